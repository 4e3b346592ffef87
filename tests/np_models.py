"""Independent numpy restatement of the convex subproblem of ONE GuSTO trip for dubins_car, astrobeeSE3 and
astrobeeSE3manifold, written straight from the reference's model files -- NOT through the oracle's row list:

  dynamics f            dubins_car.jl:161-165, astrobee_se3.jl:180-190 (+ quat_functions.jl:253-257),
                        astrobee_se3_manifold.jl:231-246
  Jacobians A, B        complex-step derivatives of f (no hand-written Jacobian table is shared with the oracle)
  trapezoid rows        dubins_car.jl:134-146, astrobee_se3_manifold.jl:169-195
  constraint registry   dubins_car.jl:184-226, astrobee_se3.jl:322-379, astrobee_se3_manifold.jl:533-608
  row functions         dynamics.jl:56-81, astrobee_se3.jl:244-263,282-305,308-311, astrobee_se3_manifold.jl:308-340,481-504
  penalisation          scp_gusto.jl:253-314 (incl. the +-eps pair of the convex_state_eq category, :297-311)
  signed distance       sphere vs AABB / sphere vs sphere in 3-D (stands in for BulletCollision.distance)

The subproblem is handed to scipy SLSQP in its full slack form (variables X, U and one slack per penalised row)."""
import numpy as np
import scipy.optimize as so

PI = np.pi


# ---- robots / models (constants from robot/astrobee3D.jl:15-33, dubins_car.jl:22-33) ------------------------------
class Dubins:
    n, m = 3, 1
    v, k = 2.0, 1.0
    x_max, u_max = np.array([100.0, 100.0, 2 * PI]), 10.0
    Delta0, eps, clearance = 1e4, 1e-6, 0.01
    has_tr = False

    @staticmethod
    def f(x, u):
        return np.array([Dubins.v * np.cos(x[2]), Dubins.v * np.sin(x[2]), Dubins.k * u[0]])


class Astrobee:
    mass, J = 7.0, 0.1083
    r = np.sqrt(3.0) * 0.5 * 0.305
    v_max, a_max, w_max, al_max = 0.5, 0.1, 45 * PI / 180, 50 * PI / 180
    clearance = 0.03


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


class AstrobeeSE3(Astrobee):
    n, m = 12, 6
    Delta0, eps = 10.0, 1e-6
    has_tr = True

    @staticmethod
    def f(x, u):
        v, p, w = x[3:6], x[6:9], x[9:12]
        F, M = u[0:3], u[3:6]
        pd = 0.25 * ((1 - np.sum(p * p)) * w - 2 * _cross(w, p) + 2 * np.sum(w * p) * p)      # mrp_derivative
        wd = (M - _cross(w, Astrobee.J * w)) / Astrobee.J
        return np.concatenate([v, F / Astrobee.mass, pd, wd])


class AstrobeeSE3Manifold(Astrobee):
    n, m = 13, 6
    Delta0, eps = 1e3, 1e-1
    has_tr = False

    @staticmethod
    def f(x, u):
        v = x[3:6]
        qw, qx, qy, qz = x[6:10]
        wx, wy, wz = x[10:13]
        F, M = u[0:3], u[3:6]
        qd = 0.5 * np.array([-wx * qx - wy * qy - wz * qz, wx * qw - wz * qy + wy * qz, wy * qw + wz * qx - wx * qz,
                             wz * qw - wy * qx + wx * qy])
        w = x[10:13]
        wd = (M - _cross(w, Astrobee.J * w)) / Astrobee.J
        return np.concatenate([v, F / Astrobee.mass, qd, wd])


def jac(model, x, u):
    """complex-step Jacobians of f"""
    n, m, h = model.n, model.m, 1e-30
    A, B = np.zeros((n, n)), np.zeros((n, m))
    for j in range(n):
        xc = x.astype(complex); xc[j] += 1j * h
        A[:, j] = model.f(xc, u.astype(complex)).imag / h
    for j in range(m):
        uc = u.astype(complex); uc[j] += 1j * h
        B[:, j] = model.f(x.astype(complex), uc).imag / h
    return A, B


# ---- signed distance of a sphere of radius r centred at c (3-D) -----------------------------------------------------
def sd_box(c, lo, hi, r):
    e = np.where(c < lo, c - lo, np.where(c > hi, c - hi, 0.0))
    if np.any(e != 0):
        d = np.linalg.norm(e)
        return d - r, e / d
    a, b = c - lo, hi - c          # inside: nearest face
    i = int(np.argmin(np.concatenate([a, b])))
    nh = np.zeros(3)
    if i < 3:
        nh[i] = -1.0
        return -a[i] - r, nh
    nh[i - 3] = 1.0
    return -b[i - 3] - r, nh


def sd_sphere(c, cs, rs, r):
    v = c - cs
    d = np.linalg.norm(v)
    return d - rs - r, v / d


def solve_subproblem(model, N, tf, x_init, goal_lo, goal_hi, Xp, Up, Delta, omega, boxes=(), spheres=(), maxiter=400):
    """One trip's convex subproblem (scp_gusto.jl:178-314) by SLSQP.  Returns X, U, objective (unscaled)."""
    n, m = model.n, model.m
    nz = n + m
    dt = tf / (N - 1)
    kappa = 1.0 / max(1.0, omega)
    toggle = Delta / 8 + model.clearance
    nzN = nz * N
    ix = lambda k, i: nz * k + i
    iu = lambda k, i: nz * k + n + i

    # ---- equality rows: init, trapezoid collocation, goal points --------------------------------------------------
    rows, rhs = [], []
    for i in range(n):
        R = np.zeros(nzN); R[ix(0, i)] = 1.0
        rows.append(R); rhs.append(x_init[i])
    lin = [(model.f(Xp[k], Up[k]),) + jac(model, Xp[k], Up[k]) for k in range(N)]
    for k in range(1, N):
        (f0, A0, B0), (f1, A1, B1) = lin[k - 1], lin[k]
        R = np.zeros((n, nzN))
        R[:, nz * (k - 1):nz * (k - 1) + n] = np.eye(n) + 0.5 * dt * A0
        R[:, nz * (k - 1) + n:nz * k] = 0.5 * dt * B0
        R[:, nz * k:nz * k + n] = -np.eye(n) + 0.5 * dt * A1
        R[:, nz * k + n:nz * (k + 1)] = 0.5 * dt * B1
        c = 0.5 * dt * (f0 - A0 @ Xp[k - 1] - B0 @ Up[k - 1] + f1 - A1 @ Xp[k] - B1 @ Up[k])
        rows.extend(R); rhs.extend(-c)
    for i in range(n):
        if goal_lo[i] == goal_hi[i]:
            R = np.zeros(nzN); R[ix(N - 1, i)] = 1.0
            rows.append(R); rhs.append(goal_lo[i])
    E, e = np.array(rows), np.array(rhs, float)

    # ---- inequality rows as (value, gradient) callables of z; hard: g <= 0; penalised: kappa (w g - off) <= s ------
    hard, pen = [], []          # entries: (fun(z) -> (val, sparse grad dict), weight, off)

    def quad(idx, coef, c0):     # sum coef_i z_i^2 + c0
        idx, coef = np.array(idx), np.array(coef, float)
        return lambda z: (float(np.sum(coef * z[idx] ** 2) + c0), (idx, 2 * coef * z[idx]))

    def lin_(idx, coef, c0):
        idx, coef = np.array(idx), np.array(coef, float)
        return lambda z: (float(np.sum(coef * z[idx]) + c0), (idx, coef))

    def quad_about(idx, ctr):    # sum (z_i - ctr_i)^2
        idx, ctr = np.array(idx), np.array(ctr, float)
        return lambda z: (float(np.sum((z[idx] - ctr) ** 2)), (idx, 2 * (z[idx] - ctr)))

    for k in range(N):
        if model is Dubins:
            for i in range(n):      # csi_max/min_bound_constraints (dynamics.jl:56-64): penalised
                pen.append((lin_([ix(k, i)], [1.0], -model.x_max[i]), omega, 0.0))
            for i in range(n):
                pen.append((lin_([ix(k, i)], [-1.0], -model.x_max[i]), omega, 0.0))
            if k < N - 1:           # cci_max/min_bound_constraints (dynamics.jl:73-81): hard, k = 1..N-1
                hard.append(lin_([iu(k, 0)], [1.0], -model.u_max))
                hard.append(lin_([iu(k, 0)], [-1.0], -model.u_max))
            continue
        man = model is AstrobeeSE3Manifold
        iw = 10 if man else 9
        if model.has_tr:            # stri_state_trust_region: omega * ||x - xp||^2 - Delta <= s
            pen.append((quad_about([ix(k, i) for i in range(n)], Xp[k]), omega, Delta))
        if man:
            qp = Xp[k, 6:10]
            qn = np.linalg.norm(qp)
            # cse_quaternion_norm h = |qp| + qp.(q - qp)/|qp| - 1, penalised as the +-eps pair (scp_gusto.jl:297-311):
            #   j = 1:  -w h - eps <= -s1   (s1 >= 0 is minimised, so this is the HARD bound w h + eps >= 0)
            #   j = 2:   w h - eps <=  s2   (the L1 penalty on h > eps / w)
            hfun = lin_([ix(k, 6 + j) for j in range(4)], qp / qn, qn - np.sum(qp * qp) / qn - 1.0)
            hard.append((lambda z, hf=hfun: (lambda v, g: (-(omega * v) - model.eps, (g[0], -omega * g[1])))(*hf(z))))
            pen.append((hfun, omega, model.eps))
            pen.append((lin_([ix(k, 6)], [-1.0], 0.0), omega, 0.0))           # csi_orientation_sign: -qw
        pen.append((quad([ix(k, 3 + j) for j in range(3)], [1.0] * 3, -model.v_max ** 2), omega, 0.0))
        pen.append((quad([ix(k, iw + j) for j in range(3)], [1.0] * 3, -model.w_max ** 2), omega, 0.0))
        r0 = Xp[k, 0:3]
        comps = [("b", b) for b in boxes] + [("s", s) for s in spheres]
        for kind, o in comps:       # ncsi_obstacle_avoidance_signed_distance_convexified
            d, nh = sd_box(r0, o[0:3], o[3:6], model.r) if kind == "b" else sd_sphere(r0, o[0:3], o[3], model.r)
            if d < toggle:
                pen.append((lin_([ix(k, j) for j in range(3)], -nh, model.clearance - d + nh @ r0), omega, 0.0))
        if k < N - 1:               # cci_translational/angular_accel_bound: hard, k = 1..N-1
            hard.append(quad([iu(k, j) for j in range(3)], [1 / model.mass ** 2] * 3, -model.a_max ** 2))
            hard.append(quad([iu(k, 3 + j) for j in range(3)], [1 / model.J ** 2] * 3, -model.al_max ** 2))
    for i in range(n):              # csbci_goal_constraints (BoxGoal): hard
        if goal_lo[i] != goal_hi[i]:
            if np.isfinite(goal_hi[i]):
                hard.append(lin_([ix(N - 1, i)], [1.0], -goal_hi[i]))
            if np.isfinite(goal_lo[i]):
                hard.append(lin_([ix(N - 1, i)], [-1.0], goal_lo[i]))

    ns = len(pen)
    w = np.full(N, dt); w[0] = w[-1] = 0.5 * dt
    uidx = np.array([[iu(k, j) for j in range(m)] for k in range(N)])

    def obj(z):
        return kappa * float(np.sum(w[:, None] * z[uidx] ** 2)) + float(z[nzN:].sum())

    def obj_grad(z):
        g = np.zeros_like(z)
        g[uidx] = 2 * kappa * w[:, None] * z[uidx]
        g[nzN:] = 1.0
        return g

    def ineq(z):                    # >= 0
        out = [z[nzN + j] - kappa * (wt * fn(z)[0] - off) for j, (fn, wt, off) in enumerate(pen)]
        out += [-fn(z)[0] for fn in hard]
        return np.array(out)

    def ineq_jac(z):
        Jm = np.zeros((ns + len(hard), len(z)))
        for j, (fn, wt, off) in enumerate(pen):
            idx, gr = fn(z)[1]
            Jm[j, idx] = -kappa * wt * gr
            Jm[j, nzN + j] = 1.0
        for j, fn in enumerate(hard):
            idx, gr = fn(z)[1]
            Jm[ns + j, idx] = -gr
        return Jm

    z0 = np.concatenate([np.hstack([Xp, Up]).ravel(), np.ones(ns)])
    Epad = np.hstack([E, np.zeros((E.shape[0], ns))])
    res = so.minimize(obj, z0, jac=obj_grad, method="SLSQP",
                      constraints=[{"type": "eq", "fun": lambda z: Epad @ z - e, "jac": lambda z: Epad},
                                   {"type": "ineq", "fun": ineq, "jac": ineq_jac}],
                      bounds=[(None, None)] * nzN + [(0, None)] * ns, options={"ftol": 1e-15, "maxiter": maxiter})
    Z = res.x[:nzN].reshape(N, nz)
    return dict(X=Z[:, :n], U=Z[:, n:], obj=res.fun / kappa, res=res, n_pen=ns, n_hard=len(hard),
                eq_violation=float(np.abs(Epad @ res.x - e).max()), ineq_min=float(ineq(res.x).min()))
