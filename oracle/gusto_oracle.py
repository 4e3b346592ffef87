"""ctypes binding of the CPU ORACLE (oracle/libgusto_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  See oracle/gusto_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("GUSTO_ORACLE_LIB") or os.path.join(_HERE, "libgusto_oracle.so")   # (a sanitizer build: oracle/Makefile asan)

MAXN, MAXM = 13, 6
FREEFLYER_SE2, DUBINS_CAR, ASTROBEE_SE3, ASTROBEE_SE3_MANIFOLD = 0, 1, 2, 3
MODEL_DIMS = {0: (6, 3), 1: (3, 1), 2: (12, 6), 3: (13, 6)}
SCP_STATUS = {0: "NA", 1: "OK", 2: "InaccurateModel", 3: "ViolatesConstraints", 4: "TrustRegionViolated"}
SOLVER_STATUS = {0: "NA", 1: "OPTIMAL", 2: "ALMOST_LOCALLY_SOLVED", 3: "FAILED"}
STOP_REASON = {0: "MaxIter", 1: "Converged", 2: "SubproblemFailed", 3: "OmegaMaxExceeded"}


class ScpParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("Delta0", "omega0", "omega_max", "eps", "rho0", "rho1", "beta_succ", "beta_fail", "gamma_fail",
                 "convergence_threshold")]


class ModelParams(C.Structure):
    _fields_ = [("mass", C.c_double), ("Jdiag", C.c_double * 3), ("radius", C.c_double), ("clearance", C.c_double),
                ("hard_limit_vel", C.c_double), ("hard_limit_accel", C.c_double), ("hard_limit_omega", C.c_double),
                ("hard_limit_alpha", C.c_double), ("dubins_v", C.c_double), ("dubins_k", C.c_double),
                ("u_max", C.c_double), ("u_min", C.c_double), ("x_max", C.c_double * MAXN),
                ("x_min", C.c_double * MAXN), ("n_robot_comp", C.c_int), ("comp_off", (C.c_double * 3) * 2)]


class IpmOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("tol_acc", C.c_double), ("mu_floor", C.c_double), ("tr_tol", C.c_double),
                ("mu_warm", C.c_double), ("max_iter", C.c_int), ("acc_iter", C.c_int), ("mu_warm_gain", C.c_double), ("mu_warm_max", C.c_double), ("sigma_max", C.c_double)]


class DistModel(C.Structure):
    """Study knob for the freeflyer signed distance (oracle/gusto_oracle.h go_dist_model)."""
    _fields_ = [("kind", C.c_int), ("n_poly", C.c_int), ("vertical_escape", C.c_int), ("poly_phase", C.c_double),
                ("margin", C.c_double), ("z_lo", C.c_double), ("z_hi", C.c_double), ("pen_mode", C.c_int),
                ("pen_value", C.c_double), ("pen_band", C.c_double)]


class TrajOptParams(C.Structure):
    """go_trajopt_params = SCPParam_TrajOpt (scp_trajopt.jl:3-30)"""
    _fields_ = [(k, C.c_double) for k in ("mu0", "s0", "c", "tau_plus", "tau_minus", "k", "ftol", "xtol", "ctol")] + \
               [(k, C.c_int) for k in ("max_penalty_iteration", "max_convex_iteration", "max_trust_iteration")]


class SubInfo(C.Structure):
    _fields_ = [("obj", C.c_double), ("res_p", C.c_double), ("res_d", C.c_double), ("mu", C.c_double),
                ("iters", C.c_int), ("status", C.c_int)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("gusto_oracle.c", "gusto_oracle.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.go_create.restype = C.c_void_p
        L.go_create.argtypes = [C.c_int, C.c_int, C.POINTER(ScpParams), C.POINTER(ModelParams), C.c_int, C.c_void_p,
                                C.c_int, C.c_void_p]
        L.go_destroy.argtypes = [C.c_void_p]
        L.go_set_ipm_opts.argtypes = [C.c_void_p, C.POINTER(IpmOpts)]
        L.go_set_distance_model.argtypes = [C.c_void_p, C.POINTER(DistModel)]
        L.go_set_trace.argtypes = [C.c_void_p, C.c_int]
        L.go_trace_len.argtypes = [C.c_void_p]
        L.go_get_trace.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
        L.go_set_problem.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, C.c_void_p, C.c_void_p]
        L.go_solve.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.go_get_traj.argtypes = [C.c_void_p, _dp, _dp]
        L.go_get_status.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
        L.go_hist_len.argtypes = [C.c_void_p]
        L.go_get_history.argtypes = [C.c_void_p] + [C.c_void_p] * 15
        L.go_get_dual.argtypes = [C.c_void_p, _dp]
        L.go_shoot.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, C.POINTER(C.c_int),
                               C.POINTER(C.c_double)]
        L.go_subproblem.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp,
                                    C.POINTER(SubInfo)]
        L.go_default_trajopt_params.argtypes = [C.c_int, C.POINTER(TrajOptParams)]
        L.go_create_trajopt.restype = C.c_void_p
        L.go_create_trajopt.argtypes = [C.c_int, C.c_int, C.POINTER(ModelParams), C.POINTER(TrajOptParams), C.c_int, C.c_void_p,
                                        C.c_int, C.c_void_p]
        L.go_trajopt_subproblem.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, C.POINTER(SubInfo)]
        L.go_solve_trajopt.argtypes = [C.c_void_p, C.c_int]
        L.go_get_trajopt_history.argtypes = [C.c_void_p] + [C.c_void_p] * 10
        L.go_trajopt_ratio.restype = C.c_double
        L.go_trajopt_ratio.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.go_trajopt_ctol.restype = C.c_double
        L.go_trajopt_ctol.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.go_rows_count.argtypes = [C.c_void_p]
        L.go_rows_get.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 13
        L.go_dynamics.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp]
        L.go_signed_distance.restype = C.c_double
        L.go_signed_distance.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int, _dp]
        L.go_trust_region_ratio.restype = C.c_double
        L.go_trust_region_ratio.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.go_cost_true.restype = C.c_double
        L.go_cost_true.argtypes = [C.c_void_p, _dp]
        L.go_convergence_metric.restype = C.c_double
        L.go_convergence_metric.argtypes = [C.c_void_p, _dp, _dp]
        L.go_convex_ineq_satisfied.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double]
        L.go_init_straightline.argtypes = [C.c_void_p, _dp, _dp]
        L.go_solve_batch.argtypes = [C.c_int, C.c_int, C.POINTER(ScpParams), C.POINTER(ModelParams), C.c_int,
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int,
                                     _dp, _dp, _ip, _ip, _ip, _ip]
        _lib = L
    return _lib


def default_params(model):
    sp, mp = ScpParams(), ModelParams()
    lib().go_default_params(C.c_int(model), C.byref(sp), C.byref(mp))
    return sp, mp


def _arr(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Oracle:
    """One oracle problem instance (model, horizon, environment)."""

    def __init__(self, model, N, boxes=None, spheres=None, scp_params=None, model_params=None, ipm_opts=None):
        self.L = lib()
        self.model, self.N = model, N
        self.n, self.m = MODEL_DIMS[model]
        sp, mp = default_params(model)
        self.sp = scp_params or sp
        self.mp = model_params or mp
        self.boxes = _arr(boxes if boxes is not None else np.zeros((0, 6))).reshape(-1, 6)
        self.spheres = _arr(spheres if spheres is not None else np.zeros((0, 4))).reshape(-1, 4)
        self.h = self.L.go_create(model, N, C.byref(self.sp), C.byref(self.mp), len(self.boxes),
                                  self.boxes.ctypes.data, len(self.spheres), self.spheres.ctypes.data)
        if not self.h:
            raise RuntimeError("go_create failed")
        if ipm_opts is not None:
            self.L.go_set_ipm_opts(self.h, C.byref(ipm_opts))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.go_destroy(self.h)
            self.h = None

    def set_distance_model(self, **kw):
        """Study knob: kind=1 selects the prism-vs-AABB model (n_poly, vertical_escape, margin, z_lo, z_hi)."""
        dm = DistModel(kind=1, n_poly=0, vertical_escape=0, poly_phase=0.0, margin=0.0, z_lo=0.0, z_hi=1.0)
        for k, v in kw.items():
            setattr(dm, k, v)
        self.L.go_set_distance_model(self.h, C.byref(dm))

    def set_trace(self, cap):
        """Record (traj_prev, subproblem optimum) of every trip of the following solves: trace()."""
        self.L.go_set_trace(self.h, int(cap))

    def trace(self):
        """Per trip t (history index t + 1): dict(Xp, Up, Xn, Un); Delta/omega of the trip are r["Delta"][t], r["omega"][t]."""
        out = []
        for t in range(self.L.go_trace_len(self.h)):
            a = [np.zeros((self.N, self.n)), np.zeros((self.N, self.m)), np.zeros((self.N, self.n)), np.zeros((self.N, self.m))]
            self.L.go_get_trace(self.h, t, *a)
            out.append(dict(Xp=a[0], Up=a[1], Xn=a[2], Un=a[3]))
        return out

    # -- problem / solve ---------------------------------------------------------------------
    def set_problem(self, x_init, goal_lo, goal_hi, tf, X0=None, U0=None):
        self._x0 = None if X0 is None else _arr(X0)
        self._u0 = None if U0 is None else _arr(U0)
        self.L.go_set_problem(self.h, _arr(x_init), _arr(goal_lo), _arr(goal_hi), float(tf),
                              None if X0 is None else self._x0.ctypes.data,
                              None if U0 is None else self._u0.ctypes.data)

    def solve(self, max_iter=30, force=False):
        self.L.go_solve(self.h, int(max_iter), int(bool(force)))
        return self.result()

    def traj(self):
        X = np.zeros((self.N, self.n))
        U = np.zeros((self.N, self.m))
        self.L.go_get_traj(self.h, X, U)
        return X, U

    def result(self):
        X, U = self.traj()
        vals = [C.c_int() for _ in range(5)]
        self.L.go_get_status(self.h, *[C.byref(v) for v in vals])
        it, conv, succ, stop, ipm = [v.value for v in vals]
        h = self.L.go_hist_len(self.h)
        cap = 4 * h + 64
        d = {k: np.zeros(cap) for k in ("J_true", "J_full", "conv", "Delta", "omega", "rho")}
        i = {k: np.zeros(cap, dtype=np.int32) for k in ("accept", "scp_status", "solver_status", "tr_sat", "cvx_sat",
                                                        "ipm_iters")}
        nJt, nJf, nrho = C.c_int(), C.c_int(), C.c_int()
        P = lambda a: a.ctypes.data
        self.L.go_get_history(self.h, P(d["J_true"]), C.addressof(nJt), P(d["J_full"]), C.addressof(nJf), P(d["conv"]),
                              P(d["Delta"]), P(d["omega"]), P(d["rho"]), C.addressof(nrho), P(i["accept"]),
                              P(i["scp_status"]), P(i["solver_status"]), P(i["tr_sat"]), P(i["cvx_sat"]),
                              P(i["ipm_iters"]))
        out = dict(X=X, U=U, iterations=it, converged=bool(conv), successful=bool(succ), stop_reason=stop,
                   total_ipm_iters=ipm)
        out["J_true"] = d["J_true"][:nJt.value].copy()
        out["J_full"] = d["J_full"][:nJf.value].copy()
        out["rho"] = d["rho"][:nrho.value].copy()
        for k in ("conv", "Delta", "omega"):
            out[k] = d[k][:h].copy()
        for k in i:
            out[k] = i[k][:h].copy()
        dual = np.zeros(self.n)
        self.L.go_get_dual(self.h, dual)
        out["dual"] = dual
        return out

    # -- pieces ------------------------------------------------------------------------------
    def subproblem(self, Xp, Up, Delta, omega, toggle):
        Xn, Un, dual = np.zeros((self.N, self.n)), np.zeros((self.N, self.m)), np.zeros(self.n)
        info = SubInfo()
        st = self.L.go_subproblem(self.h, _arr(Xp), _arr(Up), Delta, omega, toggle, Xn, Un, dual, C.byref(info))
        return dict(X=Xn, U=Un, dual=dual, status=st, obj=info.obj, iters=info.iters, res_p=info.res_p,
                    res_d=info.res_d, mu=info.mu)

    def shoot(self, p0=None, substeps=4, max_newton=100, ftol=1e-3):
        """go_shoot: indirect shooting from p0 (default: the dual of the last SCP subproblem)."""
        pv = None if p0 is None else _arr(p0)
        p_out, X, U = np.zeros(self.n), np.zeros((self.N, self.n)), np.zeros((self.N, self.m))
        it, res = C.c_int(), C.c_double()
        st = self.L.go_shoot(self.h, None if pv is None else pv.ctypes.data, substeps, max_newton, ftol, p_out, X, U,
                             C.byref(it), C.byref(res))
        return dict(status=st, p0=p_out, X=X, U=U, newton_iters=it.value, resid=res.value)

    def rows(self):
        out = []
        for i in range(self.L.go_rows_count(self.h)):
            k, isu, kind, nnz = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            idx = np.zeros(MAXN, dtype=np.int32)
            a, v0, b = np.zeros(MAXN), np.zeros(MAXN), np.zeros(MAXN)
            c0, mul, off, sl, lam = (C.c_double() for _ in range(5))
            A = C.addressof
            self.L.go_rows_get(self.h, i, A(k), A(isu), A(kind), A(nnz), idx.ctypes.data, a.ctypes.data,
                               v0.ctypes.data, b.ctypes.data, A(c0), A(mul), A(off), A(sl), A(lam))
            z = nnz.value
            out.append(dict(k=k.value, isu=isu.value, kind=kind.value, idx=idx[:z].copy(), a=a[:z].copy(),
                            v0=v0[:z].copy(), b=b[:z].copy(), c0=c0.value, mul=mul.value, off=off.value,
                            slack=sl.value, lam=lam.value))
        return out

    def dynamics(self, x, u):
        f, A, B = np.zeros(self.n), np.zeros((self.n, self.n)), np.zeros((self.n, self.m))
        self.L.go_dynamics(self.h, _arr(x), _arr(u), f, A, B)
        return f, A, B

    def signed_distance(self, comp, r, i):
        nh = np.zeros(3)
        r3 = np.zeros(3)
        r3[:len(r)] = r
        d = self.L.go_signed_distance(self.h, comp, r3, i, nh)
        return d, nh

    def trust_region_ratio(self, X, U, Xp, Up):
        return self.L.go_trust_region_ratio(self.h, _arr(X), _arr(U), _arr(Xp), _arr(Up))

    def cost_true(self, U):
        return self.L.go_cost_true(self.h, _arr(U))

    def convergence_metric(self, X, Xp):
        return self.L.go_convergence_metric(self.h, _arr(X), _arr(Xp))

    def convex_ineq_satisfied(self, X, Xp, Up, toggle):
        return bool(self.L.go_convex_ineq_satisfied(self.h, _arr(X), _arr(Xp), _arr(Up), toggle))

    def init_straightline(self):
        X, U = np.zeros((self.N, self.n)), np.zeros((self.N, self.m))
        self.L.go_init_straightline(self.h, X, U)
        return X, U


def default_trajopt_params(model):
    tp = TrajOptParams()
    lib().go_default_trajopt_params(C.c_int(model), C.byref(tp))
    return tp


class OracleTrajOpt(Oracle):
    """One TrajOpt problem instance (src/scp/scp_trajopt.jl).  Internally a knot carries (u_k, d_k), d_k = the defect of the
    interval (k, k+1) (the L1-penalised dynamics): `self.m` is m0 + n, and results come back split into U [N, m0] and D [N, n]."""

    def __init__(self, model, N, boxes=None, spheres=None, model_params=None, trajopt_params=None, ipm_opts=None):
        self.L = lib()
        self.model, self.N = model, N
        self.n, self.m0 = MODEL_DIMS[model]
        self.m = self.m0 + self.n
        sp, mp = default_params(model)
        self.sp, self.mp = sp, model_params or mp
        self.tp = trajopt_params or default_trajopt_params(model)
        self.boxes = _arr(boxes if boxes is not None else np.zeros((0, 6))).reshape(-1, 6)
        self.spheres = _arr(spheres if spheres is not None else np.zeros((0, 4))).reshape(-1, 4)
        self.h = self.L.go_create_trajopt(model, N, C.byref(self.mp), C.byref(self.tp), len(self.boxes), self.boxes.ctypes.data,
                                          len(self.spheres), self.spheres.ctypes.data)
        if not self.h:
            raise RuntimeError("go_create_trajopt failed (FreeflyerSE2 and AstrobeeSE3 have a TrajOpt variant)")
        if ipm_opts is not None:
            self.L.go_set_ipm_opts(self.h, C.byref(ipm_opts))

    def ext(self, U, D=None):
        """[N, m0] (+ defects [N, n], default 0) -> the library's [N, m0 + n]"""
        Ue = np.zeros((self.N, self.m))
        Ue[:, :self.m0] = U
        if D is not None:
            Ue[:, self.m0:] = D
        return Ue

    def set_problem(self, x_init, goal_lo, goal_hi, tf, X0=None, U0=None):
        super().set_problem(x_init, goal_lo, goal_hi, tf, X0, None if U0 is None else self.ext(U0))

    def init_straightline(self):
        X, Ue = super().init_straightline()
        return X, Ue[:, :self.m0].copy()

    def subproblem(self, Xp, Up, mu, s, Dp=None):
        Xn, Un, dual = np.zeros((self.N, self.n)), np.zeros((self.N, self.m)), np.zeros(self.n)
        info = SubInfo()
        st = self.L.go_trajopt_subproblem(self.h, _arr(Xp), self.ext(Up, Dp), float(mu), float(s), Xn, Un, dual, C.byref(info))
        return dict(X=Xn, U=Un[:, :self.m0].copy(), D=Un[:, self.m0:].copy(), dual=dual, status=st, obj=info.obj, iters=info.iters,
                    res_p=info.res_p, res_d=info.res_d, mu=info.mu)

    def ratio(self, X, U, Xp, Up):
        return self.L.go_trajopt_ratio(self.h, _arr(X), self.ext(U), _arr(Xp), self.ext(Up))

    def ctol(self, X, U, Xp, Up):
        return self.L.go_trajopt_ctol(self.h, _arr(X), self.ext(U), _arr(Xp), self.ext(Up))

    def solve_trajopt(self, max_iter=125):
        solves = self.L.go_solve_trajopt(self.h, int(max_iter))
        r = self.result()
        r["D"] = r["U"][:, self.m0:].copy()
        r["U"] = r["U"][:, :self.m0].copy()
        r["solves"] = solves
        cap = 4 * max(1, solves) + 64
        v = {k: np.zeros(cap) for k in ("s_vec", "mu_vec", "xtol_vec", "ftol_vec", "ctol_vec")}
        n = {k: C.c_int() for k in v}
        args = []
        for k in ("s_vec", "mu_vec", "xtol_vec", "ftol_vec", "ctol_vec"):
            args += [v[k].ctypes.data, C.addressof(n[k])]
        self.L.go_get_trajopt_history(self.h, *args)
        for k in v:
            r[k] = v[k][:n[k].value].copy()
        r["rho_vec"] = r["rho"]
        return r


def solve_batch(model, N, boxes, spheres, x_init, goal_lo, goal_hi, tf, max_iter=30, nthreads=0, scp_params=None,
                model_params=None):
    """CPU baseline: B independent problems through go_solve_batch (OpenMP over problems)."""
    L = lib()
    n, m = MODEL_DIMS[model]
    sp, mp = default_params(model)
    sp, mp = scp_params or sp, model_params or mp
    boxes = _arr(boxes if boxes is not None else np.zeros((0, 6))).reshape(-1, 6)
    spheres = _arr(spheres if spheres is not None else np.zeros((0, 4))).reshape(-1, 4)
    x_init, goal_lo, goal_hi, tf = _arr(x_init), _arr(goal_lo), _arr(goal_hi), _arr(tf)
    B = x_init.shape[0]
    X, U = np.zeros((B, N, n)), np.zeros((B, N, m))
    conv, succ, its, ipm = (np.zeros(B, dtype=np.int32) for _ in range(4))
    L.go_solve_batch(model, N, C.byref(sp), C.byref(mp), len(boxes), boxes.ctypes.data, len(spheres),
                     spheres.ctypes.data, B, x_init, goal_lo, goal_hi, tf, max_iter, nthreads, X, U, conv, succ, its,
                     ipm)
    return dict(X=X, U=U, converged=conv.astype(bool), successful=succ.astype(bool), iterations=its, ipm_iters=ipm)
