"""Independent numpy/scipy restatement of ONE GuSTO convex subproblem (scp_gusto.jl:178-314), used to pin the C
oracle.  Variables are the full (X, U, slacks); nothing here shares code with the oracle's interior point method."""
import numpy as np


def subproblem_matrices(o, Xp, Up, x_init, goal_idx, goal_val, dt):
    """Equality rows E z = e (init, trapezoid, goal point rows) on z = [x_0,u_0,x_1,u_1,...]."""
    N, n, m = o.N, o.n, o.m
    nz = n + m
    rows, rhs = [], []
    ix = lambda k: slice(nz * k, nz * k + n)
    iu = lambda k: slice(nz * k + n, nz * (k + 1))
    E0 = np.zeros((n, nz * N)); E0[:, ix(0)] = np.eye(n)
    rows.append(E0); rhs.append(np.asarray(x_init, float))
    lin = [o.dynamics(Xp[k], Up[k]) for k in range(N)]
    for k in range(1, N):
        f0, A0, B0 = lin[k - 1]
        f1, A1, B1 = lin[k]
        R = np.zeros((n, nz * N))
        R[:, ix(k - 1)] = np.eye(n) + 0.5 * dt * A0
        R[:, iu(k - 1)] = 0.5 * dt * B0
        R[:, ix(k)] = -np.eye(n) + 0.5 * dt * A1
        R[:, iu(k)] += 0.5 * dt * B1
        c = 0.5 * dt * (f0 - A0 @ Xp[k - 1] - B0 @ Up[k - 1] + f1 - A1 @ Xp[k] - B1 @ Up[k])
        rows.append(R); rhs.append(-c)
    if len(goal_idx):
        G = np.zeros((len(goal_idx), nz * N))
        for r, i in enumerate(goal_idx):
            G[r, nz * (N - 1) + i] = 1.0
        rows.append(G); rhs.append(np.asarray(goal_val, float))
    return np.vstack(rows), np.concatenate(rhs)


def row_value_grad(r, v):
    w = v[r["idx"]]
    val = float(np.sum(r["a"] * (w - r["v0"]) ** 2) + np.sum(r["b"] * w) + r["c0"])
    g = np.zeros_like(v)
    g[r["idx"]] = 2 * r["a"] * (w - r["v0"]) + r["b"]
    return val, g


def exact_penalty_objective(o, rows, X, U, dt, kappa):
    """kappa * cost + sum_pen max(0, ghat) -- the subproblem with slacks eliminated."""
    N = o.N
    w = np.full(N, dt); w[0] = w[-1] = 0.5 * dt
    J = kappa * float(np.sum(w[:, None] * U ** 2))
    for r in rows:
        if r["kind"] in (0, 4):
            continue
        v = (U if r["isu"] else X)[r["k"]]
        val, _ = row_value_grad(r, v)
        J += max(0.0, r["mul"] * val - r["off"])
    return J
