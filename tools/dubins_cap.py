import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
IDX = [351, 662, 1659, 2304, 3155, 3252, 3780, 3792, 5064, 6056, 6158, 6745]
x0, glo, ghi, tf = P.dubins_batch(8192, first=20000)
x0, glo, ghi, tf = x0[IDX], glo[IDX], ghi[IDX], tf[IDX]
for cap in (60, 150, 400):
    io = g.default_ipm_opts(); io.max_iter = cap
    s = g.BatchSolver(g.DUBINS_CAR, 30, len(IDX), hist_cap=64, ipm_opts=io)
    s.set_problems(x0, glo, ghi, tf); s.solve(30)
    st = s.status()
    oio = go.IpmOpts(tol=io.tol, tol_acc=io.tol_acc, mu_floor=io.mu_floor, tr_tol=io.tr_tol, mu_warm=io.mu_warm, max_iter=cap,
                     acc_iter=io.acc_iter, mu_warm_gain=io.mu_warm_gain, mu_warm_max=io.mu_warm_max, sigma_max=io.sigma_max)
    o = go.Oracle(go.DUBINS_CAR, 30, ipm_opts=oio)
    res = []
    for j in range(len(IDX)):
        o.set_problem(x0[j], glo[j], ghi[j], tf[j]); r = o.solve(30)
        res.append((r["iterations"], r["stop_reason"], r["total_ipm_iters"]))
    print("cap", cap)
    print("  gpu   ", list(zip(st["iterations"].tolist(), st["stop_reason"].tolist(), st["ipm_iters"].tolist())))
    print("  oracle", res)
