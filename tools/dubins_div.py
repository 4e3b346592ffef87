"""The dubins_car problems of profiles/r04_parity_sweep.txt whose SCP iteration counts differ between HIP and oracle:
whole solves on both sides, the first history entry that differs, both solver statuses there.  python tools/dubins_div.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
IDX = [351, 662, 1659, 2304, 3155, 3252, 3780, 3792, 5064, 6056, 6158, 6745]
x0, glo, ghi, tf = P.dubins_batch(8192, first=20000)
x0, glo, ghi, tf = x0[IDX], glo[IDX], ghi[IDX], tf[IDX]
s = g.BatchSolver(g.DUBINS_CAR, 30, len(IDX), hist_cap=64)
s.set_problems(x0, glo, ghi, tf); s.solve(30)
st, h = s.status(), s.history()
o = go.Oracle(go.DUBINS_CAR, 30)
for j, b in enumerate(IDX):
    o.set_problem(x0[j], glo[j], ghi[j], tf[j])
    r = o.solve(30)
    nh = int(h["n_hist"][j]); no = len(r["scp_status"])
    first = None
    for t in range(1, min(nh, no)):
        same = (h["scp_status"][j, t] == r["scp_status"][t] and h["accept_solution"][j, t] == r["accept"][t] and
                h["solver_status"][j, t] == r["solver_status"][t] and h["Delta"][j, t] == r["Delta"][t] and h["omega"][j, t] == r["omega"][t])
        if not same:
            first = t; break
    if first is None: first = min(nh, no)
    t = first
    def row(src, t, dev):
        if dev:
            if t >= nh: return "-"
            return f"sol {h['solver_status'][j,t]} scp {h['scp_status'][j,t]} acc {h['accept_solution'][j,t]} ipm {h['ipm_iters'][j,t]} conv {h['convergence_measure'][j,t]:.3e} rho {h['rho'][j,min(t+1, h['n_rho'][j]-1)]:.3e} D {h['Delta'][j,t]:g} w {h['omega'][j,t]:g} tr {h['trust_region_satisfied'][j,t]} cvx {h['convex_ineq_satisfied'][j,t]}"
        if t >= no: return "-"
        return f"sol {r['solver_status'][t]} scp {r['scp_status'][t]} acc {r['accept'][t]} ipm {r['ipm_iters'][t]} conv {r['conv'][t]:.3e} D {r['Delta'][t]:g} w {r['omega'][t]:g} tr {r['tr_sat'][t]} cvx {r['cvx_sat'][t]}"
    print(f"problem {b}: gpu iters {st['iterations'][j]} stop {st['stop_reason'][j]} conv {int(st['converged'][j])} | oracle iters {r['iterations']} stop {r['stop_reason']} conv {int(r['converged'])} | first differing entry {t}")
    print("   gpu   :", row(None, t, True)); print("   oracle:", row(None, t, False))
    if t > 1: print("   before: gpu", row(None, t - 1, True), "\n           ora", row(None, t - 1, False))
