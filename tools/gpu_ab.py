"""A/B timing of one freeflyer batch: python tools/gpu_ab.py [B] -- lone solve (kernel ms), with/without probe."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
import gusto_jl_amd as g
P = g.problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, glo, ghi, tf = P.freeflyer_batch(B)
env = P.freeflyer_env()
for probe in (2, 0):
    s = g.BatchSolver(g.FREEFLYER_SE2, 50, B, hist_cap=64, boxes=env)
    s.set_schedule(probe, 1)
    ms = []
    for rep in range(4):
        s.set_problems(x0, glo, ghi, tf); s.solve(30); ms.append(s.last_solve_ms())
    st = s.status()
    X, U = s.traj()
    print(f"B={B} probe={probe}: kernel ms {['%.1f' % v for v in ms]} conv {st['converged'].sum()} ipm {st['ipm_iters'].sum()} "
          f"traj/s {st['converged'].sum()/(min(ms)/1e3):.0f} checksum {float(np.abs(X).sum()):.12g}")
