import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
B = int(sys.argv[1])
x0, glo, ghi, tf = P.dubins_batch(B)
for probe in (0, 2, 1):
    s = g.BatchSolver(g.DUBINS_CAR, 30, B, hist_cap=64)
    s.set_schedule(probe, 1)
    ms = []
    for rep in range(3):
        s.set_problems(x0, glo, ghi, tf); s.solve(30); ms.append(s.last_solve_ms())
    st = s.status()
    print(f"dubins B={B} probe={probe}: kernel ms {['%.1f' % v for v in ms]} conv {st['converged'].sum()} ipm {st['ipm_iters'].sum()} iters {st['iterations'].sum()}", flush=True)
