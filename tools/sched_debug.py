import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
env = P.freeflyer_env()
B, probe, it = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x0, glo, ghi, tf = P.freeflyer_batch(B)
s = g.BatchSolver(g.FREEFLYER_SE2, 50, B, hist_cap=64, boxes=env)
s.set_schedule(probe, 1)
s.set_problems(x0, glo, ghi, tf)
t = time.time(); s.solve(it); dt = time.time() - t
st = s.status()
print(f"B={B} probe={probe} max_iter={it}: wall {dt*1e3:.1f} ms kernel {s.last_solve_ms():.1f} ms conv {st['converged'].sum()} stops {np.bincount(st['stop_reason'], minlength=5)} iters {st['iterations'].sum()}", flush=True)
