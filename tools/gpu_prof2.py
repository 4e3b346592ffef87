"""Per-problem cycle totals of a -DGUSTO_PROFILE build: clock rate of the counter and the tail of the batch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
B = int(sys.argv[1]); model = int(sys.argv[2])
spheres = None
if model == 0:
    x0, glo, ghi, tf = P.freeflyer_batch(B); boxes = P.freeflyer_env()
elif model == 1:
    x0, glo, ghi, tf = P.dubins_batch(B); boxes = None
elif model == 2:
    x0, glo, ghi, tf = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
else:
    x0, glo, ghi, tf = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True)
s = g.BatchSolver(model, 30 if model == 1 else 50, B, hist_cap=64, boxes=boxes, spheres=spheres)
for rep in range(2):
    s.set_problems(x0, glo, ghi, tf); s.solve(30)
st = s.status()
prof = np.zeros((B, 48), dtype=np.int64)
s.L.gusto_dev_get_prof.argtypes = [C.c_void_p, C.c_void_p]
s.L.gusto_dev_get_prof(s.h, prof.ctypes.data)
cyc = prof.sum(axis=1); it = st["ipm_iters"]
o = np.argsort(-cyc)[:5]
ms = s.last_solve_ms()
print(f"model {model} B={B} kernel {ms:.1f} ms, total iters {it.sum()}, total cycles {cyc.sum():.4g}")
for b in o: print(f"  problem {b}: cycles {cyc[b]:.4g} iters {it[b]} cycles/iter {cyc[b]/max(1,it[b]):.4g}  -> counter rate if it ran the whole launch: {cyc[b]/ms/1e6:.3f} GHz")
print("  iters: mean %.1f max %d; cycles/iter over all %.4g" % (it.mean(), it.max(), cyc.sum()/it.sum()))
