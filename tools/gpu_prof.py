"""Phase-cycle breakdown of the SCP kernel (needs a -DGUSTO_PROFILE build)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = int(sys.argv[2]) if len(sys.argv) > 2 else 0
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
print(f"B={B} kernel {s.last_solve_ms():.1f} ms conv {st['converged'].sum()} traj/s {st['converged'].sum()/(s.last_solve_ms()/1e3):.0f} ipm total {st['ipm_iters'].sum()}")
prof = np.zeros((B, 48), dtype=np.int64)
s.L.gusto_dev_get_prof.argtypes = [C.c_void_p, C.c_void_p]
rc = s.L.gusto_dev_get_prof(s.h, prof.ctypes.data)
names = ["RESID","BUILD","FACTOR","POSTF","RHS","BACK","MID","FWD","STEP","UPDATE","LIN","SCP","INIT","F:pre","F:AB","F:CD","F1:H","F2:Z","F3:r","F4:chol","F5:ld","F6:solve","F7:store","F8","M:th","M:reduce","M:mu+sync","M:dk/ct","M:sync2","H1:sweep","H1:back","H1:fold"]
names += ["R:prolog", "R:fixrows", "R:obsrows", "R:ctlrows", "S:prolog", "S:fixrows", "S:obsrows", "S:ctlrows",
          "FX:lds-ops", "FX:T", "FX:H,Z", "FX:Linv-lds", "FX:W,V", "FX:P'..S", "FX:Phicl", "-"]
names[0] = "R:stagecost"; names[8] = "S:tail+red"; names[16] = "F1(+S:tail p0)"
tot = prof.sum(axis=0).astype(float)
ipm = st["ipm_iters"].sum()
if model in (0, 1):   # (no matrix-core factor sweep: slots 40..44 are the per-trip stamps of scp.hpp)
    names[40:45] = ["T:obj", "T:rowcheck", "T:reduce", "T:rho", "T:accept"]
print("phase: share, cycles per IPM iteration")
for i, nm in enumerate(names):
    print(f"  {nm:7s} {100*tot[i]/tot.sum():5.1f}%  {tot[i]/ipm:9.0f}")
print("total cycles/ipm-iter", tot.sum()/ipm, " mean cycles per problem", tot.sum()/B)
