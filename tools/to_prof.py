"""Phase profile of the TrajOpt kernel (-DGUSTO_PROFILE dev build of model 4 or 5): python tools/to_prof.py <model 0|2> <B>"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, gusto_jl_amd as g
P = g.problems
model, B = int(sys.argv[1]), int(sys.argv[2])
if model == 0: batch, boxes, spheres = P.freeflyer_batch(B), P.freeflyer_env(), None
else:
    batch = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
s = g.TrajOptSolver(model, 50, B, boxes=boxes, spheres=spheres)
for _ in range(2):
    s.set_problems(*batch); s.solve(125)
st = s.status(); print("ms", s.last_solve_ms(), "ipm", st["ipm_iters"].sum(), "solves", st["iterations"].sum())
prof = np.zeros((B, 48), dtype=np.int64)
s.L.gusto_dev_get_prof.argtypes = [C.c_void_p, C.c_void_p]
s.L.gusto_dev_get_prof(s.h, prof.ctypes.data)
tot = prof.sum(axis=0).astype(float); ipm = st["ipm_iters"].sum()
names = ["RESID","BUILD","FACTOR","POSTF","RHS","BACK","MID","FWD","STEP","UPDATE","LIN","SCP","INIT","F:pre","F:ph1(T,Z)","F:sync","F:ph2(H)","F2","F3","F:ph3(chol)","F5","F6","F:ph4","F8","M:th","M:red","M:mu","M:dk","M:sync","t29","t30","t31","R:prolog","R:fixrows","R:obsrows","R:ctlrows","S:prolog","S:fixrows","S:obsrows","S:ctlrows","40","41","42","43","44","45","46","47"]
for i, nm in enumerate(names):
    if tot[i] > 0: print(f"  {nm:12s} {100*tot[i]/max(tot.sum(),1):5.1f}%  {tot[i]/ipm:9.0f}")
print("cycles per ipm iter", tot.sum()/ipm)
