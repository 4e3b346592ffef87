import sys; sys.path.insert(0,'.')
import gusto_jl_amd as g
P=g.problems
for model,N,B in ((int(sys.argv[1]) if len(sys.argv) > 1 else 1, 30 if (len(sys.argv) < 2 or sys.argv[1] == "1") else 50, 4096),):
    if model==1: batch=P.dubins_batch(B); boxes=None
    else: batch=P.freeflyer_batch(B); boxes=P.freeflyer_env()
    s=g.BatchSolver(model,N,B,hist_cap=64,boxes=boxes); s.set_problems(*batch); s.solve(30); print(model, s.launch_info())
