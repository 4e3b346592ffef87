"""Stop reasons / solver statuses of the TrajOpt batches: python tools/to_stops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
for model, B in ((g.FREEFLYER_SE2, 1024), (g.ASTROBEE_SE3, 256), (g.ASTROBEE_SE3_MANIFOLD, 128)):
    if model == g.FREEFLYER_SE2: batch, boxes, spheres = P.freeflyer_batch(B), P.freeflyer_env(), None
    elif model == g.ASTROBEE_SE3: batch = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
    else: batch = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True)
    s = g.TrajOptSolver(model, 50, B, boxes=boxes, spheres=spheres)
    s.set_problems(*batch); s.solve(125)
    st, h = s.status(), s.history()
    it = st["iterations"]
    print(f"model {model} B={B}: stop reasons (MaxIter, Converged, SubproblemFailed, -, HistFull) {np.bincount(st['stop_reason'], minlength=5).tolist()}, "
          f"converged {int(st['converged'].sum())}, solves per problem min/median/max {it.min()}/{int(np.median(it))}/{it.max()}, kernel {s.last_solve_ms():.1f} ms")
    failed = np.where(st["stop_reason"] == 2)[0]
    if len(failed):
        at = [int(it[b]) for b in failed]
        print("   SubproblemFailed after n solves:", np.bincount(at).tolist(), " ipm iterations of the failed solve:", [int(h["ipm_iters"][b, it[b] + 1]) for b in failed][:40])
