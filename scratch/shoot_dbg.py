import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, warnings
import gusto_jl_amd as g
P = g.problems; H = g.host
model = H.AstrobeeSE3Manifold()
x0, glo, ghi, tf = P.astrobee_manifold_batch(4)
orig = H.solve_shooting
def wrapped(SS, SP, **o):
    r = orig(SS, SP, **o)
    print("  shoot status", int(r["status"][0]), "it", int(r["newton_iters"][0]), "resid", float(r["resid"][0]), "max|p0|", np.abs(r["p0"]).max(), "max|U|", np.abs(r["U"]).max(), "seed max", np.abs(SP.p0).max())
    return r
H.solve_shooting = wrapped
for b in (0, 1):
    gs = H.GoalSet(); H.add_goal(gs, H.Goal(H.PointGoal(glo[b]), tf[b], model))
    TOP = H.TrajectoryOptimizationProblem(H.ProblemDefinition(H.Robot(), model, H.ISSCorner(True), x0[b], gs), 50, tf[b], True)
    TOS = H.TrajectoryOptimizationSolution(TOP)
    H.solve_SCPshooting(TOS, TOP, H.solve_gusto_hip, H.init_traj_straightline, "hip", max_iter=30)
    print(b, TOS.SS.prob_status, TOS.SS.converged)
