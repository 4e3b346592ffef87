"""GPU: worst lock-step trips of a freeflyer batch (diagnostics for tests/test_gpu_parity.py::_lockstep_parity)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gusto_jl_amd as g, gusto_oracle as go
import test_gpu_parity as T
P = g.problems
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
env = P.freeflyer_env()
x0, glo, ghi, tf = P.freeflyer_batch(64, first=first)
runs = T._oracle_runs(g.FREEFLYER_SE2, 50, env, None, x0, glo, ghi, tf, 30)
trips = [(b, t) for b, (r, tr) in enumerate(runs) for t in range(len(tr))]
bi = np.array([b for b, _ in trips]); Tn = len(trips)
Xp = np.stack([runs[b][1][t]["Xp"] for b, t in trips]); Up = np.stack([runs[b][1][t]["Up"] for b, t in trips])
Xc = np.stack([runs[b][1][t]["Xc"] for b, t in trips]); Uc = np.stack([runs[b][1][t]["Uc"] for b, t in trips])
D = np.array([runs[b][0]["Delta"][t] for b, t in trips]); W = np.array([runs[b][0]["omega"][t] for b, t in trips])
s = g.BatchSolver(g.FREEFLYER_SE2, 50, Tn, hist_cap=8, boxes=env)
s.set_problems(x0[bi], glo[bi], ghi[bi], tf[bi])
sub = s.subproblem(Xp, Up, D, W, D / 8 + 0.05)
w = np.maximum(1, W)
ex = np.abs(sub["X"] - Xc).reshape(Tn, -1).max(1); eu = np.abs(sub["U"] - Uc).reshape(Tn, -1).max(1)
o = go.Oracle(go.FREEFLYER_SE2, 50, boxes=env)
for i in np.argsort(-ex / w)[:12]:
    b, t = trips[i]
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    c = o.subproblem(Xp[i], Up[i], D[i], W[i], D[i] / 8 + 0.05)
    print(f"trip {trips[i]} omega {W[i]:g} Delta {D[i]:g}  ex {ex[i]:.2e} ex/w {ex[i]/w[i]:.2e} eu {eu[i]:.2e} | dev st {sub['status'][i]} it {sub['iters'][i]} obj {sub['obj'][i]:.12g}"
          f" | orc st {c['status']} it {c['iters']} obj {c['obj']:.12g} res_p {c['res_p']:.1e} res_d {c['res_d']:.1e} mu {c['mu']:.1e}")
print("quantiles ex/w", np.quantile(ex / w, [0.5, 0.9, 0.99, 1.0]))
