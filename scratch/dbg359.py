import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gusto_jl_amd as g, gusto_oracle as go
P = g.problems
B=1024
env = P.freeflyer_env(); x0, glo, ghi, tf = P.freeflyer_batch(B, first=20000)
s = g.BatchSolver(0, 50, B, hist_cap=64, boxes=env)
s.set_problems(x0, glo, ghi, tf); s.solve(30)
st = s.status(); h = s.history()
b=359
print({k:(v[b] if hasattr(v,'__len__') else v) for k,v in st.items()})
for k in ("Delta","omega","accept","scp_status","solver_status","rho","J_true","convergence_measure","ipm_iters"):
    if k in h: print(k, np.asarray(h[k][b])[:10])
o = go.Oracle(0, 50, boxes=env); o.set_problem(x0[b], glo[b], ghi[b], tf[b]); r=o.solve(30)
print("oracle", r["iterations"], r["converged"], r.get("solver_status"), r.get("omega")[:8] if "omega" in r else None)
