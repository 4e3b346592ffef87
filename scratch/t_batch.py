import sys, time
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo')
import numpy as np, gusto_oracle as go, gusto_jl_amd as g
P=g.problems; env=P.freeflyer_env(); N=50
B=int(sys.argv[1])
x0,glo,ghi,tf = P.freeflyer_batch(B)
t0=time.time()
r = go.solve_batch(go.FREEFLYER_SE2, N, env, None, x0, glo, ghi, tf, 30, 8)
dt=time.time()-t0
print('time',dt,'traj/s (8 thr)',B/dt,'converged',r['converged'].sum(),'successful',r['successful'].sum())
print('iters hist', np.bincount(r['iterations'],minlength=31))
print('mean ipm/iter', r['ipm_iters'].sum()/r['iterations'].sum(), 'mean ipm', r['ipm_iters'].mean())
# stop reasons through single solves for non-converged
o = go.Oracle(go.FREEFLYER_SE2, N, boxes=env)
bad = np.where(~r['converged'])[0]
stops={}
for b in bad[:200]:
    o.set_problem(x0[b],glo[b],ghi[b],tf[b]); rr=o.solve(30)
    stops[go.STOP_REASON[rr['stop_reason']]] = stops.get(go.STOP_REASON[rr['stop_reason']],0)+1
    if rr['stop_reason']==2: print('FAILED problem',b, rr['omega'][-1], rr['ipm_iters'][-3:])
print(stops)
