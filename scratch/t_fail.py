import sys
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo')
import numpy as np, gusto_oracle as go, gusto_jl_amd as g
P=g.problems; env=P.freeflyer_env(); N=50
x0,glo,ghi,tf = P.freeflyer_batch(8)
x0[0]=P.FREEFLYER_X_INIT
o = go.Oracle(go.FREEFLYER_SE2, N, boxes=env)
for b in (7,6,5,2):
    o.set_problem(x0[b],glo[b],ghi[b],tf[b])
    r=o.solve(30)
    print(b, x0[b][:2], 'iters',r['iterations'],'stop',go.STOP_REASON[r['stop_reason']], 'omega',r['omega'],'Delta',r['Delta'],'ipm',r['ipm_iters'], [go.SCP_STATUS[s] for s in r['scp_status']])
    X,U = r['X'],r['U']
    D,w = r['Delta'][-1], r['omega'][-1]
    rs = o.subproblem(X,U,D,w,D/8+0.05)
    print('   sub:', {k:v for k,v in rs.items() if k not in ('X','U','dual')})
