import sys, time
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/scratch')
import numpy as np
import gusto_oracle as go
from proto import table_env, straight, ipm, CLR
env = table_env()
N=50
o = go.Oracle(go.FREEFLYER_SE2, N, boxes=env)
x_init = np.array([0.2,2.4,0,0,0,0]); x_goal = np.array([3.,0.5,0,0.05,-0.05,0])
o.set_problem(x_init, x_goal, x_goal, 200.)
Xp,Up = o.init_straightline()
Xs,Us = straight(x_init,x_goal,N)
print('straight diff', np.abs(Xp-Xs).max())
r = o.subproblem(Xp,Up,3.,1.,3/8+CLR)
print({k:v for k,v in r.items() if k not in('X','U')})
rp = ipm(Xp,Up,x_init,np.arange(6),x_goal,N,200./49,3.,1.,env,3/8+CLR,method='dense')
print('vs proto: X',np.abs(r['X']-rp['X']).max(),'U',np.abs(r['U']-rp['U']).max(), 'obj', r['obj'], rp['obj'])
t0=time.time()
res = o.solve(30)
print('time',time.time()-t0)
for k in ('iterations','converged','successful','stop_reason','total_ipm_iters'): print(k,res[k])
print('J_true',res['J_true']); print('conv',res['conv']); print('omega',res['omega']); print('Delta',res['Delta']); print('rho',res['rho'])
print('status',[go.SCP_STATUS[s] for s in res['scp_status']]); print('ipm',res['ipm_iters']); print('dual',res['dual'])
