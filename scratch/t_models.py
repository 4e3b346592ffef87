import sys, time
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo')
import numpy as np, gusto_oracle as go, gusto_jl_amd as g
P=g.problems
# dubins
x0,glo,ghi,tf = P.dubins_batch(8)
x0[0] = [2,2,2]
o = go.Oracle(go.DUBINS_CAR, 30)
for b in range(4):
    o.set_problem(x0[b],glo[b],ghi[b],tf[b]); t0=time.time(); r=o.solve(30); 
    print('dubins',b,x0[b],'iters',r['iterations'],'conv',r['converged'],r['successful'],go.STOP_REASON[r['stop_reason']],'ipm',r['ipm_iters'],'J',r['J_true'][-1], 'conv', r['conv'][-3:], 'time',time.time()-t0)
    print('   status',[go.SCP_STATUS[s] for s in r['scp_status']], 'rho', r['rho'][-3:], 'omega', r['omega'][-1])
# astrobee se3
boxes,sph = P.iss_corner_env(True)
x0,glo,ghi,tf = P.astrobee_se3_batch(4)
o = go.Oracle(go.ASTROBEE_SE3, 50, boxes=boxes, spheres=sph)
for b in range(3):
    o.set_problem(x0[b],glo[b],ghi[b],tf[b]); t0=time.time(); r=o.solve(30)
    print('se3',b,'iters',r['iterations'],'conv',r['converged'],r['successful'],go.STOP_REASON[r['stop_reason']],'ipm',r['ipm_iters'],'J',r['J_true'][-1],'time',time.time()-t0)
    print('   status',[go.SCP_STATUS[s] for s in r['scp_status']], 'omega', r['omega'][-1], 'Delta', r['Delta'][-1])
x0,glo,ghi,tf = P.astrobee_manifold_batch(4)
o = go.Oracle(go.ASTROBEE_SE3_MANIFOLD, 50, boxes=boxes, spheres=sph)
for b in range(3):
    o.set_problem(x0[b],glo[b],ghi[b],tf[b]); t0=time.time(); r=o.solve(30)
    print('manifold',b,'iters',r['iterations'],'conv',r['converged'],r['successful'],go.STOP_REASON[r['stop_reason']],'ipm',r['ipm_iters'],'J',r['J_true'][-1],'time',time.time()-t0)
    print('   status',[go.SCP_STATUS[s] for s in r['scp_status']], 'omega', r['omega'][-1], 'qnorm', np.linalg.norm(r['X'][:,6:10],axis=1)[[0,10,25,49]])
