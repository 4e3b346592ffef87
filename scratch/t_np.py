import sys,time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, gusto_oracle as go, gusto_jl_amd as g, np_models as M
P=g.problems
def run(model_id, model, N, batch, b, boxes=None, sph=None, Delta=None, omega=1.0):
    x0,glo,ghi,tf=batch
    o=go.Oracle(model_id,N,boxes=boxes,spheres=sph); o.set_problem(x0[b],glo[b],ghi[b],tf[b])
    Xp,Up=o.init_straightline()
    D=Delta or model.Delta0
    t=time.time(); r=o.subproblem(Xp,Up,D,omega,D/8+model.clearance)
    s=M.solve_subproblem(model,N,tf[b],x0[b],glo[b],ghi[b],Xp,Up,D,omega,boxes if boxes is not None else (), sph if sph is not None else ())
    print(model.__name__,'N',N,'b',b,'oracle st',r['status'],'obj',r['obj'],'slsqp obj',s['obj'],s['res'].status,s['res'].nit,'npen',s['n_pen'],'dX',np.abs(s['X']-r['X']).max(),'dU',np.abs(s['U']-r['U']).max(),'eqv',s['eq_violation'],'ineq',s['ineq_min'],'t %.1fs'%(time.time()-t))
run(go.DUBINS_CAR,M.Dubins,12,P.dubins_batch(4),0)
run(go.DUBINS_CAR,M.Dubins,12,P.dubins_batch(4),2)
bx,sp=P.iss_corner_env(True)
run(go.ASTROBEE_SE3,M.AstrobeeSE3,8,P.astrobee_se3_batch(3),0,bx,sp)
run(go.ASTROBEE_SE3,M.AstrobeeSE3,8,P.astrobee_se3_batch(3),1,bx,sp,Delta=0.5,omega=10.0)
run(go.ASTROBEE_SE3_MANIFOLD,M.AstrobeeSE3Manifold,8,P.astrobee_manifold_batch(3),0,bx,sp)
