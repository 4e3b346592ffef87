from proto import *
import time
def cost_true(U,dt,N):
    return sum(0.5*dt*np.sum(U[k-1]**2+U[k]**2) for k in range(1,N))
def ratio(X,U,Xp,Up,env,N):
    num=den=0.
    for k in range(N-1):
        lin = f_dyn(Xp[k],Up[k]) + A_c@(X[k]-Xp[k])
        num += np.linalg.norm(f_dyn(X[k],U[k])-lin); den += np.linalg.norm(lin)
    for k in range(N):
        r0 = Xp[k,0:2]; r = X[k,0:2]
        for off in (np.array([0,0.]), np.array([0,0.15])):
            for i in range(len(env)):
                d0,nh = dist_body(r0+off, env[i])
                lin = CLR-(d0+nh@(r-r0))
                d1,_ = dist_body(r+off, env[i])
                num += abs((CLR-d1)-lin); den += abs(lin)
    return num/den if den!=0 else float('nan')
def conv_metric(X,Xp):
    return np.max(np.linalg.norm(X-Xp,axis=1))/np.max(np.linalg.norm(X,axis=1))
def cvx_sat(X,Xp,env,dtog,eps,N):
    for k in range(N):
        if X[k,3]**2+X[k,4]**2-VMAX**2 >= eps: return False
    for k in range(N):
        if X[k,5]**2-WMAX**2 >= eps: return False
    for k in range(N):
        for i in range(len(env)):
            d,nh = dist_body(Xp[k,0:2],env[i])
            if d<dtog:
                if CLR-(d+nh@(X[k,0:2]-Xp[k,0:2])) >= eps: return False
    return True
def scp(x_init,x_goal,N,tf,env,max_iter=30,verbose=True,method='riccati'):
    dt=tf/(N-1)
    D0,w0,wmax,eps,r0_,r1_,bs,bf,gf = 3.,1.,1e10,1e-2,0.1,0.3,2.,0.5,10.
    thr=1e-2
    X,U = straight(x_init,x_goal,N)
    Dv=[D0]; wv=[w0]; conv=[0.]; Jt=[cost_true(U,dt,N)]; status=['NA']; acc=[True]
    dtog = D0/8+CLR
    it=0; converged=False; tot_ipm=0
    while it<max_iter:
        r = ipm(X,U,x_init,np.arange(6),x_goal,N,dt,Dv[-1],wv[-1],env,dtog,method=method)
        tot_ipm += r['iters']
        Xn,Un = r['X'],r['U']
        conv.append(conv_metric(Xn,X))
        trs = max(np.sum((Xn-X)**2,axis=1)) - Dv[-1] <= 1e-7
        cs = cvx_sat(Xn,X,env,dtog,eps,N)
        if trs:
            rho = ratio(Xn,Un,X,U,env,N)
            if rho>r1_:
                status.append('InaccurateModel'); acc.append(False); Dv.append(bf*Dv[-1]); wv.append(wv[-1])
            else:
                acc.append(True)
                Dv.append(min(bs*Dv[-1],D0) if rho<r0_ else Dv[-1])
                if not cs: status.append('ViolatesConstraints'); wv.append(gf*wv[-1])
                else: status.append('OK'); wv.append(wv[-1])
        else:
            rho=float('nan')
            status.append('TrustRegionViolated'); acc.append(False); Dv.append(Dv[-1]); wv.append(gf*wv[-1])
        if acc[-1]:
            Jt.append(cost_true(Un,dt,N)); X,U=Xn,Un
        else: Jt.append(Jt[-1])
        dtog = Dv[-1]/8+CLR
        it+=1
        if verbose: print(f"scp {it:2d} ipm {r['iters']:2d} {status[-1]:20s} rho {rho:.4f} conv {conv[-1]:.5f} D {Dv[-1]:.4f} w {wv[-1]:.0f} J {Jt[-1]:.6f} Jfull {r['obj']:.6f}")
        if wv[-1]>wmax: break
        if not acc[-1]: continue
        if it>2 and conv[-1]+conv[-2]<=thr:
            converged=True; break
    return dict(X=X,U=U,converged=converged,iters=it,ipm=tot_ipm,status=status,J=Jt)
if __name__=='__main__':
    import sys
    N=int(sys.argv[1]) if len(sys.argv)>1 else 50
    x_init = np.array([0.2,2.4,0,0,0,0]); x_goal = np.array([3.,0.5,0,0.05,-0.05,0])
    env = table_env()
    t0=time.time()
    r=scp(x_init,x_goal,N,200.,env)
    print('converged',r['converged'],'iters',r['iters'],'ipm total',r['ipm'],'time',time.time()-t0)
    # clearance check
    print('min dist', min(dist_body(r['X'][k,0:2],env[i])[0] for k in range(N) for i in range(len(env))))
