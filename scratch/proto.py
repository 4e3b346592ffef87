"""Numpy prototype of the GuSTO convex subproblem + IPM (dense KKT and Riccati) for freeflyerSE2."""
import numpy as np
np.set_printoptions(linewidth=200, precision=6, suppress=False)

# ---------------- robot/model constants (freeflyer.jl:28-62) ----------------
MASS = 0.5*(15.36+18.08); J = 0.184; JINV = 1.0/J
R_BODY = 0.157; VMAX = 0.2; AMAX = 2*0.185/MASS; WMAX = 20*np.pi/180
ALMAX = 0.593/(J/6.43); CLR = 0.05
n, m = 6, 3

def table_env():
    ft2m = 0.3048
    lo = np.array([0., 0., -1.]); hi = np.array([12*ft2m, 9*ft2m, 1.])  # worldAABB (z pushed? length 3 given so no push)
    lo = np.array([0.,0.,0.]); hi = np.array([12.,9.,0.001])*ft2m
    a = 10.
    boxes = []
    boxes.append(([hi[0], -a, -a], [hi[0]+a, a, a]))
    boxes.append(([lo[0]-a, -a, -a], [lo[0], a, a]))
    boxes.append(([-a, hi[1], -a], [a, hi[1]+a, a]))
    boxes.append(([-a, lo[1]-a, -a], [a, lo[1], a]))
    centers = [[0.460,0.315,0.0],[0.201,1.085,0.0],[0.540,2.020,0.0],[1.374,0.196,0.0],[1.063,1.354,0.0],
               [1.365,2.322,0.0],[2.221,0.548,0.0],[2.077,1.443,0.0],[3.098,1.186,0.0],[2.837,2.064,0.0]]
    w = np.array([0.27,0.27,0.127]); infl = 0.05*np.ones(3)
    for c in centers:
        c = np.array(c)
        mn = (c-0.5*w-infl+np.array([0,0,0.5*w[0]])).astype(np.float32).astype(np.float64)
        sz = (w+2*infl).astype(np.float32).astype(np.float64)
        boxes.append((mn, mn+sz))
    return np.array([np.concatenate([b[0],b[1]]) for b in boxes])

def sdf_rect2(p, lo, hi):
    """signed distance point->rect (2D) and outward unit normal"""
    dx = max(lo[0]-p[0], 0., p[0]-hi[0]); dy = max(lo[1]-p[1], 0., p[1]-hi[1])
    if dx > 0 or dy > 0:
        q = np.minimum(np.maximum(p, lo), hi)
        v = p-q; d = np.hypot(v[0], v[1])
        return d, v/d
    cands = [(p[0]-lo[0], np.array([-1.,0.])), (hi[0]-p[0], np.array([1.,0.])),
             (p[1]-lo[1], np.array([0.,-1.])), (hi[1]-p[1], np.array([0.,1.]))]
    best = cands[0]
    for c in cands[1:]:
        if c[0] < best[0]: best = c
    return -best[0], best[1]

def dist_body(p2, box):
    d, nh = sdf_rect2(p2, box[0:2], box[3:5])
    return d - R_BODY, nh

def f_dyn(x, u):
    return np.array([x[3], x[4], x[5], u[0]/MASS, u[1]/MASS, u[2]*JINV])
A_c = np.kron(np.array([[0,1],[0,0.]]), np.eye(3))
B_c = np.zeros((6,3)); B_c[3,0]=1/MASS; B_c[4,1]=1/MASS; B_c[5,2]=JINV

class Row:
    __slots__=('k','isu','idx','a','v0','b','c0','hard')
    def __init__(s,k,isu,idx,a,v0,b,c0,hard):
        s.k,s.isu,s.idx,s.a,s.v0,s.b,s.c0,s.hard = k,isu,np.array(idx,int),np.array(a,float),np.array(v0,float),np.array(b,float),float(c0),hard
    def val(s, v):
        w = v[s.idx]
        return float(np.sum(s.a*(w-s.v0)**2) + np.sum(s.b*w) + s.c0)
    def grad(s, v):
        w = v[s.idx]
        return 2*s.a*(w-s.v0) + s.b
    def hdiag(s):
        return 2*s.a

def build_rows(Xp, Up, N, Delta, omega, env, dtog):
    rows = []
    for k in range(N):
        # TR (penalised): omega*||x-xp||^2 - Delta
        rows.append(Row(k,0,range(6),[omega]*6,Xp[k],[0]*6,-Delta,False))
        rows.append(Row(k,0,[3,4],[omega]*2,[0,0],[0,0],-omega*VMAX**2,False))
        rows.append(Row(k,0,[5],[omega],[0],[0],-omega*WMAX**2,False))
        for i in range(len(env)):
            d, nh = dist_body(Xp[k,0:2], env[i])
            if d < dtog:
                # omega*(clr - (d + nh.(r-r0)))
                rows.append(Row(k,0,[0,1],[0,0],[0,0],-omega*nh, omega*(CLR-d+nh@Xp[k,0:2]),False))
        if k < N-1:
            s2 = 1.0/(MASS*AMAX)**2
            rows.append(Row(k,1,[0,1],[s2,s2],[0,0],[0,0],-1.0,True))
            s3 = (JINV/ALMAX)**2
            rows.append(Row(k,1,[2],[s3],[0],[0],-1.0,True))
    return rows

def linearize(Xp, Up, N, dt):
    F = np.eye(n)+0.5*dt*A_c; G = np.eye(n)-0.5*dt*A_c; b = 0.5*dt*B_c
    e = np.array([f_dyn(Xp[k],Up[k]) - A_c@Xp[k] - B_c@Up[k] for k in range(N)])
    h = np.zeros((N,n))
    for k in range(1,N):
        h[k] = 0.5*dt*(e[k-1]+e[k])
    return [F]*N, [G]*N, [b]*N, h

# ----------------------------------------------------------------------------
def ipm(Xp, Up, x_init, goal_idx, goal_val, N, dt, Delta, omega, env, dtog, method='dense', verbose=False, maxit=60, tol=1e-8):
    Fk,Gk,bk,h = linearize(Xp,Up,N,dt)
    rows = build_rows(Xp,Up,N,Delta,omega,env,dtog)
    wk = np.full(N, dt); wk[0]=wk[-1]=0.5*dt
    X = Xp.copy(); U = Up.copy(); X[0] = x_init
    nr = len(rows)
    hard = np.array([r.hard for r in rows])
    t = np.zeros(nr); lam = np.zeros(nr); s = np.zeros(nr)
    for i,r in enumerate(rows):
        v = (U if r.isu else X)[r.k]
        g = r.val(v)
        if r.hard:
            t[i] = max(-g, 1e-2); lam[i] = 1.0/t[i]*1e-1  # mu0=0.1
        else:
            s[i] = max(g,0.)+1.0; t[i] = s[i]-g; lam[i]=0.5
    lamb = np.where(hard, 0., 0.5)
    nu = np.zeros((N,n))
    C = np.zeros((len(goal_idx), n)); C[np.arange(len(goal_idx)), goal_idx] = 1
    ng = len(goal_idx)
    hist = []
    for it in range(maxit):
        # residuals
        rd = np.zeros((N,n))
        for k in range(1,N):
            rd[k] = Fk[k-1]@X[k-1] + bk[k-1]@U[k-1] - Gk[k]@X[k] + bk[k]@U[k] + h[k]
        r0 = x_init - X[0]
        rg = goal_val - X[N-1][goal_idx]
        gval = np.zeros(nr); rp = np.zeros(nr)
        for i,r in enumerate(rows):
            v = (U if r.isu else X)[r.k]
            gval[i] = r.val(v)
            rp[i] = gval[i]+t[i] if r.hard else gval[i]-s[i]+t[i]
        comp = np.where(hard, t*lam, t*lam + s*lamb)
        ncomp = np.sum(hard) + 2*np.sum(~hard)
        mu = comp.sum()/ncomp
        # dual residual
        rdx = np.zeros((N,n)); rdu = np.zeros((N,m))
        for k in range(N):
            rdu[k] = 2*wk[k]*U[k]
        for i,r in enumerate(rows):
            v = (U if r.isu else X)[r.k]
            (rdu if r.isu else rdx)[r.k][r.idx] += lam[i]*r.grad(v)
        # nu[k] multiplies rd[k] (k>=1); nu[0] is init multiplier (for X[0]-x_init=0  -> sign: r0 row -(X0 - xinit))
        for k in range(N):
            if k+1<N:
                rdx[k] += Fk[k].T@nu[k+1]; rdu[k] += bk[k].T@nu[k+1]
            if k>=1:
                rdx[k] -= Gk[k].T@nu[k]; rdu[k] += bk[k].T@nu[k]
        # goal mult
        # (tracked separately)
        if it==0: mug = np.zeros(ng)
        rdx[N-1] += C.T@mug
        rdx[0] = 0  # x0 fixed: its stationarity defines nu[0]
        res_p = max(np.abs(rd).max(), np.abs(rp).max(), np.abs(rg).max() if ng else 0, np.abs(r0).max())
        res_d = max(np.abs(rdx).max(), np.abs(rdu).max())
        hist.append((it, mu, res_p, res_d))
        if verbose: print(f"it {it:2d} mu {mu:.3e} rp {res_p:.3e} rd {res_d:.3e}")
        if res_p < tol and res_d < tol*(1+np.abs(nu).max()) and mu < tol*0.1:
            break
        def build_and_solve(mu_t, kap_a=None, kap_b=None):
            """returns dX,dU,nu_new,mug_new, and row deltas"""
            if kap_a is None: kap_a = np.zeros(nr); kap_b = np.zeros(nr)
            Hx = np.zeros((N,n,n)); Hu = np.zeros((N,m,m)); gx = np.zeros((N,n)); gu = np.zeros((N,m))
            for k in range(N):
                Hu[k] = 2*wk[k]*np.eye(m); gu[k] = 2*wk[k]*U[k]
            sig = np.zeros(nr); rho0 = np.zeros(nr); Dd = np.zeros(nr)
            for i,r in enumerate(rows):
                v = (U if r.isu else X)[r.k]
                gr = r.grad(v)
                if r.hard:
                    sig[i] = lam[i]/t[i]
                    coef = (mu_t - kap_a[i] + lam[i]*rp[i])/t[i]
                    lamH = lam[i]
                else:
                    lb = lamb[i]
                    Dd[i] = t[i] + lam[i]*s[i]/lb
                    rho0[i] = mu_t - t[i]*lam[i] - kap_a[i] + lam[i]*rp[i] - (lam[i]/lb)*(mu_t - s[i]*lb - kap_b[i])
                    sig[i] = lam[i]/Dd[i]
                    coef = lam[i] + rho0[i]/Dd[i]
                    lamH = lam[i]
                H = Hu if r.isu else Hx; g = gu if r.isu else gx
                H[r.k][np.ix_(r.idx,r.idx)] += sig[i]*np.outer(gr,gr) + lamH*np.diag(r.hdiag())
                g[r.k][r.idx] += coef*gr
            if method=='dense':
                dX,dU,nun,mugn = solve_dense(Hx,Hu,gx,gu,rd,r0,rg,Fk,Gk,bk,C,N)
            else:
                dX,dU,nun,mugn = solve_riccati(Hx,Hu,gx,gu,rd,r0,rg,Fk,Gk,bk,C,N)
            # row deltas
            dlam = np.zeros(nr); ds = np.zeros(nr); dt_ = np.zeros(nr)
            for i,r in enumerate(rows):
                v = (U if r.isu else X)[r.k]; dv = (dU if r.isu else dX)[r.k]
                w = r.grad(v)@dv[r.idx]
                if r.hard:
                    dt_[i] = -rp[i]-w
                    dlam[i] = (mu_t - t[i]*lam[i] - kap_a[i] - lam[i]*dt_[i])/t[i]
                else:
                    dlam[i] = (rho0[i] + lam[i]*w)/Dd[i]
                    ds[i] = (mu_t - s[i]*lamb[i] - kap_b[i] + s[i]*dlam[i])/lamb[i]
                    dt_[i] = -rp[i] - w + ds[i]
            return dX,dU,nun,mugn,dlam,ds,dt_
        def maxstep(tau, dlam, ds, dt_):
            a = 1.0
            def upd(a, v, dv):
                neg = dv < 0
                if neg.any(): a = min(a, (tau*(-v[neg]/dv[neg])).min())
                return a
            a = upd(a, t, dt_); a = upd(a, lam, dlam)
            pen = ~hard
            a = upd(a, s[pen], ds[pen]); a = upd(a, lamb[pen], -dlam[pen])
            return a
        # predictor
        dX,dU,nun,mugn,dlam,ds,dt_ = build_and_solve(0.0)
        a_aff = maxstep(1.0, dlam, ds, dt_)
        ta = t + a_aff*dt_; la = lam + a_aff*dlam; sa = s + a_aff*ds
        lba = lamb - a_aff*dlam
        comp_aff = np.where(hard, ta*la, ta*la + sa*lba)
        mu_aff = comp_aff.sum()/ncomp
        sigma = (mu_aff/mu)**3
        kap_a = dt_*dlam; kap_b = ds*(-dlam)
        dX,dU,nun,mugn,dlam,ds,dt_ = build_and_solve(max(sigma*mu,1e-11), kap_a, kap_b)
        tau = max(0.995, 1-mu) if mu<1 else 0.995
        a = maxstep(tau, dlam, ds, dt_)
        X += a*dX; U += a*dU; t += a*dt_; lam += a*dlam; s += a*ds; lamb = np.where(hard, 0., lamb - a*dlam)
        nu += a*(nun-nu); mug += a*(mugn-mug)
        if verbose: print(f"      a_aff {a_aff:.3f} sigma {sigma:.2e} alpha {a:.4f}")
    obj = float(np.sum(wk[:,None]*U**2) + s[~hard].sum())
    return dict(X=X,U=U,s=s,t=t,lam=lam,nu=nu,mug=mug,obj=obj,iters=it,hist=hist,rows=rows)

def solve_dense(Hx,Hu,gx,gu,rd,r0,rg,Fk,Gk,bk,C,N):
    nz = (n+m)*N; ng = C.shape[0]; ne = n*N + ng
    K = np.zeros((nz+ne, nz+ne)); rhs = np.zeros(nz+ne)
    ix = lambda k: slice((n+m)*k, (n+m)*k+n)
    iu = lambda k: slice((n+m)*k+n, (n+m)*(k+1))
    for k in range(N):
        K[ix(k),ix(k)] = Hx[k]; K[iu(k),iu(k)] = Hu[k]
        rhs[ix(k)] = -gx[k]; rhs[iu(k)] = -gu[k]
    # init row: dx0 = r0  (multiplier nu0, constraint: -(x0) ... use +I)
    E = np.zeros((ne, nz)); er = np.zeros(ne)
    E[0:n, ix(0)] = np.eye(n); er[0:n] = r0
    for k in range(1,N):
        rws = slice(n*k, n*(k+1))
        E[rws, ix(k-1)] = Fk[k-1]; E[rws, iu(k-1)] = bk[k-1]
        E[rws, ix(k)] = -Gk[k]; E[rws, iu(k)] += bk[k]
        er[rws] = -rd[k]
    E[n*N:, ix(N-1)] = C; er[n*N:] = rg
    K[:nz,nz:] = E.T; K[nz:,:nz] = E; rhs[nz:] = er
    # x0 hessian irrelevant but make nonsingular
    K[ix(0),ix(0)] += np.eye(n)*0
    sol = np.linalg.solve(K + np.diag(np.r_[np.zeros(nz), np.zeros(ne)]), rhs)
    dz = sol[:nz].reshape(N,n+m)
    nu = sol[nz:nz+n*N].reshape(N,n); mug = sol[nz+n*N:]
    return dz[:,:n].copy(), dz[:,n:].copy(), nu, mug

def solve_riccati(Hx,Hu,gx,gu,rd,r0,rg,Fk,Gk,bk,C,N):
    ng = C.shape[0]
    # per-stage precompute
    Phi = np.zeros((N,n,n)); Gam = np.zeros((N,n,m)); cc = np.zeros((N,n)); QQ = np.zeros((N,n+m,n+m)); qq = np.zeros((N,n+m))
    Mk = [None]*N
    for k in range(N):
        if k==0:
            Phi[0] = 0; Gam[0] = bk[0]; cc[0] = Fk[0]@r0
            QQ[0][n:,n:] = Hu[0]; qq[0][n:] = gu[0]
            continue
        M = np.linalg.inv(Gk[k]); Mk[k]=M
        Phi[k] = Fk[k]@M; Gam[k] = Phi[k]@bk[k] + bk[k]; cc[k] = Phi[k]@rd[k]
        Qt = M.T@Hx[k]@M; Qb = Qt@bk[k]
        QQ[k][:n,:n] = Qt; QQ[k][:n,n:] = Qb; QQ[k][n:,:n] = Qb.T; QQ[k][n:,n:] = Hu[k] + bk[k].T@Qb
        gy = Qt@rd[k] + M.T@gx[k]
        qq[k][:n] = gy; qq[k][n:] = gu[k] + bk[k].T@gy
    # terminal E columns at stage N-1: [M^T C^T ; b^T M^T C^T]
    MN = Mk[N-1]
    Ey = MN.T@C.T; Eu = bk[N-1].T@Ey
    P = np.zeros((n,n)); p = np.zeros(n); Pi = np.zeros((n,ng))
    Ps = np.zeros((N,n,n)); ps = np.zeros((N,n)); Pis = np.zeros((N,n,ng))
    Ks = np.zeros((N,m,n)); d0s = np.zeros((N,m)); Ds = np.zeros((N,m,ng)); 
    Gd = np.zeros((ng,ng)); th = np.zeros(ng)
    for k in range(N-1,-1,-1):
        Ps[k]=P; ps[k]=p; Pis[k]=Pi
        PG = np.hstack([Phi[k],Gam[k]])
        T = P@PG; tp = p + P@cc[k]
        Hh = QQ[k] + PG.T@T; l = qq[k] + PG.T@tp; Z = PG.T@Pi
        if k==N-1:
            Z = Z + np.vstack([Ey,Eu])
        th = th + Pi.T@cc[k]
        S = Hh[n:,n:]; Hyu = Hh[:n,n:]
        Sinv = np.linalg.inv(S)
        K = Sinv@Hyu.T; d0 = Sinv@l[n:]; D = Sinv@Z[n:]
        Ks[k]=K; d0s[k]=d0; Ds[k]=D
        Gd = Gd + Z[n:].T@D; th = th - Z[n:].T@d0
        P = Hh[:n,:n] - Hyu@K; p = l[:n] - Hyu@d0; Pi = Z[:n] - Hyu@D
        P = 0.5*(P+P.T)
    # dual: constraint C dx_{N-1} = rg, dx_{N-1} = M(dy' + b du + rd)
    # value: d/dmu: th + (C M rd - rg) - Gd mu = 0   (dy_{-1}=0)
    if ng:
        mug = np.linalg.solve(Gd, th + C@MN@rd[N-1] - rg)
    else:
        mug = np.zeros(0)
    # forward
    dy = np.zeros(n); dU = np.zeros((N,m)); dX = np.zeros((N,n)); nu = np.zeros((N,n))
    for k in range(N):
        d = d0s[k] + Ds[k]@mug
        du = -d - Ks[k]@dy
        if k==0:
            dX[0] = r0
        else:
            dX[k] = Mk[k]@(dy + bk[k]@du + rd[k])
        dyn = Phi[k]@dy + Gam[k]@du + cc[k]
        dU[k] = du
        if k+1<N:
            nu[k+1] = Ps[k]@dyn + ps[k] + Pis[k]@mug
        dy = dyn
    # nu[0]: from stationarity of x0:  Hx0 dx0 + gx0 + I*nu0 + F0^T nu1 = 0
    nu[0] = -(Hx[0]@dX[0] + gx[0] + (Fk[0].T@nu[1] if N>1 else 0))
    return dX,dU,nu,mug

def straight(x_init, x_goal, N):
    X = np.array([x_init + (x_goal-x_init)*k/(N-1) for k in range(N)])
    return X, np.zeros((N,m))

if __name__=='__main__':
    import time
    N=50; tf=200.; dt = tf/(N-1)
    x_init = np.array([0.2,2.4,0,0,0,0]); x_goal = np.array([3.,0.5,0,0.05,-0.05,0])
    env = table_env()
    Xp,Up = straight(x_init,x_goal,N)
    Delta=3.; omega=1.
    for meth in ('dense','riccati'):
        t0=time.time()
        r = ipm(Xp,Up,x_init,np.arange(6),x_goal,N,dt,Delta,omega,env,Delta/8+CLR,method=meth,verbose=True)
        print(meth, 'obj',r['obj'],'iters',r['iters'],'time',time.time()-t0)
        print('max TR', max(np.sum((r['X']-Xp)**2,axis=1)), 'uN', r['U'][-1])
