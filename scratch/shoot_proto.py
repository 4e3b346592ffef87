import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, gusto_oracle as go, gusto_jl_amd as g
P=g.problems
v,kk=2.0,1.0
def f(z):
    x,y,th,px,py,pth=z
    u=kk/2*pth
    return np.array([v*np.cos(th), v*np.sin(th), kk*u, 0,0, px*v*np.sin(th)-py*v*np.cos(th)])
def shoot(p0,x0,tf,steps=29*4):
    z=np.concatenate([x0,p0]); h=tf/steps
    for _ in range(steps):
        k1=f(z); k2=f(z+0.5*h*k1); k3=f(z+0.5*h*k2); k4=f(z+h*k3); z=z+h/6*(k1+2*k2+2*k3+k4)
    return z[:3]
def newton(p,x0,xg,tf,ftol=1e-9,maxit=100):
    F=xg-shoot(p,x0,tf); nf=np.abs(F).max()
    for it in range(maxit):
        if nf<=ftol: return p,nf,it,True
        J=np.zeros((3,3)); h=1e-6
        for j in range(3):
            d=np.zeros(3); d[j]=h; J[:,j]=((xg-shoot(p+d,x0,tf))-F)/h
        try: dp=np.linalg.solve(J,-F)
        except np.linalg.LinAlgError: return p,nf,it,False
        a=1.0
        while a>1e-4:
            Fn=xg-shoot(p+a*dp,x0,tf); nn=np.abs(Fn).max()
            if nn<nf: break
            a*=0.5
        else: return p,nf,it,False
        p=p+a*dp; F=Fn; nf=nn
    return p,nf,maxit,nf<=ftol
x0,glo,ghi,tf=P.dubins_batch(12)
for b in range(12):
    o=go.Oracle(go.DUBINS_CAR,30); o.set_problem(x0[b],glo[b],ghi[b],tf[b])
    r=o.solve(30); d=r['dual']
    out=[]
    for sg in (1,-1):
        p,nf,it,ok=newton(sg*d,x0[b],glo[b],tf[b]); out.append((sg,ok,it,nf,np.round(p,4)))
    print(b,'scp conv',r['converged'],'it',r['iterations'],'stop',r['stop_reason'],'dual',np.round(d,4),'u0',round(r['U'][0,0],4), out, flush=True)
