from nb import *
o,r=run(max_iter=6)
Xp,Up=o.traj()
r2=o.solve(1)
X,U=o.traj()
N=200
f=lambda x,u: o.dynamics(x,u)
num_d=den_d=0
for k in range(N-1):
    fp,A,B=f(Xp[k],Up[k]); fn,_,_=f(X[k],U[k])
    lin=fp+A@(X[k]-Xp[k]); num_d+=np.linalg.norm(fn-lin); den_d+=np.linalg.norm(lin)
num_o=den_o=0
big=[]
for k in range(N):
    for c in range(2):
        for i in range(14):
            d0,nh=o.signed_distance(c,Xp[k,:2],i); d1,_=o.signed_distance(c,X[k,:2],i)
            lin=0.05-(d0+nh[:2]@(X[k,:2]-Xp[k,:2]))
            e=abs((0.05-d1)-lin); num_o+=e; den_o+=abs(lin)
            big.append((e,k,c,i,d0,d1))
print('dyn num %.4g den %.4g ; obs num %.4g den %.4g ; rho %.4g; oracle rho %s'%(num_d,den_d,num_o,den_o,(num_d+num_o)/(den_d+den_o), r2['rho'][-3:]))
big.sort(reverse=True); print(big[:8])
print('step max', np.abs(X-Xp).max(0))
