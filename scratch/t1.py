from proto import *
N=50; tf=200.; dt = tf/(N-1)
x_init = np.array([0.2,2.4,0,0,0,0]); x_goal = np.array([3.,0.5,0,0.05,-0.05,0])
env = table_env()
Xp,Up = straight(x_init,x_goal,N)
# Compare a single Newton solve dense vs riccati on random SPD data
rng = np.random.default_rng(0)
Hx = np.zeros((N,n,n)); Hu=np.zeros((N,m,m))
for k in range(N):
    a = rng.standard_normal((n,n)); Hx[k] = a@a.T*10**rng.uniform(-8,2)
    a = rng.standard_normal((m,m)); Hu[k] = a@a.T+np.eye(m)
gx = rng.standard_normal((N,n)); gu = rng.standard_normal((N,m)); rd = rng.standard_normal((N,n))*0.1; rd[0]=0
r0 = rng.standard_normal(n)*0.1; rg = rng.standard_normal(6)
Fk,Gk,bk,h = linearize(Xp,Up,N,dt)
C = np.eye(6)
a = solve_dense(Hx,Hu,gx,gu,rd,r0,rg,Fk,Gk,bk,C,N)
b = solve_riccati(Hx,Hu,gx,gu,rd,r0,rg,Fk,Gk,bk,C,N)
for x,y,nm in zip(a,b,['dX','dU','nu','mug']):
    print(nm, np.abs(x-y).max(), np.abs(x).max())
