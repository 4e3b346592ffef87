from proto import *
from scp_proto import *
N=50; tf=200.; dt=tf/(N-1)
x_init = np.array([0.2,2.4,0,0,0,0]); x_goal = np.array([3.,0.5,0,0.05,-0.05,0])
env = table_env()
X,U = straight(x_init,x_goal,N)
r = ipm(X,U,x_init,np.arange(6),x_goal,N,dt,3.,1.,env,3/8+CLR,method='dense')
r2 = ipm(r['X'],r['U'],x_init,np.arange(6),x_goal,N,dt,3.,1.,env,3/8+CLR,method='dense',verbose=True)
