from proto import *
from scp_proto import *
from t4 import gen
import sys
env = table_env()
b=int(sys.argv[1])
x_init = gen(b); x_goal = np.array([3.,0.5,0,0.05,-0.05,0])
print(x_init)
r = scp(x_init,x_goal,50,200.,env,verbose=True)
