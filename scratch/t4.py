from proto import *
from scp_proto import *
import multiprocessing as mp, warnings
warnings.filterwarnings('error')
env = table_env()
def gen(b):
    rng = np.random.default_rng(b)
    while True:
        p = np.array([rng.uniform(0.25,3.40), rng.uniform(0.25,2.49)])
        if min(dist_body(p,env[i])[0] for i in range(len(env))) >= 0.10: break
    return np.array([p[0],p[1],0,0,0,0])
def run(b):
    x_init = gen(b); x_goal = np.array([3.,0.5,0,0.05,-0.05,0])
    try:
        r = scp(x_init,x_goal,50,200.,env,verbose=False)
        X=r['X']
        md = min(dist_body(X[k,0:2],env[i])[0] for k in range(50) for i in range(len(env)))
        return b, r['converged'], r['iters'], r['ipm'], r['status'][-1], md, r['J'][-1]
    except Exception as e:
        return b, 'EXC', repr(e)[:100]
if __name__=='__main__':
    with mp.Pool(8) as p:
        for res in p.imap(run, range(32)):
            print(res)
