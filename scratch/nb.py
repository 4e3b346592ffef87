import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import gusto_oracle as go, gusto_jl_amd as g
P=g.problems
NB=dict(
 accept=[1,1,1,1,1,1,1,0,0,0,1,1,1,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1],
 conv=[0.0,0.140958,0.0771133,0.0749378,0.0593635,0.0354587,0.0181484,0.0115899,0.0116291,0.0115322,0.011723,0.00745733,0.0121744,0.00745798,0.00745974,0.00745665,0.00745854,0.00744797,0.00747751,0.00746389,0.0503231,0.0405726,0.0318271,0.0198506,0.0201292,0.0148137,0.00949141,0.0037999,0.00379924],
 omega=[1.0]*20+[10.0,10.0]+[100.0]*7,
 Delta=[3.0,3.0,3.0,3.0,3.0,3.0,3.0,1.5,0.75,0.375,0.75,1.5,3.0,1.5,0.75,0.375,0.1875,0.09375,0.046875,0.0234375,0.046875,0.09375,0.1875,0.375,0.75,1.5,3.0,3.0,3.0],
 J=[0.0,0.152419,0.0865004,0.0744733,0.0664654,0.0638019,0.0619878,0.0619878,0.0619878,0.0619878,0.0604713,0.0592238,0.0583014,0.0583014,0.0583014,0.0583014,0.0583014,0.0583014,0.0583014,0.0583014,0.0496451,0.179539,0.0644078,0.128646,0.11305,0.112113,0.111894,0.111695,0.111656])
def run(mp=None, N=200, max_iter=40, env=None, ipm=None, dm=None):
    env = P.freeflyer_env() if env is None else env
    o=go.Oracle(go.FREEFLYER_SE2,N,boxes=env, model_params=mp, ipm_opts=ipm)
    if dm: o.set_distance_model(**dm)
    o.set_problem(P.FREEFLYER_X_INIT,P.FREEFLYER_X_GOAL,P.FREEFLYER_X_GOAL,P.FREEFLYER_TF)
    return o, o.solve(max_iter)
def show(r):
    n=len(r['omega'])
    print(' k acc st  Delta   omega    J_true   (nb J)    conv    (nb conv)   rho')
    rho=list(r['rho'][2:]) if len(r['rho'])>2 else []
    ri=0
    for k in range(max(n,29)):
        s=''
        if k<n:
            rh=''
            if k>=1 and r['tr_sat'][k]:
                rh='%.4g'%rho[ri]; ri+=1
            s='%2d %d %d %8.5f %7.1f %9.6f'%(k,r['accept'][k],r['scp_status'][k],r['Delta'][k],r['omega'][k],r['J_true'][k])
            s+=' (%9.6f) %9.6f (%9.6f) %s'%(NB['J'][k] if k<29 else np.nan, r['conv'][k], NB['conv'][k] if k<29 else np.nan, rh)
        else:
            s='%2d nb: acc %d Delta %g omega %g J %g conv %g'%(k,NB['accept'][k],NB['Delta'][k],NB['omega'][k],NB['J'][k],NB['conv'][k])
        print(s)
if __name__=='__main__':
    o,r=run()
    print(r['iterations'],r['converged'],len(r['J_true']))
    show(r)
