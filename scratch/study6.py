from study1 import summary
from nb import *
res=[]
for n in (0,24,25,32,64):
  for ph in ((0.0,) if n==0 else (0.0, np.pi/max(n,1))):
    for m in np.arange(0.0,0.0031,0.00025):
        o,r=run(dm=dict(n_poly=n,poly_phase=ph,margin=m),max_iter=8)
        J=r['J_true']; err=[(J[k]-NB['J'][k])/NB['J'][k]*100 for k in range(1,7)]
        cerr=[(r['conv'][k]-NB['conv'][k])/NB['conv'][k]*100 for k in range(1,7)]
        res.append((max(abs(e) for e in err),n,ph,m,err,cerr,''.join(str(int(a)) for a in r['scp_status'][:8])))
res.sort(key=lambda t:t[0])
for t in res[:15]: print('maxerr %.2f%% n=%d ph=%.3f margin=%.5f J err %s conv err %s st %s'%(t[0],t[1],t[2],t[3],' '.join('%+.2f'%e for e in t[4]),' '.join('%+.1f'%e for e in t[5]),t[6]))
