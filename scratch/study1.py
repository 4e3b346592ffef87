from nb import *
import sys
def summary(tag, r):
    J=r['J_true']; n=min(len(J),29)
    err=[ (J[k]-NB['J'][k])/NB['J'][k]*100 for k in range(1,min(n,7))]
    acc=''.join(str(int(a)) for a in r['accept'])
    st=''.join(str(int(a)) for a in r['scp_status'])
    print('%-40s it=%2d conv=%d J1..6 err%%: %s  st=%s om_end=%g Jend=%.4f'%(tag,r['iterations'],r['converged'],' '.join('%+.2f'%e for e in err),st,r['omega'][-1],J[-1]))
if __name__=="__main__":
  o,r=run(); summary('analytic', r)
  o,r=run(dm=dict(vertical_escape=1)); summary('disc + vertical escape', r)
  for m in (0.0005,0.001,0.002,0.004):
      o,r=run(dm=dict(margin=m)); summary('disc margin %g'%m, r)
  for n in (8,12,16,25,32):
      o,r=run(dm=dict(n_poly=n)); summary('poly %d'%n, r)
      o,r=run(dm=dict(n_poly=n,vertical_escape=1)); summary('poly %d + vesc'%n, r)
