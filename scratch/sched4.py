from sched3 import *
# finer keys: level then probe ipm iterations (quantised)
def pol_fine(nb):
    def pol(b,k,om):
        ip=D[b][0][:k].sum()
        q=min(nb-1,int(ip/ (k*30.0/nb*2) ))   # crude bucket by mean ipm per trip
        return -(lvl(om[k-1])*nb+q)
    return pol
sl=lambda k0: 1 if k0<2 else 10**6
print('base probe1+1+completion', sim(pol_lvl, slice_trips=sl, ovh=0.5))
for nb in (2,4,8): print('fine',nb, sim(pol_fine(nb), slice_trips=sl, ovh=0.5))
# oracle priority = true remaining work (upper bound of what any key can do in this scheme)
def pol_true(b,k,om): return -D[b][0][k:].sum()
print('true remaining', sim(pol_true, slice_trips=sl, ovh=0.5))
print('true remaining, 1 probe', sim(pol_true, slice_trips=lambda k0: 1 if k0<1 else 10**6, ovh=0.5))
print('true remaining, 3 probe', sim(pol_true, slice_trips=lambda k0: 1 if k0<3 else 10**6, ovh=0.5))
# key = conv measure? not available. key = number of rejected steps so far?
