from study1 import summary
from nb import *
for mode in (3,4):
  for band in (1e-3,3e-3,1e-2,3e-2):
    o,r=run(dm=dict(pen_mode=mode,pen_value=1e18,pen_band=band)); summary('mode %d band %g'%(mode,band), r)
