from study1 import summary
from nb import *
for pv in (1e18, 10.0, 0.0):
    o,r=run(dm=dict(pen_mode=2,pen_value=pv)); summary('arm pen -> %g'%pv, r); 
    if pv==1e18: show(r)
