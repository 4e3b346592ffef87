from study1 import summary
from nb import *
for pv in (1e18, 1e3, 10.0, 0.0):
    o,r=run(dm=dict(pen_mode=1,pen_value=pv)); summary('pen -> %g'%pv, r); 
    if pv==1e18: show(r)
