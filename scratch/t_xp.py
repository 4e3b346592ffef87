import sys, ctypes as C
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo')
import numpy as np, gusto_oracle as go, gusto_jl_amd as g
go._LIB='/tmp/libxp.so'; go._lib=None
import os
L=go.lib.__wrapped__() if hasattr(go.lib,'__wrapped__') else None
# load custom lib manually
go._lib=None
orig=go._LIB
def load():
    Lx=C.CDLL('/tmp/libxp.so'); return Lx
Lx=load(); Lx.go_xp.argtypes=[C.c_double]*3
# monkeypatch: make go.lib() use our lib
import types
go._lib=None
go.build=lambda force=False: '/tmp/libxp.so'
Lg=go.lib()
Lg.go_xp.argtypes=[C.c_double]*3
P=g.problems; env=P.freeflyer_env(); N=50; B=512
x0,glo,ghi,tf = P.freeflyer_batch(B)
for (t0,mu0,s0) in [(1e-2,0.01,0.01),(1e-2,0.01,0.003),(1e-2,0.001,0.01),(1e-3,0.001,0.01),(1e-2,0.01,0.001),(1e-3,0.001,0.003)]:
    Lg.go_xp(t0,mu0,s0)
    r = go.solve_batch(go.FREEFLYER_SE2, N, env, None, x0, glo, ghi, tf, 30, 8)
    print((t0,mu0,s0),'ipm total',r['ipm_iters'].sum(),'per trip',r['ipm_iters'].sum()/r['iterations'].sum(),'conv',r['converged'].sum(),'trips',r['iterations'].sum(), 'max', r['ipm_iters'].max())
