import numpy as np, time, sys
from mcfost_amd.host import model as M
from oracle import Oracle
dm=float(sys.argv[3]) if len(sys.argv)>3 else 1e-2
cfg=M.small(n_rad=30,nz=20,dust_mass=dm)
N=int(float(sys.argv[1])); gam=float(sys.argv[2]); ns=6
def run(mrw, seed):
    m=M.build_model(cfg)
    if mrw: M.init_mrw(m,weights=mrw,gamma=gam)
    o=Oracle(m,N)
    r=o.run_thermal(N,seed=seed,n_threads=8)
    return o.temp_finale(r["E_abs"]).astype(float), r["counters"]
t=time.time()
A=np.array([run(None,10+s)[0] for s in range(ns)]); tb=time.time()-t
t=time.time()
B=np.array([run("dB_dT",30+s)[0] for s in range(ns)]); tm=time.time()-t
print("times",tb,tm)
ma,mb=A.mean(0),B.mean(0)
se=np.sqrt(A.var(0,ddof=1)/ns+B.var(0,ddof=1)/ns)
z=(mb-ma)/np.maximum(se,1e-12)
nz,nr=20,30
print("bias % (rows j=0..7)"); print(np.round(100*((mb-ma)/ma).reshape(nz,nr)[:8,::2],1))
print("z"); print(np.round(z.reshape(nz,nr)[:8,::2],1))
print("frac |z|>3:", np.mean(np.abs(z)>3), " mean z", z.mean(), " rms z", np.sqrt((z**2).mean()))
