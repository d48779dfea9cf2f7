import numpy as np, time, sys
from mcfost_amd.host import model as M
from oracle import Oracle
dm=1e-2
cfg=M.small(n_rad=30,nz=20,dust_mass=dm)
N=int(float(sys.argv[1])); gam=float(sys.argv[2]); ns=6
wts=sys.argv[3] if len(sys.argv)>3 else "dB_dT"
m0=M.build_model(cfg); o0=Oracle(m0,N)
prior=np.mean([o0.run_thermal(N,seed=100+s,n_threads=8)["E_abs"] for s in range(6)],axis=0)
def run(mrw, seed):
    m=M.build_model(cfg)
    if mrw: M.init_mrw(m,weights=mrw,gamma=gam)
    o=Oracle(m,N)
    r=o.run_thermal(N,seed=seed,n_threads=8,frozen=True,E_prior=prior)
    return r["E_abs"], r["counters"]
A=np.array([run(None,10+s)[0] for s in range(ns)])
B=np.array([run(wts,30+s)[0] for s in range(ns)])
ma,mb=A.mean(0),B.mean(0)
se=np.sqrt(A.var(0,ddof=1)/ns+B.var(0,ddof=1)/ns)
z=(mb-ma)/np.maximum(se,1e-300)
nz,nr=20,30
print("E bias % (rows j=0..7)"); print(np.round(100*((mb-ma)/ma).reshape(nz,nr)[:8,::2],1))
print("z"); print(np.round(z.reshape(nz,nr)[:8,::2],1))
print("frac |z|>3:", np.mean(np.abs(z)>3), " mean z", z.mean(), " rms z", np.sqrt((z**2).mean()))
