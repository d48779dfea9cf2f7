import numpy as np, time, sys
from mcfost_amd.host import model as M
from oracle import Oracle
cfg=M.small(n_rad=30,nz=20,dust_mass=1e-2)
N=int(float(sys.argv[1])) if len(sys.argv)>1 else 2000000
gam=float(sys.argv[2]) if len(sys.argv)>2 else 2.0
def run(mrw, seed, **kw):
    m=M.build_model(cfg)
    if mrw: M.init_mrw(m,weights=mrw,gamma=gam,**kw)
    o=Oracle(m,N)
    t=time.time(); r=o.run_thermal(N,seed=seed,n_threads=8); dt=time.time()-t
    return r["E_abs"], o.temp_finale(r["E_abs"]), dt, r["counters"]
Ea,Ta,ta,ca=run(None,5)
Eb,Tb,tb,cb=run(None,6)
print("brute",ta,tb)
nz,nr=20,30
for wts in ("dB_dT","B"):
    E1,T1,t1,c1=run(wts,7)
    print(wts,t1,c1["mrw_walks"],c1["mrw_steps"],c1["absorptions"])
    T0=0.5*(Ta+Tb)
    sig=np.abs(Ta-Tb)/np.sqrt(2)/T0  # ~ noise of one run (relative)
    dev=(T1-T0)/T0
    A=dev.reshape(nz,nr); S=sig.reshape(nz,nr)
    print("dev % rows j=0..5, cols ::3"); print(np.round(100*A[:6,::3],1))
    print("sig %"); print(np.round(100*S[:6,::3],1))
    deep=(slice(0,4),slice(2,20))
    print("mean dev deep %.4f  rms sig %.4f  total E ratio %.4f" % (A[deep].mean(), np.sqrt((S[deep]**2).mean()), E1.sum()/(0.5*(Ea.sum()+Eb.sum()))))
