import numpy as np, time, sys, os
from mcfost_amd.host import model as M
from oracle import Oracle
ref=np.load("gpurun_out/r2/mrw_ref.npz")
Nref=int(ref["N"])
cfg=M.small(n_rad=30,nz=20,dust_mass=1e-2)
N=int(float(sys.argv[1])); gam=float(sys.argv[2]); ns=int(sys.argv[3]) if len(sys.argv)>3 else 3
prior=ref["live1"]  # E_abs of N_ref packets: L_packet scaling: oracle built with N total -> scale prior
f1,f2=ref["frozen1"],ref["frozen2"]
Eref=0.5*(f1+f2)*(N/Nref); sref=np.abs(f1-f2)/2*(N/Nref)
def run(seed):
    m=M.build_model(cfg); M.init_mrw(m,gamma=gam,n_inter=int(os.environ.get('NINTER','5')),ext_factor=float(os.environ.get('ZE','0.4')))
    if os.environ.get('G0'): m.mrw['chi']=m.mrw['chi']*float(os.environ['G0'])
    o=Oracle(m,N)
    r=o.run_thermal(N,seed=seed,n_threads=8,frozen=True,E_prior=prior*(N/Nref))
    return r["E_abs"], r["counters"]
t=time.time()
B=np.array([run(30+s)[0] for s in range(ns)]); print("time",time.time()-t)
mb=B.mean(0); se=np.sqrt(B.var(0,ddof=1)/ns+sref**2)
nz,nr=20,30
bias=((mb-Eref)/Eref).reshape(nz,nr); z=((mb-Eref)/se).reshape(nz,nr)
np.set_printoptions(linewidth=200)
print("E bias %"); print(np.round(100*bias[:8,::2],1))
print("se %"); print(np.round(100*(se/Eref).reshape(nz,nr)[:8,::2],1))
print("deep mean bias %.4f ; all-cell rms bias %.4f; E-weighted bias %.5f" % (bias[:5,2:22].mean(), np.sqrt((bias**2).mean()), (mb.sum()-Eref.sum())/Eref.sum()))
