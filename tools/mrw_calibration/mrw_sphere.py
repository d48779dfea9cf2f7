"""Brute-force random walk out of a homogeneous sphere vs the MRW estimate (non-gray, complete redistribution)."""
import numpy as np, sys
from mcfost_amd.host import model as M
cfg=M.small(n_rad=30,nz=20,dust_mass=1e-2)
m=M.build_model(cfg); M.init_mrw(m)
T=float(sys.argv[1]) if len(sys.argv)>1 else 100.0
ti=int(np.argmin(np.abs(m.tab_Temp-T)))
chiR=m.mrw["chi"][ti]; kdep=m.mrw["kappa_dep"][ti]
cdf=m.kdB_dT_CDF.reshape(m.tab_Temp.size,-1)[ti]
kext=np.asarray(m.kappa,float); alb=np.asarray(m.albedo,float); g=np.asarray(m.tab_g_pos,float)
kabs=np.asarray(m.kappa_abs_LTE,float)
print("T",m.tab_Temp[ti],"chiR",chiR,"kdep",kdep, "kabs/kext check", np.abs(kabs-kext*(1-alb)).max())
rng=np.random.default_rng(1)
def iso(n):
    w=rng.uniform(-1,1,n); ph=rng.uniform(0,2*np.pi,n); s=np.sqrt(1-w*w)
    return np.stack([s*np.cos(ph),s*np.sin(ph),w],1)
# isotropic-scattering Rosseland mean (the walk below scatters isotropically)
wl=np.asarray(m.lam,float)*1e-6; dwl=np.asarray(m.delta_lam,float)*1e-6
cw=M.THERMAL_CONST/float(m.tab_Temp[ti])/wl; ce=np.exp(np.minimum(cw,500)); wgt=1/(wl**5*(ce-1))*dwl*cw*ce/(ce-1)
chiR=wgt.sum()/(wgt/kext).sum(); print("chiR iso",chiR, "g range",g.min(),g.max(), "albedo", alb.min(), alb.max())
for tau0 in (6.0,20.0,60.0):
    d=tau0/chiR
    n=4000
    pos=np.zeros((n,3)); lam=np.searchsorted(cdf,rng.random(n)); dirs=iso(n)
    path=np.zeros(n); dep=np.zeros(n); alive=np.ones(n,bool); nem=np.ones(n)
    it=0
    while alive.any():
        idx=np.nonzero(alive)[0]
        l=-np.log(1-rng.random(idx.size))/kext[lam[idx]]
        p0=pos[idx]; u=dirs[idx]
        # distance to sphere
        b=(p0*u).sum(1); c=(p0*p0).sum(1)-d*d
        lx=-b+np.sqrt(np.maximum(b*b-c,0))
        out=l>=lx
        lm=np.where(out,lx,l)
        path[idx]+=lm; dep[idx]+=kabs[lam[idx]]*lm
        pos[idx]=p0+u*lm[:,None]
        alive[idx[out]]=False
        ii=idx[~out]
        sc=rng.random(ii.size)<alb[lam[ii]]
        # isotropic scattering (test with g=0 transport)
        dirs[ii]=iso(ii.size)
        ab=ii[~sc]
        lam[ab]=np.searchsorted(cdf,rng.random(ab.size)); nem[ab]+=1
        it+=1
    # MRW estimate with isotropic-scattering transport opacity
    w_chi=None
    print("tau0",tau0,"brute <path*chiR>",(path.mean()*chiR),"MRW",tau0**2/2,"| brute dep",dep.mean(),"MRW dep",kdep*tau0**2/2/chiR, "n_em",nem.mean())
