"""Hybrid (MRW inside + flights near the wall) vs brute force, homogeneous sphere of optical radius R (Rosseland)."""
import numpy as np, sys
from mcfost_amd.host import model as M
cfg=M.small(n_rad=30,nz=20,dust_mass=1e-2)
m=M.build_model(cfg); M.init_mrw(m)
T=float(sys.argv[1]); R0=float(sys.argv[2]); gam=float(sys.argv[3]); mode=sys.argv[4] if len(sys.argv)>4 else "cos"
ti=int(np.argmin(np.abs(m.tab_Temp-T)))
cdf=m.kdB_dT_CDF.reshape(m.tab_Temp.size,-1)[ti]
kext=np.asarray(m.kappa,float); alb=np.asarray(m.albedo,float); kabs=np.asarray(m.kappa_abs_LTE,float)
wl=np.asarray(m.lam,float)*1e-6; dwl=np.asarray(m.delta_lam,float)*1e-6
cw=M.THERMAL_CONST/float(m.tab_Temp[ti])/wl; ce=np.exp(np.minimum(cw,500)); wgt=1/(wl**5*(ce-1))*dwl*cw*ce/(ce-1)
chiR=wgt.sum()/(wgt/kext).sum(); kdep=(wgt*kabs).sum()/wgt.sum()
import os
ze=float(os.environ.get('ZE','0'))*0.7104*(wgt/kext**2).sum()/(wgt/kext).sum()
xs=os.environ.get('XS','em')
cdfx=cdf if xs=='em' else (np.cumsum(wgt)/wgt.sum() if xs=='w' else np.cumsum(wgt/kext)/(wgt/kext).sum())
zeta=m.mrw["zeta"]; yg=np.arange(zeta.size)/(zeta.size-1)
rng=np.random.default_rng(3)
def iso(n):
    w=rng.uniform(-1,1,n); ph=rng.uniform(0,2*np.pi,n); s=np.sqrt(1-w*w)
    return np.stack([s*np.cos(ph),s*np.sin(ph),w],1)
def coslaw(nrm):
    n=nrm.shape[0]; ct=np.sqrt(rng.random(n)); st=np.sqrt(1-ct*ct); ph=rng.uniform(0,2*np.pi,n)
    a=np.where(np.abs(nrm[:,[2]])<0.9, np.array([[0,0,1.0]]), np.array([[1.0,0,0]]))
    e1=np.cross(nrm,a); e1/=np.linalg.norm(e1,axis=1)[:,None]; e2=np.cross(nrm,e1)
    return nrm*ct[:,None]+e1*(st*np.cos(ph))[:,None]+e2*(st*np.sin(ph))[:,None]
R=R0/chiR
def run(n, hybrid):
    # uniform start positions inside the sphere
    pos=iso(n)*(R*rng.random(n)**(1/3))[:,None]; lam=np.searchsorted(cdf,rng.random(n)); dirs=iso(n)
    dep=np.zeros(n); alive=np.ones(n,bool); fresh=np.ones(n,bool); inwalk=np.zeros(n,bool)
    nmrw=0
    while alive.any():
        idx=np.nonzero(alive)[0]
        fly=np.ones(idx.size,bool)
        if hybrid:
            d=R-np.linalg.norm(pos[idx],axis=1)
            go=(fresh[idx]|inwalk[idx])&(d*chiR>gam)
            ii=idx[go]; dd=d[go]
            if ii.size:
                stp=iso(ii.size); pos[ii]+=stp*dd[:,None]
                y=np.maximum(np.interp(rng.random(ii.size),zeta,yg),1e-300)
                dep[ii]+=kdep*(-np.log(y)*3/np.pi**2*chiR*(dd+ze)**2); nmrw+=ii.size
                inwalk[ii]=True; fresh[ii]=False
                d2=R-np.linalg.norm(pos[ii],axis=1)
                end=d2*chiR<=gam
                done=ii[end]
                lam[done]=np.searchsorted(cdfx,rng.random(done.size))
                dirs[done]=coslaw(stp[end]) if mode=="cos" else iso(done.size)
                inwalk[done]=False
                fly[go]=False
                fly[np.nonzero(go)[0][end]]=True
        idx=idx[fly]
        if idx.size==0: continue
        l=-np.log(1-rng.random(idx.size))/kext[lam[idx]]
        p0=pos[idx]; u=dirs[idx]
        b=(p0*u).sum(1); c=(p0*p0).sum(1)-R*R
        lx=-b+np.sqrt(np.maximum(b*b-c,0))
        out=l>=lx; lm=np.where(out,lx,l)
        dep[idx]+=kabs[lam[idx]]*lm
        pos[idx]=p0+u*lm[:,None]
        alive[idx[out]]=False
        ii=idx[~out]
        sc=rng.random(ii.size)<alb[lam[ii]]
        dirs[ii]=iso(ii.size)
        fresh[ii]=~sc
        ab=ii[~sc]; lam[ab]=np.searchsorted(cdf,rng.random(ab.size))
    return dep, nmrw
nb=int(sys.argv[5]) if len(sys.argv)>5 else 3000
db,_=run(nb,False); dh,nm=run(4*nb,True)
print("R0",R0,"gamma",gam,mode,"brute dep %.2f +- %.2f | hybrid %.2f +- %.2f  ratio %.4f  steps/packet %.2f"%(db.mean(),db.std()/np.sqrt(db.size),dh.mean(),dh.std()/np.sqrt(dh.size),dh.mean()/db.mean(),nm/(4*nb)))
