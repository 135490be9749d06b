"""Companion of standoff_model.py: how tight the lanes' radii are after the FIRST batch of a stand-off list (the 16 leaves
nearest to the group, walked the sequential way) and how many entries of the rest of the list stay alive per 16-lane row
under the reach filter with those radii -- the numbers behind pair mode and its continuation
(section 8).  CPU only:  python scratch/standoff_phase1_model.py
"""
import sys, numpy as np
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.argv=['x','0.28','-0.22']
src=open(os.path.join(ROOT,'scratch','standoff_model.py')).read()
# reuse the setup part of the model up to the group loop
setup=src.split("res=[]")[0]
exec(setup)
from scipy.spatial import cKDTree
out=[]
for g in sg[::4]:
    rows=kdorder(src,g,16); q=src[np.concatenate(rows)]
    dd,_=tree.query(q); d2=dd**2
    Ql=q.min(0); Qh=q.max(0)
    gb=np.maximum(np.maximum(blo-Qh,Ql-bhi),0); lb_box=(gb**2).sum(1)
    order=np.argsort(lb_box)
    T=0.01
    first=order[:16]
    # radii after the first batch: best distance to any point of the first 16 leaves (what sequential evaluation converges to)
    dist_first=np.sqrt(((q[:,None,None,:]-LP[first][None,:,:,:])**2).sum(3)).min(2).min(1)
    rho1=dist_first
    tight=np.sqrt(d2)
    qq=q-q.mean(0); wv,Vv=np.linalg.eigh(qq.T@qq); ng=Vv[:,0]
    def row_alive(rho, sel, cand):
        qs_=q[sel]; r_=rho[sel]; lo=qs_.min(0); hi=qs_.max(0); ctr=0.5*(lo+hi); rS=0.5*np.linalg.norm(hi-lo)
        a_=(qs_-ctr)@ng; Up=(r_-a_).max(); Um=(r_+a_).max(); rmax=r_.max()
        cc=c[cand]; nn=n[cand]; al=nn@ng; mu=np.linalg.norm(nn-al[:,None]*ng[None,:],axis=1)*rS
        s0=((ctr-cc)*nn).sum(1); beta=np.sign(s0)*al
        reach=np.where(beta>=0,beta*Up+(1-beta)*rmax,(-beta)*Um+(1+beta)*rmax)-np.abs(s0)+mu+hn[cand]
        dc=ctr-cc; gt=np.maximum(np.sqrt(np.maximum((dc**2).sum(1)-((nn*dc).sum(1))**2,0))-(rS+R[cand]),0)
        return (reach>=0)&(gt**2<=2*rmax*reach)
    rest=order[16:]
    rest=rest[lb_box[rest]<=(rho1.max())**2]
    a1=[row_alive(rho1,np.arange(16*r,16*r+16),rest).sum() for r in range(4)]
    at=[row_alive(tight,np.arange(16*r,16*r+16),rest).sum() for r in range(4)]
    # alternative first batch: per row the 4 leaves nearest to the row box (16 leaves total)
    firstR=[]
    for r in range(4):
        qr=q[16*r:16*r+16]; gl=np.maximum(np.maximum(blo-qr.max(0),qr.min(0)-bhi),0); firstR+=list(np.argsort((gl**2).sum(1))[:4])
    firstR=np.array(sorted(set(firstR)))
    rho2=np.sqrt(((q[:,None,None,:]-LP[firstR][None,:,:,:])**2).sum(3)).min(2).min(1)
    out.append((len(rest),max(a1),np.mean(a1),max(at),np.mean(at),(rho1/tight).max(),(rho1/tight).mean(),(rho2/tight).max(),(rho2/tight).mean(),len(firstR)))
r=np.array(out)
for i,nm in enumerate(["rest list (box-alive after batch 0)","row alive max (radii after batch 0)","row alive mean","row alive max (tight radii)","row alive mean (tight)","rho/tight max (group-sorted first 16)","rho/tight mean","rho/tight max (row-nearest 4x4 first)","rho/tight mean","distinct leaves in row-nearest set"]):
    print("%-44s mean %.3f p90 %.3f"%(nm,r[:,i].mean(),np.percentile(r[:,i],90)))
