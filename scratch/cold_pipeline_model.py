"""Numpy model of the round-3 cold (stand-off) search pipeline: seed -> collect -> row seeds -> row/quad filter -> lane tests.
Reuses the setup of standoff_model.py.   python scratch/cold_pipeline_model.py [cx cy]
"""
import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
args=sys.argv[1:] if len(sys.argv)>2 else ['0.28','-0.22']
sys.argv=['x']+args
src_txt=open(os.path.join(ROOT,'scratch','standoff_model.py')).read()
exec(src_txt.split("res=[]")[0])
from scipy.spatial import cKDTree
def reach_alive(q, rho, sel, ng, cand):
    qs_=q[sel]; r_=rho[sel]; lo=qs_.min(0); hi=qs_.max(0); ctr=0.5*(lo+hi); rS=0.5*np.linalg.norm(hi-lo)
    a_=(qs_-ctr)@ng; Up=(r_-a_).max(); Um=(r_+a_).max(); rmax=r_.max()
    cc=c[cand]; nn=n[cand]; al=nn@ng; mu=np.linalg.norm(nn-al[:,None]*ng[None,:],axis=1)*rS
    s0=((ctr-cc)*nn).sum(1); beta=np.sign(s0)*al
    reach=np.where(beta>=0,beta*Up+(1-beta)*rmax,(-beta)*Um+(1+beta)*rmax)-np.abs(s0)+mu+hn[cand]
    dc=ctr-cc; gt=np.maximum(np.sqrt(np.maximum((dc**2).sum(1)-((nn*dc).sum(1))**2,0))-(rS+R[cand]),0)
    return (reach>=0)&(gt**2<=2*rmax*reach)
def lane_lb(q, cand):
    qp=q[:,None,:]-c[cand][None,:,:]
    r2l=(qp**2).sum(2); al=np.abs((qp*n[cand][None]).sum(2))
    gtl=np.maximum(np.sqrt(np.maximum(r2l-al**2,0))-R[cand][None],0); gnl=np.maximum(al-hn[cand][None],0)
    return gtl**2+gnl**2
out=[]
for g in sg[::4]:
    rows=kdorder(src,g,16); q=src[np.concatenate(rows)]
    dd,_=tree.query(q); tight=dd
    Ql=q.min(0); Qh=q.max(0); Qc=0.5*(Ql+Qh)
    # (a) seed: exact NN of the group centre (or of lane 0)
    d0,i0=tree.query(Qc); p0=tgt[i0]
    rho_a=np.linalg.norm(q-p0,axis=1)
    # (b) list: box-alive at T=max rho_a^2
    gb=np.maximum(np.maximum(blo-Qh,Ql-bhi),0); lb_box=(gb**2).sum(1)
    Ta=(rho_a.max())**2
    listA=np.nonzero(lb_box<=Ta)[0]
    listT=np.nonzero(lb_box<=(tight.max())**2)[0]
    # (c) row seeds: per row the m leaves of the list nearest (centre distance) to the row centre
    qq=q-q.mean(0); wv,Vv=np.linalg.eigh(qq.T@qq); ng=Vv[:,0]
    rho_c=rho_a.copy(); 
    for m in (2,):
        for r in range(4):
            sel=np.arange(16*r,16*r+16); ctr=0.5*(q[sel].min(0)+q[sel].max(0))
            dcen=((c[listA]-ctr)**2).sum(1); pick=listA[np.argsort(dcen)[:m]]
            dmin=np.sqrt(((q[sel][:,None,None,:]-LP[pick][None])**2).sum(3)).min(2).min(1)
            rho_c[sel]=np.minimum(rho_c[sel],dmin)
    # (d) filters with rho_c
    rowal=[reach_alive(q,rho_c,np.arange(16*r,16*r+16),ng,listA) for r in range(4)]
    rowal_a=[reach_alive(q,rho_a,np.arange(16*r,16*r+16),ng,listA) for r in range(4)]
    rowcnt=[x.sum() for x in rowal]; rowcnt_a=[x.sum() for x in rowal_a]
    lb=lane_lb(q,listA)
    need_c=lb<=rho_c[:,None]**2
    need_t=lb<=tight[:,None]**2
    need_a=lb<=rho_a[:,None]**2
    # restrict lane need to row-alive
    nc=[]; 
    for r in range(4):
        sel=np.arange(16*r,16*r+16)
        nc.append(need_c[sel][:,rowal[r]].sum(1))
    nc=np.concatenate(nc)
    # quads: spatial quadrants of the row
    quadcnt=[]
    t1=Vv[:,2]; t2=Vv[:,1]
    for r in range(4):
        sel=np.arange(16*r,16*r+16); ctr=q[sel].mean(0); d_=q[sel]-ctr
        quad=((d_@t1)>0).astype(int)*2+((d_@t2)>0).astype(int)
        for qd in range(4):
            qs=sel[quad==qd]
            if len(qs)==0: continue
            qa=reach_alive(q,rho_c,qs,ng,listA)&rowal[r]; quadcnt.append(qa.sum())
    # after evaluating each lane's min-lb leaf (among needed): tight?  model as tight
    out.append((len(listT),len(listA),(rho_a/tight).mean(),(rho_a/tight).max(),(rho_c/tight).mean(),(rho_c/tight).max(),
                np.mean(rowcnt_a),np.max(rowcnt_a),np.mean(rowcnt),np.max(rowcnt),np.mean(quadcnt),np.max(quadcnt),
                need_a.sum(1).mean(),need_a.sum(1).max(),nc.mean(),nc.max(),need_t.sum(1).mean(),need_t.sum(1).max(),
                np.any(np.array(rowal),axis=0).sum()))
r=np.array(out)
names=["list @tight T","list @seed T","rho_a/tight mean","rho_a/tight max","rho_c/tight mean","rho_c/tight max",
"row alive mean (seed rho)","row alive max (seed rho)","row alive mean (row-seed rho)","row alive max (row-seed rho)","quad alive mean","quad alive max",
"need/lane mean (seed rho)","need/lane max (seed rho)","need/lane mean (row-seed)","need/lane max (row-seed)","need/lane mean (tight)","need/lane max (tight)","row union"]
for i,nm in enumerate(names): print("%-36s mean %.3f p90 %.3f"%(nm,r[:,i].mean(),np.percentile(r[:,i],90)))
