"""Numpy model of the unseeded (stand-off) search of the bench workload -- the numbers DESIGN.md sections 3 and 8 quote.

Takes a window of the 10M-point target and of the (not yet aligned) source around (cx, cy), orders both by recursive
median cuts into 16-point leaves / 64-query groups (the same shapes the device's kd order produces), builds the leaf
discs, and counts for every 64-query group, with TIGHT radii (each query's true nearest-neighbour distance):
  cur_alive / box_alive  leaves the (removed) group-level disc bound / the box bound keep alive
  new_alive, pca_*       a group bound with an exact interval along a group direction
  W1 / W2                the tilt-compensated reach filter (group direction from PCA / from the nearest leaf)
  row_alive_*            the same per 16-lane row
  union_need, need/lane  leaves some lane really needs (per-lane disc bound against its own radius)
  row / quad / lane pairs   hierarchical pair culling with the reach filter: tests and survivors per level
"MISSED" must print 0 and "lane pairs needed (found)" must equal "(all)": the reach filter never drops a leaf a lane needs.

    python scratch/standoff_model.py 0.28 -0.22
"""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from pcl_amd import synth
from scipy.spatial import cKDTree
N=10_000_000
def window(seed, lo, hi, inv=None):
    out=[]
    ch=1<<22
    for b in range(0,N,ch):
        p=synth.gaussian_surface(min(ch,N-b),seed,b)
        if inv is not None: p=synth.apply_rigid(inv,p)
        m=(p[:,0]>=lo[0])&(p[:,0]<hi[0])&(p[:,1]>=lo[1])&(p[:,1]<hi[1])
        out.append(p[m,:3].astype(np.float64))
    return np.concatenate(out)
cx,cy=float(sys.argv[1]),float(sys.argv[2])
tgt=window(synth.TARGET_SEED,(cx-0.1,cy-0.1),(cx+0.1,cy+0.1))
inv=np.linalg.inv(synth.ground_truth_transform())
src=window(synth.SOURCE_SEED,(cx-0.04,cy-0.04),(cx+0.04,cy+0.04),inv)
print(len(tgt),len(src))
def kdorder(P, idx, leaf):
    # recursive median split along widest axis until <= leaf
    if len(idx)<=leaf: return [idx]
    p=P[idx]; ax=np.argmax(p.max(0)-p.min(0))
    # split at multiple of leaf
    k=(len(idx)//2//leaf)*leaf
    if k==0: k=leaf
    o=np.argpartition(p[:,ax],k-1)  # approx
    o2=np.argsort(p[:,ax],kind='stable')
    return kdorder(P,idx[o2[:k]],leaf)+kdorder(P,idx[o2[k:]],leaf)
tl=kdorder(tgt,np.arange(len(tgt)),16)
tl=[l for l in tl if len(l)==16]
L=np.array(tl)              # nleaf x16
LP=tgt[L]                   # nleaf,16,3
c=LP.mean(1); d=LP-c[:,None,:]
A=np.einsum('lij,lik->ljk',d,d); w,V=np.linalg.eigh(A); n=V[:,:,0]
R=np.sqrt((d**2).sum(2).max(1)); hn=np.abs(np.einsum('lij,lj->li',d,n)).max(1)
blo=LP.min(1); bhi=LP.max(1)
sg=kdorder(src,np.arange(len(src)),64); sg=[g for g in sg if len(g)==64]
tree=cKDTree(tgt)
res=[]
for g in sg[::4]:
    # order within group: 4 rows of 16 by kd
    rows=kdorder(src,g,16)
    q=src[np.concatenate(rows)]
    dd,_=tree.query(q); d2=dd**2; T=d2.max()
    Ql=q.min(0); Qh=q.max(0); Qc=0.5*(Ql+Qh); rQ=0.5*np.linalg.norm(Qh-Ql)
    # current group disc lb
    l=Ql-c; h=Qh-c
    a=n*l; b=n*h
    smin=np.minimum(a,b).sum(1); smax=np.maximum(a,b).sum(1)
    gn=np.maximum(np.maximum(smin,-smax)-hn,0)
    dc=Qc-c; r2=(dc**2).sum(1); ah=np.abs((n*dc).sum(1)); b2=np.maximum(r2-ah**2,0)
    gt=np.maximum(np.sqrt(b2)-(rQ+R),0)
    lb_cur=gt**2+gn**2
    # box bound too
    gb=np.maximum(np.maximum(blo-Qh,Ql-bhi),0); lb_box=(gb**2).sum(1)
    lb_cur=np.maximum(lb_cur,lb_box)
    # new: group direction from row centroids
    rc=np.array([q[16*i:16*i+16].mean(0) for i in range(4)])
    # pick best cross among pairs
    ng=np.cross(rc[3]-rc[0],rc[2]-rc[1]); nn=np.linalg.norm(ng)
    ng=ng/nn if nn>0 else np.zeros(3)
    # tighter centre: centroid, sphere radius about centroid
    qm=q.mean(0); rS=np.sqrt(((q-qm)**2).sum(1).max())
    aq=(q-qm)@ng; amin,amax=aq.min(),aq.max()
    alpha=n@ng; m=n-alpha[:,None]*ng[None,:]; mlen=np.linalg.norm(m,axis=1)
    s0=((qm-c)*n).sum(1)
    lo=s0+np.minimum(alpha*amin,alpha*amax)-mlen*rS; hi=s0+np.maximum(alpha*amin,alpha*amax)+mlen*rS
    gn2=np.maximum(np.maximum(lo,-hi)-hn,0)
    dc2=qm-c; r22=(dc2**2).sum(1); ah2=np.abs((n*dc2).sum(1)); gt2=np.maximum(np.sqrt(np.maximum(r22-ah2**2,0))-(rS+R),0)
    lb_new=np.maximum(gt2**2+gn2**2,lb_box)
    # V3: PCA normal
    qq=q-qm; wv,Vv=np.linalg.eigh(qq.T@qq); ng3=Vv[:,0]
    aq3=qq@ng3; alpha3=n@ng3; m3=np.linalg.norm(n-alpha3[:,None]*ng3[None,:],axis=1)
    lo3=s0+np.minimum(alpha3*aq3.min(),alpha3*aq3.max())-m3*rS; hi3=s0+np.maximum(alpha3*aq3.min(),alpha3*aq3.max())+m3*rS
    gn3=np.maximum(np.maximum(lo3,-hi3)-hn,0); lb3=np.maximum(gt2**2+gn3**2,lb_box)
    # V4: tangential extents in group frame instead of sphere: |m.w| <= |m.t1|e1+|m.t2|e2 using PCA tangents
    t1=Vv[:,2]; t2=Vv[:,1]; e1=np.abs(qq@t1).max(); e2=np.abs(qq@t2).max()
    mv=n-alpha3[:,None]*ng3[None,:]
    sl=np.abs(mv@t1)*e1+np.abs(mv@t2)*e2
    lo4=s0+np.minimum(alpha3*aq3.min(),alpha3*aq3.max())-sl; hi4=s0+np.maximum(alpha3*aq3.min(),alpha3*aq3.max())+sl
    gn4=np.maximum(np.maximum(lo4,-hi4)-hn,0); lb4=np.maximum(gt2**2+gn4**2,lb_box)
    # rows: V1 bound per row with row T
    rowalive=[]; anyrow=np.zeros(len(c),bool)
    for r_ in range(4):
        qr=q[16*r_:16*r_+16]; Tr=d2[16*r_:16*r_+16].max(); qmr=qr.mean(0); rSr=np.sqrt(((qr-qmr)**2).sum(1).max())
        aqr=(qr-qmr)@ng; s0r=((qmr-c)*n).sum(1)
        lor=s0r+np.minimum(alpha*aqr.min(),alpha*aqr.max())-mlen*rSr; hir=s0r+np.maximum(alpha*aqr.min(),alpha*aqr.max())+mlen*rSr
        gnr=np.maximum(np.maximum(lor,-hir)-hn,0); dcr=qmr-c
        gtr=np.maximum(np.sqrt(np.maximum((dcr**2).sum(1)-((n*dcr).sum(1))**2,0))-(rSr+R),0)
        al_=(gtr**2+gnr**2)<=Tr; rowalive.append(al_.sum()); anyrow|=al_
    # W: tilt-compensated
    rho=np.sqrt(d2); rmax=rho.max()
    def wbound(ngv):
        a_=(q-qm)@ngv; Up=(rho-a_).max(); Um=(rho+a_).max()
        al=n@ngv; mu=np.linalg.norm(n-al[:,None]*ngv[None,:],axis=1)*rS
        s0_=((qm-c)*n).sum(1); beta=np.sign(s0_)*al
        reach=np.where(beta>=0,beta*Up+(1-beta)*rmax,(-beta)*Um+(1+beta)*rmax)-np.abs(s0_)+mu+hn
        return (reach>=0)&(gt2**2<=2*rmax*reach)&(lb_box<=T)
    w1=wbound(ng3)
    near=np.argmin(((c-qm)**2).sum(1)); w2=wbound(n[near])
    w12=w1&w2&(lb_new<=T)
    qp=q[:,None,:]-c[None,:,:]
    r2l=(qp**2).sum(2); al=np.abs((qp*n[None]).sum(2))
    gtl=np.maximum(np.sqrt(np.maximum(r2l-al**2,0))-R[None],0); gnl=np.maximum(al-hn[None],0)
    lbl=gtl**2+gnl**2
    gbl=np.maximum(np.maximum(blo[None]-q[:,None,:],q[:,None,:]-bhi[None]),0); lbbl=(gbl**2).sum(2)
    lbl=np.maximum(lbl,lbbl)
    need=(lbl<=d2[:,None])
    needbox=(lbbl<=d2[:,None])
    # hierarchical pair culling: the reach filter with the quantities of a sub-group of lanes
    def reach_alive(sel):
        qs_=q[sel]; rho_=rho[sel]; qm_=qs_.mean(0); rS_=np.sqrt(((qs_-qm_)**2).sum(1).max())
        a_=(qs_-qm_)@ng3; Up_=(rho_-a_).max(); Um_=(rho_+a_).max(); rmax_=rho_.max()
        al=n@ng3; mu=np.linalg.norm(n-al[:,None]*ng3[None,:],axis=1)*rS_
        s0_=((qm_-c)*n).sum(1); beta=np.sign(s0_)*al
        reach=np.where(beta>=0,beta*Up_+(1-beta)*rmax_,(-beta)*Um_+(1+beta)*rmax_)-np.abs(s0_)+mu+hn
        dc_=qm_-c; gt_=np.maximum(np.sqrt(np.maximum((dc_**2).sum(1)-((n*dc_).sum(1))**2,0))-(rS_+R),0)
        return (reach>=0)&(gt_**2<=2*rmax_*reach)
    row_pairs=0; quad_pairs=0; lane_pairs=0; quad_tests=0; lane_tests=0
    for r_ in range(4):
        rs=np.arange(16*r_,16*r_+16); ra=reach_alive(rs); row_pairs+=ra.sum()
        # quads: kd order inside a 16-leaf is not 2x2 spatial; model them by the 4 spatial quadrants of the row
        ctr=q[rs].mean(0); d_=q[rs]-ctr; t1=Vv[:,2]; t2=Vv[:,1]
        quad=((d_@t1)>0).astype(int)*2+((d_@t2)>0).astype(int)
        for qd in range(4):
            qsel=rs[quad==qd]
            if len(qsel)==0: continue
            quad_tests+=ra.sum()
            qa=reach_alive(qsel)&ra; quad_pairs+=qa.sum()
            lane_tests+=qa.sum()*len(qsel)
            lane_pairs+=(need[qsel][:,qa]).sum()
    missed=0
    res.append((np.sqrt(T),(lb_cur<=T).sum(),(lb_new<=T).sum(),need.any(0).sum(),need.sum(1).mean(),need.sum(1).max(),(lb_box<=T).sum(),needbox.any(0).sum(),needbox.sum(1).mean(),(lb3<=T).sum(),(lb4<=T).sum(),np.mean(rowalive),np.max(rowalive),anyrow.sum(),w1.sum(),w2.sum(),w12.sum(),(w2&~need.any(0)).sum(), (need.any(0)&~w2).sum(), row_pairs, quad_tests, quad_pairs, lane_tests, lane_pairs, need.sum()))
r=np.array(res)
print("groups",len(r))
print("standoff mean %.4f"%r[:,0].mean())
for i,nm in enumerate(["cur_alive","new_alive","union_need","need/lane mean","need/lane max","box_alive","box_union_need","boxneed/lane","pca_alive","pca_tan_alive","row_alive_mean","row_alive_max","row_union","W1 pca","W2 nearleaf","W1&W2&V1","W2 false+","W2 MISSED(must be 0)","row pairs alive (of 4 x list)","quad-level tests","quad pairs alive","lane-level tests","lane pairs needed (found)","lane pairs needed (all)"]):
    print("%-16s mean %.1f  p50 %.1f p90 %.1f"%(nm,r[:,i+1].mean(),np.median(r[:,i+1]),np.percentile(r[:,i+1],90)))
