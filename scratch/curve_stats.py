import numpy as np, sys
sys.path.insert(0,'/root/repo')
from pcl_amd import synth
n = 1<<20
pts = synth.gaussian_surface(n)[:, :3]
lo = pts.min(0); ex = (pts.max(0)-lo).max()
B = 21
q = np.minimum(((pts-lo)*(2.0**B/ex)).astype(np.uint64), 2**B-1)
def spread(v):
    x = v & np.uint64(0x1FFFFF)
    x = (x | x << np.uint64(32)) & np.uint64(0x1F00000000FFFF)
    x = (x | x << np.uint64(16)) & np.uint64(0x1F0000FF0000FF)
    x = (x | x << np.uint64(8)) & np.uint64(0x100F00F00F00F00F)
    x = (x | x << np.uint64(4)) & np.uint64(0x10C30C30C30C30C3)
    x = (x | x << np.uint64(2)) & np.uint64(0x1249249249249249)
    return x
def morton(q): return spread(q[:,0]) | (spread(q[:,1])<<np.uint64(1)) | (spread(q[:,2])<<np.uint64(2))
def hilbert(q):
    # Skilling: axes -> transpose, then interleave (x is most significant)
    X = [q[:,0].copy(), q[:,1].copy(), q[:,2].copy()]
    M = np.uint64(1) << np.uint64(B-1)
    Q = M
    while Q > 1:
        P = Q - np.uint64(1)
        for i in range(3):
            m = (X[i] & Q) != 0
            # invert low bits of X[0] where bit set
            X[0] = np.where(m, X[0] ^ P, X[0])
            # else exchange low bits of X[0] and X[i]
            t = (X[0] ^ X[i]) & P
            t = np.where(m, np.uint64(0), t)
            X[0] ^= t; X[i] ^= t
        Q >>= np.uint64(1)
    # Gray encode
    for i in range(1,3): X[i] ^= X[i-1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - np.uint64(1)), t)
        Q >>= np.uint64(1)
    for i in range(3): X[i] ^= t
    return (spread(X[0])<<np.uint64(2)) | (spread(X[1])<<np.uint64(1)) | spread(X[2])
for name, key in (("morton", morton(q)), ("hilbert", hilbert(q))):
    o = np.argsort(key, kind='stable')
    p = pts[o]
    for g in (16, 64, 1024):
        m = (n//g)*g
        pp = p[:m].reshape(-1, g, 3)
        d = np.linalg.norm(pp.max(1)-pp.min(1), axis=1)
        print(name, g, "diag mean %.4g median %.4g p99 %.4g p99.9 %.4g max %.4g  sum_area %.4g" % (d.mean(), np.median(d), np.percentile(d,99), np.percentile(d,99.9), d.max(), (np.prod(np.maximum((pp.max(1)-pp.min(1))[:, :2],1e-9),axis=1)).sum()))

def kd_order(pts, leaf=16, split=4):
    n = len(pts)
    nleaf = -(-n // leaf)
    R = 0
    while split**R < nleaf: R += 1
    order = np.arange(n)
    for r in range(1, R+1):
        seg_size = leaf * split**(R-r+1)   # segment = block being split in this round
        p = pts[order]
        seg = np.arange(n) // seg_size
        nseg = seg.max()+1
        lo = np.full((nseg,3), np.inf); hi = np.full((nseg,3), -np.inf)
        np.minimum.at(lo, seg, p); np.maximum.at(hi, seg, p)
        axis = np.argmax(hi-lo, axis=1)
        coord = p[np.arange(n), axis[seg]]
        o2 = np.lexsort((coord, seg))
        order = order[o2]
    return order
o = kd_order(pts)
p = pts[o]
for g in (16, 64, 1024):
    m = (n//g)*g
    pp = p[:m].reshape(-1, g, 3)
    d = np.linalg.norm(pp.max(1)-pp.min(1), axis=1)
    print("kd4", g, "diag mean %.4g median %.4g p99 %.4g p99.9 %.4g max %.4g  sum_area %.4g" % (d.mean(), np.median(d), np.percentile(d,99), np.percentile(d,99.9), d.max(), (np.prod(np.maximum((pp.max(1)-pp.min(1))[:, :2],1e-9),axis=1)).sum()))
