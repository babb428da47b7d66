import sys, math, random, numpy as np
sys.path.insert(0,'/root/repo')
from rvpt_amd import native, scene
from rvpt_amd.camera import Camera

def build_wide(nodes, width):
    # nodes: structured first,count,bounds (breadth-first device layout not needed for counting)
    first = nodes['first']; count = nodes['count']; b = nodes['bounds']
    def area(i):
        d = (b[i,1]-b[i,0], b[i,3]-b[i,2], b[i,5]-b[i,4]); return d[0]*d[1]+d[1]*d[2]+d[2]*d[0]
    kids = {}
    queue=[0]; qi=0
    while qi < len(queue):
        w = queue[qi]; qi+=1
        c=[int(first[w]), int(first[w])+1]
        while len(c) < width:
            pick=-1; best=-1
            for i,x in enumerate(c):
                if count[x]>0: continue
                a=area(x)
                if a>best: best=a; pick=i
            if pick<0: break
            f=int(first[c[pick]]); c[pick:pick+1]=[f,f+1]
        kids[w]=c
        for x in c:
            if count[x]==0: queue.append(x)
    return kids

def quant_boxes(nodes, kids):
    """The 64-byte form's child boxes (bvh_wide.cpp: build_quant_nodes): per wide node origin = the children's smallest minimum, scale = a power of two,
    8-bit offsets rounded outward.  -> {child binary index: conservative bounds[6]} (a binary node is the child of exactly one wide node)."""
    b = nodes['bounds'].astype(np.float64); out = {}
    for w, c in kids.items():
        for ax in range(3):
            lo = min(b[x, 2*ax] for x in c); hi = max(b[x, 2*ax+1] for x in c)
            sc = 2.0 ** math.ceil(math.log2((hi - lo) / 255.0)) if hi > lo else 2.0 ** -100
            while max(math.ceil((b[x, 2*ax+1] - lo) / sc) for x in c) > 255: sc *= 2.0
            for x in c:
                q = out.setdefault(x, [0.0] * 6)
                q[2*ax] = lo + math.floor((b[x, 2*ax] - lo) / sc) * sc
                q[2*ax+1] = lo + math.ceil((b[x, 2*ax+1] - lo) / sc) * sc
    return out

def trace(nodes, kids, tri_test, o, d, qbox=None):
    b = nodes['bounds']; count = nodes['count']; first=nodes['first']
    inv = [1.0/x if x!=0 else math.inf for x in d]
    closest = math.inf
    st = dict(steps=0, boxes=0, leaves=0, tris=0, pushes=0, pops=0, leaf_fail=0)
    def slab(i, closest, conservative=False):
        t0=0.0; t1=closest
        bb = qbox[i] if (conservative and qbox is not None) else b[i]
        for ax in range(3):
            lo=(bb[2*ax]-o[ax])*inv[ax]; hi=(bb[2*ax+1]-o[ax])*inv[ax]
            if lo>hi: lo,hi=hi,lo
            if lo>t0: t0=lo
            if hi<t1: t1=hi
        return t1>=t0, t0
    ok,_ = slab(0, closest)
    if not ok: return st, closest
    stack=[]; cur=0
    while True:
        if count[cur]>0:
            st['leaves']+=1
            leaf_ok = True
            if qbox is not None:  # the leaf's own exact box at its visit
                leaf_ok,_ = slab(cur, closest)
                if not leaf_ok: st['leaf_fail']+=1
            for k in (range(int(first[cur]), int(first[cur])+int(count[cur])) if leaf_ok else ()):
                st['tris']+=1
                t = tri_test(k,o,d)
                if t is not None and t<closest: closest=t
            # pop
            found=False
            while stack:
                e,n = stack.pop(); st['pops']+=1
                if closest>=e: cur=n; found=True; break
            if not found: break
            continue
        st['steps']+=1
        c = kids[cur]
        passed=[]
        for x in c:
            st['boxes']+=1
            ok,e = slab(x, closest, conservative=True)
            if ok: passed.append((e,x))
        if passed:
            for e,x in reversed(passed[1:]):
                stack.append((e,x)); st['pushes']+=1
            cur = passed[0][1]
        else:
            found=False
            while stack:
                e,n = stack.pop(); st['pops']+=1
                if closest>=e: cur=n; found=True; break
            if not found: break
    return st, closest

def main(which):
    if which=='cornell':
        tris,mats = scene.cornell_scene(); campos=(0.0,2.0,-1.9); rot=(0,0,0)
    elif which=='heightfield':
        tris,mats = scene.heightfield_scene(); campos=(0.0,2.5,-5.0); rot=(0,25.0,0)
    else:
        tris,mats = scene.default_scene(); campos=(0,0,0); rot=(0,0,0)
    nodes_raw, idx = native.build_bvh(tris)
    nodes = nodes_raw.view(native.NODE_DTYPE).reshape(-1)
    T = tris[idx].reshape(-1,4,4)[:,:3,:3].astype(np.float64)
    def tri_test(k,o,d):
        v0=T[k,0]; e0=T[k,1]-v0; e1=T[k,2]-v0
        pv=np.cross(d,e1); det=pv@e0
        if abs(det)<1e-300: return None
        tv=np.array(o)-v0; u=(pv@tv)/det
        if u<=0: return None
        qv=np.cross(tv,e0); v=(np.array(d)@qv)/det
        if v<=0 or u+v>=1: return None
        t=(qv@e1)/det
        return t if t>0 else None
    W,H=1920,1080
    c=Camera(W/H); c.translation=np.array(campos,float); c.rotation=np.array(rot,float)
    cam=c.get_data(); M=cam[:16].reshape(4,4).T.astype(float)
    rng=random.Random(1)
    rays=[]
    for _ in range(N):
        cx=rng.random(); cy=rng.random()
        u=cam[16]*(2*cx-1); v=2*cy-1; w=1/math.tan(0.5*cam[17])
        d=M[:3,0]*u+M[:3,1]*v+M[:3,2]*w; d/=np.linalg.norm(d)
        rays.append((tuple(M[:3,3]), tuple(d)))
    res={}
    for width in (2,4,8,'4q'):
        kids=build_wide(nodes,4 if width=='4q' else width)
        qb=quant_boxes(nodes,kids) if width=='4q' else None
        tot=dict(steps=0,boxes=0,leaves=0,tris=0,pushes=0,pops=0,leaf_fail=0); nb=0; bt=dict(tot)
        bounce=[]
        for o,d in rays:
            st,t=trace(nodes,kids,tri_test,o,d,qb)
            for k in tot: tot[k]+=st[k]
            if t<math.inf and width==2:
                p=np.array(o)+t*np.array(d)
                # a diffuse-ish bounce: random direction, offset along it
                while True:
                    s=np.array([rng.gauss(0,1) for _ in range(3)]); s/=np.linalg.norm(s)
                    break
                bounce.append((tuple(p+1e-3*s), tuple(s)))
        if width==2: brays=bounce
        for o,d in brays:
            st,t=trace(nodes,kids,tri_test,o,d,qb)
            for k in bt: bt[k]+=st[k]
        print(which,"width",width,"camera/ray:",{k:round(v/len(rays),2) for k,v in tot.items()}," bounce/ray:",{k:round(v/max(1,len(brays)),2) for k,v in bt.items()}, "n_wide", len(kids))
N=int(sys.argv[2]) if len(sys.argv)>2 else 1500
main(sys.argv[1])
