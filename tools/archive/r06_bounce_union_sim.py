"""Round 6, before the leaf boxes were built (profiles/EXPERIMENTS.md 6.1): what could a bounce round of the packet kernel skip?  A numpy brute-force path tracer of the
headline frame at 480 x 270 (default scene, default camera, Lambert bounces with numpy's RNG: statistics only), packets of 64 spatially sorted bounce rays, against the
library's own bounce table (native.bounce_rows, laboratory build; no GPU).  Prints the mean candidates per ray and the union over a packet for: the table's rows, rows
refined by the ray's octant, rows keyed on cube-map direction bins, per-ray slab tests against cluster boxes, a wave-uniform hierarchical walk, and leaf boxes over
index nibbles / bytes pruned by the row union (what was built).  Takes a few minutes.   python tools/archive/r06_bounce_union_sim.py"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from rvpt_amd import native, scene
from rvpt_amd.camera import Camera
from test_camera_rects import prepared_records
tris,mats=scene.default_scene()
prep=prepared_records(tris)
rows,scale=native.bounce_rows(tris,prep)
n=len(tris)
bits=((rows[:,:,None]>>np.arange(32,dtype=np.uint32))&1).reshape(rows.shape[0],-1).astype(bool)[:,:n]
p=prep.astype(np.float64); v0,nn,e0,e1=p[:,0:3],p[:,3:6],p[:,6:9],p[:,9:12]
verts=np.stack([v0,v0+e0,v0+e1],1)
lo,hi=verts.min(1),verts.max(1)
m=2.0**-10*scale
# octant-refined rows: oct bit k set = direction component k >= 0
oct_rows=np.zeros((2*n,8,n),bool)
for o in range(8):
    ok=np.ones((n,n),bool)  # [A,B]
    for k in range(3):
        if (o>>k)&1:  # d_k>=0: B must reach x_k >= lo_A_k - m
            ok&= hi[None,:,k] >= lo[:,None,k]-m
        else:
            ok&= lo[None,:,k] <= hi[:,None,k]+m
    oct_rows[0::2,o]=bits[0::2]&ok; oct_rows[1::2,o]=bits[1::2]&ok
print("row density",bits.mean(),"octant rows",oct_rows.mean())
# simulate paths
W,H=480,270
rng=np.random.default_rng(1)
cam=Camera(W/H)
ys,xs=np.mgrid[0:H,0:W]
# block order: tiles 16x16 row-major, within tile 16x4 blocks -> approximate wave order: sort by (ty,tx,by)
px=(xs+rng.random(xs.shape)).ravel(); py=(ys+rng.random(ys.shape)).ravel()
u=(W/H)*(2*px/W-1); v=2*(1-py/H)-1
d=np.stack([u,v,np.ones_like(u)],1); d/=np.linalg.norm(d,axis=1,keepdims=True)
o=np.zeros_like(d)
order_key=((ys//16)*1000+(xs//16)).ravel()*100+((ys%16)//4).ravel()
def intersect(o,d):
    best=np.full(len(o),np.inf); hit=np.full(len(o),-1)
    for j in range(n):
        den=d@nn[j]; num=(v0[j]-o)@nn[j]
        with np.errstate(all='ignore'):
            t=num/den
        P=o+t[:,None]*d-v0[j]
        a00,a11,a01=e1[j]@e1[j],e0[j]@e0[j],e0[j]@e1[j]
        det=a00*a11-a01*a01
        b0=P@e0[j]; b1=P@e1[j]
        uu=(a00*b0-a01*b1)/det; vv=(a11*b1-a01*b0)/det
        acc=(t>0)&(t<best)&(uu>0)&(vv>0)&(uu+vv<1)
        best=np.where(acc,t,best); hit=np.where(acc,j,hit)
    return best,hit
t,hit=intersect(o,d)
alive=hit>=0
print("camera hit frac",alive.mean())
key=order_key[alive]; o=o[alive]; d=d[alive]; t=t[alive]; hit=hit[alive]
all_rays=[]
for bounce in range(1,8):
    pos=o+t[:,None]*d
    N=nn[hit]/np.linalg.norm(nn[hit],axis=1,keepdims=True)
    side=(np.einsum('ij,ij->i',d,N)>0)  # other side
    N=np.where(side[:,None],-N,N)
    S=rng.normal(size=pos.shape); S/=np.linalg.norm(S,axis=1,keepdims=True)
    d=N+S; o=pos+0.005*N
    leave=2*hit+side.astype(int)
    all_rays.append((key.copy(),o.copy(),d.copy(),leave.copy(),np.full(len(o),bounce)))
    t,hit2=intersect(o,d)
    al=hit2>=0
    key,o,d,t,hit=key[al],o[al],d[al],t[al],hit2[al]
    print("bounce",bounce,"rays",len(al),"hit frac",al.mean())
K=np.concatenate([r[0] for r in all_rays]); O=np.concatenate([r[1] for r in all_rays]); D=np.concatenate([r[2] for r in all_rays]); L=np.concatenate([r[3] for r in all_rays]); B=np.concatenate([r[4] for r in all_rays])
octs=((D[:,0]>=0).astype(int))|((D[:,1]>=0).astype(int)<<1)|((D[:,2]>=0).astype(int)<<2)
# emulate wave-level packets: sort by key (spatial), bounce within later; rays of different bounce depth mix within a wave. approximate: order by (key//pool, random)
def measure(pool_blocks, mode):
    pool=K//(100*1)//pool_blocks  # group by tile-ish
    idx=np.lexsort((rng.random(len(K)),pool))
    res=[];res_own=[]
    # within each pool form packets
    start=0
    pools=np.split(idx,np.flatnonzero(np.diff(pool[idx]))+1)
    for pl in pools:
        if mode=='octant': pl=pl[np.argsort(octs[pl],kind='stable')]
        for s in range(0,len(pl)-63,64):
            pk=pl[s:s+64]
            if mode=='plain':
                un=bits[L[pk]].any(0).sum()
            else:
                un=oct_rows[L[pk],octs[pk]].any(0).sum()
            res.append(un)
    return np.mean(res),len(res)
for pb in (2,8,32,128):
    print("pool tiles",pb,"plain union",measure(pb,'plain'),"octant rows, unsorted",measure(pb,'oct_unsorted'),"octant rows sorted",measure(pb,'octant'))
print("own row mean",bits[L].sum(1).mean(),"own octant row mean",oct_rows[L,octs].sum(1).mean())

# ---- direction-binned rows: cube-map faces x k x k
def dir_rows(k):
    edges=np.linspace(-1,1,k+1)
    nb=6*k*k
    rows=np.zeros((n,nb,n),bool)
    pad=0.005+m
    loA,hiA=lo-pad,hi+pad; loB,hiB=lo-m,hi+m
    D0=loB[None,:,:]-hiA[:,None,:]; D1=hiB[None,:,:]-loA[:,None,:]  # [A,B,3]
    b=0
    for f in range(3):
        uax,vax=[(1,2),(0,2),(0,1)][f]
        for s in (1,-1):
            g0=np.maximum(np.where(s>0,D0[...,f],-D1[...,f]),0.0); g1=np.where(s>0,D1[...,f],-D0[...,f])
            for iu in range(k):
                for iv in range(k):
                    glo=g0.copy(); ghi=g1.copy()
                    for ax,(r0,r1) in ((uax,(edges[iu],edges[iu+1])),(vax,(edges[iv],edges[iv+1]))):
                        d0,d1=D0[...,ax],D1[...,ax]
                        # need r0*g <= d1 and r1*g >= d0
                        # r0*g<=d1: if r0>0: g<=d1/r0 ; r0<0: g>=d1/r0 ; r0==0: d1>=0
                        with np.errstate(all='ignore'):
                            if r0>0: ghi=np.minimum(ghi,d1/r0)
                            elif r0<0: glo=np.maximum(glo,d1/r0)
                            else: ghi=np.where(d1>=0,ghi,-1)
                            if r1>0: glo=np.maximum(glo,d0/r1)
                            elif r1<0: ghi=np.minimum(ghi,d0/r1)
                            else: ghi=np.where(d0<=0,ghi,-1)
                    rows[:,b,:]=(ghi>=glo)&(g1>0)
                    b+=1
    return rows,edges
def bin_of(D,k,edges):
    a=np.abs(D); f=a.argmax(1)
    s=np.take_along_axis(D,f[:,None],1)[:,0]>=0
    uax=np.array([1,0,0])[f]; vax=np.array([2,2,1])[f]
    g=np.take_along_axis(a,f[:,None],1)[:,0]
    ru=np.take_along_axis(D,uax[:,None],1)[:,0]/g; rv=np.take_along_axis(D,vax[:,None],1)[:,0]/g
    iu=np.clip(np.searchsorted(edges,ru,side='right')-1,0,k-1); iv=np.clip(np.searchsorted(edges,rv,side='right')-1,0,k-1)
    return (f*2+(~s).astype(int))*k*k+iu*k+iv
A=L>>1
for k in (1,2,4,8):
    R,edges=dir_rows(k)
    bn=bin_of(D,k,edges)
    cand=(R[A,bn]&bits[L])
    print("k",k,"bins",6*k*k,"table density",R.mean(),"cand/ray",cand.sum(1).mean(), "table KB", 2*n*6*k*k*((n+31)//32)*4/1024)
    # union over unsorted 64-packets
    idx=np.argsort(K,kind='stable'); un=[cand[idx[s:s+64]].any(0).sum() for s in range(0,len(idx)-63,64)]
    print("   union/packet",np.mean(un))

# ---- per-ray cluster filter: triangles sorted spatially, clusters of CS consecutive triangles
cent=verts.mean(1)
def morton(c):
    q=((c-c.min(0))/(c.max(0)-c.min(0)+1e-9)*1023).astype(np.int64)
    def spread(x):
        x=(x|(x<<16))&0x030000FF; x=(x|(x<<8))&0x0300F00F; x=(x|(x<<4))&0x030C30C3; x=(x|(x<<2))&0x09249249; return x
    return spread(q[:,0])|(spread(q[:,1])<<1)|(spread(q[:,2])<<2)
orderT=np.argsort(morton(cent))
for CS in (4,8,16):
    ncl=(n+CS-1)//CS
    cl_of=np.zeros(n,int); cl_of[orderT]=np.arange(n)//CS
    clo=np.array([lo[cl_of==c].min(0) for c in range(ncl)])-m; chi=np.array([hi[cl_of==c].max(0) for c in range(ncl)])+m
    with np.errstate(all='ignore'):
        inv=1.0/D
    hitc=np.zeros((len(D),ncl),bool)
    for c in range(ncl):
        t0=(clo[c]-O)*inv; t1=(chi[c]-O)*inv
        tn=np.minimum(t0,t1).max(1); tf=np.maximum(t0,t1).min(1)
        hitc[:,c]=tf>=np.maximum(tn,0)
    cand=bits[L]&hitc[:,cl_of]
    cnt=cand.sum(1)
    idx=np.argsort(K,kind='stable')
    un=[cand[idx[s:s+64]].any(0).sum() for s in range(0,len(idx)-63,64)]
    nz=[(cnt[idx[s:s+64]]>0).sum() for s in range(0,len(idx)-63,64)]
    mx=[cnt[idx[s:s+64]].max() for s in range(0,len(idx)-63,64)]
    print("CS",CS,"clusters",ncl,"cand/ray",cnt.mean(),"zero frac",(cnt==0).mean(),"union/packet",np.mean(un),"nonzero lanes/packet",np.mean(nz),"max cand in packet",np.mean(mx),"pairs/packet",cnt.mean()*64)

print("---- wave-uniform hierarchical cluster walk")
def build_tree(LS,F):
    # leaves: consecutive LS triangles in Morton order; parents: consecutive F children
    leaves=[orderT[i:i+LS] for i in range(0,n,LS)]
    levels=[[ (lo[t].min(0)-m, hi[t].max(0)+m, t) for t in leaves ]]
    while len(levels[-1])>F:
        prev=levels[-1]; cur=[]
        for i in range(0,len(prev),F):
            ch=prev[i:i+F]
            cur.append((np.min([c[0] for c in ch],0),np.max([c[1] for c in ch],0),list(range(i,min(i+F,len(prev))))))
        levels.append(cur)
    return levels[::-1]  # top first
def boxhit(O,inv,blo,bhi):
    t0=(blo-O)*inv; t1=(bhi-O)*inv
    tn=np.minimum(t0,t1).max(1); tf=np.maximum(t0,t1).min(1)
    return tf>=np.maximum(tn,0)
idx=np.argsort(K,kind='stable')
with np.errstate(all='ignore'):
    INV=1.0/D
for LS,F in ((4,4),(4,3),(8,4),(4,6),(2,4),(4,8)):
    levels=build_tree(LS,F)
    tot_nodes=[];tot_tris=[]
    for s in range(0,len(idx)-63,64*7):  # subsample packets
        pk=idx[s:s+64]
        O_,I_=O[pk],INV[pk]
        rowU=bits[L[pk]].any(0)
        active=list(range(len(levels[0]))); ntest=0
        for li,lev in enumerate(levels):
            nxt=[]
            for ni in active:
                blo,bhi,ch=lev[ni]; ntest+=1
                if boxhit(O_,I_,blo,bhi).any():
                    if li==len(levels)-1: nxt.extend(ch.tolist())
                    else: nxt.extend(ch)
            active=nxt
        tri=[t for t in active if rowU[t]]
        tot_nodes.append(ntest); tot_tris.append(len(tri))
    nn_,tt_=np.mean(tot_nodes),np.mean(tot_tris)
    print(f"LS {LS} F {F} levels {[len(l) for l in levels]}: node tests/packet {nn_:.1f}, triangle tests {tt_:.1f}, est VALU {16*nn_+36*tt_:.0f} (now {36*rowU.sum():.0f}~1070)")

print("---- leaf boxes over index nibbles (caller order = BVH leaf order), pruned by the row union")
nodes_,idxs=native.build_bvh(tris)
# NOTE: the script's `tris` must be in BVH order for this to be meaningful: redo everything with sorted triangles is heavy; instead map indices: sorted position of triangle t
pos_of=np.empty(n,int); pos_of[idxs]=np.arange(n)   # triangle t sits at sorted position pos_of[t]
for LS in (4,8):
    nl=(n+LS-1)//LS
    leaf_of=pos_of//LS
    llo=np.array([lo[leaf_of==l].min(0) for l in range(nl)])-m; lhi=np.array([hi[leaf_of==l].max(0) for l in range(nl)])+m
    wlo=np.array([lo[(pos_of//32)==w].min(0) for w in range((n+31)//32)])-m; whi=np.array([hi[(pos_of//32)==w].max(0) for w in range((n+31)//32)])+m
    res=[]
    for s in range(0,len(idx)-63,64*7):
        pk=idx[s:s+64]; O_,I_=O[pk],INV[pk]
        rowU=bits[L[pk]].any(0)
        leaves=np.unique(leaf_of[rowU])
        ntest=0; ntri=0; nword=0; ntest2=0
        words_hit=set()
        for w in np.unique(pos_of[rowU]//32):
            nword+=1
            if boxhit(O_,I_,wlo[w],whi[w]).any(): words_hit.add(w)
        for l in leaves:
            if (l*LS)//32 in words_hit: ntest2+=1
            ntest+=1
            if boxhit(O_,I_,llo[l],lhi[l]).any():
                ntri+=(rowU&(leaf_of==l)).sum()
        res.append((ntest,ntri,nword,ntest2))
    r=np.mean(res,0)
    print(f"LS {LS}: leaf tests {r[0]:.1f}, triangle tests {r[1]:.1f} -> VALU {16*r[0]+36*r[1]:.0f}; with word boxes first: word tests {r[2]:.1f} + leaf tests {r[3]:.1f} -> VALU {16*(r[2]+r[3])+36*r[1]:.0f}")
