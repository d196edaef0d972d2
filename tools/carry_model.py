#!/usr/bin/env python3
"""CPU model (numpy, on the oracle's lists): fill of blend_bwd's splat batches (8 slots) at a quarter-size config 3
when a wave may keep up to CARRY leftover survivors of a 64-entry round open for the next round, against closing a
partial batch at every round.  Needs no GPU."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import numpy as np, torch
import lvdgs
from lvdgs import synthetic
import oracle as orc
N,W,H=150000,960,540
if len(sys.argv)>1 and sys.argv[1]=="cfg3": N,W,H=500000,1920,1080
if len(sys.argv)>1 and sys.argv[1]=="kitti": N,W,H=200000,1226,370
g=synthetic.make_gaussians(N,W,H,seed=0)
cam=synthetic.make_camera(W,H)
o=orc.Oracle("f32")
f=o.forward(means3D=g["means3D"].numpy(),opacities=g["opacities"].numpy(),W=W,H=H,tanfovx=cam.tanfovx,tanfovy=cam.tanfovy,
 viewmatrix=cam.world_view_transform.numpy(),projmatrix=cam.full_proj_transform.numpy(),projmatrix_raw=cam.projection_matrix.numpy(),
 campos=cam.camera_center.numpy(),bg=np.zeros(3),scales=g["scales"].numpy(),rotations=g["rotations"].numpy(),colors_precomp=g["colors"].numpy())
ids=f["ids_sorted"].astype(np.int64); tiles=(f["keys_sorted"]>>np.uint64(32)).astype(np.int64)
gx=(W+15)//16; gy=(H+15)//16
m=f["means2D"][ids].astype(np.float64); co=f["conic_opacity"][ids].astype(np.float64)
a,b,c,op=co[:,0],co[:,1],co[:,2],co[:,3]
tx,ty=tiles%gx,tiles//gx
def reaches(x0,y0,x1,y1):
    dx_lo,dx_hi,dy_lo,dy_hi=m[:,0]-x1,m[:,0]-x0,m[:,1]-y1,m[:,1]-y0
    inside=(dx_lo<=0)&(dx_hi>=0)&(dy_lo<=0)&(dy_hi>=0)
    def along_y(dx):
        dy=np.clip(-b*dx/c,dy_lo,dy_hi); return 0.5*(a*dx*dx+c*dy*dy)+b*dx*dy
    def along_x(dy):
        dx=np.clip(-b*dy/a,dx_lo,dx_hi); return 0.5*(a*dx*dx+c*dy*dy)+b*dx*dy
    qmin=np.minimum(np.minimum(along_y(dx_lo),along_y(dx_hi)),np.minimum(along_x(dy_lo),along_x(dy_hi)))
    return (op>=1/255)&(inside|(qmin<=np.log(op*255)+0.02))
keep_tile=reaches(tx*16,ty*16,tx*16+15,ty*16+15)   # the listed pairs (tile culling)
print("rect pairs",len(ids),"listed",keep_tile.sum(),"per tile",keep_tile.sum()/(gx*gy))
surv=[reaches(tx*16+(q&1)*8,ty*16+(q>>1)*8,tx*16+(q&1)*8+7,ty*16+(q>>1)*8+7)&keep_tile for q in range(4)]
# position of every listed pair inside its tile's list
order=np.flatnonzero(keep_tile)
t=tiles[order]
start=np.r_[0,np.flatnonzero(np.diff(t))+1]
pos=np.arange(len(t))-np.repeat(start,np.diff(np.r_[start,len(t)]))
tl=np.repeat(np.diff(np.r_[start,len(t)]),np.diff(np.r_[start,len(t)]))
# rounds go back to front: round index from the front = pos // 64 (the kernel's base = r * 64)
rnd=pos//64
ntile=gx*gy; maxr=rnd.max()+1
tot=0
S=np.zeros((4,ntile,maxr),np.int64)
for q in range(4):
    s=surv[q][order]
    np.add.at(S[q],(t[s],rnd[s]),1)
    tot+=s.sum()
print("survivors",tot,"of",4*len(t),"=",tot/(4*len(t)))
def sim(carry_max, varwidth=False):
    slots=0; batches=0; steps=0; carried=0
    for q in range(4):
        for ti in range(ntile):
            k=0
            nr=int(np.ceil((S[q,ti]>0).nonzero()[0].max()+1)) if S[q,ti].any() else 0
            for r in range(nr-1,-1,-1):
                n=k+int(S[q,ti,r])
                full=n//8; L=n-8*full
                batches+=full; slots+=8*full; steps+=8*full
                if k>0 and n<8:       # carried entries must finish in this round
                    if n>0: batches+=1; slots+=8; steps+=(8 if not varwidth else (1 if n==1 else 2 if n==2 else 4 if n<=4 else 8))
                    k=0; continue
                if r>0 and L<=carry_max:
                    k=L; carried+=L
                else:
                    if L>0: batches+=1; slots+=8; steps+=(8 if not varwidth else (1 if L==1 else 2 if L==2 else 4 if L<=4 else 8))
                    k=0
    return slots,batches,steps,carried
for cm,vw in ((0,False),(0,True),(3,False),(4,False),(7,False),(4,True)):
    slots,batches,steps,carried=sim(cm,vw)
    print(f"carry<={cm} varwidth={vw}: fill {tot/slots:.3f} batches {batches} splat steps/survivor {steps/tot:.3f} carried entries {carried} ({carried/tot:.3f})")
hist=np.bincount(np.minimum(S[S>0],64))
print("survivors per (wave, round) histogram:",hist.tolist(), "mean", S[S>0].mean())
