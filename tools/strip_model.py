#!/usr/bin/env python3
"""CPU model (numpy, on the oracle's lists): how many wave iterations would the blend kernels need if the four 16-lane
groups of a wave walked DIFFERENT Gaussians -- one survivor list per 8x2 strip, 8x4 half or 4x4 block of the quadrant,
culled with the same exact ellipse-vs-rectangle test, advancing in lockstep (iterations = longest of the group lists)?
Quarter-size config 3 (same pairs per tile).  Result (recorded in profiles/experiments/README.md): 0.81x / 0.87x / 0.76x
of today's iteration count -- the Gaussians of the benchmark scene are rarely smaller than a strip, so the finer cull
buys under 20 % before any of its costs (four mask walks per iteration, per-lane record addresses, per-group batches)."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import numpy as np, torch
import lvdgs
from lvdgs import synthetic
import oracle as orc
N,W,H=150000,960,540   # quarter-size version of config 3 with the same density per tile
g=synthetic.make_gaussians(N,W,H,seed=0)
cam=synthetic.make_camera(W,H)
o=orc.Oracle("f32")
f=o.forward(means3D=g["means3D"].numpy(),opacities=g["opacities"].numpy(),W=W,H=H,tanfovx=cam.tanfovx,tanfovy=cam.tanfovy,
 viewmatrix=cam.world_view_transform.numpy(),projmatrix=cam.full_proj_transform.numpy(),projmatrix_raw=cam.projection_matrix.numpy(),
 campos=cam.camera_center.numpy(),bg=np.zeros(3),scales=g["scales"].numpy(),rotations=g["rotations"].numpy(),colors_precomp=g["colors"].numpy())
ids=f["ids_sorted"].astype(np.int64); tiles=(f["keys_sorted"]>>np.uint64(32)).astype(np.int64)
gx=(W+15)//16
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
D=len(ids); print("pairs",D, "pairs/tile", D/((W+15)//16*((H+15)//16)))
tot_q=0; it_strip8x2=0; it_half=0; tot_s=0
for q in range(4):
    qx0=tx*16+(q&1)*8; qy0=ty*16+(q>>1)*8
    sq=reaches(qx0,qy0,qx0+7,qy0+7)
    tot_q+=sq.sum()
    # strips 8x2
    cs=[]
    for s in range(4):
        ss=reaches(qx0,qy0+2*s,qx0+7,qy0+2*s+1)&sq
        cs.append(np.bincount(tiles,weights=ss,minlength=gx*((H+15)//16)))
        tot_s+=ss.sum()
    it_strip8x2+=np.max(np.stack(cs),0).sum()
    ch=[]
    for h in range(2):
        sh=reaches(qx0,qy0+4*h,qx0+7,qy0+4*h+3)&sq
        ch.append(np.bincount(tiles,weights=sh,minlength=gx*((H+15)//16)))
    it_half+=np.max(np.stack(ch),0).sum()
print("quadrant survivors / pair*4:", tot_q/(4*D))
print("wave iterations now:", tot_q)
print("strip (8x2) units:", tot_s, " lockstep iterations (max over the 4 strips per quadrant):", it_strip8x2, " ratio to now:", it_strip8x2/tot_q)
print("half (8x4) lockstep iterations:", it_half, " ratio:", it_half/tot_q)
it_blk=0
for q in range(4):
    qx0=tx*16+(q&1)*8; qy0=ty*16+(q>>1)*8
    sq=reaches(qx0,qy0,qx0+7,qy0+7)
    cb=[]
    for bl in range(4):
        bx,by=qx0+(bl&1)*4,qy0+(bl>>1)*4
        sb=reaches(bx,by,bx+3,by+3)&sq
        cb.append(np.bincount(tiles,weights=sb,minlength=gx*((H+15)//16)))
    it_blk+=np.max(np.stack(cb),0).sum()
print("block (4x4) lockstep iterations:", it_blk, " ratio:", it_blk/tot_q)
