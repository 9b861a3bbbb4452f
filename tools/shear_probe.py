import sys, math, torch, numpy as np
sys.path.insert(0,'/root/repo')
from diffdrr_amd import DRR
from diffdrr_amd.data import make_subject
from diffdrr_amd.pose import convert
D,H=512,256
drr = DRR(make_subject(torch.zeros(8,8,8)), sdd=1020.0, height=H, delx=2.4)
# fake affine for 512^3 centered
from diffdrr_amd.data import centered_affine
drr2 = DRR(make_subject(torch.zeros(D,D,2)), sdd=1020.0, height=H, delx=2.4)
def poses(B, seed):
    g = torch.Generator().manual_seed(seed)
    rot = (torch.rand(B, 3, generator=g) - 0.5) * (math.pi / 2)
    xyz = torch.tensor([0.0, 850.0, 0.0]) + (torch.rand(B, 3, generator=g) - 0.5) * 60.0
    return rot, xyz
rot,xyz = poses(32,2)
with torch.no_grad():
    pose = convert(rot, xyz, parameterization="euler_angles", convention="ZXY")
    s_w, t_w = drr.detector(pose, None)
# voxel coords: affine = diag(1) with translation -(D-1)/2
s = s_w + (D-1)/2; t = t_w + (D-1)/2
B=32
tg = t.reshape(B,H,H,3).double(); src = s[:,0].double()
t00, ei, ej = tg[:,0,0], (tg[:,-1,0]-tg[:,0,0])/(H-1), (tg[:,0,-1]-tg[:,0,0])/(H-1)
BX,BY,BZ=32,32,64
n=[D//BX, D//BY, D//BZ]
ix,iy,iz = torch.meshgrid(*[torch.arange(k) for k in n], indexing="ij")
lo = torch.stack([ix*BX, iy*BY, iz*BZ],-1).reshape(-1,3).double(); hi = lo+torch.tensor([BX,BY,BZ]).double()
corners = torch.stack([torch.where(torch.tensor([(c>>k)&1 for k in range(3)]).bool(), hi, lo) for c in range(8)],1)-0.5
w = corners[None]-src[:,None,None]
nvec = torch.linalg.cross(ei,ej); r = t00-src
det = (w*nvec[:,None,None]).sum(-1)
i = -(w*torch.linalg.cross(r,ej)[:,None,None]).sum(-1)/det
j = (w*torch.linalg.cross(r,ei)[:,None,None]).sum(-1)/det    # (B, nb, 8)
def clampbox(i0,i1,j0,j1):
    return i0.clamp(0,H-1), i1.clamp(0,H-1), j0.clamp(0,H-1), j1.clamp(0,H-1)
i0,i1 = i.amin(-1).floor(), i.amax(-1).ceil()
j0,j1 = j.amin(-1).floor(), j.amax(-1).ceil()
vis = (i1>=0)&(i0<=H-1)&(j1>=0)&(j0<=H-1)
ci0,ci1,cj0,cj1 = clampbox(i0,i1,j0,j1)
area_bbox = ((ci1-ci0+1)*(cj1-cj0+1))*vis
# hexagon area via convex hull area (shoelace on hull) - approximate by scipy
from scipy.spatial import ConvexHull
# sheared: slopes from axis directions
def width_for(k):
    jp = j - k[...,None]*i
    return (jp.amax(-1)-jp.amin(-1))
rows = (ci1-ci0+1)
best = (j1-j0+1)
ks=[]
for a in range(3):
    c1 = 1<<a
    di = i[...,c1]-i[...,0]; dj = j[...,c1]-j[...,0]
    k = torch.where(di.abs()>0.25*dj.abs(), dj/di, torch.zeros_like(di))
    wk = width_for(k).ceil()+2
    best = torch.minimum(best, wk)
area_shear = rows*best*vis
# rough: rows clamped but width not clamped to detector; fine
hull=0.0
idx = torch.nonzero(vis.reshape(-1)).reshape(-1)[::37]
tot_h=0; tot_b=0; tot_s=0
ii=i.reshape(-1,8); jj=j.reshape(-1,8)
ab=area_bbox.reshape(-1); as_=area_shear.reshape(-1)
for q in idx.tolist():
    pts=np.stack([ii[q].numpy(), jj[q].numpy()],1)
    if pts.min()<0 or pts.max()>H-1: continue
    tot_h+=ConvexHull(pts).volume; tot_b+=float(ab[q]); tot_s+=float(as_[q])
print("sampled interior bricks: hull area", tot_h, "bbox", tot_b, "sheared", tot_s, "ratios bbox/hull %.2f shear/hull %.2f"%(tot_b/tot_h, tot_s/tot_h))
print("total bbox candidates per launch %.3g ; sheared %.3g ; ratio %.3f" % (area_bbox.sum(), area_shear.sum(), area_shear.sum()/area_bbox.sum()))
# aligned-8 variants
area_bbox8 = ((ci1-ci0+1)*((cj1/8).floor()*8+7 - (cj0/8).floor()*8 +1))*vis
area_shear8 = rows*((best+7+7)/8).floor()*8*vis
print("aligned: bbox8 %.3g sheared8 %.3g ratio %.3f" % (area_bbox8.sum(), area_shear8.sum(), area_shear8.sum()/area_bbox8.sum()))
