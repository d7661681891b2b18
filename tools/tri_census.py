"""Float64 census of the benchmark frame's triangles (CPU, no GPU): for a sample of config 3's instances, every front-facing triangle whose snapped
box holds a pixel centre - box size, pixels covered (exact integer edge functions, top-left rule).  What the camera pass's records are: DESIGN.md section 5.
    python tools/tri_census.py"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from zeldaengine_amd import scenes, abi
cfg = scenes.config3()
W,H = cfg["width"], cfg["height"]
cam = cfg["camera"]
def look_at(eye, center, up):
    f = center - eye; f /= np.linalg.norm(f); s_ = np.cross(f, up); s_ /= np.linalg.norm(s_); u = np.cross(s_, f)
    m = np.eye(4); m[0,:3] = s_; m[1,:3] = u; m[2,:3] = -f; m[0,3] = -s_ @ eye; m[1,3] = -u @ eye; m[2,3] = f @ eye
    return m
def perspective(fovy, aspect, zn, zf):            # glm::perspectiveRH_ZO, then Proj[1][1] *= -1 (ZE:4617-4624)
    t = np.tan(fovy / 2.0); m = np.zeros((4,4))
    m[0,0] = 1.0 / (aspect * t); m[1,1] = -1.0 / t; m[2,2] = zf / (zn - zf); m[3,2] = -1.0; m[2,3] = -(zf * zn) / (zf - zn)
    return m
eye = np.array(list(cam.Position), dtype=np.float64); ctr = np.array(list(cam.Lookat), dtype=np.float64)
PVM = perspective(np.radians(float(cam.FOV)), W / H, float(cam.zNear), float(cam.zFar)) @ look_at(eye, ctr, np.array([0.0, 0.0, 1.0]))
obj = cfg["objects"][0]
verts, idx = obj["mesh"]; inst = obj["instances"]
pos = verts["Position"].astype(np.float64)
tri = idx.reshape(-1,3)
# instance transform of BaseInstanced.vert:38-75: p = (pos * s) * mat3(R) + t, R = mz * my * mx
def rot(e):
    x,y,z = e
    cx,sx=np.cos(x),np.sin(x); cy,sy=np.cos(y),np.sin(y); cz,sz=np.cos(z),np.sin(z)
    mx=np.array([[cx,0,sx],[0,1,0],[-sx,0,cx]]); my=np.array([[cy,-sy,0],[sy,cy,0],[0,0,1]]); mz=np.array([[1,0,0],[0,cz,-sz],[0,sz,cz]])
    return mz@my@mx
hist = {}
tot=cov_tot=0; zero=0; recs=0
boxes=[]; covs=[]
rng = np.random.default_rng(0)
sel = rng.choice(len(inst), 1500, replace=False)
for ii in sel:
    I = inst[ii]
    R = rot(I["InstanceRotation"]); 
    p = (pos*I["InstancePScale"])@R + np.array(I["InstancePosition"])
    c = np.c_[p, np.ones(len(p))]@PVM.T
    w = c[:,3]
    if (w<=0.1).any(): continue
    sx = np.round(((c[:,0]/w)*0.5+0.5)*W*256).astype(np.int64); sy = np.round(((c[:,1]/w)*0.5+0.5)*H*256).astype(np.int64)
    X = sx[tri]; Y = sy[tri]
    A = (X[:,1]-X[:,0])*(Y[:,2]-Y[:,0]) - (X[:,2]-X[:,0])*(Y[:,1]-Y[:,0])
    x0 = np.maximum((X.min(1)-128+255)>>8,0); x1=np.minimum((X.max(1)-128)>>8,W-1)
    y0 = np.maximum((Y.min(1)-128+255)>>8,0); y1=np.minimum((Y.max(1)-128)>>8,H-1)
    ok = (A<0)&(x0<=x1)&(y0<=y1)
    for t in np.nonzero(ok)[0]:
        bw, bh = x1[t]-x0[t]+1, y1[t]-y0[t]+1
        # coverage count exact
        xs = np.arange(x0[t],x1[t]+1)*256+128; ys=np.arange(y0[t],y1[t]+1)*256+128
        PX,PY = np.meshgrid(xs,ys)
        n=0
        inside = np.ones(PX.shape,bool)
        for (a,b) in ((1,2),(2,0),(0,1)):
            ex = -(X[t,b]-X[t,a]); ey = -(Y[t,b]-Y[t,a])   # sgn=-1
            e = ex*(PY-Y[t,a]) - ey*(PX-X[t,a])
            tl = 0 if ((ey<0) or (ey==0 and ex>0)) else 1
            inside &= (e - tl) >= 0
        n = int(inside.sum())
        boxes.append((bw,bh)); covs.append(n)
boxes=np.array(boxes); covs=np.array(covs)
area = boxes[:,0]*boxes[:,1]
print("triangles with a pixel centre in box:", len(covs), "mean box area", area.mean(), "mean covered", covs.mean(), "zero-coverage frac", (covs==0).mean())
for lim in (1,2,4,6,9,16):
    m = area<=lim
    print("area<=%d: frac %.3f, zero-cov within %.3f, share of all pixel tests %.3f" % (lim, m.mean(), (covs[m]==0).mean(), area[m].sum()/area.sum()))
m11 = (boxes[:,0]==1)&(boxes[:,1]==1)
print("1x1 frac", m11.mean(), "covered frac among 1x1", (covs[m11]>0).mean())
m22 = (boxes[:,0]<=2)&(boxes[:,1]<=2)
print("<=2x2 frac", m22.mean(), "zero-cov among", (covs[m22]==0).mean())
print("percentiles area", np.percentile(area,[50,75,90,95,99]))
