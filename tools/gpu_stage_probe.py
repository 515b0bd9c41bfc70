import os, sys, time, json
ROOT='/root/repo'
for p in (ROOT, ROOT+'/gaussian-splatting_amd'): sys.path.insert(0,p)
import torch
from gsr_synth import make_camera, make_scene
from diff_gaussian_rasterization import GaussianRasterizationSettings, _lib, rasterize_gaussians
dev=torch.device('cuda:0'); W,H,P=1920,1080,1_000_000
cam=make_camera(W,H); sc=make_scene(P,cam,seed=0,s_med=0.012).to(dev); camd=cam.to(dev)
rs=GaussianRasterizationSettings(H,W,cam.tanfovx,cam.tanfovy,torch.zeros(3,device=dev),1.0,camd.world_view_transform,camd.full_proj_transform,3,camd.camera_center,False,False,False)
def step():
    with torch.no_grad(): rasterize_gaussians(sc.means3D,None,sc.shs,None,sc.opacities,sc.scales,sc.rotations,None,rs,None)
for _ in range(50): step()
torch.cuda.synchronize()
res=[]
for rep in range(3):
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(200): step()
    torch.cuda.synchronize(); st=_lib.profile_read(); _lib.profile_enable(False)
    res.append({k: round(v['ms']/v['launches'],4) for k,v in st.items() if v['launches']})
print(json.dumps({'lib':os.environ.get('GSR_LIB','product')[-40:], 'stage_ms':res}))
