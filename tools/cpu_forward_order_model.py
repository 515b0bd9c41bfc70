"""Could the FORWARD blend start its heaviest blocks first?  (Its launch loses ~20 % to the tail; the backward's order is planned from
the forward's step counts, the forward itself has no such record.)  For the bench frame and the clustered scene: correlation of a tile's measured
blend steps (gpurun_out/block_steps_*.npy, written by tools/gpu_tail_model.py on the GPU box) with what is known before the blend -- the tile's list
length, and the number of list entries until an opacity x coverage prefix sum reaches 2 / 4 / 6 / 9.2 -- and what the best of them would buy in the
processor-sharing model.  Result (round 4): correlations 0.03-0.05 (uniform), 0.1-0.3 (clustered); launch-order model 1.134 -> 1.138 / 1.131
against 1.014 for the exact order: nothing cheaper than the blend predicts where the opacity saturates.

    python tools/cpu_forward_order_model.py uniform|clustered          (CPU; imports oracle/ through tests/helpers.py: analysis only)"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sched_model import F, ps_makespan
from helpers import O, make_camera, make_scene, make_clustered_scene, oracle_settings
name=sys.argv[1]
W,H=1920,1080
cam=make_camera(W,H)
sc = make_clustered_scene(1_000_000, cam, seed=0) if name=='clustered' else make_scene(1_000_000, cam, seed=0, s_med=0.012)
s=oracle_settings(cam)
with torch.no_grad():
    pre=O.preprocess(sc.means3D, sc.opacities, s, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    bins=O.bin_and_sort(pre)
bs=np.load(os.path.join(ROOT, 'gpurun_out', f'block_steps_{name}.npy')).astype(np.float64)
ranges=bins['ranges'].numpy(); pl=bins['point_list'].numpy()
op=pre['opacity'].numpy().reshape(-1); con=pre['conic'].numpy(); 
# footprint area proxy: 2*pi/sqrt(det conic) (pixels^2), coverage of a 16x16 tile = min(1, area/256)
det=np.maximum(con[:,0]*con[:,2]-con[:,1]**2,1e-12)
area=2*np.pi/np.sqrt(det)
m=op*np.minimum(1.0, area/256.0)          # mean optical-depth-ish contribution to the tile
nt=len(ranges)
n=(ranges[:,1]-ranges[:,0]).astype(np.float64)
tile_steps=bs.sum(1)
# entries until cumulative m exceeds thresholds
feat={}
for thr in (2.0,4.0,6.0,9.2):
    k=np.zeros(nt)
    for t in range(nt):
        a,b=ranges[t]
        if b<=a: continue
        c=np.cumsum(m[pl[a:b]])
        i=np.searchsorted(c,thr)
        k[t]=min(i+1,b-a)
    feat[f'entries_until_cum_{thr}']=k
feat['list_length']=n
print(name,'tiles',nt,'mean steps/tile',tile_steps.mean())
for k,v in feat.items():
    print('  corr(tile steps, %s) = %.3f'%(k,np.corrcoef(tile_steps,v)[0,1]))
# what would LPT with the best predictor give? PS model
best=max(feat,key=lambda k:abs(np.corrcoef(tile_steps,feat[k])[0,1]))
# forward launch: per block jobs; order tiles by predictor descending; blocks of a tile adjacent
order=np.argsort(-feat[best],kind='stable')
jobs_idx=bs.reshape(-1)
jobs_pred=bs[order].reshape(-1)
jobs_lpt=np.sort(bs.reshape(-1))[::-1]
for K in (8,):
    ideal=jobs_idx.sum()/(1024*F[K])
    print('  PS model K=8: index %.3f  predictor(%s) %.3f  exact LPT %.3f'%(ps_makespan(jobs_idx,1024,K)/ideal, best, ps_makespan(jobs_pred,1024,K)/ideal, ps_makespan(jobs_lpt,1024,K)/ideal))
