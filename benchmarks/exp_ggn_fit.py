import sys, time, torch
sys.path.insert(0,'.')
from oracle import nerf_oracle as O
from uncertainty_nerf_gs_amd import plugin, synthetic, ops, render, models
dev=torch.device('cuda:0')
t = synthetic.make_scene_tensors(seed=4, kind="laplace", log2T=14, prop_log2T=12)
sc = O.scene_from_tensors(t)
sd = synthetic.scene_to_device(t, dev)
H,W=6,8
o,d,_ = O.generate_rays(synthetic.orbit_c2w(2.0), 0.9*W,0.9*W,W/2,H/2,H,W); o=o.reshape(-1,3); d=d.reshape(-1,3)
gd,gr = O.laplace_ggn_diag(sc,o,d)
f=t["field"]
dm=torch.cat([f["density_w"].reshape(-1), f["density_b"].reshape(-1)]); rm=torch.cat([f["head_w"][2].reshape(-1), f["head_b"][2].reshape(-1)])
# same spacing bins as the oracle (CPU sampler) -> isolates the GGN kernels
bins,_,_ = O.proposal_sample(o,d,sc.near,sc.far,sc.prop_nets,sc.num_prop,sc.num_nerf,sc.prop_average_init_density)
a=torch.zeros(65,device=dev); b=torch.zeros(195,device=dev)
ops.laplace_ggn_diag(o.to(dev),d.to(dev),bins.to(dev).contiguous(),sd.field,dm,rm,sc.near,sc.far,a,b)
print("rel err density", ((a.cpu()-gd).abs()/ (gd.abs()+1e-6*gd.max())).max().item(), "rgb", ((b.cpu()-gr).abs()/(gr.abs()+1e-6*gr.max())).max().item())
# throughput at the real sizes
t2 = synthetic.make_scene_tensors(seed=0, kind="laplace")
sd2 = synthetic.scene_to_device(t2, dev)
g=torch.Generator().manual_seed(0)
R=4096
cam=synthetic.CAMERA_1080P
oo,ddd,_=O.generate_rays(synthetic.orbit_c2w(0.3), cam["fx"],cam["fy"],cam["cx"],cam["cy"],cam["H"],cam["W"])
idx=torch.randint(0,cam["H"]*cam["W"],(R,),generator=g)
oo=oo.reshape(-1,3)[idx].to(dev).contiguous(); ddd=ddd.reshape(-1,3)[idx].to(dev).contiguous()
f2=t2["field"]
dm2=torch.cat([f2["density_w"].reshape(-1), f2["density_b"].reshape(-1)]).to(dev); rm2=torch.cat([f2["head_w"][2].reshape(-1), f2["head_b"][2].reshape(-1)]).to(dev)
a=torch.zeros(65,device=dev); b=torch.zeros(195,device=dev)
def step():
    sb,_=render.sample_rays(sd2,oo,ddd,None,want_prop_depth=False)
    ops.laplace_ggn_diag(oo,ddd,sb,sd2.field,dm2,rm2,sd2.near,sd2.far,a,b)
for _ in range(3): step()
torch.cuda.synchronize(); t0=time.time()
for _ in range(50): step()
torch.cuda.synchronize(); dt=(time.time()-t0)/50
print("GGN batch of 4096 rays: %.3f ms -> 1000 iterations in %.2f s (%.2f Mrays/s)"%(dt*1e3, dt*1000, R/dt/1e6))
