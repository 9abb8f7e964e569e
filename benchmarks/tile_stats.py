import sys, os, torch, math
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
g.build()
from uncertainty_nerf_gs_amd import ops, splat, synthetic
dev = torch.device("cuda:0")
gp = {k: v.to(dev) for k, v in synthetic.make_splat_tensors(seed=7, N=1_000_000).items()}
H, W = 1080, 1920
c2w = synthetic.orbit_c2w(0.0, radius=2.5, height=0.5)
V = splat.viewmat_from_c2w(c2w)
pr = ops.splat_project(gp["means"], gp["scales"].contiguous(), 1.0, gp["quats"].contiguous(), V[:3], 1111.0, 1111.0, W / 2, H / 2, H, W, raw=True,
                       opacity_logits=gp["opacities"].reshape(-1).contiguous())
xys, depths, radii, conics, comp, tiles = pr[:6]
I, _, _, gids, bins = ops.splat_bin_sort(xys, depths, radii, tiles, H, W, tight=(conics, pr[7]), want_isect_ids=False)
n = (bins[:, 1] - bins[:, 0]).float().cpu()
q = torch.quantile(n, torch.tensor([0.5, 0.9, 0.99, 1.0]))
print("pairs", I, "tiles", n.numel(), "mean", n.mean().item(), "median/p90/p99/max", q.tolist())
srt = torch.sort(n, descending=True).values
print("top 10:", srt[:10].tolist(), " sum top 256 / total:", (srt[:256].sum() / n.sum()).item())
