"""Development: distribution of the tile-list lengths of the benchmark scenes (what the per-tile sort works on)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from mtgs_amd import rasterization
from mtgs_amd.synthetic import make_scene, make_camera
dev = torch.device("cuda")
for (N, W, H) in ((2_000_000, 1920, 1080), (2_000_000, 960, 540), (500_000, 1920, 1080), (8_000_000, 1920, 1080)):
    sc = make_scene(N, seed=0, sh_degree=None)
    P = {k: v.to(dev) for k, v in sc.items()}
    vm, K = make_camera(W, H)
    cols = torch.rand(N, 3, device=dev)
    r, a, info = rasterization(means=P["means"], quats=P["quats"], scales=P["scales"], opacities=P["opacities"], colors=cols,
                               viewmats=vm.to(dev), Ks=K.to(dev), width=W, height=H, packed=False, render_mode="RGB+ED",
                               rasterize_mode="antialiased")
    off = info["isect_offsets"].reshape(-1).cpu().numpy().astype(np.int64)
    M = info["isect_ids"].numel()
    L = np.diff(np.append(off, M))
    P2 = np.maximum(8, 2 ** np.ceil(np.log2(np.maximum(L, 1))))
    lg = np.log2(P2)
    work = (P2 / 2 * lg * (lg + 1) / 2).sum()
    print(f"N={N} {W}x{H}: M={M} tiles={L.size} mean={L.mean():.0f} p50={np.percentile(L,50):.0f} p90={np.percentile(L,90):.0f} "
          f"p99={np.percentile(L,99):.0f} max={L.max()}  classes<=256:{(L<=256).sum()} <=512:{((L>256)&(L<=512)).sum()} "
          f"<=1024:{((L>512)&(L<=1024)).sum()} <=2048:{((L>1024)&(L<=2048)).sum()} >2048:{(L>2048).sum()}  "
          f"padded/M={P2.sum()/M:.2f} comparators={work/1e6:.0f}M")
