#!/usr/bin/env python3
"""Per-stage timing of the hot path through the C ABI (HIP events on the launch stream), plus
tile-list statistics.  Development tool: python scripts/kbench.py [--n 2000000 --variant mtgs]"""
import argparse
import ctypes as C
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd import _lib, wrapper  # noqa: E402
from mtgs_amd._lib import call, host_i64, ptr  # noqa: E402
from mtgs_amd.synthetic import make_camera, make_scene  # noqa: E402


_FLUSH = None


def timeit(fn, reps=10, warm=2):
    """KBENCH_COLD=1: a 768 MB buffer is rewritten before every timed call, so inputs come from HBM and not from
    the 256 MB Infinity Cache a back-to-back repetition would hit (what a kernel sees inside a real step)."""
    global _FLUSH
    import os
    cold = bool(os.environ.get("KBENCH_COLD"))
    if cold and _FLUSH is None:
        _FLUSH = torch.empty(192 << 20, dtype=torch.float32, device="cuda")
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        if cold:
            _FLUSH.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--variant", default="mtgs")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--lib", default=None, help="development: another build of libmtgs_rast.so (scripts/build_variant.py)")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--extra-channels", type=int, default=0, help="extra colour channels (MTGS.py blends 3 normal channels too)")
    args = ap.parse_args()
    if args.lib:
        _lib.use_library(args.lib)
    dev = torch.device("cuda")
    W, H = args.width, args.height
    mtgs = args.variant == "mtgs"
    sc = make_scene(args.n, seed=0, sh_degree=3 if mtgs else None)
    vm, K = make_camera(W, H)
    d = {k: v.to(dev) for k, v in sc.items()}
    vm, K = vm.to(dev), K.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    N = args.n
    res = {}
    if mtgs:
        dirs = d["means"] - torch.inverse(vm)[0, :3, 3]
        rgb = torch.empty(N, 3, device=dev)
        res["sh_fwd"] = timeit(lambda: call("mtgs_sh_fwd", N, 16, 3, ptr(dirs), ptr(d["coeffs"]), None, ptr(rgb), st), args.reps)
        vcoef = torch.empty_like(d["coeffs"]); vcol = torch.randn(N, 3, device=dev)
        res["sh_bwd"] = timeit(lambda: call("mtgs_sh_bwd", N, 16, 3, ptr(dirs), ptr(d["coeffs"]), None, ptr(vcol), ptr(vcoef), None, st), args.reps)
        colors = torch.clamp(rgb + 0.5, 0, 1)
    else:
        colors = d["colors"]
    radii, means2d, depths, conics, comps = wrapper.fully_fused_projection(d["means"], None, d["quats"], d["scales"], vm, K, W, H, calc_compensations=mtgs)
    res["project_fwd"] = timeit(lambda: wrapper.fully_fused_projection(d["means"], None, d["quats"], d["scales"], vm, K, W, H, calc_compensations=mtgs), args.reps)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    res["isect_tiles(all)"] = timeit(lambda: wrapper.isect_tiles(means2d, radii, depths, 16, tw, th), args.reps)
    tpg, ids, flat = wrapper.isect_tiles(means2d, radii, depths, 16, tw, th)
    M = flat.numel()
    # sort alone
    ws = C.c_size_t(0); call("mtgs_sort_workspace_bytes", M, C.byref(ws))
    w = torch.empty(ws.value, dtype=torch.uint8, device=dev)
    _, uids, uflat = wrapper.isect_tiles(means2d, radii, depths, 16, tw, th, sort=False)
    ko, vo = torch.empty_like(uids), torch.empty_like(uflat)
    ki, vi = uids.clone(), uflat.clone()
    def do_sort():
        ki.copy_(uids); vi.copy_(uflat)
        call("mtgs_sort_pairs", M, 46, ptr(ki), ptr(vi), ptr(ko), ptr(vo), ptr(w), ws.value, st)
    res["sort(+2 copies)"] = timeit(do_sort, args.reps)
    res["offsets"] = timeit(lambda: wrapper.isect_offset_encode(ids, 1, tw, th), args.reps)
    off = wrapper.isect_offset_encode(ids, 1, tw, th)
    opac = d["opacities"][None].contiguous()
    if mtgs:
        opac = (opac * comps).contiguous()
    cols = colors[None].contiguous()
    if args.extra_channels:
        cols = torch.cat([cols, torch.rand(1, N, args.extra_channels, device=dev)], -1).contiguous()
    DC = cols.shape[-1]
    dep = depths if mtgs else None
    D = DC + (1 if mtgs else 0)
    ed = 1 if mtgs else 0
    render = torch.empty(1, H, W, D, device=dev); alphas = torch.empty(1, H, W, 1, device=dev)
    last = torch.empty(1, H, W, dtype=torch.int32, device=dev)
    import os
    order = torch.empty(tw * th, dtype=torch.int32, device=dev)
    res["tile_schedule"] = timeit(lambda: call("mtgs_tile_schedule", 1, tw, th, ptr(off), M, ptr(order), st), args.reps)
    optr = None if os.environ.get("NO_ORDER") else ptr(order)
    fwd = lambda: call("mtgs_blend_fwd", 1, N, DC, ptr(means2d), ptr(conics), ptr(cols), ptr(opac), None, ptr(dep), ed,
                       W, H, 16, tw, th, ptr(off), ptr(flat), M, ptr(render), ptr(alphas), ptr(last), optr, st)
    res["blend_fwd"] = timeit(fwd, args.reps)
    g = torch.Generator(device="cpu").manual_seed(1)
    vr = torch.randn(1, H, W, D, generator=g).to(dev); va = torch.randn(1, H, W, 1, generator=g).to(dev)
    if os.environ.get("DENSE_GRADS"):  # six dense gsplat-style arrays
        v2d = torch.zeros_like(means2d); vab = torch.zeros_like(means2d) if mtgs else None
        vcon = torch.zeros_like(conics); vcl = torch.zeros_like(cols); vop = torch.zeros_like(opac)
        vdp = torch.zeros_like(depths) if mtgs else None
        gstr, pstr = None, None
        rank = None
    else:  # views of one interleaved buffer, as mtgs_amd.wrapper passes them
        RS = -(-(8 + D) // 16) * 16
        if os.environ.get("DENSE_ROWS"):   # one row per Gaussian (rasterize_to_pixels' own backward)
            G, rank = torch.zeros(1, N, RS, device=dev), None
        else:                              # one row per VISIBLE Gaussian (the fused rasterization node)
            vis = radii.reshape(-1) > 0
            rank = (torch.cumsum(vis.int(), 0) - 1).int().contiguous()
            G = torch.zeros(1, int(vis.sum()), RS, device=dev)
        v2d, vab, vcon, vop = G[..., 0:2], (G[..., 2:4] if mtgs else None), G[..., 4:7], G[..., 7]
        vcl, vdp = G[..., 8:8 + DC], (G[..., 8 + DC] if mtgs else None)
        gstr, pstr = host_i64([RS] * 6), host_i64([RS, RS, RS, 1, RS])
    bwd = lambda: call("mtgs_blend_bwd", 1, N, DC, ptr(means2d), ptr(conics), ptr(cols), ptr(opac), None, ptr(dep), ed,
                       W, H, 16, tw, th, ptr(off), ptr(flat), M, ptr(alphas), ptr(last), ptr(render), ptr(vr), ptr(va),
                       ptr(v2d), ptr(vab), ptr(vcon), ptr(vcl), ptr(vdp), ptr(vop), gstr, ptr(rank), optr, st)
    res["blend_bwd"] = timeit(bwd, args.reps)
    vm_ = torch.empty_like(d["means"]); vq = torch.empty_like(d["quats"]); vs = torch.empty_like(d["scales"]); vvm = torch.empty_like(vm)
    vdep = torch.randn_like(depths) if (vdp is None or rank is None) else vdp
    vopn = torch.empty_like(d["opacities"])
    if rank is None and pstr is not None:
        pstr = host_i64([RS, 1, RS, 1, RS])
    dense_out = rank is not None
    vids = torch.nonzero(radii.reshape(-1) > 0).int().reshape(-1).contiguous() if dense_out else None
    n_vis_k = vids.numel() if dense_out else 0
    vws = torch.empty(max(n_vis_k, 1), 12, device=dev) if dense_out else None
    d2 = torch.empty_like(means2d) if dense_out else None
    dab = torch.empty_like(means2d) if (dense_out and mtgs) else None
    dcl = torch.empty_like(cols) if dense_out else None
    res["project_bwd"] = timeit(lambda: call("mtgs_project_bwd", 1, N, ptr(d["means"]), ptr(d["quats"]), ptr(d["scales"]), ptr(vm), ptr(K), W, H, 0.3,
                                              ptr(radii), ptr(conics), ptr(comps), ptr(d["opacities"]), ptr(v2d), ptr(vdep), ptr(vcon), None,
                                              ptr(vop), ptr(vm_), ptr(vq), ptr(vs), ptr(vvm), ptr(vopn), pstr, ptr(rank), ptr(vab), ptr(vcl), DC,
                                              host_i64([RS, RS]) if dense_out else None, ptr(d2), ptr(dab), ptr(dcl), ptr(vids), n_vis_k, ptr(vws), None, None, None, None, None, None, st), args.reps)
    n_vis = int((radii > 0).sum())
    print(f"N={N} {W}x{H} variant={args.variant} n_vis={n_vis} M={M} D={D}")
    tot = 0.0
    for k, (med, mn) in res.items():
        print(f"  {k:20s} median {med*1e3:9.1f} us   min {mn*1e3:9.1f} us")
        if k != "sort(+2 copies)":
            tot += med
    print(f"  {'sum(stages)':20s} {tot*1e3:9.1f} us")
    if args.stats:
        o = off.flatten().long()
        end = torch.cat([o[1:], torch.tensor([M], device=dev)])
        ln = (end - o).float()
        lastmax = torch.nn.functional.max_pool2d(last.float()[None], 16, ceil_mode=True).flatten()
        proc = (lastmax - o.float() + 1).clamp(min=0) * (ln > 0)
        for name, v in (("tile list length", ln), ("bwd processed per tile", proc)):
            q = torch.quantile(v, torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev))
            print(f"  {name}: mean {v.mean():.0f} p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} max {q[3]:.0f} sum {v.sum():.0f}")


if __name__ == "__main__":
    main()
