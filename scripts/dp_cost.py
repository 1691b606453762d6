#!/usr/bin/env python3
"""Local (non-communication) cost of the sparse gradient exchange at the headline size, on one GPU: the
sender side (visibility map + ordered pack) and the receiver's one-pass reduction over W senders whose
visible sets are those of W cameras at yaw r * 45 deg of the benchmark scene (the rows themselves are random)."""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd import _lib, dist as mdist, wrapper  # noqa: E402
from mtgs_amd._lib import call, ptr  # noqa: E402
from mtgs_amd.synthetic import make_camera, make_scene  # noqa: E402

import argparse
import ctypes as C

import numpy as np

_ap = argparse.ArgumentParser()
_ap.add_argument("--width", type=int, default=1920)
_ap.add_argument("--height", type=int, default=1080)
_ap.add_argument("--no-render-leg", action="store_true")
_ap.add_argument("--worlds", type=int, nargs="*", default=[2, 4, 8])
_ap.add_argument("--pmc", action="store_true", help="counter run: per world only 3 one-pass touched reductions, then 3 sparse-write ones")
_args = _ap.parse_args()
dev = torch.device("cuda")
N, K, W, H = 2_000_000, 16, _args.width, _args.height
print(f"== {N} Gaussians, {W}x{H}")
sc = {k: v.to(dev) for k, v in make_scene(N, seed=0, sh_degree=None).items()}
means = sc["means"]
g = torch.Generator().manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g).to(dev)
st = torch.cuda.current_stream().cuda_stream


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for world in _args.worlds:
    ex = mdist.SparseGradExchange(N, K, dev)
    nw, L = ex.n_words, ex.meta_len
    metas = torch.zeros(world, L, dtype=torch.int32, device=dev)
    rows, counts = [], []
    for r in range(world):
        vm, Kmat = make_camera(W, H, yaw_deg=45.0 * r)
        radii = wrapper.fully_fused_projection(means, None, sc["quats"], sc["scales"], vm.to(dev), Kmat.to(dev), W, H)[0][0]
        vis = (radii > 0)
        vg = [mk(N, 3), mk(N, 4), mk(N, 3), mk(N), mk(N, 3)]
        m = metas[r]
        pack = lambda: call("mtgs_dp_pack_ordered", N, ptr(radii), ptr(vg[0]), ptr(vg[1]), ptr(vg[2]), ptr(vg[3]), ptr(vg[4]),
                            ptr(m[4:]), ptr(m[4 + 2 * nw:]), ptr(m), ptr(ex.block_counts), ptr(ex.rows), N, st)
        if r == 0 and world == 2:
            print("sender: visibility map + ordered pack: %.1f us (n_vis %d)" % (t(pack), int(vis.sum())))
        pack()
        cam = torch.inverse(vm)[0, :3, 3].to(dev)
        m[1:4].copy_(cam.view(torch.int32))
        counts.append(int(m[0].item()))
        rows.append(ex.rows[:counts[-1]].clone())
    cap = max(counts)
    recv = torch.zeros(world, cap, 16, device=dev)
    for r in range(world):
        recv[r, :counts[r]] = rows[r]
    cams = metas[:, 1:4].contiguous().view(torch.float32)
    out = [torch.empty(N, 3, device=dev), torch.empty(N, 4, device=dev), torch.empty(N, 3, device=dev),
           torch.empty(N, device=dev), torch.empty(N, K, 3, device=dev)]
    red = lambda: call("mtgs_dp_reduce", world, N, K, 3, ptr(means), ptr(metas[:, 4:]), ptr(metas[:, 4 + 2 * nw:]), L * 4,
                       ptr(recv), cap * 16, ptr(cams), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), ptr(out[4]), 0, -1, st)
    covered = sum(counts)
    if not _args.pmc:
        print("world %d: one-pass reduce over %d rows (%d MB received): %.1f us" % (world, covered, world * cap * 64 >> 20, t(red)))
    # ---- the TOUCHED form (finish_touched / finish_touched_chunked): 40 % of every sender's rows carry a gradient (the headline
    # scene's share), compacted with their own map by mtgs_dp_touched_pack; the reduction over them, in one pass and in index chunks
    tb = {"scratch": torch.empty(nw, dtype=torch.int64, device=dev), "blocks": torch.empty(nw // 256 + 2, dtype=torch.int32, device=dev),
          "totals": torch.zeros(1, dtype=torch.int64, device=dev)}
    gz = torch.Generator().manual_seed(5)
    pad = -(-L // 16) * 16
    tcap = int(0.45 * cap) + 1024
    Lt = tcap * 16 + pad
    recv_t = torch.zeros(world, Lt, device=dev)
    t_counts, t_pack = [], None
    for r in range(world):
        rr = rows[r].clone()
        rr[(torch.rand(rr.shape[0], generator=gz) < 0.6).to(dev), :15] = 0.0
        blk = recv_t[r]
        mb = blk.data_ptr() + 4 * tcap * 16
        pk = lambda rr=rr, blk=blk, mb=mb: call("mtgs_dp_touched_pack", rr.shape[0], ptr(rr), N, ptr(tb["scratch"]), mb + 16, mb + 16 + 8 * nw, mb,
                                                ptr(tb["totals"]), ptr(tb["blocks"]), blk.data_ptr(), tcap, st)
        if r == 0:
            t_pack = t(pk)
        pk()
        blk.view(torch.int32)[tcap * 16 + 1:tcap * 16 + 4].copy_(metas[r, 1:4])
        t_counts.append(int(blk.view(torch.int32)[tcap * 16].item()))
    ri = recv_t.view(torch.int32)
    m0 = tcap * 16
    red_t = lambda b0=0, b1=-1: call("mtgs_dp_reduce_slices_cap", world, N, K, 3, ptr(means), ri.data_ptr() + 4 * (m0 + 4),
                                     ri.data_ptr() + 4 * (m0 + 4 + 2 * nw), Lt * 4, ptr(recv_t), Lt, tcap, ptr(cams), ptr(out[0]), ptr(out[1]),
                                     ptr(out[2]), ptr(out[3]), ptr(out[4]), b0, b1, C.c_uint64((1 << world) - 1), 1, K * 3, st)
    if _args.pmc:
        for o in out:
            o.zero_()
        torch.cuda.synchronize()
        for _ in range(3):
            red_t()
        for _ in range(3):
            call("mtgs_dp_reduce_slices_cap", world, N, K, 3, ptr(means), ri.data_ptr() + 4 * (m0 + 4),
                 ri.data_ptr() + 4 * (m0 + 4 + 2 * nw), Lt * 4, ptr(recv_t), Lt, tcap, ptr(cams), ptr(out[0]), ptr(out[1]),
                 ptr(out[2]), ptr(out[3]), ptr(out[4]), 0, -1, C.c_uint64((1 << world) - 1), 1 | 2, K * 3, st)
        torch.cuda.synchronize()
        print("pmc: world %d: touched rows %d, received %d MB" % (world, sum(t_counts), world * Lt * 4 >> 20))
        continue
    line_t = "world %d, touched rows only (%d of %d rows, %d MB received; pack %.1f us): one-pass reduce %.1f us" % (
        world, sum(t_counts), covered, world * Lt * 4 >> 20, t_pack, t(red_t))
    for nch in (2, 4):
        per = -(-(-(-N // nch)) // 2048) * 2048
        bounds = list(range(0, N, per)) + [N]
        line_t += "; in %d index chunks %.1f us" % (nch, t(lambda: [red_t(bounds[c], bounds[c + 1]) for c in range(len(bounds) - 1)]))
    # ... into PRE-ZEROED outputs (the zeros ride on the frame's compositing forward): touched Gaussians only, untouched tiles skipped
    for o in out:
        o.zero_()
    red_s = lambda: call("mtgs_dp_reduce_slices_cap", world, N, K, 3, ptr(means), ri.data_ptr() + 4 * (m0 + 4),
                         ri.data_ptr() + 4 * (m0 + 4 + 2 * nw), Lt * 4, ptr(recv_t), Lt, tcap, ptr(cams), ptr(out[0]), ptr(out[1]),
                         ptr(out[2]), ptr(out[3]), ptr(out[4]), 0, -1, C.c_uint64((1 << world) - 1), 1 | 2, K * 3, st)
    line_t += "; SPARSE write into pre-zeroed outputs %.1f us" % t(red_s)
    print(line_t)
    # ---- the same sums as ROWS of the union of the senders' visible sets (finish(rows=True)): one traversal (geometry + colour
    # in one pass), and one traversal PER sender (MTGS's multi-traversal step: T passes, dense [N, T, K, 3] against T row sets)
    for T in (1, world):
        masks = [(1 << world) - 1] + ([(1 << world) - 1] if T == 1 else [1 << r for r in range(world)])
        n_sub = len(masks)
        masks_dev = torch.tensor(np.asarray(masks, dtype=np.uint64).view(np.int64), dtype=torch.int64, device=dev)
        uw = torch.empty((n_sub, nw), dtype=torch.int64, device=dev)
        up = torch.empty((n_sub, nw), dtype=torch.int32, device=dev)
        totals = torch.empty(n_sub, dtype=torch.int64, device=dev)
        scratch = torch.empty(n_sub * (nw // 256 + 1), dtype=torch.int32, device=dev)
        union = lambda: call("mtgs_dp_union", world, N, ptr(metas[:, 4:]), L * 4, n_sub, ptr(masks_dev), ptr(uw), ptr(up), ptr(totals),
                             ptr(scratch), st)
        union()
        tot = [int(v) >> 32 for v in totals.tolist()]
        geo_rows = torch.empty(max(tot[0], 1), 16, device=dev)
        geo_ro, geo_ids = torch.empty(N, dtype=torch.int32, device=dev), torch.empty(max(tot[0], 1), dtype=torch.int32, device=dev)
        coef = [(torch.empty(max(tot[1 + j], 1), 3 * K, device=dev), torch.empty(N, dtype=torch.int32, device=dev)) for j in range(n_sub - 1)]

        def red_rows():
            for j in range(n_sub - 1):
                call("mtgs_dp_reduce_rows", world, N, K, 3, ptr(means), ptr(metas[:, 4:]), ptr(metas[:, 4 + 2 * nw:]), L * 4, ptr(recv),
                     cap * 16, ptr(cams), 0, -1, C.c_uint64(masks[1 + j]), ptr(geo_rows) if j == 0 else None, ptr(uw[0]), ptr(up[0]),
                     ptr(geo_ro) if j == 0 else None, ptr(geo_ids) if j == 0 else None, geo_rows.shape[0], ptr(coef[j][0]),
                     ptr(uw[1 + j]), ptr(up[1 + j]), ptr(coef[j][1]), coef[j][0].shape[0], 3 * K, st)
        # the same in ONE launch (mtgs_dp_reduce_rows_groups: a tile is walked once for the geometry and every colour group)
        from mtgs_amd.nodes import upload_table
        gt = np.zeros(n_sub - 1, dtype=mdist._DP_GROUP)
        for j in range(n_sub - 1):
            gt[j] = (masks[1 + j], coef[j][0].data_ptr(), uw[1 + j].data_ptr(), up[1 + j].data_ptr(), coef[j][1].data_ptr(), coef[j][0].shape[0])
        gtab = upload_table(gt, dev)
        red_groups = lambda: call("mtgs_dp_reduce_rows_groups", world, N, K, 3, ptr(means), ptr(metas[:, 4:]), ptr(metas[:, 4 + 2 * nw:]),
                                  L * 4, ptr(recv), cap * 16, ptr(cams), 0, -1, n_sub - 1, ptr(gtab), ptr(geo_rows), ptr(uw[0]), ptr(up[0]),
                                  ptr(geo_ro), ptr(geo_ids), geo_rows.shape[0], 3 * K, st)
        red_rows(); ref = [c[0].clone() for c in coef] + [geo_rows.clone()]
        red_groups(); torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(ref, [c[0] for c in coef] + [geo_rows]))
        line = "world %d, %d traversal(s): rows of the union (%d geometry rows, %d colour rows): union maps %.1f us + reduce %.1f us (one launch for all groups: %.1f us, bit-identical %s)" % (
            world, T, tot[0], sum(tot[1:]), t(union), t(red_rows), t(red_groups), same)
        if T > 1:
            outT = torch.empty(N, T, K, 3, device=dev)

            def red_slices():
                for tt in range(T):
                    call("mtgs_dp_reduce_slices", world, N, K, 3, ptr(means), ptr(metas[:, 4:]), ptr(metas[:, 4 + 2 * nw:]), L * 4, ptr(recv),
                         cap * 16, ptr(cams), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), outT.data_ptr() + tt * K * 3 * 4, 0, -1,
                         C.c_uint64(1 << tt), int(tt == 0), T * K * 3, st)
            line += "  | dense [N, %d, K, 3] in %d passes: %.1f us" % (T, T, t(red_slices, 5))
            del outT
        print(line)


# ---- the render leg of a data-parallel step (integrated form, one process): forward + backward down to the wire rows,
# then finish() with this rank as the only sender (the reduction over W senders is timed above)
def render_leg():
    from mtgs_amd import rasterization, spherical_harmonics
    scs = make_scene(N, seed=0, sh_degree=3)
    P = {k: v.to(dev).requires_grad_(True) for k, v in scs.items()}
    vm, Kmat = make_camera(W, H)
    vm, Kmat = vm.to(dev).requires_grad_(True), Kmat.to(dev)
    cam = torch.inverse(vm.detach())[0, :3, 3]
    gg = torch.Generator().manual_seed(1)
    Gc, Ga = torch.randn(1, H, W, 4, generator=gg).to(dev), torch.randn(1, H, W, 1, generator=gg).to(dev)
    ex = mdist.SparseGradExchange(N, K, dev, chunks=4)

    def dp_step(finish=True):
        sh = spherical_harmonics(3, P["means"].detach() - cam, P["coeffs"].detach())
        r, a, _ = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm, Kmat, W, H, cam)
        torch.autograd.backward([r, a], [Gc, Ga])
        out = ex.finish(P["means"], 3) if finish else None
        if not finish:
            ex._pending = None
        return out

    def plain_step():
        for p in P.values():
            p.grad = None
        sh = spherical_harmonics(3, P["means"].detach() - cam, P["coeffs"])
        rgb = torch.clamp(sh + 0.5, 0.0, 1.0)
        r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, Kmat, W, H, packed=False,
                                   render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
        info["means2d"].retain_grad()
        torch.autograd.backward([r, a], [Gc, Ga])

    print("single-GPU step, dense gradients (bench.py N = 1):              %.1f us" % t(plain_step, 20))
    print("data-parallel render leg (forward + backward to wire rows):     %.1f us" % t(lambda: dp_step(False), 20))
    print("data-parallel step, 1 sender (render leg + reduce of own rows): %.1f us" % t(dp_step, 20))
    print("   phases of the last step (ms):", {k: round(v, 3) for k, v in ex.phases_ms().items()})


if not _args.no_render_leg:
    render_leg()
