#!/usr/bin/env python3
"""Randomised differential test on the GPU (test infrastructure: lives under tests/ because it uses the oracle as the
checker; run directly, `python tests/fuzz_gpu.py --cases 500 --seed 3`, or through tests/test_gpu_fused.py):
  * rasterization() (one autograd node, compact gradient rows) vs the operator-by-operator composition of the same
    kernels, forward bit-for-bit and gradients to fp32-atomic accuracy, over random sizes / cameras / channel counts /
    render modes, including degenerate inputs (nothing visible, one Gaussian, huge splats, image smaller than a tile);
  * a sample of the cases against the CPU oracle;
  * node_gaussians / masked_ssim / masked_l1 / update_statistics vs their PyTorch formulations at random sizes;
  * camera_space_normals vs its oracle, output_head / depth_ncc_loss vs their PyTorch formulations, the batched
    collect_gaussians vs per-node calls.
Exits non-zero on the first mismatch, printing the failing configuration (re-run with --seed / --case)."""
import argparse
import math
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd import rasterization, wrapper  # noqa: E402
from tests.util import assert_tile_lists  # noqa: E402
from mtgs_amd.densify import update_statistics  # noqa: E402
from mtgs_amd.loss import masked_l1, masked_ssim  # noqa: E402
from mtgs_amd.nodes import node_gaussians  # noqa: E402

dev = torch.device("cuda")


def rand_case(rng):
    N = int(rng.choice([1, 2, 63, 64, 65, 300, 1023, 1025, 4000, 20000]))
    W = int(rng.integers(5, 420)); H = int(rng.integers(5, 300))
    C = int(rng.choice([1, 1, 1, 2]))
    D = int(rng.choice([1, 2, 3, 3, 3, 4, 6, 7, 8, 15, 16]))
    mode = str(rng.choice(["RGB", "RGB+D", "RGB+ED"])) if D + 1 in wrapper.SUPPORTED_CHANNELS or D in wrapper.SUPPORTED_CHANNELS else "RGB"
    if mode != "RGB" and (D + 1) not in wrapper.SUPPORTED_CHANNELS:
        mode = "RGB"
    if mode == "RGB" and D not in wrapper.SUPPORTED_CHANNELS:
        D = 3
    return dict(N=N, W=W, H=H, C=C, D=D, mode=mode, aa=bool(rng.integers(2)), absgrad=bool(rng.integers(2)),
                bg=bool(rng.integers(2)) and mode == "RGB", scale=float(rng.choice([0.02, 0.1, 0.5, 3.0])),
                spread=float(rng.choice([0.5, 3.0, 10.0])), behind=bool(rng.integers(8) == 0), seed=int(rng.integers(1 << 30)))


def build(cfg):
    g = torch.Generator().manual_seed(cfg["seed"])
    N, C = cfg["N"], cfg["C"]
    means = (torch.rand(N, 3, generator=g) * 2 - 1) * cfg["spread"]
    means[:, 2] = means[:, 2].abs() + 0.5
    if cfg["behind"]:
        means[:, 2] = -means[:, 2]
    P = {"means": means, "quats": torch.randn(N, 4, generator=g), "scales": torch.rand(N, 3, generator=g) * cfg["scale"] + 1e-3,
         "opacities": torch.rand(N, generator=g), "colors": torch.rand(N, cfg["D"], generator=g)}
    vms = []
    for c in range(C):
        a = 0.3 * c
        R = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], dtype=torch.float32)
        vm = torch.eye(4); vm[:3, :3] = R; vm[:3, 3] = torch.tensor([0.1 * c, 0.0, 0.2])
        vms.append(vm)
    K = torch.tensor([[0.8 * cfg["W"], 0, cfg["W"] / 2], [0, 0.8 * cfg["W"], cfg["H"] / 2], [0, 0, 1]], dtype=torch.float32)
    return P, torch.stack(vms), K[None].repeat(C, 1, 1)


def compose(P, vm, K, cfg, bg):
    W, H, C = cfg["W"], cfg["H"], cfg["C"]
    radii, m2d, dep, con, comp, opac = wrapper.projection_with_opacities(P["means"], P["quats"], P["scales"], vm, K, P["opacities"],
                                                                         W, H, calc_compensations=cfg["aa"])
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, ids, flat = wrapper.isect_tiles(m2d, radii, dep, 16, tw, th)
    off = wrapper.isect_offset_encode(ids, C, tw, th)
    cols = P["colors"].unsqueeze(0).expand(C, -1, -1)
    if cfg["mode"] == "RGB":
        r, a = wrapper.rasterize_to_pixels(m2d, con, cols, opac, W, H, 16, off, flat, backgrounds=bg, absgrad=cfg["absgrad"])
    else:
        r, a = wrapper.rasterize_to_pixels_with_depth(m2d, con, cols, opac, dep, cfg["mode"] == "RGB+ED", W, H, 16, off, flat,
                                                      backgrounds=bg, absgrad=cfg["absgrad"])
    return r, a, {"means2d": m2d, "radii": radii, "flatten_ids": flat, "isect_ids": ids, "isect_offsets": off}


def run_paths(cfg, P0, vm0, K0, Gc, Ga, bg0):
    W, H = cfg["W"], cfg["H"]
    res = []
    for fused in (True, False):
        P = {k: v.to(dev).requires_grad_(True) for k, v in P0.items()}
        vm = vm0.to(dev).requires_grad_(True)
        bg = None if bg0 is None else bg0.to(dev)
        if fused:
            r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, K0.to(dev), W, H,
                                       packed=False, render_mode=cfg["mode"], rasterize_mode="antialiased" if cfg["aa"] else "classic",
                                       absgrad=cfg["absgrad"], backgrounds=bg)
        else:
            r, a, info = compose(P, vm, K0.to(dev), cfg, bg)
        info["means2d"].retain_grad()
        ((r * Gc).sum() + (a * Ga).sum()).backward()
        grads = {k: P[k].grad for k in P}
        grads["viewmat"] = vm.grad
        grads["m2d"] = info["means2d"].grad
        if cfg["absgrad"]:
            grads["abs"] = info["means2d"].absgrad
        res.append((r.detach(), a.detach(), info, grads))
    return res


def check_raster(cfg, with_oracle):
    P0, vm0, K0 = build(cfg)
    W, H, C, D = cfg["W"], cfg["H"], cfg["C"], cfg["D"]
    DT = D + (cfg["mode"] != "RGB")
    g = torch.Generator().manual_seed(cfg["seed"] + 1)
    Gc, Ga = torch.randn(C, H, W, DT, generator=g).to(dev), torch.randn(C, H, W, 1, generator=g).to(dev)
    bg0 = torch.rand(C, D, generator=g) if cfg["bg"] else None
    (r1, a1, i1, g1), (r0, a0, i0, g0) = run_paths(cfg, P0, vm0, K0, Gc, Ga, bg0)
    assert torch.equal(r1, r0) and torch.equal(a1, a0), "forward differs"
    assert torch.equal(i1["radii"], i0["radii"]) and torch.equal(i1["radii"], i0["radii"]), "binning differs"
    assert_tile_lists(i1, i0)
    noise = None
    for k in g0:
        if g0[k] is None and g1[k] is None:
            continue
        a_, b_ = g1[k], g0[k]
        if a_ is None or b_ is None:   # no gradient produced on one side: must be all zero on the other
            t = a_ if a_ is not None else b_
            assert float(t.abs().max()) == 0.0, f"grad {k}: None vs non-zero"
            continue
        assert torch.isfinite(a_).all(), f"grad {k} not finite"
        scale = float(b_.abs().max())
        err = float((a_ - b_).abs().max())
        # fp32 atomics in another order: 2e-3 of the tensor's max (the bar of the oracle tests).  Scenes of image-filling
        # splats (hundreds of intersections per Gaussian, large cancelling terms) are ill-conditioned: there TWO RUNS OF
        # THE SAME PATH differ by up to ~1e-2, so the bound is calibrated on the composition path's own run-to-run noise.
        if err > 2e-3 * scale + 1e-6:
            if noise is None:
                again = run_paths(cfg, P0, vm0, K0, Gc, Ga, bg0)
                noise = {kk: float((again[1][3][kk] - g0[kk]).abs().max()) for kk in g0 if g0[kk] is not None and again[1][3][kk] is not None}
            assert err <= 4.0 * noise.get(k, 0.0) + 2e-3 * scale, f"grad {k}: {err} vs {scale} (run-to-run noise {noise.get(k)})"
    if with_oracle and cfg["C"] == 1 and cfg["N"] <= 4000:
        from oracle import oracle as orc
        a = {k: v.numpy() for k, v in P0.items()}
        rr, aa, m = orc.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], vm0.numpy(), K0.numpy(), W, H,
                                      render_mode=cfg["mode"], rasterize_mode="antialiased" if cfg["aa"] else "classic",
                                      backgrounds=None if bg0 is None else bg0.numpy())
        assert np.array_equal(i1["radii"].cpu().numpy(), m["radii"]), "radii vs oracle"
        assert_tile_lists(i1, m)
        ok = ~m["critical"] if "critical" in m else np.ones(rr.shape[:3], bool)
        d = np.abs(r1.cpu().numpy() - rr).max(-1)
        tol = 1e-4 * max(1.0, float(np.abs(rr).max()))
        assert (d[ok] <= tol).all(), f"render vs oracle: {d[ok].max()}"


def check_neighbours(rng):
    N = int(rng.choice([1, 63, 64, 65, 777, 4099]))
    K = int(rng.choice([16, 16, 9, 4]))
    model_deg = {16: 3, 9: 2, 4: 1}[K]
    n = int(rng.integers(0, model_deg + 1))
    T = int(rng.choice([0, 0, 2, 5]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    P = {"means": torch.randn(N, 3, generator=g) * 4, "scales": torch.randn(N, 3, generator=g) - 1, "quats": torch.randn(N, 4, generator=g),
         "opacities": torch.randn(N, 1, generator=g), "features_dc": torch.randn(N, 3, generator=g)}
    P["features_rest"] = torch.randn(*((N, T, K - 1, 3) if T else (N, K - 1, 3)), generator=g) * 0.3
    if T:
        P["features_adapters"] = torch.randn(N, T, 3, generator=g) * 0.2
    t = int(rng.integers(0, T)) if T else None
    c2w = torch.eye(4)[None, :3].clone(); c2w[0, :, 3] = torch.randn(3, generator=g)
    cot = {"scales": torch.randn(N, 3, generator=g), "quats": torch.randn(N, 4, generator=g), "opacities": torch.randn(N, generator=g),
           "rgbs": torch.randn(N, 3, generator=g)}
    from mtgs_amd import spherical_harmonics
    out = []
    for fused in (True, False):
        D = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
        if fused:
            o = node_gaussians(D["means"], D["scales"], D["quats"], D["opacities"], D["features_dc"], D["features_rest"], c2w.to(dev), n,
                               model_deg, features_dc_add=D.get("features_adapters"), traversal_index=t)
        else:
            dc = D["features_dc"] + D["features_adapters"][:, t] if T else D["features_dc"]
            rest = D["features_rest"][:, t] if T else D["features_rest"]
            col = torch.cat((dc[:, None], rest), 1)
            vd = D["means"].detach() - c2w.to(dev)[..., :3, 3]
            vd = vd / vd.norm(dim=-1, keepdim=True)
            o = {"scales": torch.exp(D["scales"]), "quats": D["quats"] / D["quats"].norm(dim=-1, keepdim=True),
                 "opacities": torch.sigmoid(D["opacities"]).squeeze(-1),
                 "rgbs": torch.clamp(spherical_harmonics(n, vd, col) + 0.5, 0.0, 1.0)}
        sum((o[k] * cot[k].to(dev)).sum() for k in cot).backward()
        out.append(({k: o[k].detach() for k in cot}, {k: D[k].grad for k in D if k != "means"}))
    for k in cot:
        assert torch.allclose(out[0][0][k], out[1][0][k], rtol=1e-5, atol=2e-6), f"node {k} (N={N} K={K} n={n} T={T})"
    for k in out[1][1]:
        s = float(out[1][1][k].abs().max())
        assert float((out[0][1][k] - out[1][1][k]).abs().max()) <= 2e-5 * s + 1e-7, f"node grad {k} (N={N} K={K} n={n} T={T})"
    # loss head
    H, W = int(rng.integers(11, 200)), int(rng.integers(11, 260))
    gt = torch.rand(H, W, 3, generator=g).to(dev)
    pred0 = torch.rand(H, W, 3, generator=g)
    mask = (torch.rand(H, W, 1, generator=g) > 0.3).to(dev)
    mask[H // 2, W // 2] = True
    from oracle import ssim_oracle
    vref, gref = ssim_oracle.masked_ssim(gt.cpu().numpy(), pred0.numpy(), mask.cpu().numpy(), with_grad=True)
    p = pred0.to(dev).requires_grad_(True)
    v = masked_ssim(gt, p, mask); v.backward()
    assert abs(float(v.detach()) - vref) <= 1e-5 and np.abs(p.grad.cpu().numpy() - gref).max() <= 5e-5 * np.abs(gref).max() + 1e-9, f"ssim {H}x{W}"
    p2 = pred0.to(dev).requires_grad_(True)
    l = masked_l1(gt, p2, mask); l.backward()
    pr = pred0.to(dev).double().requires_grad_(True)
    lr = torch.abs(gt.double() - pr)[mask.squeeze(-1)].mean(); lr.backward()
    assert abs(float(l.detach()) - float(lr.detach())) <= 2e-6 and torch.allclose(p2.grad.double(), pr.grad, rtol=1e-5, atol=1e-12), f"l1 {H}x{W}"


def check_later_neighbours(rng):
    """camera_space_normals vs oracle/normals_oracle.py, output_head and depth_ncc_loss vs their PyTorch formulations,
    the batched collect_gaussians vs per-node calls -- at random sizes."""
    import torch.nn.functional as F
    from mtgs_amd.loss import depth_ncc_loss, output_head
    from mtgs_amd.nodes import camera_space_normals, collect_gaussians
    from oracle import normals_oracle as no
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    # ---- normals
    N = int(rng.choice([1, 63, 64, 65, 257, 5000]))
    quats = torch.randn(N, 4, generator=g); scales = torch.exp(torch.randn(N, 3, generator=g)); means = torch.randn(N, 3, generator=g) * 8
    A = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    c2w = torch.cat([A, torch.randn(3, 1, generator=g)], 1)[None]
    G = torch.randn(N, 3, generator=g)
    q = quats.to(dev).requires_grad_(True)
    nrm = camera_space_normals(q, scales.to(dev), means.to(dev), c2w.to(dev))
    (nrm * G.to(dev)).sum().backward()
    ref_n = no.normals_fwd(quats.numpy(), scales.numpy(), means.numpy(), c2w.numpy())
    ref_g = no.normals_bwd(quats.numpy(), scales.numpy(), means.numpy(), c2w.numpy(), G.numpy())
    d = c2w[0, :, 3][None] - means
    safe = ((torch.from_numpy(ref_n).float() @ A.T * (d / d.norm(dim=-1, keepdim=True))).sum(-1).abs() > 1e-5).numpy()
    if safe.any():
        assert np.abs(nrm.detach().cpu().numpy() - ref_n)[safe].max() < 3e-6, f"normals N={N}"
        assert np.abs(q.grad.cpu().numpy() - ref_g)[safe].max() < 3e-5 * max(1.0, np.abs(ref_g).max()), f"normals grad N={N}"
    # ---- output head
    H, W = int(rng.integers(3, 120)), int(rng.integers(3, 150))
    D = int(rng.choice([4, 7, 8]))
    nc = 3 if D >= 7 else -1
    render = torch.rand(1, H, W, D, generator=g) * 1.4 - 0.2
    render[..., -1] = torch.rand(1, H, W, generator=g) * 20
    alpha = torch.rand(1, H, W, 1, generator=g) * (torch.rand(1, H, W, 1, generator=g) > 0.1)
    bg = torch.rand(3, generator=g)
    E = torch.eye(3, 4) + 0.1 * torch.randn(3, 4, generator=g)
    cots = [torch.randn(H, W, 3, generator=g), torch.randn(H, W, 3, generator=g), torch.randn(H, W, 1, generator=g), torch.randn(H, W, 3, generator=g)]

    def head_ref(render, alpha, bg, E):
        rgb = torch.clamp(render[..., :3] + (1 - alpha) * bg, 0.0, 1.0).squeeze(0)
        app = torch.clamp(rgb.matmul(E[:3, :3]) + E[None, None, :3, 3], 0, 1)
        dd = render[..., -1:]
        depth = torch.where(alpha > 0, dd, dd.detach().max()).squeeze(0)
        normal = None
        if nc >= 0:
            nm = render[..., nc:nc + 3].squeeze(0)
            normal = (nm / nm.norm(dim=-1, keepdim=True) + 1) / 2
        return rgb, app, depth, normal

    outs = []
    for fn, dt, dv in ((head_ref, torch.float64, "cpu"), (lambda r, a, b, e: output_head(r, a, b, e, depth=True, normal_channel=nc), torch.float32, dev)):
        Pp = [t.to(device=dv, dtype=dt).requires_grad_(True) for t in (render, alpha, bg, E)]
        o = fn(*Pp)
        sum((x * c.to(device=dv, dtype=dt)).sum() for x, c in zip(o, cots) if x is not None).backward()
        outs.append((o, [t.grad for t in Pp]))
    for x, y in zip(outs[0][0], outs[1][0]):
        if x is not None:
            assert float((x.detach() - y.detach().cpu().double()).abs().max()) < 5e-6, f"head value {H}x{W} D={D}"
    for x, y, nm in zip(outs[0][1], outs[1][1], ("render", "alpha", "bg", "E")):
        diff = (x - y.cpu().double()).abs()
        sc = float(x.abs().max()) + 1e-12
        if nm in ("render", "alpha"):   # a pixel within fp32 rounding of a clamp edge may take the other branch
            assert float((diff > 3e-5 * sc).double().mean()) < 5e-3, f"head grad {nm} {H}x{W} D={D}"
    # ---- depth NCC
    H, W = int(rng.integers(20, 150)), int(rng.integers(20, 200))
    k = int(rng.choice([4, 7, 16, 32])); st = int(rng.choice([k, max(1, k // 2), 5]))
    gt = torch.rand(H, W, 1, generator=g) * 30 + 1
    pred0 = gt + torch.randn(H, W, 1, generator=g)
    mask = torch.rand(H, W, 1, generator=g) > 0.002

    def ncc_ref(pd, gd, m):
        pd, gd = pd.squeeze(-1), gd.squeeze(-1)
        pad = k // 2
        mm = m.squeeze(-1).to(pd.dtype)
        pp = F.unfold(pd[None, None], kernel_size=k, padding=pad, stride=st)
        gp = F.unfold(gd[None, None], kernel_size=k, padding=pad, stride=st)
        valid = F.unfold(mm[None, None], kernel_size=k, padding=pad, stride=st).all(dim=1).squeeze(0)
        pp, gp = pp[:, :, valid], gp[:, :, valid]
        pc, gc = pp - pp.mean(dim=1, keepdim=True), gp - gp.mean(dim=1, keepdim=True)
        ps, gs = torch.sqrt((pc ** 2).mean(dim=1, keepdim=True) + 1e-8), torch.sqrt((gc ** 2).mean(dim=1, keepdim=True) + 1e-8)
        return 1 - ((pc / ps) * (gc / gs)).mean(dim=1).mean(), int(valid.sum())

    pr = pred0.double().requires_grad_(True)
    ref, nv = ncc_ref(pr, gt.double(), mask)
    pq = pred0.to(dev).requires_grad_(True)
    val = depth_ncc_loss(pq, gt.to(dev), k, st, mask=mask.to(dev))
    if nv == 0:
        assert torch.isnan(val), f"ncc {H}x{W} k={k} s={st}: expected NaN"
    else:
        ref.backward(); val.backward()
        assert abs(float(val.detach()) - float(ref.detach())) < 3e-5, f"ncc {H}x{W} k={k} s={st}"
        assert float((pq.grad.cpu().double() - pr.grad).abs().max()) <= 2e-3 * float(pr.grad.abs().max()) + 1e-9, f"ncc grad {H}x{W} k={k} s={st}"
    # ---- batched collect vs per-node
    sizes = [int(x) for x in rng.choice([0, 1, 63, 64, 65, 300, 1000], size=int(rng.integers(1, 7)))]
    if sum(sizes) == 0:
        sizes.append(5)
    c2 = torch.eye(4)[None, :3].clone(); c2[0, :, 3] = torch.randn(3, generator=g)
    base = []
    for n_ in sizes:
        p = {"means": torch.randn(n_, 3, generator=g) * 4, "scales": torch.randn(n_, 3, generator=g) - 1, "quats": torch.randn(n_, 4, generator=g),
             "opacities": torch.randn(n_, 1, generator=g), "features_dc": torch.randn(n_, 3, generator=g),
             "features_rest": torch.randn(n_, 15, 3, generator=g) * 0.3}
        if rng.integers(2):
            qq = torch.randn(4, generator=g)
            p["instance_quat"], p["instance_trans"] = qq / qq.norm(), torch.randn(3, generator=g)
        base.append(p)
    tot = sum(sizes)
    cot = {"means": torch.randn(tot, 3, generator=g), "scales": torch.randn(tot, 3, generator=g), "quats": torch.randn(tot, 4, generator=g),
           "opacities": torch.randn(tot, generator=g), "rgbs": torch.randn(tot, 3, generator=g)}
    res = []
    for batched in (True, False):
        Pn = [{k_: v.to(dev).requires_grad_(True) for k_, v in p.items()} for p in base]
        if batched:
            o = collect_gaussians(Pn, c2.to(dev), 3, 3)
        else:
            parts = [node_gaussians(p["means"], p["scales"], p["quats"], p["opacities"], p["features_dc"], p["features_rest"], c2.to(dev), 3, 3,
                                    instance_quat=p.get("instance_quat"), instance_trans=p.get("instance_trans")) for p in Pn]
            o = {k_: torch.cat([q_[k_] for q_ in parts], 0) for k_ in cot}
        sum((o[k_] * cot[k_].to(dev)).sum() for k_ in cot).backward()
        res.append(({k_: o[k_].detach() for k_ in cot}, [{k_: v.grad for k_, v in p.items()} for p in Pn]))
    for k_ in cot:
        assert torch.equal(res[0][0][k_], res[1][0][k_]), f"collect {k_} sizes={sizes}"
    for ga, gb in zip(res[0][1], res[1][1]):
        for k_ in ga:
            assert (ga[k_] is None) == (gb[k_] is None), f"collect grad {k_} sizes={sizes}"
            if ga[k_] is not None:
                if k_.startswith("instance_"):
                    assert torch.allclose(ga[k_], gb[k_], rtol=1e-4, atol=1e-4 * float(gb[k_].abs().max()) + 1e-6), f"collect pose grad sizes={sizes}"
                else:
                    assert torch.equal(ga[k_], gb[k_]), f"collect grad {k_} sizes={sizes}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    for i in range(args.cases):
        cfg = rand_case(rng)
        try:
            check_raster(cfg, with_oracle=(i % 5 == 0))
            if i % 3 == 0:
                check_neighbours(rng)
            if i % 3 == 1:
                check_later_neighbours(rng)
        except Exception as e:  # noqa: BLE001
            print(f"FAIL case {i}: {cfg}\n  {type(e).__name__}: {e}")
            raise
    print(f"fuzz ok: {args.cases} rasterization cases, {(args.cases + 2) // 3} + {(args.cases + 1) // 3} neighbour cases")


if __name__ == "__main__":
    main()
