"""The one-node rasterization() (wrapper._FusedRasterization: compact gradient rows per visible Gaussian,
projection backward with dense by-products) against the operator-by-operator composition of the same HIP
kernels (projection_with_opacities -> isect_tiles -> rasterize_to_pixels[_with_depth]), which is what
gsplat.rendering.rasterization chains (reference call site mtgs_scene_graph.py:641-662)."""
import math

import pytest
import torch

from tests.util import assert_tile_lists, listed

pytestmark = pytest.mark.gpu


def _compose(P, vm, K, W, H, render_mode, rasterize_mode, absgrad, backgrounds=None):
    from mtgs_amd import wrapper as w
    Cn = vm.shape[0]
    radii, means2d, depths, conics, comps, opac = w.projection_with_opacities(
        P["means"], P["quats"], P["scales"], vm, K, P["opacities"], W, H,
        calc_compensations=(rasterize_mode == "antialiased"))
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    _, isect_ids, flat = w.isect_tiles(means2d, radii, depths, 16, tw, th)
    off = w.isect_offset_encode(isect_ids, Cn, tw, th)
    cols = P["colors"].unsqueeze(0).expand(Cn, -1, -1)
    if render_mode == "RGB":
        render, alpha = w.rasterize_to_pixels(means2d, conics, cols, opac, W, H, 16, off, flat, backgrounds=backgrounds,
                                              absgrad=absgrad)
    else:
        render, alpha = w.rasterize_to_pixels_with_depth(means2d, conics, cols, opac, depths, render_mode == "RGB+ED",
                                                         W, H, 16, off, flat, backgrounds=backgrounds, absgrad=absgrad)
    return render, alpha, {"means2d": means2d, "depths": depths, "conics": conics, "opacities": opac, "radii": radii,
                           "flatten_ids": flat, "isect_ids": isect_ids, "isect_offsets": off}


def _scene(N, D, seed, dev):
    from mtgs_amd.synthetic import make_scene
    sc = make_scene(N, seed=seed, sh_degree=None, extent=(10.0, 3.0, 10.0))
    g = torch.Generator().manual_seed(seed + 100)
    sc["colors"] = torch.rand(N, D, generator=g)
    return {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}


@pytest.mark.parametrize("cams,D,render_mode,rasterize_mode,absgrad,extra", [
    (1, 3, "RGB+ED", "antialiased", True, False),   # what MTGS.py drives
    (1, 3, "RGB", "classic", False, True),          # + losses on info tensors
    (2, 3, "RGB+ED", "antialiased", True, True),    # several cameras: generic projection backward
    (1, 6, "RGB+ED", "antialiased", True, False),   # RGB + normals (7 blended channels)
    (1, 16, "RGB", "classic", True, False),         # wide colours: un-staged by-product stores
    (1, 15, "RGB+D", "classic", False, False),
])
def test_fused_equals_composition(hip_lib, cams, D, render_mode, rasterize_mode, absgrad, extra):
    from mtgs_amd import rasterization
    from mtgs_amd.synthetic import make_camera
    dev = torch.device("cuda")
    N, W, H = 30_000, 333, 250
    vms, Ks = zip(*[make_camera(W, H, yaw_deg=30.0 * c) for c in range(cams)])
    K = torch.cat(Ks).to(dev)
    g = torch.Generator().manual_seed(5)
    DT = D + (render_mode != "RGB")
    Gc, Ga = torch.randn(cams, H, W, DT, generator=g).to(dev), torch.randn(cams, H, W, 1, generator=g).to(dev)
    bg = torch.rand(cams, D, generator=g).to(dev).requires_grad_(True) if D == 3 else None
    Gm, Gd = torch.randn(cams, N, 2, generator=g).to(dev), torch.randn(cams, N, generator=g).to(dev)
    Gq, Go = torch.randn(cams, N, 3, generator=g).to(dev), torch.randn(cams, N, generator=g).to(dev)

    results = []
    for fused in (True, False):
        P = _scene(N, D, 3, dev)
        vm = torch.cat(vms).to(dev).requires_grad_(True)
        bgi = None if bg is None else bg.detach().clone().requires_grad_(True)
        if fused:
            render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, K, W, H,
                                                packed=False, render_mode=render_mode, rasterize_mode=rasterize_mode,
                                                absgrad=absgrad, backgrounds=bgi)
        else:
            render, alpha, info = _compose(P, vm, K, W, H, render_mode, rasterize_mode, absgrad, backgrounds=bgi)
        info["means2d"].retain_grad()
        loss = (render * Gc).sum() + (alpha * Ga).sum()
        if extra:
            vis = (info["radii"] > 0)
            loss = loss + (info["means2d"] * Gm * vis[..., None]).sum() + (info["depths"] * Gd * vis).sum() \
                + (info["conics"] * Gq * vis[..., None]).sum() * 1e-3 + (info["opacities"] * Go * vis).sum()
        loss.backward()
        grads = {k: P[k].grad for k in P}
        grads["viewmat"] = vm.grad
        if bgi is not None:
            grads["bg"] = bgi.grad
        grads["means2d.grad"] = info["means2d"].grad
        if absgrad:
            grads["means2d.absgrad"] = info["means2d"].absgrad
        results.append((render.detach(), alpha.detach(), info, grads))
    (r1, a1, i1, g1), (r0, a0, i0, g0) = results
    assert torch.equal(r1, r0) and torch.equal(a1, a0)                      # same kernels, same inputs
    assert torch.equal(i1["radii"], i0["radii"])
    assert_tile_lists(i1, i0)                     # (the fused path's lists: ordered sublists of the operator path's = gsplat's)
    assert torch.equal(i1["means2d"].detach(), i0["means2d"].detach())
    assert set(g1) == set(g0)
    for k in g0:
        assert g1[k] is not None and g0[k] is not None, k
        assert g1[k].shape == g0[k].shape and g1[k].is_contiguous(), k
        scale = float(g0[k].abs().max())
        err = float((g1[k] - g0[k]).abs().max())
        assert scale > 0 and err <= 2e-4 * scale + 1e-6, f"{k}: {err} vs {scale}"   # fp32 atomics in another order


def test_fused_no_visible_gaussians(hip_lib):
    """Every Gaussian behind the camera: zero image, zero gradients, nothing reads a row."""
    from mtgs_amd import rasterization
    from mtgs_amd.synthetic import make_camera
    dev = torch.device("cuda")
    P = _scene(1000, 3, 1, dev)
    with torch.no_grad():
        P["means"][:, 2] = -5.0 - P["means"][:, 2].abs()
    vm, K = make_camera(64, 48)
    render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm.to(dev), K.to(dev),
                                        64, 48, packed=False, render_mode="RGB+ED", absgrad=True)
    info["means2d"].retain_grad()
    (render.sum() + alpha.sum()).backward()
    assert int((info["radii"] > 0).sum()) == 0 and float(render.detach().abs().max()) == 0.0
    for k in ("means", "quats", "scales", "opacities", "colors"):
        assert P[k].grad is not None and float(P[k].grad.abs().max()) == 0.0
    assert float(info["means2d"].grad.abs().max()) == 0.0 and float(info["means2d"].absgrad.abs().max()) == 0.0


def test_randomised_differential_check(hip_lib):
    """tests/fuzz_gpu.py: 60 random configurations (sizes down to one Gaussian / images smaller than a tile, 1-2
    cameras, 1-16 channels, all render modes, nothing visible) -- fused node vs composition vs oracle, and the neighbour
    kernels vs their PyTorch formulations."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "tests" / "fuzz_gpu.py"), "--cases", "60", "--seed", "123"],
                       capture_output=True, text=True, timeout=900, cwd=str(root))
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.gpu
def test_speculative_sizing_overflow_repeats_the_frame_exactly(hip_lib):
    """The fused path enqueues binning + compositing with CAPACITIES before the host knows (n_vis, M).  Capacities that
    turn out too small (forced here) must give exactly the frame an exact-size run gives: same lists, same image, same
    gradients -- and nothing may be written out of bounds on the way (the truncated attempt is discarded)."""
    from mtgs_amd import rasterization, wrapper
    from tests.util import small_scene
    sc, vm, K = small_scene(N=4000, W=200, H=120, seed=11)
    dev = torch.device("cuda")

    def run(force):
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
        old = (wrapper._force_caps, wrapper.speculative_sizing)
        wrapper._force_caps, wrapper.speculative_sizing = force, force is not None
        try:
            r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm.to(dev), K.to(dev),
                                       200, 120, packed=False, render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
            (r.sum() + 2 * a.sum()).backward()
        finally:
            wrapper._force_caps, wrapper.speculative_sizing = old
        torch.cuda.synchronize()
        return r.detach(), a.detach(), info, {k: p.grad for k, p in P.items()}

    r0, a0, i0, g0 = run(None)
    n_vis, M = int((i0["radii"] > 0).sum()), i0["flatten_ids"].numel()      # (M: gsplat's count -- what the capacities are about)
    assert n_vis > 500 and M > 2000
    for caps in [(n_vis // 2, 1 << 16), (n_vis + 10, M // 3), (64, 64), (n_vis, M)]:
        r1, a1, i1, g1 = run(caps)
        for k in ("isect_offsets", "radii", "tiles_per_gauss"):
            assert torch.equal(i0[k], i1[k]), (caps, k)
        for k in ("flatten_ids", "isect_ids"):
            assert torch.equal(listed(i0, k), listed(i1, k)), (caps, k)
        assert torch.equal(r0, r1) and torch.equal(a0, a1), caps
        for k in g0:
            # (fp32 atomics accumulate in a different order from run to run)
            assert (g0[k] - g1[k]).abs().max() <= 2e-4 * g0[k].abs().max() + 1e-7, (caps, k)


@pytest.mark.gpu
def test_graph_mode_capture_replay_equals_eager(hip_lib):
    """mtgs_amd.graph_mode: forward + backward captured ONCE as a HIP graph (torch.cuda.graph), replayed with new
    camera / parameter VALUES in the same tensors; every replay must give the eager result of those values, the counts
    must be right on the device, and capacities that are too small must raise the overflow flag (never go out of bounds)."""
    import mtgs_amd
    from mtgs_amd import rasterization
    from mtgs_amd.synthetic import make_camera
    from tests.util import small_scene
    W, H, N = 200, 120, 4000
    sc, vm0, K = small_scene(N=N, W=W, H=H, seed=12)
    dev = torch.device("cuda")
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    vm, Kd = vm0.to(dev).clone(), K.to(dev)
    g = torch.Generator().manual_seed(4)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)

    def run():
        r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, Kd, W, H, packed=False,
                                   render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
        info["means2d"].retain_grad()
        torch.autograd.backward([r, a], [Gc, Ga])
        return r, a, info

    def eager():
        for p in P.values():
            p.grad = None
        r, a, info = run()
        torch.cuda.synchronize()
        return (r.detach().clone(), a.detach().clone(), {k: p.grad.clone() for k, p in P.items()}, info["means2d"].absgrad.clone(),
                int((info["radii"] > 0).sum()), listed(info).clone(), info["flatten_ids"].numel())

    ref0 = eager()
    n_vis, M = ref0[4], ref0[6]          # (gsplat's intersection count: the totals of the front end, the capacity's unit)
    for p in P.values():
        p.grad = torch.zeros_like(p)
    grads = [p.grad for p in P.values()]

    def capture(caps):
        gm = mtgs_amd.graph_mode(*caps)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), gm:
            torch._foreach_zero_(grads)
            run()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with gm, torch.cuda.graph(graph):
            torch._foreach_zero_(grads)
            out = run()
        return graph, out, gm

    graph, (r, a, info), gm = capture((n_vis + 500, M + 5000))
    assert info["flatten_ids"].numel() == M + 5000      # capacity-sized in graph mode

    def check(ref):
        graph.replay()
        torch.cuda.synchronize()
        assert int(info["n_visible"]) == ref[4] and int(info["n_intersections"]) == ref[6] and not bool(info["overflow"])
        assert int(info["n_listed"]) == ref[5].numel()
        assert torch.equal(info["flatten_ids"][:ref[5].numel()], ref[5])
        assert torch.equal(r, ref[0]) and torch.equal(a, ref[1])
        for k, p in P.items():
            assert (p.grad - ref[2][k]).abs().max() <= 2e-4 * ref[2][k].abs().max() + 1e-7, k
        assert (info["means2d"].absgrad - ref[3]).abs().max() <= 2e-4 * ref[3].abs().max() + 1e-7

    check(ref0)
    # new VALUES in the same tensors: another camera, perturbed parameters -> the replay follows
    vm.copy_(make_camera(W, H, yaw_deg=10.0)[0].to(dev))
    vm[0, :3, 3] += torch.tensor([0.1, -0.2, 0.3], device=dev)
    with torch.no_grad():
        P["means"] += 0.01
        P["opacities"].mul_(0.9)
    saved = [p.grad for p in P.values()]
    ref1 = eager()                      # (eager() detaches the .grad tensors; restore the graph's)
    for p, gr in zip(P.values(), saved):
        p.grad = gr
    assert ref1[4] + 500 > ref1[4] and ref1[5].numel() <= M + 5000 and ref1[4] <= n_vis + 500, "test scene: capacities must still fit"
    check(ref1)
    # capacities too small: flagged, truncated, nothing crashes
    for p in P.values():
        p.grad = torch.zeros_like(p)
    grads[:] = [p.grad for p in P.values()]
    g2, (r2, a2, i2), _ = capture((max(ref1[4] // 2, 64), max(ref1[5].numel() // 3, 1 << 16)))
    g2.replay()
    torch.cuda.synchronize()
    assert bool(i2["overflow"]) and int(i2["n_visible"]) == ref1[4] and torch.isfinite(r2).all()


@pytest.mark.parametrize("n,equal_depths", [(300, False), (1500, True), (3000, False), (4200, True), (6000, False), (40_000, True)])
def test_long_tile_lists_every_sort_path(hip_lib, lists_mode, n, equal_depths):
    """Binning without a global sort (csrc/bin3.hip): the per-tile sort has a one-wave path (< 1020 keys), a workgroup
    path (<= 4096; around 4200 Gaussians the nine lists straddle that limit inside the dispatch order's first length
    bucket), a 1024-thread path (<= 16384 keys in LDS) and a chunked path that merges through global memory.
    n Gaussians in front of a 48x40 image put ~n intersections into each of its 9 tiles; with equal_depths many of
    them share their depth bit for bit, so the order inside a tile is decided by the Gaussian index (gsplat: stable
    sort).  isect_ids / flatten_ids / offsets must equal the operator path's (one global radix sort) bit for bit."""
    from mtgs_amd import rasterization
    from mtgs_amd import wrapper as w
    from mtgs_amd.synthetic import make_camera
    dev = torch.device("cuda")
    W, H = 48, 40
    g = torch.Generator().manual_seed(n)
    vm, K = make_camera(W, H)
    vm, K = vm.to(dev), K.to(dev)
    # the camera sits at the origin looking down +z (identity view matrix): world z IS the depth, bit for bit
    z = torch.rand(n, generator=g) * 4.0 + 3.0
    if equal_depths:
        z = torch.round(z * 4.0) / 4.0                      # 17 distinct depths
    xy = (torch.rand(n, 2, generator=g) - 0.5) * 0.4
    means = torch.cat([xy * z[:, None], z[:, None]], dim=1).to(dev).contiguous()
    quats = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1).to(dev)
    scales = (torch.rand(n, 3, generator=g) * 0.6 + 0.3).to(dev)
    opac = (torch.rand(n, generator=g) * 0.2 + 0.02).to(dev)
    cols = torch.rand(n, 3, generator=g).to(dev)
    render, alpha, info = rasterization(means=means, quats=quats, scales=scales, opacities=opac, colors=cols, viewmats=vm,
                                        Ks=K, width=W, height=H, packed=False, render_mode="RGB+ED",
                                        rasterize_mode="antialiased", absgrad=True)
    radii, means2d, depths, conics, comps, oe = w.projection_with_opacities(means, quats, scales, vm, K, opac, W, H,
                                                                            calc_compensations=True)
    _, isect_ids, flat = w.isect_tiles(means2d, radii, depths, 16, 3, 3)
    off = w.isect_offset_encode(isect_ids, 1, 3, 3)
    lens = torch.diff(torch.cat([off.reshape(-1), torch.tensor([isect_ids.numel()], device=dev, dtype=off.dtype)]))
    assert int(lens.max()) > n // 2, "the construction should fill the tiles"
    if equal_depths:
        d = depths[0][radii[0] > 0]
        assert d.unique().numel() < d.numel(), "expected Gaussians with bit-identical depths"
    assert_tile_lists(info, {"isect_offsets": off, "flatten_ids": flat, "isect_ids": isect_ids})
    if lists_mode == "tight":     # (opacities 0.02 .. 0.22: the tight lists are shorter, but every sort path must still be reached)
        assert int(info["n_listed"]) > 0.5 * isect_ids.numel()
    r2, a2 = w.rasterize_to_pixels_with_depth(means2d, conics, cols.unsqueeze(0), oe, depths, True, W, H, 16, off, flat)
    assert torch.allclose(render, r2, atol=1e-5) and torch.allclose(alpha, a2, atol=1e-5)


def test_more_tile_rows_than_the_packed_path_takes(hip_lib):
    """A 32 x 66000 image has 4125 tile rows: beyond mtgs_bin3_supported (C * tile_h <= 4096), so the frame takes the
    gather-based kernels -- same ids as the operator path, and gradients flow."""
    from mtgs_amd import _lib, rasterization
    from mtgs_amd import wrapper as w
    dev = torch.device("cuda")
    W, H, n = 32, 66000, 4000
    assert not _lib.load().mtgs_bin3_supported(1, 2, 4125, 0) and _lib.load().mtgs_bin3_supported(1, 2, 4096, 0)
    g = torch.Generator().manual_seed(1)
    K = torch.tensor([[[40.0, 0.0, W / 2.0], [0.0, 40.0, H / 2.0], [0.0, 0.0, 1.0]]], device=dev)
    vm = torch.eye(4, device=dev)[None]
    z = torch.rand(n, generator=g) * 4.0 + 3.0
    y = (torch.rand(n, generator=g) - 0.5) * (H / 40.0) * z * 0.98
    x = (torch.rand(n, generator=g) - 0.5) * 0.5 * z
    means = torch.stack([x, y, z], 1).to(dev).requires_grad_(True)
    quats = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1).to(dev)
    scales = (torch.rand(n, 3, generator=g) * 0.3 + 0.05).to(dev)
    opac = (torch.rand(n, generator=g) * 0.5 + 0.2).to(dev)
    cols = torch.rand(n, 3, generator=g).to(dev)
    render, alpha, info = rasterization(means=means, quats=quats, scales=scales, opacities=opac, colors=cols, viewmats=vm, Ks=K,
                                        width=W, height=H, packed=False, render_mode="RGB+ED", rasterize_mode="antialiased")
    radii, means2d, depths, conics, comps, oe = w.projection_with_opacities(means.detach(), quats, scales, vm, K, opac, W, H,
                                                                            calc_compensations=True)
    _, isect_ids, flat = w.isect_tiles(means2d, radii, depths, 16, 2, 4125)
    assert isect_ids.numel() > n and torch.equal(info["flatten_ids"], flat) and torch.equal(info["isect_ids"], isect_ids)   # (gather path: gsplat's lists)
    (render.sum() + alpha.sum()).backward()
    assert means.grad is not None and float(means.grad.abs().sum()) > 0


@pytest.mark.parametrize("W,H", [(2560, 1440), (3840, 2160)])
def test_high_resolution_frames_take_the_packed_path(hip_lib, lists_mode, W, H):
    """14400 and 32400 tiles (one LDS histogram entry per tile: up to 32768): same ids and image as the operator path."""
    from mtgs_amd import _lib, rasterization
    from mtgs_amd import wrapper as w
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda")
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    assert _lib.load().mtgs_bin3_supported(1, tw, th, 0)
    N = 300_000
    sc = make_scene(N, seed=11, sh_degree=None)
    P = {k: v.to(dev) for k, v in sc.items()}
    vm, K = make_camera(W, H)
    vm, K = vm.to(dev), K.to(dev)
    cols = torch.rand(N, 3, device=dev)
    render, alpha, info = rasterization(means=P["means"], quats=P["quats"], scales=P["scales"], opacities=P["opacities"], colors=cols,
                                        viewmats=vm, Ks=K, width=W, height=H, packed=False, render_mode="RGB+ED",
                                        rasterize_mode="antialiased")
    radii, means2d, depths, conics, comps, oe = w.projection_with_opacities(P["means"], P["quats"], P["scales"], vm, K, P["opacities"],
                                                                            W, H, calc_compensations=True)
    _, isect_ids, flat = w.isect_tiles(means2d, radii, depths, 16, tw, th)
    off = w.isect_offset_encode(isect_ids, 1, tw, th)
    assert isect_ids.numel() > 100_000
    assert_tile_lists(info, {"isect_offsets": off, "flatten_ids": flat, "isect_ids": isect_ids})
    r2, a2 = w.rasterize_to_pixels_with_depth(means2d, conics, cols.unsqueeze(0), oe, depths, True, W, H, 16, off, flat)
    assert torch.allclose(render, r2, atol=1e-5) and torch.allclose(alpha, a2, atol=1e-5)


def test_many_cameras_in_one_call_take_the_packed_path(hip_lib, lists_mode):
    """24 cameras of 640x480: 24 x 30 = 720 (camera, tile row) bins, 28800 (camera, tile) bins -- one call, same ids as the
    operator path."""
    from mtgs_amd import _lib, rasterization
    from mtgs_amd import wrapper as w
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda")
    C, W, H, N = 24, 640, 480, 60_000
    assert _lib.load().mtgs_bin3_supported(C, 40, 30, 0)
    sc = make_scene(N, seed=4, sh_degree=None)
    P = {k: v.to(dev) for k, v in sc.items()}
    vms, Ks = zip(*[make_camera(W, H, yaw_deg=15.0 * c) for c in range(C)])
    vm, K = torch.cat(vms).to(dev), torch.cat(Ks).to(dev)
    cols = torch.rand(N, 3, device=dev)
    render, alpha, info = rasterization(means=P["means"], quats=P["quats"], scales=P["scales"], opacities=P["opacities"], colors=cols,
                                        viewmats=vm, Ks=K, width=W, height=H, packed=False, render_mode="RGB+ED",
                                        rasterize_mode="antialiased")
    radii, means2d, depths, conics, comps, oe = w.projection_with_opacities(P["means"], P["quats"], P["scales"], vm, K, P["opacities"],
                                                                            W, H, calc_compensations=True)
    _, isect_ids, flat = w.isect_tiles(means2d, radii, depths, 16, 40, 30)
    off = w.isect_offset_encode(isect_ids, C, 40, 30)
    assert_tile_lists(info, {"isect_offsets": off, "flatten_ids": flat, "isect_ids": isect_ids})


@pytest.mark.parametrize("N,W,H,C,D,needles", [(60_000, 320, 200, 1, 3, False), (200_000, 640, 480, 1, 3, False), (120_000, 640, 360, 3, 3, False),
                                               (500_000, 960, 540, 1, 6, False), (80_000, 640, 360, 2, 3, True)])
def test_tight_tile_lists_change_no_pixel_and_no_gradient(hip_lib, N, W, H, C, D, needles):
    """The opt-in tight tile lists (`with mtgs_amd.tight_lists():`) hold only the (tile, Gaussian) pairs whose {alpha >= 1/255} ellipse
    reaches a pixel centre of the tile (mtgs_bin3_build, MTGS_BIN3_TIGHT).  Against gsplat's lists (the default: every tile of the 3-sigma square): render and
    alphas are BIT-identical -- every pair left out is skipped pixel by pixel by gsplat's own `alpha < 1/255` rule -- the
    gradients agree to the order of the fp32 atomics, the lists are ordered sublists, and they are much shorter."""
    import mtgs_amd
    from mtgs_amd import rasterization
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda")
    sc = {k: v.to(dev) for k, v in make_scene(N, seed=5, sh_degree=None).items()}
    g = torch.Generator().manual_seed(2)
    cols = torch.rand(N, D, generator=g).to(dev)
    sc["opacities"] = (torch.rand(N, generator=g) ** 2).to(dev)          # (many faint Gaussians: small alpha >= 1/255 ellipses)
    if needles:      # long thin splats at every orientation (conics with a large off-diagonal term: the closed-form row spans'
        #              clamped tangent points), some of them crossing the whole image
        sc["scales"] = (torch.tensor([2.5, 0.02, 0.02]) * (0.2 + torch.rand(N, 1, generator=g))).to(dev)
        sc["quats"] = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1).to(dev)
    vms, Ks = zip(*[make_camera(W, H, yaw_deg=25.0 * c) for c in range(C)])
    vm, K = torch.cat(vms).to(dev), torch.cat(Ks).to(dev)
    Gc, Ga = torch.randn(C, H, W, D + 1, generator=g).to(dev), torch.randn(C, H, W, 1, generator=g).to(dev)
    out = {}
    for mode in ("gsplat", "tight", "gsplat again"):
        P = {k: sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")}
        c = cols.clone().requires_grad_(True)
        v = vm.clone().requires_grad_(True)
        with mtgs_amd.exact_lists(mode != "tight"):
            r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], c, v, K, W, H, packed=False,
                                       render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
            info["means2d"].retain_grad()
            torch.autograd.backward([r, a], [Gc, Ga])
        grads = {k: p.grad for k, p in P.items()}
        grads.update(colors=c.grad, viewmat=v.grad, means2d=info["means2d"].grad, absgrad=info["means2d"].absgrad)
        out[mode] = (r.detach(), a.detach(), info, grads)
    (r0, a0, i0, g0), (r1, a1, i1, g1) = out["gsplat"], out["tight"]
    assert torch.equal(r0, r1) and torch.equal(a0, a1)
    g2 = out["gsplat again"][3]
    for k in g0:
        # the order of the fp32 atomics is all that can differ.  Splats that cross hundreds of tiles sum ~10^4 contributions of both
        # signs per Gaussian: there two runs of the SAME lists differ by more than 2e-4 of the largest gradient, so the bound is
        # calibrated on that run-to-run difference (as tests/fuzz_gpu.py does)
        noise = float((g0[k] - g2[k]).abs().max())
        # (8x: ONE repeat is a small sample of that noise -- at 4x the needle case failed once in ~10 runs)
        assert float((g0[k] - g1[k]).abs().max()) <= 8.0 * noise + 2e-4 * float(g0[k].abs().max()) + 1e-7, (k, noise)
    for k in ("radii", "tiles_per_gauss"):
        assert torch.equal(i0[k], i1[k])
    assert i0.get("n_listed") is None and i0["flatten_ids"].numel() == i1["flatten_ids"].numel()     # (gsplat's M in both modes)
    assert_tile_lists(i1, {k: i0[k] for k in ("isect_offsets", "flatten_ids", "isect_ids")}, tight=True)
    assert int(i1["n_listed"]) < 0.8 * i0["flatten_ids"].numel(), (int(i1["n_listed"]), i0["flatten_ids"].numel())


@pytest.mark.parametrize("mode", ["gsplat", "tight"])
@pytest.mark.parametrize("N,W,H,C,rmode", [(60_000, 320, 200, 1, "classic"), (150_000, 640, 360, 2, "antialiased")])
def test_meta_lists_feed_rasterize_to_pixels(hip_lib, mode, N, W, H, C, rmode):
    """gsplat's pattern: the tensors of rasterization()'s `meta` go back into the operators --
        rasterize_to_pixels(info["means2d"], info["conics"], colors, info["opacities"], W, H, 16, info["isect_offsets"], info["flatten_ids"])
    -- and reproduce `render` / `alphas` BIT FOR BIT, with the lists passed as CLONES (nothing rides on the tensor objects) and
    gsplat's convention that the last tile's range ends at flatten_ids.numel().  Default call: gsplat's lists, numel == M.
    Under tight_lists(): the tensors keep gsplat's length and hold sentinels behind n_listed, which the operator stops at; the
    offsets re-encoded from the (padded) isect_ids are the tight lists' own.  Backward through the re-fed lists works too."""
    import mtgs_amd
    from mtgs_amd import rasterization, wrapper as w
    from mtgs_amd.synthetic import make_camera
    dev = torch.device("cuda")
    P = _scene(N, 3, 11, dev)
    vms, Ks = zip(*[make_camera(W, H, yaw_deg=20.0 * c) for c in range(C)])
    vm, K = torch.cat(vms).to(dev), torch.cat(Ks).to(dev)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    with mtgs_amd.tight_lists(mode == "tight"), torch.no_grad():
        render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, K, W, H, packed=False,
                                            rasterize_mode=rmode)
    M = int(info["tiles_per_gauss"].sum())
    assert info["flatten_ids"].numel() == M == info["isect_ids"].numel()          # gsplat's length in both modes
    if mode == "gsplat":
        assert "n_listed" not in info
    else:
        n = int(info["n_listed"])
        assert 0 < n < M and bool((info["flatten_ids"][n:] == -1).all()) and bool((info["flatten_ids"][:n] >= 0).all())
    flat, ids, off = info["flatten_ids"].clone(), info["isect_ids"].clone(), info["isect_offsets"].clone()
    assert torch.equal(w.isect_offset_encode(ids, C, tw, th), off)
    cols = P["colors"].detach().unsqueeze(0).expand(C, -1, -1).contiguous().requires_grad_(True)
    m2d = info["means2d"].clone().requires_grad_(True)
    r2, a2 = w.rasterize_to_pixels(m2d, info["conics"].clone(), cols, info["opacities"].clone(), W, H, 16, off, flat)
    assert torch.equal(r2, render) and torch.equal(a2, alpha)
    (r2.sum() + a2.sum()).backward()
    assert torch.isfinite(cols.grad).all() and torch.isfinite(m2d.grad).all() and float(cols.grad.abs().max()) > 0


def test_graph_mode_tensors_end_in_sentinels(hip_lib):
    """Inside graph_mode the list tensors are capacity-sized in BOTH list modes: the entries behind the listed pairs are sentinels
    (never uninitialised), info["n_listed"] is the device count of listed pairs."""
    import mtgs_amd
    from mtgs_amd import rasterization
    from mtgs_amd.synthetic import make_camera
    dev = torch.device("cuda")
    N, W, H = 50_000, 320, 200
    P = _scene(N, 3, 12, dev)
    vm, K = (t.to(dev) for t in make_camera(W, H))
    with torch.no_grad():
        _, _, eager = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, K, W, H, packed=False)
    M, n_vis = eager["flatten_ids"].numel(), int((eager["radii"] > 0).sum())
    for tight in (False, True):
        with mtgs_amd.tight_lists(tight), mtgs_amd.graph_mode(n_vis + 100, M + 777), torch.no_grad():
            _, _, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, K, W, H, packed=False)
        torch.cuda.synchronize()
        n = int(info["n_listed"])
        assert info["flatten_ids"].numel() == M + 777 and (n == M if not tight else 0 < n < M)
        assert bool((info["flatten_ids"][n:] == -1).all()) and bool((info["isect_ids"][n:] == info["isect_ids"][-1]).all())
        if not tight:
            assert torch.equal(info["flatten_ids"][:M], eager["flatten_ids"]) and torch.equal(info["isect_ids"][:M], eager["isect_ids"])
