"""MTGS's colours deferred THROUGH the clamp into rasterization() (mtgs_amd/wrapper.py::_LazySH.raster_source, csrc/viscolor.hip
mtgs_vis_color_*_dirs): `rgbs = torch.clamp(spherical_harmonics(n, viewdirs, colors) + 0.5, 0.0, 1.0)`
(/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:313-318) handed to `rasterization(colors=rgbs, ...)`
(mtgs_scene_graph.py:641-660) is evaluated for the Gaussians the projection finds VISIBLE only -- no [N, 3] colour tensor, no read of
the other coefficient rows -- and must give the render of the plain composition bit for bit, its gradients up to the order of the
compositing atomics, and fall back to the full evaluation wherever the rasterization cannot take the colours over."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(N, W, H, seed=3):
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda")
    sc = make_scene(N, seed=seed, sh_degree=3)
    vm, K = make_camera(W, H)
    vm, K = vm.to(dev), K.to(dev)
    g = torch.Generator().manual_seed(2)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
    cam = torch.inverse(vm)[0, :3, 3]
    return sc, vm, K, Gc, Ga, cam, dev


def _step(sc, vm, K, W, H, Gc, Ga, cam, dev, mode, form="mtgs", render_mode="RGB+ED", frozen=(), no_grad=False, normalise=False):
    """mode: 'raster' (deferred into the rasterization), 'fused' (SH + activation over all Gaussians), 'torch' (three torch operations)."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    P = {k: v.to(dev).requires_grad_(k not in frozen) for k, v in sc.items()}
    calls = []
    real = wrapper.call
    act = (lambda x: torch.clamp(x + 0.5, 0.0, 1.0)) if form == "mtgs" else (lambda x: torch.clamp_min(x + 0.5, 0.0))
    ctx = torch.no_grad() if no_grad else torch.enable_grad()
    try:
        wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        with wrapper.sh_lazy(mode != "torch", raster=mode == "raster"), ctx:
            dirs = P["means"].detach() - cam
            if normalise:      # (MTGS normalises in PyTorch first: vanilla_gaussian_splatting.py:314)
                dirs = dirs / dirs.norm(dim=-1, keepdim=True)
            rgb = act(spherical_harmonics(3, dirs, P["coeffs"]))
            render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False,
                                                render_mode=render_mode, absgrad=True, rasterize_mode="antialiased")
            if not no_grad:
                info["means2d"].retain_grad()
                nc = render.shape[-1]
                ((render * Gc[..., :nc]).sum() + (alpha * Ga).sum()).backward()
    finally:
        wrapper.call = real
    grads = {k: (None if v.grad is None else v.grad.clone()) for k, v in P.items()}
    if not no_grad:
        grads["means2d"], grads["absgrad"] = info["means2d"].grad.clone(), info["means2d"].absgrad.clone()
    return render.detach(), alpha.detach(), grads, calls, info


@pytest.mark.parametrize("form", ["mtgs", "gsplat"])
@pytest.mark.parametrize("render_mode", ["RGB+ED", "RGB"])
def test_deferred_colours_render_bit_for_bit_and_train_alike(hip_lib, form, render_mode):
    N, W, H = 300_000, 640, 368
    sc, vm, K, Gc, Ga, cam, dev = _scene(N, W, H)
    r1, a1, g1, c1, i1 = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "raster", form, render_mode)
    r2, a2, g2, c2, _ = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "fused", form, render_mode)
    r0, a0, g0, c0, i0 = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "torch", form, render_mode)
    # the visible Gaussians only, inside the rasterization: no SH launch of its own, no dense expansion, no torch clamp
    assert "mtgs_vis_color_fwd_dirs" in c1 and "mtgs_vis_color_bwd_dirs" in c1, c1
    assert not [n for n in c1 if n.startswith("mtgs_sh_")] and "mtgs_rows_expand" not in c1, c1
    assert "mtgs_sh_fwd_act" in c2 and "mtgs_sh_bwd_rows_act" in c2 and "mtgs_vis_color_fwd_dirs" not in c2
    assert "mtgs_sh_fwd" in c0 and "mtgs_vis_color_fwd_dirs" not in c0
    assert torch.equal(r1, r0) and torch.equal(a1, a0) and torch.equal(r2, r0)
    for k in ("radii", "means2d", "depths", "conics", "opacities", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert torch.equal(i1[k], i0[k]), k
    n_vis = int((i0["radii"] > 0).sum())
    assert 0 < n_vis < N // 2
    # d L / d coefficients: rows of the Gaussians with a cotangent -- the same Gaussians, the same values (the compositing atomics' order
    # aside); every other row is exactly zero
    assert torch.equal(g1["coeffs"] != 0, g0["coeffs"] != 0) and torch.equal(g2["coeffs"] != 0, g0["coeffs"] != 0)
    assert int((g1["coeffs"] != 0).any(dim=(1, 2)).sum()) <= n_vis
    for k in g0:
        scale = float(g0[k].abs().max())
        torch.testing.assert_close(g1[k], g0[k], rtol=1e-3, atol=1e-5 * scale)
        torch.testing.assert_close(g2[k], g0[k], rtol=1e-3, atol=1e-5 * scale)


def test_normalised_directions_frozen_coefficients_and_inference(hip_lib):
    N, W, H = 120_000, 400, 240
    sc, vm, K, Gc, Ga, cam, dev = _scene(N, W, H, seed=5)
    # MTGS's own order: directions normalised in PyTorch, then spherical_harmonics
    r1, _, g1, c1, _ = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "raster", normalise=True)
    r0, _, g0, _, _ = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "torch", normalise=True)
    assert torch.equal(r1, r0) and "mtgs_vis_color_fwd_dirs" in c1
    torch.testing.assert_close(g1["coeffs"], g0["coeffs"], rtol=1e-3, atol=1e-5 * float(g0["coeffs"].abs().max()))
    # frozen coefficients: no colour backward at all, the geometry still trains
    r1, _, g1, c1, _ = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "raster", frozen=("coeffs",))
    r0, _, g0, _, _ = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "torch", frozen=("coeffs",))
    assert torch.equal(r1, r0) and g1["coeffs"] is None and "mtgs_vis_color_bwd_dirs" not in c1 and "mtgs_vis_color_fwd_dirs" in c1
    torch.testing.assert_close(g1["means"], g0["means"], rtol=1e-3, atol=1e-5 * float(g0["means"].abs().max()))
    # only the coefficients train
    r1, _, g1, c1, _ = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "raster", frozen=("means", "quats", "scales", "opacities"))
    r0, _, g0, _, _ = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "torch", frozen=("means", "quats", "scales", "opacities"))
    assert torch.equal(r1, r0) and g1["means"] is None
    torch.testing.assert_close(g1["coeffs"], g0["coeffs"], rtol=1e-3, atol=1e-5 * float(g0["coeffs"].abs().max()))
    # inference
    r1, a1, _, c1, _ = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "raster", no_grad=True)
    r0, a0, _, _, _ = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "torch", no_grad=True)
    assert torch.equal(r1, r0) and torch.equal(a1, a0) and "mtgs_vis_color_fwd_dirs" in c1 and not r1.requires_grad


def test_what_the_rasterization_cannot_take_over_is_evaluated_in_full(hip_lib):
    """Several cameras, a background, channels in FRONT of the colours, the colours used a second time: the
    deferred object turns into the fused kernel's ordinary tensor -- same results as PyTorch's.  (Channels BEHIND the colours --
    MTGS's predict_normals -- are taken over with them.)"""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    from mtgs_amd.synthetic import make_camera
    N, W, H = 60_000, 320, 200
    sc, vm, K, Gc, Ga, cam, dev = _scene(N, W, H, seed=7)
    vm2 = torch.cat([vm, make_camera(W, H, yaw_deg=20.0)[0].to(dev)])
    K2 = torch.cat([K, K])
    bg = torch.tensor([[0.1, 0.2, 0.3]], device=dev)
    extra = torch.rand(N, 2, device=dev)

    def run(lazy, case):
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
        calls = []
        real = wrapper.call
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            with wrapper.sh_lazy(lazy):
                dirs = P["means"] - cam if case == "dirs_grad" else P["means"].detach() - cam
                rgb = torch.clamp(spherical_harmonics(3, dirs, P["coeffs"]) + 0.5, 0.0, 1.0)
                kw = dict(packed=False, render_mode="RGB", rasterize_mode="antialiased")
                reg = 0.0
                if case == "two_cameras":
                    out = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm2, K2, W, H, **kw)
                elif case == "background":
                    out = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, backgrounds=bg, **kw)
                elif case == "extra_channels":
                    out = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], torch.cat([rgb, extra], dim=-1), vm, K, W, H, **kw)
                elif case == "extra_in_front":
                    out = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], torch.cat([extra, rgb], dim=-1), vm, K, W, H, **kw)
                elif case == "used_twice":
                    out = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, **kw)
                    reg = (rgb * rgb).sum() * 1e-3
                else:
                    out = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, **kw)
                (out[0].sum() + reg).backward()
        finally:
            wrapper.call = real
        return out[0].detach(), {k: v.grad.clone() for k, v in P.items()}, calls

    for case in ("two_cameras", "background", "extra_channels", "extra_in_front", "used_twice"):
        r1, g1, c1 = run(True, case)
        r0, g0, c0 = run(False, case)
        assert torch.equal(r1, r0), case
        if case == "used_twice":      # the rasterization took the colours over; the second use evaluated them in full once more
            assert "mtgs_vis_color_fwd_dirs" in c1 and "mtgs_sh_fwd_act" in c1, c1
        elif case == "extra_channels":      # channels BEHIND the colours (MTGS's normals) stay deferred with them
            assert "mtgs_vis_color_fwd_dirs" in c1 and "mtgs_sh_fwd_act" not in c1, c1
        else:
            assert "mtgs_vis_color_fwd_dirs" not in c1 and "mtgs_sh_fwd_act" in c1, (case, c1)
        for k in g0:
            torch.testing.assert_close(g1[k], g0[k], rtol=2e-3, atol=2e-5 * float(g0[k].abs().max()), msg=lambda m: f"{case} {k}: {m}")


def test_deferred_colours_in_one_captured_graph(hip_lib):
    """The headline step -- spherical_harmonics, clamp, rasterization, backward -- captured in ONE HIP graph (mtgs_amd.graph_mode: fixed
    capacities, counts on the device) with the colours deferred into the rasterization: replays give the eager step's render bit for bit and
    its gradients; the dense coefficient gradient is the zeros that rode on the compositing forward plus the touched rows."""
    from mtgs_amd import graph_mode, rasterization, spherical_harmonics, wrapper
    N, W, H = 200_000, 640, 368
    sc, vm, K, Gc, Ga, cam, dev = _scene(N, W, H, seed=9)
    r0, a0, g0, _, i0 = _step(sc, vm, K, W, H, Gc, Ga, cam, dev, "torch")
    n_vis, M = int((i0["radii"] > 0).sum()), int(i0["flatten_ids"].numel())
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    dirs = (P["means"].detach() - cam).contiguous()
    calls = []
    real = wrapper.call

    def body():
        rgb = torch.clamp(spherical_harmonics(3, dirs, P["coeffs"]) + 0.5, 0.0, 1.0)
        render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False,
                                            render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
        ((render * Gc).sum() + (alpha * Ga).sum()).backward()
        return render, alpha, info

    gm = graph_mode(int(n_vis * 1.2) + 1024, int(M * 1.2) + 4096)
    with gm:
        body()                                  # under the mode once: its staging buffers exist before the capture
    for p in P.values():
        p.grad = None
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    try:
        wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        with gm, torch.cuda.graph(graph):
            render, alpha, info = body()
    finally:
        wrapper.call = real
    assert "mtgs_vis_color_fwd_dirs" in calls and "mtgs_vis_color_bwd_dirs" in calls and "mtgs_rows_expand" not in calls, calls
    assert not [n for n in calls if n.startswith("mtgs_sh_")], calls
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert not bool(info["overflow"]) and int(info["n_visible"]) == n_vis
    assert torch.equal(render, r0) and torch.equal(alpha, a0)
    for k in P:
        torch.testing.assert_close(P[k].grad, g0[k], rtol=1e-3, atol=1e-5 * float(g0[k].abs().max()))
    assert torch.equal(P["coeffs"].grad != 0, g0["coeffs"] != 0)


def test_the_nodes_of_a_scene_graph_concatenated_stay_deferred(hip_lib):
    """MTGS evaluates get_rgbs() per node and concatenates (mtgs_scene_graph.py:440-452: `torch.cat(value, dim=0)`, for a single node
    too): torch.cat(dim 0) of deferred activations of one degree stays deferred, rasterization() evaluates the visible Gaussians of ALL
    nodes in one launch (one descriptor per node) and every node's coefficient tensor receives ITS slice of the dense gradient.  Nodes
    of different degrees (or a plain tensor among them) are evaluated in full and concatenated by PyTorch."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    N, W, H = 150_000, 512, 288
    sc, vm, K, Gc, Ga, cam, dev = _scene(N, W, H, seed=13)
    cuts = [0, 70_000, 70_001, 120_000, N]          # four nodes, one of a single Gaussian

    def run(mode, degrees=(3, 3, 3, 3), plain_part=False):
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items() if k != "coeffs"}
        Cs = [sc["coeffs"][a:b].to(dev).clone().requires_grad_(True) for a, b in zip(cuts[:-1], cuts[1:])]
        calls = []
        real = wrapper.call
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            with wrapper.sh_lazy(mode != "torch", raster=mode == "raster"):
                parts = []
                for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                    dirs = P["means"][a:b].detach() - cam
                    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
                    rgbs = torch.clamp(spherical_harmonics(degrees[i], dirs, Cs[i]) + 0.5, 0.0, 1.0)
                    parts.append(rgbs * 1.0 if (plain_part and i == 1) else rgbs)
                rgb = torch.cat(parts, dim=0)
                deferred = type(rgb).__name__ == "_LazySH"
                render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False,
                                                    render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
                ((render * Gc).sum() + (alpha * Ga).sum()).backward()
        finally:
            wrapper.call = real
        return render.detach(), [c.grad.clone() for c in Cs], {k: v.grad.clone() for k, v in P.items()}, calls, deferred

    r0, c0, g0, _, d0 = run("torch")
    r1, c1, g1, calls, d1 = run("raster")
    assert d1 and not d0 and calls.count("mtgs_vis_color_fwd_dirs") == 1 and calls.count("mtgs_vis_color_bwd_dirs") == 1
    assert not [n for n in calls if n.startswith("mtgs_sh_")] and "mtgs_rows_expand" not in calls, calls
    assert torch.equal(r1, r0)
    for a, b in zip(c1, c0):
        assert a.shape == b.shape and torch.equal(a != 0, b != 0)
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-5 * float(max(b.abs().max(), 1e-20)))
    assert sum(int((c != 0).any()) for c in c0) >= 3      # (several nodes are seen by the camera)
    for k in g0:
        torch.testing.assert_close(g1[k], g0[k], rtol=1e-3, atol=1e-5 * float(g0[k].abs().max()))
    # a single node, concatenated alone (what MTGS does for a scene without objects)
    cuts_all, cuts[:] = list(cuts), [0, N]
    try:
        r2, c2, _, calls2, d2 = run("raster", degrees=(3,))
        r3, c3, _, _, _ = run("torch", degrees=(3,))
    finally:
        cuts[:] = cuts_all
    assert d2 and "mtgs_vis_color_fwd_dirs" in calls2 and torch.equal(r2, r3)
    torch.testing.assert_close(c2[0], c3[0], rtol=1e-3, atol=1e-5 * float(c3[0].abs().max()))
    # different degrees / a plain tensor among the parts: evaluated in full per node (fused kernel), concatenated by PyTorch
    for kw in (dict(degrees=(3, 2, 3, 3)), dict(plain_part=True)):
        r4, c4, _, calls4, d4 = run("raster", **kw)
        r5, c5, _, _, _ = run("torch", **kw)
        assert not d4 and "mtgs_vis_color_fwd_dirs" not in calls4 and calls4.count("mtgs_sh_fwd_act") == 4, (kw, calls4)
        assert torch.equal(r4, r5)
        for a, b in zip(c4, c5):
            torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-5 * float(max(b.abs().max(), 1e-20)))


def test_concatenated_nodes_with_a_frozen_node_and_in_inference(hip_lib):
    """One node's coefficients frozen (requires_grad False: MTGS freezes nodes it does not optimise): that node gets no gradient, the
    others theirs; the whole concatenation without grad mode renders the same image."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    N, W, H = 90_000, 400, 240
    sc, vm, K, Gc, Ga, cam, dev = _scene(N, W, H, seed=19)
    cuts = [0, 30_000, 65_000, N]

    def run(mode, frozen=1, no_grad=False):
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items() if k != "coeffs"}
        Cs = [sc["coeffs"][a:b].to(dev).clone().requires_grad_(i != frozen) for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
        calls = []
        real = wrapper.call
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            with wrapper.sh_lazy(mode != "torch", raster=mode == "raster"), (torch.no_grad() if no_grad else torch.enable_grad()):
                rgb = torch.cat([torch.clamp(spherical_harmonics(3, P["means"][a:b].detach() - cam, Cs[i]) + 0.5, 0.0, 1.0)
                                 for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))], dim=0)
                render, alpha, _ = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False,
                                                 render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
                if not no_grad:
                    ((render * Gc).sum() + (alpha * Ga).sum()).backward()
        finally:
            wrapper.call = real
        return render.detach(), [c.grad for c in Cs], P["means"].grad, calls

    r1, c1, m1, calls = run("raster")
    r0, c0, m0, _ = run("torch")
    assert "mtgs_vis_color_fwd_dirs" in calls and torch.equal(r1, r0)
    assert c1[1] is None and c0[1] is None
    for i in (0, 2):
        assert torch.equal(c1[i] != 0, c0[i] != 0) and float(c0[i].abs().sum()) > 0
        torch.testing.assert_close(c1[i], c0[i], rtol=1e-3, atol=1e-5 * float(c0[i].abs().max()))
    torch.testing.assert_close(m1, m0, rtol=1e-3, atol=1e-5 * float(m0.abs().max()))
    r2, _, _, calls2 = run("raster", no_grad=True)
    assert torch.equal(r2, r0) and "mtgs_vis_color_fwd_dirs" in calls2 and not r2.requires_grad
    # every node frozen, the geometry trains: no colour backward at all
    r3, c3, m3, calls3 = run("raster", frozen=None)      # (reference run with all nodes training, for the geometry gradient)

    def run_all_frozen():
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items() if k != "coeffs"}
        calls_ = []
        real = wrapper.call
        try:
            wrapper.call = lambda name, *a: (calls_.append(name), real(name, *a))[1]
            rgb = torch.cat([torch.clamp(spherical_harmonics(3, P["means"][a:b].detach() - cam, sc["coeffs"][a:b].to(dev)) + 0.5, 0.0, 1.0)
                             for a, b in zip(cuts[:-1], cuts[1:])], dim=0)
            render, alpha, _ = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False,
                                             render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
            ((render * Gc).sum() + (alpha * Ga).sum()).backward()
        finally:
            wrapper.call = real
        return render.detach(), P["means"].grad, calls_

    r4, m4, calls4 = run_all_frozen()
    assert torch.equal(r4, r0) and "mtgs_vis_color_fwd_dirs" in calls4 and "mtgs_vis_color_bwd_dirs" not in calls4
    torch.testing.assert_close(m4, m3, rtol=1e-3, atol=1e-5 * float(m3.abs().max()))


@pytest.mark.parametrize("render_mode,dx", [("RGB+ED", 3), ("RGB", 3), ("RGB+ED", 1), ("RGB", 5)])
def test_channels_concatenated_behind_the_colours_stay_deferred(hip_lib, render_mode, dx):
    """config/MTGS.py renders `torch.cat([rgbs, normals], dim=-1)` (predict_normals, mtgs_scene_graph.py:636-638; RGB+ED: seven
    blended channels): the colours stay deferred through that concatenation too -- the rasterization evaluates them for the visible
    Gaussians and takes the further channels as they are; same render bit for bit, the further channels receive their dense gradient."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    N, W, H = 150_000, 512, 288
    sc, vm, K, Gc, Ga, cam, dev = _scene(N, W, H, seed=23)
    g = torch.Generator().manual_seed(5)
    extra0 = torch.randn(N, dx, generator=g).to(dev)
    cuts = [0, 90_000, N]

    def run(mode):
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items() if k != "coeffs"}
        Cs = [sc["coeffs"][a:b].to(dev).clone().requires_grad_(True) for a, b in zip(cuts[:-1], cuts[1:])]
        extra = extra0.clone().requires_grad_(True)
        calls = []
        real = wrapper.call
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            with wrapper.sh_lazy(mode != "torch", raster=mode == "raster"):
                rgbs = torch.cat([torch.clamp(spherical_harmonics(3, P["means"][a:b].detach() - cam, Cs[i]) + 0.5, 0.0, 1.0)
                                  for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))], dim=0)
                colors = torch.cat([rgbs, extra * 0.5], dim=-1)      # (the further channels come out of the caller's own autograd graph)
                deferred = type(colors).__name__ == "_LazySH"
                render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], colors, vm, K, W, H, packed=False,
                                                    render_mode=render_mode, absgrad=True, rasterize_mode="antialiased")
                info["means2d"].retain_grad()
                Gr = torch.randn(render.shape, generator=torch.Generator().manual_seed(9)).to(dev)
                ((render * Gr).sum() + (alpha * Ga).sum()).backward()
        finally:
            wrapper.call = real
        grads = {k: v.grad.clone() for k, v in P.items()}
        grads.update(extra=extra.grad.clone(), c0=Cs[0].grad.clone(), c1=Cs[1].grad.clone(), m2d=info["means2d"].grad.clone())
        return render.detach(), alpha.detach(), grads, calls, deferred

    r1, a1, g1, calls, d1 = run("raster")
    r0, a0, g0, _, d0 = run("torch")
    assert d1 and not d0 and "mtgs_vis_color_fwd_dirs" in calls and not [n for n in calls if n.startswith("mtgs_sh_")], calls
    assert r1.shape[-1] == 3 + dx + int(render_mode != "RGB") and torch.equal(r1, r0) and torch.equal(a1, a0)
    for k in g0:
        assert torch.equal(g1[k] != 0, g0[k] != 0), k
        torch.testing.assert_close(g1[k], g0[k], rtol=1e-3, atol=1e-5 * float(g0[k].abs().max()), msg=lambda m: f"{k}: {m}")
    assert float(g0["extra"].abs().sum()) > 0 and float(g0["c1"].abs().sum()) > 0


def test_directions_with_a_gradient_stay_deferred(hip_lib):
    """config/MTGS.py trains with a camera optimizer (SO3xR3): `viewdirs = means.detach() - camera_to_worlds[..., :3, 3]`, normalised in
    PyTorch (vanilla_gaussian_splatting.py:313-314), carries a gradient.  The deferred colours still go into the rasterization: its
    backward returns d L / d viewdirs of the visible Gaussians (through the kernel's own normalisation) scattered into a dense [N, 3]
    tensor per node, and the camera position receives what PyTorch's composition gives it."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    N, W, H = 150_000, 512, 288
    sc, vm, K, Gc, Ga, cam0, dev = _scene(N, W, H, seed=29)
    cuts = [0, 80_000, N]

    def run(mode, frozen_coeffs=False):
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items() if k != "coeffs"}
        Cs = [sc["coeffs"][a:b].to(dev).clone().requires_grad_(not frozen_coeffs) for a, b in zip(cuts[:-1], cuts[1:])]
        campos = cam0.clone().requires_grad_(True)      # (the optimised camera's position)
        calls = []
        real = wrapper.call
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            with wrapper.sh_lazy(mode != "torch", raster=mode == "raster"):
                parts = []
                for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                    viewdirs = P["means"][a:b].detach() - campos
                    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
                    parts.append(torch.clamp(spherical_harmonics(3, viewdirs, Cs[i]) + 0.5, 0.0, 1.0))
                rgb = torch.cat(parts, dim=0)
                deferred = type(rgb).__name__ == "_LazySH"
                render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False,
                                                    render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
                ((render * Gc).sum() + (alpha * Ga).sum()).backward()
        finally:
            wrapper.call = real
        return render.detach(), campos.grad.clone(), [c.grad for c in Cs], P["means"].grad.clone(), calls, deferred

    r1, cam1, c1, m1, calls, d1 = run("raster")
    r0, cam0g, c0, m0, _, _ = run("torch")
    assert d1 and "mtgs_vis_color_fwd_dirs" in calls and "mtgs_vis_color_bwd_dirs" in calls and not [n for n in calls if n.startswith("mtgs_sh_")]
    assert torch.equal(r1, r0)
    assert float(cam0g.abs().max()) > 0
    torch.testing.assert_close(cam1, cam0g, rtol=2e-3, atol=2e-5 * float(cam0g.abs().max()))
    for a, b in zip(c1, c0):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-5 * float(b.abs().max()))
    torch.testing.assert_close(m1, m0, rtol=1e-3, atol=1e-5 * float(m0.abs().max()))
    # frozen coefficients, the camera still trains
    r2, cam2, c2, _, calls2, _ = run("raster", frozen_coeffs=True)
    r3, cam3, _, _, _, _ = run("torch", frozen_coeffs=True)
    assert torch.equal(r2, r3) and c2[0] is None and "mtgs_rows_expand" not in calls2
    torch.testing.assert_close(cam2, cam3, rtol=2e-3, atol=2e-5 * float(cam3.abs().max()))


@pytest.mark.parametrize("degree", [0, 1, 2])
def test_lower_degrees_of_the_sh_schedule(hip_lib, degree):
    """MTGS raises the SH degree with the step count (n = min(step // sh_degree_interval, sh_degree), vanilla_gaussian_splatting.py:315)
    on coefficient tensors that hold all 16 rows from the start: degrees 0 .. 2 on K = 16 through the rasterization, bit for bit."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    N, W, H = 100_000, 400, 240
    sc, vm, K, Gc, Ga, cam, dev = _scene(N, W, H, seed=31)

    def run(mode):
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
        calls = []
        real = wrapper.call
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            with wrapper.sh_lazy(mode != "torch", raster=mode == "raster"):
                rgb = torch.clamp(spherical_harmonics(degree, P["means"].detach() - cam, P["coeffs"]) + 0.5, 0.0, 1.0)
                render, alpha, _ = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False,
                                                 render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
                ((render * Gc).sum() + (alpha * Ga).sum()).backward()
        finally:
            wrapper.call = real
        return render.detach(), P["coeffs"].grad.clone(), calls

    r1, c1, calls = run("raster")
    r0, c0, _ = run("torch")
    assert "mtgs_vis_color_fwd_dirs" in calls and torch.equal(r1, r0)
    nb = (degree + 1) ** 2
    assert float(c1[:, nb:].abs().max()) == 0.0 and float(c0[:, nb:].abs().max()) == 0.0      # (the bands above the degree in use)
    assert torch.equal(c1 != 0, c0 != 0) and float(c0.abs().sum()) > 0
    torch.testing.assert_close(c1, c0, rtol=1e-3, atol=1e-5 * float(c0.abs().max()))


def test_the_colour_path_of_config_mtgs_py_end_to_end(hip_lib):
    """Everything config/MTGS.py does between its parameters and rasterization(), at once: per-node get_rgbs() on
    `torch.cat((features_dc[:, None, :], features_rest), dim=1)` with view directions from the OPTIMISED camera, normalised in PyTorch
    (vanilla_gaussian_splatting.py:309-322), the scene graph's `torch.cat(value, dim=0)` of every collected tensor
    (mtgs_scene_graph.py:451-452), camera-space normals concatenated behind the colours (:636-638), RGB+ED / antialiased / absgrad
    (:641-659), `retain_grad()` on means2d (:666-668).  Deferred all the way against PyTorch's evaluation: the same render bit for bit,
    the same gradients of every leaf (features_dc, features_rest, geometry, camera position)."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    N, W, H = 160_000, 640, 368
    sc, vm, K, _, Ga, cam0, dev = _scene(N, W, H, seed=37)
    cuts = [0, 100_000, 100_300, N]      # a background node, a small object node, another node
    Gr = torch.randn(1, H, W, 7, generator=torch.Generator().manual_seed(6)).to(dev)

    def run(mode):
        leaves = {}
        nodes = []
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            nd = {"means": sc["means"][a:b].to(dev).clone().requires_grad_(True), "quats": sc["quats"][a:b].to(dev).clone().requires_grad_(True),
                  "scales": sc["scales"][a:b].to(dev).clone().requires_grad_(True), "opacities": sc["opacities"][a:b].to(dev).clone().requires_grad_(True),
                  "features_dc": sc["coeffs"][a:b, 0].to(dev).clone().requires_grad_(True),
                  "features_rest": sc["coeffs"][a:b, 1:].to(dev).clone().requires_grad_(True)}
            nodes.append(nd)
            leaves.update({f"{k}{i}": v for k, v in nd.items()})
        campos = cam0.clone().requires_grad_(True)
        leaves["campos"] = campos
        calls = []
        real = wrapper.call
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            with wrapper.sh_lazy(mode != "torch", raster=mode == "raster"):
                gs = {"means": [], "quats": [], "scales": [], "opacities": [], "rgbs": []}
                for nd in nodes:
                    colors = torch.cat((nd["features_dc"][:, None, :], nd["features_rest"]), dim=1)
                    viewdirs = nd["means"].detach() - campos
                    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
                    rgbs = torch.clamp(spherical_harmonics(3, viewdirs, colors) + 0.5, 0.0, 1.0)
                    for k in ("means", "quats", "scales", "opacities"):
                        gs[k].append(nd[k])
                    gs["rgbs"].append(rgbs)
                col = {k: torch.cat(v, dim=0) for k, v in gs.items()}
                normals = torch.nn.functional.normalize(col["quats"][:, 1:] * col["scales"], dim=1)      # (a stand-in with the real one's graph shape)
                render_colors = torch.cat([col["rgbs"], normals], dim=-1)
                deferred = type(render_colors).__name__ == "_LazySH"
                render, alpha, info = rasterization(means=col["means"], quats=col["quats"], scales=col["scales"], opacities=col["opacities"],
                                                    colors=render_colors, viewmats=vm, Ks=K, width=W, height=H, tile_size=16, packed=False,
                                                    near_plane=0.01, far_plane=1e10, render_mode="RGB+ED", sparse_grad=False, absgrad=True,
                                                    rasterize_mode="antialiased")
                info["means2d"].retain_grad()
                ((render * Gr).sum() + (alpha * Ga).sum()).backward()
        finally:
            wrapper.call = real
        grads = {k: v.grad.clone() for k, v in leaves.items()}
        grads["absgrad"] = info["means2d"].absgrad.clone()
        return render.detach(), grads, calls, deferred

    r1, g1, calls, d1 = run("raster")
    r0, g0, _, d0 = run("torch")
    assert d1 and not d0 and calls.count("mtgs_vis_color_fwd_dirs") == 1 and not [n for n in calls if n.startswith("mtgs_sh_")], calls
    assert r1.shape[-1] == 7 and torch.equal(r1, r0)
    for k in g0:
        scale = float(g0[k].abs().max())
        assert scale > 0 or k.startswith(("features", "means", "quats", "scales", "opacities")), k
        torch.testing.assert_close(g1[k], g0[k], rtol=2e-3, atol=2e-5 * max(scale, 1e-20), msg=lambda m: f"{k}: {m}")
