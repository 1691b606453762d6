"""The SH backward's zeros written by the rasterizer's compositing kernels (mtgs_amd/wrapper.py::_Prefill,
csrc/blend.hip::ZeroFill -- mtgs_blend_{fwd,bwd}_packed(also_zero) --, csrc/sh.hip::sh_bwd_rows_kernel): dL/dcoeffs of `spherical_harmonics() -> clamp(. + 0.5) -> rasterization()` -- the call sequence of
/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:309-318 + mtgs_scene_graph.py:641-662 -- must be the
dense one-kernel backward's, bit for bit, whether the cotangent is sparse (only composited Gaussians) or dense (an extra loss on the
colours), with masks, inside a HIP graph capture, and when the request is never served (no rasterizer in the graph)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _sh_evaluated_by_its_own_kernels():
    """These tests are about the spherical_harmonics() autograd node and the zeros of ITS backward.  MTGS's plain step no longer has
    such a node -- the colours stay deferred into the rasterization (tests/test_gpu_sh_raster.py) -- but every composition the
    rasterization cannot take over still has (masks, several nodes concatenated, K != 16, a second use of the colours ...): the
    deferral into the rasterization is switched off here so that the plain step exercises it."""
    from mtgs_amd import wrapper
    with wrapper.sh_lazy(wrapper._lazy_sh_enabled, raster=False):
        yield


def _scene(dev, N=300_000, W=640, H=368):
    from mtgs_amd.synthetic import make_camera, make_scene
    sc = make_scene(N, seed=3, sh_degree=3)
    vm, K = make_camera(W, H)
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    g = torch.Generator().manual_seed(2)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
    return P, vm.to(dev), K.to(dev), Gc, Ga, (W, H)


def _step(P, vm, K, Gc, Ga, WH, degree=3, dense_extra=False, masks=None):
    from mtgs_amd import rasterization, spherical_harmonics
    for p in P.values():
        p.grad = None
    dirs = P["means"].detach() - torch.inverse(vm)[0, :3, 3]
    sh = spherical_harmonics(degree, dirs, P["coeffs"], masks=masks)
    rgb = torch.clamp(sh + 0.5, 0.0, 1.0)
    render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, WH[0], WH[1], packed=False,
                                        render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
    loss = (render * Gc).sum() + (alpha * Ga).sum()
    if dense_extra:
        loss = loss + 1e-3 * (sh * sh).sum()        # a cotangent on EVERY Gaussian's colour
    loss.backward()
    return {k: v.grad.clone() for k, v in P.items()}


def _step_mode(dense_extra):
    """dense_extra uses the RAW SH output a second time (`sh * sh`): with the deferred spherical_harmonics() (wrapper._LazySH) that is a
    second, plain evaluation beside the fused one -- correct, but not the single-node graph these tests reason about: they pin the
    eager evaluation for it (tests/test_gpu_sh_lazy.py covers the deferred one)."""
    from mtgs_amd import wrapper
    return wrapper.sh_lazy(not dense_extra)


@pytest.mark.parametrize("degree", [0, 1, 2, 3])
@pytest.mark.parametrize("dense_extra", [False, True])
def test_zeros_from_the_compositing_kernels_plus_rows_equal_the_dense_backward(hip_lib, degree, dense_extra):
    from mtgs_amd import wrapper
    dev = torch.device("cuda")
    args = _scene(dev)
    assert wrapper._prefill.enabled
    calls = []
    real = wrapper.call
    try:
        wrapper.call = lambda name, *a: (calls.append(name.replace("_act", "")), real(name, *a))[1]      # (the fused-activation forms count as their plain ones)
        with _step_mode(dense_extra):
            got = _step(*args, degree=degree, dense_extra=dense_extra)
    finally:
        wrapper.call = real
    assert "mtgs_sh_bwd_rows" in calls and "mtgs_sh_bwd" not in calls and "mtgs_fill_zero" not in calls, calls
    assert calls.index("mtgs_blend_bwd_packed") < calls.index("mtgs_sh_bwd_rows")
    wrapper._prefill.enabled = False
    try:
        with _step_mode(dense_extra):
            want = _step(*args, degree=degree, dense_extra=dense_extra)
    finally:
        wrapper._prefill.enabled = True
    nz = int((want["coeffs"].abs().sum(dim=(1, 2)) > 0).sum())
    assert (nz > 0.9 * want["coeffs"].shape[0]) if dense_extra else (0 < nz < 0.3 * want["coeffs"].shape[0]), nz
    # (two runs sum the compositing backward's fp32 atomics in another order: the colour cotangents differ in their last bits.  The
    #  kernels themselves are compared bit for bit on ONE cotangent below)
    assert torch.equal(got["coeffs"] != 0, want["coeffs"] != 0)
    for k in ("coeffs", "means", "quats", "scales", "opacities"):
        torch.testing.assert_close(got[k], want[k], rtol=1e-3, atol=1e-5 * float(want[k].abs().max()))


@pytest.mark.parametrize("K,degree", [(16, 3), (16, 1), (9, 2), (25, 4), (4, 0), (1, 0)])
@pytest.mark.parametrize("density", [0.0, 0.05, 1.0])
def test_rows_kernel_on_zeros_is_the_dense_kernel(hip_lib, K, degree, density):
    from mtgs_amd._lib import call, ptr, stream_of
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(K * 10 + degree)
    n = 100_003
    dirs = torch.randn(n, 3, generator=g).to(dev)
    coeffs = torch.randn(n, K, 3, generator=g).to(dev)
    v = torch.randn(n, 3, generator=g) * (torch.rand(n, 1, generator=g) < density)
    v = v.to(dev)
    masks = (torch.rand(n, generator=g) < 0.8).to(dev).to(torch.uint8)
    for m in (None, masks):
        dense = torch.full((n, K, 3), float("nan"), device=dev)
        call("mtgs_sh_bwd", n, K, degree, ptr(dirs), ptr(coeffs), ptr(m), ptr(v), ptr(dense), None, stream_of(dirs))
        rows = torch.full((n, K, 3), float("nan"), device=dev)
        call("mtgs_fill_zero", rows.data_ptr(), rows.numel() * 4, stream_of(dirs))
        call("mtgs_sh_bwd_rows", n, K, degree, ptr(dirs), ptr(m), ptr(v), ptr(rows), stream_of(dirs))
        assert torch.equal(dense, rows)
        assert density == 0.0 or float(rows.abs().sum()) > 0


def test_masks_and_a_request_nobody_serves(hip_lib):
    from mtgs_amd import spherical_harmonics, wrapper
    dev = torch.device("cuda")
    P, vm, K, Gc, Ga, WH = _scene(dev)
    masks = (torch.arange(P["means"].shape[0], device=dev) % 3) != 0
    got = _step(P, vm, K, Gc, Ga, WH, masks=masks)
    wrapper._prefill.enabled = False
    try:
        want = _step(P, vm, K, Gc, Ga, WH, masks=masks)
    finally:
        wrapper._prefill.enabled = True
    torch.testing.assert_close(got["coeffs"], want["coeffs"], rtol=1e-3, atol=1e-5 * float(want["coeffs"].abs().max()))
    assert float(got["coeffs"][~masks].abs().max()) == 0.0 and float(got["coeffs"][masks].abs().max()) > 0
    # spherical_harmonics() alone: the request stays pending, the backward is the dense kernel, and nothing is left behind
    dirs = torch.randn(P["means"].shape, device=dev)
    out = spherical_harmonics(3, dirs, P["coeffs"])
    P["coeffs"].grad = None
    out.square().sum().backward()
    ref = P["coeffs"].grad.clone()
    assert float(ref.abs().sum()) > 0
    del out
    import gc
    gc.collect()
    assert len(wrapper._prefill.pending) == 0


def test_inside_a_graph_capture(hip_lib):
    """One stream, no extra node kinds: the replayed step's gradients are the eager step's."""
    import mtgs_amd
    dev = torch.device("cuda")
    P, vm, K, Gc, Ga, WH = _scene(dev, N=200_000)
    eager = _step(P, vm, K, Gc, Ga, WH)
    n_vis = 60_000
    gm = mtgs_amd.graph_mode(n_vis, 1_500_000)
    with gm:
        _step(P, vm, K, Gc, Ga, WH)
    for p in P.values():
        p.grad = None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    from mtgs_amd import rasterization, spherical_harmonics
    box = {}
    cam_pos = torch.inverse(vm)[0, :3, 3]

    def body():
        dirs = P["means"].detach() - cam_pos
        rgb = torch.clamp(spherical_harmonics(3, dirs, P["coeffs"]) + 0.5, 0.0, 1.0)
        render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, WH[0], WH[1], packed=False,
                                            render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
        torch.autograd.backward([render, alpha], [Gc, Ga])
        box["overflow"] = info["overflow"]

    with gm, torch.cuda.graph(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert not bool(box["overflow"])
    torch.testing.assert_close(P["coeffs"].grad, eager["coeffs"], rtol=1e-3, atol=1e-5 * float(eager["coeffs"].abs().max()))
    torch.testing.assert_close(P["means"].grad, eager["means"], rtol=1e-3, atol=1e-5 * float(eager["means"].abs().max()))


@pytest.mark.parametrize("in_forward", [True, False])
def test_served_by_the_forward_or_by_the_backward_and_backward_twice(hip_lib, in_forward):
    """The region is cleared by the compositing FORWARD (default) or, for requests that only turn up later, by the compositing
    BACKWARD; a second backward through the same graph (retain_graph) finds its rows used and takes the plain paths."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    dev = torch.device("cuda")
    P, vm, K, Gc, Ga, WH = _scene(dev, N=150_000)
    want = _step(P, vm, K, Gc, Ga, WH)
    was = wrapper._prefill.in_forward
    wrapper._prefill.in_forward = in_forward
    calls = []
    real = wrapper.call
    try:
        wrapper.call = lambda name, *a: (calls.append((name.replace("_act", ""), a)), real(name, *a))[1]
        for p in P.values():
            p.grad = None
        dirs = P["means"].detach() - torch.inverse(vm)[0, :3, 3]
        rgb = torch.clamp(spherical_harmonics(3, dirs, P["coeffs"]) + 0.5, 0.0, 1.0)
        render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, WH[0], WH[1], packed=False,
                                            render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
        loss = (render * Gc).sum() + (alpha * Ga).sum()
        loss.backward(retain_graph=True)
        first = {k: v.grad.clone() for k, v in P.items()}
        for p in P.values():
            p.grad = None
        loss.backward()
        second = {k: v.grad.clone() for k, v in P.items()}
    finally:
        wrapper.call, wrapper._prefill.in_forward = real, was
    fwd_zero = [a[-2] for n, a in calls if n == "mtgs_blend_fwd_packed"]
    bwd_zero = [a[-2] for n, a in calls if n == "mtgs_blend_bwd_packed"]
    n_coeff_bytes = P["coeffs"].numel() * 4
    # (the compositing backward also clears the DENSE gradients this node returns -- tests/test_gpu_zeroed_outputs.py --: D bytes in
    #  both backwards; the SH request's bytes come on top of them in the first one when the forward did not serve it)
    D = bwd_zero[1]
    assert 0 < D < n_coeff_bytes, (fwd_zero, bwd_zero)
    assert (fwd_zero[0] >= n_coeff_bytes and bwd_zero == [D, D]) if in_forward else (fwd_zero == [0] and bwd_zero[0] == n_coeff_bytes + D), (fwd_zero, bwd_zero)
    names = [n for n, _ in calls]
    assert names.count("mtgs_sh_bwd_rows") == 1 and names.count("mtgs_sh_bwd") == 1      # (second backward: the dense kernel)
    for got in (first, second):
        for k in got:
            torch.testing.assert_close(got[k], want[k], rtol=1e-3, atol=1e-5 * float(want[k].abs().max()))


def test_two_nodes_one_rasterization_and_a_backward_on_another_stream(hip_lib):
    """A scene graph evaluates spherical_harmonics() once per node and rasterizes the concatenation (mtgs_scene_graph.py:408-461 +
    641-662): both requests are served by the one rasterization, in one region.  A spherical_harmonics() whose backward runs on
    ANOTHER stream than the rasterization that cleared its buffer must not take it (nothing orders the two): it falls back."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    dev = torch.device("cuda")
    P, vm, K, Gc, Ga, WH = _scene(dev, N=200_000)
    cam = torch.inverse(vm)[0, :3, 3]
    half = 120_000

    def run():
        for p in P.values():
            p.grad = None
        cA = P["coeffs"][:half].detach().clone().requires_grad_(True)
        cB = P["coeffs"][half:].detach().clone().requires_grad_(True)
        dirs = P["means"].detach() - cam
        shA, shB = spherical_harmonics(3, dirs[:half], cA), spherical_harmonics(2, dirs[half:], cB)
        rgb = torch.clamp(torch.cat([shA, shB]) + 0.5, 0.0, 1.0)
        render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, WH[0], WH[1], packed=False,
                                            render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
        torch.autograd.backward([render, alpha], [Gc, Ga])
        return cA.grad.clone(), cB.grad.clone()

    calls = []
    real = wrapper.call
    try:
        wrapper.call = lambda name, *a: (calls.append((name.replace("_act", ""), a)), real(name, *a))[1]
        gA, gB = run()
    finally:
        wrapper.call = real
    names = [n for n, _ in calls]
    assert names.count("mtgs_sh_bwd_rows") == 2 and "mtgs_sh_bwd" not in names
    zeroed = [a[-2] for n, a in calls if n == "mtgs_blend_fwd_packed"][0]
    assert zeroed >= (half + (200_000 - half)) * 16 * 3 * 4
    wrapper._prefill.enabled = False
    try:
        wA, wB = run()
    finally:
        wrapper._prefill.enabled = True
    for got, want in ((gA, wA), (gB, wB)):
        assert torch.equal(got != 0, want != 0)
        torch.testing.assert_close(got, want, rtol=1e-3, atol=1e-5 * float(want.abs().max()))
    assert float(gB[:, 9:].abs().max()) == 0.0      # (degree 2 of K = 16: the rows kernel leaves the higher bands zero)
    # ... a request served on one stream, its backward on another: the dense kernel
    side = torch.cuda.Stream()
    c = P["coeffs"].detach().clone().requires_grad_(True)
    sh = spherical_harmonics(3, P["means"].detach() - cam, c)
    rgb = torch.clamp(sh + 0.5, 0.0, 1.0)
    render, alpha, info = rasterization(P["means"].detach(), P["quats"].detach(), P["scales"].detach(), P["opacities"].detach(), rgb, vm, K,
                                        WH[0], WH[1], packed=False, render_mode="RGB+ED", rasterize_mode="antialiased")
    req = wrapper._Prefill.behind(rgb)[0]            # (the request on the SH node behind these colours -- fused activation or not)
    assert req is not None and req.buffer is not None
    req.stream = side.cuda_stream                     # (as if the rasterization had run over there)
    calls.clear()
    try:
        wrapper.call = lambda name, *a: (calls.append((name.replace("_act", ""), a)), real(name, *a))[1]
        torch.autograd.backward([render, alpha], [Gc, Ga])
    finally:
        wrapper.call = real
    names = [n for n, _ in calls]
    assert "mtgs_sh_bwd" in names and "mtgs_sh_bwd_rows" not in names
    assert torch.isfinite(c.grad).all() and float(c.grad.abs().sum()) > 0


def test_gsplats_sh_degree_call_style_writes_the_coefficient_gradient_in_place(hip_lib):
    """rasterization(colors=coeffs[N,16,3], sh_degree=3): the dense coefficient gradient is a buffer the compositing forward cleared,
    the backward writes the rows with a cotangent straight into it (mtgs_vis_color_bwd(dense_rows)) -- no [n_vis, 48] intermediate, no
    dense expansion pass -- and equals the expansion path."""
    from mtgs_amd import rasterization, wrapper
    dev = torch.device("cuda")
    P, vm, K, Gc, Ga, WH = _scene(dev, N=150_000)

    def run():
        for p in P.values():
            p.grad = None
        render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["coeffs"], vm, K, WH[0], WH[1],
                                            sh_degree=3, packed=False, render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
        torch.autograd.backward([render, alpha], [Gc, Ga])
        return {k: v.grad.clone() for k, v in P.items()}

    calls = []
    real = wrapper.call
    try:
        wrapper.call = lambda name, *a: (calls.append(name.replace("_act", "")), real(name, *a))[1]      # (the fused-activation forms count as their plain ones)
        got = run()
    finally:
        wrapper.call = real
    assert "mtgs_vis_color_bwd_dirs" in calls and "mtgs_rows_expand" not in calls, calls      # (the _dirs entry points, dirs = NULL here)
    wrapper._prefill.enabled = False
    try:
        want = run()
    finally:
        wrapper._prefill.enabled = True
    assert torch.equal(got["coeffs"] != 0, want["coeffs"] != 0) and float(want["coeffs"].abs().sum()) > 0
    for k in got:
        torch.testing.assert_close(got[k], want[k], rtol=1e-3, atol=1e-5 * float(want[k].abs().max()))


def test_a_thumbnail_frame_clears_the_region_with_the_fill_kernel(hip_lib):
    """Two tiles under a 58 MB coefficient gradient: mtgs_blend_fwd_packed hands the region to the fill kernel instead of making two
    waves write it (blend.hip::zero_fill_fallback); same gradients."""
    from mtgs_amd import wrapper
    dev = torch.device("cuda")
    P, vm, K, Gc, Ga, _ = _scene(dev, N=300_000)
    WH = (32, 16)
    g = torch.Generator().manual_seed(5)
    Gc, Ga = torch.randn(1, 16, 32, 4, generator=g).to(dev), torch.randn(1, 16, 32, 1, generator=g).to(dev)
    vm2, K2 = vm.clone(), K.clone()
    K2[0, 0, 0] = K2[0, 1, 1] = 25.0; K2[0, 0, 2] = 16.0; K2[0, 1, 2] = 8.0
    got = _step(P, vm2, K2, Gc, Ga, WH)
    wrapper._prefill.enabled = False
    try:
        want = _step(P, vm2, K2, Gc, Ga, WH)
    finally:
        wrapper._prefill.enabled = True
    assert float(want["coeffs"].abs().sum()) > 0 and torch.equal(got["coeffs"] != 0, want["coeffs"] != 0)
    for k in got:
        torch.testing.assert_close(got[k], want[k], rtol=2e-3, atol=1e-5 * float(want[k].abs().max()))


def test_a_second_models_rasterization_does_not_serve_the_first_models_request(hip_lib):
    """SCOPE of the prefill (round 6): two models on one device.  Model A's spherical_harmonics() forward leaves a request; model B's
    rasterization -- whose colours do NOT descend from A's SH output -- must neither allocate nor zero A's dL/dcoeffs (its region holds
    its own gradient rows only), and A's request is still served by A's own rasterization afterwards; a rasterization on ANOTHER stream
    does not serve it either; an unserved request expires at the next SH forward of the same shape."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    dev = torch.device("cuda")
    PA, vm, K, Gc, Ga, (W, H) = _scene(dev, N=200_000)
    PB, *_ = _scene(dev, N=200_000)
    pf = wrapper._prefill
    cam = torch.inverse(vm)[0, :3, 3]

    def raster(P, rgb):
        return rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False, render_mode="RGB+ED",
                             absgrad=True, rasterize_mode="antialiased")

    sh_a = spherical_harmonics(3, PA["means"].detach() - cam, PA["coeffs"])
    rgb_a = torch.clamp(sh_a + 0.5, 0.0, 1.0)          # (with the deferred SH this IS the spherical_harmonics() node: one kernel)
    (req_a,) = wrapper._Prefill.behind(rgb_a[None])
    assert req_a is not None and req_a in pf.pending
    # model B: plain colours that need a gradient (no SH behind them)
    col_b = torch.rand(200_000, 3, device=dev, requires_grad=True)
    assert wrapper._Prefill.behind(col_b) == []
    n0 = pf.regions
    rb, ab, info_b = raster(PB, col_b)
    n_vis_b = int((info_b["radii"] > 0).sum())
    assert pf.regions == n0 + 1 and req_a.buffer is None and req_a in pf.pending           # B's region: its own gradient rows only
    assert pf.last_region_bytes <= 4 * 16 * int(1.6 * n_vis_b + 4096), (pf.last_region_bytes, n_vis_b)
    # the same on another stream, with colours that DO descend from A's SH output: not served (the zeros would not be ordered)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        raster(PA, rgb_a)
    torch.cuda.current_stream().wait_stream(side)
    assert req_a.buffer is None and req_a in pf.pending
    # A's own rasterization on A's stream serves it
    ra, aa, _ = raster(PA, rgb_a)
    assert req_a.buffer is not None and req_a not in pf.pending and req_a.buffer.shape == PA["coeffs"].shape
    assert pf.last_region_bytes >= PA["coeffs"].numel() * 4
    ((ra * Gc).sum() + (aa * Ga).sum() + (rb * Gc).sum()).backward()
    got = PA["coeffs"].grad.clone()
    with wrapper.sh_prefill(enabled=False):
        for p in PA.values():
            p.grad = None
        rgb2 = torch.clamp(spherical_harmonics(3, PA["means"].detach() - cam, PA["coeffs"]) + 0.5, 0.0, 1.0)
        assert wrapper._Prefill.behind(rgb2) == []
        r2, a2, _ = raster(PA, rgb2)
        ((r2 * Gc).sum() + (a2 * Ga).sum()).backward()
    assert pf.enabled
    assert torch.equal(got != 0, PA["coeffs"].grad != 0)
    torch.testing.assert_close(got, PA["coeffs"].grad, rtol=1e-3, atol=1e-5 * float(got.abs().max()))
    # expiry: a request nobody served is dropped by the next forward of the same shape
    s1 = torch.clamp(spherical_harmonics(3, PA["means"].detach() - cam, PA["coeffs"]) + 0.5, 0.0, 1.0)
    (r1,) = wrapper._Prefill.behind(s1)
    s2 = torch.clamp(spherical_harmonics(3, PA["means"].detach() - cam, PA["coeffs"]) + 0.5, 0.0, 1.0)
    assert r1 not in pf.pending and wrapper._Prefill.behind(s2)[0] in pf.pending
    (s1.sum() + s2.sum()).backward()            # both take the dense backward
    assert bool(torch.isfinite(PA["coeffs"].grad).all())
