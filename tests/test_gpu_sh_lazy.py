"""The caller's colour activation fused into spherical_harmonics() WITHOUT touching the caller (mtgs_amd/wrapper.py::_LazySH,
csrc/sh.hip::ShAct): `torch.clamp(spherical_harmonics(n, dirs, coeffs) + 0.5, 0.0, 1.0)` -- the line behind every SH call of MTGS
(/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:317-318, multi_color_gaussian_splatting.py:96,
rigid_node.py:248, deformable_node.py:125) -- and gsplat's `clamp_min(colors + 0.5, 0.0)` must give the values and gradients of the
three separate PyTorch operations BIT FOR BIT, and every other use of the function's result must behave like an ordinary tensor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(n, K, seed, dev, scale=0.6):
    g = torch.Generator().manual_seed(seed)
    dirs = torch.randn(n, 3, generator=g).to(dev)
    coeffs = (torch.randn(n, K, 3, generator=g) * scale).to(dev)
    v = torch.randn(n, 3, generator=g).to(dev)
    return dirs, coeffs, v


FORMS = {
    "mtgs": lambda x: torch.clamp(x + 0.5, 0.0, 1.0),
    "gsplat": lambda x: torch.clamp_min(x + 0.5, 0.0),
    "kwargs": lambda x: torch.clamp(x + 0.5, min=0.0, max=1.0),
    "method": lambda x: (x + 0.5).clamp(0.0, 1.0),
    "radd_clip": lambda x: torch.clip(0.25 + x, -0.1, 0.9),
    "no_add": lambda x: x.clamp(-0.2, 0.3),
    "clamp_max": lambda x: torch.clamp_max(x + 0.5, 0.75),
    "int_bounds": lambda x: torch.clamp(x + 1, 0, 1),
}


@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("K,degree", [(16, 3), (16, 1), (9, 2), (25, 4), (4, 0)])
def test_fused_activation_is_the_three_torch_operations_bit_for_bit(hip_lib, form, K, degree):
    from mtgs_amd import spherical_harmonics, wrapper
    dev = torch.device("cuda")
    dirs, coeffs, v = _inputs(50_003, K, K * 7 + degree, dev)
    masks = (torch.rand(dirs.shape[0], device=dev) < 0.9)
    for m in (None, masks):
        calls = []
        real = wrapper.call
        c1 = coeffs.clone().requires_grad_(True)
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            x = spherical_harmonics(degree, dirs, c1, masks=m)
            assert type(x).__name__ == "_LazySH" and not calls and x.shape == dirs.shape and x.requires_grad and x.device == dirs.device
            y = FORMS[form](x)
            if form in ("mtgs", "gsplat", "kwargs", "method") and K == 16 and m is None:
                # the activations rasterization() can evaluate by itself stay deferred through the clamp (tests/test_gpu_sh_raster.py);
                # any other consumer gets the fused kernel's result, here
                assert type(y).__name__ == "_LazySH" and not calls and y.requires_grad and y.shape == dirs.shape
                y = y.contiguous()
            assert type(y) is torch.Tensor and calls == ["mtgs_sh_fwd_act"], calls
            (y * v).sum().backward()
        finally:
            wrapper.call = real
        assert calls == ["mtgs_sh_fwd_act", "mtgs_sh_bwd_act"], calls
        with wrapper.sh_lazy(False):
            c2 = coeffs.clone().requires_grad_(True)
            x2 = spherical_harmonics(degree, dirs, c2, masks=m)
            assert type(x2) is torch.Tensor
            y2 = FORMS[form](x2)
            (y2 * v).sum().backward()
        assert torch.equal(y, y2) and torch.equal(c1.grad, c2.grad), (form, K, degree, m is not None)
        if form == "mtgs" and degree > 0:      # (the clamp binds on a real share of the channels: the mask is exercised)
            frac = float(((y2 == 0.0) | (y2 == 1.0)).float().mean())
            assert 0.02 < frac < 0.98, frac


def test_bounds_are_inclusive_and_nan_propagates_like_torch(hip_lib):
    from mtgs_amd import spherical_harmonics, wrapper
    dev = torch.device("cuda")
    dirs, coeffs, v = _inputs(4096, 16, 3, dev)
    coeffs[:1000] = 0.0                       # SH = 0 exactly: x + 0.5 == 0.5 == lo -> the bound itself passes its cotangent (torch: >=)
    coeffs[1000:1010, 0, 0] = float("nan")
    for lo, hi in ((0.5, 1.0), (0.0, 0.5)):
        outs = []
        for lazy in (True, False):
            with wrapper.sh_lazy(lazy):
                c = coeffs.clone().requires_grad_(True)
                y = torch.clamp(spherical_harmonics(3, dirs, c) + 0.5, lo, hi)
                (y * v).sum().backward()
                outs.append((y, c.grad))
        (y1, g1), (y2, g2) = outs
        assert torch.equal(torch.isnan(y1), torch.isnan(y2)) and bool(torch.isnan(y1[1000:1010, 0]).all())
        assert torch.equal(torch.nan_to_num(y1, nan=7.0), torch.nan_to_num(y2, nan=7.0))
        assert torch.equal(torch.nan_to_num(g1, nan=7.0), torch.nan_to_num(g2, nan=7.0))
        assert float(g1[:1000, 0].abs().min()) > 0      # the rows AT the bound received their gradient


def test_any_other_use_is_an_ordinary_tensor(hip_lib):
    from mtgs_amd import spherical_harmonics, wrapper
    dev = torch.device("cuda")
    dirs, coeffs, v = _inputs(10_000, 16, 11, dev)
    with wrapper.sh_lazy(False):
        ref = spherical_harmonics(3, dirs, coeffs)

    def fresh():
        return spherical_harmonics(3, dirs, coeffs.clone().requires_grad_(True))

    assert torch.equal(fresh() * 2.0, ref * 2.0)
    assert torch.equal(fresh()[10:20], ref[10:20]) and torch.equal(fresh()[None].squeeze(0), ref)
    assert torch.equal(torch.cat([fresh(), fresh()]), torch.cat([ref, ref]))
    assert torch.equal(fresh() + v, ref + v)                         # tensor addend: not the deferred scalar add
    assert torch.equal(torch.add(fresh(), 0.5, alpha=2), ref + 1.0)  # alpha: not the deferred add either
    assert torch.equal(fresh().detach(), ref) and float(fresh().sum()) == float(ref.sum())
    assert np.array_equal(fresh().detach().cpu().numpy(), ref.cpu().numpy())
    assert torch.equal(torch.clamp(fresh() + 0.5, torch.zeros_like(ref), torch.ones_like(ref)), torch.clamp(ref + 0.5, 0.0, 1.0))   # tensor bounds
    x = fresh()
    assert x.grad_fn is not None and type(x.grad_fn).__name__ == "_SphericalHarmonicsBackward"
    x.retain_grad()
    (x * v).sum().backward()
    assert torch.equal(x.grad, v)
    # one object, used twice: the deferred add and the plain value agree, the SH kernel of the plain use runs once
    calls = []
    real = wrapper.call
    try:
        wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        x = fresh()
        a = x + 0.5
        s = x.sum()
        b = a * 1.0
    finally:
        wrapper.call = real
    assert calls == ["mtgs_sh_fwd"] and torch.equal(b, ref + 0.5) and float(s) == float(ref.sum())
    # in-place on the deferred object
    x = spherical_harmonics(3, dirs, coeffs)
    x.add_(0.5)
    assert torch.equal(x * 1.0, ref + 0.5)
    # no grad mode, no requires_grad: still fused, still equal
    with torch.no_grad():
        y = torch.clamp(spherical_harmonics(3, dirs, coeffs) + 0.5, 0.0, 1.0)
    assert not y.requires_grad and torch.equal(y, torch.clamp(ref + 0.5, 0.0, 1.0))


def test_the_training_step_with_the_deferred_sh_equals_the_plain_one(hip_lib):
    """spherical_harmonics -> clamp(+0.5) -> rasterization -> backward, MTGS's composition: the rasterizer sees bit-identical colours, the
    zeros of dL/dcoeffs ride on its compositing forward as before (the request sits on the fused node), the rows kernel applies the
    clamp mask; gradients equal the plain path's up to the order of the compositing atomics."""
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda")
    N, W, H = 300_000, 640, 368
    sc = make_scene(N, seed=3, sh_degree=3)
    vm, K = make_camera(W, H)
    vm, K = vm.to(dev), K.to(dev)
    g = torch.Generator().manual_seed(2)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
    cam = torch.inverse(vm)[0, :3, 3]

    def step(lazy):
        P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
        calls = []
        real = wrapper.call
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            with wrapper.sh_lazy(lazy, raster=False):
                rgb = torch.clamp(spherical_harmonics(3, P["means"].detach() - cam, P["coeffs"]) + 0.5, 0.0, 1.0)
                render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm, K, W, H, packed=False,
                                                    render_mode="RGB+ED", absgrad=True, rasterize_mode="antialiased")
                ((render * Gc).sum() + (alpha * Ga).sum()).backward()
        finally:
            wrapper.call = real
        return rgb.detach(), render.detach(), {k: v.grad.clone() for k, v in P.items()}, calls

    rgb1, r1, g1, c1 = step(True)
    rgb0, r0, g0, c0 = step(False)
    assert torch.equal(rgb1, rgb0) and torch.equal(r1, r0)
    assert "mtgs_sh_fwd_act" in c1 and "mtgs_sh_bwd_rows_act" in c1 and "mtgs_sh_bwd" not in c1 and "mtgs_sh_bwd_act" not in c1
    assert "mtgs_sh_fwd" in c0 and "mtgs_sh_bwd_rows" in c0
    assert torch.equal(g1["coeffs"] != 0, g0["coeffs"] != 0)
    for k in g0:
        torch.testing.assert_close(g1[k], g0[k], rtol=1e-3, atol=1e-5 * float(g0[k].abs().max()))
