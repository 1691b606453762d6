"""The dense gradients of rasterization() written WITHOUT a streaming pass (mtgs_amd/wrapper.py::_zeroed_outputs_plan,
csrc/project_bwd.hip::mtgs_project_bwd_zeroed): v_means / v_quats / v_scales / v_opacities, means2d.grad, .absgrad and the gradient of
extra colour channels are views of ONE region that the compositing backward clears beside its own work
(mtgs_blend_bwd_packed(also_zero)); the projection backward writes the rows of the Gaussians that have a gradient to their places.
Must equal the streaming form (rounds 1-5: project_bwd_expand_kernel) -- same non-zero pattern, same values up to the order of the
compositing atomics -- in every combination of by-products the MTGS call produces (mtgs_scene_graph.py:641-668, 1170-1178)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(zeroed, colours, retain, absgrad, render_mode, N=120_000, W=480, H=272, prefill=True):
    from mtgs_amd import rasterization, spherical_harmonics, wrapper
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda")
    sc = make_scene(N, seed=21, sh_degree=3)
    vm, K = make_camera(W, H, yaw_deg=5.0)
    vm, K = vm.to(dev).requires_grad_(True), K.to(dev)
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    cam = torch.inverse(vm.detach())[0, :3, 3]
    g = torch.Generator().manual_seed(4)
    was, was_p = wrapper._zeroed_outputs, wrapper._prefill.enabled
    wrapper._zeroed_outputs, wrapper._prefill.enabled = zeroed, prefill
    calls = []
    real = wrapper.call
    try:
        wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        if colours == "sh":          # MTGS: deferred SH colours (no dense colour gradient)
            cols = torch.clamp(spherical_harmonics(3, P["means"].detach() - cam, P["coeffs"]) + 0.5, 0.0, 1.0)
        else:                         # colours given per Gaussian (+ extra channels): a dense colour gradient is a by-product
            extra = torch.rand(N, int(colours), generator=g).to(dev).requires_grad_(True)
            P["extra"] = extra
            cols = extra
        render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], cols, vm, K, W, H, packed=False,
                                            render_mode=render_mode, absgrad=absgrad, rasterize_mode="antialiased")
        if retain:
            info["means2d"].retain_grad()
        Gc = torch.randn(render.shape, generator=g).to(dev)
        Ga = torch.randn(alpha.shape, generator=g).to(dev)
        ((render * Gc).sum() + (alpha * Ga).sum()).backward()
    finally:
        wrapper.call = real
        wrapper._zeroed_outputs, wrapper._prefill.enabled = was, was_p
    out = {k: v.grad.clone() for k, v in P.items() if v.grad is not None}
    out["viewmat"] = vm.grad.clone()
    if retain:
        out["m2d"] = info["means2d"].grad.clone()
    if absgrad:
        out["abs"] = info["means2d"].absgrad.clone()
    return out, calls, int((info["radii"] > 0).sum())


@pytest.mark.parametrize("colours,retain,absgrad,render_mode", [
    ("sh", True, True, "RGB+ED"),        # config/MTGS.py
    ("sh", False, False, "RGB"),
    ("3", True, True, "RGB+ED"),         # colours given: + a dense colour gradient
    ("7", True, False, "RGB"),           # the shipped 7-channel cell (colours + normals + ...)
    ("3", False, True, "RGB+D"),
])
def test_zeroed_outputs_equal_the_streaming_expansion(hip_lib, colours, retain, absgrad, render_mode):
    got, calls, n_vis = _run(True, colours, retain, absgrad, render_mode)
    want, calls0, _ = _run(False, colours, retain, absgrad, render_mode)
    assert "mtgs_project_bwd_zeroed" in calls and "mtgs_project_bwd" not in calls, calls
    assert "mtgs_project_bwd" in calls0 and "mtgs_project_bwd_zeroed" not in calls0
    assert set(got) == set(want)
    for k in want:
        assert got[k].shape == want[k].shape and got[k].is_contiguous(), k
        assert torch.equal(got[k] != 0, want[k] != 0), k
        torch.testing.assert_close(got[k], want[k], rtol=1e-3, atol=1e-5 * float(want[k].abs().max()), msg=lambda m: f"{k}: {m}")
    # most Gaussians have no gradient at all: their rows are the zeros the compositing backward wrote
    touched = int((got["means"] != 0).any(dim=1).sum())
    assert 0 < touched < n_vis < got["means"].shape[0] // 2


def test_without_the_prefill_switch_the_streaming_pass_runs(hip_lib):
    got, calls, _ = _run(True, "sh", True, True, "RGB+ED", prefill=False)
    assert "mtgs_project_bwd" in calls and "mtgs_project_bwd_zeroed" not in calls
    assert float(got["means"].abs().sum()) > 0


def test_a_loss_on_info_tensors_takes_the_generic_path(hip_lib):
    """gradients reaching info["means2d"] / ["depths"] directly (not MTGS, but gsplat's contract): the streaming form, same as before"""
    from mtgs_amd import rasterization, wrapper
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda")
    N, W, H = 50_000, 320, 200
    sc = make_scene(N, seed=2, sh_degree=None)
    vm, K = make_camera(W, H)
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    calls = []
    real = wrapper.call
    try:
        wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm.to(dev), K.to(dev), W, H,
                                            packed=False, render_mode="RGB")
        (render.sum() + info["depths"].sum() * 1e-3).backward()
    finally:
        wrapper.call = real
    assert "mtgs_project_bwd" in calls and "mtgs_project_bwd_zeroed" not in calls
    assert torch.isfinite(P["means"].grad).all() and float(P["means"].grad.abs().sum()) > 0
