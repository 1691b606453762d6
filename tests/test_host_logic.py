"""Host-side logic that does not need a GPU: the gsplat-compatible surface, loud failure without a
device, the synthetic generator, and the gradient bucket."""
import hashlib

import numpy as np
import pytest
import torch


def _args(N=4):
    z = lambda *s: torch.zeros(*s)
    return (z(N, 3), z(N, 4), z(N, 3), z(N), z(N, 3), torch.eye(4)[None], torch.eye(3)[None], 32, 32)


def test_shim_exposes_the_imports_mtgs_uses():
    import gsplat
    from gsplat.cuda._wrapper import spherical_harmonics
    from gsplat.rendering import rasterization
    import mtgs_amd
    assert gsplat.__version__ == "1.4.0"
    assert rasterization is mtgs_amd.rasterization and spherical_harmonics is mtgs_amd.spherical_harmonics
    import inspect
    sig = inspect.signature(rasterization)
    # the kwargs MTGS passes (mtgs_scene_graph.py:641-659) and gsplat's defaults
    for name in ("means", "quats", "scales", "opacities", "colors", "viewmats", "Ks", "width", "height", "tile_size",
                 "packed", "near_plane", "far_plane", "render_mode", "sparse_grad", "absgrad", "rasterize_mode"):
        assert name in sig.parameters
    d = {k: v.default for k, v in sig.parameters.items()}
    assert d["near_plane"] == 0.01 and d["far_plane"] == 1e10 and d["eps2d"] == 0.3 and d["radius_clip"] == 0.0
    assert d["packed"] is True and d["tile_size"] == 16 and d["render_mode"] == "RGB" and d["channel_chunk"] == 32


@pytest.mark.parametrize("kw,name", [
    (dict(packed=True), "packed"), (dict(packed=False, sparse_grad=True), "sparse_grad"),
    (dict(packed=False, distributed=True), "distributed"), (dict(packed=False, camera_model="ortho"), "camera_model"),
    (dict(packed=False, tile_size=32), "tile_size"), (dict(packed=False, covars=torch.zeros(4, 3, 3)), "covars")])
def test_unsupported_options_raise_by_name(kw, name):
    from mtgs_amd import rasterization
    with pytest.raises(NotImplementedError, match=name):
        rasterization(*_args(), **kw)


def test_no_cpu_fallback():
    """CPU tensors must fail loudly -- the product path never computes on the host."""
    from mtgs_amd import rasterization, spherical_harmonics
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rasterization(*_args(), packed=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        spherical_harmonics(0, torch.zeros(4, 3), torch.zeros(4, 1, 3))


def test_product_never_imports_the_oracle():
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1]
    for p in list((root / "mtgs_amd").rglob("*.py")) + list((root / "gsplat").rglob("*.py")):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, p


def test_shape_checks_mirror_gsplat():
    from mtgs_amd import rasterization, spherical_harmonics
    a = list(_args())
    a[3] = torch.zeros(5)                       # opacities of the wrong length
    with pytest.raises(AssertionError):
        rasterization(*a, packed=False)
    with pytest.raises(AssertionError):          # (deg+1)^2 > K
        spherical_harmonics(3, torch.zeros(4, 3), torch.zeros(4, 9, 3))


def test_wbv1_generator_is_deterministic():
    from mtgs_amd.synthetic import make_camera, make_scene
    a, b = make_scene(1000, seed=0, sh_degree=3), make_scene(1000, seed=0, sh_degree=3)
    for k in a:
        assert torch.equal(a[k], b[k])
    assert a["coeffs"].shape == (1000, 16, 3) and a["quats"].norm(dim=-1).sub(1).abs().max() < 1e-6
    assert a["scales"].min() >= 0.02 - 1e-6 and a["scales"].max() <= 0.2 + 1e-6
    vm, K = make_camera(1920, 1080, yaw_deg=45.0)
    assert vm.shape == (1, 4, 4) and K[0, 0, 0] == 0.8 * 1920 and K[0, 0, 2] == 960
    np.testing.assert_allclose((vm[0, :3, :3] @ vm[0, :3, :3].T).numpy(), np.eye(3), atol=1e-6)


def test_mtgs_c2w_to_viewmat_is_the_inverse_with_flipped_axes():
    from mtgs_amd.synthetic import mtgs_c2w_to_viewmat
    g = torch.Generator().manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(4, generator=g), dim=0)
    from oracle.torch_ref import quat_to_rotmat
    c2w = torch.eye(4)
    c2w[:3, :3] = quat_to_rotmat(q[None])[0]
    c2w[:3, 3] = torch.tensor([1.0, 2.0, 3.0])
    vm = mtgs_c2w_to_viewmat(c2w)[0]
    flip = torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0]))
    np.testing.assert_allclose((vm @ c2w @ flip).numpy(), np.eye(4), atol=1e-5)


def test_flat_grad_bucket_single_process():
    from mtgs_amd.dist import FlatGradBucket
    a = torch.randn(5, 3, requires_grad=True)
    b = torch.randn(7, requires_grad=True)
    bucket = FlatGradBucket([a, b])
    ((a * 2).sum() + (b * 3).sum()).backward()
    assert a.grad.data_ptr() == bucket.views[0].data_ptr()
    assert torch.all(bucket.flat[:15] == 2) and torch.all(bucket.flat[64:71] == 3)
    bucket.zero()
    assert bucket.flat.abs().max() == 0 and a.grad.data_ptr() == bucket.views[0].data_ptr()
    assert bucket.all_reduce() is None          # no process group: no-op


def test_neighbour_modules_have_no_cpu_fallback_and_validate_arguments():
    """mtgs_amd.nodes / loss / densify: CPU tensors raise, shapes are checked before any launch."""
    from mtgs_amd.densify import update_statistics
    from mtgs_amd.loss import masked_ssim
    from mtgs_amd.nodes import node_gaussians
    N = 8
    z = lambda *s: torch.zeros(*s)
    c2w = torch.eye(4)[None, :3]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        node_gaussians(z(N, 3), z(N, 3), torch.ones(N, 4), z(N, 1), z(N, 3), z(N, 15, 3), c2w, 3, 3)
    with pytest.raises(AssertionError):
        node_gaussians(z(N, 3), z(N, 3), torch.ones(N, 4), z(N, 1), z(N, 3), z(N, 3, 3), c2w, 3, 3)   # degree 3 needs K = 16
    with pytest.raises(NotImplementedError, match="degree"):
        node_gaussians(z(N, 3), z(N, 3), torch.ones(N, 4), z(N, 1), z(N, 3), z(N, 24, 3), c2w, 4, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        masked_ssim(z(32, 32, 3), z(32, 32, 3))
    with pytest.raises(ValueError, match="11x11"):
        masked_ssim(z(8, 32, 3), z(8, 32, 3))
    with pytest.raises(NotImplementedError, match="gt"):
        masked_ssim(z(32, 32, 3).requires_grad_(True), z(32, 32, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        update_statistics(z(N), z(N), z(N), torch.zeros(1, N, dtype=torch.int32), z(1, N, 2), 64, 48)


def test_later_neighbours_have_no_cpu_fallback_and_validate_arguments():
    """collect_gaussians (batched), camera_space_normals, output_head, masked_l1, oob_loss, update_statistics_all: CPU tensors
    raise, shapes are checked before any launch."""
    from mtgs_amd.densify import update_statistics_all
    from mtgs_amd.loss import masked_l1, oob_loss, output_head
    from mtgs_amd.nodes import camera_space_normals, collect_gaussians
    N = 8
    z = lambda *s: torch.zeros(*s)
    c2w = torch.eye(4)[None, :3]
    node = {"means": z(N, 3), "scales": z(N, 3), "quats": torch.ones(N, 4), "opacities": z(N, 1), "features_dc": z(N, 3),
            "features_rest": z(N, 15, 3)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        collect_gaussians([node, dict(node, instance_quats=torch.ones(5, 4), instance_trans=z(5, 3), frame_idx=2)], c2w, 3, 3)
    with pytest.raises(AssertionError):     # frame index outside the pose table
        collect_gaussians([dict(node, instance_quats=torch.ones(5, 4), instance_trans=z(5, 3), frame_idx=5)], c2w, 3, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        camera_space_normals(torch.ones(N, 4), torch.ones(N, 3), z(N, 3), c2w, rgbs=z(N, 3))
    with pytest.raises(AssertionError):
        camera_space_normals(torch.ones(N, 4), torch.ones(N, 2), z(N, 3), c2w)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        output_head(z(1, 16, 16, 8), z(1, 16, 16, 1), z(3), torch.eye(3, 4), depth=True, normal_channel=3)
    with pytest.raises(AssertionError):     # the normal channels would overlap the depth channel
        output_head(z(1, 16, 16, 6), z(1, 16, 16, 1), z(3), None, depth=True, normal_channel=3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        masked_l1(z(16, 16, 1), z(16, 16, 1), torch.ones(16, 16, 1, dtype=torch.bool))
    with pytest.raises(AssertionError):
        masked_l1(z(16, 16, 9), z(16, 16, 9))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        oob_loss([(z(N, 3), z(N, 1), [4.0, 2.0, 1.5])], torch.zeros(1, 20, dtype=torch.int32), [3])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        update_statistics_all([(z(N), z(N), z(N))], torch.zeros(1, N, dtype=torch.int32), z(1, N, 2), 64, 48)


def test_masked_ssim_takes_three_channels_only():
    from mtgs_amd.loss import masked_ssim
    with pytest.raises(AssertionError):
        masked_ssim(torch.zeros(32, 32, 1), torch.zeros(32, 32, 1))


def test_collect_gaussians_rejects_empty_and_wrong_dtype():
    from mtgs_amd.nodes import collect_gaussians
    c2w = torch.eye(4)[None, :3]
    with pytest.raises(ValueError, match="no nodes"):
        collect_gaussians([], c2w, 3, 3)
    N = 4
    node = {"means": torch.zeros(N, 3, dtype=torch.float64), "scales": torch.zeros(N, 3), "quats": torch.ones(N, 4),
            "opacities": torch.zeros(N, 1), "features_dc": torch.zeros(N, 3), "features_rest": torch.zeros(N, 15, 3)}
    with pytest.raises(TypeError, match="float32"):
        collect_gaussians([node], c2w, 3, 3)


def test_round3_host_logic_without_a_gpu():
    """FusedAdam refuses CPU parameters and the options it does not implement before anything touches the library; the
    descriptor record has the documented fields; the exchange exposes its phase; a ColorSource cannot hand rows over before a
    backward."""
    import pytest
    import torch
    from mtgs_amd import dist as mdist, nodes, optim
    p = torch.zeros(8, requires_grad=True)
    p.grad = torch.zeros(8)
    with pytest.raises(RuntimeError):
        optim.FusedAdam([p]).step()
    with pytest.raises(NotImplementedError):
        optim.FusedAdam([p], amsgrad=True)
    with pytest.raises(ValueError):
        optim.FusedAdam([p], lr=-1.0)
    o = optim.FusedAdam([p])
    with pytest.raises(ValueError):      # rows must be float32 [R, stride] with an int32 map of one entry per Gaussian
        o.set_row_gradient(p, torch.zeros(3, 4), torch.zeros(7, dtype=torch.int32), 0)
    assert {"p", "m", "v", "g", "rows", "row_of", "catchup", "last", "hist", "sub_width", "sub_index", "mode", "catchup_k",
            "hyper_index"} <= set(optim._GROUP.names)
    assert (optim.MODE_DENSE, optim.MODE_SLICE, optim.MODE_ROWS_CATCHUP, optim.MODE_ROWS_STEP, optim.MODE_ROWS_FLUSH) == (0, 1, 2, 3, 4)
    with pytest.raises(RuntimeError):    # row-lazy parameters live on the GPU
        o.set_row_lazy(p)
    cs0 = nodes.ColorSource(None, 1, 3, None, [], [])
    assert cs0.prepare(torch.zeros(4, dtype=torch.int32), 4) is None      # no optimizer attached: the parameters are read in place
    m, v, _ = None, None, None
    ref = optim.adam_reference_step(torch.ones(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64),
                                    torch.full((3,), 0.5, dtype=torch.float64), 1, 1e-2, eps=1e-15)
    assert torch.allclose(ref[0], torch.full((3,), 1.0 - 1e-2, dtype=torch.float64))     # first Adam step = lr * sign(g)
    ex = mdist.SparseGradExchange(1000, 16, "cpu")
    assert ex.phase == "idle" and ex.world_collectives is False
    cs = nodes.ColorSource(None, 1, 3, None, [], [])
    with pytest.raises(AssertionError):
        cs.apply_to(o)


def test_deferred_sh_dispatch_rules_on_cpu(monkeypatch):
    """mtgs_amd.wrapper._LazySH's decisions without a GPU: which expressions stay deferred, which take the fused kernel (with which
    activation), which materialise -- the SH autograd function replaced by a torch stand-in (a linear map of the coefficients)."""
    import torch
    from mtgs_amd import wrapper
    seen = []

    class Fake:
        @staticmethod
        def apply(degree, dirs, coeffs, masks, act=None):
            seen.append(act)
            x = coeffs[..., 0, :] * 2.0 + dirs
            if act is None:
                return x
            has_add, add, lo, hi = act
            return torch.clamp(x + add if has_add else x, lo, hi)

    monkeypatch.setattr(wrapper, "_SphericalHarmonics", Fake)
    d, c = torch.randn(7, 3), torch.randn(7, 16, 3, requires_grad=True)
    ref = c[..., 0, :] * 2.0 + d
    new = lambda: wrapper._LazySH(3, d, c, None)
    x = new()
    assert (x.shape, x.dtype, x.dim(), x.numel(), x.requires_grad, x.device.type) == (d.shape, torch.float32, 2, 21, True, "cpu") and not seen
    # MTGS's and gsplat's activations on K = 16 rows stay deferred THROUGH the clamp (rasterization() may take them over: visible
    # Gaussians only); any other use runs the fused kernel with that activation, once
    y = torch.clamp(x + 0.5, 0.0, 1.0)
    assert not seen and type(y) is wrapper._LazySH and y._lz_act == (0.0, 1.0) and y.shape == d.shape and y.requires_grad
    assert y.raster_source(8, 64, 64) is None      # (not this tensor's Gaussian count: the caller falls back to the full evaluation)
    assert torch.equal(y, torch.clamp(ref + 0.5, 0.0, 1.0)) and torch.equal(y * 2, torch.clamp(ref + 0.5, 0.0, 1.0) * 2)
    assert seen == [(True, 0.5, 0.0, 1.0)] and y.raster_source(7, 64, 64) is None      # (already evaluated)
    seen.clear()
    y = torch.clamp_min(0.5 + new(), 0.0)
    assert not seen and type(y) is wrapper._LazySH
    assert torch.equal(y, torch.clamp_min(ref + 0.5, 0.0)) and seen == [(True, 0.5, 0.0, float("inf"))]
    seen.clear()
    assert type(torch.clamp(y, 0.0, 0.5)) is torch.Tensor and not seen      # (a second clamp: ordinary tensors, no second evaluation)
    with wrapper.sh_lazy(True, raster=False):      # ... switched off: the clamp runs the fused kernel over all Gaussians at once
        y = torch.clamp(new() + 0.5, 0.0, 1.0)
        assert seen == [(True, 0.5, 0.0, 1.0)] and type(y) is torch.Tensor and torch.equal(y, torch.clamp(ref + 0.5, 0.0, 1.0))
    seen.clear()
    for expr, want, act in ((lambda z: torch.clamp(z + 0.25, 0.0, 1.0), torch.clamp(ref + 0.25, 0.0, 1.0), (True, 0.25, 0.0, 1.0)),
                            (lambda z: torch.clamp(z + 0.5, 0.0, 2.0), torch.clamp(ref + 0.5, 0.0, 2.0), (True, 0.5, 0.0, 2.0)),
                            (lambda z: torch.clamp(z, 0.0, 1.0), torch.clamp(ref, 0.0, 1.0), (False, 0.0, 0.0, 1.0))):
        got = expr(new())      # other activations: the fused kernel at once
        assert seen == [act] and type(got) is torch.Tensor and torch.equal(got, want)
        seen.clear()
    # torch.cat(dim 0) of deferred activations of one degree and form stays deferred; anything else concatenates ordinary tensors
    mk = lambda deg=3: torch.clamp(wrapper._LazySH(deg, d, c, None) + 0.5, 0.0, 1.0)
    z = torch.cat([mk(), mk()], dim=0)
    assert not seen and type(z) is wrapper._LazySH and z.shape == (14, 3) and z.requires_grad and len(z._lz_parts) == 2
    want = torch.clamp(ref + 0.5, 0.0, 1.0)
    assert torch.equal(z, torch.cat([want, want])) and seen == [(True, 0.5, 0.0, 1.0)] * 2
    seen.clear()
    assert type(torch.cat([mk()])) is wrapper._LazySH and type(torch.concat((mk(), mk()), 0)) is wrapper._LazySH and not seen
    for parts in ([mk(), mk(2)], [mk(), want], [mk(), torch.clamp_min(wrapper._LazySH(3, d, c, None) + 0.5, 0.0)]):
        z = torch.cat(parts)
        assert type(z) is torch.Tensor and z.shape == (14, 3)
    assert type(torch.cat([mk(), mk()], dim=1)) is torch.Tensor
    seen.clear()
    # further channels behind the colours (MTGS's predict_normals: torch.cat([rgbs, normals], dim=-1)) stay deferred too
    nrm = torch.randn(7, 3, requires_grad=True)
    z = torch.cat([mk(), nrm], dim=-1)
    assert not seen and type(z) is wrapper._LazySH and z.shape == (7, 6) and z.requires_grad and z._lz_extra is nrm
    assert torch.equal(z, torch.cat([want, nrm], dim=-1)) and seen == [(True, 0.5, 0.0, 1.0)]
    seen.clear()
    z = torch.cat([torch.cat([mk(), mk()]), torch.cat([nrm, nrm]).detach(), torch.ones(14, 1)], dim=1)      # nodes, then two tensors behind
    assert not seen and type(z) is wrapper._LazySH and z.shape == (14, 7) and z._lz_extra.shape == (14, 4)
    assert type(torch.cat([z, z])) is torch.Tensor      # (not a part of a node concatenation any more)
    seen.clear()
    for parts in ([nrm, mk()], [mk(), torch.randn(7, 6)], [mk(), torch.randn(6, 3)], [mk(), nrm.double()]):      # colours not first / > 8 channels / other N / other dtype
        assert type(torch.cat(parts, dim=-1) if parts[1].shape[0] == 7 else torch.cat([parts[0][:6], parts[1]], dim=-1)) is torch.Tensor
    seen.clear()
    # the evaluation is deferred, the inputs are not: an in-place change between the call and the first use is an error, as in autograd
    cc = torch.randn(7, 16, 3)
    z = torch.clamp(wrapper._LazySH(3, d, cc, None) + 0.5, 0.0, 1.0)
    cc.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        z * 1.0
    e2 = torch.randn(7, 3)
    z = torch.cat([mk(), e2], dim=-1)
    e2.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in\n? ?place|modified in place"):
        z * 1.0
    seen.clear()
    c8 = torch.randn(7, 9, 3)
    y = torch.clamp(wrapper._LazySH(2, d, c8, None) + 0.5, 0.0, 1.0)      # K != 16: at once
    assert seen == [(True, 0.5, 0.0, 1.0)] and type(y) is torch.Tensor
    seen.clear()
    assert torch.equal(new().clamp(max=0.25), ref.clamp(max=0.25)) and seen == [(False, 0.0, float("-inf"), 0.25)]
    seen.clear()
    # not the pattern: a tensor addend, alpha, a second add, tensor bounds, min > max, an `out=` -> ordinary tensors (act None)
    for expr, want in ((lambda z: z + torch.ones(7, 3), ref + 1), (lambda z: torch.add(z, 1.0, alpha=3), ref + 3.0),
                       (lambda z: (z + 0.5) + 0.25, ref + 0.5 + 0.25), (lambda z: torch.clamp(z, torch.zeros(7, 3), torch.ones(7, 3)), ref.clamp(0, 1)),
                       (lambda z: torch.clamp(z + 0.5, 1.0, 0.0), torch.clamp(ref + 0.5, 1.0, 0.0)), (lambda z: z * 2, ref * 2), (lambda z: z[2:4], ref[2:4])):
        seen.clear()
        got = expr(new())
        assert seen == [None] and torch.equal(got, want)
    # used twice: materialised ONCE
    seen.clear()
    z = new()
    a, b = z + 0.5, z.sum()
    assert torch.equal(a * 1, ref + 0.5) and seen == [None]
