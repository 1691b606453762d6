"""GPU parity: every stage of the HIP path, called through the C ABI, against the CPU oracle on
identical inputs.  Integer stages must be bit-exact; float stages within the stated tolerance
(north star: <= 1e-4 max abs fp32 on RGB / depth / alpha)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from tests.util import assert_grad_close, assert_image_close, assert_tile_lists, listed, small_scene, to_np

pytestmark = pytest.mark.gpu

RENDER_TOL = 1e-4


def dev(x):
    return torch.as_tensor(x).cuda()


@pytest.fixture(scope="module")
def gs(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import mtgs_amd
    return mtgs_amd


@pytest.mark.parametrize("degree,K", [(0, 1), (0, 16), (1, 16), (2, 16), (3, 16), (4, 25), (2, 9)])
@pytest.mark.parametrize("use_mask", [False, True])
def test_sh_fwd_bwd(gs, oracle, degree, K, use_mask):
    g = torch.Generator().manual_seed(degree * 10 + K)
    n = 1000 + 37
    dirs = torch.randn(n, 3, generator=g) * 3
    coeffs = torch.randn(n, K, 3, generator=g)
    masks = (torch.rand(n, generator=g) > 0.3) if use_mask else None
    vcol = torch.randn(n, 3, generator=g)
    ref = oracle.sh_fwd(degree, dirs.numpy(), coeffs.numpy(), None if masks is None else masks.numpy())
    ref_vc, ref_vd = oracle.sh_bwd(degree, dirs.numpy(), coeffs.numpy(), vcol.numpy(),
                                   None if masks is None else masks.numpy(), need_v_dirs=True)
    d, c = dev(dirs).requires_grad_(True), dev(coeffs).requires_grad_(True)
    out = gs.spherical_harmonics(degree, d, c, masks=None if masks is None else dev(masks))
    out.backward(dev(vcol))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(c.grad.cpu().numpy(), ref_vc, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(d.grad.cpu().numpy(), ref_vd, atol=5e-6, rtol=1e-4)


def test_sh_batched_dims(gs, oracle):
    g = torch.Generator().manual_seed(0)
    dirs = torch.randn(2, 50, 3, generator=g)
    coeffs = torch.randn(2, 50, 16, 3, generator=g)
    out = gs.spherical_harmonics(3, dev(dirs), dev(coeffs))
    ref = oracle.sh_fwd(3, dirs.reshape(-1, 3).numpy(), coeffs.reshape(-1, 16, 3).numpy()).reshape(2, 50, 3)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("aa", [False, True])
def test_projection_fwd_bit_exact(gs, oracle, aa):
    sc, vm, K = small_scene(N=5000, W=200, H=120)
    vm2 = torch.cat([vm, vm.clone()]); vm2[1, 0, 3] += 0.5
    K2 = torch.cat([K, K])
    a = to_np(sc)
    ref = oracle.project_fwd(a["means"], a["quats"], a["scales"], vm2.numpy(), K2.numpy(), 200, 120,
                             calc_compensations=aa)
    out = gs.fully_fused_projection(dev(sc["means"]), None, dev(sc["quats"]), dev(sc["scales"]), dev(vm2),
                                    dev(K2), 200, 120, calc_compensations=aa)
    names = ["radii", "means2d", "depths", "conics", "compensations"]
    assert (ref[0] > 0).sum() > 1000
    for nm, r, o in zip(names, ref, out):
        if r is None:
            assert o is None
            continue
        o = o.cpu().numpy()
        # the forward is compiled without FMA contraction: identical IEEE results are required
        assert np.array_equal(r.view(np.int32) if r.dtype == np.float32 else r,
                              o.view(np.int32) if o.dtype == np.float32 else o), f"{nm} not bit-exact"


def test_isect_sort_offsets_bit_exact(gs, oracle):
    sc, vm, K = small_scene(N=4000, W=333, H=211)  # not multiples of 16
    vm2 = torch.cat([vm, vm.clone()]); vm2[1, 2, 3] += 1.0
    K2 = torch.cat([K, K])
    W, H = 333, 211
    radii, means2d, depths, conics, _ = gs.fully_fused_projection(
        dev(sc["means"]), None, dev(sc["quats"]), dev(sc["scales"]), dev(vm2), dev(K2), W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = gs.isect_tiles(means2d, radii, depths, 16, tw, th)
    off = gs.isect_offset_encode(ids, 2, tw, th)
    r_tpg, r_ids, r_flat = oracle.isect_tiles(means2d.cpu().numpy(), radii.cpu().numpy(), depths.cpu().numpy(), 16, tw, th)
    r_off = oracle.isect_offset_encode(r_ids, 2, tw, th)
    assert r_ids.shape[0] > 10000
    assert np.array_equal(tpg.cpu().numpy(), r_tpg)
    assert np.array_equal(ids.cpu().numpy(), r_ids)
    assert np.array_equal(flat.cpu().numpy(), r_flat)
    assert np.array_equal(off.cpu().numpy(), r_off)
    # unsorted emission order is also specified (row-major over each rectangle)
    _, u_ids, u_flat = gs.isect_tiles(means2d, radii, depths, 16, tw, th, sort=False)
    _, ru_ids, ru_flat = oracle.isect_tiles(means2d.cpu().numpy(), radii.cpu().numpy(), depths.cpu().numpy(), 16, tw, th, sort=False)
    assert np.array_equal(u_ids.cpu().numpy(), ru_ids) and np.array_equal(u_flat.cpu().numpy(), ru_flat)


def test_sort_is_stable_on_masked_bits(gs, oracle):
    from mtgs_amd import _lib
    from mtgs_amd._lib import call, ptr
    g = torch.Generator().manual_seed(7)
    M = 100003
    keys = torch.randint(0, 1 << 20, (M,), generator=g, dtype=torch.int64) | (torch.randint(0, 4, (M,), generator=g, dtype=torch.int64) << 40)
    vals = torch.arange(M, dtype=torch.int32)
    r_k, r_v = oracle.sort_pairs(keys.numpy(), vals.numpy(), 12)  # only 12 bits significant -> many ties
    k, v = dev(keys), dev(vals)
    ko, vo = torch.empty_like(k), torch.empty_like(v)
    ws = C.c_size_t(0)
    call("mtgs_sort_workspace_bytes", M, C.byref(ws))
    w = torch.empty(ws.value, dtype=torch.uint8, device="cuda")
    call("mtgs_sort_pairs", M, 12, ptr(k), ptr(v), ptr(ko), ptr(vo), ptr(w), ws.value, torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(ko.cpu().numpy(), r_k) and np.array_equal(vo.cpu().numpy(), r_v)


@pytest.mark.parametrize("M", [1, 63, 1025, 5000, 300_000, 1 << 20, (1 << 20) + 1, 3_000_001])
@pytest.mark.parametrize("key_bits", [1, 7, 12, 23, 32, 33, 46, 64])
def test_sort_pairs_all_digit_plans(gs, M, key_bits):
    """mtgs_sort_pairs against torch's stable sort of the masked keys: up to 2^20 keys take 1024-key tiles, larger sorts
    4096-key tiles; every (size class, pass count, last-digit width) combination must be stable and must ignore the
    bits above key_bits."""
    from mtgs_amd._lib import call, ptr
    g = torch.Generator(device="cuda").manual_seed(M * 131 + key_bits)
    keys = torch.randint(-(1 << 62), 1 << 62, (M,), generator=g, dtype=torch.int64, device="cuda")
    if key_bits >= 12:   # ties in the low bits too: a few distinct values only
        keys = torch.where(torch.rand(M, device="cuda", generator=g) < 0.3, keys & 0x7, keys)
    vals = torch.arange(M, dtype=torch.int32, device="cuda")
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    ws = C.c_size_t(0)
    call("mtgs_sort_workspace_bytes", M, C.byref(ws))
    w = torch.empty(ws.value, dtype=torch.uint8, device="cuda")
    call("mtgs_sort_pairs", M, key_bits, ptr(keys), ptr(vals), ptr(ko), ptr(vo), ptr(w), ws.value,
         torch.cuda.current_stream().cuda_stream)
    masked = keys if key_bits == 64 else keys & ((1 << key_bits) - 1)
    if key_bits == 64:    # unsigned order of the 64-bit pattern
        masked = keys ^ (-(1 << 63))
    order = torch.sort(masked, stable=True).indices
    assert torch.equal(vo.long(), order)
    assert torch.equal(ko, keys[order])


CONFIGS = [
    # (render_mode, rasterize_mode, absgrad, D, backgrounds, W, H); W < 0: the camera pose carries NO gradient (|W| is the width)
    ("RGB", "classic", False, 3, False, 100, 70),
    # the 3DGS.py option cell itself (/root/reference/mtgs/config/3DGS.py:83-95: rasterize_mode classic, output_depth_during_training
    # -> RGB+ED, camera_optimizer off -> viewmats without a gradient; use_abs_grad off): classic / no absgrad / RGB+ED / D = 3
    ("RGB+ED", "classic", False, 3, False, -100, 70),
    ("RGB+ED", "antialiased", True, 3, False, 100, 70),   # 3DGS.py-like + MTGS flags
    ("RGB+ED", "antialiased", True, 6, False, 97, 61),    # MTGS.py: RGB + normals + depth = 7 channels
    ("RGB+D", "classic", True, 3, True, 64, 48),
    ("ED", "classic", False, 3, False, 64, 48),
    ("RGB", "classic", False, 12, True, 50, 40),          # padded to 16 channels, 4 waves per tile
    ("RGB", "classic", False, 40, False, 40, 30),         # channel chunks
]


@pytest.mark.parametrize("render_mode,rmode,absgrad,D,use_bg,W,H", CONFIGS)
def test_rasterization_fwd_bwd_vs_oracle(gs, oracle, render_mode, rmode, absgrad, D, use_bg, W, H):
    vm_grad, W = W > 0, abs(W)
    sc, vm, K = small_scene(N=400, W=W, H=H, D=D)
    a = to_np(sc)
    g = torch.Generator().manual_seed(11)
    n_out = {"RGB": D, "RGB+ED": D + 1, "RGB+D": D + 1, "ED": 1, "D": 1}[render_mode]
    bg = torch.rand(1, D, generator=g) if use_bg else None
    Gc = torch.randn(1, H, W, n_out, generator=g)
    Ga = torch.randn(1, H, W, 1, generator=g)
    # ---- oracle forward
    r_render, r_alpha, m = oracle.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"],
                                                vm.numpy(), K.numpy(), W, H, render_mode=render_mode,
                                                rasterize_mode=rmode, backgrounds=None if bg is None else bg.numpy())
    # ---- device forward
    P = {k: dev(v).requires_grad_(True) for k, v in sc.items()}
    vmd = dev(vm).requires_grad_(vm_grad)
    render, alpha, info = gs.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vmd, dev(K),
                                           W, H, packed=False, render_mode=render_mode, rasterize_mode=rmode,
                                           absgrad=absgrad, backgrounds=None if bg is None else dev(bg))
    info["means2d"].retain_grad()
    assert np.array_equal(info["radii"].cpu().numpy(), m["radii"])
    # the tile lists: ordered sublists of gsplat's in the default (tight) mode, gsplat's own bit for bit under exact_lists()
    assert_tile_lists(info, m, rerun=lambda: gs.rasterization(
        P["means"].detach(), P["quats"].detach(), P["scales"].detach(), P["opacities"].detach(), P["colors"].detach(), vmd.detach(),
        dev(K), W, H, packed=False, render_mode=render_mode, rasterize_mode=rmode, absgrad=absgrad,
        backgrounds=None if bg is None else dev(bg))[2])
    case = f"small {render_mode}/{rmode} D={D} {W}x{H}"
    assert_image_close(render.detach().cpu().numpy(), r_render, m["critical"], RENDER_TOL, name="render", case=case,
                       depth_channel=None if render_mode == "RGB" else -1, alpha=r_alpha)
    assert_image_close(alpha.detach().cpu().numpy(), r_alpha, m["critical"], RENDER_TOL, name="alpha", case=case)
    # ---- backward
    (render * dev(Gc)).sum().add((alpha * dev(Ga)).sum()).backward()
    Gc_raw, Ga_tot = Gc.numpy().copy(), Ga.numpy().copy()
    if render_mode in ("ED", "RGB+ED"):
        al = np.maximum(r_alpha, 1e-10)
        Gc_raw[..., -1:] = Gc.numpy()[..., -1:] / al
        Ga_tot = Ga_tot + (-(m["render_raw"][..., -1:] / al ** 2) * Gc.numpy()[..., -1:]) * (r_alpha > 1e-10)
    v2d, vabs, vcon, vcol, vop = oracle.blend_bwd(m["means2d"], m["conics"], m["colors"], m["opacities"], m["backgrounds"],
                                                  W, H, 16, m["isect_offsets"], m["flatten_ids"], r_alpha, m["last_ids"],
                                                  Gc_raw, Ga_tot)
    aa = rmode == "antialiased"
    has_depth = render_mode != "RGB"
    v_depth = vcol[..., -1].copy() if has_depth else np.zeros_like(vop)
    v_comp = vop * a["opacities"][None] if aa else None
    r_vm, r_vq, r_vs, r_vvm = oracle.project_bwd(a["means"], a["quats"], a["scales"], vm.numpy(), K.numpy(), W, H, 0.3,
                                                 m["radii"], m["conics"], m["compensations"], v2d, v_depth, vcon, v_comp)
    r_vo = (vop * (m["compensations"] if aa else 1.0)).sum(0)

    def close(name, got, ref):
        # (a 400-Gaussian scene: ONE pixel whose alpha >= 1/255 decision flips moves a gradient row by ~1e-3 of the maximum)
        assert_grad_close(name, got, ref, case=f"small {render_mode}/{rmode} D={D} {W}x{H}", rel_to_max=2e-3)

    close("means2d.grad", info["means2d"].grad, v2d)
    if absgrad:
        close("means2d.absgrad", info["means2d"].absgrad, vabs)
        assert (info["means2d"].absgrad >= info["means2d"].grad.abs() - 1e-6).all()
    close("v_means", P["means"].grad, r_vm)
    close("v_quats", P["quats"].grad, r_vq)
    close("v_scales", P["scales"].grad, r_vs)
    close("v_opacities", P["opacities"].grad, r_vo)
    if vm_grad:
        close("v_viewmats", vmd.grad[0], r_vvm[0])
    else:
        assert vmd.grad is None and not hasattr(info["means2d"], "absgrad")     # (3DGS.py: no pose gradient, no absgrad attribute)
    if render_mode not in ("ED", "D"):
        close("v_colors", P["colors"].grad, vcol[..., :D].sum(0))


def test_rasterization_sh_path(gs, oracle):
    sc, vm, K = small_scene(N=300, W=80, H=60, sh_degree=3)
    a = to_np(sc)
    r_render, r_alpha, m = oracle.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["coeffs"],
                                                vm.numpy(), K.numpy(), 80, 60, sh_degree=2)
    render, alpha, info = gs.rasterization(dev(sc["means"]), dev(sc["quats"]), dev(sc["scales"]), dev(sc["opacities"]),
                                           dev(sc["coeffs"]), dev(vm), dev(K), 80, 60, sh_degree=2, packed=False)
    assert_image_close(render.cpu().numpy(), r_render, m["critical"], RENDER_TOL, name="render")
    assert_image_close(alpha.cpu().numpy(), r_alpha, m["critical"], RENDER_TOL, name="alpha")


@pytest.mark.parametrize("N,W,H,deg,mode", [(300, 80, 60, 2, "RGB"), (20_000, 320, 200, 3, "RGB+ED")])
def test_rasterization_sh_path_backward(gs, oracle, N, W, H, deg, mode):
    """gsplat's own call style -- rasterization(colors=coeffs[N,K,3], sh_degree=n) -- forward AND backward against the oracle:
    SH masked with radii > 0, clamp_min(. + 0.5, 0), and DIFFERENTIABLE view directions (dirs = means - camera position:
    gradients reach means and, through inverse(viewmat), the view matrix).  Runs through the visibility-first one-node path
    (colours of the visible Gaussians only; dense coefficient gradient expanded from compact rows)."""
    sc, vm, K = small_scene(N=N, W=W, H=H, sh_degree=3)
    a = to_np(sc)
    g = torch.Generator().manual_seed(5)
    n_out = 3 + (mode != "RGB")
    Gc, Ga = torch.randn(1, H, W, n_out, generator=g), torch.randn(1, H, W, 1, generator=g)
    aa = mode != "RGB"
    rmode = "antialiased" if aa else "classic"
    r_render, r_alpha, m = oracle.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["coeffs"], vm.numpy(), K.numpy(),
                                                W, H, sh_degree=deg, render_mode=mode, rasterize_mode=rmode)
    P = {k: dev(v).requires_grad_(True) for k, v in sc.items() if k != "colors"}
    vmd = dev(vm).requires_grad_(True)
    render, alpha, info = gs.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["coeffs"], vmd, dev(K), W, H,
                                           sh_degree=deg, packed=False, render_mode=mode, rasterize_mode=rmode, absgrad=True)
    assert_tile_lists(info, m)
    assert_image_close(render.detach().cpu().numpy(), r_render, m["critical"], RENDER_TOL, name="render", case=f"sh_degree path N={N}",
                       depth_channel=-1 if aa else None, alpha=r_alpha)
    assert_image_close(alpha.detach().cpu().numpy(), r_alpha, m["critical"], RENDER_TOL, name="alpha", case=f"sh_degree path N={N}")
    torch.autograd.backward([render, alpha], [dev(Gc), dev(Ga)])
    # ---- oracle backward
    Gc_raw, Ga_tot = Gc.numpy().copy(), Ga.numpy().copy()
    if mode == "RGB+ED":
        al = np.maximum(r_alpha, 1e-10)
        Gc_raw[..., -1:] = Gc.numpy()[..., -1:] / al
        Ga_tot = Ga_tot + (-(m["render_raw"][..., -1:] / al ** 2) * Gc.numpy()[..., -1:]) * (r_alpha > 1e-10)
    v2d, vabs, vcon, vcol, vop = oracle.blend_bwd(m["means2d"], m["conics"], m["colors"], m["opacities"], None, W, H, 16,
                                                  m["isect_offsets"], m["flatten_ids"], r_alpha, m["last_ids"], Gc_raw, Ga_tot)
    v_depth = vcol[..., -1].copy() if aa else np.zeros_like(vop)
    v_comp = vop * a["opacities"][None] if aa else None
    r_vm, r_vq, r_vs, r_vvm = oracle.project_bwd(a["means"], a["quats"], a["scales"], vm.numpy(), K.numpy(), W, H, 0.3, m["radii"],
                                                 m["conics"], m["compensations"], v2d, v_depth, vcon, v_comp)
    campos = np.linalg.inv(vm.numpy()[0].astype(np.float64))[:3, 3].astype(np.float32)
    dirs = a["means"] - campos
    x = oracle.sh_fwd(deg, dirs, a["coeffs"], masks=m["radii"][0] > 0)
    v_rgb = vcol[0, :, :3] * (x + 0.5 >= 0.0)                     # clamp_min(x + 0.5, 0)
    r_vc, r_vd = oracle.sh_bwd(deg, dirs, a["coeffs"], v_rgb, masks=m["radii"][0] > 0, need_v_dirs=True)
    vm64 = torch.tensor(vm.numpy(), dtype=torch.float64, requires_grad=True)
    torch.inverse(vm64)[0, :3, 3].backward(torch.tensor(-r_vd.astype(np.float64).sum(0)))
    case = f"sh_degree path N={N} {mode}"
    tol = dict(rel_to_max=2e-3) if N < 1000 else {}
    assert_grad_close("v_coeffs", P["coeffs"].grad, r_vc, case=case, **tol)
    assert_grad_close("v_means (incl. view directions)", P["means"].grad, r_vm + r_vd, case=case, **tol)
    assert_grad_close("v_quats", P["quats"].grad, r_vq, case=case, **tol)
    assert_grad_close("v_viewmats (incl. camera position)", vmd.grad[0], r_vvm[0] + vm64.grad[0].numpy().astype(np.float32), case=case, **tol)


def test_empty_and_degenerate_inputs(gs):
    W, H = 40, 30
    vm = torch.eye(4)[None].cuda(); K = torch.tensor([[[30.0, 0, 20], [0, 30.0, 15], [0, 0, 1]]]).cuda()
    # every Gaussian behind the camera -> M = 0, zero image
    means = torch.tensor([[0.0, 0.0, -5.0], [1.0, 0.0, -2.0]]).cuda().requires_grad_(True)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 2).cuda(); scales = torch.full((2, 3), 0.1).cuda()
    opac = torch.full((2,), 0.5).cuda(); cols = torch.rand(2, 3).cuda()
    render, alpha, info = gs.rasterization(means, quats, scales, opac, cols, vm, K, W, H, packed=False, render_mode="RGB+ED")
    assert listed(info).numel() == 0 and (info["radii"] == 0).all()
    assert render.abs().max() == 0 and alpha.abs().max() == 0
    (render.sum() + alpha.sum()).backward()
    assert means.grad.abs().max() == 0
    # N = 0
    z = lambda *s: torch.zeros(*s).cuda()
    render, alpha, info = gs.rasterization(z(0, 3), z(0, 4), z(0, 3), z(0), z(0, 3), vm, K, W, H, packed=False)
    assert render.shape == (1, H, W, 3) and render.abs().max() == 0


def test_unsupported_options_raise(gs):
    z = lambda *s: torch.zeros(*s).cuda()
    args = (z(1, 3), z(1, 4), z(1, 3), z(1), z(1, 3), torch.eye(4)[None].cuda(), torch.eye(3)[None].cuda(), 32, 32)
    for kw in (dict(packed=True), dict(packed=False, sparse_grad=True), dict(packed=False, distributed=True),
               dict(packed=False, camera_model="fisheye"), dict(packed=False, tile_size=8),
               dict(packed=False, covars=z(1, 3, 3))):
        with pytest.raises(NotImplementedError):
            gs.rasterization(*args, **kw)


def test_depth_ordered_binning_equals_emit_then_sort(gs, oracle):
    """isect_tiles(sort=True) runs the depth-ordered binning (csrc/bin.hip); the gsplat formulation
    (emit in index order, stable 46-bit sort) must give the same arrays bit for bit, including ties
    in depth (duplicated Gaussians) and several cameras."""
    import ctypes as C
    from mtgs_amd._lib import call, ptr
    sc, vm, K = small_scene(N=3000, W=400, H=300)
    for k in ("means", "quats", "scales"):
        sc[k][1500:] = sc[k][:1500]          # exact duplicates -> equal depths, equal tiles
    vm3 = torch.cat([vm, vm.clone(), vm.clone()]); vm3[1, 0, 3] += 0.4; vm3[2, 2, 3] += 0.8
    K3 = torch.cat([K, K, K])
    W, H = 400, 300
    radii, means2d, depths, conics, _ = gs.fully_fused_projection(
        dev(sc["means"]), None, dev(sc["quats"]), dev(sc["scales"]), dev(vm3), dev(K3), W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = gs.isect_tiles(means2d, radii, depths, 16, tw, th)                 # depth-ordered path
    _, u_ids, u_flat = gs.isect_tiles(means2d, radii, depths, 16, tw, th, sort=False)   # gsplat formulation
    M = u_ids.numel()
    assert M > 20000
    ws = C.c_size_t(0)
    call("mtgs_sort_workspace_bytes", M, C.byref(ws))
    w = torch.empty(ws.value, dtype=torch.uint8, device="cuda")
    ko, vo = torch.empty_like(u_ids), torch.empty_like(u_flat)
    key_bits = 32 + int(tw * th).bit_length() + int(3).bit_length()
    call("mtgs_sort_pairs", M, key_bits, ptr(u_ids), ptr(u_flat), ptr(ko), ptr(vo), ptr(w), ws.value,
         torch.cuda.current_stream().cuda_stream)
    assert torch.equal(ids, ko) and torch.equal(flat, vo)
    r_tpg, r_ids, r_flat = oracle.isect_tiles(means2d.cpu().numpy(), radii.cpu().numpy(), depths.cpu().numpy(), 16, tw, th)
    assert np.array_equal(ids.cpu().numpy(), r_ids) and np.array_equal(flat.cpu().numpy(), r_flat)
    assert np.array_equal(tpg.cpu().numpy(), r_tpg)


GOLD = __import__("pathlib").Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("name", ["scene_classic_rgb", "scene_mtgs_like"])
def test_hip_reproduces_golden_fixture(gs, name):
    """The committed fixtures (tests/golden/make_golden.py): integer stages bit-exact, projection
    bit-exact, images <= 1e-4, gradients against the fp64 autograd values stored in the fixture."""
    z = np.load(GOLD / f"{name}.npz")
    W, H = int(z["W"]), int(z["H"])
    P = {k: dev(z[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    vm = dev(z["viewmat"]).requires_grad_(True)
    bg = dev(z["backgrounds"]) if z["backgrounds"].size else None
    render, alpha, info = gs.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm,
                                           dev(z["K"]), W, H, packed=False, render_mode=str(z["render_mode"]),
                                           rasterize_mode=str(z["rasterize_mode"]), backgrounds=bg, absgrad=True)
    for key in ("radii", "tiles_per_gauss"):
        assert np.array_equal(info[key].cpu().numpy(), z[key]), key
    assert_tile_lists(info, z, rerun=lambda: gs.rasterization(
        P["means"].detach(), P["quats"].detach(), P["scales"].detach(), P["opacities"].detach(), P["colors"].detach(), vm.detach(),
        dev(z["K"]), W, H, packed=False, render_mode=str(z["render_mode"]), rasterize_mode=str(z["rasterize_mode"]), backgrounds=bg,
        absgrad=True)[2])
    for key in ("means2d", "depths", "conics"):
        assert np.array_equal(info[key].detach().cpu().numpy(), z[key]), key
    has_depth = str(z["render_mode"]) != "RGB"
    nocrit = np.zeros(z["alpha"].shape[:3], bool)      # (the fixtures carry no critical-pixel map: every pixel is held to tol)
    assert_image_close(render.detach().cpu().numpy(), z["render"], nocrit, RENDER_TOL, name="render", case=f"fixture {name}",
                       depth_channel=-1 if has_depth else None, alpha=z["alpha"])
    assert_image_close(alpha.detach().cpu().numpy(), z["alpha"], nocrit, RENDER_TOL, name="alpha", case=f"fixture {name}")
    torch.autograd.backward([render, alpha], [dev(z["Gc"]), dev(z["Ga"])])
    # the same bars as every other comparison (fp64 autograd values in the fixture); 2e-3 of the maximum as for the other
    # few-hundred-Gaussian scenes of this file, where one flipped pixel is 1e-3 of a row
    for key, gname in [("means", "v_means"), ("quats", "v_quats"), ("scales", "v_scales"), ("opacities", "v_opacities"),
                       ("colors", "v_colors")]:
        assert_grad_close(gname, P[key].grad, z[gname], case=f"fixture {name}", rel_to_max=2e-3)
    assert_grad_close("v_viewmat", vm.grad, z["v_viewmat"], case=f"fixture {name}", rel_to_max=2e-3)


@pytest.mark.parametrize("name", ["gsplat_1_4_0_classic", "gsplat_1_4_0_mtgs"])
def test_hip_reproduces_gsplat_fixture(gs, name):
    """Outputs of the REAL gsplat 1.4.0 (tests/golden/make_gsplat_golden.py, run on a machine that has it).  Skipped until
    the two files are committed -- until then the hot path's parity is pinned by the oracle only ("parity unpinned").
    gsplat leaves culled rows of means2d / depths / conics unspecified and contracts FMAs, so the projection outputs are
    compared on visible rows with a relative tolerance; the integer stages, the image and the gradients as for the
    oracle-made fixtures."""
    path = GOLD / f"{name}.npz"
    if not path.exists():
        pytest.skip(f"{path.name} not generated yet (needs gsplat==1.4.0 + CUDA: tests/golden/make_gsplat_golden.py)")
    z = np.load(path)
    W, H = int(z["W"]), int(z["H"])
    P = {k: dev(z[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    vm = dev(z["viewmat"]).requires_grad_(True)
    bg = dev(z["backgrounds"]) if z["backgrounds"].size else None
    render, alpha, info = gs.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm,
                                           dev(z["K"]), W, H, packed=False, render_mode=str(z["render_mode"]),
                                           rasterize_mode=str(z["rasterize_mode"]), backgrounds=bg, absgrad=True)
    info["means2d"].retain_grad()
    vis = z["radii"] > 0
    assert np.array_equal(info["radii"].cpu().numpy() > 0, vis), "visibility differs from gsplat"
    assert np.array_equal(info["radii"].cpu().numpy(), z["radii"]), "radii differ from gsplat"
    for key in ("means2d", "depths", "conics", "opacities"):
        ref = z["opacities_eff" if key == "opacities" else key]
        np.testing.assert_allclose(info[key].detach().cpu().numpy()[vis], ref[vis], rtol=2e-5, atol=1e-6, err_msg=key)
    assert np.array_equal(info["tiles_per_gauss"].cpu().numpy(), z["tiles_per_gauss"])
    # (the default call returns gsplat's lists: bit for bit against the fixture; the rerun checks the opt-in tight lists as ordered
    #  sublists of the fixture's)
    assert "lists" not in z.files or str(z["lists"]) == "gsplat"
    assert info.get("n_listed") is None and info["flatten_ids"].numel() == z["flatten_ids"].shape[0]
    assert_tile_lists(info, z, rerun=lambda: gs.rasterization(
        P["means"].detach(), P["quats"].detach(), P["scales"].detach(), P["opacities"].detach(), P["colors"].detach(), vm.detach(),
        dev(z["K"]), W, H, packed=False, render_mode=str(z["render_mode"]), rasterize_mode=str(z["rasterize_mode"]), backgrounds=bg,
        absgrad=True)[2])
    case = f"gsplat fixture {name}"
    nocrit = np.zeros(z["alpha"].shape[:3], bool)
    assert_image_close(render.detach().cpu().numpy(), z["render"], nocrit, RENDER_TOL, name="render", case=case,
                       depth_channel=-1 if str(z["render_mode"]) != "RGB" else None, alpha=z["alpha"])
    assert_image_close(alpha.detach().cpu().numpy(), z["alpha"], nocrit, RENDER_TOL, name="alpha", case=case)
    torch.autograd.backward([render, alpha], [dev(z["Gc"]), dev(z["Ga"])])
    for key, gname in [("means", "v_means"), ("quats", "v_quats"), ("scales", "v_scales"), ("opacities", "v_opacities"),
                       ("colors", "v_colors")]:
        assert_grad_close(gname, P[key].grad, z[gname], case=case)
    assert_grad_close("v_viewmat", vm.grad, z["v_viewmat"], case=case)
    assert_grad_close("means2d.grad", info["means2d"].grad, z["v_means2d"], case=case)
    assert_grad_close("means2d.absgrad", info["means2d"].absgrad, z["v_means2d_abs"], case=case)


def test_camera_position_equals_torch_inverse(hip_lib):
    """mtgs_amd.rendering.camera_position = torch.inverse(viewmat)[:3, 3] (what gsplat's sh_degree path takes its view directions
    from), value and gradient with respect to the view matrix, for rigid and for general (sheared / scaled) matrices."""
    from mtgs_amd.rendering import camera_position
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(31)
    for k in range(6):
        A = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        if k >= 3:
            A = A @ torch.diag(torch.tensor([1.5, 0.7, 1.1])) + 0.1 * torch.randn(3, 3, generator=g)
        V = torch.eye(4)
        V[:3, :3], V[:3, 3] = A, torch.randn(3, generator=g) * 5
        cot = torch.randn(3, generator=g).to(dev)
        Vd = V.to(dev).double().requires_grad_(True)
        ref = torch.inverse(Vd)[:3, 3]
        (ref * cot.double()).sum().backward()
        Vf = V.to(dev).requires_grad_(True)
        got = camera_position(Vf)
        (got * cot).sum().backward()
        assert torch.allclose(got.double(), ref.detach(), rtol=1e-5, atol=1e-5)
        assert torch.allclose(Vf.grad.double(), Vd.grad, rtol=1e-4, atol=1e-4), k      # (all 16 entries, as torch.inverse reports them)
