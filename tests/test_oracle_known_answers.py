"""Known-answer tests that pin the CPU oracle WITHOUT gsplat (which cannot be run here): closed
forms, the named thresholds of the algorithm, and structural invariants.  Each constant named in
oracle/gsplat_oracle.c has a test below.

Every case runs against BOTH implementations: `impl` is the CPU oracle (always) and, under `-m gpu`, the HIP path
behind the same numpy interface (tests/hip_backend.py) -- so the edges (alpha = 1/255, T <= 1e-4, the 0.999 clamp,
the radius floor, the frustum clamp, the near plane, culling) are met by the product on purpose, not only through
random scenes."""
import math

import numpy as np
import pytest


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def impl(request):
    if request.param == "oracle":
        return request.getfixturevalue("oracle")
    request.getfixturevalue("hip_lib")
    from tests.hip_backend import HipBackend
    return HipBackend()



def cam(W, H, f=100.0):
    vm = np.eye(4, dtype=np.float32)[None]
    K = np.array([[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]], np.float32)
    return vm, K


def one_gaussian(z=5.0, s=0.2, x=0.0, y=0.0, opacity=0.8, color=(0.9, 0.5, 0.1)):
    return (np.array([[x, y, z]], np.float32), np.array([[1.0, 0, 0, 0]], np.float32),
            np.full((1, 3), s, np.float32), np.array([opacity], np.float32), np.array([color], np.float32))


def test_single_isotropic_gaussian_closed_form(impl):
    W, H, f, z, s, o = 64, 48, 100.0, 5.0, 0.2, 0.8
    vm, K = cam(W, H, f)
    means, quats, scales, opac, cols = one_gaussian(z, s, opacity=o)
    render, alpha, m = impl.rasterization(means, quats, scales, opac, cols, vm, K, W, H)
    var = (f * s / z) ** 2 + 0.3            # J Sigma J^T + eps2d I for an on-axis isotropic Gaussian
    assert m["radii"][0, 0] == math.ceil(3 * math.sqrt(var))
    np.testing.assert_allclose(m["means2d"][0, 0], [W / 2, H / 2], atol=1e-5)
    np.testing.assert_allclose(m["conics"][0, 0], [1 / var, 0, 1 / var], rtol=1e-5, atol=1e-7)
    ys, xs = np.mgrid[0:H, 0:W]
    d2 = (xs + 0.5 - W / 2) ** 2 + (ys + 0.5 - H / 2) ** 2
    a = np.minimum(0.999, o * np.exp(-0.5 * d2 / var))
    r = m["radii"][0, 0]
    # the Gaussian only exists in the tiles of its bounding square
    tx0, tx1 = math.floor((W / 2 - r) / 16), math.ceil((W / 2 + r) / 16)
    ty0, ty1 = math.floor((H / 2 - r) / 16), math.ceil((H / 2 + r) / 16)
    in_rect = (xs // 16 >= tx0) & (xs // 16 < tx1) & (ys // 16 >= ty0) & (ys // 16 < ty1)
    a = np.where((a >= 1 / 255) & in_rect, a, 0.0)
    np.testing.assert_allclose(alpha[0, ..., 0], a, atol=2e-6)
    np.testing.assert_allclose(render[0], a[..., None] * cols[0], atol=2e-6)


def test_two_gaussians_front_to_back(impl):
    W, H = 32, 32
    vm, K = cam(W, H)
    means = np.array([[0, 0, 4.0], [0, 0, 2.0]], np.float32)     # second one is nearer
    quats = np.tile(np.array([[1.0, 0, 0, 0]], np.float32), (2, 1))
    scales = np.full((2, 3), 0.5, np.float32)
    opac = np.array([0.9, 0.6], np.float32)
    cols = np.array([[1.0, 0, 0], [0, 1.0, 0]], np.float32)
    render, alpha, m = impl.rasterization(means, quats, scales, opac, cols, vm, K, W, H)
    assert list(m["flatten_ids"][:1]) == [1]                      # nearest first in every tile
    py, px = 16, 16
    def a_of(i):
        dx, dy = m["means2d"][0, i] - np.array([px + 0.5, py + 0.5])
        ca, cb, cc = m["conics"][0, i]
        return min(0.999, opac[i] * math.exp(-(0.5 * (ca * dx * dx + cc * dy * dy) + cb * dx * dy)))
    a_near, a_far = a_of(1), a_of(0)
    np.testing.assert_allclose(render[0, py, px], [a_far * (1 - a_near), a_near, 0], atol=1e-6)
    np.testing.assert_allclose(alpha[0, py, px, 0], 1 - (1 - a_near) * (1 - a_far), atol=1e-6)


def test_alpha_min_threshold(impl):                                   # ALPHA_MIN = 1/255
    W, H = 16, 16
    vm, K = cam(W, H)
    for o, expect in [(1 / 255 - 1e-4, False), (1 / 255 + 1e-4, True)]:
        means, quats, scales, opac, cols = one_gaussian(z=2.0, s=1.0, opacity=o)
        _, alpha, _ = impl.rasterization(means, quats, scales, opac, cols, vm, K, W, H)
        assert (alpha.max() > 0) == expect


def test_alpha_max_clamp(impl):                                       # ALPHA_MAX = 0.999
    W, H = 16, 16
    vm, K = cam(W, H)
    means, quats, scales, opac, cols = one_gaussian(z=2.0, s=5.0, opacity=1.0)
    _, alpha, _ = impl.rasterization(means, quats, scales, opac, cols, vm, K, W, H)
    assert abs(alpha.max() - 0.999) < 1e-6


def test_transmittance_stop_excludes_crossing_gaussian(impl):         # T_MIN = 1e-4
    W, H = 16, 16
    vm, K = cam(W, H)
    n = 3
    means = np.array([[0, 0, 2.0 + i] for i in range(n)], np.float32)
    quats = np.tile(np.array([[1.0, 0, 0, 0]], np.float32), (n, 1))
    scales = np.full((n, 3), 5.0, np.float32)
    opac = np.ones(n, np.float32)
    cols = np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]], np.float32)
    render, alpha, m = impl.rasterization(means, quats, scales, opac, cols, vm, K, W, H)
    # 1st: T = 1e-3.  2nd would give T = 1e-6 <= 1e-4 -> stop BEFORE compositing it.
    np.testing.assert_allclose(alpha[0, 8, 8, 0], 0.999, atol=1e-6)
    np.testing.assert_allclose(render[0, 8, 8], [0.999, 0, 0], atol=1e-6)
    if "last_ids" in m:   # (internal to the HIP compositing kernels; the oracle exposes it)
        assert m["last_ids"][0, 8, 8] == 0


def test_culling_rules(impl):
    W, H = 64, 48
    vm, K = cam(W, H)
    quat = np.array([[1.0, 0, 0, 0]], np.float32)
    sc = np.full((1, 3), 0.05, np.float32)
    def radius(mean, **kw):
        return impl.project_fwd(np.array([mean], np.float32), quat, sc, vm, K, W, H, **kw)[0][0, 0]
    assert radius([0, 0, 5.0]) > 0
    assert radius([0, 0, -5.0]) == 0                   # behind the camera
    assert radius([0, 0, 0.005]) == 0                  # z < near_plane (0.01)
    assert radius([0, 0, 5.0], far=4.0) == 0           # z > far_plane
    assert radius([50.0, 0, 5.0]) == 0                 # projects outside the image
    assert radius([0, 0, 5.0], radius_clip=10.0) == 0  # radius <= radius_clip
    # culled rows are zero-filled
    out = impl.project_fwd(np.array([[0, 0, -5.0]], np.float32), quat, sc, vm, K, W, H, calc_compensations=True)
    assert all(np.all(o == 0) for o in out)


def test_radius_floor_constant(impl):                                 # RADIUS_FLOOR = 0.01
    # a vanishing Gaussian: cov2d -> eps2d*I, b^2 - det = 0 -> floor 0.01 applies
    W, H = 32, 32
    vm, K = cam(W, H)
    means, quats, scales, opac, cols = one_gaussian(z=5.0, s=1e-6)
    r = impl.project_fwd(means, quats, scales, vm, K, W, H)[0][0, 0]
    assert r == math.ceil(3 * math.sqrt(0.3 + math.sqrt(0.01)))


def test_fov_clamp_margin(impl):                                      # FOV_MARGIN = 0.3
    # far off-axis Gaussian: the Jacobian uses the clamped position, the mean does not
    W, H, f = 64, 48, 100.0
    vm, K = cam(W, H, f)
    K[0, 0, 2] = 20.0                                  # off-centre principal point
    z, x, s = 2.0, 3.0, 0.3
    means, quats, scales, opac, cols = one_gaussian(z=z, s=s, x=x)
    radii, m2d, depths, conics, _ = impl.project_fwd(means, quats, scales, vm, K, W, H)
    lim_pos = (W - 20.0) / f + 0.3 * (0.5 * W / f)
    tx = z * min(lim_pos, x / z)
    J = np.array([[f / z, 0, -f * tx / z ** 2], [0, f / z, 0]])
    cov = J @ (np.eye(3) * s * s) @ J.T + 0.3 * np.eye(2)
    inv = np.linalg.inv(cov)
    # the Gaussian is culled by the image-bounds test or kept; conic is checked only when kept
    if radii[0, 0] > 0:
        np.testing.assert_allclose(conics[0, 0], [inv[0, 0], inv[0, 1], inv[1, 1]], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(m2d[0, 0, 0], f * x / z + 20.0, rtol=1e-6)
    else:
        # widen the image so it is kept
        radii, m2d, depths, conics, _ = impl.project_fwd(means, quats, scales, vm, K, 400, H)
        assert radii[0, 0] > 0


def test_compensation_in_unit_interval_and_value(impl):
    W, H, f, z, s = 64, 48, 100.0, 5.0, 0.05
    vm, K = cam(W, H, f)
    means, quats, scales, opac, cols = one_gaussian(z=z, s=s)
    comp = impl.project_fwd(means, quats, scales, vm, K, W, H, calc_compensations=True)[4][0, 0]
    v = (f * s / z) ** 2
    np.testing.assert_allclose(comp, math.sqrt(v * v / ((v + 0.3) ** 2)), rtol=1e-5)
    assert 0 < comp <= 1


def test_tile_rect_and_key_layout(impl, oracle):
    tw, th, ts = 7, 5, 16                        # 35 tiles -> tile_bits = 6
    assert oracle.tile_bits(35) == 6 and oracle.tile_bits(32) == 6 and oracle.tile_bits(31) == 5
    assert oracle.tile_bits(8160) == 13 and oracle.cam_bits(1) == 1 and oracle.cam_bits(2) == 2
    means2d = np.array([[[16.0, 16.0], [40.0, 40.0], [-100.0, 5.0], [1000.0, 1000.0]]], np.float32)
    radii = np.array([[8, 1, 10, 2000]], np.int32)
    depths = np.array([[2.0, 1.0, 3.0, 4.0]], np.float32)
    tpg, ids, flat = impl.isect_tiles(means2d, radii, depths, ts, tw, th, sort=False)
    # g0 straddles 4 tiles: [floor(8/16), ceil(24/16)) = [0,2) in x and y
    assert list(tpg[0]) == [4, 1, 0, 35]
    assert list(flat[:5]) == [0, 0, 0, 0, 1]
    assert [int(i >> 32) for i in ids[:5]] == [0, 1, 7, 8, 2 * 7 + 2]       # row-major tile ids
    assert int(ids[0] & 0xFFFFFFFF) == np.float32(2.0).view(np.int32)
    tpg, ids, flat = impl.isect_tiles(means2d, radii, depths, ts, tw, th, sort=True)
    assert np.all(np.diff(ids) >= 0)
    off = impl.isect_offset_encode(ids, 1, tw, th)
    assert off[0, 0, 0] == 0 and np.all(np.diff(off.ravel()) >= 0) and off.ravel()[-1] <= len(ids)
    # tile 0 holds g0 (depth 2) then g3 (depth 4)
    assert list(flat[off[0, 0, 0]:off[0, 0, 1]]) == [0, 3]


def test_sort_is_stable(impl):
    keys = np.array([5, 3, 5, 3, 1 << 40 | 3, 0], np.int64)
    vals = np.arange(6, dtype=np.int32)
    k, v = impl.sort_pairs(keys, vals, 8)          # the bit above bit 8 is ignored
    assert list(v) == [5, 1, 3, 4, 0, 2]


def test_offsets_with_empty_tiles_and_no_intersections(impl):
    off = impl.isect_offset_encode(np.zeros((0,), np.int64), 2, 3, 2)
    assert off.shape == (2, 2, 3) and np.all(off == 0)
    ids = np.array([(4 << 32) | 7, (4 << 32) | 9, (9 << 32) | 1], np.int64)   # cam0 tile4 x2, cam1 tile1
    off = impl.isect_offset_encode(ids, 2, 4, 2)
    assert list(off.ravel()) == [0, 0, 0, 0, 0, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3]


def test_sh_basis_orthonormal(impl):
    # Gauss-Legendre x uniform-phi quadrature of the 25 basis functions
    n_t, n_p = 32, 64
    xs, ws = np.polynomial.legendre.leggauss(n_t)
    phi = (np.arange(n_p) + 0.5) * 2 * np.pi / n_p
    ct, ph = np.meshgrid(xs, phi, indexing="ij")
    st = np.sqrt(1 - ct ** 2)
    dirs = np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3).astype(np.float32)
    w = (ws[:, None] * np.ones(n_p)[None] * 2 * np.pi / n_p).reshape(-1)
    B = np.zeros((dirs.shape[0], 25))
    for k in range(25):
        coeffs = np.zeros((dirs.shape[0], 25, 3), np.float32)
        coeffs[:, k, 0] = 1.0
        B[:, k] = impl.sh_fwd(4, dirs, coeffs)[:, 0]
    G = (B * w[:, None]).T @ B
    assert np.abs(G - np.eye(25)).max() < 2e-5


def test_sh_degree_zero_is_view_independent_and_inverts_rgb2sh(impl):
    rgb = np.random.default_rng(0).random((10, 3)).astype(np.float32)
    coeffs = np.zeros((10, 16, 3), np.float32)
    coeffs[:, 0] = (rgb - 0.5) / 0.28209479177387814
    coeffs[:, 1:] = 7.0                                  # must be ignored at degree 0
    dirs = np.random.default_rng(1).standard_normal((10, 3)).astype(np.float32)
    np.testing.assert_allclose(impl.sh_fwd(0, dirs, coeffs) + 0.5, rgb, atol=1e-6)


def test_absgrad_dominates_grad(impl):
    from tests.util import small_scene, to_np
    sc, vm, K = small_scene(N=200, W=64, H=48)
    a = to_np(sc)
    g = np.random.default_rng(0)
    Gc, Ga = g.standard_normal((1, 48, 64, 3)).astype(np.float32), g.standard_normal((1, 48, 64, 1)).astype(np.float32)
    if hasattr(impl, "grads"):
        v2d, vabs, al = impl.grads(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], vm.numpy(), K.numpy(), 64, 48,
                                   Gc, Ga)
    else:
        r, al, m = impl.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], vm.numpy(), K.numpy(), 64, 48)
        out = impl.blend_bwd(m["means2d"], m["conics"], m["colors"], m["opacities"], None, 64, 48, 16, m["isect_offsets"],
                             m["flatten_ids"], al, m["last_ids"], Gc, Ga)
        v2d, vabs = out[0], out[1]
    assert np.all(vabs >= np.abs(v2d) - 1e-6)
    assert np.all((al >= 0) & (al < 1))


def test_flip_sensitivity_of_a_threshold_critical_pixel(oracle):
    """orc_blend_bwd_ex2 (test infrastructure of the row accounting): at a threshold-critical pixel the oracle re-composites
    with the critical decision inverted and reports per Gaussian how far its per-pixel terms move.  Hand case: ONE pixel, a front
    Gaussian whose alpha sits 5e-5 above 1/255 (included by the oracle; excluding it is the flip) and an ordinary one behind:
    the front one's colour term a0 |vr| disappears, the back one's a1 T1 |vr| loses the factor (1 - a0): a change of a1 a0 |vr|.
    No mask -> zeros; the gradients do not depend on the mask; an uncritical scene has no sensitivity anywhere."""
    W = H = 16
    px, py = 8, 8
    means2d = np.array([[[px + 0.5 + 1.0, py + 0.5 + 0.5], [px + 0.5 - 0.5, py + 0.5 + 0.25]]], np.float32)
    conics = np.array([[[0.02, 0.0, 0.03], [0.05, 0.01, 0.04]]], np.float32)
    d0 = means2d[0, 0] - np.array([px + 0.5, py + 0.5], np.float32)
    sigma0 = 0.5 * (0.02 * d0[0] ** 2 + 0.03 * d0[1] ** 2)
    a_min = 1.0 / 255.0
    op0 = np.float32(a_min * (1 + 5e-5) * math.exp(sigma0))
    opac = np.array([[op0, 0.5]], np.float32)
    colors = np.array([[[0.9, 0.2, 0.4], [0.1, 0.8, 0.3]]], np.float32)
    offsets = np.zeros((1, 1, 1), np.int32)
    flat = np.array([0, 1], np.int32)
    r, al, last, crit = oracle.blend_fwd(means2d, conics, colors, opac, None, W, H, 16, offsets, flat, want_critical=True)
    assert crit[0, py, px] and oracle.blend_fwd.critical_gaussians[0, 0] and not oracle.blend_fwd.critical_gaussians[0, 1]
    g = np.random.default_rng(1)
    Gc, Ga = g.standard_normal(r.shape).astype(np.float32), np.zeros(al.shape, np.float32)
    mask = np.zeros((1, H, W), bool)
    mask[0, py, px] = True
    args = (means2d, conics, colors, opac, None, W, H, 16, offsets, flat, al, last, Gc, Ga)
    base = oracle.blend_bwd(*args, want_term_abs=True)
    out = oracle.blend_bwd(*args, want_term_abs=True, pixel_mask=mask)
    for x, y in zip(base, out[:6]):
        np.testing.assert_array_equal(x, y)
    flip = out[6]
    vr = np.abs(Gc[0, py, px])
    d1 = means2d[0, 1] - np.array([px + 0.5, py + 0.5], np.float32)
    a0 = op0 * math.exp(-sigma0)
    a1 = 0.5 * math.exp(-(0.5 * (0.05 * d1[0] ** 2 + 0.04 * d1[1] ** 2) + 0.01 * d1[0] * d1[1]))
    np.testing.assert_allclose(flip[0, 0, 6:9], a0 * vr, rtol=1e-4)
    np.testing.assert_allclose(flip[0, 1, 6:9], a1 * a0 * vr, rtol=1e-3)
    assert float(flip[0, :, :6].max()) > 0.0                              # geometry terms move too
    none = oracle.blend_bwd(*args, want_term_abs=True, pixel_mask=np.zeros_like(mask))
    assert float(np.abs(none[6]).max()) == 0.0
    opac2 = np.array([[0.7, 0.5]], np.float32)                            # nothing near a threshold: a masked pixel adds nothing
    r2, al2, last2, crit2 = oracle.blend_fwd(means2d, conics, colors, opac2, None, W, H, 16, offsets, flat, want_critical=True)
    assert not crit2.any()
    out2 = oracle.blend_bwd(means2d, conics, colors, opac2, None, W, H, 16, offsets, flat, al2, last2, Gc, Ga, want_term_abs=True,
                            pixel_mask=np.ones_like(mask))
    assert float(np.abs(out2[6]).max()) == 0.0
