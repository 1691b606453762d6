"""ctypes front end of the CPU oracle (oracle/gsplat_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gsplat_oracle.c.  PARITY UNPINNED (gsplat 1.4.0 is
not available to run; the restatement is pinned by known-answer tests and fp64 autograd instead).

All functions take / return numpy arrays on the host.  `rasterization()` restates the Python
orchestration of gsplat/rendering.py::rasterization (v1.4.0) for the option set MTGS drives
(/root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-659).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# MTGS_ORACLE_LIB: an alternative build of the same source (the sanitizer build of tests/test_oracle_sanitizers.py)
_LIB_PATH = Path(os.environ.get("MTGS_ORACLE_LIB") or _HERE / "libgsplat_oracle.so")

c_f = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
c_i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
c_i64 = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
c_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> Path:
    src = _HERE / "gsplat_oracle.c"
    if os.environ.get("MTGS_ORACLE_LIB"):
        return _LIB_PATH
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-s", "-C", str(_HERE), "-B", "libgsplat_oracle.so"])
    return _LIB_PATH


def build_native() -> Path:
    """A second build of the SAME source for bench.py's `cpu_baseline` leg only: -O3 -march=native (SURVEY.md section
    8d), FMA contraction allowed -- a timing baseline, not the checker (tests always use the -ffp-contract=off build).
    Built on the machine that runs it; the file name carries a hash of this CPU's feature flags so that a build made on
    another host is never loaded."""
    import hashlib
    flags = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = line
                break
    except OSError:
        pass
    out = _HERE / "_native" / f"libgsplat_oracle_native_{hashlib.sha1(flags.encode()).hexdigest()[:10]}.so"
    src = _HERE / "gsplat_oracle.c"
    if not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
        out.parent.mkdir(exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-std=c11", "-shared", "-o", str(out),
                               str(src), "-lm"])
    return out


def select_native() -> str:
    """Route this module's calls to the -O3 -march=native build (bench.py cpu_baseline).  Returns a label of the build in
    use; falls back to the checker build when gcc is missing."""
    global _lib
    try:
        path, label = build_native(), "gcc -O3 -march=native -fopenmp"
    except (OSError, subprocess.CalledProcessError):
        path, label = build(), "gcc -O2 -ffp-contract=off -fopenmp (checker build; gcc unavailable for a native build)"
    _lib = C.CDLL(str(path))
    _lib.orc_isect_count.restype = C.c_int64
    _lib.orc_num_threads.restype = C.c_int
    return label


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.orc_isect_count.restype = C.c_int64
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(C.c_int(n))


# ---------------------------------------------------------------- SH
def sh_fwd(degree, dirs, coeffs, masks=None):
    dirs, coeffs = _f(dirs), _f(coeffs)
    n, K = coeffs.shape[0], coeffs.shape[1]
    assert (degree + 1) ** 2 <= K and dirs.shape == (n, 3) and coeffs.shape == (n, K, 3)
    out = np.empty((n, 3), np.float32)
    m = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint8)
    lib().orc_sh_fwd(C.c_int64(n), K, degree, _p(dirs, C.c_float), _p(coeffs, C.c_float),
                     _p(m, C.c_uint8), _p(out, C.c_float))
    return out


def sh_bwd(degree, dirs, coeffs, v_colors, masks=None, need_v_dirs=False):
    dirs, coeffs, v_colors = _f(dirs), _f(coeffs), _f(v_colors)
    n, K = coeffs.shape[0], coeffs.shape[1]
    v_coeffs = np.empty_like(coeffs)
    v_dirs = np.empty_like(dirs) if need_v_dirs else None
    m = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint8)
    lib().orc_sh_bwd(C.c_int64(n), K, degree, _p(dirs, C.c_float), _p(coeffs, C.c_float),
                     _p(m, C.c_uint8), _p(v_colors, C.c_float), _p(v_coeffs, C.c_float),
                     _p(v_dirs, C.c_float))
    return v_coeffs, v_dirs


# ---------------------------------------------------------------- projection
def project_fwd(means, quats, scales, viewmats, Ks, W, H, eps2d=0.3, near=0.01, far=1e10,
                radius_clip=0.0, calc_compensations=False):
    means, quats, scales, viewmats, Ks = map(_f, (means, quats, scales, viewmats, Ks))
    N, Cc = means.shape[0], viewmats.shape[0]
    radii = np.empty((Cc, N), np.int32)
    means2d = np.empty((Cc, N, 2), np.float32)
    depths = np.empty((Cc, N), np.float32)
    conics = np.empty((Cc, N, 3), np.float32)
    comps = np.empty((Cc, N), np.float32) if calc_compensations else None
    lib().orc_project_fwd(Cc, C.c_int64(N), _p(means, C.c_float), _p(quats, C.c_float),
                          _p(scales, C.c_float), _p(viewmats, C.c_float), _p(Ks, C.c_float), W, H,
                          C.c_float(eps2d), C.c_float(near), C.c_float(far), C.c_float(radius_clip),
                          _p(radii, C.c_int32), _p(means2d, C.c_float), _p(depths, C.c_float),
                          _p(conics, C.c_float), _p(comps, C.c_float))
    return radii, means2d, depths, conics, comps


def project_bwd(means, quats, scales, viewmats, Ks, W, H, eps2d, radii, conics, comps,
                v_means2d, v_depths, v_conics, v_comps=None, need_v_viewmats=True):
    means, quats, scales, viewmats, Ks = map(_f, (means, quats, scales, viewmats, Ks))
    conics, v_means2d, v_depths, v_conics = map(_f, (conics, v_means2d, v_depths, v_conics))
    comps, v_comps = _f(comps), _f(v_comps)
    radii = np.ascontiguousarray(radii, dtype=np.int32)
    N, Cc = means.shape[0], viewmats.shape[0]
    v_means = np.empty((N, 3), np.float32)
    v_quats = np.empty((N, 4), np.float32)
    v_scales = np.empty((N, 3), np.float32)
    v_vm = np.empty((Cc, 4, 4), np.float32) if need_v_viewmats else None
    if v_comps is None:
        comps = None
    lib().orc_project_bwd(Cc, C.c_int64(N), _p(means, C.c_float), _p(quats, C.c_float),
                          _p(scales, C.c_float), _p(viewmats, C.c_float), _p(Ks, C.c_float), W, H,
                          C.c_float(eps2d), _p(radii, C.c_int32), _p(conics, C.c_float),
                          _p(comps, C.c_float), _p(v_means2d, C.c_float), _p(v_depths, C.c_float),
                          _p(v_conics, C.c_float), _p(v_comps, C.c_float), _p(v_means, C.c_float),
                          _p(v_quats, C.c_float), _p(v_scales, C.c_float), _p(v_vm, C.c_float))
    return v_means, v_quats, v_scales, v_vm


# ---------------------------------------------------------------- tiles
def tile_bits(n_tiles: int) -> int:
    return int(lib().orc_tile_bits(n_tiles))


def cam_bits(Cc: int) -> int:
    return int(lib().orc_cam_bits(Cc))


def isect_tiles(means2d, radii, depths, tile_size, tw, th, sort=True):
    """gsplat isect_tiles: returns tiles_per_gauss[C,N], isect_ids[M] (sorted), flatten_ids[M]."""
    means2d, depths = _f(means2d), _f(depths)
    radii = np.ascontiguousarray(radii, dtype=np.int32)
    Cc, N = radii.shape
    tpg = np.empty((Cc, N), np.int32)
    cum = np.empty((Cc * N,), np.int64)
    M = int(lib().orc_isect_count(Cc, C.c_int64(N), _p(means2d, C.c_float), _p(radii, C.c_int32),
                                  tile_size, tw, th, _p(tpg, C.c_int32), _p(cum, C.c_int64)))
    ids = np.empty((M,), np.int64)
    flat = np.empty((M,), np.int32)
    lib().orc_isect_emit(Cc, C.c_int64(N), _p(means2d, C.c_float), _p(radii, C.c_int32),
                         _p(depths, C.c_float), _p(cum, C.c_int64), tile_size, tw, th,
                         _p(ids, C.c_int64), _p(flat, C.c_int32))
    if sort:
        bits = 32 + tile_bits(tw * th) + cam_bits(Cc)
        lib().orc_sort_pairs(C.c_int64(M), bits, _p(ids, C.c_int64), _p(flat, C.c_int32))
    return tpg, ids, flat


def sort_pairs(keys, vals, key_bits):
    keys = np.array(keys, dtype=np.int64, copy=True)
    vals = np.array(vals, dtype=np.int32, copy=True)
    lib().orc_sort_pairs(C.c_int64(keys.shape[0]), key_bits, _p(keys, C.c_int64), _p(vals, C.c_int32))
    return keys, vals


def isect_offset_encode(isect_ids, Cc, tw, th):
    isect_ids = np.ascontiguousarray(isect_ids, dtype=np.int64)
    off = np.empty((Cc, th, tw), np.int32)
    lib().orc_isect_offsets(C.c_int64(isect_ids.shape[0]), _p(isect_ids, C.c_int64), Cc, tw, th,
                            _p(off, C.c_int32))
    return off


# ---------------------------------------------------------------- compositing
def blend_fwd(means2d, conics, colors, opacities, backgrounds, W, H, tile_size, offsets, flatten_ids,
              want_critical=False):
    means2d, conics, colors, opacities, backgrounds = map(_f, (means2d, conics, colors, opacities, backgrounds))
    Cc, N, D = colors.shape
    th, tw = offsets.shape[1:]
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    flatten_ids = np.ascontiguousarray(flatten_ids, dtype=np.int32)
    render = np.empty((Cc, H, W, D), np.float32)
    alphas = np.empty((Cc, H, W, 1), np.float32)
    last = np.empty((Cc, H, W), np.int32)
    crit = np.zeros((Cc, H, W), np.uint8) if want_critical else None
    crit_g = np.zeros((Cc, N), np.uint8) if want_critical else None
    lib().orc_blend_fwd_ex(Cc, C.c_int64(N), D, _p(means2d, C.c_float), _p(conics, C.c_float),
                           _p(colors, C.c_float), _p(opacities, C.c_float), _p(backgrounds, C.c_float),
                           W, H, tile_size, tw, th, _p(offsets, C.c_int32), _p(flatten_ids, C.c_int32),
                           C.c_int64(flatten_ids.shape[0]), _p(render, C.c_float), _p(alphas, C.c_float),
                           _p(last, C.c_int32), _p(crit, C.c_uint8), _p(crit_g, C.c_uint8))
    if want_critical:
        blend_fwd.critical_gaussians = crit_g.astype(bool)     # (side channel: the Gaussians that sit at a threshold themselves)
        return render, alphas, last, crit.astype(bool)
    return render, alphas, last


def blend_bwd(means2d, conics, colors, opacities, backgrounds, W, H, tile_size, offsets, flatten_ids,
              alphas, last_ids, v_render, v_alphas, absgrad=True, want_term_abs=False, pixel_mask=None):
    """want_term_abs: also returns term_abs[C,N,4+D] = per Gaussian the sums of |per-pixel term| of {conic 3, opacity 1,
    colour D} (test aid: the conditioning of each row's sum; for xy it is v_means2d_abs).
    pixel_mask bool[C,H,W] (with want_term_abs): additionally returns flip_terms[C,N,6+D] = the FLIP SENSITIVITY of every row
    {xy 2, conic 3, opacity 1, colour D}: every masked pixel (pass the threshold-critical ones) is composited and back-propagated
    once more per critical decision of its list with that decision inverted, and the absolute change of every Gaussian's
    per-pixel terms is summed -- how far a row can move when those decisions fall the other way (orc_blend_bwd_ex2)."""
    means2d, conics, colors, opacities, backgrounds = map(_f, (means2d, conics, colors, opacities, backgrounds))
    alphas, v_render, v_alphas = map(_f, (alphas, v_render, v_alphas))
    Cc, N, D = colors.shape
    th, tw = offsets.shape[1:]
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    flatten_ids = np.ascontiguousarray(flatten_ids, dtype=np.int32)
    last_ids = np.ascontiguousarray(last_ids, dtype=np.int32)
    v_means2d = np.empty((Cc, N, 2), np.float32)
    v_abs = np.empty((Cc, N, 2), np.float32) if absgrad else None
    v_conics = np.empty((Cc, N, 3), np.float32)
    v_colors = np.empty((Cc, N, D), np.float32)
    v_opac = np.empty((Cc, N), np.float32)
    term_abs = np.empty((Cc, N, 4 + D), np.float32) if want_term_abs else None
    mask_u8 = mask_terms = None
    if pixel_mask is not None:
        assert want_term_abs
        mask_u8 = np.ascontiguousarray(np.asarray(pixel_mask).reshape(Cc, H, W), dtype=np.uint8)
        mask_terms = np.empty((Cc, N, 6 + D), np.float32)
    lib().orc_blend_bwd_ex2(Cc, C.c_int64(N), D, _p(means2d, C.c_float), _p(conics, C.c_float),
                           _p(colors, C.c_float), _p(opacities, C.c_float), _p(backgrounds, C.c_float),
                           W, H, tile_size, tw, th, _p(offsets, C.c_int32), _p(flatten_ids, C.c_int32),
                           C.c_int64(flatten_ids.shape[0]), _p(alphas, C.c_float), _p(last_ids, C.c_int32),
                           _p(v_render, C.c_float), _p(v_alphas, C.c_float), _p(v_means2d, C.c_float),
                           _p(v_abs, C.c_float), _p(v_conics, C.c_float), _p(v_colors, C.c_float),
                           _p(v_opac, C.c_float), _p(term_abs, C.c_float), _p(mask_u8, C.c_uint8), _p(mask_terms, C.c_float))
    if mask_terms is not None:
        return v_means2d, v_abs, v_conics, v_colors, v_opac, term_abs, mask_terms
    if want_term_abs:
        return v_means2d, v_abs, v_conics, v_colors, v_opac, term_abs
    return v_means2d, v_abs, v_conics, v_colors, v_opac


def gaussians_on_pixels(pixel_mask, last_ids, offsets, flatten_ids, n_rows, tile_size=16, margin=8):
    """Test aid: bool[n_rows] -- the (camera, Gaussian) pairs that lie on the composited part of the tile list of some
    pixel of pixel_mask[C,H,W] (list start .. the pixel's last contributor + margin: a flipped alpha / transmittance decision
    at a pixel changes T for every later entry and the backward's suffix sums for every earlier one, and can move the
    point of termination by a few entries)."""
    Cc, H, W = pixel_mask.shape
    th, tw = offsets.shape[1:]
    off = np.append(offsets.reshape(-1).astype(np.int64), flatten_ids.shape[0])
    out = np.zeros(n_rows, bool)
    for c, y, x in zip(*np.nonzero(pixel_mask)):
        t = (c * th + y // tile_size) * tw + x // tile_size
        s, e = off[t], off[t + 1]
        e = min(e, int(last_ids[c, y, x]) + 1 + margin)
        if e > s:
            out[flatten_ids[s:e]] = True
    return out


# ---------------------------------------------------------------- orchestration
def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height,
                  near_plane=0.01, far_plane=1e10, radius_clip=0.0, eps2d=0.3, sh_degree=None,
                  tile_size=16, backgrounds=None, render_mode="RGB", rasterize_mode="classic"):
    """gsplat/rendering.py::rasterization (v1.4.0), packed=False, forward.  Returns
    (render[C,H,W,D'], alpha[C,H,W,1], meta) plus everything the backward needs in meta['_ctx']."""
    means, quats, scales, opacities, colors, viewmats, Ks = map(
        _f, (means, quats, scales, opacities, colors, viewmats, Ks))
    N, Cc = means.shape[0], viewmats.shape[0]
    aa = rasterize_mode == "antialiased"
    radii, means2d, depths, conics, comps = project_fwd(
        means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip, aa)
    opac = np.repeat(opacities[None, :], Cc, axis=0)
    if aa:
        opac = opac * comps
    if sh_degree is None:
        cols = np.broadcast_to(colors, (Cc,) + colors.shape[-2:]) if colors.ndim == 2 else colors
    else:
        c2w = np.linalg.inv(viewmats.astype(np.float64)).astype(np.float32)
        dirs = means[None, :, :] - c2w[:, None, :3, 3]
        shs = np.broadcast_to(colors, (Cc,) + colors.shape[-3:]) if colors.ndim == 3 else colors
        cols = np.stack([sh_fwd(sh_degree, dirs[c], shs[c], masks=radii[c] > 0) for c in range(Cc)])
        cols = np.maximum(cols + 0.5, 0.0)
    cols = np.ascontiguousarray(cols, dtype=np.float32)
    bg = backgrounds
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = np.concatenate([cols, depths[..., None]], axis=-1)
        if bg is not None:
            bg = np.concatenate([bg, np.zeros((Cc, 1), np.float32)], axis=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None].copy()
        if bg is not None:
            bg = np.zeros((Cc, 1), np.float32)
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    tpg, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, tw, th)
    offsets = isect_offset_encode(isect_ids, Cc, tw, th)
    render, alphas, last_ids, critical = blend_fwd(means2d, conics, cols, opac, bg, width, height, tile_size,
                                                   offsets, flatten_ids, want_critical=True)
    render_raw = render
    if render_mode in ("ED", "RGB+ED"):
        render = np.concatenate(
            [render[..., :-1], render[..., -1:] / np.maximum(alphas, np.float32(1e-10))], axis=-1)
    meta = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, opacities=opac,
                tile_width=tw, tile_height=th, tiles_per_gauss=tpg, isect_ids=isect_ids,
                flatten_ids=flatten_ids, isect_offsets=offsets, width=width, height=height,
                tile_size=tile_size, n_cameras=Cc, compensations=comps, colors=cols,
                backgrounds=bg, last_ids=last_ids, render_raw=render_raw, critical=critical,
                critical_gaussians=blend_fwd.critical_gaussians)
    return render, alphas, meta
