"""ctypes binding of libmtgs_rast.so (include/mtgs_rast.h).

The product path has NO fallback: if the HIP library is missing or a call fails, an exception is
raised.  Nothing here (or anywhere under mtgs_amd/) touches oracle/.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libmtgs_rast.so"


def use_library(path) -> None:
    """Development only (scripts/kbench.py --lib, A/B builds of scripts/build_variant.py): load another build of the
    library instead of the in-tree one.  Must be called before the first load(); nothing in the product reads the
    environment for this."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_library() after the library was loaded")
    LIB_PATH = Path(path)

_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t
_i64p = C.POINTER(C.c_int64)  # HOST array (row strides), nullable

# name -> argtypes (restype is int for all but the two introspection calls)
_SIGNATURES = {
    "mtgs_sh_fwd": [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "mtgs_sh_bwd": [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_sh_bwd_rows": [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "mtgs_sh_fwd_act": [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _vp, _vp],
    "mtgs_sh_bwd_act": [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_sh_bwd_rows_act": [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_fill_zero": [_vp, _sz, _vp],
    "mtgs_project_fwd": [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _f32,
                         _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "mtgs_project_bwd": [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp,
                         _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64p, _vp,
                         _vp, _vp, _i32, _i64p, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_project_bwd_zeroed": [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64p, _vp,
                                _vp, _vp, _i32, _i64p, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_project_bwd_blocks": [_i64, _i64p],
    "mtgs_isect_count": [_i32, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "mtgs_scan_workspace_bytes": [_i64, C.POINTER(_sz)],
    "mtgs_isect_scan": [_i64, _vp, _vp, _vp, _vp, _sz, _vp],
    "mtgs_isect_emit": [_i32, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "mtgs_sort_workspace_bytes": [_i64, C.POINTER(_sz)],
    "mtgs_sort_pairs": [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mtgs_bin_compact": [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp],
    "mtgs_bin_scan": [_i64, _vp, _vp, _vp, _vp, _sz, _vp],
    "mtgs_bin_emit": [_i64, _i64, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "mtgs_sort_u32_workspace_bytes": [_i64, C.POINTER(_sz)],
    "mtgs_sort_pairs_u32": [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mtgs_bin_sort_tiles": [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mtgs_bin_workspace_bytes": [_i64, _i64, C.POINTER(_sz)],
    "mtgs_bin_build": [_i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                       _vp, _sz, _vp],
    "mtgs_bin_finalize": [_i64, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "mtgs_front_workspace_bytes": [_i64, C.POINTER(_sz)],
    "mtgs_front_fwd": [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _vp, _i32, _i32,
                       _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32,
                       _vp, _vp, _i64, _vp, _sz, _vp, _sz, _vp],
    "mtgs_bin3_supported": [_i32, _i32, _i32, _i64],
    "mtgs_bin3_control_bytes": [_i32, _i32, _i32, C.POINTER(_sz)],
    "mtgs_bin3_workspace_bytes": [_i32, _i32, _i32, _i64, _i64, C.POINTER(_sz)],
    "mtgs_bin3_build": [_i32, _i64, _i32, _i32, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                        _vp, _sz, _vp],
    "mtgs_blend_fwd_packed": [_i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "mtgs_blend_touch_packed": [_i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp],
    "mtgs_blend_bwd_packed": [_i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _i64, _i32, _vp, _vp, _sz, _vp],
    "mtgs_isect_offsets": [_i64, _vp, _i32, _i32, _i32, _vp, _vp],
    "mtgs_dp_pack": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "mtgs_dp_accumulate": [_i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_dp_pack_ordered": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "mtgs_dp_touched_pack": [_i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "mtgs_dp_touched_pack_chunks": [_i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_dp_reduce_slices_cap": [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                  C.c_uint64, _i32, _i64, _vp],
    "mtgs_dp_union": [_i32, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_dp_reduce_rows_groups": [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _i64, _i64, _vp],
    "mtgs_dp_reduce_rows": [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _i64,
                            _vp, _vp, _vp, _vp, _i64, _i64, _vp],
    "mtgs_dp_reduce": [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp],
    "mtgs_dp_reduce_slices": [_i32, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                              C.c_uint64, _i32, _i64, _vp],
    "mtgs_project_bwd_rows": [_i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32,
                              _vp, _i64, _vp, _vp, _i32, _vp],
    "mtgs_node_fwd": [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64p, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                      _vp, _vp],
    "mtgs_node_bwd": [_i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                      _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "mtgs_node_desc_bytes": [],
    "mtgs_node_fwd_batch": [_i32, _vp, _i64, _i32, _vp, _vp, _vp],
    "mtgs_node_bwd_batch": [_i32, _vp, _i64, _i32, _vp, _vp],
    "mtgs_node_bwd_rows": [_i32, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp],
    "mtgs_normals_fwd": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "mtgs_normals_bwd": [_i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "mtgs_normals_bwd_rows": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "mtgs_normals_fwd_rows": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "mtgs_normals_bwd_qrows": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "mtgs_densify_stats": [_i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "mtgs_deform_embed": [_i64, _vp, _f32, _f32, _vp, _i32, _i32, _i32, _vp, _i64, _vp],
    "mtgs_fourier_dc_fwd": [_i64, _i32, _vp, _vp, _vp, _vp],
    "mtgs_fourier_dc_bwd": [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_refine_classify": [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_uint64, _i64,
                             _vp, _vp, _vp],
    "mtgs_refine_apply": [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_uint64, _i64,
                          _vp, _vp, _vp, _vp, _vp],
    "mtgs_refine_rows": [_i64, _i64, _vp, _vp, _vp, _i32, _vp, _vp],
    "mtgs_stats_desc_bytes": [],
    "mtgs_densify_stats_batch": [_i32, _vp, _i64, _vp, _vp, _i32, _i32, _vp],
    "mtgs_densify_stats_rows": [_i64, _vp, _vp, _i64, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp],
    "mtgs_ncc_patches": [_i32, _i32, _i32, _i32, _i64p],
    "mtgs_ncc_fwd": [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_ncc_bwd": [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_tv_workspace_floats": [_i32, _i32, _i32, C.POINTER(_sz)],
    "mtgs_tv_fwd": [_i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "mtgs_tv_bwd": [_i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "mtgs_oob_desc_bytes": [],
    "mtgs_oob_fwd": [_i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "mtgs_oob_bwd": [_i32, _vp, _i64, _vp, _vp, _vp, _vp],
    "mtgs_head_fwd": [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_head_workspace_floats": [_i32, _i32, C.POINTER(_sz)],
    "mtgs_head_bwd": [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_ssim_workspace_floats": [_i32, _i32, C.POINTER(_sz)],
    "mtgs_ssim_fwd": [_i32, _i32, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp],
    "mtgs_ssim_bwd": [_i32, _i32, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp],
    "mtgs_l1_workspace_floats": [_i32, _i32, C.POINTER(_sz)],
    "mtgs_l1_fwd": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_l1_bwd": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_inv_depth_l1_fwd": [_i32, _i32, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp],
    "mtgs_inv_depth_l1_bwd": [_i32, _i32, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp],
    "mtgs_campos_fwd": [_vp, _vp, _vp],
    "mtgs_campos_bwd": [_vp, _vp, _vp, _vp],
    "mtgs_loss_combine_fwd": [_i32, _vp, C.POINTER(C.c_float), C.c_uint, _f32, _vp, _vp, _vp],
    "mtgs_loss_combine_bwd": [_i32, _vp, _vp, C.POINTER(C.c_float), _vp, _vp],
    "mtgs_vis_color_fwd": [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp],
    "mtgs_vis_color_bwd": [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_vis_color_fwd_dirs": [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "mtgs_vis_color_bwd_dirs": [_i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mtgs_rows_expand": [_i64, _i32, _vp, _vp, _i64, _vp, _vp],
    "mtgs_adam_group_bytes": [],
    "mtgs_adam_block_elems": [],
    "mtgs_adam_block_rows": [],
    "mtgs_adam_block_list_rows": [],
    "mtgs_adam_step": [_i32, _vp, _vp, _i64, _i64, _i32, _vp],
    "mtgs_tile_schedule": [_i32, _i32, _i32, _vp, _i64, _vp, _vp],
    "mtgs_blend_fwd": [_i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                       _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "mtgs_blend_bwd": [_i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                       _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64p, _vp, _vp, _vp],
}
EXPORTS = ["mtgs_rast_version", "mtgs_rast_hot_version", "mtgs_rast_last_error"] + list(_SIGNATURES)
ABI_VERSION = 28
HOT_ABI_VERSION = 7      # hot-path subset (include/mtgs_rast.h MTGS_RAST_HOT_ABI_VERSION): what profiles/rNN_pmc_step.json is keyed on

_lib = None


def load() -> C.CDLL:
    """Loads the library (does not initialise the GPU).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m mtgs_amd.build` (needs hipcc). "
            "mtgs_amd has no CPU or PyTorch fallback for the rasterizer.")
    lib = C.CDLL(str(LIB_PATH))
    lib.mtgs_rast_version.restype = C.c_int
    lib.mtgs_rast_hot_version.restype = C.c_int
    lib.mtgs_rast_last_error.restype = C.c_char_p
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    v = lib.mtgs_rast_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libmtgs_rast.so ABI version {v} != expected {ABI_VERSION}; rebuild")
    if lib.mtgs_rast_hot_version() != HOT_ABI_VERSION:
        raise RuntimeError(f"libmtgs_rast.so hot-path ABI version {lib.mtgs_rast_hot_version()} != expected {HOT_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def host_i64(values):
    """HOST int64 array argument (None -> NULL)."""
    return None if values is None else (C.c_int64 * len(values))(*values)


def stream_of(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


_timed: dict = {}  # entry-point name -> list of (start_event, end_event); filled when enabled


def time_calls(names=()) -> None:
    """Bracket every call of the named entry points with HIP events on the current stream (the
    stream the kernels are launched on).  `time_calls(())` disables.  Used by bench.py only."""
    _timed.clear()
    for n in names:
        _timed[n] = []


def timed_ms() -> dict:
    """name -> list of elapsed milliseconds (call torch.cuda.synchronize() first)."""
    return {n: [s.elapsed_time(e) for s, e in evs] for n, evs in _timed.items()}


def call(name: str, *args) -> None:
    evs = _timed.get(name) if _timed else None
    if evs is not None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    rc = getattr(load(), name)(*args)
    if evs is not None:
        e.record()
        evs.append((s, e))
    if rc != 0:
        msg = load().mtgs_rast_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (code {rc}): {msg}")


def require_gpu(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "mtgs_amd: tensors must live on the HIP device (torch device 'cuda'); got "
                f"{t.device}. There is no CPU fallback in the product path.")
