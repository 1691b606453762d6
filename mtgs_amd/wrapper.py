"""Operator layer: the functions of gsplat/cuda/_wrapper.py (v1.4.0) that MTGS's rasterization path
uses, with the same names, argument meaning and error behaviour, on top of libmtgs_rast.so.

Reference call sites: spherical_harmonics at
/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:312-318 (also
multi_color_gaussian_splatting.py:96, rigid_node.py:248, deformable_node.py:125); the other four
are reached through gsplat.rendering.rasterization (mtgs_scene_graph.py:641-662).

PyTorch is used for device memory, streams and autograd plumbing only; every computation below is
a HIP kernel launched through the C ABI (include/mtgs_rast.h).
"""
from __future__ import annotations

import ctypes as C
import os
import contextlib
import threading
import time
import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import call, host_i64, ptr, require_gpu, stream_of

SUPPORTED_CHANNELS = (1, 2, 3, 4, 5, 6, 7, 8, 16, 32)
MAX_CHANNELS = 32


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32 tensor, got {t.dtype}")
    return t.contiguous()


def _strided_rows(t: Optional[Tensor], width: int):
    """(tensor, row stride in floats) for a [C, N(, width)] gradient whose C*N rows are uniformly strided with
    unit element stride -- e.g. a view of the interleaved buffer the compositing backward accumulates into;
    anything else is made contiguous."""
    if t is None:
        return None, width
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32 tensor, got {t.dtype}")
    Cn, N = t.shape[:2]
    rs = t.stride(1)
    ok = rs >= width and (t.dim() == 2 or t.stride(2) == 1 or width == 1) and (Cn == 1 or t.stride(0) == N * rs)
    if N <= 1 or not ok:
        return t.contiguous(), width
    return t, rs


# ------------------------------------------------------------------------------------- SH
class _ZeroRequest:
    """A [n, K, 3] buffer of zeros that a spherical_harmonics() backward will want (see _Prefill)."""
    __slots__ = ("shape", "device", "buffer", "stream", "thread", "fwd_stream", "__weakref__")

    def __init__(self, shape, device):
        self.shape, self.device, self.buffer, self.stream = tuple(shape), device, None, None
        # where the spherical_harmonics() forward ran: only a rasterization on the same (thread, stream) serves the request
        self.thread = threading.get_ident()
        self.fwd_stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0


class _Prefill:
    """ZEROS THE BACKWARD PASS WANTS, WRITTEN BY THE COMPOSITING KERNELS (round 5).  dL/dcoeffs of spherical_harmonics() is
    [N, K, 3] -- 384 MB at the headline workload -- of which the rows of the ~6 % composited Gaussians are non-zero; writing it is
    what the SH backward costs (81 us, HBM-bound), and the zeros depend on nothing.  The compositing kernels of the rasterizer are
    VALU-bound and leave HBM ~85 % idle for 0.16 + 0.33 ms.  So: a spherical_harmonics() forward whose coefficients need a
    gradient leaves a request on its autograd node; the rasterization() forward whose `colors` DESCEND from that node (when something
    needs a gradient) allocates the buffers -- one region, plus the gradient rows its own compositing backward accumulates into --
    and hands it to mtgs_blend_fwd_packed(also_zero), whose waves clear a slice each when their tile is done; the SH backward then
    only writes the non-zero rows (mtgs_sh_bwd_rows).  One stream, nothing to join, nothing special inside a graph capture.

    SCOPE (round 6).  A request is served only by a rasterization
      * whose colours are computed from that spherical_harmonics() output (found on the autograd graph behind `colors`:
        clamp / add / cat / expand ... up to 64 nodes), and
      * that runs on the thread and stream the SH forward ran on;
    a second model on the device, an evaluation pass with gradients, a viewer thread never pay allocations or HBM writes for someone
    else's backward.  A request nobody serves -- spherical_harmonics() without the rasterizer -- falls back to the one-kernel dense
    backward and EXPIRES at the next spherical_harmonics() forward of the same shape on its thread; a forward that is never
    differentiated has written zeros nobody reads (~10 us).
    LIFETIME: every buffer served by one take() is a view of ONE region (the kernel clears one address range): the region stays
    allocated while any of those dL/dcoeffs (a parameter's .grad) is alive -- with one model per rasterization that is the gradient
    itself plus 64 bytes per visible Gaussian.  `max_region_bytes` bounds a region (larger requests take the dense backward).
    Switches: `mtgs_amd.sh_prefill(enabled=..., in_forward=...)` (context manager); the environment variables MTGS_SH_PREFILL /
    MTGS_PREFILL_IN_FORWARD only set the defaults at import."""

    def __init__(self):
        self.lock = threading.Lock()
        self.pending = weakref.WeakSet()
        self.enabled = os.environ.get("MTGS_SH_PREFILL", "1") == "1"
        self.in_forward = os.environ.get("MTGS_PREFILL_IN_FORWARD", "1") == "1"
        self.max_region_bytes = int(os.environ.get("MTGS_PREFILL_MAX_BYTES", str(8 << 30)))
        self.regions = 0          # regions allocated so far / bytes of the last one (tests, diagnostics)
        self.last_region_bytes = 0

    def request(self, shape, device) -> Optional[_ZeroRequest]:
        if not self.enabled:
            return None
        req = _ZeroRequest(shape, device)
        with self.lock:
            for old in [r for r in self.pending if r.buffer is None and r.thread == req.thread and r.device == req.device
                        and r.shape == req.shape]:
                self.pending.discard(old)          # an unserved request of an earlier forward of this shape: expired
            self.pending.add(req)
        return req

    @staticmethod
    def behind(t, max_nodes: int = 64):
        """The requests of the spherical_harmonics() forwards tensor `t` is computed from (breadth first over its autograd graph)."""
        found, seen = [], set()
        queue = [t.grad_fn] if (t is not None and t.grad_fn is not None) else []
        while queue and len(seen) < max_nodes:
            node = queue.pop(0)
            if node is None or node in seen:
                continue
            seen.add(node)
            req = getattr(node, "zeros", None)
            if isinstance(req, _ZeroRequest):
                found.append(req)
                continue                           # (nothing of interest behind an SH node)
            queue.extend(fn for fn, _ in node.next_functions)
        return found

    def take(self, device, extra_floats: int = 0, only=()):
        """Called by a rasterization forward / backward in front of its compositing kernel: (pointer, bytes, extra) of ONE region
        holding the buffers of the still unserved requests among `only` (the requests behind this rasterization's colours; same
        device, same stream -- and, in a forward, same thread) and `extra_floats` more for the caller (`extra`: a flat view), or
        (None, 0, None)."""
        stream = torch.cuda.current_stream(device).cuda_stream      # (the kernel that clears the region is enqueued there)
        with self.lock:
            mine = [r for r in only if r in self.pending and r.device == device and r.buffer is None and r.fwd_stream == stream]
            sizes = [-(-int(torch.Size(r.shape).numel()) // 4) * 4 for r in mine]      # floats, padded to 16 bytes
            own = -(-int(extra_floats) // 4) * 4
            while mine and (sum(sizes) + own) * 4 > self.max_region_bytes:
                mine.pop(), sizes.pop()            # (beyond the budget: that request takes the dense backward)
            for r in mine:
                self.pending.discard(r)
        if not mine and not extra_floats:
            return None, 0, None
        region = torch.empty(sum(sizes) + own, dtype=torch.float32, device=device)
        self.regions, self.last_region_bytes = self.regions + 1, region.numel() * 4
        at = own
        for r, n in zip(mine, sizes):
            r.buffer, r.stream = region[at:at + int(torch.Size(r.shape).numel())].view(r.shape), stream
            at += n
        return region.data_ptr(), region.numel() * 4, (region[:int(extra_floats)] if extra_floats else None)


_prefill = _Prefill()
_sh_scope = threading.local()      # .reqs: the requests behind the colours of the rasterization being recorded (fused_rasterization)


@contextlib.contextmanager
def sh_prefill(enabled: bool = True, in_forward: bool = True):
    """Switches of the zero prefill of dL/dcoeffs (see _Prefill) for the enclosed calls: enabled = False -> every
    spherical_harmonics() backward is the one-kernel dense one; in_forward = False -> the zeros ride on the compositing BACKWARD
    instead of the forward.  Process-wide while active (autograd runs backward nodes on its own threads)."""
    old = (_prefill.enabled, _prefill.in_forward)
    _prefill.enabled, _prefill.in_forward = bool(enabled), bool(in_forward)
    try:
        yield
    finally:
        _prefill.enabled, _prefill.in_forward = old


class _SphericalHarmonics(torch.autograd.Function):
    """spherical_harmonics(), optionally with the caller's colour activation fused: act = None, or (has_add, add, lo, hi) for
    colors = clamp(SH + add, lo, hi) (csrc/sh.hip, ShAct; the decision is _LazySH's)."""

    @staticmethod
    def forward(ctx, degree: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor], act=None):
        require_gpu(dirs, coeffs, masks)
        dirs_c, coeffs_c = _f32c(dirs), _f32c(coeffs)
        K = coeffs_c.shape[-2]
        n = dirs_c.numel() // 3
        masks_c = None if masks is None else masks.contiguous().to(torch.uint8)
        colors = torch.empty(dirs_c.shape, dtype=torch.float32, device=dirs_c.device)
        passed = None
        if act is None:
            call("mtgs_sh_fwd", n, K, degree, ptr(dirs_c), ptr(coeffs_c), ptr(masks_c), ptr(colors),
                 stream_of(dirs_c))
        else:
            has_add, add, lo, hi = act
            passed = torch.empty(max(n, 1), dtype=torch.uint8, device=dirs_c.device)      # bit c: channel c passes its cotangent
            call("mtgs_sh_fwd_act", n, K, degree, ptr(dirs_c), ptr(coeffs_c), ptr(masks_c), ptr(colors), int(has_add), float(add),
                 float(lo), float(hi), ptr(passed), stream_of(dirs_c))
        ctx.save_for_backward(dirs_c, coeffs_c, masks_c, passed)
        ctx.degree, ctx.K, ctx.n = degree, K, n
        # (see _Prefill: zeros for dL/dcoeffs, filled while the rasterizer's backward runs)
        ctx.zeros = _prefill.request(coeffs_c.shape, coeffs_c.device) if (ctx.needs_input_grad[2] and not ctx.needs_input_grad[1]
                                                                          and n * K * 12 >= (1 << 22)) else None
        return colors

    @staticmethod
    def backward(ctx, v_colors: Tensor):
        dirs, coeffs, masks, passed = ctx.saved_tensors
        v_colors = _f32c(v_colors)
        need_dirs = ctx.needs_input_grad[1]
        req = ctx.zeros
        # (the zeros were written on the stream of the rasterization that served the request: only a backward ordered behind it,
        #  i.e. on the same stream, may take them)
        if req is not None and req.buffer is not None and not need_dirs and req.stream == stream_of(dirs):
            v_coeffs, req.buffer = req.buffer, None
            if passed is None:
                call("mtgs_sh_bwd_rows", ctx.n, ctx.K, ctx.degree, ptr(dirs), ptr(masks), ptr(v_colors), ptr(v_coeffs), stream_of(dirs))
            else:
                call("mtgs_sh_bwd_rows_act", ctx.n, ctx.K, ctx.degree, ptr(dirs), ptr(masks), ptr(v_colors), ptr(v_coeffs), ptr(passed),
                     stream_of(dirs))
            return None, None, v_coeffs, None, None
        v_coeffs = torch.empty_like(coeffs)
        v_dirs = torch.empty_like(dirs) if need_dirs else None
        if passed is None:
            call("mtgs_sh_bwd", ctx.n, ctx.K, ctx.degree, ptr(dirs), ptr(coeffs), ptr(masks),
                 ptr(v_colors), ptr(v_coeffs), ptr(v_dirs), stream_of(dirs))
        else:
            call("mtgs_sh_bwd_act", ctx.n, ctx.K, ctx.degree, ptr(dirs), ptr(coeffs), ptr(masks), ptr(v_colors), ptr(v_coeffs),
                 ptr(v_dirs), ptr(passed), stream_of(dirs))
        if not ctx.needs_input_grad[2]:
            v_coeffs = None
        return None, v_dirs, v_coeffs, None, None


# ---- the caller's colour activation, fused without touching the caller ------------------------------------------------------------
# MTGS writes, behind every spherical_harmonics() call,
#     rgbs = spherical_harmonics(n, viewdirs, colors);  rgbs = torch.clamp(rgbs + 0.5, 0.0, 1.0)
# (vanilla_gaussian_splatting.py:317-318, multi_color_gaussian_splatting.py:96, rigid_node.py:248, deformable_node.py:125), gsplat's own
# sh_degree path `clamp_min(colors + 0.5, 0.0)`.  Those two elementwise expressions are six launch-bound kernels per step (forward:
# add, clamp; backward: two compares, and, where -- 67 us of the 0.93 ms headline step, profiles/r05_bench_kernel_stats.csv).
# spherical_harmonics() therefore returns a DEFERRED tensor: nothing has run yet.  `x + c` (a Python scalar) stays deferred; `clamp` /
# `clip` / `clamp_min` / `clamp_max` with scalar bounds on it runs ONE kernel that evaluates SH, adds and clamps (the same fp32
# operations in the same order: bit-identical values) and records torch's clamp-backward mask, and the SH backward applies that mask
# -- one autograd node for the three.  ANY other use (arithmetic, indexing, .sum(), .grad_fn, passing it to a kernel ...)
# materialises the plain SH output once and carries on with an ordinary tensor, so every other program behaves exactly as before.
# The one cost: a program that uses the RAW output a second time besides clamping it (`clamp(x + 0.5, 0, 1)` AND, say, a regulariser on
# `x`) evaluates SH twice -- the fused kernel and a plain one, each with its own backward; MTGS never does (get_rgbs clamps at once).
# `mtgs_amd.sh_lazy(False)` (or MTGS_SH_LAZY=0) switches the deferral off.
def _is_number(v) -> bool:
    return isinstance(v, (int, float)) and not isinstance(v, bool)


class _LazySH(Tensor):
    # four states: the raw SH output (base None), `raw + c` (add), `clamp(raw [+ c], lo, hi)` (act = (lo, hi); base = the state clamped),
    # and torch.cat(dim 0) of clamped states of ONE activation and degree (parts: the nodes of a scene graph, mtgs_scene_graph.py:451-452)
    @staticmethod
    def __new__(cls, degree, dirs, coeffs, masks, base=None, add=None, act=None):
        rg = torch.is_grad_enabled() and (coeffs.requires_grad or dirs.requires_grad)
        r = Tensor._make_wrapper_subclass(cls, dirs.shape, dtype=torch.float32, device=dirs.device, requires_grad=rg)
        r._lz_sh = (degree, dirs, coeffs, masks)
        r._lz_base, r._lz_add, r._lz_act, r._lz_plain, r._lz_parts, r._lz_extra = base, add, act, None, None, None
        # (the evaluation is deferred, its inputs must not be: an in-place change of the directions or coefficients between the call
        #  and the first use would be seen by the deferred kernel and not by PyTorch's evaluation -- caught like autograd catches it)
        r._lz_ver = base._lz_ver if base is not None else (dirs._version, coeffs._version)
        return r

    def _check_inputs(self):
        if self._lz_extra is not None and self._lz_extra._version != self._lz_extra_ver:
            raise RuntimeError("torch.cat([colours, channels], dim=-1) on deferred spherical_harmonics() colours: the channels were modified in "
                               "place before the result was used; clone them, or evaluate at once with mtgs_amd.sh_lazy(False)")
        if self._lz_sh[1] is not None and (self._lz_sh[1]._version, self._lz_sh[2]._version) != self._lz_ver:
            raise RuntimeError("spherical_harmonics(): `dirs` or `coeffs` was modified in place between the call and the first use of its "
                               "(deferred) result; clone the tensor, or evaluate at once with mtgs_amd.sh_lazy(False) / MTGS_SH_LAZY=0")

    @classmethod
    def _cat(cls, parts):
        """torch.cat(parts, dim=0) of deferred activations, still deferred -- or None when the parts are not ONE composition
        rasterization() could take over (then every part is evaluated by the fused kernel and PyTorch concatenates)."""
        first = parts[0]
        for q in parts:
            if not (type(q) is cls and q._lz_act is not None and q._lz_parts is None and q._lz_extra is None and q._lz_plain is None
                    and q._lz_act == first._lz_act
                    and q._lz_sh[0] == first._lz_sh[0] and q.device == first.device):
                return None
        n = sum(int(q.shape[0]) for q in parts)
        r = Tensor._make_wrapper_subclass(cls, (n, 3), dtype=torch.float32, device=first.device,
                                          requires_grad=any(q.requires_grad for q in parts))
        r._lz_sh = (first._lz_sh[0], None, None, None)
        r._lz_base, r._lz_add, r._lz_act, r._lz_plain, r._lz_parts, r._lz_extra = None, None, first._lz_act, None, list(parts), None
        r._lz_ver = None
        return r

    @classmethod
    def _with_extra(cls, inner, extras):
        """torch.cat([deferred colours, further channels ...], dim=-1) -- MTGS's `predict_normals` appends three camera-space normal
        channels behind the colours (mtgs_scene_graph.py:636-638) --, still deferred: rasterization() evaluates the colours for the
        visible Gaussians and takes the further channels as they are; or None (then the colours are evaluated in full and PyTorch
        concatenates)."""
        n = int(inner.shape[0])
        if not (type(inner) is cls and inner._lz_act is not None and inner._lz_plain is None and inner._lz_extra is None and extras):
            return None
        for e in extras:
            if not (type(e) in (Tensor, torch.nn.Parameter) and e.dim() == 2 and e.shape[0] == n and e.dtype == torch.float32
                    and e.device == inner.device and e.shape[1] >= 1):
                return None
        if 3 + sum(int(e.shape[1]) for e in extras) > RECORD_CHANNELS:      # (what a record holds; with a depth channel one fewer: rendering.py)
            return None
        extra = extras[0] if len(extras) == 1 else torch.cat(list(extras), dim=-1)
        r = Tensor._make_wrapper_subclass(cls, (n, 3 + int(extra.shape[1])), dtype=torch.float32, device=inner.device,
                                          requires_grad=bool(inner.requires_grad or (torch.is_grad_enabled() and extra.requires_grad)))
        r._lz_sh = inner._lz_sh
        r._lz_base, r._lz_add, r._lz_act, r._lz_plain, r._lz_parts, r._lz_extra = inner, None, inner._lz_act, None, None, extra
        r._lz_ver = inner._lz_ver
        r._lz_extra_ver = extra._version
        return r

    def _materialize(self) -> Tensor:
        """The ordinary tensor this object stands for (computed once)."""
        if self._lz_plain is None:
            with torch._C.DisableTorchFunctionSubclass():
                if self._lz_extra is not None:
                    self._check_inputs()
                    self._lz_plain = torch.cat([self._lz_base._materialize(), self._lz_extra], dim=-1)
                elif self._lz_parts is not None:
                    self._lz_plain = torch.cat([q._materialize() for q in self._lz_parts], dim=0)
                elif self._lz_act is not None:
                    self._lz_plain = self._lz_base._fused(*self._lz_act)
                elif self._lz_base is not None:
                    self._lz_plain = self._lz_base._materialize() + self._lz_add
                else:
                    self._check_inputs()
                    self._lz_plain = _SphericalHarmonics.apply(*self._lz_sh)
        return self._lz_plain

    def _raster_form(self, lo: float, hi: float) -> bool:
        """clamp(self, lo, hi) is a colour activation rasterization() can evaluate for the VISIBLE Gaussians by itself
        (csrc/viscolor.hip: MTGS's clamp(x + 0.5, 0, 1), gsplat's clamp_min(x + 0.5, 0); K = 16 coefficient rows, given directions): the clamp then stays deferred too, see raster_source()."""
        degree, dirs, coeffs, masks = self._lz_sh
        return (_lazy_raster_enabled and self._lz_add == 0.5 and lo == 0.0 and hi in (1.0, float("inf")) and masks is None
                and degree <= 3 and dirs.dim() == 2 and coeffs.dim() == 3 and coeffs.shape[1] == 16)

    def exchange_source(self, n: int):
        """ColorSource for a data-parallel frame (dist.SparseGradExchange.rasterization: `sh_out` = the RAW SH output of this rank's
        camera, blended as clamp(x + 0.5, 0, 1), its gradient rebuilt by the receivers from v_rgb) when this object is that raw output,
        still deferred: SH + clamp for the visible Gaussians only, no backward of its own -- else None."""
        degree, dirs, coeffs, masks = self._lz_sh
        if not (_lazy_raster_enabled and self._lz_base is None and self._lz_act is None and self._lz_plain is None and masks is None
                and degree <= 3 and dirs.dim() == 2 and dirs.shape == (n, 3) and n > 0 and coeffs.dim() == 3 and coeffs.shape[1] == 16):
            return None
        from .nodes import sh_direction_source
        self._check_inputs()
        cs = sh_direction_source(coeffs, degree, dirs, 1)
        cs.autograd, cs.exchange = False, True
        return cs

    def raster_source(self, n: int, width: int, height: int):
        """(ColorSource, coefficients) for rasterization() when this object is a deferred activation it can take over -- the colours are
        then evaluated for the visible Gaussians only, straight into their records, and d L / d coefficients leaves the rasterization's
        backward (rows of the Gaussians with a cotangent, written into zeros that rode on the compositing forward) -- else None."""
        if self._lz_act is None or self._lz_plain is not None:
            return None
        if self._lz_extra is not None:      # colours + further channels: (source, coefficients, the further channels [N, DX])
            self._check_inputs()
            inner = self._lz_base.raster_source(n, width, height) if tuple(self.shape) == (n, 3 + self._lz_extra.shape[1]) else None
            return None if inner is None else (inner[0], inner[1], self._lz_extra)
        parts = self._lz_parts if self._lz_parts is not None else [self]
        if any(q._lz_plain is not None for q in parts):      # (a node's colours were used elsewhere meanwhile: they exist, concatenate them)
            return None
        if tuple(self.shape) != (n, 3) or n == 0 or not _bin3_ok(1, -(-width // 16), -(-height // 16), 0):
            return None
        from .nodes import sh_direction_source
        for q in parts:
            q._check_inputs()
        coeffs, dirs = [q._lz_sh[2] for q in parts], [q._lz_sh[1] for q in parts]
        dirs_grad = torch.is_grad_enabled() and any(d.requires_grad for d in dirs)      # (MTGS with a camera optimizer: viewdirs carry one)
        if dirs_grad and _graph.caps is not None:
            return None      # (graph mode: the visible list is capacity-sized; the scatter of the direction gradient is not)
        cs = sh_direction_source(coeffs, self._lz_sh[0], dirs, 1 if self._lz_act[1] == 1.0 else 4)
        cs.dirs_inputs = dirs if dirs_grad else None
        return cs, coeffs

    def _fused(self, lo: float, hi: float) -> Tensor:
        self._check_inputs()
        has_add = self._lz_add is not None
        with torch._C.DisableTorchFunctionSubclass():
            return _SphericalHarmonics.apply(*self._lz_sh, (has_add, self._lz_add if has_add else 0.0, lo, hi))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        me = args[0] if args and isinstance(args[0], _LazySH) else None
        name = getattr(func, "__name__", "")
        if func in _LAZY_CAT and args and isinstance(args[0], (list, tuple)) and len(args[0]) >= 1 and not (set(kwargs) - {"dim"}) \
                and (kwargs.get("dim", args[1] if len(args) > 1 else 0) in (0, -2)) and len(args) <= 2 \
                and all(type(q) is _LazySH for q in args[0]):
            got = _LazySH._cat(list(args[0]))
            if got is not None:
                return got
        if func in _LAZY_CAT and args and isinstance(args[0], (list, tuple)) and len(args[0]) >= 2 and not (set(kwargs) - {"dim"}) \
                and (kwargs.get("dim", args[1] if len(args) > 1 else 0) in (1, -1)) and len(args) <= 2 \
                and type(args[0][0]) is _LazySH and not any(isinstance(q, _LazySH) for q in args[0][1:]):
            got = _LazySH._with_extra(args[0][0], list(args[0][1:]))
            if got is not None:
                return got
        if me is not None and me._lz_plain is None:
            if func in _LAZY_META or (name == "__get__" and getattr(func, "__self__", None) in _LAZY_META_PROPS):
                with torch._C.DisableTorchFunctionSubclass():
                    return func(*args, **kwargs)
            # x + c, c + x with a Python scalar: stays deferred (one pending add at most)
            if func in _LAZY_ADD and me._lz_base is None and me._lz_act is None and len(args) == 2 and _is_number(args[1]) and kwargs.get("alpha", 1) == 1 \
                    and not (set(kwargs) - {"alpha"}):
                return _LazySH(*me._lz_sh, base=me, add=float(args[1]))
            # clamp(x, lo, hi) / clip / clamp_min / clamp_max with scalar bounds: the fused kernel
            if func in _LAZY_CLAMP and me._lz_act is None:
                a = list(args[1:])
                lo = kwargs.get("min", a[0] if len(a) > 0 else None)
                hi = kwargs.get("max", a[1] if len(a) > 1 else None)
                if func in _LAZY_CLAMP_MAX:
                    lo, hi = None, kwargs.get("max", a[0] if a else None)
                ok = not (set(kwargs) - {"min", "max"}) and len(a) <= 2 and (lo is None or _is_number(lo)) and (hi is None or _is_number(hi)) \
                    and not (lo is None and hi is None) and not (lo is not None and hi is not None and lo > hi)
                if ok:
                    root = me._lz_base if me._lz_base is not None else me
                    if root._lz_plain is None:
                        lo_f, hi_f = float("-inf") if lo is None else float(lo), float("inf") if hi is None else float(hi)
                        if me._raster_form(lo_f, hi_f):      # still nothing runs: rasterization() may want the visible Gaussians only
                            return _LazySH(*me._lz_sh, base=me, act=(lo_f, hi_f))
                        return me._fused(lo_f, hi_f)
        # everything else: ordinary tensors from here on
        args, kwargs = _lazy_plain(args), _lazy_plain(kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # (a path that went around __torch_function__ -- C++ callers: the same rule, ordinary tensors from here on)
        return func(*_lazy_plain(args), **_lazy_plain(kwargs or {}))


def _lazy_plain(v):
    if isinstance(v, _LazySH):
        return v._materialize()
    if isinstance(v, (list, tuple)):
        return type(v)(_lazy_plain(e) for e in v)
    if isinstance(v, dict):
        return {k: _lazy_plain(e) for k, e in v.items()}
    return v


_LAZY_META = {Tensor.size, Tensor.dim, Tensor.numel, Tensor.nelement, Tensor.ndimension, Tensor.is_floating_point, Tensor.is_complex,
              Tensor.element_size, Tensor.get_device}
_LAZY_META_PROPS = {Tensor.shape, Tensor.dtype, Tensor.device, Tensor.ndim, Tensor.is_cuda, Tensor.layout, Tensor.requires_grad,
                    Tensor.is_sparse, Tensor.is_quantized, Tensor.is_meta}
_LAZY_ADD = {torch.add, Tensor.add, Tensor.__add__, Tensor.__radd__}
_LAZY_CAT = {torch.cat, torch.concat, torch.concatenate}
_LAZY_CLAMP_MAX = {torch.clamp_max, Tensor.clamp_max}
_LAZY_CLAMP = {torch.clamp, Tensor.clamp, torch.clip, Tensor.clip, torch.clamp_min, Tensor.clamp_min} | _LAZY_CLAMP_MAX
_lazy_sh_enabled = os.environ.get("MTGS_SH_LAZY", "1") == "1"
_lazy_raster_enabled = os.environ.get("MTGS_SH_LAZY_RASTER", "1") == "1"      # ... through the clamp into rasterization() (raster_source)


@contextlib.contextmanager
def sh_lazy(enabled: bool = True, raster: Optional[bool] = None):
    """Switches the deferred evaluation of spherical_harmonics() (see _LazySH) for the enclosed calls: False = the function runs its
    kernel at once and returns an ordinary tensor, the caller's `clamp(x + 0.5, ...)` stays PyTorch's.  raster = False: the clamp
    runs the fused SH + activation kernel over ALL Gaussians at once instead of staying deferred for rasterization() (None: as it is)."""
    global _lazy_sh_enabled, _lazy_raster_enabled
    old = (_lazy_sh_enabled, _lazy_raster_enabled)
    _lazy_sh_enabled, _lazy_raster_enabled = bool(enabled), (_lazy_raster_enabled if raster is None else bool(raster))
    try:
        yield
    finally:
        _lazy_sh_enabled, _lazy_raster_enabled = old


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor,
                        masks: Optional[Tensor] = None) -> Tensor:
    """gsplat.cuda._wrapper.spherical_harmonics: dirs[...,3], coeffs[...,K,3], masks[...] -> [...,3]."""
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    assert dirs.shape[:-1] == coeffs.shape[:-2], (dirs.shape, coeffs.shape)
    assert dirs.shape[-1] == 3, dirs.shape
    assert coeffs.shape[-1] == 3, coeffs.shape
    if masks is not None:
        assert masks.shape == dirs.shape[:-1], masks.shape
    if degrees_to_use > 4:
        raise NotImplementedError("spherical_harmonics: degrees_to_use > 4")
    plain_types = (Tensor, torch.nn.Parameter)      # (not other tensor subclasses: their __torch_function__ has its own plans)
    if _lazy_sh_enabled and type(dirs) in plain_types and type(coeffs) in plain_types and dirs.is_cuda and coeffs.is_cuda \
            and dirs.dtype == torch.float32 and coeffs.dtype == torch.float32 and (masks is None or masks.is_cuda):
        return _LazySH(degrees_to_use, dirs, coeffs, masks)
    return _SphericalHarmonics.apply(degrees_to_use, dirs, coeffs, masks)


# ------------------------------------------------------------------------------------- projection
class _FullyFusedProjection(torch.autograd.Function):
    """fully_fused_projection, optionally fused with gsplat's `opacities.repeat(C,1) * compensations`
    (pass `opacities`; the sixth output is then the effective per-camera opacity)."""

    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, opacities, width, height, eps2d, near_plane,
                far_plane, radius_clip, calc_compensations):
        require_gpu(means, quats, scales, viewmats, Ks, opacities)
        means, quats, scales, viewmats, Ks, opacities = map(_f32c, (means, quats, scales, viewmats, Ks, opacities))
        N, Cn = means.shape[0], viewmats.shape[0]
        dev = means.device
        radii = torch.empty((Cn, N), dtype=torch.int32, device=dev)
        means2d = torch.empty((Cn, N, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((Cn, N), dtype=torch.float32, device=dev)
        conics = torch.empty((Cn, N, 3), dtype=torch.float32, device=dev)
        comps = torch.empty((Cn, N), dtype=torch.float32, device=dev) if calc_compensations else None
        opac_eff = torch.empty((Cn, N), dtype=torch.float32, device=dev) if opacities is not None else None
        call("mtgs_project_fwd", Cn, N, ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks),
             width, height, eps2d, near_plane, far_plane, radius_clip, ptr(opacities), ptr(radii),
             ptr(means2d), ptr(depths), ptr(conics), ptr(comps), ptr(opac_eff), 0, 0, 0, None, stream_of(means))
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii, conics, comps, opacities)
        ctx.width, ctx.height, ctx.eps2d = width, height, eps2d
        ctx.mark_non_differentiable(radii)
        return radii, means2d, depths, conics, comps, opac_eff

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_comps, v_opac_eff):
        means, quats, scales, viewmats, Ks, radii, conics, comps, opacities = ctx.saved_tensors
        N, Cn = means.shape[0], viewmats.shape[0]
        dev = means.device
        zeros = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        # incoming gradients may be views of the compositing backward's interleaved buffer: read them in place
        v_means2d, s_m2d = _strided_rows(zeros(Cn, N, 2) if v_means2d is None else v_means2d, 2)
        v_depths, s_dep = _strided_rows(zeros(Cn, N) if v_depths is None else v_depths, 1)
        v_conics, s_con = _strided_rows(zeros(Cn, N, 3) if v_conics is None else v_conics, 3)
        v_comps, s_cmp = _strided_rows(None if (comps is None or v_comps is None) else v_comps, 1)
        v_opac_eff, s_opa = _strided_rows(None if (opacities is None or v_opac_eff is None) else v_opac_eff, 1)
        v_means = torch.empty_like(means)
        v_quats = torch.empty_like(quats)
        v_scales = torch.empty_like(scales)
        v_viewmats = torch.empty_like(viewmats) if ctx.needs_input_grad[3] else None
        v_opacities = torch.empty_like(opacities) if v_opac_eff is not None else None
        call("mtgs_project_bwd", Cn, N, ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks),
             ctx.width, ctx.height, ctx.eps2d, ptr(radii), ptr(conics), ptr(comps), ptr(opacities),
             ptr(v_means2d), ptr(v_depths), ptr(v_conics), ptr(v_comps), ptr(v_opac_eff), ptr(v_means),
             ptr(v_quats), ptr(v_scales), ptr(v_viewmats), ptr(v_opacities),
             host_i64([s_m2d, s_dep, s_con, s_cmp, s_opa]), None, None, None, 0, None, None, None, None,
             None, 0, None, None, None, None, None, None, None, stream_of(means))
        g = ctx.needs_input_grad
        return (v_means if g[0] else None, v_quats if g[1] else None, v_scales if g[2] else None,
                v_viewmats, None, v_opacities if g[5] else None, None, None, None, None, None, None, None)


def _check_projection_args(means, covars, quats, scales, viewmats, Ks, packed, sparse_grad, camera_model):
    Cn, N = viewmats.size(0), means.size(0)
    assert means.size() == (N, 3), means.size()
    assert viewmats.size() == (Cn, 4, 4), viewmats.size()
    assert Ks.size() == (Cn, 3, 3), Ks.size()
    if covars is not None:
        raise NotImplementedError("fully_fused_projection: covars (pass quats and scales)")
    if packed:
        raise NotImplementedError("fully_fused_projection: packed=True")
    if sparse_grad:
        raise NotImplementedError("fully_fused_projection: sparse_grad=True (requires packed=True)")
    if camera_model != "pinhole":
        raise NotImplementedError(f"fully_fused_projection: camera_model={camera_model!r}")
    assert quats is not None and scales is not None, "quats and scales are required"
    assert quats.size() == (N, 4), quats.size()
    assert scales.size() == (N, 3), scales.size()


def fully_fused_projection(means: Tensor, covars: Optional[Tensor], quats: Optional[Tensor],
                           scales: Optional[Tensor], viewmats: Tensor, Ks: Tensor, width: int,
                           height: int, eps2d: float = 0.3, near_plane: float = 0.01,
                           far_plane: float = 1e10, radius_clip: float = 0.0, packed: bool = False,
                           sparse_grad: bool = False, calc_compensations: bool = False,
                           camera_model: str = "pinhole"):
    """gsplat.cuda._wrapper.fully_fused_projection (packed=False, pinhole).
    Returns (radii[C,N] i32, means2d[C,N,2], depths[C,N], conics[C,N,3], compensations[C,N] | None)."""
    _check_projection_args(means, covars, quats, scales, viewmats, Ks, packed, sparse_grad, camera_model)
    return _FullyFusedProjection.apply(means, quats, scales, viewmats, Ks, None, int(width), int(height),
                                       float(eps2d), float(near_plane), float(far_plane),
                                       float(radius_clip), bool(calc_compensations))[:5]


def projection_with_opacities(means, quats, scales, viewmats, Ks, opacities, width, height, eps2d=0.3,
                              near_plane=0.01, far_plane=1e10, radius_clip=0.0, calc_compensations=False):
    """fully_fused_projection + `opacities.repeat(C,1) [* compensations]` in one kernel (what
    gsplat.rendering.rasterization does right after the projection).  Returns the five projection
    outputs and opacities_eff[C,N]."""
    _check_projection_args(means, None, quats, scales, viewmats, Ks, False, False, "pinhole")
    assert opacities.shape == (means.shape[0],), opacities.shape
    return _FullyFusedProjection.apply(means, quats, scales, viewmats, Ks, opacities, int(width), int(height),
                                       float(eps2d), float(near_plane), float(far_plane),
                                       float(radius_clip), bool(calc_compensations))


# ------------------------------------------------------------------------------------- tiles
def _bit_length(v: int) -> int:
    return int(v).bit_length()


def _ws(nbytes_fn: str, n: int, dev) -> Tuple[Tensor, int]:
    nbytes = C.c_size_t(0)
    call(nbytes_fn, n, C.byref(nbytes))
    return torch.empty(nbytes.value, dtype=torch.uint8, device=dev), nbytes.value


@torch.no_grad()
def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int, tile_width: int,
                tile_height: int, sort: bool = True, packed: bool = False,
                n_cameras: Optional[int] = None, camera_ids: Optional[Tensor] = None,
                gaussian_ids: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """gsplat.cuda._wrapper.isect_tiles (packed=False): (tiles_per_gauss[C,N] i32,
    isect_ids[M] i64, flatten_ids[M] i32), sorted by (camera, tile, depth) when sort=True.

    sort=True runs the depth-ordered binning of csrc/bin.hip (same outputs, bit for bit, as the
    emit-then-sort-46-bits formulation, which sort=False + mtgs_sort_pairs still provides)."""
    if packed:
        raise NotImplementedError("isect_tiles: packed=True")
    require_gpu(means2d, radii, depths)
    Cn, N = means2d.shape[:2]
    assert means2d.shape == (Cn, N, 2), means2d.shape
    assert radii.shape == (Cn, N), radii.shape
    assert depths.shape == (Cn, N), depths.shape
    means2d, depths = _f32c(means2d), _f32c(depths)
    radii = radii.contiguous()
    if radii.dtype != torch.int32:
        radii = radii.to(torch.int32)
    dev, st = means2d.device, stream_of(means2d)
    total = Cn * N
    tiles_per_gauss = torch.empty((Cn, N), dtype=torch.int32, device=dev)
    call("mtgs_isect_count", Cn, N, ptr(means2d), ptr(radii), tile_size, tile_width, tile_height,
         ptr(tiles_per_gauss), st)
    scan_ws, scan_bytes = _ws("mtgs_scan_workspace_bytes", total, dev)
    if not sort:
        cum = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
        m_dev = torch.empty(1, dtype=torch.int64, device=dev)
        call("mtgs_isect_scan", total, ptr(tiles_per_gauss), ptr(cum), ptr(m_dev), ptr(scan_ws), scan_bytes, st)
        M = int(m_dev.item())  # the one host sync of a frame (gsplat does the same)
        isect_ids = torch.empty(M, dtype=torch.int64, device=dev)
        flatten_ids = torch.empty(M, dtype=torch.int32, device=dev)
        if M > 0:
            call("mtgs_isect_emit", Cn, N, ptr(means2d), ptr(radii), ptr(depths), ptr(cum), tile_size,
                 tile_width, tile_height, ptr(isect_ids), ptr(flatten_ids), st)
        return tiles_per_gauss, isect_ids, flatten_ids

    return _bin_depth_ordered(means2d, radii, depths, tiles_per_gauss, scan_ws, scan_bytes, tile_size, tile_width,
                              tile_height)[:3]


_mailboxes = threading.local()


def _host_mailbox():
    """(pinned int64[2] tensor, fresh tag) of the calling thread (train thread and viewer thread each get their own)."""
    mb = getattr(_mailboxes, "t", None)
    if mb is None:
        mb = torch.zeros(2, dtype=torch.int64).pin_memory()
        _mailboxes.t, _mailboxes.view, _mailboxes.tag = mb, mb.numpy(), 0
    _mailboxes.tag += 1
    return mb, _mailboxes.tag


def _wait_mailbox(mailbox, tag, device_totals, limit: int) -> Tuple[int, int]:
    """(n_vis, M) published by mtgs_bin_compact's spine kernel into the thread's pinned mailbox.

    The mailbox is `torch.Tensor.pin_memory()` memory, i.e. hipHostMalloc'ed and host-coherent (fine-grained) on ROCm:
    the kernel's system-scope release store becomes visible without a stream synchronisation.  The wait spins a few
    hundred times (the totals are normally 10-20 us away), then yields the GIL between polls so that the viewer /
    dataloader threads run while the GPU drains a backlog.  A mailbox that never answers (5 s) or answers with
    impossible totals falls back to the synchronising device read."""
    view = _mailboxes.view
    spins, t0 = 0, None
    while view[1] != tag:
        spins += 1
        if spins > 256:
            if t0 is None:
                t0 = time.perf_counter()
            elif time.perf_counter() - t0 > 5.0:
                break
            time.sleep(0)          # release the GIL
    packed = int(view[0]) if view[1] == tag else int(device_totals.item())
    n_vis, M = packed >> 32, packed & 0xFFFFFFFF
    if n_vis > limit or M >= (1 << 31):    # a carry out of the low word (M >= 2^32) would land in n_vis
        packed = int(device_totals.item())
        n_vis, M = packed >> 32, packed & 0xFFFFFFFF
        if n_vis > limit or M >= (1 << 31):
            raise RuntimeError(f"tile binning: {M} intersections / {n_vis} visible of {limit}: intersection count "
                               "must stay below 2^31 (render fewer cameras per call)")
    return n_vis, M


def _bin_depth_ordered(means2d, radii, depths, tiles_per_gauss, scan_ws, scan_bytes, tile_size, tile_width, tile_height,
                       want_rank: bool = False):
    """Depth-ordered binning (csrc/bin.hip) after mtgs_isect_count.  Returns (tiles_per_gauss, isect_ids,
    flatten_ids, offsets, tile_order, vis_ids[n_vis], vis_rank[C*N] | None)."""
    Cn, N = means2d.shape[:2]
    dev, st = means2d.device, stream_of(means2d)
    total = Cn * N
    vis_keys = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    vis_ids = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    vis_rank = torch.empty(max(total, 1), dtype=torch.int32, device=dev) if want_rank else None
    totals = torch.empty(1, dtype=torch.int64, device=dev)
    mailbox, tag = _host_mailbox()
    call("mtgs_bin_compact", Cn, N, ptr(radii), ptr(depths), ptr(tiles_per_gauss), ptr(vis_keys),
         ptr(vis_ids), ptr(vis_rank), ptr(totals), mailbox.data_ptr(), tag, ptr(scan_ws), scan_bytes, st)
    # the one host round trip of a frame: n_vis and M together.  The kernel publishes them to pinned host memory as
    # soon as they are known (one kernel before the compaction ends); polling that word instead of synchronising the
    # stream lets the host enqueue the rest of the frame while the GPU is still busy (27 us per frame otherwise).
    n_vis, M = _wait_mailbox(mailbox, tag, totals, total)
    isect_ids = torch.empty(M, dtype=torch.int64, device=dev)
    flatten_ids = torch.empty(M, dtype=torch.int32, device=dev)
    # depth sort -> scan -> emit -> tile sort (+ isect_ids) -> offsets -> tile schedule, enqueued by ONE call
    # so that the host does not starve the GPU between these short kernels.  The offsets and the tile
    # dispatch order ride along on the returned tensors (picked up by isect_offset_encode /
    # rasterize_to_pixels instead of being recomputed).
    offsets = torch.empty((Cn, tile_height, tile_width), dtype=torch.int32, device=dev)
    order = torch.empty(Cn * tile_height * tile_width, dtype=torch.int32, device=dev)
    nbytes = C.c_size_t(0)
    call("mtgs_bin_workspace_bytes", n_vis, M, C.byref(nbytes))
    bin_ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    call("mtgs_bin_build", Cn, N, n_vis, M, ptr(means2d), ptr(radii), ptr(depths), ptr(tiles_per_gauss),
         ptr(vis_keys), ptr(vis_ids), tile_size, tile_width, tile_height, ptr(isect_ids), ptr(flatten_ids),
         ptr(offsets), ptr(order), ptr(bin_ws), nbytes.value, st)
    offsets._mtgs_tile_order = order
    isect_ids._mtgs_offsets = offsets
    return tiles_per_gauss, isect_ids, flatten_ids, offsets, order, vis_ids[:n_vis], vis_rank


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """gsplat.cuda._wrapper.isect_offset_encode -> offsets[C, tile_height, tile_width] i32."""
    require_gpu(isect_ids)
    cached = getattr(isect_ids, "_mtgs_offsets", None)  # computed together with isect_ids by isect_tiles
    if cached is not None and tuple(cached.shape) == (n_cameras, tile_height, tile_width):
        return cached
    isect_ids = isect_ids.contiguous()
    offsets = torch.empty((n_cameras, tile_height, tile_width), dtype=torch.int32, device=isect_ids.device)
    call("mtgs_isect_offsets", isect_ids.numel(), ptr(isect_ids), n_cameras, tile_width, tile_height,
         ptr(offsets), stream_of(isect_ids))
    return offsets


# ------------------------------------------------------------------------------------- compositing
class _RasterizeToPixels(torch.autograd.Function):
    """rasterize_to_pixels, optionally fused with the depth channel (`depths` blended as the last
    channel instead of torch.cat) and the expected-depth normalisation (`ed`) that gsplat's
    rasterization() applies around it."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, depths, ed, width, height, tile_size,
                isect_offsets, flatten_ids, absgrad):
        require_gpu(means2d, conics, colors, opacities, backgrounds, depths, isect_offsets, flatten_ids)
        m2d, con, col, opa, bg, dep = map(_f32c, (means2d, conics, colors, opacities, backgrounds, depths))
        isect_offsets, flatten_ids = isect_offsets.contiguous(), flatten_ids.contiguous()
        Cn, N = m2d.shape[:2]
        DC = 0 if col is None else col.shape[-1]
        DT = DC + (1 if dep is not None else 0)
        th, tw = isect_offsets.shape[1:]
        dev = m2d.device
        render = torch.empty((Cn, height, width, DT), dtype=torch.float32, device=dev)
        alphas = torch.empty((Cn, height, width, 1), dtype=torch.float32, device=dev)
        last_ids = torch.empty((Cn, height, width), dtype=torch.int32, device=dev)
        M = flatten_ids.numel()
        # longest-list-first dispatch order of the tiles (scheduling aid, does not change results)
        order = getattr(isect_offsets, "_mtgs_tile_order", None)
        if order is None or order.numel() != Cn * th * tw:
            order = torch.empty(Cn * th * tw, dtype=torch.int32, device=dev)
            call("mtgs_tile_schedule", Cn, tw, th, ptr(isect_offsets), M, ptr(order), stream_of(m2d))
        call("mtgs_blend_fwd", Cn, N, DC, ptr(m2d), ptr(con), ptr(col), ptr(opa), ptr(bg), ptr(dep), int(ed),
             width, height, tile_size, tw, th, ptr(isect_offsets), ptr(flatten_ids), M, ptr(render),
             ptr(alphas), ptr(last_ids), ptr(order), stream_of(m2d))
        ctx.save_for_backward(m2d, con, col, opa, bg, dep, isect_offsets, flatten_ids, alphas, last_ids, order,
                              render if ed else None)
        ctx.dims = (width, height, tile_size, tw, th, DC, bool(ed))
        ctx.absgrad = absgrad
        ctx.means2d_ref = means2d  # the tensor MTGS calls .retain_grad() on; .absgrad is set on it
        return render, alphas

    @staticmethod
    def backward(ctx, v_render, v_alphas):
        (m2d, con, col, opa, bg, dep, isect_offsets, flatten_ids, alphas, last_ids, order,
         render) = ctx.saved_tensors
        width, height, tile_size, tw, th, DC, ed = ctx.dims
        Cn, N = m2d.shape[:2]
        CN = Cn * N
        v_render, v_alphas = _f32c(v_render), _f32c(v_alphas)
        # ONE zero-filled, interleaved buffer for every atomically accumulated gradient: rows of RS floats
        #   [xy 2 | |xy| 2 | conic 3 | opacity 1 | colour DC | depth 1 | pad]   (RS = 16 for <= 8 channels)
        # so that the 12 atomics of a (tile, Gaussian) entry land in one 64-byte line (5x cheaper than six
        # dense arrays, include/mtgs_rast.h); the gradients handed to autograd are views of it, which the
        # projection backward reads in place (_strided_rows).
        DT = DC + (1 if dep is not None else 0)
        RS = -(-(8 + DT) // 16) * 16
        G = torch.zeros((Cn, N, RS), dtype=torch.float32, device=m2d.device)
        v_means2d, v_abs_buf, v_conics, v_opacities = G[..., 0:2], G[..., 2:4], G[..., 4:7], G[..., 7]
        v_colors = G[..., 8:8 + DC] if DC else None
        v_depths = G[..., 8 + DC] if dep is not None else None
        v_abs = v_abs_buf if ctx.absgrad else None
        call("mtgs_blend_bwd", Cn, N, DC, ptr(m2d), ptr(con), ptr(col), ptr(opa), ptr(bg), ptr(dep), int(ed),
             width, height, tile_size, tw, th, ptr(isect_offsets), ptr(flatten_ids), flatten_ids.numel(),
             ptr(alphas), ptr(last_ids), ptr(render), ptr(v_render), ptr(v_alphas), ptr(v_means2d), ptr(v_abs),
             ptr(v_conics), ptr(v_colors), ptr(v_depths), ptr(v_opacities), host_i64([RS] * 6), None, ptr(order),
             stream_of(m2d))
        if ctx.absgrad:
            ctx.means2d_ref.absgrad = v_abs
        v_bg = None
        if bg is not None and ctx.needs_input_grad[4]:
            v_bg = (v_render[..., :DC] * (1.0 - alphas)).sum(dim=(1, 2))
        return (v_means2d, v_conics, v_colors, v_opacities, v_bg, v_depths, None, None, None, None, None, None, None)


# ------------------------------------------------------------------------------------- fused path
RECORD_CHANNELS = 8      # blended channels a packed record holds (csrc/raster_rec.hpp)
speculative_sizing = True  # enqueue binning + compositing before the host knows (n_vis, M); see _SizePlan
_force_caps = None       # tests: (cap_vis, cap_M) used for the speculative attempt, to exercise the overflow path
_debug_rows = None       # tests: a dict that the fused backward fills with its compact gradient rows {"G", "vis_ids", "DC"}


class _SizePlan(threading.local):
    """Per thread: the largest (n_vis, M) seen recently for a frame shape (C, N, width, height).

    With a plan, a frame's binning and compositing are enqueued with CAPACITIES (1.5x the recent maxima: the extra
    workgroups exit at once, the extra bytes are never touched) while the front kernels are still running; the kernels read
    the true sizes from device memory.  The host reads the front kernel's mailbox afterwards -- normally it is there
    already -- and repeats the frame with exact sizes in the rare case the capacities were too small."""

    def __init__(self):
        self.seen = {}

    def caps(self, key, total):
        rec = self.seen.get(key)
        if rec is None:
            return None
        n_vis, M = rec
        cap_vis = min(total, max(4096, n_vis + n_vis // 2))
        cap_M = max(1 << 16, M + M // 2)
        return cap_vis, -(-cap_M // (1 << 16)) * (1 << 16)

    def update(self, key, n_vis, M):
        old = self.seen.get(key)
        if old is not None:   # slowly forget a peak
            n_vis, M = max(n_vis, old[0] - old[0] // 16), max(M, old[1] - old[1] // 16)
        self.seen[key] = (n_vis, M)


_size_plan = _SizePlan()


class _ListMode(threading.local):
    """Per thread: which tile lists rasterization() builds (the train thread and the viewer thread of
    render_state_machine.py:142 each have their own; a context manager entered on one never changes the other's frame)."""

    def __init__(self):
        self.tight = os.environ.get("MTGS_TIGHT_LISTS", "0") == "1"


_list_mode = _ListMode()


def lists_are_tight() -> bool:
    """True while the calling thread builds tight tile lists (inside `with mtgs_amd.tight_lists():`)."""
    return bool(_list_mode.tight)


class tight_lists:
    """`with mtgs_amd.tight_lists(): ...` -- OPT-IN extension (trainers; never the default): the tile lists hold only the
    (tile, Gaussian) pairs whose {alpha >= 1/255} ellipse reaches a pixel centre of the tile (mtgs_bin3_build, flag 1).  render /
    alphas / gradients are bit-identical to the default mode (those pairs are skipped pixel by pixel anyway); the meta tensors
    isect_ids / flatten_ids / isect_offsets are ORDERED SUBLISTS of gsplat's: `info["n_listed"]` (int32 device scalar) is the number
    of listed pairs, the offsets are the prefix sums of the tight lists' own lengths, and the tensors keep gsplat's length M with
    the tail [n_listed, M) filled with SENTINELS (flatten_ids -1; isect_ids = last camera | last tile | +inf depth), so that
    gsplat's "the last tile's range ends at flatten_ids.numel()" convention stays safe: this package's rasterize_to_pixels stops
    at the first sentinel, isect_offset_encode of the padded isect_ids attributes the tail to the last tile.
    Thread-local, like exact_lists()."""

    def __init__(self, on: bool = True):
        self.on, self.prev = bool(on), None

    def __enter__(self):
        self.prev, _list_mode.tight = _list_mode.tight, self.on
        return self

    def __exit__(self, *exc):
        _list_mode.tight = self.prev
        return False


class exact_lists(tight_lists):
    """`with mtgs_amd.exact_lists(): ...` -- gsplat's tile lists (every tile of the 3-sigma square of every visible Gaussian:
    isect_ids / flatten_ids / isect_offsets bit-identical to gsplat 1.4.0 isect_tiles + isect_offset_encode).  This is the DEFAULT
    of rasterization(); the context manager pins it inside code that runs under tight_lists() (or MTGS_TIGHT_LISTS=1).
    Thread-local."""

    def __init__(self, on: bool = True):
        super().__init__(not on)


class _GraphState:
    """(Process-wide, not per thread: the backward of a captured iteration runs on autograd's device thread.  The thread that
    entered graph_mode OWNS it: a rasterization() forward from any other thread while it is active -- a viewer thread next to
    a capturing trainer -- is refused by name instead of silently inheriting the graph's capacities.  MTGS serialises the two
    with train_lock, render_state_machine.py:142.)"""
    caps = None       # (cap_vis, cap_M) while graph_mode() is active
    keep = None       # host staging buffers that captured copies read at every replay
    owner = None      # threading.get_ident() of the thread inside graph_mode


_graph = _GraphState()


class graph_mode:
    """`with mtgs_amd.graph_mode(cap_vis, cap_M): ...` -- rasterization() for HIP graph capture (torch.cuda.graph).

    Inside, a frame never talks to the host: binning and compositing run for the given CAPACITIES (visible Gaussians,
    tile intersections) and read the true counts on the device, so the calls can be captured once and replayed with new
    camera / parameter VALUES in the same tensors -- a whole training iteration becomes one graph launch instead of
    ~110 kernel launches, which is what bounds MTGS's 960x540 iteration.
    What changes for the caller while the mode is on:
      * `info["flatten_ids"]`, `info["isect_ids"]` have cap_M entries and only the first M are meaningful; the counts are
        device scalars: `info["n_visible"]`, `info["n_intersections"]` (int64), and `info["overflow"]` (bool) is True
        when a count exceeded its capacity -- the frame is then TRUNCATED (never out of bounds) and must be repeated
        with larger capacities (outside the mode, `rasterization()` does that by itself);
      * host tables that kernels read (node descriptors) are kept alive in `.keep` for the lifetime of the graph.
    Capacities: take them from an eager frame (`info["flatten_ids"].numel()`, `(info["radii"] > 0).sum()`) plus a margin."""

    def __init__(self, cap_vis: int, cap_M: int):
        self.caps, self.keep, self.cursor = (int(cap_vis), int(cap_M)), [], 0

    def __enter__(self):
        if _graph.caps is not None:
            raise RuntimeError("graph_mode is already active")
        _graph.caps, _graph.keep, _graph.owner = self.caps, self, threading.get_ident()
        self.cursor = 0      # every entry walks the same sequence of staging buffers (warm-up pass, then capture)
        return self

    def __exit__(self, *exc):
        _graph.caps = _graph.keep = _graph.owner = None
        return False


def staging_buffer(nbytes: int) -> Tensor:
    """Pinned host staging buffer for a table that a kernel reads.  Outside graph_mode: a fresh one from PyTorch's pinned
    cache.  Inside: the i-th request of a pass gets the i-th buffer of the mode object -- allocated by the warm-up pass
    (pinned allocation is not permitted while a stream is capturing), reused by the capture pass, and kept alive with the
    mode object because the captured copy re-reads it at every replay."""
    gm = _graph.keep
    if gm is None:
        return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    if gm.cursor < len(gm.keep) and gm.keep[gm.cursor].numel() == nbytes:
        buf = gm.keep[gm.cursor]
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        del gm.keep[gm.cursor:]
        gm.keep.append(buf)
    gm.cursor += 1
    return buf


def _bin3_ok(Cn, tw, th, cap_M) -> bool:
    return bool(_lib.load().mtgs_bin3_supported(Cn, tw, th, cap_M))


class _FusedRasterization(torch.autograd.Function):
    """projection -> tile binning -> compositing as ONE autograd node: what gsplat.rendering.rasterization
    chains from fully_fused_projection / isect_tiles / rasterize_to_pixels (same results).

    Forward (csrc/front.hip, bin3.hip, blend.hip): ONE kernel projects, counts tiles, ranks the visible Gaussians and
    writes a packed 64-byte record per visible Gaussian; eleven more sort / emit them into per-tile lists of record
    indices; the compositing kernels stage a candidate with ONE 64-byte gather.  No size is needed on the host to
    enqueue any of it (see _SizePlan).  Being one node also lets the backward keep the compositing gradients in COMPACT
    rows -- one 64-byte row per VISIBLE Gaussian, indexed like the records -- and lets the projection backward emit the
    gradients that leave the rasterizer (colours, means2d for retain_grad(), |means2d| for absgrad) as dense contiguous
    tensors while it reads those rows.  Channel counts above 8, more than 32768 (camera, tile) pairs, 4096 (camera, tile row) pairs or 2^30
    intersections take the earlier gather-based kernels (same results).
    Outputs: render, alphas, radii, means2d, depths, conics, compensations, opacities_eff, tiles_per_gauss,
    isect_ids, flatten_ids, isect_offsets  (the tensors of gsplat's `meta`)."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmats, Ks, backgrounds, width, height, eps2d,
                near_plane, far_plane, radius_clip, calc_compensations, with_depth, expected_depth, absgrad, dp=None, cs=None,
                sh_coeffs=None, campos=None, *sh_more):
        """dp (mtgs_amd.dist.SparseGradExchange | None): data-parallel mode -- `colors` is the SH output x[1,N,3], blended
        as clamp(x + 0.5, 0, 1); the front kernel writes the visibility map of the exchange, and the backward leaves
        the gradients as wire rows in the exchange's send buffer instead of dense tensors (see dist.py).
        cs (mtgs_amd.nodes.ColorSource | None): visibility-first colours -- channels 0..2 of the blended colours are evaluated
        from the nodes' SH coefficients for the VISIBLE Gaussians only (csrc/viscolor.hip), `colors` holds the remaining
        channels ([C,N,DX] or None); the backward leaves the coefficient gradient as compact rows in `cs` (rows, row_of).
        sh_coeffs [N,K,3], campos [3] (with cs.autograd): gsplat's own `sh_degree` call style -- the coefficient gradient is
        expanded to a dense tensor for autograd, and the view directions are differentiable (means, camera position).
        sh_more (with cs.dirs, sh_direction_source): the coefficient tensors of the further nodes, in collected order -- node i's
        gradient is the slice [start_i, start_i + n_i) of the dense one --, then (cs.dirs_inputs) every node's direction tensor when
        the directions carry a gradient: theirs is the slice of a dense [N, 3] one."""
        require_gpu(means, quats, scales, opacities, colors, viewmats, Ks, backgrounds)
        means, quats, scales, opacities, col, viewmats, Ks, bg = map(
            _f32c, (means, quats, scales, opacities, colors, viewmats, Ks, backgrounds))
        N, Cn = means.shape[0], viewmats.shape[0]
        dev, st = means.device, stream_of(means)
        tile_size = 16
        tw, th = -(-width // tile_size), -(-height // tile_size)
        n2c = getattr(cs, "camera_normals", None) if cs is not None else None     # [3,4] camera_to_world: three normal channels
        c_open = 0 if cs is None else (6 if n2c is not None else 3)                 # channels filled for the visible Gaussians
        DC = (0 if col is None else col.shape[-1]) + c_open
        DT = DC + int(with_depth)
        ed = bool(expected_depth)
        total = Cn * N
        radii = torch.empty((Cn, N), dtype=torch.int32, device=dev)
        means2d = torch.empty((Cn, N, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((Cn, N), dtype=torch.float32, device=dev)
        conics = torch.empty((Cn, N, 3), dtype=torch.float32, device=dev)
        comps = torch.empty((Cn, N), dtype=torch.float32, device=dev) if calc_compensations else None
        opac_eff = torch.empty((Cn, N), dtype=torch.float32, device=dev)
        tiles_per_gauss = torch.empty((Cn, N), dtype=torch.int32, device=dev)
        render = torch.empty((Cn, height, width, DT), dtype=torch.float32, device=dev)
        alphas = torch.empty((Cn, height, width, 1), dtype=torch.float32, device=dev)
        last_ids = torch.empty((Cn, height, width), dtype=torch.int32, device=dev)
        packed = total > 0 and 1 <= DT <= RECORD_CHANNELS and _bin3_ok(Cn, tw, th, 0)
        if dp is not None and not (packed and Cn == 1 and DC >= 3 and bg is None and (DC == 3 or dp.rows_hook is not None)):
            raise NotImplementedError("data-parallel rasterization: one camera, the SH output in the first 3 colour channels (further "
                                      "channels need SparseGradExchange.rows_hook to fold their gradient into the wire rows), no "
                                      "backgrounds")
        if cs is not None and not (packed and Cn == 1 and (dp is None or getattr(cs, "exchange", False)) and bg is None):
            raise NotImplementedError("rasterization(color_source=...): one camera, at most 8 blended channels, no backgrounds")
        if not packed:
            # ---- gather-based kernels (csrc/project.hip, bin.hip, blend.hip with dense attribute arrays)
            call("mtgs_project_fwd", Cn, N, ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks),
                 width, height, eps2d, near_plane, far_plane, radius_clip, ptr(opacities), ptr(radii),
                 ptr(means2d), ptr(depths), ptr(conics), ptr(comps), ptr(opac_eff), tile_size, tw, th,
                 ptr(tiles_per_gauss), st)     # (+ the count pass of isect_tiles)
            scan_ws, scan_bytes = _ws("mtgs_scan_workspace_bytes", total, dev)
            _, isect_ids, flatten_ids, offsets, order, vis_ids, vis_rank = _bin_depth_ordered(
                means2d, radii, depths, tiles_per_gauss, scan_ws, scan_bytes, tile_size, tw, th, want_rank=True)
            dep = depths if with_depth else None
            call("mtgs_blend_fwd", Cn, N, DC, ptr(means2d), ptr(conics), ptr(col), ptr(opac_eff), ptr(bg), ptr(dep), int(ed),
                 width, height, tile_size, tw, th, ptr(offsets), ptr(flatten_ids), flatten_ids.numel(), ptr(render),
                 ptr(alphas), ptr(last_ids), ptr(order), st)
            recs = rank_ids = None
        else:
            # ---- packed records
            key = (Cn, N, width, height)
            front_ws, front_bytes = _ws("mtgs_front_workspace_bytes", total, dev)
            # (graph mode: {packed, n_vis, M, overflow} -- words 1..3 are written by the binning, MTGS_BIN3_STATUS)
            totals = torch.empty(4 if _graph.caps is not None else 1, dtype=torch.int64, device=dev)
            vis_rank = torch.empty(total, dtype=torch.int32, device=dev)
            offsets_buf = torch.empty(Cn * th * tw + 1, dtype=torch.int32, device=dev)
            order = torch.empty(Cn * th * tw, dtype=torch.int32, device=dev)

            graph_caps = _graph.caps
            if graph_caps is not None and _graph.owner != threading.get_ident():
                raise RuntimeError("rasterization(): mtgs_amd.graph_mode is active on another thread (fixed capacities, no host "
                                   "read-back); serialise the two callers or leave graph_mode first")

            # TOUCH FIRST (ColorSource.touch_first, opt-in): the binning needs the records' geometry only, so it runs in front of the
            # colours, and one pass of the compositing DECISIONS (mtgs_blend_touch_packed) flags the Gaussians the frame composites
            # from -- a few percent of the visible ones in an opaque scene.  Peek, SH evaluation and normals work on those alone.
            touch_first = cs is not None and bool(cs.touch_first)
            # tile lists of this frame (thread-local mode, read once): gsplat's by default; mtgs_bin3_build flags: 1 = tight lists,
            # 2 = sentinel-fill the tail [n_listed, M) of flatten_ids / isect_ids (tight lists, tensors sliced to gsplat's M),
            # 4 = sentinel-fill up to the capacity (graph mode: the tensors are capacity-sized in both list modes), 16 = the frame's
            # counts and its overflow flag as device words behind `totals` (graph mode)
            tight = lists_are_tight()
            list_flags = (1 if tight else 0) | (4 | 16 if graph_caps is not None else (2 if tight else 0))

            def colours(b, flags):   # colours of the visible Gaussians, straight into their records
                cap_vis = b["cap_vis"]
                coef = cs.prepare(vis_rank, cap_vis, b["vis_ids"], totals, row_flags=flags)    # (row-lazy optimizer: up-to-date coefficient rows, compact)
                call("mtgs_vis_color_fwd_dirs", cs.n_nodes, ptr(cs.table), cs.degree, ptr(cs.cam), ptr(means), ptr(b["vis_ids"]),
                     ptr(totals), cap_vis, ptr(b["recs"]), ptr(b["vis_mask"]), ptr(coef), 0 if coef is None else coef.stride(0),
                     ptr(flags), ptr(cs.dirs), st)
                if n2c is not None:  # ... and their camera-space normals (channels 3..5)
                    call("mtgs_normals_fwd_rows", cap_vis, ptr(b["vis_ids"]), ptr(totals), ptr(quats), ptr(scales), ptr(means),
                         ptr(n2c), ptr(b["recs"]), 3, ptr(flags), st)

            def bin_ws(cap_vis, cap_M):      # the binning's workspace, 256-byte aligned: (tensor, pointer, bytes, bytes of its control words)
                nbytes, cbytes = C.c_size_t(0), C.c_size_t(0)
                call("mtgs_bin3_workspace_bytes", Cn, tw, th, cap_vis, cap_M, C.byref(nbytes))
                call("mtgs_bin3_control_bytes", Cn, tw, th, C.byref(cbytes))
                ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=dev)
                return ws, ws.data_ptr() + (-ws.data_ptr()) % 256, nbytes.value, cbytes.value

            def front(cap_vis, repeat=False, prezero=None):
                # prezero = bin_ws(...) of the binning that follows: its control words are cleared by the compaction kernel here
                # repeat=True: the capacity-overflow repeat of a frame.  The visibility map / row count of the exchange do not
                # depend on cap_vis and are already on their way: they are neither rewritten nor gathered a second time
                # (a second meta all-gather on ONE rank would desynchronise the ranks' collectives).
                dpf = dp if not repeat else None
                b = {"recs": torch.empty((cap_vis, 16), dtype=torch.float32, device=dev),
                     "vis_ids": torch.empty(cap_vis, dtype=torch.int32, device=dev),
                     "vis_keys": torch.empty(cap_vis, dtype=torch.int64, device=dev), "cap_vis": cap_vis,
                     "vis_mask": torch.empty(cap_vis, dtype=torch.uint8, device=dev) if cs is not None else None}
                mailbox, tag = _host_mailbox() if graph_caps is None else (None, 0)
                call("mtgs_front_fwd", Cn, N, ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), width, height,
                     eps2d, near_plane, far_plane, radius_clip, ptr(opacities), ptr(col), DC, int(with_depth), ptr(radii),
                     ptr(means2d), ptr(depths), ptr(conics), ptr(comps), ptr(opac_eff), tile_size, tw, th,
                     ptr(tiles_per_gauss), ptr(b["recs"]), ptr(b["vis_ids"]), ptr(b["vis_keys"]),
                     ptr(vis_rank), cap_vis, *(dpf.front_pointers() if dpf is not None else (None, None, None)),
                     (1 if dp is not None else 0) if cs is None else (3 if c_open == 6 else 2), ptr(totals),
                     None if mailbox is None else mailbox.data_ptr(), tag,
                     None if prezero is None else prezero[1], 0 if prezero is None else prezero[3],
                     ptr(front_ws), front_bytes, st)
                if cs is not None and not touch_first:
                    colours(b, None)
                b["mailbox"], b["tag"] = mailbox, tag
                if dpf is not None:
                    dpf.after_front()      # the visibility maps travel while this frame is composited
                return b

            ctx_box = {}

            def rest(b, cap_M, prezeroed=None):
                cap_alloc = max(cap_M, 1)
                out = {"rank_ids": torch.empty(cap_alloc, dtype=torch.int32, device=dev),
                       "flatten_ids": torch.empty(cap_alloc, dtype=torch.int32, device=dev),
                       "isect_ids": torch.empty(cap_alloc, dtype=torch.int64, device=dev)}
                ws, ws_ptr, ws_bytes, _ = prezeroed if prezeroed is not None else bin_ws(b["cap_vis"], cap_M)
                call("mtgs_bin3_build", Cn, N, tile_size, tw, th, ptr(totals), b["cap_vis"], cap_M, ptr(b["recs"]),
                     ptr(b["vis_ids"]), ptr(b["vis_keys"]), ptr(out["rank_ids"]),
                     ptr(out["flatten_ids"]), ptr(out["isect_ids"]), ptr(offsets_buf), ptr(order),
                     list_flags | (8 if prezeroed is not None else 0), ws_ptr, ws_bytes, st)
                if touch_first:
                    flags = torch.empty(max(b["cap_vis"], 1), dtype=torch.uint8, device=dev)
                    call("mtgs_blend_touch_packed", Cn, ptr(b["recs"]), width, height, tw, th, ptr(offsets_buf), ptr(out["rank_ids"]),
                         ptr(order), ptr(flags), b["cap_vis"], st)
                    colours(b, flags)
                # a training forward: its compositing kernel (VALU-bound, HBM mostly idle) clears what the backward pass will want
                # zeroed -- the gradient rows of the compositing backward, the dL/dcoeffs of the SH backwards (_Prefill)
                z_ptr, z_bytes = None, 0
                if _prefill.enabled and _prefill.in_forward and any(ctx.needs_input_grad) and dp is None and ctx_box.get("rows") is None:
                    RS_ = -(-(8 + DC + int(with_depth)) // 16) * 16
                    n_rows_ = max(b["cap_vis"], 1) * RS_
                    # (gsplat's sh_degree call style: the dense [N, 16, 3] coefficient gradient too -- the backward then writes the
                    #  rows of the Gaussians with a cotangent straight into it, mtgs_vis_color_bwd(dense_rows))
                    n_coef_ = N * 48 if (cs is not None and cs.autograd and cs.width == 48 and (cs.n_nodes == 1 or cs.dirs is not None)
                                         and (graph_caps is None or cs.dirs is not None)
                                         and (ctx.needs_input_grad[20] or any(ctx.needs_input_grad[22:22 + cs.n_nodes - 1]))) else 0
                    z_ptr, z_bytes, own_ = _prefill.take(dev, n_rows_ + n_coef_, only=getattr(_sh_scope, "reqs", ()))
                    ctx_box["rows"] = own_[:n_rows_].view(max(b["cap_vis"], 1), RS_)
                    if n_coef_:
                        ctx_box["coeffs"] = own_[n_rows_:].view(N, 16, 3)
                elif dp is not None and getattr(dp, "zero_region", None) is not None and any(ctx.needs_input_grad):
                    # data-parallel frame: the dense sums the exchange will return (SparseGradExchange.prezero) are cleared here too
                    z_ptr, z_bytes = dp.zero_region
                call("mtgs_blend_fwd_packed", Cn, DC, int(with_depth), ptr(b["recs"]), ptr(bg), int(ed), width, height, tw, th,
                     ptr(offsets_buf), ptr(out["rank_ids"]), ptr(render), ptr(alphas), ptr(last_ids), ptr(order), z_ptr, z_bytes, st)
                return out

            caps = _force_caps or (_size_plan.caps(key, total) if speculative_sizing else None)
            if caps is not None and not _bin3_ok(Cn, tw, th, caps[1]):
                caps = None
            if graph_caps is not None:
                # graph mode: fixed capacities, nothing waits for the host; the counts stay on the device
                if dp is not None or not _bin3_ok(Cn, tw, th, graph_caps[1]):
                    raise NotImplementedError("graph_mode: unsupported configuration (data-parallel exchange / capacity >= 2^30)")
                pz = bin_ws(min(graph_caps[0], total), graph_caps[1])
                b = front(min(graph_caps[0], total), prezero=pz)
                out = rest(b, graph_caps[1], prezeroed=pz)
                n_vis, M = b["cap_vis"], graph_caps[1]
            elif caps is not None:
                pz = bin_ws(min(caps[0], total), caps[1])
                b = front(min(caps[0], total), prezero=pz)
                out = rest(b, caps[1], prezeroed=pz)      # enqueued before the totals are known
                n_vis, M = _wait_mailbox(b["mailbox"], b["tag"], totals, total)
                if n_vis > b["cap_vis"] or M > caps[1]:   # capacities too small: repeat with exact sizes
                    if not _bin3_ok(Cn, tw, th, M):
                        raise NotImplementedError(f"rasterization: {M} tile intersections in one call (limit 2^30)")
                    if n_vis > b["cap_vis"]:
                        b = front(n_vis, repeat=True)
                    out = rest(b, M)
            else:
                b = front(total)
                n_vis, M = _wait_mailbox(b["mailbox"], b["tag"], totals, total)
                if not _bin3_ok(Cn, tw, th, M):
                    raise NotImplementedError(f"rasterization: {M} tile intersections in one call (limit 2^30)")
                b["cap_vis"] = max(n_vis, 0)              # (buffers are larger; the kernels only need a bound)
                out = rest(b, M)
            if graph_caps is None:
                _size_plan.update(key, n_vis, M)
            recs, vis_ids = b["recs"], b["vis_ids"][:n_vis]
            rank_ids, flatten_ids, isect_ids = out["rank_ids"][:M], out["flatten_ids"][:M], out["isect_ids"][:M]
            offsets = offsets_buf[:Cn * th * tw].view(Cn, th, tw)
            offsets._mtgs_tile_order = order
            if tight or graph_caps is not None:
                offsets._mtgs_n_listed = offsets_buf[Cn * th * tw]     # (device scalar: the number of pairs in the lists)
            isect_ids._mtgs_offsets = offsets
        if not packed:
            totals = torch.zeros(1, dtype=torch.int64, device=dev)
        ctx.save_for_backward(means, quats, scales, opacities, col, viewmats, Ks, bg, radii, means2d, depths, conics,
                              comps, opac_eff, offsets if not packed else offsets_buf, flatten_ids, alphas, last_ids, order,
                              vis_ids, vis_rank, render if ed else None, recs, rank_ids, totals)
        ctx.cs, ctx.vis_mask, ctx.cap_vis = cs, (b["vis_mask"] if cs is not None else None), (b["cap_vis"] if packed else 0)
        ctx.zero_rows = ctx_box.get("rows") if packed else None      # gradient rows the forward's compositing kernel cleared
        ctx.sh_reqs = [weakref.ref(r) for r in getattr(_sh_scope, "reqs", ())]      # (what the forward did not serve, the backward may)
        ctx.zero_coeffs = ctx_box.get("coeffs") if packed else None
        ctx.n2c = n2c
        ctx.graph = packed and _graph.caps is not None
        ctx.dims = (width, height, tile_size, tw, th, DC, bool(with_depth), ed, float(eps2d))
        ctx.absgrad = bool(absgrad)
        ctx.packed = packed
        ctx.dp = dp
        ctx.set_materialize_grads(False)
        # classic mode: an empty placeholder keeps the output arity fixed
        comps_out = comps if comps is not None else torch.empty(0, device=dev)
        nd = [radii, tiles_per_gauss, isect_ids, flatten_ids, offsets, totals] + ([comps_out] if comps is None else [])
        ctx.mark_non_differentiable(*nd)
        return (render, alphas, radii, means2d, depths, conics, comps_out, opac_eff, tiles_per_gauss, isect_ids,
                flatten_ids, offsets, totals)

    @staticmethod
    def backward(ctx, v_render, v_alphas, _r, g_means2d, g_depths, g_conics, g_comps, g_opac, *_ints):
        (means, quats, scales, opacities, col, viewmats, Ks, bg, radii, means2d, depths, conics, comps, opac_eff, offsets,
         flatten_ids, alphas, last_ids, order, vis_ids, vis_rank, render, recs, rank_ids, totals) = ctx.saved_tensors
        width, height, tile_size, tw, th, DC, with_depth, ed, eps2d = ctx.dims
        Cn, N = means2d.shape[:2]
        dev, st = means.device, stream_of(means)
        n_vis = vis_ids.numel()
        DT = DC + int(with_depth)
        dep = depths if with_depth else None
        # compact gradient rows, one per VISIBLE (camera, Gaussian) pair:
        #   [xy 2 | |xy| 2 | conic 3 | opacity 1 | colour DC | depth 1 | pad]
        RS = -(-(8 + DT) // 16) * 16
        G = getattr(ctx, "zero_rows", None)
        if G is not None and G.shape[1] == RS and G.shape[0] >= max(n_vis, 1):
            ctx.zero_rows = None            # (cleared by the forward's compositing kernel; a second backward of the node takes the else)
            G = G[:max(n_vis, 1)]
        else:
            G = torch.zeros((max(n_vis, 1), RS), dtype=torch.float32, device=dev)
        r_xy, r_abs, r_con, r_opa = G[:, 0:2], G[:, 2:4], G[:, 4:7], G[:, 7]
        r_col = G[:, 8:8 + DC] if DC else None
        r_dep = G[:, 8 + DC] if with_depth else None
        zeroed_out = None
        if v_render is not None or v_alphas is not None:
            if v_render is None:
                v_render = torch.zeros((Cn, height, width, DT), dtype=torch.float32, device=dev)
            if v_alphas is None:
                v_alphas = torch.zeros((Cn, height, width, 1), dtype=torch.float32, device=dev)
            v_render, v_alphas = _f32c(v_render), _f32c(v_alphas)
            if ctx.packed:
                if rank_ids.numel() > 0:
                    # (the zeros the spherical_harmonics() backwards of this pass want are written by this kernel: _Prefill)
                    # ... and the DENSE gradients this node returns (v_means, v_quats, v_scales, v_opacities, means2d.grad, .absgrad,
                    # extra colour channels: zeros for the ~94 % of the Gaussians without a gradient): the compositing backward is
                    # VALU-bound and clears them beside its own work, the projection backward then writes the rows that have a gradient
                    # to their places and its streaming pass over all N does not run (mtgs_project_bwd_zeroed)
                    zplan = _zeroed_outputs_plan(ctx, Cn, N, n_vis, DC, g_means2d, g_depths, g_conics, g_comps, g_opac)
                    z_ptr, z_bytes, z_own = _prefill.take(dev, 0 if zplan is None else zplan["floats"],
                                                          only=[r for r in (w() for w in getattr(ctx, "sh_reqs", ())) if r is not None])
                    if zplan is not None:
                        zeroed_out = {k: z_own[o:o + n_].view(shape) for k, (o, n_, shape) in zplan["views"].items()}
                    call("mtgs_blend_bwd_packed", Cn, DC, int(with_depth), ptr(recs), ptr(bg), int(ed), width, height, tw, th,
                         ptr(offsets), ptr(rank_ids), ptr(alphas), ptr(last_ids), ptr(render), ptr(v_render), ptr(v_alphas),
                         ptr(G), RS, int(ctx.absgrad), ptr(order), z_ptr, z_bytes, st)
            else:
                call("mtgs_blend_bwd", Cn, N, DC, ptr(means2d), ptr(conics), ptr(col), ptr(opac_eff), ptr(bg), ptr(dep), int(ed),
                     width, height, tile_size, tw, th, ptr(offsets), ptr(flatten_ids), flatten_ids.numel(), ptr(alphas),
                     ptr(last_ids), ptr(render), ptr(v_render), ptr(v_alphas), ptr(r_xy),
                     ptr(r_abs) if ctx.absgrad else None, ptr(r_con), ptr(r_col), ptr(r_dep), ptr(r_opa),
                     host_i64([RS] * 6), ptr(vis_rank), ptr(order), st)
        # the packed compositing backward leaves RAW MOMENT rows (blend.hip / include/mtgs_rast.h): their consumer -- the
        # projection backward, per visible Gaussian -- converts them in place to {v_xy, |v_xy|, v_conic, v_opacity_eff}
        raw = bool(ctx.packed and rank_ids.numel() > 0 and (v_render is not None or v_alphas is not None))
        if _debug_rows is not None:
            _debug_rows.update(G=G, vis_ids=vis_ids, DC=DC, with_depth=with_depth)
        cs = ctx.cs
        if cs is not None and not getattr(cs, "exchange", False):      # (data-parallel frames: the colour gradient travels as v_rgb in the wire rows)
            # visibility-first colours: d L / d (SH coefficients) of the VISIBLE Gaussians as 192-byte rows; the optimizer takes
            # them through the row map (vis_rank: rank or -1) -- no dense [N, (T,) K, 3] gradient is written
            dense_coeffs = getattr(ctx, "zero_coeffs", None) if cs.autograd else None      # (zeroed by the forward's compositing kernel)
            ctx.zero_coeffs = None
            n_c = 22 + cs.n_nodes - 1      # (inputs 20, 22 .. n_c - 1: the nodes' coefficient tensors; n_c ..: their direction tensors, if given)
            need_coef = bool(ctx.needs_input_grad[20] or any(ctx.needs_input_grad[22:n_c]))
            need_dirs = bool(getattr(cs, "dirs_inputs", None)) and any(ctx.needs_input_grad[n_c:])
            want_dirs = cs.autograd and (cs.dirs is None or need_dirs)      # (given directions carry a gradient only with a camera optimizer)
            feat = dir_rows = dir_part = None
            if not (cs.autograd and cs.dirs is not None and not (need_coef or need_dirs)):      # (frozen coefficients: nothing to do)
                feat = torch.empty((max(n_vis, 1), 48), dtype=torch.float32, device=dev) if dense_coeffs is None else None
                dir_rows = torch.empty((max(n_vis, 1), 3), dtype=torch.float32, device=dev) if want_dirs else None
                dir_part = torch.zeros((-(-max(n_vis, 1) // 64), 3), dtype=torch.float32, device=dev) if want_dirs else None      # (MTGS_VIS_COLOR_ROWS)
                call("mtgs_vis_color_bwd_dirs", cs.n_nodes, ptr(cs.table), cs.degree, ptr(cs.cam), ptr(means), ptr(vis_ids), ptr(totals),
                     n_vis, ptr(G), RS, 8, ptr(recs), ptr(ctx.vis_mask), ptr(feat), ptr(dir_rows), ptr(dir_part), ptr(dense_coeffs),
                     ptr(cs.dirs), st)
            cs.rows, cs.row_of = feat, vis_rank
        if ctx.dp is not None:
            # data-parallel mode: the per-visible VJP writes this rank's wire rows (index order) into the exchange's send
            # buffer; dense gradients are rebuilt for all ranks at once by SparseGradExchange.finish()
            if any(g is not None for g in (g_means2d, g_depths, g_conics, g_comps, g_opac)):
                raise NotImplementedError("data-parallel rasterization: gradients on info[...] tensors")
            v_viewmats = torch.empty_like(viewmats) if ctx.needs_input_grad[5] else None
            call("mtgs_project_bwd_rows", N, ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), width, height, eps2d,
                 ptr(conics), ptr(comps), ptr(opacities), ptr(G), RS, DC, int(with_depth),
                 *((ptr(col), 1) if cs is None else (ptr(ctx.vis_mask), 2)),      # (the clamp's pass-through rule: from x, or from the colour kernel's bits)
                 ptr(vis_ids), n_vis, ptr(ctx.dp.rows), ptr(v_viewmats), int(raw), st)
            if ctx.dp.rows_hook is not None:    # camera-dependent extra channels (normals): their VJP goes into the rows here
                ctx.dp.rows_hook(G, RS, vis_ids, n_vis)
            ctx.dp.after_backward(n_vis, G, vis_ids)
            return (None, None, None, None, None, v_viewmats, None, None) + (None,) * (len(ctx.needs_input_grad) - 8)
        # gradients that reached the projection outputs directly (losses on info["means2d"] / ["depths"] / ...):
        # added to the visible rows (culled pairs have no gradient path in gsplat either)
        direct = [g for g in (g_means2d, g_conics, g_opac, g_depths, g_comps) if g is not None]
        if direct and ctx.graph:
            raise NotImplementedError("graph_mode: gradients on info[...] tensors (the visible list is capacity-sized)")
        if raw and n_vis > 0 and (direct or Cn != 1):
            # (rare: a loss on info[...] tensors, or several cameras -- the generic projection backward: the rows are brought to
            #  their documented meaning by a few tensor operations first)
            o = recs[:n_vis, 5:6]
            ca, cb, cc = recs[:n_vis, 2:3], recs[:n_vis, 3:4], recs[:n_vis, 4:5]
            m1, m2 = G[:n_vis, 0:1].clone(), G[:n_vis, 1:2].clone()
            G[:n_vis, 0:1] = -o * (ca * m1 + cb * m2)
            G[:n_vis, 1:2] = -o * (cb * m1 + cc * m2)
            G[:n_vis, 2:4] *= o * 1.38629436111989061883      # (1 / (log2(e)/2): the factor the staged conic carries, include/mtgs_rast.h)
            G[:n_vis, 4:7] *= -o * G.new_tensor([0.5, 1.0, 0.5])
            raw = False
        if n_vis > 0 and direct:
            vi = vis_ids.long()
            if g_means2d is not None:
                r_xy += g_means2d.reshape(Cn * N, 2)[vi]
            if g_conics is not None:
                r_con += g_conics.reshape(Cn * N, 3)[vi]
            if g_opac is not None:
                r_opa += g_opac.reshape(Cn * N)[vi]
        r_dep_total = r_dep
        if g_depths is not None and n_vis > 0:
            r_dep_total = (r_dep if r_dep is not None else 0) + g_depths.reshape(Cn * N)[vi]
        if r_dep_total is None:
            r_dep_total = torch.zeros(max(n_vis, 1), dtype=torch.float32, device=dev)
        r_cmp = None
        if comps is not None and g_comps is not None and n_vis > 0:
            r_cmp = g_comps.reshape(Cn * N)[vi].contiguous()
        need = ctx.needs_input_grad
        geo_rows = bool(cs is not None and getattr(cs, "geometry_rows", False) and not direct and n_vis > 0)
        if geo_rows and (any(g[4] for g in cs.node_geometry) or cs.autograd):
            raise NotImplementedError("ColorSource.geometry_rows: static nodes only (a rigid node's pose gradient is a sum over its Gaussians)")
        if zeroed_out is not None and (geo_rows or direct or not raw):
            zeroed_out = None      # (cannot happen: _zeroed_outputs_plan tests the same conditions)
        Z = zeroed_out or {}
        v_means = None if geo_rows else Z.get("means", None) if Z else torch.empty_like(means)
        v_quats = None if geo_rows else Z.get("quats", None) if Z else torch.empty_like(quats)
        v_scales = None if geo_rows else Z.get("scales", None) if Z else torch.empty_like(scales)
        v_opacities = None if geo_rows else Z.get("opacities", None) if Z else torch.empty_like(opacities)
        v_viewmats = torch.empty_like(viewmats) if need[5] else None
        m2d_out = ctx.means2d_ref() if getattr(ctx, "means2d_ref", None) is not None else None
        want_m2d = m2d_out is not None and m2d_out.retains_grad
        rows_only = cs is not None and getattr(cs, "want_grad_rows", False)
        if geo_rows and not rows_only:
            raise NotImplementedError("ColorSource.geometry_rows needs want_grad_rows (no dense by-products of the projection backward)")
        if rows_only:   # the caller takes the 2-D gradients from the compact rows (densify.update_statistics_rows): no dense absgrad
            cs.grad_rows, cs.grad_row_ids, cs.grad_row_count = G, vis_ids, (totals if ctx.graph else None)
        d_m2d = (Z["m2d"] if Z else torch.empty((Cn, N, 2), dtype=torch.float32, device=dev)) if want_m2d else None
        d_abs = (Z["abs"] if Z else torch.empty((Cn, N, 2), dtype=torch.float32, device=dev)) if (ctx.absgrad and m2d_out is not None and not rows_only) else None
        c0 = 0 if cs is None else (6 if ctx.n2c is not None else 3)   # (the dense colour gradient covers the other channels only)
        q_rows = None
        if cs is not None and ctx.n2c is not None:   # the normals' gradient: quaternion rows of the visible Gaussians
            q_rows = torch.empty((max(n_vis, 1), 4), dtype=torch.float32, device=dev)
            call("mtgs_normals_bwd_qrows", n_vis, ptr(vis_ids), ptr(totals) if ctx.graph else None, ptr(quats), ptr(scales), ptr(means),
                 ptr(ctx.n2c), ptr(G), RS, 8 + 3, ptr(q_rows), st)
        d_col = (Z["col"] if Z else torch.empty((Cn, N, DC - c0), dtype=torch.float32, device=dev)) if (DC - c0 and need[4]) else None
        if geo_rows and (d_col is not None or want_m2d):
            raise NotImplementedError("ColorSource.geometry_rows: no extra colour channels with a gradient, no retain_grad() on means2d")
        vis_ws = torch.empty((max(n_vis, 1), 12), dtype=torch.float32, device=dev)  # scratch of the compact VJP
        vm_part = None      # ... and of the camera gradient: the workgroups' partial sums (added up in a fixed order, no atomics)
        if v_viewmats is not None and Cn == 1 and n_vis > 0:
            nb = C.c_int64(0)
            call("mtgs_project_bwd_blocks", n_vis, C.byref(nb))
            vm_part = torch.empty(nb.value * 12, dtype=torch.float32, device=dev)
        call("mtgs_project_bwd_zeroed" if Z else "mtgs_project_bwd", Cn, N, ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks), width, height,
             eps2d, ptr(radii), ptr(conics), ptr(comps), ptr(opacities), ptr(r_xy), ptr(r_dep_total), ptr(r_con),
             ptr(r_cmp), ptr(r_opa), ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_viewmats), ptr(v_opacities),
             host_i64([RS, r_dep_total.stride(0), RS, 1, RS]), ptr(vis_rank), ptr(r_abs),
             None if d_col is None else G.data_ptr() + 4 * (8 + c0), DC - c0 if d_col is not None else 0,
             host_i64([RS, RS]), ptr(d_m2d), ptr(d_abs), ptr(d_col), ptr(vis_ids), n_vis, ptr(vis_ws),
             ptr(totals) if ctx.graph else None, ptr(q_rows),
             ptr(dir_rows) if (cs is not None and cs.autograd and cs.dirs is None and n_vis > 0) else None,      # (differentiable view directions: dirs = means - camera position)
             ptr(G) if raw else None, ptr(recs) if (raw and Cn == 1) else None, ptr(vm_part), st)
        d_coeffs = d_campos = None
        d_more = (None,) * max(len(ctx.needs_input_grad) - 22, 0)
        if cs is not None and cs.autograd:
            if ctx.graph and cs.dirs is None:
                raise NotImplementedError("graph_mode: rasterization(sh_degree=...) (dense coefficient gradient)")
            K3 = cs.width
            if dense_coeffs is not None:
                d_coeffs = dense_coeffs
            elif feat is not None and (cs.dirs is None or need_coef):
                d_coeffs = torch.empty((N, K3 // 3, 3), dtype=torch.float32, device=dev)
                call("mtgs_rows_expand", N, K3, ptr(vis_rank), ptr(feat), 48, ptr(d_coeffs), st)
            if cs.dirs is not None:
                d_campos = None
                n_c = 22 + cs.n_nodes - 1
                need_c = [ctx.needs_input_grad[20]] + list(ctx.needs_input_grad[22:n_c])
                d_dirs = ()
                if len(ctx.needs_input_grad) > n_c:      # directions with a gradient: the visible rows scattered into a dense [N, 3]
                    v_dirs = torch.zeros((N, 3), dtype=torch.float32, device=dev)
                    if dir_rows is not None and n_vis > 0:
                        v_dirs.index_copy_(0, vis_ids[:n_vis].long(), dir_rows[:n_vis])
                    d_dirs = tuple(v_dirs[s:s + n_] if nd else None for (s, n_, *_), nd in zip(cs.node_params, ctx.needs_input_grad[n_c:]))
                if d_coeffs is not None and not any(need_c):
                    d_coeffs = None
                if cs.n_nodes > 1 and d_coeffs is not None:      # one dense buffer in collected order: every node's gradient is its slice
                    parts = [d_coeffs[s:s + n_] if nd else None for (s, n_, *_), nd in zip(cs.node_params, need_c)]
                    d_coeffs, d_more = parts[0], tuple(parts[1:])
                else:
                    d_more = (None,) * (cs.n_nodes - 1)
                d_more = d_more + d_dirs
            elif n_vis > 0:
                d_campos = -dir_part.sum(0)
            else:
                d_campos = torch.zeros(3, dtype=torch.float32, device=dev)
        if want_m2d:
            m2d_out.grad = d_m2d      # what retain_grad() would have kept: the gradient reaching means2d
        if d_abs is not None:
            m2d_out.absgrad = d_abs   # gsplat: set in rasterize_to_pixels' backward (mtgs_scene_graph.py:1172)
        v_bg = None
        if bg is not None and need[7] and v_render is not None:
            v_bg = (v_render[..., :DC] * (1.0 - alphas)).sum(dim=(1, 2))
        if geo_rows:      # the per-visible rows ARE the geometry gradient (ColorSource.apply_to -> mtgs_node_bwd_rows -> the optimizer)
            cs.geo_ws = (vis_ws, vis_ids, totals if ctx.graph else None)
            return (None, None, None, None, d_col, v_viewmats, None, v_bg) + (None,) * 12 + (d_coeffs, d_campos) + d_more
        return (v_means if need[0] else None, v_quats if need[1] else None, v_scales if need[2] else None,
                v_opacities if need[3] else None, d_col, v_viewmats, None, v_bg) + (None,) * 12 + (d_coeffs, d_campos) + d_more


def _zeroed_outputs_plan(ctx, Cn, N, n_vis, DC, g_means2d, g_depths, g_conics, g_comps, g_opac):
    """Layout of the dense gradients of a _FusedRasterization backward inside ONE region that its compositing backward clears
    (see the call site), or None when this backward does not take that form: {"floats": total, "views": {name: (offset, count, shape)}},
    every view 16-byte aligned.  Mirrors the decisions the backward takes further down (which by-products it returns)."""
    if not (_prefill.enabled and _zeroed_outputs and ctx.packed and Cn == 1 and ctx.dp is None and n_vis > 0 and N > 0):
        return None
    if any(g is not None for g in (g_means2d, g_depths, g_conics, g_comps, g_opac)):      # (a loss on info[...]: the generic path)
        return None
    cs = ctx.cs
    if cs is not None and getattr(cs, "geometry_rows", False):
        return None
    m2d_out = ctx.means2d_ref() if getattr(ctx, "means2d_ref", None) is not None else None
    want_m2d = m2d_out is not None and m2d_out.retains_grad
    rows_only = cs is not None and getattr(cs, "want_grad_rows", False)
    want_abs = bool(ctx.absgrad and m2d_out is not None and not rows_only)
    c0 = 0 if cs is None else (6 if ctx.n2c is not None else 3)
    want_col = bool(DC - c0 and ctx.needs_input_grad[4])
    items = [("means", N * 3, (N, 3)), ("quats", N * 4, (N, 4)), ("scales", N * 3, (N, 3)), ("opacities", N, (N,))]
    if want_m2d:
        items.append(("m2d", N * 2, (1, N, 2)))
    if want_abs:
        items.append(("abs", N * 2, (1, N, 2)))
    if want_col:
        items.append(("col", N * (DC - c0), (1, N, DC - c0)))
    views, at = {}, 0
    for name, n_, shape in items:
        views[name] = (at, n_, shape)
        at += -(-n_ // 4) * 4
    return {"floats": at, "views": views}


_zeroed_outputs = os.environ.get("MTGS_ZEROED_OUTPUTS", "1") == "1"      # (development switch: 0 = the streaming expansion pass of rounds 1-5)


def _sh_inputs(sh_source):
    """(sh_coeffs, campos, *further coefficient tensors) of _FusedRasterization from fused_rasterization's sh_source =
    (coefficient tensor | list of them, campos | None) | None."""
    if sh_source is None:
        return (None, None)
    coeffs, campos = sh_source
    if isinstance(coeffs, (list, tuple)):
        return (coeffs[0], campos) + tuple(coeffs[1:])
    return (coeffs, campos)


def fused_rasterization(means, quats, scales, opacities, colors, viewmats, Ks, backgrounds, width, height, eps2d,
                        near_plane, far_plane, radius_clip, calc_compensations, with_depth, expected_depth, absgrad,
                        dp=None, color_source=None, sh_source=None):
    """One-node projection + binning + compositing (see _FusedRasterization).  colors[C,N,D] | None.
    Returns (render, alphas, dict of gsplat's meta tensors)."""
    import weakref
    if colors is not None or color_source is not None:
        opened = 0 if color_source is None else (6 if getattr(color_source, "camera_normals", None) is not None else 3)
        total = (0 if colors is None else colors.shape[-1]) + opened + int(with_depth)
        if total not in SUPPORTED_CHANNELS:
            raise ValueError(f"fused_rasterization: {total} blended channels (supported: {SUPPORTED_CHANNELS})")
    # the spherical_harmonics() forwards these colours come from: only THEIR backward's zeros ride on this rasterization (_Prefill)
    _sh_scope.reqs = _prefill.behind(colors) if (_prefill.enabled and colors is not None and colors.requires_grad) else ()
    try:
        out = _FusedRasterization.apply(means, quats, scales, opacities, colors, viewmats, Ks, backgrounds, int(width),
                                        int(height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip),
                                        bool(calc_compensations), bool(with_depth), bool(expected_depth), bool(absgrad), dp,
                                        color_source, *_sh_inputs(sh_source),
                                        *(getattr(color_source, "dirs_inputs", None) or ()))
    finally:
        _sh_scope.reqs = ()
    (render, alphas, radii, means2d, depths, conics, comps, opac_eff, tiles_per_gauss, isect_ids, flatten_ids,
     offsets, totals) = out
    if render.grad_fn is not None:  # the backward sets .grad / .absgrad on this very tensor (weak: no cycle)
        render.grad_fn.means2d_ref = weakref.ref(means2d)
    meta = {"radii": radii, "means2d": means2d, "depths": depths, "conics": conics,
            "compensations": comps if calc_compensations else None, "opacities": opac_eff,
            "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids, "flatten_ids": flatten_ids, "isect_offsets": offsets}
    if getattr(offsets, "_mtgs_n_listed", None) is not None:
        meta["n_listed"] = offsets._mtgs_n_listed    # int32 device scalar: valid prefix of isect_ids / flatten_ids (tight lists)
    if _graph.caps is not None and totals.numel() == 4:   # graph mode: the counts live on the device (see graph_mode) -- views of words the binning wrote,
        # no launch (five tiny torch kernels unpacked totals[0] here until round 5)
        meta.update({"n_visible": totals[1], "n_intersections": totals[2], "overflow": totals[3:4].view(torch.bool)[0]})
    return render, alphas, meta


def _pad_channels(colors: Optional[Tensor], backgrounds: Optional[Tensor], extra: int):
    """Zero-pads the colour channels so that colours + `extra` fused channels is a supported count."""
    channels = 0 if colors is None else colors.shape[-1]
    total = channels + extra
    if total > MAX_CHANNELS:
        raise ValueError(f"rasterize_to_pixels: {total} channels > {MAX_CHANNELS}; chunk on the caller side")
    pad = next(d for d in SUPPORTED_CHANNELS if d >= total) - total
    if pad:
        Cn, N = colors.shape[:2]
        colors = torch.cat([colors, colors.new_zeros(Cn, N, pad)], dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, backgrounds.new_zeros(Cn, pad)], dim=-1)
    return colors, backgrounds, channels, pad


def rasterize_to_pixels(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor,
                        image_width: int, image_height: int, tile_size: int, isect_offsets: Tensor,
                        flatten_ids: Tensor, backgrounds: Optional[Tensor] = None,
                        masks: Optional[Tensor] = None, packed: bool = False,
                        absgrad: bool = False) -> Tuple[Tensor, Tensor]:
    """gsplat.cuda._wrapper.rasterize_to_pixels (packed=False) ->
    (render_colors[C,H,W,D], render_alphas[C,H,W,1])."""
    _check_raster_args(means2d, conics, colors, opacities, backgrounds, masks, packed, tile_size, isect_offsets,
                       image_width, image_height)
    colors, backgrounds, channels, pad = _pad_channels(colors, backgrounds, 0)
    render, alphas = _RasterizeToPixels.apply(means2d, conics, colors, opacities, backgrounds, None, False,
                                              int(image_width), int(image_height), int(tile_size),
                                              isect_offsets, flatten_ids, bool(absgrad))
    if pad:
        render = render[..., :channels]
    return render, alphas


def rasterize_to_pixels_with_depth(means2d: Tensor, conics: Tensor, colors: Optional[Tensor], opacities: Tensor,
                                   depths: Tensor, expected_depth: bool, image_width: int, image_height: int,
                                   tile_size: int, isect_offsets: Tensor, flatten_ids: Tensor,
                                   backgrounds: Optional[Tensor] = None, absgrad: bool = False):
    """rasterize_to_pixels with the depth channel of the "RGB+D" / "RGB+ED" / "D" / "ED" render modes
    blended in the same pass (last output channel; divided by clamp(alpha, 1e-10) for expected depth)
    -- what gsplat.rendering.rasterization composes from torch.cat + rasterize_to_pixels + a division."""
    if colors is not None:
        _check_raster_args(means2d, conics, colors, opacities, backgrounds, None, False, tile_size, isect_offsets,
                           image_width, image_height)
        colors, backgrounds, channels, pad = _pad_channels(colors, backgrounds, 1)
    else:
        channels, pad, backgrounds = 0, 0, None
    render, alphas = _RasterizeToPixels.apply(means2d, conics, colors, opacities, backgrounds, depths,
                                              bool(expected_depth), int(image_width), int(image_height),
                                              int(tile_size), isect_offsets, flatten_ids, bool(absgrad))
    if pad:
        render = torch.cat([render[..., :channels], render[..., -1:]], dim=-1)
    return render, alphas


def _check_raster_args(means2d, conics, colors, opacities, backgrounds, masks, packed, tile_size, isect_offsets,
                       image_width, image_height):
    if packed:
        raise NotImplementedError("rasterize_to_pixels: packed=True")
    if masks is not None:
        raise NotImplementedError("rasterize_to_pixels: tile masks")
    if tile_size != 16:
        raise NotImplementedError(f"rasterize_to_pixels: tile_size={tile_size} (only 16 is implemented)")
    Cn, N = means2d.shape[:2]
    assert means2d.shape == (Cn, N, 2), means2d.shape
    assert conics.shape == (Cn, N, 3), conics.shape
    assert colors.shape[:2] == (Cn, N), colors.shape
    assert opacities.shape == (Cn, N), opacities.shape
    if backgrounds is not None:
        assert backgrounds.shape == (Cn, colors.shape[-1]), backgrounds.shape
    th, tw = isect_offsets.shape[1:]
    assert tw * tile_size >= image_width and th * tile_size >= image_height
