"""View-parallel data parallelism for multi-traversal training: one process per GPU, replicated
Gaussians, each rank renders its own camera(s), ONE all-reduce of the Gaussian gradients per step
over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).

The reference's only collective site is nerfstudio's DDP wrap
(/root/reference/mtgs/scene_model/custom_pipeline.py:87-89), which cannot survive densification
(SURVEY.md section 5); its documented multi-GPU mode is scene-per-GPU.  This module is the
MI355X-native equivalent for the step itself: all parameter gradients live in ONE flat fp32
buffer (236 B per Gaussian at SH degree 3), so a step issues a single large collective instead
of per-tensor buckets -- xGMI is point-to-point (7 links per GPU), large messages are what keep
every link busy.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, timeout_s: Optional[float] = None) -> tuple:
    """Initialises torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, local_rank, world_size).  A single process (no env) returns (0, 0, 1).
    timeout_s: collective timeout (a rank that never arrives then fails the job instead of hanging it)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # MTGS_DIST_BACKEND=gloo lets the N > 1 path be exercised with several ranks on ONE GPU
            backend = os.environ.get("MTGS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", torch.cuda.current_device())
        if timeout_s is not None:
            import datetime
            kwargs["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def camera_for_rank(step: int, rank: int, world: int, n_cameras: int) -> int:
    """Round-robin view sharding: at step s, rank r renders camera (s*world + r) mod n_cameras."""
    return (step * world + rank) % n_cameras


class FlatGradBucket:
    """All gradients of `params` as views into one contiguous fp32 buffer.

    `param.grad` is pre-set to a view of the buffer, so autograd accumulates in place and
    `all_reduce()` is a single collective over the whole buffer (no flatten/unflatten copies)."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params: List[torch.Tensor] = list(params)
        assert self.params, "no parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        for p in self.params:
            assert p.device == dev and p.dtype == dt and p.is_leaf and p.requires_grad
        # 256-byte aligned segments keep every view 16-byte aligned for vector loads
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.views = []
        for p, o in zip(self.params, offs):
            v = self.flat[o:o + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)

    def zero(self) -> None:
        self.flat.zero_()
        for p, v in zip(self.params, self.views):  # re-attach in case someone replaced .grad
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def all_reduce(self, average: bool = False, group=None, async_op: bool = False):
        """Sum (or mean, to mimic DDP) the gradients over ranks.  No-op in a single process."""
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if average:
            if async_op:
                work.wait()
            self.flat.div_(dist.get_world_size(group))
        return work


def all_reduce_grads(params: Sequence[torch.Tensor], average: bool = False, group=None,
                     direct_bytes: int = 32 << 20) -> int:
    """Sum (or average) `p.grad` of every parameter over the ranks, in place, WITHOUT a persistent
    bucket: gradients of at least `direct_bytes` are reduced where they are (SH coefficients: 384 MB
    at 2M Gaussians -- no copy, no zero-fill, no read-modify-write accumulation), the small ones are
    packed into one temporary flat buffer so that they cost a single collective.  Returns the number
    of bytes put on the wire by this rank's buffers.  No-op in a single process."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    big = [g for g in grads if g.is_contiguous() and g.numel() * g.element_size() >= direct_bytes]
    small = [g for g in grads if not any(g is b for b in big)]
    works, nbytes = [], 0
    for g in big:
        works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True))
        nbytes += g.numel() * g.element_size()
    flat = None
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
        nbytes += flat.numel() * flat.element_size()
    for w in works:
        w.wait()
    if flat is not None:
        off = 0
        for g in small:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    if average:
        for g in grads:
            g.div_(world)
    return nbytes


_DP_GROUP = np.dtype([("mask", "<u8"), ("coef_rows", "<u8"), ("coef_words", "<u8"), ("coef_prefix", "<u8"), ("coef_row_of", "<u8"),
                      ("coef_cap", "<i8")])       # include/mtgs_rast.h: mtgs_dp_group


_DP_CHUNKS = np.dtype([("n", "<i4"), ("_pad", "<i4"), ("begin", "<i8", (17,)), ("rows", "<u8", (16,)), ("cap", "<i8", (16,))])   # mtgs_dp_chunks


class SparseGradExchange:
    """Sparse, factored replacement for the dense gradient all-reduce of view-parallel DP.

    A rank renders one camera per step, so only the Gaussians visible in it (~15 % of a road block) have a non-zero
    gradient, and the gradient of the SH coefficients is rank-1 per Gaussian:
    v_coeffs[n,k,:] = basis_k(normalize(mean_n - cam_pos)) * v_rgb[n,:].  Each rank therefore sends 64-byte rows
    {v_mean 3, v_quat 4, v_scale 3, v_opacity 1, v_rgb 3, -, index} of its visible Gaussians, in INDEX order, plus a visibility map
    (one bit per Gaussian + the row of the first Gaussian of every 64) and its camera position: 19 MB + 0.4 MB instead of
    472 MB at 2M Gaussians / SH degree 3.  Every rank rebuilds the SUM of all ranks' dense gradients in a streaming
    pass over the Gaussians (csrc/dp.hip: mtgs_dp_reduce finds every Gaussian in every sender's rows through the sender's
    map, sums in LDS, and writes each dense tensor once).  Equal to the dense all-reduce up to fp32 summation order.

    Integrated form (what bench.py and the training harness use; `rasterization()` + `finish()`):
      * the rows ARE the projection backward's per-visible output (mtgs_project_bwd_rows): no dense gradient tensor, no
        SH backward, no pack pass on the sender (the receivers' reduction writes every dense tensor, own rows included);
      * the visibility map is written by the forward's front kernels and all-gathered on a side stream WHILE the frame
        is composited; the row counts reach the host from that stream, so nothing on the critical path synchronises;
      * the rows travel in `chunks` pieces of the Gaussian index range, and chunk i is reduced while chunk i + 1 is on
        the wire.
    CONTRACT: what is summed over ranks is the gradient that flows through the rasterizer.  Gradients that other loss terms
    put on the parameters (scale / opacity regularisers, ...) reach `param.grad` through autograd as usual and are NOT
    exchanged here: they are identical on every rank when computed from the replicated parameters; if they are not,
    reduce them with `all_reduce_grads`.

    Tensor form (`exchange()`; any colour pipeline, K <= 16): this rank's dense rasterizer gradients in, sums out; only
    rows of Gaussians with radii > 0 are sent, so a gradient sitting on a Gaussian this rank's camera does not see would
    be dropped -- `check_contract=True` asserts there is none.  SH layouts the one-pass kernel does not cover (K > 16 or
    degree > 3) take the per-sender read-modify-write path (mtgs_dp_pack / mtgs_dp_accumulate)."""

    ROW = 16

    def __init__(self, n_gaussians: int, n_sh_bases: int, device, group=None, chunks: int = 4, traversals: int = 1):
        self.N, self.K, self.device, self.group = int(n_gaussians), int(n_sh_bases), device, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # tests: issue the collectives also in a one-rank group (normally short-cut), so that the RCCL code path of the
        # integrated form executes on a single-GPU box
        self.world_collectives = False
        self.grouped_receiver = True     # finish(rows=True): all colour groups + the geometry in one launch per chunk
        N = self.N
        # chunk boundaries of the index range (multiples of 2048, so that they are visibility-word and tile aligned)
        per = -(-max(N, 1) // max(int(chunks), 1))
        per = -(-per // 2048) * 2048
        self.bounds = list(range(0, N, per)) + [N] if N > 0 else [0, 0]
        self.n_chunks = len(self.bounds) - 1
        slack = min(per, N) + 64
        self.rows = torch.empty((N + slack, self.ROW), dtype=torch.float32, device=device)  # send buffer (index order)
        self.count = torch.zeros(1, dtype=torch.int64, device=device)
        # meta record of a rank, int32 words: [count, cam x, cam y, cam z | words (u64) ... | prefix (u32) ...]
        self.n_words = (N + 63) // 64
        # (two trailing words: the traversal of this rank's camera -- per-traversal appearance parameters -- and a spare;
        #  the length stays even, so every rank's u64 words stay 8-byte aligned)
        self.meta_len = 4 + 3 * self.n_words + (self.n_words & 1) + 2
        self.meta = torch.zeros(self.meta_len, dtype=torch.int32, device=device)
        self.T = int(traversals)
        assert self.T >= 1
        self.block_counts = torch.empty((N + 1023) // 1024 + 1, dtype=torch.int32, device=device)  # pack scratch
        # what the host needs from every rank's meta: the row count and the row index at every chunk boundary
        self._sample_idx = torch.tensor([0] + [4 + 2 * self.n_words + min(b // 64, max(self.n_words - 1, 0))
                                               for b in self.bounds[:-1]] + [self.meta_len - 2], dtype=torch.int64, device=device)
        is_cuda = torch.device(device).type == "cuda"
        self._samples_host = torch.zeros((self.world, len(self.bounds) + 1), dtype=torch.int32)
        if is_cuda:
            self._samples_host = self._samples_host.pin_memory()
        self.comm_stream = torch.cuda.Stream(device) if is_cuda else None
        self.last_bytes = 0
        self.phase = "idle"      # where the exchange is: named by bench.py / the harness when a collective fails
        self._pending = None
        self._events = {}
        self.grad_rows = self.vis_ids = None
        # colour channels beyond the SH output (camera-space normals ...) are functions of THIS rank's camera: the caller
        # sets rows_hook(grad_rows, row_stride, vis_ids, n_vis) to fold their gradient into the wire rows before they leave
        self.rows_hook = None
        # finish_touched(): the map of the rows that carry a gradient travels WITH the rows (one collective per step) -- a caller that
        # only ever finishes that way sets defer_maps = True, and the front end's visibility maps are not all-gathered at all
        self.defer_maps = False
        self._touched = None
        self._touched_chunks = None
        self._chunk_words = None
        self._recover = None     # the last frame finished through a truncating form: finish_recover() repeats it untruncated
        self._keep = None
        # prezero = True: the dense sums of finish_touched() / finish_touched_chunked() are allocated when the frame STARTS, as one
        # region the frame's compositing forward clears beside its own work (mtgs_blend_fwd_packed(also_zero): VALU-bound, HBM ~85 %
        # idle -- what the single-GPU step does with the SH backward's zeros), and the reduction writes the touched Gaussians only
        self.prezero = False
        self.zero_region = None      # (pointer, bytes) for the rasterization's forward; _zero_out: the five views
        self._zero_out = None

    # ---- integrated form -----------------------------------------------------------------------------------------
    def rasterization(self, means, quats, scales, opacities, sh_out, viewmats, Ks, width, height, cam_pos, near_plane=0.01,
                      far_plane=1e10, radius_clip=0.0, eps2d=0.3, render_mode="RGB+ED", rasterize_mode="antialiased",
                      absgrad=True, traversal: int = 0):
        """This rank's camera of the step.  Same outputs as `mtgs_amd.rasterization(colors=clamp(sh_out + 0.5, 0, 1), ...)`
        with MTGS's options (mtgs_scene_graph.py:641-659); `sh_out[N,3]` is the SH evaluation for THIS camera
        (`spherical_harmonics(n, means - cam_pos, coeffs)`, detached: its backward happens on the receivers).  The
        backward leaves the gradients as wire rows; call `finish()` after it.  `traversal` (with traversals = T > 1): the
        traversal this rank's camera belongs to -- MTGS's multi-colour nodes have one set of SH coefficients per traversal
        (multi_color_gaussian_splatting.py:77-101), and finish() then returns the coefficient gradient as [N, T, K, 3]
        with every sender's contribution in ITS traversal's slice."""
        from .wrapper import _LazySH, fused_rasterization
        assert viewmats.shape[0] == 1 and sh_out.dim() == 2 and sh_out.shape[0] == self.N and sh_out.shape[1] >= 3 and means.shape == (self.N, 3)
        assert render_mode in ("RGB", "RGB+D", "RGB+ED") and rasterize_mode in ("classic", "antialiased")
        assert 0 <= int(traversal) < self.T, (traversal, self.T)
        self.abandon()      # a previous frame that never reached finish() (forward-only / eval call, an exception in between)
        self.meta[1:4].copy_(cam_pos.reshape(3).to(torch.float32).contiguous().view(torch.int32))
        self.meta[self.meta_len - 2:self.meta_len - 1].fill_(int(traversal))
        self.zero_region = self._zero_out = None
        if self.prezero and torch.is_grad_enabled():
            N, K, T = self.N, self.K, self.T
            sizes = [N * 3, N * 4, N * 3, N, N * T * K * 3]
            offs = np.cumsum([0] + [-(-s // 4) * 4 for s in sizes])
            region = torch.empty(int(offs[-1]), dtype=torch.float32, device=self.device)
            shapes = [(N, 3), (N, 4), (N, 3), (N,), (N, K, 3) if T == 1 else (N, T, K, 3)]
            self._zero_out = tuple(region[int(o):int(o) + s].view(sh) for o, s, sh in zip(offs[:-1], sizes, shapes))
            self.zero_region = (region.data_ptr(), region.numel() * 4)
        self._pending = {"stage": "forward"}
        self.phase = "render (forward; meta all-gather on the side stream)"
        # `sh_out` still deferred (spherical_harmonics() returns a deferred tensor, wrapper._LazySH): SH + clamp are evaluated for the
        # Gaussians this camera sees only, by the rasterization (csrc/viscolor.hip) -- no [N, 3] tensor, 85 % of the coefficient rows unread
        cs = sh_out.exchange_source(self.N) if (type(sh_out) is _LazySH and sh_out.shape[1] == 3) else None
        cols = None if cs is not None else sh_out.detach().unsqueeze(0)
        render, alphas, m = fused_rasterization(
            means, quats, scales, opacities, cols, viewmats, Ks, None, width, height, eps2d,
            near_plane, far_plane, radius_clip, rasterize_mode == "antialiased", render_mode != "RGB",
            render_mode == "RGB+ED", absgrad, dp=self, color_source=cs)
        return render, alphas, m

    def abandon(self):
        """Drops a frame whose exchange was started (rasterization()) but never finished: waits -- on the side stream only --
        until the all-gather of its meta record no longer reads `self.meta`, which the next frame rewrites.  Forward-only
        callers (evaluation) call this instead of backward() + finish()."""
        P, self._pending, self._recover = self._pending, None, None
        if P is not None and P.get("done") is not None:
            torch.cuda.current_stream().wait_event(P["done"])

    def front_pointers(self):
        """(visibility words, row prefix per word, row count) inside this rank's meta record: written by mtgs_front_fwd."""
        base, nw = self.meta.data_ptr(), self.n_words
        return base + 16, base + 16 + 8 * nw, base

    def after_front(self):
        """Called right after the front kernels are enqueued: all-gather the meta records on the side stream and bring
        the row counts / chunk starts of every rank to pinned host memory -- overlaps the binning and the compositing."""
        if self.defer_maps:
            self._pending = {"stage": "meta", "metas": None, "done": None}
            return
        if self.comm_stream is None:
            # CPU tensors (tests/test_dist_gloo.py: the host-side bookkeeping of the exchange over gloo): the same collective and
            # the same samples, no streams or events
            if self.world > 1 or self.world_collectives:
                metas = torch.empty((self.world, self.meta_len), dtype=torch.int32, device=self.device)
                dist.all_gather_into_tensor(metas, self.meta[None].contiguous(), group=self.group)
            else:
                metas = self.meta[None]
            self._samples_host.copy_(metas[:, self._sample_idx])
            self._pending = {"stage": "meta", "metas": metas, "done": None}
            return
        ev = torch.cuda.Event()
        ev.record()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            t0.record()
            if self.world > 1 or self.world_collectives:
                metas = torch.empty((self.world, self.meta_len), dtype=torch.int32, device=self.device)
                dist.all_gather_into_tensor(metas, self.meta[None], group=self.group)
            else:
                metas = self.meta[None]
            self._samples_host.copy_(metas[:, self._sample_idx], non_blocking=True)
            t1.record()
            done = torch.cuda.Event()
            done.record()
        self._pending = {"stage": "meta", "metas": metas, "done": done}
        self._events["meta"] = (t0, t1)

    def after_backward(self, n_vis, grad_rows, vis_ids):
        assert self._pending is not None and self._pending["stage"] == "meta", "rasterization() of this exchange first"
        self._pending.update(stage="rows", n_vis=int(n_vis))
        self.phase = "render done (wire rows written)"
        self.n_vis = int(n_vis)
        # the compact gradient rows of this frame: what mtgs_amd.densify.update_statistics_rows reads (no dense means2d
        # gradient exists in this mode)
        self.grad_rows, self.vis_ids = grad_rows, vis_ids

    def finish_touched(self, means: torch.Tensor, sh_degree: int, cap_rows: int, traversal_of_rank: Sequence[int]):
        """finish_static() on the rows that CARRY a gradient only.  A rank's wire rows cover what its camera sees; in an opaque
        scene the compositing terminates long before the ends of the tile lists, and most visible Gaussians get exactly zero
        (60 % of the rows at the headline scene, 91-98 % in MTGS-like scenes).  `mtgs_dp_touched_pack` compacts the non-zero rows
        (index order kept) and builds THEIR map in the visibility map's format, both go into ONE send buffer
        [cap_rows rows | meta record], and ONE all-gather per step carries everything a receiver needs: 2.5x .. 50x fewer bytes on
        the wire and as many fewer rows through `mtgs_dp_reduce`, which runs unchanged.  No host read, no host wait (capturable
        with RCCL).  cap_rows: rows per rank on the wire (fixed; from the touched counts of earlier steps plus a margin -- a
        step's count is `self.touched_count`, an int32 device scalar); `overflow` (device bool) = some rank had more: repeat the
        step's exchange through finish_static() / finish(), which read the untouched rows buffer.
        With `defer_maps = True` the front end's visibility maps are not exchanged at all.
        Returns ((v_means, v_quats, v_scales, v_opacities, v_coeffs), overflow): the dense sums of finish() -- bit-identical
        (a zero row adds exact zeros; the order of the other rows is unchanged)."""
        from ._lib import call, ptr, stream_of
        import ctypes as _C
        P = self._pending
        assert P is not None and P["stage"] == "rows", "finish_touched() follows rasterization() + backward()"
        self._pending, self._recover = None, P       # (the frame stays recoverable -- finish_recover() -- until the next one starts)
        N, K, dev, world, nw, T = self.N, self.K, self.device, self.world, self.n_words, self.T
        assert len(traversal_of_rank) == world and all(0 <= int(t) < T for t in traversal_of_rank)
        cap = int(max(1, min(cap_rows, self.rows.shape[0])))
        means = means.detach().contiguous()
        st = stream_of(means)
        if P.get("done") is not None:
            torch.cuda.current_stream().wait_event(P["done"])      # (a frame that did all-gather its visibility maps: stream order only)
        pad = -(-self.meta_len // 16) * 16           # (every sender's block is a whole number of 64-byte rows)
        L = cap * self.ROW + pad                     # [rows | meta]; the reduction reads at most `cap` rows of a block (row_cap): a sender
        #                                              that overflowed never has its map words summed as floats
        m0 = cap * self.ROW                          # first int32 of the meta record inside a block
        tb = self._touched
        if tb is None or tb["send"].numel() != L:
            tb = self._touched = {"send": torch.zeros(L, dtype=torch.float32, device=dev),
                                  "scratch": torch.empty(max(nw, 1), dtype=torch.int64, device=dev),
                                  "blocks": torch.empty(max(nw, 1) // 256 + 2, dtype=torch.int32, device=dev),
                                  "totals": torch.zeros(1, dtype=torch.int64, device=dev)}
        send = tb["send"]
        send_i = send.view(torch.int32)
        base = send.data_ptr()
        self.phase = "exchange (touched): compaction of the rows that carry a gradient"
        send_i[m0 + 1:m0 + 4].copy_(self.meta[1:4])                                          # camera position
        send_i[m0 + self.meta_len - 2:m0 + self.meta_len - 1].copy_(self.meta[self.meta_len - 2:self.meta_len - 1])      # traversal
        mb = base + 4 * m0
        call("mtgs_dp_touched_pack", int(self.n_vis), ptr(self.rows), N, ptr(tb["scratch"]), mb + 16, mb + 16 + 8 * nw, mb,
             ptr(tb["totals"]), ptr(tb["blocks"]), base, cap, st)
        self.touched_count = send_i[m0]
        self.phase = f"exchange (touched): all-gather of {cap} rows + the touched map per rank"
        if world > 1 or self.world_collectives:
            recv = torch.empty((world, L), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(recv.view(world * L), send, group=self.group)
            self.last_bytes = world * L * 4
        else:
            recv, self.last_bytes = send[None], 0
        recv_i = recv.view(torch.int32)
        cams = recv[:, m0 + 1:m0 + 4].contiguous()
        words_all, prefix_all = recv_i[:, m0 + 4:], recv_i[:, m0 + 4 + 2 * nw:]
        rows_all = recv
        overflow = (recv_i[:, m0] > cap).any()
        out, sparse = self._dense_outputs()
        self.phase = "exchange (touched): reduction"
        stride_b, stride_f = L * 4, L
        masks = [sum(1 << r for r in range(world) if int(traversal_of_rank[r]) == t) for t in range(T)]
        for t in range(T):      # (T == 1: one pass, every sender)
            call("mtgs_dp_reduce_slices_cap", world, N, K, int(sh_degree), ptr(means), ptr(words_all), ptr(prefix_all), stride_b,
                 ptr(rows_all), stride_f, cap, ptr(cams), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]),
                 out[4].data_ptr() + t * K * 3 * 4, 0, -1, _C.c_uint64(masks[t]), int(t == 0) | sparse, T * K * 3, st)
        self.phase = "idle"
        return out, overflow

    # ---- the same exchange in CHUNKS of the index range: wire and reduction overlap, still no host read -------------------------
    def touched_chunk_counts(self) -> torch.Tensor:
        """Rows that carried a gradient in every index chunk of the LAST finish_touched_chunked() step (int64 [n_chunks], device):
        what the per-chunk capacities are agreed from in a warm-up step (MAX over the ranks + a margin)."""
        tb = self._touched_chunks
        assert tb is not None, "finish_touched_chunked() first"
        pre = tb["send0"].view(torch.int32)[4 + 2 * self.n_words:4 + 3 * self.n_words].to(torch.int64)
        total = tb["send0"].view(torch.int32)[0:1].to(torch.int64)
        starts = torch.cat([pre[self._chunk_words[:-1]], total])
        return starts[1:] - starts[:-1]

    def finish_touched_chunked(self, means: torch.Tensor, sh_degree: int, cap_rows: Sequence[int], traversal_of_rank: Sequence[int]):
        """finish_touched() with the rows in `n_chunks` pieces of the Gaussian index range (the constructor's `chunks`), each its own
        all-gather: the K collectives are issued back to back (they queue on the collective stream) and chunk c is reduced on the
        main stream as soon as it has arrived, while chunks c + 1 .. are still on the wire -- the overlap finish() has, without its
        host reads: capacities are static (cap_rows[c] rows of chunk c per rank, agreed in a warm-up step from
        touched_chunk_counts()), counts and the overflow flag stay on the device, stream order is all that joins the collectives
        and the reductions (async_op work handles: a stream wait under RCCL), so a HIP graph can hold the whole step.
        The first message carries the sender's meta record (camera, traversal, the touched rows' map for the WHOLE range: 0.37 MB
        at 2M Gaussians) in front of chunk 0's rows.  Rows are compacted in index order, so chunk c is a contiguous range of a
        sender's touched rows; a chunk with more rows than its capacity sets `overflow` (device bool; finish_recover() repeats
        the exchange untruncated).
        Returns ((v_means, v_quats, v_scales, v_opacities, v_coeffs), overflow): bit-identical to finish_touched() (the same rows
        meet every Gaussian in the same sender order)."""
        from ._lib import call, ptr, stream_of
        import ctypes as _C
        P = self._pending
        assert P is not None and P["stage"] == "rows", "finish_touched_chunked() follows rasterization() + backward()"
        self._pending, self._recover = None, P
        N, K, dev, world, nw, T, nch = self.N, self.K, self.device, self.world, self.n_words, self.T, self.n_chunks
        assert len(traversal_of_rank) == world and all(0 <= int(t) < T for t in traversal_of_rank)
        assert len(cap_rows) == nch and nch <= 16, (len(cap_rows), nch)
        caps = [int(max(1, min(int(c), self.rows.shape[0]))) for c in cap_rows]
        means = means.detach().contiguous()
        st = stream_of(means)
        if P.get("done") is not None:
            torch.cuda.current_stream().wait_event(P["done"])
        lay = self.chunk_layout(caps)
        pad = lay["pad"]                              # the meta record, padded to whole 64-byte rows, leads message 0
        tb = self._touched_chunks
        if tb is None or tb["caps"] != caps:
            sends = [torch.zeros(n, dtype=torch.float32, device=dev) for n in lay["floats"]]
            tab = np.zeros(1, dtype=_DP_CHUNKS)
            tab["n"] = nch
            tab["begin"][0, :nch + 1] = self.bounds
            for c in range(nch):
                tab["rows"][0, c] = sends[c].data_ptr() + (pad * 4 if c == 0 else 0)
                tab["cap"][0, c] = caps[c]
            tb = self._touched_chunks = {"caps": caps, "sends": sends, "send0": sends[0], "tab": tab,
                                         "scratch": torch.empty(max(nw, 1), dtype=torch.int64, device=dev),
                                         "blocks": torch.empty(max(nw, 1) // 256 + 2, dtype=torch.int32, device=dev),
                                         "totals": torch.zeros(1, dtype=torch.int64, device=dev)}
            self._chunk_words = torch.tensor([min(b // 64, max(nw - 1, 0)) for b in self.bounds], dtype=torch.int64, device=dev)
        sends, send0 = tb["sends"], tb["send0"]
        s0i = send0.view(torch.int32)
        self.phase = "exchange (touched, chunked): compaction of the rows that carry a gradient"
        s0i[1:4].copy_(self.meta[1:4])                                                        # camera position
        s0i[self.meta_len - 2:self.meta_len - 1].copy_(self.meta[self.meta_len - 2:self.meta_len - 1])      # traversal
        mb = send0.data_ptr()
        call("mtgs_dp_touched_pack_chunks", int(self.n_vis), ptr(self.rows), N, ptr(tb["scratch"]), mb + 16, mb + 16 + 8 * nw, mb,
             ptr(tb["totals"]), ptr(tb["blocks"]), tb["tab"].ctypes.data, mb + 4 * (self.meta_len - 1), st)
        self.touched_count = s0i[0]
        collectives = world > 1 or self.world_collectives
        works, recvs = self.gather_chunk_messages(sends)
        out, sparse = self._dense_outputs()
        masks = [sum(1 << r for r in range(world) if int(traversal_of_rank[r]) == t) for t in range(T)]
        r0 = recvs[0]
        r0i = r0.view(torch.int32)
        stride0_b = r0.shape[1] * 4
        cams = None
        for c in range(nch):
            self.phase = f"exchange (touched, chunked): wire + reduction of chunk {c} of {nch}"
            if works[c] is not None:
                works[c].wait()
            if c == 0:
                cams = r0[:, 1:4].contiguous()
            rows_c = recvs[c][:, pad:] if c == 0 else recvs[c]
            for t in range(T):
                call("mtgs_dp_reduce_slices_cap", world, N, K, int(sh_degree), ptr(means), r0i.data_ptr() + 16, r0i.data_ptr() + 16 + 8 * nw,
                     stride0_b, rows_c.data_ptr(), recvs[c].shape[1] if collectives else 0, caps[c], ptr(cams), ptr(out[0]), ptr(out[1]),
                     ptr(out[2]), ptr(out[3]), out[4].data_ptr() + t * K * 3 * 4, self.bounds[c], self.bounds[c + 1],
                     _C.c_uint64(masks[t]), int(t == 0) | sparse, T * K * 3, st)
        overflow = (r0i[:, self.meta_len - 1] != 0).any()
        self._keep = (recvs, works)      # (the receive buffers stay referenced until the next step: the collective stream may still own them)
        self.phase = "idle"
        return out, overflow

    def _dense_outputs(self):
        """(the five dense sums, flag): the views of the region this frame's compositing forward cleared (prezero; flag 2 = the
        reduction writes the touched Gaussians only) or fresh uninitialised tensors the reduction writes completely (flag 0)."""
        N, K, T, dev = self.N, self.K, self.T, self.device
        if self._zero_out is not None:
            out, self._zero_out, self.zero_region = self._zero_out, None, None
            return out, 2
        return (torch.empty((N, 3), dtype=torch.float32, device=dev), torch.empty((N, 4), dtype=torch.float32, device=dev),
                torch.empty((N, 3), dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev),
                torch.empty((N, K, 3) if T == 1 else (N, T, K, 3), dtype=torch.float32, device=dev)), 0

    def chunk_layout(self, caps: Sequence[int]) -> dict:
        """Messages of finish_touched_chunked(): message 0 = [meta record padded to `pad` floats | caps[0] rows], message c = caps[c]
        rows; "floats"[c] = length of message c, "row0"[c] = first float of its rows.  A pure function of the capacities (identical
        on every rank once they are agreed), which is what makes the ranks' collectives match without a host exchange per step."""
        pad = -(-self.meta_len // 16) * 16
        caps = [int(c) for c in caps]
        return {"pad": pad, "floats": [pad + caps[0] * self.ROW] + [c * self.ROW for c in caps[1:]], "row0": [pad] + [0] * (len(caps) - 1)}

    def gather_chunk_messages(self, sends: Sequence[torch.Tensor]):
        """Every chunk message's all-gather, issued back to back (async: they queue on the collective stream / the gloo thread).
        Returns (works, recvs[c] [world, len(message c)]); a single rank without `world_collectives` short-cuts to views."""
        works, recvs = [], []
        self.last_bytes = 0
        for c, s in enumerate(sends):
            self.phase = f"exchange (touched, chunked): issuing the all-gather of chunk {c} of {len(sends)} ({s.numel() * 4} bytes per rank)"
            if self.world > 1 or self.world_collectives:
                recv = torch.empty((self.world, s.numel()), dtype=s.dtype, device=s.device)
                works.append(dist.all_gather_into_tensor(recv.view(-1), s, group=self.group, async_op=True))
                self.last_bytes += recv.numel() * 4
            else:
                recv = s[None]
                works.append(None)
            recvs.append(recv)
        return works, recvs

    def finish_recover(self, means: torch.Tensor, sh_degree: int, rows: bool = False, all_colour_ranges=()):
        """After finish_touched() / finish_touched_chunked() / finish_static() reported `overflow` (the caller read the device flag):
        repeat the step's exchange UNTRUNCATED through finish(), from the rows buffer those forms leave untouched.  A frame whose
        visibility maps were never exchanged (defer_maps) all-gathers them now (blocking: this is the rare path).  Collective:
        every rank calls it (the overflow flag is the same on every rank -- it is computed from all-gathered counts)."""
        P = self._recover
        assert P is not None and P.get("stage") == "rows", "finish_recover() follows a finish_*() of the same frame"
        if P.get("metas") is None:
            if self.world > 1 or self.world_collectives:
                metas = torch.empty((self.world, self.meta_len), dtype=torch.int32, device=self.device)
                dist.all_gather_into_tensor(metas, self.meta[None].contiguous(), group=self.group)
            else:
                metas = self.meta[None]
            self._samples_host.copy_(metas[:, self._sample_idx])      # (blocking copy to pinned memory)
            if self.comm_stream is not None:
                torch.cuda.current_stream().synchronize()
            P.update(metas=metas, done=None)
        self._pending, self._recover = P, None
        return self.finish(means, sh_degree, rows=rows, all_colour_ranges=all_colour_ranges)

    def finish_static(self, means: torch.Tensor, sh_degree: int, cap_rows: int, traversal_of_rank: Sequence[int]):
        """finish() WITHOUT any host read or host wait -- the form a HIP graph can capture (with RCCL; the dynamic form sizes
        its all-gathers from the ranks' row counts, which it reads from a pinned buffer behind an event).  Every rank sends a
        FIXED number of rows, `cap_rows` (its rows are in index order from row 0, so the first min(count, cap_rows) are sent;
        take the capacity from the size plan as mtgs_amd.graph_mode does for the frame), in ONE all-gather; the counts stay
        on the device: the reduction finds every Gaussian through the senders' maps and never indexes past a sender's
        capacity, and `overflow` (device bool) says whether some rank had more rows than `cap_rows` -- the step is then
        incomplete and is to be repeated through finish() (as a frame beyond its capacities is).  traversal_of_rank: the
        traversal each rank's camera belongs to this step -- known to every rank from the schedule (camera (step * world +
        rank) mod T in the harness), so no meta record has to reach the host.
        Returns ((v_means, v_quats, v_scales, v_opacities, v_coeffs), overflow): the dense sums of finish()."""
        from ._lib import call, ptr, stream_of
        import ctypes as _C
        P = self._pending
        assert P is not None and P["stage"] == "rows", "finish_static() follows rasterization() + backward()"
        self._pending, self._recover = None, P
        N, K, dev, world, nw, T = self.N, self.K, self.device, self.world, self.n_words, self.T
        assert len(traversal_of_rank) == world and all(0 <= int(t) < T for t in traversal_of_rank)
        cap = int(max(1, min(cap_rows, self.rows.shape[0])))
        means = means.detach().contiguous()
        st = stream_of(means)
        cur = torch.cuda.current_stream()
        cur.wait_event(P["done"])            # stream order only: the meta all-gather of the side stream
        metas = P["metas"]
        cams = metas[:, 1:4].contiguous().view(torch.float32)
        words_all, prefix_all = metas[:, 4:], metas[:, 4 + 2 * nw:]
        overflow = (metas[:, 0] > cap).any()
        self.phase = f"exchange (static): all-gather of {cap} rows per rank"
        if world > 1 or self.world_collectives:
            recv = torch.empty((world, cap, self.ROW), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(recv.view(world * cap, self.ROW), self.rows[:cap], group=self.group)
            row_stride = cap * self.ROW
            self.last_bytes = world * (cap * self.ROW * 4 + self.meta_len * 4)
        else:
            recv, row_stride, self.last_bytes = self.rows, 0, 0
        out = (torch.empty((N, 3), dtype=torch.float32, device=dev), torch.empty((N, 4), dtype=torch.float32, device=dev),
               torch.empty((N, 3), dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev),
               torch.empty((N, K, 3) if T == 1 else (N, T, K, 3), dtype=torch.float32, device=dev))
        self.phase = "exchange (static): reduction"
        if T == 1:
            call("mtgs_dp_reduce", world, N, K, int(sh_degree), ptr(means), ptr(words_all), ptr(prefix_all), self.meta_len * 4, ptr(recv),
                 row_stride, ptr(cams), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), ptr(out[4]), 0, -1, st)
        else:
            masks = [sum(1 << r for r in range(world) if int(traversal_of_rank[r]) == t) for t in range(T)]
            for t in range(T):
                call("mtgs_dp_reduce_slices", world, N, K, int(sh_degree), ptr(means), ptr(words_all), ptr(prefix_all), self.meta_len * 4,
                     ptr(recv), row_stride, ptr(cams), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]),
                     out[4].data_ptr() + t * K * 3 * 4, 0, -1, _C.c_uint64(masks[t]), int(t == 0), T * K * 3, st)
        self.phase = "idle"
        return out, overflow

    # ---- host-side bookkeeping of finish(): pure functions of the samples every rank holds after the meta all-gather ----------
    def plan(self, samples) -> dict:
        """What finish() derives on the host from `samples[world, n_chunks + 2]` (row count | row index at every chunk boundary
        | traversal of the rank's camera -- the words _sample_idx picks out of every rank's meta record):
          counts[r]; starts[r][c] = first row of index chunk c in rank r's rows (+ the count as the last entry);
          caps[c] = rows per rank in the all-gather of chunk c (the largest chunk of any rank, at least 1: equal sizes for all);
          trav[r]; masks[t] = bit mask of the ranks that rendered traversal t; present = traversals some rank rendered;
          subsets = [all ranks] + [masks[t] for t in present] (subset 0: geometry / coef_all; 1 + j: colour group of present[j]);
          union_caps[i] = upper bound of the rows of subset i's union (sum of its ranks' counts, at most N).
        Identical on every rank (it is a function of the all-gathered samples only), so the ranks issue identical collectives."""
        world, T, nch = self.world, self.T, self.n_chunks
        counts = [int(samples[r][0]) for r in range(world)]
        starts = [[int(samples[r][1 + c]) for c in range(nch)] + [counts[r]] for r in range(world)]
        caps = [max(max(starts[r][c + 1] - starts[r][c] for r in range(world)), 1) for c in range(nch)]
        trav = [int(samples[r][len(self.bounds)]) for r in range(world)]
        masks = [sum(1 << r for r in range(world) if trav[r] == t) for t in range(T)]
        present = [t for t in range(T) if masks[t]]
        subsets = [(1 << world) - 1] + [masks[t] for t in present]
        union_caps = [min(self.N, sum(c for r, c in enumerate(counts) if (m >> r) & 1)) for m in subsets]
        return {"counts": counts, "starts": starts, "caps": caps, "trav": trav, "masks": masks, "present": present,
                "subsets": subsets, "union_caps": union_caps}

    def gather_rows(self, plan: dict):
        """Every chunk's all-gather, issued up front (they queue on the collective stream): chunk c of rank r's rows is
        rows[starts[r][c] : starts[r][c] + caps[c]] -- a FIXED count per rank, so the tail of a short chunk carries rows of the
        next chunk (or slack), which the receiver never reads: it finds a Gaussian's row through the sender's map
        (prefix - starts[r][c]).  Returns (works, recvs[c] [world, caps[c], ROW])."""
        world, dev = self.world, self.device
        works, recvs = [], []
        self.last_bytes = 0
        for c in range(self.n_chunks):
            cap = plan["caps"][c]
            self.phase = f"exchange: issuing the row all-gather of chunk {c} of {self.n_chunks} ({cap} rows per rank)"
            if world > 1 or self.world_collectives:
                recv = torch.empty((world, cap, self.ROW), dtype=torch.float32, device=dev)
                s0 = plan["starts"][self.rank][c]
                works.append(dist.all_gather_into_tensor(recv.view(world * cap, self.ROW), self.rows[s0:s0 + cap],
                                                         group=self.group, async_op=True))
                self.last_bytes += world * cap * self.ROW * 4
            else:
                recv = self.rows[plan["starts"][0][c]:]
                works.append(None)
            recvs.append(recv)
        self.last_bytes += world * self.meta_len * 4 if world > 1 else 0
        return works, recvs

    def finish(self, means: torch.Tensor, sh_degree: int, rows: bool = False, all_colour_ranges=()):
        """After backward(): exchange the wire rows and return (v_means, v_quats, v_scales, v_opacities, v_coeffs) --
        the dense sums over all ranks of the gradients that flowed through `rasterization()`.

        rows = True: the sums leave as compact ROWS of the union of the ranks' visible sets instead (mtgs_dp_union /
        mtgs_dp_reduce_rows): no dense [N, .] tensor -- in particular no [N, T, K, 3] coefficient gradient, 472 MB x T of
        mostly zeros per step at 2M Gaussians -- is written, and an optimizer that takes row gradients
        (FusedAdam.set_row_gradient, row-lazy for the per-traversal tensors) steps what some camera of the step saw.  Returns a
        dict: "geo_rows" [cap, 16] = {v_mean 3, v_quat 4, v_scale 3, v_opacity 1 (with respect to the ACTIVATED Gaussians, summed over
        all ranks) | the gradient of SH coefficient 0 summed over all ranks 3 | 0 | index}, "geo_row_of" int32 [N] (row or -1),
        "geo_ids" int32 [cap], "geo_totals" int64 device (count << 32: mtgs_node_bwd_rows' `totals`); "coef" {t: (rows [cap_t, 3 K],
        row_of int32 [N])} for every traversal some rank rendered (coefficient k channel c at 3 k + c, summed over THAT traversal's
        ranks); all_colour_ranges = [(begin, end)]: index ranges of nodes whose colour parameters are shared by the traversals
        -> "coef_all" (rows, row_of): the sum over ALL ranks, valid inside those ranges.  Every row equals the dense entry
        bit for bit (same summation order)."""
        from ._lib import call, ptr, stream_of
        P = self._pending
        assert P is not None and P["stage"] == "rows", "finish() follows rasterization() + backward()"
        self._pending = None
        N, K, dev, world, nw = self.N, self.K, self.device, self.world, self.n_words
        means = means.detach().contiguous()
        st = stream_of(means)
        self.phase = "exchange: waiting for the meta all-gather (visibility maps)"
        if P["done"] is not None:
            P["done"].synchronize()      # side stream only: finished while the frame was composited
        samples = self._samples_host.numpy()
        metas = P["metas"]
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        cams = metas[:, 1:4].contiguous().view(torch.float32)
        words_all, prefix_all = metas[:, 4:], metas[:, 4 + 2 * nw:]
        T = self.T
        # per-traversal appearance: the senders of traversal t (a bit mask) write slice t of the coefficient gradient
        pl = self.plan(samples)
        masks, present, subsets = pl["masks"], pl["present"], pl["subsets"]
        R = None
        if rows:
            import ctypes as _C
            caps = pl["union_caps"]
            n_sub = len(subsets)
            masks_dev = torch.tensor(np.asarray(subsets, dtype=np.uint64).view(np.int64), dtype=torch.int64, device=dev)
            uw = torch.empty((n_sub, max(nw, 1)), dtype=torch.int64, device=dev)
            up = torch.empty((n_sub, max(nw, 1)), dtype=torch.int32, device=dev)
            totals = torch.empty(n_sub, dtype=torch.int64, device=dev)
            scratch = torch.empty(n_sub * (max(nw, 1) // 256 + 1), dtype=torch.int32, device=dev)
            call("mtgs_dp_union", world, N, ptr(words_all), self.meta_len * 4, n_sub, ptr(masks_dev), ptr(uw), ptr(up), ptr(totals),
                 ptr(scratch), st)
            new = lambda *shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)
            R = {"geo_rows": new(max(caps[0], 1), 16), "geo_row_of": new(N, dt=torch.int32), "geo_ids": new(max(caps[0], 1), dt=torch.int32),
                 "geo_totals": totals[0:1], "coef": {}, "coef_all": None, "caps": caps}
            for j, t in enumerate(present):
                R["coef"][t] = (new(max(caps[1 + j], 1), 3 * K), new(N, dt=torch.int32))
            if all_colour_ranges:
                R["coef_all"] = (new(max(caps[0], 1), 3 * K), new(N, dt=torch.int32))
            out = R
            group_tab = None
            if present:      # every rendered traversal's colour group + the geometry in ONE launch per chunk (mtgs_dp_reduce_rows_groups)
                from .nodes import upload_table
                gt = np.zeros(len(present), dtype=_DP_GROUP)
                for j, t in enumerate(present):
                    gt[j] = (masks[t], R["coef"][t][0].data_ptr(), uw[1 + j].data_ptr(), up[1 + j].data_ptr(), R["coef"][t][1].data_ptr(),
                             R["coef"][t][0].shape[0])
                group_tab = upload_table(gt, dev)
        if rows != True:      # noqa: E712  (False: dense only; "both": the dense tensors beside the rows, from the same wire data: tests)
            out = (torch.empty((N, 3), dtype=torch.float32, device=dev), torch.empty((N, 4), dtype=torch.float32, device=dev),
                   torch.empty((N, 3), dtype=torch.float32, device=dev), torch.empty(N, dtype=torch.float32, device=dev),
                   torch.empty((N, K, 3) if T == 1 else (N, T, K, 3), dtype=torch.float32, device=dev))
        ev = lambda: torch.cuda.Event(enable_timing=True)
        w0, w1, red = ev(), ev(), []
        # every chunk's all-gather is issued up front (they queue on the collective stream); chunk c is reduced as soon as
        # it has arrived, while the later chunks are still on the wire
        w0.record()
        works, recvs = self.gather_rows(pl)
        caps = pl["caps"]
        for c in range(self.n_chunks):
            self.phase = f"exchange: wire + reduction of chunk {c} of {self.n_chunks}"
            if works[c] is not None:
                works[c].wait()
            if c == self.n_chunks - 1:
                w1.record()
            e0, e1 = ev(), ev()
            e0.record()
            if rows:
                rs = caps[c] * self.ROW if (world > 1 or self.world_collectives) else 0
                gb, ge = self.bounds[c], self.bounds[c + 1]

                def reduce_rows(mask, geo, coef, sub, b0=gb, b1=ge):
                    z = lambda t: ptr(t) if t is not None else None
                    call("mtgs_dp_reduce_rows", world, N, K, int(sh_degree), ptr(means), ptr(words_all), ptr(prefix_all),
                         self.meta_len * 4, ptr(recvs[c]), rs, ptr(cams), b0, b1, _C.c_uint64(mask),
                         z(R["geo_rows"] if geo else None), ptr(uw[0]), ptr(up[0]), z(R["geo_row_of"] if geo else None),
                         z(R["geo_ids"] if geo else None), R["geo_rows"].shape[0],
                         z(coef[0] if coef else None), ptr(uw[sub]), ptr(up[sub]), z(coef[1] if coef else None),
                         coef[0].shape[0] if coef else 0, 3 * K, st)
                if not present:
                    reduce_rows(0, True, None, 0)
                elif self.grouped_receiver and len(present) > 1:
                    call("mtgs_dp_reduce_rows_groups", world, N, K, int(sh_degree), ptr(means), ptr(words_all), ptr(prefix_all),
                         self.meta_len * 4, ptr(recvs[c]), rs, ptr(cams), gb, ge, len(present), ptr(group_tab), ptr(R["geo_rows"]),
                         ptr(uw[0]), ptr(up[0]), ptr(R["geo_row_of"]), ptr(R["geo_ids"]), R["geo_rows"].shape[0], 3 * K, st)
                else:
                    for j, t in enumerate(present):      # the first traversal's pass also sums the geometry over all ranks
                        reduce_rows(masks[t], j == 0, R["coef"][t], 1 + j)
                if R["coef_all"] is not None:        # nodes whose colour parameters the traversals share: all ranks, their index range
                    if c == 0:
                        R["coef_all"][1].fill_(-1)
                    # (whole chunks: a sender's received rows start at the CHUNK's first row; rows outside the ranges are
                    #  computed and never read)
                    if any(min(ge, int(e)) > max(gb, int(b)) for (b, e) in all_colour_ranges):
                        reduce_rows(subsets[0], False, R["coef_all"], 0)
            if rows == True:      # noqa: E712
                pass
            elif T == 1:
                call("mtgs_dp_reduce", world, N, K, int(sh_degree), ptr(means), ptr(words_all), ptr(prefix_all), self.meta_len * 4,
                     ptr(recvs[c]), caps[c] * self.ROW if (world > 1 or self.world_collectives) else 0, ptr(cams), ptr(out[0]),
                     ptr(out[1]), ptr(out[2]), ptr(out[3]), ptr(out[4]), self.bounds[c], self.bounds[c + 1], st)
            else:
                import ctypes as _C
                for t in range(T):      # one pass per traversal's slice; the first also sums the geometry over all senders
                    call("mtgs_dp_reduce_slices", world, N, K, int(sh_degree), ptr(means), ptr(words_all), ptr(prefix_all),
                         self.meta_len * 4, ptr(recvs[c]), caps[c] * self.ROW if (world > 1 or self.world_collectives) else 0, ptr(cams), ptr(out[0]),
                         ptr(out[1]), ptr(out[2]), ptr(out[3]), out[4].data_ptr() + t * K * 3 * 4, self.bounds[c],
                         self.bounds[c + 1], _C.c_uint64(masks[t]), int(t == 0), T * K * 3, st)
            e1.record()
            red.append((e0, e1))
        self._events.update(wire=(w0, w1), reduce=red)
        self.phase = "idle"
        return (out, R) if rows == "both" else out

    def phases_ms(self) -> dict:
        """Durations of the last step's exchange phases (synchronises): meta = all-gather of the visibility maps (side
        stream, overlapped with the compositing), wire = first row all-gather issued -> last one complete (includes the
        reductions of the earlier chunks it overlaps with), reduce = the reduction kernels."""
        torch.cuda.synchronize()
        out = {}
        if "meta" in self._events:
            out["meta"] = self._events["meta"][0].elapsed_time(self._events["meta"][1])
        if "wire" in self._events:
            out["wire"] = self._events["wire"][0].elapsed_time(self._events["wire"][1])
            out["reduce"] = sum(a.elapsed_time(b) for a, b in self._events["reduce"])
        return out

    # ---- tensor form ---------------------------------------------------------------------------------------------
    def _exchange_ordered(self, radii, means, cam_pos, v_means, v_quats, v_scales, v_opacities, v_rgb, sh_degree):
        from ._lib import call, ptr, stream_of
        N, K, dev, world = self.N, self.K, self.device, self.world
        st = stream_of(means)
        meta, nw = self.meta, self.n_words
        words, prefix = meta[4:4 + 2 * nw], meta[4 + 2 * nw:]
        meta[1:4].copy_(cam_pos.view(torch.int32))
        call("mtgs_dp_pack_ordered", N, ptr(radii), ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_opacities),
             ptr(v_rgb), ptr(words), ptr(prefix), ptr(meta), ptr(self.block_counts), ptr(self.rows), N, st)  # meta[0] = count
        if world > 1:
            metas = torch.empty((world, self.meta_len), dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(metas, meta[None], group=self.group)
            counts_h = metas[:, 0].tolist()       # host sync: buffer size for the payload exchange
            cap = max(max(counts_h), 1)
            recv = torch.empty((world, cap, self.ROW), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(recv.view(world * cap, self.ROW), self.rows[:cap], group=self.group)
            self.last_bytes = world * (cap * self.ROW * 4 + self.meta_len * 4)
        else:
            metas, recv, cap = meta[None], self.rows, N
            self.last_bytes = 0
        cams = metas[:, 1:4].contiguous().view(torch.float32)
        words_all, prefix_all = metas[:, 4:], metas[:, 4 + 2 * nw:]
        v_coeffs = torch.empty((N, K, 3), dtype=torch.float32, device=dev) if v_rgb is not None else None
        # the dense inputs were copied into the rows: they are overwritten with the sums
        call("mtgs_dp_reduce", world, N, K, int(sh_degree), ptr(means), ptr(words_all), ptr(prefix_all),
             self.meta_len * 4, ptr(recv), cap * self.ROW, ptr(cams), ptr(v_means), ptr(v_quats), ptr(v_scales),
             ptr(v_opacities), ptr(v_coeffs), 0, -1, st)
        return v_means, v_quats, v_scales, v_opacities, v_coeffs

    def exchange(self, radii: torch.Tensor, means: torch.Tensor, cam_pos: torch.Tensor, v_means: torch.Tensor,
                 v_quats: torch.Tensor, v_scales: torch.Tensor, v_opacities: torch.Tensor,
                 v_rgb: Optional[torch.Tensor], sh_degree: int, local_coeff_grad=None, check_contract: bool = False):
        """radii[N] (this rank's camera), means[N,3], cam_pos[3]; this rank's dense RASTERIZER gradients v_* (they are
        OVERWRITTEN with the sums over all ranks); v_rgb[N,3] is the gradient with respect to the SH
        OUTPUT (before the +0.5 / clamp), or None for no SH part.
        Only the rows of Gaussians with radii > 0 are sent (see the class docstring): gradients of other loss terms must
        be added AFTER the exchange.  `check_contract=True` (debugging) asserts that the inputs are zero elsewhere.
        `local_coeff_grad` is only used by the per-sender fallback path (K > 16 or degree > 3): a callable
        returning this rank's own dense v_coeffs[N,K,3], invoked while the payload is in flight.
        Returns (v_means, v_quats, v_scales, v_opacities, v_coeffs | None): dense sums over all ranks."""
        from ._lib import call, ptr, stream_of
        N, K, dev = self.N, self.K, self.device
        radii = radii.reshape(-1).contiguous()
        assert radii.numel() == N and means.shape == (N, 3)
        for t in (v_means, v_quats, v_scales, v_opacities):
            assert t.is_contiguous(), "the local dense gradients are overwritten in place"
        if check_contract:
            hidden = radii <= 0
            for name, t in (("v_means", v_means), ("v_quats", v_quats), ("v_scales", v_scales), ("v_opacities", v_opacities),
                            ("v_rgb", v_rgb)):
                if t is not None and bool((t[hidden] != 0).any()):
                    raise ValueError(f"SparseGradExchange.exchange: {name} is non-zero on Gaussians this rank's camera does not "
                                     "see; only rasterizer gradients may be passed (add regulariser gradients after the exchange)")
        v_rgb = None if v_rgb is None else v_rgb.contiguous()
        means = means.contiguous()
        cam_pos = cam_pos.reshape(3).to(torch.float32).contiguous()
        if self.world <= 64 and (v_rgb is None or (K <= 16 and sh_degree <= 3)):
            return self._exchange_ordered(radii, means, cam_pos, v_means, v_quats, v_scales, v_opacities, v_rgb,
                                          sh_degree)
        # ---- fallback: unordered rows, one read-modify-write pass per remote sender
        st = stream_of(means)
        world = self.world
        rebuild_own = v_rgb is not None and local_coeff_grad is None
        work, cams, recv = None, None, None
        if world > 1 or rebuild_own:
            call("mtgs_dp_pack", N, ptr(radii), ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_opacities), ptr(v_rgb),
                 ptr(self.rows), N, ptr(self.count), st)
        if world > 1:
            # one small collective carries both the row count and the camera position of every rank
            meta = torch.cat([self.count, cam_pos.view(torch.int32).to(torch.int64)])[None]      # [1,4] int64
            metas = torch.empty((world, 4), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(metas, meta, group=self.group)
            cams = metas[:, 1:4].to(torch.int32).contiguous().view(torch.float32)
            counts_h = metas[:, 0].tolist()       # host sync: buffer sizes for the payload exchange
            cap = max(max(counts_h), 1)
            recv = torch.empty((world * cap, self.ROW), dtype=torch.float32, device=dev)
            # [cap,16] blocks along dim 0; asynchronous, so that the local work below overlaps the transfer
            work = dist.all_gather_into_tensor(recv, self.rows[:cap], group=self.group, async_op=True)
            recv = recv.view(world, cap, self.ROW)
            self.last_bytes = world * cap * self.ROW * 4
        else:
            counts_h = [int(self.count.item())] if rebuild_own else [0]
            self.last_bytes = 0
        # this rank's own coefficient gradient: dense and WRITTEN (not accumulated), so it doubles as the zero-fill
        v_coeffs = None
        if v_rgb is not None:
            if local_coeff_grad is not None:
                v_coeffs = local_coeff_grad()
                assert v_coeffs.shape == (N, K, 3) and v_coeffs.is_contiguous()
            else:
                # no local SH backward supplied: expand the own rows' v_rgb through the same kernel (its geometry
                # part goes to a scratch buffer: the own geometry gradients are already in the dense tensors)
                v_coeffs = torch.zeros((N, K, 3), dtype=torch.float32, device=dev)
                scratch = torch.zeros(11 * N, dtype=torch.float32, device=dev)
                s_m, s_q, s_s, s_o = torch.split(scratch, [3 * N, 4 * N, 3 * N, N])
                call("mtgs_dp_accumulate", counts_h[self.rank if world > 1 else 0], ptr(self.rows), N, K, int(sh_degree),
                     ptr(means), ptr(cam_pos), ptr(s_m), ptr(s_q), ptr(s_s), ptr(s_o), ptr(v_coeffs), st)
        if work is not None:
            work.wait()
        for r in range(world):
            if world == 1 or r == self.rank:
                continue
            call("mtgs_dp_accumulate", counts_h[r], ptr(recv[r]), N, K, int(sh_degree), ptr(means), ptr(cams[r]),
                 ptr(v_means), ptr(v_quats), ptr(v_scales), ptr(v_opacities), ptr(v_coeffs), st)
        return v_means, v_quats, v_scales, v_opacities, v_coeffs


def all_reduce_stats(sum_tensors: Iterable[torch.Tensor] = (), max_tensors: Iterable[torch.Tensor] = (),
                     group=None, sum_init: Sequence[float] = ()) -> None:
    """Densification statistics must be identical on every rank before `refinement_after` reads them (SURVEY.md
    section 8e): running grad-norm sums and visibility counts are summed, max screen-space radii are max-reduced
    (/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:457-474).

    CONTRACT: the tensors are each rank's LOCAL running accumulators since the last reset, and this is called exactly
    ONCE per refinement interval, right before the refine decision (after which the reference resets them,
    vanilla_gaussian_splatting.py:571-574).  Calling it twice on the same accumulators would count the other ranks'
    contributions twice -- use `StatsReducer` below when the statistics must also be readable between refinements.
    `sum_init[i]` is the value accumulator i was initialised with (the reference starts `vis_counts` at ONE, :462): it is
    counted once, not once per rank: result = init + sum_r (local_r - init).
    One collective per reduction kind (the tensors are packed), not one per tensor."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    sum_tensors, max_tensors = list(sum_tensors), list(max_tensors)
    inits = list(sum_init) + [0.0] * (len(sum_tensors) - len(sum_init))
    for tensors, op in ((sum_tensors, dist.ReduceOp.SUM), (max_tensors, dist.ReduceOp.MAX)):
        if not tensors:
            continue
        if op == dist.ReduceOp.SUM:
            flat = torch.cat([(t - i0 if i0 else t).reshape(-1) for t, i0 in zip(tensors, inits)])
        else:
            flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.all_reduce(flat, op=op, group=group)
        off = 0
        for k, t in enumerate(tensors):
            part = flat[off:off + t.numel()].view_as(t)
            t.copy_(part + inits[k] if (op == dist.ReduceOp.SUM and inits[k]) else part)
            off += t.numel()


class StatsReducer:
    """Local accumulators + a global view for densification statistics that are read more than once per interval.

    The ranks accumulate into `local_*`; `reduce()` may be called any number of times and always returns
    (sum over ranks of the local sums [+ init], max over ranks of the local maxima) without modifying the local
    accumulators, so repeated calls never double count."""

    def __init__(self, sum_tensors: Sequence[torch.Tensor], max_tensors: Sequence[torch.Tensor], sum_init: Sequence[float] = (),
                 group=None):
        self.local_sum, self.local_max, self.group = list(sum_tensors), list(max_tensors), group
        self.sum_init = list(sum_init) + [0.0] * (len(self.local_sum) - len(sum_init))

    def reduce(self):
        g_sum = [t.clone() for t in self.local_sum]
        g_max = [t.clone() for t in self.local_max]
        all_reduce_stats(g_sum, g_max, group=self.group, sum_init=self.sum_init)
        return g_sum, g_max
