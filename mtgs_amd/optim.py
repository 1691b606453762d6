"""Fused Adam for the Gaussian parameter groups: ONE launch per step over every tensor (csrc/adam.hip).

Reference: nerfstudio builds one `torch.optim.Adam` per parameter group, each with one tensor
(/root/reference/mtgs/scene_model/custom_trainer.py:115-136; groups, learning rates and eps = 1e-15 in
/root/reference/mtgs/config/MTGS.py:121-181), and the densification edits `optimizer.state[param]` by hand
(vanilla_gaussian_splatting.py:392-446).  `FusedAdam` IS a `torch.optim.Optimizer` with torch.optim.Adam's state layout
(`state[p] = {"step", "exp_avg", "exp_avg_sq"}`, `param_groups[i] = {"params", "lr", "betas", "eps", "weight_decay"}`), so
schedulers, `state_dict()` / `load_state_dict()` (interchangeable with torch.optim.Adam's) and that kind of state surgery
keep working; only `step()` differs: every tensor of every group is updated by one kernel that reads p, m, v once and
writes them once (torch's foreach path makes ~10 passes per tensor list).

Two gradient sources per parameter:
  * `p.grad` (dense), as usual;
  * `set_row_gradient(p, rows, row_of, col)`: compact gradient rows of the VISIBLE Gaussians plus a map Gaussian -> row
    (< 0: not visible).  Gaussians without a row get the exact zero-gradient Adam update -- the moments decay, the
    parameter keeps moving along exp_avg -- without a dense gradient tensor ever being written or read.
  * both: a parameter with a row gradient AND a `p.grad` (a loss term that reaches it outside the rasterization: MTGS's
    scale / sharp-shape / out-of-box regularisers, mtgs_scene_graph.py:939-981) is stepped with their sum; for a ROW-LAZY
    parameter a dense gradient is an error (every row would have to be stepped), raised by step().

Exact lazy Adam for per-traversal tensors (`set_lazy_slices(param)` for `[N, T, ...]` parameters: MTGS's features_rest /
features_adapters): a step renders ONE traversal; the other traversals' slices only decay.  A lazy parameter's step touches
the rendered slice alone, and `prepare(t)` -- called before the forward that reads traversal t -- catches slice t up with
the zero-gradient steps it missed (the same operations in the same order as stepping every time: bit-identical), so the
optimizer's traffic no longer grows with the number of traversals.  `flush()` before anything else reads the parameters
(checkpoints, refinement, evaluation with another traversal).

Exact ROW-lazy Adam (`set_row_lazy(param)`) for parameters of which a frame reads the visible rows only -- the SH
coefficients under visibility-first colours (mtgs_amd.nodes.ColorSource): a Gaussian the frame does not see has a zero
gradient, its moments decay and its value drifts along exp_avg, but nothing reads it until it is seen again.  The step then
touches the VISIBLE rows alone, and the forward PEEKS (`ColorSource.optimizer = opt` wires `peek_rows()` between the front end
and the colour kernel): the zero-gradient steps a visible row missed are replayed in registers (the same operations in the same
order: bit-identical) into a compact buffer that the colour kernel reads and that the step takes the parameter values back from;
the optimizer's own state changes in `step()` / `flush()` only.  Optimizer traffic for the coefficients drops from every
Gaussian x every traversal to the visible rows of the rendered traversal.  `flush()` as above; `state_dict()` flushes;
`catch_up_rows()` is the in-place form of the peek for callers that read the parameters themselves.

HIP graphs: the per-step scalars (lr / (1 - beta1^t), sqrt(1 - beta2^t)) live in a small device array that `advance()`
refreshes with one copy; `step()` = `advance()` + the launch.  Capture `step()` once, then per replay call `advance()` and
replay (the copy is enqueued on the stream in front of the graph launch).
No CPU / PyTorch fallback: parameters must be float32 HIP tensors."""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from ._lib import call, load, ptr, stream_of

_GROUP = np.dtype([("p", "<u8"), ("m", "<u8"), ("v", "<u8"), ("g", "<u8"), ("rows", "<u8"), ("row_of", "<u8"), ("catchup", "<u8"),
                   ("last", "<u8"), ("hist", "<u8"), ("row_ids", "<u8"), ("row_count_dev", "<u8"), ("caught", "<u8"), ("sub_index_dev", "<u8"), ("row_flags", "<u8"),
                   ("n", "<i8"), ("first_block", "<i8"), ("row_stride", "<i8"), ("caught_stride", "<i8"), ("item_start", "<i8"), ("n_rows", "<i8"),
                   ("width", "<i4"), ("row_col", "<i4"),
                   ("vec_ok", "<i4"), ("sub_width", "<i4"), ("sub_index", "<i4"), ("mode", "<i4"), ("catchup_k", "<i4"),
                   ("hyper_index", "<i4"), ("caught_col", "<i4"), ("rank_start", "<i4"), ("rank_count", "<i4"), ("zero_probe", "<i4"), ("one_minus_beta1", "<f4"), ("beta2", "<f4"), ("one_minus_beta2", "<f4"), ("eps", "<f4"),
                   ("weight_decay", "<f4"), ("grad_scale", "<f4")], align=True)


def _put_slice(r, t) -> None:
    """The slice of a per-traversal tensor in a table row: a host int, or an int32 DEVICE scalar that the kernel reads when it
    runs (one captured step / peek for every traversal: the caller rewrites the word in front of a replay)."""
    if isinstance(t, torch.Tensor):
        if t.dtype != torch.int32 or t.numel() != 1 or not t.is_cuda:
            raise ValueError("a device slice index is one int32 on the GPU")
        r["sub_index"], r["sub_index_dev"] = 0, t.data_ptr()
    else:
        r["sub_index"], r["sub_index_dev"] = int(t), 0


MODE_DENSE, MODE_SLICE, MODE_ROWS_CATCHUP, MODE_ROWS_STEP, MODE_ROWS_FLUSH, MODE_ROWS_PEEK = 0, 1, 2, 3, 4, 5   # MTGS_ADAM_*
_checked = False


def _check_layout():
    global _checked
    if not _checked:
        want = load().mtgs_adam_group_bytes()
        if want != _GROUP.itemsize:
            raise RuntimeError(f"mtgs_adam_group is {want} bytes in libmtgs_rast.so, {_GROUP.itemsize} in mtgs_amd.optim")
        _checked = True


def _list_blocks(groups) -> int:
    """Upper bound of the workgroups of LIST-form row groups [(item_start, n_rows)] (include/mtgs_rast.h: tensors of one node share
    their count, the counts of different nodes add up to at most the number of rows)."""
    if not groups:
        return 0
    per = load().mtgs_adam_block_list_rows()
    per_node = {}
    for start, _ in groups:
        per_node[start] = per_node.get(start, 0) + 1
    return max(per_node.values()) * -(-max(n for _, n in groups) // per) + len(groups)


def _check_row_ids(row_ids):
    if row_ids is None:
        return None
    ids, start, count = row_ids
    if ids.dtype != torch.int32 or ids.dim() != 1 or not ids.is_contiguous() or \
            (count is not None and (count.dtype != torch.int64 or count.numel() < 1)):
        raise ValueError("row_ids = (int32 [R] contiguous, start, int64 device scalar | None)")
    return ids, int(start), count


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False,
                 nontemporal=True):
        if amsgrad or maximize:
            raise NotImplementedError("FusedAdam: amsgrad / maximize (MTGS's AdamOptimizerConfig uses neither)")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.nontemporal = bool(nontemporal)
        self.grad_scale = 1.0          # every gradient is multiplied by this inside the kernel (1 / world for a DDP-style mean)
        self._rows = {}                # id(param) -> (rows, row_of, col, stride, width, sub_width, sub_index, caught, row_ids)
        self._rows_more = {}           # id(param) -> further slices' sources of the same step (row-lazy per-traversal tensors)
        self._lazy = {}                # id(param) -> {"param", "T", "last": [step up to which slice t is current], "hist": [(step_size, bc2_sqrt)]}
        self._active_slice = {}        # id(param) -> slice the coming step() updates (lazy parameters)
        self._rowlazy = {}             # id(param) -> {"param", "T", "last": int32 [N * T], "hist": float32 [2 * cap], "cap"}
        self._hyper_index = {}         # id(param) -> row of the device hyper array (position in the active list)
        self._captured = False         # a step() was captured in a HIP graph: the host no longer knows which steps have run
        self._pending_host = False     # eager: _advance() ran, the launch did not yet
        self._catch_key = self._catch_table = None
        self._table_key = None
        self._table_dev = self._hyper_dev = None
        self._active = []
        self._blocks = 0

    # ---- gradient source 2 -------------------------------------------------------------------------------------------
    def set_row_gradient(self, param: torch.Tensor, rows: torch.Tensor, row_of: torch.Tensor, col: int = 0,
                         slice_index: Optional[int] = None, caught=None, row_ids=None, zero_probe: Optional[int] = None,
                         row_flags: Optional[torch.Tensor] = None) -> None:
        """For the NEXT step, `param[N, ...]`'s gradient is `rows[row_of[n], col : col + width]` (width = elements per
        Gaussian of param) where row_of[n] >= 0 and zero elsewhere; `rows` float32 [R, stride] (row-contiguous), `row_of`
        int32 [N].  slice_index = t for a per-traversal tensor `param[N, T, ...]`: only `param[:, t]` takes the row (width =
        elements of one slice), the other traversals get the zero gradient.  A `param.grad` present at step() is ADDED (terms
        outside the rasterization); row-lazy parameters refuse one.
        caught: row-lazy parameters -- the up-to-date rows peek_rows() left for THIS frame (same row numbering); the step then
        takes the parameter from them and only replays the moments of the missed steps.
        row_ids = (ids int32 [R] increasing, start, count | None): row-lazy parameters -- the frame's list of visible Gaussians
        (global index of every rank; this parameter's items are start .. start + N - 1; count: device int64 whose upper half is
        the number of valid ranks, mtgs_front_fwd's totals).  With it the rows are taken straight from the list (2-3x faster
        than scanning the row map).
        zero_probe = c: row-lazy parameters -- a row whose floats rows[r, c : c + 3] are all zero has an all-zero gradient and is
        left lazy by the step (the caller guarantees the implication: ColorSource's rows hold C0 * v_rgb there).  Most
        frustum-visible Gaussians are occluded and get no gradient; skipping them is exact -- the zero-gradient update is what
        the next catch-up replays.
        row_flags (uint8 [>= R]; row-lazy parameters in the LIST form): rows with a zero flag have a zero gradient BY CONSTRUCTION
        (the frame composites nothing from them: mtgs_blend_touch_packed) and are left lazy without anything of theirs being
        read -- unlike zero_probe without a bound on how far behind they fall: the forward that passes the same flags to
        peek_rows() does not peek them either.  Cleared by step() / zero_grad()."""
        width = param.numel() // max(param.shape[0], 1) if param.dim() else 1
        sub_w, sub_i = 0, 0
        if slice_index is not None:
            T = param.shape[1]
            if isinstance(slice_index, torch.Tensor):        # a device word (its value is the caller's promise: 0 <= t < T)
                if slice_index.dtype != torch.int32 or slice_index.numel() != 1 or not slice_index.is_cuda:
                    raise ValueError("set_row_gradient: a device slice_index is one int32 on the GPU")
                sub_w, sub_i = width // T, slice_index
            else:
                if not 0 <= int(slice_index) < T:
                    raise ValueError("set_row_gradient: slice_index")
                sub_w, sub_i = width // T, int(slice_index)
        if rows.dtype != torch.float32 or row_of.dtype != torch.int32 or row_of.numel() != param.shape[0]:
            raise ValueError("set_row_gradient: rows float32 [R, stride], row_of int32 [N]")
        if rows.dim() != 2 or rows.stride(1) != 1 or col < 0 or col + (sub_w or width) > rows.shape[1] or not row_of.is_contiguous():
            raise ValueError("set_row_gradient: row layout")
        if caught is not None:    # (buffer [R', stride] float32, column): this frame's peek_rows() output for the parameter
            cb, cc = caught
            if cb.dtype != torch.float32 or cb.dim() != 2 or cb.stride(1) != 1 or cb.shape[0] < rows.shape[0] or \
                    cc < 0 or cc + (sub_w or width) > cb.shape[1]:
                raise ValueError("set_row_gradient: caught = (float32 [R' >= R, stride], column)")
        if zero_probe is not None and not (0 <= int(zero_probe) and int(zero_probe) + 3 <= rows.shape[1]):
            raise ValueError("set_row_gradient: zero_probe")
        if row_flags is not None and (row_flags.dtype != torch.uint8 or row_flags.numel() < rows.shape[0] or not row_flags.is_contiguous()
                                      or row_ids is None):
            raise ValueError("set_row_gradient: row_flags uint8 [>= R], with row_ids (the LIST form)")
        src = (rows, row_of, int(col), int(rows.stride(0)), int(width), sub_w, sub_i, caught, _check_row_ids(row_ids),
               -1 if zero_probe is None else int(zero_probe), row_flags)
        first = self._rows.get(id(param))
        if first is not None and sub_w > 0 and first[5] > 0 and not isinstance(sub_i, torch.Tensor) and \
                not isinstance(first[6], torch.Tensor) and sub_i not in [first[6]] + [e[6] for e in self._rows_more.get(id(param), [])]:
            # ANOTHER slice of the same per-traversal tensor in the same step (data-parallel steps render several traversals:
            # every traversal's senders give that slice its own rows and row map) -- row-lazy parameters only (step())
            self._rows_more.setdefault(id(param), []).append(src)
        else:
            self._rows[id(param)] = src
            self._rows_more.pop(id(param), None)

    def zero_grad(self, set_to_none: bool = True):
        self._rows.clear()
        self._rows_more.clear()
        return super().zero_grad(set_to_none=set_to_none)

    # ---- exact lazy Adam for per-traversal tensors ----------------------------------------------------------------------
    def set_lazy_slices(self, param: torch.Tensor) -> None:
        """`param[N, T, ...]`: step() updates only the slice of the step's traversal (set_active_slice / the slice_index of
        set_row_gradient); the other slices are caught up by prepare(t) / flush().  Call before the first step."""
        assert param.dim() >= 2 and param.is_contiguous()
        # "last" is filled in at the first use from the parameter's step count at that time: every slice is taken to be current
        # then (a new optimizer that inherits state after a refinement: flush() the old one first)
        self._lazy[id(param)] = {"param": param, "T": int(param.shape[1]), "last": None, "hist": {}}

    def _last(self, L):
        if L["last"] is None:
            st = self.state.get(L["param"], {})
            s0 = int(float(st["step"])) if "step" in st else 0
            L["last"] = [s0] * L["T"]
        return L["last"]

    def set_active_slice(self, param: torch.Tensor, t: int) -> None:
        self._active_slice[id(param)] = int(t)

    def _catch_up(self, items) -> None:
        """items: [(lazy record, slice)]: apply the zero-gradient steps each slice missed, one launch for all of them."""
        _check_layout()
        elems = load().mtgs_adam_block_elems()
        todo = []
        for L, t in items:
            p = L["param"]
            st = self.state.get(p)
            if not st or "exp_avg" not in st:
                continue
            cur = int(float(st["step"]))
            k = cur - self._last(L)[t]
            if k > 0:
                todo.append((L, t, k, cur))
        if not todo:
            return
        hist = np.concatenate([np.asarray([L["hist"][j] for j in range(cur - k + 1, cur + 1)], np.float32).reshape(-1)
                               for L, t, k, cur in todo])
        dev = todo[0][0]["param"].device
        from .nodes import upload_table
        hist_dev = upload_table(hist, dev).view(torch.float32)
        tab = np.zeros(len(todo), _GROUP)
        fb, off = 0, 0
        for i, (L, t, k, cur) in enumerate(todo):
            p = L["param"]
            st = self.state[p]
            grp = next(g for g in self.param_groups if any(q is p for q in g["params"]))
            width = p.numel() // p.shape[0]
            sw = width // L["T"]
            r = tab[i]
            r["p"], r["m"], r["v"] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            r["n"], r["first_block"], r["width"], r["sub_width"], r["sub_index"], r["mode"] = p.shape[0] * sw, fb, width, sw, t, MODE_SLICE
            r["catchup"], r["catchup_k"] = hist_dev.data_ptr() + 4 * off, k
            b1, b2 = grp["betas"]
            r["one_minus_beta1"], r["beta2"], r["one_minus_beta2"] = 1.0 - b1, b2, 1.0 - b2
            r["eps"], r["weight_decay"], r["grad_scale"] = grp["eps"], grp["weight_decay"], self.grad_scale
            fb += -(-(p.shape[0] * sw) // elems)
            off += 2 * k
            L["last"][t] = cur
        table = upload_table(tab, dev)
        dummy = torch.zeros(4, dtype=torch.float32, device=dev)     # (hyper row 0: not used by catch-up groups)
        call("mtgs_adam_step", len(todo), ptr(table), ptr(dummy), fb, fb, int(self.nontemporal), stream_of(todo[0][0]["param"]))

    def prepare(self, t: int) -> None:
        """Before the forward that reads traversal t: bring slice t of every lazy parameter up to date."""
        self._catch_up([(L, int(t)) for L in self._lazy.values() if 0 <= int(t) < L["T"]])

    def flush(self) -> None:
        """Every slice of every lazy parameter up to date (checkpoints, refinement, evaluation)."""
        self._catch_up([(L, t) for L in self._lazy.values() for t in range(L["T"])])
        self._flush_rows()

    # ---- exact row-lazy Adam (visible rows only) ---------------------------------------------------------------------------
    def row_lazy_state(self, param: torch.Tensor):
        """(last int32 [N * T], hist float32 [2 * capacity], capacity) of a row-lazy parameter, or None -- what a refinement that
        moves rows WITHOUT flushing carries over to the next optimizer (set_row_lazy(..., last=, hist=))."""
        RL = self._rowlazy.get(id(param))
        return None if RL is None else (RL["last"], RL["hist"], RL["cap"])

    def set_row_lazy(self, param: torch.Tensor, traversals: Optional[int] = None, hist_capacity: int = 1 << 17,
                     last: Optional[torch.Tensor] = None, hist: Optional[torch.Tensor] = None) -> None:
        """`param[N, ...]` (traversals=None) or `param[N, T, ...]` (traversals=T: the step's set_row_gradient names the slice):
        from now on step() updates only the rows the step's row map marks visible, catch_up_rows() brings rows up to date
        before a forward reads them, flush() all of them.  The parameter's gradient must come through set_row_gradient().
        hist_capacity: optimizer steps the per-step scalar history holds (8 bytes each; grown on demand outside HIP graphs).
        last / hist: the stamps (int32 [N * T], moved with their rows) and the history of the optimizer this one continues (a
        refinement that did not flush: rows stay as lazy as they were); default: every row current as of the step count."""
        if not param.is_contiguous() or param.dtype != torch.float32 or not param.is_cuda:
            raise RuntimeError("FusedAdam.set_row_lazy: contiguous float32 HIP parameter")
        T = 1 if traversals is None else int(traversals)
        if traversals is not None and (param.dim() < 2 or param.shape[1] != T):
            raise ValueError("set_row_lazy: traversals must equal param.shape[1]")
        st = self.state.get(param, {})
        s0 = int(float(st["step"])) if "step" in st else 0
        cap = max(int(hist_capacity), s0 + 2)
        if last is not None or hist is not None:
            if last is None or hist is None or last.dtype != torch.int32 or last.numel() != param.shape[0] * T or \
                    not last.is_contiguous() or hist.dtype != torch.float32 or hist.numel() < 2 * (s0 + 2):
                raise ValueError("set_row_lazy: last int32 [N * T] contiguous and hist float32 [2 * capacity] go together")
            self._rowlazy[id(param)] = {"param": param, "T": T, "cap": hist.numel() // 2, "last": last.reshape(-1), "hist": hist}
            return
        self._rowlazy[id(param)] = {"param": param, "T": T, "cap": cap,
                                    "last": torch.full((param.shape[0] * T,), s0, dtype=torch.int32, device=param.device),
                                    "hist": torch.zeros(2 * cap, dtype=torch.float32, device=param.device)}

    def forget(self, param: torch.Tensor) -> None:
        """Drop everything held for `param` (row-lazy / lazy records with their `last` / `hist` buffers, a pending row gradient,
        its state): for parameters a refinement REPLACED when the optimizer itself lives on.  flush() first if the old values
        are still to be read.  (The records hold their parameter, so an id is never reused while a record exists.)"""
        for d in (self._rowlazy, self._lazy, self._rows, self._rows_more, self._active_slice, self._hyper_index):
            d.pop(id(param), None)
        self.state.pop(param, None)
        self._table_key = self._catch_key = None

    def is_row_lazy(self, param: torch.Tensor) -> bool:
        """True when set_row_lazy() was called for this parameter."""
        return id(param) in self._rowlazy

    def _rows_target(self, p) -> int:
        """The step zero-gradient catch-up goes up to: the steps already TAKEN.  Known to the host in eager use; once a step
        has been captured in a HIP graph only the device knows (< 0: the kernel reads it next to the step's scalars)."""
        if self._captured or torch.cuda.is_current_stream_capturing():
            return -1
        return int(float(self.state[p]["step"])) - (1 if self._pending_host else 0)

    def _rows_groups(self, items, mode, out=None):
        """Descriptor rows for [(row-lazy record | parameter, row_of | None, slice[, column of `out`])] -> (table, blocks) or
        None.  A bare parameter (PEEK only) is a tensor without lazy state: its rows are copied as they are."""
        _check_layout()
        per_block = load().mtgs_adam_block_rows()
        recs = []
        for it in items:
            RL, ro, t = it[0], it[1], it[2]
            col = it[3] if len(it) > 3 else 0
            rid = _check_row_ids(it[4]) if len(it) > 4 else None
            lazy = isinstance(RL, dict)
            p = RL["param"] if lazy else RL
            st = self.state.get(p) if lazy else None
            stateful = bool(st) and "exp_avg" in st and self._hyper_dev is not None
            if not stateful:
                if mode != MODE_ROWS_PEEK:
                    continue                   # no step taken yet: nothing to catch up
                recs.append((p, None, None, ro, t, 0, 0, col, rid))
                continue
            target = self._rows_target(p)
            hi = self._hyper_index.get(id(p))
            if hi is None:
                if target < 0:
                    raise RuntimeError("FusedAdam: a row-lazy parameter that the captured step does not update")
                hi = 0                         # (explicit target: the hyper row is not read)
            recs.append((p, RL, st, ro, t, target, hi, col, rid))
        if not recs:
            return None
        recs.sort(key=lambda rec: rec[8] is not None and mode != MODE_ROWS_FLUSH)      # (stable: LIST-form groups last)
        tab = np.zeros(len(recs), _GROUP)
        fb = 0
        lists = []
        for i, (p, RL, st, ro, t, target, hi, col, rid) in enumerate(recs):
            grp = next((g for g in self.param_groups if any(q is p for q in g["params"])),
                       {"betas": (0.0, 0.0), "eps": 0.0, "weight_decay": 0.0})      # (a tensor of another optimizer: copied)
            N = p.shape[0]
            width = p.numel() // max(N, 1)
            r = tab[i]
            r["p"], r["n"], r["first_block"], r["width"] = p.data_ptr(), N, fb, width
            if st is not None:
                r["m"], r["v"] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                r["last"], r["hist"] = RL["last"].data_ptr(), RL["hist"].data_ptr()
            if t is not None and p.dim() >= 2 and (RL["T"] > 1 if RL is not None else True):
                T = RL["T"] if RL is not None else p.shape[1]
                r["sub_width"] = width // T
                _put_slice(r, t)
            if out is not None:
                r["caught"], r["caught_stride"], r["caught_col"], r["n_rows"] = out.data_ptr(), out.stride(0), int(col), out.shape[0]
            if ro is not None:
                if ro.dtype != torch.int32 or ro.numel() != N or not ro.is_contiguous():
                    raise ValueError("catch_up_rows: row_of int32 [N]")
                r["row_of"] = ro.data_ptr()
            r["mode"], r["catchup_k"], r["hyper_index"] = mode, target, hi
            b1, b2 = grp["betas"]
            r["one_minus_beta1"], r["beta2"], r["one_minus_beta2"] = 1.0 - b1, b2, 1.0 - b2
            r["eps"], r["weight_decay"], r["grad_scale"] = grp["eps"], grp["weight_decay"], self.grad_scale
            if rid is not None and mode != MODE_ROWS_FLUSH:
                ids, start, count = rid
                cap = out.shape[0] if out is not None else ids.numel()
                r["row_ids"], r["item_start"], r["n_rows"] = ids.data_ptr(), start, min(cap, ids.numel())
                r["row_count_dev"] = 0 if count is None else count.data_ptr()
                lists.append((start, int(r["n_rows"])))
            else:
                fb += -(-N // per_block)
        return tab, fb + _list_blocks(lists), bool(lists)

    def catch_up_rows(self, items) -> None:
        """items: [(param, row_of int32 [N], slice | None[, row_ids])] -- the rows with row_of >= 0 of every row-lazy parameter among them
        are brought up to date (the zero-gradient steps since they were last touched), one launch.  Called by the forward
        between the front end (which knows the visible Gaussians) and the kernel that reads their coefficients."""
        todo = []
        for it in items:
            p, ro, t = it[0], it[1], it[2]
            RL = self._rowlazy.get(id(p))
            if RL is not None:
                if RL["T"] > 1 and t is None:
                    raise ValueError("catch_up_rows: a per-traversal parameter needs its slice")
                todo.append((RL, ro, 0 if t is None else int(t), 0, it[3] if len(it) > 3 else None))
        built = self._rows_groups(todo, MODE_ROWS_CATCHUP)
        if built is None:
            return
        tab, blocks, any_list = built
        key = tab.tobytes()
        if key != self._catch_key or torch.cuda.is_current_stream_capturing():
            from .nodes import upload_table
            self._catch_table, self._catch_key = upload_table(tab, todo[0][0]["param"].device), key
        call("mtgs_adam_step", len(tab), ptr(self._catch_table), ptr(self._hyper_dev), blocks, 0, 2 if any_list else 0,
             stream_of(todo[0][0]["param"]))

    def peek_rows(self, items, out: torch.Tensor, row_flags: Optional[torch.Tensor] = None) -> None:
        """items: [(param, row_of int32 [N], slice | None, column[, row_ids])] (row_ids as in set_row_gradient: the fast LIST form)
        -- for every Gaussian with row_of[n] = r >= 0 the UP-TO-DATE row
        of the parameter (its slice) is written to out[r, column : column + width]: row-lazy parameters are caught up in
        registers (nothing in the optimizer changes: a forward stays free of side effects), other tensors are copied.  `out`
        float32 [R, stride]; ranks >= R are skipped.  One launch.  Hand `out` back through set_row_gradient(caught=...) and
        the step reuses the caught-up parameter values instead of recomputing them.
        row_flags (uint8 [>= R], LIST-form items only): rows with a zero flag are left alone -- out[r] is NOT written (the caller
        does not read it: mtgs_blend_touch_packed's flags of the Gaussians the frame composites from)."""
        if out.dtype != torch.float32 or out.dim() != 2 or out.stride(1) != 1:
            raise ValueError("peek_rows: out float32 [R, stride]")
        if row_flags is not None and (row_flags.dtype != torch.uint8 or row_flags.numel() < out.shape[0] or not row_flags.is_contiguous()):
            raise ValueError("peek_rows: row_flags uint8 [>= rows of out]")
        todo = []
        for it in items:
            p, ro, t, col = it[0], it[1], it[2], it[3]
            RL = self._rowlazy.get(id(p))
            if RL is not None and RL["T"] > 1 and t is None:
                raise ValueError("peek_rows: a per-traversal parameter needs its slice")
            todo.append((RL if RL is not None else p, ro, t, int(col), it[4] if len(it) > 4 else None))
        built = self._rows_groups(todo, MODE_ROWS_PEEK, out)
        if built is None:
            return
        tab, blocks, any_list = built
        if row_flags is not None:
            tab["row_flags"][tab["row_ids"] != 0] = row_flags.data_ptr()
        dev = out.device
        hyper = self._hyper_dev if self._hyper_dev is not None else torch.zeros(4, dtype=torch.float32, device=dev)
        key = tab.tobytes()
        if key != self._catch_key or torch.cuda.is_current_stream_capturing():
            from .nodes import upload_table
            self._catch_table, self._catch_key = upload_table(tab, dev), key
        self._peek_keep = (out, hyper, row_flags)
        call("mtgs_adam_step", len(tab), ptr(self._catch_table), ptr(hyper), blocks, 0, 2 if any_list else 0, stream_of(out))

    def _flush_rows(self) -> None:
        items = [(RL, None, t) for RL in self._rowlazy.values() for t in range(RL["T"])]
        built = self._rows_groups(items, MODE_ROWS_FLUSH)
        if built is None:
            return
        tab, blocks, _ = built
        from .nodes import upload_table
        dev = items[0][0]["param"].device
        call("mtgs_adam_step", len(tab), ptr(upload_table(tab, dev)), ptr(self._hyper_dev), blocks, 0, 0, stream_of(items[0][0]["param"]))

    def state_dict(self):
        self.flush()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for RL in list(self._rowlazy.values()):      # everything loaded is current as of its step count
            self.set_row_lazy(RL["param"], None if RL["T"] == 1 else RL["T"], RL["cap"])
        for L in self._lazy.values():
            L["last"], L["hist"] = None, {}

    # ---- the step ----------------------------------------------------------------------------------------------------
    def _collect(self):
        """The tensors this step updates (torch.optim.Adam: those with a gradient), with lazily created state."""
        act = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                src = self._rows.get(id(p))
                if p.grad is None and src is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("FusedAdam: float32 HIP parameters only (no CPU / PyTorch fallback)")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g = None
                if p.grad is not None:
                    if src is not None and id(p) in self._rowlazy:
                        raise RuntimeError("FusedAdam: a row-lazy parameter has a dense .grad next to its row gradient (a loss term outside "
                                           "the rasterization reaches it: every row would have to be stepped) -- detach that term or "
                                           "do not make the parameter row-lazy")
                    g = p.grad
                    if g.is_sparse:
                        raise RuntimeError("FusedAdam: sparse gradients are passed with set_row_gradient()")
                    if g.dtype != torch.float32 or g.shape != p.shape:
                        raise RuntimeError("FusedAdam: gradient dtype / shape")
                    if not g.is_contiguous():
                        g = g.contiguous()
                act.append((gi, p, st, g, src))
        return act

    def _table(self, act):
        """Descriptor table on the device; rebuilt when a pointer, a size or a static hyper-parameter changed (in eager
        training autograd allocates new gradient tensors every iteration; under a HIP graph everything is static)."""
        _check_layout()
        elems = load().mtgs_adam_block_elems()
        # one table row per (tensor, gradient source): a per-traversal tensor that several traversals of the step gave rows
        # (set_row_gradient called once per slice) has one row-lazy group per slice -- disjoint elements, the same device scalars
        entries = []
        for i, (gi, p, st, g, src) in enumerate(act):
            entries.append((i, src))
            for extra in self._rows_more.get(id(p), []):
                if id(p) not in self._rowlazy:
                    raise RuntimeError("FusedAdam: row gradients for several slices of one tensor in one step need a row-lazy "
                                       "parameter (set_row_lazy(param, traversals=T))")
                entries.append((i, extra))
        tab = np.zeros(len(entries), _GROUP)
        fb = 0
        keep = []
        # table order: the row-lazy tensors last (they run as a second kernel, mtgs_adam_step's rows_from_block); the device
        # scalars stay indexed by the position in `act` (hyper_index)
        def rank(e):      # streaming groups, then row groups found through the row map, then row groups with an id list
            p, src = act[e[0]][1], e[1]
            return 0 if id(p) not in self._rowlazy else (2 if (src is not None and src[8] is not None) else 1)
        order = sorted(entries, key=rank)
        self._rows_from = None
        self._list_groups = False
        lists = []
        for j, (i, src) in enumerate(order):
            gi, p, st, g, _ = act[i]
            grp = self.param_groups[gi]
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if m.shape != p.shape or v.shape != p.shape or not m.is_contiguous() or not v.is_contiguous():
                raise RuntimeError("FusedAdam: exp_avg / exp_avg_sq must be contiguous and shaped like their parameter")
            r = tab[j]
            if id(p) in self._rowlazy and self._rows_from is None:
                self._rows_from = fb
            r["p"], r["m"], r["v"], r["n"], r["first_block"] = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), fb
            align = p.data_ptr() | m.data_ptr() | v.data_ptr()
            if g is not None:
                r["g"] = g.data_ptr()
                align |= g.data_ptr()
                keep.append(g)
            if src is not None:         # (with a dense gradient too: the kernel adds them)
                rows, row_of, col, stride, width, sub_w, sub_i, caught, rid, probe, rflags = src
                r["rows"], r["row_of"], r["row_col"], r["row_stride"], r["width"] = rows.data_ptr(), row_of.data_ptr(), col, stride, width
                r["n_rows"] = rows.shape[0]
                r["sub_width"] = sub_w
                _put_slice(r, sub_i)
                keep.append((rows, row_of))
            r["vec_ok"] = int(align % 16 == 0)
            r["hyper_index"] = i
            RL = self._rowlazy.get(id(p))
            if RL is not None:      # row-lazy: the visible rows alone (scan of the row map: ceil(N / rows per block) workgroups)
                if src is None:
                    raise RuntimeError("FusedAdam: a row-lazy parameter takes its gradient through set_row_gradient()")
                if (RL["T"] > 1) != (src[5] > 0) or (src[5] > 0 and src[4] // src[5] != RL["T"]):
                    raise RuntimeError("FusedAdam: row-lazy parameter and the slice of its row gradient do not match")
                r["mode"], r["last"], r["hist"], r["n"] = MODE_ROWS_STEP, RL["last"].data_ptr(), RL["hist"].data_ptr(), p.shape[0]
                # (the bound on the staleness of a skipped row: four visits of its slice -- a slice is rendered every T-th step)
                r["zero_probe"] = ((src[9] + 1) | (min(4 * RL["T"], 0x7fff) << 16)) if (src[9] >= 0 and grp["weight_decay"] == 0) else 0
                if src[7] is not None and grp["weight_decay"] == 0:
                    r["caught"], r["caught_stride"], r["caught_col"] = src[7][0].data_ptr(), src[7][0].stride(0), int(src[7][1])
                    keep.append(src[7][0])
                if src[8] is not None:       # LIST form: one row per 16 lanes straight from the frame's list of visible Gaussians
                    ids, start, count = src[8]
                    r["row_ids"], r["item_start"] = ids.data_ptr(), start
                    r["row_count_dev"] = 0 if count is None else count.data_ptr()
                    lists.append((start, min(int(rows.shape[0]), int(ids.numel()))))     # (workgroups: assigned on the device)
                    keep.append((ids, count))
                    if src[10] is not None:
                        r["row_flags"] = src[10].data_ptr()
                        keep.append(src[10])
                    self._list_groups = True
                else:
                    fb += -(-int(p.shape[0]) // load().mtgs_adam_block_rows())
                b1, b2 = grp["betas"]
                r["one_minus_beta1"], r["beta2"], r["one_minus_beta2"] = 1.0 - b1, b2, 1.0 - b2
                r["eps"], r["weight_decay"], r["grad_scale"] = grp["eps"], grp["weight_decay"], self.grad_scale
                continue
            L = self._lazy.get(id(p))
            if L is not None:       # lazy per-traversal tensor: this step touches the rendered slice alone
                t = self._active_slice.get(id(p), src[6] if (src is not None and src[5] > 0) else None)
                if t is None:
                    raise RuntimeError("FusedAdam: a lazy per-traversal parameter needs its slice (set_active_slice / slice_index)")
                width = p.numel() // p.shape[0]
                sw = width // L["T"]
                r["width"], r["sub_width"], r["mode"], r["n"] = width, sw, MODE_SLICE, p.shape[0] * sw
                _put_slice(r, t)      # (a host int or an int32 device scalar: the kernel's slice_of() reads either)
            b1, b2 = grp["betas"]
            r["one_minus_beta1"], r["beta2"], r["one_minus_beta2"] = 1.0 - b1, b2, 1.0 - b2   # (differences taken in double)
            r["eps"], r["weight_decay"], r["grad_scale"] = grp["eps"], grp["weight_decay"], self.grad_scale
            fb += -(-int(r["n"]) // elems)
        key = tab.tobytes()
        # (while capturing the table is always uploaded: the copy becomes part of the graph and its destination lives in the
        #  graph's memory pool -- a cached table of an earlier eager step would be freed when the next graph replaces it)
        if key != self._table_key or torch.cuda.is_current_stream_capturing():
            from .nodes import upload_table
            dev = act[0][1].device
            self._table_dev = upload_table(tab, dev)
            self._table_key = key
        if self._rows_from is None:
            self._rows_from = fb
        fb += _list_blocks(lists)
        self._blocks = fb
        self._keep = keep
        self._n_table = len(entries)
        return self._table_dev

    def inherit_layout(self, old: "FusedAdam") -> None:
        """After a refinement replaced the parameters: `self` was built over the NEW parameters with the same groups in the same
        order as `old`, and every stepped tensor has its moments already.  Takes over what an eager step() would otherwise have
        to establish before a step() can be CAPTURED: the device scalars (one row per stepped tensor) and each tensor's row in
        them, by position in the groups.  advance() fills the scalars in front of the first replay."""
        if not old._active or old._hyper_dev is None:
            return
        if self._hyper_dev is None or self._hyper_dev.numel() != 4 * len(old._active):
            self._hyper_dev = torch.zeros(4 * len(old._active), dtype=torch.float32, device=old._hyper_dev.device)
        for g_new, g_old in zip(self.param_groups, old.param_groups):
            for p_new, p_old in zip(g_new["params"], g_old["params"]):
                hi = old._hyper_index.get(id(p_old))
                if hi is not None:
                    self._hyper_index[id(p_new)] = hi

    def host_state(self):
        """The optimizer's HOST-side per-step state (the step counters the bias corrections are computed from, the lazy slices'
        bookkeeping): what a body that failed half way -- GraphedIteration's capture without warm-up -- has to get back before the
        step is run again (set_host_state)."""
        import copy
        steps = {id(p): (st["step"].clone() if torch.is_tensor(st.get("step")) else st.get("step"))
                 for g in self.param_groups for p in g["params"] for st in [self.state.get(p, {})] if "step" in st}
        lazy = {k: {"last": list(v["last"]), "hist": dict(v["hist"])} for k, v in self._lazy.items() if "last" in v and "hist" in v}
        return {"steps": steps, "lazy": lazy, "pending": self._pending_host, "active_slice": copy.copy(self._active_slice)}

    def set_host_state(self, s) -> None:
        for g in self.param_groups:
            for p in g["params"]:
                if id(p) in s["steps"] and p in self.state:
                    v = s["steps"][id(p)]
                    self.state[p]["step"] = v.clone() if torch.is_tensor(v) else v
        for k, v in s["lazy"].items():
            if k in self._lazy:
                self._lazy[k]["last"], self._lazy[k]["hist"] = list(v["last"]), dict(v["hist"])
        self._pending_host = s["pending"]
        self._active_slice = dict(s["active_slice"])

    def advance(self, active_slice: Optional[int] = None) -> None:
        """In front of every replay of a HIP graph that captured step(): increments the step count of the tensors that
        step updates and copies this step's {lr / (1 - beta1^t), sqrt(1 - beta2^t)} per tensor to the device
        (asynchronously, from a fresh pinned buffer, on the current stream).  active_slice: the traversal the replayed
        graph renders (lazy per-traversal parameters; call prepare(t) before the replay as before an eager forward)."""
        self._advance(self._active, active_slice)

    def _advance(self, act, active_slice=None) -> None:
        if not act:
            return
        hyper = np.zeros((len(act), 4), np.float32)      # {step_size, bc2_sqrt, step (int32), pending (int32)} per tensor
        hyper_i = hyper.view(np.int32)
        for i, (gi, p, st, g, src) in enumerate(act):
            grp = self.param_groups[gi]
            if id(p) in self._lazy:
                self._last(self._lazy[id(p)])        # (first use: every slice current as of the step count BEFORE this step)
            st["step"] += 1
            t = float(st["step"])
            b1, b2 = grp["betas"]
            hyper[i, 0] = grp["lr"] / (1.0 - b1 ** t)
            hyper[i, 1] = math.sqrt(1.0 - b2 ** t)
            hyper_i[i, 2], hyper_i[i, 3] = int(t), 1
            RL = self._rowlazy.get(id(p))
            if RL is not None and int(t) >= RL["cap"]:      # room for this step's scalars (the step launch appends them)
                if self._captured:
                    raise RuntimeError("FusedAdam: the row-lazy history is full; set_row_lazy(..., hist_capacity=) before capturing")
                grown = torch.zeros(4 * RL["cap"], dtype=torch.float32, device=p.device)
                grown[:2 * RL["cap"]] = RL["hist"]
                RL["hist"], RL["cap"] = grown, 2 * RL["cap"]
            L = self._lazy.get(id(p))
            if L is not None:
                step_i = int(t)
                L["hist"][step_i] = (float(hyper[i, 0]), float(hyper[i, 1]))
                ts = active_slice if active_slice is not None else \
                    self._active_slice.get(id(p), src[6] if (src is not None and src[5] > 0) else None)
                if ts is not None:
                    if L["last"][ts] != step_i - 1:
                        raise RuntimeError(f"FusedAdam: slice {ts} of a lazy parameter is {step_i - 1 - L['last'][ts]} steps behind: "
                                           "call prepare(t) before the forward of traversal t")
                    L["last"][ts] = step_i
                oldest = min(L["last"])
                for j in [j for j in L["hist"] if j <= oldest]:
                    del L["hist"][j]
        if self._hyper_dev is None or self._hyper_dev.numel() != hyper.size:
            self._hyper_dev = torch.empty(hyper.size, dtype=torch.float32, device=act[0][1].device)
        staged = torch.empty(hyper.size, dtype=torch.float32, pin_memory=True)
        staged.numpy()[:] = hyper.reshape(-1)
        self._hyper_dev.copy_(staged, non_blocking=True)
        self._pending_host = True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        act = self._collect()
        self._active = act
        if not act:
            return loss
        index = {id(a[1]): i for i, a in enumerate(act)}
        if torch.cuda.is_current_stream_capturing() and any(self._hyper_index.get(k, v) != v for k, v in index.items()):
            # (the forward of this capture peeked with the rows inherit_layout() handed over)
            raise RuntimeError("FusedAdam: the captured step orders its tensors differently from the optimizer whose layout was inherited")
        self._hyper_index = index
        if not torch.cuda.is_current_stream_capturing():
            self._advance(act)
        elif self._hyper_dev is None or self._hyper_dev.numel() != 4 * len(act):
            # (pinned allocation is not permitted while capturing, and the step count must not advance at capture time)
            raise RuntimeError("FusedAdam: run one eager step() before capturing one (state and device buffers are created there)")
        table = self._table(act)
        call("mtgs_adam_step", self._n_table, ptr(table), ptr(self._hyper_dev), self._blocks, self._rows_from,
             int(self.nontemporal) | (2 if self._list_groups else 0), stream_of(act[0][1]))
        self._rows.clear()
        self._rows_more.clear()
        self._active_slice.clear()
        self._pending_host = False
        if torch.cuda.is_current_stream_capturing():
            self._captured = True
        return loss


def adam_reference_step(p, m, v, g, step: int, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam's single-tensor update written out (amsgrad = False, maximize = False), any dtype / device --
    documentation of the arithmetic the kernel restates; the tests compare against torch.optim.Adam itself."""
    b1, b2 = betas
    if weight_decay:
        g = g + weight_decay * p
    m = m + (g - m) * (1 - b1)
    v = v * b2 + (1 - b2) * g * g
    denom = v.sqrt() / math.sqrt(1 - b2 ** step) + eps
    return p - (lr / (1 - b1 ** step)) * (m / denom), m, v
