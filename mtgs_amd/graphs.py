"""Training THROUGH HIP graphs: capture an iteration once per stretch of fixed shapes, replay it per step.

What MTGS runs per step -- `MTGSSceneModel.get_outputs` -> `get_loss_dict` -> backward -> optimizers.step ->
`update_submodel_statistics` (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:547-708, 806-987, 1157-1183) -- is ~110 kernel
launches and a dozen library calls around ~1 ms of GPU work at 960x540; launched from Python the step is bound by the host.
Between two refinements (`refinement_after`, vanilla_gaussian_splatting.py:448-577: every `refine_every` = 100 steps) the number
of Gaussians is fixed, so the whole iteration can be ONE graph launch.  `GraphedIteration` is the manager of that scheme:

  * a STRETCH = the steps between two refinements.  The first time a key (a traversal / camera; or one key for everything, see
    below) comes up in a stretch its iteration runs once eagerly under `mtgs_amd.graph_mode` -- that IS the training step, and it
    fills the mode object's pinned staging buffers -- and is captured right after (`torch.cuda.graph`: nothing executes); every
    later step of that key is `before_replay()` (e.g. `FusedAdam.advance()`) + one graph launch.  From the second stretch on the
    warm-up is skipped: the staging buffers of the key's previous graph are reused, the capture comes first and its first replay
    is the step.
  * CAPACITIES (visible Gaussians, tile intersections; `mtgs_amd.graph_mode(cap_vis, cap_M)`): the first stretch renders every key
    once the ordinary way (exact sizes; `rasterization()`'s size plan learns n_vis / M of this scene), later stretches scale the
    largest counts their predecessor saw ON THE DEVICE by the growth of N.
  * OVERFLOW: every frame's `info["overflow"]` flag is OR-ed into one device word; `poll(i)` copies it to pinned host memory every
    `poll_every` steps behind an event that is only QUERIED (the host never waits).  On overflow the graphs are dropped, every key
    renders one frame the ordinary way and is captured again with larger capacities.  (A truncated frame is never out of bounds;
    the few steps between the overflow and its detection trained on truncated tile lists.)
  * ONE MEMORY POOL for every graph of the run: a graph's private pool is hipMalloc'ed at capture and released with the graph --
    per key and per refinement that was most of the cost of re-capturing at 2M Gaussians.  Sharing is safe because the graphs are
    replayed one at a time on one stream, each replay writes everything it reads, and the only output read afterwards (the loss)
    is copied out in stream order right behind its replay.  A trivial keeper graph holds the pool across refinements.

Where MTGS plugs in: `body(key)` is one call of the trainer's train-iteration for the camera `key` --
zero_grad, `get_outputs`, `get_loss_dict`, backward, optimizer step, `update_submodel_statistics` -- returning `(loss, info)` with
`info` the rasterization's meta dict; `after_refinement()` is called behind `refinement_after`.  Everything `body` reads that
changes from step to step (the camera, the targets, the step's learning rates) must live in tensors whose VALUES are rewritten in
front of the launch -- per key (one graph per traversal) or gathered from stacked tensors through a device index (one graph for
every traversal).

Reference for what the iteration contains: scripts/mtgs_like_train.py::train_loop (BASELINE configs[4] on synthetic data), which
is a caller of this class.
"""
from __future__ import annotations

from typing import Callable, Dict, Hashable, Optional, Tuple

import torch

from . import wrapper


class GraphedIteration:
    """See the module docstring.

    body(key) -> (loss, info): ONE whole training iteration for `key`; `loss` a 0-d tensor (its storage is the graph's static
        output: copy it out in stream order), `info` the meta dict of the iteration's `rasterization()` call.  The body must not
        park the step's tensors (`info`, the render, the loss graph) in objects that outlive the call: `info["means2d"]` holds the
        step's autograd graph, and an object of the warm-up pass that dies INSIDE the capture that follows takes the capture down
        with it (hipGraph instantiation segfaults -- tests/test_gpu_graphs.py was written the wrong way first).  What the caller
        needs from a step (the loss, statistics) goes into tensors it owns, in stream order.
    n_keys: how many different keys a stretch sees (graphs per stretch; the eager frames that teach the size plan).
    size_key() -> (C, N, width, height): the key of `rasterization()`'s size plan for the CURRENT parameters.
    before_replay(): called in front of every graph launch (`FusedAdam.advance`: this step's bias corrections / learning rates,
        one small copy enqueued in front of the launch).
    on_capacities(): called when a stretch's capacities have just been planned from eager frames (a policy hook).
    can_skip_warmup(): True when a capture without a warm-up pass is valid (the optimizer's device scalars exist already).
    snapshot_host_state() / restore_host_state(s): the body's HOST side effects (optimizer step counters) around a capture without
        warm-up that fails: rolled back before the step is repeated (`FusedAdam.host_state` / `set_host_state`).
    margin: capacities = margin x the counts seen + a constant; first_cap_scale < 1 makes the FIRST capacities too small (tests).
    tight_lists: True = build the opt-in tight tile lists inside the iterations (a trainer's choice: same pixels and gradients,
        shorter lists); False = gsplat's lists; None = whatever mode the calling thread is in."""

    def __init__(self, body: Callable[[Hashable], Tuple[torch.Tensor, Dict]], n_keys: int, size_key: Callable[[], Tuple[int, int, int, int]],
                 device, before_replay: Optional[Callable[[], None]] = None, on_capacities: Optional[Callable[[], None]] = None,
                 can_skip_warmup: Optional[Callable[[], bool]] = None, poll_every: int = 16, margin: float = 1.3,
                 first_cap_scale: float = 1.0, tight_lists: Optional[bool] = True, log: Callable[..., None] = print,
                 tick: Optional[Callable[[str], None]] = None, snapshot_host_state: Optional[Callable[[], object]] = None,
                 restore_host_state: Optional[Callable[[object], None]] = None):
        self.body, self.n_keys, self.size_key, self.device = body, int(n_keys), size_key, torch.device(device)
        self.before_replay, self.on_capacities, self.can_skip_warmup = before_replay, on_capacities, can_skip_warmup
        self.poll_every, self.margin, self.tight, self.log = int(poll_every), float(margin), tight_lists, log
        self.tick = tick or (lambda name: None)
        # host state the body advances per call (the optimizer's Python step counters): snapshot() in front of a capture without
        # warm-up, restore(snapshot) when that capture fails and the step is run again (FusedAdam.host_state / set_host_state)
        self.snapshot_host_state, self.restore_host_state = snapshot_host_state, restore_host_state
        assert (snapshot_host_state is None) == (restore_host_state is None)
        self._cap_scale = float(first_cap_scale)
        self.graphs: Dict[Hashable, tuple] = {}
        self.caps: Optional[Tuple[int, int]] = None
        self.eager_left = self.n_keys
        self.staged: Dict[Hashable, list] = {}      # per key: the pinned staging buffers of its previous graph
        self.seen_dev = torch.zeros(2, dtype=torch.int64, device=self.device)    # largest n_visible / n_intersections of the graph frames
        self.ovf_dev = torch.zeros((), dtype=torch.bool, device=self.device)     # OR of the graph frames' overflow flags
        self.ovf_host = torch.zeros((), dtype=torch.bool).pin_memory()
        self._ovf_ev = None
        self.counts = {"captures": 0, "warmups": 0, "overflows": 0, "eager": 0, "replays": 0}
        self.pool = torch.cuda.graph_pool_handle()
        self._pool_keeper = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._pool_keeper, pool=self.pool):
            self._keep = torch.zeros(1, device=self.device) + 1

    # ---- one iteration, with the bookkeeping of its frame -------------------------------------------------------------------
    def _run(self, key):
        with wrapper.tight_lists(wrapper.lists_are_tight() if self.tight is None else self.tight):
            loss, info = self.body(key)
        if info is not None and "overflow" in info:       # (graph mode: the counts and the flag are device scalars)
            self.ovf_dev.logical_or_(info["overflow"])
            torch.maximum(self.seen_dev, torch.stack([info["n_visible"], info["n_intersections"]]), out=self.seen_dev)
        # (detached: a loss with its grad_fn would keep the iteration's autograd graph -- and its AccumulateGrad nodes -- alive into
        #  the capture that follows the warm-up pass)
        return loss.detach()

    def _plan_caps(self) -> Tuple[int, int]:
        n_vis, M = wrapper._size_plan.seen[tuple(self.size_key())]
        k, self._cap_scale = self._cap_scale, 1.0
        return int(self.margin * k * n_vis) + 4096, int(self.margin * k * M) + 65536

    def step(self, key) -> torch.Tensor:
        """The training step of `key`: eager while the stretch's capacities are unknown, then captured, then replayed."""
        if self.caps is None:                    # the size plan does not know this N yet: ordinary frames (exact sizes)
            loss = self._run(key)
            self.counts["eager"] += 1
            self.eager_left -= 1
            if self.eager_left <= 0:
                self.caps = self._plan_caps()
                if self.on_capacities is not None:
                    self.on_capacities()
            return loss
        if key not in self.graphs:
            self.tick("other")
            gm = wrapper.graph_mode(*self.caps)
            if key in self.staged and (self.can_skip_warmup is None or self.can_skip_warmup()):
                # a later stretch: capture at once and let the first replay BE the step.  What a warm-up would provide is there
                # already: the pinned staging buffers of the key's previous graph (same sequence of table sizes -- a mismatch would
                # allocate pinned memory while capturing and is caught below)
                gm.keep = self.staged[key]
                snap = self.snapshot_host_state() if self.snapshot_host_state is not None else None
                try:
                    g = torch.cuda.CUDAGraph()
                    with gm, torch.cuda.graph(g, pool=self.pool):
                        static = self._run(key)
                except RuntimeError as e:
                    # The EXPECTED failure only: something allocated or synchronised while capturing (a staging buffer of another size
                    # than the previous graph's: "operation not permitted when stream is capturing" and its relatives).  Nothing has
                    # executed on the device, but the body's HOST side effects have happened: the caller's hook rolls them back
                    # (FusedAdam: the step counters its bias corrections are computed from) before the step is run again the
                    # warm-up way.  Anything else -- a failed launch, a HIP error, a bug in the body -- propagates.
                    msg = str(e).lower()
                    if not any(s in msg for s in ("captur", "not permitted", "pinned", "cudamallochost", "hipmallochost", "hiphostmalloc")):
                        raise
                    self.log(f"capture without warm-up failed ({type(e).__name__}: {e}); warming up")
                    self.counts["capture_fallbacks"] = self.counts.get("capture_fallbacks", 0) + 1
                    self.staged.pop(key, None)
                    torch.cuda.synchronize()
                    if snap is not None:
                        self.restore_host_state(snap)
                    return self.step(key)
                self.graphs[key] = (g, gm, static)
                self.counts["captures"] += 1
                self.tick("capture")
                if self.before_replay is not None:
                    self.before_replay()
                g.replay()
                self.counts["replays"] += 1
                self.tick("first replay")
                return static
            with gm:                             # THE step of this iteration, eagerly, with the graph's capacities (also fills the
                loss = self._run(key)            #   mode object's staging buffers: no pinned allocation while capturing).  On the
            #                                        main stream: a side stream has its own allocator pool, every tensor of the step
            #                                        would be hipMalloc'ed afresh there (10 ms per warm-up at 2M Gaussians)
            self.tick("warm")
            g = torch.cuda.CUDAGraph()
            with gm, torch.cuda.graph(g, pool=self.pool):      # nothing executes
                static = self._run(key)
            self.graphs[key] = (g, gm, static)
            self.staged[key] = gm.keep
            self.counts["captures"] += 1
            self.counts["warmups"] += 1
            self.tick("capture")
            return loss
        g, _, static = self.graphs[key]
        if self.before_replay is not None:
            self.before_replay()                 # this step's scalars: one small copy in front of the launch
        g.replay()
        self.counts["replays"] += 1
        return static

    # ---- overflow: never blocks ----------------------------------------------------------------------------------------------
    def poll(self, i: int) -> bool:
        """Queries the event of the previous poll's copy, then (every `poll_every` steps) issues the next one.  True when an
        overflow was found: the graphs are dropped and the next `n_keys` steps run eagerly to re-learn the sizes."""
        found = False
        if self._ovf_ev is not None and self._ovf_ev.query():
            self._ovf_ev = None
            if bool(self.ovf_host):
                self.counts["overflows"] += 1
                self.log(f"step {i}: a graph frame exceeded its capacities {self.caps}; re-capturing")
                self.graphs.clear()
                self.caps, self.eager_left = None, self.n_keys
                self.ovf_dev.zero_()
                found = True
        if self._ovf_ev is None and self.graphs and i % self.poll_every == 0:
            self.ovf_host.copy_(self.ovf_dev, non_blocking=True)
            self._ovf_ev = torch.cuda.Event()
            self._ovf_ev.record()
        return found

    # ---- a refinement ended the stretch ------------------------------------------------------------------------------------------
    def after_refinement(self, n_before: int, n_after: int) -> None:
        """New parameters: new graphs, capacities = the largest counts of the last stretch scaled by the growth of N (the caller's
        refinement has synchronised already: the read of the two counts costs nothing)."""
        self.graphs.clear()
        self._ovf_ev = None
        # an overflow behind the last poll must not vanish with the flag: the read costs nothing here (seen_dev.tolist() below
        # synchronises anyway).  The stretch is over, so there is nothing to re-capture -- but the steps since the overflow trained
        # on truncated tile lists, and the counts the new capacities are scaled from are the TRUE ones (the binning reports a
        # frame's counts whether or not they fitted), so they stay valid.
        if bool(self.ovf_dev):
            self.counts["overflows"] += 1
            self.counts["overflows_found_at_refinement"] = self.counts.get("overflows_found_at_refinement", 0) + 1
            self.log(f"refinement: a graph frame of the last stretch had exceeded its capacities {self.caps} after the last poll "
                     "(its steps used truncated tile lists); the new capacities follow the counts it reported")
        seen = self.seen_dev.tolist()
        ratio = n_after / max(n_before, 1)
        self.caps = ((int(self.margin * ratio * seen[0]) + 4096, int(self.margin * ratio * seen[1]) + 65536) if seen[0] > 0 else None)
        self.eager_left = 0 if self.caps is not None else self.n_keys
        self.seen_dev.zero_()
        self.ovf_dev.zero_()

    def overflowed(self) -> bool:
        """(synchronises) a frame overflowed after the last poll"""
        return bool(self.ovf_dev)

    def close(self) -> None:
        self.graphs.clear()
