#!/usr/bin/env python3
"""Headline benchmark: rendered Mpix/s (forward + backward) of the MTGS rasterization hot path at
2M Gaussians, 1920x1080, on N MI355X GPUs (BASELINE.json `metric`).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one camera per rank, inputs resident in HBM:
    spherical_harmonics(3, dirs, coeffs[N,16,3])  ->  clamp(rgb + 0.5)              (as MTGS does:
    rasterization(..., render_mode="RGB+ED", rasterize_mode="antialiased",          vanilla_gaussian_splatting.py:309-322,
                  absgrad=True, packed=False)                                        mtgs_scene_graph.py:641-662)
    backward of  L = sum(render * Gc) + sum(alpha * Ga)  with fixed random cotangents, down to the
    gradients of means / quats / scales / opacities / SH coefficients / viewmat,
    and, for N > 1, ONE all-reduce of all Gaussian gradients (RCCL over xGMI).
Each rank renders a different camera (yaw = rank * 45 deg) of the same replicated 2M-Gaussian
WB-v1 scene, so per-GPU work is fixed as N grows: "scaling": "weak".

Launch modes (N = 1): the W + K steps run twice -- launched eagerly from Python (every kernel and library call of the step: the
HIP events around the dominant kernel live in this loop) and as ONE HIP GRAPH LAUNCH per step (the same step captured once under
mtgs_amd.graph_mode + torch.cuda.graph and replayed: same kernels, same inputs, nothing from the host but the launch).  The eager loop
follows the box's CPU load (1.00 ... 1.22 ms measured on one box within minutes, next to 0.94 ms of GPU work); the line carries the
faster of the two and names it in config.launch, both are in ms_per_step_eager / ms_per_step_graph.

The step is the caller's code as MTGS writes it; what the library makes of it is the library's business and is named in
config.colour_activation: spherical_harmonics() and the clamp return a DEFERRED tensor and rasterization() evaluates SH + clamp for
the Gaussians its projection finds visible (same render bit for bit).  Beside `ms_per_step` the line carries the same step with SH +
clamp as one kernel over all N (`ms_per_step_sh_over_all_gaussians`), with the clamp as PyTorch's kernels
(`ms_per_step_torch_activation`: what rounds 1-5 measured) and on the opt-in tight tile lists (`ms_per_step_tight_lists`).

Rank 0 prints ONE JSON line; see the task contract for the fields.  `roofline` describes the
dominant kernel (compositing backward) with its duration measured live by HIP events on the
launch stream; `cpu_baseline` times oracle/gsplat_oracle.c (the CPU restatement, "port") on the
host cores for the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

_T0 = time.time()       # (process start, before the first `import torch`)
# (the CPU oracle's OpenMP workers must not spin between its parallel regions: the iteration cells below are child processes whose
#  host-bound phases -- graph captures, refinements -- measured 25 % slower next to 128 spinning threads of this process)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# Issue cost per wave64 instruction per SIMD on gfx950, shader cycles, measured (profiles/r06_valu_ceiling.md)
VALU_COST = {"plain": 2.7, "dpp": 3.9, "swap": 7.5, "trans": 8.3}


def valu_ceiling(active_over_insts, kernel=""):
    """Cost-weighted issue ceiling (cycles per VALU instruction per SIMD) of a kernel's own mix: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU - 1 is
    the share of the 2-weight opcodes (transcendentals, permlane swaps: ~8 cycles), the compositing backward adds 7 DPP adds per 152."""
    if not active_over_insts:
        return None
    heavy = max(0.0, float(active_over_insts) - 1.0)
    dpp = 7.0 / 152.0 if kernel.startswith("blend_bwd") else 0.0
    heavy_cost = 0.5 * (VALU_COST["swap"] + VALU_COST["trans"])
    return round((1.0 - heavy - dpp) * VALU_COST["plain"] + heavy * heavy_cost + dpp * VALU_COST["dpp"], 2)


DOMINANT = ("mtgs_blend_bwd_packed", "mtgs_blend_bwd")   # the compositing backward (packed-record / gather form)
# every C-ABI entry point the step calls, with the kernels behind it (timed live in a second, untimed pass)
ENTRY_POINTS = {
    "mtgs_sh_fwd": ["sh_fwd_k16_kernel<3>"],
    "mtgs_front_fwd": ["front_project_kernel", "front_compact_kernel"],
    "mtgs_bin3_build": ["bin3_rows_count_kernel", "bin3_rows_place_kernel", "bin3_tiles_count_kernel", "bin3_tiles_place_kernel",
                        "bin3_sort_small_kernel", "bin3_sort_large_kernel", "zero"],
    "mtgs_blend_fwd_packed": ["blend_fwd_kernel<4, 2, true>"],
    "mtgs_blend_bwd_packed": ["blend_bwd_kernel<4, 4, true>"],
    "mtgs_project_bwd": ["project_bwd_vis_kernel", "project_bwd_expand_kernel"],
    # (round 6: the dense gradients are cleared beside the compositing backward's work and the per-visible pass writes the rows with a
    #  gradient to their places: no streaming pass; MTGS_ZEROED_OUTPUTS=0 gives the entry point above back)
    "mtgs_project_bwd_zeroed": ["project_bwd_vis_kernel", "viewmat_reduce_kernel"],
    "mtgs_project_bwd_rows": ["project_bwd_rows_kernel"],
    "mtgs_sh_bwd": ["sh_bwd_kernel<3>"],
    "mtgs_sh_bwd_rows": ["sh_bwd_rows_kernel<3>"],
    # (round 6: the step's `torch.clamp(sh + 0.5, 0, 1)` is fused into the SH kernels -- mtgs_amd/wrapper.py::_LazySH -- so these are
    #  the entry points the step calls; MTGS_SH_LAZY=0 gives the plain ones above and PyTorch's six elementwise kernels)
    "mtgs_sh_fwd_act": ["sh_fwd_k16_kernel<3>"],
    "mtgs_sh_bwd_act": ["sh_bwd_kernel<3>"],
    "mtgs_sh_bwd_rows_act": ["sh_bwd_rows_kernel<3>"],
    # (... and since the clamp stays deferred too, the rasterization evaluates SH + activation for the VISIBLE Gaussians by itself:
    #  these two replace the three above in the default step; mtgs_amd.sh_lazy(True, raster=False) gives those back)
    "mtgs_vis_color_fwd_dirs": ["vis_color_fwd_kernel<3>"],
    "mtgs_vis_color_bwd_dirs": ["vis_color_bwd_kernel<3>"],
    "mtgs_dp_reduce": ["dp_reduce_kernel"],
}



def _colour_activation_note():
    import mtgs_amd.wrapper as w
    if not getattr(w, "_lazy_sh_enabled", False):
        return "PyTorch elementwise kernels (MTGS_SH_LAZY=0)"
    if getattr(w, "_lazy_raster_enabled", False):
        return ("deferred into the rasterization: spherical_harmonics() and the step's own `torch.clamp(sh + 0.5, 0.0, 1.0)` "
                "(vanilla_gaussian_splatting.py:313-318) return a deferred tensor, and rasterization(colors=that) evaluates SH + clamp "
                "for the Gaussians its projection finds VISIBLE only, straight into their records -- the same render bit for bit "
                "(mtgs_amd/wrapper.py::_LazySH.raster_source, csrc/viscolor.hip, tests/test_gpu_sh_raster.py); "
                "ms_per_step_sh_over_all_gaussians = the same step with SH + clamp as one kernel over all N, "
                "ms_per_step_torch_activation = with the clamp as PyTorch's kernels")
    return ("fused: spherical_harmonics() returns a deferred tensor and the step's own `torch.clamp(sh + 0.5, 0.0, 1.0)` "
            "(vanilla_gaussian_splatting.py:318) runs inside the SH kernels, bit-identical values and gradients "
            "(mtgs_amd/wrapper.py::_LazySH, tests/test_gpu_sh_lazy.py)")

def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-gaussians", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--variant", choices=["mtgs", "lean"], default="mtgs",
                    help="mtgs: SH deg 3 + RGB+ED/antialiased/absgrad (what MTGS.py drives); "
                         "lean: colours given, RGB/classic/no absgrad")
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed CPU-oracle steps (0 disables)")
    ap.add_argument("--dp-exchange", choices=["sparse", "dense"], default="sparse",
                    help="N > 1: sparse = all-gather of 64-byte rows of the visible Gaussians, SH-coefficient "
                         "gradients rebuilt from their rank-1 factors (mtgs_amd.dist.SparseGradExchange); "
                         "dense = plain all-reduce of every gradient tensor")
    ap.add_argument("--dp-no-prezero", action="store_true", help="N > 1, touched forms: the reduction writes every dense gradient completely "
                    "(round 5) instead of only the touched Gaussians into tensors the frame's compositing forward cleared (round 6)")
    ap.add_argument("--dp-chunks", type=int, default=2, help="N > 1, --dp-finish touched-chunked / dynamic: index chunks of the exchange "
                    "(one all-gather each; the reduction of chunk c overlaps the wire time of chunks c + 1 ..)")
    ap.add_argument("--dp-finish", choices=["touched-chunked", "touched", "static", "dynamic"], default="touched-chunked",
                    help="N > 1, sparse exchange: touched-chunked (default since round 6) = SparseGradExchange.finish_touched_chunked: the rows "
                         "that carry a gradient in --dp-chunks index chunks, one fixed-capacity all-gather each issued back to back, chunk c "
                         "reduced while c + 1 .. are on the wire, no host read (profiles/r06_dp_budget.md); touched = SparseGradExchange.finish_touched: only the wire rows that CARRY a "
                         "gradient travel (40 %% of the visible ones at this scene), with their own map, in ONE fixed-capacity all-gather "
                         "per step -- no visibility-map exchange, no host read between render and reduce; static = finish_static (ONE "
                         "fixed-capacity all-gather of every visible row, capacity = the ranks' largest warm-up row count + 5 %%, "
                         "visibility maps all-gathered during the frame); dynamic = finish() (row counts read on the host, chunked "
                         "all-gathers pipelined with the reduction)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph", help="N = 1: graph (default): the K steps are timed twice -- "
                    "launched eagerly (~40 kernel launches and ~15 library calls per step from Python) and as ONE HIP graph launch each "
                    "(the step captured once under mtgs_amd.graph_mode) -- and the line carries the faster, named in config.launch; "
                    "eager: the eager loop only")
    ap.add_argument("--no-tight", action="store_true", help="skip the second graph timing pass on the opt-in tight tile lists (profiling runs: "
                                                             "the kernel trace then holds the default call's kernels only)")
    ap.add_argument("--no-also", action="store_true", help="skip the untimed extras (forward-only rate, shipped 7-channel cells): "
                                                            "profiling runs, so that the trace holds the headline step only")
    return ap.parse_args()


def build_inputs(args, rank, device):
    from mtgs_amd.synthetic import make_camera, make_scene
    sh = 3 if args.variant == "mtgs" else None
    sc = make_scene(args.n_gaussians, seed=args.seed, sh_degree=sh)
    vm, K = make_camera(args.width, args.height, yaw_deg=45.0 * rank)
    g = torch.Generator(device="cpu").manual_seed(1)
    D_out = 4 if args.variant == "mtgs" else 3
    Gc = torch.randn(1, args.height, args.width, D_out, generator=g)
    Ga = torch.randn(1, args.height, args.width, 1, generator=g)
    host = dict(sc, viewmat=vm, K=K, Gc=Gc, Ga=Ga)
    dev = {k: v.to(device) for k, v in host.items()}
    return host, dev


def make_step(args, dev, world):
    from mtgs_amd import rasterization, spherical_harmonics
    from mtgs_amd.dist import SparseGradExchange, all_reduce_grads
    names = ["means", "quats", "scales", "opacities"] + (["coeffs"] if args.variant == "mtgs" else ["colors"])
    params = {n: dev[n].requires_grad_(True) for n in names}
    viewmat = dev["viewmat"].requires_grad_(True)
    K, Gc, Ga = dev["K"], dev["Gc"], dev["Ga"]
    cam_pos = torch.inverse(dev["viewmat"].detach())[0, :3, 3]
    W, H = args.width, args.height
    all_params = list(params.values()) + [viewmat]
    info_box = {"grad_bytes": 0}
    sparse = world > 1 and args.dp_exchange == "sparse" and args.variant == "mtgs"
    exchange = SparseGradExchange(args.n_gaussians, 16, dev["means"].device, chunks=max(1, min(16, args.dp_chunks))) if sparse else None
    if sparse and args.dp_finish in ("touched", "touched-chunked"):
        exchange.defer_maps = True       # the touched rows' map travels with the rows: no visibility-map exchange during the frame
        exchange.prezero = not args.dp_no_prezero      # the dense sums' zeros ride on the compositing forward; the reduction writes touched Gaussians only

    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("start", "rows", "end")}
    info_box["events"] = ev

    def step():
        for p in all_params:
            p.grad = None
        capturing = info_box.get("capture", False)     # (timing events cannot be recorded into a HIP graph)
        if not capturing:
            ev["start"].record()
        if sparse:
            # ONE exchange per step: the sum of the Gaussian gradients over the ranks (cameras).  The exchange renders
            # (MTGS's colour activation fused), its backward leaves 64-byte wire rows of the visible Gaussians, and
            # finish() all-gathers them in chunks and rebuilds every dense gradient -- the SH-coefficient gradient of
            # every rank's rows, this rank's included, from its rank-1 factors (mtgs_amd.dist.SparseGradExchange)
            dirs = params["means"].detach() - cam_pos
            sh_out = spherical_harmonics(3, dirs, params["coeffs"].detach())
            render, alpha, info = exchange.rasterization(params["means"], params["quats"], params["scales"],
                                                         params["opacities"], sh_out, viewmat, K, W, H, cam_pos,
                                                         render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
            torch.autograd.backward([render, alpha], [Gc, Ga])
            ev["rows"].record()
            if args.dp_finish == "dynamic":
                g = exchange.finish(params["means"], 3)
            elif args.dp_finish == "touched-chunked":
                # per-chunk capacities agreed ONCE, in the first (warm-up) step, from that step's own per-chunk counts (one tiny MAX
                # all-reduce: setup, not part of a timed step); afterwards nothing of the exchange reaches the host
                caps = info_box.get("chunk_caps")
                first = caps is None
                if first:
                    caps = [args.n_gaussians] * exchange.n_chunks         # (generous: the warm-up step's messages are not the timed ones)
                g, ovf = exchange.finish_touched_chunked(params["means"], 3, caps, [0] * world)
                if first:
                    cnt = exchange.touched_chunk_counts().to(torch.float64)
                    torch.distributed.all_reduce(cnt, op=torch.distributed.ReduceOp.MAX)
                    info_box["chunk_caps"] = [int(c * 1.10) + 256 for c in cnt.tolist()]
                    info_box["static_cap"], info_box["touched"] = sum(info_box["chunk_caps"]), True
                else:
                    info_box["overflow"] = ovf if info_box.get("overflow") is None else (info_box["overflow"] | ovf)
            else:
                # no host read, no host wait: ONE fixed-capacity all-gather per step, counts stay on the device.  The capacity is
                # agreed on ONCE, in the first (warm-up) step, from that step's own counts (a tiny MAX all-reduce: setup, not part
                # of a timed step) -- the dynamic finish() is never run in these modes, so the first multi-GPU execution
                # exercises exactly the collectives of the timed steps
                touched = args.dp_finish == "touched"
                cap = info_box.get("static_cap")
                if cap is None:
                    t_cap = torch.tensor([float(exchange.n_vis)], dtype=torch.float64, device=dev["means"].device)
                    torch.distributed.all_reduce(t_cap, op=torch.distributed.ReduceOp.MAX)
                    cap = info_box["static_cap"] = int(float(t_cap.item()) * 1.05) + 1024      # (rows of the VISIBLE Gaussians)
                fin = exchange.finish_touched if touched else exchange.finish_static
                g, ovf = fin(params["means"], 3, cap, [0] * world)
                if touched and not info_box.get("touched"):
                    # ... and, for the touched form, on the rows that carry a gradient: this step's largest count + 10 %
                    t_cap = exchange.touched_count.to(torch.float64).reshape(1)
                    torch.distributed.all_reduce(t_cap, op=torch.distributed.ReduceOp.MAX)
                    info_box["static_cap"], info_box["touched"] = int(float(t_cap.item()) * 1.10) + 1024, True
                info_box["overflow"] = ovf if info_box.get("overflow") is None else (info_box["overflow"] | ovf)
            for name, t in zip(("means", "quats", "scales", "opacities", "coeffs"), g):
                params[name].grad = t
            ev["end"].record()
            info_box["grad_bytes"] = exchange.last_bytes
            info_box["info"] = info
            return render, alpha
        if args.variant == "mtgs":
            dirs = params["means"].detach() - cam_pos
            sh_out = spherical_harmonics(3, dirs, params["coeffs"])
            rgb = torch.clamp(sh_out + 0.5, 0.0, 1.0)
            render, alpha, info = rasterization(
                means=params["means"], quats=params["quats"], scales=params["scales"],
                opacities=params["opacities"], colors=rgb, viewmats=viewmat, Ks=K, width=W, height=H,
                tile_size=16, packed=False, near_plane=0.01, far_plane=1e10, render_mode="RGB+ED",
                sparse_grad=False, absgrad=True, rasterize_mode="antialiased")
        else:
            render, alpha, info = rasterization(
                means=params["means"], quats=params["quats"], scales=params["scales"],
                opacities=params["opacities"], colors=params["colors"], viewmats=viewmat, Ks=K,
                width=W, height=H, tile_size=16, packed=False, render_mode="RGB", absgrad=False,
                rasterize_mode="classic")
        info["means2d"].retain_grad()
        torch.autograd.backward([render, alpha], [Gc, Ga])
        if not capturing:
            ev["rows"].record()
        if world > 1:   # dense exchange: every Gaussian gradient tensor (the camera's own viewmat gradient stays local)
            info_box["grad_bytes"] = all_reduce_grads(list(params.values()))
        if not capturing:
            ev["end"].record()
        info_box["info"] = info
        return render, alpha

    def step_fwd():
        if args.variant == "mtgs":
            dirs = params["means"] - cam_pos
            rgb = torch.clamp(spherical_harmonics(3, dirs, params["coeffs"]) + 0.5, 0.0, 1.0)
            return rasterization(
                means=params["means"], quats=params["quats"], scales=params["scales"],
                opacities=params["opacities"], colors=rgb, viewmats=viewmat, Ks=K, width=W, height=H,
                tile_size=16, packed=False, near_plane=0.01, far_plane=1e10, render_mode="RGB+ED",
                sparse_grad=False, absgrad=True, rasterize_mode="antialiased")
        return rasterization(
            means=params["means"], quats=params["quats"], scales=params["scales"],
            opacities=params["opacities"], colors=params["colors"], viewmats=viewmat, Ks=K,
            width=W, height=H, tile_size=16, packed=False, render_mode="RGB", absgrad=False,
            rasterize_mode="classic")

    info_box["exchange"] = exchange
    return step, step_fwd, all_params, info_box


def mtgs_like_iteration_cells():
    """ms per WHOLE MTGS-style training iteration (scripts/mtgs_like_train.py: multi-traversal background + road node, 2M
    Gaussians, 960x540, the shipped option set, loss head, densification statistics, optimizer step) captured as ONE HIP graph:
    visibility-first colours with the fused Adam stepping every row, with the exact row-lazy optimizer, and with the geometry
    gradients kept as rows on top of it.  Run as child
    processes (own scene, own allocator); not part of the headline figure; None when a run fails."""
    import re
    import subprocess
    root = os.path.dirname(os.path.abspath(__file__))
    res = {}
    budget_s = float(os.environ.get("MTGS_BENCH_EXTRA_BUDGET_S", "150"))    # (a fresh box spends a minute or two importing torch: the
    #   extras must not push the run past "a few minutes"; a cell that does not fit the budget is reported as None)
    # the loop that TRAINS, through HIP graphs (train_loop(graph=True)): 1000 steps from a perturbed subset of the true Gaussians
    # with the reference's refinement rules, refinements at 300 ... 900; wall clock per step (from the step at which every
    # traversal has its first graph) with the seven re-captures and refinements inside, the GPU time per step of a stretch
    # without either, and the loss it reached
    left = budget_s - (time.time() - _T0)
    res["mtgs_like_training_ms_per_step"] = res["mtgs_like_training_steady_ms"] = res["mtgs_like_training_loss"] = None
    if left >= 30:
        try:
            r = subprocess.run([sys.executable, os.path.join(root, "scripts", "mtgs_like_train.py"), "--shipped", "--visfirst", "--optimizer",
                                "fused", "--row-lazy", "--geometry-rows", "--only", "fused", "--reps", "1", "--converge", "--grad-thresh", "1e-3",
                                "--clear-radius", "12", "--steps", "1000", "--refine-every", "100", "--densify-from", "250", "--steady", "60",
                                "260", "--train-graph", "--one-graph"], capture_output=True, text=True, timeout=min(180.0, left), cwd=root)
            m = re.search(r"timing: ([\d.]+) ms per step", r.stdout)
            sm = re.search(r"steady: ([\d.]+) ms per step", r.stdout)
            cm = re.search(r"converge: loss ([\d.]+) -> ([\d.]+) .* through (\d+) refinements", r.stdout)
            if r.returncode == 0 and m:
                res["mtgs_like_training_ms_per_step"] = float(m.group(1))
                res["mtgs_like_training_steady_ms"] = float(sm.group(1)) if sm else None
                res["mtgs_like_training_loss"] = [float(cm.group(1)), float(cm.group(2)), int(cm.group(3))] if cm else None
        except Exception:       # noqa: BLE001
            pass
    for key, extra in (("mtgs_like_iteration_graph_ms", ["--visfirst", "--optimizer", "fused"]),
                       ("mtgs_like_iteration_graph_rowlazy_ms", ["--visfirst", "--optimizer", "fused", "--row-lazy"]),
                       ("mtgs_like_iteration_graph_rowlazy_georows_ms", ["--visfirst", "--optimizer", "fused", "--row-lazy", "--geometry-rows"])):
        left = budget_s - (time.time() - _T0)
        if left < 25:
            res[key] = None
            continue
        try:
            r = subprocess.run([sys.executable, os.path.join(root, "scripts", "mtgs_like_train.py"), "--shipped", "--graph", "--reps", "24"] + extra,
                               capture_output=True, text=True, timeout=min(180.0, left), cwd=root)
            m = re.search(r"one graph launch ([\d.]+) ms wall", r.stdout)
            res[key] = float(m.group(1)) if (r.returncode == 0 and m) else None
        except Exception:       # noqa: BLE001
            res[key] = None
    return res


def c1_c2_cells(args, device):
    """BASELINE configs[0] and configs[1] on the driver's box (untimed extras; parity for both is in tests/test_gpu_fullsize.py):
    C1 = 100k Gaussians, 640x480, colours given (SH deg 0), RGB / classic, FORWARD only -- the HIP path beside the CPU oracle's
    forward on the same inputs (the C1 CPU-vs-HIP forward timing); C2 = 500k Gaussians, 1920x1080, SH deg 3,
    forward + backward with the MTGS option set.  Median of 20 (C1) / 10 (C2) after 3 warm-ups, HIP events."""
    import copy
    import numpy as np
    res = {}

    def timed(fn, reps):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    a1 = copy.copy(args)
    a1.n_gaussians, a1.width, a1.height, a1.variant = 100_000, 640, 480, "lean"
    host1, dev1 = build_inputs(a1, 0, device)
    _, fwd1, _, _ = make_step(a1, dev1, 1)
    with torch.no_grad():
        res["c1_100k_640x480_fwd_ms"] = round(timed(fwd1, 20), 4)
    try:
        from oracle import oracle as orc
        orc.build()
        orc.select_native()
        h = {k: v.numpy() for k, v in host1.items()}
        ts = []
        for k in range(8):      # (the first call builds the thread pool and touches the pages: not timed)
            t0 = time.perf_counter()
            orc.rasterization(h["means"], h["quats"], h["scales"], h["opacities"], h["colors"], h["viewmat"], h["K"], 640, 480)
            if k:
                ts.append(time.perf_counter() - t0)
        res["cpu_c1_fwd_ms"] = round(sorted(ts)[len(ts) // 2] * 1e3, 2)
        res["cpu_c1_cores"] = orc.num_threads()
    except Exception:       # noqa: BLE001
        res["cpu_c1_fwd_ms"] = None
    a2 = copy.copy(args)
    a2.n_gaussians, a2.width, a2.height, a2.variant = 500_000, 1920, 1080, "mtgs"
    _, dev2 = build_inputs(a2, 0, device)
    step2, _, _, _ = make_step(a2, dev2, 1)
    res["c2_500k_sh3_1080p_ms"] = round(timed(step2, 10), 4)
    return res


def sh_degree_cell(args, dev):
    """ms per forward + backward of gsplat's OWN call style at the headline size -- rasterization(colors=coefficients, sh_degree=3):
    SH evaluated for the visible Gaussians inside the call, view directions differentiable -- same Gaussians, camera and
    cotangents as the headline step (which is MTGS's composition spherical_harmonics() + clamp + rasterization()).  Not part
    of the headline figure."""
    from mtgs_amd import rasterization
    P = {k: dev[k].detach().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "coeffs")}
    vm = dev["viewmat"].detach().clone().requires_grad_(True)
    K, Gc, Ga = dev["K"], dev["Gc"], dev["Ga"]

    def fb():
        for q in list(P.values()) + [vm]:
            q.grad = None
        r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["coeffs"], vm, K, args.width, args.height,
                                   sh_degree=3, packed=False, render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
        torch.autograd.backward([r, a], [Gc, Ga])
    for _ in range(3):
        fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fb()
    torch.cuda.synchronize()
    return {"sh_degree_call_style_ms": round((time.perf_counter() - t0) / 10 * 1e3, 3)}


def shipped_cells(args, dev):
    """ms per rasterization() forward + backward with the option set of the shipped config/MTGS.py (6 colour channels +
    expected depth, antialiased, absgrad, viewmat gradient; colours given), same Gaussians: 1920x1080 and 960x540, and the
    forward alone at 960x540 (what eval / the viewer run).  Not part of the headline figure."""
    from mtgs_amd import rasterization
    from mtgs_amd.synthetic import make_camera
    N = args.n_gaussians
    g = torch.Generator().manual_seed(3)
    cols = torch.cat([torch.rand(N, 3, generator=g), torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)], -1)
    P = {k: dev[k].detach().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")}
    P["colors"] = cols.to(dev["means"].device).requires_grad_(True)
    res = {}
    for W, H, key in ((args.width, args.height, f"shipped_7ch_{args.width}x{args.height}_ms"), (960, 540, "shipped_7ch_960x540_ms")):
        vm, K = make_camera(W, H)
        vm, K = vm.to(P["means"].device).requires_grad_(True), K.to(P["means"].device)
        Gc = torch.randn(1, H, W, 7, generator=g).to(vm.device)
        Ga = torch.randn(1, H, W, 1, generator=g).to(vm.device)

        def fb(backward=True):
            for q in list(P.values()) + [vm]:
                q.grad = None
            r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, K, W, H, packed=False,
                                       render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
            if backward:
                info["means2d"].retain_grad()
                torch.autograd.backward([r, a], [Gc, Ga])

        def timed(fn, reps=8):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / reps * 1e3, 3)

        res[key] = timed(fb)
        if (W, H) == (args.width, args.height) and "coeffs" in dev:
            # ... and config/MTGS.py's own composition of those seven channels: rgbs = clamp(spherical_harmonics(...) + 0.5, 0, 1),
            # torch.cat([rgbs, normals], dim=-1) (mtgs_scene_graph.py:636-638) -- the colours stay deferred through the clamp AND the
            # concatenation and are evaluated for the visible Gaussians by the rasterization; beside it, SH + clamp over all N
            import mtgs_amd
            from mtgs_amd import spherical_harmonics
            coeffs = dev["coeffs"].detach().clone().requires_grad_(True)
            cam = torch.inverse(vm.detach())[0, :3, 3]
            normals = P["colors"].detach()[:, 3:].clone().requires_grad_(True)

            def fb_sh():
                for q in list(P.values()) + [vm, coeffs, normals]:
                    q.grad = None
                rgbs = torch.clamp(spherical_harmonics(3, P["means"].detach() - cam, coeffs) + 0.5, 0.0, 1.0)
                r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], torch.cat([rgbs, normals], dim=-1), vm, K, W, H,
                                           packed=False, render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
                info["means2d"].retain_grad()
                torch.autograd.backward([r, a], [Gc, Ga])

            res[f"mtgs_py_sh_plus_normals_{W}x{H}_ms"] = timed(fb_sh)
            with mtgs_amd.sh_lazy(True, raster=False):
                res[f"mtgs_py_sh_plus_normals_{W}x{H}_sh_over_all_gaussians_ms"] = timed(fb_sh)
        if (W, H) == (960, 540):
            with torch.no_grad():
                res["fwd_only_7ch_960x540_ms"] = timed(lambda: fb(False))
    return res


def cpu_baseline(args, host, steps):
    """oracle/gsplat_oracle.c (CPU restatement of the same path) on this box's host cores."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    build_label = orc.select_native()      # timing leg: -O3 -march=native build of the same source (not the checker build)
    a = {k: v.numpy() for k, v in host.items()}
    W, H = args.width, args.height
    mtgs = args.variant == "mtgs"
    cam_pos = np.linalg.inv(a["viewmat"][0].astype(np.float64))[:3, 3].astype(np.float32)

    def one():
        if mtgs:
            dirs = a["means"] - cam_pos
            rgb = np.clip(orc.sh_fwd(3, dirs, a["coeffs"]) + 0.5, 0.0, 1.0)
            r, al, m = orc.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], rgb, a["viewmat"],
                                         a["K"], W, H, render_mode="RGB+ED", rasterize_mode="antialiased")
            alc = np.maximum(al, 1e-10)
            Gc_raw = a["Gc"].copy()
            Gc_raw[..., -1:] = a["Gc"][..., -1:] / alc
            Ga_tot = a["Ga"] - (m["render_raw"][..., -1:] / alc ** 2) * a["Gc"][..., -1:] * (al > 1e-10)
        else:
            r, al, m = orc.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"],
                                         a["viewmat"], a["K"], W, H)
            Gc_raw, Ga_tot = a["Gc"], a["Ga"]
        v2d, vabs, vcon, vcol, vop = orc.blend_bwd(m["means2d"], m["conics"], m["colors"], m["opacities"], None, W, H,
                                                   16, m["isect_offsets"], m["flatten_ids"], al, m["last_ids"],
                                                   Gc_raw, Ga_tot, absgrad=mtgs)
        v_depth = vcol[..., -1].copy() if mtgs else np.zeros_like(vop)
        v_comp = vop * a["opacities"][None] if mtgs else None
        orc.project_bwd(a["means"], a["quats"], a["scales"], a["viewmat"], a["K"], W, H, 0.3, m["radii"], m["conics"],
                        m["compensations"], v2d, v_depth, vcon, v_comp)
        if mtgs:
            mask = (rgb + 0.0 > 0.0) & (rgb < 1.0)  # clamp VJP
            orc.sh_bwd(3, dirs, a["coeffs"], vcol[0, :, :3] * mask)

    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    return {"value": round(W * H / t / 1e6, 4), "unit": "Mpix/s", "cores": orc.num_threads(),
            "kind": "port", "build": build_label, "ms_per_step": round(t * 1e3, 1),
            "sample": f"{steps} full step(s) of the same workload ({args.n_gaussians} Gaussians, {W}x{H}, "
                      f"fwd+bwd, variant {args.variant}) by oracle/gsplat_oracle.c with OpenMP; median"}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (this process has not touched
    the GPU: it never calls torch.cuda.is_available() / any HIP function -- torch.cuda.device_count() does not initialise it),
    one per GPU, rendezvous on 127.0.0.1, and exit with the worst of their exit codes.  Rank 0 prints the JSON line on the
    inherited stdout.  Fewer GPUs than ranks is refused unless MTGS_DIST_BACKEND=gloo says the ranks are MEANT to share GPUs
    (a functional run of the N > 1 path on a one-GPU box, not a scaling measurement)."""
    import socket
    import subprocess
    n = args.gpus
    n_dev = torch.cuda.device_count()
    if n_dev < n and os.environ.get("MTGS_DIST_BACKEND") != "gloo":
        print(json.dumps({"bench_error": f"--gpus {n} but {n_dev} GPU(s) visible: refusing to run {n} ranks on fewer GPUs "
                                         "(set MTGS_DIST_BACKEND=gloo for a functional run with ranks sharing a GPU)"}), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    try:
        pending = list(procs)
        while pending:
            for p_ in list(pending):
                rc = p_.poll()
                if rc is None:
                    continue
                pending.remove(p_)
                if rc != 0:
                    worst = max(worst, abs(rc) or 1)
                    for q in pending:       # a rank died: the others would wait for it until the collective timeout
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
    return worst


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(args))          # (before anything touches the GPU)
    if int(env_world or "1") != args.gpus:
        # never a silent N = 1 (or N = something else) number under an `--gpus N` label
        print(json.dumps({"bench_error": f"--gpus {args.gpus} but WORLD_SIZE={env_world or 1}: launch with "
                                         f"`python bench.py --gpus {args.gpus}` (spawns its ranks) or torchrun --nproc-per-node {args.gpus}"}),
              file=sys.stderr)
        sys.exit(2)
    from mtgs_amd import _lib, dist as mdist
    rank, local_rank, world = mdist.init_from_env(timeout_s=300.0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    device = torch.device("cuda", torch.cuda.current_device())
    _lib.load()
    host, dev = build_inputs(args, rank, device)
    step, step_fwd, all_params, info_box = make_step(args, dev, world)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def fail(where, exc):
        """A failed step (a collective that errors out or times out, a kernel error): say WHICH phase, exit non-zero.  No
        in-process restart -- a process that has touched the GPU is never re-exec'ed; retry = a fresh launch, e.g. with
        --dp-exchange dense."""
        ex = info_box.get("exchange")
        phase = ex.phase if ex is not None else ("dense all-reduce" if world > 1 else "single GPU")
        print(json.dumps({"bench_error": f"{type(exc).__name__}: {exc}"[:600], "during": where, "exchange_phase": phase, "rank": rank,
                          "world": world, "dp_exchange": args.dp_exchange if world > 1 else None}), file=sys.stderr, flush=True)
        os._exit(3)

    try:
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        barrier()
        info_box["overflow"] = None       # (the capacities were set inside the first warm-up step)
    except Exception as e:      # noqa: BLE001
        fail("warm-up", e)
    _lib.time_calls(DOMINANT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
    except Exception as e:      # noqa: BLE001
        fail("timed steps", e)
    elapsed = time.perf_counter() - t0
    if info_box.get("overflow") is not None and bool(info_box["overflow"]):
        fail("timed steps", RuntimeError(f"static exchange: a rank had more wire rows than the capacity {info_box['static_cap']}"))
    kernel_ms = [t for name in DOMINANT for t in _lib.timed_ms().get(name, [])]
    _lib.time_calls(())
    # (detached: the info of a step holds means2d and with it the step's autograd graph, whose AccumulateGrad nodes would be reused
    #  by the capture on another stream)
    eager_info = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in info_box["info"].items()}
    info_box["info"] = None
    # ---- the same K steps as ONE HIP GRAPH LAUNCH each (N = 1): the step is captured once under mtgs_amd.graph_mode (fixed capacities,
    # counts on the device, nothing waits for the host) and replayed.  Same kernels on the same inputs; what goes is the host's share
    # -- ~40 kernel launches, ~15 library calls and one mailbox wait per step next to ~1 ms of GPU work, which made the eager figure
    # follow the box's CPU load (1.00 ... 1.22 ms on the same box within minutes).  The eager loop above still runs (its HIP events
    # give the dominant kernel's launch time) and is reported as also.headline_eager_ms.
    elapsed_eager, elapsed_graph, launch, graph_error = elapsed, None, "eager", None
    elapsed_graph_tight, n_listed_tight = None, None
    elapsed_graph_torch_act = elapsed_graph_all_sh = None

    def graph_time(tight, lazy_sh=True, raster=None):
        """K replays of the step captured once under mtgs_amd.graph_mode (+ the opt-in tight tile lists when `tight`; lazy_sh = False:
        spherical_harmonics() evaluated at once, the step's clamp(sh + 0.5) as PyTorch's own kernels -- the round-5 form; raster = False:
        SH + activation in one kernel over ALL Gaussians, not deferred into the rasterization -- the first round-6 form)."""
        import gc
        import mtgs_amd
        with mtgs_amd.sh_lazy(lazy_sh, raster=raster):
            return _graph_time(tight, gc, mtgs_amd)

    def _graph_time(tight, gc, mtgs_amd):
        nv_e, m_e = int((eager_info["radii"] > 0).sum().item()), int(eager_info["flatten_ids"].numel())
        gm = mtgs_amd.graph_mode(int(1.02 * nv_e) + 1024, int(1.02 * m_e) + 8192)      # (a static scene: the counts of the eager steps, a small margin)
        info_box["capture"] = True
        try:
            with mtgs_amd.tight_lists(tight):
                for p_ in all_params:
                    p_.grad = None
                gc.collect()
                with gm:
                    step()                  # (under the mode once: its staging buffers exist before the capture)
                info_box["info"] = None     # (its means2d holds that step's autograd graph: see above)
                gc.collect()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with gm, torch.cuda.graph(g):
                    step()
            overflow, n_l = info_box["info"]["overflow"], info_box["info"]["n_listed"]
            info_box["info"] = None
            for _ in range(args.warmup):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                g.replay()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if bool(overflow):
                raise RuntimeError("the captured frame exceeded its capacities")
            return t1 - t0, int(n_l)
        finally:
            info_box["capture"] = False

    if world == 1 and args.launch == "graph":
        try:
            elapsed_graph, _ = graph_time(False)
            if elapsed_graph < elapsed:      # the line carries the faster of the two launch modes and names it; the other is beside it
                elapsed, launch = elapsed_graph, "graph"
        except Exception as e:      # noqa: BLE001  (the eager figure stands)
            graph_error = f"{type(e).__name__}: {e}"[:300]
            print(f"[bench] graph launch failed, reporting the eager loop: {graph_error}", file=sys.stderr)
        try:        # beside the headline, never the headline: the same step on the opt-in tight tile lists
            if not args.no_tight:
                elapsed_graph_tight, n_listed_tight = graph_time(True)
        except Exception as e:      # noqa: BLE001
            print(f"[bench] tight-lists graph failed: {type(e).__name__}: {e}"[:300], file=sys.stderr)
        try:        # ... and with the colour activation left to PyTorch (what round 5's line measured)
            if not args.no_tight and args.variant == "mtgs":
                elapsed_graph_torch_act, _ = graph_time(False, lazy_sh=False)
                elapsed_graph_all_sh, _ = graph_time(False, raster=False)
        except Exception as e:      # noqa: BLE001
            print(f"[bench] torch-activation graph failed: {type(e).__name__}: {e}"[:300], file=sys.stderr)
    info_box["info"] = eager_info
    # per-rank phase breakdown of the last timed step (N > 1), read before anything else touches the events
    rank_phases = None
    if world > 1:
        evs = info_box["events"]
        torch.cuda.synchronize()
        rank_phases = {"render": evs["start"].elapsed_time(evs["rows"]), "exchange": evs["rows"].elapsed_time(evs["end"])}
        if info_box["exchange"] is not None:
            rank_phases.update(info_box["exchange"].phases_ms())
    # second, UNTIMED pass: every entry point of the step bracketed by HIP events on the launch stream (the events
    # serialise nothing, but they are kept out of the headline figure)
    _lib.time_calls(tuple(ENTRY_POINTS))
    for _ in range(min(args.steps, 8)):
        step()
    torch.cuda.synchronize()
    entry_ms = {n: v for n, v in _lib.timed_ms().items() if v}
    _lib.time_calls(())
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # outside the timed region: the forward-only rate SURVEY.md section 8(d) asks to be reported beside the headline
    # (what eval / the viewer run: the same calls under no_grad)
    fwd_ms = None
    if world == 1 and not args.no_also:
        params_were = [p.requires_grad for p in all_params]
        with torch.no_grad():
            for _ in range(2):
                step_fwd()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(max(args.steps // 2, 1)):
                step_fwd()
            torch.cuda.synchronize()
            fwd_ms = (time.perf_counter() - t1) / max(args.steps // 2, 1) * 1e3
        assert params_were == [p.requires_grad for p in all_params]

    info = info_box["info"]
    n_vis = int((info["radii"] > 0).sum().item())
    # (Gaussian, tile row) items of the binning (csrc/bin3.hip): the tile rows each visible Gaussian's 3-sigma square spans
    _r, _y = info["radii"].reshape(-1).float(), info["means2d"].detach().reshape(-1, 2)[:, 1]
    _th = -(-args.height // 16)
    _rows = (torch.ceil((_y + _r) / 16).clamp(0, _th) - torch.floor((_y - _r) / 16).clamp(0, _th))[_r > 0]
    n_items = int(_rows.sum().item())
    M = int(info["flatten_ids"].numel())       # gsplat's intersection count (3-sigma squares): SURVEY.md section 8(d)'s unit
    # the (tile, Gaussian) pairs the tile lists of the TIMED steps hold: the headline is the default call = gsplat's lists, so this
    # is M -- every byte figure of the line (kernels, entry points, whole step) is priced on the one unit SURVEY.md section 8(d) names
    M_l = int(info["n_listed"]) if info.get("n_listed") is not None else M
    tight_headline = M_l != M      # (only when the run was started with MTGS_TIGHT_LISTS=1)
    P = args.width * args.height
    ms_per_step = elapsed / args.steps * 1e3
    value = world * P / (elapsed / args.steps) / 1e6

    # algorithmic HBM bytes of one compositing-backward launch
    D = 4 if args.variant == "mtgs" else 3
    A = 1 if args.variant == "mtgs" else 0
    bytes_bwd = P * (4 * D + 12) + M_l * (4 + 24 + 4 * D) + n_vis * (24 + 4 * D + 8 * A)
    # Round 5: the zeros of dL/dcoeffs [N, 16, 3] that the spherical_harmonics() backward behind the rasterizer used to write are
    # written by a COMPOSITING kernel beside its own work (mtgs_blend_{fwd,bwd}_packed(also_zero), mtgs_amd/wrapper.py::_Prefill:
    # the forward's by default, together with the 64-byte gradient rows of the compositing backward) -- bytes the step has to
    # write, moved into kernels that leave HBM idle; n_coeff_rows = the Gaussians whose coefficient gradient is non-zero
    # (mtgs_sh_bwd_rows writes those rows)
    from mtgs_amd import wrapper as _wr
    sparse_dp = world > 1 and args.dp_exchange == "sparse" and args.variant == "mtgs"      # (its colours are detached: no SH autograd node)
    sh_zeros = bool(args.variant == "mtgs" and _wr._prefill.enabled and not sparse_dp)
    zeros_in_fwd = bool(_wr._prefill.enabled and _wr._prefill.in_forward and not sparse_dp)
    n_coeff_rows = 0
    _cg = dev["coeffs"].grad if (args.variant == "mtgs" and "coeffs" in dev) else None
    if _cg is not None:
        n_coeff_rows = int((_cg.abs().amax(dim=(1, 2)) > 0).sum().item())
    sh_zero_bytes = args.n_gaussians * 12 * 16 if sh_zeros else 0
    # Round 6: the DENSE gradients the rasterization returns (v_means 12 + v_quats 16 + v_scales 12 + v_opacities 4 + means2d.grad 8 +
    # absgrad 8 [+ colours] bytes per Gaussian, ~94 % zeros) are cleared by the compositing BACKWARD beside its own work; the projection
    # backward writes the rows with a gradient in place and its streaming pass over all N is gone (mtgs_project_bwd_zeroed)
    dense_zeroed = bool(world == 1 and _wr._prefill.enabled and getattr(_wr, "_zeroed_outputs", False))
    dense_zero_bytes = args.n_gaussians * (44 + 8 + 8 * A + (0 if args.variant == "mtgs" else 12)) if dense_zeroed else 0
    bytes_bwd += dense_zero_bytes
    bytes_fwd_zeros = (sh_zero_bytes + n_vis * 64) if zeros_in_fwd else 0
    if sh_zeros and not zeros_in_fwd:
        bytes_bwd += sh_zero_bytes
    # whole-step algorithmic HBM bytes, SURVEY.md section 8(d): B_F + B_B (K = 16 SH bases when the step includes SH)
    N, T = args.n_gaussians, -(-args.width // 16) * -(-args.height // 16)
    Ksh = 16 if args.variant == "mtgs" else 0
    b_fwd = (N * (12 + 12 * Ksh) + N * 12 if Ksh else 0) + N * 44 + N * 4 + n_vis * 28 + n_vis * 16 + M * 12 + 2 * M * 12 \
        + M * 8 + T * 4 + M * (28 + 4 * D) + P * (4 * D + 8)
    b_bwd = P * (4 * D + 12) + M * (28 + 4 * D) + n_vis * (24 + 4 * D + 8 * A) + n_vis * (40 + 28 + 24 + 4 * D) + n_vis * 40 + 64 \
        + (N * 12 * Ksh if Ksh else 0)
    step_bytes = b_fwd + b_bwd
    k_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
    achieved = bytes_bwd / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # algorithmic HBM bytes per launch of every kernel of the step, from THIS run's (N, n_vis, M)
    Ksh_ = 16
    alg = {
        "sh_fwd_k16_kernel<3>": N * (12 + 12 * Ksh_ + 12),
        "sh_bwd_kernel<3>": N * (24 + 12 * Ksh_),
        "sh_bwd_rows_kernel<3>": N * 12 + n_coeff_rows * (12 + 12 * Ksh_),      # cotangents in; direction in + row out where there is one
        "vis_color_fwd_kernel<3>": n_vis * (4 + 12 + 12 * Ksh_ + 12 + 1),          # id, direction, coefficient row in; colour into the record, clamp bits out
        "vis_color_bwd_kernel<3>": n_vis * (4 + 12 + 12 + 1) + n_coeff_rows * 12 * Ksh_,   # id, direction, cotangent, clamp bits in; row out where there is one
        "front_project_kernel": N * (40 + 4 + 40) + n_vis * 48,                  # (+ the chunk-local compact rows of the visible pairs)
        "front_compact_kernel": N * (4 + 4) + n_vis * (48 + 12 + 64 + 4 + 8),   # radii in, vis_rank out | staged row + colours in, record + id + key out
        # (zeroed outputs: id + row in; parameters in, 60 bytes out where there is a gradient)
        "project_bwd_vis_kernel": (n_vis * (40 + 16 + 4 + 64 + 48)) if not dense_zeroed else (n_vis * (4 + 64) + n_coeff_rows * (40 + 60)),
        "project_bwd_expand_kernel": N * (4 + 44 + 28) + n_vis * (4 + 48 + 28),
        "bin3_rows_count_kernel": n_vis * 64,
        "bin3_rows_place_kernel": n_vis * 64 + n_items * 8,
        "bin3_tiles_count_kernel": n_items * 8,
        "bin3_tiles_place_kernel": n_items * (8 + 4) + M_l * 8,
        "bin3_sort_small_kernel": M_l * (8 + 4 + 4 + 4 + 8),
        "blend_fwd_kernel<4, 2, true>": M_l * (4 + 64) + P * (4 * D + 8) + bytes_fwd_zeros,
        "blend_bwd_kernel<4, 4, true>": bytes_bwd,
    }
    # (1) measured in THIS run: HIP events around every C-ABI entry point of the step (second, untimed pass)
    entry_points = []
    for name, kerns in ENTRY_POINTS.items():
        v = entry_ms.get(name)
        if not v:
            continue
        a_bytes = sum(alg.get(k, 0) for k in kerns)
        us = sum(v) / len(v) * 1e3
        ep = {"entry_point": name, "kernels": kerns, "avg_us_live": round(us, 2), "launches_timed": len(v),
              "algorithmic_bytes": int(a_bytes) if a_bytes else None,
              "frac_of_peak_algorithmic": round(a_bytes / us / 1e3 / HBM_PEAK_GBS, 4) if a_bytes else None}
        if name == "mtgs_blend_fwd_packed" and bytes_fwd_zeros:
            # the compositing's own bytes (SURVEY.md section 8(d)'s unit) apart from the zeros that ride on the kernel for the backward pass
            own = a_bytes - bytes_fwd_zeros
            ep["algorithmic_bytes_split"] = {"compositing": int(own), "riding_zeros": int(bytes_fwd_zeros)}
            ep["frac_of_peak_compositing_only"] = round(own / us / 1e3 / HBM_PEAK_GBS, 4)
        entry_points.append(ep)
    # (2) counter-based traffic and VALU-busy fraction per kernel: COMMITTED rocprofv3 --pmc passes of this command on the
    # headline workload (scripts/pmc_step.sh -> profiles/rNN_pmc_step.json, FETCH_SIZE x2 / WRITE_SIZE x1 as calibrated there).
    # They are builder-held numbers echoed into this line, labelled as such, and dropped when the file was made with
    # another HOT-PATH ABI version of the library than the one running (include/mtgs_rast.h MTGS_RAST_HOT_ABI_VERSION: bumps of the
    # optimizer / loss entry points do not invalidate them) or in the other tile-list mode.
    traffic = valu = None
    kernels, counters_from = [], None
    headline = (args.n_gaussians, args.width, args.height, args.variant) == (2_000_000, 1920, 1080, "mtgs")
    for pmc in sorted((ROOT / "profiles").glob("r*_pmc_step.json"), reverse=True):
        if not headline:
            break
        try:
            doc = json.loads(pmc.read_text())
            if doc.get("hot_abi_version") != _lib.HOT_ABI_VERSION or doc.get("lists", "gsplat") != ("tight" if tight_headline else "gsplat"):
                continue
            rec = doc["kernels"]
            for name, a_bytes in alg.items():
                r = rec.get(name) or next((v for k, v in rec.items() if k.startswith(name + "<")), None)   # (template instances of the list mode)
                if r and r.get("hbm_bytes"):
                    cpi, ceil_ = r.get("cycles_per_valu_inst_per_simd"), valu_ceiling(r.get("active_over_insts_valu"), name)
                    frac_hbm = (r["hbm_bytes"] / r["avg_us"] / 1e3 / HBM_PEAK_GBS) if r.get("avg_us") else None
                    frac_valu = (ceil_ / cpi) if (cpi and ceil_) else None
                    kernels.append({"kernel": name, "algorithmic_bytes": int(a_bytes), "counter_bytes_committed": r["hbm_bytes"],
                                    "avg_us_committed": r.get("avg_us"), "cycles_per_valu_inst_committed": cpi,
                                    "valu_ceiling_cycles_per_inst": ceil_, "frac_of_valu_ceiling": None if frac_valu is None else round(frac_valu, 3),
                                    "frac_of_hbm_peak_counter": None if frac_hbm is None else round(frac_hbm, 3),
                                    "mean_resident_waves_per_simd": r.get("mean_resident_waves_per_simd"),
                                    # what binds, from the counters: the larger of the two fractions (a latency-bound kernel is far below both)
                                    "bound": None if (frac_hbm is None or frac_valu is None) else
                                             ("latency" if max(frac_hbm, frac_valu) < 0.35 else ("valu-issue" if frac_valu > frac_hbm else "hbm"))})
            dom = rec.get("blend_bwd_kernel<4, 4, true>", {})
            traffic = dom.get("hbm_bytes")
            if dom.get("cycles_per_valu_inst_per_simd"):
                ceil_ = valu_ceiling(dom.get("active_over_insts_valu"), "blend_bwd_kernel<4, 4, true>")
                valu = {"insts": dom.get("SQ_INSTS_VALU"), "cycles_per_inst": dom["cycles_per_valu_inst_per_simd"],
                        "ceiling_cycles_per_inst": ceil_, "frac": round(ceil_ / dom["cycles_per_valu_inst_per_simd"], 3),
                        "plain_valu_cycles_per_inst": VALU_COST["plain"], "mean_resident_waves_per_simd": dom.get("mean_resident_waves_per_simd"),
                        "note": "wave64 VALU instructions per launch and shader cycles per instruction per SIMD (GRBM_GUI_ACTIVE / 8 x 1024 / "
                                "SQ_INSTS_VALU) from the committed counter pass named in counters_from; ceiling = the cost-weighted mean over the "
                                "kernel's own instruction mix with the per-class issue costs measured in profiles/r06_valu_ceiling.md (plain 2.7, "
                                "DPP 3.9, permlane swap 7.5, transcendental 8.3 cycles); frac = ceiling / measured"}
            counters_from = pmc.name
            break
        except Exception:
            traffic = valu = None
            kernels = []

    out = {
        "metric": "rendered Mpix/s (fwd+bwd) @ 2M Gaussians 1920x1080",
        "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "ms_per_step_eager": round(elapsed_eager / args.steps * 1e3, 3),
        "ms_per_step_graph": None if elapsed_graph is None else round(elapsed_graph / args.steps * 1e3, 3),
        "ms_per_step_exact_lists": round(ms_per_step, 3) if not tight_headline else None,
        "ms_per_step_tight_lists": None if elapsed_graph_tight is None else round(elapsed_graph_tight / args.steps * 1e3, 3),
        "ms_per_step_torch_activation": None if elapsed_graph_torch_act is None else round(elapsed_graph_torch_act / args.steps * 1e3, 3),
        "ms_per_step_sh_over_all_gaussians": None if elapsed_graph_all_sh is None else round(elapsed_graph_all_sh / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[2]: {args.n_gaussians} Gaussians (WB-v1 seed {args.seed}), "
                        f"{args.width}x{args.height}, 1 camera/GPU, fwd+bwd, variant={args.variant}"
                        + (" (SH deg 3 K=16 -> RGB+ED, antialiased, absgrad, viewmat grad)" if args.variant == "mtgs"
                           else " (colours given, RGB, classic)"),
            "n_gaussians": args.n_gaussians, "width": args.width, "height": args.height,
            "n_visible": n_vis, "n_intersections": M, "n_listed": M_l,
            "lists": ("tight (MTGS_TIGHT_LISTS=1): ordered sublists of gsplat's lists, same pixels and gradients" if tight_headline else
                      "gsplat (the default call): isect_ids / flatten_ids / isect_offsets bit-identical to gsplat 1.4.0 isect_tiles + "
                      "isect_offset_encode" + ("" if n_listed_tight is None else
                                               "; the opt-in tight lists (mtgs_amd.tight_lists(): same pixels and gradients, "
                                               f"{n_listed_tight} listed pairs) are timed beside it as ms_per_step_tight_lists, never as `value`")),
            "colour_activation": _colour_activation_note(),
            "launch": ("one HIP graph launch per step: the step captured once under mtgs_amd.graph_mode + torch.cuda.graph and replayed "
                       "(the K steps were timed in both launch modes, the line carries the faster: ms_per_step_eager / ms_per_step_graph)")
                      if launch == "graph" else
                      "eager: every kernel of the step launched from Python" +
                      (f" (graph launch failed: {graph_error})" if graph_error else
                       (" (timed in both launch modes, the line carries the faster: ms_per_step_eager / ms_per_step_graph)"
                        if elapsed_graph is not None else "")),
            "parallelism": f"view-parallel dp{world}, {args.dp_exchange if world > 1 else 'no'} gradient exchange"
                           + (f" ({(('finish_touched_chunked: ' + str(len(info_box['chunk_caps'])) + ' all-gathers (index chunks, reduction overlapped) of together ') if info_box.get('chunk_caps') else 'finish_touched: one all-gather of the ' if info_box.get('touched') else 'finish_static: one all-gather of ') + str(info_box['static_cap']) + (' rows that carry a gradient + their map' if info_box.get('touched') else ' visible rows') + ' per rank, no host read' if info_box.get('static_cap') else 'finish: chunked all-gathers sized on the host'})"
                              if (world > 1 and info_box.get("exchange") is not None) else "")
                           + f", {info_box['grad_bytes']} bytes received per rank per step",
        },
        "roofline": {"kernel": "blend_bwd_kernel<4,4,packed> (mtgs_blend_bwd_packed)", "bound": "hbm", "binds": "valu-issue",
                     "binds_note": "`bound` names the roofline `achieved` / `peak` are quoted on (the contract's hbm | mfma: MFMA is unused); what "
                                   "BINDS the kernel is VALU issue: see `valu` (fraction of the issue ceiling of its own instruction mix) "
                                   "against `frac` (fraction of the HBM peak)",
                     "valu": valu,
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": bytes_bwd, "avg_launch_ms": round(k_ms, 4),
                     "launches_timed": len(kernel_ms),
                     # (round 5) zeros the backward pass wants -- dL/dcoeffs [N,16,3] of the SH backward, the compositing backward's own
                     # gradient rows -- are written by a compositing kernel beside its work: which one, and how many bytes
                     "zeros_for_the_backward_pass": {"written_by": ("mtgs_blend_fwd_packed" if zeros_in_fwd else "mtgs_blend_bwd_packed") if sh_zeros else None,
                                                     "bytes": bytes_fwd_zeros if zeros_in_fwd else sh_zero_bytes,
                                                     "sh_coefficient_rows_with_gradient": n_coeff_rows,
                                                     # (round 6) the dense gradients the node returns, cleared by the compositing BACKWARD
                                                     "dense_gradient_bytes_in_mtgs_blend_bwd_packed": dense_zero_bytes},
                     # the dominant kernel priced on gsplat's intersection count M (the lists of the default call): the same
                     # unit as whole_step below and as SURVEY.md section 8(d)
                     "algorithmic_bytes_on_gsplat_lists": P * (4 * D + 12) + M * (4 + 24 + 4 * D) + n_vis * (24 + 4 * D + 8 * A),
                     "frac_on_gsplat_lists": round((P * (4 * D + 12) + M * (4 + 24 + 4 * D) + n_vis * (24 + 4 * D + 8 * A))
                                                   / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if k_ms > 0 else 0.0,
                     "note": "avg_launch_ms: HIP events on the launch stream around every mtgs_blend_bwd_packed call of the K EAGER steps run "
                             "in front of the timed graph replays (events cannot be recorded into a graph); algorithmic bytes: P*(4D+12) + n_listed*(28+4D) + n_vis*(24+4D+8A) [+ N*192 when zeros_for_the_backward_pass.written_by names this kernel] [+ zeros_for_the_backward_pass.dense_gradient_bytes_in_mtgs_blend_bwd_packed] with n_listed = the (tile, Gaussian) pairs "
                             "of the timed steps' lists (config.n_listed; = gsplat's count config.n_intersections for the default call); "
                             "kernel is VALU-issue bound, not HBM bound (profiles/r06_valu_ceiling.md); avg_launch_ms is measured in this run; traffic "
                             "and the `valu` block come from the committed rocprofv3 --pmc passes named in counters_from (null when none "
                             "matches this library's hot-path ABI version)",
                     "counters_from": counters_from,
                     "entry_points": entry_points,
                     "entry_points_note": "measured in THIS run: HIP events on the launch stream around every C-ABI entry point of the "
                                          "step, in a second untimed pass; algorithmic bytes from this run's (N, n_vis, M)",
                     "kernels": kernels,
                     "kernels_note": "per kernel: algorithmic bytes of this run; *_committed = builder-held rocprofv3 numbers from "
                                     "counters_from, not measured in this run",
                     "whole_step": {"algorithmic_bytes": step_bytes, "unit": "GB/s",
                                    "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                    "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "formula": "SURVEY.md section 8(d) B_F + B_B"}},
    }
    if world > 1:
        # phase breakdown of the LAST timed step (HIP events; phases overlap by design, so they do not add up to
        # ms_per_step): render = forward + backward up to the wire rows (dense: up to the local gradients), meta = all-gather
        # of the visibility maps on the side stream (hidden behind the compositing), wire = first row all-gather issued ->
        # last complete, reduce = the reduction kernels, exchange = rows ready -> gradients ready.  Reduced over the ranks:
        # a straggler shows up as a max far above the min.
        keys = sorted(rank_phases)
        mine = torch.tensor([rank_phases[k] for k in keys] + [float(n_vis), float(info_box["grad_bytes"])], dtype=torch.float64,
                            device=device)
        allr = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        table = torch.stack(allr).cpu()
        out["dp_phases_ms"] = {k: round(float(table[0, i]), 3) for i, k in enumerate(keys)}       # rank 0 (as in round 2)
        out["dp_phases_ms_max"] = {k: round(float(table[:, i].max()), 3) for i, k in enumerate(keys)}
        out["dp_phases_ms_min"] = {k: round(float(table[:, i].min()), 3) for i, k in enumerate(keys)}
        out["dp_n_visible_per_rank"] = [int(v) for v in table[:, len(keys)]]
        out["dp_bytes_received_per_rank"] = [int(v) for v in table[:, len(keys) + 1]]
        # self-diagnosis of the exchange (first multi-GPU run: which assumption of profiles/r06_dp_budget.md holds?)
        ex_ms = [float(table[r, keys.index("exchange")]) for r in range(world)]
        recv = [float(v) for v in table[:, len(keys) + 1]]
        out["dp_finish"] = args.dp_finish if info_box.get("exchange") is not None else "dense all-reduce"
        out["dp_chunk_caps_rows"] = info_box.get("chunk_caps")
        out["dp_overflow"] = None if info_box.get("overflow") is None else bool(info_box["overflow"])
        # bytes a rank RECEIVES from the other ranks per step / the exchange phase: a LOWER bound of what the wire sustained (the phase
        # holds the pack and the reductions too); per link = / (world - 1) peers on a fully connected xGMI node
        out["dp_exchange_GBs_per_rank"] = [round(recv[r] * (world - 1) / world / max(ex_ms[r], 1e-6) / 1e6, 2) for r in range(world)]
        out["dp_exchange_GBs_per_link"] = [round(v / max(world - 1, 1), 2) for v in out["dp_exchange_GBs_per_rank"]]
        out["dp_rank_phases_ms"] = [{k: round(float(table[r, i]), 3) for i, k in enumerate(keys)} for r in range(world)]
        out["dp_world_size"] = torch.distributed.get_world_size()
        out["dp_backend"] = torch.distributed.get_backend()
        try:
            out["dp_rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:       # noqa: BLE001
            out["dp_rccl_version"] = None
        out["dp_env"] = {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_P2P_LEVEL", "NCCL_MIN_NCHANNELS",
                                                       "HSA_ENABLE_IPC_MODE_LEGACY", "MTGS_DIST_BACKEND")}
    if fwd_ms is not None:
        out["also"] = {"headline_eager_ms": round(elapsed_eager / args.steps * 1e3, 3), "fwd_only_ms": round(fwd_ms, 3), "fwd_only_mpix_s": round(P / fwd_ms / 1e3, 1),
                       "gaussians_per_s_fwd_bwd": round(world * args.n_gaussians / (ms_per_step * 1e-3), 0)}
        if args.variant == "mtgs":
            # what the SHIPPED config/MTGS.py drives (outside the timed region): RGB + camera-space normals + expected depth
            # = 7 blended channels, antialiased, absgrad -- at the headline size and at MTGS's training size 960x540
            out["also"].update(shipped_cells(args, dev))
            out["also"].update(sh_degree_cell(args, dev))
            out["also"].update(mtgs_like_iteration_cells())
            out["also"].update(c1_c2_cells(args, device))      # (behind the child processes: its CPU leg starts the OpenMP pool)
    if rank == 0 and world == 1 and args.cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline(args, host, args.cpu_steps)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
