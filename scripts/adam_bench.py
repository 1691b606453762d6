"""The optimizer step at the headline size: 2M Gaussians x the parameter groups of config/MTGS.py:121-181 (59 floats per
Gaussian = 472 MB of parameters).  mtgs_amd.optim.FusedAdam (one launch; dense gradients / compact rows of the visible 15 %)
against torch.optim.Adam (foreach, and torch's own fused=True).  Bytes: p, m, v read + written (24 B per element) + 4 B of
dense gradient per element."""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
if "--lib" in sys.argv:      # development: an A/B build of the library (scripts/build_variant.py)
    from mtgs_amd import _lib
    _lib.use_library(sys.argv[sys.argv.index("--lib") + 1])
from mtgs_amd.optim import FusedAdam  # noqa: E402

GROUPS = [("means", (3,), 8e-4), ("features_dc", (3,), 0.0025), ("features_rest", (15, 3), 0.0025 / 20),
          ("opacities", (1,), 0.05), ("scales", (3,), 0.005), ("quats", (4,), 0.001)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--traversals", type=int, default=3)
    ap.add_argument("--lib", default=None, help="development: path of an A/B build of the library")
    ap.add_argument("--clustered", action="store_true", help="rowlazy: visible Gaussians in runs of 4096 consecutive indices")
    ap.add_argument("--only", default="", help="profiling aid: run one variant only (fused-nt | fused | rows | rowlazy | torch-foreach | torch-fused)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    N = args.n
    g = torch.Generator(device="cuda").manual_seed(0)
    elems = sum(int(torch.tensor(s).prod()) for _, s, _ in GROUPS) * N

    def params():
        return [torch.randn(N, *s, device=dev, generator=g).requires_grad_(True) for _, s, _ in GROUPS]

    def grads(P):
        for p in P:
            p.grad = torch.randn(p.shape, device=dev, generator=g) * 0.01

    def timeit(step, reps):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) / reps * 1e3

    out = {}
    variants = {"fused-nt": lambda P: FusedAdam([{"params": [p], "lr": lr} for p, (_, _, lr) in zip(P, GROUPS)], eps=1e-15, nontemporal=True),
                "fused": lambda P: FusedAdam([{"params": [p], "lr": lr} for p, (_, _, lr) in zip(P, GROUPS)], eps=1e-15, nontemporal=False),
                "torch-foreach": lambda P: torch.optim.Adam([{"params": [p], "lr": lr} for p, (_, _, lr) in zip(P, GROUPS)], eps=1e-15, foreach=True),
                "torch-fused": lambda P: torch.optim.Adam([{"params": [p], "lr": lr} for p, (_, _, lr) in zip(P, GROUPS)], eps=1e-15, fused=True)}
    for name, mk in variants.items():
        if args.only and args.only != name:
            continue
        P = params()
        grads(P)
        opt = mk(P)
        gpu, wall = timeit(opt.step, args.reps)
        nbytes = elems * 28
        out[name] = gpu
        print(f"{name:14s} {gpu * 1e3:8.1f} us GPU  {wall * 1e3:8.1f} us wall  {nbytes / gpu / 1e6:7.0f} GB/s of p,m,v,g traffic ({nbytes / 1e6:.0f} MB)")
        del P, opt
    if not args.only or args.only == "rows":
        # gradient source 2: the visible 15 % as compact rows (64 floats per row: 11 geometry + 48 colour + padding)
        P = params()
        vis = torch.rand(N, device=dev, generator=g) < 0.15
        n_vis = int(vis.sum())
        row_of = torch.full((N,), -1, dtype=torch.int32, device=dev)
        row_of[vis] = torch.arange(n_vis, dtype=torch.int32, device=dev)
        rows = torch.randn(n_vis, 64, device=dev, generator=g) * 0.01
        cols = [0, 3, 16, 6, 7, 10]
        opt = FusedAdam([{"params": [p], "lr": lr} for p, (_, _, lr) in zip(P, GROUPS)], eps=1e-15)

        def step():
            for p, c in zip(P, cols):
                opt.set_row_gradient(p, rows, row_of, c)
            opt.step()

        gpu, wall = timeit(step, args.reps)
        nbytes = elems * 24 + n_vis * 256 + N * 4 * len(P)
        print(f"{'fused rows':14s} {gpu * 1e3:8.1f} us GPU  {wall * 1e3:8.1f} us wall  {nbytes / gpu / 1e6:7.0f} GB/s ({nbytes / 1e6:.0f} MB: p,m,v + "
              f"{n_vis} rows + the row map per tensor)")

    if not args.only or args.only == "rowlazy":
        # exact row-lazy Adam: colour tensors [N, 3], [N, T, 3], [N, T, 15, 3]; every step renders one traversal and sees a fresh
        # random 15 % (worst case for the catch-up: a row is T / 0.15 steps behind on average)
        T = args.traversals
        Pc = {"dc": torch.randn(N, 3, device=dev, generator=g), "ad": torch.randn(N, T, 3, device=dev, generator=g),
              "rest": torch.randn(N, T, 15, 3, device=dev, generator=g)}
        for variant in ("rows (every row, every traversal)", "row-lazy", "row-lazy, peek + caught", "row-lazy, id list, peek + caught"):
            P = {k: v.clone().requires_grad_(True) for k, v in Pc.items()}
            opt = FusedAdam([{"params": [P["dc"], P["ad"]], "lr": 0.0025}, {"params": [P["rest"]], "lr": 0.0025 / 20}], eps=1e-15)
            lazy = variant.startswith("row-lazy")
            peek = variant.endswith("caught")
            lst = "id list" in variant
            if lazy:
                opt.set_row_lazy(P["dc"]); opt.set_row_lazy(P["ad"], traversals=T); opt.set_row_lazy(P["rest"], traversals=T)
            frames = []
            for i in range(8):
                if args.clustered:     # visibility in runs of 4096 consecutive Gaussians (spatially sorted scenes)
                    vis = (torch.rand((N + 4095) // 4096, device=dev, generator=g) < 0.15).repeat_interleave(4096)[:N]
                else:
                    vis = torch.rand(N, device=dev, generator=g) < 0.15
                n_vis = int(vis.sum())
                row_of = torch.full((N,), -1, dtype=torch.int32, device=dev)
                row_of[vis] = torch.arange(n_vis, dtype=torch.int32, device=dev)
                frames.append((row_of, torch.randn(n_vis, 48, device=dev, generator=g) * 0.01, n_vis,
                               torch.nonzero(vis).reshape(-1).to(torch.int32)))
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.reps + 8)]
            for i in range(args.reps + 8):
                row_of, rows, n_vis, ids = frames[i % 8]
                t = i % T
                ev[i][0].record()
                ck = lambda col: {}
                rid = ((ids, 0, None),) if lst else ()
                if peek:
                    Cb = torch.empty(n_vis, 52, device=dev)
                    opt.peek_rows([(P["dc"], row_of, None, 0) + rid, (P["ad"], row_of, t, 3) + rid, (P["rest"], row_of, t, 6) + rid], Cb)
                    ck = lambda col: {"caught": (Cb, col), **({"row_ids": (ids, 0, None)} if lst else {})}
                elif lazy:
                    opt.catch_up_rows([(P["dc"], row_of, None), (P["ad"], row_of, t), (P["rest"], row_of, t)])
                ev[i][1].record()
                opt.set_row_gradient(P["dc"], rows, row_of, 0, **ck(0))
                opt.set_row_gradient(P["ad"], rows, row_of, 0, slice_index=t, **ck(3))
                opt.set_row_gradient(P["rest"], rows, row_of, 3, slice_index=t, **ck(6))
                opt.step()
                ev[i][2].record()
            torch.cuda.synchronize()
            cu = sorted(e[0].elapsed_time(e[1]) for e in ev[8:])[len(ev[8:]) // 2]
            stp = sorted(e[1].elapsed_time(e[2]) for e in ev[8:])[len(ev[8:]) // 2]
            touched = n_vis * 51 * 4
            print(f"{variant:36s} T={T}: catch-up {cu * 1e3:7.1f} us  step {stp * 1e3:7.1f} us   (visible rows: {n_vis}, "
                  f"{touched * 6 / 1e6:.0f} MB p,m,v r+w per pass; dense: {N * (3 + 48 * T) * 24 / 1e6:.0f} MB)")
            del P, opt


if __name__ == "__main__":
    main()
