#!/usr/bin/env python3
"""Development: cost of catching up rows after a long gap (exact row-lazy Adam): N rows of 45 floats stepped once, then GAP
steps without a visible row, then one catch-up of all rows."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mtgs_amd.optim import FusedAdam  # noqa: E402

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
for gap in (10, 100, 500, 1000, 2000, 4000):
    P = (torch.randn(N, 15, 3, device=dev) * 0.3).requires_grad_(True)
    opt = FusedAdam([P], lr=2e-3, eps=1e-15)
    opt.set_row_lazy(P)
    row_all = torch.arange(N, dtype=torch.int32, device=dev)
    rows = torch.randn(N, 48, device=dev) * 0.01
    none = torch.full((N,), -1, dtype=torch.int32, device=dev)
    opt.set_row_gradient(P, rows, row_all, 3)
    opt.step()
    for _ in range(gap):
        opt.set_row_gradient(P, rows, none, 3)
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.catch_up_rows([(P, row_all, None)])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"gap {gap:5d}: catch-up of {N} rows {ms:8.2f} ms  = {ms * 1e6 / (N * 45 * gap):.3f} ns per element-step "
          f"(dense streaming: {28 / 5.8e3:.4f} ns)")
