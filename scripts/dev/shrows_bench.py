#!/usr/bin/env python3
"""Development: mtgs_sh_bwd_rows alone on a 2M x 16 x 3 buffer, cotangents non-zero for 6 % of the rows (HIP events, cold buffer
between repeats).  --lib selects an A/B build."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mtgs_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--density", type=float, default=0.06)
args = ap.parse_args()
if args.lib:
    _lib.use_library(args.lib)
from mtgs_amd._lib import call, ptr, stream_of  # noqa: E402

dev = torch.device("cuda")
n, K = 2_000_000, 16
g = torch.Generator().manual_seed(0)
dirs = torch.randn(n, 3, generator=g).to(dev)
v = (torch.randn(n, 3, generator=g) * (torch.rand(n, 1, generator=g) < args.density)).to(dev)
out = torch.zeros(n, K, 3, device=dev)
junk = torch.empty(128 << 20, device=dev)
ts = []
for _ in range(8):
    out.zero_(); junk.fill_(1.0)     # (the rows' lines leave the caches)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call("mtgs_sh_bwd_rows", n, K, 3, ptr(dirs), None, ptr(v), ptr(out), stream_of(dirs))
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(f"{args.lib or 'in-tree':50s} density {args.density}: median {sorted(ts)[len(ts)//2]:.1f} us  min {min(ts):.1f} us")
