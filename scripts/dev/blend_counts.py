#!/usr/bin/env python3
"""Development: candidate / entry / slot counts of the compositing backward (library built with scripts/build_variant.py count -DMTGS_DEV -DMTGS_COUNT)."""
import ctypes as C
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mtgs_amd import _lib
_lib.use_library(sys.argv[1])
from mtgs_amd import rasterization
from mtgs_amd.synthetic import make_camera, make_scene
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
dev = torch.device("cuda")
sc = make_scene(2_000_000, seed=0)
vm, K = make_camera(W, H)
P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
g = torch.Generator().manual_seed(1)
Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
lib = _lib.load()
buf = (C.c_ulonglong * 8)()
r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm.to(dev), K.to(dev), W, H, packed=False,
                           render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
torch.cuda.synchronize()
lib.mtgs_blend_counters(buf, 1)
torch.autograd.backward([r, a], [Gc, Ga])
torch.cuda.synchronize()
lib.mtgs_blend_counters(buf, 1)
M = info["flatten_ids"].numel()
print(f"M={M} staged entries tested={buf[0]} with a valid pixel={buf[1]} slots={buf[2]} valid (entry, pixel) pairs={buf[3]}  "
      f"slots/entry={buf[2] / max(buf[1], 1):.2f} lanes/slot={buf[3] / max(buf[2], 1):.1f}")
