#!/usr/bin/env python3
"""Development: WHERE and WHEN every tile's wave of the compositing kernels ran (library built with
`scripts/build_variant.py timeline -DMTGS_DEV -DMTGS_TIMELINE`): occupancy over time, the longest tile against the kernel's
span, cycles per staged entry.  python scripts/dev/blend_timeline.py <lib.so> [W H] [out.npz]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mtgs_amd import _lib  # noqa: E402

_lib.use_library(sys.argv[1])
from mtgs_amd import rasterization  # noqa: E402
from mtgs_amd.synthetic import make_camera, make_scene  # noqa: E402

W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
out = sys.argv[4] if len(sys.argv) > 4 else None
N = 2_000_000
dev = torch.device("cuda")
sc = make_scene(N, seed=0)
vm, K = make_camera(W, H)
P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
g = torch.Generator().manual_seed(1)
Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
lib = _lib.load()
lib.mtgs_blend_timeline.argtypes = [C.c_void_p, C.c_int, C.c_int]
REC = np.dtype([("rt0", "<u8"), ("rt1", "<u8"), ("c0", "<u8"), ("c1", "<u8"), ("hw", "<u4"), ("xcc", "<u4"), ("staged", "<u4"), ("active", "<u4")])


def step():
    r, a, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm.to(dev), K.to(dev), W, H, packed=False,
                               render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
    torch.autograd.backward([r, a], [Gc, Ga])
    return info


for _ in range(4):
    info = step()
torch.cuda.synchronize()
tiles = ((W + 15) // 16) * ((H + 15) // 16)
off = info["isect_offsets"].reshape(-1).cpu().numpy().astype(np.int64)
L = np.diff(np.append(off, info["flatten_ids"].numel()))
res = {"list_len": L}
for kind, name in ((0, "fwd"), (1, "bwd")):
    buf = np.zeros(tiles, dtype=REC)
    rc = lib.mtgs_blend_timeline(buf.ctypes.data_as(C.c_void_p), kind, tiles)
    assert rc == 0, rc
    res[name] = buf
    rt0, rt1 = buf["rt0"].astype(np.int64), buf["rt1"].astype(np.int64)
    ok = rt1 > 0
    t_start, t_end = rt0[ok].min(), rt1[ok].max()
    span_us = (t_end - t_start) / 100.0          # s_memrealtime: 100 MHz
    dur_us = (rt1 - rt0)[ok] / 100.0
    cyc = (buf["c1"].astype(np.int64) - buf["c0"].astype(np.int64))[ok]
    clock_ghz = cyc.sum() / (dur_us.sum() * 1e3)
    simd = (buf["xcc"][ok].astype(np.int64) & 0xF) << 32 | (buf["hw"][ok].astype(np.int64) & 0xFFFFFFF0)
    n_simd = np.unique(simd).size
    mean_occ = dur_us.sum() / span_us / max(n_simd, 1)
    print(f"== {name}: {ok.sum()} waves, span {span_us:.1f} us, shader clock {clock_ghz:.2f} GHz, {n_simd} distinct (xcc, hw_id>>4) slots")
    print(f"   wave duration us: mean {dur_us.mean():.1f} p50 {np.percentile(dur_us, 50):.1f} p90 {np.percentile(dur_us, 90):.1f} "
          f"p99 {np.percentile(dur_us, 99):.1f} max {dur_us.max():.1f}   sum/span = {dur_us.sum() / span_us:.0f} waves resident on average "
          f"({dur_us.sum() / span_us / 1024:.2f} per SIMD of 1024)")
    st, ac = buf["staged"][ok].astype(np.float64), buf["active"][ok].astype(np.float64)
    print(f"   staged entries {st.sum():.0f}, with a valid pixel {ac.sum():.0f}; cycles per staged entry: total {cyc.sum() / st.sum():.0f}")
    # occupancy over time, 20 bins
    edges = np.linspace(t_start, t_end, 21)
    occ = [(np.minimum(rt1[ok], edges[i + 1]) - np.maximum(rt0[ok], edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) for i in range(20)]
    print("   resident waves per 5 % of the span: " + " ".join(f"{o:.0f}" for o in occ))
    # the longest-running waves: when they started, how long their lists are
    order = np.argsort(-dur_us)[:8]
    idx = np.nonzero(ok)[0][order]
    print("   longest waves: " + "; ".join(f"blk {i} start +{(rt0[i] - t_start) / 100:.0f}us dur {(rt1[i] - rt0[i]) / 100:.0f}us staged {buf['staged'][i]} "
                                            f"active {buf['active'][i]}" for i in idx))
    last = np.argsort(-rt1[ok])[:8]
    idx = np.nonzero(ok)[0][last]
    print("   last to finish: " + "; ".join(f"blk {i} start +{(rt0[i] - t_start) / 100:.0f}us dur {(rt1[i] - rt0[i]) / 100:.0f}us staged {buf['staged'][i]}"
                                             for i in idx))
    # the rate a wave runs at, against how many waves shared the chip when it ran
    per_entry = cyc / np.maximum(st, 1)
    print(f"   cycles per staged entry per wave: p10 {np.percentile(per_entry, 10):.0f} p50 {np.percentile(per_entry, 50):.0f} p90 {np.percentile(per_entry, 90):.0f}")
print(f"list lengths: mean {L.mean():.0f} p50 {np.percentile(L, 50):.0f} p90 {np.percentile(L, 90):.0f} p99 {np.percentile(L, 99):.0f} max {L.max()}")
if out:
    np.savez_compressed(out, **res)
