"""The N > 1 path on CPU: world_size 2 over gloo.  Each rank produces the gradients of ITS camera
(with the CPU oracle standing in for the device kernels -- this test is about the exchange, not
the kernels), all-reduces them through mtgs_amd.dist.FlatGradBucket, and the result must equal the
single-process two-camera backward (view-parallel DP == gradient accumulation over cameras)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _camera_grads(orc, a, vm, K, W, H, Gc, Ga):
    r, al, m = orc.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], a["colors"], vm, K, W, H)
    v2d, vabs, vcon, vcol, vop = orc.blend_bwd(m["means2d"], m["conics"], m["colors"], m["opacities"], None, W, H, 16,
                                               m["isect_offsets"], m["flatten_ids"], al, m["last_ids"], Gc, Ga)
    vm_, vq, vs, _ = orc.project_bwd(a["means"], a["quats"], a["scales"], vm, K, W, H, 0.3, m["radii"], m["conics"], None,
                                     v2d, np.zeros_like(vop), vcon, None)
    stats_sum = (m["radii"] > 0).sum(0).astype(np.float32)          # visibility counts
    stats_max = m["radii"].max(0).astype(np.float32)                # max screen radius
    return dict(means=vm_, quats=vq, scales=vs, opacities=vop.sum(0), colors=vcol.sum(0)), stats_sum, stats_max


def _scene():
    sys.path.insert(0, str(ROOT))
    from tests.util import small_scene, to_np
    W, H = 64, 48
    sc, vm, K = small_scene(N=150, W=W, H=H, seed=31)
    vm2 = torch.cat([vm, vm.clone()])
    vm2[1, 0, 3] += 0.6
    g = torch.Generator().manual_seed(5)
    Gc = torch.randn(2, H, W, 3, generator=g).numpy()
    Ga = torch.randn(2, H, W, 1, generator=g).numpy()
    return sc, to_np(sc), vm2.numpy(), torch.cat([K, K]).numpy(), W, H, Gc, Ga


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, str(ROOT))
    from mtgs_amd import dist as mdist
    from oracle import oracle as orc
    r, lr, w = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    sc, a, vm2, K2, W, H, Gc, Ga = _scene()
    cam = mdist.camera_for_rank(0, rank, world, 2)
    grads, s_sum, s_max = _camera_grads(orc, a, vm2[cam:cam + 1], K2[cam:cam + 1], W, H, Gc[cam:cam + 1], Ga[cam:cam + 1])
    names = ["means", "quats", "scales", "opacities", "colors"]
    params = [sc[n].clone().requires_grad_(True) for n in names]
    bucket = mdist.FlatGradBucket(params)
    bucket.zero()
    for p, n in zip(params, names):
        p.grad.add_(torch.from_numpy(grads[n]))        # what autograd's accumulation does
    bucket.all_reduce()
    # the bucket-free variant must give the same sums
    params2 = [sc[n].clone().requires_grad_(True) for n in names]
    for p, n in zip(params2, names):
        p.grad = torch.from_numpy(grads[n]).clone()
    sent = mdist.all_reduce_grads(params2, direct_bytes=1500)   # colours/means go direct, the rest packed
    assert sent == sum(p.grad.numel() * 4 for p in params2)
    for p, q in zip(params, params2):
        assert torch.equal(p.grad, q.grad)
    local_sum, local_max = torch.from_numpy(s_sum.copy()), torch.from_numpy(s_max.copy())
    t_sum, t_max = torch.from_numpy(s_sum), torch.from_numpy(s_max)
    mdist.all_reduce_stats([t_sum], [t_max])
    # an accumulator that starts at ONE (the reference's vis_counts, vanilla_gaussian_splatting.py:462): the initial value
    # is counted once, not once per rank; and the reducer's global view can be read twice without double counting
    t_one = local_sum + 1.0
    red = mdist.StatsReducer([t_one], [local_max], sum_init=[1.0])
    g1, _ = red.reduce()
    g2, m2 = red.reduce()
    assert torch.equal(g1[0], g2[0]), "second read differs"
    assert torch.allclose(g1[0].double(), t_sum.double() + 1.0), (g1[0][:8], t_sum[:8])
    assert torch.allclose(m2[0].double(), t_max.double())
    np.savez(Path(out_dir) / f"rank{rank}.npz", vis=t_sum.numpy(), maxr=t_max.numpy(),
             **{n: p.grad.numpy() for p, n in zip(params, names)})
    dist.barrier()
    dist.destroy_process_group()


def test_dp_allreduce_equals_two_camera_backward(tmp_path, oracle):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sc, a, vm2, K2, W, H, Gc, Ga = _scene()
    ref, s_sum, s_max = _camera_grads(oracle, a, vm2, K2, W, H, Gc, Ga)   # one process, both cameras
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for n in ["means", "quats", "scales", "opacities", "colors"]:
        assert np.array_equal(r0[n], r1[n]), f"{n}: ranks disagree after all-reduce"
        scale = np.abs(ref[n]).max()
        assert np.abs(r0[n] - ref[n]).max() <= 2e-6 * scale + 1e-7, n
    assert np.array_equal(r0["vis"], s_sum) and np.array_equal(r0["maxr"], s_max)
    assert np.array_equal(r0["vis"], r1["vis"]) and np.array_equal(r0["maxr"], r1["maxr"])


def test_camera_sharding_covers_all_cameras():
    from mtgs_amd.dist import camera_for_rank
    world, n_cam = 8, 8
    for step in range(3):
        assert sorted(camera_for_rank(step, r, world, n_cam) for r in range(world)) == list(range(8))
    assert [camera_for_rank(s, 0, 2, 5) for s in range(5)] == [0, 2, 4, 1, 3]


# ---- the default exchange (SparseGradExchange): its HOST-SIDE bookkeeping over gloo, CPU tensors, no kernels ---------------------
def _sparse_inputs(rank, N, T):
    """A rank's frame as the front / projection-backward kernels would leave it: visibility mask, meta record
    [count, cam xyz | words u64 | prefix u32 | traversal, spare] and the wire rows of its visible Gaussians in index order."""
    g = np.random.default_rng(100 + rank)
    vis = g.random(N) < (0.15 + 0.1 * rank)
    vis[N // 2: N // 2 + 300] = rank == 0            # a stretch only rank 0 sees (an index chunk where the other ranks' share is short)
    idx = np.nonzero(vis)[0]
    rows = g.standard_normal((idx.size, 16)).astype(np.float32)
    rows[:, 15] = idx.astype(np.int32).view(np.float32)
    nw = (N + 63) // 64
    bits = np.zeros(nw * 64, dtype=np.uint8)
    bits[:N] = vis
    words = np.packbits(bits.reshape(nw, 64), axis=1, bitorder="little").view(np.uint64).reshape(nw)
    prefix = np.concatenate([[0], np.cumsum(bits.reshape(nw, 64).sum(1))[:-1]]).astype(np.uint32)
    cam = np.array([1.0 + rank, -2.0, 0.5 * rank], dtype=np.float32)
    trav = (rank + 1) % T
    return vis, idx, rows, words, prefix, cam, trav


def _sparse_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, str(ROOT))
    from mtgs_amd import dist as mdist
    mdist.init_from_env(backend="gloo")
    N, T, K = 5000, 2, 16
    ex = mdist.SparseGradExchange(N, K, "cpu", chunks=3, traversals=T)
    assert ex.bounds == [0, 2048, 4096, 5000] and ex.n_chunks == 3 and (ex.world, ex.rank) == (world, rank)
    vis, idx, rows, words, prefix, cam, trav = _sparse_inputs(rank, N, T)
    nw = ex.n_words
    # what mtgs_front_fwd / SparseGradExchange.rasterization() write into the meta record, and mtgs_project_bwd_rows into the rows
    meta = np.zeros(ex.meta_len, dtype=np.int32)
    meta[0] = idx.size
    meta[1:4] = cam.view(np.int32)
    meta[4:4 + 2 * nw] = words.view(np.int32)
    meta[4 + 2 * nw:4 + 3 * nw] = prefix.view(np.int32)
    meta[ex.meta_len - 2] = trav
    ex.meta.copy_(torch.from_numpy(meta))
    ex.rows[:idx.size] = torch.from_numpy(rows)
    ex.rows[idx.size:] = float("nan")                 # slack behind the rows: may travel, must never be read
    ex._pending = {"stage": "forward"}
    ex.after_front()                                   # the meta all-gather + the samples every rank's plan is a function of
    pl = ex.plan(ex._samples_host.numpy())
    works, recvs = ex.gather_rows(pl)
    for w in works:
        w.wait()
    # ---- the plan against an independent count from every rank's mask (regenerated from its seed)
    others = [_sparse_inputs(r, N, T) for r in range(world)]
    for r, (vis_r, idx_r, _, _, _, _, trav_r) in enumerate(others):
        assert pl["counts"][r] == idx_r.size and pl["trav"][r] == trav_r
        assert pl["starts"][r] == [int(vis_r[:b].sum()) for b in ex.bounds[:-1]] + [idx_r.size]
    for c in range(ex.n_chunks):
        assert pl["caps"][c] == max(max(int(o[0][ex.bounds[c]:ex.bounds[c + 1]].sum()) for o in others), 1)
    assert pl["masks"] == [sum(1 << r for r in range(world) if others[r][6] == t) for t in range(T)]
    assert pl["present"] == [t for t in range(T) if pl["masks"][t]] and pl["subsets"][0] == (1 << world) - 1
    assert pl["subsets"][1:] == [pl["masks"][t] for t in pl["present"]]
    assert pl["union_caps"][0] == min(N, sum(o[1].size for o in others))
    assert ex.last_bytes == sum(world * cap * 64 for cap in pl["caps"]) + world * ex.meta_len * 4
    # ---- every sender's row of every Gaussian is where the receiver's kernel looks for it: block c, row prefix(n) - starts[r][c]
    metas = ex._pending["metas"].numpy()
    dense = np.zeros((N, 15), dtype=np.float64)        # what the reduction sums: all ranks, index by index
    per_trav = {t: np.zeros((N, 3), dtype=np.float64) for t in range(T)}       # the colour part goes to the SENDER's traversal
    for r in range(world):
        w64 = metas[r, 4:4 + 2 * nw].copy().view(np.uint64)
        pre = metas[r, 4 + 2 * nw:4 + 3 * nw].copy().view(np.uint32)
        assert np.array_equal(metas[r, 1:4].view(np.float32), others[r][5])
        for c in range(ex.n_chunks):
            b0, b1 = ex.bounds[c], ex.bounds[c + 1]
            for n in np.nonzero(others[r][0][b0:b1])[0] + b0:
                word, bit = int(w64[n >> 6]), int(n & 63)
                assert (word >> bit) & 1
                row_abs = int(pre[n >> 6]) + bin(word & ((1 << bit) - 1)).count("1")
                k = row_abs - pl["starts"][r][c]
                assert 0 <= k < pl["caps"][c]
                row = recvs[c][r, k].numpy()
                assert int(row[15:16].view(np.int32)[0]) == n and not np.isnan(row[:15]).any()
                dense[n] += row[:15]
                per_trav[others[r][6]][n] += row[11:14]
    ref = np.zeros((N, 15), dtype=np.float64)
    for (_, idx_r, rows_r, *_rest) in others:
        ref[idx_r] += rows_r[:, :15]
    assert np.array_equal(dense, ref)                   # ranks == accumulation over the cameras, through the wire layout
    for t in range(T):
        sel = np.zeros((N, 3))
        for (_, idx_r, rows_r, _, _, _, trav_r) in others:
            if trav_r == t:
                sel[idx_r] += rows_r[:, 11:14]
        assert np.array_equal(per_trav[t], sel)
    np.save(Path(out_dir) / f"sparse{rank}.npy", dense)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sparse_exchange_bookkeeping_over_gloo(tmp_path, world):
    """SparseGradExchange's host logic with `world` ranks over gloo on CPU tensors: meta all-gather -> samples -> plan (counts, chunk
    starts, equal per-chunk capacities, traversal masks / subsets, union capacities) -> chunked row all-gathers.  Every rank
    finds every sender's row of every visible Gaussian at block c, row prefix(n) - starts[r][c] (the receiver kernels' address
    rule), the sums equal the accumulation over all cameras, the colour part is routed to the sender's traversal, and the
    ranks agree bit for bit.  (The kernels that consume this layout are covered on the GPU: tests/test_gpu_dp.py.)"""
    port = _free_port()
    mp.spawn(_sparse_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"sparse{r}.npy") for r in range(world)]
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)


# ---- the chunked, overlapped touched exchange (SparseGradExchange.finish_touched_chunked): its wire layout over gloo, CPU tensors ------
def _touched_inputs(rank, N, T):
    """_sparse_inputs with ~60 % of the visible rows exactly zero (hidden behind others: no gradient), the touched rows' map
    (mtgs_dp_touched_pack_chunks' output format = a visibility map of the rows that carry a gradient)."""
    vis, idx, rows, _, _, cam, trav = _sparse_inputs(rank, N, T)
    g = np.random.default_rng(500 + rank)
    zero = g.random(idx.size) < 0.6
    zero[idx >= 4096] = rank == 1                    # the last index chunk: rank 1 has nothing there, the others everything
    rows[zero, :15] = 0.0
    t_idx = idx[~zero]
    nw = (N + 63) // 64
    bits = np.zeros(nw * 64, dtype=np.uint8)
    bits[t_idx] = 1
    words = np.packbits(bits.reshape(nw, 64), axis=1, bitorder="little").view(np.uint64).reshape(nw)
    prefix = np.concatenate([[0], np.cumsum(bits.reshape(nw, 64).sum(1))[:-1]]).astype(np.uint32)
    return t_idx, rows[~zero], words, prefix, cam, trav


def _chunked_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, str(ROOT))
    from mtgs_amd import dist as mdist
    mdist.init_from_env(backend="gloo")
    N, T, K = 5000, 2, 16
    ex = mdist.SparseGradExchange(N, K, "cpu", chunks=3, traversals=T)
    nw, nch, bounds = ex.n_words, ex.n_chunks, ex.bounds
    t_idx, t_rows, words, prefix, cam, trav = _touched_inputs(rank, N, T)
    # warm-up agreement: per-chunk counts -> MAX over the ranks + a margin (what bench.py does with touched_chunk_counts())
    counts = np.array([int(((t_idx >= bounds[c]) & (t_idx < bounds[c + 1])).sum()) for c in range(nch)], dtype=np.float64)
    starts = [int(prefix[bounds[c] // 64]) for c in range(nch)] + [t_idx.size]
    assert [starts[c + 1] - starts[c] for c in range(nch)] == counts.astype(int).tolist()      # the device formula of touched_chunk_counts()
    agreed = torch.from_numpy(counts.copy())
    dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
    caps = [int(c) + 4 for c in agreed.tolist()]
    lay = ex.chunk_layout(caps)
    assert lay["pad"] % 16 == 0 and lay["pad"] >= ex.meta_len and lay["floats"][0] == lay["pad"] + caps[0] * 16
    assert lay["floats"][1:] == [c * 16 for c in caps[1:]] and lay["row0"] == [lay["pad"], 0, 0]
    # the sender's messages as mtgs_dp_touched_pack_chunks leaves them: meta record in front of chunk 0's rows, NaN slack behind the rows
    sends = [np.full(n, np.nan, dtype=np.float32) for n in lay["floats"]]
    meta = sends[0][:lay["pad"]].view(np.int32)
    meta[:] = 0
    meta[0] = t_idx.size
    meta[1:4] = cam.view(np.int32)
    meta[4:4 + 2 * nw] = words.view(np.int32)
    meta[4 + 2 * nw:4 + 3 * nw] = prefix.view(np.int32)
    meta[ex.meta_len - 2] = trav
    for c in range(nch):
        blk = sends[c][lay["row0"][c]:].reshape(caps[c], 16)
        blk[:starts[c + 1] - starts[c]] = t_rows[starts[c]:starts[c + 1]]
    works, recvs = ex.gather_chunk_messages([torch.from_numpy(s) for s in sends])
    assert ex.last_bytes == world * 4 * sum(lay["floats"])
    # a receiver reduces chunk c as soon as ITS message is there: wait in order, read only that chunk
    others = [_touched_inputs(r, N, T) for r in range(world)]
    dense = np.zeros((N, 15), dtype=np.float64)
    metas = None
    for c in range(nch):
        works[c].wait()
        got = recvs[c].numpy()
        if c == 0:
            metas = got[:, :lay["pad"]].copy().view(np.int32)
        for r in range(world):
            w64 = metas[r, 4:4 + 2 * nw].copy().view(np.uint64)
            pre = metas[r, 4 + 2 * nw:4 + 3 * nw].copy().view(np.uint32)
            assert metas[r, 0] == others[r][0].size and metas[r, ex.meta_len - 2] == others[r][5]
            assert np.array_equal(metas[r, 1:4].view(np.float32), others[r][4])
            rows_rc = got[r, lay["row0"][c]:].reshape(caps[c], 16)
            start_rc = int(pre[bounds[c] // 64])                        # dp_reduce_kernel: prefix[word0]
            ids = others[r][0]
            for n in ids[(ids >= bounds[c]) & (ids < bounds[c + 1])]:
                word, bit = int(w64[n >> 6]), int(n & 63)
                assert (word >> bit) & 1
                k = int(pre[n >> 6]) + bin(word & ((1 << bit) - 1)).count("1") - start_rc
                assert 0 <= k < caps[c]                                   # (row_cap: never past the chunk's rows)
                row = rows_rc[k]
                assert int(row[15:16].view(np.int32)[0]) == n and not np.isnan(row[:15]).any()
                dense[n] += row[:15]
    ref = np.zeros((N, 15), dtype=np.float64)
    for (ids, rws, *_r) in others:
        ref[ids] += rws[:, :15]
    assert np.array_equal(dense, ref)
    np.save(Path(out_dir) / f"chunked{rank}.npy", dense)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_touched_chunked_exchange_layout_over_gloo(tmp_path, world):
    """finish_touched_chunked()'s wire protocol with `world` ranks over gloo on CPU tensors: per-chunk capacities agreed from the
    ranks' counts (MAX + margin), chunk_layout() -> one message per index chunk (the meta record with the touched rows' map leads
    message 0), gather_chunk_messages() issues them back to back, and a receiver that waits for message c alone finds every
    sender's row of every touched Gaussian of chunk c at row prefix(n) - prefix(first word of the chunk) of that message -- the
    address rule of mtgs_dp_reduce_slices_cap; sums equal the accumulation over all cameras, ranks agree bit for bit.  (The
    kernels on both sides run on the GPU: tests/test_gpu_dp.py.)"""
    port = _free_port()
    mp.spawn(_chunked_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"chunked{r}.npy") for r in range(world)]
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)
