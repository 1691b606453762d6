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
