"""The N > 1 exchange on the GPU: two ranks share the single MI355X of the test box (gloo between
them), each renders its own camera through the HIP path, and the sparse factored exchange
(mtgs_amd.dist.SparseGradExchange + csrc/dp.hip) must reproduce the dense all-reduce of the same
gradients."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.util import assert_tile_lists, listed

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MTGS_DIST_BACKEND="gloo")
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from mtgs_amd import dist as mdist, rasterization, spherical_harmonics
    from mtgs_amd.synthetic import make_camera, make_scene
    mdist.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    N, W, H, K = 30_000, 320, 240, 16
    sc = make_scene(N, seed=7, sh_degree=3, extent=(12.0, 4.0, 12.0))
    vm, Kmat = make_camera(W, H, yaw_deg=45.0 * rank)
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    cam_pos = torch.inverse(vm)[0, :3, 3].to(dev)
    g = torch.Generator().manual_seed(rank + 1)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
    # SH output as a leaf: v_rgb is the gradient with respect to it (the factor that is exchanged)
    sh_out = spherical_harmonics(3, P["means"].detach() - cam_pos, P["coeffs"].detach()).requires_grad_(True)
    rgb = torch.clamp(sh_out + 0.5, 0.0, 1.0)
    render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], rgb, vm.to(dev),
                                        Kmat.to(dev), W, H, packed=False, render_mode="RGB+ED",
                                        rasterize_mode="antialiased", absgrad=True)
    torch.autograd.backward([render, alpha], [Gc, Ga])
    # dense reference: full local SH backward, then a dense all-reduce of every gradient
    sh2 = spherical_harmonics(3, P["means"].detach() - cam_pos, P["coeffs"])
    sh2.backward(sh_out.grad)
    dense = {k: P[k].grad.clone() for k in ("means", "quats", "scales", "opacities", "coeffs")}
    for t in dense.values():
        dist.all_reduce(t)
    ex = mdist.SparseGradExchange(N, K, dev)
    loc = {k: P[k].grad.clone() for k in ("means", "quats", "scales", "opacities")}
    o = ex.exchange(info["radii"][0], P["means"].detach(), cam_pos, loc["means"], loc["quats"], loc["scales"],
                    loc["opacities"], sh_out.grad, 3)                       # coefficient gradient rebuilt from zeros
    sparse = dict(zip(("means", "quats", "scales", "opacities", "coeffs"), o))
    loc2 = {k: P[k].grad.clone() for k in ("means", "quats", "scales", "opacities")}
    own_coeffs = P["coeffs"].grad.clone()                                    # the local SH backward from above
    o2 = ex.exchange(info["radii"][0], P["means"].detach(), cam_pos, loc2["means"], loc2["quats"], loc2["scales"],
                     loc2["opacities"], sh_out.grad, 3, local_coeff_grad=lambda: own_coeffs)
    for a_, b_ in zip(o, o2):
        assert torch.allclose(a_, b_, atol=1e-5, rtol=1e-5)
    res = {}
    for k in dense:
        scale = float(dense[k].abs().max())
        res[k] = (float((dense[k] - sparse[k]).abs().max()), scale)
    n_vis = int((info["radii"] > 0).sum())
    tensor_form_bytes = ex.last_bytes
    # integrated form: the exchange renders (colour activation fused, visibility map from the front kernels, all-gathered
    # during the frame), the backward leaves wire rows, finish() exchanges them in chunks -- same sums, same image
    P2 = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    vm2 = vm.to(dev).requires_grad_(True)
    sh3 = spherical_harmonics(3, P2["means"].detach() - cam_pos, P2["coeffs"].detach())
    for chunks in (1, 3):
        ex2 = mdist.SparseGradExchange(N, K, dev, chunks=chunks)
        r2, a2, info2 = ex2.rasterization(P2["means"], P2["quats"], P2["scales"], P2["opacities"], sh3, vm2, Kmat.to(dev), W, H,
                                          cam_pos)
        assert torch.equal(r2, render) and torch.equal(a2, alpha) and torch.equal(listed(info2), listed(info))
        torch.autograd.backward([r2, a2], [Gc, Ga])
        assert all(P2[k].grad is None for k in ("means", "quats", "scales", "opacities")) and vm2.grad is not None
        o3 = ex2.finish(P2["means"], 3)
        for k, t in zip(("means", "quats", "scales", "opacities", "coeffs"), o3):
            scale = float(dense[k].abs().max())
            err = float((dense[k] - t).abs().max())
            assert err <= 1e-5 * scale + 1e-7, f"integrated form, chunks={chunks}, {k}: {err} vs {scale}"
        ph = ex2.phases_ms()
        assert set(ph) == {"meta", "wire", "reduce"} and all(v >= 0 for v in ph.values())
        vm2.grad = None
    # capacity overflow on ONE rank only (its speculative capacities were too small: the frame is repeated with exact sizes);
    # the repeat must not issue a second meta all-gather, or the ranks' collectives pair up wrongly from here on -- two
    # frames in a row, the second one checks that the exchange is still in step
    from mtgs_amd import wrapper
    ex3 = mdist.SparseGradExchange(N, K, dev, chunks=2)
    for frame in range(2):
        old = (wrapper._force_caps, wrapper.speculative_sizing)
        if rank == 0:
            wrapper._force_caps, wrapper.speculative_sizing = (n_vis // 2, 1 << 16), True
        try:
            r3, a3, info3 = ex3.rasterization(P2["means"], P2["quats"], P2["scales"], P2["opacities"], sh3, vm2, Kmat.to(dev), W, H,
                                              cam_pos)
        finally:
            wrapper._force_caps, wrapper.speculative_sizing = old
        assert n_vis > 500 and torch.equal(r3, render) and torch.equal(listed(info3), listed(info))
        torch.autograd.backward([r3, a3], [Gc, Ga])
        o4 = ex3.finish(P2["means"], 3)
        for k, t in zip(("means", "quats", "scales", "opacities", "coeffs"), o4):
            scale = float(dense[k].abs().max())
            err = float((dense[k] - t).abs().max())
            assert err <= 1e-5 * scale + 1e-7, f"overflow repeat on rank 0, frame {frame}, {k}: {err} vs {scale}"
        vm2.grad = None
    # a forward-only frame (evaluation) followed by a training frame: abandon() / the next rasterization() drop the first
    with torch.no_grad():
        ex3.rasterization(P2["means"], P2["quats"], P2["scales"], P2["opacities"], sh3, vm2, Kmat.to(dev), W, H, cam_pos)
    r3, a3, _ = ex3.rasterization(P2["means"], P2["quats"], P2["scales"], P2["opacities"], sh3, vm2, Kmat.to(dev), W, H, cam_pos)
    torch.autograd.backward([r3, a3], [Gc, Ga])
    o5 = ex3.finish(P2["means"], 3)
    assert float((o5[0] - dense["means"]).abs().max()) <= 1e-5 * float(dense["means"].abs().max()) + 1e-7
    np.save(Path(out_dir) / f"r{rank}.npy", np.array([[e, s] for e, s in res.values()] + [[n_vis, tensor_form_bytes]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sparse_exchange_equals_dense_allreduce(tmp_path, hip_lib, world):
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        a = np.load(tmp_path / f"r{r}.npy")
        for (err, scale), name in zip(a[:-1], ("means", "quats", "scales", "opacities", "coeffs")):
            assert scale > 0 and err <= 1e-5 * scale + 1e-7, f"rank {r} {name}: {err} vs {scale}"
        n_vis, nbytes = a[-1]
        assert n_vis > 1000 and 0 < nbytes < 30_000 * 236     # far fewer bytes than the dense exchange


@pytest.mark.parametrize("K,degree", [(16, 2), (16, 3), (25, 4)])
def test_sparse_exchange_single_process(hip_lib, K, degree):
    """world = 1: the exchange is a local pack + reduce of the own rows (identity on the visible rows).  K = 16 takes the
    ordered rows + one-pass reduction, K = 25 / degree 4 the per-sender read-modify-write fallback."""
    from mtgs_amd import dist as mdist
    from mtgs_amd import spherical_harmonics
    dev = torch.device("cuda")
    N = 5003
    g = torch.Generator().manual_seed(0)
    radii = (torch.rand(N, generator=g) > 0.7).int().to(dev)
    vis = radii > 0
    mk = lambda *s: (torch.randn(*s, generator=g).to(dev) * vis.view(-1, *([1] * (len(s) - 1)))).contiguous()
    v_means, v_quats, v_scales, v_opac, v_rgb = mk(N, 3), mk(N, 4), mk(N, 3), mk(N), mk(N, 3)
    means = torch.randn(N, 3, generator=g).to(dev)
    cam = torch.tensor([0.3, -0.2, 0.1], device=dev)
    coeffs = torch.zeros(N, K, 3, device=dev, requires_grad=True)
    spherical_harmonics(degree, means - cam, coeffs).backward(v_rgb)
    ex = mdist.SparseGradExchange(N, K, dev)
    ref = [t.clone() for t in (v_means, v_quats, v_scales, v_opac)] + [coeffs.grad]
    o = ex.exchange(radii, means, cam, v_means, v_quats, v_scales, v_opac, v_rgb, degree)
    for got, r in zip(o, ref):
        assert torch.allclose(got, r, atol=2e-6, rtol=1e-5)


def test_a_data_parallel_frame_evaluates_its_colours_for_the_visible_gaussians_only(hip_lib):
    """SparseGradExchange.rasterization(sh_out=spherical_harmonics(...)): the SH output arrives still deferred (wrapper._LazySH) and
    the frame evaluates SH + clamp for the Gaussians its camera sees only (mtgs_vis_color_fwd_dirs; the wire rows' v_rgb takes the
    clamp's pass-through rule from that kernel's bits, mtgs_project_bwd_rows color_mode 2): the render of the dense-colour frame
    bit for bit, its wire rows up to the order of the compositing atomics, and the same dense sums out of finish()."""
    from mtgs_amd import dist as mdist, spherical_harmonics, wrapper
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda")
    N, W, H, K = 60_000, 480, 272, 16
    sc = make_scene(N, seed=11, sh_degree=3)
    vm, Kmat = make_camera(W, H, yaw_deg=10.0)
    vm, Kmat = vm.to(dev), Kmat.to(dev)
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    cam_pos = torch.inverse(vm)[0, :3, 3]
    g = torch.Generator().manual_seed(3)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
    real = wrapper.call
    out = {}
    for lazy in (True, False):
        ex = mdist.SparseGradExchange(N, K, dev)
        calls = []
        try:
            wrapper.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
            with wrapper.sh_lazy(lazy):
                sh = spherical_harmonics(3, P["means"].detach() - cam_pos, P["coeffs"].detach())
                r, a, info = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm, Kmat, W, H, cam_pos)
                torch.autograd.backward([r, a], [Gc, Ga])
        finally:
            wrapper.call = real
        rows = ex.rows[:ex.n_vis].clone()
        sums = [o.clone() for o in ex.finish(P["means"], 3)]
        out[lazy] = (r.detach().clone(), a.detach().clone(), rows, sums, calls, int(ex.n_vis))
    (r1, a1, rows1, s1, c1, n1), (r0, a0, rows0, s0, c0, n0) = out[True], out[False]
    assert "mtgs_vis_color_fwd_dirs" in c1 and not [n for n in c1 if n.startswith("mtgs_sh_") or n.startswith("mtgs_vis_color_bwd")], c1
    assert "mtgs_sh_fwd" in c0 and "mtgs_vis_color_fwd_dirs" not in c0
    assert n1 == n0 and 0 < n1 < N // 2 and torch.equal(r1, r0) and torch.equal(a1, a0)
    assert torch.equal(rows1[:, 15].view(torch.int32), rows0[:, 15].view(torch.int32))      # the same Gaussians, in index order
    assert torch.equal(rows1[:, :14] != 0, rows0[:, :14] != 0) and float(rows0[:, 11:14].abs().sum()) > 0
    torch.testing.assert_close(rows1[:, :14], rows0[:, :14], rtol=1e-3, atol=1e-5 * float(rows0[:, :14].abs().max()))
    for got, ref in zip(s1, s0):
        torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("exchange", ["sparse", "sparse/touched", "sparse/static", "sparse/dynamic", "dense"])
def test_bench_two_ranks_driver_launch(exchange, hip_lib):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per
    rank), with both ranks on the test box's single GPU and gloo in place of RCCL: the whole N > 1 code path
    (init, exchange, barrier, max-over-ranks timing, ONE JSON line from rank 0) must run."""
    import json
    import subprocess
    env = dict(os.environ, MTGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--n-gaussians", "50000", "--width", "640", "--height", "480", "--dp-exchange", exchange.split("/")[0]]
    finish = exchange.split("/")[1] if "/" in exchange else "touched-chunked"        # (the default of --gpus N > 1)
    if "/" in exchange:
        cmd += ["--dp-finish", finish]
    exchange = exchange.split("/")[0]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    assert "cpu_baseline" not in out and out["roofline"]["launches_timed"] == 3
    assert exchange in out["config"]["parallelism"]
    # the N > 1 line carries the phase breakdown that makes a scaling run diagnostic
    ph = out["dp_phases_ms"]
    assert out["dp_world_size"] == 2 and out["dp_backend"] == "gloo"
    # (touched / static: ONE all-gather inside `exchange`; static also all-gathers the visibility maps during the frame: `meta`;
    #  dynamic: chunked all-gathers pipelined with the reduction, timed as `wire` / `reduce`)
    want = {"render", "exchange"}
    if exchange == "sparse":
        want |= {"touched-chunked": set(), "touched": set(), "static": {"meta"}, "dynamic": {"meta", "wire", "reduce"}}[finish]
        assert {"touched-chunked": "finish_touched_chunked: 2 all-gathers", "touched": "finish_touched: one", "static": "finish_static",
                "dynamic": "finish:"}[finish] in out["config"]["parallelism"]
        # the self-diagnosis of the N > 1 line (round 6): exchange form, overflow flag, per-rank phases, achieved GB/s per rank and link
        assert out["dp_finish"] == finish and len(out["dp_rank_phases_ms"]) == 2 and len(out["dp_exchange_GBs_per_link"]) == 2
        if finish in ("touched-chunked", "touched", "static"):
            assert out["dp_overflow"] is False
        if finish == "touched-chunked":
            assert len(out["dp_chunk_caps_rows"]) == 2 and all(c > 0 for c in out["dp_chunk_caps_rows"])
    assert want <= set(ph) and all(ph[k] >= 0 for k in want), ph


def test_bench_configs3_workload_eight_ranks_on_one_gpu(hip_lib):
    """BASELINE configs[3] at its own size -- 2M shared Gaussians, eight 1920x1080 cameras, one rank per camera -- with the
    eight ranks sharing the test box's single GPU and gloo in place of RCCL: everything but the wire is what the 8-GPU
    run executes (visibility maps all-gathered during the frame, 8 x 19 MB of wire rows in chunks, one-pass reduction
    over eight senders, phase timings)."""
    import json
    import subprocess
    env = dict(os.environ, MTGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "8", "--steps", "2",
           "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["dp_world_size"] == 8 and out["scaling"] == "weak"
    assert out["config"]["n_gaussians"] == 2_000_000 and out["config"]["width"] == 1920
    ph = out["dp_phases_ms"]
    assert {"render", "exchange"} <= set(ph) and "finish_touched" in out["config"]["parallelism"]      # (the default exchange form)
    # every rank receives the rows that carry a gradient of all eight cameras: ~8 x 130k x 64 B + eight 0.37 MB maps
    assert 8 * 100_000 * 64 < out["dp_bytes_received_per_rank"][0] < 8 * 200_000 * 64 + 8 * 400_000
    # every rank receives the visible rows of all eight cameras: ~8 x 300k x 64 B, an order of magnitude below the
    # 8 x 472 MB a dense all-reduce moves
    assert "sparse" in out["config"]["parallelism"]
    from tests.util import REPORT
    REPORT.append({"kind": "dp", "name": "configs[3] workload, 8 ranks on one GPU over gloo", "ms_per_step": out["ms_per_step"],
                   "phases_ms": ph, "parallelism": out["config"]["parallelism"]})


def _worker_traversals(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MTGS_DIST_BACKEND="gloo")
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from mtgs_amd import dist as mdist, spherical_harmonics
    from mtgs_amd.synthetic import make_camera, make_scene
    mdist.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    N, W, H, K, T = 20_000, 320, 240, 16, 3
    t = rank % T                                            # the traversal of this rank's camera
    sc = make_scene(N, seed=9, sh_degree=3, extent=(12.0, 4.0, 12.0))
    g = torch.Generator().manual_seed(100)
    coeffs_T = (torch.randn(N, T, K, 3, generator=g) * 0.2).to(dev)      # one set of SH coefficients per traversal
    vm, Kmat = make_camera(W, H, yaw_deg=50.0 * rank)
    vm, Kmat = vm.to(dev), Kmat.to(dev)
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items() if k != "coeffs"}
    cam_pos = torch.inverse(vm)[0, :3, 3]
    g2 = torch.Generator().manual_seed(rank + 1)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g2).to(dev), torch.randn(1, H, W, 1, generator=g2).to(dev)
    mine = coeffs_T[:, t].contiguous().requires_grad_(True)
    ex = mdist.SparseGradExchange(N, K, dev, traversals=T, chunks=2)
    sh = spherical_harmonics(3, P["means"].detach() - cam_pos, mine.detach())
    r, a, info = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm, Kmat, W, H, cam_pos, traversal=t)
    torch.autograd.backward([r, a], [Gc, Ga])
    g_means, g_quats, g_scales, g_opac, g_coeffs = ex.finish(P["means"], 3)
    assert g_coeffs.shape == (N, T, K, 3)
    # dense reference: the ordinary rasterization, the full local SH backward into this rank's traversal slice, all-reduce
    from mtgs_amd import rasterization
    P2 = {k: v.to(dev).requires_grad_(True) for k, v in sc.items() if k != "coeffs"}
    mine2 = coeffs_T[:, t].contiguous().requires_grad_(True)
    sh2 = spherical_harmonics(3, P2["means"].detach() - cam_pos, mine2)
    r2, a2, _ = rasterization(P2["means"], P2["quats"], P2["scales"], P2["opacities"], torch.clamp(sh2 + 0.5, 0.0, 1.0), vm, Kmat,
                              W, H, packed=False, render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
    torch.autograd.backward([r2, a2], [Gc, Ga])
    dense_c = torch.zeros(N, T, K, 3, device=dev)
    dense_c[:, t] = mine2.grad
    dense = [P2["means"].grad.clone(), P2["quats"].grad.clone(), P2["scales"].grad.clone(), P2["opacities"].grad.clone(), dense_c]
    for d in dense:
        dist.all_reduce(d)
    errs = []
    for got, ref in zip((g_means, g_quats, g_scales, g_opac, g_coeffs), dense):
        errs.append([float((got - ref).abs().max()), float(ref.abs().max())])
    np.save(Path(out_dir) / f"t{rank}.npy", np.array(errs))
    # finish_static(): the same sums without any host read or wait (fixed-capacity all-gather, traversals from the schedule) --
    # from a second frame of the same inputs; equal to finish() up to the order of the compositing atomics.  Then with a
    # capacity that is too small for every rank: the overflow flag, and nothing is read out of bounds.
    counts = int(ex._samples_host[rank][0])
    assert counts > 200
    for cap, want_overflow in ((N, False), (200, True)):      # (the capacity is a constant of the job: the same on every rank)
        r3, a3, _ = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm, Kmat, W, H, cam_pos, traversal=t)
        torch.autograd.backward([r3, a3], [Gc, Ga])
        out_s, ovf = ex.finish_static(P["means"], 3, cap, [q % T for q in range(world)])
        torch.cuda.synchronize()
        assert bool(ovf) == want_overflow, (rank, cap, counts)
        if not want_overflow:
            for got, ref in zip(out_s, (g_means, g_quats, g_scales, g_opac, g_coeffs)):
                assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-7, rank
        else:
            assert all(bool(torch.isfinite(o).all()) for o in out_s)
    # finish_touched(): only the rows that carry a gradient travel, with their own map, in ONE all-gather -- the front end's
    # visibility maps are not exchanged at all (defer_maps).  From the same wire rows the sums are those of finish_static()
    # BIT FOR BIT (zero rows add exact zeros, the order of the others is unchanged); then a capacity that is too small.
    ex.defer_maps = True
    for cap, want_overflow in ((N, False), (64, True)):
        r4, a4, _ = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm, Kmat, W, H, cam_pos, traversal=t)
        torch.autograd.backward([r4, a4], [Gc, Ga])
        rows_before = ex.rows[:ex.n_vis].clone()
        out_t, ovf = ex.finish_touched(P["means"], 3, cap, [q % T for q in range(world)])
        torch.cuda.synchronize()
        n_touched = int(ex.touched_count)
        assert bool(ovf) == want_overflow and 64 < n_touched < ex.n_vis, (rank, cap, n_touched, ex.n_vis)
        assert n_touched == int((rows_before[:, :14] != 0).any(1).sum())
        assert ex.last_bytes == world * 4 * (cap * 16 + -(-ex.meta_len // 16) * 16)
        if not want_overflow:
            # the reference from the SAME rows: the visibility-map form needs the maps after all -- gather them now
            ex.defer_maps = False
            ex._pending = {"stage": "forward"}
            ex.after_front()
            ex._pending.update(stage="rows", n_vis=ex.n_vis)
            out_v, ovf_v = ex.finish_static(P["means"], 3, N, [q % T for q in range(world)])
            ex.defer_maps = True
            torch.cuda.synchronize()
            assert not bool(ovf_v)
            for got, ref in zip(out_t, out_v):
                assert torch.equal(got, ref), rank
        else:
            assert all(bool(torch.isfinite(o).all()) for o in out_t)
            # the overflow's recovery through the PUBLIC API: the frame is still there, finish_recover() gathers the
            # visibility maps this frame never exchanged (defer_maps) and repeats the exchange untruncated: finish()'s sums
            out_r = ex.finish_recover(P["means"], 3)
            torch.cuda.synchronize()
            for got, ref in zip(out_r, (g_means, g_quats, g_scales, g_opac, g_coeffs)):
                assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-7, rank
    # A sender that overflows must not have the words BEHIND its rows summed as floats (the block is [rows | map], and a
    # map word of all-one bits is a NaN): every Gaussian visible AND touched makes the map dense, the capacity is far too small.
    r5, a5, _ = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm, Kmat, W, H, cam_pos, traversal=t)
    torch.autograd.backward([r5, a5], [Gc, Ga])
    ex.rows[:N, :14] = 1.0                                  # every row "carries a gradient" ...
    ex.rows[:N, 15] = torch.arange(N, device=dev, dtype=torch.int32).view(torch.float32)     # ... of Gaussian n: dense all-ones map words
    ex.n_vis = N
    out_p, ovf = ex.finish_touched(P["means"], 3, 64, [q % T for q in range(world)])
    torch.cuda.synchronize()
    assert bool(ovf) and all(bool(torch.isfinite(o).all()) for o in out_p), rank
    assert float(out_p[0].abs().max()) <= world, rank       # (sums of the 1.0 rows: nothing else was read)
    # finish_touched_chunked(): the same rows in index chunks, one all-gather per chunk issued back to back, chunk c reduced while
    # c + 1 is on the wire; bit-identical to finish_touched(); per-chunk capacities from a warm-up step's counts (MAX over ranks);
    # then one chunk too small: the overflow flag, finite sums, and the recovery
    exc = mdist.SparseGradExchange(N, K, dev, traversals=T, chunks=3)
    exc.defer_maps = True
    trav = [q % T for q in range(world)]

    def frame(e):
        rr, aa, _ = e.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm, Kmat, W, H, cam_pos, traversal=t)
        torch.autograd.backward([rr, aa], [Gc, Ga])
    frame(exc)
    _, ovf0 = exc.finish_touched_chunked(P["means"], 3, [N] * exc.n_chunks, trav)      # warm-up: generous capacities
    cnt = exc.touched_chunk_counts().to(torch.float64)
    assert not bool(ovf0) and int(cnt.sum()) == int(exc.touched_count) and exc.n_chunks == 3
    dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
    caps = [int(c) + 8 for c in cnt.tolist()]
    frame(exc)
    out_c, ovf_c = exc.finish_touched_chunked(P["means"], 3, caps, trav)
    lay = exc.chunk_layout(caps)
    assert exc.last_bytes == world * 4 * sum(lay["floats"])
    frame(exc)
    out_1, ovf_1 = exc.finish_touched(P["means"], 3, N, trav)
    torch.cuda.synchronize()
    assert not bool(ovf_c) and not bool(ovf_1)
    for got, ref in zip(out_c, out_1):
        # (two frames: the compositing atomics may sum in another order -- the exchange itself adds nothing to that)
        assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-7, rank
    # same frame, both forms: BIT identity (the chunked form re-reads the rows buffer the unchunked one left untouched)
    exc._pending = dict(exc._recover)
    out_c2, _ = exc.finish_touched_chunked(P["means"], 3, caps, trav)
    torch.cuda.synchronize()
    for got, ref in zip(out_c2, out_1):
        assert torch.equal(got, ref), rank
    # prezero: the dense sums allocated when the frame starts, cleared by the frame's compositing forward, and the reduction writes
    # the touched Gaussians only -- bit-identical to the dense write of the same frame's rows, in both forms
    exc.prezero = True
    frame(exc)
    assert exc.zero_region is not None and exc.zero_region[1] >= 4 * N * (11 + T * K * 3)
    out_z, ovf_z = exc.finish_touched_chunked(P["means"], 3, caps, trav)
    assert exc.zero_region is None
    exc._pending = dict(exc._recover)
    out_d, _ = exc.finish_touched_chunked(P["means"], 3, caps, trav)        # (no region left: the dense write)
    torch.cuda.synchronize()
    assert not bool(ovf_z)
    for got, ref in zip(out_z, out_d):
        assert torch.equal(got, ref), rank
    frame(exc)
    out_z1, _ = exc.finish_touched(P["means"], 3, N, trav)
    exc._pending = dict(exc._recover)
    out_d1, _ = exc.finish_touched(P["means"], 3, N, trav)
    torch.cuda.synchronize()
    for got, ref in zip(out_z1, out_d1):
        assert torch.equal(got, ref), rank
    exc.prezero = False
    small = list(caps)
    small[1] = 4
    frame(exc)
    out_o, ovf_o = exc.finish_touched_chunked(P["means"], 3, small, trav)
    torch.cuda.synchronize()
    assert bool(ovf_o) and all(bool(torch.isfinite(o).all()) for o in out_o), rank
    out_r = exc.finish_recover(P["means"], 3)
    torch.cuda.synchronize()
    for got, ref in zip(out_r, out_1):
        assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-7, rank
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_exchange_routes_colour_gradients_to_the_senders_traversal(tmp_path, hip_lib):
    """Per-traversal appearance (MTGS's multi-colour nodes): four ranks render cameras of traversals 0, 1, 2, 0; the
    coefficient gradient comes back as [N, T, K, 3] with every sender's factor in its own traversal's slice, equal to the
    dense all-reduce of the per-traversal gradients; the geometry sums over all ranks as before."""
    import torch.multiprocessing as mp
    world = 4
    mp.spawn(_worker_traversals, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        for (err, scale), name in zip(np.load(tmp_path / f"t{r}.npy"), ("means", "quats", "scales", "opacities", "coeffs[N,T,K,3]")):
            # (fp32 atomics in another order on the two paths: a few 1e-6 of the largest gradient)
            assert scale > 0 and err <= 3e-5 * scale + 1e-7, f"rank {r} {name}: {err} vs {scale}"


def _worker_rows(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MTGS_DIST_BACKEND="gloo")
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from mtgs_amd import dist as mdist, spherical_harmonics
    from mtgs_amd.optim import FusedAdam
    from mtgs_amd.synthetic import make_camera, make_scene
    mdist.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    N, W, H, K, T = 20_000, 320, 240, 16, 3
    sc = make_scene(N, seed=9, sh_degree=3, extent=(12.0, 4.0, 12.0))
    g = torch.Generator().manual_seed(100)
    coeffs0 = torch.randn(N, T, K, 3, generator=g) * 0.2                  # one set of SH coefficients per traversal
    geo = ("means", "quats", "scales", "opacities")

    def make():      # the rasterizer's inputs as the optimizer's parameters (the exchange sums gradients with respect to them)
        P = {k: sc[k].clone().to(dev).requires_grad_(True) for k in geo}
        P["coeffs"] = coeffs0.clone().to(dev).requires_grad_(True)
        return P, FusedAdam([{"params": [P[k]], "lr": lr} for k, lr in (("means", 1e-3), ("quats", 1e-3), ("scales", 1e-3),
                                                                         ("opacities", 1e-2), ("coeffs", 5e-3))], eps=1e-15)
    PA, oa = make()                      # dense gradients, every row stepped
    PB, ob = make()                      # row gradients, the coefficient tensor row-lazy
    ob.set_row_lazy(PB["coeffs"], traversals=T)
    every = torch.zeros(N, dtype=torch.int32, device=dev)
    ex = mdist.SparseGradExchange(N, K, dev, traversals=T, chunks=2)
    stats = {"max_rows": 0, "union": 0}
    for step in range(3):
        t = (rank + step) % T                                            # ranks 0 and 3 share a traversal in every step
        vm, Kmat = make_camera(W, H, yaw_deg=50.0 * rank + 7.0 * step)
        vm, Kmat = vm.to(dev), Kmat.to(dev)
        cam_pos = torch.inverse(vm)[0, :3, 3]
        g2 = torch.Generator().manual_seed(10 * step + rank + 1)
        Gc, Ga = torch.randn(1, H, W, 4, generator=g2).to(dev), torch.randn(1, H, W, 1, generator=g2).to(dev)
        ob.catch_up_rows([(PB["coeffs"], every, t)])                     # the sender evaluates SH for every Gaussian of slice t
        assert torch.equal(PB["coeffs"][:, t], PA["coeffs"][:, t]), (rank, step)
        for k in geo:
            assert torch.equal(PA[k], PB[k]), (rank, step, k)
        sh = spherical_harmonics(3, PB["means"].detach() - cam_pos, PB["coeffs"][:, t].detach().contiguous())
        leaves = {k: PB[k].detach().requires_grad_(True) for k in geo}
        r, a, info = ex.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], sh, vm, Kmat, W, H,
                                      cam_pos, traversal=t)
        torch.autograd.backward([r, a], [Gc, Ga])
        dense, R = ex.finish(leaves["means"], 3, rows="both")            # ONE set of wire rows, both forms of the sums
        # (a) every row equals the dense entry bit for bit, and nothing else is non-zero
        ro = R["geo_row_of"].long()
        seen = ro >= 0
        n_u = int(seen.sum())
        stats["union"], stats["max_rows"] = n_u, max(stats["max_rows"], R["geo_rows"].shape[0])
        assert int(R["geo_totals"][0] >> 32) == n_u and torch.equal(R["geo_ids"][:n_u].long(), torch.nonzero(seen).flatten())
        for k, (c0, c1) in zip(geo, ((0, 3), (3, 7), (7, 10), (10, 11))):
            d = dense[geo.index(k)].reshape(N, -1)
            assert torch.equal(R["geo_rows"][ro[seen], c0:c1], d[seen]) and not bool(d[~seen].any()), (rank, step, k)
        assert set(R["coef"]) == {(q + step) % T for q in range(world)}
        for tt in range(T):
            dt = dense[4][:, tt].reshape(N, -1)
            if tt in R["coef"]:
                rows_t, ro_t = R["coef"][tt]
                s_t = ro_t >= 0
                assert torch.equal(rows_t[ro_t.long()[s_t]], dt[s_t]) and not bool(dt[~s_t].any()), (rank, step, tt)
            else:
                assert not bool(dt.any())
        # (b) dense -> optimizer == rows -> optimizer
        for k, gk in zip(geo + ("coeffs",), dense):
            PA[k].grad = gk
        oa.step()
        for k, c0 in zip(geo, (0, 3, 7, 10)):
            ob.set_row_gradient(PB[k], R["geo_rows"], R["geo_row_of"], c0)
        for tt, (rows_t, ro_t) in R["coef"].items():
            ob.set_row_gradient(PB["coeffs"], rows_t, ro_t, 0, slice_index=tt)
        ob.step()
    ob.flush()
    same = {k: bool(torch.equal(PA[k], PB[k])) for k in PA}
    same.update({"m_" + k: bool(torch.equal(oa.state[PA[k]]["exp_avg"], ob.state[PB[k]]["exp_avg"])) for k in PA})
    moved = float((PA["coeffs"].detach().cpu() - coeffs0).abs().max())
    import json
    json.dump({"same": same, "moved": moved, "union": stats["union"], "cap": stats["max_rows"]}, open(Path(out_dir) / f"rows{rank}.json", "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_exchange_rows_into_the_optimizer_are_bit_identical_to_dense(tmp_path, hip_lib):
    """SparseGradExchange.finish(rows=True): the receiver's sums leave as compact rows of the UNION of the ranks' visible sets
    (mtgs_dp_union / mtgs_dp_reduce_rows) -- geometry over all ranks, the SH coefficient gradient per traversal over that
    traversal's ranks -- and go to FusedAdam as row gradients, one slice per rendered traversal, the [N, T, K, 3] tensor row-lazy.
    Four ranks, three traversals (two ranks share one in every step), three steps with moving cameras, from ONE set of wire rows
    per step: (a) every row equals the dense tensor's entry bit for bit and the dense tensors are zero elsewhere; (b) after three
    steps the parameters and the moments of the row-fed optimizer equal those of the optimizer fed the dense gradients, bit for
    bit (reference: the per-traversal tensors of multi_color_gaussian_splatting.py:53-80 under the DDP site custom_pipeline.py:87-89)."""
    import json
    import torch.multiprocessing as mp
    world = 4
    mp.spawn(_worker_rows, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        out = json.load(open(tmp_path / f"rows{r}.json"))
        assert all(out["same"].values()), (r, out)
        assert out["moved"] > 1e-3 and 0 < out["union"] < 20_000, out


def _worker_configs3(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MTGS_DIST_BACKEND="gloo")
    sys.path.insert(0, str(ROOT))
    import json
    import torch.distributed as dist
    from mtgs_amd import dist as mdist, spherical_harmonics
    from mtgs_amd.synthetic import make_camera, make_scene
    mdist.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    N, W, H, K = 2_000_000, 1920, 1080, 16
    sc = make_scene(N, seed=0, sh_degree=3)                       # WB-v1, as bench.py --gpus 8
    cams = [make_camera(W, H, yaw_deg=45.0 * r) for r in range(world)]

    def cotangents(r):
        g = torch.Generator().manual_seed(1000 + r)
        return torch.randn(1, H, W, 4, generator=g), torch.randn(1, H, W, 1, generator=g)

    vm, Kmat = cams[rank]
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    cam_pos = torch.inverse(vm)[0, :3, 3].to(dev)
    Gc, Ga = cotangents(rank)
    ex = mdist.SparseGradExchange(N, K, dev)
    sh = spherical_harmonics(3, P["means"].detach() - cam_pos, P["coeffs"].detach())
    r_, a_, info = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm.to(dev), Kmat.to(dev), W, H, cam_pos)
    torch.autograd.backward([r_, a_], [Gc.to(dev), Ga.to(dev)])
    sums = ex.finish(P["means"], 3)
    torch.cuda.synchronize()
    if rank == 0:
        # the oracle's eight-camera backward, summed over the cameras in fp64 (one camera at a time: ~7 s each on the box)
        from oracle import oracle as orc
        from tests import util
        orc.build()
        a = {k: v.numpy() for k, v in sc.items()}
        acc = {k: np.zeros(a[k].shape, np.float64) for k in ("means", "quats", "scales", "opacities", "coeffs")}
        # Row accounting of the 8-camera sums (round-3 review: a percentile alone averages single rows away).  Per camera the
        # oracle also gives, per visible Gaussian, the sums of |per-pixel terms| of its compositing rows -- over all pixels
        # (how well a cancelling sum can be conditioned) and over the threshold-critical pixels (how far flipped decisions can
        # move it).  Both are pushed through |J|, the element-wise absolute projection / SH VJP of that camera (seven
        # unit-vector calls of the oracle's linear VJP), and summed over the cameras: ta / ca per PARAMETER entry.
        ta = {k: np.zeros(a[k].shape, np.float64) for k in acc}
        ca = {k: np.zeros(a[k].shape, np.float64) for k in acc}

        def push_abs(vm_r, K_r, m, rows, into):
            v2d_t, dep_t, con_t, comp_t = rows                      # [1,N,2] [1,N] [1,N,3] [1,N], all >= 0
            z2, z1, z3 = np.zeros_like(v2d_t), np.zeros_like(dep_t), np.zeros_like(con_t)
            units = []
            for j in range(2):
                e = z2.copy(); e[..., j] = 1.0
                units.append(((e, z1, z3, z1), v2d_t[0, :, j]))
            units.append(((z2, np.ones_like(z1), z3, z1), dep_t[0]))
            for j in range(3):
                e = z3.copy(); e[..., j] = 1.0
                units.append(((z2, z1, e, z1), con_t[0, :, j]))
            units.append(((z2, z1, z3, np.ones_like(z1)), comp_t[0]))
            for (e2, e1, e3, ec), w in units:
                jm, jq, js, _ = orc.project_bwd(a["means"], a["quats"], a["scales"], vm_r.numpy(), K_r.numpy(), W, H, 0.3, m["radii"],
                                                m["conics"], m["compensations"], e2, e1, e3, ec, need_v_viewmats=False)
                into["means"] += np.abs(jm) * w[:, None]
                into["quats"] += np.abs(jq) * w[:, None]
                into["scales"] += np.abs(js) * w[:, None]

        for r in range(world):
            vm_r, K_r = cams[r]
            dirs = a["means"] - torch.inverse(vm_r)[0, :3, 3].numpy()
            x = orc.sh_fwd(3, dirs, a["coeffs"])
            rgb = np.clip(x + 0.5, 0.0, 1.0)
            r_ref, a_ref, m = orc.rasterization(a["means"], a["quats"], a["scales"], a["opacities"], rgb, vm_r.numpy(), K_r.numpy(),
                                                W, H, render_mode="RGB+ED", rasterize_mode="antialiased")
            if r == 0:   # this rank's own frame: the exchange renders what rasterization() renders
                assert_tile_lists(info, m)
            Gc_r, Ga_r = (t.numpy() for t in cotangents(r))
            alc = np.maximum(a_ref, 1e-10)
            Gc_raw = Gc_r.copy()
            Gc_raw[..., -1:] = Gc_r[..., -1:] / alc
            Ga_tot = Ga_r - (m["render_raw"][..., -1:] / alc ** 2) * Gc_r[..., -1:] * (a_ref > 1e-10)
            v2d, vabs, vcon, vcol, vop, tabs, ctabs = orc.blend_bwd(m["means2d"], m["conics"], m["colors"], m["opacities"], None, W, H, 16,
                                                                    m["isect_offsets"], m["flatten_ids"], a_ref, m["last_ids"], Gc_raw,
                                                                    Ga_tot, want_term_abs=True, pixel_mask=m["critical"])
            vm_, vq_, vs_, _ = orc.project_bwd(a["means"], a["quats"], a["scales"], vm_r.numpy(), K_r.numpy(), W, H, 0.3, m["radii"],
                                               m["conics"], m["compensations"], v2d, vcol[..., -1].copy(), vcon,
                                               vop * a["opacities"][None], need_v_viewmats=False)
            mask = (x + 0.5 > 0.0) & (x + 0.5 < 1.0)
            vc_, _ = orc.sh_bwd(3, dirs, a["coeffs"], vcol[0, :, :3] * mask)
            for k, v in (("means", vm_), ("quats", vq_), ("scales", vs_), ("opacities", (vop * m["compensations"]).sum(0)),
                         ("coeffs", vc_)):
                acc[k] += v
            op = a["opacities"][None]
            # (the xy terms in the form the packed backward sums them: a h dx and b h dy per pixel, tests/util.py::moment_xy_terms)
            push_abs(vm_r, K_r, m, (util.moment_xy_terms(vabs, tabs, m["conics"], m["opacities"]), tabs[..., 7], tabs[..., 0:3],
                                    tabs[..., 3] * op), ta)          # (channel 3 of D = 4: depth)
            push_abs(vm_r, K_r, m, (ctabs[..., 0:2], ctabs[..., 9], ctabs[..., 2:5], ctabs[..., 5] * op), ca)
            ta["opacities"] += (tabs[..., 3] * m["compensations"]).sum(0)
            ca["opacities"] += (ctabs[..., 5] * m["compensations"]).sum(0)
            basis_abs = np.abs(orc.sh_bwd(3, dirs, a["coeffs"], np.ones_like(vcol[0, :, :3]) * mask)[0])   # |B_k(dir)| x clamp mask
            ta["coeffs"] += basis_abs * tabs[0, :, None, 4:7]
            ca["coeffs"] += basis_abs * ctabs[0, :, None, 6:9]
        case = "configs[3]: 2M Gaussians, 8 cameras 1920x1080, sparse exchange over 8 ranks vs the oracle's 8-camera sum"
        failures = []
        for k, got in zip(("means", "quats", "scales", "opacities", "coeffs"), sums):
            try:
                # (opacity: the sum with the most cancellation under random cotangents, as at one camera)
                # every row over 1e-3: a cancelling sum within TERM_REL of its sum of |terms|, or within that plus the terms
                # its critical pixels carry (any of the eight cameras) -- nothing else
                flat = lambda t: t.reshape(t.shape[0], -1) if t.ndim > 1 else t
                util.assert_grad_close("sum v_" + k, flat(got.detach().cpu().numpy()), flat(acc[k].astype(np.float32)), case=case,
                                       row_rel_p999=2.5e-3 if k == "opacities" else 1e-3, term_abs=flat(ta[k]), crit_abs=flat(ca[k]))
            except AssertionError as e:
                failures.append(str(e))
        with open(Path(out_dir) / "report.json", "w") as f:
            json.dump({"report": util.REPORT, "failures": failures}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_configs3_full_size_gradient_sums_vs_oracle(tmp_path, hip_lib):
    """BASELINE configs[3] NUMERICALLY at its own size: 2M shared Gaussians, eight 1920x1080 cameras, one rank per camera
    (the eight ranks share the test box's one GPU, gloo in place of RCCL).  The gradient sums the sparse exchange hands to
    the optimizer -- v_means, v_quats, v_scales, v_opacities and the SH coefficient gradient rebuilt from 3 floats per row
    on the receivers -- against the oracle's eight-camera backward summed in fp64, with the per-row bars of the
    one-camera comparisons (reference semantics: one camera per step and rank, mtgs_scene_graph.py:641-690)."""
    import json
    import torch.multiprocessing as mp
    from tests.util import REPORT
    world = 8
    mp.spawn(_worker_configs3, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = json.load(open(tmp_path / "report.json"))
    REPORT.extend(out["report"])
    assert not out["failures"], out["failures"]
    assert len([r for r in out["report"] if r["kind"] == "gradient"]) == 5


def _worker_nccl_world1(out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from mtgs_amd import dist as mdist, spherical_harmonics
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    N, W, H, K = 20_000, 320, 240, 16
    sc = make_scene(N, seed=7, sh_degree=3, extent=(12.0, 4.0, 12.0))
    vm, Kmat = make_camera(W, H)
    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    cam_pos = torch.inverse(vm)[0, :3, 3].to(dev)
    ex = mdist.SparseGradExchange(N, K, dev, chunks=2)
    ex.world_collectives = True            # a one-rank group short-cuts the collectives: take their code path anyway
    sh = spherical_harmonics(3, P["means"].detach() - cam_pos, P["coeffs"].detach())
    g = torch.Generator().manual_seed(1)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
    r_, a_, info = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm.to(dev), Kmat.to(dev), W, H, cam_pos)
    torch.autograd.backward([r_, a_], [Gc, Ga])
    sums = ex.finish(P["means"], 3)
    t = torch.ones(1024, device=dev)
    dist.all_reduce(t)                      # and a plain RCCL all-reduce (the dense exchange's collective)
    # the default exchange of bench.py --gpus N: one RCCL all-gather of [rows that carry a gradient | their map]
    ex.defer_maps = True
    r2, a2, _ = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm.to(dev), Kmat.to(dev), W, H, cam_pos)
    torch.autograd.backward([r2, a2], [Gc, Ga])
    sums_t, ovf = ex.finish_touched(P["means"], 3, N, [0])
    torch.cuda.synchronize()
    same = all(float((x - y).abs().max()) <= 3e-5 * float(y.abs().max()) + 1e-7 for x, y in zip(sums_t, sums))
    # ... and its chunked, overlapped form: two RCCL all-gathers issued back to back, the reductions joined by stream order alone
    r3, a3, _ = ex.rasterization(P["means"], P["quats"], P["scales"], P["opacities"], sh, vm.to(dev), Kmat.to(dev), W, H, cam_pos)
    torch.autograd.backward([r3, a3], [Gc, Ga])
    sums_c, ovf_c = ex.finish_touched_chunked(P["means"], 3, [N] * ex.n_chunks, [0])
    torch.cuda.synchronize()
    same = same and all(float((x - y).abs().max()) <= 3e-5 * float(y.abs().max()) + 1e-7 for x, y in zip(sums_c, sums)) and not bool(ovf_c)
    ok = bool(torch.isfinite(sums[0]).all()) and float(t.sum()) == 1024.0 and float(sums[0].abs().max()) > 0 and same and not bool(ovf) \
        and ex.last_bytes > 0
    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    with open(out_path, "w") as f:
        f.write(f"{int(ok)} {ver} {dist.get_backend()}")
    dist.destroy_process_group()


def test_rccl_code_path_runs_once_on_the_box(tmp_path, hip_lib):
    """The first execution of the RCCL path itself: a one-rank `nccl` process group on the test box's GPU, the sparse
    exchange with its side-stream all-gathers FORCED onto the collectives (world = 1 normally short-cuts them), and one
    plain all-reduce.  No scaling is measured here -- only that the library loads, the communicator initialises and the
    collectives the N > 1 run issues complete on this image."""
    import subprocess
    code = (f"import sys; sys.path.insert(0, {str(ROOT)!r}); from tests.test_gpu_dp import _worker_nccl_world1; "
            f"_worker_nccl_world1({str(tmp_path / 'out.txt')!r})")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    ok, ver, backend = (tmp_path / "out.txt").read_text().split()
    assert ok == "1" and backend == "nccl", (ok, ver, backend)
    from tests.util import REPORT
    REPORT.append({"kind": "dp", "name": "one-rank RCCL process group: sparse exchange (forced collectives) + all-reduce", "rccl_version": ver})
