"""mtgs_amd.densify.update_statistics against the reference's masked-tensor formulation
(/root/reference/mtgs/scene_model/mtgs_scene_graph.py:1157-1183 and
 /root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:448-474), restated line by line."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(stats, radii, xys_absgrad, submodel_mask, W, H):
    """update_submodel_statistics + after_train for one node (tensors on CPU)."""
    xys_grad_norm, vis_counts, max_2Dsize = stats
    grads = xys_absgrad[0, submodel_mask].detach()
    image_size = grads.new_tensor([W, H]).unsqueeze(0)
    grads = (grads * image_size * 0.5).norm(dim=-1)
    node_radii = radii[0, submodel_mask]
    visible_mask = (node_radii > 0).flatten()
    vis_counts[visible_mask] += +1
    xys_grad_norm[visible_mask] += grads[visible_mask]
    newradii = node_radii.detach()[visible_mask]
    max_2Dsize[visible_mask] = torch.maximum(max_2Dsize[visible_mask], newradii)


@pytest.mark.parametrize("sizes", [(5000,), (1200, 1, 3333, 64)])
def test_update_statistics_matches_reference(hip_lib, sizes):
    from mtgs_amd.densify import update_statistics
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(4)
    N, W, H = sum(sizes), 960, 540
    model_id = torch.cat([torch.full((n,), i) for i, n in enumerate(sizes)])
    stats_ref = [[torch.rand(n, generator=g), torch.ones(n) + torch.randint(0, 5, (n,), generator=g).float(),
                  torch.rand(n, generator=g) * 30] for n in sizes]
    stats_dev = [[t.clone().to(dev) for t in s] for s in stats_ref]
    for step in range(3):
        radii = (torch.randint(0, 40, (1, N), generator=g) * (torch.rand(1, N, generator=g) < 0.3)).int()
        absgrad = torch.rand(1, N, 2, generator=g) * 1e-3
        start = 0
        for i, n in enumerate(sizes):
            _reference(stats_ref[i], radii, absgrad, model_id == i, W, H)
            update_statistics(*stats_dev[i], radii.to(dev), absgrad.to(dev), W, H, start=start)
            start += n
    for sr, sd in zip(stats_ref, stats_dev):
        for a, b, name in zip(sr, sd, ("xys_grad_norm", "vis_counts", "max_2Dsize")):
            assert torch.allclose(a, b.cpu(), rtol=1e-6, atol=1e-7), name


def test_update_statistics_all_nodes_one_launch(hip_lib):
    """update_statistics_all (one launch for the scene graph) == update_statistics per node, bit for bit, with 120 nodes
    of awkward sizes (0, 1, 255, 256, 257 ...)."""
    from mtgs_amd.densify import update_statistics, update_statistics_all
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    sizes = [4000, 0, 1, 255, 256, 257, 63, 64] + [int(x) for x in torch.randint(1, 700, (112,), generator=g)]
    N, W, H = sum(sizes), 960, 540
    mk = lambda: [[torch.rand(n, generator=g).to(dev), (torch.ones(n) + torch.randint(0, 5, (n,), generator=g).float()).to(dev),
                   (torch.rand(n, generator=g) * 30).to(dev)] for n in sizes]
    a = mk()
    b = [[t.clone() for t in s] for s in a]
    for step in range(2):
        radii = (torch.randint(0, 40, (1, N), generator=g) * (torch.rand(1, N, generator=g) < 0.3)).int().to(dev)
        absgrad = (torch.rand(1, N, 2, generator=g) * 1e-3).to(dev)
        start = 0
        for i, n in enumerate(sizes):
            update_statistics(*a[i], radii, absgrad, W, H, start=start)
            start += n
        update_statistics_all([tuple(s) for s in b], radii, absgrad, W, H)
    for sa, sb in zip(a, b):
        for x, y in zip(sa, sb):
            assert torch.equal(x, y)


def _refine_case(N, seed, T=None):
    g = torch.Generator().manual_seed(seed)
    p = {"means": (torch.rand(N, 3, generator=g) * 2 - 1) * torch.tensor([60.0, 8.0, 140.0]),   # some beyond |x| = 100
         "scales": torch.log(torch.rand(N, 3, generator=g) * 0.6 + 0.01), "quats": torch.randn(N, 4, generator=g) * 1.7,
         "opacities": torch.randn(N, 1, generator=g) * 3.0, "features_dc": torch.randn(N, 3, generator=g)}
    if T is None:
        p["features_rest"] = torch.randn(N, 15, 3, generator=g)
    else:                                            # multi-colour node: per-traversal appearance tensors
        p["features_rest"] = torch.randn(N, T, 15, 3, generator=g)
        p["features_adapters"] = torch.randn(N, T, 3, generator=g)
    stats = (torch.rand(N, generator=g) * 0.004 * 7, torch.randint(1, 12, (N,), generator=g).float(),
             torch.rand(N, generator=g) * 180.0)
    moments = {k: (torch.randn(v.shape, generator=g), torch.rand(v.shape, generator=g)) for k, v in p.items()}
    return p, stats, moments


@pytest.mark.parametrize("step,clone,T", [(500, True, None), (4000, True, None), (4000, False, 3), (16000, True, None)])
def test_refinement_on_device_equals_the_reference_restatement(hip_lib, step, clone, T):
    """split / duplicate / cull + optimizer-moment surgery of one node (mtgs_amd.densify.refine_gaussians, csrc/refine.hip)
    against the numpy restatement of refinement_after (oracle/refine_oracle.py, which follows the reference line by line),
    fed with the SAME samples (Philox4x32-10 keyed by seed / step / index / slot, restated in numpy): masks, counts and
    order identical, every row equal.  Steps: before the world-size cull starts (500), with every rule on (4000), after
    the screen-size rules stop (16000 with stop_split_at raised)."""
    from mtgs_amd.densify import RefineConfig, refine_gaussians
    from oracle import refine_oracle as ro
    N, seed = 20000, 1234567
    cfg = RefineConfig(clone_sample_means=clone, stop_split_at=20000)
    p, stats, moments = _refine_case(N, seed=step, T=T)
    dev = torch.device("cuda")
    new, new_m, info = refine_gaussians({k: v.to(dev) for k, v in p.items()}, tuple(s.to(dev) for s in stats), cfg, step, seed,
                                        moments={k: (a.to(dev), b.to(dev)) for k, (a, b) in moments.items()})
    ref, ref_m, masks = ro.refinement_after({k: v.numpy() for k, v in p.items()}, tuple(s.numpy() for s in stats), cfg, step,
                                            lambda idx, slot: ro.normals3(seed, step, idx, slot),
                                            moments={k: (a.numpy(), b.numpy()) for k, (a, b) in moments.items()})
    n_ref = ref["means"].shape[0]
    assert masks["splits"].sum() > 200 and masks["dups"].sum() > 200 and (~masks["keep"]).sum() > 200, "the case must exercise every branch"
    assert info["n_after"] == n_ref, (info["n_after"], n_ref)
    assert int(info["n_split"]) == int(masks["splits"].sum())
    assert np.array_equal(info["src_index"].cpu().numpy(), masks["src_index"]) and np.array_equal(info["kind"].cpu().numpy(), masks["kind"])
    for k in ref:
        got = new[k].cpu().numpy().astype(np.float64)
        assert got.shape == ref[k].shape, k
        tol = 2e-5 if k == "means" else 2e-6          # (means: float32 Box-Muller + rotation against float64)
        assert np.abs(got - ref[k]).max() <= tol * max(1.0, np.abs(ref[k]).max()), (k, np.abs(got - ref[k]).max())
        for j in (0, 1):
            gm = new_m[k][j].cpu().numpy().astype(np.float64)
            assert np.array_equal(gm, ref_m[k][j].astype(np.float32).astype(np.float64)), (k, j)   # moments: copied or zero, exactly


def test_refinement_is_deterministic_across_launches_and_keys(hip_lib):
    """The same (seed, step) gives bit-identical tensors on every call -- what keeps the ranks of a data-parallel job in
    lockstep without a broadcast; another step or seed gives other samples."""
    from mtgs_amd.densify import RefineConfig, refine_gaussians
    cfg = RefineConfig()
    p, stats, _ = _refine_case(5000, seed=3)
    dev = torch.device("cuda")
    pd, sd = {k: v.to(dev) for k, v in p.items()}, tuple(s.to(dev) for s in stats)
    a, _, ia = refine_gaussians(pd, sd, cfg, 4000, 99)
    b, _, ib = refine_gaussians({k: v.clone() for k, v in pd.items()}, sd, cfg, 4000, 99)
    assert ia["n_after"] == ib["n_after"] and all(torch.equal(a[k], b[k]) for k in a)
    c, _, _ = refine_gaussians(pd, sd, cfg, 4100, 99)
    d, _, _ = refine_gaussians(pd, sd, cfg, 4000, 100)
    assert c["means"].shape == a["means"].shape and not torch.equal(c["means"], a["means"])
    assert not torch.equal(d["means"], a["means"])
    # empty node and a node where nothing changes
    e, _, ie = refine_gaussians({k: v[:0] for k, v in pd.items()}, tuple(s[:0] for s in sd), cfg, 4000, 1)
    assert ie["n_after"] == 0 and e["means"].shape == (0, 3)


def test_statistics_from_the_exchange_rows_equal_the_dense_update(hip_lib):
    """In data-parallel mode the backward leaves compact gradient rows instead of a dense means2d gradient
    (SparseGradExchange.rasterization): update_statistics_rows on them = update_statistics_all on the dense absgrad of the
    same frame rendered the ordinary way."""
    from mtgs_amd import dist as mdist, rasterization, spherical_harmonics
    from mtgs_amd.densify import update_statistics_all, update_statistics_rows
    from mtgs_amd.synthetic import make_camera, make_scene
    dev = torch.device("cuda")
    N, W, H = 40_000, 320, 240
    sc = make_scene(N, seed=3, sh_degree=3, extent=(12.0, 4.0, 12.0))
    vm, K = make_camera(W, H, yaw_deg=20.0)
    vm, K = vm.to(dev), K.to(dev)
    cam_pos = torch.inverse(vm)[0, :3, 3]
    g = torch.Generator().manual_seed(1)
    Gc, Ga = torch.randn(1, H, W, 4, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)
    sizes = [25_000, 10_000, 5_000]                       # three nodes, concatenated as get_gaussians does
    mk = lambda: [(torch.zeros(n, device=dev), torch.ones(n, device=dev), torch.zeros(n, device=dev)) for n in sizes]

    P = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    sh = spherical_harmonics(3, P["means"].detach() - cam_pos, P["coeffs"].detach())
    render, alpha, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], torch.clamp(sh + 0.5, 0.0, 1.0), vm, K,
                                        W, H, packed=False, render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
    info["means2d"].retain_grad()
    torch.autograd.backward([render, alpha], [Gc, Ga])
    dense = mk()
    update_statistics_all(dense, info["radii"], info["means2d"].absgrad, W, H)

    P2 = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    ex = mdist.SparseGradExchange(N, 16, dev)
    r2, a2, info2 = ex.rasterization(P2["means"], P2["quats"], P2["scales"], P2["opacities"], sh, vm, K, W, H, cam_pos)
    torch.autograd.backward([r2, a2], [Gc, Ga])
    rows = mk()
    n_vis = int((info2["radii"] > 0).sum())
    assert ex.grad_rows is not None and ex.vis_ids.numel() >= n_vis
    update_statistics_rows(rows, info2["radii"], ex.grad_rows, ex.vis_ids, W, H, n_vis=n_vis)
    ex.finish(P2["means"], 3)
    for (a_n, a_c, a_m), (b_n, b_c, b_m) in zip(dense, rows):
        assert torch.equal(a_c, b_c) and torch.equal(a_m, b_m)
        assert float(a_n.max()) > 0 and torch.allclose(a_n, b_n, rtol=2e-4, atol=1e-6 * float(a_n.max()))   # fp32 atomics in another order
