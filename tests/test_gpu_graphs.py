"""mtgs_amd.graphs.GraphedIteration -- the capture / replay / overflow / re-capture manager of graph training -- on a small
two-camera fitting problem: rasterization() -> L1 -> backward -> FusedAdam.step, the iteration MTGS runs per step in miniature
(/root/reference/mtgs/scene_model/mtgs_scene_graph.py:547-708; refinement_after: vanilla_gaussian_splatting.py:448-577).
The graph-trained run must be the SAME training as the eager loop (up to the order of the fp32 atomics), through: the eager
frames that teach the size plan, warm-up + capture per key, replays, a forced overflow (first capacities too small) with its
non-blocking detection and re-capture, and a "refinement" (new parameter tensors, N changed) with capture-without-warm-up."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(dev, N=30_000, W=320, H=208):
    from mtgs_amd import rasterization
    from mtgs_amd.synthetic import make_camera, make_scene
    sc = make_scene(N, seed=21, sh_degree=None, extent=(8.0, 3.0, 8.0))
    cams = [tuple(t.to(dev) for t in make_camera(W, H, yaw_deg=12.0 * c)) for c in range(2)]
    true = {k: v.to(dev) for k, v in sc.items()}
    with torch.no_grad():
        targets = [rasterization(true["means"], true["quats"], true["scales"], true["opacities"], true["colors"], vm, K, W, H, packed=False)[0]
                   for vm, K in cams]
    g = torch.Generator().manual_seed(4)
    start = {"means": true["means"] + 0.02 * torch.randn(N, 3, generator=g).to(dev),
             "log_scales": true["scales"].log() + 0.1 * torch.randn(N, 3, generator=g).to(dev),
             "quats": true["quats"].clone(),
             "logit_opac": torch.logit(true["opacities"].clamp(1e-4, 1 - 1e-4)) + 0.3 * torch.randn(N, generator=g).to(dev),
             "colors": (true["colors"] + 0.2 * torch.randn(N, 3, generator=g).to(dev)).clamp(0, 1)}
    return cams, targets, start, (W, H)


def _train(dev, steps, graph, first_cap_scale=1.0, refine_at=None, counts_out=None, N=30_000):
    import mtgs_amd
    from mtgs_amd import rasterization
    from mtgs_amd.graphs import GraphedIteration
    from mtgs_amd.optim import FusedAdam
    cams, targets, start, (W, H) = _problem(dev, N=N)
    P = {k: v.clone().requires_grad_(True) for k, v in start.items()}
    lrs = {"means": 2e-3, "log_scales": 5e-3, "quats": 1e-3, "logit_opac": 2e-2, "colors": 1e-2}
    mk_opt = lambda: FusedAdam([{"params": [P[k]], "lr": lrs[k], "name": k} for k in P], eps=1e-15)
    box = {"opt": mk_opt()}
    # (one graph per key: the camera / target of a key are host-selected tensors.  The body keeps NO reference to the step's
    #  tensors -- see GraphedIteration's docstring)

    def body(c):
        opt = box["opt"]
        opt.zero_grad(set_to_none=True)
        vm, K = cams[c]
        r, a, info = rasterization(P["means"], P["quats"], P["log_scales"].exp(), torch.sigmoid(P["logit_opac"]), P["colors"], vm, K, W, H,
                                   packed=False, absgrad=True)
        loss = (r - targets[c]).abs().mean()
        loss.backward()
        opt.step()
        return loss, info

    hist = torch.zeros(steps, device=dev)
    GI = None
    if graph:
        GI = GraphedIteration(body, n_keys=2, size_key=lambda: (1, P["means"].shape[0], W, H), device=dev,
                              before_replay=lambda: box["opt"].advance(), can_skip_warmup=lambda: True, poll_every=4,
                              first_cap_scale=first_cap_scale, log=lambda *a: None)
    for i in range(steps):
        if graph:
            hist[i].copy_(GI.step(i % 2))
            GI.poll(i)
        else:
            with mtgs_amd.tight_lists():      # (the mode the graph run composites in: identical pixels either way)
                hist[i].copy_(body(i % 2)[0])
        if refine_at is not None and i + 1 == refine_at:
            # a stand-in for refinement_after: every tensor replaced (a tenth of the Gaussians culled), the moments moved
            n_before = P["means"].shape[0]
            idx = torch.arange(n_before, device=dev)
            keep = idx[idx % 10 != 0]          # (a fixed tenth: the two runs must cull the same Gaussians)
            old = box["opt"]
            state = {k: old.state[P[k]] for k in P}
            for k in list(P):
                P[k] = P[k].detach()[keep].clone().requires_grad_(True)
            box["opt"] = mk_opt()
            for k in P:
                box["opt"].state[P[k]] = {"step": state[k]["step"], "exp_avg": state[k]["exp_avg"][keep].clone(),
                                          "exp_avg_sq": state[k]["exp_avg_sq"][keep].clone()}
            box["opt"].inherit_layout(old)
            if graph:
                GI.after_refinement(n_before, P["means"].shape[0])
    torch.cuda.synchronize()
    if counts_out is not None and GI is not None:
        counts_out.update(GI.counts, overflowed=GI.overflowed())
    if GI is not None:
        GI.close()
    return hist.tolist(), {k: v.detach().clone() for k, v in P.items()}


def _same_training(a, b, pa, pb):
    la, lb = torch.tensor(a), torch.tensor(b)
    assert la[-1] < 0.8 * la[0], (la[0], la[-1])                           # it trains
    assert (la - lb).abs().max() <= 2e-3 * la.abs().max(), (la - lb).abs().max()      # the same curve, step for step
    # ... and the same parameters.  (Element by element only statistically: the two runs sum the fp32 atomics in another order, and
    # where a gradient is noise around zero Adam's normalised step turns its SIGN into lr per step.)
    for k in pa:
        assert pa[k].shape == pb[k].shape
        d, scale = (pa[k] - pb[k]).abs().flatten().float(), float(pa[k].abs().max())
        assert float(d.mean()) <= 2e-4 * scale and float(torch.quantile(d[:1_000_000], 0.999)) <= 1e-2 * scale, (k, float(d.mean()), float(d.max()))


def test_graph_training_equals_eager(hip_lib):
    dev = torch.device("cuda")
    eager = _train(dev, 40, graph=False)
    counts = {}
    graphed = _train(dev, 40, graph=True, counts_out=counts)
    _same_training(*eager[:1], *graphed[:1], eager[1], graphed[1])
    assert counts == {"captures": 2, "warmups": 2, "overflows": 0, "eager": 2, "replays": 36, "overflowed": False}, counts


def test_overflow_is_found_without_blocking_and_recaptured(hip_lib):
    """First capacities far below what the frames need (the constant floor only: 4096 visible Gaussians, 65536 intersections): the
    graph frames are truncated (never out of bounds), a poll finds the flag, the graphs are dropped, the keys render eagerly again
    (exact sizes) and are re-captured with sufficient capacities."""
    dev = torch.device("cuda")
    counts = {}
    hist, _ = _train(dev, 40, graph=True, first_cap_scale=0.0, counts_out=counts, N=120_000)
    assert counts["overflows"] == 1 and counts["captures"] == 4 and counts["eager"] == 4 and not counts["overflowed"], counts
    eager = _train(dev, 40, graph=False, N=120_000)[0]
    # the few steps in front of the detection trained on heavily truncated frames (4096 of ~20k visible Gaussians); from there on
    # it is the same optimisation, which recovers: the loss ends near the eager run's
    assert hist[-1] < 0.3 * hist[0] and abs(hist[-1] - eager[-1]) <= 0.5 * eager[-1], (hist[-1], eager[-1])


def test_refinement_recaptures_without_warmup(hip_lib):
    dev = torch.device("cuda")
    eager = _train(dev, 40, graph=False, refine_at=20)
    counts = {}
    graphed = _train(dev, 40, graph=True, refine_at=20, counts_out=counts)
    _same_training(eager[0], graphed[0], eager[1], graphed[1])
    assert eager[1]["means"].shape[0] == 27_000
    # second stretch: both keys captured again, no warm-up pass, no eager frame (capacities scaled from the device counts)
    assert counts["captures"] == 4 and counts["warmups"] == 2 and counts["eager"] == 2 and counts["overflows"] == 0, counts
