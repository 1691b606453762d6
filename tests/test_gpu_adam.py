"""Fused Adam (csrc/adam.hip, mtgs_amd.optim.FusedAdam) against torch.optim.Adam on the CPU in float64 -- the reference's
optimizer (one torch.optim.Adam per parameter group: custom_trainer.py:115-136, config/MTGS.py:121-181) -- over several steps
including a refinement (rows removed and appended with the moments following, vanilla_gaussian_splatting.py:392-446)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# the groups of config/MTGS.py:121-181 (name, shape per Gaussian, lr); eps = 1e-15 everywhere
MTGS_GROUPS = [("means", (3,), 8e-4), ("features_dc", (3,), 0.0025), ("features_rest", (15, 3), 0.0025 / 20),
               ("opacities", (1,), 0.05), ("scales", (3,), 0.005), ("quats", (4,), 0.001)]


def _params(N, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return {k: (torch.randn(N, *shp, generator=g) * (30.0 if k == "means" else 0.3)) for k, shp, _ in MTGS_GROUPS}


def _oracle_opt(P64):
    return [torch.optim.Adam([P64[k]], lr=lr, eps=1e-15) for k, _, lr in MTGS_GROUPS]


def _close(got, ref, what, lr=0.0):
    """<= 1e-6 relative on p -- of max(|p|, lr): an entry that is smaller than one step cannot be held to 1e-6 of ITSELF in
    fp32, the step that produced it carries a relative rounding of ~1e-7."""
    got, ref = got.detach().cpu().double(), ref.detach().double()
    err = (got - ref).abs()
    bound = 1e-6 * torch.clamp(ref.abs(), min=lr) + 1e-12
    assert bool((err <= bound).all()), f"{what}: max err {float(err.max()):.3e}, worst ratio {float((err / bound).max()):.2f}"


@pytest.mark.parametrize("N", [1, 1023, 50_000])
def test_fused_adam_matches_torch_adam_fp64_over_steps_and_a_refinement(hip_lib, N):
    from mtgs_amd.optim import FusedAdam
    dev = torch.device("cuda")
    P0 = _params(N, 0, dev)
    P = {k: v.clone().to(dev).requires_grad_(True) for k, v in P0.items()}
    P64 = {k: v.clone().double().requires_grad_(True) for k, v in P0.items()}
    opt = FusedAdam([{"params": [P[k]], "lr": lr} for k, _, lr in MTGS_GROUPS], eps=1e-15)
    refs = _oracle_opt(P64)
    g = torch.Generator().manual_seed(1)

    def one_step(P, P64, opt, refs, zero_rows=None):
        for k in P:
            gr = torch.randn(P[k].shape, generator=g) * 0.01
            if zero_rows is not None:          # Gaussians outside the frame: exactly-zero gradient rows
                gr[zero_rows] = 0.0
            P[k].grad = gr.to(dev)
            P64[k].grad = gr.double()
        opt.step()
        for r in refs:
            r.step()

    zero_rows = torch.rand(N, generator=g) < 0.8
    for s in range(3):
        one_step(P, P64, opt, refs, zero_rows)
        for k in P:
            lr = dict((n, l) for n, _, l in MTGS_GROUPS)[k]
            _close(P[k], P64[k], f"step {s + 1} {k}", lr)
            _close(opt.state[P[k]]["exp_avg"], refs[[n for n, _, _ in MTGS_GROUPS].index(k)].state[P64[k]]["exp_avg"], f"step {s + 1} m {k}", 1e-3)
            assert float(opt.state[P[k]]["step"]) == s + 1 == float(refs[0].state[P64["means"]]["step"])
    # ---- refinement: cull a third of the rows, append 7 new ones (zero moments), as remove_from_optim / dup_in_optim do
    keep = torch.ones(N, dtype=torch.bool)
    keep[::3] = N < 3
    n_new = 7
    newP, newP64, new_refs = {}, {}, []
    params = []
    for k, shp, lr in MTGS_GROUPS:
        add = torch.randn(n_new, *shp, generator=g)
        st, st64 = opt.state[P[k]], refs[len(new_refs)].state[P64[k]]
        q = torch.cat([P[k].detach()[keep.to(dev)], add.to(dev)]).requires_grad_(True)
        q64 = torch.cat([P64[k].detach()[keep], add.double()]).requires_grad_(True)
        z = lambda t, a: torch.cat([t[keep.to(t.device)], torch.zeros_like(a, dtype=t.dtype, device=t.device)])
        newP[k], newP64[k] = q, q64
        r = torch.optim.Adam([q64], lr=lr, eps=1e-15)
        r.state[q64] = {"step": st64["step"], "exp_avg": z(st64["exp_avg"], add), "exp_avg_sq": z(st64["exp_avg_sq"], add)}
        new_refs.append(r)
        params.append({"params": [q], "lr": lr})
    opt2 = FusedAdam(params, eps=1e-15)
    for k in newP:
        st = opt.state[P[k]]
        opt2.state[newP[k]] = {"step": st["step"], "exp_avg": z(st["exp_avg"], torch.zeros(n_new, *P[k].shape[1:])),
                               "exp_avg_sq": z(st["exp_avg_sq"], torch.zeros(n_new, *P[k].shape[1:]))}
    for s in range(2):
        one_step(newP, newP64, opt2, new_refs)
        for k in newP:
            _close(newP[k], newP64[k], f"after refinement, step {s + 1} {k}", dict((n, l) for n, _, l in MTGS_GROUPS)[k])
            assert float(opt2.state[newP[k]]["step"]) == 3 + s + 1


def test_fused_adam_row_gradients_equal_dense_gradients(hip_lib):
    """Gradient source 2: compact rows of the visible Gaussians + a row map.  Must equal the dense-gradient step with the
    rows scattered into zeros (culled Gaussians: exact zero-gradient update), bit for bit, including a tensor whose rows
    straddle the 16-byte vectors (width 45) and one that is not 16-byte aligned."""
    from mtgs_amd.optim import FusedAdam
    dev = torch.device("cuda")
    N = 20_011
    g = torch.Generator().manual_seed(3)
    vis = torch.rand(N, generator=g) < 0.15
    n_vis = int(vis.sum())
    row_of = torch.full((N,), -1, dtype=torch.int32)
    row_of[vis] = torch.arange(n_vis, dtype=torch.int32)
    STRIDE = 64
    rows = torch.randn(n_vis, STRIDE, generator=g)
    layout = [("means", (3,), 0), ("quats", (4,), 3), ("scales", (3,), 7), ("opacities", (), 10), ("rest", (15, 3), 16)]
    base = {k: torch.randn(N, *shp, generator=g) for k, shp, _ in layout}
    flat = torch.zeros(N * 3 + 1)                      # an unaligned tensor: a view one float into a buffer
    flat[1:] = base["means"].reshape(-1)

    def make():
        P = {k: v.clone().to(dev).requires_grad_(True) for k, v in base.items()}
        holder = flat.clone().to(dev)
        P["means"] = holder[1:].view(N, 3).requires_grad_(True)
        return P, FusedAdam([{"params": [p], "lr": 1e-2 * (i + 1)} for i, p in enumerate(P.values())], eps=1e-15)

    Pa, oa = make()
    Pb, ob = make()
    rows_d, row_of_d = rows.to(dev), row_of.to(dev)
    for step in range(3):
        for k, shp, col in layout:
            width = int(np.prod(shp)) if shp else 1
            dense = torch.zeros(N, width)
            dense[vis] = rows[:, col:col + width]
            Pa[k].grad = dense.view(N, *shp).to(dev)
            ob.set_row_gradient(Pb[k], rows_d, row_of_d, col)
        oa.step()
        ob.step()
        for k in Pa:
            assert torch.equal(Pa[k], Pb[k]), (step, k)
            assert torch.equal(oa.state[Pa[k]]["exp_avg_sq"], ob.state[Pb[k]]["exp_avg_sq"]), (step, k)
        rows_d = rows_d * 0.5
        rows = rows * 0.5


def test_fused_adam_adds_a_dense_gradient_to_the_row_gradient(hip_lib):
    """Both gradient sources on one parameter: `p.grad` (a loss term that reaches the parameter outside the rasterization:
    MTGS's scale regularisers, mtgs_scene_graph.py:936-981) AND set_row_gradient() (the rasterization's own gradient) -- the step
    uses their sum, bit-identical to one dense gradient holding `dense + scattered rows` (round 3 ignored p.grad silently);
    widths 3 / 4 / 45 / 1 and an unaligned tensor; a row-lazy parameter refuses the combination."""
    from mtgs_amd.optim import FusedAdam
    dev = torch.device("cuda")
    N = 9_973
    g = torch.Generator().manual_seed(8)
    vis = torch.rand(N, generator=g) < 0.2
    n_vis = int(vis.sum())
    row_of = torch.full((N,), -1, dtype=torch.int32)
    row_of[vis] = torch.arange(n_vis, dtype=torch.int32)
    rows = torch.randn(n_vis, 64, generator=g)
    layout = [("scales", (3,), 0), ("quats", (4,), 3), ("opacities", (), 7), ("rest", (15, 3), 8)]
    base = {k: torch.randn(N, *shp, generator=g) for k, shp, _ in layout}
    reg = {k: 0.3 * torch.randn(N, *shp, generator=g) for k, shp, _ in layout}

    def make():
        P = {k: v.clone().to(dev).requires_grad_(True) for k, v in base.items()}
        return P, FusedAdam([{"params": [p], "lr": 1e-2 * (i + 1)} for i, p in enumerate(P.values())], eps=1e-15)

    Pa, oa = make()
    Pb, ob = make()
    rows_d, row_of_d = rows.to(dev), row_of.to(dev)
    for step in range(3):
        for k, shp, col in layout:
            width = int(np.prod(shp)) if shp else 1
            scat = torch.zeros(N, width)
            scat[vis] = rows[:, col:col + width]
            Pa[k].grad = (reg[k].view(N, width) + scat).view(N, *shp).to(dev)           # dense + rows, added in fp32 on the host
            Pb[k].grad = reg[k].clone().to(dev)
            ob.set_row_gradient(Pb[k], rows_d, row_of_d, col)
        oa.step()
        ob.step()
        for k in Pa:
            assert torch.equal(Pa[k], Pb[k]), (step, k)
            assert torch.equal(oa.state[Pa[k]]["exp_avg"], ob.state[Pb[k]]["exp_avg"]), (step, k)
    p = base["rest"].clone().to(dev).requires_grad_(True)
    ol = FusedAdam([{"params": [p], "lr": 1e-2}], eps=1e-15)
    ol.set_row_lazy(p)
    p.grad = torch.zeros_like(p)
    ol.set_row_gradient(p, rows_d, row_of_d, 8)
    with pytest.raises(RuntimeError, match="row-lazy"):
        ol.step()


def test_fused_adam_state_dict_round_trip_with_torch_adam(hip_lib):
    """state_dict() is torch.optim.Adam's: a run can switch optimizers in either direction (checkpoints of the reference
    keep `optimizers` per group, custom_trainer.py:148-157)."""
    from mtgs_amd.optim import FusedAdam
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    w0 = torch.randn(1000, 3, generator=g)
    a = w0.clone().to(dev).requires_grad_(True)
    b = w0.clone().to(dev).requires_grad_(True)
    oa, ob = FusedAdam([a], lr=1e-2, eps=1e-15), torch.optim.Adam([b], lr=1e-2, eps=1e-15)
    for _ in range(2):
        gr = torch.randn(1000, 3, generator=g).to(dev)
        a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    ob2 = torch.optim.Adam([b], lr=1e-2, eps=1e-15)
    ob2.load_state_dict(oa.state_dict())          # fused -> torch
    oa2 = FusedAdam([a], lr=1e-2, eps=1e-15)
    oa2.load_state_dict(ob.state_dict())          # torch -> fused
    gr = torch.randn(1000, 3, generator=g).to(dev)
    a.grad, b.grad = gr.clone(), gr.clone()
    oa2.step(); ob2.step()
    assert float(oa2.state[a]["step"]) == float(ob2.state[b]["step"]) == 3
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)


def test_fused_adam_in_a_hip_graph(hip_lib):
    """step() captured once; advance() + replay per step: the same trajectory as eager steps, with a learning-rate
    schedule changing lr between replays (means: ExponentialDecay in config/MTGS.py:123-128)."""
    from mtgs_amd.optim import FusedAdam
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(7)
    w0 = torch.randn(5000, 4, generator=g)
    grads = [torch.randn(5000, 4, generator=g).to(dev) for _ in range(5)]
    lrs = [1e-2 * 0.9 ** i for i in range(5)]
    a = w0.clone().to(dev).requires_grad_(True)
    oa = FusedAdam([a], lr=lrs[0], eps=1e-15)
    for i in range(5):
        oa.param_groups[0]["lr"] = lrs[i]
        a.grad = grads[i].clone()
        oa.step()
    b = w0.clone().to(dev).requires_grad_(True)
    ob = FusedAdam([b], lr=lrs[0], eps=1e-15)
    static_g = grads[0].clone()
    b.grad = static_g
    ob.step()                                      # eager step 1 (creates state and buffers)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ob.step()
    for i in range(1, 5):
        ob.param_groups[0]["lr"] = lrs[i]
        static_g.copy_(grads[i])
        ob.advance()
        graph.replay()
    torch.cuda.synchronize()
    assert float(ob.state[b]["step"]) == 5
    assert torch.equal(a, b)


def test_fused_adam_refuses_what_it_does_not_implement(hip_lib):
    from mtgs_amd.optim import FusedAdam
    p = torch.zeros(4, device="cuda", requires_grad=True)
    with pytest.raises(NotImplementedError):
        FusedAdam([p], amsgrad=True)
    q = torch.zeros(4, requires_grad=True)
    q.grad = torch.zeros(4)
    with pytest.raises(RuntimeError):
        FusedAdam([q]).step()


def test_exact_lazy_adam_for_per_traversal_tensors(hip_lib):
    """set_lazy_slices: a step updates only the rendered traversal's slice of `[N, T, ...]` tensors; prepare(t) catches slice t
    up with the zero-gradient steps it missed.  Over a random sequence of traversals with a learning-rate schedule, after
    flush() parameters AND moments are BIT-IDENTICAL to the optimizer that touches every slice at every step; a slice is
    current after prepare(t); a step on a slice that is behind is refused."""
    from mtgs_amd.optim import FusedAdam
    dev = torch.device("cuda")
    N, T = 3001, 4
    g = torch.Generator().manual_seed(9)
    base = {"rest": torch.randn(N, T, 15, 3, generator=g) * 0.2, "adapters": torch.randn(N, T, 3, generator=g) * 0.1,
            "dc": torch.randn(N, 3, generator=g)}

    def make(lazy):
        P = {k: v.clone().to(dev).requires_grad_(True) for k, v in base.items()}
        opt = FusedAdam([{"params": [P["rest"]], "lr": 1e-2}, {"params": [P["adapters"], P["dc"]], "lr": 3e-3}], eps=1e-15)
        if lazy:
            opt.set_lazy_slices(P["rest"])
            opt.set_lazy_slices(P["adapters"])
        return P, opt

    Pa, oa = make(False)
    Pb, ob = make(True)
    seq = [0, 2, 2, 1, 0, 3, 3, 3, 1, 2, 0, 0, 1]
    for step, t in enumerate(seq):
        vis = torch.rand(N, generator=g) < 0.2
        n_vis = int(vis.sum())
        row_of = torch.full((N,), -1, dtype=torch.int32)
        row_of[vis] = torch.arange(n_vis, dtype=torch.int32)
        rows = (torch.randn(n_vis, 48, generator=g) * 0.01).to(dev)
        row_of = row_of.to(dev)
        for o in (oa, ob):
            o.param_groups[0]["lr"] = 1e-2 * 0.95 ** step
        ob.prepare(t)
        # after prepare(t) slice t equals the always-stepping optimizer's
        assert torch.equal(Pa["rest"][:, t], Pb["rest"][:, t]) and torch.equal(Pa["adapters"][:, t], Pb["adapters"][:, t]), step
        for P, o in ((Pa, oa), (Pb, ob)):
            o.set_row_gradient(P["dc"], rows, row_of, 0)
            o.set_row_gradient(P["adapters"], rows, row_of, 0, slice_index=t)
            o.set_row_gradient(P["rest"], rows, row_of, 3, slice_index=t)
            o.step()
    behind = [tt for tt in range(T) if tt != seq[-1]]
    assert any(not torch.equal(Pa["rest"][:, tt], Pb["rest"][:, tt]) for tt in behind)      # (really lazy)
    ob.flush()
    for k in base:
        assert torch.equal(Pa[k], Pb[k]), k
        assert torch.equal(oa.state[Pa[k]]["exp_avg"], ob.state[Pb[k]]["exp_avg"]), k
        assert torch.equal(oa.state[Pa[k]]["exp_avg_sq"], ob.state[Pb[k]]["exp_avg_sq"]), k
    # a step on a slice that is behind is refused
    t_bad = 1
    ob.set_row_gradient(Pb["rest"], rows, row_of, 3, slice_index=t_bad)
    ob.set_row_gradient(Pb["adapters"], rows, row_of, 0, slice_index=t_bad)
    ob.set_row_gradient(Pb["dc"], rows, row_of, 0)
    ob.step()                                   # all slices are current right after flush(): fine
    ob.set_row_gradient(Pb["rest"], rows, row_of, 3, slice_index=2)
    ob.set_row_gradient(Pb["adapters"], rows, row_of, 0, slice_index=2)
    ob.set_row_gradient(Pb["dc"], rows, row_of, 0)
    with pytest.raises(RuntimeError):
        ob.step()                               # slice 2 missed the previous step and was not prepared


def _row_lazy_case(dev, N, T, g):
    base = {"rest": torch.randn(N, T, 15, 3, generator=g) * 0.2, "adapters": torch.randn(N, T, 3, generator=g) * 0.1,
            "dc": torch.randn(N, 3, generator=g), "means": torch.randn(N, 3, generator=g)}

    def make(lazy, **kw):
        from mtgs_amd.optim import FusedAdam
        P = {k: v.clone().to(dev).requires_grad_(True) for k, v in base.items()}
        opt = FusedAdam([{"params": [P["rest"]], "lr": 1e-2}, {"params": [P["adapters"], P["dc"]], "lr": 3e-3},
                         {"params": [P["means"]], "lr": 1e-4}], eps=1e-15)
        if lazy:
            opt.set_row_lazy(P["rest"], traversals=T, **kw)
            opt.set_row_lazy(P["adapters"], traversals=T, **kw)
            opt.set_row_lazy(P["dc"], **kw)
        return P, opt
    return base, make


def _frame(N, g, dev, frac=0.2):
    vis = torch.rand(N, generator=g) < frac
    n_vis = int(vis.sum())
    row_of = torch.full((N,), -1, dtype=torch.int32)
    row_of[vis] = torch.arange(n_vis, dtype=torch.int32)
    rows = (torch.randn(max(n_vis, 1), 48, generator=g) * 0.01).to(dev)
    return vis.to(dev), row_of.to(dev), rows


def test_exact_row_lazy_adam_touches_only_the_visible_rows(hip_lib):
    """set_row_lazy: step() updates only the rows the frame saw (of the rendered traversal's slice), catch_up_rows() applies
    the zero-gradient steps a row missed right before it is read.  Random traversal order, random visibility, a learning-rate
    schedule, a history that has to grow: (1) after catch_up_rows the visible rows are BIT-IDENTICAL to the optimizer that
    steps every row every time, (2) the lazy optimizer really left the other rows alone, (3) after flush() parameters AND
    moments of every row are bit-identical, (4) state_dict() flushes."""
    dev = torch.device("cuda")
    N, T = 3001, 4
    g = torch.Generator().manual_seed(11)
    base, make = _row_lazy_case(dev, N, T, g)
    Pa, oa = make(False)
    Pb, ob = make(True, hist_capacity=4)
    seq = [0, 2, 2, 1, 0, 3, 3, 3, 1, 2, 0, 0, 1, 3, 2, 1, 1, 0]
    for step, t in enumerate(seq):
        vis, row_of, rows = _frame(N, g, dev, frac=0.05 if step % 5 == 4 else 0.25)
        for o in (oa, ob):
            o.param_groups[0]["lr"] = 1e-2 * 0.95 ** step
            o.param_groups[1]["lr"] = 3e-3 * (1.0 + 0.1 * (step % 3))
        ob.catch_up_rows([(Pb["dc"], row_of, None), (Pb["adapters"], row_of, t), (Pb["rest"], row_of, t)])
        assert torch.equal(Pa["dc"][vis], Pb["dc"][vis]), step
        assert torch.equal(Pa["rest"][vis, t], Pb["rest"][vis, t]) and torch.equal(Pa["adapters"][vis, t], Pb["adapters"][vis, t]), step
        grad_means = (torch.randn(N, 3, generator=g) * 0.1).to(dev)
        for P, o in ((Pa, oa), (Pb, ob)):
            P["means"].grad = grad_means.clone()
            o.set_row_gradient(P["dc"], rows, row_of, 0)
            o.set_row_gradient(P["adapters"], rows, row_of, 0, slice_index=t)
            o.set_row_gradient(P["rest"], rows, row_of, 3, slice_index=t)
            o.step()
        assert torch.equal(Pa["means"], Pb["means"])
        assert torch.equal(Pa["rest"][vis, t], Pb["rest"][vis, t]) and torch.equal(Pa["dc"][vis], Pb["dc"][vis]), step
    assert not torch.equal(Pa["rest"], Pb["rest"]) and not torch.equal(Pa["dc"], Pb["dc"])           # (really lazy)
    sd = ob.state_dict()                                                                             # flushes
    for k in base:
        assert torch.equal(Pa[k], Pb[k]), k
        assert torch.equal(oa.state[Pa[k]]["exp_avg"], ob.state[Pb[k]]["exp_avg"]), k
        assert torch.equal(oa.state[Pa[k]]["exp_avg_sq"], ob.state[Pb[k]]["exp_avg_sq"]), k
    # ... and goes on from a loaded state: every row current as of the loaded step count
    Pc, oc = make(True)
    with torch.no_grad():
        for k in base:
            Pc[k].copy_(Pb[k])
    oc.load_state_dict(sd)
    vis, row_of, rows = _frame(N, g, dev)
    for P, o in ((Pa, oa), (Pc, oc)):
        o.catch_up_rows([(P["dc"], row_of, None), (P["adapters"], row_of, 1), (P["rest"], row_of, 1)])
        P["means"].grad = torch.zeros(N, 3, device=dev)
        o.set_row_gradient(P["dc"], rows, row_of, 0)
        o.set_row_gradient(P["adapters"], rows, row_of, 0, slice_index=1)
        o.set_row_gradient(P["rest"], rows, row_of, 3, slice_index=1)
        o.step()
    oc.flush()
    for k in base:
        assert torch.equal(Pa[k], Pc[k]), k
    # a row-lazy parameter with a dense gradient is refused
    Pb["dc"].grad = torch.zeros_like(Pb["dc"])
    with pytest.raises(RuntimeError):
        ob.step()


def test_row_lazy_step_leaves_zero_gradient_rows_lazy(hip_lib):
    """set_row_gradient(zero_probe=c): a visible row whose gradient is exactly zero (an occluded Gaussian: most of the frustum-
    visible ones get no gradient at all) is NOT stepped -- it stays lazy like a Gaussian the frame did not see.  Eighty percent of
    the visible rows of every frame are given all-zero gradient rows: (1) those rows are really left alone by the step, (2) after
    flush() every parameter and moment is bit-identical to the optimizer that steps every row with the same (mostly zero)
    gradients; SCAN and LIST forms."""
    dev = torch.device("cuda")
    N, T = 3001, 3
    g = torch.Generator().manual_seed(23)
    base, make = _row_lazy_case(dev, N, T, g)
    for use_list in (False, True):
        Pa, oa = make(False)
        Pb, ob = make(True)
        skipped_any = False
        for step, t in enumerate([0, 1, 1, 2, 0, 2, 1, 0, 0, 2]):
            vis, row_of, rows = _frame(N, g, dev, frac=0.3)
            dead = (torch.rand(rows.shape[0], generator=g) < 0.8).to(dev)
            rows = rows.clone()
            rows[dead] = 0.0                                           # occluded: no gradient at all
            ids = torch.nonzero(vis).flatten().int().contiguous()
            kw = {"row_ids": (ids, 0, None)} if use_list else {}
            before = Pb["rest"].detach().clone()
            ob.catch_up_rows([(Pb["dc"], row_of, None), (Pb["adapters"], row_of, t), (Pb["rest"], row_of, t)])
            caught = Pb["rest"].detach().clone()
            for P, o, z in ((Pa, oa, {}), (Pb, ob, dict(zero_probe=0, **kw))):
                P["means"].grad = torch.zeros(N, 3, device=dev)
                o.set_row_gradient(P["dc"], rows, row_of, 0, **z)
                o.set_row_gradient(P["adapters"], rows, row_of, 0, slice_index=t, **z)
                o.set_row_gradient(P["rest"], rows, row_of, 3, slice_index=t, **z)
                o.step()
            dead_items = torch.nonzero(vis).flatten()[dead]
            live_items = torch.nonzero(vis).flatten()[~dead]
            assert torch.equal(Pb["rest"][dead_items, t], caught[dead_items, t])            # not stepped (only caught up before)
            if step > 0 and not torch.equal(Pa["rest"][dead_items, t], Pb["rest"][dead_items, t]):
                skipped_any = True                                                           # (the every-row optimizer moved them)
            assert torch.equal(Pa["rest"][live_items, t], Pb["rest"][live_items, t]), (use_list, step)
        assert skipped_any
        ob.flush()
        for k in base:
            assert torch.equal(Pa[k], Pb[k]), (use_list, k)
            assert torch.equal(oa.state[Pa[k]]["exp_avg"], ob.state[Pb[k]]["exp_avg"]), (use_list, k)
            assert torch.equal(oa.state[Pa[k]]["exp_avg_sq"], ob.state[Pb[k]]["exp_avg_sq"]), (use_list, k)


def test_row_lazy_adam_in_a_hip_graph(hip_lib):
    """catch_up_rows + step captured in ONE HIP graph (static row buffers, advance() per replay): after flush() bit-identical
    to the eager optimizer that steps every row -- the captured forward reads `t - 1` as the steps already taken."""
    dev = torch.device("cuda")
    N, T = 2000, 2
    g = torch.Generator().manual_seed(12)
    base, make = _row_lazy_case(dev, N, T, g)
    Pa, oa = make(False)
    Pb, ob = make(True)
    frames = [_frame(N, g, dev) for _ in range(9)]
    cap = max(f[2].shape[0] for f in frames)
    row_of_s = torch.full((N,), -1, dtype=torch.int32, device=dev)
    rows_s = torch.zeros(cap, 48, device=dev)
    gm_s = torch.zeros(N, 3, device=dev)
    Pb["means"].grad = gm_s

    def load(i):
        vis, row_of, rows = frames[i]
        row_of_s.copy_(row_of)
        rows_s.zero_()
        rows_s[:rows.shape[0]].copy_(rows)
        gm_s.copy_(torch.full((N, 3), 0.01 * (i + 1), device=dev))

    def body(o, P, ro, rw, t):
        o.catch_up_rows([(P["dc"], ro, None), (P["adapters"], ro, t), (P["rest"], ro, t)])
        o.set_row_gradient(P["dc"], rw, ro, 0)
        o.set_row_gradient(P["adapters"], rw, ro, 0, slice_index=t)
        o.set_row_gradient(P["rest"], rw, ro, 3, slice_index=t)
        o.step()

    def ref(i, t):
        vis, row_of, rows = frames[i]
        Pa["means"].grad = torch.full((N, 3), 0.01 * (i + 1), device=dev)
        body(oa, Pa, row_of, rows, t)

    import mtgs_amd
    for t in range(T):                        # plain eager steps first: optimizer state and device buffers exist from here on
        load(t)
        ref(t, t)
        body(ob, Pb, row_of_s, rows_s, t)
    graphs = []
    side = torch.cuda.Stream()
    for t in range(T):
        gm = mtgs_amd.graph_mode(1, 1)        # (the staging buffers of the tables live with the mode object)
        load(T + t)
        ref(T + t, t)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), gm:
            body(ob, Pb, row_of_s, rows_s, t)             # warm-up on the capture stream = a real step
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with gm, torch.cuda.graph(gr):
            body(ob, Pb, row_of_s, rows_s, t)             # (captured, not run)
        graphs.append((gr, gm))
    for i, t in [(4, 1), (5, 0), (6, 0), (7, 1), (8, 0), (3, 1)]:
        ref(i, t)
        load(i)
        ob.advance()
        graphs[t][0].replay()
    ob.flush()
    torch.cuda.synchronize()
    for k in base:
        assert torch.equal(Pa[k], Pb[k]), k
        assert torch.equal(oa.state[Pa[k]]["exp_avg_sq"], ob.state[Pb[k]]["exp_avg_sq"]), k


def test_row_lazy_one_graph_for_every_traversal_through_a_device_slice(hip_lib):
    """The slice of a per-traversal tensor as an int32 DEVICE word (mtgs_adam_group.sub_index_dev): peek_rows + step captured
    ONCE serve every traversal -- the word is rewritten in front of each replay.  Bit-identical (after flush) to the eager
    optimizer that steps every row with the host slice index; the peeked rows are the every-row optimizer's parameters."""
    dev = torch.device("cuda")
    N, T = 3001, 3
    g = torch.Generator().manual_seed(23)
    base, make = _row_lazy_case(dev, N, T, g)
    Pa, oa = make(False)
    Pb, ob = make(True)
    frames = [_frame(N, g, dev, frac=0.25) for _ in range(12)]
    cap = max(f[2].shape[0] for f in frames)
    row_of_s = torch.full((N,), -1, dtype=torch.int32, device=dev)
    rows_s = torch.zeros(cap, 48, device=dev)
    gm_s = torch.zeros(N, 3, device=dev)
    Pb["means"].grad = gm_s
    t_dev = torch.zeros((), dtype=torch.int32, device=dev)
    C = torch.zeros(cap, 56, device=dev)

    def load(i, t):
        vis, row_of, rows = frames[i]
        row_of_s.copy_(row_of)
        rows_s.zero_()
        rows_s[:rows.shape[0]].copy_(rows)
        gm_s.copy_(torch.full((N, 3), 0.01 * (i + 1), device=dev))
        t_dev.fill_(t)

    def body_dev():
        ob.peek_rows([(Pb["dc"], row_of_s, None, 0), (Pb["adapters"], row_of_s, t_dev, 3), (Pb["rest"], row_of_s, t_dev, 6)], C)
        ob.set_row_gradient(Pb["dc"], rows_s, row_of_s, 0, caught=(C, 0))
        ob.set_row_gradient(Pb["adapters"], rows_s, row_of_s, 0, slice_index=t_dev, caught=(C, 3))
        ob.set_row_gradient(Pb["rest"], rows_s, row_of_s, 3, slice_index=t_dev, caught=(C, 6))
        ob.step()

    def ref(i, t):
        vis, row_of, rows = frames[i]
        Pa["means"].grad = torch.full((N, 3), 0.01 * (i + 1), device=dev)
        oa.set_row_gradient(Pa["dc"], rows, row_of, 0)
        oa.set_row_gradient(Pa["adapters"], rows, row_of, 0, slice_index=t)
        oa.set_row_gradient(Pa["rest"], rows, row_of, 3, slice_index=t)
        oa.step()

    import mtgs_amd
    seq = [0, 1, 2, 2, 0, 1, 1, 1, 2, 0, 2, 0]
    gm = mtgs_amd.graph_mode(1, 1)
    for i in range(2):                        # eager steps with the device word: state and device scalars exist from here on
        load(i, seq[i])
        ref(i, seq[i])
        with gm:
            body_dev()
    gr = torch.cuda.CUDAGraph()
    with gm, torch.cuda.graph(gr):
        body_dev()
    for i in range(2, len(seq)):
        t = seq[i]
        vis, row_of, rows = frames[i]
        load(i, t)
        ob.advance()
        gr.replay()
        r = row_of[vis].long()
        assert torch.equal(C[r, 3:6], Pa["adapters"][vis, t]) and torch.equal(C[r, 6:51], Pa["rest"][vis, t].reshape(-1, 45)), i
        ref(i, t)
        assert torch.equal(Pa["rest"][vis, t], Pb["rest"][vis, t]), i
    ob.flush()
    torch.cuda.synchronize()
    for k in base:
        assert torch.equal(Pa[k], Pb[k]), k
        assert torch.equal(oa.state[Pa[k]]["exp_avg"], ob.state[Pb[k]]["exp_avg"]), k
        assert torch.equal(oa.state[Pa[k]]["exp_avg_sq"], ob.state[Pb[k]]["exp_avg_sq"]), k
    with pytest.raises(ValueError):
        ob.set_row_gradient(Pb["rest"], rows_s, row_of_s, 3, slice_index=torch.zeros((), dtype=torch.int64, device=dev))


def test_row_flags_leave_the_rows_nothing_is_composited_from_alone(hip_lib):
    """row_flags (the flags of mtgs_blend_touch_packed: the Gaussians a frame composites FROM): peek_rows() requests nothing of
    the rows with a zero flag and does not write their rows of the buffer; step() takes the parameter of a flagged row from the
    buffer and treats the others by the zero_probe rule (zero gradient: left lazy while fewer than 4 T steps behind, then committed
    from the parameter itself).  Over a random sequence with 15 % of the visible rows flagged: (1) the peeked rows of the FLAGGED
    Gaussians are the every-row optimizer's parameters, (2) unflagged rows of the peek buffer stay untouched, (3) after flush()
    parameters and moments are bit-identical, (4) the lazy optimizer really left unflagged rows behind."""
    dev = torch.device("cuda")
    N, T = 4001, 3
    g = torch.Generator().manual_seed(31)
    base, make = _row_lazy_case(dev, N, T, g)
    Pa, oa = make(False)
    Pb, ob = make(True, hist_capacity=8)
    seq = [0, 1, 2, 2, 0, 1, 1, 2, 0, 0, 1, 2, 2, 0, 1, 0]
    ids_all = torch.arange(N, dtype=torch.int32, device=dev)
    skipped_some = False
    for step, t in enumerate(seq):
        vis, row_of, rows = _frame(N, g, dev, frac=0.3)
        R = rows.shape[0]
        flags = (torch.rand(R, generator=g) < 0.15).to(torch.uint8).to(dev)
        rows = rows * flags[:, None].float()                     # an unflagged row has a zero gradient
        vis_ids = ids_all[vis].contiguous()                      # the frame's visible Gaussians in increasing order
        count = torch.tensor([R << 32], dtype=torch.int64, device=dev)
        rid = (vis_ids, 0, count)
        for o in (oa, ob):
            o.param_groups[0]["lr"] = 1e-2 * 0.93 ** step
        C = torch.full((R, 56), float("nan"), device=dev)
        ob.peek_rows([(Pb["dc"], row_of, None, 0, rid), (Pb["adapters"], row_of, t, 3, rid), (Pb["rest"], row_of, t, 6, rid)], C,
                     row_flags=flags)
        on = flags.bool()
        sel = vis_ids[on].long()
        assert torch.equal(C[on, 0:3], Pa["dc"][sel]) and torch.equal(C[on, 3:6], Pa["adapters"][sel, t]), step            # (1)
        assert torch.equal(C[on, 6:51], Pa["rest"][sel, t].reshape(-1, 45)), step
        assert torch.isnan(C[~on, :51]).all(), step                                                                        # (2)
        grad_means = (torch.randn(N, 3, generator=g) * 0.1).to(dev)
        Pa["means"].grad, Pb["means"].grad = grad_means.clone(), grad_means.clone()
        oa.set_row_gradient(Pa["dc"], rows, row_of, 0)
        oa.set_row_gradient(Pa["adapters"], rows, row_of, 0, slice_index=t)
        oa.set_row_gradient(Pa["rest"], rows, row_of, 3, slice_index=t)
        oa.step()
        kw = lambda col: {"caught": (C, col), "row_ids": rid, "row_flags": flags, "zero_probe": 0}
        ob.set_row_gradient(Pb["dc"], rows, row_of, 0, **kw(0))
        ob.set_row_gradient(Pb["adapters"], rows, row_of, 0, slice_index=t, **kw(3))
        ob.set_row_gradient(Pb["rest"], rows, row_of, 3, slice_index=t, **kw(6))
        ob.step()
        assert torch.equal(Pa["rest"][sel, t], Pb["rest"][sel, t]) and torch.equal(Pa["dc"][sel], Pb["dc"][sel]), step
        off = vis_ids[~on].long()
        skipped_some = skipped_some or not torch.equal(Pa["rest"][off, t], Pb["rest"][off, t])                             # (4)
    assert skipped_some
    ob.flush()
    for k in base:
        assert torch.equal(Pa[k], Pb[k]), k                                                                                # (3)
        assert torch.equal(oa.state[Pa[k]]["exp_avg"], ob.state[Pb[k]]["exp_avg"]), k
        assert torch.equal(oa.state[Pa[k]]["exp_avg_sq"], ob.state[Pb[k]]["exp_avg_sq"]), k


def test_row_lazy_adam_long_gaps_settle_without_changing_a_bit(hip_lib):
    """Rows unseen for thousands of steps: exp_avg reaches a fixed point of the zero-gradient recurrence and the catch-up
    switches to its one-multiplication step (csrc/adam.hip) -- the result must still be bit-identical to stepping every row
    every time, including across a learning-rate change that lifts the settled bound, rows that were never stepped, zeros
    and tiny parameters, and a history longer than the LDS window."""
    from mtgs_amd.optim import FusedAdam
    dev = torch.device("cuda")
    N = 300
    g = torch.Generator().manual_seed(13)
    base = torch.randn(N, 15, 3, generator=g) * 0.3
    base[:20] = 0.0
    base[20:40] *= 1e-24
    Pa = base.clone().to(dev).requires_grad_(True)
    Pb = base.clone().to(dev).requires_grad_(True)
    oa = FusedAdam([Pa], lr=2e-3, eps=1e-15)
    ob = FusedAdam([Pb], lr=2e-3, eps=1e-15)
    ob.set_row_lazy(Pb, hist_capacity=64)
    seen_at = {0: range(0, 200), 3: range(100, 300), 1200: range(0, 50), 2400: range(0, 300), 2401: range(250, 300), 2600: range(0, 300)}
    empty = torch.full((N,), -1, dtype=torch.int32, device=dev)
    zero_rows = torch.zeros(1, 48, device=dev)
    for step in range(2601):
        lr = 2e-3 * (0.999 ** step) * (50.0 if 1500 <= step < 1510 else 1.0)
        oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = lr
        if step in seen_at:
            idx = torch.tensor(list(seen_at[step]))
            row_of = torch.full((N,), -1, dtype=torch.int32)
            row_of[idx] = torch.arange(idx.numel(), dtype=torch.int32)
            row_of = row_of.to(dev)
            rows = (torch.randn(idx.numel(), 48, generator=g) * 0.01).to(dev)
            ob.catch_up_rows([(Pb, row_of, None)])
            sel = idx.to(dev)
            assert torch.equal(Pa[sel], Pb[sel]), step
        else:
            row_of, rows = empty, zero_rows
        for P, o in ((Pa, oa), (Pb, ob)):
            o.set_row_gradient(P, rows, row_of, 3)
            o.step()
    ob.flush()
    assert torch.equal(Pa, Pb)
    assert torch.equal(oa.state[Pa]["exp_avg"], ob.state[Pb]["exp_avg"])
    assert torch.equal(oa.state[Pa]["exp_avg_sq"], ob.state[Pb]["exp_avg_sq"])


def test_row_map_entries_beyond_the_row_buffer_are_not_read(hip_lib):
    """A frame whose visible count exceeded its capacity (graph mode) leaves ranks >= the capacity in the row map while the
    gradient rows end at the capacity: such entries count as "no row" (zero gradient) -- never an out-of-bounds read --
    for the streaming groups and for the row-lazy ones."""
    from mtgs_amd.optim import FusedAdam
    dev = torch.device("cuda")
    N, R = 5000, 600
    g = torch.Generator().manual_seed(21)
    base = torch.randn(N, 15, 3, generator=g)
    rows = (torch.randn(R, 48, generator=g) * 0.01).to(dev)
    perm = torch.randperm(N, generator=g)[:900]
    over = torch.full((N,), -1, dtype=torch.int32)
    over[perm] = torch.arange(900, dtype=torch.int32)              # ranks 600 .. 899 have no row
    cut = over.clone()
    cut[over >= R] = -1
    outs = []
    for lazy in (False, True):
        for row_of in (over, cut):
            p = base.clone().to(dev).requires_grad_(True)
            o = FusedAdam([p], lr=1e-2, eps=1e-15)
            if lazy:
                o.set_row_lazy(p)
            for _ in range(3):
                o.set_row_gradient(p, rows, row_of.to(dev), 3)
                o.step()
            o.flush()
            outs.append(p.detach().clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3])


def test_row_lazy_peek_leaves_the_optimizer_alone_and_feeds_the_step(hip_lib):
    """peek_rows: the up-to-date rows of the visible Gaussians in a compact buffer (what the colour kernel reads) WITHOUT any
    change to parameters, moments or stamps; handed back through set_row_gradient(caught=...) the step takes the parameter
    from them and replays only the moments of the missed steps.  Over a random sequence: (1) the peeked rows are bit-identical
    to the every-row optimizer's parameters, (2) a peek changes nothing, (3) a peek that is NOT followed by its step (an
    evaluation frame) is harmless, (4) after flush() everything is bit-identical, (5) tensors that are not row-lazy are
    copied, (6) ranks beyond the buffer are skipped."""
    dev = torch.device("cuda")
    N, T = 2503, 3
    g = torch.Generator().manual_seed(17)
    base, make = _row_lazy_case(dev, N, T, g)
    Pa, oa = make(False)
    Pb, ob = make(True, hist_capacity=8)
    seq = [0, 2, 2, 1, 0, 1, 1, 2, 0, 0, 1, 2, 2, 0]
    for step, t in enumerate(seq):
        vis, row_of, rows = _frame(N, g, dev, frac=0.3 if step % 4 else 0.06)
        R = rows.shape[0]
        for o in (oa, ob):
            o.param_groups[0]["lr"] = 1e-2 * 0.9 ** step
        items = [(Pb["dc"], row_of, None, 0), (Pb["adapters"], row_of, t, 3), (Pb["rest"], row_of, t, 6), (Pb["means"], row_of, None, 51)]
        before = {k: Pb[k].detach().clone() for k in base}
        mom = ob.state[Pb["rest"]]["exp_avg"].clone() if step else None
        C = torch.full((R, 56), float("nan"), device=dev)
        ob.peek_rows(items, C)
        for k in base:
            assert torch.equal(before[k], Pb[k]), (step, k)                                  # (2)
        if mom is not None:
            assert torch.equal(mom, ob.state[Pb["rest"]]["exp_avg"])
        r = row_of[vis].long()
        assert torch.equal(C[r, 0:3], Pa["dc"][vis]) and torch.equal(C[r, 3:6], Pa["adapters"][vis, t]), step       # (1)
        assert torch.equal(C[r, 6:51], Pa["rest"][vis, t].reshape(-1, 45)), step
        assert torch.equal(C[r, 51:54], Pa["means"][vis]), step                              # (5)
        if step == 5:                                                                        # (3) an evaluation frame: no step
            continue
        if step == 7:                                                                        # (6) a buffer that is too short
            short = torch.full((max(R // 2, 1), 56), float("nan"), device=dev)
            ob.peek_rows(items, short)
            keep = r < short.shape[0]
            assert torch.equal(short[r[keep], 6:51], C[r[keep], 6:51])
        grad_means = (torch.randn(N, 3, generator=g) * 0.1).to(dev)
        for P, o, ck in ((Pa, oa, None), (Pb, ob, C)):
            P["means"].grad = grad_means.clone()
            kw = (lambda col: {"caught": (ck, col)}) if ck is not None else (lambda col: {})
            o.set_row_gradient(P["dc"], rows, row_of, 0, **kw(0))
            o.set_row_gradient(P["adapters"], rows, row_of, 0, slice_index=t, **kw(3))
            o.set_row_gradient(P["rest"], rows, row_of, 3, slice_index=t, **kw(6))
            o.step()
        assert torch.equal(Pa["rest"][vis, t], Pb["rest"][vis, t]) and torch.equal(Pa["dc"][vis], Pb["dc"][vis]), step
    assert not torch.equal(Pa["rest"], Pb["rest"])
    ob.flush()
    for k in base:
        assert torch.equal(Pa[k], Pb[k]), k                                                  # (4)
        assert torch.equal(oa.state[Pa[k]]["exp_avg"], ob.state[Pb[k]]["exp_avg"]), k
        assert torch.equal(oa.state[Pa[k]]["exp_avg_sq"], ob.state[Pb[k]]["exp_avg_sq"]), k


def test_row_lazy_list_form_equals_scan_form_over_two_nodes(hip_lib):
    """The LIST form (rows straight from the frame's list of visible Gaussians: `row_ids`, one row per 16 lanes) against the
    SCAN form (row map only) for two nodes that share one id list -- peek, caught step, in-place catch-up, a device-side row
    count smaller than the buffers, an empty node range: bit-identical parameters, moments and peeked rows."""
    from mtgs_amd.optim import FusedAdam
    dev = torch.device("cuda")
    N1, N2, T = 1800, 700, 2
    N = N1 + N2
    g = torch.Generator().manual_seed(23)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.2)
    base = {"dc1": mk(N1, 3), "rest1": mk(N1, T, 15, 3), "dc2": mk(N2, 3), "rest2": mk(N2, 15, 3)}

    def make():
        P = {k: v.clone().to(dev).requires_grad_(True) for k, v in base.items()}
        o = FusedAdam([{"params": [P["dc1"], P["dc2"]], "lr": 3e-3}, {"params": [P["rest1"], P["rest2"]], "lr": 1e-2}], eps=1e-15)
        o.set_row_lazy(P["dc1"]); o.set_row_lazy(P["dc2"]); o.set_row_lazy(P["rest1"], traversals=T); o.set_row_lazy(P["rest2"])
        return P, o
    Ps, os_ = make()
    Pl, ol = make()
    for step in range(12):
        t = step % T
        frac = 0.25 if step % 5 else 0.05
        vis = torch.rand(N, generator=g) < frac
        if step == 7:
            vis[N1:] = False                                   # nothing of the second node is visible
        ids = torch.nonzero(vis).reshape(-1).to(torch.int32)
        n_vis = ids.numel()
        cap = n_vis + 37                                       # buffers larger than the count (graph mode)
        row_of = torch.full((N,), -1, dtype=torch.int32)
        row_of[vis] = torch.arange(n_vis, dtype=torch.int32)
        ids_d = torch.cat([ids, torch.full((37,), 2 ** 30, dtype=torch.int32)]).to(dev)      # garbage beyond the count
        row_of, rows = row_of.to(dev), (torch.randn(cap, 48, generator=g) * 0.01).to(dev)
        totals = torch.tensor([(n_vis << 32) | 5], dtype=torch.int64, device=dev)
        outs = []
        for P, o, lst in ((Ps, os_, False), (Pl, ol, True)):
            for grp in o.param_groups:
                grp["lr"] = grp["lr"] * 0.97
            rid = lambda start: ((ids_d, start, totals),) if lst else ()
            C = torch.zeros(cap, 52, device=dev)
            if step % 3 == 2:      # the in-place form
                o.catch_up_rows([(P["dc1"], row_of[:N1], None) + rid(0), (P["rest1"], row_of[:N1], t) + rid(0),
                                 (P["dc2"], row_of[N1:], None) + rid(N1), (P["rest2"], row_of[N1:], None) + rid(N1)])
                ck = lambda col: {}
            else:
                o.peek_rows([(P["dc1"], row_of[:N1], None, 0) + rid(0), (P["rest1"], row_of[:N1], t, 6) + rid(0),
                             (P["dc2"], row_of[N1:], None, 0) + rid(N1), (P["rest2"], row_of[N1:], None, 6) + rid(N1)], C)
                ck = lambda col: {"caught": (C, col)}
            kw = lambda start: ({"row_ids": (ids_d, start, totals)} if lst else {})
            o.set_row_gradient(P["dc1"], rows, row_of[:N1], 0, **ck(0), **kw(0))
            o.set_row_gradient(P["rest1"], rows, row_of[:N1], 3, slice_index=t, **ck(6), **kw(0))
            o.set_row_gradient(P["dc2"], rows, row_of[N1:], 0, **ck(0), **kw(N1))
            o.set_row_gradient(P["rest2"], rows, row_of[N1:], 3, **ck(6), **kw(N1))
            o.step()
            outs.append(C[:n_vis].clone())
        assert torch.equal(outs[0], outs[1]), step
        for k in base:
            assert torch.equal(Ps[k], Pl[k]), (step, k)
    os_.flush(); ol.flush()
    for k in base:
        assert torch.equal(Ps[k], Pl[k]) and torch.equal(os_.state[Ps[k]]["exp_avg_sq"], ol.state[Pl[k]]["exp_avg_sq"]), k
