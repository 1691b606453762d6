"""Fused per-node activations (mtgs_amd.nodes.node_gaussians, csrc/node.hip) against the operator chain of
VanillaGaussianSplattingModel.get_gaussians restated in plain PyTorch
(/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:299-341; multi-colour sources:
multi_color_gaussian_splatting.py:77-101), with the SH colour evaluated by the independent fp64 restatement
oracle/torch_ref.py.  Tolerances: forward 2e-6 absolute (fp32), gradients 1e-5 relative to the tensor's max."""
import pytest
import torch

from tests.util import listed

pytestmark = pytest.mark.gpu


def _reference(P, cam_pos, n, model_deg, t=None):
    """The reference's op chain, fp64, on CPU (torch_ref.spherical_harmonics is the oracle's restatement)."""
    from oracle import torch_ref
    means, scales, quats, opac = P["means"], P["scales"], P["quats"], P["opacities"]
    if t is None:
        dc, rest = P["features_dc"], P["features_rest"]
    else:  # MultiColorGaussianSplattingModel.get_pertravel_features
        dc = P["features_dc"] + P["features_adapters"][:, t, :]
        rest = P["features_rest"][:, t, :, :]
    out = {"scales": torch.exp(scales), "quats": quats / quats.norm(dim=-1, keepdim=True),
           "opacities": torch.sigmoid(opac).squeeze(-1)}
    colors = torch.cat((dc[:, None, :], rest), dim=1)
    if model_deg > 0:
        d = means.detach() - cam_pos
        d = d / d.norm(dim=-1, keepdim=True)
        out["rgbs"] = torch.clamp(torch_ref.spherical_harmonics(n, d, colors) + 0.5, 0.0, 1.0)
    else:
        out["rgbs"] = torch.sigmoid(colors[:, 0, :])
    return out


def _params(N, K, T, seed):
    g = torch.Generator().manual_seed(seed)
    P = {"means": torch.randn(N, 3, generator=g) * 5, "scales": torch.randn(N, 3, generator=g) - 2,
         "quats": torch.randn(N, 4, generator=g), "opacities": torch.randn(N, 1, generator=g),
         "features_dc": torch.randn(N, 3, generator=g) * 0.7}
    if T:
        P["features_rest"] = torch.randn(N, T, K - 1, 3, generator=g) * 0.2
        P["features_adapters"] = torch.randn(N, T, 3, generator=g) * 0.1
    else:
        P["features_rest"] = torch.randn(N, K - 1, 3, generator=g) * 0.2
    return P


@pytest.mark.parametrize("N,K,n,model_deg,T", [
    (4099, 16, 3, 3, 0), (4099, 16, 1, 3, 0), (1000, 16, 0, 3, 0),   # degree ramp 0 -> 3 with K fixed at 16 (MTGS.py:72-73)
    (777, 9, 2, 2, 0),                                               # WildGaussians-sized coefficients (sh_degree 2)
    (513, 16, 0, 0, 0),                                              # sh_degree 0 model: rgbs = sigmoid(features_dc)
    (2050, 16, 3, 3, 3),                                             # multi-colour node, traversal 1 of 3, strided views
    (2050, 16, 2, 3, -3),                                            # same, FULL per-traversal tensors + traversal_index
])
def test_node_gaussians_match_reference_chain(hip_lib, N, K, n, model_deg, T):
    full = T < 0
    T = abs(T)
    from mtgs_amd.nodes import node_gaussians
    dev = torch.device("cuda")
    P = _params(N, K, T, 11)
    c2w = torch.eye(4)[None, :3, :].clone()
    c2w[0, :3, 3] = torch.tensor([0.4, -1.1, 2.3])
    t = 1 if T else None
    g = torch.Generator().manual_seed(3)
    cot = {"scales": torch.randn(N, 3, generator=g), "quats": torch.randn(N, 4, generator=g),
           "opacities": torch.randn(N, generator=g), "rgbs": torch.randn(N, 3, generator=g)}
    # reference (fp64, CPU)
    R = {k: v.double().requires_grad_(True) for k, v in P.items()}
    ref = _reference(R, c2w[0, :3, 3].double(), n, model_deg, t)
    sum((ref[k] * cot[k].double()).sum() for k in cot).backward()
    # fused (HIP)
    D = {k: v.to(dev).requires_grad_(True) for k, v in P.items()}
    if T and full:
        out = node_gaussians(D["means"], D["scales"], D["quats"], D["opacities"], D["features_dc"], D["features_rest"],
                             c2w.to(dev), n, model_deg, features_dc_add=D["features_adapters"], traversal_index=t)
    elif T:
        out = node_gaussians(D["means"], D["scales"], D["quats"], D["opacities"], D["features_dc"],
                             D["features_rest"][:, t], c2w.to(dev), n, model_deg,
                             features_dc_add=D["features_adapters"][:, t])
    else:
        out = node_gaussians(D["means"], D["scales"], D["quats"], D["opacities"], D["features_dc"], D["features_rest"],
                             c2w.to(dev), n, model_deg)
    assert out["means"] is D["means"]
    sum((out[k] * cot[k].to(dev)).sum() for k in cot).backward()
    for k in cot:
        err = float((out[k].detach().cpu().double() - ref[k].detach()).abs().max())
        assert err <= 2e-6 * max(1.0, float(ref[k].detach().abs().max())), f"{k}: {err}"
    for k in P:
        if k == "means":
            assert D[k].grad is None and R[k].grad is None  # view directions are detached (vanilla :314)
            continue
        gd, gr = D[k].grad, R[k].grad
        assert gd is not None and gd.shape == gr.shape, k
        scale = float(gr.abs().max())
        err = float((gd.cpu().double() - gr).abs().max())
        assert err <= 1e-5 * scale + 1e-9, f"grad {k}: {err} vs {scale}"


def test_node_rgbs_golden_reference_helpers(hip_lib):
    """A constant-colour Gaussian: features_dc = RGB2SH(rgb), higher orders zero => rgbs == rgb for every view
    direction.  The expected values come from the reference's own RGB2SH (tests/golden/ref_helpers.npz,
    generated by tests/golden/make_golden.py from mtgs/scene_model/gaussian_model/utils.py)."""
    import numpy as np
    from pathlib import Path
    from mtgs_amd.nodes import node_gaussians
    z = np.load(Path(__file__).parent / "golden" / "ref_helpers.npz")
    rgb, sh = torch.from_numpy(z["rgb"]).float(), torch.from_numpy(z["rgb2sh"]).float()
    N = rgb.shape[0]
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    out = node_gaussians(torch.randn(N, 3, generator=g).to(dev), torch.zeros(N, 3, device=dev),
                         torch.randn(N, 4, generator=g).to(dev), torch.zeros(N, 1, device=dev), sh.to(dev),
                         torch.zeros(N, 15, 3, device=dev), torch.eye(4, device=dev)[None, :3], 3, 3)
    assert float((out["rgbs"].cpu() - rgb.clamp(0, 1)).abs().max()) <= 2e-6
    assert float((out["scales"] - 1.0).abs().max()) == 0.0 and float((out["opacities"] - 0.5).abs().max()) == 0.0


def test_collect_gaussians_equals_per_node_concatenation(hip_lib):
    """mtgs_amd.nodes.collect_gaussians (one autograd node, kernels write into slices) == node_gaussians per node +
    torch.cat, as MTGSSceneModel.get_gaussians concatenates (mtgs_scene_graph.py:408-461)."""
    from mtgs_amd.nodes import collect_gaussians, node_gaussians
    dev = torch.device("cuda")
    c2w = torch.eye(4)[None, :3].clone().to(dev)
    c2w[0, :3, 3] = torch.tensor([0.4, -1.1, 2.3])
    base = [_params(1500, 16, 3, 1), _params(64, 16, 0, 2), _params(1, 16, 0, 3), _params(777, 16, 4, 4)]
    travs = [1, None, None, 3]
    g = torch.Generator().manual_seed(8)
    total = sum(p["means"].shape[0] for p in base)
    cot = {"means": torch.randn(total, 3, generator=g), "scales": torch.randn(total, 3, generator=g),
           "quats": torch.randn(total, 4, generator=g), "opacities": torch.randn(total, generator=g),
           "rgbs": torch.randn(total, 3, generator=g)}
    res = []
    for collected in (True, False):
        P = [{k: v.to(dev).requires_grad_(True) for k, v in p.items()} for p in base]
        P[2]["scales"].requires_grad_(False)                     # a frozen parameter
        if collected:
            out = collect_gaussians([dict(p, traversal_index=t) if t is not None else p for p, t in zip(P, travs)], c2w, 2, 3)
        else:
            parts = [node_gaussians(p["means"], p["scales"], p["quats"], p["opacities"], p["features_dc"], p["features_rest"], c2w, 2, 3,
                                    features_dc_add=p.get("features_adapters"), traversal_index=t) for p, t in zip(P, travs)]
            out = {k: torch.cat([q[k] for q in parts], 0) for k in cot}
        sum((out[k] * cot[k].to(dev)).sum() for k in cot).backward()
        res.append(({k: out[k].detach() for k in cot}, [{k: v.grad for k, v in p.items()} for p in P], out))
    for k in cot:
        assert torch.equal(res[0][0][k], res[1][0][k]), k
    for ga, gb in zip(res[0][1], res[1][1]):
        for k in ga:
            assert (ga[k] is None) == (gb[k] is None), k
            if ga[k] is not None:
                assert torch.equal(ga[k], gb[k]), k
    mid = res[0][2]["model_id"]
    assert mid.shape == (total,) and int(mid[0]) == 0 and int(mid[-1]) == 3 and int((mid == 1).sum()) == 64


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_rigid_node_pose_reference_vectors(hip_lib, case):
    """Rigid nodes: global means / quaternions and their gradients (to the local means, the local quaternions and the
    instance pose) against vectors produced by the reference's own quat_to_rotmat / quat_mult
    (tests/golden/rigid_ref.npz, tests/golden/make_rigid_golden.py; rigid_node.py:205-216)."""
    import numpy as np
    from pathlib import Path
    from mtgs_amd.nodes import collect_gaussians, node_gaussians
    Z = np.load(Path(__file__).parent / "golden" / "rigid_ref.npz")
    dev = torch.device("cuda")
    f = lambda k: torch.from_numpy(Z[f"{case}_{k}"]).float().to(dev)
    N = Z[f"{case}_means"].shape[0]
    for collected in (False, True):
        means, quats, q, t = (f(k).requires_grad_(True) for k in ("means", "quats", "q", "t"))
        z = lambda *s: torch.zeros(*s, device=dev, requires_grad=True)
        node = {"means": means, "scales": z(N, 3), "quats": quats, "opacities": z(N, 1), "features_dc": z(N, 3),
                "features_rest": z(N, 15, 3), "instance_quat": q, "instance_trans": t}
        c2w = torch.eye(4, device=dev)[None, :3]
        if collected:
            extra = {k: v.detach().clone().requires_grad_(True) for k, v in node.items() if not k.startswith("instance")}
            out = collect_gaussians([extra, node], c2w, 3, 3)          # a static node in front: slices, not whole tensors
            gm, gq = out["means"][N:], out["quats"][N:]
            assert torch.equal(out["means"][:N], extra["means"])
        else:
            out = node_gaussians(means, node["scales"], quats, node["opacities"], node["features_dc"], node["features_rest"], c2w,
                                 3, 3, instance_quat=q, instance_trans=t)
            gm, gq = out["means"], out["quats"]
        for got, name in ((gm, "global_means"), (gq, "global_quats")):
            ref = Z[f"{case}_{name}"]
            assert np.abs(got.detach().cpu().numpy() - ref).max() <= 3e-6 * max(1.0, np.abs(ref).max()), name
        ((gm * f("Gm")).sum() + (gq * f("Gq")).sum()).backward()
        for p, name in ((means, "g_means"), (quats, "g_quats"), (q, "g_q"), (t, "g_t")):
            ref = Z[f"{case}_{name}"]
            err = np.abs(p.grad.cpu().numpy().astype(np.float64) - ref).max()
            assert err <= 2e-5 * max(1.0, np.abs(ref).max()), f"{name} ({'collected' if collected else 'single'}): {err}"


def test_rigid_node_colours_use_the_global_means(hip_lib):
    """rigid_node.py:238-251: the view directions of a rigid node come from the GLOBAL means."""
    from mtgs_amd.nodes import node_gaussians
    dev = torch.device("cuda")
    P = {k: v.to(dev) for k, v in _params(3001, 16, 0, 21).items()}
    g = torch.Generator().manual_seed(2)
    q = torch.randn(4, generator=g); q = (q / q.norm()).to(dev)
    t = (torch.randn(3, generator=g) * 4).to(dev)
    c2w = torch.eye(4, device=dev)[None, :3].clone(); c2w[0, :, 3] = torch.tensor([1.0, -2.0, 0.5], device=dev)
    rigid = node_gaussians(P["means"], P["scales"], P["quats"], P["opacities"], P["features_dc"], P["features_rest"], c2w, 3, 3,
                           instance_quat=q, instance_trans=t)
    static = node_gaussians(rigid["means"].detach(), P["scales"], P["quats"], P["opacities"], P["features_dc"], P["features_rest"],
                            c2w, 3, 3)
    assert torch.allclose(rigid["rgbs"], static["rgbs"], atol=1e-6) and torch.equal(rigid["scales"], static["scales"])
    assert not torch.allclose(rigid["means"], P["means"])


def test_collect_many_small_nodes_one_launch(hip_lib):
    """A scene graph with 150 nodes -- background, a multi-colour node, and many small rigid object nodes with awkward
    sizes (0, 1, 63, 64, 65, 255, 256, 257 ...) -- through the batched launch (mtgs_node_fwd_batch / mtgs_node_bwd_batch)
    equals the per-node calls bit for bit, forward and backward, including the pose gradients and model_id."""
    from mtgs_amd.nodes import collect_gaussians, node_gaussians
    dev = torch.device("cuda")
    c2w = torch.eye(4)[None, :3].clone().to(dev)
    c2w[0, :3, 3] = torch.tensor([0.4, -1.1, 2.3])
    g = torch.Generator().manual_seed(21)
    sizes = [5000, 3001] + [0, 1, 63, 64, 65, 255, 256, 257, 511, 513] + [int(x) for x in torch.randint(1, 900, (138,), generator=g)]
    base, travs = [], []
    for i, n in enumerate(sizes):
        p = _params(n, 16, 3 if i == 1 else 0, 100 + i)
        if i >= 2:   # rigid object node with the pose of the current frame
            q = torch.randn(4, generator=g)
            p["instance_quat"], p["instance_trans"] = q / q.norm(), torch.randn(3, generator=g) * 3
        base.append(p)
        travs.append(2 if i == 1 else None)
    total = sum(sizes)
    cot = {"means": torch.randn(total, 3, generator=g), "scales": torch.randn(total, 3, generator=g),
           "quats": torch.randn(total, 4, generator=g), "opacities": torch.randn(total, generator=g),
           "rgbs": torch.randn(total, 3, generator=g)}
    res = []
    for collected in (True, False):
        P = [{k: v.to(dev).requires_grad_(True) for k, v in p.items()} for p in base]
        P[5]["quats"].requires_grad_(False)
        P[7]["means"].requires_grad_(False)
        if collected:
            out = collect_gaussians([dict(p, traversal_index=t) if t is not None else p for p, t in zip(P, travs)], c2w, 3, 3)
        else:
            parts = [node_gaussians(p["means"], p["scales"], p["quats"], p["opacities"], p["features_dc"], p["features_rest"], c2w, 3, 3,
                                    features_dc_add=p.get("features_adapters"), traversal_index=t,
                                    instance_quat=p.get("instance_quat"), instance_trans=p.get("instance_trans"))
                     for p, t in zip(P, travs)]
            out = {k: torch.cat([q[k] for q in parts], 0) for k in cot}
        sum((out[k] * cot[k].to(dev)).sum() for k in cot).backward()
        res.append(({k: out[k].detach() for k in cot}, [{k: v.grad for k, v in p.items()} for p in P], out))
    for k in cot:
        assert torch.equal(res[0][0][k], res[1][0][k]), k
    for i, (ga, gb) in enumerate(zip(res[0][1], res[1][1])):
        for k in ga:
            assert (ga[k] is None) == (gb[k] is None), (i, k)
            if ga[k] is None:
                continue
            if k in ("instance_quat", "instance_trans"):   # per-wave atomics: summation order differs
                assert torch.allclose(ga[k], gb[k], rtol=1e-4, atol=1e-4 * float(gb[k].abs().max()) + 1e-6), (i, k)
            else:
                assert ga[k].shape == gb[k].shape and torch.equal(ga[k], gb[k]), (i, k)
    mid = res[0][2]["model_id"].cpu()
    expect = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    assert torch.equal(mid, expect)


def test_collect_rigid_nodes_with_per_frame_pose_parameters(hip_lib):
    """Rigid nodes given as their per-frame pose PARAMETERS + frame_idx: the batched launch reads row frame_idx, normalises
    the quaternion (RigidSubModel.get_object_pose, rigid_node.py:139-144) and returns the gradients of the full tables.
    Reference: the same indexing / normalisation in torch in front of the explicit-pose path."""
    from mtgs_amd.nodes import collect_gaussians
    dev = torch.device("cuda")
    c2w = torch.eye(4)[None, :3].clone().to(dev)
    c2w[0, :3, 3] = torch.tensor([0.4, -1.1, 2.3])
    g = torch.Generator().manual_seed(33)
    sizes = [2000, 700, 65, 1, 300]
    frames = [None, 12, 40, 3, 7]          # node 0 is static
    fidx = [None, 5, 39, 0, 6]
    base = []
    for i, (n, F) in enumerate(zip(sizes, frames)):
        p = _params(n, 16, 0, 200 + i)
        if F is not None:
            p["instance_quats"], p["instance_trans"] = torch.randn(F, 4, generator=g) * 1.7, torch.randn(F, 3, generator=g) * 3
        base.append(p)
    total = sum(sizes)
    cot = {k: torch.randn(total, w, generator=g).squeeze(-1) for k, w in (("means", 3), ("scales", 3), ("quats", 4), ("opacities", 1), ("rgbs", 3))}
    res = []
    for in_kernel in (True, False, "device word"):
        P = [{k: v.to(dev).requires_grad_(True) for k, v in p.items()} for p in base]
        nodes = []
        for p, f in zip(P, fidx):
            if f is None:
                nodes.append(p)
            elif in_kernel == "device word":   # the frame as an int32 DEVICE scalar (one captured iteration for every frame)
                nodes.append(dict(p, frame_idx=torch.tensor(f, dtype=torch.int32, device=dev)))
            elif in_kernel:
                nodes.append(dict(p, frame_idx=f))
            else:
                q = p["instance_quats"][f] / p["instance_quats"][f].norm(dim=-1, keepdim=True)
                nd = {k: v for k, v in p.items() if k not in ("instance_quats", "instance_trans")}
                nodes.append(dict(nd, instance_quat=q, instance_trans=p["instance_trans"][f]))
        out = collect_gaussians(nodes, c2w, 3, 3)
        sum((out[k] * cot[k].to(dev)).sum() for k in cot).backward()
        res.append(({k: out[k].detach() for k in cot}, [{k: v.grad for k, v in p.items()} for p in P]))
    for k in cot:
        assert torch.allclose(res[0][0][k], res[1][0][k], rtol=1e-5, atol=1e-5), k
        assert torch.equal(res[0][0][k], res[2][0][k]), k                      # the device word: the same kernels on the same rows
    for ga, gc in zip(res[0][1], res[2][1]):
        for k in ga:
            assert float((ga[k] - gc[k]).abs().max()) <= 2e-5 * (float(ga[k].abs().max()) + 1e-12), k      # (atomics order)
    for i, (ga, gb) in enumerate(zip(res[0][1], res[1][1])):
        for k in ga:
            assert ga[k] is not None and gb[k] is not None, (i, k)
            scale = float(gb[k].abs().max()) + 1e-12
            assert ga[k].shape == gb[k].shape and float((ga[k] - gb[k]).abs().max()) <= 2e-4 * scale, (i, k)
        if frames[i] is not None:   # only the row of the frame carries a gradient
            other = torch.ones(frames[i], dtype=torch.bool); other[fidx[i]] = False
            assert not ga["instance_quats"][other.to(dev)].any() and not ga["instance_trans"][other.to(dev)].any()
            assert ga["instance_quats"][fidx[i]].abs().sum() > 0


@pytest.mark.parametrize("degree,extra,inside", [(3, 0, False), (1, 3, False), (2, 3, True)])
def test_visibility_first_colours_equal_the_dense_node_path(hip_lib, degree, extra, inside):
    """collect_gaussians(deferred_colors=True) + rasterization(color_source=...): SH + clamp evaluated for the VISIBLE Gaussians
    only (csrc/viscolor.hip), coefficient gradients as compact rows.  Against the dense node path (colours of every Gaussian,
    dense gradients) on a scene with a vanilla node, a multi-colour node (per-traversal coefficients + adapters) and a
    shared-rest multi-colour node: same image, same geometry gradients, row gradients equal to the dense ones on the
    visible rows (zero elsewhere, and in the other traversals' slices), and the same parameters after a FusedAdam step from
    rows as from the dense gradients.  inside: the three camera-space normal channels are computed for the visible Gaussians
    inside the rasterization too (ColorSource.camera_normals; mtgs_normals_fwd_rows / mtgs_normals_bwd_qrows) instead of being
    handed over as extra colours of every Gaussian -- same image, same quaternion gradients."""
    from mtgs_amd import rasterization
    from mtgs_amd.nodes import camera_space_normals, collect_gaussians
    from mtgs_amd.optim import FusedAdam
    from mtgs_amd.synthetic import make_camera
    dev = torch.device("cuda")
    W, H, T, t = 320, 200, 3, 1
    g = torch.Generator().manual_seed(21)

    def node(n, kind, seed):
        gg = torch.Generator().manual_seed(seed)
        P = {"means": (torch.rand(n, 3, generator=gg) * 2 - 1) * torch.tensor([8.0, 2.0, 8.0]) + torch.tensor([0.0, 0.0, 6.0]),
             "scales": torch.log(torch.rand(n, 3, generator=gg) * 0.2 + 0.03), "quats": torch.randn(n, 4, generator=gg),
             "opacities": torch.randn(n, 1, generator=gg), "features_dc": torch.randn(n, 3, generator=gg) * 0.7}
        if kind == "vanilla":
            P["features_rest"] = torch.randn(n, 15, 3, generator=gg) * 0.2
        elif kind == "multi":
            P["features_rest"] = torch.randn(n, T, 15, 3, generator=gg) * 0.2
            P["features_adapters"] = torch.randn(n, T, 3, generator=gg) * 0.1
        else:   # adapters per traversal, shared features_rest
            P["features_rest"] = torch.randn(n, 15, 3, generator=gg) * 0.2
            P["features_adapters"] = torch.randn(n, T, 3, generator=gg) * 0.1
        return P

    raw = [node(9000, "vanilla", 1), node(7001, "multi", 2), node(3000, "shared", 3)]
    vm, K = make_camera(W, H)
    vm, K = vm.to(dev), K.to(dev)
    c2w = torch.inverse(vm)[:, :3, :]
    Gc = torch.randn(1, H, W, 4 + extra, generator=g).to(dev)
    Ga = torch.randn(1, H, W, 1, generator=g).to(dev)

    def run(deferred):
        P = [{k: v.clone().to(dev).requires_grad_(True) for k, v in nd.items()} for nd in raw]
        nodes = [dict(p, traversal_index=t) if "features_adapters" in p else p for p in P]
        gs = collect_gaussians(nodes, c2w, degree, deferred_colors=deferred)
        if deferred:
            cols = camera_space_normals(gs["quats"], gs["scales"], gs["means"], c2w) if (extra and not inside) else None
            if inside:
                gs["color_source"].camera_normals = c2w[0].contiguous()
            r, a, info = rasterization(gs["means"], gs["quats"], gs["scales"], gs["opacities"], cols, vm, K, W, H, packed=False,
                                       render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True, color_source=gs["color_source"])
        else:
            cols = camera_space_normals(gs["quats"], gs["scales"], gs["means"], c2w, rgbs=gs["rgbs"]) if extra else gs["rgbs"]
            r, a, info = rasterization(gs["means"], gs["quats"], gs["scales"], gs["opacities"], cols, vm, K, W, H, packed=False,
                                       render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True)
        torch.autograd.backward([r, a], [Gc, Ga])
        return P, r.detach(), a.detach(), info, gs.get("color_source")

    Pd, rd, ad, info_d, _ = run(False)
    Pv, rv, av, info_v, cs = run(True)
    n_vis = int((info_d["radii"] > 0).sum())
    assert 2000 < n_vis < 15000
    assert torch.equal(listed(info_d), listed(info_v))
    assert float((rd - rv).abs().max()) <= 2e-6 * max(1.0, float(rd.abs().max())) and float((ad - av).abs().max()) <= 2e-6
    for pd, pv in zip(Pd, Pv):
        for k in ("means", "scales", "quats", "opacities"):
            ref = pd[k].grad
            assert float((pv[k].grad - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-9, k    # (fp32 atomics order)
        assert pv["features_dc"].grad is None and pv["features_rest"].grad is None
    for (g_dc, g_ad, g_rest), pd in zip(cs.dense_gradients(), Pd):
        for got, ref in ((g_dc, pd["features_dc"].grad), (g_rest, pd["features_rest"].grad),
                         (g_ad, pd["features_adapters"].grad if "features_adapters" in pd else None)):
            if ref is None:
                assert got is None
                continue
            assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-9
    # the same FusedAdam step from the rows as from the dense gradients
    col_keys = ("features_dc", "features_rest", "features_adapters")
    od = FusedAdam([{"params": [p[k] for p in Pd for k in col_keys if k in p], "lr": 1e-2}], eps=1e-15)
    ov = FusedAdam([{"params": [p[k] for p in Pv for k in col_keys if k in p], "lr": 1e-2}], eps=1e-15)
    cs.apply_to(ov)
    od.step()
    ov.step()
    for pd, pv in zip(Pd, Pv):
        for k in col_keys:
            if k in pd:
                assert float((pd[k] - pv[k]).abs().max()) <= 2e-6, k
                # rows the frame did not see, and the other traversals' slices, took the zero-gradient update: unchanged at step 1
    vis = (info_d["radii"][0] > 0)
    n0 = raw[0]["means"].shape[0]
    assert torch.equal(Pv[0]["features_rest"][~vis[:n0]].detach().cpu(), raw[0]["features_rest"][~vis[:n0].cpu()])
    assert torch.equal(Pv[1]["features_rest"][:, 0].detach().cpu(), raw[1]["features_rest"][:, 0])


@pytest.mark.parametrize("inside,dense_scene", [(False, False), (True, True)])
def test_touch_first_changes_no_pixel_and_finds_exactly_the_rows_with_a_gradient(hip_lib, inside, dense_scene):
    """ColorSource.touch_first: the rasterization bins first, one pass of the compositing DECISIONS (mtgs_blend_touch_packed) flags
    the Gaussians the frame composites from, and the SH evaluation / normals run for those alone (the others get a constant: their
    weight is zero wherever the compositing meets them).  Against the same call without the pass: render and alphas BIT-identical,
    the same gradient rows; a Gaussian without the flag has an all-zero gradient row, and (random cotangents) every flagged one a
    non-zero one -- the flags ARE the set of rows that receive a gradient."""
    from mtgs_amd import rasterization, wrapper
    from mtgs_amd.nodes import collect_gaussians
    from mtgs_amd.synthetic import make_camera
    dev = torch.device("cuda")
    W, H, T, t = 320, 200, 3, 2
    g = torch.Generator().manual_seed(77)
    n = 60_000 if dense_scene else 12_000
    raw = {"means": (torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([8.0, 2.0, 8.0]) + torch.tensor([0.0, 0.0, 6.0]),
           "scales": torch.log(torch.rand(n, 3, generator=g) * (0.5 if dense_scene else 0.2) + 0.03), "quats": torch.randn(n, 4, generator=g),
           "opacities": torch.randn(n, 1, generator=g) + (2.0 if dense_scene else 0.0), "features_dc": torch.randn(n, 3, generator=g) * 0.7,
           "features_rest": torch.randn(n, T, 15, 3, generator=g) * 0.2, "features_adapters": torch.randn(n, T, 3, generator=g) * 0.1}
    vm, K = make_camera(W, H)
    vm, K = vm.to(dev), K.to(dev)
    c2w = torch.inverse(vm)[:, :3, :]
    D = 7 if inside else 4
    Gc, Ga = torch.randn(1, H, W, D, generator=g).to(dev), torch.randn(1, H, W, 1, generator=g).to(dev)

    def run(touch):
        P = {k: v.clone().to(dev).requires_grad_(True) for k, v in raw.items()}
        gs = collect_gaussians([dict(P, traversal_index=t)], c2w, 3, deferred_colors=True)
        cs = gs["color_source"]
        cs.touch_first = touch
        if inside:
            cs.camera_normals = c2w[0].contiguous()
        dbg = {}
        wrapper._debug_rows = dbg
        try:
            r, a, info = rasterization(gs["means"], gs["quats"], gs["scales"], gs["opacities"], None, vm, K, W, H, packed=False,
                                       render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True, color_source=cs)
            torch.autograd.backward([r, a], [Gc, Ga])
        finally:
            wrapper._debug_rows = None
        return P, r.detach(), a.detach(), info, cs, dbg["G"]

    P0, r0, a0, i0, cs0, G0 = run(False)
    P1, r1, a1, i1, cs1, G1 = run(True)
    assert cs0.row_flags is None and cs1.row_flags is not None
    assert torch.equal(r0, r1) and torch.equal(a0, a1)
    n_vis = int((i0["radii"] > 0).sum())
    flags = cs1.row_flags[:n_vis].bool()
    assert 0 < int(flags.sum()) < n_vis
    if dense_scene:
        assert int(flags.sum()) < 0.6 * n_vis, (int(flags.sum()), n_vis)       # an opaque scene hides most of what is in the frustum
    nz0, nz1 = (G0[:n_vis] != 0).any(1), (G1[:n_vis] != 0).any(1)
    assert not bool(nz1[~flags].any()) and not bool(nz0[~flags].any())          # no flag: no gradient, with or without the pass
    assert int((flags & ~nz1).sum()) <= max(2, n_vis // 2000)                   # a flag: a gradient (random cotangents)
    assert float((cs0.rows[:n_vis] - cs1.rows[:n_vis]).abs().max()) <= 2e-4 * float(cs0.rows[:n_vis].abs().max())
    for k in ("means", "scales", "quats", "opacities"):
        assert float((P0[k].grad - P1[k].grad).abs().max()) <= 2e-4 * float(P0[k].grad.abs().max()) + 1e-9, k


def test_geometry_rows_equal_the_dense_geometry_gradients(hip_lib):
    """ColorSource.geometry_rows: the rasterization returns no gradient for means / quats / scales / opacities; the projection
    backward's per-visible rows go through mtgs_node_bwd_rows (exp / normalise / sigmoid VJPs) and reach the optimizer as row
    gradients of the RAW parameters.  Against the autograd path (dense expansion + node_bwd_batch) on two static nodes: the rows,
    expanded, equal the dense gradients; the parameters' .grad stay None; the camera-space normals inside; rigid nodes refused."""
    from mtgs_amd import rasterization
    from mtgs_amd.nodes import collect_gaussians
    from mtgs_amd.optim import FusedAdam
    from mtgs_amd.synthetic import make_camera
    dev = torch.device("cuda")
    W, H, T, t = 320, 200, 2, 1
    g = torch.Generator().manual_seed(41)

    def node(n, multi, seed):
        gg = torch.Generator().manual_seed(seed)
        P = {"means": (torch.rand(n, 3, generator=gg) * 2 - 1) * torch.tensor([8.0, 2.0, 8.0]) + torch.tensor([0.0, 0.0, 6.0]),
             "scales": torch.log(torch.rand(n, 3, generator=gg) * 0.2 + 0.03), "quats": torch.randn(n, 4, generator=gg),
             "opacities": torch.randn(n, 1, generator=gg), "features_dc": torch.randn(n, 3, generator=gg) * 0.7,
             "features_rest": torch.randn(n, T, 15, 3, generator=gg) * 0.2 if multi else torch.randn(n, 15, 3, generator=gg) * 0.2}
        if multi:
            P["features_adapters"] = torch.randn(n, T, 3, generator=gg) * 0.1
        return P
    raw = [node(6000, False, 1), node(5003, True, 2)]
    vm, K = make_camera(W, H)
    vm, K = vm.to(dev), K.to(dev)
    c2w = torch.inverse(vm)[:, :3, :]
    Gc = torch.randn(1, H, W, 7, generator=g).to(dev)
    Ga = torch.randn(1, H, W, 1, generator=g).to(dev)

    def run(geo):
        P = [{k: v.clone().to(dev).requires_grad_(True) for k, v in nd.items()} for nd in raw]
        nodes = [dict(p, traversal_index=t) if "features_adapters" in p else p for p in P]
        gs = collect_gaussians(nodes, c2w, 3, deferred_colors=True)
        cs = gs["color_source"]
        cs.camera_normals = c2w[0].contiguous()
        cs.want_grad_rows = cs.geometry_rows = geo
        r, a, info = rasterization(gs["means"], gs["quats"], gs["scales"], gs["opacities"], None, vm, K, W, H, packed=False,
                                   render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True, color_source=cs)
        torch.autograd.backward([r, a], [Gc, Ga])
        return P, cs, r.detach(), info

    Pd, _, rd, info = run(False)
    Pg, cs, rg, _ = run(True)
    assert torch.equal(rd, rg)
    geo_keys = ("means", "scales", "quats", "opacities")
    for p in Pg:
        assert all(p[k].grad is None for k in geo_keys)            # nothing dense was written
    opt = FusedAdam([{"params": [p[k] for p in Pg for k in p], "lr": 1e-3}], eps=1e-15)
    # the activated tensors of run(True) are released by now (autograd outputs, freed with the graph): their blocks are handed
    # out again and overwritten before apply_to() enqueues mtgs_node_bwd_rows, which must read the RAW parameters only
    n_all = sum(p["means"].shape[0] for p in Pg)
    junk = [torch.full(shape, float("nan"), device=dev) for _ in range(6) for shape in ((n_all, 3), (n_all, 4), (n_all,))]
    cs.apply_to(opt)
    del junk
    vis = info["radii"][0] > 0
    start = 0
    for pd, pg in zip(Pd, Pg):
        n = pd["means"].shape[0]
        for k in geo_keys:
            rows, row_of, col, stride, width = opt._rows[id(pg[k])][:5]
            dense = torch.zeros(n, width, device=dev)
            sel = row_of >= 0
            dense[sel] = rows[row_of[sel].long(), col:col + width]
            ref = pd[k].grad.reshape(n, width)
            assert torch.equal(sel, vis[start:start + n])
            assert float((dense - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-9, k      # (fp32 atomics order in the rows)
        start += n
    opt.step()                                                      # (the optimizer takes them: smoke)
    # a rigid node is refused
    rig = dict(node(500, False, 3))
    Pr = {k: v.clone().to(dev).requires_grad_(True) for k, v in rig.items()}
    Pr["instance_quat"], Pr["instance_trans"] = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev), torch.zeros(3, device=dev)
    gs = collect_gaussians([Pr], c2w, 3, deferred_colors=True)
    cs2 = gs["color_source"]
    cs2.want_grad_rows = cs2.geometry_rows = True
    r, a, _ = rasterization(gs["means"], gs["quats"], gs["scales"], gs["opacities"], None, vm, K, W, H, packed=False,
                            render_mode="RGB+ED", rasterize_mode="antialiased", absgrad=True, color_source=cs2)
    with pytest.raises(NotImplementedError):
        torch.autograd.backward([r, a], [Gc[..., :4].contiguous(), Ga])
