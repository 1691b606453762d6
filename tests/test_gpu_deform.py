"""Deformable object nodes (mtgs_amd.deform, csrc/deform.hip): the deformation network against vectors produced by the
reference's own ConditionalDeformNetwork (tests/golden/deform_ref.npz), and the node composition against
DeformableSubModel.get_gaussians restated in fp64 (/root/reference/mtgs/scene_model/gaussian_model/deformable_node.py:206-247)."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden" / "deform_ref.npz"


def test_deform_network_matches_reference_vectors(hip_lib):
    from mtgs_amd.deform import deform_network
    from tests.util import REPORT
    dev = torch.device("cuda")
    g = np.load(GOLD)
    W = {k[2:]: torch.from_numpy(g[k]).to(dev).requires_grad_(True) for k in g.files if k.startswith("w.")}
    cond = torch.from_numpy(g["cond"]).to(dev).requires_grad_(True)
    means = torch.from_numpy(g["means"]).to(dev)
    d_xyz, d_quat, d_scale = deform_network(means, float(g["height"]), float(g["t"]), cond, W)
    worst = 0.0
    for got, key in ((d_xyz, "d_xyz"), (d_quat, "d_quat"), (d_scale, "d_scale")):
        ref = g[key]
        err = np.abs(got.detach().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        worst = max(worst, err)
        assert got.shape == ref.shape and err < 1e-4, (key, err)          # sin / cos of arguments up to 512 |x| in fp32
    loss = sum((o * torch.from_numpy(g[f"G_{k}"]).to(dev)).sum() for o, k in ((d_xyz, "xyz"), (d_quat, "quat"), (d_scale, "scale")))
    loss.backward()
    gw = 0.0
    for k, p in W.items():
        ref = g[f"g.{k}"]
        err = np.abs(p.grad.cpu().numpy() - ref).max() / max(1e-6, np.abs(ref).max())
        gw = max(gw, err)
        assert err < 2e-3, (k, err)
    ref = g["g_cond"]
    gc = np.abs(cond.grad.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert gc < 2e-3, gc
    REPORT.append({"kind": "neighbour", "name": "deform_network vs the reference module's vectors", "forward_rel_max": float(worst),
                   "weight_grad_rel_max": float(gw), "cond_grad_rel_max": float(gc)})


def _chain(P, q, t, cam_pos, n, deform, stop_xyz):
    """DeformableSubModel.get_gaussians, fp64 on CPU (quat_to_rotmat / quat_mult as utils.py defines them)."""
    from oracle import torch_ref
    d_xyz, d_quat, d_scale = deform
    local = (P["means"].detach() if stop_xyz else P["means"]) + d_xyz
    w, x, y, z = q
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)]),
                     torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)]),
                     torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)])])
    means = local @ R.T + t
    ql = P["quats"] / P["quats"].norm(dim=-1, keepdim=True) + d_quat
    ql = ql / ql.norm(dim=-1, keepdim=True)
    w2, x2, y2, z2 = ql.unbind(-1)
    quats = torch.stack([w * w2 - x * x2 - y * y2 - z * z2, w * x2 + x * w2 + y * z2 - z * y2,
                         w * y2 - x * z2 + y * w2 + z * x2, w * z2 + x * y2 - y * x2 + z * w2], -1)
    d = means.detach() - cam_pos
    d = d / d.norm(dim=-1, keepdim=True)
    colors = torch.cat((P["features_dc"][:, None, :], P["features_rest"]), dim=1)
    return {"means": means, "scales": torch.exp(P["scales"]) + d_scale, "quats": quats,
            "opacities": torch.sigmoid(P["opacities"]).squeeze(-1),
            "rgbs": torch.clamp(torch_ref.spherical_harmonics(n, d, colors) + 0.5, 0.0, 1.0)}


@pytest.mark.parametrize("stop_xyz", [True, False])
def test_deformable_gaussians_match_the_reference_chain(hip_lib, stop_xyz):
    from mtgs_amd.deform import deformable_gaussians
    dev = torch.device("cuda")
    N = 2500
    g = torch.Generator().manual_seed(11)
    P = {"means": torch.randn(N, 3, generator=g), "scales": torch.randn(N, 3, generator=g) - 2, "quats": torch.randn(N, 4, generator=g),
         "opacities": torch.randn(N, 1, generator=g), "features_dc": torch.randn(N, 3, generator=g) * 0.7,
         "features_rest": torch.randn(N, 15, 3, generator=g) * 0.2}
    D = [torch.randn(N, 3, generator=g) * 0.05, torch.randn(N, 4, generator=g) * 0.05, torch.randn(N, 3, generator=g) * 0.01]
    q = torch.randn(4, generator=g); q = q / q.norm()
    t = torch.randn(3, generator=g) * 4
    cam = torch.tensor([1.0, -2.0, 0.5])
    G = {k: torch.randn(N, c, generator=g) for k, c in (("means", 3), ("scales", 3), ("quats", 4), ("rgbs", 3))}
    G["opacities"] = torch.randn(N, generator=g)

    def run(f64):
        cast = (lambda v: v.double()) if f64 else (lambda v: v.to(dev))
        p = {k: cast(v).requires_grad_(True) for k, v in P.items()}
        d = [cast(v).requires_grad_(True) for v in D]
        qq, tt = cast(q).requires_grad_(True), cast(t).requires_grad_(True)
        if f64:
            out = _chain(p, qq, tt, cam.double(), 3, d, stop_xyz)
        else:
            c2w = torch.eye(4, device=dev)[None, :3].clone(); c2w[0, :, 3] = cam.to(dev)
            out = deformable_gaussians(p, qq, tt, c2w, 3, 3, deformation=tuple(d), stop_optimizing_canonical_xyz=stop_xyz)
        sum((out[k] * cast(G[k])).sum() for k in G).backward()
        leaves = dict(p); leaves.update(d_xyz=d[0], d_quat=d[1], d_scale=d[2], pose_q=qq, pose_t=tt)
        return out, leaves

    got, gl = run(False)
    ref, rl = run(True)
    for k in ref:
        assert (got[k].detach().cpu().double() - ref[k].detach()).abs().max() < 5e-6, k
    for k in rl:
        if k == "means" and stop_xyz:
            assert gl[k].grad is None or float(gl[k].grad.abs().max()) == 0.0      # the canonical means stop learning
            continue
        r = rl[k].grad
        err = float((gl[k].grad.cpu().double() - r).abs().max())
        assert err <= 3e-5 * max(1.0, float(r.abs().max())), (k, err)


def test_deformation_from_checkpoint_entries(hip_lib):
    """load_gaussian_nodes keeps `deform_network.*` / `instances_embedding` under their names; deformation_from_state feeds
    them to the network (production size: 8 x 256)."""
    from mtgs_amd.checkpoint import load_gaussian_nodes
    from mtgs_amd.deform import deform_network, deformation_from_state
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    N, E, Wd, in_ch = 1000, 16, 256, 63 + 21 + 16
    sd = {"_model.gaussian_models.ped_0.gauss_params.means": torch.randn(N, 3, generator=g) * 0.4,
          "_model.gaussian_models.ped_0.instances_embedding": torch.rand(1, E, generator=g)}
    for i in range(8):
        fan = in_ch if i == 0 else (Wd + in_ch if i == 5 else Wd)
        sd[f"_model.gaussian_models.ped_0.deform_network.linear.{i}.weight"] = torch.randn(Wd, fan, generator=g) / fan ** 0.5
        sd[f"_model.gaussian_models.ped_0.deform_network.linear.{i}.bias"] = torch.randn(Wd, generator=g) * 0.01
    for k, n in (("gaussian_warp", 3), ("gaussian_rotation", 4), ("gaussian_scaling", 3)):
        sd[f"_model.gaussian_models.ped_0.deform_network.{k}.weight"] = torch.randn(n, Wd, generator=g) / 16
        sd[f"_model.gaussian_models.ped_0.deform_network.{k}.bias"] = torch.zeros(n)
    node = {k: v.to(dev) for k, v in load_gaussian_nodes(sd)["ped_0"].items()}
    d_xyz, d_quat, d_scale = deformation_from_state(node, height=1.7, t=0.25)
    assert d_xyz.shape == (N, 3) and d_quat.shape == (N, 4) and d_scale.shape == (N, 3)
    from oracle import deform_oracle as O
    w = {k[len("deform_network."):]: v.cpu().numpy() for k, v in node.items() if k.startswith("deform_network.")}
    r_xyz, r_quat, r_scale = O.deform_network(node["means"].cpu().numpy(), 1.7, 0.25, node["instances_embedding"].cpu().numpy(), w)
    for got, ref in ((d_xyz, r_xyz), (d_quat, r_quat), (d_scale, r_scale)):
        assert np.abs(got.cpu().numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())


def test_collect_gaussians_poses_and_deforms_deformable_nodes(hip_lib):
    """mtgs_amd.checkpoint.collect_gaussians on a state dict with a deformable node = deformable_gaussians on its entries."""
    from mtgs_amd import checkpoint as ck
    from mtgs_amd.deform import deformable_gaussians, deformation_from_state
    from mtgs_amd.nodes import object_pose
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    sd = {}

    def gauss(name, n):
        b = f"_model.gaussian_models.{name}.gauss_params."
        sd[b + "means"] = torch.randn(n, 3, generator=g) * 0.5
        sd[b + "scales"] = torch.randn(n, 3, generator=g) * 0.3 - 2.0
        sd[b + "quats"] = torch.randn(n, 4, generator=g)
        sd[b + "opacities"] = torch.randn(n, 1, generator=g)
        sd[b + "features_dc"] = torch.randn(n, 3, generator=g)
        sd[b + "features_rest"] = torch.randn(n, 15, 3, generator=g) * 0.1

    gauss("road", 400)
    gauss("ped_7", 333)
    b = "_model.gaussian_models.ped_7."
    sd[b + "instance_quats"], sd[b + "instance_trans"] = torch.randn(6, 4, generator=g), torch.randn(6, 3, generator=g) * 3
    sd[b + "instances_embedding"] = torch.rand(1, 16, generator=g)
    Wd, in_ch = 32, 100
    for i in range(8):
        fan = in_ch if i == 0 else (Wd + in_ch if i == 5 else Wd)
        sd[b + f"deform_network.linear.{i}.weight"] = torch.randn(Wd, fan, generator=g) / fan ** 0.5
        sd[b + f"deform_network.linear.{i}.bias"] = torch.randn(Wd, generator=g) * 0.01
    for k, n in (("gaussian_warp", 3), ("gaussian_rotation", 4), ("gaussian_scaling", 3)):
        sd[b + f"deform_network.{k}.weight"] = torch.randn(n, Wd, generator=g) / 8
        sd[b + f"deform_network.{k}.bias"] = torch.zeros(n)
    nodes = ck.load_gaussian_nodes(sd)
    assert ck.node_kind(nodes["ped_7"]) == "deformable"
    c2w = torch.eye(4)[None, :3]
    gs = ck.collect_gaussians(nodes, c2w, 3, frame_idx=4, instance_heights={"ped_7": 1.8}, deform_time=0.4)
    p = {k: v.to(dev) for k, v in nodes["ped_7"].items()}
    q, t = object_pose(p["instance_quats"], p["instance_trans"], frame_idx=4)
    ref = deformable_gaussians(p, q.contiguous(), t.contiguous(), c2w.to(dev), 3, 3, deformation=deformation_from_state(p, 1.8, 0.4))
    for k in ("means", "scales", "quats", "opacities", "rgbs"):
        assert torch.allclose(gs[k][400:], ref[k], atol=1e-6), k
    with pytest.raises(ValueError):     # a trained deformable node without its deformation inputs is refused, not rendered wrongly
        ck.collect_gaussians(nodes, c2w, 3, frame_idx=4)
    plain = ck.collect_gaussians(nodes, c2w, 3, frame_idx=4, undeformed=True)   # before use_deformgs_after: posed, not deformed
    assert not torch.allclose(plain["means"][400:], gs["means"][400:]) and torch.equal(plain["means"][:400], gs["means"][:400])
