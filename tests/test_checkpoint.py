"""mtgs_amd.checkpoint: key layout of MTGS checkpoints (custom_trainer.py:148-157, mtgs_scene_graph.py:1185-1203)."""
import pytest
import torch


def _state(T=2):
    g = torch.Generator().manual_seed(0)
    sd = {"_model.camera_optimizer.pose_adjustment": torch.zeros(4, 6)}
    def node(name, n, multicolor=False, dynamic=False):
        base = f"_model.gaussian_models.{name}."
        sd[base + "gauss_params.means"] = torch.randn(n, 3, generator=g) * 3 + torch.tensor([0.0, 0.0, 8.0])
        sd[base + "gauss_params.scales"] = torch.randn(n, 3, generator=g) * 0.3 - 2.0
        sd[base + "gauss_params.quats"] = torch.randn(n, 4, generator=g)
        sd[base + "gauss_params.opacities"] = torch.randn(n, 1, generator=g)
        sd[base + "gauss_params.features_dc"] = torch.randn(n, 3, generator=g)
        if multicolor:
            sd[base + "gauss_params.features_rest"] = torch.randn(n, T, 15, 3, generator=g) * 0.1
            sd[base + "gauss_params.features_adapters"] = torch.randn(n, T, 3, generator=g) * 0.1
        else:
            sd[base + "gauss_params.features_rest"] = torch.randn(n, 15, 3, generator=g) * 0.1
        if dynamic:
            sd[base + "instance_quats"] = torch.randn(5, 4, generator=g)
            sd[base + "instance_trans"] = torch.randn(5, 3, generator=g)
    node("background", 300, multicolor=True)
    node("road", 200)
    node("object_vehicle_12", 50, dynamic=True)
    return sd


def test_load_nodes_from_checkpoint_file(tmp_path):
    from mtgs_amd import checkpoint as ck
    path = tmp_path / "step-000030000.ckpt"
    torch.save({"step": 30000, "pipeline": _state()}, path)
    nodes = ck.load_gaussian_nodes(str(path))
    assert list(nodes) == ["background", "road", "object_vehicle_12"]
    assert set(nodes["road"]) == {"means", "scales", "quats", "opacities", "features_dc", "features_rest"}
    assert nodes["background"]["features_rest"].shape == (300, 2, 15, 3)
    assert [ck.node_kind(nodes[n]) for n in nodes] == ["multicolor", "vanilla", "dynamic"]
    assert "instance_trans" in nodes["object_vehicle_12"]
    with pytest.raises(ValueError):
        ck.load_gaussian_nodes({"pipeline": {"_model.foo": torch.zeros(1)}})


@pytest.mark.gpu
def test_collect_and_render_checkpoint(hip_lib):
    from mtgs_amd import checkpoint as ck, rasterization
    from mtgs_amd.synthetic import make_camera
    nodes = ck.load_gaussian_nodes({"pipeline": _state()})
    c2w = torch.eye(4)[None, :3]
    with pytest.raises(NotImplementedError, match="object_vehicle_12"):
        ck.collect_gaussians(nodes, c2w, 3)
    gs = ck.collect_gaussians(nodes, c2w, 3, traversal_index=1, node_names=["background", "road"])
    assert gs["means"].shape == (500, 3) and gs["rgbs"].shape == (500, 3) and int(gs["model_id"].max()) == 1
    # the multi-colour node used traversal 1: features_dc + adapters[:, 1], features_rest[:, 1]
    from mtgs_amd.nodes import node_gaussians
    p = {k: v.cuda() for k, v in nodes["background"].items()}
    ref = node_gaussians(p["means"], p["scales"], p["quats"], p["opacities"],
                         (p["features_dc"] + p["features_adapters"][:, 1]).contiguous(),
                         p["features_rest"][:, 1].contiguous(), c2w.cuda(), 3, 3)
    assert torch.equal(gs["rgbs"][:300], ref["rgbs"])
    vm, K = make_camera(160, 120)
    render, alpha, info = rasterization(gs["means"], gs["quats"], gs["scales"], gs["opacities"], gs["rgbs"], vm.cuda(),
                                        K.cuda(), 160, 120, packed=False, render_mode="RGB+ED")
    assert render.shape == (1, 120, 160, 4) and torch.isfinite(render).all() and float(alpha.max()) > 0.1
