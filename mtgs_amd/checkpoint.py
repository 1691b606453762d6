"""Reading MTGS checkpoints into the tensors the rasterization path consumes.

MTGS's trainer writes `step-%09d.ckpt` = {"step", "pipeline": state_dict, ...}
(/root/reference/mtgs/scene_model/custom_trainer.py:148-157); the Gaussian nodes live under
`_model.gaussian_models.<node>.gauss_params.<name>` (MTGSSceneModel.load_state_dict splits the keys at the first dot
after `gaussian_models.`: /root/reference/mtgs/scene_model/mtgs_scene_graph.py:1185-1203) with
<name> in {means, scales, quats, features_dc, features_rest, opacities} (+ features_adapters for multi-colour nodes;
vanilla_gaussian_splatting.py:174-213, multi_color_gaussian_splatting.py:48-71).

`load_gaussian_nodes` returns the raw parameters per node; `collect_gaussians` does what
MTGSSceneModel.get_gaussians does for the static node types (vanilla / multi-colour: activations through the fused
node kernels, then one concatenation) so that a released checkpoint can be rendered with `rasterization`; rigid nodes are posed
with `frame_idx` or a timestamp (rigid_node.py:127-166; in-frame masks and traversal gating are the caller's: pass
`node_names`), Fourier colours are evaluated (rigid_node.py:217-229), and deformable nodes (deformable_node.py) run their
deformation network (mtgs_amd.deform) when the caller supplies what the reference keeps outside the state dict: the
instance heights.  Anything else that carries unknown per-frame state is refused by name instead of rendered wrongly.
"""
from __future__ import annotations

from typing import Dict, Iterable, Mapping, Optional, Union

import torch
from torch import Tensor

GAUSS_PARAM_NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities", "features_adapters")
_PREFIXES = ("_model.gaussian_models.", "gaussian_models.")


def load_gaussian_nodes(ckpt: Union[str, Mapping], map_location="cpu", trusted: bool = False) -> Dict[str, Dict[str, Tensor]]:
    """{node name: {parameter or buffer name: tensor}} from a checkpoint path, a checkpoint dict or a state dict.
    `gauss_params.<name>` entries are returned under <name>; anything else a node stores (instance_quats,
    instance_trans, deformation networks ...) keeps its full sub-key.

    A path is read with `torch.load(weights_only=True)` (tensors and plain containers only -- what the trainer's
    checkpoints hold).  `trusted=True` allows the full unpickler for checkpoints that carry other Python objects:
    that executes code from the file, so use it only on files you produced."""
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location=map_location, weights_only=not trusted)
    state = ckpt.get("pipeline", ckpt) if isinstance(ckpt, Mapping) else ckpt
    nodes: Dict[str, Dict[str, Tensor]] = {}
    for key, value in state.items():
        for pre in _PREFIXES:
            if key.startswith(pre):
                node, sub = key[len(pre):].split(".", maxsplit=1)       # mtgs_scene_graph.py:1190-1191
                if sub.startswith("gauss_params."):
                    sub = sub[len("gauss_params."):]
                nodes.setdefault(node, {})[sub] = value
                break
    if not nodes:
        raise ValueError("no `gaussian_models.<node>.` entries: not an MTGS checkpoint / state dict")
    return nodes


RIGID_KEYS = ("instance_quats", "instance_trans")


def node_kind(params: Mapping[str, Tensor]) -> str:
    """'vanilla' | 'multicolor' | 'rigid' (instance poses only, rigid_node.py:85-112) | 'deformable' (poses + instance
    embedding + `deform_network.*`, deformable_node.py:85-94) | 'dynamic' (anything else)."""
    extra = [k for k in params if k not in GAUSS_PARAM_NAMES]
    if extra:   # (a rigid node's features_dc is [N,3], or [N,F,3] with Fourier features: rigid_node.py:217-229)
        if set(extra) == set(RIGID_KEYS):
            return "rigid"
        rest = [k for k in extra if k not in RIGID_KEYS and k != "instances_embedding" and not k.startswith("deform_network.")]
        if not rest and "instances_embedding" in params and "deform_network.linear.0.weight" in params:
            return "deformable"
        return "dynamic"
    return "multicolor" if "features_adapters" in params else "vanilla"


def collect_gaussians(nodes: Mapping[str, Mapping[str, Tensor]], camera_to_worlds: Tensor, sh_degree_to_use: int,
                      model_sh_degree: int = 3, traversal_index: Optional[int] = None,
                      node_names: Optional[Iterable[str]] = None, device="cuda",
                      frame_idx: Optional[int] = None, timestamp: Optional[float] = None,
                      frame_timestamps: Optional[Tensor] = None, fourier: Optional[Mapping] = None,
                      instance_heights: Optional[Mapping[str, float]] = None, deform_time: Optional[float] = None,
                      undeformed: bool = False) -> Dict[str, Tensor]:
    """means / scales / quats / opacities / rgbs / model_id of the listed nodes, activated by
    mtgs_amd.nodes.node_gaussians and concatenated in order (MTGSSceneModel.get_gaussians,
    mtgs_scene_graph.py:408-461).  Multi-colour nodes need `traversal_index` (get_pertravel_features,
    multi_color_gaussian_splatting.py:77-86; None = the shared colour only, as eval_mode 'null').
    Rigid nodes are posed for `frame_idx`, or BETWEEN frames for `timestamp` + `frame_timestamps`
    (RigidSubModel.get_object_pose, rigid_node.py:127-166; objects that are not in the frame are left out, as
    get_gaussians returns None for them).  Rigid nodes with Fourier colours (features_dc[N,F,3]) need
    `fourier = {"x": normalised timestamp (temporal) | None (spatial: the camera-object yaw is computed), "scale": ..,
    "space": "temporal" | "spatial"}` -- the node's portable_config (rigid_node.py:114-125).
    Deformable nodes (deformable_node.py:206-247) are posed like rigid ones and displaced by their deformation network, which
    needs `deform_time` (the frame's timestamp as get_deformation uses it; with timestamp interpolation the reference feeds
    the interpolation FRACTION, deformable_node.py:190) and `instance_heights[name]` (instance_size[2], kept outside the state
    dict).  The reference applies the network whenever step > use_deformgs_after (deformable_node.py:230-232), so a trained
    checkpoint rendered without them would be rendered WRONGLY: that raises unless `undeformed=True` asks for the canonical
    geometry explicitly."""
    from .deform import deformation_from_state
    from .nodes import cam_obj_yaw, collect_gaussians as _collect, fourier_features_dc, object_pose
    names = list(nodes.keys()) if node_names is None else list(node_names)
    specs, scale_deltas, start = [], [], 0
    for name in names:
        p = {k: v.to(device) for k, v in nodes[name].items()}
        kind = node_kind(p)
        if kind in ("rigid", "deformable"):
            # RigidSubModel.get_object_pose (rigid_node.py:127-144): static objects store one pose, moving ones one per frame
            iq, it = p.pop("instance_quats"), p.pop("instance_trans")
            if it.dim() > 1:
                if frame_idx is None and timestamp is None:
                    raise ValueError(f"collect_gaussians: rigid node {name!r} has per-frame poses: pass frame_idx (or timestamp)")
                iq, it = object_pose(iq, it, frame_idx=frame_idx, timestamp=timestamp, frame_timestamps=frame_timestamps)
                if iq is None:
                    continue                      # not in this frame
            p["instance_quat"], p["instance_trans"] = iq.contiguous(), it.contiguous()
            if p["features_dc"].dim() == 3:       # Fourier-series colour (rigid_node.py:217-229)
                if fourier is None:
                    raise ValueError(f"collect_gaussians: rigid node {name!r} has Fourier features_dc {tuple(p['features_dc'].shape)}: "
                                     "pass fourier={'x', 'scale', 'space'}")
                space = fourier.get("space", "temporal")
                x = fourier.get("x")
                if space == "spatial":
                    x = cam_obj_yaw(camera_to_worlds.to(device), iq)
                p["features_dc"] = fourier_features_dc(p["features_dc"], x, fourier.get("scale", 1.0), space)
        if kind == "deformable":
            state = {k: p.pop(k) for k in list(p) if k == "instances_embedding" or k.startswith("deform_network.")}
            have = deform_time is not None and instance_heights is not None and name in instance_heights
            if not have and not undeformed:
                raise ValueError(f"collect_gaussians: deformable node {name!r} needs deform_time and instance_heights[{name!r}] "
                                 "(pass undeformed=True to render its canonical geometry)")
            if have:
                state["means"] = p["means"]
                d_xyz, d_quat, d_scale = deformation_from_state(state, float(instance_heights[name]), float(deform_time))
                p["means"] = p["means"].detach() + d_xyz                      # stop_optimizing_canonical_xyz (default)
                if d_quat is not None:
                    p["quats"] = p["quats"] / p["quats"].norm(dim=-1, keepdim=True) + d_quat
                if d_scale is not None:
                    scale_deltas.append((start, d_scale))
        if kind == "dynamic":
            raise NotImplementedError(f"collect_gaussians: node {name!r} carries per-frame state "
                                      f"({sorted(k for k in p if k not in GAUSS_PARAM_NAMES)[:3]}...); only vanilla, "
                                      "multi-colour, rigid and deformable nodes are supported")
        if p["scales"].shape[-1] == 1:   # isotropic nodes store one log-scale (vanilla_gaussian_splatting.py:185-196)
            p["scales"] = p["scales"].expand(-1, 3).contiguous()
        if "quats" not in p:
            p["quats"] = torch.tensor([1.0, 0.0, 0.0, 0.0], device=device).expand(p["means"].shape[0], 4).contiguous()
        if kind == "multicolor":
            if traversal_index is None:   # eval_mode "null": the shared colour only (multi_color_gaussian_splatting.py:84-86)
                p.pop("features_adapters")
                if p["features_rest"].dim() == 4:
                    p["features_rest"] = torch.zeros_like(p["features_rest"][:, 0])
            else:
                p["traversal_index"] = traversal_index
        specs.append(p)
        start += p["means"].shape[0]
    out = _collect(specs, camera_to_worlds.to(device), sh_degree_to_use, model_sh_degree)
    if scale_deltas:                              # get_scales: exp(scales) + delta_scale (deformable_node.py:115-119)
        parts, at = [], 0
        for s0, d in scale_deltas:
            parts += [out["scales"][at:s0], out["scales"][s0:s0 + d.shape[0]] + d]
            at = s0 + d.shape[0]
        out["scales"] = torch.cat(parts + [out["scales"][at:]])
    return out
