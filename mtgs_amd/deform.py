"""Deformable (non-rigid) object nodes of MTGS: the deformation network and the node's Gaussians.

`DeformableSubModel` (/root/reference/mtgs/scene_model/gaussian_model/deformable_node.py:38-247, config
mtgs/config/MTGS_deformable.py) is a rigid object node whose Gaussians are additionally displaced, rotated and rescaled per
frame by `ConditionalDeformNetwork` (utils.py:286-333): an 8 x 256 ReLU MLP with one skip connection over a frequency
embedding of (position / height, timestamp) and a learnt per-instance code, with three linear heads.

What is hand-written here is what is NOT GEMM-shaped: the embedding (csrc/deform.hip, one launch instead of ~45) and the
node activations behind the network (csrc/node.hip, shared with the rigid nodes).  The linear layers are plain library
GEMMs (torch.nn.functional.linear -> hipBLASLt): a 100 -> 256 -> ... -> 10 fp32 MLP over a few thousand Gaussians is
launch-bound, not MFMA-bound, and a fused MFMA kernel would be paid back only at >~50k Gaussians per object.
State-dict names are the reference's (`deform_network.linear.<i>.weight` ...), so a checkpoint's entries pass through.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from ._lib import call, ptr, require_gpu, stream_of


class _Embed(torch.autograd.Function):
    """[N, 3 + 6 xf + 1 + 2 tf + E] input rows of the network; only `cond` carries a gradient (deformable_node.py:181:
    `local_means.data`; the timestamp is data)."""

    @staticmethod
    def forward(ctx, means, height, t, cond, x_freqs, t_freqs):
        require_gpu(means, cond)
        N, E = means.shape[0], cond.numel()
        width = 3 + 6 * x_freqs + 1 + 2 * t_freqs + E
        out = torch.empty((N, width), dtype=torch.float32, device=means.device)
        m = means.detach().to(torch.float32).contiguous()
        c = cond.detach().to(torch.float32).contiguous()
        call("mtgs_deform_embed", N, ptr(m), float(height), float(t), ptr(c), E, int(x_freqs), int(t_freqs), ptr(out), width,
             stream_of(m))
        ctx.cond_shape, ctx.off = cond.shape, width - E
        return out

    @staticmethod
    def backward(ctx, v_out):
        return None, None, None, v_out[:, ctx.off:].sum(0).reshape(ctx.cond_shape), None, None


def deform_network(means: Tensor, height: float, t: float, cond: Tensor, weights: Mapping[str, Tensor],
                   x_multires: int = 10, t_multires: int = 10) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    """ConditionalDeformNetwork.forward on x = means.data / height * 2 (deformable_node.py:177-203): returns
    (delta_xyz[N,3], delta_quat[N,4] | None, delta_scale[N,3] | None).  `weights`: the module's state dict
    (`linear.<i>.weight/bias`, `gaussian_warp.*`, optional `gaussian_rotation.*`, `gaussian_scaling.*`); depth, width and
    the skip position are read from the shapes.  `cond`: instances_embedding[1, E]."""
    emb = _Embed.apply(means, height, t, cond, x_multires, t_multires)
    in_ch = emb.shape[1]
    D = 0
    while f"linear.{D}.weight" in weights:
        D += 1
    assert D >= 1 and weights["linear.0.weight"].shape[1] == in_ch, (D, weights["linear.0.weight"].shape, in_ch)
    h = emb
    for i in range(D):
        w, b = weights[f"linear.{i}.weight"], weights[f"linear.{i}.bias"]
        if i > 0 and w.shape[1] == h.shape[1] + in_ch:      # the layer behind the skip connection (utils.py:323-324)
            h = torch.cat([emb, h], -1)
        assert w.shape[1] == h.shape[1], (i, w.shape, h.shape)
        h = F.relu(F.linear(h, w, b))
    heads = [("gaussian_warp", 3), ("gaussian_rotation", 4), ("gaussian_scaling", 3)]
    have = [(k, n) for k, n in heads if f"{k}.weight" in weights]
    out = F.linear(h, torch.cat([weights[f"{k}.weight"] for k, _ in have]), torch.cat([weights[f"{k}.bias"] for k, _ in have]))
    parts = dict(zip([k for k, _ in have], torch.split(out, [n for _, n in have], dim=-1)))
    return parts["gaussian_warp"], parts.get("gaussian_rotation"), parts.get("gaussian_scaling")


def deformable_gaussians(params: Mapping[str, Tensor], instance_quat: Tensor, instance_trans: Tensor, camera_to_worlds: Tensor,
                         sh_degree_to_use: int, model_sh_degree: int,
                         deformation: Optional[Tuple[Tensor, Optional[Tensor], Optional[Tensor]]] = None,
                         stop_optimizing_canonical_xyz: bool = True) -> Dict[str, Tensor]:
    """DeformableSubModel.get_gaussians (deformable_node.py:206-247) for the pose `get_object_pose` returned
    (mtgs_amd.nodes.object_pose) and the `deformation` = (delta_xyz, delta_quat, delta_scale) of `deform_network`
    (None before `use_deformgs_after`): local means (+ delta_xyz; the canonical means stop learning when
    stop_optimizing_canonical_xyz) -> global frame; normalised quaternions (+ delta_quat, normalised again) composed with
    the pose; exp(scales) (+ delta_scale); colours from the view directions of the global means."""
    from .nodes import node_gaussians
    means, quats = params["means"], params["quats"]
    d_xyz, d_quat, d_scale = deformation if deformation is not None else (None, None, None)
    if d_xyz is not None:
        means = (means.detach() if stop_optimizing_canonical_xyz else means) + d_xyz
    if d_quat is not None:
        quats = quats / quats.norm(dim=-1, keepdim=True) + d_quat     # (node_gaussians normalises once more, as get_quats does)
    g = node_gaussians(means, params["scales"], quats, params["opacities"], params["features_dc"], params["features_rest"],
                       camera_to_worlds, sh_degree_to_use, model_sh_degree, instance_quat=instance_quat,
                       instance_trans=instance_trans)
    if d_scale is not None:
        g["scales"] = g["scales"] + d_scale
    return g


def deformation_from_state(node: Mapping[str, Tensor], height: float, t: float, prefix: str = "deform_network.") \
        -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    """`deform_network` on a node's checkpoint entries (mtgs_amd.checkpoint.load_gaussian_nodes): `deform_network.*` and
    `instances_embedding`; `height` = instance_size[2], which the reference keeps outside the state dict."""
    weights = {k[len(prefix):]: v for k, v in node.items() if k.startswith(prefix)}
    return deform_network(node["means"], height, t, node["instances_embedding"], weights)
