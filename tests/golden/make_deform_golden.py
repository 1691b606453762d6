"""Golden vectors for the deformation network of MTGS's deformable nodes, produced by the reference's OWN module
(/root/reference/mtgs/scene_model/gaussian_model/utils.py: ConditionalDeformNetwork, get_embedder).  Run in the build
container (the reference is not on the GPU box):  python tests/golden/make_deform_golden.py
Writes tests/golden/deform_ref.npz: inputs, the module's state dict (a small network: W = 32), outputs and autograd
gradients of sum(outputs * G) with respect to every weight and the condition."""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference/mtgs/scene_model/gaussian_model/utils.py")
spec = importlib.util.spec_from_file_location("mtgs_ref_utils", REF)
U = importlib.util.module_from_spec(spec)
spec.loader.exec_module(U)

torch.manual_seed(7)
N, E, W = 128, 16, 32
net = U.ConditionalDeformNetwork(D=8, W=W, input_ch=3, embed_dim=E)
means = (torch.rand(N, 3) - 0.5) * torch.tensor([0.8, 0.6, 1.7])
height, t = 1.7, 0.37
cond = torch.rand(1, E, requires_grad=True)
x = means / torch.full((N,), height)[:, None] * 2                       # deformable_node.py:177-181
d_xyz, d_quat, d_scale = net(x, torch.tensor(t, dtype=torch.float32).repeat(N, 1), cond.repeat(N, 1))
G = {k: torch.randn_like(v) for k, v in (("xyz", d_xyz), ("quat", d_quat), ("scale", d_scale))}
loss = (d_xyz * G["xyz"]).sum() + (d_quat * G["quat"]).sum() + (d_scale * G["scale"]).sum()
loss.backward()
out = {"means": means.numpy(), "height": np.float32(height), "t": np.float32(t), "cond": cond.detach().numpy(),
       "d_xyz": d_xyz.detach().numpy(), "d_quat": d_quat.detach().numpy(), "d_scale": d_scale.detach().numpy(),
       "g_cond": cond.grad.numpy(), "x_emb": net.embed_fn(x).detach().numpy(),
       "t_emb": net.embed_time_fn(torch.tensor([[t]], dtype=torch.float32)).numpy()}
for k, v in G.items():
    out[f"G_{k}"] = v.numpy()
for k, v in net.state_dict().items():
    out[f"w.{k}"] = v.numpy()
for k, p in net.named_parameters():
    out[f"g.{k}"] = p.grad.numpy()
dst = Path(__file__).with_name("deform_ref.npz")
np.savez_compressed(dst, **out)
print(dst, {k: v.shape for k, v in out.items() if k.startswith(("d_", "x_", "t_"))}, dst.stat().st_size)
