"""mtgs_amd.nodes.camera_space_normals (csrc/normals.hip) against the vectors the reference's own helpers produced
(tests/golden/normals_ref.npz, MTGSSceneModel._get_gaussian_camera_space_normals, mtgs_scene_graph.py:526-545) and
against oracle/normals_oracle.py on seeded inputs; tolerance 2e-6 forward, 2e-5 relative for the gradients (fp32)."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = np.load(Path(__file__).parent / "golden" / "normals_ref.npz")


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_normals_match_reference_vectors(hip_lib, case):
    from mtgs_amd.nodes import camera_space_normals
    g = {k[2:]: torch.from_numpy(GOLD[k]).float().cuda() for k in GOLD.files if k.startswith(case + "_")}
    q = g["quats"].clone().requires_grad_(True)
    n = camera_space_normals(q, g["scales"], g["means"], g["c2w"])
    assert n.shape == g["normals"].shape
    assert float((n.detach() - g["normals"]).abs().max()) < 2e-6
    (n * g["G"]).sum().backward()
    assert float((q.grad - g["g_quats"]).abs().max()) < 2e-5 * float(g["g_quats"].abs().max())


@pytest.mark.parametrize("N", [0, 1, 255, 256, 100_003])
def test_normals_with_rgbs_equal_cat_and_oracle(hip_lib, N):
    """[rgbs | normals] in one launch == torch.cat of the parts; values and gradients against the numpy oracle; the
    cotangent of the rgb columns passes through."""
    from mtgs_amd.nodes import camera_space_normals
    from oracle import normals_oracle as no
    g = torch.Generator().manual_seed(N + 5)
    quats = torch.randn(N, 4, generator=g)
    scales = torch.exp(torch.randn(N, 3, generator=g))
    means = torch.randn(N, 3, generator=g) * 10
    A = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    c2w = torch.cat([A, torch.randn(3, 1, generator=g)], 1)[None]
    rgbs = torch.rand(N, 3, generator=g)
    G = torch.randn(N, 6, generator=g)
    q = quats.cuda().requires_grad_(True)
    r = rgbs.cuda().requires_grad_(True)
    out = camera_space_normals(q, scales.cuda(), means.cuda(), c2w.cuda(), rgbs=r)
    assert out.shape == (N, 6)
    alone = camera_space_normals(quats.cuda(), scales.cuda(), means.cuda(), c2w.cuda())
    assert torch.equal(out[:, :3], rgbs.cuda()) and torch.equal(out[:, 3:], alone)
    (out * G.cuda()).sum().backward()
    ref_n = no.normals_fwd(quats.numpy(), scales.numpy(), means.numpy(), c2w.numpy())
    ref_g = no.normals_bwd(quats.numpy(), scales.numpy(), means.numpy(), c2w.numpy(), G[:, 3:].numpy())
    if N:
        # a flip decided within fp32 rounding of dot == 0 may differ from the fp64 oracle: such rows are excluded
        d = c2w[0, :, 3][None] - means
        world = torch.from_numpy(ref_n).float() @ A.T
        safe = ((world * (d / d.norm(dim=-1, keepdim=True))).sum(-1).abs() > 1e-5).numpy()
        assert safe.mean() > 0.99
        assert np.abs(out[:, 3:].detach().cpu().numpy() - ref_n)[safe].max() < 2e-6
        assert np.abs(q.grad.cpu().numpy() - ref_g)[safe].max() < 2e-5 * max(1.0, np.abs(ref_g).max())
    assert torch.equal(r.grad, G[:, :3].cuda())
