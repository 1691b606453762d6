"""mtgs_amd.loss.masked_ssim (csrc/loss.hip): (1) against the golden vectors produced by the reference's own
MaskedSSIM module (tests/golden/ssim_ref.npz), (2) against the pinned oracle at MTGS's training resolution.
Tolerances: value 5e-6 absolute, gradient 2e-5 of its max (fp32 kernel vs the reference's fp64 run)."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
Z = np.load(Path(__file__).parent / "golden" / "ssim_ref.npz")
CASES = sorted({k.split("_")[0] for k in Z.files})


@pytest.mark.parametrize("case", CASES)
def test_masked_ssim_reference_vectors(hip_lib, case):
    from mtgs_amd.loss import masked_ssim
    dev = torch.device("cuda")
    gt = torch.from_numpy(Z[f"{case}_gt"]).to(dev)
    pred = torch.from_numpy(Z[f"{case}_pred"]).to(dev).requires_grad_(True)
    mask = Z[f"{case}_mask"]
    mask = None if mask.size == 0 else torch.from_numpy(mask).to(dev)
    val = masked_ssim(gt, pred, mask)
    (1.0 - val).backward()                       # the loss MTGS forms (mtgs_scene_graph.py:831)
    assert abs(float(val.detach()) - float(Z[f"{case}_ssim_f64"])) <= 5e-6
    ref = -Z[f"{case}_grad_f64"]
    err = np.abs(pred.grad.cpu().numpy().astype(np.float64) - ref).max()
    assert err <= 2e-5 * np.abs(ref).max(), err
    with torch.no_grad():                        # forward-only path (metrics: mtgs_scene_graph.py:777)
        assert abs(float(masked_ssim(gt, pred.detach(), mask)) - float(Z[f"{case}_ssim_f64"])) <= 5e-6


def test_masked_ssim_training_resolution(hip_lib):
    from mtgs_amd.loss import masked_ssim
    from oracle import ssim_oracle
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    H, W = 540, 960
    gt = torch.rand(H, W, 3, generator=g)
    pred = (gt + 0.2 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
    mask = torch.rand(H, W, 1, generator=g) > 0.2
    val_ref, grad_ref = ssim_oracle.masked_ssim(gt.numpy(), pred.numpy(), mask.numpy(), with_grad=True)
    p = pred.to(dev).requires_grad_(True)
    val = masked_ssim(gt.to(dev), p, mask.to(dev))
    val.backward()
    assert abs(float(val.detach()) - val_ref) <= 5e-6
    assert np.abs(p.grad.cpu().numpy() - grad_ref).max() <= 2e-5 * np.abs(grad_ref).max()
    v2 = masked_ssim(gt.to(dev), p.detach(), mask.to(dev))
    assert float(v2) == float(val.detach())               # fixed summation order: bit-identical between runs


@pytest.mark.parametrize("use_mask", [True, False])
@pytest.mark.parametrize("ch", [3, 1, 4])
def test_masked_l1_matches_torch_formulation(hip_lib, use_mask, ch):
    """mtgs_scene_graph.py:823: torch.abs(gt_img - pred)[combined_mask.squeeze(-1)].mean(); one channel: the depth terms
    (:881-883, mask [H,W,1] indexing [H,W,1] images), three: the normal term (:934)."""
    from mtgs_amd.loss import masked_l1
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    H, W = 211, 333
    gt = torch.rand(H, W, ch, generator=g).to(dev)
    pred0 = torch.rand(H, W, ch, generator=g)
    pred0[5, 7] = gt[5, 7].cpu()                                    # exact ties: sign(0) = 0
    mask = (torch.rand(H, W, 1, generator=g) > 0.4).to(dev) if use_mask else None
    p_ref = pred0.to(dev).double().requires_grad_(True)
    d = torch.abs(gt.double() - p_ref)
    ref = d[mask.squeeze(-1)].mean() if use_mask else d.mean()
    (0.8 * ref).backward()
    p = pred0.to(dev).requires_grad_(True)
    val = masked_l1(gt, p, mask)
    (0.8 * val).backward()
    assert abs(float(val.detach()) - float(ref.detach())) <= 2e-6
    assert torch.allclose(p.grad.double(), p_ref.grad, rtol=1e-5, atol=1e-12)
