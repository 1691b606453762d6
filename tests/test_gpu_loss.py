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


def test_combine_losses_matches_the_written_out_sum(hip_lib):
    """combine_losses = the weighted sum of the loss dictionary with MTGS's finite-check on the normal term
    (mtgs_scene_graph.py:939), value and gradients, with the guarded term finite, NaN and inf."""
    from mtgs_amd.loss import combine_losses
    dev = torch.device("cuda")
    w = [0.8, -0.2, 0.5, 0.1, 0.1]
    for bad in (None, float("nan"), float("inf")):
        vals = [0.31, 0.87, 0.044, 0.52 if bad is None else bad, 0.09]
        ta = [torch.tensor(v, device=dev, requires_grad=True) for v in vals]
        tb = [torch.tensor(v, device=dev, dtype=torch.float64, requires_grad=True) for v in vals]
        got = combine_losses(ta, w, constant=0.2, drop_if_not_finite=(3,))
        n_term = tb[3] if bad is None else torch.zeros((), dtype=torch.float64, device=dev)
        ref = 0.8 * tb[0] + 0.2 * (1 - tb[1]) + 0.5 * tb[2] + 0.1 * n_term + 0.1 * tb[4]
        (3.0 * got).backward()
        (3.0 * ref).backward()
        assert abs(float(got) - float(ref)) <= 1e-6
        for i, (a, b) in enumerate(zip(ta, tb)):
            want = 0.0 if (i == 3 and bad is not None) else float(b.grad)
            assert abs(float(a.grad) - want) <= 1e-6, (bad, i)


@pytest.mark.parametrize("use_mask,empty", [(True, False), (False, False), (True, True)])
def test_inverse_depth_l1_matches_the_reference_formulation(hip_lib, use_mask, empty):
    """mtgs_scene_graph.py:849-858, 875-879 (lidar depth, DepthLossType.InverseL1) written out in torch float64: the range mask
    combined with the image mask, |1 / (gt + 1e-5) - 1 / (pred + 1e-5)| over it, 0 for an empty mask; value, gradient with
    respect to the predicted depth, and the mask by-product (the NCC term's mask, :891)."""
    from mtgs_amd.loss import inverse_depth_l1
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(4)
    H, W = 135, 241
    gt = (torch.rand(H, W, 1, generator=g) * 100.0)
    gt[torch.rand(H, W, 1, generator=g) < 0.3] = 0.0             # lidar: most pixels have no return
    if empty:
        gt = gt * 0.0
    pred0 = torch.rand(H, W, 1, generator=g) * 60.0 + 0.5
    pred0[3, 4] = gt[3, 4]                                        # an exact tie inside the range: sign(0) = 0
    gt = gt.to(dev)
    mask = (torch.rand(H, W, 1, generator=g) > 0.3).to(dev) if use_mask else None
    m_ref = (gt > 0.1) & (gt < 80)
    if use_mask:
        m_ref = m_ref & mask
    p_ref = pred0.to(dev).double().requires_grad_(True)
    if int(m_ref.sum()) == 0:
        ref = torch.zeros((), dtype=torch.float64, device=dev)
    else:
        ref = torch.abs(1 / (gt.double() + 1e-5) - 1 / (p_ref + 1e-5))[m_ref].mean()
        (0.5 * ref).backward()
    p = pred0.to(dev).requires_grad_(True)
    val, m = inverse_depth_l1(p, gt, mask)
    (0.5 * val).backward()
    assert m.dtype == torch.bool and m.shape == (H, W, 1) and torch.equal(m, m_ref)
    assert abs(float(val) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    want = p_ref.grad if p_ref.grad is not None else torch.zeros_like(p_ref)
    assert torch.allclose(p.grad.double(), want, rtol=2e-5, atol=1e-12)


@pytest.mark.parametrize("D,with_exposure,with_depth,normal_ch", [(8, True, True, 3), (4, True, True, -1), (3, False, False, -1),
                                                                  (7, False, True, 3)])
def test_output_head_matches_the_reference_formulation(hip_lib, D, with_exposure, with_depth, normal_ch):
    """mtgs_amd.loss.output_head against mtgs_scene_graph.py:672-690 + LearnableExposureRGBModel.forward
    (module/appearance.py:73-87) written out in torch float64: values to 2e-6, every gradient (render, alpha, background,
    exposure) to 2e-5 relative.  Values are kept away from the clamp edges by construction of the cotangent test only where
    the reference itself is continuous; exact edge hits are covered by the inclusive-mask rule (0 and 1 pass)."""
    from mtgs_amd.loss import output_head
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(D * 7 + normal_ch)
    H, W = 97, 131
    render = torch.rand(1, H, W, D, generator=g) * 1.4 - 0.2
    if with_depth:
        render[..., -1] = torch.rand(1, H, W, generator=g) * 30
    alpha = torch.rand(1, H, W, 1, generator=g)
    alpha[0, :5] = 0.0                                             # nothing hit: depth takes the maximum
    alpha[0, 7, 9] = 0.5                                           # exactly representable: lands ON the clamp edges in fp32 and fp64
    render[0, 7, 9, :3] = torch.tensor([0.0, 1.0, 0.5]) - 0.125
    bg = torch.full((3,), 0.25)
    E = (torch.eye(3, 4) + 0.1 * torch.randn(3, 4, generator=g)) if with_exposure else None
    cots = [torch.randn(H, W, 3, generator=g), torch.randn(H, W, 3, generator=g), torch.randn(H, W, 1, generator=g),
            torch.randn(H, W, 3, generator=g)]

    def reference(render, alpha, bg, E):
        rgb = torch.clamp(render[..., :3] + (1 - alpha) * bg, 0.0, 1.0).squeeze(0)
        app = torch.clamp(rgb.matmul(E[:3, :3]) + E[None, None, :3, 3], 0, 1) if E is not None else None
        depth = None
        if with_depth:
            d = render[..., -1:]
            depth = torch.where(alpha > 0, d, d.detach().max()).squeeze(0)
        normal = None
        if normal_ch >= 0:
            n = render[..., normal_ch:normal_ch + 3].squeeze(0)
            normal = (n / n.norm(dim=-1, keepdim=True) + 1) / 2
        return rgb, app, depth, normal

    def run(fn, dtype, device):
        P = [t.to(device=device, dtype=dtype).requires_grad_(True) if t is not None else None for t in (render, alpha, bg, E)]
        outs = fn(*P)
        loss = sum((o * c.to(device=device, dtype=dtype)).sum() for o, c in zip(outs, cots) if o is not None)
        loss.backward()
        return outs, [None if p is None else p.grad for p in P]

    ref_out, ref_grad = run(reference, torch.float64, "cpu")
    out, grad = run(lambda r, a, b, e: output_head(r, a, b, e, depth=with_depth, normal_channel=normal_ch), torch.float32, dev)
    for o, r, name in zip(out, ref_out, ("rgb", "rgb_appearance", "depth", "normal")):
        assert (o is None) == (r is None), name
        if o is not None:
            assert o.shape == r.shape, (name, o.shape, r.shape)
            assert float((o.detach().cpu().double() - r.detach()).abs().max()) < 3e-6, name
    for gq, gr, name in zip(grad, ref_grad, ("render", "alpha", "background", "exposure")):
        assert (gq is None) == (gr is None), name
        if gq is not None:
            scale = float(gr.abs().max()) + 1e-12
            # pixels whose fp32 value sits within rounding of a clamp edge may take the other branch than fp64
            diff = (gq.cpu().double() - gr).abs()
            if name in ("render", "alpha"):
                assert float((diff > 2e-5 * scale).double().mean()) < 1e-3, name
            else:
                assert float(diff.max()) < 5e-4 * scale, name


@pytest.mark.parametrize("n_nodes", [0, 1, 40])
def test_oob_loss_matches_the_reference_loop(hip_lib, n_nodes):
    """mtgs_amd.loss.oob_loss against the per-node loop of mtgs_scene_graph.py:949-967 written out in torch float64
    (model_id comparison, visible-node test, |means| > size / 2 + tolerance, -log(1 - sigmoid + 1e-6), mean)."""
    from mtgs_amd.loss import oob_loss
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(n_nodes + 11)
    sizes = [int(x) for x in torch.randint(1, 700, (n_nodes,), generator=g)]
    if n_nodes > 5:
        sizes[2], sizes[3] = 256, 257
    static = 1000                                                   # a static node in front: rigid nodes start after it
    starts, s = [], static
    for k in sizes:
        starts.append(s); s += k
    total = s + 333
    radii = (torch.randint(0, 30, (1, total), generator=g) * (torch.rand(1, total, generator=g) < 0.05)).int()
    if n_nodes > 5:
        radii[0, starts[4]:starts[4] + sizes[4]] = 0                # an object that is entirely invisible: skipped
    nodes = []
    for k in sizes:
        means = torch.randn(k, 3, generator=g) * 2.0
        nodes.append((means, torch.randn(k, 1, generator=g) * 2, [2.0 + torch.rand(1, generator=g).item(), 1.5, 4.0]))

    # reference loop, float64
    ops = [o.double().requires_grad_(True) for _, o, _ in nodes]
    visible = (radii > 0).flatten()
    loss, count = 0.0, 0
    for (means, _, size), o, st, k in zip(nodes, ops, starts, sizes):
        if visible[st:st + k].sum() == 0:
            continue
        oob = (means.double().abs() > (torch.tensor(size, dtype=torch.float64) / 2 + 1.5)[None]).any(-1)
        if oob.sum() != 0:
            loss = loss + (-torch.log(1 - o[oob].sigmoid() + 1e-6)).sum()
            count += int(oob.sum())
    ref = loss / count if count else torch.zeros((), dtype=torch.float64)
    if count:
        (3.0 * ref).backward()

    P = [o.to(dev).requires_grad_(True) for _, o, _ in nodes]
    val = oob_loss([(m.to(dev), p, size) for (m, _, size), p in zip(nodes, P)], radii.to(dev), starts, tolerance=1.5)
    ref = ref.detach() if torch.is_tensor(ref) else ref
    assert abs(float(val.detach()) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref)))
    if n_nodes:
        (3.0 * val).backward()
        for p, o in zip(P, ops):
            expect = o.grad if o.grad is not None else torch.zeros_like(o)
            assert p.grad.shape == o.shape
            assert torch.allclose(p.grad.cpu().double(), expect, rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("H,W,k,s", [(211, 333, 32, 16), (64, 96, 7, 7), (100, 100, 16, 5), (40, 40, 64, 16)])
def test_depth_ncc_loss_matches_the_reference_formulation(hip_lib, H, W, k, s):
    """mtgs_amd.loss.depth_ncc_loss against calculate_depth_ncc_loss (geometric_loss.py:322-348) restated with F.unfold in
    float64: value to 2e-5, gradient to 1e-3 of its maximum (fp32 sums over 1024-pixel patches); a 64-pixel patch on a
    40-pixel image has no valid patch: NaN, as the reference's mean of an empty tensor."""
    import torch.nn.functional as F
    from mtgs_amd.loss import depth_ncc_loss
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(H + k)
    gt = torch.rand(H, W, 1, generator=g) * 30 + 1
    pred0 = gt + torch.randn(H, W, 1, generator=g) * 2
    pred0[10:30, 10:30] = 7.25                                     # a flat region: variance ~ 0, the 1e-8 floor matters
    mask = torch.rand(H, W, 1, generator=g) > 0.0005
    mask[: H // 8] = False

    def reference(pred_depth, gt_depth, mask):
        pred_depth, gt_depth = pred_depth.squeeze(-1), gt_depth.squeeze(-1)
        pad = k // 2
        m = mask.squeeze(-1).to(pred_depth.dtype)
        pp = F.unfold(pred_depth[None, None], kernel_size=k, padding=pad, stride=s)
        gp = F.unfold(gt_depth[None, None], kernel_size=k, padding=pad, stride=s)
        mp = F.unfold(m[None, None], kernel_size=k, padding=pad, stride=s)
        valid = mp.all(dim=1).squeeze(0)
        pp, gp = pp[:, :, valid], gp[:, :, valid]
        pc, gc = pp - pp.mean(dim=1, keepdim=True), gp - gp.mean(dim=1, keepdim=True)
        ps = torch.sqrt((pc ** 2).mean(dim=1, keepdim=True) + 1e-8)
        gs = torch.sqrt((gc ** 2).mean(dim=1, keepdim=True) + 1e-8)
        return 1 - ((pc / ps) * (gc / gs)).mean(dim=1).mean(), int(valid.sum())

    p_ref = pred0.double().requires_grad_(True)
    ref, n_valid = reference(p_ref, gt.double(), mask)
    p = pred0.to(dev).requires_grad_(True)
    val = depth_ncc_loss(p, gt.to(dev), patch_size=k, stride=s, mask=mask.to(dev))
    if n_valid == 0:
        assert torch.isnan(val) and torch.isnan(ref)
        return
    (2.0 * ref).backward()
    (2.0 * val).backward()
    assert abs(float(val.detach()) - float(ref.detach())) <= 2e-5
    scale = float(p_ref.grad.abs().max())
    assert float((p.grad.cpu().double() - p_ref.grad).abs().max()) <= 1e-3 * scale


@pytest.mark.parametrize("H,W,C_", [(97, 131, 3), (1, 50, 3), (40, 1, 1), (64, 64, 4)])
def test_tv_loss_matches_the_reference_formulation(hip_lib, H, W, C_):
    """mtgs_amd.loss.tv_loss against TVLoss.forward (geometric_loss.py:293-303) in float64; a one-pixel-wide image has an
    empty difference tensor whose mean is NaN in the reference too."""
    from mtgs_amd.loss import tv_loss
    g = torch.Generator().manual_seed(H * 3 + W)
    x0 = torch.rand(H, W, C_, generator=g)
    if H > 5 and W > 5:
        x0[2, 3] = x0[2, 4]                       # exact ties: sign(0) = 0
    xr = x0.double().requires_grad_(True)
    ref = torch.mean(torch.abs(xr[:, :-1, :] - xr[:, 1:, :])) + torch.mean(torch.abs(xr[:-1, :, :] - xr[1:, :, :]))
    x = x0.cuda().requires_grad_(True)
    val = tv_loss(x)
    if torch.isnan(ref):
        assert torch.isnan(val)
        return
    (1.5 * ref).backward()
    (1.5 * val).backward()
    assert abs(float(val.detach()) - float(ref.detach())) < 2e-6
    assert torch.allclose(x.grad.cpu().double(), xr.grad, rtol=1e-5, atol=1e-9)
