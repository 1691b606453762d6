"""TEST INFRASTRUCTURE ONLY (checker for mtgs_amd.loss.masked_ssim; nothing under mtgs_amd/ imports this).

CPU restatement, in numpy float64, of mtgs.utils.ssim.MaskedSSIM(data_range=1.0, size_average=True, channel=3) as MTGS
calls it (/root/reference/mtgs/utils/ssim.py: `_fspecial_gauss_1d` :11-26, `gaussian_filter` :29-54, `_ssim` :57-108,
`ssim` :111-190; call site /root/reference/mtgs/scene_model/mtgs_scene_graph.py:831-841), with the analytic gradient
with respect to the second argument.  PINNED: tests/test_oracle_ssim.py checks it against tests/golden/ssim_ref.npz,
which the reference module itself produced (tests/golden/make_ssim_golden.py)."""
import numpy as np


def gauss_window(size=11, sigma=1.5):
    # ssim.py:11-26: the reference builds it with torch in float32 (torch's float32 exp, not numpy's: they differ
    # in the last bit, which moves the result by 1e-7)
    import torch
    coords = torch.arange(size, dtype=torch.float)
    coords -= size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g /= g.sum()
    return g.numpy().astype(np.float64)


def _filter_valid(img, w):
    """'valid' separable correlation over the first two axes of img[H,W,C] (ssim.py:29-54, padding=0)."""
    n = len(w)
    H, W = img.shape[:2]
    tmp = sum(w[k] * img[k:H - n + 1 + k] for k in range(n))
    return sum(w[k] * tmp[:, k:W - n + 1 + k] for k in range(n))


def _filter_full_T(g, w, H, W):
    """transpose of _filter_valid: scatters g[(H-n+1),(W-n+1),C] back onto an [H,W,C] grid."""
    n = len(w)
    tmp = np.zeros((g.shape[0], W) + g.shape[2:])
    for k in range(n):
        tmp[:, k:k + g.shape[1]] += w[k] * g
    out = np.zeros((H, W) + g.shape[2:])
    for k in range(n):
        out[k:k + g.shape[0]] += w[k] * tmp
    return out


def masked_ssim(gt, pred, mask=None, win_sigma=1.5, data_range=1.0, K=(0.01, 0.03), with_grad=False):
    """gt, pred [H,W,3]; mask [H,W] or [H,W,1] bool or None.  Returns the scalar (and d scalar / d pred)."""
    X, Y = np.asarray(gt, np.float64), np.asarray(pred, np.float64)
    H, W = X.shape[:2]
    w = gauss_window(11, win_sigma)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2          # ssim.py:82-83
    mu1, mu2 = _filter_valid(X, w), _filter_valid(Y, w)                   # :87-88
    e11, e22, e12 = _filter_valid(X * X, w), _filter_valid(Y * Y, w), _filter_valid(X * Y, w)
    s1, s2, s12 = e11 - mu1 ** 2, e22 - mu2 ** 2, e12 - mu1 * mu2         # :94-96
    dA, dB = mu1 ** 2 + mu2 ** 2 + C1, s1 + s2 + C2
    A, B = (2 * mu1 * mu2 + C1) / dA, (2 * s12 + C2) / dB                 # :98-99
    smap = A * B
    if mask is None:
        m = np.ones(smap.shape[:2] + (1,))
    else:
        m = np.asarray(mask).reshape(H, W, 1)[5:-5, 5:-5].astype(np.float64)   # ssim.py:150-153 (margin = 11 // 2)
    m3 = np.broadcast_to(m, smap.shape)
    cnt = m3.sum()
    val = (smap * m3).sum() / cnt                                        # :101-103 masked_select(...).mean()
    if not with_grad:
        return val
    g_mu = m3 * ((2 * mu1 - A * 2 * mu2) / dA * B + A * (-2 * mu1 + B * 2 * mu2) / dB)
    g_e22 = m3 * (A * (-B / dB))
    g_e12 = m3 * (A * (2.0 / dB))
    grad = _filter_full_T(g_mu, w, H, W) + 2 * Y * _filter_full_T(g_e22, w, H, W) + X * _filter_full_T(g_e12, w, H, W)
    return val, grad / cnt
