"""CPU restatement (numpy, float64) of the densification of a Gaussian node -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py line by line:
refinement_after :476-577 (densification branch), split_gaussians :630-676, dup_gaussians :678-699,
cull_gaussians :579-612, dup_in_optim :418-437, remove_from_optim :392-410.  The reference draws its samples with
torch.randn; here they are an ARGUMENT (`normals(index, slot)`), so that the device path -- which draws them from
Philox4x32-10 keyed by (seed, step, index, slot), restated below -- can be compared row for row.  Parity for this
neighbour is therefore "up to the RNG convention" (DESIGN.md section 6)."""
import numpy as np


def philox4x32_10(c, k):
    """c: [n,4] uint32 counters, k: (k0, k1); returns [n,4] uint32 (Salmon et al. 2011)."""
    c = c.astype(np.uint64).copy()
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & MASK
        c = np.stack([n0, p1 & MASK, n2, p0 & MASK], 1)
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c.astype(np.uint32)


def normals3(seed, step, index, slot):
    """The three standard normals csrc/refine.hip draws for (Gaussian index, slot) -- in float64."""
    index = np.asarray(index, dtype=np.uint32).reshape(-1)
    c = np.stack([index, np.full_like(index, slot), np.full_like(index, step), np.zeros_like(index)], 1)
    r = philox4x32_10(c, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.float64)
    u = (r + 0.5) * 2.0 ** -32
    ra, rb = np.sqrt(-2 * np.log(u[:, 0])), np.sqrt(-2 * np.log(u[:, 2]))
    return np.stack([ra * np.cos(2 * np.pi * u[:, 1]), ra * np.sin(2 * np.pi * u[:, 1]), rb * np.cos(2 * np.pi * u[:, 3])], 1)


def quat_to_rotmat(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                     2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def refinement_after(params, stats, cfg, step, normals, moments=None):
    """params: dict of float64 arrays with N rows; stats = (xys_grad_norm, vis_counts, max_2Dsize); cfg: an object with the
    control fields; normals(index_array, slot) -> [n,3].  Returns (new_params, new_moments, masks)."""
    p = {k: np.array(v, dtype=np.float64) for k, v in params.items()}
    gn, vc, m2 = (np.asarray(t, dtype=np.float64).reshape(-1) for t in stats)
    N = p["means"].shape[0]
    S = cfg.n_split_samples
    avg = gn / vc                                                                   # :496
    high = avg > cfg.densify_grad_thresh                                            # :498
    splits = (np.exp(p["scales"]).max(-1) > cfg.densify_size_thresh) & high         # :500-501
    if step < cfg.stop_screen_size_at:
        splits |= m2 > cfg.split_screen_size                                        # :503-504
    idx = np.arange(N)
    # split_gaussians (:630-676): samples are laid out sample-major (`.repeat(samps, 1)`)
    sp = idx[splits]
    z = np.concatenate([normals(sp, s) for s in range(S)], 0) if len(sp) else np.zeros((0, 3))
    rep = lambda a: np.concatenate([a[splits]] * S, 0)
    scaled = np.exp(rep(p["scales"])) * z
    q = p["quats"][splits] / np.linalg.norm(p["quats"][splits], axis=-1, keepdims=True)
    rots = quat_to_rotmat(np.concatenate([q] * S, 0)) if len(sp) else np.zeros((0, 3, 3))
    split_params = {k: rep(v) for k, v in p.items()}
    split_params["means"] = np.einsum("nij,nj->ni", rots, scaled) + rep(p["means"])
    split_params["scales"] = np.log(np.exp(rep(p["scales"])) / 1.6)
    p["scales"][splits] = np.log(np.exp(p["scales"][splits]) / 1.6)                 # :657 (in place)
    dups = (np.exp(p["scales"]).max(-1) <= cfg.densify_size_thresh) & high          # :509-510 (after the in-place shrink)
    dp = idx[dups]
    dup_params = {k: v[dups].copy() for k, v in p.items()}
    if cfg.clone_sample_means and len(dp):                                          # :686-697
        zd = normals(dp, S)
        qd = p["quats"][dups] / np.linalg.norm(p["quats"][dups], axis=-1, keepdims=True)
        dup_params["means"] = np.einsum("nij,nj->ni", quat_to_rotmat(qd), np.exp(p["scales"][dups]) * zd) + p["means"][dups]
    allp = {k: np.concatenate([p[k], split_params[k], dup_params[k]], 0) for k in p}   # :512-515
    m2_all = np.concatenate([m2, np.zeros(len(sp) * S), np.zeros(len(dp))])         # :517-524
    kind = np.concatenate([np.zeros(N, np.int64)] + [np.full(len(sp), 1 + s) for s in range(S)] + [np.full(len(dp), 1 + S)])
    src = np.concatenate([idx] + [sp] * S + [dp])
    splits_mask = np.concatenate([splits, np.zeros(len(sp) * S + len(dp), bool)])   # :532-541
    # cull_gaussians (:579-612)
    culls = 1.0 / (1.0 + np.exp(-allp["opacities"].reshape(-1))) < cfg.cull_alpha_thresh
    culls |= splits_mask
    if step > cfg.refine_every * cfg.reset_alpha_every:
        far = np.linalg.norm(allp["means"], axis=-1) > 100
        toobig = np.exp(allp["scales"]).max(-1) > np.where(far, 40.0, 1.0) * cfg.cull_scale_thresh
        if step < cfg.stop_screen_size_at:
            toobig |= m2_all > cfg.cull_screen_size
        culls |= toobig
    keep = ~culls
    new = {k: v[keep] for k, v in allp.items()}
    new_m = None
    if moments is not None:                                                         # dup_in_optim zeros + remove_from_optim
        new_m = {}
        for k, (a, b) in moments.items():
            pad = lambda t: np.concatenate([np.asarray(t, np.float64), np.zeros((len(sp) * S + len(dp),) + t.shape[1:])], 0)[keep]
            new_m[k] = (pad(a), pad(b))
    return new, new_m, {"splits": splits, "dups": dups, "keep": keep, "kind": kind[keep], "src_index": src[keep]}
