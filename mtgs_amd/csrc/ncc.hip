// ncc.hip -- patch-wise normalised cross-correlation between the rendered and the reference depth image, forward and
// backward (ncc_loss_lambda = 0.1 in the shipped config/MTGS.py:110; default patches 32 x 32, stride 16).
//
// Restates calculate_depth_ncc_loss (/root/reference/mtgs/utils/geometric_loss.py:322-348), which unfolds the two depth
// images and the mask into [1, k*k, L] patch matrices, keeps the patches whose mask is all ones with a boolean index
// (a host synchronisation; its backward is a sorting index_put) and computes per patch
//     pc = p - mean(p), gc = g - mean(g), ps = sqrt(mean(pc^2) + 1e-8), gs likewise, ncc = mean(pc gc) / (ps gs)
//     loss = 1 - mean over the valid patches of ncc.
// Zero padding of k / 2 on every side: a patch that reaches into the padding has mask 0 there and is never valid.
// Forward: one workgroup per patch, two passes over its k*k pixels (means, then centred sums -- as the reference; the
// one-pass variance cancels catastrophically on flat depth), per-patch statistics saved.  Backward: thread per PIXEL,
// gathering from the <= (k / stride)^2 patches that contain it (no atomics, no zero-fill).  Roofline: latency / L2.
#include "common.hpp"

namespace {
constexpr int NCC_BLOCK = 256, NCC_STATS = 6;   // pm, gm, ps, gs, C = mean(pc gc), valid

struct NccGrid { int H, W, k, s, pad, Lh, Lw; };

__device__ __forceinline__ float block_sum(float v, float *lds) {
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

__global__ __launch_bounds__(NCC_BLOCK) void ncc_fwd_kernel(NccGrid G, const float *__restrict__ pred, const float *__restrict__ gt,
                                                            const uint8_t *__restrict__ mask, float *__restrict__ stats) {
    __shared__ float s_red[4];
    const int patch = blockIdx.x, pi = patch / G.Lw, pj = patch - pi * G.Lw;
    const int y0 = pi * G.s - G.pad, x0 = pj * G.s - G.pad, n = G.k * G.k;
    float *st = stats + (int64_t)patch * NCC_STATS;
    const bool inside = y0 >= 0 && x0 >= 0 && y0 + G.k <= G.H && x0 + G.k <= G.W;   // wave-uniform
    float bad = 0.f, sp = 0.f, sg = 0.f;
    if (inside) {
        for (int e = threadIdx.x; e < n; e += NCC_BLOCK) {
            const int64_t q = (int64_t)(y0 + e / G.k) * G.W + x0 + e % G.k;
            if (mask && !mask[q]) bad = 1.f;
            sp += pred[q]; sg += gt[q];
        }
    }
    const float nbad = block_sum(bad, s_red);
    if (!inside || nbad > 0.f) {
        if (threadIdx.x == 0) { st[0] = st[1] = st[4] = 0.f; st[2] = st[3] = 1.f; st[5] = 0.f; }
        return;
    }
    const float inv_n = 1.0f / (float)n;
    const float pm = block_sum(sp, s_red) * inv_n, gm = block_sum(sg, s_red) * inv_n;
    float spp = 0.f, sgg = 0.f, spg = 0.f;
    for (int e = threadIdx.x; e < n; e += NCC_BLOCK) {
        const int64_t q = (int64_t)(y0 + e / G.k) * G.W + x0 + e % G.k;
        const float pc = pred[q] - pm, gc = gt[q] - gm;
        spp += pc * pc; sgg += gc * gc; spg += pc * gc;
    }
    const float vp = block_sum(spp, s_red) * inv_n, vg = block_sum(sgg, s_red) * inv_n, c = block_sum(spg, s_red) * inv_n;
    if (threadIdx.x == 0) {
        st[0] = pm; st[1] = gm; st[2] = sqrtf(vp + 1e-8f); st[3] = sqrtf(vg + 1e-8f); st[4] = c; st[5] = 1.f;
    }
}

// out[0] = 1 - mean of ncc over the valid patches (NaN when there is none, as the reference's mean of an empty tensor),
// out[1] = number of valid patches; fixed summation order
__global__ __launch_bounds__(NCC_BLOCK) void ncc_finish_kernel(int64_t L, const float *__restrict__ stats, float *__restrict__ out) {
    __shared__ float s_red[4];
    float s = 0.f, c = 0.f;
    for (int64_t p = threadIdx.x; p < L; p += NCC_BLOCK) {
        const float *st = stats + p * NCC_STATS;
        if (st[5] != 0.f) { s += st[4] / (st[2] * st[3]); c += 1.f; }
    }
    const float ts = block_sum(s, s_red), tc = block_sum(c, s_red);
    if (threadIdx.x == 0) { out[0] = 1.f - ts / tc; out[1] = tc; }
}

// d ncc / d p_i = gc_i / (n ps gs) - C pc_i / (n ps^3 gs)   (the derivatives through the two means vanish: sum pc = sum gc = 0)
__global__ __launch_bounds__(NCC_BLOCK) void ncc_bwd_kernel(NccGrid G, const float *__restrict__ pred, const float *__restrict__ gt,
                                                            const float *__restrict__ stats, const float *__restrict__ v_out,
                                                            const float *__restrict__ fwd_out, float *__restrict__ v_pred) {
    const int64_t q = (int64_t)blockIdx.x * NCC_BLOCK + threadIdx.x;
    if (q >= (int64_t)G.H * G.W) return;
    const int y = (int)(q / G.W), x = (int)(q - (int64_t)y * G.W);
    const float scale = -v_out[0] / (fwd_out[1] * (float)(G.k * G.k));
    const float p = pred[q], g = gt[q];
    // patches (i, j) with i*s - pad <= y < i*s - pad + k
    const int i_hi = min((y + G.pad) / G.s, G.Lh - 1), i_lo = max((y + G.pad - G.k + G.s) / G.s, 0);   // ceil((y+pad-k+1)/s)
    const int j_hi = min((x + G.pad) / G.s, G.Lw - 1), j_lo = max((x + G.pad - G.k + G.s) / G.s, 0);
    float acc = 0.f;
    for (int i = i_lo; i <= i_hi; ++i)
        for (int j = j_lo; j <= j_hi; ++j) {
            const float *st = stats + ((int64_t)i * G.Lw + j) * NCC_STATS;
            if (st[5] == 0.f) continue;
            const float pc = p - st[0], gc = g - st[1], ps = st[2], gs = st[3];
            acc += gc / (ps * gs) - st[4] * pc / ((ps * ps) * (ps * gs));
        }
    v_pred[q] = scale * acc;
}

inline bool make_grid(int W, int H, int k, int s, NccGrid &G) {
    if (W <= 0 || H <= 0 || k <= 0 || s <= 0) return false;
    G.H = H; G.W = W; G.k = k; G.s = s; G.pad = k / 2;
    G.Lh = (H + 2 * G.pad - k) / s + 1;
    G.Lw = (W + 2 * G.pad - k) / s + 1;
    return G.Lh > 0 && G.Lw > 0;
}
}  // namespace

extern "C" int mtgs_ncc_patches(int width, int height, int patch_size, int stride, int64_t *n) {
    NccGrid G;
    MTGS_REQUIRE(n && make_grid(width, height, patch_size, stride, G), MTGS_EINVAL, "mtgs_ncc_patches: bad arguments");
    *n = (int64_t)G.Lh * G.Lw;
    return MTGS_OK;
}

extern "C" int mtgs_ncc_fwd(int width, int height, int patch_size, int stride, const float *pred, const float *gt,
                            const uint8_t *mask, float *patch_stats, float *out, void *stream) {
    NccGrid G;
    MTGS_REQUIRE(make_grid(width, height, patch_size, stride, G), MTGS_EINVAL, "mtgs_ncc_fwd: bad sizes");
    MTGS_REQUIRE(pred && gt && patch_stats && out, MTGS_EINVAL, "mtgs_ncc_fwd: null pointer");
    const int64_t L = (int64_t)G.Lh * G.Lw;
    MTGS_REQUIRE(L < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_ncc_fwd: too many patches");
    hipStream_t st = (hipStream_t)stream;
    ncc_fwd_kernel<<<(unsigned)L, NCC_BLOCK, 0, st>>>(G, pred, gt, mask, patch_stats);
    ncc_finish_kernel<<<1, NCC_BLOCK, 0, st>>>(L, patch_stats, out);
    MTGS_CHECK_LAUNCH("mtgs_ncc_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_ncc_bwd(int width, int height, int patch_size, int stride, const float *pred, const float *gt,
                            const float *patch_stats, const float *v_out, const float *fwd_out, float *v_pred, void *stream) {
    NccGrid G;
    MTGS_REQUIRE(make_grid(width, height, patch_size, stride, G), MTGS_EINVAL, "mtgs_ncc_bwd: bad sizes");
    MTGS_REQUIRE(pred && gt && patch_stats && v_out && fwd_out && v_pred, MTGS_EINVAL, "mtgs_ncc_bwd: null pointer");
    ncc_bwd_kernel<<<(unsigned)ceil_div64((int64_t)width * height, NCC_BLOCK), NCC_BLOCK, 0, (hipStream_t)stream>>>(
        G, pred, gt, patch_stats, v_out, fwd_out, v_pred);
    MTGS_CHECK_LAUNCH("mtgs_ncc_bwd");
    return MTGS_OK;
}
