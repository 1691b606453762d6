// tv.hip -- total-variation term of the normal image (use_normal_tv_loss = True in the shipped config/MTGS.py:114).
//
// Restates TVLoss.forward (/root/reference/mtgs/utils/geometric_loss.py:293-303) for one image [H,W,C]:
//     h_diff = pred[:, :-1, :] - pred[:, 1:, :];  w_diff = pred[:-1, :, :] - pred[1:, :, :]
//     loss   = mean(|h_diff|) + mean(|w_diff|)
// (PyTorch: two slice pairs, abs, two means forward; four zero-filled full-size gradients and their adds backward.)
// Forward: per-block partial sums of both terms in a fixed order; backward: thread per element gathering the signs of
// its (up to) four differences.  NaN inputs propagate as in the reference (MTGS drops the term when it is not finite).
#include "common.hpp"

namespace {
constexpr int TV_BLOCK = 256;

__device__ __forceinline__ float block_sum(float v, float *lds) {
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

__global__ __launch_bounds__(TV_BLOCK) void tv_fwd_kernel(int H, int W, int C, const float *__restrict__ x, float *__restrict__ partials) {
    __shared__ float s_red[4];
    const int64_t n = (int64_t)H * W * C, e = (int64_t)blockIdx.x * TV_BLOCK + threadIdx.x;
    float a = 0.f, b = 0.f;
    if (e < n) {
        const int64_t pix = e / C;
        const int col = (int)(pix % W), row = (int)(pix / W);
        const float v = x[e];
        if (col + 1 < W) a = fabsf(v - x[e + C]);
        if (row + 1 < H) b = fabsf(v - x[e + (int64_t)W * C]);
    }
    const float sa = block_sum(a, s_red), sb = block_sum(b, s_red);
    if (threadIdx.x == 0) { partials[(int64_t)blockIdx.x * 2] = sa; partials[(int64_t)blockIdx.x * 2 + 1] = sb; }
}

__global__ __launch_bounds__(TV_BLOCK) void tv_finish_kernel(int64_t nblocks, float inv_a, float inv_b, const float *__restrict__ partials,
                                                             float *__restrict__ out) {
    __shared__ float s_red[4];
    float a = 0.f, b = 0.f;
    for (int64_t i = threadIdx.x; i < nblocks; i += TV_BLOCK) { a += partials[i * 2]; b += partials[i * 2 + 1]; }
    const float ta = block_sum(a, s_red), tb = block_sum(b, s_red);
    if (threadIdx.x == 0) out[0] = ta * inv_a + tb * inv_b;
}

__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : (d == 0.f ? 0.f : d)); }   // NaN stays NaN

__global__ __launch_bounds__(TV_BLOCK) void tv_bwd_kernel(int H, int W, int C, const float *__restrict__ x, const float *__restrict__ v_out,
                                                          float inv_a, float inv_b, float *__restrict__ v_x) {
    const int64_t n = (int64_t)H * W * C, e = (int64_t)blockIdx.x * TV_BLOCK + threadIdx.x;
    if (e >= n) return;
    const int64_t pix = e / C, rs = (int64_t)W * C;
    const int col = (int)(pix % W), row = (int)(pix / W);
    const float v = x[e];
    float ga = 0.f, gb = 0.f;
    if (col + 1 < W) ga += sgn(v - x[e + C]);
    if (col > 0) ga -= sgn(x[e - C] - v);
    if (row + 1 < H) gb += sgn(v - x[e + rs]);
    if (row > 0) gb -= sgn(x[e - rs] - v);
    v_x[e] = v_out[0] == 0.f ? 0.f : v_out[0] * (ga * inv_a + gb * inv_b);   // zero cotangent -> zero, also next to NaN pixels
}

inline void scales_of(int H, int W, int C, float &inv_a, float &inv_b) {   // mean over an empty tensor is NaN in torch
    const double na = (double)H * (W - 1) * C, nb = (double)(H - 1) * W * C;
    inv_a = na > 0 ? (float)(1.0 / na) : __builtin_nanf("");
    inv_b = nb > 0 ? (float)(1.0 / nb) : __builtin_nanf("");
}
}  // namespace

extern "C" int mtgs_tv_workspace_floats(int width, int height, int channels, size_t *n) {
    MTGS_REQUIRE(width > 0 && height > 0 && channels > 0 && n, MTGS_EINVAL, "mtgs_tv_workspace_floats: bad arguments");
    *n = (size_t)ceil_div64((int64_t)width * height * channels, TV_BLOCK) * 2;
    return MTGS_OK;
}

extern "C" int mtgs_tv_fwd(int width, int height, int channels, const float *image, float *partials, float *out, void *stream) {
    MTGS_REQUIRE(width > 0 && height > 0 && channels > 0, MTGS_EINVAL, "mtgs_tv_fwd: bad sizes");
    MTGS_REQUIRE(image && partials && out, MTGS_EINVAL, "mtgs_tv_fwd: null pointer");
    const int64_t nblocks = ceil_div64((int64_t)width * height * channels, TV_BLOCK);
    float ia, ib;
    scales_of(height, width, channels, ia, ib);
    hipStream_t st = (hipStream_t)stream;
    tv_fwd_kernel<<<(unsigned)nblocks, TV_BLOCK, 0, st>>>(height, width, channels, image, partials);
    tv_finish_kernel<<<1, TV_BLOCK, 0, st>>>(nblocks, ia, ib, partials, out);
    MTGS_CHECK_LAUNCH("mtgs_tv_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_tv_bwd(int width, int height, int channels, const float *image, const float *v_out, float *v_image, void *stream) {
    MTGS_REQUIRE(width > 0 && height > 0 && channels > 0, MTGS_EINVAL, "mtgs_tv_bwd: bad sizes");
    MTGS_REQUIRE(image && v_out && v_image, MTGS_EINVAL, "mtgs_tv_bwd: null pointer");
    float ia, ib;
    scales_of(height, width, channels, ia, ib);
    tv_bwd_kernel<<<(unsigned)ceil_div64((int64_t)width * height * channels, TV_BLOCK), TV_BLOCK, 0, (hipStream_t)stream>>>(
        height, width, channels, image, v_out, ia, ib, v_image);
    MTGS_CHECK_LAUNCH("mtgs_tv_bwd");
    return MTGS_OK;
}

// ---- the sum of the loss dictionary (mtgs_scene_graph.py:823-945: every term scaled by its lambda, the normal term added only
// when it is finite, :939; the trainer then adds the dictionary's values up): out = c + sum_i w_i t_i with guarded terms dropped
// when they are not finite.  PyTorch: a multiply and an add per term, isfinite + where + zeros_like for the guard, and their
// backward -- ~30 launches of one-element kernels; here one each way.
struct CombineArgs { float w[16]; unsigned guard; float c; int n; };
__global__ void combine_fwd_kernel(const float *__restrict__ terms, CombineArgs a, float *__restrict__ out, unsigned *__restrict__ kept) {
    if (threadIdx.x != 0) return;
    float s = a.c;
    unsigned k = 0;
    for (int i = 0; i < a.n; ++i) {
        const float t = terms[i];
        const bool drop = ((a.guard >> i) & 1u) && !(fabsf(t) <= 3.402823466e+38f);      // NaN or +-inf
        if (!drop) { s += a.w[i] * t; k |= 1u << i; }
    }
    out[0] = s;
    kept[0] = k;
}
__global__ void combine_bwd_kernel(const float *__restrict__ v_out, const unsigned *__restrict__ kept, CombineArgs a,
                                   float *__restrict__ v_terms) {
    const int i = threadIdx.x;
    if (i < a.n) v_terms[i] = ((kept[0] >> i) & 1u) ? a.w[i] * v_out[0] : 0.f;
}

extern "C" int mtgs_loss_combine_fwd(int n, const float *terms, const float *weights, unsigned guard_mask, float constant, float *out,
                                     uint32_t *kept, void *stream) {
    MTGS_REQUIRE(n >= 1 && n <= 16 && terms && weights && out && kept, MTGS_EINVAL, "mtgs_loss_combine_fwd: 1 .. 16 terms");
    CombineArgs a;
    for (int i = 0; i < 16; ++i) a.w[i] = i < n ? weights[i] : 0.f;      // (HOST array)
    a.guard = guard_mask; a.c = constant; a.n = n;
    combine_fwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(terms, a, out, kept);
    MTGS_CHECK_LAUNCH("mtgs_loss_combine_fwd");
    return MTGS_OK;
}
extern "C" int mtgs_loss_combine_bwd(int n, const float *v_out, const uint32_t *kept, const float *weights, float *v_terms,
                                     void *stream) {
    MTGS_REQUIRE(n >= 1 && n <= 16 && v_out && kept && weights && v_terms, MTGS_EINVAL, "mtgs_loss_combine_bwd: 1 .. 16 terms");
    CombineArgs a;
    for (int i = 0; i < 16; ++i) a.w[i] = i < n ? weights[i] : 0.f;
    a.guard = 0; a.c = 0.f; a.n = n;
    combine_bwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(v_out, kept, a, v_terms);
    MTGS_CHECK_LAUNCH("mtgs_loss_combine_bwd");
    return MTGS_OK;
}
