// refine.hip -- densification of one Gaussian node on the device: split / duplicate / cull with every per-Gaussian
// tensor (parameters AND the optimizer's moment rows) compacted and appended by kernels, and the split / clone samples
// drawn from a COUNTER-BASED generator keyed by (seed, step, Gaussian index, sample), so that every rank of a
// data-parallel job -- whatever its launch geometry -- makes bit-identical decisions and rows.
//
// Restates VanillaGaussianSplattingModel.refinement_after / split_gaussians / dup_gaussians / cull_gaussians and the
// optimizer surgery dup_in_optim / remove_from_optim
// (/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:392-446, 476-577, 579-699), which run
// as ~60 masked PyTorch ops with half a dozen .item() synchronisations and per-rank torch.randn (:642, :687):
//   avg = xys_grad_norm / vis_counts;  high = avg > densify_grad_thresh
//   split = (max exp(scales) > densify_size_thresh & high) | (max_2Dsize > split_screen_size)        [the latter early on]
//   children (n_split_samples per split): mean + R(q/|q|) (exp(scales) * z),  scales <- log(exp(scales) / 1.6)
//   (the parent's scales shrink IN PLACE too, :657 -- so)  dup = (max exp(scales') <= densify_size_thresh) & high
//   copies of dup (their mean resampled the same way when clone_sample_means, :686-697)
//   new set = [old | children, sample-major | dups];  cull: split parents, sigmoid(opacity) < cull_alpha_thresh, and (later)
//   world-space / screen-space size limits (:585-612); moments: old rows follow, new rows start at zero (:418-437).
// The result order is the reference's (boolean-mask order of that concatenation).
//
// Kernels: classify (per old Gaussian: what it becomes + how many rows survive), then -- after an exclusive scan of the
// survivor counts -- index (source row and kind of every output row), geometry (means / scales of every output row) and
// one generic row copy per remaining tensor.  Roofline: HBM (streaming + gathers), launch-bound at node sizes.
#include "common.hpp"

namespace {

struct RefineCfg {
    float grad_thresh, size_thresh, split_screen_size, cull_alpha_thresh, cull_scale_thresh, cull_screen_size;
    int nsamps, use_screen_split, cull_big, cull_screen, clone_sample_means;
    uint32_t seed_lo, seed_hi, step;
};
constexpr int MAX_SAMPS = 4;
constexpr float kSplitShrink = 1.6f;

// ---- Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): counter (index, slot, step, 0), key seed
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// three standard normals for (Gaussian index, slot): Box-Muller on uniforms (x + 0.5) * 2^-32
__device__ __forceinline__ void refine_normal3(const RefineCfg &cfg, uint32_t index, uint32_t slot, float (&z)[3]) {
    uint32_t r[4];
    philox4x32_10(index, slot, cfg.step, 0u, cfg.seed_lo, cfg.seed_hi, r);
    const float u0 = ((float)r[0] + 0.5f) * 2.3283064365386963e-10f, u1 = ((float)r[1] + 0.5f) * 2.3283064365386963e-10f;
    const float u2 = ((float)r[2] + 0.5f) * 2.3283064365386963e-10f, u3 = ((float)r[3] + 0.5f) * 2.3283064365386963e-10f;
    const float ra = sqrtf(-2.f * logf(fmaxf(u0, 1e-37f))), rb = sqrtf(-2.f * logf(fmaxf(u2, 1e-37f)));
    z[0] = ra * cosf(6.283185307179586f * u1);
    z[1] = ra * sinf(6.283185307179586f * u1);
    z[2] = rb * cosf(6.283185307179586f * u3);
}
// mean + R(q / |q|) (exp(scales) * z)     (wxyz; mtgs utils.quat_to_rotmat)
__device__ __forceinline__ void sample_mean(const float (&m)[3], const float (&sl)[3], const float4 q, const float (&z)[3], float (&out)[3]) {
    const float inv = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float w = q.x * inv, x = q.y * inv, y = q.z * inv, zz = q.w * inv;
    const float v0 = expf(sl[0]) * z[0], v1 = expf(sl[1]) * z[1], v2 = expf(sl[2]) * z[2];
    out[0] = m[0] + (1 - 2 * (y * y + zz * zz)) * v0 + 2 * (x * y - w * zz) * v1 + 2 * (x * zz + w * y) * v2;
    out[1] = m[1] + 2 * (x * y + w * zz) * v0 + (1 - 2 * (x * x + zz * zz)) * v1 + 2 * (y * zz - w * x) * v2;
    out[2] = m[2] + 2 * (x * zz - w * y) * v0 + 2 * (y * zz + w * x) * v1 + (1 - 2 * (x * x + y * y)) * v2;
}
__device__ __forceinline__ float max_exp(const float (&sl)[3]) { return fmaxf(fmaxf(expf(sl[0]), expf(sl[1])), expf(sl[2])); }
// cull_gaussians (:579-612) for one row
__device__ __forceinline__ bool culled(const RefineCfg &c, const float (&m)[3], const float (&sl)[3], float opacity_logit, float max2d) {
    bool cull = 1.0f / (1.0f + expf(-opacity_logit)) < c.cull_alpha_thresh;
    if (c.cull_big) {
        const bool far = sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]) > 100.f;
        cull = cull || max_exp(sl) > (far ? 40.f : 1.f) * c.cull_scale_thresh;
        if (c.cull_screen) cull = cull || max2d > c.cull_screen_size;
    }
    return cull;
}
struct Node {
    float m[3], sl[3], sl_cur[3];   // sl_cur: the scales after the in-place shrink of a split parent
    float4 q;
    float opac;
    bool split, dup;
};
__device__ __forceinline__ Node classify(const RefineCfg &c, int64_t i, const float *means, const float *scales, const float *quats,
                                         const float *opacities, const float *grad_norm, const float *vis_counts, const float *max2d) {
    Node n;
#pragma unroll
    for (int k = 0; k < 3; ++k) { n.m[k] = means[i * 3 + k]; n.sl[k] = scales[i * 3 + k]; }
    n.q = reinterpret_cast<const float4 *>(quats)[i];
    n.opac = opacities[i];
    const bool high = grad_norm[i] / vis_counts[i] > c.grad_thresh;
    n.split = (max_exp(n.sl) > c.size_thresh) && high;
    if (c.use_screen_split) n.split = n.split || max2d[i] > c.split_screen_size;
#pragma unroll
    for (int k = 0; k < 3; ++k) n.sl_cur[k] = n.split ? logf(expf(n.sl[k]) / kSplitShrink) : n.sl[k];
    n.dup = (max_exp(n.sl_cur) <= c.size_thresh) && high;
    return n;
}
// flags: bit 0 old row kept | bit 1+s child s kept | bit 1+nsamps dup kept | bit 7 split
__global__ __launch_bounds__(256) void refine_classify_kernel(int64_t N, const float *__restrict__ means, const float *__restrict__ scales,
                                                             const float *__restrict__ quats, const float *__restrict__ opacities,
                                                             const float *__restrict__ grad_norm, const float *__restrict__ vis_counts,
                                                             const float *__restrict__ max2d, const RefineCfg c,
                                                             int32_t *__restrict__ counts /* [2 + nsamps][N] */, uint8_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const Node n = classify(c, i, means, scales, quats, opacities, grad_norm, vis_counts, max2d);
    uint32_t f = n.split ? 0x80u : 0u;
    if (!n.split && !culled(c, n.m, n.sl, n.opac, max2d[i])) f |= 1u;
    for (int s = 0; s < c.nsamps; ++s) {
        bool keep = false;
        if (n.split) {
            float z[3], cm[3];
            refine_normal3(c, (uint32_t)i, (uint32_t)s, z);
            sample_mean(n.m, n.sl, n.q, z, cm);
            keep = !culled(c, cm, n.sl_cur, n.opac, 0.f);   // (new rows enter with max_2Dsize = 0, :517-524)
        }
        if (keep) f |= 2u << s;
        counts[(int64_t)(1 + s) * N + i] = keep ? 1 : 0;
    }
    bool keep_dup = false;
    if (n.dup) {
        float dm[3] = {n.m[0], n.m[1], n.m[2]};
        if (c.clone_sample_means) {
            float z[3];
            refine_normal3(c, (uint32_t)i, (uint32_t)c.nsamps, z);
            sample_mean(n.m, n.sl_cur, n.q, z, dm);
        }
        keep_dup = !culled(c, dm, n.sl_cur, n.opac, 0.f);
    }
    if (keep_dup) f |= 2u << c.nsamps;
    counts[i] = (int32_t)(f & 1u);
    counts[(int64_t)(1 + c.nsamps) * N + i] = keep_dup ? 1 : 0;
    flags[i] = (uint8_t)f;
}

// pos: EXCLUSIVE scans of the count columns, bases[k] = first output row of column k's block
__global__ __launch_bounds__(256) void refine_index_kernel(int64_t N, int nsamps, const uint8_t *__restrict__ flags,
                                                          const int64_t *__restrict__ pos /* [2 + nsamps][N] */,
                                                          const int64_t *__restrict__ bases, int32_t *__restrict__ src_index,
                                                          uint8_t *__restrict__ kind) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t f = flags[i];
    for (int k = 0; k < 2 + nsamps; ++k) {
        if ((f >> k) & 1u) {
            const int64_t row = bases[k] + pos[(int64_t)k * N + i];
            src_index[row] = (int32_t)i;
            kind[row] = (uint8_t)k;    // 0 old | 1 + s child of sample s | 1 + nsamps duplicate
        }
    }
}

__global__ __launch_bounds__(256) void refine_geometry_kernel(int64_t n_out, const int32_t *__restrict__ src_index,
                                                             const uint8_t *__restrict__ kind, const uint8_t *__restrict__ flags,
                                                             const float *__restrict__ means, const float *__restrict__ scales,
                                                             const float *__restrict__ quats, const RefineCfg c,
                                                             float *__restrict__ out_means, float *__restrict__ out_scales) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_out) return;
    const int64_t p = src_index[r];
    const int k = kind[r];
    float m[3], sl[3], sl_cur[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { m[j] = means[p * 3 + j]; sl[j] = scales[p * 3 + j]; }
    const bool split = (flags[p] & 0x80u) != 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) sl_cur[j] = split ? logf(expf(sl[j]) / kSplitShrink) : sl[j];
    float om[3] = {m[0], m[1], m[2]};
    if (k >= 1 && (k <= c.nsamps || c.clone_sample_means)) {
        float z[3];
        refine_normal3(c, (uint32_t)p, (uint32_t)(k - 1), z);
        const float4 q = reinterpret_cast<const float4 *>(quats)[p];
        if (k <= c.nsamps) sample_mean(m, sl, q, z, om);     // children: the parent's ORIGINAL scales spread the samples
        else sample_mean(m, sl_cur, q, z, om);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        out_means[r * 3 + j] = om[j];
        out_scales[r * 3 + j] = k == 0 ? sl[j] : sl_cur[j];
    }
}

// dst[r, :] = src[src_index[r], :]  (zero_new: rows that are not old rows become 0 -- optimizer moments of new Gaussians)
__global__ __launch_bounds__(256) void refine_rows_kernel(int64_t n_out, int64_t w, const float *__restrict__ src,
                                                         const int32_t *__restrict__ src_index, const uint8_t *__restrict__ kind,
                                                         int zero_new, float *__restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_out * w) return;
    const int64_t r = e / w, col = e - r * w;
    dst[e] = (zero_new && kind[r] != 0) ? 0.f : src[(int64_t)src_index[r] * w + col];
}

RefineCfg make_cfg(const float *th, const int *opt, uint64_t seed, int64_t step) {
    RefineCfg c;
    c.grad_thresh = th[0]; c.size_thresh = th[1]; c.split_screen_size = th[2]; c.cull_alpha_thresh = th[3];
    c.cull_scale_thresh = th[4]; c.cull_screen_size = th[5];
    c.nsamps = opt[0]; c.use_screen_split = opt[1]; c.cull_big = opt[2]; c.cull_screen = opt[3]; c.clone_sample_means = opt[4];
    c.seed_lo = (uint32_t)seed; c.seed_hi = (uint32_t)(seed >> 32); c.step = (uint32_t)step;
    return c;
}

}  // namespace

extern "C" int mtgs_refine_classify(int64_t N, const float *means, const float *scales, const float *quats,
                                    const float *opacities, const float *grad_norm, const float *vis_counts,
                                    const float *max_2dsize, const float *thresholds, const int *options, uint64_t seed,
                                    int64_t step, int32_t *counts, uint8_t *flags, void *stream) {
    MTGS_REQUIRE(N >= 0 && N < ((int64_t)1 << 31) && thresholds && options, MTGS_EINVAL, "mtgs_refine_classify: bad arguments");
    MTGS_REQUIRE(options[0] >= 1 && options[0] <= MAX_SAMPS, MTGS_EUNSUPPORTED, "mtgs_refine_classify: n_split_samples=%d (1..%d)",
                 options[0], MAX_SAMPS);
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(means && scales && quats && opacities && grad_norm && vis_counts && max_2dsize && counts && flags, MTGS_EINVAL,
                 "mtgs_refine_classify: null pointer");
    refine_classify_kernel<<<(unsigned)ceil_div64(N, 256), 256, 0, (hipStream_t)stream>>>(
        N, means, scales, quats, opacities, grad_norm, vis_counts, max_2dsize, make_cfg(thresholds, options, seed, step), counts, flags);
    MTGS_CHECK_LAUNCH("mtgs_refine_classify");
    return MTGS_OK;
}

extern "C" int mtgs_refine_apply(int64_t N, int64_t n_out, const uint8_t *flags, const int64_t *pos, const int64_t *bases,
                                 const float *means, const float *scales, const float *quats, const float *thresholds,
                                 const int *options, uint64_t seed, int64_t step, int32_t *src_index, uint8_t *kind,
                                 float *out_means, float *out_scales, void *stream) {
    MTGS_REQUIRE(N >= 0 && n_out >= 0 && n_out < ((int64_t)1 << 31) && thresholds && options, MTGS_EINVAL, "mtgs_refine_apply: bad arguments");
    if (N == 0 || n_out == 0) return MTGS_OK;
    MTGS_REQUIRE(flags && pos && bases && means && scales && quats && src_index && kind && out_means && out_scales, MTGS_EINVAL,
                 "mtgs_refine_apply: null pointer");
    const RefineCfg c = make_cfg(thresholds, options, seed, step);
    hipStream_t st = (hipStream_t)stream;
    refine_index_kernel<<<(unsigned)ceil_div64(N, 256), 256, 0, st>>>(N, c.nsamps, flags, pos, bases, src_index, kind);
    refine_geometry_kernel<<<(unsigned)ceil_div64(n_out, 256), 256, 0, st>>>(n_out, src_index, kind, flags, means, scales, quats, c,
                                                                          out_means, out_scales);
    MTGS_CHECK_LAUNCH("mtgs_refine_apply");
    return MTGS_OK;
}

extern "C" int mtgs_refine_rows(int64_t n_out, int64_t width, const float *src, const int32_t *src_index, const uint8_t *kind,
                                int zero_new, float *dst, void *stream) {
    MTGS_REQUIRE(n_out >= 0 && width >= 0, MTGS_EINVAL, "mtgs_refine_rows: bad sizes");
    if (n_out * width == 0) return MTGS_OK;
    MTGS_REQUIRE(src && src_index && kind && dst, MTGS_EINVAL, "mtgs_refine_rows: null pointer");
    refine_rows_kernel<<<(unsigned)ceil_div64(n_out * width, 256), 256, 0, (hipStream_t)stream>>>(n_out, width, src, src_index, kind,
                                                                                               zero_new, dst);
    MTGS_CHECK_LAUNCH("mtgs_refine_rows");
    return MTGS_OK;
}
