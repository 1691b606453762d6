// head.hip -- what MTGS computes from the rasterizer's output before the losses, forward and backward, in one kernel per
// direction (background composite, clamp, exposure affine, depth fill, normal image).
//
// Restates /root/reference/mtgs/scene_model/mtgs_scene_graph.py:672-690 and the shipped appearance model
// LearnableExposureRGBModel.forward (/root/reference/mtgs/scene_model/module/appearance.py:73-87):
//     rgb            = clamp(render[..., :3] + (1 - alpha) * background, 0, 1)
//     rgb_appearance = clamp(rgb @ E[:3, :3] + E[:3, 3], 0, 1)                     E = exposure_factor[camera] [3,4]
//     depth          = where(alpha > 0, render[..., -1], render[..., -1].detach().max())          (RGB+ED)
//     normal         = (n / |n| + 1) / 2,   n = render[..., 3:6]                                  (predict_normals)
// PyTorch runs this as ~20 launches forward and ~35 backward (every slice of `render` costs a zero-filled full-size
// gradient and an add); at MTGS's training resolution the iteration is bound by exactly that dispatch cost.
// Here: thread per pixel; the backward recomputes the two clamp masks, writes v_render (every channel, zeros where
// nothing arrived) and v_alpha once, and reduces the gradients of the background (3) and of E (12) through per-block
// partial sums in a fixed order.  Roofline: HBM (image-sized streams).
#include "common.hpp"

namespace {
constexpr int HEAD_BLOCK = 256, HEAD_RED = 15;   // 3 background + 12 exposure partial sums per block
struct HeadCfg {
    int D;             // channels of render
    int depth_ch;      // index of the depth channel or -1
    int normal_ch;     // first of the three normal channels or -1
};

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
__device__ __forceinline__ bool in01(float x) { return x >= 0.f && x <= 1.f; }   // torch.clamp passes the gradient inclusively

__global__ __launch_bounds__(HEAD_BLOCK) void head_fwd_kernel(int64_t P, HeadCfg cfg, const float *__restrict__ render,
                                                              const float *__restrict__ alpha, const float *__restrict__ bg,
                                                              const float *__restrict__ E, const float *__restrict__ depth_max,
                                                              float *__restrict__ rgb, float *__restrict__ app,
                                                              float *__restrict__ depth, float *__restrict__ normal) {
    const int64_t p = (int64_t)blockIdx.x * HEAD_BLOCK + threadIdx.x;
    if (p >= P) return;
    const float *r = render + p * cfg.D;
    const float a = alpha[p], t = 1.f - a;
    const float c0 = clamp01(r[0] + t * bg[0]), c1 = clamp01(r[1] + t * bg[1]), c2 = clamp01(r[2] + t * bg[2]);
    rgb[p * 3] = c0; rgb[p * 3 + 1] = c1; rgb[p * 3 + 2] = c2;
    if (app) {   // row vector times E[:3,:3] (row-major [3,4]) plus the last column
        app[p * 3] = clamp01(((c0 * E[0] + c1 * E[4]) + c2 * E[8]) + E[3]);
        app[p * 3 + 1] = clamp01(((c0 * E[1] + c1 * E[5]) + c2 * E[9]) + E[7]);
        app[p * 3 + 2] = clamp01(((c0 * E[2] + c1 * E[6]) + c2 * E[10]) + E[11]);
    }
    if (depth) depth[p] = a > 0.f ? r[cfg.depth_ch] : depth_max[0];
    if (normal) {
        const float nx = r[cfg.normal_ch], ny = r[cfg.normal_ch + 1], nz = r[cfg.normal_ch + 2];
        const float inv = 1.0f / sqrtf((nx * nx + ny * ny) + nz * nz);   // no epsilon in the reference: 0/0 = NaN where nothing was hit
        normal[p * 3] = (nx * inv + 1.f) * 0.5f; normal[p * 3 + 1] = (ny * inv + 1.f) * 0.5f; normal[p * 3 + 2] = (nz * inv + 1.f) * 0.5f;
    }
}

__device__ __forceinline__ float block_sum(float v, float *lds) {
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

__global__ __launch_bounds__(HEAD_BLOCK) void head_bwd_kernel(int64_t P, HeadCfg cfg, const float *__restrict__ render,
                                                              const float *__restrict__ alpha, const float *__restrict__ bg,
                                                              const float *__restrict__ E, const float *__restrict__ v_rgb,
                                                              const float *__restrict__ v_app, const float *__restrict__ v_depth,
                                                              const float *__restrict__ v_normal, float *__restrict__ v_render,
                                                              float *__restrict__ v_alpha, float *__restrict__ partials) {
    __shared__ float s_red[4];
    const int64_t p = (int64_t)blockIdx.x * HEAD_BLOCK + threadIdx.x;
    float red[HEAD_RED];
#pragma unroll
    for (int k = 0; k < HEAD_RED; ++k) red[k] = 0.f;
    if (p < P) {
        const float *r = render + p * cfg.D;
        float *o = v_render + p * cfg.D;
        const float a = alpha[p], t = 1.f - a;
        const float x0 = r[0] + t * bg[0], x1 = r[1] + t * bg[1], x2 = r[2] + t * bg[2];
        const float c0 = clamp01(x0), c1 = clamp01(x1), c2 = clamp01(x2);
        float g0 = v_rgb ? v_rgb[p * 3] : 0.f, g1 = v_rgb ? v_rgb[p * 3 + 1] : 0.f, g2 = v_rgb ? v_rgb[p * 3 + 2] : 0.f;
        if (E && v_app) {
            const float y0 = ((c0 * E[0] + c1 * E[4]) + c2 * E[8]) + E[3];
            const float y1 = ((c0 * E[1] + c1 * E[5]) + c2 * E[9]) + E[7];
            const float y2 = ((c0 * E[2] + c1 * E[6]) + c2 * E[10]) + E[11];
            const float h0 = in01(y0) ? v_app[p * 3] : 0.f, h1 = in01(y1) ? v_app[p * 3 + 1] : 0.f, h2 = in01(y2) ? v_app[p * 3 + 2] : 0.f;
            g0 += (h0 * E[0] + h1 * E[1]) + h2 * E[2];
            g1 += (h0 * E[4] + h1 * E[5]) + h2 * E[6];
            g2 += (h0 * E[8] + h1 * E[9]) + h2 * E[10];
            // d E[i][j] = rgb[i] * h[j],  d E[j][3] = h[j]      (partials 3 .. 14 = E row-major)
            red[3] = c0 * h0; red[4] = c0 * h1; red[5] = c0 * h2; red[6] = h0;
            red[7] = c1 * h0; red[8] = c1 * h1; red[9] = c1 * h2; red[10] = h1;
            red[11] = c2 * h0; red[12] = c2 * h1; red[13] = c2 * h2; red[14] = h2;
        }
        g0 = in01(x0) ? g0 : 0.f; g1 = in01(x1) ? g1 : 0.f; g2 = in01(x2) ? g2 : 0.f;
        red[0] = t * g0; red[1] = t * g1; red[2] = t * g2;
        for (int k = 0; k < cfg.D; ++k) o[k] = 0.f;
        o[0] = g0; o[1] = g1; o[2] = g2;
        v_alpha[p] = -((g0 * bg[0] + g1 * bg[1]) + g2 * bg[2]);
        if (cfg.depth_ch >= 0 && v_depth) o[cfg.depth_ch] = a > 0.f ? v_depth[p] : 0.f;
        if (cfg.normal_ch >= 0 && v_normal) {
            const float nx = r[cfg.normal_ch], ny = r[cfg.normal_ch + 1], nz = r[cfg.normal_ch + 2];
            const float inv = 1.0f / sqrtf((nx * nx + ny * ny) + nz * nz);
            const float ux = nx * inv, uy = ny * inv, uz = nz * inv;
            const float w0 = 0.5f * v_normal[p * 3], w1 = 0.5f * v_normal[p * 3 + 1], w2 = 0.5f * v_normal[p * 3 + 2];
            const float dot = (ux * w0 + uy * w1) + uz * w2;
            // a zero cotangent gives a zero gradient even where n = 0 (0 / 0 normal): a caller that drops a non-finite
            // normal term with torch.where instead of MTGS's host-side `if isfinite` (:939) must not get NaN back
            if (w0 != 0.f || w1 != 0.f || w2 != 0.f) {
                o[cfg.normal_ch] = (w0 - ux * dot) * inv; o[cfg.normal_ch + 1] = (w1 - uy * dot) * inv; o[cfg.normal_ch + 2] = (w2 - uz * dot) * inv;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < HEAD_RED; ++k) {
        const float s = block_sum(red[k], s_red);
        if (threadIdx.x == 0) partials[(int64_t)blockIdx.x * HEAD_RED + k] = s;
    }
}

// v_bg[3], v_E[12]: column sums of the partials in a fixed order
__global__ __launch_bounds__(HEAD_BLOCK) void head_finish_kernel(int64_t nblocks, const float *__restrict__ partials,
                                                                 float *__restrict__ v_bg, float *__restrict__ v_E) {
    __shared__ float s_red[4];
    const int k = blockIdx.x;      // one workgroup per column (15 columns one after the other in ONE workgroup: 19 us of latency)
    float s = 0.f;
    for (int64_t b = threadIdx.x; b < nblocks; b += HEAD_BLOCK) s += partials[b * HEAD_RED + k];
    const float tot = block_sum(s, s_red);
    if (threadIdx.x == 0) {
        if (k < 3) { if (v_bg) v_bg[k] = tot; }
        else if (v_E) v_E[k - 3] = tot;
    }
}
}  // namespace

extern "C" int mtgs_head_fwd(int width, int height, int channels, int depth_channel, int normal_channel, const float *render,
                             const float *alpha, const float *background, const float *exposure, const float *depth_max,
                             float *rgb, float *rgb_appearance, float *depth, float *normal, void *stream) {
    MTGS_REQUIRE(width > 0 && height > 0 && channels >= 3 && depth_channel < channels && normal_channel + 2 < channels, MTGS_EINVAL,
                 "mtgs_head_fwd: bad sizes");
    MTGS_REQUIRE(render && alpha && background && rgb, MTGS_EINVAL, "mtgs_head_fwd: null pointer");
    MTGS_REQUIRE((!rgb_appearance || exposure) && (!depth || (depth_channel >= 0 && depth_max)) && (!normal || normal_channel >= 0),
                 MTGS_EINVAL, "mtgs_head_fwd: an output was requested without its input");
    const int64_t P = (int64_t)width * height;
    head_fwd_kernel<<<(unsigned)ceil_div64(P, HEAD_BLOCK), HEAD_BLOCK, 0, (hipStream_t)stream>>>(
        P, HeadCfg{channels, depth_channel, normal_channel}, render, alpha, background, exposure, depth_max, rgb, rgb_appearance,
        depth, normal);
    MTGS_CHECK_LAUNCH("mtgs_head_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_head_workspace_floats(int width, int height, size_t *n) {
    MTGS_REQUIRE(width > 0 && height > 0 && n, MTGS_EINVAL, "mtgs_head_workspace_floats: bad arguments");
    *n = (size_t)ceil_div64((int64_t)width * height, HEAD_BLOCK) * HEAD_RED;
    return MTGS_OK;
}

extern "C" int mtgs_head_bwd(int width, int height, int channels, int depth_channel, int normal_channel, const float *render,
                             const float *alpha, const float *background, const float *exposure, const float *v_rgb,
                             const float *v_rgb_appearance, const float *v_depth, const float *v_normal, float *v_render,
                             float *v_alpha, float *v_background, float *v_exposure, float *partials, void *stream) {
    MTGS_REQUIRE(width > 0 && height > 0 && channels >= 3 && depth_channel < channels && normal_channel + 2 < channels, MTGS_EINVAL,
                 "mtgs_head_bwd: bad sizes");
    MTGS_REQUIRE(render && alpha && background && v_render && v_alpha && partials, MTGS_EINVAL, "mtgs_head_bwd: null pointer");
    const int64_t P = (int64_t)width * height, nblocks = ceil_div64(P, HEAD_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    head_bwd_kernel<<<(unsigned)nblocks, HEAD_BLOCK, 0, st>>>(P, HeadCfg{channels, depth_channel, normal_channel}, render, alpha,
                                                              background, exposure, v_rgb, v_rgb_appearance, v_depth, v_normal,
                                                              v_render, v_alpha, partials);
    if (v_background || v_exposure) head_finish_kernel<<<HEAD_RED, HEAD_BLOCK, 0, st>>>(nblocks, partials, v_background, v_exposure);
    MTGS_CHECK_LAUNCH("mtgs_head_bwd");
    return MTGS_OK;
}

// ---- camera position of a view matrix [A t; 0 1]: c = -A^-1 t (the translation column of its inverse), one thread.
// gsplat's sh_degree path forms torch.inverse(viewmats)[:, :3, 3] (rendering.py) -- an LU factorisation, two triangular solves,
// a row swap and their backward: ~25 launches of one-element kernels, 110 us of a 1.2 ms step.  A = general 3x3 (adjugate).
// bwd: with u = A^-T v_c:  v_A = -u c^T,  v_t = -u  (c = -A^-1 t  =>  dc = -A^-1 dA c - A^-1 dt), last row as torch.inverse's.
namespace {
struct Inv3 { float m[9]; };
__device__ __forceinline__ Inv3 inverse3(const float *V) {     // V: row-major 4x4, A = V[:3, :3]
    const float a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[6], g = V[8], h = V[9], i = V[10];
    const float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const float det = (a * c00 + b * c01) + c * c02;
    const float r = 1.0f / det;
    Inv3 o;
    o.m[0] = c00 * r; o.m[1] = (c * h - b * i) * r; o.m[2] = (b * f - c * e) * r;
    o.m[3] = c01 * r; o.m[4] = (a * i - c * g) * r; o.m[5] = (c * d - a * f) * r;
    o.m[6] = c02 * r; o.m[7] = (b * g - a * h) * r; o.m[8] = (a * e - b * d) * r;
    return o;
}
__global__ void campos_fwd_kernel(const float *__restrict__ V, float *__restrict__ out) {
    if (threadIdx.x != 0) return;
    const Inv3 I = inverse3(V);
    const float tx = V[3], ty = V[7], tz = V[11];
    out[0] = -((I.m[0] * tx + I.m[1] * ty) + I.m[2] * tz);
    out[1] = -((I.m[3] * tx + I.m[4] * ty) + I.m[5] * tz);
    out[2] = -((I.m[6] * tx + I.m[7] * ty) + I.m[8] * tz);
}
__global__ void campos_bwd_kernel(const float *__restrict__ V, const float *__restrict__ v_c, float *__restrict__ v_V) {
    if (threadIdx.x != 0) return;
    const Inv3 I = inverse3(V);
    const float tx = V[3], ty = V[7], tz = V[11];
    const float c[3] = {-((I.m[0] * tx + I.m[1] * ty) + I.m[2] * tz), -((I.m[3] * tx + I.m[4] * ty) + I.m[5] * tz),
                        -((I.m[6] * tx + I.m[7] * ty) + I.m[8] * tz)};
    float u[3];      // A^-T v_c
    for (int k = 0; k < 3; ++k) u[k] = (I.m[k] * v_c[0] + I.m[3 + k] * v_c[1]) + I.m[6 + k] * v_c[2];
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) v_V[r * 4 + k] = -u[r] * c[k];
        v_V[r * 4 + 3] = -u[r];
    }
    // torch.inverse treats all 16 entries as free: v_V = -C^T G C^T with C = V^-1 and G zero but for its column [v_c; 0] has the
    // last row -(c . v_c) [c; 1] as well (the caller's view matrix has a constant last row; the gradient is reported as gsplat does)
    const float cv = (c[0] * v_c[0] + c[1] * v_c[1]) + c[2] * v_c[2];
    v_V[12] = -cv * c[0]; v_V[13] = -cv * c[1]; v_V[14] = -cv * c[2]; v_V[15] = -cv;
}
}  // namespace

extern "C" int mtgs_campos_fwd(const float *viewmat, float *cam_pos, void *stream) {
    MTGS_REQUIRE(viewmat && cam_pos, MTGS_EINVAL, "mtgs_campos_fwd: null pointer");
    campos_fwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(viewmat, cam_pos);
    MTGS_CHECK_LAUNCH("mtgs_campos_fwd");
    return MTGS_OK;
}
extern "C" int mtgs_campos_bwd(const float *viewmat, const float *v_cam_pos, float *v_viewmat, void *stream) {
    MTGS_REQUIRE(viewmat && v_cam_pos && v_viewmat, MTGS_EINVAL, "mtgs_campos_bwd: null pointer");
    campos_bwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(viewmat, v_cam_pos, v_viewmat);
    MTGS_CHECK_LAUNCH("mtgs_campos_bwd");
    return MTGS_OK;
}
