// project_bwd.hip -- VJP of the per-Gaussian 3D->2D projection.
//
// Replaces gsplat 1.4.0 fully_fused_projection_bwd (pinhole, packed=False), reached through
// gsplat.rendering.rasterization (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662).
// Compiled WITH fp contraction (unlike the forward): gradients are compared at 2e-3, not bit-exact.
//
// Roofline: HBM for the dense part (radii read, 44 B of gradients written per Gaussian, zeros for the
// invisible ones) + ~1500 flops per VISIBLE Gaussian.
#include "project_common.hpp"
#include "raster_rec.hpp"

namespace {

// Sum 12 values (v_R, v_t) over the block into the block's LDS accumulator acc[12] (thread k owns
// acc[k]).  The accumulator is flushed to v_viewmats ONCE per block at the end of the kernel: the 12
// target words are shared by every block, and same-address fp32 atomics serialise at ~12 ns each.
__device__ __forceinline__ void block_reduce_viewmat(float (&vals)[12], float *__restrict__ acc,
                                                     float *lds /* [4][12] */) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float v = wave_sum_to_lane63(vals[k]);
        if (lane == 63) lds[wave * 12 + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int k = threadIdx.x;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w) v += lds[w * 12 + k];
        acc[k] += v;
    }
}
// row strides (floats) of the incoming gradients: dense gsplat arrays, or views of an interleaved buffer
struct ProjGradStrides { int64_t means2d, depths, conics, compensations, opac_eff; };
// Optional "densify" by-product for callers that keep the compositing backward's gradients in COMPACT rows
// (one row per visible Gaussian, grad_row_index): the per-Gaussian gradients that leave the rasterizer --
// v_means2d (for retain_grad), |v_means2d| (absgrad) and v_colors -- are written as dense arrays, zeros for the
// culled Gaussians, while their rows are in registers / cache anyway.
struct ProjExpand {
    const float *abs_src, *col_src;   // row-strided sources (same row index as the gradients); nullable
    int64_t abs_stride, col_stride;   // in floats
    int channels;
    float *means2d, *means2d_abs, *colors;  // dense [C,N,2] [C,N,2] [C,N,channels]; nullable
};
constexpr int PROJ_MAX_CAMS = 64;  // cameras whose v_viewmat is accumulated in LDS (MTGS: 1)

// Incoming gradients and saved forward values of ONE (camera, Gaussian) pair
struct PairIn {
    float conic[3], v_conic[3];
    float2 v_mean2d;
    float v_depth, comp, v_comp, opac, v_opac_eff;
    bool has_comp, has_vcomp, has_opac;
};
// VJP of the projection of one visible (camera, Gaussian) pair: adds to am / aq / as / ao (gradients of
// mean, quaternion, scale, opacity) and returns v_R (9) | v_t (3) in vRt.
__device__ __forceinline__ void project_vjp_pair(const float (&m)[3], const float4 q, const float (&sc)[3], const Cam &cam,
                                                 int W, int H, float eps2d, const PairIn &in, float (&am)[3],
                                                 float (&aq)[4], float (&as)[3], float &ao, float (&vRt)[12]) {
    ProjState s;
    proj_common(m, q, sc, cam, W, H, s);
    const float va = in.v_conic[0], vb = 0.5f * in.v_conic[1], vc = in.v_conic[2];
    const float a = in.conic[0], b = in.conic[1], cc = in.conic[2];
    const float t00 = a * va + b * vb, t01 = a * vb + b * vc, t10 = b * va + cc * vb, t11 = b * vb + cc * vc;
    float vcov[4];
    vcov[0] = -(t00 * a + t01 * b); vcov[1] = -(t00 * b + t01 * cc);
    vcov[2] = -(t10 * a + t11 * b); vcov[3] = -(t10 * b + t11 * cc);
    // opac_eff = opacity * compensation: the product rule feeds the compensation VJP
    if (in.has_opac) ao += in.v_opac_eff * (in.has_comp ? in.comp : 1.f);
    if (in.has_comp && (in.has_vcomp || in.has_opac)) {
        const float comp = in.comp;
        const float vcomp = (in.has_vcomp ? in.v_comp : 0.f) + (in.has_opac ? in.v_opac_eff * in.opac : 0.f);
        const float det_conic = a * cc - b * b;
        const float v_sqr = vcomp * 0.5f / (comp + kCompEps);
        const float omc = 1.f - comp * comp;
        vcov[0] += v_sqr * (omc * a - eps2d * det_conic);
        vcov[1] += v_sqr * (omc * b);
        vcov[2] += v_sqr * (omc * b);
        vcov[3] += v_sqr * (omc * cc - eps2d * det_conic);
    }
    const float *J = s.J;
    const float x = s.mean_c[0], y = s.mean_c[1];
    const float rz = s.rz, rz2 = s.rz2, rz3 = rz2 * rz, tx = s.tx, ty = s.ty;
    const float2 vm2 = in.v_mean2d;
    float G[6], G2[6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            G[i * 3 + j] = vcov[i * 2] * J[j] + vcov[i * 2 + 1] * J[3 + j];
            G2[i * 3 + j] = vcov[i] * J[j] + vcov[2 + i] * J[3 + j];  // vcov^T * J
        }
    float v_covar_c[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v_covar_c[i * 3 + j] = J[i] * G[j] + J[3 + i] * G[3 + j];
    float v_mean_c[3];
    v_mean_c[0] = cam.fx * rz * vm2.x;
    v_mean_c[1] = cam.fy * rz * vm2.y;
    v_mean_c[2] = -(cam.fx * x * vm2.x + cam.fy * y * vm2.y) * rz2;
    float vJ[6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float p = (G[i * 3] * s.covar_c[j * 3] + G[i * 3 + 1] * s.covar_c[j * 3 + 1]) + G[i * 3 + 2] * s.covar_c[j * 3 + 2];
            const float qq = (G2[i * 3] * s.covar_c[j] + G2[i * 3 + 1] * s.covar_c[3 + j]) + G2[i * 3 + 2] * s.covar_c[6 + j];
            vJ[i * 3 + j] = p + qq;
        }
    if (!s.x_clamped) v_mean_c[0] += -cam.fx * rz2 * vJ[2];
    else v_mean_c[2] += -cam.fx * rz3 * vJ[2] * tx;
    if (!s.y_clamped) v_mean_c[1] += -cam.fy * rz2 * vJ[5];
    else v_mean_c[2] += -cam.fy * rz3 * vJ[5] * ty;
    v_mean_c[2] += ((-cam.fx * rz2 * vJ[0] - cam.fy * rz2 * vJ[4]) + 2.f * cam.fx * tx * rz3 * vJ[2]) + 2.f * cam.fy * ty * rz3 * vJ[5];
    v_mean_c[2] += in.v_depth;
    const float *R = cam.R;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) vRt[i * 3 + j] = v_mean_c[i] * m[j];
        vRt[9 + i] = v_mean_c[i];
        am[i] += (R[i] * v_mean_c[0] + R[3 + i] * v_mean_c[1]) + R[6 + i] * v_mean_c[2];
    }
    float RC[9], RCt[9], tmp[9], tmp2[9], vcT[9];
    mm3(R, s.covar, RC);
    mm3_bt(R, s.covar, RCt);
    mm3(v_covar_c, RCt, tmp);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) vcT[i * 3 + j] = v_covar_c[j * 3 + i];
    mm3(vcT, RC, tmp2);
#pragma unroll
    for (int i = 0; i < 9; ++i) vRt[i] += tmp[i] + tmp2[i];
    float v_covar[9];
    mm3_at(R, v_covar_c, tmp);
    mm3(tmp, R, v_covar);
    float sym[9], vM[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) sym[i * 3 + j] = v_covar[i * 3 + j] + v_covar[j * 3 + i];
    mm3(sym, s.Mq, vM);
    float Gq[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        as[j] += (s.Rq[j] * vM[j] + s.Rq[3 + j] * vM[3 + j]) + s.Rq[6 + j] * vM[6 + j];
#pragma unroll
        for (int i = 0; i < 3; ++i) Gq[i * 3 + j] = vM[i * 3 + j] * sc[j];
    }
    const float w = s.qn[0], qx = s.qn[1], qy = s.qn[2], qz = s.qn[3];
    float vqn[4];
    vqn[0] = 2.f * ((qx * (Gq[7] - Gq[5]) + qy * (Gq[2] - Gq[6])) + qz * (Gq[3] - Gq[1]));
    vqn[1] = 2.f * (((-2.f * qx * (Gq[4] + Gq[8]) + qy * (Gq[1] + Gq[3])) + qz * (Gq[2] + Gq[6])) + w * (Gq[7] - Gq[5]));
    vqn[2] = 2.f * (((qx * (Gq[1] + Gq[3]) - 2.f * qy * (Gq[0] + Gq[8])) + qz * (Gq[5] + Gq[7])) + w * (Gq[2] - Gq[6]));
    vqn[3] = 2.f * (((qx * (Gq[2] + Gq[6]) + qy * (Gq[5] + Gq[7])) - 2.f * qz * (Gq[0] + Gq[4])) + w * (Gq[3] - Gq[1]));
    const float dot = ((vqn[0] * w + vqn[1] * qx) + vqn[2] * qy) + vqn[3] * qz;
#pragma unroll
    for (int k = 0; k < 4; ++k) aq[k] += (vqn[k] - dot * s.qn[k]) * s.inv_norm;
}

// One thread per Gaussian, looping over cameras so that v_means / v_quats / v_scales are written
// (not accumulated) exactly once.  MTGS always has C = 1.
__global__ __launch_bounds__(PROJ_BLOCK) void project_bwd_kernel(
    int C, int64_t N, const float *__restrict__ means, const float *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ viewmats,
    const float *__restrict__ Ks, int W, int H, float eps2d, const int32_t *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ compensations,
    const float *__restrict__ opacities, const float *__restrict__ v_means2d,
    const float *__restrict__ v_depths, const float *__restrict__ v_conics,
    const float *__restrict__ v_compensations, const float *__restrict__ v_opac_eff,
    float *__restrict__ v_means, float *__restrict__ v_quats, float *__restrict__ v_scales,
    float *__restrict__ v_viewmats, float *__restrict__ v_opacities, const ProjGradStrides gs,
    const int32_t *__restrict__ row_index, const ProjExpand ex) {
    __shared__ float red[(PROJ_BLOCK / 64) * 12];
    __shared__ float s_acc[PROJ_MAX_CAMS * 12];
    __shared__ int s_list[PROJ_BLOCK];
    __shared__ int s_wcnt[PROJ_BLOCK / 64];
    for (int k = threadIdx.x; k < PROJ_MAX_CAMS * 12; k += PROJ_BLOCK) s_acc[k] = 0.f;
    for (int64_t chunk = (int64_t)blockIdx.x * PROJ_BLOCK; chunk < N; chunk += (int64_t)gridDim.x * PROJ_BLOCK) {
    __syncthreads();  // s_list / s_wcnt / red reuse, s_acc initialisation
    // Only ~15 % of the Gaussians are visible and they are scattered over the index range: one thread
    // per Gaussian would leave most lanes of every wave idle behind the ~1500-instruction VJP.  The
    // block therefore (1) writes zeros for its invisible Gaussians, (2) compacts the visible ones
    // (ballot + mbcnt) and (3) lets thread j process the j-th visible one: dense waves, idle waves skip.
    const int64_t n_own = chunk + threadIdx.x;
    bool vis = false;
    if (n_own < N)
        for (int c = 0; c < C; ++c) {
            const int64_t idx = (int64_t)c * N + n_own;
            const bool v = radii[idx] > 0;
            vis = vis || v;
            if (!v) {  // dense by-products of a culled (camera, Gaussian) pair
                if (ex.means2d) reinterpret_cast<float2 *>(ex.means2d)[idx] = make_float2(0.f, 0.f);
                if (ex.means2d_abs) reinterpret_cast<float2 *>(ex.means2d_abs)[idx] = make_float2(0.f, 0.f);
                if (ex.colors)
                    for (int k = 0; k < ex.channels; ++k) ex.colors[idx * ex.channels + k] = 0.f;
            }
        }
    if (n_own < N && !vis) {
        v_means[n_own * 3] = 0.f; v_means[n_own * 3 + 1] = 0.f; v_means[n_own * 3 + 2] = 0.f;
        reinterpret_cast<float4 *>(v_quats)[n_own] = make_float4(0.f, 0.f, 0.f, 0.f);
        v_scales[n_own * 3] = 0.f; v_scales[n_own * 3 + 1] = 0.f; v_scales[n_own * 3 + 2] = 0.f;
        if (v_opacities) v_opacities[n_own] = 0.f;
    }
    const unsigned long long vmask = __ballot(vis);
    const int wave_id = threadIdx.x >> 6;
    if (lane_id() == 0) s_wcnt[wave_id] = __popcll(vmask);
    __syncthreads();
    int wbase = 0, count = 0;
#pragma unroll
    for (int w = 0; w < PROJ_BLOCK / 64; ++w) {
        if (w < wave_id) wbase += s_wcnt[w];
        count += s_wcnt[w];
    }
    if (vis)
        s_list[wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vmask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned)vmask, 0u))] = threadIdx.x;
    __syncthreads();
    if (count == 0) continue;
    const bool live = (int)threadIdx.x < count;
    const int64_t n = chunk + (live ? s_list[threadIdx.x] : 0);
    float am[3] = {0.f, 0.f, 0.f}, aq[4] = {0.f, 0.f, 0.f, 0.f}, as[3] = {0.f, 0.f, 0.f}, ao = 0.f;
    const float opac = (live && v_opac_eff) ? opacities[n] : 0.f;
    float m[3] = {0.f, 0.f, 0.f}, sc[3] = {1.f, 1.f, 1.f};
    float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
    bool loaded = false;
    for (int c = 0; c < C; ++c) {
        float vRt[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) vRt[k] = 0.f;
        const int64_t idx = (int64_t)c * N + n;
        if (live && radii[idx] > 0) {
            if (!loaded) {
                m[0] = means[n * 3]; m[1] = means[n * 3 + 1]; m[2] = means[n * 3 + 2];
                q = reinterpret_cast<const float4 *>(quats)[n];
                sc[0] = scales[n * 3]; sc[1] = scales[n * 3 + 1]; sc[2] = scales[n * 3 + 2];
                loaded = true;
            }
            const Cam cam = load_cam(viewmats + c * 16, Ks + c * 9);
            const int64_t grow = row_index ? (int64_t)row_index[idx] : idx;  // row of the incoming gradients
            PairIn in;
            in.conic[0] = conics[idx * 3]; in.conic[1] = conics[idx * 3 + 1]; in.conic[2] = conics[idx * 3 + 2];
            const float *vcon = v_conics + grow * gs.conics;
            in.v_conic[0] = vcon[0]; in.v_conic[1] = vcon[1]; in.v_conic[2] = vcon[2];
            in.v_mean2d = make_float2(v_means2d[grow * gs.means2d], v_means2d[grow * gs.means2d + 1]);
            in.v_depth = v_depths[grow * gs.depths];
            in.has_comp = compensations != nullptr; in.has_vcomp = v_compensations != nullptr; in.has_opac = v_opac_eff != nullptr;
            in.comp = in.has_comp ? compensations[idx] : 1.f;
            in.v_comp = in.has_vcomp ? v_compensations[grow * gs.compensations] : 0.f;
            in.opac = opac;
            in.v_opac_eff = in.has_opac ? v_opac_eff[grow * gs.opac_eff] : 0.f;
            if (ex.means2d) reinterpret_cast<float2 *>(ex.means2d)[idx] = in.v_mean2d;
            if (ex.means2d_abs)
                reinterpret_cast<float2 *>(ex.means2d_abs)[idx] =
                    make_float2(ex.abs_src[grow * ex.abs_stride], ex.abs_src[grow * ex.abs_stride + 1]);
            if (ex.colors)
                for (int k = 0; k < ex.channels; ++k) ex.colors[idx * ex.channels + k] = ex.col_src[grow * ex.col_stride + k];
            project_vjp_pair(m, q, sc, cam, W, H, eps2d, in, am, aq, as, ao, vRt);
        }
        if (v_viewmats) {
            __syncthreads();
            block_reduce_viewmat(vRt, s_acc + (c % PROJ_MAX_CAMS) * 12, red);
            if (c >= PROJ_MAX_CAMS && threadIdx.x < 12) {  // (not reached by MTGS) flush immediately
                const int k = threadIdx.x;
                atomicAdd(v_viewmats + c * 16 + (k < 9 ? (k / 3) * 4 + (k % 3) : (k - 9) * 4 + 3), s_acc[(c % PROJ_MAX_CAMS) * 12 + k]);
                s_acc[(c % PROJ_MAX_CAMS) * 12 + k] = 0.f;
            }
        }
    }
    if (live) {
        v_means[n * 3] = am[0]; v_means[n * 3 + 1] = am[1]; v_means[n * 3 + 2] = am[2];
        reinterpret_cast<float4 *>(v_quats)[n] = make_float4(aq[0], aq[1], aq[2], aq[3]);
        v_scales[n * 3] = as[0]; v_scales[n * 3 + 1] = as[1]; v_scales[n * 3 + 2] = as[2];
        if (v_opacities) v_opacities[n] = ao;
    }
    }  // chunk loop
    __syncthreads();
    if (v_viewmats) {
        // v_R[i][j] -> viewmat[i][j], v_t[i] -> viewmat[i][3]
        const int ncam = C < PROJ_MAX_CAMS ? C : PROJ_MAX_CAMS;
        for (int e = threadIdx.x; e < ncam * 12; e += PROJ_BLOCK) {
            const int c = e / 12, k = e % 12;
            const float v = s_acc[e];
            if (v != 0.f) atomicAdd(v_viewmats + c * 16 + (k < 9 ? (k / 3) * 4 + (k % 3) : (k - 9) * 4 + 3), v);
        }
    }
}

// ---- compact path (C == 1, rows of the incoming gradients indexed by visible rank) ---------------------------
// The dense kernel above is bound by LATENCY, not bytes: after the in-block compaction one wave per block walks
// the ~2000-instruction VJP while three wait at the barriers, and every lane issues 13 partial-line stores.  When
// the caller has the list of visible Gaussians (mtgs_bin_compact's vis_ids / vis_rank) the work splits into
//   A. project_bwd_vis_kernel    one thread per VISIBLE Gaussian (dense waves, no barrier apart from the viewmat
//                                sum), results to 48-byte rows of a workspace;
//   B. project_bwd_expand_kernel a pure streaming pass: every dense output (v_means, v_quats, v_scales,
//                                v_opacities and the by-products) is staged through LDS and leaves as fully
//                                coalesced 16-byte stores, zeros for the culled Gaussians included.
constexpr int VIS_ROW = 12;  // floats per workspace row: v_mean 3 | v_quat 4 | v_scale 3 | v_opacity 1 | pad
// RAW rows of the packed compositing backward (blend.hip, mtgs_blend_bwd_packed): with h = vis * dL/dalpha (v_sigma = -opacity h)
//   row = {sum h dx, sum h dy | k sum |h u|, k sum |h w| | sum h dx^2, sum h dx dy, sum h dy^2 | sum h = v_opacity | colours ...},  k = log2(e)/2
// -> the gradients gsplat's rasterize_to_pixels_bwd sums per pixel: v_xy = -o conic (m1, m2), |v_xy| = o / k (A1, A2),
// v_conic = -o (m3 / 2, m4, m5 / 2), applied ONCE per Gaussian here instead of once per (tile, Gaussian) in the compositing
// kernel (9 -> 3 instructions per epilogue there, one fewer per slot).  o = the opacity that was blended (opacity x
// compensation, formed as front.hip forms it).  The converted values are written back: everything behind this kernel
// (dense by-products, densification statistics, the tests' row accounting) reads rows of the documented meaning.
struct RowGrads { float2 v_xy; float v_conic[3]; float2 v_abs; };
__device__ __forceinline__ RowGrads rows_to_gradients(float *__restrict__ row, const float ca, const float cb, const float cc,
                                                      const float o) {
    const float4 a = reinterpret_cast<const float4 *>(row)[0], b = reinterpret_cast<const float4 *>(row)[1];
    const float no = -o;
    RowGrads g;
    g.v_xy = make_float2(no * (ca * a.x + cb * a.y), no * (cb * a.x + cc * a.y));
    g.v_conic[0] = 0.5f * no * b.x; g.v_conic[1] = no * b.y; g.v_conic[2] = 0.5f * no * b.z;
    const float ok = o * MTGS_HALF_LOG2E_INV;     // (the absgrad sums carry the factor log2(e)/2 of the staged conic, blend.hip)
    g.v_abs = make_float2(ok * a.z, ok * a.w);
    reinterpret_cast<float4 *>(row)[0] = make_float4(g.v_xy.x, g.v_xy.y, g.v_abs.x, g.v_abs.y);
    reinterpret_cast<float4 *>(row)[1] = make_float4(g.v_conic[0], g.v_conic[1], g.v_conic[2], b.w);
    return g;
}
// ZEROED outputs (mtgs_project_bwd_zeroed): the dense gradients were cleared by the caller -- beside the compositing backward's own
// work, mtgs_blend_bwd_packed(also_zero): that kernel is VALU-bound and leaves 88 % of the HBM bandwidth idle -- and this kernel
// writes the values of the visible Gaussians that HAVE a gradient straight to their places (15 floats in six arrays, 6 % of the
// Gaussians at the headline scene); the streaming pass behind it (project_bwd_expand_kernel: every byte of every dense output, 85 %
// of them zeros) does not run.
struct ProjSparse {
    int on;
    float *v_means, *v_quats, *v_scales, *v_opacities;     // v_opacities nullable
    ProjExpand ex;                                          // dense by-products (nullable members) and their row sources
};
__global__ __launch_bounds__(PROJ_BLOCK) void project_bwd_vis_kernel(
    int64_t n_vis, const int32_t *__restrict__ vis_ids, const float *__restrict__ means,
    const float *__restrict__ quats, const float *__restrict__ scales, const float *__restrict__ viewmats,
    const float *__restrict__ Ks, int W, int H, float eps2d, const float *__restrict__ conics,
    const float *__restrict__ compensations, const float *__restrict__ opacities,
    const float *__restrict__ v_means2d, const float *__restrict__ v_depths, const float *__restrict__ v_conics,
    const float *__restrict__ v_compensations, const float *__restrict__ v_opac_eff, const ProjGradStrides gs,
    float *__restrict__ ws, float *__restrict__ v_viewmats, const int64_t *__restrict__ n_vis_dev,
    const float *__restrict__ x_quat_rows, const float *__restrict__ x_mean_rows, float *raw_rows, int64_t raw_stride,
    const float *__restrict__ recs /* nullable: mtgs_front_fwd's records, indexed like the rows */,
    float *__restrict__ vm_partials /* nullable: [gridDim.x, 12] -- the blocks' viewmat sums, added up by viewmat_from_partials */,
    const ProjSparse sp) {
    __shared__ float red[(PROJ_BLOCK / 64) * 12];
    __shared__ float s_acc[12];
    if (threadIdx.x < 12) s_acc[threadIdx.x] = 0.f;
    if (n_vis_dev) {   // the count lives on the device (front.hip's packed totals); n_vis is the capacity of the row buffers
        const int64_t d = *n_vis_dev >> 32;
        if (d < n_vis) n_vis = d;
    }
    const Cam cam = load_cam(viewmats, Ks);
    // Only the Gaussians something was composited from carry a gradient: the tile lists of an opaque scene terminate long
    // before their ends, and 60 % (headline scene) to 98 % (MTGS-like scenes at 960x540) of the frustum-visible rows are
    // exactly zero.  The VJP (~2000 instructions) of a zero row is zero: every thread first fetches ITS row's incoming
    // gradients (one 64-byte line), the block compacts the rows that have any (ballot + LDS list, as project_bwd_kernel does
    // for the visible ones) and runs the VJP with dense waves over those; the others get a zero result row.
    __shared__ int s_list[PROJ_BLOCK];
    __shared__ int s_wcnt[PROJ_BLOCK / 64];
    struct RowIn { float2 v_xy; float v_conic[3]; float v_depth, v_comp, v_opac_eff; };
    __shared__ RowIn s_in[PROJ_BLOCK];
    for (int64_t r0 = (int64_t)blockIdx.x * PROJ_BLOCK; r0 < n_vis; r0 += (int64_t)gridDim.x * PROJ_BLOCK) {
        __syncthreads();      // s_list / s_in reuse
        {
            const int64_t r = r0 + threadIdx.x;
            bool nz = false;
            if (r < n_vis) {
                RowIn ri = {};      // (a raw row of zeros leaves v_xy / v_conic unassigned below: they must then BE zero when the depth /
                //                      compensation / normal cotangents alone make the row non-zero)
                // (everything that is read from the gradient rows is read BEFORE the raw rows are rewritten in place: the
                //  pointers alias, and a load behind the store would wait for it)
                ri.v_depth = v_depths[r * gs.depths];
                ri.v_comp = v_compensations ? v_compensations[r * gs.compensations] : 0.f;
                ri.v_opac_eff = v_opac_eff ? v_opac_eff[r * gs.opac_eff] : 0.f;
                float2 by_abs = make_float2(0.f, 0.f);      // (zeroed outputs: the absgrad pair of this row, from whichever form the rows have)
                if (sp.on && sp.ex.means2d_abs && !raw_rows) by_abs = make_float2(sp.ex.abs_src[r * sp.ex.abs_stride], sp.ex.abs_src[r * sp.ex.abs_stride + 1]);
                if (raw_rows) {
                    const float4 *row = reinterpret_cast<const float4 *>(raw_rows + r * raw_stride);
                    const float4 a4 = row[0], b4 = row[1];
                    nz = a4.x != 0.f || a4.y != 0.f || a4.z != 0.f || a4.w != 0.f || b4.x != 0.f || b4.y != 0.f || b4.z != 0.f || b4.w != 0.f;
                    if (nz) {
                        const int64_t n = vis_ids[r];
                        float o_eff, ca, cb, cc;
                        if (recs) {      // the forward's 64-byte record of this Gaussian (raster_rec.hpp) holds the conic and the blended
                            //                  opacity: one contiguous read instead of gathers from conics / opacities / compensations
                            const float4 g0 = *reinterpret_cast<const float4 *>(recs + r * REC_FLOATS);
                            const float2 g1 = *reinterpret_cast<const float2 *>(recs + r * REC_FLOATS + 4);
                            ca = g0.z; cb = g0.w; cc = g1.x; o_eff = g1.y;
                        } else {
                            o_eff = compensations ? opacities[n] * compensations[n] : opacities[n];
                            ca = conics[n * 3]; cb = conics[n * 3 + 1]; cc = conics[n * 3 + 2];
                        }
                        const RowGrads rg = rows_to_gradients(raw_rows + r * raw_stride, ca, cb, cc, o_eff);
                        ri.v_xy = rg.v_xy;
                        ri.v_conic[0] = rg.v_conic[0]; ri.v_conic[1] = rg.v_conic[1]; ri.v_conic[2] = rg.v_conic[2];
                        by_abs = rg.v_abs;
                    }
                } else {
                    const float *vcon = v_conics + r * gs.conics;
                    ri.v_conic[0] = vcon[0]; ri.v_conic[1] = vcon[1]; ri.v_conic[2] = vcon[2];
                    ri.v_xy = make_float2(v_means2d[r * gs.means2d], v_means2d[r * gs.means2d + 1]);
                    nz = ri.v_xy.x != 0.f || ri.v_xy.y != 0.f || ri.v_conic[0] != 0.f || ri.v_conic[1] != 0.f || ri.v_conic[2] != 0.f;
                }
                nz = nz || ri.v_depth != 0.f || ri.v_comp != 0.f || ri.v_opac_eff != 0.f;
                if (x_quat_rows) {
                    const float4 xq = reinterpret_cast<const float4 *>(x_quat_rows)[r];
                    nz = nz || xq.x != 0.f || xq.y != 0.f || xq.z != 0.f || xq.w != 0.f;
                }
                if (x_mean_rows) nz = nz || x_mean_rows[r * 3] != 0.f || x_mean_rows[r * 3 + 1] != 0.f || x_mean_rows[r * 3 + 2] != 0.f;
                if (sp.on) {
                    // the dense by-products of this row -- 2-D gradient (retain_grad), absgrad, extra colour channels --, where not zero
                    const bool nxy = sp.ex.means2d && (ri.v_xy.x != 0.f || ri.v_xy.y != 0.f);
                    const bool nab = sp.ex.means2d_abs && (by_abs.x != 0.f || by_abs.y != 0.f);
                    bool ncol = false;
                    if (sp.ex.colors)
                        for (int k = 0; k < sp.ex.channels; ++k) ncol = ncol || sp.ex.col_src[r * sp.ex.col_stride + k] != 0.f;
                    if (nxy || nab || ncol) {
                        const int64_t n = vis_ids[r];
                        if (nxy) reinterpret_cast<float2 *>(sp.ex.means2d)[n] = ri.v_xy;
                        if (nab) reinterpret_cast<float2 *>(sp.ex.means2d_abs)[n] = by_abs;
                        if (ncol)
                            for (int k = 0; k < sp.ex.channels; ++k) sp.ex.colors[n * sp.ex.channels + k] = sp.ex.col_src[r * sp.ex.col_stride + k];
                    }
                }
                if (nz) {
                    s_in[threadIdx.x] = ri;
                } else if (!sp.on) {
                    float4 *out = reinterpret_cast<float4 *>(ws + r * VIS_ROW);
                    out[0] = out[1] = out[2] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            const unsigned long long m = __ballot(nz);
            const int wave_id = threadIdx.x >> 6;
            if (lane_id() == 0) s_wcnt[wave_id] = __popcll(m);
            __syncthreads();
            int wbase = 0;
#pragma unroll
            for (int w = 0; w < PROJ_BLOCK / 64; ++w)
                if (w < wave_id) wbase += s_wcnt[w];
            if (nz) s_list[wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = threadIdx.x;
            __syncthreads();
        }
        int count = 0;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w) count += s_wcnt[w];
        float vRt[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) vRt[k] = 0.f;
        if ((int)threadIdx.x < count) {
            const int src = s_list[threadIdx.x];
            const int64_t r = r0 + src;
            const RowIn ri = s_in[src];
            const int64_t n = vis_ids[r];
            float m[3], sc[3], am[3] = {0.f, 0.f, 0.f}, aq[4] = {0.f, 0.f, 0.f, 0.f}, as[3] = {0.f, 0.f, 0.f}, ao = 0.f;
            float4 q;
            PairIn in;
            in.has_comp = compensations != nullptr; in.has_vcomp = v_compensations != nullptr; in.has_opac = v_opac_eff != nullptr;
            m[0] = means[n * 3]; m[1] = means[n * 3 + 1]; m[2] = means[n * 3 + 2];
            q = reinterpret_cast<const float4 *>(quats)[n];
            sc[0] = scales[n * 3]; sc[1] = scales[n * 3 + 1]; sc[2] = scales[n * 3 + 2];
            if (recs) {
                const float4 g0 = *reinterpret_cast<const float4 *>(recs + r * REC_FLOATS);
                in.conic[0] = g0.z; in.conic[1] = g0.w; in.conic[2] = recs[r * REC_FLOATS + 4];
            } else {
                in.conic[0] = conics[n * 3]; in.conic[1] = conics[n * 3 + 1]; in.conic[2] = conics[n * 3 + 2];
            }
            in.comp = in.has_comp ? compensations[n] : 1.f;
            in.opac = in.has_opac ? opacities[n] : 0.f;
            in.v_mean2d = ri.v_xy;
            in.v_conic[0] = ri.v_conic[0]; in.v_conic[1] = ri.v_conic[1]; in.v_conic[2] = ri.v_conic[2];
            in.v_depth = ri.v_depth;
            in.v_comp = ri.v_comp;
            in.v_opac_eff = ri.v_opac_eff;
            project_vjp_pair(m, q, sc, cam, W, H, eps2d, in, am, aq, as, ao, vRt);
            if (x_quat_rows) {   // quaternion gradients that reached the Gaussian beside the projection (camera-space normals)
                const float4 xq = reinterpret_cast<const float4 *>(x_quat_rows)[r];
                aq[0] += xq.x; aq[1] += xq.y; aq[2] += xq.z; aq[3] += xq.w;
            }
            if (x_mean_rows) {   // ... and position gradients (gsplat's differentiable view directions of the SH colours)
                am[0] += x_mean_rows[r * 3]; am[1] += x_mean_rows[r * 3 + 1]; am[2] += x_mean_rows[r * 3 + 2];
            }
            if (sp.on) {
                sp.v_means[n * 3] = am[0]; sp.v_means[n * 3 + 1] = am[1]; sp.v_means[n * 3 + 2] = am[2];
                reinterpret_cast<float4 *>(sp.v_quats)[n] = make_float4(aq[0], aq[1], aq[2], aq[3]);
                sp.v_scales[n * 3] = as[0]; sp.v_scales[n * 3 + 1] = as[1]; sp.v_scales[n * 3 + 2] = as[2];
                if (sp.v_opacities) sp.v_opacities[n] = ao;
            } else {
                float4 *out = reinterpret_cast<float4 *>(ws + r * VIS_ROW);
                out[0] = make_float4(am[0], am[1], am[2], aq[0]);
                out[1] = make_float4(aq[1], aq[2], aq[3], as[0]);
                out[2] = make_float4(as[1], as[2], ao, 0.f);
            }
        }
        if (count == 0) continue;      // (uniform over the block)
        if (v_viewmats) {
            __syncthreads();
            block_reduce_viewmat(vRt, s_acc, red);
        }
    }
    __syncthreads();
    if (v_viewmats && threadIdx.x < 12) {
        const int k = threadIdx.x;
        const float v = s_acc[k];
        // (1180 blocks x 12 atomics on ONE 64-byte line serialise: 5 us of this kernel's 41 at the headline workload, plus the launch
        //  that zeroed the target.  With vm_partials every block leaves its 12 sums and the pass behind adds them in a fixed order)
        if (vm_partials) vm_partials[blockIdx.x * 12 + k] = v;
        else if (v != 0.f) atomicAdd(v_viewmats + (k < 9 ? (k / 3) * 4 + (k % 3) : (k - 9) * 4 + 3), v);
    }
}

// v_viewmats[4][4] = sum over the blocks' partial sums (v_R[i][j] -> [i][j], v_t[i] -> [i][3], last row 0), in a fixed order: 12
// components x 16 threads each, then an LDS tree.  Called by the first block of the pass behind project_bwd_vis_kernel.
__device__ __forceinline__ void viewmat_from_partials(const float *__restrict__ partials, int n_blocks, float *__restrict__ v_viewmats,
                                                      float *lds /* [192] */) {
    const int t = threadIdx.x;
    if (t < 192) {
        const int k = t % 12, j = t / 12;
        float v = 0.f;
        for (int b = j; b < n_blocks; b += 16) v += partials[b * 12 + k];
        lds[t] = v;
    }
    __syncthreads();
    if (t < 16) {
        float v = 0.f;
        const int i = t >> 2, jj = t & 3;
        if (i < 3) {
            const int k = jj < 3 ? i * 3 + jj : 9 + i;
#pragma unroll
            for (int j = 0; j < 16; ++j) v += lds[j * 12 + k];
        }
        v_viewmats[t] = v;
    }
    __syncthreads();
}
// The same sum as a kernel of its own (no streaming pass behind the per-visible pass: rows only, zeroed outputs): ONE workgroup of 1024
// threads -- 12 components x 85 slices, ~14 independent loads per thread, then the slices from LDS.  (The 16-slice form above inside a
// 256-thread launch measured 19 us for 1180 blocks: 74 dependent global loads per thread.)  A fixed order: deterministic.
constexpr int VM_SLICES = 85;
__global__ __launch_bounds__(1024) void viewmat_reduce_kernel(const float *__restrict__ partials, int n_blocks, float *__restrict__ v_viewmats) {
    __shared__ float lds[VM_SLICES * 12];
    const int t = threadIdx.x;
    if (t < VM_SLICES * 12) {
        const int k = t % 12, j = t / 12;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int b = j;
        for (; b + 3 * VM_SLICES < n_blocks; b += 4 * VM_SLICES) {
            v0 += partials[b * 12 + k]; v1 += partials[(b + VM_SLICES) * 12 + k];
            v2 += partials[(b + 2 * VM_SLICES) * 12 + k]; v3 += partials[(b + 3 * VM_SLICES) * 12 + k];
        }
        for (; b < n_blocks; b += VM_SLICES) v0 += partials[b * 12 + k];
        lds[t] = (v0 + v1) + (v2 + v3);
    }
    __syncthreads();
    if (t < 16) {
        float v = 0.f;
        const int i = t >> 2, jj = t & 3;
        if (i < 3) {
            const int k = jj < 3 ? i * 3 + jj : 9 + i;
            for (int j = 0; j < VM_SLICES; ++j) v += lds[j * 12 + k];
        }
        v_viewmats[t] = v;
    }
}

// ---- wire rows (view-parallel data parallelism, mtgs_amd.dist / csrc/dp.hip) ----------------------------------------
// The same per-visible-Gaussian VJP, reading the compositing backward's compact gradient rows
//   G[r] = [v_xy 2 | |v_xy| 2 | v_conic 3 | v_opacity_eff 1 | v_colour DC | v_depth 1 | ...]
// and writing, per visible Gaussian in index order, the 64-byte row the gradient exchange puts on the wire:
//   [v_mean 3 | v_quat 4 | v_scale 3 | v_opacity 1 | v_rgb 3 | 0 | Gaussian index (int bits)]
// v_rgb is the gradient with respect to the SH OUTPUT x: with color_mode 1 the colour was clamp(x + 0.5, 0, 1)
// (front.hip), whose VJP passes the gradient where 0 <= x + 0.5 <= 1 (torch.clamp's rule); color_mode 2: the colours were evaluated for
// the visible Gaussians by mtgs_vis_color_fwd and `colors_pre` points at ITS clamp bits (uint8 per visible row, bit c = channel c passes).  No dense tensor is
// written: the receivers' reduction (mtgs_dp_reduce) rebuilds every dense gradient, this rank's included.
constexpr int WIRE_ROW = 16;
__global__ __launch_bounds__(PROJ_BLOCK) void project_bwd_rows_kernel(
    int64_t n_vis, const int32_t *__restrict__ vis_ids, const float *__restrict__ means, const float *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ viewmats, const float *__restrict__ Ks, int W, int H,
    float eps2d, const float *__restrict__ conics, const float *__restrict__ compensations,
    const float *__restrict__ opacities, float *G, int64_t gs, int DC, int with_depth,
    const float *__restrict__ colors_pre, int color_mode, float *__restrict__ wire, float *__restrict__ v_viewmats, int raw_rows) {
    __shared__ float red[(PROJ_BLOCK / 64) * 12];
    __shared__ float s_acc[12];
    __shared__ int s_list[PROJ_BLOCK];
    __shared__ int s_wcnt[PROJ_BLOCK / 64];
    if (threadIdx.x < 12) s_acc[threadIdx.x] = 0.f;
    const Cam cam = load_cam(viewmats, Ks);
    for (int64_t r0 = (int64_t)blockIdx.x * PROJ_BLOCK; r0 < n_vis; r0 += (int64_t)gridDim.x * PROJ_BLOCK) {
        // As in project_bwd_vis_kernel: most frustum-visible rows carry no gradient (nothing was composited from them).  Their wire
        // row is zeros + the index -- no gathers, no VJP; the rows that have one are compacted (ballot + LDS list) and the ~2000
        // instructions run on dense waves (round 5: the data-parallel render leg ran the VJP for every visible row).
        __syncthreads();      // s_list reuse
        {
            const int64_t r = r0 + threadIdx.x;
            bool nz = false;
            if (r < n_vis) {
                const float4 *g4 = reinterpret_cast<const float4 *>(G + r * gs);      // (the WHOLE row: extra channels -- normals -- too)
                for (int64_t q4 = 0; q4 < gs / 4; ++q4) {
                    const float4 a4 = g4[q4];
                    nz = nz || a4.x != 0.f || a4.y != 0.f || a4.z != 0.f || a4.w != 0.f;
                }
                if (!nz) {
                    float4 *out = reinterpret_cast<float4 *>(wire + r * WIRE_ROW);
                    out[0] = out[1] = out[2] = make_float4(0.f, 0.f, 0.f, 0.f);
                    out[3] = make_float4(0.f, 0.f, 0.f, __int_as_float((int)vis_ids[r]));
                }
            }
            const unsigned long long m = __ballot(nz);
            const int wave_id = threadIdx.x >> 6;
            if (lane_id() == 0) s_wcnt[wave_id] = __popcll(m);
            __syncthreads();
            int wbase = 0;
#pragma unroll
            for (int w = 0; w < PROJ_BLOCK / 64; ++w)
                if (w < wave_id) wbase += s_wcnt[w];
            if (nz) s_list[wbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = threadIdx.x;
            __syncthreads();
        }
        int count = 0;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w) count += s_wcnt[w];
        if (count == 0) continue;      // (uniform over the block)
        float vRt[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) vRt[k] = 0.f;
        if ((int)threadIdx.x < count) {
            const int64_t r = r0 + s_list[threadIdx.x];
            const int64_t n = vis_ids[r];
            float m[3], sc[3], am[3] = {0.f, 0.f, 0.f}, aq[4] = {0.f, 0.f, 0.f, 0.f}, as[3] = {0.f, 0.f, 0.f}, ao = 0.f;
            m[0] = means[n * 3]; m[1] = means[n * 3 + 1]; m[2] = means[n * 3 + 2];
            const float4 q = reinterpret_cast<const float4 *>(quats)[n];
            sc[0] = scales[n * 3]; sc[1] = scales[n * 3 + 1]; sc[2] = scales[n * 3 + 2];
            PairIn in;
            in.conic[0] = conics[n * 3]; in.conic[1] = conics[n * 3 + 1]; in.conic[2] = conics[n * 3 + 2];
            const float4 *g4 = reinterpret_cast<const float4 *>(G + r * gs);
            float4 g0 = g4[0], g1 = g4[1];
            const float4 g2 = g4[2];
            const float v_depth_far = (with_depth && DC >= 4) ? G[r * gs + 8 + DC] : 0.f;
            if (raw_rows) {      // (everything the row holds is read before it is rewritten in place: the pointers alias)
                const RowGrads rg = rows_to_gradients(G + r * gs, in.conic[0], in.conic[1], in.conic[2],
                                                      compensations ? opacities[n] * compensations[n] : opacities[n]);
                g0.x = rg.v_xy.x; g0.y = rg.v_xy.y;
                g1.x = rg.v_conic[0]; g1.y = rg.v_conic[1]; g1.z = rg.v_conic[2];
            }
            in.v_mean2d = make_float2(g0.x, g0.y);
            in.v_conic[0] = g1.x; in.v_conic[1] = g1.y; in.v_conic[2] = g1.z;
            const float gc[4] = {g2.x, g2.y, g2.z, g2.w};
            in.v_depth = with_depth ? (DC < 4 ? gc[DC] : v_depth_far) : 0.f;
            in.has_comp = compensations != nullptr; in.has_vcomp = false; in.has_opac = true;
            in.comp = in.has_comp ? compensations[n] : 1.f;
            in.v_comp = 0.f;
            in.opac = opacities[n];
            in.v_opac_eff = g1.w;
            project_vjp_pair(m, q, sc, cam, W, H, eps2d, in, am, aq, as, ao, vRt);
            float vrgb[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < DC) {
                    float v = gc[k];
                    if (color_mode == 1) {
                        const float x = colors_pre[n * DC + k] + 0.5f;
                        v = (x >= 0.f && x <= 1.f) ? v : 0.f;
                    } else if (color_mode == 2) {      // the clamp bits of mtgs_vis_color_fwd, one byte per visible row
                        v = ((reinterpret_cast<const uint8_t *>(colors_pre)[r] >> k) & 1u) ? v : 0.f;
                    }
                    vrgb[k] = v;
                }
            }
            float4 *out = reinterpret_cast<float4 *>(wire + r * WIRE_ROW);
            out[0] = make_float4(am[0], am[1], am[2], aq[0]);
            out[1] = make_float4(aq[1], aq[2], aq[3], as[0]);
            out[2] = make_float4(as[1], as[2], ao, vrgb[0]);
            out[3] = make_float4(vrgb[1], vrgb[2], 0.f, __int_as_float((int)n));
        }
        if (v_viewmats) {
            __syncthreads();
            block_reduce_viewmat(vRt, s_acc, red);
        }
    }
    __syncthreads();
    if (v_viewmats && threadIdx.x < 12) {
        const int k = threadIdx.x;
        const float v = s_acc[k];
        if (v != 0.f) atomicAdd(v_viewmats + (k < 9 ? (k / 3) * 4 + (k % 3) : (k - 9) * 4 + 3), v);
    }
}

// (NON-TEMPORAL stores for the dense gradients nobody reads inside the step -- everything but the colours, which the caller's
//  autograd chain picks up next: step 0.941 -> 0.928 ms, same box)
#ifndef MTGS_NT_EXPAND
#define MTGS_NT_EXPAND true
#endif
constexpr int EXP_STAGE_COL = 8;  // colour channels staged through LDS at most (more: per-lane stores)
// `count` floats from LDS to consecutive global addresses, whole block, 16-byte stores when aligned
template <bool NT = false>
__device__ __forceinline__ void block_store(float *__restrict__ dst, const float *lds, int count) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int n4 = count >> 2;
        for (int k = threadIdx.x; k < n4; k += PROJ_BLOCK) {
            if (NT) __builtin_nontemporal_store(reinterpret_cast<const f32x4_t *>(lds)[k], reinterpret_cast<f32x4_t *>(dst) + k);
            else reinterpret_cast<float4 *>(dst)[k] = reinterpret_cast<const float4 *>(lds)[k];
        }
        for (int k = (n4 << 2) + threadIdx.x; k < count; k += PROJ_BLOCK) dst[k] = lds[k];
    } else {
        for (int k = threadIdx.x; k < count; k += PROJ_BLOCK) dst[k] = lds[k];
    }
}
// SC = colour channels the LDS staging is sized for (1 / 4 / 8: 14 / 17 / 21 KB per workgroup -- with 8 for every call the kernel
// ran 6 workgroups per CU instead of the 8 its wave slots allow, 2 us of 44 at the headline workload)
template <int SC>
__global__ __launch_bounds__(PROJ_BLOCK) void project_bwd_expand_kernel(
    int64_t N, int64_t n_rows, const int32_t *__restrict__ radii, const int32_t *__restrict__ row_index,
    const float *__restrict__ ws, const float *__restrict__ v_means2d, int64_t m2d_stride,
    float *__restrict__ v_means, float *__restrict__ v_quats, float *__restrict__ v_scales,
    float *__restrict__ v_opacities, const ProjExpand ex, const float *__restrict__ vm_partials, int vm_blocks, float *__restrict__ v_viewmats) {
    __shared__ __attribute__((aligned(16))) float s_vm[PROJ_BLOCK * 3], s_vq[PROJ_BLOCK * 4], s_vs[PROJ_BLOCK * 3], s_vo[PROJ_BLOCK];
    if (vm_partials && blockIdx.x == 0) viewmat_from_partials(vm_partials, vm_blocks, v_viewmats, s_vm);   // (the vis kernel's block sums)
    __shared__ __attribute__((aligned(16))) float s_m2d[PROJ_BLOCK * 2], s_abs[PROJ_BLOCK * 2], s_col[PROJ_BLOCK * SC];
    const int t = threadIdx.x;
    const int64_t chunk = (int64_t)blockIdx.x * PROJ_BLOCK;
    const int n_chunk = (int)min((int64_t)PROJ_BLOCK, N - chunk);
    const int64_t n = chunk + t;
    const bool stage_col = ex.colors && ex.channels <= SC;
    bool vis = n < N && radii[n] > 0;
    float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, w2 = w0;
    float2 xy = make_float2(0.f, 0.f), ab = xy;
    int64_t r = 0;
    if (vis) {
        r = row_index[n];
        vis = r < n_rows;   // (capacity-sized row buffers, graph mode: a Gaussian beyond them has no row)
    }
    if (vis) {
        const float4 *row = reinterpret_cast<const float4 *>(ws + r * VIS_ROW);
        w0 = row[0]; w1 = row[1]; w2 = row[2];
        if (ex.means2d) xy = make_float2(v_means2d[r * m2d_stride], v_means2d[r * m2d_stride + 1]);
        if (ex.means2d_abs) ab = make_float2(ex.abs_src[r * ex.abs_stride], ex.abs_src[r * ex.abs_stride + 1]);
    }
    s_vm[t * 3] = w0.x; s_vm[t * 3 + 1] = w0.y; s_vm[t * 3 + 2] = w0.z;
    reinterpret_cast<float4 *>(s_vq)[t] = make_float4(w0.w, w1.x, w1.y, w1.z);
    s_vs[t * 3] = w1.w; s_vs[t * 3 + 1] = w2.x; s_vs[t * 3 + 2] = w2.y;
    s_vo[t] = w2.z;
    reinterpret_cast<float2 *>(s_m2d)[t] = xy;
    reinterpret_cast<float2 *>(s_abs)[t] = ab;
    if (stage_col) {
        for (int k = 0; k < ex.channels; ++k) s_col[t * ex.channels + k] = vis ? ex.col_src[r * ex.col_stride + k] : 0.f;
    } else if (ex.colors && n < N) {
        for (int k = 0; k < ex.channels; ++k) ex.colors[n * ex.channels + k] = vis ? ex.col_src[r * ex.col_stride + k] : 0.f;
    }
    __syncthreads();
    block_store<MTGS_NT_EXPAND>(v_means + chunk * 3, s_vm, n_chunk * 3);
    block_store<MTGS_NT_EXPAND>(v_quats + chunk * 4, s_vq, n_chunk * 4);
    block_store<MTGS_NT_EXPAND>(v_scales + chunk * 3, s_vs, n_chunk * 3);
    if (v_opacities) block_store<MTGS_NT_EXPAND>(v_opacities + chunk, s_vo, n_chunk);
    if (ex.means2d) block_store<MTGS_NT_EXPAND>(ex.means2d + chunk * 2, s_m2d, n_chunk * 2);
    if (ex.means2d_abs) block_store<MTGS_NT_EXPAND>(ex.means2d_abs + chunk * 2, s_abs, n_chunk * 2);
    if (stage_col) block_store(ex.colors + chunk * ex.channels, s_col, n_chunk * ex.channels);
}

}  // namespace

namespace {
inline int64_t vis_blocks(int64_t n_vis) {
    const int64_t blocks = ceil_div64(n_vis > 0 ? n_vis : 1, PROJ_BLOCK);
    return blocks < 8192 ? blocks : 8192;
}
}  // namespace

extern "C" int mtgs_project_bwd_blocks(int64_t n_vis, int64_t *blocks) {
    MTGS_REQUIRE(n_vis >= 0 && blocks, MTGS_EINVAL, "mtgs_project_bwd_blocks: bad arguments");
    *blocks = vis_blocks(n_vis);
    return MTGS_OK;
}

static int project_bwd_impl(int C, int64_t N, const float *means, const float *quats,
                                const float *scales, const float *viewmats, const float *Ks,
                                int width, int height, float eps2d, const int32_t *radii,
                                const float *conics, const float *compensations, const float *opacities,
                                const float *v_means2d, const float *v_depths, const float *v_conics,
                                const float *v_compensations, const float *v_opac_eff, float *v_means,
                                float *v_quats, float *v_scales, float *v_viewmats, float *v_opacities,
                                const int64_t *grad_row_strides, const int32_t *grad_row_index,
                                const float *x_means2d_abs, const float *x_colors, int x_channels,
                                const int64_t *x_row_strides, float *d_means2d, float *d_means2d_abs,
                                float *d_colors, const int32_t *vis_ids, int64_t n_vis, float *vis_ws,
                                const int64_t *n_vis_dev, const float *x_quat_rows, const float *x_mean_rows, float *raw_rows,
                                const float *recs, float *vm_partials, void *stream, bool zeroed);

extern "C" int mtgs_project_bwd(int C, int64_t N, const float *means, const float *quats,
                                const float *scales, const float *viewmats, const float *Ks,
                                int width, int height, float eps2d, const int32_t *radii,
                                const float *conics, const float *compensations, const float *opacities,
                                const float *v_means2d, const float *v_depths, const float *v_conics,
                                const float *v_compensations, const float *v_opac_eff, float *v_means,
                                float *v_quats, float *v_scales, float *v_viewmats, float *v_opacities,
                                const int64_t *grad_row_strides, const int32_t *grad_row_index,
                                const float *x_means2d_abs, const float *x_colors, int x_channels,
                                const int64_t *x_row_strides, float *d_means2d, float *d_means2d_abs,
                                float *d_colors, const int32_t *vis_ids, int64_t n_vis, float *vis_ws,
                                const int64_t *n_vis_dev, const float *x_quat_rows, const float *x_mean_rows, float *raw_rows,
                                const float *recs, float *vm_partials, void *stream) {
    return project_bwd_impl(C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, radii, conics, compensations, opacities,
                            v_means2d, v_depths, v_conics, v_compensations, v_opac_eff, v_means, v_quats, v_scales, v_viewmats,
                            v_opacities, grad_row_strides, grad_row_index, x_means2d_abs, x_colors, x_channels, x_row_strides, d_means2d,
                            d_means2d_abs, d_colors, vis_ids, n_vis, vis_ws, n_vis_dev, x_quat_rows, x_mean_rows, raw_rows, recs,
                            vm_partials, stream, false);
}

extern "C" int mtgs_project_bwd_zeroed(int C, int64_t N, const float *means, const float *quats,
                                       const float *scales, const float *viewmats, const float *Ks,
                                       int width, int height, float eps2d, const int32_t *radii,
                                       const float *conics, const float *compensations, const float *opacities,
                                       const float *v_means2d, const float *v_depths, const float *v_conics,
                                       const float *v_compensations, const float *v_opac_eff, float *v_means,
                                       float *v_quats, float *v_scales, float *v_viewmats, float *v_opacities,
                                       const int64_t *grad_row_strides, const int32_t *grad_row_index,
                                       const float *x_means2d_abs, const float *x_colors, int x_channels,
                                       const int64_t *x_row_strides, float *d_means2d, float *d_means2d_abs,
                                       float *d_colors, const int32_t *vis_ids, int64_t n_vis, float *vis_ws,
                                       const int64_t *n_vis_dev, const float *x_quat_rows, const float *x_mean_rows, float *raw_rows,
                                       const float *recs, float *vm_partials, void *stream) {
    MTGS_REQUIRE(C == 1 && vis_ids && vis_ws && grad_row_index && v_means && v_quats && v_scales,
                 MTGS_EINVAL, "mtgs_project_bwd_zeroed: the compact path (C == 1, vis_ids, grad_row_index) with dense outputs");
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(v_quats) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_means2d) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(d_means2d_abs) & 7) == 0, MTGS_EINVAL, "mtgs_project_bwd_zeroed: v_quats 16-byte, d_means2d(_abs) 8-byte aligned");
    return project_bwd_impl(C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, radii, conics, compensations, opacities,
                            v_means2d, v_depths, v_conics, v_compensations, v_opac_eff, v_means, v_quats, v_scales, v_viewmats,
                            v_opacities, grad_row_strides, grad_row_index, x_means2d_abs, x_colors, x_channels, x_row_strides, d_means2d,
                            d_means2d_abs, d_colors, vis_ids, n_vis, vis_ws, n_vis_dev, x_quat_rows, x_mean_rows, raw_rows, recs,
                            vm_partials, stream, true);
}

static int project_bwd_impl(int C, int64_t N, const float *means, const float *quats,
                                const float *scales, const float *viewmats, const float *Ks,
                                int width, int height, float eps2d, const int32_t *radii,
                                const float *conics, const float *compensations, const float *opacities,
                                const float *v_means2d, const float *v_depths, const float *v_conics,
                                const float *v_compensations, const float *v_opac_eff, float *v_means,
                                float *v_quats, float *v_scales, float *v_viewmats, float *v_opacities,
                                const int64_t *grad_row_strides, const int32_t *grad_row_index,
                                const float *x_means2d_abs, const float *x_colors, int x_channels,
                                const int64_t *x_row_strides, float *d_means2d, float *d_means2d_abs,
                                float *d_colors, const int32_t *vis_ids, int64_t n_vis, float *vis_ws,
                                const int64_t *n_vis_dev, const float *x_quat_rows, const float *x_mean_rows, float *raw_rows,
                                const float *recs, float *vm_partials, void *stream, bool zeroed) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, MTGS_EINVAL,
                 "mtgs_project_bwd: bad sizes C=%d N=%lld W=%d H=%d", C, (long long)N, width, height);
    hipStream_t st = (hipStream_t)stream;
    // (vm_partials: the compact path writes v_viewmats in full from the blocks' partial sums -- no zero launch, no atomics)
    const bool partial_sums = vm_partials && v_viewmats && vis_ids && vis_ws && grad_row_index && C == 1 && N > 0 && n_vis > 0;
    if (v_viewmats && C > 0 && !partial_sums) {
        if (int rc = mtgs_zero_async(v_viewmats, sizeof(float) * 16 * (size_t)C, st)) return rc;
    }
    if (N == 0 || C == 0) return MTGS_OK;
    const bool rows_only = !v_means && !v_quats && !v_scales && !v_opacities && vis_ids && vis_ws && grad_row_index && C == 1 &&
                           !d_means2d && !d_means2d_abs && !d_colors;      // (the caller keeps the workspace rows: mtgs_node_bwd_rows)
    MTGS_REQUIRE(means && quats && scales && viewmats && Ks && radii && conics && v_means2d &&
                     v_depths && v_conics && (rows_only || (v_means && v_quats && v_scales)),
                 MTGS_EINVAL, "mtgs_project_bwd: null pointer");
    MTGS_REQUIRE(!v_compensations || compensations, MTGS_EINVAL,
                 "mtgs_project_bwd: v_compensations given without compensations");
    MTGS_REQUIRE(!v_opac_eff || (opacities && (v_opacities || rows_only)), MTGS_EINVAL,
                 "mtgs_project_bwd: v_opac_eff needs opacities and v_opacities");
    const int64_t dense[5] = {2, 1, 3, 1, 1};
    int64_t rs[5];
    for (int i = 0; i < 5; ++i) {
        rs[i] = grad_row_strides ? grad_row_strides[i] : dense[i];
        MTGS_REQUIRE(rs[i] >= dense[i], MTGS_EINVAL, "mtgs_project_bwd: grad_row_strides[%d]=%lld (row width %lld)", i,
                     (long long)rs[i], (long long)dense[i]);
    }
    const ProjGradStrides gs{rs[0], rs[1], rs[2], rs[3], rs[4]};
    MTGS_REQUIRE(x_channels >= 0 && x_channels <= MTGS_MAX_CHANNELS, MTGS_EINVAL, "mtgs_project_bwd: x_channels=%d", x_channels);
    MTGS_REQUIRE((!d_means2d_abs || x_means2d_abs) && (!d_colors || (x_colors && x_channels > 0)), MTGS_EINVAL,
                 "mtgs_project_bwd: a dense by-product was requested without its source rows");
    ProjExpand ex;
    ex.abs_src = x_means2d_abs; ex.col_src = x_colors;
    ex.abs_stride = x_row_strides ? x_row_strides[0] : 2;
    ex.col_stride = x_row_strides ? x_row_strides[1] : x_channels;
    ex.channels = x_channels;
    ex.means2d = d_means2d; ex.means2d_abs = d_means2d_abs; ex.colors = d_colors;
    MTGS_REQUIRE(ex.abs_stride >= 2 && ex.col_stride >= x_channels, MTGS_EINVAL, "mtgs_project_bwd: x_row_strides too small");
    MTGS_REQUIRE(!x_quat_rows || (vis_ids && vis_ws && grad_row_index && C == 1 && (reinterpret_cast<uintptr_t>(x_quat_rows) & 15) == 0),
                 MTGS_EINVAL, "mtgs_project_bwd: x_quat_rows needs the compact path (vis_ids, vis_ws, grad_row_index, C == 1), 16-byte aligned");
    MTGS_REQUIRE(!x_mean_rows || (vis_ids && vis_ws && grad_row_index && C == 1), MTGS_EINVAL,
                 "mtgs_project_bwd: x_mean_rows needs the compact path (vis_ids, vis_ws, grad_row_index, C == 1)");
    MTGS_REQUIRE(!raw_rows || (vis_ids && vis_ws && grad_row_index && C == 1 && opacities && rs[0] >= 8 &&
                               (reinterpret_cast<uintptr_t>(raw_rows) & 15) == 0 && (rs[0] & 3) == 0),
                 MTGS_EINVAL, "mtgs_project_bwd: raw_rows needs the compact path, opacities and 16-byte aligned rows of >= 8 floats");
    MTGS_REQUIRE(!recs || (vis_ids && vis_ws && grad_row_index && C == 1 && (reinterpret_cast<uintptr_t>(recs) & 15) == 0), MTGS_EINVAL,
                 "mtgs_project_bwd: recs needs the compact path (vis_ids, vis_ws, grad_row_index, C == 1), 16-byte aligned");
    if (vis_ids && vis_ws && grad_row_index && C == 1) {
        // compact path: grad_row_index[vis_ids[r]] == r (mtgs_bin_compact's vis_ids / vis_rank)
        MTGS_REQUIRE(n_vis >= 0 && n_vis <= N, MTGS_EINVAL, "mtgs_project_bwd: n_vis=%lld", (long long)n_vis);
        if (n_vis > 0) {
            ProjSparse sp;
            sp.on = zeroed ? 1 : 0;
            sp.v_means = v_means; sp.v_quats = v_quats; sp.v_scales = v_scales; sp.v_opacities = v_opacities; sp.ex = ex;
            project_bwd_vis_kernel<<<(unsigned)vis_blocks(n_vis), PROJ_BLOCK, 0, st>>>(
                n_vis, vis_ids, means, quats, scales, viewmats, Ks, width, height, eps2d, conics, compensations, opacities,
                v_means2d, v_depths, v_conics, v_compensations, v_opac_eff, gs, vis_ws, v_viewmats, n_vis_dev, x_quat_rows, x_mean_rows,
                raw_rows, rs[0], recs, partial_sums ? vm_partials : nullptr, sp);
        }
        const int vm_blocks = (int)vis_blocks(n_vis);
        if (zeroed) {      // no streaming pass: the outputs were zeroed by the caller, the rows with a gradient are in place
            if (partial_sums) viewmat_reduce_kernel<<<1, 1024, 0, st>>>(vm_partials, vm_blocks, v_viewmats);
        } else if (!rows_only) {
            const unsigned eg = (unsigned)ceil_div64(N, PROJ_BLOCK);
            const float *vp = partial_sums ? vm_partials : nullptr;
#define MTGS_EXPAND(SC) project_bwd_expand_kernel<SC><<<eg, PROJ_BLOCK, 0, st>>>(N, n_vis, radii, grad_row_index, vis_ws, v_means2d, gs.means2d, \
                                                                                 v_means, v_quats, v_scales, v_opacities, ex, vp, vm_blocks, v_viewmats)
            if (!ex.colors || ex.channels > EXP_STAGE_COL) MTGS_EXPAND(1);      // (nothing staged: per-lane stores or no colours)
            else if (ex.channels <= 4) MTGS_EXPAND(4);
            else MTGS_EXPAND(EXP_STAGE_COL);
#undef MTGS_EXPAND
        }
        else if (partial_sums) viewmat_reduce_kernel<<<1, 1024, 0, st>>>(vm_partials, vm_blocks, v_viewmats);
        MTGS_CHECK_LAUNCH("mtgs_project_bwd");
        return MTGS_OK;
    }
    const unsigned grid = (unsigned)(ceil_div64(N, PROJ_BLOCK) < 2048 ? ceil_div64(N, PROJ_BLOCK) : 2048);
    project_bwd_kernel<<<grid, PROJ_BLOCK, 0, st>>>(C, N, means, quats, scales, viewmats, Ks, width,
                                                    height, eps2d, radii, conics, compensations, opacities,
                                                    v_means2d, v_depths, v_conics, v_compensations,
                                                    v_opac_eff, v_means, v_quats, v_scales, v_viewmats,
                                                    v_opacities, gs, grad_row_index, ex);
    MTGS_CHECK_LAUNCH("mtgs_project_bwd");
    return MTGS_OK;
}

extern "C" int mtgs_project_bwd_rows(int64_t N, const float *means, const float *quats, const float *scales,
                                     const float *viewmats, const float *Ks, int width, int height, float eps2d,
                                     const float *conics, const float *compensations, const float *opacities,
                                     float *grad_rows, int64_t row_stride, int D, int with_depth,
                                     const float *colors_pre, int color_mode, const int32_t *vis_ids, int64_t n_vis,
                                     float *wire_rows, float *v_viewmats, int raw_rows, void *stream) {
    MTGS_REQUIRE(N >= 0 && n_vis >= 0 && n_vis <= N && width > 0 && height > 0, MTGS_EINVAL, "mtgs_project_bwd_rows: bad sizes");
    // (colour channels beyond the third -- camera-space normals ... -- are not this call's: their gradient is folded into
    //  the rows by their own VJP, e.g. mtgs_normals_bwd_rows)
    MTGS_REQUIRE(D >= 0 && D <= 7 && row_stride >= 8 + D + (with_depth ? 1 : 0) && (row_stride % 4) == 0, MTGS_EINVAL,
                 "mtgs_project_bwd_rows: D=%d (0..7 colour channels) row_stride=%lld", D, (long long)row_stride);
    MTGS_REQUIRE(color_mode == 0 || ((color_mode == 1 || color_mode == 2) && D >= 3 && colors_pre), MTGS_EINVAL,
                 "mtgs_project_bwd_rows: color_mode");
    hipStream_t st = (hipStream_t)stream;
    if (v_viewmats) {
        if (int rc = mtgs_zero_async(v_viewmats, sizeof(float) * 16, st)) return rc;
    }
    if (n_vis == 0) return MTGS_OK;
    MTGS_REQUIRE(means && quats && scales && viewmats && Ks && conics && opacities && grad_rows && vis_ids && wire_rows,
                 MTGS_EINVAL, "mtgs_project_bwd_rows: null pointer");
    MTGS_REQUIRE(((reinterpret_cast<uintptr_t>(grad_rows) | reinterpret_cast<uintptr_t>(wire_rows)) & 15) == 0, MTGS_EINVAL,
                 "mtgs_project_bwd_rows: rows must be 16-byte aligned");
    const int64_t blocks = ceil_div64(n_vis, PROJ_BLOCK);
    project_bwd_rows_kernel<<<(unsigned)(blocks < 8192 ? blocks : 8192), PROJ_BLOCK, 0, st>>>(
        n_vis, vis_ids, means, quats, scales, viewmats, Ks, width, height, eps2d, conics, compensations, opacities, grad_rows,
        row_stride, D, with_depth ? 1 : 0, colors_pre, color_mode, wire_rows, v_viewmats, raw_rows);
    MTGS_CHECK_LAUNCH("mtgs_project_bwd_rows");
    return MTGS_OK;
}
