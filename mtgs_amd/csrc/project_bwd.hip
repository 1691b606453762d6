// project_bwd.hip -- VJP of the per-Gaussian 3D->2D projection.
//
// Replaces gsplat 1.4.0 fully_fused_projection_bwd (pinhole, packed=False), reached through
// gsplat.rendering.rasterization (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662).
// Compiled WITH fp contraction (unlike the forward): gradients are compared at 2e-3, not bit-exact.
//
// Roofline: HBM for the dense part (radii read, 44 B of gradients written per Gaussian, zeros for the
// invisible ones) + ~1500 flops per VISIBLE Gaussian.
#include "project_common.hpp"

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
constexpr int PROJ_MAX_CAMS = 64;  // cameras whose v_viewmat is accumulated in LDS (MTGS: 1)

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
    float *__restrict__ v_viewmats, float *__restrict__ v_opacities, const ProjGradStrides gs) {
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
        for (int c = 0; c < C; ++c) vis = vis || radii[(int64_t)c * N + n_own] > 0;
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
            ProjState s;
            proj_common(m, q, sc, cam, W, H, s);
            const float a = conics[idx * 3], b = conics[idx * 3 + 1], cc = conics[idx * 3 + 2];
            const float *vcon = v_conics + idx * gs.conics;
            const float va = vcon[0], vb = 0.5f * vcon[1], vc = vcon[2];
            const float t00 = a * va + b * vb, t01 = a * vb + b * vc, t10 = b * va + cc * vb, t11 = b * vb + cc * vc;
            float vcov[4];
            vcov[0] = -(t00 * a + t01 * b); vcov[1] = -(t00 * b + t01 * cc);
            vcov[2] = -(t10 * a + t11 * b); vcov[3] = -(t10 * b + t11 * cc);
            // opac_eff = opacity * compensation: the product rule feeds the compensation VJP
            if (v_opac_eff) ao += v_opac_eff[idx * gs.opac_eff] * (compensations ? compensations[idx] : 1.f);
            if (compensations && (v_compensations || v_opac_eff)) {
                const float comp = compensations[idx];
                const float vcomp = (v_compensations ? v_compensations[idx * gs.compensations] : 0.f) +
                                    (v_opac_eff ? v_opac_eff[idx * gs.opac_eff] * opac : 0.f);
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
            const float2 vm2 = make_float2(v_means2d[idx * gs.means2d], v_means2d[idx * gs.means2d + 1]);
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
            v_mean_c[2] += v_depths[idx * gs.depths];
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

}  // namespace

extern "C" int mtgs_project_bwd(int C, int64_t N, const float *means, const float *quats,
                                const float *scales, const float *viewmats, const float *Ks,
                                int width, int height, float eps2d, const int32_t *radii,
                                const float *conics, const float *compensations, const float *opacities,
                                const float *v_means2d, const float *v_depths, const float *v_conics,
                                const float *v_compensations, const float *v_opac_eff, float *v_means,
                                float *v_quats, float *v_scales, float *v_viewmats, float *v_opacities,
                                const int64_t *grad_row_strides, void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, MTGS_EINVAL,
                 "mtgs_project_bwd: bad sizes C=%d N=%lld W=%d H=%d", C, (long long)N, width, height);
    hipStream_t st = (hipStream_t)stream;
    if (v_viewmats && C > 0) {
        hipError_t e = hipMemsetAsync(v_viewmats, 0, sizeof(float) * 16 * (size_t)C, st);
        MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_project_bwd: memset failed: %s", hipGetErrorString(e));
    }
    if (N == 0 || C == 0) return MTGS_OK;
    MTGS_REQUIRE(means && quats && scales && viewmats && Ks && radii && conics && v_means2d &&
                     v_depths && v_conics && v_means && v_quats && v_scales,
                 MTGS_EINVAL, "mtgs_project_bwd: null pointer");
    MTGS_REQUIRE(!v_compensations || compensations, MTGS_EINVAL,
                 "mtgs_project_bwd: v_compensations given without compensations");
    MTGS_REQUIRE(!v_opac_eff || (opacities && v_opacities), MTGS_EINVAL,
                 "mtgs_project_bwd: v_opac_eff needs opacities and v_opacities");
    const int64_t dense[5] = {2, 1, 3, 1, 1};
    int64_t rs[5];
    for (int i = 0; i < 5; ++i) {
        rs[i] = grad_row_strides ? grad_row_strides[i] : dense[i];
        MTGS_REQUIRE(rs[i] >= dense[i], MTGS_EINVAL, "mtgs_project_bwd: grad_row_strides[%d]=%lld (row width %lld)", i,
                     (long long)rs[i], (long long)dense[i]);
    }
    const ProjGradStrides gs{rs[0], rs[1], rs[2], rs[3], rs[4]};
    const unsigned grid = (unsigned)(ceil_div64(N, PROJ_BLOCK) < 2048 ? ceil_div64(N, PROJ_BLOCK) : 2048);
    project_bwd_kernel<<<grid, PROJ_BLOCK, 0, st>>>(C, N, means, quats, scales, viewmats, Ks, width,
                                                    height, eps2d, radii, conics, compensations, opacities,
                                                    v_means2d, v_depths, v_conics, v_compensations,
                                                    v_opac_eff, v_means, v_quats, v_scales, v_viewmats,
                                                    v_opacities, gs);
    MTGS_CHECK_LAUNCH("mtgs_project_bwd");
    return MTGS_OK;
}
