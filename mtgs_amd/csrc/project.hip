// project.hip -- per-Gaussian 3D->2D projection (EWA splatting) forward and backward.
//
// Replaces gsplat 1.4.0 fully_fused_projection_fwd / fully_fused_projection_bwd (pinhole,
// packed=False), the first stage of gsplat.rendering.rasterization as called at
// /root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662.
//
// Roofline: HBM (pure streaming).  Algorithmic bytes per (camera, Gaussian):
//   fwd: 40 in (mean 12, quat 16, scale 12) + 4 (radius) out, + 28 (+4 compensation) for visible ones
//   bwd: 40 + 4 (radius) in; visible: + 16 (conic, comp) + 28 (+4) cotangents; 40 out
//
// This translation unit is compiled with -ffp-contract=off: the forward rounds after every
// operation in a fixed order, so radii / means2d / depths (the inputs of the integer tile-binning
// stage, whose results must be bit-exact) have exactly one IEEE-754 value per input.
#include "common.hpp"

namespace {

constexpr int PROJ_BLOCK = 256;
constexpr float kFovMargin = 0.3f;     // persp_proj: frustum clamp margin (x tan_fov)
constexpr float kRadiusFloor = 0.01f;  // sqrt(max(0.01, b^2 - det))
constexpr float kRadiusSigma = 3.0f;   // 3-sigma extent
constexpr float kCompEps = 1e-6f;      // add_blur_vjp epsilon

struct Cam {
    float R[9];
    float t[3];
    float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float *__restrict__ vm, const float *__restrict__ K) {
    Cam c;
    c.R[0] = vm[0]; c.R[1] = vm[1]; c.R[2] = vm[2];
    c.R[3] = vm[4]; c.R[4] = vm[5]; c.R[5] = vm[6];
    c.R[6] = vm[8]; c.R[7] = vm[9]; c.R[8] = vm[10];
    c.t[0] = vm[3]; c.t[1] = vm[7]; c.t[2] = vm[11];
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
    return c;
}

// C = A * B, C = A * B^T, C = A^T * B with the summation order (a0 b0 + a1 b1) + a2 b2
__device__ __forceinline__ void mm3(const float *A, const float *B, float *C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void mm3_bt(const float *A, const float *B, float *C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1]) + A[i * 3 + 2] * B[j * 3 + 2];
}
__device__ __forceinline__ void mm3_at(const float *A, const float *B, float *C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i] * B[j] + A[3 + i] * B[3 + j]) + A[6 + i] * B[6 + j];
}

struct ProjState {
    float mean_c[3];
    float Rq[9], Mq[9], covar[9], covar_c[9];
    float J[6];
    float rz, rz2, tx, ty;
    bool x_clamped, y_clamped;
    float cov2d[4];
    float qn[4], inv_norm;
};

__device__ __forceinline__ void proj_common(const float *m, const float4 q, const float *sc,
                                            const Cam &cam, int W, int H, ProjState &s) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
        s.mean_c[i] = ((cam.R[i * 3] * m[0] + cam.R[i * 3 + 1] * m[1]) + cam.R[i * 3 + 2] * m[2]) + cam.t[i];
    {
        float w = q.x, x = q.y, y = q.z, z = q.w;  // wxyz
        const float inv = 1.0f / sqrtf(((x * x + y * y) + z * z) + w * w);
        w *= inv; x *= inv; y *= inv; z *= inv;
        const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
        const float wx = w * x, wy = w * y, wz = w * z;
        s.Rq[0] = 1.f - 2.f * (y2 + z2); s.Rq[1] = 2.f * (xy - wz); s.Rq[2] = 2.f * (xz + wy);
        s.Rq[3] = 2.f * (xy + wz); s.Rq[4] = 1.f - 2.f * (x2 + z2); s.Rq[5] = 2.f * (yz - wx);
        s.Rq[6] = 2.f * (xz - wy); s.Rq[7] = 2.f * (yz + wx); s.Rq[8] = 1.f - 2.f * (x2 + y2);
        s.qn[0] = w; s.qn[1] = x; s.qn[2] = y; s.qn[3] = z;
        s.inv_norm = inv;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s.Mq[i * 3 + j] = s.Rq[i * 3 + j] * sc[j];
    mm3_bt(s.Mq, s.Mq, s.covar);
    float tmp[9];
    mm3(cam.R, s.covar, tmp);
    mm3_bt(tmp, cam.R, s.covar_c);
    const float x = s.mean_c[0], y = s.mean_c[1], z = s.mean_c[2];
    const float tan_fovx = 0.5f * (float)W / cam.fx, tan_fovy = 0.5f * (float)H / cam.fy;
    const float lim_x_pos = ((float)W - cam.cx) / cam.fx + kFovMargin * tan_fovx;
    const float lim_x_neg = cam.cx / cam.fx + kFovMargin * tan_fovx;
    const float lim_y_pos = ((float)H - cam.cy) / cam.fy + kFovMargin * tan_fovy;
    const float lim_y_neg = cam.cy / cam.fy + kFovMargin * tan_fovy;
    const float rz = 1.0f / z, rz2 = rz * rz;
    const float xz = x * rz, yz = y * rz;
    s.x_clamped = !(xz <= lim_x_pos && xz >= -lim_x_neg);
    s.y_clamped = !(yz <= lim_y_pos && yz >= -lim_y_neg);
    const float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, xz));
    const float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, yz));
    s.rz = rz; s.rz2 = rz2; s.tx = tx; s.ty = ty;
    s.J[0] = cam.fx * rz; s.J[1] = 0.f; s.J[2] = -cam.fx * tx * rz2;
    s.J[3] = 0.f; s.J[4] = cam.fy * rz; s.J[5] = -cam.fy * ty * rz2;
    float B[6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            B[i * 3 + j] = (s.J[i * 3] * s.covar_c[j] + s.J[i * 3 + 1] * s.covar_c[3 + j]) + s.J[i * 3 + 2] * s.covar_c[6 + j];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            s.cov2d[i * 2 + j] = (B[i * 3] * s.J[j * 3] + B[i * 3 + 1] * s.J[j * 3 + 1]) + B[i * 3 + 2] * s.J[j * 3 + 2];
}

__global__ __launch_bounds__(PROJ_BLOCK) void project_fwd_kernel(
    int C, int64_t N, const float *__restrict__ means, const float *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ viewmats,
    const float *__restrict__ Ks, int W, int H, float eps2d, float near_plane, float far_plane,
    float radius_clip, const float *__restrict__ opacities, int32_t *__restrict__ radii,
    float *__restrict__ means2d, float *__restrict__ depths, float *__restrict__ conics,
    float *__restrict__ compensations, float *__restrict__ opac_eff) {
    const int64_t idx = (int64_t)blockIdx.x * PROJ_BLOCK + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int c = (int)(idx / N);
    const int64_t n = idx - (int64_t)c * N;
    const Cam cam = load_cam(viewmats + c * 16, Ks + c * 9);
    const float m[3] = {means[n * 3], means[n * 3 + 1], means[n * 3 + 2]};
    int32_t r_out = 0;
    float mx = 0.f, my = 0.f, depth = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, comp = 0.f;
    const float zc = ((cam.R[6] * m[0] + cam.R[7] * m[1]) + cam.R[8] * m[2]) + cam.t[2];
    if (!(zc < near_plane || zc > far_plane)) {
        const float4 q = reinterpret_cast<const float4 *>(quats)[n];
        const float sc[3] = {scales[n * 3], scales[n * 3 + 1], scales[n * 3 + 2]};
        ProjState s;
        proj_common(m, q, sc, cam, W, H, s);
        const float pmx = cam.fx * s.mean_c[0] * s.rz + cam.cx;
        const float pmy = cam.fy * s.mean_c[1] * s.rz + cam.cy;
        float c00 = s.cov2d[0];
        const float c01 = s.cov2d[1];
        float c11 = s.cov2d[3];
        const float det_orig = c00 * c11 - c01 * c01;
        c00 += eps2d; c11 += eps2d;
        const float det = c00 * c11 - c01 * c01;
        const float cmp = sqrtf(fmaxf(0.f, det_orig / det));
        if (det > 0.f) {
            const float idet = 1.0f / det;
            const float b = 0.5f * (c00 + c11);
            const float v1 = b + sqrtf(fmaxf(kRadiusFloor, b * b - det));
            const float radius = ceilf(kRadiusSigma * sqrtf(v1));
            const bool out = radius <= radius_clip || pmx + radius <= 0.f || pmx - radius >= (float)W ||
                             pmy + radius <= 0.f || pmy - radius >= (float)H;
            if (!out) {
                r_out = (int32_t)radius;
                mx = pmx; my = pmy; depth = s.mean_c[2];
                ca = c11 * idet; cb = -c01 * idet; cc = c00 * idet;
                comp = cmp;
            }
        }
    }
    radii[idx] = r_out;
    reinterpret_cast<float2 *>(means2d)[idx] = make_float2(mx, my);
    depths[idx] = depth;
    conics[idx * 3] = ca; conics[idx * 3 + 1] = cb; conics[idx * 3 + 2] = cc;
    if (compensations) compensations[idx] = comp;
    // gsplat rendering.py: opacities.repeat(C, 1) [* compensations]
    if (opac_eff) opac_eff[idx] = r_out > 0 ? (compensations ? opacities[n] * comp : opacities[n]) : 0.f;
}

// Sum 12 values (v_R, v_t) over the block and add them to v_viewmats with one atomic per value
// per block.
__device__ __forceinline__ void block_reduce_viewmat(float (&vals)[12], float *__restrict__ out,
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
        // v_R[i][j] -> viewmat[i][j], v_t[i] -> viewmat[i][3]
        const int off = k < 9 ? (k / 3) * 4 + (k % 3) : (k - 9) * 4 + 3;
        if (v != 0.f) atomicAdd(out + off, v);
    }
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
    float *__restrict__ v_viewmats, float *__restrict__ v_opacities) {
    __shared__ float red[(PROJ_BLOCK / 64) * 12];
    const int64_t n = (int64_t)blockIdx.x * PROJ_BLOCK + threadIdx.x;
    const bool live = n < N;
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
            const float va = v_conics[idx * 3], vb = 0.5f * v_conics[idx * 3 + 1], vc = v_conics[idx * 3 + 2];
            const float t00 = a * va + b * vb, t01 = a * vb + b * vc, t10 = b * va + cc * vb, t11 = b * vb + cc * vc;
            float vcov[4];
            vcov[0] = -(t00 * a + t01 * b); vcov[1] = -(t00 * b + t01 * cc);
            vcov[2] = -(t10 * a + t11 * b); vcov[3] = -(t10 * b + t11 * cc);
            // opac_eff = opacity * compensation: the product rule feeds the compensation VJP
            if (v_opac_eff) ao += v_opac_eff[idx] * (compensations ? compensations[idx] : 1.f);
            if (compensations && (v_compensations || v_opac_eff)) {
                const float comp = compensations[idx];
                const float vcomp = (v_compensations ? v_compensations[idx] : 0.f) +
                                    (v_opac_eff ? v_opac_eff[idx] * opac : 0.f);
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
            const float2 vm2 = reinterpret_cast<const float2 *>(v_means2d)[idx];
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
            v_mean_c[2] += v_depths[idx];
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
            if (c > 0) __syncthreads();
            block_reduce_viewmat(vRt, v_viewmats + c * 16, red);
        }
    }
    if (live) {
        v_means[n * 3] = am[0]; v_means[n * 3 + 1] = am[1]; v_means[n * 3 + 2] = am[2];
        reinterpret_cast<float4 *>(v_quats)[n] = make_float4(aq[0], aq[1], aq[2], aq[3]);
        v_scales[n * 3] = as[0]; v_scales[n * 3 + 1] = as[1]; v_scales[n * 3 + 2] = as[2];
        if (v_opacities) v_opacities[n] = ao;
    }
}

}  // namespace

extern "C" int mtgs_project_fwd(int C, int64_t N, const float *means, const float *quats,
                                const float *scales, const float *viewmats, const float *Ks,
                                int width, int height, float eps2d, float near_plane, float far_plane,
                                float radius_clip, const float *opacities, int32_t *radii,
                                float *means2d, float *depths, float *conics, float *compensations,
                                float *opac_eff, void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, MTGS_EINVAL,
                 "mtgs_project_fwd: bad sizes C=%d N=%lld W=%d H=%d", C, (long long)N, width, height);
    if ((int64_t)C * N == 0) return MTGS_OK;
    MTGS_REQUIRE(means && quats && scales && viewmats && Ks && radii && means2d && depths && conics,
                 MTGS_EINVAL, "mtgs_project_fwd: null pointer");
    MTGS_REQUIRE((int64_t)C * N < ((int64_t)1 << 31), MTGS_EINVAL,
                 "mtgs_project_fwd: C*N must fit int32 (flatten_ids are int32)");
    MTGS_REQUIRE(!opac_eff || opacities, MTGS_EINVAL, "mtgs_project_fwd: opac_eff requested without opacities");
    const unsigned grid = (unsigned)ceil_div64((int64_t)C * N, PROJ_BLOCK);
    project_fwd_kernel<<<grid, PROJ_BLOCK, 0, (hipStream_t)stream>>>(
        C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
        radius_clip, opacities, radii, means2d, depths, conics, compensations, opac_eff);
    MTGS_CHECK_LAUNCH("mtgs_project_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_project_bwd(int C, int64_t N, const float *means, const float *quats,
                                const float *scales, const float *viewmats, const float *Ks,
                                int width, int height, float eps2d, const int32_t *radii,
                                const float *conics, const float *compensations, const float *opacities,
                                const float *v_means2d, const float *v_depths, const float *v_conics,
                                const float *v_compensations, const float *v_opac_eff, float *v_means,
                                float *v_quats, float *v_scales, float *v_viewmats, float *v_opacities,
                                void *stream) {
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
    const unsigned grid = (unsigned)ceil_div64(N, PROJ_BLOCK);
    project_bwd_kernel<<<grid, PROJ_BLOCK, 0, st>>>(C, N, means, quats, scales, viewmats, Ks, width,
                                                    height, eps2d, radii, conics, compensations, opacities,
                                                    v_means2d, v_depths, v_conics, v_compensations,
                                                    v_opac_eff, v_means, v_quats, v_scales, v_viewmats,
                                                    v_opacities);
    MTGS_CHECK_LAUNCH("mtgs_project_bwd");
    return MTGS_OK;
}
