// project_common.hpp -- camera / covariance / perspective-projection math shared by the projection
// forward (project.hip, compiled with -ffp-contract=off for bit-exact binning inputs) and backward
// (project_bwd.hip, compiled with the default fast contraction: its results need not be bit-exact).
// Follows gsplat 1.4.0 utils.cuh {pos_world_to_cam, quat_scale_to_covar_preci, covar_world_to_cam,
// persp_proj}; see oracle/gsplat_oracle.c for the restatement these are checked against.
#pragma once
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


}  // namespace
