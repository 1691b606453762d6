// project_fwd_body.hpp -- the projection of ONE (camera, Gaussian) pair, shared by project.hip (the gsplat-shaped
// operator) and front.hip (projection fused with compaction and the packed per-Gaussian records).  Both translation
// units are compiled with -ffp-contract=off: every operation rounds, in this fixed order, so radii / means2d / depths
// (the inputs of the integer tile-binning stage) have exactly one IEEE-754 value per input -- the one
// oracle/gsplat_oracle.c computes.  Follows gsplat 1.4.0 fully_fused_projection_fwd (pinhole).
#pragma once
#include "project_common.hpp"

namespace {

struct F3 { float x, y, z; };  // 12-byte rows move as one dwordx3 access

struct ProjOut {
    int32_t radius;   // 0 = culled (every float below is then 0)
    float mx, my, depth, ca, cb, cc, comp;
};

// near / far test of the camera-space depth (the first thing the projection decides; quaternion and scale are only
// needed when it passes)
__device__ __forceinline__ bool project_depth_ok(const F3 m3, const Cam &cam, float near_plane, float far_plane) {
    const float zc = ((cam.R[6] * m3.x + cam.R[7] * m3.y) + cam.R[8] * m3.z) + cam.t[2];
    return !(zc < near_plane || zc > far_plane);
}

// Projection of a pair whose depth test passed, from values already in registers.
__device__ __forceinline__ ProjOut project_values(const F3 m3, const float4 q, const F3 s3, const Cam &cam, int W, int H,
                                                  float eps2d, float radius_clip) {
    const float m[3] = {m3.x, m3.y, m3.z};
    const float sc[3] = {s3.x, s3.y, s3.z};
    ProjOut o{0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
            o.radius = (int32_t)radius;
            o.mx = pmx; o.my = pmy; o.depth = s.mean_c[2];
            o.ca = c11 * idet; o.cb = -c01 * idet; o.cc = c00 * idet;
            o.comp = cmp;
        }
    }
    return o;
}

__device__ __forceinline__ ProjOut project_pair(const float *__restrict__ means, const float *__restrict__ quats,
                                                const float *__restrict__ scales, const Cam &cam, int64_t n, int W, int H,
                                                float eps2d, float near_plane, float far_plane, float radius_clip) {
    const F3 m3 = *reinterpret_cast<const F3 *>(means + n * 3);
    if (!project_depth_ok(m3, cam, near_plane, far_plane)) return ProjOut{0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float4 q = reinterpret_cast<const float4 *>(quats)[n];
    const F3 s3 = *reinterpret_cast<const F3 *>(scales + n * 3);
    return project_values(m3, q, s3, cam, W, H, eps2d, radius_clip);
}

}  // namespace
