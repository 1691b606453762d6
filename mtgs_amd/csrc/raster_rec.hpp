// raster_rec.hpp -- the packed per-VISIBLE-Gaussian record of the fused rasterization path.
//
// front.hip writes ONE 64-byte record per visible (camera, Gaussian) pair, indexed by its rank in index order
// (rank = number of visible pairs with a smaller flat index); the binning kernels read the footprint from it and the
// compositing kernels stage a candidate with ONE 64-byte gather instead of six 4-12 byte gathers from six dense
// arrays (fabric traffic of the compositing backward was 2.4x its algorithmic bytes that way).
//   float  0..1  mean2d x, y            (pixels)
//   float  2..4  conic a, b, c
//   float  5     opacity (x compensation in antialiased mode)
//   float  6     s2max = 2 ln(255 opacity): alpha >= 1/255  <=>  a dx^2 + 2 b dx dy + c dy^2 <= s2max
//                (negative or NaN: the Gaussian can never reach alpha >= 1/255 and is dropped when staged)
//   int32  7     radius in pixels (the 3-sigma square of gsplat's isect_tiles)
//   float  8..15 blended channels: colours, then the depth when the render mode blends one, then zeros
#pragma once
#include "common.hpp"

namespace {

constexpr int REC_FLOATS = 16;
constexpr int REC_MAX_CHANNELS = 8;

__device__ __forceinline__ float rec_s2max(float opacity) {
    // alpha = min(0.999, op e^{-s2/2}) >= 1/255  <=>  s2 <= 2 ln(255 op)
    return 2.0f * 0.6931471805599453f * __log2f(opacity * (1.0f / MTGS_ALPHA_MIN));
}

// Does the ellipse {q(d) = a dx^2 + 2 b dx dy + c dy^2 <= s2max} around the mean reach the rectangle [X0, X1] x [Y0, Y1] (pixel
// CENTRES of a tile, relative to the mean)?  q is convex, so its minimum over the rectangle is 0 if the mean lies inside,
// otherwise it is attained on one of the four edges (a clamped 1-D parabola each).  With a margin: a pair is dropped only if
// no pixel of the tile could pass the per-pixel test s2 <= s2max of the compositing kernels, so dropping it -- when a tile's
// candidates are staged (blend.hip) or already when the tile lists are built (bin3.hip, tight lists) -- never changes a pixel.
__device__ __forceinline__ bool rec_reaches_rect(float ca, float cb, float cc, float s2max, float X0, float X1, float Y0, float Y1) {
    const float det = ca * cc - cb * cb;
    if (!(det > 0.f && ca > 0.f && cc > 0.f)) return true;
    const bool inside = X0 <= 0.f && X1 >= 0.f && Y0 <= 0.f && Y1 >= 0.f;
    const float rcc = __builtin_amdgcn_rcpf(cc), rca = __builtin_amdgcn_rcpf(ca);
    auto edge_x = [&](float xe) {  // vertical edge x = xe
        const float dy = fminf(fmaxf(-cb * xe * rcc, Y0), Y1);
        return ca * xe * xe + 2.f * cb * xe * dy + cc * dy * dy;
    };
    auto edge_y = [&](float ye) {  // horizontal edge y = ye
        const float dx = fminf(fmaxf(-cb * ye * rca, X0), X1);
        return ca * dx * dx + 2.f * cb * dx * ye + cc * ye * ye;
    };
    const float qmin = inside ? 0.f : fminf(fminf(edge_x(X0), edge_x(X1)), fminf(edge_y(Y0), edge_y(Y1)));
    // margin: never drop a candidate the per-pixel test (s2 <= s2max) could still accept
    return qmin <= s2max * 1.001f + 1e-2f;
}

}  // namespace
