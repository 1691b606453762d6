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

}  // namespace
