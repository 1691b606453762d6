// tile_rect.hpp -- the clamped tile rectangle of a projected Gaussian (gsplat 1.4.0 isect_tiles: 3-sigma square
// around the mean, half-open, clamped to the tile grid).  Shared by the count / emit kernels (isect.hip, bin.hip) and by
// the projection forward (project.hip), which can write the tile count itself.  Only additions, subtractions,
// divisions and floor / ceil: the result does not depend on the translation unit's -ffp-contract setting.
#pragma once
#include "common.hpp"

namespace {

struct Rect { int x0, y0, x1, y1; };

__device__ __forceinline__ Rect tile_rect(float mx, float my, int32_t radius, float ts, int tw, int th) {
    const float tr = (float)radius / ts, tx = mx / ts, ty = my / ts;
    Rect r;
    r.x0 = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)tw);
    r.y0 = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)th);
    r.x1 = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)tw);
    r.y1 = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)th);
    return r;
}

}  // namespace
