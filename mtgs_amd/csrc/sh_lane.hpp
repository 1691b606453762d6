// sh_lane.hpp -- ONE real spherical-harmonics basis function per lane (degree <= 3, 16 bases): the layout of
// sh_fwd_k16_kernel (sh.hip) and of the data-parallel gradient reduction (dp.hip), where lane k of a 16-lane
// row owns basis k.  Real SH factor as  b_k = (a0 + a1 z + a2 z^2 + a3 z^3) * s_k  with
// s_k in {1, x, y, 2xy, x^2-y^2, fS2, fC2}  (associated Legendre polynomial in z times the azimuthal factor);
// constants as in gsplat 1.4.0 sh_coeffs_to_color_fast (restated in oracle/gsplat_oracle.c).
#pragma once
#include "common.hpp"

namespace {

struct ShLaneConst { float a0, a1, a2, a3; int sel; };
__device__ __forceinline__ ShLaneConst sh_lane_const(int k) {
    // sel: 0 = 1, 1 = x, 2 = y, 3 = fS1 (2xy), 4 = fC1 (x^2-y^2), 5 = fS2, 6 = fC2
    switch (k) {
        case 0: return {0.2820947917738781f, 0.f, 0.f, 0.f, 0};
        case 1: return {-0.48860251190292f, 0.f, 0.f, 0.f, 2};
        case 2: return {0.f, 0.48860251190292f, 0.f, 0.f, 0};
        case 3: return {-0.48860251190292f, 0.f, 0.f, 0.f, 1};
        case 4: return {0.5462742152960395f, 0.f, 0.f, 0.f, 3};
        case 5: return {0.f, -1.092548430592079f, 0.f, 0.f, 2};
        case 6: return {-0.3153915652525201f, 0.f, 0.9461746957575601f, 0.f, 0};
        case 7: return {0.f, -1.092548430592079f, 0.f, 0.f, 1};
        case 8: return {0.5462742152960395f, 0.f, 0.f, 0.f, 4};
        case 9: return {-0.5900435899266435f, 0.f, 0.f, 0.f, 5};
        case 10: return {0.f, 1.445305721320277f, 0.f, 0.f, 3};
        case 11: return {0.4570457994644658f, 0.f, -2.285228997322329f, 0.f, 2};
        case 12: return {0.f, -1.119528997770346f, 0.f, 1.865881662950577f, 0};
        case 13: return {0.4570457994644658f, 0.f, -2.285228997322329f, 0.f, 1};
        case 14: return {0.f, 1.445305721320277f, 0.f, 0.f, 4};
        default: return {-0.5900435899266435f, 0.f, 0.f, 0.f, 6};
    }
}
// (x, y, z) must be a unit vector.  MAXDEG bounds the azimuthal factors that are evaluated.
template <int MAXDEG>
__device__ __forceinline__ float sh_lane_basis(const ShLaneConst &lc, float x, float y, float z) {
    float sfac = 1.f;
    if (MAXDEG >= 1) {
        sfac = lc.sel == 1 ? x : sfac;
        sfac = lc.sel == 2 ? y : sfac;
    }
    if (MAXDEG >= 2) {
        const float fS1 = 2.f * x * y, fC1 = x * x - y * y;
        sfac = lc.sel == 3 ? fS1 : sfac;
        sfac = lc.sel == 4 ? fC1 : sfac;
        if (MAXDEG >= 3) {
            sfac = lc.sel == 5 ? x * fS1 + y * fC1 : sfac;
            sfac = lc.sel == 6 ? x * fC1 - y * fS1 : sfac;
        }
    }
    return (lc.a0 + z * (lc.a1 + z * (lc.a2 + z * lc.a3))) * sfac;
}

}  // namespace
