// sort.hip -- stable LSD radix sort of (int64 key, int32 value) pairs on the low key_bits bits.
//
// Replaces the cub::DeviceRadixSort::SortPairs call inside gsplat 1.4.0 isect_tiles (sort=True),
// reached from gsplat.rendering.rasterization (/root/reference/mtgs/scene_model/
// mtgs_scene_graph.py:641-662).  Keys are cam | tile | fp32-depth-bits, so only
// 32 + tile_bits + cam_bits bits are significant (46 at 1920x1080, one camera).
//
// Round-1 implementation: rocPRIM's device radix sort (the ROCm counterpart of the CUB call the
// reference makes), restricted to the significant bits.  Roofline: HBM, 2 x 12 B per pair per
// 8-bit digit pass.  Bit-exact (stable) against oracle/gsplat_oracle.c::orc_sort_pairs.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

extern "C" int mtgs_sort_workspace_bytes(int64_t M, size_t *bytes) {
    MTGS_REQUIRE(M >= 0 && bytes, MTGS_EINVAL, "mtgs_sort_workspace_bytes: bad arguments");
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, (const int64_t *)nullptr, (int64_t *)nullptr,
                                             (const int32_t *)nullptr, (int32_t *)nullptr,
                                             (size_t)(M > 0 ? M : 1), 0u, 64u, (hipStream_t)0);
    MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_sort_workspace_bytes: %s", hipGetErrorString(e));
    *bytes = tmp < 16 ? 16 : tmp;
    return MTGS_OK;
}

extern "C" int mtgs_sort_pairs(int64_t M, int key_bits, int64_t *keys_in, int32_t *vals_in,
                               int64_t *keys_out, int32_t *vals_out, void *ws, size_t ws_bytes,
                               void *stream) {
    MTGS_REQUIRE(M >= 0 && key_bits > 0 && key_bits <= 64, MTGS_EINVAL,
                 "mtgs_sort_pairs: bad arguments M=%lld key_bits=%d", (long long)M, key_bits);
    if (M == 0) return MTGS_OK;
    MTGS_REQUIRE(keys_in && vals_in && keys_out && vals_out && ws, MTGS_EINVAL, "mtgs_sort_pairs: null pointer");
    size_t need = ws_bytes;
    hipError_t e = rocprim::radix_sort_pairs(ws, need, (const int64_t *)keys_in, keys_out,
                                             (const int32_t *)vals_in, vals_out, (size_t)M, 0u,
                                             (unsigned)key_bits, (hipStream_t)stream);
    MTGS_REQUIRE(e == hipSuccess, e == hipErrorInvalidValue ? MTGS_EWORKSPACE : MTGS_ELAUNCH,
                 "mtgs_sort_pairs: rocprim::radix_sort_pairs: %s", hipGetErrorString(e));
    return MTGS_OK;
}
