// sort.hip -- stable LSD radix sort of (int64 key, int32 value) pairs on the low key_bits bits.
//
// Replaces the cub::DeviceRadixSort::SortPairs call inside gsplat 1.4.0 isect_tiles (sort=True),
// reached from gsplat.rendering.rasterization (/root/reference/mtgs/scene_model/
// mtgs_scene_graph.py:641-662).  Keys are cam | tile | fp32-depth-bits, so only
// 32 + tile_bits + cam_bits bits are significant (46 at 1920x1080, one camera).
//
// Hand-written for gfx950 (radix_sort.hpp): 8-bit digits, hist / scan / reorder per pass, wave64
// ballot ranking.  Roofline: HBM, 8 B + 2 x 12 B per pair per pass.  Bit-exact (stable) against
// oracle/gsplat_oracle.c::orc_sort_pairs.  (The default binning path, bin.hip, needs only
// 32-bit keys and far fewer passes; this entry point is the gsplat-shaped formulation.)
#include "common.hpp"
#include "radix_sort.hpp"

extern "C" int mtgs_sort_workspace_bytes(int64_t M, size_t *bytes) {
    MTGS_REQUIRE(M >= 0 && bytes, MTGS_EINVAL, "mtgs_sort_workspace_bytes: bad arguments");
    *bytes = mtgs_sort::workspace_bytes<uint64_t>(M > 0 ? M : 1);
    return MTGS_OK;
}

extern "C" int mtgs_sort_pairs(int64_t M, int key_bits, int64_t *keys_in, int32_t *vals_in,
                               int64_t *keys_out, int32_t *vals_out, void *ws, size_t ws_bytes,
                               void *stream) {
    MTGS_REQUIRE(M >= 0 && key_bits > 0 && key_bits <= 64, MTGS_EINVAL,
                 "mtgs_sort_pairs: bad arguments M=%lld key_bits=%d", (long long)M, key_bits);
    if (M == 0) return MTGS_OK;
    MTGS_REQUIRE(keys_in && vals_in && keys_out && vals_out && ws, MTGS_EINVAL, "mtgs_sort_pairs: null pointer");
    return mtgs_sort::sort_pairs<uint64_t>(M, key_bits, (const uint64_t *)keys_in, vals_in, (uint64_t *)keys_out,
                                           vals_out, ws, ws_bytes, (hipStream_t)stream, "mtgs_sort_pairs");
}
