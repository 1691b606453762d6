// bin.hip -- depth-ordered tile binning: the fast path behind gsplat's isect_tiles(sort=True).
//
// gsplat 1.4.0 isect_tiles emits one 64-bit key (cam | tile | depth bits) per tile/Gaussian
// intersection and radix-sorts all M of them on 46 bits: six 8-bit passes over 12-byte pairs
// (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662 -> gsplat.rendering.rasterization
// -> isect_tiles).  M is ~13x the number of visible Gaussians, so this path splits the key instead:
//
//   1. mtgs_bin_compact   visible Gaussians -> (cam | depth bits, index) in index order, plus the
//                         two totals the host needs (n_vis and M) from ONE packed int64 scan;
//   2. mtgs_sort_pairs    stable sort of the n_vis pairs by (cam, depth)          [small]
//   3. mtgs_bin_scan      prefix sum of tiles_per_gauss in that depth order
//   4. mtgs_bin_emit      intersections emitted in depth order: (cam*n_tiles + tile, index)
//   5. mtgs_sort_pairs_u32  stable sort on the tile bits only (13 bits -> 2 passes, 8-byte pairs)
//      (both sorts are the hand-written wave64 radix sort of radix_sort.hpp)
//   6. mtgs_bin_finalize  rebuilds the 64-bit isect_ids (gsplat's meta output) from the sorted pairs
//
// A stable sort by tile of a (depth, index)-ordered sequence is the (tile, depth, index) order, i.e.
// exactly what the reference's single stable sort of emission-ordered keys produces: isect_ids and
// flatten_ids are bit-identical (tests/test_gpu_parity.py).  HBM traffic of the sorting drops from
// 6 x 2 x 12 B x M to 2 x 2 x 8 B x M (+ the small per-Gaussian sort).
//
// Roofline: HBM.  Algorithmic bytes: compact C*N*12 in + n_vis*12 out; scan n_vis*8 in + 8 out;
// emit n_vis*24 in + M*8 out; finalize M*8 in + M*8 (+4 gathered) out.
#include "common.hpp"
#include "tile_rect.hpp"
#include "radix_sort.hpp"
#include "scan.hpp"

namespace {

constexpr int BIN_BLOCK = 256;


// packed scan value: visible flag in the high word, tile count in the low word
struct PackedVisTiles {
    const int32_t *radii, *tpg;
    __device__ __forceinline__ int64_t operator()(int64_t i) const {
        return ((int64_t)(radii[i] > 0) << 32) | (int64_t)(uint32_t)tpg[i];
    }
};
struct CompactSink {
    const int32_t *radii;
    const float *depths;
    int64_t N;
    int64_t *keys;
    int32_t *ids;
    int32_t *rank;  // nullable: rank[i] = position of visible element i in the compacted list
    // called with the EXCLUSIVE packed prefix of element i
    __device__ __forceinline__ void operator()(int64_t i, int64_t excl, int64_t /*incl*/) const {
        if (radii[i] > 0) {
            const int64_t pos = excl >> 32;
            keys[pos] = ((i / N) << 32) | (int64_t)__float_as_uint(depths[i]);
            ids[pos] = (int32_t)i;
            if (rank) rank[i] = (int32_t)pos;
        }
    }
};
// final pass of mtgs_bin_compact's scan, specialised: a thread owns 8 consecutive (camera, Gaussian) pairs and reads
// their radii / tile counts as two 16-byte loads each (the generic mtgs_scan::final_kernel issues 8 element loads at a
// 32-byte lane stride per array and ran at 2 TB/s); depths are only gathered for the visible ones.
__global__ __launch_bounds__(mtgs_scan::BLOCK) void bin_compact_final_kernel(
    int64_t n, int64_t N, const int32_t *__restrict__ radii, const int32_t *__restrict__ tpg,
    const float *__restrict__ depths, const int64_t *__restrict__ partials, int64_t *__restrict__ keys,
    int32_t *__restrict__ ids, int32_t *__restrict__ rank) {
    constexpr int ITEMS = mtgs_scan::ITEMS;
    static_assert(ITEMS == 8, "two int4 loads per array");
    __shared__ int64_t lds[mtgs_scan::BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * mtgs_scan::TILE + (int64_t)threadIdx.x * ITEMS;
    int32_t r[ITEMS], t[ITEMS];
    if (base + ITEMS <= n) {
        const int4 r0 = *reinterpret_cast<const int4 *>(radii + base), r1 = *reinterpret_cast<const int4 *>(radii + base + 4);
        const int4 t0 = *reinterpret_cast<const int4 *>(tpg + base), t1 = *reinterpret_cast<const int4 *>(tpg + base + 4);
        r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r0.w; r[4] = r1.x; r[5] = r1.y; r[6] = r1.z; r[7] = r1.w;
        t[0] = t0.x; t[1] = t0.y; t[2] = t0.z; t[3] = t0.w; t[4] = t1.x; t[5] = t1.y; t[6] = t1.z; t[7] = t1.w;
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            r[i] = base + i < n ? radii[base + i] : 0;
            t[i] = base + i < n ? tpg[base + i] : 0;
        }
    }
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) s += ((int64_t)(r[i] > 0) << 32) | (int64_t)(uint32_t)t[i];
    int64_t tot;
    int64_t run = partials[blockIdx.x] + mtgs_scan::block_exclusive_scan(s, lds, tot);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        if (r[i] > 0) {   // (implies base + i < n)
            const int64_t e = base + i, pos = run >> 32;
            keys[pos] = ((e / N) << 32) | (int64_t)__float_as_uint(depths[e]);
            ids[pos] = (int32_t)e;
            if (rank) rank[e] = (int32_t)pos;
        }
        run += ((int64_t)(r[i] > 0) << 32) | (int64_t)(uint32_t)t[i];
    }
}

struct GatherTiles {
    const int32_t *ids, *tpg;
    __device__ __forceinline__ int64_t operator()(int64_t r) const { return tpg[ids[r]]; }
};
struct InclusiveSink {
    int64_t *out;
    __device__ __forceinline__ void operator()(int64_t r, int64_t /*excl*/, int64_t incl) const { out[r] = incl; }
};

// Load-balanced emission: the work is split by OUTPUT slot, not by Gaussian.  In depth order the
// largest footprints are adjacent (the nearest Gaussians cover the whole image), so a per-Gaussian
// loop would serialise thousands of stores behind the first wavefronts.  A block owns 2048
// consecutive slots: one uniform binary search in the prefix sums finds the Gaussian of its first
// slot, the next 2048 prefix sums go to LDS, and every slot locates its Gaussian with an 11-step
// search in LDS (neighbouring lanes read the same words -> broadcasts).  Stores are fully coalesced.
constexpr int EMIT_ITEMS = 8, EMIT_TILE = BIN_BLOCK * EMIT_ITEMS;
__global__ __launch_bounds__(BIN_BLOCK) void bin_emit_kernel(
    int64_t M, int64_t n_vis, const int32_t *__restrict__ ids_sorted, int64_t N,
    const float *__restrict__ means2d, const int32_t *__restrict__ radii, const int64_t *__restrict__ cum,
    float ts, int tw, int th, uint32_t *__restrict__ tile_keys, int32_t *__restrict__ gids) {
    __shared__ int32_t s_c[EMIT_TILE];
    const int tid = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * EMIT_TILE;
    // r0 = first r with cum[r] > s0   (cum is the inclusive prefix sum, so slot s0 belongs to r0).  A 256-ary
    // search: every thread probes one position per round, so the block pays 3 dependent global loads at 300k
    // Gaussians instead of the 19 of a binary search.
    int64_t lo = 0, hi = n_vis;  // the answer lies in [lo, hi]
    while (lo < hi) {
        const int64_t step = (hi - lo + BIN_BLOCK - 1) / BIN_BLOCK;
        const int64_t pos = lo + (int64_t)tid * step;
        const bool below = pos < hi && cum[pos] <= s0;           // monotone in tid: true ... true false ... false
        const int cnt = __syncthreads_count(below);
        if (cnt == 0) { hi = lo; break; }
        const int64_t last_below = lo + (int64_t)(cnt - 1) * step;
        hi = min(hi, lo + (int64_t)cnt * step);
        lo = last_below + 1;
    }
    const int64_t r0 = lo;
    // every Gaussian owns >= 1 slot, so the block's slots touch at most Gaussians r0 .. r0+EMIT_TILE-1
#pragma unroll
    for (int e = 0; e < EMIT_ITEMS; ++e) {
        const int k = e * BIN_BLOCK + tid;
        s_c[k] = (r0 + k < n_vis) ? (int32_t)cum[r0 + k] : 0x7fffffff;
    }
    const int32_t base0 = r0 > 0 ? (int32_t)cum[r0 - 1] : 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EMIT_ITEMS; ++e) {
        const int64_t i = s0 + e * BIN_BLOCK + tid;
        if (i >= M) break;
        int l = 0, h = EMIT_TILE - 1;  // first l with s_c[l] > i; it exists and is <= EMIT_TILE-1
        while (l < h) {
            const int mid = (l + h) >> 1;
            if ((int64_t)s_c[mid] <= i) l = mid + 1; else h = mid;
        }
        const int32_t excl = l > 0 ? s_c[l - 1] : base0;
        const int32_t idx = ids_sorted[r0 + l];
        const float2 m = reinterpret_cast<const float2 *>(means2d)[idx];
        const Rect q = tile_rect(m.x, m.y, radii[idx], ts, tw, th);
        const int local = (int)(i - excl), bw = q.x1 - q.x0;
        const int row = local / bw, col = local - row * bw;
        // camera of flatten id idx: a 32-bit division (C*N < 2^31), none at all for the single camera MTGS renders
        const uint32_t cam = (uint32_t)idx < (uint32_t)N ? 0u : (uint32_t)idx / (uint32_t)N;
        tile_keys[i] = cam * (uint32_t)(tw * th) + (uint32_t)((q.y0 + row) * tw + q.x0 + col);
        gids[i] = idx;
    }
}

// last-pass epilogue of the tile sort: (tile key, Gaussian index) -> gsplat's 64-bit isect_id
struct IsectIdEpilogue {
    static constexpr bool enabled = true;
    const float *depths;
    uint32_t n_tiles;
    int tile_bits;
    int32_t *flatten_ids;
    int64_t *isect_ids;
    struct G { uint32_t depth_bits; };
    __device__ __forceinline__ G gather(int32_t gid) const { return G{__float_as_uint(depths[gid])}; }
    __device__ __forceinline__ void store(uint32_t dst, uint64_t key, int32_t gid, const G &g) const {
        const int64_t cam = (uint32_t)key / n_tiles, tile = (uint32_t)key % n_tiles;
        isect_ids[dst] = (cam << (32 + tile_bits)) | (tile << 32) | (int64_t)g.depth_bits;
        flatten_ids[dst] = gid;
    }
};

__global__ __launch_bounds__(BIN_BLOCK) void bin_finalize_kernel(
    int64_t M, const uint32_t *__restrict__ tile_keys, const int32_t *__restrict__ gids,
    const float *__restrict__ depths, int n_tiles, int tile_bits, int64_t *__restrict__ isect_ids) {
    const int64_t i = (int64_t)blockIdx.x * BIN_BLOCK + threadIdx.x;
    if (i >= M) return;
    const uint32_t k = tile_keys[i];
    const int64_t cam = k / (uint32_t)n_tiles, tile = k % (uint32_t)n_tiles;
    isect_ids[i] = (cam << (32 + tile_bits)) | (tile << 32) | (int64_t)__float_as_uint(depths[gids[i]]);
}

inline int bit_length_u32(uint32_t v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

}  // namespace

extern "C" int mtgs_bin_compact(int C, int64_t N, const int32_t *radii, const float *depths,
                                const int32_t *tiles_per_gauss, int64_t *vis_keys, int32_t *vis_ids,
                                int32_t *vis_rank, int64_t *totals, int64_t *host_totals, int64_t host_tag, void *ws,
                                size_t ws_bytes, void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0, MTGS_EINVAL, "mtgs_bin_compact: bad sizes");
    const int64_t total = (int64_t)C * N;
    hipStream_t st = (hipStream_t)stream;
    MTGS_REQUIRE(totals, MTGS_EINVAL, "mtgs_bin_compact: null pointer");
    if (total == 0) {
        hipError_t e = hipMemsetAsync(totals, 0, sizeof(int64_t), st);
        MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_bin_compact: memset failed");
        if (host_totals) {  // host memory: nothing to wait for
            host_totals[0] = 0;
            __atomic_store_n(host_totals + 1, host_tag, __ATOMIC_RELEASE);
        }
        return MTGS_OK;
    }
    MTGS_REQUIRE(radii && depths && tiles_per_gauss && vis_keys && vis_ids && ws, MTGS_EINVAL, "mtgs_bin_compact: null pointer");
    MTGS_REQUIRE(ws_bytes >= mtgs_scan::workspace_bytes(total), MTGS_EWORKSPACE, "mtgs_bin_compact: workspace too small");
    if ((reinterpret_cast<uintptr_t>(radii) | reinterpret_cast<uintptr_t>(tiles_per_gauss)) & 15) {
        mtgs_scan::run(total, PackedVisTiles{radii, tiles_per_gauss}, CompactSink{radii, depths, N, vis_keys, vis_ids, vis_rank},
                       (int64_t *)ws, totals, st, host_totals, host_tag);
    } else {
        const int64_t nblocks = ceil_div64(total, mtgs_scan::TILE);
        mtgs_scan::partials_kernel<<<(unsigned)nblocks, mtgs_scan::BLOCK, 0, st>>>(total, PackedVisTiles{radii, tiles_per_gauss},
                                                                                   (int64_t *)ws);
        mtgs_scan::spine_kernel<0><<<1, mtgs_scan::BLOCK, 0, st>>>(nblocks, (int64_t *)ws, totals, host_totals, host_tag);
        bin_compact_final_kernel<<<(unsigned)nblocks, mtgs_scan::BLOCK, 0, st>>>(total, N, radii, tiles_per_gauss, depths,
                                                                                 (const int64_t *)ws, vis_keys, vis_ids, vis_rank);
    }
    MTGS_CHECK_LAUNCH("mtgs_bin_compact");
    return MTGS_OK;
}

extern "C" int mtgs_bin_scan(int64_t n_vis, const int32_t *ids_sorted, const int32_t *tiles_per_gauss,
                             int64_t *cum, void *ws, size_t ws_bytes, void *stream) {
    MTGS_REQUIRE(n_vis >= 0, MTGS_EINVAL, "mtgs_bin_scan: bad size");
    if (n_vis == 0) return MTGS_OK;
    MTGS_REQUIRE(ids_sorted && tiles_per_gauss && cum && ws, MTGS_EINVAL, "mtgs_bin_scan: null pointer");
    MTGS_REQUIRE(ws_bytes >= mtgs_scan::workspace_bytes(n_vis), MTGS_EWORKSPACE, "mtgs_bin_scan: workspace too small");
    mtgs_scan::run(n_vis, GatherTiles{ids_sorted, tiles_per_gauss}, InclusiveSink{cum}, (int64_t *)ws, nullptr,
                   (hipStream_t)stream);
    MTGS_CHECK_LAUNCH("mtgs_bin_scan");
    return MTGS_OK;
}

extern "C" int mtgs_bin_emit(int64_t M, int64_t n_vis, const int32_t *ids_sorted, int64_t N,
                             const float *means2d, const int32_t *radii, const int64_t *cum,
                             int tile_size, int tile_w, int tile_h, uint32_t *tile_keys, int32_t *gids,
                             void *stream) {
    MTGS_REQUIRE(M >= 0 && n_vis >= 0 && N >= 0 && tile_size > 0 && tile_w > 0 && tile_h > 0, MTGS_EINVAL, "mtgs_bin_emit: bad sizes");
    MTGS_REQUIRE(M < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_bin_emit: M must fit int32");
    if (n_vis == 0 || M == 0) return MTGS_OK;
    MTGS_REQUIRE(ids_sorted && means2d && radii && cum, MTGS_EINVAL, "mtgs_bin_emit: null pointer");
    bin_emit_kernel<<<(unsigned)ceil_div64(M, EMIT_TILE), BIN_BLOCK, 0, (hipStream_t)stream>>>(
        M, n_vis, ids_sorted, N, means2d, radii, cum, (float)tile_size, tile_w, tile_h, tile_keys, gids);
    MTGS_CHECK_LAUNCH("mtgs_bin_emit");
    return MTGS_OK;
}

extern "C" int mtgs_sort_u32_workspace_bytes(int64_t M, size_t *bytes) {
    MTGS_REQUIRE(M >= 0 && bytes, MTGS_EINVAL, "mtgs_sort_u32_workspace_bytes: bad arguments");
    *bytes = mtgs_sort::workspace_bytes<uint32_t>(M > 0 ? M : 1);
    return MTGS_OK;
}

extern "C" int mtgs_sort_pairs_u32(int64_t M, int key_bits, uint32_t *keys_in, int32_t *vals_in,
                                   uint32_t *keys_out, int32_t *vals_out, void *ws, size_t ws_bytes,
                                   void *stream) {
    MTGS_REQUIRE(M >= 0 && key_bits > 0 && key_bits <= 32, MTGS_EINVAL, "mtgs_sort_pairs_u32: bad arguments");
    if (M == 0) return MTGS_OK;
    MTGS_REQUIRE(keys_in && vals_in && keys_out && vals_out && ws, MTGS_EINVAL, "mtgs_sort_pairs_u32: null pointer");
    return mtgs_sort::sort_pairs<uint32_t>(M, key_bits, keys_in, vals_in, keys_out, vals_out, ws, ws_bytes,
                                           (hipStream_t)stream, "mtgs_sort_pairs_u32");
}

extern "C" int mtgs_bin_sort_tiles(int64_t M, int C, int tile_w, int tile_h, const uint32_t *tile_keys,
                                   const int32_t *gids, const float *depths, uint32_t *keys_scratch,
                                   int32_t *flatten_ids, int64_t *isect_ids, void *ws, size_t ws_bytes,
                                   void *stream) {
    MTGS_REQUIRE(M >= 0 && C > 0 && tile_w > 0 && tile_h > 0, MTGS_EINVAL, "mtgs_bin_sort_tiles: bad sizes");
    if (M == 0) return MTGS_OK;
    MTGS_REQUIRE(tile_keys && gids && depths && keys_scratch && flatten_ids && isect_ids && ws, MTGS_EINVAL,
                 "mtgs_bin_sort_tiles: null pointer");
    const uint32_t n_tiles = (uint32_t)(tile_w * tile_h);
    const int key_bits = bit_length_u32((uint32_t)C * n_tiles - 1u) > 0 ? bit_length_u32((uint32_t)C * n_tiles - 1u) : 1;
    return mtgs_sort::sort_pairs<uint32_t, IsectIdEpilogue>(
        M, key_bits, tile_keys, gids, keys_scratch, flatten_ids, ws, ws_bytes, (hipStream_t)stream,
        "mtgs_bin_sort_tiles", IsectIdEpilogue{depths, n_tiles, bit_length_u32(n_tiles), flatten_ids, isect_ids});
}

extern "C" int mtgs_bin_finalize(int64_t M, const uint32_t *tile_keys_sorted, const int32_t *flatten_ids,
                                 const float *depths, int C, int tile_w, int tile_h, int64_t *isect_ids,
                                 void *stream) {
    MTGS_REQUIRE(M >= 0 && C >= 0 && tile_w > 0 && tile_h > 0, MTGS_EINVAL, "mtgs_bin_finalize: bad sizes");
    if (M == 0) return MTGS_OK;
    MTGS_REQUIRE(tile_keys_sorted && flatten_ids && depths && isect_ids, MTGS_EINVAL, "mtgs_bin_finalize: null pointer");
    const int n_tiles = tile_w * tile_h;
    bin_finalize_kernel<<<(unsigned)ceil_div64(M, BIN_BLOCK), BIN_BLOCK, 0, (hipStream_t)stream>>>(
        M, tile_keys_sorted, flatten_ids, depths, n_tiles, bit_length_u32((uint32_t)n_tiles), isect_ids);
    MTGS_CHECK_LAUNCH("mtgs_bin_finalize");
    return MTGS_OK;
}

// ---- one call for everything after the host learns (n_vis, M) ---------------------------------
// The kernels above are short (5-40 us); launched one ctypes call at a time the host cannot keep the
// GPU fed and ~60 us of gaps open per frame.  mtgs_bin_build enqueues the whole chain back to back.
namespace {
struct BinWorkspace {
    int64_t *keys_s, *cum;
    int32_t *ids_s, *gids;
    uint32_t *tile_keys, *keys_scratch;
    void *sort_ws;
    size_t sort_bytes;
    void *scan_ws;
    size_t scan_bytes;
    size_t total;
};
inline BinWorkspace carve(char *base, int64_t n_vis, int64_t M) {
    BinWorkspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += mtgs_sort::align256(bytes); return p; };
    const int64_t nv = n_vis > 0 ? n_vis : 1, m = M > 0 ? M : 1;
    w.keys_s = (int64_t *)take((size_t)nv * 8);
    w.cum = (int64_t *)take((size_t)nv * 8);
    w.ids_s = (int32_t *)take((size_t)nv * 4);
    w.tile_keys = (uint32_t *)take((size_t)m * 4);
    w.gids = (int32_t *)take((size_t)m * 4);
    w.keys_scratch = (uint32_t *)take((size_t)m * 4);
    const size_t s1 = mtgs_sort::workspace_bytes<uint64_t>(nv), s2 = mtgs_sort::workspace_bytes<uint32_t>(m);
    w.sort_bytes = s1 > s2 ? s1 : s2;
    w.sort_ws = take(w.sort_bytes);
    w.scan_bytes = mtgs_scan::workspace_bytes(nv);
    w.scan_ws = take(w.scan_bytes);
    w.total = off;
    return w;
}
}  // namespace

extern "C" int mtgs_bin_workspace_bytes(int64_t n_vis, int64_t M, size_t *bytes) {
    MTGS_REQUIRE(n_vis >= 0 && M >= 0 && bytes, MTGS_EINVAL, "mtgs_bin_workspace_bytes: bad arguments");
    *bytes = carve(nullptr, n_vis, M).total;
    return MTGS_OK;
}

extern "C" int mtgs_bin_build(int C, int64_t N, int64_t n_vis, int64_t M, const float *means2d,
                              const int32_t *radii, const float *depths, const int32_t *tiles_per_gauss,
                              const int64_t *vis_keys, const int32_t *vis_ids, int tile_size, int tile_w,
                              int tile_h, int64_t *isect_ids, int32_t *flatten_ids, int32_t *offsets,
                              int32_t *tile_order, void *ws, size_t ws_bytes, void *stream) {
    MTGS_REQUIRE(C > 0 && N >= 0 && n_vis >= 0 && M >= 0 && tile_size > 0 && tile_w > 0 && tile_h > 0, MTGS_EINVAL,
                 "mtgs_bin_build: bad sizes");
    MTGS_REQUIRE(ws, MTGS_EINVAL, "mtgs_bin_build: null workspace");
    BinWorkspace w = carve((char *)ws, n_vis, M);
    MTGS_REQUIRE(ws_bytes >= w.total, MTGS_EWORKSPACE, "mtgs_bin_build: workspace %zu < %zu bytes", ws_bytes, w.total);
    int rc = MTGS_OK;
    if (M > 0) {
        int cam_bits = 0;
        for (uint32_t v = (uint32_t)(C - 1); v; v >>= 1) ++cam_bits;
        // the sort reads its inputs only, so vis_keys / vis_ids stay intact
        rc = mtgs_sort::sort_pairs<uint64_t>(n_vis, 32 + cam_bits, (const uint64_t *)vis_keys, vis_ids,
                                             (uint64_t *)w.keys_s, w.ids_s, w.sort_ws, w.sort_bytes,
                                             (hipStream_t)stream, "mtgs_bin_build(depth sort)");
        if (rc) return rc;
        if ((rc = mtgs_bin_scan(n_vis, w.ids_s, tiles_per_gauss, w.cum, w.scan_ws, w.scan_bytes, stream))) return rc;
        if ((rc = mtgs_bin_emit(M, n_vis, w.ids_s, N, means2d, radii, w.cum, tile_size, tile_w, tile_h, w.tile_keys,
                                w.gids, stream))) return rc;
        if ((rc = mtgs_bin_sort_tiles(M, C, tile_w, tile_h, w.tile_keys, w.gids, depths, w.keys_scratch, flatten_ids,
                                      isect_ids, w.sort_ws, w.sort_bytes, stream))) return rc;
    }
    if (offsets) {
        if ((rc = mtgs_isect_offsets(M, isect_ids, C, tile_w, tile_h, offsets, stream))) return rc;
        if (tile_order && (rc = mtgs_tile_schedule(C, tile_w, tile_h, offsets, M, tile_order, stream))) return rc;
    }
    return MTGS_OK;
}
