// isect.hip -- tile/Gaussian intersection: count, prefix sum, key emission, per-tile offsets.
//
// Replaces gsplat 1.4.0 isect_tiles (two passes around a cumsum) and isect_offset_encode, the
// binning stage of gsplat.rendering.rasterization
// (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662, tile_size=16 at :640).
// Integer stage: results are bit-exact against oracle/gsplat_oracle.c on identical inputs.
//
// Roofline: HBM.  Algorithmic bytes: count 12/Gaussian in + 4 out; scan 4 in + 8 out;
// emit 16/visible Gaussian in + 12/intersection out; offsets 8/intersection in + 4/tile out.
//
// CDNA4 mapping of the emit pass: a Gaussian's tile rectangle ranges from 1 tile to the whole
// image (sky / near Gaussians cover thousands).  One thread per Gaussian would leave 63 lanes idle
// behind the largest footprint in the wave, so lanes write small rectangles themselves and every
// LARGE rectangle is emitted by the whole wave (64 consecutive slots per step -> coalesced
// 8-byte and 4-byte stores).
#include "common.hpp"
#include "tile_rect.hpp"
#include "scan.hpp"

namespace {

constexpr int ISECT_BLOCK = 256;
constexpr int SMALL_RECT = 8;  // rectangles up to this many tiles are written by their own lane


__global__ __launch_bounds__(ISECT_BLOCK) void isect_count_kernel(
    int64_t total, const float *__restrict__ means2d, const int32_t *__restrict__ radii, float ts,
    int tw, int th, int32_t *__restrict__ tiles_per_gauss) {
    const int64_t idx = (int64_t)blockIdx.x * ISECT_BLOCK + threadIdx.x;
    if (idx >= total) return;
    const int32_t r = radii[idx];
    int32_t cnt = 0;
    if (r > 0) {
        const float2 m = reinterpret_cast<const float2 *>(means2d)[idx];
        const Rect q = tile_rect(m.x, m.y, r, ts, tw, th);
        cnt = (q.x1 - q.x0) * (q.y1 - q.y0);
    }
    tiles_per_gauss[idx] = cnt;
}

struct TilesValue {
    const int32_t *in;
    __device__ __forceinline__ int64_t operator()(int64_t i) const { return in[i]; }
};
struct InclusiveSink {
    int64_t *out;
    __device__ __forceinline__ void operator()(int64_t i, int64_t /*excl*/, int64_t incl) const { out[i] = incl; }
};

// ---- emit ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(ISECT_BLOCK) void isect_emit_kernel(
    int64_t total, int64_t N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
    const float *__restrict__ depths, const int64_t *__restrict__ cum_tiles, float ts, int tw, int th,
    int tile_bits, int64_t *__restrict__ isect_ids, int32_t *__restrict__ flatten_ids) {
    const int64_t idx = (int64_t)blockIdx.x * ISECT_BLOCK + threadIdx.x;
    const int lane = lane_id();
    Rect q = {0, 0, 0, 0};
    int cnt = 0;
    int64_t cur = 0, key_hi_lo = 0;
    if (idx < total) {
        const int32_t r = radii[idx];
        if (r > 0) {
            const float2 m = reinterpret_cast<const float2 *>(means2d)[idx];
            q = tile_rect(m.x, m.y, r, ts, tw, th);
            cnt = (q.x1 - q.x0) * (q.y1 - q.y0);
            if (cnt > 0) {
                cur = idx == 0 ? 0 : cum_tiles[idx - 1];
                const int64_t cid = idx / N;
                const uint32_t dbits = __float_as_uint(depths[idx]);
                key_hi_lo = (cid << (32 + tile_bits)) | (int64_t)dbits;
            }
        }
    }
    // small footprints: own lane
    if (cnt > 0 && cnt <= SMALL_RECT) {
        for (int i = q.y0; i < q.y1; ++i)
            for (int j = q.x0; j < q.x1; ++j) {
                const int64_t tile_id = (int64_t)i * tw + j;
                isect_ids[cur] = key_hi_lo | (tile_id << 32);
                flatten_ids[cur] = (int32_t)idx;
                ++cur;
            }
    }
    // large footprints: the whole wave emits one Gaussian at a time
    unsigned long long big = __ballot(cnt > SMALL_RECT);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bx0 = __shfl(q.x0, src, 64), by0 = __shfl(q.y0, src, 64), bx1 = __shfl(q.x1, src, 64);
        const int bcnt = __shfl(cnt, src, 64);
        const int64_t bcur = __shfl(cur, src, 64), bkey = __shfl(key_hi_lo, src, 64);
        const int32_t bidx = (int32_t)__shfl(idx, src, 64);
        const int bw = bx1 - bx0;
        for (int k = lane; k < bcnt; k += 64) {
            const int i = by0 + k / bw, j = bx0 + k % bw;
            const int64_t tile_id = (int64_t)i * tw + j;
            isect_ids[bcur + k] = bkey | (tile_id << 32);
            flatten_ids[bcur + k] = bidx;
        }
    }
}

// ---- offsets --------------------------------------------------------------------------------------
__global__ __launch_bounds__(ISECT_BLOCK) void isect_offsets_kernel(
    int64_t M, const int64_t *__restrict__ ids, int n_tiles, int tile_bits, int64_t total_tiles,
    int32_t *__restrict__ offsets) {
    const int64_t i = (int64_t)blockIdx.x * ISECT_BLOCK + threadIdx.x;
    if (i >= M) return;
    const int64_t mask = ((int64_t)1 << tile_bits) - 1;
    const int64_t hi = ids[i] >> 32;
    const int64_t slot = (hi >> tile_bits) * n_tiles + (hi & mask);
    int64_t first;  // first slot whose offset equals i
    if (i == 0) {
        first = 0;
    } else {
        const int64_t ph = ids[i - 1] >> 32;
        first = (ph >> tile_bits) * n_tiles + (ph & mask) + 1;
    }
    for (int64_t s = first; s <= slot; ++s) offsets[s] = (int32_t)i;
    if (i == M - 1)
        for (int64_t s = slot + 1; s < total_tiles; ++s) offsets[s] = (int32_t)M;
}

__global__ void fill_i32_kernel(int64_t n, int32_t v, int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

__device__ __host__ inline int bit_length(uint32_t v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

}  // namespace

extern "C" int mtgs_isect_count(int C, int64_t N, const float *means2d, const int32_t *radii,
                                int tile_size, int tile_w, int tile_h, int32_t *tiles_per_gauss,
                                void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && tile_size > 0 && tile_w > 0 && tile_h > 0, MTGS_EINVAL,
                 "mtgs_isect_count: bad sizes");
    const int64_t total = (int64_t)C * N;
    if (total == 0) return MTGS_OK;
    MTGS_REQUIRE(means2d && radii && tiles_per_gauss, MTGS_EINVAL, "mtgs_isect_count: null pointer");
    isect_count_kernel<<<(unsigned)ceil_div64(total, ISECT_BLOCK), ISECT_BLOCK, 0, (hipStream_t)stream>>>(
        total, means2d, radii, (float)tile_size, tile_w, tile_h, tiles_per_gauss);
    MTGS_CHECK_LAUNCH("mtgs_isect_count");
    return MTGS_OK;
}

extern "C" int mtgs_scan_workspace_bytes(int64_t n, size_t *bytes) {
    MTGS_REQUIRE(n >= 0 && bytes, MTGS_EINVAL, "mtgs_scan_workspace_bytes: bad arguments");
    *bytes = mtgs_scan::workspace_bytes(n);
    return MTGS_OK;
}

extern "C" int mtgs_isect_scan(int64_t n, const int32_t *tiles_per_gauss, int64_t *cum_tiles,
                               int64_t *total, void *ws, size_t ws_bytes, void *stream) {
    MTGS_REQUIRE(n >= 0, MTGS_EINVAL, "mtgs_isect_scan: negative n");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (total) {
            hipError_t e = hipMemsetAsync(total, 0, sizeof(int64_t), st);
            MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_isect_scan: memset failed");
        }
        return MTGS_OK;
    }
    MTGS_REQUIRE(tiles_per_gauss && cum_tiles && ws, MTGS_EINVAL, "mtgs_isect_scan: null pointer");
    MTGS_REQUIRE(ws_bytes >= mtgs_scan::workspace_bytes(n), MTGS_EWORKSPACE,
                 "mtgs_isect_scan: workspace %zu < %zu bytes", ws_bytes, mtgs_scan::workspace_bytes(n));
    mtgs_scan::run(n, TilesValue{tiles_per_gauss}, InclusiveSink{cum_tiles}, (int64_t *)ws, total, st);
    MTGS_CHECK_LAUNCH("mtgs_isect_scan");
    return MTGS_OK;
}

extern "C" int mtgs_isect_emit(int C, int64_t N, const float *means2d, const int32_t *radii,
                               const float *depths, const int64_t *cum_tiles, int tile_size,
                               int tile_w, int tile_h, int64_t *isect_ids, int32_t *flatten_ids,
                               void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && tile_size > 0 && tile_w > 0 && tile_h > 0, MTGS_EINVAL,
                 "mtgs_isect_emit: bad sizes");
    const int64_t total = (int64_t)C * N;
    if (total == 0) return MTGS_OK;
    MTGS_REQUIRE(means2d && radii && depths && cum_tiles, MTGS_EINVAL, "mtgs_isect_emit: null pointer");
    const int tile_bits = bit_length((uint32_t)(tile_w * tile_h));
    isect_emit_kernel<<<(unsigned)ceil_div64(total, ISECT_BLOCK), ISECT_BLOCK, 0, (hipStream_t)stream>>>(
        total, N, means2d, radii, depths, cum_tiles, (float)tile_size, tile_w, tile_h, tile_bits,
        isect_ids, flatten_ids);
    MTGS_CHECK_LAUNCH("mtgs_isect_emit");
    return MTGS_OK;
}

extern "C" int mtgs_isect_offsets(int64_t M, const int64_t *isect_ids_sorted, int C, int tile_w,
                                  int tile_h, int32_t *offsets, void *stream) {
    MTGS_REQUIRE(M >= 0 && C >= 0 && tile_w > 0 && tile_h > 0, MTGS_EINVAL, "mtgs_isect_offsets: bad sizes");
    MTGS_REQUIRE(M < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_isect_offsets: M must fit int32");
    const int n_tiles = tile_w * tile_h;
    const int64_t total_tiles = (int64_t)C * n_tiles;
    if (total_tiles == 0) return MTGS_OK;
    MTGS_REQUIRE(offsets, MTGS_EINVAL, "mtgs_isect_offsets: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        fill_i32_kernel<<<(unsigned)ceil_div64(total_tiles, 256), 256, 0, st>>>(total_tiles, 0, offsets);
    } else {
        MTGS_REQUIRE(isect_ids_sorted, MTGS_EINVAL, "mtgs_isect_offsets: null pointer");
        isect_offsets_kernel<<<(unsigned)ceil_div64(M, ISECT_BLOCK), ISECT_BLOCK, 0, st>>>(
            M, isect_ids_sorted, n_tiles, bit_length((uint32_t)n_tiles), total_tiles, offsets);
    }
    MTGS_CHECK_LAUNCH("mtgs_isect_offsets");
    return MTGS_OK;
}
