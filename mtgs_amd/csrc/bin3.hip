// bin3.hip -- tile binning of the fused rasterization path WITHOUT a global sort.
//
// Same result as gsplat 1.4.0 isect_tiles(sort=True) + isect_offset_encode (the binning stage of
// gsplat.rendering.rasterization, /root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662): flatten_ids,
// isect_ids and isect_offsets are bit-identical (tests/test_gpu_parity.py, tests/test_gpu_fused.py).
//
// gsplat sorts all M intersections on a 64-bit (camera | tile | depth) key: six radix passes over 12-byte pairs.
// bin.hip (operator level) sorts the visible Gaussians by depth and then the intersections on the tile bits: 4 + 2
// passes (its device-sized, 17-launch form was this path until round 2: 241 us at the headline size against 160 us
// here).  Here no array is sorted globally at all.  The order gsplat defines is "by tile, then by depth, then by Gaussian index"; which
// intersections belong to a tile does not depend on any order, only their arrangement INSIDE the tile's list does:
//   1. bin3_rows_{count,place}_kernel   (Gaussian, tile row) ITEMS grouped by row: 3.6 per visible Gaussian; the
//                             count pass hands every (workgroup, row) pair its base inside the row's segment with one
//                             returning atomic, the place pass numbers the items with LDS atomics;
//   2. bin3_tiles_count_kernel   a workgroup's 2048 items lie in one or two rows, so the tiles they cover are a short
//                             range of the LDS histogram: per-tile counts; the block that finishes last turns them into
//                             isect_offsets and into the longest-list-first dispatch order of the tiles;
//   3. bin3_tiles_place_kernel   the items are expanded into intersections: a workgroup reserves its slots in a tile's
//                             segment with one returning atomic per (workgroup, tile) and stores the 64-bit keys
//                             depth bits << 32 | rank  in runs of ~60: the segments now hold the right SETS, unordered;
//   4. bin3_sort_{small,large}_kernel   every tile's segment is sorted on that key by ONE wave (< 1020 keys), one
//                             workgroup (<= 4096) or one 1024-thread workgroup in LDS -- a bitonic network in its
//                             all-ascending form, 8 or 16 keys per thread, three or four stages per LDS round trip,
//                             compare-exchange = v_min_f64 + v_max_f64, XOR-swizzled layout -- and written out as
//                             rank_ids, flatten_ids and gsplat's isect_ids.  rank order == Gaussian index order, so
//                             the key order IS gsplat's (tile, depth, index) order, and the result does not depend on
//                             the order in which the atomics of steps 1-3 were served.
// Why two levels: a scatter of the M intersections straight into 8160 tile segments is M single 8-byte stores to random
// lines (measured 63 us for 4M, with or without a per-intersection atomic: 85 us); splitting by row first keeps every
// store run long -- the principle of an MSD radix sort, applied to the 1.1M row items instead of the 4M intersections.
// Segments longer than the LDS holds (16384 keys) are sorted in 16384-key chunks and merged through global memory
// by the same workgroup.
// Every kernel reads its element count from device memory and its grid is sized for a CAPACITY (speculative sizing /
// graph capture, see wrapper.py); seven launches per frame behind the one that clears the counters.
//
// Roofline: HBM / latency for 1-3 (rows 2 * n_vis*64 + items*8; tiles items*8*2 + M*8), LDS / VALU for 4
// (M*(8+16) bytes of HBM against ~log2(L)^2/2 compare-exchanges per key).
#include "common.hpp"
#include "tile_rect.hpp"
#include "devsize.hpp"
#include "raster_rec.hpp"

namespace {

using mtgs_os::SizeRef;

constexpr int B3_BLOCK = 256;
constexpr int MAX_BINS = 32768;  // (camera, tile) pairs whose counts fit the counting kernels' LDS (3840x2160: 32400 tiles)

// ---- 1. row items: (Gaussian, tile row) pairs grouped by row ------------------------------------------------------
// An item = {rank, first tile of the Gaussian in this row << 12 | number of tiles - 1}.
struct Item { uint32_t rank, span; };
constexpr int SPAN_BITS = 12;   // row widths up to 2^12 tiles, tile ids below 2^20 (MAX_BINS)
constexpr int MAX_ROWS = 4096;  // (camera, tile row) bins of the row kernels' LDS histogram (60 cameras at 1080p)

struct RowGeom {
    int row0, h, tile0, w;   // first (camera, row) bin, rows, first tile id of the first row, tiles per row
    int y0, x0;              // first tile row / column inside the camera's grid
    float mx, my, ca, cb, cc, s2max;   // TIGHT lists: the {alpha >= 1/255} ellipse of the Gaussian (raster_rec.hpp)
};
template <bool TIGHT>
__device__ __forceinline__ RowGeom row_geom(const float *__restrict__ recs, const int32_t *__restrict__ vis_ids, int64_t rank,
                                            int64_t N, int C, float ts, int tw, int th) {
    const float4 *rec = reinterpret_cast<const float4 *>(recs + rank * REC_FLOATS);
    const float4 r0 = rec[0], r1 = rec[1];
    const Rect q = tile_rect(r0.x, r0.y, __float_as_int(r1.w), ts, tw, th);
    const int cam = C == 1 ? 0 : (int)((uint32_t)vis_ids[rank] / (uint32_t)N);
    RowGeom g;
    g.row0 = cam * th + q.y0; g.h = q.y1 - q.y0; g.w = q.x1 - q.x0;
    g.tile0 = g.row0 * tw + q.x0;
    g.y0 = q.y0; g.x0 = q.x0;
    g.mx = r0.x; g.my = r0.y; g.ca = r0.z; g.cb = r0.w; g.cc = r1.x; g.s2max = r1.z;
    if (g.w <= 0) g.h = 0;
    if (TIGHT && !(g.s2max >= 0.f)) g.h = 0;   // opacity < 1/255 (or NaN): no pixel anywhere (the staging of blend.hip drops it too)
    return g;
}
// The tiles of row y (0 .. h-1 of the Gaussian's 3-sigma square) that belong to the lists: gsplat's -- all w of them -- or,
// TIGHT, those whose pixel centres the {alpha >= 1/255} ellipse reaches.  The ellipse is convex, so these are an interval of the
// row (closed form below, with the margin of the exact test the compositing kernels apply when they stage a tile's candidates,
// rec_reaches_rect): no pair with a pixel that could pass the per-pixel test is left out, so pixel for pixel nothing changes,
// but the other pairs are never counted, placed, sorted or gathered.
// first = tile id, returns the number of tiles (<= 0: none).
template <bool TIGHT>
__device__ __forceinline__ int row_span(const RowGeom &g, int y, int tw, int &first) {
    first = g.tile0 + y * tw;
    if (!TIGHT) return g.w;
    const float a = g.ca, b = g.cb, c = g.cc, det = a * c - b * b;
    if (!(det > 0.f && a > 0.f && c > 0.f)) return g.w;          // (rec_reaches_rect keeps every tile of such a conic)
    const float ty = (float)((g.y0 + y) * MTGS_TILE_SIZE);
    const float Y0 = ty + 0.5f - g.my, Y1 = ty + 15.5f - g.my;   // the band of this row's pixel centres, relative to the mean
    // The x-extent of {q <= s} inside the band, in closed form.  For a fixed dy the ellipse is dx in (-b dy -+ sqrt(a s - det dy^2)) / a;
    // the left end is convex in dy with its minimum at dy = b X / c (the ellipse's leftmost point, X = sqrt(s c / det)), so over
    // the band it is attained at that dy clamped into the band (and into the ellipse's own y range); the right end mirrors it.
    const float sm = g.s2max * 1.001f + 1e-2f;                    // (the margin of rec_reaches_rect)
    const float rdet = 1.0f / det, ymax = sqrtf(a * sm * rdet), X = sqrtf(sm * c * rdet);
    const float yb0 = fmaxf(Y0, -ymax), yb1 = fminf(Y1, ymax);
    if (!(yb0 <= yb1)) return 0;
    const float ra = 1.0f / a, tilt = b * X / c;
    const float dyl = fminf(fmaxf(tilt, yb0), yb1), dyr = fminf(fmaxf(-tilt, yb0), yb1);
    const float xl = (-b * dyl - sqrtf(fmaxf(a * sm - det * dyl * dyl, 0.f))) * ra - 2e-3f;
    const float xr = (-b * dyr + sqrtf(fmaxf(a * sm - det * dyr * dyr, 0.f))) * ra + 2e-3f;
    // tile column t holds the centres 16 t + 0.5 .. 16 t + 15.5: it meets [mx + xl, mx + xr] iff
    //   t >= (mx + xl - 15.5) / 16  and  t <= (mx + xr - 0.5) / 16
    const float inv = 1.0f / (float)MTGS_TILE_SIZE;
    // (clamped on both sides BEFORE the conversion: an ellipse far larger than the image -- or a NaN, which fmaxf / fminf turn into
    //  the bound, i.e. the full span -- must not reach an out-of-range float-to-int conversion)
    const float fx0 = (float)g.x0, fx1 = (float)(g.x0 + g.w);
    const int lo = (int)fminf(fmaxf(ceilf((g.mx + xl - 15.5f) * inv), fx0), fx1) - g.x0;
    const int hi = (int)fmaxf(fminf(floorf((g.mx + xr - 0.5f) * inv), fx1 - 1.f), fx0 - 1.f) - g.x0 + 1;
    // (the closed form errs on the wide side by its margins; any superset of the exact set is fine -- blend.hip's staging
    //  applies rec_reaches_rect to what is listed)
    first += lo;
    return hi - lo;
}

// Last-block hand-off of the two counting kernels: true in every thread of the workgroup that arrives last among the
// `active` workgroups that have data (the others never touch the counter: returning atomics on ONE address are served
// at ~90 per us, so thousands of empty workgroups of a capacity-sized grid would cost more than the kernel).  No fence:
// what the last workgroup reads was written by agent-scope atomics, which are complete when __syncthreads' vmcnt(0)
// lets their issuers pass; an agent-scope release fence would write back the XCD's L2 for nothing (microseconds each).
__device__ __forceinline__ bool arrive_last(uint32_t *done, int64_t active, int *s_flag) {
    __syncthreads();
    if (threadIdx.x == 0)
        *s_flag = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)(active - 1);
    __syncthreads();
    return *s_flag != 0;
}

constexpr int R_BLOCK = 1024;
// A thread walks the rows of its Gaussian (the tiles of its item) itself up to SERIAL_MAX of them; larger ones -- a
// Gaussian right in front of the camera covers all 68 rows x 120 tiles -- are queued in LDS and shared out over the
// lanes of a wave.  (With a low threshold the queue costs more than it saves -- 8: 49 us against 38 us for the tile
// placement -- because every queued item is re-read from global memory by the wave that expands it.)
constexpr int SERIAL_MAX = 48;

// TIGHT lists: a row costs ~40 instructions (row_span) instead of one LDS atomic, and a wave would wait for its Gaussian with the
// most rows (3.6 on average, up to SERIAL_MAX): the rows of a workgroup's 1024 Gaussians are numbered by a prefix sum and shared
// out evenly over its threads -- thread i takes rows i, i + 1024, ... and finds the owner by bisection in LDS.
struct RowShare {
    float mx[R_BLOCK], my[R_BLOCK], ca[R_BLOCK], cb[R_BLOCK], cc[R_BLOCK], s2[R_BLOCK];
    uint32_t xy[R_BLOCK];        // x0 | y0 << 16: first tile column / row inside the camera's grid
    uint32_t wr[R_BLOCK];        // (w - 1) | row0 << 16: tiles per row of the 3-sigma square, first (camera, row) bin
    uint32_t off[R_BLOCK + 1];   // exclusive prefix sum of the row counts
    uint32_t ws[R_BLOCK / 64];
};
// emit(owner thread, (camera, row) bin, first tile id, number of tiles > 0); h = this thread's rows (0: none, or shared elsewhere)
template <class F>
__device__ __forceinline__ void tight_rows_walk(RowShare &S, const RowGeom &g, int h, int tw, F &&emit) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    S.mx[tid] = g.mx; S.my[tid] = g.my; S.ca[tid] = g.ca; S.cb[tid] = g.cb; S.cc[tid] = g.cc; S.s2[tid] = g.s2max;
    S.xy[tid] = (uint32_t)g.x0 | ((uint32_t)g.y0 << 16);
    S.wr[tid] = (uint32_t)max(g.w - 1, 0) | ((uint32_t)g.row0 << 16);
    uint32_t inc = (uint32_t)h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) S.ws[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += S.ws[w];
    S.off[tid] = base + inc - (uint32_t)h;
    if (tid == R_BLOCK - 1) S.off[R_BLOCK] = base + inc;
    __syncthreads();
    const uint32_t total = S.off[R_BLOCK];
    static_assert(R_BLOCK == 1 << 10, "ten bisection steps");
    for (uint32_t idx = tid; idx < total; idx += R_BLOCK) {
        int lo = 0, hi = R_BLOCK;    // the last owner whose offset is <= idx (owners without rows share their successor's offset)
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            const int mid = (lo + hi) >> 1;
            if (S.off[mid] <= idx) lo = mid; else hi = mid;
        }
        const int y = (int)(idx - S.off[lo]);
        RowGeom q;
        q.mx = S.mx[lo]; q.my = S.my[lo]; q.ca = S.ca[lo]; q.cb = S.cb[lo]; q.cc = S.cc[lo]; q.s2max = S.s2[lo];
        const uint32_t xy = S.xy[lo], wr = S.wr[lo];
        q.x0 = (int)(xy & 0xffffu); q.y0 = (int)(xy >> 16); q.w = (int)(wr & 0xffffu) + 1; q.row0 = (int)(wr >> 16);
        q.tile0 = q.row0 * tw + q.x0; q.h = 0;
        int first;
        const int w = row_span<true>(q, y, tw, first);
        if (w > 0) emit(lo, q.row0 + y, first, w);
    }
}
// counts per (workgroup, row); every pair gets its base inside the row's segment from ONE returning atomic
template <bool TIGHT>
__global__ __launch_bounds__(R_BLOCK) void bin3_rows_count_kernel(const SizeRef n_vis_ref, const float *__restrict__ recs,
                                                                 const int32_t *__restrict__ vis_ids, int64_t N, int C, float ts,
                                                                 int tw, int th, int n_rows, uint32_t *__restrict__ row_count,
                                                                 uint32_t *__restrict__ rbase, uint32_t *__restrict__ done,
                                                                 uint32_t *__restrict__ row_start /* [n_rows + 1] */) {
    __shared__ uint32_t s_row[MAX_ROWS];
    __shared__ uint32_t s_ws[R_BLOCK / 64];
    __shared__ uint16_t s_tall[R_BLOCK];
    __shared__ uint32_t s_ntall;
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_vis = mtgs_os::size_of(n_vis_ref);
    const int64_t r = (int64_t)blockIdx.x * R_BLOCK + tid;
    if ((int64_t)blockIdx.x * R_BLOCK >= n_vis) {
        if (n_vis == 0 && blockIdx.x == 0)
            for (int b = tid; b <= n_rows; b += R_BLOCK) row_start[b] = 0;
        return;
    }
    for (int b = tid; b < n_rows; b += R_BLOCK) s_row[b] = 0;
    if (tid == 0) s_ntall = 0;
    __syncthreads();
    RowGeom g = {};
    int my_rows = 0;   // rows this thread's Gaussian contributes to the shared walk
    if (r < n_vis) {
        g = row_geom<TIGHT>(recs, vis_ids, r, N, C, ts, tw, th);
        if (g.h <= SERIAL_MAX) {
            my_rows = g.h;
            if (!TIGHT)
                for (int y = 0; y < g.h; ++y) atomicAdd(&s_row[g.row0 + y], 1u);
        } else {
            s_tall[atomicAdd(&s_ntall, 1u)] = (uint16_t)tid;   // a wave shares its rows (below)
        }
    }
    if constexpr (TIGHT) {
        __shared__ RowShare s_share;
        tight_rows_walk(s_share, g, my_rows, tw, [&](int, int bin, int, int) { atomicAdd(&s_row[bin], 1u); });
    }
    __syncthreads();
    for (uint32_t q = wave; q < s_ntall; q += R_BLOCK / 64) {
        const RowGeom g = row_geom<TIGHT>(recs, vis_ids, (int64_t)blockIdx.x * R_BLOCK + s_tall[q], N, C, ts, tw, th);
        int first;
        for (int y = lane; y < g.h; y += 64)
            if (row_span<TIGHT>(g, y, tw, first) > 0) atomicAdd(&s_row[g.row0 + y], 1u);
    }
    __syncthreads();
    for (int b = tid; b < n_rows; b += R_BLOCK) {
        const uint32_t c = s_row[b];
        if (c) rbase[(size_t)blockIdx.x * n_rows + b] = __hip_atomic_fetch_add(row_count + b, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!arrive_last(done, ceil_div64(n_vis, R_BLOCK), &s_last)) return;
    // exclusive scan of the row totals (ROWS_PER_THREAD consecutive rows per thread)
    constexpr int ROWS_PER_THREAD = MAX_ROWS / R_BLOCK;
    static_assert(MAX_ROWS % R_BLOCK == 0, "whole rows per thread");
    uint32_t v[ROWS_PER_THREAD], mine = 0;
#pragma unroll
    for (int e = 0; e < ROWS_PER_THREAD; ++e) {
        const int b = tid * ROWS_PER_THREAD + e;
        v[e] = b < n_rows ? mtgs_os::ld32(row_count + b) : 0u;
        mine += v[e];
    }
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_ws[wave] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (int w = 0; w < wave; ++w) run += s_ws[w];
#pragma unroll
    for (int e = 0; e < ROWS_PER_THREAD; ++e) {
        const int b = tid * ROWS_PER_THREAD + e;
        if (b < n_rows) row_start[b] = run;
        run += v[e];
        if (b == n_rows - 1) row_start[n_rows] = run;   // number of items
    }
}

template <bool TIGHT>
__global__ __launch_bounds__(R_BLOCK) void bin3_rows_place_kernel(const SizeRef n_vis_ref, const float *__restrict__ recs,
                                                                 const int32_t *__restrict__ vis_ids, int64_t N, int C, float ts,
                                                                 int tw, int th, int n_rows, const uint32_t *__restrict__ row_start,
                                                                 const uint32_t *__restrict__ rbase, int64_t cap_items,
                                                                 Item *__restrict__ items) {
    __shared__ uint32_t s_row[MAX_ROWS];   // next free slot of this workgroup in each row's segment
    __shared__ uint16_t s_tall[R_BLOCK];
    __shared__ uint32_t s_ntall;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_vis = mtgs_os::size_of(n_vis_ref);
    if ((int64_t)blockIdx.x * R_BLOCK >= n_vis) return;
    // (rows this workgroup has no item in were never given a base: their entry is not read below)
    for (int b = tid; b < n_rows; b += R_BLOCK) s_row[b] = row_start[b] + rbase[(size_t)blockIdx.x * n_rows + b];
    if (tid == 0) s_ntall = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * R_BLOCK + tid;
    RowGeom g = {};
    int my_rows = 0;
    if (r < n_vis) {
        g = row_geom<TIGHT>(recs, vis_ids, r, N, C, ts, tw, th);
        if (g.h <= SERIAL_MAX) {
            my_rows = g.h;
            if (!TIGHT)
                for (int y = 0; y < g.h; ++y) {
                    const int64_t pos = atomicAdd(&s_row[g.row0 + y], 1u);
                    if (pos < cap_items) items[pos] = Item{(uint32_t)r, ((uint32_t)(g.tile0 + y * tw) << SPAN_BITS) | (uint32_t)(g.w - 1)};
                }
        } else {
            s_tall[atomicAdd(&s_ntall, 1u)] = (uint16_t)tid;
        }
    }
    if constexpr (TIGHT) {
        __shared__ RowShare s_share;
        const int64_t r0 = (int64_t)blockIdx.x * R_BLOCK;
        tight_rows_walk(s_share, g, my_rows, tw, [&](int owner, int bin, int first, int w) {
            const int64_t pos = atomicAdd(&s_row[bin], 1u);
            if (pos < cap_items) items[pos] = Item{(uint32_t)(r0 + owner), ((uint32_t)first << SPAN_BITS) | (uint32_t)(w - 1)};
        });
    }
    __syncthreads();
    for (uint32_t q = wave; q < s_ntall; q += R_BLOCK / 64) {
        const int64_t rq = (int64_t)blockIdx.x * R_BLOCK + s_tall[q];
        const RowGeom g = row_geom<TIGHT>(recs, vis_ids, rq, N, C, ts, tw, th);
        for (int y = lane; y < g.h; y += 64) {
            int first;
            const int w = row_span<TIGHT>(g, y, tw, first);
            if (w <= 0) continue;
            const int64_t pos = atomicAdd(&s_row[g.row0 + y], 1u);
            if (pos < cap_items) items[pos] = Item{(uint32_t)rq, ((uint32_t)first << SPAN_BITS) | (uint32_t)(w - 1)};
        }
    }
}

// ---- 2. per-tile counts -> offsets, tile dispatch order ---------------------------------------------------------
// The items are grouped by row, so the tiles a workgroup's 1024 items touch are a short contiguous range of tile ids
// (one or two rows): only that range of the LDS histogram is cleared and flushed.
constexpr int SM_LONG_BUCKET = 255;   // (bin3_sort_small_kernel: lists of length >> 2 >= 255 are sorted by a whole workgroup)
constexpr int T_THREADS = 1024, T_ITEMS = 2, T_TILE = T_THREADS * T_ITEMS, SCHED_BUCKETS = 1024;
struct TileRange { int lo, hi; };
__device__ __forceinline__ TileRange tile_range(const Item *__restrict__ items, int64_t base, int64_t n_items, int tw) {
    const int64_t last = min(n_items, base + T_TILE) - 1;
    const int first_tile = (int)(items[base].span >> SPAN_BITS), last_tile = (int)(items[last].span >> SPAN_BITS);
    return TileRange{(first_tile / tw) * tw, (last_tile / tw + 1) * tw};
}

__global__ __launch_bounds__(T_THREADS) void bin3_tiles_count_kernel(
    const uint32_t *__restrict__ n_items_ptr, int64_t cap_items, const Item *__restrict__ items, int tw, int n_bins,
    int64_t cap_M, uint32_t *__restrict__ bins /* [n_bins], zero */, uint32_t *__restrict__ done /* zero */,
    int32_t *__restrict__ offsets /* [n_bins + 1] */, int32_t *__restrict__ order /* [n_bins] */,
    uint32_t *__restrict__ n_long /* number of lists of at least SM_LONG keys: the first n_long of the order */) {
    extern __shared__ uint32_t s_bins[];  // [n_bins]
    __shared__ uint32_t s_aux[SCHED_BUCKETS];
    __shared__ uint32_t s_ws[T_THREADS / 64];
    __shared__ uint16_t s_wide[T_TILE];
    __shared__ uint32_t s_nwide;
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_items = min((int64_t)*n_items_ptr, cap_items);
    if ((int64_t)blockIdx.x * T_TILE >= n_items) {
        if (n_items == 0 && blockIdx.x == 0) {   // nothing visible: empty lists, any order
            for (int b = tid; b <= n_bins; b += T_THREADS) offsets[b] = 0;
            for (int b = tid; b < n_bins; b += T_THREADS) order[b] = b;
            if (tid == 0) *n_long = 0;
        }
        return;
    }
    // (the grid is a few workgroups per CU, not one per chunk: a capacity-sized grid of 1024-thread workgroups that
    // find nothing to do costs more to dispatch than the kernel takes)
    for (int64_t base = (int64_t)blockIdx.x * T_TILE; base < n_items; base += (int64_t)gridDim.x * T_TILE) {
        const TileRange tr = tile_range(items, base, n_items, tw);
        for (int b = tr.lo + tid; b < tr.hi; b += T_THREADS) s_bins[b] = 0;
        if (tid == 0) s_nwide = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < T_ITEMS; ++e) {
            const int64_t i = base + e * T_THREADS + tid;
            if (i < n_items) {
                const Item it = items[i];
                const int t0 = (int)(it.span >> SPAN_BITS), w = (int)(it.span & ((1u << SPAN_BITS) - 1)) + 1;
                if (w <= SERIAL_MAX) {
                    for (int x = 0; x < w; ++x) atomicAdd(&s_bins[t0 + x], 1u);
                } else {
                    s_wide[atomicAdd(&s_nwide, 1u)] = (uint16_t)(e * T_THREADS + tid);
                }
            }
        }
        __syncthreads();
        for (uint32_t q = wave; q < s_nwide; q += T_THREADS / 64) {
            const Item it = items[base + s_wide[q]];
            const int t0 = (int)(it.span >> SPAN_BITS), w = (int)(it.span & ((1u << SPAN_BITS) - 1)) + 1;
            for (int x = lane; x < w; x += 64) atomicAdd(&s_bins[t0 + x], 1u);
        }
        __syncthreads();
        for (int b = tr.lo + tid; b < tr.hi; b += T_THREADS) {
            const uint32_t c = s_bins[b];
            if (c) {
                // the RETURNING form: its completion is what the vmcnt(0) of the barrier below waits for, and the last
                // workgroup's reads of the totals (arrive_last) rely on exactly that
                const uint32_t before = __hip_atomic_fetch_add(bins + b, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("" ::"v"(before));
            }
        }
        __syncthreads();
    }
    if (!arrive_last(done, min((int64_t)gridDim.x, ceil_div64(n_items, T_TILE)), &s_last)) return;
    for (int b = tid; b < n_bins; b += T_THREADS) s_bins[b] = mtgs_os::ld32(bins + b);
    __syncthreads();
    // exclusive scan of the counts -> offsets, clamped to the capacity of the key / id arrays: a frame beyond its
    // capacities is repeated by the caller, but every kernel behind this one must stay inside the buffers
    const int per = (n_bins + T_THREADS - 1) / T_THREADS;   // consecutive bins per thread
    const int b0 = tid * per, b1 = min(n_bins, b0 + per);
    uint32_t mine = 0;
    for (int b = b0; b < b1; ++b) mine += s_bins[b];
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_ws[wave] = inc;
    __syncthreads();
    uint32_t wb = 0;
    for (int w = 0; w < wave; ++w) wb += s_ws[w];
    uint64_t run = (uint64_t)wb + inc - mine;
    const uint64_t cap = (uint64_t)cap_M;
    for (int b = b0; b < b1; ++b) {
        const uint64_t lo = run < cap ? run : cap;
        run += s_bins[b];
        const uint64_t hi = run < cap ? run : cap;
        offsets[b] = (int32_t)lo;
        s_bins[b] = (uint32_t)(hi - lo);   // the (clamped) list length: what the dispatch order is built from
        if (b == n_bins - 1) offsets[n_bins] = (int32_t)hi;
    }
    // tile dispatch order: counting sort by decreasing list length (bucket width 4), as blend.hip::tile_schedule_kernel
    s_aux[tid] = 0;
    __syncthreads();
    auto bucket_of = [&](int t) { return SCHED_BUCKETS - 1 - min((int)(s_bins[t] >> 2), SCHED_BUCKETS - 1); };
    for (int t = tid; t < n_bins; t += T_THREADS) atomicAdd(&s_aux[bucket_of(t)], 1u);
    __syncthreads();
    const uint32_t hv = s_aux[tid];
    uint32_t hinc = hv;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(hinc, o, 64);
        if (lane >= o) hinc += up;
    }
    if (lane == 63) s_ws[wave] = hinc;
    __syncthreads();
    uint32_t hb = 0;
    for (int w = 0; w < wave; ++w) hb += s_ws[w];
    __syncthreads();
    s_aux[tid] = hb + hinc - hv;
    __syncthreads();
    if (tid == 0) *n_long = s_aux[SCHED_BUCKETS - SM_LONG_BUCKET];   // lists in the buckets in front of it: length >> 2 >= 255
    __syncthreads();
    // (ties inside a bucket land in arrival order: a scheduling aid, results do not depend on it)
    for (int t = tid; t < n_bins; t += T_THREADS) order[atomicAdd(&s_aux[bucket_of(t)], 1u)] = t;
}

struct TailFill {
    int mode;                // 0: none; 2: up to min(cap, M of `totals`); 4: up to the capacity
    const int64_t *totals;   // device: n_vis << 32 | M
    int32_t *flatten_ids;
    int64_t *isect_ids;      // nullable
    int64_t sentinel_key;
    int64_t *status;         // nullable (MTGS_BIN3_STATUS): {n_vis, M, frame beyond its capacities} for the caller, behind `totals`
    int64_t cap_vis, cap_M;
};

// ---- 3. every intersection into its tile's segment --------------------------------------------------------------
// A workgroup's intersections of one tile take CONSECUTIVE slots of the tile's segment: one returning global atomic per
// (workgroup, tile) reserves them, LDS atomics hand them out -- ~130k global atomics per frame instead of one per
// intersection, and the segment is written in runs of ~60 keys instead of single 8-byte stores.
__global__ __launch_bounds__(T_THREADS) void bin3_tiles_place_kernel(
    const uint32_t *__restrict__ n_items_ptr, int64_t cap_items, const Item *__restrict__ items, int tw, int n_bins,
    const int32_t *__restrict__ offsets, uint32_t *__restrict__ cursor /* [n_bins], zero */,
    const uint64_t *__restrict__ vis_keys, uint32_t cap_keys, uint64_t *__restrict__ keys64, const TailFill tail) {
    // (In a frame beyond its capacities the offsets are clamped and a tile's keys may spill into its neighbour's slots or
    // past cap_keys: the stores stay inside the buffer, slots may stay unwritten, and the sort's epilogue clamps the
    // rank it reads from them -- the caller repeats such a frame.)
    extern __shared__ uint32_t s_bins[];   // [n_bins]: count, then next free slot, of the tiles this chunk touches
    __shared__ uint16_t s_wide[T_TILE];
    __shared__ uint32_t s_nwide;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_items = min((int64_t)*n_items_ptr, cap_items);
    for (int64_t base = (int64_t)blockIdx.x * T_TILE; base < n_items; base += (int64_t)gridDim.x * T_TILE) {
        const TileRange tr = tile_range(items, base, n_items, tw);
        for (int b = tr.lo + tid; b < tr.hi; b += T_THREADS) s_bins[b] = 0;
        if (tid == 0) s_nwide = 0;
        __syncthreads();
        Item it[T_ITEMS];
        int t0[T_ITEMS], w[T_ITEMS];
        uint32_t depth[T_ITEMS];
#pragma unroll
        for (int e = 0; e < T_ITEMS; ++e) {
            const int64_t i = base + e * T_THREADS + tid;
            it[e] = Item{0, 0}; t0[e] = 0; w[e] = 0; depth[e] = 0;
            if (i < n_items) {
                it[e] = items[i];
                t0[e] = (int)(it[e].span >> SPAN_BITS); w[e] = (int)(it[e].span & ((1u << SPAN_BITS) - 1)) + 1;
                depth[e] = (uint32_t)vis_keys[it[e].rank];
                if (w[e] <= SERIAL_MAX) {
                    for (int x = 0; x < w[e]; ++x) atomicAdd(&s_bins[t0[e] + x], 1u);
                } else {
                    s_wide[atomicAdd(&s_nwide, 1u)] = (uint16_t)(e * T_THREADS + tid);
                    w[e] = 0;   // a wave shares its tiles (below)
                }
            }
        }
        __syncthreads();
        const uint32_t n_wide = s_nwide;
        for (uint32_t q = wave; q < n_wide; q += T_THREADS / 64) {
            const Item iw = items[base + s_wide[q]];
            const int tq = (int)(iw.span >> SPAN_BITS), wq = (int)(iw.span & ((1u << SPAN_BITS) - 1)) + 1;
            for (int x = lane; x < wq; x += 64) atomicAdd(&s_bins[tq + x], 1u);
        }
        __syncthreads();
        for (int b = tr.lo + tid; b < tr.hi; b += T_THREADS) {
            const uint32_t c = s_bins[b];
            if (c) s_bins[b] = (uint32_t)offsets[b] + __hip_atomic_fetch_add(cursor + b, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < T_ITEMS; ++e) {
            const uint64_t key = ((uint64_t)depth[e] << 32) | it[e].rank;
            for (int x = 0; x < w[e]; ++x) {
                const uint32_t pos = atomicAdd(&s_bins[t0[e] + x], 1u);
                if (pos < cap_keys) keys64[pos] = key;
            }
        }
        for (uint32_t q = wave; q < n_wide; q += T_THREADS / 64) {
            const Item iw = items[base + s_wide[q]];
            const int tq = (int)(iw.span >> SPAN_BITS), wq = (int)(iw.span & ((1u << SPAN_BITS) - 1)) + 1;
            const uint64_t key = ((uint64_t)(uint32_t)vis_keys[iw.rank] << 32) | iw.rank;
            for (int x = lane; x < wq; x += 64) {
                const uint32_t pos = atomicAdd(&s_bins[tq + x], 1u);
                if (pos < cap_keys) keys64[pos] = key;
            }
        }
        __syncthreads();
    }
    // The caller-visible tail of flatten_ids / isect_ids behind the listed pairs (tight lists: [n_listed, M); graph mode:
    // up to the capacity) is filled with SENTINELS -- flatten_ids -1, isect_ids = last camera | last tile | +inf depth -- so that
    // gsplat's convention "the last tile's range ends at flatten_ids.numel()" never walks uninitialised entries: the
    // gather-based compositing forward stops at the first negative id, isect_offset_encode of the padded isect_ids puts the
    // tail into the last tile.  offsets[n_bins] is final here (bin3_tiles_count_kernel); the sort kernels behind write [0, n_listed).
    if (tail.status && blockIdx.x == 0 && tid == 0) {
        const int64_t packed = *tail.totals, nv = packed >> 32, m = packed & 0xFFFFFFFFll;
        tail.status[0] = nv;
        tail.status[1] = m;
        tail.status[2] = (nv > tail.cap_vis || m > tail.cap_M) ? 1 : 0;
    }
    if (tail.mode) {
        const int64_t from = offsets[n_bins];
        int64_t to = (int64_t)cap_keys;
        if (tail.mode == 2) to = min(to, (int64_t)(*tail.totals & 0xFFFFFFFFll));
        for (int64_t i = from + (int64_t)blockIdx.x * T_THREADS + tid; i < to; i += (int64_t)gridDim.x * T_THREADS) {
            tail.flatten_ids[i] = -1;
            if (tail.isect_ids) tail.isect_ids[i] = tail.sentinel_key;
        }
    }
}

// ---- 4. per-tile sort in LDS ------------------------------------------------------------------------------------
struct SortEpilogue {
    int32_t *rank_ids, *flatten_ids;
    int64_t *isect_ids;  // nullable
    const int32_t *vis_ids;
    uint32_t n_tiles;
    int tile_bits;
    bool single_cam;
    uint32_t rank_max;   // cap_vis - 1: a slot a truncated frame left unwritten must not index outside the records
    __device__ __forceinline__ void store(int64_t dst, uint32_t bin, uint64_t key) const {
        const int32_t rank = (int32_t)min((uint32_t)key, rank_max);
        rank_ids[dst] = rank;
        // (NON-TEMPORAL: gsplat's two index tensors are outputs nobody in the frame reads -- the compositing kernels walk rank_ids --
        //  and 48 MB of them would evict the keys and records the kernels behind re-read)
        __builtin_nontemporal_store(vis_ids[rank], flatten_ids + dst);
        if (isect_ids) {
            const int64_t cam = single_cam ? 0 : bin / n_tiles, tile = single_cam ? bin : bin % n_tiles;
            __builtin_nontemporal_store((long long)((cam << (32 + tile_bits)) | (tile << 32) | (int64_t)(key >> 32)), reinterpret_cast<long long *>(isect_ids) + dst);
        }
    }
};

// The keys are compared as DOUBLES: depth bits of a positive finite float in the high word make the 64-bit pattern a
// positive finite double whose order is the integer order, and v_min_f64 / v_max_f64 are full-rate instructions -- a
// compare-exchange is two VALU instructions instead of a 64-bit compare and four selects (it was 7-8 issue slots).
// The network is the all-ascending form of the bitonic sort (first stage of a level mirrors, i <-> i ^ (k - 1), the
// others are plain butterflies), so no comparator needs a direction, and padding keys (+inf) never move down: work
// items that hold only padding are skipped.
constexpr uint64_t KEY_INF = 0x7ff0000000000000ull;
// LDS layout: key i lives at i ^ T(bits 5..8 of i), T linear over GF(2) with columns (15, 11, 29, 17).  ds_read_b64
// serves 32 lanes per cycle from 64 four-byte banks, i.e. it is conflict-free when the 32 keys fall into 32 different
// 8-byte columns of the 256-byte row.  The 32 lanes of a group hold work items whose keys differ in five index bits --
// the lowest five that are not stage strides of the trip: with 8 keys per thread {3..7}, {0,4..7}, {0,1,5,6,7},
// {0,1,2,6,7}, {0,1,2,3,7} or {0..4}, with 16 keys {4..8}, {0,5..8}, {0,1,6,7,8}, {0,1,2,7,8}, {0,1,2,3,8} or {0..4} --
// and with these columns each of those sets maps onto the five column bits bijectively (a padding of one key per 32
// left 45 % of the LDS cycles of the sort to bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
__host__ __device__ constexpr int swz_t(int hi4) {
    return ((hi4 & 1) ? 15 : 0) ^ ((hi4 & 2) ? 11 : 0) ^ ((hi4 & 4) ? 29 : 0) ^ ((hi4 & 8) ? 17 : 0);
}
__device__ __forceinline__ int swz(int i) {
    const int t = (int)((0x1916121d040b0f00ull >> ((i >> 2) & 0x38)) & 31);   // bytes = swz_t(0..7)
    return i ^ t ^ ((i & 256) ? 17 : 0);
}
static_assert(swz_t(1) == 0x0f && swz_t(2) == 0x0b && swz_t(3) == 0x04 && swz_t(4) == 0x1d && swz_t(5) == 0x12 &&
              swz_t(6) == 0x16 && swz_t(7) == 0x19 && swz_t(8) == 17, "packed table");
// swz(x ^ d) = swz(x) ^ swz_d(d) for any d (T is linear); the stage strides are compile-time, so a thread's keys are
// one swizzled base and XOR constants
__host__ __device__ constexpr int swz_d(int d) { return d ^ swz_t((d >> 5) & 15); }
__device__ __forceinline__ void cswap(uint64_t &a, uint64_t &b) {   // a <- min, b <- max
    const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b);
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    a = (uint64_t)__double_as_longlong(lo); b = (uint64_t)__double_as_longlong(hi);
}
template <bool WAVE> __device__ __forceinline__ void lds_sync() {
    if constexpr (WAVE) {
        // one wave: its LDS instructions execute in program order, only the compiler must not move them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// LOGE consecutive butterfly stages in ONE LDS round trip: a thread owns the E = 2^LOGE keys x + e * SLOW whose indices
// differ in the stage strides STOP, STOP/2 ... SLOW (compile-time, x has zeros at those bits); `apply` = which of the
// stages run (bit LOGE-1: STOP ...).  n = number of real keys (the rest of [0, P) is padding).
template <bool WAVE, int LOGE, int STOP>
__device__ __forceinline__ void trip(uint64_t *s, int groups, int n, int nthr, int tid, int apply) {
    constexpr int E = 1 << LOGE, SLOW = STOP >> (LOGE - 1);
    if constexpr (SLOW >= 1) {
        for (int g = tid; g < groups; g += nthr) {
            const int x = ((g & ~(SLOW - 1)) << LOGE) | (g & (SLOW - 1));
            if (x >= n) continue;
            const int px = swz(x);
            uint64_t v[E];
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = s[px ^ swz_d(e * SLOW)];
#pragma unroll
            for (int bit = E >> 1; bit >= 1; bit >>= 1)
                if (apply & bit) {
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (!(e & bit)) cswap(v[e], v[e | bit]);
                }
#pragma unroll
            for (int e = 0; e < E; ++e) s[px ^ swz_d(e * SLOW)] = v[e];
        }
        lds_sync<WAVE>();
    }
}
// The first LOGE stages of level k = 2 H: mirror (i <-> i ^ (k - 1)), then the butterflies H/2 ... B = H >> (LOGE-1).
// A thread owns E/2 keys of the lower half of a k-block (x + f B) and their mirror images in the upper half.
template <bool WAVE, int LOGE, int H>
__device__ __forceinline__ void mirror_trip(uint64_t *s, int groups, int n, int nthr, int tid) {
    constexpr int E = 1 << LOGE, HALF = E >> 1, B = H >> (LOGE - 1);
    if constexpr (B >= 1) {
        for (int g = tid; g < groups; g += nthr) {
            const int x = ((g & ~(B - 1)) << LOGE) | (g & (B - 1));
            if (x >= n) continue;
            const int low = x & (B - 1);
            const int pl = swz(x), pu = swz(x - low + H + (B - 1 - low));
            uint64_t v[E];
#pragma unroll
            for (int f = 0; f < HALF; ++f) { v[f] = s[pl ^ swz_d(f * B)]; v[HALF + f] = s[pu ^ swz_d(f * B)]; }
#pragma unroll
            for (int f = 0; f < HALF; ++f) cswap(v[f], v[E - 1 - f]);
#pragma unroll
            for (int bit = HALF >> 1; bit >= 1; bit >>= 1) {
#pragma unroll
                for (int f = 0; f < HALF; ++f)
                    if (!(f & bit)) { cswap(v[f], v[f | bit]); cswap(v[HALF + f], v[HALF + (f | bit)]); }
            }
#pragma unroll
            for (int f = 0; f < HALF; ++f) { s[pl ^ swz_d(f * B)] = v[f]; s[pu ^ swz_d(f * B)] = v[HALF + f]; }
        }
        lds_sync<WAVE>();
    }
}
#define B3_STRIDE_SWITCH(J, CALL)                                                                                   \
    switch (J) {                                                                                                     \
        case 4: CALL(4); break; case 8: CALL(8); break; case 16: CALL(16); break; case 32: CALL(32); break;          \
        case 64: CALL(64); break; case 128: CALL(128); break; case 256: CALL(256); break; case 512: CALL(512); break; \
        case 1024: CALL(1024); break; case 2048: CALL(2048); break; case 4096: CALL(4096); break;                   \
        default: CALL(8192); break;                                                                                   \
    }

// the butterfly stages from stride j down to 1 (j a power of two, 1 <= j <= 8192; groups = P >> LOGE)
template <bool WAVE, int LOGE>
__device__ __forceinline__ void level_tail(uint64_t *s, int groups, int n, int nthr, int tid, int j) {
    constexpr int E = 1 << LOGE;
    int r = 0;
    for (int q = j; q >= 1; q >>= 1) ++r;       // stages left
    while (r > LOGE) {                           // from the top, LOGE at a time, down to stride E
        const int take = r - LOGE >= LOGE ? LOGE : r - LOGE;
        const int apply = ((1 << take) - 1) << (LOGE - take);
#define B3_CALL(S) trip<WAVE, LOGE, S>(s, groups, n, nthr, tid, apply)
        B3_STRIDE_SWITCH(j, B3_CALL)
#undef B3_CALL
        j >>= take;
        r -= take;
    }
    trip<WAVE, LOGE, E / 2>(s, groups, n, nthr, tid, (1 << r) - 1);   // strides E/2 ... 1 (the last r of them)
}

// ascending sort of s[0, P) (P a power of two >= E, key i at swz(i), [n, P) holds KEY_INF); callers synchronise before
template <bool WAVE, int LOGE>
__device__ __forceinline__ void bitonic_sort_e(uint64_t *s, int P, int n, int nthr, int tid) {
    constexpr int E = 1 << LOGE;
    const int groups = P >> LOGE;
    for (int g = tid; g < groups; g += nthr) {   // levels 2 ... E on E contiguous keys, in registers
        const int x = g << LOGE;
        if (x >= n) continue;
        const int px = swz(x);
        uint64_t v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = s[px ^ e];
#pragma unroll
        for (int k = 2; k <= E; k <<= 1) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((e ^ (k - 1)) > e) cswap(v[e], v[e ^ (k - 1)]);
#pragma unroll
            for (int j = k >> 2; j >= 1; j >>= 1) {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (!(e & j)) cswap(v[e], v[e | j]);
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) s[px ^ e] = v[e];
    }
    lds_sync<WAVE>();
    for (int k = 2 * E; k <= P; k <<= 1) {
        const int h = k >> 1;
#define B3_CALL(S) mirror_trip<WAVE, LOGE, S>(s, groups, n, nthr, tid)
        B3_STRIDE_SWITCH(h, B3_CALL)
#undef B3_CALL
        if ((k >> (LOGE + 1)) >= 1) level_tail<WAVE, LOGE>(s, groups, n, nthr, tid, k >> (LOGE + 1));
    }
}
// One wave sorting 513..1024 keys holds them 16 per lane, and four waves sorting 2049..4096 keys 16 per thread (four
// stages per LDS round trip, one pass per trip instead of two: 60 -> 55 us at the headline size, at 128 VGPRs); with
// more threads than work items 8 per thread is the better split (16 keys per thread everywhere: 186 VGPRs in the
// one-wave kernel, spills in the 1024-thread one, 57 -> 90 us).
template <bool WAVE, int P16 = 1 << 30>   // P16: from this padded length on, 16 keys per thread (never: the 1024-thread kernel)
__device__ __forceinline__ void bitonic_sort(uint64_t *s, int P, int n, int nthr, int tid) {
    if constexpr (P16 < (1 << 30)) {
        if (P >= P16) { bitonic_sort_e<WAVE, 4>(s, P, n, nthr, tid); return; }
    }
    bitonic_sort_e<WAVE, 3>(s, P, n, nthr, tid);
}

__device__ __forceinline__ int pow2_ceil(int n) {
    int p = 8;
    while (p < n) p <<= 1;
    return p;
}

// one tile whose list fits the LDS buffer: load, sort, write the three output arrays
template <bool WAVE, int P16 = 1 << 30>
__device__ __forceinline__ void sort_tile_lds(uint64_t *s, const uint64_t *__restrict__ keys, int64_t o0, int L, uint32_t bin,
                                              int nthr, int tid, const SortEpilogue &epi) {
    const int P = pow2_ceil(L);
    for (int i = tid; i < P; i += nthr) s[swz(i)] = i < L ? keys[o0 + i] : KEY_INF;
    lds_sync<WAVE>();
    bitonic_sort<WAVE, P16>(s, P, L, nthr, tid);
    for (int i = tid; i < L; i += nthr) epi.store(o0 + i, bin, s[swz(i)]);
    lds_sync<WAVE>();
}

constexpr int SM_CAP = 4096, SM_WAVE_CAP = 1024, SM_TILES = 4;   // SM_CAP = SM_TILES * SM_WAVE_CAP keys of LDS
constexpr int LG_THREADS = 1024, LG_CAP = 16384;
// Position g of the dispatch order (longest lists first) is sorted by workgroup g when its list is LONG (at least
// 4 * SM_LONG_BUCKET = 1020 keys: the four waves work together, up to SM_CAP keys; beyond that the list is bin3_sort_large_kernel's), and by
// ONE wave otherwise: the short lists follow the n_long long ones in the order, four to a workgroup, no workgroup
// barrier at all.  n_long comes from the kernel that built the order.  (Four long lists per workgroup, one after the
// other, left the chip a quarter full at MTGS's 960x540, where most lists are long: 69 -> 40 us there.)
__global__ __launch_bounds__(B3_BLOCK, 4) void bin3_sort_small_kernel(const int32_t *__restrict__ offsets,
                                                                  const int32_t *__restrict__ order, int n_bins,
                                                                  const uint32_t *__restrict__ n_long_ptr, int64_t cap_M,
                                                                  const uint64_t *__restrict__ keys, const SortEpilogue epi) {
    __shared__ uint64_t s_keys[SM_CAP];
    static_assert(SM_CAP == SM_TILES * SM_WAVE_CAP, "one buffer: a long list, or four short ones");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_long = (int)*n_long_ptr;
    const bool together = (int)blockIdx.x < n_long;
    const int g = together ? (int)blockIdx.x : n_long + ((int)blockIdx.x - n_long) * SM_TILES + wave;
    if (g >= n_bins) return;
    const int bin = order[g];
    const int o0 = offsets[bin], L = offsets[bin + 1] - o0;
    if (L == 0 || L > SM_CAP || (int64_t)o0 + L > cap_M) return;   // nothing / the large kernel's / a frame beyond its capacities
    // (16 keys per thread exactly where 8 would need two passes per trip: 64 lanes x 1024 keys, 256 threads x 4096 keys)
    if (together) sort_tile_lds<false, 4096>(s_keys, keys, o0, L, (uint32_t)bin, B3_BLOCK, tid, epi);
    else if (L <= SM_WAVE_CAP) sort_tile_lds<true, 1024>(s_keys + wave * SM_WAVE_CAP, keys, o0, L, (uint32_t)bin, 64, lane, epi);
}

// Lists longer than SM_CAP (4096 keys): one 1024-thread workgroup per list, up to LG_CAP keys in LDS; beyond that the list is
// sorted in LG_CAP-key chunks and the chunks are merged with the upper levels of the all-ascending form of the
// network (first stage of a level mirrors, i <-> i ^ (k - 1), the others are plain butterflies), whose comparators
// never move a key upwards past the end of the list -- so the list needs no padding in global memory.
__global__ __launch_bounds__(LG_THREADS) void bin3_sort_large_kernel(const int32_t *__restrict__ offsets,
                                                                    const int32_t *__restrict__ order, int n_bins, int64_t cap_M,
                                                                    uint64_t *keys, const SortEpilogue epi) {
    extern __shared__ uint64_t s_big[];   // [LG_CAP]
    const int tid = threadIdx.x;
    for (int g = blockIdx.x; g < n_bins; g += gridDim.x) {
        const int bin = order[g];
        const int64_t o0 = offsets[bin];
        const int L = offsets[bin + 1] - (int)o0;
        if (L <= SM_CAP) {
            // the order is by decreasing min(length >> 2, SCHED_BUCKETS - 1): behind a list outside the first bucket
            // nothing longer follows (inside it, lists of 4092 keys and more are mixed)
            if ((L >> 2) < SCHED_BUCKETS - 1) break;
            continue;
        }
        if (o0 + L > cap_M) continue;     // a frame beyond its capacities (it is repeated)
        if (L <= LG_CAP) {
            sort_tile_lds<false>(s_big, keys, o0, L, (uint32_t)bin, LG_THREADS, tid, epi);
            continue;
        }
        uint64_t *seg = keys + o0;
        const int nch = (L + LG_CAP - 1) / LG_CAP;
        for (int c = 0; c < nch; ++c) {   // every chunk ascending
            const int c0 = c * LG_CAP, n = min(LG_CAP, L - c0);
            for (int i = tid; i < LG_CAP; i += LG_THREADS) s_big[swz(i)] = i < n ? seg[c0 + i] : KEY_INF;
            __syncthreads();
            bitonic_sort<false>(s_big, LG_CAP, n, LG_THREADS, tid);
            for (int i = tid; i < n; i += LG_THREADS) seg[c0 + i] = s_big[swz(i)];
            __syncthreads();
        }
        int64_t P = LG_CAP;
        while (P < L) P <<= 1;
        for (int64_t k = 2 * LG_CAP; k <= P; k <<= 1) {
            const int64_t half = k >> 1;
            for (int64_t i = tid; i < P / 2; i += LG_THREADS) {   // mirror stage
                const int64_t b = (i / half) * k, o = i % half, lo = b + o, hi = b + k - 1 - o;
                if (hi < L) {
                    uint64_t a = seg[lo], z = seg[hi];
                    if (a > z) { seg[lo] = z; seg[hi] = a; }
                }
            }
            __syncthreads();
            for (int64_t j = k >> 2; j >= LG_CAP; j >>= 1) {       // butterflies across chunks
                for (int64_t i = tid; i < P / 2; i += LG_THREADS) {
                    const int64_t lo = (i / j) * 2 * j + i % j, hi = lo + j;
                    if (hi < L) {
                        uint64_t a = seg[lo], z = seg[hi];
                        if (a > z) { seg[lo] = z; seg[hi] = a; }
                    }
                }
                __syncthreads();
            }
            for (int c = 0; c < nch; ++c) {                       // the remaining stages inside each chunk, in LDS
                const int c0 = c * LG_CAP, n = min(LG_CAP, L - c0);
                for (int i = tid; i < LG_CAP; i += LG_THREADS) s_big[swz(i)] = i < n ? seg[c0 + i] : KEY_INF;
                __syncthreads();
                level_tail<false, 3>(s_big, LG_CAP >> 3, n, LG_THREADS, tid, LG_CAP >> 1);
                for (int i = tid; i < n; i += LG_THREADS) seg[c0 + i] = s_big[swz(i)];
                __syncthreads();
            }
        }
        for (int i = tid; i < L; i += LG_THREADS) epi.store(o0 + i, (uint32_t)bin, seg[i]);
        __syncthreads();
    }
}

inline int bit_length_u32(uint32_t v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

struct Bin3Workspace {
    char *control;          // zeroed region
    size_t control_bytes;
    uint32_t *done_rows, *done_tiles, *n_long, *row_count, *bins, *cursor;
    uint32_t *row_start, *rbase;
    int32_t *order;
    Item *items;
    uint64_t *keys64;
    size_t total;
};
inline Bin3Workspace carve3(char *base, int64_t cap_vis, int64_t cap_M, int n_rows, int n_bins) {
    Bin3Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += mtgs_os::align256(bytes); return p; };
    const int64_t m = cap_M > 0 ? cap_M : 1, nv = cap_vis > 0 ? cap_vis : 1;
    w.control = base;
    char *misc = take(64);
    w.done_rows = (uint32_t *)misc; w.done_tiles = (uint32_t *)misc + 1; w.n_long = (uint32_t *)misc + 2;
    w.row_count = (uint32_t *)take((size_t)n_rows * 4);
    w.bins = (uint32_t *)take((size_t)n_bins * 4);
    w.cursor = (uint32_t *)take((size_t)n_bins * 4);
    w.control_bytes = off;
    w.row_start = (uint32_t *)take((size_t)(n_rows + 1) * 4);
    w.rbase = (uint32_t *)take((size_t)ceil_div64(nv, R_BLOCK) * n_rows * 4);
    w.order = (int32_t *)take((size_t)n_bins * 4);
    w.items = (Item *)take((size_t)m * sizeof(Item));
    w.keys64 = (uint64_t *)take((size_t)m * 8);
    w.total = off;
    return w;
}

}  // namespace

extern "C" int mtgs_bin3_supported(int C, int tile_w, int tile_h, int64_t cap_M) {
    return C > 0 && tile_w > 0 && tile_h > 0 && (int64_t)C * tile_w * tile_h <= MAX_BINS && (int64_t)C * tile_h <= MAX_ROWS && tile_w <= (1 << SPAN_BITS) &&
                   cap_M < ((int64_t)1 << 30)
               ? 1 : 0;
}

extern "C" int mtgs_bin3_control_bytes(int C, int tile_w, int tile_h, size_t *bytes) {
    MTGS_REQUIRE(C > 0 && tile_w > 0 && tile_h > 0 && bytes, MTGS_EINVAL, "mtgs_bin3_control_bytes: bad arguments");
    *bytes = carve3(nullptr, 1, 1, C * tile_h, C * tile_w * tile_h).control_bytes;
    return MTGS_OK;
}

extern "C" int mtgs_bin3_workspace_bytes(int C, int tile_w, int tile_h, int64_t cap_vis, int64_t cap_M, size_t *bytes) {
    MTGS_REQUIRE(C > 0 && tile_w > 0 && tile_h > 0 && cap_vis >= 0 && cap_M >= 0 && bytes, MTGS_EINVAL,
                 "mtgs_bin3_workspace_bytes: bad arguments");
    *bytes = carve3(nullptr, cap_vis, cap_M, C * tile_h, C * tile_w * tile_h).total;
    return MTGS_OK;
}

extern "C" int mtgs_bin3_build(int C, int64_t N, int tile_size, int tile_w, int tile_h, int64_t *totals,
                               int64_t cap_vis, int64_t cap_M, const float *recs, const int32_t *vis_ids,
                               const int64_t *vis_keys, int32_t *rank_ids,
                               int32_t *flatten_ids, int64_t *isect_ids, int32_t *offsets, int32_t *tile_order, int flags,
                               void *ws, size_t ws_bytes, void *stream) {
    const int tight = flags & MTGS_BIN3_TIGHT;
    MTGS_REQUIRE((flags & ~(MTGS_BIN3_TIGHT | MTGS_BIN3_FILL_TO_M | MTGS_BIN3_FILL_TO_CAP | MTGS_BIN3_PREZEROED | MTGS_BIN3_STATUS)) == 0, MTGS_EINVAL,
                 "mtgs_bin3_build: unknown flags %d", flags);
    MTGS_REQUIRE(C > 0 && N >= 0 && tile_w > 0 && tile_h > 0 && cap_vis >= 0 && cap_M >= 0, MTGS_EINVAL, "mtgs_bin3_build: bad sizes");
    MTGS_REQUIRE(tile_size == MTGS_TILE_SIZE, MTGS_EUNSUPPORTED, "mtgs_bin3_build: tile_size=%d (only 16 is implemented)", tile_size);
    MTGS_REQUIRE(mtgs_bin3_supported(C, tile_w, tile_h, cap_M), MTGS_EUNSUPPORTED,
                 "mtgs_bin3_build: %d x %d x %d (camera, tile) pairs / %lld intersections (at most %d pairs, %d (camera, tile row) "
                 "pairs and 2^30 intersections; use mtgs_bin_build)", C, tile_w, tile_h, (long long)cap_M, MAX_BINS, MAX_ROWS);
    MTGS_REQUIRE(totals && recs && vis_ids && vis_keys && rank_ids && flatten_ids && offsets && ws, MTGS_EINVAL,
                 "mtgs_bin3_build: null pointer");
    const int n_bins = C * tile_w * tile_h, n_rows = C * tile_h;
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, MTGS_EINVAL, "mtgs_bin3_build: workspace must be 256-byte aligned");
    Bin3Workspace w = carve3((char *)ws, cap_vis, cap_M, n_rows, n_bins);
    MTGS_REQUIRE(ws_bytes >= w.total, MTGS_EWORKSPACE, "mtgs_bin3_build: workspace %zu < %zu bytes", ws_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    // (MTGS_BIN3_PREZEROED: the first mtgs_bin3_control_bytes() of the workspace were cleared by a kernel in front of this call on
    //  the same stream -- mtgs_front_fwd(also_zero) -- one launch fewer per frame)
    if (!(flags & MTGS_BIN3_PREZEROED))
        if (int rc = mtgs_zero_async(w.control, w.control_bytes, st)) return rc;
    const SizeRef n_vis_ref{totals, 1, cap_vis};
    int32_t *order = tile_order ? tile_order : w.order;
    const unsigned r_grid = (unsigned)ceil_div64(cap_vis > 0 ? cap_vis : 1, R_BLOCK);
    const unsigned t_grid = (unsigned)min((int64_t)512, ceil_div64(cap_M > 0 ? cap_M : 1, T_TILE));
    const float ts = (float)tile_size;
    if (tight) {
        bin3_rows_count_kernel<true><<<r_grid, R_BLOCK, 0, st>>>(n_vis_ref, recs, vis_ids, N, C, ts, tile_w, tile_h, n_rows, w.row_count,
                                                                w.rbase, w.done_rows, w.row_start);
        bin3_rows_place_kernel<true><<<r_grid, R_BLOCK, 0, st>>>(n_vis_ref, recs, vis_ids, N, C, ts, tile_w, tile_h, n_rows, w.row_start,
                                                                w.rbase, cap_M, w.items);
    } else {
        bin3_rows_count_kernel<false><<<r_grid, R_BLOCK, 0, st>>>(n_vis_ref, recs, vis_ids, N, C, ts, tile_w, tile_h, n_rows, w.row_count,
                                                                 w.rbase, w.done_rows, w.row_start);
        bin3_rows_place_kernel<false><<<r_grid, R_BLOCK, 0, st>>>(n_vis_ref, recs, vis_ids, N, C, ts, tile_w, tile_h, n_rows, w.row_start,
                                                                 w.rbase, cap_M, w.items);
    }
    const int tile_bits_ = bit_length_u32((uint32_t)(tile_w * tile_h));
    const TailFill tail{cap_M > 0 ? ((flags & MTGS_BIN3_FILL_TO_CAP) ? 4 : ((flags & MTGS_BIN3_FILL_TO_M) ? 2 : 0)) : 0, totals, flatten_ids, isect_ids,
                        ((int64_t)(C - 1) << (32 + tile_bits_)) | ((int64_t)(tile_w * tile_h - 1) << 32) | (int64_t)0x7f800000,
                        (flags & MTGS_BIN3_STATUS) ? totals + 1 : nullptr, cap_vis, cap_M};
    bin3_tiles_count_kernel<<<t_grid, T_THREADS, (size_t)n_bins * 4, st>>>(w.row_start + n_rows, cap_M, w.items, tile_w, n_bins,
                                                                            cap_M, w.bins, w.done_tiles, offsets, order, w.n_long);
    bin3_tiles_place_kernel<<<t_grid, T_THREADS, (size_t)n_bins * 4, st>>>(w.row_start + n_rows, cap_M, w.items, tile_w, n_bins,
                                                                            offsets, w.cursor, (const uint64_t *)vis_keys, (uint32_t)(cap_M > 0 ? cap_M : 1), w.keys64, tail);
    const SortEpilogue epi{rank_ids, flatten_ids, isect_ids, vis_ids, (uint32_t)(tile_w * tile_h),
                           bit_length_u32((uint32_t)(tile_w * tile_h)), C == 1, (uint32_t)(cap_vis > 0 ? cap_vis - 1 : 0)};
    static const bool big_lds = [] {
        return hipFuncSetAttribute((const void *)bin3_sort_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   LG_CAP * 8) == hipSuccess &&
               hipFuncSetAttribute((const void *)bin3_tiles_place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   MAX_BINS * 4) == hipSuccess &&
               hipFuncSetAttribute((const void *)bin3_tiles_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   MAX_BINS * 4) == hipSuccess;
    }();
    MTGS_REQUIRE(big_lds, MTGS_ELAUNCH, "mtgs_bin3_build: cannot reserve %d bytes of LDS per workgroup", LG_CAP * 8);
    bin3_sort_large_kernel<<<(unsigned)min(n_bins, 256), LG_THREADS, (size_t)LG_CAP * 8, st>>>(
        offsets, order, n_bins, cap_M, w.keys64, epi);
    bin3_sort_small_kernel<<<(unsigned)n_bins, B3_BLOCK, 0, st>>>(offsets, order, n_bins, w.n_long, cap_M, w.keys64, epi);
    MTGS_CHECK_LAUNCH("mtgs_bin3_build");
    return MTGS_OK;
}
