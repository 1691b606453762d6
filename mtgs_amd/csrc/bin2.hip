// bin2.hip -- tile binning of the fused rasterization path: everything between front.hip and the compositing.
//
// Same result as gsplat 1.4.0 isect_tiles(sort=True) + isect_offset_encode (the binning stage of
// gsplat.rendering.rasterization, /root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662): flatten_ids,
// isect_ids and isect_offsets are bit-identical (tests/test_gpu_parity.py, tests/test_gpu_fused.py).  The algorithm
// is the depth-ordered binning of bin.hip -- sort the VISIBLE Gaussians by (camera, depth), emit the intersections
// in that order, stable-sort them on the tile bits only -- rebuilt so that a frame's binning is 14 launches instead
// of 27 and NO size has to be known by the host:
//   1. one histogram + one onesweep pass per digit (onesweep.hpp) sort (depth key, rank) of the n_vis visible
//      Gaussians; "rank" = index of the Gaussian's packed record (front.hip), so every later gather is ONE line;
//   2. bin2_scan_kernel   single-pass chained scan of the tile counts in depth order (lookback.hpp);
//   3. bin2_emit_kernel   intersections emitted in depth order, load-balanced by OUTPUT slot: (tile key, rank);
//   4. bin2_tile_hist_kernel  ONE pass over the tile keys counts the intersections of every tile; the block that
//      finishes last turns the counts into isect_offsets (exclusive scan; offsets[T] = M) and into the
//      longest-list-first tile dispatch order of the compositing kernels -- the separate offsets / schedule
//      launches are gone;
//   5. two radix passes (histogram / scan / reorder, radix_sort.hpp) sort the M (tile key, rank) pairs on the tile
//      bits; the last one writes rank_ids[M]
//      (what the compositing kernels index the records and the gradient rows with), gsplat's flatten_ids
//      (= vis_ids[rank]) and, when asked for, gsplat's 64-bit isect_ids.
// Every kernel reads its element count from device memory (min(n_vis, cap_vis) from the front kernel's packed
// totals, min(M, cap_M) from the word the scan writes) and its grid is sized for the CAPACITY, so the whole chain can be
// enqueued before the host knows n_vis and M (speculative sizing; an overflow is detected by the host from the
// front kernel's mailbox and the frame is repeated with exact sizes).
//
// Roofline: HBM.  Algorithmic bytes: depth sort n_vis*12*(1+2*4); scan n_vis*12; emit n_vis*8 + M*8;
// tile histogram M*4; tile sort M*8*2 + M*(8+4+4[+8]).
#include "common.hpp"
#include "tile_rect.hpp"
#include "lookback.hpp"
#include "onesweep.hpp"
#include "radix_sort.hpp"
#include "raster_rec.hpp"

namespace {

constexpr int B2_BLOCK = 256;
constexpr int SCAN_ITEMS = 8, SCAN_TILE = B2_BLOCK * SCAN_ITEMS;
constexpr int MAX_BINS = 12288;  // (camera, tile) pairs whose counts fit the histogram kernel's LDS
constexpr int DEPTH_ITEMS = 4;

// ---- 2. tile counts in depth order -> inclusive prefix sums; *m_eff = min(total, cap_M) ------------------
// (the count of a Gaussian rides in bits 40.. of its sort key: a streaming read, no gather)
__global__ __launch_bounds__(B2_BLOCK) void bin2_scan_kernel(const mtgs_os::SizeRef n_vis_ref,
                                                            const uint64_t *__restrict__ keys_sorted, int32_t *__restrict__ cum,
                                                            uint32_t *__restrict__ ticket, uint64_t *__restrict__ state,
                                                            uint32_t *__restrict__ err, int64_t cap_M, int64_t *__restrict__ m_eff) {
    __shared__ int s_ticket;
    __shared__ uint32_t s_w[B2_BLOCK / 64];
    __shared__ uint64_t s_excl;
    const int bid = mtgs_lb::take_ticket(ticket, &s_ticket);
    const int64_t n = mtgs_os::size_of(n_vis_ref);
    const int64_t nblocks = ceil_div64(n, SCAN_TILE);
    if (bid >= nblocks) {
        if (n == 0 && bid == 0 && threadIdx.x == 0) *m_eff = 0;
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)bid * SCAN_TILE + (int64_t)tid * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = base + i < n ? (uint32_t)(keys_sorted[base + i] >> 40) : 0u;
        sum += v[i];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t wb = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < B2_BLOCK / 64; ++w) {
        if (w < wave) wb += s_w[w];
        tot += s_w[w];
    }
    if (tid < 64) {
        const mtgs_lb::Pair e = mtgs_lb::lookback_wave(state, bid, 0, tot, err);
        if (tid == 0) s_excl = e.lo;
    }
    __syncthreads();
    uint64_t run = s_excl + wb + inc - sum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        run += v[i];
        if (base + i < n) cum[base + i] = (int32_t)(run < mtgs_lb::FIELD_MAX ? run : mtgs_lb::FIELD_MAX);
    }
    if (bid == nblocks - 1 && tid == 0) {
        const int64_t total = (int64_t)(s_excl + tot);
        *m_eff = total < cap_M ? total : cap_M;
    }
}

// ---- 3. emission in depth order, split by OUTPUT slot (see bin.hip::bin_emit_kernel) ------------------------
constexpr int EMIT_ITEMS = 8, EMIT_TILE = B2_BLOCK * EMIT_ITEMS;
// A block owns EMIT_TILE consecutive output slots, a thread EMIT_ITEMS consecutive ones: ONE search per thread finds
// the Gaussian of its first slot, the following slots advance linearly (mostly the same Gaussian: its rectangle is
// computed once), and the eight keys / values of a thread leave as two 16-byte stores each.
__global__ __launch_bounds__(B2_BLOCK) void bin2_emit_kernel(const mtgs_os::SizeRef n_vis_ref, const mtgs_os::SizeRef m_ref,
                                                            const int32_t *__restrict__ ranks_sorted,
                                                            const float *__restrict__ recs, const int32_t *__restrict__ vis_ids,
                                                            int64_t N, int C, const int32_t *__restrict__ cum, float ts, int tw,
                                                            int th, uint32_t *__restrict__ tile_keys, int32_t *__restrict__ vals) {
    __shared__ int32_t s_c[EMIT_TILE];
    const int tid = threadIdx.x;
    const int64_t n_vis = mtgs_os::size_of(n_vis_ref), M = mtgs_os::size_of(m_ref);
    const int64_t s0 = (int64_t)blockIdx.x * EMIT_TILE;
    if (s0 >= M) return;
    // r0 = first r with cum[r] > s0 (cum: inclusive prefix sums, so slot s0 belongs to r0): 256-ary search
    int64_t lo = 0, hi = n_vis;
    while (lo < hi) {
        const int64_t step = (hi - lo + B2_BLOCK - 1) / B2_BLOCK;
        const int64_t pos = lo + (int64_t)tid * step;
        const bool below = pos < hi && (int64_t)cum[pos] <= s0;
        const int cnt = __syncthreads_count(below);
        if (cnt == 0) { hi = lo; break; }
        const int64_t last_below = lo + (int64_t)(cnt - 1) * step;
        hi = min(hi, lo + (int64_t)cnt * step);
        lo = last_below + 1;
    }
    const int64_t r0 = lo;
    // every Gaussian owns >= 1 slot, so the block's slots touch at most Gaussians r0 .. r0 + EMIT_TILE - 1
#pragma unroll
    for (int e = 0; e < EMIT_ITEMS; ++e) {
        const int k = e * B2_BLOCK + tid;
        s_c[k] = (r0 + k < n_vis) ? cum[r0 + k] : 0x7fffffff;
    }
    const int32_t base0 = r0 > 0 ? cum[r0 - 1] : 0;
    __syncthreads();
    const int64_t i0 = s0 + (int64_t)tid * EMIT_ITEMS;
    if (i0 >= M) return;
    int l = 0, h = EMIT_TILE - 1;  // first l with s_c[l] > i0
    while (l < h) {
        const int mid = (l + h) >> 1;
        if ((int64_t)s_c[mid] <= i0) l = mid + 1; else h = mid;
    }
    uint32_t keys[EMIT_ITEMS];
    int32_t ranks[EMIT_ITEMS];
    int cur = -1, rank = 0, bw = 1;
    int32_t excl = 0, incl = 0;
    Rect q{0, 0, 0, 0};
    uint32_t cam_base = 0;
#pragma unroll
    for (int e = 0; e < EMIT_ITEMS; ++e) {
        const int64_t i = i0 + e;
        if (i < M) {
            while (cur < 0 || (int64_t)incl <= i) {   // next Gaussian (first slot: the searched one)
                if (cur >= 0) ++l;
                cur = l;
                excl = l > 0 ? s_c[l - 1] : base0;
                incl = s_c[l];
                rank = ranks_sorted[r0 + l];
                const float *rec = recs + (int64_t)rank * REC_FLOATS;
                const float2 m = *reinterpret_cast<const float2 *>(rec);
                q = tile_rect(m.x, m.y, __float_as_int(rec[7]), ts, tw, th);
                bw = q.x1 - q.x0;
                cam_base = C == 1 ? 0u : ((uint32_t)vis_ids[rank] / (uint32_t)N) * (uint32_t)(tw * th);
            }
            const int local = (int)(i - excl);
            const int row = local / bw, col = local - row * bw;
            keys[e] = cam_base + (uint32_t)((q.y0 + row) * tw + q.x0 + col);
            ranks[e] = rank;
        } else {
            keys[e] = 0; ranks[e] = 0;
        }
    }
    if (i0 + EMIT_ITEMS <= M) {   // (i0 is a multiple of 8: 16-byte aligned in both arrays)
        uint4 *kd = reinterpret_cast<uint4 *>(tile_keys + i0);
        int4 *vd = reinterpret_cast<int4 *>(vals + i0);
        kd[0] = make_uint4(keys[0], keys[1], keys[2], keys[3]); kd[1] = make_uint4(keys[4], keys[5], keys[6], keys[7]);
        vd[0] = make_int4(ranks[0], ranks[1], ranks[2], ranks[3]); vd[1] = make_int4(ranks[4], ranks[5], ranks[6], ranks[7]);
    } else {
#pragma unroll
        for (int e = 0; e < EMIT_ITEMS; ++e)
            if (i0 + e < M) { tile_keys[i0 + e] = keys[e]; vals[i0 + e] = ranks[e]; }
    }
}

// ---- 4. per-tile counts -> offsets, digit totals of the tile sort, tile dispatch order ----------------------
constexpr int TH_THREADS = 1024, TH_KEYS_PER_BLOCK = TH_THREADS * 32, SCHED_BUCKETS = 1024;
__global__ __launch_bounds__(TH_THREADS) void bin2_tile_hist_kernel(
    const mtgs_os::SizeRef m_ref, const uint32_t *__restrict__ tile_keys, int n_bins,
    uint32_t *__restrict__ bins /* [n_bins], zero */, uint32_t *__restrict__ done /* zero */,
    int32_t *__restrict__ offsets /* [n_bins + 1] */, int32_t *__restrict__ order /* [n_bins], nullable */) {
    extern __shared__ uint32_t s_bins[];  // [n_bins]
    __shared__ uint32_t s_aux[SCHED_BUCKETS];
    __shared__ uint32_t s_ws[TH_THREADS / 64];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t M = mtgs_os::size_of(m_ref);
    const int64_t base = (int64_t)blockIdx.x * TH_KEYS_PER_BLOCK;
    if (base < M) {
        for (int b = tid; b < n_bins; b += TH_THREADS) s_bins[b] = 0;
        __syncthreads();
        const int64_t end = min(M, base + TH_KEYS_PER_BLOCK);
        for (int64_t j0 = base + tid; j0 < end; j0 += 8 * TH_THREADS) {   // eight loads in flight per thread
            uint32_t k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) k[u] = j0 + u * TH_THREADS < end ? tile_keys[j0 + u * TH_THREADS] : 0xffffffffu;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k[u] != 0xffffffffu) atomicAdd(&s_bins[k[u]], 1u);
        }
        __syncthreads();
        for (int b = tid; b < n_bins; b += TH_THREADS) {
            const uint32_t c = s_bins[b];
            if (c) atomicAdd(&bins[b], c);
        }
    }
    // last block to arrive finishes the job (every block arrives, also the ones past the data).  ONE lane fences: an
    // agent-scope release writes back the XCD's L2 and costs microseconds per wave (all 16 waves of all blocks
    // fencing measured 96 us for this kernel); the counts themselves are agent-scope atomics, read back below with
    // agent-scope loads, so no acquire is needed on the reading side.
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    for (int b = tid; b < n_bins; b += TH_THREADS) s_bins[b] = mtgs_os::ld32(bins + b);
    __syncthreads();
    // exclusive scan of the counts -> offsets
    const int per = (n_bins + TH_THREADS - 1) / TH_THREADS;   // consecutive bins per thread
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
    uint32_t run = wb + inc - mine;
    for (int b = b0; b < b1; ++b) {
        offsets[b] = (int32_t)run;
        run += s_bins[b];
    }
    if (tid == TH_THREADS - 1) offsets[n_bins] = (int32_t)run;   // == M
    if (!order) return;
    // tile dispatch order: counting sort by decreasing list length (bucket width 4), as blend.hip::tile_schedule_kernel
    s_aux[tid] = 0;
    __syncthreads();
    auto bucket_of = [&](int t) { return SCHED_BUCKETS - 1 - min((int)(s_bins[t] >> 2), SCHED_BUCKETS - 1); };
    for (int t = tid; t < n_bins; t += TH_THREADS) atomicAdd(&s_aux[bucket_of(t)], 1u);
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
    // (ties inside a bucket land in arrival order: a scheduling aid, results do not depend on it)
    for (int t = tid; t < n_bins; t += TH_THREADS) order[atomicAdd(&s_aux[bucket_of(t)], 1u)] = t;
}

// ---- 5. last pass of the tile sort: rank_ids, flatten_ids, isect_ids ----------------------------------------
struct TileEpilogue {
    static constexpr bool enabled = true;
    int32_t *rank_ids, *flatten_ids;
    int64_t *isect_ids;  // nullable
    const int32_t *vis_ids;
    const uint64_t *vis_keys;
    uint32_t n_tiles;
    int tile_bits;
    bool single_cam;
    struct G { int32_t flat; uint32_t depth_bits; };
    __device__ __forceinline__ G gather(int32_t rank) const {
        return G{vis_ids[rank], isect_ids ? (uint32_t)vis_keys[rank] : 0u};
    }
    __device__ __forceinline__ void store(uint32_t dst, uint64_t key, int32_t rank, const G &g) const {
        rank_ids[dst] = rank;
        flatten_ids[dst] = g.flat;
        if (isect_ids) {
            // (one camera -- every MTGS call: no division)
            const int64_t cam = single_cam ? 0 : (uint32_t)key / n_tiles, tile = single_cam ? (uint32_t)key : (uint32_t)key % n_tiles;
            isect_ids[dst] = (cam << (32 + tile_bits)) | (tile << 32) | (int64_t)g.depth_bits;
        }
    }
};

inline int bit_length_u32(uint32_t v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

struct Bin2Workspace {
    // zeroed control region
    char *control;
    size_t control_bytes;
    int64_t *m_eff;
    uint32_t *scan_ticket, *scan_err, *hist_done, *bins;
    uint64_t *scan_state;
    void *depth_ctl;
    void *tile_sort_ws;
    size_t tile_sort_bytes;
    // data
    uint64_t *keys_a, *keys_b;
    int32_t *vals_a, *vals_b, *cum, *tvals_a, *tvals_b;
    uint32_t *tkeys_a, *tkeys_b;
    size_t total;
};
inline Bin2Workspace carve2(char *base, int64_t cap_vis, int64_t cap_M, int n_bins, int depth_bits, int tile_key_bits) {
    Bin2Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += mtgs_os::align256(bytes); return p; };
    const int64_t nv = cap_vis > 0 ? cap_vis : 1, m = cap_M > 0 ? cap_M : 1;
    w.control = base;
    w.m_eff = (int64_t *)take(64);
    char *misc = take(64);
    w.scan_ticket = (uint32_t *)misc; w.scan_err = (uint32_t *)misc + 1; w.hist_done = (uint32_t *)misc + 2;
    w.bins = (uint32_t *)take((size_t)n_bins * 4);
    w.scan_state = (uint64_t *)take((size_t)ceil_div64(nv, SCAN_TILE) * 8);
    w.depth_ctl = take(mtgs_os::control_bytes<DEPTH_ITEMS>(nv, mtgs_os::make_plan(depth_bits).npass));
    w.control_bytes = off;
    w.keys_a = (uint64_t *)take((size_t)nv * 8); w.keys_b = (uint64_t *)take((size_t)nv * 8);
    w.vals_a = (int32_t *)take((size_t)nv * 4); w.vals_b = (int32_t *)take((size_t)nv * 4);
    w.cum = (int32_t *)take((size_t)nv * 4);
    w.tkeys_a = (uint32_t *)take((size_t)m * 4); w.tkeys_b = (uint32_t *)take((size_t)m * 4);
    w.tvals_a = (int32_t *)take((size_t)m * 4); w.tvals_b = (int32_t *)take((size_t)m * 4);
    w.tile_sort_bytes = mtgs_sort::workspace_bytes<uint32_t>(m);
    w.tile_sort_ws = take(w.tile_sort_bytes);
    w.total = off;
    return w;
}
inline int depth_key_bits(int C) {
    int cam_bits = 0;
    for (uint32_t v = (uint32_t)(C - 1); v; v >>= 1) ++cam_bits;
    return 32 + cam_bits;
}
inline int tile_key_bits_of(int C, int tile_w, int tile_h) {
    const int b = bit_length_u32((uint32_t)C * (uint32_t)(tile_w * tile_h) - 1u);
    return b > 0 ? b : 1;
}

}  // namespace

extern "C" int mtgs_bin2_supported(int C, int tile_w, int tile_h, int64_t cap_M) {
    return C > 0 && tile_w > 0 && tile_h > 0 && (int64_t)C * tile_w * tile_h <= MAX_BINS && cap_M < ((int64_t)1 << 30) ? 1 : 0;
}

extern "C" int mtgs_bin2_workspace_bytes(int C, int tile_w, int tile_h, int64_t cap_vis, int64_t cap_M, size_t *bytes) {
    MTGS_REQUIRE(C > 0 && tile_w > 0 && tile_h > 0 && cap_vis >= 0 && cap_M >= 0 && bytes, MTGS_EINVAL,
                 "mtgs_bin2_workspace_bytes: bad arguments");
    *bytes = carve2(nullptr, cap_vis, cap_M, C * tile_w * tile_h, depth_key_bits(C), tile_key_bits_of(C, tile_w, tile_h)).total;
    return MTGS_OK;
}

extern "C" int mtgs_bin2_build(int C, int64_t N, int tile_size, int tile_w, int tile_h, const int64_t *totals,
                               int64_t cap_vis, int64_t cap_M, const float *recs, const int32_t *vis_ids,
                               const int64_t *vis_keys, int32_t *rank_ids,
                               int32_t *flatten_ids, int64_t *isect_ids, int32_t *offsets, int32_t *tile_order,
                               void *ws, size_t ws_bytes, void *stream) {
    MTGS_REQUIRE(C > 0 && N >= 0 && tile_w > 0 && tile_h > 0 && cap_vis >= 0 && cap_M >= 0, MTGS_EINVAL, "mtgs_bin2_build: bad sizes");
    MTGS_REQUIRE(tile_size == MTGS_TILE_SIZE, MTGS_EUNSUPPORTED, "mtgs_bin2_build: tile_size=%d (only 16 is implemented)", tile_size);
    MTGS_REQUIRE(mtgs_bin2_supported(C, tile_w, tile_h, cap_M), MTGS_EUNSUPPORTED,
                 "mtgs_bin2_build: %d x %d x %d (camera, tile) pairs / %lld intersections (at most %d pairs and 2^30 intersections; "
                 "use mtgs_bin_build)", C, tile_w, tile_h, (long long)cap_M, MAX_BINS);
    MTGS_REQUIRE(totals && recs && vis_ids && vis_keys && rank_ids && flatten_ids && offsets && ws, MTGS_EINVAL,
                 "mtgs_bin2_build: null pointer");
    const int n_bins = C * tile_w * tile_h;
    const int dbits = depth_key_bits(C), tbits = tile_key_bits_of(C, tile_w, tile_h);
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, MTGS_EINVAL, "mtgs_bin2_build: workspace must be 256-byte aligned");
    Bin2Workspace w = carve2((char *)ws, cap_vis, cap_M, n_bins, dbits, tbits);
    MTGS_REQUIRE(ws_bytes >= w.total, MTGS_EWORKSPACE, "mtgs_bin2_build: workspace %zu < %zu bytes", ws_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = mtgs_zero_async(w.control, w.control_bytes, st)) return rc;
    const mtgs_os::SizeRef n_vis_ref{totals, 1, cap_vis}, m_ref{w.m_eff, 0, cap_M};
    // 1. depth sort of (key, rank): the values of the first pass are the positions themselves
    {
        mtgs_os::Sorter<uint64_t, DEPTH_ITEMS> s(dbits, cap_vis, w.depth_ctl);
        s.hist(n_vis_ref, (const uint64_t *)vis_keys, st);
        const uint64_t *kin = (const uint64_t *)vis_keys;
        const int32_t *vin = nullptr;
        bool to_a = (s.plan.npass % 2) == 1;   // the last pass lands in (keys_a, vals_a)
        for (int p = 0; p < s.plan.npass; ++p) {
            uint64_t *kout = to_a ? w.keys_a : w.keys_b;
            int32_t *vout = to_a ? w.vals_a : w.vals_b;
            s.pass(p, n_vis_ref, kin, vin, kout, vout, st);
            kin = kout; vin = vout; to_a = !to_a;
        }
    }
    const int32_t *ranks_sorted = w.vals_a;
    // 2. prefix sums of the tile counts in depth order
    bin2_scan_kernel<<<(unsigned)ceil_div64(cap_vis > 0 ? cap_vis : 1, SCAN_TILE), B2_BLOCK, 0, st>>>(
        n_vis_ref, w.keys_a, w.cum, w.scan_ticket, w.scan_state, w.scan_err, cap_M, w.m_eff);
    // 3. emission
    bin2_emit_kernel<<<(unsigned)ceil_div64(cap_M > 0 ? cap_M : 1, EMIT_TILE), B2_BLOCK, 0, st>>>(
        n_vis_ref, m_ref, ranks_sorted, recs, vis_ids, N, C, w.cum, (float)tile_size, tile_w, tile_h, w.tkeys_a, w.tvals_a);
    // 4. per-tile counts -> offsets, dispatch order
    bin2_tile_hist_kernel<<<(unsigned)ceil_div64(cap_M > 0 ? cap_M : 1, TH_KEYS_PER_BLOCK), TH_THREADS, (size_t)n_bins * 4, st>>>(
        m_ref, w.tkeys_a, n_bins, w.bins, w.hist_done, offsets, tile_order);
    // 5. tile sort (13 bits at 1920x1080: 7 + 6), the last pass writes the outputs.  Histogram / scan / reorder per
    // pass (radix_sort.hpp) with the count read on the device: a one-launch pass (onesweep.hpp, as the depth sort
    // uses) measured 53 + 87 us here against 37 + 57 us -- with ~1000 resident blocks the ticket and the cross-block
    // prefix cost more than the two extra launches.
    {
        const TileEpilogue epi{rank_ids, flatten_ids, isect_ids, vis_ids, (const uint64_t *)vis_keys,
                               (uint32_t)(tile_w * tile_h), bit_length_u32((uint32_t)(tile_w * tile_h)), C == 1};
        const int rc = mtgs_sort::sort_pairs<uint32_t, TileEpilogue>(cap_M > 0 ? cap_M : 1, tbits, w.tkeys_a, w.tvals_a, w.tkeys_b,
                                                                     w.tvals_b, w.tile_sort_ws, w.tile_sort_bytes, st,
                                                                     "mtgs_bin2_build(tile sort)", epi, w.m_eff);
        if (rc) return rc;
    }
    MTGS_CHECK_LAUNCH("mtgs_bin2_build");
    return MTGS_OK;
}
