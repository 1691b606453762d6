// front.hip -- projection + compaction of the visible Gaussians + the packed records (two launches).
//
// First stage of the one-node rasterization (mtgs_amd.wrapper._FusedRasterization), i.e. of the
// gsplat.rendering.rasterization call at /root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662.
// Does what gsplat 1.4.0 spreads over fully_fused_projection_fwd, the `opacities * compensations` product of
// rendering.py, the count pass of isect_tiles and torch.cumsum:
//   1. front_project_kernel: per (camera, Gaussian) pair the projection (project_fwd_body.hpp, bit-exact against the
//      oracle) and the tile count of its 3-sigma square; all dense gsplat `meta` outputs; per 256-pair chunk (and, by
//      one atomic per block, per 64-chunk group) the number of visible pairs and intersections;
//   2. front_compact_kernel: every visible pair gets its RANK in index order (group counts + chunk counts in front +
//      an in-block scan: no inter-block dependency) and, indexed by rank: the 64-byte record (raster_rec.hpp; its geometry
//      half copied from the chunk-local compact rows kernel 1 left, round 5 -- not gathered from the dense arrays), the flat
//      index (vis_ids) and the depth key (tile count << 40 | camera << 32 | depth bits); vis_rank[flat index] =
//      rank for the backward's expansion pass; the totals (n_vis, M) go to device memory AND to a pinned host mailbox,
//      so that the host learns them without synchronising the stream;
//      optionally the visibility bitmap + per-word rank prefix that the data-parallel gradient exchange
//      (mtgs_amd.dist, csrc/dp.hip) all-gathers -- available at the START of the frame, so that exchange overlaps
//      the compositing.
// Roofline: HBM (streaming): kernel 1 per pair 40 B in + 32..36 B out (+ 48 B per visible pair into chunk-local compact rows);
// kernel 2 per pair 4 B in + 4 B out (radii, vis_rank), per visible pair 48 B (staged row) + colours in, 64 + 12 B out.
// Compiled with -ffp-contract=off (mtgs_amd/build.py) like project.hip.
#include "project_fwd_body.hpp"
#include "tile_rect.hpp"
#include "raster_rec.hpp"

namespace {

// Packed {visible pairs, tile intersections} of a chunk / group: visible << 40 | intersections (a chunk holds 256
// pairs, a group 64 chunks; intersections of a group < 2^14 * 2^19 = 2^33).
constexpr int CHUNKS_PER_GROUP = 64;
// (2 chunks per workgroup: with 8 the grid was 977 workgroups -- under four per CU for a chain of dependent loads; 16 / 8 / 4 / 2 / 1
//  chunks measured 85.1 / 81.7 / 79.9 / 79.0 / 80.2 us for the entry point)
constexpr int COMPACT_THREADS = 256, COMPACT_ROWS = 2, COMPACT_TILE = COMPACT_THREADS * COMPACT_ROWS;
constexpr int STAGE_FLOATS = 12;   // chunk-local compact row of a visible pair: x y a b | c opacity depth radius | tile count - - -
static_assert(COMPACT_THREADS == PROJ_BLOCK, "a row of the compaction kernel = one chunk of the projection kernel");
__device__ __forceinline__ uint64_t pk_vis(uint64_t w) { return w >> 40; }
__device__ __forceinline__ uint64_t pk_m(uint64_t w) { return w & ((1ull << 40) - 1ull); }

// ---- kernel 1: projection of every (camera, Gaussian) pair + the tile count, dense gsplat outputs, and the number of
// visible pairs / intersections of each 256-pair chunk (plus, atomically, of each 64-chunk group).
__global__ __launch_bounds__(PROJ_BLOCK) void front_project_kernel(
    int C, int64_t N, const float *__restrict__ means, const float *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ viewmats, const float *__restrict__ Ks, int W, int H,
    float eps2d, float near_plane, float far_plane, float radius_clip, const float *__restrict__ opacities,
    int32_t *__restrict__ radii, float *__restrict__ means2d, float *__restrict__ depths, float *__restrict__ conics,
    float *__restrict__ compensations, float *__restrict__ opac_eff, float tile_size, int tile_w, int tile_h,
    int32_t *__restrict__ tiles_per_gauss, uint64_t *__restrict__ chunk_counts, unsigned long long *__restrict__ group_counts,
    float *__restrict__ staged) {
    __shared__ uint64_t s_w[PROJ_BLOCK / 64];
    const int64_t idx = (int64_t)blockIdx.x * PROJ_BLOCK + threadIdx.x;
    uint64_t v = 0;
    float4 st0 = make_float4(0.f, 0.f, 0.f, 0.f), st1 = st0;      // the geometry half of this pair's record, if it is visible
    int32_t st_cnt = 0;
    if (idx < (int64_t)C * N) {
        const int c = C == 1 ? 0 : (int)(idx / N);
        const int64_t n = idx - (int64_t)c * N;
        const Cam cam = load_cam(viewmats + c * 16, Ks + c * 9);
        const ProjOut o = project_pair(means, quats, scales, cam, n, W, H, eps2d, near_plane, far_plane, radius_clip);
        radii[idx] = o.radius;
        // (NON-TEMPORAL: the dense meta outputs are not read again inside the frame; the compaction kernel behind re-reads radii and the
        //  chunk-local rows)
        __builtin_nontemporal_store(o.mx, means2d + idx * 2); __builtin_nontemporal_store(o.my, means2d + idx * 2 + 1);
        __builtin_nontemporal_store(o.depth, depths + idx);
        __builtin_nontemporal_store(o.ca, conics + idx * 3); __builtin_nontemporal_store(o.cb, conics + idx * 3 + 1); __builtin_nontemporal_store(o.cc, conics + idx * 3 + 2);
        if (compensations) __builtin_nontemporal_store(o.comp, compensations + idx);
        int32_t cnt = 0;
        float op = 0.f;
        if (o.radius > 0) {
            op = compensations ? opacities[n] * o.comp : opacities[n];   // gsplat rendering.py: opacities [* compensations]
            const Rect q = tile_rect(o.mx, o.my, o.radius, tile_size, tile_w, tile_h);
            cnt = (q.x1 - q.x0) * (q.y1 - q.y0);
            v = (1ull << 40) | (uint64_t)(uint32_t)cnt;
            st0 = make_float4(o.mx, o.my, o.ca, o.cb);
            st1 = make_float4(o.cc, op, o.depth, __int_as_float(o.radius));
            st_cnt = cnt;
        }
        __builtin_nontemporal_store(op, opac_eff + idx);
        __builtin_nontemporal_store(cnt, tiles_per_gauss + idx);
    }
    const uint64_t mine = v;
    // inclusive scan inside the wave (visible << 40 | tiles), the wave totals through LDS
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up(inc, d, 64);
        if ((threadIdx.x & 63) >= d) inc += up;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    // The visible pairs of the chunk leave their record geometry in CHUNK-LOCAL compact rows (staged[chunk * 256 + j], 48 bytes,
    // j = number of visible pairs in front inside the chunk): front_compact_kernel copies those ~38 contiguous rows per chunk to
    // their global ranks instead of gathering 28 bytes from five dense arrays per visible pair (its traffic was 2.1x its
    // algorithmic bytes that way: every 64-byte line of the dense arrays holds a visible pair somewhere)
    if (mine) {
        uint64_t before = inc - mine;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w)
            if (w < (int)(threadIdx.x >> 6)) before += s_w[w];
        float4 *dst = reinterpret_cast<float4 *>(staged + ((int64_t)blockIdx.x * PROJ_BLOCK + (int64_t)pk_vis(before)) * STAGE_FLOATS);
        dst[0] = st0;
        dst[1] = st1;
        dst[2] = make_float4(__int_as_float(st_cnt), 0.f, 0.f, 0.f);
    }
    if (threadIdx.x == 0) {
        uint64_t t = 0;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w) t += s_w[w];
        chunk_counts[blockIdx.x] = t;
        if (t) atomicAdd(&group_counts[blockIdx.x / CHUNKS_PER_GROUP], (unsigned long long)t);
    }
}

struct CompactArgs {
    int C;
    int64_t N;
    const int32_t *radii;
    const float *staged;  // chunk-local compact rows of front_project_kernel
    const float *colors;  // [C*N, DC] (nullable when DC == 0)
    int DC, with_depth, color_mode;
    const uint64_t *chunk_counts, *group_counts;
    // compact outputs, indexed by rank (< cap_vis)
    float *recs;
    int32_t *vis_ids;
    uint64_t *vis_keys;
    int32_t *vis_rank;  // dense [C*N]
    int64_t cap_vis;
    unsigned long long *dp_words;  // nullable
    uint32_t *dp_prefix;           // nullable
    int32_t *dp_count;             // nullable: n_vis as int32 (the first word of the exchange's meta record)
    int64_t *totals;       // device: n_vis << 32 | M
    int64_t *host_totals;  // pinned host mailbox {totals, tag}; nullable
    int64_t host_tag;
    uint32_t *also_zero;   // nullable: words this kernel clears for the caller (the binning's control words: one launch fewer per frame)
    int64_t also_zero_words;
};

// ---- kernel 2: rank of every visible pair (index order) and its packed record.  A block owns COMPACT_ROWS chunks; the number of
// visible pairs / intersections in front of it is the sum of the group counts in front of its group plus the chunk
// counts in front of it inside the group: independent loads of values kernel 1 wrote, no inter-block waiting (a
// chained scan inside ONE fused kernel measured 75 us against 40 + 20 us for these two: its ticket, look-back and
// low occupancy behind the ~700-instruction projection cost more than a second pass over 8 bytes per pair).
__global__ __launch_bounds__(COMPACT_THREADS) void front_compact_kernel(const CompactArgs a) {
    __shared__ uint64_t s_red[3][COMPACT_THREADS / 64];
    __shared__ uint64_t s_wave[COMPACT_ROWS][COMPACT_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t total = (int64_t)a.C * a.N;
    const int64_t base = (int64_t)blockIdx.x * COMPACT_TILE;
    if (a.also_zero)
        for (int64_t i = (int64_t)blockIdx.x * COMPACT_THREADS + tid; i < a.also_zero_words; i += (int64_t)gridDim.x * COMPACT_THREADS)
            a.also_zero[i] = 0u;
    const int64_t chunk0 = (int64_t)blockIdx.x * COMPACT_ROWS, group0 = chunk0 / CHUNKS_PER_GROUP;
    // prefix: groups in front, then the chunks of this group in front of the block
    uint64_t pv = 0, pm = 0;
    for (int64_t g = tid; g < group0; g += COMPACT_THREADS) { const uint64_t w = a.group_counts[g]; pv += pk_vis(w); pm += pk_m(w); }
    for (int64_t c = group0 * CHUNKS_PER_GROUP + tid; c < chunk0; c += COMPACT_THREADS) { const uint64_t w = a.chunk_counts[c]; pv += pk_vis(w); pm += pk_m(w); }
    int32_t rad[COMPACT_ROWS];
#pragma unroll
    for (int r = 0; r < COMPACT_ROWS; ++r) {
        const int64_t idx = base + r * COMPACT_THREADS + tid;
        rad[r] = idx < total ? a.radii[idx] : 0;
    }
    // (the intersections of this block's own chunks, for the totals: the per-pair prefix of the tile counts is not needed)
    uint64_t own_m = 0;
    if (tid < COMPACT_ROWS && chunk0 + tid < ceil_div64(total, PROJ_BLOCK)) own_m = pk_m(a.chunk_counts[chunk0 + tid]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pv += __shfl_xor(pv, o, 64); pm += __shfl_xor(pm, o, 64); own_m += __shfl_xor(own_m, o, 64); }
    if (lane == 0) { s_red[0][wave] = pv; s_red[1][wave] = pm; s_red[2][wave] = own_m; }
    uint64_t incl_w[COMPACT_ROWS];
#pragma unroll
    for (int r = 0; r < COMPACT_ROWS; ++r) {
        uint64_t inc = rad[r] > 0 ? (1ull << 40) : 0ull;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        incl_w[r] = inc;
        if (lane == 63) s_wave[r][wave] = inc;
    }
    __syncthreads();
    uint64_t excl_vis = 0, excl_m = 0, block_m = 0;
#pragma unroll
    for (int w = 0; w < COMPACT_THREADS / 64; ++w) { excl_vis += s_red[0][w]; excl_m += s_red[1][w]; block_m += s_red[2][w]; }
    uint64_t block_tot = 0, pre[COMPACT_ROWS], row0[COMPACT_ROWS];
#pragma unroll
    for (int r = 0; r < COMPACT_ROWS; ++r) {
        row0[r] = block_tot;        // visible pairs of this block in front of row (= chunk) r
#pragma unroll
        for (int w = 0; w < COMPACT_THREADS / 64; ++w) {
            if (w == wave) pre[r] = block_tot;
            block_tot += s_wave[r][w];
        }
    }
#pragma unroll
    for (int r = 0; r < COMPACT_ROWS; ++r) {
        const int64_t idx = base + r * COMPACT_THREADS + tid;
        const bool vis = rad[r] > 0;
        // rank = visible pairs in front: other blocks + earlier rows / waves of this block + lower lanes of this wave
        const int64_t rank = (int64_t)(excl_vis + pk_vis(pre[r]) + pk_vis(incl_w[r])) - (vis ? 1 : 0);
        if (a.dp_words && idx - lane < total) {
            const unsigned long long m = __ballot(vis);
            if (lane == 0) {
                a.dp_words[idx >> 6] = m;
                a.dp_prefix[idx >> 6] = (uint32_t)rank;   // lane 0: number of visible pairs in front of this word
            }
        }
        if (idx < total) __builtin_nontemporal_store(vis ? (int32_t)rank : -1, a.vis_rank + idx);
        if (vis) {
            if (rank < a.cap_vis) {
                // this pair's staged row: chunk r of the block, position = visible pairs of the chunk in front of it
                const int64_t local = (int64_t)(pk_vis(pre[r]) - pk_vis(row0[r]) + pk_vis(incl_w[r])) - 1;
                const float4 *sp = reinterpret_cast<const float4 *>(a.staged + ((chunk0 + r) * PROJ_BLOCK + local) * STAGE_FLOATS);
                const float4 g0 = sp[0], g1 = sp[1];
                const int32_t cnt_r = __float_as_int(sp[2].x);
                const float2 xy = make_float2(g0.x, g0.y);
                const F3 con = F3{g0.z, g0.w, g1.x};
                const float dep = g1.z, op = g1.y;
                float ch[REC_MAX_CHANNELS];
#pragma unroll
                for (int k = 0; k < REC_MAX_CHANNELS; ++k) ch[k] = 0.f;
                if (a.color_mode >= 2) {   // channels 0..2 (mode 3: 0..5, the camera-space normals too) are filled later for the
                    const int open = a.color_mode == 3 ? 6 : 3;   // visible Gaussians (viscolor.hip, normals.hip); `colors` = the rest
#pragma unroll
                    for (int k = 3; k < REC_MAX_CHANNELS; ++k)
                        if (k >= open && k < a.DC) ch[k] = a.colors[idx * (a.DC - open) + (k - open)];
                } else {
#pragma unroll
                    for (int k = 0; k < REC_MAX_CHANNELS; ++k)
                        if (k < a.DC) ch[k] = a.colors[idx * a.DC + k];
                }
                if (a.color_mode == 1) {   // MTGS's colour activation on SH output: clamp(x + 0.5, 0, 1), first 3 channels
#pragma unroll
                    for (int k = 0; k < 3; ++k) ch[k] = fminf(fmaxf(ch[k] + 0.5f, 0.f), 1.f);
                }
                if (a.with_depth) {
#pragma unroll
                    for (int k = 0; k < REC_MAX_CHANNELS; ++k)
                        if (k == a.DC) ch[k] = dep;
                }
                a.vis_ids[rank] = (int32_t)idx;
                // depth key: tile count | camera | depth bits (bin3.hip reads the depth bits: the high word of its
                // per-tile sort keys)
                a.vis_keys[rank] = ((uint64_t)(uint32_t)cnt_r << 40) | ((uint64_t)(a.C == 1 ? 0 : idx / a.N) << 32) |
                                   (uint64_t)__float_as_uint(dep);
                float4 *dst = reinterpret_cast<float4 *>(a.recs + rank * REC_FLOATS);
                dst[0] = make_float4(xy.x, xy.y, con.x, con.y);
                dst[1] = make_float4(con.z, op, rec_s2max(op), __int_as_float(rad[r]));
                dst[2] = make_float4(ch[0], ch[1], ch[2], ch[3]);
                dst[3] = make_float4(ch[4], ch[5], ch[6], ch[7]);
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        uint64_t n_vis = excl_vis + pk_vis(block_tot), M = excl_m + block_m;
        // more than 2^31 - 2 intersections: published as M = 2^31 - 1, which the host refuses (flatten_ids /
        // isect_offsets are int32, so such a frame cannot be rendered anyway)
        if (M > 0x7fffffffull) M = 0x7fffffffull;
        const int64_t packed = (int64_t)((n_vis << 32) | M);
        *a.totals = packed;
        if (a.dp_count) *a.dp_count = (int32_t)n_vis;
        if (a.host_totals) {
            __hip_atomic_store(a.host_totals, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a.host_totals + 1, a.host_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// workspace: group counts (zeroed by the call) | chunk counts | staged rows
inline size_t front_group_bytes(int64_t total) {
    const int64_t chunks = ceil_div64(total > 0 ? total : 1, PROJ_BLOCK);
    return ((size_t)ceil_div64(chunks, CHUNKS_PER_GROUP) * 8 + 255) & ~(size_t)255;
}
inline size_t front_counts_bytes(int64_t total) {
    return (front_group_bytes(total) + (size_t)ceil_div64(total > 0 ? total : 1, PROJ_BLOCK) * 8 + 255) & ~(size_t)255;
}
// ... | chunk-local compact rows of the visible pairs (capacity: every pair; only the visible ones' rows are ever touched)
inline size_t front_ws_bytes(int64_t total) {
    return front_counts_bytes(total) + (size_t)ceil_div64(total > 0 ? total : 1, PROJ_BLOCK) * PROJ_BLOCK * STAGE_FLOATS * 4;
}

}  // namespace

extern "C" int mtgs_front_workspace_bytes(int64_t total_pairs, size_t *bytes) {
    MTGS_REQUIRE(total_pairs >= 0 && bytes, MTGS_EINVAL, "mtgs_front_workspace_bytes: bad arguments");
    *bytes = front_ws_bytes(total_pairs);
    return MTGS_OK;
}

extern "C" int mtgs_front_fwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                              const float *viewmats, const float *Ks, int width, int height, float eps2d,
                              float near_plane, float far_plane, float radius_clip, const float *opacities,
                              const float *colors, int D, int with_depth, int32_t *radii, float *means2d,
                              float *depths, float *conics, float *compensations, float *opac_eff,
                              int tile_size, int tile_w, int tile_h, int32_t *tiles_per_gauss, float *recs,
                              int32_t *vis_ids, int64_t *vis_keys, int32_t *vis_rank,
                              int64_t cap_vis, uint64_t *dp_words, uint32_t *dp_prefix, int32_t *dp_count,
                              int color_mode, int64_t *totals, int64_t *host_totals, int64_t host_tag, void *also_zero,
                              size_t also_zero_bytes, void *ws, size_t ws_bytes, void *stream) {
    MTGS_REQUIRE(!also_zero || ((reinterpret_cast<uintptr_t>(also_zero) | also_zero_bytes) & 3) == 0, MTGS_EINVAL,
                 "mtgs_front_fwd: also_zero must be a 4-byte aligned region of whole words");
    MTGS_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0 && D >= 0 && cap_vis >= 0, MTGS_EINVAL,
                 "mtgs_front_fwd: bad sizes C=%d N=%lld W=%d H=%d D=%d", C, (long long)N, width, height, D);
    MTGS_REQUIRE(tile_size == MTGS_TILE_SIZE, MTGS_EUNSUPPORTED, "mtgs_front_fwd: tile_size=%d (only 16 is implemented)", tile_size);
    MTGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16, MTGS_EINVAL,
                 "mtgs_front_fwd: tile grid %dx%d does not match image %dx%d", tile_w, tile_h, width, height);
    MTGS_REQUIRE((int64_t)tile_w * tile_h < ((int64_t)1 << 19) && C <= 256, MTGS_EUNSUPPORTED,
                 "mtgs_front_fwd: fewer than 2^19 tiles and at most 256 cameras (sort key layout: count << 40 | camera << 32 | depth)");
    MTGS_REQUIRE(D + (with_depth ? 1 : 0) <= REC_MAX_CHANNELS, MTGS_EUNSUPPORTED,
                 "mtgs_front_fwd: %d blended channels (records hold at most %d; use the operator-by-operator path)",
                 D + (with_depth ? 1 : 0), REC_MAX_CHANNELS);
    MTGS_REQUIRE(totals, MTGS_EINVAL, "mtgs_front_fwd: null pointer");
    const int64_t total = (int64_t)C * N;
    hipStream_t st = (hipStream_t)stream;
    if (total == 0) {
        hipError_t e = hipMemsetAsync(totals, 0, sizeof(int64_t), st);
        MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_front_fwd: memset failed");
        if (also_zero && also_zero_bytes)
            if (int rc = mtgs_zero_async(also_zero, also_zero_bytes, st)) return rc;
        if (host_totals) {  // host memory: nothing to wait for
            host_totals[0] = 0;
            __atomic_store_n(host_totals + 1, host_tag, __ATOMIC_RELEASE);
        }
        return MTGS_OK;
    }
    MTGS_REQUIRE(total < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_front_fwd: C*N must fit int32 (flatten_ids are int32)");
    MTGS_REQUIRE(means && quats && scales && viewmats && Ks && opacities &&
                     (colors || D == 0 || (color_mode == 2 && D == 3) || (color_mode == 3 && D == 6)) && radii && means2d &&
                     depths && conics && opac_eff && tiles_per_gauss && recs && vis_ids && vis_keys &&
                     vis_rank && ws,
                 MTGS_EINVAL, "mtgs_front_fwd: null pointer");
    MTGS_REQUIRE(!dp_words == !dp_prefix, MTGS_EINVAL, "mtgs_front_fwd: dp_words and dp_prefix go together");
    MTGS_REQUIRE(color_mode == 0 || ((color_mode == 1 || color_mode == 2) && D >= 3) || (color_mode == 3 && D >= 6), MTGS_EINVAL,
                 "mtgs_front_fwd: color_mode=%d with %d channels", color_mode, D);
    MTGS_REQUIRE(ws_bytes >= front_ws_bytes(total), MTGS_EWORKSPACE, "mtgs_front_fwd: workspace %zu < %zu bytes", ws_bytes,
                 front_ws_bytes(total));
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(recs) & 15) == 0 && (reinterpret_cast<uintptr_t>(ws) & 255) == 0, MTGS_EINVAL,
                 "mtgs_front_fwd: recs must be 16-byte aligned, ws 256-byte aligned");
    if (int rc = mtgs_zero_async(ws, front_group_bytes(total), st)) return rc;
    unsigned long long *group_counts = (unsigned long long *)ws;
    uint64_t *chunk_counts = (uint64_t *)((char *)ws + front_group_bytes(total));
    float *staged = (float *)((char *)ws + front_counts_bytes(total));
    front_project_kernel<<<(unsigned)ceil_div64(total, PROJ_BLOCK), PROJ_BLOCK, 0, st>>>(
        C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip, opacities, radii,
        means2d, depths, conics, compensations, opac_eff, (float)tile_size, tile_w, tile_h, tiles_per_gauss, chunk_counts,
        group_counts, staged);
    CompactArgs a;
    a.C = C; a.N = N; a.radii = radii; a.staged = staged; a.colors = colors; a.DC = D; a.with_depth = with_depth ? 1 : 0;
    a.chunk_counts = chunk_counts; a.group_counts = (const uint64_t *)group_counts;
    a.recs = recs; a.vis_ids = vis_ids; a.vis_keys = (uint64_t *)vis_keys; a.vis_rank = vis_rank; a.cap_vis = cap_vis;
    a.dp_words = (unsigned long long *)dp_words; a.dp_prefix = dp_prefix; a.dp_count = dp_count; a.color_mode = color_mode;
    a.totals = totals; a.host_totals = host_totals; a.host_tag = host_tag;
    a.also_zero = (uint32_t *)also_zero; a.also_zero_words = also_zero ? (int64_t)(also_zero_bytes / 4) : 0;
    front_compact_kernel<<<(unsigned)ceil_div64(total, COMPACT_TILE), COMPACT_THREADS, 0, st>>>(a);
    MTGS_CHECK_LAUNCH("mtgs_front_fwd");
    return MTGS_OK;
}
