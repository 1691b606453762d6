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
//      an in-block scan: no inter-block dependency) and, indexed by rank: the 64-byte record (raster_rec.hpp), the flat
//      index (vis_ids) and the depth key (tile count << 40 | camera << 32 | depth bits); vis_rank[flat index] =
//      rank for the backward's expansion pass; the totals (n_vis, M) go to device memory AND to a pinned host mailbox,
//      so that the host learns them without synchronising the stream;
//      optionally the visibility bitmap + per-word rank prefix that the data-parallel gradient exchange
//      (mtgs_amd.dist, csrc/dp.hip) all-gathers -- available at the START of the frame, so that exchange overlaps
//      the compositing.
// Roofline: HBM (streaming): kernel 1 per pair 40 B in + 32..36 B out; kernel 2 per pair 8 B in, per visible pair
// 36 B (+ colours) gathered + 64 + 20 B out.
// Compiled with -ffp-contract=off (mtgs_amd/build.py) like project.hip.
#include "project_fwd_body.hpp"
#include "tile_rect.hpp"
#include "raster_rec.hpp"

namespace {

// Packed {visible pairs, tile intersections} of a chunk / group: visible << 40 | intersections (a chunk holds 256
// pairs, a group 64 chunks; intersections of a group < 2^14 * 2^19 = 2^33).
constexpr int CHUNKS_PER_GROUP = 64;
constexpr int COMPACT_THREADS = 256, COMPACT_ROWS = 8, COMPACT_TILE = COMPACT_THREADS * COMPACT_ROWS;  // 8 chunks
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
    int32_t *__restrict__ tiles_per_gauss, uint64_t *__restrict__ chunk_counts, unsigned long long *__restrict__ group_counts) {
    __shared__ uint64_t s_w[PROJ_BLOCK / 64];
    const int64_t idx = (int64_t)blockIdx.x * PROJ_BLOCK + threadIdx.x;
    uint64_t v = 0;
    if (idx < (int64_t)C * N) {
        const int c = C == 1 ? 0 : (int)(idx / N);
        const int64_t n = idx - (int64_t)c * N;
        const Cam cam = load_cam(viewmats + c * 16, Ks + c * 9);
        const ProjOut o = project_pair(means, quats, scales, cam, n, W, H, eps2d, near_plane, far_plane, radius_clip);
        radii[idx] = o.radius;
        reinterpret_cast<float2 *>(means2d)[idx] = make_float2(o.mx, o.my);
        depths[idx] = o.depth;
        *reinterpret_cast<F3 *>(conics + idx * 3) = F3{o.ca, o.cb, o.cc};
        if (compensations) compensations[idx] = o.comp;
        int32_t cnt = 0;
        float op = 0.f;
        if (o.radius > 0) {
            op = compensations ? opacities[n] * o.comp : opacities[n];   // gsplat rendering.py: opacities [* compensations]
            const Rect q = tile_rect(o.mx, o.my, o.radius, tile_size, tile_w, tile_h);
            cnt = (q.x1 - q.x0) * (q.y1 - q.y0);
            v = (1ull << 40) | (uint64_t)(uint32_t)cnt;
        }
        opac_eff[idx] = op;
        tiles_per_gauss[idx] = cnt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
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
    const int32_t *radii, *tiles_per_gauss;
    const float *means2d, *depths, *conics, *opac_eff;
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
};

// ---- kernel 2: rank of every visible pair (index order) and its packed record.  A block owns 8 chunks; the number of
// visible pairs / intersections in front of it is the sum of the group counts in front of its group plus the chunk
// counts in front of it inside the group: independent loads of values kernel 1 wrote, no inter-block waiting (a
// chained scan inside ONE fused kernel measured 75 us against 40 + 20 us for these two: its ticket, look-back and
// low occupancy behind the ~700-instruction projection cost more than a second pass over 8 bytes per pair).
__global__ __launch_bounds__(COMPACT_THREADS) void front_compact_kernel(const CompactArgs a) {
    __shared__ uint64_t s_red[2][COMPACT_THREADS / 64];
    __shared__ uint64_t s_wave[COMPACT_ROWS][COMPACT_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t total = (int64_t)a.C * a.N;
    const int64_t base = (int64_t)blockIdx.x * COMPACT_TILE;
    const int64_t chunk0 = (int64_t)blockIdx.x * COMPACT_ROWS, group0 = chunk0 / CHUNKS_PER_GROUP;
    // prefix: groups in front, then the chunks of this group in front of the block
    uint64_t pv = 0, pm = 0;
    for (int64_t g = tid; g < group0; g += COMPACT_THREADS) { const uint64_t w = a.group_counts[g]; pv += pk_vis(w); pm += pk_m(w); }
    for (int64_t c = group0 * CHUNKS_PER_GROUP + tid; c < chunk0; c += COMPACT_THREADS) { const uint64_t w = a.chunk_counts[c]; pv += pk_vis(w); pm += pk_m(w); }
    int32_t rad[COMPACT_ROWS], cnt[COMPACT_ROWS];
#pragma unroll
    for (int r = 0; r < COMPACT_ROWS; ++r) {
        const int64_t idx = base + r * COMPACT_THREADS + tid;
        rad[r] = idx < total ? a.radii[idx] : 0;
        cnt[r] = idx < total ? a.tiles_per_gauss[idx] : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pv += __shfl_xor(pv, o, 64); pm += __shfl_xor(pm, o, 64); }
    if (lane == 0) { s_red[0][wave] = pv; s_red[1][wave] = pm; }
    uint64_t incl_w[COMPACT_ROWS];
#pragma unroll
    for (int r = 0; r < COMPACT_ROWS; ++r) {
        uint64_t inc = rad[r] > 0 ? ((1ull << 40) | (uint64_t)(uint32_t)cnt[r]) : 0ull;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        incl_w[r] = inc;
        if (lane == 63) s_wave[r][wave] = inc;
    }
    __syncthreads();
    uint64_t excl_vis = 0, excl_m = 0;
#pragma unroll
    for (int w = 0; w < COMPACT_THREADS / 64; ++w) { excl_vis += s_red[0][w]; excl_m += s_red[1][w]; }
    uint64_t block_tot = 0, pre[COMPACT_ROWS];
#pragma unroll
    for (int r = 0; r < COMPACT_ROWS; ++r) {
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
        if (idx < total) a.vis_rank[idx] = vis ? (int32_t)rank : -1;
        if (vis) {
            if (rank < a.cap_vis) {
                const float2 xy = reinterpret_cast<const float2 *>(a.means2d)[idx];
                const F3 con = *reinterpret_cast<const F3 *>(a.conics + idx * 3);
                const float dep = a.depths[idx], op = a.opac_eff[idx];
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
                a.vis_keys[rank] = ((uint64_t)(uint32_t)cnt[r] << 40) | ((uint64_t)(a.C == 1 ? 0 : idx / a.N) << 32) |
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
        uint64_t n_vis = excl_vis + pk_vis(block_tot), M = excl_m + pk_m(block_tot);
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

// workspace: group counts (zeroed by the call) | chunk counts
inline size_t front_group_bytes(int64_t total) {
    const int64_t chunks = ceil_div64(total > 0 ? total : 1, PROJ_BLOCK);
    return ((size_t)ceil_div64(chunks, CHUNKS_PER_GROUP) * 8 + 255) & ~(size_t)255;
}
inline size_t front_ws_bytes(int64_t total) {
    return front_group_bytes(total) + (size_t)ceil_div64(total > 0 ? total : 1, PROJ_BLOCK) * 8;
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
                              int color_mode, int64_t *totals, int64_t *host_totals, int64_t host_tag, void *ws,
                              size_t ws_bytes, void *stream) {
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
    front_project_kernel<<<(unsigned)ceil_div64(total, PROJ_BLOCK), PROJ_BLOCK, 0, st>>>(
        C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip, opacities, radii,
        means2d, depths, conics, compensations, opac_eff, (float)tile_size, tile_w, tile_h, tiles_per_gauss, chunk_counts,
        group_counts);
    CompactArgs a;
    a.C = C; a.N = N; a.radii = radii; a.tiles_per_gauss = tiles_per_gauss; a.means2d = means2d; a.depths = depths;
    a.conics = conics; a.opac_eff = opac_eff; a.colors = colors; a.DC = D; a.with_depth = with_depth ? 1 : 0;
    a.chunk_counts = chunk_counts; a.group_counts = (const uint64_t *)group_counts;
    a.recs = recs; a.vis_ids = vis_ids; a.vis_keys = (uint64_t *)vis_keys; a.vis_rank = vis_rank; a.cap_vis = cap_vis;
    a.dp_words = (unsigned long long *)dp_words; a.dp_prefix = dp_prefix; a.dp_count = dp_count; a.color_mode = color_mode;
    a.totals = totals; a.host_totals = host_totals; a.host_tag = host_tag;
    front_compact_kernel<<<(unsigned)ceil_div64(total, COMPACT_TILE), COMPACT_THREADS, 0, st>>>(a);
    MTGS_CHECK_LAUNCH("mtgs_front_fwd");
    return MTGS_OK;
}
