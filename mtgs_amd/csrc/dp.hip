// dp.hip -- receiver side of the sparse, factored gradient exchange of view-parallel data
// parallelism (mtgs_amd/dist.py::SparseGradExchange).  No gsplat counterpart: the reference's only
// collective is nerfstudio's dense DDP all-reduce (/root/reference/mtgs/scene_model/
// custom_pipeline.py:87-89).
//
// Per step every rank renders ONE camera and only the Gaussians visible in it (~15 %) get a non-zero
// gradient; moreover the gradient of the SH coefficients is rank-1 per Gaussian,
//     v_coeffs[n, k, :] = basis_k(normalize(mean_n - cam_pos)) * v_rgb[n, :],
// so 48 floats are fully described by 3 (v_rgb) plus the sender's camera position.  Ranks therefore
// all-gather compact 64-byte rows {v_mean 3, v_quat 4, v_scale 3, v_opacity 1, v_rgb 3, -, index}
// (19 MB instead of 472 MB per rank at 2M Gaussians, SH degree 3) and this kernel adds one sender's
// rows into the dense, replicated gradient tensors, expanding v_rgb through the SH basis on the fly.
// The result equals the dense all-reduce up to fp32 summation order (tests/test_gpu_dp.py).
//
// Roofline: HBM, read-modify-write of (44 + 12K) bytes per row.
#include "common.hpp"
#include "sh_lane.hpp"

namespace {

__device__ __forceinline__ float sh_basis_k(int k, float x, float y, float z) {
    // gsplat spherical_harmonics (sh_coeffs_to_color_fast) basis functions, see sh.hip
    const float z2 = z * z;
    const float fC1 = x * x - y * y, fS1 = 2.f * x * y;
    switch (k) {
        case 0: return 0.2820947917738781f;
        case 1: return -0.48860251190292f * y;
        case 2: return 0.48860251190292f * z;
        case 3: return -0.48860251190292f * x;
        case 4: return 0.5462742152960395f * fS1;
        case 5: return -1.092548430592079f * z * y;
        case 6: return 0.9461746957575601f * z2 - 0.3153915652525201f;
        case 7: return -1.092548430592079f * z * x;
        case 8: return 0.5462742152960395f * fC1;
        default: break;
    }
    const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    const float fTmp1B = 1.445305721320277f * z;
    const float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    const float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    switch (k) {
        case 9: return -0.5900435899266435f * fS2;
        case 10: return fTmp1B * fS1;
        case 11: return fTmp0C * y;
        case 12: return pSH12;
        case 13: return fTmp0C * x;
        case 14: return fTmp1B * fC1;
        case 15: return -0.5900435899266435f * fC2;
        default: break;
    }
    const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    const float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    const float fTmp2B = -1.770130769779931f * z;
    const float fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
    const float pSH6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
    switch (k) {
        case 16: return 0.6258357354491763f * fS3;
        case 17: return fTmp2B * fS2;
        case 18: return fTmp1C * fS1;
        case 19: return fTmp0D * y;
        case 20: return 1.984313483298443f * z * pSH12 - 1.006230589874905f * pSH6;
        case 21: return fTmp0D * x;
        case 22: return fTmp1C * fC1;
        case 23: return fTmp2B * fC2;
        default: return 0.6258357354491763f * fC3;
    }
}

// TPR threads per row; thread j of a row owns gradient component j (geometry) and SH basis j.
template <int TPR>
__global__ __launch_bounds__(256) void dp_accumulate_kernel(
    int64_t n_rows, const float *__restrict__ rows, int K, int nb, const float *__restrict__ means,
    const float *__restrict__ cam_pos, float *__restrict__ v_means, float *__restrict__ v_quats,
    float *__restrict__ v_scales, float *__restrict__ v_opacities, float *__restrict__ v_coeffs) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / TPR;
    const int j = threadIdx.x % TPR;
    if (r >= n_rows) return;
    const float *row = rows + r * 16;
    const int64_t idx = (int64_t)__float_as_int(row[15]);
    if (j < 3) v_means[idx * 3 + j] += row[j];
    else if (j < 7) v_quats[idx * 4 + (j - 3)] += row[j];
    else if (j < 10) v_scales[idx * 3 + (j - 7)] += row[j];
    else if (j == 10) v_opacities[idx] += row[10];
    if (v_coeffs && j < nb) {
        float x = means[idx * 3] - cam_pos[0], y = means[idx * 3 + 1] - cam_pos[1], z = means[idx * 3 + 2] - cam_pos[2];
        const float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
        x *= inorm; y *= inorm; z *= inorm;
        const float b = sh_basis_k(j, x, y, z);
        float *dst = v_coeffs + (idx * K + j) * 3;
        dst[0] += b * row[11]; dst[1] += b * row[12]; dst[2] += b * row[13];
    }
}

// Sender side: compact the rows of the visible Gaussians (unordered).  A block owns a chunk of
// PACK_CHUNK Gaussians: it counts its visible ones, reserves a contiguous range of rows with ONE atomic
// (same-address atomics serialise at ~12 ns each: one per wave cost 380 us at 2M Gaussians) and then
// writes the rows, positions inside the range coming from ballots.
constexpr int PACK_CHUNK = 2048;
__global__ __launch_bounds__(256) void dp_pack_kernel(
    int64_t N, const int32_t *__restrict__ radii, const float *__restrict__ v_means,
    const float *__restrict__ v_quats, const float *__restrict__ v_scales,
    const float *__restrict__ v_opacities, const float *__restrict__ v_rgb, float *__restrict__ rows,
    int64_t capacity, int64_t *__restrict__ count) {
    __shared__ int s_w[4];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * PACK_CHUNK;
    // pass 1: number of visible Gaussians in the chunk
    int mine = 0;
    for (int i = tid; i < PACK_CHUNK; i += 256) mine += (c0 + i < N && radii[c0 + i] > 0) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) s_w[wave] = mine;
    __syncthreads();
    if (tid == 0) {
        const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        s_base = tot ? (long long)atomicAdd((unsigned long long *)count, (unsigned long long)tot) : 0;
    }
    __syncthreads();
    int64_t run = s_base;
    // pass 2: write the rows; `run` advances identically in every thread
    for (int i0 = 0; i0 < PACK_CHUNK; i0 += 256) {
        const int64_t n = c0 + i0 + tid;
        const bool vis = n < N && radii[n] > 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(vis);
        __syncthreads();
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int below = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) below += s_w[w];
            tot += s_w[w];
        }
        if (vis) {
            const int64_t slot = run + below + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (slot < capacity) {  // the caller's buffer holds N rows; never exceeded
                float4 *dst = reinterpret_cast<float4 *>(rows + slot * 16);
                dst[0] = make_float4(v_means[n * 3], v_means[n * 3 + 1], v_means[n * 3 + 2], v_quats[n * 4]);
                dst[1] = make_float4(v_quats[n * 4 + 1], v_quats[n * 4 + 2], v_quats[n * 4 + 3], v_scales[n * 3]);
                dst[2] = make_float4(v_scales[n * 3 + 1], v_scales[n * 3 + 2], v_opacities[n], v_rgb ? v_rgb[n * 3] : 0.f);
                dst[3] = make_float4(v_rgb ? v_rgb[n * 3 + 1] : 0.f, v_rgb ? v_rgb[n * 3 + 2] : 0.f, 0.f, __int_as_float((int)n));
            }
        }
        run += tot;
    }
}

// ---- ordered exchange + one-pass reduction (the default of mtgs_amd.dist.SparseGradExchange) --------------------
// A sender describes WHICH Gaussians it has rows for by a visibility map: words[N/64] (bit n%64 of word n/64 =
// radii[n] > 0) and prefix[N/64] (number of set bits in the words before).  Its rows are packed in index order, so
// row(n) = prefix[n/64] + popcount(words[n/64] below bit n%64): any receiver can look a Gaussian up in any sender's
// rows without a per-sender index array.  With that, the receiver's work is ONE streaming pass over the Gaussians
// (dp_reduce_kernel) that sums every sender's contribution in registers and WRITES the dense gradients once --
// instead of one read-modify-write pass of (44 + 12 K) bytes per row per sender plus a local SH backward
// (7 x 55 + 72 us at 8 ranks, 2M Gaussians).
// Sender, launch 1: visibility words + the number of visible Gaussians per 1024-Gaussian block.
constexpr int VIS_BLOCK = 1024;  // Gaussians per block = 16 words
__global__ __launch_bounds__(VIS_BLOCK) void dp_vis_words_kernel(int64_t N, const int32_t *__restrict__ radii,
                                                                 unsigned long long *__restrict__ words,
                                                                 uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t s_c[VIS_BLOCK / 64];
    const int64_t n = (int64_t)blockIdx.x * VIS_BLOCK + threadIdx.x;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(n < N && radii[n] > 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        if (n < N) words[n >> 6] = m;
        s_c[wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < VIS_BLOCK / 64; ++w) t += s_c[w];
        block_counts[blockIdx.x] = t;
    }
}
// Sender, launch 2: every block sums the counts of the blocks in front of it (a few thousand values), which
// gives the word prefixes (written for the receivers) and the row of each visible Gaussian; the last block
// also publishes the total.
__global__ __launch_bounds__(VIS_BLOCK) void dp_pack_ordered_kernel(
    int64_t N, const unsigned long long *__restrict__ words, const uint32_t *__restrict__ block_counts,
    uint32_t *__restrict__ prefix, int32_t *__restrict__ count, const float *__restrict__ v_means,
    const float *__restrict__ v_quats, const float *__restrict__ v_scales, const float *__restrict__ v_opacities,
    const float *__restrict__ v_rgb, float *__restrict__ rows, int64_t capacity) {
    __shared__ uint32_t s_red[VIS_BLOCK / 64];
    __shared__ uint32_t s_wpre[VIS_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t part = 0;
    for (int b = tid; b < (int)blockIdx.x; b += VIS_BLOCK) part += block_counts[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < VIS_BLOCK / 64; ++w) base += s_red[w];
    const int64_t n = (int64_t)blockIdx.x * VIS_BLOCK + tid;
    const int64_t wi = n >> 6;
    const unsigned long long w = n < N ? words[wi] : 0ull;
    if (lane == 0) s_wpre[wave] = (uint32_t)__popcll(w);
    __syncthreads();
    uint32_t wpre = base, total = base;
#pragma unroll
    for (int k = 0; k < VIS_BLOCK / 64; ++k) {
        if (k < wave) wpre += s_wpre[k];
        total += s_wpre[k];
    }
    if (lane == 0 && n < N) prefix[wi] = wpre;
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *count = (int32_t)total;
    if (n >= N || !((w >> lane) & 1ull)) return;
    const int64_t slot = (int64_t)wpre + __popcll(w & ((1ull << lane) - 1ull));
    if (slot >= capacity) return;
    float4 *dst = reinterpret_cast<float4 *>(rows + slot * 16);
    dst[0] = make_float4(v_means[n * 3], v_means[n * 3 + 1], v_means[n * 3 + 2], v_quats[n * 4]);
    dst[1] = make_float4(v_quats[n * 4 + 1], v_quats[n * 4 + 2], v_quats[n * 4 + 3], v_scales[n * 3]);
    dst[2] = make_float4(v_scales[n * 3 + 1], v_scales[n * 3 + 2], v_opacities[n], v_rgb ? v_rgb[n * 3] : 0.f);
    dst[3] = make_float4(v_rgb ? v_rgb[n * 3 + 1] : 0.f, v_rgb ? v_rgb[n * 3 + 2] : 0.f, 0.f, __int_as_float((int)n));
}

// Receiver: one WAVE per visibility word (64 consecutive Gaussians), so a sender's word and prefix are wave-uniform
// (held in scalar registers for up to DP_BATCH senders at a time) and the per-Gaussian lookup is a shift and a
// popcount.  Inside the wave, 16 lanes per Gaussian (4 Gaussians per step, 16 steps): lane k = gradient component k
// (geometry, k < 11) and SH basis k.  For every sender with the Gaussian's bit set: 64 coalesced bytes of its row,
// geometry added per lane, basis_k(normalize(mean - cam_sender)) * v_rgb accumulated per lane; then ONE write.
struct DpSenders {
    const unsigned long long *words;  // sender r: (char*)words + r * map_stride_bytes
    const uint32_t *prefix;           //           (char*)prefix + r * map_stride_bytes
    int64_t map_stride_bytes;
    const float *rows;                // sender r: rows + r * row_stride (floats); row 0 = first row of word `word0`
    int64_t row_stride;
    const float *cams;                // [W,3]
    int W;
    int64_t word0;                    // first visibility word of the Gaussian range the rows cover
    int64_t row_cap;                  // rows a sender's block HOLDS (0: row_stride / 16, the whole block is rows)
};
constexpr int DP_MAX_SENDERS = 64;  // one lane per sender computes its row span of a tile
constexpr int DP_TILE = 32;    // Gaussians per wave (half a visibility word)
constexpr int DP_MAXSTEP = DP_TILE / 4;
// ROW-centric inside a Gaussian tile: a wave owns 32 consecutive Gaussians and keeps their 16 x 4 accumulators
// (lane k: geometry component k | coefficient k x rgb) in LDS.  Rows are packed in index order, so the rows a sender
// has for the tile are CONTIGUOUS: [prefix + popcount(bits below the tile), + popcount(tile bits)).  The wave walks
// them four at a time (16 lanes per row: dense lanes, ~5 rows per sender and tile at 15 % visibility), finds the
// Gaussian from the index in the row's last word -- the 14 gradient floats leave two spare words in a 64-byte row, so
// the index costs no wire byte, and deriving the position from the map instead (j-th set bit, an LDS table per sender)
// measured 274 us against 225 us at 8 senders --, evaluates the Gaussian's basis function for the sender's camera and
// adds into the LDS accumulators; the dense gradients are written once at the end.  (A
// Gaussian-centric loop with a predicate per (Gaussian, sender) pair ran 48 % of its steps with ~30 % of the lanes
// useful: 512 us at 8 senders.)  The Gaussian range [g_begin, g_end) lets the caller reduce one CHUNK of the index
// range while the next chunk's rows are still on the wire.
// ROW outputs (mtgs_dp_reduce_rows): instead of dense [N, .] tensors the sums leave as compact rows of the UNION of the
// senders' visible sets -- geometry over all senders, colour over the senders of coeff_mask -- numbered in index order by a
// union map (words / prefix, mtgs_dp_union): row(n) = prefix[n / 64] + popcount(words[n / 64] below bit n % 64), exactly as a
// sender numbers its own rows.  geo row (16 floats) = {v_mean 3, v_quat 4, v_scale 3, v_opacity 1 | C0 * sum v_rgb 3 (the
// gradient of SH coefficient 0 over ALL senders: what MTGS's features_dc receives from every traversal) | 0 | index};
// colour row = coeff_stride floats, coefficient k channel c at 3 k + c.  row_of[N] (int32, -1: no sender sees the Gaussian) is
// the map mtgs_adam_step takes; ids[u] = the Gaussian of geo row u (mtgs_node_bwd_rows).  Rows beyond a capacity are dropped.
struct DpRowOut {
    float *geo_rows; const unsigned long long *geo_words; const uint32_t *geo_prefix; int32_t *geo_row_of, *geo_ids; int64_t geo_cap;
    float *coef_rows; const unsigned long long *coef_words; const uint32_t *coef_prefix; int32_t *coef_row_of; int64_t coef_cap;
};
constexpr float kShC0 = 0.2820947917738781f;

template <int MAXDEG, bool ROWS = false>
__global__ __launch_bounds__(256) void dp_reduce_kernel(int64_t g_begin, int64_t g_end, int K, int nb,
                                                        const float *__restrict__ means, const DpSenders S,
                                                        float *__restrict__ v_means, float *__restrict__ v_quats,
                                                        float *__restrict__ v_scales, float *__restrict__ v_opacities,
                                                        float *__restrict__ v_coeffs, unsigned long long coeff_mask, int geom,
                                                        int64_t coeff_stride, const DpRowOut R = DpRowOut{}) {
    // coeff_mask: the senders whose colour factors go into v_coeffs (per-traversal appearance: one pass per traversal
    // writes that traversal's slice, Gaussian n at v_coeffs + n * coeff_stride); geom: this pass also sums and writes
    // the geometry gradients (over ALL senders).
    __shared__ float4 s_acc[4][DP_TILE][16];
    const int lane = threadIdx.x & 63, k = lane & 15, sub = lane >> 4, wave = threadIdx.x >> 6;
    const int64_t tile = g_begin / DP_TILE + (int64_t)blockIdx.x * 4 + wave;
    const int64_t g0 = tile * DP_TILE;
    if (g0 >= g_end) return;
    const int64_t N = g_end;
    const int64_t wi = g0 >> 6;
    const int half = (int)(tile & 1);
    ShLaneConst lc = sh_lane_const(k);
    if (k >= nb) { lc.a0 = 0.f; lc.a1 = 0.f; lc.a2 = 0.f; lc.a3 = 0.f; }
    float4(*acc)[16] = s_acc[wave];
#pragma unroll
    for (int it = 0; it < DP_MAXSTEP; ++it) acc[it * 4 + sub][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the tile's means in LDS: a global gather per row would put a dependent memory round trip into every step
    __shared__ float s_mean[4][DP_TILE * 3];
    float *tmean = s_mean[wave];
    const bool any_colour = v_coeffs != nullptr || (ROWS && R.coef_rows != nullptr);
    if (any_colour) {
        for (int e = lane; e < DP_TILE * 3; e += 64) tmean[e] = g0 * 3 + e < N * 3 ? means[g0 * 3 + e] : 1.f;
    }
    // (first row, number of rows, tile bits) every sender has for this tile: one lane per sender, one memory round trip
    __shared__ int2 s_span[4][DP_MAX_SENDERS];
    unsigned my_bits = 0u;      // this sender's (lane's) Gaussians of the tile
    if (lane < S.W) {
        const char *wp = reinterpret_cast<const char *>(S.words) + lane * S.map_stride_bytes;
        const char *pp = reinterpret_cast<const char *>(S.prefix) + lane * S.map_stride_bytes;
        const unsigned long long wv = reinterpret_cast<const unsigned long long *>(wp)[wi];
        const unsigned lo = (unsigned)wv, hi = (unsigned)(wv >> 32);
        const uint32_t p_w = reinterpret_cast<const uint32_t *>(pp)[wi], p_0 = reinterpret_cast<const uint32_t *>(pp)[S.word0];
        const int start = (int)(p_w - p_0) + (half ? __builtin_popcount(lo) : 0);
        my_bits = half ? hi : lo;
        s_span[wave][lane] = make_int2(start, __builtin_popcount(my_bits));
    }
    // SPARSE WRITE (geom bit 1; the dense outputs were zeroed by the caller -- the compositing forward's riding zeros, dist.py):
    // only the Gaussians some sender has a row for are written, and a tile none has a row for is left alone altogether
    const bool sparse = !ROWS && (geom & 2) != 0;
    unsigned ubits = 0xffffffffu;
    if (sparse) {
        ubits = my_bits;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ubits |= (unsigned)__shfl_xor((int)ubits, o, 64);
        if (ubits == 0u) return;
    }
    geom &= 1;
    // Software pipeline over the senders: the rows of sender r + 1 are in flight while sender r is accumulated
    // (all of a sender's row loads are issued before any is used: one memory round trip per sender, overlapped).
    float nxt[DP_MAXSTEP];
    int nxt_cnt = 0;
    auto issue = [&](int r, float (&dst)[DP_MAXSTEP], int &cnt_out) {
        const int2 sp = s_span[wave][r];
        const int start = __builtin_amdgcn_readfirstlane(sp.x);
        int cnt = __builtin_amdgcn_readfirstlane(sp.y);
        // a sender's rows occupy [r * row_stride, (r + 1) * row_stride): rows beyond that capacity (finish_static: a rank with
        // more rows than the fixed capacity of the all-gather; the caller is told through its overflow flag) are not read
        // (row_cap: a block [rows | meta record] -- finish_touched -- must not be read past its rows: the words behind them are map
        //  bits, and a sender that overflowed its capacity would otherwise hand them to the sum as floats)
        if (S.row_stride > 0 || S.row_cap > 0) {
            const int64_t room = (S.row_cap > 0 ? S.row_cap : S.row_stride / 16) - (int64_t)start;
            cnt = room <= 0 ? 0 : (cnt < room ? cnt : (int)room);
        }
        cnt_out = cnt;
        const float *rows_r = S.rows + (int64_t)r * S.row_stride + (int64_t)start * 16;
#pragma unroll
        for (int st = 0; st < DP_MAXSTEP; ++st) {
            const int j = st * 4 + sub;
            dst[st] = j < cnt ? rows_r[j * 16 + k] : 0.f;
        }
    };
    const int row_lane0 = (lane & ~15) << 2;  // byte address of lane 0 of this 16-lane row (ds_bpermute)
    // (a colour-only pass -- one traversal's slice -- walks ITS senders only: the rows of the others are not even loaded.  With
    //  one traversal per sender the T passes of a step then read every sender's rows twice in total instead of T times.)
    auto next_needed = [&](int r) {
        while (r < S.W && !(geom || (any_colour && ((coeff_mask >> r) & 1ull)))) ++r;
        return r;
    };
    int r_next = next_needed(0);
    if (r_next < S.W) issue(r_next, nxt, nxt_cnt);
    for (int r = r_next; r < S.W; r = r_next) {
        float cur[DP_MAXSTEP];
#pragma unroll
        for (int st = 0; st < DP_MAXSTEP; ++st) cur[st] = nxt[st];
        const int cnt = nxt_cnt;
        r_next = next_needed(r + 1);
        if (r_next < S.W) issue(r_next, nxt, nxt_cnt);
        const bool colour = any_colour && ((coeff_mask >> r) & 1ull);
        if (cnt == 0) continue;  // wave-uniform: none of the tile's Gaussians
        const float cx = S.cams[r * 3], cy = S.cams[r * 3 + 1], cz = S.cams[r * 3 + 2];
#pragma unroll
        for (int st = 0; st < DP_MAXSTEP; ++st) {
            if (st * 4 >= cnt) break;  // wave-uniform
            const bool on = st * 4 + sub < cnt;
            const int vi = __float_as_int(cur[st]);
            const int idx = __builtin_amdgcn_ds_bpermute(row_lane0 + 15 * 4, vi);   // the row's Gaussian index (last word)
            const int pos = on ? (int)(idx - (int)g0) : 0;
            float4 a = acc[pos][k];
            a.x += (geom && k < 14) ? cur[st] : 0.f;      // (lanes 11-13: the sum of v_rgb over ALL senders, used by the row outputs)
            if (colour) {
                const float q0 = __int_as_float(__builtin_amdgcn_ds_bpermute(row_lane0 + 11 * 4, vi));
                const float q1 = __int_as_float(__builtin_amdgcn_ds_bpermute(row_lane0 + 12 * 4, vi));
                const float q2 = __int_as_float(__builtin_amdgcn_ds_bpermute(row_lane0 + 13 * 4, vi));
                float x = tmean[pos * 3] - cx, y = tmean[pos * 3 + 1] - cy, z = tmean[pos * 3 + 2] - cz;
                const float inorm = __builtin_amdgcn_rsqf((x * x + y * y) + z * z);
                x *= inorm; y *= inorm; z *= inorm;
                const float bk = sh_lane_basis<MAXDEG>(lc, x, y, z);
                a.y += bk * q0; a.z += bk * q1; a.w += bk * q2;
            }
            if (on) acc[pos][k] = a;
        }
    }
    if constexpr (ROWS) {
        const bool want_colour = R.coef_rows != nullptr;
        const unsigned long long uwg = geom ? R.geo_words[wi] : 0ull, uwc = want_colour ? R.coef_words[wi] : 0ull;
        const int64_t upg = geom ? (int64_t)R.geo_prefix[wi] : 0, upc = want_colour ? (int64_t)R.coef_prefix[wi] : 0;
#pragma unroll
        for (int it = 0; it < DP_MAXSTEP; ++it) {
            const int64_t n = g0 + it * 4 + sub;
            if (n >= N) continue;
            const int bit = (int)(n & 63);
            const unsigned long long below = (1ull << bit) - 1ull;
            const float4 a = acc[it * 4 + sub][k];
            if (geom) {
                const bool present = (uwg >> bit) & 1ull;
                const int64_t u = upg + __popcll(uwg & below);
                if (k == 0) R.geo_row_of[n] = present ? (int32_t)u : -1;
                if (present && u < R.geo_cap) {
                    const float v = k < 11 ? a.x : (k < 14 ? kShC0 * a.x : (k == 15 ? __int_as_float((int)n) : 0.f));
                    R.geo_rows[u * 16 + k] = v;
                    if (k == 15) R.geo_ids[u] = (int32_t)n;
                }
            }
            if (want_colour) {
                const bool present = (uwc >> bit) & 1ull;
                const int64_t u = upc + __popcll(uwc & below);
                if (k == 0) R.coef_row_of[n] = present ? (int32_t)u : -1;
                if (present && u < R.coef_cap && k < K) {
                    float *dst = R.coef_rows + u * coeff_stride + k * 3;
                    dst[0] = a.y; dst[1] = a.z; dst[2] = a.w;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < DP_MAXSTEP; ++it) {
        const int64_t n = g0 + it * 4 + sub;
        if (n >= N || !((ubits >> (it * 4 + sub)) & 1u)) continue;
        const float4 a = acc[it * 4 + sub][k];
        if (geom) {
            if (k < 3) v_means[n * 3 + k] = a.x;
            else if (k < 7) v_quats[n * 4 + (k - 3)] = a.x;
            else if (k < 10) v_scales[n * 3 + (k - 7)] = a.x;
            else if (k == 10) v_opacities[n] = a.x;
        }
        if (v_coeffs && k < K) {
            float *dst = v_coeffs + n * coeff_stride + k * 3;
            dst[0] = a.y; dst[1] = a.z; dst[2] = a.w;
        }
    }
}

// ---- every colour group of a step in ONE pass (mtgs_dp_reduce_rows_groups) ----------------------------------------------
// A data-parallel step of MTGS renders several traversals: the senders of traversal t sum into slice t of the per-traversal colour
// tensors, all senders into the geometry.  One launch of dp_reduce_kernel<.., true> per traversal walks all N / 32 tiles each time
// (clearing accumulators, loading the tile's means and every sender's span, writing a row map over all N) -- at eight traversals
// the fixed part is paid eight times (968 us at 2M Gaussians against 323 us for one).  Here a wave walks its tile ONCE: first the
// geometry of ALL senders in rank order (the order of the dense reduction: the sums are bit-identical to it), then the groups
// (sender sets, one per rendered traversal) take turns on the colour accumulators, each summing its senders in rank order as its
// own pass would.  A sender's rows are read twice in total, as with one pass per group.
template <int MAXDEG>
__global__ __launch_bounds__(256) void dp_reduce_groups_kernel(int64_t g_begin, int64_t g_end, int K, int nb,
                                                               const float *__restrict__ means, const DpSenders S, int n_groups,
                                                               const mtgs_dp_group *__restrict__ groups, int64_t coeff_stride,
                                                               const DpRowOut R) {
    __shared__ float4 s_acc[4][DP_TILE][16];
    __shared__ float s_mean[4][DP_TILE * 3];
    __shared__ int2 s_span[4][DP_MAX_SENDERS];
    const int lane = threadIdx.x & 63, k = lane & 15, sub = lane >> 4, wave = threadIdx.x >> 6;
    const int64_t tile = g_begin / DP_TILE + (int64_t)blockIdx.x * 4 + wave;
    const int64_t g0 = tile * DP_TILE;
    if (g0 >= g_end) return;
    const int64_t N = g_end;
    const int64_t wi = g0 >> 6;
    const int half = (int)(tile & 1);
    ShLaneConst lc = sh_lane_const(k);
    if (k >= nb) { lc.a0 = 0.f; lc.a1 = 0.f; lc.a2 = 0.f; lc.a3 = 0.f; }
    float4(*acc)[16] = s_acc[wave];
#pragma unroll
    for (int it = 0; it < DP_MAXSTEP; ++it) acc[it * 4 + sub][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    float *tmean = s_mean[wave];
    for (int e = lane; e < DP_TILE * 3; e += 64) tmean[e] = g0 * 3 + e < N * 3 ? means[g0 * 3 + e] : 1.f;
    if (lane < S.W) {
        const char *wp = reinterpret_cast<const char *>(S.words) + lane * S.map_stride_bytes;
        const char *pp = reinterpret_cast<const char *>(S.prefix) + lane * S.map_stride_bytes;
        const unsigned long long wv = reinterpret_cast<const unsigned long long *>(wp)[wi];
        const unsigned lo = (unsigned)wv, hi = (unsigned)(wv >> 32);
        const uint32_t p_w = reinterpret_cast<const uint32_t *>(pp)[wi], p_0 = reinterpret_cast<const uint32_t *>(pp)[S.word0];
        const int start = (int)(p_w - p_0) + (half ? __builtin_popcount(lo) : 0);
        s_span[wave][lane] = make_int2(start, __builtin_popcount(half ? hi : lo));
    }
    const int row_lane0 = (lane & ~15) << 2;
    auto issue = [&](int r, float (&dst)[DP_MAXSTEP], int &cnt_out) {
        const int2 sp = s_span[wave][r];
        const int start = __builtin_amdgcn_readfirstlane(sp.x);
        int cnt = __builtin_amdgcn_readfirstlane(sp.y);
        // (row_cap: a block [rows | meta record] -- finish_touched -- must not be read past its rows: the words behind them are map
        //  bits, and a sender that overflowed its capacity would otherwise hand them to the sum as floats)
        if (S.row_stride > 0 || S.row_cap > 0) {
            const int64_t room = (S.row_cap > 0 ? S.row_cap : S.row_stride / 16) - (int64_t)start;
            cnt = room <= 0 ? 0 : (cnt < room ? cnt : (int)room);
        }
        cnt_out = cnt;
        const float *rows_r = S.rows + (int64_t)r * S.row_stride + (int64_t)start * 16;
#pragma unroll
        for (int st = 0; st < DP_MAXSTEP; ++st) {
            const int j = st * 4 + sub;
            dst[st] = j < cnt ? rows_r[j * 16 + k] : 0.f;
        }
    };
    // the senders of `mask`, in rank order, software-pipelined as in dp_reduce_kernel.  (ONE pipeline over the whole schedule --
    // geometry, then group after group, the next pair's rows in flight across the phase boundaries -- measured SLOWER: 930 against
    // 780 us at eight groups; its bookkeeping costs more registers and branches than the eight pipeline refills.)
    auto walk = [&](unsigned long long mask, bool geom, bool colour) {
        auto next_in = [&](int r) {
            while (r < S.W && !((mask >> r) & 1ull)) ++r;
            return r;
        };
        float nxt[DP_MAXSTEP];
        int nxt_cnt = 0;
        int r_next = next_in(0);
        if (r_next < S.W) issue(r_next, nxt, nxt_cnt);
        for (int r = r_next; r < S.W; r = r_next) {
            float cur[DP_MAXSTEP];
#pragma unroll
            for (int st = 0; st < DP_MAXSTEP; ++st) cur[st] = nxt[st];
            const int cnt = nxt_cnt;
            r_next = next_in(r + 1);
            if (r_next < S.W) issue(r_next, nxt, nxt_cnt);
            if (cnt == 0) continue;
            const float cx = S.cams[r * 3], cy = S.cams[r * 3 + 1], cz = S.cams[r * 3 + 2];
#pragma unroll
            for (int st = 0; st < DP_MAXSTEP; ++st) {
                if (st * 4 >= cnt) break;
                const bool on = st * 4 + sub < cnt;
                const int vi = __float_as_int(cur[st]);
                const int idx = __builtin_amdgcn_ds_bpermute(row_lane0 + 15 * 4, vi);
                const int pos = on ? (int)(idx - (int)g0) : 0;
                float4 a = acc[pos][k];
                a.x += (geom && k < 14) ? cur[st] : 0.f;
                if (colour) {
                    const float q0 = __int_as_float(__builtin_amdgcn_ds_bpermute(row_lane0 + 11 * 4, vi));
                    const float q1 = __int_as_float(__builtin_amdgcn_ds_bpermute(row_lane0 + 12 * 4, vi));
                    const float q2 = __int_as_float(__builtin_amdgcn_ds_bpermute(row_lane0 + 13 * 4, vi));
                    float x = tmean[pos * 3] - cx, y = tmean[pos * 3 + 1] - cy, z = tmean[pos * 3 + 2] - cz;
                    const float inorm = __builtin_amdgcn_rsqf((x * x + y * y) + z * z);
                    x *= inorm; y *= inorm; z *= inorm;
                    const float bk = sh_lane_basis<MAXDEG>(lc, x, y, z);
                    a.y += bk * q0; a.z += bk * q1; a.w += bk * q2;
                }
                if (on) acc[pos][k] = a;
            }
        }
    };
    const unsigned long long all = S.W >= 64 ? ~0ull : ((1ull << S.W) - 1ull);
    if (R.geo_rows) walk(all, true, false);
    for (int j = 0; j < n_groups; ++j) {
        const mtgs_dp_group G = groups[j];
        if (j > 0) {       // the colour accumulators start from zero for every group; the geometry sums stay
#pragma unroll
            for (int it = 0; it < DP_MAXSTEP; ++it) {
                float4 a = acc[it * 4 + sub][k];
                a.y = 0.f; a.z = 0.f; a.w = 0.f;
                acc[it * 4 + sub][k] = a;
            }
        }
        walk(G.mask & all, false, true);
        const unsigned long long uwc = G.coef_words[wi];
        const int64_t upc = (int64_t)G.coef_prefix[wi];
#pragma unroll
        for (int it = 0; it < DP_MAXSTEP; ++it) {
            const int64_t n = g0 + it * 4 + sub;
            if (n >= N) continue;
            const int bit = (int)(n & 63);
            const bool present = (uwc >> bit) & 1ull;
            const int64_t u = upc + __popcll(uwc & ((1ull << bit) - 1ull));
            if (k == 0) G.coef_row_of[n] = present ? (int32_t)u : -1;
            if (present && u < G.coef_cap && k < K) {
                const float4 a = acc[it * 4 + sub][k];
                float *dst = G.coef_rows + u * coeff_stride + k * 3;
                dst[0] = a.y; dst[1] = a.z; dst[2] = a.w;
            }
        }
    }
    if (R.geo_rows) {
        const unsigned long long uwg = R.geo_words[wi];
        const int64_t upg = (int64_t)R.geo_prefix[wi];
#pragma unroll
        for (int it = 0; it < DP_MAXSTEP; ++it) {
            const int64_t n = g0 + it * 4 + sub;
            if (n >= N) continue;
            const int bit = (int)(n & 63);
            const bool present = (uwg >> bit) & 1ull;
            const int64_t u = upg + __popcll(uwg & ((1ull << bit) - 1ull));
            const float4 a = acc[it * 4 + sub][k];
            if (k == 0) R.geo_row_of[n] = present ? (int32_t)u : -1;
            if (present && u < R.geo_cap) {
                const float v = k < 11 ? a.x : (k < 14 ? kShC0 * a.x : (k == 15 ? __int_as_float((int)n) : 0.f));
                R.geo_rows[u * 16 + k] = v;
                if (k == 15) R.geo_ids[u] = (int32_t)n;
            }
        }
    }
}

// Union maps of sender subsets (mtgs_dp_union): launch 1 ORs the senders' visibility words per subset and counts the set bits
// per 256-word block; launch 2 turns the counts into word prefixes (every block sums the few hundred counts in front of it, as
// dp_pack_ordered_kernel does) and publishes the totals, packed like mtgs_front_fwd's totals (count << 32).
constexpr int UNION_BLOCK = 256;
__global__ __launch_bounds__(UNION_BLOCK) void dp_union_words_kernel(int64_t nw, int W, const unsigned long long *__restrict__ words,
                                                                     int64_t map_stride_bytes, const unsigned long long *__restrict__ masks,
                                                                     unsigned long long *__restrict__ uwords, uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t s_c[UNION_BLOCK / 64];
    const int p = blockIdx.y;
    const int64_t wi = (int64_t)blockIdx.x * UNION_BLOCK + threadIdx.x;
    const unsigned long long mask = masks[p];
    unsigned long long u = 0;
    if (wi < nw)
        for (int r = 0; r < W; ++r)
            if ((mask >> r) & 1ull) u |= reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(words) + r * map_stride_bytes)[wi];
    if (wi < nw) uwords[(int64_t)p * nw + wi] = u;
    uint32_t c = (uint32_t)__popcll(u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < UNION_BLOCK / 64; ++w) t += s_c[w];
        block_counts[(int64_t)p * gridDim.x + blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(UNION_BLOCK) void dp_union_prefix_kernel(int64_t nw, const unsigned long long *__restrict__ uwords,
                                                                      const uint32_t *__restrict__ block_counts,
                                                                      uint32_t *__restrict__ uprefix, int64_t *__restrict__ totals) {
    __shared__ uint32_t s_red[UNION_BLOCK / 64];
    __shared__ uint32_t s_w[UNION_BLOCK / 64];
    const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *bc = block_counts + (int64_t)p * gridDim.x;
    uint32_t part = 0;
    for (int b = tid; b < (int)blockIdx.x; b += UNION_BLOCK) part += bc[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) s_red[wave] = part;
    const int64_t wi = (int64_t)blockIdx.x * UNION_BLOCK + tid;
    const uint32_t c = wi < nw ? (uint32_t)__popcll(uwords[(int64_t)p * nw + wi]) : 0u;
    uint32_t inc = c;                     // inclusive scan of the word counts inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < UNION_BLOCK / 64; ++w) base += s_red[w];
    uint32_t before = base + inc - c, total = base;
#pragma unroll
    for (int w = 0; w < UNION_BLOCK / 64; ++w) {
        if (w < wave) before += s_w[w];
        total += s_w[w];
    }
    if (wi < nw) uprefix[(int64_t)p * nw + wi] = before;
    if (blockIdx.x == gridDim.x - 1 && tid == 0) totals[p] = (int64_t)total << 32;
}

// ---- sender: only the rows that CARRY a gradient travel (mtgs_dp_touched_pack) ---------------------------------------------------
// A rank's wire rows cover the Gaussians its camera SEES (the projection backward writes one per visible Gaussian, index order);
// in an opaque scene most of them are hidden behind others -- the compositing terminates before it reaches them -- and their
// rows are exactly zero: 60 % at the headline scene, 91-98 % in MTGS-like scenes.  Three small launches compact the non-zero
// rows (same order) and build THEIR map in the format of a visibility map (words + per-word prefix), so that the receivers'
// reduction runs unchanged on 2.5x .. 50x fewer rows and the wire carries as many fewer bytes.
constexpr int TOUCH_BLOCK = 256;
__device__ __forceinline__ bool wire_row_nonzero(const float *__restrict__ row) {
    const float4 *q = reinterpret_cast<const float4 *>(row);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    return a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f || b.x != 0.f || b.y != 0.f || b.z != 0.f || b.w != 0.f ||
           c.x != 0.f || c.y != 0.f || c.z != 0.f || c.w != 0.f || d.x != 0.f || d.y != 0.f;      // (d.z: pad, d.w: the index)
}
__global__ __launch_bounds__(TOUCH_BLOCK) void dp_touched_words_kernel(int64_t n_rows, const float *__restrict__ rows,
                                                                       unsigned long long *__restrict__ words /* zero */) {
    const int64_t r = (int64_t)blockIdx.x * TOUCH_BLOCK + threadIdx.x;
    if (r >= n_rows) return;
    const float *row = rows + r * 16;
    if (!wire_row_nonzero(row)) return;
    const uint32_t n = (uint32_t)__float_as_int(row[15]);
    atomicOr(words + (n >> 6), 1ull << (n & 63u));
}
__global__ __launch_bounds__(UNION_BLOCK) void dp_touched_count_kernel(int64_t nw, const unsigned long long *__restrict__ words,
                                                                       unsigned long long *__restrict__ out_words,
                                                                       uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t s_c[UNION_BLOCK / 64];
    const int64_t wi = (int64_t)blockIdx.x * UNION_BLOCK + threadIdx.x;
    const unsigned long long u = wi < nw ? words[wi] : 0ull;
    if (wi < nw) out_words[wi] = u;
    uint32_t c = (uint32_t)__popcll(u);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < UNION_BLOCK / 64; ++w) t += s_c[w];
        block_counts[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(TOUCH_BLOCK) void dp_touched_rows_kernel(int64_t n_rows, const float *__restrict__ rows,
                                                                      const unsigned long long *__restrict__ words,
                                                                      const uint32_t *__restrict__ prefix, const int64_t *__restrict__ totals,
                                                                      float *__restrict__ out_rows, int64_t capacity,
                                                                      int32_t *__restrict__ out_count) {
    const int64_t r = (int64_t)blockIdx.x * TOUCH_BLOCK + threadIdx.x;
    if (r == 0) *out_count = (int32_t)(*totals >> 32);
    if (r >= n_rows) return;
    const float4 *src = reinterpret_cast<const float4 *>(rows + r * 16);
    if (!wire_row_nonzero(rows + r * 16)) return;
    const uint32_t n = (uint32_t)__float_as_int(rows[r * 16 + 15]);
    const unsigned long long w = words[n >> 6];
    const int64_t dst = (int64_t)prefix[n >> 6] + __popcll(w & ((1ull << (n & 63u)) - 1ull));
    if (dst >= capacity) return;      // (the count says so: the caller repeats the exchange with the untruncated form)
    float4 *out = reinterpret_cast<float4 *>(out_rows + dst * 16);
    out[0] = src[0]; out[1] = src[1]; out[2] = src[2]; out[3] = src[3];
}

// The same compaction into CHUNKS of the Gaussian index range (mtgs_dp_touched_pack_chunks): chunk c's rows go to their own block
// (a separate all-gather message each, so that a receiver reduces chunk c while chunk c + 1 is on the wire), at most cap[c] of
// them; *overflow = 1 when some chunk had more.
struct DpChunkTab {
    int n;
    int64_t begin[MTGS_DP_MAX_CHUNKS + 1];
    float *rows[MTGS_DP_MAX_CHUNKS];
    int64_t cap[MTGS_DP_MAX_CHUNKS];
};
__global__ __launch_bounds__(TOUCH_BLOCK) void dp_touched_rows_chunks_kernel(int64_t n_rows, const float *__restrict__ rows,
                                                                             const unsigned long long *__restrict__ words,
                                                                             const uint32_t *__restrict__ prefix,
                                                                             const int64_t *__restrict__ totals, const DpChunkTab tab,
                                                                             int32_t *__restrict__ out_count, int32_t *__restrict__ overflow) {
    const int64_t r = (int64_t)blockIdx.x * TOUCH_BLOCK + threadIdx.x;
    if (r == 0) *out_count = (int32_t)(*totals >> 32);
    if (r >= n_rows) return;
    const float4 *src = reinterpret_cast<const float4 *>(rows + r * 16);
    if (!wire_row_nonzero(rows + r * 16)) return;
    const uint32_t n = (uint32_t)__float_as_int(rows[r * 16 + 15]);
    const unsigned long long w = words[n >> 6];
    const int64_t dst = (int64_t)prefix[n >> 6] + __popcll(w & ((1ull << (n & 63u)) - 1ull));
    int c = 0;
#pragma unroll 1
    while (c + 1 < tab.n && (int64_t)n >= tab.begin[c + 1]) ++c;
    const int64_t local = dst - (int64_t)prefix[tab.begin[c] >> 6];      // (chunks begin on word boundaries)
    if (local >= tab.cap[c]) { *overflow = 1; return; }
    float4 *out = reinterpret_cast<float4 *>(tab.rows[c] + local * 16);
    out[0] = src[0]; out[1] = src[1]; out[2] = src[2]; out[3] = src[3];
}

}  // namespace

extern "C" int mtgs_dp_touched_pack_chunks(int64_t n_rows, const float *rows, int64_t N, uint64_t *scratch_words, uint64_t *out_words,
                                           uint32_t *out_prefix, int32_t *out_count, int64_t *totals, uint32_t *block_counts,
                                           const mtgs_dp_chunks *chunks, int32_t *overflow, void *stream) {
    MTGS_REQUIRE(n_rows >= 0 && N >= 0 && N < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_dp_touched_pack_chunks: bad sizes");
    MTGS_REQUIRE(scratch_words && out_words && out_prefix && out_count && totals && block_counts && chunks && overflow &&
                     (n_rows == 0 || rows), MTGS_EINVAL, "mtgs_dp_touched_pack_chunks: null pointer");
    MTGS_REQUIRE(chunks->n >= 1 && chunks->n <= MTGS_DP_MAX_CHUNKS, MTGS_EINVAL, "mtgs_dp_touched_pack_chunks: %d chunks (1..%d)", chunks->n,
                 MTGS_DP_MAX_CHUNKS);
    DpChunkTab tab;
    tab.n = chunks->n;
    for (int c = 0; c <= chunks->n; ++c) {
        const int64_t b = chunks->begin[c];
        MTGS_REQUIRE(b >= 0 && b <= N && (c == 0 ? b == 0 : b >= chunks->begin[c - 1]) && (c == chunks->n ? b == N : (b % 64) == 0), MTGS_EINVAL,
                     "mtgs_dp_touched_pack_chunks: chunk boundary %d = %lld (ascending multiples of 64 from 0 to N = %lld)", c, (long long)b,
                     (long long)N);
        tab.begin[c] = b;
    }
    for (int c = 0; c < chunks->n; ++c) {
        MTGS_REQUIRE(chunks->cap[c] >= 0 && (chunks->cap[c] == 0 || chunks->rows[c]) &&
                         (reinterpret_cast<uintptr_t>(chunks->rows[c]) & 15) == 0, MTGS_EINVAL,
                     "mtgs_dp_touched_pack_chunks: chunk %d: capacity %lld, rows %p (16-byte aligned)", c, (long long)chunks->cap[c],
                     (void *)chunks->rows[c]);
        tab.rows[c] = chunks->rows[c];
        tab.cap[c] = chunks->cap[c];
    }
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(rows) & 15) == 0, MTGS_EINVAL, "mtgs_dp_touched_pack_chunks: rows must be 16-byte aligned");
    const int64_t nw = (N + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = mtgs_zero_async(overflow, sizeof(int32_t), st)) return rc;
    if (nw == 0) return mtgs_zero_async(out_count, sizeof(int32_t), st);
    if (int rc = mtgs_zero_async(scratch_words, (size_t)nw * 8, st)) return rc;
    if (n_rows > 0)
        dp_touched_words_kernel<<<(unsigned)ceil_div64(n_rows, TOUCH_BLOCK), TOUCH_BLOCK, 0, st>>>(n_rows, rows, (unsigned long long *)scratch_words);
    const unsigned grid = (unsigned)ceil_div64(nw, UNION_BLOCK);
    dp_touched_count_kernel<<<grid, UNION_BLOCK, 0, st>>>(nw, (const unsigned long long *)scratch_words, (unsigned long long *)out_words, block_counts);
    dp_union_prefix_kernel<<<dim3(grid, 1), UNION_BLOCK, 0, st>>>(nw, (const unsigned long long *)out_words, block_counts, out_prefix, totals);
    dp_touched_rows_chunks_kernel<<<(unsigned)ceil_div64(n_rows > 0 ? n_rows : 1, TOUCH_BLOCK), TOUCH_BLOCK, 0, st>>>(
        n_rows, rows, (const unsigned long long *)out_words, out_prefix, totals, tab, out_count, overflow);
    MTGS_CHECK_LAUNCH("mtgs_dp_touched_pack_chunks");
    return MTGS_OK;
}

extern "C" int mtgs_dp_touched_pack(int64_t n_rows, const float *rows, int64_t N, uint64_t *scratch_words, uint64_t *out_words,
                                    uint32_t *out_prefix, int32_t *out_count, int64_t *totals, uint32_t *block_counts,
                                    float *out_rows, int64_t capacity, void *stream) {
    MTGS_REQUIRE(n_rows >= 0 && N >= 0 && N < ((int64_t)1 << 31) && capacity >= 0, MTGS_EINVAL, "mtgs_dp_touched_pack: bad sizes");
    MTGS_REQUIRE(scratch_words && out_words && out_prefix && out_count && totals && block_counts && (n_rows == 0 || (rows && out_rows)),
                 MTGS_EINVAL, "mtgs_dp_touched_pack: null pointer");
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(rows) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_rows) & 15) == 0, MTGS_EINVAL,
                 "mtgs_dp_touched_pack: rows must be 16-byte aligned");
    const int64_t nw = (N + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    if (nw == 0) {
        hipError_t e = hipMemsetAsync(out_count, 0, sizeof(int32_t), st);
        MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_dp_touched_pack: memset failed");
        return MTGS_OK;
    }
    if (int rc = mtgs_zero_async(scratch_words, (size_t)nw * 8, st)) return rc;
    if (n_rows > 0)
        dp_touched_words_kernel<<<(unsigned)ceil_div64(n_rows, TOUCH_BLOCK), TOUCH_BLOCK, 0, st>>>(n_rows, rows, (unsigned long long *)scratch_words);
    const unsigned grid = (unsigned)ceil_div64(nw, UNION_BLOCK);
    dp_touched_count_kernel<<<grid, UNION_BLOCK, 0, st>>>(nw, (const unsigned long long *)scratch_words, (unsigned long long *)out_words, block_counts);
    dp_union_prefix_kernel<<<dim3(grid, 1), UNION_BLOCK, 0, st>>>(nw, (const unsigned long long *)out_words, block_counts, out_prefix, totals);
    dp_touched_rows_kernel<<<(unsigned)ceil_div64(n_rows > 0 ? n_rows : 1, TOUCH_BLOCK), TOUCH_BLOCK, 0, st>>>(
        n_rows, rows, (const unsigned long long *)out_words, out_prefix, totals, out_rows, capacity, out_count);
    MTGS_CHECK_LAUNCH("mtgs_dp_touched_pack");
    return MTGS_OK;
}

extern "C" int mtgs_dp_union(int W, int64_t N, const uint64_t *words, int64_t map_stride_bytes, int P, const uint64_t *masks,
                             uint64_t *union_words, uint32_t *union_prefix, int64_t *totals, uint32_t *block_counts, void *stream) {
    MTGS_REQUIRE(W >= 1 && W <= DP_MAX_SENDERS && N >= 0 && P >= 1 && map_stride_bytes >= 0, MTGS_EINVAL, "mtgs_dp_union: bad sizes");
    MTGS_REQUIRE(words && masks && union_words && union_prefix && totals && block_counts, MTGS_EINVAL, "mtgs_dp_union: null pointer");
    const int64_t nw = (N + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    if (nw == 0) {
        hipError_t e = hipMemsetAsync(totals, 0, sizeof(int64_t) * P, st);
        MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_dp_union: memset failed");
        return MTGS_OK;
    }
    const dim3 grid((unsigned)ceil_div64(nw, UNION_BLOCK), (unsigned)P);
    dp_union_words_kernel<<<grid, UNION_BLOCK, 0, st>>>(nw, W, (const unsigned long long *)words, map_stride_bytes,
                                                        (const unsigned long long *)masks, (unsigned long long *)union_words, block_counts);
    dp_union_prefix_kernel<<<grid, UNION_BLOCK, 0, st>>>(nw, (const unsigned long long *)union_words, block_counts, union_prefix, totals);
    MTGS_CHECK_LAUNCH("mtgs_dp_union");
    return MTGS_OK;
}

extern "C" int mtgs_dp_reduce_rows(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words,
                                   const uint32_t *prefix, int64_t map_stride_bytes, const float *rows, int64_t row_stride,
                                   const float *cams, int64_t g_begin, int64_t g_end, uint64_t coeff_mask,
                                   float *geo_rows, const uint64_t *geo_words, const uint32_t *geo_prefix, int32_t *geo_row_of,
                                   int32_t *geo_ids, int64_t geo_cap, float *coef_rows, const uint64_t *coef_words,
                                   const uint32_t *coef_prefix, int32_t *coef_row_of, int64_t coef_cap, int64_t coef_stride,
                                   void *stream) {
    MTGS_REQUIRE(W >= 1 && N >= 0 && map_stride_bytes >= 0 && row_stride >= 0 && geo_cap >= 0 && coef_cap >= 0, MTGS_EINVAL,
                 "mtgs_dp_reduce_rows: bad sizes");
    MTGS_REQUIRE(W <= DP_MAX_SENDERS, MTGS_EUNSUPPORTED, "mtgs_dp_reduce_rows: %d senders (at most %d)", W, DP_MAX_SENDERS);
    if (g_end < 0) g_end = N;
    MTGS_REQUIRE(g_begin >= 0 && g_begin <= g_end && g_end <= N && (g_begin % 64) == 0, MTGS_EINVAL,
                 "mtgs_dp_reduce_rows: range [%lld, %lld) of %lld (the start must be a multiple of 64)", (long long)g_begin,
                 (long long)g_end, (long long)N);
    if (g_end == g_begin) return MTGS_OK;
    MTGS_REQUIRE(geo_rows || coef_rows, MTGS_EINVAL, "mtgs_dp_reduce_rows: nothing to write");
    MTGS_REQUIRE(words && prefix && rows && means, MTGS_EINVAL, "mtgs_dp_reduce_rows: null pointer");
    MTGS_REQUIRE(!geo_rows || (geo_words && geo_prefix && geo_row_of && geo_ids), MTGS_EINVAL, "mtgs_dp_reduce_rows: geometry outputs");
    int nb = 0;
    if (coef_rows) {
        MTGS_REQUIRE(coef_words && coef_prefix && coef_row_of && cams, MTGS_EINVAL, "mtgs_dp_reduce_rows: colour outputs");
        MTGS_REQUIRE(degree >= 0 && degree <= 3 && K <= 16 && (degree + 1) * (degree + 1) <= K && coef_stride >= (int64_t)K * 3,
                     MTGS_EUNSUPPORTED, "mtgs_dp_reduce_rows: degree %d / K %d / stride %lld", degree, K, (long long)coef_stride);
        nb = (degree + 1) * (degree + 1);
    }
    const DpSenders S{(const unsigned long long *)words, prefix, map_stride_bytes, rows, row_stride, cams, W, g_begin / 64, 0};
    const DpRowOut R{geo_rows, (const unsigned long long *)geo_words, geo_prefix, geo_row_of, geo_ids, geo_cap,
                     coef_rows, (const unsigned long long *)coef_words, coef_prefix, coef_row_of, coef_cap};
    const unsigned grid = (unsigned)ceil_div64(ceil_div64(g_end - g_begin, DP_TILE), 4);
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long cm = coeff_mask;
    const int geom = geo_rows ? 1 : 0;
    float *none = nullptr;
    switch (coef_rows ? degree : 0) {
        case 0: dp_reduce_kernel<0, true><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, none, none, none, none, none, cm, geom, coef_stride, R); break;
        case 1: dp_reduce_kernel<1, true><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, none, none, none, none, none, cm, geom, coef_stride, R); break;
        case 2: dp_reduce_kernel<2, true><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, none, none, none, none, none, cm, geom, coef_stride, R); break;
        default: dp_reduce_kernel<3, true><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, none, none, none, none, none, cm, geom, coef_stride, R); break;
    }
    MTGS_CHECK_LAUNCH("mtgs_dp_reduce_rows");
    return MTGS_OK;
}

extern "C" int mtgs_dp_pack(int64_t N, const int32_t *radii, const float *v_means, const float *v_quats,
                            const float *v_scales, const float *v_opacities, const float *v_rgb, float *rows,
                            int64_t capacity, int64_t *count, void *stream) {
    MTGS_REQUIRE(N >= 0 && capacity >= 0, MTGS_EINVAL, "mtgs_dp_pack: bad sizes");
    MTGS_REQUIRE(N < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_dp_pack: N must fit int32");
    MTGS_REQUIRE(count, MTGS_EINVAL, "mtgs_dp_pack: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int64_t), st);
    MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_dp_pack: memset failed");
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(radii && v_means && v_quats && v_scales && v_opacities && rows, MTGS_EINVAL, "mtgs_dp_pack: null pointer");
    dp_pack_kernel<<<(unsigned)ceil_div64(N, PACK_CHUNK), 256, 0, st>>>(N, radii, v_means, v_quats, v_scales, v_opacities, v_rgb,
                                                                 rows, capacity, count);
    MTGS_CHECK_LAUNCH("mtgs_dp_pack");
    return MTGS_OK;
}

extern "C" int mtgs_dp_accumulate(int64_t n_rows, const float *rows, int64_t N, int K, int degree,
                                  const float *means, const float *cam_pos, float *v_means, float *v_quats,
                                  float *v_scales, float *v_opacities, float *v_coeffs, void *stream) {
    MTGS_REQUIRE(n_rows >= 0 && N >= 0, MTGS_EINVAL, "mtgs_dp_accumulate: bad sizes");
    if (n_rows == 0) return MTGS_OK;
    MTGS_REQUIRE(rows && v_means && v_quats && v_scales && v_opacities, MTGS_EINVAL, "mtgs_dp_accumulate: null pointer");
    int nb = 0;
    if (v_coeffs) {
        MTGS_REQUIRE(means && cam_pos && degree >= 0 && degree <= MTGS_MAX_SH_DEGREE && (degree + 1) * (degree + 1) <= K,
                     MTGS_EINVAL, "mtgs_dp_accumulate: bad SH arguments (degree %d, K %d)", degree, K);
        nb = (degree + 1) * (degree + 1);
    }
    hipStream_t st = (hipStream_t)stream;
    if (nb <= 16)
        dp_accumulate_kernel<16><<<(unsigned)ceil_div64(n_rows * 16, 256), 256, 0, st>>>(
            n_rows, rows, K, nb, means, cam_pos, v_means, v_quats, v_scales, v_opacities, v_coeffs);
    else
        dp_accumulate_kernel<32><<<(unsigned)ceil_div64(n_rows * 32, 256), 256, 0, st>>>(
            n_rows, rows, K, nb, means, cam_pos, v_means, v_quats, v_scales, v_opacities, v_coeffs);
    MTGS_CHECK_LAUNCH("mtgs_dp_accumulate");
    return MTGS_OK;
}

extern "C" int mtgs_dp_pack_ordered(int64_t N, const int32_t *radii, const float *v_means, const float *v_quats,
                                    const float *v_scales, const float *v_opacities, const float *v_rgb,
                                    uint64_t *words, uint32_t *prefix, int32_t *count, uint32_t *block_counts,
                                    float *rows, int64_t capacity, void *stream) {
    MTGS_REQUIRE(N >= 0 && N < ((int64_t)1 << 31) && capacity >= 0, MTGS_EINVAL, "mtgs_dp_pack_ordered: bad sizes");
    MTGS_REQUIRE(count, MTGS_EINVAL, "mtgs_dp_pack_ordered: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), st);
        MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_dp_pack_ordered: memset failed");
        return MTGS_OK;
    }
    MTGS_REQUIRE(radii && words && prefix && block_counts && v_means && v_quats && v_scales && v_opacities && rows,
                 MTGS_EINVAL, "mtgs_dp_pack_ordered: null pointer");
    const unsigned grid = (unsigned)ceil_div64(N, VIS_BLOCK);
    dp_vis_words_kernel<<<grid, VIS_BLOCK, 0, st>>>(N, radii, (unsigned long long *)words, block_counts);
    dp_pack_ordered_kernel<<<grid, VIS_BLOCK, 0, st>>>(N, (const unsigned long long *)words, block_counts, prefix, count,
                                                      v_means, v_quats, v_scales, v_opacities, v_rgb, rows, capacity);
    MTGS_CHECK_LAUNCH("mtgs_dp_pack_ordered");
    return MTGS_OK;
}

extern "C" int mtgs_dp_reduce_rows_groups(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words,
                                          const uint32_t *prefix, int64_t map_stride_bytes, const float *rows, int64_t row_stride,
                                          const float *cams, int64_t g_begin, int64_t g_end, int n_groups, const mtgs_dp_group *groups,
                                          float *geo_rows, const uint64_t *geo_words, const uint32_t *geo_prefix, int32_t *geo_row_of,
                                          int32_t *geo_ids, int64_t geo_cap, int64_t coef_stride, void *stream) {
    MTGS_REQUIRE(W >= 1 && N >= 0 && map_stride_bytes >= 0 && row_stride >= 0 && geo_cap >= 0 && n_groups >= 1, MTGS_EINVAL,
                 "mtgs_dp_reduce_rows_groups: bad sizes");
    MTGS_REQUIRE(W <= DP_MAX_SENDERS, MTGS_EUNSUPPORTED, "mtgs_dp_reduce_rows_groups: %d senders (at most %d)", W, DP_MAX_SENDERS);
    if (g_end < 0) g_end = N;
    MTGS_REQUIRE(g_begin >= 0 && g_begin <= g_end && g_end <= N && (g_begin % 64) == 0, MTGS_EINVAL,
                 "mtgs_dp_reduce_rows_groups: range [%lld, %lld) of %lld (the start must be a multiple of 64)", (long long)g_begin,
                 (long long)g_end, (long long)N);
    if (g_end == g_begin) return MTGS_OK;
    MTGS_REQUIRE(words && prefix && rows && means && cams && groups, MTGS_EINVAL, "mtgs_dp_reduce_rows_groups: null pointer");
    MTGS_REQUIRE(!geo_rows || (geo_words && geo_prefix && geo_row_of && geo_ids), MTGS_EINVAL, "mtgs_dp_reduce_rows_groups: geometry outputs");
    MTGS_REQUIRE(degree >= 0 && degree <= 3 && K <= 16 && (degree + 1) * (degree + 1) <= K && coef_stride >= (int64_t)K * 3,
                 MTGS_EUNSUPPORTED, "mtgs_dp_reduce_rows_groups: degree %d / K %d / stride %lld", degree, K, (long long)coef_stride);
    const int nb = (degree + 1) * (degree + 1);
    const DpSenders S{(const unsigned long long *)words, prefix, map_stride_bytes, rows, row_stride, cams, W, g_begin / 64, 0};
    const DpRowOut R{geo_rows, (const unsigned long long *)geo_words, geo_prefix, geo_row_of, geo_ids, geo_cap, nullptr, nullptr, nullptr,
                     nullptr, 0};
    const unsigned grid = (unsigned)ceil_div64(ceil_div64(g_end - g_begin, DP_TILE), 4);
    hipStream_t st = (hipStream_t)stream;
    switch (degree) {
        case 0: dp_reduce_groups_kernel<0><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, n_groups, groups, coef_stride, R); break;
        case 1: dp_reduce_groups_kernel<1><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, n_groups, groups, coef_stride, R); break;
        case 2: dp_reduce_groups_kernel<2><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, n_groups, groups, coef_stride, R); break;
        default: dp_reduce_groups_kernel<3><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, n_groups, groups, coef_stride, R); break;
    }
    MTGS_CHECK_LAUNCH("mtgs_dp_reduce_rows_groups");
    return MTGS_OK;
}

extern "C" int mtgs_dp_reduce_slices_cap(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words,
                                         const uint32_t *prefix, int64_t map_stride_bytes, const float *rows,
                                         int64_t row_stride, int64_t row_cap, const float *cams, float *v_means, float *v_quats,
                                         float *v_scales, float *v_opacities, float *v_coeffs, int64_t g_begin, int64_t g_end,
                                         uint64_t coeff_mask, int write_geometry, int64_t coeff_stride, void *stream) {
    MTGS_REQUIRE(W >= 1 && N >= 0 && map_stride_bytes >= 0 && row_stride >= 0, MTGS_EINVAL, "mtgs_dp_reduce: bad sizes");
    MTGS_REQUIRE(row_cap >= 0 && (row_cap == 0 || row_stride == 0 || row_cap <= row_stride / 16), MTGS_EINVAL,
                 "mtgs_dp_reduce: row_cap %lld rows do not fit a block of %lld floats", (long long)row_cap, (long long)row_stride);
    MTGS_REQUIRE(W <= DP_MAX_SENDERS, MTGS_EUNSUPPORTED, "mtgs_dp_reduce: %d senders (at most %d; use mtgs_dp_accumulate)", W,
                 DP_MAX_SENDERS);
    if (g_end < 0) g_end = N;
    MTGS_REQUIRE(g_begin >= 0 && g_begin <= g_end && g_end <= N && (g_begin % 64) == 0, MTGS_EINVAL,
                 "mtgs_dp_reduce: range [%lld, %lld) of %lld (the start must be a multiple of 64)", (long long)g_begin,
                 (long long)g_end, (long long)N);
    if (g_end == g_begin) return MTGS_OK;
    MTGS_REQUIRE(words && prefix && rows && (!(write_geometry & 1) || (v_means && v_quats && v_scales && v_opacities)), MTGS_EINVAL,
                 "mtgs_dp_reduce: null pointer");
    MTGS_REQUIRE((write_geometry & 1) || v_coeffs, MTGS_EINVAL, "mtgs_dp_reduce: nothing to write");
    int nb = 0;
    if (v_coeffs) {
        MTGS_REQUIRE(means && cams && degree >= 0 && (degree + 1) * (degree + 1) <= K, MTGS_EINVAL,
                     "mtgs_dp_reduce: bad SH arguments (degree %d, K %d)", degree, K);
        MTGS_REQUIRE(degree <= 3 && K <= 16, MTGS_EUNSUPPORTED,
                     "mtgs_dp_reduce: degree %d / K %d (one basis per lane of a 16-lane row: degree <= 3, K <= 16; "
                     "use mtgs_dp_accumulate)", degree, K);
        MTGS_REQUIRE(coeff_stride >= (int64_t)K * 3, MTGS_EINVAL, "mtgs_dp_reduce: coefficient stride %lld < K * 3", (long long)coeff_stride);
        nb = (degree + 1) * (degree + 1);
    }
    MTGS_REQUIRE(means, MTGS_EINVAL, "mtgs_dp_reduce: null pointer");
    const DpSenders S{(const unsigned long long *)words, prefix, map_stride_bytes, rows, row_stride, cams, W, g_begin / 64, row_cap};
    const unsigned grid = (unsigned)ceil_div64(ceil_div64(g_end - g_begin, DP_TILE), 4);  // one wave per 32-Gaussian tile
    hipStream_t st = (hipStream_t)stream;
    const unsigned long long cm = coeff_mask;
    const int geom = write_geometry & 3;      // bit 0: write the geometry gradients; bit 1: SPARSE write into pre-zeroed outputs
    // (the kernel bounds its writes by g_end: the last range ends at N)
    switch (degree) {
        case 0: dp_reduce_kernel<0><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, v_means, v_quats, v_scales, v_opacities, v_coeffs, cm, geom, coeff_stride); break;
        case 1: dp_reduce_kernel<1><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, v_means, v_quats, v_scales, v_opacities, v_coeffs, cm, geom, coeff_stride); break;
        case 2: dp_reduce_kernel<2><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, v_means, v_quats, v_scales, v_opacities, v_coeffs, cm, geom, coeff_stride); break;
        default: dp_reduce_kernel<3><<<grid, 256, 0, st>>>(g_begin, g_end, K, nb, means, S, v_means, v_quats, v_scales, v_opacities, v_coeffs, cm, geom, coeff_stride); break;
    }
    MTGS_CHECK_LAUNCH("mtgs_dp_reduce");
    return MTGS_OK;
}

extern "C" int mtgs_dp_reduce_slices(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words,
                                     const uint32_t *prefix, int64_t map_stride_bytes, const float *rows,
                                     int64_t row_stride, const float *cams, float *v_means, float *v_quats,
                                     float *v_scales, float *v_opacities, float *v_coeffs, int64_t g_begin, int64_t g_end,
                                     uint64_t coeff_mask, int write_geometry, int64_t coeff_stride, void *stream) {
    return mtgs_dp_reduce_slices_cap(W, N, K, degree, means, words, prefix, map_stride_bytes, rows, row_stride, 0, cams, v_means, v_quats,
                                     v_scales, v_opacities, v_coeffs, g_begin, g_end, coeff_mask, write_geometry, coeff_stride, stream);
}

extern "C" int mtgs_dp_reduce(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words,
                              const uint32_t *prefix, int64_t map_stride_bytes, const float *rows,
                              int64_t row_stride, const float *cams, float *v_means, float *v_quats,
                              float *v_scales, float *v_opacities, float *v_coeffs, int64_t g_begin, int64_t g_end,
                              void *stream) {
    return mtgs_dp_reduce_slices(W, N, K, degree, means, words, prefix, map_stride_bytes, rows, row_stride, cams, v_means, v_quats,
                                 v_scales, v_opacities, v_coeffs, g_begin, g_end, ~0ull, 1, (int64_t)K * 3, stream);
}
