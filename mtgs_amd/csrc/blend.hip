// blend.hip -- per-tile front-to-back alpha compositing, forward and backward.
//
// Replaces gsplat 1.4.0 rasterize_to_pixels_fwd / rasterize_to_pixels_bwd, the last stage of
// gsplat.rendering.rasterization (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662).
//
// CDNA4 design (not gsplat's 256-thread / 1-pixel-per-thread CUDA layout):
//  * ONE WAVE PER 16x16 TILE, 4 PIXELS PER LANE (lane l owns column l%16 of rows l/16 + {0,4,8,12}).
//    A tile is a single wavefront, so there is no workgroup barrier anywhere in the hot loop,
//    early termination is one ballot, and in the backward the cross-lane reduction of every
//    per-Gaussian gradient is amortised over 4 pixels per lane and issues ONE atomic instruction
//    per (tile, Gaussian) instead of one set per 32-thread warp (8 per tile in the reference layout).
//  * Per-tile Gaussian chunks are staged through LDS 64 at a time (coalesced flatten_ids read one
//    chunk AHEAD, gathered attribute rows), then consumed as wave-uniform broadcast ds_read_b128.
//  * "Does this Gaussian touch any pixel of the tile" is an OR of the per-pixel compare masks
//    (SGPR pairs) -- one s_cbranch, no cross-lane traffic.
//  * Cross-lane sums: all per-Gaussian gradient components are reduced TOGETHER with the transposed
//    v_permlane16/32_swap + DPP reduction of wave_reduce.hpp (30 VALU ops for 12 values, no LDS),
//    which leaves value j in the lanes of row j%4 of register j/4: lane 16*(j%4) + j/4 issues the
//    global_atomic_add_f32 (unsafeAtomicAdd = the hardware fp32 atomic), i.e. ONE atomic
//    instruction per (tile, Gaussian).
//  * 1/(1-alpha) is v_rcp_f32 and exp is v_exp_f32 (the reference is built with --use_fast_math).
//  * Tiles are dispatched longest-list-first (mtgs_tile_schedule): 8160 single-wave workgroups over
//    1024 SIMDs leave a long tail otherwise.
//  * Wide channel counts (D > 8) fall back to 1 pixel per lane / 4 waves per tile.
//
// Roofline: the kernels are VALU bound (about 25 / 70 flops per pixel x Gaussian pair, fwd / bwd)
// -- MFMA is deliberately unused, there is no dense contraction.  Algorithmic HBM bytes:
//   fwd: M*(4 + 24 + 4D) gathered attributes + P*(4D + 8) written
//   bwd: P*(4D + 12) read + M*(4 + 24 + 4D) gathered + N_vis*(24 + 4D (+8 absgrad)) accumulated
#include "common.hpp"
#include "wave_reduce.hpp"

namespace {

constexpr float kAlphaMax = MTGS_ALPHA_MAX;
constexpr float kAlphaMin = MTGS_ALPHA_MIN;
constexpr float kTMin = MTGS_T_MIN;

template <int D>
struct Rec {  // LDS record per staged Gaussian, in floats: x y a b | c opac col[D] (padded to x4)
    static constexpr int N = ((6 + D + 3) / 4) * 4;
};

// Block -> tile mapping.  `order` (from mtgs_tile_schedule) lists tiles by decreasing work so the
// hardware dispatcher, which starts workgroups in blockIdx order as slots free up, runs the
// longest tiles first (LPT scheduling) -- measured: balance matters more than keeping
// neighbouring tiles on one XCD's L2 (contiguous per-XCD bands were 20 % slower).
__device__ __forceinline__ int64_t block_to_tile(const int32_t *__restrict__ order) {
    return order ? (int64_t)order[blockIdx.x] : (int64_t)blockIdx.x;
}

// Gather one Gaussian's attributes (row g of the per-camera arrays) into this thread's LDS record.
template <int D>
__device__ __forceinline__ void stage_record(float *__restrict__ s_rec, const float *__restrict__ means2d,
                                             const float *__restrict__ conics,
                                             const float *__restrict__ colors,
                                             const float *__restrict__ opacities, int32_t g) {
    constexpr int REC = Rec<D>::N;
    const float2 xy = reinterpret_cast<const float2 *>(means2d)[g];
    const float ca = conics[(int64_t)g * 3], cb = conics[(int64_t)g * 3 + 1], cc = conics[(int64_t)g * 3 + 2];
    const float op = opacities[g];
    float r[REC];
    r[0] = xy.x; r[1] = xy.y; r[2] = ca; r[3] = cb; r[4] = cc; r[5] = op;
#pragma unroll
    for (int k = 0; k < D; ++k) r[6 + k] = colors[(int64_t)g * D + k];
#pragma unroll
    for (int k = 6 + D; k < REC; ++k) r[k] = 0.f;
    float4 *dst = reinterpret_cast<float4 *>(s_rec + threadIdx.x * REC);
#pragma unroll
    for (int k = 0; k < REC / 4; ++k) dst[k] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
}

// alpha of one Gaussian at one pixel.  u = a dx + b dy and w = b dx + c dy are kept: the backward
// needs them for the mean gradient.  sigma = 0.5 (dx u + dy w) == 0.5 (a dx^2 + c dy^2) + b dx dy.
struct GaussEval { float u, w, sigma, vis, alpha_raw; };
__device__ __forceinline__ GaussEval eval_gauss(float a, float b, float c, float opac, float dx, float dy) {
    GaussEval e;
    e.u = a * dx + b * dy;
    e.w = b * dx + c * dy;
    e.sigma = 0.5f * (dx * e.u + dy * e.w);
    e.vis = __expf(-e.sigma);
    e.alpha_raw = opac * e.vis;
    return e;
}

// ------------------------------------------------------------------------------------------------
template <int D, int PPL>
__global__ __launch_bounds__(256 / PPL) void blend_fwd_kernel(
    int C, const float *__restrict__ means2d, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ backgrounds, int W, int H, int tw, int th,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int64_t M,
    float *__restrict__ render, float *__restrict__ alphas, int32_t *__restrict__ last_ids,
    const int32_t *__restrict__ order) {
    constexpr int NT = 256 / PPL, ROWS = NT / 16, REC = Rec<D>::N;
    __shared__ __attribute__((aligned(16))) float s_rec[NT * REC];
    const int64_t n_tiles = (int64_t)tw * th, total_tiles = (int64_t)C * n_tiles;
    const int64_t tile = block_to_tile(order);
    const int cam = (int)(tile / n_tiles);
    const int t_in = (int)(tile - (int64_t)cam * n_tiles);
    const int ty = t_in / tw, tx = t_in - ty * tw;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int ix = tx * 16 + lx;
    const float px = (float)ix + 0.5f;
    int iy[PPL];
    float py[PPL], T[PPL], acc[PPL][D];
    int32_t last[PPL];
    bool done[PPL], inside[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        iy[p] = ty * 16 + ly + p * ROWS;
        py[p] = (float)iy[p] + 0.5f;
        inside[p] = ix < W && iy[p] < H;
        done[p] = !inside[p];
        T[p] = 1.f;
        last[p] = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) acc[p][k] = 0.f;
    }
    const int64_t start = offsets[tile];
    const int64_t end = (tile == total_tiles - 1) ? M : (int64_t)offsets[tile + 1];

    // flatten_ids of the NEXT chunk are fetched while the current chunk is composited
    int32_t g_next = (start + tid < end) ? flatten_ids[start + tid] : 0;
    for (int64_t b0 = start; b0 < end; b0 += NT) {
        bool all_done = true;
#pragma unroll
        for (int p = 0; p < PPL; ++p) all_done = all_done && done[p];
        if (__syncthreads_and(all_done)) break;
        const int32_t g_cur = g_next;
        if (b0 + tid < end) stage_record<D>(s_rec, means2d, conics, colors, opacities, g_cur);
        if (b0 + NT + tid < end) g_next = flatten_ids[b0 + NT + tid];
        __syncthreads();
        const int bsz = (int)min((int64_t)NT, end - b0);
        for (int t = 0; t < bsz; ++t) {
            const float4 r0 = *reinterpret_cast<const float4 *>(s_rec + t * REC);
            const float2 r1 = *reinterpret_cast<const float2 *>(s_rec + t * REC + 4);
            const float dx = r0.x - px;
            float alpha[PPL];
            bool valid[PPL];
            unsigned long long any = 0;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                const GaussEval e = eval_gauss(r0.z, r0.w, r1.x, r1.y, dx, r0.y - py[p]);
                alpha[p] = fminf(kAlphaMax, e.alpha_raw);
                valid[p] = !done[p] && e.sigma >= 0.f && alpha[p] >= kAlphaMin;
                any |= __ballot(valid[p]);
            }
            if (any == 0) continue;
            float col[D];
#pragma unroll
            for (int k = 0; k < D; ++k) col[k] = s_rec[t * REC + 6 + k];
            unsigned long long stopped = 0;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                const float next_T = T[p] * (1.f - alpha[p]);
                const bool stop = valid[p] && next_T <= kTMin;
                const bool use = valid[p] && !stop;
                stopped |= __ballot(stop);
                done[p] = done[p] || stop;
                const float w = use ? alpha[p] * T[p] : 0.f;
#pragma unroll
                for (int k = 0; k < D; ++k) acc[p][k] += col[k] * w;
                last[p] = use ? (int32_t)(b0 + t) : last[p];
                T[p] = use ? next_T : T[p];
            }
            if (stopped) {
                bool ad = true;
#pragma unroll
                for (int p = 0; p < PPL; ++p) ad = ad && done[p];
                if (__all(ad)) break;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        if (inside[p]) {
            const int64_t pid = ((int64_t)cam * H + iy[p]) * W + ix;
            alphas[pid] = 1.f - T[p];
            last_ids[pid] = last[p];
#pragma unroll
            for (int k = 0; k < D; ++k)
                render[pid * D + k] = backgrounds ? acc[p][k] + T[p] * backgrounds[cam * D + k] : acc[p][k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Gradient components per Gaussian, in reduction order: xy(2) |xy|(2) conic(3) opacity(1) colour(D)
template <int D>
struct GradLayout {
    static constexpr int NV = 8 + D;
    static constexpr int NR = (NV + 3) / 4;
};

template <int D, int PPL>
__global__ __launch_bounds__(256 / PPL) void blend_bwd_kernel(
    int C, const float *__restrict__ means2d, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ backgrounds, int W, int H, int tw, int th,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int64_t M,
    const float *__restrict__ alphas, const int32_t *__restrict__ last_ids,
    const float *__restrict__ v_render, const float *__restrict__ v_alphas,
    float *__restrict__ v_means2d, float *__restrict__ v_means2d_abs, float *__restrict__ v_conics,
    float *__restrict__ v_colors, float *__restrict__ v_opacities,
    const int32_t *__restrict__ order) {
    constexpr int NT = 256 / PPL, ROWS = NT / 16, REC = Rec<D>::N;
    constexpr int NV = GradLayout<D>::NV, NR = GradLayout<D>::NR;
    __shared__ __attribute__((aligned(16))) float s_rec[NT * REC];
    __shared__ int32_t s_id[NT];
    __shared__ int32_t s_max[NT / 64];
    const int64_t n_tiles = (int64_t)tw * th, total_tiles = (int64_t)C * n_tiles;
    const int64_t tile = block_to_tile(order);
    const int64_t start = offsets[tile];
    const int64_t end = (tile == total_tiles - 1) ? M : (int64_t)offsets[tile + 1];
    if (end <= start) return;
    const int cam = (int)(tile / n_tiles);
    const int t_in = (int)(tile - (int64_t)cam * n_tiles);
    const int ty = t_in / tw, tx = t_in - ty * tw;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4, lane = tid & 63;
    const int ix = tx * 16 + lx;
    const float px = (float)ix + 0.5f;
    float py[PPL], T[PPL], Tf_va[PPL], buf[PPL][D], vr[PPL][D];
    int32_t bin_final[PPL];
    int32_t my_max = -1;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int iy = ty * 16 + ly + p * ROWS;
        py[p] = (float)iy + 0.5f;
        const bool inside = ix < W && iy < H;
        const int64_t pid = ((int64_t)cam * H + (inside ? iy : 0)) * W + (inside ? ix : 0);
        const float T_final = 1.f - alphas[pid];
        T[p] = T_final;
        bin_final[p] = inside ? last_ids[pid] : -1;
        my_max = max(my_max, bin_final[p]);
        float bgd = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            buf[p][k] = 0.f;
            vr[p][k] = v_render[pid * D + k];
            if (backgrounds) bgd += backgrounds[cam * D + k] * vr[p][k];
        }
        // d(alpha_out)/d(alpha_i) and the background term share the factor T_final / (1 - alpha_i)
        Tf_va[p] = T_final * (v_alphas[pid] - bgd);
    }
    // tile-wide newest contributor
    int32_t wmax = wave_max_i32(my_max);
    if (NT > 64) {
        if (lane == 0) s_max[tid >> 6] = wmax;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) wmax = max(wmax, s_max[w]);
    }
    const int64_t top = wmax;  // sorted index of the last Gaussian any pixel of the tile used
    if (top < start) return;

    // which gradient component this lane adds to memory after the transposed reduction:
    // component j = 4*col + row lives in row `row` of register `col`.
    const int a_col = lane & 15, a_row = lane >> 4;
    const int j = 4 * a_col + a_row;
    float *a_base = nullptr;
    int a_stride = 0;
    if (a_col < NR && j < NV) {
        if (j < 2) { a_base = v_means2d + j; a_stride = 2; }
        else if (j < 4) { a_base = v_means2d_abs ? v_means2d_abs + (j - 2) : nullptr; a_stride = 2; }
        else if (j < 7) { a_base = v_conics + (j - 4); a_stride = 3; }
        else if (j < 8) { a_base = v_opacities; a_stride = 1; }
        else { a_base = v_colors + (j - 8); a_stride = D; }
    }

    int32_t g_next = (top - tid >= start) ? flatten_ids[top - tid] : 0;
    for (int64_t hi = top; hi >= start; hi -= NT) {
        if (hi != top) __syncthreads();
        const int32_t g_cur = g_next;
        if (hi - tid >= start) {
            stage_record<D>(s_rec, means2d, conics, colors, opacities, g_cur);
            s_id[tid] = g_cur;
        }
        if (hi - NT - tid >= start) g_next = flatten_ids[hi - NT - tid];
        __syncthreads();
        const int bsz = (int)min((int64_t)NT, hi - start + 1);
        for (int t = 0; t < bsz; ++t) {
            const int32_t idx = (int32_t)(hi - t);
            const float4 r0 = *reinterpret_cast<const float4 *>(s_rec + t * REC);
            const float2 r1 = *reinterpret_cast<const float2 *>(s_rec + t * REC + 4);
            const float opac = r1.y;
            const float dx = r0.x - px;
            float dy[PPL], alpha[PPL];
            GaussEval e[PPL];
            bool valid[PPL];
            unsigned long long any = 0;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                dy[p] = r0.y - py[p];
                e[p] = eval_gauss(r0.z, r0.w, r1.x, opac, dx, dy[p]);
                alpha[p] = fminf(kAlphaMax, e[p].alpha_raw);
                valid[p] = idx <= bin_final[p] && e[p].sigma >= 0.f && alpha[p] >= kAlphaMin;
                any |= __ballot(valid[p]);
            }
            if (any == 0) continue;
            float col[D];
#pragma unroll
            for (int k = 0; k < D; ++k) col[k] = s_rec[t * REC + 6 + k];
            float gv[4 * NR];
#pragma unroll
            for (int k = 0; k < 4 * NR; ++k) gv[k] = 0.f;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                if (valid[p]) {
                    const float ra = __builtin_amdgcn_rcpf(1.0f - alpha[p]);
                    T[p] *= ra;
                    const float fac = alpha[p] * T[p];
                    float v_alpha = Tf_va[p] * ra;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        gv[8 + k] += fac * vr[p][k];
                        v_alpha += (col[k] * T[p] - buf[p][k] * ra) * vr[p][k];
                        buf[p][k] += col[k] * fac;
                    }
                    if (e[p].alpha_raw <= kAlphaMax) {
                        const float v_sigma = -e[p].alpha_raw * v_alpha;
                        const float hs = 0.5f * v_sigma;
                        gv[4] += hs * dx * dx;
                        gv[5] += v_sigma * dx * dy[p];
                        gv[6] += hs * dy[p] * dy[p];
                        const float vx = v_sigma * e[p].u, vy = v_sigma * e[p].w;
                        gv[0] += vx; gv[1] += vy;
                        gv[2] += fabsf(vx); gv[3] += fabsf(vy);
                        gv[7] += e[p].vis * v_alpha;
                    }
                }
            }
            float red[NR];
            wave_reduce_x4<NR>(gv, red);
            float val = red[0];
#pragma unroll
            for (int i = 1; i < NR; ++i) val = (a_col == i) ? red[i] : val;
            if (a_base) unsafeAtomicAdd(a_base + (int64_t)s_id[t] * a_stride, val);
        }
    }
}

template <int D, int PPL>
int launch_fwd(int C, const float *means2d, const float *conics, const float *colors,
               const float *opacities, const float *backgrounds, int W, int H, int tw, int th,
               const int32_t *offsets, const int32_t *flatten_ids, int64_t M, float *render,
               float *alphas, int32_t *last_ids, const int32_t *order, hipStream_t st) {
    const int64_t total = (int64_t)C * tw * th;
    const unsigned grid = (unsigned)total;
    blend_fwd_kernel<D, PPL><<<grid, 256 / PPL, 0, st>>>(C, means2d, conics, colors, opacities,
                                                         backgrounds, W, H, tw, th, offsets,
                                                         flatten_ids, M, render, alphas, last_ids, order);
    return 0;
}

template <int D, int PPL>
int launch_bwd(int C, const float *means2d, const float *conics, const float *colors,
               const float *opacities, const float *backgrounds, int W, int H, int tw, int th,
               const int32_t *offsets, const int32_t *flatten_ids, int64_t M, const float *alphas,
               const int32_t *last_ids, const float *v_render, const float *v_alphas,
               float *v_means2d, float *v_means2d_abs, float *v_conics, float *v_colors,
               float *v_opacities, const int32_t *order, hipStream_t st) {
    const int64_t total = (int64_t)C * tw * th;
    const unsigned grid = (unsigned)total;
    blend_bwd_kernel<D, PPL><<<grid, 256 / PPL, 0, st>>>(
        C, means2d, conics, colors, opacities, backgrounds, W, H, tw, th, offsets, flatten_ids, M,
        alphas, last_ids, v_render, v_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, order);
    return 0;
}

// Counting sort of the tiles by decreasing list length (bucket width 4, single workgroup).
constexpr int SCHED_THREADS = 1024, SCHED_BUCKETS = 1024;
__global__ __launch_bounds__(SCHED_THREADS) void tile_schedule_kernel(int total_tiles, const int32_t *__restrict__ offsets,
                                                                     int64_t M, int32_t *__restrict__ order) {
    __shared__ int hist[SCHED_BUCKETS];
    __shared__ int wsum[SCHED_THREADS / 64];
    const int tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    auto bucket_of = [&](int t) {
        const int64_t end = (t == total_tiles - 1) ? M : (int64_t)offsets[t + 1];
        const int len = (int)(end - offsets[t]);
        return SCHED_BUCKETS - 1 - min(len >> 2, SCHED_BUCKETS - 1);  // descending
    };
    for (int t = tid; t < total_tiles; t += SCHED_THREADS) atomicAdd(&hist[bucket_of(t)], 1);
    __syncthreads();
    // exclusive scan of hist (one bucket per thread)
    const int v = hist[tid];
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += up;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    __syncthreads();
    hist[tid] = base + inc - v;
    __syncthreads();
    for (int t = tid; t < total_tiles; t += SCHED_THREADS) order[atomicAdd(&hist[bucket_of(t)], 1)] = t;
}

bool supported_channels(int D) { return (D >= 1 && D <= 8) || D == 16 || D == 32; }

}  // namespace

#define MTGS_DISPATCH_D(FN, ...)                         \
    switch (D) {                                         \
        case 1: FN<1, 4>(__VA_ARGS__); break;            \
        case 2: FN<2, 4>(__VA_ARGS__); break;            \
        case 3: FN<3, 4>(__VA_ARGS__); break;            \
        case 4: FN<4, 4>(__VA_ARGS__); break;            \
        case 5: FN<5, 4>(__VA_ARGS__); break;            \
        case 6: FN<6, 4>(__VA_ARGS__); break;            \
        case 7: FN<7, 4>(__VA_ARGS__); break;            \
        case 8: FN<8, 4>(__VA_ARGS__); break;            \
        case 16: FN<16, 1>(__VA_ARGS__); break;          \
        default: FN<32, 1>(__VA_ARGS__); break;          \
    }

extern "C" int mtgs_blend_fwd(int C, int64_t N, int D, const float *means2d, const float *conics,
                              const float *colors, const float *opacities, const float *backgrounds,
                              int width, int height, int tile_size, int tile_w, int tile_h,
                              const int32_t *offsets, const int32_t *flatten_ids, int64_t M,
                              float *render, float *alphas, int32_t *last_ids,
                              const int32_t *tile_order, void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && M >= 0 && width > 0 && height > 0, MTGS_EINVAL, "mtgs_blend_fwd: bad sizes");
    MTGS_REQUIRE(tile_size == MTGS_TILE_SIZE, MTGS_EUNSUPPORTED, "mtgs_blend_fwd: tile_size=%d (only 16 is implemented)", tile_size);
    MTGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16, MTGS_EINVAL,
                 "mtgs_blend_fwd: tile grid %dx%d does not match image %dx%d", tile_w, tile_h, width, height);
    MTGS_REQUIRE(supported_channels(D), MTGS_EUNSUPPORTED,
                 "mtgs_blend_fwd: D=%d channels (supported: 1..8, 16, 32; pad on the host)", D);
    if (C == 0) return MTGS_OK;
    MTGS_REQUIRE(offsets && render && alphas && last_ids && (M == 0 || (means2d && conics && colors && opacities && flatten_ids)),
                 MTGS_EINVAL, "mtgs_blend_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    MTGS_DISPATCH_D(launch_fwd, C, means2d, conics, colors, opacities, backgrounds, width, height,
                    tile_w, tile_h, offsets, flatten_ids, M, render, alphas, last_ids, tile_order, st);
    MTGS_CHECK_LAUNCH("mtgs_blend_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_blend_bwd(int C, int64_t N, int D, const float *means2d, const float *conics,
                              const float *colors, const float *opacities, const float *backgrounds,
                              int width, int height, int tile_size, int tile_w, int tile_h,
                              const int32_t *offsets, const int32_t *flatten_ids, int64_t M,
                              const float *alphas, const int32_t *last_ids, const float *v_render,
                              const float *v_alphas, float *v_means2d, float *v_means2d_abs,
                              float *v_conics, float *v_colors, float *v_opacities,
                              const int32_t *tile_order, void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && M >= 0 && width > 0 && height > 0, MTGS_EINVAL, "mtgs_blend_bwd: bad sizes");
    MTGS_REQUIRE(tile_size == MTGS_TILE_SIZE, MTGS_EUNSUPPORTED, "mtgs_blend_bwd: tile_size=%d (only 16 is implemented)", tile_size);
    MTGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16, MTGS_EINVAL,
                 "mtgs_blend_bwd: tile grid %dx%d does not match image %dx%d", tile_w, tile_h, width, height);
    MTGS_REQUIRE(supported_channels(D), MTGS_EUNSUPPORTED,
                 "mtgs_blend_bwd: D=%d channels (supported: 1..8, 16, 32; pad on the host)", D);
    if (C == 0 || M == 0) return MTGS_OK;
    MTGS_REQUIRE(means2d && conics && colors && opacities && offsets && flatten_ids && alphas &&
                     last_ids && v_render && v_alphas && v_means2d && v_conics && v_colors && v_opacities,
                 MTGS_EINVAL, "mtgs_blend_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    MTGS_DISPATCH_D(launch_bwd, C, means2d, conics, colors, opacities, backgrounds, width, height,
                    tile_w, tile_h, offsets, flatten_ids, M, alphas, last_ids, v_render, v_alphas,
                    v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, tile_order, st);
    MTGS_CHECK_LAUNCH("mtgs_blend_bwd");
    return MTGS_OK;
}

extern "C" int mtgs_tile_schedule(int C, int tile_w, int tile_h, const int32_t *offsets, int64_t M,
                                  int32_t *tile_order, void *stream) {
    MTGS_REQUIRE(C >= 0 && tile_w > 0 && tile_h > 0 && M >= 0, MTGS_EINVAL, "mtgs_tile_schedule: bad sizes");
    const int64_t total = (int64_t)C * tile_w * tile_h;
    if (total == 0) return MTGS_OK;
    MTGS_REQUIRE(total < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_tile_schedule: too many tiles");
    MTGS_REQUIRE(offsets && tile_order, MTGS_EINVAL, "mtgs_tile_schedule: null pointer");
    tile_schedule_kernel<<<1, SCHED_THREADS, 0, (hipStream_t)stream>>>((int)total, offsets, M, tile_order);
    MTGS_CHECK_LAUNCH("mtgs_tile_schedule");
    return MTGS_OK;
}
