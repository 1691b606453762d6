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
//  * Small images (fewer than ~6000 tiles; MTGS trains at 960x540 = 2040 tiles) cannot fill the chip
//    with one wave per tile: they run 2 or 4 waves per tile (2 / 1 pixels per lane), same code.
//    Wide channel counts (D > 8) always use 1 pixel per lane.
//
// Roofline: the kernels are VALU bound (about 25 / 70 flops per pixel x Gaussian pair, fwd / bwd)
// -- MFMA is deliberately unused, there is no dense contraction.  Algorithmic HBM bytes:
//   fwd: M*(4 + 24 + 4D) gathered attributes + P*(4D + 8) written
//   bwd: P*(4D + 12) read + M*(4 + 24 + 4D) gathered + N_vis*(24 + 4D (+8 absgrad)) accumulated
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "wave_reduce.hpp"
#include "raster_rec.hpp"

// Development hooks (candidate / slot counters of the backward, the MTGS_PPL override of kbench.py's sweeps, the per-tile timeline
// of round 6 and its ablation switches) live in dev/blend_dev.hpp and exist in VARIANT builds only
// (scripts/build_variant.py NAME -DMTGS_DEV [-DMTGS_COUNT | -DMTGS_TIMELINE | -DMTGS_ABL_NOREDUCE ...]): the product library
// compiles the hook sites below to nothing.
#ifdef MTGS_DEV
#include "dev/blend_dev.hpp"
#else
#define MTGS_COUNT_ADD(i, v) do { } while (0)
#define MTGS_COUNT_SLOTS(vmask) do { } while (0)
#define MTGS_DEV_PPL_OVERRIDE() do { } while (0)
#define MTGS_TL_DECL() do { } while (0)
#define MTGS_TL_STAGED(n) do { } while (0)
#define MTGS_TL_ACTIVE() do { } while (0)
#define MTGS_TL_END(kind) do { } while (0)
#endif

namespace {

constexpr float kAlphaMax = MTGS_ALPHA_MAX;
constexpr float kTMin = MTGS_T_MIN;
// alpha = opacity * exp(-sigma) <= opacity (sigma >= 0): below this opacity the min(0.999, .) of the reference can never bind,
// and a batch of candidates without such a Gaussian runs the loop without the clamp and its gradient mask
constexpr float kNoClampOpacity = 0.9989f;
constexpr float kHalfLog2e = MTGS_HALF_LOG2E;   // exp(-s2/2) = exp2(-s2 * log2(e)/2)

// LDS record per staged Gaussian, in floats:  x y a b | c opac s2max idx | col[D] (padded to x4)
//   a b c   = conic x log2(e)/2,  s2max = 2 ln(255 opac) x log2(e)/2  (alpha >= 1/255  <=>  s2 <= s2max; alpha = opac exp2(-s2)),
//   idx     = position in the sorted intersection list (int bits).
template <int D>
struct Rec {
    static constexpr int N = ((8 + D + 3) / 4) * 4;
};

// Block -> tile mapping.  `order` (from mtgs_tile_schedule) lists tiles by decreasing work so the
// hardware dispatcher, which starts workgroups in blockIdx order as slots free up, runs the
// longest tiles first (LPT scheduling) -- measured: balance matters more than keeping
// neighbouring tiles on one XCD's L2 (contiguous per-XCD bands were 20 % slower).
__device__ __forceinline__ int64_t block_to_tile(const int32_t *__restrict__ order) {
    return order ? (int64_t)order[blockIdx.x] : (int64_t)blockIdx.x;
}

__device__ __forceinline__ int lanes_below(unsigned long long m) {  // popcount of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Stage up to CAND candidates of the tile's sorted list into LDS, DROPPING those that cannot reach
// alpha >= 1/255 at any pixel centre of the tile, and compacting the survivors in list order
// (wave ballot + mbcnt prefix).  A candidate is dropped only if its {alpha >= 1/255} ellipse (with a
// safety margin) does not reach the rectangle spanned by the tile's pixel centres -- an exact
// ellipse/rectangle test, about 35 % of the 3-sigma-square list at the headline workload -- so every
// dropped candidate would have been skipped pixel by pixel anyway: results are unchanged, but the
// per-pixel loop never sees it.  The binning stage itself keeps gsplat's conservative 3-sigma
// squares, so isect_ids / flatten_ids / offsets stay bit-identical to the reference.
//   FWD: candidate k of the batch is sorted index base + k;  BWD: base - k (back to front).
// Returns the number of survivors (uniform over the workgroup).  With more than one wave per tile
// the per-wave ballots are chained through a small LDS array (two workgroup barriers per round).
// PK (packed input, the fused rasterization path): the list holds RANKS and a candidate is ONE 64-byte record of
// front.hip (raster_rec.hpp) -- xy, conic, opacity, the precomputed s2max and the blended channels in the LDS layout
// -- instead of six gathers from six dense arrays and a log2 per candidate and tile.
template <int D, int NT, int CAND, bool BWD, bool CULL, bool PK>
__device__ __forceinline__ int stage_batch(float *__restrict__ s_rec, int32_t *__restrict__ s_id,
                                           int *__restrict__ s_wc, const float *__restrict__ recs,
                                           const float *__restrict__ means2d,
                                           const float *__restrict__ conics,
                                           const float *__restrict__ colors,
                                           const float *__restrict__ depths, int DC,
                                           const float *__restrict__ opacities,
                                           const int32_t *__restrict__ row_index,
                                           const int32_t (&g)[CAND / NT], int64_t base, int n_cand,
                                           float tile_x0, float tile_y0, bool *may_clamp = nullptr) {
    constexpr int REC = Rec<D>::N;
    const int tid = threadIdx.x;
    int count = 0;
    unsigned long long clamp_any = 0;   // some kept candidate has an opacity whose alpha can reach the 0.999 clamp
#pragma unroll
    for (int r = 0; r < CAND / NT; ++r) {
        const int k = r * NT + tid;
        const bool in_range = k < n_cand && (PK || g[r] >= 0);   // (gather form: a sentinel behind the listed pairs)
        bool keep = in_range;
        float2 xy = make_float2(0.f, 0.f);
        float ca = 0.f, cb = 0.f, cc = 0.f, op = 0.f, s2max = 0.f;
        float4 pk[REC / 4];
        if (in_range) {
            if (PK) {
                static_assert(!PK || REC <= REC_FLOATS, "packed records hold at most 8 channels");
                const float4 *rp = reinterpret_cast<const float4 *>(recs + (int64_t)g[r] * REC_FLOATS);
#pragma unroll
                for (int c = 0; c < REC / 4; ++c) pk[c] = rp[c];
                xy = make_float2(pk[0].x, pk[0].y);
                ca = pk[0].z; cb = pk[0].w; cc = pk[1].x; op = pk[1].y; s2max = pk[1].z;
            } else {
                xy = reinterpret_cast<const float2 *>(means2d)[g[r]];
                ca = conics[(int64_t)g[r] * 3]; cb = conics[(int64_t)g[r] * 3 + 1]; cc = conics[(int64_t)g[r] * 3 + 2];
                op = opacities[g[r]];
                s2max = rec_s2max(op);   // alpha = min(0.999, op e^{-s2/2}) >= 1/255  <=>  s2 <= 2 ln(255 op)
            }
            // opacity < 1/255 (or NaN): alpha >= 1/255 is unreachable.  Dropped here unconditionally -- the
            // per-pixel range test compares bit patterns and relies on s2max >= 0.
            keep = s2max >= 0.f;
            if (CULL && keep)   // the exact ellipse / rectangle test (raster_rec.hpp)
                keep = rec_reaches_rect(ca, cb, cc, s2max, tile_x0 + 0.5f - xy.x, tile_x0 + 15.5f - xy.x, tile_y0 + 0.5f - xy.y,
                                        tile_y0 + 15.5f - xy.y);
        }
        // exp(-sigma) = exp2(-s2 log2(e) / 2): the factor is folded into the STAGED conic and threshold (4 multiplies per
        // candidate and tile here instead of one per pixel slot in every loop below); forward, decision pass and backward
        // stage through this one function, so they evaluate the same expression and take identical per-pixel decisions
        ca *= kHalfLog2e; cb *= kHalfLog2e; cc *= kHalfLog2e; s2max *= kHalfLog2e;
        clamp_any |= __ballot(keep && op > kNoClampOpacity);
        int slot = k;
        if (CULL) {
            const unsigned long long m = __ballot(keep);
            if (NT == 64) {
                slot = count + lanes_below(m);
                count += __popcll(m);
            } else {
                const int wave = tid >> 6;
                if ((tid & 63) == 0) s_wc[wave] = __popcll(m);
                __syncthreads();
                int below = 0, total = 0;
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) {
                    const int c = s_wc[w];
                    if (w < wave) below += c;
                    total += c;
                }
                slot = count + below + lanes_below(m);
                count += total;
                __syncthreads();
            }
        }
        if (keep) {
            float4 *dst = reinterpret_cast<float4 *>(s_rec + slot * REC);
            const float idxf = __int_as_float((int32_t)(BWD ? base - k : base + k));
            if (PK) {
                pk[0].z = ca; pk[0].w = cb; pk[1].x = cc; pk[1].z = s2max;   // (scaled, see above)
                pk[1].w = idxf;   // (the radius is not needed any more)
#pragma unroll
                for (int c = 0; c < REC / 4; ++c) dst[c] = pk[c];
            } else {
                float rec[REC];
                rec[0] = xy.x; rec[1] = xy.y; rec[2] = ca; rec[3] = cb; rec[4] = cc; rec[5] = op; rec[6] = s2max;
                rec[7] = idxf;
#pragma unroll
                for (int c = 0; c < D; ++c) rec[8 + c] = c < DC ? colors[(int64_t)g[r] * DC + c] : depths[g[r]];
#pragma unroll
                for (int c = 8 + D; c < REC; ++c) rec[c] = 0.f;
#pragma unroll
                for (int c = 0; c < REC / 4; ++c) dst[c] = make_float4(rec[4 * c], rec[4 * c + 1], rec[4 * c + 2], rec[4 * c + 3]);
            }
            if (s_id) s_id[slot] = (!PK && row_index) ? row_index[g[r]] : g[r];  // gradient row of this Gaussian
        }
    }
    if (may_clamp) {
        if (NT == 64) {
            *may_clamp = clamp_any != 0;
        } else {   // (the staging barriers above separate the rounds; one more pair for the flag)
            if ((tid & 63) == 0) s_wc[tid >> 6] = clamp_any != 0;
            __syncthreads();
            int f = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) f |= s_wc[w];
            __syncthreads();
            *may_clamp = f != 0;
        }
    }
    return CULL ? count : n_cand;
}

// One Gaussian at one pixel.  With d = mean - pixel centre and the conic (a, b, c):
//   s2 = a dx^2 + 2 b dx dy + c dy^2 = 2 sigma,  evaluated as  q0 + dy (2 b dx + c dy),  q0 = a dx^2
// (dx, q0 and 2 b dx are shared by the pixels of a lane: 3 VALU ops per pixel).  The forward and the backward
// kernel use the same expression, so they take identical per-pixel decisions.
__device__ __forceinline__ float eval_s2(float q0, float b2dx, float c, float dy) {
    return fmaf(dy, fmaf(c, dy, b2dx), q0);
}
// "0 <= s2 <= s2max" as ONE unsigned integer compare of the bit patterns: non-negative floats order like their
// bits, while negative values, NaN and (for finite s2max) +inf all have larger patterns.  Returns the wave's
// lane mask straight from v_cmp (an SGPR pair; no per-lane boolean is materialised).
__device__ __forceinline__ unsigned long long in_range_mask(float s2, float s2max) {
    return __builtin_amdgcn_uicmp(__float_as_uint(s2), __float_as_uint(s2max), 37 /* ICMP_ULE */);
}
// Colour row of staged entry t.  The loads are issued by hand (inline asm ds_read_b128) at the top of the
// entry and waited for right before the any-lane validity branch, so that their LDS latency overlaps the ~20
// VALU ops of the validity test: left to the compiler they sink behind the branch (or next to the wait) and
// every entry stalls on LDS.  The compiler's own lgkmcnt bookkeeping stays conservative with an extra
// in-flight LDS load (the counter only over-counts), and the destination registers are live between the two
// asm statements, so nothing else is allocated to them while the load is in flight.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_offset(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
#ifndef MTGS_EARLY_COLOR_LOAD
#define MTGS_EARLY_COLOR_LOAD 1
#endif
template <int D, int REC>
struct ColorRow {
    static constexpr int NQ = (REC - 8) / 4;
    f32x4 q[NQ];
    // `anchor` is a value the validity test depends on (the mean's x): naming it as an in/out operand keeps the
    // instruction scheduler from moving the load below the test.
    __device__ __forceinline__ void load(const float *s_rec, int t, float &anchor) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (MTGS_EARLY_COLOR_LOAD)
                asm volatile("ds_read_b128 %0, %2" : "=v"(q[i]), "+v"(anchor) : "v"(lds_offset(s_rec + t * REC + 8 + 4 * i)));
            else
                q[i] = *reinterpret_cast<const f32x4 *>(s_rec + t * REC + 8 + 4 * i);
        }
    }
    // the same with the record's LDS address in a VGPR the caller shares between all reads of a pair of entries
    // (one v_mov per pair instead of one per read): byte offset OFF + 32 + 16 i as the instruction's immediate
    template <int OFF>
    __device__ __forceinline__ void load_at(uint32_t base, float &anchor) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            asm volatile("ds_read_b128 %0, %2 offset:%3" : "=v"(q[i]), "+v"(anchor) : "v"(base), "n"(OFF + 32 + 16 * i));
    }
    __device__ __forceinline__ void wait() {
        if (MTGS_EARLY_COLOR_LOAD) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[i]));
        }
    }
    __device__ __forceinline__ float operator[](int k) const { return q[k / 4][k % 4]; }
};
// base + row * stride_bytes with one v_mad_u64_u32 (32 x 32 + 64 bits)
__device__ __forceinline__ float *row_address(float *base, uint32_t row, uint32_t stride_bytes) {
    uint64_t addr, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(addr), "=s"(carry) : "v"(row), "v"(stride_bytes), "v"((uint64_t)base));
    return reinterpret_cast<float *>(addr);
}
// alpha = opacity * exp(-sigma) as a ROUNDED product: the reference rounds alpha before it forms 1 - alpha; left to the
// compiler the product is contracted into 1 - opacity * e (one fma), which moves T by an ulp per Gaussian and doubles the
// number of pixels whose T <= 1e-4 / alpha >= 1/255 decisions differ from the oracle's (measured at the headline size).
__device__ __forceinline__ float alpha_rounded(float opac, float e) {
#pragma clang fp contract(off)
    return opac * e;
}

// ------------------------------------------------------------------------------------------------
#ifndef MTGS_FWD_WAVES
#define MTGS_FWD_WAVES 6
#endif
// A region a compositing kernel clears for the caller while it runs (mtgs_blend_{fwd,bwd}_packed(also_zero)): the kernels are
// VALU-bound and leave HBM ~85 % idle, so every wave writes one slice of zeros when its tile is done -- the tiles finish at very
// different times, which spreads the stores over the kernel's duration.  (The SH backward's dL/dcoeffs, 384 MB of which 94 % are
// zeros at the headline workload: written here it costs the compositing ~15 us, on its own 58 us; a fill on a second stream
// beside the kernel cost more -- queue switches inside a captured graph, and a burst that starved the kernel's own loads.)
struct ZeroFill {
    uint4 *p;
    unsigned long long n16;   // 16-byte words
    uint32_t per_wave;        // 16-byte words per wave (n16 <= per_wave * waves of the launch)
};

template <int NT>
__device__ __forceinline__ void zero_fill_slice(const ZeroFill &zf) {
    if (zf.p == nullptr) return;
    const unsigned long long w0 = ((unsigned long long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * zf.per_wave;
    // NON-TEMPORAL stores: the region is not read again before the kernels behind this one have streamed far more than the caches
    // hold, and the compositing kernel's own working set (records, lists) must stay cached while 400 MB of zeros pass by
    // (same-box A/B of the step: 0.958 -> 0.940 ms; a plain fill kernel measured SLOWER with them, scripts/dev/write_bench.hip)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (uint32_t i = threadIdx.x & 63; i < zf.per_wave; i += 64)
        if (w0 + i < zf.n16) __builtin_nontemporal_store(z, reinterpret_cast<u32x4 *>(zf.p) + w0 + i);
}
// A guard for degenerate frames: with a handful of tiles under a large region (a 16x16 thumbnail of a 2M-Gaussian scene) the
// compositing kernel would become a fill by a few waves -- the region is then cleared by the fill kernel in front of it (zero.hip)
// and the kernel gets none.  Measured at 2M Gaussians x 384 MB: riding wins down to 320x208 (260 tiles, 1.5 MB per tile: step 1.33
// against 1.41 ms with the separate fill; 1920x1080: 0.94 / 0.99, 960x540: 0.74 / 0.80, 480x270: 0.88 / 0.96).
#ifndef MTGS_ZERO_FILL_MAX_KB
#define MTGS_ZERO_FILL_MAX_KB 8192
#endif
constexpr size_t ZERO_FILL_MAX_PER_TILE = (size_t)MTGS_ZERO_FILL_MAX_KB << 10;
extern "C" int mtgs_fill_zero(void *p, size_t bytes, void *stream);
inline int zero_fill_fallback(void *&also_zero, size_t also_zero_bytes, int64_t tiles, hipStream_t st) {
    if (!also_zero || also_zero_bytes == 0 || also_zero_bytes / (size_t)(tiles > 0 ? tiles : 1) <= ZERO_FILL_MAX_PER_TILE) return MTGS_OK;
    void *p = also_zero;
    also_zero = nullptr;
    return mtgs_fill_zero(p, also_zero_bytes, (void *)st);
}
inline ZeroFill make_zero_fill(void *also_zero, size_t also_zero_bytes, unsigned grid, int waves_per_block) {
    ZeroFill zf{nullptr, 0ull, 0u};
    if (also_zero && also_zero_bytes) {
        const unsigned long long n16 = also_zero_bytes / 16, waves = (unsigned long long)grid * waves_per_block;
        zf = ZeroFill{(uint4 *)also_zero, n16, (uint32_t)((n16 + waves - 1) / waves)};
    }
    return zf;
}

template <int D, int PPL, bool PK>
__global__ __launch_bounds__(256 / PPL, (PPL == 4 && D <= 4) ? MTGS_FWD_WAVES : 1) void blend_fwd_kernel(
    int C, const float *__restrict__ recs, const float *__restrict__ means2d, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ backgrounds, const float *__restrict__ depths, int DC, int ed, int W, int H,
    int tw, int th, const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int64_t M,
    float *__restrict__ render, float *__restrict__ alphas, int32_t *__restrict__ last_ids,
    const int32_t *__restrict__ order, const ZeroFill zf) {
    constexpr int NT = 256 / PPL, ROWS = NT / 16, REC = Rec<D>::N;
    constexpr bool CULL = true;
    constexpr int CAND = NT == 64 ? 128 : 256, NR = CAND / NT;
    __shared__ __attribute__((aligned(16))) float s_rec[CAND * REC];
    __shared__ int s_wc[NT / 64];
    MTGS_TL_DECL();
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
    // Finished pixels (T would drop to <= 1e-4, or outside the image) are kept as one WAVE MASK per pixel slot -- an SGPR
    // pair: removing them from a candidate's validity mask is a scalar AND, and a slot's updates are three selects under
    // ONE mask {valid and not stopping} instead of the "alpha = 0 for invalid lanes" select plus four that froze a stopping
    // pixel's weight / T / last index / finished flag (17 -> 15 VALU per slot).
    unsigned long long done[PPL];
    bool inside[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        iy[p] = ty * 16 + ly + p * ROWS;
        py[p] = (float)iy[p] + 0.5f;
        inside[p] = ix < W && iy[p] < H;
        done[p] = __ballot(!inside[p]);
        T[p] = 1.f;
        last[p] = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) acc[p][k] = 0.f;
    }
    const int64_t start = offsets[tile];
    // (packed input: offsets has one more entry, the total -- M is not known when the launch is enqueued)
    const int64_t end = (!PK && tile == total_tiles - 1) ? M : (int64_t)offsets[tile + 1];

    // flatten_ids of the NEXT batch are fetched while the current batch is composited
    int32_t g_next[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) g_next[r] = (start + r * NT + tid < end) ? flatten_ids[start + r * NT + tid] : 0;
    for (int64_t b0 = start; b0 < end; b0 += CAND) {
        bool all_done = true;
#pragma unroll
        for (int p = 0; p < PPL; ++p) all_done = all_done && done[p] == ~0ull;
        if (__syncthreads_and(all_done)) break;
        // gather form: a negative id is the SENTINEL behind the listed pairs of tight lists / capacity-sized tensors
        // (mtgs_bin3_build, MTGS_BIN3_FILL_*): gsplat ends the last tile's range at flatten_ids.numel(), the list ends here
        if (!PK && flatten_ids[b0] < 0) break;
        int32_t g_cur[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) g_cur[r] = g_next[r];
        const int n_cand = (int)min((int64_t)CAND, end - b0);
        bool may_clamp;
        const int bsz = stage_batch<D, NT, CAND, false, CULL, PK>(s_rec, nullptr, s_wc, recs, means2d, conics, colors, depths, DC,
                                                              opacities, nullptr, g_cur, b0, n_cand, (float)(tx * 16),
                                                              (float)(ty * 16), &may_clamp);
        MTGS_TL_STAGED(bsz);
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (b0 + CAND + r * NT + tid < end) g_next[r] = flatten_ids[b0 + CAND + r * NT + tid];
        __syncthreads();
        // One staged entry against the lane's pixels; returns true once every pixel of the wave is finished.
        // OFF / va: the entry's record is at LDS address va + OFF * REC * 4 (one address VGPR per pair of entries);
        // CLAMP = false: no candidate of the batch can reach alpha = 0.999 (stage_batch).
        auto entry = [&](const float4 &r0, const float4 &r1, const uint32_t va, auto off_tag, auto clamp_tag) -> bool {
            constexpr int OFF = decltype(off_tag)::value;
            constexpr bool CLAMP = decltype(clamp_tag)::value;
            ColorRow<D, REC> col;
            col.template load_at<OFF * REC * 4>(va, py[0]);
            const float dx = r0.x - px;
            const float adx = r0.z * dx, bdx = r0.w * dx;
            const float q0 = adx * dx, b2dx = bdx + bdx;
            float s2[PPL];
            unsigned long long vmask[PPL], any = 0;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                s2[p] = eval_s2(q0, b2dx, r1.x, r0.y - py[p]);
                vmask[p] = in_range_mask(s2[p], r1.z) & ~done[p];
                any |= vmask[p];
            }
            col.wait();
            if (any == 0) return false;
            MTGS_TL_ACTIVE();
            const int32_t idx = __float_as_int(r1.w);
            unsigned long long stopped = 0;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                if (vmask[p] != 0) {  // wave-uniform: skip the strips of the tile this Gaussian does not reach
                    // every lane computes (an invalid lane's s2 is anything from a large number to NaN: its alpha is 0,
                    // tiny or NaN and never used); the masks stay wave-uniform values in SGPRs
                    const float e = __builtin_amdgcn_exp2f(-s2[p]);
                    const float alpha = CLAMP ? fminf(kAlphaMax, alpha_rounded(r1.y, e)) : alpha_rounded(r1.y, e);
                    const float w = alpha * T[p];
                    const float next_T = T[p] * (1.f - alpha);     // (the reference's expression: the T <= 1e-4 decision hangs on it)
                    const unsigned long long sm = __builtin_amdgcn_fcmpf(next_T, kTMin, 5 /* FCMP_OLE */) & vmask[p];
                    done[p] |= sm;
                    stopped |= sm;
                    // the Gaussian that would cross T = 1e-4 is not composited: lanes that are invalid or stopping keep
                    // their state (weight 0: colour + 0)
                    const bool upd = __builtin_amdgcn_inverse_ballot_w64(vmask[p] & ~sm);
                    const float wu = upd ? w : 0.f;
#pragma unroll
                    for (int k = 0; k < D; ++k) acc[p][k] = fmaf(col[k], wu, acc[p][k]);
                    T[p] = upd ? next_T : T[p];
                    last[p] = upd ? idx : last[p];
                }
            }
            if (stopped) {
                unsigned long long ad = done[0];
#pragma unroll
                for (int p = 1; p < PPL; ++p) ad &= done[p];
                return ad == ~0ull;
            }
            return false;
        };
        // Entries are consumed two per iteration from two alternating register sets (A, B): the record of the
        // next entry is in flight while the current one is processed, without register copies.
        auto run = [&](auto clamp_tag) {
            using Z = std::integral_constant<int, 0>;
            using O = std::integral_constant<int, 1>;
            const uint32_t rec0 = lds_offset(s_rec);
            auto rec_at = [&](uint32_t va, int q) {
                const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(va + 16 * q));
                return make_float4(v.x, v.y, v.z, v.w);
            };
            uint32_t va;
            asm volatile("v_mov_b32 %0, %1" : "=v"(va) : "s"(rec0));
            float4 ra0 = rec_at(va, 0), ra1 = rec_at(va, 1), rb0 = ra0, rb1 = ra1;
            bool fin = false;     // (a single loop exit: with `break`s out of the entries the register allocator kept two
                                  //  copies of the per-pixel state and moved 24 registers on every skipped candidate)
            for (int t = 0; t < bsz && !fin; t += 2) {
                asm volatile("v_mov_b32 %0, %1" : "=v"(va) : "s"(rec0 + (uint32_t)t * (REC * 4)));
                if (t + 1 < bsz) { rb0 = rec_at(va, REC / 4); rb1 = rec_at(va, REC / 4 + 1); }
                fin = entry(ra0, ra1, va, Z{}, clamp_tag);
                if (t + 1 < bsz && !fin) {
                    if (t + 2 < bsz) { ra0 = rec_at(va, 2 * (REC / 4)); ra1 = rec_at(va, 2 * (REC / 4) + 1); }
                    fin = entry(rb0, rb1, va, O{}, clamp_tag);
                }
            }
        };
        if (may_clamp) run(std::true_type{}); else run(std::false_type{});
    }
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        if (inside[p]) {
            const int64_t pid = ((int64_t)cam * H + iy[p]) * W + ix;
            const float alpha_out = 1.f - T[p];
            alphas[pid] = alpha_out;
            last_ids[pid] = last[p];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                float v = (backgrounds && k < DC) ? acc[p][k] + T[p] * backgrounds[cam * DC + k] : acc[p][k];
                // expected depth ("ED"): accumulated depth / clamp(alpha, min=1e-10)  (gsplat rendering.py)
                if (ed && k == D - 1) v = v / fmaxf(alpha_out, 1e-10f);
                render[pid * D + k] = v;
            }
        }
    }
    zero_fill_slice<NT>(zf);      // (see ZeroFill)
    MTGS_TL_END(0);
}

// ------------------------------------------------------------------------------------------------
// The forward's per-pixel DECISIONS without its colours: which Gaussians does the frame composite FROM?  Same staging, same
// validity test, same alpha / T expressions and the same termination as blend_fwd_kernel (so exactly the entries the forward
// gives a non-zero weight -- and the backward a gradient -- are found), but no colour row is read and nothing is accumulated:
// an entry with at least one updating pixel sets touched[rank] = 1.  In an opaque scene most frustum-visible Gaussians are
// never reached (91-98 % in MTGS-like scenes): everything that is per VISIBLE Gaussian behind the front end -- the optimizer's
// peek of the coefficient rows, the SH evaluation, the normals, the optimizer step -- can then run over the touched ones alone.
// Records are staged without their colour half (32 of 64 bytes gathered).
template <int PPL>
__global__ __launch_bounds__(256 / PPL) void blend_touch_kernel(int C, const float *__restrict__ recs, int W, int H, int tw, int th,
                                                                const int32_t *__restrict__ offsets,
                                                                const int32_t *__restrict__ rank_ids,
                                                                const int32_t *__restrict__ order, uint8_t *__restrict__ touched) {
    constexpr int NT = 256 / PPL, ROWS = NT / 16, REC = Rec<0>::N;
    constexpr int CAND = NT == 64 ? 128 : 256, NR = CAND / NT;
    static_assert(REC == 8, "geometry half of the record");
    __shared__ __attribute__((aligned(16))) float s_rec[CAND * REC];
    __shared__ int32_t s_id[CAND];
    __shared__ int s_wc[NT / 64];
    const int64_t n_tiles = (int64_t)tw * th;
    const int64_t tile = block_to_tile(order);
    const int cam = (int)(tile / n_tiles);
    const int t_in = (int)(tile - (int64_t)cam * n_tiles);
    const int ty = t_in / tw, tx = t_in - ty * tw;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int ix = tx * 16 + lx;
    const float px = (float)ix + 0.5f;
    float py[PPL], T[PPL];
    unsigned long long done[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int iy = ty * 16 + ly + p * ROWS;
        py[p] = (float)iy + 0.5f;
        done[p] = __ballot(!(ix < W && iy < H));
        T[p] = 1.f;
    }
    const int64_t start = offsets[tile], end = offsets[tile + 1];
    for (int64_t b0 = start; b0 < end; b0 += CAND) {
        bool all_done = true;
#pragma unroll
        for (int p = 0; p < PPL; ++p) all_done = all_done && done[p] == ~0ull;
        if (__syncthreads_and(all_done)) break;
        int32_t g_cur[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) g_cur[r] = (b0 + r * NT + tid < end) ? rank_ids[b0 + r * NT + tid] : 0;
        const int n_cand = (int)min((int64_t)CAND, end - b0);
        const int bsz = stage_batch<0, NT, CAND, false, true, true>(s_rec, s_id, s_wc, recs, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                                                                    nullptr, g_cur, b0, n_cand, (float)(tx * 16), (float)(ty * 16));
        __syncthreads();
        bool fin = false;
        for (int t = 0; t < bsz && !fin; ++t) {
            const float4 r0 = *reinterpret_cast<const float4 *>(s_rec + t * REC);
            const float4 r1 = *reinterpret_cast<const float4 *>(s_rec + t * REC + 4);
            const float dx = r0.x - px;
            const float adx = r0.z * dx, bdx = r0.w * dx;
            const float q0 = adx * dx, b2dx = bdx + bdx;
            float s2[PPL];
            unsigned long long vmask[PPL], any = 0;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                s2[p] = eval_s2(q0, b2dx, r1.x, r0.y - py[p]);
                vmask[p] = in_range_mask(s2[p], r1.z) & ~done[p];
                any |= vmask[p];
            }
            if (any == 0) continue;
            unsigned long long stopped = 0, hit = 0;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                if (vmask[p] != 0) {
                    const float e = __builtin_amdgcn_exp2f(-s2[p]);
                    const float alpha = fminf(kAlphaMax, alpha_rounded(r1.y, e));
                    const float next_T = T[p] * (1.f - alpha);
                    const unsigned long long sm = __builtin_amdgcn_fcmpf(next_T, kTMin, 5 /* FCMP_OLE */) & vmask[p];
                    done[p] |= sm;
                    stopped |= sm;
                    const unsigned long long um = vmask[p] & ~sm;
                    hit |= um;
                    const bool upd = __builtin_amdgcn_inverse_ballot_w64(um);
                    T[p] = upd ? next_T : T[p];
                }
            }
            if (hit != 0 && (tid & 63) == 0) touched[s_id[t]] = 1;     // (several waves / tiles may store the same 1)
            if (stopped) {
                unsigned long long ad = done[0];
#pragma unroll
                for (int p = 1; p < PPL; ++p) ad &= done[p];
                fin = ad == ~0ull;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Row strides (bytes) of the six gradient outputs.  Dense gsplat arrays by default; the Python layer passes
// views of ONE interleaved [C,N,16] buffer instead (xy, |xy|, conic, opacity, colour.. in reduction order), so
// that the 12 lanes of an entry's atomic instruction fall into a single 64-byte line: measured 0.06 ns per
// (instruction x line) chip-wide, i.e. 638 us for this kernel's ~2M instructions on six dense arrays against
// 126 us on one line (scripts/dev/atomic_bench.hip) -- the atomics, not VALU, bounded the dense layout.
struct GradRowBytes { uint32_t means2d, means2d_abs, conics, colors, depths, opacities; };

// Gradient components per Gaussian, in reduction order: xy(2) |xy|(2) conic(3) opacity(1) colour(D)
template <int D>
struct GradLayout {
    static constexpr int NV = 8 + D;
    static constexpr int NR = (NV + 3) / 4;
};

// (second launch-bound argument = waves per SIMD the register allocator must leave room for: the
//  one-wave-per-tile mapping is latency-sensitive, 5 waves/SIMD measured better than 4)
#ifndef MTGS_BWD_WAVES
#define MTGS_BWD_WAVES 4
#endif

template <int D, int PPL, bool PK>
__global__ __launch_bounds__(256 / PPL, (PPL == 4 && D <= 4) ? MTGS_BWD_WAVES : 1) void blend_bwd_kernel(
    int C, const float *__restrict__ recs, const float *__restrict__ means2d, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ backgrounds, const float *__restrict__ depths, int DC, int ed, int W, int H,
    int tw, int th, const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids, int64_t M,
    const float *__restrict__ alphas, const int32_t *__restrict__ last_ids, const float *__restrict__ render,
    const float *__restrict__ v_render, const float *__restrict__ v_alphas,
    float *__restrict__ v_means2d, float *__restrict__ v_means2d_abs, float *__restrict__ v_conics,
    float *__restrict__ v_colors, float *__restrict__ v_depths, float *__restrict__ v_opacities,
    const GradRowBytes gs, const int32_t *__restrict__ row_index, const int32_t *__restrict__ order, const ZeroFill zf) {
    constexpr int NT = 256 / PPL, ROWS = NT / 16, REC = Rec<D>::N;
    constexpr bool CULL = true;
    constexpr int CAND = NT == 64 ? 128 : 256, NRD = CAND / NT;
    constexpr int NV = GradLayout<D>::NV, NR = GradLayout<D>::NR;
    MTGS_TL_DECL();
    auto zero_slice = [&]() { zero_fill_slice<NT>(zf); MTGS_TL_END(1); };      // this wave's slice of the caller's region (see ZeroFill)
    __shared__ __attribute__((aligned(16))) float s_rec[CAND * REC];
    __shared__ int32_t s_id[CAND];
    __shared__ int32_t s_max[NT / 64];
    __shared__ int s_wc[NT / 64];
    const int64_t n_tiles = (int64_t)tw * th, total_tiles = (int64_t)C * n_tiles;
    const int64_t tile = block_to_tile(order);
    const int64_t start = offsets[tile];
    const int64_t end = (!PK && tile == total_tiles - 1) ? M : (int64_t)offsets[tile + 1];
    if (end <= start) { zero_slice(); return; }
    const int cam = (int)(tile / n_tiles);
    const int t_in = (int)(tile - (int64_t)cam * n_tiles);
    const int ty = t_in / tw, tx = t_in - ty * tw;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4, lane = tid & 63;
    const int ix = tx * 16 + lx;
    const float px = (float)ix + 0.5f;
    // Per pixel: T (transmittance behind the current Gaussian), vr = dL/d(out), and the scalar
    //   Bq = <colour accumulated behind, vr> - T_final * (dL/dalpha_out - <background, vr>)
    // (the reference keeps the accumulated colour per channel; only its product with vr is ever used).
    float py[PPL], T[PPL], Bq[PPL], vr[PPL][D];
    int32_t bin_final[PPL];
    int32_t my_max = -1;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int iy = ty * 16 + ly + p * ROWS;
        py[p] = (float)iy + 0.5f;
        const bool inside = ix < W && iy < H;
        const int64_t pid = ((int64_t)cam * H + (inside ? iy : 0)) * W + (inside ? ix : 0);
        const float alpha_out = alphas[pid];
        const float T_final = 1.f - alpha_out;
        T[p] = T_final;
        bin_final[p] = inside ? last_ids[pid] : -1;
        my_max = max(my_max, bin_final[p]);
        float bgd = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            vr[p][k] = v_render[pid * D + k];
            if (backgrounds && k < DC) bgd += backgrounds[cam * DC + k] * vr[p][k];
        }
        float va = v_alphas[pid];
        if (ed) {
            // VJP of E = A / clamp(alpha, min=1e-10): v_A = v_E / clamp(alpha), v_alpha += -v_E E / alpha
            const float vE = vr[p][D - 1];
            vr[p][D - 1] = vE / fmaxf(alpha_out, 1e-10f);
            if (alpha_out > 1e-10f) va += -vE * render[pid * D + D - 1] / alpha_out;
        }
        // d(alpha_out)/d(alpha_i) and the background term share the factor T_final / (1 - alpha_i)
        Bq[p] = -T_final * (va - bgd);
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
    if (top < start) { zero_slice(); return; }

    // After the transposed reduction component jr = 4*col + row lives in row `row` of register `col`.
    // One wave (or two) per tile: lane 16*row + col issues the atomic for its component right away.
    // FOUR waves per tile (small images): the reduced values of a whole batch are parked in LDS
    // (s_grad[wave][entry][16]) and flushed once per batch, so the four waves' sums are formed in LDS and
    // the global atomics are quartered (measured 392 -> 325 us at 640x480; with 1-2 waves per tile the
    // extra barrier and LDS traffic cost more than they save).  In the flush, lane L adds component
    // L % 16 of entry L / 16 (+ 16 per step).
    constexpr bool BATCH_FLUSH = NV <= 16 && NT == 256;
    constexpr int NW = NT / 64;
    __shared__ float s_grad[BATCH_FLUSH ? NW * CAND * 16 : 1];
    const int a_col = lane & 15, a_row = lane >> 4;
    // 12 or 16 components: the PACKED reduction leaves component packed_component(lane) in every lane of a quad
    // (wave_reduce.hpp); the first lane of the quad issues the atomic.  Other counts: one register per 4 components.
    constexpr bool PACKED = NR == 3 || NR == 4;
    const int j_red = PACKED ? packed_component(lane) : 4 * a_col + a_row;
    const bool red_lane = PACKED ? (lane & 3) == 0 : a_col < NR;
    const int j = BATCH_FLUSH ? (tid & 15) : j_red;
    float *a_base = nullptr;
    uint32_t a_stride_bytes = 0;
    if ((BATCH_FLUSH || red_lane) && j < NV) {
        if (j < 2) { a_base = v_means2d + j; a_stride_bytes = gs.means2d; }
        else if (j < 4) { a_base = v_means2d_abs ? v_means2d_abs + (j - 2) : nullptr; a_stride_bytes = gs.means2d_abs; }
        else if (j < 7) { a_base = v_conics + (j - 4); a_stride_bytes = gs.conics; }
        else if (j < 8) { a_base = v_opacities; a_stride_bytes = gs.opacities; }
        else if (j - 8 < DC) { a_base = v_colors + (j - 8); a_stride_bytes = gs.colors; }
        else { a_base = v_depths; a_stride_bytes = gs.depths; }
    }

    int32_t g_next[NRD];
#pragma unroll
    for (int r = 0; r < NRD; ++r) g_next[r] = (top - r * NT - tid >= start) ? flatten_ids[top - r * NT - tid] : 0;
    for (int64_t hi = top; hi >= start; hi -= CAND) {
        if (hi != top) __syncthreads();
        int32_t g_cur[NRD];
#pragma unroll
        for (int r = 0; r < NRD; ++r) g_cur[r] = g_next[r];
        const int n_cand = (int)min((int64_t)CAND, hi - start + 1);
        bool may_clamp;
        const int bsz = stage_batch<D, NT, CAND, true, CULL, PK>(s_rec, s_id, s_wc, recs, means2d, conics, colors, depths, DC, opacities,
                                                             row_index, g_cur, hi, n_cand, (float)(tx * 16), (float)(ty * 16),
                                                             &may_clamp);
        MTGS_TL_STAGED(bsz);
#pragma unroll
        for (int r = 0; r < NRD; ++r)
            if (hi - CAND - r * NT - tid >= start) g_next[r] = flatten_ids[hi - CAND - r * NT - tid];
        if (BATCH_FLUSH) {  // zero the batch's gradient slots (entries that touch no pixel never write theirs)
            float4 *z = reinterpret_cast<float4 *>(s_grad);
            for (int e = tid; e < NW * bsz * 4; e += NT) z[(e / (bsz * 4)) * (CAND * 4) + e % (bsz * 4)] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        // One staged entry against the lane's pixels.  OFF = byte offset of the entry's record from the address in `va`
        // (a VGPR shared by the two entries of an iteration: every LDS read of the loop is `va` / `vi` + an immediate),
        // CLAMP = false: no candidate of the batch can reach alpha = 0.999 (stage_batch), so the clamp and the mask of
        // its gradient are compiled out.
        auto entry = [&](const float4 &r0, const float4 &r1, const int t, const uint32_t va, const uint32_t vi, auto off_tag,
                         auto clamp_tag) {
            constexpr int OFF = decltype(off_tag)::value;
            constexpr bool CLAMP = decltype(clamp_tag)::value;
            ColorRow<D, REC> col;
            // (the anchor of the hand-issued loads is the lane's first pixel y: the validity test below depends on it, so
            //  the scheduler cannot sink the loads under the test, and no copy of a record register is needed)
            col.template load_at<OFF * REC * 4>(va, py[0]);
            int32_t gid;  // Gaussian row for the atomics at the end of the entry, fetched the same way
            asm volatile("ds_read_b32 %0, %2 offset:%3" : "=v"(gid), "+v"(py[0]) : "v"(vi), "n"(OFF * 4));
            const int32_t idx = __float_as_int(r1.w);
            const float opac = r1.y;
            const float dx = r0.x - px;
            const float adx = r0.z * dx, bdx = r0.w * dx;
            const float q0 = adx * dx, b2dx = bdx + bdx;
            float dy[PPL], s2[PPL];
            unsigned long long vmask[PPL], any = 0;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                dy[p] = r0.y - py[p];
                s2[p] = eval_s2(q0, b2dx, r1.x, dy[p]);
                // contributes to this pixel: not newer than the pixel's last contributor, and alpha >= 1/255
                vmask[p] = in_range_mask(s2[p], r1.z) & __builtin_amdgcn_sicmp(bin_final[p], idx, 39 /* ICMP_SGE */);
                any |= vmask[p];
            }
            col.wait();
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gid));
            MTGS_COUNT_ADD(0, 1);
            if (any == 0) return;
            MTGS_COUNT_ADD(1, 1);
            MTGS_COUNT_SLOTS(vmask);
            MTGS_TL_ACTIVE();
            float gv[4 * NR];
#pragma unroll
            for (int k = 0; k < 4 * NR; ++k) gv[k] = 0.f;
            // sums over this lane's pixels of v_sigma * {1, dy, dy^2}: dx is common to the lane's pixels
            // (same column), so the conic and mean gradients are recovered from these three numbers.
            // PK (raw rows): the sums are of h = vis * g instead -- v_sigma = -opacity * h -- and the factor -opacity, like
            // the conic map of the position gradient, is applied ONCE per Gaussian by the consumer of the rows
            // (project_bwd.hip, rows_to_gradients): the slot loses the v_opacity accumulation (it IS sum h), the epilogue
            // forms three products instead of nine.
            float S0 = 0.f, S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                // Wave-uniform branch per pixel slot; inside, invalid lanes run with vis = 0, hence alpha = 0
                // (then 1/(1-alpha) == 1 exactly, fac == 0: T and Bq are unchanged and every contribution
                // is 0), so there is no exec-mask divergence in the gradient math (switching the lanes off instead
                // saves the select but costs the explicit zeroing of ten accumulators per entry: measured worse).
                if (vmask[p] != 0) {
                    const bool valid = __builtin_amdgcn_inverse_ballot_w64(vmask[p]);
#if defined(MTGS_DEV) && defined(MTGS_ABL_NOTRANS)       // (ablation: the two transcendentals of a slot replaced by plain VALU)
                    const float vis = valid ? (0.25f - 1e-3f * s2[p]) : 0.f;
#else
                    const float vis = valid ? __builtin_amdgcn_exp2f(-s2[p]) : 0.f;
#endif
                    const float alpha_raw = alpha_rounded(opac, vis);
                    const float alpha = CLAMP ? fminf(kAlphaMax, alpha_raw) : alpha_raw;
#if defined(MTGS_DEV) && defined(MTGS_ABL_NOTRANS)
                    const float ra = 1.0f + alpha;
#else
                    const float ra = __builtin_amdgcn_rcpf(1.0f - alpha);
#endif
                    T[p] *= ra;
                    const float fac = alpha * T[p];
                    float A = 0.f;  // <colour of this Gaussian, vr>
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        gv[8 + k] += fac * vr[p][k];
                        A += col[k] * vr[p][k];
                    }
                    // dL/dalpha_i = T_i <c_i, vr> - (<behind, vr> - T_final (va - <bg, vr>)) / (1 - alpha_i)
                    const float v_alpha = A * T[p] - ra * Bq[p];
                    Bq[p] += fac * A;
                    // alpha clamped at 0.999: no gradient through sigma / opacity (invalid lanes: vis == 0)
                    const float g = (!CLAMP || alpha_raw <= kAlphaMax) ? v_alpha : 0.f;
                    const float v_sigma = PK ? vis * g : -alpha_raw * g;      // (PK: h)
                    const float vsdy = v_sigma * dy[p];
                    S0 += v_sigma;
                    S1 += vsdy;
                    S2 += vsdy * dy[p];
                    // |v_sigma u|, u = a dx + b dy and |v_sigma w|, w = b dx + c dy  (|x| is a source modifier: 2 FMAs each)
                    gv[2] = fmaf(fabsf(v_sigma), fabsf(fmaf(r0.w, dy[p], adx)), gv[2]);
                    gv[3] = fmaf(fabsf(v_sigma), fabsf(fmaf(r1.x, dy[p], bdx)), gv[3]);
                    if (!PK) gv[7] += vis * g;
                }
            }
            if (PK) {
                // RAW MOMENTS of h over the tile: {sum h dx, sum h dy | k sum |h u|, k sum |h w| | sum h dx^2, sum h dx dy, sum h dy^2 | sum h}
                // (k = log2(e)/2 rides on the staged conic: the consumer of the rows divides it out, rows_to_gradients)
                gv[0] = dx * S0;
                gv[1] = S1;
                gv[4] = dx * gv[0];
                gv[5] = dx * S1;
                gv[6] = S2;
                gv[7] = S0;
            } else {
                // sum_p v_sigma u_p with u_p = a dx + b dy_p (and w_p = b dx + c dy_p); conic: 1/2 v_sigma d d^T
                // (the staged conic carries the factor log2(e)/2: taken out of the four sums it appears in)
                gv[0] = (adx * S0 + r0.w * S1) * MTGS_HALF_LOG2E_INV;
                gv[1] = (bdx * S0 + r1.x * S1) * MTGS_HALF_LOG2E_INV;
                gv[2] *= MTGS_HALF_LOG2E_INV;
                gv[3] *= MTGS_HALF_LOG2E_INV;
                gv[4] = 0.5f * dx * dx * S0;
                gv[5] = dx * S1;
                gv[6] = 0.5f * S2;
            }
            float val;
#if defined(MTGS_DEV) && defined(MTGS_ABL_NOREDUCE)      // (round-6 ablation: what the cross-lane reduction costs -- profiles/r06_valu_ceiling.md)
            val = 0.f;
#pragma unroll
            for (int k = 0; k < 4 * NR; ++k) val += gv[k];
            if (false)
#endif
            if constexpr (PACKED) {
                val = wave_reduce_x4_packed<NR>(gv);
            } else {
                float red[NR];
                wave_reduce_x4<NR>(gv, red);
                val = red[0];
#pragma unroll
                for (int i = 1; i < NR; ++i) val = (a_col == i) ? red[i] : val;
            }
            if (BATCH_FLUSH) {
                if (red_lane && j_red < 16) s_grad[((tid >> 6) * CAND + t) * 16 + j_red] = val;
            } else {
                // (what the atomics cost was measured with two experimental builds, since removed: none at all 423 -> 411 us,
                //  plain stores 424 us -- DESIGN.md section 4)
#if defined(MTGS_DEV) && defined(MTGS_ABL_NOATOMIC)
                if (a_base && val == 1.2345e-30f) unsafeAtomicAdd(row_address(a_base, (uint32_t)gid, a_stride_bytes), val);
#else
                if (a_base) unsafeAtomicAdd(row_address(a_base, (uint32_t)gid, a_stride_bytes), val);
#endif
            }
        };
        // Entries are consumed two per iteration from two alternating register sets (A, B): the record of the
        // next entry is in flight while the current one is processed, without register copies.  `va` / `vi` hold the
        // LDS addresses of the pair's first record / row id.
        auto run = [&](auto clamp_tag) {
            using Z = std::integral_constant<int, 0>;
            using O = std::integral_constant<int, 1>;
            const uint32_t rec0 = lds_offset(s_rec), id0 = lds_offset(s_id);
            auto rec_at = [&](uint32_t va, int q) {
                const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(va + 16 * q));
                return make_float4(v.x, v.y, v.z, v.w);
            };
            uint32_t va, vi;
            asm volatile("v_mov_b32 %0, %1" : "=v"(va) : "s"(rec0));
            float4 ra0 = rec_at(va, 0), ra1 = rec_at(va, 1), rb0 = ra0, rb1 = ra1;
            for (int t = 0; t < bsz; t += 2) {
                asm volatile("v_mov_b32 %0, %1" : "=v"(va) : "s"(rec0 + (uint32_t)t * (REC * 4)));
                asm volatile("v_mov_b32 %0, %1" : "=v"(vi) : "s"(id0 + (uint32_t)t * 4));
                if (t + 1 < bsz) { rb0 = rec_at(va, REC / 4); rb1 = rec_at(va, REC / 4 + 1); }
                entry(ra0, ra1, t, va, vi, Z{}, clamp_tag);
                if (t + 1 >= bsz) break;
                if (t + 2 < bsz) { ra0 = rec_at(va, 2 * (REC / 4)); ra1 = rec_at(va, 2 * (REC / 4) + 1); }
                entry(rb0, rb1, t + 1, va, vi, O{}, clamp_tag);
            }
        };
        if (may_clamp) run(std::true_type{}); else run(std::false_type{});
        if (BATCH_FLUSH) {
            __syncthreads();
            for (int e = tid; e < bsz * 16; e += NT) {
                const int t = e >> 4;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) v += s_grad[(w * CAND + t) * 16 + j];
                if (a_base && v != 0.f) unsafeAtomicAdd(row_address(a_base, (uint32_t)s_id[t], a_stride_bytes), v);
            }
        }
    }
    zero_slice();
}

template <int D, int PPL, bool PK = false>
int launch_fwd(int C, const float *recs, const float *means2d, const float *conics, const float *colors,
               const float *opacities, const float *backgrounds, const float *depths, int DC, int ed, int W,
               int H, int tw, int th,
               const int32_t *offsets, const int32_t *flatten_ids, int64_t M, float *render,
               float *alphas, int32_t *last_ids, const int32_t *order, hipStream_t st, void *also_zero = nullptr,
               size_t also_zero_bytes = 0) {
    const int64_t total = (int64_t)C * tw * th;
    const unsigned grid = (unsigned)total;
    blend_fwd_kernel<D, PPL, PK><<<grid, 256 / PPL, 0, st>>>(C, recs, means2d, conics, colors, opacities,
                                                         backgrounds, depths, DC, ed, W, H, tw, th, offsets,
                                                         flatten_ids, M, render, alphas, last_ids, order,
                                                         make_zero_fill(also_zero, also_zero_bytes, grid, 256 / PPL / 64));
    return 0;
}

template <int D, int PPL, bool PK = false>
int launch_bwd(int C, const float *recs, const float *means2d, const float *conics, const float *colors,
               const float *opacities, const float *backgrounds, const float *depths, int DC, int ed, int W,
               int H, int tw, int th, const int32_t *offsets, const int32_t *flatten_ids, int64_t M,
               const float *alphas, const int32_t *last_ids, const float *render, const float *v_render,
               const float *v_alphas, float *v_means2d, float *v_means2d_abs, float *v_conics, float *v_colors,
               float *v_depths, float *v_opacities, const GradRowBytes gs, const int32_t *row_index, const int32_t *order,
               hipStream_t st, void *also_zero = nullptr, size_t also_zero_bytes = 0) {
    const int64_t total = (int64_t)C * tw * th;
    const unsigned grid = (unsigned)total;
    const ZeroFill zf = make_zero_fill(also_zero, also_zero_bytes, grid, 256 / PPL / 64);
    blend_bwd_kernel<D, PPL, PK><<<grid, 256 / PPL, 0, st>>>(
        C, recs, means2d, conics, colors, opacities, backgrounds, depths, DC, ed, W, H, tw, th, offsets, flatten_ids, M,
        alphas, last_ids, render, v_render, v_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_depths,
        v_opacities, gs, row_index, order, zf);
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

// Pixels per lane.  One wave per tile (4 pixels per lane) is the most instruction-efficient
// mapping, but it needs enough tiles to fill 1024 SIMDs several times over; small images (MTGS
// trains at 960x540 = 2040 tiles) get 2 or 4 waves per tile instead.  Thresholds from kbench.py.
static int pick_ppl(int64_t total_tiles, int DT, bool backward) {
    if (DT > 8) return 1;
    MTGS_DEV_PPL_OVERRIDE();
    // measured on MI355X, N = 2M (us, pixels per lane 4 / 2 / 1):
    //   fwd, 4 channels: 1200 tiles 405/291/220   2040 tiles 282/202/159   3600 tiles 237/187/169   8160 tiles 230/-/-
    //   bwd, 4 channels: 1200 tiles 424/344/324   2040 tiles 319/277/408   2800 tiles 320/310/484   8160 tiles 430/-/-
    //   bwd, 7 channels: 1200 tiles  - /389/600   2040 tiles 362/323/804   3600 tiles 349/383/ -    8160 tiles 505/591/-
    // (re-measured after the compact gradient rows and the packed reduction: two waves per tile now win the middle range)
    // round 4, tight tile lists (half the pairs per tile; us, pixels per lane 1 / 2 / 4, packed kernels, median of 12-30):
    //   fwd, 4 channels: 2040 tiles 109/132/193   3600 tiles 124/119/162   8160 tiles 179/152-157/156-158
    //   fwd, 7 channels: 2040 tiles 126/140/193   3600 tiles 142/130/158   8160 tiles 204/197/200
    //   bwd, 4 channels: 2040 tiles 336/230/294   3600 tiles 464/256/263   8160 tiles 795/382/330
    //   bwd, 7 channels: 2040 tiles 691/261/329   3600 tiles 948/298/304   8160 tiles 1640/480/392
    if (backward) {
        if (total_tiles >= 6000) return 4;
        return (DT <= 4 && total_tiles < 1536) ? 1 : 2;
    }
    return total_tiles >= 16000 ? 4 : (total_tiles >= 3000 ? 2 : 1);
}

#define MTGS_DISPATCH_ONE(FN, DD, ...)                                         \
    if (ppl == 4) FN<DD, 4>(__VA_ARGS__);                                      \
    else if (ppl == 2) FN<DD, 2>(__VA_ARGS__);                                 \
    else FN<DD, 1>(__VA_ARGS__);

#define MTGS_DISPATCH_D(FN, ...)                                      \
    switch (DT) {                                                     \
        case 1: MTGS_DISPATCH_ONE(FN, 1, __VA_ARGS__) break;          \
        case 2: MTGS_DISPATCH_ONE(FN, 2, __VA_ARGS__) break;          \
        case 3: MTGS_DISPATCH_ONE(FN, 3, __VA_ARGS__) break;          \
        case 4: MTGS_DISPATCH_ONE(FN, 4, __VA_ARGS__) break;          \
        case 5: MTGS_DISPATCH_ONE(FN, 5, __VA_ARGS__) break;          \
        case 6: MTGS_DISPATCH_ONE(FN, 6, __VA_ARGS__) break;          \
        case 7: MTGS_DISPATCH_ONE(FN, 7, __VA_ARGS__) break;          \
        case 8: MTGS_DISPATCH_ONE(FN, 8, __VA_ARGS__) break;          \
        case 16: FN<16, 1>(__VA_ARGS__); break;                       \
        default: FN<32, 1>(__VA_ARGS__); break;                       \
    }

extern "C" int mtgs_blend_fwd(int C, int64_t N, int D, const float *means2d, const float *conics,
                              const float *colors, const float *opacities, const float *backgrounds,
                              const float *depths, int ed_normalize, int width, int height, int tile_size,
                              int tile_w, int tile_h, const int32_t *offsets, const int32_t *flatten_ids,
                              int64_t M, float *render, float *alphas, int32_t *last_ids,
                              const int32_t *tile_order, void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && M >= 0 && D >= 0 && width > 0 && height > 0, MTGS_EINVAL, "mtgs_blend_fwd: bad sizes");
    MTGS_REQUIRE(tile_size == MTGS_TILE_SIZE, MTGS_EUNSUPPORTED, "mtgs_blend_fwd: tile_size=%d (only 16 is implemented)", tile_size);
    MTGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16, MTGS_EINVAL,
                 "mtgs_blend_fwd: tile grid %dx%d does not match image %dx%d", tile_w, tile_h, width, height);
    const int DT = D + (depths ? 1 : 0);
    MTGS_REQUIRE(supported_channels(DT), MTGS_EUNSUPPORTED,
                 "mtgs_blend_fwd: D=%d channels in total (supported: 1..8, 16, 32; pad on the host)", DT);
    MTGS_REQUIRE(!ed_normalize || depths, MTGS_EINVAL, "mtgs_blend_fwd: ed_normalize needs depths");
    if (C == 0) return MTGS_OK;
    MTGS_REQUIRE(offsets && render && alphas && last_ids &&
                     (M == 0 || (means2d && conics && (colors || D == 0) && opacities && flatten_ids)),
                 MTGS_EINVAL, "mtgs_blend_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int ppl = pick_ppl((int64_t)C * tile_w * tile_h, DT, false);
    MTGS_DISPATCH_D(launch_fwd, C, nullptr, means2d, conics, colors, opacities, backgrounds, depths, D, ed_normalize, width,
                    height, tile_w, tile_h, offsets, flatten_ids, M, render, alphas, last_ids, tile_order, st);
    MTGS_CHECK_LAUNCH("mtgs_blend_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_blend_bwd(int C, int64_t N, int D, const float *means2d, const float *conics,
                              const float *colors, const float *opacities, const float *backgrounds,
                              const float *depths, int ed_normalize, int width, int height, int tile_size,
                              int tile_w, int tile_h, const int32_t *offsets, const int32_t *flatten_ids,
                              int64_t M, const float *alphas, const int32_t *last_ids, const float *render,
                              const float *v_render, const float *v_alphas, float *v_means2d,
                              float *v_means2d_abs, float *v_conics, float *v_colors, float *v_depths,
                              float *v_opacities, const int64_t *grad_row_strides, const int32_t *grad_row_index,
                              const int32_t *tile_order, void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && M >= 0 && D >= 0 && width > 0 && height > 0, MTGS_EINVAL, "mtgs_blend_bwd: bad sizes");
    MTGS_REQUIRE(tile_size == MTGS_TILE_SIZE, MTGS_EUNSUPPORTED, "mtgs_blend_bwd: tile_size=%d (only 16 is implemented)", tile_size);
    MTGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16, MTGS_EINVAL,
                 "mtgs_blend_bwd: tile grid %dx%d does not match image %dx%d", tile_w, tile_h, width, height);
    const int DT = D + (depths ? 1 : 0);
    MTGS_REQUIRE(supported_channels(DT), MTGS_EUNSUPPORTED,
                 "mtgs_blend_bwd: D=%d channels in total (supported: 1..8, 16, 32; pad on the host)", DT);
    MTGS_REQUIRE(!ed_normalize || (depths && render), MTGS_EINVAL, "mtgs_blend_bwd: ed_normalize needs depths and render");
    if (C == 0 || M == 0) return MTGS_OK;
    MTGS_REQUIRE(means2d && conics && (colors || D == 0) && opacities && offsets && flatten_ids && alphas &&
                     last_ids && v_render && v_alphas && v_means2d && v_conics && (v_colors || D == 0) &&
                     (v_depths || !depths) && v_opacities,
                 MTGS_EINVAL, "mtgs_blend_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t dense[6] = {2, 2, 3, D, 1, 1};
    const int64_t min_w[6] = {2, 2, 3, D, 1, 1};
    uint32_t sb[6];
    for (int i = 0; i < 6; ++i) {
        const int64_t rs = grad_row_strides ? grad_row_strides[i] : dense[i];
        MTGS_REQUIRE(rs >= min_w[i] && rs < ((int64_t)1 << 30), MTGS_EINVAL,
                     "mtgs_blend_bwd: grad_row_strides[%d]=%lld (row width %lld)", i, (long long)rs, (long long)min_w[i]);
        sb[i] = (uint32_t)(rs * 4);
    }
    const GradRowBytes gs{sb[0], sb[1], sb[2], sb[3], sb[4], sb[5]};
    const int ppl = pick_ppl((int64_t)C * tile_w * tile_h, DT, true);
    MTGS_DISPATCH_D(launch_bwd, C, nullptr, means2d, conics, colors, opacities, backgrounds, depths, D, ed_normalize, width,
                    height, tile_w, tile_h, offsets, flatten_ids, M, alphas, last_ids, render, v_render, v_alphas,
                    v_means2d, v_means2d_abs, v_conics, v_colors, v_depths, v_opacities, gs, grad_row_index, tile_order,
                    st);
    MTGS_CHECK_LAUNCH("mtgs_blend_bwd");
    return MTGS_OK;
}

// ---- packed input (fused rasterization path): records of front.hip + rank_ids / offsets[T + 1] of bin3.hip ----
#define MTGS_DISPATCH_PK_ONE(FN, DD, ...)                                      \
    if (ppl == 4) FN<DD, 4, true>(__VA_ARGS__);                                \
    else if (ppl == 2) FN<DD, 2, true>(__VA_ARGS__);                           \
    else FN<DD, 1, true>(__VA_ARGS__);

#define MTGS_DISPATCH_PK(FN, ...)                                     \
    switch (DT) {                                                     \
        case 1: MTGS_DISPATCH_PK_ONE(FN, 1, __VA_ARGS__) break;       \
        case 2: MTGS_DISPATCH_PK_ONE(FN, 2, __VA_ARGS__) break;       \
        case 3: MTGS_DISPATCH_PK_ONE(FN, 3, __VA_ARGS__) break;       \
        case 4: MTGS_DISPATCH_PK_ONE(FN, 4, __VA_ARGS__) break;       \
        case 5: MTGS_DISPATCH_PK_ONE(FN, 5, __VA_ARGS__) break;       \
        case 6: MTGS_DISPATCH_PK_ONE(FN, 6, __VA_ARGS__) break;       \
        case 7: MTGS_DISPATCH_PK_ONE(FN, 7, __VA_ARGS__) break;       \
        default: MTGS_DISPATCH_PK_ONE(FN, 8, __VA_ARGS__) break;      \
    }

extern "C" int mtgs_blend_fwd_packed(int C, int D, int with_depth, const float *recs, const float *backgrounds,
                                     int ed_normalize, int width, int height, int tile_w, int tile_h,
                                     const int32_t *offsets, const int32_t *rank_ids, float *render, float *alphas,
                                     int32_t *last_ids, const int32_t *tile_order, void *also_zero, size_t also_zero_bytes,
                                     void *stream) {
    MTGS_REQUIRE(C >= 0 && D >= 0 && width > 0 && height > 0, MTGS_EINVAL, "mtgs_blend_fwd_packed: bad sizes");
    MTGS_REQUIRE(!also_zero || ((reinterpret_cast<uintptr_t>(also_zero) | also_zero_bytes) & 15) == 0, MTGS_EINVAL,
                 "mtgs_blend_fwd_packed: also_zero must be a 16-byte aligned region of whole 16-byte words");
    MTGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16, MTGS_EINVAL,
                 "mtgs_blend_fwd_packed: tile grid %dx%d does not match image %dx%d", tile_w, tile_h, width, height);
    const int DT = D + (with_depth ? 1 : 0);
    MTGS_REQUIRE(DT >= 1 && DT <= REC_MAX_CHANNELS, MTGS_EUNSUPPORTED, "mtgs_blend_fwd_packed: %d blended channels (1..%d)", DT,
                 REC_MAX_CHANNELS);
    MTGS_REQUIRE(!ed_normalize || with_depth, MTGS_EINVAL, "mtgs_blend_fwd_packed: ed_normalize needs the depth channel");
    if (C == 0) return also_zero ? mtgs_zero_async(also_zero, also_zero_bytes, (hipStream_t)stream) : MTGS_OK;
    MTGS_REQUIRE(recs && offsets && rank_ids && render && alphas && last_ids, MTGS_EINVAL, "mtgs_blend_fwd_packed: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = zero_fill_fallback(also_zero, also_zero_bytes, (int64_t)C * tile_w * tile_h, st)) return rc;
    const int ppl = pick_ppl((int64_t)C * tile_w * tile_h, DT, false);
    MTGS_DISPATCH_PK(launch_fwd, C, recs, nullptr, nullptr, nullptr, nullptr, backgrounds, nullptr, D, ed_normalize, width, height,
                     tile_w, tile_h, offsets, rank_ids, (int64_t)-1, render, alphas, last_ids, tile_order, st, also_zero, also_zero_bytes);
    MTGS_CHECK_LAUNCH("mtgs_blend_fwd_packed");
    return MTGS_OK;
}

extern "C" int mtgs_blend_touch_packed(int C, const float *recs, int width, int height, int tile_w, int tile_h, const int32_t *offsets,
                                       const int32_t *rank_ids, const int32_t *tile_order, uint8_t *touched, int64_t cap_vis,
                                       void *stream) {
    MTGS_REQUIRE(C >= 0 && width > 0 && height > 0 && cap_vis >= 0, MTGS_EINVAL, "mtgs_blend_touch_packed: bad sizes");
    MTGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16, MTGS_EINVAL,
                 "mtgs_blend_touch_packed: tile grid %dx%d does not match image %dx%d", tile_w, tile_h, width, height);
    if (C == 0 || cap_vis == 0) return MTGS_OK;
    MTGS_REQUIRE(recs && offsets && rank_ids && touched, MTGS_EINVAL, "mtgs_blend_touch_packed: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(touched, 0, (size_t)cap_vis, st);
    MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_blend_touch_packed: memset failed");
    const int64_t total_tiles = (int64_t)C * tile_w * tile_h;
    const int ppl = pick_ppl(total_tiles, 4, false);
    const unsigned grid = (unsigned)total_tiles;
    if (ppl == 4) blend_touch_kernel<4><<<grid, 64, 0, st>>>(C, recs, width, height, tile_w, tile_h, offsets, rank_ids, tile_order, touched);
    else if (ppl == 2) blend_touch_kernel<2><<<grid, 128, 0, st>>>(C, recs, width, height, tile_w, tile_h, offsets, rank_ids, tile_order, touched);
    else blend_touch_kernel<1><<<grid, 256, 0, st>>>(C, recs, width, height, tile_w, tile_h, offsets, rank_ids, tile_order, touched);
    MTGS_CHECK_LAUNCH("mtgs_blend_touch_packed");
    return MTGS_OK;
}

extern "C" int mtgs_blend_bwd_packed(int C, int D, int with_depth, const float *recs, const float *backgrounds,
                                     int ed_normalize, int width, int height, int tile_w, int tile_h,
                                     const int32_t *offsets, const int32_t *rank_ids, const float *alphas,
                                     const int32_t *last_ids, const float *render, const float *v_render,
                                     const float *v_alphas, float *grad_rows, int64_t row_stride, int absgrad,
                                     const int32_t *tile_order, void *also_zero, size_t also_zero_bytes, void *stream) {
    MTGS_REQUIRE(C >= 0 && D >= 0 && width > 0 && height > 0, MTGS_EINVAL, "mtgs_blend_bwd_packed: bad sizes");
    MTGS_REQUIRE(!also_zero || ((reinterpret_cast<uintptr_t>(also_zero) | also_zero_bytes) & 15) == 0, MTGS_EINVAL,
                 "mtgs_blend_bwd_packed: also_zero must be a 16-byte aligned region of whole 16-byte words");
    MTGS_REQUIRE(also_zero_bytes / 16 < ((size_t)1 << 32) * 64, MTGS_EINVAL, "mtgs_blend_bwd_packed: also_zero too large");
    MTGS_REQUIRE(tile_w == (width + 15) / 16 && tile_h == (height + 15) / 16, MTGS_EINVAL,
                 "mtgs_blend_bwd_packed: tile grid %dx%d does not match image %dx%d", tile_w, tile_h, width, height);
    const int DT = D + (with_depth ? 1 : 0);
    MTGS_REQUIRE(DT >= 1 && DT <= REC_MAX_CHANNELS, MTGS_EUNSUPPORTED, "mtgs_blend_bwd_packed: %d blended channels (1..%d)", DT,
                 REC_MAX_CHANNELS);
    MTGS_REQUIRE(!ed_normalize || (with_depth && render), MTGS_EINVAL, "mtgs_blend_bwd_packed: ed_normalize needs depth and render");
    MTGS_REQUIRE(row_stride >= 8 + DT && row_stride < ((int64_t)1 << 28), MTGS_EINVAL, "mtgs_blend_bwd_packed: row_stride=%lld",
                 (long long)row_stride);
    if (C == 0) return also_zero ? mtgs_zero_async(also_zero, also_zero_bytes, (hipStream_t)stream) : MTGS_OK;
    MTGS_REQUIRE(recs && offsets && rank_ids && alphas && last_ids && v_render && v_alphas && grad_rows, MTGS_EINVAL,
                 "mtgs_blend_bwd_packed: null pointer");
    hipStream_t st = (hipStream_t)stream;
    // rows: [xy 2 | |xy| 2 | conic 3 | opacity 1 | colour D | depth 1 | pad]
    const uint32_t sb = (uint32_t)(row_stride * 4);
    const GradRowBytes gs{sb, sb, sb, sb, sb, sb};
    if (int rc = zero_fill_fallback(also_zero, also_zero_bytes, (int64_t)C * tile_w * tile_h, st)) return rc;
    const int ppl = pick_ppl((int64_t)C * tile_w * tile_h, DT, true);
    MTGS_DISPATCH_PK(launch_bwd, C, recs, nullptr, nullptr, nullptr, nullptr, backgrounds, nullptr, D, ed_normalize, width, height,
                     tile_w, tile_h, offsets, rank_ids, (int64_t)-1, alphas, last_ids, render, v_render, v_alphas, grad_rows,
                     absgrad ? grad_rows + 2 : nullptr, grad_rows + 4, grad_rows + 8, grad_rows + 8 + D, grad_rows + 7, gs, nullptr,
                     tile_order, st, also_zero, also_zero_bytes);
    MTGS_CHECK_LAUNCH("mtgs_blend_bwd_packed");
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
