// radix_sort.hpp -- hand-written stable LSD radix sort of (key, value) pairs for gfx950.
//
// Replaces the cub::DeviceRadixSort::SortPairs call of gsplat 1.4.0's isect_tiles.  One pass =
//   hist    : per-block digit histogram (LDS atomics)            -> g_hist[digit][block]
//   scan    : one workgroup per digit, exclusive scan over blocks -> g_hist in place, digit totals
//   reorder : every block re-reads its tile, ranks its keys STABLY and scatters keys + values.
// Stable ranking inside a block is wave64-native: a wave walks its sub-tile 64 keys at a time; the
// set of lanes holding the same digit ("peers") is built from RADIX_BITS ballots, the rank is
// popcount(peers below me) + a wave-private running counter in LDS, and the four waves' sub-tiles
// are chained by a per-digit prefix over waves.  No inter-workgroup synchronisation inside a launch
// (nothing to deadlock), no atomics on global memory.  The tile is rebuilt in LDS grouped by digit
// before it is written, so consecutive lanes store consecutive elements of a digit run (coalesced).
//
// Roofline: HBM.  Bytes per pass: hist n*sizeof(K) read; reorder n*(sizeof(K)+4) read + written.
#pragma once
#include "common.hpp"

namespace mtgs_sort {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

template <typename K>
__device__ __forceinline__ unsigned digit_of(K key, int shift, unsigned mask) {
    return (unsigned)(key >> shift) & mask;
}

// n_dev (nullable): the element count lives in DEVICE memory (the host sized buffers and grids for a
// capacity `n` before it knew the count); the kernels then work on min(n, *n_dev) elements.
__device__ __forceinline__ int64_t effective_n(int64_t n, const int64_t *n_dev) {
    if (n_dev) {
        const int64_t d = *n_dev;
        if (d < n) n = d;
    }
    return n;
}

template <typename K, int ITEMS>
__global__ __launch_bounds__(THREADS) void hist_kernel(int64_t n, const K *__restrict__ keys, int shift,
                                                       unsigned mask, int nblocks,
                                                       uint32_t *__restrict__ g_hist, const int64_t *__restrict__ n_dev = nullptr) {
    __shared__ uint32_t s_hist[RADIX];
    const int tid = threadIdx.x;
    s_hist[tid] = 0;
    __syncthreads();
    n = effective_n(n, n_dev);
    const int64_t base = (int64_t)blockIdx.x * (THREADS * ITEMS);
    K k[ITEMS];   // all loads first: one memory round trip per block
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t j = base + (int64_t)i * THREADS + tid;
        k[i] = j < n ? keys[j] : (K)0;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (base + (int64_t)i * THREADS + tid < n) atomicAdd(&s_hist[digit_of(k[i], shift, mask)], 1u);
    __syncthreads();
    g_hist[(int64_t)tid * nblocks + blockIdx.x] = s_hist[tid];
}

// one workgroup per digit: exclusive scan of g_hist[digit][0..nblocks) in place; total -> totals[digit]
template <int DUMMY = 0>
__global__ __launch_bounds__(THREADS) void scan_kernel(int nblocks, uint32_t *__restrict__ g_hist,
                                                       uint32_t *__restrict__ totals) {
    __shared__ uint32_t s_w[WAVES];
    uint32_t *row = g_hist + (int64_t)blockIdx.x * nblocks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += THREADS) {
        const int j = b0 + tid;
        const uint32_t v = j < nblocks ? row[j] : 0;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t t = s_w[w];
            if (w < wave) wbase += t;
            tot += t;
        }
        if (j < nblocks) row[j] = carry + wbase + inc - v;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry;
}

// Optional epilogue of the LAST pass: gather(value) fetches what the outputs need (issued for four elements before the
// first store), store(dst, key, value, gathered) writes them INSTEAD of the key / value stores (bin.hip: gsplat's
// isect_ids).
struct NoEpilogue {
    static constexpr bool enabled = false;
    struct G {};
    __device__ __forceinline__ G gather(int32_t) const { return G{}; }
    __device__ __forceinline__ void store(uint32_t, uint64_t, int32_t, const G &) const {}
};

template <typename K, int ITEMS, class Epi>
__global__ __launch_bounds__(THREADS) void reorder_kernel(int64_t n, const K *__restrict__ keys_in,
                                                          const int32_t *__restrict__ vals_in,
                                                          K *__restrict__ keys_out, int32_t *__restrict__ vals_out,
                                                          int shift, unsigned mask, int bits, int nblocks,
                                                          const uint32_t *__restrict__ g_hist,
                                                          const uint32_t *__restrict__ totals, Epi epi,
                                                          const int64_t *__restrict__ n_dev = nullptr) {
    constexpr int TILE = THREADS * ITEMS;
    __shared__ uint32_t s_off[WAVES][RADIX];  // per-wave digit counts, then running LOCAL offsets
    __shared__ uint32_t s_gbase[RADIX];       // global position of local slot 0 of each digit
    __shared__ uint32_t s_w[WAVES];
    __shared__ K s_key[TILE];                 // the tile, reordered by digit (stable)
    __shared__ int32_t s_val[TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s_off[w][tid] = 0;
    __syncthreads();
    // ---- load this wave's sub-tile (order: wave, iteration, lane == increasing index) + count digits
    n = effective_n(n, n_dev);
    const int64_t tile_base = (int64_t)blockIdx.x * TILE;
    if (tile_base >= n) return;   // (block-uniform; capacity-sized grids)
    const int64_t wbase = tile_base + (int64_t)wave * (64 * ITEMS);
    K key[ITEMS];
    int32_t val[ITEMS];
    unsigned dig[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {   // every load of the tile is issued before the first is used
        const int64_t j = wbase + i * 64 + lane;
        key[i] = j < n ? keys_in[j] : (K)0;
        val[i] = j < n ? vals_in[j] : 0;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        dig[i] = digit_of(key[i], shift, mask);
        if (wbase + i * 64 + lane < n) atomicAdd(&s_off[wave][dig[i]], 1u);
        else dig[i] = 0;
    }
    __syncthreads();
    // ---- per digit (one thread each): count in this block, local exclusive offset, global base
    {
        uint32_t cnt = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) cnt += s_off[w][tid];
        // exclusive scan of the block's digit counts -> local start of every digit
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        // exclusive scan of the GLOBAL digit totals
        const uint32_t tot = totals[tid];
        uint32_t ginc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(ginc, o, 64);
            if (lane >= o) ginc += up;
        }
        if (lane == 63) { s_w[wave] = inc; s_gbase[wave] = ginc; }  // s_gbase[0..3] used as scratch
        __syncthreads();
        uint32_t lb = 0, gb = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
            if (w < wave) { lb += s_w[w]; gb += s_gbase[w]; }
        __syncthreads();
        const uint32_t local_start = lb + inc - cnt;
        // global position of the first element of this digit coming from this block
        s_gbase[tid] = gb + ginc - tot + g_hist[(int64_t)tid * nblocks + blockIdx.x] - local_start;
        uint32_t running = local_start;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = s_off[w][tid];
            s_off[w][tid] = running;
            running += c;
        }
    }
    __syncthreads();
    // ---- stable LOCAL rank, 64 keys per step; the tile is rebuilt in LDS grouped by digit
    uint32_t *my_off = s_off[wave];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t j = wbase + i * 64 + lane;
        const bool live = j < n;
        unsigned long long peers = __ballot(live);
        for (int b = 0; b < bits; ++b) {
            const bool bit = (dig[i] >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t below = (uint32_t)__builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
        uint32_t pos = 0;
        if (live) pos = my_off[dig[i]] + below;
        __builtin_amdgcn_wave_barrier();
        if (live && below == 0) my_off[dig[i]] += (uint32_t)__popcll(peers);  // one lane per distinct digit
        __builtin_amdgcn_wave_barrier();
        if (live) {
            s_key[pos] = key[i];
            s_val[pos] = val[i];
        }
    }
    __syncthreads();
    // ---- write out: consecutive threads hold consecutive elements of a digit run -> coalesced runs
    const int count = (int)min((int64_t)TILE, n - tile_base);
    constexpr int WB = 4;
    for (int k0 = tid; k0 < count; k0 += WB * THREADS) {
        K kk[WB];
        int32_t vv[WB];
        typename Epi::G gg[WB];
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int k = k0 + u * THREADS;
            kk[u] = k < count ? s_key[k] : (K)0;
            vv[u] = k < count ? s_val[k] : 0;
        }
        if (Epi::enabled) {
#pragma unroll
            for (int u = 0; u < WB; ++u)
                if (k0 + u * THREADS < count) gg[u] = epi.gather(vv[u]);
        }
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int k = k0 + u * THREADS;
            if (k < count) {
                const uint32_t dst = s_gbase[digit_of(kk[u], shift, mask)] + (uint32_t)k;
                if (Epi::enabled) {
                    epi.store(dst, (uint64_t)kk[u], vv[u], gg[u]);
                } else {
                    keys_out[dst] = kk[u];
                    vals_out[dst] = vv[u];
                }
            }
        }
    }
}

inline int64_t tile_items(int64_t n) { return n <= (1 << 20) ? THREADS * 4 : THREADS * 16; }
inline int64_t num_blocks(int64_t n) { return ceil_div64(n, tile_items(n)); }
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
// workspace: g_hist[RADIX][nblocks] + totals[RADIX] (uint32), then one (key, value) ping-pong buffer
template <typename K>
inline size_t workspace_bytes(int64_t n) {
    return align256((size_t)(RADIX * (num_blocks(n) + 1)) * sizeof(uint32_t)) + align256((size_t)n * sizeof(K)) +
           align256((size_t)n * sizeof(int32_t));
}

// Sorts on key bits [0, key_bits): result in (keys_out, vals_out); the inputs are only read.
// Passes ping-pong between the output pair and a temporary pair in the workspace, starting on the
// side that makes the LAST pass land in the output pair.
template <typename K, class Epi = NoEpilogue>
int sort_pairs(int64_t n, int key_bits, const K *keys_in, const int32_t *vals_in, K *keys_out, int32_t *vals_out,
               void *ws, size_t ws_bytes, hipStream_t st, const char *who, Epi epi = Epi(),
               const int64_t *n_dev = nullptr) {
    if (n == 0) return MTGS_OK;
    MTGS_REQUIRE(ws_bytes >= workspace_bytes<K>(n), MTGS_EWORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes,
                 workspace_bytes<K>(n));
    MTGS_REQUIRE(n < ((int64_t)1 << 32), MTGS_EINVAL, "%s: n must fit 32 bits", who);
    const int npass = (key_bits + RADIX_BITS - 1) / RADIX_BITS;
    const int nblocks = (int)num_blocks(n);
    const bool small = tile_items(n) == THREADS * 4;
    char *w = (char *)ws;
    uint32_t *g_hist = (uint32_t *)w;
    uint32_t *totals = g_hist + (size_t)RADIX * nblocks;
    w += align256((size_t)(RADIX * (nblocks + 1)) * sizeof(uint32_t));
    K *keys_tmp = (K *)w;
    w += align256((size_t)n * sizeof(K));
    int32_t *vals_tmp = (int32_t *)w;
    const K *kin = keys_in;
    const int32_t *vin = vals_in;
    bool to_out = (npass % 2) == 1;  // odd: in->out->tmp->out ; even: in->tmp->out
    int shift = 0;
    for (int p = 0; p < npass; ++p) {
        // the remaining bits are spread evenly over the remaining passes (13 tile bits: 7 + 6, not 8 + 5): fewer
        // digits in a pass mean longer per-digit runs in a block's tile, i.e. longer contiguous stores
        const int bits = (key_bits - shift + (npass - p) - 1) / (npass - p);
        const unsigned mask = (1u << bits) - 1u;
        K *kout = to_out ? keys_out : keys_tmp;
        int32_t *vout = to_out ? vals_out : vals_tmp;
        const bool last = p == npass - 1;
        if (small) {
            hist_kernel<K, 4><<<nblocks, THREADS, 0, st>>>(n, kin, shift, mask, nblocks, g_hist, n_dev);
            scan_kernel<0><<<RADIX, THREADS, 0, st>>>(nblocks, g_hist, totals);
            if (last && Epi::enabled)
                reorder_kernel<K, 4, Epi><<<nblocks, THREADS, 0, st>>>(n, kin, vin, kout, vout, shift, mask, bits,
                                                                      nblocks, g_hist, totals, epi, n_dev);
            else
                reorder_kernel<K, 4, NoEpilogue><<<nblocks, THREADS, 0, st>>>(n, kin, vin, kout, vout, shift, mask, bits,
                                                                             nblocks, g_hist, totals, NoEpilogue(), n_dev);
        } else {
            hist_kernel<K, 16><<<nblocks, THREADS, 0, st>>>(n, kin, shift, mask, nblocks, g_hist, n_dev);
            scan_kernel<0><<<RADIX, THREADS, 0, st>>>(nblocks, g_hist, totals);
            if (last && Epi::enabled)
                reorder_kernel<K, 16, Epi><<<nblocks, THREADS, 0, st>>>(n, kin, vin, kout, vout, shift, mask, bits,
                                                                       nblocks, g_hist, totals, epi, n_dev);
            else
                reorder_kernel<K, 16, NoEpilogue><<<nblocks, THREADS, 0, st>>>(n, kin, vin, kout, vout, shift, mask, bits,
                                                                              nblocks, g_hist, totals, NoEpilogue(), n_dev);
        }
        shift += bits;
        kin = kout;
        vin = vout;
        to_out = !to_out;
    }
    hipError_t e = hipGetLastError();
    MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "%s: launch failed: %s", who, hipGetErrorString(e));
    return MTGS_OK;
}

}  // namespace mtgs_sort
