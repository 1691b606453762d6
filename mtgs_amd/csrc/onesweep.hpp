// onesweep.hpp -- stable LSD radix sort of (key, int32 value) pairs in ONE launch per digit pass (gfx950).
//
// radix_sort.hpp spends three launches per pass (histogram, scan over blocks, reorder); on the 300k-key depth sort
// and the 4M-key tile sort of a frame that is 18 launches of 2-40 us each, i.e. mostly launch boundaries.  Here:
//   * ONE histogram kernel per sort counts the digits of EVERY pass up front (the multiset of keys does not change
//     between passes) -> global digit totals per pass;
//   * a pass is one kernel: a block counts its tile's digits, publishes the counts, obtains the number of equal
//     digits in the blocks in front of it from the other blocks' published counts, then ranks its keys stably
//     (wave64 ballot match, as radix_sort.hpp) and scatters keys + values.  The cross-block prefix is TWO-LEVEL,
//     not a chained look-back: a memory-side round trip costs 1.5-2 us on this chip and with every block of a pass
//     resident at once a chained look-back degenerates into ~sqrt(2 blocks / depth) dependent hops (measured: 20 us
//     per pass over 300k keys, 46 us over 4M).  Blocks are grouped by GROUP consecutive tickets; a block sums the
//     counts of the blocks in front of it INSIDE its group (independent loads), the last block of a group publishes
//     the group total, and every block adds the totals of the groups in front: two dependent hops, whatever the size.
//     One 32-bit {ready, count} word per (block, digit) and per (group, digit); lanes = digits.
//   * blocks take their logical index from an atomic ticket (forward progress, see lookback.hpp); every spin is
//     bounded (error bit instead of a hang);
//   * the number of elements is read from DEVICE memory (sizes[which], clamped to the capacity the grid was sized
//     for), so the host can enqueue a frame's sorts before it knows how many intersections the frame has.
// Stability, digit plan (remaining bits spread evenly over the passes) and the optional last-pass epilogue are those
// of radix_sort.hpp; results are bit-identical (tests/test_gpu_parity.py::test_onesweep_*).
//
// Roofline: HBM.  Per pass n*(sizeof(K)+4) read + written; histogram n*sizeof(K) read once per sort.
#pragma once
#include "common.hpp"

namespace mtgs_os {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int MAX_PASSES = 8;
constexpr uint32_t READY = 1u << 31, VAL_MASK = ~READY;
constexpr int SPIN_LIMIT = 1 << 22;

struct Plan {
    int npass;
    int shift[MAX_PASSES], bits[MAX_PASSES];
};
inline Plan make_plan(int key_bits) {
    Plan p;
    p.npass = (key_bits + RADIX_BITS - 1) / RADIX_BITS;
    int shift = 0;
    for (int i = 0; i < p.npass; ++i) {
        const int bits = (key_bits - shift + (p.npass - i) - 1) / (p.npass - i);
        p.shift[i] = shift; p.bits[i] = bits;
        shift += bits;
    }
    return p;
}

__device__ __forceinline__ uint32_t ld32(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st32(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Element count read from DEVICE memory: mode 0 = *p, mode 1 = *p >> 32 (n_vis of front.hip's packed totals); clamped
// to the capacity the buffers and the grid were sized for.
struct SizeRef {
    const int64_t *p;
    int mode;
    int64_t cap;
};
__device__ __forceinline__ int64_t size_of(const SizeRef r) {
    int64_t n = *r.p;
    if (r.mode == 1) n >>= 32;
    return n < r.cap ? n : r.cap;
}
__device__ __forceinline__ uint64_t ld64(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st64(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr uint64_t READY2 = ((uint64_t)READY << 32) | READY;
// Publishes one row of RADIX {READY | count} words (s_row, LDS) as 128 8-byte stores (threads 0..127).
__device__ __forceinline__ void publish_row(uint32_t *dst_row, const uint32_t *s_row) {
    const int t = threadIdx.x;
    if (t < RADIX / 2) st64(reinterpret_cast<uint64_t *>(dst_row) + t, reinterpret_cast<const uint64_t *>(s_row)[t]);
}
// Sum over `count` published rows rows[(first + i) * RADIX + d], i in [0, count), for every digit d: the rows are
// dealt to the four waves (row i -> wave i % 4), a lane covers 4 digits of a row with two 8-byte loads, and ALL of a
// wave's loads are issued before the first is waited for -- one memory round trip for up to 4 * LB_ROWS rows.  The
// partial sums meet in s_part[WAVES][RADIX]; the caller adds the four after a barrier.  Waits for rows that are not
// published yet (READY bit of every word).
constexpr int LB_ROWS = 16;
__device__ __forceinline__ void sum_rows(const uint32_t *rows, int first, int count, uint32_t (*s_part)[RADIX], uint32_t *err) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc[4] = {0u, 0u, 0u, 0u};
    int spins = 0;
    for (int i0 = wave; i0 < count; i0 += WAVES * LB_ROWS) {
        uint64_t v[LB_ROWS][2];
#pragma unroll
        for (int u = 0; u < LB_ROWS; ++u) {
            const int i = i0 + u * WAVES;
            const uint64_t *p = reinterpret_cast<const uint64_t *>(rows + (int64_t)(first + i) * RADIX) + 2 * lane;
            v[u][0] = i < count ? ld64(p) : READY2;
            v[u][1] = i < count ? ld64(p + 1) : READY2;
        }
#pragma unroll
        for (int u = 0; u < LB_ROWS; ++u) {
            const int i = i0 + u * WAVES;
            while ((v[u][0] & READY2) != READY2 || (v[u][1] & READY2) != READY2) {
                if (++spins > SPIN_LIMIT) { atomicOr(err, 2u); break; }
                __builtin_amdgcn_s_sleep(1);
                const uint64_t *p = reinterpret_cast<const uint64_t *>(rows + (int64_t)(first + i) * RADIX) + 2 * lane;
                v[u][0] = ld64(p);
                v[u][1] = ld64(p + 1);
            }
            acc[0] += (uint32_t)v[u][0] & VAL_MASK; acc[1] += (uint32_t)(v[u][0] >> 32) & VAL_MASK;
            acc[2] += (uint32_t)v[u][1] & VAL_MASK; acc[3] += (uint32_t)(v[u][1] >> 32) & VAL_MASK;
        }
    }
    *reinterpret_cast<uint4 *>(&s_part[wave][4 * lane]) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
}

// Digit totals of every pass: g_hist[pass][digit] += ...   (g_hist zeroed by the caller).  One block = 16 keys/thread.
template <typename K>
__global__ __launch_bounds__(THREADS) void hist_all_kernel(const SizeRef size, const K *__restrict__ keys, const Plan plan,
                                                          uint32_t *__restrict__ g_hist) {
    __shared__ uint32_t s_hist[MAX_PASSES][RADIX];
    const int64_t n = size_of(size);
    const int64_t base = (int64_t)blockIdx.x * (THREADS * 16);
    if (base >= n) return;
    const int tid = threadIdx.x;
    for (int p = 0; p < plan.npass; ++p) s_hist[p][tid] = 0;
    __syncthreads();
    K k[16];   // all loads first: one memory round trip per block, not one per key
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t j = base + (int64_t)i * THREADS + tid;
        k[i] = j < n ? keys[j] : (K)0;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (base + (int64_t)i * THREADS + tid < n) {
            for (int p = 0; p < plan.npass; ++p)
                atomicAdd(&s_hist[p][(unsigned)(k[i] >> plan.shift[p]) & ((1u << plan.bits[p]) - 1u)], 1u);
        }
    }
    __syncthreads();
    for (int p = 0; p < plan.npass; ++p) {
        const uint32_t c = s_hist[p][tid];
        if (c) atomicAdd(&g_hist[p * RADIX + tid], c);
    }
}

// Last-pass epilogue: gather(value) fetches what the outputs need (issued for four elements before the first store),
// store(dst, key, value, gathered) writes them INSTEAD of the key / value stores.
struct NoEpilogue {
    static constexpr bool enabled = false;
    int unused = 0;   // (never an EMPTY struct as a by-value kernel argument)
    struct G {};
    __device__ __forceinline__ G gather(int32_t) const { return G{}; }
    __device__ __forceinline__ void store(uint32_t, uint64_t, int32_t, const G &) const {}
};

// One digit pass.  vals_in == nullptr: the value of element j is j (first pass of a sort of iota-valued pairs).
// Epi (last pass only): called as epi(dst, key, value) INSTEAD of the key / value stores.
template <typename K, int ITEMS, class Epi>
__global__ __launch_bounds__(THREADS) void pass_kernel(const SizeRef size, const K *__restrict__ keys_in,
                                                      const int32_t *__restrict__ vals_in, K *__restrict__ keys_out,
                                                      int32_t *__restrict__ vals_out, int shift, int bits,
                                                      const uint32_t *__restrict__ g_hist /* [RADIX], this pass */,
                                                      uint32_t *__restrict__ ticket, uint32_t *__restrict__ state /* [blocks][RADIX] */,
                                                      uint32_t *__restrict__ gstate /* [groups][RADIX] */, int group,
                                                      uint32_t *__restrict__ err, Epi epi) {
    constexpr int TILE = THREADS * ITEMS;
    __shared__ uint32_t s_off[WAVES][RADIX];  // per-wave digit counts, then running LOCAL offsets
    __shared__ uint32_t s_gbase[RADIX];       // global position of local slot 0 of each digit
    __shared__ uint32_t s_w[WAVES], s_g[WAVES];
    __shared__ __attribute__((aligned(16))) uint32_t s_row[RADIX];   // the row this block publishes
    __shared__ __attribute__((aligned(16))) uint32_t s_part[WAVES][RADIX];
    __shared__ K s_key[TILE];
    __shared__ int32_t s_val[TILE];
    __shared__ int s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = (int)atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s_off[w][tid] = 0;
    __syncthreads();
    const int bid = s_ticket;
    const int64_t n = size_of(size);
    const int64_t tile_base = (int64_t)bid * TILE;
    if (tile_base >= n) return;   // (tickets past the data: nobody waits on them)
    const unsigned mask = (1u << bits) - 1u;
    // ---- load this wave's sub-tile (order: wave, iteration, lane == increasing index) + count digits
    const int64_t wbase = tile_base + (int64_t)wave * (64 * ITEMS);
    K key[ITEMS];
    int32_t val[ITEMS];
    unsigned dig[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {   // every load of the tile is issued before the first is used
        const int64_t j = wbase + i * 64 + lane;
        key[i] = j < n ? keys_in[j] : (K)0;
        val[i] = j < n ? (vals_in ? vals_in[j] : (int32_t)j) : 0;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        dig[i] = (unsigned)(key[i] >> shift) & mask;
        if (wbase + i * 64 + lane < n) atomicAdd(&s_off[wave][dig[i]], 1u);
        else dig[i] = 0;
    }
    __syncthreads();
    // ---- per digit (thread d): publish the block's counts; local offsets
    uint32_t cnt = 0, gbase_excl, local_start;
    {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) cnt += s_off[w][tid];
        s_row[tid] = READY | cnt;
        // local exclusive scan of the block's digit counts, and exclusive scan of the global digit totals
        const uint32_t tot = g_hist[tid];
        uint32_t inc = cnt, ginc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64), gup = __shfl_up(ginc, o, 64);
            if (lane >= o) { inc += up; ginc += gup; }
        }
        if (lane == 63) { s_w[wave] = inc; s_g[wave] = ginc; }
        __syncthreads();
        publish_row(state + (int64_t)bid * RADIX, s_row);
        uint32_t lb = 0, gb = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
            if (w < wave) { lb += s_w[w]; gb += s_g[w]; }
        local_start = lb + inc - cnt;
        gbase_excl = gb + ginc - tot;
        uint32_t running = local_start;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = s_off[w][tid];
            s_off[w][tid] = running;
            running += c;
        }
    }
    __syncthreads();
    // ---- stable LOCAL rank, 64 keys per step; the tile is rebuilt in LDS grouped by digit.  This needs nothing from the
    // other blocks, so it runs BETWEEN publishing the counts and reading the others': the memory-side round trips of
    // the cross-block prefix (1.5-2 us each) are hidden behind it instead of stalling every block.
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
    // ---- elements with this digit in the blocks in front: inside the group, then the groups in front
    {
        const int g = bid / group, q = bid - g * group;
        sum_rows(state, g * group, q, s_part, err);
        __syncthreads();
        uint32_t prev = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) prev += s_part[w][tid];
        if (q == group - 1) {   // (block-uniform)
            s_row[tid] = READY | ((prev + cnt) & VAL_MASK);
            __syncthreads();
            publish_row(gstate + (int64_t)g * RADIX, s_row);
        }
        __syncthreads();
        sum_rows(gstate, 0, g, s_part, err);
        __syncthreads();
#pragma unroll
        for (int w = 0; w < WAVES; ++w) prev += s_part[w][tid];
        s_gbase[tid] = gbase_excl + prev - local_start;
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
                const uint32_t dst = s_gbase[(unsigned)(kk[u] >> shift) & mask] + (uint32_t)k;
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

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
template <int ITEMS>
inline int64_t pass_blocks(int64_t cap) { return ceil_div64(cap > 0 ? cap : 1, THREADS * ITEMS); }
inline int group_size(int64_t blocks) {   // ~sqrt(blocks), at least 32: both levels of the prefix stay short
    int g = 32;
    while ((int64_t)g * g < blocks) g *= 2;
    return g;
}
// control words of one sort (must be ZERO before the sort starts): ticket[MAX_PASSES] | err | g_hist[MAX_PASSES][RADIX] |
// per pass: state[blocks][RADIX] | gstate[groups][RADIX]
template <int ITEMS>
inline size_t pass_state_bytes(int64_t cap) {
    const int64_t blocks = pass_blocks<ITEMS>(cap);
    return align256((size_t)(blocks + ceil_div64(blocks, group_size(blocks))) * RADIX * 4);
}
template <int ITEMS>
inline size_t control_bytes(int64_t cap, int npass) {
    return align256((size_t)(MAX_PASSES + 8) * 4) + align256((size_t)MAX_PASSES * RADIX * 4) + (size_t)npass * pass_state_bytes<ITEMS>(cap);
}

template <typename K, int ITEMS>
struct Sorter {
    Plan plan;
    int64_t cap, blocks;
    int group;
    uint32_t *ticket, *err, *g_hist, *state;
    size_t state_stride;  // uint32 words per pass
    Sorter(int key_bits, int64_t cap_, void *control) : plan(make_plan(key_bits)), cap(cap_) {
        blocks = pass_blocks<ITEMS>(cap);
        group = group_size(blocks);
        char *w = (char *)control;
        ticket = (uint32_t *)w;
        err = ticket + MAX_PASSES;
        w += align256((size_t)(MAX_PASSES + 8) * 4);
        g_hist = (uint32_t *)w;
        w += align256((size_t)MAX_PASSES * RADIX * 4);
        state = (uint32_t *)w;
        state_stride = pass_state_bytes<ITEMS>(cap) / 4;
    }
    void hist(const SizeRef size, const K *keys, hipStream_t st) const {
        hist_all_kernel<K><<<(unsigned)ceil_div64(cap > 0 ? cap : 1, THREADS * 16), THREADS, 0, st>>>(size, keys, plan, g_hist);
    }
    template <class Epi = NoEpilogue>
    void pass(int p, const SizeRef size, const K *kin, const int32_t *vin, K *kout, int32_t *vout, hipStream_t st,
              Epi epi = Epi()) const {
        uint32_t *stp = state + (size_t)p * state_stride;
        pass_kernel<K, ITEMS, Epi><<<(unsigned)blocks, THREADS, 0, st>>>(size, kin, vin, kout, vout, plan.shift[p], plan.bits[p],
                                                                        g_hist + p * RADIX, ticket + p, stp,
                                                                        stp + (size_t)blocks * RADIX, group, err, epi);
    }
};

}  // namespace mtgs_os
