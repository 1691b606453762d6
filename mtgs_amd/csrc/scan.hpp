// scan.hpp -- three-phase int64 prefix sum with a value functor and an output sink (gfx950).
//   value(i)            -> int64 element i
//   sink(i, excl, incl) -> consumes the exclusive / inclusive prefix of element i
// Phases: per-block partial sums (2048 elements per block) -> single-block spine scan -> per-block
// scan + sink.  Workspace: one int64 per block (+1).
#pragma once
#include "common.hpp"

namespace mtgs_scan {

constexpr int BLOCK = 256;
constexpr int ITEMS = 8;
constexpr int TILE = BLOCK * ITEMS;

inline size_t workspace_bytes(int64_t n) { return (size_t)(ceil_div64(n, TILE) + 1) * sizeof(int64_t); }

__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t *lds, int64_t &block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        const int64_t t = lds[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    block_total = tot;
    return base + inc - v;
}

template <class Value>
__global__ __launch_bounds__(BLOCK) void partials_kernel(int64_t n, Value value, int64_t *__restrict__ partials) {
    __shared__ int64_t lds[BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * TILE;
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t j = base + (int64_t)i * BLOCK + threadIdx.x;
        if (j < n) s += value(j);
    }
    int64_t tot;
    block_exclusive_scan(s, lds, tot);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// host_total (nullable): HOST-visible (pinned) int64[2].  The grand total is known here, one kernel before the
// per-element pass: it is published as {total, tag} with system-scope ordering so that a host thread polling
// host_total[1] for `tag` learns the total while the GPU is still busy (no stream synchronisation).
template <int DUMMY = 0>
__global__ __launch_bounds__(BLOCK) void spine_kernel(int64_t nblocks, int64_t *__restrict__ partials,
                                                      int64_t *__restrict__ total_out, int64_t *host_total = nullptr,
                                                      int64_t tag = 0) {
    __shared__ int64_t lds[BLOCK / 64];
    int64_t carry = 0;
    for (int64_t b0 = 0; b0 < nblocks; b0 += BLOCK) {
        const int64_t j = b0 + threadIdx.x;
        const int64_t v = j < nblocks ? partials[j] : 0;
        int64_t tot;
        const int64_t ex = block_exclusive_scan(v, lds, tot);
        if (j < nblocks) partials[j] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
    if (threadIdx.x == 0 && host_total) {
        __hip_atomic_store(host_total, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_total + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <class Value, class Sink>
__global__ __launch_bounds__(BLOCK) void final_kernel(int64_t n, Value value, Sink sink,
                                                      const int64_t *__restrict__ partials) {
    __shared__ int64_t lds[BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * ITEMS;
    int64_t v[ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        v[i] = base + i < n ? value(base + i) : 0;
        s += v[i];
    }
    int64_t tot;
    int64_t run = partials[blockIdx.x] + block_exclusive_scan(s, lds, tot);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t excl = run;
        run += v[i];
        if (base + i < n) sink(base + i, excl, run);
    }
}

template <class Value, class Sink>
inline void run(int64_t n, Value value, Sink sink, int64_t *partials, int64_t *total_out, hipStream_t st,
                int64_t *host_total = nullptr, int64_t tag = 0) {
    const int64_t nblocks = ceil_div64(n, TILE);
    partials_kernel<<<(unsigned)nblocks, BLOCK, 0, st>>>(n, value, partials);
    spine_kernel<0><<<1, BLOCK, 0, st>>>(nblocks, partials, total_out, host_total, tag);
    final_kernel<<<(unsigned)nblocks, BLOCK, 0, st>>>(n, value, sink, partials);
}

}  // namespace mtgs_scan
