// lookback.hpp -- decoupled look-back (single-pass chained scan) for gfx950.
//
// A workgroup publishes its AGGREGATE as soon as it is known, then walks back over its predecessors, summing
// aggregates until it meets a block whose INCLUSIVE prefix is already published.  One 8-byte word per block carries
// {status, value}, read and written with relaxed agent-scope atomics (sc1 accesses: served by the memory side, so
// the per-XCD L2s and the per-CU L1s never hold a stale copy; "8-B agent atomics both sides" is one of the valid
// hand-off forms of MI355X_MICROARCH.md).  The word IS the payload, so no separate flag / fence is needed.
//
// Forward progress: blocks take their logical index from an atomic TICKET, so a block only ever waits on blocks
// that already started (HIP promises nothing about dispatch order).  Every spin is bounded: a bound that trips
// sets bit 1 of *err and the block carries on with a wrong prefix instead of hanging the GPU.
//
// The value is a pair of 31-bit counters {hi, lo} (visible Gaussians, tile intersections); the walk sums the two
// fields separately in 64 bits so that an overflowing `lo` is detected by the caller, not carried into `hi`.
#pragma once
#include "common.hpp"

namespace mtgs_lb {

constexpr uint64_t ST_AGG = 1ull << 62, ST_INCL = 2ull << 62, ST_MASK = 3ull << 62, VAL_MASK = ~ST_MASK;
constexpr uint64_t FIELD_MAX = 0x7fffffffull;
constexpr int SPIN_LIMIT = 1 << 22;
constexpr uint32_t ERR_OVERFLOW = 1u, ERR_SPIN = 2u;

__device__ __forceinline__ uint64_t ld(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t pack(uint64_t hi, uint64_t lo) { return (hi << 31) | lo; }

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct Pair { uint64_t hi, lo; };

// Ticket of this workgroup (its logical block index), broadcast through *slot (LDS).  Includes a barrier.
__device__ __forceinline__ int take_ticket(uint32_t *counter, int *slot) {
    if (threadIdx.x == 0) *slot = (int)atomicAdd(counter, 1u);
    __syncthreads();
    return *slot;
}

// Called by ALL 64 lanes of ONE wave.  Publishes the block's aggregate {agg_hi, agg_lo}, returns the EXCLUSIVE prefix
// (sum over the blocks with a smaller ticket) in every lane, then publishes the inclusive prefix (lo saturating at
// FIELD_MAX with ERR_OVERFLOW).
__device__ __forceinline__ Pair lookback_wave(uint64_t *state, int bid, uint64_t agg_hi, uint64_t agg_lo, uint32_t *err) {
    const int lane = (int)(threadIdx.x & 63);
    uint64_t ehi = 0, elo = 0;
    if (bid > 0) {
        if (lane == 0) st(state + bid, ST_AGG | pack(agg_hi, agg_lo));
        int base = bid - 1, spins = 0;
        while (true) {
            const int j = base - lane;
            const uint64_t s = j >= 0 ? ld(state + j) : ST_INCL;  // "block -1": inclusive prefix 0
            const unsigned long long none = __ballot((s & ST_MASK) == 0);
            const unsigned long long incl = __ballot((s & ST_MASK) == ST_INCL);
            const int first = incl ? __builtin_ctzll(incl) : 63;   // lanes 0..first are needed
            const unsigned long long need = first == 63 ? ~0ull : ((2ull << first) - 1ull);
            if (none & need) {
                if (++spins > SPIN_LIMIT) {
                    if (lane == 0) atomicOr(err, ERR_SPIN);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            const uint64_t v = lane <= first ? (s & VAL_MASK) : 0ull;
            ehi += wave_sum_u64(v >> 31);
            elo += wave_sum_u64(v & FIELD_MAX);
            if (incl) break;
            base -= 64;
        }
    }
    uint64_t ilo = elo + agg_lo;
    if (ilo > FIELD_MAX) {
        if (lane == 0) atomicOr(err, ERR_OVERFLOW);
        ilo = FIELD_MAX;
    }
    if (elo > FIELD_MAX) elo = FIELD_MAX;
    if (lane == 0) st(state + bid, ST_INCL | pack(ehi + agg_hi, ilo));
    return Pair{ehi, elo};
}

}  // namespace mtgs_lb
