// devsize.hpp -- element counts that live in DEVICE memory (speculative sizing / graph capture: the host sizes buffers
// and grids for a capacity, the kernels read the true count) and relaxed agent-scope accessors for words that
// workgroups of one launch hand to each other.
#pragma once
#include "common.hpp"

namespace mtgs_os {

__device__ __forceinline__ uint32_t ld32(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// mode 0 = *p, mode 1 = *p >> 32 (n_vis of front.hip's packed totals); clamped to the capacity the buffers and the grid
// were sized for.
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
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace mtgs_os
