// zero.hip -- kernel-side zero fill (see common.hpp: why not hipMemsetAsync).
#include "common.hpp"

namespace {
__global__ __launch_bounds__(256) void zero_kernel(uint32_t *__restrict__ p, size_t n_words) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n4 = n_words >> 2;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        if (i < n4) reinterpret_cast<uint4 *>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (i < (n_words & 3)) p[(n4 << 2) + i] = 0u;
    } else {
        for (size_t k = i; k < n_words; k += (size_t)gridDim.x * 256) p[k] = 0u;
    }
}
// 16 consecutive 16-byte stores per thread, block-contiguous: the fastest pure write measured on this chip (6.8 TB/s for 384 MiB,
// scripts/dev/write_bench.hip) -- for the large fills the Python layer overlaps with compute-bound kernels
__global__ __launch_bounds__(256) void fill_zero_kernel(uint4 *__restrict__ p, size_t n16) {
    const size_t base = (size_t)blockIdx.x * 256 * 16 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}
}  // namespace

int mtgs_zero_async(void *p, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return MTGS_OK;
    MTGS_REQUIRE(p && (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0, MTGS_EINVAL, "mtgs_zero_async: unaligned");
    const size_t n_words = bytes >> 2;
    const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    // aligned: one 16-byte store per thread (+ up to 3 tail words by the first threads); otherwise a strided loop
    size_t threads = aligned ? (n_words >> 2) : n_words;
    if (threads < 4) threads = 4;
    size_t blocks = (threads + 255) / 256;
    if (!aligned && blocks > 65536) blocks = 65536;
    zero_kernel<<<(unsigned)blocks, 256, 0, stream>>>((uint32_t *)p, n_words);
    MTGS_CHECK_LAUNCH("mtgs_zero_async");
    return MTGS_OK;
}

extern "C" int mtgs_fill_zero(void *p, size_t bytes, void *stream) {
    if (bytes == 0) return MTGS_OK;
    MTGS_REQUIRE(p && (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0, MTGS_EINVAL, "mtgs_fill_zero: whole, aligned 4-byte words");
    hipStream_t st = (hipStream_t)stream;
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0 || bytes < ((size_t)1 << 20)) return mtgs_zero_async(p, bytes, st);
    const size_t n16 = bytes >> 4;
    fill_zero_kernel<<<(unsigned)((n16 + 4095) / 4096), 256, 0, st>>>((uint4 *)p, n16);
    MTGS_CHECK_LAUNCH("mtgs_fill_zero");
    return mtgs_zero_async((char *)p + (n16 << 4), bytes - (n16 << 4), st);
}
