// scratch: which store pattern streams 384 MiB to HBM fastest on this chip (rocclr's fillBufferAligned was seen at 8.2 TB/s,
// a one-float4-per-thread kernel at 6.65 TB/s)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void w_one(f4 *__restrict__ p, size_t n) {   // one 16 B store per thread
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    if (i < n) { if (MODE) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}
template <int MODE, int U> __global__ __launch_bounds__(256) void w_strided(f4 *__restrict__ p, size_t n) {  // U stores per thread, block-contiguous chunks
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256; if (i < n) { if (MODE) __builtin_nontemporal_store(v, p + i); else p[i] = v; } }
}
template <int MODE> __global__ __launch_bounds__(256) void w_persist(f4 *__restrict__ p, size_t n) {  // grid-stride, few blocks
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { if (MODE) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}
template <int MODE> __global__ __launch_bounds__(256) void w_rows192(float *__restrict__ p, size_t rows) {  // sh_bwd-like: 128 rows x 192 B per 128 threads
    const size_t g0 = (size_t)blockIdx.x * 256;
    f4 *dst = (f4 *)(p + g0 * 48);
    const size_t n4 = (rows - g0 < 256 ? rows - g0 : 256) * 12;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = threadIdx.x; i < n4; i += 256) { if (MODE) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}
template <class F> void timeit(const char *name, size_t bytes, F &&launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("%-34s %7.1f us %6.0f GB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e9);
}
int main() {
    const size_t bytes = (size_t)384 << 20, n = bytes / 16;
    f4 *p; hipMalloc(&p, bytes);
    timeit("hipMemsetAsync", bytes, [&] { hipMemsetAsync(p, 0, bytes, 0); });
    timeit("one f4/thread", bytes, [&] { w_one<0><<<(unsigned)((n + 255) / 256), 256>>>(p, n); });
    timeit("one f4/thread nt", bytes, [&] { w_one<1><<<(unsigned)((n + 255) / 256), 256>>>(p, n); });
    timeit("4 f4/thread", bytes, [&] { w_strided<0, 4><<<(unsigned)((n + 1023) / 1024), 256>>>(p, n); });
    timeit("4 f4/thread nt", bytes, [&] { w_strided<1, 4><<<(unsigned)((n + 1023) / 1024), 256>>>(p, n); });
    timeit("16 f4/thread", bytes, [&] { w_strided<0, 16><<<(unsigned)((n + 4095) / 4096), 256>>>(p, n); });
    timeit("16 f4/thread nt", bytes, [&] { w_strided<1, 16><<<(unsigned)((n + 4095) / 4096), 256>>>(p, n); });
    for (int g : {1024, 2048, 4096, 8192}) {
        char nm[64]; snprintf(nm, sizeof nm, "grid-stride %d blocks", g);
        timeit(nm, bytes, [&] { w_persist<0><<<g, 256>>>(p, n); });
        snprintf(nm, sizeof nm, "grid-stride %d blocks nt", g);
        timeit(nm, bytes, [&] { w_persist<1><<<g, 256>>>(p, n); });
    }
    const size_t rows = bytes / 192;
    timeit("rows of 192 B, 256/block", bytes, [&] { w_rows192<0><<<(unsigned)((rows + 255) / 256), 256>>>((float *)p, rows); });
    timeit("rows of 192 B, 256/block nt", bytes, [&] { w_rows192<1><<<(unsigned)((rows + 255) / 256), 256>>>((float *)p, rows); });
    return 0;
}
