// Development microbenchmark: HBM-cold streaming READ bandwidth on MI355X for the access shapes of the SH kernels:
// 16-byte vs 12-byte per-lane loads of a contiguous 384 MB array, U loads in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 -o read_bench read_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct F3 { float x, y, z; };
template <typename T, int U>
__global__ __launch_bounds__(256) void rd(const T *__restrict__ p, size_t n_per_wave_iter, size_t n, float *out) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t base = wave * 64 * U;
    if (base >= n) return;
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[min(base + (size_t)u * 64 + lane, n - 1)];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += reinterpret_cast<float *>(&v[u])[0] + reinterpret_cast<float *>(&v[u])[sizeof(T) / 4 - 1];
    if (s == 123.456f) out[0] = s;
}
// mode 0: rewrite the flush buffer (leaves up to 256 MB of DIRTY lines in the Infinity Cache, whose write-back then
// competes with the timed reads -- the situation inside a real step); mode 1: only read it (clean eviction)
__global__ void flush(float *p, size_t n, int mode, float *out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (mode == 0) p[i] += 1.f; else s += p[i];
    }
    if (s == 123.456f) out[1] = s;
}
template <typename T, int U>
void run(const char *name, void *buf, size_t bytes, float *fl, size_t fl_n, float *out, int mode) {
    const size_t n = bytes / sizeof(T);
    const size_t waves = (n + 64 * U - 1) / (64 * U);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        flush<<<4096, 256>>>(fl, fl_n, mode, out);
        hipEventRecord(e0);
        rd<T, U><<<(unsigned)((waves + 3) / 4), 256>>>((const T *)buf, 0, n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-22s %s %7.1f us  %6.0f GB/s\n", name, mode ? "after clean flush" : "after dirty flush", best * 1e3, bytes / (best * 1e-3) / 1e9);
}
// pure streaming WRITE (16 bytes per lane), for the WRITE_SIZE calibration
__global__ __launch_bounds__(256) void wr(float4 *__restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main(int argc, char **argv) {
    if (argc > 1) {   // calibration mode for rocprofv3 --pmc: exactly ONE kind of access per kernel, known byte counts
        const size_t bytes = (size_t)384 << 20;
        void *buf; float *out;
        hipMalloc(&buf, bytes); hipMalloc(&out, 16);
        hipMemset(buf, 0, bytes);
        for (int rep = 0; rep < 3; ++rep) {
            const size_t n4 = bytes / 16, waves = (n4 + 64 * 16 - 1) / (64 * 16);
            rd<float4, 16><<<(unsigned)((waves + 3) / 4), 256>>>((const float4 *)buf, 0, n4, out);     // reads 384 MiB
            const size_t n1 = bytes / 4, waves1 = (n1 + 64 * 16 - 1) / (64 * 16);
            rd<float, 16><<<(unsigned)((waves1 + 3) / 4), 256>>>((const float *)buf, 0, n1, out);       // reads 384 MiB, 4 B/lane
            wr<<<(unsigned)((n4 + 255) / 256), 256>>>((float4 *)buf, n4);                                // writes 384 MiB
        }
        hipDeviceSynchronize();
        printf("calibration kernels done: each moves %zu bytes\n", bytes);
        return 0;
    }
    const size_t bytes = (size_t)384 << 20, fl_n = (size_t)192 << 20;
    void *buf; float *fl, *out;
    hipMalloc(&buf, bytes); hipMalloc(&fl, fl_n * 4); hipMalloc(&out, 16);
    hipMemset(buf, 0, bytes); hipMemset(fl, 0, fl_n * 4);
    for (int mode = 0; mode < 2; ++mode) {
        run<float4, 4>("float4 x4 in flight", buf, bytes, fl, fl_n, out, mode);
        run<float4, 16>("float4 x16", buf, bytes, fl, fl_n, out, mode);
        run<F3, 16>("12-byte x16", buf, bytes, fl, fl_n, out, mode);
        run<float, 16>("4-byte x16", buf, bytes, fl, fl_n, out, mode);
    }
    return 0;
}
