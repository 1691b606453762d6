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
// sh_bwd-like: per block 128 rows: each thread reads 2 x 12 bytes of its row's inputs, stages 48 floats in LDS, barrier, block writes 24.5 KB coalesced
template <int BLOCK, int WORK> __global__ __launch_bounds__(BLOCK) void w_shlike(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ p, size_t rows) {
    __shared__ float lds[BLOCK * 49];
    const size_t g0 = (size_t)blockIdx.x * BLOCK;
    const int cnt = (int)(rows - g0 < BLOCK ? rows - g0 : BLOCK), tid = threadIdx.x;
    if (tid < cnt) {
        const size_t g = g0 + tid;
        float x = a[g * 3], y = a[g * 3 + 1], z = a[g * 3 + 2], v0 = b[g * 3], v1 = b[g * 3 + 1], v2 = b[g * 3 + 2];
        float acc = x;
#pragma unroll
        for (int w = 0; w < WORK; ++w) acc = acc * y + z;      // stand-in for the basis evaluation
        float *row = lds + tid * 49;
#pragma unroll
        for (int k = 0; k < 16; ++k) { row[3 * k] = acc * v0 + k; row[3 * k + 1] = acc * v1; row[3 * k + 2] = acc * v2; }
    }
    __syncthreads();
    f4 *dst = (f4 *)(p + g0 * 48);
    const int n4 = cnt * 12;
    for (int i4 = tid; i4 < n4; i4 += BLOCK) {
        const int g = i4 / 12, j = (i4 % 12) * 4;
        const float *s = lds + g * 49;
        f4 v = {s[j], s[j + 1], s[j + 2], s[j + 3]};
        dst[i4] = v;
    }
}
// the same with 16-byte LDS accesses: 12 ds_write_b128 + 12 ds_read_b128 per thread instead of 48 + 48 four-byte ones (row stride 52 floats)
template <int BLOCK, int WORK, int RS> __global__ __launch_bounds__(BLOCK) void w_shlike128(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ p, size_t rows) {
    __shared__ __attribute__((aligned(16))) float lds[BLOCK * RS];
    const size_t g0 = (size_t)blockIdx.x * BLOCK;
    const int cnt = (int)(rows - g0 < BLOCK ? rows - g0 : BLOCK), tid = threadIdx.x;
    if (tid < cnt) {
        const size_t g = g0 + tid;
        float x = a[g * 3], y = a[g * 3 + 1], z = a[g * 3 + 2], v[3] = {b[g * 3], b[g * 3 + 1], b[g * 3 + 2]};
        float acc = x;
#pragma unroll
        for (int w = 0; w < WORK; ++w) acc = acc * y + z;
        f4 *row = (f4 *)(lds + tid * RS);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            f4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int e = 4 * q + i; o[i] = (acc + (float)(e / 3)) * v[e % 3]; }
            row[q] = o;
        }
    }
    __syncthreads();
    f4 *dst = (f4 *)(p + g0 * 48);
    const int n4 = cnt * 12;
    for (int i4 = tid; i4 < n4; i4 += BLOCK) {
        const int g = i4 / 12, q = i4 % 12;
        dst[i4] = *(const f4 *)(lds + g * RS + q * 4);
    }
}
// sparse cotangents (the real frame: ~6 % of the Gaussians carry a gradient): only those rows are computed and staged; the block's
// coalesced write takes zeros from registers for the others (flag per row in LDS)
template <int BLOCK, int WORK, int RS> __global__ __launch_bounds__(BLOCK) void w_shsparse(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ p, size_t rows) {
    __shared__ __attribute__((aligned(16))) float lds[BLOCK * RS];
    __shared__ unsigned char s_nz[BLOCK];
    const size_t g0 = (size_t)blockIdx.x * BLOCK;
    const int cnt = (int)(rows - g0 < BLOCK ? rows - g0 : BLOCK), tid = threadIdx.x;
    bool nz = false;
    if (tid < cnt) {
        const size_t g = g0 + tid;
        float v[3] = {b[g * 3], b[g * 3 + 1], b[g * 3 + 2]};
        nz = v[0] != 0.f || v[1] != 0.f || v[2] != 0.f;
        if (nz) {
            float x = a[g * 3], y = a[g * 3 + 1], z = a[g * 3 + 2];
            float acc = x;
#pragma unroll
            for (int w = 0; w < WORK; ++w) acc = acc * y + z;
            f4 *row = (f4 *)(lds + tid * RS);
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                f4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int e = 4 * q + i; o[i] = (acc + (float)(e / 3)) * v[e % 3]; }
                row[q] = o;
            }
        }
    }
    s_nz[tid] = nz;
    __syncthreads();
    f4 *dst = (f4 *)(p + g0 * 48);
    const int n4 = cnt * 12;
    for (int i4 = tid; i4 < n4; i4 += BLOCK) {
        const int g = i4 / 12, q = i4 % 12;
        f4 o = {0.f, 0.f, 0.f, 0.f};
        if (s_nz[g]) o = *(const f4 *)(lds + g * RS + q * 4);
        dst[i4] = o;
    }
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
    float *a, *b; hipMalloc(&a, rows * 12); hipMalloc(&b, rows * 12); hipMemset(a, 0, rows * 12); hipMemset(b, 0, rows * 12);
    const size_t tot = bytes + rows * 24;
    timeit("sh_bwd-like 128/block, work 0", tot, [&] { w_shlike<128, 0><<<(unsigned)((rows + 127) / 128), 128>>>(a, b, (float *)p, rows); });
    timeit("sh_bwd-like 128/block, work 40", tot, [&] { w_shlike<128, 40><<<(unsigned)((rows + 127) / 128), 128>>>(a, b, (float *)p, rows); });
    timeit("sh_bwd-like 256/block, work 40", tot, [&] { w_shlike<256, 40><<<(unsigned)((rows + 255) / 256), 256>>>(a, b, (float *)p, rows); });
    timeit("sh_bwd-like b128 128/blk rs52 w0", tot, [&] { w_shlike128<128, 0, 52><<<(unsigned)((rows + 127) / 128), 128>>>(a, b, (float *)p, rows); });
    timeit("sh_bwd-like b128 128/blk rs52 w40", tot, [&] { w_shlike128<128, 40, 52><<<(unsigned)((rows + 127) / 128), 128>>>(a, b, (float *)p, rows); });
    timeit("sh_bwd-like b128 256/blk rs52 w40", tot, [&] { w_shlike128<256, 40, 52><<<(unsigned)((rows + 255) / 256), 256>>>(a, b, (float *)p, rows); });
    timeit("sh_bwd-like b128 128/blk rs48 w40", tot, [&] { w_shlike128<128, 40, 48><<<(unsigned)((rows + 127) / 128), 128>>>(a, b, (float *)p, rows); });
    {   // 6 % of the rows carry a cotangent
        float *hb = (float *)malloc(rows * 12); for (size_t r = 0; r < rows; ++r) { const bool on = (r * 2654435761u) % 100 < 6; hb[3 * r] = hb[3 * r + 1] = hb[3 * r + 2] = on ? 1.f : 0.f; }
        hipMemcpy(b, hb, rows * 12, hipMemcpyHostToDevice); free(hb);
    }
    timeit("sh_bwd-like 128/blk, 6% rows", tot, [&] { w_shlike128<128, 40, 52><<<(unsigned)((rows + 127) / 128), 128>>>(a, b, (float *)p, rows); });
    timeit("sparse-staged 128/blk, 6% rows", tot, [&] { w_shsparse<128, 40, 52><<<(unsigned)((rows + 127) / 128), 128>>>(a, b, (float *)p, rows); });
    timeit("sparse-staged 256/blk, 6% rows", tot, [&] { w_shsparse<256, 40, 52><<<(unsigned)((rows + 255) / 256), 256>>>(a, b, (float *)p, rows); });
    timeit("sparse-staged 64/blk, 6% rows", tot, [&] { w_shsparse<64, 40, 52><<<(unsigned)((rows + 63) / 64), 64>>>(a, b, (float *)p, rows); });
    timeit("rows of 192 B again", bytes, [&] { w_rows192<0><<<(unsigned)((rows + 255) / 256), 256>>>((float *)p, rows); });
    timeit("sh_bwd-like 64/block, work 40", tot, [&] { w_shlike<64, 40><<<(unsigned)((rows + 63) / 64), 64>>>(a, b, (float *)p, rows); });
    return 0;
}
