// Development: what rate do scattered 180-byte pieces move at?  (the row-lazy optimizer's access pattern: a [N, T, 45] float
// tensor, 15 % of the rows of one slice, from three arrays, optionally written back)
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void gather_kernel(const int *__restrict__ ids, int n_rows, float *a, float *b, float *c, int T,
                                                     int write, int rows_per_group, float *out) {
    const int grp = (blockIdx.x * 256 + threadIdx.x) >> 4, c0 = threadIdx.x & 15;
    float acc = 0.f;
    for (int k = 0; k < rows_per_group; ++k) {
        const int r = grp * rows_per_group + k;
        if (r >= n_rows) break;
        const long base = (long)ids[r] * T * 45;
        float x[3], y[3], z[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int col = 16 * u + c0;
            x[u] = y[u] = z[u] = 0.f;
            if (col < 45) { x[u] = a[base + col]; y[u] = b[base + col]; z[u] = c[base + col]; }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int col = 16 * u + c0;
            const float s = x[u] * 0.999f + y[u] * 0.5f + z[u];
            acc += s;
            if (write && col < 45) { a[base + col] = s; b[base + col] = y[u] * 0.9f; c[base + col] = z[u] * 0.999f; }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const int N = 1600000, T = 3;
    const size_t elems = (size_t)N * T * 45;
    float *a, *b, *c, *out;
    hipMalloc(&a, elems * 4); hipMalloc(&b, elems * 4); hipMalloc(&c, elems * 4); hipMalloc(&out, 4);
    hipMemset(a, 0, elems * 4); hipMemset(b, 0, elems * 4); hipMemset(c, 0, elems * 4);
    std::vector<int> ids;
    srand(1);
    for (int i = 0; i < N; ++i) if (rand() % 100 < 19) ids.push_back(i);
    int *d_ids; hipMalloc(&d_ids, ids.size() * 4);
    hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
    const int n = (int)ids.size();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int write = 0; write < 2; ++write)
        for (int rpg : {1, 2, 4, 8}) {
            const int groups = (n + rpg - 1) / rpg, blocks = (groups * 16 + 255) / 256;
            for (int w = 0; w < 3; ++w) gather_kernel<<<blocks, 256>>>(d_ids, n, a, b, c, T, write, rpg, out);
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) gather_kernel<<<blocks, 256>>>(d_ids, n, a, b, c, T, write, rpg, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)n * 180 * 3 * (write ? 2 : 1);
            printf("%s rows/group %d: %7.1f us  %6.0f GB/s of 180-byte pieces (%d rows, %.0f MB)\n", write ? "read+write" : "read only ",
                   rpg, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9, n, bytes / 1e6);
        }
    return 0;
}
