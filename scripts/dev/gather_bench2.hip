// Development: the floor of the row-lazy optimizer's access pattern -- per visible Gaussian three tensors (dc [N,3], adapter
// [N,T,3], rest [N,T,45]), each with p, m, v and a `last` stamp, plus compact rows -- as a plain kernel over a precomputed id list.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Tens { float *p, *m, *v; int *last; int width, sw, off; };   // row stride `width`, slice [off, off + sw)

__global__ __launch_bounds__(256) void pattern_kernel(const int *__restrict__ ids, int n_rows, Tens dc, Tens ad, Tens rs, float *comp,
                                                      const float *grad, int step_mode, int rpg, float *out) {
    const int grp = (blockIdx.x * 256 + threadIdx.x) >> 4, c0 = threadIdx.x & 15;
    float acc = 0.f;
    for (int k = 0; k < rpg; ++k) {
        const int r = grp * rpg + k;
        if (r >= n_rows) break;
        const long i = ids[r];
        const Tens *ts[3] = {&dc, &ad, &rs};
        const int cols[3] = {0, 3, 6};
        for (int q = 0; q < 3; ++q) {
            const Tens &t = *ts[q];
            const int L = t.last[i];
            for (int u = 0; u < 3; ++u) {
                const int c = 16 * u + c0;
                if (c >= t.sw) continue;
                const long a = i * t.width + t.off + c;
                if (!step_mode) {       // peek: read p, m, v; write the compact row
                    const float s = t.p[a] + t.m[a] * 0.9f + t.v[a] * (float)L;
                    comp[(long)r * 52 + cols[q] + c] = s;
                } else {                // step: read m, v, compact p, gradient; write p, m, v
                    const float pc = comp[(long)r * 52 + cols[q] + c], g = grad[(long)r * 48 + (q == 2 ? 3 : 0) + c];
                    const float m = t.m[a] * 0.9f + g, v = t.v[a] * 0.999f + g * g + (float)L;
                    t.p[a] = pc - m; t.m[a] = m; t.v[a] = v;
                    acc += m;
                }
            }
            if (step_mode && c0 == 0) t.last[i] = L + 1;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

static Tens make(long N, int width, int sw, int off) {
    Tens t; t.width = width; t.sw = sw; t.off = off;
    hipMalloc(&t.p, N * width * 4); hipMalloc(&t.m, N * width * 4); hipMalloc(&t.v, N * width * 4); hipMalloc(&t.last, N * 4 * (width / sw));
    hipMemset(t.p, 0, N * width * 4); hipMemset(t.m, 0, N * width * 4); hipMemset(t.v, 0, N * width * 4); hipMemset(t.last, 0, N * 4);
    return t;
}

int main() {
    const int N = 1600000, T = 3;
    Tens dc = make(N, 3, 3, 0), ad = make(N, 3 * T, 3, 3), rs = make(N, 45 * T, 45, 45);
    std::vector<int> ids;
    srand(1);
    for (int i = 0; i < N; ++i) if (rand() % 100 < 19) ids.push_back(i);
    const int n = (int)ids.size();
    int *d_ids; hipMalloc(&d_ids, n * 4); hipMemcpy(d_ids, ids.data(), n * 4, hipMemcpyHostToDevice);
    float *comp, *grad, *out; hipMalloc(&comp, (size_t)n * 52 * 4); hipMalloc(&grad, (size_t)n * 48 * 4); hipMalloc(&out, 4);
    hipMemset(grad, 0, (size_t)n * 48 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int rpg : {1, 4, 8}) {
            const int groups = (n + rpg - 1) / rpg, blocks = (groups * 16 + 255) / 256;
            for (int w = 0; w < 3; ++w) pattern_kernel<<<blocks, 256>>>(d_ids, n, dc, ad, rs, comp, grad, mode, rpg, out);
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) pattern_kernel<<<blocks, 256>>>(d_ids, n, dc, ad, rs, comp, grad, mode, rpg, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s rows/group %d: %7.1f us (%d visible rows, three tensors + stamps)\n", mode ? "step-like" : "peek-like", rpg, ms * 100, n);
        }
    return 0;
}
