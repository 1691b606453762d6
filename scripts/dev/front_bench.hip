// scratch: the I/O shape of front_project_kernel without its arithmetic -- per pair 12 + 16 + 12 + 4 bytes in from four arrays, 40 bytes
// out into seven arrays (4, 8, 4, 12, 4, 4, 4) -- to find what bounds the kernel (3.45 TB/s of counter traffic at 53 us, round 5):
//   direct : every thread loads and stores its own pair (the kernel's form)
//   wide   : the block's results go through LDS and every array is written with 16 bytes per lane
//   loads / stores : one side only
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/front_bench scripts/dev/front_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <functional>
typedef float f4 __attribute__((ext_vector_type(4)));
struct F3 { float x, y, z; };

template <int LOADS, int STORES>
__global__ __launch_bounds__(256) void direct(size_t n, const float *__restrict__ means, const f4 *__restrict__ quats, const float *__restrict__ scales,
                                              const float *__restrict__ opac, int *__restrict__ radii, float2 *__restrict__ m2d, float *__restrict__ depths,
                                              float *__restrict__ conics, float *__restrict__ comps, float *__restrict__ opeff, int *__restrict__ tiles) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    if (LOADS) {
        const F3 m = *reinterpret_cast<const F3 *>(means + i * 3);
        const f4 q = quats[i];
        const F3 s = *reinterpret_cast<const F3 *>(scales + i * 3);
        acc = m.x + m.y * m.z + q.x * q.y + q.z * q.w + s.x * s.y + s.z + opac[i];
    }
    if (STORES) {
        radii[i] = (int)acc;
        m2d[i] = make_float2(acc, acc + 1.f);
        depths[i] = acc;
        *reinterpret_cast<F3 *>(conics + i * 3) = F3{acc, acc, acc};
        comps[i] = acc; opeff[i] = acc; tiles[i] = (int)acc;
    } else if (acc == 12345.f) radii[i] = 1;
}

// wide stores: per block of 256 pairs, every output array's chunk (1 KB per 4-byte field) is written by 64 lanes x 16 bytes
__global__ __launch_bounds__(256) void wide(size_t n, const float *__restrict__ means, const f4 *__restrict__ quats, const float *__restrict__ scales,
                                            const float *__restrict__ opac, int *__restrict__ radii, float2 *__restrict__ m2d, float *__restrict__ depths,
                                            float *__restrict__ conics, float *__restrict__ comps, float *__restrict__ opeff, int *__restrict__ tiles) {
    __shared__ __attribute__((aligned(16))) float s[10][256];   // radius | x y (interleaved: 2 rows) | depth | conic (3 rows, interleaved) | comp | op | tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t i0 = (size_t)blockIdx.x * 256, i = i0 + tid;
    float acc = 0.f;
    if (i < n) {
        const F3 m = *reinterpret_cast<const F3 *>(means + i * 3);
        const f4 q = quats[i];
        const F3 sc = *reinterpret_cast<const F3 *>(scales + i * 3);
        acc = m.x + m.y * m.z + q.x * q.y + q.z * q.w + sc.x * sc.y + sc.z + opac[i];
    }
    s[0][tid] = acc;
    (&s[1][0])[tid * 2] = acc; (&s[1][0])[tid * 2 + 1] = acc + 1.f;
    s[3][tid] = acc;
    (&s[4][0])[tid * 3] = acc; (&s[4][0])[tid * 3 + 1] = acc; (&s[4][0])[tid * 3 + 2] = acc;
    s[7][tid] = acc; s[8][tid] = acc; s[9][tid] = acc;
    __syncthreads();
    if (i0 + 256 > n) return;   // (scratch: full blocks only)
    // 10 KB = 640 float4: wave w writes rows {0,3 | 1,2 | 4,5,6 | 7,8,9}
    auto put = [&](void *dst, const float *row, int n4) {
        for (int k = lane; k < n4; k += 64) reinterpret_cast<f4 *>(dst)[k] = reinterpret_cast<const f4 *>(row)[k];
    };
    if (wave == 0) { put(radii + i0, s[0], 64); put(depths + i0, s[3], 64); }
    if (wave == 1) put(m2d + i0, s[1], 128);
    if (wave == 2) put(conics + i0 * 3, s[4], 192);
    if (wave == 3) { put(comps + i0, s[7], 64); put(opeff + i0, s[8], 64); put(tiles + i0, s[9], 64); }
}

static void timeit(const char *name, size_t bytes, const std::function<void()> &f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %7.1f us  %6.2f TB/s\n", name, ms / 20 * 1000, bytes / (ms / 20 * 1e-3) / 1e12);
}

int main() {
    const size_t n = 2000000;
    float *means, *scales, *opac, *depths, *conics, *comps, *opeff; f4 *quats; int *radii, *tiles; float2 *m2d;
    hipMalloc(&means, n * 12); hipMalloc(&scales, n * 12); hipMalloc(&opac, n * 4); hipMalloc(&quats, n * 16);
    hipMalloc(&radii, n * 4); hipMalloc(&tiles, n * 4); hipMalloc(&m2d, n * 8); hipMalloc(&depths, n * 4); hipMalloc(&conics, n * 12);
    hipMalloc(&comps, n * 4); hipMalloc(&opeff, n * 4);
    hipMemset(means, 0, n * 12); hipMemset(scales, 0, n * 12); hipMemset(opac, 0, n * 4); hipMemset(quats, 0, n * 16);
    const unsigned g = (unsigned)((n + 255) / 256);
#define ARGS n, means, quats, scales, opac, radii, m2d, depths, conics, comps, opeff, tiles
    timeit("direct (44 in + 40 out)", n * 84, [&] { direct<1, 1><<<g, 256>>>(ARGS); });
    timeit("wide stores", n * 84, [&] { wide<<<g, 256>>>(ARGS); });
    timeit("loads only (44)", n * 44, [&] { direct<1, 0><<<g, 256>>>(ARGS); });
    timeit("stores only (40)", n * 40, [&] { direct<0, 1><<<g, 256>>>(ARGS); });
    timeit("direct again", n * 84, [&] { direct<1, 1><<<g, 256>>>(ARGS); });
    return 0;
}
