// Development microbenchmark: cost of the compositing backward's gradient atomics on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_bench atomic_bench.hip
// Each wave issues ITER atomic instructions with `lanes` active lanes; rows are pseudo-random in [0, rows).
//   mode 0: SoA  -- lane j adds to array j (12 separate arrays, one per component group as in blend_bwd: 2,2,3,1,3,1)
//   mode 1: AoS  -- lane j adds to row*16 + j (one 64-byte line per row)
//   mode 2: AoS, plain stores (no atomics) for reference
//   mode 3: SoA, every wave-instruction uses 64 lanes on 64 different rows of ONE array (classic scatter)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

__global__ __launch_bounds__(64) void k(int mode, int iters, int lanes, uint32_t rows, float *buf, int spread) {
    const int lane = threadIdx.x;
    const uint32_t wave = blockIdx.x;
    // SoA arrays: xy(2) abs(2) conic(3) opac(1) col(3) depth(1)
    const int grp_off[12] = {0, 0, 1, 1, 2, 2, 2, 3, 4, 4, 4, 5};
    const int grp_idx[12] = {0, 1, 0, 1, 0, 1, 2, 0, 0, 1, 2, 0};
    const int grp_w[6] = {2, 2, 3, 1, 3, 1};
    size_t arr_base[6];
    size_t o = 0;
    for (int g = 0; g < 6; ++g) { arr_base[g] = o; o += (size_t)rows * grp_w[g]; }
    for (int it = 0; it < iters; ++it) {
        // rows of neighbouring waves overlap (spread) like neighbouring tiles share Gaussians
        const uint32_t row = hash32((wave / spread) * 7919u + it) % rows;
        if (mode == 0) {
            if (lane < lanes) {
                const int g = grp_off[lane];
                unsafeAtomicAdd(buf + arr_base[g] + (size_t)row * grp_w[g] + grp_idx[lane], 1.0f);
            }
        } else if (mode == 1) {
            if (lane < lanes) unsafeAtomicAdd(buf + (size_t)row * 16 + lane, 1.0f);
        } else if (mode == 2) {
            if (lane < lanes) buf[(size_t)row * 16 + lane] = 1.0f;
        } else {
            const uint32_t r2 = hash32(row * 64u + lane) % rows;
            unsafeAtomicAdd(buf + r2, 1.0f);
        }
    }
}

int main(int argc, char **argv) {
    const uint32_t rows = 300000;
    const int waves = 8160, iters = 256;
    float *buf;
    hipMalloc(&buf, (size_t)rows * 16 * 4);
    hipMemset(buf, 0, (size_t)rows * 16 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spread : {1, 4, 16}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int lanes : {12, 4, 1}) {
                if (mode == 3 && lanes != 12) continue;
                k<<<waves, 64>>>(mode, iters, lanes, rows, buf, spread);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                k<<<waves, 64>>>(mode, iters, lanes, rows, buf, spread);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double instr = (double)waves * iters;
                printf("spread %2d mode %d lanes %2d: %8.1f us  %7.2f ns/instr/chip  %6.1f G lane-ops/s\n", spread, mode, lanes,
                       ms * 1e3, ms * 1e6 / instr, instr * (mode == 3 ? 64 : lanes) / (ms * 1e6));
            }
        }
    }
    return 0;
}
