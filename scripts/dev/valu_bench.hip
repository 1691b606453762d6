// Development microbenchmark: VALU issue rate on MI355X (cycles per wave64 instruction per SIMD) as a
// function of waves per SIMD and of the instruction mix (fma / exp / rcp / pk_fma / dependent chain).
//   hipcc --offload-arch=gfx950 -O3 -o valu_bench valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(int iters, float *out) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float b = 1.0001f, c = 0.0001f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // 8 independent fma chains
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (MODE == 1) { // one dependent chain
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        } else if (MODE == 2) { // exp2, independent
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == 3) { // packed fma on register pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 *p = reinterpret_cast<f2 *>(a);
            f2 bb = {b, b}, cc = {c, c};
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(bb), "v"(cc));
        } else if (MODE == 4) { // v_cmp + cndmask mix, independent
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (MODE == 5) { // rcp
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char *name, float *buf) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        const int waves = 1024 * wps;
        k<MODE><<<waves, 64>>>(iters, buf);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MODE><<<waves, 64>>>(iters, buf);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)wps * iters * 32;
        printf("%-10s waves/SIMD %d: %8.1f us  %.2f cycles/instr/SIMD @2.4GHz\n", name, wps, ms * 1e3,
               ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
}

int main() {
    float *buf;
    hipMalloc(&buf, 1024);
    run<0>("fma_indep", buf);
    run<1>("fma_chain", buf);
    run<2>("exp_indep", buf);
    run<3>("pk_fma", buf);
    run<4>("mul_indep", buf);
    run<5>("rcp_indep", buf);
    return 0;
}
