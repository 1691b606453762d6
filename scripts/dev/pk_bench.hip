// scratch: issue rate of v_pk_fma_f32 / v_pk_mul_f32 against v_fma_f32 (same number of INSTRUCTIONS per loop, 2x the flops)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *o, int iters, float s) {
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, (float)i};
    f2 c = {s, s * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(c.x), "v"(c.y)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(c.x), "v"(c.y)); }
            if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c)); }
            if (MODE == 2) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
            if (MODE == 3) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i].x)); asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].y)); }
            if (MODE == 4) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(c.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(c.y)); }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    o[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, float *o) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 8;
    k<MODE><<<blocks, 256>>>(o, 10, 1.0001f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(o, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters * 16;   // wave-instructions
    printf("%-28s %.3f ms  %.1f G wave-instr/s  (%.2f cycles per instr per SIMD at 2.4 GHz)\n", name, ms, instr / ms / 1e6,
           2.4e9 * 1024 * ms * 1e-3 / instr);
}
int main() {
    float *o; hipMalloc(&o, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", o); run<1>("v_pk_fma_f32", o); run<2>("v_pk_mul/add_f32", o); run<3>("v_exp/v_rcp_f32", o); run<4>("v_cndmask/v_mul", o);
    return 0;
}
