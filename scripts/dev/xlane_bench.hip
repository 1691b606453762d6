// Development microbenchmark (round 6): issue cost of the CROSS-LANE instructions of the compositing backward's reduction on gfx950
// -- v_permlane16_swap / v_permlane32_swap, DPP adds (quad_perm, row_ror, row_half_mirror, row_bcast), ds_swizzle, ds_bpermute --
// against a plain v_add_f32, in shader cycles per instruction per SIMD at 4 and 8 waves per SIMD (wall time x 2.1 GHz, s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o xlane_bench scripts/dev/xlane_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int MODE>
__global__ __launch_bounds__(64) void xl(int iters, float *out, unsigned long long *cyc) {
    float a = threadIdx.x * 0.001f, b = a + 1.f, c = a + 2.f, d = a + 3.f, e = a + 4.f, f = a + 5.f, g = a + 6.f, h = a + 7.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 16 independent v_add_f32 (8 chains x 2)
            REP4(asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %4, %4, %5\n\tv_add_f32 %6, %6, %7"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (MODE == 1) {   // 16 v_permlane16_swap on 4 independent register pairs
            REP4(asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (MODE == 2) {   // 16 v_permlane32_swap
            REP4(asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (MODE == 3) {   // 16 DPP adds, quad_perm, independent (dst != src chains: a += perm(b) ...)
            REP4(asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                              "v_add_f32_dpp %2, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                              "v_add_f32_dpp %4, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                              "v_add_f32_dpp %6, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (MODE == 4) {   // 16 DPP adds, row_ror:8
            REP4(asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                              "v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                              "v_add_f32_dpp %4, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                              "v_add_f32_dpp %6, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (MODE == 5) {   // 16 DPP adds, row_bcast:15 (gfx9 cross-row broadcast)
            REP4(asm volatile("v_add_f32_dpp %0, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                              "v_add_f32_dpp %2, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                              "v_add_f32_dpp %4, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                              "v_add_f32_dpp %6, %7, %7 row_bcast:15 row_mask:0xa bank_mask:0xf"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (MODE == 6) {   // 16 ds_swizzle (LDS crossbar, no memory): swap 16-lane halves of 32 = xor 0x10
            REP4(asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM,\"1pppp\")\n\tds_swizzle_b32 %1, %1 offset:swizzle(BITMASK_PERM,\"1pppp\")\n\t"
                              "ds_swizzle_b32 %2, %2 offset:swizzle(BITMASK_PERM,\"1pppp\")\n\tds_swizzle_b32 %3, %3 offset:swizzle(BITMASK_PERM,\"1pppp\")\n\ts_waitcnt lgkmcnt(0)"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (MODE == 7) {   // 16 v_mov_b32_dpp row_half_mirror
            REP4(asm volatile("v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                              "v_mov_b32_dpp %4, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %6, %7 row_half_mirror row_mask:0xf bank_mask:0xf"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (MODE == 8) {   // 16 v_add with s_nop 1 in front of each (the hazard padding of the reduction)
            REP4(asm volatile("s_nop 1\n\tv_add_f32 %0, %0, %1\n\ts_nop 1\n\tv_add_f32 %2, %2, %3\n\ts_nop 1\n\tv_add_f32 %4, %4, %5\n\ts_nop 1\n\tv_add_f32 %6, %6, %7"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (MODE == 9) {   // 16 x v_readlane_b32 -> SGPR (another way across lanes)
            REP4(asm volatile("v_readlane_b32 s20, %0, 5\n\tv_readlane_b32 s21, %1, 17\n\tv_readlane_b32 s22, %2, 33\n\tv_readlane_b32 s23, %3, 60"
                              : : "v"(a), "v"(b), "v"(c), "v"(d) : "s20", "s21", "s22", "s23");)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = a + b + c + d + e + f + g + h;
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, float *buf, unsigned long long *cyc, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 8}) {
        const int waves = 1024 * wps;
        xl<MODE><<<waves, 64>>>(iters / 4, buf, cyc);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        xl<MODE><<<waves, 64>>>(iters, buf, cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(waves);
        (void)hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= waves;
        const double per_simd = (double)iters * 16 * wps;
        printf("%-34s waves/SIMD %d: %8.1f us   cycles/instr/SIMD (wall x 2.1 GHz) %6.2f   one wave: %6.2f cycles between its instructions\n", name, wps,
               ms * 1e3, ms * 1e-3 * 2.1e9 / per_simd, mean / ((double)iters * 16));
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float *buf; unsigned long long *cyc;
    (void)hipMalloc(&buf, 1024); (void)hipMalloc(&cyc, 8192 * 8);
    run<0>("v_add_f32", buf, cyc, iters);
    run<8>("s_nop 1 + v_add_f32", buf, cyc, iters);
    run<1>("v_permlane16_swap_b32", buf, cyc, iters);
    run<2>("v_permlane32_swap_b32", buf, cyc, iters);
    run<3>("v_add_f32_dpp quad_perm", buf, cyc, iters);
    run<4>("v_add_f32_dpp row_ror:8", buf, cyc, iters);
    run<5>("v_add_f32_dpp row_bcast:15", buf, cyc, iters);
    run<7>("v_mov_b32_dpp row_half_mirror", buf, cyc, iters);
    run<6>("ds_swizzle_b32 (x4 + wait)", buf, cyc, iters);
    run<9>("v_readlane_b32", buf, cyc, iters);
    return 0;
}
