// Standalone check of the transposed wave reduction used by blend_bwd (dev tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../mtgs_amd/csrc/wave_reduce.hpp"

__global__ void k(const float* in, float* out) {  // in[12][64] -> out[64][3]
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = in[i * 64 + threadIdx.x];
    float r[3];
    wave_reduce_x4<3>(v, r);
    for (int i = 0; i < 3; ++i) out[threadIdx.x * 3 + i] = r[i];
}
template <int NR>
__global__ void kp(const float* in, float* out) {  // in[4*NR][64] -> out[64]
    float v[4 * NR];
    for (int i = 0; i < 4 * NR; ++i) v[i] = in[i * 64 + threadIdx.x];
    out[threadIdx.x] = wave_reduce_x4_packed<NR>(v);
}
template <int NR>
int test_packed() {
    float h[4 * NR * 64], *d, *o, ho[64];
    for (int i = 0; i < 4 * NR * 64; ++i) h[i] = (float)(rand() % 1000) / 7.0f;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    kp<NR><<<1, 64>>>(d, o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int q = (l >> 2) & 3, row = l >> 4;
        const int reg = q == 0 ? 0 : (q == 2 ? 1 : (q == 1 ? 2 : 3));
        if (reg >= NR) continue;
        const int val = 4 * reg + row;
        double s = 0; for (int k2 = 0; k2 < 64; ++k2) s += h[val * 64 + k2];
        if (fabs(ho[l] - s) > 1e-2) { if (bad < 10) printf("packed<%d> lane %d val %d got %f want %f\n", NR, l, val, ho[l], s); ++bad; }
    }
    printf(bad ? "packed<%d> FAIL %d\n" : "packed<%d> OK\n", NR, bad);
    return bad;
}
int main() {
    if (test_packed<3>() || test_packed<4>()) return 1;
    float h[12 * 64], *d, *o, ho[64 * 3];
    for (int i = 0; i < 12 * 64; ++i) h[i] = (float)(rand() % 1000) / 7.0f;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int val = 0; val < 12; ++val) {
        double s = 0; for (int l = 0; l < 64; ++l) s += h[val * 64 + l];
        int reg = val / 4, row = val % 4;
        for (int l = row * 16; l < row * 16 + 16; ++l) {
            float got = ho[l * 3 + reg];
            if (fabs(got - s) > 1e-2) { if (bad < 10) printf("val %d lane %d got %f want %f\n", val, l, got, s); ++bad; }
        }
    }
    printf(bad ? "FAIL %d\n" : "OK\n", bad);
    return bad != 0;
}
