// wave_reduce.hpp -- transposed wave64 reduction for gfx950.
//
// Summing K per-lane values over the 64 lanes one value at a time costs 6 DPP adds each.  Here
// 4*NR values are reduced together: v_permlane16_swap / v_permlane32_swap (gfx950) exchange half of
// the lanes of TWO registers in one instruction, so each cross-row step halves the number of live
// registers; only the last 4 in-row steps (DPP quad_perm / row_half_mirror / row_mirror) are paid
// per surviving register.  12 values: 6+6 + 3+3 + 3*4 = 30 VALU instructions instead of 72.
//
// Result layout: out[i], row r (lanes 16r..16r+15, every lane of the row) = sum over the wave of
// in[4*i + r].
#pragma once
#include <hip/hip_runtime.h>

// The clang builtins __builtin_amdgcn_permlane{16,32}_swap of ROCm 7.2 return the first result in
// both vector elements, so the instructions are issued directly.  The s_nop covers the VALU-write ->
// permlane-swap-read hazard the compiler would otherwise pad (it cannot see inside the asm).
__device__ __forceinline__ void permlane16_swap(float &a, float &b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane32_swap(float &a, float &b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// in: 4*NR values per lane (clobbered).  out[NR] as described above.
template <int NR>
__device__ __forceinline__ void wave_reduce_x4(float *in, float *out) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
        // rows: a -> [a01, b01, a23, b23]
        permlane16_swap(a, b);
        a += b;
        permlane16_swap(c, d);
        c += d;
        // [a01,b01,a23,b23] x [c01,d01,c23,d23] -> [a, b, c, d] partial sums, one value per row
        permlane32_swap(a, c);
        a += c;
        a += dpp_perm<0xB1>(a);   // quad_perm [1,0,3,2]
        a += dpp_perm<0x4E>(a);   // quad_perm [2,3,0,1]
        a += dpp_perm<0x141>(a);  // row_half_mirror
        a += dpp_perm<0x140>(a);  // row_mirror
        out[i] = a;
    }
}
