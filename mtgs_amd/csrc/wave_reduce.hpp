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

// ---- packed variant: the in-row steps are transposed too ------------------------------------------------------------
// After the cross-row swaps register i holds, in row r, 16 per-lane partial sums of value 4*i + r.  Summing each
// register separately costs 4 DPP adds per register.  Instead two registers share one (lanes 0-7 of a row keep X,
// lanes 8-15 keep Y: v_add_dpp with a bank mask), and at the next level two such registers share the quads, so that
// ONE register finishes with one value per quad:
//      quad 0 (lanes 0-3 of the row): register 0     quad 2 (lanes 8-11):  register 1
//      quad 1 (lanes 4-7):            register 2     quad 3 (lanes 12-15): register 3
// 12 values: 6+6 cross-row + 7 in-row instead of + 12;  16 values: 8+8 + 8 instead of + 16.
// packed_component(lane) gives the value index 4*i + r a lane ends up holding (every lane of a quad holds the same).
__device__ __forceinline__ void row_pair(float &dst, float y) {
    // dst (banks 2,3 = lanes 8-15 of every row) = y + y rotated by 8 lanes; lanes 0-7 keep dst
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(dst) : "v"(y));
}
__device__ __forceinline__ void quad_pair(float &dst, float w) {
    // dst (banks 1,3 = quads 1 and 3 of every row) = w + half-mirrored w; quads 0 and 2 keep dst
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xa" : "+v"(dst) : "v"(w));
}
__device__ __forceinline__ int packed_component(int lane) {
    const int q = (lane >> 2) & 3, row = lane >> 4;
    const int reg = q == 0 ? 0 : (q == 2 ? 1 : (q == 1 ? 2 : 3));
    return 4 * reg + row;
}
// in: 4*NR values per lane (clobbered), NR = 3 or 4.  Returns the packed register described above.
template <int NR>
__device__ __forceinline__ float wave_reduce_x4_packed(float *in) {
    static_assert(NR == 3 || NR == 4, "packed reduction handles 12 or 16 values");
    float reg[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
        permlane16_swap(a, b);
        a += b;
        permlane16_swap(c, d);
        c += d;
        permlane32_swap(a, c);
        reg[i] = a + c;
    }
    // level 1: halves of a row
    float w01 = reg[0] + dpp_perm<0x128>(reg[0]);   // row_ror:8 -> lanes i and i^8 summed, in every lane
    row_pair(w01, reg[1]);                          // lanes 8-15 <- register 1
    float w23;
    if (NR == 4) {
        w23 = reg[2] + dpp_perm<0x128>(reg[2]);
        row_pair(w23, reg[3]);
    } else {
        w23 = reg[2] + dpp_perm<0x128>(reg[2]);     // both halves hold register 2
    }
    // level 2: quads
    float v = w01 + dpp_perm<0x141>(w01);           // row_half_mirror: the 4 distinct pair sums of a half, twice
    quad_pair(v, w23);                              // quads 1 and 3 <- registers 2 / 3
    v += dpp_perm<0xB1>(v);                         // quad_perm [1,0,3,2]
    v += dpp_perm<0x4E>(v);                         // quad_perm [2,3,0,1]
    return v;
}
