// common.hpp -- shared host/device helpers for libmtgs_rast.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mtgs_rast.h"

#define MTGS_WAVE 64

// Thread-local error message (mtgs_rast_last_error).  Defined in abi.hip.
void mtgs_set_error(const char *fmt, ...);

#define MTGS_REQUIRE(cond, code, ...) \
    do {                              \
        if (!(cond)) {                \
            mtgs_set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

// Launch errors are surfaced without synchronising the device.
#define MTGS_CHECK_LAUNCH(name)                                              \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            mtgs_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return MTGS_ELAUNCH;                                             \
        }                                                                    \
    } while (0)

static inline __host__ __device__ int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Zero-fill of control words that KERNELS then update with atomics / flags, as a kernel of this library (zero.hip) --
// not hipMemsetAsync: inside a captured HIP graph a memset node may run on a copy engine, whose writes bypass the
// XCDs' L2s, and graph nodes are not separated by the cache maintenance that separates eager launches; stale L2 lines
// of the "zeroed" words then survive into the kernels (observed: look-back flags of the previous replay -> wild
// scatter addresses -> memory fault on replay).  `bytes` and `p` must be multiples of 4.
int mtgs_zero_async(void *p, size_t bytes, hipStream_t stream);

#ifdef __HIPCC__
// ---- wave64 cross-lane helpers (DPP; CDNA has row_bcast) ---------------------------------------
// v_add_f32 with a DPP-permuted operand is ONE VALU instruction; a full 64-lane sum is 6 of them
// and lands in lane 63.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, BOUND));
}
// Sum over the wave; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_mov<0xB1>(v);               // quad_perm [1,0,3,2]  : xor 1
    v += dpp_mov<0x4E>(v);               // quad_perm [2,3,0,1]  : xor 2
    v += dpp_mov<0x141>(v);              // row_half_mirror      : 8 lanes
    v += dpp_mov<0x140>(v);              // row_mirror           : 16 lanes (a row)
    v += dpp_mov<0x142, 0xa, 0xf, false>(v); // row_bcast15 -> rows 1,3
    v += dpp_mov<0x143, 0xc, 0xf, false>(v); // row_bcast31 -> rows 2,3
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
#endif
