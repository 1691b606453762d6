// sh.hip -- real spherical harmonics (degree <= 4) colour evaluation, forward and backward.
//
// Replaces gsplat 1.4.0 compute_sh_fwd / compute_sh_bwd, reached from MTGS through
// gsplat.cuda._wrapper.spherical_harmonics (/root/reference/mtgs/scene_model/gaussian_model/
// vanilla_gaussian_splatting.py:312-318 and siblings).
//
// Roofline: HBM.  Algorithmic bytes per Gaussian: fwd 12 (dir) + 12*nb (active coeffs) + 12 (out);
// bwd 12 + 12 (v_colors) + 12*K (v_coeffs written in full, zeros above the active degree).
//
// CDNA4 mapping: coeffs are [n,K,3] AoS, i.e. one Gaussian's coefficients are 12*K contiguous
// bytes.  One thread per Gaussian reading its own row would make every wave load touch 64
// different lines; instead a block streams the rows of its 128 Gaussians through LDS with
// fully-coalesced (float4 when the row is a multiple of 16 B) accesses, and each lane then walks
// its row in LDS with an ODD row stride (bank = (lane*stride + j) mod 32 -> conflict-free).
#include "common.hpp"
#include "sh_lane.hpp"

namespace {

constexpr int SH_BLOCK = 128;

__device__ __forceinline__ void sh_bases_dev(int degree, float x, float y, float z, float *b) {
    b[0] = 0.2820947917738781f;
    if (degree < 1) return;
    b[1] = -0.48860251190292f * y;
    b[2] = 0.48860251190292f * z;
    b[3] = -0.48860251190292f * x;
    if (degree < 2) return;
    float z2 = z * z;
    float fTmp0B = -1.092548430592079f * z;
    float fC1 = x * x - y * y;
    float fS1 = 2.f * x * y;
    b[4] = 0.5462742152960395f * fS1;
    b[5] = fTmp0B * y;
    b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    b[7] = fTmp0B * x;
    b[8] = 0.5462742152960395f * fC1;
    if (degree < 3) return;
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    float fTmp1B = 1.445305721320277f * z;
    float fC2 = x * fC1 - y * fS1;
    float fS2 = x * fS1 + y * fC1;
    b[9] = -0.5900435899266435f * fS2;
    b[10] = fTmp1B * fS1;
    b[11] = fTmp0C * y;
    b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
    b[13] = fTmp0C * x;
    b[14] = fTmp1B * fC1;
    b[15] = -0.5900435899266435f * fC2;
    if (degree < 4) return;
    float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    float fTmp2B = -1.770130769779931f * z;
    float fC3 = x * fC2 - y * fS2;
    float fS3 = x * fS2 + y * fC2;
    b[16] = 0.6258357354491763f * fS3;
    b[17] = fTmp2B * fS2;
    b[18] = fTmp1C * fS1;
    b[19] = fTmp0D * y;
    b[20] = 1.984313483298443f * z * b[12] - 1.006230589874905f * b[6];
    b[21] = fTmp0D * x;
    b[22] = fTmp1C * fC1;
    b[23] = fTmp2B * fC2;
    b[24] = 0.6258357354491763f * fC3;
}

// d(basis_k)/d(x,y,z), contracted on the fly with s[k] = <coeffs[k,:], v_color>.
template <int DEG>
__device__ __forceinline__ void sh_dir_grad(float x, float y, float z, const float *s, float &vx,
                                            float &vy, float &vz) {
    vx = vy = vz = 0.f;
    if (DEG < 1) return;
    vy += -0.48860251190292f * s[1];
    vz += 0.48860251190292f * s[2];
    vx += -0.48860251190292f * s[3];
    if (DEG < 2) return;
    float z2 = z * z;
    float fTmp0B = -1.092548430592079f * z, fTmp0B_z = -1.092548430592079f;
    float fC1 = x * x - y * y, fC1_x = 2.f * x, fC1_y = -2.f * y;
    float fS1 = 2.f * x * y, fS1_x = 2.f * y, fS1_y = 2.f * x;
    vx += 0.5462742152960395f * fS1_x * s[4]; vy += 0.5462742152960395f * fS1_y * s[4];
    vy += fTmp0B * s[5]; vz += fTmp0B_z * y * s[5];
    float pSH6_z = 2.f * 0.9461746957575601f * z;
    vz += pSH6_z * s[6];
    vx += fTmp0B * s[7]; vz += fTmp0B_z * x * s[7];
    vx += 0.5462742152960395f * fC1_x * s[8]; vy += 0.5462742152960395f * fC1_y * s[8];
    if (DEG < 3) return;
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f, fTmp0C_z = -2.285228997322329f * 2.f * z;
    float fTmp1B = 1.445305721320277f * z, fTmp1B_z = 1.445305721320277f;
    float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    float fC2_x = fC1 + x * fC1_x - y * fS1_x, fC2_y = x * fC1_y - fS1 - y * fS1_y;
    float fS2_x = fS1 + x * fS1_x + y * fC1_x, fS2_y = x * fS1_y + fC1 + y * fC1_y;
    vx += -0.5900435899266435f * fS2_x * s[9]; vy += -0.5900435899266435f * fS2_y * s[9];
    vx += fTmp1B * fS1_x * s[10]; vy += fTmp1B * fS1_y * s[10]; vz += fTmp1B_z * fS1 * s[10];
    vy += fTmp0C * s[11]; vz += fTmp0C_z * y * s[11];
    float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    float pSH12_z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
    vz += pSH12_z * s[12];
    vx += fTmp0C * s[13]; vz += fTmp0C_z * x * s[13];
    vx += fTmp1B * fC1_x * s[14]; vy += fTmp1B * fC1_y * s[14]; vz += fTmp1B_z * fC1 * s[14];
    vx += -0.5900435899266435f * fC2_x * s[15]; vy += -0.5900435899266435f * fC2_y * s[15];
    if (DEG < 4) return;
    float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float fTmp0D_z = 3.f * -4.683325804901025f * z2 + 2.007139630671868f;
    float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f, fTmp1C_z = 2.f * 3.31161143515146f * z;
    float fTmp2B = -1.770130769779931f * z, fTmp2B_z = -1.770130769779931f;
    float fC3_x = fC2 + x * fC2_x - y * fS2_x, fC3_y = x * fC2_y - fS2 - y * fS2_y;
    float fS3_x = fS2 + x * fS2_x + y * fC2_x, fS3_y = x * fS2_y + fC2 + y * fC2_y;
    vx += 0.6258357354491763f * fS3_x * s[16]; vy += 0.6258357354491763f * fS3_y * s[16];
    vx += fTmp2B * fS2_x * s[17]; vy += fTmp2B * fS2_y * s[17]; vz += fTmp2B_z * fS2 * s[17];
    vx += fTmp1C * fS1_x * s[18]; vy += fTmp1C * fS1_y * s[18]; vz += fTmp1C_z * fS1 * s[18];
    vy += fTmp0D * s[19]; vz += fTmp0D_z * y * s[19];
    vz += (1.984313483298443f * (pSH12 + z * pSH12_z) - 1.006230589874905f * pSH6_z) * s[20];
    vx += fTmp0D * s[21]; vz += fTmp0D_z * x * s[21];
    vx += fTmp1C * fC1_x * s[22]; vy += fTmp1C * fC1_y * s[22]; vz += fTmp1C_z * fC1 * s[22];
    vx += fTmp2B * fC2_x * s[23]; vy += fTmp2B * fC2_y * s[23]; vz += fTmp2B_z * fC2 * s[23];
    vx += 0.6258357354491763f * fC3_x * s[24]; vy += 0.6258357354491763f * fC3_y * s[24];
}

// Stream the active part (NB3 floats) of each Gaussian's coefficient row into LDS.
template <int NB3, int STRIDE>
__device__ __forceinline__ void stage_coeffs(float *lds, const float *__restrict__ coeffs,
                                             const uint8_t *__restrict__ masks, int64_t g0, int cnt,
                                             int K) {
    const int tid = threadIdx.x;
    if ((NB3 % 4 == 0) && K * 3 == NB3) {  // rows contiguous and 16-B aligned: float4 stream
        const float4 *src = reinterpret_cast<const float4 *>(coeffs + g0 * NB3);
        const int n4 = cnt * (NB3 / 4);
        for (int i4 = tid; i4 < n4; i4 += SH_BLOCK) {
            const int g = i4 / (NB3 / 4), j = (i4 % (NB3 / 4)) * 4;
            if (masks && !masks[g0 + g]) continue;
            const float4 v = src[i4];
            float *d = lds + g * STRIDE + j;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        const int total = cnt * NB3;
        for (int i = tid; i < total; i += SH_BLOCK) {
            const int g = i / NB3, j = i % NB3;
            if (masks && !masks[g0 + g]) continue;
            lds[g * STRIDE + j] = coeffs[(g0 + g) * (int64_t)K * 3 + j];
        }
    }
}

// ---- the colour activation of the callers, fused (round 6) --------------------------------------------------------------------
// MTGS turns the SH output into a colour with `torch.clamp(rgbs + 0.5, 0.0, 1.0)` (vanilla_gaussian_splatting.py:318, and the same
// line in the multi-colour / rigid / deformable nodes), gsplat's own sh_degree path with `clamp_min(colors + 0.5, 0)`
// (rendering.py): two more passes over [N, 3] forward and four backward, all launch-bound (67 us of a 0.93 ms step).  With `act.pass`
// given the forward kernels write y = clamp(x + add, lo, hi) -- the same two fp32 operations in the same order, NaN propagating as
// torch.clamp does -- and one byte per Gaussian whose bit c says "channel c passes its cotangent" (lo <= x + add <= hi: torch's
// clamp backward); the backward kernels mask the incoming cotangent with it.  The Python layer decides when the caller's
// expression is exactly this one (mtgs_amd/wrapper.py::_LazySH).
struct ShAct {
    uint8_t *pass;     // nullable: no activation
    float add, lo, hi;
    int has_add;       // 0: y = clamp(x) (no `+ 0.0`: the sign of a zero would change)
};
__device__ __forceinline__ float sh_act_apply(const ShAct &a, float x, int c, unsigned &bits) {
#pragma clang fp contract(off)
    const float v = a.has_add ? x + a.add : x;
    if (v >= a.lo && v <= a.hi) bits |= 1u << c;
    return v < a.lo ? a.lo : (v > a.hi ? a.hi : v);       // (a NaN fails both compares and stays a NaN, as in torch.clamp)
}

template <int DEG>
__global__ __launch_bounds__(SH_BLOCK) void sh_fwd_kernel(int64_t n, int K,
                                                         const float *__restrict__ dirs,
                                                         const float *__restrict__ coeffs,
                                                         const uint8_t *__restrict__ masks,
                                                         float *__restrict__ colors, const ShAct act) {
    constexpr int NB = (DEG + 1) * (DEG + 1), NB3 = NB * 3, STRIDE = NB3 | 1;
    __shared__ float lds[SH_BLOCK * STRIDE];
    const int64_t g0 = (int64_t)blockIdx.x * SH_BLOCK;
    const int cnt = (int)min((int64_t)SH_BLOCK, n - g0);
    stage_coeffs<NB3, STRIDE>(lds, coeffs, masks, g0, cnt, K);
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid >= cnt) return;
    const int64_t g = g0 + tid;
    float r = 0.f, gg = 0.f, bb = 0.f;
    if (!masks || masks[g]) {
        float x = dirs[g * 3], y = dirs[g * 3 + 1], z = dirs[g * 3 + 2];
        const float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
        x *= inorm; y *= inorm; z *= inorm;
        float b[NB];
        sh_bases_dev(DEG, x, y, z, b);
        const float *c = lds + tid * STRIDE;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            r += b[k] * c[k * 3];
            gg += b[k] * c[k * 3 + 1];
            bb += b[k] * c[k * 3 + 2];
        }
    }
    if (act.pass) {
        unsigned bits = 0u;
        r = sh_act_apply(act, r, 0, bits); gg = sh_act_apply(act, gg, 1, bits); bb = sh_act_apply(act, bb, 2, bits);
        act.pass[g] = (uint8_t)bits;
    }
    colors[g * 3] = r; colors[g * 3 + 1] = gg; colors[g * 3 + 2] = bb;
}

// ---- K == 16 forward: one 16-lane DPP row per Gaussian, no LDS ---------------------------------
// MTGS allocates K = 16 coefficients per Gaussian (sh_degree 3) from step 0.  With K = 16, lane (g, k) of a
// wave = basis k of Gaussian g: the 64 lanes of a load instruction read 64 x 12 B = 768 CONTIGUOUS bytes
// (4 whole rows), nothing is staged through LDS, there is no barrier, and the sum over k is a 4-step DPP
// butterfly inside the 16-lane row.  Each lane evaluates only ITS basis function: real SH factor as
//   b_k = (a0 + a1 z + a2 z^2 + a3 z^3) * s_k,  s_k in {1, x, y, 2xy, x^2-y^2, fS2, fC2}
// (associated Legendre polynomial in z times the azimuthal factor), with per-lane constants a0..a3.
// Measured 125 -> see DESIGN.md table (sh_fwd_k16_kernel).
// The addend must be a ROUNDED product (mul_rounded at the call sites; HIP's __fmul_rn is a plain, contractable `*`): if the compiler contracts the multiply into the
// first add, lane L gets fma(b_L, c_L, round(b_M c_M)) and its partner M the mirror image -- 1 ulp apart, which made a
// Gaussian's colour depend on its position in the wave (tests/test_gpu_large.py renders a subset bit-identically).
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, in every lane
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}
// A wave owns 64 consecutive Gaussians.  Directions, masks and results use the lane-per-Gaussian mapping (768
// contiguous bytes per instruction; the direction is normalised ONCE per Gaussian), the coefficients the row mapping:
// in step `it` row `sub` works on Gaussian 4*it + sub, whose direction it fetches from lane 4*it + sub by
// ds_bpermute; lane k of a row keeps the result of step k, which the owner lane l = 4k + sub reads back from lane
// 16*(l % 4) + l / 4.  All 16 coefficient loads of a lane are issued before the first is used.
constexpr int SH16_BLOCK = 256, SH16_STEPS = 16, SH16_PER_WAVE = 64;
struct F3 { float x, y, z; };
template <int DEG>
__global__ __launch_bounds__(SH16_BLOCK) void sh_fwd_k16_kernel(int64_t n, const float *__restrict__ dirs,
                                                                const float *__restrict__ coeffs,
                                                                const uint8_t *__restrict__ masks,
                                                                float *__restrict__ colors, const ShAct act) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const int lane = threadIdx.x & 63, k = lane & 15, sub = lane >> 4;
    const ShLaneConst lc = sh_lane_const(k);
    const bool active = k < NB;
    const int64_t g0 = ((int64_t)blockIdx.x * (SH16_BLOCK / 64) + (threadIdx.x >> 6)) * SH16_PER_WAVE;
    if (g0 >= n) return;
    // ---- lane-per-Gaussian: direction (normalised once) and mask
    const int64_t gl = g0 + lane;
    const bool onl = gl < n && (!masks || masks[gl]);
    float dx = 0.f, dy = 0.f, dz = 1.f;
    if (onl) {
        const F3 d = *reinterpret_cast<const F3 *>(dirs + gl * 3);
        const float inorm = 1.0f / sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z);
        dx = d.x * inorm; dy = d.y * inorm; dz = d.z * inorm;
    }
    const unsigned long long on_mask = __builtin_amdgcn_ballot_w64(onl);
    // ---- rows
    F3 c[SH16_STEPS];
#pragma unroll
    for (int it = 0; it < SH16_STEPS; ++it) {
        const int gi = it * 4 + sub;
        c[it] = F3{0.f, 0.f, 0.f};
        if (active && ((on_mask >> gi) & 1ull)) {
            // non-temporal: 384 MB of coefficients are read once per frame and fit no cache; a plain load allocates
            // every line in the 256 MB Infinity Cache and has to push out the dirty lines the previous kernels left
            // there first (inside the training step this kernel ran at 4.0 TB/s, alone at 5.8; with `nt` 6.1 TB/s
            // inside the step).  The projection's parameter loads do NOT get the hint: its backward gathers the same
            // rows again and finds them cached (tried: 33 -> 39 us for project_bwd_vis).
            typedef float f3v __attribute__((ext_vector_type(3)));
            const f3v v = __builtin_nontemporal_load(reinterpret_cast<const f3v *>(coeffs + ((g0 + gi) * 16 + k) * 3));
            c[it] = F3{v.x, v.y, v.z};
        }
    }
    float myr = 0.f, myg = 0.f, myb = 0.f;
#pragma unroll
    for (int it = 0; it < SH16_STEPS; ++it) {
        const int src = (it * 4 + sub) << 2;
        const float x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dx)));
        const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dy)));
        const float z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dz)));
        const float b = sh_lane_basis<DEG>(lc, x, y, z);
        const float r = row16_sum(mul_rounded(b, c[it].x)), gg = row16_sum(mul_rounded(b, c[it].y)), bb = row16_sum(mul_rounded(b, c[it].z));
        const bool mine = k == it;
        myr = mine ? r : myr; myg = mine ? gg : myg; myb = mine ? bb : myb;
    }
    const int back = ((lane & 3) * 16 + (lane >> 2)) << 2;
    myr = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(myr)));
    myg = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(myg)));
    myb = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(myb)));
    if (gl < n) {
        F3 o = onl ? F3{myr, myg, myb} : F3{0.f, 0.f, 0.f};
        if (act.pass) {
            unsigned bits = 0u;
            o.x = sh_act_apply(act, o.x, 0, bits); o.y = sh_act_apply(act, o.y, 1, bits); o.z = sh_act_apply(act, o.z, 2, bits);
            act.pass[gl] = (uint8_t)bits;
        }
        *reinterpret_cast<F3 *>(colors + gl * 3) = o;
    }
}

template <int DEG>
__global__ __launch_bounds__(SH_BLOCK) void sh_bwd_kernel(int64_t n, int K,
                                                         const float *__restrict__ dirs,
                                                         const float *__restrict__ coeffs,
                                                         const uint8_t *__restrict__ masks,
                                                         const float *__restrict__ v_colors,
                                                         float *__restrict__ v_coeffs,
                                                         float *__restrict__ v_dirs, const uint8_t *__restrict__ pass) {
    constexpr int NB = (DEG + 1) * (DEG + 1), NB3 = NB * 3, STRIDE = NB3 | 1;
    __shared__ float lds[SH_BLOCK * STRIDE];
    const int64_t g0 = (int64_t)blockIdx.x * SH_BLOCK;
    const int cnt = (int)min((int64_t)SH_BLOCK, n - g0);
    const int tid = threadIdx.x;
    if (v_dirs) {
        stage_coeffs<NB3, STRIDE>(lds, coeffs, masks, g0, cnt, K);
        __syncthreads();
    }
    if (tid < cnt) {
        const int64_t g = g0 + tid;
        float *row = lds + tid * STRIDE;
        const bool on = !masks || masks[g];
        float vdx = 0.f, vdy = 0.f, vdz = 0.f;
        if (on) {
            float x = dirs[g * 3], y = dirs[g * 3 + 1], z = dirs[g * 3 + 2];
            const float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
            x *= inorm; y *= inorm; z *= inorm;
            float b[NB];
            sh_bases_dev(DEG, x, y, z, b);
            float v0 = v_colors[g * 3], v1 = v_colors[g * 3 + 1], v2 = v_colors[g * 3 + 2];
            if (pass) {      // the fused activation's backward: a clamped channel passes no cotangent
                const unsigned pb = pass[g];
                v0 = (pb & 1u) ? v0 : 0.f; v1 = (pb & 2u) ? v1 : 0.f; v2 = (pb & 4u) ? v2 : 0.f;
            }
            if (v_dirs) {
                float s[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    s[k] = (row[k * 3] * v0 + row[k * 3 + 1] * v1) + row[k * 3 + 2] * v2;
                float vx, vy, vz;
                sh_dir_grad<DEG>(x, y, z, s, vx, vy, vz);
                const float dot = (vx * x + vy * y) + vz * z;
                vdx = (vx - dot * x) * inorm; vdy = (vy - dot * y) * inorm; vdz = (vz - dot * z) * inorm;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                row[k * 3] = b[k] * v0; row[k * 3 + 1] = b[k] * v1; row[k * 3 + 2] = b[k] * v2;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB3; ++j) row[j] = 0.f;
        }
        if (v_dirs) { v_dirs[g * 3] = vdx; v_dirs[g * 3 + 1] = vdy; v_dirs[g * 3 + 2] = vdz; }
    }
    __syncthreads();
    // coalesced write of the full [cnt, K*3] block, zeros above the active degree
    const int K3 = K * 3;
    float *dst = v_coeffs + g0 * K3;
    if (K3 % 4 == 0) {
        const int n4 = cnt * (K3 / 4);
        for (int i4 = tid; i4 < n4; i4 += SH_BLOCK) {
            const int g = i4 / (K3 / 4), j = (i4 % (K3 / 4)) * 4;
            const float *s = lds + g * STRIDE;
            float4 v;
            v.x = j < NB3 ? s[j] : 0.f;
            v.y = j + 1 < NB3 ? s[j + 1] : 0.f;
            v.z = j + 2 < NB3 ? s[j + 2] : 0.f;
            v.w = j + 3 < NB3 ? s[j + 3] : 0.f;
            reinterpret_cast<float4 *>(dst)[i4] = v;
        }
    } else {
        const int total = cnt * K3;
        for (int i = tid; i < total; i += SH_BLOCK) {
            const int g = i / K3, j = i % K3;
            dst[i] = j < NB3 ? lds[g * STRIDE + j] : 0.f;
        }
    }
}


// ---- sparse backward: only the rows whose cotangent is non-zero --------------------------------------------------------------
// The colours feed the rasterizer, so dL/dcolour is zero for every Gaussian nothing was composited from -- ~94 % of them at the
// headline workload -- and 12 K bytes of dL/dcoeffs per Gaussian are then zeros, which need neither the directions nor a kernel
// that waits for its loads before it stores (sh_bwd_kernel: 5.3 TB/s against 6.8 TB/s of a pure fill).  With v_coeffs ZERO ALREADY
// (mtgs_fill_zero -- the Python layer issues it on a second stream while the compute-bound compositing backward runs,
// mtgs_amd/wrapper.py::_prefill) this kernel reads the 12-byte cotangents and writes the 12 K-byte rows of the others: one thread
// per Gaussian, its row as 16-byte stores (few lanes of a wave are active, the pieces of a line merge in L2).  Cost grows
// linearly with the density of the cotangent; at 100 % it is the partial-line pattern that measured 195 us at 2M rows.
// One thread per Gaussian finds the rows; a row is then WRITTEN by K3/4 lanes, one 16-byte piece each, so that one store
// instruction of the wave carries 64 / (K3/4) whole rows (5 at K = 16): the products go through a per-wave LDS buffer (in-order DS
// instructions of one wave: no barrier).  With each lane storing its own row, a wave issued 12 store instructions for ~4 active
// lanes -- 20 us of the 36 us this kernel took on 2M rows, 6 % of them with a cotangent (cold; scripts/dev/shrows_bench.py).
template <int DEG>
__global__ __launch_bounds__(256) void sh_bwd_rows_kernel(int64_t n, int K, const float *__restrict__ dirs, const uint8_t *__restrict__ masks,
                                                          const float *__restrict__ v_colors, float *__restrict__ v_coeffs,
                                                          const uint8_t *__restrict__ pass) {
    constexpr int NB = (DEG + 1) * (DEG + 1), NB3 = NB * 3;
    __shared__ __attribute__((aligned(16))) float s_p[4][256];
    __shared__ int s_row[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * 256 + wave * 64;      // first Gaussian of the wave
    const int64_t g = base + lane;
    float v[3] = {0.f, 0.f, 0.f};
    bool nz = false;
    if (g < n) {
        v[0] = v_colors[g * 3]; v[1] = v_colors[g * 3 + 1]; v[2] = v_colors[g * 3 + 2];
        if (pass && (v[0] != 0.f || v[1] != 0.f || v[2] != 0.f)) {      // (the byte is only fetched for a Gaussian with a cotangent)
            const unsigned pb = pass[g];
            v[0] = (pb & 1u) ? v[0] : 0.f; v[1] = (pb & 2u) ? v[1] : 0.f; v[2] = (pb & 4u) ? v[2] : 0.f;
        }
        nz = !(v[0] == 0.f && v[1] == 0.f && v[2] == 0.f) && !(masks && !masks[g]);
    }
    unsigned long long m = __ballot(nz);
    if (m == 0ull) return;
    const int K3 = K * 3;
    const bool coop = K3 % 4 == 0 && K3 <= 256;
    const int pieces = coop ? K3 / 4 : 1, rows_per = coop ? 64 / pieces : 0;
    float b[NB];
    if (nz) {
        float x = dirs[g * 3], y = dirs[g * 3 + 1], z = dirs[g * 3 + 2];
        const float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
        x *= inorm; y *= inorm; z *= inorm;
        sh_bases_dev(DEG, x, y, z, b);
    }
    if (!coop) {
        if (nz) {
            float *dst = v_coeffs + g * K3;
#pragma unroll
            for (int e = 0; e < NB3; ++e) dst[e] = b[e / 3] * v[e % 3];
        }
        return;
    }
    int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));   // rows with work in front
    int left = __popcll(m);
    while (left > 0) {
        const int cnt = min(left, rows_per);
        if (nz && rank >= 0 && rank < cnt) {
            float *row = &s_p[wave][rank * K3];
#pragma unroll
            for (int q = 0; q < (NB3 + 3) / 4; ++q) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int e = 4 * q + i; o[i] = e < NB3 ? b[e / 3] * v[e % 3] : 0.f; }
                *reinterpret_cast<float4 *>(row + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
            }
            for (int q = (NB3 + 3) / 4; q < pieces; ++q) *reinterpret_cast<float4 *>(row + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
            s_row[wave][rank] = lane;
        }
        __builtin_amdgcn_wave_barrier();
        const int slot = lane / pieces, piece = lane - slot * pieces;
        if (slot < cnt) {
            const int64_t gr = base + s_row[wave][slot];
            reinterpret_cast<float4 *>(v_coeffs + gr * K3)[piece] = *reinterpret_cast<const float4 *>(&s_p[wave][slot * K3 + 4 * piece]);
        }
        __builtin_amdgcn_wave_barrier();
        rank -= cnt;
        left -= cnt;
    }
}

}  // namespace

extern "C" int mtgs_sh_fwd(int64_t n, int K, int degree, const float *dirs, const float *coeffs,
                           const uint8_t *masks, float *colors, void *stream) {
    return mtgs_sh_fwd_act(n, K, degree, dirs, coeffs, masks, colors, 0, 0.f, 0.f, 0.f, nullptr, stream);
}

extern "C" int mtgs_sh_fwd_act(int64_t n, int K, int degree, const float *dirs, const float *coeffs, const uint8_t *masks, float *colors,
                               int has_add, float add, float lo, float hi, uint8_t *pass, void *stream) {
    MTGS_REQUIRE(n >= 0 && K > 0, MTGS_EINVAL, "mtgs_sh_fwd: bad sizes n=%lld K=%d", (long long)n, K);
    MTGS_REQUIRE(!pass || lo <= hi, MTGS_EINVAL, "mtgs_sh_fwd_act: lo %g > hi %g", (double)lo, (double)hi);
    const ShAct act{pass, add, lo, hi, has_add ? 1 : 0};
    MTGS_REQUIRE(degree >= 0 && degree <= MTGS_MAX_SH_DEGREE && (degree + 1) * (degree + 1) <= K,
                 MTGS_EINVAL, "mtgs_sh_fwd: degree %d needs (degree+1)^2 <= K=%d and degree <= 4", degree, K);
    if (n == 0) return MTGS_OK;
    MTGS_REQUIRE(dirs && coeffs && colors, MTGS_EINVAL, "mtgs_sh_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (K == 16 && degree <= 3) {
        const unsigned g16 = (unsigned)ceil_div64(n, SH16_PER_WAVE * (SH16_BLOCK / 64));
        switch (degree) {
            case 0: sh_fwd_k16_kernel<0><<<g16, SH16_BLOCK, 0, st>>>(n, dirs, coeffs, masks, colors, act); break;
            case 1: sh_fwd_k16_kernel<1><<<g16, SH16_BLOCK, 0, st>>>(n, dirs, coeffs, masks, colors, act); break;
            case 2: sh_fwd_k16_kernel<2><<<g16, SH16_BLOCK, 0, st>>>(n, dirs, coeffs, masks, colors, act); break;
            default: sh_fwd_k16_kernel<3><<<g16, SH16_BLOCK, 0, st>>>(n, dirs, coeffs, masks, colors, act); break;
        }
        MTGS_CHECK_LAUNCH("mtgs_sh_fwd");
        return MTGS_OK;
    }
    const unsigned grid = (unsigned)ceil_div64(n, SH_BLOCK);
    switch (degree) {
        case 0: sh_fwd_kernel<0><<<grid, SH_BLOCK, 0, st>>>(n, K, dirs, coeffs, masks, colors, act); break;
        case 1: sh_fwd_kernel<1><<<grid, SH_BLOCK, 0, st>>>(n, K, dirs, coeffs, masks, colors, act); break;
        case 2: sh_fwd_kernel<2><<<grid, SH_BLOCK, 0, st>>>(n, K, dirs, coeffs, masks, colors, act); break;
        case 3: sh_fwd_kernel<3><<<grid, SH_BLOCK, 0, st>>>(n, K, dirs, coeffs, masks, colors, act); break;
        default: sh_fwd_kernel<4><<<grid, SH_BLOCK, 0, st>>>(n, K, dirs, coeffs, masks, colors, act); break;
    }
    MTGS_CHECK_LAUNCH("mtgs_sh_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_sh_bwd(int64_t n, int K, int degree, const float *dirs, const float *coeffs,
                           const uint8_t *masks, const float *v_colors, float *v_coeffs,
                           float *v_dirs, void *stream) {
    return mtgs_sh_bwd_act(n, K, degree, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, nullptr, stream);
}

extern "C" int mtgs_sh_bwd_act(int64_t n, int K, int degree, const float *dirs, const float *coeffs, const uint8_t *masks,
                               const float *v_colors, float *v_coeffs, float *v_dirs, const uint8_t *pass, void *stream) {
    MTGS_REQUIRE(n >= 0 && K > 0, MTGS_EINVAL, "mtgs_sh_bwd: bad sizes n=%lld K=%d", (long long)n, K);
    MTGS_REQUIRE(degree >= 0 && degree <= MTGS_MAX_SH_DEGREE && (degree + 1) * (degree + 1) <= K,
                 MTGS_EINVAL, "mtgs_sh_bwd: degree %d needs (degree+1)^2 <= K=%d and degree <= 4", degree, K);
    if (n == 0) return MTGS_OK;
    MTGS_REQUIRE(dirs && v_colors && v_coeffs && (coeffs || !v_dirs), MTGS_EINVAL, "mtgs_sh_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div64(n, SH_BLOCK);
#define MTGS_SH_BWD(DG) \
    sh_bwd_kernel<DG><<<grid, SH_BLOCK, 0, st>>>(n, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, pass)
    switch (degree) {
        case 0: MTGS_SH_BWD(0); break;
        case 1: MTGS_SH_BWD(1); break;
        case 2: MTGS_SH_BWD(2); break;
        case 3: MTGS_SH_BWD(3); break;
        default: MTGS_SH_BWD(4); break;
    }
#undef MTGS_SH_BWD
    MTGS_CHECK_LAUNCH("mtgs_sh_bwd");
    return MTGS_OK;
}

extern "C" int mtgs_sh_bwd_rows(int64_t n, int K, int degree, const float *dirs, const uint8_t *masks, const float *v_colors,
                                float *v_coeffs, void *stream) {
    return mtgs_sh_bwd_rows_act(n, K, degree, dirs, masks, v_colors, v_coeffs, nullptr, stream);
}

extern "C" int mtgs_sh_bwd_rows_act(int64_t n, int K, int degree, const float *dirs, const uint8_t *masks, const float *v_colors,
                                    float *v_coeffs, const uint8_t *pass, void *stream) {
    MTGS_REQUIRE(n >= 0 && K > 0, MTGS_EINVAL, "mtgs_sh_bwd_rows: bad sizes n=%lld K=%d", (long long)n, K);
    MTGS_REQUIRE(degree >= 0 && degree <= MTGS_MAX_SH_DEGREE && (degree + 1) * (degree + 1) <= K, MTGS_EINVAL,
                 "mtgs_sh_bwd_rows: degree %d needs (degree+1)^2 <= K=%d and degree <= 4", degree, K);
    if (n == 0) return MTGS_OK;
    MTGS_REQUIRE(dirs && v_colors && v_coeffs, MTGS_EINVAL, "mtgs_sh_bwd_rows: null pointer");
    MTGS_REQUIRE((K * 3) % 4 != 0 || (reinterpret_cast<uintptr_t>(v_coeffs) & 15) == 0, MTGS_EINVAL, "mtgs_sh_bwd_rows: v_coeffs must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div64(n, 256);
#define MTGS_SH_ROWS(DG) sh_bwd_rows_kernel<DG><<<grid, 256, 0, st>>>(n, K, dirs, masks, v_colors, v_coeffs, pass)
    switch (degree) {
        case 0: MTGS_SH_ROWS(0); break;
        case 1: MTGS_SH_ROWS(1); break;
        case 2: MTGS_SH_ROWS(2); break;
        case 3: MTGS_SH_ROWS(3); break;
        default: MTGS_SH_ROWS(4); break;
    }
#undef MTGS_SH_ROWS
    MTGS_CHECK_LAUNCH("mtgs_sh_bwd_rows");
    return MTGS_OK;
}
