// Development microbenchmark (round 6): the VALU ISSUE CEILING of gfx950 for the instruction kinds of the compositing
// kernels, in cycles per wave64 instruction per SIMD, as a function of the waves resident per SIMD -- measured THREE ways:
//   (1) wall time (HIP events) x the nominal 2.4 GHz,
//   (2) s_memtime ticks between the first and the last instruction of every wave (shader cycles at the clock the chip
//       really ran at; mean and max over the waves),
//   (3) under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE` (program directly behind `--`): every mode is
//       its own kernel (template argument in the name), every occupancy its own Grid_Size.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I mtgs_amd/csrc -o valu_ceiling scripts/dev/valu_ceiling.hip
// One-wave workgroups (as blend_{fwd,bwd}_kernel<., 4, .>): grid = 1024 x waves-per-SIMD, all resident at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "wave_reduce.hpp"

enum Mode {
    FMA_INDEP = 0, FMA_CHAIN, MUL_INDEP, CNDMASK_VCC, EXP_INDEP, RCP_INDEP, CMP_SAND_CNDMASK, REDUCE12, LDS_BCAST,
    SLOT_BLOCK, SLOT_BRANCHY, SLOT_BRANCHY_LDS, FMA_CHAIN2, FMA_CHAIN4, N_MODES
};
static const char *kNames[N_MODES] = {
    "fma_indep(8 chains)", "fma_chain(1 dependent)", "mul_indep", "cndmask_vcc", "exp_indep", "rcp_indep",
    "cmp>sgpr>s_and>cndmask", "reduce12(permlane_swap+dpp)", "lds_bcast_b128", "bwd_slot x4, one block", "bwd_slot x4, branch per slot",
    "bwd entry: lds+test+4 slots+reduce", "fma 2 chains", "fma 4 chains"};
// VALU instructions per loop iteration (hand-counted for the asm modes; the C++ modes are counted from the ISA, see --count and
// the PMC pass, and this table is only used for the time-based columns)
static int kValuPerIter[N_MODES] = {32, 32, 32, 32, 32, 32, 32, 25, 0, 4 * 27, 4 * 27, 0, 32, 32};

struct SlotState {
    float T[4], Bq[4], S0, S1, S2, g2, g3, gc[4];
};

// the backward's pixel slot (blend.hip, PK, no clamp, D = 4): 27 VALU
__device__ __forceinline__ void bwd_slot(float s2, float dy, float opac, const float (&col)[4], const float (&vr)[4], float r0w, float r1x,
                                         float adx, float bdx, unsigned long long vmask, float &T, float &Bq, SlotState &st) {
    const bool valid = __builtin_amdgcn_inverse_ballot_w64(vmask);
    const float vis = valid ? __builtin_amdgcn_exp2f(-s2) : 0.f;
    float alpha;
    {
#pragma clang fp contract(off)
        alpha = opac * vis;
    }
    const float ra = __builtin_amdgcn_rcpf(1.0f - alpha);
    T *= ra;
    const float fac = alpha * T;
    float A = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        st.gc[k] += fac * vr[k];
        A += col[k] * vr[k];
    }
    const float v_alpha = A * T - ra * Bq;
    Bq += fac * A;
    const float v_sigma = vis * v_alpha;
    const float vsdy = v_sigma * dy;
    st.S0 += v_sigma;
    st.S1 += vsdy;
    st.S2 += vsdy * dy;
    st.g2 = fmaf(fabsf(v_sigma), fabsf(fmaf(r0w, dy, adx)), st.g2);
    st.g3 = fmaf(fabsf(v_sigma), fabsf(fmaf(r1x, dy, bdx)), st.g3);
}

template <int MODE>
__global__ __launch_bounds__(64) void vc(int iters, float *out, unsigned long long *cyc, float seed) {
    __shared__ __attribute__((aligned(16))) float s_rec[128 * 12];
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i + seed;
    const float b = 1.0001f, c = 0.0001f;
    for (int i = threadIdx.x; i < 128 * 12; i += 64) s_rec[i] = 0.5f + 0.001f * i + seed;
    __syncthreads();
    SlotState st;
    float py[4], vr[4][4];
    for (int p = 0; p < 4; ++p) {
        st.T[p] = 0.9f; st.Bq[p] = 0.1f + seed; py[p] = (threadIdx.x >> 4) + 4 * p + 0.5f;
        for (int k = 0; k < 4; ++k) vr[p][k] = 0.01f * (k + 1) + seed;
    }
    st.S0 = st.S1 = st.S2 = st.g2 = st.g3 = 0.f;
    for (int k = 0; k < 4; ++k) st.gc[k] = 0.f;
    const float px = (threadIdx.x & 15) + 0.5f;
    // wave-uniform masks the compiler cannot see through (all lanes on)
    unsigned long long m_all = __ballot(a[0] > -1e30f);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == FMA_INDEP) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (MODE == FMA_CHAIN) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        } else if (MODE == FMA_CHAIN2) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (MODE == FMA_CHAIN4) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (MODE == MUL_INDEP) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (MODE == CNDMASK_VCC) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
        } else if (MODE == EXP_INDEP) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == RCP_INDEP) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == CMP_SAND_CNDMASK) {
            // 16 x { v_cmp -> SGPR pair, s_and with a mask, v_cndmask on that SGPR pair }: 32 VALU + 16 SALU, dependent VALU->SALU->VALU
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    unsigned long long m;
                    asm volatile("v_cmp_le_u32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b));
                    asm volatile("s_and_b64 %0, %0, %1" : "+s"(m) : "s"(m_all));
                    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(m));
                }
        } else if (MODE == REDUCE12) {
            float gv[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) gv[k] = a[k & 7] + (float)k;
            const float v = wave_reduce_x4_packed<3>(gv);
            a[0] += v;   // (12 adds to build the inputs + the reduction: counted from the ISA)
        } else if (MODE == LDS_BCAST) {
            // 8 wave-uniform 16-byte reads per iteration, consumed by one add each
            uint32_t va = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)s_rec + (uint32_t)(it & 63) * 48u;
            asm volatile("v_mov_b32 %0, %0" : "+v"(va));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 q;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(va), "n"(16 * i));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q));
                a[i] += q.x;
            }
        } else if (MODE == SLOT_BLOCK || MODE == SLOT_BRANCHY) {
            float s2[4], dy[4], col[4], opac = a[4], r0w = a[5], r1x = a[6], adx = a[7], bdx = a[0];
#pragma unroll
            for (int p = 0; p < 4; ++p) { s2[p] = a[p]; dy[p] = py[p]; asm volatile("" : "+v"(s2[p]), "+v"(dy[p])); }
#pragma unroll
            for (int k = 0; k < 4; ++k) { col[k] = a[k + 4]; asm volatile("" : "+v"(col[k])); }
            asm volatile("" : "+v"(opac), "+v"(r0w), "+v"(r1x), "+v"(adx), "+v"(bdx));
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned long long vm = m_all;
                asm volatile("" : "+s"(vm));
                if (MODE == SLOT_BLOCK || vm != 0) bwd_slot(s2[p], dy[p], opac, col, vr[p], r0w, r1x, adx, bdx, vm, st.T[p], st.Bq[p], st);
            }
        } else if (MODE == SLOT_BRANCHY_LDS) {
            // one ENTRY of the backward as blend.hip runs it: two broadcast record reads + the colour row, the validity test of the
            // lane's four pixels, the any-lane branch, four slots behind wave-uniform branches, the raw-moment epilogue and the
            // 12-value reduction (no atomic)
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const uint32_t va = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)s_rec + (uint32_t)(it & 127) * 48u;
            const f32x4 r0 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)va);
            const f32x4 r1 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(va + 16));
            const f32x4 cq = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)(va + 32));
            const float col[4] = {cq.x, cq.y, cq.z, cq.w};
            const float dx = r0.x - px;
            const float adx = r0.z * dx, bdx = r0.w * dx;
            const float q0 = adx * dx, b2dx = bdx + bdx;
            float dy[4], s2[4];
            unsigned long long vmask[4], any = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                dy[p] = r0.y - py[p];
                s2[p] = fmaf(dy[p], fmaf(r1.x, dy[p], b2dx), q0);
                vmask[p] = __builtin_amdgcn_uicmp(__float_as_uint(s2[p]), __float_as_uint(r1.z + 1e30f), 37) | m_all;
                any |= vmask[p];
            }
            if (any != 0) {
                st.S0 = st.S1 = st.S2 = st.g2 = st.g3 = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) st.gc[k] = 0.f;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (vmask[p] != 0) bwd_slot(s2[p] * 1e-3f, dy[p], r1.y * 1e-3f, col, vr[p], r0.w, r1.x, adx, bdx, vmask[p], st.T[p], st.Bq[p], st);
                float gv[12];
                gv[0] = dx * st.S0; gv[1] = st.S1; gv[2] = st.g2; gv[3] = st.g3; gv[4] = dx * gv[0]; gv[5] = dx * st.S1; gv[6] = st.S2; gv[7] = st.S0;
#pragma unroll
                for (int k = 0; k < 4; ++k) gv[8 + k] = st.gc[k];
                a[0] += wave_reduce_x4_packed<3>(gv);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int p = 0; p < 4; ++p) s += st.T[p] + st.Bq[p];
    s += st.S0 + st.S1 + st.S2 + st.g2 + st.g3 + st.gc[0] + st.gc[1] + st.gc[2] + st.gc[3];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(float *buf, unsigned long long *cyc, std::vector<unsigned long long> &h, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const int waves = 1024 * wps;
        vc<MODE><<<waves, 64>>>(iters / 4, buf, cyc, 0.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        vc<MODE><<<waves, 64>>>(iters, buf, cyc, 0.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0, mx = 0;
        for (int i = 0; i < waves; ++i) { mean += (double)h[i]; mx = std::max(mx, (double)h[i]); }
        mean /= waves;
        const double ipw = (double)iters * kValuPerIter[MODE];   // VALU per wave (0: counted from the PMC pass)
        const double per_simd = ipw * wps;
        printf("%-38s mode %2d waves/SIMD %d grid %5d iters %6d: %8.1f us | memtime/wave mean %10.0f max %10.0f", kNames[MODE], MODE, wps, waves, iters,
               ms * 1e3, mean, mx);
        if (ipw > 0)
            printf(" | cyc/VALU/SIMD: wall@2.4GHz %.2f  memtime(max) %.2f  | per wave: %.2f cyc between its VALU", ms * 1e-3 * 2.4e9 / per_simd,
                   mx / per_simd, mean / ipw);
        printf("\n");
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float *buf;
    unsigned long long *cyc;
    hipMalloc(&buf, 1024);
    hipMalloc(&cyc, 8192 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(8192);
    run<FMA_INDEP>(buf, cyc, h, iters);
    run<FMA_CHAIN>(buf, cyc, h, iters);
    run<FMA_CHAIN2>(buf, cyc, h, iters);
    run<FMA_CHAIN4>(buf, cyc, h, iters);
    run<MUL_INDEP>(buf, cyc, h, iters);
    run<CNDMASK_VCC>(buf, cyc, h, iters);
    run<EXP_INDEP>(buf, cyc, h, iters);
    run<RCP_INDEP>(buf, cyc, h, iters);
    run<CMP_SAND_CNDMASK>(buf, cyc, h, iters);
    run<REDUCE12>(buf, cyc, h, iters);
    run<LDS_BCAST>(buf, cyc, h, iters);
    run<SLOT_BLOCK>(buf, cyc, h, iters / 2);
    run<SLOT_BRANCHY>(buf, cyc, h, iters / 2);
    run<SLOT_BRANCHY_LDS>(buf, cyc, h, iters / 2);
    return 0;
}
