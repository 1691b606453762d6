// dev/blend_dev.hpp -- development hooks of blend.hip, compiled into VARIANT builds only
// (scripts/build_variant.py NAME -DMTGS_DEV [-DMTGS_COUNT]); never part of libmtgs_rast.so.
//   -DMTGS_DEV     MTGS_PPL=<1|2|4> in the environment overrides pick_ppl() (scripts/kbench.py sweeps)
//   -DMTGS_COUNT   candidate / entry / slot / lane counters of the compositing backward, read with mtgs_blend_counters()
//                  (scripts/dev/blend_counts.py; profiles/r03_blend_isa_budget.md)
//   -DMTGS_TIMELINE  every workgroup of the packed compositing kernels records {realtime, shader clock} at its first and last
//                  instruction, where it ran (HW_ID / XCC_ID) and how many staged entries / entries with a valid pixel it saw;
//                  read with mtgs_blend_timeline() (scripts/dev/blend_timeline.py; profiles/r06_valu_ceiling.md)
#pragma once
#include <stdlib.h>

#define MTGS_DEV_PPL_OVERRIDE() do { if (const char *e_ = getenv("MTGS_PPL")) return atoi(e_); } while (0)

#ifdef MTGS_COUNT
__device__ unsigned long long g_blend_counters[8];
#define MTGS_COUNT_ADD(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_blend_counters[i], (unsigned long long)(v)); } while (0)
#define MTGS_COUNT_SLOTS(vmask) do { int ns_ = 0, nl_ = 0; for (int p_ = 0; p_ < PPL; ++p_) { ns_ += (vmask)[p_] != 0; nl_ += __popcll((vmask)[p_]); } \
                                     MTGS_COUNT_ADD(2, ns_); MTGS_COUNT_ADD(3, nl_); } while (0)
extern "C" int mtgs_blend_counters(unsigned long long *out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blend_counters), sizeof(g_blend_counters));
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_blend_counters), z, sizeof(z)); }
    return 0;
}
#else
#define MTGS_COUNT_ADD(i, v) do { } while (0)
#define MTGS_COUNT_SLOTS(vmask) do { } while (0)
#endif

#ifdef MTGS_TIMELINE
struct MtgsTlRec { unsigned long long rt0, rt1, c0, c1; unsigned hw, xcc, staged, active; };
__device__ MtgsTlRec g_blend_tl[2][32768];
#define MTGS_TL_DECL() unsigned long long tl_rt0_ = __builtin_amdgcn_s_memrealtime(), tl_c0_ = __builtin_readcyclecounter(); unsigned tl_staged_ = 0, tl_active_ = 0
#define MTGS_TL_STAGED(n) do { tl_staged_ += (unsigned)(n); } while (0)
#define MTGS_TL_ACTIVE() do { tl_active_ += 1u; } while (0)
#define MTGS_TL_END(kind) do { if (threadIdx.x == 0 && blockIdx.x < 32768) { MtgsTlRec r_; r_.rt0 = tl_rt0_; r_.c0 = tl_c0_; \
        r_.rt1 = __builtin_amdgcn_s_memrealtime(); r_.c1 = __builtin_readcyclecounter(); \
        r_.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4); r_.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); \
        r_.staged = tl_staged_; r_.active = tl_active_; g_blend_tl[kind][blockIdx.x] = r_; } } while (0)
extern "C" int mtgs_blend_timeline(void *out, int kind, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blend_tl), sizeof(MtgsTlRec) * (size_t)n, sizeof(MtgsTlRec) * 32768 * (size_t)kind);
}
#else
#define MTGS_TL_DECL() do { } while (0)
#define MTGS_TL_STAGED(n) do { } while (0)
#define MTGS_TL_ACTIVE() do { } while (0)
#define MTGS_TL_END(kind) do { } while (0)
#endif
