// dev/blend_dev.hpp -- development hooks of blend.hip, compiled into VARIANT builds only
// (scripts/build_variant.py NAME -DMTGS_DEV [-DMTGS_COUNT]); never part of libmtgs_rast.so.
//   -DMTGS_DEV     MTGS_PPL=<1|2|4> in the environment overrides pick_ppl() (scripts/kbench.py sweeps)
//   -DMTGS_COUNT   candidate / entry / slot / lane counters of the compositing backward, read with mtgs_blend_counters()
//                  (scripts/dev/blend_counts.py; profiles/r03_blend_isa_budget.md)
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
