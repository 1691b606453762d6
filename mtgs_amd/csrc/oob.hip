// oob.hip -- the out-of-box regulariser of the rigid object nodes, every node of the frame in one pass (forward:
// two launches + a finish, backward: one).  On in the shipped config (config/MTGS.py:117 oob_lambda = 1.0).
//
// Restates /root/reference/mtgs/scene_model/mtgs_scene_graph.py:949-967, which loops over the rigid models in Python
// with a full-size `model_id == id` comparison, two boolean-mask gathers and two host synchronisations PER NODE:
//     for every rigid node present in the frame:
//         if no Gaussian of the node is visible (radii > 0): skip the node
//         oob   = any(|means_local| > instance_size / 2 + tolerance, dim=-1)            (detached)
//         loss += sum(-log(1 - sigmoid(opacities[oob]) + 1e-6));  count += sum(oob)
//     loss = loss / count   (only when count != 0)
// Table of descriptors in device memory as for the node activations (include/mtgs_rast.h: mtgs_oob_desc).
#include "common.hpp"

namespace {
constexpr int OOB_BLOCK = 256;

__device__ __forceinline__ const mtgs_oob_desc &desc_of_block(const mtgs_oob_desc *__restrict__ table, int n_nodes, int &node) {
    int lo = 0, hi = n_nodes - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    node = __builtin_amdgcn_readfirstlane(lo);
    return table[node];
}

// flags[node] = 1 when any Gaussian of the node is visible (flags zeroed by the caller)
__global__ __launch_bounds__(OOB_BLOCK) void oob_visible_kernel(const mtgs_oob_desc *__restrict__ table, int n_nodes,
                                                                const int32_t *__restrict__ radii, int32_t *__restrict__ flags) {
    int node;
    const mtgs_oob_desc &d = desc_of_block(table, n_nodes, node);
    const int64_t i = ((int64_t)blockIdx.x - d.first_block) * OOB_BLOCK + threadIdx.x;
    const bool vis = i < d.n && radii[d.start + i] > 0;
    if (__builtin_amdgcn_ballot_w64(vis) != 0 && (threadIdx.x & 63) == 0) flags[node] = 1;   // same value from every writer
}

__device__ __forceinline__ bool is_oob(const mtgs_oob_desc &d, int64_t i) {
    const float x = d.means[i * 3], y = d.means[i * 3 + 1], z = d.means[i * 3 + 2];
    return fabsf(x) > d.limit[0] || fabsf(y) > d.limit[1] || fabsf(z) > d.limit[2];
}

__device__ __forceinline__ float block_sum(float v, float *lds) {
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

__global__ __launch_bounds__(OOB_BLOCK) void oob_fwd_kernel(const mtgs_oob_desc *__restrict__ table, int n_nodes,
                                                            const int32_t *__restrict__ flags, float *__restrict__ partials) {
    __shared__ float s_red[4];
    int node;
    const mtgs_oob_desc &d = desc_of_block(table, n_nodes, node);
    const int64_t i = ((int64_t)blockIdx.x - d.first_block) * OOB_BLOCK + threadIdx.x;
    float term = 0.f, cnt = 0.f;
    if (i < d.n && flags[node] && is_oob(d, i)) {
        const float s = 1.f / (1.f + expf(-d.opacities[i]));
        term = -logf((1.f - s) + 1e-6f);
        cnt = 1.f;
    }
    const float bs = block_sum(term, s_red), bc = block_sum(cnt, s_red);
    if (threadIdx.x == 0) { partials[(int64_t)blockIdx.x * 2] = bs; partials[(int64_t)blockIdx.x * 2 + 1] = bc; }
}

// out[0] = sum / count (0 when nothing is out of its box), out[1] = count
__global__ __launch_bounds__(OOB_BLOCK) void oob_finish_kernel(int64_t nblocks, const float *__restrict__ partials,
                                                               float *__restrict__ out) {
    __shared__ float s_red[4];
    float s = 0.f, c = 0.f;
    for (int64_t b = threadIdx.x; b < nblocks; b += OOB_BLOCK) { s += partials[b * 2]; c += partials[b * 2 + 1]; }
    const float ts = block_sum(s, s_red), tc = block_sum(c, s_red);
    if (threadIdx.x == 0) { out[0] = tc > 0.f ? ts / tc : 0.f; out[1] = tc; }
}

// d/do [-log(1 - sigmoid(o) + 1e-6)] = s (1 - s) / (1 - s + 1e-6)
__global__ __launch_bounds__(OOB_BLOCK) void oob_bwd_kernel(const mtgs_oob_desc *__restrict__ table, int n_nodes,
                                                            const int32_t *__restrict__ flags, const float *__restrict__ v_out,
                                                            const float *__restrict__ fwd_out) {
    int node;
    const mtgs_oob_desc &d = desc_of_block(table, n_nodes, node);
    const int64_t i = ((int64_t)blockIdx.x - d.first_block) * OOB_BLOCK + threadIdx.x;
    if (i >= d.n) return;
    float g = 0.f;
    if (flags[node] && fwd_out[1] > 0.f && is_oob(d, i)) {
        const float s = 1.f / (1.f + expf(-d.opacities[i]));
        g = v_out[0] / fwd_out[1] * (s * (1.f - s) / ((1.f - s) + 1e-6f));
    }
    d.g_opacities[i] = g;
}
}  // namespace

extern "C" int mtgs_oob_desc_bytes(void) { return (int)sizeof(mtgs_oob_desc); }

extern "C" int mtgs_oob_fwd(int n_nodes, const mtgs_oob_desc *table, int64_t total_blocks, const int32_t *radii, int32_t *flags,
                            float *partials, float *out, void *stream) {
    MTGS_REQUIRE(n_nodes >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_oob_fwd: bad sizes");
    MTGS_REQUIRE(out && (n_nodes == 0 || (table && flags)) && (total_blocks == 0 || (radii && partials)), MTGS_EINVAL,
                 "mtgs_oob_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n_nodes > 0) {
        hipError_t e = mtgs_zero_async(flags, (size_t)n_nodes * sizeof(int32_t), st) == MTGS_OK ? hipSuccess : hipErrorLaunchFailure;
        MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_oob_fwd: memset failed: %s", hipGetErrorString(e));
    }
    if (total_blocks > 0) {
        oob_visible_kernel<<<(unsigned)total_blocks, OOB_BLOCK, 0, st>>>(table, n_nodes, radii, flags);
        oob_fwd_kernel<<<(unsigned)total_blocks, OOB_BLOCK, 0, st>>>(table, n_nodes, flags, partials);
    }
    oob_finish_kernel<<<1, OOB_BLOCK, 0, st>>>(total_blocks, partials, out);
    MTGS_CHECK_LAUNCH("mtgs_oob_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_oob_bwd(int n_nodes, const mtgs_oob_desc *table, int64_t total_blocks, const int32_t *flags,
                            const float *v_out, const float *fwd_out, void *stream) {
    MTGS_REQUIRE(n_nodes >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_oob_bwd: bad sizes");
    if (n_nodes == 0 || total_blocks == 0) return MTGS_OK;
    MTGS_REQUIRE(table && flags && v_out && fwd_out, MTGS_EINVAL, "mtgs_oob_bwd: null pointer");
    oob_bwd_kernel<<<(unsigned)total_blocks, OOB_BLOCK, 0, (hipStream_t)stream>>>(table, n_nodes, flags, v_out, fwd_out);
    MTGS_CHECK_LAUNCH("mtgs_oob_bwd");
    return MTGS_OK;
}
