// dp.hip -- receiver side of the sparse, factored gradient exchange of view-parallel data
// parallelism (mtgs_amd/dist.py::SparseGradExchange).  No gsplat counterpart: the reference's only
// collective is nerfstudio's dense DDP all-reduce (/root/reference/mtgs/scene_model/
// custom_pipeline.py:87-89).
//
// Per step every rank renders ONE camera and only the Gaussians visible in it (~15 %) get a non-zero
// gradient; moreover the gradient of the SH coefficients is rank-1 per Gaussian,
//     v_coeffs[n, k, :] = basis_k(normalize(mean_n - cam_pos)) * v_rgb[n, :],
// so 48 floats are fully described by 3 (v_rgb) plus the sender's camera position.  Ranks therefore
// all-gather compact 64-byte rows {v_mean 3, v_quat 4, v_scale 3, v_opacity 1, v_rgb 3, -, index}
// (19 MB instead of 472 MB per rank at 2M Gaussians, SH degree 3) and this kernel adds one sender's
// rows into the dense, replicated gradient tensors, expanding v_rgb through the SH basis on the fly.
// The result equals the dense all-reduce up to fp32 summation order (tests/test_gpu_dp.py).
//
// Roofline: HBM, read-modify-write of (44 + 12K) bytes per row.
#include "common.hpp"

namespace {

__device__ __forceinline__ float sh_basis_k(int k, float x, float y, float z) {
    // gsplat spherical_harmonics (sh_coeffs_to_color_fast) basis functions, see sh.hip
    const float z2 = z * z;
    const float fC1 = x * x - y * y, fS1 = 2.f * x * y;
    switch (k) {
        case 0: return 0.2820947917738781f;
        case 1: return -0.48860251190292f * y;
        case 2: return 0.48860251190292f * z;
        case 3: return -0.48860251190292f * x;
        case 4: return 0.5462742152960395f * fS1;
        case 5: return -1.092548430592079f * z * y;
        case 6: return 0.9461746957575601f * z2 - 0.3153915652525201f;
        case 7: return -1.092548430592079f * z * x;
        case 8: return 0.5462742152960395f * fC1;
        default: break;
    }
    const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    const float fTmp1B = 1.445305721320277f * z;
    const float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    const float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    switch (k) {
        case 9: return -0.5900435899266435f * fS2;
        case 10: return fTmp1B * fS1;
        case 11: return fTmp0C * y;
        case 12: return pSH12;
        case 13: return fTmp0C * x;
        case 14: return fTmp1B * fC1;
        case 15: return -0.5900435899266435f * fC2;
        default: break;
    }
    const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    const float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    const float fTmp2B = -1.770130769779931f * z;
    const float fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
    const float pSH6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
    switch (k) {
        case 16: return 0.6258357354491763f * fS3;
        case 17: return fTmp2B * fS2;
        case 18: return fTmp1C * fS1;
        case 19: return fTmp0D * y;
        case 20: return 1.984313483298443f * z * pSH12 - 1.006230589874905f * pSH6;
        case 21: return fTmp0D * x;
        case 22: return fTmp1C * fC1;
        case 23: return fTmp2B * fC2;
        default: return 0.6258357354491763f * fC3;
    }
}

// TPR threads per row; thread j of a row owns gradient component j (geometry) and SH basis j.
template <int TPR>
__global__ __launch_bounds__(256) void dp_accumulate_kernel(
    int64_t n_rows, const float *__restrict__ rows, int K, int nb, const float *__restrict__ means,
    const float *__restrict__ cam_pos, float *__restrict__ v_means, float *__restrict__ v_quats,
    float *__restrict__ v_scales, float *__restrict__ v_opacities, float *__restrict__ v_coeffs) {
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / TPR;
    const int j = threadIdx.x % TPR;
    if (r >= n_rows) return;
    const float *row = rows + r * 16;
    const int64_t idx = (int64_t)__float_as_int(row[15]);
    if (j < 3) v_means[idx * 3 + j] += row[j];
    else if (j < 7) v_quats[idx * 4 + (j - 3)] += row[j];
    else if (j < 10) v_scales[idx * 3 + (j - 7)] += row[j];
    else if (j == 10) v_opacities[idx] += row[10];
    if (v_coeffs && j < nb) {
        float x = means[idx * 3] - cam_pos[0], y = means[idx * 3 + 1] - cam_pos[1], z = means[idx * 3 + 2] - cam_pos[2];
        const float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
        x *= inorm; y *= inorm; z *= inorm;
        const float b = sh_basis_k(j, x, y, z);
        float *dst = v_coeffs + (idx * K + j) * 3;
        dst[0] += b * row[11]; dst[1] += b * row[12]; dst[2] += b * row[13];
    }
}

// Sender side: compact the rows of the visible Gaussians (unordered).  A block owns a chunk of
// PACK_CHUNK Gaussians: it counts its visible ones, reserves a contiguous range of rows with ONE atomic
// (same-address atomics serialise at ~12 ns each: one per wave cost 380 us at 2M Gaussians) and then
// writes the rows, positions inside the range coming from ballots.
constexpr int PACK_CHUNK = 2048;
__global__ __launch_bounds__(256) void dp_pack_kernel(
    int64_t N, const int32_t *__restrict__ radii, const float *__restrict__ v_means,
    const float *__restrict__ v_quats, const float *__restrict__ v_scales,
    const float *__restrict__ v_opacities, const float *__restrict__ v_rgb, float *__restrict__ rows,
    int64_t capacity, int64_t *__restrict__ count) {
    __shared__ int s_w[4];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * PACK_CHUNK;
    // pass 1: number of visible Gaussians in the chunk
    int mine = 0;
    for (int i = tid; i < PACK_CHUNK; i += 256) mine += (c0 + i < N && radii[c0 + i] > 0) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane == 0) s_w[wave] = mine;
    __syncthreads();
    if (tid == 0) {
        const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        s_base = tot ? (long long)atomicAdd((unsigned long long *)count, (unsigned long long)tot) : 0;
    }
    __syncthreads();
    int64_t run = s_base;
    // pass 2: write the rows; `run` advances identically in every thread
    for (int i0 = 0; i0 < PACK_CHUNK; i0 += 256) {
        const int64_t n = c0 + i0 + tid;
        const bool vis = n < N && radii[n] > 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(vis);
        __syncthreads();
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int below = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) below += s_w[w];
            tot += s_w[w];
        }
        if (vis) {
            const int64_t slot = run + below + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (slot < capacity) {  // the caller's buffer holds N rows; never exceeded
                float4 *dst = reinterpret_cast<float4 *>(rows + slot * 16);
                dst[0] = make_float4(v_means[n * 3], v_means[n * 3 + 1], v_means[n * 3 + 2], v_quats[n * 4]);
                dst[1] = make_float4(v_quats[n * 4 + 1], v_quats[n * 4 + 2], v_quats[n * 4 + 3], v_scales[n * 3]);
                dst[2] = make_float4(v_scales[n * 3 + 1], v_scales[n * 3 + 2], v_opacities[n], v_rgb ? v_rgb[n * 3] : 0.f);
                dst[3] = make_float4(v_rgb ? v_rgb[n * 3 + 1] : 0.f, v_rgb ? v_rgb[n * 3 + 2] : 0.f, 0.f, __int_as_float((int)n));
            }
        }
        run += tot;
    }
}

}  // namespace

extern "C" int mtgs_dp_pack(int64_t N, const int32_t *radii, const float *v_means, const float *v_quats,
                            const float *v_scales, const float *v_opacities, const float *v_rgb, float *rows,
                            int64_t capacity, int64_t *count, void *stream) {
    MTGS_REQUIRE(N >= 0 && capacity >= 0, MTGS_EINVAL, "mtgs_dp_pack: bad sizes");
    MTGS_REQUIRE(N < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_dp_pack: N must fit int32");
    MTGS_REQUIRE(count, MTGS_EINVAL, "mtgs_dp_pack: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int64_t), st);
    MTGS_REQUIRE(e == hipSuccess, MTGS_ELAUNCH, "mtgs_dp_pack: memset failed");
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(radii && v_means && v_quats && v_scales && v_opacities && rows, MTGS_EINVAL, "mtgs_dp_pack: null pointer");
    dp_pack_kernel<<<(unsigned)ceil_div64(N, PACK_CHUNK), 256, 0, st>>>(N, radii, v_means, v_quats, v_scales, v_opacities, v_rgb,
                                                                 rows, capacity, count);
    MTGS_CHECK_LAUNCH("mtgs_dp_pack");
    return MTGS_OK;
}

extern "C" int mtgs_dp_accumulate(int64_t n_rows, const float *rows, int64_t N, int K, int degree,
                                  const float *means, const float *cam_pos, float *v_means, float *v_quats,
                                  float *v_scales, float *v_opacities, float *v_coeffs, void *stream) {
    MTGS_REQUIRE(n_rows >= 0 && N >= 0, MTGS_EINVAL, "mtgs_dp_accumulate: bad sizes");
    if (n_rows == 0) return MTGS_OK;
    MTGS_REQUIRE(rows && v_means && v_quats && v_scales && v_opacities, MTGS_EINVAL, "mtgs_dp_accumulate: null pointer");
    int nb = 0;
    if (v_coeffs) {
        MTGS_REQUIRE(means && cam_pos && degree >= 0 && degree <= MTGS_MAX_SH_DEGREE && (degree + 1) * (degree + 1) <= K,
                     MTGS_EINVAL, "mtgs_dp_accumulate: bad SH arguments (degree %d, K %d)", degree, K);
        nb = (degree + 1) * (degree + 1);
    }
    hipStream_t st = (hipStream_t)stream;
    if (nb <= 16)
        dp_accumulate_kernel<16><<<(unsigned)ceil_div64(n_rows * 16, 256), 256, 0, st>>>(
            n_rows, rows, K, nb, means, cam_pos, v_means, v_quats, v_scales, v_opacities, v_coeffs);
    else
        dp_accumulate_kernel<32><<<(unsigned)ceil_div64(n_rows * 32, 256), 256, 0, st>>>(
            n_rows, rows, K, nb, means, cam_pos, v_means, v_quats, v_scales, v_opacities, v_coeffs);
    MTGS_CHECK_LAUNCH("mtgs_dp_accumulate");
    return MTGS_OK;
}
