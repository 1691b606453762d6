// fourier.hip -- Fourier-series features_dc of rigid object nodes.
//
// Restates RigidSubModel.get_fourier_features (/root/reference/mtgs/scene_model/gaussian_model/rigid_node.py:217-221):
//     true_features_dc[n, :] = sum_f features_dc[n, f, :] * idft_base[f]
// with idft_base = IDFT(x * fourier_features_scale, F, ...) (utils.py:335-352; F weights shared by the whole node: x is the
// frame's normalised timestamp or the camera-object yaw).  PyTorch runs it as a broadcast multiply that materialises
// [N, F, 3] plus a reduction; here one pass reads the parameter once (forward) and the backward writes
// v_features_dc[n, f, :] = w[f] * v_dc[n, :] in one pass, plus v_w[f] = sum_n <features_dc[n, f, :], v_dc[n, :]> (the yaw
// depends on the object pose, so the weights can carry a gradient) through per-block partial sums in a fixed order.
// Roofline: HBM, N * F * 12 bytes per direction.
#include "common.hpp"

namespace {
constexpr int FOURIER_MAX_DIM = 32;

__global__ __launch_bounds__(256) void fourier_dc_fwd_kernel(int64_t N, int F, const float *__restrict__ features_dc,
                                                            const float *__restrict__ w, float *__restrict__ dc) {
    __shared__ float s_w[FOURIER_MAX_DIM];
    if (threadIdx.x < F) s_w[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (Gaussian, channel)
    if (e >= N * 3) return;
    const int64_t n = e / 3;
    const int c = (int)(e - n * 3);
    float acc = 0.f;
    for (int f = 0; f < F; ++f) acc += features_dc[(n * F + f) * 3 + c] * s_w[f];
    dc[e] = acc;
}

__global__ __launch_bounds__(256) void fourier_dc_bwd_kernel(int64_t N, int F, const float *__restrict__ features_dc,
                                                            const float *__restrict__ w, const float *__restrict__ v_dc,
                                                            float *__restrict__ v_features_dc, float *__restrict__ partial_w) {
    __shared__ float s_w[FOURIER_MAX_DIM];
    __shared__ float s_red[256 / 64][FOURIER_MAX_DIM];
    if (threadIdx.x < F) s_w[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = e < N * 3;
    const int64_t n = live ? e / 3 : 0;
    const int c = live ? (int)(e - n * 3) : 0;
    const float g = live ? v_dc[e] : 0.f;
    for (int f = 0; f < F; ++f) {
        float t = 0.f;
        if (live) {
            const int64_t k = (n * F + f) * 3 + c;
            v_features_dc[k] = s_w[f] * g;
            if (partial_w) t = features_dc[k] * g;
        }
        if (partial_w) {
            t = wave_sum_to_lane63(t);
            if ((threadIdx.x & 63) == 63) s_red[threadIdx.x >> 6][f] = t;
        }
    }
    if (partial_w) {
        __syncthreads();
        if (threadIdx.x < F) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 256 / 64; ++k) t += s_red[k][threadIdx.x];
            partial_w[(int64_t)blockIdx.x * F + threadIdx.x] = t;
        }
    }
}
}  // namespace

extern "C" int mtgs_fourier_dc_fwd(int64_t N, int F, const float *features_dc, const float *w, float *dc, void *stream) {
    MTGS_REQUIRE(N >= 0 && F >= 1 && F <= FOURIER_MAX_DIM, MTGS_EINVAL, "mtgs_fourier_dc_fwd: N=%lld F=%d (1..%d)", (long long)N, F,
                 FOURIER_MAX_DIM);
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(features_dc && w && dc, MTGS_EINVAL, "mtgs_fourier_dc_fwd: null pointer");
    fourier_dc_fwd_kernel<<<(unsigned)ceil_div64(N * 3, 256), 256, 0, (hipStream_t)stream>>>(N, F, features_dc, w, dc);
    MTGS_CHECK_LAUNCH("mtgs_fourier_dc_fwd");
    return MTGS_OK;
}

/* partial_w (nullable): [ceil(3 N / 256), F] per-block partial sums of the weight gradient; the caller adds them up */
extern "C" int mtgs_fourier_dc_bwd(int64_t N, int F, const float *features_dc, const float *w, const float *v_dc,
                                   float *v_features_dc, float *partial_w, void *stream) {
    MTGS_REQUIRE(N >= 0 && F >= 1 && F <= FOURIER_MAX_DIM, MTGS_EINVAL, "mtgs_fourier_dc_bwd: N=%lld F=%d (1..%d)", (long long)N, F,
                 FOURIER_MAX_DIM);
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(features_dc && w && v_dc && v_features_dc, MTGS_EINVAL, "mtgs_fourier_dc_bwd: null pointer");
    fourier_dc_bwd_kernel<<<(unsigned)ceil_div64(N * 3, 256), 256, 0, (hipStream_t)stream>>>(N, F, features_dc, w, v_dc, v_features_dc,
                                                                                         partial_w);
    MTGS_CHECK_LAUNCH("mtgs_fourier_dc_bwd");
    return MTGS_OK;
}
