// deform.hip -- input embedding of the deformation network of MTGS's deformable (non-rigid) object nodes.
//
// DeformableSubModel.get_deformation (/root/reference/mtgs/scene_model/gaussian_model/deformable_node.py:173-203)
// feeds ConditionalDeformNetwork (utils.py:286-333) with, per Gaussian n:
//     x = means[n] / height * 2,      t = the frame's timestamp,      condition = the instance embedding
//     row = [ x, sin(x f_0), cos(x f_0), ... sin(x f_9), cos(x f_9) | t, sin(t f_0), cos(t f_0), ... | condition ]
// with f_i = 2^i (get_embedder: include_input, log sampling, utils.py:235-283).  PyTorch builds the row from 42
// element-wise launches, two `repeat`s and three `cat`s per frame and node; here ONE launch writes the [N, ld] matrix
// the first linear layer (a library GEMM) consumes -- directly into the left columns of the buffer the skip layer
// reads again, so that the skip connection's `cat` is a view.  Nothing but the condition carries a gradient
// (means.data, a timestamp), and that gradient is a column sum of the GEMM's input gradient.
// Roofline: HBM (writes): N * width * 4 bytes.
#include "common.hpp"

namespace {
constexpr int DEFORM_MAX_COND = 64, DEFORM_MAX_FREQS = 16;

__global__ __launch_bounds__(256) void deform_embed_kernel(int64_t N, const float *__restrict__ means, float height, float t,
                                                          const float *__restrict__ cond, int E, int xf, int tf,
                                                          float *__restrict__ out, int64_t ld) {
    __shared__ float s_tail[1 + 2 * DEFORM_MAX_FREQS + DEFORM_MAX_COND];   // the part of the row every Gaussian shares
    const int tw = 1 + 2 * tf, xw = 3 + 6 * xf;
    for (int j = threadIdx.x; j < tw + E; j += 256) {
        float v;
        if (j == 0) v = t;
        else if (j < tw) {
            const int q = j - 1, i = q >> 1;
            const float a = t * exp2f((float)i);
            v = (q & 1) ? cosf(a) : sinf(a);
        } else v = cond[j - tw];
        s_tail[j] = v;
    }
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float *row = out + n * ld;
    float x[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { x[c] = means[n * 3 + c] / height * 2.f; row[c] = x[c]; }
    for (int i = 0; i < xf; ++i) {
        const float f = exp2f((float)i);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = x[c] * f;
            row[3 + 6 * i + c] = sinf(a);
            row[3 + 6 * i + 3 + c] = cosf(a);
        }
    }
    for (int j = 0; j < tw + E; ++j) row[xw + j] = s_tail[j];
}
}  // namespace

/* out[N, ld] (ld >= 3 + 6 x_freqs + 1 + 2 t_freqs + E floats; only those columns are written). */
extern "C" int mtgs_deform_embed(int64_t N, const float *means, float height, float t, const float *cond, int E, int x_freqs,
                                 int t_freqs, float *out, int64_t ld, void *stream) {
    MTGS_REQUIRE(N >= 0 && E >= 0 && E <= DEFORM_MAX_COND && x_freqs >= 0 && x_freqs <= DEFORM_MAX_FREQS && t_freqs >= 0 &&
                     t_freqs <= DEFORM_MAX_FREQS,
                 MTGS_EINVAL, "mtgs_deform_embed: N=%lld E=%d (<= %d) frequencies %d / %d (<= %d)", (long long)N, E, DEFORM_MAX_COND,
                 x_freqs, t_freqs, DEFORM_MAX_FREQS);
    MTGS_REQUIRE(ld >= 3 + 6 * x_freqs + 1 + 2 * t_freqs + E, MTGS_EINVAL, "mtgs_deform_embed: row stride %lld too small", (long long)ld);
    MTGS_REQUIRE(height > 0.f, MTGS_EINVAL, "mtgs_deform_embed: instance height must be positive");
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(means && out && (cond || E == 0), MTGS_EINVAL, "mtgs_deform_embed: null pointer");
    deform_embed_kernel<<<(unsigned)ceil_div64(N, 256), 256, 0, (hipStream_t)stream>>>(N, means, height, t, cond, E, x_freqs, t_freqs,
                                                                                      out, ld);
    MTGS_CHECK_LAUNCH("mtgs_deform_embed");
    return MTGS_OK;
}
