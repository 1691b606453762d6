// loss.hip -- masked SSIM between the rendered and the ground-truth image, forward and backward, fused
// (the consumer side of the rasterization path).
//
// Restates mtgs.utils.ssim.MaskedSSIM(data_range=1.0, size_average=True, channel=3)(gt, pred, mask)
// (/root/reference/mtgs/utils/ssim.py:57-108 `_ssim`, :111-190 `ssim`; called from
// /root/reference/mtgs/scene_model/mtgs_scene_graph.py:322, :831-841): an 11-tap separable Gaussian window
// (sigma 1.5, 'valid' correlation) gives mu_x, mu_y, E[x^2], E[y^2], E[xy] per pixel and channel,
//     ssim = ((2 mu_x mu_y + C1) / (mu_x^2 + mu_y^2 + C1)) * ((2 s_xy + C2) / (s_x^2 + s_y^2 + C2)),
// averaged over the masked elements of the (H-10) x (W-10) map (mask cropped by the window margin).
// PyTorch runs this as 5 grouped convolutions x 2 passes + ~15 elementwise kernels forward and twice that backward
// over NCHW copies of the images; here the images stay in the rasterizer's [H,W,3] layout, one kernel computes the
// five filtered maps in LDS and the per-pixel SSIM together with the three gradient maps the backward needs, and a
// second kernel applies the transposed window to those maps.  The masked mean is a two-level sum in a fixed order
// (per-block partials, then one block): deterministic.
//
// Roofline: HBM (forward reads 2 x 12 B per pixel and writes 36 B of gradient maps; backward reads 36 + 24, writes 12).
#include "common.hpp"

namespace {

constexpr int WIN = 11, HALO = WIN - 1, TILE = 16, IN_TILE = TILE + HALO;  // 26
struct Window { float w[WIN]; };

__device__ __forceinline__ float block_sum_256(float v, float *lds) {
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// gmaps[(oy * OW + ox) * 9 + c * 3 + {0,1,2}] = mask * d ssim / d {mu_y, E[y^2], E[xy]}   (nullable)
__global__ __launch_bounds__(256) void ssim_fwd_kernel(int H, int W, const float *__restrict__ X, const float *__restrict__ Y,
                                                       const uint8_t *__restrict__ mask, const Window win, float C1, float C2,
                                                       float *__restrict__ gmaps, float *__restrict__ partials) {
    __shared__ float sX[IN_TILE][IN_TILE + 1], sY[IN_TILE][IN_TILE + 1];
    __shared__ float sH[5][IN_TILE][TILE + 1];
    __shared__ float s_red[4];
    const int OW = W - HALO, OH = H - HALO;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    const int ox = x0 + tx, oy = y0 + ty;
    const bool valid = ox < OW && oy < OH;
    // the mask is cropped by the window margin: output (oy, ox) <-> image pixel (oy + 5, ox + 5)
    const float m = valid ? (mask ? (mask[(int64_t)(oy + HALO / 2) * W + ox + HALO / 2] ? 1.f : 0.f) : 1.f) : 0.f;
    float sum = 0.f;
    // The three channels are three passes over the LDS tiles; the NEXT channel's pixels are fetched into registers while the
    // current one is filtered (a workgroup is resident for the whole image at 960x540: its time was three global round trips).
    constexpr int PER = (IN_TILE * IN_TILE + 255) / 256;
    float rx[PER], ry[PER];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * 256;
            const int r = i / IN_TILE, q = i - r * IN_TILE;
            const int gy = y0 + r, gx = x0 + q;
            const bool in = i < IN_TILE * IN_TILE && gy < H && gx < W;
            const int64_t a = ((int64_t)gy * W + gx) * 3 + c;
            rx[u] = in ? X[a] : 0.f;
            ry[u] = in ? Y[a] : 0.f;
        }
    };
    fetch(0);
    for (int c = 0; c < 3; ++c) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * 256;
            if (i < IN_TILE * IN_TILE) {
                const int r = i / IN_TILE, q = i - r * IN_TILE;
                sX[r][q] = rx[u];
                sY[r][q] = ry[u];
            }
        }
        __syncthreads();
        if (c < 2) fetch(c + 1);
        for (int i = tid; i < IN_TILE * TILE; i += 256) {   // horizontal pass: 26 rows x 16 columns
            const int r = i >> 4, q = i & 15;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float x = sX[r][q + k], y = sY[r][q + k], w = win.w[k];
                a0 += w * x; a1 += w * y; a2 += w * (x * x); a3 += w * (y * y); a4 += w * (x * y);
            }
            sH[0][r][q] = a0; sH[1][r][q] = a1; sH[2][r][q] = a2; sH[3][r][q] = a3; sH[4][r][q] = a4;
        }
        __syncthreads();
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;   // vertical pass
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float w = win.w[k];
            mu1 += w * sH[0][ty + k][tx]; mu2 += w * sH[1][ty + k][tx];
            e11 += w * sH[2][ty + k][tx]; e22 += w * sH[3][ty + k][tx]; e12 += w * sH[4][ty + k][tx];
        }
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
        const float nA = 2.f * mu12 + C1, dA = mu1_sq + mu2_sq + C1;
        const float nB = 2.f * s12 + C2, dB = s1 + s2 + C2;
        const float A = nA / dA, B = nB / dB;
        sum += m * (A * B);
        if (gmaps && valid) {
            // y = the second argument (the prediction).  d/d mu_y, d/d E[y^2], d/d E[xy]:
            const float dA_dmu2 = (2.f * mu1 - A * 2.f * mu2) / dA;
            const float dB_dmu2 = (-2.f * mu1 + B * 2.f * mu2) / dB;   // n_B has -2 mu_x, d_B has -2 mu_y
            const float g_mu = m * (dA_dmu2 * B + A * dB_dmu2);
            const float g_e22 = m * (A * (-B / dB));
            const float g_e12 = m * (A * (2.f / dB));
            float *g = gmaps + ((int64_t)oy * OW + ox) * 9 + c * 3;
            g[0] = g_mu; g[1] = g_e22; g[2] = g_e12;
        }
    }
    const float bs = block_sum_256(sum, s_red);
    const float bc = block_sum_256(3.f * m, s_red);
    if (tid == 0) {
        const int64_t b = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
        partials[b * 2] = bs; partials[b * 2 + 1] = bc;
    }
}

// out[0] = masked mean, out[1] = number of masked elements (fixed summation order)
__global__ __launch_bounds__(256) void ssim_finish_kernel(int64_t nblocks, const float *__restrict__ partials,
                                                          float *__restrict__ out) {
    __shared__ float s_red[4];
    float s = 0.f, c = 0.f;
    for (int64_t b = threadIdx.x; b < nblocks; b += 256) { s += partials[b * 2]; c += partials[b * 2 + 1]; }
    const float ts = block_sum_256(s, s_red);
    const float tc = block_sum_256(c, s_red);
    if (threadIdx.x == 0) { out[0] = ts / tc; out[1] = tc; }
}

// the same with the reference's guard for an empty mask (mtgs_scene_graph.py:857-858: the depth term is 0 then)
__global__ __launch_bounds__(256) void mean_or_zero_finish_kernel(int64_t nblocks, const float *__restrict__ partials,
                                                                  float *__restrict__ out) {
    __shared__ float s_red[4];
    float s = 0.f, c = 0.f;
    for (int64_t b = threadIdx.x; b < nblocks; b += 256) { s += partials[b * 2]; c += partials[b * 2 + 1]; }
    const float ts = block_sum_256(s, s_red);
    const float tc = block_sum_256(c, s_red);
    if (threadIdx.x == 0) { out[0] = tc > 0.f ? ts / tc : 0.f; out[1] = tc; }
}

// v_Y[q] = (v_out / count) * sum_p w2d(q - p) * (g_mu[p] + 2 Y[q] g_e22[p] + X[q] g_e12[p])
__global__ __launch_bounds__(256) void ssim_bwd_kernel(int H, int W, const float *__restrict__ X, const float *__restrict__ Y,
                                                       const float *__restrict__ gmaps, const Window win,
                                                       const float *__restrict__ v_out, const float *__restrict__ fwd_out,
                                                       float *__restrict__ v_Y) {
    __shared__ float sG[3][IN_TILE][IN_TILE + 1];
    __shared__ float sH[3][IN_TILE][TILE + 1];
    const int OW = W - HALO, OH = H - HALO;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    const int px = x0 + tx, py = y0 + ty;
    const float scale = v_out[0] / fwd_out[1];
    // (the next channel's maps and pixels are fetched while the current channel is filtered, as in the forward)
    constexpr int PER = (IN_TILE * IN_TILE + 255) / 256;
    float rg[PER][3], xn = 0.f, yn = 0.f;
    const bool own = px < W && py < H;
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            // pixel (py, px) receives from outputs (py - a, px - b), a, b in [0, 10]: rows y0-10 .. y0+15
            const int i = tid + u * 256;
            const int r = i / IN_TILE, q = i - r * IN_TILE;
            const int gy = y0 - HALO + r, gx = x0 - HALO + q;
            const bool in = i < IN_TILE * IN_TILE && gy >= 0 && gx >= 0 && gy < OH && gx < OW;
            const float *g = gmaps + ((int64_t)gy * OW + gx) * 9 + c * 3;
            rg[u][0] = in ? g[0] : 0.f; rg[u][1] = in ? g[1] : 0.f; rg[u][2] = in ? g[2] : 0.f;
        }
        if (own) { const int64_t a = ((int64_t)py * W + px) * 3 + c; xn = X[a]; yn = Y[a]; }
    };
    fetch(0);
    for (int c = 0; c < 3; ++c) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * 256;
            if (i < IN_TILE * IN_TILE) {
                const int r = i / IN_TILE, q = i - r * IN_TILE;
                sG[0][r][q] = rg[u][0]; sG[1][r][q] = rg[u][1]; sG[2][r][q] = rg[u][2];
            }
        }
        const float xc = xn, yc = yn;
        __syncthreads();
        if (c < 2) fetch(c + 1);
        for (int i = tid; i < IN_TILE * TILE; i += 256) {   // horizontal: column q <-> tile columns q .. q+10, weight w[10-k]
            const int r = i >> 4, q = i & 15;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float w = win.w[HALO - k];
                a0 += w * sG[0][r][q + k]; a1 += w * sG[1][r][q + k]; a2 += w * sG[2][r][q + k];
            }
            sH[0][r][q] = a0; sH[1][r][q] = a1; sH[2][r][q] = a2;
        }
        __syncthreads();
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float w = win.w[HALO - k];
            t0 += w * sH[0][ty + k][tx]; t1 += w * sH[1][ty + k][tx]; t2 += w * sH[2][ty + k][tx];
        }
        if (own) {
            const int64_t a = ((int64_t)py * W + px) * 3 + c;
            v_Y[a] = scale * (t0 + 2.f * yc * t1 + xc * t2);
        }
    }
}

// ---- masked L1 of the same loss head: torch.abs(gt - pred)[mask].mean()  (mtgs_scene_graph.py:823) ------------------
// PyTorch runs the boolean-mask indexing as nonzero + gather forward and a sorted index_put backward (~300 us per
// iteration at 960x540); here: per-block partial sums in a fixed order, and d/d pred = mask * sign(pred - gt) / count.
constexpr int L1_PIX = 1024;  // pixels per block
__global__ __launch_bounds__(256) void l1_fwd_kernel(int64_t n_pix, int ch, const float *__restrict__ X, const float *__restrict__ Y,
                                                     const uint8_t *__restrict__ mask, float *__restrict__ partials) {
    __shared__ float s_red[4];
    float s = 0.f, c = 0.f;
    const int64_t p0 = (int64_t)blockIdx.x * L1_PIX;
    for (int i = threadIdx.x; i < L1_PIX; i += 256) {
        const int64_t p = p0 + i;
        if (p < n_pix && (!mask || mask[p])) {
            if (ch == 3) {
                s += (fabsf(X[p * 3] - Y[p * 3]) + fabsf(X[p * 3 + 1] - Y[p * 3 + 1])) + fabsf(X[p * 3 + 2] - Y[p * 3 + 2]);
            } else {
                float t = 0.f;
                for (int k = 0; k < ch; ++k) t += fabsf(X[p * ch + k] - Y[p * ch + k]);
                s += t;
            }
            c += (float)ch;
        }
    }
    const float bs = block_sum_256(s, s_red);
    const float bc = block_sum_256(c, s_red);
    if (threadIdx.x == 0) { partials[blockIdx.x * 2] = bs; partials[blockIdx.x * 2 + 1] = bc; }
}
__global__ __launch_bounds__(256) void l1_bwd_kernel(int64_t n_pix, int ch, const float *__restrict__ X, const float *__restrict__ Y,
                                                     const uint8_t *__restrict__ mask, const float *__restrict__ v_out,
                                                     const float *__restrict__ fwd_out, float *__restrict__ v_Y) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pix) return;
    const float scale = (!mask || mask[p]) ? v_out[0] / fwd_out[1] : 0.f;
    for (int c = 0; c < ch; ++c) {
        const float d = Y[p * ch + c] - X[p * ch + c];
        v_Y[p * ch + c] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);   // torch: sign(0) = 0
    }
}

// ---- the lidar depth term of the same loss head (mtgs_scene_graph.py:849-856, 875-879):
//   mask = (gt > lo) & (gt < hi) & combined_mask;  loss = |1 / (gt + eps) - 1 / (pred + eps)|[mask].mean()
// PyTorch: two compares, two ands, two adds, two reciprocals (1 / x is reciprocal then mul), the masked mean and their
// backward chain -- a dozen launches for one number per pixel.  One launch per direction; the mask is a by-product (the depth
// NCC term uses the same one, :891).  Arithmetic as torch: 1 / x = IEEE division, differences in that order.
__global__ __launch_bounds__(256) void inv_depth_l1_fwd_kernel(int64_t n_pix, const float *__restrict__ gt, const float *__restrict__ pred,
                                                               const uint8_t *__restrict__ mask, float lo, float hi, float eps,
                                                               uint8_t *__restrict__ mask_out, float *__restrict__ partials) {
    __shared__ float s_red[4];
    float s = 0.f, c = 0.f;
    const int64_t p0 = (int64_t)blockIdx.x * L1_PIX;
    for (int i = threadIdx.x; i < L1_PIX; i += 256) {
        const int64_t p = p0 + i;
        if (p >= n_pix) continue;
        const float g = gt[p];
        const bool m = g > lo && g < hi && (!mask || mask[p]);
        if (mask_out) mask_out[p] = m ? 1 : 0;
        if (m) { s += fabsf(1.0f / (g + eps) - 1.0f / (pred[p] + eps)); c += 1.f; }
    }
    const float bs = block_sum_256(s, s_red);
    const float bc = block_sum_256(c, s_red);
    if (threadIdx.x == 0) { partials[blockIdx.x * 2] = bs; partials[blockIdx.x * 2 + 1] = bc; }
}
// d/d pred of |a - 1 / (pred + eps)| = sign(1 / (pred + eps) - a) * (-1 / (pred + eps)^2); torch's chain: grad of abs = sign(.),
// of the subtraction -1, of reciprocal -y^2.
__global__ __launch_bounds__(256) void inv_depth_l1_bwd_kernel(int64_t n_pix, const float *__restrict__ gt, const float *__restrict__ pred,
                                                               const uint8_t *__restrict__ mask, float lo, float hi, float eps,
                                                               const float *__restrict__ v_out, const float *__restrict__ fwd_out,
                                                               float *__restrict__ v_pred) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pix) return;
    const float g = gt[p];
    float out = 0.f;
    if (g > lo && g < hi && (!mask || mask[p])) {
        const float a = 1.0f / (g + eps), y = 1.0f / (pred[p] + eps), d = a - y;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        out = (v_out[0] / fwd_out[1]) * sg * (y * y);      // d |a - y| / d y = -sign(a - y); d y / d pred = -y^2
    }
    v_pred[p] = out;
}

}  // namespace

static Window make_window(float sigma) {
    Window w;
    double s = 0.0;
    for (int i = 0; i < WIN; ++i) {
        // _fspecial_gauss_1d in float32: coords = arange(size) - size // 2; g = exp(-(coords**2) / (2 sigma**2)); g /= g.sum()
        const float c = (float)(i - WIN / 2);
        w.w[i] = expf(-(c * c) / (2.f * sigma * sigma));
        s += (double)w.w[i];
    }
    float fs = 0.f;
    for (int i = 0; i < WIN; ++i) fs += w.w[i];
    (void)s;
    for (int i = 0; i < WIN; ++i) w.w[i] /= fs;
    return w;
}

extern "C" int mtgs_ssim_workspace_floats(int width, int height, size_t *n) {
    MTGS_REQUIRE(width > HALO && height > HALO && n, MTGS_EINVAL, "mtgs_ssim_workspace_floats: image smaller than the 11x11 window");
    *n = (size_t)ceil_div64(width - HALO, TILE) * (size_t)ceil_div64(height - HALO, TILE) * 2;
    return MTGS_OK;
}

extern "C" int mtgs_ssim_fwd(int width, int height, const float *gt, const float *pred, const uint8_t *mask,
                             float win_sigma, float data_range, float K1, float K2, float *gmaps, float *partials,
                             float *out, void *stream) {
    MTGS_REQUIRE(width > HALO && height > HALO, MTGS_EINVAL, "mtgs_ssim_fwd: image %dx%d smaller than the 11x11 window", width, height);
    MTGS_REQUIRE(gt && pred && partials && out, MTGS_EINVAL, "mtgs_ssim_fwd: null pointer");
    const Window win = make_window(win_sigma);
    const float C1 = (K1 * data_range) * (K1 * data_range), C2 = (K2 * data_range) * (K2 * data_range);
    const dim3 grid((unsigned)ceil_div64(width - HALO, TILE), (unsigned)ceil_div64(height - HALO, TILE));
    hipStream_t st = (hipStream_t)stream;
    ssim_fwd_kernel<<<grid, 256, 0, st>>>(height, width, gt, pred, mask, win, C1, C2, gmaps, partials);
    ssim_finish_kernel<<<1, 256, 0, st>>>((int64_t)grid.x * grid.y, partials, out);
    MTGS_CHECK_LAUNCH("mtgs_ssim_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_ssim_bwd(int width, int height, const float *gt, const float *pred, const float *gmaps,
                             float win_sigma, const float *v_out, const float *fwd_out, float *v_pred, void *stream) {
    MTGS_REQUIRE(width > HALO && height > HALO, MTGS_EINVAL, "mtgs_ssim_bwd: image %dx%d smaller than the 11x11 window", width, height);
    MTGS_REQUIRE(gt && pred && gmaps && v_out && fwd_out && v_pred, MTGS_EINVAL, "mtgs_ssim_bwd: null pointer");
    const Window win = make_window(win_sigma);
    const dim3 grid((unsigned)ceil_div64(width, TILE), (unsigned)ceil_div64(height, TILE));
    ssim_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(height, width, gt, pred, gmaps, win, v_out, fwd_out, v_pred);
    MTGS_CHECK_LAUNCH("mtgs_ssim_bwd");
    return MTGS_OK;
}

extern "C" int mtgs_l1_workspace_floats(int width, int height, size_t *n) {
    MTGS_REQUIRE(width > 0 && height > 0 && n, MTGS_EINVAL, "mtgs_l1_workspace_floats: bad arguments");
    *n = (size_t)ceil_div64((int64_t)width * height, L1_PIX) * 2;
    return MTGS_OK;
}

extern "C" int mtgs_l1_fwd(int width, int height, int channels, const float *gt, const float *pred, const uint8_t *mask,
                           float *partials, float *out, void *stream) {
    MTGS_REQUIRE(width > 0 && height > 0 && channels >= 1 && channels <= 8, MTGS_EINVAL, "mtgs_l1_fwd: bad sizes");
    MTGS_REQUIRE(gt && pred && partials && out, MTGS_EINVAL, "mtgs_l1_fwd: null pointer");
    const int64_t n_pix = (int64_t)width * height, nblocks = ceil_div64(n_pix, L1_PIX);
    hipStream_t st = (hipStream_t)stream;
    l1_fwd_kernel<<<(unsigned)nblocks, 256, 0, st>>>(n_pix, channels, gt, pred, mask, partials);
    ssim_finish_kernel<<<1, 256, 0, st>>>(nblocks, partials, out);
    MTGS_CHECK_LAUNCH("mtgs_l1_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_l1_bwd(int width, int height, int channels, const float *gt, const float *pred, const uint8_t *mask,
                           const float *v_out, const float *fwd_out, float *v_pred, void *stream) {
    MTGS_REQUIRE(width > 0 && height > 0 && channels >= 1 && channels <= 8, MTGS_EINVAL, "mtgs_l1_bwd: bad sizes");
    MTGS_REQUIRE(gt && pred && v_out && fwd_out && v_pred, MTGS_EINVAL, "mtgs_l1_bwd: null pointer");
    const int64_t n_pix = (int64_t)width * height;
    l1_bwd_kernel<<<(unsigned)ceil_div64(n_pix, 256), 256, 0, (hipStream_t)stream>>>(n_pix, channels, gt, pred, mask, v_out, fwd_out, v_pred);
    MTGS_CHECK_LAUNCH("mtgs_l1_bwd");
    return MTGS_OK;
}

extern "C" int mtgs_inv_depth_l1_fwd(int width, int height, const float *gt_depth, const float *pred_depth, const uint8_t *mask,
                                     float lo, float hi, float eps, uint8_t *mask_out, float *partials, float *out, void *stream) {
    MTGS_REQUIRE(width > 0 && height > 0, MTGS_EINVAL, "mtgs_inv_depth_l1_fwd: bad sizes");
    MTGS_REQUIRE(gt_depth && pred_depth && partials && out, MTGS_EINVAL, "mtgs_inv_depth_l1_fwd: null pointer");
    const int64_t n_pix = (int64_t)width * height, nblocks = ceil_div64(n_pix, L1_PIX);
    hipStream_t st = (hipStream_t)stream;
    inv_depth_l1_fwd_kernel<<<(unsigned)nblocks, 256, 0, st>>>(n_pix, gt_depth, pred_depth, mask, lo, hi, eps, mask_out, partials);
    mean_or_zero_finish_kernel<<<1, 256, 0, st>>>(nblocks, partials, out);
    MTGS_CHECK_LAUNCH("mtgs_inv_depth_l1_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_inv_depth_l1_bwd(int width, int height, const float *gt_depth, const float *pred_depth, const uint8_t *mask,
                                     float lo, float hi, float eps, const float *v_out, const float *fwd_out, float *v_pred,
                                     void *stream) {
    MTGS_REQUIRE(width > 0 && height > 0, MTGS_EINVAL, "mtgs_inv_depth_l1_bwd: bad sizes");
    MTGS_REQUIRE(gt_depth && pred_depth && v_out && fwd_out && v_pred, MTGS_EINVAL, "mtgs_inv_depth_l1_bwd: null pointer");
    const int64_t n_pix = (int64_t)width * height;
    inv_depth_l1_bwd_kernel<<<(unsigned)ceil_div64(n_pix, 256), 256, 0, (hipStream_t)stream>>>(n_pix, gt_depth, pred_depth, mask, lo, hi,
                                                                                              eps, v_out, fwd_out, v_pred);
    MTGS_CHECK_LAUNCH("mtgs_inv_depth_l1_bwd");
    return MTGS_OK;
}
