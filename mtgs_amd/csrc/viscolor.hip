// viscolor.hip -- colours of the VISIBLE Gaussians only, forward and backward (visibility-first node path).
//
// MTGS evaluates spherical harmonics + clamp for every Gaussian of every node at every step
// (/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:309-322, multi_color_gaussian_splatting.py:77-101,
// collected by mtgs_scene_graph.py:408-461), 192 B of coefficients per Gaussian, although one camera sees ~15 % of a road
// block; gsplat's own `sh_degree` path masks SH with radii > 0.  Here the node kernels run geometry-only (node.hip,
// skip_colors), the front end projects and ranks the visible Gaussians with the colour channels of their records left open
// (front.hip, color_mode 2), and this file fills them: one 16-lane DPP row per visible Gaussian, lane k = SH basis k
// (sh_lane.hpp, the layout of sh_fwd_k16_kernel / node.hip), coefficients read in place through the node's row strides.
// Backward: the compositing backward's compact rows hold d L / d rgb per visible Gaussian; lane k writes
// basis_k(dir) * mask * v_rgb, i.e. a 192-byte coefficient-gradient ROW per visible Gaussian -- the optimizer consumes the
// rows through a row map (adam.hip) and the dense [N, (T,) K, 3] gradients (zeros for ~85 % of the rows, and for every
// other traversal) are never written.  Roofline: HBM, ~250 B per VISIBLE Gaussian per direction.
#include "common.hpp"
#include "sh_lane.hpp"
#include "raster_rec.hpp"

namespace {

constexpr int VC_BLOCK = 256, VC_ROWS_PER_BLOCK = VC_BLOCK / 16;

__device__ __forceinline__ float mul_rounded(float a, float b) {   // (see node.hip: keeps the colour independent of the lane)
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
struct F3 { float x, y, z; };
// (non-temporal, as sh_fwd_k16_kernel's: the coefficient rows are read once per frame and would only push the records and gradient
//  rows of this frame out of the Infinity Cache -- 46.7 -> 41.6 us for the call at the headline workload, with 64 instead of 128
//  Gaussians per workgroup 37.6: profiles/r06_viscolor_ab.txt)
__device__ __forceinline__ F3 vc_load3(const float *p) {
#ifndef MTGS_VC_PLAIN_LOADS
    typedef float f3v __attribute__((ext_vector_type(3)));
    const f3v v = __builtin_nontemporal_load(reinterpret_cast<const f3v *>(p));
    return F3{v.x, v.y, v.z};
#else
    return *reinterpret_cast<const F3 *>(p);
#endif
}

// A workgroup owns VC_ROWS = 64 consecutive visible Gaussians.  Phase 1: thread i < 64 resolves row i -- Gaussian index,
// node (binary search over the table's `start`, in LDS when the table is small), the three coefficient row addresses --
// into LDS and ISSUES the load of its direction; phase 2: every 16-lane DPP row walks VC_STEPS = 4 of them with all coefficient
// loads issued before the first use, and only then the directions are normalised and published (publish_dirs): index -> {direction,
// coefficients} are two dependent round trips, not three.  (One row per 16 lanes and four dependent global round trips per wave measured 77 us for 465k visible Gaussians:
// latency, not bandwidth.)
#ifndef MTGS_VC_STEPS
#define MTGS_VC_STEPS 4
#endif
constexpr int VC_STEPS = MTGS_VC_STEPS, VC_ROWS = VC_ROWS_PER_BLOCK * VC_STEPS, VC_LDS_NODES = 128;
static_assert(VC_ROWS == MTGS_VIS_COLOR_ROWS || MTGS_VC_STEPS != 4, "include/mtgs_rast.h: MTGS_VIS_COLOR_ROWS");
struct RowInfo {
    const float *dc, *dc_add, *rest;   // addresses of this Gaussian's coefficient rows (dc_add nullable)
    float dx, dy, dz;                  // unit view direction
    float inorm;                       // 1 / |mean - cam|
    int32_t k_rest, use_sh;
};

// coef_rows (nullable): the coefficients of visible Gaussian r as ONE compact row [dc 3 | dc_add 3 | rest 3 k_rest] at
// coef_rows + r * coef_stride (the optimizer's peek: mtgs_adam_step, MTGS_ADAM_ROWS_PEEK) instead of the nodes' tensors.
struct RawDir { float x, y, z; bool on; };
__device__ __forceinline__ RawDir resolve_rows(const mtgs_node_desc *__restrict__ table, int n_nodes, const float *__restrict__ cam_pos,
                                               const float *__restrict__ means, const int32_t *__restrict__ vis_ids, int64_t r0,
                                               int64_t n_vis, RowInfo *s_row, int64_t *s_start,
                                               const float *__restrict__ coef_rows = nullptr, int64_t coef_stride = 0,
                                               const uint8_t *__restrict__ row_flags = nullptr,
                                               const float *__restrict__ dirs = nullptr,
                                               const float *__restrict__ cotangents = nullptr, int64_t cot_stride = 0) {
    // cotangents (backward): d L / d rgb of row r at cotangents + r * cot_stride -- a row whose three values are zero (a visible
    // Gaussian the frame composited nothing from: ~60 % of them in an opaque scene) has an all-zero gradient whatever its direction,
    // so its direction is not fetched (a scattered 12-byte read each)
    const int tid = threadIdx.x;
    const bool small = n_nodes <= VC_LDS_NODES;
    if (small && n_nodes > 1) {
        for (int i = tid; i < n_nodes; i += VC_BLOCK) s_start[i] = table[i].start;
        __syncthreads();
    }
    RawDir rd{0.f, 0.f, 1.f, false};
    if (tid < VC_ROWS) {
        const int64_t r = r0 + tid;
        RowInfo ri;
        ri.dc = nullptr; ri.dc_add = nullptr; ri.rest = nullptr; ri.dx = 0.f; ri.dy = 0.f; ri.dz = 1.f; ri.inorm = 0.f; ri.k_rest = 0; ri.use_sh = 1;
        // row_flags: a Gaussian the frame composites nothing from gets a constant colour (whatever a zero SH sum activates to:
        // finite, and multiplied by a zero weight wherever the compositing meets it) -- nothing of it is read
        if (r < n_vis && (!row_flags || row_flags[r])) {
            const int64_t g = vis_ids[r];
            int lo = 0, hi = n_nodes - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                const int64_t st = small ? s_start[mid] : table[mid].start;
                if (st <= g) lo = mid; else hi = mid - 1;
            }
            const mtgs_node_desc &d = table[lo];
            const int64_t gl = g - d.start;
            ri.dc = d.features_dc + gl * d.dc_stride;
            ri.dc_add = d.features_dc_add ? d.features_dc_add + gl * d.dc_add_stride : nullptr;
            ri.rest = d.features_rest + gl * d.rest_stride;
            if (coef_rows) {
                ri.dc = coef_rows + r * coef_stride;
                ri.dc_add = d.features_dc_add ? ri.dc + 3 : nullptr;
                ri.rest = ri.dc + 6;
            }
            ri.k_rest = d.k_rest; ri.use_sh = d.use_sh;
            bool want_dir = true;
            if (cotangents) {
                const float *cg = cotangents + r * cot_stride;
                want_dir = cg[0] != 0.f || cg[1] != 0.f || cg[2] != 0.f;
            }
            // dirs (mtgs_vis_color_fwd_dirs): the caller's own view directions [N, 3] (MTGS: spherical_harmonics(n, viewdirs, colors) with
            // viewdirs computed in PyTorch) instead of mean - camera position; normalised as sh_fwd_k16_kernel does (sh.hip)
            rd.on = want_dir;
            if (!want_dir) {
            } else if (dirs) {
                const F3 dv = *reinterpret_cast<const F3 *>(dirs + g * 3);
                rd.x = dv.x; rd.y = dv.y; rd.z = dv.z;
            } else {
                const F3 m = *reinterpret_cast<const F3 *>(means + g * 3);
                rd.x = m.x - cam_pos[0]; rd.y = m.y - cam_pos[1]; rd.z = m.z - cam_pos[2];
            }
        }
        s_row[tid] = ri;
    }
    __syncthreads();
    return rd;
}
// ... the second half of phase 1: the resolving threads normalise the direction they loaded and publish it
__device__ __forceinline__ void publish_dirs(RowInfo *s_row, const RawDir rd) {
    if (threadIdx.x < VC_ROWS && rd.on) {
        const float inorm = 1.0f / sqrtf((rd.x * rd.x + rd.y * rd.y) + rd.z * rd.z);
        RowInfo &ri = s_row[threadIdx.x];
        ri.dx = rd.x * inorm; ri.dy = rd.y * inorm; ri.dz = rd.z * inorm; ri.inorm = inorm;
    }
    __syncthreads();
}

template <int DEG>
__global__ __launch_bounds__(VC_BLOCK) void vis_color_fwd_kernel(const mtgs_node_desc *__restrict__ table, int n_nodes,
                                                                 const float *__restrict__ cam_pos, const float *__restrict__ means,
                                                                 const int32_t *__restrict__ vis_ids, const int64_t *__restrict__ totals,
                                                                 int64_t cap_vis, float *__restrict__ recs, uint8_t *__restrict__ vis_mask,
                                                                 const float *__restrict__ coef_rows, int64_t coef_stride,
                                                                 const uint8_t *__restrict__ row_flags, const float *__restrict__ dirs) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    __shared__ RowInfo s_row[VC_ROWS];
    __shared__ int64_t s_start[VC_LDS_NODES];
    int64_t n_vis = *totals >> 32;
    if (n_vis > cap_vis) n_vis = cap_vis;
    const int64_t r0 = (int64_t)blockIdx.x * VC_ROWS;
    if (r0 >= n_vis) return;
    const RawDir rd = resolve_rows(table, n_nodes, cam_pos, means, vis_ids, r0, n_vis, s_row, s_start, coef_rows, coef_stride, row_flags, dirs);
    const int k = threadIdx.x & 15, sub = threadIdx.x >> 4;
    const ShLaneConst lc = sh_lane_const(k);
    F3 c[VC_STEPS];
#pragma unroll
    for (int it = 0; it < VC_STEPS; ++it) {      // every load of the 8 rows in flight before the first use
        const RowInfo &ri = s_row[it * VC_ROWS_PER_BLOCK + sub];
        c[it] = F3{0.f, 0.f, 0.f};
        const bool active = ri.dc && (ri.use_sh ? (k < NB && k - 1 < ri.k_rest) : (k == 0));
        if (active) {
            if (k == 0) {
                c[it] = vc_load3(ri.dc);
                if (ri.dc_add) {
                    const F3 a = vc_load3(ri.dc_add);
                    c[it].x += a.x; c[it].y += a.y; c[it].z += a.z;
                }
            } else {
                c[it] = vc_load3(ri.rest + (k - 1) * 3);
            }
        }
    }
    publish_dirs(s_row, rd);
#pragma unroll
    for (int it = 0; it < VC_STEPS; ++it) {
        const int row = it * VC_ROWS_PER_BLOCK + sub;
        const RowInfo &ri = s_row[row];
        const float b = ri.use_sh ? sh_lane_basis<DEG>(lc, ri.dx, ri.dy, ri.dz) : 1.f;
        const float sr = row16_sum(mul_rounded(b, c[it].x)), sg = row16_sum(mul_rounded(b, c[it].y)), sb = row16_sum(mul_rounded(b, c[it].z));
        const int64_t r = r0 + row;
        if (k != 0 || r >= n_vis) continue;
        F3 rgb;
        uint8_t mk = 7;
        if (ri.use_sh == 4) {   // gsplat's own sh_degree path (rendering.py): clamp_min(SH + 0.5, 0), no upper clamp
            const float x = sr + 0.5f, y = sg + 0.5f, z = sb + 0.5f;
            rgb = F3{x < 0.f ? 0.f : x, y < 0.f ? 0.f : y, z < 0.f ? 0.f : z};      // (a NaN stays a NaN, as in torch.clamp_min)
            mk = (uint8_t)((x >= 0.f) | ((y >= 0.f) << 1) | ((z >= 0.f) << 2));
        } else if (ri.use_sh) {
            const float x = sr + 0.5f, y = sg + 0.5f, z = sb + 0.5f;
            rgb = F3{x < 0.f ? 0.f : (x > 1.f ? 1.f : x), y < 0.f ? 0.f : (y > 1.f ? 1.f : y), z < 0.f ? 0.f : (z > 1.f ? 1.f : z)};
            // torch.clamp passes the gradient where min <= x <= max (inclusive): one bit per channel
            mk = (uint8_t)((x >= 0.f && x <= 1.f) | ((y >= 0.f && y <= 1.f) << 1) | ((z >= 0.f && z <= 1.f) << 2));
        } else {
            rgb = F3{1.f / (1.f + expf(-sr)), 1.f / (1.f + expf(-sg)), 1.f / (1.f + expf(-sb))};
        }
        float *dst = recs + r * REC_FLOATS + 8;
        dst[0] = rgb.x; dst[1] = rgb.y; dst[2] = rgb.z;
        vis_mask[r] = mk;
    }
}

template <int DEG>
__global__ __launch_bounds__(VC_BLOCK) void vis_color_bwd_kernel(const mtgs_node_desc *__restrict__ table, int n_nodes,
                                                                 const float *__restrict__ cam_pos, const float *__restrict__ means,
                                                                 const int32_t *__restrict__ vis_ids, const int64_t *__restrict__ totals,
                                                                 int64_t cap_vis, const float *__restrict__ grad_rows, int64_t row_stride,
                                                                 int col, const float *__restrict__ recs,
                                                                 const uint8_t *__restrict__ vis_mask, float *__restrict__ feat_rows,
                                                                 float *__restrict__ dir_rows, float *__restrict__ dir_part,
                                                                 float *__restrict__ dense_rows, const float *__restrict__ dirs) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    __shared__ RowInfo s_row[VC_ROWS];
    __shared__ int64_t s_start[VC_LDS_NODES];
    int64_t n_vis = *totals >> 32;
    if (n_vis > cap_vis) n_vis = cap_vis;
    const int64_t r0 = (int64_t)blockIdx.x * VC_ROWS;
    if (r0 >= n_vis) return;
    const int k = threadIdx.x & 15, sub = threadIdx.x >> 4;
    // the cotangents and masks of the workgroup's rows do not depend on the Gaussian index: in flight beside the index -> direction chain
    F3 vin[VC_STEPS];
    unsigned mkin[VC_STEPS];
#pragma unroll
    for (int it = 0; it < VC_STEPS; ++it) {
        const int64_t r = r0 + it * VC_ROWS_PER_BLOCK + sub;
        vin[it] = F3{0.f, 0.f, 0.f}; mkin[it] = 0u;
        if (r < n_vis) {
            const float *gr = grad_rows + r * row_stride + col;
            vin[it] = F3{gr[0], gr[1], gr[2]};
            mkin[it] = vis_mask[r];
        }
    }
    const RawDir rd = resolve_rows(table, n_nodes, cam_pos, means, vis_ids, r0, n_vis, s_row, s_start, nullptr, 0, nullptr, dirs,
                                   grad_rows + col, row_stride);
    publish_dirs(s_row, rd);
    const ShLaneConst lc = sh_lane_const(k);
#pragma unroll
    for (int it = 0; it < VC_STEPS; ++it) {
        const int row = it * VC_ROWS_PER_BLOCK + sub;
        const int64_t r = r0 + row;
        if (r >= n_vis) continue;
        const RowInfo &ri = s_row[row];
        F3 v = vin[it];
        float b;
        if (ri.use_sh) {
            const unsigned mk = mkin[it];   // torch.clamp passes the gradient where min <= x <= max
            v.x = (mk & 1u) ? v.x : 0.f; v.y = (mk & 2u) ? v.y : 0.f; v.z = (mk & 4u) ? v.z : 0.f;
            b = (k < NB && k - 1 < ri.k_rest) ? sh_lane_basis<DEG>(lc, ri.dx, ri.dy, ri.dz) : 0.f;
        } else {
            const float *y = recs + r * REC_FLOATS + 8;   // d sigmoid = y (1 - y)
            v.x *= y[0] * (1.f - y[0]); v.y *= y[1] * (1.f - y[1]); v.z *= y[2] * (1.f - y[2]);
            b = k == 0 ? 1.f : 0.f;
        }
        // colour gradients below 4 x the smallest normal float count as zero, so that "the coefficient-0 gradient b_0 v is zero"
        // and "every coefficient's gradient is zero" are the SAME statement (b_0 = 0.282: its product with such a v would
        // round to zero where a larger basis value's would not): mtgs_adam_step's zero_probe leaves exactly the rows lazy
        // whose gradient is all zero
        v.x = fabsf(v.x) < 4.8e-38f ? 0.f : v.x; v.y = fabsf(v.y) < 4.8e-38f ? 0.f : v.y; v.z = fabsf(v.z) < 4.8e-38f ? 0.f : v.z;
        if (dense_rows) {      // a ZEROED [N, 16, 3] gradient: the row of this Gaussian, if its cotangent is not zero (the zeros were
            //                    written beside a compositing kernel's work: mtgs_blend_fwd_packed(also_zero))
            if (v.x != 0.f || v.y != 0.f || v.z != 0.f)
                *reinterpret_cast<F3 *>(dense_rows + (int64_t)vis_ids[r] * 48 + k * 3) = F3{b * v.x, b * v.y, b * v.z};
        } else {
            *reinterpret_cast<F3 *>(feat_rows + r * 48 + k * 3) = F3{b * v.x, b * v.y, b * v.z};
        }
    }
    if (!dir_rows) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    // gsplat's view directions are differentiable (dirs = means - camera position, rendering.py): d L / d dirs of the
    // visible rows.  Lane k: w = <coefficient k, masked v_rgb>, times the gradient of ITS basis function
    // b_k = P(z) S(x, y) with respect to the unit direction; summed over the row, then through the normalisation.
#pragma unroll
    for (int it = 0; it < VC_STEPS; ++it) {
        const int row = it * VC_ROWS_PER_BLOCK + sub;
        const int64_t r = r0 + row;
        const RowInfo &ri = s_row[row];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (r < n_vis && ri.use_sh && k < NB && k - 1 < ri.k_rest && k > 0) {
            const F3 c = *reinterpret_cast<const F3 *>(ri.rest + (k - 1) * 3);
            const float *gr = grad_rows + r * row_stride + col;
            const unsigned mk = vis_mask[r];
            const float w = ((mk & 1u) ? c.x * gr[0] : 0.f) + ((mk & 2u) ? c.y * gr[1] : 0.f) + ((mk & 4u) ? c.z * gr[2] : 0.f);
            const float x = ri.dx, y = ri.dy, z = ri.dz;
            const float P = lc.a0 + z * (lc.a1 + z * (lc.a2 + z * lc.a3)), dP = lc.a1 + z * (2.f * lc.a2 + 3.f * z * lc.a3);
            float S = 1.f, Sx = 0.f, Sy = 0.f;
            switch (lc.sel) {
                case 1: S = x; Sx = 1.f; break;
                case 2: S = y; Sy = 1.f; break;
                case 3: S = 2.f * x * y; Sx = 2.f * y; Sy = 2.f * x; break;
                case 4: S = x * x - y * y; Sx = 2.f * x; Sy = -2.f * y; break;
                case 5: S = 3.f * x * x * y - y * y * y; Sx = 6.f * x * y; Sy = 3.f * (x * x - y * y); break;
                case 6: S = x * x * x - 3.f * x * y * y; Sx = 3.f * (x * x - y * y); Sy = -6.f * x * y; break;
                default: break;
            }
            gx = w * P * Sx; gy = w * P * Sy; gz = w * dP * S;
        }
        gx = row16_sum(gx); gy = row16_sum(gy); gz = row16_sum(gz);
        if (k == 0 && r < n_vis) {
            const float dot = gx * ri.dx + gy * ri.dy + gz * ri.dz;      // d (d / |d|): (v - n <n, v>) / |d|
            const F3 o = F3{(gx - ri.dx * dot) * ri.inorm, (gy - ri.dy * dot) * ri.inorm, (gz - ri.dz * dot) * ri.inorm};
            *reinterpret_cast<F3 *>(dir_rows + r * 3) = o;
            sx += o.x; sy += o.y; sz += o.z;
        }
    }
    // sum over the workgroup's rows (the camera position's gradient is minus the sum over all rows): one partial per
    // workgroup, summed by the caller -- 3 x n_vis same-address atomics would be the kernel time
    sx = wave_sum_to_lane63(sx); sy = wave_sum_to_lane63(sy); sz = wave_sum_to_lane63(sz);
    __shared__ float s_part[VC_BLOCK / 64][3];
    if ((threadIdx.x & 63) == 63) { s_part[threadIdx.x >> 6][0] = sx; s_part[threadIdx.x >> 6][1] = sy; s_part[threadIdx.x >> 6][2] = sz; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < VC_BLOCK / 64; ++w) t += s_part[w][threadIdx.x];
        dir_part[(int64_t)blockIdx.x * 3 + threadIdx.x] = t;
    }
}

// Dense expansion of the coefficient-gradient rows for autograd callers (gsplat's rasterization(sh_degree=...) hands its
// `colors` gradient back as a dense [N, K, 3] tensor): out[n, :width] = row_of[n] >= 0 ? rows[row_of[n], :width] : 0, every
// byte written once (width a multiple of 4 floats... the tail of a row is written element-wise).
__global__ __launch_bounds__(256) void rows_expand_kernel(int64_t N, int width, const int32_t *__restrict__ row_of,
                                                          const float *__restrict__ rows, int64_t row_stride, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N * width) return;
    const int64_t n = e / width;
    const int c = (int)(e - n * width);
    const int32_t r = row_of[n];
    out[e] = r >= 0 ? rows[(int64_t)r * row_stride + c] : 0.f;
}
// the same, 16 bytes per lane (width and row_stride multiples of 4, 16-byte aligned pointers): four float4 per thread, the 64-bit
// division once per workgroup, the dense output streamed past the caches (it is 384 MB at 2M Gaussians x 48 floats and nobody
// reads it before the optimizer)
typedef float vc_f4 __attribute__((ext_vector_type(4)));
constexpr int RX_PER = 4;
__global__ __launch_bounds__(256) void rows_expand4_kernel(int64_t N, int width4, const int32_t *__restrict__ row_of,
                                                           const float4 *__restrict__ rows, int64_t row_stride4, float4 *__restrict__ out) {
    const int64_t base = (int64_t)blockIdx.x * (256 * RX_PER), total = N * width4;
    const int64_t n0 = base / width4;
    const unsigned rem0 = (unsigned)(base - n0 * width4);
    float4 v[RX_PER];
    int64_t e[RX_PER];
#pragma unroll
    for (int u = 0; u < RX_PER; ++u) {
        const unsigned local = rem0 + (unsigned)(u * 256 + threadIdx.x);
        const unsigned dn = local / (unsigned)width4, c = local - dn * (unsigned)width4;
        e[u] = base + u * 256 + threadIdx.x;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e[u] < total) {
            const int32_t r = row_of[n0 + dn];
            if (r >= 0) v[u] = rows[(int64_t)r * row_stride4 + c];
        }
    }
#pragma unroll
    for (int u = 0; u < RX_PER; ++u)
        if (e[u] < total) {
            vc_f4 t = {v[u].x, v[u].y, v[u].z, v[u].w};
            __builtin_nontemporal_store(t, reinterpret_cast<vc_f4 *>(out + e[u]));
        }
}

#define MTGS_VC_DISPATCH(KERNEL, ...)                                                              \
    switch (degree) {                                                                              \
        case 0: KERNEL<0><<<grid, VC_BLOCK, 0, st>>>(__VA_ARGS__); break;                          \
        case 1: KERNEL<1><<<grid, VC_BLOCK, 0, st>>>(__VA_ARGS__); break;                          \
        case 2: KERNEL<2><<<grid, VC_BLOCK, 0, st>>>(__VA_ARGS__); break;                          \
        default: KERNEL<3><<<grid, VC_BLOCK, 0, st>>>(__VA_ARGS__); break;                         \
    }

}  // namespace

extern "C" int mtgs_vis_color_fwd(int n_nodes, const mtgs_node_desc *table, int degree, const float *cam_pos, const float *means,
                                  const int32_t *vis_ids, const int64_t *totals, int64_t cap_vis, float *recs, uint8_t *vis_mask,
                                  const float *coef_rows, int64_t coef_stride, const uint8_t *row_flags, void *stream) {
    return mtgs_vis_color_fwd_dirs(n_nodes, table, degree, cam_pos, means, vis_ids, totals, cap_vis, recs, vis_mask, coef_rows, coef_stride,
                                   row_flags, nullptr, stream);
}

extern "C" int mtgs_vis_color_fwd_dirs(int n_nodes, const mtgs_node_desc *table, int degree, const float *cam_pos, const float *means,
                                       const int32_t *vis_ids, const int64_t *totals, int64_t cap_vis, float *recs, uint8_t *vis_mask,
                                       const float *coef_rows, int64_t coef_stride, const uint8_t *row_flags, const float *dirs,
                                       void *stream) {
    MTGS_REQUIRE(n_nodes > 0 && degree >= 0 && degree <= 3 && cap_vis >= 0, MTGS_EINVAL, "mtgs_vis_color_fwd: bad sizes (degree <= 3)");
    MTGS_REQUIRE(!coef_rows || coef_stride >= 51, MTGS_EINVAL, "mtgs_vis_color_fwd: coef_stride=%lld (a row is dc 3 | dc_add 3 | rest 45)",
                 (long long)coef_stride);
    if (cap_vis == 0) return MTGS_OK;
    MTGS_REQUIRE(table && (dirs || (cam_pos && means)) && vis_ids && totals && recs && vis_mask, MTGS_EINVAL,
                 "mtgs_vis_color_fwd: null pointer (directions: dirs, or means and cam_pos)");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div64(cap_vis, VC_ROWS);
    MTGS_VC_DISPATCH(vis_color_fwd_kernel, table, n_nodes, cam_pos, means, vis_ids, totals, cap_vis, recs, vis_mask, coef_rows,
                     coef_stride, row_flags, dirs)
    MTGS_CHECK_LAUNCH("mtgs_vis_color_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_vis_color_bwd(int n_nodes, const mtgs_node_desc *table, int degree, const float *cam_pos, const float *means,
                                  const int32_t *vis_ids, const int64_t *totals, int64_t cap_vis, const float *grad_rows,
                                  int64_t row_stride, int col, const float *recs, const uint8_t *vis_mask, float *feat_rows,
                                  float *dir_rows, float *dir_part, float *dense_rows, void *stream) {
    return mtgs_vis_color_bwd_dirs(n_nodes, table, degree, cam_pos, means, vis_ids, totals, cap_vis, grad_rows, row_stride, col, recs,
                                   vis_mask, feat_rows, dir_rows, dir_part, dense_rows, nullptr, stream);
}

extern "C" int mtgs_vis_color_bwd_dirs(int n_nodes, const mtgs_node_desc *table, int degree, const float *cam_pos, const float *means,
                                       const int32_t *vis_ids, const int64_t *totals, int64_t cap_vis, const float *grad_rows,
                                       int64_t row_stride, int col, const float *recs, const uint8_t *vis_mask, float *feat_rows,
                                       float *dir_rows, float *dir_part, float *dense_rows, const float *dirs, void *stream) {
    MTGS_REQUIRE(n_nodes > 0 && degree >= 0 && degree <= 3 && cap_vis >= 0 && row_stride >= col + 3 && col >= 0, MTGS_EINVAL,
                 "mtgs_vis_color_bwd: bad sizes");
    if (cap_vis == 0) return MTGS_OK;
    MTGS_REQUIRE(table && (dirs || (cam_pos && means)) && vis_ids && totals && grad_rows && recs && vis_mask && (feat_rows || dense_rows) &&
                     (!dir_rows == !dir_part),
                 MTGS_EINVAL, "mtgs_vis_color_bwd: null pointer (dir_rows and dir_part go together; feat_rows or dense_rows)");
    // (dense_rows: ONE zeroed [N, 16, 3] buffer in collected order -- with several nodes, each node's gradient is its slice
    //  [start, start + n) of it, which takes plain K = 16 coefficient rows in every node: sh_direction_source / sh_coefficient_source)
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div64(cap_vis, VC_ROWS);
    MTGS_VC_DISPATCH(vis_color_bwd_kernel, table, n_nodes, cam_pos, means, vis_ids, totals, cap_vis, grad_rows, row_stride, col, recs,
                     vis_mask, feat_rows, dir_rows, dir_part, dense_rows, dirs)
    MTGS_CHECK_LAUNCH("mtgs_vis_color_bwd");
    return MTGS_OK;
}

extern "C" int mtgs_rows_expand(int64_t N, int width, const int32_t *row_of, const float *rows, int64_t row_stride, float *out,
                                void *stream) {
    MTGS_REQUIRE(N >= 0 && width > 0 && row_stride >= width, MTGS_EINVAL, "mtgs_rows_expand: bad sizes");
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(row_of && rows && out, MTGS_EINVAL, "mtgs_rows_expand: null pointer");
    MTGS_REQUIRE(N * width < ((int64_t)1 << 39), MTGS_EINVAL, "mtgs_rows_expand: too many elements");
    if (width % 4 == 0 && row_stride % 4 == 0 && (((uintptr_t)rows | (uintptr_t)out) & 15) == 0)
        rows_expand4_kernel<<<(unsigned)ceil_div64(N * (width / 4), 256 * RX_PER), 256, 0, (hipStream_t)stream>>>(
            N, width / 4, row_of, (const float4 *)rows, row_stride / 4, (float4 *)out);
    else
        rows_expand_kernel<<<(unsigned)ceil_div64(N * width, 256), 256, 0, (hipStream_t)stream>>>(N, width, row_of, rows, row_stride, out);
    MTGS_CHECK_LAUNCH("mtgs_rows_expand");
    return MTGS_OK;
}
