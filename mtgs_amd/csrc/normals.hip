// normals.hip -- camera-space normals of the Gaussians, forward and backward (caller side of the path, SURVEY.md
// section 8a1: with `predict_normals` -- the shipped MTGS.py config -- MTGS blends 3 normal channels next to RGB).
//
// Restates MTGSSceneModel._get_gaussian_camera_space_normals
// (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:526-545; quat_to_rotmat = utils.py:14-41, wxyz, not normalised):
//     k       = argmin(scales)                         first of equal minima
//     col     = column k of quat_to_rotmat(quats)
//     n0      = col / max(|col|, 1e-12)
//     s       = dot(n0, normalize(cam_pos - means)) < 0 ? -1 : +1        (means detached)
//     normals = (s n0) @ camera_to_worlds[:3, :3]
// which PyTorch runs as ~25 launches per direction over all N Gaussians (one_hot, the [N,3,3] rotation matrices, bmm,
// a boolean-mask write that synchronises the host, matmul) followed by torch.cat([rgbs, normals]).  Here: one streaming
// kernel per direction, thread per Gaussian; the forward can write straight into columns 3..5 of the [N,6] colour
// tensor (and copy rgbs into columns 0..2), so there is no cat either.  Only `quats` receives a gradient.
// Roofline: HBM; N * (40 + 12) B forward (+ 24 B with the rgbs copy), N * (40 + 12 + 16) B backward.
#include "common.hpp"
#include "raster_rec.hpp"

namespace {
struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };

struct NormalGeom { F3 n0; float inv_len, sign; int k; };

__device__ __forceinline__ int argmin3(const F3 s) {   // torch.argmin: first of equal minima
    int k = 0;
    float m = s.x;
    if (s.y < m) { m = s.y; k = 1; }
    if (s.z < m) k = 2;
    return k;
}
__device__ __forceinline__ F3 rot_column(const F4 q, int k) {   // q = (w, x, y, z)
    const float w = q.x, x = q.y, y = q.z, z = q.w;
    if (k == 0) return F3{1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y)};
    if (k == 1) return F3{2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x)};
    return F3{2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y)};
}
__device__ __forceinline__ NormalGeom geometry(const F4 q, const F3 s, const F3 m, const float *__restrict__ c2w) {
    NormalGeom g;
    g.k = argmin3(s);
    const F3 col = rot_column(q, g.k);
    const float len = sqrtf((col.x * col.x + col.y * col.y) + col.z * col.z);
    g.inv_len = 1.0f / fmaxf(len, 1e-12f);
    g.n0 = F3{col.x * g.inv_len, col.y * g.inv_len, col.z * g.inv_len};
    float dx = c2w[3] - m.x, dy = c2w[7] - m.y, dz = c2w[11] - m.z;
    const float dinv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
    dx *= dinv; dy *= dinv; dz *= dinv;
    const float dot = (g.n0.x * dx + g.n0.y * dy) + g.n0.z * dz;
    g.sign = dot < 0.f ? -1.f : 1.f;   // NaN (camera exactly at the mean): no flip, as `dots < 0` in the reference
    return g;
}

// v (cotangent of the camera-space normal) -> cotangent of the quaternion
__device__ __forceinline__ F4 normal_vjp(const F4 q, const NormalGeom &g, const F3 v, const float *__restrict__ c2w) {
    // back through the camera rotation (v_world = R v) and the flip
    float vx = g.sign * ((c2w[0] * v.x + c2w[1] * v.y) + c2w[2] * v.z);
    float vy = g.sign * ((c2w[4] * v.x + c2w[5] * v.y) + c2w[6] * v.z);
    float vz = g.sign * ((c2w[8] * v.x + c2w[9] * v.y) + c2w[10] * v.z);
    // F.normalize: d (col / |col|) = (v - n0 <n0, v>) / |col|
    const float dot = (g.n0.x * vx + g.n0.y * vy) + g.n0.z * vz;
    vx = (vx - g.n0.x * dot) * g.inv_len; vy = (vy - g.n0.y * dot) * g.inv_len; vz = (vz - g.n0.z * dot) * g.inv_len;
    const float w = q.x, x = q.y, y = q.z, z = q.w;
    if (g.k == 0)        // col = (1 - 2(yy + zz), 2(xy + wz), 2(xz - wy))
        return F4{2.f * (z * vy - y * vz), 2.f * (y * vy + z * vz), (-4.f * y * vx + 2.f * x * vy) - 2.f * w * vz,
                  (-4.f * z * vx + 2.f * w * vy) + 2.f * x * vz};
    if (g.k == 1)        // col = (2(xy - wz), 1 - 2(xx + zz), 2(yz + wx))
        return F4{2.f * (x * vz - z * vx), (2.f * y * vx - 4.f * x * vy) + 2.f * w * vz, 2.f * (x * vx + z * vz),
                  (-2.f * w * vx - 4.f * z * vy) + 2.f * y * vz};
    // col = (2(xz + wy), 2(yz - wx), 1 - 2(xx + yy))
    return F4{2.f * (y * vx - x * vy), (2.f * z * vx - 2.f * w * vy) - 4.f * x * vz, (2.f * w * vx + 2.f * z * vy) - 4.f * y * vz,
              2.f * (x * vx + y * vy)};
}

__global__ __launch_bounds__(256) void normals_fwd_kernel(int64_t N, const float *__restrict__ quats,
                                                          const float *__restrict__ scales, const float *__restrict__ means,
                                                          const float *__restrict__ c2w, const float *__restrict__ rgbs,
                                                          float *__restrict__ out, int64_t out_stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const F4 q = *reinterpret_cast<const F4 *>(quats + i * 4);
    const F3 s = *reinterpret_cast<const F3 *>(scales + i * 3);
    const F3 m = *reinterpret_cast<const F3 *>(means + i * 3);
    const NormalGeom g = geometry(q, s, m, c2w);
    const float nx = g.sign * g.n0.x, ny = g.sign * g.n0.y, nz = g.sign * g.n0.z;
    // row vector times camera_to_worlds[:3, :3] (row-major [3,4])
    const F3 nc = F3{(nx * c2w[0] + ny * c2w[4]) + nz * c2w[8], (nx * c2w[1] + ny * c2w[5]) + nz * c2w[9],
                     (nx * c2w[2] + ny * c2w[6]) + nz * c2w[10]};
    float *row = out + i * out_stride;
    if (rgbs) {
        *reinterpret_cast<F3 *>(row) = *reinterpret_cast<const F3 *>(rgbs + i * 3);
        row += 3;
    }
    *reinterpret_cast<F3 *>(row) = nc;
}

__global__ __launch_bounds__(256) void normals_bwd_kernel(int64_t N, const float *__restrict__ quats,
                                                          const float *__restrict__ scales, const float *__restrict__ means,
                                                          const float *__restrict__ c2w, const float *__restrict__ v_out,
                                                          int64_t v_stride, float *__restrict__ g_quats) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const F4 q = *reinterpret_cast<const F4 *>(quats + i * 4);
    const F3 s = *reinterpret_cast<const F3 *>(scales + i * 3);
    const F3 m = *reinterpret_cast<const F3 *>(means + i * 3);
    const NormalGeom g = geometry(q, s, m, c2w);
    const F3 v = *reinterpret_cast<const F3 *>(v_out + i * v_stride);
    const F4 o = normal_vjp(q, g, v, c2w);
    *reinterpret_cast<F4 *>(g_quats + i * 4) = o;
}
// The backward for the VISIBLE Gaussians of one frame, added into the wire rows of the data-parallel gradient exchange
// (csrc/dp.hip, project_bwd.hip: row r = [v_mean 3 | v_quat 4 | ...] of Gaussian vis_ids[r]).  The normal channels are
// a function of each rank's OWN camera, so their gradient cannot be summed over the ranks and pulled back afterwards:
// the sender folds it into the quaternion gradient of its row before the row leaves.  v_normal = columns
// col .. col + 2 of the compositing backward's compact gradient rows G[n_vis, row_stride].
__global__ __launch_bounds__(256) void normals_bwd_rows_kernel(int64_t n_vis, const int32_t *__restrict__ vis_ids,
                                                               const float *__restrict__ quats, const float *__restrict__ scales,
                                                               const float *__restrict__ means, const float *__restrict__ c2w,
                                                               const float *__restrict__ G, int64_t row_stride, int col,
                                                               float *__restrict__ wire) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_vis) return;
    const int64_t i = vis_ids[r];
    const F4 q = *reinterpret_cast<const F4 *>(quats + i * 4);
    const F3 s = *reinterpret_cast<const F3 *>(scales + i * 3);
    const F3 m = *reinterpret_cast<const F3 *>(means + i * 3);
    const NormalGeom g = geometry(q, s, m, c2w);
    const F3 v = F3{G[r * row_stride + col], G[r * row_stride + col + 1], G[r * row_stride + col + 2]};
    const F4 o = normal_vjp(q, g, v, c2w);
    float *w = wire + r * 16 + 3;
    w[0] += o.x; w[1] += o.y; w[2] += o.z; w[3] += o.w;
}
// Visibility first (single process): the normals of the VISIBLE Gaussians only, straight into their records (channels
// `channel` .. + 2; mtgs_front_fwd color_mode 3 left them open), and their backward as rows [n_vis, 4] of quaternion
// gradients that mtgs_project_bwd adds to its own (x_quat_rows) -- no [N, 3] normal tensor, no dense gradient of it.
__global__ __launch_bounds__(256) void normals_fwd_rows_kernel(int64_t cap_vis, const int32_t *__restrict__ vis_ids,
                                                               const int64_t *__restrict__ totals, const float *__restrict__ quats,
                                                               const float *__restrict__ scales, const float *__restrict__ means,
                                                               const float *__restrict__ c2w, float *__restrict__ recs, int channel,
                                                               const uint8_t *__restrict__ row_flags) {
    int64_t n_vis = totals ? *totals >> 32 : cap_vis;
    if (n_vis > cap_vis) n_vis = cap_vis;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_vis) return;
    if (row_flags && !row_flags[r]) {   // nothing is composited from this Gaussian: any finite value will do, nothing is gathered
        float *dst = recs + r * REC_FLOATS + 8 + channel;
        dst[0] = 0.f; dst[1] = 0.f; dst[2] = 0.f;
        return;
    }
    const int64_t i = vis_ids[r];
    const F4 q = *reinterpret_cast<const F4 *>(quats + i * 4);
    const F3 s = *reinterpret_cast<const F3 *>(scales + i * 3);
    const F3 m = *reinterpret_cast<const F3 *>(means + i * 3);
    const NormalGeom g = geometry(q, s, m, c2w);
    const float nx = g.sign * g.n0.x, ny = g.sign * g.n0.y, nz = g.sign * g.n0.z;
    float *dst = recs + r * REC_FLOATS + 8 + channel;
    dst[0] = (nx * c2w[0] + ny * c2w[4]) + nz * c2w[8];
    dst[1] = (nx * c2w[1] + ny * c2w[5]) + nz * c2w[9];
    dst[2] = (nx * c2w[2] + ny * c2w[6]) + nz * c2w[10];
}
__global__ __launch_bounds__(256) void normals_bwd_qrows_kernel(int64_t cap_vis, const int32_t *__restrict__ vis_ids,
                                                                const int64_t *__restrict__ totals, const float *__restrict__ quats,
                                                                const float *__restrict__ scales, const float *__restrict__ means,
                                                                const float *__restrict__ c2w, const float *__restrict__ G,
                                                                int64_t row_stride, int col, float *__restrict__ qrows) {
    int64_t n_vis = totals ? *totals >> 32 : cap_vis;
    if (n_vis > cap_vis) n_vis = cap_vis;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_vis) return;
    const F3 v = F3{G[r * row_stride + col], G[r * row_stride + col + 1], G[r * row_stride + col + 2]};
    if (v.x == 0.f && v.y == 0.f && v.z == 0.f) {
        // nothing was composited from this Gaussian (occluded: most frustum-visible ones): the VJP of a zero cotangent is zero,
        // and its quaternion / scale / mean are not even gathered
        *reinterpret_cast<F4 *>(qrows + r * 4) = F4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int64_t i = vis_ids[r];
    const F4 q = *reinterpret_cast<const F4 *>(quats + i * 4);
    const F3 s = *reinterpret_cast<const F3 *>(scales + i * 3);
    const F3 m = *reinterpret_cast<const F3 *>(means + i * 3);
    const NormalGeom g = geometry(q, s, m, c2w);
    *reinterpret_cast<F4 *>(qrows + r * 4) = normal_vjp(q, g, v, c2w);
}
}  // namespace

extern "C" int mtgs_normals_fwd_rows(int64_t cap_vis, const int32_t *vis_ids, const int64_t *totals, const float *quats,
                                     const float *scales, const float *means, const float *c2w, float *recs, int channel,
                                     const uint8_t *row_flags,
                                     void *stream) {
    MTGS_REQUIRE(cap_vis >= 0 && channel >= 0 && channel + 3 <= REC_MAX_CHANNELS, MTGS_EINVAL, "mtgs_normals_fwd_rows: bad sizes");
    if (cap_vis == 0) return MTGS_OK;
    MTGS_REQUIRE(vis_ids && quats && scales && means && c2w && recs, MTGS_EINVAL, "mtgs_normals_fwd_rows: null pointer");
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(quats) & 15) == 0, MTGS_EINVAL, "mtgs_normals_fwd_rows: quats must be 16-byte aligned");
    normals_fwd_rows_kernel<<<(unsigned)ceil_div64(cap_vis, 256), 256, 0, (hipStream_t)stream>>>(cap_vis, vis_ids, totals, quats, scales,
                                                                                                 means, c2w, recs, channel, row_flags);
    MTGS_CHECK_LAUNCH("mtgs_normals_fwd_rows");
    return MTGS_OK;
}

extern "C" int mtgs_normals_bwd_qrows(int64_t cap_vis, const int32_t *vis_ids, const int64_t *totals, const float *quats,
                                      const float *scales, const float *means, const float *c2w, const float *grad_rows,
                                      int64_t row_stride, int col, float *quat_rows, void *stream) {
    MTGS_REQUIRE(cap_vis >= 0 && col >= 0 && row_stride >= col + 3, MTGS_EINVAL, "mtgs_normals_bwd_qrows: bad sizes");
    if (cap_vis == 0) return MTGS_OK;
    MTGS_REQUIRE(vis_ids && quats && scales && means && c2w && grad_rows && quat_rows, MTGS_EINVAL, "mtgs_normals_bwd_qrows: null pointer");
    MTGS_REQUIRE(((reinterpret_cast<uintptr_t>(quats) | reinterpret_cast<uintptr_t>(quat_rows)) & 15) == 0, MTGS_EINVAL,
                 "mtgs_normals_bwd_qrows: quats / quat_rows must be 16-byte aligned");
    normals_bwd_qrows_kernel<<<(unsigned)ceil_div64(cap_vis, 256), 256, 0, (hipStream_t)stream>>>(
        cap_vis, vis_ids, totals, quats, scales, means, c2w, grad_rows, row_stride, col, quat_rows);
    MTGS_CHECK_LAUNCH("mtgs_normals_bwd_qrows");
    return MTGS_OK;
}

extern "C" int mtgs_normals_bwd_rows(int64_t n_vis, const int32_t *vis_ids, const float *quats, const float *scales,
                                     const float *means, const float *c2w, const float *grad_rows, int64_t row_stride, int col,
                                     float *wire_rows, void *stream) {
    MTGS_REQUIRE(n_vis >= 0 && col >= 0 && row_stride >= col + 3, MTGS_EINVAL, "mtgs_normals_bwd_rows: bad sizes");
    if (n_vis == 0) return MTGS_OK;
    MTGS_REQUIRE(vis_ids && quats && scales && means && c2w && grad_rows && wire_rows, MTGS_EINVAL, "mtgs_normals_bwd_rows: null pointer");
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(quats) & 15) == 0, MTGS_EINVAL, "mtgs_normals_bwd_rows: quats must be 16-byte aligned");
    normals_bwd_rows_kernel<<<(unsigned)ceil_div64(n_vis, 256), 256, 0, (hipStream_t)stream>>>(n_vis, vis_ids, quats, scales, means,
                                                                                              c2w, grad_rows, row_stride, col, wire_rows);
    MTGS_CHECK_LAUNCH("mtgs_normals_bwd_rows");
    return MTGS_OK;
}

extern "C" int mtgs_normals_fwd(int64_t N, const float *quats, const float *scales, const float *means, const float *c2w,
                                const float *rgbs, float *out, int64_t out_stride, void *stream) {
    MTGS_REQUIRE(N >= 0 && out_stride >= (rgbs ? 6 : 3), MTGS_EINVAL, "mtgs_normals_fwd: bad sizes N=%lld out_stride=%lld",
                 (long long)N, (long long)out_stride);
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(quats && scales && means && c2w && out, MTGS_EINVAL, "mtgs_normals_fwd: null pointer");
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(quats) & 15) == 0, MTGS_EINVAL, "mtgs_normals_fwd: quats must be 16-byte aligned");
    normals_fwd_kernel<<<(unsigned)ceil_div64(N, 256), 256, 0, (hipStream_t)stream>>>(N, quats, scales, means, c2w, rgbs, out,
                                                                                      out_stride);
    MTGS_CHECK_LAUNCH("mtgs_normals_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_normals_bwd(int64_t N, const float *quats, const float *scales, const float *means, const float *c2w,
                                const float *v_normals, int64_t v_stride, float *g_quats, void *stream) {
    MTGS_REQUIRE(N >= 0 && v_stride >= 3, MTGS_EINVAL, "mtgs_normals_bwd: bad sizes");
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(quats && scales && means && c2w && v_normals && g_quats, MTGS_EINVAL, "mtgs_normals_bwd: null pointer");
    MTGS_REQUIRE(((reinterpret_cast<uintptr_t>(quats) | reinterpret_cast<uintptr_t>(g_quats)) & 15) == 0, MTGS_EINVAL,
                 "mtgs_normals_bwd: quats / g_quats must be 16-byte aligned");
    normals_bwd_kernel<<<(unsigned)ceil_div64(N, 256), 256, 0, (hipStream_t)stream>>>(N, quats, scales, means, c2w, v_normals,
                                                                                      v_stride, g_quats);
    MTGS_CHECK_LAUNCH("mtgs_normals_bwd");
    return MTGS_OK;
}
