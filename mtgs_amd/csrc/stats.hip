// stats.hip -- densification statistics of one Gaussian node, one launch.
//
// Restates MTGSSceneModel.update_submodel_statistics + VanillaGaussianSplattingModel.after_train
// (/root/reference/mtgs/scene_model/mtgs_scene_graph.py:1157-1183,
//  /root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:448-474): per step and node
//     grads = (xys.absgrad[0, mask] * (W, H) * 0.5).norm(dim=-1)        (or xys.grad without absgrad)
//     vis   = radii[mask] > 0
//     vis_counts[vis] += 1;  xys_grad_norm[vis] += grads[vis];  max_2Dsize[vis] = max(max_2Dsize[vis], radii[vis])
// which PyTorch runs as ~12 masked gathers / scatters per node (a rigid-node scene has hundreds of nodes).  A node's
// Gaussians are a contiguous slice [start, start + n) of the collected arrays (torch.cat in get_gaussians, :408-461).
//
// Roofline: HBM; n * 12 B read + 24 B read-modify-write per visible Gaussian.
#include "common.hpp"

namespace {
__global__ __launch_bounds__(256) void densify_stats_kernel(int64_t n, const int32_t *__restrict__ radii,
                                                            const float *__restrict__ grad2d, float half_w, float half_h,
                                                            float *__restrict__ grad_norm, float *__restrict__ vis_counts,
                                                            float *__restrict__ max_2dsize) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t r = radii[i];
    if (r <= 0) return;
    const float2 g = reinterpret_cast<const float2 *>(grad2d)[i];
    const float gx = g.x * half_w, gy = g.y * half_h;
    grad_norm[i] += sqrtf(gx * gx + gy * gy);
    vis_counts[i] += 1.f;
    max_2dsize[i] = fmaxf(max_2dsize[i], (float)r);
}
// every node of the scene graph in one launch: workgroup b serves 256 Gaussians of the node with
// first_block <= b < next first_block (include/mtgs_rast.h: mtgs_stats_desc)
__global__ __launch_bounds__(256) void densify_stats_batch_kernel(const mtgs_stats_desc *__restrict__ table, int n_nodes,
                                                                  const int32_t *__restrict__ radii,
                                                                  const float *__restrict__ grad2d, float half_w, float half_h) {
    int lo = 0, hi = n_nodes - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const mtgs_stats_desc &d = table[__builtin_amdgcn_readfirstlane(lo)];
    const int64_t i = ((int64_t)blockIdx.x - d.first_block) * 256 + threadIdx.x;
    if (i >= d.n) return;
    const int32_t r = radii[d.start + i];
    if (r <= 0) return;
    const float2 g = reinterpret_cast<const float2 *>(grad2d)[d.start + i];
    const float gx = g.x * half_w, gy = g.y * half_h;
    d.xys_grad_norm[i] += sqrtf(gx * gx + gy * gy);
    d.vis_counts[i] += 1.f;
    d.max_2dsize[i] = fmaxf(d.max_2dsize[i], (float)r);
}
// The same update from the COMPACT gradient rows of the one-node rasterization (one 64-byte row per visible Gaussian:
// wrapper._FusedRasterization; what the data-parallel exchange keeps instead of a dense means2d gradient): thread per
// visible Gaussian, its node found by binary search over the nodes' first rows.  Only visible Gaussians are touched at
// all -- 300k rows instead of 2M radii at the headline size.
__global__ __launch_bounds__(256) void densify_stats_rows_kernel(int64_t n_vis, const int32_t *__restrict__ vis_ids,
                                                                 const float *__restrict__ rows, int64_t row_stride, int col,
                                                                 const int32_t *__restrict__ radii,
                                                                 const mtgs_stats_desc *__restrict__ table, int n_nodes,
                                                                 float half_w, float half_h, const int64_t *__restrict__ n_vis_dev) {
    if (n_vis_dev) {       // the count lives on the device (mtgs_front_fwd's packed totals); n_vis is the capacity of the rows
        const int64_t c = *n_vis_dev >> 32;
        if (c < n_vis) n_vis = c;
    }
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_vis) return;
    const int64_t id = vis_ids[r];
    int lo = 0, hi = n_nodes - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].start <= id) lo = mid; else hi = mid - 1;
    }
    const mtgs_stats_desc &d = table[lo];
    const int64_t i = id - d.start;
    if (i < 0 || i >= d.n) return;                      // a Gaussian of no listed node
    const float gx = rows[r * row_stride + col] * half_w, gy = rows[r * row_stride + col + 1] * half_h;
    d.xys_grad_norm[i] += sqrtf(gx * gx + gy * gy);
    d.vis_counts[i] += 1.f;
    d.max_2dsize[i] = fmaxf(d.max_2dsize[i], (float)radii[id]);
}
}  // namespace

/* table: n_nodes descriptors sorted by `start` (first_block unused); rows[n_vis, row_stride] floats, the 2-D gradient of
 * visible Gaussian r (flat index vis_ids[r]) in columns col, col + 1 (0: means2d.grad, 2: means2d.absgrad). */
extern "C" int mtgs_densify_stats_rows(int64_t n_vis, const int32_t *vis_ids, const float *rows, int64_t row_stride, int col,
                                       const int32_t *radii, int n_nodes, const mtgs_stats_desc *table, int width, int height,
                                       const int64_t *n_vis_dev, void *stream) {
    MTGS_REQUIRE(n_vis >= 0 && n_nodes >= 0 && row_stride >= 4 && (col == 0 || col == 2) && width > 0 && height > 0, MTGS_EINVAL,
                 "mtgs_densify_stats_rows: bad arguments");
    if (n_vis == 0 || n_nodes == 0) return MTGS_OK;
    MTGS_REQUIRE(vis_ids && rows && radii && table, MTGS_EINVAL, "mtgs_densify_stats_rows: null pointer");
    densify_stats_rows_kernel<<<(unsigned)ceil_div64(n_vis, 256), 256, 0, (hipStream_t)stream>>>(
        n_vis, vis_ids, rows, row_stride, col, radii, table, n_nodes, 0.5f * (float)width, 0.5f * (float)height, n_vis_dev);
    MTGS_CHECK_LAUNCH("mtgs_densify_stats_rows");
    return MTGS_OK;
}

extern "C" int mtgs_stats_desc_bytes(void) { return (int)sizeof(mtgs_stats_desc); }

extern "C" int mtgs_densify_stats_batch(int n_nodes, const mtgs_stats_desc *table, int64_t total_blocks, const int32_t *radii,
                                        const float *grad2d, int width, int height, void *stream) {
    MTGS_REQUIRE(n_nodes >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31) && width > 0 && height > 0, MTGS_EINVAL,
                 "mtgs_densify_stats_batch: bad sizes");
    if (n_nodes == 0 || total_blocks == 0) return MTGS_OK;
    MTGS_REQUIRE(table && radii && grad2d, MTGS_EINVAL, "mtgs_densify_stats_batch: null pointer");
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(grad2d) & 7) == 0, MTGS_EINVAL, "mtgs_densify_stats_batch: grad2d must be 8-byte aligned");
    densify_stats_batch_kernel<<<(unsigned)total_blocks, 256, 0, (hipStream_t)stream>>>(table, n_nodes, radii, grad2d,
                                                                                      0.5f * (float)width, 0.5f * (float)height);
    MTGS_CHECK_LAUNCH("mtgs_densify_stats_batch");
    return MTGS_OK;
}

extern "C" int mtgs_densify_stats(int64_t n, const int32_t *radii, const float *grad2d, int width, int height,
                                  float *xys_grad_norm, float *vis_counts, float *max_2dsize, void *stream) {
    MTGS_REQUIRE(n >= 0 && width > 0 && height > 0, MTGS_EINVAL, "mtgs_densify_stats: bad sizes");
    if (n == 0) return MTGS_OK;
    MTGS_REQUIRE(radii && grad2d && xys_grad_norm && vis_counts && max_2dsize, MTGS_EINVAL, "mtgs_densify_stats: null pointer");
    MTGS_REQUIRE((reinterpret_cast<uintptr_t>(grad2d) & 7) == 0, MTGS_EINVAL, "mtgs_densify_stats: grad2d must be 8-byte aligned");
    densify_stats_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, (hipStream_t)stream>>>(
        n, radii, grad2d, 0.5f * (float)width, 0.5f * (float)height, xys_grad_norm, vis_counts, max_2dsize);
    MTGS_CHECK_LAUNCH("mtgs_densify_stats");
    return MTGS_OK;
}
