// project.hip -- per-Gaussian 3D->2D projection (EWA splatting) forward and backward.
//
// Replaces gsplat 1.4.0 fully_fused_projection_fwd / fully_fused_projection_bwd (pinhole,
// packed=False), the first stage of gsplat.rendering.rasterization as called at
// /root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662.
//
// Roofline: HBM (pure streaming).  Algorithmic bytes per (camera, Gaussian):
//   fwd: 40 in (mean 12, quat 16, scale 12) + 4 (radius) out, + 28 (+4 compensation) for visible ones
//   bwd: 40 + 4 (radius) in; visible: + 16 (conic, comp) + 28 (+4) cotangents; 40 out
//
// This translation unit is compiled with -ffp-contract=off: the forward rounds after every
// operation in a fixed order, so radii / means2d / depths (the inputs of the integer tile-binning
// stage, whose results must be bit-exact) have exactly one IEEE-754 value per input.
#include "project_fwd_body.hpp"
#include "tile_rect.hpp"

namespace {

__global__ __launch_bounds__(PROJ_BLOCK) void project_fwd_kernel(
    int C, int64_t N, const float *__restrict__ means, const float *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ viewmats,
    const float *__restrict__ Ks, int W, int H, float eps2d, float near_plane, float far_plane,
    float radius_clip, const float *__restrict__ opacities, int32_t *__restrict__ radii,
    float *__restrict__ means2d, float *__restrict__ depths, float *__restrict__ conics,
    float *__restrict__ compensations, float *__restrict__ opac_eff, float tile_size, int tile_w, int tile_h,
    int32_t *__restrict__ tiles_per_gauss) {
    const int64_t idx = (int64_t)blockIdx.x * PROJ_BLOCK + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int c = (int)(idx / N);
    const int64_t n = idx - (int64_t)c * N;
    const Cam cam = load_cam(viewmats + c * 16, Ks + c * 9);
    const ProjOut o = project_pair(means, quats, scales, cam, n, W, H, eps2d, near_plane, far_plane, radius_clip);
    radii[idx] = o.radius;
    reinterpret_cast<float2 *>(means2d)[idx] = make_float2(o.mx, o.my);
    depths[idx] = o.depth;
    *reinterpret_cast<F3 *>(conics + idx * 3) = F3{o.ca, o.cb, o.cc};
    if (compensations) compensations[idx] = o.comp;
    // gsplat rendering.py: opacities.repeat(C, 1) [* compensations]
    if (opac_eff) opac_eff[idx] = o.radius > 0 ? (compensations ? opacities[n] * o.comp : opacities[n]) : 0.f;
    if (tiles_per_gauss) {  // gsplat isect_tiles, count pass (what mtgs_isect_count computes from the arrays above)
        int32_t cnt = 0;
        if (o.radius > 0) {
            const Rect q = tile_rect(o.mx, o.my, o.radius, tile_size, tile_w, tile_h);
            cnt = (q.x1 - q.x0) * (q.y1 - q.y0);
        }
        tiles_per_gauss[idx] = cnt;
    }
}

}  // namespace

extern "C" int mtgs_project_fwd(int C, int64_t N, const float *means, const float *quats,
                                const float *scales, const float *viewmats, const float *Ks,
                                int width, int height, float eps2d, float near_plane, float far_plane,
                                float radius_clip, const float *opacities, int32_t *radii,
                                float *means2d, float *depths, float *conics, float *compensations,
                                float *opac_eff, int tile_size, int tile_w, int tile_h, int32_t *tiles_per_gauss,
                                void *stream) {
    MTGS_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, MTGS_EINVAL,
                 "mtgs_project_fwd: bad sizes C=%d N=%lld W=%d H=%d", C, (long long)N, width, height);
    if ((int64_t)C * N == 0) return MTGS_OK;
    MTGS_REQUIRE(means && quats && scales && viewmats && Ks && radii && means2d && depths && conics,
                 MTGS_EINVAL, "mtgs_project_fwd: null pointer");
    MTGS_REQUIRE((int64_t)C * N < ((int64_t)1 << 31), MTGS_EINVAL,
                 "mtgs_project_fwd: C*N must fit int32 (flatten_ids are int32)");
    MTGS_REQUIRE(!opac_eff || opacities, MTGS_EINVAL, "mtgs_project_fwd: opac_eff requested without opacities");
    MTGS_REQUIRE(!tiles_per_gauss || (tile_size > 0 && tile_w > 0 && tile_h > 0), MTGS_EINVAL,
                 "mtgs_project_fwd: tiles_per_gauss requested without a tile grid");
    const unsigned grid = (unsigned)ceil_div64((int64_t)C * N, PROJ_BLOCK);
    project_fwd_kernel<<<grid, PROJ_BLOCK, 0, (hipStream_t)stream>>>(
        C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
        radius_clip, opacities, radii, means2d, depths, conics, compensations, opac_eff, (float)tile_size, tile_w, tile_h,
        tiles_per_gauss);
    MTGS_CHECK_LAUNCH("mtgs_project_fwd");
    return MTGS_OK;
}

