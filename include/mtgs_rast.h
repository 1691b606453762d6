/*
 * mtgs_rast.h -- C ABI of libmtgs_rast.so, the MI355X (gfx950) Gaussian-splatting rasterizer
 * that sits behind gsplat 1.4.0's Python API for OpenDriveLab/MTGS.
 *
 * What each entry point replaces
 * ------------------------------
 * MTGS reaches the rasterizer through exactly two Python functions (reference @ 2025-09-12):
 *   gsplat.rendering.rasterization(...)          mtgs/scene_model/mtgs_scene_graph.py:20-23, :641-662
 *   gsplat.cuda._wrapper.spherical_harmonics()   mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:15-18, :317
 *                                                multi_color_gaussian_splatting.py:96, rigid_node.py:248, deformable_node.py:125
 * gsplat (pinned v1.4.0, requirements.txt:12) lowers those two functions onto a fixed set of
 * native operators (its `gsplat.cuda._wrapper._make_lazy_cuda_func` table: compute_sh_fwd/bwd,
 * fully_fused_projection_fwd/bwd, isect_tiles, isect_offset_encode, rasterize_to_pixels_fwd/bwd,
 * plus cub::DeviceRadixSort inside isect_tiles).  The functions below are that operator table,
 * one for one, as a plain C ABI: the binding a maintainer would write is in INTEGRATION.md.
 *
 * Conventions
 * -----------
 *  - Every pointer is a DEVICE pointer into caller-owned memory (PyTorch's allocator in the
 *    shipped host code).  The library never allocates, frees, retains a pointer or synchronises.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *    All work is enqueued on it; the call returns immediately.
 *  - Dense row-major fp32 / int32 / int64 arrays, shapes given per argument.  C = cameras,
 *    N = Gaussians, M = tile/Gaussian intersections, D = colour channels, K = SH bases.
 *  - Return value: 0 = MTGS_OK, otherwise an MTGS_E* code; mtgs_rast_last_error() returns a
 *    thread-local message for the last failing call on this thread.
 *  - Re-entrant: no global mutable state apart from the thread-local error string.
 *  - Optional pointers are marked "nullable".
 */
#ifndef MTGS_RAST_H
#define MTGS_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTGS_RAST_ABI_VERSION 28
/* Version of the HOT-PATH subset (mtgs_sh_*, mtgs_vis_color_*_dirs, mtgs_front_fwd, mtgs_bin3_build, mtgs_blend_*_packed, mtgs_project_bwd*): bumped only
 * when one of THOSE kernels or signatures changes, so that committed per-kernel counter files (profiles/rNN_pmc_step.json, keyed
 * on it) survive bumps of the optimizer / loss / node entry points.  mtgs_rast_hot_version() returns it. */
#define MTGS_RAST_HOT_ABI_VERSION 7
#define MTGS_VIS_COLOR_ROWS 64   /* visible Gaussians per workgroup of mtgs_vis_color_*: the granularity of dir_part */
#define MTGS_BIN3_TIGHT 1
#define MTGS_BIN3_FILL_TO_M 2
#define MTGS_BIN3_FILL_TO_CAP 4
#define MTGS_BIN3_PREZEROED 8
#define MTGS_BIN3_STATUS 16

enum {
    MTGS_OK = 0,
    MTGS_EINVAL = 1,      /* bad argument (null pointer, negative size, unsupported option) */
    MTGS_ELAUNCH = 2,     /* hipGetLastError() after a launch reported a failure */
    MTGS_EWORKSPACE = 3,  /* workspace too small */
    MTGS_EUNSUPPORTED = 4 /* valid in gsplat, not implemented here (named in the message) */
};

/* Fixed algorithm constants (gsplat 1.4.0 semantics; see oracle/gsplat_oracle.c for citations) */
#define MTGS_TILE_SIZE 16
#define MTGS_ALPHA_MAX 0.999f
/* exp(-sigma) is evaluated as exp2(-s2 * log2(e)/2); the compositing kernels fold the factor into the staged conic (blend.hip) */
#define MTGS_HALF_LOG2E 0.72134752044448170368f
#define MTGS_HALF_LOG2E_INV 1.38629436111989061883f
#define MTGS_ALPHA_MIN (1.0f / 255.0f)
#define MTGS_T_MIN 1e-4f
#define MTGS_MAX_SH_DEGREE 4
#define MTGS_MAX_CHANNELS 32 /* blended colour channels per launch (gsplat channel_chunk default) */

int mtgs_rast_version(void);
int mtgs_rast_hot_version(void);   /* MTGS_RAST_HOT_ABI_VERSION of the built library */
const char *mtgs_rast_last_error(void);

/* ---- spherical harmonics: gsplat compute_sh_fwd / compute_sh_bwd ------------------------------
 * dirs[n,3] (need not be unit), coeffs[n,K,3], masks[n] (nullable, 0 = skip: output/grads zero),
 * colors[n,3].  degree <= 4 and (degree+1)^2 <= K.  v_dirs nullable (MTGS passes detached dirs). */
int mtgs_sh_fwd(int64_t n, int K, int degree, const float *dirs, const float *coeffs,
                const uint8_t *masks, float *colors, void *stream);
int mtgs_sh_bwd(int64_t n, int K, int degree, const float *dirs, const float *coeffs,
                const uint8_t *masks, const float *v_colors, float *v_coeffs, float *v_dirs,
                void *stream);
/* The same v_coeffs for a cotangent that is zero for most Gaussians (the colours feed the rasterizer: only composited Gaussians
 * have one) -- v_coeffs[n,K,3] must be ZERO on entry (mtgs_fill_zero, which a caller can overlap with other work: it depends on
 * nothing -- mtgs_blend_bwd_packed(also_zero) writes it beside its own work); the call reads v_colors and writes the rows of the Gaussians whose cotangent is non-zero (and whose mask is set).
 * Same values as mtgs_sh_bwd (a zero cotangent gives a zero row there too).  No v_dirs. */
int mtgs_sh_bwd_rows(int64_t n, int K, int degree, const float *dirs, const uint8_t *masks, const float *v_colors,
                     float *v_coeffs, void *stream);
/* Round 6 (hot-path ABI v6): the caller's colour activation fused into the SH kernels.  MTGS writes `torch.clamp(rgbs + 0.5, 0.0, 1.0)`
 * behind every spherical_harmonics() call (vanilla_gaussian_splatting.py:318; multi_color / rigid / deformable nodes alike), gsplat's
 * own sh_degree path `clamp_min(colors + 0.5, 0)`: two elementwise passes forward and four backward over [n, 3].
 * mtgs_sh_fwd_act: colors = clamp(x + add, lo, hi) with x the SH output (has_add = 0: clamp(x); hi = +inf: clamp_min), the same fp32
 * operations in the same order as the torch expression, NaN propagating; pass[n] u8 (required) receives per Gaussian the bits of the
 * channels with lo <= x + add <= hi -- torch's clamp backward mask.  pass = NULL: mtgs_sh_fwd.
 * mtgs_sh_bwd_act / mtgs_sh_bwd_rows_act: v_colors is the cotangent of the ACTIVATED colours; channels whose bit is clear pass nothing.
 * The Python layer uses them when the caller's expression is exactly that one (mtgs_amd/wrapper.py::_LazySH). */
int mtgs_sh_fwd_act(int64_t n, int K, int degree, const float *dirs, const float *coeffs, const uint8_t *masks, float *colors,
                    int has_add, float add, float lo, float hi, uint8_t *pass, void *stream);
int mtgs_sh_bwd_act(int64_t n, int K, int degree, const float *dirs, const float *coeffs, const uint8_t *masks, const float *v_colors,
                    float *v_coeffs, float *v_dirs, const uint8_t *pass, void *stream);
int mtgs_sh_bwd_rows_act(int64_t n, int K, int degree, const float *dirs, const uint8_t *masks, const float *v_colors, float *v_coeffs,
                         const uint8_t *pass, void *stream);
/* bytes of zeros at p (4-byte aligned, a whole number of words), stream-ordered: 16-byte stores, the chip's fastest pure write. */
int mtgs_fill_zero(void *p, size_t bytes, void *stream);

/* ---- projection: gsplat fully_fused_projection_fwd / _bwd (pinhole, packed=False) -------------
 * means[N,3] quats[N,4] (wxyz, any norm) scales[N,3] viewmats[C,4,4] (world->cam) Ks[C,3,3].
 * out: radii[C,N] i32 (0 = culled), means2d[C,N,2], depths[C,N], conics[C,N,3] (a,b,c of the
 * inverse 2x2 covariance), compensations[C,N] (nullable; antialiased mode).  Culled rows of the
 * float outputs are written as zeros.
 * Fusion of gsplat's `opacities.repeat(C,1) * compensations` (rendering.py): with opacities[N]
 * (nullable) given, opac_eff[C,N] = opacities * compensations (or opacities in classic mode) is
 * written by the forward, and the backward takes v_opac_eff[C,N] (nullable) and writes
 * v_opacities[N] (nullable), folding v_opac_eff * opacities into the compensation VJP.
 * tiles_per_gauss[C,N] i32 (nullable, with tile_size / tile_w / tile_h): the count pass of isect_tiles
 * (mtgs_isect_count) written by the same kernel -- one launch and one pass over means2d / radii less per frame.
 * bwd: v_means[N,3] v_quats[N,4] v_scales[N,3] are OVERWRITTEN (summed over cameras);
 * v_viewmats[C,4,4] nullable, overwritten.
 * grad_row_strides (HOST pointer, nullable): row strides in floats of the incoming gradients
 * {v_means2d, v_depths, v_conics, v_compensations, v_opac_eff}; NULL = dense {2,1,3,1,1}.  Lets the
 * caller hand over views of the interleaved buffer mtgs_blend_bwd accumulated into (see there).
 * grad_row_index[C*N] i32 (nullable): row of the incoming gradients that belongs to (camera, Gaussian) pair
 * c*N+n (read for radii > 0 only); NULL = row c*N+n.  With mtgs_bin_compact's vis_rank the caller keeps ONE
 * compact 64-byte gradient row per VISIBLE Gaussian (19 MB instead of 128 MB at 2M Gaussians: no dense
 * zero-fill, atomics and re-reads stay in cache).
 * Dense by-products (all nullable): d_means2d[C,N,2] = the incoming v_means2d rows, d_means2d_abs[C,N,2] =
 * rows of x_means2d_abs, d_colors[C,N,x_channels] = rows of x_colors (same row index; x_row_strides = HOST
 * {abs, colors} row strides in floats, NULL = {2, x_channels}), zeros for culled pairs -- the gradients that
 * leave the rasterizer per Gaussian (retain_grad / absgrad / colours), expanded while their rows are read.
 * vis_ids[n_vis] i32 + vis_ws[n_vis*12] f32 scratch (both nullable): the visible Gaussians in increasing order
 * with grad_row_index[vis_ids[r]] == r (mtgs_bin_compact's vis_ids / vis_rank).  With them (and C == 1) the
 * VJP runs one thread per VISIBLE Gaussian and a streaming pass writes every dense output coalesced.
 * recs (nullable; hot ABI v5; compact path): mtgs_front_fwd's 64-byte records, indexed like the rows -- conic and blended opacity of a
 * visible Gaussian are then read from its record (contiguous) instead of being gathered from conics / opacities / compensations.
 * vm_partials (nullable; hot ABI v5; compact path, with v_viewmats): scratch of 12 * blocks floats (mtgs_project_bwd_blocks) -- every
 * workgroup of the per-visible pass leaves its 12 sums of the camera gradient there and the pass behind adds them in a fixed order into
 * v_viewmats (written in full): no launch to zero it, no same-address atomics (5 us at the headline workload), a deterministic sum.
 * n_vis_dev (nullable, device): mtgs_front_fwd's packed totals; the number of rows is then min(n_vis, *n_vis_dev >> 32)
 * and n_vis is only the capacity of the row buffers (graph mode: the host never learns the count).
 * Rows only: with v_means = v_quats = v_scales = v_opacities = NULL (compact path, no dense by-products) the streaming pass is
 * skipped and vis_ws[n_vis, 12] = [v_mean 3 | v_quat 4 | v_scale 3 | v_opacity 1 | pad] IS the result (mtgs_node_bwd_rows).
 * x_quat_rows[n_vis, 4] (nullable, compact path only, 16-byte aligned): quaternion gradients of the visible Gaussians that did
 * not come through the projection -- the camera-space normals' (mtgs_normals_bwd_qrows) -- added to v_quats; x_mean_rows[n_vis, 3]
 * (nullable, compact path only): likewise position gradients (the view directions of gsplat's sh_degree colours:
 * mtgs_vis_color_bwd's dir_rows), added to v_means.
 * raw_rows (nullable, compact path only; ABI v22): the base of the compact gradient rows v_means2d / v_conics / x_means2d_abs point
 * into (row stride = grad_row_strides[0]) when they hold mtgs_blend_bwd_packed's RAW MOMENT rows: every visible row's first eight
 * floats are converted IN PLACE to {v_xy 2, |v_xy| 2, v_conic 3, v_opacity_eff} before they are used (see mtgs_blend_bwd_packed). */
int mtgs_project_fwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                     const float *viewmats, const float *Ks, int width, int height, float eps2d,
                     float near_plane, float far_plane, float radius_clip, const float *opacities,
                     int32_t *radii, float *means2d, float *depths, float *conics,
                     float *compensations, float *opac_eff, int tile_size, int tile_w, int tile_h,
                     int32_t *tiles_per_gauss, void *stream);
int mtgs_project_bwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                     const float *viewmats, const float *Ks, int width, int height, float eps2d,
                     const int32_t *radii, const float *conics, const float *compensations,
                     const float *opacities, const float *v_means2d, const float *v_depths,
                     const float *v_conics, const float *v_compensations, const float *v_opac_eff,
                     float *v_means, float *v_quats, float *v_scales, float *v_viewmats,
                     float *v_opacities, const int64_t *grad_row_strides, const int32_t *grad_row_index,
                     const float *x_means2d_abs, const float *x_colors, int x_channels,
                     const int64_t *x_row_strides, float *d_means2d, float *d_means2d_abs, float *d_colors,
                     const int32_t *vis_ids, int64_t n_vis, float *vis_ws, const int64_t *n_vis_dev,
                     const float *x_quat_rows, const float *x_mean_rows, float *raw_rows, const float *recs, float *vm_partials,
                     void *stream);
/* mtgs_project_bwd_zeroed (ABI v28, hot ABI v7): the compact path of mtgs_project_bwd (C == 1, vis_ids, vis_ws, grad_row_index) for
 * dense outputs the CALLER ZEROED -- v_means, v_quats (16-byte aligned), v_scales, v_opacities, d_means2d, d_means2d_abs (8-byte
 * aligned), d_colors: typically as one region cleared beside the compositing backward's own work (mtgs_blend_bwd_packed(also_zero):
 * that kernel is VALU-bound, the bandwidth is idle).  The per-visible pass then writes the values of the Gaussians that HAVE a gradient
 * straight to their places (6 % of all Gaussians at the headline scene) and the streaming pass over all N -- every byte of every dense
 * output, 85 % of them zeros -- does not run; vis_ws is not written.  Same values as mtgs_project_bwd (a zero gradient is +0.0 here
 * where the VJP of an all-zero row may produce -0.0).  Same parameters. */
int mtgs_project_bwd_zeroed(int C, int64_t N, const float *means, const float *quats, const float *scales,
                     const float *viewmats, const float *Ks, int width, int height, float eps2d,
                     const int32_t *radii, const float *conics, const float *compensations,
                     const float *opacities, const float *v_means2d, const float *v_depths,
                     const float *v_conics, const float *v_compensations, const float *v_opac_eff,
                     float *v_means, float *v_quats, float *v_scales, float *v_viewmats,
                     float *v_opacities, const int64_t *grad_row_strides, const int32_t *grad_row_index,
                     const float *x_means2d_abs, const float *x_colors, int x_channels,
                     const int64_t *x_row_strides, float *d_means2d, float *d_means2d_abs, float *d_colors,
                     const int32_t *vis_ids, int64_t n_vis, float *vis_ws, const int64_t *n_vis_dev,
                     const float *x_quat_rows, const float *x_mean_rows, float *raw_rows, const float *recs, float *vm_partials,
                     void *stream);
/* workgroups of the compact path's per-visible pass for n_vis rows: vm_partials holds 12 floats for each */
int mtgs_project_bwd_blocks(int64_t n_vis, int64_t *blocks);

/* ---- tile intersection: gsplat isect_tiles (count pass / cumsum / emit pass) -------------------
 * mtgs_isect_count : tiles_per_gauss[C,N] i32 = #tiles of the clamped bounding square.
 * mtgs_isect_scan  : cum_tiles[C*N] i64 = inclusive prefix sum; total[1] i64 = M (device).
 *                    ws from mtgs_scan_workspace_bytes.
 * mtgs_isect_emit  : isect_ids[M] i64 = cam<<(32+tile_bits) | tile<<32 | bits(depth),
 *                    flatten_ids[M] i32 = c*N+n, row-major over each Gaussian's tile rectangle. */
int mtgs_isect_count(int C, int64_t N, const float *means2d, const int32_t *radii, int tile_size,
                     int tile_w, int tile_h, int32_t *tiles_per_gauss, void *stream);
int mtgs_scan_workspace_bytes(int64_t n, size_t *bytes);
int mtgs_isect_scan(int64_t n, const int32_t *tiles_per_gauss, int64_t *cum_tiles, int64_t *total,
                    void *ws, size_t ws_bytes, void *stream);
int mtgs_isect_emit(int C, int64_t N, const float *means2d, const int32_t *radii,
                    const float *depths, const int64_t *cum_tiles, int tile_size, int tile_w,
                    int tile_h, int64_t *isect_ids, int32_t *flatten_ids, void *stream);

/* ---- radix sort: the cub::DeviceRadixSort::SortPairs call inside gsplat isect_tiles ----------
 * Stable LSD sort of (i64 key, i32 value) on key bits [0, key_bits).  Output in keys_out/vals_out;
 * the inputs are only read (the ping-pong buffer lives in the workspace). */
int mtgs_sort_workspace_bytes(int64_t M, size_t *bytes);
int mtgs_sort_pairs(int64_t M, int key_bits, int64_t *keys_in, int32_t *vals_in, int64_t *keys_out,
                    int32_t *vals_out, void *ws, size_t ws_bytes, void *stream);

/* ---- depth-ordered binning: the fast path behind isect_tiles(sort=True) (mtgs_amd/csrc/bin.hip).
 * Produces the SAME isect_ids / flatten_ids as count+scan+emit+sort_pairs above (bit-identical),
 * by sorting the visible Gaussians by (camera, depth) first and then stably by tile only.
 *  mtgs_bin_compact  : vis_keys[<=C*N] i64 = cam<<32 | bits(depth), vis_ids[<=C*N] i32 = c*N+n, in
 *                      index order; totals[1] i64 (device) = n_vis<<32 | M.  ws: mtgs_scan_workspace_bytes(C*N).
 *                      vis_rank[C*N] i32 (nullable): position of every visible c*N+n in that list (other
 *                      entries are left untouched) -- the grad_row_index of the two backward kernels.
 *                      host_totals (nullable): PINNED HOST int64[2]; {totals, host_tag} is published (system scope)
 *                      as soon as the totals are known, one kernel before the call's work ends, so a host thread
 *                      polling host_totals[1] == host_tag reads n_vis and M without synchronising the stream.
 *  mtgs_bin_scan     : cum[n_vis] i64 = inclusive sum of tiles_per_gauss[ids_sorted[r]].
 *  mtgs_bin_emit     : tile_keys[M] u32 = cam*n_tiles + tile, gids[M] i32, in depth order.
 *  mtgs_sort_pairs_u32 : stable LSD sort of (u32 key, i32 value) on key bits [0, key_bits).
 *  mtgs_bin_sort_tiles : the same sort on the tile bits, with gsplat's isect_ids[M] i64 written by the
 *                      last pass directly (keys_scratch[M] u32 is a scratch buffer).
 *  mtgs_bin_finalize : isect_ids[M] i64 from already sorted (tile key, index) pairs and depths. */
int mtgs_bin_compact(int C, int64_t N, const int32_t *radii, const float *depths,
                     const int32_t *tiles_per_gauss, int64_t *vis_keys, int32_t *vis_ids,
                     int32_t *vis_rank, int64_t *totals, int64_t *host_totals, int64_t host_tag, void *ws,
                     size_t ws_bytes, void *stream);
int mtgs_bin_scan(int64_t n_vis, const int32_t *ids_sorted, const int32_t *tiles_per_gauss,
                  int64_t *cum, void *ws, size_t ws_bytes, void *stream);
int mtgs_bin_emit(int64_t M, int64_t n_vis, const int32_t *ids_sorted, int64_t N,
                  const float *means2d, const int32_t *radii, const int64_t *cum, int tile_size,
                  int tile_w, int tile_h, uint32_t *tile_keys, int32_t *gids, void *stream);
int mtgs_sort_u32_workspace_bytes(int64_t M, size_t *bytes);
int mtgs_sort_pairs_u32(int64_t M, int key_bits, uint32_t *keys_in, int32_t *vals_in,
                        uint32_t *keys_out, int32_t *vals_out, void *ws, size_t ws_bytes, void *stream);
int mtgs_bin_sort_tiles(int64_t M, int C, int tile_w, int tile_h, const uint32_t *tile_keys,
                        const int32_t *gids, const float *depths, uint32_t *keys_scratch,
                        int32_t *flatten_ids, int64_t *isect_ids, void *ws, size_t ws_bytes,
                        void *stream);
/* Everything after the host has read (n_vis, M) from mtgs_bin_compact's totals, in ONE call:
 * depth sort -> mtgs_bin_scan -> mtgs_bin_emit -> mtgs_bin_sort_tiles -> mtgs_isect_offsets (if offsets
 * non-null) -> mtgs_tile_schedule (if tile_order non-null).  ws: mtgs_bin_workspace_bytes. */
int mtgs_bin_workspace_bytes(int64_t n_vis, int64_t M, size_t *bytes);
int mtgs_bin_build(int C, int64_t N, int64_t n_vis, int64_t M, const float *means2d,
                   const int32_t *radii, const float *depths, const int32_t *tiles_per_gauss,
                   const int64_t *vis_keys, const int32_t *vis_ids, int tile_size, int tile_w, int tile_h,
                   int64_t *isect_ids, int32_t *flatten_ids, int32_t *offsets, int32_t *tile_order,
                   void *ws, size_t ws_bytes, void *stream);
int mtgs_bin_finalize(int64_t M, const uint32_t *tile_keys_sorted, const int32_t *flatten_ids,
                      const float *depths, int C, int tile_w, int tile_h, int64_t *isect_ids,
                      void *stream);

/* ---- gsplat isect_offset_encode: offsets[C,tile_h,tile_w] i32 = first sorted index per tile --- */
int mtgs_isect_offsets(int64_t M, const int64_t *isect_ids_sorted, int C, int tile_w, int tile_h,
                       int32_t *offsets, void *stream);

/* ---- compositing: gsplat rasterize_to_pixels_fwd / _bwd ---------------------------------------
 * means2d[C,N,2] conics[C,N,3] colors[C,N,D] opacities[C,N] backgrounds[C,D] (nullable).
 * out: render[C,H,W,DT] alphas[C,H,W] last_ids[C,H,W] i32 (index into the sorted list).
 * Two fusions of what gsplat's Python does around the operator (rendering.py, "RGB+D"/"RGB+ED"/"D"/"ED"):
 *   depths[C,N] (nullable): blended as one extra, LAST channel (DT = D + 1) instead of torch.cat;
 *                           D may then be 0 ("D"/"ED" modes, colors null).  backgrounds stays [C,D].
 *   ed_normalize          : the last channel is divided by clamp(alpha, min=1e-10) (expected depth).
 * With depths = NULL and ed_normalize = 0 this is exactly rasterize_to_pixels (DT = D).
 * bwd: v_means2d[C,N,2] v_conics[C,N,3] v_colors[C,N,D] v_depths[C,N] v_opacities[C,N] and
 * v_means2d_abs (nullable, absgrad) must be ZERO-FILLED by the caller; gradients are accumulated with
 * atomics.  render (the forward output) is only read when ed_normalize is set.
 * grad_row_strides (HOST pointer, nullable): row strides in floats of {v_means2d, v_means2d_abs, v_conics,
 * v_colors, v_depths, v_opacities}; NULL = the dense gsplat arrays {2,2,3,D,1,1}.  The memory-side fp32
 * atomics cost one request per 64-byte line an instruction touches, so callers should interleave the six
 * outputs in ONE buffer of 16-float rows (xy, |xy|, conic, opacity, colour.., depth) and pass views of it:
 * one line per (tile, Gaussian) instead of six (5x less atomic time on MI355X).
 * grad_row_index[C*N] i32 (nullable): gradient row of flatten id c*N+n (NULL = c*N+n), see mtgs_project_bwd.
 * tile_order[C*tile_h*tile_w] (nullable) is a permutation of the tile indices giving the order in
 * which tiles are dispatched (results do not depend on it); mtgs_tile_schedule fills it with the
 * tiles sorted by decreasing list length (no gsplat counterpart: a scheduling aid for the
 * one-wave-per-tile kernels). */
int mtgs_tile_schedule(int C, int tile_w, int tile_h, const int32_t *offsets, int64_t M,
                       int32_t *tile_order, void *stream);
int mtgs_blend_fwd(int C, int64_t N, int D, const float *means2d, const float *conics,
                   const float *colors, const float *opacities, const float *backgrounds,
                   const float *depths, int ed_normalize, int width, int height, int tile_size,
                   int tile_w, int tile_h, const int32_t *offsets, const int32_t *flatten_ids, int64_t M,
                   float *render, float *alphas, int32_t *last_ids, const int32_t *tile_order,
                   void *stream);
int mtgs_blend_bwd(int C, int64_t N, int D, const float *means2d, const float *conics,
                   const float *colors, const float *opacities, const float *backgrounds,
                   const float *depths, int ed_normalize, int width, int height, int tile_size,
                   int tile_w, int tile_h, const int32_t *offsets, const int32_t *flatten_ids, int64_t M,
                   const float *alphas, const int32_t *last_ids, const float *render,
                   const float *v_render, const float *v_alphas, float *v_means2d,
                   float *v_means2d_abs, float *v_conics, float *v_colors, float *v_depths,
                   float *v_opacities, const int64_t *grad_row_strides, const int32_t *grad_row_index,
                   const int32_t *tile_order, void *stream);

/* ---- input embedding of the deformation network of deformable object nodes (deformable_node.py:173-203,
 * utils.py:235-333; csrc/deform.hip): row n of out[N, ld] = [x, sin/cos(x 2^i) i < x_freqs | t, sin/cos(t 2^i) i < t_freqs |
 * cond[E]] with x = means[n] / height * 2.  The linear layers behind it are library GEMMs (mtgs_amd/deform.py). */
int mtgs_deform_embed(int64_t N, const float *means, float height, float t, const float *cond, int E, int x_freqs,
                      int t_freqs, float *out, int64_t ld, void *stream);

/* ---- Fourier-series features_dc of rigid object nodes (rigid_node.py:217-221; csrc/fourier.hip) -------------------------
 * dc[N,3] = sum_f features_dc[N,F,3] * w[F]  (F <= 32; w = IDFT(x) computed by the caller, utils.py:335-352).
 * bwd: v_features_dc[N,F,3] = w[f] * v_dc[N,3]; partial_w[ceil(3N/256), F] (nullable): per-block partial sums of v_w. */
int mtgs_fourier_dc_fwd(int64_t N, int F, const float *features_dc, const float *w, float *dc, void *stream);
int mtgs_fourier_dc_bwd(int64_t N, int F, const float *features_dc, const float *w, const float *v_dc,
                        float *v_features_dc, float *partial_w, void *stream);

/* ---- densification of a Gaussian node on the device (csrc/refine.hip) -------------------
 * Restates refinement_after / split_gaussians / dup_gaussians / cull_gaussians + the optimizer surgery of
 * mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:392-446, 476-699.
 * thresholds (HOST float[6]): densify_grad_thresh, densify_size_thresh, split_screen_size, cull_alpha_thresh,
 *   cull_scale_thresh, cull_screen_size.  options (HOST int[5]): n_split_samples (1..4), split by screen size
 *   (step < stop_screen_size_at), cull by world size (step > refine_every * reset_alpha_every), cull by screen size,
 *   clone_sample_means.  Samples: Philox4x32-10 keyed by (seed, step, Gaussian index, sample) -- identical on every rank.
 * mtgs_refine_classify: counts[(2 + n_split_samples), N] i32 (old row kept | child of sample s kept | duplicate kept),
 *   flags[N] u8.  The caller takes exclusive prefix sums of the count columns (pos, i64) and the first output row of each
 *   column's block (bases: old rows, then children sample-major, then duplicates -- the reference's order) and the total.
 * mtgs_refine_apply: src_index[n_out] i32 / kind[n_out] u8 of every output row and out_means / out_scales[n_out,3].
 * mtgs_refine_rows: dst[n_out,width] = src[src_index,:]; zero_new: rows of new Gaussians are zero (Adam moments). */
int mtgs_refine_classify(int64_t N, const float *means, const float *scales, const float *quats,
                         const float *opacities, const float *grad_norm, const float *vis_counts,
                         const float *max_2dsize, const float *thresholds, const int *options, uint64_t seed,
                         int64_t step, int32_t *counts, uint8_t *flags, void *stream);
int mtgs_refine_apply(int64_t N, int64_t n_out, const uint8_t *flags, const int64_t *pos, const int64_t *bases,
                      const float *means, const float *scales, const float *quats, const float *thresholds,
                      const int *options, uint64_t seed, int64_t step, int32_t *src_index, uint8_t *kind,
                      float *out_means, float *out_scales, void *stream);
int mtgs_refine_rows(int64_t n_out, int64_t width, const float *src, const int32_t *src_index, const uint8_t *kind,
                     int zero_new, float *dst, void *stream);

/* ---- fused rasterization front end + binning (ABI v7; what mtgs_amd.wrapper._FusedRasterization runs) ------------
 * The gsplat-shaped operators above stay; these entry points run the same stages of gsplat.rendering.rasterization
 * (mtgs/scene_model/mtgs_scene_graph.py:641-662) with fewer launches, one gather per intersection, and WITHOUT the host
 * knowing n_vis / M when it enqueues them.
 *
 * mtgs_front_fwd: mtgs_project_fwd (+ opacities * compensations + the tile count) fused with the compaction of the
 * visible (camera, Gaussian) pairs.  Dense outputs as mtgs_project_fwd.  Compact outputs, indexed by RANK (= number of
 * visible pairs with a smaller flat index c*N+n; entries with rank >= cap_vis are not written):
 *   recs[cap_vis,16] f32  packed record: x y | conic a b c | opacity_eff | s2max = 2 ln(255 opacity_eff) | radius (i32 bits)
 *                         | colours[D] (from colors[C*N,D]), depth (if with_depth), zeros   (D + with_depth <= 8)
 *   vis_ids[cap_vis] i32  flat index;  vis_keys[cap_vis] i64 = tile count << 40 | camera << 32 | bits(depth)
 *   vis_rank[C*N] i32     rank of every visible pair (dense; -1 for culled pairs: the row map of mtgs_adam_step)
 *   dp_words[ceil(C*N/64)] u64, dp_prefix[ceil(C*N/64)] u32 (both nullable): visibility bitmap and rank of the first
 *                         pair of each word (the map mtgs_dp_reduce reads); dp_count[1] i32 (nullable) = n_vis
 *   color_mode            0: colours as given; 1: the first three channels are SH output x, blended as
 *                         clamp(x + 0.5, 0, 1) (MTGS's colour activation, vanilla_gaussian_splatting.py:318);
 *                         2: the first three channels of the records are left open for mtgs_vis_color_fwd (colours of the
 *                         visible Gaussians only); `colors` then holds the OTHER D - 3 channels [C*N, D - 3] (nullable when D = 3)
 *   totals[1] i64 (device) = n_vis << 32 | M; host_totals (nullable): PINNED HOST int64[2], receives {totals, host_tag}
 *   as soon as the last block finishes (system-scope release store), so the host can poll instead of synchronising.
 *   M = 2^31 - 1 signals more than 2^31 - 2 intersections (or an internal failure): the frame cannot be rendered.
 * ws: mtgs_front_workspace_bytes(C*N), 256-byte aligned.
 *
 * mtgs_bin3_build: the tile binning of the frame WITHOUT a global sort -- (Gaussian, tile row) items grouped by row,
 * expanded into per-tile segments (unordered), every segment sorted on (depth bits, rank) by one workgroup in LDS:
 * offsets[C*th*tw + 1] (last entry = M) + tile_order (nullable), rank_ids[cap_M] (record / gradient-row index of
 * every intersection), flatten_ids[cap_M] (gsplat), isect_ids[cap_M] (gsplat; nullable) -- bit-identical to gsplat's
 * isect_tiles(sort=True) + isect_offset_encode.  Sizes come from `totals` on the device, clamped to (cap_vis, cap_M):
 * the caller sizes buffers and this call sizes grids for the capacities; if the true totals exceed them the outputs are
 * truncated (never out of bounds) and the caller repeats with larger ones.  Seven launches.
 * flags (ABI v25; was `tight`): MTGS_BIN3_TIGHT (1) -- TIGHT lists -- a (tile, Gaussian) pair of gsplat's 3-sigma square is listed only if the Gaussian's
 * {alpha >= 1/255} ellipse reaches a pixel centre of the tile (the exact ellipse / rectangle test, with a margin, that
 * mtgs_blend_*_packed apply when they stage a tile's candidates: 32 % of gsplat's pairs at the headline workload).  The
 * lists are sublists of gsplat's in the same order; every pair left out would be skipped pixel by pixel (gsplat's
 * `alpha < 1/255: continue`), so render, alphas and all gradients are those of the full lists, but the pair is never counted,
 * placed, sorted or gathered.  offsets[last] = the number of pairs listed (<= M of `totals`); tiles_per_gauss of
 * mtgs_front_fwd and M stay gsplat's.  Without the flag: gsplat's lists, bit-identical (the default of rasterization()).
 * MTGS_BIN3_PREZEROED (8): the workspace's control words are zero already (mtgs_front_fwd(also_zero)).
 * MTGS_BIN3_STATUS (16; hot ABI v5): `totals` points at FOUR words and the call writes words 1..3 = {n_vis, M, 1 if n_vis > cap_vis
 * or M > cap_M (the frame was truncated: repeat it with larger capacities) else 0} -- the device scalars a graph-captured caller
 * hands out as info["n_visible"] / ["n_intersections"] / ["overflow"] without launching anything to unpack word 0.
 * MTGS_BIN3_FILL_TO_M (2) / MTGS_BIN3_FILL_TO_CAP (4): the entries of flatten_ids / isect_ids behind the listed pairs -- up to
 * min(cap_M, M of `totals`) / up to cap_M -- are filled with sentinels (flatten_ids -1; isect_ids = last camera | last tile |
 * +inf depth bits): a caller that slices the tensors to gsplat's M (tight lists) or hands out capacity-sized tensors (graph
 * capture) never exposes uninitialised entries, and gsplat's "last range ends at numel" convention stays safe
 * (mtgs_blend_fwd stops at the first negative id; mtgs_isect_offsets of the padded ids puts the tail into the last tile).
 * Supported when mtgs_bin3_supported(C, tile_w, tile_h, cap_M) (C*tile_w*tile_h <= 32768 (3840x2160 has 32400 tiles), C*tile_h <= 4096, tile_w <= 4096,
 * cap_M < 2^30), else use mtgs_bin_build.  ws: mtgs_bin3_workspace_bytes, 256-byte aligned.
 *
 * mtgs_blend_fwd_packed / mtgs_blend_bwd_packed: mtgs_blend_fwd / _bwd reading the records through rank_ids.
 * grad_rows[n_vis, row_stride] f32 (zeroed by the caller): [xy 2 | |xy| 2 (absgrad) | conic 3 | opacity 1 | colours D |
 * depth 1 | pad], accumulated with atomics, one 64-byte line per visible pair for row_stride = 16. */
/* mtgs_front_fwd color_mode: 0 = `colors` as they are; 1 = clamp(colors[:3] + 0.5, 0, 1) (MTGS's activation on raw SH, data-parallel
 * exchange); 2 = channels 0..2 of the records are left open for mtgs_vis_color_fwd and `colors` holds the other D - 3; 3 = channels
 * 0..5 are left open (colours, then the camera-space normals: mtgs_normals_fwd_rows) and `colors` holds the other D - 6. */
int mtgs_front_workspace_bytes(int64_t total_pairs, size_t *bytes);
int mtgs_front_fwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                   const float *viewmats, const float *Ks, int width, int height, float eps2d, float near_plane,
                   float far_plane, float radius_clip, const float *opacities, const float *colors, int D,
                   int with_depth, int32_t *radii, float *means2d, float *depths, float *conics,
                   float *compensations, float *opac_eff, int tile_size, int tile_w, int tile_h,
                   int32_t *tiles_per_gauss, float *recs, int32_t *vis_ids, int64_t *vis_keys,
                   int32_t *vis_rank, int64_t cap_vis, uint64_t *dp_words, uint32_t *dp_prefix, int32_t *dp_count,
                   int color_mode, int64_t *totals, int64_t *host_totals, int64_t host_tag, void *also_zero,
                   size_t also_zero_bytes, void *ws, size_t ws_bytes, void *stream);
/* also_zero (nullable; hot ABI v4): a region of whole 4-byte words that mtgs_front_fwd's compaction kernel clears for the caller --
 * the control words at the start of mtgs_bin3_build's workspace (mtgs_bin3_control_bytes), so that the binning, called with
 * MTGS_BIN3_PREZEROED behind it on the same stream, needs no launch of its own for them. */
int mtgs_bin3_control_bytes(int C, int tile_w, int tile_h, size_t *bytes);
int mtgs_bin3_supported(int C, int tile_w, int tile_h, int64_t cap_M);
int mtgs_bin3_workspace_bytes(int C, int tile_w, int tile_h, int64_t cap_vis, int64_t cap_M, size_t *bytes);
int mtgs_bin3_build(int C, int64_t N, int tile_size, int tile_w, int tile_h, int64_t *totals,
                    int64_t cap_vis, int64_t cap_M, const float *recs, const int32_t *vis_ids,
                    const int64_t *vis_keys, int32_t *rank_ids, int32_t *flatten_ids, int64_t *isect_ids,
                    int32_t *offsets, int32_t *tile_order, int flags, void *ws, size_t ws_bytes, void *stream);
int mtgs_blend_fwd_packed(int C, int D, int with_depth, const float *recs, const float *backgrounds,
                          int ed_normalize, int width, int height, int tile_w, int tile_h, const int32_t *offsets,
                          const int32_t *rank_ids, float *render, float *alphas, int32_t *last_ids,
                          const int32_t *tile_order, void *also_zero, size_t also_zero_bytes, void *stream);
/* also_zero: as for mtgs_blend_bwd_packed below (a training forward clears what its backward pass will want zeroed: the gradient
 * rows of the compositing backward, the dL/dcoeffs of the spherical_harmonics() backwards). */
/* mtgs_blend_touch_packed (ABI v24): the forward's per-pixel DECISIONS without its colours -- touched[cap_vis] (uint8, cleared by the
 * call) gets 1 for every visible Gaussian (rank) that the frame composites FROM, i.e. that has a non-zero weight at some pixel: same
 * staging, validity test, alpha / T expressions and termination as mtgs_blend_fwd_packed, reading the geometry half (32 bytes) of the
 * records only (the colour channels may still be open).  Exactly those Gaussians receive a gradient in mtgs_blend_bwd_packed; in an
 * opaque scene they are a few percent of the frustum-visible ones.  What is per visible Gaussian behind the front end takes the flags
 * (row_flags of mtgs_vis_color_fwd / mtgs_normals_fwd_rows / mtgs_adam_group) and leaves the others alone -- their colour never
 * reaches a pixel (weight 0 wherever they are met), their gradient is zero. */
int mtgs_blend_touch_packed(int C, const float *recs, int width, int height, int tile_w, int tile_h, const int32_t *offsets,
                            const int32_t *rank_ids, const int32_t *tile_order, uint8_t *touched, int64_t cap_vis, void *stream);
/* mtgs_blend_bwd_packed (ABI v22): grad_rows[n_vis, row_stride] holds RAW MOMENT rows -- with h = vis * dL/dalpha per (pixel,
 * Gaussian) pair (gsplat's v_sigma = -opacity h): {sum h dx, sum h dy | k sum |h u|, k sum |h w| | sum h dx^2, sum h dx dy, sum h dy^2 |
 * sum h | colours D | depth}; u = a dx + b dy, w = b dx + c dy, k = MTGS_HALF_LOG2E (hot ABI v2: the kernels stage the conic with
 * that factor folded in, so that alpha = opacity * exp2(-s2) costs no multiply per pixel).  The conic map of the position gradient and the factor -opacity
 * are applied once per Gaussian by the row's consumer (mtgs_project_bwd(raw_rows), mtgs_project_bwd_rows(raw_rows = 1)), which
 * writes {v_xy, |v_xy|, v_conic, v_opacity_eff} back in place.  (mtgs_blend_bwd writes those directly.) */
int mtgs_blend_bwd_packed(int C, int D, int with_depth, const float *recs, const float *backgrounds,
                          int ed_normalize, int width, int height, int tile_w, int tile_h, const int32_t *offsets,
                          const int32_t *rank_ids, const float *alphas, const int32_t *last_ids,
                          const float *render, const float *v_render, const float *v_alphas, float *grad_rows,
                          int64_t row_stride, int absgrad, const int32_t *tile_order, void *also_zero, size_t also_zero_bytes,
                          void *stream);
/* also_zero (nullable; hot ABI v5): a 16-byte aligned region of whole 16-byte words that the kernel CLEARS for the caller while it
 * runs -- it is VALU-bound and leaves HBM ~85 % idle; every wave writes one slice of zeros when its tile is done.  The Python layer
 * passes the dL/dcoeffs buffers of the spherical_harmonics() backwards that follow in the same backward pass (384 MB at the headline
 * workload, 94 % of it zeros: mtgs_sh_bwd_rows then writes the other rows). */

/* ---- view-parallel data parallelism: sparse, factored gradient exchange (mtgs_amd/csrc/dp.hip) ------
 * No gsplat counterpart.  Rows are 16 floats: v_mean 3, v_quat 4, v_scale 3, v_opacity 1, v_rgb 3, spare,
 * Gaussian index (int bits).  mtgs_dp_pack compacts the rows of the Gaussians with radii > 0 (unordered)
 * and writes their number to count[1] (device); mtgs_dp_accumulate adds one sender's rows into the dense
 * gradient tensors, expanding v_rgb through the SH basis evaluated at normalize(mean - cam_pos) into
 * v_coeffs[N,K,3] (nullable). Indices within one call must be unique (they are: one camera per sender). */
int mtgs_dp_pack(int64_t N, const int32_t *radii, const float *v_means, const float *v_quats,
                 const float *v_scales, const float *v_opacities, const float *v_rgb, float *rows,
                 int64_t capacity, int64_t *count, void *stream);
int mtgs_dp_accumulate(int64_t n_rows, const float *rows, int64_t N, int K, int degree, const float *means,
                       const float *cam_pos, float *v_means, float *v_quats, float *v_scales,
                       float *v_opacities, float *v_coeffs, void *stream);


/* Ordered variant + one-pass reduction (what mtgs_amd.dist.SparseGradExchange uses for K <= 16, degree <= 3).
 * A sender's visibility map: words[ceil(N/64)] u64 (bit n%64 of word n/64 = radii[n] > 0) and prefix[ceil(N/64)] u32
 * (set bits in the words before).  mtgs_dp_pack_ordered builds both, count[1] i32 (device) = number of rows, and
 * writes the rows in INDEX order: row(n) = prefix[n/64] + popcount(words[n/64] below bit n%64), so a receiver finds
 * any Gaussian in any sender's rows from the sender's map alone (block_counts[ceil(N/1024)] u32 is scratch).
 * mtgs_dp_reduce: W senders' maps (sender r's words at (char*)words + r*map_stride_bytes, prefix likewise), rows
 * (sender r at rows + r*row_stride floats), cams[W,3]; ONE pass over the N Gaussians sums every sender's rows in
 * registers (v_rgb expanded through basis(normalize(mean - cam_r))) and WRITES v_means[N,3] v_quats[N,4]
 * v_scales[N,3] v_opacities[N] v_coeffs[N,K,3] (nullable) -- the sum over senders, zeros where no sender has a row.
 * [g_begin, g_end) (g_begin a multiple of 64; g_end < 0 = N): only that range of Gaussians is reduced and written, and
 * every sender's `rows` then starts at ITS first row of the range (row of Gaussian n = prefix[n/64] - prefix[g_begin/64]
 * + ...): the caller exchanges the rows in chunks of the index range and reduces a chunk while the next is on the wire. */
int mtgs_dp_pack_ordered(int64_t N, const int32_t *radii, const float *v_means, const float *v_quats,
                         const float *v_scales, const float *v_opacities, const float *v_rgb, uint64_t *words,
                         uint32_t *prefix, int32_t *count, uint32_t *block_counts, float *rows, int64_t capacity,
                         void *stream);
/* mtgs_dp_touched_pack (ABI v25): a sender's wire rows[n_rows,16] (index order, one per VISIBLE Gaussian, mtgs_project_bwd_rows) ->
 * only the rows that carry a gradient (any of the 14 gradient floats non-zero), compacted in the same order into
 * out_rows[capacity,16], with THEIR map in the format of a visibility map: out_words[ceil(N/64)] (bit n = Gaussian n has a row),
 * out_prefix[ceil(N/64)] (rows in front of the word), out_count[1] i32 and totals[1] i64 (count << 32).  The receivers' reduction
 * (mtgs_dp_reduce*) runs unchanged on the shorter rows.  scratch_words[ceil(N/64)] u64 and block_counts[ceil(ceil(N/64)/256)] are
 * scratch; rows beyond `capacity` are dropped (out_count tells: the caller falls back to the untruncated exchange). */
int mtgs_dp_touched_pack(int64_t n_rows, const float *rows, int64_t N, uint64_t *scratch_words, uint64_t *out_words,
                         uint32_t *out_prefix, int32_t *out_count, int64_t *totals, uint32_t *block_counts, float *out_rows,
                         int64_t capacity, void *stream);
/* mtgs_dp_touched_pack_chunks (ABI v27): the same compaction into CHUNKS of the Gaussian index range -- the rows of the Gaussians
 * in [begin[c], begin[c + 1]) go to rows[c] (at most cap[c] of them, same order), so that every chunk is its own all-gather message
 * and a receiver reduces chunk c (mtgs_dp_reduce_slices_cap over [begin[c], begin[c + 1]), rows = the gathered chunk) while chunk
 * c + 1 is on the wire.  begin[0] = 0, begin[n] = N, the others multiples of 64.  The map (out_words / out_prefix / out_count /
 * totals) is the WHOLE range's, as mtgs_dp_touched_pack writes it; overflow[1] i32 (device) is set to 1 when a chunk held more rows
 * than its capacity (they are dropped), 0 otherwise.  `chunks` is read on the host during the call. */
#define MTGS_DP_MAX_CHUNKS 16
typedef struct mtgs_dp_chunks {
    int n;
    int64_t begin[MTGS_DP_MAX_CHUNKS + 1];
    float *rows[MTGS_DP_MAX_CHUNKS];
    int64_t cap[MTGS_DP_MAX_CHUNKS];
} mtgs_dp_chunks;
int mtgs_dp_touched_pack_chunks(int64_t n_rows, const float *rows, int64_t N, uint64_t *scratch_words, uint64_t *out_words,
                                uint32_t *out_prefix, int32_t *out_count, int64_t *totals, uint32_t *block_counts,
                                const mtgs_dp_chunks *chunks, int32_t *overflow, void *stream);
int mtgs_dp_reduce(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words,
                   const uint32_t *prefix, int64_t map_stride_bytes, const float *rows, int64_t row_stride,
                   const float *cams, float *v_means, float *v_quats, float *v_scales, float *v_opacities,
                   float *v_coeffs, int64_t g_begin, int64_t g_end, void *stream);
/* mtgs_dp_reduce_slices: the same pass for PER-TRAVERSAL appearance parameters (MTGS's multi-colour nodes: ranks of one
 * step render cameras of different traversals, and a sender's colour gradient belongs to ITS traversal's coefficients).
 * coeff_mask: bit r = sender r contributes to v_coeffs; Gaussian n's coefficients start at v_coeffs + n * coeff_stride
 * floats (a slice of [N, T, K, 3]: coeff_stride = T K 3, v_coeffs pre-offset by t K 3); write_geometry = 0 skips the four
 * geometry gradients (summed over ALL senders, written by one of the T passes). */
int mtgs_dp_reduce_slices(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words,
                          const uint32_t *prefix, int64_t map_stride_bytes, const float *rows, int64_t row_stride,
                          const float *cams, float *v_means, float *v_quats, float *v_scales, float *v_opacities,
                          float *v_coeffs, int64_t g_begin, int64_t g_end, uint64_t coeff_mask, int write_geometry,
                          int64_t coeff_stride, void *stream);
/* mtgs_dp_reduce_slices_cap (ABI v27): mtgs_dp_reduce_slices with the number of ROWS a sender's block holds given apart from the
 * block stride: a block may carry more than rows (finish_touched: [rows | the sender's map]), and a sender that had more rows than
 * the agreed capacity must not have the words behind its rows summed as floats.  row_cap = 0: row_stride / 16 (the whole block).
 * write_geometry: bit 0 = this pass writes the four geometry gradients; bit 1 (ABI v27) = SPARSE write: the outputs were zeroed by the
 * caller (mtgs_blend_fwd_packed(also_zero) does it beside the compositing), so only the Gaussians some sender has a row for are
 * written and 32-Gaussian tiles without any are skipped -- same values as the dense write. */
int mtgs_dp_reduce_slices_cap(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words,
                              const uint32_t *prefix, int64_t map_stride_bytes, const float *rows, int64_t row_stride, int64_t row_cap,
                              const float *cams, float *v_means, float *v_quats, float *v_scales, float *v_opacities,
                              float *v_coeffs, int64_t g_begin, int64_t g_end, uint64_t coeff_mask, int write_geometry,
                              int64_t coeff_stride, void *stream);
/* ROWS out of the exchange (ABI v21): the receiver's sums as compact rows of the UNION of the senders' visible sets instead of
 * dense [N, .] tensors -- what lets the optimizer step the rows some camera of the step saw, and only those
 * (mtgs_amd.dist.SparseGradExchange.finish(rows=True) -> FusedAdam.set_row_gradient; reference: the per-traversal tensors of
 * multi_color_gaussian_splatting.py:53-80 under the data-parallel site custom_pipeline.py:87-89).
 * mtgs_dp_union: P sender subsets (masks[p], bit r = sender r) -> union_words[P, ceil(N/64)], union_prefix[P, ceil(N/64)] (the
 * map format of a sender: row(n) = prefix[n/64] + popcount(words[n/64] below bit n%64)), totals[p] = number of rows << 32 (the
 * packing of mtgs_front_fwd's totals); block_counts[P * ceil(ceil(N/64) / 256)] is scratch.
 * mtgs_dp_reduce_rows: the pass of mtgs_dp_reduce_slices over [g_begin, g_end) with row outputs, either half optional:
 *   geometry (sum over ALL senders): geo_rows[geo_cap, 16] = {v_mean 3, v_quat 4, v_scale 3, v_opacity 1, C0 * sum v_rgb 3 (the
 *     gradient of SH coefficient 0: features_dc), 0, Gaussian index}, geo_row_of[N] (row or -1), geo_ids[geo_cap] (row -> Gaussian),
 *     numbered by the union map (geo_words, geo_prefix) of the full subset;
 *   colour (sum over the senders of coeff_mask): coef_rows[coef_cap, coef_stride >= 3 K], coefficient k channel c at 3 k + c,
 *     coef_row_of[N], numbered by the union map of that subset.
 * The sums are formed in the order of mtgs_dp_reduce: a row equals the dense entry bit for bit.  Rows beyond a capacity are
 * dropped (their row_of entry still names them: mtgs_adam_step ignores rows >= n_rows). */
int mtgs_dp_union(int W, int64_t N, const uint64_t *words, int64_t map_stride_bytes, int P, const uint64_t *masks,
                  uint64_t *union_words, uint32_t *union_prefix, int64_t *totals, uint32_t *block_counts, void *stream);
int mtgs_dp_reduce_rows(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words, const uint32_t *prefix,
                        int64_t map_stride_bytes, const float *rows, int64_t row_stride, const float *cams, int64_t g_begin,
                        int64_t g_end, uint64_t coeff_mask, float *geo_rows, const uint64_t *geo_words, const uint32_t *geo_prefix,
                        int32_t *geo_row_of, int32_t *geo_ids, int64_t geo_cap, float *coef_rows, const uint64_t *coef_words,
                        const uint32_t *coef_prefix, int32_t *coef_row_of, int64_t coef_cap, int64_t coef_stride, void *stream);
/* mtgs_dp_reduce_rows_groups (ABI v24): mtgs_dp_reduce_rows for EVERY colour group of a step in one launch -- the geometry over all
 * senders (geo_rows nullable: colour groups only) and, for each of the n_groups entries of the DEVICE table `groups`, the colour rows of
 * the senders of its mask through its own union map.  A tile of 32 Gaussians is walked once (accumulators, means, the senders' spans,
 * the maps): at eight traversals 968 -> ~400 us at 2M Gaussians.  Bit-identical to one mtgs_dp_reduce_rows call per group. */
typedef struct mtgs_dp_group {
    uint64_t mask;               /* the senders (bit r = rank r) whose colour factors this group sums */
    float *coef_rows;            /* [coef_cap, coef_stride] */
    const uint64_t *coef_words;  /* union map of the group's senders (mtgs_dp_union) */
    const uint32_t *coef_prefix;
    int32_t *coef_row_of;        /* [N] */
    int64_t coef_cap;
} mtgs_dp_group;
int mtgs_dp_reduce_rows_groups(int W, int64_t N, int K, int degree, const float *means, const uint64_t *words, const uint32_t *prefix,
                               int64_t map_stride_bytes, const float *rows, int64_t row_stride, const float *cams, int64_t g_begin,
                               int64_t g_end, int n_groups, const mtgs_dp_group *groups, float *geo_rows, const uint64_t *geo_words,
                               const uint32_t *geo_prefix, int32_t *geo_row_of, int32_t *geo_ids, int64_t geo_cap, int64_t coef_stride,
                               void *stream);
/* Wire rows straight from the compositing backward's compact gradient rows (no dense tensor, no pack pass): the VJP of
 * the projection per VISIBLE Gaussian (vis_ids[n_vis], index order; C = 1) writes wire_rows[n_vis,16] =
 * {v_mean 3, v_quat 4, v_scale 3, v_opacity 1, v_rgb 3, 0, Gaussian index (int bits)}.  grad_rows[n_vis,row_stride] as mtgs_blend_bwd_packed
 * leaves them (the first three of D <= 7 colour channels; further channels are folded into the rows by their own VJP,
 * e.g. mtgs_normals_bwd_rows).  color_mode 1: colours were clamp(colors_pre + 0.5, 0, 1) (mtgs_front_fwd),
 * v_rgb is the gradient with respect to colors_pre[N,3].  color_mode 2 (hot ABI v7): the colours were written into the records by
 * mtgs_vis_color_fwd(_dirs) (use_sh = 1) and `colors_pre` points at its vis_mask -- uint8[n_vis], bit c = channel c passes its
 * cotangent -- instead of a dense [N,3] tensor.  v_viewmats[1,4,4] nullable, overwritten. */
int mtgs_project_bwd_rows(int64_t N, const float *means, const float *quats, const float *scales,
                          const float *viewmats, const float *Ks, int width, int height, float eps2d,
                          const float *conics, const float *compensations, const float *opacities,
                          float *grad_rows, int64_t row_stride, int D, int with_depth, const float *colors_pre,
                          int color_mode, const int32_t *vis_ids, int64_t n_vis, float *wire_rows, float *v_viewmats,
                          int raw_rows, void *stream);

/* ---- caller side of the path: fused per-node activations ----------------------------
 * One kernel per direction for what VanillaGaussianSplattingModel.get_gaussians does per step with a dozen PyTorch
 * launches (mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:299-341; multi_color_gaussian_splatting.py
 * :77-101 for the per-traversal coefficient sources):
 *   scales = exp(scales_raw), quats = quats_raw / |quats_raw|, opacities = sigmoid(opacities_raw),
 *   rgbs = clamp(SH(degree, normalize(means - cam_pos), [features_dc (+ features_dc_add) | features_rest]) + 0.5, 0, 1)
 *          (use_sh = 0: rgbs = sigmoid(features_dc (+ features_dc_add)), the sh_degree-0 model).
 * features_dc / features_dc_add (nullable) / features_rest are read in place: row_strides (HOST) = their row strides
 * in floats {>= 3, >= 3, >= K_rest*3}, so a per-traversal slice of [N,T,..] needs no gather and nothing is concatenated.
 * K_rest = K - 1 <= 15, degree <= 3.  clamp_mask[N] u8 (bit c: channel c passed the clamp) is saved for the backward.
 * bwd: gradients with respect to the RAW parameters; g_features_dc[N,3] (also the gradient of features_dc_add) and
 * g_features_rest[N,K_rest,3] are dense and fully written (zeros above the active degree).  means / cam_pos get no
 * gradient: MTGS detaches the view directions (vanilla_gaussian_splatting.py:314).
 * n_traversals > 0 (multi-colour nodes): the coefficients were slice `traversal` of per-traversal parameters
 * [N,T,K_rest,3] / [N,T,3]; the backward then writes the gradients of the FULL tensors -- g_features_rest[N,T,K_rest,3]
 * and g_features_dc_add[N,T,3] (nullable): values in slice `traversal`, zeros elsewhere -- in the same pass, instead
 * of the zero-fill + strided slice copy autograd performs for an indexed view.
 * Rigid nodes (rigid_node.py:205-216, shipped configs have fourier_features_dim = None): pose[7] (DEVICE, nullable) =
 * instance quaternion wxyz | translation; global mean = quat_to_rotmat(q) m + t (utils.py:14-41, q is NOT normalised
 * there), global quaternion = quat_mult(q, q_local / |q_local|) (utils.py:60-70), view directions from the global
 * mean.  means_out[N,3] (nullable) receives the (global) means.  bwd: v_means[N,3] (nullable) is the gradient that
 * reached the global means, g_means[N,3] (nullable) = R^T v_means (v_means without pose), g_pose[7] (nullable) is
 * ACCUMULATED with atomics (zero it first): d quaternion from both the rotation and the quaternion product, d t. */
int mtgs_node_fwd(int64_t N, int K_rest, int degree, int use_sh, const float *means, const float *scales_raw,
                  const float *quats_raw, const float *opacities_raw, const float *features_dc,
                  const float *features_dc_add, const float *features_rest, const int64_t *row_strides,
                  const float *cam_pos, float *scales, float *quats, float *opacities, float *rgbs,
                  uint8_t *clamp_mask, const float *pose, float *means_out, void *stream);
int mtgs_node_bwd(int64_t N, int K_rest, int degree, int use_sh, const float *means, const float *quats_raw,
                  const float *cam_pos, const float *scales, const float *opacities, const float *rgbs,
                  const uint8_t *clamp_mask, const float *v_scales, const float *v_quats, const float *v_opacities,
                  const float *v_rgbs, float *g_scales_raw, float *g_quats_raw, float *g_opacities_raw,
                  float *g_features_dc, float *g_features_rest, float *g_features_dc_add, int n_traversals,
                  int traversal, const float *pose, const float *v_means, float *g_means, float *g_pose,
                  void *stream);

/* ---- all nodes of a scene in ONE launch -------------------------------------------------------------------------------
 * MTGSSceneModel.get_gaussians (mtgs_scene_graph.py:408-461) loops over the nodes of the scene graph -- background, road,
 * and one rigid node per object instance present in the frame (rigid_node.py:259-261): tens to hundreds of nodes of a few
 * thousand Gaussians each, i.e. hundreds of launches of microseconds of work.  The batched entry points take a TABLE of
 * descriptors in device memory (one per node, in output order) and run mtgs_node_fwd / mtgs_node_bwd for every node in a
 * single launch: workgroup b serves 256 Gaussians of the node with first_block <= b < next first_block.
 * All pointers are device pointers with the meaning of the same-named arguments of mtgs_node_fwd / mtgs_node_bwd; the
 * output pointers are the node's slices of the collected tensors.  The forward ignores the v_* / g_* fields; the backward
 * reads features_* only through k_rest / use_sh (gradient rows are dense: [n,3], [n,k_rest,3] or, with n_traversals > 0,
 * [n,T,k_rest,3] / [n,T,3]). */
typedef struct mtgs_node_desc {
    int64_t n;                 /* Gaussians of this node */
    int64_t first_block;       /* index of the node's first workgroup: sum over earlier nodes of ceil(n / 256) */
    int64_t start;             /* offset of the node's first Gaussian in the collected tensors (model_id) */
    const float *means, *scales_raw, *quats_raw, *opacities_raw;
    const float *features_dc, *features_dc_add, *features_rest;
    int64_t dc_stride, dc_add_stride, rest_stride;          /* row strides in floats */
    const float *pose;         /* [4] instance quaternion wxyz of a rigid node, or NULL (static node) */
    const float *pose_trans;   /* [3] instance translation */
    int32_t k_rest, use_sh, n_traversals, traversal;
    int32_t pose_normalize;    /* 1: `pose` is a raw row of the per-frame parameters instance_quats[frame]: the kernel normalises
                                * it as RigidSubModel.get_object_pose does (rigid_node.py:142) */
    int32_t skip_colors;       /* 1: geometry only (no coefficient reads, no rgbs / clamp_mask writes): the colours of the VISIBLE
                                * Gaussians are evaluated by mtgs_vis_color_fwd from this same table */
    float *scales, *quats, *opacities, *rgbs;               /* forward outputs = activations saved for the backward */
    uint8_t *clamp_mask;
    float *means_out;          /* global means (written when non-NULL) */
    const float *v_scales, *v_quats, *v_opacities, *v_rgbs, *v_means;
    float *g_scales_raw, *g_quats_raw, *g_opacities_raw, *g_features_dc, *g_features_rest, *g_features_dc_add, *g_means,
        *g_pose;               /* g_pose[7]: zeroed accumulator (atomics), gradient of (normalised quaternion | translation) */
    float *g_pose_quat_row, *g_pose_trans_row;   /* pose_normalize: rows [4] / [3] of the gradients of the per-frame parameters,
                                                  * written after the launch from g_pose (nullable) */
    const int32_t *frame_dev;  /* (ABI v24, nullable) the frame of this step in DEVICE memory: pose / pose_trans / g_pose_*_row then name
                                * row 0 of the node's per-frame tables and the kernels add the frame when they run (one captured
                                * iteration for every frame) */
} mtgs_node_desc;
int mtgs_node_desc_bytes(void);   /* sizeof(mtgs_node_desc): bindings check their layout against it */
/* total_blocks = sum over nodes of ceil(n / 256); `degree` = sh_degree_to_use of the step (all nodes), or -1 for
 * mtgs_node_fwd_batch when EVERY descriptor has skip_colors = 1: the lean geometry-only kernel (same results);
 * model_id[sum n] (nullable) receives the node index of every collected Gaussian (mtgs_scene_graph.py:449-455). */
int mtgs_node_fwd_batch(int n_nodes, const mtgs_node_desc *table, int64_t total_blocks, int degree, const float *cam_pos,
                        int64_t *model_id, void *stream);
int mtgs_node_bwd_batch(int n_nodes, const mtgs_node_desc *table, int64_t total_blocks, int degree, const float *cam_pos,
                        void *stream);
/* The geometry half of the node backward for the VISIBLE Gaussians: ws_rows[cap_vis, ws_stride >= 12] (mtgs_project_bwd, rows only: 12;
 * mtgs_dp_reduce_rows' geometry rows: 16) -> 
 * param_rows[cap_vis, 12] = [means 3 | scales 3 | quats 4 | opacities 1 | pad], the gradients with respect to the RAW parameters of
 * Gaussian vis_ids[r] (exp / normalise / sigmoid VJPs through the descriptors' scales / quats_raw / opacities), r < min(cap_vis,
 * *totals >> 32) (totals nullable).  Static nodes only (pose = NULL).  The rows go to mtgs_adam_step through a row map: the dense
 * geometry gradients (zeros for ~85 % of the Gaussians) are neither written nor read. */
int mtgs_node_bwd_rows(int n_nodes, const mtgs_node_desc *table, const int32_t *vis_ids, const int64_t *totals, int64_t cap_vis,
                       const float *ws_rows, int ws_stride, float *param_rows, void *stream);

/* ---- camera-space normals of the Gaussians (predict_normals, the shipped MTGS.py config: 3 more blended channels) ------
 * MTGSSceneModel._get_gaussian_camera_space_normals (mtgs_scene_graph.py:526-545), ~25 PyTorch launches per direction and a
 * host synchronisation (boolean-mask write), in one kernel per direction:
 *   k = argmin(scales[n]) (first of equal minima); col = column k of quat_to_rotmat(quats[n]) (utils.py:14-41, wxyz, the
 *   quaternion as given); n0 = col / max(|col|, 1e-12); s = -1 if <n0, normalize(c2w[:,3] - means[n])> < 0 else +1;
 *   normals[n] = (s n0) @ c2w[:3,:3].
 * quats[N,4] (16-byte aligned), scales[N,3] (activated or raw: only their order matters), means[N,3], c2w = DEVICE
 * pointer to camera_to_worlds[3,4] row-major (no host copy of the camera).  out: row n starts at out + n * out_stride;
 * rgbs (nullable [N,3]): when given the row is [rgbs[n] | normals[n]] (out_stride >= 6: the torch.cat of :638 is gone),
 * otherwise the normals alone (out_stride >= 3).
 * bwd: v_normals = cotangent of the NORMAL columns (pointer to the first of them, row stride v_stride floats);
 * g_quats[N,4] is fully written.  scales, means and the camera get no gradient (argmin; .detach(); data). */
int mtgs_normals_fwd(int64_t N, const float *quats, const float *scales, const float *means, const float *c2w,
                     const float *rgbs, float *out, int64_t out_stride, void *stream);
int mtgs_normals_bwd(int64_t N, const float *quats, const float *scales, const float *means, const float *c2w,
                     const float *v_normals, int64_t v_stride, float *g_quats, void *stream);
/* The normal channels' gradient of the VISIBLE Gaussians folded into the quaternion gradient of their data-parallel wire
 * rows (wire_rows[r, 16], Gaussian vis_ids[r]): v_normal = grad_rows[r, col .. col + 2] (the compositing backward's compact
 * rows).  The camera-space normal depends on the sender's camera, so this happens before the rows are exchanged. */
int mtgs_normals_bwd_rows(int64_t n_vis, const int32_t *vis_ids, const float *quats, const float *scales, const float *means,
                          const float *c2w, const float *grad_rows, int64_t row_stride, int col, float *wire_rows,
                          void *stream);
/* Visibility first (one process): the normals of the VISIBLE Gaussians only.  fwd_rows: the camera-space normal of Gaussian
 * vis_ids[r] goes into channels `channel` .. + 2 of its record (mtgs_front_fwd color_mode 3 left them open), r < min(cap_vis,
 * *totals >> 32) (totals nullable: cap_vis rows).  bwd_qrows: quat_rows[r, 4] = the quaternion gradient from v_normal =
 * grad_rows[r, col .. col + 2], handed to mtgs_project_bwd as x_quat_rows.  No [N, 3] normal tensor and no dense gradient of it. */
/* row_flags (ABI v24, nullable): rows with row_flags[r] == 0 (nothing is composited from them: mtgs_blend_touch_packed) get zeros and
 * nothing of theirs is gathered */
int mtgs_normals_fwd_rows(int64_t cap_vis, const int32_t *vis_ids, const int64_t *totals, const float *quats, const float *scales,
                          const float *means, const float *c2w, float *recs, int channel, const uint8_t *row_flags, void *stream);
int mtgs_normals_bwd_qrows(int64_t cap_vis, const int32_t *vis_ids, const int64_t *totals, const float *quats, const float *scales,
                           const float *means, const float *c2w, const float *grad_rows, int64_t row_stride, int col,
                           float *quat_rows, void *stream);

/* ---- densification statistics of one node in one launch ------------------------------
 * mtgs_scene_graph.py:1157-1183 + vanilla_gaussian_splatting.py:448-474: for the n Gaussians of a node (a contiguous
 * slice of the collected arrays; pass pointers to the slice) with radii > 0:
 *   xys_grad_norm += |grad2d * (width, height) * 0.5|,  vis_counts += 1,  max_2dsize = max(max_2dsize, radii).
 * radii[n] i32, grad2d[n,2] (means2d.absgrad or .grad of the rasterization), the three statistics [n] f32 in place. */
int mtgs_densify_stats(int64_t n, const int32_t *radii, const float *grad2d, int width, int height,
                       float *xys_grad_norm, float *vis_counts, float *max_2dsize, void *stream);

/* The same for every node of the scene graph in ONE launch (update_submodel_statistics loops over the nodes,
 * mtgs_scene_graph.py:1157-1183): a table of descriptors in device memory, in any order; node i owns rows
 * [start, start + n) of radii / grad2d (the collected arrays) and its own three statistics arrays [n].
 * total_blocks = sum over nodes of ceil(n / 256); first_block = the running sum. */
typedef struct mtgs_stats_desc {
    int64_t n, first_block, start;
    float *xys_grad_norm, *vis_counts, *max_2dsize;
} mtgs_stats_desc;
/* The same statistics from the compact gradient rows of the one-node rasterization (csrc/stats.hip): table sorted by
 * `start`; the 2-D gradient of visible Gaussian r = rows[r * row_stride + col .. col + 1], col 0 (grad) or 2 (absgrad).
 * n_vis_dev (nullable, device): mtgs_front_fwd's packed totals -- the number of rows is then min(n_vis, *n_vis_dev >> 32)
 * (graph mode: n_vis is the capacity of the row buffers). */
int mtgs_densify_stats_rows(int64_t n_vis, const int32_t *vis_ids, const float *rows, int64_t row_stride, int col,
                            const int32_t *radii, int n_nodes, const mtgs_stats_desc *table, int width, int height,
                            const int64_t *n_vis_dev, void *stream);
int mtgs_stats_desc_bytes(void);
int mtgs_densify_stats_batch(int n_nodes, const mtgs_stats_desc *table, int64_t total_blocks, const int32_t *radii,
                             const float *grad2d, int width, int height, void *stream);

/* ---- patch-wise depth NCC of the loss head (ncc_loss_lambda = 0.1 in config/MTGS.py; patches 32 x 32, stride 16) ---------
 * calculate_depth_ncc_loss (mtgs/utils/geometric_loss.py:322-348): F.unfold of pred / gt / mask with padding k/2, patches
 * whose mask is all ones (boolean index: host sync, sorting backward), per patch pc = p - mean(p), gc = g - mean(g),
 * ncc = mean(pc gc) / (sqrt(mean(pc^2) + 1e-8) sqrt(mean(gc^2) + 1e-8));  out[0] = 1 - mean over valid patches (NaN when
 * none), out[1] = their number.  pred, gt [H,W] f32, mask [H,W] u8 (nullable).  patch_stats[mtgs_ncc_patches * 6] is
 * written by the forward and read by the backward; bwd: v_pred[H,W] fully written (gt gets no gradient). */
int mtgs_ncc_patches(int width, int height, int patch_size, int stride, int64_t *n);
int mtgs_ncc_fwd(int width, int height, int patch_size, int stride, const float *pred, const float *gt, const uint8_t *mask,
                 float *patch_stats, float *out, void *stream);
int mtgs_ncc_bwd(int width, int height, int patch_size, int stride, const float *pred, const float *gt,
                 const float *patch_stats, const float *v_out, const float *fwd_out, float *v_pred, void *stream);

/* ---- total variation of the normal image (use_normal_tv_loss = True in config/MTGS.py:114) ---------------------------------
 * TVLoss.forward (mtgs/utils/geometric_loss.py:293-303) for one image[H,W,channels]:
 *   out[0] = mean |image[:, :-1] - image[:, 1:]| + mean |image[:-1] - image[1:]|      (NaN inputs propagate)
 * partials: mtgs_tv_workspace_floats; bwd: v_image fully written (v_out = DEVICE pointer to the scalar cotangent). */
int mtgs_tv_workspace_floats(int width, int height, int channels, size_t *n);
int mtgs_tv_fwd(int width, int height, int channels, const float *image, float *partials, float *out, void *stream);
int mtgs_tv_bwd(int width, int height, int channels, const float *image, const float *v_out, float *v_image, void *stream);

/* ---- out-of-box regulariser of the rigid object nodes (config/MTGS.py:117 oob_lambda = 1.0) -----------------------------
 * mtgs_scene_graph.py:949-967 loops over the rigid models with a full-size `model_id == id` comparison, boolean-mask
 * gathers and two host synchronisations per node; here every node of the frame in one pass.  For the nodes with at least
 * one visible Gaussian (radii[start .. start + n) > 0): oob = any(|means_local| > limit), limit = instance_size / 2 +
 * tolerance;  out[0] = sum over the oob Gaussians of -log(1 - sigmoid(opacities) + 1e-6) / their number (0 if none),
 * out[1] = that number.  flags[n_nodes] (int32 scratch, receives the per-node visibility), partials[2 * total_blocks].
 * bwd: g_opacities[n] of every node fully written (zeros outside the oob set). */
typedef struct mtgs_oob_desc {
    int64_t n, first_block, start;      /* Gaussians of the node; its first 256-thread block; its offset in radii */
    const float *means;                 /* [n,3] LOCAL means of the rigid node */
    const float *opacities;             /* [n] logits */
    float *g_opacities;                 /* [n] (backward) */
    float limit[3];
    float reserved;
} mtgs_oob_desc;
int mtgs_oob_desc_bytes(void);
int mtgs_oob_fwd(int n_nodes, const mtgs_oob_desc *table, int64_t total_blocks, const int32_t *radii, int32_t *flags,
                 float *partials, float *out, void *stream);
int mtgs_oob_bwd(int n_nodes, const mtgs_oob_desc *table, int64_t total_blocks, const int32_t *flags, const float *v_out,
                 const float *fwd_out, void *stream);

/* ---- the output head between the rasterizer and the losses ------------------------
 * mtgs_scene_graph.py:672-690 + LearnableExposureRGBModel.forward (module/appearance.py:73-87), one kernel per direction:
 *   rgb[H,W,3]            = clamp(render[..., :3] + (1 - alpha) * background, 0, 1)
 *   rgb_appearance[H,W,3] = clamp(rgb @ E[:3,:3] + E[:3,3], 0, 1)         E = exposure[3,4] of the camera (nullable: skipped)
 *   depth[H,W]            = alpha > 0 ? render[..., depth_channel] : depth_max[0]     (depth_channel < 0: skipped;
 *                           depth_max = DEVICE scalar, the detached maximum of the depth channel)
 *   normal[H,W,3]         = (n / |n| + 1) / 2,  n = render[..., normal_channel : normal_channel + 3]  (< 0: skipped; no eps)
 * render[H,W,channels], alpha[H,W]; background[3], exposure[12] are DEVICE pointers.
 * bwd: any of the four cotangents may be NULL; v_render[H,W,channels] (every channel written) and v_alpha[H,W] are the
 * complete gradients of this head; v_background[3], v_exposure[12] (nullable) are reduced through `partials`
 * (mtgs_head_workspace_floats) in a fixed order. */
int mtgs_head_fwd(int width, int height, int channels, int depth_channel, int normal_channel, const float *render,
                  const float *alpha, const float *background, const float *exposure, const float *depth_max, float *rgb,
                  float *rgb_appearance, float *depth, float *normal, void *stream);
int mtgs_head_workspace_floats(int width, int height, size_t *n);
int mtgs_head_bwd(int width, int height, int channels, int depth_channel, int normal_channel, const float *render,
                  const float *alpha, const float *background, const float *exposure, const float *v_rgb,
                  const float *v_rgb_appearance, const float *v_depth, const float *v_normal, float *v_render, float *v_alpha,
                  float *v_background, float *v_exposure, float *partials, void *stream);

/* ---- masked SSIM of the loss head, fused ---------------------------------------------
 * mtgs.utils.ssim.MaskedSSIM(data_range=1.0, size_average=True, channel=3)(gt, pred, mask)  (mtgs/utils/ssim.py:57-190,
 * mtgs_scene_graph.py:322, :831-841).  gt, pred: [H,W,3] f32 (the rasterizer's layout: no NCHW copies); mask[H,W] u8
 * (nullable = all ones), cropped by the window margin as the reference does.  11 taps, `win_sigma` (reference: 1.5),
 * K1 = 0.01, K2 = 0.03.  out[0] = masked mean of the (H-10) x (W-10) x 3 SSIM map, out[1] = number of masked elements.
 * gmaps[(H-10)*(W-10)*9] (nullable: forward only) receives mask * d ssim / d {mu_pred, E[pred^2], E[gt pred]} per
 * pixel and channel for the backward; partials: mtgs_ssim_workspace_floats.  Sums in a fixed order (deterministic).
 * bwd: v_pred[H,W,3] = v_out[0] * d out[0] / d pred  (v_out, fwd_out = DEVICE pointers; gt gets no gradient). */
int mtgs_ssim_workspace_floats(int width, int height, size_t *n);
int mtgs_ssim_fwd(int width, int height, const float *gt, const float *pred, const uint8_t *mask, float win_sigma,
                  float data_range, float K1, float K2, float *gmaps, float *partials, float *out, void *stream);
int mtgs_ssim_bwd(int width, int height, const float *gt, const float *pred, const float *gmaps, float win_sigma,
                  const float *v_out, const float *fwd_out, float *v_pred, void *stream);
/* Masked L1 of the same loss head: torch.abs(gt - pred)[mask].mean() -- the RGB term (mtgs_scene_graph.py:823, 3 channels),
 * the depth terms (:881-883, 1 channel) and the normal term (:934, 3 channels); gt, pred [H,W,channels], 1 <= channels <= 8;
 * out[0] = mean over the masked pixels x channels, out[1] = their number; v_pred = v_out[0] * mask * sign(pred - gt) /
 * out[1].  Replaces a nonzero + gather forward and a sorted index_put backward. */
int mtgs_l1_workspace_floats(int width, int height, size_t *n);
int mtgs_l1_fwd(int width, int height, int channels, const float *gt, const float *pred, const uint8_t *mask,
                float *partials, float *out, void *stream);
int mtgs_l1_bwd(int width, int height, int channels, const float *gt, const float *pred, const uint8_t *mask,
                const float *v_out, const float *fwd_out, float *v_pred, void *stream);
/* The lidar depth term (mtgs_scene_graph.py:849-856, 875-879, DepthLossType.InverseL1) in one launch per direction:
 *   m = (gt > lo) & (gt < hi) & mask;   out[0] = |1 / (gt + eps) - 1 / (pred + eps)|[m].mean()  (0 when m is empty, :857),
 *   out[1] = count(m);  mask_out[H*W] (nullable) = m, which the depth NCC term reuses (:891).  gt, pred [H,W] f32; mask
 * [H,W] u8, nullable.  PyTorch forms it with a dozen elementwise launches per direction.  partials: mtgs_l1_workspace_floats. */
int mtgs_inv_depth_l1_fwd(int width, int height, const float *gt_depth, const float *pred_depth, const uint8_t *mask,
                          float lo, float hi, float eps, uint8_t *mask_out, float *partials, float *out, void *stream);
int mtgs_inv_depth_l1_bwd(int width, int height, const float *gt_depth, const float *pred_depth, const uint8_t *mask,
                          float lo, float hi, float eps, const float *v_out, const float *fwd_out, float *v_pred, void *stream);

/* Camera position of ONE view matrix [A t; 0 1] (row-major 4x4, device): cam_pos = -A^-1 t = torch.inverse(viewmat)[:3, 3], which
 * gsplat's sh_degree path forms for the view directions (rendering.py) -- there an LU factorisation with its backward, ~25 launches;
 * here one each way.  bwd: v_viewmat[4,4] (fully written) from v_cam_pos[3]. */
int mtgs_campos_fwd(const float *viewmat, float *cam_pos, void *stream);
int mtgs_campos_bwd(const float *viewmat, const float *v_cam_pos, float *v_viewmat, void *stream);

/* The sum of the loss dictionary (mtgs_scene_graph.py:823-945 scales every term by its lambda and adds the normal term only when
 * it is finite, :939; the trainer adds the values up): out[0] = constant + sum_i weights[i] * terms[i] over the n <= 16 DEVICE
 * scalars terms[i], a term whose bit is set in guard_mask being dropped when it is NaN / inf; kept[0] = bit mask of the terms that
 * counted.  bwd: v_terms[i] = kept_i ? weights[i] * v_out[0] : 0.  weights: HOST array.  One launch each way instead of ~30
 * one-element PyTorch kernels. */
int mtgs_loss_combine_fwd(int n, const float *terms, const float *weights, unsigned guard_mask, float constant, float *out,
                          uint32_t *kept, void *stream);
int mtgs_loss_combine_bwd(int n, const float *v_out, const uint32_t *kept, const float *weights, float *v_terms, void *stream);

/* ---- colours of the VISIBLE Gaussians only (visibility-first node path) ------------------------------------------------
 * MTGS evaluates SH + clamp for every Gaussian of every node each step (vanilla_gaussian_splatting.py:309-322,
 * multi_color_gaussian_splatting.py:77-101) although a camera sees ~15 % of a road block; gsplat's own sh_degree path masks
 * SH with radii > 0 (rendering.py).  Here the node kernels run geometry-only (mtgs_node_desc.skip_colors) and, after
 * mtgs_front_fwd (called with color_offset = 3: the first three channels of the records are left open), this call fills
 * them for the n_vis = totals >> 32 visible Gaussians: Gaussian g = vis_ids[r] belongs to the node with start <= g < start + n
 * (table sorted by start: the collected order); colour = clamp(SH_degree(normalize(means[g] - cam_pos), coefficients) + 0.5,
 * 0, 1) for use_sh = 1, sigmoid(features_dc [+ dc_add]) for use_sh = 0, coefficients read in place through the node's row
 * strides.  vis_mask[r] = the clamp's pass-through bits.  C = 1.  `means` = the collected (global) means [N,3].
 * bwd: v_rgb = grad_rows[r * row_stride + col .. + 2] (the compositing backward's compact rows) -> feat_rows[r, 48] =
 * d L / d coefficient k, channel c at [3 k + c] (k = 0: features_dc (and the adapter), k >= 1: features_rest[k - 1]; zeros above the
 * degree in use): the gradient of the VISIBLE rows only -- consumed as rows by mtgs_adam_step, never expanded. */
/* coef_rows (nullable): the coefficients of visible Gaussian r are read from the compact row
 * coef_rows[r * coef_stride ..] = [dc 3 | dc_add 3 | rest 3 k_rest] (coef_stride >= 51 floats) instead of the nodes' tensors --
 * what mtgs_adam_step's MTGS_ADAM_ROWS_PEEK groups leave (row-lazy optimizer: up-to-date values without touching the parameters). */
int mtgs_vis_color_fwd(int n_nodes, const mtgs_node_desc *table, int degree, const float *cam_pos, const float *means,
                       const int32_t *vis_ids, const int64_t *totals, int64_t cap_vis, float *recs, uint8_t *vis_mask,
                       const float *coef_rows, int64_t coef_stride, const uint8_t *row_flags, void *stream);
int mtgs_vis_color_bwd(int n_nodes, const mtgs_node_desc *table, int degree, const float *cam_pos, const float *means,
                       const int32_t *vis_ids, const int64_t *totals, int64_t cap_vis, const float *grad_rows,
                       int64_t row_stride, int col, const float *recs, const uint8_t *vis_mask, float *feat_rows,
                       float *dir_rows, float *dir_part, float *dense_rows, void *stream);
/* mtgs_vis_color_fwd_dirs / mtgs_vis_color_bwd_dirs (ABI v28, hot ABI v7): the same with the view directions GIVEN -- dirs[N, 3]
 * (nullable: then as above), indexed by the Gaussian like `means`, normalised by the kernel exactly as mtgs_sh_fwd does -- instead of
 * means[g] - cam_pos (which may then be NULL).  This is MTGS's own call style, gsplat.cuda._wrapper.spherical_harmonics(n, viewdirs,
 * colors) followed by torch.clamp(. + 0.5, 0, 1) (vanilla_gaussian_splatting.py:313-318), evaluated for the Gaussians the projection
 * found visible when the caller hands the result to rasterization() (mtgs_amd/wrapper.py, _LazySH): bit-identical colours to
 * mtgs_sh_fwd_act with K = 16 for those Gaussians, no [N, 3] colour tensor, no read of the other ~85 % of the coefficient rows.
 * dir_rows / dir_part with dirs: d L / d dirs[vis_ids[r]] of the visible rows (through the kernel's normalisation), for callers whose
 * directions carry a gradient (MTGS with its camera optimizer: viewdirs = means.detach() - camera_to_worlds[..., :3, 3]). */
int mtgs_vis_color_fwd_dirs(int n_nodes, const mtgs_node_desc *table, int degree, const float *cam_pos, const float *means,
                            const int32_t *vis_ids, const int64_t *totals, int64_t cap_vis, float *recs, uint8_t *vis_mask,
                            const float *coef_rows, int64_t coef_stride, const uint8_t *row_flags, const float *dirs, void *stream);
int mtgs_vis_color_bwd_dirs(int n_nodes, const mtgs_node_desc *table, int degree, const float *cam_pos, const float *means,
                            const int32_t *vis_ids, const int64_t *totals, int64_t cap_vis, const float *grad_rows,
                            int64_t row_stride, int col, const float *recs, const uint8_t *vis_mask, float *feat_rows,
                            float *dir_rows, float *dir_part, float *dense_rows, const float *dirs, void *stream);
/* dense_rows (nullable; ABI v26; since v28 any number of nodes whose coefficient tensors are plain [n_i, 16, 3] rows: node i's gradient
 * is the slice [start_i, start_i + n_i) of the buffer): a ZEROED [N, 16, 3] coefficient gradient -- the rows of the visible Gaussians whose colour
 * cotangent is not zero are written straight into it (row = vis_ids[r]) and feat_rows may be NULL: gsplat's `sh_degree` call style
 * without the [n_vis, 48] intermediate and the dense expansion pass behind it. */
/* use_sh = 4 in a descriptor: gsplat's own sh_degree path -- clamp_min(SH + 0.5, 0) (gsplat/rendering.py), and dir_rows
 * (nullable [cap_vis, 3]) receives d L / d (means - camera position) of the visible rows (gsplat's view directions are
 * differentiable; MTGS detaches them) and dir_part[ceil(cap_vis / MTGS_VIS_COLOR_ROWS), 3] (zeroed by the caller; 64 rows per workgroup
 * since ABI v28, 128 before) their per-workgroup sums
 * (minus their total is the camera position's gradient).  mtgs_rows_expand: dense expansion of gradient rows for autograd callers:
 * out[n, c] = row_of[n] >= 0 ? rows[row_of[n] * row_stride + c] : 0 for c < width. */
int mtgs_rows_expand(int64_t N, int width, const int32_t *row_of, const float *rows, int64_t row_stride, float *out,
                     void *stream);

/* ---- SURVEY.md section 8f, rank 2 (second half): the optimizer step of every Gaussian parameter group in ONE launch ----
 * Reference: one torch.optim.Adam per parameter group with one tensor each (mtgs/scene_model/custom_trainer.py:115-136;
 * groups, learning rates and eps = 1e-15 in mtgs/config/MTGS.py:121-181); the densification moves the moments with their
 * rows (vanilla_gaussian_splatting.py:392-446 -> mtgs_refine_rows).  Arithmetic = torch.optim.Adam (amsgrad = False,
 * maximize = False) in fp32:  g = grad_scale * g + weight_decay * p;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;
 * p -= step_size * m / (sqrt(v) / bc2_sqrt + eps)  with step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) computed
 * by the caller in double for the step t being taken and passed as `hyper[.][4]` = {step_size f32, bc2_sqrt f32, t i32,
 * pending i32} per group in DEVICE memory (row hyper_index of the group; t and pending are read by the row-lazy groups
 * below only) -- apart from the table, so that a step captured in a HIP graph is advanced by one small copy per replay
 * (learning-rate schedules and the bias corrections change every step; the table does not).
 * A table of descriptors in DEVICE memory (8-byte aligned), one per tensor; workgroup b works on group i with
 * first_block[i] <= b < first_block[i + 1], first_block = running sum of ceil(n / mtgs_adam_block_elems()).
 * Gradient source of a group: `g` (dense, n floats), or `rows` + `row_of` -- element e belongs to item i = e / width
 * (a Gaussian: width floats of this tensor), its gradient is rows[row_of[i] * row_stride + row_col + e % width], or 0 when
 * row_of[i] < 0 (a Gaussian the frame did not see: exact zero-gradient update, no dense gradient tensor) -- or neither
 * (all-zero gradient).  vec_ok = 1: p, m, v (and g) are 16-byte aligned. */
typedef struct mtgs_adam_group {
    float *p, *m, *v;           /* parameter, exp_avg, exp_avg_sq: n floats each, updated in place */
    const float *g;             /* dense gradient or NULL */
    const float *rows;          /* compact gradient rows or NULL; streaming groups (DENSE / SLICE) with BOTH take g + the row */
    const int32_t *row_of;      /* [n / width] row of every item, < 0: none */
    const float *catchup;       /* catchup_k > 0: {step_size, bc2_sqrt} of the catchup_k steps to apply, oldest first (DEVICE) */
    int32_t *last;              /* row-lazy groups: [n, T] step up to which slice (i, t) of item i is current */
    float *hist;                /* row-lazy groups: {step_size, bc2_sqrt} of step j at hist[2 j] (DEVICE; the step launch appends) */
    const int32_t *row_ids;     /* row-lazy groups, LIST form (nullable): the frame's visible Gaussians in increasing order,
                                 * row_ids[r] = global index of rank r (mtgs_front_fwd's vis_ids); this tensor's items are the
                                 * indices item_start .. item_start + n - 1.  NULL: SCAN form, the rows are found through row_of */
    const int64_t *row_count_dev; /* LIST: number of valid ranks = min(n_rows, *row_count_dev >> 32) (mtgs_front_fwd's totals; nullable) */
    float *caught;              /* ROWS_PEEK: destination, ROWS_STEP: source (nullable) -- the up-to-date parameter rows of the
                                 * frame, row r = row_of[i] at caught[r * caught_stride + caught_col ..] (n_rows rows) */
    const int32_t *sub_index_dev; /* (ABI v24, nullable) the slice in DEVICE memory: read when the kernel runs, overrides sub_index -- ONE
                                 * captured step / peek serves every traversal of a per-traversal tensor (the caller rewrites the
                                 * word in front of a replay) */
    const uint8_t *row_flags;   /* (ABI v24, nullable) LIST groups, ROWS_PEEK / ROWS_STEP: row r (rank) is worked on only if row_flags[r] != 0 --
                                 * the Gaussians the frame composites FROM (mtgs_blend_touch_packed); the others have a zero gradient
                                 * and nobody reads their peeked row: they are skipped before anything of theirs is requested */
    int64_t n, first_block;
    int64_t row_stride;         /* floats between rows */
    int64_t caught_stride;      /* floats between rows of `caught` */
    int64_t item_start;         /* LIST: global index of this tensor's item 0 */
    int64_t n_rows;             /* rows in `rows` (and `caught`): a row_of entry >= n_rows (a capacity overflow of the frame that produced the
                                 * map, graph mode) is treated as "no row" instead of being read */
    int32_t width, row_col;
    int32_t vec_ok;
    int32_t sub_width, sub_index;   /* sub_width > 0: an item is `width / sub_width` slices of sub_width floats (a per-traversal
                                     * tensor [N, T, ...]); only slice sub_index takes the row's gradient
                                     * rows[.., row_col + e % sub_width], the other slices get zero */
    int32_t mode;                   /* MTGS_ADAM_DENSE: every element, as above.
                                     * MTGS_ADAM_SLICE (with sub_width): the group IS slice sub_index of a [N, T, ...] tensor -- n = N * sub_width
                                     * virtual elements, element e lives at p[(e / sub_width) * width + sub_index * sub_width + e % sub_width];
                                     * the other slices are neither read nor written (exact lazy Adam, below).
                                     * MTGS_ADAM_ROWS_CATCHUP / _ROWS_STEP / _ROWS_FLUSH: row-lazy groups, below (n = number of ITEMS) */
    int32_t catchup_k;              /* SLICE: > 0: no gradient step -- apply catchup_k ZERO-gradient steps with the scalars in `catchup`.
                                     * ROWS_CATCHUP / ROWS_FLUSH: the step to catch up to, or < 0: hyper.t - hyper.pending */
    int32_t hyper_index;            /* row of `hyper` that belongs to this group */
    int32_t caught_col;
    int32_t rank_start, rank_count; /* LIST: the ranks of this tensor's items -- WRITTEN by mtgs_adam_step (flags bit 1), which also
                                     * rewrites first_block of the LIST groups from the counts */
    int32_t zero_probe;             /* ROWS_STEP (ABI v23): low 16 bits c + 1 > 0: a row whose three floats rows[r * row_stride + c ..] are
                                     * all zero has an all-zero gradient (mtgs_vis_color_bwd's rows: the coefficient-0 gradient C0 * v_rgb
                                     * is zero iff every coefficient's is) and is NOT stepped while it is fewer than K = (zero_probe >> 16)
                                     * (0: 6) steps behind -- it stays lazy, exactly like a Gaussian the frame did not see (the
                                     * zero-gradient update is what a later catch-up replays: bit-identical); K bounds the history a
                                     * forward's peek replays for a row that is visible in every frame and never receives a gradient.
                                     * Most frustum-visible Gaussians are occluded and receive no gradient at all (measured 91 ... 98 % at
                                     * 960x540 in the harness scenes).  0: every row with row_of >= 0 is stepped */
    float one_minus_beta1, beta2, one_minus_beta2;   /* 1 - beta rounded from double by the caller (1 - 0.999f is 5e-5 off) */
    float eps, weight_decay, grad_scale;
} mtgs_adam_group;
/* Exact lazy Adam for per-traversal tensors [N, T, ...] (MTGS's features_rest / features_adapters,
 * multi_color_gaussian_splatting.py:53-71): a step renders ONE traversal; the other traversals' slices get the zero gradient,
 * i.e. their moments only decay and p moves along exp_avg -- a recurrence in per-step scalars the host knows.  Such a slice
 * can be left untouched (slice_only groups step the rendered slice alone) and CAUGHT UP before its traversal is rendered
 * again: catchup_k zero-gradient steps per element in registers, the same operations in the same order as stepping every
 * time (bit-identical), at 24 B per element of ONE slice instead of 24 B x T per step. */
enum { MTGS_ADAM_DENSE = 0, MTGS_ADAM_SLICE = 1, MTGS_ADAM_ROWS_CATCHUP = 2, MTGS_ADAM_ROWS_STEP = 3, MTGS_ADAM_ROWS_FLUSH = 4,
       MTGS_ADAM_ROWS_PEEK = 5 };
/* Row-lazy Adam (exact) for tensors of which a frame READS ONLY THE VISIBLE ROWS -- with visibility-first colours
 * (mtgs_vis_color_fwd) the SH coefficients: 48 of a Gaussian's 59 floats, of which a camera needs ~15 %.  A Gaussian the
 * frame does not see gets the zero gradient: its moments decay and p drifts along exp_avg, which nothing reads until the
 * Gaussian is seen again.  So `last[i, t]` records the step up to which slice t of item i is current, `hist` the per-step
 * scalars, and
 *   ROWS_CATCHUP (the forward, after mtgs_front_fwd and before mtgs_vis_color_fwd): for every item with row_of[i] >= 0 apply
 *                the zero-gradient steps last + 1 .. target in registers (the same operations in the same order as stepping
 *                every time: BIT-IDENTICAL), target = catchup_k or, < 0, hyper.t - hyper.pending = the steps already taken;
 *   ROWS_STEP    (the optimizer step, same launch as the other groups): for the items with row_of[i] >= 0 catch up to t - 1 if
 *                needed, apply step t with the gradient rows[row_of[i] * row_stride + row_col + c], set last = t; one
 *                thread appends hist[t] and clears hyper.pending (which the caller's copy of this step's scalars had set:
 *                a forward captured in the same HIP graph as its step reads t - 1 as the steps already taken);
 *   ROWS_FLUSH   every item behind the target is caught up (checkpoints, refinement, anything else that reads the tensor);
 *   ROWS_PEEK    ROWS_CATCHUP without side effects: the caught-up PARAMETER rows go to the compact buffer `caught` (row
 *                row_of[i]; p, m, v and `last` stay as they are) -- mtgs_vis_color_fwd reads its coefficients from there,
 *                coalesced, and a forward no longer changes optimizer state.  A ROWS_STEP group that is handed the same
 *                frame's `caught` rows takes p from them and replays only the moment recurrences of the missed steps
 *                (m, v do not depend on p when weight_decay = 0; otherwise `caught` is ignored).  A PEEK group with m = v =
 *                last = NULL is a plain row copy (tensors that are not row-lazy).
 * SCAN form (row_ids = NULL): a workgroup scans mtgs_adam_block_rows() items through row_of / last; items untouched by a
 * launch cost 4 bytes; first_block advances by ceil(n / mtgs_adam_block_rows()).  LIST form (row_ids given; not for FLUSH): one
 * row per 16 lanes straight from the frame's list of visible Gaussians.  LIST groups come last in the table; only the FIRST
 * one's first_block matters (where the LIST workgroups begin), the others are assigned on the device from the number of
 * visible items of each tensor.  total_blocks must be an upper bound: tensors that share row_ids and item_start (one node)
 * have the same count and the counts of different nodes add up to at most n_rows, so
 * (largest number of tensors per node) * ceil(n_rows / mtgs_adam_block_list_rows()) + (number of LIST groups) is one. */
int mtgs_adam_group_bytes(void);    /* sizeof(mtgs_adam_group): bindings check their layout against it */
int mtgs_adam_block_elems(void);    /* elements one workgroup updates */
int mtgs_adam_block_rows(void);     /* items one workgroup scans (row-lazy groups, SCAN form) */
int mtgs_adam_block_list_rows(void);  /* rows one workgroup handles (row-lazy groups, LIST form) */
/* flags bit 0 (nontemporal): moments (and a dense gradient) are streamed past the caches (they are touched once per step). */
/* rows_from_block: the row-lazy groups come LAST in the table and own the workgroups [rows_from_block, total_blocks) -- they run
 * as a second kernel with its own register budget (= total_blocks when the table has none, 0 when it has nothing else).
 * flags: bit 0 = nontemporal (above), bit 1 = the table has LIST-form groups: their rank_start is resolved first (one small launch;
 * the table is written). */
int mtgs_adam_step(int n_groups, mtgs_adam_group *table, float *hyper, int64_t total_blocks, int64_t rows_from_block,
                   int flags, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MTGS_RAST_H */
