// node.hip -- fused per-node activations of an MTGS Gaussian node, forward and backward (SURVEY.md section 8f,
// rank 1: the caller side of the rasterization path).
//
// Restates, in one kernel per direction, what VanillaGaussianSplattingModel.get_gaussians computes every step with
// ~12 PyTorch launches (/root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:299-341;
// MultiColorGaussianSplattingModel.get_rgbs, multi_color_gaussian_splatting.py:77-101, differs only in WHERE the
// coefficients come from: features_dc + features_adapters[:, t], features_rest[:, t]):
//     scales    = exp(scales_raw)                         quats = quats_raw / |quats_raw|
//     opacities = sigmoid(opacities_raw)
//     colors    = cat(features_dc[:, None], features_rest)          <- a 384 MB copy at 2M Gaussians, K = 16
//     dirs      = normalize(means.detach() - cam_pos)
//     rgbs      = clamp(spherical_harmonics(n, dirs, colors) + 0.5, 0, 1)     (sigmoid(features_dc) when the model
//                                                                              has sh_degree 0)
// The fused kernels read features_dc / features_rest in place (row strides given, so a per-traversal slice of
// [N,T,K-1,3] needs no gather copy) and the backward writes v_features_dc / v_features_rest directly -- no cat, no
// split, no dirs / clamp temporaries: algorithmic bytes N*(40 + 12 K) + N*44 forward, the same plus N*12 K backward,
// against ~4x that through the operator chain.
//
// Layout (as sh_fwd_k16_kernel): one 16-lane DPP row per Gaussian, lane k = SH basis k; the same lanes also carry the
// small activations (lanes 0-2 scales, 4-7 quaternion -- an aligned quad, so |q|^2 is two quad_perm adds --, lane 8
// opacity).  Roofline: HBM.
#include "common.hpp"
#include "sh_lane.hpp"

namespace {

// The addend must be a ROUNDED product (mul_rounded at the call sites; HIP's __fmul_rn is a plain, contractable `*`): if the compiler contracts the multiply into the
// first add, lane L gets fma(b_L, c_L, round(b_M c_M)) and its partner M the mirror image -- 1 ulp apart, which made a
// Gaussian's colour depend on its position in the wave (tests/test_gpu_large.py renders a subset bit-identically).
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, in every lane
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float quad_sum(float v) {   // sum over the 4 lanes of a quad, in every lane
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}
struct F3 { float x, y, z; };

struct NodeParams {
    const float *means, *scales_raw, *quats_raw, *opac_raw;  // [N,3] [N,3] [N,4] [N]
    const float *dc, *dc_add, *rest;                         // rows of 3, 3 (nullable), Kr*3 floats
    int64_t dc_stride, dc_add_stride, rest_stride;           // row strides in floats
    const float *cam_pos;                                    // [3] device
    int Kr;                                                  // SH bases in `rest` (K - 1)
    int use_sh;                                              // 0: rgbs = sigmoid(dc [+ dc_add]); 1: clamp(SH + 0.5, 0, 1); 2: SH
    const float *pose;                                       // [4] device: instance quaternion wxyz; nullable (static node)
    const float *pose_t;                                     // [3] device: instance translation
    int pose_norm;                                           // 1: the quaternion is a raw parameter row, normalise it
    int skip_colors;                                         // 1: geometry only -- the colours are evaluated for the VISIBLE
                                                             // Gaussians by the rasterizer's front end (viscolor.hip)
};

// Rigid nodes (rigid_node.py:205-216): global mean = R(q) m + t with mtgs utils.quat_to_rotmat (NO normalisation of q:
// get_object_pose hands over a unit quaternion), global quaternion = utils.quat_mult(q, q_local / |q_local|).
struct Pose { float w, x, y, z, tx, ty, tz; float R[9]; };
__device__ __forceinline__ Pose load_pose(const float *q, const float *t, int normalize) {
    Pose o;
    o.w = q[0]; o.x = q[1]; o.y = q[2]; o.z = q[3]; o.tx = t[0]; o.ty = t[1]; o.tz = t[2];
    if (normalize) {   // RigidSubModel.get_object_pose: instance_quats[frame] / |instance_quats[frame]| (rigid_node.py:142)
        const float inv = 1.0f / sqrtf(((o.w * o.w + o.x * o.x) + o.y * o.y) + o.z * o.z);
        o.w *= inv; o.x *= inv; o.y *= inv; o.z *= inv;
    }
    const float xx = o.x * o.x, yy = o.y * o.y, zz = o.z * o.z, xy = o.x * o.y, xz = o.x * o.z, yz = o.y * o.z;
    const float wx = o.w * o.x, wy = o.w * o.y, wz = o.w * o.z;
    o.R[0] = 1.f - 2.f * (yy + zz); o.R[1] = 2.f * (xy - wz); o.R[2] = 2.f * (xz + wy);
    o.R[3] = 2.f * (xy + wz); o.R[4] = 1.f - 2.f * (xx + zz); o.R[5] = 2.f * (yz - wx);
    o.R[6] = 2.f * (xz - wy); o.R[7] = 2.f * (yz + wx); o.R[8] = 1.f - 2.f * (xx + yy);
    return o;
}

// A wave owns 64 consecutive Gaussians and uses TWO lane mappings:
//   * lane-per-Gaussian for the small activations and for everything that is 3..4 floats per Gaussian
//     (64 x 12 / 16 / 4 contiguous bytes per instruction -- per-row 4-byte accesses cost one instruction per 4
//     Gaussians and bounded the first version at 2.7 TB/s);
//   * the 16-lane-row mapping for the coefficients: in step `it`, row `sub` works on Gaussian 4*it + sub, so one
//     wave instruction covers 4 CONSECUTIVE Gaussians (720-768 contiguous bytes; with 16*sub + it the four 180-byte
//     segments were 2.9 KB apart and the forward ran at 3.0 TB/s).  Values move between the mappings with one
//     ds_bpermute each: lane k of row `sub` keeps the result of step k, i.e. of Gaussian 4k + sub, which the owner
//     lane l = 4k + sub fetches from lane 16*(l % 4) + l / 4; in the backward row `sub` reads lane 4*it + sub.
constexpr int NODE_BLOCK = 256, NODE_PER_WAVE = 64, NODE_STEPS = 16;
struct F4 { float x, y, z, w; };

// The work of ONE wave: Gaussians [g0, g0 + 64) of a node with N Gaussians.
template <int DEG>
__device__ __forceinline__ void node_fwd_wave(const int64_t N, const NodeParams &P, const int64_t g0, float *__restrict__ scales,
                                              float *__restrict__ quats, float *__restrict__ opacities,
                                              float *__restrict__ rgbs, uint8_t *__restrict__ clamp_mask,
                                              float *__restrict__ means_out) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const int lane = threadIdx.x & 63, k = lane & 15, sub = lane >> 4;
    const ShLaneConst lc = sh_lane_const(k);
    const bool active = !P.skip_colors && (P.use_sh ? (k < NB && k - 1 < P.Kr) : (k == 0));
    const float camx = P.cam_pos[0], camy = P.cam_pos[1], camz = P.cam_pos[2];
    // ---- lane-per-Gaussian loads
    const int64_t gl = g0 + lane;
    const bool okl = gl < N;
    F3 sr = F3{0.f, 0.f, 0.f};
    F4 qr = F4{1.f, 0.f, 0.f, 0.f};
    float orw = 0.f;
    if (okl) {
        sr = *reinterpret_cast<const F3 *>(P.scales_raw + gl * 3);
        qr = *reinterpret_cast<const F4 *>(P.quats_raw + gl * 4);
        orw = P.opac_raw[gl];
    }
    // unit view direction, once per Gaussian in the lane-per-Gaussian mapping (the rows fetch it by ds_bpermute)
    float dx = 0.f, dy = 0.f, dz = 1.f;
    Pose ps;
    if (P.pose) ps = load_pose(P.pose, P.pose_t, P.pose_norm);
    if (((P.use_sh && !P.skip_colors) || P.pose || means_out) && okl) {
        F3 mn = *reinterpret_cast<const F3 *>(P.means + gl * 3);
        if (P.pose)   // rigid node: the Gaussian lives in the object frame
            mn = F3{(ps.R[0] * mn.x + ps.R[1] * mn.y) + ps.R[2] * mn.z + ps.tx, (ps.R[3] * mn.x + ps.R[4] * mn.y) + ps.R[5] * mn.z + ps.ty,
                    (ps.R[6] * mn.x + ps.R[7] * mn.y) + ps.R[8] * mn.z + ps.tz};
        if (means_out) *reinterpret_cast<F3 *>(means_out + gl * 3) = mn;
        dx = mn.x - camx; dy = mn.y - camy; dz = mn.z - camz;
        const float inorm = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
        dx *= inorm; dy *= inorm; dz *= inorm;
    }
    // ---- coefficient rows: 16 steps of 4 consecutive Gaussians, every load issued before the first use
    F3 c[NODE_STEPS];
#pragma unroll
    for (int it = 0; it < NODE_STEPS; ++it) {
        const int64_t g = g0 + it * 4 + sub;
        c[it] = F3{0.f, 0.f, 0.f};
        if (g < N && active) {
            if (k == 0) {
                c[it] = *reinterpret_cast<const F3 *>(P.dc + g * P.dc_stride);
                if (P.dc_add) {
                    const F3 a = *reinterpret_cast<const F3 *>(P.dc_add + g * P.dc_add_stride);
                    c[it].x += a.x; c[it].y += a.y; c[it].z += a.z;
                }
            } else {
                c[it] = *reinterpret_cast<const F3 *>(P.rest + g * P.rest_stride + (k - 1) * 3);
            }
        }
    }
    float myr = 0.f, myg = 0.f, myb = 0.f;
    if (!P.skip_colors) {
#pragma unroll
    for (int it = 0; it < NODE_STEPS; ++it) {
        float r, gg, bb;
        if (P.use_sh) {
            const int src = (it * 4 + sub) << 2;   // the lane that owns Gaussian 4*it + sub
            const float x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dx)));
            const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dy)));
            const float z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dz)));
            const float b = sh_lane_basis<DEG>(lc, x, y, z);
            r = row16_sum(mul_rounded(b, c[it].x)); gg = row16_sum(mul_rounded(b, c[it].y)); bb = row16_sum(mul_rounded(b, c[it].z));
        } else {  // lane 0 of the row holds the coefficients; every lane of the row gets them
            r = row16_sum(c[it].x); gg = row16_sum(c[it].y); bb = row16_sum(c[it].z);
        }
        const bool mine = k == it;  // lane k of the row keeps the result of step k
        myr = mine ? r : myr; myg = mine ? gg : myg; myb = mine ? bb : myb;
    }
    {   // the owner of Gaussian l fetches its colour from lane 16 * (l % 4) + l / 4
        const int src = ((lane & 3) * 16 + (lane >> 2)) << 2;
        myr = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(myr)));
        myg = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(myg)));
        myb = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(myb)));
    }
    }
    if (!okl) return;
    // ---- lane-per-Gaussian results
    F3 rgb;
    if (P.skip_colors) {
        rgb = F3{0.f, 0.f, 0.f};
    } else if (P.use_sh == 2) {   // the raw SH value: the colour activation is the consumer's (data-parallel exchange, front.hip)
        rgb = F3{myr, myg, myb};
        clamp_mask[gl] = 7;
    } else if (P.use_sh) {
        const float x = myr + 0.5f, y = myg + 0.5f, z = myb + 0.5f;
        rgb = F3{fminf(fmaxf(x, 0.f), 1.f), fminf(fmaxf(y, 0.f), 1.f), fminf(fmaxf(z, 0.f), 1.f)};
        // torch.clamp passes the gradient where min <= x <= max (inclusive): one bit per channel
        clamp_mask[gl] = (uint8_t)((x >= 0.f && x <= 1.f) | ((y >= 0.f && y <= 1.f) << 1) | ((z >= 0.f && z <= 1.f) << 2));
    } else {
        rgb = F3{1.f / (1.f + expf(-myr)), 1.f / (1.f + expf(-myg)), 1.f / (1.f + expf(-myb))};
        clamp_mask[gl] = 7;
    }
    if (!P.skip_colors) *reinterpret_cast<F3 *>(rgbs + gl * 3) = rgb;
    *reinterpret_cast<F3 *>(scales + gl * 3) = F3{expf(sr.x), expf(sr.y), expf(sr.z)};
    const float qinv = 1.0f / sqrtf(((qr.x * qr.x + qr.y * qr.y) + qr.z * qr.z) + qr.w * qr.w);
    F4 qn = F4{qr.x * qinv, qr.y * qinv, qr.z * qinv, qr.w * qinv};   // (w, x, y, z)
    if (P.pose) {   // utils.quat_mult(q_instance, q_local)
        const float w2 = qn.x, x2 = qn.y, y2 = qn.z, z2 = qn.w;
        qn = F4{((ps.w * w2 - ps.x * x2) - ps.y * y2) - ps.z * z2, ((ps.w * x2 + ps.x * w2) + ps.y * z2) - ps.z * y2,
                ((ps.w * y2 - ps.x * z2) + ps.y * w2) + ps.z * x2, ((ps.w * z2 + ps.x * y2) - ps.y * x2) + ps.z * w2};
    }
    *reinterpret_cast<F4 *>(quats + gl * 4) = qn;
    opacities[gl] = 1.f / (1.f + expf(-orw));
}

template <int DEG>
__global__ __launch_bounds__(NODE_BLOCK) void node_fwd_kernel(int64_t N, const NodeParams P, float *__restrict__ scales,
                                                              float *__restrict__ quats, float *__restrict__ opacities,
                                                              float *__restrict__ rgbs, uint8_t *__restrict__ clamp_mask,
                                                              float *__restrict__ means_out) {
    const int64_t g0 = ((int64_t)blockIdx.x * (NODE_BLOCK / 64) + (threadIdx.x >> 6)) * NODE_PER_WAVE;
    if (g0 >= N) return;
    node_fwd_wave<DEG>(N, P, g0, scales, quats, opacities, rgbs, clamp_mask, means_out);
}

// ---- all nodes of a scene in ONE launch: a table of descriptors in device memory (include/mtgs_rast.h) --------------
// Workgroup b belongs to the node i with table[i].first_block <= b < table[i + 1].first_block (binary search; the index
// is wave-uniform, so the descriptor arrives through scalar loads) and handles 256 of its Gaussians.
__device__ __forceinline__ int node_of_block(const mtgs_node_desc *__restrict__ table, int n_nodes, int64_t b) {
    int lo = 0, hi = n_nodes - 1;   // last node whose first_block <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    return __builtin_amdgcn_readfirstlane(lo);
}
// frame_dev (nullable): `pose` / `pose_trans` / the gradient rows are the FIRST rows of the node's per-frame tables and the frame of
// this step is read from device memory when the kernel runs (one captured iteration for every frame: the caller rewrites the word)
__device__ __forceinline__ int64_t frame_of(const mtgs_node_desc &d) { return d.frame_dev ? (int64_t)*d.frame_dev : 0; }
__device__ __forceinline__ const float *pose_q(const mtgs_node_desc &d) { return d.pose ? d.pose + 4 * frame_of(d) : nullptr; }
__device__ __forceinline__ const float *pose_t(const mtgs_node_desc &d) { return d.pose_trans ? d.pose_trans + 3 * frame_of(d) : nullptr; }
__device__ __forceinline__ NodeParams params_of(const mtgs_node_desc &d, const float *cam_pos) {
    return NodeParams{d.means, d.scales_raw, d.quats_raw, d.opacities_raw, d.features_dc, d.features_dc_add, d.features_rest,
                      d.dc_stride, d.dc_add_stride, d.rest_stride, cam_pos, d.k_rest, d.use_sh, pose_q(d), pose_t(d), d.pose_normalize,
                      d.skip_colors};
}

template <int DEG>
__global__ __launch_bounds__(NODE_BLOCK) void node_fwd_batch_kernel(const mtgs_node_desc *__restrict__ table, int n_nodes,
                                                                    const float *__restrict__ cam_pos,
                                                                    int64_t *__restrict__ model_id) {
    const int i = node_of_block(table, n_nodes, blockIdx.x);
    const mtgs_node_desc &d = table[i];
    const int64_t g0 = (((int64_t)blockIdx.x - d.first_block) * (NODE_BLOCK / 64) + (threadIdx.x >> 6)) * NODE_PER_WAVE;
    if (g0 >= d.n) return;
    if (model_id) {
        const int64_t gl = g0 + (threadIdx.x & 63);
        if (gl < d.n) model_id[d.start + gl] = i;
    }
    node_fwd_wave<DEG>(d.n, params_of(d, cam_pos), g0, d.scales, d.quats, d.opacities, d.rgbs, d.clamp_mask, d.means_out);
}

// Geometry only (every descriptor has skip_colors = 1: visibility-first colours), `degree = -1` of mtgs_node_fwd_batch: the
// activations alone -- exp, normalise (+ the rigid pose), sigmoid, the means -- one Gaussian per lane, 44 bytes in and 52 out.
// Separate from node_fwd_wave because that kernel's 16 coefficient rows per lane fix its register count (and the skipped colour
// loop still issued its address arithmetic): 86 us for 192 MB there.  Same expressions in the same order: same bits.
__global__ __launch_bounds__(NODE_BLOCK) void node_fwd_geom_batch_kernel(const mtgs_node_desc *__restrict__ table, int n_nodes,
                                                                         int64_t *__restrict__ model_id) {
    const int i = node_of_block(table, n_nodes, blockIdx.x);
    const mtgs_node_desc &d = table[i];
    const int64_t gl = ((int64_t)blockIdx.x - d.first_block) * NODE_BLOCK + threadIdx.x;
    if (gl >= d.n) return;
    const F3 sr = *reinterpret_cast<const F3 *>(d.scales_raw + gl * 3);
    const F4 qr = *reinterpret_cast<const F4 *>(d.quats_raw + gl * 4);
    const float orw = d.opacities_raw[gl];
    F3 mn = F3{0.f, 0.f, 0.f};
    if (d.pose || d.means_out) mn = *reinterpret_cast<const F3 *>(d.means + gl * 3);
    if (model_id) model_id[d.start + gl] = i;
    Pose ps;
    if (d.pose) {
        ps = load_pose(pose_q(d), pose_t(d), d.pose_normalize);
        mn = F3{(ps.R[0] * mn.x + ps.R[1] * mn.y) + ps.R[2] * mn.z + ps.tx, (ps.R[3] * mn.x + ps.R[4] * mn.y) + ps.R[5] * mn.z + ps.ty,
                (ps.R[6] * mn.x + ps.R[7] * mn.y) + ps.R[8] * mn.z + ps.tz};
    }
    if (d.means_out) *reinterpret_cast<F3 *>(d.means_out + gl * 3) = mn;
    *reinterpret_cast<F3 *>(d.scales + gl * 3) = F3{expf(sr.x), expf(sr.y), expf(sr.z)};
    const float qinv = 1.0f / sqrtf(((qr.x * qr.x + qr.y * qr.y) + qr.z * qr.z) + qr.w * qr.w);
    F4 qn = F4{qr.x * qinv, qr.y * qinv, qr.z * qinv, qr.w * qinv};   // (w, x, y, z)
    if (d.pose) {   // utils.quat_mult(q_instance, q_local)
        const float w2 = qn.x, x2 = qn.y, y2 = qn.z, z2 = qn.w;
        qn = F4{((ps.w * w2 - ps.x * x2) - ps.y * y2) - ps.z * z2, ((ps.w * x2 + ps.x * w2) + ps.y * z2) - ps.z * y2,
                ((ps.w * y2 - ps.x * z2) + ps.y * w2) + ps.z * x2, ((ps.w * z2 + ps.x * y2) - ps.y * x2) + ps.z * w2};
    }
    *reinterpret_cast<F4 *>(d.quats + gl * 4) = qn;
    d.opacities[gl] = 1.f / (1.f + expf(-orw));
}

// The geometry half of the node backward for the VISIBLE Gaussians only: the projection backward's workspace rows
// ws[r] = [v_mean 3 | v_quat 4 | v_scale 3 | v_opacity 1 | pad] (gradients with respect to the ACTIVATED Gaussian vis_ids[r])
// -> rows of gradients with respect to the raw parameters, out[r] = [means 3 | scales 3 | quats 4 | opacities 1 | pad]: the
// exp / normalise / sigmoid VJPs of node_bwd_wave, same expressions.  Static nodes (a rigid node reduces a pose gradient over
// all of its Gaussians: refused by the caller).  The optimizer takes the rows through its row map (adam.hip): no dense
// [N, .] gradient of the geometry is written or read.
__global__ __launch_bounds__(NODE_BLOCK) void node_bwd_rows_kernel(const mtgs_node_desc *__restrict__ table, int n_nodes,
                                                                   const int32_t *__restrict__ vis_ids, const int64_t *__restrict__ totals,
                                                                   int64_t cap_vis, const float *__restrict__ ws, int ws_stride,
                                                                   float *__restrict__ out) {
    int64_t n_vis = totals ? *totals >> 32 : cap_vis;
    if (n_vis > cap_vis) n_vis = cap_vis;
    const int64_t r = (int64_t)blockIdx.x * NODE_BLOCK + threadIdx.x;
    if (r >= n_vis) return;
    const float4 *w = reinterpret_cast<const float4 *>(ws + r * ws_stride);
    const float4 w0 = w[0], w1 = w[1], w2 = w[2];       // (v_mean xyz, vq.w) (vq.xyz', vs.x) (vs.yz, v_opacity, -)
    if (w0.x == 0.f && w0.y == 0.f && w0.z == 0.f && w0.w == 0.f && w1.x == 0.f && w1.y == 0.f && w1.z == 0.f && w1.w == 0.f &&
        w2.x == 0.f && w2.y == 0.f && w2.z == 0.f) {
        // an occluded Gaussian (no gradient reached it: most frustum-visible ones): zero row, no parameter gathers, no node search
        float4 *dz = reinterpret_cast<float4 *>(out + r * 12);
        dz[0] = dz[1] = dz[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int64_t g = vis_ids[r];
    int lo = 0, hi = n_nodes - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].start <= g) lo = mid; else hi = mid - 1;
    }
    const mtgs_node_desc &d = table[lo];
    const int64_t gl = g - d.start;
    // exp / sigmoid recomputed from the RAW parameters (the expressions of the forward, so the same bits): the activated
    // tensors are autograd outputs that are released when the backward has run, i.e. before this kernel is enqueued
    const F3 sr = *reinterpret_cast<const F3 *>(d.scales_raw + gl * 3);
    const F3 s = F3{expf(sr.x), expf(sr.y), expf(sr.z)};
    const F4 q = *reinterpret_cast<const F4 *>(d.quats_raw + gl * 4);
    const F4 vq = F4{w0.w, w1.x, w1.y, w1.z};
    const float qinv = 1.0f / sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    const F4 qn = F4{q.x * qinv, q.y * qinv, q.z * qinv, q.w * qinv};
    const float dot = ((vq.x * qn.x + vq.y * qn.y) + vq.z * qn.z) + vq.w * qn.w;                // d (q / |q|)
    const float o = 1.f / (1.f + expf(-d.opacities_raw[gl]));
    float4 *dst = reinterpret_cast<float4 *>(out + r * 12);
    dst[0] = make_float4(w0.x, w0.y, w0.z, w1.w * s.x);                                          // means | d exp = exp
    dst[1] = make_float4(w2.x * s.y, w2.y * s.z, (vq.x - dot * qn.x) * qinv, (vq.y - dot * qn.y) * qinv);
    dst[2] = make_float4((vq.z - dot * qn.z) * qinv, (vq.w - dot * qn.w) * qinv, w2.z * o * (1.f - o), 0.f);   // d sigmoid
}

struct NodeGrads {   // cotangents of the activated Gaussians and the gradients of the raw parameters
    const float *v_scales, *v_quats, *v_opacities, *v_rgbs, *v_means;
    float *g_scales_raw, *g_quats_raw, *g_opac_raw, *g_dc, *g_rest, *g_dc_add, *g_means, *g_pose;
    int n_trav, trav;
};

template <int DEG>
__device__ __forceinline__ void node_bwd_wave(const int64_t N, const NodeParams &P, const int64_t g0,
                                              const float *__restrict__ scales, const float *__restrict__ opacities,
                                              const float *__restrict__ rgbs, const uint8_t *__restrict__ clamp_mask,
                                              const NodeGrads &G) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const int lane = threadIdx.x & 63, k = lane & 15, sub = lane >> 4;
    const ShLaneConst lc = sh_lane_const(k);
    const float *__restrict__ v_scales = G.v_scales, *__restrict__ v_quats = G.v_quats, *__restrict__ v_opacities = G.v_opacities,
                *__restrict__ v_rgbs = G.v_rgbs, *__restrict__ v_means = G.v_means;
    float *__restrict__ g_scales_raw = G.g_scales_raw, *__restrict__ g_quats_raw = G.g_quats_raw, *__restrict__ g_opac_raw = G.g_opac_raw,
          *__restrict__ g_dc = G.g_dc, *__restrict__ g_rest = G.g_rest, *__restrict__ g_dc_add = G.g_dc_add,
          *__restrict__ g_means = G.g_means, *__restrict__ g_pose = G.g_pose;
    const int n_trav = G.n_trav, trav = G.trav;
    const float camx = P.cam_pos[0], camy = P.cam_pos[1], camz = P.cam_pos[2];
    // ---- lane-per-Gaussian: activations, and the colour cotangent with the clamp / sigmoid VJP applied
    const int64_t gl = g0 + lane;
    const bool okl = gl < N;
    F3 v = F3{0.f, 0.f, 0.f}, mn = F3{0.f, 0.f, 1.f};
    Pose ps;
    if (P.pose) ps = load_pose(P.pose, P.pose_t, P.pose_norm);
    float pq0 = 0.f, pq1 = 0.f, pq2 = 0.f, pq3 = 0.f, pt0 = 0.f, pt1 = 0.f, pt2 = 0.f;   // this Gaussian's part of d pose
    if (okl) {
        if (v_rgbs) v = *reinterpret_cast<const F3 *>(v_rgbs + gl * 3);
        if (!v_rgbs) {
        } else if (P.use_sh) {
            const unsigned mk = clamp_mask[gl];
            v.x = (mk & 1u) ? v.x : 0.f; v.y = (mk & 2u) ? v.y : 0.f; v.z = (mk & 4u) ? v.z : 0.f;
        } else {
            const F3 y = *reinterpret_cast<const F3 *>(rgbs + gl * 3);
            v.x *= y.x * (1.f - y.x); v.y *= y.y * (1.f - y.y); v.z *= y.z * (1.f - y.z);
        }
        if (P.use_sh || P.pose) {
            const F3 ml = *reinterpret_cast<const F3 *>(P.means + gl * 3);
            mn = ml;
            if (P.pose)
                mn = F3{(ps.R[0] * ml.x + ps.R[1] * ml.y) + ps.R[2] * ml.z + ps.tx, (ps.R[3] * ml.x + ps.R[4] * ml.y) + ps.R[5] * ml.z + ps.ty,
                        (ps.R[6] * ml.x + ps.R[7] * ml.y) + ps.R[8] * ml.z + ps.tz};
            if (g_means || (P.pose && g_pose)) {
                const F3 vm = v_means ? *reinterpret_cast<const F3 *>(v_means + gl * 3) : F3{0.f, 0.f, 0.f};
                if (P.pose) {
                    // m_g = R m + t:  d m = R^T v,  d t = v,  d R[i][j] = v_i m_j  contracted with d R / d q (quat_to_rotmat)
                    if (g_means)
                        *reinterpret_cast<F3 *>(g_means + gl * 3) =
                            F3{(ps.R[0] * vm.x + ps.R[3] * vm.y) + ps.R[6] * vm.z, (ps.R[1] * vm.x + ps.R[4] * vm.y) + ps.R[7] * vm.z,
                               (ps.R[2] * vm.x + ps.R[5] * vm.y) + ps.R[8] * vm.z};
                    pt0 = vm.x; pt1 = vm.y; pt2 = vm.z;
                    const float r00 = vm.x * ml.x, r01 = vm.x * ml.y, r02 = vm.x * ml.z, r10 = vm.y * ml.x, r11 = vm.y * ml.y,
                                r12 = vm.y * ml.z, r20 = vm.z * ml.x, r21 = vm.z * ml.y, r22 = vm.z * ml.z;
                    pq0 = 2.f * (((-ps.z * r01 + ps.y * r02) + (ps.z * r10 - ps.x * r12)) + (-ps.y * r20 + ps.x * r21));
                    pq1 = 2.f * ((((ps.y * r01 + ps.z * r02) + ps.y * r10) - 2.f * ps.x * r11 - ps.w * r12) + ((ps.z * r20 + ps.w * r21) - 2.f * ps.x * r22));
                    pq2 = 2.f * ((((-2.f * ps.y * r00 + ps.x * r01) + ps.w * r02) + (ps.x * r10 + ps.z * r12)) + ((-ps.w * r20 + ps.z * r21) - 2.f * ps.y * r22));
                    pq3 = 2.f * ((((-2.f * ps.z * r00 - ps.w * r01) + ps.x * r02) + ((ps.w * r10 - 2.f * ps.z * r11) + ps.y * r12)) + (ps.x * r20 + ps.y * r21));
                } else if (g_means) {
                    *reinterpret_cast<F3 *>(g_means + gl * 3) = vm;
                }
            }
        }
        const F3 s = *reinterpret_cast<const F3 *>(scales + gl * 3), vs = *reinterpret_cast<const F3 *>(v_scales + gl * 3);
        *reinterpret_cast<F3 *>(g_scales_raw + gl * 3) = F3{vs.x * s.x, vs.y * s.y, vs.z * s.z};  // d exp = exp
        const F4 q = *reinterpret_cast<const F4 *>(P.quats_raw + gl * 4);
        F4 vq = *reinterpret_cast<const F4 *>(v_quats + gl * 4);
        const float qinv = 1.0f / sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
        const F4 qn = F4{q.x * qinv, q.y * qinv, q.z * qinv, q.w * qinv};
        if (P.pose) {   // VJP of utils.quat_mult(q_instance, qn) with respect to both factors
            const float vw = vq.x, vx = vq.y, vy = vq.z, vz = vq.w, w2 = qn.x, x2 = qn.y, y2 = qn.z, z2 = qn.w;
            pq0 += ((vw * w2 + vx * x2) + vy * y2) + vz * z2;
            pq1 += ((-vw * x2 + vx * w2) - vy * z2) + vz * y2;
            pq2 += ((-vw * y2 + vx * z2) + vy * w2) - vz * x2;
            pq3 += ((-vw * z2 - vx * y2) + vy * x2) + vz * w2;
            vq = F4{((vw * ps.w + vx * ps.x) + vy * ps.y) + vz * ps.z, ((-vw * ps.x + vx * ps.w) + vy * ps.z) - vz * ps.y,
                    ((-vw * ps.y - vx * ps.z) + vy * ps.w) + vz * ps.x, ((-vw * ps.z + vx * ps.y) - vy * ps.x) + vz * ps.w};
        }
        const float dot = ((vq.x * qn.x + vq.y * qn.y) + vq.z * qn.z) + vq.w * qn.w;                // d (q / |q|)
        *reinterpret_cast<F4 *>(g_quats_raw + gl * 4) =
            F4{(vq.x - dot * qn.x) * qinv, (vq.y - dot * qn.y) * qinv, (vq.z - dot * qn.z) * qinv, (vq.w - dot * qn.w) * qinv};
        const float o = opacities[gl];
        g_opac_raw[gl] = v_opacities[gl] * o * (1.f - o);                                           // d sigmoid
    }
    if (P.pose && g_pose) {   // every lane takes part (zeros outside the node): one atomic per wave and component
        const float t0 = wave_sum_to_lane63(pq0), t1 = wave_sum_to_lane63(pq1), t2 = wave_sum_to_lane63(pq2), t3 = wave_sum_to_lane63(pq3);
        const float t4 = wave_sum_to_lane63(pt0), t5 = wave_sum_to_lane63(pt1), t6 = wave_sum_to_lane63(pt2);
        if (lane == 63) {
            atomicAdd(g_pose + 0, t0); atomicAdd(g_pose + 1, t1); atomicAdd(g_pose + 2, t2); atomicAdd(g_pose + 3, t3);
            atomicAdd(g_pose + 4, t4); atomicAdd(g_pose + 5, t5); atomicAdd(g_pose + 6, t6);
        }
    }
    // no colour gradient asked for (the data-parallel exchange rebuilds it on the receivers): the coefficient rows are
    // not written at all -- N * K * 12 bytes, the largest stream of this kernel
    if (!g_dc && !g_rest) return;
    // the basis needs the unit view direction: computed once per Gaussian here, moved to the rows below
    float dx = mn.x - camx, dy = mn.y - camy, dz = mn.z - camz;
    const float inorm = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
    dx *= inorm; dy *= inorm; dz *= inorm;
    // ---- rows: v_coeff[k, :] = basis_k(dir) * v
    if (n_trav > 0) {
        // Per-traversal parameters [N, T, ...]: the gradient of the FULL tensors is written here -- slice `trav` gets the
        // values, the other traversals zeros -- instead of autograd's zero-fill + strided slice copy.  Row `sub` of step
        // s handles (Gaussian, traversal) pair 4 s + sub, so the four rows of a store instruction cover four CONSECUTIVE
        // [K-1, 3] blocks (720 contiguous bytes); the pair's Gaussian is found with a multiply-high (T is small).
        if (okl) *reinterpret_cast<F3 *>(g_dc + gl * 3) = P.use_sh ? F3{0.2820947917738781f * v.x, 0.2820947917738781f * v.y, 0.2820947917738781f * v.z} : v;
        const uint32_t magic = 0xFFFFFFFFu / (uint32_t)n_trav + 1u;
        const int n_here = (int)min((int64_t)NODE_PER_WAVE, N - g0);
        const int n_steps = (n_here * n_trav + 3) >> 2;
        for (int st = 0; st < n_steps; ++st) {
            const uint32_t q = (uint32_t)(4 * st + sub);
            const uint32_t gi = __umulhi(q, magic);          // q / n_trav
            const int tt = (int)(q - gi * (uint32_t)n_trav);
            const int src = (int)(gi & 63u) << 2;            // the lane that owns Gaussian gi
            const float vx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.x)));
            const float vy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.y)));
            const float vz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.z)));
            float b = 0.f;
            if (P.use_sh) {
                const float x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dx)));
                const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dy)));
                const float z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dz)));
                b = (k < NB && tt == trav) ? sh_lane_basis<DEG>(lc, x, y, z) : 0.f;
            } else {
                b = (k == 0 && tt == trav) ? 1.f : 0.f;
            }
            if ((int)gi >= n_here) continue;
            const int64_t pair = (g0 + gi) * n_trav + tt;
            const F3 o = F3{b * vx, b * vy, b * vz};
            if (k == 0) { if (g_dc_add) *reinterpret_cast<F3 *>(g_dc_add + pair * 3) = o; }
            else if (k - 1 < P.Kr) *reinterpret_cast<F3 *>(g_rest + (pair * P.Kr + (k - 1)) * 3) = o;
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < NODE_STEPS; ++it) {
        const int src = (it * 4 + sub) << 2;   // the lane that owns Gaussian 4*it + sub
        const float vx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.x)));
        const float vy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.y)));
        const float vz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v.z)));
        float b;
        if (P.use_sh) {
            const float x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dx)));
            const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dy)));
            const float z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(dz)));
            b = k < NB ? sh_lane_basis<DEG>(lc, x, y, z) : 0.f;
        } else {
            b = k == 0 ? 1.f : 0.f;
        }
        const int64_t g = g0 + it * 4 + sub;
        if (g >= N) continue;
        const F3 o = F3{b * vx, b * vy, b * vz};
        if (k == 0) *reinterpret_cast<F3 *>(g_dc + g * 3) = o;
        else if (k - 1 < P.Kr) *reinterpret_cast<F3 *>(g_rest + (g * P.Kr + (k - 1)) * 3) = o;
    }
}

template <int DEG>
__global__ __launch_bounds__(NODE_BLOCK) void node_bwd_kernel(int64_t N, const NodeParams P, const float *__restrict__ scales,
                                                              const float *__restrict__ opacities,
                                                              const float *__restrict__ rgbs,
                                                              const uint8_t *__restrict__ clamp_mask, const NodeGrads G) {
    const int64_t g0 = ((int64_t)blockIdx.x * (NODE_BLOCK / 64) + (threadIdx.x >> 6)) * NODE_PER_WAVE;
    if (g0 >= N) return;
    node_bwd_wave<DEG>(N, P, g0, scales, opacities, rgbs, clamp_mask, G);
}

template <int DEG>
__global__ __launch_bounds__(NODE_BLOCK) void node_bwd_batch_kernel(const mtgs_node_desc *__restrict__ table, int n_nodes,
                                                                    const float *__restrict__ cam_pos) {
    const int i = node_of_block(table, n_nodes, blockIdx.x);
    const mtgs_node_desc &d = table[i];
    const int64_t g0 = (((int64_t)blockIdx.x - d.first_block) * (NODE_BLOCK / 64) + (threadIdx.x >> 6)) * NODE_PER_WAVE;
    if (g0 >= d.n) return;
    NodeParams P = params_of(d, cam_pos);
    P.dc_stride = 3; P.dc_add_stride = 3; P.rest_stride = (int64_t)d.k_rest * 3;   // the gradients are dense rows
    const NodeGrads G{d.v_scales, d.v_quats, d.v_opacities, d.v_rgbs, d.v_means, d.g_scales_raw, d.g_quats_raw, d.g_opacities_raw,
                      d.g_features_dc, d.g_features_rest, d.g_features_dc_add, d.g_means, d.g_pose, d.n_traversals, d.traversal};
    node_bwd_wave<DEG>(d.n, P, g0, d.scales, d.opacities, d.rgbs, d.clamp_mask, G);
}

// After the batched backward: for the nodes whose pose is a row of the per-frame parameter tables, turn the accumulated
// gradient of the NORMALISED quaternion into the gradient of the raw row (d (q / |q|)) and store both rows.
__global__ void node_pose_finalize_kernel(const mtgs_node_desc *__restrict__ table, int n_nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const mtgs_node_desc &d = table[i];
    if (!d.pose || !d.pose_normalize || !d.g_pose) return;
    const float *pq = pose_q(d);
    const int64_t f = frame_of(d);
    const float w = pq[0], x = pq[1], y = pq[2], z = pq[3];
    const float inv = 1.0f / sqrtf(((w * w + x * x) + y * y) + z * z);
    const float qn[4] = {w * inv, x * inv, y * inv, z * inv};
    const float dot = ((d.g_pose[0] * qn[0] + d.g_pose[1] * qn[1]) + d.g_pose[2] * qn[2]) + d.g_pose[3] * qn[3];
    if (d.g_pose_quat_row)
        for (int k = 0; k < 4; ++k) d.g_pose_quat_row[4 * f + k] = (d.g_pose[k] - dot * qn[k]) * inv;
    if (d.g_pose_trans_row)
        for (int k = 0; k < 3; ++k) d.g_pose_trans_row[3 * f + k] = d.g_pose[4 + k];
}

}  // namespace

#define MTGS_NODE_DISPATCH(KERNEL, ...)                                                 \
    switch (degree) {                                                                   \
        case 0: KERNEL<0><<<grid, NODE_BLOCK, 0, st>>>(__VA_ARGS__); break;             \
        case 1: KERNEL<1><<<grid, NODE_BLOCK, 0, st>>>(__VA_ARGS__); break;             \
        case 2: KERNEL<2><<<grid, NODE_BLOCK, 0, st>>>(__VA_ARGS__); break;             \
        default: KERNEL<3><<<grid, NODE_BLOCK, 0, st>>>(__VA_ARGS__); break;            \
    }

static int node_check(const char *who, int64_t N, int K_rest, int degree, int use_sh, const int64_t *strides) {
    MTGS_REQUIRE(N >= 0 && K_rest >= 0, MTGS_EINVAL, "%s: bad sizes", who);
    MTGS_REQUIRE(degree >= 0 && degree <= 3 && K_rest <= 15, MTGS_EUNSUPPORTED,
                 "%s: degree %d / %d higher-order bases (one basis per lane of a 16-lane row: degree <= 3, K <= 16)", who,
                 degree, K_rest);
    MTGS_REQUIRE(!use_sh || (degree + 1) * (degree + 1) <= K_rest + 1, MTGS_EINVAL,
                 "%s: degree %d needs (degree+1)^2 <= K = %d", who, degree, K_rest + 1);
    MTGS_REQUIRE(strides && strides[0] >= 3 && strides[2] >= (int64_t)K_rest * 3, MTGS_EINVAL, "%s: bad row strides", who);
    return MTGS_OK;
}

extern "C" int mtgs_node_fwd(int64_t N, int K_rest, int degree, int use_sh, const float *means, const float *scales_raw,
                             const float *quats_raw, const float *opacities_raw, const float *features_dc,
                             const float *features_dc_add, const float *features_rest, const int64_t *row_strides,
                             const float *cam_pos, float *scales, float *quats, float *opacities, float *rgbs,
                             uint8_t *clamp_mask, const float *pose, float *means_out, void *stream) {
    if (int rc = node_check("mtgs_node_fwd", N, K_rest, degree, use_sh, row_strides)) return rc;
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(means && scales_raw && quats_raw && opacities_raw && features_dc && (features_rest || K_rest == 0) && cam_pos &&
                     scales && quats && opacities && rgbs && clamp_mask,
                 MTGS_EINVAL, "mtgs_node_fwd: null pointer");
    const NodeParams P{means, scales_raw, quats_raw, opacities_raw, features_dc, features_dc_add, features_rest,
                       row_strides[0], row_strides[1], row_strides[2], cam_pos, K_rest, use_sh, pose, pose ? pose + 4 : nullptr, 0};
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div64(N, NODE_PER_WAVE * (NODE_BLOCK / 64));
    MTGS_NODE_DISPATCH(node_fwd_kernel, N, P, scales, quats, opacities, rgbs, clamp_mask, means_out)
    MTGS_CHECK_LAUNCH("mtgs_node_fwd");
    return MTGS_OK;
}

extern "C" int mtgs_node_bwd(int64_t N, int K_rest, int degree, int use_sh, const float *means, const float *quats_raw,
                             const float *cam_pos, const float *scales, const float *opacities, const float *rgbs,
                             const uint8_t *clamp_mask, const float *v_scales, const float *v_quats,
                             const float *v_opacities, const float *v_rgbs, float *g_scales_raw, float *g_quats_raw,
                             float *g_opacities_raw, float *g_features_dc, float *g_features_rest,
                             float *g_features_dc_add, int n_traversals, int traversal, const float *pose,
                             const float *v_means, float *g_means, float *g_pose, void *stream) {
    const int64_t strides[3] = {3, 3, (int64_t)K_rest * 3};
    if (int rc = node_check("mtgs_node_bwd", N, K_rest, degree, use_sh, strides)) return rc;
    if (N == 0) return MTGS_OK;
    MTGS_REQUIRE(means && quats_raw && cam_pos && scales && opacities && rgbs && clamp_mask && v_scales && v_quats &&
                     v_opacities && v_rgbs && g_scales_raw && g_quats_raw && g_opacities_raw && g_features_dc &&
                     (g_features_rest || K_rest == 0),
                 MTGS_EINVAL, "mtgs_node_bwd: null pointer");
    MTGS_REQUIRE(n_traversals >= 0 && (n_traversals == 0 || (traversal >= 0 && traversal < n_traversals)), MTGS_EINVAL,
                 "mtgs_node_bwd: traversal %d of %d", traversal, n_traversals);
    const NodeParams P{means, nullptr, quats_raw, nullptr, nullptr, nullptr, nullptr, 3, 3, (int64_t)K_rest * 3, cam_pos, K_rest, use_sh, pose,
                       pose ? pose + 4 : nullptr, 0};
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div64(N, NODE_PER_WAVE * (NODE_BLOCK / 64));
    const NodeGrads G{v_scales, v_quats, v_opacities, v_rgbs, v_means, g_scales_raw, g_quats_raw, g_opacities_raw, g_features_dc,
                      g_features_rest, g_features_dc_add, g_means, g_pose, n_traversals, traversal};
    MTGS_NODE_DISPATCH(node_bwd_kernel, N, P, scales, opacities, rgbs, clamp_mask, G)
    MTGS_CHECK_LAUNCH("mtgs_node_bwd");
    return MTGS_OK;
}

extern "C" int mtgs_node_desc_bytes(void) { return (int)sizeof(mtgs_node_desc); }

extern "C" int mtgs_node_fwd_batch(int n_nodes, const mtgs_node_desc *table, int64_t total_blocks, int degree,
                                   const float *cam_pos, int64_t *model_id, void *stream) {
    MTGS_REQUIRE(n_nodes >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_node_fwd_batch: bad sizes");
    MTGS_REQUIRE(degree >= -1 && degree <= 3, MTGS_EUNSUPPORTED, "mtgs_node_fwd_batch: degree %d (-1 .. 3)", degree);
    if (n_nodes == 0 || total_blocks == 0) return MTGS_OK;
    MTGS_REQUIRE(table && cam_pos, MTGS_EINVAL, "mtgs_node_fwd_batch: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)total_blocks;
    if (degree < 0) {   // geometry only: every descriptor has skip_colors = 1
        node_fwd_geom_batch_kernel<<<grid, NODE_BLOCK, 0, st>>>(table, n_nodes, model_id);
        MTGS_CHECK_LAUNCH("mtgs_node_fwd_batch");
        return MTGS_OK;
    }
    MTGS_NODE_DISPATCH(node_fwd_batch_kernel, table, n_nodes, cam_pos, model_id)
    MTGS_CHECK_LAUNCH("mtgs_node_fwd_batch");
    return MTGS_OK;
}

extern "C" int mtgs_node_bwd_batch(int n_nodes, const mtgs_node_desc *table, int64_t total_blocks, int degree,
                                   const float *cam_pos, void *stream) {
    MTGS_REQUIRE(n_nodes >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_node_bwd_batch: bad sizes");
    MTGS_REQUIRE(degree >= 0 && degree <= 3, MTGS_EUNSUPPORTED, "mtgs_node_bwd_batch: degree %d (<= 3)", degree);
    if (n_nodes == 0 || total_blocks == 0) return MTGS_OK;
    MTGS_REQUIRE(table && cam_pos, MTGS_EINVAL, "mtgs_node_bwd_batch: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)total_blocks;
    MTGS_NODE_DISPATCH(node_bwd_batch_kernel, table, n_nodes, cam_pos)
    node_pose_finalize_kernel<<<(unsigned)((n_nodes + 255) / 256), 256, 0, st>>>(table, n_nodes);
    MTGS_CHECK_LAUNCH("mtgs_node_bwd_batch");
    return MTGS_OK;
}

extern "C" int mtgs_node_bwd_rows(int n_nodes, const mtgs_node_desc *table, const int32_t *vis_ids, const int64_t *totals,
                                  int64_t cap_vis, const float *ws_rows, int ws_stride, float *param_rows, void *stream) {
    MTGS_REQUIRE(n_nodes > 0 && cap_vis >= 0 && ws_stride >= 12 && (ws_stride & 3) == 0, MTGS_EINVAL, "mtgs_node_bwd_rows: bad sizes");
    if (cap_vis == 0) return MTGS_OK;
    MTGS_REQUIRE(table && vis_ids && ws_rows && param_rows, MTGS_EINVAL, "mtgs_node_bwd_rows: null pointer");
    MTGS_REQUIRE(((reinterpret_cast<uintptr_t>(ws_rows) | reinterpret_cast<uintptr_t>(param_rows)) & 15) == 0, MTGS_EINVAL,
                 "mtgs_node_bwd_rows: rows must be 16-byte aligned");
    node_bwd_rows_kernel<<<(unsigned)ceil_div64(cap_vis, NODE_BLOCK), NODE_BLOCK, 0, (hipStream_t)stream>>>(table, n_nodes, vis_ids, totals,
                                                                                                       cap_vis, ws_rows, ws_stride, param_rows);
    MTGS_CHECK_LAUNCH("mtgs_node_bwd_rows");
    return MTGS_OK;
}
