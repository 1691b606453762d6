/*
 * gsplat_oracle.c -- CPU restatement of the gsplat-1.4.0 rasterization path used by MTGS.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mtgs_amd/ (the product) may import, link or execute
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and there
 * only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the algorithm lives in the third-party dependency gsplat==1.4.0
 * (/root/reference/requirements.txt:12), which is neither vendored under /root/reference nor
 * installable in the build container, and the reference holds no tests, golden vectors or
 * fixtures for this path.  This file therefore restates gsplat
 * 1.4.0's PUBLISHED algorithm (gsplat/cuda/csrc/{utils.cuh, fully_fused_projection_{fwd,bwd}.cu,
 * isect_tiles.cu (named intersect*.cu upstream), rasterize_to_pixels_{fwd,bwd}.cu,
 * spherical_harmonics_{fwd,bwd}.cu} and gsplat/cuda/_torch_impl.py), anchored on the reference's
 * call sites:
 *   rasterization(**gsplat_kwargs)      /root/reference/mtgs/scene_model/mtgs_scene_graph.py:641-662
 *   spherical_harmonics(n, dirs, coefs) /root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:312-318
 * and is pinned by closed-form known-answer tests, fp64 autograd of an independent torch
 * restatement (oracle/torch_ref.py) and finite differences (tests/test_oracle_*.py).
 *
 * Arithmetic: per-element math in IEEE fp32 with a fixed operation order (build with
 * -ffp-contract=off; see Makefile); gradient SUMS over pixels / cameras are accumulated in fp64
 * and rounded once, so the oracle is the order-independent value the device's atomics approximate.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- named constants (each has a test in tests/test_oracle_known_answers.py) ---------------- */
#define ALPHA_MAX 0.999f          /* rasterize_to_pixels_fwd: alpha = min(0.999f, opac * vis)          */
#define ALPHA_MIN (1.0f / 255.0f) /* rasterize_to_pixels_fwd: skip if alpha < 1/255                    */
#define T_MIN 1e-4f               /* rasterize_to_pixels_fwd: stop when next_T <= 1e-4 (excl. this one) */
#define RADIUS_FLOOR 0.01f        /* fully_fused_projection_fwd: sqrt(max(0.01f, b*b - det))           */
#define RADIUS_SIGMA 3.0f         /* fully_fused_projection_fwd: ceil(3 * sqrt(lambda_max))            */
#define FOV_MARGIN 0.3f           /* utils.cuh persp_proj: lim = .. + 0.3 * tan_fov                    */
#define COMP_EPS 1e-6f            /* utils.cuh add_blur_vjp: 0.5 / (compensation + 1e-6)               */

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ============================ spherical harmonics ============================================ */
/* gsplat spherical_harmonics_fwd.cu sh_coeffs_to_color_fast / _torch_impl._eval_sh_bases_fast.
 * Evaluates the (degree+1)^2 real SH basis values b[] at unit direction (x,y,z). */
static void sh_bases(int degree, float x, float y, float z, float *b) {
    b[0] = 0.2820947917738781f;
    if (degree < 1) return;
    b[1] = -0.48860251190292f * y;
    b[2] = 0.48860251190292f * z;
    b[3] = -0.48860251190292f * x;
    if (degree < 2) return;
    float z2 = z * z;
    float fTmp0B = -1.092548430592079f * z;
    float fC1 = x * x - y * y;
    float fS1 = 2.f * x * y;
    b[4] = 0.5462742152960395f * fS1;
    b[5] = fTmp0B * y;
    b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    b[7] = fTmp0B * x;
    b[8] = 0.5462742152960395f * fC1;
    if (degree < 3) return;
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    float fTmp1B = 1.445305721320277f * z;
    float fC2 = x * fC1 - y * fS1;
    float fS2 = x * fS1 + y * fC1;
    b[9] = -0.5900435899266435f * fS2;
    b[10] = fTmp1B * fS1;
    b[11] = fTmp0C * y;
    b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
    b[13] = fTmp0C * x;
    b[14] = fTmp1B * fC1;
    b[15] = -0.5900435899266435f * fC2;
    if (degree < 4) return;
    float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    float fTmp2B = -1.770130769779931f * z;
    float fC3 = x * fC2 - y * fS2;
    float fS3 = x * fS2 + y * fC2;
    b[16] = 0.6258357354491763f * fS3;
    b[17] = fTmp2B * fS2;
    b[18] = fTmp1C * fS1;
    b[19] = fTmp0D * y;
    b[20] = 1.984313483298443f * z * b[12] - 1.006230589874905f * b[6];
    b[21] = fTmp0D * x;
    b[22] = fTmp1C * fC1;
    b[23] = fTmp2B * fC2;
    b[24] = 0.6258357354491763f * fC3;
}

/* d(bases)/d(x,y,z) at unit direction, same recurrences differentiated term by term. */
static void sh_bases_grad(int degree, float x, float y, float z, float *dx, float *dy, float *dz) {
    int nb = (degree + 1) * (degree + 1);
    for (int i = 0; i < nb; ++i) dx[i] = dy[i] = dz[i] = 0.f;
    if (degree < 1) return;
    dy[1] = -0.48860251190292f;
    dz[2] = 0.48860251190292f;
    dx[3] = -0.48860251190292f;
    if (degree < 2) return;
    float z2 = z * z;
    float fTmp0B = -1.092548430592079f * z, fTmp0B_z = -1.092548430592079f;
    float fC1 = x * x - y * y, fC1_x = 2.f * x, fC1_y = -2.f * y;
    float fS1 = 2.f * x * y, fS1_x = 2.f * y, fS1_y = 2.f * x;
    dx[4] = 0.5462742152960395f * fS1_x; dy[4] = 0.5462742152960395f * fS1_y;
    dy[5] = fTmp0B; dz[5] = fTmp0B_z * y;
    dz[6] = 2.f * 0.9461746957575601f * z;
    dx[7] = fTmp0B; dz[7] = fTmp0B_z * x;
    dx[8] = 0.5462742152960395f * fC1_x; dy[8] = 0.5462742152960395f * fC1_y;
    if (degree < 3) return;
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f, fTmp0C_z = -2.285228997322329f * 2.f * z;
    float fTmp1B = 1.445305721320277f * z, fTmp1B_z = 1.445305721320277f;
    float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    float fC2_x = fC1 + x * fC1_x - y * fS1_x, fC2_y = x * fC1_y - fS1 - y * fS1_y;
    float fS2_x = fS1 + x * fS1_x + y * fC1_x, fS2_y = x * fS1_y + fC1 + y * fC1_y;
    dx[9] = -0.5900435899266435f * fS2_x; dy[9] = -0.5900435899266435f * fS2_y;
    dx[10] = fTmp1B * fS1_x; dy[10] = fTmp1B * fS1_y; dz[10] = fTmp1B_z * fS1;
    dy[11] = fTmp0C; dz[11] = fTmp0C_z * y;
    float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    float pSH12_z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
    dz[12] = pSH12_z;
    dx[13] = fTmp0C; dz[13] = fTmp0C_z * x;
    dx[14] = fTmp1B * fC1_x; dy[14] = fTmp1B * fC1_y; dz[14] = fTmp1B_z * fC1;
    dx[15] = -0.5900435899266435f * fC2_x; dy[15] = -0.5900435899266435f * fC2_y;
    if (degree < 4) return;
    float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float fTmp0D_z = 3.f * -4.683325804901025f * z2 + 2.007139630671868f;
    float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f, fTmp1C_z = 2.f * 3.31161143515146f * z;
    float fTmp2B = -1.770130769779931f * z, fTmp2B_z = -1.770130769779931f;
    float fC3_x = fC2 + x * fC2_x - y * fS2_x, fC3_y = x * fC2_y - fS2 - y * fS2_y;
    float fS3_x = fS2 + x * fS2_x + y * fC2_x, fS3_y = x * fS2_y + fC2 + y * fC2_y;
    float pSH6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
    dx[16] = 0.6258357354491763f * fS3_x; dy[16] = 0.6258357354491763f * fS3_y;
    dx[17] = fTmp2B * fS2_x; dy[17] = fTmp2B * fS2_y; dz[17] = fTmp2B_z * fS2;
    dx[18] = fTmp1C * fS1_x; dy[18] = fTmp1C * fS1_y; dz[18] = fTmp1C_z * fS1;
    dy[19] = fTmp0D; dz[19] = fTmp0D_z * y;
    dz[20] = 1.984313483298443f * (pSH12 + z * pSH12_z) - 1.006230589874905f * dz[6];
    (void)pSH6;
    dx[21] = fTmp0D; dz[21] = fTmp0D_z * x;
    dx[22] = fTmp1C * fC1_x; dy[22] = fTmp1C * fC1_y; dz[22] = fTmp1C_z * fC1;
    dx[23] = fTmp2B * fC2_x; dy[23] = fTmp2B * fC2_y; dz[23] = fTmp2B_z * fC2;
    dx[24] = 0.6258357354491763f * fC3_x; dy[24] = 0.6258357354491763f * fC3_y;
}

/* compute_sh_fwd: colors[n,3] = sum_k basis_k(normalize(dir)) * coeffs[n,k,:]; masks nullable. */
void orc_sh_fwd(int64_t n, int K, int degree, const float *dirs, const float *coeffs,
                const uint8_t *masks, float *colors) {
    int nb = (degree + 1) * (degree + 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float *o = colors + i * 3;
        if (masks && !masks[i]) { o[0] = o[1] = o[2] = 0.f; continue; }
        float x = dirs[i * 3], y = dirs[i * 3 + 1], z = dirs[i * 3 + 2];
        float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
        x *= inorm; y *= inorm; z *= inorm;
        float b[25];
        sh_bases(degree, x, y, z, b);
        const float *c = coeffs + i * (int64_t)K * 3;
        for (int ch = 0; ch < 3; ++ch) {
            float acc = 0.f;
            for (int k = 0; k < nb; ++k) acc += b[k] * c[k * 3 + ch];
            o[ch] = acc;
        }
    }
}

/* compute_sh_bwd: v_coeffs[n,K,3] (zero above the active degree); v_dirs[n,3] nullable, includes
 * the VJP of the normalisation (v_dir = (v_d - (v_d.d) d) / |dir|). */
void orc_sh_bwd(int64_t n, int K, int degree, const float *dirs, const float *coeffs,
                const uint8_t *masks, const float *v_colors, float *v_coeffs, float *v_dirs) {
    int nb = (degree + 1) * (degree + 1);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float *vc = v_coeffs + i * (int64_t)K * 3;
        for (int k = 0; k < K * 3; ++k) vc[k] = 0.f;
        if (v_dirs) v_dirs[i * 3] = v_dirs[i * 3 + 1] = v_dirs[i * 3 + 2] = 0.f;
        if (masks && !masks[i]) continue;
        float x = dirs[i * 3], y = dirs[i * 3 + 1], z = dirs[i * 3 + 2];
        float inorm = 1.0f / sqrtf((x * x + y * y) + z * z);
        x *= inorm; y *= inorm; z *= inorm;
        float b[25];
        sh_bases(degree, x, y, z, b);
        const float *vo = v_colors + i * 3;
        for (int k = 0; k < nb; ++k)
            for (int ch = 0; ch < 3; ++ch) vc[k * 3 + ch] = b[k] * vo[ch];
        if (v_dirs) {
            float dx[25], dy[25], dz[25];
            sh_bases_grad(degree, x, y, z, dx, dy, dz);
            const float *c = coeffs + i * (int64_t)K * 3;
            float vx = 0.f, vy = 0.f, vz = 0.f;
            for (int k = 0; k < nb; ++k) {
                float s = (c[k * 3] * vo[0] + c[k * 3 + 1] * vo[1]) + c[k * 3 + 2] * vo[2];
                vx += dx[k] * s; vy += dy[k] * s; vz += dz[k] * s;
            }
            float dot = (vx * x + vy * y) + vz * z;
            v_dirs[i * 3] = (vx - dot * x) * inorm;
            v_dirs[i * 3 + 1] = (vy - dot * y) * inorm;
            v_dirs[i * 3 + 2] = (vz - dot * z) * inorm;
        }
    }
}

/* ============================ projection ====================================================== */
/* 3x3 row-major helpers with a fixed summation order: (a0*b0 + a1*b1) + a2*b2 */
static inline void mm3(const float *A, const float *B, float *C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j];
}
static inline void mm3_bt(const float *A, const float *B, float *C) { /* C = A * B^T */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1]) + A[i * 3 + 2] * B[j * 3 + 2];
}
static inline void mm3_at(const float *A, const float *B, float *C) { /* C = A^T * B */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i] * B[j] + A[3 + i] * B[3 + j]) + A[6 + i] * B[6 + j];
}

/* utils.cuh quat_to_rotmat: wxyz, normalised inside. Row-major R. Returns 1/|q| in *inv_norm. */
static inline void quat_to_rotmat(const float *q, float *R, float *qn, float *inv_norm) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float inv = 1.0f / sqrtf(((x * x + y * y) + z * z) + w * w);
    w *= inv; x *= inv; y *= inv; z *= inv;
    float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (y2 + z2); R[1] = 2.f * (xy - wz); R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz); R[4] = 1.f - 2.f * (x2 + z2); R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy); R[7] = 2.f * (yz + wx); R[8] = 1.f - 2.f * (x2 + y2);
    if (qn) { qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z; }
    if (inv_norm) *inv_norm = inv;
}

typedef struct {
    float mean_c[3];
    float Rq[9], Mq[9]; /* quaternion rotation, Rq*diag(s) */
    float covar[9], covar_c[9];
    float J[6]; /* 2x3 row-major */
    float rz, rz2, tx, ty;
    int x_clamped, y_clamped;
    float cov2d[4]; /* before blur: [0]=xx [1]=xy [2]=yx [3]=yy */
    float qn[4], inv_norm;
} proj_state;

/* Shared front half of fully_fused_projection_{fwd,bwd}: world->cam, covariance, persp_proj. */
static void proj_common(const float *mean, const float *quat, const float *scale, const float *vm,
                        const float *Kmat, int W, int H, proj_state *s) {
    float R[9] = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
    float t[3] = {vm[3], vm[7], vm[11]};
    for (int i = 0; i < 3; ++i) /* utils.cuh pos_world_to_cam */
        s->mean_c[i] = ((R[i * 3] * mean[0] + R[i * 3 + 1] * mean[1]) + R[i * 3 + 2] * mean[2]) + t[i];
    quat_to_rotmat(quat, s->Rq, s->qn, &s->inv_norm); /* utils.cuh quat_scale_to_covar_preci */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) s->Mq[i * 3 + j] = s->Rq[i * 3 + j] * scale[j];
    mm3_bt(s->Mq, s->Mq, s->covar);
    float tmp[9];
    mm3(R, s->covar, tmp); /* utils.cuh covar_world_to_cam: R * covar * R^T */
    mm3_bt(tmp, R, s->covar_c);
    /* utils.cuh persp_proj */
    float fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
    float x = s->mean_c[0], y = s->mean_c[1], z = s->mean_c[2];
    float tan_fovx = 0.5f * (float)W / fx, tan_fovy = 0.5f * (float)H / fy;
    float lim_x_pos = ((float)W - cx) / fx + FOV_MARGIN * tan_fovx;
    float lim_x_neg = cx / fx + FOV_MARGIN * tan_fovx;
    float lim_y_pos = ((float)H - cy) / fy + FOV_MARGIN * tan_fovy;
    float lim_y_neg = cy / fy + FOV_MARGIN * tan_fovy;
    float rz = 1.0f / z, rz2 = rz * rz;
    float xz = x * rz, yz = y * rz;
    s->x_clamped = !(xz <= lim_x_pos && xz >= -lim_x_neg);
    s->y_clamped = !(yz <= lim_y_pos && yz >= -lim_y_neg);
    float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, xz));
    float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, yz));
    s->rz = rz; s->rz2 = rz2; s->tx = tx; s->ty = ty;
    float *J = s->J;
    J[0] = fx * rz; J[1] = 0.f; J[2] = -fx * tx * rz2;
    J[3] = 0.f; J[4] = fy * rz; J[5] = -fy * ty * rz2;
    /* cov2d = J * covar_c * J^T */
    float B[6];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            B[i * 3 + j] = (J[i * 3] * s->covar_c[j] + J[i * 3 + 1] * s->covar_c[3 + j]) + J[i * 3 + 2] * s->covar_c[6 + j];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            s->cov2d[i * 2 + j] = (B[i * 3] * J[j * 3] + B[i * 3 + 1] * J[j * 3 + 1]) + B[i * 3 + 2] * J[j * 3 + 2];
}

/* fully_fused_projection_fwd (pinhole, packed=False).  compensations nullable. */
void orc_project_fwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                     const float *viewmats, const float *Ks, int W, int H, float eps2d,
                     float near_plane, float far_plane, float radius_clip, int32_t *radii,
                     float *means2d, float *depths, float *conics, float *compensations) {
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < (int64_t)C * N; ++idx) {
        int c = (int)(idx / N);
        int64_t n = idx % N;
        const float *vm = viewmats + c * 16, *Kmat = Ks + c * 9;
        radii[idx] = 0;
        means2d[idx * 2] = means2d[idx * 2 + 1] = 0.f;
        depths[idx] = 0.f;
        conics[idx * 3] = conics[idx * 3 + 1] = conics[idx * 3 + 2] = 0.f;
        if (compensations) compensations[idx] = 0.f;
        /* near/far test needs only z */
        const float *m = means + n * 3;
        float zc = ((vm[8] * m[0] + vm[9] * m[1]) + vm[10] * m[2]) + vm[11];
        if (zc < near_plane || zc > far_plane) continue;
        proj_state s;
        proj_common(m, quats + n * 4, scales + n * 3, vm, Kmat, W, H, &s);
        float fx = Kmat[0], fy = Kmat[4], cx = Kmat[2], cy = Kmat[5];
        float mx = fx * s.mean_c[0] * s.rz + cx, my = fy * s.mean_c[1] * s.rz + cy;
        /* utils.cuh add_blur */
        float c00 = s.cov2d[0], c01 = s.cov2d[1], c11 = s.cov2d[3];
        float det_orig = c00 * c11 - c01 * c01;
        c00 += eps2d; c11 += eps2d;
        float det = c00 * c11 - c01 * c01;
        float comp = sqrtf(fmaxf(0.f, det_orig / det));
        if (!(det > 0.f)) continue;
        /* inverse */
        float idet = 1.0f / det;
        float ca = c11 * idet, cb = -c01 * idet, cc = c00 * idet;
        /* 3-sigma radius */
        float b = 0.5f * (c00 + c11);
        float v1 = b + sqrtf(fmaxf(RADIUS_FLOOR, b * b - det));
        float radius = ceilf(RADIUS_SIGMA * sqrtf(v1));
        if (radius <= radius_clip) continue;
        if (mx + radius <= 0.f || mx - radius >= (float)W || my + radius <= 0.f || my - radius >= (float)H) continue;
        radii[idx] = (int32_t)radius;
        means2d[idx * 2] = mx; means2d[idx * 2 + 1] = my;
        depths[idx] = s.mean_c[2];
        conics[idx * 3] = ca; conics[idx * 3 + 1] = cb; conics[idx * 3 + 2] = cc;
        if (compensations) compensations[idx] = comp;
    }
}

/* fully_fused_projection_bwd.  Outputs are overwritten; camera sums in fp64. v_viewmats nullable,
 * v_compensations/compensations nullable (together). */
void orc_project_bwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                     const float *viewmats, const float *Ks, int W, int H, float eps2d,
                     const int32_t *radii, const float *conics, const float *compensations,
                     const float *v_means2d, const float *v_depths, const float *v_conics,
                     const float *v_compensations, float *v_means, float *v_quats, float *v_scales,
                     float *v_viewmats) {
    double *vvm = (double *)calloc((size_t)C * 12, sizeof(double)); /* per camera: v_R[9], v_t[3] */
#pragma omp parallel
    {
        double *loc = (double *)calloc((size_t)C * 12, sizeof(double));
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            double am[3] = {0, 0, 0}, aq[4] = {0, 0, 0, 0}, as[3] = {0, 0, 0};
            for (int c = 0; c < C; ++c) {
                int64_t idx = (int64_t)c * N + n;
                if (radii[idx] <= 0) continue;
                const float *vm = viewmats + c * 16, *Kmat = Ks + c * 9;
                float R[9] = {vm[0], vm[1], vm[2], vm[4], vm[5], vm[6], vm[8], vm[9], vm[10]};
                proj_state s;
                proj_common(means + n * 3, quats + n * 4, scales + n * 3, vm, Kmat, W, H, &s);
                float fx = Kmat[0], fy = Kmat[4];
                /* vjp of conic = inverse(cov2d_blur): v_cov = -conic * V * conic, V symmetric
                 * with the off-diagonal gradient split in two (fully_fused_projection_bwd.cu). */
                float a = conics[idx * 3], b = conics[idx * 3 + 1], cc = conics[idx * 3 + 2];
                float va = v_conics[idx * 3], vb = 0.5f * v_conics[idx * 3 + 1], vc = v_conics[idx * 3 + 2];
                /* T = conic * V */
                float t00 = a * va + b * vb, t01 = a * vb + b * vc, t10 = b * va + cc * vb, t11 = b * vb + cc * vc;
                float vcov[4];
                vcov[0] = -(t00 * a + t01 * b); vcov[1] = -(t00 * b + t01 * cc);
                vcov[2] = -(t10 * a + t11 * b); vcov[3] = -(t10 * b + t11 * cc);
                if (v_compensations) { /* utils.cuh add_blur_vjp */
                    float comp = compensations[idx], vcomp = v_compensations[idx];
                    float det_conic = a * cc - b * b;
                    float v_sqr = vcomp * 0.5f / (comp + COMP_EPS);
                    float omc = 1.f - comp * comp;
                    vcov[0] += v_sqr * (omc * a - eps2d * det_conic);
                    vcov[1] += v_sqr * (omc * b);
                    vcov[2] += v_sqr * (omc * b);
                    vcov[3] += v_sqr * (omc * cc - eps2d * det_conic);
                }
                /* utils.cuh persp_proj_vjp */
                const float *J = s.J;
                float x = s.mean_c[0], y = s.mean_c[1];
                float rz = s.rz, rz2 = s.rz2, rz3 = rz2 * rz, tx = s.tx, ty = s.ty;
                float vmx = v_means2d[idx * 2], vmy = v_means2d[idx * 2 + 1];
                /* v_covar_c = J^T vcov J */
                float G[6]; /* vcov * J : 2x3 */
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 3; ++j) G[i * 3 + j] = vcov[i * 2] * J[j] + vcov[i * 2 + 1] * J[3 + j];
                float v_covar_c[9];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) v_covar_c[i * 3 + j] = J[i] * G[j] + J[3 + i] * G[3 + j];
                float v_mean_c[3];
                v_mean_c[0] = fx * rz * vmx;
                v_mean_c[1] = fy * rz * vmy;
                v_mean_c[2] = -(fx * x * vmx + fy * y * vmy) * rz2;
                /* v_J = vcov * J * covar_c^T + vcov^T * J * covar_c */
                float vcovT[4] = {vcov[0], vcov[2], vcov[1], vcov[3]};
                float G2[6];
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 3; ++j) G2[i * 3 + j] = vcovT[i * 2] * J[j] + vcovT[i * 2 + 1] * J[3 + j];
                float vJ[6];
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 3; ++j) {
                        float p = (G[i * 3] * s.covar_c[j * 3] + G[i * 3 + 1] * s.covar_c[j * 3 + 1]) + G[i * 3 + 2] * s.covar_c[j * 3 + 2];
                        float q = (G2[i * 3] * s.covar_c[j] + G2[i * 3 + 1] * s.covar_c[3 + j]) + G2[i * 3 + 2] * s.covar_c[6 + j];
                        vJ[i * 3 + j] = p + q;
                    }
                if (!s.x_clamped) v_mean_c[0] += -fx * rz2 * vJ[2];
                else v_mean_c[2] += -fx * rz3 * vJ[2] * tx;
                if (!s.y_clamped) v_mean_c[1] += -fy * rz2 * vJ[5];
                else v_mean_c[2] += -fy * rz3 * vJ[5] * ty;
                v_mean_c[2] += ((-fx * rz2 * vJ[0] - fy * rz2 * vJ[4]) + 2.f * fx * tx * rz3 * vJ[2]) + 2.f * fy * ty * rz3 * vJ[5];
                v_mean_c[2] += v_depths[idx];
                /* utils.cuh pos_world_to_cam_vjp / covar_world_to_cam_vjp */
                const float *m = means + n * 3;
                float vR[9], vt[3], vmean[3];
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) vR[i * 3 + j] = v_mean_c[i] * m[j];
                    vt[i] = v_mean_c[i];
                    vmean[i] = (R[i] * v_mean_c[0] + R[3 + i] * v_mean_c[1]) + R[6 + i] * v_mean_c[2];
                }
                /* v_R += v_covar_c * R * covar^T + v_covar_c^T * R * covar ; v_covar = R^T v_covar_c R */
                float RC[9], RCt[9], tmp[9], tmp2[9], vcT[9];
                mm3(R, s.covar, RC);     /* R * covar   */
                mm3_bt(R, s.covar, RCt); /* R * covar^T */
                mm3(v_covar_c, RCt, tmp);
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) vcT[i * 3 + j] = v_covar_c[j * 3 + i];
                mm3(vcT, RC, tmp2);
                for (int i = 0; i < 9; ++i) vR[i] += tmp[i] + tmp2[i];
                float v_covar[9];
                mm3_at(R, v_covar_c, tmp);
                mm3(tmp, R, v_covar);
                /* utils.cuh quat_scale_to_covar_vjp: v_M = (v_covar + v_covar^T) M */
                float sym[9], vM[9];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) sym[i * 3 + j] = v_covar[i * 3 + j] + v_covar[j * 3 + i];
                mm3(sym, s.Mq, vM);
                const float *sc = scales + n * 3;
                float Gq[9], vs[3];
                for (int j = 0; j < 3; ++j) {
                    vs[j] = (s.Rq[j] * vM[j] + s.Rq[3 + j] * vM[3 + j]) + s.Rq[6 + j] * vM[6 + j];
                    for (int i = 0; i < 3; ++i) Gq[i * 3 + j] = vM[i * 3 + j] * sc[j];
                }
                /* utils.cuh quat_to_rotmat_vjp (row-major G) */
                float w = s.qn[0], qx = s.qn[1], qy = s.qn[2], qz = s.qn[3];
                float vqn[4];
                vqn[0] = 2.f * ((qx * (Gq[7] - Gq[5]) + qy * (Gq[2] - Gq[6])) + qz * (Gq[3] - Gq[1]));
                vqn[1] = 2.f * (((-2.f * qx * (Gq[4] + Gq[8]) + qy * (Gq[1] + Gq[3])) + qz * (Gq[2] + Gq[6])) + w * (Gq[7] - Gq[5]));
                vqn[2] = 2.f * (((qx * (Gq[1] + Gq[3]) - 2.f * qy * (Gq[0] + Gq[8])) + qz * (Gq[5] + Gq[7])) + w * (Gq[2] - Gq[6]));
                vqn[3] = 2.f * (((qx * (Gq[2] + Gq[6]) + qy * (Gq[5] + Gq[7])) - 2.f * qz * (Gq[0] + Gq[4])) + w * (Gq[3] - Gq[1]));
                float dot = ((vqn[0] * w + vqn[1] * qx) + vqn[2] * qy) + vqn[3] * qz;
                for (int k = 0; k < 4; ++k) aq[k] += (double)((vqn[k] - dot * s.qn[k]) * s.inv_norm);
                for (int k = 0; k < 3; ++k) { am[k] += (double)vmean[k]; as[k] += (double)vs[k]; }
                for (int k = 0; k < 9; ++k) loc[c * 12 + k] += (double)vR[k];
                for (int k = 0; k < 3; ++k) loc[c * 12 + 9 + k] += (double)vt[k];
            }
            for (int k = 0; k < 3; ++k) { v_means[n * 3 + k] = (float)am[k]; v_scales[n * 3 + k] = (float)as[k]; }
            for (int k = 0; k < 4; ++k) v_quats[n * 4 + k] = (float)aq[k];
        }
#pragma omp critical
        for (int k = 0; k < C * 12; ++k) vvm[k] += loc[k];
        free(loc);
    }
    if (v_viewmats) {
        for (int c = 0; c < C; ++c) {
            float *o = v_viewmats + c * 16;
            for (int i = 0; i < 16; ++i) o[i] = 0.f;
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) o[i * 4 + j] = (float)vvm[c * 12 + i * 3 + j];
                o[i * 4 + 3] = (float)vvm[c * 12 + 9 + i];
            }
        }
    }
    free(vvm);
}

/* ============================ tile intersection ============================================== */
static inline int bit_length_u32(uint32_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }
/* isect_tiles: tile_n_bits = floor(log2(n_tiles)) + 1, cam_n_bits = floor(log2(C)) + 1 */
int orc_tile_bits(int n_tiles) { return bit_length_u32((uint32_t)n_tiles); }
int orc_cam_bits(int C) { return bit_length_u32((uint32_t)C); }

static inline void tile_rect(float mx, float my, int32_t radius, int tile_size, int tw, int th,
                             int *x0, int *y0, int *x1, int *y1) {
    float ts = (float)tile_size;
    float tr = (float)radius / ts, tx = mx / ts, ty = my / ts;
    /* (uint32_t) cast of a negative float saturates to 0 on the device; min() clamps the top */
    *x0 = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)tw);
    *y0 = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)th);
    *x1 = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)tw);
    *y1 = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)th);
}

/* isect_tiles pass 1 + cumsum: tiles_per_gauss[C*N] i32, cum[C*N] i64 inclusive. Returns M. */
int64_t orc_isect_count(int C, int64_t N, const float *means2d, const int32_t *radii, int tile_size,
                        int tw, int th, int32_t *tiles_per_gauss, int64_t *cum) {
    int64_t total = 0;
    for (int64_t idx = 0; idx < (int64_t)C * N; ++idx) {
        int32_t cnt = 0;
        if (radii[idx] > 0) {
            int x0, y0, x1, y1;
            tile_rect(means2d[idx * 2], means2d[idx * 2 + 1], radii[idx], tile_size, tw, th, &x0, &y0, &x1, &y1);
            cnt = (x1 - x0) * (y1 - y0);
        }
        tiles_per_gauss[idx] = cnt;
        total += cnt;
        if (cum) cum[idx] = total;
    }
    return total;
}

/* isect_tiles pass 2 */
void orc_isect_emit(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                    const int64_t *cum, int tile_size, int tw, int th, int64_t *isect_ids,
                    int32_t *flatten_ids) {
    int tile_bits = orc_tile_bits(tw * th);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t idx = 0; idx < (int64_t)C * N; ++idx) {
        if (radii[idx] <= 0) continue;
        int x0, y0, x1, y1;
        tile_rect(means2d[idx * 2], means2d[idx * 2 + 1], radii[idx], tile_size, tw, th, &x0, &y0, &x1, &y1);
        int64_t cur = idx == 0 ? 0 : cum[idx - 1];
        int64_t cid = idx / N;
        uint32_t dbits;
        memcpy(&dbits, depths + idx, 4);
        int64_t hi = cid << (32 + tile_bits);
        for (int i = y0; i < y1; ++i)
            for (int j = x0; j < x1; ++j) {
                int64_t tile_id = (int64_t)i * tw + j;
                isect_ids[cur] = hi | (tile_id << 32) | (int64_t)dbits;
                flatten_ids[cur] = (int32_t)idx;
                ++cur;
            }
    }
}

/* cub::DeviceRadixSort::SortPairs restated: STABLE sort on key bits [0,key_bits). */
typedef struct { int64_t k; int32_t v; } kv_t;
static void merge_sort_kv(kv_t *a, kv_t *tmp, int64_t n, int64_t mask) {
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) tmp[o++] = ((a[j].k & mask) < (a[i].k & mask)) ? a[j++] : a[i++];
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(kv_t));
    }
}
void orc_sort_pairs(int64_t M, int key_bits, int64_t *keys, int32_t *vals) {
    if (M <= 1) return;
    kv_t *a = (kv_t *)malloc((size_t)M * sizeof(kv_t)), *t = (kv_t *)malloc((size_t)M * sizeof(kv_t));
    for (int64_t i = 0; i < M; ++i) { a[i].k = keys[i]; a[i].v = vals[i]; }
    int64_t mask = key_bits >= 64 ? (int64_t)-1 : (((int64_t)1 << key_bits) - 1);
    merge_sort_kv(a, t, M, mask);
    for (int64_t i = 0; i < M; ++i) { keys[i] = a[i].k; vals[i] = a[i].v; }
    free(a); free(t);
}

/* isect_offset_encode */
void orc_isect_offsets(int64_t M, const int64_t *ids_sorted, int C, int tw, int th, int32_t *offsets) {
    int n_tiles = tw * th;
    int tile_bits = orc_tile_bits(n_tiles);
    int64_t total = (int64_t)C * n_tiles;
    int64_t next = 0; /* next (cam,tile) slot whose offset is not yet written */
    for (int64_t i = 0; i < M; ++i) {
        int64_t hi = ids_sorted[i] >> 32;
        int64_t cid = hi >> tile_bits, tid = hi & (((int64_t)1 << tile_bits) - 1);
        int64_t slot = cid * n_tiles + tid;
        while (next <= slot && next < total) offsets[next++] = (int32_t)i;
    }
    while (next < total) offsets[next++] = (int32_t)M;
}

/* ============================ compositing ==================================================== */
/* rasterize_to_pixels_fwd.  render[C,H,W,D], alphas[C,H,W], last_ids[C,H,W].
 * critical[C,H,W] (nullable, test aid): 1 where some Gaussian of the pixel's list sits within a
 * relative 1e-4 of one of the algorithm's two DISCONTINUITIES (alpha == 1/255 skip threshold,
 * next_T == 1e-4 stop threshold).  There an fp32-rounding-level change of alpha flips a discrete
 * decision and moves the pixel by up to alpha_min * |colour|, so comparisons against another
 * fp32 implementation (different exp, FMA contraction) are ill-conditioned at exactly those pixels. */
/* _ex: plus critical_gauss[C*N] (nullable, zero-initialised by the caller, test aid): 1 for a Gaussian that ITSELF sits at
 * one of the two thresholds at some pixel -- its own contribution at that pixel is what a flipped decision adds or removes. */
void orc_blend_fwd_ex(int C, int64_t N, int D, const float *means2d, const float *conics,
                      const float *colors, const float *opacities, const float *backgrounds, int W, int H,
                      int tile_size, int tw, int th, const int32_t *offsets, const int32_t *flatten_ids,
                      int64_t M, float *render, float *alphas, int32_t *last_ids, uint8_t *critical,
                      uint8_t *critical_gauss) {
    (void)N;
    int64_t n_tiles = (int64_t)C * tw * th;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t t = 0; t < n_tiles; ++t) {
        int c = (int)(t / ((int64_t)tw * th));
        int ty = (int)((t / tw) % th), tx = (int)(t % tw);
        int64_t start = offsets[t], end = (t == n_tiles - 1) ? M : offsets[t + 1];
        for (int py = ty * tile_size; py < (ty + 1) * tile_size && py < H; ++py)
            for (int px = tx * tile_size; px < (tx + 1) * tile_size && px < W; ++px) {
                float fxp = (float)px + 0.5f, fyp = (float)py + 0.5f;
                float T = 1.f;
                float acc[64];
                for (int k = 0; k < D; ++k) acc[k] = 0.f;
                int32_t last = 0;
                uint8_t crit = 0;
                for (int64_t i = start; i < end; ++i) {
                    int32_t g = flatten_ids[i];
                    float dx = means2d[g * 2] - fxp, dy = means2d[g * 2 + 1] - fyp;
                    float a = conics[g * 3], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
                    float sigma = 0.5f * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                    float alpha = fminf(ALPHA_MAX, opacities[g] * expf(-sigma));
                    if (fabsf(alpha - ALPHA_MIN) <= 1e-4f * ALPHA_MIN && sigma >= -1e-6f) { crit = 1; if (critical_gauss) critical_gauss[g] = 1; }
                    if (fabsf(sigma) <= 1e-6f && alpha >= ALPHA_MIN) { crit = 1; if (critical_gauss) critical_gauss[g] = 1; }
                    if (sigma < 0.f || alpha < ALPHA_MIN) continue;
                    float next_T = T * (1.f - alpha);
                    if (fabsf(next_T - T_MIN) <= 1e-4f * T_MIN) { crit = 1; if (critical_gauss) critical_gauss[g] = 1; }
                    if (next_T <= T_MIN) break;
                    float vis = alpha * T;
                    const float *col = colors + (int64_t)g * D;
                    for (int k = 0; k < D; ++k) acc[k] += col[k] * vis;
                    last = (int32_t)i;
                    T = next_T;
                }
                int64_t pid = ((int64_t)c * H + py) * W + px;
                alphas[pid] = 1.f - T;
                for (int k = 0; k < D; ++k)
                    render[pid * D + k] = backgrounds ? acc[k] + T * backgrounds[c * D + k] : acc[k];
                last_ids[pid] = last;
                if (critical) critical[pid] = crit;
            }
    }
}

void orc_blend_fwd(int C, int64_t N, int D, const float *means2d, const float *conics,
                   const float *colors, const float *opacities, const float *backgrounds, int W, int H,
                   int tile_size, int tw, int th, const int32_t *offsets, const int32_t *flatten_ids,
                   int64_t M, float *render, float *alphas, int32_t *last_ids, uint8_t *critical) {
    orc_blend_fwd_ex(C, N, D, means2d, conics, colors, opacities, backgrounds, W, H, tile_size, tw, th, offsets, flatten_ids, M,
                     render, alphas, last_ids, critical, NULL);
}

/* rasterize_to_pixels_bwd.  Per-Gaussian sums in fp64, written (overwriting) at the end.
 * v_means2d_abs nullable. */
/* _ex: the same, plus term_abs[C,N,4+D] (nullable, test aid): per Gaussian the sums of the ABSOLUTE values of the
 * per-pixel terms of {conic 3, opacity 1, colour D} (for xy that sum is v_means2d_abs itself).  |sum| << sum of |terms|
 * marks a row whose terms cancel: its fp32-summed device value cannot be expected within a relative bound of the sum,
 * only within (per-term relative error) x (sum of |terms|) -- tests/util.py::assert_grad_close. */
/* Test aid of orc_blend_bwd_ex2: ONE pixel's forward and backward (the loops of orc_blend_fwd_ex / orc_blend_bwd_ex2) with at
 * most one discrete decision overridden -- ov_kind 1: the alpha >= 1/255 / sigma >= 0 decision of list position ov_pos
 * inverted, 2: its T <= 1e-4 stop decision inverted, 0: none.  terms[(end - start)][6 + D]: the SIGNED per-pixel gradient
 * terms {xy 2, conic 3, opacity 1, colour D} of every list position (zero where the Gaussian does not contribute). */
static void pixel_pass(int D, const float *means2d, const float *conics, const float *colors, const float *opacities,
                       const float *bg, float fxp, float fyp, int64_t start, int64_t end, const int32_t *flatten_ids,
                       const float *vr, float va_out, int64_t ov_pos, int ov_kind, float *terms) {
    int SM = 6 + D;
    int64_t L = end - start;
    memset(terms, 0, (size_t)L * SM * sizeof(float));
    uint8_t *inc = (uint8_t *)calloc((size_t)L, 1);
    float T = 1.f;
    int64_t last = start - 1;
    for (int64_t i = start; i < end; ++i) {
        int32_t g = flatten_ids[i];
        float dx = means2d[g * 2] - fxp, dy = means2d[g * 2 + 1] - fyp;
        float a = conics[g * 3], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
        float sigma = 0.5f * (a * dx * dx + cc * dy * dy) + b * dx * dy;
        float alpha = fminf(ALPHA_MAX, opacities[g] * expf(-sigma));
        int skip = sigma < 0.f || alpha < ALPHA_MIN;
        if (i == ov_pos && ov_kind == 1) skip = !skip;
        if (skip) continue;
        float next_T = T * (1.f - alpha);
        int stop = next_T <= T_MIN;
        if (i == ov_pos && ov_kind == 2) stop = !stop;
        if (stop) break;
        inc[i - start] = 1;
        last = i;
        T = next_T;
    }
    float T_final = T;
    float buffer[64];
    for (int k = 0; k < D; ++k) buffer[k] = 0.f;
    float bg_dot = 0.f;
    if (bg)
        for (int k = 0; k < D; ++k) bg_dot += bg[k] * vr[k];
    for (int64_t i = last; i >= start; --i) {
        if (!inc[i - start]) continue;
        int32_t g = flatten_ids[i];
        float dx = means2d[g * 2] - fxp, dy = means2d[g * 2 + 1] - fyp;
        float a = conics[g * 3], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
        float opac = opacities[g];
        float sigma = 0.5f * (a * dx * dx + cc * dy * dy) + b * dx * dy;
        float vis = expf(-sigma);
        float alpha = fminf(ALPHA_MAX, opac * vis);
        float ra = 1.0f / (1.0f - alpha);
        T *= ra;
        float fac = alpha * T;
        const float *col = colors + (int64_t)g * D;
        float v_alpha = 0.f;
        for (int k = 0; k < D; ++k) v_alpha += (col[k] * T - buffer[k] * ra) * vr[k];
        v_alpha += T_final * ra * va_out;
        if (bg) v_alpha += -T_final * ra * bg_dot;
        float *Q = terms + (i - start) * SM;
        if (opac * vis <= ALPHA_MAX) {
            float v_sigma = -opac * vis * v_alpha;
            Q[0] = v_sigma * (a * dx + b * dy);
            Q[1] = v_sigma * (b * dx + cc * dy);
            Q[2] = 0.5f * v_sigma * dx * dx;
            Q[3] = v_sigma * dx * dy;
            Q[4] = 0.5f * v_sigma * dy * dy;
            Q[5] = vis * v_alpha;
        }
        for (int k = 0; k < D; ++k) {
            Q[6 + k] = fac * vr[k];
            buffer[k] += col[k] * fac;
        }
    }
    free(inc);
}

/* _ex2: plus pixel_mask[C,H,W] / mask_terms[C,N,6+D] (both nullable, test aid): the FLIP SENSITIVITY of every gradient row.
 * For every pixel with pixel_mask != 0 (the threshold-critical pixels of orc_blend_fwd_ex) and every threshold-critical
 * decision on its list (alpha within 1e-4 of 1/255, sigma at 0, T (1 - alpha) within 1e-4 of 1e-4) the pixel is composited and
 * back-propagated ONCE MORE with that one decision inverted (pixel_pass), and per Gaussian of the list the absolute change of
 * its per-pixel terms {xy 2, conic 3, opacity 1, colour D} is summed.  That is how far a row can move when the discrete
 * decisions of its critical pixels fall the other way on another fp32 implementation (first order in the number of flips per
 * pixel) -- NOT the size of the row's own terms there: a flipped Gaussian behind j changes j's v_alpha through the
 * accumulated colour, whatever j's own term is.  The magnitude bound of the "flipped" / "self-critical" classes of
 * tests/util.py::assert_grad_close. */
void orc_blend_bwd_ex2(int C, int64_t N, int D, const float *means2d, const float *conics,
                       const float *colors, const float *opacities, const float *backgrounds, int W, int H,
                       int tile_size, int tw, int th, const int32_t *offsets, const int32_t *flatten_ids,
                       int64_t M, const float *alphas, const int32_t *last_ids, const float *v_render,
                       const float *v_alphas, float *v_means2d, float *v_means2d_abs, float *v_conics,
                       float *v_colors, float *v_opacities, float *term_abs, const uint8_t *pixel_mask, float *mask_terms) {
    int64_t CN = (int64_t)C * N;
    int S = 8 + D; /* xy(2) abs(2) conic(3) opac(1) colour(D) */
    int SA = 4 + D;
    int SM = 6 + D;
    double *acc = (double *)calloc((size_t)CN * S, sizeof(double));
    double *aab = term_abs ? (double *)calloc((size_t)CN * SA, sizeof(double)) : NULL;
    double *amk = (pixel_mask && mask_terms) ? (double *)calloc((size_t)CN * SM, sizeof(double)) : NULL;
    int64_t n_tiles = (int64_t)C * tw * th;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t t = 0; t < n_tiles; ++t) {
        int c = (int)(t / ((int64_t)tw * th));
        int ty = (int)((t / tw) % th), tx = (int)(t % tw);
        int64_t start = offsets[t], end = (t == n_tiles - 1) ? M : offsets[t + 1];
        if (end <= start) continue;
        for (int py = ty * tile_size; py < (ty + 1) * tile_size && py < H; ++py)
            for (int px = tx * tile_size; px < (tx + 1) * tile_size && px < W; ++px) {
                float fxp = (float)px + 0.5f, fyp = (float)py + 0.5f;
                int64_t pid = ((int64_t)c * H + py) * W + px;
                float T_final = 1.f - alphas[pid];
                float T = T_final;
                float buffer[64];
                for (int k = 0; k < D; ++k) buffer[k] = 0.f;
                const float *vr = v_render + pid * D;
                float va_out = v_alphas[pid];
                float bg_dot = 0.f;
                if (backgrounds)
                    for (int k = 0; k < D; ++k) bg_dot += backgrounds[c * D + k] * vr[k];
                int64_t bin_final = last_ids[pid];
                for (int64_t i = bin_final; i >= start; --i) {
                    int32_t g = flatten_ids[i];
                    float dx = means2d[g * 2] - fxp, dy = means2d[g * 2 + 1] - fyp;
                    float a = conics[g * 3], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
                    float opac = opacities[g];
                    float sigma = 0.5f * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                    float vis = expf(-sigma);
                    float alpha = fminf(ALPHA_MAX, opac * vis);
                    if (sigma < 0.f || alpha < ALPHA_MIN) continue;
                    float ra = 1.0f / (1.0f - alpha);
                    T *= ra;
                    float fac = alpha * T;
                    const float *col = colors + (int64_t)g * D;
                    double *A = acc + (int64_t)g * S;
                    float v_alpha = 0.f;
                    for (int k = 0; k < D; ++k) {
                        v_alpha += (col[k] * T - buffer[k] * ra) * vr[k];
                    }
                    v_alpha += T_final * ra * va_out;
                    if (backgrounds) v_alpha += -T_final * ra * bg_dot;
                    float vxy0 = 0.f, vxy1 = 0.f, vc0 = 0.f, vc1 = 0.f, vc2 = 0.f, vo = 0.f;
                    if (opac * vis <= ALPHA_MAX) {
                        float v_sigma = -opac * vis * v_alpha;
                        vc0 = 0.5f * v_sigma * dx * dx;
                        vc1 = v_sigma * dx * dy;
                        vc2 = 0.5f * v_sigma * dy * dy;
                        vxy0 = v_sigma * (a * dx + b * dy);
                        vxy1 = v_sigma * (b * dx + cc * dy);
                        vo = vis * v_alpha;
                    }
                    if (aab) {
                        double *B = aab + (int64_t)g * SA;
#pragma omp atomic
                        B[0] += (double)fabsf(vc0);
#pragma omp atomic
                        B[1] += (double)fabsf(vc1);
#pragma omp atomic
                        B[2] += (double)fabsf(vc2);
#pragma omp atomic
                        B[3] += (double)fabsf(vo);
                        for (int k = 0; k < D; ++k) {
#pragma omp atomic
                            B[4 + k] += (double)fabsf(fac * vr[k]);
                        }
                    }
                    for (int k = 0; k < D; ++k) {
                        float vcol = fac * vr[k];
#pragma omp atomic
                        A[8 + k] += (double)vcol;
                        buffer[k] += col[k] * fac;
                    }
#pragma omp atomic
                    A[0] += (double)vxy0;
#pragma omp atomic
                    A[1] += (double)vxy1;
#pragma omp atomic
                    A[2] += (double)fabsf(vxy0);
#pragma omp atomic
                    A[3] += (double)fabsf(vxy1);
#pragma omp atomic
                    A[4] += (double)vc0;
#pragma omp atomic
                    A[5] += (double)vc1;
#pragma omp atomic
                    A[6] += (double)vc2;
#pragma omp atomic
                    A[7] += (double)vo;
                }
            }
    }
    if (amk) {
#pragma omp parallel for schedule(dynamic, 4)
        for (int64_t t = 0; t < n_tiles; ++t) {
            int c = (int)(t / ((int64_t)tw * th));
            int ty = (int)((t / tw) % th), tx = (int)(t % tw);
            int64_t start = offsets[t], end = (t == n_tiles - 1) ? M : offsets[t + 1];
            if (end <= start) continue;
            int64_t L = end - start;
            float *base = NULL, *alt = NULL;
            for (int py = ty * tile_size; py < (ty + 1) * tile_size && py < H; ++py)
                for (int px = tx * tile_size; px < (tx + 1) * tile_size && px < W; ++px) {
                    int64_t pid = ((int64_t)c * H + py) * W + px;
                    if (!pixel_mask[pid]) continue;
                    if (!base) { base = (float *)malloc((size_t)L * SM * sizeof(float)); alt = (float *)malloc((size_t)L * SM * sizeof(float)); }
                    const float *vr = v_render + pid * D;
                    const float *bg = backgrounds ? backgrounds + (int64_t)c * D : NULL;
                    pixel_pass(D, means2d, conics, colors, opacities, bg, (float)px + 0.5f, (float)py + 0.5f, start, end, flatten_ids,
                               vr, v_alphas[pid], -1, 0, base);
                    /* the threshold-critical decisions of this pixel, found the way orc_blend_fwd_ex flags them */
                    float T = 1.f;
                    for (int64_t i = start; i < end; ++i) {
                        int32_t g = flatten_ids[i];
                        float dx = means2d[g * 2] - ((float)px + 0.5f), dy = means2d[g * 2 + 1] - ((float)py + 0.5f);
                        float a = conics[g * 3], b = conics[g * 3 + 1], cc = conics[g * 3 + 2];
                        float sigma = 0.5f * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                        float alpha = fminf(ALPHA_MAX, opacities[g] * expf(-sigma));
                        int crit_inc = (fabsf(alpha - ALPHA_MIN) <= 1e-4f * ALPHA_MIN && sigma >= -1e-6f) || (fabsf(sigma) <= 1e-6f && alpha >= ALPHA_MIN);
                        int skip = sigma < 0.f || alpha < ALPHA_MIN;
                        int crit_stop = 0, stop = 0;
                        float next_T = T;
                        if (!skip) {
                            next_T = T * (1.f - alpha);
                            crit_stop = fabsf(next_T - T_MIN) <= 1e-4f * T_MIN;
                            stop = next_T <= T_MIN;
                        }
                        for (int kind = 1; kind <= 2; ++kind) {
                            if (!(kind == 1 ? crit_inc : crit_stop)) continue;
                            pixel_pass(D, means2d, conics, colors, opacities, bg, (float)px + 0.5f, (float)py + 0.5f, start, end,
                                       flatten_ids, vr, v_alphas[pid], i, kind, alt);
                            for (int64_t q = 0; q < L; ++q) {
                                double *Q = amk + (int64_t)flatten_ids[start + q] * SM;
                                for (int k = 0; k < SM; ++k) {
                                    float d = fabsf(alt[q * SM + k] - base[q * SM + k]);
                                    if (d != 0.f) {
#pragma omp atomic
                                        Q[k] += (double)d;
                                    }
                                }
                            }
                        }
                        if (skip) continue;
                        if (stop) break;
                        T = next_T;
                    }
                }
            free(base);
            free(alt);
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < CN; ++g) {
        const double *A = acc + g * S;
        v_means2d[g * 2] = (float)A[0]; v_means2d[g * 2 + 1] = (float)A[1];
        if (v_means2d_abs) { v_means2d_abs[g * 2] = (float)A[2]; v_means2d_abs[g * 2 + 1] = (float)A[3]; }
        v_conics[g * 3] = (float)A[4]; v_conics[g * 3 + 1] = (float)A[5]; v_conics[g * 3 + 2] = (float)A[6];
        v_opacities[g] = (float)A[7];
        for (int k = 0; k < D; ++k) v_colors[g * D + k] = (float)A[8 + k];
        if (aab)
            for (int k = 0; k < SA; ++k) term_abs[g * SA + k] = (float)aab[g * SA + k];
        if (amk)
            for (int k = 0; k < SM; ++k) mask_terms[g * SM + k] = (float)amk[g * SM + k];
    }
    free(acc);
    free(aab);
    free(amk);
}

void orc_blend_bwd_ex(int C, int64_t N, int D, const float *means2d, const float *conics,
                      const float *colors, const float *opacities, const float *backgrounds, int W, int H,
                      int tile_size, int tw, int th, const int32_t *offsets, const int32_t *flatten_ids,
                      int64_t M, const float *alphas, const int32_t *last_ids, const float *v_render,
                      const float *v_alphas, float *v_means2d, float *v_means2d_abs, float *v_conics,
                      float *v_colors, float *v_opacities, float *term_abs) {
    orc_blend_bwd_ex2(C, N, D, means2d, conics, colors, opacities, backgrounds, W, H, tile_size, tw, th, offsets, flatten_ids,
                      M, alphas, last_ids, v_render, v_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities,
                      term_abs, NULL, NULL);
}

void orc_blend_bwd(int C, int64_t N, int D, const float *means2d, const float *conics,
                   const float *colors, const float *opacities, const float *backgrounds, int W, int H,
                   int tile_size, int tw, int th, const int32_t *offsets, const int32_t *flatten_ids,
                   int64_t M, const float *alphas, const int32_t *last_ids, const float *v_render,
                   const float *v_alphas, float *v_means2d, float *v_means2d_abs, float *v_conics,
                   float *v_colors, float *v_opacities) {
    orc_blend_bwd_ex(C, N, D, means2d, conics, colors, opacities, backgrounds, W, H, tile_size, tw, th, offsets, flatten_ids,
                     M, alphas, last_ids, v_render, v_alphas, v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities,
                     NULL);
}
