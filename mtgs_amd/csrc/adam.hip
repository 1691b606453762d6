// adam.hip -- the optimizer step of every Gaussian parameter group in ONE launch.
//
// Reference: one torch.optim.Adam per parameter group, each holding one tensor
// (/root/reference/mtgs/scene_model/custom_trainer.py:115-136, groups / learning rates / eps = 1e-15 in
// mtgs/config/MTGS.py:121-181); the densification moves the moments with their rows (vanilla_gaussian_splatting.py:392-446,
// here csrc/refine.hip).  torch runs that as ~10 multi-tensor passes per group set; this is one streaming pass:
// p, m, v are read once and written once (24 B per element) plus the gradient (4 B) when it is dense.
//
// Two gradient sources per group:
//   dense  g[n]                                -- what autograd leaves in param.grad;
//   rows   rows[row_of[i] * row_stride + col + c], i = e / width, c = e % width, row_of[i] < 0 -> 0
//          -- compact gradient rows of the VISIBLE Gaussians (the rasterizer's backward works per visible Gaussian; a frame
//          sees ~15 % of a road block).  Culled Gaussians get the exact zero-gradient update (their moments decay, the
//          parameter keeps moving along exp_avg) without a dense gradient tensor ever being written or read; for a
//          per-traversal tensor [N, T, ...] only the slice of the frame's traversal takes the row (sub_width / sub_index).
//   both   (streaming groups) g + the row: the loss terms that reach a parameter outside the rasterization (MTGS's scale /
//          sharp-shape / out-of-box regularisers, mtgs_scene_graph.py:939-981) leave a dense param.grad while the
//          rasterization's own gradient arrives as rows.
//
// Arithmetic = torch.optim.Adam (amsgrad = False, maximize = False), fp32, in torch's operation order:
//   g += weight_decay * p;  m += (g - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g * g;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),   bc1 = 1 - beta1^t, bc2 = 1 - beta2^t computed by the HOST in double
// and handed over as hyper[group] = {lr / bc1, sqrt(bc2)} in device memory: the only per-step state, so a step captured
// in a HIP graph is advanced by one 8-byte-per-group copy in front of the replay.
// HBM-bound: 28 B per element dense, 24 B + the visible rows otherwise.
//
// Row-lazy groups (MTGS_ADAM_ROWS_*, include/mtgs_rast.h): tensors of which a frame reads the visible rows only (the SH
// coefficients under visibility-first colours); a workgroup scans ADAM_ROWS items and touches the selected rows (adam_rows).
#include "common.hpp"

#define ADAM_BLOCK 256
#define ADAM_VEC 4
#define ADAM_UNROLL 4
#define ADAM_ELEMS (ADAM_BLOCK * ADAM_VEC * ADAM_UNROLL)   // elements per workgroup
#define ADAM_ROWS 256                                       // items a workgroup scans (row-lazy groups): one per thread
#define ADAM_SEG 3                                          // 16-float segments of a row per pass (48 floats: degree-3 SH)

namespace {

struct Hyper {
    float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, wd, gscale;
};

// the slice (traversal) a group works on: a host value in the descriptor, or -- sub_index_dev -- a DEVICE word that is read when
// the kernel runs, so that ONE captured step serves every traversal (the caller rewrites the word in front of a replay)
__device__ __forceinline__ int slice_of(const mtgs_adam_group &d) { return d.sub_index_dev ? *d.sub_index_dev : d.sub_index; }

// The update in two halves with the operation sequence PINNED (no contraction beyond the fused multiply-adds written out): the
// row-lazy groups replay the moment half alone where the parameter half has already been done (adam_rows), and all paths
// -- streaming, slice, row catch-up, row step -- must produce the same bits.
__device__ __forceinline__ void adam_moments(const float p, float &m, float &v, float g, const Hyper &h) {
#pragma clang fp contract(off)
    g = g * h.gscale;
    if (h.wd != 0.f) g = fmaf(h.wd, p, g);
    m = fmaf(g - m, h.one_minus_b1, m);
    v = fmaf(v, h.b2, (h.one_minus_b2 * g) * g);
}
__device__ __forceinline__ void adam_param(float &p, const float m, const float v, const Hyper &h) {
#pragma clang fp contract(off)
    const float denom = __fsqrt_rn(v) / h.bc2_sqrt + h.eps;
    p = p - h.step_size * (m / denom);
}
__device__ __forceinline__ void adam_update(float &p, float &m, float &v, float g, const Hyper &h) {
    adam_moments(p, m, v, g, h);
    adam_param(p, m, v, h);
}

// Workgroup b belongs to group i with table[i].first_block <= b < table[i + 1].first_block (wave-uniform: scalar loads).
__device__ __forceinline__ int find_group(const mtgs_adam_group *table, int n, int64_t b) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4(const float *p) {
    if (NT) {
        const f4v r = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));   // global_load_dwordx4 ... nt
        return make_float4(r.x, r.y, r.z, r.w);
    }
    return *reinterpret_cast<const float4 *>(p);
}
template <bool NT>
__device__ __forceinline__ void st4(float *p, float4 r) {
    if (NT) {
        f4v t = {r.x, r.y, r.z, r.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(p));
        return;
    }
    *reinterpret_cast<float4 *>(p) = r;
}

// gradient of element e of a rows-source group (i = e / width tracked incrementally by the caller).  sub_width > 0: the
// item is width / sub_width slices (a per-traversal tensor [N, T, ...]) and only slice sub_index has a gradient.
__device__ __forceinline__ float row_grad(const mtgs_adam_group &d, int64_t i, int c) {
    const int32_t r = d.row_of[i];
    if (r < 0 || r >= d.n_rows) return 0.f;
    if (d.sub_width > 0) {
        const int s = c / d.sub_width;
        if (s != slice_of(d)) return 0.f;
        c -= s * d.sub_width;
    }
    return d.rows[(int64_t)r * d.row_stride + d.row_col + c];
}

// ---- row-lazy groups (include/mtgs_rast.h, MTGS_ADAM_ROWS_*) ---------------------------------------------------------------
// Modes: CATCHUP / FLUSH replay the missed zero-gradient steps in place; PEEK does the same in registers and leaves the
// up-to-date PARAMETER rows in the compact buffer `caught` (row r = the Gaussian's rank) without touching p, m, v or `last` --
// the forward reads its coefficients from there, coalesced, and has no side effect on the optimizer; STEP applies the step,
// and when the same frame's `caught` rows are handed back it takes p from them and replays only the MOMENT half of the
// missed steps (3 instructions each instead of ~40: the parameter half needs a square root and two divisions).
// Two ways to find the rows:
//   LIST (row_ids given: the frame's visible Gaussians in increasing order, rank -> index): rows straight from the list --
//        ids[r], then `last`, p, m, v, gradient / caught row all in flight together: ONE dependent round trip per row, 128 rows
//        per workgroup, the workgroups of every tensor assigned on the device from its number of visible items;
//   SCAN (row map only): a workgroup reads row_of / last of ADAM_ROWS items, compacts the selected ones in LDS (all `last` are
//        read before any is rewritten: a row belongs to one workgroup) and walks them; for callers without the id list and for
//        FLUSH, which selects by `last`.
// What the pattern costs on this chip (scripts/dev/gather_bench{,2}.hip: plain kernels over a precomputed id list, no arithmetic,
// 304k visible of 1.6M Gaussians, T = 3): the 180-byte pieces of ONE [N, T, 45] tensor from three arrays: 39 us read-only
// (4.2 TB/s), 111 us read + write; MTGS's three colour tensors (dc [N, 3], adapter [N, T, 3], rest [N, T, 45]) each with p, m, v
// and a stamp: 112 ... 148 us peek-like, 222 ... 253 us step-like -- the 12-byte and 4-byte pieces cost a memory transaction
// each.  Measured here, arithmetic included (MTGS-like iteration, 960x540): peek 155 ... 180 us (SCAN 175 ... 200), step 243 ... 297.
#define ADAM_HWIN 64     // steps of history in LDS; older steps: one load per LPR / 2 steps, shared by the row's lanes (row_work)
#ifndef ADAM_ZERO_SKIP_MAX
#define ADAM_ZERO_SKIP_MAX 6     // zero_probe without a bound of its own: a zero-gradient row is committed once it is this many steps behind
#endif
#define ADAM_LIST_ROWS 128
struct RowCtx {
    bool step, flush, peek, has_state;
    int t_now, target, win0, sw, T, slice;
    int64_t off;
    const float *s_hist;
};
__device__ __forceinline__ RowCtx row_ctx(const mtgs_adam_group &d, const Hyper &h, float *hy, bool first_thread, float *s_hist) {
    RowCtx x;
    x.sw = d.sub_width > 0 ? d.sub_width : d.width;          // floats of the slice this group works on
    x.T = d.sub_width > 0 ? d.width / d.sub_width : 1;
    x.slice = d.sub_width > 0 ? slice_of(d) : 0;
    x.off = (int64_t)x.slice * x.sw;
    x.t_now = reinterpret_cast<const int32_t *>(hy)[2];
    const int pending = reinterpret_cast<const int32_t *>(hy)[3];
    x.step = d.mode == MTGS_ADAM_ROWS_STEP; x.flush = d.mode == MTGS_ADAM_ROWS_FLUSH; x.peek = d.mode == MTGS_ADAM_ROWS_PEEK;
    x.has_state = d.m != nullptr && d.last != nullptr;      // (PEEK of a tensor that is not row-lazy: a plain copy)
    x.target = x.step ? x.t_now - 1 : (d.catchup_k >= 0 ? d.catchup_k : x.t_now - pending);   // zero-gradient steps up to here
    if (x.step && first_thread) {
        d.hist[2 * (int64_t)x.t_now] = h.step_size;
        d.hist[2 * (int64_t)x.t_now + 1] = h.bc2_sqrt;
        reinterpret_cast<int32_t *>(hy)[3] = 0;      // (nothing in a step launch reads it)
    }
    x.win0 = x.target - ADAM_HWIN + 1;               // s_hist[2 (j - win0)] = scalars of step j
    if (threadIdx.x < 2 * ADAM_HWIN && x.has_state) {
        const int j = x.win0 + ((int)threadIdx.x >> 1);
        s_hist[threadIdx.x] = j >= 1 ? d.hist[2 * (int64_t)j + (threadIdx.x & 1)] : 0.f;
    }
    x.s_hist = s_hist;
    return x;
}

// One row (item i, rank r) by one 16-lane group.  LIST: `last` is loaded here, together with the row (L_in is ignored).
template <bool LIST, int LPR>
__device__ __forceinline__ void row_work(const mtgs_adam_group &d, const Hyper &h, const RowCtx &x, const int64_t i, const int r,
                                         const int L_in, const int c0) {
    // row_flags (LIST): the rows the frame composites FROM (mtgs_blend_touch_packed).  PEEK: nobody reads the colour of the others,
    // so nothing of theirs is requested (the stamp, the parameter and moment pieces are a memory transaction each) and their
    // `caught` row is not written.  STEP: the others have a zero gradient and NO caught row -- they follow the zero_probe rule
    // (left lazy while fewer than K steps behind, then committed, so that no row the frames keep seeing falls far behind and the
    // replay of a row that becomes visible from behind an occluder stays short), taking p from the parameter itself.
    const bool flagged = !(LIST && d.row_flags && r >= 0 && r < d.n_rows) || d.row_flags[r] != 0;
    if (x.peek && !flagged) return;
    int32_t *lastp = x.has_state ? d.last + i * x.T + x.slice : nullptr;
    int L = L_in;
    if (LIST) L = x.has_state ? *lastp : x.target;
    if (x.step && d.zero_probe > 0 && x.has_state && r >= 0 && r < d.n_rows &&
        x.target - L < ((d.zero_probe >> 16) > 0 ? (d.zero_probe >> 16) : ADAM_ZERO_SKIP_MAX)) {
        // A visible row nothing was composited from (occluded: 90 % and more of the frustum-visible Gaussians) has an all-zero
        // gradient: it may stay lazy like an unseen one -- but the forward peeks every visible row, and a row that is visible in
        // every frame of its traversal and never stepped would have an ever longer history replayed by every peek.  So it is
        // left alone only while it is fewer than K steps behind (zero_probe >> 16; the caller scales it with the number of traversals:
        // a slice is rendered every T-th step); then this step commits it like a row with a
        // gradient.  One 12-byte probe of the compact gradient row (+ the stamp) decides, before any of the scattered
        // parameter / moment pieces is requested.
        const float *pr = d.rows + (int64_t)r * d.row_stride + ((d.zero_probe & 0xffff) - 1);
        const bool nz = c0 < 3 && pr[c0] != 0.f;
        const unsigned long long b = __ballot(nz);
        const int g0 = ((int)(threadIdx.x & 63) / LPR) * LPR;
        if (((b >> g0) & ((1ull << LPR) - 1ull)) == 0ull) return;
    }
    const bool in_rows = r >= 0 && r < d.n_rows;
    // STEP with this frame's caught rows: p comes from them, the missed steps are replayed for the moments only
    const bool use_caught = x.step && d.caught != nullptr && in_rows && h.wd == 0.f && flagged;
    const bool need_state = LIST ? x.has_state : (x.has_state && (x.step || L < x.target));
#pragma unroll 1
    for (int cb = 0; cb < x.sw; cb += LPR * ADAM_SEG) {
        float p[ADAM_SEG], m[ADAM_SEG], v[ADAM_SEG], g[ADAM_SEG];
        int64_t phys[ADAM_SEG];
#pragma unroll
        for (int u = 0; u < ADAM_SEG; ++u) {
            const int c = cb + LPR * u + c0;
            phys[u] = c < x.sw ? i * d.width + x.off + c : -1;
            g[u] = 0.f; m[u] = 0.f; v[u] = 0.f; p[u] = 0.f;
            if (phys[u] >= 0) {
                p[u] = use_caught ? d.caught[(int64_t)r * d.caught_stride + d.caught_col + c] : d.p[phys[u]];
                if (need_state) { m[u] = d.m[phys[u]]; v[u] = d.v[phys[u]]; }
                if (x.step && in_rows) g[u] = d.rows[(int64_t)r * d.row_stride + d.row_col + c];
            }
        }
        if (need_state) {
            Hyper hj = h;
            if (use_caught) {
                for (int j = L + 1; j <= x.target; ++j) {
#pragma unroll
                    for (int u = 0; u < ADAM_SEG; ++u) adam_moments(0.f, m[u], v[u], 0.f, h);   // (wd == 0: p is not read)
                }
            } else {
                // A row that has not been seen for ~900 steps has exp_avg at a FIXED POINT of the zero-gradient recurrence
                // (0, or a denormal that m * (1 - beta1) no longer moves), and step_size * m / (sqrt(v) / bc2 + eps) -- at most
                // step_size * |m| / eps -- is below a quarter ulp of p: from then on a step leaves m and p bit-for-bit alone
                // and only multiplies v by beta2.  `settled` lanes take that one-instruction step as long as the bound holds
                // for the step's scalar (checked per step: learning rates move), so a gap costs ~40 instructions per element
                // for its first ~900 steps and 2 for the rest -- still the same bits as stepping every time.
                bool settled = false;
                float m_max = 0.f, lim_min = 0.f, h_cache = 0.f;
                int h_base = -1;
                for (int j = L + 1; j <= x.target; ++j) {      // the zero-gradient steps this row missed, oldest first
                    if (j >= x.win0) { hj.step_size = x.s_hist[2 * (j - x.win0)]; hj.bc2_sqrt = x.s_hist[2 * (j - x.win0) + 1]; }
                    else {
                        // older than the LDS window: the scalars of LPR / 2 steps per load, one float per lane of the row's group
                        // (all lanes of a group walk the same j), handed round with a group-wide shuffle -- a long replay waits for
                        // memory once per LPR / 2 steps instead of once per step
                        constexpr int CH = LPR / 2;
                        const int jb = j - (j % CH);
                        if (jb != h_base) { h_cache = d.hist[2 * (int64_t)jb + c0]; h_base = jb; }
                        hj.step_size = __shfl(h_cache, 2 * (j - jb), LPR);
                        hj.bc2_sqrt = __shfl(h_cache, 2 * (j - jb) + 1, LPR);
                    }
                    if (settled && hj.step_size * m_max < lim_min) {
#pragma unroll
                        for (int u = 0; u < ADAM_SEG; ++u) adam_moments(0.f, m[u], v[u], 0.f, h);   // m stays, v *= beta2
                        continue;
                    }
                    bool fixed = h.wd == 0.f && h.eps > 0.f;
                    m_max = 0.f; lim_min = 3.0e38f;
#pragma unroll
                    for (int u = 0; u < ADAM_SEG; ++u) {
                        if (phys[u] < 0) continue;
                        const float m_old = m[u];
                        adam_update(p[u], m[u], v[u], 0.f, hj);
                        fixed = fixed && m[u] == m_old;
                        m_max = fmaxf(m_max, fabsf(m_old));
                        lim_min = fminf(lim_min, h.eps * fabsf(p[u]) * 7.450580596923828e-09f);   // eps |p| 2^-27
                    }
                    settled = fixed;
                }
            }
        }
        const bool dirty = x.step || L < x.target;       // (LIST, catch-up in place: a row that is current is left alone)
#pragma unroll
        for (int u = 0; u < ADAM_SEG; ++u) {
            if (phys[u] < 0) continue;
            if (x.peek) {
                if (in_rows) d.caught[(int64_t)r * d.caught_stride + d.caught_col + (cb + LPR * u + c0)] = p[u];
                continue;
            }
            if (!dirty) continue;
            if (x.step) adam_update(p[u], m[u], v[u], g[u], h);
            d.p[phys[u]] = p[u]; d.m[phys[u]] = m[u]; d.v[phys[u]] = v[u];
        }
    }
    // (SCAN: every lane of the workgroup has read its `last` before the barrier; LIST: the row's 16 lanes read it above, same wave)
    if (c0 == 0 && !x.peek && x.has_state && (x.step || L < x.target)) *lastp = x.step ? x.t_now : x.target;
}

__device__ __forceinline__ void adam_rows_scan(const mtgs_adam_group &d, const Hyper &h, float *hy, int64_t block_in_group) {
    __shared__ int s_item[ADAM_ROWS], s_L[ADAM_ROWS], s_r[ADAM_ROWS];
    __shared__ float s_hist[2 * ADAM_HWIN];
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const RowCtx x = row_ctx(d, h, hy, block_in_group == 0 && tid == 0, s_hist);
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const int64_t base = block_in_group * ADAM_ROWS;
    {
        const int64_t i = base + tid;
        bool sel = false;
        int L = x.target, r = -1;
        if (i < d.n) {
            if (x.has_state) L = d.last[i * x.T + x.slice];
            if (!x.flush) r = d.row_of[i];
            sel = x.flush ? L < x.target : (x.peek ? (r >= 0 && r < d.n_rows) : (r >= 0 && (x.step || L < x.target)));
        }
        const unsigned long long mask = __ballot(sel);
        if (mask != 0) {
            int wbase = 0;
            if ((tid & 63) == 0) wbase = atomicAdd(&s_cnt, __popcll(mask));
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (sel) {
                const int at = wbase + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                s_item[at] = tid; s_L[at] = L; s_r[at] = r;
            }
        }
    }
    __syncthreads();      // every `last` of this workgroup's items has been read; the list is complete
    const int cnt = s_cnt;
#pragma unroll 1
    for (int q = tid >> 4; q < cnt; q += ADAM_BLOCK / 16) row_work<false, 16>(d, h, x, base + s_item[q], s_r[q], s_L[q], tid & 15);
}

// LIST: a workgroup takes ADAM_LIST_ROWS consecutive ranks; rows of more than 4 floats by 16 lanes each (8 rows per lane group),
// narrower ones (features_dc, one traversal's adapter: 3 floats) by 4 lanes each (2 rows per lane group) -- the prologue
// (descriptor, scalars, history window, one barrier) is paid once per 128 rows.
template <int LPR>
__device__ __forceinline__ void adam_rows_list_t(const mtgs_adam_group &d, const Hyper &h, const RowCtx &x, int64_t block_in_group) {
    const int tid = threadIdx.x;
    constexpr int GROUPS = ADAM_BLOCK / LPR, TRIPS = ADAM_LIST_ROWS / GROUPS;
    const int64_t q0 = block_in_group * ADAM_LIST_ROWS + tid / LPR;      // position among this tensor's ranks
    int32_t id[TRIPS];
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {       // the block's ids first: TRIPS independent loads
        const int64_t q = q0 + (int64_t)k * GROUPS;
        id[k] = q < d.rank_count ? d.row_ids[d.rank_start + q] : -1;
    }
#pragma unroll 1
    for (int k = 0; k < TRIPS; ++k) {
        if (id[k] < 0) continue;
        row_work<true, LPR>(d, h, x, (int64_t)id[k] - d.item_start, (int)(d.rank_start + q0 + (int64_t)k * GROUPS), 0, tid % LPR);
    }
}
__device__ __forceinline__ void adam_rows_list(const mtgs_adam_group &d, const Hyper &h, float *hy, int64_t block_in_group) {
    __shared__ float s_hist[2 * ADAM_HWIN];
    const RowCtx x = row_ctx(d, h, hy, false, s_hist);      // (the step's bookkeeping: adam_list_search_kernel)
    __syncthreads();      // s_hist
    if (x.sw <= 4) adam_rows_list_t<4>(d, h, x, block_in_group);
    else adam_rows_list_t<16>(d, h, x, block_in_group);
}

// Scheduling of the LIST groups (they are the LAST groups of a table), one workgroup before the row kernel: for every LIST group
// the ranks of its tensor's items -- rank_start = first rank whose Gaussian index is >= item_start, rank_count = ranks below
// item_start + n (row_ids is increasing; a 64-ary search by one wave each, three or four dependent loads for 300k rows) -- and, from
// the counts, the workgroups each group really needs: first_block of the LIST groups is REWRITTEN here (the host launches an
// upper bound; surplus workgroups leave after one descriptor load).
__device__ __forceinline__ int64_t wave_lower_bound(const int32_t *__restrict__ ids, int64_t count, int64_t key) {
    const int lane = threadIdx.x & 63;
    int64_t lo = 0, hi = count;          // ids[k] < key for k < lo, ids[k] >= key for k >= hi
    while (hi > lo) {
        const int64_t stride = (hi - lo + 63) / 64;
        const int64_t k = lo + (int64_t)lane * stride;
        const bool ge = k >= hi || (int64_t)ids[k] >= key;
        const unsigned long long m = __ballot(ge);
        const int f = m ? __builtin_ctzll(m) : 64;      // first probe that is >= key (lane 0 probes lo itself)
        if (f == 0) { hi = lo; break; }
        const int64_t new_lo = lo + (int64_t)(f - 1) * stride + 1;
        const int64_t new_hi = f < 64 ? (lo + (int64_t)f * stride < hi ? lo + (int64_t)f * stride : hi) : hi;
        lo = new_lo; hi = new_hi;
    }
    return lo;
}
#define ADAM_SCHED_THREADS 1024
#define ADAM_SEARCH_THREADS 256
// 1. the searches, one per wave, spread over as many workgroups as there are tasks (a scene graph with a hundred object nodes has
//    ~300 LIST groups = 600 searches of three or four dependent loads each: in ONE workgroup they took 172 us per launch):
//    task 2 g = the start of group g's ranks, 2 g + 1 = their end (parked in rank_count until the assignment below)
__global__ void __launch_bounds__(ADAM_SEARCH_THREADS) adam_list_search_kernel(mtgs_adam_group *__restrict__ table,
                                                                              float *__restrict__ hyper, int n_groups) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int task = blockIdx.x * (ADAM_SEARCH_THREADS / 64) + wave;
    if (task >= 2 * n_groups) return;
    const int g = task >> 1, which = task & 1;
    mtgs_adam_group &d = table[g];
    if (d.mode < MTGS_ADAM_ROWS_CATCHUP || d.row_ids == nullptr) return;
    if (which == 0 && d.mode == MTGS_ADAM_ROWS_STEP && lane == 0) {
        // the step's bookkeeping (row_ctx does it for SCAN groups): a LIST group may have no workgroup at all -- a node
        // none of whose Gaussians the frame saw -- and the step still counts
        float *hy = hyper + 4 * (int64_t)d.hyper_index;
        const int t_now = reinterpret_cast<const int32_t *>(hy)[2];
        d.hist[2 * (int64_t)t_now] = hy[0];
        d.hist[2 * (int64_t)t_now + 1] = hy[1];
        reinterpret_cast<int32_t *>(hy)[3] = 0;
    }
    int64_t count = d.n_rows;
    if (d.row_count_dev) { const int64_t c = *d.row_count_dev >> 32; if (c < count) count = c; }
    const int64_t r = wave_lower_bound(d.row_ids, count, which ? d.item_start + d.n : d.item_start);
    if (lane == 0) {
        if (which) d.rank_count = (int32_t)r; else d.rank_start = (int32_t)r;
    }
}
// 2. the workgroups every LIST group really needs: first_block of the LIST groups is REWRITTEN (the host launches an upper bound;
//    surplus workgroups leave after one descriptor load).  One workgroup: the bounds are gathered into LDS by all threads, the prefix
//    itself is a chain of stores by one thread, not of dependent loads.
__global__ void __launch_bounds__(ADAM_SCHED_THREADS) adam_list_assign_kernel(mtgs_adam_group *__restrict__ table, int n_groups) {
    constexpr int MAXG = 2 * ADAM_SCHED_THREADS;   // groups handled by the parallel prefix (two per thread); more: one thread walks them
    __shared__ int64_t s_ws[ADAM_SCHED_THREADS / 64];
    __shared__ int64_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n_groups <= MAXG) {
        // exclusive prefix of the LIST groups' workgroup counts, two consecutive groups per thread; the first LIST group keeps the
        // host's first_block, the others follow it
        int cnt[2];
        int64_t blocks[2], first[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int g = 2 * tid + e;
            cnt[e] = -1; blocks[e] = 0; first[e] = 0;
            if (g < n_groups) {
                const mtgs_adam_group &d = table[g];
                if (d.mode >= MTGS_ADAM_ROWS_CATCHUP && d.row_ids != nullptr) {
                    cnt[e] = d.rank_count - d.rank_start;
                    blocks[e] = ((int64_t)cnt[e] + ADAM_LIST_ROWS - 1) / ADAM_LIST_ROWS;
                    first[e] = d.first_block;
                }
            }
        }
        // the first LIST group: smallest index with cnt >= 0
        if (tid == 0) s_base = -1;
        __syncthreads();
        const int mine = cnt[0] >= 0 ? 2 * tid : (cnt[1] >= 0 ? 2 * tid + 1 : 0x7fffffff);
        int best = mine;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
        __shared__ int s_first_idx[ADAM_SCHED_THREADS / 64];
        if (lane == 0) s_first_idx[wave] = best;
        const int64_t sum = blocks[0] + blocks[1];
        int64_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_ws[wave] = inc;
        __syncthreads();
        int first_idx = 0x7fffffff;
        for (int w = 0; w < ADAM_SCHED_THREADS / 64; ++w) first_idx = min(first_idx, s_first_idx[w]);
        if (first_idx == 0x7fffffff) return;                      // no LIST group
        if (2 * tid == (first_idx & ~1)) s_base = first[first_idx & 1];
        __syncthreads();
        int64_t run = s_base + inc - sum;
        for (int w = 0; w < wave; ++w) run += s_ws[w];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (cnt[e] >= 0) {
                table[2 * tid + e].rank_count = cnt[e];
                table[2 * tid + e].first_block = run;
            }
            run += blocks[e];
        }
        return;
    }
    if (tid == 0) {
        int64_t next = -1;
        for (int g = 0; g < n_groups; ++g) {
            mtgs_adam_group &e = table[g];
            if (e.mode < MTGS_ADAM_ROWS_CATCHUP || e.row_ids == nullptr) continue;
            const int cnt = e.rank_count - e.rank_start;
            e.rank_count = cnt;
            if (next < 0) next = e.first_block;      // (the first LIST group keeps the host's value)
            e.first_block = next;
            next += ((int64_t)cnt + ADAM_LIST_ROWS - 1) / ADAM_LIST_ROWS;
        }
    }
}

// item (Gaussian) and column of element e of a tensor with `width` floats per item.  The widths of the narrow geometry
// tensors (1, 3, 4) and element indices below 2^32 -- every MTGS tensor -- take a shift or a multiply-high instead of the
// 64-bit software division a run-time divisor costs (~60 instructions per float4 of a streaming group with a row map).
__device__ __forceinline__ void item_of(const int64_t e, const int width, int64_t &i, int &c) {
    if (e < ((int64_t)1 << 32)) {
        const uint32_t e32 = (uint32_t)e;
        uint32_t q;
        switch (width) {
            case 1: q = e32; break;
            case 3: q = e32 / 3u; break;
            case 4: q = e32 >> 2; break;
            default: q = e32 / (uint32_t)width; break;
        }
        i = q;
        c = (int)(e32 - q * (uint32_t)width);
        return;
    }
    i = e / width;
    c = (int)(e - i * width);
}

template <bool NT>
__global__ void __launch_bounds__(ADAM_BLOCK) adam_kernel(const mtgs_adam_group *__restrict__ table,
                                                          float *__restrict__ hyper, int n_groups) {
    const int gi = find_group(table, n_groups, (int64_t)blockIdx.x);
    const mtgs_adam_group d = table[gi];
    float *hy = hyper + 4 * (int64_t)d.hyper_index;
    Hyper h;
    h.one_minus_b1 = d.one_minus_beta1; h.b2 = d.beta2; h.one_minus_b2 = d.one_minus_beta2;
    h.step_size = hy[0]; h.bc2_sqrt = hy[1]; h.eps = d.eps; h.wd = d.weight_decay; h.gscale = d.grad_scale;
    const int64_t base = ((int64_t)blockIdx.x - d.first_block) * ADAM_ELEMS;
    float *__restrict__ P = d.p, *__restrict__ M = d.m, *__restrict__ V = d.v;
    const bool dense = d.g != nullptr;
    const bool rows = d.rows != nullptr;
    if (d.mode == MTGS_ADAM_SLICE) {
        // ONE slice of a per-traversal tensor: virtual element e -> p[(e / sub_width) * width + sub_index * sub_width + e % sub_width];
        // the other slices are not touched.  A slice row is sub_width contiguous floats at a 4-byte-aligned offset: 4-byte
        // accesses, coalesced across the wave.  (A 16-elements-per-lane version with multiply-high row indices and
        // non-temporal moments measured SLOWER, 3.70 against 3.41 ms for the T = 8 iteration: 48 more live registers.  The
        // slices are 180-byte pieces every T * 180 bytes, so the fabric moves whole lines around them: at T = 3 the lazy
        // scheme's two passes over one slice cost more than one pass over three, it pays from T ~ 5.)
        const int64_t end = base + ADAM_ELEMS < d.n ? base + ADAM_ELEMS : d.n;
        const int sw = d.sub_width;
        const int64_t off = (int64_t)slice_of(d) * sw;
        for (int64_t e0 = base + threadIdx.x; e0 < end; e0 += ADAM_BLOCK * ADAM_UNROLL) {
            float p[ADAM_UNROLL], m[ADAM_UNROLL], v[ADAM_UNROLL], g[ADAM_UNROLL];
            int64_t phys[ADAM_UNROLL];
#pragma unroll
            for (int u = 0; u < ADAM_UNROLL; ++u) {
                const int64_t e = e0 + (int64_t)u * ADAM_BLOCK;
                phys[u] = -1;
                if (e < end) {
                    const int64_t i = e / sw;
                    const int c = (int)(e - i * sw);
                    phys[u] = i * d.width + off + c;
                    p[u] = P[phys[u]]; m[u] = M[phys[u]]; v[u] = V[phys[u]];
                    g[u] = 0.f;
                    if (d.catchup_k == 0) {
                        if (dense) g[u] = d.g[phys[u]];
                        if (rows) { const int32_t r = d.row_of[i]; g[u] += (r < 0 || r >= d.n_rows) ? 0.f : d.rows[(int64_t)r * d.row_stride + d.row_col + c]; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < ADAM_UNROLL; ++u) {
                if (phys[u] < 0) continue;
                if (d.catchup_k > 0) {
                    Hyper hj = h;
                    for (int j = 0; j < d.catchup_k; ++j) {      // the missed zero-gradient steps, oldest first
                        hj.step_size = d.catchup[2 * j]; hj.bc2_sqrt = d.catchup[2 * j + 1];
                        adam_update(p[u], m[u], v[u], 0.f, hj);
                    }
                } else {
                    adam_update(p[u], m[u], v[u], g[u], h);
                }
                P[phys[u]] = p[u]; M[phys[u]] = m[u]; V[phys[u]] = v[u];
            }
        }
        return;
    }
    if (d.vec_ok && base + ADAM_ELEMS <= d.n) {
        // full chunk, 16-byte accesses: UNROLL x 3 (4) loads per lane in flight before the first use
        float4 p[ADAM_UNROLL], m[ADAM_UNROLL], v[ADAM_UNROLL], g[ADAM_UNROLL];
#pragma unroll
        for (int u = 0; u < ADAM_UNROLL; ++u) {
            const int64_t e = base + ((int64_t)u * ADAM_BLOCK + threadIdx.x) * ADAM_VEC;
            p[u] = ld4<false>(P + e);
            m[u] = ld4<NT>(M + e);
            v[u] = ld4<NT>(V + e);
            if (dense) g[u] = ld4<NT>(d.g + e);
        }
        if (!dense || rows) {      // (both sources: the dense gradient of the terms outside the rasterization + the rows)
#pragma unroll
            for (int u = 0; u < ADAM_UNROLL; ++u) {
                if (!dense) g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rows) {
                    const int64_t e = base + ((int64_t)u * ADAM_BLOCK + threadIdx.x) * ADAM_VEC;
                    int64_t i;
                    int c;
                    item_of(e, d.width, i, c);
                    float t[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        t[k] = row_grad(d, i, c);
                        if (++c == d.width) { c = 0; ++i; }
                    }
                    g[u] = make_float4(g[u].x + t[0], g[u].y + t[1], g[u].z + t[2], g[u].w + t[3]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < ADAM_UNROLL; ++u) {
            const int64_t e = base + ((int64_t)u * ADAM_BLOCK + threadIdx.x) * ADAM_VEC;
            adam_update(p[u].x, m[u].x, v[u].x, g[u].x, h);
            adam_update(p[u].y, m[u].y, v[u].y, g[u].y, h);
            adam_update(p[u].z, m[u].z, v[u].z, g[u].z, h);
            adam_update(p[u].w, m[u].w, v[u].w, g[u].w, h);
            st4<false>(P + e, p[u]);
            st4<NT>(M + e, m[u]);
            st4<NT>(V + e, v[u]);
        }
        return;
    }
    // ragged tail of a group, or a group whose tensors are not 16-byte aligned: one element per lane and trip
    const int64_t end = base + ADAM_ELEMS < d.n ? base + ADAM_ELEMS : d.n;
    for (int64_t e = base + threadIdx.x; e < end; e += ADAM_BLOCK) {
        float g = 0.f;
        if (dense) g = d.g[e];
        if (rows) { int64_t i; int c; item_of(e, d.width, i, c); g += row_grad(d, i, c); }
        float p = P[e], m = M[e], v = V[e];
        adam_update(p, m, v, g, h);
        P[e] = p; M[e] = m; V[e] = v;
    }
}

// The row-lazy groups of a table (the last ones): their own kernel, because the streaming kernel above holds 64 data registers
// per lane (4 waves per SIMD) and this one lives on memory round trips.  Measured (adam_bench --only rowlazy, catch-up / step):
// two rows per 16-lane group at 4 waves per SIMD 453 / 218 us, one row at 6 waves (66 registers) 400 / 212 us, at 8 waves 393 /
// 220 us; forcing the two-row version to 6 waves spills 25 registers: 502 / 402 us.  The step's ~2 TB/s is the rate 180-byte
// pieces at 540-byte pitch move at on this chip (vis_color_fwd's coefficient gather sees the same), not an occupancy limit.
#ifndef MTGS_ADAM_ROWS_WAVES
#define MTGS_ADAM_ROWS_WAVES 6
#endif
__global__ void __launch_bounds__(ADAM_BLOCK, MTGS_ADAM_ROWS_WAVES) adam_rows_kernel(const mtgs_adam_group *__restrict__ table,
                                                                                     float *__restrict__ hyper, int n_groups,
                                                                                     int64_t block_offset) {
    const int64_t b = (int64_t)blockIdx.x + block_offset;
    const int gi = find_group(table, n_groups, b);
    const mtgs_adam_group d = table[gi];
    if (d.row_ids && (b - d.first_block) * ADAM_LIST_ROWS >= d.rank_count) return;      // (LIST: the grid is an upper bound)
    float *hy = hyper + 4 * (int64_t)d.hyper_index;
    Hyper h;
    h.one_minus_b1 = d.one_minus_beta1; h.b2 = d.beta2; h.one_minus_b2 = d.one_minus_beta2;
    h.step_size = hy[0]; h.bc2_sqrt = hy[1]; h.eps = d.eps; h.wd = d.weight_decay; h.gscale = d.grad_scale;
    if (d.row_ids) {
        adam_rows_list(d, h, hy, b - d.first_block);
    } else {
        adam_rows_scan(d, h, hy, b - d.first_block);
    }
}

}  // namespace

extern "C" int mtgs_adam_group_bytes(void) { return (int)sizeof(mtgs_adam_group); }
extern "C" int mtgs_adam_block_elems(void) { return ADAM_ELEMS; }
extern "C" int mtgs_adam_block_rows(void) { return ADAM_ROWS; }

extern "C" int mtgs_adam_block_list_rows(void) { return ADAM_LIST_ROWS; }

extern "C" int mtgs_adam_step(int n_groups, mtgs_adam_group *table, float *hyper, int64_t total_blocks, int64_t rows_from_block,
                              int flags, void *stream) {
    MTGS_REQUIRE(n_groups >= 0 && total_blocks >= 0 && rows_from_block >= 0 && rows_from_block <= total_blocks, MTGS_EINVAL,
                 "mtgs_adam_step: bad sizes");
    if (n_groups == 0 || total_blocks == 0) return MTGS_OK;
    MTGS_REQUIRE(table != nullptr && hyper != nullptr, MTGS_EINVAL, "mtgs_adam_step: null table");
    MTGS_REQUIRE(((uintptr_t)table & 7) == 0, MTGS_EINVAL, "mtgs_adam_step: table must be 8-byte aligned");
    MTGS_REQUIRE(total_blocks < ((int64_t)1 << 31), MTGS_EINVAL, "mtgs_adam_step: more than 2^31 workgroups");
    hipStream_t st = (hipStream_t)stream;
    const bool nontemporal = flags & 1;
    if (rows_from_block > 0) {
        if (nontemporal)
            hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)rows_from_block), dim3(ADAM_BLOCK), 0, st, table, hyper, n_groups);
        else
            hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)rows_from_block), dim3(ADAM_BLOCK), 0, st, table, hyper, n_groups);
    }
    if (rows_from_block < total_blocks) {
        if (flags & 2) {
            const unsigned sb = (unsigned)((2 * (int64_t)n_groups + ADAM_SEARCH_THREADS / 64 - 1) / (ADAM_SEARCH_THREADS / 64));
            hipLaunchKernelGGL(adam_list_search_kernel, dim3(sb), dim3(ADAM_SEARCH_THREADS), 0, st, table, hyper, n_groups);
            hipLaunchKernelGGL(adam_list_assign_kernel, dim3(1), dim3(ADAM_SCHED_THREADS), 0, st, table, n_groups);
        }
        hipLaunchKernelGGL(adam_rows_kernel, dim3((unsigned)(total_blocks - rows_from_block)), dim3(ADAM_BLOCK), 0, st, table, hyper,
                           n_groups, rows_from_block);
    }
    MTGS_CHECK_LAUNCH("mtgs_adam_step");
    return MTGS_OK;
}
