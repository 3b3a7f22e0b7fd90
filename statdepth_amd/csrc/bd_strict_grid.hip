// bd_strict_grid.hip -- K3 strict band depth (relax=False, J = 2) of SHORT series at LARGE n without testing every pair of points:
// the L-infinity (box) depth of a point cloud in R^3 (SURVEY.md 8 P4 = _functional.py:246-251 with T = d = 3; config 5: 10^6 points).
//
// Per target x the quantity is a function of the 27 counts h[c] of the OTHER points by state vector c in {tie, above, below}^3
// (strict_class_wg_kernel, bd_strict.hip: contained ordered pairs = sum over compatible classes, by the wild-card transform).
// That kernel classifies every point against every target: O(n^2), 427 ms at n = 10^6.  Here the counts come from a GRID:
//   * every coordinate is replaced by its integer rank B (points strictly below; equal values share B): the large-n rank
//     route's image mode, one pass over the 3 x n matrix (mbd_rank_big.hip).  Comparing ranks IS comparing values, ties included.
//   * a coordinate's cell = B >> shift, 256 cells per coordinate (value-based: equal values share a cell, and a smaller cell
//     index means a smaller value).  A point whose three cell indices ALL differ from the target's has its state decided by
//     the cell indices alone: its count per strict orthant of cells comes from a 3-D prefix sum over the 256^3 cell histogram
//     (27 look-ups per target).
//   * the other points share a cell index with the target in some coordinate: a slab of about n / 256 points per coordinate,
//     contiguous when the points are ordered by that coordinate.  They are classified explicitly, in THREE passes: pass k orders
//     points and targets by coordinate k, a workgroup takes 16 consecutive targets (one or two cells) and streams their slab
//     once for all 16 -- lanes = points, the target's ranks wave-uniform, one LDS atomic per (point, target) on shared
//     histograms, exactly the form of strict_class_wg_kernel -- counting a point only in the FIRST coordinate whose cell it
//     shares with the target.  3 n / 256 classifications per target instead of n: 85 times fewer at n = 10^6.
//   * a last kernel adds the prefix part to the three passes' counts and runs the 27-entry transform.
// Exact whatever the data (ties, duplicated points, clusters: a crowded cell only makes its slab longer); NaN-free data only --
// a NaN anywhere (the rank route counts them) leaves everything to the state-class kernels, which run behind this one and
// return at once otherwise.  All points are targets, or a subset of them (the others' positions are skipped).
#include "sd_common.h"

namespace sd {

int launch_rank_big_image(const double *Y, i64 T, i64 n, u32 *img, u32 *nnan, void *ws, size_t ws_bytes, hipStream_t s);   // mbd_rank_big.hip

constexpr int SG_LG = 8, SG_G = 1 << SG_LG;     // cells per coordinate
constexpr int SG_TG = 16;                       // targets per workgroup of a slab pass
constexpr int SG_COPIES = 16;                   // copies of a target's 27 counters (a lane counts into copy lane % 16)
constexpr int SG_NT = 256, SG_PTS = 4;
constexpr i64 SG_MIN_N = 32768;                 // measured: 10^5 points 0.74 ms against 4.97 ms of the state-class kernel (O(n^2)), 3 x 10^5: 2.0 / 39.5

static inline int sg_shift(i64 n) {
    int bits = 1;
    while (((i64)1 << bits) < n) ++bits;
    return bits > SG_LG ? bits - SG_LG : 0;
}

bool bd_strict_grid_applies(i64 T, i64 n, int J) { return J == 2 && T == 3 && n >= SG_MIN_N && n < ((i64)1 << 31) && mbd_rank_big_supported(T, n, 2); }

struct SgPlan {
    size_t off_R, off_nnan, off_cursor, off_rec, off_cellcnt, off_start, off_hc, off_c27, off_slot, off_big, big_bytes, total;
};
static SgPlan sg_plan(i64 n, bool subset) {
    SgPlan p;
    size_t o = 256;                                                  // the class kernels' flag
    auto take = [&](size_t b) { size_t r = o; o += align_up(b, 256); return r; };
    p.off_R = take((size_t)3 * n * 4);
    p.off_nnan = take(3 * 4);
    p.off_cursor = take((size_t)3 * n * 4);
    p.off_rec = take((size_t)3 * n * 16);
    p.off_cellcnt = take((size_t)3 * SG_G * 4);
    p.off_start = take((size_t)3 * (SG_G + 1) * 4);
    p.off_hc = take((size_t)SG_G * SG_G * SG_G * 4);
    p.off_c27 = take((size_t)n * 27 * 4);
    p.off_slot = take(subset ? (size_t)n * 4 : 0);
    p.big_bytes = mbd_rank_big_workspace_bytes(3, n, 2);
    p.off_big = take(p.big_bytes);
    p.total = o;
    return p;
}
size_t bd_strict_grid_workspace_bytes(i64 n, bool subset) { return sg_plan(n, subset).total; }

// ---- flag: a NaN in the image (the rank route's per-row NaN counts) ----
__global__ void sg_flag_kernel(const u32 *__restrict__ nnan, u32 *__restrict__ flag) {
    if (threadIdx.x == 0 && (nnan[0] | nnan[1] | nnan[2])) flag[0] = 1;
}

// ---- output slots of a subset of targets: slot[id] = position in the call's target list, -1 for the others ----
__global__ __launch_bounds__(256) void sg_slots_kernel(const i64 *__restrict__ targets, i64 m, int *__restrict__ slot) {
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q < m) slot[targets[q]] = (int)q;
}

// ---- records in the order of coordinate K: position = rank + arrival among the points that share it; per-coordinate cell
//      counts (LDS histogram per workgroup), and (K == 0) the 3-D cell histogram ----
template <int K>
__global__ __launch_bounds__(256) void sg_scatter_kernel(const u32 *__restrict__ R, i64 n, int shift, u32 *__restrict__ cursor,
                                                         uint4 *__restrict__ rec, u32 *__restrict__ cellcnt, u32 *__restrict__ hc,
                                                         const u32 *__restrict__ flag) {
    if (flag[0]) return;
    __shared__ u32 lh[SG_G];
    for (int c = threadIdx.x; c < SG_G; c += 256) lh[c] = 0;
    __syncthreads();
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        const u32 r0 = R[i] & 0x7FFFFFFFu, r1 = R[n + i] & 0x7FFFFFFFu, r2 = R[2 * n + i] & 0x7FFFFFFFu;
        const u32 rk = K == 0 ? r0 : (K == 1 ? r1 : r2);
        const u32 pos = rk + atomicAdd(&cursor[(size_t)K * n + rk], 1u);
        rec[(size_t)K * n + pos] = make_uint4(r0, r1, r2, (u32)i);
        atomicAdd(&lh[rk >> shift], 1u);
        if (K == 0) atomicAdd(&hc[((size_t)(r0 >> shift) * SG_G + (r1 >> shift)) * SG_G + (r2 >> shift)], 1u);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < SG_G; c += 256)
        if (lh[c]) atomicAdd(&cellcnt[K * SG_G + c], lh[c]);
}

// ---- start[k][c] = points with cell_k < c (the first position of cell c in order k), c = 0 .. 256; one workgroup ----
__global__ __launch_bounds__(256) void sg_starts_kernel(const u32 *__restrict__ cellcnt, u32 *__restrict__ start, const u32 *__restrict__ flag) {
    if (flag[0]) return;
    __shared__ u32 sc[SG_G];
    for (int k = 0; k < 3; ++k) {
        sc[threadIdx.x] = cellcnt[k * SG_G + threadIdx.x];
        __syncthreads();
        u32 acc = 0;
        for (int c = 0; c < (int)threadIdx.x; ++c) acc += sc[c];
        start[k * (SG_G + 1) + threadIdx.x] = acc;
        if (threadIdx.x == SG_G - 1) start[k * (SG_G + 1) + SG_G] = acc + sc[SG_G - 1];
        __syncthreads();
    }
}

// ---- inclusive prefix sums of the cell histogram along one axis (stride 1: a wave scan per line; else a thread per line) ----
__global__ __launch_bounds__(256) void sg_scan_inner_kernel(u32 *__restrict__ hc, const u32 *__restrict__ flag) {
    if (flag[0]) return;
    // one wave per line of 256 cells: 4 cells per lane
    const int lane = threadIdx.x & 63;
    const size_t line = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint4 *p = reinterpret_cast<uint4 *>(hc + line * SG_G) + lane;
    uint4 v = *p;
    v.y += v.x; v.z += v.y; v.w += v.z;
    u32 run = v.w;
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = __shfl_up(run, d);
        if (lane >= d) run += o;
    }
    const u32 before = run - v.w;
    v.x += before; v.y += before; v.z += before; v.w += before;
    *p = v;
}
__global__ __launch_bounds__(256) void sg_scan_outer_kernel(u32 *__restrict__ hc, size_t stride, const u32 *__restrict__ flag) {
    if (flag[0]) return;
    // thread = one line along the axis of the given stride (256 or 65 536 cells); neighbouring threads neighbouring cells
    const size_t lid = (size_t)blockIdx.x * 256 + threadIdx.x;       // 0 .. 65 535
    const size_t base = stride == SG_G ? (lid / SG_G) * SG_G * SG_G + (lid % SG_G) : lid;
    u32 run = 0;
#pragma unroll 8
    for (int c = 0; c < SG_G; ++c) {
        run += hc[base + (size_t)c * stride];
        hc[base + (size_t)c * stride] = run;
    }
}

__device__ __forceinline__ u32 sg_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// ---- slab pass K: see the header.  c27[id][27] += the classes of the points that share cell_K with target id and no cell of
//      an earlier coordinate ----
template <int K>
__global__ __launch_bounds__(SG_NT) void sg_pass_kernel(const uint4 *__restrict__ rec, const u32 *__restrict__ start, i64 n, int shift,
                                                       const int *__restrict__ slot, u32 *__restrict__ c27,
                                                       const u32 *__restrict__ flag) {
    if (flag[0]) return;
    constexpr int NC = 27, GS = SG_COPIES * NC;
    __shared__ u32 hist[SG_TG * GS];
    __shared__ uint4 xs[SG_TG];
    const int tid = threadIdx.x, lane = tid & 63;
    const i64 q0 = (i64)blockIdx.x * SG_TG;
    const int gc = (int)(n - q0 < SG_TG ? n - q0 : SG_TG);
    for (int c = tid; c < SG_TG * GS; c += SG_NT) hist[c] = 0;
    if (tid < SG_TG) xs[tid] = tid < gc ? rec[q0 + tid] : make_uint4(0, 0, 0, 0xFFFFFFFFu);
    __syncthreads();
    auto rk = [](const uint4 &v) -> u32 { return K == 0 ? v.x : (K == 1 ? v.y : v.z); };
    const u32 cmin = rk(xs[0]) >> shift, cmax = rk(xs[gc - 1]) >> shift;
    const i64 pbeg = start[K * (SG_G + 1) + cmin], pend = start[K * (SG_G + 1) + cmax + 1];
    // a subset of the targets: the positions that are none are skipped (block-uniform per g)
    for (i64 i0 = pbeg + tid; i0 < pend; i0 += (i64)SG_NT * SG_PTS) {
        uint4 p[SG_PTS];
        bool ok[SG_PTS];
#pragma unroll
        for (int k = 0; k < SG_PTS; ++k) {
            const i64 i = i0 + (i64)k * SG_NT;
            ok[k] = i < pend;
            p[k] = ok[k] ? rec[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll 1
        for (int g = 0; g < gc; ++g) {
            const u32 x0 = sg_uniform(xs[g].x), x1 = sg_uniform(xs[g].y), x2 = sg_uniform(xs[g].z), xid = sg_uniform(xs[g].w);
            if (slot && slot[xid] < 0) continue;                        // wave-uniform
            u32 *hg = hist + g * GS + (lane & (SG_COPIES - 1)) * NC;
            const u32 c0 = x0 >> shift, c1 = x1 >> shift, c2 = x2 >> shift;
#pragma unroll
            for (int k = 0; k < SG_PTS; ++k) {
                const u32 code = (p[k].x > x0 ? 1u : 0u) + (p[k].x < x0 ? 2u : 0u) + 3u * ((p[k].y > x1 ? 1u : 0u) + (p[k].y < x1 ? 2u : 0u)) +
                                 9u * ((p[k].z > x2 ? 1u : 0u) + (p[k].z < x2 ? 2u : 0u));
                bool use = ok[k] && p[k].w != xid;
                if (K == 0) use = use && (p[k].x >> shift) == c0;
                if (K == 1) use = use && (p[k].y >> shift) == c1 && (p[k].x >> shift) != c0;
                if (K == 2) use = use && (p[k].z >> shift) == c2 && (p[k].x >> shift) != c0 && (p[k].y >> shift) != c1;
                if (use) atomicAdd(&hg[code], 1u);
            }
        }
    }
    __syncthreads();
    for (int w = tid; w < gc * NC; w += SG_NT) {
        const int g = w / NC, c = w % NC;
        const u32 *h = hist + g * GS + c;
        u32 v = 0;
#pragma unroll
        for (int r = 0; r < SG_COPIES; ++r) v += h[r * NC];
        const u32 id = xs[g].w;
        if (v && !(slot && slot[id] < 0)) c27[(size_t)id * NC + c] += v;   // this pass owns the target: no other workgroup adds to it
    }
}

// ---- per target: the eight strict orthants of cells from the prefix sums + the three passes' counts, then the transform of
//      strict_class_wg_kernel (per coordinate the tie slot becomes the sum of the three states; sum of signed squares) ----
__global__ __launch_bounds__(256) void sg_final_kernel(const u32 *__restrict__ R, i64 n, int shift, const u32 *__restrict__ hc,
                                                      const u32 *__restrict__ c27, const i64 *__restrict__ targets, i64 m,
                                                      u64 *__restrict__ out, int jcols, const u32 *__restrict__ flag) {
    if (flag[0]) return;
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    const i64 id = targets ? targets[q] : q;
    const u32 cc[3] = {(R[id] & 0x7FFFFFFFu) >> shift, (R[n + id] & 0x7FFFFFFFu) >> shift, (R[2 * n + id] & 0x7FFFFFFFu) >> shift};
    // F(i, j, k) = points with cell_0 < i, cell_1 < j, cell_2 < k, from the inclusive sums
    auto F = [&](int i, int j, int k) -> long long {
        return (i > 0 && j > 0 && k > 0) ? (long long)hc[((size_t)(i - 1) * SG_G + (j - 1)) * SG_G + (k - 1)] : 0ll;
    };
    long long h[27];
#pragma unroll
    for (int c = 0; c < 27; ++c) h[c] = (long long)c27[(size_t)id * 27 + c];
    // state digits: 1 = above, 2 = below.  below: cells [0, c); above: cells [c + 1, G)
#pragma unroll
    for (int s0 = 1; s0 <= 2; ++s0)
#pragma unroll
        for (int s1 = 1; s1 <= 2; ++s1)
#pragma unroll
            for (int s2 = 1; s2 <= 2; ++s2) {
                long long cnt = 0;
#pragma unroll
                for (int e0 = 0; e0 < 2; ++e0)
#pragma unroll
                    for (int e1 = 0; e1 < 2; ++e1)
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            // below: the single term F(c); above: F(G) - F(c + 1)
                            if ((s0 == 2 && e0) || (s1 == 2 && e1) || (s2 == 2 && e2)) continue;
                            const int i = s0 == 2 ? (int)cc[0] : (e0 ? (int)cc[0] + 1 : SG_G);
                            const int j = s1 == 2 ? (int)cc[1] : (e1 ? (int)cc[1] + 1 : SG_G);
                            const int k = s2 == 2 ? (int)cc[2] : (e2 ? (int)cc[2] + 1 : SG_G);
                            const int neg = (s0 == 1 && e0) + (s1 == 1 && e1) + (s2 == 1 && e2);
                            cnt += (neg & 1) ? -F(i, j, k) : F(i, j, k);
                        }
                h[s0 + 3 * s1 + 9 * s2] += cnt;
            }
    const long long ties = h[0];                                      // the points that tie with the target everywhere
#pragma unroll
    for (int stride = 1; stride < 27; stride *= 3)
#pragma unroll
        for (int idx = 0; idx < 9; ++idx) {
            const int b = (idx / stride) * stride * 3 + (idx % stride);
            h[b] += h[b + stride] + h[b + 2 * stride];
        }
    long long total = 0;
#pragma unroll
    for (int c = 0; c < 27; ++c) {
        const int strict_digits = ((c % 3) != 0) + (((c / 3) % 3) != 0) + ((c / 9) != 0);
        total += (strict_digits & 1) ? -h[c] * h[c] : h[c] * h[c];
    }
    out[q * jcols] = ((u64)total - (u64)ties) / 2;
}

// counts of every target (targets == nullptr: all n points, in order) into out[q * jcols]; flag[0] (the class kernels' NaN flag,
// zeroed by the caller) is set when the data holds a NaN, and nothing is written then
int launch_bd_strict_grid(const double *Y, i64 n, const i64 *targets, i64 m, u64 *out, int jcols, u32 *flag, void *ws, size_t ws_bytes,
                          hipStream_t s) {
    const bool subset = targets != nullptr;
    const SgPlan p = sg_plan(n, subset);
    if (!ws || ws_bytes < p.total) return fail(SD_ERR_WORKSPACE, "strict grid workspace too small");
    char *w = (char *)ws;
    u32 *R = (u32 *)(w + p.off_R), *nnan = (u32 *)(w + p.off_nnan), *cursor = (u32 *)(w + p.off_cursor);
    uint4 *rec = (uint4 *)(w + p.off_rec);
    u32 *cellcnt = (u32 *)(w + p.off_cellcnt), *start = (u32 *)(w + p.off_start), *hc = (u32 *)(w + p.off_hc), *c27 = (u32 *)(w + p.off_c27);
    int *slot = subset ? (int *)(w + p.off_slot) : nullptr;
    const int shift = sg_shift(n);
    int rc = launch_rank_big_image(Y, 3, n, R, nnan, w + p.off_big, p.big_bytes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(sg_flag_kernel, dim3(1), dim3(64), 0, s, (const u32 *)nnan, flag);
    // cursor | ... | c27 are zeroed in three pieces (what lies between them is written before it is read)
    SD_HIP(hipMemsetAsync(cursor, 0, (size_t)3 * n * 4, s));
    SD_HIP(hipMemsetAsync(cellcnt, 0, (size_t)3 * SG_G * 4, s));
    SD_HIP(hipMemsetAsync(hc, 0, (size_t)SG_G * SG_G * SG_G * 4, s));
    SD_HIP(hipMemsetAsync(c27, 0, (size_t)n * 27 * 4, s));
    if (subset) {
        SD_HIP(hipMemsetAsync(slot, 0xFF, (size_t)n * 4, s));
        hipLaunchKernelGGL(sg_slots_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, targets, m, slot);
    }
    const unsigned gs = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(sg_scatter_kernel<0>, dim3(gs), dim3(256), 0, s, (const u32 *)R, n, shift, cursor, rec, cellcnt, hc, (const u32 *)flag);
    hipLaunchKernelGGL(sg_scatter_kernel<1>, dim3(gs), dim3(256), 0, s, (const u32 *)R, n, shift, cursor, rec, cellcnt, hc, (const u32 *)flag);
    hipLaunchKernelGGL(sg_scatter_kernel<2>, dim3(gs), dim3(256), 0, s, (const u32 *)R, n, shift, cursor, rec, cellcnt, hc, (const u32 *)flag);
    hipLaunchKernelGGL(sg_starts_kernel, dim3(1), dim3(SG_G), 0, s, (const u32 *)cellcnt, start, (const u32 *)flag);
    hipLaunchKernelGGL(sg_scan_inner_kernel, dim3(SG_G * SG_G / 4), dim3(256), 0, s, hc, (const u32 *)flag);
    hipLaunchKernelGGL(sg_scan_outer_kernel, dim3(SG_G * SG_G / 256), dim3(256), 0, s, hc, (size_t)SG_G, (const u32 *)flag);
    hipLaunchKernelGGL(sg_scan_outer_kernel, dim3(SG_G * SG_G / 256), dim3(256), 0, s, hc, (size_t)SG_G * SG_G, (const u32 *)flag);
    const unsigned gp = (unsigned)((n + SG_TG - 1) / SG_TG);
    hipLaunchKernelGGL(sg_pass_kernel<0>, dim3(gp), dim3(SG_NT), 0, s, (const uint4 *)rec, (const u32 *)start, n, shift, (const int *)slot, c27, (const u32 *)flag);
    hipLaunchKernelGGL(sg_pass_kernel<1>, dim3(gp), dim3(SG_NT), 0, s, (const uint4 *)(rec + n), (const u32 *)start, n, shift, (const int *)slot, c27, (const u32 *)flag);
    hipLaunchKernelGGL(sg_pass_kernel<2>, dim3(gp), dim3(SG_NT), 0, s, (const uint4 *)(rec + 2 * n), (const u32 *)start, n, shift, (const int *)slot, c27, (const u32 *)flag);
    hipLaunchKernelGGL(sg_final_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const u32 *)R, n, shift, (const u32 *)hc, (const u32 *)c27,
                       targets, m, out, jcols, (const u32 *)flag);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

}  // namespace sd
