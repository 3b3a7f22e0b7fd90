// bd_strict_grid.hip -- K3 strict band depth (relax=False, J = 2) of SHORT series at LARGE n without testing every pair of points:
// the L-infinity (box) depth of a point cloud in R^2, R^3 or R^4 (SURVEY.md 8 P4 = _functional.py:246-251 with T = d; config 5: 10^6 points
// in R^3).  Written out for three coordinates below; two and four work the same way (2 048 / 64 cells per coordinate, 9 / 81 classes).
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

constexpr int SG_TG = 16;                       // targets per workgroup of a slab pass
constexpr int SG_NT = 256, SG_PTS = 4;
constexpr i64 SG_MIN_N = 32768;                 // measured (R^3): 10^5 points 0.74 ms against 4.97 ms of the state-class kernel (O(n^2)), 3 x 10^5: 2.0 / 39.5

template <int TT> struct SgCfg {
    static_assert(TT >= 2 && TT <= 4, "two to four coordinates");
    static constexpr int LG = TT == 2 ? 11 : (TT == 3 ? 8 : 6);       // cells per coordinate: 2 048 / 256 / 64 (4 / 16.7 / 16.7 M cells in all)
    static constexpr int G = 1 << LG;
    static constexpr int NC = TT == 2 ? 9 : (TT == 3 ? 27 : 81);      // state classes
    static constexpr int RW = TT <= 3 ? 4 : 8;                        // words of a record: TT ranks, the point's index, padding
    static constexpr int IDW = TT <= 3 ? 3 : 4;                       // the word that holds the index (TT = 2: word 2 is unused)
    static constexpr int COPIES = TT == 4 ? 4 : 16;                   // copies of a target's counters (a lane counts into copy lane % COPIES)
    static constexpr size_t CELLS = (size_t)1 << (LG * TT);
};

static inline int sg_shift(i64 n, int lg) {
    int bits = 1;
    while (((i64)1 << bits) < n) ++bits;
    return bits > lg ? bits - lg : 0;
}

bool bd_strict_grid_applies(i64 T, i64 n, int J) {
    return J == 2 && T >= 2 && T <= 4 && n >= SG_MIN_N && n < ((i64)1 << 31) && mbd_rank_big_supported(T, n, 2);
}

struct SgPlan {
    size_t off_R, off_nnan, off_cursor, off_rec, off_cellcnt, off_start, off_hc, off_cnt, off_slot, off_big, big_bytes, total;
};
template <int TT>
static SgPlan sg_plan(i64 n, bool subset) {
    using C = SgCfg<TT>;
    SgPlan p;
    size_t o = 256;                                                  // the class kernels' flag
    auto take = [&](size_t b) { size_t r = o; o += align_up(b, 256); return r; };
    p.off_R = take((size_t)TT * n * 4);
    p.off_nnan = take(TT * 4);
    p.off_cursor = take((size_t)TT * n * 4);
    p.off_rec = take((size_t)TT * n * C::RW * 4);
    p.off_cellcnt = take((size_t)TT * C::G * 4);
    p.off_start = take((size_t)TT * (C::G + 1) * 4);
    p.off_hc = take(C::CELLS * 4);
    p.off_cnt = take((size_t)n * C::NC * 4);
    p.off_slot = take(subset ? (size_t)n * 4 : 0);
    p.big_bytes = mbd_rank_big_workspace_bytes(TT, n, 2);
    p.off_big = take(p.big_bytes);
    p.total = o;
    return p;
}
size_t bd_strict_grid_workspace_bytes(i64 T, i64 n, bool subset) {
    return T == 2 ? sg_plan<2>(n, subset).total : (T == 3 ? sg_plan<3>(n, subset).total : sg_plan<4>(n, subset).total);
}

// ---- flag: a NaN in the image (the rank route's per-row NaN counts) ----
__global__ void sg_flag_kernel(const u32 *__restrict__ nnan, int TT, u32 *__restrict__ flag) {
    if (threadIdx.x == 0) {
        u32 any = 0;
        for (int t = 0; t < TT; ++t) any |= nnan[t];
        if (any) flag[0] = 1;
    }
}

// ---- output slots of a subset of targets: slot[id] = position in the call's target list, -1 for the others ----
__global__ __launch_bounds__(256) void sg_slots_kernel(const i64 *__restrict__ targets, i64 m, int *__restrict__ slot) {
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q < m) slot[targets[q]] = (int)q;
}

// ---- records in the order of coordinate K: position = rank + arrival among the points that share it; per-coordinate cell
//      counts (LDS histogram per workgroup), and (K == 0) the cell histogram ----
template <int TT, int K>
__global__ __launch_bounds__(256) void sg_scatter_kernel(const u32 *__restrict__ R, i64 n, int shift, u32 *__restrict__ cursor,
                                                         u32 *__restrict__ rec, u32 *__restrict__ cellcnt, u32 *__restrict__ hc,
                                                         const u32 *__restrict__ flag) {
    if (flag[0]) return;
    using C = SgCfg<TT>;
    __shared__ u32 lh[C::G];
    for (int c = threadIdx.x; c < C::G; c += 256) lh[c] = 0;
    __syncthreads();
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
        u32 r[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) r[t] = R[(size_t)t * n + i] & 0x7FFFFFFFu;
        const u32 pos = r[K] + atomicAdd(&cursor[(size_t)K * n + r[K]], 1u);
        uint4 *dst = reinterpret_cast<uint4 *>(rec + ((size_t)K * n + pos) * C::RW);
        if constexpr (TT == 2) dst[0] = make_uint4(r[0], r[1], 0u, (u32)i);
        if constexpr (TT == 3) dst[0] = make_uint4(r[0], r[1], r[2], (u32)i);
        if constexpr (TT == 4) { dst[0] = make_uint4(r[0], r[1], r[2], r[3]); dst[1] = make_uint4((u32)i, 0u, 0u, 0u); }
        atomicAdd(&lh[r[K] >> shift], 1u);
        if (K == 0) {
            size_t cell = 0;
#pragma unroll
            for (int t = 0; t < TT; ++t) cell = (cell << C::LG) | (r[t] >> shift);
            atomicAdd(&hc[cell], 1u);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C::G; c += 256)
        if (lh[c]) atomicAdd(&cellcnt[K * C::G + c], lh[c]);
}

// ---- start[k][c] = points with cell_k < c (the first position of cell c in order k), c = 0 .. G; one workgroup ----
template <int TT>
__global__ __launch_bounds__(256) void sg_starts_kernel(const u32 *__restrict__ cellcnt, u32 *__restrict__ start, const u32 *__restrict__ flag) {
    if (flag[0]) return;
    using C = SgCfg<TT>;
    __shared__ u32 sc[C::G];
    __shared__ u32 part[256];
    constexpr int PER = C::G >= 256 ? C::G / 256 : 1;                  // cells per thread
    for (int k = 0; k < TT; ++k) {
        for (int c = threadIdx.x; c < C::G; c += 256) sc[c] = cellcnt[k * C::G + c];
        __syncthreads();
        u32 mine = 0;
        for (int i = 0; i < PER; ++i) { const int c = (int)threadIdx.x * PER + i; if (c < C::G) mine += sc[c]; }
        part[threadIdx.x] = mine;
        __syncthreads();
        u32 acc = 0;
        for (int w = 0; w < (int)threadIdx.x; ++w) acc += part[w];
        for (int i = 0; i < PER; ++i) {
            const int c = (int)threadIdx.x * PER + i;
            if (c < C::G) { start[k * (C::G + 1) + c] = acc; acc += sc[c]; }
        }
        if ((int)threadIdx.x * PER + PER == C::G || (C::G < 256 && (int)threadIdx.x == C::G - 1)) start[k * (C::G + 1) + C::G] = acc;
        __syncthreads();
    }
}

// ---- inclusive prefix sums of the cell histogram along one axis: innermost (a wave per line of G cells, G / 64 consecutive cells
//      per lane) and the others (a thread per line, neighbouring threads neighbouring cells) ----
template <int TT>
__global__ __launch_bounds__(256) void sg_scan_inner_kernel(u32 *__restrict__ hc, const u32 *__restrict__ flag) {
    if (flag[0]) return;
    using C = SgCfg<TT>;
    constexpr int PER = C::G / 64;
    const int lane = threadIdx.x & 63;
    const size_t line = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    u32 *p = hc + line * C::G + (size_t)lane * PER;
    u32 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = p[i];
#pragma unroll
    for (int i = 1; i < PER; ++i) v[i] += v[i - 1];
    u32 run = v[PER - 1];
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = __shfl_up(run, d);
        if (lane >= d) run += o;
    }
    const u32 before = run - v[PER - 1];
#pragma unroll
    for (int i = 0; i < PER; ++i) p[i] = v[i] + before;
}
template <int TT>
__global__ __launch_bounds__(256) void sg_scan_outer_kernel(u32 *__restrict__ hc, size_t stride, const u32 *__restrict__ flag) {
    if (flag[0]) return;
    using C = SgCfg<TT>;
    const size_t lid = (size_t)blockIdx.x * 256 + threadIdx.x;       // a line: all cells but the scanned axis
    const size_t base = (lid / stride) * stride * C::G + (lid % stride);
    u32 run = 0;
#pragma unroll 8
    for (int c = 0; c < C::G; ++c) {
        run += hc[base + (size_t)c * stride];
        hc[base + (size_t)c * stride] = run;
    }
}

__device__ __forceinline__ u32 sg_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// ---- slab pass K: see the header.  cnt[id][NC] += the classes of the points that share cell_K with target id and no cell of
//      an earlier coordinate ----
template <int TT, int K>
__global__ __launch_bounds__(SG_NT) void sg_pass_kernel(const u32 *__restrict__ rec, const u32 *__restrict__ start, i64 n, int shift,
                                                       const int *__restrict__ slot, u32 *__restrict__ cnt,
                                                       const u32 *__restrict__ flag) {
    if (flag[0]) return;
    using C = SgCfg<TT>;
    constexpr int NC = C::NC, GS = C::COPIES * NC, RW = C::RW;
    __shared__ u32 hist[SG_TG * GS];
    __shared__ u32 xs[SG_TG][8];
    const int tid = threadIdx.x, lane = tid & 63;
    const i64 q0 = (i64)blockIdx.x * SG_TG;
    const int gc = (int)(n - q0 < SG_TG ? n - q0 : SG_TG);
    for (int c = tid; c < SG_TG * GS; c += SG_NT) hist[c] = 0;
    if (tid < SG_TG * RW) {
        const int g = tid / RW, w = tid % RW;
        xs[g][w] = g < gc ? rec[(size_t)(q0 + g) * RW + w] : 0xFFFFFFFFu;
    }
    __syncthreads();
    const u32 cmin = xs[0][K] >> shift, cmax = xs[gc - 1][K] >> shift;
    const i64 pbeg = start[K * (C::G + 1) + cmin], pend = start[K * (C::G + 1) + cmax + 1];
    for (i64 i0 = pbeg + tid; i0 < pend; i0 += (i64)SG_NT * SG_PTS) {
        u32 p[SG_PTS][TT], pid[SG_PTS];
        bool ok[SG_PTS];
#pragma unroll
        for (int k = 0; k < SG_PTS; ++k) {
            const i64 i = i0 + (i64)k * SG_NT;
            ok[k] = i < pend;
            const uint4 *src = reinterpret_cast<const uint4 *>(rec + (size_t)(ok[k] ? i : pbeg) * RW);
            const uint4 a = src[0];
            p[k][0] = a.x; p[k][1] = a.y;
            if constexpr (TT >= 3) p[k][2] = a.z;
            if constexpr (TT == 4) { p[k][3] = a.w; pid[k] = src[1].x; } else pid[k] = a.w;
        }
#pragma unroll 1
        for (int g = 0; g < gc; ++g) {
            u32 x[TT], cx[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) { x[t] = sg_uniform(xs[g][t]); cx[t] = x[t] >> shift; }
            const u32 xid = sg_uniform(xs[g][C::IDW]);
            if (slot && slot[xid] < 0) continue;                        // wave-uniform: not a target of this call
            u32 *hg = hist + g * GS + (lane & (C::COPIES - 1)) * NC;
#pragma unroll
            for (int k = 0; k < SG_PTS; ++k) {
                u32 code = 0;
                bool use = ok[k] && pid[k] != xid && (p[k][K] >> shift) == cx[K];
#pragma unroll
                for (int t = TT - 1; t >= 0; --t) {
                    code = code * 3u + (p[k][t] > x[t] ? 1u : 0u) + (p[k][t] < x[t] ? 2u : 0u);
                    if (t < K) use = use && (p[k][t] >> shift) != cx[t];
                }
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
        for (int r = 0; r < C::COPIES; ++r) v += h[r * NC];
        const u32 id = xs[g][C::IDW];
        if (v && !(slot && slot[id] < 0)) cnt[(size_t)id * NC + c] += v;   // this pass owns the target: no other workgroup adds to it
    }
}

// ---- per target: the strict orthants of cells from the prefix sums + the passes' counts, then the transform of
//      strict_class_wg_kernel (per coordinate the tie slot becomes the sum of the three states; sum of signed squares) ----
template <int TT>
__global__ __launch_bounds__(256) void sg_final_kernel(const u32 *__restrict__ R, i64 n, int shift, const u32 *__restrict__ hc,
                                                      const u32 *__restrict__ cnt, const i64 *__restrict__ targets, i64 m,
                                                      u64 *__restrict__ out, int jcols, const u32 *__restrict__ flag) {
    if (flag[0]) return;
    using C = SgCfg<TT>;
    constexpr int NC = C::NC, G = C::G;
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    const i64 id = targets ? targets[q] : q;
    int cc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) cc[t] = (int)((R[(size_t)t * n + id] & 0x7FFFFFFFu) >> shift);
    long long h[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) h[c] = (long long)cnt[(size_t)id * NC + c];
    // state digits: 1 = above, 2 = below.  below: cells [0, c): the single term F(c); above: cells [c + 1, G): F(G) - F(c + 1),
    // where F(i_0, ..) = points with cell_t < i_t for every t, from the inclusive sums
#pragma unroll
    for (int sv = 0; sv < (1 << TT); ++sv) {                            // bit t of sv: coordinate t is "above"
        long long sum = 0;
        int code = 0, pw = 1;
#pragma unroll
        for (int t = 0; t < TT; ++t) { code += pw * (((sv >> t) & 1) ? 1 : 2); pw *= 3; }
#pragma unroll
        for (int ev = 0; ev < (1 << TT); ++ev) {                        // bit t of ev: the subtracted term of an "above" coordinate
            if (ev & ~sv) continue;
            size_t cell = 0;
            bool zero = false;
            int neg = 0;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int i = ((sv >> t) & 1) ? (((ev >> t) & 1) ? cc[t] + 1 : G) : cc[t];
                neg += (ev >> t) & 1;
                zero = zero || i <= 0;
                cell = (cell << C::LG) | (size_t)(i > 0 ? i - 1 : 0);
            }
            const long long f = zero ? 0ll : (long long)hc[cell];
            sum += (neg & 1) ? -f : f;
        }
        h[code] += sum;
    }
    const long long ties = h[0];                                      // the points that tie with the target everywhere
#pragma unroll
    for (int stride = 1; stride < NC; stride *= 3)
#pragma unroll
        for (int idx = 0; idx < NC / 3; ++idx) {
            const int b = (idx / stride) * stride * 3 + (idx % stride);
            h[b] += h[b + stride] + h[b + 2 * stride];
        }
    long long total = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int strict_digits = 0;
#pragma unroll
        for (int t = 0, d = c; t < TT; ++t, d /= 3) strict_digits += (d % 3) != 0;
        total += (strict_digits & 1) ? -h[c] * h[c] : h[c] * h[c];
    }
    out[q * jcols] = ((u64)total - (u64)ties) / 2;
}

template <int TT, int K>
static void sg_launch_scatter(const u32 *R, i64 n, int shift, u32 *cursor, u32 *rec, u32 *cellcnt, u32 *hc, const u32 *flag, hipStream_t s) {
    const unsigned gs = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL((sg_scatter_kernel<TT, K>), dim3(gs), dim3(256), 0, s, R, n, shift, cursor, rec, cellcnt, hc, flag);
}
template <int TT, int K>
static void sg_launch_pass(const u32 *rec, const u32 *start, i64 n, int shift, const int *slot, u32 *cnt, const u32 *flag, hipStream_t s) {
    hipLaunchKernelGGL((sg_pass_kernel<TT, K>), dim3((unsigned)((n + SG_TG - 1) / SG_TG)), dim3(SG_NT), 0, s,
                       rec + (size_t)K * n * SgCfg<TT>::RW, start, n, shift, slot, cnt, flag);
}

template <int TT>
static int sg_run(const double *Y, i64 n, const i64 *targets, i64 m, u64 *out, int jcols, u32 *flag, void *ws, size_t ws_bytes, hipStream_t s) {
    using C = SgCfg<TT>;
    const bool subset = targets != nullptr;
    const SgPlan p = sg_plan<TT>(n, subset);
    if (!ws || ws_bytes < p.total) return fail(SD_ERR_WORKSPACE, "strict grid workspace too small");
    char *w = (char *)ws;
    u32 *R = (u32 *)(w + p.off_R), *nnan = (u32 *)(w + p.off_nnan), *cursor = (u32 *)(w + p.off_cursor), *rec = (u32 *)(w + p.off_rec);
    u32 *cellcnt = (u32 *)(w + p.off_cellcnt), *start = (u32 *)(w + p.off_start), *hc = (u32 *)(w + p.off_hc), *cnt = (u32 *)(w + p.off_cnt);
    int *slot = subset ? (int *)(w + p.off_slot) : nullptr;
    const int shift = sg_shift(n, C::LG);
    int rc = launch_rank_big_image(Y, TT, n, R, nnan, w + p.off_big, p.big_bytes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(sg_flag_kernel, dim3(1), dim3(64), 0, s, (const u32 *)nnan, TT, flag);
    // zeroed: the cursors, the cell counts, the cell histogram, the class counts (what lies between them is written before it is read)
    SD_HIP(hipMemsetAsync(cursor, 0, (size_t)TT * n * 4, s));
    SD_HIP(hipMemsetAsync(cellcnt, 0, (size_t)TT * C::G * 4, s));
    SD_HIP(hipMemsetAsync(hc, 0, C::CELLS * 4, s));
    SD_HIP(hipMemsetAsync(cnt, 0, (size_t)n * C::NC * 4, s));
    if (subset) {
        SD_HIP(hipMemsetAsync(slot, 0xFF, (size_t)n * 4, s));
        hipLaunchKernelGGL(sg_slots_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, targets, m, slot);
    }
    const u32 *cf = flag;
    sg_launch_scatter<TT, 0>(R, n, shift, cursor, rec, cellcnt, hc, cf, s);
    sg_launch_scatter<TT, 1>(R, n, shift, cursor, rec, cellcnt, hc, cf, s);
    if constexpr (TT >= 3) sg_launch_scatter<TT, 2>(R, n, shift, cursor, rec, cellcnt, hc, cf, s);
    if constexpr (TT >= 4) sg_launch_scatter<TT, 3>(R, n, shift, cursor, rec, cellcnt, hc, cf, s);
    hipLaunchKernelGGL((sg_starts_kernel<TT>), dim3(1), dim3(256), 0, s, (const u32 *)cellcnt, start, cf);
    const size_t lines = C::CELLS / C::G;
    hipLaunchKernelGGL((sg_scan_inner_kernel<TT>), dim3((unsigned)(lines / 4)), dim3(256), 0, s, hc, cf);
    for (int a = 1; a < TT; ++a) {
        size_t stride = 1;
        for (int k = 0; k < a; ++k) stride *= C::G;
        hipLaunchKernelGGL((sg_scan_outer_kernel<TT>), dim3((unsigned)(lines / 256)), dim3(256), 0, s, hc, stride, cf);
    }
    sg_launch_pass<TT, 0>(rec, start, n, shift, slot, cnt, cf, s);
    sg_launch_pass<TT, 1>(rec, start, n, shift, slot, cnt, cf, s);
    if constexpr (TT >= 3) sg_launch_pass<TT, 2>(rec, start, n, shift, slot, cnt, cf, s);
    if constexpr (TT >= 4) sg_launch_pass<TT, 3>(rec, start, n, shift, slot, cnt, cf, s);
    hipLaunchKernelGGL((sg_final_kernel<TT>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const u32 *)R, n, shift, (const u32 *)hc,
                       (const u32 *)cnt, targets, m, out, jcols, cf);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// counts of every target (targets == nullptr: all n points, in order) into out[q * jcols]; flag[0] (the class kernels' NaN flag,
// zeroed by the caller) is set when the data holds a NaN, and nothing is written then
int launch_bd_strict_grid(const double *Y, i64 T, i64 n, const i64 *targets, i64 m, u64 *out, int jcols, u32 *flag, void *ws, size_t ws_bytes,
                          hipStream_t s) {
    if (T == 2) return sg_run<2>(Y, n, targets, m, out, jcols, flag, ws, ws_bytes, s);
    if (T == 3) return sg_run<3>(Y, n, targets, m, out, jcols, flag, ws, ws_bytes, s);
    if (T == 4) return sg_run<4>(Y, n, targets, m, out, jcols, flag, ws, ws_bytes, s);
    return fail(SD_ERR_UNSUPPORTED, "the grid of cells covers two to four coordinates");
}

}  // namespace sd
