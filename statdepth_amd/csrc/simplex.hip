// simplex.hip -- K4: batched point-in-simplex containment counts.
//
// Replaces _is_in_simplex (_containment.py:138-176; an LP feasibility problem handed
// to scipy.optimize.linprog: sum l_k p_k = x, sum l_k = 1, l >= 0) inside the subset
// loops of _pointwisedepth (_pointcloud.py:50-54) and _simplex_depth /
// _simplex_containment (_functional.py:281-285, _containment.py:130-136).
//
// point_in_hull() is the same restatement as oracle_point_in_hull (oracle/oracle.c):
// complete-pivoting elimination of [P^T;1] l = [x;1]; full rank -> unique l, inside
// iff min l >= -tol; rank deficient (degenerate simplex) -> enumerate `rank`-subsets
// (Caratheodory).  Same operation order as the oracle so decisions agree bit for bit
// (this file is compiled with -ffp-contract=off).
//
// Work decomposition: threads own contiguous ranges of the lexicographic subset
// enumeration (unrank once, then step), or one sampled subset each in the sampled
// estimators.  Small dense fp64 systems per thread: VALU bound, no MFMA (systems are
// (d+1)x(d+1) with d <= 8 and data-dependent pivoting).
#include <stdlib.h>

#include "sd_common.h"

namespace sd {

constexpr int SMAX = 10;   // d + 1 <= 9
constexpr int SX_THREADS = 256;

__device__ static bool solve_subset(const double (*R)[SMAX + 1], int rank, const int *cols, double tol,
                                    double rank_eps) {
    double A[SMAX][SMAX + 1];
    for (int r = 0; r < rank; ++r) {
        for (int c = 0; c < rank; ++c) A[r][c] = R[r][cols[c]];
        A[r][rank] = R[r][SMAX];
    }
    for (int c = 0; c < rank; ++c) {
        int p = c;
        double best = fabs(A[c][c]);
        for (int r = c + 1; r < rank; ++r)
            if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); p = r; }
        if (best <= rank_eps) return false;
        if (p != c)
            for (int k = 0; k <= rank; ++k) { double tmp = A[c][k]; A[c][k] = A[p][k]; A[p][k] = tmp; }
        for (int r = c + 1; r < rank; ++r) {
            double f = A[r][c] / A[c][c];
            for (int k = c; k <= rank; ++k) A[r][k] -= f * A[c][k];
        }
    }
    double l[SMAX];
    for (int c = rank - 1; c >= 0; --c) {
        double s = A[c][rank];
        for (int k = c + 1; k < rank; ++k) s -= A[c][k] * l[k];
        l[c] = s / A[c][c];
    }
    for (int c = 0; c < rank; ++c)
        if (!(l[c] >= -tol)) return false;
    return true;
}

// pts: kpts x d (row-major, local), x: d
__device__ static bool point_in_hull(const double *pts, int kpts, int d, const double *x, double tol) {
    int rows = d + 1;
    double R[SMAX][SMAX + 1];
    double scale = 1.0;
    for (int r = 0; r < d; ++r) {
        for (int c = 0; c < kpts; ++c) {
            double v = pts[c * d + r];
            if (v != v) return false;
            R[r][c] = v;
            if (fabs(v) > scale) scale = fabs(v);
        }
        if (x[r] != x[r]) return false;
        R[r][SMAX] = x[r];
        if (fabs(x[r]) > scale) scale = fabs(x[r]);
    }
    for (int c = 0; c < kpts; ++c) R[d][c] = 1.0;
    R[d][SMAX] = 1.0;
    if (isinf(scale)) return false;
    double rank_eps = 1e-10 * scale;
    int rank = 0;
    int lim = rows < kpts ? rows : kpts;
    for (; rank < lim; ++rank) {
        int pr = -1, pc = -1;
        double best = rank_eps;
        for (int r = rank; r < rows; ++r)
            for (int c = rank; c < kpts; ++c)
                if (fabs(R[r][c]) > best) { best = fabs(R[r][c]); pr = r; pc = c; }
        if (pr < 0) break;
        if (pr != rank)
            for (int k = 0; k <= SMAX; ++k) { double tmp = R[rank][k]; R[rank][k] = R[pr][k]; R[pr][k] = tmp; }
        if (pc != rank)
            for (int r = 0; r < rows; ++r) { double tmp = R[r][rank]; R[r][rank] = R[r][pc]; R[r][pc] = tmp; }
        for (int r = rank + 1; r < rows; ++r) {
            double f = R[r][rank] / R[rank][rank];
            if (f != 0.0) {
                for (int c = rank; c < kpts; ++c) R[r][c] -= f * R[rank][c];
                R[r][SMAX] -= f * R[rank][SMAX];
            }
        }
    }
    for (int r = rank; r < rows; ++r)
        if (fabs(R[r][SMAX]) > 1e-7 * scale) return false;
    int cols[SMAX];
    for (int c = 0; c < rank; ++c) cols[c] = c;
    if (rank == kpts) return solve_subset(R, rank, cols, tol, rank_eps);
    for (;;) {
        if (solve_subset(R, rank, cols, tol, rank_eps)) return true;
        int k = rank - 1;
        while (k >= 0 && cols[k] == kpts - rank + k) --k;
        if (k < 0) break;
        ++cols[k];
        for (int l = k + 1; l < rank; ++l) cols[l] = cols[l - 1] + 1;
    }
    return false;
}

__device__ static u64 binom_dev(u64 a, int k) {
    if (k < 0 || (u64)k > a) return 0;
    u64 c = 1;
    for (int j = 1; j <= k; ++j) c = c * (a - (u64)j + 1) / (u64)j;
    return c;
}

// lexicographic unranking of the r-th k-subset of {0..no-1}
__device__ static void unrank_comb(u64 r, int k, i64 no, i64 *idx) {
    i64 c = 0;
    for (int p = 0; p < k; ++p) {
        for (;; ++c) {
            u64 cnt = binom_dev((u64)(no - 1 - c), k - 1 - p);   // subsets starting with c at position p
            if (r < cnt) break;
            r -= cnt;
        }
        idx[p] = c++;
    }
}

__device__ static bool next_comb(i64 *idx, int k, i64 no) {
    int i = k - 1;
    while (i >= 0 && idx[i] == no - k + i) --i;
    if (i < 0) return false;
    ++idx[i];
    for (int l = i + 1; l < k; ++l) idx[l] = idx[l - 1] + 1;
    return true;
}

// splitmix64 finaliser: counter-based draws keyed by (seed, target, sample, draw)
__device__ __host__ static inline u64 mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// The sampled estimators' subsets (round 4): shared sample s is ONE (d+1)-subset of all n rows for every target -- k distinct rows
// drawn with the counter generator keyed by (seed, s): index = floor(r * n / 2^64), rejection on duplicates, sorted ascending -- so
// that its factorisation can be shared by all targets (sxw_factor_kernel / sxw_apply_kernel).  A target uses the FIRST `samples`
// shared subsets that do not contain it (s = 0, 1, 2, ... in order): each is uniform over the subsets of the other rows.
// Restated identically in oracle.c (sample_rows / oracle_simplex_sampled).
constexpr u64 SX_SHARED_KEY = 0xFFFFFFFFFFFFFFFFull;
__device__ static void sample_rows(u64 seed, u64 keyv, u64 sample, int k, i64 n, i64 *idx) {
    u64 key = mix64(seed ^ mix64(keyv * 0xD1342543DE82EF95ull + sample));
    u64 ctr = 0;
    for (int p = 0; p < k;) {
        u64 r = mix64(key + ctr++);
        // index in [0, n) as the high half of r * n (a 64 x 64 -> 128 bit product: a handful of instructions; the 64-bit
        // modulo it replaces was ~480 of the 715 lane-instructions of a sampled 4-point test, profiles/r02_issue_roofline.json)
        i64 c = (i64)__umul64hi(r, (u64)n);
        bool dup = false;
        for (int l = 0; l < p; ++l) dup |= (idx[l] == c);
        if (!dup) idx[p++] = c;
    }
    for (int a = 1; a < k; ++a) {   // insertion sort
        i64 v = idx[a];
        int b = a - 1;
        while (b >= 0 && idx[b] > v) { idx[b + 1] = idx[b]; --b; }
        idx[b + 1] = v;
    }
}
// the next shared subset from index *s on that does not contain row tg; *s is left behind it
__device__ static void sample_next_valid(u64 seed, i64 tg, u64 *s, int k, i64 n, i64 *idx) {
    for (;;) {
        sample_rows(seed, SX_SHARED_KEY, (*s)++, k, n, idx);
        bool member = false;
        for (int l = 0; l < k; ++l) member |= idx[l] == tg;
        if (!member) return;
    }
}

__device__ __forceinline__ u64 block_sum_u64(u64 v, u64 *scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 r = 0;
    if (threadIdx.x == 0)
        for (int k = 0; k < SX_THREADS / 64; ++k) r += scratch[k];
    return r;
}

// Which rows of P a block's subsets are drawn from, and where its target lies:
//   default   -- the n - 1 rows other than targets[q] (the subset loops of _pointwisedepth / _simplex_depth);
//   members   -- an explicit block of rows per target (int32[nb][bs], -1 padded; the block's OTHERS first, its target
//                LAST): the K-block sampled estimator (_samplepointwisedepth, _pointcloud.py:107-121) in one launch;
//   Q         -- EXTERNAL targets (m x d): every row of P is an "other" (the homogeneity coefficients' depth of a
//                point of G inside F u {g}, homogeneity.py:172-186, for all of G at once).  Point clouds only (T = 0).
struct SxSel {
    const int *members;
    int bs;
    const double *Q;
    int d;
};

struct SxBlock {
    const int *mem;
    const double *xq;
    i64 tg, no;
    u64 total, key;
};

__device__ __forceinline__ SxBlock sx_block(const SxSel &sel, const i64 *targets, i64 n, i64 q, int k, u64 total) {
    SxBlock b;
    b.mem = nullptr;
    b.xq = nullptr;
    if (sel.members) {
        const int *mem = sel.members + q * sel.bs;
        int cnt = 0;
        while (cnt < sel.bs && mem[cnt] >= 0) ++cnt;
        b.mem = mem;
        b.no = cnt > 0 ? cnt - 1 : 0;
        b.tg = cnt > 0 ? mem[cnt - 1] : 0;
        b.total = cnt > 0 ? binom_dev((u64)b.no, k) : 0;        // exhaustive inside the block
        b.key = (u64)b.tg;
    } else if (sel.Q) {
        b.xq = sel.Q + q * sel.d;
        b.no = n;
        b.tg = -1;
        b.total = total;
        b.key = (u64)(n + q);
    } else {
        b.tg = targets ? targets[q] : q;
        b.no = n - 1;
        b.total = total;
        b.key = (u64)b.tg;
    }
    return b;
}

__device__ __forceinline__ i64 sx_src(const SxBlock &b, i64 i, bool sampled = false) {
    if (b.mem) return b.mem[i];
    if (b.xq || sampled) return i;                               // (a sampled subset holds rows, never the target's)
    return i < b.tg ? i : i + 1;                                 // skip the target itself
}

// grid = (chunks, m).  T == 0 selects the pointcloud form (P is n x d);
// otherwise P is n x T x d and c counts timepoints (relax / strict reduction).
__global__ __launch_bounds__(SX_THREADS) void simplex_kernel(
    const double *__restrict__ P, i64 n, i64 T, int d, const i64 *__restrict__ targets, int relax, double tol,
    u64 total, u64 per_thread, i64 samples, u64 seed, i64 q0, u64 *__restrict__ out, SxSel sel) {
    __shared__ u64 scratch[SX_THREADS / 64];
    i64 q = q0 + blockIdx.y;
    int k = d + 1;
    const SxBlock blk = sx_block(sel, targets, n, q, k, total);
    const i64 tg = blk.tg, no = blk.no;
    total = blk.total;
    u64 tid = (u64)blockIdx.x * SX_THREADS + threadIdx.x;
    u64 first = tid * per_thread;
    u64 acc = 0;
    if (first < total) {
        u64 last = first + per_thread < total ? first + per_thread : total;
        i64 idx[SMAX];
        double pts[SMAX * 8], x[8];
        if (samples < 0) unrank_comb(first, k, no, idx);
        u64 snext = 0;                                                  // sampled: the shared index behind the last subset taken
        if (samples >= 0)
            for (u64 r = 0; r < first; ++r) sample_next_valid(seed, tg, &snext, k, n, idx);   // this thread's first valid subset
        i64 TT = T > 0 ? T : 1;
        for (u64 r = first; r < last; ++r) {
            if (samples >= 0) sample_next_valid(seed, tg, &snext, k, n, idx);
            u64 cnt = 0;
            for (i64 t = 0; t < TT; ++t) {
                for (int c = 0; c < k; ++c) {
                    i64 src = sx_src(blk, idx[c], samples >= 0);
                    const double *pp = P + (src * TT + t) * d;
                    for (int e = 0; e < d; ++e) pts[c * d + e] = pp[e];
                }
                const double *xx = blk.xq ? blk.xq : P + (tg * TT + t) * d;
                for (int e = 0; e < d; ++e) x[e] = xx[e];
                cnt += point_in_hull(pts, k, d, x, tol);
            }
            acc += (T > 0 && !relax) ? (cnt / (u64)TT) : cnt;
            if (samples < 0 && r + 1 < last) next_comb(idx, k, no);
        }
    }
    u64 tot = block_sum_u64(acc, scratch);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[q], tot);
}

// ---------------------------------------------------------------------------------------------------
// Register-resident form of the same test for a compile-time dimension D and a NON-degenerate simplex
// (D + 1 points of full rank -- every simplex of continuous data).  The (D+1) x (D+2) system lives in VGPRs:
// every array index below is a compile-time constant, and the data-dependent row / column exchanges of the
// pivoting are predicated exchanges with each candidate (v_cndmask) instead of indexed moves through scratch
// memory.  The arithmetic is, operation for operation and in the same order, that of point_in_hull() /
// solve_subset() above (pivot search order and strict ">" included), so the decision is the same bit for bit.
// Returns 0 / 1 (outside / inside) or 2: the simplex is rank deficient, the caller falls back to point_in_hull().
// ---------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ int hull_full_rank(double (&R)[D + 1][D + 2], double scale, double tol) {
    constexpr int K = D + 1;                  // points = rows = K; column K is the right-hand side
    if (isinf(scale)) return 0;
    const double rank_eps = 1e-10 * scale;
    // ---- complete-pivoting elimination (point_in_hull) ----
#pragma unroll
    for (int rank = 0; rank < K; ++rank) {
        int pr = -1, pc = -1;
        double best = rank_eps;
#pragma unroll
        for (int r = rank; r < K; ++r)
#pragma unroll
            for (int c = rank; c < K; ++c) {
                const double a = fabs(R[r][c]);
                const bool gt = a > best;
                best = gt ? a : best;
                pr = gt ? r : pr;
                pc = gt ? c : pc;
            }
        if (pr < 0) return 2;                 // rank deficient: Caratheodory enumeration in the generic code
#pragma unroll
        for (int rr = rank + 1; rr < K; ++rr) {           // exchange rows rank <-> pr
            const bool sw = pr == rr;
#pragma unroll
            for (int k = 0; k <= K; ++k) {
                const double a = R[rank][k], b = R[rr][k];
                R[rank][k] = sw ? b : a;
                R[rr][k] = sw ? a : b;
            }
        }
#pragma unroll
        for (int cc = rank + 1; cc < K; ++cc) {           // exchange columns rank <-> pc
            const bool sw = pc == cc;
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const double a = R[r][rank], b = R[r][cc];
                R[r][rank] = sw ? b : a;
                R[r][cc] = sw ? a : b;
            }
        }
#pragma unroll
        for (int r = rank + 1; r < K; ++r) {
            const double f = R[r][rank] / R[rank][rank];
            const bool nz = f != 0.0;
#pragma unroll
            for (int c = rank; c <= K; ++c) {
                const double v = R[r][c] - f * R[rank][c];
                R[r][c] = nz ? v : R[r][c];
            }
        }
    }
    // ---- solve_subset with all K columns: partial pivoting on the (already triangular) system ----
#pragma unroll
    for (int c = 0; c < K; ++c) {
        int p = c;
        double best = fabs(R[c][c]);
#pragma unroll
        for (int r = c + 1; r < K; ++r) {
            const double a = fabs(R[r][c]);
            const bool gt = a > best;
            best = gt ? a : best;
            p = gt ? r : p;
        }
        if (best <= rank_eps) return 0;
#pragma unroll
        for (int rr = c + 1; rr < K; ++rr) {
            const bool sw = p == rr;
#pragma unroll
            for (int k = 0; k <= K; ++k) {
                const double a = R[c][k], b = R[rr][k];
                R[c][k] = sw ? b : a;
                R[rr][k] = sw ? a : b;
            }
        }
#pragma unroll
        for (int r = c + 1; r < K; ++r) {
            const double f = R[r][c] / R[c][c];
#pragma unroll
            for (int k = c; k <= K; ++k) R[r][k] -= f * R[c][k];
        }
    }
    double l[K];
#pragma unroll
    for (int c = K - 1; c >= 0; --c) {
        double sacc = R[c][K];
#pragma unroll
        for (int k = c + 1; k < K; ++k) sacc -= R[c][k] * l[k];
        l[c] = sacc / R[c][c];
    }
    bool in = true;
#pragma unroll
    for (int c = 0; c < K; ++c) in = in && (l[c] >= -tol);
    return in ? 1 : 0;
}

// grid = (chunks, m), the work decomposition of simplex_kernel; D = d known at compile time.
template <int D>
__global__ __launch_bounds__(SX_THREADS) void simplex_kernel_fast(
    const double *__restrict__ P, i64 n, i64 T, const i64 *__restrict__ targets, int relax, double tol,
    u64 total, u64 per_thread, i64 samples, u64 seed, i64 q0, u64 *__restrict__ out, SxSel sel) {
    __shared__ u64 scratch[SX_THREADS / 64];
    constexpr int K = D + 1;
    const i64 q = q0 + blockIdx.y;
    const SxBlock blk = sx_block(sel, targets, n, q, K, total);
    const i64 tg = blk.tg, no = blk.no;
    total = blk.total;
    const u64 tid = (u64)blockIdx.x * SX_THREADS + threadIdx.x;
    const u64 first = tid * per_thread;
    u64 acc = 0;
    if (first < total) {
        const u64 last = first + per_thread < total ? first + per_thread : total;
        i64 idx[SMAX];
        if (samples < 0) unrank_comb(first, K, no, idx);
        u64 snext = 0;                                                  // sampled: the shared index behind the last subset taken
        if (samples >= 0)
            for (u64 r = 0; r < first; ++r) sample_next_valid(seed, tg, &snext, K, n, idx);   // this thread's first valid subset
        const i64 TT = T > 0 ? T : 1;
        for (u64 r = first; r < last; ++r) {
            if (samples >= 0) sample_next_valid(seed, tg, &snext, K, n, idx);
            u64 cnt = 0;
            for (i64 t = 0; t < TT; ++t) {
                double R[K][K + 1];
                double scale = 1.0;
                bool anynan = false;
                const double *xx = blk.xq ? blk.xq : P + (tg * TT + t) * D;
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const i64 src = sx_src(blk, idx[c], samples >= 0);
                    const double *pp = P + (src * TT + t) * D;
#pragma unroll
                    for (int e = 0; e < D; ++e) {
                        const double v = pp[e];
                        anynan |= v != v;
                        R[e][c] = v;
                        scale = fabs(v) > scale ? fabs(v) : scale;
                    }
                    R[D][c] = 1.0;
                }
#pragma unroll
                for (int e = 0; e < D; ++e) {
                    const double v = xx[e];
                    anynan |= v != v;
                    R[e][K] = v;
                    scale = fabs(v) > scale ? fabs(v) : scale;
                }
                R[D][K] = 1.0;
                int res = anynan ? 0 : hull_full_rank<D>(R, scale, tol);
                if (res == 2) {                                         // degenerate simplex: generic code, from the data
                    double pts[SMAX * 8], x[8];
                    for (int c = 0; c < K; ++c) {
                        const i64 src = sx_src(blk, idx[c], samples >= 0);
                        const double *pp = P + (src * TT + t) * D;
                        for (int e = 0; e < D; ++e) pts[c * D + e] = pp[e];
                    }
                    for (int e = 0; e < D; ++e) x[e] = xx[e];
                    res = point_in_hull(pts, K, D, x, tol) ? 1 : 0;
                }
                cnt += (u64)res;
            }
            acc += (T > 0 && !relax) ? (cnt / (u64)TT) : cnt;
            if (samples < 0 && r + 1 < last) next_comb(idx, K, no);
        }
    }
    u64 tot = block_sum_u64(acc, scratch);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[q], tot);
}


// ---------------------------------------------------------------------------------------------------
// Sampled estimators, every target's sample s the SAME subset (sample_rows above): the subset's matrix [P_S^T; 1] is
// eliminated ONCE and every target only carries its right-hand side [x; 1] through the recorded row operations.
// point_in_hull() / solve_subset() touch the right-hand side in a fixed sequence of operations that depends on the matrix
// alone -- the row exchanges of the complete-pivoting elimination, b_r -= f b_k for its multipliers f (skipped where f = 0),
// the second (partial-pivoting) elimination of solve_subset on the triangular system with its own multipliers, the back
// substitution -- so replaying that sequence gives the barycentric coordinates of the per-target elimination BIT FOR BIT,
// at about 2 K^2 multiply-subtracts + K divisions per test instead of a K x K elimination (d = 8: ~400 instructions
// against ~7 500; d = 3: ~90 against ~700, and no random gathers: the members are read once per subset).
// What depends on the target besides the right-hand side is `scale` (it includes |x|) through rank_eps = 1e-10 scale: a
// target whose own scale would put the smallest pivot at or below rank_eps, and a subset that is rank deficient, holds a NaN
// or an infinity, or exchanges rows in the second elimination, take the per-target code (point_in_hull), decision for decision
// the old kernel's.  A target that is a MEMBER of the subset skips it and takes a spare shared subset instead (the first
// `samples` shared subsets that do not contain it: sxw_members_kernel counts what it skips, sxw_spare_kernel makes it up).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double scw_u(double v) {                    // a wave-uniform double into SGPRs
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)b), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---------------------------------------------------------------------------------------------------
// Two kernels over a caller's workspace (sd_simplex_sampled_workspace_bytes): the records go to HBM, a thread per (timepoint,
// sample) with the whole register file for its elimination, and the replay reads a record as one coalesced load per wave.
// Batches of pairs as large as the workspace holds.  Without a workspace the per-target kernels run (same subsets).
// ---------------------------------------------------------------------------------------------------
// the per-target code for a target the shared replay cannot serve (see above): out of line, so that its scratch arrays and
// registers are not the replay loop's
template <int D>
__device__ __noinline__ int sxw_special(const double *P, i64 TT, i64 t, const int *mem, const double *xv, double tol) {
    constexpr int K = D + 1;
    double pts[SMAX * 8], x[8];
    for (int c = 0; c < K; ++c) {
        const double *pp = P + ((i64)mem[c] * TT + t) * D;
        for (int e = 0; e < D; ++e) pts[c * D + e] = pp[e];
    }
    for (int e = 0; e < D; ++e) x[e] = xv[e];
    return point_in_hull(pts, K, D, x, tol) ? 1 : 0;
}

template <int D> struct SxwCfg {
    static constexpr int K = D + 1, NL = K * (K - 1) / 2, NU = K * (K + 1) / 2;
    static constexpr int DW = 2 * NL + NU + 2;                          // doubles: f1, f2, u, scale, minpiv
    static constexpr int IW = ((2 * K + 1 + 3) / 4) * 4;                // ints: perm, members, status; padded to 16 bytes
    static constexpr size_t REC = (size_t)DW * 8 + (size_t)IW * 4;      // bytes of a record
    static constexpr int TPT = D >= 5 ? 2 : 4;                          // targets per thread of the replay
};

// pairs [w0, w0 + cnt) of the call, pair w = timepoint w / samples, sample w % samples; one thread each
template <int D>
__global__ __launch_bounds__(64) void sxw_factor_kernel(const double *__restrict__ P, i64 n, i64 T, i64 samples, u64 seed, i64 w0, i64 cnt,
                                                        double *__restrict__ recd, int *__restrict__ reci, double *__restrict__ m1g) {
    using C = SxwCfg<D>;
    constexpr int K = C::K, NL = C::NL, NU = C::NU, DW = C::DW, IW = C::IW;
    const i64 v = (i64)blockIdx.x * 64 + threadIdx.x;
    if (v >= cnt) return;
    const i64 TT = T > 0 ? T : 1;
    const i64 t = (w0 + v) / samples, sidx = (w0 + v) % samples;
    i64 idx[K];
    sample_rows(seed, SX_SHARED_KEY, (u64)sidx, K, n, idx);
    double *rd = recd + (size_t)v * DW;
    int *ri = reci + (size_t)v * IW;
    double *m1l = m1g + (size_t)v * (K * (K - 1));                      // the first elimination's multipliers by ORIGINAL row
    double R[K][K];
    int rowid[K];
    double scale = 1.0;
    bool irregular = false;
#pragma unroll
    for (int c = 0; c < K; ++c) {
        const double *pp = P + (idx[c] * TT + t) * D;
#pragma unroll
        for (int e = 0; e < D; ++e) {
            const double x = pp[e];
            irregular |= x != x;
            R[e][c] = x;
            scale = fabs(x) > scale ? fabs(x) : scale;
        }
        R[D][c] = 1.0;
    }
#pragma unroll
    for (int i = 0; i < K; ++i) rowid[i] = i;
    irregular |= isinf(scale);
    const double rank_eps = 1e-10 * scale;
    double minpiv = __builtin_huge_val();
#pragma unroll
    for (int rank = 0; rank < K; ++rank) {
        int pr = -1, pc = -1;
        double best = rank_eps;
#pragma unroll
        for (int r = rank; r < K; ++r)
#pragma unroll
            for (int c = rank; c < K; ++c) {
                const double av = fabs(R[r][c]);
                const bool gt = av > best;
                best = gt ? av : best;
                pr = gt ? r : pr;
                pc = gt ? c : pc;
            }
        irregular |= pr < 0;                                            // rank deficient: the per-target code's business
        minpiv = best < minpiv ? best : minpiv;
#pragma unroll
        for (int rr = rank + 1; rr < K; ++rr) {                         // exchange rows rank <-> pr
            const bool sw = pr == rr;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double x0 = R[rank][k], x1 = R[rr][k];
                R[rank][k] = sw ? x1 : x0;
                R[rr][k] = sw ? x0 : x1;
            }
            const int i0 = rowid[rank], i1 = rowid[rr];
            rowid[rank] = sw ? i1 : i0;
            rowid[rr] = sw ? i0 : i1;
        }
#pragma unroll
        for (int cc = rank + 1; cc < K; ++cc) {                         // exchange columns rank <-> pc
            const bool sw = pc == cc;
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const double x0 = R[r][rank], x1 = R[r][cc];
                R[r][rank] = sw ? x1 : x0;
                R[r][cc] = sw ? x0 : x1;
            }
        }
#pragma unroll
        for (int r = rank + 1; r < K; ++r) {
            const double f = R[r][rank] / R[rank][rank];
            if (rank < K - 1) m1l[rowid[r] * (K - 1) + rank] = f;
            const bool nz = f != 0.0;
#pragma unroll
            for (int c = rank; c < K; ++c) {
                const double x = R[r][c] - f * R[rank][c];
                R[r][c] = nz ? x : R[r][c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < K; ++c) {
        const double best = fabs(R[c][c]);
#pragma unroll
        for (int r = c + 1; r < K; ++r) irregular |= fabs(R[r][c]) > best;
        minpiv = best < minpiv ? best : minpiv;
#pragma unroll
        for (int r = c + 1; r < K; ++r) {
            const double f = R[r][c] / R[c][c];
            rd[NL + r * (r - 1) / 2 + c] = f;
#pragma unroll
            for (int k = c; k < K; ++k) R[r][k] -= f * R[c][k];
        }
    }
    __threadfence_block();                                              // this thread's own multipliers, read back by row below
#pragma unroll
    for (int i = 1; i < K; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) rd[i * (i - 1) / 2 + k] = m1l[rowid[i] * (K - 1) + k];
    {
        int w = 0;
#pragma unroll
        for (int c = 0; c < K; ++c)
#pragma unroll
            for (int k = c; k < K; ++k) rd[2 * NL + w++] = R[c][k];
    }
    rd[2 * NL + NU] = scale;
    rd[2 * NL + NU + 1] = minpiv;
#pragma unroll
    for (int i = 0; i < K; ++i) { ri[i] = rowid[i]; ri[K + i] = (int)idx[i]; }
    ri[2 * K] = irregular ? 1 : 0;
}

// replay of pairs [w0, w0 + cnt) on a tile of targets: grid = (tiles, splits); counts leave as one atomic per target
template <int D>
__global__ __launch_bounds__(256) void sxw_apply_kernel(const double *__restrict__ P, i64 n, i64 T, const i64 *__restrict__ targets, i64 m,
                                                       double tol, i64 samples, u64 seed, i64 w0, i64 cnt,
                                                       const double *__restrict__ recd, const int *__restrict__ reci,
                                                       u64 *__restrict__ out) {
    using C = SxwCfg<D>;
    constexpr int K = C::K, NL = C::NL, NU = C::NU, DW = C::DW, IW = C::IW, TPT = C::TPT;
    const int tid = threadIdx.x;
    const i64 TT = T > 0 ? T : 1;
    const i64 q0 = (i64)blockIdx.x * (256 * TPT);
    i64 tg[TPT];
    bool live[TPT];
    u64 acc[TPT];
#pragma unroll
    for (int p = 0; p < TPT; ++p) {
        const i64 q = q0 + tid + (i64)p * 256;
        live[p] = q < m;
        tg[p] = live[p] ? (targets ? targets[q] : q) : 0;
        acc[p] = 0;
    }
    // this workgroup's slice of the batch: contiguous (the timepoint changes rarely)
    const i64 per = (cnt + gridDim.y - 1) / gridDim.y;
    const i64 vbeg = (i64)blockIdx.y * per, vend = vbeg + per < cnt ? vbeg + per : cnt;
    i64 tcur = -1;
    double xx[TPT][K], sx[TPT];
    // A record travels as ONE coalesced load per wave (lane L holds doubles L and 64 + L, and int L), prefetched a pair ahead;
    // its values reach the arithmetic through v_readlane with compile-time lane numbers: wave-uniform operands in SGPRs without
    // the scalar cache's latency per value (a scalar load per value, waited for one by one, was ten times slower) and without LDS.
    static_assert(DW <= 128 && IW <= 64, "a record fits two doubles and one int per lane");
    const int lane = tid & 63;
    double c0 = 0.0, c1 = 0.0, n0 = 0.0, n1 = 0.0;
    int ci = 0, ni = 0;
    auto fetch = [&](i64 v, double &d0, double &d1, int &di) {
        const double *rd = recd + (size_t)v * DW;
        d0 = lane < DW ? rd[lane] : 0.0;
        d1 = 64 + lane < DW ? rd[64 + lane] : 0.0;
        di = lane < IW ? reci[(size_t)v * IW + lane] : 0;
    };
    if (vbeg < vend) fetch(vbeg, n0, n1, ni);
#pragma unroll 1
    for (i64 v = vbeg; v < vend; ++v) {
        c0 = n0; c1 = n1; ci = ni;
        if (v + 1 < vend) fetch(v + 1, n0, n1, ni);
        auto RD = [&](int k) -> double {                                // k: a compile-time constant wherever this is called
            const unsigned long long bits = (unsigned long long)__double_as_longlong(k < 64 ? c0 : c1);
            const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)bits, k & 63);
            const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(bits >> 32), k & 63);
            return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        };
        auto RI = [&](int k) -> int { return __builtin_amdgcn_readlane(ci, k); };
        const i64 t = (w0 + v) / samples;
        if (t != tcur) {                                                // block-uniform
            tcur = t;
#pragma unroll
            for (int p = 0; p < TPT; ++p) {
                const double *xp = P + (tg[p] * TT + t) * D;
                double sc = 1.0;
                bool isn = false;
#pragma unroll
                for (int e = 0; e < D; ++e) {
                    const double x = xp[e];
                    xx[p][e] = x;
                    isn |= x != x;
                    sc = fabs(x) > sc ? fabs(x) : sc;
                }
                xx[p][D] = 1.0;
                sx[p] = isn ? __builtin_nan("") : sc;
            }
        }
        const double scaleS = RD(2 * NL + NU), minpiv = RD(2 * NL + NU + 1);
        const int status = RI(2 * K);
        int perm[K], mem[K];
#pragma unroll
        for (int i = 0; i < K; ++i) { perm[i] = RI(i); mem[i] = RI(K + i); }
        bool special[TPT], in[TPT], skip[TPT];
        double b[TPT][K], l[TPT][K];
#pragma unroll
        for (int p = 0; p < TPT; ++p) {
            bool member = false;
#pragma unroll
            for (int i = 0; i < K; ++i) member |= (i64)mem[i] == tg[p];
            const double sc = sx[p] > scaleS ? sx[p] : scaleS;          // NaN x: sc = scaleS, and the per-target code says no
            special[p] = status != 0 || !(sx[p] == sx[p]) || !(minpiv > 1e-10 * sc) || isinf(sc);
            skip[p] = member;                                           // this target takes a spare subset instead (sxw_spare_kernel)
            in[p] = true;
#pragma unroll
            for (int i = 0; i < K; ++i) b[p][i] = xx[p][perm[i]];
        }
        if (status == 0) {                                              // block-uniform
            int w = 0;
#pragma unroll
            for (int i = 1; i < K; ++i)
#pragma unroll
                for (int k = 0; k < i; ++k) {
                    const double f = RD(w++);
                    if (f != 0.0) {                                     // block-uniform
#pragma unroll
                        for (int p = 0; p < TPT; ++p) b[p][i] -= f * b[p][k];
                    }
                }
            w = 0;
#pragma unroll
            for (int i = 1; i < K; ++i)
#pragma unroll
                for (int k = 0; k < i; ++k) {
                    const double f = RD(NL + w++);
#pragma unroll
                    for (int p = 0; p < TPT; ++p) b[p][i] -= f * b[p][k];
                }
#pragma unroll
            for (int c = K - 1; c >= 0; --c) {
                const int base = 2 * NL + c * K - c * (c - 1) / 2;      // u[c][c]
                double sacc[TPT];
#pragma unroll
                for (int p = 0; p < TPT; ++p) sacc[p] = b[p][c];
#pragma unroll
                for (int k = c + 1; k < K; ++k) {
                    const double u = RD(base + (k - c));
#pragma unroll
                    for (int p = 0; p < TPT; ++p) sacc[p] -= u * l[p][k];
                }
                const double dg = RD(base);
#pragma unroll
                for (int p = 0; p < TPT; ++p) {
                    l[p][c] = sacc[p] / dg;
                    in[p] = in[p] && (l[p][c] >= -tol);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < TPT; ++p) {
            if (!live[p] || skip[p]) continue;
            int inside = in[p] ? 1 : 0;
            if (special[p]) {
                double xv[D];
#pragma unroll
                for (int e = 0; e < D; ++e) xv[e] = xx[p][e];
                inside = sxw_special<D>(P, TT, t, mem, xv, tol);
            }
            acc[p] += (u64)inside;
        }
    }
#pragma unroll
    for (int p = 0; p < TPT; ++p)
        if (live[p] && acc[p]) atomicAdd(&out[q0 + tid + (i64)p * 256], acc[p]);
}

// cmem[row] = shared subsets s < samples that contain the row: what a target of that row skips in the replay
__global__ __launch_bounds__(256) void sxw_members_kernel(i64 n, i64 samples, u64 seed, int k, int *__restrict__ cmem) {
    const i64 sidx = (i64)blockIdx.x * 256 + threadIdx.x;
    if (sidx >= samples) return;
    i64 idx[SMAX];
    sample_rows(seed, SX_SHARED_KEY, (u64)sidx, k, n, idx);
    for (int i = 0; i < k; ++i) atomicAdd(&cmem[idx[i]], 1);
}

// the spare subsets of the targets that skipped some: a thread per (target, timepoint) walks the shared subsets from index
// `samples` on, takes those that do not contain its target until it has made up what it skipped, and tests each with the
// per-target code (hull_full_rank, point_in_hull for a degenerate simplex).  K / n of the tests go this way.
template <int D>
__global__ __launch_bounds__(256) void sxw_spare_kernel(const double *__restrict__ P, i64 n, i64 T, const i64 *__restrict__ targets, i64 m,
                                                       double tol, i64 samples, u64 seed, const int *__restrict__ cmem,
                                                       u64 *__restrict__ out) {
    constexpr int K = D + 1;
    const i64 TT = T > 0 ? T : 1;
    const i64 id = (i64)blockIdx.x * 256 + threadIdx.x;
    if (id >= m * TT) return;
    const i64 q = id / TT, t = id % TT;
    const i64 tg = targets ? targets[q] : q;
    const int need = cmem[tg];
    if (need == 0) return;
    u64 snext = (u64)samples, acc = 0;
    const double *xx = P + (tg * TT + t) * D;
    for (int got = 0; got < need; ++got) {
        i64 idx[K];
        sample_next_valid(seed, tg, &snext, K, n, idx);
        double R[K][K + 1];
        double scale = 1.0;
        bool anynan = false;
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const double *pp = P + (idx[c] * TT + t) * D;
#pragma unroll
            for (int e = 0; e < D; ++e) {
                const double v = pp[e];
                anynan |= v != v;
                R[e][c] = v;
                scale = fabs(v) > scale ? fabs(v) : scale;
            }
            R[D][c] = 1.0;
        }
#pragma unroll
        for (int e = 0; e < D; ++e) {
            const double v = xx[e];
            anynan |= v != v;
            R[e][K] = v;
            scale = fabs(v) > scale ? fabs(v) : scale;
        }
        R[D][K] = 1.0;
        int res = anynan ? 0 : hull_full_rank<D>(R, scale, tol);
        if (res == 2) {                                                 // degenerate simplex: generic code, from the data
            double pts[SMAX * 8], x[8];
            for (int c = 0; c < K; ++c) {
                const double *pp = P + (idx[c] * TT + t) * D;
                for (int e = 0; e < D; ++e) pts[c * D + e] = pp[e];
            }
            for (int e = 0; e < D; ++e) x[e] = xx[e];
            res = point_in_hull(pts, K, D, x, tol) ? 1 : 0;
        }
        acc += (u64)res;
    }
    if (acc) atomicAdd(&out[q], acc);
}

template <int D>
static size_t sxw_pair_bytes() { return SxwCfg<D>::REC + (size_t)(D + 1) * D * 8; }
size_t simplex_sampled_workspace_bytes(i64 n, i64 T, int d, i64 samples) {
    if (d < 1 || d > 8 || samples <= 0) return 0;
    const i64 TT = T > 0 ? T : 1;
    size_t per = 0;
    switch (d) {
#define SX_PB(D_) case D_: per = sxw_pair_bytes<D_>(); break;
        SX_PB(1) SX_PB(2) SX_PB(3) SX_PB(4) SX_PB(5) SX_PB(6) SX_PB(7) SX_PB(8)
#undef SX_PB
    }
    i64 pairs = samples * TT;
    if (pairs > ((i64)1 << 20)) pairs = (i64)1 << 20;                   // batches of up to 2^20 pairs: at most ~1.7 GB at d = 8
    return (size_t)pairs * per + align_up((size_t)n * 4, 256) + 1024;
}

template <int D>
static int launch_simplex_ws(const double *P, i64 n, i64 T, const i64 *targets, i64 m, double tol, i64 samples, u64 seed, u64 *out,
                             void *ws, size_t ws_bytes, hipStream_t s) {
    using C = SxwCfg<D>;
    const i64 TT = T > 0 ? T : 1;
    const i64 pairs = samples * TT;
    const size_t per = sxw_pair_bytes<D>();
    const size_t cm_bytes = align_up((size_t)n * 4, 256);
    if (ws_bytes < 768 + cm_bytes + (size_t)(pairs < 256 ? pairs : 256) * per) return fail(SD_ERR_WORKSPACE, "sampled simplex workspace too small");
    i64 batch = (i64)((ws_bytes - 768 - cm_bytes) / per);
    if (batch > pairs) batch = pairs;
    char *w = (char *)(((size_t)ws + 255) / 256 * 256);
    int *cmem = (int *)w;
    double *recd = (double *)(w + cm_bytes);
    double *m1g = recd + (size_t)batch * C::DW;
    int *reci = (int *)(m1g + (size_t)batch * (C::K * (C::K - 1)));
    SD_HIP(hipMemsetAsync(cmem, 0, (size_t)n * 4, s));
    hipLaunchKernelGGL(sxw_members_kernel, dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, s, n, samples, seed, C::K, cmem);
    const i64 tiles = (m + 256 * C::TPT - 1) / (256 * C::TPT);
    for (i64 w0 = 0; w0 < pairs; w0 += batch) {
        const i64 cnt = pairs - w0 < batch ? pairs - w0 : batch;
        hipLaunchKernelGGL((sxw_factor_kernel<D>), dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, s, P, n, T, samples, seed, w0, cnt, recd, reci, m1g);
        i64 splits = (4096 + tiles - 1) / tiles;                        // about 4 096 workgroups in all
        if (splits > cnt) splits = cnt;
        if (splits > 65535) splits = 65535;
        hipLaunchKernelGGL((sxw_apply_kernel<D>), dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, s, P, n, T, targets, m, tol, samples,
                           seed, w0, cnt, (const double *)recd, (const int *)reci, out);
    }
    hipLaunchKernelGGL((sxw_spare_kernel<D>), dim3((unsigned)((m * TT + 255) / 256)), dim3(256), 0, s, P, n, T, targets, m, tol, samples, seed,
                       (const int *)cmem, out);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

static int launch_simplex_common(const double *P, i64 n, i64 T, int d, const i64 *targets, i64 m, int relax,
                                 double tol, i64 samples, u64 seed, u64 *out, hipStream_t s,
                                 SxSel sel = SxSel{nullptr, 0, nullptr, 0}, void *ws = nullptr, size_t ws_bytes = 0) {
    SD_HIP(hipMemsetAsync(out, 0, sizeof(u64) * m, s));
    u64 total;
    // subsets per target: of the n - 1 others / of all n rows (external targets) / of the largest block's others
    const i64 pool = sel.members ? (i64)sel.bs - 1 : (sel.Q ? n : n - 1);
    if (samples >= 0) total = (u64)samples;
    else if (!binom_u64_checked((u64)pool, d + 1, &total)) return fail(SD_ERR_OVERFLOW, "subset count overflow");
    if (total == 0) return SD_OK;
    // sampled, targets of the set, counts that add over the timepoints (point clouds, relax=True): the shared factorisation
    if (samples > 0 && !sel.members && !sel.Q && (T == 0 || relax) && d >= 1 && d <= 8 && n >= d + 3 && xswitch("SD_SIMPLEX_GENERIC") != 1 &&
        xswitch("SD_SIMPLEX_PERTARGET") != 1) {
        // with a workspace that holds at least a few hundred records: factor into HBM, replay from there (else the per-target kernels)
        if (ws && ws_bytes >= simplex_sampled_workspace_bytes(n, T, d, 256 < samples ? 256 : samples)) {
            switch (d) {
#define SX_WS(D_) case D_: return launch_simplex_ws<D_>(P, n, T, targets, m, tol, samples, seed, out, ws, ws_bytes, s);
                SX_WS(1) SX_WS(2) SX_WS(3) SX_WS(4) SX_WS(5) SX_WS(6) SX_WS(7) SX_WS(8)
#undef SX_WS
            }
        }
    }
    // aim for ~2^18 threads over all targets, at least 1 subset per thread
    u64 want_threads = ((u64)1 << 18) / (u64)(m > 0 ? m : 1);
    if (want_threads < SX_THREADS) want_threads = SX_THREADS;
    u64 per_thread = (total + want_threads - 1) / want_threads;
    if (per_thread < 1) per_thread = 1;
    u64 threads = (total + per_thread - 1) / per_thread;
    u64 blocks = (threads + SX_THREADS - 1) / SX_THREADS;
    if (blocks > 0x7fffffffull) return fail(SD_ERR_UNSUPPORTED, "too many subsets per target for one launch");
    for (i64 q0 = 0; q0 < m; q0 += 65535) {   // grid.y limit: fold large m into several launches
        i64 mm = m - q0 < 65535 ? m - q0 : 65535;
        dim3 grid((unsigned)blocks, (unsigned)mm);
#define SX_FAST(D_) case D_: hipLaunchKernelGGL((simplex_kernel_fast<D_>), grid, dim3(SX_THREADS), 0, s, P, n, T, targets, \
                                                 relax, tol, total, per_thread, samples, seed, q0, out, sel); break;
        if (xswitch("SD_SIMPLEX_GENERIC") == 1) {            // cross-check builds: the generic (scratch-memory) kernel
            hipLaunchKernelGGL(simplex_kernel, grid, dim3(SX_THREADS), 0, s, P, n, T, d, targets, relax, tol, total,
                               per_thread, samples, seed, q0, out, sel);
        } else {
            switch (d) {
                SX_FAST(1) SX_FAST(2) SX_FAST(3) SX_FAST(4) SX_FAST(5) SX_FAST(6) SX_FAST(7) SX_FAST(8)
                default:
                    hipLaunchKernelGGL(simplex_kernel, grid, dim3(SX_THREADS), 0, s, P, n, T, d, targets, relax, tol,
                                       total, per_thread, samples, seed, q0, out, sel);
            }
        }
#undef SX_FAST
    }
    SD_HIP(hipGetLastError());
    return SD_OK;
}

int launch_pointcloud_simplex(const double *P, i64 n, int d, const i64 *targets, i64 m, double tol,
                              i64 samples, u64 seed, u64 *out, hipStream_t s, void *ws, size_t ws_bytes) {
    return launch_simplex_common(P, n, 0, d, targets, m, 1, tol, samples, seed, out, s, SxSel{nullptr, 0, nullptr, 0}, ws, ws_bytes);
}

// external targets Q (m x d): out[q] = #{(d+1)-subsets of ALL n rows of P whose simplex contains Q[q]}
int launch_pointcloud_simplex_external(const double *P, i64 n, int d, const double *Q, i64 m, double tol, u64 *out,
                                       hipStream_t s) {
    return launch_simplex_common(P, n, 0, d, nullptr, m, 1, tol, -1, 0, out, s, SxSel{nullptr, 0, Q, d});
}

// explicit blocks: out[k] = #{(d+1)-subsets of block k's others whose simplex contains the block's target (its last row)}
int launch_pointcloud_simplex_subsets(const double *P, i64 n, int d, const int *members, i64 nb, int bs, double tol,
                                      u64 *out, hipStream_t s) {
    return launch_simplex_common(P, n, 0, d, nullptr, nb, 1, tol, -1, 0, out, s, SxSel{members, bs, nullptr, d});
}

int launch_multi_simplex(const double *P, i64 n, i64 T, int d, const i64 *targets, i64 m, int relax,
                         double tol, i64 samples, u64 seed, u64 *out, hipStream_t s, void *ws, size_t ws_bytes) {
    return launch_simplex_common(P, n, T, d, targets, m, relax, tol, samples, seed, out, s, SxSel{nullptr, 0, nullptr, 0}, ws, ws_bytes);
}

}  // namespace sd
