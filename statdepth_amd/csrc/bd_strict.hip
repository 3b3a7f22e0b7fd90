// bd_strict.hip -- K3: strict band depth (relax=False, the reference's default).
//
// Replaces the subset loop of _univariate_band_depth (_functional.py:246-251) with
// `containment // len(curve)` (_containment.py:80): a j-subset of the other curves
// counts only if its band contains the target at EVERY timepoint.
//
// Per target, every curve i gets two T-bit masks over the timepoints:
//   UN_i[t] = (x_i > x_q) or x_i is NaN,   DN_i[t] = (x_i < x_q) or x_i is NaN.
// With pandas' skipna min/max (_containment.py:68-69) a subset fails at t iff all
// its members are in UN or all are in DN, so it is contained at every t iff
//   AND_members(UN) == 0 and AND_members(DN) == 0      (as T-bit masks).
// A NaN in the target fails everything (count 0).
//
// What runs, by case (launch_bd_strict_impl):
//   J = 2, 6 <= T <= 8, any n, NaN-free   strict_class_wg_kernel: the same classes counted by a workgroup per few targets
//   J = 2, T <= 5, any n      strict_class_kernel: the masks ARE classes (4^T or 3^T of them); per target one pass over
//                             the curves and a class transform.  The L-infinity depth of point clouds.
//   J = 2, n <= 131 071       per batch of targets: masks (strict_masks_rank_kernel from the bucket kernel's rank image
//                             for n <= 32 767, else strict_masks2_kernel from the values) -> digests of the canonical
//                             masks (strict_hash_kernel) -> pairs of CLEAN curves counted by grouping complementary
//                             masks (strict_match_lds_kernel; global-memory table behind it) -> pairs with a DIRTY
//                             curve (tie / NaN) tested by strict_pairs2_kernel, for the targets that have any.
//                             T > 1024: those targets through the first generation instead.
//   J = 2, n > 131 071        pair kernel for every pair (refused when that would take hours).
//   J = 3, 4                  first-generation masks + prefix enumeration (small n, like the reference).
//   external targets, explicit blocks: launch_bd_strict_external, launch_bd_strict_subsets.
// Integer / compare work throughout: VALU-issue and LDS bound, no MFMA.
#include <stdlib.h>

#include "sd_common.h"

namespace sd {

int launch_rank_bucket_image(const double *Y, i64 n, i64 row0, i64 rows, u32 *AB, u32 *nnan, hipStream_t s);   // mbd_rank_bucket.hip
int launch_rank_medium_image(const double *Y, i64 n, i64 row0, i64 rows, u32 *AB, u32 *nnan, hipStream_t s);   // ... its two-block form
int launch_rank_big_image(const double *Y, i64 T, i64 n, u32 *img, u32 *nnan, void *ws, size_t ws_bytes, hipStream_t s);   // mbd_rank_big.hip
// bd_strict_grid.hip: two to four coordinates at large n through a grid of cells instead of every pair of points
bool bd_strict_grid_applies(i64 T, i64 n, int J);
size_t bd_strict_grid_workspace_bytes(i64 T, i64 n, bool subset);
int launch_bd_strict_grid(const double *Y, i64 T, i64 n, const i64 *targets, i64 m, u64 *out, int jcols, u32 *flag, void *ws,
                          size_t ws_bytes, hipStream_t s);
// ... for all targets, or a subset that is not small (else the state-class kernel's O(m n) is less work)
static inline bool strict_grid_wanted(i64 T, i64 n, i64 m, int J) {
    return bd_strict_grid_applies(T, n, J) && m * 32 >= n && xswitch("SD_STRICT_V1") != 1 && xswitch("SD_STRICT_NOCLASS") != 1 &&
           xswitch("SD_STRICT_NOGRID") != 1;
}
// 32 767 < n <= 131 071: 32-bit ranks from the large-n route's B image (strict_masks_rank32_kernel)
static inline bool strict_rank32_applies(i64 T, i64 n, int J) {
    return J == 2 && n > 32767 && n <= 131071 && mbd_rank_big_supported(T, n, 2);
}
constexpr i64 ST_RANK_MAXN = 32767;          // ranks below 2^15 (two per register in the mask kernel); images: bucket kernel up to
                                             // 16 384 curves, its column-block form beyond

constexpr int ST_THREADS = 256;
constexpr int ST_WREG = 16;          // mask words kept in registers (T <= 1024)

static inline i64 strict_words(i64 T) { return (T + 63) / 64; }

// complement matching: the global-memory table behind the LDS one (strict_match_insert_kernel): slot = 64-bit key + two counters
constexpr i64 ST_MATCH_MAXN = 131071;                    // the keys carry a 17-bit curve id
static inline i64 strict_table_slots(i64 n) {
    i64 s = 64;
    while (s < 2 * n) s <<= 1;
    return s;
}
// a series is "short" for the class kernel up to 5 timepoints: 81 / 243 counters per lane leave two to six waves per CU, and
// still O(n) per target beats masks + matching at every size (10^5 x 4: 21 against 755 ms; 10^4 x 5: 1.3 against 2.7 ms)
static inline bool strict_class_applies(i64 T, i64 n, int J) {
    (void)n;
    return J == 2 && T <= 5;
}
// 6 ... 8 timepoints: 729 ... 6 561 three-state classes per target are too many for a histogram per lane; a workgroup takes a
// few targets and counts into shared histograms (strict_class_wg_kernel).  NaN-free data only: four states would be 4^T
// counters (256 KB at T = 8), so data with NaN goes the way it went before (masks + matching up to 131 071 curves).
static inline bool strict_class_wg_applies(i64 T, int J) { return J == 2 && T >= 6 && T <= 8; }
// ... and beyond the matching's reach, with more pairs than the pair kernel is allowed, they are the ONLY route
static inline bool strict_class_wg_only(i64 T, i64 n, i64 m, int J) {
    return strict_class_wg_applies(T, J) && n > 131071 && (double)m * (double)n * (double)n * 0.5 > 2.0e14;
}
static inline bool strict_match_applies(i64 T, i64 n, int J) { return J == 2 && (T + 31) / 32 <= 65535 && n <= ST_MATCH_MAXN; }

static i64 strict_batch(i64 T, i64 n, i64 m) {
    size_t per = (size_t)n * 2 * strict_words(T) * 8 + (size_t)strict_table_slots(n) * 16 + 20 + (size_t)((n + 63) / 64) * 8 + (size_t)((T + 31) / 32) * 256 + (size_t)n * 13 + 16;
    i64 b = (i64)(((size_t)2048 << 20) / (per ? per : 1));   // up to 2 GiB of masks and tables per batch ...
    if (b < 1024) {                                           // ... or 16 GiB when that is what 1024 targets take: the matching
        const i64 b16 = (i64)(((size_t)16384 << 20) / (per ? per : 1));   // kernel runs one workgroup per target
        b = b16 < 1024 ? b16 : 1024;
    }
    if (b < 1) b = 1;
    if (b > m) b = m;
    if (b > 65535) b = 65535;
    return b;
}

// bytes of the mask pipeline's workspace for batches of b targets (the layout launch_bd_strict_impl carves)
static size_t strict_ws_for_batch(i64 T, i64 n, i64 b) {
    return align_up((size_t)b * n * 2 * strict_words(T) * 8, 256) + align_up((size_t)b * 4, 256) +
           align_up((size_t)b * (strict_table_slots(n) * 16 + 16 + ((n + 63) / 64) * 8), 256) +
           align_up((size_t)((T + 31) / 32) * 4, 256) + align_up((size_t)b * ((T + 31) / 32) * 256, 256) +
           align_up((size_t)b * n * 8, 256) + align_up((size_t)b * n, 256) + align_up((size_t)(b + 1) * 4, 256) +
           align_up((size_t)b * (n + 4) * 4, 256) +
           (n <= ST_RANK_MAXN ? align_up((size_t)T * n * 4, 256) + 2 * align_up((size_t)T * 4, 256) : 0) +
           (strict_rank32_applies(T, n, 2) ? align_up((size_t)T * n * 4, 256) + 2 * align_up((size_t)T * 4, 256) +
                                                 align_up(mbd_rank_big_workspace_bytes(T, n, 2), 256) : 0) + 2560;
}

// The RECOMMENDED size (batches of strict_batch targets).  The launchers take any workspace that holds a batch of one
// target and size their batches to what they are given (strict_batch_for_ws), so a caller short of memory may pass less:
// bd_strict_min_workspace_bytes is the floor.
size_t bd_strict_workspace_bytes(i64 T, i64 n, i64 m, int J) {
    // short series go through the class kernel (launch_bd_strict_classes): a flag, no images -- three coordinates at large n
    // through the grid of cells (rank image, records in three orders, cell histogram, class counts)
    if (strict_grid_wanted(T, n, m, J)) return bd_strict_grid_workspace_bytes(T, n, m < n) + 4096;
    if (strict_class_applies(T, n, J) && xswitch("SD_STRICT_V1") != 1 && xswitch("SD_STRICT_NOCLASS") != 1) return 4096;
    // 6 ... 8 timepoints: the class kernel's flag in front of what the mask pipeline takes should the data hold NaN (beyond
    // the matching's reach there is no such fallback: the flag only)
    if (strict_class_wg_only(T, n, m, J)) return 4096;
    return strict_ws_for_batch(T, n, strict_batch(T, n, m)) + (strict_class_wg_applies(T, J) ? 256 : 0);
}
// what a caller who KNOWS the data NaN-free needs: the flag alone where the state classes of 6 ... 8 timepoints take such data
size_t bd_strict_nanfree_workspace_bytes(i64 T, i64 n, i64 m, int J) {
    if (strict_class_wg_applies(T, J) && xswitch("SD_STRICT_V1") != 1 && xswitch("SD_STRICT_NOCLASS") != 1) return 4096;
    return bd_strict_workspace_bytes(T, n, m, J);
}
size_t bd_strict_min_workspace_bytes(i64 T, i64 n, i64 m, int J) {
    if (strict_class_applies(T, n, J) && xswitch("SD_STRICT_V1") != 1 && xswitch("SD_STRICT_NOCLASS") != 1) return 4096;
    if (strict_class_wg_only(T, n, m, J)) return 4096;
    return strict_ws_for_batch(T, n, 1) + (strict_class_wg_applies(T, J) ? 256 : 0);
}
// largest batch (<= the recommended one) whose layout fits ws_bytes; 0: not even one target fits
static i64 strict_batch_for_ws(i64 T, i64 n, i64 m, size_t ws_bytes) {
    i64 hi = strict_batch(T, n, m);
    if (strict_ws_for_batch(T, n, hi) <= ws_bytes) return hi;
    if (strict_ws_for_batch(T, n, 1) > ws_bytes) return 0;
    i64 lo = 1;                                               // fits
    while (hi - lo > 1) {
        const i64 mid = lo + (hi - lo) / 2;
        if (strict_ws_for_batch(T, n, mid) <= ws_bytes) lo = mid; else hi = mid;
    }
    return lo;
}

// masks[b][i][0..W) = UN, masks[b][i][W..2W) = DN
__global__ __launch_bounds__(ST_THREADS) void strict_masks_kernel(
    const double *__restrict__ Y, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0,
    u64 *__restrict__ masks, u32 *__restrict__ xnan, const u32 *__restrict__ gate, const double *__restrict__ Yt) {
    i64 i = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    i64 b = blockIdx.y;
    if (gate && gate[b * 4] == 0) return;      // target already counted by complement matching
    i64 q = q0 + b;
    i64 tg = targets ? targets[q] : q;
    const double *tq = Yt ? Yt + b * (((T + 31) / 32) * 32) : nullptr;   // external targets: the gathered copy
    i64 W = (T + 63) / 64;
    u64 *mrow = masks + ((size_t)b * n + (i < n ? i : 0)) * 2 * W;
    bool anynan = false;
    for (i64 w = 0; w < W; ++w) {
        u64 un = 0, dn = 0;
        i64 tend = (w + 1) * 64 < T ? (w + 1) * 64 : T;
        for (i64 t = w * 64; t < tend; ++t) {
            double xq = tq ? tq[t] : Y[t * n + tg];
            double xi = i < n ? Y[t * n + i] : 0.0;
            anynan |= (xq != xq);
            u64 bit = (u64)1 << (t & 63);
            bool isn = xi != xi;
            if (xi > xq || isn) un |= bit;
            if (xi < xq || isn) dn |= bit;
        }
        if (i < n) {
            mrow[w] = un;
            mrow[W + w] = dn;
        }
    }
    if (anynan && blockIdx.x == 0 && threadIdx.x == 0) xnan[b] = 1;
}

template <typename Tv>
__device__ __forceinline__ Tv block_sum(Tv v, Tv *scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    Tv r = 0;
    if (threadIdx.x == 0)
        for (unsigned k = 0; k < (blockDim.x + 63) / 64; ++k) r += scratch[k];
    return r;
}

// J = 2: pairs (a < b).  grid = (a tiles, b chunks, batch)
constexpr int ST_BCHUNK = 512;

template <bool REG>
__global__ __launch_bounds__(ST_THREADS) void strict_pairs_kernel(
    const u64 *__restrict__ masks, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0,
    const u32 *__restrict__ xnan, const u32 *__restrict__ gate, u64 *__restrict__ out, int jcols) {
    __shared__ u64 scratch[ST_THREADS / 64];
    i64 b = blockIdx.z;
    i64 q = q0 + b;
    if (xnan[b]) return;                       // NaN in the target: nothing is contained
    if (gate && gate[b * 4] == 0) return;      // target already counted by complement matching
    i64 tg = targets ? targets[q] : q;
    i64 a = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    i64 b0 = (i64)blockIdx.y * ST_BCHUNK;
    i64 b1 = b0 + ST_BCHUNK < n ? b0 + ST_BCHUNK : n;
    i64 amin = (i64)blockIdx.x * ST_THREADS;
    if (b1 - 1 <= amin) return;                // whole chunk at or below the tile: no a < b pair
    int W = (int)((T + 63) / 64);
    const u64 *mb = masks + (size_t)b * n * 2 * W;
    bool alive = a < n && a != tg;
    const u64 *ma = mb + (size_t)(alive ? a : 0) * 2 * W;
    u64 un[ST_WREG], dn[ST_WREG];
    if constexpr (REG) {
#pragma unroll
        for (int w = 0; w < ST_WREG; ++w) {
            un[w] = (w < W) ? ma[w] : 0;
            dn[w] = (w < W) ? ma[W + w] : 0;
        }
    }
    u64 good = 0;
    for (i64 c = b0; c < b1; ++c) {
        if (c == tg) continue;
        const u64 *mc = mb + (size_t)c * 2 * W;   // wave-uniform: scalar loads
        // lanes that cannot count any more (dead lane, a >= c, or already in conflict with c) are "settled";
        // once the whole wave is settled the remaining words are skipped.  Most pairs conflict in the first
        // 64 timepoints, so this cuts the mask work by up to W (16 at T = 1000).
        u64 bad = (alive && a < c) ? 0 : ~0ull;
        if constexpr (REG) {
#pragma unroll
            for (int w = 0; w < ST_WREG; ++w) {
                if (w < W) {
                    bad |= (un[w] & mc[w]) | (dn[w] & mc[W + w]);
                    if ((w & 1) == 0 && __ballot(bad == 0) == 0) break;
                }
            }
        } else {
            for (int w = 0; w < W; ++w) {
                bad |= (ma[w] & mc[w]) | (ma[W + w] & mc[W + w]);
                if (__ballot(bad == 0) == 0) break;
            }
        }
        good += (bad == 0);
    }
    u64 tot = block_sum(good, scratch);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[q * jcols], tot);
}

// ---------------------------------------------------------------------------------------------------
// J = 2, second generation (T <= 1024): 32-bit mask words in a word-major image m32[b][k][i] (k < W32: UN word k,
// k >= W32: DN word k - W32; the n curves of one word are contiguous, so every access below is coalesced).
//  strict_masks2_kernel  thread = curve, block = 256 curves x one word (32 timepoints) x a group of 32 targets: the
//      curve's 32 values stay in VGPRs across the targets (the first generation re-read the matrix once per target:
//      m n T loads from L2), the target's values are wave-uniform scalar loads.
//  strict_pairs2_kernel  the lane's own masks live in VGPRs (32 + 32 words); the partners come in sub-chunks of 64
//      whose masks are staged in LDS (16 KiB).  Pass 1 tests the first four words (128 timepoints) of every partner
//      branch-free (wave-uniform LDS addresses: broadcasts) and records the survivors as one bit per partner; pass 2
//      walks the remaining words only for the survivors (most pairs conflict early).  The first generation read the
//      partner's words through the scalar cache inside a loop with an early exit, which serialised on the
//      scalar-load latency (~1500 cycles per partner and wave).
// ---------------------------------------------------------------------------------------------------
constexpr int ST_W32 = 32;                  // mask words per kind kept in registers (T <= 1024)
constexpr int ST_TG = 32;                   // targets per block of the mask kernel
constexpr int ST_SUB = 64;

// How the mask kernels store a (target, curve, word) pair.  FULL: UN and DN words (the pair kernel's input).  With
// matching on, the first pass over a batch stores UN only and notes in dflag[b][i] whether the curve is dirty for the
// target (tie or NaN at a timepoint that counts: the only thing the matching needs DN for); DN words -- half of the
// largest traffic of the strict path -- are then written by a second pass over the targets that have dirty curves
// (dlist / *dcount, appended by the matching), which continuous data never runs.
struct StrictMaskOut {
    u32 *m32;
    unsigned char *dflag;      // null: FULL
    const u32 *cmask;          // with dflag: timepoints that do not count
    const u32 *tiemask;        // rank kernel: timepoints at which some two curves hold the same value (null: unknown)
    const u32 *dlist;          // null: every target of the batch, else the listed ones
    const u32 *dcount;
};
__device__ __forceinline__ void strict_store_masks(const StrictMaskOut &o, i64 b, int k, int W32, i64 n, i64 i, u32 un, u32 dn,
                                                   u32 valid) {
    if (i >= n) return;
    o.m32[((size_t)b * 2 * W32 + k) * n + i] = un;
    if (!o.dflag) {
        o.m32[((size_t)b * 2 * W32 + W32 + k) * n + i] = dn;
    } else {
        const u32 v = valid & ~o.cmask[k];
        if ((((un ^ dn) & v) != v) || (un & dn)) o.dflag[(size_t)b * n + i] = 1;      // rare: continuous data never stores
    }
}

// The batch's target curves, gathered curve-major and padded to whole words: Yt[b][t], t < Tp = 32 W32 (zero beyond T).
// The mask kernel then fetches a target's 32 values of a word with a few wide scalar loads instead of 32 strided ones;
// a NaN anywhere in the target is noted here.  grid = nb
__global__ __launch_bounds__(ST_THREADS) void strict_gather_targets_kernel(
    const double *__restrict__ Y, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0, const double *__restrict__ Q,
    i64 mq, double *__restrict__ Yt, u32 *__restrict__ xnan) {
    const i64 b = blockIdx.x;
    const i64 tg = Q ? q0 + b : (targets ? targets[q0 + b] : q0 + b);      // Q: external targets, T x mq time-major
    const double *src = Q ? Q : Y;
    const i64 ld = Q ? mq : n;
    const i64 Tp = ((T + 31) / 32) * 32;
    bool isn = false;
    for (i64 t = threadIdx.x; t < Tp; t += ST_THREADS) {
        const double v = t < T ? src[t * ld + tg] : 0.0;
        isn |= v != v;
        Yt[b * Tp + t] = v;
    }
    if (__syncthreads_or(isn) && threadIdx.x == 0) xnan[b] = 1;
}

// grid = (ceil(n / 256), W32, ceil(nb / ST_TG))
__global__ __launch_bounds__(ST_THREADS) void strict_masks2_kernel(
    const double *__restrict__ Y, const double *__restrict__ Yt, i64 T, i64 n, i64 nb, StrictMaskOut o) {
    const i64 cnt = o.dlist ? (i64)*o.dcount : nb;
    const i64 z0 = (i64)blockIdx.z * ST_TG;
    if (z0 >= cnt) return;
    const i64 zend = z0 + ST_TG < cnt ? z0 + ST_TG : cnt;
    const i64 i = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    const int k = blockIdx.y;
    const int W32 = (int)((T + 31) / 32);
    const i64 t0 = (i64)k * 32;
    const int tl = (int)(T - t0 < 32 ? T - t0 : 32);
    const u32 valid = tl == 32 ? 0xFFFFFFFFu : ((1u << tl) - 1u);
    double x[32];
    u32 nanbits = 0;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        x[t] = (i < n && t < tl) ? Y[(t0 + t) * n + i] : 0.0;
        nanbits |= (x[t] != x[t]) ? (1u << t) : 0u;
    }
    for (i64 z = z0; z < zend; ++z) {
        const i64 b = o.dlist ? (i64)o.dlist[z] : z;
        const double *__restrict__ xq = Yt + (b * W32 + k) * 32;    // wave-uniform, contiguous: wide scalar loads
        u32 un = nanbits, dn = nanbits;
#pragma unroll
        for (int t = 0; t < 32; ++t) {                              // no branch in here: the loads issue together
            const double q = xq[t];
            un |= (x[t] > q) ? (1u << t) : 0u;
            dn |= (x[t] < q) ? (1u << t) : 0u;
        }
        strict_store_masks(o, b, k, W32, n, i, un & valid, dn & valid, valid);
    }
}

// The same masks from INTEGER ranks (n <= 32 767).  The bucket kernel's image mode (mbd_rank_bucket.hip) turns the
// matrix into R[t][i] = B | A << 16 with B = number of curves strictly below curve i at t (0xFFFFFFFF: NaN): B is order-
// and tie-preserving (x_i > x_q <=> B_i > B_q, equal values share B), costs one pass of the headline kernel (0.07 ms at
// 10 000 x 1 000), and makes the mask kernel's compares 32-bit (full rate; the fp64 ones run at half) on half the
// registers.  Rt: the batch's targets gathered as in strict_gather_targets_kernel (B only, zero beyond T).
constexpr u32 ST_RANK_NAN = 0xFFFFFFFFu;
// Which timepoints have two curves with the same value at all?  (A + B + 1 < number of non-NaN curves for some key of
// the row.)  Where none has, no curve can tie with any target and the mask kernel skips the tie test.  grid = T
__global__ __launch_bounds__(ST_THREADS) void strict_row_ties_kernel(const u32 *__restrict__ R, const u32 *__restrict__ rnan,
                                                                    i64 n, u32 *__restrict__ tiemask) {
    const i64 t = blockIdx.x;
    const u32 nv = (u32)n - rnan[t];
    bool tie = false;
    for (i64 i = threadIdx.x; i < n; i += ST_THREADS) {
        const u32 w = R[t * n + i];
        tie |= w != ST_RANK_NAN && (w >> 16) + (w & 0xFFFFu) + 1u != nv;
    }
    if (__syncthreads_or(tie) && threadIdx.x == 0) atomicOr(&tiemask[t >> 5], 1u << (t & 31));
}

__global__ __launch_bounds__(ST_THREADS) void strict_gather_rank_targets_kernel(
    const u32 *__restrict__ R, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0, u32 *__restrict__ Rt,
    u32 *__restrict__ xnan) {
    const i64 b = blockIdx.x;
    const i64 tg = targets ? targets[q0 + b] : q0 + b;
    const i64 Tp = ((T + 31) / 32) * 32;
    bool isn = false;
    // packed for the mask kernel: word k's timepoints j and j + 16 share Rt[b][16 k + j] (low / high half)
    for (i64 e = threadIdx.x; e < Tp / 2; e += ST_THREADS) {
        const i64 t = (e / 16) * 32 + (e % 16);
        const u32 w0 = t < T ? R[t * n + tg] : 0u;
        const u32 w1 = t + 16 < T ? R[(t + 16) * n + tg] : 0u;
        isn |= w0 == ST_RANK_NAN || w1 == ST_RANK_NAN;
        Rt[b * (Tp / 2) + e] = (w0 & 0xFFFFu) | ((w1 & 0xFFFFu) << 16);
    }
    if (__syncthreads_or(isn) && threadIdx.x == 0) xnan[b] = 1;
}

// grid = (ceil(n / 256), W32, ceil(nb / ST_TG))
__global__ __launch_bounds__(ST_THREADS) void strict_masks_rank_kernel(
    const u32 *__restrict__ R, const u32 *__restrict__ Rt, i64 T, i64 n, i64 nb, StrictMaskOut o) {
    const i64 cnt = o.dlist ? (i64)*o.dcount : nb;
    const i64 z0 = (i64)blockIdx.z * ST_TG;
    if (z0 >= cnt) return;
    const i64 zend = z0 + ST_TG < cnt ? z0 + ST_TG : cnt;
    const i64 i = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    const int k = blockIdx.y;
    const int W32 = (int)((T + 31) / 32);
    const i64 t0 = (i64)k * 32;
    const int tl = (int)(T - t0 < 32 ? T - t0 : 32);
    const u32 valid = tl == 32 ? 0xFFFFFFFFu : ((1u << tl) - 1u);
    // two timepoints per register: x2 = rank at t0 + j | rank at t0 + j + 16 << 16 (ranks < 2^15: bit 15 of each half is
    // free), kept as x2 + H and x2 - H (H = 0x80008000) for the two subtractions below
    u32 xh[16], xm[16];
    u32 nanbits = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const u32 w0 = (i < n && j < tl) ? R[(t0 + j) * n + i] : 0u;
        const u32 w1 = (i < n && j + 16 < tl) ? R[(t0 + j + 16) * n + i] : 0u;
        const bool n0 = w0 == ST_RANK_NAN, n1 = w1 == ST_RANK_NAN;
        nanbits |= (n0 ? (1u << j) : 0u) | (n1 ? (1u << (j + 16)) : 0u);
        const u32 x2 = (n0 ? 0u : (w0 & 0xFFFFu)) | ((n1 ? 0u : (w1 & 0xFFFFu)) << 16);
        xh[j] = x2 + 0x80008000u;
        xm[j] = x2 - 0x80008000u;
    }
    // First pass of a matching batch over a full word in which every timepoint counts: only UN is stored, and of DN
    // only "is there a tie" is wanted -- a packed running minimum of x2 ^ q2 (a zero half = equal ranks = a tie)
    // instead of the second subtract / rotate / insert: 2.5 instead of 3 instructions per timepoint.
    // ... and where no two curves of the data set share a value at any timepoint of the word (tiemask: continuous data,
    // everywhere) not even that: 1.5 instructions per timepoint.
    const bool un_only = o.dflag != nullptr && tl == 32 && o.cmask[k] == 0;
    const bool no_ties = un_only && o.tiemask != nullptr && o.tiemask[k] == 0;
    // Both halves at once with plain 32-bit subtractions: each half of (q2 + H) - x2 = q2 - (x2 - H) is
    // q + 0x8000 - x > 0 (no borrow between the halves) and its bit 15 says x <= q; likewise (x2 + H) - q2 for
    // x >= q.  The accumulators rotate right by one per step and take the two bits at 15 and 31 (v_bfi): after 16
    // steps timepoint j sits in bit j and j + 16 in bit 16 + j.  1.5 full-rate instructions per (timepoint,
    // direction); the accumulated sense is inverted (NOT above / NOT below).
    // The next target's 16 rank pairs are fetched (one s_load_dwordx16) while this one's are used: the kernel was
    // waiting on that scalar load for half of its cycles (SQ_WAIT_INST_ANY 0.55 of the wave cycles).
    typedef u32 u32x16 __attribute__((ext_vector_type(16)));
    const u32 H = 0x80008000u;
    // Round 3, after the instruction costs were measured (tools/experiments/valu_rate.hip): v_alignbit_b32 and v_bfi_b32 issue in
    // 4.1 cycles each, v_and_b32 / v_lshrrev_b32 / v_or_b32 in 2.0 - 2.1.  The accumulator starts at 0 and each half is a 16-bit
    // shift register that is full after exactly 16 steps, so a plain 32-bit shift never carries a set bit from the upper half
    // into bit 15: shift, mask, or -- three two-cycle instructions (6.2 cycles) instead of rotate + insert (8.2).
#ifdef SD_MASK_ROTATE_BFI
#define ST_SHIFT_IN(acc, d) asm("v_alignbit_b32 %0, %0, %0, 1\n\tv_bfi_b32 %0, %2, %1, %0" : "+v"(acc) : "v"(d), "s"(H))
#else
#define ST_SHIFT_IN(acc, d) asm("v_lshrrev_b32 %0, 1, %0\n\tv_and_b32 %1, %2, %1\n\tv_or_b32 %0, %0, %1" : "+v"(acc), "+v"(d) : "s"(H))
#endif
    i64 b = o.dlist ? (i64)o.dlist[z0] : z0;
    u32x16 qv = *reinterpret_cast<const u32x16 *>(Rt + (b * W32 + k) * 16);
    for (i64 z = z0; z < zend; ++z) {
        const i64 bn = z + 1 < zend ? (o.dlist ? (i64)o.dlist[z + 1] : z + 1) : b;
        const u32x16 qn = *reinterpret_cast<const u32x16 *>(Rt + (bn * W32 + k) * 16);
        u32 na = 0, nb_ = 0;
        if (no_ties) {                                                  // block-uniform
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                u32 d1 = qv[j] - xm[j];                                 // (q + 0x8000) - x per half; bit 15 / 31: x <= q
                ST_SHIFT_IN(na, d1);
            }
            if (i < n) {
                o.m32[((size_t)b * 2 * W32 + k) * n + i] = ~na | nanbits;
                if (nanbits) o.dflag[(size_t)b * n + i] = 1;
            }
        } else if (un_only) {                                           // block-uniform
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            u16x2 mn = {0xFFFF, 0xFFFF};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const u32 q2 = qv[j];
                u32 d1 = q2 - xm[j];                                    // (q + 0x8000) - x per half; bit 15 / 31: x <= q
                ST_SHIFT_IN(na, d1);
                const u32 e = (xh[j] ^ H) ^ q2;                         // x2 ^ q2 (xh = x2 | H)
                mn = __builtin_elementwise_min(mn, __builtin_bit_cast(u16x2, e));
            }
            const u32 un = ~na | nanbits;
            if (i < n) {
                o.m32[((size_t)b * 2 * W32 + k) * n + i] = un;
                if (mn.x == 0 || mn.y == 0 || nanbits) o.dflag[(size_t)b * n + i] = 1;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const u32 q2 = qv[j];
                u32 d1 = q2 - xm[j];                                    // (q + 0x8000) - x per half; bit 15 / 31: x <= q
                u32 d2 = xh[j] - q2;                                    // (x + 0x8000) - q per half; bit 15 / 31: x >= q
                // rotate right by one, then take bits 15 and 31 from the difference (the compiler's own rendering of
                // this costs a third instruction)
                ST_SHIFT_IN(na, d1);
                ST_SHIFT_IN(nb_, d2);
            }
            strict_store_masks(o, b, k, W32, n, i, (~na | nanbits) & valid, (~nb_ | nanbits) & valid, valid);
        }
        qv = qn;
        b = bn;
    }
}

// ---------------------------------------------------------------------------------------------------
// The same masks from 32-BIT ranks (32 767 < n <= 131 071: config 3's size).  R[t][i] = the large-n route's B word
// (mbd_rank_big.hip, launch_rank_big_image): curves strictly below curve i at t, bit 31 set when the curve ties with
// another one there, 0xFFFFFFFF for NaN.  B is order- and tie-preserving, so x_i > x_q <=> B_i > B_q <=> bit 31 of
// B_q - B_i (ranks < 2^31): v_sub_u32 + v_alignbit_b32 per (timepoint, direction) against two half-rate v_cmp_f64 +
// two v_cndmask + v_or of the fp64 kernel.  Thread = curve, block = 256 curves x one word x 32 targets, as above.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ST_THREADS) void strict_row_ties32_kernel(const u32 *__restrict__ R, i64 n, u32 *__restrict__ tiemask) {
    const i64 t = blockIdx.x;
    bool tie = false;
    for (i64 i = threadIdx.x; i < n; i += ST_THREADS) {
        const u32 w = R[t * n + i];
        tie |= w != ST_RANK_NAN && (w >> 31) != 0;
    }
    if (__syncthreads_or(tie) && threadIdx.x == 0) atomicOr(&tiemask[t >> 5], 1u << (t & 31));
}

// Rt[b][t] = the target's rank at t (tie bit cleared; 0 beyond T and for NaN, which is noted in xnan)
__global__ __launch_bounds__(ST_THREADS) void strict_gather_rank32_targets_kernel(
    const u32 *__restrict__ R, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0, u32 *__restrict__ Rt,
    u32 *__restrict__ xnan) {
    const i64 b = blockIdx.x;
    const i64 tg = targets ? targets[q0 + b] : q0 + b;
    const i64 Tp = ((T + 31) / 32) * 32;
    bool isn = false;
    for (i64 t = threadIdx.x; t < Tp; t += ST_THREADS) {
        const u32 w = t < T ? R[t * n + tg] : 0u;
        isn |= w == ST_RANK_NAN;
        Rt[b * Tp + t] = w == ST_RANK_NAN ? 0u : (w & 0x7FFFFFFFu);
    }
    if (__syncthreads_or(isn) && threadIdx.x == 0) xnan[b] = 1;
}

// grid = (ceil(n / 256), W32, ceil(nb / ST_TG))
__global__ __launch_bounds__(ST_THREADS) void strict_masks_rank32_kernel(
    const u32 *__restrict__ R, const u32 *__restrict__ Rt, i64 T, i64 n, i64 nb, StrictMaskOut o) {
    const i64 cnt = o.dlist ? (i64)*o.dcount : nb;
    const i64 z0 = (i64)blockIdx.z * ST_TG;
    if (z0 >= cnt) return;
    const i64 zend = z0 + ST_TG < cnt ? z0 + ST_TG : cnt;
    const i64 i = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    const int k = blockIdx.y;
    const int W32 = (int)((T + 31) / 32);
    const i64 t0 = (i64)k * 32;
    const int tl = (int)(T - t0 < 32 ? T - t0 : 32);
    const u32 valid = tl == 32 ? 0xFFFFFFFFu : ((1u << tl) - 1u);
    u32 x[32];
    u32 nanbits = 0;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const u32 w = (i < n && t < tl) ? R[(t0 + t) * n + i] : 0u;
        const bool isn = w == ST_RANK_NAN;
        nanbits |= isn ? (1u << t) : 0u;
        x[t] = isn ? 0u : (w & 0x7FFFFFFFu);
    }
    const bool un_only = o.dflag != nullptr && tl == 32 && o.cmask[k] == 0;
    const bool no_ties = un_only && o.tiemask != nullptr && o.tiemask[k] == 0;
    typedef u32 u32x16 __attribute__((ext_vector_type(16)));
    for (i64 z = z0; z < zend; ++z) {
        const i64 b = o.dlist ? (i64)o.dlist[z] : z;
        const u32 *qp = Rt + (b * W32 + k) * 32;                      // wave-uniform: two s_load_dwordx16
        const u32x16 qa = *reinterpret_cast<const u32x16 *>(qp), qb = *reinterpret_cast<const u32x16 *>(qp + 16);
        u32 un = 0, dn = 0;
        if (no_ties) {                                                // block-uniform: continuous data, everywhere
#pragma unroll
            for (int t = 31; t >= 0; --t) un = __builtin_amdgcn_alignbit(un, (t < 16 ? qa[t] : qb[t - 16]) - x[t], 31);
            if (i < n) {
                o.m32[((size_t)b * 2 * W32 + k) * n + i] = un | nanbits;
                if (nanbits) o.dflag[(size_t)b * n + i] = 1;
            }
        } else if (un_only) {                                         // block-uniform: of DN only "is there a tie" is wanted
            u32 mn = 0xFFFFFFFFu;
#pragma unroll
            for (int t = 31; t >= 0; --t) {
                const u32 q = t < 16 ? qa[t] : qb[t - 16];
                un = __builtin_amdgcn_alignbit(un, q - x[t], 31);
                const u32 e = q ^ x[t];
                mn = mn < e ? mn : e;
            }
            if (i < n) {
                o.m32[((size_t)b * 2 * W32 + k) * n + i] = un | nanbits;
                if (mn == 0u || nanbits) o.dflag[(size_t)b * n + i] = 1;
            }
        } else {
#pragma unroll
            for (int t = 31; t >= 0; --t) {
                const u32 q = t < 16 ? qa[t] : qb[t - 16];
                un = __builtin_amdgcn_alignbit(un, q - x[t], 31);     // x above q
                dn = __builtin_amdgcn_alignbit(dn, x[t] - q, 31);     // x below q
            }
            strict_store_masks(o, b, k, W32, n, i, (un | nanbits) & valid, (dn | nanbits) & valid, valid);
        }
    }
}

// What the complement matching below needs per (target, curve), read off the UN masks once:
//   HF[b][i] = payload (60 bits) << 4 | exact << 3 | side << 2 | canonical mask non-empty << 1 | clean
// payload: the canonical mask itself when it has one non-zero word (exact = 1: word index << 32 | word), else 60 hash bits
// (canonical form, side, clean: see the matching section; cmask marks the timepoints that do not count).
// grid = (ceil(n / 256), nb); the words of a curve are n apart (word-major image): every load is coalesced.
// (Folding this into the mask kernel was measured: carrying 8 or 16 targets' hashes in VGPRs across all words costs
// the occupancy and the 32-target reuse of the curve values that strict_masks2_kernel lives on -- 30 and 47 ms instead
// of 17.6 + 6 ms at 10 000 x 1 000.)
__device__ __forceinline__ u64 strict_mix(u64 h, u32 w) {
    h = (h ^ w) * 0xff51afd7ed558ccdull;
    return h ^ (h >> 29);
}

struct StrictFirst {            // the first timepoint that counts: word and bit (wave-uniform)
    int wf;
    u32 vf;
};
__device__ __forceinline__ StrictFirst strict_first_counting(const u32 *__restrict__ cmask, int W32, u32 lastvalid) {
    StrictFirst f{0, 0u};
    for (; f.wf < W32; ++f.wf) {
        f.vf = (f.wf == W32 - 1 ? lastvalid : 0xFFFFFFFFu) & ~cmask[f.wf];
        if (f.vf) break;
    }
    return f;
}

__global__ __launch_bounds__(ST_THREADS) void strict_hash_kernel(const u32 *__restrict__ m32, i64 T, i64 n,
                                                                const u32 *__restrict__ xnan, const u32 *__restrict__ cmask,
                                                                const unsigned char *__restrict__ dflag, u64 *__restrict__ HF) {
    const i64 b = blockIdx.y;
    if (xnan[b]) return;
    const i64 a = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    if (a >= n) return;
    const int W32 = (int)((T + 31) / 32);
    const u32 lastvalid = (T & 31) ? ((1u << (T & 31)) - 1u) : 0xFFFFFFFFu;
    const u32 *mb = m32 + (size_t)b * 2 * W32 * n;
    const StrictFirst fc = strict_first_counting(cmask, W32, lastvalid);
    // dirty (tie / NaN where it counts): noted by the mask kernel; every row constant: nothing is clean (pair kernel)
    const bool clean = fc.vf != 0 && dflag[(size_t)b * n + a] == 0;
    u32 side = 0, flip = 0;
    if (fc.vf) {
        side = (mb[(size_t)fc.wf * n + a] >> (__ffs((int)fc.vf) - 1)) & 1u;
        flip = side ? 0xFFFFFFFFu : 0u;
    }
    u64 h = 0x9E3779B97F4A7C15ull;
    u32 one = 0, onek = 0, nz = 0;                            // the only non-zero canonical word, its index, how many there are
    for (int w0 = 0; w0 < W32; w0 += 16) {
        u32 un[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {                        // sixteen loads in flight
            const int w = w0 + j < W32 ? w0 + j : W32 - 1;
            un[j] = mb[(size_t)w * n + a];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int w = w0 + j;
            if (w < W32) {
                const u32 v = (w == W32 - 1 ? lastvalid : 0xFFFFFFFFu) & ~cmask[w];
                const u32 cw = (un[j] ^ flip) & v;
                if (cw) { one = cw; onek = (u32)w; ++nz; }
                h = strict_mix(h, cw);
            }
        }
    }
    // a canonical mask with a single non-zero word is described EXACTLY by (index, word): equal payloads are equal
    // masks and need no look at the masks (most groups of real data: curves that cross the target within one stretch
    // of 32 timepoints and stay on one side otherwise)
    const bool exact = nz <= 1;
    u64 payload;
    if (exact) {
        payload = ((u64)onek << 32) | one;
    } else {
        h ^= h >> 32;
        h *= 0xc4ceb9fe1a85ec53ull;
        h ^= h >> 33;
        payload = h >> 4;
    }
    HF[(size_t)b * n + a] = (payload << 4) | ((exact ? 1u : 0u) << 3) | (side << 2) | ((nz ? 1u : 0u) << 1) | (clean ? 1u : 0u);
}

// one (a tile, b chunk) of target b; db = the target's dirty bitmap when matching has counted the clean-clean pairs
__device__ __forceinline__ void strict_pairs2_target(const u32 *__restrict__ m32, i64 T, i64 n, const i64 *__restrict__ targets,
                                                     i64 q0, const u64 *__restrict__ dbits, u64 *__restrict__ out, int jcols,
                                                     const i64 b) {
    __shared__ u64 scratch[ST_THREADS / 64];
    __shared__ u32 orparts[ST_THREADS / 64][2][ST_SUB];   // per wave: OR of its quarter of partner j's UN / DN words
    __shared__ __attribute__((aligned(16))) u32 cm[ST_SUB][2 * ST_W32 + 4];   // partner j: UN words 0..31, DN words 0..31 (zero
                                                                                // beyond W32); rows padded by 16 bytes: the survivors'
                                                                                // per-lane rows fall on different banks
    const i64 q = q0 + b;
    // with matching on, the pairs of two clean curves are counted there; here only pairs with a dirty member remain
    const u64 *db = dbits ? dbits + (size_t)b * ((n + 63) / 64) : nullptr;
    const i64 tg = targets ? targets[q] : q;
    const i64 a = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    const i64 b0 = (i64)blockIdx.y * ST_BCHUNK;
    const i64 b1 = b0 + ST_BCHUNK < n ? b0 + ST_BCHUNK : n;
    const i64 amin = (i64)blockIdx.x * ST_THREADS;
    if (b1 - 1 <= amin) return;                // whole chunk at or below the tile: no a < b pair
    const int W32 = (int)((T + 31) / 32);
    const u32 *mb = m32 + (size_t)b * 2 * W32 * n;
    const bool alive = a < n && a != tg;
    u32 un[ST_W32], dn[ST_W32];
#pragma unroll
    for (int w = 0; w < ST_W32; ++w) {
        un[w] = (alive && w < W32) ? mb[(size_t)w * n + a] : 0u;
        dn[w] = (alive && w < W32) ? mb[(size_t)(W32 + w) * n + a] : 0u;
    }
    // a curve that is never above the target (UN empty) cannot conflict with one that is never below it (DN empty):
    // such pairs -- most of the contained pairs of banded data -- need no walk over the masks
    u32 ora = 0, ord_ = 0;
#pragma unroll
    for (int w = 0; w < ST_W32; ++w) { ora |= un[w]; ord_ |= dn[w]; }
    const bool u0a = ora == 0, d0a = ord_ == 0;
    const bool dirty_a = !db || (alive && ((db[a >> 6] >> (a & 63)) & 1));
    const bool tile_dirty = __syncthreads_or(dirty_a) != 0;     // block-uniform
    u64 good = 0;
    for (i64 s0 = b0; s0 < b1; s0 += ST_SUB) {
        const int len = (int)(b1 - s0 < ST_SUB ? b1 - s0 : ST_SUB);
        const u64 dsub = db ? db[s0 >> 6] : ~0ull;              // dirty partners of this sub-chunk (s0 is a multiple of 64)
        if (!tile_dirty && dsub == 0) continue;                 // clean tile x clean partners: nothing to count here
        __syncthreads();                                        // the previous sub-chunk has been read
        for (int e = threadIdx.x; e < ST_SUB * 2 * ST_W32; e += ST_THREADS) {
            const int w2 = e / ST_SUB, j = e % ST_SUB;          // consecutive threads: consecutive partners of one word
            const int w = w2 % ST_W32;
            u32 v = 0;
            if (j < len && w < W32) v = mb[(size_t)(w2 < ST_W32 ? w : W32 + w) * n + s0 + j];
            cm[j][w2] = v;
        }
        __syncthreads();
        // which partners have an empty UN / DN mask: wave w ORs words 8w..8w+7 of partner j = lane, the four waves meet in LDS
        {
            const int j = threadIdx.x & 63, part = threadIdx.x >> 6;
            u32 pu = 0, pd = 0;
#pragma unroll
            for (int w = 0; w < 8; w += 4) {
                const uint4 cu = *reinterpret_cast<const uint4 *>(&cm[j][part * 8 + w]);
                const uint4 cd = *reinterpret_cast<const uint4 *>(&cm[j][ST_W32 + part * 8 + w]);
                pu |= cu.x | cu.y | cu.z | cu.w;
                pd |= cd.x | cd.y | cd.z | cd.w;
            }
            orparts[part][0][j] = pu;
            orparts[part][1][j] = pd;
        }
        __syncthreads();
        u64 U0, D0;                                             // bit j: partner j has no UN / no DN bit (wave-uniform)
        {
            const int j = threadIdx.x & 63;
            U0 = __ballot((orparts[0][0][j] | orparts[1][0][j] | orparts[2][0][j] | orparts[3][0][j]) == 0);
            D0 = __ballot((orparts[0][1][j] | orparts[1][1][j] | orparts[2][1][j] | orparts[3][1][j]) == 0);
        }
        // ---- pass 1: first four words of every partner; survivors as bits ----
        u64 surv = 0;
#pragma unroll 8
        for (int j = 0; j < ST_SUB; ++j) {
            const uint4 cu = *reinterpret_cast<const uint4 *>(&cm[j][0]);
            const uint4 cd = *reinterpret_cast<const uint4 *>(&cm[j][ST_W32]);
            const u32 bad = (un[0] & cu.x) | (un[1] & cu.y) | (un[2] & cu.z) | (un[3] & cu.w) | (dn[0] & cd.x) |
                            (dn[1] & cd.y) | (dn[2] & cd.z) | (dn[3] & cd.w);
            surv |= (u64)(bad == 0) << j;
        }
        // real partners only (a curve that ties with the target everywhere has empty masks and conflicts with nothing,
        // not even with the padding or the target's own slot): block-uniform mask; then c > a only
        u64 real = len == ST_SUB ? ~0ull : (((u64)1 << len) - 1);
        if (tg >= s0 && tg < s0 + len) real &= ~((u64)1 << (int)(tg - s0));
        surv = alive ? (surv & real) : 0;
        if (!dirty_a) surv &= dsub;
        const i64 jf = a + 1 - s0;
        surv = (jf <= 0) ? surv : (jf >= 64 ? 0 : (surv >> (int)jf) << (int)jf);
        if (W32 <= 4) {
            good += (u64)__popcll(surv);
        } else {
            // pairs that cannot conflict by class: counted without the walk
            const u64 sure = surv & ((u0a ? D0 : 0ull) | (d0a ? U0 : 0ull));
            good += (u64)__popcll(sure);
            surv &= ~sure;
            // ---- pass 2: the survivors' remaining words ----
            while (surv) {
                const int j = __ffsll((long long)surv) - 1;
                surv &= surv - 1;
                u32 bad = 0;
#pragma unroll
                for (int w = 4; w < ST_W32; w += 4) {                   // 16-byte LDS reads (per-lane partner j)
                    const uint4 cu = *reinterpret_cast<const uint4 *>(&cm[j][w]);
                    const uint4 cd = *reinterpret_cast<const uint4 *>(&cm[j][ST_W32 + w]);
                    bad |= (un[w] & cu.x) | (un[w + 1] & cu.y) | (un[w + 2] & cu.z) | (un[w + 3] & cu.w) | (dn[w] & cd.x) |
                           (dn[w + 1] & cd.y) | (dn[w + 2] & cd.z) | (dn[w + 3] & cd.w);
                }
                good += (bad == 0);
            }
        }
    }
    u64 tot = block_sum(good, scratch);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[q * jcols], tot);
}

// Without matching: grid = (a tiles, b chunks, nb), block z = target z.  With matching: grid.z is a fixed number of
// layers that share out the batch's targets WITH dirty curves (dlist, *dcount: appended by the matching kernels) --
// continuous data has none, and a grid over all targets would cost more in empty blocks than everything else here.
// The same count over the target's LIST of curves that can be in a pair the matching has not counted (strict_match_lds_kernel
// writes it): the dirty curves (ties / NaN with the target; bit 31 of their entry) and the clean curves that cross the target.
// A clean curve that stays below (above) the target throughout contains it with a dirty curve a exactly when a is never
// below (never above) it -- DN_a (UN_a) empty -- so those partners, the bulk of banded or quantised data, are counted as
// z0 (z1) per dirty curve without a look at their masks.  ilist[b]: {entries, z0, z1, -, entry...}; the a < c rule runs
// on list positions.
// Tiles of ST_P3_THREADS = 1 024 list positions, each walking the WHOLE list: a partner sub-chunk is staged once per 1 024 lanes
// and a lane's own masks are fetched once per target (staging was half of the 256-thread, 512-partner form's time).
constexpr int ST_P3_THREADS = 1024;
__device__ __forceinline__ void strict_pairs3_target(const u32 *__restrict__ m32, i64 T, i64 n, i64 q0,
                                                     const u32 *__restrict__ ilist, u64 *__restrict__ out, int jcols,
                                                     const i64 b, const int tile, const int chunk, const int chunks) {
    __shared__ u64 scratch[ST_P3_THREADS / 64];
    __shared__ u32 orparts[4][2][ST_SUB];   // per wave: OR of its quarter of partner j's UN / DN words
    __shared__ __attribute__((aligned(16))) u32 cm[ST_SUB][2 * ST_W32 + 4];   // partner j: UN words 0..31, DN words 0..31 (zero
                                                                                // beyond W32); rows padded by 16 bytes: the survivors'
                                                                                // per-lane rows fall on different banks
    __shared__ u32 sidx[ST_SUB];                            // the sub-chunk's curve indices (bit 31: dirty)
    const i64 q = q0 + b;
    const u32 *lst = ilist + (size_t)b * (n + 4);
    const i64 nd_l = (i64)lst[0], nc_l = (i64)lst[3];          // dirty entries first, then the clean ones (stored from the back)
    const i64 len_l = nd_l + nc_l;
    const u64 z0 = lst[1], z1 = lst[2];
    lst += 4;
    auto entry = [&](i64 pos) -> u32 { return pos < nd_l ? lst[pos] : lst[n - 1 - (pos - nd_l)]; };
    const i64 apos = (i64)tile * ST_P3_THREADS + threadIdx.x;
    const i64 amin = (i64)tile * ST_P3_THREADS;
    if (amin >= nd_l) return;                               // a tile without a dirty lane: every pair of its lanes with a dirty
                                                            // curve is counted by that curve's lane (the earlier position)
    // This block's share of the partners.  A tile with many dirty lanes walks the whole list itself (one staging per 1 024
    // lanes, its masks fetched once); a few dirty lanes against a long list (a handful of ties in continuous data) are
    // latency-bound on one block, so the list is cut into `chunks` pieces that run side by side.
    const i64 dirty_here = nd_l - amin < ST_P3_THREADS ? nd_l - amin : ST_P3_THREADS;
    const i64 nchunk = dirty_here >= 128 ? 1 : (i64)chunks;
    const i64 csz = ((len_l + nchunk - 1) / nchunk + ST_SUB - 1) / ST_SUB * ST_SUB;
    if ((i64)chunk >= nchunk) return;
    const i64 b0 = (i64)chunk * csz;
    const i64 b1 = b0 + csz < len_l ? b0 + csz : len_l;
    const bool closed = chunk == 0;                   // this block also adds its lanes' pairs with the z0 / z1 curves
    if ((b0 >= len_l || b1 - 1 <= amin) && !closed) return;   // no partner here, or the whole chunk at or below the tile
    const int W32 = (int)((T + 31) / 32);
    const u32 *mb = m32 + (size_t)b * 2 * W32 * n;
    const bool alive = apos < len_l;
    const u32 ent = alive ? entry(apos) : 0u;
    const i64 a = (i64)(ent & 0x7FFFFFFFu);
    u32 un[ST_W32], dn[ST_W32];
#pragma unroll
    for (int w = 0; w < ST_W32; ++w) {
        un[w] = (alive && w < W32) ? mb[(size_t)w * n + a] : 0u;
        dn[w] = (alive && w < W32) ? mb[(size_t)(W32 + w) * n + a] : 0u;
    }
    // a curve that is never above the target (UN empty) cannot conflict with one that is never below it (DN empty):
    // such pairs -- most of the contained pairs of banded data -- need no walk over the masks
    u32 ora = 0, ord_ = 0;
#pragma unroll
    for (int w = 0; w < ST_W32; ++w) { ora |= un[w]; ord_ |= dn[w]; }
    const bool u0a = ora == 0, d0a = ord_ == 0;
    const bool dirty_a = alive && (ent >> 31);
    const bool tile_dirty = __syncthreads_or(dirty_a) != 0;     // block-uniform
    u64 good = 0;
    if (closed && dirty_a) good += (d0a ? z0 : 0ull) + (u0a ? z1 : 0ull);
    for (i64 s0 = b0; s0 < b1; s0 += ST_SUB) {
        if (s0 + ST_SUB <= amin + 1) continue;                  // every partner of this sub-chunk at or below every lane's position
        const int len = (int)(b1 - s0 < ST_SUB ? b1 - s0 : ST_SUB);
        __syncthreads();                                        // the previous sub-chunk has been read
        if (threadIdx.x < ST_SUB) sidx[threadIdx.x] = threadIdx.x < len ? entry(s0 + threadIdx.x) : 0u;
        __syncthreads();
        const u64 dsub = __ballot((int)(threadIdx.x & 63) < len && (sidx[threadIdx.x & 63] >> 31));   // dirty partners of this sub-chunk
        if (!tile_dirty && dsub == 0) continue;                 // clean tile x clean partners: counted by the matching
        for (int e = threadIdx.x; e < ST_SUB * 2 * ST_W32; e += ST_P3_THREADS) {
            const int w2 = e / ST_SUB, j = e % ST_SUB;          // consecutive threads: consecutive partners of one word
            const int w = w2 % ST_W32;
            u32 v = 0;
            if (j < len && w < W32) v = mb[(size_t)(w2 < ST_W32 ? w : W32 + w) * n + (sidx[j] & 0x7FFFFFFFu)];
            cm[j][w2] = v;
        }
        __syncthreads();
        // which partners have an empty UN / DN mask: wave w ORs words 8w..8w+7 of partner j = lane, the four waves meet in LDS
        if (threadIdx.x < 256) {
            const int j = threadIdx.x & 63, part = threadIdx.x >> 6;
            u32 pu = 0, pd = 0;
#pragma unroll
            for (int w = 0; w < 8; w += 4) {
                const uint4 cu = *reinterpret_cast<const uint4 *>(&cm[j][part * 8 + w]);
                const uint4 cd = *reinterpret_cast<const uint4 *>(&cm[j][ST_W32 + part * 8 + w]);
                pu |= cu.x | cu.y | cu.z | cu.w;
                pd |= cd.x | cd.y | cd.z | cd.w;
            }
            orparts[part][0][j] = pu;
            orparts[part][1][j] = pd;
        }
        __syncthreads();
        u64 U0, D0;                                             // bit j: partner j has no UN / no DN bit (wave-uniform)
        {
            const int j = threadIdx.x & 63;
            U0 = __ballot((orparts[0][0][j] | orparts[1][0][j] | orparts[2][0][j] | orparts[3][0][j]) == 0);
            D0 = __ballot((orparts[0][1][j] | orparts[1][1][j] | orparts[2][1][j] | orparts[3][1][j]) == 0);
        }
        // ---- pass 1: first four words of every partner; survivors as bits ----
        u64 surv = 0;
#pragma unroll 8
        for (int j = 0; j < ST_SUB; ++j) {
            const uint4 cu = *reinterpret_cast<const uint4 *>(&cm[j][0]);
            const uint4 cd = *reinterpret_cast<const uint4 *>(&cm[j][ST_W32]);
            const u32 bad = (un[0] & cu.x) | (un[1] & cu.y) | (un[2] & cu.z) | (un[3] & cu.w) | (dn[0] & cd.x) |
                            (dn[1] & cd.y) | (dn[2] & cd.z) | (dn[3] & cd.w);
            surv |= (u64)(bad == 0) << j;
        }
        // real partners only (a curve that ties with the target everywhere has empty masks and conflicts with nothing,
        // not even with the padding or the target's own slot): block-uniform mask; then c > a only
        const u64 real = len == ST_SUB ? ~0ull : (((u64)1 << len) - 1);
        surv = alive ? (surv & real) : 0;
        if (!dirty_a) surv &= dsub;
        const i64 jf = apos + 1 - s0;
        surv = (jf <= 0) ? surv : (jf >= 64 ? 0 : (surv >> (int)jf) << (int)jf);
        if (W32 <= 4) {
            good += (u64)__popcll(surv);
        } else {
            // pairs that cannot conflict by class: counted without the walk
            const u64 sure = surv & ((u0a ? D0 : 0ull) | (d0a ? U0 : 0ull));
            good += (u64)__popcll(sure);
            surv &= ~sure;
            // ---- pass 2: the survivors' remaining words ----
            while (surv) {
                const int j = __ffsll((long long)surv) - 1;
                surv &= surv - 1;
                u32 bad = 0;
#pragma unroll
                for (int w = 4; w < ST_W32; w += 4) {                   // 16-byte LDS reads (per-lane partner j)
                    const uint4 cu = *reinterpret_cast<const uint4 *>(&cm[j][w]);
                    const uint4 cd = *reinterpret_cast<const uint4 *>(&cm[j][ST_W32 + w]);
                    bad |= (un[w] & cu.x) | (un[w + 1] & cu.y) | (un[w + 2] & cu.z) | (un[w + 3] & cu.w) | (dn[w] & cd.x) |
                           (dn[w + 1] & cd.y) | (dn[w + 2] & cd.z) | (dn[w + 3] & cd.w);
                }
                good += (bad == 0);
            }
        }
    }
    u64 tot = block_sum(good, scratch);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[q * jcols], tot);
}

// Without matching: grid = (a tiles, b chunks, nb), block z = target z.  With matching: grid.z is a fixed number of
// layers that share out the batch's targets WITH dirty curves (dlist, *dcount: appended by the matching kernels) --
// continuous data has none, and a grid over all targets would cost more in empty blocks than everything else here.
constexpr int ST_PAIR_LAYERS = 32;
__global__ __launch_bounds__(ST_THREADS) void strict_pairs2_kernel(
    const u32 *__restrict__ m32, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0,
    const u32 *__restrict__ xnan, const u32 *__restrict__ dlist, const u32 *__restrict__ dcount,
    const u64 *__restrict__ dbits, u64 *__restrict__ out, int jcols) {
    if (!dlist) {
        if (xnan[blockIdx.z]) return;              // NaN in the target: nothing is contained
        strict_pairs2_target(m32, T, n, targets, q0, nullptr, out, jcols, (i64)blockIdx.z);
        return;
    }
    const u32 cnt = *dcount;
    for (u32 zi = blockIdx.z; zi < cnt; zi += gridDim.z) {
        strict_pairs2_target(m32, T, n, targets, q0, dbits, out, jcols, (i64)dlist[zi]);
        __syncthreads();                           // the target's shared arrays are reused by the next one
    }
}

// A persistent 1-D grid over the items (dirty target of the matching's work list, tile of 1 024 list positions, chunk of
// partners); an item without work (a tile beyond the target's dirty entries, a chunk the tile does not use) costs its block
// a look at the list's header, not a workgroup launch: thousands of 1 024-thread workgroups that return at once cost more
// in the dispatcher than the real ones cost in the kernel (measured: 3.3 -> 5.2 ms at 2 000 x 1 000 for twice the grid).
__global__ __launch_bounds__(ST_P3_THREADS) void strict_pairs3_kernel(
    const u32 *__restrict__ m32, i64 T, i64 n, i64 q0, const u32 *__restrict__ dlist, const u32 *__restrict__ dcount,
    const u32 *__restrict__ ilist, u64 *__restrict__ out, int jcols, int tiles, int chunks) {
    const i64 cnt = (i64)*dcount, items = cnt * tiles * chunks;
    // item = (chunk, tile, target), targets fastest: the items with work (chunk 0 of the first tiles of every target) are
    // consecutive and spread evenly over the blocks
    for (i64 it = blockIdx.x; it < items; it += gridDim.x) {
        const i64 zi = it % cnt;
        const int rem = (int)(it / cnt);
        strict_pairs3_target(m32, T, n, q0, ilist, out, jcols, (i64)dlist[zi], rem % tiles, rem / tiles, chunks);
        __syncthreads();                           // the item's shared arrays are reused by the next one
    }
}


// ---------------------------------------------------------------------------------------------------
// J = 2 by complement matching: O(n T) per target instead of O(n^2 T).
// Call a curve CLEAN for a target when at every timepoint that counts it is strictly above or strictly below it (no
// tie, no NaN): DN = ~UN.  A pair of clean curves is contained at every timepoint iff at every timepoint exactly one
// of them is above, i.e. iff UN_b == ~UN_a as T-bit masks: an EQUALITY, so the pairs can be counted by grouping
// instead of testing.  Canonical form of a mask: itself (side 0) or its complement (side 1), whichever is 0 at the
// first timepoint that counts; complementary masks share their canonical form and sit on opposite sides, so
//     clean-clean contained pairs = sum over distinct canonical masks of  count(side 0) * count(side 1).
// The empty canonical mask (curves below / above the target throughout) is counted directly; the others are grouped
// in an open-addressing table per target keyed by a digest of the canonical mask (strict_hash_kernel).  A slot holds hash bits + the id of the first curve that claimed it (one CAS, no lock,
// nothing to wait for); a curve that meets a key with its hash compares its full canonical mask with that first
// curve's before joining, so hash collisions cost a probe and never a wrong count.
// Curves that are not clean (ties, NaN) are DIRTY: pairs with a dirty member are counted by strict_pairs2_kernel,
// which skips everything else (dirty bitmap per target); continuous data has none and the pair kernel exits at once.
// ---------------------------------------------------------------------------------------------------
// Timepoints at which all curves hold the same value (a common start, say) constrain no pair: cmask has their bits, and
// the matching treats them as absent (they would otherwise make every curve "tie" and send every target to the pair
// kernel).
__global__ __launch_bounds__(ST_THREADS) void strict_const_rows_kernel(const double *__restrict__ Y, i64 T, i64 n,
                                                                      u32 *__restrict__ cmask) {
    // grid = T blocks, one per timepoint (cmask zeroed before): a row of equal values sets its bit
    const i64 t = blockIdx.x;
    const double *row = Y + t * n;
    const double first = row[0];
    bool differs = false;
    for (i64 i0 = 0; i0 < n; i0 += (i64)ST_THREADS * 8) {       // eight strides between the block-wide checks
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const i64 i = i0 + (i64)u * ST_THREADS + threadIdx.x;
            if (i < n && !(row[i] == first)) differs = true;     // NaN differs from everything
        }
        if (__syncthreads_or(differs)) return;                   // block-uniform
    }
    if (threadIdx.x == 0) atomicOr(&cmask[t >> 5], 1u << (t & 31));
}

// do curves a and rep have the same canonical mask for this target?  Equal HF words (payload, exact flag; sides apart)
// are required; exact payloads settle it, hashed ones (masks spread over several words) are confirmed on the masks.
__device__ __forceinline__ bool strict_same_canonical(const u32 *__restrict__ mb, i64 n, int W32, u32 lastvalid,
                                                      const u32 *__restrict__ cmask, i64 a, u64 hfa, i64 rep, u64 hfr) {
    if ((hfa >> 3) != (hfr >> 3)) return false;
    if (hfa & 8) return true;
    const u32 flip = (hfa & 4) ? 0xFFFFFFFFu : 0u, rflip = (hfr & 4) ? 0xFFFFFFFFu : 0u;
    u32 diff = 0;
    for (int w0 = 0; w0 < W32; w0 += 8) {
        u32 ma[8], mr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                         // sixteen loads in flight
            const int w = w0 + j < W32 ? w0 + j : W32 - 1;
            ma[j] = mb[(size_t)w * n + a];
            mr[j] = mb[(size_t)w * n + rep];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = w0 + j < W32 ? w0 + j : W32 - 1;
            const u32 v = (w == W32 - 1 ? lastvalid : 0xFFFFFFFFu) & ~cmask[w];
            diff |= ((ma[j] ^ flip) ^ (mr[j] ^ rflip)) & v;
        }
    }
    return diff == 0;
}

// One block per target, the table in LDS (8 bytes per slot: key = 14 tag bits | side | 17-bit id
// of the first curve with that canonical mask; counter = members after the first, side 0 in the low half, side 1 in
// the high half).  A curve that meets its own canonical mask joins the group with one LDS atomic on the counter; the
// value it gets back is the group before it, so `members on the other side so far` summed over the joiners is exactly
// count(side 0) * count(side 1).  Also writes the target's dirty bitmap and count for the pair kernel, and its total.
constexpr i64 ST_MATCH_LDS_CAP = 13107;                       // candidates the LDS table takes: 16 384 slots (128 KiB), load 0.8
constexpr int ST_ML_THREADS = 1024;
constexpr int ST_ML_SEEN = 65536;                             // bits per side of the "digest seen" filter (2 x 8 KiB of LDS)
static inline i64 strict_lds_slots(i64 n) {
    i64 s = 64;
    while (s * 4 < n * 5 && s < 16384) s <<= 1;
    return s;
}
__device__ __forceinline__ u64 strict_spread(u64 hf) {        // slot and tag bits from the payload (exact ones are not hashed yet)
    u64 h = (hf >> 3) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 31);
}
__global__ __launch_bounds__(ST_ML_THREADS) void strict_match_lds_kernel(
    const u32 *__restrict__ m32, const u64 *__restrict__ HF, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0,
    const u32 *__restrict__ xnan, const u32 *__restrict__ cmask, u32 *__restrict__ meta, u64 *__restrict__ dbits,
    u32 *__restrict__ dlist, u32 *__restrict__ dcount, int slots, int whole_targets, int force_overflow,
    u64 *__restrict__ out, int jcols, u32 *__restrict__ ilist) {
    extern __shared__ u32 tabl[];                             // keys [slots] | counters [slots]
    __shared__ u64 red[ST_ML_THREADS / 64][3];
    __shared__ u32 lcount, ccount;                            // the pair kernel's curve list so far: dirty entries (from the front), clean ones (from the back)
    const i64 b = blockIdx.x;
    if (xnan[b]) return;
    const i64 tg = targets ? targets[q0 + b] : q0 + b;
    const int tid = threadIdx.x;
    const int W32 = (int)((T + 31) / 32);
    const u32 lastvalid = (T & 31) ? ((1u << (T & 31)) - 1u) : 0xFFFFFFFFu;
    const u32 *mb = m32 + (size_t)b * 2 * W32 * n;
    const u64 *hb = HF + (size_t)b * n;
    u32 *cntl = tabl + slots;
    u32 *seen = cntl + slots;                                 // [2][ST_ML_SEEN / 32]: digests present on side 0 / side 1
    for (int e = tid; e < 2 * slots + 2 * (ST_ML_SEEN / 32); e += ST_ML_THREADS) tabl[e] = 0;
    if (tid == 0) { lcount = 0; ccount = 0; }
    u32 *lst = ilist ? ilist + (size_t)b * (n + 4) + 4 : nullptr;
    __syncthreads();
    u64 acc = 0;
    u32 z0 = 0, z1 = 0, nd = 0;
    // More curves than the table takes at a load of 0.8 (n > 13 107: slots = 16 384): the 2 x 8 KiB filter would fill up
    // (at n = 10^5 half of the crossing curves passed it and the table was filled in 5 - 6 rounds, every round a pass over the
    // digests: 0.52 of config-3-size's 0.83 s).  Then the whole table area serves as the filter first -- 2^19 bits per
    // side, load < 0.13 -- the few curves that pass it are remembered as one bit per (thread, iteration) in registers, and
    // the table is built from them alone: three passes over the digests, no rounds.
    const bool must_count = (i64)n * 5 > (i64)slots * 4;
    const bool bigf = must_count && slots == 16384;                         // block-uniform
    u32 *fbits = bigf ? tabl : seen;                                        // [2][FB / 32]
    const u32 FB = bigf ? (1u << 19) : (u32)ST_ML_SEEN;
    // Pass 1: dirty curves, the two empty-mask counts, and one bit per (side, digest): a curve can only pair with a
    // curve whose canonical mask -- hence digest -- it shares on the OTHER side, and in continuous data hardly any
    // digest occurs on both sides.  One LDS atomic per curve, no probing.
    for (i64 a0 = 0; a0 < n; a0 += ST_ML_THREADS) {
        const i64 a = a0 + tid;
        const bool active = a < n && a != tg;
        const u64 hf = a < n ? hb[a] : 0;
        const bool clean = hf & 1;
        const u32 side = (u32)(hf >> 2) & 1u;
        const bool isdirty = active && !clean;
        const u64 dw = __ballot(isdirty);
        if ((tid & 63) == 0 && a0 + (tid & ~63) < n) dbits[(size_t)b * ((n + 63) / 64) + (a >> 6)] = dw;
        nd += isdirty;
        if (lst) {
            // the pair kernel's list: dirty curves (bit 31) from the front, clean curves that cross the target from the back
            // (n - 1 entries at most in all); wave-aggregated appends
            const bool wantc = active && clean && (hf & 2);
            const u64 wd = __ballot(isdirty), wc = __ballot(wantc);
            if (wd) {
                u32 base = 0;
                if ((tid & 63) == 0) base = atomicAdd(&lcount, (u32)__popcll(wd));
                base = (u32)__builtin_amdgcn_readfirstlane((int)base);
                if (isdirty) lst[base + (u32)__popcll(wd & ((1ull << (tid & 63)) - 1ull))] = (u32)a | 0x80000000u;
            }
            if (wc) {
                u32 base = 0;
                if ((tid & 63) == 0) base = atomicAdd(&ccount, (u32)__popcll(wc));
                base = (u32)__builtin_amdgcn_readfirstlane((int)base);
                if (wantc) lst[(u32)n - 1u - (base + (u32)__popcll(wc & ((1ull << (tid & 63)) - 1ull)))] = (u32)a;
            }
        }
        if (active && clean) {
            if (!(hf & 2)) {                       // empty canonical mask: below throughout (side 0) / above throughout
                z0 += side == 0;
                z1 += side == 1;
            } else {
                const u32 bit = (u32)(strict_spread(hf) >> 24) & (FB - 1);
                atomicOr(&fbits[side * (FB / 32) + (bit >> 5)], 1u << (bit & 31));
            }
        }
    }
    __syncthreads();
    // How many candidates?  (The table takes 0.6 of its slots per round; see below.)  Not counted when even all n curves
    // would fit at a load of 0.8.
    int cand = 0;
    u64 f01 = 0, f23 = 0;                                                   // bigf: this thread's candidates, bit = iteration (n < 2^17)
    {
        int it = 0;
        for (i64 a0 = 0; a0 < n && must_count; a0 += ST_ML_THREADS, ++it) {
            const i64 a = a0 + tid;
            const u64 hf = a < n ? hb[a] : 0;
            bool c = false;
            if (a < n && a != tg && (hf & 3) == 3) {
                const u32 side = (u32)(hf >> 2) & 1u;
                const u32 bit = (u32)(strict_spread(hf) >> 24) & (FB - 1);
                c = (fbits[(1u - side) * (FB / 32) + (bit >> 5)] >> (bit & 31)) & 1u;
            }
            if (c) { if (it < 64) f01 |= 1ull << it; else f23 |= 1ull << (it - 64); }
            cand += __syncthreads_count(c);
        }
    }
    if (bigf) {                                                             // the filter has served: its area becomes the table
        __syncthreads();
        for (int e = tid; e < 2 * slots; e += ST_ML_THREADS) tabl[e] = 0;
        __syncthreads();
    }
    // More candidates than the table takes (n in the tens of thousands: the filter fills up and lets half of the crossing
    // curves through): the digests are dealt into `rounds` classes by hash bits of their own and the table is filled
    // class by class (a group lies in one class); more than 16 rounds, or a class that still does not fit: flagged.
    int rounds = (int)(((i64)cand * 5 + (i64)slots * 3 - 1) / ((i64)slots * 3));       // load 0.6 per round
    if (rounds < 1) rounds = 1;
    bool overflow = force_overflow || rounds > 16;                               // block-uniform
    // Pass 2: only the curves whose digest was seen on the other side enter the table (digests again from L2).
    for (int rd = 0; rd < rounds && !overflow; ++rd) {
    if (rd > 0) {
        __syncthreads();
        for (int e = tid; e < 2 * slots; e += ST_ML_THREADS) tabl[e] = 0;
        __syncthreads();
    }
    bool stuck = false;
    int it = 0;
    for (i64 a0 = 0; a0 < n; a0 += ST_ML_THREADS, ++it) {
        const i64 a = a0 + tid;
        if (bigf && !(((it < 64 ? f01 >> it : f23 >> (it - 64))) & 1ull)) continue;   // not a candidate: its digest is not even read
        const u64 hf = a < n ? hb[a] : 0;
        if (a < n && a != tg && (hf & 3) == 3) {
            const u32 side = (u32)(hf >> 2) & 1u;
            const u64 h = strict_spread(hf);
            const u32 bit = (u32)(h >> 24) & (ST_ML_SEEN - 1);
            if ((int)((u32)(h >> 53) % (u32)rounds) == rd &&
                (bigf || ((seen[(1u - side) * (ST_ML_SEEN / 32) + (bit >> 5)] >> (bit & 31)) & 1u))) {
                const u32 tag = ((u32)(h >> 40) & 0x1FFFu) | 0x2000u;       // 14 bits, never zero
                const u32 mine = (tag << 18) | (side << 17) | (u32)a;       // a < 2^17
                int slot = (int)(h & (u64)(slots - 1));
                for (int probe = 0; probe < slots; ++probe) {                // load <= 0.8: always ends early
                    u32 cur = __hip_atomic_load(&tabl[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (cur == 0) {
                        cur = atomicCAS(&tabl[slot], 0u, mine);
                        if (cur == 0) break;
                    }
                    if ((cur >> 18) == tag) {
                        const i64 rep = (i64)(cur & 0x1FFFFu);
                        const u64 hr = hb[rep];
                        if (strict_same_canonical(mb, n, W32, lastvalid, cmask, a, hf, rep, hr)) {
                            const u32 rside = (cur >> 17) & 1u;
                            const u32 old = atomicAdd(&cntl[slot], side ? 0x10000u : 1u);
                            acc += side ? (old & 0xFFFFu) + (rside == 0) : (old >> 16) + (rside == 1);
                            // the two side counts share one word: a group with more than 65 535 later members on one
                            // side would carry into the other half -- such a target goes to the global-memory table
                            // (full 32-bit counters), which recounts ALL its groups
                            stuck |= (side ? old >> 16 : old & 0xFFFFu) == 0xFFFFu;
                            break;
                        }
                    }
                    slot = (slot + 1) & (slots - 1);
                    stuck |= probe == slots - 1;                             // table full: never with load 0.6, but exact
                }
            }
        }
    }
    if (__syncthreads_or(stuck)) overflow = true;
    }                                                                         // rounds
    if (overflow) acc = 0;                                                    // the global-memory table counts ALL groups
    u64 r0 = acc, r1 = ((u64)z0 << 32) | z1, r2 = nd;
    for (int o = 32; o > 0; o >>= 1) {
        r0 += __shfl_down(r0, o);
        r1 += __shfl_down(r1, o);
        r2 += __shfl_down(r2, o);
    }
    if ((tid & 63) == 0) { red[tid >> 6][0] = r0; red[tid >> 6][1] = r1; red[tid >> 6][2] = r2; }
    __syncthreads();
    if (tid == 0) {
        u64 t0 = 0, t1 = 0, t2 = 0;
        for (int k = 0; k < ST_ML_THREADS / 64; ++k) { t0 += red[k][0]; t1 += red[k][1]; t2 += red[k][2]; }
        meta[b * 4] = (u32)t2;
        meta[b * 4 + 3] = overflow ? 1u : 0u;                 // the global-table kernels add this target's groups
        if (ilist) {
            u32 *hd = ilist + (size_t)b * (n + 4);
            hd[0] = lcount;                                   // dirty entries lst[0 .. hd[0]); clean ones lst[n - hd[3] .. n)
            hd[3] = ccount;
            hd[1] = (u32)(t1 >> 32);                          // z0: clean curves below the target throughout
            hd[2] = (u32)t1;                                  // z1: ... above it throughout
        }
        if (t2) dlist[atomicAdd(dcount, 1u)] = (u32)b;        // the pair kernel's work list
        // with whole_targets the pair kernel that follows counts ALL pairs of a target that has dirty curves
        if (!(whole_targets && t2)) out[(q0 + b) * jcols] = t0 + (t1 >> 32) * (t1 & 0xFFFFFFFFull);
    }
}

// Targets the LDS kernel flagged (meta[b][3]: more candidates than its table takes): the same grouping in a table in
// global memory (64-bit keys: 47 tag bits | 17-bit id; two counters per slot; >= 2n slots).  The LDS kernel has done
// the rest (dirty curves, empty masks, work list, out = z0 * z1); these kernels add the groups' products.
// grid = (ceil(slots / 4096), nb): zero the flagged targets' tables
__global__ __launch_bounds__(ST_THREADS) void strict_match_clear_kernel(const u32 *__restrict__ meta, unsigned long long *__restrict__ keys,
                                                                       u32 *__restrict__ cnt, i64 slots) {
    const i64 b = blockIdx.y;
    if (!meta[b * 4 + 3]) return;
    const i64 i0 = (i64)blockIdx.x * 4096;
    for (i64 i = i0 + threadIdx.x; i < i0 + 4096 && i < slots; i += ST_THREADS) {
        keys[(size_t)b * slots + i] = 0;
        cnt[((size_t)b * slots + i) * 2] = 0;
        cnt[((size_t)b * slots + i) * 2 + 1] = 0;
    }
}

// grid = (ceil(n / 256), nb)
__global__ __launch_bounds__(ST_THREADS) void strict_match_insert_kernel(
    const u32 *__restrict__ m32, const u64 *__restrict__ HF, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0,
    const u32 *__restrict__ xnan, const u32 *__restrict__ cmask, const u32 *__restrict__ meta,
    unsigned long long *__restrict__ keys, u32 *__restrict__ cnt, i64 slots) {
    const i64 b = blockIdx.y;
    if (xnan[b] || !meta[b * 4 + 3]) return;
    const i64 tg = targets ? targets[q0 + b] : q0 + b;
    const i64 a = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    if (a >= n || a == tg) return;
    const int W32 = (int)((T + 31) / 32);
    const u32 lastvalid = (T & 31) ? ((1u << (T & 31)) - 1u) : 0xFFFFFFFFu;
    const u32 *mb = m32 + (size_t)b * 2 * W32 * n;
    const u64 hf = HF[(size_t)b * n + a];
    if ((hf & 3) != 3) return;                                // dirty, or the empty canonical mask: counted already
    const u32 side = (u32)(hf >> 2) & 1u;
    const u64 h = strict_spread(hf);
    const u64 tag = (h >> 17) | ((u64)1 << 46);               // 47 bits, never zero
    const unsigned long long mine = (tag << 17) | (u64)a;
    unsigned long long *kb = keys + (size_t)b * slots;
    u32 *cb = cnt + (size_t)b * slots * 2;
    i64 slot = (i64)(h & (u64)(slots - 1));
    for (i64 probe = 0; probe < slots; ++probe) {             // the table has >= 2n slots: always ends early
        unsigned long long cur = __hip_atomic_load(&kb[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            cur = atomicCAS(&kb[slot], 0ull, mine);
            if (cur == 0) break;
        }
        if ((cur >> 17) == tag) {
            const i64 rep = (i64)(cur & 0x1FFFF);
            if (strict_same_canonical(mb, n, W32, lastvalid, cmask, a, hf, rep, HF[(size_t)b * n + rep])) break;
        }
        slot = (slot + 1) & (slots - 1);
    }
    atomicAdd(&cb[slot * 2 + side], 1u);
}

// grid = (nb, ceil(slots / ST_TOTAL_CHUNK)): the flagged targets' sums of count(side 0) * count(side 1)
constexpr int ST_TOTAL_CHUNK = 4096;
__global__ __launch_bounds__(ST_THREADS) void strict_match_total_kernel(
    const u32 *__restrict__ cnt, i64 slots, i64 q0, const u32 *__restrict__ xnan, const u32 *__restrict__ meta,
    int whole_targets, u64 *__restrict__ out, int jcols) {
    __shared__ u64 scratch[ST_THREADS / 64];
    const i64 b = blockIdx.x;
    if (xnan[b] || !meta[b * 4 + 3]) return;
    if (whole_targets && meta[b * 4] != 0) return;   // the pair kernel that follows counts ALL pairs of such a target
    const uint2 *cb = reinterpret_cast<const uint2 *>(cnt) + (size_t)b * slots;
    const i64 i0 = (i64)blockIdx.y * ST_TOTAL_CHUNK;
    const i64 i1 = i0 + ST_TOTAL_CHUNK < slots ? i0 + ST_TOTAL_CHUNK : slots;
    u64 acc = 0;
    for (i64 i = i0 + threadIdx.x; i < i1; i += ST_THREADS) {
        const uint2 c = cb[i];
        acc += (u64)c.x * (u64)c.y;
    }
    const u64 tot = block_sum(acc, scratch);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[(q0 + b) * jcols], tot);
}

// J = 3 / 4: one thread per (J-1)-prefix, loop over the last member.
template <int J>
__global__ __launch_bounds__(ST_THREADS) void strict_subsets_kernel(
    const u64 *__restrict__ masks, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0,
    const u32 *__restrict__ xnan, u64 *__restrict__ out, int jcols) {
    __shared__ u64 scratch[ST_THREADS / 64];
    i64 b = blockIdx.z;
    i64 q = q0 + b;
    if (xnan[b]) return;
    i64 tg = targets ? targets[q] : q;
    int W = (int)((T + 63) / 64);
    const u64 *mb = masks + (size_t)b * n * 2 * W;
    // prefix (i0 < i1 [< i2]) from a flat index over n^(J-1)
    i64 flat = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    i64 idx[3];
    bool ok = true;
    i64 f = flat;
    for (int k = J - 2; k >= 0; --k) { idx[k] = f % n; f /= n; }
    if (f != 0) ok = false;
    for (int k = 0; k < J - 1; ++k) {
        if (idx[k] == tg) ok = false;
        if (k > 0 && idx[k] <= idx[k - 1]) ok = false;
    }
    u64 good = 0;
    if (ok) {
        for (i64 c = idx[J - 2] + 1; c < n; ++c) {
            if (c == tg) continue;
            u64 bad = 0;
            for (int w = 0; w < W; ++w) {
                u64 u = mb[(size_t)c * 2 * W + w], d = mb[(size_t)c * 2 * W + W + w];
                for (int k = 0; k < J - 1; ++k) {
                    u &= mb[(size_t)idx[k] * 2 * W + w];
                    d &= mb[(size_t)idx[k] * 2 * W + W + w];
                }
                bad |= u | d;
            }
            good += (bad == 0);
        }
    }
    u64 tot = block_sum(good, scratch);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[q * jcols + (J - 2)], tot);
}

// ---------------------------------------------------------------------------------------------------
// Strict band depth (J = 2) of one target inside an explicit subset of the curves, for nb (subset, target) pairs in one
// launch: the K-block sampled estimator with the reference's default relax=False (_samplefunctionaldepth,
// _functional.py:170-182 calls _univariate_band_depth on n*K small blocks).  One workgroup per pair: the block's masks
// against its target are built in LDS (u32[members][2 W32 + 1], the + 1 keeps rows on different banks), then every
// thread walks the pairs (a, b > a) of its members a with an early exit per four words.  Blocks are small (n / K
// curves), so the whole pair fits the LDS: members * (2 W32 + 1) * 4 + bs * 4 bytes; larger ones keep their masks in a
// slice of the workspace instead (strict_subset_big_kernel below; refused only when bs * 4 + 64 exceeds the LDS).
// ---------------------------------------------------------------------------------------------------
constexpr int ST_SUB_THREADS = 512;
constexpr int ST_SUB_GRID = 2048;                               // workgroups (and scratch slices) of the large-block form
static inline size_t strict_subset_lds(i64 T, int bs) { return (size_t)bs * (2 * ((T + 31) / 32) + 1) * 4 + (size_t)bs * 4 + 64; }
bool bd_strict_subsets_supported(i64 T, int bs) { return strict_subset_lds(T, bs) <= 160 * 1024 - 2048; }
// blocks whose masks do not fit the LDS keep them in a slice of the workspace (L2-resident: a slice is read bs times)
size_t bd_strict_subsets_workspace_bytes(i64 T, i64 nb, int bs) {
    if (bd_strict_subsets_supported(T, bs)) return 0;
    const i64 g = nb < ST_SUB_GRID ? nb : ST_SUB_GRID;
    return (size_t)g * bs * (2 * ((T + 31) / 32) + 1) * 4 + 256;
}

// grid-stride over the (subset, target) pairs; scratch == nullptr: masks in LDS
__global__ __launch_bounds__(ST_SUB_THREADS) void strict_subset_kernel(const double *__restrict__ Y, i64 T, i64 n,
                                                                      const int *__restrict__ members, i64 nb, int bs,
                                                                      const int *__restrict__ target, u32 *__restrict__ scratch,
                                                                      u64 *__restrict__ out) {
    extern __shared__ u32 sm[];
    __shared__ u64 red[ST_SUB_THREADS / 64];
    __shared__ int s_cnt;
    const int W32 = (int)((T + 31) / 32);
    const int RW = 2 * W32 + 1;
    int *ids = reinterpret_cast<int *>(sm);                     // [bs] the block's other members
    u32 *mk = scratch ? scratch + (size_t)blockIdx.x * bs * RW : sm + bs;     // [cnt][RW]
    const int tid = threadIdx.x;
    for (i64 k = blockIdx.x; k < nb; k += gridDim.x) {
        __syncthreads();                                        // the previous pair's ids / masks / sums are done with
        const int tg = target[k];
        if (tid == 0) {
            int c = 0;
            for (int e = 0; e < bs; ++e) {
                const int col = members[k * bs + e];
                if (col >= 0 && col != tg) ids[c++] = col;
            }
            s_cnt = c;
        }
        bool tnan = false;
        for (i64 t = tid; t < T; t += ST_SUB_THREADS) {
            const double q = Y[t * n + tg];
            tnan |= q != q;
        }
        const bool anynan = __syncthreads_or(tnan) != 0;        // also publishes ids / s_cnt
        if (anynan) {                                           // NaN in the target: nothing is contained
            if (tid == 0) out[k] = 0;
            continue;
        }
        const int cnt = s_cnt;
        for (int e = tid; e < cnt * W32; e += ST_SUB_THREADS) {
            const int c = e / W32, w = e % W32;
            const i64 col = ids[c];
            u32 un = 0, dn = 0;
            const i64 t0 = (i64)w * 32;
            const int tl = (int)(T - t0 < 32 ? T - t0 : 32);
            for (int t = 0; t < tl; ++t) {
                const double x = Y[(t0 + t) * n + col], q = Y[(t0 + t) * n + tg];
                const bool isn = x != x;
                un |= (x > q || isn) ? (1u << t) : 0u;
                dn |= (x < q || isn) ? (1u << t) : 0u;
            }
            mk[c * RW + w] = un;
            mk[c * RW + W32 + w] = dn;
        }
        __syncthreads();                                        // (global stores of this workgroup are visible to it after the barrier)
        u64 good = 0;
        for (int a = tid; a < cnt; a += ST_SUB_THREADS) {
            const u32 *ra = mk + (size_t)a * RW;
            for (int b = a + 1; b < cnt; ++b) {
                const u32 *rb = mk + (size_t)b * RW;
                u32 bad = 0;
                for (int w = 0; w < W32 && !bad; w += 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (w + j < W32) bad |= (ra[w + j] & rb[w + j]) | (ra[W32 + w + j] & rb[W32 + w + j]);
                }
                good += bad == 0;
            }
        }
        for (int o = 32; o > 0; o >>= 1) good += __shfl_down(good, o);
        if ((tid & 63) == 0) red[tid >> 6] = good;
        __syncthreads();
        if (tid == 0) {
            u64 tot = 0;
            for (int j = 0; j < ST_SUB_THREADS / 64; ++j) tot += red[j];
            out[k] = tot;
        }
    }
}

int launch_bd_strict_subsets(const double *Y, i64 T, i64 n, const int *members, i64 nb, int bs, const int *target, u64 *out,
                             void *ws, size_t ws_bytes, hipStream_t s) {
    const bool in_lds = bd_strict_subsets_supported(T, bs);
    if ((size_t)bs * 4 + 64 > 160 * 1024 - 2048)
        return fail(SD_ERR_UNSUPPORTED, "strict subset depth: blocks of %d curves (the member list alone exceeds the LDS)", bs);
    u32 *scratch = nullptr;
    if (!in_lds) {
        const size_t need = bd_strict_subsets_workspace_bytes(T, nb, bs);
        if (!ws || ws_bytes < need)
            return fail(SD_ERR_WORKSPACE, "strict subset depth: %zu bytes of workspace for blocks of %d curves x %lld timepoints "
                        "(sd_bd_strict_subset_workspace_bytes)", need, bs, (long long)T);
        scratch = (u32 *)(((size_t)ws + 255) / 256 * 256);
    }
    const size_t lds = in_lds ? strict_subset_lds(T, bs) : (size_t)bs * 4 + 64;
    SD_HIP(hipFuncSetAttribute((const void *)strict_subset_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const i64 g = in_lds ? (nb < 65535 * 16 ? nb : 65535 * 16) : (nb < ST_SUB_GRID ? nb : ST_SUB_GRID);
    hipLaunchKernelGGL(strict_subset_kernel, dim3((unsigned)g), dim3(ST_SUB_THREADS), lds, s, Y, T, n, members, nb, bs, target, scratch,
                       out);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// ---------------------------------------------------------------------------------------------------
// J = 2 over at most three timepoints, any n: the L-infinity / box containment of point clouds (SURVEY 8 P4:
// FunctionalDepth([points.T]) -- "curves" = points, "timepoints" = coordinates; config 5 is 10^6 points in R^3, where
// the reference's default relax=False asks for the pairs of points whose bounding box contains the target).
// Per (target, other point) a STATE per coordinate in two bits -- above, below, neither (tie), both (NaN) -- i.e. a
// class c < 4^T; a pair is contained at every coordinate iff c_a & c_b == 0.  With h[c] = points per class,
//     ordered contained pairs = sum over c, c' with c & c' == 0 of h[c] h[c'] = sum over masks (-1)^popc(mask) U[mask]^2,
// U = superset sums of h (inclusion-exclusion over the 2T bits), so a target costs one pass over the points and a
// 64-entry transform -- O(n) per target instead of O(n^2), exact, no limit on n.
// Lanes = targets (coordinates in VGPRs, a private 4^T-counter histogram per lane in LDS: [class][lane], no atomics
// between lanes); the points stream through the scalar cache, eight per load, the same for every lane of the block.
// ---------------------------------------------------------------------------------------------------
constexpr int ST_CL_THREADS = 128;
// lanes (= targets) per block: the private histograms of a block must fit the LDS -- 3^T (NaN-free) or 4^T counters per lane
__host__ __device__ constexpr int st_cl_threads(int TT, bool NANS) {
    return TT <= 3 ? ST_CL_THREADS : (TT == 4 ? (NANS ? 64 : 128) : (NANS ? 32 : 64));
}

// is there a NaN anywhere in the data?  (flag[0] = 1)  NaN-free data -- the rule -- needs three states per coordinate
// instead of four: 27 counters per lane instead of 64 at T = 3, and 2.4 x the waves per SIMD that hide this kernel's
// LDS and scalar-load latencies.
__global__ __launch_bounds__(ST_THREADS) void strict_any_nan_kernel(const double *__restrict__ A, i64 na, const double *__restrict__ B,
                                                                   i64 nbv, u32 *__restrict__ flag) {
    bool isn = false;
    for (i64 i = (i64)blockIdx.x * ST_THREADS + threadIdx.x; i < na + nbv; i += (i64)gridDim.x * ST_THREADS) {
        const double v = i < na ? A[i] : B[i - na];
        isn |= v != v;
    }
    if (__syncthreads_or(isn) && threadIdx.x == 0) flag[0] = 1u;
}

template <int TT, bool NANS>
__global__ __launch_bounds__(st_cl_threads(TT, NANS)) void strict_class_kernel(const double *__restrict__ Y, i64 n, const i64 *__restrict__ targets,
                                                                    const double *__restrict__ Q, i64 m, const u32 *__restrict__ nanflag,
                                                                    u64 *__restrict__ out, int jcols) {
    if ((nanflag[0] != 0) != NANS) return;                                   // the other instantiation serves this data
    constexpr int P3 = TT == 1 ? 3 : (TT == 2 ? 9 : (TT == 3 ? 27 : (TT == 4 ? 81 : 243)));
    constexpr int NC = NANS ? (1 << (2 * TT)) : P3;
    constexpr int CLT = st_cl_threads(TT, NANS);
    __shared__ u32 hist[NC][CLT];
    const int tid = threadIdx.x;
    const i64 q = (i64)blockIdx.x * CLT + tid;
    const bool active = q < m;
    const i64 tg = (active && !Q) ? (targets ? targets[q] : q) : -1;        // its own column is not one of the others
    double x[TT];
    bool tnan = false;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        x[t] = !active ? 0.0 : (Q ? Q[t * m + q] : Y[t * n + tg]);
        tnan |= x[t] != x[t];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) hist[c][tid] = 0;
    // Per (target, point): 2 T fp64 compares turned into the class code (two bits per coordinate, NaN coordinates of the
    // point -- a scalar: the point is the same for every lane -- set both; or base 3 without NaN), one increment of the
    // lane's own counter.  The target meets itself in the stream (class 0: every coordinate ties) and is taken off
    // afterwards.  (Measured and not kept: compare + add-with-carry chains, one instruction per bit instead of two, and
    // ds_add instead of read / add / write: 1.9 and 1.8 s against 1.46 s at 10^6 points.)
    auto visit = [&](const double (&p)[TT]) {
        u32 code = 0;
        if constexpr (NANS) {
            u32 nanbits = 0;
#pragma unroll
            for (int t = TT - 1; t >= 0; --t) {
                const unsigned long long pb = (unsigned long long)__double_as_longlong(p[t]);
                const u32 hi = (u32)(pb >> 32) & 0x7FFFFFFFu, lo = (u32)pb;     // 32-bit tests: scalar ALU
                nanbits = (nanbits << 2) | ((hi > 0x7FF00000u || (hi == 0x7FF00000u && lo != 0u)) ? 3u : 0u);
                code |= (p[t] > x[t] ? 1u : 0u) << (2 * t);                      // above
                code |= (p[t] < x[t] ? 2u : 0u) << (2 * t);                      // below
            }
            code |= nanbits;
        } else {
#pragma unroll
            for (int t = TT - 1; t >= 0; --t) code = code * 3u + (p[t] > x[t] ? 1u : 0u) + (p[t] < x[t] ? 2u : 0u);
        }
        hist[code][tid] += 1u;                                               // own counter: no atomic needed
    };
    i64 i = 0;
    for (; i + 8 <= n; i += 8) {
        double blkp[TT][8];
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) blkp[t][k] = Y[t * n + i + k];      // wave-uniform addresses: scalar loads
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double p[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) p[t] = blkp[t][k];
            visit(p);
        }
    }
    for (; i < n; ++i) {
        double p[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) p[t] = Y[t * n + i];
        visit(p);
    }
    if (!active) return;
    if (tg >= 0 && !tnan) hist[0][tid] -= 1u;                                // the target itself (a NaN target counts nothing)
    // class 0 = the points that tie with the target in every coordinate: the only ones compatible with themselves
    const u64 ties = hist[0][tid];
    long long total = 0;
    if constexpr (NANS) {
        // superset sums over the 2T bits, in place, then inclusion-exclusion
#pragma unroll
        for (int bit = 1; bit < NC; bit <<= 1)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (!(c & bit)) hist[c][tid] += hist[c | bit][tid];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const long long u = (long long)hist[c][tid];
            total += (__builtin_popcount((unsigned)c) & 1) ? -u * u : u * u;
        }
    } else if constexpr (TT >= 4) {
        // three states, 81 / 243 classes: too many for registers.  A pair is compatible iff in no coordinate both are above or
        // both below: prod_t (1 - [both above at t] - [both below at t]) = sum over subsets S of the coordinates of (-1)^|S| [equal
        // and strict on S].  In place, per coordinate, the tie slot becomes the sum of the three states (a wild card); entry c
        // then counts the points that match c's strict digits, and the ordered pairs are sum_c (-1)^(strict digits of c) entry(c)^2.
#pragma unroll 1
        for (int stride = 1; stride < NC; stride *= 3)
#pragma unroll 1
            for (int g = 0; g < NC / 3; ++g) {
                const int base = (g / stride) * stride * 3 + (g % stride);
                hist[base][tid] += hist[base + stride][tid] + hist[base + 2 * stride][tid];
            }
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {
            int strict_digits = 0;
            for (int d = c; d; d /= 3) strict_digits += (d % 3) != 0;
            const long long u = (long long)hist[c][tid];
            total += (strict_digits & 1) ? -u * u : u * u;
        }
    } else {
        // three states: z = (M x ... x M) h in registers (tie ~ all, above ~ {tie, below}, below ~ {tie, above}), then h . z
        u64 h[NC], z[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { h[c] = hist[c][tid]; z[c] = h[c]; }
#pragma unroll
        for (int stride = 1; stride < NC; stride *= 3)
#pragma unroll
            for (int g = 0; g < NC / 3; ++g) {
                const int base = (g / stride) * stride * 3 + (g % stride);
                const u64 s0 = z[base], s1 = z[base + stride], s2 = z[base + 2 * stride];
                z[base] = s0 + s1 + s2;
                z[base + stride] = s0 + s2;
                z[base + 2 * stride] = s0 + s1;
            }
#pragma unroll
        for (int c = 0; c < NC; ++c) total += (long long)(h[c] * z[c]);
    }
    out[q * jcols] = tnan ? 0ull : ((u64)total - ties) / 2;                  // NaN in the target: nothing is contained
}

template <int TT>
static int launch_class_wg(const double *Y, i64 n, const i64 *targets, const double *Q, i64 m, const u32 *nanflag, u64 *out, int jcols,
                           hipStream_t s);

int launch_bd_strict_classes(const double *Y, i64 T, i64 n, const i64 *targets, const double *Q, i64 m, u64 *out, int jcols,
                             void *ws, size_t ws_bytes, hipStream_t s) {
    Carver cv(ws, ws_bytes);
    u32 *flag = (u32 *)cv.take(256);
    if (!flag) return fail(SD_ERR_WORKSPACE, "strict-depth workspace too small");
    SD_HIP(hipMemsetAsync(flag, 0, 4, s));
#define ST_CL_LAUNCH(TT_)                                                                                                      \
    hipLaunchKernelGGL((strict_class_kernel<TT_, false>), dim3((unsigned)((m + st_cl_threads(TT_, false) - 1) / st_cl_threads(TT_, false))), \
                       dim3(st_cl_threads(TT_, false)), 0, s, Y, n, targets, Q, m, (const u32 *)flag, out, jcols);             \
    hipLaunchKernelGGL((strict_class_kernel<TT_, true>), dim3((unsigned)((m + st_cl_threads(TT_, true) - 1) / st_cl_threads(TT_, true))),   \
                       dim3(st_cl_threads(TT_, true)), 0, s, Y, n, targets, Q, m, (const u32 *)flag, out, jcols);
#define ST_CL_NAN(TT_)                                                                                                         \
    hipLaunchKernelGGL((strict_class_kernel<TT_, true>), dim3((unsigned)((m + st_cl_threads(TT_, true) - 1) / st_cl_threads(TT_, true))),   \
                       dim3(st_cl_threads(TT_, true)), 0, s, Y, n, targets, Q, m, (const u32 *)flag, out, jcols);
    // two to four coordinates at large n, external targets aside: the grid of cells (bd_strict_grid.hip) when the workspace holds it (a
    // caller that passed the floor keeps the O(m n) kernels); it sets the flag itself from the rank route's NaN counts
    if (!Q && strict_grid_wanted(T, n, m, 2) && ws_bytes >= bd_strict_grid_workspace_bytes(T, n, targets != nullptr)) {
        int rc = launch_bd_strict_grid(Y, T, n, targets, m, out, jcols, flag, ws, ws_bytes, s);
        if (rc) return rc;
        // data with NaN (the flag is set): the four-state lane kernel, which returns at once otherwise
        if (T == 2) { ST_CL_NAN(2) } else if (T == 3) { ST_CL_NAN(3) } else { ST_CL_NAN(4) }
        SD_HIP(hipGetLastError());
        return SD_OK;
    }
    hipLaunchKernelGGL(strict_any_nan_kernel, dim3(1024), dim3(ST_THREADS), 0, s, Y, T * n, Q ? Q : Y, Q ? T * m : (i64)0, flag);
    switch ((int)T) {
        case 1: ST_CL_LAUNCH(1) break;
        case 2: ST_CL_LAUNCH(2) break;
        case 3:
        // T = 3, 4, 5 without NaN: the workgroup form (lanes = points, shared histograms: 10^5 x 3 / 4 / 5 in 4.9 / 6.7 / 8.5 ms against
        // 10.7 / 21 / 46 with a histogram per lane; 10^6 x 3: 427 against 506 ms); with NaN the four-state lane kernel.  Cross-check
        // builds, SD_STRICT_LANECLASS = 1: the lane kernel for both.
        case 4:
        case 5:
            if (xswitch("SD_STRICT_LANECLASS") == 1) {
                if (T == 3) { ST_CL_LAUNCH(3) } else if (T == 4) { ST_CL_LAUNCH(4) } else { ST_CL_LAUNCH(5) }
                break;
            }
#define ST_CL_WG(TT_)                                                                                                          \
            {                                                                                                                  \
                int rc = launch_class_wg<TT_>(Y, n, targets, Q, m, flag, out, jcols, s);                                       \
                if (rc) return rc;                                                                                             \
                hipLaunchKernelGGL((strict_class_kernel<TT_, true>),                                                           \
                                   dim3((unsigned)((m + st_cl_threads(TT_, true) - 1) / st_cl_threads(TT_, true))),            \
                                   dim3(st_cl_threads(TT_, true)), 0, s, Y, n, targets, Q, m, (const u32 *)flag, out, jcols);   \
            }
            if (T == 3) ST_CL_WG(3) else if (T == 4) ST_CL_WG(4) else ST_CL_WG(5)
#undef ST_CL_WG
            break;
        default: return fail(SD_ERR_UNSUPPORTED, "the class kernel covers up to five timepoints");
    }
#undef ST_CL_LAUNCH
#undef ST_CL_NAN
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// ---------------------------------------------------------------------------------------------------
// J = 2 over 6 ... 8 timepoints, NaN-free data, any n: the same state classes, counted by a WORKGROUP per G targets.
// Lanes = points (their coordinates in VGPRs, SCW_PTS points per thread and trip, coalesced loads that serve all G targets);
// the target's coordinates are wave-uniform (SGPRs), a pair costs 2 T compares, the base-3 code and one LDS atomic on the
// target's histogram (3^T counters).  Then the in-place wild-card transform of strict_class_kernel, coordinate by
// coordinate with the workgroup's threads, and sum_c (-1)^(strict digits of c) entry(c)^2.
// ---------------------------------------------------------------------------------------------------
constexpr int SCW_PTS = 4;
template <int TT> struct ScwCfg {
    static constexpr int NC = TT == 3 ? 27 : (TT == 4 ? 81 : (TT == 5 ? 243 : (TT == 6 ? 729 : (TT == 7 ? 2187 : 6561))));
    static constexpr int G = TT <= 5 ? 16 : (TT <= 7 ? 8 : 4);          // (T >= 6:) 46 / 70 / 105 KB of histograms
    static constexpr int NT = TT == 8 ? 1024 : 512;                     // three / two / one workgroup per CU
    // few classes: the lanes of a wave meet on the same counter (most points are strictly above or below in every coordinate:
    // 2^T classes) and the LDS serialises them -- R copies of a target's histogram, a lane counts into copy lane % R
    // T = 6: 8 targets x 2 copies against 16 x 1: 10.1 against 15.6 ms on random walks (correlated coordinates: fewer classes
    // occur), the same on independent ones; T = 7 / 8 with copies (4 x 2 / 2 x 2 targets): 13.2 / 23.3 against 15.1 / 20.0 on walks,
    // 13.1 / 23.4 against 12.2 / 18.5 on independent coordinates -- not taken
    // (T = 3: 16 copies; 8 the same, 32 slower; 32 targets per workgroup slower)
    static constexpr int R = TT == 3 ? 16 : (TT == 4 ? 8 : (TT == 5 ? 4 : (TT == 6 ? 2 : 1)));
    static constexpr size_t LDS = (size_t)G * R * NC * 4;
};
__device__ __forceinline__ double scw_uniform(double v) {               // a wave-uniform double into SGPRs
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)b), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int TT>
__global__ __launch_bounds__(ScwCfg<TT>::NT) void strict_class_wg_kernel(const double *__restrict__ Y, i64 n,
                                                                         const i64 *__restrict__ targets,
                                                                         const double *__restrict__ Q, i64 m,
                                                                         const u32 *__restrict__ nanflag,
                                                                         u64 *__restrict__ out, int jcols) {
    if (nanflag && nanflag[0] != 0) return;                             // (T <= 5: the four-state kernel serves this data)
    using C = ScwCfg<TT>;
    constexpr int NC = C::NC, G = C::G, NT = C::NT, NW = NT / 64, R = C::R, GS = R * NC;
    extern __shared__ u32 scw_hist[];                                   // [G][R][NC]
    __shared__ double xs[G][8];
    __shared__ long long red[G][NW];
    __shared__ u32 s_ties[G];
    u32 *hist = scw_hist;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 q0 = (i64)blockIdx.x * G;
    const int gc = (int)(m - q0 < G ? m - q0 : G);                      // targets of this workgroup
    for (int c = tid; c < G * GS; c += NT) hist[c] = 0;
    if (tid < G * TT) {
        const int g = tid / TT, t = tid % TT;
        double v = 0.0;
        if (g < gc) {
            const i64 q = q0 + g;
            v = Q ? Q[t * m + q] : Y[t * n + (targets ? targets[q] : q)];
        }
        xs[g][t] = v;
    }
    __syncthreads();
    for (i64 i0 = tid; i0 < n; i0 += (i64)NT * SCW_PTS) {
        double p[SCW_PTS][TT];
        bool ok[SCW_PTS];
#pragma unroll
        for (int k = 0; k < SCW_PTS; ++k) {
            const i64 i = i0 + (i64)k * NT;
            ok[k] = i < n;
#pragma unroll
            for (int t = 0; t < TT; ++t) p[k][t] = ok[k] ? Y[t * n + i] : 0.0;
        }
#pragma unroll 1
        for (int g = 0; g < gc; ++g) {
            double x[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) x[t] = scw_uniform(xs[g][t]);
            u32 *hg = hist + g * GS + (lane & (R - 1)) * NC;
#pragma unroll
            for (int k = 0; k < SCW_PTS; ++k) {
                u32 code = 0;
#pragma unroll
                for (int t = TT - 1; t >= 0; --t) code = code * 3u + (p[k][t] > x[t] ? 1u : 0u) + (p[k][t] < x[t] ? 2u : 0u);
                if (ok[k]) atomicAdd(&hg[code], 1u);
            }
        }
    }
    __syncthreads();
    // the target itself met in the stream as class 0 (every coordinate ties): taken off; class 0 = the points that tie with
    // the target everywhere, the only ones compatible with themselves
    if constexpr (R > 1) {
        for (int w = tid; w < gc * NC; w += NT) {
            u32 *h = hist + (w / NC) * GS + (w % NC);
            u32 v = h[0];
#pragma unroll
            for (int r = 1; r < R; ++r) v += h[r * NC];
            h[0] = v;
        }
        __syncthreads();
    }
    if (tid < gc) {
        const bool self = !Q && (targets ? targets[q0 + tid] : q0 + tid) >= 0;
        if (self) hist[tid * GS] -= 1u;
        s_ties[tid] = hist[tid * GS];
    }
    __syncthreads();
    // per coordinate the tie slot becomes the sum of the three states (see strict_class_kernel)
#pragma unroll 1
    for (int stride = 1; stride < NC; stride *= 3) {
        for (int w = tid; w < gc * (NC / 3); w += NT) {
            const int g = w / (NC / 3), idx = w % (NC / 3);
            u32 *h = hist + g * GS + (idx / stride) * stride * 3 + (idx % stride);
            h[0] += h[stride] + h[2 * stride];
        }
        __syncthreads();
    }
    long long acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0;
    for (int c = tid; c < NC; c += NT) {
        int strict_digits = 0;
#pragma unroll
        for (int t = 0, d = c; t < TT; ++t, d /= 3) strict_digits += (d % 3) != 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const long long u = (long long)hist[g * GS + c];
            acc[g] += (strict_digits & 1) ? -u * u : u * u;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        long long v = acc[g];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0) red[g][wave] = v;
    }
    __syncthreads();
    if (tid < gc) {
        long long total = 0;
        for (int w = 0; w < NW; ++w) total += red[tid][w];
        out[(q0 + tid) * jcols] = ((u64)total - (u64)s_ties[tid]) / 2;
    }
}

template <int TT>
static int launch_class_wg(const double *Y, i64 n, const i64 *targets, const double *Q, i64 m, const u32 *nanflag, u64 *out, int jcols,
                           hipStream_t s) {
    using C = ScwCfg<TT>;
    auto k = strict_class_wg_kernel<TT>;
    SD_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
    hipLaunchKernelGGL(k, dim3((unsigned)((m + C::G - 1) / C::G)), dim3(C::NT), C::LDS, s, Y, n, targets, Q, m, nanflag, out, jcols);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// Q != nullptr: the m targets are EXTERNAL curves (T x m, time-major), every curve of Y is an "other" (J = 2 only).
static int launch_bd_strict_impl(const double *Y, i64 T, i64 n, const i64 *targets, const double *Q, i64 m, int J,
                                 u64 *out, void *ws, size_t ws_bytes, hipStream_t s);

int launch_bd_strict(const double *Y, i64 T, i64 n, const i64 *targets, i64 m, int J,
                     u64 *out, void *ws, size_t ws_bytes, hipStream_t s) {
    return launch_bd_strict_impl(Y, T, n, targets, nullptr, m, J, out, ws, ws_bytes, s);
}

size_t bd_strict_external_workspace_bytes(i64 T, i64 n, i64 m) { return bd_strict_workspace_bytes(T, n, m, 2) + align_up((size_t)m * 8, 256); }

int launch_bd_strict_external(const double *Y, i64 T, i64 n, const double *Q, i64 m, u64 *out, void *ws, size_t ws_bytes,
                              hipStream_t s) {
    // the kernels exclude "the target itself" from the others by its column index: -1 for every external target
    Carver cv(ws, ws_bytes);
    i64 *none = (i64 *)cv.take((size_t)m * 8);
    const size_t need = cv.rest();                            // everything behind the index block: the batches adapt to it
    void *sws = cv.take(need);
    if (!none || !sws) return fail(SD_ERR_WORKSPACE, "strict-depth workspace too small (sd_bd_strict_external_workspace_bytes)");
    SD_HIP(hipMemsetAsync(none, 0xFF, (size_t)m * 8, s));
    return launch_bd_strict_impl(Y, T, n, none, Q, m, 2, out, sws, need, s);
}

static int launch_bd_strict_impl(const double *Y, i64 T, i64 n, const i64 *targets, const double *Q, i64 m, int J,
                                 u64 *out, void *ws, size_t ws_bytes, hipStream_t s) {
    // cross-check builds, SD_STRICT_NOCLASS = 1: short series through the mask kernels like any other
    if (strict_class_applies(T, n, J) && xswitch("SD_STRICT_V1") != 1 && xswitch("SD_STRICT_NOCLASS") != 1)
        return launch_bd_strict_classes(Y, T, n, Q ? nullptr : targets, Q, m, out, 1, ws, ws_bytes, s);
    // 6 ... 8 timepoints: the workgroup form of the class kernel when no value is NaN.  That is a property of the data: the
    // flag comes back to the host (one 4-byte copy and a wait on the stream -- the only route that waits), data with NaN goes
    // on to the mask pipeline below
    if (strict_class_wg_applies(T, J) && xswitch("SD_STRICT_V1") != 1 && xswitch("SD_STRICT_NOCLASS") != 1 && m > 0 && n > 0) {
        if (!ws || ws_bytes < 256) return fail(SD_ERR_WORKSPACE, "strict-depth workspace too small");
        u32 *flag = (u32 *)ws;
        u32 hflag = 0;
        SD_HIP(hipMemsetAsync(flag, 0, 4, s));
        hipLaunchKernelGGL(strict_any_nan_kernel, dim3(1024), dim3(ST_THREADS), 0, s, Y, T * n, Q ? Q : Y, Q ? T * m : (i64)0, flag);
        SD_HIP(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, s));
        SD_HIP(hipStreamSynchronize(s));
        if (!hflag) {
            const i64 *tg = Q ? nullptr : targets;
            switch ((int)T) {
                case 6: return launch_class_wg<6>(Y, n, tg, Q, m, nullptr, out, 1, s);
                case 7: return launch_class_wg<7>(Y, n, tg, Q, m, nullptr, out, 1, s);
                default: return launch_class_wg<8>(Y, n, tg, Q, m, nullptr, out, 1, s);
            }
        }
        if (strict_class_wg_only(T, n, m, J))
            return fail(SD_ERR_UNSUPPORTED, "strict band depth of %lld curves over %lld timepoints with NaN in the data: the state "
                        "classes of 6 to 8 timepoints take NaN-free data, matching takes up to %lld curves", (long long)n,
                        (long long)T, (long long)ST_MATCH_MAXN);
        ws = (char *)ws + 256;
        ws_bytes -= 256;
    }
    i64 W = strict_words(T);
    const i64 B = strict_batch_for_ws(T, n, m, ws_bytes);     // the recommended batch, or what the caller's workspace holds
    if (B < 1) return fail(SD_ERR_WORKSPACE, "strict-depth workspace too small: %zu bytes, one target takes %zu "
                           "(sd_bd_strict_min_workspace_bytes)", ws_bytes, strict_ws_for_batch(T, n, 1));
    Carver cv(ws, ws_bytes);
    u64 *masks = (u64 *)cv.take((size_t)B * n * 2 * W * 8);
    u32 *xnan = (u32 *)cv.take((size_t)B * 4);
    const i64 slots = strict_table_slots(n);
    // cross-check builds, SD_STRICT_NOMATCH = 1: every target through the pair kernel
    const bool match = strict_match_applies(T, n, J) && xswitch("SD_STRICT_V1") != 1 && xswitch("SD_STRICT_NOMATCH") != 1;
    // Without matching every pair of curves is tested for every target: refuse what would keep the GPU for hours
    // (m n^2 / 2 pair tests at ~2e11 per second) instead of starting it
    if (!match && J == 2 && (double)m * (double)n * (double)n * 0.5 > 2.0e14)
        return fail(SD_ERR_UNSUPPORTED, "strict band depth of %lld curves over %lld timepoints: pairs are counted by matching for up to "
                    "%lld curves (any number for T <= 5, and for T <= 8 without NaN); beyond that every pair is tested, %.1e tests here", (long long)n,
                    (long long)T, (long long)ST_MATCH_MAXN, (double)m * (double)n * (double)n * 0.5);
    const i64 dwords = (n + 63) / 64;
    // keys | counters | per target {dirty, below, above, -} | dirty bitmaps
    unsigned char *tab = (unsigned char *)cv.take((size_t)B * (slots * 16 + 16 + dwords * 8));
    u32 *cmask = (u32 *)cv.take((size_t)((T + 31) / 32) * 4);
    double *Yt = (double *)cv.take((size_t)B * ((T + 31) / 32) * 256);
    u64 *HF = (u64 *)cv.take((size_t)B * n * 8);                                    // hash + flags per (target, curve)
    unsigned char *dflag = (unsigned char *)cv.take((size_t)B * n);                 // dirty (target, curve) pairs
    u32 *dlist = (u32 *)cv.take((size_t)(B + 1) * 4);                               // dirty targets of the batch | their number
    u32 *ilist = (u32 *)cv.take((size_t)B * (n + 4) * 4);                           // per target: the pair kernel's list of curves
    if (!masks || !xnan || !tab || !cmask || !Yt || !HF || !dflag || !dlist || !ilist)
        return fail(SD_ERR_WORKSPACE, "strict-depth workspace too small");
    u32 *dcount = dlist + B;
    // cross-check builds, SD_STRICT_FP64_MASKS = 1: masks from the fp64 values at any n
    const bool rankmasks = !Q && n >= 2 && n <= ST_RANK_MAXN && J == 2 && xswitch("SD_STRICT_V1") != 1 && xswitch("SD_STRICT_FP64_MASKS") != 1;
    const bool rank32 = !Q && strict_rank32_applies(T, n, J) && xswitch("SD_STRICT_V1") != 1 && xswitch("SD_STRICT_FP64_MASKS") != 1;
    u32 *R = nullptr, *rnan = nullptr, *tiemask = nullptr;
    void *bigws = nullptr;
    size_t bigws_bytes = 0;
    if (rank32) {
        R = (u32 *)cv.take((size_t)T * n * 4);
        rnan = (u32 *)cv.take((size_t)T * 4);
        tiemask = (u32 *)cv.take((size_t)((T + 31) / 32) * 4);
        bigws_bytes = mbd_rank_big_workspace_bytes(T, n, 2);
        bigws = cv.take(bigws_bytes);
        if (!R || !rnan || !tiemask || !bigws) return fail(SD_ERR_WORKSPACE, "strict-depth workspace too small");
    }
    if (rankmasks) {
        R = (u32 *)cv.take((size_t)T * n * 4);
        rnan = (u32 *)cv.take((size_t)T * 4);
        tiemask = (u32 *)cv.take((size_t)((T + 31) / 32) * 4);
        if (!R || !rnan || !tiemask) return fail(SD_ERR_WORKSPACE, "strict-depth workspace too small");
    }
    // cross-check builds, SD_STRICT_GLOBAL_TABLE = 1: every target's groups through the global-memory table (the route
    // of targets with more candidates than the LDS table takes)
    const bool force_global = xswitch("SD_STRICT_GLOBAL_TABLE") == 1;
    const bool global_possible = match && (force_global || n > ST_MATCH_LDS_CAP);
    u64 *dbits = (u64 *)(tab + (size_t)B * (slots * 16 + 16));
    unsigned long long *keys = (unsigned long long *)tab;
    u32 *cnt = (u32 *)(tab + (size_t)B * slots * 8);
    u32 *dirty = (u32 *)(tab + (size_t)B * slots * 16);
    int jcols = J - 1;
    SD_HIP(hipMemsetAsync(out, 0, sizeof(u64) * m * jcols, s));
    if (J >= 3) {
        double threads = 1.0;
        for (int k = 0; k < J - 1; ++k) threads *= (double)n;
        if (threads > 4.0e9) return fail(SD_ERR_UNSUPPORTED, "strict J=%d enumeration too large for n=%lld", J, (long long)n);
    }
    if (match) {
        SD_HIP(hipMemsetAsync(cmask, 0, (size_t)((T + 31) / 32) * 4, s));
        // (an external target does not share a value all of Y's curves share: every timepoint counts for it)
        if (!Q) hipLaunchKernelGGL(strict_const_rows_kernel, dim3((unsigned)T), dim3(ST_THREADS), 0, s, Y, T, n, cmask);
    }
    if (rankmasks) {
        // the image launcher takes at most 2048 rows per workgroup
        const i64 step = 2048 * 64;
        for (i64 r0 = 0; r0 < T; r0 += step) {
            const i64 rows = T - r0 < step ? T - r0 : step;
            int rc = n <= 16384 ? launch_rank_bucket_image(Y, n, r0, rows, R + r0 * n, rnan + r0, s)
                                : launch_rank_medium_image(Y, n, r0, rows, R + r0 * n, rnan + r0, s);
            if (rc) return rc;
        }
        SD_HIP(hipMemsetAsync(tiemask, 0, (size_t)((T + 31) / 32) * 4, s));
        hipLaunchKernelGGL(strict_row_ties_kernel, dim3((unsigned)T), dim3(ST_THREADS), 0, s, (const u32 *)R, (const u32 *)rnan, n,
                           tiemask);
    }
    if (rank32) {
        int rc = launch_rank_big_image(Y, T, n, R, rnan, bigws, bigws_bytes, s);
        if (rc) return rc;
        SD_HIP(hipMemsetAsync(tiemask, 0, (size_t)((T + 31) / 32) * 4, s));
        hipLaunchKernelGGL(strict_row_ties32_kernel, dim3((unsigned)T), dim3(ST_THREADS), 0, s, (const u32 *)R, n, tiemask);
    }
    for (i64 q0 = 0; q0 < m; q0 += B) {
        i64 nb = m - q0 < B ? m - q0 : B;
        SD_HIP(hipMemsetAsync(xnan, 0, (size_t)nb * 4, s));
        dim3 g1((unsigned)((n + ST_THREADS - 1) / ST_THREADS), (unsigned)nb);
        dim3 g2((unsigned)((n + ST_THREADS - 1) / ST_THREADS), (unsigned)((n + ST_BCHUNK - 1) / ST_BCHUNK), (unsigned)nb);
        const i64 W32 = (T + 31) / 32;
        // cross-check builds, SD_STRICT_V1 = 1: the first-generation kernels only (they serve J > 2 anyway)
        const bool gen2 = xswitch("SD_STRICT_V1") != 1 && J == 2 && W32 <= ST_W32;
        const u32 *gate = nullptr;
        if (gen2 || match) {
            // second generation: 32-bit words, word-major image (fits the same workspace: 2 W32 n u32 <= 2 W n u64)
            dim3 g1b((unsigned)((n + ST_THREADS - 1) / ST_THREADS), (unsigned)W32, (unsigned)((nb + ST_TG - 1) / ST_TG));
            // with matching, the first pass stores UN words and dirty flags only (see StrictMaskOut)
            StrictMaskOut mo{(u32 *)masks, match ? dflag : nullptr, cmask, tiemask, nullptr, nullptr};
            if (match) SD_HIP(hipMemsetAsync(dflag, 0, (size_t)nb * n, s));
            if (rankmasks) {
                hipLaunchKernelGGL(strict_gather_rank_targets_kernel, dim3((unsigned)nb), dim3(ST_THREADS), 0, s, (const u32 *)R, T, n,
                                   targets, q0, (u32 *)Yt, xnan);
                hipLaunchKernelGGL(strict_masks_rank_kernel, g1b, dim3(ST_THREADS), 0, s, (const u32 *)R, (const u32 *)Yt, T, n, nb, mo);
            } else if (rank32) {
                hipLaunchKernelGGL(strict_gather_rank32_targets_kernel, dim3((unsigned)nb), dim3(ST_THREADS), 0, s, (const u32 *)R, T, n,
                                   targets, q0, (u32 *)Yt, xnan);
                hipLaunchKernelGGL(strict_masks_rank32_kernel, g1b, dim3(ST_THREADS), 0, s, (const u32 *)R, (const u32 *)Yt, T, n, nb, mo);
            } else {
                hipLaunchKernelGGL(strict_gather_targets_kernel, dim3((unsigned)nb), dim3(ST_THREADS), 0, s, Y, T, n, targets, q0, Q, m,
                                   Yt, xnan);
                hipLaunchKernelGGL(strict_masks2_kernel, g1b, dim3(ST_THREADS), 0, s, Y, (const double *)Yt, T, n, nb, mo);
            }
            if (match) {
                hipLaunchKernelGGL(strict_hash_kernel, g1, dim3(ST_THREADS), 0, s, (const u32 *)masks, T, n, xnan, (const u32 *)cmask,
                                   (const unsigned char *)dflag, HF);
                SD_HIP(hipMemsetAsync(dirty, 0, (size_t)B * 16, s));
                SD_HIP(hipMemsetAsync(dcount, 0, 4, s));
                const i64 lslots = strict_lds_slots(n);
                const size_t tb = (size_t)lslots * 8 + 2 * (ST_ML_SEEN / 8);
                SD_HIP(hipFuncSetAttribute((const void *)strict_match_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb));
                hipLaunchKernelGGL(strict_match_lds_kernel, dim3((unsigned)nb), dim3(ST_ML_THREADS), tb, s, (const u32 *)masks,
                                   (const u64 *)HF, T, n, targets, q0, xnan, (const u32 *)cmask, dirty, dbits, dlist, dcount,
                                   (int)lslots, gen2 ? 0 : 1, force_global ? 1 : 0, out, jcols,
                                   (gen2 && xswitch("SD_STRICT_PAIRS2") != 1) ? ilist : (u32 *)nullptr);
                if (global_possible) {
                    hipLaunchKernelGGL(strict_match_clear_kernel, dim3((unsigned)((slots + 4095) / 4096), (unsigned)nb), dim3(ST_THREADS),
                                       0, s, (const u32 *)dirty, keys, cnt, slots);
                    hipLaunchKernelGGL(strict_match_insert_kernel, g1, dim3(ST_THREADS), 0, s, (const u32 *)masks, (const u64 *)HF, T, n,
                                       targets, q0, xnan, (const u32 *)cmask, (const u32 *)dirty, keys, cnt, slots);
                    hipLaunchKernelGGL(strict_match_total_kernel,
                                       dim3((unsigned)nb, (unsigned)((slots + ST_TOTAL_CHUNK - 1) / ST_TOTAL_CHUNK)), dim3(ST_THREADS),
                                       0, s, (const u32 *)cnt, slots, q0, xnan, (const u32 *)dirty, gen2 ? 0 : 1, out, jcols);
                }
                gate = dirty;
                if (gen2) {
                    // second pass: full masks (UN and DN) for the targets on the pair kernel's work list
                    StrictMaskOut mf{(u32 *)masks, nullptr, cmask, tiemask, dlist, dcount};
                    if (rankmasks)
                        hipLaunchKernelGGL(strict_masks_rank_kernel, g1b, dim3(ST_THREADS), 0, s, (const u32 *)R, (const u32 *)Yt, T, n, nb,
                                           mf);
                    else if (rank32)
                        hipLaunchKernelGGL(strict_masks_rank32_kernel, g1b, dim3(ST_THREADS), 0, s, (const u32 *)R, (const u32 *)Yt, T, n,
                                           nb, mf);
                    else
                        hipLaunchKernelGGL(strict_masks2_kernel, g1b, dim3(ST_THREADS), 0, s, Y, (const double *)Yt, T, n, nb, mf);
                }
            }
            if (gen2) {
                if (gate) {
                    // about 8 192 blocks in all (32 layers at least while a layer is small): with no dirty target every
                    // block only reads the list's length and leaves
                    const i64 per_layer = (i64)g2.x * g2.y;
                    i64 layers = (8192 + per_layer - 1) / per_layer;
                    if (layers < ST_PAIR_LAYERS && per_layer <= 256) layers = ST_PAIR_LAYERS;
                    if (layers > nb) layers = nb;
                    dim3 g2m(g2.x, g2.y, (unsigned)layers);
                    // cross-check builds, SD_STRICT_PAIRS2 = 1: every partner of a dirty curve through the mask test
                    if (xswitch("SD_STRICT_PAIRS2") == 1)
                        hipLaunchKernelGGL(strict_pairs2_kernel, g2m, dim3(ST_THREADS), 0, s, (const u32 *)masks, T, n, targets, q0, xnan,
                                           (const u32 *)dlist, (const u32 *)dcount, (const u64 *)dbits, out, jcols);
                    else
                    {
                        const int tiles = (int)((n + ST_P3_THREADS - 1) / ST_P3_THREADS);
                        const int chunks = n > 4096 ? 8 : (n > 1024 ? 4 : 1);
                        i64 grid = (i64)nb * tiles * chunks;
                        if (grid > 1024) grid = 1024;
                        hipLaunchKernelGGL(strict_pairs3_kernel, dim3((unsigned)grid), dim3(ST_P3_THREADS), 0, s, (const u32 *)masks, T, n,
                                           q0, (const u32 *)dlist, (const u32 *)dcount, (const u32 *)ilist, out, jcols, tiles, chunks);
                    }
                } else {
                    hipLaunchKernelGGL(strict_pairs2_kernel, g2, dim3(ST_THREADS), 0, s, (const u32 *)masks, T, n, targets, q0, xnan,
                                       (const u32 *)nullptr, (const u32 *)nullptr, (const u64 *)nullptr, out, jcols);
                }
                SD_HIP(hipGetLastError());
                continue;
            }
            // T > 1024: the pair kernel of the second generation keeps 2 x 32 words in registers and does not apply; the
            // targets that matching could not take (ties, NaN) go through the first generation below, which rebuilds
            // their masks in its own layout over the image just consumed (stream order)
        }
        if (Q && !(gen2 || match))      // external targets reach the first generation through their gathered copy
            hipLaunchKernelGGL(strict_gather_targets_kernel, dim3((unsigned)nb), dim3(ST_THREADS), 0, s, Y, T, n, targets, q0, Q, m,
                               Yt, xnan);
        hipLaunchKernelGGL(strict_masks_kernel, g1, dim3(ST_THREADS), 0, s, Y, T, n, targets, q0, masks, xnan, gate,
                           Q ? (const double *)Yt : (const double *)nullptr);
        if (W <= ST_WREG)
            hipLaunchKernelGGL((strict_pairs_kernel<true>), g2, dim3(ST_THREADS), 0, s, masks, T, n, targets, q0, xnan, gate, out, jcols);
        else
            hipLaunchKernelGGL((strict_pairs_kernel<false>), g2, dim3(ST_THREADS), 0, s, masks, T, n, targets, q0, xnan, gate, out, jcols);
        if (J >= 3) {
            i64 flat = n * n;
            dim3 g3((unsigned)((flat + ST_THREADS - 1) / ST_THREADS), 1, (unsigned)nb);
            hipLaunchKernelGGL((strict_subsets_kernel<3>), g3, dim3(ST_THREADS), 0, s, masks, T, n, targets, q0, xnan, out, jcols);
        }
        if (J >= 4) {
            i64 flat = n * n * n;
            dim3 g4((unsigned)((flat + ST_THREADS - 1) / ST_THREADS), 1, (unsigned)nb);
            hipLaunchKernelGGL((strict_subsets_kernel<4>), g4, dim3(ST_THREADS), 0, s, masks, T, n, targets, q0, xnan, out, jcols);
        }
        SD_HIP(hipGetLastError());
    }
    return SD_OK;
}

}  // namespace sd
