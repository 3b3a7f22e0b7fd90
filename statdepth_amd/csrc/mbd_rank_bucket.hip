// mbd_rank_bucket.hip -- K1+K2 rank formulation without a sort (default path of sd_mbd_counts, 2 <= n <= 16384, J <= 3).
//
// Same integers as the pairwise kernel and the reference's enumeration (_functional.py:246-251,
// _containment.py:75-77): per (curve, timepoint) the counts B (others strictly below) and A (strictly above),
// folded as C(n-1,j) - C(A,j) - C(B,j) over t.  A rank needs no sorted row, only "how many keys are below
// mine", and that splits into a coarse part any monotone bucketing answers with a histogram and a fine part
// that only the few keys sharing my bucket can change:
//
//   F  rank_bucket_kernel   -- one workgroup per row (persistent over rows g, g+G, ...), thread t owns curves
//      t, t+NT, ... for the whole launch.  Per row: (0) min/max of the row; (1) bucket b(x) = floor((x-lo) *
//      NB/(hi-lo)) -- rounding is monotone, so b is a non-decreasing function of x whatever the data -- and one
//      LDS atomic per key on a packed u16 histogram, whose return value is the key's slot inside its bucket;
//      (2) exclusive prefix sum of the NB counters; (3) keys scattered into bucket order; (4) every key
//      compares itself with the members of its own bucket: B = base + #(y < x), A = n - base - #(y <= x).
//      Keys past the end of a bucket belong to later buckets and compare greater (NaN sentinels follow the
//      last one), so the member loop needs no bounds.  Ties are exact by construction (equal values share a bucket).  The per-curve band counts
//      are accumulated in REGISTERS across the workgroup's rows; one partial total per (workgroup, curve)
//      goes to HBM at the end: the matrix is read once and nothing per (row, curve) is ever written.
//      NaNs stay out of the histogram and weigh in as pandas' skipna does; +-inf take the end buckets.
//      Tie-heavy rows (a bucket of 16 keys or more): when every bucket of the row holds ONE value -- quantised or
//      integer data, a row of equal values -- the ranks are closed form (less = 0, le = count) and the row needs no
//      member pass at all.  A row with an infinity, or with a bucket above CAP keys that mixes values, is set aside
//      and sorted by the same workgroup behind its row loop (rb_slow_row).
//      Rows much wider than their bulk (heavy tails, outlying curves): a row whose histogram shows a crowded bucket and
//      whose range exceeds the central bracket of its threads' own keys many times over is histogrammed again under the
//      three-piece map of rank_bucket.h (linear core + float-like tails, monotone for any data); the rows behind it take
//      their bracket with their range.  The external-target and medium kernels below use the same map.
//   Z  rank_finalize_kernel / rank_finalize4_kernel -- totals of the requested targets = sum of the workgroups'
//      partials (+ the fold of pair-image rows when the caller supplies one).
//
// HBM traffic per call: the matrix once (8 nT) + G partials of 8 n (J-1) bytes written and read once.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "sd_common.h"
#include "rank_sort.h"
#include "rank_bucket.h"

namespace sd {

constexpr u32 RB_AB_SPECIAL = 0xFFFFFFFFu;     // same encodings as mbd_rank_ab.hip
constexpr int RB_MEDIUM_MAXN = 40960;          // rank_medium_image_kernel: 2 or 3 blocks (measured: 4 blocks lose to the large-n route)
// RB_SMALL_E > 0 builds the kernels with at most that many keys per thread for 64 VGPRs and launches two workgroups
// (two rows) per CU.  Measured (n = 600..4096, T = 1000): no faster than one workgroup per CU -- the kernel is bound
// by VALU + LDS throughput, not by latency or barrier stalls -- so it is off.
#ifndef RB_SMALL_E
#define RB_SMALL_E 0
#endif
#define RB_WAVES_PER_EU(E) ((E) <= RB_SMALL_E ? 8 : 4)
#ifndef RB_REDIRECT
#define RB_REDIRECT 1                          // window reads a bucket does not need go to one shared NaN pair
#endif
// Rows whose range is much wider than their bulk (heavy tails, outlying curves): the three-piece map of rank_bucket.h.  Rounds
// 1 - 3 clamped the tails into the two end buckets of a range clipped to the waves' innermost extremes: three outlying curves
// were fine, heavy tails at every timepoint set every row aside for the sort (n = 14 000 Cauchy rows: 3.4 x the Gaussian time).
#ifndef RB_TIES
#define RB_TIES 1                              // tie-heavy rows: closed form when every bucket holds one value
#endif
constexpr int RB_PAD = 8;                      // NaN sentinels behind the bucket-ordered keys (never < or <= anything)

// ---------------------------------------------------------------------------------------------------
// The row the bucket map cannot spread (an infinite bracket, a crowded bucket that mixes values): in-LDS sort of the
// plain values + binary search, the method of rank_search_kernel (mbd_rank_ab.hip), for ONE row inside the
// bucket kernel, run behind the main row loop for the rows it set aside.  Thread t gets the pairs
// of its curves t + 1024 e in ab[e] (B | A << 16, RB_AB_SPECIAL where the curve is NaN or beyond n).
// The sort image overlays the LDS at Sm; s_cnt is one LDS word outside it.
// ---------------------------------------------------------------------------------------------------
constexpr int RB_SNT = 1024, RB_SE = 16;
__device__ __forceinline__ void rb_slow_row(const double *__restrict__ row, int n, double *Sm, u32 *s_cnt, u32 *ab,
                                         u32 *nnan_out) {
    using SC = R2Cfg<RB_SNT, RB_SE>;
    using Sorter = R2Sorter<RB_SNT, RB_SE>;
    constexpr int N = SC::N, LE = SC::LE, WB = SC::WB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n_act = ((n + WB - 1) / WB) * WB;
    const bool wreal = wave * WB < n_act;
    const double INF = __builtin_huge_val();
    if (t == 0) *s_cnt = 0;
    __syncthreads();
    double k[RB_SE];
    u32 mynan = 0;
    if (wreal) {
        const int i0 = wave * WB + lane;
#pragma unroll
        for (int e = 0; e < RB_SE; ++e) {
            double x = (i0 + e * 64 < n) ? row[i0 + e * 64] : INF;
            const bool isn = x != x;                                  // NaN -> +inf, counted (pandas skipna)
            mynan += isn ? 1u : 0u;
            k[e] = isn ? INF : x;
        }
    }
    if (mynan) atomicAdd(s_cnt, mynan);
    Sorter::sort(k, Sm, t, n_act, wreal, INF);
    if (wreal) {
        double *Sw = Sm + r2_base<0, LE>(t);
#pragma unroll
        for (int e = 0; e < RB_SE; ++e) Sw[r2_off<0, LE>(e)] = k[e];
    }
    __syncthreads();
    const u32 nnan = *s_cnt;
#pragma unroll 4
    for (int e = 0; e < RB_SE; ++e) {
        u32 r = RB_AB_SPECIAL;
        if (t + e * RB_SNT < n) {
            const double x = row[t + e * RB_SNT];
            if (x == x) {
                const int lo = r2_bound<N, SlotPad<LE>, false, false>(Sm, n_act, x, INF);   // x is in the row
                int hi = lo + 1, step = 1;                            // gallop over the tie run of x
                while (hi + step <= n_act && Sm[r2_phys<LE>(hi + step - 1)] <= x) { hi += step; step <<= 1; }
                while (step > 1) {
                    step >>= 1;
                    if (hi + step <= n_act && Sm[r2_phys<LE>(hi + step - 1)] <= x) hi += step;
                }
                // keys <= x within [0, n_act) are real non-NaN values unless x = +inf
                const u32 A = (x == INF) ? 0u : (u32)(n - hi) - nnan;
                r = (u32)lo | (A << 16);
            }
        }
        ab[e] = r;
    }
    *nnan_out = nnan;
    __syncthreads();                                                  // the image is free again
}

template <int NT, int E, int LNB, int U2>
struct RBCfg {
    static constexpr int NB = 1 << LNB;
    static constexpr int NW = NT / 64;
    static constexpr int W = NB / 2 / NT;                       // histogram words per thread in the prefix sum
    static constexpr int QW = W / 4;                            // ... as 16-byte quads
    static_assert(W >= 4 && W % 4 == 0, "whole quads of histogram words per thread");
    static_assert(NW == 16 || NW == 8, "cross-wave reductions are laid out for rows of 16 lanes");
    static_assert(NT == RB_SNT || NT == 512, "rb_slow_row is compiled for 1024 threads (512: two workgroups per CU, no cold path)");
    // positions: [0, n) keys, [n, n + RB_PAD) sentinels, DUMMY.. a scratch pair range for keys that are NaN
    static __host__ __device__ constexpr int dummy_pos(int n) { return (n + RB_PAD + 1) & ~1; }
    static __host__ __device__ constexpr size_t keys_slots(int n) { return (size_t)dummy_pos(n) + 2 * U2 + 2; }
    static constexpr int DEFW = 64;                              // bitmap of set-aside rows: 2048 rows per workgroup and launch
    // min/max partials, wave totals, bitmap, 64 spare bytes, the waves' brackets ([3][NW] float2: rb3_wave_bracket)
    static constexpr size_t HDR = (size_t)4 * NW * 8 + (size_t)NW * 4 + (size_t)DEFW * 4 + 64 + (size_t)6 * NW * 4;
    static constexpr int BRK_WORD = NW + DEFW + 16;                // of the brackets, in u32 words behind the min/max partials
    static __host__ __device__ constexpr size_t lds_bytes(int n) {
        const size_t a = keys_slots(n) * 8 + (size_t)(NB / 2 + 4) * 4;
        const size_t n_act = (size_t)((n + 1023) / 1024) * 1024;
        const size_t b = NT == RB_SNT ? (n_act + n_act / 16) * 8 : 0;  // sort image of rb_slow_row (R2Cfg<1024,16> slots)
        return HDR + (a > b ? a : b);
    }
};

// DBG (timing experiments only; 1-4: results invalid): 1 = stop after the range, 2 = after the histogram, 3 = after the
// prefix sum, 4 = after the scatter; 5 = full kernel with cycle stamps per phase printed by wave 0 of workgroup 0
// SEL (J == 2, every curve a target): the second and last launch behind rank_bucket32_kernel (mbd_rank_bucket32.hip).
// (1) Each workgroup sums its slice of curves over that kernel's Gsum u32 partial blocks and adds the totals to out_tot
// (zeroed by rank_bucket32_kernel) -- the finalize step, without a launch of its own.  (2) Only when that kernel flagged
// rows (*gate == epoch; else return): the flagged rows are ranked here in fp64 and their band counts added to out_tot
// as well.  Every workgroup derives the same ordered list of flagged rows and takes entries g, g + G, ...
template <int NT, int E, int LNB, int J, int CAP, int U2, int DBG = 0, bool A32 = false, bool SEL = false>
__global__ __launch_bounds__(NT, (NT == 512 ? 4 : RB_WAVES_PER_EU(E))) void rank_bucket_kernel(const double *__restrict__ Y, i64 n64, i64 row0, i64 rows,
                                                         u64 *__restrict__ partial, int p32, u32 *__restrict__ nnan_img,
                                                         const unsigned char *__restrict__ rowflag = nullptr,
                                                         const u32 *__restrict__ gate = nullptr, u32 epoch = 0,
                                                         u64 *__restrict__ out_tot = nullptr, int Gsum = 0,
                                                         const u32 *__restrict__ listbuf = nullptr,
                                                         u32 *__restrict__ fblocks = nullptr, u32 *__restrict__ done = nullptr,
                                                         int kspin = 1) {
    // J == 0: image mode (J >= 4 on the host side): no fold, the pairs (B | A << 16) of every (row, curve) go to the
    // pair image `partial` (as u32[rows][n]) and the rows' NaN counts to nnan_img; rank_accumulate*_kernel folds them
    using C = RBCfg<NT, E, LNB, U2>;
    constexpr int NB = C::NB, NW = C::NW, QW = C::QW;
    static_assert(2 * U2 - 1 <= RB_PAD, "the first member pass reads at most RB_PAD keys past the end");
    static_assert(CAP == 127 && NB <= 32768, "packing of (base, count, slot); the crowding test reads the counters' bits");
    constexpr int TRYB = (E == 16) ? 5 : 4;     // buckets of 16 (32 at NB = 8192 / n > 15360) keys: ties rather than density
    constexpr int NACC = (J == 2 || J == 0) ? 1 : (J - 1);
    u32 *ABimg = reinterpret_cast<u32 *>(partial);
    extern __shared__ double Sm[];
    const int n = (int)n64;
    double *red = Sm;                                                 // [2][NW][2] min/max partials
    u32 *wtot = reinterpret_cast<u32 *>(red + 4 * NW);                // [NW]
    u32 *defer = wtot + NW;                                           // [DEFW] bitmap of the rows set aside
    float2 *brk = reinterpret_cast<float2 *>(wtot + C::BRK_WORD);     // [3][NW] the waves' brackets: two parities + the second try
    // histogram before the keys: every LDS offset except the keys' end is a compile-time constant
    u32 *H = reinterpret_cast<u32 *>(Sm + C::HDR / 8);                // NB packed u16 counters, then bases
    double *S = reinterpret_cast<double *>(H + NB / 2 + 4);           // keys in bucket order + sentinels + dummy
    double *IMG = reinterpret_cast<double *>(H);                      // sort image of the rows set aside (overlays both)
    const unsigned short *H16 = reinterpret_cast<const unsigned short *>(H);
    const int t0 = threadIdx.x;
    const double INF = __builtin_huge_val();
    const double QNAN = __builtin_nan("");
    const int DUMMY = C::dummy_pos(n);
    const double2 *NANP = reinterpret_cast<const double2 *>(S + ((n + 1) & ~1));   // two of the sentinels, 16-byte aligned
    int t = t0;
    // SEL: the ordered list of flagged rows behind everything else in LDS (u16 row indices, rows <= 4096)
    const unsigned short *sel = reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(Sm) + C::lds_bytes(n));
    if constexpr (SEL) {
        static_assert(!SEL || (J == 2 && A32 && NT == RB_SNT), "second launch behind the 32-bit kernel: J = 2, u32 totals");
        {   // (1) finalize: totals of this workgroup's slice of curves = sum of the Gsum partial blocks (16-byte loads)
            const u32 *P = reinterpret_cast<const u32 *>(partial);
            u64 *redl = reinterpret_cast<u64 *>(Sm);                  // [64 slices][16 quads][4]
            const int nst = (n + 3) & ~3, nq = nst >> 2;              // the blocks lie nst words apart: 16-byte aligned whatever n
            const int q0 = (int)((i64)blockIdx.x * nq / gridDim.x), q1 = (int)((i64)(blockIdx.x + 1) * nq / gridDim.x);
            const int qx = t0 & 15, y = t0 >> 4;
            for (int qb = q0; qb < q1; qb += 16) {
                u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                if (qb + qx < q1) {
                    const u32 *pp = P + 4 * (qb + qx);
#pragma unroll 4
                    for (int g = y; g < Gsum; g += 64) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(pp + (size_t)g * nst);
                        a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
                    }
                }
                u64 *rl = redl + ((size_t)y * 16 + qx) * 4;
                rl[0] = a0; rl[1] = a1; rl[2] = a2; rl[3] = a3;
                __syncthreads();
                if (t0 < 64 && qb + (t0 >> 2) < q1 && 4 * (qb + (t0 >> 2)) + (t0 & 3) < n) {
                    u64 tot = 0;
#pragma unroll 8
                    for (int k = 0; k < 64; ++k) tot += redl[((size_t)k * 16 + (t0 >> 2)) * 4 + (t0 & 3)];
                    atomicAdd(&out_tot[4 * (qb + (t0 >> 2)) + (t0 & 3)], tot);
                }
                __syncthreads();
            }
        }
        {   // (1b) the keys rank_bucket32_kernel set aside (equal 32-bit images): their rows are NaN-free and finite, a group
            //      of equal images is complete in its workgroup's list, so B and A follow from the list and the doubles.
            //      Both lists of this workgroup at once, 64 lanes each: every lane fetches its own key's value (ONE round trip
            //      to memory; a loop with a load per partner was a chain of them), the groups meet in LDS.
            constexpr int LCAP = 64, LW = 1 + 2 * LCAP;               // R32_LCAP, R32_LIST_WORDS (mbd_rank_bucket32.hip)
            double *xs = reinterpret_cast<double *>(Sm);              // [2][LCAP] values, then keys and (B0 | E0 << 16)
            u32 *ks = reinterpret_cast<u32 *>(xs + 2 * LCAP), *bs_ = ks + 2 * LCAP;
            const int k = t0 >> 6, i = t0 & 63;
            for (int w0 = blockIdx.x; w0 < Gsum; w0 += 2 * (int)gridDim.x) {   // one trip: Gsum <= 2 gridDim on the host's grids
                u32 L = 0, key = 0, be = 0;
                double xv = 0.0;
                if (t0 < 2 * LCAP) {
                    const int w = w0 + k * (int)gridDim.x;
                    if (w < Gsum) {
                        const u32 *lb = listbuf + (size_t)w * LW;
                        L = lb[0];
                        if (L > (u32)LCAP) L = 0;
                        if ((u32)i < L) {
                            key = lb[1 + i];
                            be = lb[1 + LCAP + i];
                            xv = Y[(row0 + (i64)w + (i64)(key >> 14) * Gsum) * n + (key & 0x3FFFu)];
                        }
                    }
                    xs[t0] = xv; ks[t0] = key; bs_[t0] = be;
                }
                __syncthreads();
                if (t0 < 2 * LCAP && (u32)i < L) {
                    u32 B = be & 0xFFFFu, A = (u32)n - B - (be >> 16);
                    for (u32 j = 0; j < L; ++j) {
                        const u32 kj = ks[k * LCAP + j];
                        if (j != (u32)i && (kj >> 14) == (key >> 14) && (bs_[k * LCAP + j] & 0xFFFFu) == (be & 0xFFFFu)) {
                            const double xj = xs[k * LCAP + j];
                            B += (xj < xv) ? 1u : 0u;
                            A += (xj > xv) ? 1u : 0u;
                        }
                    }
                    const u64 v = (u64)n - 1;
                    atomicAdd(&out_tot[key & 0x3FFFu], (v * (v - 1) - (u64)A * (A - 1) - (u64)B * (B - 1)) >> 1);
                }
                __syncthreads();                                      // the next trip, and Sm is reused below
            }
        }
        if (*gate != epoch) return;                                   // (2) nothing was flagged
        unsigned short *selw = const_cast<unsigned short *>(sel);
        u32 nsel = 0;
        for (i64 c0 = 0; c0 < rows; c0 += NT) {
            const bool f = (c0 + t0 < rows) && rowflag[c0 + t0];
            const u64 m = __ballot(f);
            if ((t0 & 63) == 0) wtot[t0 >> 6] = (u32)__popcll(m);
            __syncthreads();
            u32 off = nsel, tot = 0;
            for (int w = 0; w < NW; ++w) {
                const u32 c = wtot[w];
                off += (w < (t0 >> 6)) ? c : 0u;
                tot += c;
            }
            if (f) selw[off + (u32)__popcll(m & ((1ull << (t0 & 63)) - 1ull))] = (unsigned short)(c0 + t0);
            nsel += tot;
            __syncthreads();
        }
        rows = nsel;                                                  // from here on "row r" is the r-th flagged row
    }
    auto rowptr = [&](i64 r) -> const double * {
        if constexpr (SEL) return Y + (row0 + (i64)sel[r]) * n;
        else return Y + (row0 + r) * n;
    };

    // LDS setup (once, and again behind rb_slow_row): sentinels + dummy range, empty histogram
    auto lds_setup = [&]() {
        for (int p = n + t; p < (int)C::keys_slots(n); p += NT) S[p] = QNAN;
        uint4 *Hq = reinterpret_cast<uint4 *>(H);
#pragma unroll
        for (int i = 0; i < QW; ++i) Hq[i * NT + t] = make_uint4(0, 0, 0, 0);
        if (t < 4) H[NB / 2 + t] = 0;
    };
    lds_setup();
    if (t < C::DEFW) defer[t] = 0;
    u32 ndefer = 0, rowidx = 0;                                       // block-uniform

    // curves of thread t: t, t + NT, ...; slots beyond n read as NaN and are treated like any other NaN
    double k[E];
    auto load_row = [&](i64 r) {
        const double *rp = rowptr(r) + t;
#pragma unroll
        for (int e = 0; e < E; ++e) k[e] = (e < E - 1 || t + (E - 1) * NT < n) ? rp[e * NT] : QNAN;
    };
    // J == 2: acc[e][0] = 2 * (sum of band counts) (one accumulator, see the fold below); else acc[e][j-2].
    // A32 (J == 2, p32 == 2: the host found ceil(rows / G) * n^2 < 2^32): 32-bit accumulators -- E registers fewer
    static_assert(!A32 || J == 2, "32-bit accumulators: J == 2 only");
    using ACC = std::conditional_t<A32, u32, u64>;
    ACC acc[E][NACC];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[e][j] = 0;

    bool heavy = false;                                               // block-uniform: the last row went under the three-piece map
    // (0) range of a row = min / max of its keys (NaN never wins: pandas skipna, _containment.py:68-69), per wave into
    // LDS.  It is computed for the NEXT row at the end of every iteration, where its VALU work fills the waits of
    // the LDS-bound member passes, and once here for the first row.
    auto row_range = [&](int parity) {
        double mn = INF, mx = -INF;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mn = rb_mm<false>(mn, k[e]);
            mx = rb_mm<true>(mx, k[e]);
        }
        if (heavy) {                                                  // block-uniform: the bracket of a row that follows a clipped one
            const float2 wb = rb3_wave_bracket<E, NT>(mn, mx);        // (from the threads' own extremes, before they merge)
            if ((t & 63) == 63) brk[parity * NW + (t >> 6)] = wb;
        }
        mn = rb_wave_allreduce<false>(mn);
        mx = rb_wave_allreduce<true>(mx);
        double *rp = red + parity * 2 * NW;
        if ((t & 63) == 63) { rp[2 * (t >> 6)] = mn; rp[2 * (t >> 6) + 1] = mx; }
    };
    long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;        // DBG == 5: cycles per phase, thread 0 of workgroup 0
    auto mark = [&](int ph) {
        if constexpr (DBG == 5) {
            const long long now = (long long)__builtin_readcyclecounter();
            stamp[ph] += now - tlast;
            tlast = now;
        }
    };
    int par = 0;
    if ((i64)blockIdx.x < rows) {
        load_row(blockIdx.x);
        row_range(0);
    }
    if constexpr (DBG == 5) tlast = (long long)__builtin_readcyclecounter();
    for (i64 r = blockIdx.x; r < rows; r += gridDim.x) {
        const i64 rnext = r + gridDim.x;
        // Per-row opaque copy of the thread id: every address below derives from it, so the compiler recomputes
        // those few ALU ops per row instead of hoisting loop-invariant address registers out of the row loop
        // and spilling the accumulators to make room (same device as in mbd_rank_ab.hip).
        t = t0;
        asm volatile("" : "+v"(t));
        const int lane = t & 63;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        double *redp = red + par * 2 * NW;
        const float2 *brkp = brk + par * NW;
        par ^= 1;
        mark(0);
        __syncthreads();                                              // barrier 1 (histogram is zero, S is free)
        mark(1);
        double lo, hi;
        {
            const double2 p = reinterpret_cast<const double2 *>(redp)[lane & (NW - 1)];
            lo = rb_readlane_f64(rb_row_allreduce<false>(p.x), 0);
            hi = rb_readlane_f64(rb_row_allreduce<true>(p.y), 0);
        }
        // range overflow -> 0 -> one crowded bucket; all values equal -> 0 -> one bucket of one value (closed form below)
        const double scale = (hi > lo) ? (double)NB / (hi - lo) : 0.0;
        // Every decision below is block-uniform.  The next row is loaded at ONE place (two load sites would
        // keep two copies of the key registers alive across the loop).  A row with an infinity or without any
        // value is set aside for the sort behind the loop.
        bool go = (hi >= lo) && (scale < INF) && (lo > -INF) && (hi < INF);
        if constexpr (DBG == 1) go = false;
        u32 bs[E];
        u32 nn = 0;                                                   // NaN others of this row (block-uniform)
        // block-uniform: a bucket above CAP keys / a bucket of 2^TRYB keys or more (ties?  see the member phase)
        bool crowded = false, trypure = false;
        // A row whose range is much wider than its bulk (heavy tails, outlying curves) goes under the three-piece map of
        // rank_bucket.h.  Such a row shows in the histogram of the linear map (a bucket of 2^TRYB keys or more, below) and is
        // histogrammed again; rows of one matrix resemble each other, so after the first one (`heavy`) the bracket is taken
        // with the row's range, BEFORE the histogram, until a row needs no clipping again.  Rows that spread well pay nothing.
        auto bracket_map = [&]() -> Rb3 {
            double mn = INF, mx = -INF;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                mn = rb_mm<false>(mn, k[e]);
                mx = rb_mm<true>(mx, k[e]);
            }
            const float2 wb = rb3_wave_bracket<E, NT>(mn, mx);
            if (lane == 63) brk[2 * NW + wave] = wb;
            __syncthreads();
            return rb3_make<NB>(lo, hi, brk[2 * NW + (lane & (NW - 1))], n);
        };
        auto histogram_clip = [&](const Rb3 &m3) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double x = k[e];
                u32 b = rb3_bucket<NB>(m3, x);
                b = (x == x) ? b : (u32)(NB + 2);
                const u32 sh = (b & 1u) * 16u;
                const u32 old = atomicAdd(&H[b >> 1], 1u << sh);
                bs[e] = b | (((old >> sh) & 0xFFFFu) << 16);
            }
        };
        if (go) {
            Rb3 m3;
            m3.clip = false;
            if (heavy) {                                              // block-uniform: the bracket came with the range
                m3 = rb3_make<NB>(lo, hi, brkp[lane & (NW - 1)], n);
                heavy = m3.clip;
            }
            // ---- (1) bucket + slot, branch-free.  min(fl(fl(x - lo) * scale), NB-1), negative -> 0, is non-decreasing
            //      in x for ANY lo and scale >= 0; NaNs count into a dummy word behind the histogram ----
            if (m3.clip) histogram_clip(m3);
            else
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double x = k[e];
                double u = (x - lo) * scale;
                u = u < (double)(NB - 1) ? u : (double)(NB - 1);      // NaN -> NB-1 (overridden below)
                u32 b;
                asm("v_cvt_u32_f64 %0, %1" : "=v"(b) : "v"(u));       // saturating: below the range -> bucket 0
                b = (x == x) ? b : (u32)(NB + 2);
                const u32 sh = (b & 1u) * 16u;
                const u32 old = atomicAdd(&H[b >> 1], 1u << sh);
                bs[e] = b | (((old >> sh) & 0xFFFFu) << 16);
            }
            mark(2);
            __syncthreads();                                          // barrier 2
            uint4 *Hq = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);   // this wave's 64*QW quads, quad i*64+lane
            if constexpr (DBG == 2) {
#pragma unroll
                for (int i = 0; i < QW; ++i) Hq[i * 64 + lane] = make_uint4(0, 0, 0, 0);
                go = false;
            }
            if constexpr (DBG != 2) {
                // ---- (2) exclusive prefix sum over the counters (conflict-free 16-byte accesses: lane <-> quad);
                //      a crowded bucket defers the row ----
                uint4 hq[QW];
                u32 runq[QW], inclq[QW], offq[QW];
                auto prefix_a = [&]() {
                    u32 ov = 0, wsum = 0;
#pragma unroll
                    for (int i = 0; i < QW; ++i) {
                        hq[i] = Hq[i * 64 + lane];
                        const u32 s4 = hq[i].x + hq[i].y + hq[i].z + hq[i].w;   // both halves at once: no half exceeds 16384
                        ov |= hq[i].x | hq[i].y | hq[i].z | hq[i].w;            // bit k of a half set <=> some counter has it
                        runq[i] = (s4 & 0xFFFFu) + (s4 >> 16);
                        inclq[i] = rb_wave_incl_scan(runq[i]);
                        offq[i] = wsum;
                        wsum += rb_readlane(inclq[i], 63);
                    }
                    // some counter > CAP (= 2^7 - 1) / some counter >= 2^TRYB: any of the bits from there up is set
                    constexpr u32 HIM = (0xFFFFu & ~(u32)CAP) * 0x10001u, TRM = (0xFFFFu & ~((1u << TRYB) - 1u)) * 0x10001u;
                    const bool wover = __ballot((ov & HIM) != 0) != 0, wtry = __ballot((ov & TRM) != 0) != 0;
                    if (lane == 63) wtot[wave] = wsum | (wover ? 0x80000000u : 0u) | (wtry ? 0x40000000u : 0u);
                };
                prefix_a();
                mark(3);
                __syncthreads();                                      // barrier 3
                u32 wt = wtot[lane & (NW - 1)];
                crowded = __ballot((wt >> 31) != 0) != 0;
                trypure = RB_TIES && __ballot((wt & 0x40000000u) != 0) != 0;
                if (__ballot((wt & 0x40000000u) != 0) != 0 && !m3.clip) {   // block-uniform, rare
                    // A bucket of 2^TRYB keys or more under the linear map: tie-heavy data -- or a range much wider than the
                    // row's bulk.  The bracket tells them apart; the second kind is histogrammed again under core + tails.
                    m3 = bracket_map();
                    if (m3.clip) {                                    // block-uniform
                        heavy = true;
#pragma unroll
                        for (int i = 0; i < QW; ++i) Hq[i * 64 + lane] = make_uint4(0, 0, 0, 0);
                        if (t < 4) H[NB / 2 + t] = 0;
                        __syncthreads();
                        histogram_clip(m3);
                        __syncthreads();
                        prefix_a();
                        __syncthreads();
                        wt = wtot[lane & (NW - 1)];
                        crowded = __ballot((wt >> 31) != 0) != 0;
                        trypure = RB_TIES && __ballot((wt & 0x40000000u) != 0) != 0;
                    }
                }
                const u32 wscan = rb_row_incl_scan(wt & 0x3FFFFFFFu);
                const u32 woff = wave ? rb_readlane(wscan, wave - 1) : 0u;
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    u32 base = woff + offq[i] + inclq[i] - runq[i];
                    uint4 o;
                    o.x = base | ((base + (hq[i].x & 0xFFFFu)) << 16);
                    base += (hq[i].x & 0xFFFFu) + (hq[i].x >> 16);
                    o.y = base | ((base + (hq[i].y & 0xFFFFu)) << 16);
                    base += (hq[i].y & 0xFFFFu) + (hq[i].y >> 16);
                    o.z = base | ((base + (hq[i].z & 0xFFFFu)) << 16);
                    base += (hq[i].z & 0xFFFFu) + (hq[i].z >> 16);
                    o.w = base | ((base + (hq[i].w & 0xFFFFu)) << 16);
                    base += (hq[i].w & 0xFFFFu) + (hq[i].w >> 16);
                    Hq[i * 64 + lane] = o;
                    // base past the last bucket = number of non-NaN keys
                    if (i == QW - 1 && t == NT - 1) H[NB / 2] = base;
                }
            }
        }
        if constexpr (DBG == 3) {
            if (go) {
                __syncthreads();
                uint4 *Hq = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);
#pragma unroll
                for (int i = 0; i < QW; ++i) Hq[i * 64 + lane] = make_uint4(0, 0, 0, 0);
            }
            go = false;
        }
        if constexpr (DBG == 0) {
            if (!go) {
                if (lo > hi) {
                    // no value at this timepoint (every curve NaN): nothing is contained, nothing to rank
                    if constexpr (J == 0) {
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            if (e < E - 1 || t + (E - 1) * NT < n) ABimg[r * n + t + e * NT] = RB_AB_SPECIAL;
                        if (t == 0) nnan_img[r] = (u32)n;
                    }
                } else {                                              // set aside: sorted behind the loop
                    if (t == 0) defer[rowidx >> 5] |= 1u << (rowidx & 31);
                    ++ndefer;
                }
            }
        }
        ++rowidx;
        u32 bc[E];                                                    // base | count << 16 | slot << 24; count 0: a NaN
        u32 (&pk)[E] = bs;                                            // member phase: less | le << 16, in bs's registers
        if (go) {
            mark(4);
            __syncthreads();                                          // barrier 4
            // ---- (3) scatter into bucket order (branch-free; NaNs write the dummy slot) ----
            const u32 nv = H[NB / 2];
            nn = (u32)n - nv;
            if (!trypure) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const u32 b = bs[e] & 0xFFFFu, slot = bs[e] >> 16;
                    const u32 base = H16[b], end = H16[b + 1];
                    const bool isk = b < (u32)NB;
                    S[isk ? base + slot : (u32)DUMMY] = k[e];
                    bc[e] = isk ? (base | ((end - base) << 16) | (slot << 24)) : 0u;
                }
            } else {                                                  // counts and slots beyond 8 bits: two words
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const u32 b = bs[e] & 0xFFFFu, slot = bs[e] >> 16;
                    const u32 base = H16[b], end = H16[b + 1];
                    const bool isk = b < (u32)NB;
                    S[isk ? base + slot : (u32)DUMMY] = k[e];
                    bc[e] = base | (slot << 16);
                    pk[e] = isk ? end - base : 0u;
                }
            }
            if (nn && t < RB_PAD) S[nv + t] = QNAN;                   // sentinels behind a row shortened by NaNs
        }
        if (rnext < rows) load_row(rnext);                            // next row in flight under the member passes
        if constexpr (DBG == 4) {
            if (go) {
                __syncthreads();
                uint4 *Hq = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);
#pragma unroll
                for (int i = 0; i < QW; ++i) Hq[i * 64 + lane] = make_uint4(0, 0, 0, 0);
            }
            go = false;
        }
        if (go) {
            mark(5);
            __syncthreads();                                          // barrier 5
            mark(6);
            // the histogram is dead until the next row's atomics (behind its barrier 1)
            {
                uint4 *Hq = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);
#pragma unroll
                for (int i = 0; i < QW; ++i) Hq[i * 64 + lane] = make_uint4(0, 0, 0, 0);
            }
            // ---- (4) rank inside the bucket.  First pass, branch-free: 2*U2 keys from the even position at or
            //      below the bucket's base (16-byte reads); a key in front of an odd base belongs to an earlier
            //      bucket, compares below, and is taken off again.  Second pass: the rest of longer buckets. ----
            // A row with a bucket of 2^TRYB keys or more is tie-heavy data more often than a dense cluster: if every
            // bucket of the row holds ONE value (all members equal their bucket's first key), the ranks are closed
            // form -- less = 0, le = count -- and there is no member pass at all.  Else the row goes the normal way,
            // or to the sort behind the loop when a bucket is above CAP.
            int mode = 0;                                             // 0: member passes, 1: closed form, 2: set aside
            if (trypure) {                                            // block-uniform
                bool pure = true;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const u32 base = bc[e] & 0xFFFFu, slot = bc[e] >> 16;
                    if (pk[e]) pure = pure && (S[base + slot] == S[base]);
                }
                if (__syncthreads_and(pure)) {
                    mode = 1;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const u32 cnt = pk[e];
                        bc[e] = (bc[e] & 0xFFFFu) | (cnt ? 0x10000u : 0u);
                        pk[e] = cnt << 16;
                    }
                } else if (crowded) {
                    mode = 2;
                    if constexpr (DBG == 0) {
                        if (t == 0) defer[(rowidx - 1) >> 5] |= 1u << ((rowidx - 1) & 31);
                        ++ndefer;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const u32 cnt = pk[e];
                        bc[e] = cnt ? ((bc[e] & 0xFFFFu) | (cnt << 16) | ((bc[e] >> 16) << 24)) : 0u;
                    }
                }
            }
            bool more = false;
            if (mode == 0) {
            // A key's window: its reads are issued together, ahead of the compares.  A lane whose bucket ends before
            // a pair reads the NaN pair behind the keys instead: same counts (NaN compares false), and lanes sharing
            // one address cost no bank-conflict cycles.  (Two windows in flight per lane: no faster, see DESIGN.)
            double xw[1];
            double2 yw[1][U2];
            auto window = [&](int e, int w) {
                const u32 base = bc[e] & 0xFFFFu, cnt = (bc[e] >> 16) & 0xFFu, slot = bc[e] >> 24;
                const u32 odd = base & 1u;
                xw[w] = S[base + slot];                               // own key (its registers hold the next row)
                const double2 *Sq = reinterpret_cast<const double2 *>(S + (base - odd));
#pragma unroll
                for (int u = 0; u < U2; ++u) {
#if RB_REDIRECT
                    yw[w][u] = *((u == 0 || cnt + odd > (u32)(2 * u)) ? Sq + u : NANP);
#else
                    yw[w][u] = Sq[u];
#endif
                }
            };
            window(0, 0);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int w = 0;
                const u32 base = bc[e] & 0xFFFFu, cnt = (bc[e] >> 16) & 0xFFu;
                const u32 odd = base & 1u;
                const double x = xw[w];
                u32 less = 0, le = 0;
#pragma unroll
                for (int u = 0; u < U2; ++u) {
                    less += (yw[w][u].x < x) ? 1u : 0u;
                    le += (yw[w][u].x <= x) ? 1u : 0u;
                    less += (yw[w][u].y < x) ? 1u : 0u;
                    le += (yw[w][u].y <= x) ? 1u : 0u;
                }
                pk[e] = (less - odd) | ((le - odd) << 16);
                more |= cnt + odd > (u32)(2 * U2);
                if (e + 1 < E) window(e + 1, 0);
            }
            if (__ballot(more && !(DBG == 6 && n > 0)) != 0) {   // DBG 6: never taken, code kept (timing experiment)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    u32 bce = bc[e];
                    asm volatile("" : "+v"(bce));                     // unpack again here: nothing kept alive from pass 1
                    const u32 base = bce & 0xFFFFu, cnt = (bce >> 16) & 0xFFu, slot = bce >> 24;
                    const u32 odd = base & 1u;
                    if (cnt + odd > (u32)(2 * U2)) {
                        const double x = S[base + slot];
                        const double2 *Sq = reinterpret_cast<const double2 *>(S + (base - odd));
                        u32 less = 0, le = 0;
#pragma unroll 1
                        for (u32 kk = 2 * U2; kk < cnt + odd; kk += 2) {
                            const double2 y = Sq[kk >> 1];
                            less += (y.x < x) ? 1u : 0u;
                            le += (y.x <= x) ? 1u : 0u;
                            less += (y.y < x) ? 1u : 0u;
                            le += (y.y <= x) ? 1u : 0u;
                        }
                        pk[e] += less | (le << 16);
                    }
                }
            }
            }                                                         // mode == 0
            // ---- fold ----
            if (mode != 2) {
            const u32 nv = (u32)n - nn;                               // non-NaN keys of the row
            const u32 v = nv - 1u;                                    // valid others of a non-NaN target
            const u32 R2 = v * (v - 1u + 2u * nn);                    // J == 2: 2 * [N v + C(v,2)]  (< 2^30)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const u32 base = bc[e] & 0xFFFFu, cnt = (bc[e] >> 16) & 0xFFu;
                const u32 B = base + (pk[e] & 0xFFFFu), A = nv - base - (pk[e] >> 16);
                if constexpr (J == 0) {
                    if (e < E - 1 || t + (E - 1) * NT < n) ABimg[r * n + t + e * NT] = cnt ? (B | (A << 16)) : RB_AB_SPECIAL;
                } else if constexpr (J == 2) {
                    // 2 * contained_2 = 2 N (v - A - B) + v(v-1) - A(A-1) - B(B-1)   (all terms < 2^30)
                    u32 q = __umul24(A, A - 1u) + __umul24(B, B - 1u);   // A, B < 2^15: 24-bit multiplies are exact
                    if (nn) q += 2u * __umul24(nn, A + B);
                    acc[e][0] += (ACC)(cnt ? R2 - q : 0u);
                } else {
                    if (cnt) {
                        u64 a7[JMAX - 1] = {0, 0, 0, 0, 0, 0, 0};
                        band_counts_add<(J > 0 ? J : 2)>(A, B, nn, (u64)(n - 1), a7);
#pragma unroll
                        for (int j = 0; j < J - 1; ++j) acc[e][j] += a7[j];
                    }
                }
            }
            }                                                         // mode != 2
        }
        if constexpr (J == 0) {
            if (go && t == 0) nnan_img[r] = nn;
        }
        if (rnext < rows) row_range(par);                             // the next row's keys have landed by now
        mark(7);
    }
    t = t0;
    if constexpr (DBG == 5) {   // in-kernel stamps (cdna_hip_programming.md, 7): where one wave's cycles go, barrier waits included
        if (blockIdx.x == 0 && t0 == 0)
            printf("rb stamps (cycles, wave 0 of workgroup 0): to-b1 %lld | b1-wait %lld | phase1 %lld | b2+prefixA %lld | b3+prefixB %lld | b4+scatter %lld | b5-wait %lld | members+fold+range %lld\n",
                   stamp[0], stamp[1], stamp[2], stamp[3], stamp[4], stamp[5], stamp[6], stamp[7]);
    }
    // ---- this workgroup's partial totals (u32 when the host found that they fit: J = 2, few rows per workgroup) ----
    u64 *P = partial + (size_t)blockIdx.x * (J > 0 ? J - 1 : 0) * n;
    // SEL: the totals of the flagged rows go to this workgroup's own u32 block (summed by the last workgroups to arrive, below)
    u32 *P32 = SEL ? fblocks + (size_t)blockIdx.x * ((n + 3) & ~3) : reinterpret_cast<u32 *>(partial) + (size_t)blockIdx.x * n;
#pragma unroll
    for (int e = 0; e < E; ++e)
        if ((J > 0) && (e < E - 1 || t + (E - 1) * NT < n)) {
            if constexpr (J == 2) {
                if constexpr (SEL) { if ((i64)blockIdx.x < rows) P32[t + e * NT] = (u32)(acc[e][0] >> 1); }   // (else: never read)
                else if (p32) P32[t + e * NT] = (u32)(acc[e][0] >> 1);
                else P[t + e * NT] = acc[e][0] >> 1;
            } else {
#pragma unroll
                for (int j = 0; j < J - 1; ++j) P[(size_t)j * n + t + e * NT] = acc[e][j];
            }
        }
    // ---- the rows the bucket map could not spread (an infinity, all values equal, a crowded bucket): sort + search.
    //      The main loop's accumulators are already in HBM and dead here, so this cold code shares no registers
    //      with the hot loop; each thread adds its own curves' band counts to what it stored above. ----
#ifndef RB_NO_COLD
    if constexpr (NT == RB_SNT)
    if (ndefer) {                                                     // block-uniform
        __syncthreads();
        rowidx = 0;
        for (i64 r = blockIdx.x; r < rows; r += gridDim.x, ++rowidx) {
            if (!((defer[rowidx >> 5] >> (rowidx & 31)) & 1u)) continue;
            u32 ab[RB_SE], nnan_s;
            rb_slow_row(rowptr(r), n, IMG, wtot, ab, &nnan_s);
#pragma unroll 1
            for (int e = 0; e < RB_SE; ++e) {
                if constexpr (J == 0) {
                    if (t + e * NT < n) ABimg[r * n + t + e * NT] = ab[e];
                    if (e == 0 && t == 0) nnan_img[r] = nnan_s;
                } else if (ab[e] != RB_AB_SPECIAL) {                         // implies t + e * NT < n
                    u64 a7[JMAX - 1] = {0, 0, 0, 0, 0, 0, 0};
                    band_counts_add<(J > 0 ? J : 2)>(ab[e] >> 16, ab[e] & 0xFFFFu, nnan_s, (u64)(n - 1), a7);
                    if (SEL || (J == 2 && p32)) P32[t + e * NT] += (u32)a7[0];
                    else {
#pragma unroll
                        for (int j = 0; j < J - 1; ++j) P[(size_t)j * n + t + e * NT] += a7[j];
                    }
                }
            }
        }
    }
#endif
    if constexpr (SEL) {
        // ---- the flagged rows' totals: G blocks of u32 -> out_tot.  Adding them with atomics from every workgroup costs
        //      G same-address atomics per curve (20 us at G = 256); instead each workgroup publishes its block, takes a
        //      ticket, and the LAST K workgroups to arrive each sum one slice of the curves over all blocks.  A waiting
        //      workgroup holds its CU, so the wait ends only if the workgroups that have not started yet still find one: the
        //      host passes K = kspin <= (workgroups the launch's CUs hold at once) - 1 (launch_bucket_sel_cfg: occupancy x the
        //      stream's CU mask; every workgroup resident at once: 16), and K = 1 never waits. ----
        const int G = (int)gridDim.x, K = G < kspin ? G : (kspin < 1 ? 1 : kspin);
        const int Gs = rows < (i64)G ? (int)rows : G;                 // workgroups that had a flagged row: the others' blocks are not read
        __syncthreads();                                              // this workgroup's block is complete (stores acknowledged) ...
        if (t0 == 0) {
            __threadfence();                                          // ... and written back for the device (ONE fence per workgroup)
            const u32 ticket = atomicAdd(done, 1u);                   // *done was zeroed by rank_bucket32_kernel
            wtot[0] = ticket;
            if (ticket >= (u32)(G - K))
                while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u32)G) __builtin_amdgcn_s_sleep(8);
            __threadfence();                                          // acquire: the other workgroups' blocks
        }
        __syncthreads();
        const u32 ticket = wtot[0];
        if (ticket < (u32)(G - K)) return;
        const int my = (int)ticket - (G - K);
        u64 *redl = reinterpret_cast<u64 *>(Sm);                      // [16 slices][64 quads][4]
        const int nst = (n + 3) & ~3, nq = nst >> 2;
        const int q0 = (int)((i64)my * nq / K), q1 = (int)((i64)(my + 1) * nq / K);
        const int qx = t0 & 63, y = t0 >> 6;
        for (int qb = q0; qb < q1; qb += 64) {
            u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            if (qb + qx < q1) {
                const u32 *pp = fblocks + 4 * (qb + qx);
#pragma unroll 8
                for (int g = y; g < Gs; g += 16) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(pp + (size_t)g * nst);
                    a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
                }
            }
            u64 *rl = redl + ((size_t)y * 64 + qx) * 4;
            rl[0] = a0; rl[1] = a1; rl[2] = a2; rl[3] = a3;
            __syncthreads();
            if (t0 < 256 && qb + (t0 >> 2) < q1 && 4 * (qb + (t0 >> 2)) + (t0 & 3) < n) {
                u64 tot = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) tot += redl[((size_t)k * 64 + (t0 >> 2)) * 4 + (t0 & 3)];
                atomicAdd(&out_tot[4 * (qb + (t0 >> 2)) + (t0 & 3)], tot);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// X: band totals of EXTERNAL curves (targets that are not members of the set; the homogeneity coefficients'
// `FunctionalDepth(F u {g}, to_compute=[g])`, homogeneity.py:101-128) through the same bucket structure.  Per row the
// set is histogrammed, prefix-summed and scattered exactly as in rank_bucket_kernel; then every external value x looks
// up its own bucket -- b(x) is monotone, so the keys of earlier buckets are below x and those of later ones above --
// and compares itself with that bucket's members: B = base + #(y < x), A = n_valid - base - #(y <= x).  O(n + m)
// per row instead of the pairwise kernel's O(n m).  No row is set aside: a row the map cannot spread (equal values,
// an infinity) simply lands in one bucket, and m targets walking n keys is still cheap.
// Thread t owns targets t, t + 1024, ... (EQ of them) and their totals for the whole launch.
// ---------------------------------------------------------------------------------------------------
template <int NT, int E, int LNB, int J, int EQ>
__global__ __launch_bounds__(NT) void rank_external_kernel(const double *__restrict__ Y, i64 n64, i64 rows,
                                                           const double *__restrict__ Q, i64 m64, i64 qstride,
                                                           u64 *__restrict__ partial) {
    using C = RBCfg<NT, E, LNB, 3>;
    constexpr int NB = C::NB, NW = C::NW, QW = C::QW;
    extern __shared__ double Sm[];
    const int n = (int)n64, m = (int)m64;
    double *red = Sm;                                                 // [2][NW][2] min/max partials
    u32 *wtot = reinterpret_cast<u32 *>(red + 4 * NW);                // [NW]
    float2 *brk = reinterpret_cast<float2 *>(wtot + C::BRK_WORD);     // [NW] the waves' brackets
    u32 *H = reinterpret_cast<u32 *>(Sm + C::HDR / 8);                // NB packed u16 counters, then bases
    double *S = reinterpret_cast<double *>(H + NB / 2 + 4);           // keys in bucket order
    const unsigned short *H16 = reinterpret_cast<const unsigned short *>(H);
    const int t0 = threadIdx.x;
    const double INF = __builtin_huge_val();
    const double QNAN = __builtin_nan("");
    const int DUMMY = C::dummy_pos(n);
    int t = t0;
    {
        uint4 *Hq = reinterpret_cast<uint4 *>(H);
#pragma unroll
        for (int i = 0; i < QW; ++i) Hq[i * NT + t] = make_uint4(0, 0, 0, 0);
        if (t < 4) H[NB / 2 + t] = 0;
    }
    double k[E];
    auto load_row = [&](i64 r) {
        const double *rp = Y + r * n + t;
#pragma unroll
        for (int e = 0; e < E; ++e) k[e] = (e < E - 1 || t + (E - 1) * NT < n) ? rp[e * NT] : QNAN;
    };
    u64 acc[EQ][JMAX - 1];
#pragma unroll
    for (int q = 0; q < EQ; ++q)
#pragma unroll
        for (int j = 0; j < JMAX - 1; ++j) acc[q][j] = 0;
    int par = 0;
    bool heavy = false;                                               // block-uniform: the last row went under the three-piece map
    if ((i64)blockIdx.x < rows) load_row(blockIdx.x);
    for (i64 r = blockIdx.x; r < rows; r += gridDim.x) {
        const i64 rnext = r + gridDim.x;
        t = t0;
        asm volatile("" : "+v"(t));                                   // per-row opaque thread id (see rank_bucket_kernel)
        const int lane = t & 63;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        // the row's external values (in flight under the set's phases)
        double xq[EQ];
#pragma unroll
        for (int q = 0; q < EQ; ++q) xq[q] = (t + q * NT < m) ? Q[r * qstride + t + q * NT] : QNAN;
        // ---- (0) range ----
        double mn = INF, mx = -INF;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mn = rb_mm<false>(mn, k[e]);
            mx = rb_mm<true>(mx, k[e]);
        }
        const double tmn = mn, tmx = mx;                              // this thread's own extremes (the bracket's groups)
        mn = rb_wave_allreduce<false>(mn);
        mx = rb_wave_allreduce<true>(mx);
        double *redp = red + par * 2 * NW;
        if (lane == 63) { redp[2 * wave] = mn; redp[2 * wave + 1] = mx; }
        par ^= 1;
        __syncthreads();                                              // barrier 1
        double lo, hi;
        Rb3 m3;
        m3.clip = false;
        {
            const double2 p = reinterpret_cast<const double2 *>(redp)[lane & (NW - 1)];
            lo = rb_readlane_f64(rb_row_allreduce<false>(p.x), 0);
            hi = rb_readlane_f64(rb_row_allreduce<true>(p.y), 0);
        }
        double scale = (double)NB / (hi - lo);
        // equal values, an infinity in the range, a range too small or too large: everything into one bucket
        if (!((hi > lo) && (scale < INF) && (lo > -INF) && (hi < INF))) scale = 0.0;
        auto bucket_of = [&](double x) {
            if (m3.clip) return rb3_bucket<NB>(m3, x);                // block-uniform
            double u = (x - lo) * scale;
            u = u > 0.0 ? u : 0.0;                                    // below the range, and NaN (0 * inf) -> 0
            u = u < (double)(NB - 1) ? u : (double)(NB - 1);
            return (u32)u;
        };
        // ---- (1) histogram (rows that resemble a clipped one take their bracket first: see rank_bucket_kernel) ----
        auto bracket_map = [&]() -> Rb3 {
            const float2 wb = rb3_wave_bracket<E, NT>(tmn, tmx);
            if (lane == 63) brk[wave] = wb;
            __syncthreads();
            return rb3_make<NB>(lo, hi, brk[lane & (NW - 1)], n);
        };
        if (heavy) {                                                  // block-uniform
            m3 = bracket_map();
            heavy = m3.clip;
        }
        u32 bs[E];
        auto histogram = [&]() {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double x = k[e];
                u32 b = bucket_of(x);
                b = (x == x) ? b : (u32)(NB + 2);
                const u32 sh = (b & 1u) * 16u;
                const u32 old = atomicAdd(&H[b >> 1], 1u << sh);
                bs[e] = b | (((old >> sh) & 0xFFFFu) << 16);
            }
        };
        histogram();
        __syncthreads();                                              // barrier 2
        // ---- (2) exclusive prefix sum ----
        {
            uint4 *Hq = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);
            uint4 hq[QW];
            u32 runq[QW], inclq[QW], offq[QW];
            auto prefix_a = [&]() {
                u32 wsum = 0, ov = 0;
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    hq[i] = Hq[i * 64 + lane];
                    const u32 lo16 = (hq[i].x & 0xFFFFu) + (hq[i].y & 0xFFFFu) + (hq[i].z & 0xFFFFu) + (hq[i].w & 0xFFFFu);
                    const u32 hi16 = (hq[i].x >> 16) + (hq[i].y >> 16) + (hq[i].z >> 16) + (hq[i].w >> 16);
                    ov |= hq[i].x | hq[i].y | hq[i].z | hq[i].w;      // bit k of a half set <=> some counter has it
                    runq[i] = lo16 + hi16;                            // counters reach n here: no packed addition
                    inclq[i] = rb_wave_incl_scan(runq[i]);
                    offq[i] = wsum;
                    wsum += rb_readlane(inclq[i], 63);
                }
                const bool wtry = __ballot((ov & 0xFFE0FFE0u) != 0) != 0;   // some bucket of 32 keys or more
                if (lane == 63) wtot[wave] = wsum | (wtry ? 0x40000000u : 0u);
            };
            prefix_a();
            __syncthreads();                                          // barrier 3
            u32 wt = wtot[lane & 15];
            if (__ballot((wt & 0x40000000u) != 0) != 0 && !m3.clip) { // block-uniform, rare: ties -- or a range much wider than
                m3 = bracket_map();                                   // the bulk (see rank_bucket_kernel): again under core + tails
                if (m3.clip) {
                    heavy = true;
#pragma unroll
                    for (int i = 0; i < QW; ++i) Hq[i * 64 + lane] = make_uint4(0, 0, 0, 0);
                    if (t < 4) H[NB / 2 + t] = 0;
                    __syncthreads();
                    histogram();
                    __syncthreads();
                    prefix_a();
                    __syncthreads();
                    wt = wtot[lane & 15];
                }
            }
            const u32 wscan = rb_row_incl_scan(wt & 0x3FFFFFFFu);
            const u32 woff = wave ? rb_readlane(wscan, wave - 1) : 0u;
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                u32 base = woff + offq[i] + inclq[i] - runq[i];
                uint4 o;
                o.x = base | ((base + (hq[i].x & 0xFFFFu)) << 16);
                base += (hq[i].x & 0xFFFFu) + (hq[i].x >> 16);
                o.y = base | ((base + (hq[i].y & 0xFFFFu)) << 16);
                base += (hq[i].y & 0xFFFFu) + (hq[i].y >> 16);
                o.z = base | ((base + (hq[i].z & 0xFFFFu)) << 16);
                base += (hq[i].z & 0xFFFFu) + (hq[i].z >> 16);
                o.w = base | ((base + (hq[i].w & 0xFFFFu)) << 16);
                base += (hq[i].w & 0xFFFFu) + (hq[i].w >> 16);
                Hq[i * 64 + lane] = o;
                if (i == QW - 1 && t == NT - 1) H[NB / 2] = base;     // number of non-NaN keys
            }
        }
        __syncthreads();                                              // barrier 4
        // ---- (3) scatter ----
        const u32 nv = H[NB / 2];
        const u32 nn = (u32)n - nv;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u32 b = bs[e] & 0xFFFFu, slot = bs[e] >> 16;
            const u32 base = H16[b];
            S[(b < (u32)NB) ? base + slot : (u32)DUMMY] = k[e];
        }
        if (rnext < rows) load_row(rnext);
        __syncthreads();                                              // barrier 5
        // ---- (4x) every external value against the members of its bucket ----
#pragma unroll
        for (int q = 0; q < EQ; ++q) {
            const double x = xq[q];
            if (x == x) {                                             // NaN target (and target slots beyond m): nothing
                const u32 b = bucket_of(x);
                const u32 base = H16[b], end = H16[b + 1];
                u32 less = 0, le = 0;
                // a value outside the set's range sits in an end bucket and still compares correctly
#pragma unroll 1
                for (u32 j = base; j < end; ++j) {
                    const double y = S[j];
                    less += (y < x) ? 1u : 0u;
                    le += (y <= x) ? 1u : 0u;
                }
                band_counts_add<J>(nv - base - le, base + less, nn, (u64)n, acc[q]);
            }
        }
        __syncthreads();                                              // barrier 6: S and the bases have been read
        {
            uint4 *Hq = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);
#pragma unroll
            for (int i = 0; i < QW; ++i) Hq[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        }
    }
    t = t0;
    u64 *P = partial + (size_t)blockIdx.x * (J - 1) * m;
#pragma unroll
    for (int q = 0; q < EQ; ++q)
        if (t + q * NT < m) {
#pragma unroll
            for (int j = 0; j < J - 1; ++j) P[(size_t)j * m + t + q * NT] = acc[q][j];
        }
}


// ---------------------------------------------------------------------------------------------------
// Z: out[q][j] (+)= sum_g partial[g][j][i(q)] + fold of the pair-image rows flagged in rowflag.
// block = 32 targets x 32 slices (of workgroups g, of rows), LDS tree over the slices.  rowflag == nullptr: no pair image
// (the bucket kernel ranks every row itself).
// ---------------------------------------------------------------------------------------------------
template <int J>
__global__ __launch_bounds__(1024) void rank_finalize_kernel(const u64 *__restrict__ partial, int G, int p32,
                                                             const u32 *__restrict__ AB, const u32 *__restrict__ nnan,
                                                             const unsigned char *__restrict__ rowflag, i64 rows, i64 n,
                                                             const i64 *__restrict__ targets, i64 tbegin, i64 m,
                                                             u64 *__restrict__ out, int first) {
    __shared__ u64 red[32][33];
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
    const i64 q = (i64)blockIdx.x * 32 + x;
    const i64 i = (q < m) ? (targets ? targets[q] : tbegin + q) : 0;
    // any deferred row at all?  (one pass over the flags by the whole block; normally none)
    int any = 0;
    if (rowflag) {                                                    // block-uniform
        for (i64 r = threadIdx.x; r < rows; r += 1024) any |= rowflag[r];
        any = __syncthreads_or(any);
    }
    u64 acc[JMAX - 1];
#pragma unroll
    for (int j = 0; j < JMAX - 1; ++j) acc[j] = 0;
    if (q < m) {
        if (J == 2 && p32) {
            const u32 *p = reinterpret_cast<const u32 *>(partial) + i;
#pragma unroll 8
            for (int g = y; g < G; g += 32) acc[0] += p[(size_t)g * n];
        } else {
            const u64 *p = partial + i;
#pragma unroll 8
            for (int g = y; g < G; g += 32) {
#pragma unroll
                for (int j = 0; j < J - 1; ++j) acc[j] += p[((size_t)g * (J - 1) + j) * n];
            }
        }
        if (any) {
            for (i64 r = y; r < rows; r += 32) {
                if (rowflag[r]) {
                    const u32 ab = AB[r * n + i];
                    if (ab != RB_AB_SPECIAL) band_counts_add<J>(ab >> 16, ab & 0xFFFFu, nnan[r], (u64)(n - 1), acc);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < J - 1; ++j) {
        red[y][x] = acc[j];
        __syncthreads();
        if (y == 0 && q < m) {
            u64 tot = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) tot += red[k][x];
            if (first) out[q * (J - 1) + j] = tot;
            else out[q * (J - 1) + j] += tot;
        }
        __syncthreads();
    }
}

// Z for the common case (J = 2, u32 partial totals, a contiguous 4-aligned block of targets, no pair image): thread
// (x, y) of a 256-thread block sums slice y of the G partial blocks for the 4 curves tbegin + 4 (8 blockIdx + x) ..
// with 16-byte loads; block = 32 curves x 32 slices.
__global__ __launch_bounds__(256) void rank_finalize4_kernel(const u32 *__restrict__ partial, int G, i64 n, i64 tbegin, i64 m,
                                                            u64 *__restrict__ out, int first) {
    __shared__ u64 red[32][8][4 + 1];
    const int x = threadIdx.x & 7, y = threadIdx.x >> 3;
    const i64 q = ((i64)blockIdx.x * 8 + x) * 4;                      // first of this thread's 4 targets
    u64 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (q < m) {
        const u32 *p = partial + tbegin + q;
#pragma unroll 8
        for (int g = y; g < G; g += 32) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + (size_t)g * n);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    }
    red[y][x][0] = a0; red[y][x][1] = a1; red[y][x][2] = a2; red[y][x][3] = a3;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int xx = threadIdx.x >> 2, c = threadIdx.x & 3;
        const i64 qq = ((i64)blockIdx.x * 8 + xx) * 4 + c;
        if (qq < m) {
            u64 tot = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) tot += red[k][xx][c];
            if (first) out[qq] = tot;
            else out[qq] += tot;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static int rb_cus() {
    static int cached[64];                                // per device; 0 = not asked yet (benign race: same value)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        if (dev >= 0 && dev < 64 && cached[dev] > 0) return cached[dev];
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        if (dev >= 0 && dev < 64) cached[dev] = cus;
    }
    return cus;
}

bool mbd_rank_bucket_supported(i64 T, i64 n, int J) {
    (void)T;
    i64 nmin = 1;                          // measured faster than the sort kernels from n = 600 to 16384
#ifdef SD_TUNING
    if (const char *e = getenv("SD_RB_MIN_N")) nmin = atoll(e);   // tuning experiments
#endif
    return n > nmin && n <= 16384 && J >= 2 && J <= 3;
}

// upper bound of the grid the launcher will use (the partial totals are sized by it)
int mbd_rank_bucket_max_grid() { return 2 * rb_cus(); }

// the workgroups' partial totals: all this path keeps in HBM
size_t mbd_rank_bucket_partial_bytes(i64 n, int J) { return align_up((size_t)mbd_rank_bucket_max_grid() * (J - 1) * n * 8, 256); }
size_t mbd_rank_bucket_workspace_bytes(i64 rows, i64 n, int J) {
    (void)rows;
    return mbd_rank_bucket_partial_bytes(n, J) + 512;
}

#ifndef RB_CAP
#define RB_CAP 127
#endif
template <int NT, int E, int LNB, int J, int U2>
static int launch_bucket_cfg(const double *Y, i64 n, i64 row0, i64 rows, u64 *partial, int p32, int G, hipStream_t s,
                             u32 *nnan_img = nullptr) {
    using C = RBCfg<NT, E, LNB, U2>;
    auto kf = rank_bucket_kernel<NT, E, LNB, J, RB_CAP, U2>;
    if constexpr (J == 2) {
        if (p32 == 2) kf = rank_bucket_kernel<NT, E, LNB, J, RB_CAP, U2, 0, true>;
    }
#ifdef SD_TUNING
    if constexpr (E == 10 && J == 2 && LNB == 15 && U2 == 3) {
        if (const char *d = getenv("SD_RB_DBG")) {        // timing experiments: truncated kernels (results invalid)
            switch (atoi(d)) {
                case 1: kf = rank_bucket_kernel<NT, E, LNB, J, RB_CAP, U2, 1>; break;
                case 2: kf = rank_bucket_kernel<NT, E, LNB, J, RB_CAP, U2, 2>; break;
                case 3: kf = rank_bucket_kernel<NT, E, LNB, J, RB_CAP, U2, 3>; break;
                case 4: kf = rank_bucket_kernel<NT, E, LNB, J, RB_CAP, U2, 4>; break;
                case 5: kf = rank_bucket_kernel<NT, E, LNB, J, RB_CAP, U2, 5>; break;
                case 6: kf = rank_bucket_kernel<NT, E, LNB, J, RB_CAP, U2, 6>; break;
            }
        }
    }
#endif
    const size_t lds = C::lds_bytes((int)n);
    if (lds > 163840) return fail(SD_ERR_UNSUPPORTED, "bucket kernel: %zu bytes of LDS for n=%lld", lds, (long long)n);
    SD_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kf, dim3(G), dim3(NT), lds, s, Y, n, row0, rows, partial, p32, nnan_img, (const unsigned char *)nullptr,
                       (const u32 *)nullptr, 0u, (u64 *)nullptr, 0, (const u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, 1);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// second launch behind rank_bucket32_kernel: the rows it flagged, totals added to its partial blocks
template <int E, int LNB>
static int launch_bucket_sel_cfg(const double *Y, i64 n, i64 row0, i64 rows, u64 *partial, int G, const unsigned char *rowflag,
                                 const u32 *gate, u32 epoch, u64 *out, int Gsum, const u32 *listbuf, u32 *fblocks, u32 *done,
                                 hipStream_t s) {
    using C = RBCfg<1024, E, LNB, 3>;
    auto kf = rank_bucket_kernel<1024, E, LNB, 2, RB_CAP, 3, 0, true, true>;
    const size_t lds = C::lds_bytes((int)n) + 8192;
    if (lds > 163840) return fail(SD_ERR_UNSUPPORTED, "bucket kernel: %zu bytes of LDS for n=%lld", lds, (long long)n);
    SD_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // workgroups that may wait for the others at the end of the kernel: fewer than the launch's CUs hold at once
    int kspin = 1;
    {
        static int occ_cached = 0;                            // per instantiation (benign race: same value)
        int occ = occ_cached;
        if (occ <= 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kf, 1024, lds) != hipSuccess || occ < 1) occ = 1;
            occ_cached = occ;
        }
        int usable = rb_cus();
        bool masked = false;
        uint32_t mask[16] = {0};
        if (hipExtStreamGetCUMask(s, 16, mask) == hipSuccess) {
            int bits = 0;
            for (int i = 0; i < 16; ++i) bits += __builtin_popcount(mask[i]);
            if (bits > 0 && bits < usable) { usable = bits; masked = true; }
        } else {
            (void)hipGetLastError();
        }
        const long cap = (long)occ * usable;
        kspin = cap >= G ? 16 : (int)(cap - 1 < 16 ? cap - 1 : 16);
        // A masked stream: the launch's CUs are not one pool.  Workgroups are dealt to the 8 XCDs in turn and stay there, so a
        // waiter on an XCD that the mask left one CU starves that XCD's remaining workgroups however many CUs the others have
        // free (found with tools/cu_mask_check.py: 8 CUs, one per XCD, 7 waiters -- the kernel never ended).  K = 1 never waits.
        if (masked) kspin = 1;
        if (kspin < 1) kspin = 1;
    }
    hipLaunchKernelGGL(kf, dim3(G), dim3(1024), lds, s, Y, n, row0, rows, partial, 2, (u32 *)nullptr, rowflag, gate, epoch, out, Gsum, listbuf,
                       fblocks, done, kspin);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

template <int J>
static int launch_bucket_j(const double *Y, i64 n, i64 row0, i64 rows, u64 *partial, int p32, int G, hipStream_t s) {
    // E = ceil(n / 1024); 8192 buckets up to n = 4096, 16384 above while keys + histogram fit the 160 KiB of LDS
    // (8192 at E = 16), 32768 at E = 10, 11 for J = 2 and the pair-image mode (measured: -2..-3.5 % on continuous
    // rows, -7 % with outlying curves, +6 % on tie-heavy rows whose closed form only pays the longer prefix sum;
    // J = 3 stays at 16384: the wider prefix arrays push it further into scratch); first member pass: 3 x 16 bytes
#define RB_ARGS Y, n, row0, rows, partial, p32, G, s
    const int E = (int)((n + 1023) / 1024);
#ifdef RB_HALF                                            // timing experiment: two 512-thread workgroups per CU (no cold path)
    if constexpr (J == 2) {
        if (E == 5) return launch_bucket_cfg<512, 10, 14, 2, 3>(RB_ARGS);
    }
#endif
#ifdef SD_TUNING
    if (E == 10 && J == 2) {                              // tuning experiments on the config-2 shape
        const char *eu = getenv("SD_RB_U2"), *el = getenv("SD_RB_LNB");
        const int u2 = eu ? atoi(eu) : 3, lnb = el ? atoi(el) : 15;
        if (lnb == 14 && u2 == 2) return launch_bucket_cfg<1024, 10, 14, 2, 2>(RB_ARGS);
        if (lnb == 14 && u2 == 4) return launch_bucket_cfg<1024, 10, 14, 2, 4>(RB_ARGS);
        if (lnb == 15 && u2 == 2) return launch_bucket_cfg<1024, 10, 15, 2, 2>(RB_ARGS);
        if (lnb == 14 && u2 == 3) return launch_bucket_cfg<1024, 10, 14, 2, 3>(RB_ARGS);
        if (lnb == 13 && u2 == 3) return launch_bucket_cfg<1024, 10, 13, 2, 3>(RB_ARGS);
        if (lnb == 13 && u2 == 4) return launch_bucket_cfg<1024, 10, 13, 2, 4>(RB_ARGS);
    }
#endif
    switch (E) {
        case 1: return launch_bucket_cfg<1024, 1, 13, J, 3>(RB_ARGS);
        case 2: return launch_bucket_cfg<1024, 2, 13, J, 3>(RB_ARGS);
        case 3: return launch_bucket_cfg<1024, 3, 13, J, 3>(RB_ARGS);
        case 4: return launch_bucket_cfg<1024, 4, 13, J, 3>(RB_ARGS);
        case 5: return launch_bucket_cfg<1024, 5, 14, J, 3>(RB_ARGS);
        case 6: return launch_bucket_cfg<1024, 6, 14, J, 3>(RB_ARGS);
        case 7: return launch_bucket_cfg<1024, 7, 14, J, 3>(RB_ARGS);
        case 8: return launch_bucket_cfg<1024, 8, 14, J, 3>(RB_ARGS);
        case 9: return launch_bucket_cfg<1024, 9, 14, J, 3>(RB_ARGS);
        case 10: return launch_bucket_cfg<1024, 10, (J == 3 ? 14 : 15), J, 3>(RB_ARGS);
        case 11: return launch_bucket_cfg<1024, 11, (J == 3 ? 14 : 15), J, 3>(RB_ARGS);
        case 12: return launch_bucket_cfg<1024, 12, 14, J, 3>(RB_ARGS);
        case 13: return launch_bucket_cfg<1024, 13, 14, J, 3>(RB_ARGS);
        case 14: return launch_bucket_cfg<1024, 14, 14, J, 3>(RB_ARGS);
        case 15: return launch_bucket_cfg<1024, 15, 14, J, 3>(RB_ARGS);
        case 16: return launch_bucket_cfg<1024, 16, 13, J, 3>(RB_ARGS);
    }
#undef RB_ARGS
    return fail(SD_ERR_UNSUPPORTED, "bucket kernel covers n <= 16384");
}

// image mode (J >= 4 on the host side): pairs of every (row, curve) to AB, NaN counts per row to nnan
int launch_rank_bucket_image(const double *Y, i64 n, i64 row0, i64 rows, u32 *AB, u32 *nnan, hipStream_t s) {
    const int cus = rb_cus();
    const int G = (int)(rows < cus ? rows : cus);
    if (rows > (i64)G * 2048) return fail(SD_ERR_INVALID, "bucket kernel: more than 2048 rows per workgroup in one launch");
    u64 *img = reinterpret_cast<u64 *>(AB);
#define RB_IM(E_, L_) case E_: return launch_bucket_cfg<1024, E_, L_, 0, 3>(Y, n, row0, rows, img, 0, G, s, nnan);
    switch ((int)((n + 1023) / 1024)) {
        RB_IM(1, 13) RB_IM(2, 13) RB_IM(3, 13) RB_IM(4, 13) RB_IM(5, 14) RB_IM(6, 14) RB_IM(7, 14) RB_IM(8, 14)
        RB_IM(9, 14) RB_IM(10, 15) RB_IM(11, 15) RB_IM(12, 14) RB_IM(13, 14) RB_IM(14, 14) RB_IM(15, 14) RB_IM(16, 13)
    }
#undef RB_IM
    return fail(SD_ERR_UNSUPPORTED, "bucket kernel covers n <= 16384");
}

// mbd_rank_bucket32.hip: 32-bit key images, two workgroups per CU
bool rank_bucket32_supported(i64 n, i64 rows, int cus);
size_t rank_bucket32_extra_bytes(i64 rows);
int launch_rank_bucket32(const double *Y, i64 n, i64 row0, i64 rows, u32 *partial, unsigned char *rowflag, u32 *gate, u32 epoch,
                         u64 *out_zero, u32 *listbuf, int G, hipStream_t s);
size_t rank_bucket32_list_bytes(int G);
u32 rank_bucket32_epoch();

// J = 2, 4096 < n <= 11264, every curve a target: rank_bucket32_kernel (two workgroups per CU; zeroes out when `first`) + the
// fp64 kernel's SEL form (sums the partial blocks into out; ranks the rows the first kernel flagged).  Two launches.  The
// flags and the gate word sit behind the 2 * cus u32 partial blocks, inside the space sized for u64 blocks.
// what the two-launch path carves out of the partial-block space: its u32 blocks, row flags, gate words, lists, the second
// launch's blocks -- and, last, n u64 totals of ALL curves for calls that ask for a subset of the targets
static size_t two_level_used_bytes(i64 n, i64 rows) {
    const size_t nst = (size_t)((n + 3) & ~3);
    return (size_t)2 * rb_cus() * nst * 4 + align_up((size_t)rows, 64) + 64 + 256 + rank_bucket32_list_bytes(2 * rb_cus()) + 256 +
           (size_t)rb_cus() * nst * 4 + 256;
}
bool rank_bucket_two_level_supported(i64 n, i64 rows) {
    // the all-totals block sits behind the largest batch's region (rank_bucket_two_level_all_totals): the same expression here
    return rank_bucket32_supported(n, rows, rb_cus()) &&
           align_up(two_level_used_bytes(n, 4096), 256) + (size_t)n * 8 <= mbd_rank_bucket_partial_bytes(n, 2);
}
u64 *rank_bucket_two_level_all_totals(u64 *partial, i64 n) {        // sized for the largest batch (4 096 rows)
    return reinterpret_cast<u64 *>(reinterpret_cast<char *>(partial) + align_up(two_level_used_bytes(n, 4096), 256));
}

// out[q] (=|+=) all[target q]: the subset of the totals a call asked for
__global__ __launch_bounds__(256) void rank_gather_totals_kernel(const u64 *__restrict__ all, const i64 *__restrict__ targets, i64 tbegin,
                                                                i64 m, u64 *__restrict__ out) {
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q < m) out[q] = all[targets ? targets[q] : tbegin + q];
}
int launch_rank_gather_totals(const u64 *all, const i64 *targets, i64 tbegin, i64 m, u64 *out, hipStream_t s) {
    hipLaunchKernelGGL(rank_gather_totals_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, all, targets, tbegin, m, out);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

int launch_rank_bucket_two_level(const double *Y, i64 n, i64 row0, i64 rows, u64 *partial, u64 *out, int first, hipStream_t s) {
    const int cus = rb_cus();
    const int G = (int)(rows < 2 * cus ? rows : 2 * cus);
    u32 *P32 = reinterpret_cast<u32 *>(partial);
    unsigned char *rowflag = reinterpret_cast<unsigned char *>(P32 + (size_t)2 * cus * ((n + 3) & ~3));
    u32 *gate = reinterpret_cast<u32 *>(rowflag + align_up((size_t)rows, 64));
    u32 *listbuf = gate + 16;                                         // a list of set-aside keys per workgroup of the first launch
    // the second launch's own blocks (totals of the flagged rows), 16-byte aligned, behind the lists; gate[2]: arrival counter
    u32 *fblocks = reinterpret_cast<u32 *>(align_up((size_t)(listbuf) + rank_bucket32_list_bytes(G), 256));
    const u32 epoch = rank_bucket32_epoch();
    int rc = launch_rank_bucket32(Y, n, row0, rows, P32, rowflag, gate, epoch, first ? out : nullptr, listbuf, G, s);
    if (rc) return rc;
    const int G2 = G < cus ? G : cus;
#define RB_SEL(E_) case E_: return launch_bucket_sel_cfg<E_, 14>(Y, n, row0, rows, partial, G2, rowflag, gate, epoch, out, G, listbuf, fblocks, gate + 2, s);
    switch ((int)((n + 1023) / 1024)) {
        RB_SEL(3) RB_SEL(4) RB_SEL(5) RB_SEL(6) RB_SEL(7) RB_SEL(8) RB_SEL(9) RB_SEL(11)
        case 10: return launch_bucket_sel_cfg<10, 15>(Y, n, row0, rows, partial, G2, rowflag, gate, epoch, out, G, listbuf, fblocks, gate + 2, s);
    }
#undef RB_SEL
    return fail(SD_ERR_UNSUPPORTED, "two-level bucket path covers 4096 < n <= 11264");
}

// rows [row0, row0 + rows): partial totals of every curve per workgroup; returns the grid used (number of partial
// blocks) in *G_out
int launch_rank_bucket(const double *Y, i64 n, i64 row0, i64 rows, int J, u64 *partial, int *p32_out, int *G_out,
                       hipStream_t s) {
    // one workgroup per CU; two where the kernel is built for 64 VGPRs (few keys per thread)
    int cus = rb_cus() * (((n + 1023) / 1024) <= RB_SMALL_E ? 2 : 1);
#ifdef RB_HALF
    if ((n + 1023) / 1024 == 5 && J == 2) cus = 2 * rb_cus();
#endif
    const int G = (int)(rows < cus ? rows : cus);
    *G_out = G;
    if (rows > (i64)G * 2048) return fail(SD_ERR_INVALID, "bucket kernel: more than 2048 rows per workgroup in one launch");
    // J = 2: a workgroup's total is at most ceil(rows / G) * C(n-1, 2); below 2^32 the partials travel as u32
    const u64 per_wg = (u64)((rows + G - 1) / G) * ((u64)(n - 1) * (u64)(n - 2) / 2);
    int p32 = (J == 2 && per_wg < ((u64)1 << 32)) ? 1 : 0;
    // ... and 2 * total stays below 2^32 too (2 * C(v,2) + 2 N v < n^2 per row): the kernel accumulates in 32 bits
    if (p32 && (u64)((rows + G - 1) / G) * (u64)n * (u64)n < ((u64)1 << 32) && xswitch("SD_RB_ACC64") == 0) p32 = 2;
    *p32_out = p32;
    if (J == 2) return launch_bucket_j<2>(Y, n, row0, rows, partial, p32, G, s);
    if (J == 3) return launch_bucket_j<3>(Y, n, row0, rows, partial, p32, G, s);
    return fail(SD_ERR_UNSUPPORTED, "bucket kernel covers J in [2,3]");
}

int launch_rank_finalize(const u64 *partial, int G, int p32, const u32 *AB, const u32 *nnan, const unsigned char *rowflag,
                         i64 rows, i64 n, const i64 *targets, i64 tbegin, i64 m, int J, u64 *out, int first,
                         hipStream_t s) {
    dim3 grid((unsigned)((m + 31) / 32));
    if (J == 2 && p32 && !targets && !rowflag && n % 4 == 0 && tbegin % 4 == 0 && m % 4 == 0 && xswitch("SD_RB_FINAL1") == 0)
        hipLaunchKernelGGL(rank_finalize4_kernel, grid, dim3(256), 0, s, reinterpret_cast<const u32 *>(partial), G, n, tbegin, m,
                           out, first);
    else if (J == 2)
        hipLaunchKernelGGL((rank_finalize_kernel<2>), grid, dim3(1024), 0, s, partial, G, p32, AB, nnan, rowflag, rows, n, targets,
                           tbegin, m, out, first);
    else
        hipLaunchKernelGGL((rank_finalize_kernel<3>), grid, dim3(1024), 0, s, partial, G, p32, AB, nnan, rowflag, rows, n, targets,
                           tbegin, m, out, first);
    SD_HIP(hipGetLastError());
    return SD_OK;
}


// ---------------------------------------------------------------------------------------------------
// M: pair image of rows of 16 384 < n <= 40 960 curves (the "medium" sizes a 2- to 4-GPU time-sharded step sees:
// n = N * 10^4) as 2 or 3 column blocks through ONE workgroup.  Ranks are additive over column blocks: per row each
// block in turn is histogrammed, prefix-summed and scattered exactly as in rank_external_kernel, and EVERY key of the
// row -- the block's own (still in registers) and the other blocks' (read again: the row sits in L2) -- looks up its
// bucket there: B += base + #(y < x), A += n_valid - base - #(y <= x) (a key meets itself in its own block: `<=` keeps
// it out of A, `<` out of B).  The running counts wait in the image itself (this thread's own words).  One build and
// NBLK look-ups per key: about the bucket kernel's cost per key at two blocks, against three passes over global memory
// on the large-n route.  Output as the bucket kernel's image mode (B | A << 16, 0xFFFFFFFF for NaN; counts < 2^16),
// folded by the same rank_accumulate kernels.
// ---------------------------------------------------------------------------------------------------
template <int NT, int E, int LNB>
__global__ __launch_bounds__(NT) void rank_medium_image_kernel(const double *__restrict__ Y, i64 n64, i64 row0, i64 rows, int nblk,
                                                               u32 *__restrict__ AB, u32 *__restrict__ nnan_img) {
    using C = RBCfg<NT, E, LNB, 3>;
    constexpr int NB = C::NB, NW = C::NW, QW = C::QW;
    extern __shared__ double Sm[];
    const int n = (int)n64;
    constexpr int BS = E * NT;                                        // block j: curves [j BS, min((j + 1) BS, n))
    double *red = Sm;                                                 // [2][NW][2] min/max partials
    u32 *wtot = reinterpret_cast<u32 *>(red + 4 * NW);                // [NW]
    float2 *brk = reinterpret_cast<float2 *>(wtot + C::BRK_WORD);     // [NW] the waves' brackets
    u32 *H = reinterpret_cast<u32 *>(Sm + C::HDR / 8);                // NB packed u16 counters, then bases
    double *S = reinterpret_cast<double *>(H + NB / 2 + 4);           // keys in bucket order
    const unsigned short *H16 = reinterpret_cast<const unsigned short *>(H);
    const int t0 = threadIdx.x;
    const double INF = __builtin_huge_val();
    const double QNAN = __builtin_nan("");
    const int DUMMY = C::dummy_pos(BS);
    int t = t0;
    {
        uint4 *Hq = reinterpret_cast<uint4 *>(H);
#pragma unroll
        for (int i = 0; i < QW; ++i) Hq[i * NT + t] = make_uint4(0, 0, 0, 0);
        if (t < 4) H[NB / 2 + t] = 0;
    }
    int par = 0;
    bool heavy = false;                                               // block-uniform: the last block went under the three-piece map
    for (i64 r = blockIdx.x; r < rows; r += gridDim.x) {
        t = t0;
        asm volatile("" : "+v"(t));                                   // per-row opaque thread id (see rank_bucket_kernel)
        const int lane = t & 63;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        const double *row = Y + (row0 + r) * n;
        u32 *img = AB + r * n;                                        // B | A << 16 so far, per key
        u32 nn_tot = 0;
        for (int jb = 0; jb < nblk; ++jb) {
            const int boff = jb * BS;
            const int bsz = n - boff < BS ? n - boff : BS;
            double kk[E];
#pragma unroll
            for (int e = 0; e < E; ++e) kk[e] = (t + e * NT < bsz) ? row[boff + t + e * NT] : QNAN;
            // ---- (0) range of the block ----
            double mn = INF, mx = -INF;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double xf = (kk[e] == INF || kk[e] == -INF) ? QNAN : kk[e];   // infinities take the end buckets anyway
                mn = rb_mm<false>(mn, xf);
                mx = rb_mm<true>(mx, xf);
            }
            const double tmn = mn, tmx = mx;                          // this thread's own extremes (the bracket's groups)
            mn = rb_wave_allreduce<false>(mn);
            mx = rb_wave_allreduce<true>(mx);
            double *redp = red + par * 2 * NW;
            if (lane == 63) { redp[2 * wave] = mn; redp[2 * wave + 1] = mx; }
            par ^= 1;
            __syncthreads();                                          // barrier 1
            double lo, hi;
            Rb3 m3;
            m3.clip = false;
            {
                const double2 p = reinterpret_cast<const double2 *>(redp)[lane & (NW - 1)];
                lo = rb_readlane_f64(rb_row_allreduce<false>(p.x), 0);
                hi = rb_readlane_f64(rb_row_allreduce<true>(p.y), 0);
            }
            const bool flat = hi == lo;                               // every finite key of the block holds one value
            double scale = (double)NB / (hi - lo);
            // equal values, an infinity in the range, a range too small or too large: everything into one bucket
            if (!((hi > lo) && (scale < INF) && (lo > -INF) && (hi < INF))) scale = 0.0;
            auto bucket_of = [&](double x) {
                if (m3.clip) return rb3_bucket<NB>(m3, x);            // block-uniform
                double u = (x - lo) * scale;
                u = u > 0.0 ? u : 0.0;                                // below the range, and NaN (0 * inf) -> 0
                u = u < (double)(NB - 1) ? u : (double)(NB - 1);
                return (u32)u;
            };
            // ---- (1) histogram ----
            u32 bs[E];
            bool trypure = false;                                     // block-uniform
            auto bracket_map = [&]() -> Rb3 {
                const float2 wb = rb3_wave_bracket<E, NT>(tmn, tmx);
                if (lane == 63) brk[wave] = wb;
                __syncthreads();
                return rb3_make<NB>(lo, hi, brk[lane & (NW - 1)], bsz);
            };
            if (heavy && !flat) {                                     // block-uniform: the last block went under core + tails
                m3 = bracket_map();
                heavy = m3.clip;
            }
            auto histogram = [&]() {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double x = kk[e];
                    u32 b = bucket_of(x);
                    b = (x == x) ? b : (u32)(NB + 2);
                    const u32 sh = (b & 1u) * 16u;
                    const u32 old = atomicAdd(&H[b >> 1], 1u << sh);
                    bs[e] = b | (((old >> sh) & 0xFFFFu) << 16);
                }
            };
            histogram();
            __syncthreads();                                          // barrier 2
            // ---- (2) exclusive prefix sum ----
            {
                uint4 *Hq = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);
                uint4 hq[QW];
                u32 runq[QW], inclq[QW], offq[QW];
                auto prefix_a = [&]() {
                    u32 wsum = 0, ov = 0;
#pragma unroll
                    for (int i = 0; i < QW; ++i) {
                        hq[i] = Hq[i * 64 + lane];
                        const u32 lo16 = (hq[i].x & 0xFFFFu) + (hq[i].y & 0xFFFFu) + (hq[i].z & 0xFFFFu) + (hq[i].w & 0xFFFFu);
                        const u32 hi16 = (hq[i].x >> 16) + (hq[i].y >> 16) + (hq[i].z >> 16) + (hq[i].w >> 16);
                        ov |= hq[i].x | hq[i].y | hq[i].z | hq[i].w;  // bit k of a half set <=> some counter has it
                        runq[i] = lo16 + hi16;
                        inclq[i] = rb_wave_incl_scan(runq[i]);
                        offq[i] = wsum;
                        wsum += rb_readlane(inclq[i], 63);
                    }
                    // a bucket of 16 keys or more is tie-heavy data more often than a dense cluster (see rank_bucket_kernel)
                    const bool wtry = __ballot((ov & 0xFFF0FFF0u) != 0) != 0;
                    if (lane == 63) wtot[wave] = wsum | (wtry ? 0x40000000u : 0u);
                };
                prefix_a();
                __syncthreads();                                      // barrier 3
                u32 wt = wtot[lane & 15];
                trypure = __ballot((wt & 0x40000000u) != 0) != 0;
                if (trypure && !flat && !m3.clip) {                   // block-uniform, rare: ties -- or a range much wider than
                    m3 = bracket_map();                               // the bulk (see rank_bucket_kernel): again under core + tails
                    if (m3.clip) {
                        heavy = true;
#pragma unroll
                        for (int i = 0; i < QW; ++i) Hq[i * 64 + lane] = make_uint4(0, 0, 0, 0);
                        if (t < 4) H[NB / 2 + t] = 0;
                        __syncthreads();
                        histogram();
                        __syncthreads();
                        prefix_a();
                        __syncthreads();
                        wt = wtot[lane & 15];
                        trypure = __ballot((wt & 0x40000000u) != 0) != 0;
                    }
                }
                const u32 wscan = rb_row_incl_scan(wt & 0x3FFFFFFFu);
                const u32 woff = wave ? rb_readlane(wscan, wave - 1) : 0u;
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    u32 base = woff + offq[i] + inclq[i] - runq[i];
                    uint4 o;
                    o.x = base | ((base + (hq[i].x & 0xFFFFu)) << 16);
                    base += (hq[i].x & 0xFFFFu) + (hq[i].x >> 16);
                    o.y = base | ((base + (hq[i].y & 0xFFFFu)) << 16);
                    base += (hq[i].y & 0xFFFFu) + (hq[i].y >> 16);
                    o.z = base | ((base + (hq[i].z & 0xFFFFu)) << 16);
                    base += (hq[i].z & 0xFFFFu) + (hq[i].z >> 16);
                    o.w = base | ((base + (hq[i].w & 0xFFFFu)) << 16);
                    base += (hq[i].w & 0xFFFFu) + (hq[i].w >> 16);
                    Hq[i * 64 + lane] = o;
                    if (i == QW - 1 && t == NT - 1) H[NB / 2] = base; // number of non-NaN keys of the block
                }
            }
            __syncthreads();                                          // barrier 4
            // ---- (3) scatter ----
            const u32 nv = H[NB / 2];
            nn_tot += (u32)bsz - nv;                                  // pad slots beyond the block are NaN too: not in bsz
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const u32 b = bs[e] & 0xFFFFu, slot = bs[e] >> 16;
                const u32 base = H16[b];
                S[(b < (u32)NB) ? base + slot : (u32)DUMMY] = kk[e];
            }
            // Tie-heavy rows (values on a grid, duplicated curves): if every bucket of the block holds ONE value -- every key
            // equals its bucket's first -- a look-up is a compare with that value, not a walk over hundreds of members
            bool allpure = false;                                     // block-uniform
            if (trypure && !flat) {
                __syncthreads();
                bool pure = true;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const u32 b = bs[e] & 0xFFFFu;
                    if (b < (u32)NB) pure = pure && (S[H16[b]] == kk[e]);
                }
                allpure = __syncthreads_and(pure) != 0;
            }
            u32 flat_lo = 0, flat_hi = 0;                             // a flat block's -inf / +inf keys
            if (flat) {
                u32 ml = 0, mh = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) { ml += kk[e] == -INF; mh += kk[e] == INF; }
                for (int k2 = 0; k2 < E; ++k2) {                      // block-wide sums, one count at a time (rare path)
                    flat_lo += __syncthreads_count(k2 < (int)ml);
                    flat_hi += __syncthreads_count(k2 < (int)mh);
                }
            }
            __syncthreads();                                          // barrier 5
            // ---- (4) every key of the row against the members of its bucket in this block ----
            // (a six-key window read ahead of the compares, as in rank_bucket_kernel, was measured here and changed nothing
            // at E = 10 and cost 15 % at E = 15: the look-ups are not what this kernel waits for -- its global loads are)
            auto lookup = [&](double x) -> u32 {
                if (flat) {                                           // block-uniform: one bucket of equal keys (and infinities)
                    const u32 neq = nv - flat_lo - flat_hi;
                    const u32 less = (x > -INF ? flat_lo : 0u) + (x > lo ? neq : 0u);
                    const u32 le = flat_lo + (x >= lo ? neq : 0u) + (x >= INF ? flat_hi : 0u);
                    return (x == x) ? (less | ((nv - le) << 16)) : 0u;
                }
                u32 c = 0;
                if (x == x) {
                    const u32 b = bucket_of(x);
                    const u32 base = H16[b], end = H16[b + 1];
                    u32 less = 0, le = 0;
                    if (allpure) {                                    // block-uniform: one value per bucket
                        const double y = S[base < end ? base : 0u];
                        less = (base < end && y < x) ? end - base : 0u;
                        le = (base < end && y <= x) ? end - base : 0u;
                    } else
#pragma unroll 1
                    for (u32 j = base; j < end; ++j) {
                        const double y = S[j];
                        less += (y < x) ? 1u : 0u;
                        le += (y <= x) ? 1u : 0u;
                    }
                    c = (base + less) | ((nv - base - le) << 16);
                }
                return c;
            };
            const bool first = jb == 0, last = jb == nblk - 1;
            for (int ib = 0; ib < nblk; ++ib) {                       // block-uniform
                const int ioff = ib * BS;
                const int isz = n - ioff < BS ? n - ioff : BS;
                // the other block's keys and the running counts first, all in flight together (one at a time they cost a
                // trip to L2 per key: 48 us per row of 20 000 instead of 30), then the look-ups, then the stores
                double xk[E];
                u32 cw[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool in = t + e * NT < isz;
                    xk[e] = ib == jb ? kk[e] : (in ? row[ioff + t + e * NT] : QNAN);
                    cw[e] = (!first && in) ? img[ioff + t + e * NT] : 0u;
                }
#pragma unroll
                for (int e = 0; e < E; ++e) cw[e] += lookup(xk[e]);
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (t + e * NT < isz) img[ioff + t + e * NT] = (last && !(xk[e] == xk[e])) ? RB_AB_SPECIAL : cw[e];
            }
            __syncthreads();                                          // barrier 6: S and the bases have been read
            {
                uint4 *Hq = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);
#pragma unroll
                for (int i = 0; i < QW; ++i) Hq[i * 64 + lane] = make_uint4(0, 0, 0, 0);
            }
        }
        if (t == 0) nnan_img[r] = nn_tot;
    }
}

// medium sizes: pair image through 2 or 3 column blocks of at most 16 384 curves per workgroup (10^7 keys, host call
// included: n = 20 000: 0.159 ms against 0.223 on the large-n route; 30 000: 0.208 / 0.228; 40 000: 0.228 / 0.239;
// 60 000 as four blocks: 0.355 / 0.243 -- not taken)
bool rank_medium_supported(i64 n) { return n > 16384 && n <= RB_MEDIUM_MAXN; }
int launch_rank_medium_image(const double *Y, i64 n, i64 row0, i64 rows, u32 *AB, u32 *nnan, hipStream_t s) {
    if (!rank_medium_supported(n)) return fail(SD_ERR_UNSUPPORTED, "medium image kernel covers 16384 < n <= %d", RB_MEDIUM_MAXN);
    const int cus = rb_cus();
    const int G = (int)(rows < cus ? rows : cus);
    const int nblk = (int)((n + 16383) / 16384);                                     // fewest blocks ...
    const int E = (int)((n + (i64)nblk * 1024 - 1) / ((i64)nblk * 1024));             // ... of equal size, whole thousands
#define RB_MD(E_, L_)                                                                                               \
    case E_: {                                                                                                       \
        using C = RBCfg<1024, E_, L_, 3>;                                                                            \
        const size_t lds = C::lds_bytes(E_ * 1024);                                                                  \
        auto kf = rank_medium_image_kernel<1024, E_, L_>;                                                            \
        SD_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipLaunchKernelGGL(kf, dim3(G), dim3(1024), lds, s, Y, n, row0, rows, nblk, AB, nnan);                       \
    } break;
    switch (E) {
        RB_MD(9, 14) RB_MD(10, 14) RB_MD(11, 14) RB_MD(12, 14) RB_MD(13, 14) RB_MD(14, 14) RB_MD(15, 14) RB_MD(16, 13)
        default: return fail(SD_ERR_UNSUPPORTED, "medium image kernel: no instantiation for n=%lld", (long long)n);
    }
#undef RB_MD
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// ---- external targets through the bucket structure ----
bool mbd_rank_external_supported(i64 T, i64 n, i64 m, int J) {
    (void)T;
    return n >= 2 && n <= 16384 && J >= 2 && J <= 3 && m >= 1;
}

size_t mbd_rank_external_workspace_bytes(i64 T, i64 n, i64 m, int J) {
    if (!mbd_rank_external_supported(T, n, m, J)) return 0;
    const i64 mc = m < 8192 ? m : 8192;
    return align_up((size_t)rb_cus() * (J - 1) * mc * 8, 256) + 512;
}

template <int E, int LNB, int J>
static int launch_external_cfg(const double *Y, i64 n, i64 rows, const double *Q, i64 m, i64 qstride, u64 *partial, int G,
                               hipStream_t s) {
    using C = RBCfg<1024, E, LNB, 3>;
    const size_t lds = C::lds_bytes((int)n);
    if (lds > 163840) return fail(SD_ERR_UNSUPPORTED, "external kernel: %zu bytes of LDS for n=%lld", lds, (long long)n);
#define RB_XL(EQ_)                                                                                                  \
    {                                                                                                                \
        auto kf = rank_external_kernel<1024, E, LNB, J, EQ_>;                                                        \
        SD_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipLaunchKernelGGL(kf, dim3(G), dim3(1024), lds, s, Y, n, rows, Q, m, qstride, partial);                              \
    }
    if (m <= 1024) RB_XL(1) else if (m <= 2048) RB_XL(2) else if (m <= 4096) RB_XL(4) else RB_XL(8)
#undef RB_XL
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// Q: T x m time-major; out[q][j] = totals.  Targets go in groups of 8 192 (8 per thread).
int launch_mbd_external_rank(const double *Y, i64 T, i64 n, const double *Q, i64 m, int J, u64 *out, void *ws,
                             size_t ws_bytes, hipStream_t s) {
    if (!mbd_rank_external_supported(T, n, m, J)) return fail(SD_ERR_UNSUPPORTED, "external rank kernel: unsupported shape");
    if (!ws || ws_bytes < mbd_rank_external_workspace_bytes(T, n, m, J))
        return fail(SD_ERR_WORKSPACE, "external rank workspace too small");
    u64 *partial = (u64 *)(((size_t)ws + 255) / 256 * 256);
    const int cus = rb_cus();
    const int G = (int)(T < cus ? T : cus);
    const int E = (int)((n + 1023) / 1024);
    for (i64 q0 = 0; q0 < m; q0 += 8192) {
        const i64 mc = m - q0 < 8192 ? m - q0 : 8192;
        int rc = SD_OK;
#define RB_XE(E_, L_) case E_: rc = (J == 2) ? launch_external_cfg<E_, L_, 2>(Y, n, T, Q + q0, mc, m, partial, G, s) \
                                             : launch_external_cfg<E_, L_, 3>(Y, n, T, Q + q0, mc, m, partial, G, s); break;
        switch (E) {
            RB_XE(1, 13) RB_XE(2, 13) RB_XE(3, 13) RB_XE(4, 13) RB_XE(5, 14) RB_XE(6, 14) RB_XE(7, 14) RB_XE(8, 14)
            RB_XE(9, 14) RB_XE(10, 14) RB_XE(11, 14) RB_XE(12, 14) RB_XE(13, 14) RB_XE(14, 14) RB_XE(15, 14) RB_XE(16, 13)
            default: return fail(SD_ERR_UNSUPPORTED, "external rank kernel covers n <= 16384");
        }
#undef RB_XE
        if (rc) return rc;
        if ((rc = launch_rank_finalize(partial, G, 0, nullptr, nullptr, nullptr, T, mc, nullptr, 0, mc, J,
                                       out + q0 * (J - 1), 1, s)))
            return rc;
    }
    return SD_OK;
}

}  // namespace sd
