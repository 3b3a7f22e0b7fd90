// mbd_rank_big.hip -- K1+K2 rank formulation for rows that do not fit one CU's LDS
// (16 384 < n < 2^31 curves; BASELINE.json config 3, the north-star stretch case, and every multi-GPU run).
//
// Same integers as the other K1+K2 kernels (reference loops _functional.py:246-251, _containment.py:75-77).
// Per row the ranks B (others strictly below) and A (strictly above) of every curve are written as a
// (uint32, uint32) pair image; rank_accumulate2_kernel folds C(v,j) - C(A,j) - C(B,j) over the rows.
//
// Route 1 (default): sample-partition into value buckets, so that no rank needs another bucket's keys.
//   S  bucket_splitters_kernel  one workgroup per row sorts a 4 096-element strided sample in LDS and
//                               publishes NB-1 splitters (NB ~ n / 5 500 buckets of capacity 8 192).
//   P  bucket_partition_kernel  every element finds its bucket (binary search over the splitters in LDS;
//                               equal values always land together), slots are handed out with one LDS
//                               atomic per element and one global atomic per (workgroup, bucket); values
//                               and curve ids are scattered into the bucket arrays.  NaNs never enter a
//                               bucket: they are counted and marked in the pair image right here.
//   A  bucket_packed_kernel     one workgroup per (row, bucket): packed-key sort (slot index in the low
//                               mantissa bits, rank_sort.h), rank = bucket base + position, handed to the
//                               slot's owner through LDS and scattered to the curve's pair.  A bucket with
//                               ties or near-ties is flagged for
//   B  bucket_search_kernel     sort of the plain values + binary search inside the bucket (exact for ties).
// Route 2 (overflow fallback, SD_BIG_IMPL=1 forces it): a row whose partition overflowed a bucket (a
//   quarter of the row tied on one value, say) is cut into chunks of 16 384 keys in curve order:
//   chunk_sort_kernel sorts every chunk, chunk_search_kernel streams every sorted chunk of the row through
//   LDS and sums lower/upper bounds per chunk: O((n/C)^2) chunk searches per row instead of none.
// The product path launches B and route 2 as ONE kernel per batch, big_fallback_kernel: flagged buckets first, then -- for
//   the rows whose partition overflowed -- chunk sort, a meeting of the (CU-resident) grid at a counter, chunk search; with
//   nothing flagged every workgroup reads two gate words and leaves.  The separate kernels stay for the cross-check
//   switches (chunked route for every row, earlier generations).
// Work per row, route 1: one streaming pass + n/5 500 LDS sorts; config 3 (10^5 x 256) on one MI355X:
// see DESIGN.md.
#include <stdlib.h>

#include <atomic>

#include "sd_common.h"
#include "rank_sort.h"
#include "rank_bucket.h"

namespace sd {

// The pair image of a batch of rows: for every (row, curve) ONE u32 -- B (others strictly below), with bit 31 set when
// the curve ties with another one at this timepoint -- and, for the tied keys only, A (strictly above) in a second
// image; an untied key has A = n_real - 1 - B.  Continuous data writes and reads 4 bytes per key instead of 8 (the
// scattered 8-byte pairs were 596 MB of the ranking kernel's traffic at config 3, profiles/r02_config3_pmc_traffic.json).
struct AB2 {
    u32 *B;
    u32 *A;
    unsigned short *H;                         // half-word image (n <= AB2_H_MAXN, fold mode) or nullptr
};
constexpr u32 AB2_NAN = 0xFFFFFFFFu;           // B word: the curve is NaN at this timepoint
constexpr u32 AB2_TIE = 0x80000000u;           // B word: A is in the second image
// The fold wants C(v, j) - C(A, j) - C(B, j), symmetric in A and B, and an untied key has A + B = nreal - 1: min(A, B) says
// it all and fits 16 bits up to 131 070 curves -- 2 bytes per (row, curve) written and read instead of 4 (config 3: 102 MB
// less traffic).  AB2_H_WORD: look at the B word (NaN, or a tied key with its A in the second image).
constexpr unsigned short AB2_H_WORD = 0xFFFFu;
constexpr i64 AB2_H_MAXN = 131070;

// an untied key: A = nreal - 1 - B
__device__ __forceinline__ void ab_store_untied(const AB2 &ab, size_t idx, u32 B, u32 nreal) {
    if (ab.H) {
        const u32 A = nreal - 1u - B;
        ab.H[idx] = (unsigned short)(A < B ? A : B);
    } else {
        ab.B[idx] = B;
    }
}
__device__ __forceinline__ void ab_store_nan(const AB2 &ab, size_t idx) {
    ab.B[idx] = AB2_NAN;
    if (ab.H) ab.H[idx] = AB2_H_WORD;
}
__device__ __forceinline__ void ab_store(const AB2 &ab, size_t idx, u32 B, u32 A, u32 nreal) {
    if (A + B + 1u == nreal) {
        ab_store_untied(ab, idx, B, nreal);
    } else {
        ab.B[idx] = B | AB2_TIE;
        ab.A[idx] = A;
        if (ab.H) ab.H[idx] = AB2_H_WORD;
    }
}

// =====================================================================================================
// route 2: chunks in curve order
// =====================================================================================================
constexpr int BIG_NT = 1024, BIG_E = 16;
constexpr int BIG_C = BIG_NT * BIG_E;          // 16384 keys per chunk
using BigCfg = R2Cfg<BIG_NT, BIG_E>;

// persistent 1-D grid over (chunk, row); rowflag != nullptr: only rows with a non-zero flag (none: every workgroup reads
// a few flags and leaves)
__device__ __forceinline__ void chunk_sort_items(const double *__restrict__ Y, i64 n, i64 row0, i64 rows, i64 nch,
                                                 double *sorted, i64 sstride, u32 *nanrow,
                                                 const u32 *__restrict__ rowflag, double *Sm) {
    constexpr int E = BIG_E, WB = BigCfg::WB;
    for (i64 v = blockIdx.x; v < rows * nch; v += gridDim.x) {
    const i64 c = v % nch, rb = v / nch;
    if (rowflag && !rowflag[rb]) continue;
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));                               // per-item opaque thread id: no address hoisted out of the loop
    const int lane = t & 63, wave = t >> 6;
    __syncthreads();                                          // the previous item's sort image is no longer in use
    const i64 base = c * BIG_C;
    const int nc = (int)((n - base) < BIG_C ? (n - base) : BIG_C);
    const int n_act = ((nc + WB - 1) / WB) * WB;
    const bool wreal = wave * WB < n_act;
    const double INF = __builtin_huge_val();
    const double *rp = Y + (row0 + rb) * n + base + (wave * WB + lane);
    double k[E];
    u32 mynan = 0;
    if (wreal) {
#pragma unroll
        for (int e = 0; e < E; ++e) k[e] = (wave * WB + lane + e * 64 < nc) ? rp[e * 64] : INF;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            bool isn = k[e] != k[e];
            mynan += isn ? 1u : 0u;
            k[e] = isn ? INF : k[e];
        }
    }
    for (int o = 32; o > 0; o >>= 1) mynan += __shfl_down(mynan, o);
    if (lane == 0 && mynan) atomicAdd(&nanrow[rb], mynan);
    R2Sorter<BIG_NT, BIG_E>::sort(k, Sm, t, n_act, wreal, INF);
    if (wreal) {
        // layout 0: thread t holds sorted positions 16 t .. 16 t + 15 (two 64-byte runs per thread)
        double *dst = sorted + rb * sstride + base + (i64)t * E;
#pragma unroll
        for (int r = 0; r < E; ++r) dst[r] = k[r];
    }
    }
}

__global__ __launch_bounds__(BIG_NT) void chunk_sort_kernel(const double *__restrict__ Y, i64 n, i64 row0, i64 rows, i64 nch,
                                                            double *__restrict__ sorted, i64 sstride,
                                                            u32 *__restrict__ nanrow, const u32 *__restrict__ rowflag,
                                                            const u32 *__restrict__ gate, u32 epoch) {
    extern __shared__ double Sm[];
    if (gate && *gate != epoch) return;                       // no row of this batch overflowed its value buckets
    chunk_sort_items(Y, n, row0, rows, nch, sorted, sstride, nanrow, rowflag, Sm);
}

// persistent 1-D grid over the items (row, query chunk): the chunk's curves are searched in every sorted chunk of the row
__device__ __forceinline__ void chunk_search_items(const double *__restrict__ Y, i64 n, i64 row0, i64 rows,
                                                   const double *sorted, i64 sstride, const u32 *nanrow, int nchunks,
                                                   const u32 *__restrict__ rowflag, const AB2 &ab, double *Sm,
                                                   i64 vfirst, i64 vstride) {
    constexpr int E = BIG_E, WB = BigCfg::WB, N = BIG_C;
    const double INF = __builtin_huge_val();
    for (i64 v = vfirst; v < rows * nchunks; v += vstride) {
        const i64 rb = v / nchunks;
        if (rowflag && !rowflag[rb]) continue;
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));                           // per-item opaque thread id
        const int qc = (int)(v % nchunks);
        const i64 qbase = (i64)qc * BIG_C;
        const int nq = (int)((n - qbase) < BIG_C ? (n - qbase) : BIG_C);
        const double *xp = Y + (row0 + rb) * n + qbase + t;
        double x[E];
        u32 lo[E], hi[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            x[e] = (t + e * BIG_NT < nq) ? xp[e * BIG_NT] : INF;
            lo[e] = 0;
            hi[e] = 0;
        }
        for (int c = 0; c < nchunks; ++c) {
            const i64 base = (i64)c * BIG_C;
            const int nc = (int)((n - base) < BIG_C ? (n - base) : BIG_C);
            const int n_act = ((nc + WB - 1) / WB) * WB;
            const double *src = sorted + rb * sstride + base;
            __syncthreads();                          // previous chunk's searches are done
            for (int p = t; p < n_act; p += BIG_NT) Sm[r2_swz(p)] = src[p];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if ((e & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                if (t + e * BIG_NT < nq && x[e] == x[e]) {
                    int l = r2_bound<N, SlotSwz, false>(Sm, n_act, x[e], INF);
                    int h = l;
                    // keys equal to x in this chunk?  (always true once: in x's own chunk)
                    double nx = (l < n_act) ? Sm[r2_swz(l)] : INF;
                    if (l < n_act && nx <= x[e]) h = r2_bound<N, SlotSwz, true>(Sm, n_act, x[e], INF);
                    lo[e] += (u32)l;
                    hi[e] += (u32)h;
                }
            }
        }
        const u32 nnan = nanrow[rb];
        const size_t dst = (size_t)(rb * n + qbase + t);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (t + e * BIG_NT < nq) {
                if (x[e] == x[e])
                    ab_store(ab, dst + e * BIG_NT, lo[e], (x[e] == INF) ? 0u : (u32)(n - hi[e]) - nnan, (u32)n - nnan);
                else
                    ab_store_nan(ab, (size_t)(dst + e * BIG_NT));
            }
        }
    }
}

__global__ __launch_bounds__(BIG_NT) void chunk_search_kernel(const double *__restrict__ Y, i64 n, i64 row0,
                                                              i64 rows, const double *__restrict__ sorted,
                                                              i64 sstride, const u32 *__restrict__ nanrow,
                                                              int nchunks, const u32 *__restrict__ rowflag,
                                                              const u32 *__restrict__ gate, u32 epoch, AB2 ab) {
    extern __shared__ double Sm[];
    if (gate && *gate != epoch) return;                       // no row of this batch overflowed its value buckets
    chunk_search_items(Y, n, row0, rows, sorted, sstride, nanrow, nchunks, rowflag, ab, Sm, blockIdx.x, gridDim.x);
}

// =====================================================================================================
// route 1: value buckets
// =====================================================================================================
#ifndef SD_BK_CE
#define SD_BK_CE 16                            // keys per thread of the ranking kernels = bucket capacity / 512
#endif
constexpr int BK_NT = 512, BK_E = 16;          // the search kernel's sort: 8 192 slots
constexpr int BK_C = BK_NT * SD_BK_CE;         // bucket capacity (8 192)
static_assert(BK_C <= BK_NT * BK_E, "a value bucket fits the search kernel's sort");
#ifndef SD_BK_FILL
#define SD_BK_FILL 5500
#endif
constexpr int BK_FILL = SD_BK_FILL;            // target mean fill
constexpr int BK_MAXNB = 1024;
using BkCfg = R2Cfg<BK_NT, BK_E>;

static inline int bucket_count(i64 n) {
    // mean fill 5 500 of 8 192; past ~300 buckets even the 16 384-value sample leaves < 50 samples per bucket and the
    // fills scatter too much: aim lower
    const i64 fill = n > 1600000 ? (i64)BK_FILL * 3800 / 5500 : BK_FILL;
    i64 nb = (n + fill - 1) / fill;
    if (nb < 2) nb = 2;
    return (int)nb;
}

// S: grid = rows; spl[r][0..NB-2] ascending.  SNT threads sort a strided sample of SE SNT values: 2 048 for up to 24
// value buckets, 4 096 up to 72, 16 384 above (a bucket's fill scatters with 1 / sqrt(samples per bucket); at n = 10^6 the small
// sample overflowed the 8 192-key buckets and sent every row to the chunked route).
#ifdef SD_CROSSCHECK
template <int SNT, int SE>
__global__ __launch_bounds__(SNT) void bucket_splitters_kernel(const double *__restrict__ Y, i64 n, i64 row0, int NB,
                                                               double *__restrict__ spl) {
    using Cfg = R2Cfg<SNT, SE>;
    constexpr int E = SE, LE = Cfg::LE, SS = SNT * SE;
    extern __shared__ double Sm[];
    const int t = threadIdx.x;
    const i64 rb = blockIdx.x;
    const double *row = Y + (row0 + rb) * n;
    const double INF = __builtin_huge_val();
    double k[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const i64 s = (i64)t * E + e;                       // sample index, any assignment of samples to slots works
        double v = row[(s * n) / SS];
        k[e] = (v == v) ? v : INF;
    }
    R2Sorter<SNT, SE>::sort(k, Sm, t, SS, true, INF);
    double *Sw = Sm + r2_base<0, LE>(t);
#pragma unroll
    for (int e = 0; e < E; ++e) Sw[r2_off<0, LE>(e)] = k[e];
    __syncthreads();
    for (int b = t; b < NB - 1; b += SNT) {
        const int q = (int)(((i64)(b + 1) * SS) / NB);
        spl[rb * (NB - 1) + b] = Sm[r2_phys<LE>(q)];
    }
}

// P: grid = (ceil(n / 16384), rows)
__global__ __launch_bounds__(1024) void bucket_partition_kernel(const double *__restrict__ Y, i64 n, i64 row0, int NB,
                                                                const double *__restrict__ spl,
                                                                u32 *__restrict__ bcnt, u32 *__restrict__ nnanrow,
                                                                u32 *__restrict__ ovf, double *__restrict__ bval,
                                                                u32 *__restrict__ bidx, AB2 ab, int dbg) {
    __shared__ double s_spl[BK_MAXNB];
    __shared__ u32 s_hist[BK_MAXNB];
    __shared__ u32 s_base[BK_MAXNB];
    const int t = threadIdx.x;
    const i64 rb = blockIdx.y;
    const i64 base = (i64)blockIdx.x * 16384;
    for (int b = t; b < NB; b += 1024) {
        if (b < NB - 1) s_spl[b] = spl[rb * (NB - 1) + b];
        s_hist[b] = 0;
    }
    __syncthreads();
    const double *row = Y + (row0 + rb) * n;
    double x[16];
    u32 bk[16], off[16];
    u32 mynan = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const i64 i = base + t + e * 1024;
        x[e] = (i < n) ? row[i] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const i64 i = base + t + e * 1024;
        bk[e] = 0xFFFFFFFFu;
        if (i < n) {
            if (x[e] == x[e]) {
                // bucket = number of splitters < x (equal values always share a bucket)
                int lo = 0, hi = NB - 1;
                if (dbg) lo = hi = (int)((u32)(e + t) % (u32)NB);     // timing experiment: no search (results invalid)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_spl[mid] < x[e]) lo = mid + 1;
                    else hi = mid;
                }
                bk[e] = (u32)lo;
                off[e] = atomicAdd(&s_hist[lo], 1u);
            } else {
                ++mynan;
                ab_store_nan(ab, (size_t)(rb * n + i));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) mynan += __shfl_down(mynan, o);
    if ((t & 63) == 0 && mynan) atomicAdd(&nnanrow[rb], mynan);
    __syncthreads();
    for (int b = t; b < NB; b += 1024) s_base[b] = s_hist[b] ? atomicAdd(&bcnt[rb * NB + b], s_hist[b]) : 0u;
    __syncthreads();
    bool over = false;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (bk[e] != 0xFFFFFFFFu) {
            const u32 pos = s_base[bk[e]] + off[e];
            if (pos < (u32)BK_C) {
                const size_t slot = ((size_t)rb * NB + bk[e]) * BK_C + pos;
                bval[slot] = x[e];
                bidx[slot] = (u32)(base + t + e * 1024);
            } else {
                over = true;
            }
        }
    }
    if (over) ovf[rb] = 1u;
}

// P' (second generation): the same partition with the scatter staged through LDS.  One workgroup takes 8 192 consecutive curves
// of one row, orders them by value bucket inside LDS (local slot = LDS-atomic offset + local exclusive prefix of the
// workgroup's bucket counts) and copies the ordered block out: consecutive threads write consecutive elements of a
// bucket's run, so the 12-byte records leave as coalesced stores instead of ~19 interleaved partial runs per wave
// instruction.  grid = (ceil(n / 8192), rows).
constexpr int BP2_NT = 1024, BP2_E = 8, BP2_C = BP2_NT * BP2_E;
__global__ __launch_bounds__(BP2_NT) void bucket_partition2_kernel(const double *__restrict__ Y, i64 n, i64 row0, int NB,
                                                                   const double *__restrict__ spl,
                                                                   u32 *__restrict__ bcnt, u32 *__restrict__ nnanrow,
                                                                   u32 *__restrict__ ovf, double *__restrict__ bval,
                                                                   u32 *__restrict__ bidx, AB2 ab) {
    extern __shared__ double Sm2[];
    double *Skey = Sm2;                                               // [BP2_C]
    u32 *Sid = reinterpret_cast<u32 *>(Skey + BP2_C);                 // [BP2_C]
    unsigned short *Sbk = reinterpret_cast<unsigned short *>(Sid + BP2_C);   // [BP2_C]
    __shared__ double s_spl[BK_MAXNB];
    __shared__ u32 s_hist[BK_MAXNB];
    __shared__ u32 s_gbase[BK_MAXNB];
    __shared__ u32 s_lbase[BK_MAXNB + 1];
    __shared__ u32 s_wtot[BP2_NT / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const i64 rb = blockIdx.y;
    const i64 base = (i64)blockIdx.x * BP2_C;
    for (int b = t; b < BK_MAXNB; b += BP2_NT) {
        if (b < NB - 1) s_spl[b] = spl[rb * (NB - 1) + b];
        s_hist[b] = 0;
    }
    __syncthreads();
    const double *row = Y + (row0 + rb) * n;
    double x[BP2_E];
    u32 bk[BP2_E], off[BP2_E];
    u32 mynan = 0;
#pragma unroll
    for (int e = 0; e < BP2_E; ++e) {
        const i64 i = base + t + e * BP2_NT;
        x[e] = (i < n) ? row[i] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < BP2_E; ++e) {
        const i64 i = base + t + e * BP2_NT;
        bk[e] = 0xFFFFFFFFu;
        off[e] = 0;
        if (i < n) {
            if (x[e] == x[e]) {
                int lo = 0, hi = NB - 1;                    // bucket = number of splitters < x (equal values share a bucket)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_spl[mid] < x[e]) lo = mid + 1;
                    else hi = mid;
                }
                bk[e] = (u32)lo;
                off[e] = atomicAdd(&s_hist[lo], 1u);
            } else {
                ++mynan;
                ab_store_nan(ab, (size_t)(rb * n + i));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) mynan += __shfl_down(mynan, o);
    if (lane == 0 && mynan) atomicAdd(&nnanrow[rb], mynan);
    __syncthreads();
    // global base of this workgroup's run in every bucket; local exclusive prefix of the counts (one thread per bucket)
    {
        const u32 c = (t < NB) ? s_hist[t] : 0u;
        if (t < NB) s_gbase[t] = c ? atomicAdd(&bcnt[rb * NB + t], c) : 0u;
        const u32 incl = rb_wave_incl_scan(c);
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        const u32 wt = (lane < BP2_NT / 64) ? s_wtot[lane] : 0u;
        const u32 wscan = rb_row_incl_scan(wt);
        const u32 woff = wave ? rb_readlane(wscan, wave - 1) : 0u;
        if (t < NB) s_lbase[t] = woff + incl - c;
        if (t == BP2_NT - 1) s_lbase[NB] = woff + incl;             // number of non-NaN keys of the block (NB <= 1024)
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < BP2_E; ++e) {
        if (bk[e] != 0xFFFFFFFFu) {
            const u32 lp = s_lbase[bk[e]] + off[e];
            Skey[lp] = x[e];
            Sid[lp] = (u32)(base + t + e * BP2_NT);
            Sbk[lp] = (unsigned short)bk[e];
        }
    }
    __syncthreads();
    const u32 nval = s_lbase[NB];
    bool over = false;
    for (u32 p = t; p < nval; p += BP2_NT) {
        const u32 b = Sbk[p];
        const u32 g = s_gbase[b] + (p - s_lbase[b]);
        if (g < (u32)BK_C) {
            const size_t slot = ((size_t)rb * NB + b) * BK_C + g;
            bval[slot] = Skey[p];
            bidx[slot] = Sid[p];
        } else {
            over = true;
        }
    }
    if (over) ovf[rb] = 1u;
}
#endif  // SD_CROSSCHECK

template <int NT, int E>
struct BkKeys {
    using C = R2Cfg<NT, E>;
    static constexpr int LN = C::LN;
    static constexpr u64 MASK = (u64)C::N - 1;
    static constexpr u64 TOPM = ((0xFFFFFFFFFFFFFull >> LN) << LN);
    static constexpr u64 H3 = (0x7FEull << 52) | TOPM;         // padding class (largest)
    static constexpr u64 H1 = H3 - ((u64)2 << LN);             // +inf class (H3 - 1 class stays unused here: no NaN)
    static constexpr u64 SIGN = 0x8000000000000000ull;
    static constexpr u64 LOW = (u64)1 << LN;
};

__device__ __forceinline__ u64 bk_bits(double v) { return (u64)__double_as_longlong(v); }
__device__ __forceinline__ double bk_dbl(u64 b) { return __longlong_as_double((long long)b); }

// A: grid = (NB, rows)
#ifdef SD_CROSSCHECK
__global__ __launch_bounds__(BK_NT) void bucket_packed_kernel(i64 n, int NB, const u32 *__restrict__ bcnt,
                                                              const u32 *__restrict__ nnanrow,
                                                              const u32 *__restrict__ ovf,
                                                              const double *__restrict__ bval,
                                                              const u32 *__restrict__ bidx, u32 *__restrict__ bflag,
                                                              AB2 ab) {
    using C = BkCfg;
    using K = BkKeys<BK_NT, BK_E>;
    constexpr int E = BK_E, NT = BK_NT, LN = C::LN, WB = C::WB;
    constexpr u64 MASK = K::MASK, CLS_PAD = K::H3 >> LN;
    extern __shared__ double Sm[];
    double *firstkey = Sm + C::SLOTS;
    __shared__ u32 s_basecnt;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.x;
    const i64 rb = blockIdx.y;
    if (ovf[rb]) return;
    const int cnt = (int)bcnt[rb * NB + b];
    if (cnt == 0) return;
    if (t == 0) {
        u32 s = 0;
        for (int q = 0; q < b; ++q) s += bcnt[rb * NB + q];
        s_basecnt = s;
    }
    const int n_act = ((cnt + WB - 1) / WB) * WB;
    const bool wreal = wave * WB < n_act;
    const double INF = __builtin_huge_val();
    const double MAXK = bk_dbl(K::H3 | MASK);
    const size_t slot0 = ((size_t)rb * NB + b) * BK_C;
    const int i0 = wave * WB + lane;
    double k[E];
    int forcefull = 0;
    if (wreal) {
        const double *rp = bval + slot0 + i0;
#pragma unroll
        for (int e = 0; e < E; ++e) k[e] = (i0 + e * 64 < cnt) ? rp[e * 64] : INF;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = i0 + e * 64;
            const u64 bits = bk_bits(k[e]);
            const u64 a = bits & ~K::SIGN;
            u64 kb = bits & ~MASK;
            if (__builtin_expect((a - K::LOW) >= (K::H1 - K::LOW), 0)) {
                if (a == 0x7FF0000000000000ull) kb = (bits & K::SIGN) ? (K::SIGN | K::H3) : K::H1;
                else if (a == 0) kb = 0;
                else forcefull |= (j < cnt);
            }
            kb = (j < cnt) ? kb : K::H3;
            k[e] = bk_dbl(kb | (u64)j);
        }
    }
    R2Sorter<NT, E>::sort(k, Sm, t, n_act, wreal, MAXK);
    if (wreal) firstkey[t] = k[0];
    __syncthreads();
    int anytie = 0;
    if (wreal) {
        u64 nextb = ~0ull;
        if ((t + 1) * E < n_act) nextb = bk_bits(firstkey[t + 1]);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u64 c0 = bk_bits(k[e]) >> LN;
            const u64 c1 = ((e < E - 1) ? bk_bits(k[e + 1]) : nextb) >> LN;
            anytie |= (c0 == c1) & (c0 != CLS_PAD);
        }
    }
    if (__syncthreads_or(anytie | forcefull)) {
        if (t == 0) bflag[rb * NB + b] = 1u;
        return;
    }
    u32 *R = reinterpret_cast<u32 *>(Sm);
    if (wreal) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = (int)(bk_bits(k[e]) & MASK);
            if (j < cnt) R[j] = (u32)(t * E + e);
        }
    }
    __syncthreads();
    const u32 base = s_basecnt;
    for (int j = t; j < cnt; j += NT) {
        ab.B[rb * n + bidx[slot0 + j]] = base + R[j];               // distinct keys: A = (n - NaNs of the row) - 1 - B
    }
}
#endif  // SD_CROSSCHECK


// A' (default): grid = 8 * NB * ceil(rows / 8) (see the mapping below).  Ranks inside one value bucket WITHOUT a sort, the method of mbd_rank_bucket.hip:
// a monotone map of the bucket's keys onto NBF fine buckets (LDS histogram, the atomic's return value is the slot),
// exclusive prefix sum, scatter into fine-bucket order, and every key counts the members of its own fine bucket that
// are < / <= itself: B = (keys in earlier value buckets) + base + less, A = n_real - (... + base + le).  Ties are exact.
// A value bucket whose keys are all equal is closed-form; one with a fine bucket above BR_CAP keys (heavy ties that
// are not all equal, an infinity stretching the range) is flagged for bucket_search_kernel like before.
#ifndef SD_BR_U2
#define SD_BR_U2 3
#endif
constexpr int BR_NT = 512, BR_E = SD_BK_CE, BR_LNB = 12, BR_NBF = 1 << BR_LNB, BR_CAP = 63, BR_TRYB = 4, BR_U2 = SD_BR_U2, BR_PAD = 8;
static_assert(((BR_CAP + 1) & BR_CAP) == 0, "the crowding test reads the counters' bits");
constexpr int BR_NW = BR_NT / 64;
static_assert(BR_NT * BR_E == BK_C, "one thread slot per key of a full value bucket");
static_assert(BR_NBF / 2 / BR_NT == 4, "one 16-byte quad of histogram words per thread");
constexpr size_t BR_HDR = 256;                                         // min/max partials [NW][2] doubles, wave totals [NW]
constexpr size_t BR_LDS = BR_HDR + (size_t)(BR_NBF / 2 + 4) * 4 + (size_t)(BK_C + BR_PAD + 2 * BR_U2 + 4) * 8;

__device__ __forceinline__ void bucket_rank_item(const int w, i64 n, i64 rows, int NB, const u32 *__restrict__ bcnt,
                                                 const u32 *__restrict__ nnanrow, const u32 *__restrict__ ovf,
                                                 const u32 *__restrict__ rowtied, const double *__restrict__ bval,
                                                 const u32 *__restrict__ bidx, u32 *__restrict__ bflag,
                                                 u32 *__restrict__ gate, u32 epoch, AB2 ab) {
    constexpr int E = BR_E, NT = BR_NT, NBF = BR_NBF, NW = BR_NW, U2 = BR_U2;
    extern __shared__ double Sm[];
    double *red = Sm;                                                 // [NW][2]
    u32 *wtot = reinterpret_cast<u32 *>(red + 2 * NW);                // [NW], then the sum of the earlier buckets' counts
    u32 *H = reinterpret_cast<u32 *>(Sm + BR_HDR / 8);                // NBF packed u16 counters, then bases
    double *S = reinterpret_cast<double *>(H + NBF / 2 + 4);          // keys in fine-bucket order + NaN sentinels
    const unsigned short *H16 = reinterpret_cast<const unsigned short *>(H);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // XCD-aware mapping (workgroups go to the 8 XCDs round-robin): all value buckets of a row run on ONE XCD, close in
    // time, so the 8-byte pair writes they scatter over that row of the image meet in that XCD's L2 and leave it as
    // whole lines.  grid.x = 8 * NB * ceil(rows / 8).
    const int b = (w >> 3) % NB;
    const i64 rb = (i64)((w >> 3) / NB) * 8 + (w & 7);
    if (rb >= rows) return;
    if (ovf[rb]) return;
    if (rowtied && !rowtied[rb]) return;                              // third generation: only the rows flagged "tied"
    const u32 *rowcnt = bcnt + rb * NB;
    const int cnt = (int)rowcnt[b];
    if (cnt == 0) return;
    const double INF = __builtin_huge_val();
    const double QNAN = __builtin_nan("");
    const size_t slot0 = ((size_t)rb * NB + b) * BK_C;

    // keys of thread t: slots t, t + NT, ... (coalesced); slots beyond cnt read as NaN = "no key"
    double k[E];
    {
        const double *rp = bval + slot0 + t;
#pragma unroll
        for (int e = 0; e < E; ++e) k[e] = (t + e * NT < cnt) ? rp[e * NT] : QNAN;
    }
    // keys in earlier value buckets of this row
    u32 gsum = 0;
    for (int q = t; q < b; q += NT) gsum += rowcnt[q];
    gsum = rb_wave_incl_scan(gsum);
    // LDS setup: empty histogram, sentinels behind the last key
    reinterpret_cast<uint4 *>(H)[t] = make_uint4(0, 0, 0, 0);
    if (t < 4) H[NBF / 2 + t] = 0;
    if (t < BR_PAD + 2 * U2 + 4) S[cnt + t] = QNAN;
    // range
    double mn = INF, mx = -INF;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e * NT >= cnt) break;                                     // block-uniform: key slots beyond the bucket's fill
        mn = rb_mm<false>(mn, k[e]);
        mx = rb_mm<true>(mx, k[e]);
    }
    mn = rb_wave_allreduce<false>(mn);
    mx = rb_wave_allreduce<true>(mx);
    if (lane == 63) { red[2 * wave] = mn; red[2 * wave + 1] = mx; wtot[wave] = gsum; }
    __syncthreads();                                                  // barrier 1
    double lo, hi;
    u32 gbase;
    {
        const double2 p = reinterpret_cast<const double2 *>(red)[lane & (NW - 1)];
        lo = rb_readlane_f64(rb_row_allreduce<false>(p.x), 0);        // rotations over 16 lanes see each of the 8 twice
        hi = rb_readlane_f64(rb_row_allreduce<true>(p.y), 0);
        const u32 g = (lane < NW) ? wtot[lane] : 0u;
        gbase = rb_readlane(rb_row_incl_scan(g), 15);
    }
    const u32 nreal = (u32)n - nnanrow[rb];
    const size_t abrow = (size_t)(rb * n);
    const u32 *idp = bidx + slot0 + t;
    if (!(hi > lo)) {
        // every key of the bucket has the same value (or there is one key): all tied
        if (hi == lo) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (t + e * NT < cnt) ab_store(ab, abrow + idp[e * NT], gbase, nreal - gbase - (u32)cnt, nreal);
        } else if (t == 0) { bflag[rb * NB + b] = 1u; if (gate) gate[0] = epoch; }                   // a signalling NaN poisoned the range: sort it
        return;
    }
    const double scale = (double)NBF / (hi - lo);                     // infinite range -> 0 -> one crowded fine bucket
    if (!(scale < INF)) {                                             // block-uniform: denormal range
        if (t == 0) { bflag[rb * NB + b] = 1u; if (gate) gate[0] = epoch; }
        return;
    }
    // ---- (1) fine bucket + slot ----
    u32 bs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e * NT >= cnt) break;
        const double x = k[e];
        double u = (x - lo) * scale;
        u = u > 0.0 ? u : 0.0;                                        // -inf (and NaN) -> 0
        u = u < (double)(NBF - 1) ? u : (double)(NBF - 1);
        u32 fb = (u32)u;
        fb = (x == x) ? fb : (u32)(NBF + 2);                          // no key: dummy counter
        const u32 sh = (fb & 1u) * 16u;
        const u32 old = atomicAdd(&H[fb >> 1], 1u << sh);
        bs[e] = fb | (((old >> sh) & 0xFFFFu) << 16);
    }
    __syncthreads();                                                  // barrier 2
    // ---- (2) exclusive prefix sum; a fine bucket of 2^BR_TRYB keys or more: ties?  (checked behind the scatter) ----
    bool anyover = false, anytry = false;
    {
        const uint4 hq = reinterpret_cast<const uint4 *>(H)[t];
        // some counter > BR_CAP / >= 2^BR_TRYB: bit k of a half-word of the OR is set iff some counter has it
        constexpr u32 HIM = (0xFFFFu & ~(u32)BR_CAP) * 0x10001u, TRM = (0xFFFFu & ~((1u << BR_TRYB) - 1u)) * 0x10001u;
        const u32 s4 = hq.x + hq.y + hq.z + hq.w;
        const u32 ov = hq.x | hq.y | hq.z | hq.w;
        const u32 run = (s4 & 0xFFFFu) + (s4 >> 16);
        const u32 incl = rb_wave_incl_scan(run);
        const bool wover = __ballot((ov & HIM) != 0) != 0, wtry = __ballot((ov & TRM) != 0) != 0;
        if (lane == 63) wtot[wave] = incl | (wover ? 0x80000000u : 0u) | (wtry ? 0x40000000u : 0u);
        __syncthreads();                                              // barrier 3
        const u32 wt = (lane < NW) ? wtot[lane] : 0u;
        anyover = __ballot((wt >> 31) != 0) != 0;                     // block-uniform
        anytry = __ballot((wt & 0x40000000u) != 0) != 0;
        const u32 wscan = rb_row_incl_scan(wt & 0x3FFFFFFFu);
        u32 base = (wave ? rb_readlane(wscan, wave - 1) : 0u) + incl - run;
        uint4 o;
        o.x = base | ((base + (hq.x & 0xFFFFu)) << 16);
        base += (hq.x & 0xFFFFu) + (hq.x >> 16);
        o.y = base | ((base + (hq.y & 0xFFFFu)) << 16);
        base += (hq.y & 0xFFFFu) + (hq.y >> 16);
        o.z = base | ((base + (hq.z & 0xFFFFu)) << 16);
        base += (hq.z & 0xFFFFu) + (hq.z >> 16);
        o.w = base | ((base + (hq.w & 0xFFFFu)) << 16);
        base += (hq.w & 0xFFFFu) + (hq.w >> 16);
        reinterpret_cast<uint4 *>(H)[t] = o;
        if (t == NT - 1) H[NBF / 2] = base;                           // = cnt
    }
    __syncthreads();                                                  // barrier 4
    // ---- (3) scatter into fine-bucket order ----
    u32 bc[E];                                                        // base | count << 16; count 0: no key
    const u32 dummy = (u32)(cnt + BR_PAD + 1) & ~1u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e * NT >= cnt) break;
        const u32 fb = bs[e] & 0xFFFFu, slot = bs[e] >> 16;
        const u32 base = H16[fb], end = H16[fb + 1];
        const bool isk = fb < (u32)NBF;
        S[isk ? base + slot : dummy] = k[e];
        bc[e] = isk ? (base | ((end - base) << 16)) : 0u;
    }
    __syncthreads();                                                  // barrier 5
    if (anytry) {                                                     // block-uniform
        // tie-heavy data: when every fine bucket holds ONE value, less = 0 and le = count -- no member pass.
        // Else: the normal way, or the search kernel when a fine bucket is above BR_CAP keys.
        bool pure = true;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (e * NT >= cnt) break;
            if (bc[e] >> 16) pure = pure && (S[bc[e] & 0xFFFFu] == k[e]);
        }
        if (__syncthreads_and(pure)) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (e * NT >= cnt) break;
                const u32 base = bc[e] & 0xFFFFu, fc = bc[e] >> 16;
                if (fc) ab_store(ab, abrow + idp[e * NT], gbase + base, nreal - (gbase + base + fc), nreal);
            }
            return;
        }
        if (anyover) {
            if (t == 0) { bflag[rb * NB + b] = 1u; if (gate) gate[0] = epoch; }
            return;
        }
    }
    // ---- (4) rank inside the fine bucket (the keys are still in registers), write the pairs ----
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e * NT >= cnt) break;
        if ((e & 1) == 0) __builtin_amdgcn_sched_barrier(0);
        const u32 base = bc[e] & 0xFFFFu, fc = bc[e] >> 16;
        const u32 odd = base & 1u;
        const double x = k[e];
        const double2 *Sq = reinterpret_cast<const double2 *>(S + (base - odd));
        u32 less = 0, le = 0;
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const double2 y = Sq[u];
            less += (y.x < x) ? 1u : 0u;
            le += (y.x <= x) ? 1u : 0u;
            less += (y.y < x) ? 1u : 0u;
            le += (y.y <= x) ? 1u : 0u;
        }
        less -= odd;
        le -= odd;
        if (fc + odd > (u32)(2 * U2)) {                               // a fine bucket longer than the window
            for (u32 kk = 2 * U2; kk < fc + odd; kk += 2) {
                const double2 y = Sq[kk >> 1];
                less += (y.x < x) ? 1u : 0u;
                le += (y.x <= x) ? 1u : 0u;
                less += (y.y < x) ? 1u : 0u;
                le += (y.y <= x) ? 1u : 0u;
            }
        }
        if (fc) ab_store(ab, abrow + idp[e * NT], gbase + base + less, nreal - (gbase + base + le), nreal);
    }
}


#ifdef SD_CROSSCHECK
// A' as a kernel of its own (second generation, cross-check builds): grid = 8 * NB * ceil(rows / 8), every row
__global__ __launch_bounds__(BR_NT) void bucket_rank_kernel(i64 n, i64 rows, int NB, const u32 *__restrict__ bcnt,
                                                            const u32 *__restrict__ nnanrow,
                                                            const u32 *__restrict__ ovf,
                                                            const double *__restrict__ bval,
                                                            const u32 *__restrict__ bidx, u32 *__restrict__ bflag,
                                                            AB2 ab) {
    bucket_rank_item(blockIdx.x, n, rows, NB, bcnt, nnanrow, ovf, nullptr, bval, bidx, bflag, nullptr, 0u, ab);
}
#endif

// =====================================================================================================
// route 1, third generation (round 3): 8-byte records, table-driven partition, 32-bit ranking
// =====================================================================================================
// What changed against S / P' / A' above, and why (profiles/r02b_config3_*): P' spent 91 VALU + 46 SALU per key on a
// binary search over the splitters and wrote 12-byte records; A' compared fp64 keys (half rate on MI355X) out of 8-byte
// LDS slots with 63 % of its LDS cycles lost to bank conflicts; three fallback launches of full grids did nothing.
//   S3 bucket_setup_kernel      the sample sort of S, then per row: NB + 1 splitters (the sample's extremes are splitters
//                               too: the keys beyond them form two small end buckets), a LOOK-UP TABLE over TB_C cells of
//                               the monotone map c(x) = trunc((x - lo) * scale) holding, per cell, the number of splitters
//                               in earlier cells and up to this cell; a "tied" flag when the sorted sample has two equal
//                               neighbours; and the zeroing of the row's counters (no memset launch).
//   P3 bucket_partition3_kernel bucket(x) = number of splitters < x as before, but from the table: a compare is needed
//                               only in cells that a splitter cuts.  Records are 8 bytes: the curve index and a 32-bit
//                               IMAGE q of the key, monotone inside its bucket (interior buckets: linear between the two
//                               splitters; end buckets: a float-like code of the distance to the splitter in representable
//                               doubles) -- equal keys have equal images, different keys almost always different ones.
//                               Rows flagged "tied" keep the 12-byte fp64 records (their images would collide en masse).
//   A3 bucket_rank32_kernel     the bucket ranking of A' on the 32-bit images: 4-byte LDS slots, integer compares, no
//                               range reduction in fp64.  A key whose image equals another member's settles the order of
//                               those members with their fp64 values (gathered from the matrix through the records' curve
//                               indices): exact whatever the data, cheap because it is rare on continuous data.
//                               Tied rows go through bucket_rank_kernel (fp64) as before.
constexpr int TB_C = 2048;                                             // cells of the partition's look-up table
#ifndef SD_S3_RUN
#define SD_S3_RUN 4
#endif
constexpr int S3_RUN = SD_S3_RUN;                                      // neighbours per sample position (bucket_setup_kernel)
constexpr u32 Q_MAX = 0xFFFFFFFEu;                                     // largest image (0xFFFFFFFF = "no key" in LDS)

__device__ __forceinline__ u32 tb_cell(double x, double lo, double scale) {
    double u = (x - lo) * scale;                                       // monotone in x for any lo and any scale >= 0
    u = u > 0.0 ? u : 0.0;                                             // NaN (inf * 0) -> 0
    u = u < (double)(TB_C - 1) ? u : (double)(TB_C - 1);
    return (u32)u;
}
// order-preserving 64-bit pattern of a double (zeros canonicalised by the caller: x + 0.0)
__device__ __forceinline__ u64 q_ord(double x) {
    const u64 b = (u64)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// monotone 32-bit code of a 64-bit distance: exact below 2^27, then 26 mantissa bits under a 6-bit length field
__device__ __forceinline__ u32 q_code(u64 d) {
    if (d < (1ull << 27)) return (u32)d;
    const int e = 64 - __builtin_clzll(d);                              // 28 .. 64
    const u32 mant = (u32)(d >> (e - 27));                              // [2^26, 2^27)
    return ((u32)(e - 26) << 26) + (mant - (1u << 26));
}

template <bool MAX>
__device__ __forceinline__ u32 rb_mmu(u32 a, u32 b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); }
template <bool MAX>
__device__ __forceinline__ u32 rb_wave_allreduce_u32(u32 v) {
    v = rb_mmu<MAX>(v, (u32)__builtin_amdgcn_mov_dpp((int)v, 0x121, 0xF, 0xF, false));
    v = rb_mmu<MAX>(v, (u32)__builtin_amdgcn_mov_dpp((int)v, 0x122, 0xF, 0xF, false));
    v = rb_mmu<MAX>(v, (u32)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xF, 0xF, false));
    v = rb_mmu<MAX>(v, (u32)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, false));
    u32 r = rb_readlane(v, 0);
    r = rb_mmu<MAX>(r, rb_readlane(v, 16));
    r = rb_mmu<MAX>(r, rb_readlane(v, 32));
    r = rb_mmu<MAX>(r, rb_readlane(v, 48));
    return r;
}

// S3: grid = rows.  NB = interior buckets; NBT = NB + 2 buckets and NS = NB + 1 splitters per row.
constexpr int FB_MEET_WORDS_S3 = 2 + 1024;                             // = FB_MEET_WORDS (checked where that is defined)
template <int SNT, int SE>
__global__ __launch_bounds__(SNT) void bucket_setup_kernel(const double *__restrict__ Y, i64 n, i64 row0, int NB,
                                                           double *__restrict__ spl, double *__restrict__ mk,
                                                           u32 *__restrict__ tab, double2 *__restrict__ rp,
                                                           u32 *__restrict__ rowtied, u32 *__restrict__ bcnt,
                                                           u32 *__restrict__ bflag, u32 *__restrict__ nnanrow,
                                                           u32 *__restrict__ ovf, u32 *__restrict__ nanf,
                                                           u32 *__restrict__ meet) {
    using Cfg = R2Cfg<SNT, SE>;
    constexpr int E = SE, LE = Cfg::LE, SS = SNT * SE, NWV = SNT / 64, CPT = TB_C / SNT;
    static_assert(TB_C % SNT == 0, "whole cells per thread");
    extern __shared__ double Sm[];
    __shared__ u32 s_cnt[TB_C];
    __shared__ u32 s_w[16];
    __shared__ u32 s_inf[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const i64 rb = blockIdx.x;
    const int NBT = NB + 2, NS = NB + 1;
    const double *row = Y + (row0 + rb) * n;
    const double INF = __builtin_huge_val();
    double k[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const i64 sidx = (i64)t * E + e;
        // the sample in runs of S3_RUN neighbours (32 bytes: one memory request fetches four keys, S3 22.3 -> 15.1 us and
        // 33 -> 9 MB at config 3).  Runs of 8 / 16 gain 1 / 2 us more and leave fewer independent positions when
        // neighbouring curves resemble each other.
        i64 sp = ((sidx / S3_RUN) * n) / (SS / S3_RUN) + (sidx % S3_RUN);
        sp = sp < n ? sp : n - 1;
        double v = row[sp];
        k[e] = (v == v) ? v : INF;
    }
    for (int c = t; c < TB_C; c += SNT) s_cnt[c] = 0;
    if (t < 2) s_inf[t] = 0;
    // the row's counters (the partition adds to them): zeroed here, no memset launch
    for (int b = t; b < NBT; b += SNT) {
        bcnt[rb * NBT + b] = 0;
        bflag[rb * NBT + b] = 0;
    }
    if (t == 0) { nnanrow[rb] = 0; ovf[rb] = 0; nanf[rb] = 0; }
    if (rb == 0)                                                       // the batch's meeting words (big_fallback_kernel)
        for (int c = t; c < FB_MEET_WORDS_S3; c += SNT) meet[c] = 0;
    R2Sorter<SNT, SE>::sort(k, Sm, t, SS, true, INF);
    double *Sw = Sm + r2_base<0, LE>(t);
#pragma unroll
    for (int e = 0; e < E; ++e) Sw[r2_off<0, LE>(e)] = k[e];
    u32 cpos = 0, cneg = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        cpos += (k[e] == INF) ? 1u : 0u;
        cneg += (k[e] == -INF) ? 1u : 0u;
    }
    if (cpos) atomicAdd(&s_inf[0], cpos);
    if (cneg) atomicAdd(&s_inf[1], cneg);
    __syncthreads();
    // two equal neighbours in the sorted sample: the row is tie-heavy (or holds several NaN / infinities)
    bool tie = false;
#pragma unroll
    for (int e = 0; e + 1 < E; ++e) tie |= k[e] == k[e + 1];
    if (t + 1 < SNT) tie |= k[E - 1] == Sm[r2_phys<LE>((t + 1) * E)];
    const int anytie = __syncthreads_or(tie);
    // cell map over the finite part of the sample
    const int ilo = (int)(s_inf[1] < (u32)(SS - 1) ? s_inf[1] : (u32)(SS - 1));
    const int ihi = SS - 1 - (int)(s_inf[0] < (u32)(SS - 1) ? s_inf[0] : (u32)(SS - 1));
    const double lo = Sm[r2_phys<LE>(ilo)], hi = Sm[r2_phys<LE>(ihi)];
    double scale = (double)TB_C / (hi - lo);
    if (!(scale > 0.0 && scale < INF)) scale = 0.0;                     // degenerate range: one cell, plain search
    for (int j = t; j < NS; j += SNT) {
        const int pos = (j == 0) ? 0 : (j == NS - 1) ? SS - 1 : (int)(((i64)j * SS) / NB);
        const double sj = Sm[r2_phys<LE>(pos)] + 0.0;                   // -0 -> +0
        spl[rb * NS + j] = sj;
        atomicAdd(&s_cnt[tb_cell(sj, lo, scale)], 1u);
        if (j >= 1) {                                                   // interior bucket j = (s_{j-1}, s_j]
            const int pp = (j - 1 == 0) ? 0 : (int)(((i64)(j - 1) * SS) / NB);
            const double sp = Sm[r2_phys<LE>(pp)] + 0.0;
            double m = (double)Q_MAX / (sj - sp);
            // The outermost interior buckets reach from the first / last quantile splitter to the sample's extreme: under heavy
            // tails nearly all of their keys sit at the inner end and a linear image crowds them into a few of A3's fine buckets
            // (Cauchy rows: 81 keys in the first, above its 63 -- the value bucket went to the search kernel, 120 us at config
            // 3's size).  Where the sample's median inside such a bucket lies in the inner eighth of its range the bucket takes
            // the float-like code of the distance to its inner splitter, like the end buckets beyond it (mk = 0 says so).
            if (j == 1 && NS >= 3) {
                const double med = Sm[r2_phys<LE>(pos / 2)];
                if ((sj - med) * 8.0 < (sj - sp)) m = 0.0;
            } else if (j == NS - 1 && NS >= 3) {
                const double med = Sm[r2_phys<LE>((pp + pos) / 2)];
                if ((med - sp) * 8.0 < (sj - sp)) m = 0.0;
            }
            mk[rb * NBT + j] = m;
        }
    }
    if (t == 0) {
        mk[rb * NBT] = 0.0;
        mk[rb * NBT + NBT - 1] = 0.0;
        rp[rb] = make_double2(lo, scale);
        rowtied[rb] = anytie ? 1u : 0u;
    }
    __syncthreads();
    // per cell: splitters in earlier cells | splitters up to and including this cell
    u32 c[CPT], run = 0;
#pragma unroll
    for (int q = 0; q < CPT; ++q) { c[q] = s_cnt[t * CPT + q]; run += c[q]; }
    const u32 incl = rb_wave_incl_scan(run);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    u32 base = incl - run;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    (void)NWV;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        tab[rb * TB_C + t * CPT + q] = base | ((base + c[q]) << 16);
        base += c[q];
    }
}

// P3: grid = (ceil(n / 3072), rows), 512 threads x 6 keys
#ifndef SD_P3_E
#define SD_P3_E 6                      // 8: 4 096 keys per block and three workgroups per CU -- + 2 ... 3 % at every size measured
#endif
constexpr int P3_NT = 512, P3_E = SD_P3_E, P3_C = P3_NT * P3_E;
static inline size_t p3_lds_bytes(int NBT) {
    return (size_t)P3_C * 8 + (size_t)(2 * NBT + 2) * 8 + (size_t)TB_C * 4 + (size_t)(3 * NBT + 4 + P3_NT / 64) * 4 + (size_t)P3_C * 2 + 64;
}
__global__ __launch_bounds__(P3_NT) void bucket_partition3_kernel(const double *__restrict__ Y, i64 n, i64 row0, int NBT,
                                                                  const double *__restrict__ spl,
                                                                  const double *__restrict__ mk,
                                                                  const u32 *__restrict__ tab,
                                                                  const double2 *__restrict__ rp,
                                                                  const u32 *__restrict__ rowtied,
                                                                  u32 *__restrict__ bcnt, u32 *__restrict__ nnanrow,
                                                                  u32 *__restrict__ ovf, u64 *__restrict__ rec,
                                                                  u32 *__restrict__ bidx, u32 *__restrict__ gate,
                                                                  u32 epoch, AB2 ab) {
    // One block of 3 072 curves of one row per workgroup, four workgroups per CU up to ~30 value buckets (40 KB of LDS; with
    // 4 096 curves: three).  Keys per thread 4 / 5 / 6 / 7 / 8 / 10 at config 3: 0.272 / 0.265 / 0.267 / 0.278 / 0.274 / 0.293 ms
    // (profiles/r04_p3_keys_per_thread.txt).  (Measured and dropped: 4 consecutive
    // blocks per workgroup with the next block's keys prefetched and the row's table loaded once -- 95 VGPRs, two workgroups
    // per CU, 128 us against 111 us at config 3: the third resident workgroup hides more latency than the prefetch.)
    extern __shared__ double Sm3[];
    const int NS = NBT - 1;
    u64 *stage = reinterpret_cast<u64 *>(Sm3);                         // [P3_C] records in bucket order
    double *s_spl = Sm3 + P3_C;                                        // [NS]
    double *s_mk = s_spl + NBT;                                        // [NBT]
    u32 *s_tab = reinterpret_cast<u32 *>(s_mk + NBT + 2);              // [TB_C]
    u32 *s_hist = s_tab + TB_C;                                        // [NBT]
    u32 *s_gbase = s_hist + NBT;                                       // [NBT]
    u32 *s_lbase = s_gbase + NBT;                                      // [NBT + 1]
    u32 *s_wtot = s_lbase + NBT + 2;                                   // [P3_NT / 64]
    unsigned short *sbk = reinterpret_cast<unsigned short *>(s_wtot + P3_NT / 64 + 1);   // [P3_C]
    const i64 rb = blockIdx.y;
    const double *row = Y + (row0 + rb) * n;
    const i64 base = (i64)blockIdx.x * P3_C;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef SD_P3_STAGGER
    {   // three workgroups per CU: the second and third dispatch rounds start a third / two thirds of a block late
        const i64 lin = (i64)blockIdx.y * gridDim.x + blockIdx.x;
        if (lin >= SD_P3_STAGGER && lin < 2 * SD_P3_STAGGER) __builtin_amdgcn_s_sleep(64);
        if (lin >= 2 * SD_P3_STAGGER && lin < 3 * SD_P3_STAGGER) __builtin_amdgcn_s_sleep(127);
    }
#endif
    double x[P3_E];
#pragma unroll
    for (int e = 0; e < P3_E; ++e) {
        const i64 i = base + t + e * P3_NT;
        x[e] = (i < n) ? row[i] : 0.0;
    }
    for (int c = t; c < TB_C; c += P3_NT) s_tab[c] = tab[rb * TB_C + c];
    for (int b = t; b < NBT; b += P3_NT) {
        if (b < NS) s_spl[b] = spl[rb * NS + b];
        s_mk[b] = mk[rb * NBT + b];
        s_hist[b] = 0;
    }
    const double2 prm = rp[rb];
    const bool tied = rowtied[rb] != 0;                                // block-uniform
    const int per = (NBT + P3_NT - 1) / P3_NT;                         // buckets per thread in the prefix phase (<= 3)
    u32 mynan = 0;
    bool over = false;
    __syncthreads();
    {
#if defined(SD_TUNING) && defined(SD_P3_STOP)
        if (SD_P3_STOP == 1) { double a = 0; for (int e = 0; e < P3_E; ++e) a += x[e]; if (a == 1.2345e300) bidx[0] = 1; return; }
#endif
        u32 bk[P3_E], off[P3_E], q[P3_E];
#pragma unroll
        for (int e = 0; e < P3_E; ++e) {
            const i64 i = base + t + e * P3_NT;
            bk[e] = 0xFFFFFFFFu;
            off[e] = 0;
            q[e] = 0;
            if (i < n) {
                const double xv = x[e];
                if (xv == xv) {
                    const u32 tb = s_tab[tb_cell(xv, prm.x, prm.y)];
                    int a = (int)(tb & 0xFFFFu), b = (int)(tb >> 16);   // bucket = number of splitters < x, in [a, b]
                    while (a < b) {
                        const int mid = (a + b) >> 1;
                        if (s_spl[mid] < xv) a = mid + 1;
                        else b = mid;
                    }
                    bk[e] = (u32)a;
                    {
                        // curves ordered by level put a whole wave's keys into one bucket: 64 lanes on one counter are 64
                        // serialised LDS atomics (P3 117 -> 168 us on such data).  One atomic for the wave then, the lanes
                        // take their slots in lane order; any other wave as before.
                        const u64 act = __ballot(1);
                        const u32 a0 = (u32)__builtin_amdgcn_readfirstlane(a);
                        if (__ballot((u32)a == a0) == act) {
                            const int leader = __ffsll((long long)act) - 1;
                            u32 old = 0;
                            if (lane == leader) old = atomicAdd(&s_hist[a0], (u32)__popcll(act));
                            old = rb_readlane(old, leader);
                            off[e] = old + __builtin_amdgcn_mbcnt_hi((u32)(act >> 32), __builtin_amdgcn_mbcnt_lo((u32)act, 0u));
                        } else {
                            off[e] = atomicAdd(&s_hist[a], 1u);
                        }
                    }
                    if (!tied) {
                        const double xc = xv + 0.0;
                        u32 qq;
                        if (s_mk[a] == 0.0 && a <= 1) {                 // below the sample / a skewed first interior bucket (S3)
                            const u32 cd = q_code(q_ord(s_spl[a]) - q_ord(xc));
                            qq = Q_MAX - (cd < Q_MAX ? cd : Q_MAX);
                        } else if (s_mk[a] == 0.0) {                    // above the sample / a skewed last interior bucket / no width
                            const u32 cd = q_code(q_ord(xc) - q_ord(s_spl[a - 1]));
                            qq = cd < Q_MAX ? cd : Q_MAX;
                        } else {
                            double v = (xc - s_spl[a - 1]) * s_mk[a];
                            v = v < (double)Q_MAX ? v : (double)Q_MAX; // NaN -> Q_MAX
                            qq = (u32)v;
                        }
                        q[e] = qq;
                    }
                } else {
                    ++mynan;
                    ab_store_nan(ab, (size_t)(rb * n + i));
                }
            }
        }
        __syncthreads();                                               // B1: the block's histogram is complete
#if defined(SD_TUNING) && defined(SD_P3_STOP)
        if (SD_P3_STOP == 2) { u32 a = 0; for (int e = 0; e < P3_E; ++e) a ^= bk[e] ^ off[e] ^ q[e]; if (a == 0x12345678u) bidx[0] = a; return; }
#endif
        // global base of this workgroup's run in every bucket (one returning atomic per bucket: its round trip to L2 is
        // spent under the local prefix sum and the LDS scatter, the result is only needed for the copy-out); local exclusive
        // prefix of the workgroup's counts.  Thread t owns the buckets [t * per, t * per + per), per <= 3.
        u32 gb0 = 0, gb1 = 0, gb2 = 0;
        {
            u32 run = 0;
            {
                const int b = t * per;
                if (b < NBT) { const u32 c = s_hist[b]; gb0 = c ? atomicAdd(&bcnt[rb * NBT + b], c) : 0u; run += c; }
            }
            if (per > 1) {
                const int b = t * per + 1;
                if (b < NBT) { const u32 c = s_hist[b]; gb1 = c ? atomicAdd(&bcnt[rb * NBT + b], c) : 0u; run += c; }
            }
            if (per > 2) {
                const int b = t * per + 2;
                if (b < NBT) { const u32 c = s_hist[b]; gb2 = c ? atomicAdd(&bcnt[rb * NBT + b], c) : 0u; run += c; }
            }
            const u32 incl = rb_wave_incl_scan(run);
            if (lane == 63) s_wtot[wave] = incl;
            __syncthreads();                                           // B2
            u32 o = incl - run;
            for (int w = 0; w < wave; ++w) o += s_wtot[w];
            for (int r = 0; r < per; ++r) {
                const int b = t * per + r;
                if (b < NBT) {
                    s_lbase[b] = o;
                    o += s_hist[b];
                }
            }
            if (t == P3_NT - 1) s_lbase[NBT] = o;                       // non-NaN keys of the block
        }
        __syncthreads();                                               // B3
#pragma unroll
        for (int e = 0; e < P3_E; ++e) {
            if (bk[e] != 0xFFFFFFFFu) {
                const u32 lp = s_lbase[bk[e]] + off[e];
                const u32 id = (u32)(base + t + e * P3_NT);
                stage[lp] = tied ? (u64)__double_as_longlong(x[e]) : ((u64)q[e] | ((u64)id << 32));
                sbk[lp] = (unsigned short)bk[e];
            }
        }
        if (t * per < NBT) s_gbase[t * per] = gb0;
        if (per > 1 && t * per + 1 < NBT) s_gbase[t * per + 1] = gb1;
        if (per > 2 && t * per + 2 < NBT) s_gbase[t * per + 2] = gb2;
        __syncthreads();                                               // B4
#if defined(SD_TUNING) && defined(SD_P3_STOP)
        if (SD_P3_STOP == 4) { if (stage[t] == 0x123456789ull && s_gbase[0] == 77u) bidx[0] = 1; return; }
#endif
        const u32 nval = s_lbase[NBT];
        for (u32 p = t; p < nval; p += P3_NT) {
            const u32 b = sbk[p];
            const u32 g = s_gbase[b] + (p - s_lbase[b]);
            if (g < (u32)BK_C) rec[((size_t)rb * NBT + b) * BK_C + g] = stage[p];
            else over = true;
        }
        if (tied) {                                                    // fp64 records: the curve indices in a second round
            __syncthreads();
            u32 *stage32 = reinterpret_cast<u32 *>(stage);
#pragma unroll
            for (int e = 0; e < P3_E; ++e)
                if (bk[e] != 0xFFFFFFFFu) stage32[s_lbase[bk[e]] + off[e]] = (u32)(base + t + e * P3_NT);
            __syncthreads();
            for (u32 p = t; p < nval; p += P3_NT) {
                const u32 b = sbk[p];
                const u32 g = s_gbase[b] + (p - s_lbase[b]);
                if (g < (u32)BK_C) bidx[((size_t)rb * NBT + b) * BK_C + g] = stage32[p];
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) mynan += __shfl_down(mynan, o);
    if (lane == 0 && mynan) atomicAdd(&nnanrow[rb], mynan);
    if (over) { ovf[rb] = 1u; gate[1] = epoch; }                       // a gate word is "set" when it holds the batch's epoch
}

// A3: grid = 8 * NBT * ceil(rows / 8) (the XCD-aware mapping of bucket_rank_kernel), 512 threads x 16 keys
constexpr int A3_NT = 512, A3_E = SD_BK_CE, A3_LNB = 12, A3_NBF = 1 << A3_LNB, A3_CAP = 63, A3_U = 2, A3_PAD = 32;
constexpr int A3_NW = A3_NT / 64;
static_assert(A3_NT * A3_E == BK_C, "one thread slot per key of a full value bucket");
static_assert(A3_NBF / 2 / A3_NT == 4, "one 16-byte quad of histogram words per thread");
constexpr size_t A3_HDR = 256;
constexpr size_t A3_LDS32 = A3_HDR + (size_t)(A3_NBF / 2 + 4) * 4 + (size_t)(BK_C + A3_PAD) * 4 + (size_t)(BK_C + A3_PAD) * 2;
#if defined(SD_TUNING) && defined(SD_A3_EXP)
constexpr size_t A3_LDS = A3_HDR + (size_t)(A3_NBF / 2 + 4) * 4 + (size_t)(BK_C + A3_PAD) * 4 + 64;   // timing experiment: no Jx, no tied path
#define SD_A3_ATTR __attribute__((amdgpu_waves_per_eu(SD_A3_EXP, SD_A3_EXP)))
#else
constexpr size_t A3_LDS = A3_LDS32 > BR_LDS ? A3_LDS32 : BR_LDS;      // rows flagged "tied" run A' inside this kernel
#define SD_A3_ATTR
#endif

__global__ __launch_bounds__(A3_NT) SD_A3_ATTR void bucket_rank32_kernel(const double *__restrict__ Y, i64 n, i64 row0, i64 rows, int NBT,
                                                              const u32 *__restrict__ bcnt,
                                                              const u32 *__restrict__ nnanrow,
                                                              const u32 *__restrict__ ovf,
                                                              const u32 *__restrict__ rowtied,
                                                              const u64 *__restrict__ rec,
                                                              const u32 *__restrict__ bidx, u32 *__restrict__ bflag,
                                                              u32 *__restrict__ gate, u32 epoch, AB2 ab) {
    constexpr int E = A3_E, NT = A3_NT, NBF = A3_NBF, NW = A3_NW, U = A3_U;
    extern __shared__ double Sm[];
    u32 *red = reinterpret_cast<u32 *>(Sm);                            // [NW][2] min / max, then [NW] earlier-bucket sums
    u32 *wtot = red + 2 * NW;                                          // [NW]
    u32 *H = reinterpret_cast<u32 *>(Sm + A3_HDR / 8);                 // NBF packed u16 counters, then bases
    u32 *S = H + NBF / 2 + 4;                                          // images in fine-bucket order + sentinels
    unsigned short *Jx = reinterpret_cast<unsigned short *>(S + BK_C + A3_PAD);   // record slot of every image
    const unsigned short *H16 = reinterpret_cast<const unsigned short *>(H);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // the groups of 8 rows NEWEST first: the partition wrote the last rows' records last, and those are the ones the 256 MB
    // Infinity Cache still holds when this kernel starts (- 1 % at every size measured; blockIdx order: 0.2667 / 0.9026 ms at
    // 10^5 x 256 / x 1 000, this order 0.2631 / 0.8949)
    const int w = (int)(((gridDim.x >> 3) / NBT - 1 - (blockIdx.x >> 3) / NBT) * NBT + (blockIdx.x >> 3) % NBT) * 8 + (int)(blockIdx.x & 7);
#ifdef SD_A3_STAGGER
    // Two workgroups share a CU and would run their load / LDS / store phases in lock-step (the phase times of the kernel add
    // up: profiles/r03e_phase_times_config3.txt).  The workgroups of the second dispatch round start half an item late;
    // their successors inherit the offset.
    if (w >= SD_A3_STAGGER && w < 2 * SD_A3_STAGGER) {
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int b = (w >> 3) % NBT;
    const i64 rb = (i64)((w >> 3) / NBT) * 8 + (w & 7);
    if (rb >= rows) return;
    if (ovf[rb]) return;
    __builtin_amdgcn_s_setprio(2);                                    // load / histogram / scatter phases go first, the member pass of the
                                                                      // CU's other workgroup fills in (A3 146 -> 137 us at config 3)
#if !(defined(SD_TUNING) && defined(SD_A3_EXP))
    if (rowtied[rb]) {                                                // block-uniform: fp64 records, A' (same LDS, same grid)
        bucket_rank_item(w, n, rows, NBT, bcnt, nnanrow, ovf, rowtied, reinterpret_cast<const double *>(rec), bidx, bflag,
                         gate, epoch, ab);
        return;
    }
#endif
    const u32 *rowcnt = bcnt + rb * NBT;
    const int cnt = (int)rowcnt[b];
    if (cnt == 0) return;
    const size_t slot0 = ((size_t)rb * NBT + b) * BK_C;

    u32 q[E], id[E];
    {
        const u64 *rp = rec + slot0 + t;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u64 r = (t + e * NT < cnt) ? rp[e * NT] : 0xFFFFFFFFull;
            q[e] = (u32)r;
            id[e] = (u32)(r >> 32);
        }
    }
    u32 gsum = 0;
    for (int k = t; k < b; k += NT) gsum += rowcnt[k];
    gsum = rb_wave_incl_scan(gsum);
    reinterpret_cast<uint4 *>(H)[t] = make_uint4(0, 0, 0, 0);
    if (t < 4) H[NBF / 2 + t] = 0;
    if (t < A3_PAD) S[cnt + t] = 0xFFFFFFFFu;
    u32 mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e * NT >= cnt) break;
        mn = mn < q[e] ? mn : q[e];
        const u32 v = (t + e * NT < cnt) ? q[e] : 0u;
        mx = mx > v ? mx : v;
    }
    mn = rb_wave_allreduce_u32<false>(mn);
    mx = rb_wave_allreduce_u32<true>(mx);
    if (lane == 63) { red[2 * wave] = mn; red[2 * wave + 1] = mx; wtot[wave] = gsum; }
    __syncthreads();                                                  // barrier 1
#if defined(SD_TUNING) && defined(SD_A3_STOP)
    if (SD_A3_STOP == 1) { if (mn + gsum == 0x12345678u) ab.B[0] = mn; return; }   // timing experiment (results invalid)
#endif
    u32 lo, hi, gbase;
    {
        const uint2 pmm = reinterpret_cast<const uint2 *>(red)[lane & (NW - 1)];
        lo = rb_wave_allreduce_u32<false>(pmm.x);
        hi = rb_wave_allreduce_u32<true>(pmm.y);
        const u32 g = (lane < NW) ? wtot[lane] : 0u;
        gbase = rb_readlane(rb_row_incl_scan(g), 15);
    }
    const u32 nreal = (u32)n - nnanrow[rb];
    const size_t abrow = (size_t)(rb * n);
    const float fs = (float)NBF / ((float)(hi - lo) + 1.0f);
    // ---- (1) fine bucket + slot ----
    u32 bs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e * NT >= cnt) break;
        u32 fb = (u32)((float)(q[e] - lo) * fs);                      // monotone in q
        fb = fb < (u32)(NBF - 1) ? fb : (u32)(NBF - 1);
        fb = (t + e * NT < cnt) ? fb : (u32)(NBF + 2);                // no key: dummy counter
        const u32 sh = (fb & 1u) * 16u;
        const u32 old = atomicAdd(&H[fb >> 1], 1u << sh);
        bs[e] = fb | (((old >> sh) & 0xFFFFu) << 16);
    }
    __syncthreads();                                                  // barrier 2
#if defined(SD_TUNING) && defined(SD_A3_STOP)
    if (SD_A3_STOP == 2) { u32 a = 0; for (int e = 0; e < E; ++e) a ^= bs[e]; if (a == 0x12345678u) ab.B[0] = a; return; }
#endif
    // ---- (2) exclusive prefix sum; a fine bucket above A3_CAP keys: the search kernel takes the value bucket ----
    bool anyover = false;
    {
        const uint4 hq = reinterpret_cast<const uint4 *>(H)[t];
        constexpr u32 HIM = (0xFFFFu & ~(u32)A3_CAP) * 0x10001u;
        const u32 s4 = hq.x + hq.y + hq.z + hq.w;
        const u32 ov = hq.x | hq.y | hq.z | hq.w;
        const u32 run = (s4 & 0xFFFFu) + (s4 >> 16);
        const u32 incl = rb_wave_incl_scan(run);
        const bool wover = __ballot((ov & HIM) != 0) != 0;
        if (lane == 63) wtot[wave] = incl | (wover ? 0x80000000u : 0u);
        __syncthreads();                                              // barrier 3
        const u32 wt = (lane < NW) ? wtot[lane] : 0u;
        anyover = __ballot((wt >> 31) != 0) != 0;                     // block-uniform
        const u32 wscan = rb_row_incl_scan(wt & 0x7FFFFFFFu);
        u32 base = (wave ? rb_readlane(wscan, wave - 1) : 0u) + incl - run;
        uint4 o;
        o.x = base | ((base + (hq.x & 0xFFFFu)) << 16);
        base += (hq.x & 0xFFFFu) + (hq.x >> 16);
        o.y = base | ((base + (hq.y & 0xFFFFu)) << 16);
        base += (hq.y & 0xFFFFu) + (hq.y >> 16);
        o.z = base | ((base + (hq.z & 0xFFFFu)) << 16);
        base += (hq.z & 0xFFFFu) + (hq.z >> 16);
        o.w = base | ((base + (hq.w & 0xFFFFu)) << 16);
        base += (hq.w & 0xFFFFu) + (hq.w >> 16);
        reinterpret_cast<uint4 *>(H)[t] = o;
        if (t == NT - 1) H[NBF / 2] = base;                           // = cnt
    }
    if (anyover) {                                                    // heavy ties the sample did not show
        if (t == 0) { bflag[rb * NBT + b] = 1u; gate[0] = epoch; }
        return;
    }
    __syncthreads();                                                  // barrier 4
#if defined(SD_TUNING) && defined(SD_A3_STOP)
    if (SD_A3_STOP == 3) { u32 a = H[t]; for (int e = 0; e < E; ++e) a ^= bs[e]; if (a == 0x12345678u) ab.B[0] = a; return; }
#endif
    // ---- (3) scatter into fine-bucket order ----
    u32 bc[E];                                                        // base | count << 16; count 0: no key
    const u32 dummy = (u32)(cnt + A3_PAD - 1);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e * NT >= cnt) break;
        const u32 fb = bs[e] & 0xFFFFu, slot = bs[e] >> 16;
        const u32 base = H16[fb], end = H16[fb + 1];
        const bool isk = fb < (u32)NBF;
        const u32 pos = isk ? base + slot : dummy;
        S[pos] = isk ? q[e] : 0xFFFFFFFFu;
#if !(defined(SD_TUNING) && defined(SD_A3_EXP))
        Jx[pos] = (unsigned short)(t + e * NT);
#endif
        bc[e] = isk ? (base | ((end - base) << 16)) : 0u;
    }
    __syncthreads();                                                  // barrier 5
#if defined(SD_TUNING) && defined(SD_A3_STOP)
    if (SD_A3_STOP == 4) { u32 a = S[t]; for (int e = 0; e < E; ++e) a ^= bc[e] ^ id[e]; if (a == 0x12345678u) ab.B[0] = a; return; }
#endif
    // ---- (4) rank inside the fine bucket, write B ----
    __builtin_amdgcn_s_setprio(0);
    const double *yrow = Y + (row0 + rb) * n;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e * NT >= cnt) break;
        if ((e & 1) == 0) __builtin_amdgcn_sched_barrier(0);
        const u32 base = bc[e] & 0xFFFFu, fc = bc[e] >> 16;
        const u32 offq = base & 3u;
        const u32 x = q[e];
        const uint4 *Sq = reinterpret_cast<const uint4 *>(S + (base - offq));
        u32 less = 0, eq = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint4 y = Sq[u];
            less += (y.x < x) + (y.y < x) + (y.z < x) + (y.w < x);
            eq += (y.x == x) + (y.y == x) + (y.z == x) + (y.w == x);
        }
        if (fc + offq > (u32)(4 * U)) {                               // a fine bucket longer than the window
            for (u32 kk = 4 * U; kk < fc + offq; kk += 4) {
                const uint4 y = Sq[kk >> 2];
                less += (y.x < x) + (y.y < x) + (y.z < x) + (y.w < x);
                eq += (y.x == x) + (y.y == x) + (y.z == x) + (y.w == x);
            }
        }
        less -= offq;                                                 // keys in front of the base: earlier fine buckets, all smaller
        if (fc) {
            if (eq == 1u) {
#if defined(SD_TUNING) && defined(SD_A3_STOP)
                if (SD_A3_STOP == 5) { if (gbase + base + less == 0xFFFFFFF0u) ab.B[abrow + id[e]] = 1; } else     // no store
#endif
                ab_store_untied(ab, abrow + id[e], gbase + base + less, nreal);
            } else {
                // another member carries the same image: the fp64 values of those members decide
                const double xv = yrow[id[e]];
                u32 lt = 0, eqv = 0;
                for (u32 m = base; m < base + fc; ++m) {
                    if (S[m] == x) {
#if defined(SD_TUNING) && defined(SD_A3_EXP)
                        const u32 im = id[e];                          // timing experiment: no partner look-up (results invalid)
#else
                        const u32 im = (u32)(rec[slot0 + Jx[m]] >> 32);
#endif
                        const double yv = yrow[im];
                        lt += (yv < xv) ? 1u : 0u;
                        eqv += (yv == xv) ? 1u : 0u;
                    }
                }
                const u32 Bv = gbase + base + less + lt;
                ab_store(ab, abrow + id[e], Bv, nreal - (Bv + eqv), nreal);
            }
        }
    }
}

// B: persistent 1-D grid over the (row, bucket) pairs, flagged buckets only (nothing flagged: every workgroup reads a few
// flags and leaves).  rowtied == nullptr or rowtied[row]: fp64 records (bval, bidx); else 8-byte records whose keys are
// gathered from the matrix through their curve indices.
template <int NT, int E>
__device__ __forceinline__ void bucket_search_items(const double *__restrict__ Y, i64 n, i64 row0, i64 rows, int NB,
                                                    const u32 *__restrict__ bcnt, const u32 *__restrict__ nnanrow,
                                                    const u32 *__restrict__ bflag, const u32 *__restrict__ rowtied,
                                                    const double *__restrict__ bval, const u32 *__restrict__ bidx,
                                                    const AB2 &ab, double *Sm) {
    using C = R2Cfg<NT, E>;
    constexpr int LE = C::LE, WB = C::WB, N = C::N;
    static_assert(NT * E >= BK_C, "a value bucket fits the sort");
    __shared__ u32 s_basecnt;
    const u64 *rec = reinterpret_cast<const u64 *>(bval);
    for (i64 v = blockIdx.x; v < rows * NB; v += gridDim.x) {
        if (!bflag[v]) continue;                                      // block-uniform
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));                                   // per-item opaque thread id
        const int lane = t & 63, wave = t >> 6;
        const int b = (int)(v % NB);
        const i64 rb = v / NB;
        const bool packed = rowtied && !rowtied[rb];
        const double *yrow = Y + (row0 + rb) * n;
        const int cnt = (int)bcnt[rb * NB + b];
        __syncthreads();                                              // the previous item's image is no longer read
        if (t == 0) {
            u32 sum = 0;
            for (int q = 0; q < b; ++q) sum += bcnt[rb * NB + q];
            s_basecnt = sum;
        }
        const int n_act = ((cnt + WB - 1) / WB) * WB;
        const bool wreal = wave * WB < n_act;
        const double INF = __builtin_huge_val();
        const size_t slot0 = ((size_t)rb * NB + b) * BK_C;
        const int i0 = wave * WB + lane;
        double k[E];
        if (wreal) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int j = i0 + e * 64;
                k[e] = (j < cnt) ? (packed ? yrow[(u32)(rec[slot0 + j] >> 32)] : bval[slot0 + j]) : INF;
            }
        }
        R2Sorter<NT, E>::sort(k, Sm, t, n_act, wreal, INF);
        if (wreal) {
            double *Sw = Sm + r2_base<0, LE>(t);
#pragma unroll
            for (int e = 0; e < E; ++e) Sw[r2_off<0, LE>(e)] = k[e];
        }
        __syncthreads();
        const u32 base = s_basecnt;
        const u32 nreal = (u32)n - nnanrow[rb];
        for (int j = t; j < cnt; j += NT) {
            const u32 id = packed ? (u32)(rec[slot0 + j] >> 32) : bidx[slot0 + j];
            const double x = packed ? yrow[id] : bval[slot0 + j];
            int lo = r2_bound<N, SlotPad<LE>, false, false>(Sm, n_act, x, INF);       // x is in the bucket
            int hi = lo + 1, step = 1;
            while (hi + step <= n_act && Sm[r2_phys<LE>(hi + step - 1)] <= x) { hi += step; step <<= 1; }
            while (step > 1) {
                step >>= 1;
                if (hi + step <= n_act && Sm[r2_phys<LE>(hi + step - 1)] <= x) hi += step;
            }
            // values above x: everything real beyond x's tie run (x = +inf: the padding ties with it, nothing is above)
            ab_store(ab, (size_t)(rb * n + id), base + (u32)lo, (x == INF) ? 0u : nreal - (base + (u32)hi), nreal);
        }
    }
}

__global__ __launch_bounds__(BK_NT) void bucket_search_kernel(const double *__restrict__ Y, i64 n, i64 row0, i64 rows, int NB,
                                                              const u32 *__restrict__ bcnt,
                                                              const u32 *__restrict__ nnanrow,
                                                              const u32 *__restrict__ bflag,
                                                              const u32 *__restrict__ rowtied,
                                                              const double *__restrict__ bval,
                                                              const u32 *__restrict__ bidx,
                                                              const u32 *__restrict__ gate, u32 epoch, AB2 ab) {
    extern __shared__ double Sm[];
    if (gate && *gate != epoch) return;                               // no bucket of this batch was flagged
    bucket_search_items<BK_NT, BK_E>(Y, n, row0, rows, NB, bcnt, nnanrow, bflag, rowtied, bval, bidx, ab, Sm);
}

// The three fall-backs of a batch in ONE launch (product path): flagged value buckets (gate[0]), then the chunked route for
// the rows whose partition overflowed (gate[1]): every workgroup sorts its share of the rows' chunks, the grid meets at a
// counter, every workgroup searches its share of (row, chunk) items.  Nothing flagged: every workgroup reads the two gate
// words and leaves.
// The meeting (meet[], zeroed by S3 every batch): [0] arrivals, [1] workgroups that gave up waiting, [2 + w] who searches
// workgroup w's items.  The grid is one workgroup per CU the stream may use, so every workgroup the others wait for is
// running -- unless something the host cannot see keeps CUs from this launch (a process-wide mask, another process).  So the
// wait is bounded: a workgroup that has waited spin_limit polls announces it, looks once more, and leaves its items to the
// LAST workgroup to arrive, which exists whatever happens (nobody waits for a workgroup that has not sorted yet without
// eventually making room for it) and sweeps the claim words when anybody gave up.  A claim word changes once, by
// compare-and-swap (0 -> FB_SELF by its owner, FB_LEFT by its owner on leaving, FB_TAKEN by the last arriver): every item is
// searched exactly once.
constexpr u32 FB_SELF = 1u, FB_LEFT = 2u, FB_TAKEN = 3u;
constexpr int FB_MAXGRID = 1024;
constexpr int FB_MEET_WORDS = 2 + FB_MAXGRID;
static_assert(FB_MEET_WORDS == FB_MEET_WORDS_S3, "S3 zeroes the meeting words");
__global__ __launch_bounds__(BIG_NT) void big_fallback_kernel(const double *__restrict__ Y, i64 n, i64 row0, i64 rows, int NBT,
                                                              const u32 *__restrict__ bcnt,
                                                              const u32 *__restrict__ nnanrow,
                                                              const u32 *__restrict__ bflag,
                                                              const u32 *__restrict__ rowtied,
                                                              const double *__restrict__ bval,
                                                              const u32 *__restrict__ bidx, double *sorted, i64 sstride,
                                                              u32 *nanf, int nch, const u32 *__restrict__ ovf,
                                                              const u32 *__restrict__ gate, u32 epoch, u32 *meet,
                                                              u32 spin_limit, AB2 ab) {
    extern __shared__ double Sm[];
    __shared__ u32 s_role;                                            // 0: leave, 1: search my items, 2: ... and sweep
    const bool flagged = gate[0] == epoch, over = gate[1] == epoch;   // block-uniform
    if (!flagged && !over) return;
    if (flagged) bucket_search_items<BIG_NT, BIG_E>(Y, n, row0, rows, NBT, bcnt, nnanrow, bflag, rowtied, bval, bidx, ab, Sm);
    if (!over) return;
    __syncthreads();
    chunk_sort_items(Y, n, row0, rows, nch, sorted, sstride, nanf, ovf, Sm);
    __syncthreads();
    const u32 G = gridDim.x, me = blockIdx.x;
    auto arrived = [&]() { return __hip_atomic_load(&meet[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (threadIdx.x == 0) {
        __threadfence();                                              // the sorted chunks and the NaN counts, device-wide
        const u32 ticket = atomicAdd(&meet[0], 1u);
        const bool last = ticket + 1u == G;
        bool full = last;
        for (u32 spins = 0; !full && spins < spin_limit; ++spins) {
            __builtin_amdgcn_s_sleep(32);
            full = arrived() >= G;
        }
        if (!full) {                                                  // waited long enough: say so, then look once more
            atomicAdd(&meet[1], 1u);
            __threadfence();
            full = arrived() >= G;
        }
        u32 role = 0;
        if (full) role = atomicCAS(&meet[2 + me], 0u, FB_SELF) == 0u ? 1u : 0u;
        else atomicCAS(&meet[2 + me], 0u, FB_LEFT);
        if (last) role |= 2u;
        __threadfence();
        s_role = role;
    }
    __syncthreads();
    const u32 role = s_role;
    if (role & 1u) chunk_search_items(Y, n, row0, rows, sorted, sstride, nanf, nch, ovf, ab, Sm, me, G);
    if (!(role & 2u)) return;
    // the last arriver: did anybody give up?  (Whoever did said so before its last look at the arrivals, which did not
    // include this workgroup's.)
    __syncthreads();
    if (threadIdx.x == 0) s_role = __hip_atomic_load(&meet[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_role == 0u) return;
    for (u32 w = 0; w < G; ++w) {
        if (w == me) continue;
        __syncthreads();
        if (threadIdx.x == 0) {
            const u32 v = atomicCAS(&meet[2 + w], 0u, FB_TAKEN);
            s_role = (v == 0u || v == FB_LEFT) ? 1u : 0u;
        }
        __syncthreads();
        if (s_role) chunk_search_items(Y, n, row0, rows, sorted, sstride, nanf, nch, ovf, ab, Sm, w, G);
    }
}

// =====================================================================================================
// fold the pair image into the totals of the targets: block = 64 targets x 16 row slices
// =====================================================================================================
template <int J>
__global__ __launch_bounds__(1024) void rank_accumulate2_kernel(AB2 ab, const u32 *__restrict__ nnan,
                                                                i64 rows, i64 n, const i64 *__restrict__ targets,
                                                                i64 tbegin, i64 m, u64 *__restrict__ out, int first) {
    __shared__ u64 red[16][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const i64 q = (i64)blockIdx.x * 64 + x;
    const i64 i = (q < m) ? (targets ? targets[q] : tbegin + q) : 0;
    u64 acc[JMAX - 1];
#pragma unroll
    for (int j = 0; j < JMAX - 1; ++j) acc[j] = 0;
    if (q < m) {
        i64 r = y;
        auto fold = [&](u32 w, u32 nn, i64 row) {
            if (w == AB2_NAN) return;
            const u32 B = w & ~AB2_TIE;
            if (J == 2 && nn == 0 && !(w & AB2_TIE)) {                // untied key of a NaN-free row: contained = A * B
                acc[0] += (u64)B * (u64)((u32)n - 1u - B);
                return;
            }
            const u32 A = (w & AB2_TIE) ? ab.A[row * n + i] : (u32)n - nn - 1u - B;     // tied keys carry their A
            band_counts_add<J>(A, B, nn, (u64)(n - 1), acc);
        };
        // the word of (row, curve): from the half-word image where there is one (min(A, B) of an untied key stands for B:
        // the counts are symmetric in A and B), else / on its marker from the B image
        auto word = [&](i64 row) -> u32 {
            if (ab.H) {
                const u32 h = ab.H[row * n + i];
                if (h != (u32)AB2_H_WORD) return h;
            }
            return ab.B[row * n + i];
        };
        for (; r + 16 * 7 < rows; r += 16 * 8) {
            u32 w[8], nn[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                w[u] = word(r + 16 * u);
                nn[u] = nnan[r + 16 * u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) fold(w[u], nn[u], r + 16 * u);
        }
        for (; r < rows; r += 16) fold(word(r), nnan[r], r);
    }
#pragma unroll
    for (int j = 0; j < J - 1; ++j) {
        red[y][x] = acc[j];
        __syncthreads();
        if (y == 0 && q < m) {
            u64 tot = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) tot += red[k][x];
            if (first) out[q * (J - 1) + j] = tot;
            else out[q * (J - 1) + j] += tot;
        }
        __syncthreads();
    }
}

// The same fold for a contiguous, 4-aligned block of targets (the usual call: every curve): a thread takes FOUR neighbouring
// curves through 16-byte loads, so half a wave reads 512 bytes of a row instead of 128.  block = 32 quads x 16 row slices;
// needs n % 4 == 0 and tbegin % 4 == 0 (the image rows stay 16-byte aligned).  J <= 3 (accumulators in registers).
template <int J>
__global__ __launch_bounds__(512) void rank_accumulate2x4_kernel(AB2 ab, const u32 *__restrict__ nnan, i64 rows, i64 n,
                                                                  i64 tbegin, i64 m, u64 *__restrict__ out, int first) {
    __shared__ u64 red[16][32];
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
    const i64 q0 = ((i64)blockIdx.x * 32 + x) * 4;                    // first of this thread's four targets
    const i64 i0 = tbegin + q0;
    u64 acc[4][JMAX - 1];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < JMAX - 1; ++j) acc[c][j] = 0;
    if (q0 < m) {
        auto fold4 = [&](const uint4 w, u32 nn, i64 row) {
            const u32 ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (ww[c] == AB2_NAN) continue;
                const u32 B = ww[c] & ~AB2_TIE;
                if (J == 2 && nn == 0 && !(ww[c] & AB2_TIE)) {
                    // untied key of a NaN-free row: A + B = n - 1, and C(A+B, 2) - C(A, 2) - C(B, 2) = A * B
                    acc[c][0] += (u64)B * (u64)((u32)n - 1u - B);
                    continue;
                }
                const u32 A = (ww[c] & AB2_TIE) ? ab.A[row * n + i0 + c] : (u32)n - nn - 1u - B;
                band_counts_add<J>(A, B, nn, (u64)(n - 1), acc[c]);
            }
        };
        const uint4 *img = reinterpret_cast<const uint4 *>(ab.B);
        const uint2 *himg = reinterpret_cast<const uint2 *>(ab.H);
        // four half words (8-byte loads) where there is a half-word image; a marker sends that curve to the B image
        auto words = [&](i64 row) -> uint4 {
            if (!himg) return img[(row * n + i0) >> 2];
            const uint2 h = himg[(row * n + i0) >> 2];
            uint4 w = make_uint4(h.x & 0xFFFFu, h.x >> 16, h.y & 0xFFFFu, h.y >> 16);
            if (w.x == (u32)AB2_H_WORD) w.x = ab.B[row * n + i0];
            if (w.y == (u32)AB2_H_WORD) w.y = ab.B[row * n + i0 + 1];
            if (w.z == (u32)AB2_H_WORD) w.z = ab.B[row * n + i0 + 2];
            if (w.w == (u32)AB2_H_WORD) w.w = ab.B[row * n + i0 + 3];
            return w;
        };
        i64 r = y;
        for (; r + 16 * 3 < rows; r += 16 * 4) {
            uint4 w[4];
            u32 nn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[u] = words(r + 16 * u);
                nn[u] = nnan[r + 16 * u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) fold4(w[u], nn[u], r + 16 * u);
        }
        for (; r < rows; r += 16) fold4(words(r), nnan[r], r);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int j = 0; j < J - 1; ++j) {
            red[y][x] = acc[c][j];
            __syncthreads();
            if (y == 0 && q0 + c < m) {
                u64 tot = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) tot += red[k][x];
                if (first) out[(q0 + c) * (J - 1) + j] = tot;
                else out[(q0 + c) * (J - 1) + j] += tot;
            }
            __syncthreads();
        }
    }
}

// The fold over the HALF-WORD image for a contiguous, 8-aligned block of targets: a thread takes EIGHT neighbouring curves
// through 16-byte loads (16 lanes read 256 bytes of a row); block = 16 octets x 32 row slices.  A marker among the eight (NaN
// / tied key: rare) sends that load through the B words.  Needs n % 8 == 0 and tbegin % 8 == 0.  J <= 3.
template <int J>
__global__ __launch_bounds__(512) void rank_accumulate_h8_kernel(AB2 ab, const u32 *__restrict__ nnan, i64 rows, i64 n,
                                                                 i64 tbegin, i64 m, u64 *__restrict__ out, int first) {
    __shared__ u64 red[32][16][9];                                    // (+1: the row slices fall on different banks)
    const int x = threadIdx.x & 15, y = threadIdx.x >> 4;
    const i64 q0 = ((i64)blockIdx.x * 16 + x) * 8;                    // first of this thread's eight targets
    const i64 i0 = tbegin + q0;
    u64 acc[8][JMAX - 1];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int j = 0; j < JMAX - 1; ++j) acc[c][j] = 0;
    if (q0 < m) {
        const uint4 *himg = reinterpret_cast<const uint4 *>(ab.H);
        const u32 nm1 = (u32)n - 1u;
        auto fold8 = [&](const uint4 v, u32 nn, i64 row) {
            const u32 p[4] = {v.x, v.y, v.z, v.w};
            u32 anym = 0;                                             // a half word 0xFFFF <=> a zero half word in ~p
#pragma unroll
            for (int k = 0; k < 4; ++k) anym |= ((~p[k]) - 0x00010001u) & p[k] & 0x80008000u;
            if (J == 2 && nn == 0 && !anym) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const u32 h = (c & 1) ? (p[c >> 1] >> 16) : (p[c >> 1] & 0xFFFFu);
                    acc[c][0] += (u64)h * (u64)(nm1 - h);              // untied key of a NaN-free row: A * B
                }
                return;
            }
            // markers: the eight B words, and the eight A words when one of them is tied, through 16-byte loads as well (a
            // tie-heavy row is markers throughout)
            u32 bw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, aw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (anym) {
                const uint4 *bp = reinterpret_cast<const uint4 *>(ab.B + row * n + i0);
                const uint4 b0 = bp[0], b1 = bp[1];
                bw[0] = b0.x; bw[1] = b0.y; bw[2] = b0.z; bw[3] = b0.w; bw[4] = b1.x; bw[5] = b1.y; bw[6] = b1.z; bw[7] = b1.w;
                u32 anyt = 0;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const u32 h = (c & 1) ? (p[c >> 1] >> 16) : (p[c >> 1] & 0xFFFFu);
                    anyt |= (h == (u32)AB2_H_WORD && bw[c] != AB2_NAN) ? (bw[c] & AB2_TIE) : 0u;
                }
                if (anyt) {
                    const uint4 *ap = reinterpret_cast<const uint4 *>(ab.A + row * n + i0);
                    const uint4 a0 = ap[0], a1 = ap[1];
                    aw[0] = a0.x; aw[1] = a0.y; aw[2] = a0.z; aw[3] = a0.w; aw[4] = a1.x; aw[5] = a1.y; aw[6] = a1.z; aw[7] = a1.w;
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                u32 w = (c & 1) ? (p[c >> 1] >> 16) : (p[c >> 1] & 0xFFFFu);
                if (w == (u32)AB2_H_WORD) w = bw[c];
                if (w == AB2_NAN) continue;
                const u32 B = w & ~AB2_TIE;
                const u32 A = (w & AB2_TIE) ? aw[c] : (u32)n - nn - 1u - B;
                band_counts_add<J>(A, B, nn, (u64)(n - 1), acc[c]);
            }
        };
        i64 r = y;
        for (; r + 32 * 3 < rows; r += 32 * 4) {
            uint4 w[4];
            u32 nn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[u] = himg[((r + 32 * u) * n + i0) >> 3];
                nn[u] = nnan[r + 32 * u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) fold8(w[u], nn[u], r + 32 * u);
        }
        for (; r < rows; r += 32) fold8(himg[(r * n + i0) >> 3], nnan[r], r);
    }
#pragma unroll
    for (int j = 0; j < J - 1; ++j) {
#pragma unroll
        for (int c = 0; c < 8; ++c) red[y][x][c] = acc[c][j];
        __syncthreads();
        if (threadIdx.x < 128) {
            const int xx = threadIdx.x >> 3, c = threadIdx.x & 7;
            const i64 q = ((i64)blockIdx.x * 16 + xx) * 8 + c;
            if (q < m) {
                u64 tot = 0;
#pragma unroll
                for (int k = 0; k < 32; ++k) tot += red[k][xx][c];
                if (first) out[q * (J - 1) + j] = tot;
                else out[q * (J - 1) + j] += tot;
            }
        }
        __syncthreads();
    }
}

// =====================================================================================================
// host side
// =====================================================================================================
static inline i64 big_nchunks(i64 n) { return (n + BIG_C - 1) / BIG_C; }

struct BigPlan {
    i64 nch, sstride, rpb;
    int NB, NBT;                                         // interior value buckets; NBT = NB + 2 with the two end buckets
    size_t off_ab, off_h, off_sorted, off_bval, off_bidx, off_spl, off_mk, off_tab, off_rp, off_zero, zero_bytes, total;
    // zeroed block: bcnt[rpb*NBT] | nnanrow[rpb] | ovf[rpb] | bflag[rpb*NBT] | nanrow_f[rpb] | rowtied[rpb]
    size_t z_bcnt, z_nnan, z_ovf, z_bflag, z_nanf, z_tied, z_gate, z_meet;
};

#ifndef SD_BIG_SCRATCH_BYTES
#define SD_BIG_SCRATCH_BYTES ((size_t)1 << 32)
#endif
#ifndef SD_S3_SMALL_NB
#define SD_S3_SMALL_NB 24
#endif
constexpr int S3_SMALL_NB = SD_S3_SMALL_NB;             // up to that many value buckets: the 2 048-key sample
#ifndef SD_S3_MID_NB
#define SD_S3_MID_NB 72
#endif
constexpr int S3_MID_NB = SD_S3_MID_NB;                 // ... the 4 096-key sample; beyond: 16 384 keys
static BigPlan big_plan(i64 T, i64 n) {
    BigPlan p;
    p.nch = big_nchunks(n);
    p.sstride = p.nch * BIG_C;
    p.NB = bucket_count(n);
    p.NBT = p.NB + 2;
    const size_t per_row = (size_t)n * 8 + (size_t)p.sstride * 8 + (size_t)p.NBT * BK_C * 12 + (size_t)p.NBT * 32 +
                           (size_t)TB_C * 4 + 128;
    // Scratch per batch: every batch costs 40 - 45 us by itself (S3's latency in front of the partition, the kernels' tails, the
    // fold's launch; profiles/r04_experiment_rows_per_batch.txt), and 288 GB of HBM have room: 4 GiB take 10^5 curves x 1 000
    // timepoints in one batch (1.5 GiB: three).
    i64 r = (i64)((size_t)SD_BIG_SCRATCH_BYTES / per_row);
    const i64 vb = xswitch("SD_RANK_ROWS_PER_BATCH");          // cross-check builds: several batches on small inputs
    if (vb > 0 && vb < r) r = vb;
    if (r < 1) r = 1;
    if (r > T) r = T;
    if (r > 65535) r = 65535;
    p.rpb = r;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
    p.off_ab = take((size_t)r * n * 8);
    p.off_h = take(n <= AB2_H_MAXN ? (size_t)r * n * 2 : 0);
    p.off_sorted = take((size_t)r * p.sstride * 8);
    p.off_bval = take((size_t)r * p.NBT * BK_C * 8);
    p.off_bidx = take((size_t)r * p.NBT * BK_C * 4);
    p.off_spl = take((size_t)r * p.NBT * 8);
    p.off_mk = take((size_t)r * p.NBT * 8);
    p.off_tab = take((size_t)r * TB_C * 4);
    p.off_rp = take((size_t)r * 16);
    p.off_zero = o;
    size_t z = 0;
    auto ztake = [&](size_t bytes) { size_t at = z; z = align_up(z + bytes, 256); return at; };
    p.z_bcnt = ztake((size_t)r * p.NBT * 4);
    p.z_nnan = ztake((size_t)r * 4);
    p.z_ovf = ztake((size_t)r * 4);
    p.z_bflag = ztake((size_t)r * p.NBT * 4);
    p.z_nanf = ztake((size_t)r * 4);
    p.z_tied = ztake((size_t)r * 4);
    p.z_gate = ztake(16 * 4);                                // per batch: [0] some bucket flagged, [1] some row overflowed
    p.z_meet = ztake((size_t)FB_MEET_WORDS * 4);             // the fall-back kernel's meeting
    p.zero_bytes = z;
    p.total = o + z + 256;
    return p;
}

bool mbd_rank_big_supported(i64 T, i64 n, int J) {
    (void)T;
    return n > 16384 && n < ((i64)1 << 31) && J >= 2 && J <= JMAX && big_nchunks(n) <= 1024 &&
           bucket_count(n) + 2 <= BK_MAXNB;
}

size_t mbd_rank_big_workspace_bytes(i64 T, i64 n, int J) {
    if (!mbd_rank_big_supported(T, n, J)) return 0;
    return big_plan(T, n).total + 1024;
}

// img_out != nullptr: no fold -- the B words of every (row, curve) go to img_out[T][n] (bit 31: the curve ties with another
// one at this timepoint; 0xFFFFFFFF: NaN) and the rows' NaN counts to nnan_out[T]: the strict path's rank image for n > 32 767
static int big_run(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J, u64 *out, void *ws,
                   size_t ws_bytes, hipStream_t s, u32 *img_out, u32 *nnan_out);

int launch_mbd_rank_big(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                        u64 *out, void *ws, size_t ws_bytes, hipStream_t s) {
    return big_run(Y, T, n, targets, tbegin, m, J, out, ws, ws_bytes, s, nullptr, nullptr);
}

int launch_rank_big_image(const double *Y, i64 T, i64 n, u32 *img, u32 *nnan, void *ws, size_t ws_bytes, hipStream_t s) {
    return big_run(Y, T, n, nullptr, 0, n, 2, nullptr, ws, ws_bytes, s, img, nnan);
}

static int big_run(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J, u64 *out, void *ws,
                   size_t ws_bytes, hipStream_t s, u32 *img_out, u32 *nnan_out) {
    if (!mbd_rank_big_supported(T, n, J)) return fail(SD_ERR_UNSUPPORTED, "large-n rank kernels cover n > 16384");
    const BigPlan p = big_plan(T, n);
    if (!ws || ws_bytes < p.total) return fail(SD_ERR_WORKSPACE, "large-n rank workspace too small");
    char *w = (char *)(((size_t)ws + 255) / 256 * 256);
    AB2 ab;                                                 // per batch: rpb * n words of B, then as many of A
    ab.B = (u32 *)(w + p.off_ab);
    ab.A = ab.B + (size_t)p.rpb * n;
    ab.H = nullptr;
    double *sorted = (double *)(w + p.off_sorted);
    double *bval = (double *)(w + p.off_bval);
    u32 *bidx = (u32 *)(w + p.off_bidx);
    double *spl = (double *)(w + p.off_spl);
    double *mk = (double *)(w + p.off_mk);
    u32 *tab = (u32 *)(w + p.off_tab);
    double2 *rp = (double2 *)(w + p.off_rp);
    char *zb = w + p.off_zero;
    u32 *bcnt = (u32 *)(zb + p.z_bcnt), *nnanrow = (u32 *)(zb + p.z_nnan), *ovf = (u32 *)(zb + p.z_ovf);
    u32 *bflag = (u32 *)(zb + p.z_bflag), *nanf = (u32 *)(zb + p.z_nanf), *rowtied = (u32 *)(zb + p.z_tied);
    u32 *gate = (u32 *)(zb + p.z_gate);
    u32 *meet = (u32 *)(zb + p.z_meet);
    // A gate word is "set" when it holds the batch's epoch (a process-wide counter, never 0): nothing has to zero it, and
    // stale workspace contents can at worst make a fallback kernel scan flags that S3 has zeroed -- time, never results.
    static std::atomic<u32> epoch_counter{0};

    const bool buckets = xswitch("SD_BIG_IMPL") != 1;       // cross-check builds, 1: chunked route for every row
#ifdef SD_CROSSCHECK
    const bool gen2 = xswitch("SD_BIG_GEN2") == 1 || xswitch("SD_BIG_SORT") == 1 || xswitch("SD_BIG_PART1") == 1;
#else
    const bool gen2 = false;
#endif
    // fold mode up to 131 070 curves: the half-word image (cross-check builds, SD_BIG_NOHALF = 1: B words as before; the second
    // generation's packed kernel writes B words only)
    if (!img_out && !gen2 && n <= AB2_H_MAXN && xswitch("SD_BIG_NOHALF") != 1) ab.H = (unsigned short *)(w + p.off_h);
    const int NB = p.NB, NBT = p.NBT;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const unsigned pgrid = (unsigned)cus;                   // small persistent grids of the fallback kernels
    // The merged fall-back kernel's workgroups wait for each other: its grid is what the launch's CUs hold at once
    // (one workgroup on each CU the stream may use), never more.
    unsigned fgrid = 1;
    // polls (about a microsecond each) before a waiting workgroup leaves its items to the last arriver; cross-check builds:
    // SD_BIG_SPIN = 1 makes every workgroup but the last give up at once (the take-over path under test)
    u32 spin_limit = 1u << 20;
    if (xswitch("SD_BIG_SPIN") > 0) spin_limit = (u32)xswitch("SD_BIG_SPIN") - 1u;
    {
        int usable = cus;
        uint32_t mask[16] = {0};
        if (hipExtStreamGetCUMask(s, 16, mask) == hipSuccess) {
            int bits = 0;
            for (int i = 0; i < 16; ++i) bits += __builtin_popcount(mask[i]);
            if (bits > 0 && bits < usable) usable = bits;
        } else {
            (void)hipGetLastError();
        }
        fgrid = (unsigned)(usable < 1 ? 1 : usable);         // one workgroup per CU (a kernel that launches at all fits once)
        if (fgrid > (unsigned)FB_MAXGRID) fgrid = FB_MAXGRID;
    }
    auto k_cs = chunk_sort_kernel;
    auto k_cq = chunk_search_kernel;
    // sample per row: 2 048 values up to 24 value buckets (>= 85 samples per bucket), 4 096 up to 72, 16 384 above.
    // The sort of the sample by ONE workgroup is pure latency in front of the partition (4 096 keys: 33 us, as 256 x 16
    // or 1024 x 4 alike), so the sample is no larger than the buckets' capacity margin needs.
    auto k_s3_small = bucket_setup_kernel<128, 16>;
    auto k_s3 = bucket_setup_kernel<256, 16>;
    auto k_s3_big = bucket_setup_kernel<1024, 16>;
    constexpr size_t lds_sp_small = R2Cfg<128, 16>::LDS_BYTES, lds_sp = R2Cfg<256, 16>::LDS_BYTES;
    constexpr size_t lds_sp_big = R2Cfg<1024, 16>::LDS_BYTES;
    auto k_bs = bucket_search_kernel;
    const size_t lds_p3 = p3_lds_bytes(NBT);
    SD_HIP(hipFuncSetAttribute((const void *)k_cs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BigCfg::LDS_BYTES));
    SD_HIP(hipFuncSetAttribute((const void *)k_cq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BigCfg::LDS_BYTES));
    SD_HIP(hipFuncSetAttribute((const void *)k_s3_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp_small));
    SD_HIP(hipFuncSetAttribute((const void *)k_s3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp));
    SD_HIP(hipFuncSetAttribute((const void *)k_s3_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp_big));
    SD_HIP(hipFuncSetAttribute((const void *)bucket_partition3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p3));
    SD_HIP(hipFuncSetAttribute((const void *)bucket_rank32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)A3_LDS));
    SD_HIP(hipFuncSetAttribute((const void *)k_bs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BkCfg::LDS_BYTES));
    SD_HIP(hipFuncSetAttribute((const void *)big_fallback_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BigCfg::LDS_BYTES));
#ifdef SD_CROSSCHECK
    auto k_br = bucket_rank_kernel;
    SD_HIP(hipFuncSetAttribute((const void *)k_br, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BR_LDS));
    auto k_bp = bucket_packed_kernel;
    auto k_sp_small = bucket_splitters_kernel<128, 16>;
    auto k_sp = bucket_splitters_kernel<256, 16>;
    auto k_sp_big = bucket_splitters_kernel<1024, 16>;
    const size_t lds_bk = BkCfg::LDS_BYTES + (size_t)BK_NT * 8;
    SD_HIP(hipFuncSetAttribute((const void *)k_sp_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp_small));
    SD_HIP(hipFuncSetAttribute((const void *)k_sp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp));
    SD_HIP(hipFuncSetAttribute((const void *)k_sp_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp_big));
    SD_HIP(hipFuncSetAttribute((const void *)k_bp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bk));
    SD_HIP(hipFuncSetAttribute((const void *)bucket_partition2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)((size_t)BP2_C * 14)));
#endif

    for (i64 row0 = 0; row0 < T; row0 += p.rpb) {
        const i64 rows = T - row0 < p.rpb ? T - row0 : p.rpb;
        const u32 *fallback_rows = nullptr;                  // chunked route: every row
        const u32 *nn_for_fold = nanf;
        const u32 *gate_o = nullptr;                         // no gate: the chunk kernels look at every row flag
        bool merged = false;
        if (img_out) ab.B = img_out + (size_t)row0 * n;      // image mode: the B words straight into the caller's image
        u32 epoch = ++epoch_counter;
        if (epoch == 0) epoch = ++epoch_counter;
        if (!buckets || gen2) SD_HIP(hipMemsetAsync(zb, 0, p.zero_bytes, s));      // (the third generation's S3 zeroes)
        if (buckets && !gen2) {
            // ---- third generation: S3 -> P3 -> A3 (untied rows) / A' (tied rows) ----
            if (NB <= S3_SMALL_NB)
                hipLaunchKernelGGL(k_s3_small, dim3((unsigned)rows), dim3(128), lds_sp_small, s, Y, n, row0, NB, spl, mk, tab,
                                   rp, rowtied, bcnt, bflag, nnanrow, ovf, nanf, meet);
            else if (NB <= S3_MID_NB)
                hipLaunchKernelGGL(k_s3, dim3((unsigned)rows), dim3(256), lds_sp, s, Y, n, row0, NB, spl, mk, tab, rp,
                                   rowtied, bcnt, bflag, nnanrow, ovf, nanf, meet);
            else
                hipLaunchKernelGGL(k_s3_big, dim3((unsigned)rows), dim3(1024), lds_sp_big, s, Y, n, row0, NB, spl, mk, tab,
                                   rp, rowtied, bcnt, bflag, nnanrow, ovf, nanf, meet);
            hipLaunchKernelGGL(bucket_partition3_kernel, dim3((unsigned)((n + P3_C - 1) / P3_C), (unsigned)rows), dim3(P3_NT),
                               lds_p3, s, Y, n, row0, NBT, (const double *)spl, (const double *)mk, (const u32 *)tab,
                               (const double2 *)rp, (const u32 *)rowtied, bcnt, nnanrow, ovf, (u64 *)bval, bidx, gate, epoch, ab);
            hipLaunchKernelGGL(bucket_rank32_kernel, dim3((unsigned)(8 * NBT * ((rows + 7) / 8))), dim3(A3_NT), A3_LDS, s, Y, n,
                               row0, rows, NBT, (const u32 *)bcnt, (const u32 *)nnanrow, (const u32 *)ovf,
                               (const u32 *)rowtied, (const u64 *)bval, (const u32 *)bidx, bflag, gate, epoch, ab);
            // the fall-backs (flagged buckets; chunked route for rows whose partition overflowed) behind their gate words
            hipLaunchKernelGGL(big_fallback_kernel, dim3(fgrid), dim3(BIG_NT), BigCfg::LDS_BYTES, s, Y, n, row0, rows, NBT,
                               (const u32 *)bcnt, (const u32 *)nnanrow, (const u32 *)bflag, (const u32 *)rowtied,
                               (const double *)bval, (const u32 *)bidx, sorted, p.sstride, nanf, (int)p.nch,
                               (const u32 *)ovf, (const u32 *)gate, epoch, meet, spin_limit, ab);
            fallback_rows = ovf;
            nn_for_fold = nnanrow;
            gate_o = gate + 1;
            merged = true;
        }
#ifdef SD_CROSSCHECK
        if (buckets && gen2) {
            // ---- second generation (cross-check builds): S -> P' (or P) -> A' (or the packed-key sort) ----
            const bool rank_nosort = xswitch("SD_BIG_SORT") != 1;
            if (NB <= 24)
                hipLaunchKernelGGL(k_sp_small, dim3((unsigned)rows), dim3(128), lds_sp_small, s, Y, n, row0, NB, spl);
            else if (NB <= 72)
                hipLaunchKernelGGL(k_sp, dim3((unsigned)rows), dim3(256), lds_sp, s, Y, n, row0, NB, spl);
            else
                hipLaunchKernelGGL(k_sp_big, dim3((unsigned)rows), dim3(1024), lds_sp_big, s, Y, n, row0, NB, spl);
            if (xswitch("SD_BIG_PART1") == 1)                    // first-generation partition (direct scatter)
                hipLaunchKernelGGL(bucket_partition_kernel, dim3((unsigned)((n + 16383) / 16384), (unsigned)rows), dim3(1024),
                                   0, s, Y, n, row0, NB, (const double *)spl, bcnt, nnanrow, ovf, bval, bidx, ab, 0);
            else
                hipLaunchKernelGGL(bucket_partition2_kernel, dim3((unsigned)((n + BP2_C - 1) / BP2_C), (unsigned)rows),
                                   dim3(BP2_NT), (size_t)BP2_C * 14, s, Y, n, row0, NB, (const double *)spl, bcnt, nnanrow,
                                   ovf, bval, bidx, ab);
            if (!rank_nosort)
                hipLaunchKernelGGL(k_bp, dim3((unsigned)NB, (unsigned)rows), dim3(BK_NT), lds_bk, s, n, NB, (const u32 *)bcnt,
                                   (const u32 *)nnanrow, (const u32 *)ovf, (const double *)bval, (const u32 *)bidx, bflag, ab);
            else
                hipLaunchKernelGGL(k_br, dim3((unsigned)(8 * NB * ((rows + 7) / 8))), dim3(BR_NT), BR_LDS, s, n, rows, NB,
                                   (const u32 *)bcnt, (const u32 *)nnanrow, (const u32 *)ovf,
                                   (const double *)bval, (const u32 *)bidx, bflag, ab);
            hipLaunchKernelGGL(k_bs, dim3(pgrid), dim3(BK_NT), BkCfg::LDS_BYTES, s, Y, n, row0, rows, NB, (const u32 *)bcnt,
                               (const u32 *)nnanrow, (const u32 *)bflag, (const u32 *)nullptr, (const double *)bval,
                               (const u32 *)bidx, (const u32 *)nullptr, 0u, ab);
            fallback_rows = ovf;
            nn_for_fold = nnanrow;
        }
#endif
        // chunked route: every row (cross-check switches)
        if (!merged) {
        const unsigned csgrid = fallback_rows ? pgrid : (unsigned)(p.nch * rows < 65535 * 16 ? p.nch * rows : 65535 * 16);
        hipLaunchKernelGGL(k_cs, dim3(csgrid), dim3(BIG_NT), BigCfg::LDS_BYTES, s, Y, n, row0, rows, p.nch,
                           sorted, p.sstride, nanf, fallback_rows, gate_o, epoch);
        i64 rgroups = cus / p.nch;
        if (rgroups < 1) rgroups = 1;
        if (rgroups > rows) rgroups = rows;
        hipLaunchKernelGGL(k_cq, dim3((unsigned)(rgroups * p.nch)), dim3(BIG_NT), BigCfg::LDS_BYTES, s, Y, n, row0, rows,
                           (const double *)sorted, p.sstride, (const u32 *)nanf, (int)p.nch, fallback_rows, gate_o, epoch, ab);
        }
        SD_HIP(hipGetLastError());
        if (img_out) {
            SD_HIP(hipMemcpyAsync(nnan_out + row0, nn_for_fold, (size_t)rows * 4, hipMemcpyDeviceToDevice, s));
            continue;
        }
        const int first = row0 == 0;
        if (ab.H && !targets && (n & 7) == 0 && (tbegin & 7) == 0 && J <= 3) {
            dim3 grid8((unsigned)((m + 127) / 128));
            if (J == 2) hipLaunchKernelGGL((rank_accumulate_h8_kernel<2>), grid8, dim3(512), 0, s, ab, nn_for_fold, rows, n, tbegin, m, out, first);
            else hipLaunchKernelGGL((rank_accumulate_h8_kernel<3>), grid8, dim3(512), 0, s, ab, nn_for_fold, rows, n, tbegin, m, out, first);
        } else if (!targets && (n & 3) == 0 && (tbegin & 3) == 0 && J <= 3) {
            dim3 grid4((unsigned)((m + 127) / 128));
            if (J == 2) hipLaunchKernelGGL((rank_accumulate2x4_kernel<2>), grid4, dim3(512), 0, s, ab, nn_for_fold, rows, n, tbegin, m, out, first);
            else hipLaunchKernelGGL((rank_accumulate2x4_kernel<3>), grid4, dim3(512), 0, s, ab, nn_for_fold, rows, n, tbegin, m, out, first);
        } else {
            dim3 grid((unsigned)((m + 63) / 64));
            SD_DISPATCH_J(J, hipLaunchKernelGGL((rank_accumulate2_kernel<J_>), grid, dim3(1024), 0, s, ab,
                                                nn_for_fold, rows, n, targets, tbegin, m, out, first));
        }
        SD_HIP(hipGetLastError());
    }
    return SD_OK;
}

}  // namespace sd
