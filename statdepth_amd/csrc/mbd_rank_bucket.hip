// mbd_rank_bucket.hip -- K1+K2 rank formulation without a sort (default path of sd_mbd_counts, n <= 16384, J <= 3).
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
//      A row whose finite values are all equal or that has a bucket of more than CAP keys (heavy ties,
//      clustered data) is left to the sort + search kernel (rank_search_kernel, mbd_rank_ab.hip), which
//      writes that row of the pair image.
//   Z  rank_finalize_kernel -- totals of the requested targets = sum of the workgroups' partials + the fold
//      of the pair-image rows the search kernel produced.
//
// HBM traffic per call: the matrix once (8 nT) + G partials of 8 n (J-1) bytes written and read once.
#include <stdlib.h>

#include "sd_common.h"

namespace sd {

constexpr u32 RB_AB_SPECIAL = 0xFFFFFFFFu;     // same encodings as mbd_rank_ab.hip
constexpr u32 RB_ROW_DEFERRED = 0xFFFFFFFFu;
constexpr int RB_PAD = 8;                      // NaN sentinels behind the bucket-ordered keys (never < or <= anything)
constexpr u32 RB_NOKEY = 0xFFFFFFFFu;          // bs[e] / bc[e]: no curve here, or a NaN

// ---- wave64 cross-lane helpers on DPP (no LDS traffic, no ds_bpermute latency chain) ----
// row_shr:1,2,4,8 scan inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row totals up.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ u32 rb_dpp(u32 old, u32 v) {
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWMASK, 0xF, false);
}

__device__ __forceinline__ u32 rb_wave_incl_scan(u32 v) {
    v += rb_dpp<0x111, 0xF>(0u, v);
    v += rb_dpp<0x112, 0xF>(0u, v);
    v += rb_dpp<0x114, 0xF>(0u, v);
    v += rb_dpp<0x118, 0xF>(0u, v);
    v += rb_dpp<0x142, 0xA>(0u, v);
    v += rb_dpp<0x143, 0xC>(0u, v);
    return v;
}

template <int CTRL, int ROWMASK, bool MAX>
__device__ __forceinline__ double rb_mm_step(double v) {
    const u64 b = (u64)__double_as_longlong(v);
    const u32 l = rb_dpp<CTRL, ROWMASK>((u32)b, (u32)b), h = rb_dpp<CTRL, ROWMASK>((u32)(b >> 32), (u32)(b >> 32));
    const double o = __longlong_as_double((long long)(((u64)h << 32) | l));
    return MAX ? (o > v ? o : v) : (o < v ? o : v);
}

// min (or max) over the wave, valid in lane 63
template <bool MAX>
__device__ __forceinline__ double rb_wave_minmax_last(double v) {
    v = rb_mm_step<0x111, 0xF, MAX>(v);
    v = rb_mm_step<0x112, 0xF, MAX>(v);
    v = rb_mm_step<0x114, 0xF, MAX>(v);
    v = rb_mm_step<0x118, 0xF, MAX>(v);
    v = rb_mm_step<0x142, 0xA, MAX>(v);
    v = rb_mm_step<0x143, 0xC, MAX>(v);
    return v;
}

template <int NT, int E, int LNB>
struct RBCfg {
    static constexpr int NB = 1 << LNB;
    static constexpr int NW = NT / 64;
    static constexpr int W = NB / 2 / NT;                       // histogram words per thread in the prefix sum
    static_assert(W >= 1, "at least one histogram word per thread");
    static __host__ __device__ constexpr size_t keys_slots(int n) { return (size_t)((n + RB_PAD + 1) & ~1); }
    static __host__ __device__ constexpr size_t lds_bytes(int n) {
        return keys_slots(n) * 8 + (size_t)(NB / 2 + 2) * 4 + (size_t)4 * NW * 8 + (size_t)NW * 4 + 16;
    }
};

// DBG (timing experiments only, results invalid): 1 = stop after the range, 2 = after the histogram, 3 = after the
// prefix sum, 4 = after the scatter
template <int NT, int E, int LNB, int J, int CAP, int U, int DBG = 0>
__global__ __launch_bounds__(NT) void rank_bucket_kernel(const double *__restrict__ Y, i64 n64, i64 row0, i64 rows,
                                                         u64 *__restrict__ partial, u32 *__restrict__ nnan_out,
                                                         unsigned char *__restrict__ rowflag) {
    using C = RBCfg<NT, E, LNB>;
    constexpr int NB = C::NB, NW = C::NW, W = C::W;
    static_assert(U >= 1 && U - 1 <= RB_PAD, "member loop reads at most RB_PAD keys past the end");
    static_assert(CAP < 255 && NB <= 32768, "packing of (base, count, slot)");
    constexpr int NACC = (J == 2) ? 1 : (J - 1);
    extern __shared__ double Sm[];
    const int n = (int)n64;
    double *S = Sm;                                                   // keys in bucket order + sentinels
    u32 *H = reinterpret_cast<u32 *>(S + C::keys_slots(n));           // NB packed u16 counters, then bases
    double *red = reinterpret_cast<double *>(H + NB / 2 + 2);         // [2][NW][2] min/max partials
    u32 *wtot = reinterpret_cast<u32 *>(red + 4 * NW);                // [NW]
    const unsigned short *H16 = reinterpret_cast<const unsigned short *>(H);
    const int t0 = threadIdx.x;
    const double INF = __builtin_huge_val();
    int t = t0;

    // one-time LDS setup: sentinels, empty histogram
    if (t < RB_PAD) S[n + t] = __builtin_nan("");
#pragma unroll
    for (int w = 0; w < W; ++w) H[t * W + w] = 0;
    if (t < 2) H[NB / 2 + t] = 0;

    // the host picks E = ceil(n / NT): only the last of a thread's E curves can lie beyond n
    auto valid = [&](int e) { return e < E - 1 || t + (E - 1) * NT < n; };
    double k[E];
    auto load_row = [&](i64 r, double (&dst)[E]) {
        const double *rp = Y + (row0 + r) * n + t;
#pragma unroll
        for (int e = 0; e < E; ++e) dst[e] = valid(e) ? rp[e * NT] : __builtin_nan("");
    };
    // J == 2: acc[e][0] = 2 * (sum of band counts) (one accumulator, see the fold below); else acc[e][j-2]
    u64 acc[E][NACC];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[e][j] = 0;

    if ((i64)blockIdx.x < rows) load_row(blockIdx.x, k);
    int par = 0;
    for (i64 r = blockIdx.x; r < rows; r += gridDim.x) {
        const i64 rnext = r + gridDim.x;
        // Per-row opaque copy of the thread id: every address below derives from it, so the compiler recomputes
        // those few ALU ops per row instead of hoisting loop-invariant address registers out of the row loop
        // and spilling the accumulators to make room (same device as in mbd_rank_ab.hip).
        t = t0;
        asm volatile("" : "+v"(t));
        const int lane = t & 63, wave = t >> 6;
        // ---- (0) range of the finite values of the row (NaN: pandas skipna, _containment.py:68-69) ----
        double mn = INF, mx = -INF;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const double x = k[e];                                    // slots beyond n hold NaN
            const bool fin = __builtin_fabs(x) < INF;
            mn = (fin && x < mn) ? x : mn;
            mx = (fin && x > mx) ? x : mx;
        }
        mn = rb_wave_minmax_last<false>(mn);
        mx = rb_wave_minmax_last<true>(mx);
        double *redp = red + par * 2 * NW;
        if (lane == 63) { redp[2 * wave] = mn; redp[2 * wave + 1] = mx; }
        par ^= 1;
        __syncthreads();                                              // barrier 1 (histogram is zero, S is free)
        double lo = INF, hi = -INF;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const double a = redp[2 * w], b = redp[2 * w + 1];
            lo = a < lo ? a : lo;
            hi = b > hi ? b : hi;
        }
        const double scale = (double)NB / (hi - lo);                  // range overflow -> 0 -> one crowded bucket
        // Every decision below is block-uniform.  The next row is loaded at ONE place (two load sites would
        // keep two copies of the key registers alive across the loop).
        bool go = (hi > lo) && (scale < INF);                         // else: all finite values equal, or none
        if constexpr (DBG == 1) go = false;
        u32 bs[E];
        u32 nn = 0;                                                   // NaN others of this row (block-uniform)
        if (go) {
            // ---- (1) bucket + slot.  clamp(fl(fl(x - lo) * scale)) is non-decreasing in x; -inf and +inf
            //      land in the first and last bucket ----
#pragma unroll
            for (int e = 0; e < E; ++e) {
                bs[e] = RB_NOKEY;
                const double x = k[e];
                if (x == x) {
                    double u = (x - lo) * scale;
                    u = u > 0.0 ? u : 0.0;
                    u = u < (double)(NB - 1) ? u : (double)(NB - 1);
                    const u32 b = (u32)u;
                    const u32 sh = (b & 1u) * 16u;
                    const u32 old = atomicAdd(&H[b >> 1], 1u << sh);
                    bs[e] = b | (((old >> sh) & 0xFFFFu) << 16);
                }
            }
            __syncthreads();                                          // barrier 2
            if constexpr (DBG == 2) {
#pragma unroll
                for (int w = 0; w < W; ++w) H[t * W + w] = 0;
                go = false;
            }
            if constexpr (DBG != 2) {
                // ---- (2) exclusive prefix sum over the counters; a crowded bucket defers the row ----
                u32 hw[W], sum = 0, ov = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    hw[w] = H[t * W + w];
                    sum += hw[w];                                     // both halves at once: no half exceeds 16384
                    ov |= hw[w] + (u32)(0x7FFF - CAP) * 0x10001u;     // bit 15 / 31 set iff that counter > CAP
                }
                const u32 run = (sum & 0xFFFFu) + (sum >> 16);
                const u32 incl = rb_wave_incl_scan(run);
                const bool wover = __ballot((ov & 0x80008000u) != 0) != 0;
                if (lane == 63) wtot[wave] = incl | (wover ? 0x80000000u : 0u);
                __syncthreads();                                      // barrier 3
                u32 base0 = incl - run, anyover = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const u32 x = wtot[w];
                    anyover |= x >> 31;
                    base0 += (w < wave) ? (x & 0x7FFFFFFFu) : 0u;
                }
                go = !anyover;
                // a crowded row leaves zeros behind (the next user is the next row's histogram, behind barrier 1)
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const u32 c0 = hw[w] & 0xFFFFu, c1 = hw[w] >> 16;
                    H[t * W + w] = go ? (base0 | ((base0 + c0) << 16)) : 0u;
                    base0 += c0 + c1;
                }
                if (t == NT - 1) H[NB / 2] = base0;                   // = number of non-NaN keys: base past the last bucket
            }
        }
        if constexpr (DBG == 3) {
            if (go) {
                __syncthreads();
#pragma unroll
                for (int w = 0; w < W; ++w) H[t * W + w] = 0;
            }
            go = false;
        }
        u32 bc[E];
        if (go) {
            __syncthreads();                                          // barrier 4
            // ---- (3) scatter into bucket order ----
            const u32 nv = H[NB / 2];
            nn = (u32)n - nv;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                bc[e] = RB_NOKEY;
                if (bs[e] != RB_NOKEY) {
                    const u32 b = bs[e] & 0xFFFFu, slot = bs[e] >> 16;
                    const u32 base = H16[b], end = H16[b + 1];
                    S[base + slot] = k[e];
                    bc[e] = base | ((end - base) << 16) | (slot << 24);   // base < 2^15, cnt and slot <= CAP < 255
                }
            }
            if (nn && t < RB_PAD) S[nv + t] = __builtin_nan("");    // sentinels behind a row shortened by NaNs
        }
        if (rnext < rows) load_row(rnext, k);                         // next row in flight under the member loop
        if (t == 0) { nnan_out[r] = (go || DBG) ? 0u : RB_ROW_DEFERRED; rowflag[r] = (go || DBG) ? 0 : 1; }
        if constexpr (DBG == 4) {
            if (go) {
                __syncthreads();
#pragma unroll
                for (int w = 0; w < W; ++w) H[t * W + w] = 0;
            }
            go = false;
        }
        if (go) {
            __syncthreads();                                          // barrier 5
            // the histogram is dead until the next row's atomics (behind its barrier 1)
#pragma unroll
            for (int w = 0; w < W; ++w) H[t * W + w] = 0;
            // ---- (4) rank inside the bucket, fold ----
            const u32 nv = (u32)n - nn;                               // non-NaN keys of the row
            const u32 v = nv - 1u;                                    // valid others of a non-NaN target
            const u32 R2 = v * (v - 1u + 2u * nn);                    // J == 2: 2 * [N v + C(v,2)]  (< 2^30)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (bc[e] != RB_NOKEY) {
                    const u32 base = bc[e] & 0xFFFFu, cnt = (bc[e] >> 16) & 0xFFu;
                    const double *Sp = S + base;
                    const double x = Sp[bc[e] >> 24];                 // own key (its registers hold the next row)
                    u32 less = 0, le = 0;
#pragma unroll 1
                    for (u32 kk = 0; kk < cnt; kk += U) {
                        double y[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) y[u] = Sp[kk + u];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            less += (y[u] < x) ? 1u : 0u;
                            le += (y[u] <= x) ? 1u : 0u;
                        }
                    }
                    const u32 B = base + less, A = nv - base - le;
                    if constexpr (J == 2) {
                        // 2 * contained_2 = 2 N (v - A - B) + v(v-1) - A(A-1) - B(B-1)   (all terms < 2^30)
                        u32 q = A * (A - 1u) + B * (B - 1u);
                        if (nn) q += 2u * nn * (A + B);
                        acc[e][0] += (u64)(R2 - q);
                    } else {
                        u64 a7[JMAX - 1] = {0, 0, 0, 0, 0, 0, 0};
                        band_counts_add<J>(A, B, nn, (u64)(n - 1), a7);
#pragma unroll
                        for (int j = 0; j < J - 1; ++j) acc[e][j] += a7[j];
                    }
                }
            }
        }
    }
    t = t0;
    // ---- this workgroup's partial totals ----
    u64 *P = partial + (size_t)blockIdx.x * (J - 1) * n;
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (valid(e)) {
#pragma unroll
            for (int j = 0; j < J - 1; ++j) P[(size_t)j * n + t + e * NT] = (J == 2) ? (acc[e][0] >> 1) : acc[e][j];
        }
}

// ---------------------------------------------------------------------------------------------------
// Z: out[q][j] (+)= sum_g partial[g][j][i(q)] + fold of the pair-image rows flagged in rowflag.
// block = 32 targets x 32 slices (of workgroups g, of rows), LDS tree over the slices.
// ---------------------------------------------------------------------------------------------------
template <int J>
__global__ __launch_bounds__(1024) void rank_finalize_kernel(const u64 *__restrict__ partial, int G,
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
    for (i64 r = threadIdx.x; r < rows; r += 1024) any |= rowflag[r];
    any = __syncthreads_or(any);
    u64 acc[JMAX - 1];
#pragma unroll
    for (int j = 0; j < JMAX - 1; ++j) acc[j] = 0;
    if (q < m) {
        const u64 *p = partial + i;
#pragma unroll 8
        for (int g = y; g < G; g += 32) {
#pragma unroll
            for (int j = 0; j < J - 1; ++j) acc[j] += p[((size_t)g * (J - 1) + j) * n];
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

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static int rb_cus() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;
}

bool mbd_rank_bucket_supported(i64 T, i64 n, int J) {
    (void)T;
    return n > 8192 && n <= 16384 && J >= 2 && J <= 3;
}

// upper bound of the grid the launcher will use (the partial totals are sized by it)
int mbd_rank_bucket_max_grid() { return rb_cus(); }

size_t mbd_rank_bucket_workspace_bytes(i64 rows, i64 n, int J) {
    return align_up((size_t)mbd_rank_bucket_max_grid() * (J - 1) * n * 8, 256) + align_up((size_t)rows, 256) + 512;
}

template <int NT, int E, int LNB, int J>
static int launch_bucket_cfg(const double *Y, i64 n, i64 row0, i64 rows, u64 *partial, u32 *nnan, unsigned char *rowflag,
                             int G, hipStream_t s) {
    using C = RBCfg<NT, E, LNB>;
    auto kf = rank_bucket_kernel<NT, E, LNB, J, 40, 4>;
    if constexpr (E == 10 && J == 2) {
        if (const char *d = getenv("SD_RB_DBG")) {
            switch (atoi(d)) {
                case 1: kf = rank_bucket_kernel<NT, E, LNB, J, 40, 4, 1>; break;
                case 2: kf = rank_bucket_kernel<NT, E, LNB, J, 40, 4, 2>; break;
                case 3: kf = rank_bucket_kernel<NT, E, LNB, J, 40, 4, 3>; break;
                case 4: kf = rank_bucket_kernel<NT, E, LNB, J, 40, 4, 4>; break;
            }
        }
    }
    const size_t lds = C::lds_bytes((int)n);
    if (lds > 163840) return fail(SD_ERR_UNSUPPORTED, "bucket kernel: %zu bytes of LDS for n=%lld", lds, (long long)n);
    SD_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kf, dim3(G), dim3(NT), lds, s, Y, n, row0, rows, partial, nnan, rowflag);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

template <int J>
static int launch_bucket_j(const double *Y, i64 n, i64 row0, i64 rows, u64 *partial, u32 *nnan, unsigned char *rowflag,
                           int G, hipStream_t s) {
    // E = ceil(n / 1024); 16384 buckets while keys + histogram fit the 160 KiB of LDS, 8192 above
    switch ((int)((n + 1023) / 1024)) {
        case 9: return launch_bucket_cfg<1024, 9, 14, J>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
        case 10: return launch_bucket_cfg<1024, 10, 14, J>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
        case 11: return launch_bucket_cfg<1024, 11, 14, J>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
        case 12: return launch_bucket_cfg<1024, 12, 14, J>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
        case 13: return launch_bucket_cfg<1024, 13, 14, J>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
        case 14: return launch_bucket_cfg<1024, 14, 14, J>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
        case 15: return launch_bucket_cfg<1024, 15, 14, J>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
        case 16: return launch_bucket_cfg<1024, 16, 13, J>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
    }
    return fail(SD_ERR_UNSUPPORTED, "bucket kernel covers 8192 < n <= 16384");
}

// rows [row0, row0 + rows): bucket kernel; returns the grid used (number of partial blocks) in *G_out
int launch_rank_bucket(const double *Y, i64 n, i64 row0, i64 rows, int J, u64 *partial, u32 *nnan,
                       unsigned char *rowflag, int *G_out, hipStream_t s) {
    const int cus = mbd_rank_bucket_max_grid();
    const int G = (int)(rows < cus ? rows : cus);
    *G_out = G;
    if (J == 2) return launch_bucket_j<2>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
    if (J == 3) return launch_bucket_j<3>(Y, n, row0, rows, partial, nnan, rowflag, G, s);
    return fail(SD_ERR_UNSUPPORTED, "bucket kernel covers J in [2,3]");
}

int launch_rank_finalize(const u64 *partial, int G, const u32 *AB, const u32 *nnan, const unsigned char *rowflag,
                         i64 rows, i64 n, const i64 *targets, i64 tbegin, i64 m, int J, u64 *out, int first,
                         hipStream_t s) {
    dim3 grid((unsigned)((m + 31) / 32));
    if (J == 2)
        hipLaunchKernelGGL((rank_finalize_kernel<2>), grid, dim3(1024), 0, s, partial, G, AB, nnan, rowflag, rows, n, targets,
                           tbegin, m, out, first);
    else
        hipLaunchKernelGGL((rank_finalize_kernel<3>), grid, dim3(1024), 0, s, partial, G, AB, nnan, rowflag, rows, n, targets,
                           tbegin, m, out, first);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

}  // namespace sd
