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
//      Keys past the end of a bucket belong to later buckets and compare greater, so the member loop needs
//      no bounds.  Ties are exact by construction (equal values share a bucket).  The per-curve band counts
//      are accumulated in REGISTERS across the workgroup's rows; one partial total per (workgroup, curve)
//      goes to HBM at the end: the matrix is read once and nothing per (row, curve) is ever written.
//      A row with a NaN or an infinity, a degenerate range or a bucket of more than CAP keys (heavy ties,
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
constexpr int RB_PAD = 8;                      // +inf sentinels behind the bucket-ordered keys

__device__ __forceinline__ u32 rb_wave_incl_scan(u32 v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
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

template <int NT, int E, int LNB, int J, int CAP, int U>
__global__ __launch_bounds__(NT) void rank_bucket_kernel(const double *__restrict__ Y, i64 n64, i64 row0, i64 rows,
                                                         u64 *__restrict__ partial, u32 *__restrict__ nnan_out,
                                                         unsigned char *__restrict__ rowflag) {
    using C = RBCfg<NT, E, LNB>;
    constexpr int NB = C::NB, NW = C::NW, W = C::W;
    static_assert(U >= 1 && U - 1 <= RB_PAD, "member loop reads at most RB_PAD keys past the end");
    static_assert(CAP < 256 && NB <= 32768, "packing of (base, count, slot)");
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
    if (t < RB_PAD) S[n + t] = INF;
#pragma unroll
    for (int w = 0; w < W; ++w) H[t * W + w] = 0;
    if (t < 2) H[NB / 2 + t] = 0;

    // the host picks E = ceil(n / NT): only the last of a thread's E curves can lie beyond n
    auto valid = [&](int e) { return e < E - 1 || t + (E - 1) * NT < n; };
    double k[E];
    auto load_row = [&](i64 r, double (&dst)[E]) {
        const double *rp = Y + (row0 + r) * n + t;
#pragma unroll
        for (int e = 0; e < E; ++e) dst[e] = valid(e) ? rp[e * NT] : 0.0;
    };
    u64 acc[E][JMAX - 1];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int j = 0; j < JMAX - 1; ++j) acc[e][j] = 0;

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
        // ---- (0) range of the row; non-finite values send the row to the search kernel ----
        double mn = INF, mx = -INF;
        int bad = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (valid(e)) {
                const double x = k[e];
                bad |= !(__builtin_fabs(x) < INF);                    // NaN or +-inf
                mn = x < mn ? x : mn;
                mx = x > mx ? x : mx;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        double *redp = red + par * 2 * NW;
        if (lane == 0) { redp[2 * wave] = mn; redp[2 * wave + 1] = mx; }
        par ^= 1;
        const int anybad = __syncthreads_or(bad);                     // barrier 1 (histogram is zero, S is free)
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
        bool go = !(anybad || !(hi > lo) || !(scale < INF));
        u32 bs[E];
        if (go) {
            // ---- (1) bucket + slot ----
#pragma unroll
            for (int e = 0; e < E; ++e) {
                bs[e] = 0;
                if (valid(e)) {
                    const double u = (k[e] - lo) * scale;             // finite, >= 0, monotone in k[e]
                    const u32 b = (u32)(u < (double)(NB - 1) ? u : (double)(NB - 1));
                    const u32 sh = (b & 1u) * 16u;
                    const u32 old = atomicAdd(&H[b >> 1], 1u << sh);
                    bs[e] = b | (((old >> sh) & 0xFFFFu) << 16);
                }
            }
            __syncthreads();                                          // barrier 2
            // ---- (2) exclusive prefix sum over the counters; crowded bucket -> defer ----
            u32 ex[2 * W], run = 0;
            int over = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const u32 x = H[t * W + w];
                const u32 c0 = x & 0xFFFFu, c1 = x >> 16;
                over |= (c0 > (u32)CAP) | (c1 > (u32)CAP);
                ex[2 * w] = run;
                run += c0;
                ex[2 * w + 1] = run;
                run += c1;
            }
            const u32 incl = rb_wave_incl_scan(run, lane);
            if (lane == 63) wtot[wave] = incl;
            go = !__syncthreads_or(over);                             // barrier 3
            u32 base0 = incl - run;
#pragma unroll
            for (int w = 0; w < NW; ++w) base0 += (w < wave) ? wtot[w] : 0u;
            // a crowded row leaves zeros behind (its next user is the next row's histogram, behind barrier 1)
#pragma unroll
            for (int w = 0; w < W; ++w)
                H[t * W + w] = go ? ((base0 + ex[2 * w]) | ((base0 + ex[2 * w + 1]) << 16)) : 0u;
            if (t == NT - 1) H[NB / 2] = (u32)n;                      // base of the bucket past the last one
        }
        u32 bc[E];
        if (go) {
            __syncthreads();                                          // barrier 4
            // ---- (3) scatter into bucket order ----
#pragma unroll
            for (int e = 0; e < E; ++e) {
                bc[e] = 0;
                if (valid(e)) {
                    const u32 b = bs[e] & 0xFFFFu, slot = bs[e] >> 16;
                    const u32 base = H16[b], end = H16[b + 1];
                    S[base + slot] = k[e];
                    bc[e] = base | ((end - base) << 16) | (slot << 24);   // base < 2^15, cnt and slot <= CAP < 2^8
                }
            }
        }
        if (rnext < rows) load_row(rnext, k);                         // next row in flight under the member loop
        if (t == 0) { nnan_out[r] = go ? 0u : RB_ROW_DEFERRED; rowflag[r] = go ? 0 : 1; }
        if (go) {
            __syncthreads();                                          // barrier 5
            // the histogram is dead until the next row's atomics (behind its barrier 1)
#pragma unroll
            for (int w = 0; w < W; ++w) H[t * W + w] = 0;
            // ---- (4) rank inside the bucket, fold ----
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (valid(e)) {
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
                    const u32 B = base + less, A = (u32)n - base - le;
                    band_counts_add<J>(A, B, 0u, (u64)(n - 1), acc[e]);
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
            for (int j = 0; j < J - 1; ++j) P[(size_t)j * n + t + e * NT] = acc[e][j];
        }
}

// ---------------------------------------------------------------------------------------------------
// Z: out[q][j] (+)= sum_g partial[g][j][i(q)] + fold of the pair-image rows flagged in rowflag.
// block = 64 targets x 16 slices, LDS tree over the slices.
// ---------------------------------------------------------------------------------------------------
template <int J>
__global__ __launch_bounds__(1024) void rank_finalize_kernel(const u64 *__restrict__ partial, int G,
                                                             const u32 *__restrict__ AB, const u32 *__restrict__ nnan,
                                                             const unsigned char *__restrict__ rowflag, i64 rows, i64 n,
                                                             const i64 *__restrict__ targets, i64 tbegin, i64 m,
                                                             u64 *__restrict__ out, int first) {
    __shared__ u64 red[16][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const i64 q = (i64)blockIdx.x * 64 + x;
    const i64 i = (q < m) ? (targets ? targets[q] : tbegin + q) : 0;
    u64 acc[JMAX - 1];
#pragma unroll
    for (int j = 0; j < JMAX - 1; ++j) acc[j] = 0;
    if (q < m) {
        for (int g = y; g < G; g += 16) {
#pragma unroll
            for (int j = 0; j < J - 1; ++j) acc[j] += partial[((size_t)g * (J - 1) + j) * n + i];
        }
        for (i64 r = y; r < rows; r += 16) {
            if (rowflag[r]) {
                const u32 ab = AB[r * n + i];
                if (ab != RB_AB_SPECIAL) band_counts_add<J>(ab >> 16, ab & 0xFFFFu, nnan[r], (u64)(n - 1), acc);
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
            for (int k = 0; k < 16; ++k) tot += red[k][x];
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
    dim3 grid((unsigned)((m + 63) / 64));
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
