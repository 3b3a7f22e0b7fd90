// mbd_rank_ab.hip -- K1+K2 rank formulation for n <= 16384: the launcher of the rank path and the kernels that fold a
// pair image.
//
// Same integers as the pairwise kernel and the reference's enumeration (_functional.py:246-251,
// _containment.py:75-77): per (curve, timepoint) the counts B (others strictly below) and A (strictly above), and
// C(v,j) - C(A,j) - C(B,j) summed over t.
//
// In the product library the rows are ranked by rank_bucket_kernel (mbd_rank_bucket.hip): J <= 3 folds in registers,
// J >= 4 writes the pairs (B, A) of every (row, curve) as a uint16 pair image and
//  C  rank_accumulate_kernel / rank_accumulate4_kernel -- fold the pairs and the per-row NaN counts into the int64
//     totals of the requested targets.
//
// -DSD_CROSSCHECK builds (libstatdepth_hip_xcheck.so, loaded by the tests only) also carry the sort-based predecessors
// the bucket kernel replaced, as independent implementations to compare against (SD_RANK_IMPL = 3, 2, 1):
//  A  rank_packed_kernel  -- one workgroup sorts one row in LDS (rank_sort.h).  The curve index rides in
//     the low log2(N) mantissa bits of the fp64 key, so v_min_f64 / v_max_f64 sort value and owner
//     together and the holder of sorted position p knows which curve sits there: rank = p, handed to the
//     owner through LDS, written out as (B, A) pairs of uint16.  Exact unless two keys agree above the
//     index field (values within ~2^-38 relative, incl. exact ties): such a row is flagged for kernel B
//     instead, and a workgroup that met one leaves the rest of its rows to kernel B directly.  NaN / +-inf /
//     padding map to sentinel classes beyond every finite class; finite values that would fall into a
//     sentinel class or (non-zero) into the zero class are left to kernel B as well.
//  B  rank_search_kernel  -- the flagged rows: sort of the plain values, then every curve binary-searches
//     its own value (lower bound = B, upper bound gives A; ties are exact by construction).
#include <stdlib.h>

#include "sd_common.h"
#include "rank_sort.h"

namespace sd {

constexpr u32 AB_SPECIAL = 0xFFFFFFFFu;      // the curve is NaN at this timepoint: contributes nothing

#ifdef SD_CROSSCHECK
constexpr u32 ROW_DEFERRED = 0xFFFFFFFFu;    // nnan_out[r]: the packed kernel left row r to the search kernel

template <int NT, int E>
struct PKeys {
    using C = R2Cfg<NT, E>;
    static constexpr int LN = C::LN;
    static constexpr u64 MASK = (u64)C::N - 1;                 // index field
    static constexpr u64 TOPM = ((0xFFFFFFFFFFFFFull >> LN) << LN);
    static constexpr u64 H3 = (0x7FEull << 52) | TOPM;         // padding class (largest)
    static constexpr u64 H2 = H3 - ((u64)1 << LN);             // NaN class
    static constexpr u64 H1 = H3 - ((u64)2 << LN);             // +inf class
    static constexpr u64 SIGN = 0x8000000000000000ull;
    static constexpr u64 LOW = (u64)1 << LN;                   // magnitudes below this share the zero class
};

__device__ __forceinline__ u64 pk_bits(double v) { return (u64)__double_as_longlong(v); }
__device__ __forceinline__ double pk_dbl(u64 b) { return __longlong_as_double((long long)b); }

// ---------------------------------------------------------------------------------------------------
// A: packed keys, rank = position.  Rows [row0, row0 + rows) of Y; AB and nnan are indexed by row - row0.
// ---------------------------------------------------------------------------------------------------
template <int NT, int E, int DBG = 0>
__global__ __launch_bounds__(NT) void rank_packed_kernel(const double *__restrict__ Y, i64 n64, i64 row0, i64 rows,
                                                         u32 *__restrict__ AB, u32 *__restrict__ nnan_out) {
    using C = R2Cfg<NT, E>;
    using K = PKeys<NT, E>;
    using Sorter = R2Sorter<NT, E>;
    constexpr int LN = C::LN, WB = C::WB;
    constexpr u64 MASK = K::MASK;
    constexpr u64 CLS_NAN = K::H2 >> LN, CLS_PAD = K::H3 >> LN;
    extern __shared__ double Sm[];
    double *firstkey = Sm + C::SLOTS;                          // NT doubles behind the sort image
    __shared__ u32 s_nnan[2];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int n = (int)n64;
    const int n_act = ((n + WB - 1) / WB) * WB;
    const bool wreal = wave * WB < n_act;
    const double INF = __builtin_huge_val();
    const double MAXK = pk_dbl(K::H3 | MASK);
    if (t < 2) s_nnan[t] = 0;

    const int i0 = wave * WB + lane;                           // loaded curves: i0 + 64 e (512 B per wave instruction)
    double k[E];
    auto load_row = [&](i64 r) {
        const double *rp = Y + (row0 + r) * n + i0;
#pragma unroll
        for (int e = 0; e < E; ++e) k[e] = (i0 + e * 64 < n) ? rp[e * 64] : INF;
    };
    if (wreal && (i64)blockIdx.x < rows) load_row(blockIdx.x);
    __syncthreads();

    int par = 0;
    bool defer = false;                     // after one listed row this workgroup stops trying the packed path
    for (i64 r = blockIdx.x; r < rows; r += gridDim.x) {
        if (defer) {
            if (t == 0) nnan_out[r] = ROW_DEFERRED;
            continue;
        }
        // Per-row opaque copy of the thread id: every LDS address below derives from it, so the compiler
        // recomputes those few ALU ops per row instead of hoisting ~45 loop-invariant address registers out
        // of the row loop and spilling them (measured: 190 B/lane of scratch, +95 MB of HBM traffic per launch).
        int tv = t;
        asm volatile("" : "+v"(tv));
        // ---- pack: value bits above the index field | curve index ----
        int forcefull = 0;
        u32 mynan = 0;
        if (wreal) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = i0 + e * 64;
                const u64 b = pk_bits(k[e]);
                const u64 a = b & ~K::SIGN;
                u64 kb = b & ~MASK;
                // one unsigned range test flags zero class, sentinel classes, +-inf and NaN
                if (__builtin_expect((a - K::LOW) >= (K::H1 - K::LOW), 0)) {
                    if (a > 0x7FF0000000000000ull) { kb = K::H2; mynan += (i < n); }
                    else if (a == 0x7FF0000000000000ull) kb = (b & K::SIGN) ? (K::SIGN | K::H3) : K::H1;
                    else if (a == 0) kb = 0;                                   // -0 -> +0
                    else forcefull |= (i < n);         // finite value inside a sentinel / the zero class
                }
                kb = (i < n) ? kb : K::H3;
                k[e] = pk_dbl(kb | (u64)i);
            }
        }
        if (mynan) atomicAdd(&s_nnan[par], mynan);
        if constexpr (!(DBG & 2)) Sorter::sort(k, Sm, tv, n_act, wreal, MAXK);

        // ---- any class with several members?  (holders of the sorted positions, layout 0: p = t*E + e) ----
        if (wreal) firstkey[tv] = k[0];
        __syncthreads();
        if (t == 0) s_nnan[par ^ 1] = 0;                       // nobody touches the other parity during this row
        const u32 nnan = s_nnan[par];
        par ^= 1;
        int anytie = 0;
        if (wreal) {
            u64 nextb = ~0ull;
            if ((tv + 1) * E < n_act) nextb = pk_bits(firstkey[tv + 1]);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const u64 c0 = pk_bits(k[e]) >> LN;
                const u64 c1 = ((e < E - 1) ? pk_bits(k[e + 1]) : nextb) >> LN;
                anytie |= (c0 == c1) & (c0 != CLS_NAN) & (c0 != CLS_PAD);
            }
        }
        const int mode = __syncthreads_or(anytie | forcefull);
        const i64 rnext = r + gridDim.x;
        if (mode && !(DBG & 1)) {
            if (t == 0) nnan_out[r] = ROW_DEFERRED;
            defer = true;
            continue;
        }
        // ---- ranks are positions: scatter to the owners' slots (the sort image is dead) ----
        const u32 nreal = (u32)n - nnan;                       // non-NaN values occupy sorted positions [0, nreal)
        u32 *R = reinterpret_cast<u32 *>(Sm);
        if (wreal) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const u64 b = pk_bits(k[e]);
                const int idx = (int)(b & MASK);
                const u32 p = (u32)(tv * E + e);
                if (idx < n) R[idx] = ((b >> LN) == CLS_NAN) ? AB_SPECIAL : (p | ((nreal - 1u - p) << 16));
            }
        }
        __syncthreads();
        if (wreal && rnext < rows) load_row(rnext);            // key registers are free: next row in flight
        u32 *dst = AB + r * n + tv;                            // coalesced: curve t + e*NT
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (tv + e * NT < n) dst[e * NT] = R[tv + e * NT];
        if (t == 0) nnan_out[r] = nnan;
        __syncthreads();                                       // LDS is reused by the next row
    }
}

// ---------------------------------------------------------------------------------------------------
// B: plain values + search.  only_deferred == 0: every row of [row0, row0 + rows); otherwise only the rows
// the packed kernel marked ROW_DEFERRED -- workgroup g of this kernel looks at exactly the rows workgroup g
// of the packed kernel owned (same grid), so no list, counter or memset is needed.
// ---------------------------------------------------------------------------------------------------
template <int NT, int E, int DBG = 0>
__global__ __launch_bounds__(NT) void rank_search_kernel(const double *__restrict__ Y, i64 n64, i64 row0, i64 rows,
                                                         u32 *__restrict__ AB, u32 *__restrict__ nnan_out,
                                                         int only_deferred) {
    using C = R2Cfg<NT, E>;
    using Sorter = R2Sorter<NT, E>;
    constexpr int N = C::N, LE = C::LE, WB = C::WB;
    extern __shared__ double Sm[];
    __shared__ u32 s_nnan[2];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int n = (int)n64;
    const int n_act = ((n + WB - 1) / WB) * WB;
    const bool wreal = wave * WB < n_act;
    const double INF = __builtin_huge_val();
    if (t < 2) s_nnan[t] = 0;

    const int i0 = wave * WB + lane;
    double k[E];
    auto load_row = [&](i64 r) {
        const double *rp = Y + (row0 + r) * n + i0;
#pragma unroll
        for (int e = 0; e < E; ++e) k[e] = (i0 + e * 64 < n) ? rp[e * 64] : INF;
    };
    __syncthreads();

    int par = 0;
    for (i64 r = blockIdx.x; r < rows; r += gridDim.x) {
        if (only_deferred && nnan_out[r] != ROW_DEFERRED) continue;     // block-uniform
        const double *__restrict__ row = Y + (row0 + r) * n;
        if (wreal) load_row(r);
        int tv = t;                         // per-row opaque copy: keeps LDS addresses out of loop-invariant spills
        asm volatile("" : "+v"(tv));
        // NaN -> +inf, counted (pandas skipna, _containment.py:68-69)
        u32 mynan = 0;
        if (wreal) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                bool isn = k[e] != k[e];
                mynan += isn ? 1u : 0u;
                k[e] = isn ? INF : k[e];
            }
        }
        if (mynan) atomicAdd(&s_nnan[par], mynan);
        if constexpr (!(DBG & 2)) Sorter::sort(k, Sm, tv, n_act, wreal, INF);
        if (wreal) {
            double *Sw = Sm + r2_base<0, LE>(tv);
#pragma unroll
            for (int e = 0; e < E; ++e) Sw[r2_off<0, LE>(e)] = k[e];
        }
        __syncthreads();
        if (t == 0) s_nnan[par ^ 1] = 0;    // nobody touches the other parity during this row
        const u32 nnan = s_nnan[par];
        par ^= 1;
        // every wave searches (curve t + e*NT belongs to thread t) although only the waves below n_act
        // sorted: the search is a chain of dependent LDS reads and needs all the parallelism it can get
        if (!(DBG & 1)) {
            const double *xp = row + t;
            u32 *dst = AB + r * n + t;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if ((e & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // at most 4 searches in flight (VGPRs)
                if (t + e * NT < n) {
                    double x = xp[e * NT];
                    u32 ab = AB_SPECIAL;
                    if (x == x) {
                        int lo = r2_bound<N, SlotPad<LE>, false, false>(Sm, n_act, x, INF);   // x is in the row
                        // upper bound: x sits at lo; gallop over its tie run (1 probe if untied, ~2 log2(run) otherwise)
                        int hi = lo + 1, step = 1;
                        while (hi + step <= n_act && Sm[r2_phys<LE>(hi + step - 1)] <= x) { hi += step; step <<= 1; }
                        while (step > 1) {
                            step >>= 1;
                            if (hi + step <= n_act && Sm[r2_phys<LE>(hi + step - 1)] <= x) hi += step;
                        }
                        // keys <= x within [0, n_act) are real non-NaN values unless x = +inf
                        u32 A = (x == INF) ? 0u : (u32)(n - hi) - nnan;
                        ab = (u32)lo | (A << 16);
                    }
                    dst[e * NT] = ab;
                }
            }
        }
        __syncthreads();                    // LDS is reused by the next row; every thread has read the flag
        if (t == 0) nnan_out[r] = nnan;
    }
}
#endif  // SD_CROSSCHECK (retired sort kernels)


// ---------------------------------------------------------------------------------------------------
// C: out[q][j] += sum over the batch's rows of the band counts of target q.
// block = 64 targets x 16 row slices, LDS tree over the slices.
// ---------------------------------------------------------------------------------------------------
template <int J>
__global__ __launch_bounds__(1024) void rank_accumulate_kernel(const u32 *__restrict__ AB, const u32 *__restrict__ nnan,
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
        // 8 independent loads in flight per thread: this kernel is a pure stream over the pair image
        i64 r = y;
        for (; r + 16 * 7 < rows; r += 16 * 8) {
            u32 ab[8], nn[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ab[u] = AB[(r + 16 * u) * n + i];
                nn[u] = nnan[r + 16 * u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (ab[u] != AB_SPECIAL) band_counts_add<J>(ab[u] >> 16, ab[u] & 0xFFFFu, nn[u], (u64)(n - 1), acc);
        }
        for (; r < rows; r += 16) {
            const u32 ab = AB[r * n + i];
            if (ab != AB_SPECIAL) band_counts_add<J>(ab >> 16, ab & 0xFFFFu, nnan[r], (u64)(n - 1), acc);
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

// Same fold for a contiguous, 4-aligned target block: 16-byte loads (4 curves per lane, 1 KiB per wave
// instruction instead of 256 B).  block = 16 curve quads x 64 row slices.
template <int J>
__global__ __launch_bounds__(1024) void rank_accumulate4_kernel(const u32 *__restrict__ AB, const u32 *__restrict__ nnan,
                                                                i64 rows, i64 n, i64 tbegin, i64 m,
                                                                u64 *__restrict__ out, int first) {
    __shared__ u64 red[64][65];
    const int x = threadIdx.x & 15, y = threadIdx.x >> 4;            // quad within block, row slice
    const i64 q4 = ((i64)blockIdx.x * 16 + x) * 4;                   // first of this thread's 4 targets
    const bool live = q4 < m;
    u64 acc[4][JMAX - 1];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < JMAX - 1; ++j) acc[c][j] = 0;
    if (live) {
        const u32 *src = AB + tbegin + q4;
        i64 r = y;
        for (; r + 64 * 3 < rows; r += 64 * 4) {
            uint4 v[4];
            u32 nn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = *reinterpret_cast<const uint4 *>(src + (r + 64 * u) * n);
                nn[u] = nnan[r + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 ab[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (ab[c] != AB_SPECIAL) band_counts_add<J>(ab[c] >> 16, ab[c] & 0xFFFFu, nn[u], (u64)(n - 1), acc[c]);
            }
        }
        for (; r < rows; r += 64) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + r * n);
            const u32 nn = nnan[r];
            const u32 ab[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (ab[c] != AB_SPECIAL) band_counts_add<J>(ab[c] >> 16, ab[c] & 0xFFFFu, nn, (u64)(n - 1), acc[c]);
        }
    }
#pragma unroll
    for (int j = 0; j < J - 1; ++j) {
#pragma unroll
        for (int c = 0; c < 4; ++c) red[y][x * 4 + c] = acc[c][j];
        __syncthreads();
        if (threadIdx.x < 64) {
            const i64 q = (i64)blockIdx.x * 64 + threadIdx.x;
            if (q < m) {
                u64 tot = 0;
                for (int k = 0; k < 64; ++k) tot += red[k][threadIdx.x];
                if (first) out[q * (J - 1) + j] = tot;
                else out[q * (J - 1) + j] += tot;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
#ifdef SD_CROSSCHECK
static int ab_grid(i64 rows, int per_cu = 1) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    cus *= per_cu;
    return (int)(rows < cus ? rows : cus);
}
#endif

static i64 ab_rows_per_batch(i64 T, i64 n) {
    i64 r = ((i64)1 << 30) / (n * 4);       // pair image <= 1 GiB
    if (r > 65536) r = 65536;               // and the bucket kernel's per-workgroup bitmap of set-aside rows (2048 bits)
    const i64 v = xswitch("SD_RANK_ROWS_PER_BATCH");   // cross-check builds: force several batches on small inputs
    if (v > 0 && v < r) r = v;
    if (r < 1) r = 1;
    return r < T ? r : T;
}

// mbd_rank_bucket.hip
bool mbd_rank_bucket_supported(i64 T, i64 n, int J);
int mbd_rank_bucket_max_grid();
size_t mbd_rank_bucket_partial_bytes(i64 n, int J);
size_t mbd_rank_bucket_workspace_bytes(i64 rows, i64 n, int J);
int launch_rank_bucket(const double *Y, i64 n, i64 row0, i64 rows, int J, u64 *partial, int *p32_out, int *G_out,
                       hipStream_t s);
int launch_rank_bucket_image(const double *Y, i64 n, i64 row0, i64 rows, u32 *AB, u32 *nnan, hipStream_t s);
bool rank_bucket_two_level_supported(i64 n, i64 rows);
int launch_rank_bucket_two_level(const double *Y, i64 n, i64 row0, i64 rows, u64 *partial, u64 *out, int first, hipStream_t s);
u64 *rank_bucket_two_level_all_totals(u64 *partial, i64 n);
int launch_rank_gather_totals(const u64 *all, const i64 *targets, i64 tbegin, i64 m, u64 *out, hipStream_t s);
int launch_rank_finalize(const u64 *partial, int G, int p32, const u32 *AB, const u32 *nnan, const unsigned char *rowflag,
                         i64 rows, i64 n, const i64 *targets, i64 tbegin, i64 m, int J, u64 *out, int first,
                         hipStream_t s);

bool mbd_rank_supported(i64 T, i64 n, int J) {
    (void)T;
    return n >= 2 && n <= 16384 && J >= 2 && J <= JMAX;
}

// Which implementation ranks the rows: 4 = bucket kernel (J <= 3), 5 = bucket kernel in pair-image mode +
// fold (J >= 4).  In -DSD_CROSSCHECK builds SD_RANK_IMPL selects the sort-based predecessors 3, 2, 1.
static int rank_impl(i64 T, i64 n, int J) {
    const int forced = (int)xswitch("SD_RANK_IMPL");
    int impl = forced ? forced : 4;
    if (impl == 4 && !mbd_rank_bucket_supported(T, n, J)) impl = (J >= 4) ? 5 : 3;
    return impl;
}

size_t mbd_rank_workspace_bytes(i64 T, i64 n, int J) {
    if (!mbd_rank_supported(T, n, J)) return 0;
    const i64 rpb = ab_rows_per_batch(T, n);
    const int impl = rank_impl(T, n, J);
    if (impl == 4) return mbd_rank_bucket_workspace_bytes(rpb, n, J);      // no pair image on this path
    size_t need = align_up((size_t)rpb * n * 4, 256) + 2 * align_up((size_t)rpb * 4, 256) + 512;
    size_t v1 = (size_t)(T < 1024 ? T : 1024) * (J <= 3 ? J - 1 : 0) * n * 8;   // first-generation kernel's partial sums
    return need > v1 ? need : v1;
}

#ifdef SD_CROSSCHECK
template <int NT, int E>
static int launch_sorts(const double *Y, i64 n, i64 row0, i64 rows, u32 *AB, u32 *nnan, int impl, hipStream_t s) {
    using C = R2Cfg<NT, E>;
    // workgroups per CU: limited by LDS (160 KiB) and by 16 waves per CU at up to 128 VGPRs per lane
    constexpr int BY_LDS = (int)(163840 / (C::LDS_BYTES + 512)), BY_WAVES = 1024 / NT;
    constexpr int PER_CU = BY_LDS < BY_WAVES ? (BY_LDS < 1 ? 1 : BY_LDS) : BY_WAVES;
    const int G = ab_grid(rows, PER_CU);
    auto ks = rank_search_kernel<NT, E>;
    SD_HIP(hipFuncSetAttribute((const void *)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    if (impl == 2) {   // full keys + search for every row (A/B timing, cross-check)
        hipLaunchKernelGGL(ks, dim3(G), dim3(NT), C::LDS_BYTES, s, Y, n, row0, rows, AB, nnan, 0);
        SD_HIP(hipGetLastError());
        return SD_OK;
    }
    if (impl == 4) {   // the bucket kernel ranked the rows; only those it deferred are sorted here
        hipLaunchKernelGGL(ks, dim3(G), dim3(NT), C::LDS_BYTES, s, Y, n, row0, rows, AB, nnan, 1);
        SD_HIP(hipGetLastError());
        return SD_OK;
    }
    auto kp = rank_packed_kernel<NT, E>;
    {
#ifdef SD_TUNING
        const char *dbg = getenv("SD_RANKP_DBG");   // timing experiments only: 3 = no sort, ties ignored
        if (dbg && atoi(dbg) == 3) kp = rank_packed_kernel<NT, E, 3>;
#endif
    }
    const size_t lds = C::LDS_BYTES + (size_t)NT * 8;
    SD_HIP(hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kp, dim3(G), dim3(NT), lds, s, Y, n, row0, rows, AB, nnan);
    hipLaunchKernelGGL(ks, dim3(G), dim3(NT), C::LDS_BYTES, s, Y, n, row0, rows, AB, nnan, 1);   // same grid!
    SD_HIP(hipGetLastError());
    return SD_OK;
}

int launch_mbd_rank_v1(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                       u64 *out, void *ws, size_t ws_bytes, hipStream_t s);
#endif

int launch_mbd_rank(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                    u64 *out, void *ws, size_t ws_bytes, hipStream_t s) {
    if (!mbd_rank_supported(T, n, J)) return fail(SD_ERR_UNSUPPORTED, "rank kernels cover 2 <= n <= 16384");
    const int impl = rank_impl(T, n, J);
#ifdef SD_CROSSCHECK
    if (impl == 1 && J <= 3) return launch_mbd_rank_v1(Y, T, n, targets, tbegin, m, J, out, ws, ws_bytes, s);
#else
    if (impl != 4 && impl != 5) return fail(SD_ERR_UNSUPPORTED, "rank implementation %d exists in cross-check builds only", impl);
#endif
    const i64 rpb = ab_rows_per_batch(T, n);
    Carver cv(ws, ws_bytes);
    if (impl == 4) {
        // the bucket kernel ranks every row itself: partial totals per workgroup, no pair image, no search launch
        u64 *partial = (u64 *)cv.take(mbd_rank_bucket_partial_bytes(n, J));
        if (!partial) return fail(SD_ERR_WORKSPACE, "rank workspace too small (bucket kernel)");
        // the two-launch path takes up to 4 096 rows per batch (its list of flagged rows sits in LDS): longer series go
        // through it in batches of that size, the last one through whichever path takes its length
        // (a call for a subset of the targets ranks every row all the same: the path computes the totals of ALL curves into
        // the workspace and the subset is gathered at the end -- provided every batch of the call takes this path)
        const bool all_targets = !targets && tbegin == 0 && m == n;
        const i64 step = (J == 2 && rpb > 4096 && rank_bucket_two_level_supported(n, 4096)) ? 4096 : rpb;
        bool every = J == 2;                                            // does every batch of this call take the two-launch path?
        for (i64 row0 = 0; row0 < T && every; row0 += step) every = rank_bucket_two_level_supported(n, T - row0 < step ? T - row0 : step);
        const bool subset_tl = !all_targets && every && m >= 1;
        const bool two_level = J == 2 && (all_targets || subset_tl);
        u64 *const tl_out = subset_tl ? rank_bucket_two_level_all_totals(partial, n) : out;
        for (i64 row0 = 0; row0 < T; row0 += step) {
            const i64 rows = T - row0 < step ? T - row0 : step;
            int rc, G = 0, p32 = 0;
            if (two_level && rank_bucket_two_level_supported(n, rows)) {
                // 32-bit key images, two workgroups per CU; the second launch finalizes (and ranks what the first flagged)
                if ((rc = launch_rank_bucket_two_level(Y, n, row0, rows, partial, tl_out, row0 == 0, s))) return rc;
                continue;
            }
            if ((rc = launch_rank_bucket(Y, n, row0, rows, J, partial, &p32, &G, s))) return rc;
            if ((rc = launch_rank_finalize(partial, G, p32, nullptr, nullptr, nullptr, rows, n, targets, tbegin, m, J, out,
                                           row0 == 0, s)))
                return rc;
        }
        if (tl_out != out) return launch_rank_gather_totals(tl_out, targets, tbegin, m, out, s);
        return SD_OK;
    }
    u32 *AB = (u32 *)cv.take((size_t)rpb * n * 4);
    u32 *nnan = (u32 *)cv.take((size_t)rpb * 4);
    if (!AB || !nnan) return fail(SD_ERR_WORKSPACE, "rank workspace too small");
    for (i64 row0 = 0; row0 < T; row0 += rpb) {
        const i64 rows = T - row0 < rpb ? T - row0 : rpb;
        int rc;
        // E = 16 keys per thread throughout; smaller rows take smaller workgroups so that several rows are in
        // flight per CU (n = 4000: 4 workgroups of 256 threads per CU, 0.053 ms against 0.091 ms for 1024 x 4)
        if (impl == 5) rc = launch_rank_bucket_image(Y, n, row0, rows, AB, nnan, s);
#ifdef SD_CROSSCHECK
        else if (n <= 1024) rc = launch_sorts<64, 16>(Y, n, row0, rows, AB, nnan, impl, s);
        else if (n <= 2048) rc = launch_sorts<128, 16>(Y, n, row0, rows, AB, nnan, impl, s);
        else if (n <= 4096) rc = launch_sorts<256, 16>(Y, n, row0, rows, AB, nnan, impl, s);
        else if (n <= 8192) rc = launch_sorts<512, 16>(Y, n, row0, rows, AB, nnan, impl, s);
        else rc = launch_sorts<1024, 16>(Y, n, row0, rows, AB, nnan, impl, s);
#else
        else rc = fail(SD_ERR_UNSUPPORTED, "sort-based rank kernels exist in cross-check builds only");
#endif
        if (rc) return rc;
        dim3 grid((unsigned)((m + 63) / 64));
        const int first = row0 == 0;
        if (!targets && (n % 4) == 0 && (tbegin % 4) == 0 && (m % 4) == 0 && J <= 3) {
            SD_DISPATCH_J(J, hipLaunchKernelGGL((rank_accumulate4_kernel<J_>), grid, dim3(1024), 0, s, (const u32 *)AB,
                                                (const u32 *)nnan, rows, n, tbegin, m, out, first));
        } else {
            SD_DISPATCH_J(J, hipLaunchKernelGGL((rank_accumulate_kernel<J_>), grid, dim3(1024), 0, s, (const u32 *)AB,
                                                (const u32 *)nnan, rows, n, targets, tbegin, m, out, first));
        }
        SD_HIP(hipGetLastError());
    }
    return SD_OK;
}

// 16 384 < n <= 40 960: pair image by rank_medium_image_kernel (2 or 3 column blocks per workgroup, mbd_rank_bucket.hip),
// folded by the same accumulate kernels as the bucket kernel's image mode.
bool rank_medium_supported(i64 n);
int launch_rank_medium_image(const double *Y, i64 n, i64 row0, i64 rows, u32 *AB, u32 *nnan, hipStream_t s);

bool mbd_rank_medium_supported(i64 T, i64 n, int J) {
    // one workgroup per row: with few rows the large-n route, which spreads a row over several workgroups, fills the chip better
    // ... and since the large-n route's third generation (round 3) three column blocks lose to it: 40 960 x 500 in 0.322 against
    // 0.287 ms, 32 768 (two blocks) 0.263 against 0.267.  Cross-check builds, SD_MEDIUM_WIDE = 1: up to the kernel's 40 960
    const i64 top = xswitch("SD_MEDIUM_WIDE") == 1 ? (i64)40960 : (i64)32768;
    return rank_medium_supported(n) && n <= top && T >= 96 && J >= 2 && J <= JMAX && xswitch("SD_BIG_NOMEDIUM") != 1;
}

size_t mbd_rank_medium_workspace_bytes(i64 T, i64 n, int J) {
    if (!mbd_rank_medium_supported(T, n, J)) return 0;
    const i64 rpb = ab_rows_per_batch(T, n);
    return align_up((size_t)rpb * n * 4, 256) + align_up((size_t)rpb * 4, 256) + 512;
}

int launch_mbd_rank_medium(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J, u64 *out, void *ws,
                           size_t ws_bytes, hipStream_t s) {
    if (!mbd_rank_medium_supported(T, n, J)) return fail(SD_ERR_UNSUPPORTED, "medium rank route covers 16384 < n <= 32768 with T >= 96");
    const i64 rpb = ab_rows_per_batch(T, n);
    Carver cv(ws, ws_bytes);
    u32 *AB = (u32 *)cv.take((size_t)rpb * n * 4);
    u32 *nnan = (u32 *)cv.take((size_t)rpb * 4);
    if (!AB || !nnan) return fail(SD_ERR_WORKSPACE, "rank workspace too small (medium route)");
    for (i64 row0 = 0; row0 < T; row0 += rpb) {
        const i64 rows = T - row0 < rpb ? T - row0 : rpb;
        int rc = launch_rank_medium_image(Y, n, row0, rows, AB, nnan, s);
        if (rc) return rc;
        dim3 grid((unsigned)((m + 63) / 64));
        const int first = row0 == 0;
        if (!targets && (n % 4) == 0 && (tbegin % 4) == 0 && (m % 4) == 0 && J <= 3) {
            SD_DISPATCH_J(J, hipLaunchKernelGGL((rank_accumulate4_kernel<J_>), grid, dim3(1024), 0, s, (const u32 *)AB,
                                                (const u32 *)nnan, rows, n, tbegin, m, out, first));
        } else {
            SD_DISPATCH_J(J, hipLaunchKernelGGL((rank_accumulate_kernel<J_>), grid, dim3(1024), 0, s, (const u32 *)AB,
                                                (const u32 *)nnan, rows, n, targets, tbegin, m, out, first));
        }
        SD_HIP(hipGetLastError());
    }
    return SD_OK;
}

}  // namespace sd
