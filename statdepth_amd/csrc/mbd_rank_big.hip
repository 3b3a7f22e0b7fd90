// mbd_rank_big.hip -- K1+K2 rank formulation for rows that do not fit one CU's LDS
// (16 384 < n < 2^31 curves; BASELINE.json config 3 and the north-star stretch case).
//
// A row of n curves is cut into chunks of C = 16 384 keys.
//   kernel A  chunk_sort_kernel:   one workgroup sorts one (row, chunk) in LDS (rank_sort.h,
//             NaN -> +inf and counted per row) and writes the sorted chunk to a scratch image
//             in HBM (rows x chunks x C fp64; config 3: 235 MB of the 288 GB).
//   kernel B  chunk_search_kernel: a persistent workgroup owns one chunk of *curves* and a
//             strided set of rows; per row it streams every sorted chunk of that row through
//             LDS (coalesced 128 KiB copies) and every thread adds, for its own 16 curves,
//             lower_bound / upper_bound within that chunk.  Summed over the chunks these are
//             exactly B (others strictly below) and n_valid - A (A strictly above), so the
//             totals are the integers of the other formulations.
// Work per row: n/C sorts + (n/C)^2 chunk searches, i.e. O(n log C + n^2 log C / C) instead
// of the pairwise kernel's O(n^2) -- ~300x less work at n = 10^5.  Totals are accumulated in
// registers over the rows and added to the int64 outputs with one atomic per curve.
// Replaces the same reference loops as the other K1+K2 kernels (_functional.py:246-251,
// _containment.py:75-77).
#include <stdlib.h>

#include "sd_common.h"
#include "rank_sort.h"

namespace sd {

constexpr int BIG_NT = 1024, BIG_E = 16;
constexpr int BIG_C = BIG_NT * BIG_E;          // 16384 keys per chunk
using BigCfg = R2Cfg<BIG_NT, BIG_E>;

// grid = (nchunks, rows_in_batch)
__global__ __launch_bounds__(BIG_NT) void chunk_sort_kernel(const double *__restrict__ Y, i64 n, i64 row0,
                                                            double *__restrict__ sorted, i64 sstride,
                                                            u32 *__restrict__ nanrow) {
    constexpr int E = BIG_E, WB = BigCfg::WB;
    extern __shared__ double Sm[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const i64 c = blockIdx.x, rb = blockIdx.y;
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

// grid = G persistent workgroups; workgroup g owns curve chunk g % nchunks and rows g / nchunks + k * (G / nchunks)
template <int J>
__global__ __launch_bounds__(BIG_NT) void chunk_search_kernel(const double *__restrict__ Y, i64 n, i64 row0,
                                                              i64 rows, const double *__restrict__ sorted,
                                                              i64 sstride, const u32 *__restrict__ nanrow,
                                                              int nchunks, int qc0, int nqc,
                                                              u64 *__restrict__ totals) {
    constexpr int E = BIG_E, WB = BigCfg::WB, N = BIG_C;
    extern __shared__ double Sm[];
    const int t = threadIdx.x;
    // only the curve chunks [qc0, qc0 + nqc) hold targets
    const int qc = qc0 + (int)(blockIdx.x % nqc);
    const int rgroups = gridDim.x / nqc;
    const i64 qbase = (i64)qc * BIG_C;
    const int nq = (int)((n - qbase) < BIG_C ? (n - qbase) : BIG_C);
    const double INF = __builtin_huge_val();
    u64 acc[E][JMAX - 1];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int j = 0; j < J - 1; ++j) acc[e][j] = 0;

    for (i64 rb = blockIdx.x / nqc; rb < rows; rb += rgroups) {
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
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (t + e * BIG_NT < nq && x[e] == x[e]) {
                u32 B = lo[e];
                u32 A = (x[e] == INF) ? 0u : (u32)(n - hi[e]) - nnan;
                band_counts_add<J>(A, B, nnan, (u64)(n - 1), acc[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        i64 i = qbase + t + e * BIG_NT;
        if (t + e * BIG_NT < nq) {
#pragma unroll
            for (int j = 0; j < J - 1; ++j)
                if (acc[e][j]) atomicAdd(&totals[(size_t)j * n + i], acc[e][j]);
        }
    }
}

// out[q*(J-1)+j] = totals[j][targets[q]]
__global__ __launch_bounds__(256) void big_gather_kernel(const u64 *__restrict__ totals, i64 n, int jc,
                                                         const i64 *__restrict__ targets, i64 tbegin, i64 m,
                                                         u64 *__restrict__ out) {
    i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    i64 i = targets ? targets[q] : tbegin + q;
    for (int j = 0; j < jc; ++j) out[q * jc + j] = totals[(size_t)j * n + i];
}

static inline i64 big_nchunks(i64 n) { return (n + BIG_C - 1) / BIG_C; }

// rows per batch so that the sorted image stays below ~1 GiB
static i64 big_rows_per_batch(i64 T, i64 n) {
    i64 sstride = big_nchunks(n) * BIG_C;
    i64 r = ((i64)1 << 30) / (sstride * 8);
    if (const char *e = getenv("SD_RANK_ROWS_PER_BATCH")) {   // tests: force several batches on small inputs
        i64 v = atoll(e);
        if (v > 0 && v < r) r = v;
    }
    if (r < 1) r = 1;
    if (r > T) r = T;
    if (r > 65535) r = 65535;
    return r;
}

bool mbd_rank_big_supported(i64 T, i64 n, int J) {
    (void)T;
    return n > 16384 && n < ((i64)1 << 31) && (J == 2 || J == 3) && big_nchunks(n) <= 1024;
}

size_t mbd_rank_big_workspace_bytes(i64 T, i64 n, int J) {
    if (!mbd_rank_big_supported(T, n, J)) return 0;
    i64 rows = big_rows_per_batch(T, n);
    size_t b = align_up((size_t)rows * big_nchunks(n) * BIG_C * 8, 256);
    b += align_up((size_t)rows * 4, 256);
    b += align_up((size_t)(J - 1) * n * 8, 256);
    return b + 1024;
}

int launch_mbd_rank_big(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                        u64 *out, void *ws, size_t ws_bytes, hipStream_t s) {
    if (!mbd_rank_big_supported(T, n, J)) return fail(SD_ERR_UNSUPPORTED, "chunked rank kernel covers n > 16384, J in {2,3}");
    const i64 nch = big_nchunks(n);
    const i64 sstride = nch * BIG_C;
    const i64 rpb = big_rows_per_batch(T, n);
    Carver cv(ws, ws_bytes);
    double *sorted = (double *)cv.take((size_t)rpb * sstride * 8);
    u32 *nanrow = (u32 *)cv.take((size_t)rpb * 4);
    u64 *totals = (u64 *)cv.take((size_t)(J - 1) * n * 8);
    if (!sorted || !nanrow || !totals) return fail(SD_ERR_WORKSPACE, "chunked rank workspace too small");
    SD_HIP(hipMemsetAsync(totals, 0, (size_t)(J - 1) * n * 8, s));
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    auto ksort = chunk_sort_kernel;
    SD_HIP(hipFuncSetAttribute((const void *)ksort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BigCfg::LDS_BYTES));
    auto ks2 = chunk_search_kernel<2>;
    auto ks3 = chunk_search_kernel<3>;
    SD_HIP(hipFuncSetAttribute((const void *)ks2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BigCfg::LDS_BYTES));
    SD_HIP(hipFuncSetAttribute((const void *)ks3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BigCfg::LDS_BYTES));
    for (i64 row0 = 0; row0 < T; row0 += rpb) {
        i64 rows = T - row0 < rpb ? T - row0 : rpb;
        SD_HIP(hipMemsetAsync(nanrow, 0, (size_t)rows * 4, s));
        hipLaunchKernelGGL(ksort, dim3((unsigned)nch, (unsigned)rows), dim3(BIG_NT), BigCfg::LDS_BYTES, s, Y, n, row0,
                           sorted, sstride, nanrow);
        // persistent search grid: a multiple of the number of target chunks, about one workgroup per CU.
        // An explicit target list may point anywhere, so every chunk is searched; a contiguous block
        // (targets == NULL) restricts the search to the chunks it overlaps.
        i64 qc0 = targets ? 0 : tbegin / BIG_C;
        i64 qc1 = targets ? nch : (tbegin + m + BIG_C - 1) / BIG_C;
        i64 nqc = qc1 - qc0;
        i64 rgroups = cus / nqc;
        if (rgroups < 1) rgroups = 1;
        if (rgroups > rows) rgroups = rows;
        unsigned G = (unsigned)(rgroups * nqc);
        if (J == 2)
            hipLaunchKernelGGL(ks2, dim3(G), dim3(BIG_NT), BigCfg::LDS_BYTES, s, Y, n, row0, rows, sorted, sstride,
                               nanrow, (int)nch, (int)qc0, (int)nqc, totals);
        else
            hipLaunchKernelGGL(ks3, dim3(G), dim3(BIG_NT), BigCfg::LDS_BYTES, s, Y, n, row0, rows, sorted, sstride,
                               nanrow, (int)nch, (int)qc0, (int)nqc, totals);
        SD_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(big_gather_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, totals, n, J - 1, targets,
                       tbegin, m, out);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

}  // namespace sd
