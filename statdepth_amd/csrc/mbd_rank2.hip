// mbd_rank2.hip -- K1+K2 rank formulation, second generation (the default for n <= 16384).
//
// Same integers as mbd_rank.hip / mbd_pairwise.hip (the reference's band counts,
// _functional.py:246-251 + _containment.py:75-77).  Differences from the first generation:
//
//  1. Normalised bitonic network: every compare-exchange is ascending; each merge stage first
//     reverses the lower half of its blocks, and that permutation is folded into the LDS read
//     that opens the stage (registers of the lower half read mirrored slots).  Padding then
//     never moves, so positions >= n_act = roundup(n, 64*E) are never stored, moved or
//     compared: whole waves idle through the wave-local stages (10 000 curves occupy 10 of
//     the 16 waves), and in the cross-wave windows their registers are neither written to
//     nor read from LDS.  No sign flips at stage boundaries either.
//  2. Rows are loaded wave-block-wise (coalesced 512 B per wave instruction) into the waves
//     below n_act; the rank search afterwards is spread over all waves (curve i belongs to
//     thread i mod NT), because it is a chain of dependent LDS reads.
//
// Layout/window machinery (registers hold position bits [b, b+log2 E), transposes through a
// padded LDS image, windows with b <= 6 are wave-local) is as in mbd_rank.hip.
#include <stdlib.h>

#include "sd_common.h"
#include "rank_sort.h"

namespace sd {

// DBG: timing-experiment mask (1 = skip search, 2 = skip sort); production = 0
template <int NT, int E, int J, int DBG = 0>
__global__ __launch_bounds__(NT) void mbd_rank2_kernel(const double *__restrict__ Y, i64 T, i64 n64,
                                                       u64 *__restrict__ partial) {
    using C = R2Cfg<NT, E>;
    using Sorter = R2Sorter<NT, E>;
    constexpr int N = C::N, LE = C::LE, WB = C::WB;
    constexpr int SEARCH_ILP = 4;
    extern __shared__ double Sm[];
    __shared__ u32 s_nnan[2];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int n = (int)n64;
    const int n_act = ((n + WB - 1) / WB) * WB;
    const bool wreal = wave * WB < n_act;
    const double INF = __builtin_huge_val();

    u64 acc[E][JMAX - 1];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int j = 0; j < J - 1; ++j) acc[e][j] = 0;
    if (t < 2) s_nnan[t] = 0;

    // own curves: i = wave*WB + e*64 + lane  (one address, immediate offsets e*512 B)
    const int i0 = wave * WB + lane;
    double k[E];
    auto load_row = [&](i64 tp) {
        const double *rp = Y + tp * n + i0;
#pragma unroll
        for (int e = 0; e < E; ++e) k[e] = (i0 + e * 64 < n) ? rp[e * 64] : INF;
    };
    if (wreal && (i64)blockIdx.x < T) load_row(blockIdx.x);
    __syncthreads();

    int par = 0;
    for (i64 tp = blockIdx.x; tp < T; tp += gridDim.x, par ^= 1) {
        const double *__restrict__ row = Y + tp * n;
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
        if constexpr (!(DBG & 2)) Sorter::sort(k, Sm, t, n_act, wreal, INF);
        if (wreal) {
            double *Sw = Sm + r2_base<0, LE>(t);
#pragma unroll
            for (int r = 0; r < E; ++r) Sw[r2_off<0, LE>(r)] = k[r];
        }
        __syncthreads();
        if (t == 0) s_nnan[par ^ 1] = 0;    // nobody touches the other parity during this row
        const u32 nnan = s_nnan[par];
        const i64 tnext = tp + gridDim.x;
        // every wave searches (curve i = t + e*NT belongs to thread t), although only the waves below
        // n_act sorted: the search is a chain of dependent LDS reads and needs all the parallelism it can get
        if (!(DBG & 1)) {
            const double *xp = row + t;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if ((e & (SEARCH_ILP - 1)) == 0) __builtin_amdgcn_sched_barrier(0);   // searches in flight (VGPRs)
                if (t + e * NT < n) {
                    double x = xp[e * NT];
                    if (x == x) {
                        int lo = r2_bound<N, SlotPad<LE>, false, false>(Sm, n_act, x, INF);   // x is in the row
                        // upper bound: one probe settles it unless x is tied with its successor
                        int hi = lo + 1;
                        double nx = (hi < n_act) ? Sm[r2_phys<LE>(hi)] : INF;
                        if (hi < n_act && nx <= x) hi = r2_bound<N, SlotPad<LE>, true>(Sm, n_act, x, INF);
                        u32 B = (u32)lo;
                        // keys <= x within [0, n_act) are real non-NaN values unless x = +inf
                        u32 A = (x == INF) ? 0u : (u32)(n - hi) - nnan;
                        band_counts_add<J>(A, B, nnan, (u64)(n - 1), acc[e]);
                    }
                }
            }
        }
        if (wreal && tnext < T) load_row(tnext);   // in flight across the barrier
        __syncthreads();                    // LDS is reused by the next row
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int i = t + e * NT;
        if (i < n) {
#pragma unroll
            for (int j = 0; j < J - 1; ++j) partial[((size_t)blockIdx.x * (J - 1) + j) * n + i] = acc[e][j];
        }
    }
}

// (reduction kernel shared with mbd_rank.hip)
void launch_rank_reduce(const u64 *partial, int G, i64 n, int jc, const i64 *targets, i64 tbegin, i64 m, u64 *out,
                        hipStream_t s);
int rank_grid_for(i64 T);

template <int NT, int E, int J, int DBG = 0>
static int launch_rank2_cfg(const double *Y, i64 T, i64 n, u64 *partial, int G, hipStream_t s) {
    using C = R2Cfg<NT, E>;
    auto kern = mbd_rank2_kernel<NT, E, J, DBG>;
    SD_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(G), dim3(NT), C::LDS_BYTES, s, Y, T, n, partial);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

template <int J>
static int launch_rank2_j(const double *Y, i64 T, i64 n, u64 *partial, int G, hipStream_t s) {
    if (n <= 1024) return launch_rank2_cfg<256, 4, J>(Y, T, n, partial, G, s);
    if (n <= 2048) return launch_rank2_cfg<256, 8, J>(Y, T, n, partial, G, s);
    if (n <= 4096) return launch_rank2_cfg<1024, 4, J>(Y, T, n, partial, G, s);
    if (n <= 8192) return launch_rank2_cfg<1024, 8, J>(Y, T, n, partial, G, s);
    const char *dbg = getenv("SD_RANK2_DBG");   // timing experiments only
    int dv = dbg ? atoi(dbg) : 0;
    if (J == 2 && dv == 1) return launch_rank2_cfg<1024, 16, 2, 1>(Y, T, n, partial, G, s);
    if (J == 2 && dv == 3) return launch_rank2_cfg<1024, 16, 2, 3>(Y, T, n, partial, G, s);
    return launch_rank2_cfg<1024, 16, J>(Y, T, n, partial, G, s);
}

int launch_mbd_rank2(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                     u64 *out, void *ws, size_t ws_bytes, hipStream_t s) {
    int G = rank_grid_for(T);
    size_t need = (size_t)G * (J - 1) * n * 8;
    if (!ws || ws_bytes < need) return fail(SD_ERR_WORKSPACE, "rank workspace too small");
    u64 *partial = (u64 *)ws;
    int rc = (J == 2) ? launch_rank2_j<2>(Y, T, n, partial, G, s) : launch_rank2_j<3>(Y, T, n, partial, G, s);
    if (rc) return rc;
    launch_rank_reduce(partial, G, n, J - 1, targets, tbegin, m, out, s);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

}  // namespace sd
