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

namespace sd {

template <int NT, int E>
struct R2Cfg {
    static constexpr int N = NT * E;
    static constexpr int LE = (E == 1) ? 0 : (E == 2) ? 1 : (E == 4) ? 2 : (E == 8) ? 3 : (E == 16) ? 4 : 5;
    static constexpr int LT = (NT == 256) ? 8 : (NT == 512) ? 9 : 10;
    static constexpr int LN = LE + LT;
    static constexpr int WB = 64 * E;                  // positions owned by one wave in wave-local layouts
    static constexpr int SLOTS = N + (N >> LE);
    static constexpr size_t LDS_BYTES = (size_t)SLOTS * 8;
};

template <int LE>
__device__ __forceinline__ int r2_phys(int p) { return p + (p >> LE); }

template <int B, int LE>
__device__ __forceinline__ int r2_base(int t) {
    int u = ((t >> B) << (B + LE)) | (t & ((1 << B) - 1));
    return u + (u >> LE);
}
template <int B, int LE>
__device__ __forceinline__ constexpr int r2_off(int r) { return (r << B) + ((r << B) >> LE); }

__device__ __forceinline__ void r2_cmpx(double &a, double &b) {
    // exactly two instructions: the builtin fmin/fmax add a canonicalising v_max_f64 x,x,x per operand
    // after every LDS load (keys are never NaN here, so no quieting is needed)
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo;
    b = hi;
}

__device__ __forceinline__ void r2_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NT, int E>
struct R2Sorter {
    using C = R2Cfg<NT, E>;
    static constexpr int LE = C::LE;
    static constexpr int LN = C::LN;

    static constexpr int wb(int S, int k) { return (S - (k + 1) * LE) > 0 ? (S - (k + 1) * LE) : 0; }

    // number of real registers of thread t in window B: positions ((t>>B) << (B+LE)) + (r << B) + (t & (2^B-1))
    // below n_act.  n_act is a multiple of 64*E, so the count is wave-uniform for B >= 6.
    template <int B>
    static __device__ __forceinline__ int cnt_of(int t, int n_act) {
        int c = (n_act - ((t >> B) << (B + LE)) - (t & ((1 << B) - 1)) + (1 << B) - 1) >> B;
        c = c < 0 ? 0 : (c > E ? E : c);
        return __builtin_amdgcn_readfirstlane(c);
    }

    // wave-local transpose (both windows <= 6); REV: lower-half registers read the reversed lower half
    template <int BF, int BT, bool REV>
    static __device__ __forceinline__ void transpose_local(double (&k)[E], double *S, int t) {
        double *Sw = S + r2_base<BF, LE>(t);
#pragma unroll
        for (int r = 0; r < E; ++r) Sw[r2_off<BF, LE>(r)] = k[r];
        r2_wave_sync();
        const double *Sr = S + r2_base<BT, LE>(t);
        if constexpr (REV) {
            const double *Sf = S + r2_base<BT, LE>(t ^ ((1 << BT) - 1));
#pragma unroll
            for (int r = 0; r < E; ++r)
                k[r] = (r < E / 2) ? Sf[r2_off<BT, LE>(r ^ (E / 2 - 1))] : Sr[r2_off<BT, LE>(r)];
        } else {
#pragma unroll
            for (int r = 0; r < E; ++r) k[r] = Sr[r2_off<BT, LE>(r)];
        }
    }

    // transpose through a workgroup barrier (BF or BT > 6); S = stage (for the reversal's activity test)
    template <int BF, int BT, bool REV, int S>
    static __device__ __forceinline__ void transpose_global(double (&k)[E], double *Sm, int t, int n_act,
                                                            bool wreal, double maxkey) {
        if constexpr (BF <= 6) {
            if (wreal) {
                double *Sw = Sm + r2_base<BF, LE>(t);
#pragma unroll
                for (int r = 0; r < E; ++r) Sw[r2_off<BF, LE>(r)] = k[r];
            }
        } else {
            const int cf = cnt_of<BF>(t, n_act);
            double *Sw = Sm + r2_base<BF, LE>(t);
#pragma unroll
            for (int r = 0; r < E; ++r)
                if (r < cf) Sw[r2_off<BF, LE>(r)] = k[r];
        }
        __syncthreads();
        if constexpr (BT <= 6) {
            static_assert(!REV || BT > 6, "a stage's first window is entered from layout 0");
            if (wreal) {
                const double *Sr = Sm + r2_base<BT, LE>(t);
#pragma unroll
                for (int r = 0; r < E; ++r) k[r] = Sr[r2_off<BT, LE>(r)];
            }
        } else {
            const int ct = cnt_of<BT>(t, n_act);
            const double *Sr = Sm + r2_base<BT, LE>(t);
            bool act = false;
            if constexpr (REV) {
                int a = n_act > (((t >> BT) << S) + (1 << (S - 1)));
                act = __builtin_amdgcn_readfirstlane(a) != 0;
            }
            if (REV && act) {
                // active block: its lower half is entirely real and is read reversed
                const double *Sf = Sm + r2_base<BT, LE>(t ^ ((1 << BT) - 1));
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    if (r < E / 2) k[r] = Sf[r2_off<BT, LE>(r ^ (E / 2 - 1))];
                    else k[r] = (r < ct) ? Sr[r2_off<BT, LE>(r)] : maxkey;
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; ++r) k[r] = (r < ct) ? Sr[r2_off<BT, LE>(r)] : maxkey;
            }
            // the reversed read took slots that other waves own in this window: they must not be
            // rewritten (next transpose) before every wave has read them
            if constexpr (REV) __syncthreads();
        }
    }

    template <int B, int HI, int LO>
    static __device__ __forceinline__ void levels(double (&k)[E]) {
#pragma unroll
        for (int j = HI; j >= LO; --j) {
            const int jr = j - B;
#pragma unroll
            for (int r = 0; r < E; ++r)
                if (!((r >> jr) & 1)) r2_cmpx(k[r], k[r | (1 << jr)]);
        }
    }

    template <int S, int K, int BPREV>
    static __device__ __forceinline__ void windows(double (&k)[E], double *Sm, int t, int n_act, bool wreal,
                                                   double maxkey) {
        constexpr int B = wb(S, K);
        constexpr int HI = (K == 0) ? S - 1 : BPREV - 1;
        constexpr bool REV = (K == 0);
        constexpr bool GLOBAL = (B > 6) || (BPREV > 6);
        if constexpr (B != BPREV) {
            if constexpr (GLOBAL) transpose_global<BPREV, B, REV, S>(k, Sm, t, n_act, wreal, maxkey);
            else if (wreal) transpose_local<BPREV, B, REV>(k, Sm, t);
        }
        if (B > 6 || wreal) levels<B, HI, B>(k);
        if constexpr (B > 0) windows<S, K + 1, B>(k, Sm, t, n_act, wreal, maxkey);
    }

    template <int S, int SLIM = 99>
    static __device__ __forceinline__ void stage(double (&k)[E], double *Sm, int t, int n_act, bool wreal,
                                                 double maxkey) {
        if constexpr (S <= LE) {
            if (wreal) {
                // mirror comparators of the normalised network, all inside the thread
#pragma unroll
                for (int r = 0; r < E; ++r)
                    if (!((r >> (S - 1)) & 1)) r2_cmpx(k[r], k[r ^ ((1 << S) - 1)]);
                if constexpr (S >= 2) levels<0, S - 2, 0>(k);
            }
        } else {
            windows<S, 0, 0>(k, Sm, t, n_act, wreal, maxkey);
        }
        if constexpr (S < LN && S < SLIM) stage<S + 1, SLIM>(k, Sm, t, n_act, wreal, maxkey);
    }

    template <int SLIM = 99>
    static __device__ __forceinline__ void sort(double (&k)[E], double *Sm, int t, int n_act, bool wreal,
                                                double maxkey) {
        stage<1, SLIM>(k, Sm, t, n_act, wreal, maxkey);
    }
};

// fixed-depth descent over the sorted LDS image restricted to [0, n_act)
template <int N, int LE, bool INCL>
__device__ __forceinline__ int r2_bound(const double *Sm, int n_act, double x, double big) {
    int c = 0;
#pragma unroll
    for (int s = N >> 1; s >= 1; s >>= 1) {
        int pos = c + s - 1;
        double a = (pos < n_act) ? Sm[r2_phys<LE>(pos)] : big;
        bool go = INCL ? (a <= x) : (a < x);
        c += go ? s : 0;
    }
    if (INCL) {
        double a = (c < n_act) ? Sm[r2_phys<LE>(c)] : big;
        c += (a <= x) ? 1 : 0;
    }
    return c;
}

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
                        int lo = r2_bound<N, LE, false>(Sm, n_act, x, INF);
                        // upper bound: one probe settles it unless x is tied with its successor
                        int hi = lo + 1;
                        double nx = (hi < n_act) ? Sm[r2_phys<LE>(hi)] : INF;
                        if (hi < n_act && nx <= x) hi = r2_bound<N, LE, true>(Sm, n_act, x, INF);
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
void launch_rank_reduce(const u64 *partial, int G, i64 n, int jc, const i64 *targets, i64 m, u64 *out, hipStream_t s);
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

int launch_mbd_rank2(const double *Y, i64 T, i64 n, const i64 *targets, i64 m, int J,
                     u64 *out, void *ws, size_t ws_bytes, hipStream_t s) {
    int G = rank_grid_for(T);
    size_t need = (size_t)G * (J - 1) * n * 8;
    if (!ws || ws_bytes < need) return fail(SD_ERR_WORKSPACE, "rank workspace too small");
    u64 *partial = (u64 *)ws;
    int rc = (J == 2) ? launch_rank2_j<2>(Y, T, n, partial, G, s) : launch_rank2_j<3>(Y, T, n, partial, G, s);
    if (rc) return rc;
    launch_rank_reduce(partial, G, n, J - 1, targets, m, out, s);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

}  // namespace sd
