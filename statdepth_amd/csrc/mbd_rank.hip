// mbd_rank.hip -- K1+K2, rank formulation, FIRST GENERATION (kept as an independent cross-check,
// SD_RANK_IMPL=1; the default path is mbd_rank_ab.hip): the same integers as the pairwise kernel in
// O(n T log n).
//
// For one timepoint the counts A (others strictly above) and B (strictly below) of
// every curve follow from its position in the sorted row: B = lower_bound(x),
// A = n_valid - upper_bound(x).  One workgroup sorts a whole row X[t, 0:n] in LDS
// (n <= 16384: 128 KiB of fp64 keys inside the CU's 160 KiB), every thread then
// looks its own curves up by binary search and folds C(v,j) - C(A,j) - C(B,j) into
// per-curve register accumulators that persist across the timepoints the workgroup
// owns.  The matrix is read from HBM exactly once (coalesced rows), so this is the
// formulation whose floor is the HBM roofline; what it actually pays for is the
// LDS/VALU cost of the sort (DESIGN.md has the budget).
//
// Sort: bitonic network on fp64 keys only (no payload), v_min_f64 / v_max_f64 as the
// compare-exchange.  Each thread keeps E keys in VGPRs; a level whose partner
// distance lies inside the thread's register window is pure VALU, and the window is
// moved by transposing through LDS ("layout b": register index = position bits
// [b, b+log2 E)).  Windows with b <= 6 keep every wave inside its own 64*E-key block,
// so those transposes need no workgroup barrier; only the four last stages cross
// waves (8 barriers per row).  Descending sub-sequences are stored negated (sign-bit
// flip at stage boundaries), so every compare-exchange is ascending and costs exactly
// two fp64 instructions.  NaN keys are replaced by +inf and counted (pandas skipna,
// _containment.py:68-69); padding up to the power of two is +inf as well.
//
// Replaces the same reference loops as mbd_pairwise.hip (_functional.py:246-251,
// _containment.py:75-77); results are bit-identical to it and to the oracle.
#include <stdlib.h>

#include "sd_common.h"

namespace sd {

constexpr int RK_MAXG = 1024;   // max persistent workgroups (partial-sum rows)

template <int NT, int E>
struct RankCfg {
    static constexpr int N = NT * E;
    static constexpr int LE = (E == 1) ? 0 : (E == 2) ? 1 : (E == 4) ? 2 : (E == 8) ? 3 : (E == 16) ? 4 : 5;
    static constexpr int LT = (NT == 256) ? 8 : (NT == 512) ? 9 : 10;
    static constexpr int LN = LE + LT;
    static constexpr int SLOTS = N + (N >> LE);       // padded: one slot per E keys
    static constexpr size_t LDS_BYTES = (size_t)SLOTS * 8;
};

template <int LE>
__device__ __forceinline__ int phys(int p) { return p + (p >> LE); }

// Position of (thread t, register r) when registers hold position bits [B, B+LE):
//   p = ((t >> B) << (B+LE)) | (r << B) | (t & (2^B - 1)).
// The bit fields are disjoint, so the padded LDS slot splits into a per-thread base and a
// per-register constant: phys(p) = lds_base<B>(t) + lds_off<B>(r) -- one address VGPR per layout,
// the rest folds into the ds_read/ds_write immediate offsets.
template <int B, int LE>
__device__ __forceinline__ int lds_base(int t) {
    int u = ((t >> B) << (B + LE)) | (t & ((1 << B) - 1));
    return u + (u >> LE);
}
template <int B, int LE>
__device__ __forceinline__ constexpr int lds_off(int r) {
    return (r << B) + ((r << B) >> LE);
}

__device__ __forceinline__ void cmpx(double &a, double &b) {
    // exactly two instructions: the builtin fmin/fmax add a canonicalising v_max_f64 x,x,x per operand
    // after every LDS load (keys are never NaN here, so no quieting is needed)
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo;
    b = hi;
}

__device__ __forceinline__ double flip_sign(double v, unsigned m) {   // m = 0 or 0x80000000
    unsigned long long u = __double_as_longlong(v);
    u ^= ((unsigned long long)m) << 32;
    return __longlong_as_double(u);
}

template <int NT, int E, int SK = 0>
struct RankSorter {
    using C = RankCfg<NT, E>;
    static constexpr int LE = C::LE;
    static constexpr int LN = C::LN;

    // window k of stage S: register bits start at wb(S,k)
    static constexpr int wb(int S, int k) { return (S - (k + 1) * LE) > 0 ? (S - (k + 1) * LE) : 0; }

    template <int BF, int BT>
    static __device__ __forceinline__ void transpose(double (&k)[E], double *S, int t) {
        constexpr bool global = (BF > 6) || (BT > 6);
        double *Sw = S + lds_base<BF, LE>(t);
#pragma unroll
        for (int r = 0; r < E; ++r) Sw[lds_off<BF, LE>(r)] = k[r];
        if constexpr (global) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const double *Sr = S + lds_base<BT, LE>(t);
#pragma unroll
        for (int r = 0; r < E; ++r) k[r] = Sr[lds_off<BT, LE>(r)];
    }

    // levels HI..LO (position bits) with register window at B
    template <int B, int HI, int LO>
    static __device__ __forceinline__ void levels(double (&k)[E]) {
#pragma unroll
        for (int j = HI; j >= LO; --j) {
            const int jr = j - B;
#pragma unroll
            for (int r = 0; r < E; ++r)
                if (!((r >> jr) & 1)) cmpx(k[r], k[r | (1 << jr)]);
        }
    }

    template <int S, int K, int BPREV>
    static __device__ __forceinline__ void windows(double (&k)[E], double *Sm, int t) {
        constexpr int B = wb(S, K);
        constexpr int HI = (K == 0) ? S - 1 : BPREV - 1;
        if constexpr (B != BPREV && !(SK & 2)) transpose<BPREV, B>(k, Sm, t);
        if constexpr (!(SK & 1)) levels<B, HI, B>(k);
        if constexpr (B > 0) windows<S, K + 1, B>(k, Sm, t);
    }

    template <int S>
    static __device__ __forceinline__ void stage(double (&k)[E], double *Sm, int t) {
        // stored = (-1)^g(p) * x with g_S(p) = bit S of p (0 for the last stage): flip where g changes
#pragma unroll
        for (int r = 0; r < E; ++r) {
            int p = t * E + r;
            unsigned gs = (S < LN) ? ((p >> S) & 1) : 0;
            unsigned gp = (S > 1) ? ((p >> (S - 1)) & 1) : 0;
            k[r] = flip_sign(k[r], (gs ^ gp) << 31);
        }
        windows<S, 0, 0>(k, Sm, t);
        if constexpr (S < LN) stage<S + 1>(k, Sm, t);
    }

    static __device__ __forceinline__ void sort(double (&k)[E], double *Sm, int t) { stage<1>(k, Sm, t); }
};

// grid = G persistent workgroups; workgroup g owns timepoints g, g+G, ...
// partial[(g*(J-1) + j)*n + i] = sum over its timepoints of the band counts of curve i
// PH: debug ablation mask (1 = sort, 2 = search); production launches use PH = 3
template <int NT, int E, int J, int PH = 3>
__global__ __launch_bounds__(NT) void mbd_rank_kernel(const double *__restrict__ Y, i64 T, i64 n,
                                                      u64 *__restrict__ partial) {
    using C = RankCfg<NT, E>;
    constexpr int N = C::N;
    constexpr int LE = C::LE;
    extern __shared__ double Sm[];
    __shared__ u32 s_nnan;
    const int t = threadIdx.x;
    u64 acc[E][JMAX - 1];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int j = 0; j < J - 1; ++j) acc[e][j] = 0;
    const double INF = __builtin_huge_val();
    const u32 npad = (u32)(N - n);

    for (i64 tp = blockIdx.x; tp < T; tp += gridDim.x) {
        const double *__restrict__ row = Y + tp * n;
        if (t == 0) s_nnan = 0;
        double k[E];
        u32 mynan = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            i64 i = (i64)t + (i64)e * NT;
            double v = (i < n) ? row[i] : INF;
            if (v != v) { v = INF; ++mynan; }
            k[e] = v;
        }
        __syncthreads();                       // s_nnan zeroed; previous row's searches finished
        if (mynan) atomicAdd(&s_nnan, mynan);
        if constexpr (PH & 1) RankSorter<NT, E, (PH >> 2)>::sort(k, Sm, t);
        {
            double *Sw = Sm + lds_base<0, LE>(t);
#pragma unroll
            for (int r = 0; r < E; ++r) Sw[lds_off<0, LE>(r)] = k[r];
        }
        __syncthreads();
        const u32 nnan = s_nnan;
        // look every own curve up in the sorted row
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if ((e & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // keep at most 4 searches in flight (VGPRs)
            i64 i = (i64)t + (i64)e * NT;
            if ((PH & 2) && i < n) {
                double x = row[i];
                if (x == x) {
                    // lower bound by a fixed-depth descent; x itself is in the row, so Sm[lo] == x afterwards
                    int lo = 0;
#pragma unroll
                    for (int s = N >> 1; s >= 1; s >>= 1) {
                        double a = Sm[phys<LE>(lo + s - 1)];
                        lo += (a < x) ? s : 0;
                    }
                    // upper bound: one probe settles it unless x is tied with its successor
                    int hi = lo + 1;
                    if (hi < N && Sm[phys<LE>(hi)] <= x) {
                        hi = 0;
#pragma unroll
                        for (int s = N >> 1; s >= 1; s >>= 1) {
                            double b = Sm[phys<LE>(hi + s - 1)];
                            hi += (b <= x) ? s : 0;
                        }
                        hi += (Sm[phys<LE>(hi)] <= x) ? 1 : 0;
                    }
                    u32 B = (u32)lo;
                    u32 A = (x == INF) ? 0u : (u32)(N - hi) - npad - nnan;
                    band_counts_add<J>(A, B, nnan, (u64)(n - 1), acc[e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        i64 i = (i64)t + (i64)e * NT;
        if (i < n) {
#pragma unroll
            for (int j = 0; j < J - 1; ++j) partial[((size_t)blockIdx.x * (J - 1) + j) * n + i] = acc[e][j];
        }
    }
}

// out[q*(J-1)+j] = sum_g partial[g][j][targets[q]]
// block = 64 targets x 16 slices of g; LDS tree over the slices
__global__ __launch_bounds__(1024) void rank_reduce_kernel(const u64 *__restrict__ partial, int G, i64 n, int jc,
                                                           const i64 *__restrict__ targets, i64 tbegin, i64 m,
                                                           u64 *__restrict__ out) {
    __shared__ u64 red[16][64];
    int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    i64 q = (i64)blockIdx.x * 64 + x;
    i64 i = (q < m) ? (targets ? targets[q] : tbegin + q) : 0;
    for (int j = 0; j < jc; ++j) {
        u64 s = 0;
        if (q < m)
            for (int g = y; g < G; g += 16) s += partial[((size_t)g * jc + j) * n + i];
        red[y][x] = s;
        __syncthreads();
        if (y == 0 && q < m) {
            u64 tot = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) tot += red[k][x];
            out[q * jc + j] = tot;
        }
        __syncthreads();
    }
}

static int rank_grid(i64 T) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    i64 g = T < cus ? T : cus;
    if (g > RK_MAXG) g = RK_MAXG;
    return (int)g;
}

template <int NT, int E, int J, int PH = 3>
static int launch_rank_cfg(const double *Y, i64 T, i64 n, u64 *partial, int G, hipStream_t s) {
    using C = RankCfg<NT, E>;
    auto kern = mbd_rank_kernel<NT, E, J, PH>;
    SD_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(G), dim3(NT), C::LDS_BYTES, s, Y, T, n, partial);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

template <int J>
static int launch_rank_j(const double *Y, i64 T, i64 n, u64 *partial, int G, hipStream_t s) {
    if (n <= 1024) return launch_rank_cfg<256, 4, J>(Y, T, n, partial, G, s);
    if (n <= 2048) return launch_rank_cfg<256, 8, J>(Y, T, n, partial, G, s);
    if (n <= 4096) return launch_rank_cfg<1024, 4, J>(Y, T, n, partial, G, s);
    if (n <= 8192) return launch_rank_cfg<1024, 8, J>(Y, T, n, partial, G, s);
    // debug knobs (timing experiments only): SD_RANK_CFG=2 -> 512x32, SD_RANK_PH = ablation mask
    const char *cfg = getenv("SD_RANK_CFG");
    const char *ph = getenv("SD_RANK_PH");
    int phv = ph ? atoi(ph) : 3;
    if (J == 2 && phv == 1) return launch_rank_cfg<1024, 16, 2, 1>(Y, T, n, partial, G, s);
    if (J == 2 && phv == 2) return launch_rank_cfg<1024, 16, 2, 2>(Y, T, n, partial, G, s);
    if (J == 2 && phv == 0) return launch_rank_cfg<1024, 16, 2, 0>(Y, T, n, partial, G, s);
    if (J == 2 && phv == 5) return launch_rank_cfg<1024, 16, 2, 5>(Y, T, n, partial, G, s);    // sort without CE
    if (J == 2 && phv == 9) return launch_rank_cfg<1024, 16, 2, 9>(Y, T, n, partial, G, s);    // sort without transposes
    if (cfg && atoi(cfg) == 2) return launch_rank_cfg<512, 32, J>(Y, T, n, partial, G, s);
    return launch_rank_cfg<1024, 16, J>(Y, T, n, partial, G, s);
}

// first-generation rank kernel, selectable with SD_RANK_IMPL=1 (J <= 3) as an independent cross-check
int launch_mbd_rank_v1(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                       u64 *out, void *ws, size_t ws_bytes, hipStream_t s) {
    int G = rank_grid(T);
    size_t need = (size_t)G * (J - 1) * n * 8;
    if (!ws || ws_bytes < need) return fail(SD_ERR_WORKSPACE, "rank workspace too small");
    u64 *partial = (u64 *)ws;
    int rc = (J == 2) ? launch_rank_j<2>(Y, T, n, partial, G, s) : launch_rank_j<3>(Y, T, n, partial, G, s);
    if (rc) return rc;
    hipLaunchKernelGGL(rank_reduce_kernel, dim3((unsigned)((m + 63) / 64)), dim3(1024), 0, s, partial, G, n, J - 1,
                       targets, tbegin, m, out);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

}  // namespace sd
