// mbd_pairwise.hip -- K1+K2, pairwise formulation: every curve is streamed against
// every target; per (target, timepoint) the strictly-above / strictly-below counts
// A, B are formed and reduced to band-containment totals in the epilogue.
//
// Replaces the pair loop of _univariate_band_depth (_functional.py:246-251) and the
// per-timepoint test of _r2_containment (_containment.py:75-77) in the reference.
//
// Mapping (gfx950): lanes = targets (each lane keeps its own x_q(t), A, B in VGPRs),
// the streamed row X[t, :] is wave-uniform and arrives through the scalar data
// cache (s_load_dwordx8/x16), so the inner loop is 2 x (v_cmp_*_f64 + add-with-carry)
// per streamed value and no cross-lane traffic at all.  Bound: fp64 compare issue
// (VALU), not HBM -- see DESIGN.md.  The rank kernel (mbd_rank.hip) produces the same
// integers in O(n T log n).
#include "sd_common.h"

namespace sd {

// ---------------------------------------------------------------------------
// strided (t,i) -> time-major, 32x32 tiles through LDS
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_time_major_kernel(const double *__restrict__ X, i64 T, i64 n,
                                                            i64 st, i64 sn, double *__restrict__ Y) {
    __shared__ double tile[32][33];
    i64 i0 = (i64)blockIdx.x * 32, t0 = (i64)blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    // read with the fast index following the input's unit stride
    if (st == 1) {
        for (int k = ty; k < 32; k += 8) {
            i64 i = i0 + k, t = t0 + tx;
            if (i < n && t < T) tile[k][tx] = X[t * st + i * sn];
        }
    } else {
        for (int k = ty; k < 32; k += 8) {
            i64 t = t0 + k, i = i0 + tx;
            if (i < n && t < T) tile[tx][k] = X[t * st + i * sn];
        }
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        i64 t = t0 + k, i = i0 + tx;
        if (i < n && t < T) Y[t * n + i] = tile[tx][k];
    }
}

int launch_to_time_major(const double *X, i64 T, i64 n, i64 st, i64 sn, double *Y, hipStream_t s) {
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((T + 31) / 32));
    hipLaunchKernelGGL(to_time_major_kernel, grid, dim3(256), 0, s, X, T, n, st, sn, Y);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// ---------------------------------------------------------------------------
// per-row NaN counts (N(t) of the containment formula)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nan_count_rows_kernel(const double *__restrict__ Y, i64 T, i64 n,
                                                             u32 *__restrict__ nan_cnt) {
    i64 t = blockIdx.x;
    const double *row = Y + t * n;
    u32 c = 0;
    for (i64 i = threadIdx.x; i < n; i += 256) {
        double v = row[i];
        c += (v != v);
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    __shared__ u32 part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) nan_cnt[t] = part[0] + part[1] + part[2] + part[3];
}

int launch_nan_count_rows(const double *Y, i64 T, i64 n, u32 *nan_cnt, hipStream_t s) {
    hipLaunchKernelGGL(nan_count_rows_kernel, dim3((unsigned)T), dim3(256), 0, s, Y, T, n, nan_cnt);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// ---------------------------------------------------------------------------
// K1+K2 pairwise
// ---------------------------------------------------------------------------
constexpr int PW_THREADS = 256;
constexpr int PW_TB = 4;   // timepoints per workgroup

// Q != nullptr: the targets are EXTERNAL curves Q[t*m + q] (not members of Y); all n curves of Y are "others".
template <int J, bool WRITE_AB>
__global__ __launch_bounds__(PW_THREADS) void mbd_pairwise_kernel(
    const double *__restrict__ Y, i64 T, i64 n, const i64 *__restrict__ targets, i64 tbegin, i64 m,
    const u32 *__restrict__ nan_cnt, u64 *__restrict__ out, u32 *__restrict__ AB,
    const double *__restrict__ Q) {
    i64 q = (i64)blockIdx.x * PW_THREADS + threadIdx.x;
    bool valid = q < m;
    i64 tg = (valid && !Q) ? (targets ? targets[q] : tbegin + q) : 0;
    i64 t0 = (i64)blockIdx.y * PW_TB;
    u64 acc[JMAX - 1];
#pragma unroll
    for (int j = 0; j < JMAX - 1; ++j) acc[j] = 0;

    for (int tt = 0; tt < PW_TB; ++tt) {
        i64 t = t0 + tt;
        if (t >= T) break;
        const double *__restrict__ row = Y + t * n;
        double xq = Q ? Q[t * m + (valid ? q : 0)] : row[tg];
        u32 A = 0, B = 0;
        i64 i = 0;
        // wave-uniform addresses: the compiler issues scalar loads for row[i]
        for (; i + 8 <= n; i += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                double xi = row[i + u];
                A += (xi > xq);
                B += (xi < xq);
            }
        }
        for (; i < n; ++i) {
            double xi = row[i];
            A += (xi > xq);
            B += (xi < xq);
        }
        if constexpr (WRITE_AB) {
            if (valid) {
                AB[(q * T + t) * 2 + 0] = A;
                AB[(q * T + t) * 2 + 1] = B;
            }
        } else {
            if (valid && xq == xq) band_counts_add<J>(A, B, nan_cnt[t], (u64)(Q ? n : n - 1), acc);
        }
    }
    if constexpr (!WRITE_AB) {
        if (valid) {
#pragma unroll
            for (int j = 2; j <= J; ++j) atomicAdd(&out[q * (J - 1) + (j - 2)], acc[j - 2]);
        }
    }
}

int launch_mbd_pairwise(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                        const u32 *nan_cnt, u64 *out, hipStream_t s) {
    SD_HIP(hipMemsetAsync(out, 0, sizeof(u64) * m * (J - 1), s));
    dim3 grid((unsigned)((m + PW_THREADS - 1) / PW_THREADS), (unsigned)((T + PW_TB - 1) / PW_TB));
    SD_DISPATCH_J(J, hipLaunchKernelGGL((mbd_pairwise_kernel<J_, false>), grid, dim3(PW_THREADS), 0, s,
                                        Y, T, n, targets, tbegin, m, nan_cnt, out, (u32 *)nullptr,
                                        (const double *)nullptr));
    SD_HIP(hipGetLastError());
    return SD_OK;
}

int launch_mbd_external(const double *Y, i64 T, i64 n, const double *Q, i64 m, int J, const u32 *nan_cnt, u64 *out,
                        hipStream_t s) {
    SD_HIP(hipMemsetAsync(out, 0, sizeof(u64) * m * (J - 1), s));
    dim3 grid((unsigned)((m + PW_THREADS - 1) / PW_THREADS), (unsigned)((T + PW_TB - 1) / PW_TB));
    SD_DISPATCH_J(J, hipLaunchKernelGGL((mbd_pairwise_kernel<J_, false>), grid, dim3(PW_THREADS), 0, s,
                                        Y, T, n, (const i64 *)nullptr, (i64)0, m, nan_cnt, out, (u32 *)nullptr, Q));
    SD_HIP(hipGetLastError());
    return SD_OK;
}

int launch_above_below(const double *Y, i64 T, i64 n, const i64 *targets, i64 m, u32 *AB, hipStream_t s) {
    dim3 grid((unsigned)((m + PW_THREADS - 1) / PW_THREADS), (unsigned)((T + PW_TB - 1) / PW_TB));
    hipLaunchKernelGGL((mbd_pairwise_kernel<2, true>), grid, dim3(PW_THREADS), 0, s, Y, T, n, targets, (i64)0, m,
                       (const u32 *)nullptr, (u64 *)nullptr, AB, (const double *)nullptr);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// ---------------------------------------------------------------------------
// Band totals of one target inside an explicit SUBSET of the curves, for many (subset, target) pairs at once:
// the K-block sampled estimator (_samplefunctionaldepth, _functional.py:170-182) calls
// _univariate_band_depth once per (target, block); here every block is one workgroup of one launch.
// Lanes = members of the block (gathered columns), waves = timepoints; A/B/NaN counts by ballot + popcount.
// members[k*bs + c] = column index or -1 (padding); the target column must be among the members.
// ---------------------------------------------------------------------------
template <int J>
__global__ __launch_bounds__(256) void mbd_subset_kernel(const double *__restrict__ Y, i64 T, i64 n,
                                                         const int *__restrict__ members, int bs,
                                                         const int *__restrict__ target, u64 *__restrict__ out) {
    __shared__ u64 red[4][JMAX - 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 k = blockIdx.x;
    const int *mem = members + k * bs;
    const int tg = target[k];
    u32 nmem = 0;                                    // members other than the target
    for (int c = lane; c < bs; c += 64) nmem += (mem[c] >= 0 && mem[c] != tg);
    for (int o = 32; o > 0; o >>= 1) nmem += __shfl_xor(nmem, o);
    u64 acc[JMAX - 1];
#pragma unroll
    for (int j = 0; j < JMAX - 1; ++j) acc[j] = 0;
    for (i64 t = wave; t < T; t += 4) {
        const double *row = Y + t * n;
        const double x = row[tg];
        u32 A = 0, B = 0, N = 0;
        for (int c0 = 0; c0 < bs; c0 += 64) {
            const int c = c0 + lane;
            const int col = (c < bs) ? mem[c] : -1;
            const bool live = col >= 0 && col != tg;
            const double v = live ? row[col] : x;
            A += (u32)__popcll(__ballot(live && v > x));
            B += (u32)__popcll(__ballot(live && v < x));
            N += (u32)__popcll(__ballot(live && v != v));
        }
        if (x == x) band_counts_add<J>(A, B, N, (u64)nmem, acc);
    }
    if (lane == 0)
#pragma unroll
        for (int j = 0; j < J - 1; ++j) red[wave][j] = acc[j];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int j = 0; j < J - 1; ++j) out[k * (J - 1) + j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
}

int launch_mbd_subsets(const double *Y, i64 T, i64 n, const int *members, i64 nb, int bs, const int *target, int J,
                       u64 *out, hipStream_t s) {
    SD_DISPATCH_J(J, hipLaunchKernelGGL((mbd_subset_kernel<J_>), dim3((unsigned)nb), dim3(256), 0, s, Y, T, n, members,
                                        bs, target, out));
    SD_HIP(hipGetLastError());
    return SD_OK;
}

}  // namespace sd
