// l1_depth.hip -- K5: L1 (spatial) depth of a point cloud.
// Replaces _L1_depth (_pointcloud.py:125-150): per point x,
//   e = sum_{y != x} (y - x)/||x - y||   (:145-146, in index order)
//   depth = 1 - ||e|| / n                (:148,150; n includes x).
// Lanes = target points; the streamed point y is wave-uniform (scalar cache).
// fp64 VALU bound.  The unit vector is (y - x) * r with r = 1/sqrt(s) from v_rsq_f64 + two Newton steps (error
// about one ulp) instead of an IEEE sqrt and d IEEE divisions: a third of the instructions; the summation order is
// the oracle's, the result agrees with oracle_l1_depth far inside the 1e-12 the tests allow (floating point: the
// tolerance row of north_star, not the bit-exact one).  Squared distances outside [1e-280, 1e280] and coincident
// points (s = 0 -> NaN like the reference's 0/0) take the IEEE sqrt + division.
#include "sd_common.h"

namespace sd {

// Besides the default form (target = row targets[q] of P, others = the remaining rows) two selections the callers of
// the depth need in ONE launch: EXTERNAL targets Q (m x d; every row of P is an "other", the sample counts n + 1
// points: depth of a point of G inside F u {g}, homogeneity.py:172-186) and explicit BLOCKS of rows per target
// (members int32[m][bs], -1 padded, the block's others first and its target last: the K-block sampled estimator,
// _pointcloud.py:107-121).
template <int D>
__global__ __launch_bounds__(256) void l1_depth_kernel(const double *__restrict__ P, i64 n, int d_rt,
                                                       const i64 *__restrict__ targets, i64 m,
                                                       double *__restrict__ out, const double *__restrict__ Q,
                                                       const int *__restrict__ members, int bs) {
    i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    constexpr int DM = D > 0 ? D : 64;
    const int d = D > 0 ? D : d_rt;
    const int *mem = members ? members + q * bs : nullptr;
    i64 cnt = n;                                     // rows streamed; npts = size of the sample the depth refers to
    i64 tg = -1;
    double npts = (double)n;
    if (mem) {
        int c = 0;
        while (c < bs && mem[c] >= 0) ++c;
        if (c == 0) { out[q] = __builtin_nan(""); return; }
        tg = mem[c - 1];
        cnt = c - 1;
        npts = (double)c;
    } else if (Q) {
        npts = (double)(n + 1);
    } else {
        tg = targets ? targets[q] : q;
    }
    const double *xp = (Q && !mem) ? Q + q * d : P + tg * d;
    double x[DM], e[DM];
#pragma unroll
    for (int c = 0; c < DM; ++c)
        if (c < d) { x[c] = xp[c]; e[c] = 0.0; }
    for (i64 i = 0; i < cnt; ++i) {
        if (!mem && i == tg) continue;
        const double *y = P + (mem ? (i64)mem[i] : i) * d;
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < DM; ++c)
            if (c < d) { double df = x[c] - y[c]; s += df * df; }
        double r;
        if (__builtin_expect(s > 1e-280 && s < 1e280, 1)) {
            r = __builtin_amdgcn_rsq(s);
            r = r * (1.5 - (0.5 * s) * (r * r));
            r = r * (1.5 - (0.5 * s) * (r * r));
        } else {
            r = 1.0 / sqrt(s);                                        // 0 -> inf -> 0 * inf = NaN below
        }
#pragma unroll
        for (int c = 0; c < DM; ++c)
            if (c < d) e[c] += (y[c] - x[c]) * r;
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < DM; ++c)
        if (c < d) s += e[c] * e[c];
    out[q] = 1.0 - sqrt(s) / npts;
}

static int launch_l1_common(const double *P, i64 n, int d, const i64 *targets, i64 m, double *out, const double *Q,
                            const int *members, int bs, hipStream_t s) {
    dim3 grid((unsigned)((m + 255) / 256));
#define L1_CASE(DD) case DD: hipLaunchKernelGGL((l1_depth_kernel<DD>), grid, dim3(256), 0, s, P, n, d, targets, m, out, Q, members, bs); break;
    switch (d) {
        L1_CASE(1) L1_CASE(2) L1_CASE(3) L1_CASE(4) L1_CASE(5) L1_CASE(6) L1_CASE(7) L1_CASE(8)
        default: hipLaunchKernelGGL((l1_depth_kernel<0>), grid, dim3(256), 0, s, P, n, d, targets, m, out, Q, members, bs);
    }
#undef L1_CASE
    SD_HIP(hipGetLastError());
    return SD_OK;
}

int launch_l1_depth(const double *P, i64 n, int d, const i64 *targets, i64 m, double *out, hipStream_t s) {
    return launch_l1_common(P, n, d, targets, m, out, nullptr, nullptr, 0, s);
}

int launch_l1_external(const double *P, i64 n, int d, const double *Q, i64 m, double *out, hipStream_t s) {
    return launch_l1_common(P, n, d, nullptr, m, out, Q, nullptr, 0, s);
}

int launch_l1_subsets(const double *P, i64 n, int d, const int *members, i64 nb, int bs, double *out, hipStream_t s) {
    return launch_l1_common(P, n, d, nullptr, nb, out, nullptr, members, bs, s);
}

}  // namespace sd
