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

template <int D>
__global__ __launch_bounds__(256) void l1_depth_kernel(const double *__restrict__ P, i64 n, int d_rt,
                                                       const i64 *__restrict__ targets, i64 m,
                                                       double *__restrict__ out) {
    i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    i64 tg = targets ? targets[q] : q;
    constexpr int DM = D > 0 ? D : 64;
    const int d = D > 0 ? D : d_rt;
    double x[DM], e[DM];
#pragma unroll
    for (int c = 0; c < DM; ++c)
        if (c < d) { x[c] = P[tg * d + c]; e[c] = 0.0; }
    for (i64 i = 0; i < n; ++i) {
        if (i == tg) continue;
        const double *y = P + i * d;
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
    out[q] = 1.0 - sqrt(s) / (double)n;
}

int launch_l1_depth(const double *P, i64 n, int d, const i64 *targets, i64 m, double *out, hipStream_t s) {
    dim3 grid((unsigned)((m + 255) / 256));
#define L1_CASE(DD) case DD: hipLaunchKernelGGL((l1_depth_kernel<DD>), grid, dim3(256), 0, s, P, n, d, targets, m, out); break;
    switch (d) {
        L1_CASE(1) L1_CASE(2) L1_CASE(3) L1_CASE(4) L1_CASE(5) L1_CASE(6) L1_CASE(7) L1_CASE(8)
        default: hipLaunchKernelGGL((l1_depth_kernel<0>), grid, dim3(256), 0, s, P, n, d, targets, m, out);
    }
#undef L1_CASE
    SD_HIP(hipGetLastError());
    return SD_OK;
}

}  // namespace sd
