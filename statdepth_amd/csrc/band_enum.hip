// band_enum.hip -- K6: componentwise band containment of multivariate curves ('r2_enum', relax, J = 2).
//
// The reference declares this containment and leaves it unimplemented (`_r2_enum_containment`,
// _containment.py:83-103: "treat each component in the vector valued function as a real valued function, and
// calculate containment for each one.  If all the components are contained ... we say the function is contained";
// the body is `raise NotImplementedError`).  Built here as the band depth of _univariate_band_depth
// (_functional.py:238-253) with that predicate: for a target x and a pair {a, b} of the other curves,
//     contained at t  <=>  for EVERY feature f:  min(a_f(t), b_f(t)) <= x_f(t) <= max(a_f(t), b_f(t)),
// out[q] = sum over t of the number of containing pairs; depth = out / T / C(n, 2) on the host (d = 1 is the univariate
// modified band depth exactly).  The strict form (contained at every t) needs no kernel of its own: it is the strict
// univariate band depth (K3) of the T*d component series, which the host calls directly.
//
// Counting the pairs without enumerating them.  Per (target, t) every other curve a has a STATE per feature --
// tie / above / below -- and a pair is contained iff in no feature both are above or both are below.  That
// compatibility is a product over the features, so with h[c] = number of curves in state vector c (3^d classes) and
// M the 3 x 3 per-feature compatibility (tie ~ anything, above ~ {tie, below}, below ~ {tie, above})
//     ordered containing pairs = h^T (M x M x ... x M) h,
// evaluated as d passes of a 3-point transform over the 3^d counters in LDS (d * 3^d adds) instead of n^2 mask tests:
// 5 000 curves with d = 8: 1.6e5 LDS operations per (target, t) instead of 2.5e7 tests.
// The states come from INTEGER images of the values: sd_above_below's B(a, t, f) = number of curves strictly below
// a's value in feature f at t is order- and tie-preserving (a_f > x_f <=> B(a) > B(x); equal values share B), so one
// O(n^2 T d) pass of the pairwise kernel turns the fp64 data into u16 ranks, all n * d of which for one timepoint fit
// the LDS (80 KB at config 4) and serve every target of that timepoint.
// NaN-free input only (a NaN component would need a fourth state: 4^d counters do not fit); the host refuses NaN.
#include "sd_common.h"

namespace sd {

int launch_rank_bucket_image(const double *Y, i64 n, i64 row0, i64 rows, u32 *AB, u32 *nnan, hipStream_t s);   // mbd_rank_bucket.hip

template <int D>
struct BECfg {
    static constexpr int pow3(int k) { return k == 0 ? 1 : 3 * pow3(k - 1); }
    static constexpr int NC = pow3(D);                         // state vectors
    static constexpr int NT = 1024;
    static size_t lds_bytes(i64 n) { return (size_t)n * D * 2 + (size_t)NC * 8 + 256; }
};

// grid = (T, QG): block (t, g) serves targets g, g + QG, ... at timepoint t.
// AB[(a * R + r) * 2 + 1] = B of curve a in component row r = t * D + f (sd_above_below over the R = T * D rows: n >
// 16 384), or ranks[r][a] & 0xFFFF = the same B from the bucket kernel's image mode (n <= 16 384: one pass of the
// headline kernel instead of the O(n^2) pairwise one -- 12 of config 4's 81 ms).
template <int D>
__global__ __launch_bounds__(1024) void band_class_kernel(const u32 *__restrict__ AB, const u32 *__restrict__ ranks, i64 n64, i64 T,
                                                         const i64 *__restrict__ targets, i64 m, u64 *__restrict__ out) {
    using C = BECfg<D>;
    constexpr int NC = C::NC, NT = C::NT;
    extern __shared__ unsigned char smem[];
    const int n = (int)n64;
    unsigned short *R = reinterpret_cast<unsigned short *>(smem);                  // [n][D] ranks at this timepoint
    u32 *h0 = reinterpret_cast<u32 *>(smem + (((size_t)n * D * 2 + 15) / 16) * 16);  // [NC] curves per state vector
    u32 *z = h0 + NC;                                                             // [NC] transformed counters
    __shared__ u64 red[NT / 64][2];
    const int t = threadIdx.x;
    const i64 tp = blockIdx.x;
    const i64 RR = T * D;
    if (ranks) {            // bucket kernel's image: ranks[row][curve] = B | A << 16, the curves of a row contiguous
        for (i64 idx = t; idx < (i64)n * D; idx += NT) {
            const i64 f = idx / n, a = idx % n;
            R[a * D + f] = (unsigned short)(ranks[(tp * D + f) * n + a] & 0xFFFFu);
        }
    } else {
        for (i64 idx = t; idx < (i64)n * D; idx += NT) {
            const i64 a = idx / D, f = idx % D;
            R[idx] = (unsigned short)AB[((a * RR) + tp * D + f) * 2 + 1];
        }
    }
    __syncthreads();
    // Two targets per sweep: their counters share the 32-bit words (low / high half: counts and transformed counts are
    // <= n < 2^16, and every step of the transform is an addition), so zeroing, the transform and its barriers are paid
    // once for both; the transform takes two features per pass (nine-point groups, M applied along either).
    for (i64 q0 = 2 * (i64)blockIdx.y; q0 < m; q0 += 2 * (i64)gridDim.y) {
        const bool two = q0 + 1 < m;
        const int tgA = (int)(targets ? targets[q0] : q0);
        const int tgB = two ? (int)(targets ? targets[q0 + 1] : q0 + 1) : tgA;
        u32 rqA[D], rqB[D];
#pragma unroll
        for (int f = 0; f < D; ++f) {
            rqA[f] = R[(size_t)tgA * D + f];
            rqB[f] = R[(size_t)tgB * D + f];
        }
        for (int c = t; c < NC; c += NT) h0[c] = 0;
        __syncthreads();
        for (int a = t; a < n; a += NT) {
            u32 codeA = 0, codeB = 0, w = 1;
#pragma unroll
            for (int f = 0; f < D; ++f) {
                const u32 ra = R[(size_t)a * D + f];
                codeA += w * (ra > rqA[f] ? 1u : (ra < rqA[f] ? 2u : 0u));
                codeB += w * (ra > rqB[f] ? 1u : (ra < rqB[f] ? 2u : 0u));
                w *= 3u;
            }
            if (a != tgA) atomicAdd(&h0[codeA], 1u);
            if (a != tgB) atomicAdd(&h0[codeB], 0x10000u);
        }
        __syncthreads();
        for (int c = t; c < NC; c += NT) z[c] = h0[c];
        __syncthreads();
        // z <- (M x ... x M) h, M = tie: compatible with every state; above: with tie and below; below: with tie and above
        int stride = 1;
#pragma unroll
        for (int f = 0; f < D; f += 2) {
            if (f + 1 < D) {
                for (int g = t; g < NC / 9; g += NT) {
                    const int base = (g / stride) * stride * 9 + (g % stride);
                    u32 v[3][3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const u32 s0 = z[base + (3 * j) * stride], s1 = z[base + (3 * j + 1) * stride], s2 = z[base + (3 * j + 2) * stride];
                        v[j][0] = s0 + s1 + s2;
                        v[j][1] = s0 + s2;
                        v[j][2] = s0 + s1;
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        z[base + i * stride] = v[0][i] + v[1][i] + v[2][i];
                        z[base + (3 + i) * stride] = v[0][i] + v[2][i];
                        z[base + (6 + i) * stride] = v[0][i] + v[1][i];
                    }
                }
                stride *= 9;
            } else {
                for (int i = t; i < NC / 3; i += NT) {
                    const int base = (i / stride) * stride * 3 + (i % stride);
                    const u32 s0 = z[base], s1 = z[base + stride], s2 = z[base + 2 * stride];
                    z[base] = s0 + s1 + s2;
                    z[base + stride] = s0 + s2;
                    z[base + 2 * stride] = s0 + s1;
                }
                stride *= 3;
            }
            __syncthreads();
        }
        u64 accA = 0, accB = 0;
        for (int c = t; c < NC; c += NT) {
            const u32 h = h0[c], zz = z[c];
            accA += (u64)(h & 0xFFFFu) * (u64)(zz & 0xFFFFu);
            accB += (u64)(h >> 16) * (u64)(zz >> 16);
        }
        for (int o = 32; o > 0; o >>= 1) {
            accA += __shfl_down(accA, o);
            accB += __shfl_down(accB, o);
        }
        if ((t & 63) == 0) { red[t >> 6][0] = accA; red[t >> 6][1] = accB; }
        __syncthreads();
        if (t == 0) {
            u64 oA = 0, oB = 0;
            for (int k = 0; k < NT / 64; ++k) { oA += red[k][0]; oB += red[k][1]; }
            // a curve is compatible with itself iff it ties with the target in every feature (state vector 0)
            const u64 pA = (oA - (u64)(h0[0] & 0xFFFFu)) / 2, pB = (oB - (u64)(h0[0] >> 16)) / 2;
            if (pA) atomicAdd(&out[q0], pA);
            if (two && pB) atomicAdd(&out[q0 + 1], pB);
        }
        __syncthreads();
    }
}

size_t multi_band_workspace_bytes(i64 n, i64 T, int d) {
    const size_t rows = (size_t)T * d;
    return align_up(rows * n * 8, 256) + align_up(rows * n * 8, 256) + 1024;      // time-major copy + (A, B) image
}

bool multi_band_supported(i64 n, i64 T, int d) {
    if (d < 1 || d > 8 || n < 2 || n > 65535 || T < 1) return false;
    size_t lds = 0;
    switch (d) {
        case 1: lds = BECfg<1>::lds_bytes(n); break;
        case 2: lds = BECfg<2>::lds_bytes(n); break;
        case 3: lds = BECfg<3>::lds_bytes(n); break;
        case 4: lds = BECfg<4>::lds_bytes(n); break;
        case 5: lds = BECfg<5>::lds_bytes(n); break;
        case 6: lds = BECfg<6>::lds_bytes(n); break;
        case 7: lds = BECfg<7>::lds_bytes(n); break;
        default: lds = BECfg<8>::lds_bytes(n); break;
    }
    return lds + 512 <= 163840;
}

// P: n x T x d (curve, timepoint, feature) row-major, i.e. the T*d component series of curve a are contiguous: as a
// univariate data set of R = T*d "timepoints" it is curve-major (st = 1, sn = R).
int launch_multi_band(const double *P, i64 n, i64 T, int d, const i64 *targets, i64 m, u64 *out, void *ws, size_t ws_bytes,
                      hipStream_t s) {
    if (!multi_band_supported(n, T, d))
        return fail(SD_ERR_UNSUPPORTED, "componentwise band containment: n=%lld d=%d outside what the LDS holds (n*d*2 + 8*3^d bytes)",
                    (long long)n, d);
    const i64 R = T * d;
    Carver cv(ws, ws_bytes);
    double *Y = (double *)cv.take((size_t)R * n * 8);
    u32 *AB = (u32 *)cv.take((size_t)R * n * 8);
    if (!Y || !AB) return fail(SD_ERR_WORKSPACE, "workspace too small (sd_multi_band_workspace_bytes)");
    int rc;
    if ((rc = launch_to_time_major(P, R, n, 1, R, Y, s))) return rc;
    const bool image = n <= 16384;
    if (image) {
        // every curve's B in every component row: image mode of the bucket kernel (at most 2048 rows per workgroup and launch)
        u32 *nn = AB + (size_t)R * n;                                               // NaN counts per row (unused: NaN-free input)
        const i64 step = 2048 * 64;
        for (i64 r0 = 0; r0 < R; r0 += step)
            if ((rc = launch_rank_bucket_image(Y, n, r0, R - r0 < step ? R - r0 : step, AB + r0 * n, nn + r0, s))) return rc;
    } else if ((rc = launch_above_below(Y, R, n, nullptr, n, AB, s))) {
        return rc;
    }
    SD_HIP(hipMemsetAsync(out, 0, sizeof(u64) * m, s));
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    i64 qg = (2 * (i64)cus + T - 1) / T;                                            // about two blocks per CU in all
    if (qg < 1) qg = 1;
    if (qg > m) qg = m;
    if (qg > 65535) qg = 65535;
    dim3 grid((unsigned)T, (unsigned)qg);
#define BE_CASE(D_)                                                                                                  \
    case D_: {                                                                                                       \
        auto kf = band_class_kernel<D_>;                                                                             \
        const size_t lds = BECfg<D_>::lds_bytes(n);                                                                  \
        SD_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipLaunchKernelGGL(kf, grid, dim3(1024), lds, s, image ? (const u32 *)nullptr : (const u32 *)AB,          \
                           image ? (const u32 *)AB : (const u32 *)nullptr, n, T, targets, m, out);                  \
    } break;
    switch (d) {
        BE_CASE(1) BE_CASE(2) BE_CASE(3) BE_CASE(4) BE_CASE(5) BE_CASE(6) BE_CASE(7) BE_CASE(8)
    }
#undef BE_CASE
    SD_HIP(hipGetLastError());
    return SD_OK;
}

}  // namespace sd
