// bd_strict.hip -- K3: strict band depth (relax=False).
//
// Replaces the subset loop of _univariate_band_depth (_functional.py:246-251) with
// `containment // len(curve)` (_containment.py:80): a j-subset of the other curves
// counts only if its band contains the target at EVERY timepoint.
//
// Per target, every curve i gets two T-bit masks over the timepoints:
//   UN_i[t] = (x_i > x_q) or x_i is NaN,   DN_i[t] = (x_i < x_q) or x_i is NaN.
// With pandas' skipna min/max (_containment.py:68-69) a subset fails at t iff all
// its members are in UN or all are in DN, so it is contained at every t iff
//   AND_members(UN) == 0 and AND_members(DN) == 0      (as T-bit masks).
// A NaN in the target fails everything (count 0).
// Phase 1 builds the masks (coalesced row reads, target value wave-uniform);
// phase 2 counts subsets: lanes = curve a (masks in VGPRs), partner b wave-uniform
// through the scalar cache.  Integer work, VALU-bound (64-bit AND/OR), no MFMA.
#include "sd_common.h"

namespace sd {

constexpr int ST_THREADS = 256;
constexpr int ST_WREG = 16;          // mask words kept in registers (T <= 1024)

static inline i64 strict_words(i64 T) { return (T + 63) / 64; }

static i64 strict_batch(i64 T, i64 n, i64 m) {
    size_t per = (size_t)n * 2 * strict_words(T) * 8;
    i64 b = (i64)(((size_t)256 << 20) / (per ? per : 1));
    if (b < 1) b = 1;
    if (b > m) b = m;
    if (b > 65535) b = 65535;
    return b;
}

size_t bd_strict_workspace_bytes(i64 T, i64 n, i64 m, int J) {
    (void)J;
    i64 b = strict_batch(T, n, m);
    return align_up((size_t)b * n * 2 * strict_words(T) * 8, 256) + align_up((size_t)b * 4, 256) + 512;
}

// masks[b][i][0..W) = UN, masks[b][i][W..2W) = DN
__global__ __launch_bounds__(ST_THREADS) void strict_masks_kernel(
    const double *__restrict__ Y, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0,
    u64 *__restrict__ masks, u32 *__restrict__ xnan) {
    i64 i = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    i64 b = blockIdx.y;
    i64 q = q0 + b;
    i64 tg = targets ? targets[q] : q;
    i64 W = (T + 63) / 64;
    u64 *mrow = masks + ((size_t)b * n + (i < n ? i : 0)) * 2 * W;
    bool anynan = false;
    for (i64 w = 0; w < W; ++w) {
        u64 un = 0, dn = 0;
        i64 tend = (w + 1) * 64 < T ? (w + 1) * 64 : T;
        for (i64 t = w * 64; t < tend; ++t) {
            double xq = Y[t * n + tg];
            double xi = i < n ? Y[t * n + i] : 0.0;
            anynan |= (xq != xq);
            u64 bit = (u64)1 << (t & 63);
            bool isn = xi != xi;
            if (xi > xq || isn) un |= bit;
            if (xi < xq || isn) dn |= bit;
        }
        if (i < n) {
            mrow[w] = un;
            mrow[W + w] = dn;
        }
    }
    if (anynan && blockIdx.x == 0 && threadIdx.x == 0) xnan[b] = 1;
}

template <typename Tv>
__device__ __forceinline__ Tv block_sum(Tv v, Tv *scratch) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    Tv r = 0;
    if (threadIdx.x == 0)
        for (unsigned k = 0; k < (blockDim.x + 63) / 64; ++k) r += scratch[k];
    return r;
}

// J = 2: pairs (a < b).  grid = (a tiles, b chunks, batch)
constexpr int ST_BCHUNK = 512;

template <bool REG>
__global__ __launch_bounds__(ST_THREADS) void strict_pairs_kernel(
    const u64 *__restrict__ masks, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0,
    const u32 *__restrict__ xnan, u64 *__restrict__ out, int jcols) {
    __shared__ u64 scratch[ST_THREADS / 64];
    i64 b = blockIdx.z;
    i64 q = q0 + b;
    if (xnan[b]) return;                       // NaN in the target: nothing is contained
    i64 tg = targets ? targets[q] : q;
    i64 a = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    i64 b0 = (i64)blockIdx.y * ST_BCHUNK;
    i64 b1 = b0 + ST_BCHUNK < n ? b0 + ST_BCHUNK : n;
    i64 amin = (i64)blockIdx.x * ST_THREADS;
    if (b1 - 1 <= amin) return;                // whole chunk at or below the tile: no a < b pair
    int W = (int)((T + 63) / 64);
    const u64 *mb = masks + (size_t)b * n * 2 * W;
    bool alive = a < n && a != tg;
    const u64 *ma = mb + (size_t)(alive ? a : 0) * 2 * W;
    u64 un[ST_WREG], dn[ST_WREG];
    if constexpr (REG) {
#pragma unroll
        for (int w = 0; w < ST_WREG; ++w) {
            un[w] = (w < W) ? ma[w] : 0;
            dn[w] = (w < W) ? ma[W + w] : 0;
        }
    }
    u64 good = 0;
    for (i64 c = b0; c < b1; ++c) {
        if (c == tg) continue;
        const u64 *mc = mb + (size_t)c * 2 * W;   // wave-uniform: scalar loads
        // lanes that cannot count any more (dead lane, a >= c, or already in conflict with c) are "settled";
        // once the whole wave is settled the remaining words are skipped.  Most pairs conflict in the first
        // 64 timepoints, so this cuts the mask work by up to W (16 at T = 1000).
        u64 bad = (alive && a < c) ? 0 : ~0ull;
        if constexpr (REG) {
#pragma unroll
            for (int w = 0; w < ST_WREG; ++w) {
                if (w < W) {
                    bad |= (un[w] & mc[w]) | (dn[w] & mc[W + w]);
                    if ((w & 1) == 0 && __ballot(bad == 0) == 0) break;
                }
            }
        } else {
            for (int w = 0; w < W; ++w) {
                bad |= (ma[w] & mc[w]) | (ma[W + w] & mc[W + w]);
                if (__ballot(bad == 0) == 0) break;
            }
        }
        good += (bad == 0);
    }
    u64 tot = block_sum(good, scratch);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[q * jcols], tot);
}

// J = 3 / 4: one thread per (J-1)-prefix, loop over the last member.
template <int J>
__global__ __launch_bounds__(ST_THREADS) void strict_subsets_kernel(
    const u64 *__restrict__ masks, i64 T, i64 n, const i64 *__restrict__ targets, i64 q0,
    const u32 *__restrict__ xnan, u64 *__restrict__ out, int jcols) {
    __shared__ u64 scratch[ST_THREADS / 64];
    i64 b = blockIdx.z;
    i64 q = q0 + b;
    if (xnan[b]) return;
    i64 tg = targets ? targets[q] : q;
    int W = (int)((T + 63) / 64);
    const u64 *mb = masks + (size_t)b * n * 2 * W;
    // prefix (i0 < i1 [< i2]) from a flat index over n^(J-1)
    i64 flat = (i64)blockIdx.x * ST_THREADS + threadIdx.x;
    i64 idx[3];
    bool ok = true;
    i64 f = flat;
    for (int k = J - 2; k >= 0; --k) { idx[k] = f % n; f /= n; }
    if (f != 0) ok = false;
    for (int k = 0; k < J - 1; ++k) {
        if (idx[k] == tg) ok = false;
        if (k > 0 && idx[k] <= idx[k - 1]) ok = false;
    }
    u64 good = 0;
    if (ok) {
        for (i64 c = idx[J - 2] + 1; c < n; ++c) {
            if (c == tg) continue;
            u64 bad = 0;
            for (int w = 0; w < W; ++w) {
                u64 u = mb[(size_t)c * 2 * W + w], d = mb[(size_t)c * 2 * W + W + w];
                for (int k = 0; k < J - 1; ++k) {
                    u &= mb[(size_t)idx[k] * 2 * W + w];
                    d &= mb[(size_t)idx[k] * 2 * W + W + w];
                }
                bad |= u | d;
            }
            good += (bad == 0);
        }
    }
    u64 tot = block_sum(good, scratch);
    if (threadIdx.x == 0 && tot) atomicAdd(&out[q * jcols + (J - 2)], tot);
}

int launch_bd_strict(const double *Y, i64 T, i64 n, const i64 *targets, i64 m, int J,
                     u64 *out, void *ws, size_t ws_bytes, hipStream_t s) {
    i64 W = strict_words(T);
    i64 B = strict_batch(T, n, m);
    Carver cv(ws, ws_bytes);
    u64 *masks = (u64 *)cv.take((size_t)B * n * 2 * W * 8);
    u32 *xnan = (u32 *)cv.take((size_t)B * 4);
    if (!masks || !xnan) return fail(SD_ERR_WORKSPACE, "strict-depth workspace too small");
    int jcols = J - 1;
    SD_HIP(hipMemsetAsync(out, 0, sizeof(u64) * m * jcols, s));
    if (J >= 3) {
        double threads = 1.0;
        for (int k = 0; k < J - 1; ++k) threads *= (double)n;
        if (threads > 4.0e9) return fail(SD_ERR_UNSUPPORTED, "strict J=%d enumeration too large for n=%lld", J, (long long)n);
    }
    for (i64 q0 = 0; q0 < m; q0 += B) {
        i64 nb = m - q0 < B ? m - q0 : B;
        SD_HIP(hipMemsetAsync(xnan, 0, (size_t)nb * 4, s));
        dim3 g1((unsigned)((n + ST_THREADS - 1) / ST_THREADS), (unsigned)nb);
        hipLaunchKernelGGL(strict_masks_kernel, g1, dim3(ST_THREADS), 0, s, Y, T, n, targets, q0, masks, xnan);
        dim3 g2((unsigned)((n + ST_THREADS - 1) / ST_THREADS), (unsigned)((n + ST_BCHUNK - 1) / ST_BCHUNK), (unsigned)nb);
        if (W <= ST_WREG)
            hipLaunchKernelGGL((strict_pairs_kernel<true>), g2, dim3(ST_THREADS), 0, s, masks, T, n, targets, q0, xnan, out, jcols);
        else
            hipLaunchKernelGGL((strict_pairs_kernel<false>), g2, dim3(ST_THREADS), 0, s, masks, T, n, targets, q0, xnan, out, jcols);
        if (J >= 3) {
            i64 flat = n * n;
            dim3 g3((unsigned)((flat + ST_THREADS - 1) / ST_THREADS), 1, (unsigned)nb);
            hipLaunchKernelGGL((strict_subsets_kernel<3>), g3, dim3(ST_THREADS), 0, s, masks, T, n, targets, q0, xnan, out, jcols);
        }
        if (J >= 4) {
            i64 flat = n * n * n;
            dim3 g4((unsigned)((flat + ST_THREADS - 1) / ST_THREADS), 1, (unsigned)nb);
            hipLaunchKernelGGL((strict_subsets_kernel<4>), g4, dim3(ST_THREADS), 0, s, masks, T, n, targets, q0, xnan, out, jcols);
        }
        SD_HIP(hipGetLastError());
    }
    return SD_OK;
}

}  // namespace sd
