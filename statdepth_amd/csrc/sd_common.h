// sd_common.h -- shared declarations of the gfx950 band-depth engine (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "statdepth_hip.h"

typedef unsigned long long u64;
typedef int64_t i64;
typedef unsigned int u32;

namespace sd {

constexpr int JMAX_HOST = 8;   // J range of the count kernels

// thread-local error message returned by sd_last_error()
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define SD_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return sd::fail(SD_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                              \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// carve-out helper for the caller-provided workspace
struct Carver {
    char *base;
    size_t off, cap;
    Carver(void *p, size_t bytes) : base((char *)p), off(0), cap(bytes) {}
    void *take(size_t bytes) {
        size_t o = align_up(off, 256);
        if (!base || o + bytes > cap) { off = cap + 1; return nullptr; }
        off = o + bytes;
        return base + o;
    }
    size_t rest() const { size_t o = align_up(off, 256); return o < cap ? cap - o : 0; }   // what the next take() could get
    bool ok() const { return off <= cap; }
};

// Value of an environment switch that selects an alternative implementation for cross-checks (SD_RANK_IMPL,
// SD_BIG_IMPL, SD_SIMPLEX_GENERIC, ...).  The product library (libstatdepth_hip.so) is built without -DSD_CROSSCHECK:
// there this returns 0 for every name and no environment variable is read; libstatdepth_hip_xcheck.so (tests only)
// reads the variable on every call.  xcheck.hip.
long long xswitch(const char *name);

// ---- launchers implemented in the kernel translation units ----
// strided (t,i) -> time-major Y[t*n+i]
int launch_to_time_major(const double *X, i64 T, i64 n, i64 st, i64 sn, double *Y, hipStream_t s);
// per-row NaN counts of a time-major matrix, plus a global any-NaN counter
int launch_nan_count_rows(const double *Y, i64 T, i64 n, u32 *nan_cnt, hipStream_t s);
// K1+K2 pairwise
// targets == nullptr means the contiguous block [tbegin, tbegin + m)
int launch_mbd_pairwise(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                        const u32 *nan_cnt, u64 *out, hipStream_t s);
// external targets Q (T x m, time-major) against the n curves of Y
int launch_mbd_external(const double *Y, i64 T, i64 n, const double *Q, i64 m, int J, const u32 *nan_cnt, u64 *out,
                        hipStream_t s);
int launch_mbd_subsets(const double *Y, i64 T, i64 n, const int *members, i64 nb, int bs, const int *target, int J,
                       u64 *out, hipStream_t s);
int launch_above_below(const double *Y, i64 T, i64 n, const i64 *targets, i64 m, u32 *AB, hipStream_t s);
bool bd_strict_subsets_supported(i64 T, int bs);
size_t bd_strict_subsets_workspace_bytes(i64 T, i64 nb, int bs);
int launch_bd_strict_subsets(const double *Y, i64 T, i64 n, const int *members, i64 nb, int bs, const int *target, u64 *out,
                             void *ws, size_t ws_bytes, hipStream_t s);
// K1+K2 rank formulation
size_t mbd_rank_workspace_bytes(i64 T, i64 n, int J);
bool mbd_rank_supported(i64 T, i64 n, int J);
int launch_mbd_rank(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                    u64 *out, void *ws, size_t ws_bytes, hipStream_t s);
// external targets through the bucket structure (n <= 16384, J <= 3)
bool mbd_rank_external_supported(i64 T, i64 n, i64 m, int J);
size_t mbd_rank_external_workspace_bytes(i64 T, i64 n, i64 m, int J);
int launch_mbd_external_rank(const double *Y, i64 T, i64 n, const double *Q, i64 m, int J, u64 *out, void *ws,
                             size_t ws_bytes, hipStream_t s);
// K1+K2 rank formulation for n > 16384 (chunked)
bool mbd_rank_big_supported(i64 T, i64 n, int J);
// medium sizes (16 384 < n <= 40 960, T >= 96): 2 or 3 column blocks per workgroup, pair image + fold
bool mbd_rank_medium_supported(i64 T, i64 n, int J);
size_t mbd_rank_medium_workspace_bytes(i64 T, i64 n, int J);
int launch_mbd_rank_medium(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J, u64 *out, void *ws,
                           size_t ws_bytes, hipStream_t s);
size_t mbd_rank_big_workspace_bytes(i64 T, i64 n, int J);
int launch_mbd_rank_big(const double *Y, i64 T, i64 n, const i64 *targets, i64 tbegin, i64 m, int J,
                        u64 *out, void *ws, size_t ws_bytes, hipStream_t s);
// K3 strict
size_t bd_strict_workspace_bytes(i64 T, i64 n, i64 m, int J);
size_t bd_strict_min_workspace_bytes(i64 T, i64 n, i64 m, int J);
size_t bd_strict_nanfree_workspace_bytes(i64 T, i64 n, i64 m, int J);
int launch_bd_strict(const double *Y, i64 T, i64 n, const i64 *targets, i64 m, int J,
                     u64 *out, void *ws, size_t ws_bytes, hipStream_t s);
size_t bd_strict_external_workspace_bytes(i64 T, i64 n, i64 m);
int launch_bd_strict_external(const double *Y, i64 T, i64 n, const double *Q, i64 m, u64 *out, void *ws, size_t ws_bytes,
                              hipStream_t s);
// K5 l1
int launch_l1_depth(const double *P, i64 n, int d, const i64 *targets, i64 m, double *out, hipStream_t s);
int launch_l1_external(const double *P, i64 n, int d, const double *Q, i64 m, double *out, hipStream_t s);
int launch_l1_subsets(const double *P, i64 n, int d, const int *members, i64 nb, int bs, double *out, hipStream_t s);
// K4 simplex
int launch_pointcloud_simplex_external(const double *P, i64 n, int d, const double *Q, i64 m, double tol, u64 *out,
                                       hipStream_t s);
int launch_pointcloud_simplex_subsets(const double *P, i64 n, int d, const int *members, i64 nb, int bs, double tol,
                                      u64 *out, hipStream_t s);
int launch_pointcloud_simplex(const double *P, i64 n, int d, const i64 *targets, i64 m, double tol,
                              i64 samples, u64 seed, u64 *out, hipStream_t s, void *ws = nullptr, size_t ws_bytes = 0);
int launch_multi_simplex(const double *P, i64 n, i64 T, int d, const i64 *targets, i64 m, int relax,
                         double tol, i64 samples, u64 seed, u64 *out, hipStream_t s, void *ws = nullptr, size_t ws_bytes = 0);
// records of the sampled estimators' shared factorisation (a batch of (timepoint, sample) pairs)
size_t simplex_sampled_workspace_bytes(i64 n, i64 T, int d, i64 samples);

// K6 componentwise band containment of multivariate curves (band_enum.hip)
bool multi_band_supported(i64 n, i64 T, int d);
size_t multi_band_workspace_bytes(i64 n, i64 T, int d);
int launch_multi_band(const double *P, i64 n, i64 T, int d, const i64 *targets, i64 m, u64 *out, void *ws, size_t ws_bytes,
                      hipStream_t s);

// exact C(a,k) on the host in u64 with overflow detection (returns false on overflow)
bool binom_u64_checked(u64 a, int k, u64 *out);

}  // namespace sd

// ---- device helpers -------------------------------------------------------
#ifdef __HIPCC__
namespace sd {

constexpr int JMAX = 8;

// C(a,k) for k = 0..K by the exact recurrence C(a,k) = C(a,k-1)*(a-k+1)/k.
// Caller guarantees k*C(a,k) < 2^64 (checked on the host from n, J, T).
template <int K>
__device__ __forceinline__ void binoms(u64 a, u64 (&c)[K + 1]) {
    c[0] = 1;
#pragma unroll
    for (int k = 1; k <= K; ++k) c[k] = c[k - 1] * (a - (u64)(k - 1)) / (u64)k;
}

// number of j-subsets (j = 2..J) of the n-1 other curves whose band contains the
// target at one timepoint, from A (above), B (below), N (NaN others), nm1 = n-1.
// acc[j-2] += count.  Restates oracle_mbd_counts (oracle/oracle.c).
template <int J>
__device__ __forceinline__ void band_counts_add(u32 A, u32 B, u32 N, u64 nm1, u64 (&acc)[JMAX - 1]) {
    u64 v = nm1 - N;
    if (N == 0) {
        if constexpr (J == 2) {
            acc[0] += (v * (v - 1) >> 1) - ((u64)A * (A - 1) >> 1) - ((u64)B * (B - 1) >> 1);
        } else {
            u64 cv[J + 1], ca[J + 1], cb[J + 1];
            binoms<J>(v, cv);
            binoms<J>(A, ca);
            binoms<J>(B, cb);
#pragma unroll
            for (int j = 2; j <= J; ++j) acc[j - 2] += cv[j] - ca[j] - cb[j];
        }
    } else {
        u64 cv[J + 1], ca[J + 1], cb[J + 1], cn[J + 1];
        binoms<J>(v, cv);
        binoms<J>(A, ca);
        binoms<J>(B, cb);
        binoms<J>(N, cn);
#pragma unroll
        for (int j = 2; j <= J; ++j) {
            u64 s = 0;
#pragma unroll
            for (int k = 1; k <= j; ++k) s += cn[j - k] * (cv[k] - ca[k] - cb[k]);
            acc[j - 2] += s;
        }
    }
}

// dispatch a runtime J in [2, JMAX] to a template instantiation
#define SD_DISPATCH_J(J, ...)                         \
    switch (J) {                                      \
        case 2: { constexpr int J_ = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int J_ = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int J_ = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int J_ = 5; __VA_ARGS__; } break; \
        case 6: { constexpr int J_ = 6; __VA_ARGS__; } break; \
        case 7: { constexpr int J_ = 7; __VA_ARGS__; } break; \
        case 8: { constexpr int J_ = 8; __VA_ARGS__; } break; \
        default: return sd::fail(SD_ERR_INVALID, "J=%d outside [2,8]", (int)(J)); \
    }

}  // namespace sd
#endif
