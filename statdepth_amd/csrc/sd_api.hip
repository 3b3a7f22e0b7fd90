// sd_api.hip -- the extern "C" boundary declared in include/statdepth_hip.h.
// Argument validation, workspace carving and kernel sequencing; no arithmetic.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "sd_common.h"

namespace sd {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

bool binom_u64_checked(u64 a, int k, u64 *out) {
    if (k < 0 || (u64)k > a) { *out = 0; return true; }
    unsigned __int128 c = 1;
    for (int j = 1; j <= k; ++j) {
        c = c * (a - (u64)j + 1) / (u64)j;
        if (c >> 64) return false;
    }
    *out = (u64)c;
    return true;
}

// J, n, T limits shared by the count kernels: totals must fit int64 and the
// device recurrence C(a,k) = C(a,k-1)*(a-k+1)/k must not overflow u64.
static int check_count_range(i64 T, i64 n, int J) {
    if (J < 2 || J > JMAX_HOST) return fail(SD_ERR_INVALID, "J=%d outside [2,%d]", J, JMAX_HOST);
    u64 c;
    if (!binom_u64_checked((u64)(n > 0 ? n - 1 : 0), J, &c))
        return fail(SD_ERR_OVERFLOW, "C(n-1,J) overflows 64 bits (n=%lld, J=%d)", (long long)n, J);
    unsigned __int128 tot = (unsigned __int128)c * (u64)(T > J ? T : J);
    if (tot >> 63)
        return fail(SD_ERR_OVERFLOW, "T*C(n-1,J) >= 2^63 (T=%lld, n=%lld, J=%d): int64 totals would overflow",
                    (long long)T, (long long)n, J);
    return SD_OK;
}

static int check_matrix(const double *X, i64 T, i64 n, i64 st, i64 sn, const i64 *targets, i64 m) {
    if (!X) return fail(SD_ERR_INVALID, "X is null");
    if (T <= 0 || n <= 0) return fail(SD_ERR_INVALID, "empty matrix (T=%lld, n=%lld)", (long long)T, (long long)n);
    if (n >= ((i64)1 << 31) || T >= ((i64)1 << 31))
        return fail(SD_ERR_UNSUPPORTED, "T and n must be < 2^31");
    if (st <= 0 || sn <= 0) return fail(SD_ERR_INVALID, "strides must be positive (st=%lld, sn=%lld)", (long long)st, (long long)sn);
    if (!((sn == 1 && st >= n) || (st == 1 && sn >= T)))
        return fail(SD_ERR_INVALID, "matrix must be time-major (sn=1, st>=n) or curve-major (st=1, sn>=T)");
    if (m < 0) return fail(SD_ERR_INVALID, "m < 0");
    if (targets == nullptr && m != n) return fail(SD_ERR_INVALID, "targets=NULL requires m == n");
    return SD_OK;
}

static bool is_time_major_dense(i64 n, i64 st, i64 sn) { return sn == 1 && st == n; }

// wide[i] (two 64-bit limbs) (+)= part[i]: the chunk totals of sd_mbd_counts_wide
__global__ void wide_add_kernel(const u64 *__restrict__ part, i64 count, u64 *__restrict__ wide, int first) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64 v = part[i];
    if (first) {
        wide[2 * i] = v;
        wide[2 * i + 1] = 0;
    } else {
        const u64 lo = wide[2 * i] + v;
        wide[2 * i + 1] += lo < v ? 1u : 0u;
        wide[2 * i] = lo;
    }
}


}  // namespace sd

using namespace sd;

extern "C" {

int sd_abi_version(void) { return SD_ABI_VERSION; }

const char *sd_last_error(void) { return g_err; }

int sd_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int sd_device_info(int dev, char *name, int name_len, int *cu_count, size_t *hbm_bytes) {
    hipDeviceProp_t p;
    SD_HIP(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return SD_OK;
}

int sd_set_device(int dev) {
    SD_HIP(hipSetDevice(dev));
    return SD_OK;
}

int sd_malloc(void **dptr, size_t bytes) {
    if (!dptr) return fail(SD_ERR_INVALID, "dptr is null");
    SD_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return SD_OK;
}

int sd_free(void *dptr) {
    if (dptr) SD_HIP(hipFree(dptr));
    return SD_OK;
}

int sd_memcpy_h2d(void *dst, const void *src_host, size_t bytes, void *stream) {
    SD_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return SD_OK;
}

int sd_memcpy_d2h(void *dst_host, const void *src, size_t bytes, void *stream) {
    SD_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SD_OK;
}

int sd_memset(void *dst, int value, size_t bytes, void *stream) {
    SD_HIP(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return SD_OK;
}

int sd_stream_synchronize(void *stream) {
    SD_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SD_OK;
}

// ---------------------------------------------------------------------------
// K1+K2
// ---------------------------------------------------------------------------
static int resolve_mbd_algo(int algo, i64 T, i64 n, i64 m, int J) {
    if (algo == SD_MBD_PAIRWISE || algo == SD_MBD_RANK) return algo;
    // rank costs ~n log n per timepoint whatever m is; pairwise costs m*n.
    if (mbd_rank_supported(T, n, J) && m >= 32) return SD_MBD_RANK;
    // chunked rank: work ~ (n/C) sorts + (n/C)^2 searches per row whatever m is; pairwise costs m*n
    if (mbd_rank_big_supported(T, n, J) && m >= 2048) return SD_MBD_RANK;
    return SD_MBD_PAIRWISE;
}

size_t sd_mbd_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m, int J, int algo) {
    if (T <= 0 || n <= 0) return 0;
    size_t b = 0;
    if (!is_time_major_dense(n, st, sn)) b += align_up((size_t)T * n * 8, 256);
    b += align_up((size_t)T * 4, 256);   // nan_cnt
    int a = resolve_mbd_algo(algo, T, n, m, J);
    if (a == SD_MBD_RANK || algo == SD_MBD_AUTO)
        b += align_up(mbd_rank_workspace_bytes(T, n, J) + mbd_rank_big_workspace_bytes(T, n, J) +
                      mbd_rank_medium_workspace_bytes(T, n, J), 256) + 256;
    return b + 1024;
}

static int mbd_counts_impl(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                           const int64_t *targets, int64_t tbegin, int64_t m, int J, int algo,
                           int64_t *out, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_matrix(X, T, n, st, sn, targets ? targets : (const i64 *)1, m);
    if (rc) return rc;
    if (!targets && (tbegin < 0 || tbegin + m > n))
        return fail(SD_ERR_INVALID, "target block [%lld, %lld) outside [0, %lld)", (long long)tbegin,
                    (long long)(tbegin + m), (long long)n);
    if ((rc = check_count_range(T, n, J))) return rc;
    if (!out) return fail(SD_ERR_INVALID, "out is null");
    if (algo < SD_MBD_AUTO || algo > SD_MBD_RANK) return fail(SD_ERR_INVALID, "unknown algo %d", algo);
    if (m == 0) return SD_OK;
    hipStream_t s = (hipStream_t)stream;
    Carver cv(ws, ws_bytes);
    const double *Y = X;
    if (!is_time_major_dense(n, st, sn)) {
        double *Yw = (double *)cv.take((size_t)T * n * 8);
        if (!Yw) return fail(SD_ERR_WORKSPACE, "workspace too small for the time-major copy");
        if ((rc = launch_to_time_major(X, T, n, st, sn, Yw, s))) return rc;
        Y = Yw;
    }
    int a = resolve_mbd_algo(algo, T, n, m, J);
    if (a == SD_MBD_RANK && mbd_rank_medium_supported(T, n, J)) {
        size_t need = mbd_rank_medium_workspace_bytes(T, n, J);
        void *rws = cv.take(need);
        if (!rws) return fail(SD_ERR_WORKSPACE, "workspace too small for the medium rank route");
        return launch_mbd_rank_medium(Y, T, n, targets, tbegin, m, J, (u64 *)out, rws, need, s);
    }
    if (a == SD_MBD_RANK && mbd_rank_big_supported(T, n, J)) {
        size_t need = mbd_rank_big_workspace_bytes(T, n, J);
        void *rws = cv.take(need);
        if (!rws) return fail(SD_ERR_WORKSPACE, "workspace too small for the chunked rank kernel");
        return launch_mbd_rank_big(Y, T, n, targets, tbegin, m, J, (u64 *)out, rws, need, s);
    }
    if (a == SD_MBD_RANK) {
        if (!mbd_rank_supported(T, n, J))
            return fail(SD_ERR_UNSUPPORTED, "rank kernel does not cover T=%lld n=%lld J=%d", (long long)T, (long long)n, J);
        size_t need = mbd_rank_workspace_bytes(T, n, J);
        void *rws = cv.take(need);
        if (need && !rws) return fail(SD_ERR_WORKSPACE, "workspace too small for the rank kernel");
        return launch_mbd_rank(Y, T, n, targets, tbegin, m, J, (u64 *)out, rws, need, s);
    }
    u32 *nan_cnt = (u32 *)cv.take((size_t)T * 4);
    if (!nan_cnt) return fail(SD_ERR_WORKSPACE, "workspace too small (nan counts)");
    if ((rc = launch_nan_count_rows(Y, T, n, nan_cnt, s))) return rc;
    return launch_mbd_pairwise(Y, T, n, targets, tbegin, m, J, nan_cnt, (u64 *)out, s);
}

int sd_mbd_counts(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                  const int64_t *targets, int64_t m, int J, int algo,
                  int64_t *out, void *ws, size_t ws_bytes, void *stream) {
    if (!targets && m != n) return fail(SD_ERR_INVALID, "targets=NULL requires m == n");
    return mbd_counts_impl(X, T, n, st, sn, targets, 0, m, J, algo, out, ws, ws_bytes, stream);
}

int sd_mbd_counts_range(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                        int64_t target_begin, int64_t m, int J, int algo,
                        int64_t *out, void *ws, size_t ws_bytes, void *stream) {
    return mbd_counts_impl(X, T, n, st, sn, nullptr, target_begin, m, J, algo, out, ws, ws_bytes, stream);
}

// ---- two-limb totals: T * C(n-1, J) beyond int64 (J >= 4 at n = 10^5) -------------------------------------------
// The kernels accumulate per-timepoint band counts (each < 2^64 by check_count_range's recurrence bound) in 64 bits;
// here the timepoints are cut into chunks whose totals provably fit int64, every chunk runs through the ordinary path,
// and wide_add_kernel adds the chunk's totals into (lo, hi) pairs with carry.
size_t sd_mbd_wide_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m, int J, int algo) {
    if (T <= 0 || n <= 0 || m < 0 || J < 2) return 0;
    return sd_mbd_workspace_bytes(T, n, st, sn, m, J, algo) + align_up((size_t)m * (J - 1) * 8, 256) + 256;
}

int sd_mbd_counts_wide(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                       const int64_t *targets, int64_t m, int J, int algo,
                       uint64_t *out, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_matrix(X, T, n, st, sn, targets, m);
    if (rc) return rc;
    if (!out) return fail(SD_ERR_INVALID, "out is null");
    if (J < 2 || J > JMAX_HOST) return fail(SD_ERR_INVALID, "J=%d outside [2,%d]", J, JMAX_HOST);
    if (m == 0) return SD_OK;
    // rows per chunk: the largest Tc with Tc * C(n-1, J) < 2^63 (check_count_range's own condition)
    u64 c;
    if (!binom_u64_checked((u64)(n - 1), J, &c))
        return fail(SD_ERR_OVERFLOW, "C(n-1,J) itself overflows 64 bits (n=%lld, J=%d): beyond two-limb totals too",
                    (long long)n, J);
    i64 Tc = c ? (i64)((((u64)1 << 63) - 1) / c) : T;
    if (Tc > T) Tc = T;
    while (Tc > (i64)J && check_count_range(Tc, n, J) != SD_OK) --Tc;      // the check uses max(T, J) rows
    if (Tc < 1 || check_count_range(Tc, n, J) != SD_OK)
        return fail(SD_ERR_OVERFLOW, "J*C(n-1,J) >= 2^63 (n=%lld, J=%d): a single chunk of timepoints does not fit int64",
                    (long long)n, J);
    Carver cv(ws, ws_bytes);
    u64 *part = (u64 *)cv.take((size_t)m * (J - 1) * 8);
    const size_t inner = sd_mbd_workspace_bytes(Tc, n, st, sn, m, J, algo);
    void *iws = cv.take(inner);
    if (!part || !iws) return fail(SD_ERR_WORKSPACE, "workspace too small (sd_mbd_wide_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    const i64 count = m * (J - 1);
    for (i64 t0 = 0; t0 < T; t0 += Tc) {
        const i64 rows = T - t0 < Tc ? T - t0 : Tc;
        // rows [t0, t0 + rows) of either layout: x(t, i) = X[t*st + i*sn]; a curve-major block keeps its curve stride
        if ((rc = mbd_counts_impl(X + t0 * st, rows, n, st, sn, targets, 0, m, J, algo, (int64_t *)part, iws, inner, stream)))
            return rc;
        hipLaunchKernelGGL(wide_add_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, part, count,
                           (u64 *)out, t0 == 0 ? 1 : 0);
        SD_HIP(hipGetLastError());
    }
    return SD_OK;
}

int sd_mbd_external_counts(const double *X, int64_t T, int64_t n, const double *Q, int64_t m, int J,
                           int64_t *out, void *ws, size_t ws_bytes, void *stream) {
    if (!X || !Q || !out) return fail(SD_ERR_INVALID, "null pointer");
    if (T <= 0 || n <= 0 || m < 0) return fail(SD_ERR_INVALID, "bad shape (T=%lld, n=%lld, m=%lld)", (long long)T,
                                                (long long)n, (long long)m);
    int rc = check_count_range(T, n + 1, J);
    if (rc) return rc;
    if (m == 0) return SD_OK;
    hipStream_t s = (hipStream_t)stream;
    // bucket structure per row + one look-up per target (O(n + m) per row) when the caller gave the workspace for it
    // (sd_mbd_external_workspace_bytes); the pairwise kernel (O(n m) per row) otherwise
    if (mbd_rank_external_supported(T, n, m, J) && m >= 16 && ws &&
        ws_bytes >= mbd_rank_external_workspace_bytes(T, n, m, J) + 256)
        return launch_mbd_external_rank(X, T, n, Q, m, J, (u64 *)out, ws, ws_bytes, s);
    Carver cv(ws, ws_bytes);
    u32 *nan_cnt = (u32 *)cv.take((size_t)T * 4);
    if (!nan_cnt) return fail(SD_ERR_WORKSPACE, "workspace too small (need T*4 + 256 bytes)");
    if ((rc = launch_nan_count_rows(X, T, n, nan_cnt, s))) return rc;
    return launch_mbd_external(X, T, n, Q, m, J, nan_cnt, (u64 *)out, s);
}

size_t sd_mbd_external_workspace_bytes(int64_t T, int64_t n, int64_t m, int J) {
    if (T <= 0 || n <= 0 || m <= 0) return 0;
    size_t pw = align_up((size_t)T * 4, 256) + 256;                       // pairwise kernel: per-row NaN counts
    size_t rk = mbd_rank_external_workspace_bytes(T, n, m, J);           // 0 where the bucket kernel does not apply
    return (rk ? rk + 256 : 0) > pw ? rk + 256 : pw;
}

int sd_mbd_subset_counts(const double *X, int64_t T, int64_t n, const int32_t *members, int64_t nb, int bs,
                         const int32_t *target, int J, int64_t *out, void *stream) {
    if (!X || !members || !target || !out) return fail(SD_ERR_INVALID, "null pointer");
    if (T <= 0 || n <= 0 || nb < 0 || bs <= 0) return fail(SD_ERR_INVALID, "bad shape");
    int rc = check_count_range(T, bs, J);
    if (rc) return rc;
    if (nb == 0) return SD_OK;
    return launch_mbd_subsets(X, T, n, members, nb, bs, target, J, (u64 *)out, (hipStream_t)stream);
}

int sd_above_below(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                   const int64_t *targets, int64_t m, uint32_t *AB,
                   void *ws, size_t ws_bytes, void *stream) {
    int rc = check_matrix(X, T, n, st, sn, targets, m);
    if (rc) return rc;
    if (!AB) return fail(SD_ERR_INVALID, "AB is null");
    if (m == 0) return SD_OK;
    hipStream_t s = (hipStream_t)stream;
    Carver cv(ws, ws_bytes);
    const double *Y = X;
    if (!is_time_major_dense(n, st, sn)) {
        double *Yw = (double *)cv.take((size_t)T * n * 8);
        if (!Yw) return fail(SD_ERR_WORKSPACE, "workspace too small for the time-major copy");
        if ((rc = launch_to_time_major(X, T, n, st, sn, Yw, s))) return rc;
        Y = Yw;
    }
    return launch_above_below(Y, T, n, targets, m, AB, s);
}

// ---------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------
size_t sd_bd_strict_j_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m, int J) {
    if (T <= 0 || n <= 0) return 0;
    size_t b = 0;
    if (!is_time_major_dense(n, st, sn)) b += align_up((size_t)T * n * 8, 256);
    return b + bd_strict_workspace_bytes(T, n, m, J) + 1024;
}

size_t sd_bd_strict_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m) {
    return sd_bd_strict_j_workspace_bytes(T, n, st, sn, m, 2);
}

size_t sd_bd_strict_nanfree_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m) {
    if (T <= 0 || n <= 0) return 0;
    size_t b = 0;
    if (!is_time_major_dense(n, st, sn)) b += align_up((size_t)T * n * 8, 256);
    return b + bd_strict_nanfree_workspace_bytes(T, n, m, 2) + 1024;
}

size_t sd_bd_strict_min_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m, int J) {
    if (T <= 0 || n <= 0) return 0;
    size_t b = 0;
    if (!is_time_major_dense(n, st, sn)) b += align_up((size_t)T * n * 8, 256);
    return b + bd_strict_min_workspace_bytes(T, n, m, J) + 1024;
}

int sd_bd_strict_j_counts(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                          const int64_t *targets, int64_t m, int J,
                          int64_t *out, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_matrix(X, T, n, st, sn, targets, m);
    if (rc) return rc;
    if (J < 2 || J > 4) return fail(SD_ERR_UNSUPPORTED, "strict band depth covers J in [2,4], got %d", J);
    if ((rc = check_count_range(1, n, J))) return rc;
    if (!out) return fail(SD_ERR_INVALID, "out is null");
    if (m == 0) return SD_OK;
    hipStream_t s = (hipStream_t)stream;
    Carver cv(ws, ws_bytes);
    const double *Y = X;
    if (!is_time_major_dense(n, st, sn)) {
        double *Yw = (double *)cv.take((size_t)T * n * 8);
        if (!Yw) return fail(SD_ERR_WORKSPACE, "workspace too small for the time-major copy");
        if ((rc = launch_to_time_major(X, T, n, st, sn, Yw, s))) return rc;
        Y = Yw;
    }
    size_t need = cv.rest();                                   // batches are sized to what the caller passed
    void *sws = cv.take(need);
    // 6 ... 8 timepoints, J = 2: NaN-free data is counted through state classes and takes 256 bytes (a flag); whether the data
    // is NaN-free the launcher finds out itself, and data with NaN then meets the mask pipeline's own check of what it was given
    const size_t floor = (J == 2 && T >= 6 && T <= 8) ? (size_t)256 : bd_strict_min_workspace_bytes(T, n, m, J);
    if (!sws || need < floor)
        return fail(SD_ERR_WORKSPACE, "workspace too small for the strict-depth masks (sd_bd_strict_min_workspace_bytes)");
    return launch_bd_strict(Y, T, n, targets, m, J, (u64 *)out, sws, need, s);
}

size_t sd_bd_strict_external_workspace_bytes(int64_t T, int64_t n, int64_t m) {
    if (T <= 0 || n <= 0 || m <= 0) return 0;
    return bd_strict_external_workspace_bytes(T, n, m) + 512;
}

int sd_bd_strict_external_counts(const double *X, int64_t T, int64_t n, const double *Q, int64_t m, int64_t *out, void *ws,
                                 size_t ws_bytes, void *stream) {
    if (!X || !Q || !out) return fail(SD_ERR_INVALID, "null pointer");
    if (T <= 0 || n <= 0 || m < 0) return fail(SD_ERR_INVALID, "bad shape");
    int rc = check_count_range(1, n + 1, 2);
    if (rc) return rc;
    if (m == 0) return SD_OK;
    return launch_bd_strict_external(X, T, n, Q, m, (u64 *)out, ws, ws_bytes, (hipStream_t)stream);
}

size_t sd_bd_strict_subset_workspace_bytes(int64_t T, int64_t nb, int bs) {
    if (T <= 0 || nb <= 0 || bs <= 0) return 0;
    const size_t b = bd_strict_subsets_workspace_bytes(T, nb, bs);
    return b ? b + 512 : 0;
}

int sd_bd_strict_subset_counts(const double *X, int64_t T, int64_t n, const int32_t *members, int64_t nb, int bs,
                               const int32_t *target, int64_t *out, void *ws, size_t ws_bytes, void *stream) {
    if (!X || !members || !target || !out) return fail(SD_ERR_INVALID, "null pointer");
    if (T <= 0 || n <= 0 || nb < 0 || bs <= 0) return fail(SD_ERR_INVALID, "bad shape");
    int rc = check_count_range(1, bs, 2);
    if (rc) return rc;
    if (nb == 0) return SD_OK;
    return launch_bd_strict_subsets(X, T, n, members, nb, bs, target, (u64 *)out, ws, ws_bytes, (hipStream_t)stream);
}

int sd_bd_strict_subset_supported(int64_t T, int bs) { return T > 0 && bs > 0 && bd_strict_subsets_supported(T, bs) ? 1 : 0; }

int sd_bd_strict_counts(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                        const int64_t *targets, int64_t m,
                        int64_t *out, void *ws, size_t ws_bytes, void *stream) {
    return sd_bd_strict_j_counts(X, T, n, st, sn, targets, m, 2, out, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------
// K5
// ---------------------------------------------------------------------------
int sd_l1_depth(const double *P, int64_t n, int d, const int64_t *targets, int64_t m,
                double *out, void *stream) {
    if (!P || !out) return fail(SD_ERR_INVALID, "null pointer");
    if (n <= 0 || d <= 0) return fail(SD_ERR_INVALID, "empty point cloud");
    if (d > 64) return fail(SD_ERR_UNSUPPORTED, "l1 depth covers d <= 64");
    if (!targets && m != n) return fail(SD_ERR_INVALID, "targets=NULL requires m == n");
    if (m == 0) return SD_OK;
    return launch_l1_depth(P, n, d, targets, m, out, (hipStream_t)stream);
}

int sd_l1_external_depth(const double *P, int64_t n, int d, const double *Q, int64_t m, double *out, void *stream) {
    if (!P || !Q || !out) return fail(SD_ERR_INVALID, "null pointer");
    if (n <= 0 || d <= 0 || m < 0) return fail(SD_ERR_INVALID, "bad shape");
    if (d > 64) return fail(SD_ERR_UNSUPPORTED, "l1 depth covers d <= 64");
    if (m == 0) return SD_OK;
    return launch_l1_external(P, n, d, Q, m, out, (hipStream_t)stream);
}

int sd_l1_subset_depth(const double *P, int64_t n, int d, const int32_t *members, int64_t nb, int bs, double *out,
                       void *stream) {
    if (!P || !members || !out) return fail(SD_ERR_INVALID, "null pointer");
    if (n <= 0 || d <= 0 || nb < 0 || bs <= 0) return fail(SD_ERR_INVALID, "bad shape");
    if (d > 64) return fail(SD_ERR_UNSUPPORTED, "l1 depth covers d <= 64");
    if (nb == 0) return SD_OK;
    return launch_l1_subsets(P, n, d, members, nb, bs, out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// K4
// ---------------------------------------------------------------------------
static int check_simplex(const double *P, i64 n, int d, const i64 *targets, i64 m, const void *out, bool exhaustive,
                         i64 n_others) {
    if (!P || !out) return fail(SD_ERR_INVALID, "null pointer");
    if (n <= 0) return fail(SD_ERR_INVALID, "empty input");
    if (d < 1 || d > 8) return fail(SD_ERR_UNSUPPORTED, "simplex containment covers d in [1,8], got %d", d);
    if (!targets && m != n) return fail(SD_ERR_INVALID, "targets=NULL requires m == n");
    if (exhaustive) {
        u64 c;
        if (!binom_u64_checked((u64)n_others, d + 1, &c) || (c >> 62))
            return fail(SD_ERR_OVERFLOW, "C(%lld,%d) subsets per target is not enumerable", (long long)n_others, d + 1);
    }
    return SD_OK;
}

int sd_pointcloud_simplex_counts(const double *P, int64_t n, int d, const int64_t *targets, int64_t m,
                                 double tol, int64_t *out, void *stream) {
    int rc = check_simplex(P, n, d, targets, m, out, true, n - 1);
    if (rc) return rc;
    if (m == 0) return SD_OK;
    return launch_pointcloud_simplex(P, n, d, targets, m, tol, -1, 0, (u64 *)out, (hipStream_t)stream);
}

size_t sd_simplex_sampled_workspace_bytes(int64_t n, int64_t T, int d, int64_t samples) {
    return simplex_sampled_workspace_bytes(n, T, d, samples);
}

int sd_pointcloud_simplex_sampled(const double *P, int64_t n, int d, const int64_t *targets, int64_t m,
                                  double tol, int64_t samples, uint64_t seed, int64_t *out, void *ws, size_t ws_bytes,
                                  void *stream) {
    int rc = check_simplex(P, n, d, targets, m, out, false, n - 1);
    if (rc) return rc;
    if (samples <= 0) return fail(SD_ERR_INVALID, "samples must be positive");
    if (n - 1 < d + 1) return fail(SD_ERR_INVALID, "need at least d+2 points");
    if (m == 0) return SD_OK;
    return launch_pointcloud_simplex(P, n, d, targets, m, tol, samples, seed, (u64 *)out, (hipStream_t)stream, ws, ws_bytes);
}

int sd_pointcloud_simplex_external_counts(const double *P, int64_t n, int d, const double *Q, int64_t m, double tol,
                                          int64_t *out, void *stream) {
    if (!Q) return fail(SD_ERR_INVALID, "null pointer");
    int rc = check_simplex(P, n, d, (const i64 *)1, m, out, true, n);
    if (rc) return rc;
    if (m < 0) return fail(SD_ERR_INVALID, "m < 0");
    if (m == 0) return SD_OK;
    return launch_pointcloud_simplex_external(P, n, d, Q, m, tol, (u64 *)out, (hipStream_t)stream);
}

int sd_pointcloud_simplex_subset_counts(const double *P, int64_t n, int d, const int32_t *members, int64_t nb, int bs,
                                        double tol, int64_t *out, void *stream) {
    if (!members) return fail(SD_ERR_INVALID, "null pointer");
    if (bs <= 0 || nb < 0) return fail(SD_ERR_INVALID, "bad shape");
    int rc = check_simplex(P, n, d, (const i64 *)1, nb, out, true, bs - 1);
    if (rc) return rc;
    if (nb == 0) return SD_OK;
    return launch_pointcloud_simplex_subsets(P, n, d, members, nb, bs, tol, (u64 *)out, (hipStream_t)stream);
}

int sd_multi_simplex_counts(const double *P, int64_t n, int64_t T, int d, const int64_t *targets, int64_t m,
                            int relax, double tol, int64_t *out, void *stream) {
    int rc = check_simplex(P, n, d, targets, m, out, true, n - 1);
    if (rc) return rc;
    if (T <= 0) return fail(SD_ERR_INVALID, "T <= 0");
    if (m == 0) return SD_OK;
    return launch_multi_simplex(P, n, T, d, targets, m, relax, tol, -1, 0, (u64 *)out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// K6
// ---------------------------------------------------------------------------
size_t sd_multi_band_workspace_bytes(int64_t n, int64_t T, int d) {
    if (n <= 0 || T <= 0 || d <= 0) return 0;
    return multi_band_workspace_bytes(n, T, d);
}

int sd_multi_band_counts(const double *P, int64_t n, int64_t T, int d, const int64_t *targets, int64_t m,
                         int64_t *out, void *ws, size_t ws_bytes, void *stream) {
    if (!P || !out) return fail(SD_ERR_INVALID, "null pointer");
    if (n <= 0 || T <= 0) return fail(SD_ERR_INVALID, "empty input");
    if (d < 1 || d > 8) return fail(SD_ERR_UNSUPPORTED, "componentwise band containment covers d in [1,8], got %d", d);
    if (!targets && m != n) return fail(SD_ERR_INVALID, "targets=NULL requires m == n");
    if (m < 0) return fail(SD_ERR_INVALID, "m < 0");
    int rc = check_count_range(T, n, 2);
    if (rc) return rc;
    if (m == 0) return SD_OK;
    return launch_multi_band(P, n, T, d, targets, m, (u64 *)out, ws, ws_bytes, (hipStream_t)stream);
}

int sd_multi_simplex_sampled(const double *P, int64_t n, int64_t T, int d, const int64_t *targets, int64_t m,
                             int relax, double tol, int64_t samples, uint64_t seed, int64_t *out, void *ws, size_t ws_bytes,
                             void *stream) {
    int rc = check_simplex(P, n, d, targets, m, out, false, n - 1);
    if (rc) return rc;
    if (T <= 0) return fail(SD_ERR_INVALID, "T <= 0");
    if (samples <= 0) return fail(SD_ERR_INVALID, "samples must be positive");
    if (n - 1 < d + 1) return fail(SD_ERR_INVALID, "need at least d+2 curves");
    if (m == 0) return SD_OK;
    return launch_multi_simplex(P, n, T, d, targets, m, relax, tol, samples, seed, (u64 *)out, (hipStream_t)stream, ws, ws_bytes);
}

}  // extern "C"
