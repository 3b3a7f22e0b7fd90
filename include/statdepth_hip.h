/*
 * statdepth_hip.h -- C ABI of the MI355X (gfx950) band-depth engine.
 *
 * This is the drop-in boundary for statdepth's calculation layer
 * (the files under statdepth/depth/calculations/ in the reference).  The reference has no
 * FFI of its own -- its hot path is Python loops -- so each entry point below
 * names the reference function whose inner loops it replaces; the host side
 * (statdepth_amd/, Python) keeps the reference's FunctionalDepth /
 * PointcloudDepth signatures and calls these through ctypes.  INTEGRATION.md
 * shows the stub a statdepth maintainer would add to bind them.
 *
 * Conventions
 *  - plain C types only; every data pointer is a DEVICE pointer (HBM) unless
 *    the name says host; `stream` is a hipStream_t passed as void* (NULL = the
 *    null stream).  Calls enqueue work and return; they do not synchronise.
 *  - a univariate data set of n curves observed at T timepoints is addressed
 *    as x(t,i) = X[t*st + i*sn] (strides in elements), so both pandas layouts
 *    are accepted in place: C-contiguous T x n ("time-major", st=n, sn=1) and
 *    F-contiguous ("curve-major", st=1, sn=T).  Kernels run time-major; a
 *    curve-major input is transposed on the device into the workspace first.
 *  - `targets` is an int64 device array of m curve / point indices (the
 *    reference's `to_compute`), or NULL for "all, in order" (then m must be n).
 *  - integer outputs are the tested contract (bit-exact vs the reference);
 *    the fp64 normalisers (/T, /C(n,j) ...) stay on the host.
 *  - every function returns SD_OK (0) or an error code; sd_last_error() gives
 *    the message for the calling thread.
 */
#ifndef STATDEPTH_HIP_H
#define STATDEPTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_ABI_VERSION 1

enum sd_status {
    SD_OK = 0,
    SD_ERR_INVALID = 1,     /* bad argument (shape, stride, J, null pointer ...) */
    SD_ERR_HIP = 2,         /* a HIP runtime call failed */
    SD_ERR_NO_DEVICE = 3,   /* no gfx950 device visible */
    SD_ERR_UNSUPPORTED = 4, /* valid request outside what the kernels cover */
    SD_ERR_OVERFLOW = 5,    /* an int64 total could overflow (T*C(n-1,J) >= 2^63) */
    SD_ERR_WORKSPACE = 6    /* workspace too small */
};

/* algorithm selector of sd_mbd_counts */
enum sd_mbd_algo {
    SD_MBD_AUTO = 0,
    SD_MBD_PAIRWISE = 1, /* O(m n T): stream every curve against every target, compare + count */
    SD_MBD_RANK = 2      /* O(n T log n): per-timepoint sort in LDS, ranks give the same integers */
};

/* ---- library / device ---------------------------------------------------- */
int sd_abi_version(void);
/* 0 for the product library.  1 for libstatdepth_hip_xcheck.so, the -DSD_CROSSCHECK build of the same sources that
 * also holds the retired kernel generations and honours the SD_* environment switches selecting them (tests only). */
int sd_is_crosscheck_build(void);
const char *sd_last_error(void);
int sd_device_count(void);
/* name (e.g. "gfx950...") and CU count of device `dev`; name buffer >= 64 bytes */
int sd_device_info(int dev, char *name, int name_len, int *cu_count, size_t *hbm_bytes);
int sd_set_device(int dev);

/* ---- memory / stream helpers for clients without their own allocator ------ */
int sd_malloc(void **dptr, size_t bytes);
int sd_free(void *dptr);
int sd_memcpy_h2d(void *dst, const void *src_host, size_t bytes, void *stream);
int sd_memcpy_d2h(void *dst_host, const void *src, size_t bytes, void *stream);
int sd_memset(void *dst, int value, size_t bytes, void *stream);
int sd_stream_synchronize(void *stream);

/* ---- K1+K2: modified band depth totals (relax=True) ------------------------
 * Replaces: the subset loop of _univariate_band_depth (_functional.py:238-253)
 * with _r2_containment(relax=True) (_containment.py:45-80) inside it, for all
 * targets of _functionaldepth's loop (_functional.py:74-75).
 *
 * out[q*(J-1) + (j-2)] = sum over t of #{j-subsets of the OTHER n-1 curves whose
 * band [min,max] (pandas skipna, inclusive ends) contains target q at t},
 * j = 2..J.  The host forms S_nj = out/T and depth = sum_j S_nj / C(n,j).
 * Computed from per-timepoint counts A (others strictly above), B (strictly
 * below), N (NaN others), v = n-1-N:
 *     sum_{k=1..j} C(N,j-k) * [C(v,k) - C(A,k) - C(B,k)]       (0 if x is NaN)
 * which equals the reference's enumeration exactly (tests/test_oracle_golden.py).
 * J in [2, 8]; SD_ERR_OVERFLOW if T*C(n-1,J) >= 2^63.
 * `ws`/`ws_bytes`: device scratch of at least sd_mbd_workspace_bytes(...).
 */
size_t sd_mbd_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m, int J, int algo);
int sd_mbd_counts(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                  const int64_t *targets, int64_t m, int J, int algo,
                  int64_t *out, void *ws, size_t ws_bytes, void *stream);

/* Same totals for a CONTIGUOUS block of targets [target_begin, target_begin + m) -- the form the
 * target-sharded multi-GPU path uses (rank r owns one block of curves of the gathered set).  Lets the
 * chunked rank kernel search only the chunks that hold targets. */
int sd_mbd_counts_range(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                        int64_t target_begin, int64_t m, int J, int algo,
                        int64_t *out, void *ws, size_t ws_bytes, void *stream);

/* The same totals as TWO-LIMB unsigned integers for problems whose totals do not fit int64 (sd_mbd_counts returns
 * SD_ERR_OVERFLOW there: J >= 4 at n = 10^5, or very long T): out[(q*(J-1) + (j-2))*2 + {0,1}] = (low, high) 64 bits
 * of sum_t contained_j.  Requires J * C(n-1, J) < 2^63 (one timepoint's count fits 64 bits); the timepoints are
 * processed in chunks whose totals fit int64 and added with carry.  Workspace: sd_mbd_wide_workspace_bytes. */
size_t sd_mbd_wide_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m, int J, int algo);
int sd_mbd_counts_wide(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                       const int64_t *targets, int64_t m, int J, int algo,
                       uint64_t *out, void *ws, size_t ws_bytes, void *stream);

/* Band totals of m EXTERNAL curves Q (T x m, time-major dense) with respect to the n curves of X (time-major
 * dense, st = n, sn = 1): every curve of X is an "other".  This is what the reference's homogeneity
 * coefficients do |G| times with a temporary column (homogeneity.py:101-112,125-128: append g to F, call
 * FunctionalDepth(to_compute=[g]), drop g) -- here one launch for all of G.
 * out[q*(J-1)+(j-2)] = sum_t #{j-subsets of X's curves whose band contains Q[:,q] at t};
 * depth of g within F u {g} = sum_j out/T / C(n+1, j) on the host (_functional.py:229,253). */
int sd_mbd_external_counts(const double *X, int64_t T, int64_t n, const double *Q, int64_t m, int J,
                           int64_t *out, void *ws, size_t ws_bytes, void *stream);
/* Workspace for sd_mbd_external_counts.  With at least this much the call ranks the targets through a per-row bucket
 * structure (O(n + m) per timepoint, n <= 16384, J <= 3); with T*4 + 256 bytes it still works (pairwise, O(n m)). */
size_t sd_mbd_external_workspace_bytes(int64_t T, int64_t n, int64_t m, int J);

/* The same for the reference's default relax=False, J = 2: out[q] = number of pairs of X's curves whose band contains
 * Q[:,q] at EVERY timepoint (`c // T`, _containment.py:80); depth of g within F u {g} = out / C(n+1, 2).  One launch for
 * all of G (complement matching, see sd_bd_strict_counts). */
size_t sd_bd_strict_external_workspace_bytes(int64_t T, int64_t n, int64_t m);
int sd_bd_strict_external_counts(const double *X, int64_t T, int64_t n, const double *Q, int64_t m, int64_t *out, void *ws,
                                 size_t ws_bytes, void *stream);

/* Band totals of one target inside an explicit subset of the curves, for nb (subset, target) pairs in one launch:
 * the K-block sampled estimator (_samplefunctionaldepth, _functional.py:170-182) evaluates
 * _univariate_band_depth on n*K small blocks.  X time-major dense (st = n, sn = 1).
 * members: int32[nb*bs] column indices, -1 = padding; target: int32[nb], each a member of its block.
 * out[k*(J-1)+(j-2)] = sum_t #{j-subsets of the block's OTHER members whose band contains the target at t}.
 * Host: depth = sum_j out/T / C(block size, j). */
int sd_mbd_subset_counts(const double *X, int64_t T, int64_t n, const int32_t *members, int64_t nb, int bs,
                         const int32_t *target, int J, int64_t *out, void *stream);

/* The same estimator with the reference's default relax=False (`c // T`, _containment.py:80), J = 2: the number of pairs
 * of the block's OTHER members whose band contains the target at EVERY timepoint, for nb (subset, target) pairs in one
 * launch (a workgroup per pair, the block's masks in LDS).  Arguments as sd_mbd_subset_counts; out: int64[nb].
 * Host: depth = out / C(block size, 2).  Blocks whose masks do not fit the LDS (sd_bd_strict_subset_supported == 0:
 * bs * (2 * ceil(T/32) + 2) * 4 bytes > ~158 KB) keep them in the workspace instead
 * (sd_bd_strict_subset_workspace_bytes; 0 when the LDS suffices, ws may then be NULL). */
size_t sd_bd_strict_subset_workspace_bytes(int64_t T, int64_t nb, int bs);
int sd_bd_strict_subset_counts(const double *X, int64_t T, int64_t n, const int32_t *members, int64_t nb, int bs,
                               const int32_t *target, int64_t *out, void *ws, size_t ws_bytes, void *stream);
int sd_bd_strict_subset_supported(int64_t T, int bs);

/* Finest-granularity form of K1 (tests, diagnostics): AB[(q*T + t)*2 + {0,1}] =
 * (#curves strictly above, #strictly below) target q at t, as uint32. */
int sd_above_below(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                   const int64_t *targets, int64_t m, uint32_t *AB,
                   void *ws, size_t ws_bytes, void *stream);

/* ---- K3: strict band depth (relax=False), J = 2 ------------------------------
 * Replaces: the same loop (_functional.py:246-251) with `containment // len(curve)`
 * (_containment.py:80): a pair counts only if its band contains the target at
 * EVERY timepoint.  out[q] = number of such unordered pairs of other curves.
 * depth = out / C(n,2) on the host.
 * How the pairs are counted (always the same integers):
 *   T <= 5 (point clouds as `FunctionalDepth([points.T])`: the L-infinity / box depth), any n: per target one pass over
 *     the curves into 3^T / 4^T state classes and a class transform -- O(n) per target;
 *   T = 6 ... 8 without NaN anywhere, any n: the same (3^T classes; "NaN anywhere" is a flag the call reads back from the
 *     device: for these T the call waits on `stream` once before it launches the counting;
 *     sd_bd_strict_nanfree_workspace_bytes is the workspace such data needs).  With NaN: as below;
 *   n <= 131 071: curves that are strictly above or below the target at every timepoint ("clean") pair up exactly when
 *     their above-masks are complements, so those pairs are counted by grouping masks; pairs with a curve that ties
 *     with the target or holds NaN are tested one by one (only the targets that have such curves);
 *   beyond: every pair is tested; calls that would need more than 2e14 pair tests are refused (SD_ERR_UNSUPPORTED).
 * NaN in the target: out[q] = 0; NaN in another curve: it joins both masks (pandas' skipna min / max).
 * The workspace holds up to 16 GiB of masks for a batch of targets when n is large (sd_bd_strict_workspace_bytes: the
 * RECOMMENDED size).  A caller short of memory may pass less, down to sd_bd_strict_min_workspace_bytes (one target per
 * batch): the launcher sizes its batches to what it is given -- same integers, more launches.
 */
size_t sd_bd_strict_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m);
size_t sd_bd_strict_min_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m, int J);
/* The size for a caller who knows that X holds no NaN (J = 2): a few KB for 6 ... 8 timepoints (the state classes need a flag,
 * none of the mask pipeline's buffers), sd_bd_strict_workspace_bytes otherwise.  Passing it for data WITH NaN is safe: the
 * call returns SD_ERR_WORKSPACE. */
size_t sd_bd_strict_nanfree_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m);
int sd_bd_strict_counts(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                        const int64_t *targets, int64_t m,
                        int64_t *out, void *ws, size_t ws_bytes, void *stream);

/* ---- K3b: strict band depth, general J (subset enumeration over bit masks) ----
 * out[q*(J-1)+(j-2)] = #{j-subsets of others containing target q at every t}.
 * J in [2, 4]; work grows as C(n-1,J) per target.
 */
size_t sd_bd_strict_j_workspace_bytes(int64_t T, int64_t n, int64_t st, int64_t sn, int64_t m, int J);
int sd_bd_strict_j_counts(const double *X, int64_t T, int64_t n, int64_t st, int64_t sn,
                          const int64_t *targets, int64_t m, int J,
                          int64_t *out, void *ws, size_t ws_bytes, void *stream);

/* ---- K5: L1 (spatial) depth of a point cloud ----------------------------------
 * Replaces: _L1_depth (_pointcloud.py:125-150).  P is n x d row-major (rows =
 * points).  out[q] = 1 - || sum_{y != x} (y-x)/||x-y|| || / n  (fp64; coincident
 * points give NaN like the reference's 0/0).
 */
int sd_l1_depth(const double *P, int64_t n, int d, const int64_t *targets, int64_t m,
                double *out, void *stream);

/* The same depth for m EXTERNAL points Q (m x d): every row of P is an "other" and the sample counts n + 1 points --
 * what the reference's point-cloud homogeneity does once per point of G with a temporary row
 * (homogeneity.py:172-175,183-186: append g to F, PointcloudDepth(to_compute=[g]), drop g); here one launch for all. */
int sd_l1_external_depth(const double *P, int64_t n, int d, const double *Q, int64_t m, double *out, void *stream);
/* ... and inside explicit BLOCKS of rows, nb (block, target) pairs in one launch: the K-block sampled estimator
 * (_samplepointwisedepth, _pointcloud.py:107-121) calls _pointwisedepth on len(to_compute) * (n // K) samples.
 * members: int32[nb*bs] row indices, -1 = padding; in every block the OTHER rows first and the target LAST.
 * out[k] = 1 - ||sum over the block's others|| / (block size). */
int sd_l1_subset_depth(const double *P, int64_t n, int d, const int32_t *members, int64_t nb, int bs, double *out,
                       void *stream);

/* ---- K4: simplex containment counts ---------------------------------------------
 * Replaces: _is_in_simplex (_containment.py:138-176, an LP feasibility test
 * through scipy.optimize.linprog) inside
 *   - _pointwisedepth's subset loop (_pointcloud.py:50-54): sd_pointcloud_simplex_counts,
 *     P n x d row-major, out[q] = #{(d+1)-subsets of the other n-1 points whose
 *     simplex contains point q}; depth = out / C(n,d+1) on the host;
 *   - _simplex_depth / _simplex_containment (_functional.py:281-285,
 *     _containment.py:130-136): sd_multi_simplex_counts, P n x T x d row-major
 *     (curve, timepoint, feature); per target and (d+1)-subset of the other
 *     curves c = #{t: x_q(t) in simplex}; out[q] = sum c (relax != 0) or
 *     sum [c == T] (relax == 0); depth = out / (T or 1) / C(n-1,d+1) on the host.
 * Containment = the closed simplex (degenerate point sets allowed) with
 * feasibility tolerance `tol` on the barycentric coordinates (1e-7 mirrors the
 * LP solver's default).  d in [1, 8].
 * subset enumeration is exhaustive: C(n-1,d+1) per target must be < 2^62.
 */
int sd_pointcloud_simplex_counts(const double *P, int64_t n, int d,
                                 const int64_t *targets, int64_t m, double tol,
                                 int64_t *out, void *stream);
int sd_multi_simplex_counts(const double *P, int64_t n, int64_t T, int d,
                            const int64_t *targets, int64_t m, int relax, double tol,
                            int64_t *out, void *stream);

/* External targets / explicit blocks, as for sd_l1_external_depth / sd_l1_subset_depth above:
 *   out[q] = #{(d+1)-subsets of ALL n rows of P whose simplex contains Q[q]}; depth = out / C(n+1, d+1) on the host
 *   (_pointcloud.py:38,56 with the temporary row counted: homogeneity.py:172-175,183-186);
 *   out[k] = #{(d+1)-subsets of block k's other rows whose simplex contains the block's target (its last row)};
 *   depth = out / C(block size, d+1) (_pointcloud.py:107-121 -> :50-56 on the sample). */
int sd_pointcloud_simplex_external_counts(const double *P, int64_t n, int d, const double *Q, int64_t m, double tol,
                                          int64_t *out, void *stream);
int sd_pointcloud_simplex_subset_counts(const double *P, int64_t n, int d, const int32_t *members, int64_t nb, int bs,
                                        double tol, int64_t *out, void *stream);

/* Seeded uniform subset-sampling estimators for sizes where exhaustive
 * enumeration is impossible (BASELINE.json configs 4 and 5; not in the
 * reference).  For each target, `samples` (d+1)-subsets of the
 * other items are drawn with a counter-based generator (keys below); out[q] = number of
 * containing simplices (pointcloud) or sum over samples of c / [c == T] (multi).
 */
/* Round 4: sample s is ONE (d+1)-subset of all n items for every target (generator keyed by (seed, s)); a target that is
 * itself a member gets that member replaced by an item drawn with the generator keyed by (seed, target, s) -- every target
 * sees `samples` uniform subsets of the other items, and the subset's elimination is shared by all targets (each target only
 * carries its right-hand side through the recorded row operations: the same barycentric coordinates bit for bit).
 * ws / ws_bytes: optional (NULL / 0 allowed) records of a batch of (timepoint, sample) pairs, sd_simplex_sampled_workspace_bytes;
 * with it the factorisation and the replay are two kernels (d >= 4: several times faster), without it one. */
size_t sd_simplex_sampled_workspace_bytes(int64_t n, int64_t T, int d, int64_t samples);
int sd_pointcloud_simplex_sampled(const double *P, int64_t n, int d,
                                  const int64_t *targets, int64_t m, double tol,
                                  int64_t samples, uint64_t seed, int64_t *out,
                                  void *ws, size_t ws_bytes, void *stream);
int sd_multi_simplex_sampled(const double *P, int64_t n, int64_t T, int d,
                             const int64_t *targets, int64_t m, int relax, double tol,
                             int64_t samples, uint64_t seed, int64_t *out,
                             void *ws, size_t ws_bytes, void *stream);

/* ---- K6: componentwise band containment of multivariate curves ('r2_enum') --------
 * Fills: _r2_enum_containment (_containment.py:83-103), which the reference declares -- "treat each component in the
 * vector valued function as a real valued function ... if all the components are contained ... the function is
 * contained" -- and leaves as `raise NotImplementedError`; built as the predicate of _univariate_band_depth's pair
 * loop (_functional.py:238-253), J = 2, relax=True.  P is n x T x d row-major (curve, timepoint, feature), NaN-free.
 *   out[q] = sum_t #{pairs {a,b} of the other curves: for every feature f,
 *                    min(a_f(t), b_f(t)) <= x_q,f(t) <= max(a_f(t), b_f(t))};   depth = out / T / C(n,2) on the host
 * (d = 1 gives sd_mbd_counts' totals).  The strict form (contained at every t) is sd_bd_strict_j_counts over the
 * T*d component series of each curve (st = 1, sn = T*d) and needs no entry point of its own.
 * n <= 65535 and n*d*2 + 8*3^d bytes of LDS (config 4: 5000 curves, d = 8: 132 KB). */
size_t sd_multi_band_workspace_bytes(int64_t n, int64_t T, int d);
int sd_multi_band_counts(const double *P, int64_t n, int64_t T, int d, const int64_t *targets, int64_t m,
                         int64_t *out, void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STATDEPTH_HIP_H */
