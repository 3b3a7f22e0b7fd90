/*
 * oracle.c -- CPU restatement of statdepth's band-depth / containment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Nothing under statdepth_amd/ (the product) imports, links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked in
 * tests/test_oracle_golden.py against golden vectors produced by running the
 * reference itself (tests/golden/make_golden.py imports /root/reference in the
 * build container; fixtures = inputs + the reference's outputs).
 *
 * Each function cites the reference lines it restates (paths relative to the
 * reference checkout, statdepth/depth/calculations/...).
 *
 * Layout convention: a univariate data set is addressed as x(t,i) = X[t*st + i*sn]
 * (t = timepoint/row of the DataFrame, i = curve/column), so both pandas
 * layouts (C-contiguous T x n: st=n,sn=1; F-contiguous: st=1,sn=T) are read in
 * place.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;
typedef uint64_t u64;

#define XAT(t, i) X[(t) * st + (i) * sn]

/* exact C(a,k) in u64 by the multiplicative recurrence C(a,j)=C(a,j-1)*(a-j+1)/j
 * (each step exact).  Caller guarantees k*C(a,k) < 2^64. */
static u64 binom_u64(u64 a, int k) {
    if (k < 0 || (u64)k > a) return 0;
    u64 c = 1;
    for (int j = 1; j <= k; ++j) c = c * (a - (u64)j + 1) / (u64)j;
    return c;
}

void oracle_set_num_threads(int k) {
#ifdef _OPENMP
    if (k > 0) omp_set_num_threads(k);
#else
    (void)k;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* --------------------------------------------------------------------------
 * A1  _r2_containment(data, curve, relax)           _containment.py:45-80
 *   mins = data.min(axis=1); maxs = data.max(axis=1)   (:68-69, pandas skipna)
 *   c = #{t : mins[t] <= curve[t] <= maxs[t]}          (:75-77, inclusive)
 * Returns c (the caller applies c/T or c//T, :80).
 * `band` lists the j column indices forming the band.
 * NaN semantics follow pandas: min/max skip NaN; an all-NaN row gives NaN
 * bounds, and any comparison with NaN is false.
 * -------------------------------------------------------------------------- */
static long r2_containment_count(const double *X, long T, long st, long sn,
                                 const long *band, int j, long target) {
    long c = 0;
    for (long t = 0; t < T; ++t) {
        double mn = NAN, mx = NAN;
        for (int k = 0; k < j; ++k) {
            double v = XAT(t, band[k]);
            if (isnan(v)) continue;                 /* skipna */
            if (isnan(mn) || v < mn) mn = v;
            if (isnan(mx) || v > mx) mx = v;
        }
        double x = XAT(t, target);
        if (mn <= x && x <= mx) ++c;                /* false whenever a NaN is involved */
    }
    return c;
}

/* --------------------------------------------------------------------------
 * A2 (literal)  _univariate_band_depth              _functional.py:198-255
 *   for j in 2..J: for every j-subset of the OTHER n-1 curves (:235,243-246)
 *       S_nj += containment(subset, curve, relax)   (:251)
 * Literal enumeration, exponential in J: small cases only.  Output per target
 * and per j: relax -> sum over subsets of c (so S_nj = out/T); strict -> number
 * of subsets with c == T (c//T, _containment.py:80).
 * out is m x (J-1), row-major.
 * -------------------------------------------------------------------------- */
int oracle_band_enum(const double *X, long T, long n, long st, long sn,
                     const long *targets, long m, int J, int relax, i64 *out) {
    if (J < 2 || J > 8) return -1;
    for (long q = 0; q < m; ++q) {
        long tg = targets[q];
        long others[n > 1 ? n - 1 : 1];
        long no = 0;
        for (long i = 0; i < n; ++i)
            if (i != tg) others[no++] = i;
        for (int j = 2; j <= J; ++j) {
            i64 acc = 0;
            if (no >= j) {
                long idx[8], band[8];
                for (int k = 0; k < j; ++k) idx[k] = k;
                for (;;) {
                    for (int k = 0; k < j; ++k) band[k] = others[idx[k]];
                    long c = r2_containment_count(X, T, st, sn, band, j, tg);
                    acc += relax ? c : (c / T);
                    int k = j - 1;
                    while (k >= 0 && idx[k] == no - j + k) --k;
                    if (k < 0) break;
                    ++idx[k];
                    for (int l = k + 1; l < j; ++l) idx[l] = idx[l - 1] + 1;
                }
            }
            out[q * (J - 1) + (j - 2)] = acc;
        }
    }
    return 0;
}

/* --------------------------------------------------------------------------
 * 'r2_enum' (literal)  _r2_enum_containment          _containment.py:83-103
 * The reference declares this containment -- every component of a vector valued
 * function treated as a real valued function, "if all the components are
 * contained ... the function is contained" -- and leaves the body unimplemented
 * (raise NotImplementedError).  Restated as the docstring defines it, inside the
 * subset loop of _univariate_band_depth (_functional.py:238-253): for every
 * j-subset of the OTHER curves, c = #{t: for every feature f the band
 * [min, max] (skipna, inclusive) of the subset's f-th components contains the
 * target's}.  out[q][j-2] = sum over subsets of c (relax) or of [c == T] (strict).
 * P is n x T x d row-major.  Literal enumeration: small cases only.
 * -------------------------------------------------------------------------- */
int oracle_multi_band_enum(const double *P, long n, long T, int d, const long *targets, long m, int J, int relax,
                           i64 *out) {
    if (J < 2 || J > 8) return -1;
    for (long q = 0; q < m; ++q) {
        long tg = targets[q];
        long others[n > 1 ? n - 1 : 1];
        long no = 0;
        for (long i = 0; i < n; ++i)
            if (i != tg) others[no++] = i;
        for (int j = 2; j <= J; ++j) {
            i64 acc = 0;
            if (no >= j) {
                long idx[8];
                for (int k = 0; k < j; ++k) idx[k] = k;
                for (;;) {
                    long c = 0;
                    for (long t = 0; t < T; ++t) {
                        int all = 1;
                        for (int f = 0; f < d && all; ++f) {
                            double mn = NAN, mx = NAN;
                            for (int k = 0; k < j; ++k) {
                                double v = P[(others[idx[k]] * T + t) * d + f];
                                if (isnan(v)) continue;
                                if (isnan(mn) || v < mn) mn = v;
                                if (isnan(mx) || v > mx) mx = v;
                            }
                            double x = P[(tg * T + t) * d + f];
                            all = (mn <= x && x <= mx);
                        }
                        c += all;
                    }
                    acc += relax ? c : (c / T);
                    int k = j - 1;
                    while (k >= 0 && idx[k] == no - j + k) --k;
                    if (k < 0) break;
                    ++idx[k];
                    for (int l = k + 1; l < j; ++l) idx[l] = idx[l - 1] + 1;
                }
            }
            out[q * (J - 1) + (j - 2)] = acc;
        }
    }
    return 0;
}

/* --------------------------------------------------------------------------
 * A2 (closed form, relax=True)    SURVEY.md 8(a) A2
 * Per target i and timepoint t, over the n-1 OTHER curves:
 *   A = #strictly above, B = #strictly below, E = #equal, N = #NaN.
 * v = A+B+E valid others.  A j-subset with k>=1 valid members contains x iff not
 * all valid members are above and not all are below (min<=x<=max over the valid
 * members, _containment.py:68-77 with skipna):
 *   contained_j(t) = sum_{k=1..j} C(N, j-k) * [C(v,k) - C(A,k) - C(B,k)]
 * (no NaN: C(n-1,j) - C(A,j) - C(B,j)).  x NaN -> 0.
 * out[q][j-2] = sum_t contained_j(t)   (so S_nj = out / T, depth_j = S_nj / C(n,j)).
 * O(m*n*T).  This is also the function timed as bench.py's cpu_baseline.
 * -------------------------------------------------------------------------- */
int oracle_mbd_counts(const double *X, long T, long n, long st, long sn,
                      const long *targets, long m, int J, i64 *out) {
    if (J < 2 || J > 16) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (long q = 0; q < m; ++q) {
        long tg = targets[q];
        u64 acc[16];
        memset(acc, 0, sizeof acc);
        for (long t = 0; t < T; ++t) {
            double x = XAT(t, tg);
            if (isnan(x)) continue;
            u64 A = 0, B = 0, N = 0;
            if (sn == 1) {
                const double *row = X + t * st;
                for (long i = 0; i < n; ++i) {
                    double v = row[i];
                    A += (v > x);
                    B += (v < x);
                    N += (v != v);
                }
            } else {
                for (long i = 0; i < n; ++i) {
                    double v = XAT(t, i);
                    A += (v > x);
                    B += (v < x);
                    N += (v != v);
                }
            }
            /* the target itself is neither above, below nor NaN; it is "equal" */
            u64 v = (u64)(n - 1) - N;
            for (int j = 2; j <= J; ++j) {
                u64 s = 0;
                for (int k = 1; k <= j; ++k) {
                    u64 w = binom_u64(N, j - k);
                    if (!w) continue;
                    s += w * (binom_u64(v, k) - binom_u64(A, k) - binom_u64(B, k));
                }
                acc[j - 2] += s;
            }
        }
        for (int j = 2; j <= J; ++j) out[q * (J - 1) + (j - 2)] = (i64)acc[j - 2];
    }
    return 0;
}

/* The same totals by the RANK formulation the GPU's default path uses: per timepoint one sort of
 * the row, B = first position of the value, A = valid - position past its tie run; O(n T log n) instead of O(n^2 T).
 * Same integers as oracle_mbd_counts (checked against it and the fixtures in tests/test_oracle_golden.py); bench.py
 * times it as the like-for-like CPU baseline of the rank kernels.  Threads split the timepoints and add their
 * per-curve totals at the end. */
typedef struct { double v; long i; } oracle_kv;
static int oracle_kv_cmp(const void *a, const void *b) {
    double x = ((const oracle_kv *)a)->v, y = ((const oracle_kv *)b)->v;
    return (x > y) - (x < y);
}
int oracle_mbd_counts_ranksort(const double *X, long T, long n, long st, long sn, int J, i64 *out) {
    if (J < 2 || J > 16) return -1;
    memset(out, 0, sizeof(i64) * (size_t)n * (size_t)(J - 1));
    int fail = 0;
#pragma omp parallel
    {
        oracle_kv *kv = (oracle_kv *)malloc(sizeof(oracle_kv) * (size_t)n);
        u64 *acc = (u64 *)calloc((size_t)n * (size_t)(J - 1), sizeof(u64));
        if (!kv || !acc) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (long t = 0; t < T; ++t) {
                long nv = 0;
                for (long i = 0; i < n; ++i) {
                    double v = XAT(t, i);
                    if (v == v) { kv[nv].v = v; kv[nv].i = i; ++nv; }
                }
                qsort(kv, (size_t)nv, sizeof(oracle_kv), oracle_kv_cmp);
                u64 N = (u64)(n - nv);                       /* NaN curves: others of every valid target */
                for (long p = 0; p < nv;) {
                    long e = p + 1;
                    while (e < nv && kv[e].v == kv[p].v) ++e;   /* tie run [p, e) */
                    u64 B = (u64)p, A = (u64)(nv - e), v = (u64)(nv - 1);
                    for (int j = 2; j <= J; ++j) {
                        u64 s = 0;
                        for (int k = 1; k <= j; ++k) {
                            u64 w = binom_u64(N, j - k);
                            if (!w) continue;
                            s += w * (binom_u64(v, k) - binom_u64(A, k) - binom_u64(B, k));
                        }
                        for (long r = p; r < e; ++r) acc[kv[r].i * (J - 1) + (j - 2)] += s;
                    }
                    p = e;
                }
            }
#pragma omp critical
            for (long i = 0; i < n * (J - 1); ++i) out[i] += (i64)acc[i];
        }
        free(kv);
        free(acc);
    }
    return fail ? -2 : 0;
}

/* Per-(target,timepoint) above/below counts (the integer quantity K1 produces):
 * AB[q][t][0]=A, [1]=B, over all n curves != target position (NaN counted in
 * neither).  Used by tests to check kernels at the finest granularity. */
int oracle_above_below(const double *X, long T, long n, long st, long sn,
                       const long *targets, long m, i64 *AB) {
#pragma omp parallel for schedule(dynamic, 1)
    for (long q = 0; q < m; ++q) {
        long tg = targets[q];
        for (long t = 0; t < T; ++t) {
            double x = XAT(t, tg);
            i64 A = 0, B = 0;
            for (long i = 0; i < n; ++i) {
                double v = XAT(t, i);
                A += (v > x);
                B += (v < x);
            }
            AB[(q * T + t) * 2 + 0] = A;
            AB[(q * T + t) * 2 + 1] = B;
        }
    }
    return 0;
}

/* --------------------------------------------------------------------------
 * A2 (relax=False, J=2)   _functional.py:238-253 with _containment.py:80 (c//T)
 * out[q] = #{unordered pairs (a,b) of others : for every t,
 *            min(a,b) <= x <= max(a,b) with pandas skipna semantics}.
 * Per other curve a, per t, a state: U (a>x), D (a<x), E (a==x), N (a NaN).
 * The pair fails at t iff both U, both D, both N, or one is N and the other is
 * not E.  A NaN in the target fails every pair.  Bit masks over t, O(m*n^2*T/64).
 * -------------------------------------------------------------------------- */
int oracle_bd_strict_counts(const double *X, long T, long n, long st, long sn,
                            const long *targets, long m, i64 *out) {
    long W = (T + 63) / 64;
#pragma omp parallel
    {
        u64 *U = (u64 *)malloc((size_t)n * W * 8);
        u64 *D = (u64 *)malloc((size_t)n * W * 8);
        u64 *Nn = (u64 *)malloc((size_t)n * W * 8);
#pragma omp for schedule(dynamic, 1)
        for (long q = 0; q < m; ++q) {
            long tg = targets[q];
            int xnan = 0;
            memset(U, 0, (size_t)n * W * 8);
            memset(D, 0, (size_t)n * W * 8);
            memset(Nn, 0, (size_t)n * W * 8);
            for (long t = 0; t < T; ++t) {
                double x = XAT(t, tg);
                if (isnan(x)) { xnan = 1; break; }
                for (long i = 0; i < n; ++i) {
                    double v = XAT(t, i);
                    u64 bit = (u64)1 << (t & 63);
                    if (v > x) U[i * W + (t >> 6)] |= bit;
                    else if (v < x) D[i * W + (t >> 6)] |= bit;
                    else if (v != v) Nn[i * W + (t >> 6)] |= bit;
                }
            }
            i64 good = 0;
            if (!xnan) {
                for (long a = 0; a < n; ++a) {
                    if (a == tg) continue;
                    for (long b = a + 1; b < n; ++b) {
                        if (b == tg) continue;
                        u64 bad = 0;
                        for (long w = 0; w < W && !bad; ++w) {
                            u64 ua = U[a * W + w], da = D[a * W + w], na = Nn[a * W + w];
                            u64 ub = U[b * W + w], db = D[b * W + w], nb = Nn[b * W + w];
                            bad = (ua & ub) | (da & db) | (na & (ub | db | nb)) | (nb & (ua | da));
                        }
                        good += !bad;
                    }
                }
            }
            out[q] = good;
        }
        free(U); free(D); free(Nn);
    }
    return 0;
}

/* --------------------------------------------------------------------------
 * M2  _is_in_simplex(simplex_points, point)          _containment.py:138-176
 * Reference: feasibility of  sum_k l_k p_k = x, sum_k l_k = 1, l >= 0  through
 * scipy.optimize.linprog (third party: SciPy 1.15.3 / HiGHS here; the reference
 * pins scipy 1.5.2 in environment.yml:136).  Not vendored, so the published
 * problem statement is restated: x is in the convex hull of the k points, with a
 * feasibility tolerance, degenerate (affinely dependent) point sets allowed.
 *
 * Restatement: complete-pivoting elimination of M l = b, M = [P^T ; 1],
 * b = [x ; 1].  Full rank -> unique l, inside iff min l >= -tol.  Rank deficient
 * -> (Caratheodory) x is in the hull iff it is in the hull of some `rank`
 * affinely independent points among them: enumerate column subsets of size rank
 * in the pivoted row space.  Tolerances: rank threshold and consistency are
 * relative to the data scale; `tol` bounds l from below (l is scale free).
 * kpts = number of points (d+1 for a simplex), d = dimension.  kpts, d+1 <= 12.
 * -------------------------------------------------------------------------- */
#define SMAX 12

static int solve_subset(const double R[SMAX][SMAX + 1], int rank, const int *cols, double tol, double rank_eps) {
    /* solve the rank x rank system  R[0..rank)[cols] l = R[.][rhs]  by partial pivoting */
    double A[SMAX][SMAX + 1];
    int kp = SMAX; /* rhs column index in R */
    for (int r = 0; r < rank; ++r) {
        for (int c = 0; c < rank; ++c) A[r][c] = R[r][cols[c]];
        A[r][rank] = R[r][kp];
    }
    for (int c = 0; c < rank; ++c) {
        int p = c;
        double best = fabs(A[c][c]);
        for (int r = c + 1; r < rank; ++r)
            if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); p = r; }
        if (best <= rank_eps) return 0;            /* subset itself dependent */
        if (p != c)
            for (int k = 0; k <= rank; ++k) { double tmp = A[c][k]; A[c][k] = A[p][k]; A[p][k] = tmp; }
        for (int r = c + 1; r < rank; ++r) {
            double f = A[r][c] / A[c][c];
            for (int k = c; k <= rank; ++k) A[r][k] -= f * A[c][k];
        }
    }
    double l[SMAX];
    for (int c = rank - 1; c >= 0; --c) {
        double s = A[c][rank];
        for (int k = c + 1; k < rank; ++k) s -= A[c][k] * l[k];
        l[c] = s / A[c][c];
    }
    for (int c = 0; c < rank; ++c)
        if (!(l[c] >= -tol)) return 0;
    return 1;
}

int oracle_point_in_hull(const double *P, int kpts, int d, const double *x, double tol) {
    if (kpts > SMAX || d + 1 > SMAX || kpts < 1) return -1;
    int rows = d + 1;
    /* R = [M | b], rhs kept in column SMAX */
    double R[SMAX][SMAX + 1];
    double scale = 1.0;
    for (int r = 0; r < d; ++r) {
        for (int c = 0; c < kpts; ++c) {
            double v = P[c * d + r];
            if (v != v) return 0;
            R[r][c] = v;
            if (fabs(v) > scale) scale = fabs(v);
        }
        if (x[r] != x[r]) return 0;
        R[r][SMAX] = x[r];
        if (fabs(x[r]) > scale) scale = fabs(x[r]);
    }
    for (int c = 0; c < kpts; ++c) R[d][c] = 1.0;
    R[d][SMAX] = 1.0;
    if (isinf(scale)) return 0;
    double rank_eps = 1e-10 * scale;
    int colperm[SMAX];
    for (int c = 0; c < kpts; ++c) colperm[c] = c;
    int rank = 0;
    int lim = rows < kpts ? rows : kpts;
    for (; rank < lim; ++rank) {
        /* complete pivoting over the remaining block */
        int pr = -1, pc = -1;
        double best = rank_eps;
        for (int r = rank; r < rows; ++r)
            for (int c = rank; c < kpts; ++c)
                if (fabs(R[r][c]) > best) { best = fabs(R[r][c]); pr = r; pc = c; }
        if (pr < 0) break;
        if (pr != rank)
            for (int k = 0; k <= SMAX; ++k) { double tmp = R[rank][k]; R[rank][k] = R[pr][k]; R[pr][k] = tmp; }
        if (pc != rank) {
            for (int r = 0; r < rows; ++r) { double tmp = R[r][rank]; R[r][rank] = R[r][pc]; R[r][pc] = tmp; }
            int ti = colperm[rank]; colperm[rank] = colperm[pc]; colperm[pc] = ti;
        }
        for (int r = rank + 1; r < rows; ++r) {
            double f = R[r][rank] / R[rank][rank];
            if (f != 0.0) {
                for (int c = rank; c < kpts; ++c) R[r][c] -= f * R[rank][c];
                R[r][SMAX] -= f * R[rank][SMAX];
            }
        }
    }
    /* consistency: eliminated rows must have zero rhs (x in the affine hull) */
    for (int r = rank; r < rows; ++r)
        if (fabs(R[r][SMAX]) > 1e-7 * scale) return 0;
    if (rank == kpts) {
        int cols[SMAX];
        for (int c = 0; c < rank; ++c) cols[c] = c;
        return solve_subset(R, rank, cols, tol, rank_eps);
    }
    /* rank deficient: enumerate `rank`-subsets of the kpts columns (upper-trapezoidal rows 0..rank) */
    int cols[SMAX];
    for (int c = 0; c < rank; ++c) cols[c] = c;
    for (;;) {
        if (solve_subset(R, rank, cols, tol, rank_eps)) return 1;
        int k = rank - 1;
        while (k >= 0 && cols[k] == kpts - rank + k) --k;
        if (k < 0) break;
        ++cols[k];
        for (int l = k + 1; l < rank; ++l) cols[l] = cols[l - 1] + 1;
    }
    return 0;
}

/* next k-combination of {0..n-1} in lexicographic order; returns 0 when done */
static int next_comb(long *idx, int k, long n) {
    int i = k - 1;
    while (i >= 0 && idx[i] == n - k + i) --i;
    if (i < 0) return 0;
    ++idx[i];
    for (int l = i + 1; l < k; ++l) idx[l] = idx[l - 1] + 1;
    return 1;
}

/* --------------------------------------------------------------------------
 * P1  _pointwisedepth simplex branch                  _pointcloud.py:44-56
 *   for each point: S = #{(d+1)-subsets of the other n-1 points whose simplex
 *   contains it} (:50-54); depth = S / C(n, d+1), n INCLUDES the point (:38,56).
 * P is n x d row-major.  out[q] = S.
 * -------------------------------------------------------------------------- */
int oracle_pointcloud_simplex_counts(const double *P, long n, int d, const long *targets, long m,
                                     double tol, i64 *out) {
    int k = d + 1;
    if (k > SMAX) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (long q = 0; q < m; ++q) {
        long tg = targets[q];
        long no = n - 1;
        i64 S = 0;
        if (no >= k) {
            long idx[SMAX];
            double pts[SMAX * SMAX];
            for (int c = 0; c < k; ++c) idx[c] = c;
            do {
                for (int c = 0; c < k; ++c) {
                    long src = idx[c] < tg ? idx[c] : idx[c] + 1;   /* skip the target row */
                    memcpy(pts + c * d, P + src * d, sizeof(double) * d);
                }
                S += oracle_point_in_hull(pts, k, d, P + tg * d, tol) > 0;
            } while (next_comb(idx, k, no));
        }
        out[q] = S;
    }
    return 0;
}

/* --------------------------------------------------------------------------
 * M1  _simplex_depth + _simplex_containment   _functional.py:257-286, _containment.py:105-136
 *   curves: n arrays T x d (P is n x T x d row-major).  For target q and each
 *   (d+1)-subset of the n-1 others: c = #{t : x_q(t) in simplex of the subset at t}
 *   (:130-132); contribution c (relax, the caller divides by T) or c//T (strict) (:136).
 *   depth = sum / C(n-1, d+1)  (_functional.py:278,286: n there = number of OTHERS).
 * out[q] = sum over subsets of c (relax) or of [c == T] (strict).
 * -------------------------------------------------------------------------- */
int oracle_multi_simplex_counts(const double *P, long n, long T, int d, const long *targets, long m,
                                int relax, double tol, i64 *out) {
    int k = d + 1;
    if (k > SMAX) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (long q = 0; q < m; ++q) {
        long tg = targets[q];
        long no = n - 1;
        i64 S = 0;
        if (no >= k) {
            long idx[SMAX];
            double pts[SMAX * SMAX];
            for (int c = 0; c < k; ++c) idx[c] = c;
            do {
                long cnt = 0;
                for (long t = 0; t < T; ++t) {
                    for (int c = 0; c < k; ++c) {
                        long src = idx[c] < tg ? idx[c] : idx[c] + 1;
                        memcpy(pts + c * d, P + (src * T + t) * d, sizeof(double) * d);
                    }
                    cnt += oracle_point_in_hull(pts, k, d, P + (tg * T + t) * d, tol) > 0;
                }
                S += relax ? cnt : (cnt / T);
            } while (next_comb(idx, k, no));
        }
        out[q] = S;
    }
    return 0;
}

/* --------------------------------------------------------------------------
 * Seeded subset-sampling estimators (BASELINE.json configs 4 and 5; NOT in the reference, which can only
 * enumerate).  The sampler is the build's own definition, restated here so that the HIP kernels
 * (csrc/simplex.hip: sample_subset) can be checked draw for draw: splitmix64 counter generator, k distinct rows by
 * rejection, sorted ascending (see sample_rows below for the keys).
 * -------------------------------------------------------------------------- */
static u64 mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Round 4: shared sample s is ONE (d+1)-subset of all n rows for every target (generator keyed by (seed, s)), so that the
 * HIP kernels can share its factorisation between the targets; a target uses the FIRST `samples` shared subsets that do not
 * contain it (s = 0, 1, 2, ... in order): each is uniform over the subsets of the other rows. */
#define SX_SHARED_KEY 0xFFFFFFFFFFFFFFFFull
static void sample_rows(u64 seed, u64 keyv, u64 sample, int k, long n, long *idx) {
    u64 key = mix64(seed ^ mix64(keyv * 0xD1342543DE82EF95ull + sample));
    u64 ctr = 0;
    for (int p = 0; p < k;) {
        u64 r = mix64(key + ctr++);
        long c = (long)(((unsigned __int128)r * (unsigned __int128)(u64)n) >> 64);   /* floor(r * n / 2^64) */
        int dup = 0;
        for (int l = 0; l < p; ++l) dup |= (idx[l] == c);
        if (!dup) idx[p++] = c;
    }
    for (int a = 1; a < k; ++a) {
        long v = idx[a];
        int b = a - 1;
        while (b >= 0 && idx[b] > v) { idx[b + 1] = idx[b]; --b; }
        idx[b + 1] = v;
    }
}

/* the next shared subset from index *s on that does not contain row tg; *s is left behind it */
static void sample_next_valid(u64 seed, long tg, u64 *s, int k, long n, long *idx) {
    for (;;) {
        sample_rows(seed, SX_SHARED_KEY, (*s)++, k, n, idx);
        int member = 0;
        for (int l = 0; l < k; ++l) member |= idx[l] == tg;
        if (!member) return;
    }
}

/* P is n x T x d (T = 1: point cloud).  out[q] = sum over `samples` sampled (d+1)-subsets of the others of
 * c (relax) or [c == T] (strict), c = #timepoints where the target lies in the subset's simplex. */
int oracle_simplex_sampled(const double *P, long n, long T, int d, const long *targets, long m, int relax,
                           double tol, long samples, u64 seed, i64 *out) {
    int k = d + 1;
    if (k > SMAX || n - 1 < k) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (long q = 0; q < m; ++q) {
        long tg = targets[q];
        i64 S = 0;
        long idx[SMAX];
        double pts[SMAX * SMAX];
        u64 snext = 0;
        for (long s = 0; s < samples; ++s) {
            sample_next_valid(seed, tg, &snext, k, n, idx);
            long cnt = 0;
            for (long t = 0; t < T; ++t) {
                for (int c = 0; c < k; ++c) {
                    long src = idx[c];                              /* rows, never the target's */
                    memcpy(pts + c * d, P + (src * T + t) * d, sizeof(double) * d);
                }
                cnt += oracle_point_in_hull(pts, k, d, P + (tg * T + t) * d, tol) > 0;
            }
            S += (relax || T == 1) ? cnt : (cnt / T);
        }
        out[q] = S;
    }
    return 0;
}

/* --------------------------------------------------------------------------
 * P3  _L1_depth                                        _pointcloud.py:125-150
 *   e = sum_{y != x} (y - x)/||x - y||  (:145-146, in index order);
 *   depth = 1 - ||e|| / n  (:148,150), n includes x.  Coincident points: 0/0=NaN.
 * -------------------------------------------------------------------------- */
int oracle_l1_depth(const double *P, long n, int d, const long *targets, long m, double *out) {
#pragma omp parallel for schedule(static)
    for (long q = 0; q < m; ++q) {
        long tg = targets[q];
        double e[64];
        if (d > 64) { out[q] = NAN; continue; }
        for (int c = 0; c < d; ++c) e[c] = 0.0;
        for (long i = 0; i < n; ++i) {
            if (i == tg) continue;
            double s = 0.0;
            for (int c = 0; c < d; ++c) {
                double df = P[tg * d + c] - P[i * d + c];
                s += df * df;
            }
            double nr = sqrt(s);
            for (int c = 0; c < d; ++c) e[c] += (P[i * d + c] - P[tg * d + c]) / nr;
        }
        double s = 0.0;
        for (int c = 0; c < d; ++c) s += e[c] * e[c];
        out[q] = 1.0 - sqrt(s) / (double)n;
    }
    return 0;
}
