/* selftest.c -- drives every entry point of oracle.c on small random inputs; built with -fsanitize=address,undefined by
 * `make -C oracle sanitize` (test infrastructure: the checker checked for memory errors and undefined behaviour). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int64_t i64;
int oracle_band_enum(const double *, long, long, long, long, const long *, long, int, int, i64 *);
int oracle_multi_band_enum(const double *, long, long, int, const long *, long, int, int, i64 *);
int oracle_mbd_counts(const double *, long, long, long, long, const long *, long, int, i64 *);
int oracle_mbd_counts_ranksort(const double *, long, long, long, long, int, i64 *);
int oracle_above_below(const double *, long, long, long, long, const long *, long, i64 *);
int oracle_bd_strict_counts(const double *, long, long, long, long, const long *, long, i64 *);
int oracle_pointcloud_simplex_counts(const double *, long, int, const long *, long, double, i64 *);
int oracle_multi_simplex_counts(const double *, long, long, int, const long *, long, int, double, i64 *);
int oracle_simplex_sampled(const double *, long, long, int, const long *, long, int, double, long, uint64_t, i64 *);
int oracle_l1_depth(const double *, long, int, const long *, long, double *);

static double rnd(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (double)(*s >> 8) / 16777216.0 - 0.5; }

int main(void) {
    unsigned seed = 12345u;
    enum { T = 7, N = 11, D = 3 };
    double X[T * N], P[N * T * D], Q[N * D], l1[N];
    long tg[N];
    i64 out[N * 8], ab[N * T * 2];
    for (int i = 0; i < T * N; ++i) X[i] = floor(rnd(&seed) * 8.0) / 2.0;      /* ties */
    X[5] = NAN; X[17] = INFINITY; X[23] = -INFINITY;
    for (int i = 0; i < N * T * D; ++i) P[i] = floor(rnd(&seed) * 6.0);
    for (int i = 0; i < N * D; ++i) Q[i] = rnd(&seed);
    for (int i = 0; i < N; ++i) tg[i] = i;
    int rc = 0;
    for (int relax = 0; relax < 2; ++relax) rc |= oracle_band_enum(X, T, N, N, 1, tg, N, 3, relax, out);
    for (int relax = 0; relax < 2; ++relax) rc |= oracle_multi_band_enum(P, N, T, D, tg, N, 3, relax, out);
    rc |= oracle_mbd_counts(X, T, N, N, 1, tg, N, 4, out);
    rc |= oracle_mbd_counts(X, N, T, 1, N, tg, T, 2, out);                        /* the other layout */
    rc |= oracle_mbd_counts_ranksort(X, T, N, N, 1, 4, out);
    rc |= oracle_above_below(X, T, N, N, 1, tg, N, ab);
    rc |= oracle_bd_strict_counts(X, T, N, N, 1, tg, N, out);
    rc |= oracle_pointcloud_simplex_counts(Q, N, D, tg, N, 1e-7, out);
    rc |= oracle_multi_simplex_counts(P, N, T, D, tg, N, 1, 1e-7, out);
    rc |= oracle_multi_simplex_counts(P, N, T, D, tg, N, 0, 1e-7, out);
    rc |= oracle_simplex_sampled(Q, N, 0, D, tg, N, 1, 1e-7, 50, 7u, out);
    rc |= oracle_simplex_sampled(P, N, T, D, tg, N, 0, 1e-7, 20, 9u, out);
    rc |= oracle_l1_depth(Q, N, D, tg, N, l1);
    printf(rc ? "oracle selftest: an entry point returned an error\n" : "oracle selftest ok (ASan + UBSan clean)\n");
    return rc ? 1 : 0;
}
