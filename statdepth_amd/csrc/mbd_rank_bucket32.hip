// mbd_rank_bucket32.hip -- K1+K2 bucket ranking on 32-bit key images, TWO workgroups per CU (J = 2, 3072 < n <= 11264).
//
// Same integers as rank_bucket_kernel (mbd_rank_bucket.hip), the pairwise kernel and the reference's enumeration
// (_functional.py:246-251, _containment.py:75-77).  The fp64 kernel keeps one row per CU in LDS and its five
// barrier-separated phases leave the VALU idle while the LDS works and the other way round (SQ counters: VALU busy 57 %,
// LDS busy 37 %, together the kernel's whole time).  Here a row's keys live in LDS as 31-bit images q(x), non-decreasing in x
// whatever the data and whatever the map's constants (round 4: a THREE-PIECE map -- a linear core of 14 336 buckets over the
// row's range, clipped to a central bracket taken from a sample of 1 024 keys when the range is much wider than the bulk
// (heavy tails, outlying curves), and below and above the core 1 024 buckets each for a float-like code of the distance to
// the core's edge, bits(d + c) - bits(c) with c = core width / 448: linear with the core's slope next to the edge, halving
// per octave, 32 octaves; every piece is monotone and the pieces are ordered),
// so that a row of up to 11 264 curves + its 16 384-bucket histogram take 77 KiB and two 512-thread workgroups, half a
// row apart in phase, share a CU: one's compares run under the other's LDS traffic.  Order is decided by the images
// wherever they differ; keys whose images coincide (two values within range / 2^31, or equal values) are NOT ranked
// here: they are set aside with what the images do say (B0 = keys with a smaller image, E0 = keys with the same image)
// and settled exactly, in fp64, among themselves by the second launch (a group of equal images is complete in the list).
// Tie-heavy rows (quantised data: a bucket of 16 keys or more whose keys share an image) are ranked in closed form when
// every bucket of the row provably holds one value.
// Rows this kernel does not take -- a NaN, an infinity, a bucket of 64 keys or more, ties mixed with near-ties, and
// every row of a workgroup whose list of set-aside keys overflows -- are flagged and ranked by rank_bucket_kernel's
// SEL form in a second launch that returns at once when nothing was flagged (gate word = the call's epoch).
//
// Per-curve totals stay in registers (u32: the host checked that a workgroup's total fits); one partial block per
// workgroup goes to HBM and the second launch (rank_bucket_kernel's SEL form) sums the blocks into the totals.  HBM traffic: the matrix once + the partial blocks.
#include <atomic>
#include <stdio.h>

#include "sd_common.h"
#include "rank_bucket.h"

namespace sd {

#ifndef R32_PRIO_LDS
#define R32_PRIO_LDS 2                           // s_setprio in the load / histogram / prefix / scatter phases (issue on arrival)
#define R32_PRIO_VALU 0                          // ... and in the member pass, which is bound by VALU issue
#endif
#ifndef R32_MIN_N
#define R32_MIN_N 3072                           // the two-launch path is used above this n (measured: -10 % at 4 096, nothing at 3 072)
#endif
#ifndef R32_LNB
#define R32_LNB 14                               // 16 384 buckets: histogram 32 KiB
#endif
constexpr int R32_NT = 512, R32_NW = 8;
constexpr int R32_LCAP = 64;                    // set-aside keys per workgroup before it hands all its rows over
constexpr int R32_LIST_WORDS = 1 + 2 * R32_LCAP;   // a workgroup's list in the workspace: count, keys, (B0 | E0 << 16)
constexpr int R32_PAD = 12;                     // sentinel images behind the keys (two quads past the last partial quad)
constexpr u32 R32_TB = 1024;                    // buckets of each tail
constexpr int R32_TSH = 30;                     // tail code = (bits(d + c) - bits(c)) >> 30: 2^22 codes = 32 buckets per octave
constexpr double R32_CDIV = 1.0 / 448.0;        // c = core width / 448: the first octave continues the core's slope (32 * 448 = 14 336)
#ifndef R32_BETA
#define R32_BETA 2.0                             // the central bracket of the sample is widened by this many spans on either side
#endif
constexpr u32 R32_SENT = 0x7FFFFFFFu;           // sentinel image behind the keys: above every key image, below 2^31

template <int E, int LNB>
struct R32Cfg {
    static constexpr int NT = R32_NT, NW = R32_NW, NB = 1 << LNB;
    static constexpr int QW = NB / 2 / NT / 4;                  // 16-byte quads of histogram words per thread
    static_assert(QW >= 1 && NB / 2 == QW * 4 * NT, "whole quads of histogram words per thread");
    static constexpr int SH = 31 - LNB;
    static constexpr u32 C0 = R32_TB << SH, C1 = ((u32)NB - R32_TB) << SH;   // images of the core: [C0, C1)
    static constexpr u32 NANIMG = ((u32)NB + 2u) << SH;         // a NaN's image: its bucket is the dummy counter NB + 2
    static_assert(LNB == 14, "the first tail octave continues the core's slope for 14 336 core buckets");
    static constexpr size_t HDR = 8 * NW * 8 + NW * 4 + 32 + (size_t)R32_LCAP * 8;
    static_assert(HDR % 16 == 0, "the histogram starts on a 16-byte boundary");
    static __host__ __device__ constexpr int al4(int n) { return (n + 3) & ~3; }
    static __host__ __device__ constexpr int dummy_pos(int n) { return al4(n) + R32_PAD; }
    static __host__ __device__ constexpr size_t lds_bytes(int n) { return HDR + (size_t)(NB / 2 + 4) * 4 + (size_t)(dummy_pos(n) + 4) * 4; }
};

template <int E, int LNB>
__global__ __launch_bounds__(R32_NT, 4) void rank_bucket32_kernel(const double *__restrict__ Y, i64 n64, i64 row0, i64 rows,
                                                                 u32 *__restrict__ partial, unsigned char *__restrict__ rowflag,
                                                                 u32 *__restrict__ gate, u32 epoch, u64 *__restrict__ out_zero,
                                                                 u32 *__restrict__ listbuf) {
    using C = R32Cfg<E, LNB>;
    constexpr int NT = C::NT, NW = C::NW, NB = C::NB, QW = C::QW, SH = C::SH;
    constexpr u32 C0 = C::C0, C1 = C::C1, NANIMG = C::NANIMG;
    extern __shared__ double Sm[];
    const int n = (int)n64;
    double *red = Sm;                                                 // [2][NW][4] min / max of the row, central bracket of its sample
    u32 *wtot = reinterpret_cast<u32 *>(red + 8 * NW);                // [NW]
    u32 *misc = wtot + NW;                                            // [0]: set-aside keys so far, [1]: running sum of the ranks
    u32 *lkey = misc + 8, *lbe = lkey + R32_LCAP;                     // (row index << 14 | curve), B0 | E0 << 16
    u32 *H = reinterpret_cast<u32 *>(reinterpret_cast<char *>(Sm) + C::HDR);
    u32 *S = H + NB / 2 + 4;                                          // key images in bucket order + sentinels + dummy
    const unsigned short *H16 = reinterpret_cast<const unsigned short *>(H);
    const int t0 = threadIdx.x;
    const double INF = __builtin_huge_val();
    const double QNAN = __builtin_nan("");
    const int DUMMY = C::dummy_pos(n);
    const uint4 *SENT = reinterpret_cast<const uint4 *>(S + C::al4(n));   // a quad of sentinels, shared by every lane that needs one
    const u32 nm1 = (u32)n - 1u;
    const u32 ranksum = (u32)(((u64)n * (u64)(n - 1)) >> 1);          // sum of the ranks of a row without equal images
    int t = t0;
#ifdef R32_STAMPS
    const long long t_entry = (long long)__builtin_readcyclecounter();
    const long long rt_entry = (long long)__builtin_amdgcn_s_memrealtime();
#endif

    if (blockIdx.x == 0 && t == 0) gate[2] = 0;                       // the second launch's arrival counter
    if (out_zero)                                                     // the second launch adds every total to out
        for (i64 c = (i64)blockIdx.x * NT + t; c < n; c += (i64)gridDim.x * NT) out_zero[c] = 0;
    for (int p = n + t; p < DUMMY + 4; p += NT) S[p] = R32_SENT;
    {
        uint4 *Hq = reinterpret_cast<uint4 *>(H);
#pragma unroll
        for (int i = 0; i < QW; ++i) Hq[i * NT + t] = make_uint4(0, 0, 0, 0);
        if (t < 4) H[NB / 2 + t] = 0;
        if (t < 5) misc[t] = 0;
    }

    // curves of thread t: t, t + NT, ...; slots beyond n read as NaN (they count into the dummy word like any NaN)
    double x[E];
    auto load_row = [&](i64 r) {
        const double *rp = Y + (row0 + r) * n + t;
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = (e < E - 2 || t + e * NT < n) ? rp[e * NT] : QNAN;   // n > (E - 2) * NT
    };
    u32 acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0;
    // the sample of a row: one key per thread at evenly spaced curve indices -- thread t takes position j = 397 t mod 512 of the 512,
    // so that the 16 lanes of a DPP row hold keys from all over the curve index (curves ordered by level would otherwise make
    // every group one level).  The innermost of the 32 groups' minima and maxima bracket the bulk (about the quartiles).
    double smp;
    auto sample_issue = [&](i64 r) { smp = Y[(row0 + r) * n + (((((u32)t * 397u) & 511u) * (u32)n) >> 9)]; };
    auto row_range = [&](int parity) {
        double mn = INF, mx = -INF;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mn = rb_mm<false>(mn, x[e]);
            mx = rb_mm<true>(mx, x[e]);
        }
        mn = rb_wave_allreduce<false>(mn);
        mx = rb_wave_allreduce<true>(mx);
        // the sample's central bracket, in float (it only places the core: the map is monotone whatever it says): a NaN drops
        // out of v_min / v_max, a group without a value keeps +-inf and switches the bracket off
        const float sf = (float)smp;
        const float rmn = rb_row_allreduce_f32<false>(sf == sf ? sf : __builtin_huge_valf());
        const float rmx = rb_row_allreduce_f32<true>(sf == sf ? sf : -__builtin_huge_valf());
        auto rl = [](float v, int l) -> float { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        float imn = fmaxf(fmaxf(rl(rmn, 0), rl(rmn, 16)), fmaxf(rl(rmn, 32), rl(rmn, 48)));
        float imx = fminf(fminf(rl(rmx, 0), rl(rmx, 16)), fminf(rl(rmx, 32), rl(rmx, 48)));
        // Two sample values of a group of 16 equal (positions scattered over the row): values on a grid.  Such a row is ranked in
        // closed form when every bucket holds ONE value, which the tail buckets of a clipped map -- octaves wide -- would spoil:
        // an inverted bracket switches the clipping off for the row (its tails are no heavier for being rounded; if they are
        // heavy, the row is crowded and handed over as in round 3).
        {
            bool eqs = false;                                         // every pair of the group of 16 but those 8 lanes apart
#define R32_NB(k) eqs = eqs || (__int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sf), 0x120 + k, 0xF, 0xF, false)) == sf);   // row_ror:k
            R32_NB(1) R32_NB(2) R32_NB(3) R32_NB(4) R32_NB(5) R32_NB(6) R32_NB(7)
#undef R32_NB
            if (__ballot(eqs) != 0) { imn = __builtin_huge_valf(); imx = -__builtin_huge_valf(); }
        }
        double *rp = red + parity * 4 * NW;
        if ((t & 63) == 63) {
            double *wp = rp + 4 * (t >> 6);
            wp[0] = mn; wp[1] = mx;
            reinterpret_cast<float *>(wp + 2)[0] = imn;
            reinterpret_cast<float *>(wp + 2)[1] = imx;
        }
    };
#ifdef R32_STAMPS                                  // timing experiments: cycles per phase, one wave
    long long stamp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter();
#define R32_MARK(ph) { const long long now_ = (long long)__builtin_readcyclecounter(); stamp[ph] += now_ - tlast; tlast = now_; }
#else
#define R32_MARK(ph)
#endif
    int par = 0;
    u32 rowidx = 0, nbad = 0, expect = 0;                             // block-uniform
    bool handover = false;                                            // the list overflowed: every row of this workgroup is handed over
    bool stop = false;
    u32 kb[E];                                                        // image, then B = keys with a smaller image (the rank)
    i64 r = blockIdx.x;
    for (; r < rows; r += gridDim.x) {
        t = t0;
        asm volatile("" : "+v"(t));                                   // addresses are recomputed per row, not hoisted and spilled
        const int lane = t & 63;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        R32_MARK(0)
        // Tie-heavy / NaN-ridden data: when 8 of the first 16 workgroups have found their first row bad, everybody leaves what is left to
        // the second launch (gate[1] = epoch << 8 | count; block-uniform scalar load, a few rows stale at worst)
        if (rowidx && t == 0) misc[3] = __hip_atomic_load(gate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ONE load per workgroup
        __builtin_amdgcn_s_setprio(R32_PRIO_LDS);                     // latency-bound phases go first, the compares of the member
        sample_issue(r);
        load_row(r);                                                  // pass (the other workgroup's, half a row away) fill in
        row_range(par);
        R32_MARK(1)
        double *redp = red + par * 4 * NW;
        par ^= 1;
        __syncthreads();                                              // barrier 1 (histogram is zero, S is free)
        R32_MARK(2)
        if (rowidx) {                                                 // block-uniform: everybody reads the word thread 0 fetched
            const u32 w = misc[3];
            if ((w >> 8) == (epoch & 0xFFFFFFu) && (w & 0xFFu) >= 8u) { stop = true; break; }
        }
        double lo, hi;                                                // the row's minimum and maximum, then the core of the map
        double mscale, moff, mc;
        u64 mcb;
        u32 mc0, mcw;                                                 // the core's first image and its width in images
        bool go, bad;
        {
            const double2 p = reinterpret_cast<const double2 *>(redp)[2 * (lane & (NW - 1))];
            const float2 pb = reinterpret_cast<const float2 *>(redp + 4 * (lane & (NW - 1)) + 2)[0];
            lo = rb_readlane_f64(rb_row_allreduce<false>(p.x), 0);
            hi = rb_readlane_f64(rb_row_allreduce<true>(p.y), 0);
            const double imn = (double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rb_row_allreduce_f32<true>(pb.x))));
            const double imx = (double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rb_row_allreduce_f32<false>(pb.y))));
            // hi < lo: no value at this timepoint (every curve NaN), nothing is contained.  Else a row with an infinity, an
            // overflowing range or all values equal (one bucket) is handed over.
            const bool fin = (hi > lo) && (lo > -INF) && (hi < INF) && ((hi - lo) < INF);
            bad = !fin && (hi >= lo);
            // the core: the row's range, clipped to the sample's central bracket widened by R32_BETA spans on either side (a
            // light-tailed row keeps its whole range and never sees the tail code; heavy tails and outlying curves go to the tails)
            const double si = imx - imn;
            u32 c0 = 256u, c1 = 0x7FFFFE00u;                          // a light-tailed row: the core takes the whole image range
            if (fin && si > 0.0 && si < INF && (hi - lo) > (1.5 * (1.0 + 2.0 * R32_BETA)) * si) {   // block-uniform: the range is
                const double lo2 = imn - R32_BETA * si, hi2 = imx + R32_BETA * si;                  // much wider than the bulk
                lo = lo2 > lo ? lo2 : lo;
                hi = hi2 < hi ? hi2 : hi;
                c0 = C0;
                c1 = C1;
            }
            mc0 = c0;
            mcw = c1 - c0;
            const double wc = hi - lo;
            mscale = ((double)mcw - 64.0) / wc;                       // the largest key of an unclipped row stays inside the core
            moff = __builtin_fma(-lo, mscale, (double)c0);
            mc = wc * R32_CDIV;
            mcb = (u64)__double_as_longlong(mc);
            go = fin && (wc > 0.0) && (mscale < INF) && (mc > 0.0);
            bad = bad || (fin && !go);
        }
        // images of four keys: core keys cost fma + cvt + the core test; the tail code is computed only when some lane of the wave
        // has a key outside the core (or a NaN: cvt gives 0)
        auto convert4 = [&](const double (&xv)[4], u32 (&q)[4]) {
            bool anyout = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double u = __builtin_fma(xv[i], mscale, moff);
                u32 qi;
                asm("v_cvt_u32_f64 %0, %1" : "=v"(qi) : "v"(u));      // saturating: below -> 0, above -> 2^32 - 1, NaN -> 0
                q[i] = qi;
                anyout = anyout || (qi - mc0 >= mcw);
            }
            if (__ballot(anyout) != 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool outc = q[i] - mc0 >= mcw, low = q[i] < mc0;
                    double d = low ? lo - xv[i] : xv[i] - hi;
                    d = rb_mm<true>(d, 0.0);                          // a key the rounding put just outside: code 0
                    const u64 tc = ((u64)__double_as_longlong(d + mc) - mcb) >> R32_TSH;
                    // below the core: mc0 - 1 downwards; above: from the core's end upwards; clamped to what is left of the
                    // image range (a clipped core leaves 1 024 buckets on either side, an unclipped one a few images: only a
                    // key the rounding put just outside comes here then)
                    const u32 room = low ? mc0 - 1u : 0x7FFFFEFFu - (mc0 + mcw);
                    const u32 tcc = tc < (u64)room ? (u32)tc : room;
                    u32 qt = low ? (mc0 - 1u) - tcc : (mc0 + mcw) + tcc;
                    qt = (xv[i] == xv[i]) ? qt : NANIMG;
                    q[i] = outc ? qt : q[i];
                }
            }
        };
        u32 sl[(E + 3) / 4];
        if (go) {
            // ---- (1) image, bucket, slot: trunc(min(fl(fl(x - lo) * scale), TOP)), negative -> 0, is non-decreasing in x ----
#pragma unroll
            for (int e = 0; e < (E + 3) / 4; ++e) sl[e] = 0;
            // batches of 8 atomics in flight; their return values (the slots) are packed 4 to a register per batch
#pragma unroll
            for (int e0 = 0; e0 < E; e0 += 8) {
                u32 old[8];
#pragma unroll
                for (int h = 0; h < 8; h += 4) {
                    if (e0 + h < E) {
                        double xv[4];
                        u32 q[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) xv[i] = e0 + h + i < E ? x[e0 + h + i] : x[e0 + h];
                        convert4(xv, q);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (e0 + h + i < E) kb[e0 + h + i] = q[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = e0 + i;
                    if (e < E) {
                        const u32 b = kb[e] >> SH;                    // a NaN's image: the dummy counter NB + 2
                        old[i] = atomicAdd(&H[b >> 1], 1u << ((b & 1u) * 16u));
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = e0 + i;
                    if (e < E)   // a NaN key's slot is read from the wrong half: its row is not taken anyway
                        sl[e >> 2] |= ((old[i] >> (((kb[e] >> SH) & 1u) * 16u)) & 0xFFu) << (8 * (e & 3));
                }
                asm volatile("" : "+v"(sl[e0 >> 2]));                 // packed HERE: not (old, shift) pairs kept for the scatter
                if (e0 + 4 < E) asm volatile("" : "+v"(sl[(e0 >> 2) + 1]));
            }
            R32_MARK(3)
            __syncthreads();                                          // barrier 2
            R32_MARK(4)
            // ---- (2) exclusive prefix sum over the counters (conflict-free 16-byte accesses: lane <-> quad) ----
            uint4 *Hq = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);
            uint4 hq[QW];
            u32 runq[QW], inclq[QW], offq[QW], ov = 0, wsum = 0;
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                hq[i] = Hq[i * 64 + lane];
                const u32 s4 = hq[i].x + hq[i].y + hq[i].z + hq[i].w;
                ov |= hq[i].x | hq[i].y | hq[i].z | hq[i].w;          // bit k of a half set <=> some counter has it
                runq[i] = (s4 & 0xFFFFu) + (s4 >> 16);
                inclq[i] = rb_wave_incl_scan(runq[i]);
                offq[i] = wsum;
                wsum += rb_readlane(inclq[i], 63);
            }
            const bool wover = __ballot((ov & 0xFFC0FFC0u) != 0) != 0;   // some bucket holds 64 keys or more
            const bool wtry = __ballot((ov & 0xFFF0FFF0u) != 0) != 0;    // ... 16 or more: ties rather than density?
            if (lane == 63) wtot[wave] = wsum | (wover ? 0x80000000u : 0u) | (wtry ? 0x40000000u : 0u);
            R32_MARK(5)
            __syncthreads();                                          // barrier 3
            const u32 wt = wtot[lane & (NW - 1)];
            const bool crowded = __ballot((wt >> 31) != 0) != 0;
            const bool trypure = __ballot((wt & 0x40000000u) != 0) != 0;
            const u32 wscan = rb_row_incl_scan(wt & 0x3FFFFFFFu);
            const u32 woff = wave ? rb_readlane(wscan, wave - 1) : 0u;
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                u32 base = woff + offq[i] + inclq[i] - runq[i];
                uint4 o;
                o.x = base | ((base + (hq[i].x & 0xFFFFu)) << 16);
                base += (hq[i].x & 0xFFFFu) + (hq[i].x >> 16);
                o.y = base | ((base + (hq[i].y & 0xFFFFu)) << 16);
                base += (hq[i].y & 0xFFFFu) + (hq[i].y >> 16);
                o.z = base | ((base + (hq[i].z & 0xFFFFu)) << 16);
                base += (hq[i].z & 0xFFFFu) + (hq[i].z >> 16);
                o.w = base | ((base + (hq[i].w & 0xFFFFu)) << 16);
                base += (hq[i].w & 0xFFFFu) + (hq[i].w >> 16);
                Hq[i * 64 + lane] = o;
                if (i == QW - 1 && t == NT - 1) H[NB / 2] = base;     // base past the last bucket = number of non-NaN keys
            }
            __syncthreads();                                          // barrier 4
            R32_MARK(6)
            const u32 nv = H[NB / 2];
            // block-uniform.  A crowded row (a bucket of 64 keys or more) is scattered too: its slots wrap at 256 but stay inside
            // their buckets, which is all the closed form below asks of S; its member pass is never run.
            bool take = nv == (u32)n;
            u32 bc[E];                                                // base | count << 14
            if (take) {
                // ---- (3) scatter into bucket order ----
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const u32 b = kb[e] >> SH, slot = (sl[e >> 2] >> (8 * (e & 3))) & 0xFFu;
                    // two 2-byte reads, kept apart: merged into one 4-byte read they sit on an odd halfword for every
                    // odd b, and the LDS replays such a read for 64 cycles (SQ_LDS_UNALIGNED_STALL)
                    u32 b1 = b + 1u;
                    asm("" : "+v"(b1));                               // adjacency hidden from the load vectoriser
                    const u32 base = H16[b], end = H16[b1];
                    const bool isk = e < E - 2 || t + e * NT < n;
                    S[isk ? base + slot : (u32)DUMMY] = kb[e];
                    // a slot beyond n carries the NaN image: its "bucket" is the dummy counter, not a range of S -- one key at 0
                    bc[e] = isk ? base | ((end - base) << 14) : (1u << 14);
                }
            }
            R32_MARK(7)
            __syncthreads();                                          // barrier 5
            R32_MARK(8)
            bool pure = false;                                        // block-uniform: the row is ranked in closed form
            if (take && (trypure || crowded)) {
                // A bucket of 16 keys or more is tie-heavy (quantised) data more often than a dense cluster.  When a key
                // there shares its image with the first two keys of its bucket, the row is tried in closed form: if every
                // bucket of the row holds ONE value, the ranks are B = base, A = n - base - count and there is no member
                // pass at all.  One value per bucket is PROVEN on the doubles: equal images (the first key's, in S), then
                // equal high and low words -- every key of a bucket writes its words to the bucket's first two slots (one
                // key's land) and all compare with what landed.  A row that fails any of it is handed over.
                const u32 tag = rowidx + 1u;                          // misc words tagged by the row: nothing to reset
                bool vote = false, same = true;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const u32 base = bc[e] & 0x3FFFu, cnt = bc[e] >> 14;
                    if (e < E - 2 || t + e * NT < n) {
                        const u32 first = S[base];
                        same = same && (first == kb[e]);
                        if (cnt >= 16u) vote = vote || (first == kb[e] && S[base + 1] == kb[e]);
                    }
                }
                if (vote) misc[2] = tag;
                if (!same) misc[4] = tag;
                __syncthreads();
                // tie-heavy: closed form, or hand-over when a bucket is beyond the member pass (64 keys); a row with a bucket of 16
                // equal keys whose other buckets mix values (a clipped map's tail buckets are octaves wide) takes the member pass
                if ((misc[2] == tag && misc[4] != tag) || crowded) {
                    take = false;
                    bool ok = misc[2] == tag && misc[4] != tag;       // block-uniform
                    if (ok) {
                        // the doubles again (L2; their registers went to the images); -0 counts as +0.  A bucket of two keys
                        // or more takes the high word in its first slot and the low word in its second.  Four keys at a time
                        // (loaded again for the comparison): this cold path must not cost the row loop its registers.
                        const double *rp = Y + (row0 + r) * n + t;
#pragma unroll
                        for (int pass = 0; pass < 2; ++pass) {
                            bool eq = true;
#pragma unroll
                            for (int e0 = 0; e0 < E; e0 += 4) {
                                int eo = e0 * NT;                     // opaque: one address computation per group, none hoisted
                                asm volatile("" : "+v"(eo));
                                const double *rq = rp + eo;
                                u64 w[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const bool isk = e0 + i < E && (e0 + i < E - 2 || t + (e0 + i) * NT < n);
                                    w[i] = (u64)__double_as_longlong((isk ? rq[i * NT] : 0.0) + 0.0);
                                }
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    if (e0 + i < E && (e0 + i < E - 2 || t + (e0 + i) * NT < n) && (bc[e0 + i] >> 14) >= 2u) {
                                        u32 *sp = S + (bc[e0 + i] & 0x3FFFu);
                                        if (pass == 0) { sp[0] = (u32)(w[i] >> 32); sp[1] = (u32)w[i]; }
                                        else eq = eq && sp[0] == (u32)(w[i] >> 32) && sp[1] == (u32)w[i];
                                    }
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (pass == 1 && !eq) misc[4] = tag;
                            __syncthreads();
                        }
                        ok = misc[4] != tag;
                    }
                    if (ok) {
                        pure = true;
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const u32 B = bc[e] & 0x3FFFu, A = (u32)n - B - (bc[e] >> 14);
                            acc[e] += (e < E - 2 || t + e * NT < n) ? (nm1 * (nm1 - 1u) - A * (A - 1u) - B * (B - 1u)) >> 1 : 0u;
                        }
                    }
                }
            }
#if defined(R32_SECPRIO)                          // timing experiment: the workgroup dispatched second to a CU one level up in the member pass
            if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_setprio(R32_PRIO_VALU + 1);
            else
#endif
            __builtin_amdgcn_s_setprio(R32_PRIO_VALU);
            {                                                         // the histogram is dead until the next row's atomics
                uint4 *Hz = reinterpret_cast<uint4 *>(H) + wave * (64 * QW);
#pragma unroll
                for (int i = 0; i < QW; ++i) Hz[i * 64 + lane] = make_uint4(0, 0, 0, 0);
                if (t < 2) H[NB / 2 + t] = 0;
            }
            if (take) {
                // ---- (4) rank inside the bucket: two quads from the 16-byte boundary at or below the bucket's base; the
                //      keys in front of the base belong to earlier buckets (smaller images) and are taken off again, keys
                //      past the bucket's end have larger images, sentinels follow the last key.  Only `<` is counted:
                //      equal images show in the sum of the row's ranks (fold above) ----
                const uint4 *S4 = reinterpret_cast<const uint4 *>(S);
                uint4 y0, y1;
                auto window = [&](int e) {
                    const u32 base = bc[e] & 0x3FFFu, cnt = bc[e] >> 14;
                    const uint4 *p = S4 + (base >> 2);
                    y0 = p[0];
#ifdef R32_NOREDIRECT
                    y1 = p[1];
#else
                    y1 = *((cnt + (base & 3u) > 4u) ? p + 1 : SENT);
#endif
                };
                u32 sB = 0;
                window(0);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const u32 base = bc[e] & 0x3FFFu, cnt = bc[e] >> 14, off = base & 3u;
                    const u32 qe = kb[e];
                    // y < q <=> bit 31 of y - q (images below 2^31); v_alignbit shifts it into a bit list: two full-rate
                    // instructions per member and no compare -> carry chain through vcc (gfx950 pays wait states on those)
                    u32 lt = 0;
                    lt = __builtin_amdgcn_alignbit(lt, y0.x - qe, 31);
                    lt = __builtin_amdgcn_alignbit(lt, y0.y - qe, 31);
                    lt = __builtin_amdgcn_alignbit(lt, y0.z - qe, 31);
                    lt = __builtin_amdgcn_alignbit(lt, y0.w - qe, 31);
                    lt = __builtin_amdgcn_alignbit(lt, y1.x - qe, 31);
                    lt = __builtin_amdgcn_alignbit(lt, y1.y - qe, 31);
                    lt = __builtin_amdgcn_alignbit(lt, y1.z - qe, 31);
                    lt = __builtin_amdgcn_alignbit(lt, y1.w - qe, 31);
                    if (e + 1 < E) window(e + 1);
                    u32 less = (u32)__popc(lt);
                    if (cnt + off > 8u) {                             // the rest of a long bucket (a few lanes per visit)
                        const uint4 *p = S4 + (base >> 2);
#pragma unroll 1
                        for (u32 kk = 8; kk < cnt + off; kk += 4) {   // up to 63 + 3 positions: the bit list is counted per quad
                            const uint4 y = p[kk >> 2];
                            u32 l4 = 0;
                            l4 = __builtin_amdgcn_alignbit(l4, y.x - qe, 31);
                            l4 = __builtin_amdgcn_alignbit(l4, y.y - qe, 31);
                            l4 = __builtin_amdgcn_alignbit(l4, y.z - qe, 31);
                            l4 = __builtin_amdgcn_alignbit(l4, y.w - qe, 31);
                            less += (u32)__popc(l4);
                        }
                    }
                    const u32 B = base - off + less;                  // keys with a smaller image
                    kb[e] = B;
                    const bool isk = e < E - 2 || t + e * NT < n;
                    sB += isk ? B : 0u;
                    acc[e] += isk ? __umul24(B, nm1 - B) : 0u;        // contained pairs = A * B, taken back below if B is not the rank
                    __builtin_amdgcn_sched_barrier(0);                // one window ahead, not more: the registers are counted
                }
                sB = rb_wave_incl_scan(sB);
                if (lane == 63) atomicAdd(&misc[1], sB);
                __syncthreads();                                      // barrier 6: the row's rank sum is complete
                {
                    // ---- fold of the row.  Its ranks are exact iff no two images coincide, i.e. iff they are a
                    //      permutation of 0 .. n-1, i.e. iff they sum to n(n-1)/2 (equal images share the smaller rank) ----
                    const u32 have_sum = misc[1];
                    expect += ranksum;
                    if (have_sum != expect) {
                        // Keys with equal images have equal B: counted through the (empty) histogram as u16 counters indexed by
                        // B.  The tied ones take their product back and are set aside with B0 = B and the size of their group.
                        expect = have_sum;
                        const u32 l0 = misc[0];                       // the list before this row (no push since the last barrier)
#ifdef R32_STAMPS
                        stamp[0] += 1000000;
#endif
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            if (e < E - 2 || t + e * NT < n) atomicAdd(&H[kb[e] >> 1], 1u << ((kb[e] & 1u) * 16u));
                        __syncthreads();
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            if (e < E - 2 || t + e * NT < n) {
                                const u32 B = kb[e], c = H16[B];
                                if (c != 1u) {
                                    acc[e] -= __umul24(B, nm1 - B);
                                    const u32 idx = atomicAdd(&misc[0], 1u);
                                    if (idx < (u32)R32_LCAP) {
                                        lkey[idx] = (rowidx << 14) | (u32)(t + e * NT);
                                        lbe[idx] = B | (c << 16);
                                    }
                                }
                            }
                        }
                        __syncthreads();
#ifndef R32_NO_TIEPROOF                           // (timing experiments: the list alone)
                        // (E <= 16, n <= 8 192: from E = 18 on this cold code costs the row loop two accumulators in scratch,
                        // + 5 % on config 2; there a bucket of 16 equal keys is common on such data and the closed form above runs)
                        if (E <= 16 && misc[0] > (u32)R32_LCAP) {     // block-uniform
                            // More tied keys than the list takes: values on a grid (rounded measurements) with a few equal keys
                            // per bucket -- too few for the closed form above to be tried.  Keys with equal images share B, the
                            // group of B occupies the sorted positions B .. B + c - 1, so its first two slots of S (dead: the
                            // member pass is over) take the words of its doubles as above: if every group holds ONE value, a tied
                            // key has B curves strictly below and n - B - c strictly above, and the list is not needed.
                            const u32 tag = (rowidx + 1u) | 0x40000000u;       // (not the closed form's tag above: it may have failed on this row)
                            const double *rp = Y + (row0 + r) * n + t;
#pragma unroll
                            for (int pass = 0; pass < 2; ++pass) {
                                bool eq = true;
#pragma unroll
                                for (int e0 = 0; e0 < E; e0 += 2) {
                                    int eo = e0 * NT;
                                    asm volatile("" : "+v"(eo));
                                    const double *rq = rp + eo;
                                    u64 w[2];
#pragma unroll
                                    for (int i = 0; i < 2; ++i) {
                                        const bool isk = e0 + i < E && (e0 + i < E - 2 || t + (e0 + i) * NT < n);
                                        w[i] = (u64)__double_as_longlong((isk ? rq[i * NT] : 0.0) + 0.0);
                                    }
#pragma unroll
                                    for (int i = 0; i < 2; ++i) {
                                        if (e0 + i < E && (e0 + i < E - 2 || t + (e0 + i) * NT < n) && H16[kb[e0 + i]] >= 2u) {
                                            u32 *sp = S + kb[e0 + i];
                                            if (pass == 0) { sp[0] = (u32)(w[i] >> 32); sp[1] = (u32)w[i]; }
                                            else eq = eq && sp[0] == (u32)(w[i] >> 32) && sp[1] == (u32)w[i];
                                        }
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                                if (pass == 1 && !eq) misc[4] = tag;
                                __syncthreads();
                            }
                            if (misc[4] != tag) {                     // block-uniform: proven
#pragma unroll
                                for (int e = 0; e < E; ++e) {
                                    if (e < E - 2 || t + e * NT < n) {
                                        const u32 B = kb[e], c = H16[B];
                                        if (c != 1u) {
                                            const u32 A = (u32)n - B - c;
                                            acc[e] += (nm1 * (nm1 - 1u) - A * (A - 1u) - B * (B - 1u)) >> 1;
                                        }
                                    }
                                }
                                if (t == 0) misc[0] = l0;             // this row's entries are dropped
                            }
                            __syncthreads();
                        }
#endif
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            if (e < E - 2 || t + e * NT < n) H[kb[e] >> 1] = 0;
                        __syncthreads();
                        if (misc[0] > (u32)R32_LCAP) handover = true;     // block-uniform
                    }
                }
            } else {
                bad = !pure;
            }
        }
        R32_MARK(9)
        if (t == 0) rowflag[r] = bad ? 1 : 0;
        nbad += bad ? 1u : 0u;
        ++rowidx;
        if (handover) break;
        if (bad && rowidx == 1 && t == 0 && blockIdx.x < 16) {        // my first row was bad: count me in (16 contenders at most)
            u32 old = gate[1], assumed;
            do {
                assumed = old;
                const u32 want = ((assumed >> 8) != (epoch & 0xFFFFFFu)) ? (((epoch & 0xFFFFFFu) << 8) | 1u)
                                 : ((assumed & 0xFFu) == 0xFFu ? assumed : assumed + 1u);
                old = atomicCAS(gate + 1, assumed, want);
            } while (old != assumed);
        }
        if (nbad >= 2 && 2 * nbad > rowidx) { stop = true; r += gridDim.x; break; }   // this data is not for this kernel: leave the rest
    }
    t = t0;
    __syncthreads();
    if (handover) {                                                   // every row of this workgroup, ranked or not
        for (i64 rr = blockIdx.x + (i64)t * gridDim.x; rr < rows; rr += (i64)NT * gridDim.x) rowflag[rr] = 1;
        nbad = 1;
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0;
    } else if (stop) {
        for (i64 rr = r + (i64)t * gridDim.x; rr < rows; rr += (i64)NT * gridDim.x) rowflag[rr] = 1;   // rows left behind
    }
    if (t == 0 && nbad) *gate = epoch;
    u32 *P = partial + (size_t)blockIdx.x * C::al4(n);               // blocks on 16-byte boundaries whatever n
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e < E - 2 || t + e * NT < n) P[t + e * NT] = acc[e];
    // ---- keys whose images coincide go to the second launch, which settles them among themselves in fp64 (their rows are
    //      NaN-free and finite): count, then (row index << 14 | curve), then (B0 | E0 << 16) per key ----
    {
        const u32 L = handover ? 0u : misc[0];
        u32 *lb = listbuf + (size_t)blockIdx.x * R32_LIST_WORDS;
        if (t == 0) lb[0] = L;
        if ((u32)t < L) { lb[1 + t] = lkey[t]; lb[1 + R32_LCAP + t] = lbe[t]; }
    }
#ifdef R32_STAMPS
    __syncthreads();
    if ((t0 & 63) == 0 && (t0 == 0 || t0 == NT - 64)) {                // the oldest and the youngest wave of the workgroup
        u32 *dbg = listbuf + (size_t)gridDim.x * R32_LIST_WORDS + (size_t)blockIdx.x * 32 + (t0 ? 16 : 0);
        dbg[0] = (u32)((long long)__builtin_readcyclecounter() - t_entry);
        dbg[1] = (u32)rt_entry;
        dbg[2] = (u32)__builtin_amdgcn_s_memrealtime();
        dbg[3] = misc[0] | (rowidx << 16);
        for (int i = 0; i < 10; ++i) dbg[4 + i] = (u32)stamp[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool rank_bucket32_supported(i64 n, i64 rows, int cus) {
    // u32 totals per workgroup, for this kernel's workgroups and for the fp64 kernel's when every row is handed over
    const u64 per = (u64)((rows + cus - 1) / cus) * (u64)n * (u64)n;
    return n > R32_MIN_N && n <= 11264 && rows >= cus && rows <= 4096 && per < ((u64)1 << 32) && xswitch("SD_RB_NO32") == 0;
}

size_t rank_bucket32_extra_bytes(i64 rows) { return align_up((size_t)rows + 64, 256); }

template <int E>
static int launch32_cfg(const double *Y, i64 n, i64 row0, i64 rows, u32 *partial, unsigned char *rowflag, u32 *gate, u32 epoch,
                        u64 *out_zero, u32 *listbuf, int G, hipStream_t s) {
    using C = R32Cfg<E, R32_LNB>;
    auto kf = rank_bucket32_kernel<E, R32_LNB>;
    const size_t lds = C::lds_bytes((int)n);
    if (lds > 81920) return fail(SD_ERR_UNSUPPORTED, "bucket32 kernel: %zu bytes of LDS for n=%lld", lds, (long long)n);
    SD_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kf, dim3(G), dim3(R32_NT), lds, s, Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// rows [row0, row0 + rows): u32 partial totals of every curve per workgroup (G blocks of n), flags of the rows left to the
// fp64 kernel in rowflag[rows], *gate = epoch when there is any
#ifdef R32_STAMPS
size_t rank_bucket32_list_bytes(int G) { return (size_t)G * R32_LIST_WORDS * 4 + (size_t)G * 128; }
#else
size_t rank_bucket32_list_bytes(int G) { return (size_t)G * R32_LIST_WORDS * 4; }
#endif

int launch_rank_bucket32(const double *Y, i64 n, i64 row0, i64 rows, u32 *partial, unsigned char *rowflag, u32 *gate, u32 epoch,
                         u64 *out_zero, u32 *listbuf, int G, hipStream_t s) {
    switch ((int)((n + 1023) / 1024)) {
        case 3: return launch32_cfg<6>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 4: return launch32_cfg<8>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 5: return launch32_cfg<10>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 6: return launch32_cfg<12>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 7: return launch32_cfg<14>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 8: return launch32_cfg<16>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 9: return launch32_cfg<18>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 10: return launch32_cfg<20>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 11: return launch32_cfg<22>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
    }
    return fail(SD_ERR_UNSUPPORTED, "bucket32 kernel covers 3072 < n <= 11264");
}

u32 rank_bucket32_epoch() {
    static std::atomic<u32> counter{0};
    u32 e = ++counter;
    if (e == 0) e = ++counter;
    return e;
}

}  // namespace sd
