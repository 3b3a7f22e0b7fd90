// rank_bucket.h -- wave64 cross-lane helpers shared by the bucket-ranking kernels (mbd_rank_bucket.hip, mbd_rank_big.hip).
#pragma once
#include "sd_common.h"

namespace sd {

// ---- wave64 cross-lane helpers on DPP (no LDS traffic, no ds_bpermute latency chain) ----
// row_shr:1,2,4,8 scan inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row totals up.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ u32 rb_dpp(u32 old, u32 v) {
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWMASK, 0xF, false);
}

__device__ __forceinline__ u32 rb_row_incl_scan(u32 v) {       // inclusive scan inside each row of 16 lanes
    v += rb_dpp<0x111, 0xF>(0u, v);
    v += rb_dpp<0x112, 0xF>(0u, v);
    v += rb_dpp<0x114, 0xF>(0u, v);
    v += rb_dpp<0x118, 0xF>(0u, v);
    return v;
}

__device__ __forceinline__ u32 rb_wave_incl_scan(u32 v) {
    v = rb_row_incl_scan(v);
    v += rb_dpp<0x142, 0xA>(0u, v);
    v += rb_dpp<0x143, 0xC>(0u, v);
    return v;
}

__device__ __forceinline__ u32 rb_readlane(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }

__device__ __forceinline__ double rb_readlane_f64(double v, int l) {
    const u64 b = (u64)__double_as_longlong(v);
    const u64 r = (u64)rb_readlane((u32)b, l) | ((u64)rb_readlane((u32)(b >> 32), l) << 32);
    return __longlong_as_double((long long)r);
}

// v_min_f64 / v_max_f64 as single instructions (the builtins add a canonicalising v_max x,x,x per operand).  A quiet
// NaN operand yields the other operand, so NaNs drop out of a reduction; a signalling NaN would poison it, which
// only sends that row to the sort.
template <bool MAX>
__device__ __forceinline__ double rb_mm(double a, double b) {
    double r;
    if constexpr (MAX) asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int CTRL>
__device__ __forceinline__ double rb_dpp_f64(double v) {      // every lane has a source under row_ror: no old value
    const u64 b = (u64)__double_as_longlong(v);
    const u32 l = (u32)__builtin_amdgcn_mov_dpp((int)(u32)b, CTRL, 0xF, 0xF, false);
    const u32 h = (u32)__builtin_amdgcn_mov_dpp((int)(u32)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((u64)h << 32) | l));
}

// min (or max) over each row of 16 lanes, in every lane of the row (rotations by 1, 2, 4, 8)
template <bool MAX>
__device__ __forceinline__ double rb_row_allreduce(double v) {
    v = rb_mm<MAX>(v, rb_dpp_f64<0x121>(v));
    v = rb_mm<MAX>(v, rb_dpp_f64<0x122>(v));
    v = rb_mm<MAX>(v, rb_dpp_f64<0x124>(v));
    v = rb_mm<MAX>(v, rb_dpp_f64<0x128>(v));
    return v;
}

// ... over the whole wave, wave-uniform (the four row results meet through SGPRs)
template <bool MAX>
__device__ __forceinline__ double rb_wave_allreduce(double v) {
    v = rb_row_allreduce<MAX>(v);
    double r = rb_readlane_f64(v, 0);
    r = rb_mm<MAX>(r, rb_readlane_f64(v, 16));
    r = rb_mm<MAX>(r, rb_readlane_f64(v, 32));
    r = rb_mm<MAX>(r, rb_readlane_f64(v, 48));
    return r;
}


// max (or min) over each row of 16 lanes of a FLOAT, in every lane of the row: the rotation is a DPP modifier of the
// v_max_f32 / v_min_f32 itself (one instruction per step; s_nop: a DPP source written by the previous VALU needs two
// wait states).  For heuristics only -- the values are rounded.
template <bool MAX>
__device__ __forceinline__ float rb_row_allreduce_f32(float v) {
    if constexpr (MAX)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
            : "+v"(v));
    else
        asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
            : "+v"(v));
    return v;
}

// ---------------------------------------------------------------------------------------------------
// Three-piece bucket map of the fp64 bucket kernels (round 4; the 32-bit kernel's map, mbd_rank_bucket32.hip, on bucket
// indices).  A row whose range is much wider than its bulk -- heavy tails at every timepoint, outlying curves or entries --
// squeezed nearly every key into a few buckets of the linear map and sent the row to the sort (or, in the kernels that have
// none, into member loops of hundreds of keys).  Such a row keeps a linear core over a central bracket widened by RB3_BETA
// spans on either side (NB - 2 TB buckets) and gives the keys beyond it float-like codes of their distance d to the core's
// end, (bits(d + c) - bits(c)) >> SH with c = core width / 448: TB / 32 buckets per octave, the first octave continuing the
// core's slope, 32 octaves per tail, the last bucket takes what lies beyond.  Each piece is non-decreasing in x and the pieces'
// bucket ranges are disjoint and ordered, so the map is monotone for ANY data and any bracket: a bad bracket costs time, never
// a rank.  The bracket: every thread's own keys are a strided subset of the row; the minima and maxima of groups of 16 - 30
// such keys (1, 2, 4 or 16 neighbouring lanes; up to 48 keys below 4 keys per thread), 32 groups per workgroup, and of those the INNERMOST pair -- the largest
// minimum and the smallest maximum, swapped when they cross.  A few wild values cannot move it.  Heuristic values: float.
// ---------------------------------------------------------------------------------------------------
constexpr double RB3_BETA = 2.0;
constexpr double RB3_RATIO = 1.5 * (1.0 + 2.0 * RB3_BETA);    // clip only when the range exceeds this many bracket spans

template <bool MAX, int CTRL>
__device__ __forceinline__ float rb3_dpp_f32(float v) {       // v = min / max(v, v of the lane CTRL names) inside a quad
    if constexpr (MAX) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[%1,%2,%3,%4] row_mask:0xf bank_mask:0xf"
                           : "+v"(v) : "n"(CTRL & 3), "n"((CTRL >> 2) & 3), "n"((CTRL >> 4) & 3), "n"((CTRL >> 6) & 3));
    else asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[%1,%2,%3,%4] row_mask:0xf bank_mask:0xf"
             : "+v"(v) : "n"(CTRL & 3), "n"((CTRL >> 2) & 3), "n"((CTRL >> 4) & 3), "n"((CTRL >> 6) & 3));
    return v;
}

// mn / mx: this thread's minimum / maximum over its E keys (+inf / -inf when it holds none).  Returns, wave-uniform, the
// largest group minimum (.x) and the smallest group maximum (.y) of this wave's selected groups; a group without a key
// (or with values beyond float) is neutral.
template <int E, int NT>
__device__ __forceinline__ float2 rb3_wave_bracket(double mn, double mx) {
#ifdef RB3_OFF                                   // timing experiments: the linear map alone
    return make_float2(-__builtin_huge_valf(), __builtin_huge_valf());
#endif
    float a = (float)mn, b = (float)mx;
    constexpr int GL = E >= 16 ? 1 : E >= 8 ? 2 : E >= 4 ? 4 : 16;
    if constexpr (GL == 2 || GL == 4) {
        a = rb3_dpp_f32<false, 0xB1>(a);                              // quad_perm [1, 0, 3, 2]
        b = rb3_dpp_f32<true, 0xB1>(b);
    }
    if constexpr (GL == 4) {
        a = rb3_dpp_f32<false, 0x4E>(a);                              // quad_perm [2, 3, 0, 1]
        b = rb3_dpp_f32<true, 0x4E>(b);
    }
    if constexpr (GL == 16) {
        a = rb_row_allreduce_f32<false>(a);
        b = rb_row_allreduce_f32<true>(b);
    }
    const float FINF = __builtin_huge_valf();
    a = (a < FINF) ? a : -FINF;                                       // no key in the group (or NaN): neutral for the maximum
    b = (b > -FINF) ? b : FINF;
    auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    float2 r;
    r.x = __builtin_fmaxf(rl(a, 0), rl(a, 32));
    r.y = __builtin_fminf(rl(b, 0), rl(b, 32));
    if constexpr (NT <= 512) {                                        // 8 waves: four groups per wave
        r.x = __builtin_fmaxf(r.x, __builtin_fmaxf(rl(a, 16), rl(a, 48)));
        r.y = __builtin_fminf(r.y, __builtin_fminf(rl(b, 16), rl(b, 48)));
    }
    return r;
}

struct Rb3 {
    double lo, hi, scale, c;                                          // the core [lo, hi), NB - 2 TB buckets; c = width / 448
    u64 cb;                                                           // bits(c)
    bool clip;                                                        // block-uniform; false: the kernel's own linear map
};

// q: the wave brackets as written by the waves' last lanes, read by lane & (NW - 1); lo / hi: the row's exact extremes; n: keys
// of the row.  The range is clipped when it exceeds RB3_RATIO bracket spans AND so many that the bulk would crowd its buckets
// (about n * ratio / NB keys each): a short row spreads over NB buckets whatever its tails, and its Gaussian rows -- whose small
// sample makes a narrow bracket now and then -- stay on the linear map.
template <int NB>
__device__ __forceinline__ Rb3 rb3_make(double lo, double hi, float2 q, int n) {
    constexpr int TB = NB / 16, NCORE = NB - 2 * TB;
    const double INF = __builtin_huge_val();
#ifdef RB3_OFF
    { Rb3 m0; m0.lo = lo; m0.hi = hi; m0.scale = 0.0; m0.c = 0.0; m0.cb = 0; m0.clip = false; return m0; }
#endif
    const float l2 = rb_row_allreduce_f32<true>(q.x), h2 = rb_row_allreduce_f32<false>(q.y);
    const double a = (double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(l2)));
    const double b = (double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(h2)));
    const double s0 = a < b ? a : b, s1 = a < b ? b : a, si = s1 - s0;
    Rb3 m;
    m.lo = lo; m.hi = hi; m.scale = 0.0; m.c = 0.0; m.cb = 0; m.clip = false;
    const bool fin = (hi > lo) && (lo > -INF) && (hi < INF) && ((hi - lo) < INF);
    const double crowd = 4.0 * (double)NB / (double)n;
    if (fin && si > 0.0 && si < INF && (hi - lo) > (crowd > RB3_RATIO ? crowd : RB3_RATIO) * si) {   // block-uniform
        double lo2 = s0 - RB3_BETA * si, hi2 = s1 + RB3_BETA * si;
        lo2 = lo2 > lo ? lo2 : lo;
        hi2 = hi2 < hi ? hi2 : hi;
        const double wc = hi2 - lo2, sc = (double)NCORE / wc, c = wc * (1.0 / 448.0);
        if (wc > 0.0 && sc < INF && c > 0.0) {                        // (uniform values computed by the VALU: back into SGPRs)
            m.lo = rb_readlane_f64(lo2, 0); m.hi = rb_readlane_f64(hi2, 0); m.scale = rb_readlane_f64(sc, 0);
            m.c = rb_readlane_f64(c, 0); m.cb = (u64)__double_as_longlong(m.c); m.clip = true;
        }
    }
    return m;
}

// bucket of x under a clipped map (m.clip): [0, TB) low tail, [TB, NB - TB) core, [NB - TB, NB) high tail.  NaN -> TB.
template <int NB>
__device__ __forceinline__ u32 rb3_bucket(const Rb3 &m, double x) {
    constexpr int TB = NB / 16, NCORE = NB - 2 * TB;
    constexpr int LPO = (TB == 64 ? 1 : TB == 128 ? 2 : TB == 256 ? 3 : TB == 512 ? 4 : TB == 1024 ? 5 : 6);   // log2(TB / 32)
    static_assert(TB >= 64 && TB <= 2048 && (TB >> LPO) == 32, "32 octaves per tail");
    const double u = (x - m.lo) * m.scale;
    const double uc = u < (double)(NCORE - 1) ? u : (double)(NCORE - 1);
    u32 b;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(b) : "v"(uc));                  // saturating: negative and NaN -> 0
    b += (u32)TB;
    const bool low = u < 0.0, high = u >= (double)NCORE;              // fl(x - lo) < 0 <=> x < lo exactly
    if (low || high) {
        double d = low ? m.lo - x : x - m.hi;
        d = rb_mm<true>(d, 0.0);                                      // a key the rounding put just past the core's end: code 0
        const u64 tc = ((u64)__double_as_longlong(d + m.c) - m.cb) >> (52 - LPO);
        const u32 tcc = tc < (u64)(TB - 1) ? (u32)tc : (u32)(TB - 1);
        b = low ? (u32)(TB - 1) - tcc : (u32)(TB + NCORE) + tcc;
    }
    return b;
}

}  // namespace sd
