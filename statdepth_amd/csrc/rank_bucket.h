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

}  // namespace sd
