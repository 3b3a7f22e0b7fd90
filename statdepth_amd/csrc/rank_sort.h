// rank_sort.h -- in-LDS sort machinery shared by the rank kernels (mbd_rank2.hip, mbd_rank_big.hip):
// normalised bitonic network over fp64 keys held E per thread, register windows moved by transposes
// through a padded LDS image, positions >= n_act never touched.  See mbd_rank2.hip for the description.
#pragma once
#include "sd_common.h"

namespace sd {

template <int NT, int E>
struct R2Cfg {
    static constexpr int N = NT * E;
    static constexpr int LE = (E == 1) ? 0 : (E == 2) ? 1 : (E == 4) ? 2 : (E == 8) ? 3 : (E == 16) ? 4 : 5;
    static constexpr int LT = (NT == 64) ? 6 : (NT == 128) ? 7 : (NT == 256) ? 8 : (NT == 512) ? 9 : 10;
    static constexpr int LN = LE + LT;
    static constexpr int WB = 64 * E;                  // positions owned by one wave in wave-local layouts
    static constexpr int SLOTS = N + (N >> LE);
    static constexpr size_t LDS_BYTES = (size_t)SLOTS * 8;
};

template <int LE>
__device__ __forceinline__ int r2_phys(int p) { return p + (p >> LE); }

template <int B, int LE>
__device__ __forceinline__ int r2_base(int t) {
    int u = ((t >> B) << (B + LE)) | (t & ((1 << B) - 1));
    return u + (u >> LE);
}
template <int B, int LE>
__device__ __forceinline__ constexpr int r2_off(int r) { return (r << B) + ((r << B) >> LE); }

__device__ __forceinline__ void r2_cmpx(double &a, double &b) {
    // exactly two instructions: the builtin fmin/fmax add a canonicalising v_max_f64 x,x,x per operand
    // after every LDS load (keys are never NaN here, so no quieting is needed)
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo;
    b = hi;
}

__device__ __forceinline__ void r2_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NT, int E>
struct R2Sorter {
    using C = R2Cfg<NT, E>;
    static constexpr int LE = C::LE;
    static constexpr int LN = C::LN;

    static constexpr int wb(int S, int k) { return (S - (k + 1) * LE) > 0 ? (S - (k + 1) * LE) : 0; }

    // number of real registers of thread t in window B: positions ((t>>B) << (B+LE)) + (r << B) + (t & (2^B-1))
    // below n_act.  n_act is a multiple of 64*E, so the count is wave-uniform for B >= 6.
    template <int B>
    static __device__ __forceinline__ int cnt_of(int t, int n_act) {
        int c = (n_act - ((t >> B) << (B + LE)) - (t & ((1 << B) - 1)) + (1 << B) - 1) >> B;
        c = c < 0 ? 0 : (c > E ? E : c);
        return __builtin_amdgcn_readfirstlane(c);
    }

    // wave-local transpose (both windows <= 6); REV: lower-half registers read the reversed lower half
    template <int BF, int BT, bool REV>
    static __device__ __forceinline__ void transpose_local(double (&k)[E], double *S, int t) {
        double *Sw = S + r2_base<BF, LE>(t);
#pragma unroll
        for (int r = 0; r < E; ++r) Sw[r2_off<BF, LE>(r)] = k[r];
        r2_wave_sync();
        const double *Sr = S + r2_base<BT, LE>(t);
        if constexpr (REV) {
            const double *Sf = S + r2_base<BT, LE>(t ^ ((1 << BT) - 1));
#pragma unroll
            for (int r = 0; r < E; ++r)
                k[r] = (r < E / 2) ? Sf[r2_off<BT, LE>(r ^ (E / 2 - 1))] : Sr[r2_off<BT, LE>(r)];
        } else {
#pragma unroll
            for (int r = 0; r < E; ++r) k[r] = Sr[r2_off<BT, LE>(r)];
        }
    }

    // transpose through a workgroup barrier (BF or BT > 6); S = stage (for the reversal's activity test)
    template <int BF, int BT, bool REV, int S>
    static __device__ __forceinline__ void transpose_global(double (&k)[E], double *Sm, int t, int n_act,
                                                            bool wreal, double maxkey) {
        if constexpr (BF <= 6) {
            if (wreal) {
                double *Sw = Sm + r2_base<BF, LE>(t);
#pragma unroll
                for (int r = 0; r < E; ++r) Sw[r2_off<BF, LE>(r)] = k[r];
            }
        } else {
            const int cf = cnt_of<BF>(t, n_act);
            double *Sw = Sm + r2_base<BF, LE>(t);
#pragma unroll
            for (int r = 0; r < E; ++r)
                if (r < cf) Sw[r2_off<BF, LE>(r)] = k[r];
        }
        __syncthreads();
        if constexpr (BT <= 6) {
            static_assert(!REV || BT > 6, "a stage's first window is entered from layout 0");
            if (wreal) {
                const double *Sr = Sm + r2_base<BT, LE>(t);
#pragma unroll
                for (int r = 0; r < E; ++r) k[r] = Sr[r2_off<BT, LE>(r)];
            }
        } else {
            const int ct = cnt_of<BT>(t, n_act);
            const double *Sr = Sm + r2_base<BT, LE>(t);
            bool act = false;
            if constexpr (REV) {
                int a = n_act > (((t >> BT) << S) + (1 << (S - 1)));
                act = __builtin_amdgcn_readfirstlane(a) != 0;
            }
            if (REV && act) {
                // active block: its lower half is entirely real and is read reversed
                const double *Sf = Sm + r2_base<BT, LE>(t ^ ((1 << BT) - 1));
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    if (r < E / 2) k[r] = Sf[r2_off<BT, LE>(r ^ (E / 2 - 1))];
                    else k[r] = (r < ct) ? Sr[r2_off<BT, LE>(r)] : maxkey;
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; ++r) k[r] = (r < ct) ? Sr[r2_off<BT, LE>(r)] : maxkey;
            }
            // the reversed read took slots that other waves own in this window: they must not be
            // rewritten (next transpose) before every wave has read them
            if constexpr (REV) __syncthreads();
        }
    }

    template <int B, int HI, int LO>
    static __device__ __forceinline__ void levels(double (&k)[E]) {
#pragma unroll
        for (int j = HI; j >= LO; --j) {
            const int jr = j - B;
#pragma unroll
            for (int r = 0; r < E; ++r)
                if (!((r >> jr) & 1)) r2_cmpx(k[r], k[r | (1 << jr)]);
        }
    }

    template <int S, int K, int BPREV>
    static __device__ __forceinline__ void windows(double (&k)[E], double *Sm, int t, int n_act, bool wreal,
                                                   double maxkey) {
        constexpr int B = wb(S, K);
        constexpr int HI = (K == 0) ? S - 1 : BPREV - 1;
        constexpr bool REV = (K == 0);
        constexpr bool GLOBAL = (B > 6) || (BPREV > 6);
        if constexpr (B != BPREV) {
            if constexpr (GLOBAL) transpose_global<BPREV, B, REV, S>(k, Sm, t, n_act, wreal, maxkey);
            else if (wreal) transpose_local<BPREV, B, REV>(k, Sm, t);
        }
        if (B > 6 || wreal) levels<B, HI, B>(k);
        if constexpr (B > 0) windows<S, K + 1, B>(k, Sm, t, n_act, wreal, maxkey);
    }

    template <int S, int SLIM = 99>
    static __device__ __forceinline__ void stage(double (&k)[E], double *Sm, int t, int n_act, bool wreal,
                                                 double maxkey) {
        if constexpr (S <= LE) {
            if (wreal) {
                // mirror comparators of the normalised network, all inside the thread
#pragma unroll
                for (int r = 0; r < E; ++r)
                    if (!((r >> (S - 1)) & 1)) r2_cmpx(k[r], k[r ^ ((1 << S) - 1)]);
                if constexpr (S >= 2) levels<0, S - 2, 0>(k);
            }
        } else {
            windows<S, 0, 0>(k, Sm, t, n_act, wreal, maxkey);
        }
        if constexpr (S < LN && S < SLIM) stage<S + 1, SLIM>(k, Sm, t, n_act, wreal, maxkey);
    }

    // SLIM: stop after stage SLIM (timing experiments).  A variant that replaced the cross-wave stages by
    // merge-path rounds (partition search + E sequential take-the-smaller steps per thread and round)
    // was measured 1.5x slower than those stages: ~40 dependent LDS reads per round at ~300 cycles each.
    template <int SLIM = 99>
    static __device__ __forceinline__ void sort(double (&k)[E], double *Sm, int t, int n_act, bool wreal,
                                                double maxkey) {
        stage<1, SLIM>(k, Sm, t, n_act, wreal, maxkey);
    }
};

// Alternative slot map for a search image that is filled by a plain coalesced copy (mbd_rank_big): the
// probes of a descent are p = m * 2^j - 1, which a linear or padded image puts on one or two banks for
// the upper levels; XOR-ing bits 5..9 and 10..14 of p into the low five bits spreads them.  Measured:
// -13 % on the chunked kernel; on mbd_rank2 (image written from the sort's registers, one extra
// barrier, more address arithmetic per probe) the padded image is faster, so that kernel keeps it.
__device__ __forceinline__ int r2_swz(int p) { return p ^ (((p >> 5) ^ (p >> 10)) & 31); }

// Fixed-depth descent over the sorted search image restricted to [0, n_act): number of keys < x
// (INCL = false) or <= x (INCL = true).  The descent alone tops out at N - 1; FINAL adds the
// closing probe that makes N reachable.  It can be omitted for the strict count when x itself
// is one of the keys (then at most N - 1 keys are below it).
// SLOT: slot map of the image: SlotPad<LE> (the transpose image, mbd_rank2) or SlotSwz (mbd_rank_big).
template <int LE>
struct SlotPad {
    static __device__ __forceinline__ int at(int p) { return r2_phys<LE>(p); }
};
struct SlotSwz {
    static __device__ __forceinline__ int at(int p) { return r2_swz(p); }
};

template <int N, class SLOT, bool INCL, bool FINAL = true>
__device__ __forceinline__ int r2_bound(const double *Sm, int n_act, double x, double big) {
    int c = 0;
#pragma unroll
    for (int s = N >> 1; s >= 1; s >>= 1) {
        int pos = c + s - 1;
        double a = (pos < n_act) ? Sm[SLOT::at(pos)] : big;
        bool go = INCL ? (a <= x) : (a < x);
        c += go ? s : 0;
    }
    if (FINAL) {
        double a = (c < n_act) ? Sm[SLOT::at(c)] : big;
        bool go = INCL ? (a <= x) : (a < x);
        c += go ? 1 : 0;
    }
    return c;
}

}  // namespace sd
