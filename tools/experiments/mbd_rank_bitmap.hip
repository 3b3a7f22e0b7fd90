// mbd_rank_bitmap.hip -- K1+K2 for J = 2, n <= 10240: ranks from an occupancy BITMAP instead of a histogram of keys.
//
// Same integers as every other path (the reference's enumeration, _functional.py:246-251 with _containment.py:75-77
// inside: per (curve, timepoint) B = others strictly below, A = strictly above, folded as C(n-1,2) - C(A,2) - C(B,2)).
//
// rank_bucket_kernel (mbd_rank_bucket.hip) sorts every key into a bucket of a 2^15-bucket histogram and then lets
// every key compare itself with its bucket mates: scatter + three random 16-byte LDS reads + 12 fp64 compares per
// key.  Here a key is first turned into a 32-bit image q = floor((x - lo) * 2^32 / (hi - lo)) (monotone in x for ANY
// lo, hi: fl(fl(x - lo) * s) and the truncation are non-decreasing), and its top 18 bits address ONE BIT of a
// 2^18-cell occupancy map -- 16 times the resolution a 16-bit counter array of the same size gives.  With 10^4 keys
// in 2.6 * 10^5 cells nine keys out of ten are alone in their cell, and for those
//      B = (keys in the 32-cell blocks before mine) + popcount(occupied cells of my block below mine),  A = n-1-B,
// with no scatter, no member reads and no compares: a cell order IS the value order.  Per 32-cell block one 8-byte
// record {occupancy word, counter}: a key that finds its bit already set is an "extra" and counts itself on the
// record's second word; a prefix sum over popcount(word) + extras turns the second word into (keys before the block |
// keys in the block << 16).  A block whose key count differs from its popcount holds a collision somewhere: all its
// keys (typically 4-6, about 15 % of a config-2 row) go the slow way -- COMPACTED into a per-wave work list first, so
// that the slow code runs on full waves instead of 15 % of the lanes of every instruction: scatter of the 32-bit
// images into rank order (block base + slot; first arrivals take the popcount slot, extras the counter slot), one
// barrier, every listed key counts the images of its block that are < / <= its own.  Images that TIE (two keys within
// range / 2^32 of each other, or equal values) are settled by that key alone with the fp64 values of the tied curves
// (their indices travel in the low 14 bits of the scattered word).  The fold of a slow key goes to a per-curve LDS
// accumulator by ds_add, and so does a fast key's (contained = A B when nothing ties), once per row.
//
// Rows this structure is not made for -- a NaN or an infinity, all values equal, a block above 64 keys (tie-heavy or
// clustered data), more slow keys than a wave's list holds -- are SET ASIDE: flagged in `rowsel` and ranked by
// rank_bucket_kernel in a second launch that returns at once when nothing was flagged.  A workgroup that had to set a
// row aside for crowding stops trying (tie-heavy data costs one attempt per workgroup, not one per row).
//
// HBM traffic: the matrix once + G partial blocks of 4 n bytes, as before.
#include <stdio.h>
#include <stdlib.h>

#include "sd_common.h"
#include "rank_bucket.h"

namespace sd {

template <int E_>
struct BMCfg {
    static constexpr int E = E_;
    static constexpr int NT = 1024, NW = NT / 64;
    static constexpr int LM = 18;                        // 2^LM cells per row
    static constexpr int QSH = 32 - LM;                  // q >> QSH = cell
    static constexpr int NREC = 1 << (LM - 5);           // records: one per 32 cells
    static constexpr int QW = NREC / 2 / NT;             // 16-byte quads (2 records) per thread in the prefix sum
    static constexpr int CAPC = 64;                      // keys per block above which the row is set aside
    static constexpr int WLCAP = 192;                    // slow keys per wave and row
    static constexpr int NPAD = E * NT;
    static constexpr size_t HDR = 1024;
    static_assert(QW >= 1, "at least one quad per thread");
    static_assert(NPAD * 4 <= NREC * 8, "the scattered images overlay the records");
    static_assert(NPAD <= 16384, "curve index travels in 14 bits");
    static constexpr size_t lds_bytes() {
        return HDR + (size_t)NREC * 8 + (size_t)NPAD * 4 + (size_t)NW * WLCAP * 8;
    }
};

// DBG (timing experiments, SD_TUNING builds only; results invalid): 1 = stop after the range, 2 = after the bit phase,
// 3 = after the prefix sum, 4 = after the rank phase (no slow keys)
template <int E, int DBG = 0>
__global__ __launch_bounds__(1024, 4) void rank_bitmap_kernel(const double *__restrict__ Y, i64 n64, i64 row0, i64 rows,
                                                             u32 *__restrict__ partial, unsigned char *__restrict__ rowsel,
                                                             u32 *__restrict__ wgdefer) {
    using C = BMCfg<E>;
    constexpr int NT = C::NT, NW = C::NW, NREC = C::NREC, QW = C::QW, QSH = C::QSH;
    constexpr int BATCH = 5;                                          // keys of a thread whose LDS round trips overlap
    extern __shared__ double Sm[];
    const int n = (int)n64;
    double *red = Sm;                                                 // [2][NW][2] min / max partials
    u32 *wnan = reinterpret_cast<u32 *>(red + 4 * NW);                // [2][NW] "this wave saw a NaN"
    u32 *wtot = wnan + 2 * NW;                                        // [NW] wave totals of the prefix sum
    u32 *flg = wtot + NW;                                             // [0]: a wave's work list overflowed
    u32 *REC = reinterpret_cast<u32 *>(Sm + C::HDR / 8);              // NREC x {occupancy word, counter / prefix}
    u32 *S = REC;                                                     // slow keys' images in rank order (overlay)
    u32 *accL = REC + 2 * NREC;                                       // per-curve totals of the slow keys
    uint2 *WL = reinterpret_cast<uint2 *>(accL + C::NPAD);            // [NW][WLCAP] work lists
    const int t0 = threadIdx.x;
    const double INF = __builtin_huge_val();
    int t = t0;
    const bool vlast = t0 + (E - 1) * NT < n;                         // the thread's last key exists

    {   // LDS setup: empty records, zero totals
        uint4 *R4 = reinterpret_cast<uint4 *>(REC);
#pragma unroll
        for (int i = 0; i < QW; ++i) R4[i * NT + t] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < E; ++e) accL[t + e * NT] = 0;
        if (t < 4) flg[t] = 0;
    }

    double k[E];
    // the next row's loads go out in two halves -- behind the bit phase and behind the rank phase -- so that the key
    // registers in flight never coexist with all of the images (E = 10 would not fit 128 VGPRs otherwise)
    constexpr int EH = E / 2;
    auto load_part = [&](i64 r, int e0, int e1) {
        const double *rp = Y + (row0 + r) * n + t;
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (e >= e0 && e < e1) k[e] = (e < E - 1 || vlast) ? rp[e * NT] : rp[0];   // a missing key repeats the first
    };
    auto load_row = [&](i64 r) { load_part(r, 0, E); };
    // range of a row + "has a NaN" per wave into LDS; computed for the NEXT row at the end of every iteration
    auto row_range = [&](int parity) {
        double mn = INF, mx = -INF;
        bool isn = false;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mn = rb_mm<false>(mn, k[e]);
            mx = rb_mm<true>(mx, k[e]);
            isn |= k[e] != k[e];
        }
        mn = rb_wave_allreduce<false>(mn);
        mx = rb_wave_allreduce<true>(mx);
        const bool wn = __ballot(isn) != 0;
        if ((t & 63) == 63) {
            double *rp = red + parity * 2 * NW;
            rp[2 * (t >> 6)] = mn;
            rp[2 * (t >> 6) + 1] = mx;
            wnan[parity * NW + (t >> 6)] = wn ? 1u : 0u;
        }
    };

    // DBG == 9: cycles between the marks below, summed over the rows, for waves 0 and 15 of workgroup 0 (the kernel is
    // otherwise complete; cdna_hip_programming.md 7, in-kernel stamps)
    long long stamp[16], tlast = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) stamp[i] = 0;
    auto mark = [&](int ph) {
        if constexpr (DBG == 9) {
            const long long now = (long long)__builtin_readcyclecounter();
            stamp[ph] += now - tlast;
            tlast = now;
        }
    };
    const u32 nm1 = (u32)n - 1u;
    const u32 R2 = nm1 * (nm1 - 1u);                                  // 2 C(n-1, 2)  (< 2^28)
    u32 ndefer = 0;                                                   // block-uniform
    bool giveup = false;                                              // crowded data: stop trying, set every row aside
    int par = 0;
    if ((i64)blockIdx.x < rows) {
        load_row(blockIdx.x);
        row_range(0);
    }
    if constexpr (DBG == 9) tlast = (long long)__builtin_readcyclecounter();
    for (i64 r = blockIdx.x; r < rows; r += gridDim.x) {
        const i64 rnext = r + gridDim.x;
        // per-row opaque copy of the thread id (see rank_bucket_kernel: keeps address registers out of the row loop)
        t = t0;
        asm volatile("" : "+v"(t));
        const int lane = t & 63;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        if (giveup) {                                                 // block-uniform
            if (t == 0) rowsel[r] = 1;
            ++ndefer;
            continue;
        }
        const double *redp = red + par * 2 * NW;
        const u32 *wnp = wnan + par * NW;
        par ^= 1;
        mark(0);
        __syncthreads();                                              // barrier 1: range partials; records are empty
        mark(1);
        if (t == 0) flg[0] = 0;                                       // every wave has read the previous row's flag
        double lo, hi;
        bool anynan;
        {
            const double2 p = reinterpret_cast<const double2 *>(redp)[lane & 15];
            lo = rb_readlane_f64(rb_row_allreduce<false>(p.x), 0);
            hi = rb_readlane_f64(rb_row_allreduce<true>(p.y), 0);
            anynan = __ballot(wnp[lane & 15] != 0) != 0;
            if constexpr (E >= 2) {
                // outlier-robust range, as in rank_bucket_kernel: the waves hold 16 random subsets of the row, the
                // innermost of their minima / maxima bracket the bulk; tails clamp into the end cells, tie there and
                // are settled in fp64 (or crowd the end block and set the row aside): monotone for ANY lo, hi
                const float l2 = rb_row_allreduce_f32<true>((float)p.x), h2 = rb_row_allreduce_f32<false>((float)p.y);
                const double lo2 = (double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(l2)));
                const double hi2 = (double)__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(h2)));
                const double sp = hi2 - lo2;
                if (sp > 0.0 && sp < INF && (hi - lo) > 8.0 * sp) {
                    const double nlo = lo2 - 1.5 * sp, nhi = hi2 + 1.5 * sp;
                    lo = nlo > lo ? nlo : lo;
                    hi = nhi < hi ? nhi : hi;
                }
            }
        }
        const double scale = 4294967296.0 / (hi - lo);
        // a NaN, an infinity, no two different values, a range that over- or underflows: not for this kernel
        bool go = !anynan && (hi > lo) && (scale < INF) && (scale > 0.0) && (lo > -INF) && (hi < INF);
        if constexpr (DBG == 1) go = false;
        u32 q[E], kk[(E + 3) / 4];                                    // kk: extras' slot + 1, a byte per key
#pragma unroll
        for (int e = 0; e < (E + 3) / 4; ++e) kk[e] = 0;
        bool crowded = false, over = false;
        const bool dirty = go;                                        // the records get bits: empty them afterwards
        if (go) {
            // ---- (1) image, occupancy bit, extras count themselves.  Branch-free and in batches -- all the atomics of a
            //      thread's keys are in flight together (a use inside a branch per key would wait for every LDS round
            //      trip in turn); the extras' second atomic has few active lanes and its result is used last ----
#pragma unroll
            for (int e0 = 0; e0 < E; e0 += BATCH) {
                u32 old[BATCH], bit[BATCH];
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int e = e0 + b;
                    if (e >= E) continue;
                    const double u = (k[e] - lo) * scale;
                    u32 qq;
                    asm("v_cvt_u32_f64 %0, %1" : "=v"(qq) : "v"(u));  // saturating; below the range -> 0
                    q[e] = qq;
                    bit[b] = (e < E - 1 || vlast) ? 1u << ((qq >> QSH) & 31u) : 0u;   // a missing key sets nothing
                    old[b] = atomicOr(&REC[2 * (qq >> (QSH + 5))], bit[b]);
                }
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int e = e0 + b;
                    if (e >= E) continue;
                    const bool extra = (old[b] & bit[b]) != 0;
                    bit[b] = extra ? 1u : 0u;
                    old[b] = 0;
                    if (extra) old[b] = atomicAdd(&REC[2 * (q[e] >> (QSH + 5)) + 1], 1u);   // few lanes: cheap in the LDS
                }
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int e = e0 + b;
                    if (e >= E) continue;
                    const u32 kx = old[b] < 254u ? old[b] + 1u : 255u;
                    kk[e / 4] |= (bit[b] ? kx : 0u) << (8 * (e % 4));
                }
                asm volatile("" ::: "memory");                        // one batch at a time (register budget)
            }
        }
        mark(2);
        if (rnext < rows) load_part(rnext, 0, EH);                    // the key registers are free: half of the next row in flight
        if (go) {
            __syncthreads();                                          // barrier 2
            mark(3);
            if constexpr (DBG == 2) go = false;
        }
        if (go) {
            // ---- (2) exclusive prefix sum over popcount(word) + extras (16-byte accesses, lane <-> quad) ----
            uint4 *R4 = reinterpret_cast<uint4 *>(REC) + wave * (64 * QW);
            u32 c0[QW], c1[QW], incl[QW], offq[QW], wsum = 0, cmax = 0;
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                const uint4 rq = R4[i * 64 + lane];
                c0[i] = __popc(rq.x) + rq.y;
                c1[i] = __popc(rq.z) + rq.w;
                cmax = max(cmax, max(c0[i], c1[i]));
                incl[i] = rb_wave_incl_scan(c0[i] + c1[i]);
                offq[i] = wsum;
                wsum += rb_readlane(incl[i], 63);
            }
            const bool wcrowd = __ballot(cmax > (u32)C::CAPC) != 0;
            if (lane == 63) wtot[wave] = wsum | (wcrowd ? 0x80000000u : 0u);
            mark(4);
            __syncthreads();                                          // barrier 3
            mark(5);
            const u32 wt = wtot[lane & 15];
            crowded = __ballot((wt >> 31) != 0) != 0;
            if (!crowded) {
                const u32 wscan = rb_row_incl_scan(wt & 0x7FFFFFFFu);
                const u32 woff = wave ? rb_readlane(wscan, wave - 1) : 0u;
                u32 *R1 = reinterpret_cast<u32 *>(R4);
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    const u32 base = woff + offq[i] + incl[i] - c0[i] - c1[i];
                    R1[(i * 64 + lane) * 4 + 1] = base | (c0[i] << 16);
                    R1[(i * 64 + lane) * 4 + 3] = (base + c0[i]) | (c1[i] << 16);
                }
                mark(6);
                __syncthreads();                                      // barrier 4: records complete
                mark(7);
            }
        }
        bool rankit = go && !crowded;
        if constexpr (DBG == 3) rankit = false;
        u32 wcnt = 0;                                                 // slow keys of this wave (wave-uniform)
        bool half2 = false;                                           // second half of the next row requested
        if (rankit) {
            // ---- (3) rank: fast keys in place, keys of blocks with a collision go to the wave's work list.  First
            //      all the record reads and the arithmetic (no branch: the reads of a thread's keys are in flight
            //      together), then the appends ----
            uint2 *WLw = WL + wave * C::WLCAP;
#pragma unroll
            for (int e0 = 0; e0 < E; e0 += BATCH) {
                u32 w1[BATCH], slowbits = 0;
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int e = e0 + b;
                    if (e >= E) continue;
                    const u32 qq = q[e];
                    const uint2 rc = *reinterpret_cast<const uint2 *>(&REC[2 * (qq >> (QSH + 5))]);
                    const u32 below = (1u << ((qq >> QSH) & 31u)) - 1u;
                    const u32 inb = __popc(rc.x & below);
                    const u32 p0 = rc.y & 0xFFFFu, cnt = rc.y >> 16, cw = __popc(rc.x);
                    const bool valid = (e < E - 1 || vlast);
                    const bool slow = valid && (cnt != cw);
                    const u32 B = p0 + inb;
                    const u32 kx = (kk[e / 4] >> (8 * (e % 4))) & 0xFFu;
                    const u32 slot = kx ? cw + kx - 1u : inb;
                    w1[b] = p0 | (cnt << 14) | (slot << 21);
                    slowbits |= slow ? 1u << b : 0u;
                    // nothing ties around a fast key: contained = A B.  A slow key keeps its image, without the lowest
                    // bit: 19 image bits tell the keys of a block apart and the curve index takes 14 of the word's 32
                    // (a tie in the remaining 31 bits is settled in fp64 like any other)
                    q[e] = slow ? ((qq >> 1) << 14) | (u32)(t + e * NT) : (valid ? __umul24(B, nm1 - B) : 0u);
                }
#pragma unroll
                for (int b = 0; b < BATCH; ++b) {
                    const int e = e0 + b;
                    if (e >= E) continue;
                    const bool slow = (slowbits >> b) & 1u;
                    const u64 sm = __ballot(slow);
                    const u32 pos = wcnt + __builtin_amdgcn_mbcnt_hi((u32)(sm >> 32), __builtin_amdgcn_mbcnt_lo((u32)sm, 0u));
                    if (slow && pos < (u32)C::WLCAP) WLw[pos] = make_uint2(q[e], w1[b]);
                    wcnt += (u32)__popcll(sm);
                    q[e] = slow ? 0u : q[e];
                }
                asm volatile("" ::: "memory");                        // one batch at a time (register budget)
            }
            wcnt = __builtin_amdgcn_readfirstlane(wcnt);
            if (wcnt > (u32)C::WLCAP && lane == 0) flg[0] = 1u;
            if (rnext < rows) load_part(rnext, EH, E);                // the other half of the next row
            half2 = true;
            mark(8);
            __syncthreads();                                          // barrier 5: records are dead, lists complete
            mark(9);
            over = flg[0] != 0;
            if constexpr (DBG == 4) over = true;
            if (!over) {
                // the row counts: commit the fast keys' folds (per-curve totals live in LDS: ten accumulator registers
                // would not fit beside the row in flight; a conflict-free ds_add per key)
#pragma unroll
                for (int e = 0; e < E; ++e) atomicAdd(&accL[t + e * NT], q[e]);
                // ---- (4) slow keys: scatter the images into rank order ----
                for (u32 j = lane; j < wcnt; j += 64) {
                    const uint2 en = WLw[j];
                    S[(en.y & 0x3FFFu) + (en.y >> 21)] = en.x;
                }
                mark(10);
                __syncthreads();                                      // barrier 6
                mark(11);
                // ---- (5) ... and count the block's images below / not above the own one ----
                const double *Yr = Y + (row0 + r) * n;
                // the wave's (at most WLCAP = 3 x 64) entries are worked on together: all the list reads, then the first
                // four images of every block, are in flight at once (a chunk at a time would expose three LDS round
                // trips per chunk)
                constexpr int NCH = C::WLCAP / 64;
                uint2 en[NCH];
                u32 less[NCH], le[NCH], cntc[NCH];
                bool act[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    act[c] = c * 64 + lane < wcnt;
                    en[c] = WLw[act[c] ? c * 64 + lane : 0];
                }
                u32 y[NCH][4];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const u32 base = en[c].y & 0x3FFFu;
                    cntc[c] = act[c] ? (en[c].y >> 14) & 0x7Fu : 0u;
                    const u32 lastp = base + (cntc[c] ? cntc[c] - 1u : 0u);
#pragma unroll
                    for (int u = 0; u < 4; ++u) y[c][u] = S[min(base + u, lastp)];
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const u32 xlo = en[c].x & ~0x3FFFu, xhi = en[c].x | 0x3FFFu;
                    less[c] = le[c] = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool in = (u32)u < cntc[c];
                        less[c] += (in && y[c][u] < xlo) ? 1u : 0u;
                        le[c] += (in && y[c][u] <= xhi) ? 1u : 0u;
                    }
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const u32 base = en[c].y & 0x3FFFu, cnt = cntc[c];
                    const u32 xlo = en[c].x & ~0x3FFFu, xhi = en[c].x | 0x3FFFu;
                    const u32 lastp = base + (cnt ? cnt - 1u : 0u);
                    for (u32 i = 4; __ballot(i < cnt) != 0; i += 4) {    // blocks of more than four keys
                        u32 z[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) z[u] = S[min(base + i + u, lastp)];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool in = i + u < cnt;
                            less[c] += (in && z[u] < xlo) ? 1u : 0u;
                            le[c] += (in && z[u] <= xhi) ? 1u : 0u;
                        }
                    }
                    const bool tie = le[c] - less[c] > 1u;            // another key shares my image: settle in fp64
                    if (__ballot(tie) != 0) {
                        if (tie) {
                            const double vx = Yr[en[c].x & 0x3FFFu];
                            for (u32 i = 0; i < cnt; ++i) {
                                const u32 yy = S[base + i];
                                if ((yy >> 14) == (en[c].x >> 14) && yy != en[c].x) {
                                    const double vy = Yr[yy & 0x3FFFu];
                                    less[c] += (vy < vx) ? 1u : 0u;
                                    le[c] -= (vy <= vx) ? 0u : 1u;
                                }
                            }
                        }
                    }
                    if (act[c]) {
                        const u32 B = base + less[c], A = (u32)n - base - le[c];
                        const u32 c2 = R2 - __umul24(A, A - 1u) - __umul24(B, B - 1u);
                        atomicAdd(&accL[en[c].x & 0x3FFFu], c2 >> 1);
                    }
                }
                mark(12);
                mark(13);
                __syncthreads();                                      // barrier 7: the images are dead
                mark(14);
            }
        }
        if (!half2 && rnext < rows) load_part(rnext, EH, E);
        if (dirty) {
            // records (and whatever overlays them) back to empty for the next row
            uint4 *R4 = reinterpret_cast<uint4 *>(REC);
#pragma unroll
            for (int i = 0; i < QW; ++i) R4[i * NT + t] = make_uint4(0, 0, 0, 0);
        }
        const bool aside = (DBG == 0 || DBG == 9) && (!go || crowded || over);
        if (t == 0) rowsel[r] = aside ? 1 : 0;
        if (aside) ++ndefer;
        if (aside && (crowded || over)) giveup = true;
        if (rnext < rows && !giveup) row_range(par);
        mark(15);
    }
    if constexpr (DBG == 9) {
        if (blockIdx.x == 0 && (t0 == 0 || t0 == 960))
            printf("bm stamps wave %d (cycles over the workgroup's rows): ->b1 %lld | b1 wait %lld | range+bits %lld | b2 wait %lld | "
                   "prefixA %lld | b3 wait %lld | prefixB %lld | b4 wait %lld | rank+append %lld | b5 wait %lld | scatter %lld | "
                   "b6 wait %lld | slow compare %lld | commit %lld | b7 wait %lld | zero+next range %lld\n",
                   t0 >> 6, stamp[0], stamp[1], stamp[2], stamp[3], stamp[4], stamp[5], stamp[6], stamp[7], stamp[8], stamp[9],
                   stamp[10], stamp[11], stamp[12], stamp[13], stamp[14], stamp[15]);
    }
    t = t0;
    __syncthreads();
    // ---- this workgroup's partial totals (u32: the host checked ceil(rows / G) * C(n-1,2) < 2^32) ----
    u32 *P32 = partial + (size_t)blockIdx.x * n;
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e < E - 1 || vlast) P32[t + e * NT] = accL[t + e * NT];
    if (t == 0) wgdefer[blockIdx.x] = ndefer;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static bool bm_enabled() {
    const char *e = getenv("SD_RB_BITMAP");                           // 0: rank_bucket_kernel for every row
    return !(e && atoi(e) == 0);
}

bool mbd_rank_bitmap_supported(i64 rows, i64 n, int J, int G) {
    if (!bm_enabled() || J != 2 || n < 1025 || n > 10240 || G < 1) return false;
    const u64 per_wg = (u64)((rows + G - 1) / G) * ((u64)(n - 1) * (u64)(n - 2) / 2);
    return per_wg < ((u64)1 << 32);
}

template <int E>
static int launch_bitmap_cfg(const double *Y, i64 n, i64 row0, i64 rows, u32 *partial, unsigned char *rowsel, u32 *wgdefer,
                             int G, hipStream_t s) {
    using C = BMCfg<E>;
    auto kf = rank_bitmap_kernel<E>;
#ifdef SD_TUNING
    if constexpr (E == 10) {
        if (const char *d = getenv("SD_BM_DBG")) {
            switch (atoi(d)) {
                case 1: kf = rank_bitmap_kernel<E, 1>; break;
                case 2: kf = rank_bitmap_kernel<E, 2>; break;
                case 3: kf = rank_bitmap_kernel<E, 3>; break;
                case 4: kf = rank_bitmap_kernel<E, 4>; break;
                case 9: kf = rank_bitmap_kernel<E, 9>; break;
            }
        }
    }
#endif
    constexpr size_t lds = C::lds_bytes();
    static_assert(lds <= 163840, "LDS budget");
    SD_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kf, dim3(G), dim3(1024), lds, s, Y, n, row0, rows, partial, rowsel, wgdefer);
    SD_HIP(hipGetLastError());
    return SD_OK;
}

// rows [row0, row0 + rows) through the bitmap kernel: partial totals as u32[G][n], rowsel[r] = 1 for the rows it set
// aside, wgdefer[g] = how many of them workgroup g owns (rows g, g + G, ...)
int launch_rank_bitmap(const double *Y, i64 n, i64 row0, i64 rows, u32 *partial, unsigned char *rowsel, u32 *wgdefer, int G,
                       hipStream_t s) {
#define BM_ARGS Y, n, row0, rows, partial, rowsel, wgdefer, G, s
    switch ((int)((n + 1023) / 1024)) {
        case 2: return launch_bitmap_cfg<2>(BM_ARGS);
        case 3: return launch_bitmap_cfg<3>(BM_ARGS);
        case 4: return launch_bitmap_cfg<4>(BM_ARGS);
        case 5: return launch_bitmap_cfg<5>(BM_ARGS);
        case 6: return launch_bitmap_cfg<6>(BM_ARGS);
        case 7: return launch_bitmap_cfg<7>(BM_ARGS);
        case 8: return launch_bitmap_cfg<8>(BM_ARGS);
        case 9: return launch_bitmap_cfg<9>(BM_ARGS);
        case 10: return launch_bitmap_cfg<10>(BM_ARGS);
    }
#undef BM_ARGS
    return fail(SD_ERR_UNSUPPORTED, "bitmap kernel covers 1024 < n <= 10240");
}

}  // namespace sd
