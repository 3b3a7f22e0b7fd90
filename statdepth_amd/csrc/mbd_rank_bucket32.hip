// mbd_rank_bucket32.hip -- K1+K2 bucket ranking on 32-bit key images, TWO workgroups per CU (J = 2, 4096 < n <= 11264).
//
// Same integers as rank_bucket_kernel (mbd_rank_bucket.hip), the pairwise kernel and the reference's enumeration
// (_functional.py:246-251, _containment.py:75-77).  The fp64 kernel keeps one row per CU in LDS and its five
// barrier-separated phases leave the VALU idle while the LDS works and the other way round (SQ counters: VALU busy 57 %,
// LDS busy 37 %, together the kernel's whole time).  Here a row's keys live in LDS as 32-bit images
//     q(x) = trunc((x - lo) * (2^31 - 256) / (hi - lo))          -- non-decreasing in x whatever the data
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
#ifndef R32_LNB
#define R32_LNB 14                               // 16 384 buckets: histogram 32 KiB
#endif
constexpr int R32_NT = 512, R32_NW = 8;
constexpr int R32_LCAP = 64;                    // set-aside keys per workgroup before it hands all its rows over
constexpr int R32_LIST_WORDS = 1 + 2 * R32_LCAP;   // a workgroup's list in the workspace: count, keys, (B0 | E0 << 16)
constexpr int R32_PAD = 12;                     // sentinel images behind the keys (two quads past the last partial quad)
constexpr double R32_TOP = 2147483392.0;        // 2^31 - 256: the largest key image (31 bits: y < q <=> bit 31 of y - q)
constexpr u32 R32_SENT = 0x7FFFFFFFu;           // sentinel image behind the keys: above every key image, below 2^31

template <int E, int LNB>
struct R32Cfg {
    static constexpr int NT = R32_NT, NW = R32_NW, NB = 1 << LNB;
    static constexpr int QW = NB / 2 / NT / 4;                  // 16-byte quads of histogram words per thread
    static_assert(QW >= 1 && NB / 2 == QW * 4 * NT, "whole quads of histogram words per thread");
    static constexpr size_t HDR = 4 * NW * 8 + NW * 4 + 32 + (size_t)R32_LCAP * 8;
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
    constexpr int NT = C::NT, NW = C::NW, NB = C::NB, QW = C::QW, SH = 31 - LNB;
    extern __shared__ double Sm[];
    const int n = (int)n64;
    double *red = Sm;                                                 // [2][NW][2] min/max partials
    u32 *wtot = reinterpret_cast<u32 *>(red + 4 * NW);                // [NW]
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
    auto row_range = [&](int parity) {
        double mn = INF, mx = -INF;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mn = rb_mm<false>(mn, x[e]);
            mx = rb_mm<true>(mx, x[e]);
        }
        mn = rb_wave_allreduce<false>(mn);
        mx = rb_wave_allreduce<true>(mx);
        double *rp = red + parity * 2 * NW;
        if ((t & 63) == 63) { rp[2 * (t >> 6)] = mn; rp[2 * (t >> 6) + 1] = mx; }
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
        load_row(r);                                                  // pass (the other workgroup's, half a row away) fill in
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
                    if (cnt + off > 8u) {                             // the rest of a long bucket (a few lanes per visit)
                        const uint4 *p = S4 + (base >> 2);
#pragma unroll 1
                        for (u32 kk = 8; kk < cnt + off; kk += 4) {
                            const uint4 y = p[kk >> 2];
                            lt = __builtin_amdgcn_alignbit(lt, y.x - qe, 31);
                            lt = __builtin_amdgcn_alignbit(lt, y.y - qe, 31);
                            lt = __builtin_amdgcn_alignbit(lt, y.z - qe, 31);
                            lt = __builtin_amdgcn_alignbit(lt, y.w - qe, 31);
                        }
                    }
                    const u32 B = base - off + (u32)__popc(lt);       // keys with a smaller image
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
    u32 *P = partial + (size_t)blockIdx.x * n;
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
    if (t0 == 0) {
        u32 *dbg = listbuf + (size_t)gridDim.x * R32_LIST_WORDS + (size_t)blockIdx.x * 4;
        dbg[0] = (u32)((long long)__builtin_readcyclecounter() - t_entry);
        dbg[1] = (u32)rt_entry;
        dbg[2] = (u32)__builtin_amdgcn_s_memrealtime();
        dbg[3] = misc[0] | (rowidx << 16);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool rank_bucket32_supported(i64 n, i64 rows, int cus) {
    // u32 totals per workgroup, for this kernel's workgroups and for the fp64 kernel's when every row is handed over
    const u64 per = (u64)((rows + cus - 1) / cus) * (u64)n * (u64)n;
    return n > 4096 && n <= 11264 && rows >= cus && rows <= 4096 && per < ((u64)1 << 32) && xswitch("SD_RB_NO32") == 0;
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
size_t rank_bucket32_list_bytes(int G) { return (size_t)G * R32_LIST_WORDS * 4 + (size_t)G * 16; }

int launch_rank_bucket32(const double *Y, i64 n, i64 row0, i64 rows, u32 *partial, unsigned char *rowflag, u32 *gate, u32 epoch,
                         u64 *out_zero, u32 *listbuf, int G, hipStream_t s) {
    switch ((int)((n + 1023) / 1024)) {
        case 5: return launch32_cfg<10>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 6: return launch32_cfg<12>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 7: return launch32_cfg<14>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 8: return launch32_cfg<16>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 9: return launch32_cfg<18>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 10: return launch32_cfg<20>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
        case 11: return launch32_cfg<22>(Y, n, row0, rows, partial, rowflag, gate, epoch, out_zero, listbuf, G, s);
    }
    return fail(SD_ERR_UNSUPPORTED, "bucket32 kernel covers 4096 < n <= 11264");
}

u32 rank_bucket32_epoch() {
    static std::atomic<u32> counter{0};
    u32 e = ++counter;
    if (e == 0) e = ++counter;
    return e;
}

}  // namespace sd
