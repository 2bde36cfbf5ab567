// wave_ops.h -- wave64 cross-lane primitives for gfx950 (CDNA4): DPP scans/reductions and the
// in-register k-th-smallest selection used by the cross-recurrence thresholding.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acoss {

// DPP controls (GFX9 encoding)
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
constexpr int DPP_WAVE_SHR1 = 0x138;

struct OpAdd { static __device__ inline int apply(int a, int b) { return a + b; } };
struct OpUMin { static __device__ inline int apply(int a, int b) { return (int)min((unsigned)a, (unsigned)b); } };
struct OpUMax { static __device__ inline int apply(int a, int b) { return (int)max((unsigned)a, (unsigned)b); } };

// Inclusive scan over the 64 lanes of a wave; lane 63 ends up with the reduction of all lanes.
// Lanes whose DPP source is outside the row keep `identity` (bound_ctrl = 0, old = identity).
template <typename Op>
__device__ inline int wave_scan(int v, int identity)
{
    v = Op::apply(v, __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR1, 0xf, 0xf, false));
    v = Op::apply(v, __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR2, 0xf, 0xf, false));
    v = Op::apply(v, __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR4, 0xf, 0xf, false));
    v = Op::apply(v, __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_SHR8, 0xf, 0xf, false));
    v = Op::apply(v, __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_BCAST15, 0xa, 0xf, false));
    v = Op::apply(v, __builtin_amdgcn_update_dpp(identity, v, DPP_ROW_BCAST31, 0xc, 0xf, false));
    return v;
}

// Wave-uniform reduction result (an SGPR after readlane).
template <typename Op>
__device__ inline int wave_reduce(int v, int identity)
{
    return __builtin_amdgcn_readlane(wave_scan<Op>(v, identity), 63);
}

__device__ inline int wave_sum(int v) { return wave_reduce<OpAdd>(v, 0); }
__device__ inline unsigned wave_umin(unsigned v) { return (unsigned)wave_reduce<OpUMin>((int)v, -1); }
__device__ inline unsigned wave_umax(unsigned v) { return (unsigned)wave_reduce<OpUMax>((int)v, 0); }

// Value of lane-1 (lane 0 receives `fill`).
__device__ inline int lane_shr1(int v, int fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHR1, 0xf, 0xf, false);
}
__device__ inline float lane_shr1(float v, float fill)
{
    return __int_as_float(lane_shr1(__float_as_int(v), __float_as_int(fill)));
}

// Order-preserving map of float64 onto uint64 (total order of the reals; -0.0 == +0.0).
__device__ inline uint64_t f64_key(double x)
{
    x = x + 0.0;  // -0.0 -> +0.0
    uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double f64_from_key(uint64_t k)
{
    uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// Result of a selection: mask rule is  (key < thr_key) || (key == thr_key && index <= cut).
// Number of lanes of the wave for which `pred` holds: the compare writes its lane mask straight into
// an SGPR pair and the scalar unit counts the bits, so a count costs one VALU instruction and no
// cross-lane reduction.
__device__ inline int wave_count(bool pred) { return __popcll(__ballot(pred)); }

struct SelectResult {
    uint64_t thr_key;
    int cut;
};

// k-th smallest (1-based, 1 <= k <= n) of the n keys a wave holds in registers, EPL per lane.
// idx[e] is the position (column or row number) of this lane's element e; elements with
// idx >= n are padding (their key must be UINT64_MAX).  Ties at the threshold are cut
// lowest-position first.  Wave-uniform result.
//
// Method: bit-serial search from the most significant differing bit over the high 32-bit words:
// per bit one v_cmp + v_addc per element and one DPP reduction (measured equal in time to the
// ballot + s_bcnt1 form, at a third of the registers; the rare low-word / tie phases use ballots).  The search tracks how many
// elements the current bucket [prefix, prefix + 2^b) still holds and stops as soon as it holds
// exactly one -- on real data after ~log2(n) bits -- fetching that element with a masked wave
// maximum.  Only if several elements share the whole high word does it continue over their low
// words, and only if full keys are equal does it rank positions.
template <int EPL>
__device__ inline SelectResult wave_select_kth(const uint64_t (&key)[EPL], const int (&idx)[EPL], int n, int k)
{
    unsigned hi[EPL], lo[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        hi[e] = (unsigned)(key[e] >> 32);
        lo[e] = (unsigned)key[e];
    }
    unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        const bool valid = idx[e] < n;
        mn = min(mn, valid ? hi[e] : 0xffffffffu);
        mx = max(mx, valid ? hi[e] : 0u);
    }
    mn = wave_umin(mn);
    mx = wave_umax(mx);
    SelectResult res;
    res.cut = 0x7fffffff;
    unsigned vh = mn;
    int c_lo = 0, c_hi = n;          // elements below the bucket / below its end
    int b = -1;
    if (mn != mx) {
        const int top = 31 - __clz((int)(mn ^ mx));
        vh = (top == 31) ? 0u : (mn >> (top + 1)) << (top + 1);
        for (b = top; b >= 0 && c_hi - c_lo > 1; b--) {
            const unsigned cand = vh | (1u << b);
            int c = 0;
#pragma unroll
            for (int e = 0; e < EPL; e++) c += (hi[e] < cand) ? 1 : 0;
            c = wave_sum(c);
            if (c <= k - 1) { vh = cand; c_lo = c; } else { c_hi = c; }
        }
    }
    if (c_hi - c_lo == 1) {
        // one element left in [vh, vh + 2^(b+1)): it is the k-th smallest
        const unsigned span = (b + 1 >= 32) ? 0xffffffffu : ((1u << (b + 1)) - 1u);
        unsigned mh = 0, ml = 0;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const bool in = (hi[e] - vh) <= span && idx[e] < n;
            mh = max(mh, in ? hi[e] : 0u);
            ml = max(ml, in ? lo[e] : 0u);
        }
        res.thr_key = ((uint64_t)wave_umax(mh) << 32) | wave_umax(ml);
        return res;
    }
    // several elements share the high word vh: rank k2 among them by the low word
    const int k2 = k - c_lo;
    int d_lo = 0, d_hi = c_hi - c_lo;
    unsigned vl = 0;
    for (b = 31; b >= 0 && d_hi - d_lo > 1; b--) {
        const unsigned cand = vl | (1u << b);
        int c = 0;
#pragma unroll
        for (int e = 0; e < EPL; e++) c += wave_count(hi[e] == vh && lo[e] < cand && idx[e] < n);
        if (c <= k2 - 1) { vl = cand; d_lo = c; } else { d_hi = c; }
    }
    if (d_hi - d_lo == 1) {
        const unsigned span = (b + 1 >= 32) ? 0xffffffffu : ((1u << (b + 1)) - 1u);
        unsigned ml = 0;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const bool in = hi[e] == vh && (lo[e] - vl) <= span && idx[e] < n;
            ml = max(ml, in ? lo[e] : 0u);
        }
        res.thr_key = ((uint64_t)vh << 32) | wave_umax(ml);
        return res;
    }
    // exact ties: d_hi - d_lo elements equal (vh, vl); `need` of them are taken, lowest position first
    res.thr_key = ((uint64_t)vh << 32) | vl;
    const int equal = d_hi - d_lo, need = k2 - d_lo;
    if (equal > need) {
        // the need-th smallest position among the tied elements (bit-serial over positions)
        int cut = 0;
        for (int pb = 30; pb >= 0; pb--) {
            const int cand = cut | (1 << pb);
            int c = 0;
#pragma unroll
            for (int e = 0; e < EPL; e++) c += wave_count(key[e] == res.thr_key && idx[e] < cand && idx[e] < n);
            if (c <= need - 1) cut = cand;
        }
        res.cut = cut;
    }
    return res;
}

// ---------------------------------------------------------------------------------------------
// Faster selection for 16 elements per lane.  Counting "how many keys are below the probe" costs
// 16 x (v_cmp + v_addc) = 32 half-rate VALU instructions per probe in wave_select_kth; here each lane
// first sorts the high words of its 16 keys once (63-comparator odd-even merge network, v_min/v_max),
// after which a probe is a 4-level binary search through v_cndmask trees (5 compares + 11 selects).
// The search itself bisects a value bracket [L, U) with known counts and can be warm-started from the
// threshold of the neighbouring row (rows of the windowed sums change slowly): it gallops outwards
// from the previous threshold and bisects the last galloping interval, ~10 probes instead of ~19.
// ---------------------------------------------------------------------------------------------
__device__ inline void sort16_u32(unsigned (&s)[16])
{
    constexpr int net[63][2] = {
        {0,1},{2,3},{0,2},{1,3},{1,2},{4,5},{6,7},{4,6},{5,7},{5,6},{0,4},{2,6},{2,4},{1,5},{3,7},{3,5},
        {1,2},{3,4},{5,6},{8,9},{10,11},{8,10},{9,11},{9,10},{12,13},{14,15},{12,14},{13,15},{13,14},
        {8,12},{10,14},{10,12},{9,13},{11,15},{11,13},{9,10},{11,12},{13,14},{0,8},{4,12},{4,8},{2,10},
        {6,14},{6,10},{2,4},{6,8},{10,12},{1,9},{5,13},{5,9},{3,11},{7,15},{7,11},{3,5},{7,9},{11,13},
        {1,2},{3,4},{5,6},{7,8},{9,10},{11,12},{13,14}};
#pragma unroll
    for (int c = 0; c < 63; c++) {
        const unsigned a = s[net[c][0]], b = s[net[c][1]];
        s[net[c][0]] = min(a, b);
        s[net[c][1]] = max(a, b);
    }
}

// number of entries of the ascending s[0..15] that are < cand
__device__ inline int count_below_sorted16(const unsigned (&s)[16], unsigned cand)
{
    const bool b3 = s[7] < cand;
    const unsigned t2 = b3 ? s[11] : s[3];
    const bool b2 = t2 < cand;
    const unsigned u0 = b2 ? s[5] : s[1], u1 = b2 ? s[13] : s[9];
    const bool b1 = (b3 ? u1 : u0) < cand;
    const unsigned w0 = b1 ? s[2] : s[0], w1 = b1 ? s[6] : s[4], w2 = b1 ? s[10] : s[8], w3 = b1 ? s[14] : s[12];
    const unsigned x0 = b2 ? w1 : w0, x1 = b2 ? w3 : w2;
    const bool b0 = (b3 ? x1 : x0) < cand;
    return (b3 ? 8 : 0) + (b2 ? 4 : 0) + (b1 ? 2 : 0) + (b0 ? 1 : 0) + (s[15] < cand ? 1 : 0);
}

// order-preserving 32-bit halves of a float64's key (the value must already have -0.0 folded to +0.0)
__device__ inline unsigned key_hi(double x)
{
    const unsigned h = (unsigned)__double2hiint(x);
    return (h & 0x80000000u) ? ~h : (h | 0x80000000u);
}
__device__ inline unsigned key_lo(double x)
{
    const unsigned l = (unsigned)__double2loint(x);
    return ((unsigned)__double2hiint(x) & 0x80000000u) ? ~l : l;
}
__device__ inline uint64_t key_of(double x) { return ((uint64_t)key_hi(x) << 32) | key_lo(x); }

// element positions of the two register layouts used by the selection kernels
struct RowIdx {   // rows: lane l holds elements 128*q + 2*l + {0,1}
    int lane;
    __device__ inline int operator()(int e) const { return 128 * (e >> 1) + 2 * lane + (e & 1); }
};
struct ColIdx {   // columns staged through LDS: element e of lane l is row e*64 + l
    int lane;
    __device__ inline int operator()(int e) const { return e * 64 + lane; }
};

// cut value with which wave_select16 marks a row it could not resolve on its fast path (several keys share
// the winning high word, or exact ties): select_fix_kernel re-does those rows with wave_select_kth.
constexpr int SELECT_UNRESOLVED = -2;

// k-th smallest (1-based) of the n values a wave holds, 16 per lane (x[e] at position idx_of(e); positions
// >= n are padding), ties cut lowest-position first, or cut == SELECT_UNRESOLVED (see above).  Works from the
// raw values: the only arrays are x and the lane-sorted high words.  warm_hi: in = high word of a nearby threshold (0 = none), out = this one's.
template <typename IdxFn>
__device__ inline SelectResult wave_select16(const double (&x)[16], IdxFn idx_of, int n, int k, unsigned &warm_hi)
{
    unsigned s[16];
    unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const bool valid = idx_of(e) < n;
        const unsigned h = key_hi(x[e]);
        s[e] = valid ? h : 0xffffffffu;
        mn = min(mn, s[e]);
        mx = max(mx, valid ? h : 0u);
    }
    sort16_u32(s);
    mn = wave_umin(mn);
    mx = wave_umax(mx);
    // invariant: count(hi < L) = cL <= k-1 < cU = count(hi < U), the k-th smallest has hi in [L, U)
    uint64_t L = mn, U = (uint64_t)mx + 1;
    int cL = 0, cU = n;
    auto probe = [&](const unsigned cand) {
        const int c = wave_sum(count_below_sorted16(s, cand));
        if (c <= k - 1) { L = cand; cL = c; } else { U = cand; cU = c; }
        return c <= k - 1;
    };
    if (warm_hi > mn && warm_hi <= mx) {
        // gallop away from the neighbouring threshold until the bracket closes on the other side
        const bool right = probe(warm_hi);
        uint64_t step = 1u << 11;
        for (int g = 0; g < 24 && cU - cL > 1; g++) {
            const uint64_t c64 = right ? (uint64_t)warm_hi + step : ((uint64_t)warm_hi > step ? (uint64_t)warm_hi - step : 0);
            if (c64 <= L || c64 >= U) break;              // ran into the other end of the bracket
            if (probe((unsigned)c64) != right) break;     // overshot: the answer is inside [L, U)
            step <<= 1;
        }
    }
    while (cU - cL > 1 && U - L > 1) probe((unsigned)(L + ((U - L) >> 1)));
    if (cU - cL == 1) {
        // the single element with hi in [L, U) is the k-th smallest
        unsigned mh = 0, ml = 0;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const unsigned h = key_hi(x[e]);
            const bool in = ((uint64_t)h >= L) & ((uint64_t)h < U) & (idx_of(e) < n);
            mh = max(mh, in ? h : 0u);
            ml = max(ml, in ? key_lo(x[e]) : 0u);
        }
        SelectResult res;
        res.cut = 0x7fffffff;
        mh = wave_umax(mh);
        res.thr_key = ((uint64_t)mh << 32) | wave_umax(ml);
        warm_hi = mh;
        return res;
    }
    // several keys share the high word L (or exact ties): left to the fix-up pass (~0.1 % of rows), so that
    // the general routine's key arrays do not count against this kernel's registers
    SelectResult res;
    res.thr_key = 0;
    res.cut = SELECT_UNRESOLVED;
    warm_hi = (unsigned)L;
    return res;
}

// ---------------------------------------------------------------------------------------------
// Histogram form of the 16-per-lane selection (the product path).
//
// The high words of the keys are binned linearly between the row's minimum and maximum into
// HIST_BINS wave-private LDS counters (one ds_add per element), a two-level scan of the counters
// finds the bin that holds the k-th smallest and how many elements lie below it, and only the
// handful of elements in that bin are ranked against each other by value.  Against the sorted /
// probing form above this replaces the 63-comparator network and the ~8 counting probes by ~4
// VALU instructions per element.  Anything unusual (two candidates in one lane, equal candidate
// values) is left to the fix-up pass, exactly like wave_select16.
//
// hist: HIST_WORDS counters private to the wave, zero on entry and zero again on return (the 64
// trailing words absorb the padding slots and are never read).
// ---------------------------------------------------------------------------------------------
constexpr int HIST_BPL = 16;
constexpr int HIST_BINS = 64 * HIST_BPL;
constexpr int HIST_LOG2 = 10;
constexpr int HIST_WORDS = HIST_BINS + 64;
static_assert((1 << HIST_LOG2) == HIST_BINS, "HIST_LOG2");

__device__ inline void hist_clear(unsigned *hist, int lane)
{
#pragma unroll
    for (int t = 0; t < HIST_BPL / 4; t++)
        reinterpret_cast<uint4 *>(hist + HIST_BPL * lane)[t] = make_uint4(0, 0, 0, 0);
    hist[HIST_BINS + lane] = 0;
}

// Window state carried from one row to the next by the wave that walks them: the high word of the last
// threshold and the bin width (log2) to use around it.  hi == 0: no prediction, bin the whole range.
struct HistWarm {
    unsigned hi;
    int shift;
};
constexpr int HIST_WARM_SHIFT0 = 8, HIST_WARM_SHIFT_MAX = 14;

template <typename IdxFn>
__device__ inline SelectResult wave_select16_hist(const double (&x)[16], IdxFn idx_of, int n, int k, unsigned *hist,
                                                  int lane, HistWarm &warm)
{
    SelectResult res;
    res.thr_key = 0;
    res.cut = SELECT_UNRESOLVED;
    unsigned bin[16];
    // pass kinds: a window predicted from the neighbouring row, the full range, or the refinement of one bin
    // of the previous pass (when two of its elements sit in the same lane)
    enum { PREDICTED, FULL, REFINE };
    int kind = warm.hi != 0 ? PREDICTED : FULL;
    unsigned lo = 0;
    int shift = warm.shift;
    if (kind == PREDICTED) {
        const unsigned half = (unsigned)(HIST_BINS / 2) << shift;
        lo = max(warm.hi, half) - half;
    }
    int r = 0, cstar = 0;
    double cv = 0.0;
    uint64_t any = 0;
    for (;;) {
        // bins [lo, lo + (HIST_BINS << shift)) of the high words; everything else (and the padding slots)
        // goes to the lane's spill word; `below` = number of elements under lo
        int below = 0;
        if (kind == FULL) {
            unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const unsigned h = key_hi(x[e]);   // padding slots repeat real elements of the row: harmless here
                mn = min(mn, h);
                mx = max(mx, h);
            }
            mn = wave_umin(mn);
            mx = wave_umax(mx);
            lo = mn;
            shift = max(0, 32 - (int)__clz(mx - mn) - HIST_LOG2);
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++)
                below += __popcll(__ballot((int)(key_hi(x[e]) < lo) & (int)(idx_of(e) < n)));
        }
        const unsigned spill = (unsigned)(HIST_BINS + lane);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            unsigned b = min((key_hi(x[e]) - lo) >> shift, spill);
            asm("" : "+v"(b));      // opaque: hipcc 7.2 crashes in instruction selection on the folded LDS address
            bin[e] = idx_of(e) < n ? b : spill;
            atomicAdd(&hist[bin[e]], 1u);
        }
        const int kk = k - below;           // rank among the elements >= lo
        // level 1: HIST_BPL counters per lane
        uint4 c4[HIST_BPL / 4];
        int tot = 0;
#pragma unroll
        for (int t = 0; t < HIST_BPL / 4; t++) {
            c4[t] = reinterpret_cast<const uint4 *>(hist + HIST_BPL * lane)[t];
            tot += (int)(c4[t].x + c4[t].y + c4[t].z + c4[t].w);
        }
        const int incl = wave_scan<OpAdd>(tot, 0);
        const uint64_t m1 = __ballot((incl - tot < kk) & (kk <= incl));
        if (m1 == 0) {
            hist_clear(hist, lane);
            if (kind != PREDICTED) return res;      // k outside 1..n: cannot happen
            // the k-th smallest lies outside the predicted window: widen it for the rows to come and bin the
            // full range now
            warm.shift = min(warm.shift + 1, HIST_WARM_SHIFT_MAX);
            kind = FULL;
            continue;
        }
        const int ls = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)m1) - 1);
        const int r0 = kk - (__builtin_amdgcn_readlane(incl, ls) - __builtin_amdgcn_readlane(tot, ls));
        // level 2: the HIST_BPL counters of lane ls, one per lane
        const int c2 = lane < HIST_BPL ? (int)hist[HIST_BPL * ls + lane] : 0;
        const int inc2 = wave_scan<OpAdd>(c2, 0);
        hist_clear(hist, lane);
        const uint64_t m2 = __ballot((lane < HIST_BPL) & (inc2 >= r0));
        const int ts = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)m2) - 1);
        cstar = __builtin_amdgcn_readlane(c2, ts);
        r = r0 - (__builtin_amdgcn_readlane(inc2, ts) - cstar);      // 1-based rank inside the bin
        const unsigned bstar = (unsigned)(HIST_BPL * ls + ts);
        // candidates: the elements of bin bstar, one per lane
        uint64_t dup = 0;
        any = 0;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const bool in = bin[e] == bstar;
            const uint64_t m = __ballot(in);
            dup |= any & m;
            any |= m;
            cv = in ? x[e] : cv;
        }
        if (dup == 0) break;
        if (shift == 0) return res;         // same high word in one lane: fix-up pass
        lo += bstar << shift;
        shift = max(shift - HIST_LOG2, 0);
        kind = REFINE;
    }
    const bool mine = (any >> lane) & 1;
    int less = 0, equal = 0;
    if (cstar > 1) {
        for (uint64_t rest = any; rest != 0; rest &= rest - 1) {
            const int c = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)rest) - 1);
            const double vc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cv), c),
                                               __builtin_amdgcn_readlane(__double2loint(cv), c));
            less += vc < cv;
            equal += vc == cv;
        }
        if (__ballot(mine & (equal > 1)) != 0) return res;      // equal values: position order, fix-up pass
    }
    const uint64_t win = __ballot(mine & (less == r - 1));
    if (win == 0) return res;
    const int wl = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)win) - 1);
    const double tv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cv), wl),
                                       __builtin_amdgcn_readlane(__double2loint(cv), wl));
    res.thr_key = f64_key(tv);
    res.cut = 0x7fffffff;
    warm.hi = key_hi(tv);
    return res;
}

// 64 x 64 bit transpose across a wave: lane r holds row r (bit c = column c) -> lane c holds column c (bit r = row r).
// Six butterfly steps (block sizes 32 .. 1): lanes r and r ^ j exchange the off-diagonal j x j blocks of every 2j x 2j
// block; ~70 instructions instead of 64 ballots.  (One template instance per step: written as a loop over j, hipcc left it
// a run-time loop with a switch for the masks wherever the function was inlined more than once.)
template <int J>
__device__ inline uint64_t wave_transpose64_step(uint64_t x, int lane)
{
    // columns c with (c & J) == 0
    constexpr uint64_t m = J == 32 ? 0x00000000ffffffffull : J == 16 ? 0x0000ffff0000ffffull : J == 8 ? 0x00ff00ff00ff00ffull
                         : J == 4 ? 0x0f0f0f0f0f0f0f0full : J == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
    const unsigned ylo = (unsigned)__shfl_xor((int)(unsigned)x, J);
    const unsigned yhi = (unsigned)__shfl_xor((int)(unsigned)(x >> 32), J);
    const uint64_t y = ((uint64_t)yhi << 32) | ylo;
    return (lane & J) ? ((x & ~m) | ((y & ~m) >> J)) : ((x & m) | ((y & m) << J));
}

__device__ inline uint64_t wave_transpose64(uint64_t x, int lane)
{
    x = wave_transpose64_step<32>(x, lane);
    x = wave_transpose64_step<16>(x, lane);
    x = wave_transpose64_step<8>(x, lane);
    x = wave_transpose64_step<4>(x, lane);
    x = wave_transpose64_step<2>(x, lane);
    x = wave_transpose64_step<1>(x, lane);
    return x;
}

}  // namespace acoss
