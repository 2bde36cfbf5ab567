// planar_select.h -- device helpers of the kNN selection on uint32 keys (planar_kernels.hip, band_kernels.hip):
// ballot bit planes, the histogram selection, the error band of float32-approximate keys and the exact float64
// refinement of rows / columns the selection could not resolve.
#pragma once

#include "common.h"
#include "kernel_utils.h"
#include "thresh_work.h"
#include "wave_ops.h"

namespace acoss {

// ---- shared with crp_kernels.hip (same definitions; both are internal) -------------------------------------
__device__ inline void planar_put_lane_u64(unsigned &lo, unsigned &hi, uint64_t m, int e)
{
    // s_nop: VALU-writes-SGPR -> VALU-reads-SGPR wait states the hazard recogniser does not insert around asm
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(lo), "+v"(hi)
        : "s"((unsigned)m), "s"((unsigned)(m >> 32)), "n"(e));
}

// four masks into lanes e .. e+3 (one wait state for the four VALU-written SGPR pairs instead of one each)
__device__ inline void planar_put_lanes4_u64(unsigned &lo, unsigned &hi, uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3, int e)
{
    asm("s_nop 1\n\t"
        "v_writelane_b32 %0, %2, %10\n\tv_writelane_b32 %1, %3, %10\n\t"
        "v_writelane_b32 %0, %4, %11\n\tv_writelane_b32 %1, %5, %11\n\t"
        "v_writelane_b32 %0, %6, %12\n\tv_writelane_b32 %1, %7, %12\n\t"
        "v_writelane_b32 %0, %8, %13\n\tv_writelane_b32 %1, %9, %13"
        : "+v"(lo), "+v"(hi)
        : "s"((unsigned)m0), "s"((unsigned)(m0 >> 32)), "s"((unsigned)m1), "s"((unsigned)(m1 >> 32)),
          "s"((unsigned)m2), "s"((unsigned)(m2 >> 32)), "s"((unsigned)m3), "s"((unsigned)(m3 >> 32)),
          "n"(e), "n"(e + 1), "n"(e + 2), "n"(e + 3));
}

// lane e (< E) = lane mask of "position e*64 + lane exists" (computed in the lane: no ballots)
template <int E>
__device__ inline uint64_t planar_slot_valid(int n, int lane)
{
    const int rem = n - 64 * lane;          // positions of word `lane` that exist
    return rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
}

// Padding convention of the 16-keys-per-lane selection (wave_select_hist256_u32): positions >= n hold PLANAR_PAD, a key
// no finite value maps to (the NaN pattern) and that therefore sorts above every real key -- it is never below a
// window, never inside an error band, never selected, and lands in a bin of its own above the real keys.  The counting
// loops then need no validity mask.  (The 32-keys-per-lane selection repeats a real element there and masks instead.)
constexpr unsigned PLANAR_PAD = 0xffffffffu;

// h[e] = PLANAR_PAD where position e*64 + lane >= n.  last_only: the caller knows that only slot E-1 can run past n.
template <int E>
__device__ inline void planar_pad_keys(unsigned (&h)[E], int n, int lane, bool last_only)
{
    if (last_only) {
        h[E - 1] = (E - 1) * 64 + lane < n ? h[E - 1] : PLANAR_PAD;
        return;
    }
#pragma unroll
    for (int e = 0; e < E; e++) h[e] = e * 64 + lane < n ? h[e] : PLANAR_PAD;
}

// word e of out = lanes whose element e is selected (high word <= the threshold's), positions e*64 + lane
template <int E>
__device__ inline void planar_emit_bits(const unsigned (&h)[E], unsigned thr_hi, uint64_t valid, uint64_t *out, int lane,
                                        int64_t stride = 1)
{
    unsigned lo = 0, hi = 0;
    static_assert(E % 4 == 0, "ballots are moved into their lanes four at a time");
#pragma unroll
    for (int e = 0; e < E; e += 4) {
        const uint64_t m0 = __ballot(h[e] <= thr_hi), m1 = __ballot(h[e + 1] <= thr_hi);
        const uint64_t m2 = __ballot(h[e + 2] <= thr_hi), m3 = __ballot(h[e + 3] <= thr_hi);
        planar_put_lanes4_u64(lo, hi, m0, m1, m2, m3, e);
    }
    if (lane < E) out[lane * stride] = (((uint64_t)hi << 32) | lo) & valid;
}

// k-th smallest of the n high words a wave holds (h[e] = position e*64 + lane; positions >= n repeat a real
// element and are ignored), by the histogram method of wave_select16_hist (wave_ops.h).  Returns the winning
// high word with cut = INT_MAX when no other element shares it, else cut = SELECT_UNRESOLVED.
// E = 16 (<= 1024 elements) or 32 (<= 2048; the bin numbers are recomputed instead of held in registers).
template <int E>
__device__ inline SelectResult wave_select_hist_u32(const unsigned (&h)[E], int n, int k, unsigned *hist, int lane,
                                                    HistWarm &warm)
{
    constexpr bool KEEP = E <= 16;
    SelectResult res;
    res.thr_key = 0;
    res.cut = SELECT_UNRESOLVED;
    unsigned bin[KEEP ? E : 1];
    enum { PREDICTED, FULL, REFINE };
    int kind = warm.hi != 0 ? PREDICTED : FULL;
    unsigned lo = 0;
    int shift = warm.shift;
    if (kind == PREDICTED) {
        const unsigned half = (unsigned)(HIST_BINS / 2) << shift;
        lo = max(warm.hi, half) - half;
    }
    int r = 0, cstar = 0;
    unsigned ch = 0;
    uint64_t any = 0;
    for (;;) {
        int below = 0;
        if (kind == FULL) {
            unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
            for (int e = 0; e < E; e++) {
                mn = min(mn, h[e]);
                mx = max(mx, h[e]);
            }
            mn = wave_umin(mn);
            mx = wave_umax(mx);
            lo = mn;
            shift = max(0, 32 - (int)__clz(mx - mn) - HIST_LOG2);
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) below += __popcll(__ballot((h[e] < lo) & (e * 64 + lane < n)));
        }
        const unsigned spill = (unsigned)(HIST_BINS + lane);
#pragma unroll
        for (int e = 0; e < E; e++) {
            unsigned b = min((h[e] - lo) >> shift, spill);
            asm("" : "+v"(b));      // opaque: hipcc 7.2 crashes in instruction selection on the folded LDS address
            b = e * 64 + lane < n ? b : spill;
            if constexpr (KEEP) bin[e] = b;
            atomicAdd(&hist[b], 1u);
        }
        const int kk = k - below;
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
            if (kind != PREDICTED) return res;
            warm.shift = min(warm.shift + 1, HIST_WARM_SHIFT_MAX + 3);
            kind = FULL;
            continue;
        }
        const int ls = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)m1) - 1);
        const int r0 = kk - (__builtin_amdgcn_readlane(incl, ls) - __builtin_amdgcn_readlane(tot, ls));
        const int c2 = lane < HIST_BPL ? (int)hist[HIST_BPL * ls + lane] : 0;
        const int inc2 = wave_scan<OpAdd>(c2, 0);
        hist_clear(hist, lane);
        const uint64_t m2 = __ballot((lane < HIST_BPL) & (inc2 >= r0));
        const int ts = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)m2) - 1);
        cstar = __builtin_amdgcn_readlane(c2, ts);
        r = r0 - (__builtin_amdgcn_readlane(inc2, ts) - cstar);
        const unsigned bstar = (unsigned)(HIST_BPL * ls + ts);
        uint64_t dup = 0;
        any = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            bool in;
            if constexpr (KEEP) in = bin[e] == bstar;
            else in = (((h[e] - lo) >> shift) == bstar) & (h[e] >= lo) & (e * 64 + lane < n);
            const uint64_t m = __ballot(in);
            dup |= any & m;
            any |= m;
            ch = in ? h[e] : ch;
        }
        if (dup == 0) break;
        if (shift == 0) {                   // equal high words in one lane: exact values needed, fix-up pass
            res.thr_key = (uint64_t)(lo + bstar) << 32;       // (tells it which high word the ties share)
            return res;
        }
        lo += bstar << shift;
        shift = max(shift - HIST_LOG2, 0);
        kind = REFINE;
    }
    const bool mine = (any >> lane) & 1;
    int less = 0, equal = 1;
    if (cstar > 1) {
        equal = 0;
        for (uint64_t rest = any; rest != 0; rest &= rest - 1) {
            const int c = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)rest) - 1);
            const unsigned vc = (unsigned)__builtin_amdgcn_readlane((int)ch, c);
            less += vc < ch;
            equal += vc == ch;
        }
    }
    // the winner: `less` smaller candidates, and with ties less < r <= less + equal
    const uint64_t win = __ballot(mine & (less < r) & (r <= less + equal));
    if (win == 0) return res;
    const int wl = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)win) - 1);
    const unsigned th = (unsigned)__builtin_amdgcn_readlane((int)ch, wl);
    if (__builtin_amdgcn_readlane(equal, wl) > 1) {                 // shared high word: exact values decide, fix-up pass
        res.thr_key = (uint64_t)th << 32;
        return res;
    }
    res.thr_key = ((uint64_t)th << 32) | 0xffffffffull;
    res.cut = 0x7fffffff;
    warm.hi = th;
    return res;
}

// The same selection with a single-level histogram of 256 bins per wave (16 keys per lane): lane l owns bins 4l .. 4l+3,
// so the counters are read and cleared with ONE conflict-free 16-byte access per lane -- wave_select_hist_u32 reads
// 4 x 16 bytes at a 64-byte lane stride (4-way bank conflicts on the reads and on the clears) and needs a second, dependent
// LDS read for its second level.  Bins are 4x wider for the same window; the 1-3 keys of the winning bin are ranked by
// v_readlane as before.  hist: 256 + 64 words, zero on entry and on return.  warm.shift is in this function's own bins
// (HIST256_SHIFT0 for float64 high words, + 3 for float32 keys).  h: positions >= n hold PLANAR_PAD (planar_pad_keys).
constexpr int HIST256_LOG2 = 8, HIST256_BINS = 1 << HIST256_LOG2;
constexpr int HIST256_SHIFT0 = HIST_WARM_SHIFT0 + 2, HIST256_SHIFT_MAX = HIST_WARM_SHIFT_MAX + 5;

__device__ inline void hist256_clear(unsigned *hist, int lane)
{
    reinterpret_cast<uint4 *>(hist)[lane] = make_uint4(0, 0, 0, 0);
    hist[HIST256_BINS + lane] = 0;
}

__device__ inline void band_limits(unsigned th, const float *pair_band, unsigned &lo, unsigned &hi);
template <int E>
__device__ inline bool band_crowded(const unsigned (&h)[E], unsigned lo, unsigned hi);

// unsigned saturating subtraction (v_sub_u32 with the clamp bit): max(a, b) - b in one instruction; b wave-uniform
__device__ inline unsigned sat_sub_u32(unsigned a, unsigned b_uniform)
{
    return __builtin_elementwise_sub_sat(a, b_uniform);
}

// Instruction budget: at eight waves per SIMD these kernels are bound by instruction issue, not by latency (the
// selection alone on made-up keys takes as long as the loads alone, tools/planar_probe.py), so every per-key
// instruction counts sixteen times per row.  Per key there are: the bin (saturating subtract, shift, min), the LDS
// add, and compare + select for the candidates.  There is no count of the keys below the window: bin 0 takes them (the
// subtraction saturates), so ranks stay global in every pass; and the error band of approximate keys (pair_band != null:
// band_limits) is checked against the one to three candidates of the winning bin whenever it lies inside that bin --
// against all keys only when it reaches over the bin's edge.
__device__ inline SelectResult wave_select_hist256_u32(const unsigned (&h)[16], int n, int k, unsigned *hist, int lane, HistWarm &warm,
                                                       const float *pair_band = nullptr)
{
    constexpr int E = 16;
    SelectResult res;
    res.thr_key = 0;
    res.cut = SELECT_UNRESOLVED;
    unsigned bin[E];
    enum { PREDICTED, FULL, REFINE };
    int kind = warm.hi != 0 ? PREDICTED : FULL;
    unsigned lo0 = 0;                       // bin b >= 1 holds the keys lo0 + (b << shift) ... ; bin 0 everything below bin 1
    int shift = warm.shift;
    if (kind == PREDICTED) {
        const unsigned back = (unsigned)(HIST256_BINS / 2 + 1) << shift;      // the predicted key sits in the middle of bins 1 .. 255
        lo0 = max(warm.hi, back) - back;
    }
    int r = 0, cstar = 0;
    unsigned ch = 0, bstar = 0;
    uint64_t any = 0;
    for (;;) {
        if (kind == FULL) {
            unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
            for (int e = 0; e < E; e++) {
                // (the key goes through an opaque volatile asm: otherwise the per-lane minimum and maximum -- loop-invariant
                // and free of side effects -- are hoisted in front of the pass loop and computed for EVERY row, 56
                // instructions that only the first row of a wave and the rows with a failed prediction need)
                unsigned v = h[e];
                asm volatile("" : "+v"(v));
                mn = min(mn, v);
                mx = max(mx, v == PLANAR_PAD ? 0u : v);
            }
            mn = wave_umin(mn);
            mx = wave_umax(mx);
            lo0 = mn;                       // nothing lies below: bin 0 is an ordinary bin in this pass
            shift = max(0, 32 - (int)__clz(mx - mn) - HIST256_LOG2);
        }
        lo0 = (unsigned)__builtin_amdgcn_readfirstlane((int)lo0);
        const unsigned spill = (unsigned)(HIST256_BINS + lane);
#pragma unroll
        for (int e = 0; e < E; e++) {
            const unsigned b = min(sat_sub_u32(h[e], lo0) >> shift, spill);      // above the window or padding: a spill word
            bin[e] = b;
            atomicAdd(&hist[b], 1u);
        }
        const uint4 c4 = reinterpret_cast<const uint4 *>(hist)[lane];
        const int tot = (int)(c4.x + c4.y + c4.z + c4.w);
        const int incl = wave_scan<OpAdd>(tot, 0);
        hist256_clear(hist, lane);
        const uint64_t m1 = __ballot((incl - tot < k) & (k <= incl));
        int ls = 0, ts = 0;
        if (m1 != 0) {
            ls = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)m1) - 1);
            r = k - (__builtin_amdgcn_readlane(incl, ls) - __builtin_amdgcn_readlane(tot, ls));
            const int c0 = __builtin_amdgcn_readlane((int)c4.x, ls), c1 = __builtin_amdgcn_readlane((int)c4.y, ls);
            const int c2 = __builtin_amdgcn_readlane((int)c4.z, ls), c3 = __builtin_amdgcn_readlane((int)c4.w, ls);
            cstar = c0;
            if (r > c0) {
                r -= c0; ts = 1; cstar = c1;
                if (r > c1) {
                    r -= c1; ts = 2; cstar = c2;
                    if (r > c2) { r -= c2; ts = 3; cstar = c3; }
                }
            }
        }
        bstar = (unsigned)(4 * ls + ts);
        if (m1 == 0 || (kind == PREDICTED && bstar == 0)) {
            if (kind != PREDICTED) return res;      // k outside 1..n: cannot happen
            // the k-th smallest lies outside the predicted window: widen it for the rows to come, bin the full range now
            warm.shift = min(warm.shift + 1, HIST256_SHIFT_MAX);
            kind = FULL;
            continue;
        }
        // candidates: the keys of bin bstar, one per lane (keys are never 0: their sign bit is set)
        ch = 0;
#pragma unroll
        for (int e = 0; e < E; e++) ch = bin[e] == bstar ? h[e] : ch;
        any = __ballot(ch != 0);
        if (__popcll(any) == cstar) break;          // no lane holds two of them
        if (shift == 0) {                   // equal high words in one lane: exact values needed, fix-up pass
            res.thr_key = (uint64_t)(lo0 + bstar) << 32;       // (tells it which high word the ties share)
            return res;
        }
        // bin bstar again, 128 times finer (bins 1 .. 128; bin 0 keeps everything below it, so k stays the rank)
        const int fine = max(shift - (HIST256_LOG2 - 1), 0);
        lo0 = lo0 + (bstar << shift) - (1u << fine);
        shift = fine;
        kind = REFINE;
    }
    const bool mine = (any >> lane) & 1;
    int less = 0, equal = 1;
    if (cstar > 1) {
        equal = 0;
        for (uint64_t rest = any; rest != 0; rest &= rest - 1) {
            const int c = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)rest) - 1);
            const unsigned vc = (unsigned)__builtin_amdgcn_readlane((int)ch, c);
            less += vc < ch;
            equal += vc == ch;
        }
    }
    const uint64_t win = __ballot(mine & (less < r) & (r <= less + equal));
    if (win == 0) return res;
    const int wl = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)win) - 1);
    const unsigned th = (unsigned)__builtin_amdgcn_readlane((int)ch, wl);
    res.thr_key = (uint64_t)th << 32;
    if (__builtin_amdgcn_readlane(equal, wl) > 1) return res;       // shared high word: exact values decide, fix-up pass
    if (pair_band != nullptr) {
        // approximate keys: the result stands only if the winner is alone in its error band
        unsigned blo, bhi;
        band_limits(th, pair_band, blo, bhi);
        const unsigned bin_lo = lo0 + (bstar << shift), bin_hi = bin_lo + ((1u << shift) - 1u);
        bool crowded;
        if (blo >= bin_lo && bhi <= bin_hi && !(kind != FULL && bstar == 0))
            crowded = cstar > 1 && __popcll(__ballot(mine & ((ch - blo) <= (bhi - blo)))) > 1;
        else
            crowded = band_crowded<E>(h, blo, bhi);
        if (crowded) return res;
    }
    res.thr_key |= 0xffffffffull;
    res.cut = 0x7fffffff;
    warm.hi = th;
    return res;
}

// ---- float32-approximate keys: the error band around a threshold ----------------------------------------------
// Keys are float32 bit patterns (values >= +0) with the sign bit set.  [lo, hi] = the keys of the values within `band`
// of the value of key th, widened by one ulp each way for the rounding of the two float operations.
// band = base + slope * value: twice the error bound of a value of that size (the caller's two floats per pair).
__device__ inline void band_limits(unsigned th, const float *pair_band, unsigned &lo, unsigned &hi)
{
    const float a = __uint_as_float(th & 0x7fffffffu);
    const float band = fmaf(pair_band[1], a, pair_band[0]);
    const float l = a - band, h = a + band;
    lo = l > 0.0f ? (__float_as_uint(l) | 0x80000000u) - 1u : 0x80000000u;
    hi = (__float_as_uint(h) | 0x80000000u) + 1u;
}

// number of the wave's keys inside [lo, hi] (wave-uniform).  Padding slots (positions >= n repeat a real element) are
// counted too: that can only turn a resolved row into one for the fix-up pass, never the other way round.
template <int E>
__device__ inline int band_count(const unsigned (&h)[E], unsigned lo, unsigned hi)
{
    const unsigned width = hi - lo;
    int c = 0;
#pragma unroll
    for (int e = 0; e < E; e++) c += __popcll(__ballot((h[e] - lo) <= width));
    return c;
}

// more than one of the wave's keys inside [lo, hi]? (wave-uniform; counted per lane, two ballots at the end)
template <int E>
__device__ inline bool band_crowded(const unsigned (&h)[E], unsigned lo, unsigned hi)
{
    const unsigned width = hi - lo;
    int c = 0;
#pragma unroll
    for (int e = 0; e < E; e++) c += (h[e] - lo) <= width;
    const uint64_t some = __ballot(c > 0);
    return __ballot(c > 1) != 0 || (some & (some - 1)) != 0;
}

// after the selection on approximate keys: the result stands only if the winner is alone in its error band
template <int E>
__device__ inline void band_resolve(const unsigned (&h)[E], int n, const float *band, int p, int lane, SelectResult &res)
{
    if (band == nullptr || res.cut == SELECT_UNRESOLVED || res.cut < 0 || res.thr_key == ~0ull) return;
    const unsigned th = (unsigned)(res.thr_key >> 32);
    unsigned lo, hi;
    band_limits(th, band + 2 * p, lo, hi);
    if (band_crowded<E>(h, lo, hi)) {
        res.thr_key = (uint64_t)th << 32;
        res.cut = SELECT_UNRESOLVED;
    }
}

// word index of element idx of the float64 layout
__device__ inline int64_t planar_word(int64_t idx) { return idx; }

__device__ inline bool planar_trivial(int k, int n, SelectResult &r)
{
    if (k <= 0) { r.thr_key = 0ull; r.cut = -1; return true; }
    if (k >= n) { r.thr_key = ~0ull; r.cut = 0x7fffffff; return true; }
    return false;
}

__device__ inline int knn_count(int k_mode, double kv, int len)
{
    // CRPUtils.py:190-193; half-even rounding (np.round) = rint under the default rounding mode
    return k_mode == 0 ? (int)rint(kv * (double)len) : (k_mode == 1 ? (int)kv : len);
}

// ---- fix-up: rows / columns whose winner shares its high word -----------------------------------------------
// One term of a windowed sum as the EXACT path defines it, from the dot product of a frame pair and the two squared norms:
//   float64 features (crema chroma): C = max(fma(-2, dot, |x|^2 + |y|^2), 0) in float64 (crp_strip_kernel's arithmetic);
//   float32 features (the reference's mfcc_htk / hpcp: get_csm follows its inputs' dtype, CRPUtils.py:82, and sliding_csm
//   squares in that dtype before it promotes, :40-41): c in float32, r = sqrtf(max(c, 0)), the term is (double)(r * r) --
//   crp_kernel<float>'s arithmetic, which the float32 filter's values differ from only by the root-square and the float32
//   window sum (round 4: the 16-bit-key path for float32 corpora).
__device__ inline double exact_term(double dot, double nsum) { return fmax(fma(-2.0, dot, nsum), 0.0); }
__device__ inline double exact_term(float dot, float nsum)
{
    const float r = sqrtf(fmaxf(fmaf(-2.0f, dot, nsum), 0.0f));
    return (double)(r * r);
}

// One windowed sum, exactly: dot product as an FMA chain over the bins of the rolled x frame (the matrix-core and VALU forms
// of that chain agree bit for bit: tests/test_gpu_fast_path.py), exact_term(), the window's terms added in order in float64.
template <typename FT>
__device__ inline double planar_exact_value(const FT *__restrict__ feats, const FT *__restrict__ norms, int d,
                                            const acoss_pair_desc &ds, int win, int i, int j)
{
    double s = 0.0;
    for (int k = 0; k < win; k++) {
        const FT *x = feats + (ds.x_row0 + i + k) * d, *y = feats + (ds.y_row0 + j + k) * d;
        FT acc = 0;
        for (int b = 0; b < d; b++) {
            int src = b - ds.shift;
            if (src < 0) src += d;
            acc = fma(x[src], y[b], acc);
        }
        s += exact_term(acc, norms[ds.x_row0 + i + k] + norms[ds.y_row0 + j + k]);
    }
    return s;
}

// One row (DIR 0) or column (DIR 1) the selection kernels left unresolved, the general way: exact values for the
// elements that share the winner's key (or, with approximate keys, lie inside its error band), then the bit-serial
// selection over 64-bit keys.
// Bit planes of the band kernel (band_kernels.hip) are stored shifted up by plane_shift bits: bit l of word e = position
// 64 e + l - plane_shift.  Lane e holds natural word e; returns the shifted word e.
__device__ inline uint64_t shifted_plane_word(uint64_t mine, int plane_shift, int lane)
{
    const unsigned plo = (unsigned)__shfl_up((int)(unsigned)mine, 1), phi = (unsigned)__shfl_up((int)(unsigned)(mine >> 32), 1);
    const uint64_t prev = lane == 0 ? 0ull : (((uint64_t)phi << 32) | plo);
    return (mine << plane_shift) | (prev >> (64 - plane_shift));
}

// key_at(q): the uint32 key of position q of the row / column (its source: the key matrix, or a compact copy of the row).
// thr_hi: the high word the selection left for this row; thr / cut (may be null): where the final threshold goes.
// (have_range: [blo, bhi] given by the caller -- the 16-bit keys of keys16.h; else derived from thr_hi and the pair's band)
template <int DIR, int E, typename KeyAt, typename FT>
__device__ inline void fix_row_generic_impl(KeyAt key_at, unsigned thr_hi, bool have_range, unsigned blo, unsigned bhi, const FT *__restrict__ feats,
                                       const FT *__restrict__ norms, int d, const acoss_pair_desc &ds, int win,
                                       const ThreshWork &w, int p, int which, int len, int k, uint64_t *thr, int *cut, int lane, int plane_shift = 0)
{
    // the selection kernel left the high word the tied elements share: only those few need their exact value;
    // every other element is ordered by its high word alone
    const unsigned th = have_range ? 1u : thr_hi;
    if (!have_range && w.band != nullptr) band_limits(th, w.band + 2 * p, blo, bhi);
    const bool banded = have_range || w.band != nullptr;
    uint64_t key[E];
    int idx[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        idx[e] = e * 64 + lane;
        const int q = min(idx[e], len - 1);
        const unsigned h = key_at(q);
        uint64_t kx = (uint64_t)h << 32;
        bool exact = th == 0u || h == th;
        if (banded) {
            // approximate keys: below the band certainly selected, above it certainly not, inside it exact values decide
            kx = h < blo ? 0ull : ~0ull - 1ull;
            exact = th == 0u || (h >= blo && h <= bhi);
        }
        if (exact)
            kx = f64_key(DIR == 0 ? planar_exact_value(feats, norms, d, ds, win, which, q)
                                  : planar_exact_value(feats, norms, d, ds, win, q, which));
        key[e] = idx[e] < len ? kx : ~0ull;
    }
    const SelectResult res = wave_select_kth<E>(key, idx, len, k);
    if (lane == 0 && thr != nullptr) {
        thr[which] = res.thr_key;
        cut[which] = res.cut;
    }
    uint64_t *bits = DIR == 0 ? w.row_bits : w.col_bits;
    if (bits) {
        bits = DIR == 0 ? w.row_bits + ((int64_t)p * w.max_m + which) * E : w.col_word(p, which, 0);
        const int64_t bstride = DIR == 0 ? 1 : w.max_n;
        uint64_t mine = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const bool on = (idx[e] < len) & ((key[e] < res.thr_key) | ((key[e] == res.thr_key) & (idx[e] <= res.cut)));
            const uint64_t m = __ballot(on);
            if (lane == e) mine = m;
        }
        if (plane_shift > 0) mine = shifted_plane_word(mine, plane_shift, lane);
        if (lane < E) bits[lane * bstride] = mine;
    }
}

template <int DIR, int E, typename KeyAt, typename FT>
__device__ inline void fix_row_generic(KeyAt key_at, unsigned thr_hi, const FT *__restrict__ feats,
                                       const FT *__restrict__ norms, int d, const acoss_pair_desc &ds, int win,
                                       const ThreshWork &w, int p, int which, int len, int k, uint64_t *thr, int *cut, int lane, int plane_shift = 0)
{
    fix_row_generic_impl<DIR, E>(key_at, thr_hi, false, 0u, 0u, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane, plane_shift);
}

template <int DIR, int E, typename KeyAt, typename FT>
__device__ inline void fix_row_generic_range(KeyAt key_at, unsigned blo, unsigned bhi, const FT *__restrict__ feats,
                                             const FT *__restrict__ norms, int d, const acoss_pair_desc &ds, int win,
                                             const ThreshWork &w, int p, int which, int len, int k, uint64_t *thr, int *cut, int lane)
{
    fix_row_generic_impl<DIR, E>(key_at, 1u, true, blo, bhi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane, 0);
}

// The same for approximate keys when few elements (<= 64) lie inside the error band, which is the rule: the wave works
// on the band elements together.  Seven elements at a time, lane (g, kk) forms C[i + kk][j + kk] of element g with the
// strip kernel's arithmetic (FMA chain over the rolled bins, all of a frame pair's loads in flight at once), lane g adds
// the nine values in window order; every band element then counts the band elements that precede it in (exact value,
// position) order, and the first k - (elements below the band) of them are selected.  Returns false if the row has to
// go the general way.
constexpr int FIX_MAXD = 16;
struct FixSmem {
    int pos[64];
    unsigned long long key[64];
    double cval[7 * 9];
    unsigned char sel[64];
};

template <int DIR, int E, typename KeyAt, typename FT>
__device__ inline bool fix_row_band_range(FixSmem &sm, KeyAt key_at, unsigned blo, unsigned bhi, const FT *__restrict__ feats,
                                    const FT *__restrict__ norms, int d, const acoss_pair_desc &ds, int win,
                                    const ThreshWork &w, int p, int which, int len, int k, uint64_t *thr, int *cut, int lane, int plane_shift = 0)
{
    if (win != 9 || d > FIX_MAXD) return false;
    unsigned h[E];
    int below = 0, n = 0;
    int myidx[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int pos = e * 64 + lane;
        const int q = min(pos, len - 1);
        h[e] = key_at(q);
        const bool valid = pos < len;
        below += __popcll(__ballot(valid & (h[e] < blo)));
        const bool in = valid & (h[e] >= blo) & (h[e] <= bhi);
        const unsigned long long m = __ballot(in);
        myidx[e] = in ? n + __popcll(m & ((1ull << lane) - 1ull)) : -1;
        n += __popcll(m);
    }
    const int need = k - below;
    if (n > 64 || need < 1 || need > n) return false;      // (the last two cannot happen with a valid error band)
#pragma unroll
    for (int e = 0; e < E; e++)
        if (myidx[e] >= 0) sm.pos[myidx[e]] = e * 64 + lane;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 7) {
        const int g = lane / 9, kk = lane - 9 * g, el = c0 + g;
        if (lane < 63 && el < n) {
            const int pos = sm.pos[el];
            const int i = DIR == 0 ? which : pos, j = DIR == 0 ? pos : which;
            const FT *x = feats + (ds.x_row0 + i + kk) * d, *y = feats + (ds.y_row0 + j + kk) * d;
            FT xv[FIX_MAXD], yv[FIX_MAXD];
#pragma unroll
            for (int b = 0; b < FIX_MAXD; b++) {
                int src = b - ds.shift;
                if (src < 0) src += d;
                xv[b] = b < d ? x[src] : (FT)0;
                yv[b] = b < d ? y[b] : (FT)0;
            }
            const FT nn = norms[ds.x_row0 + i + kk] + norms[ds.y_row0 + j + kk];
            FT acc = 0;
#pragma unroll
            for (int b = 0; b < FIX_MAXD; b++)
                if (b < d) acc = fma(xv[b], yv[b], acc);
            sm.cval[g * 9 + kk] = exact_term(acc, nn);
        }
        __syncthreads();
        if (lane < 7 && c0 + lane < n) {
            double s_ = 0.0;
#pragma unroll
            for (int q = 0; q < 9; q++) s_ += sm.cval[lane * 9 + q];
            sm.key[c0 + lane] = f64_key(s_);
        }
        __syncthreads();
    }
    // rank of band element `lane` among the band elements: (exact key, position) order
    int rank = 1;
    unsigned long long mykey = 0;
    int mypos = 0;
    if (lane < n) { mykey = sm.key[lane]; mypos = sm.pos[lane]; }
    for (int m = 0; m < n; m++) {
        const unsigned long long km = sm.key[m];
        const int pm = sm.pos[m];
        rank += (km < mykey) | ((km == mykey) & (pm < mypos));
    }
    if (lane < n) sm.sel[lane] = rank <= need;
    const unsigned long long last = __ballot(lane < n && rank == need);      // exactly one lane
    const int ll = __ffsll((long long)last) - 1;
    if (lane == ll && thr != nullptr) {
        thr[which] = mykey;
        cut[which] = 0x7fffffff;        // (ties in exact value inside the band were cut by position in the ranking above)
    }
    __syncthreads();
    uint64_t *bits = DIR == 0 ? w.row_bits : w.col_bits;
    if (bits) {
        bits = DIR == 0 ? w.row_bits + ((int64_t)p * w.max_m + which) * E : w.col_word(p, which, 0);
        const int64_t bstride = DIR == 0 ? 1 : w.max_n;
        uint64_t mine = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const bool valid = e * 64 + lane < len;
            const bool on = valid & ((h[e] < blo) | (myidx[e] >= 0 && sm.sel[myidx[e] & 63]));
            const uint64_t m = __ballot(on);
            if (lane == e) mine = m;
        }
        if (plane_shift > 0) mine = shifted_plane_word(mine, plane_shift, lane);
        if (lane < E) bits[lane * bstride] = mine;
    }
    __syncthreads();
    return true;
}

template <int DIR, int E, typename KeyAt, typename FT>
__device__ inline bool fix_row_band(FixSmem &sm, KeyAt key_at, unsigned thr_hi, const FT *__restrict__ feats,
                                    const FT *__restrict__ norms, int d, const acoss_pair_desc &ds, int win,
                                    const ThreshWork &w, int p, int which, int len, int k, uint64_t *thr, int *cut, int lane, int plane_shift = 0)
{
    if (thr_hi == 0u) return false;
    unsigned blo, bhi;
    band_limits(thr_hi, w.band + 2 * p, blo, bhi);
    return fix_row_band_range<DIR, E>(sm, key_at, blo, bhi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane, plane_shift);
}

}  // namespace acoss
