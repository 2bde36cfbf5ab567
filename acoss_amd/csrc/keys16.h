// keys16.h -- the kNN selection (CRPUtils.py:169-219) on 16-bit keys (round 3).
//
// The float32 strip kernel's windowed sums T~ are only a FILTER (strip32_kernels.hip): rows and columns whose k-th smallest
// has a neighbour inside the float32 error band are finished in float64.  The selection kernels are bound by the bytes they
// read and the instructions they spend per key, so the filter's keys shrink to 16 bits:
//
//     k' = bits(T~) -sat koff_p                                       bits = the float32 bit pattern (T~ >= +0: monotone)
//     key16(T~) = min(max(k' >> 11, (k' >> 9) -sat 49152), 0xFFFE)
//
// koff_p = the pattern of 2 W_p 2^-7 per pair (W_p = the pair's bound on the window norm sums; no windowed sum exceeds 2 W_p).
// Two resolutions, one monotone map: the three octaves below 2 W_p -- where the thresholds of real rows sit (1.5 to 2.9 octaves
// below 2 W_p on every corpus of the benchmarks) -- keep 14 mantissa bits (keys 16384 .. 65534), the four octaves below those
// 12 bits (keys 0 .. 16383).  Values at the top clamp to 0xFFFE, values below the range to 0, 0xFFFF pads positions past the
// end of a row.  The strip kernel (row-band form, OUT = 1)
// writes ONLY this plane: 2 bytes per cell written, 2 read by the row selection, 2 by the column selection (6 bytes per cell
// and pair instead of 12).
//
// What 16 bits cannot decide, in order of cost:
//  1. nothing else within the key range that the float32 error band of the winner can reach (92 % of the rows of the
//     benchmark corpus): the masks follow from key16 <= threshold alone;
//  2. otherwise the few cells in that range (two or three) get their float32 value RECOMPUTED from the features by the
//     selecting wave, one cell per lane, with the strip kernel's arithmetic bit for bit (an FMA chain over the bins, the nine
//     terms added by window_sum9(): tests/test_gpu_fast_path.py pins the matrix-core form against exactly this chain) -- the
//     full 32-bit keys of those cells then decide as the 32-bit selection does;
//  3. if the winner is still not alone in its error band (0.5 %), or the threshold clamps: the row's keys go to the side buffer
//     and select_fix_side16_kernel finishes it in float64 (exact values of every cell in the reachable range).
// Masks are identical to the float64 path's in every case.
#pragma once

#include "planar_select.h"

namespace acoss {

constexpr unsigned K16_FINE = 16384u;                 // first key of the fine region; K16_FINE_BIAS = (4 << 14) - K16_FINE ... 49152
constexpr unsigned K16_FINE_BIAS = 49152u;
constexpr unsigned K16_MAX = 0xFFFEu, K16_PAD = 0xFFFFu;
#ifndef K16_SHIFT0_V
#define K16_SHIFT0_V 4
#endif
constexpr int K16_SHIFT0 = K16_SHIFT0_V, K16_SHIFT_MAX = 10;     // histogram bin widths (log2, in keys): the 32-bit selection's window / 2^9
constexpr int K16_SCRATCH = 256;                      // words of wave-private LDS behind the histogram (slow path): positions,
                                                      // C values, float32 keys, extra mask bits (the last quarter: zero between uses)
constexpr int K16_HIST_WORDS = HIST256_BINS + 64 + K16_SCRATCH;

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ inline u16x2 k16_splat(unsigned v) { return (u16x2){(unsigned short)v, (unsigned short)v}; }
__device__ inline u16x2 k16_from_u32(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ inline unsigned k16_to_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

__device__ inline unsigned key16_of_bits(unsigned fbits, unsigned koff)
{
    const unsigned kp = __builtin_elementwise_sub_sat(fbits, koff);
    return min(max(kp >> 11, __builtin_elementwise_sub_sat(kp >> 9, K16_FINE_BIAS)), K16_MAX);
}

// the float32 bit patterns a key stands for: [first, first + span)
__device__ inline void key16_bits_range(unsigned key, unsigned koff, unsigned &first, unsigned &span)
{
    if (key >= K16_FINE) { first = koff + ((key + K16_FINE_BIAS) << 9); span = 1u << 9; }
    else { first = koff + (key << 11); span = 1u << 11; }
}

// what a lane needs to recompute a cell's float32 windowed sum
#ifdef ACOSS_PROBES
#define K16_STAT(ctr, slot) do { if (lane == 0 && (ctr) != nullptr) atomicAdd((ctr) + 16 * (slot), 1); } while (0)
#else
#define K16_STAT(ctr, slot) do { } while (0)
#endif

// Development (probes build, K16Ctx.flags & 4): where a wave's cycles go, phase by phase -- s_memtime at the phase boundaries (after
// the waits the code has there anyway), summed per wave and added to stats[4 + phase] in units of 64 cycles when the wave ends (one block in 64 reports; stats[11] = the rows
// or columns those waves handled).
struct K16Probe {
#ifdef ACOSS_PROBES
    uint64_t last;
    uint32_t acc[8];
    bool on;
    __device__ inline void start(bool enable)
    {
        on = enable;
        for (int s = 0; s < 8; s++) acc[s] = 0u;
        last = on ? __builtin_readcyclecounter() : 0u;
    }
    __device__ inline void lap(int phase)
    {
        if (!on) return;
        const uint64_t t = __builtin_readcyclecounter();
        acc[phase] += (uint32_t)(t - last);
        last = t;
    }
    __device__ inline void count() { if (on) acc[7]++; }            // one more row / column (phase 7 = their number)
    __device__ inline void flush(int *stats, int lane)
    {
        // (one block in 64 reports: a million atomics on one line would take longer than the kernel)
        if (!on || stats == nullptr || lane != 0 || (blockIdx.x & 63) != 0) return;
        for (int s = 0; s < 7; s++) atomicAdd(stats + 4 + s, (int)(acc[s] >> 6));
        atomicAdd(stats + 4 + 7, (int)acc[7]);
    }
#else
    __device__ inline void start(bool) {}
    __device__ inline void lap(int) {}
    __device__ inline void count() {}
    __device__ inline void flush(int *, int) {}
#endif
};

struct K16Ctx {
    int flags;              // development (probes build): 1 = every row decided by its 16-bit keys (timing only), 4 = K16Probe
    int *stats;             // development counters (probes build): [1] float32 recomputes, [2] full-range passes, [3] finer passes
    const float *xp;        // packed x frames of the batch (pack_x32): [pair][max_nx][16]
    int max_nx;
    const float *f32;       // the corpus' float32 copy and its squared norms
    const float *n32;
    const uint32_t *koff;   // per pair
};

// C[i][j] of T~ exactly as the strip kernels form it (strip32_kernels.hip): dot = FMA chain over the bins of the rolled x
// frame, C = max(fma(-2, dot, |x|^2 + |y|^2), 0); the nine C values of a window are then added by window_sum9()
template <int D>
__device__ inline float k16_c_value(const float *__restrict__ xrow0, const float *__restrict__ f32, const float *__restrict__ n32,
                                    const acoss_pair_desc &ds, int i, int j)
{
    const float4 *x4 = reinterpret_cast<const float4 *>(xrow0 + (int64_t)i * 16);
    const float *y = f32 + (ds.y_row0 + j) * D;
    const float4 a = x4[0], b = x4[1], c = x4[2], e = x4[3];
    float yv[D];
    if constexpr (D % 4 == 0) {
#pragma unroll
        for (int q = 0; q < D / 4; q++) {
            const float4 t = reinterpret_cast<const float4 *>(y)[q];
            yv[4 * q] = t.x; yv[4 * q + 1] = t.y; yv[4 * q + 2] = t.z; yv[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int bin = 0; bin < D; bin++) yv[bin] = y[bin];
    }
    const float yn = n32[ds.y_row0 + j];
    const float xv[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, e.x, e.y, e.z, e.w};
    float acc = 0.0f;
#pragma unroll
    for (int bin = 0; bin < D; bin++) acc = fmaf(xv[bin], yv[bin], acc);
    return fmaxf(fmaf(-2.0f, acc, xv[D] + yn), 0.0f);
}

// [h_lo, h_hi]: the 16-bit keys that can hold a cell within the float32 error band of ANY value whose key is th.
// Keys above K16_FINE belong to values a >= 2 W 2^-3 (koff is the pattern of 2 W 2^-7, the fine region starts four octaves
// above it) and step by at least 2^-15 a: there the band, (2 (d + 4.5) W + 19 a) 2^-24 rounded up <= (8 (d + 4.5) + 19) 2^-24 a
// < 2^-16.7 a for d <= 13, is narrower than a key, so nothing outside [th - 1, th + 1] is in reach: no float arithmetic on the
// common path.
//
// That shortcut rests on the caller's arrays agreeing with each other -- band = (2 (d + 4.5) W, 19) 2^-24 and koff = the pattern
// of the same 2 W 2^-7 -- and acoss_mask_bits_keys16_batch takes them as independent arguments, so it is CHECKED per pair
// (k16_reach_adjacent_ok, once per wave): the band at the bottom of the fine region -- the value of key K16_FINE, four octaves
// above koff's -- and its slope must both stay below one key step, 2^-15 of the value; a pair that fails (a wider band, a koff
// from a smaller W, koff == 0: a pair too quiet for the window to mean anything) takes the general branch for every threshold.
__device__ inline bool k16_reach_adjacent_ok(unsigned koff, const float *pair_band)
{
    if (koff == 0u) return false;
    const float a0 = __uint_as_float(koff + ((K16_FINE + K16_FINE_BIAS) << 9));
    return pair_band[1] < 0x1p-15f && fmaf(pair_band[1], a0, pair_band[0]) < 0x1p-15f * a0;
}

__device__ inline void k16_reach(unsigned th, unsigned koff, bool adjacent_ok, const float *pair_band, unsigned &h_lo, unsigned &h_hi)
{
    if (adjacent_ok && th > K16_FINE && th < K16_MAX) {
        h_lo = th - 1u;
        h_hi = min(th + 1u, K16_MAX);
        return;
    }
    unsigned b_lo, span;
    key16_bits_range(th, koff, b_lo, span);
    const unsigned b_hi = b_lo + span - 1u;
    const float a_lo = __uint_as_float(b_lo), a_hi = __uint_as_float(b_hi);
    const float band = fmaf(pair_band[1], a_hi, pair_band[0]);
    const float l = a_lo - band, h = a_hi + band;
    const unsigned lb = l > 0.0f ? __float_as_uint(l) - 1u : 0u;
    h_lo = th == 0u ? 0u : key16_of_bits(lb, koff);
    h_hi = th >= K16_MAX ? K16_MAX : key16_of_bits(__float_as_uint(h) + 1u, koff);
}

// a packed constant the optimiser cannot see through: min(x, 1) on packed halves stays ONE v_pk_min_u16 (with a visible 1 it
// becomes two compares, two selects and a permute per register), x * 4 + b one v_pk_mad_u16
__device__ inline u16x2 k16_opaque(unsigned packed)
{
    asm("" : "+v"(packed));
    return k16_from_u32(packed);
}
__device__ inline u16x2 k16_opaque_ones() { return k16_opaque(0x00010001u); }

// LDS by byte offset (wave-private histogram words): the offset of a __shared__ word, and += 1 at an offset
typedef __attribute__((address_space(3))) unsigned k16_lds_word;
__device__ inline unsigned k16_lds_offset(unsigned *p) { return (unsigned)(uintptr_t)(k16_lds_word *)p; }
__device__ inline void k16_lds_add1(unsigned byte_offset)
{
    __hip_atomic_fetch_add((k16_lds_word *)(uintptr_t)byte_offset, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct Sel16 {
    unsigned th;        // the k-th smallest key
    int rk;             // the k-th smallest is the rk-th (1-based) among the keys equal to th
    int ceq;            // keys equal to th
    unsigned binlo;     // the histogram bin the search ended in: [binlo, binlo + binw]
    unsigned binw;
    unsigned ch;        // per lane: its key of that bin (valid where `any`)
    uint64_t any;
    int cstar;          // keys in that bin
    bool ok;
};

// ---- the pieces of wave_select_k16 ---------------------------------------------------------------------------------------------
// One histogram pass: bin the wave's keys (bin = (key -sat lo0) >> shift, everything from bin 256 up in a per-lane spill word),
// find the bin of the k-th smallest.  lo0 wave-uniform.  hit = false: the k-th smallest lies above the 256 bins.
struct K16Pass {
    int r;              // the k-th smallest is the r-th of its bin
    int cstar;          // keys in that bin
    unsigned bstar;     // the bin
    bool hit;
};

__device__ inline K16Pass k16_hist_pass(const u16x2 (&h)[8], int k, unsigned *hist, unsigned hist_lds, int lane, unsigned lo0, int shift)
{
    // bins of the 16 keys, stage by stage over the registers (independent neighbours: no wait states between the packed
    // instructions), then their LDS byte addresses in packed form too: hist sits below 64 KB, 4 * bin + base fits 16 bits
    const u16x2 lo_pk = k16_splat(lo0), sh_pk = k16_splat((unsigned)shift), spill_pk = k16_splat((unsigned)(HIST256_BINS + lane));
    const u16x2 base_pk = k16_splat(hist_lds), four_pk = k16_opaque(0x00040004u);      // (opaque: one v_pk_mad_u16, not shift + add)
    u16x2 b[8];
#pragma unroll
    for (int v = 0; v < 8; v++) b[v] = __builtin_elementwise_sub_sat(h[v], lo_pk);
#pragma unroll
    for (int v = 0; v < 8; v++) b[v] = b[v] >> sh_pk;
#pragma unroll
    for (int v = 0; v < 8; v++) b[v] = __builtin_elementwise_min(b[v], spill_pk);
#pragma unroll
    for (int v = 0; v < 8; v++) b[v] = b[v] * four_pk + base_pk;
#pragma unroll
    for (int v = 0; v < 8; v++) {
        const unsigned a2 = k16_to_u32(b[v]);
        k16_lds_add1(a2 & 0xFFFFu);
        k16_lds_add1(a2 >> 16);
    }
    const uint4 c4 = reinterpret_cast<const uint4 *>(hist)[lane];
    const int tot = (int)(c4.x + c4.y + c4.z + c4.w);
    const int incl = wave_scan<OpAdd>(tot, 0);
    hist256_clear(hist, lane);
    // The lane whose four bins hold the k-th smallest decodes it by itself, one word for the whole wave to read: with
    // q_t = k - 1 - (keys before its bin t) the bin is the LAST t with q_t >= 0, so the unsigned minimum over t of
    // (q_t << 13 | (3 - t) << 11 | c_t) -- negative q_t are huge, equal q_t (an empty bin in front) fall to the later bin --
    // carries the rank inside the bin (q_t + 1), the bin and its count (<= 1024: 11 bits).
    const int q0 = k - 1 - (incl - tot), q1 = q0 - (int)c4.x, q2 = q1 - (int)c4.y, q3 = q2 - (int)c4.z;
    const unsigned e0 = ((unsigned)q0 << 13) + (c4.x + (3u << 11)), e1 = ((unsigned)q1 << 13) + (c4.y + (2u << 11));
    const unsigned e2 = ((unsigned)q2 << 13) + (c4.z + (1u << 11)), e3 = ((unsigned)q3 << 13) + c4.w;
    const unsigned code = min(min(e0, e1), min(e2, e3));
    const uint64_t m1 = __ballot((unsigned)q0 < (unsigned)tot);
    K16Pass ps;
    ps.hit = m1 != 0;
    ps.r = 0; ps.cstar = 0; ps.bstar = 0;
    if (ps.hit) {
        const int ls = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)m1) - 1);
        const unsigned cd = (unsigned)__builtin_amdgcn_readlane((int)code, ls);
        ps.r = (int)(cd >> 13) + 1;
        ps.cstar = (int)(cd & 0x7FFu);
        ps.bstar = (unsigned)(4 * ls + 3) - ((cd >> 11) & 3u);
    }
    return ps;
}

// the keys of [binlo, binlo + binw], one per lane: the smallest offset from binlo.  (Offsets are taken modulo 2^16: a key below
// binlo wraps to 2^16 - (binlo - key), which a bin that reaches past the top of the key range could mistake for one of its own;
// the limit stops at key 0xFFFF.)
__device__ inline uint64_t k16_bin_keys(const u16x2 (&h)[8], unsigned binlo, unsigned binw, unsigned &ch)
{
    const u16x2 bl_pk = k16_splat(binlo);
    u16x2 mo = k16_splat(0xFFFFu);
#pragma unroll
    for (int v = 0; v < 8; v++) mo = __builtin_elementwise_min(mo, h[v] - bl_pk);
    const unsigned off = min((unsigned)mo.x, (unsigned)mo.y);
    ch = binlo + off;
#ifdef K16_REGRESSION_D26CA8D          // (tools/build_variant.py: the line as it stood before d26ca8d, for tests/test_gpu_keys16.py's regression test)
    return __ballot(off <= binw);
#else
    return __ballot(off <= min(binw, 0xFFFFu - binlo));
#endif
}

// k-th smallest of the wave's 16-bit keys: lane l holds positions 16 l .. 16 l + 15 as eight packed pairs (position
// 16 l + 2 v in the low half of h[v]); positions past the end hold K16_PAD.  hist: K16_HIST_WORDS words, zero on entry and on
// return.  Same method as wave_select_hist256_u32: 256 bins around the key predicted by the previous row, bin 0 catches
// everything below the window (the subtraction saturates), a miss re-bins the whole range, a bin holding two keys of one lane
// is re-binned finer.  The predicted pass that hits is straight-line code (nine rows in ten); everything else is the loop behind
// it.  ok = false: cannot happen for 1 <= k <= n.
__device__ inline Sel16 wave_select_k16(const u16x2 (&h)[8], int k, unsigned *hist, int lane, HistWarm &warm, int *stats, K16Probe &pr)
{
    Sel16 res;
    res.ok = false;
    res.th = 0;
    res.rk = 1;
    res.ceq = 1;
    const unsigned hist_lds = k16_lds_offset(hist);
    int r = 0, cstar = 0;
    unsigned ch = 0, binlo = 0, binw = 0;
    uint64_t any = 0;
    // what the loop starts with when the straight path does not finish: a pass over the full range, or a finer one
    bool full = true, found = false;
    unsigned lo0 = 0;
    int shift = 0;
    if (warm.hi != 0) {
        shift = warm.shift;
        const unsigned back = (unsigned)(HIST256_BINS / 2 + 1) << shift;
        lo0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(max(warm.hi, back) - back));
        const K16Pass ps = k16_hist_pass(h, k, hist, hist_lds, lane, lo0, shift);
        pr.lap(1);
        if (ps.hit && ps.bstar != 0) {
            r = ps.r;
            cstar = ps.cstar;
            binlo = lo0 + (ps.bstar << shift);
            binw = (1u << shift) - 1u;
            any = k16_bin_keys(h, binlo, binw, ch);
            found = __popcll(any) == cstar;
            full = false;                              // (not found: two keys of one lane in the bin -- finer bins)
        } else
            warm.shift = min(warm.shift + 1, K16_SHIFT_MAX);
    }
    while (!found) {
        if (full) {
            K16_STAT(stats, 2);
            u16x2 mn = k16_splat(0xFFFFu), mx1 = k16_splat(0u);
#pragma unroll
            for (int v = 0; v < 8; v++) {
                unsigned raw = k16_to_u32(h[v]);
                asm volatile("" : "+v"(raw));        // (not hoisted in front of the pass loop: see wave_select_hist256_u32)
                const u16x2 x = k16_from_u32(raw);
                mn = __builtin_elementwise_min(mn, x);
                mx1 = __builtin_elementwise_max(mx1, x + k16_splat(1u));      // padding wraps to 0
            }
            const unsigned mnl = wave_umin(min((unsigned)mn.x, (unsigned)mn.y));
            const unsigned mxl = wave_umax(max((unsigned)mx1.x, (unsigned)mx1.y));
            lo0 = mnl;
            const unsigned span = mxl > mnl ? mxl - 1u - mnl : 0u;
            shift = max(0, 32 - (int)__clz(span) - HIST256_LOG2);
        } else {
            if (shift == 0) {
                // equal keys inside one lane: the threshold key is binlo itself; ranks among equals are positions' business
                res.th = binlo;
                res.ceq = cstar;
                res.rk = r;
                res.binlo = binlo;
                res.binw = 0;
                res.ch = ch;
                res.any = any;
                res.cstar = cstar;
                res.ok = true;
                return res;
            }
            K16_STAT(stats, 3);
            const int fine = max(shift - (HIST256_LOG2 - 1), 0);
            lo0 = binlo - min(binlo, 1u << fine);
            // (bins 1 .. 128 cover the old bin when lo0 = binlo - 2^fine; at the bottom of the key range bin 0 shares it)
            shift = fine;
        }
        lo0 = (unsigned)__builtin_amdgcn_readfirstlane((int)lo0);
        const K16Pass ps = k16_hist_pass(h, k, hist, hist_lds, lane, lo0, shift);
        if (!ps.hit) return res;
        r = ps.r;
        cstar = ps.cstar;
        binlo = lo0 + (ps.bstar << shift);
        binw = (1u << shift) - 1u;
        if (!full && ps.bstar == 0) { binlo = 0; binw = lo0 + binw; }        // (finer pass: bin 0 = everything below bin 1)
        any = k16_bin_keys(h, binlo, binw, ch);
        found = __popcll(any) == cstar;
        full = false;
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
    res.th = (unsigned)__builtin_amdgcn_readlane((int)ch, wl);
    res.ceq = __builtin_amdgcn_readlane(equal, wl);
    res.rk = r - __builtin_amdgcn_readlane(less, wl);
    res.binlo = binlo;
    res.binw = binw;
    res.ch = ch;
    res.any = any;
    res.cstar = cstar;
    res.ok = true;
    warm.hi = max(res.th, 1u);
    return res;
}

// the 0 / 1 flags of a lane's eight registers (low half: position 2 v, high half: position 2 v + 1) -> its 16 mask bits
__device__ inline unsigned k16_gather_flags(const u16x2 (&f)[8])
{
    unsigned acc = k16_to_u32(f[0]);
#pragma unroll
    for (int v = 1; v < 8; v++) acc |= k16_to_u32(f[v]) << (2 * v);
    return (acc | (acc >> 15)) & 0xFFFFu;
}

// bit t of the result = (key of position 16 l + t) <= thr   (thr < K16_PAD, so padding is never set)
__device__ inline unsigned k16_bits_le(const u16x2 (&h)[8], unsigned thr)
{
    const u16x2 t1_pk = k16_splat(thr + 1u), one = k16_opaque_ones();
    u16x2 f[8];
#pragma unroll
    for (int v = 0; v < 8; v++) f[v] = __builtin_elementwise_min(__builtin_elementwise_sub_sat(t1_pk, h[v]), one);    // thr + 1 -sat key > 0
    return k16_gather_flags(f);
}

// bit t = key of position 16 l + t lies in [lo, hi]   (hi < K16_PAD)
__device__ inline unsigned k16_bits_in(const u16x2 (&h)[8], unsigned lo, unsigned hi)
{
    const u16x2 lo_pk = k16_splat(lo), w1_pk = k16_splat(hi - lo + 1u), one = k16_opaque_ones();
    u16x2 f[8];
#pragma unroll
    for (int v = 0; v < 8; v++) f[v] = __builtin_elementwise_min(__builtin_elementwise_sub_sat(w1_pk, h[v] - lo_pk), one);
    return k16_gather_flags(f);
}

enum { K16_DECIDED = 0, K16_HANDOVER = 1 };

// One row (DIR 0) or column (DIR 1) after wave_select_k16: decide from the 16-bit keys if nothing else can lie in the
// winner's error band, else from recomputed float32 values of the few cells in reach; sel = this lane's 16 mask bits.
// Returns K16_HANDOVER when only exact float64 values can decide (the caller hands the keys to the side buffer).
// scratch: K16_SCRATCH words of wave-private LDS, its last 64 words zero on entry and on return.
template <int D, int DIR>
__device__ inline int k16_decide(const u16x2 (&h)[8], const Sel16 &s, int k, unsigned *scratch, int lane, const K16Ctx &cx,
                                 const float *pair_band, unsigned koff, bool adjacent_ok, const acoss_pair_desc &ds, int p, int which,
                                 unsigned &sel)
{
    sel = 0;
    if (s.th == 0u || s.th >= K16_MAX) return K16_HANDOVER;              // the threshold left the key range
    unsigned h_lo, h_hi;
    k16_reach(s.th, koff, adjacent_ok, pair_band, h_lo, h_hi);
    const bool mine = (s.any >> lane) & 1;
    bool alone;
    if (h_lo >= s.binlo && h_hi <= s.binlo + s.binw)                     // the reach lies inside the last bin: its few keys only
        alone = s.cstar == 1 || __popcll(__ballot(mine & ((s.ch - h_lo) <= (h_hi - h_lo)))) == 1;
    else {
        const unsigned in = k16_bits_in(h, h_lo, h_hi);
        const uint64_t some = __ballot(in != 0u);
        alone = (some & (some - 1)) == 0 && __ballot((in & (in - 1u)) != 0u) == 0;
    }
#ifdef ACOSS_PROBES
    if (cx.flags & 1) { sel = k16_bits_le(h, s.th); return K16_DECIDED; }
#endif
    if (alone && s.ceq == 1) {
        sel = k16_bits_le(h, s.th);
        return K16_DECIDED;
    }
    // ---- the cells in reach, one per lane: float32 values as the strip kernel formed them
    K16_STAT(cx.stats, 1);
    unsigned in = k16_bits_in(h, h_lo, h_hi);
    const int cnt = __popc(in);
    const int incl = wave_scan<OpAdd>(cnt, 0);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total > 64) return K16_HANDOVER;
    int slot = incl - cnt;
    while (in) {
        const int t = __ffs((int)in) - 1;
        in &= in - 1u;
        scratch[slot++] = (unsigned)(16 * lane + t);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // seven cells at a time: lane (g, kk) forms C[i + kk][j + kk] of cell g (all of the wave's loads in flight at once), lane g
    // adds the nine values as the strip kernel does (window_sum9 of kernel_utils.h: the association follows the row)
    const float *xrow0 = cx.xp + (int64_t)p * cx.max_nx * 16;
    for (int c0 = 0; c0 < total; c0 += 7) {
        const int g = lane / 9, kk = lane - 9 * g, el = c0 + g;
        if (lane < 63 && el < total) {
            const int ps = (int)scratch[el];
            const int ci = (DIR == 0 ? which : ps) + kk, cj = (DIR == 0 ? ps : which) + kk;
            scratch[64 + lane] = __float_as_uint(k16_c_value<D>(xrow0, cx.f32, cx.n32, ds, ci, cj));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 7 && c0 + lane < total) {
            float cv[9];
#pragma unroll
            for (int q = 0; q < 9; q++) cv[q] = __uint_as_float(scratch[64 + 9 * lane + q]);
            const int row = DIR == 0 ? which : (int)scratch[c0 + lane];
            const bool odd = window_sum9_odd(row);
            // (both associations, one select: no divergent branch)
            const float se = window_sum9(cv, false), so = window_sum9(cv, true);
            scratch[128 + c0 + lane] = __float_as_uint(fabsf(odd ? so : se));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned key32 = 0xFFFFFFFFu, k16v = K16_PAD;
    int pos = -1;
    if (lane < total) {
        pos = (int)scratch[lane];
        key32 = scratch[128 + lane];
        k16v = key16_of_bits(key32, koff);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the rk-th smallest float32 key among the cells whose 16-bit key is th
    int lt = 0, eq = 0;
    for (int c = 0; c < total; c++) {
        const unsigned kc = (unsigned)__builtin_amdgcn_readlane((int)key32, c);
        const unsigned hc = (unsigned)__builtin_amdgcn_readlane((int)k16v, c);
        lt += (hc == s.th) & (kc < key32);
        eq += (hc == s.th) & (kc == key32);
    }
    const uint64_t wm = __ballot((lane < total) & (k16v == s.th) & (lt < s.rk) & (s.rk <= lt + eq));
    if (wm == 0) return K16_HANDOVER;                  // (recomputed and stored keys disagree: cannot happen)
    const int wl = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)wm) - 1);
    if (__builtin_amdgcn_readlane(eq, wl) > 1) return K16_HANDOVER;       // equal float32 values: exact values decide
    const unsigned wkey = (unsigned)__builtin_amdgcn_readlane((int)key32, wl);
    unsigned blo, bhi;
    band_limits(wkey | 0x80000000u, pair_band, blo, bhi);
    const unsigned mk = key32 | 0x80000000u;
    if (__popcll(__ballot((lane < total) & ((mk - blo) <= (bhi - blo)))) > 1) return K16_HANDOVER;
    // decided: below the threshold key by the 16-bit keys, among its equals by the float32 keys
    const bool on = (lane < total) & (k16v == s.th) & (key32 <= wkey);
    if (on) atomicOr(&scratch[192 + (pos >> 4)], 1u << (pos & 15));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned extra = scratch[192 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    scratch[192 + lane] = 0u;
    sel = (s.th > 0u ? k16_bits_le(h, s.th - 1u) : 0u) | extra;
    return K16_DECIDED;
}

}  // namespace acoss
