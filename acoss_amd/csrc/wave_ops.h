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

}  // namespace acoss
