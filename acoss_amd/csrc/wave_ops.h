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
struct SelectResult {
    uint64_t thr_key;
    int cut;
};

// k-th smallest (1-based, 1 <= k <= n) of the n keys a wave holds in registers, EPL per lane,
// element e of lane l having index e*64 + l (indices >= n hold UINT64_MAX and are never chosen).
// Ties at the threshold are cut lowest-index first.  Wave-uniform result.
//
// Method: bit-serial search from the most significant bit, first over the high 32-bit words
// (counts by v_cmp + v_addc per element and one DPP reduction per bit), then -- only if several
// elements share the winning high word -- over the low words of those elements.
template <int EPL>
__device__ inline SelectResult wave_select_kth(const uint64_t (&key)[EPL], int n, int k)
{
    const int lane = threadIdx.x & 63;
    unsigned hi[EPL], lo[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        hi[e] = (unsigned)(key[e] >> 32);
        lo[e] = (unsigned)key[e];
    }
    // common leading bits of the valid high words
    unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        const bool valid = e * 64 + lane < n;
        mn = min(mn, valid ? hi[e] : 0xffffffffu);
        mx = max(mx, valid ? hi[e] : 0u);
    }
    mn = wave_umin(mn);
    mx = wave_umax(mx);
    unsigned vh = mn;
    if (mn != mx) {
        const int top = 31 - __clz((int)(mn ^ mx));   // highest differing bit
        vh = (top == 31) ? 0u : (mn >> (top + 1)) << (top + 1);
        for (int b = top; b >= 0; b--) {
            const unsigned cand = vh | (1u << b);
            int c = 0;
#pragma unroll
            for (int e = 0; e < EPL; e++) c += (hi[e] < cand) ? 1 : 0;
            if (wave_sum(c) <= k - 1) vh = cand;
        }
    }
    // how many strictly below / equal in the high word
    int cl = 0, ce = 0;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        cl += (hi[e] < vh) ? 1 : 0;
        ce += (hi[e] == vh) ? 1 : 0;
    }
    cl = wave_sum(cl);
    ce = wave_sum(ce);
    unsigned vl = 0;
    if (ce == 1) {
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < EPL; e++) m = max(m, hi[e] == vh ? lo[e] : 0u);
        vl = wave_umax(m);
    } else {
        const int k2 = k - cl;   // rank among the elements sharing the high word
        for (int b = 31; b >= 0; b--) {
            const unsigned cand = vl | (1u << b);
            int c = 0;
#pragma unroll
            for (int e = 0; e < EPL; e++) c += (hi[e] == vh && lo[e] < cand) ? 1 : 0;
            if (wave_sum(c) <= k2 - 1) vl = cand;
        }
    }
    SelectResult res;
    res.thr_key = ((uint64_t)vh << 32) | vl;
    // ties: `need` of the elements equal to the threshold are taken, lowest index first
    int below = 0, equal = 0;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        below += (key[e] < res.thr_key) ? 1 : 0;
        equal += (key[e] == res.thr_key) ? 1 : 0;
    }
    below = wave_sum(below);
    equal = wave_sum(equal);
    int need = k - below;
    res.cut = 0x7fffffff;
    if (equal > need) {   // rare: exact ties across the cut
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            unsigned long long m = __ballot(key[e] == res.thr_key);
            const int c = __popcll(m);
            if (need > 0 && need <= c) {
                for (int t = 1; t < need; t++) m &= m - 1;   // drop the need-1 lowest set bits
                res.cut = e * 64 + (__ffsll((long long)m) - 1);
                need = 0;
            } else if (need > 0) {
                need -= c;
            }
        }
    }
    return res;
}

}  // namespace acoss
