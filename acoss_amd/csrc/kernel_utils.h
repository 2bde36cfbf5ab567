// kernel_utils.h -- device helpers shared by the CRP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

namespace acoss {

// ---------------------------------------------------------------------------------------------
// XCD-aware block remap: hardware deals consecutive block ids round-robin over the 8 XCDs; this
// maps them back so that logically consecutive blocks (tiles of one pair, which share the two
// songs' feature rows) run on one XCD and hit its L2.  Bijective for any grid size.
// ---------------------------------------------------------------------------------------------
__device__ inline int xcd_remap(int b, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)), i.e. it waits for every outstanding global store and prefetch of the
// wave -- a full HBM round trip per barrier in a kernel that streams results out while it iterates.
// Use this where the data exchanged across the barrier lives in LDS and global results are never
// read back by the block.
__device__ inline void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Pin the completion of an earlier global load HERE.  hipcc places the s_waitcnt vmcnt(N) of a load at
// its first use; if that use sits inside a loop that also issues stores or prefetches, the counted wait
// is re-executed every iteration and -- the counter being in-order -- drains those younger operations
// too.  Touching the value before the loop moves the wait out of it.
__device__ inline void settle(double v) { asm volatile("" ::"v"(v)); }
__device__ inline void settle(float v) { asm volatile("" ::"v"(v)); }

// sqrt(max(c, 0)) for the CSM epilogue.  float64: v_rsq_f64 seed, one Goldschmidt step and two
// Newton corrections (the sequence the compiler's own sqrt uses, correctly rounded), but without
// the per-element exponent rescaling and class tests: zero flows through the iteration exactly
// (0 * finite seed), and the only inputs the fast form cannot take -- positive values below
// 2^-900 -- are sent to the library sqrt by a wave-uniform branch that is never taken on real
// features.
__device__ inline double csm_sqrt(double c)
{
    c = fmax(c, 0.0);
    const double tiny = 0x1.0p-900;
    if (__builtin_expect(__any(c > 0.0 && c < tiny), 0)) return sqrt(c);
    const double y = __builtin_amdgcn_rsq(fmax(c, tiny));
    double g = c * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, c), h, g);
    g = fma(fma(-g, g, c), h, g);
    return g;
}
__device__ inline float csm_sqrt(float c) { return sqrtf(fmaxf(c, 0.0f)); }
// The two halves of csm_sqrt(double) for kernels that take many roots at once: one wave-uniform test over all of a lane's
// values, then the branch-free iteration (c >= 0, not in (0, 2^-900)) -- the same instructions, the same bits.
__device__ inline bool csm_sqrt_is_tiny(double c) { return c > 0.0 && c < 0x1.0p-900; }
template <bool NONZERO = false>
__device__ inline double csm_sqrt_fast(double c)
{
    // (NONZERO: the caller has established c >= 2^-900, the seed needs no guard against 1 / sqrt(0))
    const double y = __builtin_amdgcn_rsq(NONZERO ? c : fmax(c, 0x1.0p-900));
    double g = c * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, c), h, g);
    g = fma(fma(-g, g, c), h, g);
    return g;
}

// The float32 window sum T~ of output row i, as every float32 kernel forms it (strip32_kernels.hip; the recomputation in
// keys16.h): nine non-negative C values of one diagonal, c[k] = C[i + k][j + k], added pairwise (depth 4: the bound
// 9.5 u T of engine.PLANAR32_BOUND_T covers it with room to spare).  The association depends on the row,
// odd = (i mod 7) & 1: a wave of the row-band kernel forms seven consecutive rows from fifteen C values of a diagonal
// and shares the partial sums between neighbouring rows (24 additions for seven sums; summing each window on its own: 56).
//     even rows:  c0 + (((c1 + c2) + (c3 + c4)) + ((c5 + c6) + (c7 + c8)))
//     odd rows:   (((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7))) + c8
template <typename T>
__device__ inline T window_sum9(const T *c, bool odd)
{
    if (odd) return (((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]))) + c[8];
    return c[0] + (((c[1] + c[2]) + (c[3] + c[4])) + ((c[5] + c[6]) + (c[7] + c[8])));
}
__device__ inline bool window_sum9_odd(int row) { return (((row % 7) + 7) % 7) & 1; }


}  // namespace acoss
