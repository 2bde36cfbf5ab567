// planar_kernels.hip -- kNN selection (CRPUtils.py:169-219) on the high words of the windowed sums' order-preserving
// keys, the uint32 matrix written by crp_strip_kernel<..., PLANAR> (same element indexing as the float64 matrix).
//
// Why: the selection kernels of crp_kernels.hip run at 75 % of their load-only time -- they are bound by
// reading 8 bytes per element of T, twice (rows, columns).  The k-th smallest of a row is decided by the high
// words alone unless another element shares the winner's high word (~6e-4 of the rows at 992 columns); those
// rows and columns go to the fix-up kernel, which recomputes the exact values of the tied elements from the
// features (the strip kernel does not write the low words at all).  So the two big kernels read 4 bytes per
// element, hold 16 instead of 32 registers of data per lane, and need no key conversion.
//
// Outputs are the same as acoss_mask_bits_batch: thresholds (key = high word : 0xffffffff, which selects
// exactly the same elements as the full key of the k-th smallest when its high word is unique), tie cuts,
// and the row / column bit planes that combine_bits_kernel turns into the bit-packed mutual mask.
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

// lane e (< E) = lane mask of "position e*64 + lane exists"
template <int E>
__device__ inline uint64_t planar_slot_valid(int n, int lane)
{
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int e = 0; e < E; e++) planar_put_lane_u64(lo, hi, __ballot(e * 64 + lane < n), e);
    return ((uint64_t)hi << 32) | lo;
}

// word e of out = lanes whose element e is selected (high word <= the threshold's), positions e*64 + lane
template <int E>
__device__ inline void planar_emit_bits(const unsigned (&h)[E], unsigned thr_hi, uint64_t valid, uint64_t *out, int lane,
                                        int64_t stride = 1)
{
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int e = 0; e < E; e++) planar_put_lane_u64(lo, hi, __ballot(h[e] <= thr_hi), e);
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

// after the selection on approximate keys: the result stands only if the winner is alone in its error band
template <int E>
__device__ inline void band_resolve(const unsigned (&h)[E], int n, const float *band, int p, int lane, SelectResult &res)
{
    if (band == nullptr || res.cut == SELECT_UNRESOLVED || res.cut < 0 || res.thr_key == ~0ull) return;
    const unsigned th = (unsigned)(res.thr_key >> 32);
    unsigned lo, hi;
    band_limits(th, band + 2 * p, lo, hi);
    if (band_count<E>(h, lo, hi) > 1) {
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

// ---- rows ------------------------------------------------------------------------------------------------
constexpr int PL_ROWS_PER_WAVE = 8;

// MODE (development probes): 1 = loads only
template <int MODE = 0, int E = 16>
__global__ __launch_bounds__(256, E == 16 ? 6 : 4) void select_rows_planar_kernel(const uint32_t *__restrict__ Thi,
                                                                   const acoss_pair_desc *__restrict__ descs, int win,
                                                                   double kv, int k_mode, ThreshWork w, int rows_blocks)
{
    __shared__ __attribute__((aligned(16))) unsigned hist_all[4 * HIST_WORDS];
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / rows_blocks;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = ((lb % rows_blocks) * 4 + wave) * PL_ROWS_PER_WAVE;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (r0 >= M) return;
    const int r1 = min(r0 + PL_ROWS_PER_WAVE, M);
    const int lane = threadIdx.x & 63;
    const int k = knn_count(k_mode, kv, N);
    uint64_t *thr = w.row_thr + (int64_t)p * w.max_m;
    int *cut = w.row_cut + (int64_t)p * w.max_m;
    unsigned *hist = hist_all + wave * HIST_WORDS;
    hist_clear(hist, lane);
    // (float32 keys carry 23 mantissa bits where float64 high words carry 20: the same window of values is 8x as many keys)
    HistWarm warm{0, HIST_WARM_SHIFT0 + (w.band != nullptr ? 3 : 0)};
    const uint64_t valid = planar_slot_valid<E>(N, lane);
    const bool wide = N > (E - 1) * 64;       // wave-uniform: only the last slot can run past the row
    for (int i = r0; i < r1; i++) {
        const uint32_t *row = Thi + ds.crp_off + (int64_t)i * ds.crp_pitch;
        unsigned h[E];
        if (wide) {
            // wave-uniform row pointer + one lane offset + immediates: 256 contiguous bytes per load instruction
#pragma unroll
            for (int e = 0; e < E - 1; e++) h[e] = row[(unsigned)(e * 64 + lane)];
            h[E - 1] = row[(unsigned)min((E - 1) * 64 + lane, N - 1)];
        } else {
            // (cold path: the lane number is laundered through an empty asm so that the sixteen clamped offsets are
            // computed here instead of being hoisted in front of the row loop, where they would set the register
            // count of the whole kernel)
            int lc = lane;
            asm volatile("" : "+v"(lc));
#pragma unroll
            for (int e = 0; e < E; e++) h[e] = row[(unsigned)min(e * 64 + lc, N - 1)];
        }
        if constexpr (MODE == 1) {
            unsigned acc = 0;
#pragma unroll
            for (int e = 0; e < E; e++) acc += h[e];
            if (acc == 0x12345u) thr[i] = 0;
            continue;
        }
        SelectResult res;
        if (!planar_trivial(k, N, res)) {
            res = wave_select_hist_u32<E>(h, N, k, hist, lane, warm);
            band_resolve<E>(h, N, w.band, p, lane, res);
        }
        if (lane == 0) {
            thr[i] = res.thr_key;
            cut[i] = res.cut;
        }
        if (w.row_bits && res.cut != SELECT_UNRESOLVED)
            planar_emit_bits<E>(h, res.cut < 0 ? 0u : (unsigned)(res.thr_key >> 32), res.cut < 0 ? 0ull : valid,
                                w.row_bits + ((int64_t)p * w.max_m + i) * E, lane);
    }
}

// ---- columns ---------------------------------------------------------------------------------------------
// One 8-wave block stages 16 columns (64-byte row segments, 16-byte loads) through LDS; wave v then selects
// in columns 2v and 2v + 1, the second one inside the window predicted by the first.  Once a wave holds its
// two columns in registers their LDS slots become its histogram.
constexpr int PL_COLS = 16;
constexpr int PL_LDC = 512 + 36;        // words per staged half column (two of them hold one wave's histogram; column pairs 8 banks apart)

static_assert(2 * (512 + 36) >= HIST_WORDS, "a wave's two half-column slots hold its histogram");

template <int MODE = 0>
__global__ __launch_bounds__(512, 4) void select_cols_planar_kernel(const uint32_t *__restrict__ Thi,
                                                                   const acoss_pair_desc *__restrict__ descs, int win,
                                                                   double kv, int k_mode, ThreshWork w, int col_blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned pcolbuf[];      // [PL_COLS][PL_LDC]
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / col_blocks;
    const int j0 = (lb % col_blocks) * PL_COLS;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (j0 >= N) return;
    unsigned ha[16], hb[16];
    {
        // 64 rows x 8 column pairs per sweep (8-byte loads, 64-byte row segments), 16 sweeps (M <= 1024).
        // (16-byte loads with 4 lanes per row segment take 2.3x as long: measured.)
        const int c2 = threadIdx.x & 7, rr = threadIdx.x >> 3;
        // block-uniform: even pitch / offset (8-byte loads) and a full column group -> wave-uniform base pointer +
        // 32-bit word offsets
        const bool fast = ((ds.crp_pitch & 1) == 0) && ((ds.crp_off & 1) == 0) && (j0 + PL_COLS <= N);
        uint2 tmp[16];
        if (fast) {
            const uint32_t *pb = Thi + ds.crp_off + j0 + 2 * c2;
#pragma unroll
            for (int s = 0; s < 16; s++)
                tmp[s] = *reinterpret_cast<const uint2 *>(pb + (unsigned)(min(s * 64 + rr, M - 1) * ds.crp_pitch));
        } else {
            // (cold path; thread ids laundered so that its address arithmetic is not hoisted: see the row kernel)
            int cc = c2, rc = rr;
            asm volatile("" : "+v"(cc), "+v"(rc));
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const int64_t ri = ds.crp_off + (int64_t)min(s * 64 + rc, M - 1) * ds.crp_pitch;
                tmp[s].x = Thi[ri + min(j0 + 2 * cc + 0, N - 1)];
                tmp[s].y = Thi[ri + min(j0 + 2 * cc + 1, N - 1)];
            }
        }
        // staged in two halves of 512 rows (35 KB of LDS: three blocks per CU instead of two).  Slot s*64 + rr of a
        // column holds row min(s*64 + rr, M-1), so the readers need no clamp.
        const int wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int lane_ = threadIdx.x & 63;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (half) __syncthreads();              // every wave has read the first half
#pragma unroll
            for (int s = 0; s < 8; s++) {
                unsigned *dst = pcolbuf + (2 * c2) * PL_LDC + s * 64 + rr;
                dst[0 * PL_LDC] = tmp[half * 8 + s].x;
                dst[1 * PL_LDC] = tmp[half * 8 + s].y;
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; e++) {
                ha[half * 8 + e] = pcolbuf[(2 * wave_) * PL_LDC + e * 64 + lane_];
                hb[half * 8 + e] = pcolbuf[(2 * wave_ + 1) * PL_LDC + e * 64 + lane_];
            }
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int ja = j0 + 2 * wave;
    if (ja >= N) return;
    if constexpr (MODE == 1) {
        unsigned acc = 0;
#pragma unroll
        for (int e = 0; e < 16; e++) acc += ha[e] + hb[e];
        if (acc == 0x12345u) w.col_cut[0] = 0;
        return;
    }
    unsigned *hist = pcolbuf + (2 * wave) * PL_LDC;      // the wave's two slots: 2 * PL_LDC words >= HIST_WORDS, 16-byte aligned
    hist_clear(hist, lane);
    const int k = knn_count(k_mode, kv, M);
    const uint64_t valid = planar_slot_valid<16>(M, lane);
    // (float32 keys carry 23 mantissa bits where float64 high words carry 20: the same window of values is 8x as many keys)
    HistWarm warm{0, HIST_WARM_SHIFT0 + (w.band != nullptr ? 3 : 0)};
    auto column = [&](const unsigned (&h)[16], const int j) {
        SelectResult res;
        if (MODE == 2) { res.thr_key = ((uint64_t)__builtin_amdgcn_readfirstlane((int)h[1]) << 32) | 0xffffffffull; res.cut = 0x7fffffff; }
        else if (!planar_trivial(k, M, res)) {
            res = wave_select_hist_u32<16>(h, M, k, hist, lane, warm);
            band_resolve<16>(h, M, w.band, p, lane, res);
        }
        if (lane == 0) {
            w.col_thr[(int64_t)p * w.max_n + j] = res.thr_key;
            w.col_cut[(int64_t)p * w.max_n + j] = res.cut;
        }
        if (w.col_bits && res.cut != SELECT_UNRESOLVED)
            planar_emit_bits<16>(h, res.cut < 0 ? 0u : (unsigned)(res.thr_key >> 32), res.cut < 0 ? 0ull : valid,
                                 w.col_word(p, j, 0), lane, w.max_n);
    };
    column(ha, ja);
    if (ja + 1 < N) column(hb, ja + 1);
}

// Columns of matrices with up to 2048 rows: COLS waves, one column each (32 values per lane), the block's
// columns staged whole (8 columns = 66 KB of LDS: two blocks per CU, one loading while the other selects).
// Same selection and outputs as above.
constexpr int PLW_LDC = 2048 + 4;

template <int COLS>
__global__ __launch_bounds__(64 * COLS) void select_cols_planar_wide_kernel(const uint32_t *__restrict__ Thi,
                                                                      const acoss_pair_desc *__restrict__ descs, int win,
                                                                      double kv, int k_mode, ThreshWork w, int col_blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned pcolbuf[];      // [COLS][PLW_LDC]
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / col_blocks;
    const int j0 = (lb % col_blocks) * COLS;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (j0 >= N) return;
    {
        // 128 rows x COLS / 2 column pairs per sweep, 16 sweeps (M <= 2048); sweeps past the last row are skipped
        const int c2 = threadIdx.x & (COLS / 2 - 1), rr = threadIdx.x / (COLS / 2);
        const bool fast = ((ds.crp_pitch & 1) == 0) && ((ds.crp_off & 1) == 0) && (j0 + COLS <= N);
        const int sweeps = (M + 127) >> 7;
        const uint32_t *pb = Thi + ds.crp_off + j0 + 2 * c2;        // (fast) block-uniform base + 32-bit word offsets
        uint2 tmp[16];
#pragma unroll
        for (int s = 0; s < 16; s++) {
            if (s < sweeps) {          // block-uniform: no loads for sweeps past the last row
                const int row = min(s * 128 + rr, M - 1);
                if (fast) {
                    tmp[s] = *reinterpret_cast<const uint2 *>(pb + (unsigned)(row * ds.crp_pitch));
                } else {
                    const int64_t ri = ds.crp_off + (int64_t)row * ds.crp_pitch;
                    tmp[s].x = Thi[ri + min(j0 + 2 * c2 + 0, N - 1)];
                    tmp[s].y = Thi[ri + min(j0 + 2 * c2 + 1, N - 1)];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 16; s++) {
            if (s < sweeps) {
                unsigned *dst = pcolbuf + (2 * c2) * PLW_LDC + s * 128 + rr;
                dst[0 * PLW_LDC] = tmp[s].x;
                dst[1 * PLW_LDC] = tmp[s].y;
            }
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = j0 + wave;
    if (j >= N) return;
    unsigned h[32];
#pragma unroll
    for (int e = 0; e < 32; e++) h[e] = pcolbuf[wave * PLW_LDC + min(e * 64 + lane, M - 1)];
    unsigned *hist = pcolbuf + wave * PLW_LDC;      // the wave's own column slot: PLW_LDC words >= HIST_WORDS
    hist_clear(hist, lane);
    const int k = knn_count(k_mode, kv, M);
    const uint64_t valid = planar_slot_valid<32>(M, lane);
    // (float32 keys carry 23 mantissa bits where float64 high words carry 20: the same window of values is 8x as many keys)
    HistWarm warm{0, HIST_WARM_SHIFT0 + (w.band != nullptr ? 3 : 0)};
    SelectResult res;
    if (!planar_trivial(k, M, res)) {
        res = wave_select_hist_u32<32>(h, M, k, hist, lane, warm);
        band_resolve<32>(h, M, w.band, p, lane, res);
    }
    if (lane == 0) {
        w.col_thr[(int64_t)p * w.max_n + j] = res.thr_key;
        w.col_cut[(int64_t)p * w.max_n + j] = res.cut;
    }
    if (w.col_bits && res.cut != SELECT_UNRESOLVED)
        planar_emit_bits<32>(h, res.cut < 0 ? 0u : (unsigned)(res.thr_key >> 32), res.cut < 0 ? 0ull : valid,
                             w.col_word(p, j, 0), lane, w.max_n);
}

// ---- fix-up: rows / columns whose winner shares its high word -----------------------------------------------
// One windowed sum, exactly as crp_strip_kernel forms it: dot product as an FMA chain over the bins of the rolled
// x frame, C = max(fma(-2, dot, |x|^2 + |y|^2), 0), the window's C values added in order (the matrix-core and VALU
// forms of that chain agree bit for bit: tests/test_gpu_fast_path.py).
__device__ inline double planar_exact_value(const double *__restrict__ feats, const double *__restrict__ norms, int d,
                                            const acoss_pair_desc &ds, int win, int i, int j)
{
    double s = 0.0;
    for (int k = 0; k < win; k++) {
        const double *x = feats + (ds.x_row0 + i + k) * d, *y = feats + (ds.y_row0 + j + k) * d;
        double acc = 0.0;
        for (int b = 0; b < d; b++) {
            int src = b - ds.shift;
            if (src < 0) src += d;
            acc = fma(x[src], y[b], acc);
        }
        s += fmax(fma(-2.0, acc, norms[ds.x_row0 + i + k] + norms[ds.y_row0 + j + k]), 0.0);
    }
    return s;
}

// One row (DIR 0) or column (DIR 1) the selection kernels left unresolved, the general way: exact values for the
// elements that share the winner's key (or, with approximate keys, lie inside its error band), then the bit-serial
// selection over 64-bit keys.
template <int DIR, int E>
__device__ inline void fix_row_generic(const uint32_t *__restrict__ Thi, const double *__restrict__ feats,
                                       const double *__restrict__ norms, int d, const acoss_pair_desc &ds, int win,
                                       const ThreshWork &w, int p, int which, int len, int k, uint64_t *thr, int *cut, int lane)
{
    // the selection kernel left the high word the tied elements share: only those few need their exact value;
    // every other element is ordered by its high word alone
    const unsigned th = (unsigned)(thr[which] >> 32);
    unsigned blo = 0u, bhi = 0u;
    if (w.band != nullptr) band_limits(th, w.band + 2 * p, blo, bhi);
    uint64_t key[E];
    int idx[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        idx[e] = e * 64 + lane;
        const int q = min(idx[e], len - 1);
        const unsigned h = Thi[planar_word(ds.crp_off + (DIR == 0 ? (int64_t)which * ds.crp_pitch + q : (int64_t)q * ds.crp_pitch + which))];
        uint64_t kx = (uint64_t)h << 32;
        bool exact = th == 0u || h == th;
        if (w.band != nullptr) {
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
    if (lane == 0) {
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
        if (lane < E) bits[lane * bstride] = mine;
    }
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

template <int DIR, int E>
__device__ inline bool fix_row_band(FixSmem &sm, const uint32_t *__restrict__ Thi, const double *__restrict__ feats,
                                    const double *__restrict__ norms, int d, const acoss_pair_desc &ds, int win,
                                    const ThreshWork &w, int p, int which, int len, int k, uint64_t *thr, int *cut, int lane)
{
    const unsigned th = (unsigned)(thr[which] >> 32);
    if (th == 0u || win != 9 || d > FIX_MAXD) return false;
    unsigned blo, bhi;
    band_limits(th, w.band + 2 * p, blo, bhi);
    unsigned h[E];
    int below = 0, n = 0;
    int myidx[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int pos = e * 64 + lane;
        const int q = min(pos, len - 1);
        h[e] = Thi[planar_word(ds.crp_off + (DIR == 0 ? (int64_t)which * ds.crp_pitch + q : (int64_t)q * ds.crp_pitch + which))];
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
            const double *x = feats + (ds.x_row0 + i + kk) * d, *y = feats + (ds.y_row0 + j + kk) * d;
            double xv[FIX_MAXD], yv[FIX_MAXD];
#pragma unroll
            for (int b = 0; b < FIX_MAXD; b++) {
                int src = b - ds.shift;
                if (src < 0) src += d;
                xv[b] = b < d ? x[src] : 0.0;
                yv[b] = b < d ? y[b] : 0.0;
            }
            const double nn = norms[ds.x_row0 + i + kk] + norms[ds.y_row0 + j + kk];
            double acc = 0.0;
#pragma unroll
            for (int b = 0; b < FIX_MAXD; b++)
                if (b < d) acc = fma(xv[b], yv[b], acc);
            sm.cval[g * 9 + kk] = fmax(fma(-2.0, acc, nn), 0.0);
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
    if (lane == ll) {
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
        if (lane < E) bits[lane * bstride] = mine;
    }
    __syncthreads();
    return true;
}

template <int DIR, int E = 16>
__global__ __launch_bounds__(64) void select_fix_planar_kernel(const uint32_t *__restrict__ Thi, const double *__restrict__ feats,
                                                               const double *__restrict__ norms, int d,
                                                               const acoss_pair_desc *__restrict__ descs, int win,
                                                               double kv, int k_mode, ThreshWork w, int groups)
{
    __shared__ FixSmem sm;
    const int p = blockIdx.x / groups, g = blockIdx.x % groups;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int count = DIR == 0 ? M : N;
    const int len = DIR == 0 ? N : M;
    const int lane = threadIdx.x;
    const int t = g * 64 + lane;
    uint64_t *thr = (DIR == 0 ? w.row_thr + (int64_t)p * w.max_m : w.col_thr + (int64_t)p * w.max_n);
    int *cut = (DIR == 0 ? w.row_cut + (int64_t)p * w.max_m : w.col_cut + (int64_t)p * w.max_n);
    unsigned long long todo = __ballot(t < count && cut[t] == SELECT_UNRESOLVED);
    if (todo == 0) return;
    const int k = knn_count(k_mode, kv, len);
    while (todo) {
        const int which = g * 64 + (__ffsll((long long)todo) - 1);     // wave-uniform
        todo &= todo - 1;
        if (w.band != nullptr && fix_row_band<DIR, E>(sm, Thi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane)) continue;
        fix_row_generic<DIR, E>(Thi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane);
    }
}

// defined in crp_kernels.hip
int launch_combine_bits(const acoss_pair_desc *descs, int K, int win, int mutual, ThreshWork w, uint64_t *bits, hipStream_t st);

static void kappa_mode_planar(double kappa, double &kv, int &mode)
{
    if (kappa == 0.0) { kv = 0.0; mode = 2; }       // CRPUtils.py:188-189
    else if (kappa < 1.0) { kv = kappa; mode = 0; }  // :190-191
    else { kv = kappa; mode = 1; }                   // :192-193
}

static int run_planar(int probe, const uint32_t *planes, const double *feats, const double *norms, int d,
                      const acoss_pair_desc *descs, int K, int win,
                      int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits, void *work, size_t work_bytes,
                      hipStream_t st, const float *band = nullptr)
{
    if (!planes || !feats || !norms || d < 1 || !descs || !work || K < 0 || win < 1 || max_nx < win || max_ny < win || kappa < 0.0) {
        set_error("mask_bits_planar: bad argument");
        return ACOSS_EINVAL;
    }
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    if (max_m > 2048 || max_n > 2048) {
        set_error("mask_bits_planar: matrices larger than 2048 x 2048 are not supported");
        return ACOSS_ENOTSUP;
    }
    if (work_bytes < thresh_work_bytes(K, max_m, max_n, true)) {
        set_error("mask_bits_planar: workspace too small");
        return ACOSS_EINVAL;
    }
    ThreshWork w = thresh_work_layout(work, K, max_m, max_n, true);
    w.band = band;
    if (K == 0) return ACOSS_OK;
    double kv;
    int mode;
    kappa_mode_planar(kappa, kv, mode);
    const int rb = ceil_div(max_m, 4 * PL_ROWS_PER_WAVE);
    const int cb = ceil_div(max_n, PL_COLS);
    const size_t lds = sizeof(unsigned) * PL_COLS * PL_LDC;
    if ((int64_t)K * rb > 0x7fffffffLL || (int64_t)K * cb > 0x7fffffffLL) { set_error("mask_bits_planar: batch too large"); return ACOSS_ENOTSUP; }
    if (w.wpr == 32) {
        // up to 2048 x 2048: 32 values per lane, one column per wave (the probes are for the 16-word kernels)
        if (probe != 0) { set_error("mask_bits_planar: no probes for matrices beyond 1024 x 1024"); return ACOSS_ENOTSUP; }
        hipLaunchKernelGGL((select_rows_planar_kernel<0, 32>), dim3((unsigned)((int64_t)K * rb)), dim3(256), 0, st, planes, descs, win, kv, mode, w, rb);
        int rc = launch_check("select_rows_planar_kernel<32>");
        if (rc) return rc;
        const int gm = ceil_div(max_m, 64), gn = ceil_div(max_n, 64);
        hipLaunchKernelGGL((select_fix_planar_kernel<0, 32>), dim3((unsigned)((int64_t)K * gm)), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, gm);
        rc = launch_check("select_fix_planar_kernel<rows, 32>");
        if (rc) return rc;
        if (mutual) {
            constexpr int WC = 8;
            const int cbw = ceil_div(max_n, WC);
            const size_t ldw = sizeof(unsigned) * WC * PLW_LDC;
            ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_planar_wide_kernel<WC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldw));
            hipLaunchKernelGGL(select_cols_planar_wide_kernel<WC>, dim3((unsigned)((int64_t)K * cbw)), dim3(64 * WC), ldw, st, planes, descs, win, kv, mode, w, cbw);
            rc = launch_check("select_cols_planar_wide_kernel");
            if (rc) return rc;
            hipLaunchKernelGGL((select_fix_planar_kernel<1, 32>), dim3((unsigned)((int64_t)K * gn)), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, gn);
            rc = launch_check("select_fix_planar_kernel<cols, 32>");
            if (rc) return rc;
        }
        return launch_combine_bits(descs, K, win, mutual, w, bits, st);
    }
    if (probe == 1) {
        hipLaunchKernelGGL(select_rows_planar_kernel<1>, dim3((unsigned)((int64_t)K * rb)), dim3(256), 0, st, planes, descs, win, kv, mode, w, rb);
        return launch_check("select_rows_planar probe");
    }
    if (probe == 13) {      // column kernel without the selection itself (staging, reload, bit emission)
        ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_planar_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(select_cols_planar_kernel<2>, dim3((unsigned)((int64_t)K * cb)), dim3(512), lds, st, planes, descs, win, kv, mode, w, cb);
        return launch_check("select_cols_planar probe");
    }
    if (probe == 11) {
        ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_planar_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(select_cols_planar_kernel<1>, dim3((unsigned)((int64_t)K * cb)), dim3(512), lds, st, planes, descs, win, kv, mode, w, cb);
        return launch_check("select_cols_planar probe");
    }
    if (probe == 0 || probe == 2) {
        hipLaunchKernelGGL(select_rows_planar_kernel<0>, dim3((unsigned)((int64_t)K * rb)), dim3(256), 0, st, planes, descs, win, kv, mode, w, rb);
        int rc = launch_check("select_rows_planar_kernel");
        if (rc) return rc;
        if (probe == 2) return ACOSS_OK;
        const int groups = ceil_div(max_m, 64);
        hipLaunchKernelGGL(select_fix_planar_kernel<0>, dim3((unsigned)((int64_t)K * groups)), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, groups);
        rc = launch_check("select_fix_planar_kernel<rows>");
        if (rc) return rc;
    }
    if (mutual || probe == 12) {
        ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_planar_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(select_cols_planar_kernel<0>, dim3((unsigned)((int64_t)K * cb)), dim3(512), lds, st, planes, descs, win, kv, mode, w, cb);
        int rc = launch_check("select_cols_planar_kernel");
        if (rc) return rc;
        if (probe == 12) return ACOSS_OK;
        const int groups = ceil_div(max_n, 64);
        hipLaunchKernelGGL(select_fix_planar_kernel<1>, dim3((unsigned)((int64_t)K * groups)), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, groups);
        rc = launch_check("select_fix_planar_kernel<cols>");
        if (rc) return rc;
    }
    return launch_combine_bits(descs, K, win, mutual, w, bits, st);
}

}  // namespace acoss

using namespace acoss;

extern "C" {

int acoss_mask_bits_planar_batch(const uint32_t *planes, const double *feats, const double *norms, int d,
                                 const acoss_pair_desc *descs, int K, int win,
                                 int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits, void *work,
                                 size_t work_bytes, void *stream)
{
    if (!bits) { set_error("mask_bits_planar_batch: bad argument"); return ACOSS_EINVAL; }
    return run_planar(0, planes, feats, norms, d, descs, K, win, max_nx, max_ny, kappa, mutual, bits, work, work_bytes,
                      (hipStream_t)stream);
}

int acoss_mask_bits_planar32_batch(const uint32_t *keys, const float *band, const double *feats, const double *norms, int d,
                                   const acoss_pair_desc *descs, int K, int win,
                                   int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits, void *work,
                                   size_t work_bytes, void *stream)
{
    if (!bits || !band) { set_error("mask_bits_planar32_batch: bad argument"); return ACOSS_EINVAL; }
    return run_planar(0, keys, feats, norms, d, descs, K, win, max_nx, max_ny, kappa, mutual, bits, work, work_bytes,
                      (hipStream_t)stream, band);
}

// development probe (not part of the public ABI): 1 = row loads only, 2 = row selection kernel alone,
// 11 = column loads only, 12 = column selection kernel alone
int acoss_dev_planar_probe(int probe, const uint32_t *planes, const double *feats, const double *norms, int d,
                           const acoss_pair_desc *descs, int K,
                           int win, int max_nx, int max_ny, double kappa, void *work, size_t work_bytes, void *stream)
{
    return run_planar(probe, planes, feats, norms, d, descs, K, win, max_nx, max_ny, kappa, 1, nullptr, work, work_bytes,
                      (hipStream_t)stream);
}

}  // extern "C"
