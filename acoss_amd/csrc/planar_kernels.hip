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
#include "planar_select.h"

namespace acoss {


// ---- side buffer: undecided rows / columns of the float32-approximate selection ------------------------------
// The wave that could not decide a row holds its keys in registers in position order: it writes them to a slot (4 KB,
// coalesced) and the refinement kernel below finishes the row from there -- no second strided walk over the key matrix
// (the column refinement of select_fix_planar_kernel reads 3.3 GB per 4096 pairs that way).
constexpr int SELECT_HANDED_OVER = -3;

template <int E>
__device__ inline void side_hand_over(const unsigned (&h)[E], const ThreshWork &w, int p, int dir, int which, int len,
                                      SelectResult &res, int lane)
{
    if (w.side_keys == nullptr) return;
    int slot = 0;
    if (lane == 0) slot = atomicAdd(w.side_counter, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= w.side_cap) return;            // no room: stays unresolved for the strided fix-up kernel
    uint32_t *dst = w.side_keys + (int64_t)slot * 1024;
#pragma unroll
    for (int e = 0; e < E; e++)
        if (e * 64 + lane < len) dst[e * 64 + lane] = h[e];
    if (lane == 0) w.side_slots[slot] = make_int4(p, dir, which, (int)(unsigned)(res.thr_key >> 32));
    res.cut = SELECT_HANDED_OVER;
}

__global__ __launch_bounds__(64) void select_fix_side_kernel(const double *__restrict__ feats, const double *__restrict__ norms, int d,
                                                             const acoss_pair_desc *__restrict__ descs, int win, double kv, int k_mode,
                                                             ThreshWork w)
{
    __shared__ FixSmem sm;
    const int slot = blockIdx.x;
    if (slot >= min(*w.side_counter, w.side_cap)) return;
    const int4 rec = w.side_slots[slot];
    const int p = rec.x, dir = rec.y, which = rec.z;
    const unsigned thr_hi = (unsigned)rec.w;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int len = dir ? M : N;
    const int lane = threadIdx.x;
    const int k = knn_count(k_mode, kv, len);
    const uint32_t *keys = w.side_keys + (int64_t)slot * 1024;
    auto key_at = [&](int q) { return keys[q]; };
    uint64_t *thr = dir ? w.col_thr + (int64_t)p * w.max_n : w.row_thr + (int64_t)p * w.max_m;
    int *cut = dir ? w.col_cut + (int64_t)p * w.max_n : w.row_cut + (int64_t)p * w.max_m;
    if (dir == 0) {
        if (fix_row_band<0, 16>(sm, key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane)) return;
        fix_row_generic<0, 16>(key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane);
    } else {
        if (fix_row_band<1, 16>(sm, key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane)) return;
        fix_row_generic<1, 16>(key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane);
    }
}

// ---- rows ------------------------------------------------------------------------------------------------
#ifndef PL_RPW
#define PL_RPW 16
#endif
#ifndef PL_WARM32
#define PL_WARM32 3          // float32 keys: the same window of values is 2^3 as many keys as float64 high words
#endif
constexpr int PL_ROWS_PER_WAVE = PL_RPW;

// MODE (development probes): 1 = loads only
template <int MODE = 0, int E = 16>
__global__ __launch_bounds__(256, E == 16 ? 8 : 4) void select_rows_planar_kernel(const uint32_t *__restrict__ Thi,
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
    HistWarm warm{0, (E == 16 ? HIST256_SHIFT0 : HIST_WARM_SHIFT0) + (w.band != nullptr ? PL_WARM32 : 0)};
    const uint64_t valid = planar_slot_valid<E>(N, lane);
    const bool wide = N > (E - 1) * 64;       // wave-uniform: only the last slot can run past the row
    for (int i = r0; i < r1; i++) {
        const uint32_t *row = Thi + ds.crp_off + (int64_t)i * ds.crp_pitch;
        unsigned h[E];
        if constexpr (MODE == 3) {
            // probe: no loads, keys made up in registers (a smooth ramp + hashed low bits, so that the window prediction
            // works as on real rows): the selection alone at full occupancy
#pragma unroll
            for (int e = 0; e < E; e++) {
                const unsigned q = (unsigned)(e * 64 + lane);
                h[e] = 0xbf000000u + (((q * 2654435761u) ^ ((unsigned)i * 40503u)) >> 10) + ((q * 2246822519u) >> 12);
            }
        } else
        if (wide) {
            // wave-uniform row pointer + one lane offset + immediates: 256 contiguous bytes per load instruction
#pragma unroll
            for (int e = 0; e < E - 1; e++) h[e] = __builtin_nontemporal_load(row + (unsigned)(e * 64 + lane));      // read once: no cache allocation
            h[E - 1] = __builtin_nontemporal_load(row + (unsigned)min((E - 1) * 64 + lane, N - 1));
        } else {
            // (cold path: the lane number is laundered through an empty asm so that the sixteen clamped offsets are
            // computed here instead of being hoisted in front of the row loop, where they would set the register
            // count of the whole kernel)
            int lc = lane;
            asm volatile("" : "+v"(lc));
#pragma unroll
            for (int e = 0; e < E; e++) h[e] = row[(unsigned)min(e * 64 + lc, N - 1)];
        }
        if constexpr (E == 16) planar_pad_keys<E>(h, N, lane, wide);
        if constexpr (MODE == 1) {
            unsigned acc = 0;
#pragma unroll
            for (int e = 0; e < E; e++) acc += h[e];
            if (acc == 0x12345u) thr[i] = 0;
            continue;
        }
        SelectResult res;
        if (!planar_trivial(k, N, res)) {
            if constexpr (E == 16) {
                res = wave_select_hist256_u32(h, N, k, hist, lane, warm, w.band != nullptr ? w.band + 2 * p : nullptr);
            } else {
                res = wave_select_hist_u32<E>(h, N, k, hist, lane, warm);
                band_resolve<E>(h, N, w.band, p, lane, res);
            }
        }
        if constexpr (E == 16) {
            if (w.band != nullptr && res.cut == SELECT_UNRESOLVED) side_hand_over<E>(h, w, p, 0, i, N, res, lane);
        }
        if (lane == 0) {
            thr[i] = res.thr_key;
            cut[i] = res.cut;
        }
        if (w.row_bits && res.cut != SELECT_UNRESOLVED && res.cut != SELECT_HANDED_OVER)
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
    planar_pad_keys<16>(ha, M, lane, M > 15 * 64);
    planar_pad_keys<16>(hb, M, lane, M > 15 * 64);
    if constexpr (MODE == 1) {
        unsigned acc = 0;
#pragma unroll
        for (int e = 0; e < 16; e++) acc += ha[e] + hb[e];
        if (acc == 0x12345u) w.col_cut[0] = 0;
        return;
    }
    unsigned *hist = pcolbuf + (2 * wave) * PL_LDC;      // the wave's two slots: 2 * PL_LDC words >= HIST_WORDS, 16-byte aligned
    hist256_clear(hist, lane);
    const int k = knn_count(k_mode, kv, M);
    const uint64_t valid = planar_slot_valid<16>(M, lane);
    // (float32 keys carry 23 mantissa bits where float64 high words carry 20: the same window of values is 8x as many keys)
    HistWarm warm{0, HIST256_SHIFT0 + (w.band != nullptr ? PL_WARM32 : 0)};
    auto column = [&](const unsigned (&h)[16], const int j) {
        SelectResult res;
        if (MODE == 2) { res.thr_key = ((uint64_t)__builtin_amdgcn_readfirstlane((int)h[1]) << 32) | 0xffffffffull; res.cut = 0x7fffffff; }
        else if (!planar_trivial(k, M, res)) {
            res = wave_select_hist256_u32(h, M, k, hist, lane, warm, w.band != nullptr ? w.band + 2 * p : nullptr);
        }
        if (w.band != nullptr && res.cut == SELECT_UNRESOLVED) side_hand_over<16>(h, w, p, 1, j, M, res, lane);
        if (lane == 0) {
            w.col_thr[(int64_t)p * w.max_n + j] = res.thr_key;
            w.col_cut[(int64_t)p * w.max_n + j] = res.cut;
        }
        if (w.col_bits && res.cut != SELECT_UNRESOLVED && res.cut != SELECT_HANDED_OVER)
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
    HistWarm warm{0, HIST_WARM_SHIFT0 + (w.band != nullptr ? PL_WARM32 : 0)};
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


template <int DIR, int E = 16>
__global__ __launch_bounds__(64) void select_fix_planar_kernel(const uint32_t *__restrict__ Thi, const double *__restrict__ feats,
                                                               const double *__restrict__ norms, int d,
                                                               const acoss_pair_desc *__restrict__ descs, int win,
                                                               double kv, int k_mode, ThreshWork w, int groups, int64_t total)
{
    __shared__ FixSmem sm;
    const int lane = threadIdx.x;
    // behind the side-buffer refinement this kernel only takes what did not fit there: nothing, as a rule
    if (w.side_counter != nullptr && *w.side_counter <= w.side_cap) return;
    for (int64_t blk = blockIdx.x; blk < total; blk += gridDim.x) {
        const int p = (int)(blk / groups), g = (int)(blk % groups);
        const acoss_pair_desc ds = descs[p];
        const int M = ds.nx - win + 1, N = ds.ny - win + 1;
        const int count = DIR == 0 ? M : N;
        const int len = DIR == 0 ? N : M;
        const int t = g * 64 + lane;
        uint64_t *thr = (DIR == 0 ? w.row_thr + (int64_t)p * w.max_m : w.col_thr + (int64_t)p * w.max_n);
        int *cut = (DIR == 0 ? w.row_cut + (int64_t)p * w.max_m : w.col_cut + (int64_t)p * w.max_n);
        unsigned long long todo = __ballot(t < count && cut[t] == SELECT_UNRESOLVED);
        if (todo == 0) continue;
        const int k = knn_count(k_mode, kv, len);
        while (todo) {
            const int which = g * 64 + (__ffsll((long long)todo) - 1);     // wave-uniform
            todo &= todo - 1;
            auto key_at = [&](int q) {
                return Thi[planar_word(ds.crp_off + (DIR == 0 ? (int64_t)which * ds.crp_pitch + q : (int64_t)q * ds.crp_pitch + which))];
            };
            const unsigned thr_hi = (unsigned)(thr[which] >> 32);
            if (w.band != nullptr && fix_row_band<DIR, E>(sm, key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane)) continue;
            fix_row_generic<DIR, E>(key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane);
        }
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
    if (band == nullptr || probe != 0) {       // exact keys: nothing is handed over
        w.side_counter = nullptr;
        w.side_slots = nullptr;
        w.side_keys = nullptr;
        w.side_cap = 0;
    }
    if (K == 0) return ACOSS_OK;
    if (w.side_counter) ACOSS_HIP(hipMemsetAsync(w.side_counter, 0, 256, st));
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
        hipLaunchKernelGGL((select_fix_planar_kernel<0, 32>), dim3((unsigned)((int64_t)K * gm)), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, gm, (int64_t)K * gm);
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
            hipLaunchKernelGGL((select_fix_planar_kernel<1, 32>), dim3((unsigned)((int64_t)K * gn)), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, gn, (int64_t)K * gn);
            rc = launch_check("select_fix_planar_kernel<cols, 32>");
            if (rc) return rc;
        }
        return launch_combine_bits(descs, K, win, mutual, w, bits, st);
    }
#ifdef ACOSS_PROBES
    if (probe == 3) {
        hipLaunchKernelGGL(select_rows_planar_kernel<3>, dim3((unsigned)((int64_t)K * rb)), dim3(256), 0, st, planes, descs, win, kv, mode, w, rb);
        return launch_check("select_rows_planar probe");
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
#else
    if (probe != 0) { set_error("mask_bits_planar: probes need a --probes build"); return ACOSS_ENOTSUP; }
#endif
    if (probe == 0 || probe == 2) {
        hipLaunchKernelGGL(select_rows_planar_kernel<0>, dim3((unsigned)((int64_t)K * rb)), dim3(256), 0, st, planes, descs, win, kv, mode, w, rb);
        int rc = launch_check("select_rows_planar_kernel");
        if (rc) return rc;
        if (probe == 2) return ACOSS_OK;
        if (!w.side_keys) {
            const int groups = ceil_div(max_m, 64);
            hipLaunchKernelGGL(select_fix_planar_kernel<0>, dim3((unsigned)((int64_t)K * groups)), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, groups, (int64_t)K * groups);
            rc = launch_check("select_fix_planar_kernel<rows>");
            if (rc) return rc;
        }
    }
    if (mutual || probe == 12) {
        ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_planar_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(select_cols_planar_kernel<0>, dim3((unsigned)((int64_t)K * cb)), dim3(512), lds, st, planes, descs, win, kv, mode, w, cb);
        int rc = launch_check("select_cols_planar_kernel");
        if (rc) return rc;
        if (probe == 12) return ACOSS_OK;
        if (!w.side_keys) {
            const int groups = ceil_div(max_n, 64);
            hipLaunchKernelGGL(select_fix_planar_kernel<1>, dim3((unsigned)((int64_t)K * groups)), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, groups, (int64_t)K * groups);
            rc = launch_check("select_fix_planar_kernel<cols>");
            if (rc) return rc;
        }
    }
    if (w.side_keys) {
        // undecided rows and columns: one wave per side-buffer slot; what did not fit (never, with a 2 % buffer, on real
        // features) is still marked unresolved and goes the strided way
        hipLaunchKernelGGL(select_fix_side_kernel, dim3((unsigned)w.side_cap), dim3(64), 0, st, feats, norms, d, descs, win, kv, mode, w);
        int rc = launch_check("select_fix_side_kernel");
        if (rc) return rc;
        const int gm = ceil_div(max_m, 64), gn = ceil_div(max_n, 64);
        hipLaunchKernelGGL(select_fix_planar_kernel<0>, dim3(2048), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, gm, (int64_t)K * gm);
        if (mutual) hipLaunchKernelGGL(select_fix_planar_kernel<1>, dim3(2048), dim3(64), 0, st, planes, feats, norms, d, descs, win, kv, mode, w, gn, (int64_t)K * gn);
        rc = launch_check("select_fix_planar_kernel (overflow)");
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

#ifdef ACOSS_PROBES
// development probe (not part of the public ABI): 1 = row loads only, 2 = row selection kernel alone,
// 11 = column loads only, 12 = column selection kernel alone
int acoss_dev_planar_probe(int probe, const uint32_t *planes, const double *feats, const double *norms, int d,
                           const acoss_pair_desc *descs, int K,
                           int win, int max_nx, int max_ny, double kappa, void *work, size_t work_bytes, void *stream)
{
    return run_planar(probe, planes, feats, norms, d, descs, K, win, max_nx, max_ny, kappa, 1, nullptr, work, work_bytes,
                      (hipStream_t)stream);
}
#endif  // ACOSS_PROBES

}  // extern "C"
