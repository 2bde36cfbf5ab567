// keys16_kernels.hip -- row and column kNN selection on the 16-bit key plane (keys16.h) and the float64 refinement of
// what neither the 16-bit keys nor recomputed float32 values can decide.  Outputs: the row / column bit planes of
// ThreshWork (thresh_work.h), which combine_bits_kernel (crp_kernels.hip) turns into the bit-packed mutual mask --
// the same planes, bit for bit, as select_rows_planar_kernel / select_cols_planar_kernel leave.
//
// Data layout in registers: a lane owns 16 CONSECUTIVE positions of its row (column), two keys per register.  A row is then
// read with two 16-byte loads per lane (2 KB per wave, contiguous), the lane's 16 mask bits are exactly one uint16 of the
// row's bit vector (no ballots, no lane transposition: 128 contiguous bytes per row leave the wave), and the per-key work
// runs on packed 16-bit instructions.
#include "radix16.h"

#include <stdlib.h>
#include <string.h>

namespace acoss {

__device__ __attribute__((aligned(32))) const uint32_t k16_pad_block[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                                                            0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};

// positions >= n of a lane's 16 -> K16_PAD (lanes that straddle the end of the row only)
__device__ inline void k16_pad_tail(u16x2 (&h)[8], int n, int lane)
{
    const int nv = n - 16 * lane;          // valid positions of this lane
    if (nv >= 16) return;
#pragma unroll
    for (int v = 0; v < 8; v++) {
        if (2 * v >= nv) h[v] = k16_splat(K16_PAD);
        else if (2 * v + 1 >= nv) h[v].y = (unsigned short)K16_PAD;
    }
}

// the keys a wave could not decide: 2 KB to a side-buffer slot, position order
template <bool OPAQUE_LANE>
__device__ inline bool k16_hand_over(const u16x2 (&h)[8], const ThreshWork &w, int p, int dir, int which, unsigned th, int lane)
{
    if (w.side_keys == nullptr) return false;
    int slot = 0;
    if (lane == 0) slot = atomicAdd(w.side_counter, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= w.side_cap) return false;
    // (row kernel: the lane index behind an opaque move -- hipcc would otherwise form this rare path's lane address at kernel entry and
    //  spill it; in the column kernel the same move makes the allocator spill four key registers on the common path instead: A/B)
    int hl = lane;
    if (OPAQUE_LANE) asm volatile("" : "+v"(hl));
    uint4 *dst = reinterpret_cast<uint4 *>(w.side_keys + (int64_t)slot * 1024) + 2 * hl;
    dst[0] = make_uint4(k16_to_u32(h[0]), k16_to_u32(h[1]), k16_to_u32(h[2]), k16_to_u32(h[3]));
    dst[1] = make_uint4(k16_to_u32(h[4]), k16_to_u32(h[5]), k16_to_u32(h[6]), k16_to_u32(h[7]));
    if (lane == 0) w.side_slots[slot] = make_int4(p, dir, which, (int)th);
    return true;
}

// A pair's rows are dealt evenly over the 4 * rows_blocks waves the grid gives it (rows_blocks covers max_m at up to
// K16_ROWS_PER_WAVE rows per wave): equal wave loads for every song length, and long runs of rows inside one predicted window.
constexpr int K16_ROWS_PER_WAVE = 32;
#ifndef K16_ROWS_WPS
#define K16_ROWS_WPS 7                     // 72 registers: no spills, and the kernel keeps 7.1 of 8 waves resident anyway
#endif

// ---- rows ------------------------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void select_rows_k16_body(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                     int win, double kv, int k_mode, const ThreshWork &w, int rows_blocks, const K16Ctx &cx,
                                                     const int p, const int sub)
{
    __shared__ __attribute__((aligned(16))) unsigned hist_all[4 * K16_HIST_WORDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int rpw = (M + 4 * rows_blocks - 1) / (4 * rows_blocks);
    const int r0 = (sub * 4 + wave) * rpw;
    if (r0 >= M) return;
    const int r1 = min(r0 + rpw, M);
    const int lane = threadIdx.x & 63;
    const int k = knn_count(k_mode, kv, N);
    unsigned *hist = hist_all + wave * K16_HIST_WORDS;
    hist256_clear(hist, lane);
    hist[HIST256_BINS + 64 + 192 + lane] = 0u;
    HistWarm warm{0, K16_SHIFT0};
    const unsigned koff = cx.koff[p];
    const float *pair_band = w.band + 2 * p;
    const bool adjacent_ok = k16_reach_adjacent_ok(koff, pair_band);
    // lane l reads bytes [32 l, 32 l + 32) of the row; lanes past the end read a block of padding keys instead (a per-lane
    // pointer and step: nothing about the addresses is recomputed per row)
    const int n_lanes = (N + 15) >> 4;
    // the lane that straddles the end of a row finds K16_PAD behind it: acoss_crp_keys16_batch pads every row to a multiple of 16
    // columns (strip32_kernels.hip) -- unless the pitch leaves no room for that
    const bool tail = (N & 15) != 0 && ds.crp_pitch < ((N + 15) & ~15);            // wave-uniform
    const bool inside = lane < n_lanes;
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    const char *rp = inside ? reinterpret_cast<const char *>(keys + ds.crp_off + 16 * lane + (int64_t)r0 * ds.crp_pitch)
                            : reinterpret_cast<const char *>(k16_pad_block);
    const int64_t rstep = inside ? 2 * (int64_t)ds.crp_pitch : 0;
    u32x4v na = __builtin_nontemporal_load(reinterpret_cast<const u32x4v *>(rp)), nb = __builtin_nontemporal_load(reinterpret_cast<const u32x4v *>(rp) + 1);
    // the row's 16 mask bits per lane leave through a raw buffer store: wave-uniform row offset + a constant lane offset
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(w.row_bits + ((int64_t)p * w.max_m) * 16, 0, M * 128, 0x00020000);
    int cut = 0;                                 // lane t: row_cut of row r0 + t (rpw <= K16_ROWS_PER_WAVE <= 64), stored once
    K16Probe pr;                                 // phases: 0 waiting for the row, 1 histogram pass, 2 rest of the selection, 3 decision, 4 stores
    pr.start((cx.flags & 4) != 0);
    for (int i = r0; i < r1; i++) {
#ifdef ACOSS_PROBES
        if (cx.flags & 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pr.lap(0); }
#endif
        u16x2 h[8] = {k16_from_u32(na.x), k16_from_u32(na.y), k16_from_u32(na.z), k16_from_u32(na.w),
                      k16_from_u32(nb.x), k16_from_u32(nb.y), k16_from_u32(nb.z), k16_from_u32(nb.w)};
        if (i + 1 < r1) {
            rp += rstep;
            na = __builtin_nontemporal_load(reinterpret_cast<const u32x4v *>(rp));
            nb = __builtin_nontemporal_load(reinterpret_cast<const u32x4v *>(rp) + 1);
        }
        if (tail) {
            // (the lane index behind an opaque move: hipcc would otherwise hoist this rare path's sixteen lane conditions out of the
            //  row loop into SGPR pairs, and spill other scalars of the loop to make room)
            int tl = lane;
            asm volatile("" : "+v"(tl));
            k16_pad_tail(h, N, tl);
        }
        unsigned sel = 0;
        int state = K16_DECIDED;
        if (k <= 0) sel = 0u;
        else if (k >= N) {
            const int nv = N - 16 * lane;
            sel = nv >= 16 ? 0xFFFFu : (nv <= 0 ? 0u : ((1u << nv) - 1u));
        } else {
            const Sel16 s = wave_select_k16(h, k, hist, lane, warm, cx.stats, pr);
            pr.lap(2);
            state = s.ok ? k16_decide<D, 0>(h, s, k, hist + HIST256_BINS + 64, lane, cx, pair_band, koff, adjacent_ok, ds, p, i, sel) : K16_HANDOVER;
            pr.lap(3);
            if (state == K16_HANDOVER) {
                if (!k16_hand_over<true>(h, w, p, 0, i, s.ok ? s.th : 0u, lane)) {
                    // no room in the side buffer: marked for the strided refinement kernel
                    if (lane == 0) w.row_thr[(int64_t)p * w.max_m + i] = (uint64_t)(s.ok ? s.th : 0u) << 32;
                    cut = lane == i - r0 ? SELECT_UNRESOLVED : cut;
                    continue;
                }
            }
        }
        cut = lane == i - r0 ? (state == K16_DECIDED ? 0x7fffffff : -3) : cut;
        if (state == K16_DECIDED) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)sel, brsrc, 2 * lane, i * 128, 0);
        pr.lap(4);
        pr.count();
    }
    pr.flush(cx.stats, lane);
    if (lane < r1 - r0) w.row_cut[(int64_t)p * w.max_m + r0 + lane] = cut;
    // a hint for the column kernel, which starts every wave cold: the last threshold key of the pair's first rows (the low
    // word of a slot that only an unresolved column 0 ever uses, and then with a zero low word = no hint)
    if (lane == 0 && r0 == 0 && warm.hi != 0u) reinterpret_cast<unsigned *>(w.col_thr + (int64_t)p * w.max_n)[0] = warm.hi;
}


template <int D>
__global__ __launch_bounds__(256, K16_ROWS_WPS) void select_rows_k16_kernel(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                                 int win, double kv, int k_mode, ThreshWork w, int rows_blocks, K16Ctx cx)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    select_rows_k16_body<D>(keys, descs, win, kv, k_mode, w, rows_blocks, cx, lb / rows_blocks, lb % rows_blocks);
}

// The same for the pairs of a list (round 5: the pairs the radix selection hands back -- exact ties); the grid covers `slots` pairs at a
// time.  list_n: a device int.
template <int D>
__global__ __launch_bounds__(256, K16_ROWS_WPS) void select_rows_k16_list_kernel(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                                 int win, double kv, int k_mode, ThreshWork w, int rows_blocks, K16Ctx cx,
                                                                 const int *__restrict__ list, const int *__restrict__ list_n, int slots)
{
    const int n = *list_n;
    for (int s = blockIdx.x / rows_blocks; s < n; s += slots) {
        select_rows_k16_body<D>(keys, descs, win, kv, k_mode, w, rows_blocks, cx, list[s], blockIdx.x % rows_blocks);
        __syncthreads();
    }
}

// ---- columns ---------------------------------------------------------------------------------------------------------------
// One 8-wave block stages 32 columns (64-byte row segments) through LDS in two halves of 512 rows; wave v then selects in
// columns 4v .. 4v+3, each inside the window predicted by the one before.  A thread loads 8 bytes (four columns) of two
// adjacent rows and packs them into one word per column; a wave's loads cover 16 consecutive rows.  Staged column c keeps the
// packed rows (16 l + 2 u, 16 l + 2 u + 1) of lane l at word 33 u + (l & 31): reads (u fixed, lanes consecutive) and
// writes (column stride 265 words = 9 banks ... 1 mod 8 with the four columns of a thread, u = 0..3 across the 32 lanes of a
// write) are conflict-free.
#ifndef K16_COL_WAVES
#define K16_COL_WAVES 8                   // waves per block, four columns each
#endif
constexpr int K16_COLS = 4 * K16_COL_WAVES;
#ifndef K16_COL_OPAQUE
#define K16_COL_OPAQUE 0
#endif
#ifndef K16_COL_SEED
#define K16_COL_SEED 1
#endif
#ifndef K16_COL_SEED_WIDEN
#define K16_COL_SEED_WIDEN 2              // log2: the first window of a wave is this much wider than K16_SHIFT0's
#endif
constexpr int K16_LDC = 265;            // words per staged half column: 8 x 33 + 1

template <int D>
__device__ __forceinline__ void select_cols_k16_body(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                     int win, double kv, int k_mode, const ThreshWork &w, int col_blocks, const K16Ctx &cx,
                                                     const int p, const int sub)
{
    __shared__ __attribute__((aligned(16))) unsigned colbuf[K16_COLS * K16_LDC + 8];
    __shared__ __attribute__((aligned(16))) unsigned hist_all[K16_COL_WAVES * K16_HIST_WORDS];
    const int j0 = sub * K16_COLS;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (j0 >= N) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    K16Probe pr;                                 // phases: 0 staging, 1 histogram pass, 2 rest of the selection, 3 decision, 4 stores
    pr.start((cx.flags & 4) != 0);
    u16x2 hc[4][8];
    {
        // thread (c2, rp): columns 4 c2 .. 4 c2 + 3 of the row pair 64 s + rp (rows 128 s + 2 rp, + 1) for s = 0..7; that pair
        // belongs to lane 8 s + (rp >> 3) of the selecting waves, as its word u = rp & 7
        const int c2 = threadIdx.x & (K16_COL_WAVES - 1), rp = threadIdx.x / K16_COL_WAVES;
        const bool fast = ((ds.crp_pitch & 3) == 0) && ((ds.crp_off & 3) == 0) && (j0 + K16_COLS <= N);      // block-uniform: 8-byte loads
        uint2 ta[8], tb[8];
        if (fast) {
            const uint16_t *pb = keys + ds.crp_off + j0 + 4 * c2;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const int ra = min(128 * s + 2 * rp, M - 1), rb = min(128 * s + 2 * rp + 1, M - 1);
                ta[s] = *reinterpret_cast<const uint2 *>(pb + (unsigned)(ra * ds.crp_pitch));
                tb[s] = *reinterpret_cast<const uint2 *>(pb + (unsigned)(rb * ds.crp_pitch));
            }
        } else {
            int cc = c2, rc = rp;
            asm volatile("" : "+v"(cc), "+v"(rc));
#pragma unroll
            for (int s = 0; s < 8; s++) {
                unsigned short e[2][4];
#pragma unroll
                for (int z = 0; z < 2; z++) {
                    const int64_t ri = ds.crp_off + (int64_t)min(128 * s + 2 * rc + z, M - 1) * ds.crp_pitch;
#pragma unroll
                    for (int u = 0; u < 4; u++) e[z][u] = keys[ri + min(j0 + 4 * cc + u, N - 1)];
                }
                ta[s] = make_uint2((unsigned)e[0][0] | ((unsigned)e[0][1] << 16), (unsigned)e[0][2] | ((unsigned)e[0][3] << 16));
                tb[s] = make_uint2((unsigned)e[1][0] | ((unsigned)e[1][1] << 16), (unsigned)e[1][2] | ((unsigned)e[1][3] << 16));
            }
        }
        // rows past the end of the column become padding (the clamped loads above repeat the last row)
#pragma unroll
        for (int s = 0; s < 8; s++) {
            if (128 * s + 2 * rp >= M) ta[s] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            if (128 * s + 2 * rp + 1 >= M) tb[s] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        }
        unsigned *dst = colbuf + (4 * c2) * K16_LDC + 33 * (rp & 7) + (rp >> 3);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (half) __syncthreads();              // every wave has read the first half
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const int s = 4 * half + s4;        // lanes 8 s .. 8 s + 7: slots 8 s4 .. of this half
                dst[0 * K16_LDC + 8 * s4] = (ta[s].x & 0xFFFFu) | (tb[s].x << 16);
                dst[1 * K16_LDC + 8 * s4] = (ta[s].x >> 16) | (tb[s].x & 0xFFFF0000u);
                dst[2 * K16_LDC + 8 * s4] = (ta[s].y & 0xFFFFu) | (tb[s].y << 16);
                dst[3 * K16_LDC + 8 * s4] = (ta[s].y >> 16) | (tb[s].y & 0xFFFF0000u);
            }
            __syncthreads();
            if ((lane >> 5) == half) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
#pragma unroll
                    for (int u = 0; u < 8; u++) hc[c][u] = k16_from_u32(colbuf[(4 * wave + c) * K16_LDC + 33 * u + (lane & 31)]);
                }
            }
        }
    }
    pr.lap(0);
    const int ja = j0 + 4 * wave;
    if (ja >= N) return;
    unsigned *hist = hist_all + wave * K16_HIST_WORDS;
    hist256_clear(hist, lane);
    hist[HIST256_BINS + 64 + 192 + lane] = 0u;
    const int k = knn_count(k_mode, kv, M);
    // first window: around the row kernel's hint, wider than between neighbouring columns
    HistWarm warm{K16_COL_SEED ? reinterpret_cast<const unsigned *>(w.col_thr + (int64_t)p * w.max_n)[0] : 0u, K16_SHIFT0 + K16_COL_SEED_WIDEN};
    if (warm.hi >= K16_MAX) warm.hi = 0u;
    const unsigned koff = cx.koff[p];
    const float *pair_band = w.band + 2 * p;
    const bool adjacent_ok = k16_reach_adjacent_ok(koff, pair_band);
    auto column = [&](const u16x2 (&h)[8], const int j) {
        unsigned sel = 0;
        int state = K16_DECIDED;
#ifdef ACOSS_PROBES
        if (cx.flags & 2) { reinterpret_cast<uint16_t *>(w.col_word(p, j, lane >> 2))[lane & 3] = (uint16_t)(h[0].x & 1); return; }
#endif
        if (k <= 0) sel = 0u;
        else if (k >= M) {
            const int nv = M - 16 * lane;
            sel = nv >= 16 ? 0xFFFFu : (nv <= 0 ? 0u : ((1u << nv) - 1u));
        } else {
            const Sel16 s = wave_select_k16(h, k, hist, lane, warm, cx.stats, pr);
            pr.lap(2);
            state = s.ok ? k16_decide<D, 1>(h, s, k, hist + HIST256_BINS + 64, lane, cx, pair_band, koff, adjacent_ok, ds, p, j, sel) : K16_HANDOVER;
            pr.lap(3);
            if (state == K16_HANDOVER) {
                if (!k16_hand_over<(K16_COL_OPAQUE != 0)>(h, w, p, 1, j, s.ok ? s.th : 0u, lane)) {
                    if (lane == 0) {
                        w.col_thr[(int64_t)p * w.max_n + j] = (uint64_t)(s.ok ? s.th : 0u) << 32;
                        w.col_cut[(int64_t)p * w.max_n + j] = SELECT_UNRESOLVED;
                    }
                    return;
                }
            }
        }
        if (lane == 0) w.col_cut[(int64_t)p * w.max_n + j] = state == K16_DECIDED ? 0x7fffffff : -3;
        // rows 16 l .. 16 l + 15 of column j = quarter (l & 3) of word l >> 2 of the column's bit vector
        if (state == K16_DECIDED) reinterpret_cast<uint16_t *>(w.col_word(p, j, lane >> 2))[lane & 3] = (uint16_t)sel;
        pr.lap(4);
        pr.count();
    };
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (ja + c < N) column(hc[c], ja + c);
        if (c == 0) warm.shift = K16_SHIFT0;       // neighbouring columns predict each other closely
    }
    pr.flush(cx.stats, lane);
}


template <int D>
__global__ __launch_bounds__(64 * K16_COL_WAVES, 6) void select_cols_k16_kernel(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                                 int win, double kv, int k_mode, ThreshWork w, int col_blocks, K16Ctx cx)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    select_cols_k16_body<D>(keys, descs, win, kv, k_mode, w, col_blocks, cx, lb / col_blocks, lb % col_blocks);
}

template <int D>
__global__ __launch_bounds__(64 * K16_COL_WAVES, 6) void select_cols_k16_list_kernel(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                                 int win, double kv, int k_mode, ThreshWork w, int col_blocks, K16Ctx cx,
                                                                 const int *__restrict__ list, const int *__restrict__ list_n, int slots)
{
    const int n = *list_n;
    for (int s = blockIdx.x / col_blocks; s < n; s += slots) {
        select_cols_k16_body<D>(keys, descs, win, kv, k_mode, w, col_blocks, cx, list[s], blockIdx.x % col_blocks);
        __syncthreads();
    }
}

// ---- refinement ------------------------------------------------------------------------------------------------------------
// One wave per side-buffer slot: exact float64 values of every cell whose 16-bit key the winner's error band can reach
// (fix_row_band; the general bit-serial selection when more than 64 cells are in reach), cells with smaller keys are selected.
template <typename FT>
__device__ __forceinline__ void select_fix_side16_body(const FT *__restrict__ feats, const FT *__restrict__ norms, int d,
                                                       const acoss_pair_desc *__restrict__ descs, int win, double kv, int k_mode,
                                                       const ThreshWork &w, const uint32_t *__restrict__ koff, const int slot)
{
    __shared__ FixSmem sm;
    const int4 rec = w.side_slots[slot];
    const int p = rec.x, dir = rec.y, which = rec.z;
    const unsigned th = (unsigned)rec.w;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int len = dir ? M : N;
    const int lane = threadIdx.x;
    const int k = knn_count(k_mode, kv, len);
    const uint16_t *keys = reinterpret_cast<const uint16_t *>(w.side_keys + (int64_t)slot * 1024);
    auto key_at = [&](int q) { return (unsigned)keys[q]; };
    uint64_t *thr = dir ? w.col_thr + (int64_t)p * w.max_n : w.row_thr + (int64_t)p * w.max_m;
    int *cut = dir ? w.col_cut + (int64_t)p * w.max_n : w.row_cut + (int64_t)p * w.max_m;
    unsigned h_lo, h_hi;
    k16_reach(th, koff[p], k16_reach_adjacent_ok(koff[p], w.band + 2 * p), w.band + 2 * p, h_lo, h_hi);
    if (dir == 0) {
        if (fix_row_band_range<0, 16>(sm, key_at, h_lo, h_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane)) return;
        fix_row_generic_range<0, 16>(key_at, h_lo, h_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane);
    } else {
        if (fix_row_band_range<1, 16>(sm, key_at, h_lo, h_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane)) return;
        fix_row_generic_range<1, 16>(key_at, h_lo, h_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane);
    }
}


template <typename FT>
__global__ __launch_bounds__(64) void select_fix_side16_kernel(const FT *__restrict__ feats, const FT *__restrict__ norms, int d,
                                                               const acoss_pair_desc *__restrict__ descs, int win, double kv, int k_mode,
                                                               ThreshWork w, const uint32_t *__restrict__ koff)
{
    // (one slot per block when the grid is the slot capacity; a small grid walks the slots: the list form of the selection)
    const int n = min(*w.side_counter, w.side_cap);
    for (int slot = blockIdx.x; slot < n; slot += gridDim.x) {
        select_fix_side16_body<FT>(feats, norms, d, descs, win, kv, k_mode, w, koff, slot);
        __syncthreads();
    }
}

// rows / columns that found no room in the side buffer (never, with its 2 % capacity, on real features): the same refinement
// from the key plane itself
template <int DIR, typename FT>
__global__ __launch_bounds__(64) void select_fix_k16_kernel(const uint16_t *__restrict__ keys16, const FT *__restrict__ feats,
                                                            const FT *__restrict__ norms, int d, const acoss_pair_desc *__restrict__ descs,
                                                            int win, double kv, int k_mode, ThreshWork w, const uint32_t *__restrict__ koff,
                                                            int groups, int64_t total, const int *__restrict__ list, const int *__restrict__ list_n)
{
    __shared__ FixSmem sm;
    const int lane = threadIdx.x;
    if (w.side_counter != nullptr && *w.side_counter <= w.side_cap) return;
    if (list != nullptr) total = (int64_t)(*list_n) * groups;
    for (int64_t blk = blockIdx.x; blk < total; blk += gridDim.x) {
        const int p = list != nullptr ? list[blk / groups] : (int)(blk / groups), g = (int)(blk % groups);
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
            const int which = g * 64 + (__ffsll((long long)todo) - 1);
            todo &= todo - 1;
            auto key_at = [&](int q) {
                return (unsigned)keys16[ds.crp_off + (DIR == 0 ? (int64_t)which * ds.crp_pitch + q : (int64_t)q * ds.crp_pitch + which)];
            };
            unsigned h_lo, h_hi;
            k16_reach((unsigned)(thr[which] >> 32), koff[p], k16_reach_adjacent_ok(koff[p], w.band + 2 * p), w.band + 2 * p, h_lo, h_hi);
            if (fix_row_band_range<DIR, 16>(sm, key_at, h_lo, h_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane)) continue;
            fix_row_generic_range<DIR, 16>(key_at, h_lo, h_hi, feats, norms, d, ds, win, w, p, which, len, k, thr, cut, lane);
        }
    }
}

// ---- key offsets for float32 corpora ---------------------------------------------------------------------------------------------
// For a float64 corpus koff follows from the window norm sums W of the CENTRED copy the filter runs on (engine.keys16_koff): no
// windowed sum exceeds 2 W and thresholds sit 1.5-6 octaves below it.  A float32 corpus is its own filter operand (no centring:
// acoss_mask_bits_keys16_f32_batch), and features with a common offset -- real MFCC: the energy-like coefficient -- have raw norm
// sums far above every distance: a key range hung on them would lie above all thresholds.  But |x - y|^2 <= 2 (|x - m|^2 +
// |y - m|^2) for ANY vector m, so the range hangs on norms centred PER PAIR, m = the midpoint of the two songs' mean frames
// (pairs of songs that share an offset are covered too).  One block per pair: mean frames, centred squared norms into LDS, the
// largest 9-frame window sum of each song -> top = 2 (Wx + Wy) (+ the float32 noise of the raw operands, 2^-17 of the raw norm
// sums: a value that clamps at the top all the same is merely handed over), koff = the pattern of top 2^-7.
template <int D>
__global__ __launch_bounds__(256) void k16_koff_pair_kernel(const float *__restrict__ xp, int max_nx, const float *__restrict__ f32,
                                                            const float *__restrict__ n32, const acoss_pair_desc *__restrict__ descs,
                                                            int win, uint32_t *__restrict__ koff)
{
    __shared__ float mean[2][16];
    __shared__ float cn[2][2064 + 16];             // centred squared norms of the x and y frames
    __shared__ unsigned wmax[4];                   // [song] centred, [2 + song] raw: window-sum maxima as float bit patterns (>= 0)
    const int p = blockIdx.x;
    const acoss_pair_desc ds = descs[p];
    const int tid = threadIdx.x;
    if (ds.nx > 2064 || ds.ny > 2064) { if (tid == 0) koff[p] = 0u; return; }      // (block-uniform; the 16-bit path stops at 2056 frames)
    if (tid < 32) mean[tid >> 4][tid & 15] = 0.0f;
    if (tid < 4) wmax[tid] = 0u;
    __syncthreads();
    const float *xs = xp + (int64_t)p * max_nx * 16;                 // x frames rotated by the OTI: 16 floats each
    const float *ys = f32 + ds.y_row0 * D;
    auto frame = [&](int song, int f, int b) { return song == 0 ? xs[(int64_t)f * 16 + b] : ys[(int64_t)f * D + b]; };
    for (int song = 0; song < 2; song++) {
        const int n = song == 0 ? ds.nx : ds.ny;
        float acc[D];
#pragma unroll
        for (int b = 0; b < D; b++) acc[b] = 0.0f;
        for (int f = tid; f < n; f += 256) {
#pragma unroll
            for (int b = 0; b < D; b++) acc[b] += frame(song, f, b);
        }
#pragma unroll
        for (int b = 0; b < D; b++) atomicAdd(&mean[song][b], acc[b] / (float)n);
    }
    __syncthreads();
    float m[D];
#pragma unroll
    for (int b = 0; b < D; b++) m[b] = 0.5f * (mean[0][b] + mean[1][b]);
    for (int song = 0; song < 2; song++) {
        const int n = song == 0 ? ds.nx : ds.ny;
        for (int f = tid; f < n; f += 256) {
            float q = 0.0f;
#pragma unroll
            for (int b = 0; b < D; b++) {
                const float v = frame(song, f, b) - m[b];
                q = fmaf(v, v, q);
            }
            cn[song][f] = q;
        }
    }
    __syncthreads();
    for (int song = 0; song < 2; song++) {
        const int n = song == 0 ? ds.nx : ds.ny;
        const float *raw = song == 0 ? nullptr : n32 + ds.y_row0;
        float best = 0.0f, best_raw = 0.0f;
        for (int i = tid; i + win <= n; i += 256) {
            float w = 0.0f, wr = 0.0f;
            for (int k = 0; k < win; k++) {
                w += cn[song][i + k];
                wr += song == 0 ? xs[(int64_t)(i + k) * 16 + D] : raw[i + k];
            }
            best = fmaxf(best, w);
            best_raw = fmaxf(best_raw, wr);
        }
        if (best == best) atomicMax(&wmax[song], __float_as_uint(fmaxf(best, 0.0f)));
        else atomicMax(&wmax[song], 0x7f800000u);                    // a NaN anywhere: no usable range
        atomicMax(&wmax[2 + song], best_raw == best_raw ? __float_as_uint(fmaxf(best_raw, 0.0f)) : 0x7f800000u);
    }
    __syncthreads();
    if (tid == 0) {
        const float wc = __uint_as_float(wmax[0]) + __uint_as_float(wmax[1]), wr = __uint_as_float(wmax[2]) + __uint_as_float(wmax[3]);
        const float top = fmaf(0x1p-17f, wr, 2.0f * wc * (1.0f + 0x1p-10f));
        koff[p] = (top > 0x1p-100f && top < INFINITY) ? __float_as_uint(top) + 1u - (7u << 23) : 0u;
    }
}

// defined in crp_kernels.hip
int launch_combine_bits(const acoss_pair_desc *descs, int K, int win, int mutual, ThreshWork w, uint64_t *bits, hipStream_t st);
int launch_combine_bits_list(const acoss_pair_desc *descs, int K, int win, int mutual, ThreshWork w, uint64_t *bits, const int *list,
                             const int *list_n, int slots, hipStream_t st);

// ACOSS_RADIX16 (default on; "0", "false", "no", "" switch it off): the radix selection of radix16_kernels.hip in front, the
// wave-per-row kernels of this file only for the pairs it hands back.  Read at every call.
static bool radix16_enabled()
{
    const char *e = getenv("ACOSS_RADIX16");
    if (e == nullptr) return true;
    return !(e[0] == 0 || strcmp(e, "0") == 0 || strcmp(e, "false") == 0 || strcmp(e, "no") == 0);
}
constexpr int K16_LIST_SLOTS = 128;        // pairs the list kernels cover at a time (their blocks find the list empty on the benchmark's data;
                                           // corpora with repeated frames hand back a fifth of their pairs)

}  // namespace acoss

using namespace acoss;

// FT = the type of the corpus whose EXACT windowed sums the masks are defined by (planar_select.h: exact_term): double for
// float64 features, float for float32 features (then feats / norms are the very arrays the filter was computed from)
template <typename FT>
static int mask_bits_keys16_impl(const uint16_t *keys16, const float *band, const uint32_t *koff, const float *xp,
                                 const float *f32, const float *n32, const FT *feats, const FT *norms, int d,
                                 const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, double kappa,
                                 int mutual, uint64_t *bits, void *work, size_t work_bytes, void *stream)
{
    if (!keys16 || !band || !koff || !xp || !f32 || !n32 || !feats || !norms || !descs || !bits || !work || K < 0 || max_nx < win ||
        max_ny < win || kappa < 0.0) {
        set_error("mask_bits_keys16_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if ((d != 12 && d != 13) || win != 9) { set_error("mask_bits_keys16_batch: supports d in {12, 13} and win == 9"); return ACOSS_ENOTSUP; }
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    if (max_m > 2048 || max_n > 2048) { set_error("mask_bits_keys16_batch: matrices up to 2048 x 2048"); return ACOSS_ENOTSUP; }
    const bool lng = max_m > 1024 || max_n > 1024;
    if (lng && !((mutual == 0 || mutual == 1) && radix16_enabled())) {
        set_error("mask_bits_keys16_batch: matrices beyond 1024 x 1024 need the radix selection (ACOSS_RADIX16)");
        return ACOSS_ENOTSUP;
    }
    if (work_bytes < thresh_work_bytes(K, max_m, max_n, true)) { set_error("mask_bits_keys16_batch: workspace too small"); return ACOSS_EINVAL; }
    if (K == 0) return ACOSS_OK;
    hipStream_t st = (hipStream_t)stream;
    ThreshWork w = thresh_work_layout(work, K, max_m, max_n, true);
    if (lng) {
        // ---- the long form (a side of 1025 .. 2048): the radix selection alone.  Pairs it cannot express stay UNRESOLVED (their
        // mask rows are undefined): acoss_mask_bits_keys16_unresolved() lists them, the caller redoes them on the float64 path
        // (acoss_crp_planar_batch_f64 + acoss_mask_bits_planar_batch; acoss_serra09_scores does)
        return r16_run<FT>(7, keys16, band, koff, feats, norms, d, descs, K, win, max_nx, max_ny, kappa, mutual, bits, w.radix, st);
    }
    w.band = band;
    ACOSS_HIP(hipMemsetAsync(w.side_counter, 0, 256, st));
    double kv;
    int mode;
    if (kappa == 0.0) { kv = 0.0; mode = 2; } else if (kappa < 1.0) { kv = kappa; mode = 0; } else { kv = kappa; mode = 1; }
    K16Ctx cx;
    cx.xp = xp; cx.max_nx = max_nx; cx.f32 = f32; cx.n32 = n32; cx.koff = koff;
    cx.stats = nullptr;
    cx.flags = 0;
#ifdef ACOSS_PROBES
    if (getenv("ACOSS_K16_STATS")) cx.stats = w.side_counter;
    if (getenv("ACOSS_K16_FLAGS")) cx.flags = atoi(getenv("ACOSS_K16_FLAGS"));
#endif
    const int rb = ceil_div(max_m, 4 * K16_ROWS_PER_WAVE), cb = ceil_div(max_n, K16_COLS);
    if ((int64_t)K * rb > 0x7fffffffLL || (int64_t)K * cb > 0x7fffffffLL) { set_error("mask_bits_keys16_batch: batch too large"); return ACOSS_ENOTSUP; }
    int rc = ACOSS_OK;
    auto rows = [&](unsigned blocks, hipStream_t s) {
        if (d == 12) hipLaunchKernelGGL(select_rows_k16_kernel<12>, dim3(blocks), dim3(256), 0, s, keys16, descs, win, kv, mode, w, rb, cx);
        else hipLaunchKernelGGL(select_rows_k16_kernel<13>, dim3(blocks), dim3(256), 0, s, keys16, descs, win, kv, mode, w, rb, cx);
        return launch_check("select_rows_k16_kernel");
    };
    auto cols = [&](hipStream_t s) {
        if (d == 12) hipLaunchKernelGGL(select_cols_k16_kernel<12>, dim3((unsigned)((int64_t)K * cb)), dim3(64 * K16_COL_WAVES), 0, s, keys16, descs, win, kv, mode, w, cb, cx);
        else hipLaunchKernelGGL(select_cols_k16_kernel<13>, dim3((unsigned)((int64_t)K * cb)), dim3(64 * K16_COL_WAVES), 0, s, keys16, descs, win, kv, mode, w, cb, cx);
        return launch_check("select_cols_k16_kernel");
    };
    if ((mutual == 0 || mutual == 1) && w.radix != nullptr && radix16_enabled()) {
        // ---- round 5: radix selection (keys in registers, two passes), exact values for the few cells in reach, and the
        // wave-per-row kernels below only for the pairs it cannot express (exact ties): their list stays on the device
        rc = r16_run<FT>(7, keys16, band, koff, feats, norms, d, descs, K, win, max_nx, max_ny, kappa, mutual, bits, w.radix, st);
        if (rc) return rc;
        const R16Work rw = r16_work_layout(w.radix, K, max_m, max_n);
        const int slots = K < K16_LIST_SLOTS ? K : K16_LIST_SLOTS;
        const int *list = rw.pair_list, *list_n = rw.counters + 2;
        if (d == 12) hipLaunchKernelGGL(select_rows_k16_list_kernel<12>, dim3((unsigned)(slots * rb)), dim3(256), 0, st, keys16, descs, win, kv, mode, w, rb, cx, list, list_n, slots);
        else hipLaunchKernelGGL(select_rows_k16_list_kernel<13>, dim3((unsigned)(slots * rb)), dim3(256), 0, st, keys16, descs, win, kv, mode, w, rb, cx, list, list_n, slots);
        if (mutual) {
            if (d == 12) hipLaunchKernelGGL(select_cols_k16_list_kernel<12>, dim3((unsigned)(slots * cb)), dim3(64 * K16_COL_WAVES), 0, st, keys16, descs, win, kv, mode, w, cb, cx, list, list_n, slots);
            else hipLaunchKernelGGL(select_cols_k16_list_kernel<13>, dim3((unsigned)(slots * cb)), dim3(64 * K16_COL_WAVES), 0, st, keys16, descs, win, kv, mode, w, cb, cx, list, list_n, slots);
        }
        hipLaunchKernelGGL(select_fix_side16_kernel<FT>, dim3(2048), dim3(64), 0, st, feats, norms, d, descs, win, kv, mode, w, koff);      // (grid-stride over the slots asked for)
        const int gm2 = ceil_div(max_m, 64), gn2 = ceil_div(max_n, 64);
        hipLaunchKernelGGL((select_fix_k16_kernel<0, FT>), dim3(256), dim3(64), 0, st, keys16, feats, norms, d, descs, win, kv, mode, w, koff, gm2, (int64_t)K * gm2, list, list_n);
        if (mutual) hipLaunchKernelGGL((select_fix_k16_kernel<1, FT>), dim3(256), dim3(64), 0, st, keys16, feats, norms, d, descs, win, kv, mode, w, koff, gn2, (int64_t)K * gn2, list, list_n);
        rc = launch_check("mask_bits_keys16_batch: list kernels");
        if (rc) return rc;
        return launch_combine_bits_list(descs, K, win, mutual, w, bits, list, list_n, slots, st);
    }
    if (mutual != 3) {                          // (3 = measurement: the column selection kernel alone)
        rc = rows((unsigned)((int64_t)K * rb), st);
        if (rc) return rc;
    }
    if (mutual == 2) return ACOSS_OK;           // measurement: the row selection kernel alone (bench.py's roofline_selection)
    if (mutual) {
        rc = cols(st);
        if (rc) return rc;
        if (mutual == 3) return ACOSS_OK;
    }
    hipLaunchKernelGGL(select_fix_side16_kernel<FT>, dim3((unsigned)w.side_cap), dim3(64), 0, st, feats, norms, d, descs, win, kv, mode, w, koff);
    rc = launch_check("select_fix_side16_kernel");
    if (rc) return rc;
    const int gm = ceil_div(max_m, 64), gn = ceil_div(max_n, 64);
    hipLaunchKernelGGL((select_fix_k16_kernel<0, FT>), dim3(2048), dim3(64), 0, st, keys16, feats, norms, d, descs, win, kv, mode, w, koff, gm, (int64_t)K * gm, (const int *)nullptr, (const int *)nullptr);
    if (mutual) hipLaunchKernelGGL((select_fix_k16_kernel<1, FT>), dim3(2048), dim3(64), 0, st, keys16, feats, norms, d, descs, win, kv, mode, w, koff, gn, (int64_t)K * gn, (const int *)nullptr, (const int *)nullptr);
    rc = launch_check("select_fix_k16_kernel (overflow)");
    if (rc) return rc;
    return launch_combine_bits(descs, K, win, mutual, w, bits, st);
}

extern "C" int acoss_mask_bits_keys16_batch(const uint16_t *keys16, const float *band, const uint32_t *koff, const float *xp,
                                            const float *f32, const float *n32, const double *feats, const double *norms, int d,
                                            const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, double kappa,
                                            int mutual, uint64_t *bits, void *work, size_t work_bytes, void *stream)
{
    return mask_bits_keys16_impl<double>(keys16, band, koff, xp, f32, n32, feats, norms, d, descs, K, win, max_nx, max_ny, kappa, mutual,
                                         bits, work, work_bytes, stream);
}

// The same for a corpus of float32 features (the reference's mfcc_htk / hpcp: CRPUtils.py:82, :40-41): the masks equal those of
// acoss_crp_batch_f32 + acoss_mask_bits_batch.  f32 / n32 are the corpus itself and its norms (acoss_frame_norms_f32) -- not a
// centred copy: the filter's cross-similarity values are then bit for bit the exact path's, and only the root-square and the
// float32 window sum separate the two (|T~ - T| <= 7 u T: inside the band of acoss_mask_bits_planar32_batch).
extern "C" int acoss_mask_bits_keys16_f32_batch(const uint16_t *keys16, const float *band, const uint32_t *koff, const float *xp,
                                                const float *f32, const float *n32, int d, const acoss_pair_desc *descs, int K, int win,
                                                int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits, void *work,
                                                size_t work_bytes, void *stream)
{
    return mask_bits_keys16_impl<float>(keys16, band, koff, xp, f32, n32, f32, n32, d, descs, K, win, max_nx, max_ny, kappa, mutual, bits,
                                        work, work_bytes, stream);
}

// What the last acoss_mask_bits_keys16(_f32)_batch call on this workspace did (synchronises with the device; tests and tools):
// out[0] work items beyond the tiles' own slots, out[1] rows + columns that flagged their pair, out[2] pairs handed to the
// wave-per-row kernels, out[3] reserved; out[8 .. 19]: why lines flagged their pair (radix16_kernels.hip).  out: 20 ints.
extern "C" int acoss_mask_bits_keys16_stats(void *work, int K, int max_nx, int max_ny, int win, int *out)
{
    if (!work || !out || K <= 0) { set_error("mask_bits_keys16_stats: bad argument"); return ACOSS_EINVAL; }
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    const ThreshWork w = thresh_work_layout(work, K, max_m, max_n, true);
    if (w.radix == nullptr) { set_error("mask_bits_keys16_stats: matrices up to 2048 x 2048"); return ACOSS_ENOTSUP; }
    const R16Work rw = r16_work_layout(w.radix, K, max_m, max_n);
    ACOSS_HIP(hipDeviceSynchronize());
    ACOSS_HIP(hipMemcpy(out, rw.counters, 20 * sizeof(int), hipMemcpyDeviceToHost));
    return ACOSS_OK;
}

// The pairs the last acoss_mask_bits_keys16(_f32)_batch call on this workspace left UNRESOLVED (waits for `stream`).  Matrices up
// to 1024 x 1024: always none (the wave-per-row kernels redo what the radix selection hands back).  Beyond (up to 2048 x 2048):
// the pairs whose rows or columns hold more cells inside the reach of their k-th smallest key than a work item holds (exact
// ties, thresholds below the key range) -- their rows of `bits` are undefined and the caller takes them through the float64 path.
// list: room for `cap` pair indices (K is always enough); *n: how many there are (may exceed cap: then only cap are listed).
extern "C" int acoss_mask_bits_keys16_unresolved(void *work, int K, int max_nx, int max_ny, int win, int32_t *list, int cap, int *n,
                                                 void *stream)
{
    if (!work || !n || K < 0 || cap < 0 || (cap > 0 && !list)) { set_error("mask_bits_keys16_unresolved: bad argument"); return ACOSS_EINVAL; }
    *n = 0;
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    if (K == 0 || (max_m <= 1024 && max_n <= 1024)) return ACOSS_OK;
    const ThreshWork w = thresh_work_layout(work, K, max_m, max_n, true);
    if (w.radix == nullptr) { set_error("mask_bits_keys16_unresolved: matrices up to 2048 x 2048"); return ACOSS_ENOTSUP; }
    const R16Work rw = r16_work_layout(w.radix, K, max_m, max_n);
    hipStream_t st = (hipStream_t)stream;
    int count = 0;
    ACOSS_HIP(hipMemcpyAsync(&count, rw.counters + 2, sizeof(int), hipMemcpyDeviceToHost, st));
    ACOSS_HIP(hipStreamSynchronize(st));
    count = count < 0 ? 0 : (count > K ? K : count);
    *n = count;
    const int take = count < cap ? count : cap;
    if (take > 0) {
        ACOSS_HIP(hipMemcpyAsync(list, rw.pair_list, sizeof(int) * (size_t)take, hipMemcpyDeviceToHost, st));
        ACOSS_HIP(hipStreamSynchronize(st));
    }
    return ACOSS_OK;
}

extern "C" int acoss_radix16_enabled(void) { return radix16_enabled() ? 1 : 0; }

// koff[pair] for acoss_crp_keys16_batch / acoss_mask_bits_keys16_f32_batch on a float32 corpus (k16_koff_pair_kernel: the key
// range hung on norms centred per pair).  xp / f32 / n32 / descs: exactly what acoss_crp_keys16_batch will be given.
extern "C" int acoss_keys16_koff_f32_batch(const float *xp, const float *f32, const float *n32, int d, const acoss_pair_desc *descs,
                                           int K, int win, int max_nx, int max_ny, uint32_t *koff, void *stream)
{
    if (!xp || !f32 || !n32 || !descs || !koff || K < 0 || max_nx < win || max_ny < win) {
        set_error("keys16_koff_f32_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if ((d != 12 && d != 13) || win != 9) { set_error("keys16_koff_f32_batch: supports d in {12, 13} and win == 9"); return ACOSS_ENOTSUP; }
    if (K == 0) return ACOSS_OK;
    hipStream_t st = (hipStream_t)stream;
    if (d == 12) hipLaunchKernelGGL(k16_koff_pair_kernel<12>, dim3((unsigned)K), dim3(256), 0, st, xp, max_nx, f32, n32, descs, win, koff);
    else hipLaunchKernelGGL(k16_koff_pair_kernel<13>, dim3((unsigned)K), dim3(256), 0, st, xp, max_nx, f32, n32, descs, win, koff);
    return launch_check("k16_koff_pair_kernel");
}

#ifdef ACOSS_PROBES
// development: device address of the side-buffer counter block of a workspace (int[64]: [0] hand-overs, [1..3] see K16Ctx)
extern "C" void *acoss_dev_side_counter(void *work, int K, int max_nx, int max_ny, int win)
{
    return thresh_work_layout(work, K, max_nx - win + 1, max_ny - win + 1, true).side_counter;
}
#endif
