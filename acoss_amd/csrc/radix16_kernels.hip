// radix16_kernels.hip -- the kNN selection (CRPUtils.py:169-219) on the 16-bit key plane as a two-pass radix selection with the
// keys held in registers (round 5).
//
// Why: the wave-per-row kernels of keys16_kernels.hip spend most of their instructions per ROW, not per key -- histogram scan
// and decode across the lanes, the rank among the winner's bin, the decision, the mask bits: 215 vector + 120 scalar
// instructions for 16 keys per lane, of which the binning is 48 -- and the column form pays a transposition through LDS on top.
// Here a block takes a TILE of 64 columns (32 rows) of one pair, every thread keeps its share of the tile's keys in registers
// (32 dwords) and the block finds all 64 (32) thresholds together:
//   pass 1   every key adds 1 to the 256-bin histogram of its column (row) over the key's HIGH byte -- LDS atomics, no value
//            returned; in the column kernel a lane owns a column pair, so the 32 lanes of an LDS lane group never meet in a bank;
//   decode 1 one lane per column walks its 256 counts: bin B of the k-th smallest, cb = keys below that bin;
//   pass 2   every key is compared with its column's window [256 B - 1, 256 B + 256] (one packed subtraction, two compares);
//            the few keys inside (7.7 on average at 1000 frames) go to the column's list with their row;
//   decode 2 sixteen lanes per column rank the list: the k-th smallest key th, how many keys are <= th, whether a key in the
//            reach of th's float32 error band lies above it.  If exactly k keys are <= th and none above is in reach, the
//            column's selection is `key <= th` (96 % of the benchmark's rows and columns); otherwise the cells in reach become a
//            work item and exact float64 values decide among them (r16_exact_kernel) -- the same set fix_row_band_range
//            (planar_select.h) selects, so the masks are the float64 path's bit for bit.
// The row kernel does the same per row (a wave owns four rows: coalesced 1 KB loads, bank conflicts in pass 1 accepted) and,
// with both thresholds known, writes the mutual mask's base bits `key < min(t1_row, t1_col)` straight from its registers, one
// byte per lane and 8 columns; r16_apply_kernel adds the few cells the work items select.  No per-row cross-lane scan, no
// transposition, no bit planes, no combine kernel.  Per key: ~14 vector instructions in the column kernel, ~19 in the row
// kernel (the wave-per-row kernels: 14 + 18 lane operations per cell) -- both kernels wait for HBM.
#include "radix16.h"

#include <stdlib.h>

namespace acoss {

typedef __attribute__((address_space(3))) unsigned r16_lds_word;
__device__ inline unsigned r16_lds_off(unsigned *p) { return (unsigned)(uintptr_t)(r16_lds_word *)p; }
__device__ inline void r16_lds_add(unsigned byte_off, unsigned v)
{
    __hip_atomic_fetch_add((r16_lds_word *)(uintptr_t)byte_off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

constexpr unsigned R16_XMAX = 257u;      // captured window: x = key - (256 B - 1) in [0, 257]
constexpr unsigned R16_XNONE = 0x1FFu;

template <int LINES>
struct R16Lds {
    unsigned hist[8192];                 // columns: [bin][32 column pairs], two 16-bit counts per word; rows: [row][bin]
    unsigned S[32 * LINES];              // sums over groups of eight bins: [group][line]
    unsigned lists[LINES * R16_LS];      // x | position << 9
    unsigned cnt[LINES];
    int lineB[LINES], linecb[LINES];
    unsigned thx[LINES], cle[LINES], red[LINES], rcnt[LINES];
    int item[LINES];
    unsigned t1[LINES];
};

// ---- decode 1: the high byte of every line's k-th smallest key -------------------------------------------------------------------
// S holds the line's counts summed over groups of eight bins; lane `line` walks the 32 groups, then the eight bins of its group.
template <int LINES, typename CountFn>
__device__ inline void r16_decode1(R16Lds<LINES> &sm, int line, int k, CountFn count_of)
{
    unsigned cum = 0, cbg = 0;
    int G = -1;
#pragma unroll 4
    for (int g = 0; g < 32; g++) {
        const unsigned s = sm.S[g * LINES + line];
        const bool take = (G < 0) & ((int)(cum + s) >= k);
        G = take ? g : G;
        cbg = take ? cum : cbg;
        cum += s;
    }
    G = max(G, 0);                                        // (k <= number of keys: a group is always found)
    int B = -1;
    unsigned cb = 0;
    cum = cbg;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const unsigned c = count_of(line, 8 * G + b);
        const bool take = (B < 0) & ((int)(cum + c) >= k);
        B = take ? 8 * G + b : B;
        cb = take ? cum : cb;
        cum += c;
    }
    sm.lineB[line] = max(B, 0);
    sm.linecb[line] = (int)cb;
    sm.cnt[line] = 0u;
    sm.thx[line] = 0x3FFu;
    sm.red[line] = 0u;
    sm.rcnt[line] = 0u;
}

// ---- decode 2: sixteen lanes per line rank the line's list ----------------------------------------------------------------------
// e = lane within the sixteen; `valid`: the line exists (its results are stored).  Writes t1 / item of the line (lane e == 0) and,
// for a line whose reach holds other cells, the work item.  Returns the line's t1 (all sixteen lanes).
template <int LINES>
__device__ inline unsigned r16_decode2(R16Lds<LINES> &sm, int line, int e, bool valid, int k, int p, int dir, int which,
                                       unsigned koff, bool adjacent_ok, const float *pair_band, const R16Work &w)
{
    const unsigned nraw = sm.cnt[line];
    const int n = (int)min(nraw, (unsigned)R16_CAP);
    const int B = sm.lineB[line], cb = sm.linecb[line];
    const int need = k - cb;
    const unsigned *lst = sm.lists + line * R16_LS;
    unsigned ent[4], xe[4];
    int lt[4], le[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int m = e + 16 * a;
        ent[a] = m < n ? lst[m] : R16_XNONE;
        xe[a] = ent[a] & 0x1FFu;
        lt[a] = le[a] = 0;
    }
    for (int m = 0; m < R16_CAP; m++) {
        const bool act = m < n;
        if (__ballot(act) == 0) break;
        const unsigned xm = act ? (lst[m] & 0x1FFu) : R16_XNONE;
        const bool inb = xm - 1u < 256u;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            lt[a] += (inb & (xm < xe[a])) ? 1 : 0;
            le[a] += (inb & (xm <= xe[a])) ? 1 : 0;
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const bool win = (xe[a] - 1u < 256u) & (lt[a] < need) & (need <= le[a]);
        if (win) { sm.thx[line] = xe[a]; sm.cle[line] = (unsigned)le[a]; }      // (equal keys: every winner writes the same pair)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned thx = sm.thx[line];
    const int cle = (int)sm.cle[line];
    const int w0 = 256 * B - 1;                            // the key of x == 0
    const unsigned th16 = (unsigned)(w0 + (int)thx);
    bool ok = (thx <= 256u) & (nraw <= (unsigned)R16_CAP) & (B > 0) & (th16 < K16_MAX);
    unsigned h_lo = 0, h_hi = 0;
    k16_reach(ok ? th16 : (K16_FINE + 1u), koff, adjacent_ok, pair_band, h_lo, h_hi);
    const int hlo_x = (int)h_lo - w0, hhi_x = (int)h_hi - w0;
    ok = ok & (hlo_x >= 0) & (hhi_x <= (int)R16_XMAX) & (hlo_x <= (int)thx);
    unsigned packed = 0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int x = (int)xe[a];
        if (x > (int)R16_XMAX) continue;
        packed += ((x > (int)thx) & (x <= hhi_x)) ? 1u : 0u;                     // above the threshold key, inside the reach
        packed += (x == 0) ? (1u << 8) : 0u;
        packed += ((x >= 1) & (x < hlo_x)) ? (1u << 16) : 0u;
        packed += ((x >= hlo_x) & (x <= hhi_x)) ? (1u << 24) : 0u;
    }
    if (packed) atomicAdd(&sm.red[line], packed);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned red = sm.red[line];
    const int above = (int)(red & 0xFFu), n0 = (int)((red >> 8) & 0xFFu), nlow = (int)((red >> 16) & 0xFFu), nR = (int)(red >> 24);
    const bool clean = ok & (cb + cle == k) & (above == 0);
    unsigned t1 = 0;
    int item = -1;
    if (clean) t1 = th16 + 1u;
    else if (ok) {
        const int below = hlo_x == 0 ? cb - n0 : cb + nlow;
        const int need2 = k - below;
        ok = (need2 >= 1) & (need2 <= nR) & (nR <= R16_CAP);
        if (ok) {
            t1 = h_lo;
            if (e == 0) {
                int idx = valid ? atomicAdd(&w.counters[0], 1) : -1;
                if (idx >= w.item_cap) idx = -2;
                sm.item[line] = idx;
                if (idx >= 0) {
                    R16Item *it = w.items + idx;
                    it->p = p; it->dir = dir; it->which = which; it->need = need2;
                    it->n = nR; it->sel_lo = 0u; it->sel_hi = 0u; it->reserved = 0;
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            item = sm.item[line];
            if (item >= 0) {
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const int x = (int)xe[a];
                    if ((x <= (int)R16_XMAX) & (x >= hlo_x) & (x <= hhi_x)) {
                        const unsigned slot = atomicAdd(&sm.rcnt[line], 1u);
                        w.items[item].pos[slot & (R16_CAP - 1)] = (uint16_t)(ent[a] >> 9);
                    }
                }
            } else if (item == -2) ok = false;
        }
    }
    if (!ok & valid & (e == 0)) {
        w.pair_flag[p] = 1;
        atomicAdd(&w.counters[1], 1);
    }
    if (!ok) { t1 = 0; item = -1; }
    return (t1 & 0xFFFFu) | ((unsigned)(item >= 0) << 16);
}

// ---- columns --------------------------------------------------------------------------------------------------------------------
// Block = 1024 threads = a tile of 64 columns x all rows (<= 1024) of one pair.  Thread (pi = t & 31, rs = t >> 5): column pair pi,
// rows rs + 32 q -- a wave instruction reads two rows x 128 contiguous bytes.
constexpr int R16C_THREADS = 1024;
constexpr int R16C_COLS = 64;

__global__ __launch_bounds__(R16C_THREADS, 8) void r16_cols_kernel(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                                   int win, double kv, int k_mode, R16Work w, int ldn, int col_blocks,
                                                                   const float *__restrict__ band, const uint32_t *__restrict__ koff_of, int dbg)
{
    __shared__ __attribute__((aligned(16))) R16Lds<R16C_COLS> sm;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / col_blocks;
    const int j0 = (lb % col_blocks) * R16C_COLS;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (j0 >= N) return;
    const int t = threadIdx.x;
    const int k = knn_count(k_mode, kv, M);
    uint16_t *t1_out = w.t1_col + (int64_t)p * ldn;
    int *item_out = w.item_col + (int64_t)p * ldn;
    if (k <= 0 || k >= M) {                                 // block-uniform: nothing / everything
        if (t < R16C_COLS && j0 + t < N) { t1_out[j0 + t] = k <= 0 ? (uint16_t)0 : (uint16_t)0xFFFFu; item_out[j0 + t] = -1; }
        return;
    }
    const int pi = t & 31, rs = t >> 5;
    // ---- the tile's keys: 32 dwords per thread, all loads in flight before the first is used
    const bool fast = ((ds.crp_pitch & 1) == 0) && ((ds.crp_off & 1) == 0);
    const int colp = min(j0 + 2 * pi, max(ds.crp_pitch - 2, 0));
    const uint16_t *base = keys + ds.crp_off + colp + (int64_t)rs * ds.crp_pitch;
    const int64_t rstep = 32 * (int64_t)ds.crp_pitch;
    unsigned wv[32];
    if (fast) {
#pragma unroll
        for (int q = 0; q < 32; q++)
            wv[q] = rs + 32 * q < M ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(base + q * rstep)) : 0xFFFFFFFFu;
    } else {
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const uint16_t *s = base + q * rstep;
            wv[q] = rs + 32 * q < M ? ((unsigned)s[0] | ((unsigned)s[ds.crp_pitch > 1 ? 1 : 0] << 16)) : 0xFFFFFFFFu;
        }
    }
    if (dbg == 1) {                                         // development: the loads alone
        unsigned a = 0;
#pragma unroll
        for (int q = 0; q < 32; q++) a ^= wv[q];
        if (a == 0x12345678u) t1_out[0] = 1;
        return;
    }
    {
        uint4 *hz = reinterpret_cast<uint4 *>(sm.hist);
        hz[t] = make_uint4(0u, 0u, 0u, 0u);
        hz[t + 1024] = make_uint4(0u, 0u, 0u, 0u);
    }
    lds_barrier();
    // ---- pass 1: histogram over the high bytes.  Word [bin][pi]: low half = the pair's even column, high half = its odd column
    const unsigned lanebase = r16_lds_off(sm.hist) + 4u * (unsigned)pi;
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const unsigned x = wv[q];
        r16_lds_add(lanebase + ((x >> 1) & 0x7F80u), 1u);
        r16_lds_add(lanebase + ((x >> 17) & 0x7F80u), 0x10000u);
    }
    lds_barrier();
    if (dbg == 2) { if (sm.hist[t] == 0x12345678u) t1_out[0] = 1; return; }
    // ---- decode 1
    {
        unsigned s = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) s += sm.hist[(8 * rs + b) * 32 + pi];
        sm.S[rs * R16C_COLS + 2 * pi] = s & 0xFFFFu;
        sm.S[rs * R16C_COLS + 2 * pi + 1] = s >> 16;
    }
    lds_barrier();
    if (t < R16C_COLS)
        r16_decode1<R16C_COLS>(sm, t, k, [&](int line, int bin) { return (sm.hist[bin * 32 + (line >> 1)] >> (16 * (line & 1))) & 0xFFFFu; });
    lds_barrier();
    if (dbg == 3) { if (sm.lineB[t & 63] == 0x12345678) t1_out[0] = 1; return; }
    // ---- pass 2: the keys inside each column's window, with their rows
    {
        const unsigned b0 = (unsigned)(256 * sm.lineB[2 * pi] - 1) & 0xFFFFu, b1 = (unsigned)(256 * sm.lineB[2 * pi + 1] - 1) & 0xFFFFu;
        const u16x2 bsh = k16_from_u32(b0 | (b1 << 16));
        unsigned *l0 = sm.lists + (2 * pi) * R16_LS, *l1 = l0 + R16_LS;
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const unsigned x = k16_to_u32(k16_from_u32(wv[q]) - bsh);
            const unsigned xl = x & 0xFFFFu, xh = x >> 16;
            const unsigned row = (unsigned)(rs + 32 * q);
            if (xl <= R16_XMAX) {
                const unsigned s = atomicAdd(&sm.cnt[2 * pi], 1u);
                if (s < (unsigned)R16_CAP) l0[s] = xl | (row << 9);
            }
            if (xh <= R16_XMAX) {
                const unsigned s = atomicAdd(&sm.cnt[2 * pi + 1], 1u);
                if (s < (unsigned)R16_CAP) l1[s] = xh | (row << 9);
            }
        }
    }
    lds_barrier();
    if (dbg == 4) { if (sm.cnt[t & 63] == 0x12345678u) t1_out[0] = 1; return; }
    // ---- decode 2
    {
        const int line = t >> 4, e = t & 15;
        const bool valid = j0 + line < N;
        const unsigned koff = koff_of[p];
        const float *pair_band = band + 2 * p;
        const bool adjacent_ok = k16_reach_adjacent_ok(koff, pair_band);
        const unsigned r = r16_decode2<R16C_COLS>(sm, line, e, valid, k, p, 1, j0 + line, koff, adjacent_ok, pair_band, w);
        if (valid && e == 0) {
            t1_out[j0 + line] = (uint16_t)(r & 0xFFFFu);
            item_out[j0 + line] = (r >> 16) ? sm.item[line] : -1;
        }
    }
}

// ---- rows + the mutual mask's base bits -------------------------------------------------------------------------------------------
// Block = 512 threads = a tile of 32 rows; wave v owns rows 4 v .. 4 v + 3, lane l the 8-column pieces l and l + 64 of each
// (a wave instruction reads 1 KB of one row).  After the selection the same registers give the mask: bit = key < min(t1_row, t1_col),
// one byte per lane and piece, 64 contiguous bytes per wave instruction.
constexpr int R16R_THREADS = 512;
constexpr int R16R_ROWS = 32;

__global__ __launch_bounds__(R16R_THREADS, 6) void r16_rows_kernel(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                                   int win, double kv, int k_mode, R16Work w, int ldm, int ldn, int row_blocks,
                                                                   const float *__restrict__ band, const uint32_t *__restrict__ koff_of,
                                                                   int mutual, uint64_t *__restrict__ bits, int dbg)
{
    __shared__ __attribute__((aligned(16))) R16Lds<R16R_ROWS> sm;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / row_blocks;
    const int i0 = (lb % row_blocks) * R16R_ROWS;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (i0 >= M) return;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int k = knn_count(k_mode, kv, N);
    const bool trivial = k <= 0 || k >= N;                  // block-uniform
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    // ---- keys: rows i0 + 4 wave + rr, pieces lane and lane + 64
    const bool fast = ((ds.crp_pitch & 7) == 0) && ((ds.crp_off & 7) == 0);
    unsigned wv[4][2][4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int i = i0 + 4 * wave + rr;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c0 = 8 * (lane + 64 * h);
            u32x4v v = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (i < M && c0 < N) {
                const uint16_t *src = keys + ds.crp_off + (int64_t)i * ds.crp_pitch + c0;
                if (fast && c0 + 8 <= ds.crp_pitch) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4v *>(src));
                else {
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const unsigned a = c0 + 2 * d < N ? (unsigned)src[2 * d] : 0xFFFFu, b = c0 + 2 * d + 1 < N ? (unsigned)src[2 * d + 1] : 0xFFFFu;
                        v[d] = a | (b << 16);
                    }
                }
                if (c0 + 8 > N) {                           // the piece that straddles the end of the row: padding behind it
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        if (c0 + 2 * d >= N) v[d] |= 0xFFFFu;
                        if (c0 + 2 * d + 1 >= N) v[d] |= 0xFFFF0000u;
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < 4; d++) wv[rr][h][d] = v[d];
        }
    }
    // the column bounds of this lane's two pieces (the same for all its rows)
    u32x4v tc[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int c0 = 8 * (lane + 64 * h);
        tc[h] = (u32x4v){0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (mutual && c0 < N) tc[h] = *reinterpret_cast<const u32x4v *>(w.t1_col + (int64_t)p * ldn + c0);
    }
    uint16_t *t1_out = w.t1_row + (int64_t)p * ldm;
    int *item_out = w.item_row + (int64_t)p * ldm;
    if (dbg == 1) {
        unsigned a = tc[0][0] ^ tc[1][1];
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int d = 0; d < 4; d++) a ^= wv[rr][h][d];
        if (a == 0x12345678u) t1_out[0] = 1;
        return;
    }
    if (!trivial) {
        {
            uint4 *hz = reinterpret_cast<uint4 *>(sm.hist);
#pragma unroll
            for (int z = 0; z < 4; z++) hz[t + 512 * z] = make_uint4(0u, 0u, 0u, 0u);
        }
        lds_barrier();
        // ---- pass 1: hist[row][bin]
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const unsigned rowbase = r16_lds_off(sm.hist) + 1024u * (unsigned)(4 * wave + rr);
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned x = wv[rr][h][d];
                    r16_lds_add(rowbase + ((x >> 6) & 0x3FCu), 1u);
                    r16_lds_add(rowbase + ((x >> 22) & 0x3FCu), 1u);
                }
            }
        }
        lds_barrier();
        if (dbg == 2) { if (sm.hist[t] == 0x12345678u) t1_out[0] = 1; return; }
        // ---- decode 1: thread (g = t & 31, r = t >> 5): rows r and r + 16, bins 8 g .. 8 g + 7
        {
            const int g = t & 31, r = t >> 5;
#pragma unroll
            for (int z = 0; z < 2; z++) {
                const uint4 a = reinterpret_cast<const uint4 *>(sm.hist + (r + 16 * z) * 256 + 8 * g)[0];
                const uint4 b = reinterpret_cast<const uint4 *>(sm.hist + (r + 16 * z) * 256 + 8 * g)[1];
                sm.S[g * R16R_ROWS + r + 16 * z] = (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w);
            }
        }
        lds_barrier();
        if (t < R16R_ROWS) r16_decode1<R16R_ROWS>(sm, t, k, [&](int line, int bin) { return sm.hist[line * 256 + bin]; });
        lds_barrier();
        if (dbg == 3) { if (sm.lineB[t & 31] == 0x12345678) t1_out[0] = 1; return; }
        // ---- pass 2
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int line = 4 * wave + rr;
            const unsigned b0 = (unsigned)(256 * sm.lineB[line] - 1) & 0xFFFFu;
            const u16x2 bsh = k16_from_u32(b0 | (b0 << 16));
            unsigned *l0 = sm.lists + line * R16_LS;
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned x = k16_to_u32(k16_from_u32(wv[rr][h][d]) - bsh);
                    const unsigned xl = x & 0xFFFFu, xh = x >> 16;
                    const unsigned col = (unsigned)(8 * (lane + 64 * h) + 2 * d);
                    if (xl <= R16_XMAX) {
                        const unsigned s = atomicAdd(&sm.cnt[line], 1u);
                        if (s < (unsigned)R16_CAP) l0[s] = xl | (col << 9);
                    }
                    if (xh <= R16_XMAX) {
                        const unsigned s = atomicAdd(&sm.cnt[line], 1u);
                        if (s < (unsigned)R16_CAP) l0[s] = xh | ((col + 1u) << 9);
                    }
                }
            }
        }
        lds_barrier();
        if (dbg == 4) { if (sm.cnt[t & 31] == 0x12345678u) t1_out[0] = 1; return; }
        // ---- decode 2
        {
            const int line = t >> 4, e = t & 15;
            const bool valid = i0 + line < M;
            const unsigned koff = koff_of[p];
            const float *pair_band = band + 2 * p;
            const bool adjacent_ok = k16_reach_adjacent_ok(koff, pair_band);
            const unsigned r = r16_decode2<R16R_ROWS>(sm, line, e, valid, k, p, 0, i0 + line, koff, adjacent_ok, pair_band, w);
            if (e == 0) {
                sm.t1[line] = r & 0xFFFFu;
                if (valid) {
                    t1_out[i0 + line] = (uint16_t)(r & 0xFFFFu);
                    item_out[i0 + line] = (r >> 16) ? sm.item[line] : -1;
                }
            }
        }
        lds_barrier();
    } else {
        if (t < R16R_ROWS) {
            sm.t1[t] = k <= 0 ? 0u : 0xFFFFu;
            if (i0 + t < M) { t1_out[i0 + t] = k <= 0 ? (uint16_t)0 : (uint16_t)0xFFFFu; item_out[i0 + t] = -1; }
        }
        lds_barrier();
    }
    if (dbg == 5) return;
    // ---- the mask's base bits
    const u16x2 one = k16_opaque_ones();
    unsigned char *ob = reinterpret_cast<unsigned char *>(bits) + ((int64_t)p * w.max_m + i0) * 128;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int line = 4 * wave + rr;
        if (i0 + line >= M) break;                          // wave-uniform
        const u16x2 tr = k16_splat(sm.t1[line]);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            unsigned acc = 0;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const u16x2 m1 = __builtin_elementwise_min(k16_from_u32(tc[h][d]), tr);
                const u16x2 f = __builtin_elementwise_min(__builtin_elementwise_sub_sat(m1, k16_from_u32(wv[rr][h][d])), one);
                acc |= k16_to_u32(f) << (2 * d);
            }
            ob[line * 128 + lane + 64 * h] = (unsigned char)((acc | (acc >> 15)) & 0xFFu);
        }
    }
}

// ---- exact values for the work items -------------------------------------------------------------------------------------------------
// One wave per item: the item's cells (<= 64) get their exact windowed sums -- the arithmetic of fix_row_band_range
// (planar_select.h): FMA chain over the bins of the rolled x frame, exact_term(), the nine terms added in window order in
// float64 -- and the `need` smallest by (value, position) are selected.
template <typename FT>
__global__ __launch_bounds__(64) void r16_exact_kernel(const FT *__restrict__ feats, const FT *__restrict__ norms, int d,
                                                       const acoss_pair_desc *__restrict__ descs, int win, R16Work w)
{
    __shared__ double cval[64];
    __shared__ unsigned long long key[R16_CAP];
    __shared__ int posv[R16_CAP];
    const int lane = threadIdx.x;
    const int count = min(w.counters[0], w.item_cap);
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
        R16Item *item = w.items + it;
        const int p = item->p;
        if (w.pair_flag[p]) continue;
        const int n = min(item->n, R16_CAP), need = item->need, dir = item->dir, which = item->which;
        const acoss_pair_desc ds = descs[p];
        if (lane < n) posv[lane] = (int)item->pos[lane];
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += 7) {
            const int g = lane / 9, kk = lane - 9 * g, el = c0 + g;
            if (lane < 63 && el < n) {
                const int pos = posv[el];
                const int i = dir == 0 ? which : pos, j = dir == 0 ? pos : which;
                const FT *x = feats + (ds.x_row0 + i + kk) * d, *y = feats + (ds.y_row0 + j + kk) * d;
                FT acc = 0;
                for (int b = 0; b < d; b++) {
                    int src = b - ds.shift;
                    if (src < 0) src += d;
                    acc = fma(x[src], y[b], acc);
                }
                cval[g * 9 + kk] = exact_term(acc, norms[ds.x_row0 + i + kk] + norms[ds.y_row0 + j + kk]);
            }
            __syncthreads();
            if (lane < 7 && c0 + lane < n) {
                double s_ = 0.0;
                for (int q = 0; q < win; q++) s_ += cval[lane * 9 + q];
                key[c0 + lane] = f64_key(s_);
            }
            __syncthreads();
        }
        int rank = 1;
        unsigned long long mykey = 0;
        int mypos = 0;
        if (lane < n) { mykey = key[lane]; mypos = posv[lane]; }
        for (int m = 0; m < n; m++) {
            const unsigned long long km = key[m];
            const int pm = posv[m];
            rank += ((km < mykey) | ((km == mykey) & (pm < mypos))) ? 1 : 0;
        }
        const unsigned long long sel = __ballot(lane < n && rank <= need);
        if (lane == 0) { item->sel_lo = (unsigned)sel; item->sel_hi = (unsigned)(sel >> 32); }
        __syncthreads();
    }
}

// ---- the cells the items select, where the other side selects them too ---------------------------------------------------------------
__global__ __launch_bounds__(64) void r16_apply_kernel(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                       R16Work w, int ldm, int ldn, int mutual, uint64_t *__restrict__ bits)
{
    const int lane = threadIdx.x;
    const int count = min(w.counters[0], w.item_cap);
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
        const R16Item *item = w.items + it;
        const int p = item->p;
        if (w.pair_flag[p]) continue;
        const uint64_t sel = ((uint64_t)item->sel_hi << 32) | item->sel_lo;
        if (!((sel >> lane) & 1)) continue;
        const int pos = (int)item->pos[lane];
        const int i = item->dir == 0 ? item->which : pos, j = item->dir == 0 ? pos : item->which;
        bool other = true;
        if (mutual) {
            const acoss_pair_desc ds = descs[p];
            const unsigned key = keys[ds.crp_off + (int64_t)i * ds.crp_pitch + j];
            const int oi = item->dir == 0 ? w.item_col[(int64_t)p * ldn + j] : w.item_row[(int64_t)p * ldm + i];
            const unsigned t1 = item->dir == 0 ? w.t1_col[(int64_t)p * ldn + j] : w.t1_row[(int64_t)p * ldm + i];
            other = key < t1;
            if (!other && oi >= 0) {
                const R16Item *o = w.items + oi;
                const uint64_t osel = ((uint64_t)o->sel_hi << 32) | o->sel_lo;
                const int want = item->dir == 0 ? i : j;
                for (int m = 0; m < min(o->n, R16_CAP); m++) other |= ((osel >> m) & 1) && (int)o->pos[m] == want;
            }
        }
        if (other) atomicOr(reinterpret_cast<unsigned long long *>(bits) + ((int64_t)p * w.max_m + i) * 16 + (j >> 6), 1ull << (j & 63));
    }
}

// the flagged pairs, compacted (one block)
__global__ __launch_bounds__(256) void r16_flag_list_kernel(R16Work w, int K)
{
    __shared__ int n;
    if (threadIdx.x == 0) n = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < K; p += 256)
        if (w.pair_flag[p]) w.pair_list[atomicAdd(&n, 1)] = p;
    __syncthreads();
    if (threadIdx.x == 0) w.counters[2] = n;
}

}  // namespace acoss

using namespace acoss;

// Development / stage entry points (round 5): the two selection kernels alone.  `work`: acoss_radix16_work_bytes() bytes.
extern "C" size_t acoss_radix16_work_bytes(int K, int max_nx, int max_ny, int win)
{
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    return r16_work_bytes(K, (max_m + 7) & ~7, (max_n + 7) & ~7);
}

extern "C" int acoss_radix16_layout(void *work, int K, int max_nx, int max_ny, int win, void **ptrs, int *dims)
{
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    const int ldm = (max_m + 7) & ~7, ldn = (max_n + 7) & ~7;
    const R16Work w = r16_work_layout(work, K, ldm, ldn);
    ptrs[0] = w.t1_row; ptrs[1] = w.t1_col; ptrs[2] = w.item_row; ptrs[3] = w.item_col; ptrs[4] = w.counters; ptrs[5] = w.items;
    ptrs[6] = w.pair_flag; ptrs[7] = w.pair_list;
    dims[0] = ldm; dims[1] = ldn; dims[2] = w.item_cap; dims[3] = (int)sizeof(R16Item);
    return ACOSS_OK;
}

// what = 1: columns; 2: rows + base bits (needs the columns' t1 when mutual); 4: exact + apply.  Bits of `what` may be combined.
template <typename FT>
static int radix16_run(int what, const uint16_t *keys16, const float *band, const uint32_t *koff, const FT *feats, const FT *norms, int d,
                       const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits,
                       void *work, hipStream_t st)
{
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    const int ldm = (max_m + 7) & ~7, ldn = (max_n + 7) & ~7;
    R16Work w = r16_work_layout(work, K, ldm, ldn);
    w.max_m = max_m;
    w.max_n = max_n;
    double kv;
    int mode;
    if (kappa == 0.0) { kv = 0.0; mode = 2; } else if (kappa < 1.0) { kv = kappa; mode = 0; } else { kv = kappa; mode = 1; }
    const int cb = ceil_div(max_n, R16C_COLS), rb = ceil_div(max_m, R16R_ROWS);
    if ((int64_t)K * cb > 0x7fffffffLL || (int64_t)K * rb > 0x7fffffffLL) { set_error("radix16: batch too large"); return ACOSS_ENOTSUP; }
    if (what & 1) {
        ACOSS_HIP(hipMemsetAsync(w.counters, 0, 256, st));
        ACOSS_HIP(hipMemsetAsync(w.pair_flag, 0, (size_t)K, st));
        if (mutual) {
            hipLaunchKernelGGL(r16_cols_kernel, dim3((unsigned)((int64_t)K * cb)), dim3(R16C_THREADS), 0, st, keys16, descs, win, kv, mode, w, ldn, cb, band, koff, what >> 8);
            const int rc = launch_check("r16_cols_kernel");
            if (rc) return rc;
        }
    }
    if (what & 2) {
        hipLaunchKernelGGL(r16_rows_kernel, dim3((unsigned)((int64_t)K * rb)), dim3(R16R_THREADS), 0, st, keys16, descs, win, kv, mode, w, ldm, ldn, rb, band, koff, mutual, bits, what >> 8);
        const int rc = launch_check("r16_rows_kernel");
        if (rc) return rc;
    }
    if (what & 4) {
        hipLaunchKernelGGL(r16_exact_kernel<FT>, dim3(8192), dim3(64), 0, st, feats, norms, d, descs, win, w);
        hipLaunchKernelGGL(r16_apply_kernel, dim3(8192), dim3(64), 0, st, keys16, descs, w, ldm, ldn, mutual, bits);
        hipLaunchKernelGGL(r16_flag_list_kernel, dim3(1), dim3(256), 0, st, w, K);
        const int rc = launch_check("r16_exact_kernel / r16_apply_kernel");
        if (rc) return rc;
    }
    return ACOSS_OK;
}

extern "C" int acoss_radix16_stage(int what, const uint16_t *keys16, const float *band, const uint32_t *koff, const double *feats,
                                   const double *norms, int d, const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                                   double kappa, int mutual, uint64_t *bits, void *work, size_t work_bytes, void *stream)
{
    if (!keys16 || !band || !koff || !descs || !work || K < 0 || max_nx < win || max_ny < win || kappa < 0.0) {
        set_error("radix16_stage: bad argument");
        return ACOSS_EINVAL;
    }
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    if (max_m > 1024 || max_n > 1024) { set_error("radix16_stage: matrices up to 1024 x 1024"); return ACOSS_ENOTSUP; }
    if (work_bytes < acoss_radix16_work_bytes(K, max_nx, max_ny, win)) { set_error("radix16_stage: workspace too small"); return ACOSS_EINVAL; }
    if (K == 0) return ACOSS_OK;
    return radix16_run<double>(what, keys16, band, koff, feats, norms, d, descs, K, win, max_nx, max_ny, kappa, mutual, bits, work, (hipStream_t)stream);
}
