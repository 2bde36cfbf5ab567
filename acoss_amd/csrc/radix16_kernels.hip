// radix16_kernels.hip -- the kNN selection (CRPUtils.py:169-219) on the 16-bit key plane as a two-pass radix selection with the
// keys held in registers (round 5).
//
// Why: the wave-per-row kernels of keys16_kernels.hip spend most of their instructions per ROW, not per key -- histogram scan
// and decode across the lanes, the rank among the winner's bin, the decision, the mask bits: 215 vector + 120 scalar
// instructions for 16 keys per lane, of which the binning is 48 -- and the column form pays a transposition through LDS on top.
// Here a block takes a TILE of 32 lines (rows or columns) x all positions of one pair, sixteen threads per line; every thread keeps
// its share of the tile's keys in registers (32 dwords; 48 / 64 for sides up to 1536 / 2048: NQ) and the block finds all 32
// thresholds together:
//   pass 1   every key adds 1 to the 256-bin histogram of its line over the key's HIGH byte -- LDS atomics, no value returned; in
//            the column kernel a lane owns a column pair, so the 32 lanes of an LDS lane group never meet in a bank;
//   decode 1 one lane per line walks its 256 counts: bin B of the k-th smallest, cb = keys below that bin;
//   pass 2   every key is compared with its line's window [256 B - 1, 256 B + 256] (one packed subtraction, two compares, one
//            wave-uniform branch); the few keys inside (7.7 on average at 1000 frames) are kept in the thread's own LDS slots and
//            counted (by sixteens, by key) after the sweep;
//   decode 2 sixteen lanes per line find the k-th smallest key th, how many keys are <= th, whether a key in the reach of th's
//            float32 error band lies beside it.  If exactly k keys are <= th and none above is in reach, the line's selection is
//            `key <= th` (96 % of the benchmark's rows and columns); otherwise the cells in reach become a work item (every thread
//            contributes the ones among its own hits) and exact float64 values decide among them (r16_exact_tiles_kernel) -- the
//            same set fix_row_band_range (planar_select.h) selects, so the masks are the float64 path's bit for bit.
// The row kernel does the same per row (a wave owns four rows: coalesced 1 KB loads, bank conflicts in pass 1 accepted) and,
// with both thresholds known, writes the mutual mask's base bits `key < min(t1_row, t1_col)` straight from its registers, one
// byte per lane and 8 columns; r16_apply_kernel adds the few cells the work items select.  No per-row cross-lane scan, no
// transposition, no bit planes, no combine kernel.  Per key ~14 vector + 8 scalar instructions in either kernel; PMC: the vector
// ALUs are active 0.91 / 0.81 of the cycles (columns / rows) -- bound by instruction issue.
#include "radix16.h"

#include <stdlib.h>

namespace acoss {

typedef __attribute__((address_space(3))) unsigned r16_lds_word;
__device__ inline unsigned r16_lds_off(unsigned *p) { return (unsigned)(uintptr_t)(r16_lds_word *)p; }
__device__ inline void r16_lds_add(unsigned byte_off, unsigned v)
{
    __hip_atomic_fetch_add((r16_lds_word *)(uintptr_t)byte_off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// column loads: a 32-column tile reads half a cache line per row and the neighbouring tile (the next block) the other half --
// plain loads leave the line in L2 for it, non-temporal ones do not
#ifndef R16_COL_NT
#define R16_COL_NT 0
#endif
#ifndef R16_HIT_BRANCH
#define R16_HIT_BRANCH 1
#endif
#if R16_COL_NT
#define R16_COL_LOAD(p) __builtin_nontemporal_load(p)
#else
#define R16_COL_LOAD(p) (*(p))
#endif
constexpr unsigned R16_XMAX = 257u;      // captured window: x = key - (256 B - 1) in [0, 257]
constexpr int R16_THREADS = 16 * R16_TILE_LINES;   // sixteen threads per line
constexpr int R16_WORDS = R16_TILE_LINES / 2;      // histogram words per bin (two lines each)
constexpr int R16_BINSHIFT = R16_WORDS == 32 ? 1 : 2;        // key >> (8 + BINSHIFT) << ... : byte offset of a bin = bin * 4 * WORDS
constexpr unsigned R16_BINMASK = 0xFFu * 4u * R16_WORDS;
static_assert(R16_TILE_LINES == 64 || R16_TILE_LINES == 32, "block geometry");
constexpr int R16_LINES = R16_TILE_LINES; // columns (rows) per block

// Line l of a tile has its pass-1 counts in half (DIR ? l & 1 : l >> 5) of word (DIR ? l >> 1 : l & 31) of every histogram bin:
// a column kernel lane owns two neighbouring columns (one dword of a row), the rows of a row-kernel LDS lane group differ in
// their low five bits -- either way the lanes of a group add into different banks.
struct R16Lds {
    unsigned hist[256 * R16_WORDS];            // pass 1: [bin][words]; pass 2: every thread's hits [slot][thread]; row kernel, last: the mask bytes
    unsigned S[32 * R16_LINES];                // decode 1: the counts summed over groups of eight bins, [group][line]; then hcnt[thread]
    unsigned fine[(R16_XMAX + 1) * (R16_LINES / 4)];      // [x][line >> 2]: byte (line & 3) = the line's keys with that x
    unsigned sub[16 * R16_LINES];              // [(x - 1) >> 4][line]: the line's keys inside bin B, by sixteens
    int lineB[R16_LINES], linecb[R16_LINES];
    unsigned t1[R16_LINES];
    unsigned dec[R16_LINES], rcnt[R16_LINES];
    int ditem[R16_LINES], nr[R16_LINES];
    unsigned sweep3, item_n;
    uint4 tcol[256];                           // row kernel: the pair's column bounds (2048 columns in the long form)
};
constexpr int R16_LINE_SHIFT = 20;             // a hit: x (9 bits) | position << 9 (11 bits) | line << 20
constexpr unsigned R16_POS_MASK = 0x7FFu;
constexpr int R16_SLOTS = 8;                   // hits a thread can hold.  A thread with more
                                               // (temporally smooth features: runs of neighbouring cells inside one window) still
                                               // COUNTS them all; the block then finds the items' cells by a third sweep

template <int DIR> __device__ inline int r16_word(int line) { return DIR ? line >> 1 : line & (R16_WORDS - 1); }
template <int DIR> __device__ inline int r16_half(int line) { return DIR ? line & 1 : line / R16_WORDS; }

// ---- decode 1: the high byte of every line's k-th smallest key (lane = line) ---------------------------------------------------
template <int DIR>
__device__ inline void r16_decode1(R16Lds &sm, int line, int k)
{
    const unsigned *S = sm.S;
    unsigned cum = 0, cbg = 0;
    int G = -1;
#pragma unroll 8
    for (int g = 0; g < 32; g++) {
        const unsigned s = S[g * R16_LINES + line];
        const bool take = (G < 0) & ((int)(cum + s) >= k);
        G = take ? g : G;
        cbg = take ? cum : cbg;
        cum += s;
    }
    G = max(G, 0);                                        // (k <= number of keys: a group is always found)
    int B = -1;
    unsigned cb = 0;
    cum = cbg;
    const int wd = r16_word<DIR>(line), sh = 16 * r16_half<DIR>(line);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const unsigned c = (sm.hist[(8 * G + b) * R16_WORDS + wd] >> sh) & 0xFFFFu;
        const bool take = (B < 0) & ((int)(cum + c) >= k);
        B = take ? 8 * G + b : B;
        cb = take ? cum : cb;
        cum += c;
    }
    sm.lineB[line] = max(B, 0);
    sm.linecb[line] = (int)cb;
}

// ---- decode 2: sixteen lanes per line (one DPP row), every thread of the block at once --------------------------------------------
// Lane e of a line takes the line's e-th count by sixteens; a row scan finds the sixteen that holds the k-th smallest; lane e
// then takes key 16 s + 1 + e of it and a second scan finds the key itself.  The lane that holds it decides (clean / work item)
// and, for an item, the sixteen look through the hit slots of the threads that own the line's keys for the cells in reach.
__device__ inline int r16_row_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR2, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR4, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR8, 0xf, 0xf, false);
    return v;
}

template <int DIR>
__device__ inline void r16_decode2(R16Lds &sm, int line, int e, bool valid, int k, int p, int which, int item0, int tile_slot,
                                   unsigned koff, bool adjacent_ok, const float *pair_band, const R16Work &w, uint16_t *t1_out,
                                   int *item_out, int dbg)
{
    const int B = sm.lineB[line], cb = sm.linecb[line];
    const int need = k - cb;                               // >= 1: the rank of the k-th smallest inside bin B
    const int c = (int)sm.sub[e * R16_LINES + line];
    const int incl = r16_row_scan(c);
    if ((incl - c < need) & (need <= incl)) sm.dec[line] = (unsigned)e | ((unsigned)(need - (incl - c)) << 8) | ((unsigned)(incl - c) << 20);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned d1 = sm.dec[line];
    // (B == 0, a threshold in the lowest 256 keys: the window starts at key -1, which no cell has -- the 0xFFFF pads behind a
    //  line's end land on x == 0 there; the reach of a key never goes below key 0, so x == 0 is never read for such a line)
    const bool ok = d1 != 0x7FFFFFFFu;
    const int sstar = (int)(d1 & 15u), r = (int)((d1 >> 8) & 0xFFFu), cums = (int)(d1 >> 20) & 0xFFF;
    const int fq = line >> 2, fs = 8 * (line & 3);
    constexpr int FW = R16_LINES / 4;
    const unsigned *fbase = sm.fine + (16 * sstar) * FW + fq;        // x = 16 sstar + i
    const int f = (int)((fbase[(1 + e) * FW] >> fs) & 0xFFu);       // this lane: x = 16 sstar + 1 + e
    const int incl2 = r16_row_scan(f);
    // (the counts by key are bytes, four lines to a word: 256 equal keys in one line would carry into its neighbour -- then the
    //  sixteen counts no longer add up to the count of their sixteen, in either line)
    const int total16 = __shfl(incl2, (threadIdx.x & 48) + 15, 64);
    const bool mine = ok & (incl2 - f < r) & (r <= incl2);           // one lane of the sixteen
    unsigned t1 = 0;
    int item = -1, hlo_x = 0, hhi_x = -1;
    if (mine) {
        const int thx = 16 * sstar + 1 + e;                // 1 .. 256
        const int cle = cb + cums + incl2;                 // keys <= the threshold key
        const int w0 = 256 * B - 1;                        // the key of x == 0
        const unsigned th16 = (unsigned)(w0 + thx);
        int why = th16 < K16_MAX ? 0 : 1;
        if (total16 != (int)sm.sub[sstar * R16_LINES + line]) why = 2;
        bool good = why == 0;
        unsigned h_lo = 0, h_hi = 0;
        k16_reach(good ? th16 : (K16_FINE + 1u), koff, adjacent_ok, pair_band, h_lo, h_hi);
        hlo_x = (int)h_lo - w0;
        hhi_x = (int)h_hi - w0;
        if (good & !((hlo_x >= thx - 1) & (hlo_x <= thx) & (hhi_x >= thx) & (hhi_x <= thx + 1))) { good = false; why = 3; }
        const int c_m1 = (int)((fbase[e * FW] >> fs) & 0xFFu), c_p1 = (int)((fbase[(2 + e) * FW] >> fs) & 0xFFu);
        const int above = hhi_x > thx ? c_p1 : 0, in_lo = hlo_x < thx ? c_m1 : 0;
        int nR_keep = 0;
        if (good & (cle == k) & (above == 0)) t1 = th16 + 1u;
        else if (good) {
            // the cells in reach decide by exact value: those below the reach are selected, `need2` of the reach's cells too
            const int nR = in_lo + f + above, below = cle - f - in_lo, need2 = k - below;
            good = (need2 >= 1) & (need2 <= nR) & (nR <= R16_CAP);
            if (!good) why = nR > R16_CAP ? 4 : 5;
            nR_keep = nR;
            if (good) {
                t1 = h_lo;
                if (valid && dbg != 8) {
                    const int li = (int)atomicAdd(&sm.item_n, 1u);
                    if (li < R16_TILE_ITEMS) item = item0 + li;
                    else {
                        item = w.static_items + atomicAdd(&w.counters[0], 1);
                        if (item >= w.item_cap) { item = -1; good = false; why = 6; }
                    }
                    if (item >= 0) {
                        R16Item *it = w.items + item;
                        it->p = p; it->dir = DIR; it->which = which; it->need = need2;
                        it->n = nR; it->sel_lo = 0u; it->sel_hi = 0u; it->reserved = 0;
                    }
                }
            }
        }
        if (!good) { t1 = 0; item = -1; }
        sm.t1[line] = t1;
        sm.dec[line] = 0x80000000u | (unsigned)(item >= 0) | ((unsigned)(hlo_x & 0x1FF) << 1) | ((unsigned)(hhi_x & 0x1FF) << 10) | ((unsigned)good << 19);
        sm.ditem[line] = item;
        sm.nr[line] = item >= 0 ? nR_keep : 0;
        if (valid) {
            if (!good) {
                w.pair_flag[p] = 1;
                atomicAdd(&w.counters[1], 1);
                atomicAdd(&w.counters[8 + why], 1);           // statistics: [9] key at the top, [10] byte counts, [11] reach, [12] ties, [13] need, [14] items
            }
            t1_out[which] = (uint16_t)t1;
            item_out[which] = item;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned d2 = sm.dec[line];
    if (!(d2 & 0x80000000u)) {                              // no lane found the key (counts that do not add up: cannot happen)
        if (e == 0) {
            sm.t1[line] = 0u;
            if (valid) { w.pair_flag[p] = 1; atomicAdd(&w.counters[1], 1); atomicAdd(&w.counters[B > 0 ? 15 : 16], 1); t1_out[which] = 0; item_out[which] = -1; }
        }
        return;
    }
}

// ---- the selection kernel ------------------------------------------------------------------------------------------------------------
// DIR 1, columns: a tile of 64 columns x all rows of one pair.  Thread (pi = t & 31, rs = t >> 5): column pair pi (the two halves
//        of its dwords), rows rs + 32 q -- a wave instruction reads two rows x 128 contiguous bytes.
// DIR 0, rows: a tile of 64 rows x all columns.  Wave v, lane l: row 16 (v >> 2) + (l & 15), 8-column pieces j + 16 q with
//        j = (l >> 4) + 4 (v & 3) -- a wave instruction reads 64 contiguous bytes of each of sixteen rows, and the 32 lanes of an LDS
//        lane group belong to sixteen different rows.  With both bounds known the row kernel writes the mutual mask's base bits
//        key < min(t1_row, t1_col) from its registers.
template <int DIR, int NQ>
__global__ __launch_bounds__(R16_THREADS, NQ == 32 ? 8 : (NQ == 48 ? 6 : 4)) void r16_select_kernel(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                                    int win, double kv, int k_mode, R16Work w, int ldm, int ldn, int tiles,
                                                                    const float *__restrict__ band, const uint32_t *__restrict__ koff_of,
                                                                    int mutual, uint64_t *__restrict__ bits, int wpr, int item0, int dbg)
{
    __shared__ __attribute__((aligned(16))) R16Lds sm;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / tiles;
    const int l0 = (lb % tiles) * R16_LINES;               // first line of the tile
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int n_lines = DIR ? N : M, n_pos = DIR ? M : N;
    if (l0 >= n_lines) {
        if (threadIdx.x == 0) w.tile_used[item0 / R16_TILE_ITEMS + lb] = 0;
        return;
    }
    const int t = threadIdx.x, lane = t & 63;
    const int k = knn_count(k_mode, kv, n_pos);
    const bool trivial = k <= 0 || k >= n_pos;              // block-uniform: nothing / everything
    uint16_t *t1_out = DIR ? w.t1_col + (int64_t)p * ldn : w.t1_row + (int64_t)p * ldm;
    int *item_out = DIR ? w.item_col + (int64_t)p * ldn : w.item_row + (int64_t)p * ldm;
    if (DIR == 1 && trivial) {
        if (t < R16_LINES && l0 + t < n_lines) { t1_out[l0 + t] = k <= 0 ? (uint16_t)0 : (uint16_t)0xFFFFu; item_out[l0 + t] = -1; }
        if (t == 0) w.tile_used[item0 / R16_TILE_ITEMS + lb] = 0;
        return;
    }
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    // ---- the tile's keys: 32 dwords per thread, every load in flight before the first is used
    unsigned wv[NQ];
    const int pi = t & (R16_WORDS - 1), rs = t / R16_WORDS;   // columns (rs: 0 .. 31)
    const int wave = t >> 6;
    const int rr = 16 * (wave >> 2) + (lane & 15), jj = (lane >> 4) + 4 * (wave & 3);      // rows
    if (DIR == 1) {
        const bool fast = ((ds.crp_pitch & 1) == 0) && ((ds.crp_off & 1) == 0);
        const int c0 = l0 + 2 * pi;
        const int64_t rstep = 32 * (int64_t)ds.crp_pitch;
        if (fast) {
            // (column pairs past the pitch repeat the last pair: their lines do not exist)
            const uint16_t *base = keys + ds.crp_off + min(c0, max(ds.crp_pitch - 2, 0)) + (int64_t)rs * ds.crp_pitch;
#pragma unroll
            for (int q = 0; q < NQ; q++)
                wv[q] = rs + 32 * q < M ? R16_COL_LOAD(reinterpret_cast<const unsigned *>(base + q * rstep)) : 0xFFFFFFFFu;
        } else {
            const bool ha = c0 < ds.crp_pitch, hb = c0 + 1 < ds.crp_pitch;
            const uint16_t *base = keys + ds.crp_off + (ha ? c0 : 0) + (int64_t)rs * ds.crp_pitch;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const uint16_t *s = base + q * rstep;
                wv[q] = rs + 32 * q < M ? ((ha ? (unsigned)s[0] : 0xFFFFu) | ((hb ? (unsigned)s[1] : 0xFFFFu) << 16)) : 0xFFFFFFFFu;
            }
        }
    } else {
        const bool fast = ((ds.crp_pitch & 7) == 0) && ((ds.crp_off & 7) == 0);
        const int i = l0 + rr;
#pragma unroll
        for (int q = 0; q < NQ / 4; q++) {
            const int c0 = 8 * (jj + 16 * q);
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
            for (int d = 0; d < 4; d++) wv[4 * q + d] = v[d];
        }
        if (t < 4 * NQ) {
            u32x4v tc = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (mutual && 8 * t < N) tc = *reinterpret_cast<const u32x4v *>(w.t1_col + (int64_t)p * ldn + 8 * t);
            sm.tcol[t] = make_uint4(tc[0], tc[1], tc[2], tc[3]);
        }
    }
    if (dbg == 1) {                                         // development: the loads alone
        unsigned a = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) a ^= wv[q];
        if (a == 0x12345678u) t1_out[0] = 1;
        return;
    }
    // register dwords that can hold keys of the matrix (block-uniform; songs shorter than 1024 frames: the sweeps stop there)
    const int q_end = DIR ? min(NQ, (M + 31) >> 5) : min(NQ, 4 * ((N + 127) >> 7));
    if (!trivial) {
        {
            uint4 *hz = reinterpret_cast<uint4 *>(sm.hist);
            hz[t] = make_uint4(0u, 0u, 0u, 0u);
            hz[t + R16_THREADS] = make_uint4(0u, 0u, 0u, 0u);
            uint4 *fz = reinterpret_cast<uint4 *>(sm.fine);
            fz[t] = make_uint4(0u, 0u, 0u, 0u);
            if (t < (int)(sizeof(sm.fine) / 16) - R16_THREADS) fz[t + R16_THREADS] = make_uint4(0u, 0u, 0u, 0u);
            sm.sub[t] = 0u;
            if (t == 0) { sm.sweep3 = 0u; sm.item_n = 0u; }
            if (t < R16_LINES) { sm.dec[t] = 0x7FFFFFFFu; sm.rcnt[t] = 0u; sm.ditem[t] = -1; sm.nr[t] = 0; }
        }
        lds_barrier();
        // ---- pass 1: histogram over the high bytes
        {
            const unsigned lanebase = r16_lds_off(sm.hist) + 4u * (unsigned)(DIR ? pi : (rr & (R16_WORDS - 1)));
            const unsigned vlo = DIR ? 1u : (rr < R16_WORDS ? 1u : 0x10000u), vhi = DIR ? 0x10000u : vlo;
            auto bin = [&](const int q) {
                const unsigned x = wv[q];
                r16_lds_add(lanebase + ((x >> R16_BINSHIFT) & R16_BINMASK), vlo);
                r16_lds_add(lanebase + ((x >> (16 + R16_BINSHIFT)) & R16_BINMASK), vhi);
            };
            if (q_end == NQ) {                              // (block-uniform: the full-size form has no tests inside)
#pragma unroll
                for (int q = 0; q < NQ; q++) bin(q);
            } else {
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    if (q < q_end) bin(q);
            }
        }
        lds_barrier();
        if (dbg == 2) { if (sm.hist[t] == 0x12345678u) t1_out[0] = 1; return; }
        // ---- decode 1
        {
            unsigned s = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) s += sm.hist[(8 * rs + b) * R16_WORDS + pi];
            unsigned *S = sm.S;
            S[rs * R16_LINES + (DIR ? 2 * pi : pi)] = s & 0xFFFFu;
            S[rs * R16_LINES + (DIR ? 2 * pi + 1 : pi + R16_WORDS)] = s >> 16;
        }
        lds_barrier();
        if (t < R16_LINES) r16_decode1<DIR>(sm, t, k);
        lds_barrier();
        if (dbg == 3) { if (sm.lineB[t & (R16_LINES - 1)] == 0x12345678) t1_out[0] = 1; return; }
        // ---- pass 2: the keys inside each line's window: counted by sixteens and by key, listed with their positions
        {
            const int la = DIR ? 2 * pi : rr, lb2 = DIR ? 2 * pi + 1 : rr;
            const unsigned b0 = (unsigned)(256 * sm.lineB[la] - 1) & 0xFFFFu, b1 = (unsigned)(256 * sm.lineB[lb2] - 1) & 0xFFFFu;
            const u16x2 bsh = k16_from_u32(b0 | (b1 << 16));
            // A hit is only KEPT inside the sweep -- x | position << 9 | line << 20 into the thread's own column of the slot table
            // that takes the place of the histogram -- and counted afterwards, all lanes their q-th hit together: a block of the
            // sweep runs for the whole wave when one lane hits (0.78 times per dword), the counting once per hit.
            // (lines past the end of the matrix -- padding, or whatever lies behind a row -- take no part)
            const unsigned lim_a = l0 + la < n_lines ? R16_XMAX + 1u : 0u, lim_b = l0 + lb2 < n_lines ? R16_XMAX + 1u : 0u;
            unsigned hc = 0;
            const unsigned slotbase = r16_lds_off(sm.hist) + 4u * (unsigned)t;
            auto count = [&](const unsigned rec) {
                const unsigned x = rec & 0x1FFu, line = rec >> R16_LINE_SHIFT;
                if (x - 1u < 256u) atomicAdd(&sm.sub[((x - 1u) >> 4) * R16_LINES + line], 1u);
                atomicAdd(&sm.fine[x * (R16_LINES / 4) + (line >> 2)], 1u << (8 * (line & 3)));
            };
            auto hit = [&](const unsigned rec) {
                if (hc < (unsigned)R16_SLOTS) *(r16_lds_word *)(uintptr_t)(slotbase + hc * (4u * R16_THREADS)) = rec;
                else {
                    count(rec);
                    sm.sweep3 = 1u;
                }
                hc++;
            };
            const unsigned rec_a = (unsigned)la << R16_LINE_SHIFT, rec_b = ((unsigned)lb2 << R16_LINE_SHIFT) | (DIR ? 0u : 1u << 9);
            auto test = [&](const int q) {
                const unsigned x = k16_to_u32(k16_from_u32(wv[q]) - bsh);
                const unsigned xl = x & 0xFFFFu, xh = x >> 16;
                const unsigned pos = DIR ? (unsigned)(rs + 32 * q) : (unsigned)(8 * (jj + 16 * (q >> 2)) + 2 * (q & 3));
                // (a real branch around both hit blocks: hipcc predicates them otherwise -- eight vector instructions and two LDS
                //  writes per dword with all lanes off; some lane of the wave hits in 0.63 of the dwords)
                const bool ha = xl < lim_a, hb = xh < lim_b;
#if R16_HIT_BRANCH
                if (__ballot(ha | hb) != 0ull) {
                    if (ha) hit((xl | (pos << 9)) + rec_a);
                    if (hb) hit((xh | (pos << 9)) + rec_b);
                }
#else
                if (ha) hit((xl | (pos << 9)) + rec_a);
                if (hb) hit((xh | (pos << 9)) + rec_b);
#endif
            };
            if (q_end == NQ) {
#pragma unroll
                for (int q = 0; q < NQ; q++) test(q);
            } else {
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    if (q < q_end) test(q);
            }
            const unsigned kept = min(hc, (unsigned)R16_SLOTS);
            sm.S[t] = kept;
            for (unsigned q = 0; q < (unsigned)R16_SLOTS; q++) {
                if (__ballot(q < kept) == 0) break;
                if (q < kept) count(*(const r16_lds_word *)(uintptr_t)(slotbase + q * (4u * R16_THREADS)));
            }
        }
        lds_barrier();
        if (dbg == 4) { if (sm.S[t] == 0x12345678u) t1_out[0] = 1; return; }
        // ---- decode 2
        {
            const unsigned koff = koff_of[p];
            const float *pair_band = band + 2 * p;
            const bool adjacent_ok = k16_reach_adjacent_ok(koff, pair_band);
            const int line = t >> 4;
            r16_decode2<DIR>(sm, line, t & 15, l0 + line < n_lines, k, p, l0 + line, item0 + R16_TILE_ITEMS * lb, lb, koff, adjacent_ok,
                             pair_band, w, t1_out, item_out, dbg);
            lds_barrier();
            if (t == 0) w.tile_used[item0 / R16_TILE_ITEMS + lb] = (int)min(sm.item_n, (unsigned)R16_TILE_ITEMS);
        }
        // ---- the items' cells: every thread looks through its OWN hits (one on average) for those of a line with an item whose x lies in
        // the item's reach -- all 512 threads at once; sixteen lanes per line walking the slots of the 16 or 32 threads that own
        // the line's keys took 0.2-0.4 ms of each kernel (a block waits for its slowest wave)
        if (sm.sweep3 == 0u) {
            const unsigned kept = sm.S[t];
            const unsigned slotbase = r16_lds_off(sm.hist) + 4u * (unsigned)t;
            for (unsigned q = 0; q < (unsigned)R16_SLOTS; q++) {
                if (__ballot(q < kept) == 0) break;
                if (q < kept) {
                    const unsigned rec = *(const r16_lds_word *)(uintptr_t)(slotbase + q * (4u * R16_THREADS));
                    const unsigned line = rec >> R16_LINE_SHIFT, d = sm.dec[line];
                    const unsigned x = rec & 0x1FFu, lo = (d >> 1) & 0x1FFu, hi = (d >> 10) & 0x1FFu;
                    if (((d & 0x80000001u) == 0x80000001u) & (x >= lo) & (x <= hi)) {
                        const unsigned slot = atomicAdd(&sm.rcnt[line], 1u);
                        if (slot < (unsigned)R16_CAP) (w.items + sm.ditem[line])->pos[slot] = (uint16_t)((rec >> 9) & R16_POS_MASK);
                    }
                }
            }
        }
        // ---- sweep 3, only in a block where some thread had more hits than slots: the items' cells straight from the registers
        if (sm.sweep3 != 0u) {
            const int la = DIR ? 2 * pi : rr, lb2 = DIR ? 2 * pi + 1 : rr;
            const unsigned b0 = (unsigned)(256 * sm.lineB[la] - 1) & 0xFFFFu, b1 = (unsigned)(256 * sm.lineB[lb2] - 1) & 0xFFFFu;
            const u16x2 bsh = k16_from_u32(b0 | (b1 << 16));
            const unsigned da = sm.dec[la], db = sm.dec[lb2];
            const bool ha = (da & 0x80000001u) == 0x80000001u, hb = (db & 0x80000001u) == 0x80000001u;
            const unsigned lo_a = (da >> 1) & 0x1FFu, lo_b = (db >> 1) & 0x1FFu;
            const unsigned wid_a = ha ? ((da >> 10) & 0x1FFu) - lo_a + 1u : 0u, wid_b = hb ? ((db >> 10) & 0x1FFu) - lo_b + 1u : 0u;
            R16Item *ia = w.items + (ha ? sm.ditem[la] : 0), *ib = w.items + (hb ? sm.ditem[lb2] : 0);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                if (q < q_end) {
                    const unsigned x = k16_to_u32(k16_from_u32(wv[q]) - bsh);
                    const unsigned xl = x & 0xFFFFu, xh = x >> 16;
                    const unsigned pos = DIR ? (unsigned)(rs + 32 * q) : (unsigned)(8 * (jj + 16 * (q >> 2)) + 2 * (q & 3));
                    if (xl - lo_a < wid_a) {
                        const unsigned slot = atomicAdd(&sm.rcnt[la], 1u);
                        if (slot < (unsigned)R16_CAP) ia->pos[slot] = (uint16_t)pos;
                    }
                    if (xh - lo_b < wid_b) {
                        const unsigned slot = atomicAdd(&sm.rcnt[lb2], 1u);
                        if (slot < (unsigned)R16_CAP) ib->pos[slot] = (uint16_t)(DIR ? pos : pos + 1u);
                    }
                }
            }
        }
        // every item must have found exactly its cells (slots or third sweep): anything else hands the pair back
        lds_barrier();
        if (t < R16_LINES && l0 + t < n_lines && sm.ditem[t] >= 0 && (int)sm.rcnt[t] != sm.nr[t]) {
            w.pair_flag[p] = 1;
            atomicAdd(&w.counters[1], 1);
            atomicAdd(&w.counters[8 + 9], 1);
        }
    } else {                                                // (rows only)
        if (t < R16_LINES) {
            sm.t1[t] = k <= 0 ? 0u : 0xFFFFu;
            if (l0 + t < n_lines) { t1_out[l0 + t] = k <= 0 ? (uint16_t)0 : (uint16_t)0xFFFFu; item_out[l0 + t] = -1; }
        }
        if (t == 0) w.tile_used[item0 / R16_TILE_ITEMS + lb] = 0;
        lds_barrier();
    }
    if (DIR == 1) return;
    if (dbg == 5) return;
    // ---- the mask's base bits: one byte per piece into LDS, then the tile's 64 x 128 bytes leave as one contiguous run
    {
        const u16x2 one = k16_opaque_ones();
        const u16x2 tr = k16_splat(sm.t1[rr]);
        unsigned char *mb = reinterpret_cast<unsigned char *>(sm.hist);
        constexpr int MBS = 4 * NQ + 16;                    // bytes per row (144 for 1024 columns: the sixteen rows of a wave instruction meet two to a bank)
#pragma unroll
        for (int q = 0; q < NQ / 4; q++) {
            const int piece = jj + 16 * q;
            const uint4 tc = sm.tcol[piece];
            const unsigned tcv[4] = {tc.x, tc.y, tc.z, tc.w};
            unsigned acc = 0;
            if (4 * q < q_end) {                            // (block-uniform; pieces behind the last column: zero bits)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const u16x2 m1 = __builtin_elementwise_min(k16_from_u32(tcv[d]), tr);
                    const u16x2 f = __builtin_elementwise_min(__builtin_elementwise_sub_sat(m1, k16_from_u32(wv[4 * q + d])), one);
                    acc |= k16_to_u32(f) << (2 * d);
                }
            }
            mb[rr * MBS + piece] = (unsigned char)((acc | (acc >> 15)) & 0xFFu);
        }
        lds_barrier();
        constexpr int RQ = NQ / 4;                          // 16-byte runs per row that hold columns (NQ / 2 words); the row's other words: zero
        const int rsh = wpr == 16 ? 3 : 4;                  // a row of the mask: wpr / 2 runs
        for (int u = t; u < (R16_LINES << rsh); u += R16_THREADS) {
            const int row = u >> rsh, part = u & ((1 << rsh) - 1);
            if (l0 + row < M)
                reinterpret_cast<uint4 *>(bits)[(((int64_t)p * w.max_m + l0) << rsh) + u] =
                    part < RQ ? reinterpret_cast<const uint4 *>(mb)[row * (MBS / 16) + part] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

// ---- exact values for the work items -------------------------------------------------------------------------------------------------
// Items live at R16_TILE_ITEMS * tile + slot (slot < tile_used[tile]) and behind w.static_items (counters[0] of them).  A wave
// takes 64 consecutive indices, finds the ones that hold an item of an unflagged pair and works through them.
// ---- exact values for the work items -------------------------------------------------------------------------------------------------
// The arithmetic of fix_row_band_range (planar_select.h): per cell nine terms -- FMA chain over the bins of the rolled x frame,
// exact_term() -- added in window order in float64; within an item the `need` smallest cells by (value, position) are selected.
// lanes (g, kk), g < 7: term kk of cell g; is_row: the cell is (which, pos), else (pos, which)
template <typename FT>
__device__ inline void r16_exact_term(const FT *__restrict__ feats, const FT *__restrict__ norms, int d, const acoss_pair_desc &ds,
                                      bool is_row, int which, int pos, int kk, double *out)
{
    const int i = is_row ? which : pos, j = is_row ? pos : which;
    const FT *x = feats + (ds.x_row0 + i + kk) * d, *y = feats + (ds.y_row0 + j + kk) * d;
    // (all of a frame pair's loads in flight at once; d <= FIX_MAXD)
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
    *out = exact_term(acc, nn);
}

struct R16ExactSmem {
    double cval[64];
    unsigned long long key[64];
    int posv[64], citem[64];
    uint4 rec[R16_TILE_ITEMS * 5];               // a tile's item records (80 bytes each)
};

// one item, the wave together (items with more than a handful of cells, and the items behind the tiles' own slots)
template <typename FT>
__device__ inline void r16_exact_item(R16ExactSmem &sm, R16Item *item, const FT *__restrict__ feats, const FT *__restrict__ norms, int d,
                                      const acoss_pair_desc &ds, int win, int lane)
{
    const int n = min(item->n, R16_CAP), need = item->need, dir = item->dir, which = item->which;
    if (lane < n) sm.posv[lane] = (int)item->pos[lane];
    __syncthreads();
    for (int e0 = 0; e0 < n; e0 += 7) {
        const int g = lane / 9, kk = lane - 9 * g, el = e0 + g;
        if (lane < 63 && el < n) r16_exact_term<FT>(feats, norms, d, ds, dir == 0, which, sm.posv[el], kk, &sm.cval[g * 9 + kk]);
        __syncthreads();
        if (lane < 7 && e0 + lane < n) {
            double s_ = 0.0;
            for (int q = 0; q < win; q++) s_ += sm.cval[lane * 9 + q];
            sm.key[e0 + lane] = f64_key(s_);
        }
        __syncthreads();
    }
    int rank = 1;
    unsigned long long mykey = 0;
    int mypos = 0;
    if (lane < n) { mykey = sm.key[lane]; mypos = sm.posv[lane]; }
    for (int m = 0; m < n; m++) {
        const unsigned long long km = sm.key[m];
        const int pm = sm.posv[m];
        rank += ((km < mykey) | ((km == mykey) & (pm < mypos))) ? 1 : 0;
    }
    const unsigned long long sel = __ballot(lane < n && rank <= need);
    if (lane == 0) { item->sel_lo = (unsigned)sel; item->sel_hi = (unsigned)(sel >> 32); }
    __syncthreads();
}

// A wave per TILE: the tile's eight item records arrive in one round trip (the pair follows from the tile's index), the cells of
// all its items are evaluated together -- two dependent round trips to memory per tile instead of four per item.
template <typename FT>
__global__ __launch_bounds__(64) void r16_exact_tiles_kernel(const FT *__restrict__ feats, const FT *__restrict__ norms, int d,
                                                             const acoss_pair_desc *__restrict__ descs, int win, R16Work w,
                                                             int row_tiles, int rb, int cb)
{
    __shared__ R16ExactSmem sm;
    const int lane = threadIdx.x;
    const int n_tiles = w.static_items / R16_TILE_ITEMS;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int used = min(w.tile_used[tile], R16_TILE_ITEMS);
        if (used <= 0) continue;
        const int p = tile < row_tiles ? tile / rb : (tile - row_tiles) / cb;
        if (w.pair_flag[p]) continue;
        R16Item *items = w.items + (int64_t)R16_TILE_ITEMS * tile;
        const uint4 *src = reinterpret_cast<const uint4 *>(items);
        static_assert(R16_TILE_ITEMS * 5 <= 64, "one load per lane");
        if (lane < R16_TILE_ITEMS * 5) sm.rec[lane] = src[lane];
        const acoss_pair_desc ds = descs[p];
        __syncthreads();
        // the tile's cells, item after item
        int start = 0, mine = -1, n_mine = 0, total = 0;
#pragma unroll
        for (int l = 0; l < R16_TILE_ITEMS; l++) {
            const int nl = l < used ? min((int)sm.rec[5 * l + 1].x, R16_CAP) : 0;
            if (lane >= total && lane < total + nl) { mine = l; start = total; n_mine = nl; }
            total += nl;
        }
        if (total > 64) {                                    // (rare: an item with many cells in reach) one item at a time
            for (int l = 0; l < used; l++) r16_exact_item<FT>(sm, items + l, feats, norms, d, ds, win, lane);
            continue;
        }
        if (mine >= 0) {
            sm.citem[lane] = mine;
            sm.posv[lane] = (int)reinterpret_cast<const uint16_t *>(&sm.rec[5 * mine + 2])[lane - start];
        }
        __syncthreads();
        for (int e0 = 0; e0 < total; e0 += 7) {
            const int g = lane / 9, kk = lane - 9 * g, el = e0 + g;
            if (lane < 63 && el < total) {
                const int l = sm.citem[el];
                const uint4 hdr = sm.rec[5 * l];
                r16_exact_term<FT>(feats, norms, d, ds, hdr.y == 0u, (int)hdr.z, sm.posv[el], kk, &sm.cval[g * 9 + kk]);
            }
            __syncthreads();
            if (lane < 7 && e0 + lane < total) {
                double s_ = 0.0;
                for (int q = 0; q < win; q++) s_ += sm.cval[lane * 9 + q];
                sm.key[e0 + lane] = f64_key(s_);
            }
            __syncthreads();
        }
        bool sel = false;
        if (mine >= 0) {
            const unsigned long long mykey = sm.key[lane];
            const int mypos = sm.posv[lane], need = (int)sm.rec[5 * mine].w;
            int rank = 1;
            for (int m = start; m < start + n_mine; m++) {
                const unsigned long long km = sm.key[m];
                const int pm = sm.posv[m];
                rank += ((km < mykey) | ((km == mykey) & (pm < mypos))) ? 1 : 0;
            }
            sel = rank <= need;
        }
        const unsigned long long all = __ballot(sel);
        // lane l < used writes item l's selection: its cells are lanes [start_l, start_l + n_l)
        {
            int st = 0;
            unsigned long long m_l = 0;
#pragma unroll
            for (int l = 0; l < R16_TILE_ITEMS; l++) {
                const int nl = l < used ? min((int)sm.rec[5 * l + 1].x, R16_CAP) : 0;
                if (l == lane) m_l = nl >= 64 ? all : ((all >> st) & ((1ull << nl) - 1ull));
                st += nl;
            }
            if (lane < used) { items[lane].sel_lo = (unsigned)m_l; items[lane].sel_hi = (unsigned)(m_l >> 32); }
        }
        __syncthreads();
    }
}

// the items behind the tiles' own slots (tiles with more than R16_TILE_ITEMS unclean lines: a few hundred per batch)
template <typename FT>
__global__ __launch_bounds__(64) void r16_exact_extra_kernel(const FT *__restrict__ feats, const FT *__restrict__ norms, int d,
                                                             const acoss_pair_desc *__restrict__ descs, int win, R16Work w)
{
    __shared__ R16ExactSmem sm;
    const int lane = threadIdx.x;
    const int total = min(w.static_items + w.counters[0], w.item_cap);
    for (int it = w.static_items + blockIdx.x; it < total; it += gridDim.x) {
        R16Item *item = w.items + it;
        const int p = item->p;
        if (w.pair_flag[p]) continue;
        const acoss_pair_desc ds = descs[p];
        r16_exact_item<FT>(sm, item, feats, norms, d, ds, win, lane);
    }
}

// ---- the cells the items select, where the other side selects them too (lane = item) ---------------------------------------------------
__global__ __launch_bounds__(64) void r16_apply_kernel(const uint16_t *__restrict__ keys, const acoss_pair_desc *__restrict__ descs,
                                                       R16Work w, int ldm, int ldn, int mutual, uint64_t *__restrict__ bits, int wpr)
{
    const int total = min(w.static_items + w.counters[0], w.item_cap);
    for (int s = blockIdx.x * 64 + threadIdx.x; s < total; s += 64 * gridDim.x) {
        if (s < w.static_items && (s & (R16_TILE_ITEMS - 1)) >= w.tile_used[s / R16_TILE_ITEMS]) continue;
        const R16Item *item = w.items + s;
        const int p = item->p, dir = item->dir, which = item->which;
        if (w.pair_flag[p]) continue;
        const uint64_t sel = ((uint64_t)item->sel_hi << 32) | item->sel_lo;
        const acoss_pair_desc ds = descs[p];
        for (uint64_t rest = sel; rest != 0; rest &= rest - 1) {
            const int pos = (int)item->pos[__ffsll((unsigned long long)rest) - 1];
            const int i = dir == 0 ? which : pos, j = dir == 0 ? pos : which;
            if (i >= ds.nx - 8 || j >= ds.ny - 8) continue;             // (cannot happen: positions come from the selection kernels)
            bool other = true;
            if (mutual) {
                const unsigned key = keys[ds.crp_off + (int64_t)i * ds.crp_pitch + j];
                const int oi = dir == 0 ? w.item_col[(int64_t)p * ldn + j] : w.item_row[(int64_t)p * ldm + i];
                const unsigned t1 = dir == 0 ? w.t1_col[(int64_t)p * ldn + j] : w.t1_row[(int64_t)p * ldm + i];
                other = key < t1;
                if (!other && oi >= 0) {
                    const R16Item *o = w.items + oi;
                    const uint64_t osel = ((uint64_t)o->sel_hi << 32) | o->sel_lo;
                    const int want = dir == 0 ? i : j;
                    for (int m = 0; m < min(o->n, R16_CAP); m++) other |= ((osel >> m) & 1) && (int)o->pos[m] == want;
                }
            }
            if (other) atomicOr(reinterpret_cast<unsigned long long *>(bits) + ((int64_t)p * w.max_m + i) * wpr + (j >> 6), 1ull << (j & 63));
        }
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
    return r16_work_bytes(K, max_m, max_n);
}

extern "C" int acoss_radix16_layout(void *work, int K, int max_nx, int max_ny, int win, void **ptrs, int *dims)
{
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    const int ldm = (max_m + 7) & ~7, ldn = (max_n + 7) & ~7;
    const R16Work w = r16_work_layout(work, K, max_m, max_n);
    ptrs[0] = w.t1_row; ptrs[1] = w.t1_col; ptrs[2] = w.item_row; ptrs[3] = w.item_col; ptrs[4] = w.counters; ptrs[5] = w.items;
    ptrs[6] = w.pair_flag; ptrs[7] = w.pair_list;
    dims[0] = ldm; dims[1] = ldn; dims[2] = w.item_cap; dims[3] = (int)sizeof(R16Item);
    return ACOSS_OK;
}

// what = 1: columns; 2: rows + base bits (needs the columns' t1 when mutual); 4: item list + exact values (| 64: stop there)
// + the items' cells + the list of flagged pairs.  Bits of `what` may be combined; bits 8-11 / 12-15: development phase cuts of
// the column / row kernel.
namespace acoss {

template <typename FT>
int r16_run(int what, const uint16_t *keys16, const float *band, const uint32_t *koff, const FT *feats, const FT *norms, int d,
            const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits,
            void *work, hipStream_t st)
{
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    const int ldm = (max_m + 7) & ~7, ldn = (max_n + 7) & ~7;
    R16Work w = r16_work_layout(work, K, max_m, max_n);
    double kv;
    int mode;
    if (kappa == 0.0) { kv = 0.0; mode = 2; } else if (kappa < 1.0) { kv = kappa; mode = 0; } else { kv = kappa; mode = 1; }
    const int cb = ceil_div(max_n, R16_LINES), rb = ceil_div(max_m, R16_LINES);
    if ((int64_t)K * cb > 0x7fffffffLL || (int64_t)K * rb > 0x7fffffffLL) { set_error("radix16: batch too large"); return ACOSS_ENOTSUP; }
    if (max_m > 2048 || max_n > 2048) { set_error("radix16: matrices up to 2048 x 2048"); return ACOSS_ENOTSUP; }
    // long forms (a side beyond 1024): 48 or 64 dwords of keys per thread; 32 words per row of the mask when either side is long
    const int wpr = mask_bits_words(max_m, max_n);
    if (what & 1) {
        ACOSS_HIP(hipMemsetAsync(w.counters, 0, 256, st));
        ACOSS_HIP(hipMemsetAsync(w.pair_flag, 0, (size_t)K, st));
        if (mutual) {
#define R16_LAUNCH_SELECT(DIR_, NQ_, blocks_, tiles_, item0_, dbg_)                                                                              \
    hipLaunchKernelGGL((r16_select_kernel<DIR_, NQ_>), dim3((unsigned)(blocks_)), dim3(R16_THREADS), 0, st, keys16, descs, win, kv, mode, w, ldm, \
                       ldn, tiles_, band, koff, mutual, bits, wpr, item0_, dbg_)
            const int nq = max_m <= 1024 ? 32 : (max_m <= 1536 ? 48 : 64);         // dwords of keys per thread: the columns' positions are rows
            if (nq == 32) R16_LAUNCH_SELECT(1, 32, (int64_t)K * cb, cb, R16_TILE_ITEMS * K * rb, (what >> 8) & 15);
            else if (nq == 48) R16_LAUNCH_SELECT(1, 48, (int64_t)K * cb, cb, R16_TILE_ITEMS * K * rb, (what >> 8) & 15);
            else R16_LAUNCH_SELECT(1, 64, (int64_t)K * cb, cb, R16_TILE_ITEMS * K * rb, (what >> 8) & 15);
            const int rc = launch_check("r16_select_kernel<columns>");
            if (rc) return rc;
        } else ACOSS_HIP(hipMemsetAsync(w.tile_used + (size_t)K * rb, 0, (size_t)K * cb * sizeof(int), st));
    }
    if (what & 2) {
        const int nq = max_n <= 1024 ? 32 : (max_n <= 1536 ? 48 : 64);
        if (nq == 32) R16_LAUNCH_SELECT(0, 32, (int64_t)K * rb, rb, 0, (what >> 12) & 15);
        else if (nq == 48) R16_LAUNCH_SELECT(0, 48, (int64_t)K * rb, rb, 0, (what >> 12) & 15);
        else R16_LAUNCH_SELECT(0, 64, (int64_t)K * rb, rb, 0, (what >> 12) & 15);
        const int rc = launch_check("r16_select_kernel<rows>");
        if (rc) return rc;
    }
    if (what & 4) {
        hipLaunchKernelGGL(r16_exact_tiles_kernel<FT>, dim3(8192), dim3(64), 0, st, feats, norms, d, descs, win, w, K * rb, rb, cb);
        hipLaunchKernelGGL(r16_exact_extra_kernel<FT>, dim3(2048), dim3(64), 0, st, feats, norms, d, descs, win, w);      // (none on the benchmark; smooth features: 10^5 items)
        if (what & 64) return launch_check("r16_exact_tiles_kernel");
        hipLaunchKernelGGL(r16_apply_kernel, dim3(4096), dim3(64), 0, st, keys16, descs, w, ldm, ldn, mutual, bits, wpr);
        hipLaunchKernelGGL(r16_flag_list_kernel, dim3(1), dim3(256), 0, st, w, K);
        const int rc = launch_check("r16_exact_kernel / r16_apply_kernel");
        if (rc) return rc;
    }
    return ACOSS_OK;
}

template int r16_run<double>(int, const uint16_t *, const float *, const uint32_t *, const double *, const double *, int, const acoss_pair_desc *,
                             int, int, int, int, double, int, uint64_t *, void *, hipStream_t);
template int r16_run<float>(int, const uint16_t *, const float *, const uint32_t *, const float *, const float *, int, const acoss_pair_desc *,
                            int, int, int, int, double, int, uint64_t *, void *, hipStream_t);

}  // namespace acoss

extern "C" int acoss_radix16_stage(int what, const uint16_t *keys16, const float *band, const uint32_t *koff, const double *feats,
                                   const double *norms, int d, const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                                   double kappa, int mutual, uint64_t *bits, void *work, size_t work_bytes, void *stream)
{
    if (!keys16 || !band || !koff || !descs || !work || K < 0 || max_nx < win || max_ny < win || kappa < 0.0) {
        set_error("radix16_stage: bad argument");
        return ACOSS_EINVAL;
    }
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    if (max_m > 2048 || max_n > 2048) { set_error("radix16_stage: matrices up to 2048 x 2048"); return ACOSS_ENOTSUP; }
    if (work_bytes < acoss_radix16_work_bytes(K, max_nx, max_ny, win)) { set_error("radix16_stage: workspace too small"); return ACOSS_EINVAL; }
    if (K == 0) return ACOSS_OK;
    return r16_run<double>(what, keys16, band, koff, feats, norms, d, descs, K, win, max_nx, max_ny, kappa, mutual, bits, work, (hipStream_t)stream);
}
