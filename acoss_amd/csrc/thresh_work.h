// thresh_work.h -- layout of the kNN threshold workspace shared by the selection, mask and fused
// alignment kernels: per pair and per row (column) the order-preserving key of the k-th smallest value
// and the tie cut position.
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace acoss {

struct ThreshWork {
    uint64_t *row_thr;   // [K][max_m]
    uint64_t *col_thr;   // [K][max_n]
    int *row_cut;        // [K][max_m]
    int *col_cut;        // [K][max_n]
    int max_m, max_n;
    // optional (bit-mask path): the selected positions of every row / column, wpr x uint64 each (bit l of word e =
    // position 64 e + l); wpr = mask_bits_words(max_m, max_n)
    uint64_t *row_bits;  // [K][max_m][wpr]
    uint64_t *col_bits;  // [K][wpr][max_n]: word-major, so that word e of 64 neighbouring columns is one 512-byte run
                         // (what the transposing combine kernel reads); column j's word e = col_word(p, j, e)
    int wpr;
    // float32-approximate keys (strip32_kernels.hip): per pair two floats (base, slope): twice the error bound of an
    // approximate value v is base + slope * v; a row or column is resolved only if no other key lies within that distance
    // of its k-th smallest.  nullptr: exact high words.
    const float *band;
    // optional (float32-approximate keys): rows / columns the selection kernels could not decide hand their keys (position
    // order, 4 KB each) to the refinement kernel through these slots; side_counter[0] = slots asked for (beyond side_cap
    // the row stays marked unresolved in row_cut / col_cut and the strided fix-up kernels pick it up)
    int *side_counter;
    int4 *side_slots;       // {pair, direction, index, high word the selection left}
    uint32_t *side_keys;    // [side_cap][1024]
    int side_cap;
    void *radix;            // (bit-mask path up to 2048 x 2048) the radix selection's workspace: r16_work_layout(radix, K, max_m, max_n)
    __host__ __device__ uint64_t *col_word(int p, int j, int e) const { return col_bits + ((size_t)p * wpr + e) * max_n + j; }
};

// ---- workspace of the radix selection (radix16.h lays it out; up to 2048 x 2048): behind the side buffer / the bit planes ---------------------
// Every tile of lines owns R16_TILE_ITEMS item slots, one for four lines (no global atomic on the selection kernels' path; 4 % of
// the lines need an item on the benchmark, 15-25 % on temporally smooth features); a tile with more asks for single items behind
// them: room for 2 % of the lines.
#ifndef R16_TILE_LINES_V
#define R16_TILE_LINES_V 32              // lines (rows or columns) per block of the selection kernel: 32 (eight waves, four blocks per CU: measured 4.1 ms per 4096 pairs against 4.8) or 64 (sixteen)
#endif
constexpr int R16_TILE_LINES = R16_TILE_LINES_V;
constexpr int R16_TILE_ITEMS = R16_TILE_LINES / 4;
constexpr size_t R16_ITEM_BYTES = 80;

inline int r16_tiles(int K, int max_m, int max_n) { return K * ((max_m + R16_TILE_LINES - 1) / R16_TILE_LINES + (max_n + R16_TILE_LINES - 1) / R16_TILE_LINES); }

inline int r16_item_cap(int K, int max_m, int max_n)
{
    const double lines = (double)K * (double)(max_m + max_n);
    const double cap = (double)R16_TILE_ITEMS * (double)r16_tiles(K, max_m, max_n) + lines * 0.02 + 4096.0;
    return cap > 4.0e7 ? 40000000 : (int)cap;
}

inline size_t r16_align(size_t b) { return (b + 255) & ~(size_t)255; }

inline size_t r16_work_bytes(int K, int max_m, int max_n)
{
    const int ldm = (max_m + 7) & ~7, ldn = (max_n + 7) & ~7;      // row strides of the t1 / item arrays
    size_t b = 512;
    b += r16_align((size_t)K * ldm * sizeof(uint16_t)) + r16_align((size_t)K * ldn * sizeof(uint16_t));
    b += r16_align((size_t)K * ldm * sizeof(int)) + r16_align((size_t)K * ldn * sizeof(int));
    b += r16_align((size_t)K * sizeof(int)) + r16_align((size_t)K);
    b += r16_align((size_t)r16_tiles(K, max_m, max_n) * sizeof(int));
    b += r16_align((size_t)r16_item_cap(K, max_m, max_n) * R16_ITEM_BYTES);
    return b + 256;
}

// uint64 words per row of the bit planes and of the bit-packed mask: 16 up to 1024 x 1024, 32 up to 2048 x 2048
inline int mask_bits_words(int max_m, int max_n) { return (max_m > 1024 || max_n > 1024) ? 32 : 16; }

// capacity of the side buffer that travels with the bit planes: 2 % of the rows and columns (0.5 - 1 % ask for a slot)
inline int thresh_side_cap(int K, int max_m, int max_n)
{
    const double rows = (double)K * (double)(max_m + max_n);
    const double cap = rows * 0.02 < 1024.0 ? 1024.0 : rows * 0.02;
    return cap > 4.0e6 ? 4000000 : (int)cap;
}

inline size_t thresh_work_bytes(int K, int max_m, int max_n, bool with_bits)
{
    size_t b = (size_t)K * (size_t)(max_m + max_n) * (sizeof(uint64_t) + sizeof(int)) + 64;
    if (with_bits) {
        b += (size_t)K * (size_t)(max_m + max_n) * mask_bits_words(max_m, max_n) * sizeof(uint64_t) + 64;
        if (mask_bits_words(max_m, max_n) == 16)
            b += 512 + (size_t)thresh_side_cap(K, max_m, max_n) * (sizeof(int4) + 1024 * sizeof(uint32_t)) + 256 + r16_work_bytes(K, max_m, max_n);
        else if (max_m <= 2048 && max_n <= 2048)
            b += 512 + r16_work_bytes(K, max_m, max_n);
    }
    return b;
}

inline ThreshWork thresh_work_layout(void *work, int K, int max_m, int max_n, bool with_bits = false)
{
    ThreshWork w;
    w.max_m = max_m;
    w.max_n = max_n;
    w.row_thr = (uint64_t *)work;
    w.col_thr = w.row_thr + (size_t)K * max_m;
    w.row_cut = (int *)(w.col_thr + (size_t)K * max_n);
    w.col_cut = w.row_cut + (size_t)K * max_m;
    w.row_bits = nullptr;
    w.col_bits = nullptr;
    w.band = nullptr;
    w.side_counter = nullptr;
    w.side_slots = nullptr;
    w.side_keys = nullptr;
    w.side_cap = 0;
    w.radix = nullptr;
    w.wpr = mask_bits_words(max_m, max_n);
    if (with_bits) {
        uintptr_t a = (uintptr_t)(w.col_cut + (size_t)K * max_n);
        a = (a + 63) & ~(uintptr_t)63;
        w.row_bits = (uint64_t *)a;
        w.col_bits = w.row_bits + (size_t)K * max_m * w.wpr;
        if (w.wpr == 16) {
            a = (uintptr_t)(w.col_bits + (size_t)K * max_n * w.wpr);
            a = (a + 255) & ~(uintptr_t)255;
            w.side_counter = (int *)a;
            w.side_cap = thresh_side_cap(K, max_m, max_n);
            w.side_slots = (int4 *)(a + 256);
            w.side_keys = (uint32_t *)(a + 256 + (size_t)w.side_cap * sizeof(int4));
            w.radix = (void *)(((uintptr_t)(w.side_keys + (size_t)w.side_cap * 1024) + 255) & ~(uintptr_t)255);
        } else if (max_m <= 2048 && max_n <= 2048) {
            // (long form of the 16-bit keys: the radix selection alone -- no side buffer, the bit planes stay unused)
            w.radix = (void *)(((uintptr_t)(w.col_bits + (size_t)K * max_n * w.wpr) + 255) & ~(uintptr_t)255);
        }
    }
    return w;
}

}  // namespace acoss
