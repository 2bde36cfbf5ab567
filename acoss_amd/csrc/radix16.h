// radix16.h -- workspace of the radix form of the kNN selection on the 16-bit key plane (radix16_kernels.hip, round 5).
//
// Every row and column leaves the selection kernels as ONE number and, rarely, one work item:
//     t1       cells whose key is < t1 are selected                                   (0: none, 0xFFFF: every cell)
//     item     -1, or the index of an R16Item: the cells of the row / column whose keys lie in the reach of the k-th smallest's
//              float32 error band (keys16.h: k16_reach) -- positions only, at most R16_CAP of them -- of which the `need`
//              smallest by EXACT value (position breaks ties) are selected too.  r16_exact_kernel fills `sel`.
// The mutual mask is then   key < min(t1_row[i], t1_col[j])   (written by the row kernel)   OR-ed with the cells the items select
// on one side that the other side selects too (r16_apply_kernel).
// A pair the kernels cannot express this way (more than R16_CAP cells inside the captured key window of a row or column: exact
// ties -- a threshold outside the window its reach needs -- no room for an item) is FLAGGED and redone by the wave-per-row
// kernels of keys16_kernels.hip.
#pragma once

#include "keys16.h"

namespace acoss {

constexpr int R16_CAP = 24;              // cells in reach an item can hold (2-4 on the benchmark; more: exact ties, the pair is handed back)

struct __attribute__((aligned(16))) R16Item {
    int p, dir, which, need;
    int n;
    unsigned sel_lo, sel_hi;             // bit m: cell pos[m] is selected (r16_exact_kernel)
    int reserved;
    uint16_t pos[R16_CAP];
};                                        // 80 bytes

struct R16Work {
    uint16_t *t1_row, *t1_col;           // [K][max_m], [K][max_n]
    int *item_row, *item_col;
    int *counters;                       // [0] items beyond the tiles' own slots, [1] lines that flagged their pair (statistics), [2] pairs in pair_list, [8 + why] why lines flagged their pair
    int *tile_used;                      // [row tiles of every pair][column tiles of every pair]: items in the tile's R16_TILE_ITEMS slots
    int static_items;                    // R16_TILE_ITEMS x tiles: item index = R16_TILE_ITEMS * tile + slot; the rest of `items` is asked for one by one
    int *pair_list;                      // [K]: the flagged pairs, compacted (r16_flag_list_kernel)
    unsigned char *pair_flag;            // [K]
    R16Item *items;
    int item_cap;
    int max_m, max_n;
};

static_assert(sizeof(R16Item) == R16_ITEM_BYTES, "thresh_work.h sizes the workspace");

inline R16Work r16_work_layout(void *work, int K, int max_m, int max_n)
{
    const int ldm = (max_m + 7) & ~7, ldn = (max_n + 7) & ~7;
    R16Work w;
    uintptr_t a = ((uintptr_t)work + 255) & ~(uintptr_t)255;
    w.counters = (int *)a;                  a += 256;
    w.t1_row = (uint16_t *)a;               a += r16_align((size_t)K * ldm * sizeof(uint16_t));
    w.t1_col = (uint16_t *)a;               a += r16_align((size_t)K * ldn * sizeof(uint16_t));
    w.item_row = (int *)a;                  a += r16_align((size_t)K * ldm * sizeof(int));
    w.item_col = (int *)a;                  a += r16_align((size_t)K * ldn * sizeof(int));
    w.pair_list = (int *)a;                 a += r16_align((size_t)K * sizeof(int));
    w.pair_flag = (unsigned char *)a;       a += r16_align((size_t)K);
    w.tile_used = (int *)a;                 a += r16_align((size_t)r16_tiles(K, max_m, max_n) * sizeof(int));
    w.items = (R16Item *)a;
    w.item_cap = r16_item_cap(K, max_m, max_n);
    w.static_items = R16_TILE_ITEMS * r16_tiles(K, max_m, max_n);
    w.max_m = max_m;
    w.max_n = max_n;
    return w;
}

// the selection on the keys of one batch (radix16_kernels.hip).  what: 1 columns, 2 rows + base bits, 4 exact values + the items'
// cells + the list of flagged pairs (w.pair_list, w.counters[2]); `work`: r16_work_bytes() bytes (ThreshWork::radix)
template <typename FT>
int r16_run(int what, const uint16_t *keys16, const float *band, const uint32_t *koff, const FT *feats, const FT *norms, int d,
            const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits,
            void *work, hipStream_t st);

}  // namespace acoss
