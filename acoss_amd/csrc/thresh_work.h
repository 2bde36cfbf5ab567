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
};

inline ThreshWork thresh_work_layout(void *work, int K, int max_m, int max_n)
{
    ThreshWork w;
    w.max_m = max_m;
    w.max_n = max_n;
    w.row_thr = (uint64_t *)work;
    w.col_thr = w.row_thr + (size_t)K * max_m;
    w.row_cut = (int *)(w.col_thr + (size_t)K * max_n);
    w.col_cut = w.row_cut + (size_t)K * max_m;
    return w;
}

}  // namespace acoss
