// gemm_f64.h -- one 128 x 128 output tile of C = A . B^T (both operands row-major along the contraction axis) on
// v_mfma_f64_16x16x4_f64, shared by the wide-feature CSM, the FTM2D all-pairs similarity and the cross-diffusion
// products of the similarity network fusion.
//
// 512 threads = 8 waves, wave tile 32 x 64 (2 x 4 MFMA tiles, 64 accumulator VGPRs); the contraction axis is walked
// in chunks of 32 staged through LDS (row stride 36 doubles: the 64-lane operand reads take their minimum two
// passes), with the next chunk's 16 values per thread fetched into registers while the current one is multiplied.
// A 128 x 128 tile does 16 flop per byte staged, which keeps the L2 -> LDS traffic under 5 TB/s at the float64
// matrix peak; the 64 x 64 tile of the first version (8 flop / byte) stalled at half of it.
// Accumulation is in contraction order (k ascending, one FMA per k inside the MFMA), i.e. the same chain as a
// scalar loop -- kernels built on this agree bit for bit with their scalar counterparts.
#pragma once

#include <hip/hip_runtime.h>

namespace acoss {

typedef double v4f64_g __attribute__((ext_vector_type(4)));
constexpr int GM_T = 128, GM_KC = 32, GM_LD = GM_KC + 4, GM_THREADS = 512;

struct GemmSmem {
    double a[GM_T][GM_LD];
    double b[GM_T][GM_LD];
};

// loadA(r, k) / loadB(r, k): element k of row r (0..127) of the tile's operand, 0.0 outside the matrix;
// store(i, j, v): the finished element (i, j) of the tile (0..127 each); called for every element.
template <typename LoadA, typename LoadB, typename Store>
__device__ inline void gemm_nt_tile_f64(GemmSmem &sm, int kdim, LoadA loadA, LoadB loadB, Store store)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 64;
    v4f64_g acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (v4f64_g){0.0, 0.0, 0.0, 0.0};
    double ra[8], rb[8];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int e = threadIdx.x + GM_THREADS * q;
            ra[q] = loadA(e >> 5, k0 + (e & 31));
            rb[q] = loadB(e >> 5, k0 + (e & 31));
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < kdim; k0 += GM_KC) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int e = threadIdx.x + GM_THREADS * q;
            sm.a[e >> 5][e & 31] = ra[q];
            sm.b[e >> 5][e & 31] = rb[q];
        }
        __syncthreads();
        if (k0 + GM_KC < kdim) fetch(k0 + GM_KC);
#pragma unroll
        for (int kk = 0; kk < GM_KC; kk += 4) {
            double a[2], b[4];
#pragma unroll
            for (int t = 0; t < 2; t++) a[t] = sm.a[wi + 16 * t + lr][kk + lk];
#pragma unroll
            for (int t = 0; t < 4; t++) b[t] = sm.b[wj + 16 * t + lr][kk + lk];
#pragma unroll
            for (int ta = 0; ta < 2; ta++)
#pragma unroll
                for (int tb = 0; tb < 4; tb++)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int ta = 0; ta < 2; ta++)
#pragma unroll
        for (int tb = 0; tb < 4; tb++)
#pragma unroll
            for (int r = 0; r < 4; r++) store(wi + 16 * ta + lk + 4 * r, wj + 16 * tb + lr, acc[ta][tb][r]);
}

// The same tile for operands that lie in memory as rows of `kdim` contiguous doubles (A: rows i0 .., leading dimension lda, rowsA of
// them exist; B likewise): what the lambdas of gemm_nt_tile_f64 cost per loaded value -- a 64-bit multiply-add for the address and
// two range tests, ~8 vector instructions, sixteen values per thread and chunk -- is the eighth of the loop that the float64
// matrix instruction does not hide (it runs on the vector pipe: 64 of them take 2048 cycles per wave and chunk, those loads 500).
// Here a thread keeps ONE 32-bit offset per operand and the tile's interior (every row present, whole chunks) loads 16 bytes at
// a time with one test per chunk (a ragged last chunk); tiles at the matrix edge take the general form.  Same contraction order.
template <typename Store>
__device__ inline void gemm_nt_tile_f64_rows(GemmSmem &sm, int kdim, const double *__restrict__ A, int lda, int rowsA,
                                             const double *__restrict__ B, int ldb, int rowsB, Store store)
{
    const bool interior = rowsA >= GM_T && rowsB >= GM_T && (kdim & 1) == 0 && (lda & 1) == 0 && (ldb & 1) == 0 &&
                          ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0;            // block-uniform
    if (!interior) {
        gemm_nt_tile_f64(
            sm, kdim, [&](const int r, const int k) { return (r < rowsA && k < kdim) ? A[(int64_t)r * lda + k] : 0.0; },
            [&](const int r, const int k) { return (r < rowsB && k < kdim) ? B[(int64_t)r * ldb + k] : 0.0; }, store);
        return;
    }
    typedef double v2f64_g __attribute__((ext_vector_type(2)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 64;
    v4f64_g acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (v4f64_g){0.0, 0.0, 0.0, 0.0};
    // thread t: rows (t >> 4) + 32 q (q < 4), doubles 2 (t & 15), + 1 of the chunk
    const int tr = threadIdx.x >> 4, tk = 2 * (threadIdx.x & 15);
    const double *pa = A + (int64_t)tr * lda + tk, *pb = B + (int64_t)tr * ldb + tk;
    const int64_t sa = 32 * (int64_t)lda, sb = 32 * (int64_t)ldb;
    v2f64_g ra[4], rb[4];
    auto fetch = [&](const int k0) {
        const bool in = k0 + tk < kdim;                   // (the last chunk may be ragged: kdim is even, so pairs are whole)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ra[q] = in ? *reinterpret_cast<const v2f64_g *>(pa + q * sa + k0) : (v2f64_g){0.0, 0.0};
            rb[q] = in ? *reinterpret_cast<const v2f64_g *>(pb + q * sb + k0) : (v2f64_g){0.0, 0.0};
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < kdim; k0 += GM_KC) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            *reinterpret_cast<v2f64_g *>(&sm.a[tr + 32 * q][tk]) = ra[q];
            *reinterpret_cast<v2f64_g *>(&sm.b[tr + 32 * q][tk]) = rb[q];
        }
        __syncthreads();
        if (k0 + GM_KC < kdim) fetch(k0 + GM_KC);
#pragma unroll
        for (int kk = 0; kk < GM_KC; kk += 4) {
            double a[2], b[4];
#pragma unroll
            for (int t = 0; t < 2; t++) a[t] = sm.a[wi + 16 * t + lr][kk + lk];
#pragma unroll
            for (int t = 0; t < 4; t++) b[t] = sm.b[wj + 16 * t + lr][kk + lk];
#pragma unroll
            for (int ta = 0; ta < 2; ta++)
#pragma unroll
                for (int tb = 0; tb < 4; tb++)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int ta = 0; ta < 2; ta++)
#pragma unroll
        for (int tb = 0; tb < 4; tb++)
#pragma unroll
            for (int r = 0; r < 4; r++) store(wi + 16 * ta + lk + 4 * r, wj + 16 * tb + lr, acc[ta][tb][r]);
}

}  // namespace acoss
