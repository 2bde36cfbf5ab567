// gemm_f32.h -- one output tile (128 x 128 or 256 x 128) of C = A . B^T (both operands row-major along the contraction axis) in float32
// on v_mfma_f32_16x16x4_f32 (exact float32 products and sums; 256 flop / clk / CU = 157 TFLOP/s on MI355X), for the
// float32 wide-feature CSM (the reference keeps its 20 736-dimensional scattering features in float32, Serra09.py:187).
//
// Same decomposition as gemm_f64.h: 512 threads = 8 waves, wave tile 32 x 64 (2 x 4 MFMA tiles, 32 accumulator VGPRs),
// the contraction axis walked in chunks of 32 staged through LDS with the next chunk's 16 values per thread on their way
// while the current one is multiplied.  What differs:
//   * a lane holds FOUR consecutive k of its row (one ds_read_b128) and feeds them to four successive MFMAs, i.e. MFMA s
//     of a group contracts k = 4 q + s over the four lane groups q -- a permutation of the contraction order, the same
//     for both operands.  Six 16-byte LDS reads feed 32 MFMAs.  (Float32 results are compared within a bound, not bit
//     for bit: numpy's own sgemm fixes no order either.)
//   * row stride 36 floats: the sixteen rows a 16-lane group reads start 4 banks apart (36 r mod 64 runs through the
//     multiples of 4), so every 256-byte pass of a ds_read_b128 is conflict-free, and the 16-byte stores of the staging
//     are too.
// 32 flop per staged byte: 4.9 TB/s of L2 -> LDS traffic at the matrix peak.
#pragma once

#include <hip/hip_runtime.h>

namespace acoss {

typedef float v4f32_g __attribute__((ext_vector_type(4)));
constexpr int GM32_KC = 32, GM32_LD = GM32_KC + 4, GM32_THREADS = 512;

// Output tile = (4 * 16 * WM) x (2 * 16 * WN): the 8 waves form a 4 x 2 grid, each with WM x WN MFMA tiles of 16 x 16.
// WM = 2, WN = 4: 128 x 128 (32 accumulator registers, 32 flop per staged byte); WM = 4, WN = 4: 256 x 128 (64
// accumulator registers, 43 flop per staged byte, 8 instead of 12 LDS reads per 64 MFMAs).
template <int WM, int WN>
struct Gemm32Smem {
    static constexpr int TM = 64 * WM, TN = 32 * WN;
    float a[TM][GM32_LD];
    float b[TN][GM32_LD];
};

// loadA(r, k) / loadB(r, k): elements k .. k+3 (k a multiple of 4) of row r of the tile's operand (r < TM resp. TN), 0
// outside the matrix; store(i, j, v): the finished element (i, j) of the tile; called for every element.
template <int WM, int WN, typename LoadA, typename LoadB, typename Store>
__device__ inline void gemm_nt_tile_f32(Gemm32Smem<WM, WN> &sm, int kdim, LoadA loadA, LoadB loadB, Store store)
{
    constexpr int TM = 64 * WM, TN = 32 * WN;
    constexpr int QA = TM * 8 / GM32_THREADS, QB = TN * 8 / GM32_THREADS;      // 16-byte quads per thread and chunk
    static_assert(QA >= 1 && QB >= 1, "tile too small for 512 threads");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * (16 * WM), wj = (wave & 1) * (16 * WN);
    v4f32_g acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; a++)
#pragma unroll
        for (int b = 0; b < WN; b++) acc[a][b] = (v4f32_g){0.f, 0.f, 0.f, 0.f};
    float4 ra[QA], rb[QB];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int q = 0; q < QA; q++) {
            const int e = threadIdx.x + GM32_THREADS * q;          // TM rows x 8 quads
            ra[q] = loadA(e >> 3, k0 + 4 * (e & 7));
        }
#pragma unroll
        for (int q = 0; q < QB; q++) {
            const int e = threadIdx.x + GM32_THREADS * q;
            rb[q] = loadB(e >> 3, k0 + 4 * (e & 7));
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < kdim; k0 += GM32_KC) {
#if !defined(GM32_PROBE) || GM32_PROBE != 2          // (timing-only builds: 1 no global loads inside the walk, 2 no staging and no barriers, 3 no products)
#pragma unroll
        for (int q = 0; q < QA; q++) {
            const int e = threadIdx.x + GM32_THREADS * q;
            *reinterpret_cast<float4 *>(&sm.a[e >> 3][4 * (e & 7)]) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < QB; q++) {
            const int e = threadIdx.x + GM32_THREADS * q;
            *reinterpret_cast<float4 *>(&sm.b[e >> 3][4 * (e & 7)]) = rb[q];
        }
        __syncthreads();
#endif
#if !defined(GM32_PROBE) || GM32_PROBE != 1
        if (k0 + GM32_KC < kdim) fetch(k0 + GM32_KC);
#endif
#if !defined(GM32_PROBE) || GM32_PROBE != 3
#pragma unroll
        for (int kk = 0; kk < GM32_KC; kk += 16) {
            float4 a[WM], b[WN];
#pragma unroll
            for (int t = 0; t < WM; t++) a[t] = *reinterpret_cast<const float4 *>(&sm.a[wi + 16 * t + lr][kk + 4 * lk]);
#pragma unroll
            for (int t = 0; t < WN; t++) b[t] = *reinterpret_cast<const float4 *>(&sm.b[wj + 16 * t + lr][kk + 4 * lk]);
#pragma unroll
            for (int s = 0; s < 4; s++) {
#pragma unroll
                for (int ta = 0; ta < WM; ta++)
#pragma unroll
                    for (int tb = 0; tb < WN; tb++) {
                        const float av = s == 0 ? a[ta].x : s == 1 ? a[ta].y : s == 2 ? a[ta].z : a[ta].w;
                        const float bv = s == 0 ? b[tb].x : s == 1 ? b[tb].y : s == 2 ? b[tb].z : b[tb].w;
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[ta][tb], 0, 0, 0);
                    }
            }
        }
#endif
#if !defined(GM32_PROBE) || GM32_PROBE != 2
        __syncthreads();
#endif
    }
    // accumulator register r of a lane = row 4 * lk + r, column lr of its 16 x 16 tile
#pragma unroll
    for (int ta = 0; ta < WM; ta++)
#pragma unroll
        for (int tb = 0; tb < WN; tb++)
#pragma unroll
            for (int r = 0; r < 4; r++) store(wi + 16 * ta + 4 * lk + r, wj + 16 * tb + lr, acc[ta][tb][r]);
}

// The same tile for operands that lie in memory as rows of `kdim` contiguous floats (A: rows from A on, leading dimension lda, rowsA of them
// exist; B likewise): one pointer per thread and staged quad instead of the lambdas' per-quad address arithmetic and range tests.  The
// float32 matrix instruction occupies the vector pipe, so every vector instruction of the staging path is time the products do not get
// (round 5, timing-only builds of the 20 736-d CSM, 28 pairs: whole kernel 10.7 ms; without the loads inside the walk 8.1; the multiply
// loop alone 8.0 = 0.91 of the float32 matrix peak).  Rows past the last one repeat it (their results are dropped by `store`); quads
// behind the end of the contraction axis are zero.  Needs kdim, lda, ldb multiples of 4 and 16-byte aligned bases (the caller checks).
template <int WM, int WN, typename Store>
__device__ inline void gemm_nt_tile_f32_rows(Gemm32Smem<WM, WN> &sm, int kdim, const float *__restrict__ A, int lda, int rowsA,
                                             const float *__restrict__ B, int ldb, int rowsB, Store store)
{
    constexpr int TM = 64 * WM, TN = 32 * WN;
    constexpr int QA = TM * 8 / GM32_THREADS, QB = TN * 8 / GM32_THREADS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 1) * (16 * WM), wj = (wave & 1) * (16 * WN);
    v4f32_g acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; a++)
#pragma unroll
        for (int b = 0; b < WN; b++) acc[a][b] = (v4f32_g){0.f, 0.f, 0.f, 0.f};
    // thread t stages quad (t & 7) of rows (t >> 3) + 64 q
    const int tr = threadIdx.x >> 3, tq = 4 * (threadIdx.x & 7);
    const float *pa[QA], *pb[QB];
#pragma unroll
    for (int q = 0; q < QA; q++) pa[q] = A + (int64_t)min(tr + 64 * q, rowsA - 1) * lda + tq;
#pragma unroll
    for (int q = 0; q < QB; q++) pb[q] = B + (int64_t)min(tr + 64 * q, rowsB - 1) * ldb + tq;
    float4 ra[QA], rb[QB];
    auto fetch = [&](const int k0) {
        const bool in = k0 + tq < kdim;
#pragma unroll
        for (int q = 0; q < QA; q++) ra[q] = in ? *reinterpret_cast<const float4 *>(pa[q] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < QB; q++) rb[q] = in ? *reinterpret_cast<const float4 *>(pb[q] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    fetch(0);
    for (int k0 = 0; k0 < kdim; k0 += GM32_KC) {
#pragma unroll
        for (int q = 0; q < QA; q++) *reinterpret_cast<float4 *>(&sm.a[tr + 64 * q][tq]) = ra[q];
#pragma unroll
        for (int q = 0; q < QB; q++) *reinterpret_cast<float4 *>(&sm.b[tr + 64 * q][tq]) = rb[q];
        __syncthreads();
        if (k0 + GM32_KC < kdim) fetch(k0 + GM32_KC);
#pragma unroll
        for (int kk = 0; kk < GM32_KC; kk += 16) {
            float4 a[WM], b[WN];
#pragma unroll
            for (int t = 0; t < WM; t++) a[t] = *reinterpret_cast<const float4 *>(&sm.a[wi + 16 * t + lr][kk + 4 * lk]);
#pragma unroll
            for (int t = 0; t < WN; t++) b[t] = *reinterpret_cast<const float4 *>(&sm.b[wj + 16 * t + lr][kk + 4 * lk]);
#pragma unroll
            for (int s_ = 0; s_ < 4; s_++) {
#pragma unroll
                for (int ta = 0; ta < WM; ta++)
#pragma unroll
                    for (int tb = 0; tb < WN; tb++) {
                        const float av = s_ == 0 ? a[ta].x : s_ == 1 ? a[ta].y : s_ == 2 ? a[ta].z : a[ta].w;
                        const float bv = s_ == 0 ? b[tb].x : s_ == 1 ? b[tb].y : s_ == 2 ? b[tb].z : b[tb].w;
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[ta][tb], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int ta = 0; ta < WM; ta++)
#pragma unroll
        for (int tb = 0; tb < WN; tb++)
#pragma unroll
            for (int r = 0; r < 4; r++) store(wi + 16 * ta + 4 * lk + r, wj + 16 * tb + lr, acc[ta][tb][r]);
}

// (Built and measured, not kept: the DMA form of gemm_f64.h for this tile -- two images of 33 KB, groups of eight rows 1040 bytes apart, one
//  barrier per chunk, bit-identical: 9.16 ms against 8.97 for the pointer form above on the 20 736-d CSM of 28 pairs.)

}  // namespace acoss
