// gemm_f64.h -- one 128 x 64 output tile of C = A . B^T (both operands row-major along the contraction axis) on
// v_mfma_f64_16x16x4_f64, shared by the wide-feature CSM, the FTM2D all-pairs similarity and the cross-diffusion
// products of the similarity network fusion.
//
// 512 threads = 8 waves, wave tile 32 x 32 (2 x 2 MFMA tiles, 32 accumulator VGPRs); the contraction axis is walked in chunks of
// 32 staged through LDS (row stride 36 doubles: the 64-lane operand reads take their minimum two passes), with the next chunk's
// values fetched into registers while the current one is multiplied.  116 VGPRs and 54 KB of LDS: TWO blocks per CU, and that is
// the point (round 5): the waves of a block run in step (two barriers per chunk) and write LDS together with the matrix pipe idle;
// a second block on the CU multiplies meanwhile.  Measured per product of 32 pairs of 1984^3 (rocprofv3 kernel trace; 6.83 ms =
// the f64 matrix peak): 128 x 128 tiles, one block of 8 waves (32 x 64 wave tiles, 180 VGPRs) per CU 9.1-9.2 ms; the same tile as
// 16 waves of 32 x 32 9.4; this form 8.75 (0.78 of the peak) although it stages 1.5 x the bytes; 64 x 64 tiles, three blocks of four
// waves 9.96 (12 flop per byte staged is not enough).  tools/ubench/mfma_f64_peak.hip: the instruction alone sustains 0.92-0.98 of
// the peak, with its operands read from LDS as here 0.95.
// Accumulation is in contraction order (k ascending, one FMA per k inside the MFMA), i.e. the same chain as a
// scalar loop -- kernels built on this agree bit for bit with their scalar counterparts.
#pragma once

#include <hip/hip_runtime.h>

namespace acoss {

typedef double v4f64_g __attribute__((ext_vector_type(4)));
#ifndef GM_THREADS_V
#define GM_THREADS_V 512
#endif
#ifndef GM_KK_UNROLL
#define GM_KK_UNROLL 8
#endif
#ifndef GM_TJ_V
#define GM_TJ_V 64
#endif
#ifndef GM_T_V
#define GM_T_V 128
#endif
constexpr int GM_T = GM_T_V, GM_TJ = GM_TJ_V, GM_KC = 32, GM_LD = GM_KC + 4, GM_THREADS = GM_THREADS_V;    // tile: GM_T rows x GM_TJ columns
// wave tile 32 x GM_WJ (GM_T / 32 rows of waves x GM_WCOLS columns)
constexpr int GM_WJ = GM_T * GM_TJ / (32 * (GM_THREADS / 64)), GM_TB = GM_WJ / 16, GM_WCOLS = GM_TJ / GM_WJ;
constexpr int GM_FQ = GM_T * GM_KC / GM_THREADS, GM_FQB = GM_TJ * GM_KC / GM_THREADS;           // values per thread and chunk: A, B
static_assert(GM_WJ % 16 == 0 && GM_TJ % GM_WJ == 0 && (GM_T / 32) * GM_WCOLS == GM_THREADS / 64, "block geometry");

struct GemmSmem {
    double a[GM_T][GM_LD];
    double b[GM_TJ][GM_LD];
};

// loadA(r, k) / loadB(r, k): element k of row r (0..127) of the tile's operand, 0.0 outside the matrix;
// store(i, j, v): the finished element (i, j) of the tile (0..127 each); called for every element.
template <typename LoadA, typename LoadB, typename Store>
__device__ inline void gemm_nt_tile_f64(GemmSmem &sm, int kdim, LoadA loadA, LoadB loadB, Store store)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int wi = (wave / GM_WCOLS) * 32, wj = (wave % GM_WCOLS) * GM_WJ;
    v4f64_g acc[2][GM_TB];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < GM_TB; b++) acc[a][b] = (v4f64_g){0.0, 0.0, 0.0, 0.0};
    double ra[GM_FQ], rb[GM_FQB];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int q = 0; q < GM_FQ; q++) {
            const int e = threadIdx.x + GM_THREADS * q;
            ra[q] = loadA(e >> 5, k0 + (e & 31));
            if (q < GM_FQB) rb[q] = loadB(e >> 5, k0 + (e & 31));
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < kdim; k0 += GM_KC) {
#pragma unroll
        for (int q = 0; q < GM_FQ; q++) {
            const int e = threadIdx.x + GM_THREADS * q;
            sm.a[e >> 5][e & 31] = ra[q];
            if (q < GM_FQB) sm.b[e >> 5][e & 31] = rb[q];
        }
        __syncthreads();
        if (k0 + GM_KC < kdim) fetch(k0 + GM_KC);
#pragma unroll GM_KK_UNROLL
        for (int kk = 0; kk < GM_KC; kk += 4) {
            double a[2], b[GM_TB];
#pragma unroll
            for (int t = 0; t < 2; t++) a[t] = sm.a[wi + 16 * t + lr][kk + lk];
#pragma unroll
            for (int t = 0; t < GM_TB; t++) b[t] = sm.b[wj + 16 * t + lr][kk + lk];
#pragma unroll
            for (int ta = 0; ta < 2; ta++)
#pragma unroll
                for (int tb = 0; tb < GM_TB; tb++)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int ta = 0; ta < 2; ta++)
#pragma unroll
        for (int tb = 0; tb < GM_TB; tb++)
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
    // (tiles at the matrix edge too: rows past the last one repeat it -- element (i, j) depends on row i of A and row j of B only,
    //  and `store` drops what lies outside; until round 5 those tiles took the general form, at three times an interior tile's time)
    const bool interior = rowsA >= 1 && rowsB >= 1 && (kdim & 1) == 0 && (lda & 1) == 0 && (ldb & 1) == 0 &&
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
    const int wi = (wave / GM_WCOLS) * 32, wj = (wave % GM_WCOLS) * GM_WJ;
    v4f64_g acc[2][GM_TB];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < GM_TB; b++) acc[a][b] = (v4f64_g){0.0, 0.0, 0.0, 0.0};
    // thread t: rows (t >> 4) + RS q (q < RQ), doubles 2 (t & 15), + 1 of the chunk
    constexpr int RS = GM_THREADS / 16, RQ = GM_T / RS, RQB = GM_TJ / RS;
    const int tr = threadIdx.x >> 4, tk = 2 * (threadIdx.x & 15);
    const double *pa[RQ], *pb[RQB];
#pragma unroll
    for (int q = 0; q < RQ; q++) pa[q] = A + (int64_t)min(tr + RS * q, rowsA - 1) * lda + tk;
#pragma unroll
    for (int q = 0; q < RQB; q++) pb[q] = B + (int64_t)min(tr + RS * q, rowsB - 1) * ldb + tk;
    const bool live = wi < rowsA && wj < rowsB;           // (wave-uniform) a wave tile wholly outside the matrix multiplies nothing
    v2f64_g ra[RQ], rb[RQB];
    auto fetch = [&](const int k0) {
        const bool in = k0 + tk < kdim;                   // (the last chunk may be ragged: kdim is even, so pairs are whole)
#pragma unroll
        for (int q = 0; q < RQ; q++) ra[q] = in ? *reinterpret_cast<const v2f64_g *>(pa[q] + k0) : (v2f64_g){0.0, 0.0};
#pragma unroll
        for (int q = 0; q < RQB; q++) rb[q] = in ? *reinterpret_cast<const v2f64_g *>(pb[q] + k0) : (v2f64_g){0.0, 0.0};
    };
    // Timing-only builds (-DGM_PROBE=1 no global loads inside the walk, 2 no staging and no barriers, 3 no products) of the 128 x 128
    // form, 32 products of 1984^3 per launch under rocprofv3 (round 5): whole kernel 9.2-9.4 ms, 1: 8.7, 2: 7.1-7.5 (0.91-0.97 of the
    // matrix peak: the multiply loop itself is fine), 3: 3.1 -- a third of the load + staging time was not hidden.  Built on that form
    // and measured, none faster: two LDS buffers and one barrier per chunk (9.4 ms); operands one pair of steps ahead in a second
    // register set (9.9; also slower on this form: 9.7 against 8.75); the staging writes between the products of a chunk's third
    // pair of steps (11.1); 128 VGPRs for two blocks per CU (spills: 215-240 pairs/s against 266).  What helped is a second block.
    fetch(0);
    for (int k0 = 0; k0 < kdim; k0 += GM_KC) {
#if !defined(GM_PROBE) || GM_PROBE != 2
#pragma unroll
        for (int q = 0; q < RQ; q++) *reinterpret_cast<v2f64_g *>(&sm.a[tr + RS * q][tk]) = ra[q];
#pragma unroll
        for (int q = 0; q < RQB; q++) *reinterpret_cast<v2f64_g *>(&sm.b[tr + RS * q][tk]) = rb[q];
        __syncthreads();
#endif
#if !defined(GM_PROBE) || GM_PROBE != 1
        if (k0 + GM_KC < kdim) fetch(k0 + GM_KC);
#endif
#if !defined(GM_PROBE) || GM_PROBE != 3
        if (live) {
#pragma unroll
            for (int kk = 0; kk < GM_KC; kk += 4) {
                double a[2], b[GM_TB];
#pragma unroll
                for (int t = 0; t < 2; t++) a[t] = sm.a[wi + 16 * t + lr][kk + lk];
#pragma unroll
                for (int t = 0; t < GM_TB; t++) b[t] = sm.b[wj + 16 * t + lr][kk + lk];
#pragma unroll
                for (int ta = 0; ta < 2; ta++)
#pragma unroll
                    for (int tb = 0; tb < GM_TB; tb++)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
            }
        }
#endif
#if !defined(GM_PROBE) || GM_PROBE != 2
        __syncthreads();
#endif
    }
#pragma unroll
    for (int ta = 0; ta < 2; ta++)
#pragma unroll
        for (int tb = 0; tb < GM_TB; tb++)
#pragma unroll
            for (int r = 0; r < 4; r++) store(wi + 16 * ta + lk + 4 * r, wj + 16 * tb + lr, acc[ta][tb][r]);
}

}  // namespace acoss

// ---- the same product with its operands brought into LDS by the DMA path (global_load_lds_dwordx4: no staging registers, no ds_write)
// into TWO images, one barrier per chunk (round 5; snf_kernels.hip and ftm2d_kernels.hip use it where the operands' leading
// dimensions are even: -DGM_DMA=0 switches it off).  Per product of 32 pairs of 1984^3: 8.19 ms against 8.75 for the register-staged 128 x 64
// form above (0.83 of the f64 matrix peak; operands a pair of steps ahead in a second register set: no change).  128 x 128 tile, 1024 threads = sixteen
// waves of 32 x 32, one block per CU (133 KB of LDS).  A wave instruction of the DMA path writes 1 KB = four rows x 32 doubles
// contiguously; the image keeps those groups 1040 bytes apart and puts rows r, r + 32, r + 64, r + 96 into one group, so the sixteen
// rows of an operand read (r .. r + 15: sixteen different groups) meet in different banks with plain immediate offsets for the steps.
namespace acoss {

constexpr int GD_T = 128, GD_KC = 32, GD_THREADS = 1024, GD_GROUP = 1024 + 16, GD_OPB = 32 * GD_GROUP;

struct GemmDmaSmem {
    __attribute__((aligned(16))) char img[2][2][GD_OPB];          // [buffer][A / B][32 groups]
};

static __device__ __attribute__((aligned(16))) double gd_zero[2] = {0.0, 0.0};       // what lanes behind the end of the contraction axis load

// operands as gemm_nt_tile_f64_rows's; needs lda, ldb, kdim even and 16-byte aligned bases (the caller checks)
template <typename Store>
__device__ inline void gemm_nt_tile_f64_dma(GemmDmaSmem &sm, int kdim, const double *__restrict__ A, int lda, int rowsA,
                                            const double *__restrict__ B, int ldb, int rowsB, Store store)
{
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 2) * 32, wj = (wave & 3) * 32;
    v4f64_g acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (v4f64_g){0.0, 0.0, 0.0, 0.0};
    // wave w brings groups w and w + 16 of either operand: lane (s = lane >> 4, piece = lane & 15) -> row group + 32 s, doubles 2 piece, + 1
    const int piece = lane & 15, slot = lane >> 4;
    const double *srcA[2], *srcB[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int row = wave + 16 * q + 32 * slot;
        srcA[q] = A + (int64_t)min(row, rowsA - 1) * lda + 2 * piece;
        srcB[q] = B + (int64_t)min(row, rowsB - 1) * ldb + 2 * piece;
    }
    auto dma = [&](const int buf, const int k0) {
        const bool in = k0 + 2 * piece < kdim;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            __builtin_amdgcn_global_load_lds((gptr_t)(in ? srcA[q] + k0 : gd_zero), (lptr_t)(sm.img[buf][0] + (wave + 16 * q) * GD_GROUP), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(in ? srcB[q] + k0 : gd_zero), (lptr_t)(sm.img[buf][1] + (wave + 16 * q) * GD_GROUP), 16, 0, 0);
        }
    };
    const bool live = wi < rowsA && wj < rowsB;
    const int offA = lr * GD_GROUP + (wave >> 2) * 256 + lk * 8, offB = lr * GD_GROUP + (wave & 3) * 256 + lk * 8;
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < kdim; k0 += GD_KC) {
        if (k0 + GD_KC < kdim) dma(cur ^ 1, k0 + GD_KC);
        if (live) {
            const char *ia = sm.img[cur][0] + offA, *ib = sm.img[cur][1] + offB;
#pragma unroll
            for (int kk = 0; kk < GD_KC; kk += 4) {
                double a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    a[t] = *reinterpret_cast<const double *>(ia + 16 * t * GD_GROUP + kk * 8);
                    b[t] = *reinterpret_cast<const double *>(ib + 16 * t * GD_GROUP + kk * 8);
                }
#pragma unroll
                for (int ta = 0; ta < 2; ta++)
#pragma unroll
                    for (int tb = 0; tb < 2; tb++)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int ta = 0; ta < 2; ta++)
#pragma unroll
        for (int tb = 0; tb < 2; tb++)
#pragma unroll
            for (int r = 0; r < 4; r++) store(wi + 16 * ta + lk + 4 * r, wj + 16 * tb + lr, acc[ta][tb][r]);
}

}  // namespace acoss
