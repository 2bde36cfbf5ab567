// crp_kernels.hip -- batched stage kernels for the cross-recurrence-plot half of the hot path:
//   frame norms, OTI (CRPUtils.py:109), cross-similarity (CRPUtils.py:67), sliding window
//   (CRPUtils.py:24), kNN thresholds + binarisation (CRPUtils.py:169/201).
// gfx950 (CDNA4) only: wave64, DPP reductions, scalar (SGPR) broadcast of wave-uniform rows.
#include "common.h"
#include "wave_ops.h"
#include "gemm_f32.h"
#include "gemm_f64.h"
#include "kernel_utils.h"
#include "thresh_work.h"

#include <math.h>

namespace acoss {

// ---------------------------------------------------------------------------------------------
// per-frame squared norms (FMA chain in bin order; this order is part of the kernels' contract:
// every kernel that forms |x|^2+|y|^2-2x.y reads these and therefore agrees bit for bit)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void frame_norms_kernel(const T *__restrict__ feats, int64_t n_frames,
                                                          int d, T *__restrict__ norms)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_frames) return;
    const T *x = feats + f * d;
    T acc = 0;
    for (int b = 0; b < d; b++) acc = fma(x[b], x[b], acc);
    norms[f] = acc;
}

// ---------------------------------------------------------------------------------------------
// OTI: one thread per pair.  The 12 products are rounded, then summed in numpy's pairwise order
// for a contiguous run (eight accumulators combined as a tree, tail sequential), so the scores
// -- and the first-maximum argmax -- are bit-identical to np.sum(np.roll(C1, s) * C2).
// ---------------------------------------------------------------------------------------------
__device__ inline double np_sum_small(const double *a, int n)   // n <= 128
{
    if (n < 8) {
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc = __dadd_rn(acc, a[i]);
        return acc;
    }
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] = __dadd_rn(r[k], a[i + k]);
    double acc = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                           __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    for (; i < n; i++) acc = __dadd_rn(acc, a[i]);
    return acc;
}

__global__ __launch_bounds__(64) void oti_kernel(const double *__restrict__ gchroma, int nbins,
                                                 acoss_pair_desc *descs, int K)
{
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= K) return;
    const double *c1 = gchroma + (int64_t)descs[p].song_x * nbins;
    const double *c2 = gchroma + (int64_t)descs[p].song_y * nbins;
    double prod[64];
    int best = 0;
    double best_score = 0.0;
    for (int s = 0; s < nbins; s++) {
        for (int b = 0; b < nbins; b++) {
            int src = b - s;
            if (src < 0) src += nbins;
            prod[b] = __dmul_rn(c1[src], c2[b]);
        }
        const double score = np_sum_small(prod, nbins);
        if (s == 0 || score > best_score) {
            best_score = score;
            best = s;
        }
    }
    descs[p].shift = best;
}

// ---------------------------------------------------------------------------------------------
// Cross-similarity matrix, materialising (the HBM-roofline-graded kernel).
//
// Block = 4 waves, tile = 128 rows x 128 columns.  A lane owns 2 adjacent columns and keeps the
// two y frames (2 x D values) in VGPRs for the whole tile; a wave walks 32 rows, and the x frame
// of a row is wave-uniform, so it is fetched with scalar loads and fed to v_fma as an SGPR
// operand -- no LDS, no per-lane x traffic.  Every store instruction of a wave writes one
// contiguous 128 x sizeof(T) run of an output row (16 B per lane for float64).
// Algorithmic traffic: sizeof(T) * (nx*ny + d*(nx+ny)) bytes per pair; the feature re-reads
// across tiles are served by L2 (tiles of one pair are placed on one XCD).
// ---------------------------------------------------------------------------------------------
constexpr int CSM_TM = 128, CSM_TN = 128, CSM_ROWS_PER_WAVE = 32;

template <typename T>
__device__ inline T clamp_sqrt(T c)
{
    c = c < (T)0 ? (T)0 : c;   // CRPUtils.py:83
    return sqrt(c);
}

// MODE (development probes, product = 0): 1 = stores only (no arithmetic), 2 = arithmetic only (store
// suppressed), 3 = assume shift == 0 (contiguous scalar loads of the x frame)
template <typename T, int D, int MODE = 0>
__global__ __launch_bounds__(256) void csm_kernel(const T *__restrict__ feats, const T *__restrict__ norms,
                                                  const acoss_pair_desc *__restrict__ descs,
                                                  int tiles_m, int tiles_n, T *__restrict__ out)
{
    const int tiles = tiles_m * tiles_n;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / tiles, t = lb % tiles;
    const acoss_pair_desc ds = descs[p];
    const int i0 = (t / tiles_n) * CSM_TM, j0 = (t % tiles_n) * CSM_TN;
    if (i0 >= ds.nx || j0 >= ds.ny) return;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int shift = MODE == 3 ? 0 : ds.shift;

    // this lane's two y frames
    const int j = j0 + 2 * lane;
    const bool ok0 = j < ds.ny, ok1 = j + 1 < ds.ny;
    T y0[D], y1[D];
    {
        const T *yp0 = feats + (ds.y_row0 + (ok0 ? j : 0)) * D;
        const T *yp1 = feats + (ds.y_row0 + (ok1 ? j + 1 : 0)) * D;
#pragma unroll
        for (int b = 0; b < D; b++) {
            y0[b] = yp0[b];
            y1[b] = yp1[b];
        }
    }
    const T yy0 = norms[ds.y_row0 + (ok0 ? j : 0)];
    const T yy1 = norms[ds.y_row0 + (ok1 ? j + 1 : 0)];

    // rolled x: X1[f][b] = X[f][(b - shift) mod D]   (np.roll(chroma, oti, axis=0), Serra09.py:167)
    int rot[D];
#pragma unroll
    for (int b = 0; b < D; b++) {
        int src = b - shift;
        rot[b] = src < 0 ? src + D : src;
    }

    T *orow = out + ds.csm_off + (int64_t)j;
    const int r_begin = i0 + wave * CSM_ROWS_PER_WAVE;
    const int r_end = min(r_begin + CSM_ROWS_PER_WAVE, ds.nx);
    // wave-uniform: whole 128-column strip inside the matrix and every row 16-byte aligned
    const bool full = (j0 + CSM_TN <= ds.ny) && ((ds.csm_pitch & 1) == 0) && ((ds.csm_off & 1) == 0);
    if (full) {
        for (int i = r_begin; i < r_end; i++) {
            const T *xp = feats + (ds.x_row0 + i) * D;   // wave-uniform address -> scalar loads
            T c0, c1;
            if constexpr (MODE == 1) {
                c0 = (T)i; c1 = yy0;
            } else {
                const T xx = norms[ds.x_row0 + i];
                T a0 = 0, a1 = 0;
#pragma unroll
                for (int b = 0; b < D; b++) {
                    const T xb = xp[rot[b]];
                    a0 = fma(xb, y0[b], a0);
                    a1 = fma(xb, y1[b], a1);
                }
                c0 = csm_sqrt(fma((T)-2, a0, xx + yy0));
                c1 = csm_sqrt(fma((T)-2, a1, xx + yy1));
            }
            T *dst = orow + (int64_t)i * ds.csm_pitch;
            if constexpr (MODE == 2) {
                if (c0 == (T)-1.25) dst[0] = c1;      // never true: keeps the arithmetic alive
            } else
            if constexpr (sizeof(T) == 8) {
                *reinterpret_cast<double2 *>(dst) = make_double2(c0, c1);   // one 16-byte store per lane
            } else {
                *reinterpret_cast<float2 *>(dst) = make_float2(c0, c1);
            }
        }
    } else {
        for (int i = r_begin; i < r_end; i++) {
            const T *xp = feats + (ds.x_row0 + i) * D;
            const T xx = norms[ds.x_row0 + i];
            T a0 = 0, a1 = 0;
#pragma unroll
            for (int b = 0; b < D; b++) {
                const T xb = xp[rot[b]];
                a0 = fma(xb, y0[b], a0);
                a1 = fma(xb, y1[b], a1);
            }
            const T c0 = csm_sqrt(fma((T)-2, a0, xx + yy0));
            const T c1 = csm_sqrt(fma((T)-2, a1, xx + yy1));
            T *dst = orow + (int64_t)i * ds.csm_pitch;
            if (ok0) dst[0] = c0;
            if (ok1) dst[1] = c1;
        }
    }
}

// Any feature dimension (d up to a few thousand): one thread per output element, plain loops.
// Correct-first path for shapes other than the chroma / MFCC ones.
template <typename T>
__global__ __launch_bounds__(256) void csm_generic_kernel(const T *__restrict__ feats, const T *__restrict__ norms,
                                                          int d, const acoss_pair_desc *__restrict__ descs,
                                                          int tiles_m, int tiles_n, T *__restrict__ out)
{
    const int tiles = tiles_m * tiles_n;
    const int p = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const acoss_pair_desc ds = descs[p];
    const int i = (t / tiles_n) * 16 + (threadIdx.x >> 4), j = (t % tiles_n) * 16 + (threadIdx.x & 15);
    if (i >= ds.nx || j >= ds.ny) return;
    const T *x = feats + (ds.x_row0 + i) * d, *y = feats + (ds.y_row0 + j) * d;
    T acc = 0;
    for (int b = 0; b < d; b++) {
        int src = b - ds.shift;
        if (src < 0) src += d;
        acc = fma(x[src], y[b], acc);
    }
    const T c = fma((T)-2, acc, norms[ds.x_row0 + i] + norms[ds.y_row0 + j]);
    out[ds.csm_off + (int64_t)i * ds.csm_pitch + j] = clamp_sqrt(c);
}

// float64 features of any width on the matrix cores (the 20 736-dimensional scattering features of Serra09.py:187,
// the 'ssms' blocks of EarlySNF.py:72-74): one 128 x 128 tile per block (gemm_f64.h), accumulating in feature
// order -- the same FMA chain as csm_generic_kernel, so the two agree bit for bit.
__global__ __launch_bounds__(GM_THREADS) void csm_gemm_kernel(const double *__restrict__ feats, const double *__restrict__ norms,
                                                              int d, const acoss_pair_desc *__restrict__ descs,
                                                              int tiles_m, int tiles_n, double *__restrict__ out)
{
    __shared__ GemmSmem sm;
    const int tiles = tiles_m * tiles_n;
    const int p = blockIdx.x / tiles, t = blockIdx.x % tiles;
    const acoss_pair_desc ds = descs[p];
    const int i0 = (t / tiles_n) * GM_T, j0 = (t % tiles_n) * GM_TJ;
    if (i0 >= ds.nx || j0 >= ds.ny) return;
    auto store = [&](const int i, const int j, const double v) {
        if (i0 + i < ds.nx && j0 + j < ds.ny) {
            const double c = fma(-2.0, v, norms[ds.x_row0 + i0 + i] + norms[ds.y_row0 + j0 + j]);
            out[ds.csm_off + (int64_t)(i0 + i) * ds.csm_pitch + j0 + j] = clamp_sqrt(c);
        }
    };
    if (ds.shift == 0)      // (no rotation: rows as they lie -- the form with one pointer per staged pair; it checks alignment itself)
        gemm_nt_tile_f64_rows(sm, d, feats + (ds.x_row0 + i0) * (int64_t)d, d, ds.nx - i0, feats + (ds.y_row0 + j0) * (int64_t)d, d, ds.ny - j0, store);
    else
        gemm_nt_tile_f64(
            sm, d,
            [&](const int r, const int k) {
                int src = k - ds.shift;                    // np.roll(chroma_i, oti) (Serra09.py:167)
                if (src < 0) src += d;
                return (i0 + r < ds.nx && k < d) ? feats[(ds.x_row0 + i0 + r) * d + src] : 0.0;
            },
            [&](const int r, const int k) { return (j0 + r < ds.ny && k < d) ? feats[(ds.y_row0 + j0 + r) * d + k] : 0.0; }, store);
}

// The same for float32 features (the reference keeps the scattering features in float32, Serra09.py:187-192, and
// get_csm follows the dtype of its inputs, CRPUtils.py:82): gemm_f32.h on v_mfma_f32_16x16x4_f32, float32 output.
// Rows are read as 16-byte quads when the layout allows it (no roll, d a multiple of 4: the scattering case).
// |result^2 - exact^2| <= (d + 4) 2^-24 (|x|^2 + |y|^2), the bound of any float32 dot product of d terms plus the
// three roundings of the norms' sum and the final FMA; tests/test_gpu_stages.py compares within twice that (both
// sides of the comparison carry it).
template <int WM>
__global__ __launch_bounds__(GM32_THREADS, 2) void csm_gemm32_kernel(const float *__restrict__ feats, const float *__restrict__ norms,
                                                                     int d, const acoss_pair_desc *__restrict__ descs,
                                                                     int tiles_m, int tiles_n, float *__restrict__ out)
{
    constexpr int TM = 64 * WM, TN = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char gemm32_smem[];
    Gemm32Smem<WM, 4> &sm = *reinterpret_cast<Gemm32Smem<WM, 4> *>(gemm32_smem);
    const int tiles = tiles_m * tiles_n;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);      // neighbouring tiles (same x rows) share an L2
    const int p = lb / tiles, t = lb % tiles;
    const acoss_pair_desc ds = descs[p];
    const int i0 = (t / tiles_n) * TM, j0 = (t % tiles_n) * TN;
    if (i0 >= ds.nx || j0 >= ds.ny) return;
    const bool quads = ds.shift == 0 && (d & 3) == 0 && (reinterpret_cast<uintptr_t>(feats) & 15) == 0;      // block-uniform
    auto row_quad = [&](const int64_t row0, const int rows, const int r, const int k, const int shift) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= rows || k >= d) return v;
        const float *src = feats + (row0 + r) * d;
        if (quads) return *reinterpret_cast<const float4 *>(src + k);
        float e[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int q = k + c - shift;                     // np.roll(chroma_i, oti) (Serra09.py:167)
            if (q < 0) q += d;
            e[c] = k + c < d ? src[q] : 0.f;
        }
        return make_float4(e[0], e[1], e[2], e[3]);
    };
    auto store = [&](const int i, const int j, const float v) {
        if (i0 + i < ds.nx && j0 + j < ds.ny) {
            const float c = fmaf(-2.0f, v, norms[ds.x_row0 + i0 + i] + norms[ds.y_row0 + j0 + j]);
            out[ds.csm_off + (int64_t)(i0 + i) * ds.csm_pitch + j0 + j] = clamp_sqrt(c);
        }
    };
    if (quads)          // (no rotation, whole quads, aligned rows: one pointer per staged quad -- the scattering features' case)
        gemm_nt_tile_f32_rows(sm, d, feats + (ds.x_row0 + i0) * (int64_t)d, d, ds.nx - i0, feats + (ds.y_row0 + j0) * (int64_t)d, d, ds.ny - j0, store);
    else
        gemm_nt_tile_f32(
            sm, d,
            [&](const int r, const int k) { return row_quad(ds.x_row0 + i0, ds.nx - i0, r, k, ds.shift); },
            [&](const int r, const int k) { return row_quad(ds.y_row0 + j0, ds.ny - j0, r, k, 0); }, store);
}

// ---------------------------------------------------------------------------------------------
// Sliding-window (delay embedding) of a materialised CSM: S[i][j] = sqrt(sum_k csm[i+k][j+k]^2).
// Tile of 32 x 64 outputs; the (32+win-1) x (64+win-1) squared inputs are staged in LDS as
// float64 (float32 inputs are squared in float32 first, CRPUtils.py:40-41), each thread then
// sums its diagonals directly (k ascending).  LDS reads are lane-contiguous: conflict-free.
// ---------------------------------------------------------------------------------------------
constexpr int SL_TM = 32, SL_TN = 64;

template <typename T>
__global__ __launch_bounds__(256) void sliding_kernel(const T *__restrict__ csm,
                                                      const acoss_pair_desc *__restrict__ descs, int win,
                                                      int tiles_m, int tiles_n, double *__restrict__ S)
{
    extern __shared__ double sq[];
    const int tiles = tiles_m * tiles_n;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / tiles, t = lb % tiles;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int i0 = (t / tiles_n) * SL_TM, j0 = (t % tiles_n) * SL_TN;
    if (i0 >= M || j0 >= N) return;
    const int in_rows = SL_TM + win - 1, in_cols = SL_TN + win - 1;
    const int ld = in_cols | 1;   // odd pitch
    const T *src = csm + ds.csm_off;
    for (int idx = threadIdx.x; idx < in_rows * in_cols; idx += 256) {
        const int r = idx / in_cols, c = idx % in_cols;
        const int gi = i0 + r, gj = j0 + c;
        double v = 0.0;
        if (gi < ds.nx && gj < ds.ny) {
            const T x = src[(int64_t)gi * ds.csm_pitch + gj];
            const T x2 = x * x;
            v = (double)x2;
        }
        sq[r * ld + c] = v;
    }
    __syncthreads();
    const int c = threadIdx.x & 63;
    const int gj = j0 + c;
    if (gj >= N) return;
    for (int r = threadIdx.x >> 6; r < SL_TM; r += 4) {
        const int gi = i0 + r;
        if (gi >= M) break;
        double acc = 0.0;
        for (int k = 0; k < win; k++) acc += sq[(r + k) * ld + (c + k)];
        S[ds.crp_off + (int64_t)gi * ds.crp_pitch + gj] = sqrt(acc);
    }
}

// ---------------------------------------------------------------------------------------------
// kNN thresholds.  One wave per row: the row's values sit in registers (EPL per lane), and
// wave_select_kth() finds the k-th smallest.  The column pass stages 8 adjacent columns of the
// matrix through LDS (64-byte coalesced reads) and then runs the same in-register selection
// with one wave per column.
// thr[] holds the order-preserving key of the threshold value, cut[] the tie cut index.
// ---------------------------------------------------------------------------------------------

__device__ inline void store_uniform_select(int k, int n, uint64_t *thr, int *cut, const SelectResult &r)
{
    if ((threadIdx.x & 63) == 0) {
        *thr = r.thr_key;
        *cut = r.cut;
    }
}

// k <= 0: nothing selected; k >= n: everything.  Encoded so that the mask rule needs no branch.
__device__ inline bool trivial_select(int k, int n, SelectResult &r)
{
    if (k <= 0) { r.thr_key = 0ull; r.cut = -1; return true; }
    if (k >= n) { r.thr_key = ~0ull; r.cut = 0x7fffffff; return true; }
    return false;
}

// Rows: each wave walks SEL_ROWS_PER_WAVE consecutive rows; while it selects in row r the 16-byte
// loads of row r+1 are already in flight (two register buffers, statically alternated), so the
// HBM latency of a row hides behind the selection of the previous one.  Lane l holds elements
// 128*q + 2*l + {0,1} of the row (one 16-byte load per q).
constexpr int SEL_ROWS_PER_WAVE = 8;

// Branch-free: positions past the row end are clamped to the row's last aligned pair (in bounds,
// the row pitch is even) and masked out when the keys are formed, so all EPL/2 loads of a row issue
// back to back.
template <int EPL>
__device__ inline void load_row_pairs(const double *row, int N, int lane, bool vec_ok, double (&buf)[EPL])
{
    if (vec_ok) {   // wave-uniform
        const int last = ((N + 1) & ~1) - 2;
#pragma unroll
        for (int q = 0; q < EPL / 2; q++) {
            const int j = min(128 * q + 2 * lane, last);
            const double2 v = *reinterpret_cast<const double2 *>(row + j);
            buf[2 * q] = v.x;
            buf[2 * q + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < EPL / 2; q++) {
            const int j = 128 * q + 2 * lane;
            buf[2 * q] = row[min(j, N - 1)];
            buf[2 * q + 1] = row[min(j + 1, N - 1)];
        }
    }
}

// 16-per-lane form: lane l holds positions e*64 + l (8-byte loads, 512 contiguous bytes per instruction), so
// the elements of a lane are 64 positions apart -- neighbouring, strongly correlated values sit in different
// lanes, which the histogram selection wants (one candidate per lane), and the bit words come out as the plain
// bit vector of the row.
__device__ inline void load_row_strided(const double *row, int N, int lane, double (&buf)[16])
{
    // unsigned 32-bit element offsets from the wave-uniform row pointer: SGPR base + one offset VGPR (+ immediates)
    // instead of sixteen 64-bit address pairs
    if (N > 15 * 64) {      // wave-uniform: only the last slot can run past the row
#pragma unroll
        for (int e = 0; e < 15; e++) buf[e] = row[(unsigned)(e * 64 + lane)];
        buf[15] = row[(unsigned)min(15 * 64 + lane, N - 1)];
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) buf[e] = row[(unsigned)min(e * 64 + lane, N - 1)];
    }
}

// The selected positions of one row / column as 16 x uint64: word e = ballot over the lanes of register slot e,
// i.e. bit l of word e = position idx_of(e) of lane l.  For columns (ColIdx: position e*64 + l) that is the
// plain bit vector; for rows (RowIdx: position 128*(e>>1) + 2*l + (e&1)) words 2q and 2q+1 hold the even and
// odd positions of the 128-position group q -- combine_bits_kernel interleaves them.
// Positions follow the mask rule (value < threshold, or value == threshold and position <= cut); the key
// order is the order of the doubles (-0.0 already folded), so values are compared, not keys.
// lane `e` of (lo, hi) := the wave-uniform m.  The s_nop covers gfx950's VALU-writes-SGPR -> VALU-reads-SGPR
// wait states, which the hazard recogniser does not insert around inline asm.
__device__ inline void put_lane_u64(unsigned &lo, unsigned &hi, uint64_t m, int e)
{
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(lo), "+v"(hi)
        : "s"((unsigned)m), "s"((unsigned)(m >> 32)), "n"(e));
}

// lane e (< 16) gets the lane mask of "position idx_of(e) exists": computed once per wave, ANDed into every
// emitted word
template <typename IdxFn>
__device__ inline uint64_t slot_valid_masks(IdxFn idx_of, int n, int lane)
{
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const uint64_t m = __ballot(idx_of(e) < n);
        put_lane_u64(lo, hi, m, e);
    }
    return ((uint64_t)hi << 32) | lo;
}

template <typename IdxFn>
__device__ inline void emit_select_bits(const double (&x)[16], IdxFn idx_of, uint64_t valid, const SelectResult &r,
                                        uint64_t *out, int lane, int64_t stride = 1)
{
    const double tv = r.thr_key == ~0ull ? INFINITY : (r.thr_key == 0ull ? -INFINITY : f64_from_key(r.thr_key));
    unsigned lo = 0, hi = 0;
    if (r.cut == 0x7fffffff) {        // no tie cut (almost always): one compare per slot
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const uint64_t m = __ballot(x[e] <= tv);
            put_lane_u64(lo, hi, m, e);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const uint64_t m = __ballot((x[e] < tv) | ((x[e] == tv) & (idx_of(e) <= r.cut)));
            put_lane_u64(lo, hi, m, e);
        }
    }
    if (lane < 16) out[lane * stride] = (((uint64_t)hi << 32) | lo) & valid;
}

template <int EPL, bool HIST = false>
__device__ inline SelectResult select_from_buf(double (&buf)[EPL], int N, int k, int lane, unsigned &warm,
                                               unsigned *hist, HistWarm &hwarm)
{
    SelectResult res;
    if (trivial_select(k, N, res)) return res;
#pragma unroll
    for (int e = 0; e < EPL; e++) buf[e] = buf[e] + 0.0;     // -0.0 -> +0.0
    if constexpr (EPL == 16) {
        if constexpr (HIST) return wave_select16_hist(buf, ColIdx{lane}, N, k, hist, lane, hwarm);
        else return wave_select16(buf, ColIdx{lane}, N, k, warm);
    } else {
        uint64_t key[EPL];
        int idx[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            idx[e] = 128 * (e >> 1) + 2 * lane + (e & 1);
            key[e] = idx[e] < N ? key_of(buf[e]) : ~0ull;
        }
        return wave_select_kth<EPL>(key, idx, N, k);
    }
}

// MODE (development probes, product = 0): 1 = row loads only, 2 = selection only (synthetic values),
// 3 = the sorted / probing selection instead of the histogram one
template <int EPL, int MODE = 0>
__global__ __launch_bounds__(256, EPL == 16 ? 4 : 1) void select_rows_kernel(const double *__restrict__ S,
                                                          const acoss_pair_desc *__restrict__ descs,
                                                          int win, double kappa_k_fixed, int k_mode,
                                                          ThreshWork w, int rows_blocks)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / rows_blocks;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = ((lb % rows_blocks) * 4 + wave) * SEL_ROWS_PER_WAVE;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (r0 >= M) return;
    const int r1 = min(r0 + SEL_ROWS_PER_WAVE, M);
    const int lane = threadIdx.x & 63;
    // neighbour count from the number of columns (CRPUtils.py:190-193); k_mode 0: fraction
    // (half-even rounding, rint under the default rounding mode), 1: absolute count, 2: all
    const int k = k_mode == 0 ? (int)rint(kappa_k_fixed * (double)N) : (k_mode == 1 ? (int)kappa_k_fixed : N);
    const double *base = S + ds.crp_off;
    const bool vec_ok = ((ds.crp_pitch & 1) == 0) && ((ds.crp_off & 1) == 0);
    uint64_t *thr = w.row_thr + (int64_t)p * w.max_m;
    int *cut = w.row_cut + (int64_t)p * w.max_m;
    double bufA[EPL];
    if constexpr (MODE == 1) {
        for (int i = r0; i < r1; i++) {
            load_row_pairs<EPL>(base + (int64_t)i * ds.crp_pitch, N, lane, vec_ok, bufA);
            double acc = 0;
#pragma unroll
            for (int e = 0; e < EPL; e++) acc += bufA[e];
            if (acc == 1.2345) thr[i] = 0;
        }
        return;
    }
    if constexpr (MODE == 2) {
        for (int i = r0; i < r1; i++) {
#pragma unroll
            for (int e = 0; e < EPL; e++) bufA[e] = (double)((lane * 2654435761u + e * 40503u + i * 97u) & 0xfffff) * 1e-3 + 0.5;
            unsigned cold = 0;
            HistWarm nohw{0, HIST_WARM_SHIFT0};
            SelectResult res = select_from_buf<EPL>(bufA, N, k, lane, cold, nullptr, nohw);
            store_uniform_select(k, N, thr + i, cut + i, res);
        }
        return;
    }
    unsigned warm = 0;      // high word of the previous row's threshold (rows of T change slowly)
    constexpr bool HIST = (EPL == 16) && (MODE == 0);
    __shared__ __attribute__((aligned(16))) unsigned hist_all[HIST ? 4 * HIST_WORDS : 4];
    unsigned *hist = hist_all + (HIST ? wave * HIST_WORDS : 0);
    if constexpr (HIST) hist_clear(hist, lane);
    HistWarm hwarm{0, HIST_WARM_SHIFT0};
    uint64_t slot_valid = 0;
    if constexpr (EPL == 16) {
        if (w.row_bits) slot_valid = slot_valid_masks(ColIdx{lane}, N, lane);
    }
    auto load = [&](double (&buf)[EPL], const int i) {
        if constexpr (EPL == 16) load_row_strided(base + (int64_t)i * ds.crp_pitch, N, lane, buf);
        else load_row_pairs<EPL>(base + (int64_t)i * ds.crp_pitch, N, lane, vec_ok, buf);
    };
    auto process = [&](double (&buf)[EPL], const int i) {
        const SelectResult res = select_from_buf<EPL, HIST>(buf, N, k, lane, warm, hist, hwarm);
        store_uniform_select(k, N, thr + i, cut + i, res);
        if constexpr (EPL == 16) {
            if (w.row_bits && res.cut != SELECT_UNRESOLVED)
                emit_select_bits(buf, ColIdx{lane}, slot_valid, res, w.row_bits + ((int64_t)p * w.max_m + i) * 16, lane);
        }
    };
    // One row buffer: a second, prefetched buffer costs more registers (occupancy / spills) than it gains,
    // with either selection (measured).  The kernel runs at ~75 % of its load-only time.
    for (int i = r0; i < r1; i++) {
        load(bufA, i);
        process(bufA, i);
    }
}

constexpr int SEL_COLS_PER_BLOCK = 8;       // = waves per block; 64-byte row segments (4 columns / block: loads alone 4.4 ms)
constexpr int SEL_COLS_THREADS = SEL_COLS_PER_BLOCK * 64;

template <int EPL, int MODE = 0>
__global__ __launch_bounds__(SEL_COLS_THREADS, 4) void select_cols_kernel(const double *__restrict__ S,
                                                          const acoss_pair_desc *__restrict__ descs,
                                                          int win, double kappa_k_fixed, int k_mode,
                                                          ThreshWork w, int col_blocks)
{
    extern __shared__ double colbuf[];   // [8][ldc]
    // neighbouring column blocks read the two 64-byte halves of the same lines: keep them on one XCD
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / col_blocks;
    const int j0 = (lb % col_blocks) * SEL_COLS_PER_BLOCK;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (j0 >= N) return;
    constexpr int ldc = EPL * 64 + 2;
    const double *base = S + ds.crp_off;
    // 512 threads: 64 rows x 8 columns per sweep, 64-byte segments per row.  Fully unrolled and
    // branch-free (clamped addresses) so that all EPL loads of a thread are in flight together.
    {
        const int c = threadIdx.x % SEL_COLS_PER_BLOCK, rr = threadIdx.x / SEL_COLS_PER_BLOCK;
        const int cc = min(j0 + c, N - 1);
        double tmp[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) tmp[e] = base[(int64_t)min(e * 64 + rr, M - 1) * ds.crp_pitch + cc];
#pragma unroll
        for (int e = 0; e < EPL; e++) colbuf[c * ldc + e * 64 + rr] = tmp[e];
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = j0 + wave;
    if (j >= N) return;
    const int lane = threadIdx.x & 63;
    if constexpr (MODE == 1) {       // probe: loads only
        if (colbuf[wave * ldc + lane] == -1.2345) w.col_cut[0] = 0;
        return;
    }
    int k = k_mode == 0 ? (int)rint(kappa_k_fixed * (double)M) : (k_mode == 1 ? (int)kappa_k_fixed : M);
    SelectResult res;
    if (trivial_select(k, M, res)) {
        if constexpr (EPL == 16) {
            if (w.col_bits) {
                double x[16];
#pragma unroll
                for (int e = 0; e < 16; e++) x[e] = 0.0;
                emit_select_bits(x, ColIdx{lane}, slot_valid_masks(ColIdx{lane}, M, lane), res, w.col_word(p, j, 0), lane, w.max_n);
            }
        }
    } else {
        if constexpr (EPL == 16) {
            double x[16];
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = colbuf[wave * ldc + min(e * 64 + lane, M - 1)] + 0.0;
            if constexpr (MODE == 3) {
                unsigned cold = 0;
                res = wave_select16(x, ColIdx{lane}, M, k, cold);
            } else {
                // the wave's column is in registers now: its LDS slot becomes the histogram
                unsigned *hist = reinterpret_cast<unsigned *>(colbuf + wave * ldc);
                hist_clear(hist, lane);
                HistWarm hw{0, HIST_WARM_SHIFT0};
                res = wave_select16_hist(x, ColIdx{lane}, M, k, hist, lane, hw);
            }
            if (w.col_bits && res.cut != SELECT_UNRESOLVED)
                emit_select_bits(x, ColIdx{lane}, slot_valid_masks(ColIdx{lane}, M, lane), res, w.col_word(p, j, 0), lane, w.max_n);
        } else {
            uint64_t key[EPL];
            int idx[EPL];
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                idx[e] = e * 64 + lane;
                key[e] = idx[e] < M ? f64_key(colbuf[wave * ldc + idx[e]]) : ~0ull;
            }
            res = wave_select_kth<EPL>(key, idx, M, k);
        }
    }
    store_uniform_select(k, M, w.col_thr + (int64_t)p * w.max_n + j, w.col_cut + (int64_t)p * w.max_n + j, res);
}

// Fix-up pass of the 16-per-lane selection: one wave scans 64 thresholds of one pair; rows (DIR 0) or
// columns (DIR 1) marked SELECT_UNRESOLVED are re-selected with the general key-array routine.  On real
// data ~0.1 % of the rows are marked, so this kernel reads 4 bytes per row and returns.
template <int DIR>
__global__ __launch_bounds__(64) void select_fix_kernel(const double *__restrict__ S,
                                                        const acoss_pair_desc *__restrict__ descs, int win,
                                                        double kappa_k_fixed, int k_mode, ThreshWork w, int groups)
{
    const int p = blockIdx.x / groups, g = blockIdx.x % groups;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int count = DIR == 0 ? M : N;          // how many thresholds
    const int len = DIR == 0 ? N : M;            // elements behind each
    const int lane = threadIdx.x;
    const int t = g * 64 + lane;
    uint64_t *thr = (DIR == 0 ? w.row_thr + (int64_t)p * w.max_m : w.col_thr + (int64_t)p * w.max_n);
    int *cut = (DIR == 0 ? w.row_cut + (int64_t)p * w.max_m : w.col_cut + (int64_t)p * w.max_n);
    unsigned long long todo = __ballot(t < count && cut[t] == SELECT_UNRESOLVED);
    if (todo == 0) return;
    const int k = k_mode == 0 ? (int)rint(kappa_k_fixed * (double)len) : (k_mode == 1 ? (int)kappa_k_fixed : len);
    const double *base = S + ds.crp_off;
    while (todo) {
        const int which = g * 64 + (__ffsll((long long)todo) - 1);     // wave-uniform
        todo &= todo - 1;
        uint64_t key[16];
        int idx[16];
#pragma unroll
        for (int e = 0; e < 16; e++) {
            idx[e] = ColIdx{lane}(e);     // the bit words must match the producers'
            const int q = min(idx[e], len - 1);
            const double v = DIR == 0 ? base[(int64_t)which * ds.crp_pitch + q] : base[(int64_t)q * ds.crp_pitch + which];
            key[e] = idx[e] < len ? f64_key(v) : ~0ull;
        }
        const SelectResult res = wave_select_kth<16>(key, idx, len, k);
        if (lane == 0) {
            thr[which] = res.thr_key;
            cut[which] = res.cut;
        }
        uint64_t *bits = DIR == 0 ? w.row_bits : w.col_bits;
        if (bits) {
            bits = DIR == 0 ? w.row_bits + ((int64_t)p * w.max_m + which) * 16 : w.col_word(p, which, 0);
            const int64_t bstride = DIR == 0 ? 1 : w.max_n;
            uint64_t mine = 0;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const bool on = (idx[e] < len) & ((key[e] < res.thr_key) | ((key[e] == res.thr_key) & (idx[e] <= res.cut)));
                const uint64_t m = __ballot(on);
                if (lane == e) mine = m;
            }
            if (lane < 16) bits[lane * bstride] = mine;
        }
    }
}

// bit k of the argument -> bit 2k of the result
__device__ inline uint64_t spread_bits32(unsigned a)
{
    uint64_t x = a;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// Bit-packed mutual mask: out[p][i][cw] (uint64, bit c = column cw*64 + c) = row_bits[i][cw] & the transpose
// of col_bits.  One block per 64 rows of a pair: the rows' words come in and go out through LDS with fully
// coalesced accesses (64 rows x W words are contiguous in both arrays); wave v takes the 64 x 64 tiles cw = v, v+4, ...:
// lane l loads the column word of column cw*64 + l covering rows ri*64 .. ri*64+63, a butterfly transpose
// (wave_transpose64) turns it, lane r ends with the row word of row ri*64 + r.
constexpr int COMBINE_MAXW = 32;

// (list != nullptr: the pairs list[0 .. *list_n), `slots` of them at a time -- the pairs the radix selection hands back)
__global__ __launch_bounds__(256) void combine_bits_kernel(const acoss_pair_desc *__restrict__ descs, int K, int win,
                                                           int mutual, ThreshWork w, int tiles_m,
                                                           uint64_t *__restrict__ out, const int *__restrict__ list,
                                                           const int *__restrict__ list_n, int slots)
{
    __shared__ uint64_t rowbuf[64 * (COMBINE_MAXW + 1)];
    const int ri = blockIdx.x % tiles_m;
    const int n_list = list != nullptr ? *list_n : 1;
    for (int s = list != nullptr ? blockIdx.x / tiles_m : 0; s < n_list; s += slots) {
    const int p = list != nullptr ? list[s] : blockIdx.x / tiles_m;
    __syncthreads();
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (ri * 64 >= M) continue;
    const int W = w.wpr, ld = W + 1;
    const int rows = min(64, M - ri * 64);
    const int64_t base = ((int64_t)p * w.max_m + ri * 64) * W;          // the block's rows are contiguous: rows * W words
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // the column words of this wave's tiles are requested together with the rows (one round trip instead of two)
    constexpr int TPW = COMBINE_MAXW / 4;
    uint64_t cwd[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int j = (wave + 4 * t) * 64 + lane;
        cwd[t] = (mutual && j < N && wave + 4 * t < W) ? *w.col_word(p, j, ri) : 0ull;
    }
    for (int idx = threadIdx.x; idx < rows * W; idx += 256) rowbuf[(idx / W) * ld + idx % W] = w.row_bits[base + idx];
    __syncthreads();
    if (mutual) {
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int cw = wave + 4 * t;
            if (cw * 64 < N) {
                const uint64_t tr = wave_transpose64(cwd[t], lane);
                if (lane < rows) rowbuf[lane * ld + cw] &= tr;
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < rows * W; idx += 256) out[base + idx] = rowbuf[(idx / W) * ld + idx % W];
    }
}


// ---------------------------------------------------------------------------------------------
// Selection for matrices of any size (beyond 2048 x 2048, where a row no longer fits the registers of one wave):
// one 256-thread block per row (DIR 0) or column (DIR 1), most-significant-digit-first radix select with 8-bit
// digits over the order-preserving keys.  The row is read from memory (L2) once per digit and the search stops as
// soon as the bucket holds a single element; exact ties are cut lowest position first, like every other selection
// here.  Slow next to the register-resident kernels (a column walk is strided), but correct for every length.
// ---------------------------------------------------------------------------------------------
template <int DIR>
__global__ __launch_bounds__(256) void select_generic_kernel(const double *__restrict__ S,
                                                             const acoss_pair_desc *__restrict__ descs, int win,
                                                             double kv, int k_mode, ThreshWork w, int per_pair)
{
    __shared__ __attribute__((aligned(16))) unsigned hist[256];
    __shared__ unsigned long long sh_key;
    __shared__ int sh_digit, sh_rank, sh_count, sh_cut, sh_wave[4];
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / per_pair, which = lb % per_pair;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int count = DIR == 0 ? M : N, len = DIR == 0 ? N : M;
    if (which >= count) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int k = k_mode == 0 ? (int)rint(kv * (double)len) : (k_mode == 1 ? (int)kv : len);
    uint64_t *thr = DIR == 0 ? w.row_thr + (int64_t)p * w.max_m : w.col_thr + (int64_t)p * w.max_n;
    int *cut = DIR == 0 ? w.row_cut + (int64_t)p * w.max_m : w.col_cut + (int64_t)p * w.max_n;
    SelectResult res;
    if (trivial_select(k, len, res)) {
        if (tid == 0) { thr[which] = res.thr_key; cut[which] = res.cut; }
        return;
    }
    const double *base = S + ds.crp_off + (DIR == 0 ? (int64_t)which * ds.crp_pitch : (int64_t)which);
    const int64_t stride = DIR == 0 ? 1 : ds.crp_pitch;
    uint64_t prefix = 0, mask = 0;
    int rank = k, bucket = len, shift = 56;
    for (; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        for (int t = tid; t < len; t += 256) {
            const uint64_t key = f64_key(base[t * stride]);
            if ((key & mask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            // lane l owns digits 4l .. 4l+3
            const uint4 c = reinterpret_cast<const uint4 *>(hist)[lane];
            const int tot = (int)(c.x + c.y + c.z + c.w);
            const int incl = wave_scan<OpAdd>(tot, 0);
            const bool mine = (incl - tot < rank) & (rank <= incl);       // exactly one lane
            if (mine) {
                int r = rank - (incl - tot), d = 0, cnt = (int)c.x;
                if (r > cnt) { r -= cnt; d = 1; cnt = (int)c.y;
                    if (r > cnt) { r -= cnt; d = 2; cnt = (int)c.z;
                        if (r > cnt) { r -= cnt; d = 3; cnt = (int)c.w; } } }
                sh_digit = 4 * lane + d;
                sh_rank = r;
                sh_count = cnt;
            }
        }
        __syncthreads();
        prefix |= (uint64_t)(unsigned)sh_digit << shift;
        mask |= 255ull << shift;
        rank = sh_rank;
        bucket = sh_count;
        if (bucket == 1) break;
    }
    res.cut = 0x7fffffff;
    if (shift > 0) {
        // one element left in the bucket: it is the k-th smallest
        for (int t = tid; t < len; t += 256) {
            const uint64_t key = f64_key(base[t * stride]);
            if ((key & mask) == prefix) sh_key = key;
        }
        __syncthreads();
        res.thr_key = sh_key;
    } else {
        res.thr_key = prefix;
        if (bucket > rank) {
            // `bucket` equal keys, `rank` of them are taken: the rank-th lowest position is the cut
            int seen = 0;
            if (tid == 0) sh_cut = 0x7fffffff;
            for (int c0 = 0; c0 < len && seen < rank; c0 += 256) {
                const int t = c0 + tid;
                const bool eq = t < len && f64_key(base[t * stride]) == prefix;
                const uint64_t bal = __ballot(eq);
                __syncthreads();
                if (lane == 0) sh_wave[tid >> 6] = __popcll(bal);
                __syncthreads();
                int before = seen;
                for (int v = 0; v < (tid >> 6); v++) before += sh_wave[v];
                before += __popcll(bal & ((1ull << lane) - 1ull));
                if (eq && before + 1 == rank) sh_cut = t;
                seen += sh_wave[0] + sh_wave[1] + sh_wave[2] + sh_wave[3];
            }
            __syncthreads();
            res.cut = sh_cut;
        }
    }
    if (tid == 0) { thr[which] = res.thr_key; cut[which] = res.cut; }
}


// B[i][j] = row rule (and column rule when mutual).  A thread owns 4 adjacent columns (its column
// thresholds stay in registers) and walks MASK_ROWS rows, with all of its 32-byte row loads issued
// before the first compare; a wave covers 256 contiguous columns (2 KB) of each row.
constexpr int MASK_ROWS = 8;

__global__ __launch_bounds__(256) void mask_kernel(const double *__restrict__ S,
                                                   const acoss_pair_desc *__restrict__ descs, int win,
                                                   int mutual, ThreshWork w, int col_blocks, int row_blocks,
                                                   uint8_t *__restrict__ B)
{
    const int per_pair = col_blocks * row_blocks;
    const int p = blockIdx.x / per_pair, t = blockIdx.x % per_pair;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int j = (t % col_blocks) * 1024 + threadIdx.x * 4;
    const int i0 = (t / col_blocks) * MASK_ROWS;
    if (i0 >= M || j >= N) return;
    const bool vec = ((ds.crp_pitch & 3) == 0) && ((ds.crp_off & 3) == 0) && j + 3 < N;
    uint64_t ct[4];
    int cc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int jc = min(j + c, N - 1);
        ct[c] = mutual ? w.col_thr[(int64_t)p * w.max_n + jc] : ~0ull;
        cc[c] = mutual ? w.col_cut[(int64_t)p * w.max_n + jc] : 0x7fffffff;
    }
    const double *base = S + ds.crp_off + j;
    double v[MASK_ROWS][4];
    uint64_t rt[MASK_ROWS];
    int rc[MASK_ROWS];
#pragma unroll
    for (int r = 0; r < MASK_ROWS; r++) {
        const int i = min(i0 + r, M - 1);
        const double *row = base + (int64_t)i * ds.crp_pitch;
        if (vec) {
            const double2 a = *reinterpret_cast<const double2 *>(row), b = *reinterpret_cast<const double2 *>(row + 2);
            v[r][0] = a.x; v[r][1] = a.y; v[r][2] = b.x; v[r][3] = b.y;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) v[r][c] = row[min(c, N - 1 - j)];
        }
        rt[r] = w.row_thr[(int64_t)p * w.max_m + i];
        rc[r] = w.row_cut[(int64_t)p * w.max_m + i];
    }
#pragma unroll
    for (int r = 0; r < MASK_ROWS; r++) {
        const int i = i0 + r;
        if (i >= M) break;
        uint32_t packed = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint64_t key = f64_key(v[r][c]);
            bool on = key < rt[r] || (key == rt[r] && j + c <= rc[r]);
            on = on && (key < ct[c] || (key == ct[c] && i <= cc[c]));
            packed |= (on ? 1u : 0u) << (8 * c);
        }
        uint8_t *brow = B + ds.crp_off + (int64_t)i * ds.crp_pitch + j;
        if (vec) {
            *reinterpret_cast<uint32_t *>(brow) = packed;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (j + c < N) brow[c] = (uint8_t)((packed >> (8 * c)) & 0xff);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
template <typename T>
static int launch_norms(const T *feats, int64_t n_frames, int d, T *norms, hipStream_t st)
{
    if (!feats || !norms || n_frames < 0 || d < 1) { set_error("frame_norms: bad argument"); return ACOSS_EINVAL; }
    if (n_frames == 0) return ACOSS_OK;
    hipLaunchKernelGGL(frame_norms_kernel<T>, dim3((unsigned)ceil_div64(n_frames, 256)), dim3(256), 0, st,
                       feats, n_frames, d, norms);
    return launch_check("frame_norms_kernel");
}

template <typename T>
static int launch_csm(const T *feats, const T *norms, int d, const acoss_pair_desc *descs, int K,
                      int max_nx, int max_ny, T *csm, hipStream_t st)
{
    if (!feats || !norms || !descs || !csm || K < 0 || d < 1 || max_nx < 1 || max_ny < 1) {
        set_error("csm_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if (K == 0) return ACOSS_OK;
    if (d == 12 || d == 13) {
        const int tm = ceil_div(max_nx, CSM_TM), tn = ceil_div(max_ny, CSM_TN);
        const int64_t blocks = (int64_t)K * tm * tn;
        if (blocks > 0x7fffffffLL) { set_error("csm_batch: batch too large for one launch"); return ACOSS_ENOTSUP; }
        if (d == 12)
            hipLaunchKernelGGL((csm_kernel<T, 12>), dim3((unsigned)blocks), dim3(256), 0, st, feats, norms, descs, tm, tn, csm);
        else
            hipLaunchKernelGGL((csm_kernel<T, 13>), dim3((unsigned)blocks), dim3(256), 0, st, feats, norms, descs, tm, tn, csm);
        return launch_check("csm_kernel");
    }
    if constexpr (sizeof(T) == 8) {
        if (d >= 32) {        // wide float64 features: matrix cores
            const int tm = ceil_div(max_nx, GM_T), tn = ceil_div(max_ny, GM_TJ);
            const int64_t blocks = (int64_t)K * tm * tn;
            if (blocks > 0x7fffffffLL) { set_error("csm_batch: batch too large for one launch"); return ACOSS_ENOTSUP; }
            hipLaunchKernelGGL(csm_gemm_kernel, dim3((unsigned)blocks), dim3(GM_THREADS), 0, st, feats, norms, d, descs, tm, tn, csm);
            return launch_check("csm_gemm_kernel");
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (d >= 32) {        // wide float32 features: matrix cores
            // (128 x 128 tiles, two blocks per CU: 111 TFLOP/s at 992 x 20736 x 992; 256 x 128 tiles with one block per CU: 87)
            const int tm = ceil_div(max_nx, 128), tn = ceil_div(max_ny, 128);
            const int64_t blocks = (int64_t)K * tm * tn;
            if (blocks > 0x7fffffffLL) { set_error("csm_batch: batch too large for one launch"); return ACOSS_ENOTSUP; }
            const size_t lds = sizeof(Gemm32Smem<2, 4>);
            hipLaunchKernelGGL(csm_gemm32_kernel<2>, dim3((unsigned)blocks), dim3(GM32_THREADS), lds, st, feats, norms, d, descs, tm, tn, csm);
            return launch_check("csm_gemm32_kernel");
        }
    }
    const int tm = ceil_div(max_nx, 16), tn = ceil_div(max_ny, 16);
    const int64_t blocks = (int64_t)K * tm * tn;
    if (blocks > 0x7fffffffLL) { set_error("csm_batch: batch too large for one launch"); return ACOSS_ENOTSUP; }
    hipLaunchKernelGGL(csm_generic_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, feats, norms, d, descs, tm, tn, csm);
    return launch_check("csm_generic_kernel");
}

template <typename T>
static int launch_sliding(const T *csm, const acoss_pair_desc *descs, int K, int win, int max_nx,
                          int max_ny, double *S, hipStream_t st)
{
    if (!csm || !descs || !S || K < 0 || win < 1 || win > 64 || max_nx < win || max_ny < win) {
        set_error("sliding_batch: bad argument (1 <= win <= 64, every song >= win frames)");
        return ACOSS_EINVAL;
    }
    if (K == 0) return ACOSS_OK;
    const int tm = ceil_div(max_nx - win + 1, SL_TM), tn = ceil_div(max_ny - win + 1, SL_TN);
    const int64_t blocks = (int64_t)K * tm * tn;
    if (blocks > 0x7fffffffLL) { set_error("sliding_batch: batch too large for one launch"); return ACOSS_ENOTSUP; }
    const size_t lds = sizeof(double) * (size_t)(SL_TM + win - 1) * (size_t)((SL_TN + win - 1) | 1);
    hipLaunchKernelGGL(sliding_kernel<T>, dim3((unsigned)blocks), dim3(256), lds, st, csm, descs, win, tm, tn, S);
    return launch_check("sliding_kernel");
}

static void kappa_mode(double kappa, double &kv, int &mode)
{
    if (kappa == 0.0) { kv = 0.0; mode = 2; }       // CRPUtils.py:188-189
    else if (kappa < 1.0) { kv = kappa; mode = 0; }  // :190-191
    else { kv = kappa; mode = 1; }                   // :192-193
}

// shared with planar_kernels.hip
int launch_combine_bits(const acoss_pair_desc *descs, int K, int win, int mutual, ThreshWork w, uint64_t *bits, hipStream_t st)
{
    const int tm = ceil_div(w.max_m, 64);       // every word of every row is written
    if (w.wpr > COMBINE_MAXW || (int64_t)K * tm > 0x7fffffffLL) { set_error("combine_bits: batch too large"); return ACOSS_ENOTSUP; }
    hipLaunchKernelGGL(combine_bits_kernel, dim3((unsigned)((int64_t)K * tm)), dim3(256), 0, st, descs, K, win, mutual, w, tm, bits,
                       (const int *)nullptr, (const int *)nullptr, 1);
    return launch_check("combine_bits_kernel");
}

int launch_combine_bits_list(const acoss_pair_desc *descs, int K, int win, int mutual, ThreshWork w, uint64_t *bits, const int *list,
                             const int *list_n, int slots, hipStream_t st)
{
    const int tm = ceil_div(w.max_m, 64);
    if (w.wpr > COMBINE_MAXW) { set_error("combine_bits: batch too large"); return ACOSS_ENOTSUP; }
    hipLaunchKernelGGL(combine_bits_kernel, dim3((unsigned)(slots * tm)), dim3(256), 0, st, descs, K, win, mutual, w, tm, bits, list, list_n, slots);
    return launch_check("combine_bits_kernel (list)");
}

}  // namespace acoss

using namespace acoss;

extern "C" {

int acoss_frame_norms_f64(const double *feats, int64_t n_frames, int d, double *norms, void *stream)
{
    return launch_norms<double>(feats, n_frames, d, norms, (hipStream_t)stream);
}
int acoss_frame_norms_f32(const float *feats, int64_t n_frames, int d, float *norms, void *stream)
{
    return launch_norms<float>(feats, n_frames, d, norms, (hipStream_t)stream);
}

int acoss_oti_batch(const double *gchroma, int nbins, acoss_pair_desc *descs, int K, void *stream)
{
    if (!gchroma || !descs || K < 0 || nbins < 1 || nbins > 64) {
        set_error("oti_batch: bad argument (1 <= nbins <= 64)");
        return ACOSS_EINVAL;
    }
    if (K == 0) return ACOSS_OK;
    hipLaunchKernelGGL(oti_kernel, dim3(ceil_div(K, 64)), dim3(64), 0, (hipStream_t)stream, gchroma, nbins, descs, K);
    return launch_check("oti_kernel");
}

int acoss_csm_batch_f64(const double *feats, const double *norms, int d, const acoss_pair_desc *descs,
                        int K, int max_nx, int max_ny, double *csm, void *stream)
{
    return launch_csm<double>(feats, norms, d, descs, K, max_nx, max_ny, csm, (hipStream_t)stream);
}
int acoss_csm_batch_f32(const float *feats, const float *norms, int d, const acoss_pair_desc *descs,
                        int K, int max_nx, int max_ny, float *csm, void *stream)
{
    return launch_csm<float>(feats, norms, d, descs, K, max_nx, max_ny, csm, (hipStream_t)stream);
}

#ifdef ACOSS_PROBES      // python -m acoss_amd.build --probes: the MODE != 0 instantiations exist only in that build
// development probe (not part of the public ABI): select_rows in a probe MODE
int acoss_dev_select_probe(int mode, const double *S, const acoss_pair_desc *descs, int K, int win, int max_nx,
                           int max_ny, double kappa, void *work, void *stream)
{
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    ThreshWork w = thresh_work_layout(work, K, max_m, max_n);
    const int rb = ceil_div(max_m, 4 * SEL_ROWS_PER_WAVE);
    const unsigned blocks = (unsigned)((int64_t)K * rb);
    hipStream_t st = (hipStream_t)stream;
    if (mode >= 10) {       // column kernel: 10 = product, 11 = loads only, 13 = sorted / probing selection
        const int cb = ceil_div(max_n, SEL_COLS_PER_BLOCK);
        const size_t lds = (size_t)SEL_COLS_PER_BLOCK * (16 * 64 + 2) * sizeof(double);
        const unsigned cblocks = (unsigned)((int64_t)K * cb);
        if (mode == 11) {
            ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_kernel<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((select_cols_kernel<16, 1>), dim3(cblocks), dim3(SEL_COLS_THREADS), lds, st, S, descs, win, kappa, 0, w, cb);
        } else if (mode == 13) {
            ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_kernel<16, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((select_cols_kernel<16, 3>), dim3(cblocks), dim3(SEL_COLS_THREADS), lds, st, S, descs, win, kappa, 0, w, cb);
        } else {
            ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_kernel<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((select_cols_kernel<16, 0>), dim3(cblocks), dim3(SEL_COLS_THREADS), lds, st, S, descs, win, kappa, 0, w, cb);
        }
        return launch_check("select_cols probe");
    }
    if (mode == 1) hipLaunchKernelGGL((select_rows_kernel<16, 1>), dim3(blocks), dim3(256), 0, st, S, descs, win, kappa, 0, w, rb);
    else if (mode == 2) hipLaunchKernelGGL((select_rows_kernel<16, 2>), dim3(blocks), dim3(256), 0, st, S, descs, win, kappa, 0, w, rb);
    else if (mode == 3) hipLaunchKernelGGL((select_rows_kernel<16, 3>), dim3(blocks), dim3(256), 0, st, S, descs, win, kappa, 0, w, rb);
    else hipLaunchKernelGGL((select_rows_kernel<16, 0>), dim3(blocks), dim3(256), 0, st, S, descs, win, kappa, 0, w, rb);
    return launch_check("select_rows probe");
}

// development probe (not part of the public ABI): the float64 d=12 CSM kernel in a probe MODE
int acoss_dev_csm_probe(int mode, const double *feats, const double *norms, const acoss_pair_desc *descs,
                        int K, int max_nx, int max_ny, double *csm, void *stream)
{
    const int tm = ceil_div(max_nx, CSM_TM), tn = ceil_div(max_ny, CSM_TN);
    const unsigned blocks = (unsigned)((int64_t)K * tm * tn);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1) hipLaunchKernelGGL((csm_kernel<double, 12, 1>), dim3(blocks), dim3(256), 0, st, feats, norms, descs, tm, tn, csm);
    else if (mode == 2) hipLaunchKernelGGL((csm_kernel<double, 12, 2>), dim3(blocks), dim3(256), 0, st, feats, norms, descs, tm, tn, csm);
    else if (mode == 3) hipLaunchKernelGGL((csm_kernel<double, 12, 3>), dim3(blocks), dim3(256), 0, st, feats, norms, descs, tm, tn, csm);
    else hipLaunchKernelGGL((csm_kernel<double, 12, 0>), dim3(blocks), dim3(256), 0, st, feats, norms, descs, tm, tn, csm);
    return launch_check("csm_kernel probe");
}
#endif  // ACOSS_PROBES

int acoss_sliding_batch_f64(const double *csm, const acoss_pair_desc *descs, int K, int win,
                            int max_nx, int max_ny, double *S, void *stream)
{
    return launch_sliding<double>(csm, descs, K, win, max_nx, max_ny, S, (hipStream_t)stream);
}
int acoss_sliding_batch_f32(const float *csm, const acoss_pair_desc *descs, int K, int win,
                            int max_nx, int max_ny, double *S, void *stream)
{
    return launch_sliding<float>(csm, descs, K, win, max_nx, max_ny, S, (hipStream_t)stream);
}

size_t acoss_binarize_work_bytes(int K, int max_nx, int max_ny, int win)
{
    const size_t m = (size_t)(max_nx - win + 1 > 0 ? max_nx - win + 1 : 0);
    const size_t n = (size_t)(max_ny - win + 1 > 0 ? max_ny - win + 1 : 0);
    return (size_t)(K > 0 ? K : 0) * (m + n) * (sizeof(uint64_t) + sizeof(int)) + 64;
}

static int run_thresholds(const double *S, const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                          double kappa, int mutual, void *work, size_t work_bytes, hipStream_t st, ThreshWork &w,
                          bool with_bits = false)
{
    if (!S || !descs || !work || K < 0 || win < 1 || max_nx < win || max_ny < win || kappa < 0.0) {
        set_error("thresholds: bad argument");
        return ACOSS_EINVAL;
    }
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    if (work_bytes < thresh_work_bytes(K, max_m, max_n, with_bits)) {
        set_error("thresholds: workspace too small");
        return ACOSS_EINVAL;
    }
    if (with_bits && (max_m > 1024 || max_n > 1024)) {
        set_error("mask_bits: matrices larger than 1024 x 1024 are not supported (acoss_mask_bits_planar_batch goes to 2048)");
        return ACOSS_ENOTSUP;
    }
    w = thresh_work_layout(work, K, max_m, max_n, with_bits);
    if (K == 0) return ACOSS_OK;
    double kv;
    int mode;
    kappa_mode(kappa, kv, mode);
    {
        const int rb = ceil_div(max_m, 4 * SEL_ROWS_PER_WAVE);
        if (max_n > 2048) {
            if ((int64_t)K * max_m > 0x7fffffffLL) { set_error("thresholds: batch too large"); return ACOSS_ENOTSUP; }
            hipLaunchKernelGGL(select_generic_kernel<0>, dim3((unsigned)((int64_t)K * max_m)), dim3(256), 0, st, S, descs, win, kv, mode, w, max_m);
        } else if (max_n <= 1024)
            hipLaunchKernelGGL((select_rows_kernel<16>), dim3((unsigned)((int64_t)K * rb)), dim3(256), 0, st, S, descs, win, kv, mode, w, rb);
        else
            hipLaunchKernelGGL((select_rows_kernel<32>), dim3((unsigned)((int64_t)K * rb)), dim3(256), 0, st, S, descs, win, kv, mode, w, rb);
        int rc = launch_check("select_rows_kernel");
        if (rc) return rc;
        if (max_n <= 1024) {
            const int groups = ceil_div(max_m, 64);
            hipLaunchKernelGGL(select_fix_kernel<0>, dim3((unsigned)((int64_t)K * groups)), dim3(64), 0, st, S, descs, win, kv, mode, w, groups);
            rc = launch_check("select_fix_kernel<rows>");
            if (rc) return rc;
        }
    }
    if (mutual) {
        const int cb = ceil_div(max_n, SEL_COLS_PER_BLOCK);
        if (max_m > 2048) {
            if ((int64_t)K * max_n > 0x7fffffffLL) { set_error("thresholds: batch too large"); return ACOSS_ENOTSUP; }
            hipLaunchKernelGGL(select_generic_kernel<1>, dim3((unsigned)((int64_t)K * max_n)), dim3(256), 0, st, S, descs, win, kv, mode, w, max_n);
        } else if (max_m <= 1024) {
            const size_t lds = sizeof(double) * SEL_COLS_PER_BLOCK * (16 * 64 + 2);
            ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(select_cols_kernel<16>, dim3((unsigned)((int64_t)K * cb)), dim3(SEL_COLS_THREADS), lds, st, S, descs, win, kv, mode, w, cb);
        } else {
            const size_t lds = sizeof(double) * SEL_COLS_PER_BLOCK * (32 * 64 + 2);
            ACOSS_HIP(hipFuncSetAttribute((const void *)select_cols_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(select_cols_kernel<32>, dim3((unsigned)((int64_t)K * cb)), dim3(SEL_COLS_THREADS), lds, st, S, descs, win, kv, mode, w, cb);
        }
        int rc = launch_check("select_cols_kernel");
        if (rc) return rc;
        if (max_m <= 1024) {
            const int groups = ceil_div(max_n, 64);
            hipLaunchKernelGGL(select_fix_kernel<1>, dim3((unsigned)((int64_t)K * groups)), dim3(64), 0, st, S, descs, win, kv, mode, w, groups);
            rc = launch_check("select_fix_kernel<cols>");
            if (rc) return rc;
        }
    }
    return ACOSS_OK;
}

int acoss_thresholds_batch(const double *S, const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                           double kappa, int mutual, void *work, size_t work_bytes, void *stream)
{
    ThreshWork w;
    return run_thresholds(S, descs, K, win, max_nx, max_ny, kappa, mutual, work, work_bytes, (hipStream_t)stream, w);
}

int acoss_mask_bits_words(int max_nx, int max_ny, int win)
{
    return mask_bits_words(max_nx - win + 1, max_ny - win + 1);
}

size_t acoss_mask_bits_work_bytes(int K, int max_nx, int max_ny, int win)
{
    return thresh_work_bytes(K > 0 ? K : 0, max_nx - win + 1 > 0 ? max_nx - win + 1 : 0,
                             max_ny - win + 1 > 0 ? max_ny - win + 1 : 0, true);
}

int acoss_mask_bits_batch(const double *S, const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                          double kappa, int mutual, uint64_t *bits, void *work, size_t work_bytes, void *stream)
{
    if (!bits) { set_error("mask_bits_batch: bad argument"); return ACOSS_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ThreshWork w;
    int rc = run_thresholds(S, descs, K, win, max_nx, max_ny, kappa, mutual, work, work_bytes, st, w, true);
    if (rc || K == 0) return rc;
    return launch_combine_bits(descs, K, win, mutual, w, bits, st);
}

int acoss_binarize_batch(const double *S, const acoss_pair_desc *descs, int K, int win, int max_nx,
                         int max_ny, double kappa, int mutual, uint8_t *B, void *work,
                         size_t work_bytes, void *stream)
{
    if (!B) { set_error("binarize_batch: bad argument"); return ACOSS_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ThreshWork w;
    int rc = run_thresholds(S, descs, K, win, max_nx, max_ny, kappa, mutual, work, work_bytes, st, w);
    if (rc || K == 0) return rc;
    const int cbk = ceil_div(w.max_n, 1024), rbk = ceil_div(w.max_m, MASK_ROWS);
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)((int64_t)K * cbk * rbk)), dim3(256), 0, st, S, descs, win, mutual, w, cbk, rbk, B);
    return launch_check("mask_kernel");
}

}  // extern "C"
