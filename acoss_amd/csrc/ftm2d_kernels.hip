// ftm2d_kernels.hip -- the 2D-Fourier-transform-magnitude cover-song feature and its all-pairs similarity
// (benchmarking/FTM2D.py).
//
// Per song (FTM2D.py:92-100), from beat-synchronous chroma (nbeats x 12):
//   chrompwr (:9-25)              columns raised to the power P, norms preserved
//   btchroma_to_fftmat (:29-48)   for every window of 75 beats: |fft2| of the 12 x 75 patch, fftshift, flattened
//   per-window L2 norm, log(C x / norm + 1), median over the windows, L2 normalisation  ->  900 numbers
// Similarity of two songs (:122-126): exp(-|s1 - s2|^2); for all pairs at once that is one N x 900 x N product on
// the float64 matrix cores with an exp epilogue.
//
// The transform is done as two direct DFTs (12-point down the chroma axis, 75-point along the beats) in float64:
// 75 = 3 * 5^2 has no power-of-two structure worth the trouble, and one window is only 67 500 complex
// multiply-adds.
#include "common.h"
#include "wave_ops.h"
#include "gemm_f64.h"

namespace acoss {

constexpr int FT_BINS = 12, FT_WIN = 75, FT_DIM = FT_BINS * FT_WIN;      // 900
constexpr int FT_MAX_WINDOWS = 2048;                                     // median by 32-per-lane selection

// ---- chrompwr, one thread per beat --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ftm2d_chrompwr_kernel(const double *__restrict__ bt, int64_t n_beats, double P,
                                                             double *__restrict__ out)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_beats) return;
    double x[FT_BINS], s = 0.0;
#pragma unroll
    for (int c = 0; c < FT_BINS; c++) {
        x[c] = bt[b * FT_BINS + c];
        s += x[c] * x[c];
    }
    double cmn = sqrt(s);                    // FTM2D.py:17-18
    if (cmn == 0.0) cmn = 1.0;
    double s2 = 0.0;
#pragma unroll
    for (int c = 0; c < FT_BINS; c++) {
        x[c] = pow(x[c] / cmn, P);           // :21
        s2 += x[c] * x[c];
    }
    double cmpn = sqrt(s2);                  // :23-24
    if (cmpn == 0.0) cmpn = 1.0;
#pragma unroll
    for (int c = 0; c < FT_BINS; c++) out[b * FT_BINS + c] = cmn * (x[c] / cmpn);      // :25
}

// ---- one block per window: |fft2|, fftshift, norm, log -------------------------------------------------------
__global__ __launch_bounds__(256) void ftm2d_window_kernel(const double *__restrict__ chroma,      // [beats][12], after chrompwr
                                                           const int64_t *__restrict__ beat_off,  // [n_songs + 1]
                                                           const int64_t *__restrict__ win_off,   // [n_songs + 1]
                                                           int n_songs, double C, double *__restrict__ V)     // [windows][900]
{
    __shared__ double xw[FT_WIN][FT_BINS + 1];
    __shared__ double yre[FT_BINS][FT_WIN + 1], yim[FT_BINS][FT_WIN + 1];
    __shared__ double tw_re[FT_WIN], tw_im[FT_WIN], t12_re[FT_BINS], t12_im[FT_BINS];
    __shared__ double red[4];
    const int64_t w = blockIdx.x;
    // the song this window belongs to: last s with win_off[s] <= w
    int lo = 0, hi = n_songs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (win_off[mid] <= w) lo = mid; else hi = mid;
    }
    const int64_t b0 = beat_off[lo] + (w - win_off[lo]);
    for (int e = threadIdx.x; e < FT_WIN * FT_BINS; e += 256) xw[e / FT_BINS][e % FT_BINS] = chroma[b0 * FT_BINS + e];
    if (threadIdx.x < FT_WIN) sincospi(-2.0 * (double)threadIdx.x / FT_WIN, &tw_im[threadIdx.x], &tw_re[threadIdx.x]);
    if (threadIdx.x >= 128 && threadIdx.x < 128 + FT_BINS)
        sincospi(-2.0 * (double)(threadIdx.x - 128) / FT_BINS, &t12_im[threadIdx.x - 128], &t12_re[threadIdx.x - 128]);
    __syncthreads();
    // 12-point DFT down the chroma axis: Y[k][t] = sum_c x[t][c] e^{-2 pi i k c / 12}
    for (int e = threadIdx.x; e < FT_DIM; e += 256) {
        const int k = e / FT_WIN, t = e % FT_WIN;
        double re = 0.0, im = 0.0;
        int m = 0;
#pragma unroll
        for (int c = 0; c < FT_BINS; c++) {
            re = fma(xw[t][c], t12_re[m], re);
            im = fma(xw[t][c], t12_im[m], im);
            m += k;
            if (m >= FT_BINS) m -= FT_BINS;
        }
        yre[k][t] = re;
        yim[k][t] = im;
    }
    __syncthreads();
    // 75-point DFT along the beats, magnitude, fftshift (scipy.fftpack.fftshift: index i -> (i + n/2) % n)
    double mag[4] = {0.0, 0.0, 0.0, 0.0}, ss = 0.0;
    int place[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int e = threadIdx.x + 256 * q;
        if (e < FT_DIM) {
            const int k = e / FT_WIN, l = e % FT_WIN;
            double re = 0.0, im = 0.0;
            int m = 0;
            for (int t = 0; t < FT_WIN; t++) {
                const double a = yre[k][t], b = yim[k][t], c = tw_re[m], s = tw_im[m];
                re = fma(a, c, fma(-b, s, re));
                im = fma(a, s, fma(b, c, im));
                m += l;
                if (m >= FT_WIN) m -= FT_WIN;
            }
            mag[q] = sqrt(re * re + im * im);                     // FTM2D.py:45
            ss += mag[q] * mag[q];
            place[q] = ((k + FT_BINS / 2) % FT_BINS) * FT_WIN + (l + FT_WIN / 2) % FT_WIN;
        }
    }
    // per-window norm (:95-96): wave sums, then the four partials in a fixed order
    {
        double v = ss;
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    }
    __syncthreads();
    double norm = sqrt((red[0] + red[1]) + (red[2] + red[3]));
    if (norm == 0.0) norm = 1.0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (threadIdx.x + 256 * q < FT_DIM) V[w * FT_DIM + place[q]] = log(C * mag[q] / norm + 1.0);      // :97
}

// ---- median over a song's windows (:98), one wave per dimension ----------------------------------------------
__global__ __launch_bounds__(256) void ftm2d_median_kernel(const double *__restrict__ V, const int64_t *__restrict__ win_off,
                                                           double *__restrict__ med)      // [n_songs][900]
{
    const int s = blockIdx.y;
    const int dim = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (dim >= FT_DIM) return;
    const int64_t w0 = win_off[s];
    const int W = (int)(win_off[s + 1] - w0);
    if (W <= 0) {
        if (lane == 0) med[(int64_t)s * FT_DIM + dim] = 0.0;         // too few beats: zeros(900) (:87-90)
        return;
    }
    uint64_t key[32];
    int idx[32];
#pragma unroll
    for (int e = 0; e < 32; e++) {
        idx[e] = e * 64 + lane;
        key[e] = idx[e] < W ? f64_key(V[(w0 + idx[e]) * FT_DIM + dim]) : ~0ull;
    }
    const int k = (W + 1) / 2;                                        // lower middle (1-based)
    double m = f64_from_key(wave_select_kth<32>(key, idx, W, k).thr_key);
    if ((W & 1) == 0) m = (m + f64_from_key(wave_select_kth<32>(key, idx, W, k + 1).thr_key)) / 2.0;      // np.median: mean of the two
    if (lane == 0) med[(int64_t)s * FT_DIM + dim] = m;
}

// ---- L2 normalisation (:99), one block per song ---------------------------------------------------------------
__global__ __launch_bounds__(256) void ftm2d_normalize_kernel(const double *__restrict__ med, const int64_t *__restrict__ win_off,
                                                              double *__restrict__ shingles)
{
    __shared__ double red[4];
    const int s = blockIdx.x;
    const double *m = med + (int64_t)s * FT_DIM;
    const bool empty = win_off[s + 1] == win_off[s];
    double ss = 0.0;
    for (int e = threadIdx.x; e < FT_DIM; e += 256) ss += m[e] * m[e];
    for (int d = 32; d > 0; d >>= 1) ss += __shfl_down(ss, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const double norm = sqrt((red[0] + red[1]) + (red[2] + red[3]));
    for (int e = threadIdx.x; e < FT_DIM; e += 256) shingles[(int64_t)s * FT_DIM + e] = empty ? 0.0 : m[e] / norm;
}

// ---- similarity of listed pairs (:117-127), one wave per pair ---------------------------------------------------
__global__ __launch_bounds__(256) void ftm2d_pairs_kernel(const double *__restrict__ shingles, const int32_t *__restrict__ pairs,
                                                          int K, double *__restrict__ sims)
{
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= K) return;
    const int lane = threadIdx.x & 63;
    const double *a = shingles + (int64_t)pairs[2 * p] * FT_DIM, *b = shingles + (int64_t)pairs[2 * p + 1] * FT_DIM;
    double d = 0.0;
    for (int e = lane; e < FT_DIM; e += 64) {
        const double t = a[e] - b[e];
        d = fma(t, t, d);
    }
    for (int s = 32; s > 0; s >>= 1) d += __shfl_down(d, s, 64);
    if (lane == 0) sims[p] = exp(-d);
}

// ---- all pairs: exp(-(|a|^2 + |b|^2 - 2 a.b)) on v_mfma_f64_16x16x4_f64 (gemm_f64.h), norms from a pre-pass ---------
__global__ __launch_bounds__(256) void ftm2d_norms_kernel(const double *__restrict__ S, int n, double *__restrict__ norms)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n) return;
    double a = 0.0;
    for (int e = lane; e < FT_DIM; e += 64) a = fma(S[(int64_t)s * FT_DIM + e], S[(int64_t)s * FT_DIM + e], a);
    for (int d = 32; d > 0; d >>= 1) a += __shfl_down(a, d, 64);
    if (lane == 0) norms[s] = a;
}

__global__ __launch_bounds__(GM_THREADS) void ftm2d_gram_kernel(const double *__restrict__ S, const double *__restrict__ norms,
                                                                int n, double *__restrict__ out)
{
    __shared__ GemmSmem sm;
    const int i0 = blockIdx.y * GM_T, j0 = blockIdx.x * GM_TJ;
    gemm_nt_tile_f64_rows(
        sm, FT_DIM, S + (int64_t)i0 * FT_DIM, FT_DIM, n - i0, S + (int64_t)j0 * FT_DIM, FT_DIM, n - j0,
        [&](const int i, const int j, const double v) {
            if (i0 + i < n && j0 + j < n) {
                const double d = fmax(fma(-2.0, v, norms[i0 + i] + norms[j0 + j]), 0.0);
                out[(int64_t)(i0 + i) * n + j0 + j] = exp(-d);
            }
        });
}

#ifndef GM_DMA
#define GM_DMA 1
#endif
// the same through gemm_nt_tile_f64_dma (gemm_f64.h: operands into LDS by the DMA path, 128 x 128 tiles): FT_DIM is even, the shingle
// matrix 16-byte aligned
__global__ __launch_bounds__(GD_THREADS) void ftm2d_gram_dma_kernel(const double *__restrict__ S, const double *__restrict__ norms,
                                                                    int n, double *__restrict__ out)
{
    __shared__ GemmDmaSmem sm;
    const int i0 = blockIdx.y * GD_T, j0 = blockIdx.x * GD_T;
    gemm_nt_tile_f64_dma(
        sm, FT_DIM, S + (int64_t)i0 * FT_DIM, FT_DIM, n - i0, S + (int64_t)j0 * FT_DIM, FT_DIM, n - j0,
        [&](const int i, const int j, const double v) {
            if (i0 + i < n && j0 + j < n) {
                const double d = fmax(fma(-2.0, v, norms[i0 + i] + norms[j0 + j]), 0.0);
                out[(int64_t)(i0 + i) * n + j0 + j] = exp(-d);
            }
        });
}

}  // namespace acoss

using namespace acoss;

extern "C" {

size_t acoss_ftm2d_scratch_bytes(int64_t total_beats, int64_t total_windows, int n_songs)
{
    return sizeof(double) * ((size_t)total_beats * FT_BINS + (size_t)total_windows * FT_DIM + (size_t)n_songs * FT_DIM) + 256;
}

int acoss_ftm2d_shingles(const double *btchroma, const int64_t *beat_off_host, int n_songs, double pwr, double C,
                         void *scratch, size_t scratch_bytes, double *shingles, void *stream)
{
    if (!btchroma || !beat_off_host || !scratch || !shingles || n_songs < 1) {
        set_error("ftm2d_shingles: bad argument");
        return ACOSS_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    // beat offsets relative to the first song, and window offsets (W = nbeats - 74, FTM2D.py:38-43; none below 75 beats, :87)
    const int64_t b_first = beat_off_host[0];
    const int64_t total_beats = beat_off_host[n_songs] - b_first;
    int64_t total_windows = 0;
    int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(n_songs + 1));
    if (!offs) { set_error("ftm2d_shingles: out of host memory"); return ACOSS_ENOMEM; }
    int64_t *woff = offs + (n_songs + 1);
    for (int s = 0; s <= n_songs; s++) offs[s] = beat_off_host[s] - b_first;
    for (int s = 0; s < n_songs; s++) {
        const int64_t nb = offs[s + 1] - offs[s];
        const int64_t W = nb >= FT_WIN ? nb - FT_WIN + 1 : 0;
        if (nb < 0 || W > FT_MAX_WINDOWS) {
            set_error("ftm2d_shingles: song %d has %lld beats (supported: up to %d)", s, (long long)nb, FT_MAX_WINDOWS + FT_WIN - 1);
            free(offs);
            return nb < 0 ? ACOSS_EINVAL : ACOSS_ENOTSUP;
        }
        woff[s] = total_windows;
        total_windows += W;
    }
    woff[n_songs] = total_windows;
    if (scratch_bytes < acoss_ftm2d_scratch_bytes(total_beats, total_windows, n_songs)) {
        free(offs);
        set_error("ftm2d_shingles: scratch too small");
        return ACOSS_EINVAL;
    }
    // scratch: chrompwr output (total_beats x 12) | V (total_windows x 900) | medians (n_songs x 900)
    double *d_chroma = (double *)scratch;
    double *d_V = d_chroma + total_beats * FT_BINS;
    double *d_med = d_V + total_windows * FT_DIM;
    int64_t *d_tables = nullptr;
    if (hipMalloc((void **)&d_tables, sizeof(int64_t) * 2 * (size_t)(n_songs + 1)) != hipSuccess) {
        free(offs);
        set_error("ftm2d_shingles: device allocation failed");
        return ACOSS_ENOMEM;
    }
    int rc = ACOSS_OK;
    if (hipMemcpyAsync(d_tables, offs, sizeof(int64_t) * 2 * (size_t)(n_songs + 1), hipMemcpyHostToDevice, st) != hipSuccess) {
        rc = ACOSS_EIO;
        set_error("ftm2d_shingles: offset upload failed");
    }
    if (rc == ACOSS_OK && total_beats > 0) {
        hipLaunchKernelGGL(ftm2d_chrompwr_kernel, dim3((unsigned)ceil_div64(total_beats, 256)), dim3(256), 0, st,
                           btchroma + b_first * FT_BINS, total_beats, pwr, d_chroma);
        rc = launch_check("ftm2d_chrompwr_kernel");
    }
    if (rc == ACOSS_OK && total_windows > 0) {
        hipLaunchKernelGGL(ftm2d_window_kernel, dim3((unsigned)total_windows), dim3(256), 0, st, d_chroma, d_tables,
                           d_tables + (n_songs + 1), n_songs, C, d_V);
        rc = launch_check("ftm2d_window_kernel");
    }
    if (rc == ACOSS_OK) {
        hipLaunchKernelGGL(ftm2d_median_kernel, dim3(FT_DIM / 4, (unsigned)n_songs), dim3(256), 0, st, d_V, d_tables + (n_songs + 1), d_med);
        rc = launch_check("ftm2d_median_kernel");
    }
    if (rc == ACOSS_OK) {
        hipLaunchKernelGGL(ftm2d_normalize_kernel, dim3((unsigned)n_songs), dim3(256), 0, st, d_med, d_tables + (n_songs + 1), shingles);
        rc = launch_check("ftm2d_normalize_kernel");
    }
    (void)hipStreamSynchronize(st);      // the offset tables (host and device copies) must outlive the kernels
    (void)hipFree(d_tables);
    free(offs);
    return rc;
}

int acoss_ftm2d_pairs(const double *shingles, const int32_t *pairs, int K, double *sims, void *stream)
{
    if (!shingles || !pairs || !sims || K < 0) { set_error("ftm2d_pairs: bad argument"); return ACOSS_EINVAL; }
    if (K == 0) return ACOSS_OK;
    hipLaunchKernelGGL(ftm2d_pairs_kernel, dim3((unsigned)ceil_div(K, 4)), dim3(256), 0, (hipStream_t)stream, shingles, pairs, K, sims);
    return launch_check("ftm2d_pairs_kernel");
}

int acoss_ftm2d_gram(const double *shingles, int n, double *sims, void *stream)
{
    if (!shingles || !sims || n < 1) { set_error("ftm2d_gram: bad argument"); return ACOSS_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    double *norms = nullptr;
    if (hipMalloc((void **)&norms, sizeof(double) * (size_t)n) != hipSuccess) { set_error("ftm2d_gram: device allocation failed"); return ACOSS_ENOMEM; }
    hipLaunchKernelGGL(ftm2d_norms_kernel, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, st, shingles, n, norms);
    if (GM_DMA && (FT_DIM & 1) == 0 && ((uintptr_t)shingles & 15) == 0)
        hipLaunchKernelGGL(ftm2d_gram_dma_kernel, dim3((unsigned)ceil_div(n, GD_T), (unsigned)ceil_div(n, GD_T)), dim3(GD_THREADS), 0, st, shingles, norms, n, sims);
    else
    hipLaunchKernelGGL(ftm2d_gram_kernel, dim3((unsigned)ceil_div(n, GM_TJ), (unsigned)ceil_div(n, GM_T)), dim3(GM_THREADS), 0, st, shingles, norms, n, sims);
    const int rc = launch_check("ftm2d_gram_kernel");
    (void)hipStreamSynchronize(st);
    (void)hipFree(norms);
    return rc;
}

}  // extern "C"
