// snf_kernels.hip -- similarity network fusion of a song pair's block affinity matrices
// (benchmarking/SimilarityFusion.py; driven per pair by benchmarking/EarlySNF.py:41-90).
//
// Per pair (A: M frames, B: N frames, L = M + N) and per feature:
//   affinity  W = [[W(SSMA), W(CSM)], [W(CSM)^T, W(SSMB)]]  (get_WCSMSSM, :94-134): exponential kernels whose local
//             scale comes from the mean of the k smallest entries of each row / column (get_W :50-73, get_WCSM :76-92)
//   P         row-normalised W with the diagonal regularised to 1/2 (get_P, :136-157)
//   S         every row's K largest entries of W, L1-normalised (get_S, :159-180; kept dense here)
//   3 cross-diffusion steps  P_i <- reg(S_i . mean_{k != i} P_k . S_i^T)  (snf_ws, :207-277), as two L x L x L
//             float64 products on the matrix cores each -- S has 9.5 % non-zeros, but a gather-based sparse product
//             would read 188 rows of P per output row (6 GB per product through L2); the dense MFMA product is faster
//   result    the negated cross block of the mean of the P_i, in the layout of the pair's cross-recurrence matrix,
//             ready for the kNN mask and alignment kernels.
#include "common.h"
#include "wave_ops.h"
#include "gemm_f64.h"

namespace acoss {

struct SnfPair {            // one per pair, device table
    int64_t w_off;          // element offset of the pair's L x L matrices inside every matrix buffer: rows ld apart
    int64_t md_off;         // offset of its 2 L local-scale means
    int64_t c_off;          // its offset in a buffer of packed L x L matrices (debug outputs)
    int M, N, k1, k2, K, L;
    int ld;                 // L rounded up to even (16-byte rows: the DMA form of the products); the pad column holds zeros
};

struct SnfBlocks {          // the three distance matrices of one feature and their pair layouts
    const double *ssma, *ssmb, *csm;
    const acoss_pair_desc *da, *db, *dc;
    int win;
};

__device__ inline double snf_block_value(const SnfBlocks &f, int p, int M, int r, int c)
{
    // entry (r, c) of [[SSMA, CSM], [CSM^T, SSMB]] with the self-similarity diagonals forced to 0 (get_W :58)
    if (r < M) {
        if (c < M) return r == c ? 0.0 : f.ssma[f.da[p].crp_off + (int64_t)r * f.da[p].crp_pitch + c];
        return f.csm[f.dc[p].crp_off + (int64_t)r * f.dc[p].crp_pitch + (c - M)];
    }
    if (c < M) return f.csm[f.dc[p].crp_off + (int64_t)c * f.dc[p].crp_pitch + (r - M)];
    return r == c ? 0.0 : f.ssmb[f.db[p].crp_off + (int64_t)(r - M) * f.db[p].crp_pitch + (c - M)];
}

// mean of the k smallest of the n values a wave holds, 32 per lane (positions e*64 + lane): the k-th smallest by
// selection, then the sum of everything below it plus the threshold value for the remaining places
__device__ inline double wave_ksmallest_mean(const double (&x)[32], int n, int k, int lane)
{
    uint64_t key[32];
    int idx[32];
#pragma unroll
    for (int e = 0; e < 32; e++) {
        idx[e] = e * 64 + lane;
        key[e] = idx[e] < n ? f64_key(x[e]) : ~0ull;
    }
    const SelectResult r = wave_select_kth<32>(key, idx, n, k);
    const double tv = f64_from_key(r.thr_key);
    double s = 0.0;
    int below = 0;
#pragma unroll
    for (int e = 0; e < 32; e++) {
        const bool lt = key[e] < r.thr_key;
        s += lt ? x[e] : 0.0;
        below += lt;
    }
    for (int d = 32; d > 0; d >>= 1) {
        s += __shfl_down(s, d, 64);
        below += __shfl_down(below, d, 64);
    }
    s = __shfl(s, 0, 64);
    below = __shfl(below, 0, 64);
    return (s + (double)(k - below) * tv) / (double)k;
}

// local-scale means: vectors 0..M-1 rows of SSMA (k1 + 1 smallest incl. the zero diagonal, rescaled, :60-61),
// M..L-1 rows of SSMB (k2), L..L+M-1 rows of the CSM (k2 smallest, :87-88), L+M..2L-1 its columns (k1, :89-90)
__global__ __launch_bounds__(256) void snf_stats_kernel(SnfBlocks f, const SnfPair *__restrict__ pairs, int max_vec,
                                                        double *__restrict__ md)
{
    const int p = blockIdx.y;
    const SnfPair pr = pairs[p];
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (v >= 2 * pr.L) return;
    const int M = pr.M, N = pr.N;
    double x[32];
    int n, k;
    double scale = 1.0;
    if (v < pr.L) {             // a row of one of the self-similarity blocks
        const bool inA = v < M;
        n = inA ? M : N;
        const int kk = inA ? pr.k1 : pr.k2;
        k = kk + 1;
        scale = (double)(kk + 1) / (double)kk;
        const int r = v, c0 = inA ? 0 : M;
#pragma unroll
        for (int e = 0; e < 32; e++) x[e] = snf_block_value(f, p, M, r, c0 + min(e * 64 + lane, n - 1));
    } else if (v < pr.L + M) {  // a row of the cross block
        n = N;
        k = pr.k2;
        const int r = v - pr.L;
#pragma unroll
        for (int e = 0; e < 32; e++) x[e] = snf_block_value(f, p, M, r, M + min(e * 64 + lane, n - 1));
    } else {                    // a column of the cross block
        n = M;
        k = pr.k1;
        const int c = v - pr.L - M;
#pragma unroll
        for (int e = 0; e < 32; e++) x[e] = snf_block_value(f, p, M, min(e * 64 + lane, n - 1), M + c);
    }
    const double m = wave_ksmallest_mean(x, n, k, lane) * scale;
    if (lane == 0) md[pr.md_off + v] = m;
}

// W(r, c) = exp(-d^2 / (2 (Mu Eps)^2)), Eps = (md_r + md_c + d) / 3
__global__ __launch_bounds__(256) void snf_wfill_kernel(SnfBlocks f, const SnfPair *__restrict__ pairs, double Mu,
                                                        const double *__restrict__ md, double *__restrict__ W)
{
    const int p = blockIdx.z;
    const SnfPair pr = pairs[p];
    const int r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (r >= pr.L || c >= pr.ld) return;
    if (c >= pr.L) { W[pr.w_off + (int64_t)r * pr.ld + c] = 0.0; return; }      // the pad column
    const int M = pr.M, L = pr.L;
    const double *m = md + pr.md_off;
    const double d = snf_block_value(f, p, M, r, c);
    const bool self = (r < M) == (c < M);
    double mr, mc;
    if (self) { mr = m[r]; mc = m[c]; }                                 // get_W: MeanDist of both rows
    else if (r < M) { mr = m[L + r]; mc = m[L + M + (c - M)]; }         // get_WCSM: row mean + column mean
    else { mr = m[L + M + (r - M)]; mc = m[L + c]; }                    // the transposed block
    const double eps = (mr + mc + d) / 3.0;
    double den = 2.0 * ((Mu * eps) * (Mu * eps));
    if (self && den == 0.0) den = 1.0;                                  // :69-70 (get_W only)
    W[pr.w_off + (int64_t)r * pr.ld + c] = exp(-(d * d) / den);
}

// row sums without the diagonal (get_P :147-149, snf_ws :258-260), one wave per row, fixed order
__global__ __launch_bounds__(256) void snf_rowsum_kernel(const double *__restrict__ X, const SnfPair *__restrict__ pairs,
                                                         double *__restrict__ rowsum)
{
    const int p = blockIdx.y;
    const SnfPair pr = pairs[p];
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= pr.L) return;
    const double *row = X + pr.w_off + (int64_t)r * pr.ld;
    double s = 0.0;
    for (int c = lane; c < pr.L; c += 64) s += c == r ? 0.0 : row[c];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (lane == 0) rowsum[pr.md_off / 2 + r] = s;
}

// P = I / 2 + X_nodiag / (2 rowsum)  (:145-151, :255-262); in place is fine
__global__ __launch_bounds__(256) void snf_reg_kernel(const double *__restrict__ X, const SnfPair *__restrict__ pairs,
                                                      const double *__restrict__ rowsum, double *__restrict__ P)
{
    const int p = blockIdx.z;
    const SnfPair pr = pairs[p];
    const int r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (r >= pr.L || c >= pr.ld) return;
    double rs = rowsum[pr.md_off / 2 + r];
    if (rs == 0.0) rs = 1.0;
    const int64_t at = pr.w_off + (int64_t)r * pr.ld + c;
    P[at] = c >= pr.L ? 0.0 : (r == c ? 0.5 : 0.5 * X[at] / rs);
}

// S: the K largest entries of every row of W, divided by their sum (:170-180); ties cut lowest column first
__global__ __launch_bounds__(256) void snf_topk_kernel(const double *__restrict__ W, const SnfPair *__restrict__ pairs,
                                                       double *__restrict__ S)
{
    const int p = blockIdx.y;
    const SnfPair pr = pairs[p];
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= pr.L) return;
    const double *row = W + pr.w_off + (int64_t)r * pr.ld;
    double x[32];
    uint64_t key[32];
    int idx[32];
#pragma unroll
    for (int e = 0; e < 32; e++) {
        idx[e] = e * 64 + lane;
        x[e] = row[min(idx[e], pr.L - 1)];
        key[e] = idx[e] < pr.L ? f64_key(-x[e]) : ~0ull;
    }
    const SelectResult sel = wave_select_kth<32>(key, idx, pr.L, pr.K);
    double s = 0.0;
    bool on[32];
#pragma unroll
    for (int e = 0; e < 32; e++) {
        on[e] = (idx[e] < pr.L) & ((key[e] < sel.thr_key) | ((key[e] == sel.thr_key) & (idx[e] <= sel.cut)));
        s += on[e] ? x[e] : 0.0;
    }
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    s = __shfl(s, 0, 64);
    if (s == 0.0) s = 1.0;
    double *out = S + pr.w_off + (int64_t)r * pr.ld;
#pragma unroll
    for (int e = 0; e < 32; e++)
        if (idx[e] < pr.ld) out[idx[e]] = (idx[e] < pr.L && on[e]) ? x[e] / s : 0.0;
}

// the pair's L x L matrix out of its padded rows, into a buffer of packed matrices (debug outputs)
__global__ __launch_bounds__(256) void snf_unpad_kernel(const double *__restrict__ src, const SnfPair *__restrict__ pairs, double *__restrict__ dst)
{
    const SnfPair pr = pairs[blockIdx.z];
    const int r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (r >= pr.L || c >= pr.L) return;
    dst[pr.c_off + (int64_t)r * pr.L + c] = src[pr.w_off + (int64_t)r * pr.ld + c];
}

// out = mean of n_src matrices (:241-246), used when more than two features are fused
__global__ __launch_bounds__(256) void snf_mean_kernel(const double *const *__restrict__ src, int n_src, int64_t total,
                                                       double *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    double s = 0.0;
    for (int k = 0; k < n_src; k++) s += src[k][e];
    out[e] = s / (double)n_src;
}

// C = X . Y^T per pair (all L x L, row-major): one 128 x 128 tile per block (gemm_f64.h).  rows_m: only the tiles that hold rows
// below M -- the last product of the last iteration, whose other rows nobody reads (EarlySNF.py:84-85 takes [0:M, M:]).
__global__ __launch_bounds__(GM_THREADS) void snf_gemm_nt_kernel(const double *__restrict__ X, const double *__restrict__ Y,
                                                                 const SnfPair *__restrict__ pairs, double *__restrict__ C, int rows_m)
{
    __shared__ GemmSmem sm;
    const SnfPair pr = pairs[blockIdx.z];
    const int L = pr.L, ld = pr.ld;
    const int i0 = blockIdx.y * GM_T, j0 = blockIdx.x * GM_TJ;
    if (i0 >= (rows_m ? pr.M : L) || j0 >= L) return;
    const double *Xp = X + pr.w_off, *Yp = Y + pr.w_off;
    double *Cp = C + pr.w_off;
    // (the contraction runs over ld: the pad column of both operands holds zeros; the result's pad column is written as zero)
    gemm_nt_tile_f64_rows(sm, ld, Xp + (int64_t)i0 * ld, ld, L - i0, Yp + (int64_t)j0 * ld, ld, L - j0,
                          [&](const int i, const int j, const double v) {
                              if (i0 + i < L && j0 + j < ld) Cp[(int64_t)(i0 + i) * ld + j0 + j] = j0 + j < L ? v : 0.0;
                          });
}

#ifndef GM_DMA
#define GM_DMA 1
#endif
// the same product through gemm_nt_tile_f64_dma (gemm_f64.h): rows are ld = L rounded up to even doubles apart (16-byte rows)
__global__ __launch_bounds__(GD_THREADS) void snf_gemm_nt_dma_kernel(const double *__restrict__ X, const double *__restrict__ Y,
                                                                     const SnfPair *__restrict__ pairs, double *__restrict__ C, int rows_m)
{
    __shared__ GemmDmaSmem sm;
    const SnfPair pr = pairs[blockIdx.z];
    const int L = pr.L, ld = pr.ld;
    const int i0 = blockIdx.y * GD_T, j0 = blockIdx.x * GD_T;
    if (i0 >= (rows_m ? pr.M : L) || j0 >= L) return;
    const double *Xp = X + pr.w_off, *Yp = Y + pr.w_off;
    double *Cp = C + pr.w_off;
    gemm_nt_tile_f64_dma(sm, ld, Xp + (int64_t)i0 * ld, ld, L - i0, Yp + (int64_t)j0 * ld, ld, L - j0,
                         [&](const int i, const int j, const double v) {
                             if (i0 + i < L && j0 + j < ld) Cp[(int64_t)(i0 + i) * ld + j0 + j] = j0 + j < L ? v : 0.0;
                         });
}

// -(mean_f P_f)[0:M, M:] into the pair's cross-recurrence layout (EarlySNF.py:84-85)
__global__ __launch_bounds__(256) void snf_cross_kernel(const double *const *__restrict__ P, int n_feat,
                                                        const SnfPair *__restrict__ pairs,
                                                        const acoss_pair_desc *__restrict__ dout, double *__restrict__ out)
{
    const int p = blockIdx.z;
    const SnfPair pr = pairs[p];
    const int r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (r >= pr.M || c >= pr.N) return;
    double s = 0.0;
    for (int k = 0; k < n_feat; k++) s += P[k][pr.w_off + (int64_t)r * pr.ld + pr.M + c];      // fused_score, :182-189
    out[dout[p].crp_off + (int64_t)r * dout[p].crp_pitch + c] = -(s / (double)n_feat);
}

}  // namespace acoss

using namespace acoss;

extern "C" {

// host-side sizes: total elements of the L x L matrices of a batch
static int64_t snf_total_elems(const int32_t *M, const int32_t *N, int K)
{
    int64_t t = 0;
    for (int p = 0; p < K; p++) { const int64_t L = (int64_t)M[p] + N[p]; t += L * ((L + 1) & ~(int64_t)1); }      // rows ld apart
    return t;
}

size_t acoss_snf_scratch_bytes(const int32_t *M, const int32_t *N, int K, int n_feat)
{
    // per feature: W/P, S, P'   + two temporaries, + the local-scale means / row sums, + tables
    int64_t tl = 0;
    for (int p = 0; p < K; p++) tl += (int64_t)M[p] + N[p];
    const int64_t tot = snf_total_elems(M, N, K);
    return sizeof(double) * (size_t)((3 * (int64_t)n_feat + 2) * tot + 3 * tl) + sizeof(SnfPair) * (size_t)K + 8 * (size_t)n_feat * 4 + 1024;
}

int acoss_snf_cross_batch(const acoss_snf_feature *feats, int n_feat, int K, const int32_t *M, const int32_t *N,
                          double kappa, double mu, int niters, void *scratch, size_t scratch_bytes,
                          const acoss_pair_desc *dout, double *cross_out, double *debug_W, double *debug_fused,
                          void *stream)
{
    if (!feats || n_feat < 2 || n_feat > 8 || K < 0 || !M || !N || !scratch || !dout || !cross_out || niters < 0 || kappa <= 0.0) {
        set_error("snf_cross_batch: bad argument (2..8 features)");
        return ACOSS_EINVAL;
    }
    if (K == 0) return ACOSS_OK;
    if (scratch_bytes < acoss_snf_scratch_bytes(M, N, K, n_feat)) { set_error("snf_cross_batch: scratch too small"); return ACOSS_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    // per-pair table (EarlySNF.py:51: K = int(kappa (M + N)); SimilarityFusion.py:127-128: k1, k2)
    SnfPair *tab = (SnfPair *)malloc(sizeof(SnfPair) * (size_t)K);
    if (!tab) { set_error("snf_cross_batch: out of host memory"); return ACOSS_ENOMEM; }
    int64_t woff = 0, mdoff = 0, coff = 0;
    int maxL = 0;
    for (int p = 0; p < K; p++) {
        SnfPair &t = tab[p];
        t.M = M[p]; t.N = N[p]; t.L = M[p] + N[p];
        t.K = (int)(kappa * (double)t.L);
        t.k1 = (int)((double)t.K * (double)t.M / (double)t.L);
        t.k2 = t.K - t.k1;
        if (t.M < 1 || t.N < 1 || t.L > 2048 || t.k1 < 1 || t.k2 < 1 || t.K >= t.L) {
            set_error("snf_cross_batch: pair %d (M %d, N %d, K %d): needs M + N <= 2048 and at least one neighbour per block", p, t.M, t.N, t.K);
            free(tab);
            return t.L > 2048 ? ACOSS_ENOTSUP : ACOSS_EINVAL;
        }
        t.ld = (t.L + 1) & ~1;
        t.w_off = woff; t.md_off = mdoff; t.c_off = coff;
        woff += (int64_t)t.L * t.ld;
        coff += (int64_t)t.L * t.L;
        mdoff += 2 * (int64_t)t.L;
        maxL = t.L > maxL ? t.L : maxL;
    }
    const int64_t tot = woff;
    // scratch layout
    double *base = (double *)scratch;
    double *Pm[8], *Sm[8], *Pn[8];
    for (int f = 0; f < n_feat; f++) { Pm[f] = base; base += tot; Sm[f] = base; base += tot; Pn[f] = base; base += tot; }
    double *Am = base; base += tot;
    double *Xm = base; base += tot;
    double *md = base; base += mdoff;                   // 2 L per pair
    double *rowsum = base; base += mdoff / 2;           // L per pair (indexed by md_off / 2)
    SnfPair *d_tab = (SnfPair *)(((uintptr_t)base + 63) & ~(uintptr_t)63);
    const double **d_ptrs = (const double **)(d_tab + K);
    int rc = ACOSS_OK;
    if (hipMemcpyAsync(d_tab, tab, sizeof(SnfPair) * (size_t)K, hipMemcpyHostToDevice, st) != hipSuccess) rc = ACOSS_EIO;
    const dim3 g_el((unsigned)ceil_div(maxL + 1, 256), (unsigned)maxL, (unsigned)K), g_row((unsigned)ceil_div(maxL, 4), (unsigned)K);
    const dim3 g_mm((unsigned)ceil_div(maxL, GM_TJ), (unsigned)ceil_div(maxL, GM_T), (unsigned)K);
    const dim3 g_dma((unsigned)ceil_div(maxL, GD_T), (unsigned)ceil_div(maxL, GD_T), (unsigned)K);
    const bool use_dma = GM_DMA != 0 && ((uintptr_t)scratch & 15) == 0;          // (every row starts on 16 bytes: ld and every w_off are even)
    for (int f = 0; f < n_feat && rc == ACOSS_OK; f++) {
        const SnfBlocks b{feats[f].ssma, feats[f].ssmb, feats[f].csm, feats[f].da, feats[f].db, feats[f].dc, feats[f].win};
        if (!b.ssma || !b.ssmb || !b.csm || !b.da || !b.db || !b.dc) { set_error("snf_cross_batch: feature %d has a null pointer", f); rc = ACOSS_EINVAL; break; }
        hipLaunchKernelGGL(snf_stats_kernel, dim3((unsigned)ceil_div(2 * maxL, 4), (unsigned)K), dim3(256), 0, st, b, d_tab, 2 * maxL, md);
        hipLaunchKernelGGL(snf_wfill_kernel, g_el, dim3(256), 0, st, b, d_tab, mu, md, Pm[f]);            // W lives in the P buffer
        if ((rc = launch_check("snf affinity kernels")) != ACOSS_OK) break;
        if (debug_W) hipLaunchKernelGGL(snf_unpad_kernel, g_el, dim3(256), 0, st, Pm[f], d_tab, debug_W + (int64_t)f * coff);
        hipLaunchKernelGGL(snf_topk_kernel, g_row, dim3(256), 0, st, Pm[f], d_tab, Sm[f]);
        hipLaunchKernelGGL(snf_rowsum_kernel, g_row, dim3(256), 0, st, Pm[f], d_tab, rowsum);
        hipLaunchKernelGGL(snf_reg_kernel, g_el, dim3(256), 0, st, Pm[f], d_tab, rowsum, Pm[f]);          // P = get_P(W, reg_diag)
        if (rc == ACOSS_OK) rc = launch_check("snf P / S kernels");
    }
    // cross diffusion (:238-270).  Iteration 0 reads the initial P of every other feature; from iteration 1 on the
    // reference's Pts and nextPts are one list, so feature i reads the already-updated features k < i.
    for (int it = 0; it < niters && rc == ACOSS_OK; it++) {
        for (int i = 0; i < n_feat && rc == ACOSS_OK; i++) {
            const double *src;
            if (n_feat == 2) {
                src = Pm[1 - i];
                if (it == 0 && i == 1) src = Pm[0];     // iteration 0 writes into Pn, so Pm[0] still holds the initial P
            } else {
                const double *list[8];
                int n = 0;
                for (int k = 0; k < n_feat; k++)
                    if (k != i) list[n++] = (it == 0) ? Pm[k] : Pm[k];
                if (hipMemcpyAsync((void *)d_ptrs, list, sizeof(double *) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) { rc = ACOSS_EIO; break; }
                (void)hipStreamSynchronize(st);       // `list` is a stack array
                hipLaunchKernelGGL(snf_mean_kernel, dim3((unsigned)ceil_div64(tot, 256)), dim3(256), 0, st, d_ptrs, n, tot, Xm);
                src = Xm;
            }
            if (use_dma) hipLaunchKernelGGL(snf_gemm_nt_dma_kernel, g_dma, dim3(GD_THREADS), 0, st, Sm[i], src, d_tab, Am, 0);
            else
            hipLaunchKernelGGL(snf_gemm_nt_kernel, g_mm, dim3(GM_THREADS), 0, st, Sm[i], src, d_tab, Am, 0);    // A = S . P^T   (:251)
            double *dst = (it == 0) ? Pn[i] : Pm[i];
            // the very last product: only rows 0 .. M - 1 are ever read (by snf_cross_kernel) -- unless the caller wants the whole
            // fused matrix (debug_fused), or further features of this iteration read it (they read the features before them)
            const int rows_m = (it == niters - 1 && i == n_feat - 1 && debug_fused == nullptr) ? 1 : 0;
            if (use_dma) hipLaunchKernelGGL(snf_gemm_nt_dma_kernel, g_dma, dim3(GD_THREADS), 0, st, Sm[i], Am, d_tab, dst, rows_m);
            else
            hipLaunchKernelGGL(snf_gemm_nt_kernel, g_mm, dim3(GM_THREADS), 0, st, Sm[i], Am, d_tab, dst, rows_m);       // S . A^T       (:252)
            hipLaunchKernelGGL(snf_rowsum_kernel, g_row, dim3(256), 0, st, dst, d_tab, rowsum);
            hipLaunchKernelGGL(snf_reg_kernel, g_el, dim3(256), 0, st, dst, d_tab, rowsum, dst);          // :254-262
            rc = launch_check("snf diffusion kernels");
        }
        if (it == 0 && rc == ACOSS_OK)
            for (int f = 0; f < n_feat; f++) { double *t = Pm[f]; Pm[f] = Pn[f]; Pn[f] = t; }          // Pts = nextPts (:270)
    }
    if (rc == ACOSS_OK) {
        const double *list[8];
        for (int f = 0; f < n_feat; f++) list[f] = Pm[f];
        if (hipMemcpyAsync((void *)d_ptrs, list, sizeof(double *) * (size_t)n_feat, hipMemcpyHostToDevice, st) != hipSuccess) rc = ACOSS_EIO;
        (void)hipStreamSynchronize(st);
        if (rc == ACOSS_OK) {
            int maxM = 0, maxN = 0;
            for (int p = 0; p < K; p++) { maxM = M[p] > maxM ? M[p] : maxM; maxN = N[p] > maxN ? N[p] : maxN; }
            hipLaunchKernelGGL(snf_cross_kernel, dim3((unsigned)ceil_div(maxN, 256), (unsigned)maxM, (unsigned)K), dim3(256), 0, st,
                               d_ptrs, n_feat, d_tab, dout, cross_out);
            rc = launch_check("snf_cross_kernel");
            if (rc == ACOSS_OK && debug_fused) {
                hipLaunchKernelGGL(snf_mean_kernel, dim3((unsigned)ceil_div64(tot, 256)), dim3(256), 0, st, d_ptrs, n_feat, tot, Xm);
                hipLaunchKernelGGL(snf_unpad_kernel, g_el, dim3(256), 0, st, Xm, d_tab, debug_fused);
                rc = launch_check("snf_mean_kernel");
            }
        }
    }
    (void)hipStreamSynchronize(st);       // the host table must outlive its upload
    free(tab);
    return rc;
}

}  // extern "C"
