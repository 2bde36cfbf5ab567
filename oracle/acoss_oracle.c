/*
 * acoss_oracle.c -- CPU oracle (plain C99) for the acoss pairwise scoring hot path.
 *
 * TEST INFRASTRUCTURE ONLY: see acoss_oracle.h for who may use this and for the parity
 * status (pinned against tests/golden/, generated from the reference itself).
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off so no FMA contraction changes the
 * rounding the reference's numpy code performs).
 */
#include "acoss_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* numpy summation order                                                                 */
/* ------------------------------------------------------------------------------------ */

/* np.sum over a contiguous float64 run = numpy's pairwise sum: < 8 elements sequential from
 * 0.0; <= 128 elements eight running accumulators combined as a balanced tree, tail added
 * sequentially; larger runs split in halves (multiple of 8).  Checked bit-for-bit against
 * np.sum in tests/test_oracle_golden.py. */
double orc_np_sum(const double *a, long n)
{
    if (n < 8) {
        double acc = 0.0;
        for (long i = 0; i < n; i++) acc += a[i];
        return acc;
    }
    if (n <= 128) {
        double r[8];
        long i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double acc = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) acc += a[i];
        return acc;
    }
    long half = n / 2;
    half -= half % 8;
    return orc_np_sum(a, half) + orc_np_sum(a + half, n - half);
}

static float np_sum_f32(const float *a, long n)
{
    /* same blocking as the float64 loop, in float32 */
    if (n < 8) {
        float acc = 0.0f;
        for (long i = 0; i < n; i++) acc += a[i];
        return acc;
    }
    if (n <= 128) {
        float r[8];
        long i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        float acc = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) acc += a[i];
        return acc;
    }
    long half = n / 2;
    half -= half % 8;
    return np_sum_f32(a, half) + np_sum_f32(a + half, n - half);
}

/* ------------------------------------------------------------------------------------ */
/* Serra09.py:24-28 global_chroma                                                        */
/* ------------------------------------------------------------------------------------ */
int orc_global_chroma(const double *chroma, long nframes, int nbins, double *out)
{
    if (nbins != 12 && nbins != 24 && nbins != 36) return -1; /* Serra09.py:26-27 */
    /* chroma.sum(axis=0) on a C-contiguous (n, nbins) array accumulates frame by frame */
    for (int b = 0; b < nbins; b++) out[b] = 0.0;
    for (long f = 0; f < nframes; f++)
        for (int b = 0; b < nbins; b++) out[b] += chroma[f * nbins + b];
    double top = out[0];
    for (int b = 1; b < nbins; b++)
        if (out[b] > top) top = out[b];
    for (int b = 0; b < nbins; b++) out[b] = out[b] / top;
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* CRPUtils.py:109-136 get_oti                                                           */
/* ------------------------------------------------------------------------------------ */
int orc_get_oti(const double *c1, const double *c2, int nbins)
{
    double prod[64];
    double *buf = prod;
    if (nbins > 64) buf = (double *)malloc(sizeof(double) * (size_t)nbins);
    int best = 0;
    double best_score = 0.0;
    for (int s = 0; s < nbins; s++) {
        /* np.roll(C1, s)[b] == C1[(b - s) mod nbins]  (CRPUtils.py:130) */
        for (int b = 0; b < nbins; b++) {
            int src = b - s;
            if (src < 0) src += nbins;
            buf[b] = c1[src] * c2[b];
        }
        double score = orc_np_sum(buf, nbins);
        if (s == 0 || score > best_score) { /* np.argmax: first maximum wins */
            best_score = score;
            best = s;
        }
    }
    if (buf != prod) free(buf);
    return best;
}

/* ------------------------------------------------------------------------------------ */
/* CRPUtils.py:67-84 get_csm                                                             */
/* ------------------------------------------------------------------------------------ */
void orc_csm_f64(const double *X, long M, const double *Y, long N, int d, int shift,
                 double *out)
{
    double *Xr = (double *)malloc(sizeof(double) * (size_t)M * d);
    double *xx = (double *)malloc(sizeof(double) * (size_t)M);
    double *yy = (double *)malloc(sizeof(double) * (size_t)N);
    double *sq = (double *)malloc(sizeof(double) * (size_t)d);
    shift = ((shift % d) + d) % d;
    for (long i = 0; i < M; i++)
        for (int b = 0; b < d; b++) Xr[i * d + (b + shift) % d] = X[i * d + b];
    for (long i = 0; i < M; i++) {
        for (int b = 0; b < d; b++) sq[b] = Xr[i * d + b] * Xr[i * d + b];
        xx[i] = orc_np_sum(sq, d); /* np.sum(X**2, 1)  CRPUtils.py:82 */
    }
    for (long j = 0; j < N; j++) {
        for (int b = 0; b < d; b++) sq[b] = Y[j * d + b] * Y[j * d + b];
        yy[j] = orc_np_sum(sq, d);
    }
    for (long i = 0; i < M; i++) {
        const double *x = Xr + i * d;
        for (long j = 0; j < N; j++) {
            const double *y = Y + j * d;
            double dot = 0.0; /* X.dot(Y.T): BLAS order is not knowable; sequential here */
            for (int b = 0; b < d; b++) dot += x[b] * y[b];
            double c = (xx[i] + yy[j]) - 2.0 * dot;
            if (c < 0.0) c = 0.0; /* CRPUtils.py:83 */
            out[i * N + j] = sqrt(c);
        }
    }
    free(Xr); free(xx); free(yy); free(sq);
}

void orc_csm_f32(const float *X, long M, const float *Y, long N, int d, int shift,
                 float *out)
{
    float *Xr = (float *)malloc(sizeof(float) * (size_t)M * d);
    float *xx = (float *)malloc(sizeof(float) * (size_t)M);
    float *yy = (float *)malloc(sizeof(float) * (size_t)N);
    float *sq = (float *)malloc(sizeof(float) * (size_t)d);
    shift = ((shift % d) + d) % d;
    for (long i = 0; i < M; i++)
        for (int b = 0; b < d; b++) Xr[i * d + (b + shift) % d] = X[i * d + b];
    for (long i = 0; i < M; i++) {
        for (int b = 0; b < d; b++) sq[b] = Xr[i * d + b] * Xr[i * d + b];
        xx[i] = np_sum_f32(sq, d);
    }
    for (long j = 0; j < N; j++) {
        for (int b = 0; b < d; b++) sq[b] = Y[j * d + b] * Y[j * d + b];
        yy[j] = np_sum_f32(sq, d);
    }
    for (long i = 0; i < M; i++) {
        const float *x = Xr + i * d;
        for (long j = 0; j < N; j++) {
            const float *y = Y + j * d;
            float dot = 0.0f;
            for (int b = 0; b < d; b++) dot += x[b] * y[b];
            float c = (xx[i] + yy[j]) - 2.0f * dot;
            if (c < 0.0f) c = 0.0f;
            out[i * N + j] = sqrtf(c);
        }
    }
    free(Xr); free(xx); free(yy); free(sq);
}

/* ------------------------------------------------------------------------------------ */
/* CRPUtils.py:24-45 sliding_csm                                                         */
/* ------------------------------------------------------------------------------------ */

/* One diagonal: sq[0..len) are the squared entries (already float64).  cum = cumsum([0]+sq)
 * (CRPUtils.py:41-42), S = sqrt(cum[win:] - cum[:-win]) (:43-44). */
static void sliding_one_diag(const double *sq, long len, int win, double *cum, double *S,
                             long r0, long c0, long N)
{
    cum[0] = 0.0;
    for (long t = 0; t < len; t++) cum[t + 1] = cum[t] + sq[t];
    for (long t = 0; t + win <= len; t++)
        S[(r0 + t) * N + (c0 + t)] = sqrt(cum[t + win] - cum[t]);
}

int orc_sliding_csm_f64(const double *D, long M0, long N0, int win, double *S)
{
    long M = M0 - win + 1, N = N0 - win + 1;
    if (M < 1 || N < 1 || win < 1) return -1;
    long cap = (M0 < N0 ? M0 : N0) + 1;
    double *sq = (double *)malloc(sizeof(double) * (size_t)cap);
    double *cum = (double *)malloc(sizeof(double) * (size_t)(cap + 1));
    for (long off = -(M - 1); off <= N - 1; off++) { /* CRPUtils.py:39 */
        long r0 = off < 0 ? -off : 0, c0 = off < 0 ? 0 : off;
        long len = M0 - r0 < N0 - c0 ? M0 - r0 : N0 - c0; /* np.diag(D, off) */
        for (long t = 0; t < len; t++) {
            double v = D[(r0 + t) * N0 + (c0 + t)];
            sq[t] = v * v; /* CRPUtils.py:40 */
        }
        sliding_one_diag(sq, len, win, cum, S, r0, c0, N);
    }
    free(sq); free(cum);
    return 0;
}

int orc_sliding_csm_f32(const float *D, long M0, long N0, int win, double *S)
{
    long M = M0 - win + 1, N = N0 - win + 1;
    if (M < 1 || N < 1 || win < 1) return -1;
    long cap = (M0 < N0 ? M0 : N0) + 1;
    double *sq = (double *)malloc(sizeof(double) * (size_t)cap);
    double *cum = (double *)malloc(sizeof(double) * (size_t)(cap + 1));
    for (long off = -(M - 1); off <= N - 1; off++) {
        long r0 = off < 0 ? -off : 0, c0 = off < 0 ? 0 : off;
        long len = M0 - r0 < N0 - c0 ? M0 - r0 : N0 - c0;
        for (long t = 0; t < len; t++) {
            float v = D[(r0 + t) * N0 + (c0 + t)];
            float v2 = v * v; /* squared in float32 (CRPUtils.py:40) ... */
            sq[t] = (double)v2; /* ... promoted by .tolist() (:41) */
        }
        sliding_one_diag(sq, len, win, cum, S, r0, c0, N);
    }
    free(sq); free(cum);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* CRPUtils.py:169-219 csm_to_binary / csm_to_binary_mutual                              */
/* ------------------------------------------------------------------------------------ */
long orc_nneighbs(double kappa, long ncols)
{
    if (kappa == 0.0) return ncols;                 /* CRPUtils.py:188-189: all ones */
    if (kappa < 1.0) return (long)rint(kappa * (double)ncols); /* :190-191 np.round = half-even */
    return (long)kappa;                             /* :192-193 */
}

/* k-th smallest (1-based) of v[0..n) by Hoare quickselect on a scratch copy */
static double kth_smallest(double *w, long n, long k)
{
    long lo = 0, hi = n - 1, target = k - 1;
    while (lo < hi) {
        double pivot = w[lo + (hi - lo) / 2];
        long a = lo, b = hi;
        while (a <= b) {
            while (w[a] < pivot) a++;
            while (w[b] > pivot) b--;
            if (a <= b) {
                double t = w[a]; w[a] = w[b]; w[b] = t;
                a++; b--;
            }
        }
        if (target <= b) hi = b;
        else if (target >= a) lo = a;
        else break;
    }
    return w[target];
}

/* ones at the k smallest of a strided vector; ties: value, then index */
static void mark_k_smallest(const double *v, long n, long stride, long k, double *scratch,
                            uint8_t *out, long ostride)
{
    if (k <= 0) { for (long j = 0; j < n; j++) out[j * ostride] = 0; return; }
    if (k >= n) { for (long j = 0; j < n; j++) out[j * ostride] = 1; return; }
    for (long j = 0; j < n; j++) scratch[j] = v[j * stride];
    double thr = kth_smallest(scratch, n, k);
    long below = 0;
    for (long j = 0; j < n; j++) below += (v[j * stride] < thr);
    long at_thr = k - below; /* how many entries equal to thr are taken, lowest index first */
    for (long j = 0; j < n; j++) {
        double x = v[j * stride];
        uint8_t bit = 0;
        if (x < thr) bit = 1;
        else if (x == thr && at_thr > 0) { bit = 1; at_thr--; }
        out[j * ostride] = bit;
    }
}

void orc_csm_to_binary(const double *D, long M, long N, double kappa, uint8_t *B)
{
    long k = orc_nneighbs(kappa, N);
    double *scratch = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    for (long i = 0; i < M; i++)
        mark_k_smallest(D + i * N, N, 1, k, scratch, B + i * N, 1);
    free(scratch);
}

void orc_csm_to_binary_mutual(const double *D, long M, long N, double kappa, uint8_t *B)
{
    /* csm_to_binary(D, kappa) * csm_to_binary(D.T, kappa).T   (CRPUtils.py:219):
     * the transposed call sees ncols = M, so its neighbour count is from M. */
    long kc = orc_nneighbs(kappa, M);
    uint8_t *Bc = (uint8_t *)malloc((size_t)(M * N > 0 ? M * N : 1));
    double *scratch = (double *)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
    orc_csm_to_binary(D, M, N, kappa, B);
    for (long j = 0; j < N; j++)
        mark_k_smallest(D + j, M, N, kc, scratch, Bc + j, N);
    for (long t = 0; t < M * N; t++) B[t] = (uint8_t)(B[t] * Bc[t]);
    free(Bc); free(scratch);
}

/* ------------------------------------------------------------------------------------ */
/* SequenceAlignment.c alignment recurrences                                             */
/* ------------------------------------------------------------------------------------ */
static inline float fmax2(float a, float b) { return b > a ? b : a; }

/* SequenceAlignment.c:104-111 gammaState: onset for a recurrence point, extension otherwise
 * (both 0.5 in the reference). */
static const float GAMMA_ONSET = 0.5f, GAMMA_EXT = 0.5f;
static inline float gap_cost(unsigned char s) { return s == 1 ? GAMMA_ONSET : GAMMA_EXT; }

/* SequenceAlignment.c:113-143 */
float orc_qmax(const unsigned char *S, float *D, int M, int N)
{
    if (M < 3 || N < 3) return 0.0f; /* :117-119 */
    float best = 0.0f;
    for (int i = 2; i < M; i++) {
        const unsigned char *s_cur = S + (size_t)i * N, *s_p1 = s_cur - N, *s_p2 = s_p1 - N;
        float *d_cur = D + (size_t)i * N;
        const float *d_p1 = d_cur - N, *d_p2 = d_p1 - N;
        for (int j = 2; j < N; j++) {
            float diag = d_p1[j - 1], up2 = d_p2[j - 1], left2 = d_p1[j - 2];
            float v;
            if (s_cur[j] == 1) { /* :124-129 */
                v = fmax2(fmax2(diag, up2), left2) + 1.0f;
            } else {             /* :130-136 */
                diag -= gap_cost(s_p1[j - 1]);
                up2 -= gap_cost(s_p2[j - 1]);
                left2 -= gap_cost(s_p1[j - 2]);
                v = fmax2(fmax2(fmax2(diag, up2), left2), 0.0f);
            }
            d_cur[j] = v;
            if (v > best) best = v;
        }
    }
    return best;
}

/* SequenceAlignment.c:147-180 */
float orc_dmax(const unsigned char *S, float *D, int M, int N)
{
    if (M < 4 || N < 4) return 0.0f; /* :151-153 */
    float best = 0.0f;
    for (int i = 3; i < M; i++) {
        const unsigned char *s0 = S + (size_t)i * N, *s1 = s0 - N, *s2 = s1 - N, *s3 = s2 - N;
        float *d0 = D + (size_t)i * N;
        const float *d1 = d0 - N, *d2 = d1 - N, *d3 = d2 - N;
        for (int j = 3; j < N; j++) {
            /* the five predecessors with the skipped cells' S values added (:159-162) */
            float p1 = d1[j - 1];
            float p2 = d2[j - 1] + (float)s1[j];
            float p3 = d1[j - 2] + (float)s0[j - 1];
            float p4 = d3[j - 1] + (float)s2[j] + (float)s1[j];
            float p5 = d1[j - 3] + (float)s0[j - 2] + (float)s0[j - 1];
            float v;
            if (s0[j] == 1) { /* :158-164 */
                v = fmax2(fmax2(fmax2(fmax2(p1, p2), p3), p4), p5) + 1.0f;
            } else {          /* :165-173 */
                p1 -= gap_cost(s1[j - 1]);
                p2 -= gap_cost(s2[j - 1]);
                p3 -= gap_cost(s1[j - 2]);
                p4 -= gap_cost(s3[j - 1]);
                p5 -= gap_cost(s1[j - 3]);
                v = fmax2(fmax2(fmax2(fmax2(fmax2(p1, p2), p3), p4), p5), 0.0f);
            }
            d0[j] = v;
            if (v > best) best = v;
        }
    }
    return best;
}

/* SequenceAlignment.c:43-63 Delta / Match */
static inline float sw_gap(unsigned char prev, unsigned char cur)
{
    if (cur > 0) return 0.0f;
    if (prev > 0) return -0.5f; /* gap opening */
    return -0.7f;               /* gap extension */
}

/* SequenceAlignment.c:73-99.  S is (N,M) row-major, D is (N+1,M+1) row-major. */
float orc_swc(const unsigned char *S, float *D, int N, int M)
{
    int R = N + 1, C = M + 1; /* :77 */
    if (R < 4 || C < 4) return 0.0f;
    float best = 0.0f;
    for (int i = 3; i < R; i++) {
        for (int j = 3; j < C; j++) {
            unsigned char cur = S[(size_t)(i - 1) * M + (j - 1)];
            float ms = cur == 0 ? -1.0f : 1.0f;
            float a = D[(size_t)(i - 1) * C + (j - 1)] + ms + sw_gap(S[(size_t)(i - 2) * M + (j - 2)], cur);
            float b = D[(size_t)(i - 2) * C + (j - 1)] + ms + sw_gap(S[(size_t)(i - 3) * M + (j - 2)], cur);
            float c = D[(size_t)(i - 1) * C + (j - 2)] + ms + sw_gap(S[(size_t)(i - 2) * M + (j - 3)], cur);
            float v = fmax2(fmax2(fmax2(a, b), c), 0.0f);
            D[(size_t)i * C + j] = v;
            if (v > best) best = v;
        }
    }
    return best;
}

/* ------------------------------------------------------------------------------------ */
/* Serra09.py:166-175 one pair's chain                                                   */
/* ------------------------------------------------------------------------------------ */
/* scratch for one pair of at most (max_n x max_n) frames; reused across pairs by one thread */
typedef struct {
    double *csm, *S;
    uint8_t *B;
    float *D;
} pair_ws;

static int ws_alloc(pair_ws *w, long max_nx, long max_ny)
{
    size_t cells = (size_t)max_nx * (size_t)max_ny;
    w->csm = (double *)malloc(sizeof(double) * cells);
    w->S = (double *)malloc(sizeof(double) * cells);
    w->B = (uint8_t *)malloc(cells);
    w->D = (float *)malloc(sizeof(float) * cells);
    return (w->csm && w->S && w->B && w->D) ? 0 : -1;
}

static void ws_free(pair_ws *w)
{
    free(w->csm); free(w->S); free(w->B); free(w->D);
}

static int serra09_pair_ws(const double *Xi, const double *gi, long ni,
                           const double *Xj, const double *gj, long nj,
                           int d, int m, double kappa, int do_oti, pair_ws *w,
                           double *qmax_out, double *dmax_out)
{
    long M = ni - m + 1, N = nj - m + 1;
    if (M < 1 || N < 1) return -1;
    int shift = do_oti ? orc_get_oti(gi, gj, d) : 0;        /* Serra09.py:166 */
    orc_csm_f64(Xi, ni, Xj, nj, d, shift, w->csm);          /* :167-169 */
    orc_sliding_csm_f64(w->csm, ni, nj, m, w->S);           /* :170 */
    orc_csm_to_binary_mutual(w->S, M, N, kappa, w->B);      /* :171 */
    memset(w->D, 0, sizeof(float) * (size_t)M * (size_t)N); /* :173 */
    float q = orc_qmax(w->B, w->D, (int)M, (int)N);         /* :174 */
    if (qmax_out) *qmax_out = (double)q / (double)(M + N);
    if (dmax_out) {                                         /* skipped when only chroma_qmax is wanted */
        float dm = orc_dmax(w->B, w->D, (int)M, (int)N);    /* :175 -- D not re-zeroed */
        *dmax_out = (double)dm / (double)(M + N);
    }
    return 0;
}

int orc_serra09_pair(const double *Xi, const double *gi, long ni,
                     const double *Xj, const double *gj, long nj,
                     int d, int m, double kappa, int do_oti,
                     double *qmax_out, double *dmax_out)
{
    pair_ws w;
    if (ni < m || nj < m) return -1;
    if (ws_alloc(&w, ni, nj) != 0) { ws_free(&w); return -1; }
    int rc = serra09_pair_ws(Xi, gi, ni, Xj, gj, nj, d, m, kappa, do_oti, &w, qmax_out, dmax_out);
    ws_free(&w);
    return rc;
}

int orc_serra09_pairs(const double *feats, const int64_t *frame_off, const double *gchroma,
                      const int32_t *pairs, long K, int d, int m, double kappa, int do_oti,
                      int nthreads, double *qmax_out, double *dmax_out)
{
    int used = 1;
    long max_n = 1;
    for (long p = 0; p < K; p++) {
        for (int s = 0; s < 2; s++) {
            long n = frame_off[pairs[2 * p + s] + 1] - frame_off[pairs[2 * p + s]];
            if (n > max_n) max_n = n;
        }
    }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel
#endif
    {
        pair_ws w;   /* one scratch per thread, reused: no allocator traffic inside the timed loop */
        int ok = ws_alloc(&w, max_n, max_n) == 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long p = 0; p < K; p++) {
            int a = pairs[2 * p], b = pairs[2 * p + 1];
            double q = 0.0, dm = 0.0;
            if (ok)
                serra09_pair_ws(feats + frame_off[a] * d, gchroma + (long)a * d,
                                frame_off[a + 1] - frame_off[a],
                                feats + frame_off[b] * d, gchroma + (long)b * d,
                                frame_off[b + 1] - frame_off[b],
                                d, m, kappa, do_oti, &w, &q, dmax_out ? &dm : NULL);
            if (qmax_out) qmax_out[p] = q;
            if (dmax_out) dmax_out[p] = dm;
        }
        ws_free(&w);
    }
    return used;
}
