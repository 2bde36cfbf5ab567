/*
 * asan_driver.c -- runs the CPU oracle (acoss_oracle.c) under -fsanitize=address,undefined on the reference's own
 * fixtures.  TEST INFRASTRUCTURE ONLY (oracle/acoss_oracle.h).  The reference compiles its alignment kernel with
 * boundscheck(False) / wraparound(False) and -Ofast (benchmarking/pySeqAlign.pyx:5-6, setup.py:45) and has no test of
 * its own for out-of-range reads; this driver gives every buffer its exact size on the heap, so a read or write one
 * element past a matrix is a sanitizer report, and checks every result against the fixture.
 *
 * Input (written by tests/test_oracle_sanitize.py from tests/golden/*.npz): a stream of records
 *   "DP  " int32 M N; uint8 S[M*N]; float32 Dq[M*N] Dd_reused[M*N] Dd_fresh[M*N] Dsw[(M+1)(N+1)]; float64 scores[4]
 *   "STG " int32 nx ny d m oti; float64 kappa; float64 X[nx*d] Y[ny*d] gX[d] gY[d] CSM[nx*ny] S[M*N]; uint8 B1[M*N] B[M*N]
 *   "END "
 * Exit status 0 = every record reproduced, no sanitizer report.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "acoss_oracle.h"

static void *take(FILE *f, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) { fprintf(stderr, "asan_driver: short read\n"); exit(2); }
    return p;
}

static int fail(const char *what, int rec)
{
    fprintf(stderr, "asan_driver: record %d: %s differs from the fixture\n", rec, what);
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s cases.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int rec = 0, bad = 0, n_dp = 0, n_stage = 0;
    for (;; rec++) {
        char tag[4];
        if (fread(tag, 1, 4, f) != 4) { fprintf(stderr, "asan_driver: no END record\n"); return 2; }
        if (!memcmp(tag, "END ", 4)) break;
        if (!memcmp(tag, "DP  ", 4)) {
            int32_t dim[2];
            if (fread(dim, 4, 2, f) != 2) return 2;
            const int M = dim[0], N = dim[1];
            const size_t mn = (size_t)M * N, sw = (size_t)(M + 1) * (N + 1);
            unsigned char *S = take(f, mn);
            float *Dq = take(f, 4 * mn), *Ddr = take(f, 4 * mn), *Ddf = take(f, 4 * mn), *Dsw = take(f, 4 * sw);
            double *sc = take(f, 8 * 4);
            float *D = calloc(mn ? mn : 1, 4);
            if (orc_qmax(S, D, M, N) != (float)sc[0] || memcmp(D, Dq, 4 * mn)) bad += fail("qmax", rec);
            if (orc_dmax(S, D, M, N) != (float)sc[1] || memcmp(D, Ddr, 4 * mn)) bad += fail("dmax on qmax's D", rec);      /* Serra09.py:173-175 */
            memset(D, 0, 4 * mn);
            if (orc_dmax(S, D, M, N) != (float)sc[2] || memcmp(D, Ddf, 4 * mn)) bad += fail("dmax", rec);
            float *W = calloc(sw, 4);
            if (fabs((double)orc_swc(S, W, M, N) - sc[3]) > 1e-5) bad += fail("swalignimpconstrained score", rec);
            for (size_t i = 0; i < sw; i++)
                if (fabs((double)W[i] - (double)Dsw[i]) > 1e-5) { bad += fail("swalignimpconstrained D", rec); break; }
            free(S); free(Dq); free(Ddr); free(Ddf); free(Dsw); free(sc); free(D); free(W);
            n_dp++;
        } else if (!memcmp(tag, "STG ", 4)) {
            int32_t h[5];
            double kappa;
            if (fread(h, 4, 5, f) != 5 || fread(&kappa, 8, 1, f) != 1) return 2;
            const int nx = h[0], ny = h[1], d = h[2], m = h[3], oti = h[4];
            const long M = nx - m + 1, N = ny - m + 1;
            double *X = take(f, 8 * (size_t)nx * d), *Y = take(f, 8 * (size_t)ny * d), *gX = take(f, 8 * (size_t)d), *gY = take(f, 8 * (size_t)d);
            double *C0 = take(f, 8 * (size_t)nx * ny), *S0 = take(f, 8 * (size_t)M * N);
            uint8_t *B1 = take(f, (size_t)M * N), *B0 = take(f, (size_t)M * N);
            if (orc_get_oti(gX, gY, d) != oti) bad += fail("get_oti", rec);
            double *C = malloc(8 * (size_t)nx * ny), *S = malloc(8 * (size_t)M * N);
            uint8_t *B = malloc((size_t)M * N);
            orc_csm_f64(X, nx, Y, ny, d, oti, C);
            for (size_t i = 0; i < (size_t)nx * ny; i++)
                if (fabs(C[i] - C0[i]) > 1e-9) { bad += fail("get_csm", rec); break; }      /* BLAS summation order */
            if (orc_sliding_csm_f64(C0, nx, ny, m, S) != 0 || memcmp(S, S0, 8 * (size_t)M * N)) bad += fail("sliding_csm", rec);
            orc_csm_to_binary(S0, M, N, kappa, B);
            if (memcmp(B, B1, (size_t)M * N)) bad += fail("csm_to_binary", rec);
            orc_csm_to_binary_mutual(S0, M, N, kappa, B);
            if (memcmp(B, B0, (size_t)M * N)) bad += fail("csm_to_binary_mutual", rec);
            double q, dm;
            if (orc_serra09_pair(X, gX, nx, Y, gY, ny, d, m, kappa, 1, &q, &dm) != 0) bad += fail("serra09_pair", rec);
            free(X); free(Y); free(gX); free(gY); free(C0); free(S0); free(B1); free(B0); free(C); free(S); free(B);
            n_stage++;
        } else {
            fprintf(stderr, "asan_driver: unknown record tag\n");
            return 2;
        }
    }
    fclose(f);
    /* degenerate shapes the fixtures do not hold: matrices smaller than the recurrences' borders, windows as long as a song */
    for (int M = 0; M <= 4; M++)
        for (int N = 0; N <= 4; N++) {
            unsigned char *S = malloc((size_t)M * N + 1);
            float *D = calloc((size_t)M * N + 1, 4), *W = calloc((size_t)(M + 1) * (N + 1), 4);
            memset(S, 1, (size_t)M * N + 1);
            (void)orc_qmax(S, D, M, N);
            (void)orc_dmax(S, D, M, N);
            (void)orc_swc(S, W, M, N);
            free(S); free(D); free(W);
        }
    {
        double Z[9 * 12], g[12], S1[1], qd[2];
        for (int i = 0; i < 9 * 12; i++) Z[i] = (double)((i * 37) % 11) / 11.0 + 0.01;
        if (orc_global_chroma(Z, 9, 12, g) != 0) bad += fail("global_chroma", -1);
        double *C = malloc(8 * 81);
        orc_csm_f64(Z, 9, Z, 9, 12, 3, C);
        if (orc_sliding_csm_f64(C, 9, 9, 9, S1) != 0) bad += fail("sliding_csm 1x1", -1);
        if (orc_sliding_csm_f64(C, 9, 9, 10, S1) != -1) bad += fail("sliding_csm too short", -1);
        if (orc_serra09_pair(Z, g, 9, Z, g, 9, 12, 9, 0.095, 1, &qd[0], &qd[1]) != 0) bad += fail("serra09_pair 1x1", -1);
        free(C);
    }
    printf("asan_driver: %d alignment records, %d stage records, %d mismatches\n", n_dp, n_stage, bad);
    return bad ? 1 : 0;
}
