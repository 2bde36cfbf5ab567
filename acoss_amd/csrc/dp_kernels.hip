// dp_kernels.hip -- the three alignment recurrences of benchmarking/SequenceAlignment.c as
// row-sweep kernels for gfx950:  qmax (:113-143), dmax (:147-180), constrained
// Smith-Waterman (:73-99).
//
// Every predecessor of cell (i, j) lies in rows i-1..i-3, so all cells of a row are independent
// given the previous rows: ONE WAVE owns one matrix and sweeps it row by row; lane l owns CPL
// adjacent columns whose previous-row values live in VGPRs, the left halo comes from lane l-1
// through DPP wave_shr, and there are no barriers and no LDS.  The mask row is fetched with one
// 16-byte load per lane (prefetched a few rows ahead) when the rows are 16-byte aligned.
// The only consumed output in the reference's callers is the maximum cell; D is written only
// when the caller passes a buffer (parity / drop-in use).
#include "common.h"

#include <stdlib.h>
#include <type_traits>
#include "wave_ops.h"
#include "thresh_work.h"

#include <math.h>

namespace acoss {

enum { KIND_QMAX = 0, KIND_DMAX = 1, KIND_SWC = 2 };

struct DpParams {
    float g_on, g_ext;                  // gammaState
    float sw_match, sw_mismatch, sw_open, sw_ext;
    int boundary;                       // dmax: 1 = boundary left behind by qmax on a shared D
};

__device__ inline float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// byte c of a CPL-byte row segment held in CPL/4 dwords; c in [-4, -1] reads the halo dword
// (the last dword of lane l-1's segment)
template <int NW>
__device__ inline unsigned seg_byte(const unsigned (&w)[NW], unsigned halo, int c)
{
    const unsigned word = c < 0 ? halo : w[c >> 2];
    return (word >> (8 * (c & 3))) & 0xffu;
}

template <int NW>
__device__ inline void load_mask_row(const uint8_t *row, int j0, int N, bool aligned, unsigned (&w)[NW])
{
    // row == nullptr: rows above the matrix (all zero)
    if (row == nullptr) {
#pragma unroll
        for (int q = 0; q < NW; q++) w[q] = 0;
        return;
    }
    if (aligned) {
        // wave-uniform branch; lanes past the row end re-read its last aligned 16 bytes (in bounds:
        // the pitch is a multiple of 16) -- their cells are masked out by the column validity test
        const int last = ((N + 15) & ~15) - 16;
#pragma unroll
        for (int q = 0; q < NW; q += 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(row + min(j0 + 4 * q, last));
            w[q] = v.x; w[q + 1] = v.y; w[q + 2] = v.z; w[q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NW; q++) {
            unsigned x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int j = j0 + 4 * q + b;
                if (j < N) x |= (unsigned)row[j] << (8 * b);
            }
            w[q] = x;
        }
    }
}

// One wave per matrix, CPL columns per lane (cols <= 64*CPL).
template <int KIND, int CPL>
__global__ __launch_bounds__(256) void dp_wave_kernel(const uint8_t *__restrict__ S,
                                                      const acoss_mat_desc *__restrict__ mats, int K,
                                                      float *__restrict__ D, DpParams prm,
                                                      float *__restrict__ scores)
{
    constexpr int NW = CPL / 4;
    constexpr int FIRST = (KIND == KIND_DMAX) ? 3 : 2;   // first row/column the recurrence writes
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * 4 + wave;
    if (p >= K) return;
    const int lane = threadIdx.x & 63;
    const acoss_mat_desc md = mats[p];
    const int M = md.rows, N = md.cols;
    if (M < FIRST + 1 || N < FIRST + 1) {   // SequenceAlignment.c:78-80 / 117-119 / 151-153
        if (lane == 0) scores[p] = 0.0f;
        return;
    }
    const uint8_t *sbase = S + md.s_off;
    const bool aligned = ((md.s_pitch & 15) == 0) && ((md.s_off & 15) == 0);
    // For the Smith-Waterman the D grid is offset by one row and one column (SequenceAlignment.c:77,85)
    const int dshift = (KIND == KIND_SWC) ? 1 : 0;
    float *dbase = D ? D + md.d_off + (int64_t)dshift * md.d_pitch + dshift : nullptr;
    const int j0 = lane * CPL;

    float d1[CPL], d2[CPL], d3[CPL];           // rows i-1, i-2, i-3 of D
    unsigned s1[NW], s2[NW], s3[NW];           // rows i-1, i-2, i-3 of the mask
    // ---- initial rows (0 .. FIRST-1)
#pragma unroll
    for (int c = 0; c < CPL; c++) { d1[c] = 0.f; d2[c] = 0.f; d3[c] = 0.f; }
    if (dbase) {
        // in/out semantics of the reference: untouched cells of D are boundary values
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int j = j0 + c;
            if (j < N) {
                d1[c] = dbase[(int64_t)(FIRST - 1) * md.d_pitch + j];
                d2[c] = dbase[(int64_t)(FIRST - 2) * md.d_pitch + j];
                if (KIND == KIND_DMAX) d3[c] = dbase[j];
            }
        }
    }
    load_mask_row<NW>(sbase + (int64_t)(FIRST - 1) * md.s_pitch, j0, N, aligned, s1);
    load_mask_row<NW>(sbase + (int64_t)(FIRST - 2) * md.s_pitch, j0, N, aligned, s2);
    if (KIND == KIND_DMAX) load_mask_row<NW>(sbase, j0, N, aligned, s3);
    else load_mask_row<NW>(nullptr, j0, N, aligned, s3);
    if (KIND == KIND_DMAX && !dbase && prm.boundary) {
        // qmax leaves D[2][j] = (S[2][j] == 1) for j >= 2 on a zeroed D (Serra09.py:173-175)
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int j = j0 + c;
            d1[c] = (j >= 2 && j < N && seg_byte<NW>(s1, 0, c) == 1u) ? 1.0f : 0.0f;
        }
    }

    float best = 0.0f;
    // Mask rows are prefetched PF rows ahead into a register ring (the row sweep is a serial chain:
    // without this every step would expose a full HBM/L2 round trip).  The ring is indexed
    // statically by unrolling the row loop PF times; the slot a row is consumed from is refilled at
    // once with the row PF steps ahead.
    constexpr int PF = (NW <= 4) ? 8 : 4;
    unsigned ring[PF][NW];
#pragma unroll
    for (int u = 0; u < PF; u++) {
        if (FIRST + u < M) load_mask_row<NW>(sbase + (int64_t)(FIRST + u) * md.s_pitch, j0, N, aligned, ring[u]);
        else load_mask_row<NW>(nullptr, j0, N, aligned, ring[u]);
    }
    auto do_row = [&](const int i, const unsigned (&s0)[NW]) {
            // left halo from lane l-1 (lane 0 gets zeros; its columns < FIRST are never computed)
            const float h1a = lane_shr1(d1[CPL - 1], 0.f), h1b = lane_shr1(d1[CPL - 2], 0.f);
            const float h1c = lane_shr1(d1[CPL - 3], 0.f);
            const float h2a = lane_shr1(d2[CPL - 1], 0.f), h3a = lane_shr1(d3[CPL - 1], 0.f);
            const unsigned hs0 = (unsigned)lane_shr1((int)s0[NW - 1], 0);
            const unsigned hs1 = (unsigned)lane_shr1((int)s1[NW - 1], 0);
            const unsigned hs2 = (unsigned)lane_shr1((int)s2[NW - 1], 0);
            const unsigned hs3 = (unsigned)lane_shr1((int)s3[NW - 1], 0);

            float nd[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int j = j0 + c;
                const float p_diag = c >= 1 ? d1[c - 1] : h1a;                       // D[i-1][j-1]
                const float p_up2 = c >= 1 ? d2[c - 1] : h2a;                        // D[i-2][j-1]
                const float p_left2 = c >= 2 ? d1[c - 2] : (c == 1 ? h1a : h1b);     // D[i-1][j-2]
                const unsigned cur = seg_byte<NW>(s0, hs0, c);
                float v;
                if (KIND == KIND_QMAX) {
                    if (cur == 1u) {
                        v = max3f(p_diag, p_up2, p_left2) + 1.0f;
                    } else {
                        const float g1 = seg_byte<NW>(s1, hs1, c - 1) == 1u ? prm.g_on : prm.g_ext;
                        const float g2 = seg_byte<NW>(s2, hs2, c - 1) == 1u ? prm.g_on : prm.g_ext;
                        const float g3 = seg_byte<NW>(s1, hs1, c - 2) == 1u ? prm.g_on : prm.g_ext;
                        v = fmaxf(max3f(p_diag - g1, p_up2 - g2, p_left2 - g3), 0.0f);
                    }
                } else if (KIND == KIND_DMAX) {
                    const float p_up3 = c >= 1 ? d3[c - 1] : h3a;                                   // D[i-3][j-1]
                    const float p_left3 = c >= 3 ? d1[c - 3] : (c == 2 ? h1a : (c == 1 ? h1b : h1c));  // D[i-1][j-3]
                    const float s_up1 = (float)seg_byte<NW>(s1, hs1, c);        // S[i-1][j]
                    const float s_up2 = (float)seg_byte<NW>(s2, hs2, c);        // S[i-2][j]
                    const float s_l1 = (float)seg_byte<NW>(s0, hs0, c - 1);     // S[i][j-1]
                    const float s_l2 = (float)seg_byte<NW>(s0, hs0, c - 2);     // S[i][j-2]
                    float c1 = p_diag;
                    float c2 = p_up2 + s_up1;
                    float c3 = p_left2 + s_l1;
                    float c4 = (p_up3 + s_up2) + s_up1;
                    float c5 = (p_left3 + s_l2) + s_l1;
                    if (cur == 1u) {
                        v = fmaxf(fmaxf(max3f(c1, c2, c3), c4), c5) + 1.0f;
                    } else {
                        c1 -= seg_byte<NW>(s1, hs1, c - 1) == 1u ? prm.g_on : prm.g_ext;
                        c2 -= seg_byte<NW>(s2, hs2, c - 1) == 1u ? prm.g_on : prm.g_ext;
                        c3 -= seg_byte<NW>(s1, hs1, c - 2) == 1u ? prm.g_on : prm.g_ext;
                        c4 -= seg_byte<NW>(s3, hs3, c - 1) == 1u ? prm.g_on : prm.g_ext;
                        c5 -= seg_byte<NW>(s1, hs1, c - 3) == 1u ? prm.g_on : prm.g_ext;
                        v = fmaxf(fmaxf(fmaxf(max3f(c1, c2, c3), c4), c5), 0.0f);
                    }
                } else {   // KIND_SWC, in mask coordinates (a, b) = (i, j), D cell (a+1, b+1)
                    const float ms = cur == 0u ? prm.sw_mismatch : prm.sw_match;
                    const unsigned q1 = seg_byte<NW>(s1, hs1, c - 1);   // S[a-1][b-1]
                    const unsigned q2 = seg_byte<NW>(s2, hs2, c - 1);   // S[a-2][b-1]
                    const unsigned q3 = seg_byte<NW>(s1, hs1, c - 2);   // S[a-1][b-2]
                    const float e1 = cur > 0u ? 0.0f : (q1 > 0u ? prm.sw_open : prm.sw_ext);
                    const float e2 = cur > 0u ? 0.0f : (q2 > 0u ? prm.sw_open : prm.sw_ext);
                    const float e3 = cur > 0u ? 0.0f : (q3 > 0u ? prm.sw_open : prm.sw_ext);
                    v = fmaxf(max3f((p_diag + ms) + e1, (p_up2 + ms) + e2, (p_left2 + ms) + e3), 0.0f);
                }
                const bool valid = j >= FIRST && j < N;
                nd[c] = valid ? v : 0.0f;
            }
            // columns the recurrence never writes keep their boundary values
            if (j0 < FIRST) {
#pragma unroll
                for (int c = 0; c < FIRST; c++) {
                    if (dbase) {
                        nd[c] = dbase[(int64_t)i * md.d_pitch + c];
                    } else if (KIND == KIND_DMAX && prm.boundary && c == 2) {
                        nd[c] = seg_byte<NW>(s0, 0, 2) == 1u ? 1.0f : 0.0f;   // qmax's D[i][2]
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int j = j0 + c;
                const bool valid = j >= FIRST && j < N;
                if (valid) best = fmaxf(best, nd[c]);
                if (dbase && valid) dbase[(int64_t)i * md.d_pitch + j] = nd[c];
            }
#pragma unroll
            for (int c = 0; c < CPL; c++) { d3[c] = d2[c]; d2[c] = d1[c]; d1[c] = nd[c]; }
#pragma unroll
            for (int q = 0; q < NW; q++) { s3[q] = s2[q]; s2[q] = s1[q]; s1[q] = s0[q]; }
    };
    for (int ib = FIRST; ib < M; ib += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int i = ib + u;
            if (i < M) {
                unsigned s0[NW];
#pragma unroll
                for (int q = 0; q < NW; q++) s0[q] = ring[u][q];
                if (i + PF < M) load_mask_row<NW>(sbase + (int64_t)(i + PF) * md.s_pitch, j0, N, aligned, ring[u]);
                do_row(i, s0);
            }
        }
    }
    // wave max
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    if (lane == 0) scores[p] = best;
}

// Any width: one workgroup per matrix, the last three D rows and mask rows in LDS, one barrier
// per row.  Slow path for matrices wider than 64*32 columns.
template <int KIND>
__global__ __launch_bounds__(256) void dp_block_kernel(const uint8_t *__restrict__ S,
                                                       const acoss_mat_desc *__restrict__ mats,
                                                       float *__restrict__ D, DpParams prm,
                                                       float *__restrict__ scores, int ld)
{
    extern __shared__ float lds_f[];
    constexpr int FIRST = (KIND == KIND_DMAX) ? 3 : 2;
    float *drow = lds_f;                                    // [4][ld]
    uint8_t *srow = reinterpret_cast<uint8_t *>(drow + 4 * ld);   // [4][ld]
    __shared__ float red[4];
    const acoss_mat_desc md = mats[blockIdx.x];
    const int M = md.rows, N = md.cols;
    if (M < FIRST + 1 || N < FIRST + 1) {
        if (threadIdx.x == 0) scores[blockIdx.x] = 0.0f;
        return;
    }
    const uint8_t *sbase = S + md.s_off;
    const int dshift = (KIND == KIND_SWC) ? 1 : 0;
    float *dbase = D ? D + md.d_off + (int64_t)dshift * md.d_pitch + dshift : nullptr;
    for (int r = 0; r < FIRST; r++)
        for (int j = threadIdx.x; j < N; j += 256) {
            float dv = dbase ? dbase[(int64_t)r * md.d_pitch + j] : 0.0f;
            const uint8_t sv = sbase[(int64_t)r * md.s_pitch + j];
            if (KIND == KIND_DMAX && !dbase && prm.boundary && r == 2) dv = (j >= 2 && sv == 1) ? 1.0f : 0.0f;
            drow[(r & 3) * ld + j] = dv;
            srow[(r & 3) * ld + j] = sv;
        }
    float best = 0.0f;
    for (int i = FIRST; i < M; i++) {
        __syncthreads();
        const float *e1 = drow + ((i - 1) & 3) * ld, *e2 = drow + ((i - 2) & 3) * ld, *e3 = drow + ((i - 3) & 3) * ld;
        const uint8_t *t1 = srow + ((i - 1) & 3) * ld, *t2 = srow + ((i - 2) & 3) * ld, *t3 = srow + ((i - 3) & 3) * ld;
        float *e0 = drow + (i & 3) * ld;
        uint8_t *t0 = srow + (i & 3) * ld;
        const uint8_t *grow = sbase + (int64_t)i * md.s_pitch;
        for (int j = threadIdx.x; j < N; j += 256) t0[j] = grow[j];
        __syncthreads();
        for (int j = threadIdx.x; j < N; j += 256) {
            float v;
            if (j < FIRST) {
                v = dbase ? dbase[(int64_t)i * md.d_pitch + j] : 0.0f;
                if (KIND == KIND_DMAX && !dbase && prm.boundary && j == 2) v = t0[2] == 1 ? 1.0f : 0.0f;
                e0[j] = v;
                continue;
            }
            const unsigned cur = t0[j];
            if (KIND == KIND_QMAX) {
                if (cur == 1u) v = max3f(e1[j - 1], e2[j - 1], e1[j - 2]) + 1.0f;
                else v = fmaxf(max3f(e1[j - 1] - (t1[j - 1] == 1 ? prm.g_on : prm.g_ext),
                                     e2[j - 1] - (t2[j - 1] == 1 ? prm.g_on : prm.g_ext),
                                     e1[j - 2] - (t1[j - 2] == 1 ? prm.g_on : prm.g_ext)), 0.0f);
            } else if (KIND == KIND_DMAX) {
                float c1 = e1[j - 1];
                float c2 = e2[j - 1] + (float)t1[j];
                float c3 = e1[j - 2] + (float)t0[j - 1];
                float c4 = (e3[j - 1] + (float)t2[j]) + (float)t1[j];
                float c5 = (e1[j - 3] + (float)t0[j - 2]) + (float)t0[j - 1];
                if (cur == 1u) v = fmaxf(fmaxf(max3f(c1, c2, c3), c4), c5) + 1.0f;
                else {
                    c1 -= t1[j - 1] == 1 ? prm.g_on : prm.g_ext;
                    c2 -= t2[j - 1] == 1 ? prm.g_on : prm.g_ext;
                    c3 -= t1[j - 2] == 1 ? prm.g_on : prm.g_ext;
                    c4 -= t3[j - 1] == 1 ? prm.g_on : prm.g_ext;
                    c5 -= t1[j - 3] == 1 ? prm.g_on : prm.g_ext;
                    v = fmaxf(fmaxf(fmaxf(max3f(c1, c2, c3), c4), c5), 0.0f);
                }
            } else {
                const float ms = cur == 0u ? prm.sw_mismatch : prm.sw_match;
                const float x1 = cur > 0u ? 0.0f : (t1[j - 1] > 0 ? prm.sw_open : prm.sw_ext);
                const float x2 = cur > 0u ? 0.0f : (t2[j - 1] > 0 ? prm.sw_open : prm.sw_ext);
                const float x3 = cur > 0u ? 0.0f : (t1[j - 2] > 0 ? prm.sw_open : prm.sw_ext);
                v = fmaxf(max3f((e1[j - 1] + ms) + x1, (e2[j - 1] + ms) + x2, (e1[j - 2] + ms) + x3), 0.0f);
            }
            e0[j] = v;
            best = fmaxf(best, v);
            if (dbase) dbase[(int64_t)i * md.d_pitch + j] = v;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) scores[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---------------------------------------------------------------------------------------------
// Mask + alignment fused (the product path): the wave that sweeps a pair's rows reads the rows of the
// windowed squared sums (acoss_crp_batch output, non-negative) and the kNN thresholds, forms the mutual
// mask bits of its 16 columns in registers and feeds them to the recurrence.  The uint8 mask is never
// written or read; per pair the kernel streams 8 bytes per cell once and emits one float.
// Constant gap penalty only (gamma_onset == gamma_extension, the reference's 0.5 / 0.5), columns <= 1024.
// ---------------------------------------------------------------------------------------------
__device__ inline double thr_value(uint64_t key)
{
    // keys of non-negative values have the sign bit set; 0 = "select nothing", ~0 = "select everything"
    if (key == ~0ull) return INFINITY;
    if ((key >> 63) == 0) return -1.0;
    return __longlong_as_double((long long)(key & 0x7fffffffffffffffull));
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void dp_fused_kernel(const double *__restrict__ T,
                                                          const acoss_pair_desc *__restrict__ descs, int K, int win,
                                                          ThreshWork w, int mutual, float gamma, int boundary,
                                                          float *__restrict__ scores)
{
    constexpr int CPL = 16;
    constexpr int FIRST = (KIND == KIND_DMAX) ? 3 : 2;
    constexpr int R0 = (KIND == KIND_DMAX) ? 1 : 2;     // first row whose mask is needed
    constexpr int PF = 2;                               // rows of T in flight (32 VGPRs each)
    // The 16 column thresholds of a lane are re-read every row: they live in LDS ([column][lane], so a
    // wave's read of one column slot is 512 contiguous bytes), not in 48 VGPRs.  Every lane reads only
    // what it wrote itself, so no barrier is involved.
    __shared__ double ct_lds[4][CPL][64];
    __shared__ int cc_lds[4][CPL][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * 4 + wave;
    if (p >= K) return;
    const int lane = threadIdx.x & 63;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (M < FIRST + 1 || N < FIRST + 1) {
        if (lane == 0) scores[p] = 0.0f;
        return;
    }
    const int j0 = lane * CPL;
    const double *base = T + ds.crp_off;
    const int last = ((N + 1) & ~1) - 2;     // last aligned pair of a row (the pitch is even and >= N)
    double (*ctw)[64] = ct_lds[wave];
    int (*ccw)[64] = cc_lds[wave];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int j = j0 + c;
        double tv = -1.0;      // columns past the matrix: never selected
        int cv = -1;
        if (j < N) {
            tv = mutual ? thr_value(w.col_thr[(int64_t)p * w.max_n + j]) : INFINITY;
            cv = mutual ? w.col_cut[(int64_t)p * w.max_n + j] : 0x7fffffff;
        }
        ctw[c][lane] = tv;
        ccw[c][lane] = cv;
    }
    // bit c set iff column j0 + c exists: slots past the row end hold clamped duplicates or, for odd N,
    // one uninitialised padding element -- never let them into the mask
    unsigned colmask = 0;
#pragma unroll
    for (int c = 0; c < CPL; c++) colmask |= (j0 + c < N ? 1u : 0u) << c;
    auto load_row = [&](const int i, double (&dst)[CPL]) {
        const double *row = base + (int64_t)i * ds.crp_pitch;
#pragma unroll
        for (int q = 0; q < CPL / 2; q++) {
            const double2 v = *reinterpret_cast<const double2 *>(row + min(j0 + 2 * q, last));
            dst[2 * q] = v.x;
            dst[2 * q + 1] = v.y;
        }
    };
    double ring[PF][CPL];
#pragma unroll
    for (int u = 0; u < PF; u++) load_row(min(R0 + u, M - 1), ring[u]);

    float d1[CPL], d2[CPL], d3[CPL];
    unsigned m1 = 0, m2 = 0;               // mask bits of rows i-1, i-2 (dmax)
#pragma unroll
    for (int c = 0; c < CPL; c++) { d1[c] = d2[c] = d3[c] = 0.f; }
    float best = 0.0f;
    const bool l0 = lane == 0;
    double rt_lane = -1.0;     // row thresholds of rows (i & ~63) + lane
    int rc_lane = -1;
    int rt_block = -1;

    auto do_row = [&](const int i, double (&t)[CPL]) {
        if ((i >> 6) != rt_block) {          // wave-uniform: fetch the next 64 row thresholds
            rt_block = i >> 6;
            const int r = min((rt_block << 6) + lane, M - 1);
            rt_lane = thr_value(w.row_thr[(int64_t)p * w.max_m + r]);
            rc_lane = w.row_cut[(int64_t)p * w.max_m + r];
        }
        const int sel = i & 63;
        const double rt = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rt_lane), sel),
                                           __builtin_amdgcn_readlane(__double2loint(rt_lane), sel));
        const int rc = __builtin_amdgcn_readlane(rc_lane, sel);
        unsigned m0 = 0;                     // this row's mask bits, bit c = column j0 + c
        bool col_tie = false;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int j = j0 + c;
            const double ctv = ctw[c][lane];
            // bitwise, not short-circuit, logic: no divergent branches per element
            const bool row_on = (t[c] < rt) | ((t[c] == rt) & (j <= rc));
            const bool col_on = t[c] < ctv;
            col_tie |= (t[c] == ctv);
            m0 |= ((row_on & col_on) ? 1u : 0u) << c;
        }
        if (__any(col_tie)) {                // a value equal to its column threshold: apply the tie cut
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int j = j0 + c;
                const bool row_on = (t[c] < rt) | ((t[c] == rt) & (j <= rc));
                const bool eq = (t[c] == ctw[c][lane]) & (i <= ccw[c][lane]);
                m0 |= ((row_on & eq) ? 1u : 0u) << c;
            }
        }
        m0 &= colmask;
        // the slot is consumed: refill it with the row PF steps ahead before the recurrence
        if (i + PF < M) load_row(i + PF, t);
        if (i >= FIRST) {
            const float h1a = lane_shr1(d1[CPL - 1], 0.f), h1b = lane_shr1(d1[CPL - 2], 0.f);
            const float h2a = lane_shr1(d2[CPL - 1], 0.f);
            float h1c = 0.f, h3a = 0.f;
            unsigned h0 = 0;                 // mask bits of lane l-1 (columns j0-16 .. j0-1)
            if (KIND == KIND_DMAX) {
                h1c = lane_shr1(d1[CPL - 3], 0.f);
                h3a = lane_shr1(d3[CPL - 1], 0.f);
                h0 = (unsigned)lane_shr1((int)m0, 0);
            }
            const unsigned ext = (m0 << 2) | ((h0 >> 14) & 3u);     // bit c+2 = S[i][j0 + c], bits 1,0 = j0-1, j0-2
            float nd[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const float p_diag = c >= 1 ? d1[c - 1] : h1a;
                const float p_up2 = c >= 1 ? d2[c - 1] : h2a;
                const float p_left2 = c >= 2 ? d1[c - 2] : (c == 1 ? h1a : h1b);
                float m = max3f(p_diag, p_up2, p_left2);
                if (KIND == KIND_DMAX) {
                    const float p_up3 = c >= 1 ? d3[c - 1] : h3a;
                    const float p_left3 = c >= 3 ? d1[c - 3] : (c == 2 ? h1a : (c == 1 ? h1b : h1c));
                    const float s_u1 = (float)((m1 >> c) & 1u);            // S[i-1][j]
                    const float s_u2 = (float)((m2 >> c) & 1u);            // S[i-2][j]
                    const float s_l1 = (float)((ext >> (c + 1)) & 1u);     // S[i][j-1]
                    const float s_l2 = (float)((ext >> c) & 1u);           // S[i][j-2]
                    const float c2 = p_up2 + s_u1;
                    const float c3 = p_left2 + s_l1;
                    const float c4 = (p_up3 + s_u2) + s_u1;
                    const float c5 = (p_left3 + s_l2) + s_l1;
                    m = fmaxf(fmaxf(max3f(p_diag, c2, c3), c4), c5);
                }
                const bool on = (m0 >> c) & 1u;
                float v = fmaxf(m + (on ? 1.0f : -gamma), 0.0f);
                if (c < FIRST) {
                    // columns the recurrence never writes (lane 0 only): zero, or what qmax leaves in
                    // column 2 of a shared D (Serra09.py:173-175)
                    const float bval = (KIND == KIND_DMAX && c == 2 && boundary) ? (float)((m0 >> 2) & 1u) : 0.0f;
                    best = fmaxf(best, l0 ? 0.0f : v);
                    v = l0 ? bval : v;
                } else {
                    best = fmaxf(best, v);
                }
                nd[c] = v;
            }
#pragma unroll
            for (int c = 0; c < CPL; c++) { d3[c] = d2[c]; d2[c] = d1[c]; d1[c] = nd[c]; }
        } else if (KIND == KIND_DMAX && i == 2) {
            // row 2 of D as qmax leaves it on a shared buffer: (S[2][j] == 1) for j >= 2, else zero
#pragma unroll
            for (int c = 0; c < CPL; c++) d1[c] = (boundary && j0 + c >= 2) ? (float)((m0 >> c) & 1u) : 0.0f;
        }
        m2 = m1;
        m1 = m0;
    };
    for (int ib = R0; ib < M; ib += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int i = ib + u;
            if (i < M) do_row(i, ring[u]);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    if (lane == 0) scores[p] = best;
}

// ---------------------------------------------------------------------------------------------
// Alignment from the bit-packed mutual mask (acoss_mask_bits_batch): one wave per pair, a lane reads the
// 16 mask bits of its columns as one uint16 per row (128 contiguous bytes per wave and row), 16 rows ahead.
// With ~60 VGPRs eight such waves share a SIMD, and a pair's whole mask is 124 KB: this sweep costs a
// fraction of the selection passes.  Constant gap penalty, <= 1024 columns.
// ---------------------------------------------------------------------------------------------
// the mask bits of a lane's CPL columns in one row: a uint16 / uint32 load; `ext` holds them shifted left by two
template <int CPL> struct BitsRow;
template <> struct BitsRow<16> { using type = unsigned short; using ext = unsigned; };
template <> struct BitsRow<32> { using type = unsigned; using ext = uint64_t; };

template <int KIND, int CPL>
__global__ __launch_bounds__(256, CPL == 16 ? 4 : 2) void dp_bits_kernel(const uint64_t *__restrict__ bits,
                                                      const acoss_pair_desc *__restrict__ descs, int K, int win,
                                                      int max_m, float gamma, int boundary, float4 sw,
                                                      float *__restrict__ scores)
{
    // sw = (match, mismatch, gap open, gap extension) of swalignimpconstrained (SequenceAlignment.c:43-63)
    using row_t = typename BitsRow<CPL>::type;
    using ext_t = typename BitsRow<CPL>::ext;
    constexpr int FIRST = (KIND == KIND_DMAX) ? 3 : 2;
    constexpr int R0 = (KIND == KIND_DMAX) ? 1 : 2;
    // rows per trip of the main loop = rows of mask held ahead in registers: a multiple of 3, so that the three value rows
    // (i-1, i-2, i-3) return to their registers at the end of a trip and the loop carries no copies
    constexpr int PF = CPL == 16 ? 12 : 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * 4 + wave;
    if (p >= K) return;
    const int lane = threadIdx.x & 63;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (M < FIRST + 1 || N < FIRST + 1) {
        if (lane == 0) scores[p] = 0.0f;
        return;
    }
    const int j0 = lane * CPL;
    const row_t *rowp = reinterpret_cast<const row_t *>(bits + (int64_t)p * max_m * CPL) + lane;
    float d1[CPL], d2[CPL], d3[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) { d1[c] = d2[c] = d3[c] = 0.f; }
    unsigned m1 = 0, m2 = 0;
    float best = 0.0f;
    const bool l0 = lane == 0;
    // MAIN (compile-time): row i >= FIRST for certain (the main loop): no test
    auto do_row = [&](const int i, const unsigned m0, auto main_tag) {
        if (decltype(main_tag)::value || i >= FIRST) {
            const float h1a = lane_shr1(d1[CPL - 1], 0.f), h1b = lane_shr1(d1[CPL - 2], 0.f);
            const float h2a = lane_shr1(d2[CPL - 1], 0.f);
            float h1c = 0.f, h3a = 0.f;
            unsigned h0 = 0;
            if (KIND == KIND_DMAX) {
                h1c = lane_shr1(d1[CPL - 3], 0.f);
                h3a = lane_shr1(d3[CPL - 1], 0.f);
                h0 = (unsigned)lane_shr1((int)m0, 0);
            }
            const ext_t ext = ((ext_t)m0 << 2) | ((h0 >> (CPL - 2)) & 3u);
            ext_t ext1 = 0, ext2 = 0;            // rows i-1, i-2 shifted so that bit c+2 = column c (KIND_SWC)
            if (KIND == KIND_SWC) {
                ext1 = ((ext_t)m1 << 2) | (((unsigned)lane_shr1((int)m1, 0) >> (CPL - 2)) & 3u);
                ext2 = ((ext_t)m2 << 2) | (((unsigned)lane_shr1((int)m2, 0) >> (CPL - 2)) & 3u);
            }
            float nd[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const float p_diag = c >= 1 ? d1[c - 1] : h1a;
                const float p_up2 = c >= 1 ? d2[c - 1] : h2a;
                const float p_left2 = c >= 2 ? d1[c - 2] : (c == 1 ? h1a : h1b);
                float m = max3f(p_diag, p_up2, p_left2);
                if (KIND == KIND_DMAX) {
                    const float p_up3 = c >= 1 ? d3[c - 1] : h3a;
                    const float p_left3 = c >= 3 ? d1[c - 3] : (c == 2 ? h1a : (c == 1 ? h1b : h1c));
                    const float s_u1 = (float)((m1 >> c) & 1u);
                    const float s_u2 = (float)((m2 >> c) & 1u);
                    const float s_l1 = (float)((unsigned)(ext >> (c + 1)) & 1u);
                    const float s_l2 = (float)((unsigned)(ext >> c) & 1u);
                    const float c2 = p_up2 + s_u1;
                    const float c3 = p_left2 + s_l1;
                    const float c4 = (p_up3 + s_u2) + s_u1;
                    const float c5 = (p_left3 + s_l2) + s_l1;
                    m = fmaxf(fmaxf(max3f(p_diag, c2, c3), c4), c5);
                }
                const bool on = (m0 >> c) & 1u;
                float v = fmaxf(m + (on ? 1.0f : -gamma), 0.0f);
                if (KIND == KIND_SWC) {
                    // same expression order as dp_wave_kernel<KIND_SWC> (the -0.7 penalty is inexact in float32)
                    const float ms = on ? sw.x : sw.y;
                    const float e1 = on ? 0.0f : (((unsigned)(ext1 >> (c + 1)) & 1u) ? sw.z : sw.w);     // S[a-1][b-1]
                    const float e2 = on ? 0.0f : (((unsigned)(ext2 >> (c + 1)) & 1u) ? sw.z : sw.w);     // S[a-2][b-1]
                    const float e3 = on ? 0.0f : (((unsigned)(ext1 >> c) & 1u) ? sw.z : sw.w);           // S[a-1][b-2]
                    v = fmaxf(max3f((p_diag + ms) + e1, (p_up2 + ms) + e2, (p_left2 + ms) + e3), 0.0f);
                }
                if (c < FIRST) {
                    const float bval = (KIND == KIND_DMAX && c == 2 && boundary) ? (float)((m0 >> 2) & 1u) : 0.0f;
                    best = fmaxf(best, l0 ? 0.0f : v);
                    v = l0 ? bval : v;
                } else {
                    best = fmaxf(best, v);
                }
                nd[c] = v;
            }
#pragma unroll
            for (int c = 0; c < CPL; c++) { d3[c] = d2[c]; d2[c] = d1[c]; d1[c] = nd[c]; }
        } else if (KIND == KIND_DMAX && i == 2) {
#pragma unroll
            for (int c = 0; c < CPL; c++) d1[c] = (boundary && j0 + c >= 2) ? (float)((m0 >> c) & 1u) : 0.0f;
        }
        m2 = m1;
        m1 = m0;
    };
    // (bits past column N are zero by construction)
    int i = R0;
#pragma unroll 1
    for (; i < min(FIRST, M); i++) do_row(i, (unsigned)rowp[(int64_t)i * 64], std::false_type{});       // rows before the first computed one
    unsigned ring[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) ring[u] = rowp[(int64_t)min(i + u, M - 1) * 64];
    for (; i + PF <= M; i += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const unsigned m0 = ring[u];
            ring[u] = rowp[(int64_t)min(i + u + PF, M - 1) * 64];
            do_row(i + u, m0, std::true_type{});
        }
    }
#pragma unroll 1
    for (; i < M; i++) do_row(i, (unsigned)rowp[(int64_t)i * 64], std::true_type{});        // fewer than PF rows left
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
    if (lane == 0) scores[p] = best;
}

// ---------------------------------------------------------------------------------------------
// qmax in 16-bit integers (round 3).  With gamma = 0.5 for onset and extension (gammaState's values, SequenceAlignment.c:104)
// every D value is a multiple of 0.5 below min(M, N) <= 1024, so E = 2 D is an integer below 2048 and the recurrence
// (SequenceAlignment.c:113-143) reads: match E = max3 + 2, mismatch E = max(max3 - 1, 0) -- a saturating subtraction.  Two
// cells per register, on the packed 16-bit instructions: per register and row two v_pk_max_u16, one v_pk_add_u16 of 3 or 0
// (the two cells' match bits, four registers at a time from a 256-entry table in LDS indexed by a byte of the row's mask bits)
// and one v_pk_sub_u16 clamp of 1 -- match: + 3 - 1, mismatch: -sat 1 --, the running maximum, and one v_alignbit that forms
// the row shifted by one cell (used as the diagonal predecessor of the next row and as the (i-2, j-1) predecessor of the one
// after).  ~59 vector instructions per row against 112 in float32; scores identical by construction (integers; best / 2 is exact).  Same data layout and
// row pipeline as dp_bits_kernel<KIND_QMAX, 16>.
// ---------------------------------------------------------------------------------------------
typedef unsigned short dp_u16x2 __attribute__((ext_vector_type(2)));
#ifndef DP_QD16_DEFAULT
#define DP_QD16_DEFAULT 1             // acoss_align_bits_qd_batch: 1 = the one-sweep 16-bit kernel (round 4), 0 = the two 16-bit kernels
#endif

__global__ __launch_bounds__(256, 6) void dp_bits_q16_kernel(const uint64_t *__restrict__ bits,
                                                              const acoss_pair_desc *__restrict__ descs, int K, int win,
                                                              int max_m, float *__restrict__ scores)
{
    __shared__ __attribute__((aligned(16))) uint4 lut[256];
    {
        // entry b: register r (cells 2r, 2r+1 of a byte's eight) -> 3 in the half of every set bit
        const unsigned b = threadIdx.x;
        unsigned e[4];
#pragma unroll
        for (int r = 0; r < 4; r++) e[r] = (((b >> (2 * r)) & 1u) ? 3u : 0u) | (((b >> (2 * r + 1)) & 1u) ? 0x30000u : 0u);
        lut[b] = make_uint4(e[0], e[1], e[2], e[3]);
    }
    __syncthreads();
    constexpr int PF = 12;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * 4 + wave;
    if (p >= K) return;
    const int lane = threadIdx.x & 63;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (M < 3 || N < 3) {                        // SequenceAlignment.c:117-119
        if (lane == 0) scores[p] = 0.0f;
        return;
    }
    const unsigned short *rowp = reinterpret_cast<const unsigned short *>(bits + (int64_t)p * max_m * 16) + lane;
    unsigned d1[8], d1s[8], d2s[8];              // row i-1; row i-1 and row i-2 shifted right by one cell
#pragma unroll
    for (int k = 0; k < 8; k++) d1[k] = d1s[k] = d2s[k] = 0u;
    unsigned best = 0u;
    const bool l0 = lane == 0;
    const dp_u16x2 one = (dp_u16x2){1, 1};
    auto pk = [](unsigned v) { return __builtin_bit_cast(dp_u16x2, v); };
    auto un = [](dp_u16x2 v) { return __builtin_bit_cast(unsigned, v); };
    // (the masks of a row are requested from the table one row ahead)
    auto do_row = [&](const uint4 ma, const uint4 mb) {
        const unsigned mask[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
        const unsigned halo = (unsigned)lane_shr1((int)d1[7], 0);       // cells -2, -1: the previous lane's last register
        unsigned nd[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned left2 = k >= 1 ? d1[k - 1] : halo;
            const dp_u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(pk(d1s[k]), pk(d2s[k])), pk(left2));
            unsigned v = un(__builtin_elementwise_sub_sat(m + pk(mask[k]), one));        // match: + 3 - 1; mismatch: -sat 1
            if (k == 0) v = l0 ? 0u : v;         // columns 0 and 1 are never written (SequenceAlignment.c:121)
            best = un(__builtin_elementwise_max(pk(best), pk(v)));
            nd[k] = v;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) d2s[k] = d1s[k];
#pragma unroll
        for (int k = 7; k >= 1; k--) d1s[k] = __builtin_amdgcn_alignbit(nd[k], nd[k - 1], 16);
        d1s[0] = __builtin_amdgcn_alignbit(nd[0], (unsigned)lane_shr1((int)nd[7], 0), 16);
#pragma unroll
        for (int k = 0; k < 8; k++) d1[k] = nd[k];
    };
    // (bits past column N are zero by construction; rows 0 and 1 are never written)
    int i = 2;
    unsigned ring[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) ring[u] = rowp[(int64_t)min(i + u, M - 1) * 64];
    uint4 na = lut[ring[0] & 0xFFu], nb = lut[(ring[0] >> 8) & 0xFFu];
    for (; i + PF <= M; i += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const uint4 ma = na, mb = nb;
            ring[u] = rowp[(int64_t)min(i + u + PF, M - 1) * 64];
            const unsigned mn = ring[(u + 1) % PF];          // row i + u + 1 (u = PF - 1: the row requested at u = 0 of this trip)
            na = lut[mn & 0xFFu];
            nb = lut[(mn >> 8) & 0xFFu];
            do_row(ma, mb);
        }
    }
#pragma unroll 1
    for (; i < M; i++) {
        const unsigned m0 = rowp[(int64_t)i * 64];
        do_row(lut[m0 & 0xFFu], lut[(m0 >> 8) & 0xFFu]);
    }
    unsigned b = max(best & 0xFFFFu, best >> 16);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, off));
    if (lane == 0) scores[p] = 0.5f * (float)b;
}

// dmax in 16-bit integers: the same construction for SequenceAlignment.c:147-180 (E = 2 D <= 6 * 1024: every step adds at most
// 1 + two mask values).  Five predecessors, four of them with mask values of the skipped cells added (2 per set bit in E
// units): (i-2, j-1) + S[i-1][j]; (i-1, j-2) + S[i][j-1]; (i-3, j-1) + S[i-2][j] + S[i-1][j]; (i-1, j-3) + S[i][j-2] + S[i][j-1].
// Rows shifted by one cell are kept for i-1, i-2, i-3 (one v_alignbit per register and row forms the new one), shifts by two
// and three are the neighbouring registers of the plain and the shifted row.  boundary = 1: D as qmax leaves it (Serra09.py:
// 173-175): row 2 and column 2 hold the mask values.
__global__ __launch_bounds__(256, 4) void dp_bits_d16_kernel(const uint64_t *__restrict__ bits,
                                                              const acoss_pair_desc *__restrict__ descs, int K, int win,
                                                              int max_m, int boundary, float *__restrict__ scores)
{
    __shared__ __attribute__((aligned(16))) uint4 lut[256];
    {
        const unsigned b = threadIdx.x;
        unsigned e[4];
#pragma unroll
        for (int r = 0; r < 4; r++) e[r] = (((b >> (2 * r)) & 1u) ? 3u : 0u) | (((b >> (2 * r + 1)) & 1u) ? 0x30000u : 0u);
        lut[b] = make_uint4(e[0], e[1], e[2], e[3]);
    }
    __syncthreads();
    constexpr int PF = 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * 4 + wave;
    if (p >= K) return;
    const int lane = threadIdx.x & 63;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (M < 4 || N < 4) {                        // SequenceAlignment.c:151-153
        if (lane == 0) scores[p] = 0.0f;
        return;
    }
    const unsigned short *rowp = reinterpret_cast<const unsigned short *>(bits + (int64_t)p * max_m * 16) + lane;
    auto pk = [](unsigned v) { return __builtin_bit_cast(dp_u16x2, v); };
    auto un = [](dp_u16x2 v) { return __builtin_bit_cast(unsigned, v); };
    const dp_u16x2 one = (dp_u16x2){1, 1};
    const bool l0 = lane == 0;
    // the mask values (2 per set bit) of a row's cells, per register
    auto row_vals = [&](const unsigned m0, unsigned (&m3)[8], unsigned (&a)[8]) {
        const uint4 ma = lut[m0 & 0xFFu], mb = lut[(m0 >> 8) & 0xFFu];
        m3[0] = ma.x; m3[1] = ma.y; m3[2] = ma.z; m3[3] = ma.w; m3[4] = mb.x; m3[5] = mb.y; m3[6] = mb.z; m3[7] = mb.w;
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = m3[k] & 0x00020002u;
    };
    unsigned d1[8], d1s[8], d2s[8], d3s[8];      // row i-1; rows i-1, i-2, i-3 shifted right by one cell
    unsigned a1[8], a2[8];                       // mask values of rows i-1, i-2
#pragma unroll
    for (int k = 0; k < 8; k++) d1[k] = d1s[k] = d2s[k] = d3s[k] = a1[k] = a2[k] = 0u;
    // rows 1 and 2 only leave their mask values behind; with the boundary row 2 of D holds them too (columns >= 2)
    {
        unsigned m3[8];
        row_vals((unsigned)rowp[(int64_t)1 * 64], m3, a2);
        row_vals((unsigned)rowp[(int64_t)2 * 64], m3, a1);
        if (boundary) {
#pragma unroll
            for (int k = 0; k < 8; k++) d1[k] = a1[k];
            if (l0) d1[0] = 0u;
#pragma unroll
            for (int k = 7; k >= 1; k--) d1s[k] = __builtin_amdgcn_alignbit(d1[k], d1[k - 1], 16);
            d1s[0] = __builtin_amdgcn_alignbit(d1[0], (unsigned)lane_shr1((int)d1[7], 0), 16);
        }
    }
    unsigned best = 0u;
    auto do_row = [&](const unsigned m0) {
        unsigned m3[8], a0[8];
        row_vals(m0, m3, a0);
        const unsigned a0_prev = (unsigned)lane_shr1((int)a0[7], 0);     // cells -2, -1 of this row's mask values
        const unsigned d1_prev = (unsigned)lane_shr1((int)d1[7], 0);
        const unsigned d1s_prev = (unsigned)lane_shr1((int)d1s[7], 0);
        unsigned nd[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned a0m1 = k >= 1 ? a0[k - 1] : a0_prev;                    // S[i][j-2] (x 2)
            const dp_u16x2 sl1 = pk(__builtin_amdgcn_alignbit(a0[k], a0m1, 16));   // S[i][j-1]
            const dp_u16x2 su1 = pk(a1[k]), su2 = pk(a2[k]);                        // S[i-1][j], S[i-2][j]
            const dp_u16x2 c1 = pk(d1s[k]);                                          // (i-1, j-1)
            const dp_u16x2 c2 = pk(d2s[k]) + su1;                                    // (i-2, j-1)
            const dp_u16x2 c3 = pk(k >= 1 ? d1[k - 1] : d1_prev) + sl1;              // (i-1, j-2)
            const dp_u16x2 c4 = pk(d3s[k]) + (su2 + su1);                            // (i-3, j-1)
            const dp_u16x2 c5 = pk(k >= 1 ? d1s[k - 1] : d1s_prev) + (pk(a0m1) + sl1);      // (i-1, j-3)
            const dp_u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_max(c1, c2), __builtin_elementwise_max(c3, c4)), c5);
            unsigned v = un(__builtin_elementwise_sub_sat(m + pk(m3[k]), one));
            unsigned vb = v;                     // what counts for the maximum: columns >= 3
            if (k == 0) { v = l0 ? 0u : v; vb = v; }
            if (k == 1) {
                vb = l0 ? (v & 0xFFFF0000u) : v;
                v = l0 ? ((v & 0xFFFF0000u) | (boundary ? (a0[1] & 0xFFFFu) : 0u)) : v;       // column 2 = S[i][2] with the boundary
            }
            best = un(__builtin_elementwise_max(pk(best), pk(vb)));
            nd[k] = v;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { d3s[k] = d2s[k]; d2s[k] = d1s[k]; a2[k] = a1[k]; a1[k] = a0[k]; }
#pragma unroll
        for (int k = 7; k >= 1; k--) d1s[k] = __builtin_amdgcn_alignbit(nd[k], nd[k - 1], 16);
        d1s[0] = __builtin_amdgcn_alignbit(nd[0], (unsigned)lane_shr1((int)nd[7], 0), 16);
#pragma unroll
        for (int k = 0; k < 8; k++) d1[k] = nd[k];
    };
    int i = 3;
    unsigned ring[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) ring[u] = rowp[(int64_t)min(i + u, M - 1) * 64];
    for (; i + PF <= M; i += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const unsigned m0 = ring[u];
            ring[u] = rowp[(int64_t)min(i + u + PF, M - 1) * 64];
            do_row(m0);
        }
    }
#pragma unroll 1
    for (; i < M; i++) do_row((unsigned)rowp[(int64_t)i * 64]);
    unsigned b = max(best & 0xFFFFu, best >> 16);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, off));
    if (lane == 0) scores[p] = 0.5f * (float)b;
}

// qmax AND dmax in 16-bit integers in ONE sweep over the mask (round 4): what Serra09.similarity asks for per mask
// (Serra09.py:173-175; `boundary` = dmax on the D that qmax leaves behind).  The two recurrences of dp_bits_q16_kernel and
// dp_bits_d16_kernel are independent given the mask rows, so their dependent chains interleave in one instruction stream: the
// mask row, its table look-ups and the loop are shared, and a wave always has a second chain to issue from.  Three changes
// against the separate dmax kernel: the mask values (2 per set bit) come from a second table instead of an AND per register;
// the (i-3, j-1) predecessor is carried -- c4(i) = D[i-3][j-1] + S[i-2][j] + S[i-1][j] = c2(i-1) + S[i-1][j], where c2(i-1) is
// the previous row's (i-2, j-1) term -- so neither the row i-3 nor the mask values of row i-2 are kept.  Scores identical by construction
// (integers).
template <int R>
__global__ __launch_bounds__(256, R == 8 ? 4 : (R == 12 ? 3 : 2)) void dp_bits_qd16_kernel(const uint64_t *__restrict__ bits,
                                                               const acoss_pair_desc *__restrict__ descs, int K, int win,
                                                               int max_m, int boundary, float *__restrict__ qscores,
                                                               float *__restrict__ dscores)
{
    __shared__ __attribute__((aligned(16))) uint4 lut3[256], lut2[256];
    {
        // entry b: register r (cells 2r, 2r+1 of a byte's eight) -> 3 (lut3) / 2 (lut2) in the half of every set bit
        const unsigned b = threadIdx.x;
        unsigned e[4];
#pragma unroll
        for (int r = 0; r < 4; r++) e[r] = (((b >> (2 * r)) & 1u) ? 1u : 0u) | (((b >> (2 * r + 1)) & 1u) ? 0x10000u : 0u);
        lut3[b] = make_uint4(3u * e[0], 3u * e[1], 3u * e[2], 3u * e[3]);
        lut2[b] = make_uint4(2u * e[0], 2u * e[1], 2u * e[2], 2u * e[3]);
    }
    __syncthreads();
    constexpr int PF = 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * 4 + wave;
    if (p >= K) return;
    const int lane = threadIdx.x & 63;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (M < 3 || N < 3) {                        // SequenceAlignment.c:117-119 (and :151-153)
        if (lane == 0) { qscores[p] = 0.0f; dscores[p] = 0.0f; }
        return;
    }
    const bool with_d = M >= 4 && N >= 4;        // SequenceAlignment.c:151-153: dmax of a smaller matrix is 0 (wave-uniform)
    // R registers of two 16-bit cells per lane: 2 R columns, their mask bits 2 R bits from bit 2 R lane of the row.  R = 8: rows of
    // 16 words (<= 1024 columns); R = 12 / 16: rows of 32 words, <= 1536 / 2048 columns (E = 2 D < 2 min(M, N) + 2 <= 4098 either way)
    constexpr int ROWB = R == 8 ? 128 : 256;                 // bytes per row of the mask
    const unsigned char *rowb = reinterpret_cast<const unsigned char *>(bits) + (int64_t)p * max_m * ROWB + lane * (R / 4);
    auto load_row = [&](const int i) -> unsigned {
        const unsigned char *src = rowb + (int64_t)i * ROWB;
        if (R == 8) return (unsigned)*reinterpret_cast<const unsigned short *>(src);
        if (R == 16) return *reinterpret_cast<const unsigned *>(src);
        unsigned v;                                           // R == 12: three bytes from byte 3 lane (the fourth is the next lane's or the row's unused tail)
        __builtin_memcpy(&v, src, 4);
        return v & 0xFFFFFFu;
    };
    auto pk = [](unsigned v) { return __builtin_bit_cast(dp_u16x2, v); };
    auto un = [](dp_u16x2 v) { return __builtin_bit_cast(unsigned, v); };
    const dp_u16x2 one = (dp_u16x2){1, 1};
    const bool l0 = lane == 0;
    // qmax state: row i-1; rows i-1 and i-2 shifted right by one cell
    unsigned q1[R], q1s[R], q2s[R], qbest = 0u;
    // dmax state: row i-1; rows i-1 and i-2 shifted right by one cell; the previous row's (i-2, j-1) term; mask values of row i-1
    unsigned d1[R], d1s[R], d2s[R], c2p[R], a1[R], dbest = 0u;
#pragma unroll
    for (int k = 0; k < R; k++) q1[k] = q1s[k] = q2s[k] = d1[k] = d1s[k] = d2s[k] = c2p[k] = a1[k] = 0u;
    // a row's mask bits -> per register the two cells' values (3 / 2 per set bit), a byte of bits at a time
    auto unpack = [](const uint4 *lut, const unsigned m, unsigned (&o)[R]) {
#pragma unroll
        for (int b = 0; b < R / 4; b++) {
            const uint4 a = lut[(m >> (8 * b)) & 0xFFu];
            o[4 * b] = a.x; o[4 * b + 1] = a.y; o[4 * b + 2] = a.z; o[4 * b + 3] = a.w;
        }
    };
    // one row of qmax (dp_bits_q16_kernel's)
    auto q_row = [&](const unsigned (&m3)[R]) {
        const unsigned halo = (unsigned)lane_shr1((int)q1[R - 1], 0);
        unsigned nd[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const unsigned left2 = k >= 1 ? q1[k - 1] : halo;
            const dp_u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(pk(q1s[k]), pk(q2s[k])), pk(left2));
            unsigned v = un(__builtin_elementwise_sub_sat(m + pk(m3[k]), one));
            if (k == 0) v = l0 ? 0u : v;
            qbest = un(__builtin_elementwise_max(pk(qbest), pk(v)));
            nd[k] = v;
        }
#pragma unroll
        for (int k = 0; k < R; k++) q2s[k] = q1s[k];
#pragma unroll
        for (int k = R - 1; k >= 1; k--) q1s[k] = __builtin_amdgcn_alignbit(nd[k], nd[k - 1], 16);
        q1s[0] = __builtin_amdgcn_alignbit(nd[0], (unsigned)lane_shr1((int)nd[R - 1], 0), 16);
#pragma unroll
        for (int k = 0; k < R; k++) q1[k] = nd[k];
    };
    // one row of dmax (dp_bits_d16_kernel's, with the carried (i-3, j-1) term)
    auto d_row = [&](const unsigned (&m3)[R], const unsigned (&a0)[R]) {
        const unsigned a0_prev = (unsigned)lane_shr1((int)a0[R - 1], 0);
        const unsigned d1_prev = (unsigned)lane_shr1((int)d1[R - 1], 0);
        const unsigned d1s_prev = (unsigned)lane_shr1((int)d1s[R - 1], 0);
        unsigned nd[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            const unsigned a0m1 = k >= 1 ? a0[k - 1] : a0_prev;                    // S[i][j-2] (x 2)
            const dp_u16x2 sl1 = pk(__builtin_amdgcn_alignbit(a0[k], a0m1, 16));   // S[i][j-1]
            const dp_u16x2 su1 = pk(a1[k]);                                          // S[i-1][j]
            const dp_u16x2 c1 = pk(d1s[k]);                                          // (i-1, j-1)
            const dp_u16x2 c2 = pk(d2s[k]) + su1;                                    // (i-2, j-1) + S[i-1][j]
            const dp_u16x2 c3 = pk(k >= 1 ? d1[k - 1] : d1_prev) + sl1;              // (i-1, j-2) + S[i][j-1]
            const dp_u16x2 c4 = pk(c2p[k]) + su1;                                    // (i-3, j-1) + S[i-2][j] + S[i-1][j]
            const dp_u16x2 c5 = pk(k >= 1 ? d1s[k - 1] : d1s_prev) + (pk(a0m1) + sl1);      // (i-1, j-3) + S[i][j-2] + S[i][j-1]
            const dp_u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_max(c1, c2), __builtin_elementwise_max(c3, c4)), c5);
            unsigned v = un(__builtin_elementwise_sub_sat(m + pk(m3[k]), one));
            unsigned vb = v;                     // what counts for the maximum: columns >= 3
            if (k == 0) { v = l0 ? 0u : v; vb = v; }
            if (k == 1) {
                vb = l0 ? (v & 0xFFFF0000u) : v;
                v = l0 ? ((v & 0xFFFF0000u) | (boundary ? (a0[1] & 0xFFFFu) : 0u)) : v;       // column 2 = S[i][2] with the boundary
            }
            dbest = un(__builtin_elementwise_max(pk(dbest), pk(vb)));
            nd[k] = v;
            c2p[k] = un(c2);
        }
#pragma unroll
        for (int k = 0; k < R; k++) { d2s[k] = d1s[k]; a1[k] = a0[k]; }
#pragma unroll
        for (int k = R - 1; k >= 1; k--) d1s[k] = __builtin_amdgcn_alignbit(nd[k], nd[k - 1], 16);
        d1s[0] = __builtin_amdgcn_alignbit(nd[0], (unsigned)lane_shr1((int)nd[R - 1], 0), 16);
#pragma unroll
        for (int k = 0; k < R; k++) d1[k] = nd[k];
    };
    // rows 1 and 2: dmax only takes their mask values (c2p = 0 + S[1][j]: the carried term of row 3; with the boundary row 2 of D
    // holds its mask values in columns >= 2); qmax computes row 2
    {
        const unsigned r1 = load_row(1), r2 = load_row(2);
        unsigned m3[R];
        unpack(lut2, r1, c2p);
        unpack(lut2, r2, a1);
        unpack(lut3, r2, m3);
        if (boundary) {
#pragma unroll
            for (int k = 0; k < R; k++) d1[k] = a1[k];
            if (l0) d1[0] = 0u;
#pragma unroll
            for (int k = R - 1; k >= 1; k--) d1s[k] = __builtin_amdgcn_alignbit(d1[k], d1[k - 1], 16);
            d1s[0] = __builtin_amdgcn_alignbit(d1[0], (unsigned)lane_shr1((int)d1[R - 1], 0), 16);
        }
        q_row(m3);
    }
    int i = 3;
    if (with_d) {
        unsigned ring[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) ring[u] = load_row(min(i + u, M - 1));
        for (; i + PF <= M; i += PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const unsigned m0 = ring[u];
                ring[u] = load_row(min(i + u + PF, M - 1));
                unsigned m3[R], a0[R];
                unpack(lut3, m0, m3);
                unpack(lut2, m0, a0);
                q_row(m3);
                d_row(m3, a0);
            }
        }
    }
#pragma unroll 1
    for (; i < M; i++) {
        const unsigned m0 = load_row(i);
        unsigned m3[R], a0[R];
        unpack(lut3, m0, m3);
        unpack(lut2, m0, a0);
        q_row(m3);
        if (with_d) d_row(m3, a0);
    }
    unsigned bq = max(qbest & 0xFFFFu, qbest >> 16), bd = max(dbest & 0xFFFFu, dbest >> 16);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        bq = max(bq, (unsigned)__shfl_xor((int)bq, off));
        bd = max(bd, (unsigned)__shfl_xor((int)bd, off));
    }
    if (lane == 0) {
        qscores[p] = 0.5f * (float)bq;
        dscores[p] = with_d ? 0.5f * (float)bd : 0.0f;
    }
}

// qmax and dmax of the same mask in ONE sweep (what Serra09.similarity asks for, Serra09.py:173-175: dmax on the D
// that qmax leaves behind = `boundary`): the two recurrences are independent given the mask rows, so their
// dependent chains interleave in one instruction stream and the mask is read once.  Same arithmetic per kind as
// dp_bits_kernel.
template <int CPL>
__global__ __launch_bounds__(256, CPL == 16 ? 2 : 1) void dp_bits_qd_kernel(const uint64_t *__restrict__ bits,
                                                         const acoss_pair_desc *__restrict__ descs, int K, int win,
                                                         int max_m, float gamma, int boundary,
                                                         float *__restrict__ qscores, float *__restrict__ dscores)
{
    using row_t = typename BitsRow<CPL>::type;
    using ext_t = typename BitsRow<CPL>::ext;
    constexpr int PF = 6;                       // a multiple of 2 and 3: both recurrences' rows are back in place after a trip
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * 4 + wave;
    if (p >= K) return;
    const int lane = threadIdx.x & 63;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const bool do_q = M >= 3 && N >= 3, do_d = M >= 4 && N >= 4;      // SequenceAlignment.c:117-119 / 151-153
    if (!do_q) {
        if (lane == 0) { qscores[p] = 0.0f; dscores[p] = 0.0f; }
        return;
    }
    const int j0 = lane * CPL;
    const row_t *rowp = reinterpret_cast<const row_t *>(bits + (int64_t)p * max_m * CPL) + lane;
    float q1[CPL], q2[CPL], e1[CPL], e2[CPL], e3[CPL];              // qmax rows i-1, i-2; dmax rows i-1, i-2, i-3
#pragma unroll
    for (int c = 0; c < CPL; c++) { q1[c] = q2[c] = e1[c] = e2[c] = e3[c] = 0.f; }
    unsigned m1 = 0, m2 = 0;
    float qbest = 0.0f, dbest = 0.0f;
    const bool l0 = lane == 0;
    auto do_row = [&](const int i, const unsigned m0, auto main_tag) {
        constexpr bool MAIN = decltype(main_tag)::value;        // row i >= 3 for certain
        if (MAIN || i >= 2) {       // ---- qmax (rows >= 2)
            const float h1a = lane_shr1(q1[CPL - 1], 0.f), h1b = lane_shr1(q1[CPL - 2], 0.f);
            const float h2a = lane_shr1(q2[CPL - 1], 0.f);
            float nd[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const float p_diag = c >= 1 ? q1[c - 1] : h1a;
                const float p_up2 = c >= 1 ? q2[c - 1] : h2a;
                const float p_left2 = c >= 2 ? q1[c - 2] : (c == 1 ? h1a : h1b);
                const bool on = (m0 >> c) & 1u;
                float v = fmaxf(max3f(p_diag, p_up2, p_left2) + (on ? 1.0f : -gamma), 0.0f);
                if (c < 2) {
                    qbest = fmaxf(qbest, l0 ? 0.0f : v);
                    v = l0 ? 0.0f : v;
                } else {
                    qbest = fmaxf(qbest, v);
                }
                nd[c] = v;
            }
#pragma unroll
            for (int c = 0; c < CPL; c++) { q2[c] = q1[c]; q1[c] = nd[c]; }
        }
        if (do_d) {
            if (MAIN || i >= 3) {   // ---- dmax (rows >= 3)
                const float h1a = lane_shr1(e1[CPL - 1], 0.f), h1b = lane_shr1(e1[CPL - 2], 0.f), h1c = lane_shr1(e1[CPL - 3], 0.f);
                const float h2a = lane_shr1(e2[CPL - 1], 0.f), h3a = lane_shr1(e3[CPL - 1], 0.f);
                const unsigned h0 = (unsigned)lane_shr1((int)m0, 0);
                const ext_t ext = ((ext_t)m0 << 2) | ((h0 >> (CPL - 2)) & 3u);
                float nd[CPL];
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const float p_diag = c >= 1 ? e1[c - 1] : h1a;
                    const float p_up2 = c >= 1 ? e2[c - 1] : h2a;
                    const float p_left2 = c >= 2 ? e1[c - 2] : (c == 1 ? h1a : h1b);
                    const float p_up3 = c >= 1 ? e3[c - 1] : h3a;
                    const float p_left3 = c >= 3 ? e1[c - 3] : (c == 2 ? h1a : (c == 1 ? h1b : h1c));
                    const float s_u1 = (float)((m1 >> c) & 1u);
                    const float s_u2 = (float)((m2 >> c) & 1u);
                    const float s_l1 = (float)((unsigned)(ext >> (c + 1)) & 1u);
                    const float s_l2 = (float)((unsigned)(ext >> c) & 1u);
                    const float c2 = p_up2 + s_u1;
                    const float c3 = p_left2 + s_l1;
                    const float c4 = (p_up3 + s_u2) + s_u1;
                    const float c5 = (p_left3 + s_l2) + s_l1;
                    const float m = fmaxf(fmaxf(max3f(p_diag, c2, c3), c4), c5);
                    const bool on = (m0 >> c) & 1u;
                    float v = fmaxf(m + (on ? 1.0f : -gamma), 0.0f);
                    if (c < 3) {
                        const float bval = (c == 2 && boundary) ? (float)((m0 >> 2) & 1u) : 0.0f;
                        dbest = fmaxf(dbest, l0 ? 0.0f : v);
                        v = l0 ? bval : v;
                    } else {
                        dbest = fmaxf(dbest, v);
                    }
                    nd[c] = v;
                }
#pragma unroll
                for (int c = 0; c < CPL; c++) { e3[c] = e2[c]; e2[c] = e1[c]; e1[c] = nd[c]; }
            } else if (i == 2) {
#pragma unroll
                for (int c = 0; c < CPL; c++) e1[c] = (boundary && j0 + c >= 2) ? (float)((m0 >> c) & 1u) : 0.0f;
            }
        }
        m2 = m1;
        m1 = m0;
    };
    int i = 1;
#pragma unroll 1
    for (; i < min(3, M); i++) do_row(i, (unsigned)rowp[(int64_t)i * 64], std::false_type{});
    unsigned ring[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) ring[u] = rowp[(int64_t)min(i + u, M - 1) * 64];
    for (; i + PF <= M; i += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const unsigned m0 = ring[u];        // bits past column N are zero by construction
            ring[u] = rowp[(int64_t)min(i + u + PF, M - 1) * 64];
            do_row(i + u, m0, std::true_type{});
        }
    }
#pragma unroll 1
    for (; i < M; i++) do_row(i, (unsigned)rowp[(int64_t)i * 64], std::true_type{});
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        qbest = fmaxf(qbest, __shfl_xor(qbest, off));
        dbest = fmaxf(dbest, __shfl_xor(dbest, off));
    }
    if (lane == 0) {
        qscores[p] = qbest;
        dscores[p] = do_d ? dbest : 0.0f;
    }
}

template <int KIND>
static int launch_dp(const uint8_t *S, const acoss_mat_desc *mats, int K, int max_cols, float *D,
                     int boundary, const acoss_align_params *params, float *scores, hipStream_t st)
{
    if (!S || !mats || !scores || K < 0 || max_cols < 0) { set_error("align batch: bad argument"); return ACOSS_EINVAL; }
    if (K == 0) return ACOSS_OK;
    acoss_align_params ap;
    if (params) ap = *params; else acoss_default_align_params(&ap);
    DpParams prm{ap.gamma_onset, ap.gamma_extension, ap.sw_match, ap.sw_mismatch, ap.sw_gap_open, ap.sw_gap_ext, boundary};
    if (max_cols <= 64 * 16) {
        hipLaunchKernelGGL((dp_wave_kernel<KIND, 16>), dim3(ceil_div(K, 4)), dim3(256), 0, st, S, mats, K, D, prm, scores);
        return launch_check("dp_wave_kernel<16>");
    }
    if (max_cols <= 64 * 32) {
        hipLaunchKernelGGL((dp_wave_kernel<KIND, 32>), dim3(ceil_div(K, 4)), dim3(256), 0, st, S, mats, K, D, prm, scores);
        return launch_check("dp_wave_kernel<32>");
    }
    const int ld = (max_cols + 3) & ~3;
    const size_t lds = (size_t)ld * 4 * (sizeof(float) + 1);
    if (lds > 160 * 1024 - 64) { set_error("align batch: %d columns exceed the LDS-resident limit", max_cols); return ACOSS_ENOTSUP; }
    ACOSS_HIP(hipFuncSetAttribute((const void *)dp_block_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(dp_block_kernel<KIND>, dim3(K), dim3(256), lds, st, S, mats, D, prm, scores, ld);
    return launch_check("dp_block_kernel");
}

}  // namespace acoss

using namespace acoss;

extern "C" {

int acoss_qmax_batch(const uint8_t *S, const acoss_mat_desc *mats, int K, int max_cols, float *D,
                     const acoss_align_params *params, float *scores, void *stream)
{
    return launch_dp<KIND_QMAX>(S, mats, K, max_cols, D, 0, params, scores, (hipStream_t)stream);
}

int acoss_dmax_batch(const uint8_t *S, const acoss_mat_desc *mats, int K, int max_cols, float *D,
                     int boundary, const acoss_align_params *params, float *scores, void *stream)
{
    return launch_dp<KIND_DMAX>(S, mats, K, max_cols, D, boundary, params, scores, (hipStream_t)stream);
}

int acoss_swc_batch(const uint8_t *S, const acoss_mat_desc *mats, int K, int max_cols, float *D,
                    const acoss_align_params *params, float *scores, void *stream)
{
    return launch_dp<KIND_SWC>(S, mats, K, max_cols, D, 0, params, scores, (hipStream_t)stream);
}

int acoss_align_bits_batch(int kind, const uint64_t *bits, const acoss_pair_desc *descs, int K, int win, int max_nx,
                           int max_ny, int boundary, const acoss_align_params *params, float *scores, void *stream)
{
    if (!bits || !descs || !scores || K < 0 || win < 1 || max_nx < win || max_ny < win || kind < 0 || kind > 2) {
        set_error("align_bits_batch: bad argument (kind 0 = qmax, 1 = dmax, 2 = swalignimpconstrained)");
        return ACOSS_EINVAL;
    }
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    acoss_align_params ap;
    if (params) ap = *params; else acoss_default_align_params(&ap);
    if (max_n > 2048 || max_m > 2048 || ap.gamma_onset != ap.gamma_extension) {
        set_error("align_bits_batch: needs <= 2048 x 2048 matrices and gamma_onset == gamma_extension");
        return ACOSS_ENOTSUP;
    }
    if (K == 0) return ACOSS_OK;
    hipStream_t st = (hipStream_t)stream;
    const float4 sw = make_float4(ap.sw_match, ap.sw_mismatch, ap.sw_gap_open, ap.sw_gap_ext);
    const dim3 grid(ceil_div(K, 4));
    if (mask_bits_words(max_m, max_n) == 16) {
        static const bool q16 = []() { const char *e = getenv("ACOSS_DP_Q16"); return !(e && e[0] == '0'); }();
        if (kind == 0 && q16 && ap.gamma_onset == 0.5f)        // (checked above: gamma_extension equals it)
            hipLaunchKernelGGL(dp_bits_q16_kernel, grid, dim3(256), 0, st, bits, descs, K, win, max_m, scores);
        else if (kind == 1 && q16 && ap.gamma_onset == 0.5f)
            hipLaunchKernelGGL(dp_bits_d16_kernel, grid, dim3(256), 0, st, bits, descs, K, win, max_m, boundary, scores);
        else if (kind == 0)
            hipLaunchKernelGGL((dp_bits_kernel<KIND_QMAX, 16>), grid, dim3(256), 0, st, bits, descs, K, win, max_m, ap.gamma_onset, 0, sw, scores);
        else if (kind == 1)
            hipLaunchKernelGGL((dp_bits_kernel<KIND_DMAX, 16>), grid, dim3(256), 0, st, bits, descs, K, win, max_m, ap.gamma_onset, boundary, sw, scores);
        else
            hipLaunchKernelGGL((dp_bits_kernel<KIND_SWC, 16>), grid, dim3(256), 0, st, bits, descs, K, win, max_m, ap.gamma_onset, 0, sw, scores);
    } else {
        if (kind == 0)
            hipLaunchKernelGGL((dp_bits_kernel<KIND_QMAX, 32>), grid, dim3(256), 0, st, bits, descs, K, win, max_m, ap.gamma_onset, 0, sw, scores);
        else if (kind == 1)
            hipLaunchKernelGGL((dp_bits_kernel<KIND_DMAX, 32>), grid, dim3(256), 0, st, bits, descs, K, win, max_m, ap.gamma_onset, boundary, sw, scores);
        else
            hipLaunchKernelGGL((dp_bits_kernel<KIND_SWC, 32>), grid, dim3(256), 0, st, bits, descs, K, win, max_m, ap.gamma_onset, 0, sw, scores);
    }
    return launch_check("dp_bits_kernel");
}

int acoss_align_bits_qd_batch(const uint64_t *bits, const acoss_pair_desc *descs, int K, int win, int max_nx,
                              int max_ny, int boundary, const acoss_align_params *params, float *qmax_scores,
                              float *dmax_scores, void *stream)
{
    if (!bits || !descs || !qmax_scores || !dmax_scores || K < 0 || win < 1 || max_nx < win || max_ny < win) {
        set_error("align_bits_qd_batch: bad argument");
        return ACOSS_EINVAL;
    }
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    acoss_align_params ap;
    if (params) ap = *params; else acoss_default_align_params(&ap);
    if (max_n > 2048 || max_m > 2048 || ap.gamma_onset != ap.gamma_extension) {
        set_error("align_bits_qd_batch: needs <= 2048 x 2048 matrices and gamma_onset == gamma_extension");
        return ACOSS_ENOTSUP;
    }
    if (K == 0) return ACOSS_OK;
    // Round 3: since the single-recurrence kernels lost their per-row branches and register copies (a trip of the main loop is
    // 12 unconditional rows), qmax and dmax as two launches (0.69 + 1.24 ms per 4096 pairs of 1000-frame songs) beat the
    // one-sweep kernel (2.02 ms: 230 registers, two waves per SIMD).  ACOSS_DP_ONE_SWEEP=1 keeps the old form.
    static const bool one_sweep = []() { const char *e = getenv("ACOSS_DP_ONE_SWEEP"); return e && e[0] == '1'; }();
    static const bool q16 = []() { const char *e = getenv("ACOSS_DP_Q16"); return !(e && e[0] == '0'); }();
    if (DP_QD16_DEFAULT && !one_sweep && q16 && ap.gamma_onset == 0.5f) {
        // round 4: both recurrences in 16-bit integers in one sweep (dp_bits_qd16_kernel); round 5: also on the 32-word rows of
        // matrices beyond 1024 columns, 24 or 32 columns per lane
        const dim3 grid(ceil_div(K, 4));
        if (mask_bits_words(max_m, max_n) == 16)
            hipLaunchKernelGGL(dp_bits_qd16_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, bits, descs, K, win, max_m, boundary, qmax_scores, dmax_scores);
        else if (max_n <= 1536)
            hipLaunchKernelGGL(dp_bits_qd16_kernel<12>, grid, dim3(256), 0, (hipStream_t)stream, bits, descs, K, win, max_m, boundary, qmax_scores, dmax_scores);
        else
            hipLaunchKernelGGL(dp_bits_qd16_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, bits, descs, K, win, max_m, boundary, qmax_scores, dmax_scores);
        return launch_check("dp_bits_qd16_kernel");
    }
    if (!one_sweep) {
        int rc = acoss_align_bits_batch(0, bits, descs, K, win, max_nx, max_ny, 0, params, qmax_scores, stream);
        if (rc == ACOSS_OK) rc = acoss_align_bits_batch(1, bits, descs, K, win, max_nx, max_ny, boundary, params, dmax_scores, stream);
        return rc;
    }
    if (mask_bits_words(max_m, max_n) == 16)
        hipLaunchKernelGGL(dp_bits_qd_kernel<16>, dim3(ceil_div(K, 4)), dim3(256), 0, (hipStream_t)stream, bits, descs, K, win, max_m,
                           ap.gamma_onset, boundary, qmax_scores, dmax_scores);
    else
        hipLaunchKernelGGL(dp_bits_qd_kernel<32>, dim3(ceil_div(K, 4)), dim3(256), 0, (hipStream_t)stream, bits, descs, K, win, max_m,
                           ap.gamma_onset, boundary, qmax_scores, dmax_scores);
    return launch_check("dp_bits_qd_kernel");
}

int acoss_align_fused_batch(int kind, const double *T, const acoss_pair_desc *descs, int K, int win, int max_nx,
                            int max_ny, int mutual, const void *work, size_t work_bytes, int boundary,
                            const acoss_align_params *params, float *scores, void *stream)
{
    if (!T || !descs || !work || !scores || K < 0 || win < 1 || max_nx < win || max_ny < win || (kind != 0 && kind != 1)) {
        set_error("align_fused_batch: bad argument (kind 0 = qmax, 1 = dmax)");
        return ACOSS_EINVAL;
    }
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    acoss_align_params ap;
    if (params) ap = *params; else acoss_default_align_params(&ap);
    if (max_n > 1024 || ap.gamma_onset != ap.gamma_extension) {
        set_error("align_fused_batch: needs <= 1024 columns and gamma_onset == gamma_extension "
                  "(use acoss_binarize_batch + acoss_qmax_batch / acoss_dmax_batch otherwise)");
        return ACOSS_ENOTSUP;
    }
    if (work_bytes < (size_t)K * (size_t)(max_m + max_n) * 12) { set_error("align_fused_batch: workspace too small"); return ACOSS_EINVAL; }
    if (K == 0) return ACOSS_OK;
    ThreshWork w = thresh_work_layout(const_cast<void *>(work), K, max_m, max_n);
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0)
        hipLaunchKernelGGL(dp_fused_kernel<KIND_QMAX>, dim3(ceil_div(K, 4)), dim3(256), 0, st, T, descs, K, win, w, mutual, ap.gamma_onset, 0, scores);
    else
        hipLaunchKernelGGL(dp_fused_kernel<KIND_DMAX>, dim3(ceil_div(K, 4)), dim3(256), 0, st, T, descs, K, win, w, mutual, ap.gamma_onset, boundary, scores);
    return launch_check("dp_fused_kernel");
}

}  // extern "C"
