// fused_kernels.hip -- the fast forms of the cross-similarity half of the path (gfx950):
//
//   pack_x_kernel        per pair, the x song's frames rotated by the OTI (Serra09.py:167) and
//                        packed one frame per 128-byte line [d values | squared norm | zeros], so
//                        that a kernel that walks x frames as its wave-uniform operand fetches a
//                        whole frame with two scalar loads instead of d+1 rotated ones;
//   csm_packed_kernel    get_csm (CRPUtils.py:67-84) on packed x: same tiling as csm_kernel;
//   crp_kernel           get_csm + sliding_csm fused (CRPUtils.py:67-84 + :24-45): the windowed sums
//                        of squared distances S^2[i][j] = sum_k C[i+k][j+k] (or their sqrt) straight
//                        from the features; the CSM never touches HBM.
#include "common.h"
#include "kernel_utils.h"

#include <type_traits>


namespace acoss {

constexpr int XP_STRIDE = 16;   // elements per packed frame (128 B for float64)

template <typename T>
__global__ __launch_bounds__(256) void pack_x_kernel(const T *__restrict__ feats, const T *__restrict__ norms,
                                                     int d, const acoss_pair_desc *__restrict__ descs,
                                                     int max_nx, int blocks_per_pair, T *__restrict__ xp)
{
    const int p = blockIdx.x / blocks_per_pair;
    const acoss_pair_desc ds = descs[p];
    const int f = (blockIdx.x % blocks_per_pair) * 16 + (threadIdx.x >> 4);
    const int slot = threadIdx.x & 15;
    if (f >= ds.nx) return;
    T v = 0;
    if (slot < d) {
        int src = slot - ds.shift;   // X1[f][b] = X[f][(b - shift) mod d]
        if (src < 0) src += d;
        v = feats[(ds.x_row0 + f) * d + src];
    } else if (slot == d) {
        v = norms[ds.x_row0 + f];
    }
    xp[((int64_t)p * max_nx + f) * XP_STRIDE + slot] = v;
}

// ---------------------------------------------------------------------------------------------
// CSM on packed x.  Identical arithmetic to csm_kernel (same FMA order over the rolled bins, same
// norms), so the two agree bit for bit; only the x-frame fetch differs.
// ---------------------------------------------------------------------------------------------
constexpr int CSM_TM = 128, CSM_TN = 128, CSM_ROWS_PER_WAVE = 32;

template <typename T, int D>
__global__ __launch_bounds__(256) void csm_packed_kernel(const T *__restrict__ xp, int max_nx,
                                                         const T *__restrict__ feats, const T *__restrict__ norms,
                                                         const acoss_pair_desc *__restrict__ descs,
                                                         int tiles_m, int tiles_n, T *__restrict__ out)
{
    static_assert(D < XP_STRIDE, "packed frame holds d values and the norm");
    // the tile's 128 packed x frames: one coalesced 16-byte-per-lane sweep into LDS, then every
    // wave reads the frame of its current row as LDS broadcasts (all lanes, same address)
    __shared__ __attribute__((aligned(16))) T xs[CSM_TM * XP_STRIDE];
    const int tiles = tiles_m * tiles_n;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / tiles, t = lb % tiles;
    const acoss_pair_desc ds = descs[p];
    const int i0 = (t / tiles_n) * CSM_TM, j0 = (t % tiles_n) * CSM_TN;
    if (i0 >= ds.nx || j0 >= ds.ny) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    {
        constexpr int VEC = 16 / sizeof(T);                       // elements per 16-byte load
        constexpr int CHUNKS = CSM_TM * XP_STRIDE / VEC;          // 16-byte chunks in the tile's x block
        const T *xsrc = xp + ((int64_t)p * max_nx + i0) * XP_STRIDE;
        const int valid_chunks = min(CSM_TM, ds.nx - i0) * XP_STRIDE / VEC;
#pragma unroll
        for (int c = threadIdx.x; c < CHUNKS; c += 256) {
            const int cc = min(c, valid_chunks - 1);              // clamp: rows past the song are never used
            reinterpret_cast<uint4 *>(xs)[c] = reinterpret_cast<const uint4 *>(xsrc)[cc];
        }
    }

    const int j = j0 + 2 * lane;
    const bool ok0 = j < ds.ny, ok1 = j + 1 < ds.ny;
    const int ja = min(j, ds.ny - 1), jb = min(j + 1, ds.ny - 1);
    T y0[D], y1[D];
    {
        const T *yp0 = feats + (ds.y_row0 + ja) * D;
        const T *yp1 = feats + (ds.y_row0 + jb) * D;
#pragma unroll
        for (int b = 0; b < D; b++) {
            y0[b] = yp0[b];
            y1[b] = yp1[b];
        }
    }
    const T yy0 = norms[ds.y_row0 + ja];
    const T yy1 = norms[ds.y_row0 + jb];
    __syncthreads();

    T *orow = out + ds.csm_off + (int64_t)j;
    const int r_begin = wave * CSM_ROWS_PER_WAVE;
    const int r_end = min(r_begin + CSM_ROWS_PER_WAVE, ds.nx - i0);
    const bool full = (j0 + CSM_TN <= ds.ny) && ((ds.csm_pitch & 1) == 0) && ((ds.csm_off & 1) == 0);
#pragma unroll 4
    for (int r = r_begin; r < r_end; r++) {
        const T *xr = xs + r * XP_STRIDE;
        T a0 = 0, a1 = 0;
#pragma unroll
        for (int b = 0; b < D; b++) {
            a0 = fma(xr[b], y0[b], a0);
            a1 = fma(xr[b], y1[b], a1);
        }
        const T c0 = csm_sqrt(fma((T)-2, a0, xr[D] + yy0));
        const T c1 = csm_sqrt(fma((T)-2, a1, xr[D] + yy1));
        T *dst = orow + (int64_t)(i0 + r) * ds.csm_pitch;
        if (full) {   // wave-uniform
            if constexpr (sizeof(T) == 8) {
                *reinterpret_cast<double2 *>(dst) = make_double2(c0, c1);
            } else {
                *reinterpret_cast<float2 *>(dst) = make_float2(c0, c1);
            }
        } else {
            if (ok0) dst[0] = c0;
            if (ok1) dst[1] = c1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused CSM + sliding window.
//
// Block = 4 waves.  Phase 1 computes a 32 x 128 tile of squared distances C (clamped at 0) into LDS
// exactly like the CSM kernel does (lane = 2 columns with its y frames in VGPRs, x frame by scalar
// loads from the packed line, 8 rows per wave); phase 2 forms the (32-w+1) x (128-w+1) outputs
// sum_{k<w} C[r+k][c+k] with w lane-contiguous (conflict-free) 8-byte LDS reads each and streams them
// out, 512 contiguous bytes per wave instruction.  For w = 9 a 24 x 120 output tile costs 1.42 x its
// own size in C evaluations; nothing but the result is written to HBM.
//   float64 features: sum of the clamped squared distances (the reference squares the square roots
//                     again, CRPUtils.py:40 -- identical up to one rounding);
//   float32 features: sqrtf, square in float32, promote, sum in float64 (CRPUtils.py:40-41 exactly).
// ---------------------------------------------------------------------------------------------
constexpr int CRP_RT = 32, CRP_CT = 128, CRP_LD = CRP_CT + 2;

template <typename T, int D, int WIN, bool SQRT_OUT>
__global__ __launch_bounds__(256) void crp_kernel(const T *__restrict__ xp, int max_nx,
                                                  const T *__restrict__ feats, const T *__restrict__ norms,
                                                  const acoss_pair_desc *__restrict__ descs, int win_rt,
                                                  int tiles_m, int tiles_n, double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double cl[CRP_RT * CRP_LD];
    __shared__ __attribute__((aligned(16))) T xs[CRP_RT * XP_STRIDE];   // the tile's 32 packed x frames
    const int win = WIN > 0 ? WIN : win_rt;
    const int TM = CRP_RT - (win - 1), TN = CRP_CT - (win - 1);
    const int tiles = tiles_m * tiles_n;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / tiles, t = lb % tiles;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int i0 = (t / tiles_n) * TM, j0 = (t % tiles_n) * TN;
    if (i0 >= M || j0 >= N) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- phase 1: C tile -> LDS
    {
        constexpr int VEC = 16 / sizeof(T);
        constexpr int CHUNKS = CRP_RT * XP_STRIDE / VEC;           // 256 (f64) / 128 (f32) 16-byte chunks
        const T *xsrc = xp + ((int64_t)p * max_nx + i0) * XP_STRIDE;
        const int valid_chunks = min(CRP_RT, ds.nx - i0) * XP_STRIDE / VEC;
        if (threadIdx.x < CHUNKS)
            reinterpret_cast<uint4 *>(xs)[threadIdx.x] =
                reinterpret_cast<const uint4 *>(xsrc)[min((int)threadIdx.x, valid_chunks - 1)];
    }
    {
        const int j = j0 + 2 * lane;
        const int ja = min(j, ds.ny - 1), jb = min(j + 1, ds.ny - 1);   // clamped: out-of-range cells are never read back
        T y0[D], y1[D];
        const T *yp0 = feats + (ds.y_row0 + ja) * D;
        const T *yp1 = feats + (ds.y_row0 + jb) * D;
#pragma unroll
        for (int b = 0; b < D; b++) {
            y0[b] = yp0[b];
            y1[b] = yp1[b];
        }
        const T yy0 = norms[ds.y_row0 + ja], yy1 = norms[ds.y_row0 + jb];
        __syncthreads();
#pragma unroll
        for (int rr = wave * (CRP_RT / 4); rr < (wave + 1) * (CRP_RT / 4); rr++) {
            const T *xr = xs + rr * XP_STRIDE;   // rows past the song hold a clamped copy: never read back
            T a0 = 0, a1 = 0;
#pragma unroll
            for (int b = 0; b < D; b++) {
                a0 = fma(xr[b], y0[b], a0);
                a1 = fma(xr[b], y1[b], a1);
            }
            T c0 = fma((T)-2, a0, xr[D] + yy0);
            T c1 = fma((T)-2, a1, xr[D] + yy1);
            double q0, q1;
            if constexpr (sizeof(T) == 8) {
                q0 = fmax(c0, 0.0);
                q1 = fmax(c1, 0.0);
            } else {
                const float r0 = sqrtf(fmaxf(c0, 0.0f)), r1 = sqrtf(fmaxf(c1, 0.0f));
                q0 = (double)(r0 * r0);
                q1 = (double)(r1 * r1);
            }
            *reinterpret_cast<double2 *>(&cl[rr * CRP_LD + 2 * lane]) = make_double2(q0, q1);
        }
    }
    __syncthreads();
    // ---- phase 2: diagonal window sums
    double *obase = out + ds.crp_off + j0;
    const int ca = lane, cb = lane + 64;
    const bool oka = ca < TN && j0 + ca < N, okb = cb < TN && j0 + cb < N;
    for (int r = wave; r < TM; r += 4) {
        const int gi = i0 + r;
        if (gi >= M) break;
        double sa = 0.0, sb = 0.0;
        if (WIN > 0) {
#pragma unroll
            for (int k = 0; k < (WIN > 0 ? WIN : 1); k++) {
                sa += cl[(r + k) * CRP_LD + ca + k];
                sb += cl[(r + k) * CRP_LD + (okb ? cb : ca) + k];
            }
        } else {
            for (int k = 0; k < win; k++) {
                sa += cl[(r + k) * CRP_LD + ca + k];
                sb += cl[(r + k) * CRP_LD + (okb ? cb : ca) + k];
            }
        }
        if (SQRT_OUT) {
            sa = csm_sqrt(sa);
            sb = csm_sqrt(sb);
        }
        double *orow = obase + (int64_t)gi * ds.crp_pitch;
        if (oka) orow[ca] = sa;
        if (okb) orow[cb] = sb;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused CSM + sliding window, float64, phase 1 on the matrix cores.
//
// The K = d contraction x_i . y_j of the 32 x 128 C tile runs as v_mfma_f64_16x16x4_f64 (the f64
// matrix pipe has the same peak as the f64 VALU but is a separate pipe, so the VALU is left with the
// epilogue and the window sums): each wave owns two 16-column blocks x two 16-row blocks = four
// 16x16 accumulators, ceil(d/4) MFMAs each.  A fragments come from the LDS copy of the packed x
// frames (row stride 18 doubles: conflict-free for the 16-rows x 4-bins fragment read), B fragments
// straight from the y frames in global memory (bins >= d are fed as zeros, which also cancels the
// norm slot of the packed line).  The accumulator layout (lane = column, 4 rows per register set)
// writes to the LDS C tile as four 128-byte row segments per instruction.
// ---------------------------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int XS_LD = 18;

template <int D, int WIN, bool SQRT_OUT>
__global__ __launch_bounds__(256) void crp_mfma_kernel(const double *__restrict__ xp, int max_nx,
                                                       const double *__restrict__ feats, const double *__restrict__ norms,
                                                       const acoss_pair_desc *__restrict__ descs, int win_rt,
                                                       int tiles_m, int tiles_n, double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double cl[CRP_RT * CRP_LD];
    __shared__ __attribute__((aligned(16))) double xs[CRP_RT * XS_LD];
    constexpr int KSTEPS = (D + 3) / 4;
    const int win = WIN > 0 ? WIN : win_rt;
    const int TM = CRP_RT - (win - 1), TN = CRP_CT - (win - 1);
    const int tiles = tiles_m * tiles_n;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / tiles, t = lb % tiles;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int i0 = (t / tiles_n) * TM, j0 = (t % tiles_n) * TN;
    if (i0 >= M || j0 >= N) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- packed x frames of the tile -> LDS (one 16-byte chunk per thread)
    {
        const double *xsrc = xp + ((int64_t)p * max_nx + i0) * XP_STRIDE;
        const int valid_chunks = min(CRP_RT, ds.nx - i0) * (XP_STRIDE / 2);
        const int c = min((int)threadIdx.x, valid_chunks - 1);
        const double2 v = reinterpret_cast<const double2 *>(xsrc)[c];
        *reinterpret_cast<double2 *>(&xs[(threadIdx.x >> 3) * XS_LD + (threadIdx.x & 7) * 2]) = v;
    }
    // ---- B fragments and column norms from the y frames
    const int lr = lane & 15, lk = lane >> 4;
    double bfrag[2][KSTEPS], yy[2];
#pragma unroll
    for (int cbi = 0; cbi < 2; cbi++) {
        const int jc = min(j0 + 16 * (2 * wave + cbi) + lr, ds.ny - 1);
        const double *yp = feats + (ds.y_row0 + jc) * D;
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            const int bin = 4 * s + lk;
            bfrag[cbi][s] = bin < D ? yp[min(bin, D - 1)] : 0.0;
        }
        yy[cbi] = norms[ds.y_row0 + jc];
    }
    __syncthreads();
    // ---- phase 1: C tile by MFMA -> LDS
    {
        v4f64 acc[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int cbi = 0; cbi < 2; cbi++) acc[rb][cbi] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const double a = xs[(16 * rb + lr) * XS_LD + 4 * s + lk];
#pragma unroll
                for (int cbi = 0; cbi < 2; cbi++)
                    acc[rb][cbi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bfrag[cbi][s], acc[rb][cbi], 0, 0, 0);
            }
        }
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * rb + lk + 4 * r;
                const double xx = xs[row * XS_LD + D];
#pragma unroll
                for (int cbi = 0; cbi < 2; cbi++) {
                    const double c = fma(-2.0, acc[rb][cbi][r], xx + yy[cbi]);
                    cl[row * CRP_LD + 16 * (2 * wave + cbi) + lr] = fmax(c, 0.0);
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: diagonal window sums
    double *obase = out + ds.crp_off + j0;
    const int ca = lane, cb = lane + 64;
    const bool oka = ca < TN && j0 + ca < N, okb = cb < TN && j0 + cb < N;
    for (int r = wave; r < TM; r += 4) {
        const int gi = i0 + r;
        if (gi >= M) break;
        double sa = 0.0, sb = 0.0;
        if (WIN > 0) {
#pragma unroll
            for (int k = 0; k < (WIN > 0 ? WIN : 1); k++) {
                sa += cl[(r + k) * CRP_LD + ca + k];
                sb += cl[(r + k) * CRP_LD + (okb ? cb : ca) + k];
            }
        } else {
            for (int k = 0; k < win; k++) {
                sa += cl[(r + k) * CRP_LD + ca + k];
                sb += cl[(r + k) * CRP_LD + (okb ? cb : ca) + k];
            }
        }
        if (SQRT_OUT) {
            sa = csm_sqrt(sa);
            sb = csm_sqrt(sb);
        }
        double *orow = obase + (int64_t)gi * ds.crp_pitch;
        if (oka) orow[ca] = sa;
        if (okb) orow[cb] = sb;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused CSM + sliding window, float64, persistent column strips (the product kernel).
//
// One 8-wave block owns a strip of (128 - w + 1) output columns of one pair and walks DOWN it in
// steps of 32 rows:
//   * its y fragments (wave v = columns 16v..16v+15 of the 128-wide C strip) and column norms are
//     loaded once;
//   * the 32 packed x frames of step t+1 are fetched (one 16-byte chunk per thread) while step t is
//     computed, and dropped into the other half of a double-buffered LDS slot;
//   * step t computes C rows [32t, 32t+32) on the matrix cores into a ring of 32 + w - 1 LDS rows and
//     then writes the window sums of output rows [32t - w + 1, 32t + 33 - w): every C row is computed
//     exactly once per strip (no row halo), and the only global traffic besides the x/y frames is the
//     result, 512 contiguous bytes per wave instruction.
// Two barriers per step separate ring writes from ring reads.
// ---------------------------------------------------------------------------------------------
constexpr int STRIP_ROWS = 32;
// Output columns per strip: the largest multiple of 16 that the 128 C columns allow.  Row pieces of the result then
// start and end on 64-byte boundaries whenever the pitch does (whole 128-byte lines for float64): a pure store kernel
// writes such pieces 25-37 % faster than the 120-column pieces a window of 9 would allow (tools/store_probe.py).
constexpr int strip_tn(int win) { return ((CRP_CT - (win - 1)) / 16) * 16; }
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
constexpr int BUFFER_RSRC_WORD3 = 0x00020000;      // gfx9 raw buffer: 32-bit data format, no swizzle

// MODE (development probes, product = 0; 4 = 1 + 2, 5 = 1 + 3): 1 = no result stores, 2 = no window sums (store a C value),
// 3 = no MFMA phase (stale LDS contents summed)
// CSM_LAYOUT: write at the pair's csm_off / csm_pitch instead of crp_off / crp_pitch (WIN = 1 with
// SQRT_OUT is then exactly get_csm, CRPUtils.py:67-84).
// PLANAR: instead of the float64 sums the kernel writes a uint32 matrix with the same element indexing (crp_off +
// i * crp_pitch + j): the high words of the values' order-preserving keys (wave_ops.h:f64_key; the sums are >= +0.0,
// so that is the upper half of the bit pattern with the sign bit set).  Half the bytes of the float64 form leave the
// chip and the selection kernels read 4 bytes per element; the low words are needed for ~0.1 % of the rows /
// columns only, whose tied elements the fix-up kernel recomputes from the features.
// Cache policy of the result stores (raw buffer store aux bits on gfx950: 2 = nt).  The key-word forms (4 bytes per cell:
// a wave's stores cover lines only partly) gain from non-temporal stores -- 5.79 against 6.10 ms here, 3.3 against 3.7 ms
// in the float32 kernel, per 4096 pairs, same buffer; the float64 form (16 bytes per lane, whole lines) loses: 7.8 against
// 6.7 ms (tools/ab_strip.py, tools/ab_strip32.py).
#define STRIP_STORE_POLICY (PLANAR ? 2 : 0)

template <int D, int WIN, bool SQRT_OUT, int MODE = 0, bool CSM_LAYOUT = false, bool PLANAR = false>
__global__ __launch_bounds__(512) void crp_strip_kernel(const double *__restrict__ xp, int max_nx,
                                                        const double *__restrict__ feats, const double *__restrict__ norms,
                                                        const acoss_pair_desc *__restrict__ descs, int strips,
                                                        double *__restrict__ out)
{
    // C rows of the current step live in rows [HALO, HALO + 32) of `cbuf`; rows [0, HALO) hold the last
    // HALO rows of the previous step (copied through registers across the step boundary), so every LDS
    // address in the window sums is (a per-wave base) + (a compile-time offset): no ring arithmetic.
    constexpr int HALO = WIN - 1;
    constexpr int CROWS = STRIP_ROWS + HALO;
    constexpr int TN = strip_tn(WIN);
    constexpr int KSTEPS = (D + 3) / 4;
    constexpr int ROWS_PER_WAVE = STRIP_ROWS / 8;
    // DIAG: window sums by diagonal runs (see the sum phase); needs ROWS_PER_WAVE spare columns left of the strip
    constexpr bool DIAG = (TN + ROWS_PER_WAVE <= CRP_CT) && (ROWS_PER_WAVE % 2 == 0);
    constexpr int CPAD = 8;       // the diagonal runs of the edge lanes start up to 4 columns outside a row
    __shared__ __attribute__((aligned(16))) double cbuf_raw[CROWS * CRP_LD + 2 * CPAD];
    __shared__ __attribute__((aligned(16))) double xs[STRIP_ROWS * XS_LD];
    double *const cbuf = cbuf_raw + CPAD;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / strips;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - WIN + 1, N = ds.ny - WIN + 1;
    const int j0 = (lb % strips) * TN;
    if (j0 >= N) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7
    const int lr = lane & 15, lk = lane >> 4;
    const int n_steps = (ds.nx + STRIP_ROWS - 1) / STRIP_ROWS;
    const double *xsrc = xp + (int64_t)p * max_nx * XP_STRIDE;
    const int last_chunk = ds.nx * (XP_STRIDE / 2) - 1;                  // 16-byte chunks of valid frames

    // y fragments of this wave's 16 columns (kept for the whole strip)
    double bfrag[KSTEPS], yy;
    {
        const int jc = min(j0 + 16 * wave + lr, ds.ny - 1);
        const double *yp = feats + (ds.y_row0 + jc) * D;
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            const int bin = 4 * s + lk;
            bfrag[s] = bin < D ? yp[min(bin, D - 1)] : 0.0;
        }
        yy = norms[ds.y_row0 + jc];
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) settle(bfrag[s]);
        settle(yy);
    }
    // x frames of step 0 (threads 0..255: one 16-byte chunk each)
    const bool loader = threadIdx.x < STRIP_ROWS * (XP_STRIDE / 2);
    double *xs_dst = &xs[(threadIdx.x >> 3) * XS_LD + (threadIdx.x & 7) * 2];
    if (loader)
        *reinterpret_cast<double2 *>(xs_dst) = reinterpret_cast<const double2 *>(xsrc)[min((int)threadIdx.x, last_chunk)];
    __syncthreads();

    // (the per-lane read / write bases into cbuf, the diagonal-run start column, the halo-copy and x-loader roles are
    // derived inside `step`, see there: rda / rdb / rdd = this wave's ROWS_PER_WAVE output rows at the lane's columns,
    // wr = its 16 C columns, the lane owns the two adjacent diagonals starting at columns dcol, dcol + 1; thread
    // h < HALO * 64 moves two elements of rows [32, 32 + HALO) to rows [0, HALO))
    const int64_t o_off = CSM_LAYOUT ? ds.csm_off : ds.crp_off;
    const int o_pitch = CSM_LAYOUT ? ds.csm_pitch : ds.crp_pitch;
    double *orow = out + o_off + j0 + (int64_t)(wave * ROWS_PER_WAVE - HALO) * o_pitch;
    // The pair's result matrix as a raw buffer resource (diagonal-run form): every store is "wave-uniform byte
    // offset of the row (SGPR) + 32-bit lane offset", so no 64-bit address pairs live in VGPRs, and the hardware
    // range check drops anything outside the matrix.  (Time-neutral against plain pointer stores in a same-buffer
    // A/B -- the strip kernel's time moves by up to 10 % with the placement of its output allocation,
    // tools/align_probe.py -- but 12-16 fewer VGPRs.)
    constexpr int OB = PLANAR ? 4 : 8;          // bytes per output cell
    const int o_rows = CSM_LAYOUT ? ds.nx : M;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(out) + OB * o_off, 0, (int)(OB * (int64_t)o_rows * o_pitch), BUFFER_RSRC_WORD3);
    // 32-bit offsets below: the launchers refuse shapes beyond strip_offsets_fit(); a descriptor whose pitch breaks that
    // bound all the same gets no stores at all rather than stores at wrapped offsets
    if ((int64_t)OB * ((int64_t)o_rows + 2 * STRIP_ROWS) * o_pitch > 0x7fffffffLL) return;
    const int orow0 = (wave * ROWS_PER_WAVE - HALO) * o_pitch + j0;   // element index of (first row of the wave, strip column 0)
    static_assert(!PLANAR || ((TN + ROWS_PER_WAVE <= CRP_CT) && !SQRT_OUT && !CSM_LAYOUT), "planar output: diagonal-run form only");

    // x frames are fetched two steps ahead (registers), so their HBM latency spans a whole step
    double2 xn1 = make_double2(0.0, 0.0), xn2 = make_double2(0.0, 0.0);
    if (loader && n_steps > 1)
        xn1 = reinterpret_cast<const double2 *>(xsrc)[min(STRIP_ROWS * (XP_STRIDE / 2) + (int)threadIdx.x, last_chunk)];

    // One step.  CHECKED = false is the interior form: every output row and column of the step exists,
    // so each wave issues exactly 2 * ROWS_PER_WAVE unconditional stores -- a static count the compiler
    // can put in its vmcnt waits, which keeps the x prefetch and the result stores in flight across
    // steps (with data-dependent store counts it falls back to vmcnt(0): a full HBM round trip per step).
    auto step = [&](const int t, auto checked_tag) {
        // Everything that depends on the thread number is re-derived here from a laundered copy of it: kept live
        // across the loop these per-thread constants (LDS pointers, column numbers, role flags) cost ~12 VGPRs,
        // and the kernel needs <= 80 for a third block per CU; re-deriving them costs ~25 VALU per step.
        int tid_ = threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const int lane_ = tid_ & 63;
        const int lr_ = lane_ & 15, lk_ = lane_ >> 4;
        const bool loader_ = tid_ < STRIP_ROWS * (XP_STRIDE / 2);
        // (24-bit multiplies and unsigned shifts: the laundered thread id is opaque to the compiler, which would emit
        // quarter-rate 32-bit multiplies and signed-division sequences for these)
        const unsigned ut_ = (unsigned)tid_;
        double *xs_dst_ = &xs[__umul24(ut_ >> 3, (unsigned)XS_LD) + (ut_ & 7u) * 2u];
        const int ca_ = lane_, cb_ = lane_ + 64;
        const bool oka_ = ca_ < TN && j0 + ca_ < N, okb_ = cb_ < TN && j0 + cb_ < N;
        const int cbr_ = okb_ ? cb_ : ca_;
        const double *rda_ = cbuf + (wave * ROWS_PER_WAVE) * CRP_LD + ca_;
        const double *rdb_ = cbuf + (wave * ROWS_PER_WAVE) * CRP_LD + cbr_;
        double *wr_ = cbuf + (HALO + lk_) * CRP_LD + 16 * wave + lr_;
        const int dcol_ = 2 * lane_ - ROWS_PER_WAVE;
        const double *rdd_ = cbuf + (wave * ROWS_PER_WAVE) * CRP_LD + dcol_;
        const bool copier_ = tid_ < HALO * (CRP_CT / 2);
        static_assert(CRP_CT == 128, "halo copy: two elements per thread, 64 threads per row");
        const unsigned hrow_ = ut_ >> 6, hcol_ = (2u * ut_) & 127u;
        double *halo_src_ = cbuf + __umul24(copier_ ? STRIP_ROWS + hrow_ : 0u, (unsigned)CRP_LD) + hcol_;
        double *halo_dst_ = cbuf + __umul24(copier_ ? hrow_ : 0u, (unsigned)CRP_LD) + hcol_;
        (void)rda_; (void)rdb_; (void)rdd_; (void)oka_; (void)okb_; (void)cb_;
        constexpr bool CHECKED = decltype(checked_tag)::value;
        const bool more = t + 1 < n_steps;
        if (loader_ && t + 2 < n_steps)
            xn2 = reinterpret_cast<const double2 *>(xsrc)[min((t + 2) * STRIP_ROWS * (XP_STRIDE / 2) + tid_, last_chunk)];
        // ---- C rows [32t, 32t+32) of this wave's 16 columns -> cbuf rows [HALO, HALO+32)
        if (MODE != 3 && MODE != 5) {
            // the norms of the lane's 8 x rows are read together with the A fragments, and |x|^2 + |y|^2 is formed while
            // the matrix-core chain runs: read -> wait -> write once per value would put 8 LDS round trips in a row here
            double nsum[2][4];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
#pragma unroll
                for (int r = 0; r < 4; r++) nsum[rb][r] = xs[(16 * rb + lk_ + 4 * r) * XS_LD + D];
            }
            v4f64 acc[2];
            acc[0] = (v4f64){0.0, 0.0, 0.0, 0.0};
            acc[1] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) {
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
                    const double a = xs[(16 * rb + lr_) * XS_LD + 4 * s + lk_];
                    acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bfrag[s], acc[rb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
#pragma unroll
                for (int r = 0; r < 4; r++) nsum[rb][r] = nsum[rb][r] + yy;
            }
            double cv[2][4];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
#pragma unroll
                for (int r = 0; r < 4; r++) cv[rb][r] = fmax(fma(-2.0, acc[rb][r], nsum[rb][r]), 0.0);
            }
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
#pragma unroll
                for (int r = 0; r < 4; r++) wr_[(16 * rb + 4 * r) * CRP_LD] = cv[rb][r];
            }
        }
        lds_barrier();
        // next step's x frames may now replace the current ones (nobody reads xs until the next step)
        if (loader_ && more) *reinterpret_cast<double2 *>(xs_dst_) = xn1;
        xn1 = xn2;
        // ---- output rows [32t - HALO, 32t + 32 - HALO): ROWS_PER_WAVE per wave, window sums from LDS
        const int g0 = t * STRIP_ROWS - HALO + wave * ROWS_PER_WAVE;          // wave-uniform
        if constexpr (DIAG) {
            // Output (q, c) and (q + 1, c + 1) share WIN - 1 of their WIN addends, so a lane_ walks two adjacent
            // diagonals for ROWS_PER_WAVE rows: ROWS_PER_WAVE + WIN - 1 two-element LDS reads instead of
            // 2 * ROWS_PER_WAVE * WIN single ones, each output still summed in the order k = 0..WIN-1.
            // Row q of the wave then sits at columns dcol_ + q, dcol_ + q + 1: an aligned pair for even q; for
            // odd q the pair is (own second diagonal, first diagonal of the next lane_), one DPP shift.
            double va[ROWS_PER_WAVE + HALO], vb[ROWS_PER_WAVE + HALO];
#pragma unroll
            for (int m = 0; m < ROWS_PER_WAVE + HALO; m++) {
                va[m] = rdd_[m * (CRP_LD + 1)];
                vb[m] = rdd_[m * (CRP_LD + 1) + 1];
            }
#pragma unroll
            for (int q = 0; q < ROWS_PER_WAVE; q++) {
                const int gi = g0 + q;
                // (the C values are never -0.0 -- max(fma(-2, dot, |x|^2 + |y|^2), 0) with non-negative norms -- so starting
                // the chain at the first addend gives the bits of 0.0 + c0 + c1 + ..., the form every other kernel uses)
                double sa = va[q], sb = vb[q];
                if (MODE != 2 && MODE != 4) {
#pragma unroll
                    for (int k = 1; k < WIN; k++) {
                        sa += va[q + k];
                        sb += vb[q + k];
                    }
                }
                if (SQRT_OUT) {
                    sa = csm_sqrt(sa);
                    sb = csm_sqrt(sb);
                }
                const int col = dcol_ + q;
                if constexpr (PLANAR) {
                    const int soff = 4 * (orow0 + (t * STRIP_ROWS + q) * o_pitch);      // wave-uniform byte offset of (row, strip column 0)
                    auto hw = [&](const int c) { return 4 * c; };
                    // only the high words leave the kernel: half the bytes of the float64 form
                    const uint32_t ha = (uint32_t)__double2hiint(sa) | 0x80000000u;
                    const uint32_t hb = (uint32_t)__double2hiint(sb) | 0x80000000u;
                    if (CHECKED) {
                        const bool row_ok = gi >= 0 && gi < M;
                        if (row_ok && col >= 0 && col < TN && j0 + col < N) __builtin_amdgcn_raw_buffer_store_b32(ha, orsrc, hw(col), soff, STRIP_STORE_POLICY);
                        if (row_ok && col + 1 >= 0 && col + 1 < TN && j0 + col + 1 < N) __builtin_amdgcn_raw_buffer_store_b32(hb, orsrc, hw(col + 1), soff, STRIP_STORE_POLICY);
                    } else {
                        // this lane's pair of the row: columns ps, ps + 1 (ps even): one 8-byte store, 512 contiguous bytes per wave
                        uint32_t h0 = ha, h1 = hb;
                        int ps = col;
                        if ((q & 1) != 0) {
                            h0 = hb;
                            h1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ha, 0x130, 0xf, 0xf, true);   // wave_shl:1
                            ps = col + 1;
                        }
                        if (ps >= 0 && ps < TN) __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){h0, h1}, orsrc, hw(ps), soff, STRIP_STORE_POLICY);
                    }
                } else {
                    const int soff = 8 * (orow0 + (t * STRIP_ROWS + q) * o_pitch);      // wave-uniform byte offset of (row, strip column 0)
                    const u32x2_t wa = {(unsigned)__double2loint(sa), (unsigned)__double2hiint(sa)};
                    const u32x2_t wb = {(unsigned)__double2loint(sb), (unsigned)__double2hiint(sb)};
                    if (MODE == 1 || MODE == 4 || MODE == 5) {
                        if (sa == -1.25) __builtin_amdgcn_raw_buffer_store_b64(wb, orsrc, 8 * (col & 63), soff, 0);
                    } else if (CHECKED) {
                        const bool row_ok = gi >= 0 && gi < M;
                        if (row_ok && col >= 0 && col < TN && j0 + col < N) __builtin_amdgcn_raw_buffer_store_b64(wa, orsrc, 8 * col, soff, STRIP_STORE_POLICY);
                        if (row_ok && col + 1 >= 0 && col + 1 < TN && j0 + col + 1 < N) __builtin_amdgcn_raw_buffer_store_b64(wb, orsrc, 8 * col + 8, soff, STRIP_STORE_POLICY);
                    } else if ((q & 1) == 0) {
                        if (col >= 0 && col < TN) __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){wa.x, wa.y, wb.x, wb.y}, orsrc, 8 * col, soff, STRIP_STORE_POLICY);
                    } else {
                        const unsigned glo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)wa.x, 0x130, 0xf, 0xf, true);   // wave_shl:1
                        const unsigned ghi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)wa.y, 0x130, 0xf, 0xf, true);
                        if (col + 1 >= 0 && col + 1 < TN)
                            __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){wb.x, wb.y, glo, ghi}, orsrc, 8 * col + 8, soff, STRIP_STORE_POLICY);
                    }
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < ROWS_PER_WAVE; q++) {
            const int gi = g0 + q;
            if (!CHECKED || (gi >= 0 && gi < M)) {
                double sa = 0.0, sb = 0.0;
                if (MODE == 2) {
                    sa = rda_[q * CRP_LD];
                    sb = rdb_[q * CRP_LD];
                } else {
#pragma unroll
                    for (int k = 0; k < WIN; k++) {
                        sa += rda_[(q + k) * CRP_LD + k];
                        sb += rdb_[(q + k) * CRP_LD + k];
                    }
                }
                if (SQRT_OUT) {
                    sa = csm_sqrt(sa);
                    sb = csm_sqrt(sb);
                }
                double *o = orow + (int64_t)(t * STRIP_ROWS + q) * o_pitch;
                if (MODE == 1) {
                    if (sa == -1.25) o[ca_] = sb;
                } else if (CHECKED) {
                    if (oka_) o[ca_] = sa;
                    if (okb_) o[cb_] = sb;
                } else {
                    // pair up adjacent columns across neighbouring lanes (one DPP swap) so that every lane_
                    // issues ONE 16-byte store: even lanes write columns (l, l+1), odd lanes (63+l, 64+l)
                    const bool odd = lane_ & 1;
                    const double give = odd ? sa : sb;
                    const int glo = __builtin_amdgcn_update_dpp(0, __double2loint(give), 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                    const int ghi = __builtin_amdgcn_update_dpp(0, __double2hiint(give), 0xB1, 0xf, 0xf, true);
                    const double got = __hiloint2double(ghi, glo);
                    const double2 v = odd ? make_double2(got, sb) : make_double2(sa, got);
                    const int col = odd ? 63 + lane_ : lane_;
                    if (col + 1 < TN) *reinterpret_cast<double2 *>(o + col) = v;
                }
            }
        }
        }
        // carry the last HALO C rows over to the next step: read before the barrier, write after it
        double2 hv = make_double2(0.0, 0.0);
        if (copier_) hv = *reinterpret_cast<const double2 *>(halo_src_);
        lds_barrier();
        if (copier_) *reinterpret_cast<double2 *>(halo_dst_) = hv;
    };
    const bool full_strip = (j0 + TN <= N) && ((o_pitch & 1) == 0) && ((o_off & 1) == 0);   // block-uniform
    step(0, std::true_type{});
    if (full_strip) {
        for (int t = 1; t < n_steps - 1; t++) step(t, std::false_type{});
    } else {
        for (int t = 1; t < n_steps - 1; t++) step(t, std::true_type{});
    }
    if (n_steps > 1) step(n_steps - 1, std::true_type{});
}

template <typename T>
static int launch_pack(const T *feats, const T *norms, int d, const acoss_pair_desc *descs, int K, int max_nx,
                       T *xp, hipStream_t st)
{
    if (!feats || !norms || !descs || !xp || K < 0 || d < 1 || d >= XP_STRIDE || max_nx < 1) {
        set_error("pack_x: bad argument (1 <= d <= 15)");
        return ACOSS_EINVAL;
    }
    if (K == 0) return ACOSS_OK;
    const int bpp = ceil_div(max_nx, 16);
    const int64_t blocks = (int64_t)K * bpp;
    if (blocks > 0x7fffffffLL) { set_error("pack_x: batch too large"); return ACOSS_ENOTSUP; }
    hipLaunchKernelGGL(pack_x_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, feats, norms, d, descs, max_nx, bpp, xp);
    return launch_check("pack_x_kernel");
}

template <typename T>
static int launch_csm_packed(const T *xp, const T *feats, const T *norms, int d, const acoss_pair_desc *descs,
                             int K, int max_nx, int max_ny, T *csm, hipStream_t st)
{
    if (!xp || !feats || !norms || !descs || !csm || K < 0 || max_nx < 1 || max_ny < 1) {
        set_error("csm_packed_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if (d != 12 && d != 13) { set_error("csm_packed_batch: d must be 12 or 13 (use acoss_csm_batch for other sizes)"); return ACOSS_ENOTSUP; }
    if (K == 0) return ACOSS_OK;
    const int tm = ceil_div(max_nx, CSM_TM), tn = ceil_div(max_ny, CSM_TN);
    const int64_t blocks = (int64_t)K * tm * tn;
    if (blocks > 0x7fffffffLL) { set_error("csm_packed_batch: batch too large"); return ACOSS_ENOTSUP; }
    if (d == 12)
        hipLaunchKernelGGL((csm_packed_kernel<T, 12>), dim3((unsigned)blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, tm, tn, csm);
    else
        hipLaunchKernelGGL((csm_packed_kernel<T, 13>), dim3((unsigned)blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, tm, tn, csm);
    return launch_check("csm_packed_kernel");
}

template <int D, int WIN>
static void launch_crp_strip(const double *xp, int max_nx, const double *feats, const double *norms,
                             const acoss_pair_desc *descs, int K, int max_ny, int sqrt_out, double *out, hipStream_t st)
{
    const int strips = ceil_div(max_ny - WIN + 1, strip_tn(WIN));
    const unsigned blocks = (unsigned)((int64_t)K * strips);
    if (sqrt_out) hipLaunchKernelGGL((crp_strip_kernel<D, WIN, true>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, out);
    else hipLaunchKernelGGL((crp_strip_kernel<D, WIN, false>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, out);
}

template <int D>
static void launch_crp_strip_planar(const double *xp, int max_nx, const double *feats, const double *norms,
                                    const acoss_pair_desc *descs, int K, int max_ny, uint32_t *planes, hipStream_t st)
{
    const int strips = ceil_div(max_ny - 9 + 1, strip_tn(9));
    const unsigned blocks = (unsigned)((int64_t)K * strips);
    hipLaunchKernelGGL((crp_strip_kernel<D, 9, false, 0, false, true>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms,
                       descs, strips, reinterpret_cast<double *>(planes));
}

template <int D>
static void launch_crp_mfma(const double *xp, int max_nx, const double *feats, const double *norms,
                            const acoss_pair_desc *descs, int win, int tm, int tn, unsigned blocks, int sqrt_out,
                            double *out, hipStream_t st)
{
    if (win == 9) {
        if (sqrt_out) hipLaunchKernelGGL((crp_mfma_kernel<D, 9, true>), dim3(blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, win, tm, tn, out);
        else hipLaunchKernelGGL((crp_mfma_kernel<D, 9, false>), dim3(blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, win, tm, tn, out);
    } else {
        if (sqrt_out) hipLaunchKernelGGL((crp_mfma_kernel<D, 0, true>), dim3(blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, win, tm, tn, out);
        else hipLaunchKernelGGL((crp_mfma_kernel<D, 0, false>), dim3(blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, win, tm, tn, out);
    }
}

template <typename T, int D>
static void launch_crp_d(const T *xp, int max_nx, const T *feats, const T *norms, const acoss_pair_desc *descs,
                         int win, int tm, int tn, unsigned blocks, int sqrt_out, double *out, hipStream_t st)
{
    // flags: bit 0 = write sqrt of the sums, bit 1 = force the VALU form of the float64 kernel
    const bool force_valu = (sqrt_out & 2) != 0;
    sqrt_out &= 1;
    if constexpr (sizeof(T) == 8) {
        if (!force_valu) {
            launch_crp_mfma<D>(xp, max_nx, feats, norms, descs, win, tm, tn, blocks, sqrt_out, out, st);
            return;
        }
    }
    if (win == 9) {
        if (sqrt_out) hipLaunchKernelGGL((crp_kernel<T, D, 9, true>), dim3(blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, win, tm, tn, out);
        else hipLaunchKernelGGL((crp_kernel<T, D, 9, false>), dim3(blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, win, tm, tn, out);
    } else {
        if (sqrt_out) hipLaunchKernelGGL((crp_kernel<T, D, 0, true>), dim3(blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, win, tm, tn, out);
        else hipLaunchKernelGGL((crp_kernel<T, D, 0, false>), dim3(blocks), dim3(256), 0, st, xp, max_nx, feats, norms, descs, win, tm, tn, out);
    }
}

template <typename T>
static int launch_crp(const T *xp, const T *feats, const T *norms, int d, const acoss_pair_desc *descs, int K,
                      int win, int max_nx, int max_ny, int sqrt_out, double *out, hipStream_t st)
{
    if (!xp || !feats || !norms || !descs || !out || K < 0 || win < 1 || max_nx < win || max_ny < win) {
        set_error("crp_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if ((d != 12 && d != 13) || win > 16) {
        set_error("crp_batch: supports d in {12, 13} and win <= 16 (use csm_batch + sliding_batch otherwise)");
        return ACOSS_ENOTSUP;
    }
    if (K == 0) return ACOSS_OK;
    const int TM = CRP_RT - (win - 1), TN = CRP_CT - (win - 1);
    const int tm = ceil_div(max_nx - win + 1, TM), tn = ceil_div(max_ny - win + 1, TN);
    const int64_t blocks = (int64_t)K * tm * tn;
    if (blocks > 0x7fffffffLL) { set_error("crp_batch: batch too large"); return ACOSS_ENOTSUP; }
    if (!strip_offsets_fit(max_nx, max_ny, 8)) {
        set_error("crp_batch: a pair's result matrix must stay below 2 GiB (32-bit store offsets)");
        return ACOSS_ENOTSUP;
    }
    // flags: bit 0 = sqrt, bit 1 = all-VALU tile kernel, bit 2 = matrix-core tile kernel;
    // default for float64 and the window of the paper (m = 9): the persistent strip kernel
    if constexpr (sizeof(T) == 8) {
        if (win == 9 && (sqrt_out & 6) == 0) {
            if (d == 12) launch_crp_strip<12, 9>(xp, max_nx, feats, norms, descs, K, max_ny, sqrt_out & 1, out, st);
            else launch_crp_strip<13, 9>(xp, max_nx, feats, norms, descs, K, max_ny, sqrt_out & 1, out, st);
            return launch_check("crp_strip_kernel");
        }
    }
    sqrt_out &= 3;
    if (d == 12) launch_crp_d<T, 12>(xp, max_nx, feats, norms, descs, win, tm, tn, (unsigned)blocks, sqrt_out, out, st);
    else launch_crp_d<T, 13>(xp, max_nx, feats, norms, descs, win, tm, tn, (unsigned)blocks, sqrt_out, out, st);
    return launch_check("crp_kernel");
}

}  // namespace acoss

using namespace acoss;

extern "C" {

#ifdef ACOSS_PROBES      // python -m acoss_amd.build --probes
// development probe (not part of the public ABI): strip kernel in a probe MODE
int acoss_dev_crp_probe(int mode, const double *xp, const double *feats, const double *norms,
                        const acoss_pair_desc *descs, int K, int max_nx, int max_ny, double *out, void *stream)
{
    const int strips = ceil_div(max_ny - 9 + 1, strip_tn(9));
    const unsigned blocks = (unsigned)((int64_t)K * strips);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1) hipLaunchKernelGGL((crp_strip_kernel<12, 9, false, 1>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, out);
    else if (mode == 2) hipLaunchKernelGGL((crp_strip_kernel<12, 9, false, 2>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, out);
    else if (mode == 3) hipLaunchKernelGGL((crp_strip_kernel<12, 9, false, 3>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, out);
    else if (mode == 4) hipLaunchKernelGGL((crp_strip_kernel<12, 9, false, 4>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, out);
    else if (mode == 5) hipLaunchKernelGGL((crp_strip_kernel<12, 9, false, 5>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, out);
    else hipLaunchKernelGGL((crp_strip_kernel<12, 9, false, 0>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, out);
    return launch_check("crp_strip probe");
}
#endif  // ACOSS_PROBES

int64_t acoss_xpack_elems(int K, int max_nx)
{
    return (int64_t)(K > 0 ? K : 0) * (int64_t)(max_nx > 0 ? max_nx : 0) * XP_STRIDE;
}

int acoss_pack_x_f64(const double *feats, const double *norms, int d, const acoss_pair_desc *descs, int K,
                     int max_nx, double *xp, void *stream)
{
    return launch_pack<double>(feats, norms, d, descs, K, max_nx, xp, (hipStream_t)stream);
}
int acoss_pack_x_f32(const float *feats, const float *norms, int d, const acoss_pair_desc *descs, int K,
                     int max_nx, float *xp, void *stream)
{
    return launch_pack<float>(feats, norms, d, descs, K, max_nx, xp, (hipStream_t)stream);
}

int acoss_csm_packed_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                               const acoss_pair_desc *descs, int K, int max_nx, int max_ny, double *csm, void *stream)
{
    return launch_csm_packed<double>(xp, feats, norms, d, descs, K, max_nx, max_ny, csm, (hipStream_t)stream);
}
int acoss_csm_packed_batch_f32(const float *xp, const float *feats, const float *norms, int d,
                               const acoss_pair_desc *descs, int K, int max_nx, int max_ny, float *csm, void *stream)
{
    return launch_csm_packed<float>(xp, feats, norms, d, descs, K, max_nx, max_ny, csm, (hipStream_t)stream);
}

// get_csm through the persistent strip kernel (window 1, sqrt): the fast float64 CSM
int acoss_csm_strip_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                              const acoss_pair_desc *descs, int K, int max_nx, int max_ny, double *csm, void *stream)
{
    if (!xp || !feats || !norms || !descs || !csm || K < 0 || max_nx < 1 || max_ny < 1) {
        set_error("csm_strip_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if (d != 12 && d != 13) { set_error("csm_strip_batch: d must be 12 or 13"); return ACOSS_ENOTSUP; }
    if (!strip_offsets_fit(max_nx, max_ny, 8)) {
        set_error("csm_strip_batch: a pair's matrix must stay below 2 GiB (32-bit store offsets)");
        return ACOSS_ENOTSUP;
    }
    if (K == 0) return ACOSS_OK;
    const int strips = ceil_div(max_ny, CRP_CT);
    const unsigned blocks = (unsigned)((int64_t)K * strips);
    hipStream_t st = (hipStream_t)stream;
    if (d == 12) hipLaunchKernelGGL((crp_strip_kernel<12, 1, true, 0, true>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, csm);
    else hipLaunchKernelGGL((crp_strip_kernel<13, 1, true, 0, true>), dim3(blocks), dim3(512), 0, st, xp, max_nx, feats, norms, descs, strips, csm);
    return launch_check("crp_strip_kernel (csm)");
}

int acoss_crp_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                        const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, int sqrt_out,
                        double *out, void *stream)
{
    return launch_crp<double>(xp, feats, norms, d, descs, K, win, max_nx, max_ny, sqrt_out, out, (hipStream_t)stream);
}
int acoss_crp_planar_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                               const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                               uint32_t *planes, void *stream)
{
    if (!xp || !feats || !norms || !descs || !planes || K < 0 || max_nx < win || max_ny < win) {
        set_error("crp_planar_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if ((d != 12 && d != 13) || win != 9) {
        set_error("crp_planar_batch: supports d in {12, 13} and win == 9 (use crp_batch otherwise)");
        return ACOSS_ENOTSUP;
    }
    if (K == 0) return ACOSS_OK;
    if ((int64_t)K * ceil_div(max_ny - win + 1, strip_tn(win)) > 0x7fffffffLL || !strip_offsets_fit(max_nx, max_ny, 4)) {
        set_error("crp_planar_batch: batch too large");
        return ACOSS_ENOTSUP;
    }
    if (d == 12) launch_crp_strip_planar<12>(xp, max_nx, feats, norms, descs, K, max_ny, planes, (hipStream_t)stream);
    else launch_crp_strip_planar<13>(xp, max_nx, feats, norms, descs, K, max_ny, planes, (hipStream_t)stream);
    return launch_check("crp_strip_kernel<planar>");
}
int acoss_crp_batch_f32(const float *xp, const float *feats, const float *norms, int d,
                        const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, int sqrt_out,
                        double *out, void *stream)
{
    return launch_crp<float>(xp, feats, norms, d, descs, K, win, max_nx, max_ny, sqrt_out, out, (hipStream_t)stream);
}

}  // extern "C"
