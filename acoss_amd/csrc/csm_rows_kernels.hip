// csm_rows_kernels.hip -- get_csm (CRPUtils.py:67-84) as a row-band kernel (round 4): the kernel the north star grades
// against the HBM roofline (8 bytes written per cell, nothing else of note).
//
// x stationary, y streaming, and NO shared memory: a block of four waves owns a band of 32 rows of one pair (two matrix-core
// row tiles; every wave holds the band's x frames in registers as ready-made A operands for the whole walk) and steps RIGHT
// over the columns, 128 per step; wave w takes the w-th piece of 32 columns (two column tiles, interleaved: lane lr's B
// operands are y frames 2 lr and 2 lr + 1 of the piece).  The accumulators of
// v_mfma_f64_16x16x4_f64 then hold, per lane, two ADJACENT cells of four rows per row tile (rows lk, 4 + lk, 8 + lk, 12 + lk): the epilogue
// sqrt(max(fma(-2, x.y, |x|^2 + |y|^2), 0)) runs in that layout and each result pair leaves with one 16-byte store -- a wave
// instruction writes four rows x 256 contiguous, line-aligned bytes (whole 128-byte lines: no read-modify-write in the L2),
// non-temporal.  No LDS round trip for the result, no barrier anywhere: waves are independent.
//
// Why this form: as a pure store pattern (tools/ubench/csm_store.hip, 4096 pairs, no arithmetic) it reaches 6.3 TB/s where
// the column strips of crp_strip_kernel<D,1,sqrt> (1 KB row pieces at an 8 KB pitch, walking down) reach 5.25 (5.8 with
// non-temporal stores); the strip kernel itself ran at 4.76.
//
// Arithmetic contract: dot = the k-ordered FMA chain over the bins of the rolled x frame (a float64
// matrix-core instruction IS that chain), C = fma(-2, dot, |x|^2 + |y|^2), csm_sqrt: bit for bit csm_kernel's values
// (tests/test_gpu_fast_path.py compares whole matrices).
#include "common.h"
#include "kernel_utils.h"

#include <type_traits>

namespace acoss {

typedef double v4f64r __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2r __attribute__((ext_vector_type(2)));
constexpr int CR_BR = 32;            // rows per band: two row tiles
constexpr int CR_CW = 32;            // columns per wave and step: two column tiles
constexpr int CR_WAVES = 4;          // waves per block (neighbouring 32-column pieces of the same band)
constexpr int CR_XP = 16;            // elements per packed x frame (pack_x_kernel: d bins, the squared norm, zeros)
constexpr int CR_RSRC_WORD3 = 0x00020000;
#ifndef CR_STORE_POLICY
#define CR_STORE_POLICY 2            // nt
#endif
#ifndef CR_WPS
#define CR_WPS 4
#endif
#ifndef CR_PROBE
#define CR_PROBE 0                   // development (tools/build_variant.py), wrong values, timing only: bit 0 = no square roots, bit 1 = no matrix
#endif                               // instructions, bit 2 = no y loads inside the walk (the first step's operands again), bit 3 = no stores

template <int D>
__global__ __launch_bounds__(64 * CR_WAVES, D <= 12 ? CR_WPS : CR_WPS - 1) void csm_rows_kernel(const double *__restrict__ xp, int max_nx,
                                                                        const double *__restrict__ feats, const double *__restrict__ norms,
                                                                        const acoss_pair_desc *__restrict__ descs, int bands,
                                                                        double *__restrict__ out)
{
    constexpr int KSTEPS = (D + 3) / 4;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / bands;
    const acoss_pair_desc ds = descs[p];
    const int R0 = (lb % bands) * CR_BR;
    if (R0 >= ds.nx) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    if ((int64_t)8 * ((int64_t)ds.nx + CR_BR) * ds.csm_pitch > 0x7fffffffLL) return;        // 32-bit offsets (strip_offsets_fit)
    // A operands of the band's two row tiles (row 16 u + lr, bins 4 s + lk) and the squared norms of the rows this lane's
    // accumulators hold (v_mfma_f64_16x16x4_f64: element r of lane (lr, lk) is row 4 r + lk, column lr of the tile)
    const double *xsrc = xp + (int64_t)p * max_nx * CR_XP;
    double a[2][KSTEPS], xn[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const double *xr = xsrc + (int64_t)min(R0 + 16 * u + lr, ds.nx - 1) * CR_XP;
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            const int bin = 4 * s + lk;
            a[u][s] = bin < D ? xr[min(bin, D - 1)] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) xn[u][r] = xsrc[(int64_t)min(R0 + 16 * u + 4 * r + lk, ds.nx - 1) * CR_XP + D];
    }
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(out) + 8 * ds.csm_off, 0, (int)(8 * (int64_t)ds.nx * ds.csm_pitch), CR_RSRC_WORD3);
    const bool even_layout = ((ds.csm_pitch & 1) == 0) && ((ds.csm_off & 1) == 0);           // block-uniform: 16-byte stores
    const int row_off0 = 8 * ((R0 + lk) * ds.csm_pitch);                                     // row lk of the band, bytes
    const int n_steps = (ds.ny + CR_WAVES * CR_CW - 1) / (CR_WAVES * CR_CW);
    // y fragments of step t for this wave's 32 columns: column tile v = frames c0 + 2 lr + v
    auto load_y = [&](const int t, double (&b)[2][KSTEPS], double (&yn)[2]) {
        const int c0 = (t * CR_WAVES + wave) * CR_CW + 2 * lr;
#pragma unroll
        for (int v = 0; v < 2; v++) {
            const int64_t j = ds.y_row0 + min(c0 + v, ds.ny - 1);
            const double *yp = feats + j * D;
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) {
                const int bin = 4 * s + lk;
                b[v][s] = bin < D ? yp[min(bin, D - 1)] : 0.0;
            }
            yn[v] = norms[j];
        }
    };
    // The y operands of step t + 1 are requested right behind step t's matrix instructions, into the registers those have
    // just read: they travel while the epilogue runs (no second operand set: the kernel needs every register for four
    // waves per SIMD), only the two norms are double-buffered (the epilogue still reads the current ones).
    double bcur[2][KSTEPS], ycur[2], ynext[2] = {0.0, 0.0};
    load_y(0, bcur, ycur);
    for (int t = 0; t < n_steps; t++) {
        const int c0 = (t * CR_WAVES + wave) * CR_CW + 2 * lr;          // this lane's first column
        // (laundered: kept as loop invariants the eight per-row store offsets cost eight registers the kernel does not have;
        // recomputed, one addition of a scalar per store)
        int cell_off = row_off0 + 8 * c0;
        asm volatile("" : "+v"(cell_off));
        v4f64r acc[2][2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
#pragma unroll
            for (int v = 0; v < 2; v++) acc[u][v] = (v4f64r){0.0, 0.0, 0.0, 0.0};
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
#pragma unroll
                for (int v = 0; v < 2; v++) {
                    if (CR_PROBE & 2) acc[u][v][s & 3] += a[u][s] + bcur[v][s];
                    else acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][s], bcur[v][s], acc[u][v], 0, 0, 0);
                }
            }
        }
        if (!(CR_PROBE & 4) && t + 1 < n_steps) load_y(t + 1, bcur, ynext);
        // epilogue in the accumulator layout: cv[u][r][v] = cell (row 16 u + 4 r + lk, column c0 + v)
        // (the test for values the fast square root cannot take runs on the HIGH WORDS, full-rate integer minima instead of
        // two float64 compares per value: a value >= +0 lies below 2^-900 iff its high word is below 0x07B00000 -- which
        // sends exact zeros to the general form too, where they take the same iteration as here)
        double cv[2][4][2];
        unsigned min_hi = 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 2; u++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int v = 0; v < 2; v++) {
                    cv[u][r][v] = fmax(fma(-2.0, acc[u][v][r], xn[u][r] + ycur[v]), 0.0);
                    min_hi = min(min_hi, (unsigned)__double2hiint(cv[u][r][v]));
                }
            }
        }
        const bool tiny = min_hi < 0x07B00000u;
        // stores: four rows x 256 contiguous bytes per wave instruction.  Rows past the end of the song fall outside the
        // buffer resource (the hardware drops them); a piece that reaches past the last column, or a pair whose matrix
        // does not start on a 16-byte boundary, takes the checked 8-byte form.
        const bool whole = even_layout && (t * CR_WAVES + wave + 1) * CR_CW <= ds.ny;        // wave-uniform
        // COMMON = the piece lies inside the row, 16-byte stores, every value >= 2^-900 (one wave-uniform test; exact zeros --
        // identical frames -- and values the iteration cannot take go the other way); the other form checks every store and scales tiny values before the iteration
        auto finish = [&](auto common_tag) {
            constexpr bool COMMON = decltype(common_tag)::value;
#pragma unroll
            for (int u = 0; u < 2; u++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double c[2];
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        c[v] = cv[u][r][v];
                        if (CR_PROBE & 1) {
                        } else if (COMMON) {
                            c[v] = csm_sqrt_fast<true>(c[v]);
                        } else {
                            // below 2^-900 the iteration's intermediates leave the normal range: take the root of c 2^200 (exact
                            // scaling by an even power of two) and scale back -- what a correctly rounded sqrt returns
                            const bool tn = csm_sqrt_is_tiny(c[v]);
                            c[v] = csm_sqrt_fast(tn ? c[v] * 0x1.0p200 : c[v]) * (tn ? 0x1.0p-100 : 1.0);
                        }
                    }
                    const int soff = cell_off + 8 * ((16 * u + 4 * r) * ds.csm_pitch);   // per lane (row 16 u + 4 r + lk): a vector offset
                    const u32x2r w0 = {(unsigned)__double2loint(c[0]), (unsigned)__double2hiint(c[0])};
                    const u32x2r w1 = {(unsigned)__double2loint(c[1]), (unsigned)__double2hiint(c[1])};
                    if (CR_PROBE & 8) {
                        if (c[0] == -1.25) __builtin_amdgcn_raw_buffer_store_b64(w1, orsrc, soff, 0, CR_STORE_POLICY);
                    } else if (COMMON) {
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4r){w0.x, w0.y, w1.x, w1.y}, orsrc, soff, 0, CR_STORE_POLICY);
                        // (left alone the scheduler interleaves all sixteen root iterations and spills)
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        if (c0 < ds.ny) __builtin_amdgcn_raw_buffer_store_b64(w0, orsrc, soff, 0, CR_STORE_POLICY);
                        if (c0 + 1 < ds.ny) __builtin_amdgcn_raw_buffer_store_b64(w1, orsrc, soff + 8, 0, CR_STORE_POLICY);
                    }
                }
            }
        };
        if (__builtin_expect(whole && !__any(tiny), 1)) finish(std::true_type{});
        else finish(std::false_type{});
        ycur[0] = ynext[0];
        ycur[1] = ynext[1];
    }
}

}  // namespace acoss

using namespace acoss;

// get_csm (CRPUtils.py:67-84) for a batch of pairs, float64, row-band form; same arguments as acoss_csm_strip_batch_f64
extern "C" int acoss_csm_rows_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                                        const acoss_pair_desc *descs, int K, int max_nx, int max_ny, double *csm, void *stream)
{
    if (!xp || !feats || !norms || !descs || !csm || K < 0 || max_nx < 1 || max_ny < 1) {
        set_error("csm_rows_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if (d != 12 && d != 13) { set_error("csm_rows_batch: d must be 12 or 13"); return ACOSS_ENOTSUP; }
    if (!strip_offsets_fit(max_nx, max_ny, 8)) {
        set_error("csm_rows_batch: a pair's matrix must stay below 2 GiB (32-bit store offsets)");
        return ACOSS_ENOTSUP;
    }
    if (K == 0) return ACOSS_OK;
    const int bands = ceil_div(max_nx, CR_BR);
    if ((int64_t)K * bands > 0x7fffffffLL) { set_error("csm_rows_batch: batch too large"); return ACOSS_ENOTSUP; }
    const unsigned blocks = (unsigned)((int64_t)K * bands);
    hipStream_t st = (hipStream_t)stream;
    if (d == 12) hipLaunchKernelGGL(csm_rows_kernel<12>, dim3(blocks), dim3(64 * CR_WAVES), 0, st, xp, max_nx, feats, norms, descs, bands, csm);
    else hipLaunchKernelGGL(csm_rows_kernel<13>, dim3(blocks), dim3(64 * CR_WAVES), 0, st, xp, max_nx, feats, norms, descs, bands, csm);
    return launch_check("csm_rows_kernel");
}
