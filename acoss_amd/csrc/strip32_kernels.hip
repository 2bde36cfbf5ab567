// strip32_kernels.hip -- cross-similarity + sliding window in float32: the FILTER in front of the kNN selection
// (CRPUtils.py:67-84 + :24-45 fused; crp_rows32_kernel, the dominant kernel of the product path).
//
// Why float32: the masks need the windowed sums T exactly only where they are compared against a row's / column's k-th
// smallest value; everywhere else an approximation with a known error bound decides the same way.  This kernel produces that
// approximation -- T~ on v_mfma_f32_16x16x4_f32 (a k-ordered chain of round-to-nearest FMAs: pinned by
// tests/test_gpu_fast_path.py against a host emulation) with the window sums as packed float32 adds -- and writes it either
// as 16-bit keys (keys16.h: the product path, 2 bytes per cell) or as order-preserving uint32 keys (4 bytes per cell: the
// bit pattern of a float32 >= +0 with the sign bit set).  Rows and columns in which another value lies within the error
// band of the k-th smallest are finished in float64 by the refinement kernels, so the masks equal the float64 path's bit for bit.
//
// Error bound used by the callers: |T~ - T| <= 2^-24 * ((d + 4.5) * sum_k (|x_{i+k}|^2 + |y_{j+k}|^2) + 9.5 T)
// for the operands it is given (the host passes the corpus with its mean subtracted: same distances, a third of the norms).
//
// (Round 2's column-strip form of this kernel -- a block per 112-column strip walking down, 448-byte row pieces -- was removed
// in round 4: it cannot write 2-byte cells efficiently and its 4-byte form depended on the placement of the output buffer.)
#include "common.h"
#include "kernel_utils.h"

#include <stdlib.h>
#include <type_traits>

namespace acoss {

typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef float v2f32 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2s_t __attribute__((ext_vector_type(2)));

constexpr int S32_WIN = 9, S32_HALO = S32_WIN - 1;
constexpr int S32_XP = 16;              // packed x line: 16 floats (d bins, the squared norm, zeros)
constexpr int S32_CPAD = 8;
constexpr int S32_RSRC_WORD3 = 0x00020000;

// cache policy of the key stores (raw buffer store aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1): the matrix is written
// once and read back from HBM long after it has left every cache
#ifndef S32_STORE_POLICY
#define S32_STORE_POLICY 2
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Row bands (round 3): x stationary, y streaming.  A block owns a BAND of 56 output rows
// of one pair (64 C rows: the window's 8-row halo is recomputed per band, the same 14 % the column strips recompute) and
// walks RIGHT over the columns in chunks of 128 C columns; the last 15 columns of a chunk are carried to the next one
// inside LDS.  Why: a column-strip kernel stores 448-byte row pieces at a 4 KB pitch (5.1 TB/s as a pure store pattern,
// tools/ubench/plane_layout.hip); here every wave instruction stores 512 contiguous, line-aligned bytes of ONE row and the
// next chunk continues that row (5.8 TB/s), and -- what the column strips cannot do at all without halving their store
// rate -- the same rows can leave as two 16-bit planes of 256-byte pieces (5.6 TB/s) for the selection kernels to read
// at 2 bytes per cell.
//
// Per chunk: C[64][128] on the matrix cores into LDS (wave w: columns 16w .. 16w+15, four 16-row tiles; the y fragments
// of the next chunk are already in flight), LDS-only barrier, then wave w forms rows 7w .. 7w+6 of the band: a lane walks
// two adjacent diagonals down 7 rows (15 single 8-byte LDS reads for 14 outputs); the seven window sums share their partial
// sums (24 packed additions; window_sum9() of kernel_utils.h defines the association; tests/test_gpu_fast_path.py pins every
// key against the host emulation of that chain).  Walking diagonals, the columns a lane holds drift by one per row, so row q
// of a chunk covers columns
// [A + q, A + q + 128) with A = 128 t - 15 (+ 1 in odd waves: 8-byte alignment of the diagonal reads).  Stores must be
// line-aligned (the same pieces shifted by a few cells store at half the rate), so each row is brought into place in
// registers: the aligned block [128 (t-1), 128 t) of a row is the tail of what the previous chunk produced (kept in two
// registers per row) followed by the first few lanes of the current chunk's -- one select and one ds_bpermute per
// value -- and is stored one chunk late; a last pass without arithmetic flushes the final block.
constexpr int R32_BR = 56;              // output rows per band
constexpr int R32_CR = R32_BR + S32_HALO;     // C rows per band (64 = four matrix-core tiles)
constexpr int R32_CW = 128;             // C columns per chunk
constexpr int R32_CARRY = 15;           // C columns carried from chunk to chunk
constexpr int R32_LD = 147;             // odd (the diagonal stride LD + 1 keeps 8-byte alignment); >= CARRY + CW
constexpr int R32_RPW = R32_BR / 8;     // output rows per wave
#ifndef R32_WPS
#define R32_WPS 6                       // waves per SIMD the register allocation aims at: three blocks per CU (42 KB of LDS each)
#endif

// OUT: 0 = uint32 keys; 1 = the 16-bit key plane of keys16.h (2 bytes per cell, two
// resolutions in one monotone map: see there; `out` is then a uint16 matrix with the same element indexing);
// 2 = no stores (development probe)
template <int D, int OUT = 0>
__global__ __launch_bounds__(512, R32_WPS) void crp_rows32_kernel(const float *__restrict__ xp, int max_nx,
                                                         const float *__restrict__ feats, const float *__restrict__ norms,
                                                         const acoss_pair_desc *__restrict__ descs, int bands,
                                                         uint32_t *__restrict__ out, const uint32_t *__restrict__ koff_of)
{
    constexpr int KSTEPS = (D + 3) / 4;
    __shared__ __attribute__((aligned(16))) float cbuf_raw[R32_CR * R32_LD + 2 * S32_CPAD + 1];
    // the band's x frames as matrix-core operands: row r, lane group lk -> the four floats [bin lk, bin 4 + lk, bin 8 + lk,
    // bin 12 + lk] (one 16-byte read per tile hands a lane its operand of every contraction step), and their squared norms.
    // Laid out [tile][lk][row in tile]: the 16-byte unit a lane reads is unit `lane` of its tile, the one image every lane
    // group of ds_read_b128 takes without bank conflicts ([row][lk] made each group of 16 lanes hit 4 x 4 banks: 16 LDS
    // cycles per read instead of 4)
    __shared__ __attribute__((aligned(16))) float xa[R32_CR * 16];
    __shared__ __attribute__((aligned(16))) float xn[R32_CR];
    float *const cbuf = cbuf_raw + S32_CPAD;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / bands;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - S32_WIN + 1, N = ds.ny - S32_WIN + 1;
    const int R0 = (lb % bands) * R32_BR;
    if (R0 >= M) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    if ((int64_t)4 * ((int64_t)M + 64) * ds.crp_pitch > 0x7fffffffLL) return;      // 32-bit offsets, see strip_offsets_fit()
    // the band's 64 x frames (rotated by the OTI, 16 floats each: D bins, the norm, zeros): 256 chunks of 16 bytes
    if (tid < R32_CR * (S32_XP / 4)) {
        const float *xsrc = xp + (int64_t)p * max_nx * S32_XP;
        const int r = tid >> 2, c = tid & 3;
        const int row = min(R0 + r, ds.nx - 1);
        const float4 v = reinterpret_cast<const float4 *>(xsrc)[row * (S32_XP / 4) + c];
        const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = 4 * c + i;              // position in the packed line; (e & 3) == i
            if (e == D) xn[r] = e4[i];
            xa[(((r >> 4) * 4 + i) * 16 + (r & 15)) * 4 + c] = e < D ? e4[i] : 0.0f;      // bin e = 4 c + i: group i, step c
        }
    }
    // chunks: until every output column (< ny - 8) has left the lanes that hold real sums: a row's last block takes its
    // final 15 - q columns from lanes that only a further chunk fills (its y frames are clamped copies, never summed into
    // a stored cell)
    const int n_chunks = (ds.ny + S32_HALO - 1 + R32_CW - 1) / R32_CW;
    // y fragments of chunk t for this wave's 16 columns
    auto load_y = [&](const int t, float (&bf)[KSTEPS], float &yy) {
        const int jc = min(t * R32_CW + 16 * wave + lr, ds.ny - 1);
        const float *yp = feats + (ds.y_row0 + jc) * D;
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            const int bin = 4 * s + lk;
            bf[s] = bin < D ? yp[min(bin, D - 1)] : 0.0f;
        }
        yy = norms[ds.y_row0 + jc];
    };
    float bcur[KSTEPS], bnext[KSTEPS], ycur, ynext = 0.0f;
    load_y(0, bcur, ycur);
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) bnext[s] = 0.0f;
    __syncthreads();

    constexpr int CELL = OUT == 1 ? 2 : 4;   // bytes per cell of the result
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(out) + CELL * ds.crp_off, 0, (int)(CELL * (int64_t)M * ds.crp_pitch), S32_RSRC_WORD3);
    const bool even_layout = ((ds.crp_pitch & 1) == 0) && ((ds.crp_off & 1) == 0);       // block-uniform: one store per lane pair
    const uint32_t koff = OUT == 1 ? koff_of[p] : 0u;
    // the 16-bit plane's rows end on a multiple of 16 columns, K16_PAD (0xFFFF) behind the last key: the row selection reads 16
    // keys per lane and needs no tail handling then (keys16_kernels.hip)
    const int Npad = min((N + 15) & ~15, ds.crp_pitch);
    const int bw = wave & 1;                 // odd waves sit one column to the right (see above)
    float *const wr = cbuf + (4 * lk) * R32_LD + R32_CARRY + 16 * wave + lr;
    // (LDS byte address of the lane's first diagonal read: cbuf + (7 wave) rows + bw + 2 lane)
    const unsigned rdd_lds = (unsigned)(uintptr_t)(cbuf + (wave * R32_RPW) * R32_LD + bw + 2 * lane);
    const v4f32 *const xa_rd = reinterpret_cast<const v4f32 *>(xa) + lane;                // + 64 per row tile
    const v4f32 *const xn_rd = reinterpret_cast<const v4f32 *>(xn) + lk;                  // + 4 per row tile
    // carry: columns [128, 143) of all 64 rows -> [0, 15): 960 floats, two per thread for the first 480 threads
    const int ce0 = 2 * tid, ce1 = 2 * tid + 1;
    const int cr0 = ce0 / R32_CARRY, cc0 = ce0 - cr0 * R32_CARRY, cr1 = ce1 / R32_CARRY, cc1 = ce1 - cr1 * R32_CARRY;
    const bool carrier = tid < R32_CR * R32_CARRY / 2;
    float pa[R32_RPW], pb[R32_RPW];          // the previous chunk's sums of this lane, per row
    uint32_t ppk[R32_RPW];                   // (OUT == 1) ... as packed 16-bit keys: the lane's left column in the low half
#pragma unroll
    for (int q = 0; q < R32_RPW; q++) { pa[q] = pb[q] = 0.0f; ppk[q] = 0u; }

    // block [128 (t-1), 128 t) of row q: previous chunk's lanes [sh, 64) then the current chunk's lanes [0, sh).
    // FAST (block-uniform): whole band inside the matrix, even layout, block inside the row: one unconditional 8-byte store.
    // BW = the wave's parity (compile-time here: which of a lane's two sums opens an aligned pair depends on it).
    // Keys are the bit patterns of the sums (>= +0) with the sign bit set: -|s|.
    auto emit = [&](const int t, const int q, const float ca, const float cb, auto fast_tag, auto bw_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr int BW = decltype(bw_tag)::value;
        const int sigma = R32_CARRY - BW - q;          // cells between the block's first column and the lane pair that holds it
        const bool odd = (sigma & 1) != 0;
        const int sh0 = sigma >> 1, sh1 = (sigma + 1) >> 1;
        const float c0 = odd ? cb : ca, p0 = odd ? pb[q] : pa[q];
        const float c1 = odd ? ca : cb, p1 = odd ? pa[q] : pb[q];
        const float x0 = lane < sh0 ? -fabsf(c0) : -fabsf(p0), x1 = lane < sh1 ? -fabsf(c1) : -fabsf(p1);
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + sh0) & 63) << 2, (int)__float_as_uint(x0));
        const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + sh1) & 63) << 2, (int)__float_as_uint(x1));
        const int gi = R0 + wave * R32_RPW + q;
        const int soff = CELL * (gi * ds.crp_pitch + (t - 1) * R32_CW);
        if (OUT == 2) {
            if (r0 == 0x12345u) __builtin_amdgcn_raw_buffer_store_b32(r1, orsrc, 4 * (lane & 63), 0, 0);
        } else if (OUT == 1) {
            const uint32_t k0 = __builtin_elementwise_sub_sat(r0 & 0x7fffffffu, koff), k1 = __builtin_elementwise_sub_sat(r1 & 0x7fffffffu, koff);
            const uint32_t h0 = min(max(k0 >> 11, __builtin_elementwise_sub_sat(k0 >> 9, 49152u)), 0xFFFEu);
            const uint32_t h1 = min(max(k1 >> 11, __builtin_elementwise_sub_sat(k1 >> 9, 49152u)), 0xFFFEu);
            if (FAST) {
                __builtin_amdgcn_raw_buffer_store_b32(h0 | (h1 << 16), orsrc, 4 * lane, soff, S32_STORE_POLICY);
            } else if (gi < M) {
                const int col = (t - 1) * R32_CW + 2 * lane;
                if (col < Npad) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(col < N ? h0 : 0xFFFFu), orsrc, 4 * lane, soff, S32_STORE_POLICY);
                if (col + 1 < Npad) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(col + 1 < N ? h1 : 0xFFFFu), orsrc, 4 * lane + 2, soff, S32_STORE_POLICY);
            }
        } else if (FAST) {
            __builtin_amdgcn_raw_buffer_store_b64((u32x2s_t){r0, r1}, orsrc, 8 * lane, soff, S32_STORE_POLICY);
        } else if (gi < M) {
            const int col = (t - 1) * R32_CW + 2 * lane;
            if (col < N) __builtin_amdgcn_raw_buffer_store_b32(r0, orsrc, 8 * lane, soff, S32_STORE_POLICY);
            if (col + 1 < N) __builtin_amdgcn_raw_buffer_store_b32(r1, orsrc, 8 * lane + 4, soff, S32_STORE_POLICY);
        }
    };
    // the same for the 16-bit plane: keys are formed BEFORE the rows are brought into place, two per register, so an even
    // shift moves both with one select and one ds_bpermute, an odd one takes the two halves from neighbouring source lanes
    // (the sums are >= +0 with the sign bit clear: every C is max(., +0), an exact cancellation in the fma rounds to +0.  The
    // two keys of a lane are clamped to 16 bits by the saturating pack, then to 0xFFFE together)
    typedef unsigned short u16x2k __attribute__((ext_vector_type(2)));
    auto keys16 = [&](const float a, const float b) {
        const uint32_t ka = __builtin_elementwise_sub_sat(__float_as_uint(a), koff), kb = __builtin_elementwise_sub_sat(__float_as_uint(b), koff);
        const u16x2k coarse = __builtin_amdgcn_cvt_pk_u16(ka >> 11, kb >> 11);
        const u16x2k fine = __builtin_amdgcn_cvt_pk_u16(__builtin_elementwise_sub_sat(ka >> 9, 49152u), __builtin_elementwise_sub_sat(kb >> 9, 49152u));
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(coarse, fine),
                                                                      (u16x2k){(unsigned short)0xFFFEu, (unsigned short)0xFFFEu}));
    };
    auto emit16 = [&](const int t, const int q, const uint32_t cpk, auto fast_tag, auto bw_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr int BW = decltype(bw_tag)::value;
        constexpr int dummy = 0;
        (void)dummy;
        const int sigma = R32_CARRY - BW - q;
        const int sh0 = sigma >> 1, sh1 = (sigma + 1) >> 1;
        uint32_t r;
        if ((sigma & 1) == 0) {
            const uint32_t x = lane < sh0 ? cpk : ppk[q];
            r = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + sh0) & 63) << 2, (int)x);
        } else {
            const uint32_t x0 = lane < sh0 ? cpk : ppk[q], x1 = lane < sh1 ? cpk : ppk[q];
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + sh0) & 63) << 2, (int)x0);      // its high half opens the pair
            const uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + sh1) & 63) << 2, (int)x1);      // its low half closes it
            r = __builtin_amdgcn_alignbit(b1, b0, 16);
        }
        const int gi = R0 + wave * R32_RPW + q;
        const int soff = 2 * (gi * ds.crp_pitch + (t - 1) * R32_CW);
        if (FAST) {
            __builtin_amdgcn_raw_buffer_store_b32(r, orsrc, 4 * lane, soff, S32_STORE_POLICY);
        } else if (gi < M) {
            const int col = (t - 1) * R32_CW + 2 * lane;
            if (col < Npad) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(col < N ? (r & 0xFFFFu) : 0xFFFFu), orsrc, 4 * lane, soff, S32_STORE_POLICY);
            if (col + 1 < Npad) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(col + 1 < N ? (r >> 16) : 0xFFFFu), orsrc, 4 * lane + 2, soff, S32_STORE_POLICY);
        }
    };
    const bool fast_band = even_layout && (R0 + R32_BR <= M);      // block-uniform

    auto chunk = [&](const int t, auto fast_tag, auto bw_tag) {
        if (t + 1 < n_chunks) load_y(t + 1, bnext, ynext);
        // ---- C rows [R0, R0 + 64) x columns [128 t, 128 t + 128): this wave's 16 columns, two row tiles at a time: the LDS
        // operands of a half are requested before the first is used (left to itself under the 80-register budget hipcc
        // waits for every single read)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            v4f32 av[2], nv[2];
#pragma unroll
            for (int u = 0; u < 2; u++) av[u] = xa_rd[64 * (2 * h + u)];
#pragma unroll
            for (int u = 0; u < 2; u++) nv[u] = xn_rd[4 * (2 * h + u)];
            __builtin_amdgcn_sched_barrier(0);
            v4f32 acc[2];
            acc[0] = (v4f32){0.f, 0.f, 0.f, 0.f};
            acc[1] = (v4f32){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) {
#pragma unroll
                for (int u = 0; u < 2; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][s], bcur[s], acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    wr[(16 * (2 * h + u) + r) * R32_LD] = fmaxf(fmaf(-2.0f, acc[u][r], nv[u][r] + ycur), 0.0f);
            }
        }
        lds_barrier();
        // ---- rows 7w .. 7w+6 of the band: window sums along the lane's two diagonals
        // carry the chunk's last 15 columns: read before the barrier, write after it
        float k0 = 0.f, k1 = 0.f;
        if (carrier) {
            k0 = cbuf[cr0 * R32_LD + R32_CW + cc0];
            k1 = cbuf[cr1 * R32_LD + R32_CW + cc1];
        }
        // fifteen single ds_read_b64 (2 LDS cycles each: two groups of 32 lanes over 64 banks); left to the compiler they
        // pair up into ds_read2_b64, which the LDS serves as two 4 x 16-lane accesses: 8 cycles per pair
        v2f32 v[R32_RPW + S32_HALO];
#pragma unroll
        for (int m = 0; m < R32_RPW + S32_HALO; m++)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[m]) : "v"(rdd_lds), "n"(m * (R32_LD + 1) * 4) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                       "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(k0), "+v"(k1));
        // window sums of the wave's seven rows (row R0 + 7 wave + q: q = row mod 7, bands start at multiples of 56) with shared
        // partial sums: exactly window_sum9(&v[q], q & 1) of kernel_utils.h, in 24 packed additions instead of 56
        v2f32 ws[R32_RPW];
        {
            const v2f32 p1 = v[1] + v[2], p3 = v[3] + v[4], p5 = v[5] + v[6], p7 = v[7] + v[8];
            const v2f32 q1 = p1 + p3, q5 = p5 + p7;
            const v2f32 m0 = q1 + q5;
            ws[0] = v[0] + m0;
            ws[1] = m0 + v[9];
            const v2f32 p9 = v[9] + v[10];
            const v2f32 q3 = p3 + p5, q7 = p7 + p9;
            const v2f32 m2 = q3 + q7;
            ws[2] = v[2] + m2;
            ws[3] = m2 + v[11];
            const v2f32 p11 = v[11] + v[12];
            const v2f32 q9 = p9 + p11;
            const v2f32 m4 = q5 + q9;
            ws[4] = v[4] + m4;
            ws[5] = m4 + v[13];
            const v2f32 p13 = v[13] + v[14];
            const v2f32 q11 = p11 + p13;
            ws[6] = v[6] + (q7 + q11);
        }
#pragma unroll
        for (int q = 0; q < R32_RPW; q++) {
            const v2f32 s = ws[q];
            if constexpr (OUT == 1) {
                const uint32_t cpk = keys16(s.x, s.y);
                if (t > 0) emit16(t, q, cpk, fast_tag, bw_tag);
                ppk[q] = cpk;
            } else {
                if (t > 0) emit(t, q, s.x, s.y, fast_tag, bw_tag);
                pa[q] = s.x;
                pb[q] = s.y;
            }
        }
        lds_barrier();
        if (carrier) {
            cbuf[cr0 * R32_LD + cc0] = k0;
            cbuf[cr1 * R32_LD + cc1] = k1;
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) bcur[s] = bnext[s];
        ycur = ynext;
    };
    // blocks [128 (t-1), 128 t) that lie inside the row take the unconditional stores
    const int t_fast = fast_band ? min(N / R32_CW, n_chunks - 1) : 0;
    if (bw) {
        int t = 0;
        for (; t <= t_fast; t++) chunk(t, std::true_type{}, std::integral_constant<int, 1>{});
        for (; t < n_chunks; t++) chunk(t, std::false_type{}, std::integral_constant<int, 1>{});
#pragma unroll
        for (int q = 0; q < R32_RPW; q++) {
            if constexpr (OUT == 1) emit16(n_chunks, q, 0u, std::false_type{}, std::integral_constant<int, 1>{});
            else emit(n_chunks, q, 0.0f, 0.0f, std::false_type{}, std::integral_constant<int, 1>{});
        }
    } else {
        int t = 0;
        for (; t <= t_fast; t++) chunk(t, std::true_type{}, std::integral_constant<int, 0>{});
        for (; t < n_chunks; t++) chunk(t, std::false_type{}, std::integral_constant<int, 0>{});
#pragma unroll
        for (int q = 0; q < R32_RPW; q++) {
            if constexpr (OUT == 1) emit16(n_chunks, q, 0u, std::false_type{}, std::integral_constant<int, 0>{});
            else emit(n_chunks, q, 0.0f, 0.0f, std::false_type{}, std::integral_constant<int, 0>{});
        }
    }
}

}  // namespace acoss

using namespace acoss;

extern "C" int acoss_crp_planar32_batch(const float *xp, const float *feats, const float *norms, int d,
                                        const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                                        uint32_t *out, void *stream)
{
    if (!xp || !feats || !norms || !descs || !out || K < 0 || max_nx < win || max_ny < win) {
        set_error("crp_planar32_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if ((d != 12 && d != 13) || win != S32_WIN) {
        set_error("crp_planar32_batch: supports d in {12, 13} and win == 9");
        return ACOSS_ENOTSUP;
    }
    if (K == 0) return ACOSS_OK;
    const int bands = ceil_div(max_nx - win + 1, R32_BR);
    if ((int64_t)K * bands > 0x7fffffffLL || !strip_offsets_fit(max_nx, max_ny, 4)) { set_error("crp_planar32_batch: batch too large"); return ACOSS_ENOTSUP; }
    const unsigned nb = (unsigned)((int64_t)K * bands);
    hipStream_t st = (hipStream_t)stream;
    if (d == 12) hipLaunchKernelGGL(crp_rows32_kernel<12>, dim3(nb), dim3(512), 0, st, xp, max_nx, feats, norms, descs, bands, out, (const uint32_t *)nullptr);
    else hipLaunchKernelGGL(crp_rows32_kernel<13>, dim3(nb), dim3(512), 0, st, xp, max_nx, feats, norms, descs, bands, out, (const uint32_t *)nullptr);
    return launch_check("crp_rows32_kernel");
}

extern "C" int acoss_crp_keys16_batch(const float *xp, const float *feats, const float *norms, int d, const acoss_pair_desc *descs,
                                      int K, int win, int max_nx, int max_ny, const uint32_t *koff, uint16_t *out, void *stream)
{
    if (!xp || !feats || !norms || !descs || !koff || !out || K < 0 || max_nx < win || max_ny < win) {
        set_error("crp_keys16_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if ((d != 12 && d != 13) || win != S32_WIN) { set_error("crp_keys16_batch: supports d in {12, 13} and win == 9"); return ACOSS_ENOTSUP; }
    if (K == 0) return ACOSS_OK;
    const int bands = ceil_div(max_nx - win + 1, R32_BR);
    if ((int64_t)K * bands > 0x7fffffffLL || !strip_offsets_fit(max_nx, max_ny, 4)) { set_error("crp_keys16_batch: batch too large"); return ACOSS_ENOTSUP; }
    const unsigned nb = (unsigned)((int64_t)K * bands);
    hipStream_t st = (hipStream_t)stream;
    if (d == 12) hipLaunchKernelGGL((crp_rows32_kernel<12, 1>), dim3(nb), dim3(512), 0, st, xp, max_nx, feats, norms, descs, bands, reinterpret_cast<uint32_t *>(out), koff);
    else hipLaunchKernelGGL((crp_rows32_kernel<13, 1>), dim3(nb), dim3(512), 0, st, xp, max_nx, feats, norms, descs, bands, reinterpret_cast<uint32_t *>(out), koff);
    return launch_check("crp_rows32_kernel<keys16>");
}
