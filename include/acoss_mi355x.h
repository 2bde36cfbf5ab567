/*
 * acoss_mi355x.h -- C ABI of libacoss_mi355x.so: the MI355X (gfx950) implementation of acoss's
 * pairwise cover-song scoring hot path (Serra09: OTI -> cross-similarity -> sliding window ->
 * mutual kNN cross-recurrence plot -> qmax / dmax / constrained Smith-Waterman).
 *
 * Three groups of entry points:
 *
 *  (1) The reference's own native interface, same names and signatures, HOST pointers.  These
 *      are exactly what benchmarking/pySeqAlign.pxd:3-10 binds from
 *      benchmarking/SequenceAlignment.c (qmax_c :113, dmax_c :147, swalignimpconstrained :73);
 *      here they run on the GPU (copy in, one kernel, copy out).
 *
 *  (2) Batched stage kernels on DEVICE pointers, one per numeric function of
 *      benchmarking/CRPUtils.py on the path (get_oti :109, get_csm :67, sliding_csm :24,
 *      csm_to_binary :169, csm_to_binary_mutual :201) and per alignment recurrence.  The
 *      reference has no FFI for these (they are numpy); they are what a GPU port of
 *      Serra09.similarity (Serra09.py:158-196) calls instead.
 *
 *  (3) The whole chain in one call: acoss_corpus_create() puts a feature set into HBM,
 *      acoss_serra09_scores() takes a pair list through planning, OTI, cross-similarity +
 *      sliding window, mutual kNN masks and the alignment recurrences, and returns
 *      chroma_qmax / chroma_dmax (/ constrained Smith-Waterman) scores divided by (M + N):
 *      what Serra09.similarity (Serra09.py:161-184) computes per feature type.
 *
 * Conventions: every function returns 0 on success or a negative errno-style code
 * (ACOSS_E*); acoss_last_error() gives the message for the calling thread.  Nothing here
 * allocates device memory except the group-(1) host-pointer calls; group (2)/(3) work in
 * caller-provided buffers and are asynchronous on `stream` (a hipStream_t passed as void*,
 * NULL = the default stream).  No torch types appear in any signature.
 */
#ifndef ACOSS_MI355X_H
#define ACOSS_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACOSS_OK        0
#define ACOSS_EINVAL  (-22)   /* bad argument (null pointer, non-positive size, bad dtype) */
#define ACOSS_ENOMEM  (-12)   /* device allocation failed (host-pointer calls only) */
#define ACOSS_EIO      (-5)   /* HIP runtime error; see acoss_last_error() */
#define ACOSS_ENOTSUP (-95)   /* size beyond what the kernels support (see each function) */

#define ACOSS_ABI_VERSION 1

/* One (X, Y) song pair of a batch.  Songs live in one concatenated, frames-major feature
 * array `feats` (total_frames x d); x_row0 / y_row0 are the first frame of each song.  The
 * pair's (nx, ny) cross-similarity matrix lives at csm_off (elements, row pitch csm_pitch) of
 * the CSM buffer and its (nx-win+1, ny-win+1) cross-recurrence matrices at crp_off (elements,
 * row pitch crp_pitch) of the S / B buffers.  `shift` is the optimal transposition index
 * applied to X's bins (np.roll(chroma_i, oti, axis=0), Serra09.py:167); acoss_oti_batch fills
 * it.  acoss_plan_pairs() builds an array of these on the host. */
typedef struct acoss_pair_desc {
    int64_t x_row0;
    int64_t y_row0;
    int64_t csm_off;
    int64_t crp_off;
    int32_t nx;
    int32_t ny;
    int32_t csm_pitch;
    int32_t crp_pitch;
    int32_t shift;
    int32_t song_x;     /* indices into gchroma (and the caller's song list) */
    int32_t song_y;
    int32_t reserved;
} acoss_pair_desc;      /* 64 bytes */

/* One binary matrix handed to an alignment kernel: S is (rows, cols) uint8 with row pitch
 * s_pitch at s_off; D (optional) is float32 with row pitch d_pitch at d_off.  For qmax / dmax
 * D is (rows, cols); for the constrained Smith-Waterman it is (rows+1, cols+1)
 * (SequenceAlignment.c:65-69). */
typedef struct acoss_mat_desc {
    int64_t s_off;
    int64_t d_off;
    int32_t rows;
    int32_t cols;
    int32_t s_pitch;
    int32_t d_pitch;
} acoss_mat_desc;       /* 32 bytes */

/* Alignment penalties; the defaults are the constants hard-coded in the reference
 * (SequenceAlignment.c:45-46, 57-58, 105-106). */
typedef struct acoss_align_params {
    float gamma_onset;      /* 0.5  qmax/dmax gap after a recurrence point */
    float gamma_extension;  /* 0.5  qmax/dmax gap otherwise */
    float sw_match;         /* +1   */
    float sw_mismatch;      /* -1   */
    float sw_gap_open;      /* -0.5 */
    float sw_gap_ext;       /* -0.7 */
} acoss_align_params;

/* ---------------------------------------------------------------------------------------
 * library / device
 * ------------------------------------------------------------------------------------- */
int         acoss_abi_version(void);
const char *acoss_last_error(void);
/* Number of visible HIP devices (<0 on error) and selection of the current one. */
int         acoss_device_count(void);
int         acoss_set_device(int device);
void        acoss_default_align_params(acoss_align_params *p);

/* ---------------------------------------------------------------------------------------
 * (1) reference native interface -- host pointers, synchronous
 * ------------------------------------------------------------------------------------- */
/* SequenceAlignment.c:113  S uint8[M*N] row-major, D float32[M*N] in/out (caller zeroes it),
 * returns the maximum cell; 0.0 and D untouched when M<3 or N<3.  Errors (HIP failure, size
 * not supported) are reported as a NaN return value and through acoss_last_error(): the
 * reference signature has no error channel. */
float qmax_c(unsigned char *S, float *D, int M, int N);
/* SequenceAlignment.c:147  same buffers; 0.0 when M<4 or N<4.  D is read as well as written:
 * calling it on the D that qmax_c just filled reproduces Serra09.py:173-175. */
float dmax_c(unsigned char *S, float *D, int M, int N);
/* SequenceAlignment.c:73   S uint8[N*M] (N rows), D float32[(N+1)*(M+1)]; note (N, M) order. */
float swalignimpconstrained(unsigned char *S, float *D, int N, int M);

/* ---------------------------------------------------------------------------------------
 * host-side planning helper (pure CPU, no HIP)
 * ------------------------------------------------------------------------------------- */
/* Fills descs[K] from frame offsets (frame_off[n_songs+1], in frames) and a pair list
 * (pairs[K][2] song indices): matrix offsets are packed back to back with row pitches
 * rounded up to `pitch_align` elements (>=1; 16 makes every mask row 16-byte aligned).
 * total_csm / total_crp receive the element counts of the CSM and S/B buffers.
 * Serra09.py:162-172 shapes: CSM (nx, ny), CRP (nx-win+1, ny-win+1).  Returns ACOSS_EINVAL if
 * a song is shorter than `win`. */
int acoss_plan_pairs(const int64_t *frame_off, int n_songs, const int32_t *pairs, int K,
                     int win, int pitch_align, acoss_pair_desc *descs,
                     int64_t *total_csm, int64_t *total_crp);

/* ---------------------------------------------------------------------------------------
 * (2) batched stage kernels -- device pointers, asynchronous on `stream`
 * ------------------------------------------------------------------------------------- */
/* Per-frame squared norms sum_b x[b]^2 of a (n_frames, d) feature array; the CSM kernels take
 * them as input (the np.sum(X**2, 1) terms of CRPUtils.py:82). */
int acoss_frame_norms_f64(const double *feats, int64_t n_frames, int d, double *norms, void *stream);
int acoss_frame_norms_f32(const float *feats, int64_t n_frames, int d, float *norms, void *stream);

/* CRPUtils.py:109-136 get_oti for every pair: descs[p].shift = argmax_s sum_b
 * roll(g[song_x], s)[b] * g[song_y][b], first maximum wins; gchroma is (n_songs, nbins)
 * float64 (Serra09.py:24-28).  nbins <= 64. */
int acoss_oti_batch(const double *gchroma, int nbins, acoss_pair_desc *descs, int K, void *stream);

/* CRPUtils.py:67-84 get_csm for every pair: csm[csm_off + i*csm_pitch + j] =
 * sqrt(max(0, |x_i|^2 + |y_j|^2 - 2 x_i.y_j)) with X's bins rotated by descs[p].shift.
 * Output dtype = input dtype (float64 for crema chroma, float32 for essentia HPCP).
 * max_nx / max_ny bound the launch grid (>= every nx / ny of the batch).  d <= 64. */
int acoss_csm_batch_f64(const double *feats, const double *norms, int d,
                        const acoss_pair_desc *descs, int K, int max_nx, int max_ny,
                        double *csm, void *stream);
int acoss_csm_batch_f32(const float *feats, const float *norms, int d,
                        const acoss_pair_desc *descs, int K, int max_nx, int max_ny,
                        float *csm, void *stream);

/* Fast forms.  acoss_pack_x_* writes, per pair, the x song's frames rotated by descs[p].shift and
 * packed one frame per 16-element line [d values | squared norm | zeros] into xp
 * (acoss_xpack_elems(K, max_nx) elements, pair p at p*max_nx*16); 1 <= d <= 15.  The two consumers
 * below then fetch an x frame with two scalar loads:
 *   acoss_csm_packed_batch_*  = acoss_csm_batch_* (bit-identical output), d in {12, 13};
 *   acoss_crp_batch_*         = get_csm followed by sliding_csm without the CSM ever reaching HBM:
 *                               out[crp_off + i*crp_pitch + j] = sum_{k<win} C[i+k][j+k] when sqrt_out == 0
 *                               (C = clamped squared distance; the square of what sliding_csm returns --
 *                               the kNN selection only needs the order), or its sqrt when bit 0 of sqrt_out
 *                               is set.  d in {12, 13}, win <= 16.  The float64 kernel runs the K = d
 *                               contraction on the matrix cores (v_mfma_f64_16x16x4_f64); bit 1 of sqrt_out
 *                               selects the all-VALU form instead (A/B tests). */
int64_t acoss_xpack_elems(int K, int max_nx);
int acoss_pack_x_f64(const double *feats, const double *norms, int d, const acoss_pair_desc *descs, int K,
                     int max_nx, double *xp, void *stream);
int acoss_pack_x_f32(const float *feats, const float *norms, int d, const acoss_pair_desc *descs, int K,
                     int max_nx, float *xp, void *stream);
int acoss_csm_packed_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                               const acoss_pair_desc *descs, int K, int max_nx, int max_ny, double *csm,
                               void *stream);
int acoss_csm_packed_batch_f32(const float *xp, const float *feats, const float *norms, int d,
                               const acoss_pair_desc *descs, int K, int max_nx, int max_ny, float *csm,
                               void *stream);
/* float64 get_csm through the persistent matrix-core strip kernel (same output as acoss_csm_batch_f64
 * up to the rounding of one FMA-chain order; d in {12, 13}).
 * Size limit of the strip kernels (this one, acoss_crp_batch_f64 with win == 9, acoss_crp_planar_batch_f64,
 * acoss_crp_planar32_batch): they address ONE pair's result matrix with 32-bit byte offsets, so
 * cell_bytes * (max_nx + 64) * (max_ny + 256) must stay below 2^31 (cell_bytes 8, or 4 for the planar forms) --
 * about 16000 x 16000 frames for float64 -- and the descriptors' pitches at most max_ny + 256.  Larger shapes
 * return ACOSS_ENOTSUP (use acoss_csm_batch + acoss_sliding_batch, which index with 64 bits). */
int acoss_csm_strip_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                              const acoss_pair_desc *descs, int K, int max_nx, int max_ny, double *csm,
                              void *stream);
/* get_csm (CRPUtils.py:67-84) in row-band form (round 4; csrc/csm_rows_kernels.hip): the same arguments and bit for bit the
 * same matrix as acoss_csm_strip_batch_f64 -- x stationary in registers, y streaming, no shared memory; every wave store
 * writes four rows x 256 contiguous line-aligned bytes straight from the matrix-core accumulators (non-temporal).  The
 * HBM-bound kernel of the library: 8 bytes written per cell. */
int acoss_csm_rows_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                             const acoss_pair_desc *descs, int K, int max_nx, int max_ny, double *csm,
                             void *stream);
int acoss_crp_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                        const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, int sqrt_out,
                        double *out, void *stream);
/* Float32 approximation of acoss_crp_planar_batch_f64's matrix (xp, feats, norms: the float64 corpus rounded to
 * float32, packed by acoss_pack_x_f32): the windowed sums computed in float32 and written as order-preserving uint32
 * keys (the float32 bit pattern with the sign bit set), same element indexing.  |approx - exact| <= 2^-24 * ((d + 4.5) *
 * (sum over the window of |x_{i+k}|^2 + |y_{j+k}|^2) + 9.5 * exact).  CRPUtils.py:67-84 + :24-45, d in {12, 13}, win == 9. */
int acoss_crp_planar32_batch(const float *xp, const float *feats, const float *norms, int d,
                             const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                             uint32_t *out, void *stream);
int acoss_crp_batch_f32(const float *xp, const float *feats, const float *norms, int d,
                        const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, int sqrt_out,
                        double *out, void *stream);

/* The same windowed sums (win == 9, d in {12, 13}, float64) as a uint32 matrix of order-preserving key HIGH
 * WORDS (the IEEE bit pattern's upper half with the sign bit set; the sums are >= +0.0), indexed like the float64
 * matrix: out[crp_off + i * crp_pitch + j].  Half the bytes of acoss_crp_batch_f64.  The kNN selection
 * (acoss_mask_bits_planar_batch) is decided by the high words for all but ~0.1 % of the rows / columns and
 * recomputes the tied elements of those from the features.  crp_off and crp_pitch even for the fast stores
 * (acoss_plan_pairs with pitch_align >= 2). */
int acoss_crp_planar_batch_f64(const double *xp, const double *feats, const double *norms, int d,
                               const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                               uint32_t *out, void *stream);

/* CRPUtils.py:24-45 sliding_csm: S[i][j] = sqrt(sum_{k<win} csm[i+k][j+k]^2), always float64
 * out.  S is written at crp_off with pitch crp_pitch.  1 <= win <= 64. */
int acoss_sliding_batch_f64(const double *csm, const acoss_pair_desc *descs, int K, int win,
                            int max_nx, int max_ny, double *S, void *stream);
int acoss_sliding_batch_f32(const float *csm, const acoss_pair_desc *descs, int K, int win,
                            int max_nx, int max_ny, double *S, void *stream);

/* CRPUtils.py:169-199 / 201-219 csm_to_binary(_mutual): B[i][j] = 1 iff S[i][j] is among the
 * k_row smallest of row i (and, when mutual != 0, among the k_col smallest of column j);
 * k_row = round_half_even(kappa*cols) for 0<kappa<1, (int)kappa for kappa>=1, everything for
 * kappa == 0; k_col likewise from the row count.  Ties at the k-th value are resolved
 * lowest-index first (the reference's tie order is np.argpartition-internal).
 * `work` must hold acoss_binarize_work_bytes() bytes.  Supports rows, cols <= 2048. */
size_t acoss_binarize_work_bytes(int K, int max_nx, int max_ny, int win);
int acoss_binarize_batch(const double *S, const acoss_pair_desc *descs, int K, int win,
                         int max_nx, int max_ny, double kappa, int mutual, uint8_t *B,
                         void *work, size_t work_bytes, void *stream);

/* The two halves of acoss_binarize_batch separately.  acoss_thresholds_batch computes only the per-row
 * (and, when mutual, per-column) kNN thresholds of S into `work` (acoss_binarize_work_bytes()).
 * acoss_align_fused_batch then runs qmax (kind 0) or dmax (kind 1) directly from S and those thresholds,
 * forming the mask bits in registers inside the alignment sweep: the uint8 mask never exists in memory.
 * Requirements: S non-negative (the output of acoss_crp_batch_*), every pitch and offset of the descs
 * even, at most 1024 columns, gamma_onset == gamma_extension; otherwise ACOSS_ENOTSUP and the caller
 * uses acoss_binarize_batch + acoss_qmax_batch / acoss_dmax_batch.  `boundary` as for acoss_dmax_batch. */
int acoss_thresholds_batch(const double *S, const acoss_pair_desc *descs, int K, int win, int max_nx,
                           int max_ny, double kappa, int mutual, void *work, size_t work_bytes, void *stream);
int acoss_align_fused_batch(int kind, const double *T, const acoss_pair_desc *descs, int K, int win,
                            int max_nx, int max_ny, int mutual, const void *work, size_t work_bytes,
                            int boundary, const acoss_align_params *params, float *scores, void *stream);

/* Bit-mask path.  acoss_mask_bits_batch (matrices up to 1024 x 1024) = acoss_binarize_batch with the mask
 * bit-packed: bits[(p*max_m + i)*W + w] holds columns 64w .. 64w+63 of row i of pair p (max_m = max_nx-win+1;
 * bit c = column 64w + c; bits past the row end are zero; W = acoss_mask_bits_words() = 16 when both matrix
 * dimensions of the batch are <= 1024, else 32: acoss_mask_bits_planar_batch and the acoss_align_bits_* calls go
 * up to 2048 x 2048).  The selection kernels emit each row's and each
 * column's selected positions as bit vectors by ballot; a tile kernel transposes the column vectors and ANDs.
 * `work` needs acoss_mask_bits_work_bytes() bytes (thresholds + the two bit planes).
 * acoss_align_bits_batch runs qmax (kind 0) / dmax (kind 1) / swalignimpconstrained (kind 2) from those bits, one wave per pair
 * (gamma_onset == gamma_extension required; `boundary` as for acoss_dmax_batch).  With gamma = 0.5 (gammaState's value,
 * SequenceAlignment.c:104) and matrices up to 1024 x 1024, qmax and dmax run in 16-bit integers on E = 2 D (every D is a
 * multiple of 0.5; the reference's float32 arithmetic is exact on them): the same scores, about half the instructions.
 * Environment ACOSS_DP_Q16=0 (read once per process) keeps the float32 kernels. */
int acoss_mask_bits_words(int max_nx, int max_ny, int win);
size_t acoss_mask_bits_work_bytes(int K, int max_nx, int max_ny, int win);
int acoss_mask_bits_batch(const double *S, const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                          double kappa, int mutual, uint64_t *bits, void *work, size_t work_bytes, void *stream);

/* The same bit-packed mask from the high words written by acoss_crp_planar_batch_f64 (same work and bits
 * sizes as acoss_mask_bits_batch).  Rows or columns whose k-th smallest value shares its high word with another
 * element are finished exactly: their values are recomputed from the features (feats, norms, d and the descriptors'
 * shifts must be the ones the high words were built from) with the arithmetic of the strip kernel.
 * Results are identical to acoss_mask_bits_batch on the float64 matrix. */
int acoss_mask_bits_planar_batch(const uint32_t *planes, const double *feats, const double *norms, int d,
                                 const acoss_pair_desc *descs, int K, int win,
                                 int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits,
                                 void *work, size_t work_bytes, void *stream);
/* acoss_mask_bits_planar_batch on the float32-approximate keys of acoss_crp_planar32_batch.  band holds two floats per
 * pair, (base, slope): twice the error bound of an approximate value v of pair p is band[2p] + band[2p+1] * v
 * (base = 2 * (d + 4.5) * 2^-24 * (largest window sum of squared norms of song x + of song y), slope = 2 * 9.5 * 2^-24, both
 * rounded up; derivation in DESIGN.md section 4).  A row / column whose k-th smallest approximate value has another
 * value within that distance is finished in float64: the values inside the band are recomputed exactly from feats /
 * norms (the float64 corpus), the rest is decided by the approximation.  The masks equal
 * acoss_mask_bits_planar_batch's bit for bit. */
int acoss_mask_bits_planar32_batch(const uint32_t *keys, const float *band, const double *feats, const double *norms,
                                   int d, const acoss_pair_desc *descs, int K, int win,
                                   int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits,
                                   void *work, size_t work_bytes, void *stream);
/* The same filter with 16-bit keys (the default of acoss_serra09_scores since round 3; csrc/keys16.h): the float32 windowed
 * sums of acoss_crp_planar32_batch leave the chip as ONE uint16 per cell,
 *     k' = float32 bits -saturating koff[pair],   key16 = min(max(k' >> 11, (k' >> 9) -saturating 49152), 0xFFFE),
 *     koff[pair] = bit pattern of 2 W_pair 2^-7 (of the float32 not below 2 W_pair)
 * (W_pair = the largest window sum of squared norms of song x + that of song y, i.e. base / (2 * (d + 4.5) * 2^-24) of `band`; the kernels check
 * per pair that `band` and `koff` agree in this sense and fall back to the float evaluation of the reach where they do not; no
 * windowed sum exceeds 2 W_pair): 14 mantissa bits over the three octaves below 2 W_pair, 12 over the four below those, one
 * monotone map; same element indexing as the uint32 matrix (element = 2 bytes).  Behind the last key of a row the kernel writes
 * 0xFFFF up to the next multiple of 16 columns (as far as crp_pitch has room: acoss_plan_pairs with pitch_align a multiple of
 * 16 always has): the row selection reads 16 keys per lane and takes such rows without any tail handling; a plane whose pitch
 * is tighter is still selected correctly (the kernel masks the tail itself then).  A `keys16` plane handed to
 * acoss_mask_bits_keys16_batch MUST come from acoss_crp_keys16_batch with the same descriptors (or carry that padding): the
 * wave-per-row kernels trust it wherever the pitch has room for it; the radix selection masks every tail itself.
 * acoss_mask_bits_keys16_batch selects on those keys; where the winner's error band can reach another key it recomputes
 * the float32 values of the few cells involved from xp / f32 / n32 (the operands acoss_crp_keys16_batch was given, same
 * descriptors) with the strip kernel's arithmetic, and what float32 cannot decide is finished exactly in float64 from
 * feats / norms as in acoss_mask_bits_planar32_batch.  Masks identical to acoss_mask_bits_batch on the float64 sums.
 * Round 5: the selection itself is a radix selection with the keys in registers (csrc/radix16_kernels.hip; ACOSS_RADIX16, default on)
 * that settles 96 % of the rows and columns on the keys alone and the rest by exact float64 values of the two or three cells in
 * reach; the recompute tiers above run only for the pairs it hands back (exact ties) and with ACOSS_RADIX16=0.
 * CRPUtils.py:67-84 + :24-45 + :169-219; d in {12, 13}, win == 9, matrices up to 2048 x 2048 (beyond 1024: see
 * acoss_mask_bits_keys16_unresolved below); work / bits sizes as for acoss_mask_bits_batch.  (mutual == 2 / 3 launch the row / the column selection kernel alone and leave `bits` untouched:
 * measurement hooks for bench.py's per-kernel rooflines.) */
int acoss_crp_keys16_batch(const float *xp, const float *feats, const float *norms, int d, const acoss_pair_desc *descs,
                           int K, int win, int max_nx, int max_ny, const uint32_t *koff, uint16_t *out, void *stream);
int acoss_mask_bits_keys16_batch(const uint16_t *keys16, const float *band, const uint32_t *koff, const float *xp,
                                 const float *f32, const float *n32, const double *feats, const double *norms, int d,
                                 const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, double kappa,
                                 int mutual, uint64_t *bits, void *work, size_t work_bytes, void *stream);
/* The 16-bit-key filter for a corpus of FLOAT32 features (round 4; the reference's mfcc_htk / hpcp: get_csm follows its inputs'
 * dtype, CRPUtils.py:82, and sliding_csm squares in that dtype before it promotes to float64, :40-41).  f32 / n32 are the corpus
 * itself and its norms (acoss_frame_norms_f32), xp = acoss_pack_x_f32 of them, keys16 = acoss_crp_keys16_batch on them, band /
 * koff as above from the corpus' own norms: no centred copy -- the filter's cross-similarity values are then bit for bit the
 * exact path's and only the root-square and the float32 window sum separate the two (within the band's slope).  What the keys
 * and recomputed float32 values cannot decide is finished with the exact float32-input arithmetic
 * (sum over the window of (double)(sqrtf(max(c, 0))^2)): masks identical to acoss_crp_batch_f32 + acoss_mask_bits_batch. */
/* koff[pair] for a float32 corpus.  Such a corpus is its own filter operand (not centred), and features with a common offset --
 * real MFCC -- have raw norm sums far above every distance; but |x - y|^2 <= 2 (|x - m|^2 + |y - m|^2) for any m, so the key
 * range hangs on squared norms centred PER PAIR (m = the midpoint of the two songs' mean frames): top = 2 (Wx + Wy), koff = the
 * pattern of top 2^-7.  One block per pair.  Arguments as acoss_crp_keys16_batch's. */
int acoss_keys16_koff_f32_batch(const float *xp, const float *f32, const float *n32, int d, const acoss_pair_desc *descs, int K,
                                int win, int max_nx, int max_ny, uint32_t *koff, void *stream);
int acoss_mask_bits_keys16_f32_batch(const uint16_t *keys16, const float *band, const uint32_t *koff, const float *xp,
                                     const float *f32, const float *n32, int d, const acoss_pair_desc *descs, int K, int win,
                                     int max_nx, int max_ny, double kappa, int mutual, uint64_t *bits, void *work,
                                     size_t work_bytes, void *stream);
/* What the last acoss_mask_bits_keys16(_f32)_batch call on `work` did (synchronises with the device; tests and tools): out[0]
 * work items beyond the tiles' own slots, out[1] rows + columns that flagged their pair, out[2] pairs handed to the
 * wave-per-row kernels, out[3] reserved, out[8 .. 19] why lines flagged their pair (development statistics).  out: 20 ints. */
int acoss_mask_bits_keys16_stats(void *work, int K, int max_nx, int max_ny, int win, int *out);
/* Matrices with a side of 1025 .. 2048 (round 5: the long form of the radix selection, 64 dwords of keys per thread; needs
 * ACOSS_RADIX16 on, the default -- acoss_radix16_enabled()): acoss_mask_bits_keys16(_f32)_batch runs the radix selection alone,
 * and the pairs it cannot express (exact ties: more cells inside the reach of a row's k-th smallest key than a work item
 * holds; thresholds below the key range) stay UNRESOLVED -- their rows of `bits` are undefined.  This call waits for `stream`
 * and lists them: *n pairs (indices into the batch; at most `cap` are written to `list`).  The caller redoes them on the float64
 * path (acoss_crp_planar_batch_f64 + acoss_mask_bits_planar_batch), as acoss_serra09_scores does.  Matrices up to 1024 x 1024:
 * *n = 0 always (the call does not wait). */
int acoss_mask_bits_keys16_unresolved(void *work, int K, int max_nx, int max_ny, int win, int32_t *list, int cap, int *n,
                                      void *stream);
/* 1 unless the environment variable ACOSS_RADIX16 is "0", "false", "no" or empty (read at every call). */
int acoss_radix16_enabled(void);
/* Round 5: the selection of acoss_mask_bits_keys16_batch as a two-pass radix selection with the keys in registers
 * (csrc/radix16_kernels.hip) -- the form acoss_mask_bits_keys16_batch itself runs; these are its stages, exposed for the
 * per-kernel measurements of bench.py and the stage tests.  `what` is a bit set: 1 = the column kernel (every column's bound
 * t1: cells with key < t1 are selected; plus work items for the columns whose k-th smallest has another key inside its float32
 * error band), 2 = the row kernel (the same per row, then the base bits key < min(t1_row, t1_col) of the mutual mask into
 * `bits`), 4 = exact float64 values for the work items, their cells OR-ed into `bits`, and the list of pairs that have to
 * go through the wave-per-row kernels (exact ties).  acoss_radix16_layout: device pointers into `work`
 * (ptrs[8]: t1_row, t1_col, item_row, item_col, counters, items, pair_flag, pair_list; dims[4]: row stride of the t1 / item
 * arrays for rows and columns, item capacity, item size).  CRPUtils.py:169-219. */
size_t acoss_radix16_work_bytes(int K, int max_nx, int max_ny, int win);
int acoss_radix16_layout(void *work, int K, int max_nx, int max_ny, int win, void **ptrs, int *dims);
int acoss_radix16_stage(int what, const uint16_t *keys16, const float *band, const uint32_t *koff, const double *feats,
                        const double *norms, int d, const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny,
                        double kappa, int mutual, uint64_t *bits, void *work, size_t work_bytes, void *stream);
/* The product path: get_csm + sliding_csm + csm_to_binary_mutual (CRPUtils.py:67-84, :24-45, :201-219) without any
 * matrix in HBM.  One kernel forms a 24-row band of a pair's windowed sums in float32 on the matrix cores, keeps the
 * band's keys in registers, selects each row's k-th smallest and writes only the row's bit plane; run on (x, y) it
 * gives the row planes, on (y, x) the column planes; acoss's mutual mask is their AND (bits: same layout as
 * acoss_mask_bits_batch, W = 16).  Rows whose k-th smallest key has another key within the float32 error band (band:
 * as for acoss_mask_bits_planar32_batch) are finished exactly in float64 from feats / norms, so the masks equal
 * acoss_mask_bits_batch's on the float64 windowed sums bit for bit.
 *   pk      packed float32 frames of the whole corpus (acoss_pack_frames_f32): 16 floats per frame, [d values |
 *           squared norm | 0 ...], the corpus centred and scaled as the caller likes (distances only need to be
 *           consistent with `band`); the OTI rotation of song x is applied on the fly from descs[p].shift
 *   work    acoss_mask_bits_fused_work_bytes(K, max_nx, max_ny, win, side_rows) bytes; side_rows = capacity of the
 *           side buffer that carries undecided rows to the refinement kernel (4 KB each; 2 % of K * (M + N) is ample).
 *           After the stream has run, *acoss_mask_bits_fused_counter(work) (a DEVICE int) holds the number of rows that
 *           asked for a slot: if it exceeds side_rows the masks are incomplete and the call must be repeated with at
 *           least that many side rows.
 * Supported: d in {12, 13}, win == 9, matrices up to 1014 x 1014 (acoss_mask_bits_fused_supported). */
int acoss_pack_frames_f32(const float *feats, const float *norms, int d, int64_t n_frames, float *out, void *stream);
int acoss_mask_bits_fused_supported(int d, int win, int max_nx, int max_ny);
size_t acoss_mask_bits_fused_work_bytes(int K, int max_nx, int max_ny, int win, int side_rows);
int acoss_mask_bits_fused_batch(const float *pk, const float *band, const double *feats, const double *norms, int d,
                                const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, double kappa,
                                int mutual, uint64_t *bits, void *work, size_t work_bytes, int side_rows, void *stream);
const int *acoss_mask_bits_fused_counter(void *work);
int acoss_align_bits_batch(int kind, const uint64_t *bits, const acoss_pair_desc *descs, int K, int win,
                           int max_nx, int max_ny, int boundary, const acoss_align_params *params,
                           float *scores, void *stream);
/* qmax and dmax of the same bit mask in one sweep (Serra09.similarity's pair, Serra09.py:173-175; `boundary` = 1 is
 * its dmax on the D qmax leaves behind): same results as the two acoss_align_bits_batch calls. */
int acoss_align_bits_qd_batch(const uint64_t *bits, const acoss_pair_desc *descs, int K, int win, int max_nx,
                              int max_ny, int boundary, const acoss_align_params *params, float *qmax_scores,
                              float *dmax_scores, void *stream);

/* SequenceAlignment.c:113 / :147 / :73 over a batch of matrices.  S and D are the bases the
 * descs' offsets refer to; D may be NULL (scores only: the reference's callers never read D,
 * Serra09.py:174-175).  When D is given it is read AND written exactly like the reference's
 * in-place buffer: cells the recurrence never writes keep their previous contents and act as
 * boundary values.  scores[K] receives the maximum cell (not yet divided by M+N).
 * max_cols >= every mats[p].cols selects the kernel variant (one wave per matrix up to 2048
 * columns, one workgroup per matrix with LDS-resident rows up to ~8000 columns).
 * dmax `boundary`: 0 = zero boundary (fresh D, ChenFusion.py:66), 1 = the boundary qmax
 * leaves behind when D is not re-zeroed (Serra09.py:173-175); ignored when D is given. */
int acoss_qmax_batch(const uint8_t *S, const acoss_mat_desc *mats, int K, int max_cols, float *D,
                     const acoss_align_params *params, float *scores, void *stream);
int acoss_dmax_batch(const uint8_t *S, const acoss_mat_desc *mats, int K, int max_cols, float *D,
                     int boundary, const acoss_align_params *params, float *scores, void *stream);
int acoss_swc_batch(const uint8_t *S, const acoss_mat_desc *mats, int K, int max_cols, float *D,
                    const acoss_align_params *params, float *scores, void *stream);

/* ---------------------------------------------------------------------------------------
 * (4) evaluation -- the rank computation of CoverAlgorithm.getEvalStatistics (CoverAlgorithm.py:362-390)
 * ------------------------------------------------------------------------------------- */
/* D: device (N x N) float32 score matrix, row pitch row_pitch elements (higher = more similar; the diagonal is
 * ignored, as the reference sets it to -inf, :360).  clique_id[N]: device, songs with equal ids are versions of
 * one another.  mate_off[N + 1]: device prefix sums of (clique size - 1) per song.  For every song i the 1-based
 * ranks of its clique mates in the descending order of row i (the positions np.argsort(-D, 1) would give them,
 * :362/:381) are written, sorted ascending, to mate_ranks[mate_off[i] .. mate_off[i+1]).  Equal scores rank in
 * song-index order (the reference's unstable argsort leaves that order unspecified).  max_mates = the largest
 * clique size - 1 (<= 4096). */
int acoss_eval_ranks(const float *D, int N, int64_t row_pitch, const int32_t *clique_id, const int64_t *mate_off,
                     int max_mates, int32_t *mate_ranks, void *stream);

/* ---------------------------------------------------------------------------------------
 * (5) FTM2D -- the 2D-Fourier-transform-magnitude feature and its similarity (benchmarking/FTM2D.py)
 * ------------------------------------------------------------------------------------- */
/* btchroma: device, beat-synchronous chroma of all songs, beats-major (total_beats x 12) float64 (the transpose of
 * what FTM2D.py:91 produces per song); beat_off[n_songs + 1]: HOST array of first beats.  Per song: chrompwr (:9-25,
 * power `pwr`), |fft2| of every 12 x 75 window with fftshift (:29-48), per-window L2 norm and log(C x / norm + 1)
 * (:95-97), median over the windows (:98), L2 normalisation (:99) -> shingles[n_songs][900] (device); zeros for
 * songs with fewer than 75 beats (:87-90).  Up to 2122 beats per song.  scratch: device,
 * acoss_ftm2d_scratch_bytes(total_beats, total_windows, n_songs) bytes with total_windows = sum of
 * max(0, nbeats - 74).  Synchronises `stream` before returning. */
size_t acoss_ftm2d_scratch_bytes(int64_t total_beats, int64_t total_windows, int n_songs);
int acoss_ftm2d_shingles(const double *btchroma, const int64_t *beat_off, int n_songs, double pwr, double C,
                         void *scratch, size_t scratch_bytes, double *shingles, void *stream);
/* FTM2D.similarity (:117-127): sims[k] = exp(-|s_i - s_j|^2) for pairs[k] = (i, j); all device pointers. */
int acoss_ftm2d_pairs(const double *shingles, const int32_t *pairs, int K, double *sims, void *stream);
/* The same for all n x n pairs as one 900-deep product on the float64 matrix cores:
 * sims[i][j] = exp(-max(0, |s_i|^2 + |s_j|^2 - 2 s_i.s_j)). */
int acoss_ftm2d_gram(const double *shingles, int n, double *sims, void *stream);

/* ---------------------------------------------------------------------------------------
 * (6) similarity network fusion of a pair's block affinity matrices (benchmarking/SimilarityFusion.py,
 *     driven by benchmarking/EarlySNF.py:41-90)
 * ------------------------------------------------------------------------------------- */
/* One feature of the fusion: for every pair p the self-similarity matrices of song A (M x M, layout da[p]) and
 * song B (N x N, db[p]) and their cross-similarity matrix (M x N, dc[p]); the layouts are the crp_off / crp_pitch
 * fields of pair descriptors (e.g. planned for the pair lists (i, i), (j, j) and (i, j)).  All device pointers. */
typedef struct acoss_snf_feature {
    const double *ssma, *ssmb, *csm;
    const acoss_pair_desc *da, *db, *dc;
    int32_t win;                 /* unused by the kernels; the embedding window the matrices were built with */
} acoss_snf_feature;

/* For every pair: W_f = get_WCSMSSM(ssma, ssmb, csm, K) per feature (SimilarityFusion.py:94-134) with
 * K = int(kappa * (M + N)) (EarlySNF.py:51), snf_ws(Ws, K, niters, reg_diag=True) (:207-277), and
 * cross_out = -fused[0:M, M:] (EarlySNF.py:84-85) in the layout dout[p] (crp_off / crp_pitch), ready for
 * acoss_mask_bits_batch / acoss_binarize_batch.  feats, M, N: HOST arrays (n_feat, K, K); M + N <= 2048.
 * debug_W (n_feat x sum L^2) / debug_fused (sum L^2): optional device buffers receiving the affinity matrices and
 * the fused matrix, pair after pair, L x L row-major (tests).  Synchronises `stream` before returning. */
size_t acoss_snf_scratch_bytes(const int32_t *M, const int32_t *N, int K, int n_feat);
int acoss_snf_cross_batch(const acoss_snf_feature *feats, int n_feat, int K, const int32_t *M, const int32_t *N,
                          double kappa, double mu, int niters, void *scratch, size_t scratch_bytes,
                          const acoss_pair_desc *dout, double *cross_out, double *debug_W, double *debug_fused,
                          void *stream);

/* ---------------------------------------------------------------------------------------
 * (3) the chain in one call
 * ------------------------------------------------------------------------------------- */
/* A feature set resident in HBM: the songs of one feature type of Serra09.load_features' dict (Serra09.py:154), frames-
 * major.  acoss_corpus_create takes HOST arrays -- feats (total_frames x d float64, songs back to back), frame_off
 * (n_songs + 1 frame offsets), gchroma (n_songs x nbins float64 global chroma, Serra09.py:24-28; NULL for features
 * without OTI) -- uploads them, forms the per-frame norms and the centred float32 copy the matrix-core kernels use, and
 * owns that device memory until acoss_corpus_destroy.  acoss_corpus_wrap builds the same handle around arrays the
 * caller already holds on the DEVICE (f32 / n32: the corpus minus its mean, times a power of two, rounded to float32,
 * and its squared norms; norms_scaled: those norms in float64 on the HOST; all three NULL: float64 kernels only); the
 * caller keeps ownership.  Thread safety: acoss_serra09_scores calls on ONE handle are serialised inside the library (a
 * per-handle mutex: the handle owns the call's pinned staging); calls on different handles, each with its own scratch
 * and stream, run concurrently.  Environment switches (the same names the Python engine reads; README.md "Switches"):
 * ACOSS_PLANAR32=0 (float64 kernels only) and ACOSS_KEYS16=0 (32-bit instead of 16-bit keys for the float32 filter) are read
 * once, when the handle is made; ACOSS_DP_Q16=0 and ACOSS_DP_ONE_SWEEP=1 (alignment kernels) once per process. */
typedef struct acoss_corpus acoss_corpus;
int acoss_corpus_create(const double *feats, const int64_t *frame_off, int n_songs, int d, const double *gchroma,
                        int nbins, acoss_corpus **out);
int acoss_corpus_wrap(const double *feats, const double *norms, const double *gchroma, int nbins, const float *f32,
                      const float *n32, const double *norms_scaled, const int64_t *frame_off, int n_songs, int d,
                      acoss_corpus **out);
void acoss_corpus_destroy(acoss_corpus *c);

/* Serra09.py:161-184 for K pairs (pairs[K][2] song indices, HOST): per pair OTI (do_oti; CRPUtils.py:109) -> get_csm
 * (:67) -> sliding_csm (:24, window win) -> csm_to_binary_mutual (:201, kappa) -> the recurrences asked for in `want`
 * (bit 0: qmax_c, bit 1: dmax_c on the D qmax leaves behind as Serra09.py:173-175 drives it, bit 2:
 * swalignimpconstrained), each divided by (M + N) into its HOST output array (float64, position = position in `pairs`;
 * arrays of recurrences not asked for may be NULL).  Pairs are split by size class (matrices up to 1024 x 1024: float32
 * filter + exact float64 refinement, identical to float64; up to 2048 x 2048: float64 key high words; beyond, or feature
 * widths / windows without a fused kernel: one kernel per function with a byte mask) and run in batches of batch_pairs
 * (<= 0: sized from the song lengths, ~4096 pairs of 1000-frame songs); host planning of a batch overlaps the GPU work of
 * the one before it, and the call synchronises `stream` once, before it returns.  scratch: DEVICE memory,
 * acoss_serra09_scratch_bytes() of the same arguments.  Errors: ACOSS_EINVAL (bad indices, a song shorter than win,
 * scratch too small), ACOSS_EIO. */
size_t acoss_serra09_scratch_bytes(const acoss_corpus *c, const int32_t *pairs, int K, int win, int batch_pairs);
int acoss_serra09_scores(acoss_corpus *c, const int32_t *pairs, int K, int win, double kappa, int do_oti, int want,
                         int batch_pairs, void *scratch, size_t scratch_bytes, double *qmax, double *dmax, double *swc,
                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ACOSS_MI355X_H */
