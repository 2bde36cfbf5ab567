// band_kernels.hip -- get_csm + sliding_csm + csm_to_binary (CRPUtils.py:67-84, :24-45, :169-219) in ONE kernel:
// the windowed sums never go to HBM.
//
// Why: the materialising fast path writes the 3.9 MB key matrix of a pair once and reads it twice (rows, columns):
// 53 GB per 4096-pair step against 0.8 GB of features.  Here a block owns a band of 24 rows of a pair over all of
// its columns: it forms the band's squared distances on the matrix cores chunk by chunk (128 columns), sums the
// 9-long diagonal windows out of LDS straight into registers -- 16 keys per lane and row, the layout the histogram
// selection works on -- selects each row's k-th smallest key and leaves only the row's bit plane (128 bytes).  The
// column planes come from the same kernel run on the swapped pair (rows of (y, x) = columns of (x, y)); the two
// plane sets are ANDed by combine_bits_kernel as before.
//
// Arithmetic: float32 (v_mfma_f32_16x16x4_f32, a k-ordered chain of round-to-nearest FMAs; window sums in the order
// k = 0..8), i.e. the approximate keys of strip32_kernels.hip with the same error bound.  A row whose k-th smallest
// key has another key inside its error band is not decided here: its keys go to a compact side buffer and
// band_fix_kernel finishes it exactly in float64 (planar_select.h), so the masks equal the float64 path's bit for bit.
//
// Work decomposition (gfx950): 8 waves per block, 2 blocks per CU.
//   phase A (per 128-column chunk): wave v computes the 32 x 16 tile of columns 16v .. 16v+15 (two 16-row MFMA blocks
//     share the y fragment), epilogue C = max(fma(-2, dot, |x|^2 + |y|^2), 0) into the chunk buffer in LDS.  The buffer
//     keeps two interleaved column streams (columns u and u + 64 in adjacent words) so that phase B reads the
//     elements of two diagonals 64 columns apart with one aligned 8-byte ds_read per row.
//   phase B: wave v owns output rows 3v .. 3v+2: a lane walks its two diagonals down 11 C rows (11 ds_read_b64) and
//     forms 3 x 2 window sums -> registers h[q][2 chunk + stream].  One LDS-only barrier per chunk (chunk buffers are
//     double-buffered, the 10 carried columns are copied across).
//   selection: per row the histogram selection of planar_select.h on the register-resident keys, error-band check,
//     64-bit ballots funnel-shifted into column order, 128 bytes stored per row.
#include "planar_select.h"

namespace acoss {

typedef float bd_v4f32 __attribute__((ext_vector_type(4)));
typedef float bd_v2f32 __attribute__((ext_vector_type(2)));

constexpr int BD_WIN = 9, BD_HALO = BD_WIN - 1;
constexpr int BD_WAVES = 8;
constexpr int BD_RPW = 3;                       // output rows per wave
constexpr int BD_R = BD_WAVES * BD_RPW;         // 24 output rows per block
constexpr int BD_CROWS = BD_R + BD_HALO;        // 32 C rows = two 16-row MFMA blocks
constexpr int BD_HC = BD_HALO + BD_RPW - 1;     // columns carried from one chunk to the next (10)
constexpr int BD_CHUNK = 128;
constexpr int BD_SW = 64 + BD_HC;               // entries per column stream
constexpr int BD_PITCH = 2 * BD_SW;             // floats per C row in LDS (even: 8-byte reads stay aligned)
constexpr int BD_E = 16;                        // key registers per lane and row
constexpr int BD_MAXN = 64 * BD_E - BD_HC;      // longest row the register layout holds (1014)
constexpr int BD_MAXCH = (BD_MAXN + BD_HC + BD_CHUNK - 1) / BD_CHUNK;      // 8 chunks
constexpr int BD_XP = 16;                       // floats per packed frame: [d bins | squared norm | 0 ...]
constexpr int BD_SIDE_WORDS = 64 * BD_E;        // words per side-buffer slot
constexpr int BD_FLD = 20;                      // floats per staged frame in LDS (16 + 4: see the kernel)

static_assert(BD_CROWS == 32, "two MFMA row blocks");
static_assert(BD_MAXCH == 8, "chunk loop is unrolled 8 times");

// Where the band kernel puts rows it could not decide, and what the fix-up kernel needs to find them.
struct BandWork {
    uint64_t *row_bits;     // [K][max_m][16]
    uint64_t *col_bits;     // [K][16][max_n] (word-major, see thresh_work.h)
    int max_m, max_n;
    int *counter;           // [0]: slots asked for (may exceed cap: the caller re-runs with a larger side buffer)
    int4 *slots;            // [cap]: {pair, orientation, row, key of the tentative threshold}
    uint32_t *side;         // [cap][BD_SIDE_WORDS]: the row's keys in column order (float32 bits | sign bit)
    int cap;
    const float *band;      // per pair (base, slope) of the error band (engine.planar32_band)
};

__device__ inline int bd_srcbin(int bin, int shift, int d)
{
    // np.roll(chroma, oti, axis=0) (Serra09.py:167): rolled[b] = orig[(b - oti) mod d]; the norm slot and the padding stay
    if (bin >= d) return bin;
    int s = bin - shift;
    return s < 0 ? s + d : s;
}

// k-th smallest (1-based) of the wave's keys h[e] (invalid positions hold 0xffffffff, every valid key is a float32 >= +0
// bit pattern).  true: `thr` is the k-th smallest and no other key equals it; false: undecided here (`thr` = the key
// the candidates share, or 0).
//
// Histogram selection as in planar_select.h, with a single-level histogram of 256 bins per wave: lane l owns bins
// 4l .. 4l+3, so the counters are read and cleared with ONE conflict-free 16-byte access per lane (the 1024-bin form
// reads 4 x 16 bytes at a 64-byte lane stride -- 4-way bank conflicts on reads and clears -- and needs a second,
// dependent LDS read for its second level; LDS is what bounds the band kernel).  The window around the previous row's
// threshold is as wide as before (bins 4x wider): ~0.6 keys per bin, the 1-3 keys of the winning bin are ranked by
// v_readlane; two of them in one lane, or a cold start on the full range, re-bin that bin 256x finer.
constexpr int BH_LOG2 = 8, BH_BINS = 1 << BH_LOG2, BH_WORDS = BH_BINS + 64;
constexpr int BH_SHIFT0 = 13, BH_SHIFT_MAX = 19;

__device__ inline void bh_clear(unsigned *hist, int lane)
{
    reinterpret_cast<uint4 *>(hist)[lane] = make_uint4(0, 0, 0, 0);
    hist[BH_BINS + lane] = 0;
}

__device__ inline bool band_select(const unsigned (&h)[BD_E], int k, unsigned *hist, int lane, HistWarm &warm, unsigned &thr)
{
    unsigned bin[BD_E];
    enum { PREDICTED, FULL, REFINE };
    int kind = warm.hi != 0 ? PREDICTED : FULL;
    unsigned lo = 0;
    int shift = warm.shift;
    thr = 0;
    if (kind == PREDICTED) {
        const unsigned half = (unsigned)(BH_BINS / 2) << shift;
        lo = max(warm.hi, half) - half;
    }
    int r = 0, cstar = 0;
    unsigned ch = 0;
    uint64_t any = 0;
    for (;;) {
        int below = 0;
        if (kind == FULL) {
            unsigned mn = 0xffffffffu, mx = 0u;
#pragma unroll
            for (int e = 0; e < BD_E; e++) {
                mn = min(mn, h[e]);
                mx = max(mx, h[e] == 0xffffffffu ? 0u : h[e]);
            }
            mn = wave_umin(mn);
            mx = wave_umax(mx);
            lo = mn;
            shift = mx > mn ? max(0, 32 - (int)__clz(mx - mn) - BH_LOG2) : 0;
        }
        // keys below lo wrap to >= 2^31 and the invalid ones sit >= 2^31 above every valid key: with shift <= 23 both
        // land in the lane's spill word (a wave-wide ds_add on 64 consecutive words runs at the full LDS atomic rate,
        // 4.2 cycles per instruction and CU: tools/ubench/lds_atomic.hip; masking lanes off costs more than it saves).
        // The number of keys below lo comes from the borrow of the subtraction.
        const unsigned spill = (unsigned)(BH_BINS + lane);
        const unsigned lo_s = (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
#pragma unroll
        for (int e = 0; e < BD_E; e++) {
            unsigned d;
            uint64_t borrow;
            asm("v_sub_co_u32 %0, %1, %2, %3" : "=v"(d), "=s"(borrow) : "v"(h[e]), "s"(lo_s));
            below += __popcll(borrow);
            unsigned b = min(d >> shift, spill);
            asm("" : "+v"(b));      // opaque: hipcc 7.2 crashes in instruction selection on the folded LDS address
            bin[e] = b;
            atomicAdd(&hist[b], 1u);
        }
        const int kk = k - below;
        const uint4 c4 = reinterpret_cast<const uint4 *>(hist)[lane];
        const int tot = (int)(c4.x + c4.y + c4.z + c4.w);
        const int incl = wave_scan<OpAdd>(tot, 0);
        bh_clear(hist, lane);
        const uint64_t m1 = __ballot((incl - tot < kk) & (kk <= incl));
        if (m1 == 0) {
            if (kind != PREDICTED) return false;
            warm.shift = min(warm.shift + 1, BH_SHIFT_MAX);
            kind = FULL;
            continue;
        }
        const int ls = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)m1) - 1);
        r = kk - (__builtin_amdgcn_readlane(incl, ls) - __builtin_amdgcn_readlane(tot, ls));
        // the four counters of lane ls (wave-uniform from here on)
        const int c0 = __builtin_amdgcn_readlane((int)c4.x, ls), c1 = __builtin_amdgcn_readlane((int)c4.y, ls);
        const int c2 = __builtin_amdgcn_readlane((int)c4.z, ls), c3 = __builtin_amdgcn_readlane((int)c4.w, ls);
        int ts = 0;
        cstar = c0;
        if (r > c0) {
            r -= c0; ts = 1; cstar = c1;
            if (r > c1) {
                r -= c1; ts = 2; cstar = c2;
                if (r > c2) { r -= c2; ts = 3; cstar = c3; }
            }
        }
        const unsigned bstar = (unsigned)(4 * ls + ts);
        uint64_t dup = 0;
        any = 0;
#pragma unroll
        for (int e = 0; e < BD_E; e++) {
            const bool in = bin[e] == bstar;
            const uint64_t m = __ballot(in);
            dup |= any & m;
            any |= m;
            ch = in ? h[e] : ch;
        }
        if (dup == 0) break;
        if (shift == 0) {                   // equal keys in one lane: exact values needed
            thr = lo + bstar;
            return false;
        }
        lo += bstar << shift;
        shift = max(shift - BH_LOG2, 0);
        kind = REFINE;
    }
    const bool mine = (any >> lane) & 1;
    int less = 0, equal = 1;
    if (cstar > 1) {
        equal = 0;
        for (uint64_t rest = any; rest != 0; rest &= rest - 1) {
            const int c = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)rest) - 1);
            const unsigned vc = (unsigned)__builtin_amdgcn_readlane((int)ch, c);
            less += vc < ch;
            equal += vc == ch;
        }
    }
    const uint64_t win = __ballot(mine & (less < r) & (r <= less + equal));
    if (win == 0) return false;
    const int wl = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)win) - 1);
    thr = (unsigned)__builtin_amdgcn_readlane((int)ch, wl);
    if (__builtin_amdgcn_readlane(equal, wl) > 1) return false;
    warm.hi = thr;
    return true;
}

// [lo, hi]: the keys of the values within the error band of the value of key th (plain float32 bit patterns), widened
// by one ulp each way for the rounding of the two float operations
__device__ inline void bd_band_limits(unsigned th, const float *pair_band, unsigned &lo, unsigned &hi)
{
    const float a = __uint_as_float(th);
    const float band = fmaf(pair_band[1], a, pair_band[0]);
    const float l = a - band, h = a + band;
    lo = l > 0.0f ? __float_as_uint(l) - 1u : 0u;
    hi = __float_as_uint(h) + 1u;
}

template <int D>
__global__ __launch_bounds__(64 * BD_WAVES, 4) void crp_band_kernel(const float *__restrict__ pk,
                                                                    const acoss_pair_desc *__restrict__ descs,
                                                                    int bands_m, int bands_n, double kv, int k_mode,
                                                                    BandWork bw, int mode)
{
    // mode (development ablations, product = 0): 1 = no selection, 2 = no window sums, 4 = no matrix-core phase
    constexpr int KSTEPS = (D + 3) / 4;
    // LDS: the two chunk buffers, per wave the packed frames of its 16 columns of a chunk, the band's 32 x frames, the
    // column-plane staging of orientation 1 and the waves' histograms
    __shared__ __attribute__((aligned(16))) float cbuf_raw[2 * BD_CROWS * BD_PITCH];
    __shared__ __attribute__((aligned(16))) float ybuf_all[BD_WAVES * 16 * BD_FLD];
    __shared__ __attribute__((aligned(16))) float abuf[BD_CROWS * BD_FLD];
    __shared__ __attribute__((aligned(16))) uint64_t obuf[BD_E * BD_R];
    __shared__ __attribute__((aligned(16))) unsigned hist_all[BD_WAVES * BH_WORDS];

    const int per_pair = bands_m + bands_n;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / per_pair, t = lb % per_pair;
    const int orient = t >= bands_m ? 1 : 0;
    const acoss_pair_desc ds = descs[p];
    // rows of this orientation: frames of A; columns: frames of B
    const int nA = orient ? ds.ny : ds.nx, nB = orient ? ds.nx : ds.ny;
    const int64_t rowA0 = orient ? ds.y_row0 : ds.x_row0, rowB0 = orient ? ds.x_row0 : ds.y_row0;
    const int shiftA = orient ? 0 : ds.shift, shiftB = orient ? ds.shift : 0;
    const int Mo = nA - BD_HALO, No = nB - BD_HALO;
    const int i0 = (orient ? t - bands_m : t) * BD_R;
    if (i0 >= Mo) return;
    if (mode & 8) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int nchunks = min(BD_MAXCH, (No + BD_HC + BD_CHUNK - 1) / BD_CHUNK);

    // ---- operands: whole packed frames (64 bytes) by 16-byte loads, 4 lanes per frame, through LDS (row stride 20
    // floats: the fragment reads are 2-way bank conflicts instead of 8-way); fragments are read with the OTI rotation
    const int st_f = lane >> 2, st_p = lane & 3;
    float *const ybuf = ybuf_all + wave * (16 * BD_FLD);
    auto gload_b = [&](const int ch) {
        return *reinterpret_cast<const float4 *>(pk + (rowB0 + min(BD_CHUNK * ch + 16 * wave + st_f, nB - 1)) * BD_XP + 4 * st_p);
    };
    float4 bq0 = gload_b(0), bq1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nchunks > 1) bq1 = gload_b(1);
    if (tid < 4 * BD_CROWS) {
        const int f = tid >> 2, part = tid & 3;
        *reinterpret_cast<float4 *>(abuf + f * BD_FLD + 4 * part) =
            *reinterpret_cast<const float4 *>(pk + (rowA0 + min(i0 + f, nA - 1)) * BD_XP + 4 * part);
    }
    __syncthreads();
    float afrag[2][KSTEPS], nxv[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; rb++) {
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            const int bin = 4 * s + lk;
            const float v = abuf[(16 * rb + lr) * BD_FLD + bd_srcbin(bin, shiftA, D)];
            afrag[rb][s] = bin < D ? v : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) nxv[rb][r] = abuf[(16 * rb + 4 * lk + r) * BD_FLD + D];
    }
    int boff[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) boff[s] = lr * BD_FLD + bd_srcbin(4 * s + lk, shiftB, D);
    float bfrag[KSTEPS], yy;
    // the wave's tile of chunk c: registers -> its LDS slot -> fragments (wave-private: ordered by the wave's own waits)
    auto stage_b = [&](const float4 q) {
        *reinterpret_cast<float4 *>(ybuf + st_f * BD_FLD + 4 * st_p) = q;
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            const float v = ybuf[boff[s]];
            bfrag[s] = 4 * s + lk < D ? v : 0.0f;
        }
        yy = ybuf[lr * BD_FLD + D];
    };
    stage_b(bq0);
    if (nchunks > 2) bq0 = gload_b(2);

    // LDS addresses.  Chunk-local column u = HC + 16 wave + lr of the new columns; stream 0 holds u in [0, SW) at
    // word 2u, stream 1 holds u in [64, 64 + SW) at word 2 (u - 64) + 1; columns in both ranges are written twice.
    const int u = BD_HC + 16 * wave + lr;
    const int slot_a = wave < 4 ? 2 * u : 2 * (u - 64) + 1;
    const bool dual = (wave == 3) & (u >= 64);
    const int slot_b = 2 * (u - 64) + 1;
    const int wr_off = (4 * lk) * BD_PITCH + slot_a;
    const int wr_off2 = (4 * lk) * BD_PITCH + slot_b;
    const int rd_off = (BD_RPW * wave) * BD_PITCH + 2 * lane;
    // carried columns: u in [128, 128 + HC) of this chunk = u - 128 of the next one
    const bool carrier = tid < BD_CROWS * BD_HC;
    const int c_row = tid / BD_HC, c_col = tid - c_row * BD_HC;
    const int carry_src = c_row * BD_PITCH + 2 * (64 + c_col) + 1, carry_dst = c_row * BD_PITCH + 2 * c_col;

    unsigned h[BD_RPW][BD_E];
#pragma unroll
    for (int q = 0; q < BD_RPW; q++) {
#pragma unroll
        for (int e = 0; e < BD_E; e++) h[q][e] = 0xffffffffu;
    }

#pragma unroll
    for (int ch = 0; ch < BD_MAXCH; ch++) {
        if (ch < nchunks) {          // block-uniform
            float *const cb = cbuf_raw + (ch & 1) * (BD_CROWS * BD_PITCH);
            // ---- phase A: C rows [i0, i0 + 32) x columns [128 ch + 16 wave, + 16)
            if (!(mode & 4)) {
                bd_v4f32 acc[2];
                acc[0] = (bd_v4f32){0.f, 0.f, 0.f, 0.f};
                acc[1] = (bd_v4f32){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KSTEPS; s++) {
#pragma unroll
                    for (int rb = 0; rb < 2; rb++)
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[rb][s], bfrag[s], acc[rb], 0, 0, 0);
                }
#pragma unroll
                for (int rb = 0; rb < 2; rb++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float c = fmaxf(fmaf(-2.0f, acc[rb][r], nxv[rb][r] + yy), 0.0f);
                        cb[wr_off + (16 * rb + r) * BD_PITCH] = c;
                        if (dual) cb[wr_off2 + (16 * rb + r) * BD_PITCH] = c;
                    }
                }
            }
            lds_barrier();
            // next chunk's fragments (its tile was loaded two chunks ago), and the load for the chunk after next
            if (ch + 1 < nchunks) {
                stage_b((ch & 1) ? bq0 : bq1);
                if (ch + 3 < nchunks) {
                    if (ch & 1) bq0 = gload_b(ch + 3);
                    else bq1 = gload_b(ch + 3);
                }
            }
            // ---- phase B: window sums of rows 3 wave + q, columns 128 ch - HC + 64 s + lane + q
            if (mode & 2) {
#pragma unroll
                for (int q = 0; q < BD_RPW; q++) {
                    h[q][2 * ch] = __float_as_uint(afrag[0][0]) + lane * 977u + q;
                    h[q][2 * ch + 1] = __float_as_uint(nxv[0][0]) + lane * 1471u + q;
                }
            } else {
                bd_v2f32 v[BD_RPW + BD_HALO];
#pragma unroll
                for (int m = 0; m < BD_RPW + BD_HALO; m++)
                    v[m] = *reinterpret_cast<const bd_v2f32 *>(cb + rd_off + m * (BD_PITCH + 2));
                if (ch + 1 < nchunks && carrier) cbuf_raw[((ch + 1) & 1) * (BD_CROWS * BD_PITCH) + carry_dst] = cb[carry_src];
#pragma unroll
                for (int q = 0; q < BD_RPW; q++) {
                    bd_v2f32 s = v[q];
#pragma unroll
                    for (int k = 1; k < BD_WIN; k++) s += v[q + k];
                    h[q][2 * ch] = __float_as_uint(s.x);
                    h[q][2 * ch + 1] = __float_as_uint(s.y);
                }
            }
        }
    }

    // ---- selection: row i0 + 3 wave + q; position (e, lane) of its registers is column 64 e + lane - (HC - q)
    unsigned *hist = hist_all + wave * BH_WORDS;
    bh_clear(hist, lane);
    const int k = knn_count(k_mode, kv, No);
    const float *pair_band = bw.band + 2 * p;
    HistWarm warm{0, BH_SHIFT0};
#pragma unroll
    for (int q = 0; q < BD_RPW; q++) {
        const int i = i0 + BD_RPW * wave + q;
        if (i < Mo) {            // wave-uniform
            const int dq = BD_HC - q;
            unsigned(&hq)[BD_E] = h[q];
            // positions outside the row (before column 0, from column No on) get the largest key
#pragma unroll
            for (int e = 0; e < BD_E; e++) {
                if (64 * e - dq < 0 || 64 * e + 63 - dq >= No) {
                    const int col = 64 * e + lane - dq;
                    hq[e] = ((col >= 0) & (col < No)) ? hq[e] : 0xffffffffu;
                }
            }
            uint64_t word[BD_E];
            bool decided = true;
            unsigned thr = 0;
            if (k <= 0) {
#pragma unroll
                for (int e = 0; e < BD_E; e++) word[e] = 0ull;
            } else if (k >= No) {
#pragma unroll
                for (int e = 0; e < BD_E; e++) {
                    const int left = No - 64 * e;
                    word[e] = left >= 64 ? ~0ull : (left > 0 ? (1ull << left) - 1ull : 0ull);
                }
            } else if (mode & 1) {
#pragma unroll
                for (int e = 0; e < BD_E; e++) word[e] = __ballot(hq[e] <= hq[0]);
            } else {
                decided = band_select(hq, k, hist, lane, warm, thr);
                uint64_t bal[BD_E + 1];
                if (decided) {
                    // thr is the k-th smallest and unique.  It stands if no other key lies in its error band [blo, bhi]:
                    // exactly k keys <= bhi and k - 1 keys < blo; the keys <= bhi are then the row's selection.
                    unsigned blo, bhi;
                    bd_band_limits(thr, pair_band, blo, bhi);
                    int ca = 0, cb = 0;
#pragma unroll
                    for (int e = 0; e < BD_E; e++) {
                        bal[e] = __ballot(hq[e] <= bhi);
                        ca += __popcll(bal[e]);
                        cb += __popcll(__ballot(hq[e] < blo));
                    }
                    decided = (ca == k) & (cb == k - 1);
                }
                if (decided) {
                    bal[BD_E] = 0ull;
#pragma unroll
                    for (int e = 0; e < BD_E; e++) word[e] = (bal[e] >> dq) | (bal[e + 1] << (64 - dq));
                } else {
                    // undecided: hand the row's keys (column order, sign bit set as the fix-up code expects) to the fix-up kernel
                    int slot = 0;
                    if (lane == 0) slot = atomicAdd(bw.counter, 1);
                    slot = __builtin_amdgcn_readfirstlane(slot);
                    if (slot < bw.cap) {
                        uint32_t *dst = bw.side + (int64_t)slot * BD_SIDE_WORDS;
#pragma unroll
                        for (int e = 0; e < BD_E; e++) {
                            const int col = 64 * e + lane - dq;
                            if ((col >= 0) & (col < No)) dst[col] = hq[e] | 0x80000000u;
                        }
                        if (lane == 0) bw.slots[slot] = make_int4(p, orient, i, (int)(thr != 0 ? (thr | 0x80000000u) : 0u));
                    }
#pragma unroll
                    for (int e = 0; e < BD_E; e++) word[e] = 0ull;
                }
            }
            unsigned wlo = 0, whi = 0;
            if (mode & 64) { wlo = (unsigned)word[3]; whi = (unsigned)(word[5] >> 32); }
            else
#pragma unroll
            for (int e = 0; e < BD_E; e++) planar_put_lane_u64(wlo, whi, word[e], e);
            const uint64_t mine = ((uint64_t)whi << 32) | wlo;
            if (orient == 0) {
                if (lane < BD_E) bw.row_bits[((int64_t)p * bw.max_m + i) * BD_E + lane] = mine;
            } else {
                if (lane < BD_E) obuf[lane * BD_R + BD_RPW * wave + q] = mine;
            }
        }
    }
    if (orient) {
        // the band's 24 columns x 16 words leave word-major: 16 runs of 24 consecutive words
        __syncthreads();
        if (tid < BD_E * BD_R) {
            const int e = tid / BD_R, rl = tid - e * BD_R;
            if (i0 + rl < Mo) bw.col_bits[((int64_t)p * BD_E + e) * bw.max_n + i0 + rl] = obuf[tid];
        }
    }
}

// One wave per side-buffer slot: the exact float64 refinement of planar_select.h on the row's keys.
__global__ __launch_bounds__(64) void band_fix_kernel(const double *__restrict__ feats, const double *__restrict__ norms, int d,
                                                      const acoss_pair_desc *__restrict__ descs, int win, double kv, int k_mode,
                                                      BandWork bw)
{
    __shared__ FixSmem sm;
    const int slot = blockIdx.x;
    if (slot >= min(*bw.counter, bw.cap)) return;
    const int4 rec = bw.slots[slot];
    const int p = rec.x, orient = rec.y, which = rec.z;
    const unsigned thr_hi = (unsigned)rec.w;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    const int len = orient ? M : N;
    const int lane = threadIdx.x;
    const int k = knn_count(k_mode, kv, len);
    const uint32_t *keys = bw.side + (int64_t)slot * BD_SIDE_WORDS;
    auto key_at = [&](int q) { return keys[q]; };
    ThreshWork w;
    w.row_thr = w.col_thr = nullptr;
    w.row_cut = w.col_cut = nullptr;
    w.max_m = bw.max_m;
    w.max_n = bw.max_n;
    w.row_bits = bw.row_bits;
    w.col_bits = bw.col_bits;
    w.wpr = BD_E;
    w.band = bw.band;
    if (orient == 0) {
        if (fix_row_band<0, BD_E>(sm, key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, nullptr, nullptr, lane)) return;
        fix_row_generic<0, BD_E>(key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, nullptr, nullptr, lane);
    } else {
        if (fix_row_band<1, BD_E>(sm, key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, nullptr, nullptr, lane)) return;
        fix_row_generic<1, BD_E>(key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, nullptr, nullptr, lane);
    }
}

// packed float32 frames: [d values | squared norm | zeros] per frame, 64 bytes
__global__ void pack_frames_kernel(const float *__restrict__ feats, const float *__restrict__ norms, int d, int64_t n_frames,
                                   float *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t f = g >> 4;
    const int c = (int)(g & 15);
    if (f >= n_frames) return;
    out[g] = c < d ? feats[f * d + c] : (c == d ? norms[f] : 0.0f);
}

// defined in crp_kernels.hip
int launch_combine_bits(const acoss_pair_desc *descs, int K, int win, int mutual, ThreshWork w, uint64_t *bits, hipStream_t st);

static size_t bd_align(size_t b) { return (b + 255) & ~(size_t)255; }

static size_t band_work_layout(void *work, int K, int max_m, int max_n, int side_rows, BandWork &bw)
{
    char *base = (char *)work;
    size_t off = 0;
    bw.max_m = max_m;
    bw.max_n = max_n;
    bw.counter = (int *)(base + off);
    off += 256;
    bw.row_bits = (uint64_t *)(base + off);
    off += bd_align((size_t)K * max_m * BD_E * sizeof(uint64_t));
    bw.col_bits = (uint64_t *)(base + off);
    off += bd_align((size_t)K * max_n * BD_E * sizeof(uint64_t));
    bw.slots = (int4 *)(base + off);
    off += bd_align((size_t)side_rows * sizeof(int4));
    bw.side = (uint32_t *)(base + off);
    off += bd_align((size_t)side_rows * BD_SIDE_WORDS * sizeof(uint32_t));
    bw.cap = side_rows;
    bw.band = nullptr;
    return off;
}

}  // namespace acoss

using namespace acoss;

extern "C" {

int acoss_pack_frames_f32(const float *feats, const float *norms, int d, int64_t n_frames, float *out, void *stream)
{
    if (!feats || !norms || !out || d < 1 || d > 15 || n_frames < 0) { set_error("pack_frames_f32: bad argument (1 <= d <= 15)"); return ACOSS_EINVAL; }
    if (n_frames == 0) return ACOSS_OK;
    const int64_t total = n_frames * 16;
    hipLaunchKernelGGL(pack_frames_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, feats, norms, d, n_frames, out);
    return launch_check("pack_frames_kernel");
}

int acoss_mask_bits_fused_supported(int d, int win, int max_nx, int max_ny)
{
    return (d == 12 || d == 13) && win == BD_WIN && max_nx >= win && max_ny >= win &&
           max_nx - win + 1 <= BD_MAXN && max_ny - win + 1 <= BD_MAXN;
}

size_t acoss_mask_bits_fused_work_bytes(int K, int max_nx, int max_ny, int win, int side_rows)
{
    if (K < 0 || win < 1 || max_nx < win || max_ny < win || side_rows < 0) return 0;
    BandWork bw;
    return band_work_layout(nullptr, K, max_nx - win + 1, max_ny - win + 1, side_rows, bw) + 256;
}

int acoss_mask_bits_fused_batch(const float *pk, const float *band, const double *feats, const double *norms, int d,
                                const acoss_pair_desc *descs, int K, int win, int max_nx, int max_ny, double kappa,
                                int mutual, uint64_t *bits, void *work, size_t work_bytes, int side_rows, void *stream)
{
    if (!pk || !band || !feats || !norms || !descs || !bits || !work || K < 0 || kappa < 0.0 || side_rows < 1) {
        set_error("mask_bits_fused_batch: bad argument");
        return ACOSS_EINVAL;
    }
    if (!acoss_mask_bits_fused_supported(d, win, max_nx, max_ny)) {
        set_error("mask_bits_fused_batch: supports d in {12, 13}, win == 9 and matrices up to %d x %d", BD_MAXN, BD_MAXN);
        return ACOSS_ENOTSUP;
    }
    if (work_bytes < acoss_mask_bits_fused_work_bytes(K, max_nx, max_ny, win, side_rows)) {
        set_error("mask_bits_fused_batch: workspace too small");
        return ACOSS_EINVAL;
    }
    if (K == 0) return ACOSS_OK;
    const int max_m = max_nx - win + 1, max_n = max_ny - win + 1;
    BandWork bw;
    // 256-byte aligned carve of the caller's buffer
    void *aligned = (void *)(((uintptr_t)work + 255) & ~(uintptr_t)255);
    band_work_layout(aligned, K, max_m, max_n, side_rows, bw);
    bw.band = band;
    hipStream_t st = (hipStream_t)stream;
    double kv;
    int mode;
    if (kappa == 0.0) { kv = 0.0; mode = 2; }        // CRPUtils.py:188-189
    else if (kappa < 1.0) { kv = kappa; mode = 0; }   // :190-191
    else { kv = kappa; mode = 1; }                    // :192-193
    const int bands_m = ceil_div(max_m, BD_R), bands_n = mutual ? ceil_div(max_n, BD_R) : 0;
    const int64_t blocks = (int64_t)K * (bands_m + bands_n);
    if (blocks > 0x7fffffffLL) { set_error("mask_bits_fused_batch: batch too large"); return ACOSS_ENOTSUP; }
    ACOSS_HIP(hipMemsetAsync(bw.counter, 0, 256, st));
    const char *dm = getenv("ACOSS_BAND_MODE");       // development ablations (see the kernel)
    const int dev_mode = dm ? atoi(dm) : 0;
    if (d == 12) hipLaunchKernelGGL(crp_band_kernel<12>, dim3((unsigned)blocks), dim3(64 * BD_WAVES), 0, st, pk, descs, bands_m, bands_n, kv, mode, bw, dev_mode);
    else hipLaunchKernelGGL(crp_band_kernel<13>, dim3((unsigned)blocks), dim3(64 * BD_WAVES), 0, st, pk, descs, bands_m, bands_n, kv, mode, bw, dev_mode);
    int rc = launch_check("crp_band_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(band_fix_kernel, dim3((unsigned)side_rows), dim3(64), 0, st, feats, norms, d, descs, win, kv, mode, bw);
    rc = launch_check("band_fix_kernel");
    if (rc) return rc;
    ThreshWork w;
    w.row_thr = w.col_thr = nullptr;
    w.row_cut = w.col_cut = nullptr;
    w.max_m = max_m;
    w.max_n = max_n;
    w.row_bits = bw.row_bits;
    w.col_bits = bw.col_bits;
    w.wpr = BD_E;
    w.band = band;
    return launch_combine_bits(descs, K, win, mutual, w, bits, st);
}

// rows the last acoss_mask_bits_fused_batch on `work` could not decide in its own kernel (device int, valid once the
// stream has run): more than side_rows means some were dropped and the call must be repeated with a larger side buffer
const int *acoss_mask_bits_fused_counter(void *work)
{
    return (const int *)(((uintptr_t)work + 255) & ~(uintptr_t)255);
}

}  // extern "C"
