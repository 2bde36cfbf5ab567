// band_kernels.hip -- get_csm + sliding_csm + csm_to_binary (CRPUtils.py:67-84, :24-45, :169-219) in ONE kernel:
// the windowed sums never go to HBM.
//
// Why: the materialising fast path writes the 3.9 MB key matrix of a pair once and reads it twice (rows, columns):
// 53 GB per 4096-pair step against 0.8 GB of features.  Here a block owns bands of 24 rows of a pair over all of
// its columns: it forms a band's squared distances on the matrix cores chunk by chunk (128 columns), sums the
// 9-long diagonal windows out of LDS straight into registers -- 16 keys per lane and row, the layout the histogram
// selection works on -- selects each row's k-th smallest key and leaves only the row's bit plane (128 bytes).  The
// column planes come from the same kernel run on the swapped pair (rows of (y, x) = columns of (x, y)); the two
// plane sets are ANDed by combine_planes_kernel.
//
// Arithmetic: float32 on v_mfma_f32_16x16x4_f32 (a k-ordered chain of round-to-nearest FMAs).  The squared norms ride
// in the matrix product: A = [-2 x | |x|^2 | 1], B = [y | 1 | |y|^2], so the accumulator leaves as |x|^2 + |y|^2 - 2 x.y
// and the epilogue is one max(., 0); window sums in the order k = 0..8.  |T~ - T| <= 2^-24 (17.5 W + 9.5 T) (W = window
// sums of the squared norms; derivation in DESIGN.md section 4).  A row whose k-th smallest key has another key inside
// its error band is not decided here: its keys go to a compact side buffer and band_fix_kernel finishes it exactly
// in float64 (planar_select.h), so the masks equal the float64 path's bit for bit.
//
// Work decomposition (gfx950): 8 waves per block, 2 blocks per CU; a block walks a run of consecutive bands of one
// (pair, orientation), so the per-block set-up is paid once per run and each band's selection starts from the
// thresholds of the band above it.
//   phase A (per 128-column chunk): wave v computes the 32 x 16 tile of columns 16v .. 16v+15 (two 16-row MFMA blocks
//     share the y fragment) into the chunk buffer in LDS.  The buffer keeps two interleaved column streams (columns u
//     and u + 64 in adjacent words) so that phase B reads the elements of two diagonals 64 columns apart with one
//     aligned 8-byte ds_read per row.  Operands reach the waves as whole 64-byte frames (16-byte loads, through LDS).
//   phase B: wave v owns output rows 3v .. 3v+2: a lane walks its two diagonals down 11 C rows (11 ds_read_b64) and
//     forms 3 x 2 window sums -> registers h[q][2 chunk + stream].  One LDS-only barrier per chunk (chunk buffers are
//     double-buffered, the 10 carried columns are copied across).
//   selection: the wave's three rows in lockstep through ONE histogram pass (band_select3: 512 bins around the hint,
//     three 10-bit counters per word), so the LDS round trips -- what a selection waits for -- are paid once per three
//     rows; raw 64-bit ballots leave as the row's plane (bit l of word e = column 64 e + l - d, d = 10 - row mod 3:
//     the diagonal walk's offset; combine_planes_kernel shifts it out).
#include "planar_select.h"

namespace acoss {

typedef float bd_v4f32 __attribute__((ext_vector_type(4)));
typedef float bd_v2f32 __attribute__((ext_vector_type(2)));

constexpr int BD_WIN = 9, BD_HALO = BD_WIN - 1;
constexpr int BD_WAVES = 8;
constexpr int BD_RPW = 3;                       // output rows per wave
constexpr int BD_R = BD_WAVES * BD_RPW;         // 24 output rows per band
constexpr int BD_CROWS = BD_R + BD_HALO;        // 32 C rows = two 16-row MFMA blocks
constexpr int BD_HC = BD_HALO + BD_RPW - 1;     // columns carried from one chunk to the next (10)
constexpr int BD_CHUNK = 128;
constexpr int BD_SW = 64 + BD_HC;               // entries per column stream
constexpr int BD_PITCH = 2 * BD_SW;             // floats per C row in LDS (even: 8-byte reads stay aligned)
constexpr int BD_E = 16;                        // key registers per lane and row
constexpr int BD_MAXN = 64 * BD_E - BD_HC;      // longest row the register layout holds (1014)
constexpr int BD_MAXCH = (BD_MAXN + BD_HC + BD_CHUNK - 1) / BD_CHUNK;      // 8 chunks
constexpr int BD_XP = 16;                       // floats per packed frame: [d bins | squared norm | 1 | 0 ...]
constexpr int BD_SIDE_WORDS = 64 * BD_E;        // words per side-buffer slot
constexpr int BD_FLD = 20;                      // floats per staged frame in LDS (16 + 4: see the kernel)
constexpr unsigned BD_INVALID = 0x7fffffffu;    // key of a position outside the row (above every float32 >= +0)

static_assert(BD_CROWS == 32, "two MFMA row blocks");
static_assert(BD_MAXCH == 8, "chunk loop is unrolled 8 times");

__host__ __device__ inline int bd_plane_shift(int row) { return BD_HC - (row % BD_R) % BD_RPW; }

// Where the band kernel puts rows it could not decide, and what the fix-up kernel needs to find them.
struct BandWork {
    uint64_t *row_bits;     // [K][max_m][16]    planes (shifted, see the file header)
    uint64_t *col_bits;     // [K][16][max_n]    word-major planes of the columns
    int max_m, max_n;
    int *counter;           // [0]: slots asked for (may exceed cap: the caller re-runs with a larger side buffer)
    int4 *slots;            // [cap]: {pair, orientation, row, key of the tentative threshold}
    uint32_t *side;         // [cap][BD_SIDE_WORDS]: the row's keys in column order (float32 bits | sign bit)
    int cap;
    const float *band;      // per pair (base, slope) of the error band (engine.planar32_band)
};

__device__ inline int bd_srcbin(int bin, int shift, int d)
{
    // np.roll(chroma, oti, axis=0) (Serra09.py:167): rolled[b] = orig[(b - oti) mod d]; the norm slots and the padding stay
    if (bin >= d) return bin;
    int s = bin - shift;
    return s < 0 ? s + d : s;
}

// ---------------------------------------------------------------------------------------------------------------
// One row on its own (the first band of a block, and rows the lockstep pass below could not place).
// k-th smallest (1-based) of the wave's keys h[e] (positions outside the row hold BD_INVALID, every valid key is a
// float32 >= +0 bit pattern).  true: `thr` is the k-th smallest and no other key equals it; false: undecided here
// (`thr` = the key the candidates share, or 0).
//
// Histogram selection as in planar_select.h with a single-level histogram of 256 bins per wave: lane l owns bins
// 4l .. 4l+3, so the counters are read and cleared with one conflict-free 16-byte access per lane.  The 1-3 keys of the
// winning bin are ranked by v_readlane; two of them in one lane, or a cold start on the full range, re-bin that bin
// 256x finer.
// ---------------------------------------------------------------------------------------------------------------
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
                mx = max(mx, h[e] == BD_INVALID ? 0u : h[e]);
            }
            mn = wave_umin(mn);
            mx = wave_umax(mx);
            lo = mn;
            shift = mx > mn ? max(0, 32 - (int)__clz(mx - mn) - BH_LOG2) : 0;
        }
        // the positions outside the row must fall above the window (always, unless the values reach 2^120)
        if (((BD_INVALID - lo) >> shift) < (unsigned)BH_BINS) return false;
        // keys below lo wrap to >= 2^31 and land, like the keys above the window, in the lane's spill word (a wave-wide
        // ds_add on 64 consecutive words runs at the full LDS atomic rate, 4.2 cycles per instruction and CU:
        // tools/ubench/lds_atomic.hip; masking lanes off costs more than it saves).  The number of keys below lo comes
        // from the borrow of the subtraction.
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

// thr is the k-th smallest key of the row and unique: it stands if no other key lies in its error band -- exactly k keys
// <= bhi and k - 1 keys < blo (counting form; the lockstep pass below decides most rows without it)
__device__ inline bool band_alone_by_count(const unsigned (&h)[BD_E], int k, unsigned thr, const float *pair_band)
{
    unsigned blo, bhi;
    bd_band_limits(thr, pair_band, blo, bhi);
    int ca = 0, cb = 0;
#pragma unroll
    for (int e = 0; e < BD_E; e++) {
        ca += __popcll(__ballot(h[e] <= bhi));
        cb += __popcll(__ballot(h[e] < blo));
    }
    return (ca == k) & (cb == k - 1);
}

// ---------------------------------------------------------------------------------------------------------------
// The wave's three rows in lockstep.  All three share the window [lo, lo + 512 << 15) of key space centred on `hint`
// (a threshold of a nearby row: one binade either way, thresholds of rows of one pair differ by less than that), and
// ONE histogram: word b + 64 counts bin b for every row in its own 10-bit field (a row has at most 1014 keys), words
// 0 .. 63 collect the keys below the window, words 576 .. 639 those above it (per-lane words: no two lanes of an
// instruction on one address).  One pass = 48 atomics, one wait, one counter read (2 x 16 bytes per lane), a packed
// scan, per row a second-level read of 8 counters; the winning bin's 1-3 keys are found by a range test and ranked.
// A row is DECIDED here when its k-th smallest key is unique and alone in its error band: with the band inside the
// winning bin only that bin's keys can lie in it.  Rows that miss the window, have two candidates in one lane or a band
// across a bin edge are left to the caller (found / decided bits clear).
// ---------------------------------------------------------------------------------------------------------------
constexpr int BM_LOG2 = 9, BM_BINS = 1 << BM_LOG2, BM_SHIFT = 15;
constexpr int BM_WORDS = 64 + BM_BINS + 64;

__device__ inline int bd_med3_i32(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

// inclusive scan over lanes 0 .. 7 of their row of 16 (other lanes: don't care)
__device__ inline int bd_scan8(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR2, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR4, 0xf, 0xf, false);
    return v;
}

__device__ inline void band_select3(const unsigned (&h)[BD_RPW][BD_E], const int k, const unsigned hint, unsigned *hist,
                                    const int lane, const float *pair_band, unsigned (&thr)[BD_RPW], unsigned &found,
                                    unsigned &decided)
{
    static_assert(BD_RPW == 3, "three 10-bit counters per word");
    found = 0;
    decided = 0;
    const unsigned half = (unsigned)(BM_BINS / 2) << BM_SHIFT;
    const unsigned lo = max(hint, half) - half;
    if (((BD_INVALID - lo) >> BM_SHIFT) < (unsigned)BM_BINS) return;      // positions outside the rows must fall above the window
    const int lo_t = lane - 64, hi_t = BM_BINS + lane;
#pragma unroll
    for (int s = 0; s < BD_RPW; s++) {
#pragma unroll
        for (int e = 0; e < BD_E; e++) {
            // bin number, negative below the window; clamped into the lane's spill words either side
            int t = (int)(h[s][e] - lo) >> BM_SHIFT;
            t = bd_med3_i32(t, lo_t, hi_t);
            atomicAdd(&hist[64 + t], 1u << (10 * s));
        }
    }
    // lane l: bins 8l .. 8l+7
    const uint4 wa = reinterpret_cast<const uint4 *>(hist + 64)[2 * lane], wb = reinterpret_cast<const uint4 *>(hist + 64)[2 * lane + 1];
    const unsigned bel = hist[lane];
    const unsigned sum8 = wa.x + wa.y + wa.z + wa.w + wb.x + wb.y + wb.z + wb.w;       // field-wise: a row's total < 1024
    const unsigned incl = (unsigned)wave_scan<OpAdd>((int)sum8, 0);
    const unsigned belsum = (unsigned)wave_reduce<OpAdd>((int)bel, 0);
    int ls[BD_RPW], r0[BD_RPW];
    unsigned c2w[BD_RPW];
    bool ok[BD_RPW];
#pragma unroll
    for (int s = 0; s < BD_RPW; s++) {
        const int kk = k - (int)((belsum >> (10 * s)) & 1023u);
        const int tot = (int)((sum8 >> (10 * s)) & 1023u), inc = (int)((incl >> (10 * s)) & 1023u);
        const uint64_t m1 = __ballot((inc - tot < kk) & (kk <= inc));
        ok[s] = m1 != 0;
        ls[s] = __builtin_amdgcn_readfirstlane(ok[s] ? __ffsll((unsigned long long)m1) - 1 : 0);
        // keys before lane ls's bins (the packed difference has no borrows: incl >= sum8 in every field)
        const unsigned before = (unsigned)__builtin_amdgcn_readlane((int)(incl - sum8), ls[s]);
        r0[s] = kk - (int)((before >> (10 * s)) & 1023u);
        c2w[s] = hist[64 + 8 * ls[s] + (lane & 7)];
    }
    reinterpret_cast<uint4 *>(hist + 64)[2 * lane] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4 *>(hist + 64)[2 * lane + 1] = make_uint4(0, 0, 0, 0);
    hist[lane] = 0;
    hist[64 + BM_BINS + lane] = 0;
    unsigned lo2[BD_RPW];
    int cstar[BD_RPW], r[BD_RPW];
#pragma unroll
    for (int s = 0; s < BD_RPW; s++) {
        const int c2 = (int)((c2w[s] >> (10 * s)) & 1023u);
        const int inc2 = bd_scan8(c2);
        const uint64_t m2 = __ballot((lane < 8) & (inc2 >= r0[s]));
        ok[s] = ok[s] & (m2 != 0);
        const int ts = __builtin_amdgcn_readfirstlane(m2 != 0 ? __ffsll((unsigned long long)m2) - 1 : 0);
        cstar[s] = __builtin_amdgcn_readlane(c2, ts);
        r[s] = r0[s] - (__builtin_amdgcn_readlane(inc2, ts) - cstar[s]);
        lo2[s] = lo + ((unsigned)(8 * ls[s] + ts) << BM_SHIFT);
    }
    // the keys of the winning bins: one per lane or the row goes to the caller
    uint64_t any[BD_RPW];
    unsigned ch[BD_RPW];
#pragma unroll
    for (int s = 0; s < BD_RPW; s++) {
        any[s] = 0;
        ch[s] = 0;
        const unsigned lo2s = (unsigned)__builtin_amdgcn_readfirstlane((int)lo2[s]);
#pragma unroll
        for (int e = 0; e < BD_E; e++) {
            const bool in = (h[s][e] - lo2s) < (1u << BM_SHIFT);
            any[s] |= __ballot(in);
            ch[s] = in ? h[s][e] : ch[s];
        }
        ok[s] = ok[s] & (__popcll(any[s]) == cstar[s]);
    }
#pragma unroll
    for (int s = 0; s < BD_RPW; s++) {
        if (!ok[s]) continue;        // wave-uniform
        const bool mine = (any[s] >> lane) & 1;
        int less = 0, equal = 1;
        if (cstar[s] > 1) {
            equal = 0;
            for (uint64_t rest = any[s]; rest != 0; rest &= rest - 1) {
                const int c = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)rest) - 1);
                const unsigned vc = (unsigned)__builtin_amdgcn_readlane((int)ch[s], c);
                less += vc < ch[s];
                equal += vc == ch[s];
            }
        }
        const uint64_t win = __ballot(mine & (less < r[s]) & (r[s] <= less + equal));
        if (win == 0) continue;
        const int wl = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)win) - 1);
        const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)ch[s], wl);
        thr[s] = t;
        if (__builtin_amdgcn_readlane(equal, wl) > 1) {         // equal keys: exact values decide
            found |= 9u << s;
            continue;
        }
        found |= 1u << s;
        unsigned blo, bhi;
        bd_band_limits(t, pair_band, blo, bhi);
        if ((blo >= lo2[s]) & (bhi < lo2[s] + (1u << BM_SHIFT))) {
            // the band lies inside the winning bin: only that bin's keys can be in it
            const uint64_t near = __ballot(mine & ((ch[s] - blo) <= (bhi - blo)));
            if (__popcll(near) == 1) decided |= 1u << s;
            else found |= 8u << s;          // another key shares the band: undecidable here whatever the caller counts
        }
    }
}

// Diagnostic build of the band kernel (STAMP): wave-cycles per phase, summed over the launch (tools/band_stamps.py).
constexpr int BD_NSTAMP = 12;
__device__ unsigned long long g_band_stamps[BD_NSTAMP];

template <int D, bool STAMP = false>
__global__ __launch_bounds__(64 * BD_WAVES, 4) void crp_band_kernel(const float *__restrict__ pk,
                                                                    const acoss_pair_desc *__restrict__ descs,
                                                                    int runs_m, int runs_n, int run_bands, double kv, int k_mode,
                                                                    BandWork bw, int mode)
{
    // mode (development ablations, product = 0): 1 = no selection, 8 = exit at once
    constexpr int KA = D + 2;                   // contraction depth with the two norm columns
    constexpr int KSTEPS = (KA + 3) / 4;
    static_assert(KA <= BD_XP, "packed frame holds the augmented vector");
    // LDS: the two chunk buffers, per wave the packed frames of its 16 columns of a chunk, the band's 32 x frames, the
    // column-plane staging of orientation 1 and the waves' histograms
    __shared__ __attribute__((aligned(16))) float cbuf_raw[2 * BD_CROWS * BD_PITCH];
    __shared__ __attribute__((aligned(16))) float ybuf_all[BD_WAVES * 16 * BD_FLD];
    __shared__ __attribute__((aligned(16))) float abuf[BD_CROWS * BD_FLD];
    __shared__ __attribute__((aligned(16))) uint64_t obuf[BD_E * BD_R];
    __shared__ __attribute__((aligned(16))) unsigned hist_all[BD_WAVES * BM_WORDS];
    static_assert(BM_WORDS >= BH_WORDS, "the single-row histogram shares the wave's words");

    const int per_pair = runs_m + runs_n;
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / per_pair, t = lb % per_pair;
    const int orient = t >= runs_m ? 1 : 0;
    const acoss_pair_desc ds = descs[p];
    // rows of this orientation: frames of A; columns: frames of B
    const int nA = orient ? ds.ny : ds.nx, nB = orient ? ds.nx : ds.ny;
    const int64_t rowA0 = orient ? ds.y_row0 : ds.x_row0, rowB0 = orient ? ds.x_row0 : ds.y_row0;
    const int shiftA = orient ? 0 : ds.shift, shiftB = orient ? ds.shift : 0;
    const int Mo = nA - BD_HALO, No = nB - BD_HALO;
    const int band_lo = (orient ? t - runs_m : t) * run_bands;
    const int band_hi = min(band_lo + run_bands, (Mo + BD_R - 1) / BD_R);
    if (band_lo >= band_hi) return;
    if (mode & 8) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nchunks = min(BD_MAXCH, (No + BD_HC + BD_CHUNK - 1) / BD_CHUNK);
    const bool a_loader = threadIdx.x < 4 * BD_CROWS;
    auto gload_a = [&](const int band) {
        const int tid = threadIdx.x;
        return *reinterpret_cast<const float4 *>(pk + (rowA0 + min(band * BD_R + (tid >> 2), nA - 1)) * BD_XP + 4 * (tid & 3));
    };
    if (a_loader) *reinterpret_cast<float4 *>(abuf + (threadIdx.x >> 2) * BD_FLD + 4 * (threadIdx.x & 3)) = gload_a(band_lo);
    {
        unsigned *hz = hist_all + wave * BM_WORDS;
        for (int i = threadIdx.x & 63; i < BM_WORDS; i += 64) hz[i] = 0;
    }
    const int k = knn_count(k_mode, kv, No);
    const float *pair_band = bw.band + 2 * p;
    unsigned hint = 0;          // threshold key of a row of the band above (0: none yet)
    unsigned long long st_acc[BD_NSTAMP] = {}, st_prev = 0;
    auto stamp = [&](const int phase) {
        if constexpr (STAMP) {
            unsigned long long now;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            if (phase >= 0) st_acc[phase] += now - st_prev;
            st_prev = now;
        }
    };
    stamp(-1);

    for (int band = band_lo; band < band_hi; band++) {
        const int i0 = band * BD_R;
        // Per-lane constants are formed here, every band, from a laundered thread id: kept across the band loop they
        // (and everything hipcc hoists with them) cost ~40 registers and spills in the selection.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int lr = lane & 15, lk = lane >> 4;
        unsigned *hist = hist_all + wave * BM_WORDS;
        // ---- operands: whole packed frames (64 bytes) by 16-byte loads, 4 lanes per frame, through LDS (row stride 20
        // floats: the fragment reads are 2-way bank conflicts instead of 8-way); fragments are read with the OTI rotation.
        // A side: bins scaled by -2 (exact), then |x|^2, then the packed 1; B side: bins, then the packed 1, then |y|^2.
        float *const ybuf = ybuf_all + wave * (16 * BD_FLD);
        const int st_off = (lane >> 2) * BD_FLD + 4 * (lane & 3);
        auto gload_b = [&](const int ch) {
            return *reinterpret_cast<const float4 *>(pk + (rowB0 + min(BD_CHUNK * ch + 16 * wave + (lane >> 2), nB - 1)) * BD_XP + 4 * (lane & 3));
        };
        int boff[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            const int kx = 4 * s + lk;          // index into the augmented vector: B side kx < D: bin; D: the 1 slot; D + 1: norm slot
            boff[s] = lr * BD_FLD + (kx < D ? bd_srcbin(kx, shiftB, D) : (kx == D ? D + 1 : (kx == D + 1 ? D : min(kx, BD_XP - 1))));
        }
        // LDS addresses.  Chunk-local column u = HC + 16 wave + lr of the new columns; stream 0 holds u in [0, SW) at
        // word 2u, stream 1 holds u in [64, 64 + SW) at word 2 (u - 64) + 1; columns in both ranges are written twice
        // (wave 3 only: its second copy sits 127 words below the first).
        const int u = BD_HC + 16 * wave + lr;
        const bool dual = (wave == 3) & (u >= 64);
        const int wr_off = (4 * lk) * BD_PITCH + (wave < 4 ? 2 * u : 2 * (u - 64) + 1);
        const int rd_off = (BD_RPW * wave) * BD_PITCH + 2 * lane;
        // carried columns: u in [128, 128 + HC) of this chunk = u - 128 of the next one (129 words below)
        const bool carrier = tid < BD_CROWS * BD_HC;
        const int c_row = tid / BD_HC;
        const int carry_src = c_row * BD_PITCH + 2 * (64 + tid - c_row * BD_HC) + 1;

        float4 bq = gload_b(0);
        stamp(0);
        __syncthreads();            // the band's x frames are in LDS; every wave is done with the band above
        stamp(1);
        float afrag[2][KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            // A side of the augmented vector: kx < D: bin (rotated) x -2; D: norm slot; D + 1: the 1 slot
            const int kx = 4 * s + lk;
            const int aoff = kx < D ? bd_srcbin(kx, shiftA, D) : min(kx, BD_XP - 1);
            const float ascale = kx < D ? -2.0f : (kx < KA ? 1.0f : 0.0f);
#pragma unroll
            for (int rb = 0; rb < 2; rb++) afrag[rb][s] = ascale * abuf[(16 * rb + lr) * BD_FLD + aoff];
        }
        float bfrag[KSTEPS];
        // the wave's tile of chunk c: registers -> its LDS slot -> fragments (wave-private: ordered by the wave's own waits)
        auto stage_b = [&](const float4 q) {
            *reinterpret_cast<float4 *>(ybuf + st_off) = q;
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) {
                const float v = ybuf[boff[s]];
                bfrag[s] = 4 * s + lk < KA ? v : 0.0f;
            }
        };
        stage_b(bq);
        if (nchunks > 1) bq = gload_b(1);
        stamp(2);

        unsigned h[BD_RPW][BD_E];
#pragma unroll
        for (int q = 0; q < BD_RPW; q++) {
#pragma unroll
            for (int e = 0; e < BD_E; e++) h[q][e] = BD_INVALID;
        }

        // The chunk loop is software-pipelined: between two barriers a wave issues the window-sum reads of chunk j, runs the
        // matrix-core work of chunk j + 1 while they travel, stages the y tile of chunk j + 2, forms the sums of chunk j and
        // only then writes the C tile of chunk j + 1 (into the other chunk buffer).
        bd_v4f32 acc[2];
        auto mfma_tile = [&]() {
            acc[0] = (bd_v4f32){0.f, 0.f, 0.f, 0.f};
            acc[1] = (bd_v4f32){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) {
#pragma unroll
                for (int rb = 0; rb < 2; rb++)
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[rb][s], bfrag[s], acc[rb], 0, 0, 0);
            }
        };
        auto write_tile = [&](float *const cb) {
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float c = fmaxf(acc[rb][r], 0.0f);
                    cb[wr_off + (16 * rb + r) * BD_PITCH] = c;
                    if (dual) cb[wr_off - 127 + (16 * rb + r) * BD_PITCH] = c;
                }
            }
        };
        // C rows [i0, i0 + 32) x columns [0, 128) of chunk 0, and the fragments of chunk 1
        mfma_tile();
        write_tile(cbuf_raw);
        if (nchunks > 1) {
            stage_b(bq);
            if (nchunks > 2) bq = gload_b(2);
        }
        stamp(3);
        lds_barrier();
        stamp(4);
#pragma unroll
        for (int ch = 0; ch < BD_MAXCH; ch++) {
            if (ch < nchunks) {          // block-uniform
                float *const cb = cbuf_raw + (ch & 1) * (BD_CROWS * BD_PITCH);
                float *const cn = cbuf_raw + ((ch + 1) & 1) * (BD_CROWS * BD_PITCH);
                // ---- window sums of chunk ch: rows 3 wave + q, columns 128 ch - HC + 64 s + lane + q; the reads first
                bd_v2f32 v[BD_RPW + BD_HALO];
                float carried = 0.0f;
#pragma unroll
                for (int m = 0; m < BD_RPW + BD_HALO; m++)
                    v[m] = *reinterpret_cast<const bd_v2f32 *>(cb + rd_off + m * (BD_PITCH + 2));
                if (ch + 1 < nchunks && carrier) carried = cb[carry_src];
                // ---- matrix-core work of chunk ch + 1, then the fragments of chunk ch + 2
                if (ch + 1 < nchunks) {
                    mfma_tile();
                    if (ch + 2 < nchunks) {
                        stage_b(bq);
                        if (ch + 3 < nchunks) bq = gload_b(ch + 3);
                    }
                }
                stamp(5);
                {
                    // every output sums its 9 elements in the order k = 0 .. 8
#pragma unroll
                    for (int q = 0; q < BD_RPW; q++) {
                        bd_v2f32 s = v[q];
#pragma unroll
                        for (int kk = 1; kk < BD_WIN; kk++) s += v[q + kk];
                        h[q][2 * ch] = __float_as_uint(s.x);
                        h[q][2 * ch + 1] = __float_as_uint(s.y);
                    }
                    if (ch + 1 < nchunks && carrier) cn[carry_src - 129] = carried;
                }
                stamp(6);
                if (ch + 1 < nchunks) {
                    write_tile(cn);
                    stamp(3);
                    lds_barrier();
                    stamp(4);
                }
            }
        }

        // ---- selection: row i0 + 3 wave + q; position (e, lane) of its registers is column 64 e + lane - (HC - q)
        // (the next band's x frames travel meanwhile; every wave read this band's at the top, so their LDS slots are free)
        float4 a_next = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_loader && band + 1 < band_hi) a_next = gload_a(band + 1);
        const int irow0 = i0 + BD_RPW * wave;
        // positions outside the rows (before column 0, from column No on) get the largest key
#pragma unroll
        for (int q = 0; q < BD_RPW; q++) {
            const int dq = BD_HC - q;
#pragma unroll
            for (int e = 0; e < BD_E; e++) {
                if (64 * e - dq < 0 || 64 * e + 63 - dq >= No) {
                    const int col = 64 * e + lane - dq;
                    h[q][e] = ((col >= 0) & (col < No)) ? h[q][e] : BD_INVALID;
                }
            }
        }
        const bool trivial = (k <= 0) | (k >= No);
        unsigned thr3[BD_RPW] = {0u, 0u, 0u}, found = 0, decided = 0;
        if (!trivial && !(mode & 1) && irow0 < Mo) {
            if (hint == 0) {
                // nothing to start from: the middle row on its own, from the full range
                HistWarm warm{0, BH_SHIFT0};
                unsigned t1;
                if (band_select(h[1], k, hist, lane, warm, t1)) hint = t1;
                else {
                    HistWarm warm0{0, BH_SHIFT0};
                    if (band_select(h[0], k, hist, lane, warm0, t1)) hint = t1;
                }
            }
            stamp(7);
            if (hint != 0) band_select3(h, k, hint, hist, lane, pair_band, thr3, found, decided);
        }
        stamp(8);
#pragma unroll
        for (int q = 0; q < BD_RPW; q++) {
            const int i = irow0 + q;
            if (i < Mo) {            // wave-uniform
                const int dq = BD_HC - q;
                unsigned(&hq)[BD_E] = h[q];
                uint64_t word[BD_E];
                if (k <= 0) {
#pragma unroll
                    for (int e = 0; e < BD_E; e++) word[e] = 0ull;
                } else if (k >= No) {
#pragma unroll
                    for (int e = 0; e < BD_E; e++) word[e] = __ballot(hq[e] != BD_INVALID);
                } else if (mode & 1) {
#pragma unroll
                    for (int e = 0; e < BD_E; e++) word[e] = __ballot(hq[e] <= hq[0]);
                } else {
                    unsigned thr = thr3[q];
                    bool dec = (decided >> q) & 1;
                    if (!((found >> q) & 1)) {
                        // not placed by the lockstep pass: on its own, then the counting form of the band test
                        HistWarm warm{hint, BH_SHIFT0 + 2};
                        dec = band_select(hq, k, hist, lane, warm, thr) && band_alone_by_count(hq, k, thr, pair_band);
                    } else if (!dec && !((found >> (q + 3)) & 1)) {
                        dec = band_alone_by_count(hq, k, thr, pair_band);      // band across a bin edge
                    }
                    // (wave-uniform by construction; said so to the compiler, or the ballots below end up in vector registers)
                    dec = __builtin_amdgcn_readfirstlane((int)dec) != 0;
                    thr = (unsigned)__builtin_amdgcn_readfirstlane((int)thr);
                    if (dec) {
                        if (q == 1 || hint == 0) hint = thr;
#pragma unroll
                        for (int e = 0; e < BD_E; e++) word[e] = __ballot(hq[e] <= thr);
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
                // the 16 ballots -> lanes 0 .. 15, four at a time (one s_nop covers the v_cmp -> v_writelane hazard of a group)
                unsigned wlo = 0, whi = 0;
#pragma unroll
                for (int g = 0; g < BD_E; g += 4) {
                    asm volatile("s_nop 3" : "+s"(word[g]), "+s"(word[g + 1]), "+s"(word[g + 2]), "+s"(word[g + 3]));
#pragma unroll
                    for (int e = g; e < g + 4; e++) {
                        asm("v_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                            : "+v"(wlo), "+v"(whi)
                            : "s"((unsigned)word[e]), "s"((unsigned)(word[e] >> 32)), "n"(e));
                    }
                }
                const uint64_t mine = ((uint64_t)whi << 32) | wlo;
                if (orient == 0) {
                    if (lane < BD_E) bw.row_bits[((int64_t)p * bw.max_m + i) * BD_E + lane] = mine;
                } else {
                    if (lane < BD_E) obuf[lane * BD_R + BD_RPW * wave + q] = mine;
                }
            }
        }
        stamp(9);
        if (a_loader && band + 1 < band_hi) *reinterpret_cast<float4 *>(abuf + (threadIdx.x >> 2) * BD_FLD + 4 * (threadIdx.x & 3)) = a_next;
        if (orient) {
            // the band's 24 columns x 16 words leave word-major: 16 runs of 24 consecutive words
            __syncthreads();
            if (tid < BD_E * BD_R) {
                const int e = tid / BD_R, rl = tid - e * BD_R;
                if (i0 + rl < Mo) bw.col_bits[((int64_t)p * BD_E + e) * bw.max_n + i0 + rl] = obuf[tid];
            }
        }
        stamp(10);
    }
    if constexpr (STAMP) {
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < BD_NSTAMP; i++) atomicAdd(&g_band_stamps[i], st_acc[i]);
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
    const int sh = bd_plane_shift(which);
    if (orient == 0) {
        if (fix_row_band<0, BD_E>(sm, key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, nullptr, nullptr, lane, sh)) return;
        fix_row_generic<0, BD_E>(key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, nullptr, nullptr, lane, sh);
    } else {
        if (fix_row_band<1, BD_E>(sm, key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, nullptr, nullptr, lane, sh)) return;
        fix_row_generic<1, BD_E>(key_at, thr_hi, feats, norms, d, ds, win, w, p, which, len, k, nullptr, nullptr, lane, sh);
    }
}

// Bit-packed mutual mask from the planes of the band kernel: out[p][i][cw] (uint64, bit c = column cw*64 + c) = the
// row plane of row i AND the transposed column planes, both with their plane shift taken out.  One block per 64 rows of
// a pair (cf. combine_bits_kernel, crp_kernels.hip: same staging and the same 64 x 64 butterfly transpose).
__device__ inline uint64_t bd_transpose64(uint64_t x, int lane) { return wave_transpose64(x, lane); }      // wave_ops.h

__global__ __launch_bounds__(256) void combine_planes_kernel(const acoss_pair_desc *__restrict__ descs, int win, int mutual,
                                                             BandWork bw, int tiles_m, uint64_t *__restrict__ out)
{
    constexpr int W = BD_E, LD = W + 1;
    __shared__ uint64_t rowbuf[64 * LD];
    const int p = blockIdx.x / tiles_m, ri = blockIdx.x % tiles_m;
    const acoss_pair_desc ds = descs[p];
    const int M = ds.nx - win + 1, N = ds.ny - win + 1;
    if (ri * 64 >= M) return;
    const int rows = min(64, M - ri * 64);
    const int64_t base = ((int64_t)p * bw.max_m + ri * 64) * W;          // the block's rows are contiguous: rows * W words
    for (int idx = threadIdx.x; idx < rows * W; idx += 256) {
        const int rl = idx / W, wd = idx % W;
        const int sh = bd_plane_shift(ri * 64 + rl);
        const uint64_t a = bw.row_bits[base + idx], b = wd + 1 < W ? bw.row_bits[base + idx + 1] : 0ull;
        rowbuf[rl * LD + wd] = (a >> sh) | (b << (64 - sh));
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (mutual) {
        for (int cw = wave; cw * 64 < N; cw += 4) {
            const int j = cw * 64 + lane;
            uint64_t cwd = 0ull;
            if (j < N) {
                const int sh = bd_plane_shift(j);
                const uint64_t *cp = bw.col_bits + ((int64_t)p * W + ri) * bw.max_n + j;
                const uint64_t a = cp[0], b = ri + 1 < W ? cp[bw.max_n] : 0ull;
                cwd = (a >> sh) | (b << (64 - sh));
            }
            const uint64_t tr = bd_transpose64(cwd, lane);
            if (lane < rows) rowbuf[lane * LD + cw] &= tr;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < rows * W; idx += 256) out[base + idx] = rowbuf[(idx / W) * LD + idx % W];
}

// packed float32 frames: [d values | squared norm | 1 | zeros] per frame, 64 bytes
__global__ void pack_frames_kernel(const float *__restrict__ feats, const float *__restrict__ norms, int d, int64_t n_frames,
                                   float *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t f = g >> 4;
    const int c = (int)(g & 15);
    if (f >= n_frames) return;
    out[g] = c < d ? feats[f * d + c] : (c == d ? norms[f] : (c == d + 1 ? 1.0f : 0.0f));
}

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
    if (!feats || !norms || !out || d < 1 || d > 14 || n_frames < 0) { set_error("pack_frames_f32: bad argument (1 <= d <= 14)"); return ACOSS_EINVAL; }
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
    // a block walks a run of consecutive bands: long enough to amortise its set-up and to hand thresholds down, short
    // enough that a batch still has thousands of blocks (small batches get shorter runs)
    const int bands_m = ceil_div(max_m, BD_R), bands_n = ceil_div(max_n, BD_R);
    int run_bands = 14;
    while (run_bands > 1 && (int64_t)K * (ceil_div(bands_m, run_bands) + (mutual ? ceil_div(bands_n, run_bands) : 0)) < 4096) run_bands = (run_bands + 1) / 2;
    const int runs_m = ceil_div(bands_m, run_bands), runs_n = mutual ? ceil_div(bands_n, run_bands) : 0;
    const int64_t blocks = (int64_t)K * (runs_m + runs_n);
    if (blocks > 0x7fffffffLL) { set_error("mask_bits_fused_batch: batch too large"); return ACOSS_ENOTSUP; }
    ACOSS_HIP(hipMemsetAsync(bw.counter, 0, 256, st));
    const int dev_mode = 0;      // (development ablations of the kernel: tools removed in round 4)
    if (d == 12) hipLaunchKernelGGL(crp_band_kernel<12>, dim3((unsigned)blocks), dim3(64 * BD_WAVES), 0, st, pk, descs, runs_m, runs_n, run_bands, kv, mode, bw, dev_mode);
    else hipLaunchKernelGGL(crp_band_kernel<13>, dim3((unsigned)blocks), dim3(64 * BD_WAVES), 0, st, pk, descs, runs_m, runs_n, run_bands, kv, mode, bw, dev_mode);
    int rc = launch_check("crp_band_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(band_fix_kernel, dim3((unsigned)side_rows), dim3(64), 0, st, feats, norms, d, descs, win, kv, mode, bw);
    rc = launch_check("band_fix_kernel");
    if (rc) return rc;
    const int tm = ceil_div(max_m, 64);
    if ((int64_t)K * tm > 0x7fffffffLL) { set_error("mask_bits_fused_batch: batch too large"); return ACOSS_ENOTSUP; }
    hipLaunchKernelGGL(combine_planes_kernel, dim3((unsigned)((int64_t)K * tm)), dim3(256), 0, st, descs, win, mutual, bw, tm, bits);
    return launch_check("combine_planes_kernel");
}


// rows the last acoss_mask_bits_fused_batch on `work` could not decide in its own kernel (device int, valid once the
// stream has run): more than side_rows means some were dropped and the call must be repeated with a larger side buffer
const int *acoss_mask_bits_fused_counter(void *work)
{
    return (const int *)(((uintptr_t)work + 255) & ~(uintptr_t)255);
}

}  // extern "C"
