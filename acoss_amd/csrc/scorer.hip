// scorer.hip -- group (3) of the C ABI: Serra09.similarity's chroma / MFCC chain (benchmarking/Serra09.py:161-184) for a
// whole pair list in one call.  acoss_corpus_* holds a feature set resident in HBM (float64 features, per-frame norms,
// the centred float32 copy the matrix-core kernels work on); acoss_serra09_scores plans the batches, splits the pairs by
// size class, runs the chain of stage kernels (the same entry points group (2) exports) and returns scores already
// divided by (M + N).  Everything a host binding needs to reach the product path lives here: the Python engine
// (acoss_amd/engine.py: serra09_scores) is one of its callers.
//
// Host side of the pipeline: descriptors and error bands of batch b + 1 are planned while batch b runs (all launches and
// copies are asynchronous on the caller's stream, staging memory is pinned), scores come back through one pinned buffer
// and the call synchronises ONCE, at its end.
#include "common.h"
#include "thresh_work.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

struct acoss_corpus {
    int n_songs = 0, d = 0, nbins = 0;
    int64_t n_frames = 0;
    bool owns = false, f32_ok = false;
    bool force_f64 = false;                     // ACOSS_PLANAR32=0, read once when the handle is made
    bool keys16 = true;                         // ACOSS_KEYS16=0: 32-bit keys for the float32 filter (round 2's form)
    std::mutex call_mutex;                      // the staging below belongs to one acoss_serra09_scores call at a time
    std::vector<int64_t> frame_off;
    std::vector<double> norms_scaled;           // squared norms of the centred, scaled frames (float64, host)
    double *feats = nullptr, *norms = nullptr, *gchroma = nullptr;      // device
    float *f32 = nullptr, *n32 = nullptr;       // device: centred, scaled corpus rounded to float32, and its norms
    std::map<int, std::vector<double>> wmax;    // per window length: largest window sum of norms_scaled per song
    // pinned staging, grown on demand
    void *pin = nullptr;
    size_t pin_bytes = 0;
};

namespace acoss {

// |approx - exact| <= bound_w * (window sums of squared norms) + bound_t * exact for the float32 strip kernel
// (engine.planar32_bound_w): a d-step FMA chain, norms added outside the matrix product
static double bound_w32(int d) { return (d + 4.5) * ldexp(1.0, -24); }
static const double BOUND_T32 = 9.5 * 0.000000059604644775390625;      // 9.5 * 2^-24

static const std::vector<double> &song_wmax(acoss_corpus *c, int win)
{
    auto it = c->wmax.find(win);
    if (it != c->wmax.end()) return it->second;
    std::vector<double> out((size_t)c->n_songs, 0.0);
    for (int s = 0; s < c->n_songs; s++) {
        const int64_t a = c->frame_off[s], b = c->frame_off[s + 1];
        const double *n = c->norms_scaled.data();
        if (b - a >= win) {
            // differences of cumulative sums, as engine.DeviceCorpus.song_wmax forms them
            std::vector<double> cs((size_t)(b - a) + 1, 0.0);
            for (int64_t i = a; i < b; i++) cs[(size_t)(i - a) + 1] = cs[(size_t)(i - a)] + n[i];
            double m = 0.0;
            for (int64_t i = 0; i + win <= b - a; i++) m = std::max(m, cs[(size_t)(i + win)] - cs[(size_t)i]);
            out[(size_t)s] = m * (1.0 + 1e-12);
        } else {
            double m = 0.0;
            for (int64_t i = a; i < b; i++) m += n[i];
            out[(size_t)s] = m;
        }
    }
    return c->wmax.emplace(win, std::move(out)).first->second;
}

static int pin_reserve(acoss_corpus *c, size_t bytes)
{
    if (c->pin_bytes >= bytes) return ACOSS_OK;
    if (c->pin) (void)hipHostFree(c->pin);
    c->pin = nullptr;
    c->pin_bytes = 0;
    if (hipHostMalloc(&c->pin, bytes, hipHostMallocDefault) != hipSuccess) {
        set_error("serra09_scores: pinned staging allocation of %zu bytes failed", bytes);
        return ACOSS_ENOMEM;
    }
    c->pin_bytes = bytes;
    return ACOSS_OK;
}

// size class of a pair: 0 = matrices up to 1024 x 1024 (16 keys per lane), 1 = up to 2048 x 2048, 2 = beyond; 3 = shapes
// the fused strip kernels do not cover (feature width other than 12 / 13 or a window other than 9): one kernel per function
static int pair_class(const acoss_corpus *c, int win, int i, int j)
{
    if ((c->d != 12 && c->d != 13) || win != 9) return 3;
    const int64_t li = c->frame_off[i + 1] - c->frame_off[i], lj = c->frame_off[j + 1] - c->frame_off[j];
    const int64_t side = std::max(li, lj) - win + 1;
    return side > 2048 ? 2 : (side > 1024 ? 1 : 0);
}

struct BatchPlan {
    int cls;
    bool exact = false;         // (class 1) the float64 path, whatever the handle's switches say: pairs the 16-bit keys left unresolved
    std::vector<int> idx;       // positions in the caller's pair list
    int max_nx, max_ny;
};

static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

// device scratch of one batch of class `cls`: {descs, band, xp, T, work, bits / byte mask, C, scores}
struct Carve {
    size_t descs, band, koff, xp, T, work, bits, C, scores, total;
};

static Carve carve(const acoss_corpus *c, int cls, int K, int max_nx, int max_ny, int win, int64_t total_csm, int64_t total_crp)
{
    Carve v;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += al(bytes); return at; };
    const int max_m = max_nx - win + 1;
    v.descs = take(sizeof(acoss_pair_desc) * (size_t)K);
    v.band = take(2 * sizeof(float) * (size_t)K);
    v.koff = take(sizeof(uint32_t) * (size_t)K);
    v.xp = take((cls == 3 ? 0 : (size_t)acoss_xpack_elems(K, max_nx)) * sizeof(double));
    v.T = take(((size_t)std::max<int64_t>(total_crp, 2) + 16) * (cls == 0 || cls == 1 ? 4 : 8));
    const size_t wb = cls <= 1 ? acoss_mask_bits_work_bytes(K, max_nx, max_ny, win) : acoss_binarize_work_bytes(K, max_nx, max_ny, win);
    v.work = take(wb);
    v.bits = take(cls <= 1 ? (size_t)K * max_m * acoss_mask_bits_words(max_nx, max_ny, win) * 8 : (size_t)std::max<int64_t>(total_crp, 1));
    v.C = take(cls == 3 ? (size_t)std::max<int64_t>(total_csm, 1) * 8 : 0);
    v.scores = take(3 * sizeof(float) * (size_t)K);
    v.total = off + 256;
    (void)c;
    return v;
}

// pairs -> batches: by class, then in the caller's order; batch length from the memory a pair of the class needs
static int plan_batches(const acoss_corpus *c, const int32_t *pairs, int K, int win, int batch_pairs, std::vector<BatchPlan> &out)
{
    std::vector<int> by_class[4];
    for (int p = 0; p < K; p++) {
        const int i = pairs[2 * p], j = pairs[2 * p + 1];
        if (i < 0 || j < 0 || i >= c->n_songs || j >= c->n_songs) { set_error("serra09_scores: pair %d names song %d / %d of %d", p, i, j, c->n_songs); return ACOSS_EINVAL; }
        if (c->frame_off[i + 1] - c->frame_off[i] < win || c->frame_off[j + 1] - c->frame_off[j] < win) {
            set_error("serra09_scores: a song of pair %d is shorter than the window", p);
            return ACOSS_EINVAL;
        }
        by_class[pair_class(c, win, i, j)].push_back(p);
    }
    for (int cls = 0; cls < 4; cls++) {
        const std::vector<int> &v = by_class[cls];
        if (v.empty()) continue;
        int64_t longest = 0;
        for (int p : v)
            for (int t = 0; t < 2; t++) longest = std::max(longest, c->frame_off[pairs[2 * p + t] + 1] - c->frame_off[pairs[2 * p + t]]);
        int bp = batch_pairs;
        if (bp <= 0) {
            // ~4096 pairs of 1000-frame songs per launch batch (the one-wave-per-pair alignment kernels want thousands in flight);
            // ragged corpora are budgeted by the cells their pairs really have (mean over the class, the caller's order is
            // arbitrary), plus what every pair costs whatever its size (bit planes, packed x frames, thresholds): up to 8192
            // pairs = eight alignment waves per SIMD
            double cells = 0.0;
            for (int p : v)
                cells += (double)(c->frame_off[pairs[2 * p] + 1] - c->frame_off[pairs[2 * p]]) * (double)(c->frame_off[pairs[2 * p + 1] + 1] - c->frame_off[pairs[2 * p + 1]]);
            cells /= (double)v.size();
            const double per_pair = cls == 3 ? (double)longest * (double)longest * 17.0
                                             : std::min((double)longest * (double)longest, 1.15 * cells) * 8.8 + (double)longest * 400.0;
            bp = (int)std::max(1.0, std::min((double)v.size(), (cls == 3 ? 8.0 : 36.0) * 1073741824.0 / std::max(per_pair, 1.0)));
            if (cls <= 1) bp = std::min(bp, 8192);
            // the alignment kernels run one wave per pair: a batch of 4203 pairs puts a fifth wave on some of the 1024 SIMDs and
            // the launch takes as long as 5120 would; whole multiples of 4096 (of 1024 for smaller budgets) leave no such tail
            if (bp >= 4096 && (int)v.size() > bp) bp -= bp % 4096;
            else if (bp >= 1024 && (int)v.size() > bp) bp -= bp % 1024;
        }
        for (size_t lo = 0; lo < v.size(); lo += (size_t)bp) {
            BatchPlan b;
            b.cls = cls;
            b.idx.assign(v.begin() + (long)lo, v.begin() + (long)std::min(v.size(), lo + (size_t)bp));
            b.max_nx = b.max_ny = 0;
            for (int p : b.idx) {
                b.max_nx = std::max<int>(b.max_nx, (int)(c->frame_off[pairs[2 * p] + 1] - c->frame_off[pairs[2 * p]]));
                b.max_ny = std::max<int>(b.max_ny, (int)(c->frame_off[pairs[2 * p + 1] + 1] - c->frame_off[pairs[2 * p + 1]]));
            }
            out.push_back(std::move(b));
        }
    }
    return ACOSS_OK;
}

static int pitch_of(int cls) { return cls <= 1 ? 32 : 16; }

}  // namespace acoss

using namespace acoss;

extern "C" {

int acoss_corpus_wrap(const double *feats, const double *norms, const double *gchroma, int nbins, const float *f32,
                      const float *n32, const double *norms_scaled, const int64_t *frame_off, int n_songs, int d,
                      acoss_corpus **out)
{
    if (!feats || !norms || !frame_off || !out || n_songs < 1 || d < 1 || (gchroma && nbins < 1)) {
        set_error("corpus_wrap: bad argument");
        return ACOSS_EINVAL;
    }
    acoss_corpus *c = new acoss_corpus();
    c->n_songs = n_songs;
    c->d = d;
    c->nbins = nbins;
    c->frame_off.assign(frame_off, frame_off + n_songs + 1);
    c->n_frames = frame_off[n_songs];
    c->feats = const_cast<double *>(feats);
    c->norms = const_cast<double *>(norms);
    c->gchroma = const_cast<double *>(gchroma);
    c->f32 = const_cast<float *>(f32);
    c->n32 = const_cast<float *>(n32);
    c->f32_ok = f32 && n32 && norms_scaled;
    // the two switches of the product chain (README.md, "Switches"), read once, when the handle is made
    // (the Python engine's parser: "0", "false", "no" and the empty string switch off)
    auto env_off = [](const char *name) {
        const char *e = getenv(name);
        return e && (e[0] == 0 || strcmp(e, "0") == 0 || strcmp(e, "false") == 0 || strcmp(e, "no") == 0);
    };
    c->force_f64 = env_off("ACOSS_PLANAR32");
    c->keys16 = !env_off("ACOSS_KEYS16");
    if (c->f32_ok) c->norms_scaled.assign(norms_scaled, norms_scaled + c->n_frames);
    *out = c;
    return ACOSS_OK;
}

int acoss_corpus_create(const double *feats, const int64_t *frame_off, int n_songs, int d, const double *gchroma, int nbins,
                        acoss_corpus **out)
{
    if (!feats || !frame_off || !out || n_songs < 1 || d < 1 || (gchroma && nbins < 1)) {
        set_error("corpus_create: bad argument");
        return ACOSS_EINVAL;
    }
    const int64_t nf = frame_off[n_songs];
    const size_t ne = (size_t)nf * d;
    // the float32 copy: every entry minus the corpus mean (distances between frames do not change, the squared norms --
    // the scale of the float32 error bound -- shrink to the variance part), scaled by the power of two that brings the
    // largest norm to ~1 (exact in both precisions, order-preserving; no float32 overflow or denormals)
    double mu = 0.0;
    for (size_t i = 0; i < ne; i++) mu += feats[i];
    mu = ne ? mu / (double)ne : 0.0;
    std::vector<double> n64((size_t)nf, 0.0);
    double top = 0.0;
    for (int64_t f = 0; f < nf; f++) {
        double s = 0.0;
        for (int b = 0; b < d; b++) { const double v = feats[f * d + b] - mu; s += v * v; }
        n64[(size_t)f] = s;
        top = std::max(top, s);
    }
    const bool ok = std::isfinite(top) && top > 0.0;
    const double scale = ok ? ldexp(1.0, -(int)lrint(0.5 * log2(top))) : 1.0;
    std::vector<float> f32(ne), n32((size_t)nf);
    for (size_t i = 0; i < ne; i++) f32[i] = (float)((feats[i] - mu) * scale);
    for (int64_t f = 0; f < nf; f++) { n64[(size_t)f] *= scale * scale; n32[(size_t)f] = (float)n64[(size_t)f]; }
    double *dfe = nullptr, *dno = nullptr, *dgc = nullptr;
    float *df32 = nullptr, *dn32 = nullptr;
    auto fail = [&](const char *what) {
        (void)hipFree(dfe); (void)hipFree(dno); (void)hipFree(dgc); (void)hipFree(df32); (void)hipFree(dn32);
        set_error("corpus_create: %s failed", what);
        return ACOSS_ENOMEM;
    };
    if (hipMalloc(&dfe, std::max<size_t>(ne, 1) * 8) != hipSuccess || hipMalloc(&dno, std::max<size_t>((size_t)nf, 1) * 8) != hipSuccess ||
        hipMalloc(&df32, std::max<size_t>(ne, 1) * 4) != hipSuccess || hipMalloc(&dn32, std::max<size_t>((size_t)nf, 1) * 4) != hipSuccess)
        return fail("hipMalloc");
    if (gchroma && hipMalloc(&dgc, (size_t)n_songs * nbins * 8) != hipSuccess) return fail("hipMalloc");
    if (hipMemcpy(dfe, feats, ne * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(df32, f32.data(), ne * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dn32, n32.data(), (size_t)nf * 4, hipMemcpyHostToDevice) != hipSuccess ||
        (gchroma && hipMemcpy(dgc, gchroma, (size_t)n_songs * nbins * 8, hipMemcpyHostToDevice) != hipSuccess))
        return fail("hipMemcpy");
    int rc = acoss_frame_norms_f64(dfe, nf, d, dno, nullptr);
    if (rc == ACOSS_OK && hipDeviceSynchronize() != hipSuccess) rc = ACOSS_EIO;
    if (rc != ACOSS_OK) { (void)fail("frame_norms"); return rc; }
    rc = acoss_corpus_wrap(dfe, dno, dgc, nbins, ok ? df32 : nullptr, ok ? dn32 : nullptr, ok ? n64.data() : nullptr, frame_off, n_songs, d, out);
    if (rc != ACOSS_OK) { (void)fail("wrap"); return rc; }
    (*out)->owns = true;
    if (!ok) { (void)hipFree(df32); (void)hipFree(dn32); }
    return ACOSS_OK;
}

void acoss_corpus_destroy(acoss_corpus *c)
{
    if (!c) return;
    if (c->owns) {
        (void)hipFree(c->feats); (void)hipFree(c->norms); (void)hipFree(c->gchroma); (void)hipFree(c->f32); (void)hipFree(c->n32);
    }
    if (c->pin) (void)hipHostFree(c->pin);
    delete c;
}

size_t acoss_serra09_scratch_bytes(const acoss_corpus *c, const int32_t *pairs, int K, int win, int batch_pairs)
{
    if (!c || !pairs || K < 0 || win < 1) return 0;
    std::vector<BatchPlan> plan;
    if (plan_batches(c, pairs, K, win, batch_pairs, plan) != ACOSS_OK) return 0;
    // sized from the cells the batches really hold (acoss_plan_pairs, as acoss_serra09_scores will plan them) -- every pair at its
    // batch's largest shape would ask for three to four times the budget on length-skewed corpora
    size_t need = 256;
    std::vector<int32_t> bp;
    std::vector<acoss_pair_desc> hd;
    for (const BatchPlan &b : plan) {
        const int B = (int)b.idx.size();
        bp.resize(2 * (size_t)B);
        hd.resize((size_t)B);
        for (int t = 0; t < B; t++) { bp[2 * (size_t)t] = pairs[2 * b.idx[(size_t)t]]; bp[2 * (size_t)t + 1] = pairs[2 * b.idx[(size_t)t] + 1]; }
        int64_t csm = 0, crp = 0;
        if (acoss_plan_pairs(c->frame_off.data(), c->n_songs, bp.data(), B, win, pitch_of(b.cls), hd.data(), &csm, &crp) != ACOSS_OK) return 0;
        need = std::max(need, carve(c, b.cls, B, b.max_nx, b.max_ny, win, csm, crp).total);
    }
    return need;
}

int acoss_serra09_scores(acoss_corpus *c, const int32_t *pairs, int K, int win, double kappa, int do_oti, int want,
                         int batch_pairs, void *scratch, size_t scratch_bytes, double *qmax, double *dmax, double *swc,
                         void *stream)
{
    if (!c || !pairs || K < 0 || win < 1 || kappa < 0.0 || (want & ~7) || !want ||
        ((want & 1) && !qmax) || ((want & 2) && !dmax) || ((want & 4) && !swc)) {
        set_error("serra09_scores: bad argument (want: 1 = qmax, 2 = dmax, 4 = swalignimpconstrained, each with its output array)");
        return ACOSS_EINVAL;
    }
    if (do_oti && !c->gchroma) { set_error("serra09_scores: do_oti without global chroma in the corpus"); return ACOSS_EINVAL; }
    if (K == 0) return ACOSS_OK;
    // calls on one handle are serialised (they share its pinned staging); calls on different handles run concurrently
    std::lock_guard<std::mutex> one_call_at_a_time(c->call_mutex);
    std::vector<BatchPlan> plan;
    int rc = plan_batches(c, pairs, K, win, batch_pairs, plan);
    if (rc != ACOSS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    // pinned staging: every batch's descriptors and bands (read by asynchronous copies until the stream gets there), the scores
    // (twice K: pairs of 1033 .. 2056 frames the 16-bit keys leave unresolved -- exact ties -- come round again in batches of their own)
    const size_t pin_descs = al(sizeof(acoss_pair_desc) * 2 * (size_t)K), pin_band = al(16 * (size_t)K), pin_scores = al(24 * (size_t)K), pin_koff = al(8 * (size_t)K);
    rc = pin_reserve(c, pin_descs + pin_band + pin_scores + pin_koff);
    if (rc != ACOSS_OK) return rc;
    acoss_pair_desc *h_descs = (acoss_pair_desc *)c->pin;
    float *h_band = (float *)((char *)c->pin + pin_descs);
    float *h_scores = (float *)((char *)c->pin + pin_descs + pin_band);      // [3][K] in batch order
    uint32_t *h_koff = (uint32_t *)((char *)c->pin + pin_descs + pin_band + pin_scores);
    char *base = (char *)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    const size_t avail = scratch_bytes - (size_t)(base - (char *)scratch);
    const int nq = (want & 1) ? 1 : 0, nd = (want & 2) ? 1 : 0, ns = (want & 4) ? 1 : 0;
    size_t done = 0;
    std::vector<int32_t> unresolved;
    for (size_t bi = 0; bi < plan.size(); bi++) {
        const BatchPlan b = plan[bi];                   // (a copy: the loop appends to `plan`)
        const int B = (int)b.idx.size(), cls = b.cls;
        std::vector<int32_t> bp(2 * (size_t)B);
        for (int t = 0; t < B; t++) { bp[2 * (size_t)t] = pairs[2 * b.idx[(size_t)t]]; bp[2 * (size_t)t + 1] = pairs[2 * b.idx[(size_t)t] + 1]; }
        acoss_pair_desc *hd = h_descs + done;
        int64_t tcsm = 0, tcrp = 0;
        rc = acoss_plan_pairs(c->frame_off.data(), c->n_songs, bp.data(), B, win, pitch_of(cls), hd, &tcsm, &tcrp);
        if (rc != ACOSS_OK) return rc;
        const Carve v = carve(c, cls, B, b.max_nx, b.max_ny, win, tcsm, tcrp);
        if (v.total > avail) { set_error("serra09_scores: scratch too small (%zu bytes needed, acoss_serra09_scratch_bytes)", v.total); return ACOSS_EINVAL; }
        acoss_pair_desc *d_descs = (acoss_pair_desc *)(base + v.descs);
        float *d_band = (float *)(base + v.band), *d_scores = (float *)(base + v.scores);
        ACOSS_HIP(hipMemcpyAsync(d_descs, hd, sizeof(acoss_pair_desc) * (size_t)B, hipMemcpyHostToDevice, st));
        if (do_oti) { rc = acoss_oti_batch(c->gchroma, c->nbins, d_descs, B, st); if (rc) return rc; }
        // 16-bit keys up to 2048 x 2048: class 1 takes them when the radix selection is on (its long form; the wave-per-row kernels
        // behind it stop at 1024)
        const bool long16 = cls == 1 && !b.exact && c->f32_ok && !c->force_f64 && c->keys16 && acoss_radix16_enabled();
        const bool use32 = (cls == 0 && c->f32_ok && !c->force_f64) || long16;
        uint64_t *bits = (uint64_t *)(base + v.bits);
        if (cls <= 1) {
            if (use32) {
                const std::vector<double> &w = song_wmax(c, win);
                float *hb = h_band + 2 * done;
                const double up = 1.0 + ldexp(1.0, -10);        // covers the float32 rounding of base, slope and base + slope * v
                for (int t = 0; t < B; t++) {
                    const double band2[2] = {2.0 * bound_w32(c->d) * (w[(size_t)bp[2 * (size_t)t]] + w[(size_t)bp[2 * (size_t)t + 1]]) * up, 2.0 * BOUND_T32 * up};
                    for (int u = 0; u < 2; u++) {
                        float f = (float)band2[u];
                        if ((double)f < band2[u]) f = nextafterf(f, INFINITY);
                        hb[2 * t + u] = f;
                    }
                }
                ACOSS_HIP(hipMemcpyAsync(d_band, hb, 8 * (size_t)B, hipMemcpyHostToDevice, st));
                float *xp = (float *)(base + v.xp);
                rc = acoss_pack_x_f32(c->f32, c->n32, c->d, d_descs, B, b.max_nx, xp, st);
                if (c->keys16) {
                    // 16-bit keys (csrc/keys16.h): koff = the pattern of the float32 not below 2 W, seven octaves down
                    uint32_t *hk = h_koff + done, *d_koff = (uint32_t *)(base + v.koff);
                    for (int t = 0; t < B; t++) {
                        const double w2 = 2.0 * (w[(size_t)bp[2 * (size_t)t]] + w[(size_t)bp[2 * (size_t)t + 1]]);
                        float f = (float)w2;
                        if ((double)f < w2) f = nextafterf(f, INFINITY);
                        uint32_t fb;
                        memcpy(&fb, &f, 4);
                        hk[t] = (std::isfinite(f) && f > 0x1p-100f)      /* the same limit as engine.keys16_koff */ ? fb - (7u << 23) : 0u;
                    }
                    ACOSS_HIP(hipMemcpyAsync(d_koff, hk, 4 * (size_t)B, hipMemcpyHostToDevice, st));
                    if (!rc) rc = acoss_crp_keys16_batch(xp, c->f32, c->n32, c->d, d_descs, B, win, b.max_nx, b.max_ny, d_koff, (uint16_t *)(base + v.T), st);
                    if (!rc) rc = acoss_mask_bits_keys16_batch((const uint16_t *)(base + v.T), d_band, d_koff, xp, c->f32, c->n32, c->feats, c->norms, c->d,
                                                               d_descs, B, win, b.max_nx, b.max_ny, kappa, 1, bits, base + v.work, v.bits - v.work, st);
                } else {
                    if (!rc) rc = acoss_crp_planar32_batch(xp, c->f32, c->n32, c->d, d_descs, B, win, b.max_nx, b.max_ny, (uint32_t *)(base + v.T), st);
                    if (!rc) rc = acoss_mask_bits_planar32_batch((const uint32_t *)(base + v.T), d_band, c->feats, c->norms, c->d, d_descs, B, win, b.max_nx,
                                                                 b.max_ny, kappa, 1, bits, base + v.work, v.bits - v.work, st);
                }
            } else {
                double *xp = (double *)(base + v.xp);
                rc = acoss_pack_x_f64(c->feats, c->norms, c->d, d_descs, B, b.max_nx, xp, st);
                if (!rc) rc = acoss_crp_planar_batch_f64(xp, c->feats, c->norms, c->d, d_descs, B, win, b.max_nx, b.max_ny, (uint32_t *)(base + v.T), st);
                if (!rc) rc = acoss_mask_bits_planar_batch((const uint32_t *)(base + v.T), c->feats, c->norms, c->d, d_descs, B, win, b.max_nx, b.max_ny,
                                                           kappa, 1, bits, base + v.work, v.bits - v.work, st);
            }
            if (rc) return rc;
            // Serra09.py:173-175: qmax, then dmax on the D qmax leaves behind (boundary = 1)
            if (nq && nd) rc = acoss_align_bits_qd_batch(bits, d_descs, B, win, b.max_nx, b.max_ny, 1, nullptr, d_scores, d_scores + B, st);
            else if (nq) rc = acoss_align_bits_batch(0, bits, d_descs, B, win, b.max_nx, b.max_ny, 0, nullptr, d_scores, st);
            else if (nd) rc = acoss_align_bits_batch(1, bits, d_descs, B, win, b.max_nx, b.max_ny, 1, nullptr, d_scores + B, st);
            if (!rc && ns) rc = acoss_align_bits_batch(2, bits, d_descs, B, win, b.max_nx, b.max_ny, 0, nullptr, d_scores + 2 * B, st);
            if (rc) return rc;
            if (long16) {
                // pairs the radix selection could not express: once more, on the float64 path, in a batch of their own (it is a
                // subset of this one: the scratch fits); their scores overwrite the ones just computed from undefined masks
                unresolved.resize((size_t)B);
                int n_un = 0;
                rc = acoss_mask_bits_keys16_unresolved(base + v.work, B, b.max_nx, b.max_ny, win, unresolved.data(), B, &n_un, st);
                if (rc) return rc;
                if (n_un > 0) {
                    BatchPlan r;
                    r.cls = 1;
                    r.exact = true;
                    r.max_nx = b.max_nx;
                    r.max_ny = b.max_ny;
                    for (int t = 0; t < n_un; t++) r.idx.push_back(b.idx[(size_t)unresolved[(size_t)t]]);
                    plan.push_back(std::move(r));
                }
            }
        } else {
            // byte mask: long songs (fused strip kernel, any-length selection) or shapes without a fused kernel (one kernel per function)
            double *T = (double *)(base + v.T);
            uint8_t *Bm = (uint8_t *)(base + v.bits);
            if (cls == 2) {
                double *xp = (double *)(base + v.xp);
                rc = acoss_pack_x_f64(c->feats, c->norms, c->d, d_descs, B, b.max_nx, xp, st);
                if (!rc) rc = acoss_crp_batch_f64(xp, c->feats, c->norms, c->d, d_descs, B, win, b.max_nx, b.max_ny, 0, T, st);
            } else {
                double *C = (double *)(base + v.C);
                rc = acoss_csm_batch_f64(c->feats, c->norms, c->d, d_descs, B, b.max_nx, b.max_ny, C, st);
                if (!rc) rc = acoss_sliding_batch_f64(C, d_descs, B, win, b.max_nx, b.max_ny, T, st);
            }
            if (rc) return rc;
            ACOSS_HIP(hipMemsetAsync(Bm, 0, (size_t)std::max<int64_t>(tcrp, 1), st));
            rc = acoss_binarize_batch(T, d_descs, B, win, b.max_nx, b.max_ny, kappa, 1, Bm, base + v.work, v.bits - v.work, st);
            if (rc) return rc;
            // alignment descriptors in the (now free) window-sum buffer
            std::vector<acoss_mat_desc> mats((size_t)B);
            for (int t = 0; t < B; t++) {
                mats[(size_t)t].s_off = hd[t].crp_off; mats[(size_t)t].d_off = 0;
                mats[(size_t)t].rows = hd[t].nx - win + 1; mats[(size_t)t].cols = hd[t].ny - win + 1;
                mats[(size_t)t].s_pitch = hd[t].crp_pitch; mats[(size_t)t].d_pitch = 0;
            }
            acoss_mat_desc *d_mats = (acoss_mat_desc *)T;
            if ((size_t)B * sizeof(acoss_mat_desc) > v.work - v.T) { set_error("serra09_scores: internal carve"); return ACOSS_EIO; }
            ACOSS_HIP(hipMemcpyAsync(d_mats, mats.data(), sizeof(acoss_mat_desc) * (size_t)B, hipMemcpyHostToDevice, st));
            ACOSS_HIP(hipStreamSynchronize(st));        // (mats is a local: rare path, long songs only)
            const int max_cols = b.max_ny - win + 1;
            if (nq) rc = acoss_qmax_batch(Bm, d_mats, B, max_cols, nullptr, nullptr, d_scores, st);
            if (!rc && nd) rc = acoss_dmax_batch(Bm, d_mats, B, max_cols, nullptr, 1, nullptr, d_scores + B, st);
            if (!rc && ns) rc = acoss_swc_batch(Bm, d_mats, B, max_cols, nullptr, nullptr, d_scores + 2 * B, st);
            if (rc) return rc;
        }
        float *hs = h_scores + 3 * done;
        ACOSS_HIP(hipMemcpyAsync(hs, d_scores, 12 * (size_t)B, hipMemcpyDeviceToHost, st));
        done += (size_t)B;
    }
    ACOSS_HIP(hipStreamSynchronize(st));
    done = 0;
    for (const BatchPlan &b : plan) {
        const int B = (int)b.idx.size();
        const float *hs = h_scores + 3 * done;
        const acoss_pair_desc *hd = h_descs + done;
        for (int t = 0; t < B; t++) {
            const double denom = (double)(hd[t].nx - win + 1) + (double)(hd[t].ny - win + 1);       // Serra09.py:174-175: / (M + N)
            const int p = b.idx[(size_t)t];
            if (nq) qmax[p] = (double)hs[t] / denom;
            if (nd) dmax[p] = (double)hs[B + t] / denom;
            if (ns) swc[p] = (double)hs[2 * B + t] / denom;
        }
        done += (size_t)B;
    }
    return ACOSS_OK;
}

}  // extern "C"
