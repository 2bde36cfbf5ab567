// capi.hip -- library-level pieces of the C ABI: error text, device selection, the host-side
// pair planner, and the reference's own native interface (qmax_c / dmax_c /
// swalignimpconstrained on HOST pointers, benchmarking/pySeqAlign.pxd:3-10) executed on the GPU.
#include "common.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace acoss {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

long nneighbs(double kappa, long ncols)
{
    if (kappa == 0.0) return ncols;
    if (kappa < 1.0) return (long)rint(kappa * (double)ncols);
    return (long)kappa;
}

// RAII device buffer for the synchronous host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu) failed", bytes);
            return ACOSS_ENOMEM;
        }
        return ACOSS_OK;
    }
};

typedef int (*batch_fn)(const uint8_t *, const acoss_mat_desc *, int, int, float *, float *);

// One matrix through a batched alignment kernel with the reference's in-place-D contract.
static float align_host(unsigned char *S, float *D, int rows, int cols, int d_rows, int d_cols, int kind)
{
    if (!S || !D) { set_error("alignment: null buffer"); return NAN; }
    if (rows < 1 || cols < 1) return 0.0f;   // the reference returns 0.0 for degenerate sizes
    const size_t sbytes = (size_t)rows * cols, dbytes = sizeof(float) * (size_t)d_rows * d_cols;
    DevBuf ds, dd, dm, dsc;
    if (ds.alloc(sbytes) || dd.alloc(dbytes) || dm.alloc(sizeof(acoss_mat_desc)) || dsc.alloc(sizeof(float))) return NAN;
    acoss_mat_desc md;
    md.s_off = 0; md.d_off = 0; md.rows = rows; md.cols = cols; md.s_pitch = cols; md.d_pitch = d_cols;
    if (hipMemcpy(ds.p, S, sbytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dd.p, D, dbytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dm.p, &md, sizeof(md), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("alignment: host-to-device copy failed");
        return NAN;
    }
    int rc;
    if (kind == 0) rc = acoss_qmax_batch((const uint8_t *)ds.p, (const acoss_mat_desc *)dm.p, 1, cols, (float *)dd.p, nullptr, (float *)dsc.p, nullptr);
    else if (kind == 1) rc = acoss_dmax_batch((const uint8_t *)ds.p, (const acoss_mat_desc *)dm.p, 1, cols, (float *)dd.p, 0, nullptr, (float *)dsc.p, nullptr);
    else rc = acoss_swc_batch((const uint8_t *)ds.p, (const acoss_mat_desc *)dm.p, 1, cols, (float *)dd.p, nullptr, (float *)dsc.p, nullptr);
    if (rc != ACOSS_OK) return NAN;
    float score = NAN;
    if (hipMemcpy(&score, dsc.p, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(D, dd.p, dbytes, hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("alignment: device-to-host copy failed: %s", hipGetErrorString(hipGetLastError()));
        return NAN;
    }
    return score;
}

}  // namespace acoss

using namespace acoss;

namespace acoss {

int device_cus()
{
    static thread_local int cached_dev = -1, cached_cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 256; }
        cached_dev = dev;
        cached_cus = n;
    }
    return cached_cus;
}

}  // namespace acoss

extern "C" {

int acoss_abi_version(void) { return ACOSS_ABI_VERSION; }

const char *acoss_last_error(void) { return g_err; }

int acoss_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipGetDeviceCount failed (no HIP device visible)");
        return ACOSS_EIO;
    }
    return n;
}

int acoss_set_device(int device)
{
    ACOSS_HIP(hipSetDevice(device));
    return ACOSS_OK;
}

void acoss_default_align_params(acoss_align_params *p)
{
    if (!p) return;
    p->gamma_onset = 0.5f;       // SequenceAlignment.c:105
    p->gamma_extension = 0.5f;   // :106
    p->sw_match = 1.0f;          // :57
    p->sw_mismatch = -1.0f;      // :58
    p->sw_gap_open = -0.5f;      // :45
    p->sw_gap_ext = -0.7f;       // :46
}

int acoss_plan_pairs(const int64_t *frame_off, int n_songs, const int32_t *pairs, int K, int win,
                     int pitch_align, acoss_pair_desc *descs, int64_t *total_csm, int64_t *total_crp)
{
    if (!frame_off || !pairs || !descs || n_songs < 1 || K < 0 || win < 1 || pitch_align < 1) {
        set_error("plan_pairs: bad argument");
        return ACOSS_EINVAL;
    }
    int64_t csm = 0, crp = 0;
    for (int p = 0; p < K; p++) {
        const int a = pairs[2 * p], b = pairs[2 * p + 1];
        if (a < 0 || a >= n_songs || b < 0 || b >= n_songs) {
            set_error("plan_pairs: pair %d references song outside [0, %d)", p, n_songs);
            return ACOSS_EINVAL;
        }
        const int64_t nx = frame_off[a + 1] - frame_off[a], ny = frame_off[b + 1] - frame_off[b];
        if (nx < win || ny < win || nx > 0x7fffffff || ny > 0x7fffffff) {
            set_error("plan_pairs: pair %d has a song shorter than the window (%lld, %lld frames, win %d)",
                      p, (long long)nx, (long long)ny, win);
            return ACOSS_EINVAL;
        }
        acoss_pair_desc &d = descs[p];
        memset(&d, 0, sizeof(d));
        d.x_row0 = frame_off[a];
        d.y_row0 = frame_off[b];
        d.nx = (int32_t)nx;
        d.ny = (int32_t)ny;
        d.song_x = a;
        d.song_y = b;
        d.csm_pitch = (int32_t)(ceil_div64(ny, pitch_align) * pitch_align);
        d.crp_pitch = (int32_t)(ceil_div64(ny - win + 1, pitch_align) * pitch_align);
        d.csm_off = csm;
        d.crp_off = crp;
        csm += nx * d.csm_pitch;
        crp += (nx - win + 1) * d.crp_pitch;
        // keep every matrix base aligned like its rows
        csm = ceil_div64(csm, pitch_align) * pitch_align;
        crp = ceil_div64(crp, pitch_align) * pitch_align;
    }
    if (total_csm) *total_csm = csm;
    if (total_crp) *total_crp = crp;
    return ACOSS_OK;
}

// SequenceAlignment.c:113
float qmax_c(unsigned char *S, float *D, int M, int N)
{
    if (M < 3 || N < 3) return 0.0f;
    return align_host(S, D, M, N, M, N, 0);
}

// SequenceAlignment.c:147
float dmax_c(unsigned char *S, float *D, int M, int N)
{
    if (M < 4 || N < 4) return 0.0f;
    return align_host(S, D, M, N, M, N, 1);
}

// SequenceAlignment.c:73 (argument order N rows, M columns)
float swalignimpconstrained(unsigned char *S, float *D, int N, int M)
{
    if (N + 1 < 4 || M + 1 < 4) return 0.0f;
    return align_host(S, D, N, M, N + 1, M + 1, 2);
}

}  // extern "C"
