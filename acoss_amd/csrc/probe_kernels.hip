// probe_kernels.hip -- development probes (not part of the public ABI): store-pattern bandwidth.
// How fast can a kernel write a batch of row-major float64 matrices (the CSM output) depending on WHICH bytes the
// concurrently running workgroups touch?  All modes write every element of K matrices of rows x cols once, 16 bytes
// per lane and store instruction.
//   mode 0: linear  -- workgroup b writes the contiguous chunk b of the whole buffer (what a fill kernel does)
//   mode 1: tiles   -- workgroup = one 128 x 128 tile (128 row pieces of 1 KB at a pitch of cols * 8 bytes)
//   mode 2: strips  -- workgroup = one 120-column strip of a matrix, walking down 32 rows at a time (960-byte pieces)
//   mode 3: bands   -- workgroup = 16 full rows of a matrix (one contiguous 16 * cols * 8 byte region)
//   mode 4: band of 64 rows, column-chunk-major -- wave w takes the 128-column chunks w, w+4, ... and writes the 64 row
//           pieces of a chunk before moving on (a y-stationary kernel that owns whole rows)
//   mode 6: the planar form: a uint32 matrix (4 bytes per cell, rows x cols cells in the first half of each matrix's
//           space), 120-column strips walked down, one 8-byte store per lane: 480-byte row pieces
//   mode 7: the same with 240-column strips, one 16-byte store per lane: 960-byte row pieces
//   mode 5: band of 32 rows walked in 120-column steps -- each step writes 32 pieces of 960 bytes (the strip kernel
//           turned by 90 degrees)
#include "common.h"
#include "kernel_utils.h"

namespace acoss {

__global__ __launch_bounds__(256) void store_probe_kernel(double *__restrict__ out, int K, int rows, int cols, int mode,
                                                          int per_mat)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const double2 v = make_double2(1.0 + tid, 2.0);
    if (mode == 0) {
        const int64_t total = (int64_t)K * rows * cols;
        const int64_t chunk = 256 * 2 * 64;                 // 256 KB per workgroup
        const int64_t base = (int64_t)lb * chunk;
        for (int it = 0; it < 64; it++) {
            const int64_t idx = base + (int64_t)it * 512 + tid * 2;
            if (idx + 1 < total) *reinterpret_cast<double2 *>(out + idx) = v;
        }
        return;
    }
    const int p = lb / per_mat, t = lb % per_mat;
    double *M = out + (int64_t)p * rows * cols;
    if (mode == 1) {
        const int tn = (cols + 127) / 128;
        const int r0 = (t / tn) * 128, c0 = (t % tn) * 128;
        const int c = c0 + (tid & 63) * 2;                   // a wave = one 1 KB row piece
        for (int r = r0 + (tid >> 6); r < min(r0 + 128, rows); r += 4)
            if (c + 1 < cols) *reinterpret_cast<double2 *>(M + (int64_t)r * cols + c) = v;
    } else if (mode == 2) {
        const int c0 = t * 120;
        const int c = c0 + (tid & 63) * 2;
        const bool on = (tid & 63) < 60 && c + 1 < cols;
        for (int r = tid >> 6; r < rows; r += 4)
            if (on) *reinterpret_cast<double2 *>(M + (int64_t)r * cols + c) = v;
    } else if (mode == 3) {
        const int r0 = t * 16;
        const int64_t n = (int64_t)min(16, rows - r0) * cols;
        double *B = M + (int64_t)r0 * cols;
        for (int64_t i = tid * 2; i + 1 < n; i += 512) *reinterpret_cast<double2 *>(B + i) = v;
    } else if (mode == 4) {
        const int r0 = t * 64, r1 = min(r0 + 64, rows);
        for (int c0 = (tid >> 6) * 128; c0 < cols; c0 += 512) {
            const int c = c0 + (tid & 63) * 2;
            for (int r = r0; r < r1; r++)
                if (c + 1 < cols) *reinterpret_cast<double2 *>(M + (int64_t)r * cols + c) = v;
        }
    } else if (mode == 10) {
        // uint32 matrix, a band of 32 rows walked right in 112-column steps (448-byte pieces, the strip kernel turned by 90 degrees)
        uint32_t *U = reinterpret_cast<uint32_t *>(out) + (int64_t)p * rows * cols;
        const int r0 = t * 32, r1 = min(r0 + 32, rows);
        for (int c0 = 0; c0 < cols; c0 += 112) {
            const int c = c0 + (tid & 63) * 2;
            const bool on = (tid & 63) < 56 && c + 1 < cols;
            for (int r = r0 + (tid >> 6); r < r1; r += 4)
                if (on) *reinterpret_cast<uint2 *>(U + (int64_t)r * cols + c) = make_uint2(tid, r);
        }
    } else if (mode == 9) {
        // float64, 112-column strips: 896-byte row pieces = 7 whole 128-byte lines when the pitch is line-aligned
        const int c = t * 112 + (tid & 63) * 2;
        const bool on = (tid & 63) < 56 && c + 1 < cols;
        for (int r = tid >> 6; r < rows; r += 4)
            if (on) *reinterpret_cast<double2 *>(M + (int64_t)r * cols + c) = v;
    } else if (mode == 8) {
        // 112-column strips of a uint32 matrix: 448-byte row pieces, every piece 64-byte aligned when the pitch is
        uint32_t *U = reinterpret_cast<uint32_t *>(out) + (int64_t)p * rows * cols;
        const int c = t * 112 + (tid & 63) * 2;
        const bool on = (tid & 63) < 56 && c + 1 < cols;
        for (int r = tid >> 6; r < rows; r += 4)
            if (on) *reinterpret_cast<uint2 *>(U + (int64_t)r * cols + c) = make_uint2(tid, r);
    } else if (mode == 6 || mode == 7) {
        uint32_t *U = reinterpret_cast<uint32_t *>(out) + (int64_t)p * rows * cols;        // 4 bytes per cell
        const int w = mode == 6 ? 120 : 240, per = mode == 6 ? 2 : 4;
        const int c = t * w + (tid & 63) * per;
        const bool on = (tid & 63) < 60 && c + per - 1 < cols;
        for (int r = tid >> 6; r < rows; r += 4) {
            if (on) {
                if (mode == 6) *reinterpret_cast<uint2 *>(U + (int64_t)r * cols + c) = make_uint2(tid, r);
                else *reinterpret_cast<uint4 *>(U + (int64_t)r * cols + c) = make_uint4(tid, r, 3, 4);
            }
        }
    } else {
        const int r0 = t * 32, r1 = min(r0 + 32, rows);
        for (int c0 = 0; c0 < cols; c0 += 120) {
            const int c = c0 + (tid & 63) * 2;
            const bool on = (tid & 63) < 60 && c + 1 < cols;
            for (int r = r0 + (tid >> 6); r < r1; r += 4)
                if (on) *reinterpret_cast<double2 *>(M + (int64_t)r * cols + c) = v;
        }
    }
}

}  // namespace acoss

using namespace acoss;

extern "C" int acoss_dev_store_probe(double *out, int K, int rows, int cols, int mode, void *stream)
{
    if (!out || K < 1 || rows < 1 || cols < 2 || (cols & 3) || mode < 0 || mode > 10) { set_error("store_probe: bad argument"); return ACOSS_EINVAL; }
    int per_mat = 1;
    int64_t blocks;
    if (mode == 0) blocks = ((int64_t)K * rows * cols + 256 * 2 * 64 - 1) / (256 * 2 * 64);
    else {
        per_mat = mode == 1 ? ((rows + 127) / 128) * ((cols + 127) / 128) : (mode == 2 ? (cols + 119) / 120 : (mode == 3 ? (rows + 15) / 16 : (mode == 4 ? (rows + 63) / 64 : (mode == 5 ? (rows + 31) / 32 : (mode == 6 ? (cols + 119) / 120 : (mode == 7 ? (cols + 239) / 240 : (cols + 111) / 112))))));
        if (mode == 9) per_mat = (cols + 111) / 112;
        if (mode == 10) per_mat = (rows + 31) / 32;
        blocks = (int64_t)K * per_mat;
    }
    if (blocks > 0x7fffffffLL) { set_error("store_probe: too large"); return ACOSS_ENOTSUP; }
    hipLaunchKernelGGL(store_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, K, rows, cols, mode, per_mat);
    return launch_check("store_probe_kernel");
}
