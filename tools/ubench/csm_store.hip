// Micro-benchmark (dev tool, round 4): pure store patterns for the float64 cross-similarity matrix (get_csm, CRPUtils.py:67-84):
// K pairs x 1000 x 1000 float64 at a row pitch of 1024 elements (8 KB).  No arithmetic.
//   S: column strips of 128 cells, the block walks DOWN in 32-row steps (crp_strip_kernel<12,1,sqrt> today)
//   B: row bands, the block walks RIGHT in 128-cell chunks
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/csm_store.hip -o tools/ubench/csm_store && tools/ubench/csm_store [pairs]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NR = 1000, NC = 1000, PITCH = 8192;
constexpr int WORD3 = 0x00020000;

__device__ inline int xcd_remap(int b, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// MODE 0: lanes interleaved (even lanes cols l, l+1; odd lanes cols 63+l, 64+l), 16 B per lane [today]
// MODE 1: lane l -> cols 2l, 2l+1 (1 KB contiguous per instruction)
template <int MODE, int POLICY>
__global__ __launch_bounds__(512) void strip_k(unsigned char *out, int strips)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / strips, s = lb % strips;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = s * 128;
    unsigned char *base = out + (size_t)p * NR * PITCH;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, NR * PITCH, WORD3);
    const int col = MODE == 0 ? ((lane & 1) ? 63 + lane : lane) : 2 * lane;
    for (int t = 0; t < (NR + 31) / 32; t++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = t * 32 + 4 * wave + q;
            const unsigned v = (unsigned)(row * 977 + lane);
            if (row < NR && j0 + col + 1 < NC) __builtin_amdgcn_raw_buffer_store_b128((u32x4){v, v + 1, v + 2, v + 3}, rs, 8 * col, row * PITCH + 8 * j0, POLICY);
        }
    }
}

// band of BR rows (8 waves x BR / 8 rows), chunks of 128 cells; MODE 0: one b128 per lane (1 KB per instruction);
// MODE 1: two b64 per lane (2 x 512 B); MODE 2: chunks of 64 cells, one b64 per lane (512 B per instruction)
template <int BR, int MODE, int POLICY>
__global__ __launch_bounds__(512) void band_k(unsigned char *out, int bands)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / bands, b = lb % bands;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *base = out + (size_t)p * NR * PITCH;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, NR * PITCH, WORD3);
    constexpr int RPW = BR / 8;
    constexpr int CW = MODE == 2 ? 64 : 128;
    for (int t = 0; t < (NC + CW - 1) / CW; t++) {
#pragma unroll
        for (int q = 0; q < RPW; q++) {
            const int row = b * BR + RPW * wave + q;
            if (row >= NR) continue;
            const unsigned v = (unsigned)(row * 977 + lane);
            if (MODE == 0) {
                const int col = CW * t + 2 * lane;
                if (col + 1 < NC) __builtin_amdgcn_raw_buffer_store_b128((u32x4){v, v + 1, v + 2, v + 3}, rs, 8 * col, row * PITCH, POLICY);
            } else if (MODE == 1) {
                const int c0 = CW * t + lane, c1 = c0 + 64;
                if (c0 < NC) __builtin_amdgcn_raw_buffer_store_b64((u32x2){v, v + 1}, rs, 8 * c0, row * PITCH, POLICY);
                if (c1 < NC) __builtin_amdgcn_raw_buffer_store_b64((u32x2){v, v + 1}, rs, 8 * c1, row * PITCH, POLICY);
            } else {
                const int c0 = CW * t + lane;
                if (c0 < NC) __builtin_amdgcn_raw_buffer_store_b64((u32x2){v, v + 1}, rs, 8 * c0, row * PITCH, POLICY);
            }
        }
    }
}

// the matrix-core layout stored as it is (no LDS re-layout): a block owns a band of BR rows and walks right in 128-cell chunks,
// wave w = cells 16w .. 16w+15 of the chunk; lane (lr, lk) holds C[16 u + 4 lk + r][lr] of row tile u: one b64 store per (u, r),
// i.e. a wave instruction writes four rows x 128 bytes (whole lines)
// MODE 1: two adjacent column tiles interleaved (lane lr: cells 2 lr, 2 lr + 1 of a 32-cell piece): b128 stores, four rows x 256 bytes;
// the wave then owns 32 cells and the chunk is 256 wide
template <int BR, int MODE, int POLICY>
__global__ __launch_bounds__(512) void mfma_layout_k(unsigned char *out, int bands)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / bands, b = lb % bands;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    unsigned char *base = out + (size_t)p * NR * PITCH;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, NR * PITCH, WORD3);
    constexpr int CW = MODE == 1 ? 256 : 128;
    for (int t = 0; t < (NC + CW - 1) / CW; t++) {
#pragma unroll
        for (int u = 0; u < BR / 16; u++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = b * BR + 16 * u + 4 * lk + r;
                const unsigned v = (unsigned)(row * 977 + lane);
                if (MODE == 0) {
                    const int col = CW * t + 16 * wave + lr;
                    if (row < NR && col < NC) __builtin_amdgcn_raw_buffer_store_b64((u32x2){v, v + 1}, rs, 8 * col + row * PITCH, 0, POLICY);
                } else {
                    const int col = CW * t + 32 * wave + 2 * lr;
                    if (row < NR && col + 1 < NC) __builtin_amdgcn_raw_buffer_store_b128((u32x4){v, v + 1, v + 2, v + 3}, rs, 8 * col + row * PITCH, 0, POLICY);
                }
            }
        }
    }
}

template <typename F>
static float timed(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 4096;
    unsigned char *buf;
    const size_t bytes = (size_t)K * NR * PITCH;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0x11, bytes);
    const double B = (double)K * NR * NC * 8.0;
    float t;
    t = timed([&] { hipMemsetAsync(buf, 0x22, bytes, 0); });
    printf("%-64s %.3f ms  %.2f TB/s (of the %d-byte pitch: all bytes)\n", "hipMemset of the whole buffer", t, (double)bytes / t / 1e9, PITCH);
#define RUN(name, launch) t = timed([&] { launch; }); printf("%-64s %.3f ms  %.2f TB/s\n", name, t, B / t / 1e9);
    RUN("strips 128, down, lanes interleaved, default [today]", (strip_k<0, 0><<<K * 8, 512>>>(buf, 8)))
    RUN("strips 128, down, lanes interleaved, nt", (strip_k<0, 2><<<K * 8, 512>>>(buf, 8)))
    RUN("strips 128, down, lanes linear, default", (strip_k<1, 0><<<K * 8, 512>>>(buf, 8)))
    RUN("strips 128, down, lanes linear, nt", (strip_k<1, 2><<<K * 8, 512>>>(buf, 8)))
    RUN("bands 32 rows, right, b128 (1 KB / instr), default", (band_k<32, 0, 0><<<K * 32, 512>>>(buf, 32)))
    RUN("bands 32 rows, right, b128 (1 KB / instr), nt", (band_k<32, 0, 2><<<K * 32, 512>>>(buf, 32)))
    RUN("bands 64 rows, right, b128, default", (band_k<64, 0, 0><<<K * 16, 512>>>(buf, 16)))
    RUN("bands 64 rows, right, b128, nt", (band_k<64, 0, 2><<<K * 16, 512>>>(buf, 16)))
    RUN("bands 16 rows, right, b128, default", (band_k<16, 0, 0><<<K * 63, 512>>>(buf, 63)))
    RUN("bands 32 rows, right, 2 x b64 (2 x 512 B), default", (band_k<32, 1, 0><<<K * 32, 512>>>(buf, 32)))
    RUN("bands 32 rows, right, 2 x b64, nt", (band_k<32, 1, 2><<<K * 32, 512>>>(buf, 32)))
    RUN("bands 32 rows, right, chunks of 64: b64 (512 B / instr), default", (band_k<32, 2, 0><<<K * 32, 512>>>(buf, 32)))
    RUN("bands 8 rows (1 row per wave), right, b128, default", (band_k<8, 0, 0><<<K * 125, 512>>>(buf, 125)))
    RUN("matrix-core layout, bands 64, 4 rows x 128 B per instr, default", (mfma_layout_k<64, 0, 0><<<K * 16, 512>>>(buf, 16)))
    RUN("matrix-core layout, bands 64, 4 rows x 128 B per instr, nt", (mfma_layout_k<64, 0, 2><<<K * 16, 512>>>(buf, 16)))
    RUN("matrix-core layout, bands 32, 4 rows x 128 B per instr, default", (mfma_layout_k<32, 0, 0><<<K * 32, 512>>>(buf, 32)))
    RUN("matrix-core layout, bands 32, 4 rows x 128 B per instr, nt", (mfma_layout_k<32, 0, 2><<<K * 32, 512>>>(buf, 32)))
    RUN("matrix-core layout, bands 64, 4 rows x 256 B per instr (b128), default", (mfma_layout_k<64, 1, 0><<<K * 16, 512>>>(buf, 16)))
    RUN("matrix-core layout, bands 64, 4 rows x 256 B per instr (b128), nt", (mfma_layout_k<64, 1, 2><<<K * 16, 512>>>(buf, 16)))
    RUN("matrix-core layout, bands 32, 4 rows x 256 B per instr (b128), nt", (mfma_layout_k<32, 1, 2><<<K * 32, 512>>>(buf, 32)))
    RUN("matrix-core layout, bands 16, 4 rows x 256 B per instr (b128), nt", (mfma_layout_k<16, 1, 2><<<K * 63, 512>>>(buf, 63)))
    return 0;
}
