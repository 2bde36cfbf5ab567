// Micro-benchmark (dev tool, round 3): store and load patterns of candidate layouts of the key matrix when it is split
// into a 16-bit HIGH plane and a 16-bit LOW plane.  Pure memory patterns, no arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/plane_layout.hip -o tools/ubench/plane_layout && tools/ubench/plane_layout [pairs]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int M = 992, ROWS = 32;
constexpr int WORD3 = 0x00020000;

__device__ inline int xcd_remap(int b, int n) { const int per = n / 8; return b < per * 8 ? (b & 7) * per + (b >> 3) : b; }

// ---- stores: one 512-thread block per (pair, strip), walking down in 32-row steps, wave w owns rows 4w .. 4w+3 of a step
// MODE 0: u32 cells, 448-byte pieces (TN = 112), pitch 4096 B, one 8-byte store per lane        [today]
// MODE 1: two u16 planes, row-major, pitch 2048 B each, 224-byte pieces (odd strips start 32 B off a 64-byte line)
// MODE 3: strip-blocked: [row][strip][HIGH 224 B | LOW 224 B], pitch 4096 B
// MODE 4: two u16 planes, TN = 96 (192-byte pieces, all 64-byte aligned)
// MODE 5: HIGH plane only (TN = 112)
// MODE 6: two u16 planes, pieces of 224 B as in MODE 1, but each wave stores whole 8-byte (4 cells) per lane: 28 lanes
template <int MODE, int POLICY>
__global__ __launch_bounds__(512) void store_k(unsigned char *out, int strips)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / strips, s = lb % strips;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int TN = MODE == 4 ? 96 : 112;
    const int j0 = s * TN;
    if (j0 >= M) return;
    unsigned char *base = out + (size_t)p * M * 4096;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, M * 4096, WORD3);
    const int np = min(TN, M - j0) / 2;          // column pairs of this strip
    for (int t = 0; t < M / ROWS; t++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = t * ROWS + 4 * wave + q;
            const unsigned v = (unsigned)(row * 977 + lane);
            if (MODE == 0) {
                if (lane < np) __builtin_amdgcn_raw_buffer_store_b64((u32x2){v, v + 1}, rs, 8 * lane, row * 4096 + 4 * j0, POLICY);
            } else if (MODE == 1) {
                if (lane < np) {
                    __builtin_amdgcn_raw_buffer_store_b32(v, rs, 4 * lane, row * 2048 + 2 * j0, POLICY);
                    __builtin_amdgcn_raw_buffer_store_b32(v + 1, rs, 4 * lane, M * 2048 + row * 2048 + 2 * j0, POLICY);
                }
            } else if (MODE == 3) {
                if (lane < np) {
                    __builtin_amdgcn_raw_buffer_store_b32(v, rs, 4 * lane, row * 4096 + 448 * s, POLICY);
                    __builtin_amdgcn_raw_buffer_store_b32(v + 1, rs, 4 * lane, row * 4096 + 448 * s + 224, POLICY);
                }
            } else if (MODE == 4) {
                if (lane < np) {
                    __builtin_amdgcn_raw_buffer_store_b32(v, rs, 4 * lane, row * 2048 + 2 * j0, POLICY);
                    __builtin_amdgcn_raw_buffer_store_b32(v + 1, rs, 4 * lane, M * 2048 + row * 2048 + 2 * j0, POLICY);
                }
            } else if (MODE == 5) {
                if (lane < np) __builtin_amdgcn_raw_buffer_store_b32(v, rs, 4 * lane, row * 2048 + 2 * j0, POLICY);
            } else if (MODE == 6) {
                // lanes 0..27: HIGH (8 bytes = 4 cells each), lanes 32..59: LOW
                const int l = lane & 31;
                if (l < np / 2) __builtin_amdgcn_raw_buffer_store_b64((u32x2){v, v + 1}, rs, 8 * l, (lane >> 5) * M * 2048 + row * 2048 + 2 * j0, POLICY);
            }
        }
    }
}


// ---- stores of a ROTATED strip kernel: one 512-thread block per (pair, band of 56 rows), walking RIGHT in 128-column chunks;
// wave w owns rows 7w .. 7w+6 of the band, a row piece = 128 cells per chunk
// MODE 0: u32 cells, 512-byte pieces aligned to the chunk              MODE 1: the same, piece start shifted by (q - 14) cells (rounded to even)
// MODE 2: two u16 planes, 256-byte pieces aligned                      MODE 3: two u16 planes, shifted by (q - 14) cells (rounded to even)
// MODE 4: HIGH plane only, aligned
template <int MODE, int POLICY>
__global__ __launch_bounds__(512) void band_store_k(unsigned char *out, int bands)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / bands, b = lb % bands;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *base = out + (size_t)p * M * 4096;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, M * 4096, WORD3);
    for (int t = 0; t < 8; t++) {
#pragma unroll
        for (int q = 0; q < 7; q++) {
            const int row = b * 56 + 7 * wave + q;
            if (row >= M) continue;
            const unsigned v = (unsigned)(row * 977 + lane);
            const int shift = (MODE == 1 || MODE == 3) ? ((q - 14) & ~1) : 0;
            const int col = 128 * t + shift + 2 * lane;
            if (col < 0 || col + 1 >= M) continue;
            if (MODE == 0 || MODE == 1) {
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){v, v + 1}, rs, 4 * col, row * 4096, POLICY);
            } else if (MODE == 2 || MODE == 3) {
                __builtin_amdgcn_raw_buffer_store_b32(v, rs, 2 * col, row * 2048, POLICY);
                __builtin_amdgcn_raw_buffer_store_b32(v + 1, rs, 2 * col, M * 2048 + row * 2048, POLICY);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(v, rs, 2 * col, row * 2048, POLICY);
            }
        }
    }
}

// ---- row reads: 256-thread blocks, wave = 16 rows
// MODE 0: u32 cells, 16 dword loads, lanes contiguous (256 B per instruction), pitch 4096      [today]
// MODE 1: u16 HIGH plane, pitch 2048: lane = 16 consecutive cells = two 16-byte loads at a 32-byte lane stride
// MODE 2: u16 HIGH plane, pitch 2048: 8 dword loads, lanes contiguous
// MODE 3: strip-blocked HIGH: lane l -> strip l / 7, group l % 7: two 16-byte loads
// MODE 4: as MODE 1 with default-policy loads (no nt)
template <int MODE>
__global__ __launch_bounds__(256) void rows_k(const unsigned char *in, unsigned *sink, int rows_blocks)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / rows_blocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = ((lb % rows_blocks) * 4 + wave) * 16;
    if (r0 >= M) return;
    const unsigned char *base = in + (size_t)p * M * 4096;
    unsigned acc = 0;
    for (int i = r0; i < min(r0 + 16, M); i++) {
        if (MODE == 0) {
            const unsigned *row = (const unsigned *)(base + (size_t)i * 4096);
#pragma unroll
            for (int e = 0; e < 16; e++) acc += __builtin_nontemporal_load(row + min(e * 64 + lane, M - 1));
        } else if (MODE == 1 || MODE == 4) {
            const u32x4 *row = (const u32x4 *)(base + (size_t)i * 2048);
            const int l = min(lane, M / 16 - 1);
            u32x4 a, b;
            if (MODE == 1) { a = __builtin_nontemporal_load(row + 2 * l); b = __builtin_nontemporal_load(row + 2 * l + 1); }
            else { a = row[2 * l]; b = row[2 * l + 1]; }
            acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        } else if (MODE == 2) {
            const unsigned *row = (const unsigned *)(base + (size_t)i * 2048);
#pragma unroll
            for (int e = 0; e < 8; e++) acc += __builtin_nontemporal_load(row + min(e * 64 + lane, M / 2 - 1));
        } else if (MODE == 3) {
            const int l = min(lane, M / 16 - 1);
            const u32x4 *q = (const u32x4 *)(base + (size_t)i * 4096 + (l / 7) * 448 + (l % 7) * 32);
            const u32x4 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
            acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// ---- column tile reads: 512-thread blocks, one block = CW bytes of every row of a pair (8 lanes x 8 bytes per row segment)
// MODE 0: u32 cells: 16 columns = 64-byte segments, pitch 4096    [today]
// MODE 1: u16 HIGH plane: 32 columns = 64-byte segments, pitch 2048
// MODE 2: u16 HIGH plane: 16 columns = 32-byte segments (4 lanes x 8 bytes), pitch 2048
template <int MODE>
__global__ __launch_bounds__(512) void cols_k(const unsigned char *in, unsigned *sink, int col_blocks)
{
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / col_blocks, cb = lb % col_blocks;
    const unsigned char *base = in + (size_t)p * M * 4096;
    unsigned acc = 0;
    if (MODE == 0 || MODE == 1) {
        const int pitch = MODE == 0 ? 4096 : 2048;
        const int c2 = threadIdx.x & 7, rr = threadIdx.x >> 3;
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const u32x2 v = *(const u32x2 *)(base + (size_t)min(s * 64 + rr, M - 1) * pitch + cb * 64 + c2 * 8);
            acc += v.x + v.y;
        }
    } else {
        const int c2 = threadIdx.x & 3, rr = threadIdx.x >> 2;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const u32x2 v = *(const u32x2 *)(base + (size_t)min(s * 128 + rr, M - 1) * 2048 + cb * 32 + c2 * 8);
            acc += v.x + v.y;
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}


// ---- column tile reads again, under the occupancy of the real column kernel (LDS sized for three blocks per CU) and with a
// TILED plane: tiles of 64 rows x 64 cells (8 KB), so that the 1024 row segments a block fetches lie in 16 tiles instead of
// 1024 different 2 KB rows (TLB reach, DRAM page locality)
// MODE 0: row-major u16 plane, pitch 2048 B    MODE 1: tiled plane
template <int MODE>
__global__ __launch_bounds__(512) void cols2_k(const unsigned char *in, unsigned *sink, int col_blocks)
{
    __shared__ unsigned pad[12800];          // 51 KB: three blocks per CU
    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lb / col_blocks, cb = lb % col_blocks;
    const unsigned char *base = in + (size_t)p * M * 4096;
    unsigned acc = 0;
    const int c2 = threadIdx.x & 7, rp = threadIdx.x >> 3;
#pragma unroll
    for (int s = 0; s < 8; s++) {
#pragma unroll
        for (int z = 0; z < 2; z++) {
            const int row = min(128 * s + 2 * rp + z, M - 1);
            size_t off;
            if (MODE == 0) off = (size_t)row * 2048 + cb * 64 + c2 * 8;
            else off = ((size_t)(row >> 6) * 16 + (cb >> 1)) * 8192 + (row & 63) * 128 + (cb & 1) * 64 + c2 * 8;
            const u32x2 v = *(const u32x2 *)(base + off);
            acc += v.x + v.y;
        }
    }
    pad[threadIdx.x] = acc;
    __syncthreads();
    acc = pad[(threadIdx.x * 7) & 511];
    if (acc == 0x12345u) sink[0] = acc;
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
    unsigned char *buf; unsigned *sink;
    const size_t bytes = (size_t)K * M * 4096;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0x11, bytes);
    const double cells = (double)K * M * M;
    float t;
#define ST(MODE, POL, strips, B, name) t = timed([&] { store_k<MODE, POL><<<K * strips, 512>>>(buf, strips); }); \
    printf("store %-58s %.3f ms  %.2f TB/s\n", name, t, cells * (B) / t / 1e9);
    ST(0, 2, 9, 4, "u32 cells, 448 B pieces, nt [today]")
    ST(0, 0, 9, 4, "u32 cells, 448 B pieces, default policy")
    ST(1, 2, 9, 4, "two u16 planes row-major, 224 B pieces (odd strips 32 B off), nt")
    ST(1, 0, 9, 4, "two u16 planes row-major, 224 B pieces, default policy")
    ST(3, 2, 9, 4, "strip-blocked [HIGH 224 | LOW 224], nt")
    ST(3, 0, 9, 4, "strip-blocked [HIGH 224 | LOW 224], default policy")
    ST(4, 2, 11, 4, "two u16 planes, TN = 96 (192 B pieces, aligned), nt")
    ST(4, 0, 11, 4, "two u16 planes, TN = 96, default policy")
    ST(5, 2, 9, 2, "HIGH plane only, 224 B pieces, nt")
    ST(5, 0, 9, 2, "HIGH plane only, 224 B pieces, default policy")
    ST(6, 2, 9, 4, "two u16 planes, 8-byte stores (28 lanes per plane), nt")
    ST(6, 0, 9, 4, "two u16 planes, 8-byte stores, default policy")
#define BS(MODE, POL, B, name) t = timed([&] { band_store_k<MODE, POL><<<K * 18, 512>>>(buf, 18); }); \
    printf("bstore %-57s %.3f ms  %.2f TB/s\n", name, t, cells * (B) / t / 1e9);
    BS(0, 2, 4, "rotated: u32 cells, 512 B pieces aligned, nt")
    BS(0, 0, 4, "rotated: u32 cells, 512 B pieces aligned, default")
    BS(1, 2, 4, "rotated: u32 cells, pieces shifted by q - 14 cells, nt")
    BS(1, 0, 4, "rotated: u32 cells, pieces shifted by q - 14 cells, default")
    BS(2, 2, 4, "rotated: two u16 planes, 256 B pieces aligned, nt")
    BS(2, 0, 4, "rotated: two u16 planes, 256 B pieces aligned, default")
    BS(3, 2, 4, "rotated: two u16 planes, shifted, nt")
    BS(3, 0, 4, "rotated: two u16 planes, shifted, default")
    BS(4, 2, 2, "rotated: HIGH plane only, aligned, nt")
    BS(4, 0, 2, "rotated: HIGH plane only, aligned, default")
    const int rb = (M + 63) / 64;
#define RD(MODE, B, name) t = timed([&] { rows_k<MODE><<<K * rb, 256>>>(buf, sink, rb); }); \
    printf("rows  %-58s %.3f ms  %.2f TB/s\n", name, t, cells * (B) / t / 1e9);
    RD(0, 4, "u32 cells, 16 dword loads, nt [today]")
    RD(1, 2, "u16 plane, lane = 16 consecutive cells (2 x 16 B), nt")
    RD(4, 2, "u16 plane, lane = 16 consecutive cells (2 x 16 B), default")
    RD(2, 2, "u16 plane, 8 dword loads lanes contiguous, nt")
    RD(3, 2, "strip-blocked HIGH (224 B of every 448), 2 x 16 B, nt")
#define CL(MODE, CB, B, name) t = timed([&] { cols_k<MODE><<<K * CB, 512>>>(buf, sink, CB); }); \
    printf("cols  %-58s %.3f ms  %.2f TB/s\n", name, t, cells * (B) / t / 1e9);
    CL(0, 62, 4, "u32 cells, 16 columns (64 B segments) [today]")
    CL(1, 31, 2, "u16 plane, 32 columns (64 B segments)")
    CL(2, 62, 2, "u16 plane, 16 columns (32 B segments)")
#define CL2(MODE, name) t = timed([&] { cols2_k<MODE><<<K * 31, 512>>>(buf, sink, 31); }); \
    printf("cols2 %-57s %.3f ms  %.2f TB/s\n", name, t, cells * 2 / t / 1e9);
    CL2(0, "u16 plane row-major, 32 columns, 3 blocks / CU")
    CL2(1, "u16 plane in 64 x 64 tiles, 32 columns, 3 blocks / CU")
    return 0;
}
