// Micro-benchmark (dev tool): issue cost of counting patterns on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 8192
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned *in, unsigned *out)
{
    unsigned hi[16];
    for (int e = 0; e < 16; e++) hi[e] = in[threadIdx.x * 16 + e];
    unsigned cand = in[0];
    int c = 0;
    for (int it = 0; it < N_IT; it++) {
        cand = cand * 1664525u + 1013904223u;
        unsigned cd = cand >> 1;
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) c += (hi[e] < cd) ? 1 : 0;
        } else if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 16; e++) c += (int)((hi[e] - cd) >> 31);
        } else if (MODE == 2) {
#pragma unroll
            for (int e = 0; e < 16; e++) c += __popcll(__ballot(hi[e] < cd));
        } else if (MODE == 3) {   // f64 adds for reference
            double a = __hiloint2double(hi[0], hi[1]);
#pragma unroll
            for (int e = 0; e < 16; e++) a += (double)cd;
            c += __double2hiint(a);
        } else if (MODE == 4) {   // 32-bit adds for reference
#pragma unroll
            for (int e = 0; e < 16; e++) c += hi[e] ^ cd;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c;
}
int main()
{
    unsigned *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 24);
    hipMemset(in, 0x3c, 1 << 20);
    const int blocks = 256 * 8;   // 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"cmp+addc", "sub+shr+add", "ballot+bcnt", "f64 add", "xor+add"};
    for (int m = 0; m < 5; m++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (m == 0) k<0><<<blocks, 256>>>(in, out); else if (m == 1) k<1><<<blocks, 256>>>(in, out);
            else if (m == 2) k<2><<<blocks, 256>>>(in, out); else if (m == 3) k<3><<<blocks, 256>>>(in, out); else k<4><<<blocks, 256>>>(in, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) {
                // per SIMD: 8 waves x N_IT x 16 element-ops
                double cyc = ms * 1e-3 * 2.4e9 / (8.0 * N_IT * 16);
                printf("%-12s %.3f ms  -> %.2f cycles per element-op per wave-slot (at 2.4 GHz)\n", names[m], ms, cyc);
            }
        }
    }
    return 0;
}
