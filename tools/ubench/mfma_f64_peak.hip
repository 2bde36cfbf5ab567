// Micro-benchmark (dev tool, round 5): what v_mfma_f64_16x16x4_f64 sustains on a whole MI355X with nothing beside it -- W waves per SIMD,
// each a chain of MFMAs over A independent accumulators, operands in registers (random normal values, or zeros), no memory traffic
// inside the loop.  Answers whether EarlySNF's product kernel (0.69 of the 78.6 TFLOP/s f64 matrix peak, the matrix cores busy 0.70 of
// the cycles with or without its loads, staging and barriers) is held by the chip or by the kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f64_peak.hip -o tools/ubench/mfma_f64_peak && tools/ubench/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef double v4d __attribute__((ext_vector_type(4)));

// LDS > 0: before every 8 MFMAs the wave reads its next operands from shared memory (4 x ds_read2_b64, as the product kernel does: the
// operands of the MFMAs then change every step); LDS == 2: those reads are issued one step ahead (two register sets)
template <int LDS>
__global__ __launch_bounds__(1024) void k_mfma_lds(const double *in, double *out, int trips)
{
    __shared__ double sm[2][128][36];
    for (int e = threadIdx.x; e < 2 * 128 * 36; e += blockDim.x) (&sm[0][0][0])[e] = in[e & 255];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wi = (wave >> 2) * 32, wj = (wave & 3) * 32;
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a[2][2][2], b[2][2][2];
    auto operands = [&](const int set, const int kk) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                a[set][h][t] = sm[0][wi + 16 * t + lr][kk + 4 * h + lk];
                b[set][h][t] = sm[1][wj + 16 * t + lr][kk + 4 * h + lk];
            }
    };
    operands(0, 0);
    for (int it = 0; it < trips; it++) {
#pragma unroll
        for (int st = 0; st < 4; st++) {
            if (LDS == 2) operands((st + 1) & 1, 8 * ((st + 1) & 3));
            if (LDS == 1) operands(st & 1, 8 * st);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int ta = 0; ta < 2; ta++)
#pragma unroll
                    for (int tb = 0; tb < 2; tb++)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[LDS ? (st & 1) : 0][h][ta], b[LDS ? (st & 1) : 0][h][tb], acc[ta][tb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int LDS>
static void run_lds(const char *what, const double *d_in, double *d_out, int trips)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma_lds<LDS>, dim3(256), dim3(1024), 0, 0, d_in, d_out, trips / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma_lds<LDS>, dim3(256), dim3(1024), 0, 0, d_in, d_out, trips);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 16.0 * trips * 32.0 * 2048.0;
    printf("%-8s 16 waves / CU of 32 x 32 wave tiles, operands %s: %8.3f ms  %6.1f TFLOP/s  (%.3f of 78.6)\n", what,
           LDS == 0 ? "fixed registers" : (LDS == 1 ? "from LDS before each pair of steps" : "from LDS one pair of steps ahead"), ms, flop / ms / 1e9, flop / ms / 1e9 / 78.6);
}

template <int A>
__global__ __launch_bounds__(256) void k_mfma(const double *in, double *out, int trips)
{
    double a[2], b[2];
    a[0] = in[threadIdx.x & 63]; a[1] = in[64 + (threadIdx.x & 63)];
    b[0] = in[128 + (threadIdx.x & 63)]; b[1] = in[192 + (threadIdx.x & 63)];
    v4d acc[A];
#pragma unroll
    for (int i = 0; i < A; i++) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < trips; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < A; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(u + i) & 1], b[(u >> 1) & 1], acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < A; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int A>
static void run(const char *what, const double *d_in, double *d_out, int blocks_per_cu, int trips)
{
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<A>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, trips / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<A>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, trips);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4.0 * trips * 8.0 * A * 2048.0;
    printf("%-8s %d waves/SIMD, %d accumulators: %8.3f ms  %6.1f TFLOP/s  (%.3f of 78.6)\n", what, blocks_per_cu, A, ms, flop / ms / 1e9, flop / ms / 1e9 / 78.6);
}

int main()
{
    double h[256], *d_in, *d_out;
    hipMalloc(&d_in, sizeof h);
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(double));
    for (int pass = 0; pass < 2; pass++) {
        srand(1);
        for (int i = 0; i < 256; i++) {
            const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
            h[i] = pass == 0 ? sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) * 1e-3 : 0.0;
        }
        hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
        const char *what = pass == 0 ? "random" : "zeros";
        run<4>(what, d_in, d_out, 1, 20000);
        run<4>(what, d_in, d_out, 2, 20000);
        run<4>(what, d_in, d_out, 4, 10000);
        run<8>(what, d_in, d_out, 4, 5000);
        run<2>(what, d_in, d_out, 8, 10000);
        run_lds<0>(what, d_in, d_out, 4000);
        run_lds<1>(what, d_in, d_out, 4000);
        run_lds<2>(what, d_in, d_out, 4000);
    }
    return 0;
}
