// LDS atomic throughput on gfx950: cycles per ds_add_u32 wave-instruction for the address patterns of the histogram
// selection (csrc/planar_select.h, band_kernels.hip).  Build: hipcc -O3 --offload-arch=gfx950 lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int N_OPS = 256;

// pattern: per (op, lane) a word index into the wave's 1088-word histogram and an active flag
__global__ __launch_bounds__(1024) void probe(const unsigned short *__restrict__ idx, const unsigned long long *__restrict__ mask,
                                              int use_exec, unsigned long long *out)
{
    __shared__ unsigned hist[16 * 1088];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned *h = hist + wave * 1088;
    for (int i = lane; i < 1088; i += 64) h[i] = 0;
    __syncthreads();
    unsigned a[16];
    unsigned long long m[16];
    unsigned long long t0 = 0, t1 = 0;
    unsigned long long total = 0;
    for (int rep = 0; rep < N_OPS / 16; rep++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            a[e] = idx[(rep * 16 + e) * 64 + lane];
            m[e] = mask[rep * 16 + e];
        }
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (use_exec == 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) atomicAdd(&h[a[e]], 1u);
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                if ((m[e] >> lane) & 1) atomicAdd(&h[a[e]], 1u);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        total += t1 - t0;
    }
    if (lane == 0) out[blockIdx.x * 16 + wave] = total;
    if (h[lane] == 0xdeadbeef) out[0] = 1;
}

int main()
{
    const int waves_list[] = {1, 4, 16};
    const char *names[] = {"64 lanes consecutive (spill words)", "58 consecutive + 6 random bins", "6 random lanes active (exec-masked)",
                           "16 random lanes active (exec-masked)", "64 lanes random bins", "64 lanes same word", "no lanes active"};
    unsigned short *d_idx;
    unsigned long long *d_mask, *d_out;
    hipMalloc(&d_idx, N_OPS * 64 * sizeof(unsigned short));
    hipMalloc(&d_mask, N_OPS * sizeof(unsigned long long));
    hipMalloc(&d_out, 256 * 16 * sizeof(unsigned long long));
    srand(1);
    for (int pat = 0; pat < 7; pat++) {
        std::vector<unsigned short> idx(N_OPS * 64);
        std::vector<unsigned long long> mask(N_OPS);
        for (int op = 0; op < N_OPS; op++) {
            unsigned long long mk = 0;
            int nact = pat == 2 ? 6 : pat == 3 ? 16 : 0;
            while (__builtin_popcountll(mk) < nact) mk |= 1ull << (rand() % 64);
            if (pat == 1) { while (__builtin_popcountll(mk) < 6) mk |= 1ull << (rand() % 64); }
            for (int l = 0; l < 64; l++) {
                unsigned short v = 1024 + l;
                if (pat == 1 && ((mk >> l) & 1)) v = rand() % 1024;
                if (pat == 2 || pat == 3 || pat == 4) v = rand() % 1024;
                if (pat == 5) v = 7;
                idx[op * 64 + l] = v;
            }
            mask[op] = (pat == 2 || pat == 3) ? mk : (pat == 6 ? 0ull : ~0ull);
        }
        hipMemcpy(d_idx, idx.data(), idx.size() * sizeof(unsigned short), hipMemcpyHostToDevice);
        hipMemcpy(d_mask, mask.data(), mask.size() * sizeof(unsigned long long), hipMemcpyHostToDevice);
        for (int wi = 0; wi < 3; wi++) {
            const int waves = waves_list[wi];
            const int use_exec = (pat == 2 || pat == 3 || pat == 6) ? 1 : 0;
            hipLaunchKernelGGL(probe, dim3(256), dim3(64 * waves), 0, 0, d_idx, d_mask, use_exec, d_out);
            hipDeviceSynchronize();
            std::vector<unsigned long long> out(256 * 16);
            hipMemcpy(out.data(), d_out, out.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double sum = 0;
            for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) sum += (double)out[b * 16 + w];
            const double per_wave_op = sum / (256.0 * waves) / N_OPS;
            printf("%-40s waves/CU %2d: %7.1f cycles per ds_add per wave -> %6.1f per CU-instruction\n", names[pat], waves, per_wave_op,
                   per_wave_op / waves);
        }
    }
    return 0;
}
