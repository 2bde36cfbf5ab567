// Do scalar stores work on gfx950, and are they visible to a later kernel's vector loads?
// hipcc -O3 --offload-arch=gfx950 sstore.hip -o sstore
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void writer(unsigned long long *out, int n)
{
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    // 16 wave-uniform 64-bit words derived from ballots (as the band kernel's emission)
    unsigned long long *dst = out + (size_t)wave * 16;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const unsigned long long w = __ballot(((lane * 2654435761u + e * 40503u + wave) >> 7) & 1);
        asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(w), "s"(dst), "n"(8 * e) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

__global__ void reader(const unsigned long long *in, unsigned long long *sum, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * 16) atomicAdd(sum, in[i]);
}

int main()
{
    const int n = 1 << 16;
    unsigned long long *d, *dsum;
    hipMalloc(&d, (size_t)n * 16 * 8);
    hipMalloc(&dsum, 8);
    hipMemset(d, 0xff, (size_t)n * 16 * 8);
    hipMemset(dsum, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(writer, dim3(n / 4), dim3(256), 0, 0, d, n);
    hipEventRecord(e1);
    hipLaunchKernelGGL(reader, dim3(n * 16 / 256), dim3(256), 0, 0, d, dsum, n);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> host((size_t)n * 16);
    hipMemcpy(host.data(), d, host.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long dev_sum = 0;
    hipMemcpy(&dev_sum, dsum, 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    unsigned long long ref_sum = 0;
    for (int w = 0; w < n; w++)
        for (int e = 0; e < 16; e++) {
            unsigned long long ref = 0;
            for (unsigned lane = 0; lane < 64; lane++)
                if (((lane * 2654435761u + e * 40503u + (unsigned)w) >> 7) & 1) ref |= 1ull << lane;
            ref_sum += ref;
            bad += host[(size_t)w * 16 + e] != ref;
        }
    printf("sync: %s; scalar stores: %zu of %zu words wrong; reader kernel sum %s; writer %.3f ms for %d waves\n",
           hipGetErrorString(err), bad, host.size(), dev_sum == ref_sum ? "ok" : "WRONG", ms, n);
    return bad != 0;
}
