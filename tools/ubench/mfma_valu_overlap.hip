// Micro-benchmark (dev tool, round 4): does v_mfma_f32_16x16x4_f32 overlap with vector instructions on gfx950?
//   (a) inside ONE wave's stream: K independent fillers after every MFMA, one wave per SIMD (cycles per MFMA gap);
//   (b) between the waves of a SIMD: waves that only issue MFMAs beside waves that only issue fillers;
//   (c) the strip kernel's shape: 12 MFMAs + 228 fillers per trip at 6 waves per SIMD, phased (12 MFMAs, then the
//       fillers) against interleaved (19 fillers after every MFMA).
// Fillers: v_add_u32 (integer), v_add_f32, v_pk_add_f32, v_pk_max_u16 -- the kinds crp_rows32_kernel issues.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap && tools/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define MFMA(acc) "v_mfma_f32_16x16x4_f32 %[" #acc "], %[a], %[b], %[" #acc "]\n"
#define F_U32(r) "v_add_u32 %[" #r "], %[" #r "], %[one]\n"
#define F_F32(r) "v_add_f32 %[" #r "], %[" #r "], %[onef]\n"
#define F_PKF(r) "v_pk_add_f32 %[" #r "], %[" #r "], %[p1]\n"
#define F_PKU(r) "v_pk_max_u16 %[" #r "], %[" #r "], %[one]\n"

#define OPS                                                                                                              \
    : [c0] "+v"(c0), [c1] "+v"(c1), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4), [f5] "+v"(f5), \
      [f6] "+v"(f6), [f7] "+v"(f7), [q0] "+v"(q0), [q1] "+v"(q1), [q2] "+v"(q2), [q3] "+v"(q3)                              \
    : [a] "v"(a), [b] "v"(b), [one] "v"(one), [onef] "v"(onef), [p1] "v"(p1)

// filler number i of kind T (registers cycle: eight independent chains; four for the packed float form)
#define FILL(T, i) FILL_##T(i)
#define FILL_0(i) F_U32(f##i)
#define FILL_1(i) F_F32(f##i)
#define FILL_3(i) F_PKU(f##i)
#define FILL_2(i) FILL2_##i
#define FILL2_0 F_PKF(q0)
#define FILL2_1 F_PKF(q1)
#define FILL2_2 F_PKF(q2)
#define FILL2_3 F_PKF(q3)
#define FILL2_4 F_PKF(q0)
#define FILL2_5 F_PKF(q1)
#define FILL2_6 F_PKF(q2)
#define FILL2_7 F_PKF(q3)

#define FK0(T)
#define FK1(T) FILL(T, 0)
#define FK2(T) FK1(T) FILL(T, 1)
#define FK3(T) FK2(T) FILL(T, 2)
#define FK4(T) FK3(T) FILL(T, 3)
#define FK5(T) FK4(T) FILL(T, 4)
#define FK6(T) FK5(T) FILL(T, 5)
#define FK7(T) FK6(T) FILL(T, 6)
#define FK8(T) FK7(T) FILL(T, 7)
#define FK12(T) FK8(T) FK4(T)
#define FK16(T) FK8(T) FK8(T)
#define FK19(T) FK16(T) FK3(T)

#define STATE                                                                                                  \
    v4f c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};                                                  \
    float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)], onef = in[1];                               \
    unsigned one = (unsigned)in[2] | 1u;                                                                         \
    unsigned f0 = 1, f1 = 2, f2 = 3, f3 = 4, f4 = 5, f5 = 6, f6 = 7, f7 = 8;                                     \
    v2f q0 = {1.f, 2.f}, q1 = {3.f, 4.f}, q2 = {5.f, 6.f}, q3 = {7.f, 8.f}, p1 = {in[3], in[4]};

#define SINK                                                                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + (float)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + q0[0] + q1[1] + q2[0] + q3[1];

// (a) one stream: per trip two MFMAs (alternating accumulators: a dependent one is 64 cycles away), each followed by K fillers
#define DEF_A(T, K)                                                                                             \
    __global__ __launch_bounds__(256) void ka_##T##_##K(const float *in, float *out, long long *cyc, int trips) \
    {                                                                                                           \
        STATE                                                                                                   \
        const long long t0 = __builtin_readcyclecounter();                                                      \
        for (int it = 0; it < trips; it++) {                                                                    \
            asm volatile(MFMA(c0) FK##K(T) MFMA(c1) FK##K(T) MFMA(c0) FK##K(T) MFMA(c1) FK##K(T) OPS);          \
        }                                                                                                       \
        const long long t1 = __builtin_readcyclecounter();                                                      \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;        \
        SINK                                                                                                    \
    }
#define DEF_A_ALL(T) DEF_A(T, 0) DEF_A(T, 1) DEF_A(T, 2) DEF_A(T, 3) DEF_A(T, 4) DEF_A(T, 5) DEF_A(T, 6) DEF_A(T, 8) DEF_A(T, 12) DEF_A(T, 19)
DEF_A_ALL(0)
DEF_A_ALL(1)
DEF_A_ALL(2)
DEF_A_ALL(3)

// fillers alone (no MFMA): 4 x K per trip
#define DEF_F(T)                                                                                                \
    __global__ __launch_bounds__(256) void kf_##T(const float *in, float *out, long long *cyc, int trips)      \
    {                                                                                                           \
        STATE                                                                                                   \
        const long long t0 = __builtin_readcyclecounter();                                                      \
        for (int it = 0; it < trips; it++) {                                                                    \
            asm volatile(FK8(T) FK8(T) FK8(T) FK8(T) OPS);                                                      \
        }                                                                                                       \
        const long long t1 = __builtin_readcyclecounter();                                                      \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;        \
        SINK                                                                                                    \
    }
DEF_F(0)
DEF_F(1)
DEF_F(2)
DEF_F(3)

// (b) roles: in a block of 64 * 4 * W threads, waves with (wave >> 2) < n_mfma_rows issue only MFMAs (4 per trip), the others
// only fillers (32 per trip = the same 128 issue cycles if a filler costs 4); `which` = 1: MFMA waves only run, 2: filler
// waves only, 3: both
template <int T>
__global__ __launch_bounds__(1024) void kb(const float *in, float *out, long long *cyc, int trips, int n_mfma_rows, int which, unsigned *hwid)
{
    STATE
    const int wave = threadIdx.x >> 6;
    const bool mf = (wave >> 2) < n_mfma_rows;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) hwid[wave] = id;
    const long long t0 = __builtin_readcyclecounter();
    if (mf) {
        if (which & 1)
            for (int it = 0; it < trips; it++) asm volatile(MFMA(c0) MFMA(c1) MFMA(c0) MFMA(c1) OPS);
    } else if (which & 2) {
        for (int it = 0; it < trips; it++) {
            if (T == 0) asm volatile(FK8(0) FK8(0) FK8(0) FK8(0) OPS);
            if (T == 1) asm volatile(FK8(1) FK8(1) FK8(1) FK8(1) OPS);
            if (T == 2) asm volatile(FK8(2) FK8(2) FK8(2) FK8(2) OPS);
            if (T == 3) asm volatile(FK8(3) FK8(3) FK8(3) FK8(3) OPS);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    SINK
}

// (c) the strip kernel's shape per trip: 12 MFMAs and 12 x 19 fillers; PH = 1 phased, 0 interleaved
template <int T, int PH>
__global__ __launch_bounds__(512) void kc(const float *in, float *out, long long *cyc, int trips)
{
    STATE
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < trips; it++) {
        if (PH) {
            asm volatile(MFMA(c0) MFMA(c1) MFMA(c0) MFMA(c1) MFMA(c0) MFMA(c1) MFMA(c0) MFMA(c1) MFMA(c0) MFMA(c1) MFMA(c0) MFMA(c1) OPS);
            for (int r = 0; r < 3; r++) {
                if (T == 0) asm volatile(FK19(0) FK19(0) FK19(0) FK19(0) OPS);
                if (T == 1) asm volatile(FK19(1) FK19(1) FK19(1) FK19(1) OPS);
                if (T == 3) asm volatile(FK19(3) FK19(3) FK19(3) FK19(3) OPS);
            }
        } else {
            for (int r = 0; r < 3; r++) {
                if (T == 0) asm volatile(MFMA(c0) FK19(0) MFMA(c1) FK19(0) MFMA(c0) FK19(0) MFMA(c1) FK19(0) OPS);
                if (T == 1) asm volatile(MFMA(c0) FK19(1) MFMA(c1) FK19(1) MFMA(c0) FK19(1) MFMA(c1) FK19(1) OPS);
                if (T == 3) asm volatile(MFMA(c0) FK19(3) MFMA(c1) FK19(3) MFMA(c0) FK19(3) MFMA(c1) FK19(3) OPS);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    SINK
}

static float *d_in, *d_out;
static long long *d_cyc;
static unsigned *d_hw;
static hipEvent_t e0, e1;

template <typename F>
static void run(const char *name, F launch, int blocks, int threads, int trips, double per_trip_units, const char *unit)
{
    const int waves = blocks * (threads / 64);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long *h = (long long *)malloc(sizeof(long long) * waves);
    hipMemcpy(h, d_cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    int n = 0;
    for (int i = 0; i < waves; i++) if (h[i] > 0) { sum += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; n++; }
    free(h);
    // s_memtime ticks at 100 MHz on gfx950?  report both ticks and wall
    printf("%-46s %8.3f ms   ticks/trip avg %9.2f max %9.2f   wall ns per %s %.3f\n", name, ms, n ? sum / n / trips : 0.0, mx / trips,
           unit, ms * 1e6 / (trips * per_trip_units));
    hipMemset(d_cyc, 0, sizeof(long long) * 65536);
}

int main()
{
    hipMalloc(&d_in, 4096);
    hipMalloc(&d_out, 256 * 16 * 1024 * 4);
    hipMalloc(&d_cyc, sizeof(long long) * 65536);
    hipMalloc(&d_hw, 4096);
    float hin[128];
    for (int i = 0; i < 128; i++) hin[i] = 0.37f + 0.01f * (float)(i % 17);
    hin[1] = 1.0f; hin[2] = 1.0f; hin[3] = 0.5f; hin[4] = 0.25f;
    hipMemcpy(d_in, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipMemset(d_cyc, 0, sizeof(long long) * 65536);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int trips = 20000;
    const char *kinds[4] = {"v_add_u32", "v_add_f32", "v_pk_add_f32", "v_pk_max_u16"};
    printf("== (a) one wave per SIMD, one stream: 4 MFMAs per trip, K fillers after each (gap = wall ns per MFMA x clock)\n");
#define RUN_A(T, K) run((sprintf(nm, "%s K=%d", kinds[T], K), nm), [&] { hipLaunchKernelGGL(ka_##T##_##K, dim3(256), dim3(256), 0, 0, d_in, d_out, d_cyc, trips); }, 256, 256, trips, 4.0, "MFMA")
#define RUN_A_ALL(T) RUN_A(T, 0); RUN_A(T, 1); RUN_A(T, 2); RUN_A(T, 3); RUN_A(T, 4); RUN_A(T, 5); RUN_A(T, 6); RUN_A(T, 8); RUN_A(T, 12); RUN_A(T, 19)
    char nm[128];
    RUN_A_ALL(0);
    RUN_A_ALL(1);
    RUN_A_ALL(2);
    RUN_A_ALL(3);
    printf("== fillers alone, one wave per SIMD: 32 per trip\n");
    run("v_add_u32 x32", [&] { hipLaunchKernelGGL(kf_0, dim3(256), dim3(256), 0, 0, d_in, d_out, d_cyc, trips); }, 256, 256, trips, 32.0, "filler");
    run("v_add_f32 x32", [&] { hipLaunchKernelGGL(kf_1, dim3(256), dim3(256), 0, 0, d_in, d_out, d_cyc, trips); }, 256, 256, trips, 32.0, "filler");
    run("v_pk_add_f32 x32", [&] { hipLaunchKernelGGL(kf_2, dim3(256), dim3(256), 0, 0, d_in, d_out, d_cyc, trips); }, 256, 256, trips, 32.0, "filler");
    run("v_pk_max_u16 x32", [&] { hipLaunchKernelGGL(kf_3, dim3(256), dim3(256), 0, 0, d_in, d_out, d_cyc, trips); }, 256, 256, trips, 32.0, "filler");
    printf("== (b) roles on one SIMD (1024-thread blocks = 4 waves per SIMD, one block per CU): rows of MFMA-only waves beside rows of filler-only waves\n");
    for (int T = 0; T < 4; T++) {
        for (int nm_rows = 1; nm_rows <= 2; nm_rows++) {
            for (int which = 1; which <= 3; which++) {
                sprintf(nm, "%s: %d MFMA rows / %d filler rows, run %s", kinds[T], nm_rows, 4 - nm_rows, which == 1 ? "MFMA only" : which == 2 ? "fillers only" : "both");
                auto l = [&] {
                    if (T == 0) hipLaunchKernelGGL(kb<0>, dim3(256), dim3(1024), 0, 0, d_in, d_out, d_cyc, trips, nm_rows, which, d_hw);
                    if (T == 1) hipLaunchKernelGGL(kb<1>, dim3(256), dim3(1024), 0, 0, d_in, d_out, d_cyc, trips, nm_rows, which, d_hw);
                    if (T == 2) hipLaunchKernelGGL(kb<2>, dim3(256), dim3(1024), 0, 0, d_in, d_out, d_cyc, trips, nm_rows, which, d_hw);
                    if (T == 3) hipLaunchKernelGGL(kb<3>, dim3(256), dim3(1024), 0, 0, d_in, d_out, d_cyc, trips, nm_rows, which, d_hw);
                };
                run(nm, l, 256, 1024, trips, 1.0, "trip");
            }
        }
    }
    unsigned hw[16];
    hipMemcpy(hw, d_hw, sizeof(hw), hipMemcpyDeviceToHost);
    printf("   HW_ID of block 0's waves (simd = bits 5:4):");
    for (int i = 0; i < 16; i++) printf(" w%d:simd%u", i, (hw[i] >> 4) & 3);
    printf("\n");
    printf("== (c) the strip kernel's shape: 12 MFMAs + 228 fillers per trip, 512-thread blocks, 3 blocks per CU = 6 waves per SIMD\n");
    const int tc = 4000;
    run("v_add_u32 phased", [&] { hipLaunchKernelGGL((kc<0, 1>), dim3(768), dim3(512), 0, 0, d_in, d_out, d_cyc, tc); }, 768, 512, tc, 1.0, "trip");
    run("v_add_u32 interleaved", [&] { hipLaunchKernelGGL((kc<0, 0>), dim3(768), dim3(512), 0, 0, d_in, d_out, d_cyc, tc); }, 768, 512, tc, 1.0, "trip");
    run("v_add_f32 phased", [&] { hipLaunchKernelGGL((kc<1, 1>), dim3(768), dim3(512), 0, 0, d_in, d_out, d_cyc, tc); }, 768, 512, tc, 1.0, "trip");
    run("v_add_f32 interleaved", [&] { hipLaunchKernelGGL((kc<1, 0>), dim3(768), dim3(512), 0, 0, d_in, d_out, d_cyc, tc); }, 768, 512, tc, 1.0, "trip");
    run("v_pk_max_u16 phased", [&] { hipLaunchKernelGGL((kc<3, 1>), dim3(768), dim3(512), 0, 0, d_in, d_out, d_cyc, tc); }, 768, 512, tc, 1.0, "trip");
    run("v_pk_max_u16 interleaved", [&] { hipLaunchKernelGGL((kc<3, 0>), dim3(768), dim3(512), 0, 0, d_in, d_out, d_cyc, tc); }, 768, 512, tc, 1.0, "trip");
    printf("   (one wave per SIMD, same two kernels, 256 blocks of 256 threads)\n");
    run("v_add_u32 phased, 1 wave/SIMD", [&] { hipLaunchKernelGGL((kc<0, 1>), dim3(256), dim3(256), 0, 0, d_in, d_out, d_cyc, tc); }, 256, 256, tc, 1.0, "trip");
    run("v_add_u32 interleaved, 1 wave/SIMD", [&] { hipLaunchKernelGGL((kc<0, 0>), dim3(256), dim3(256), 0, 0, d_in, d_out, d_cyc, tc); }, 256, 256, tc, 1.0, "trip");
    return 0;
}
