// Micro-benchmark (dev tool, round 4): issue rate of the vector instructions the selection / strip / alignment kernels
// are made of, on gfx950, at 1, 2 and 8 waves per SIMD.  Each kernel repeats ONE instruction (eight independent
// register chains) 64 times per trip; the table gives cycles per wave-instruction per SIMD (s_memtime ticks of the
// slowest wave / (trips x 64 x waves per SIMD)).
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/valu_rate.hip -o tools/ubench/valu_rate && tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define VADD(N) "v_add_u32 %[r" #N "], %[r" #N "], %[a]\n"
#define SADD(N) "s_add_u32 s2" #N ", s2" #N ", %[s]\n"

#define OPS32 \
    : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [r4] "+v"(r4), [r5] "+v"(r5), [r6] "+v"(r6), [r7] "+v"(r7) \
    : [a] "v"(a), [b] "v"(b), [s] "s"(sv) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"
#define OPS64 \
    : [r0] "+v"(q0), [r1] "+v"(q1), [r2] "+v"(q2), [r3] "+v"(q3), [r4] "+v"(q4), [r5] "+v"(q5), [r6] "+v"(q6), [r7] "+v"(q7) \
    : [a] "v"(qa), [b] "v"(qb), [s] "s"(sv) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"

// NAME, instruction template with %[rN] as destination/accumulator N, 32-bit registers
#define DEF32(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(const unsigned *in, unsigned *out, long long *cyc, int trips)      \
    {                                                                                                                  \
        unsigned r0 = in[threadIdx.x], r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        unsigned a = in[64 + (threadIdx.x & 63)] | 1u, b = in[128 + (threadIdx.x & 63)] & 15u;                         \
        unsigned sv = __builtin_amdgcn_readfirstlane(in[3]);                                                           \
        const long long t0 = __builtin_readcyclecounter();                                                             \
        for (int it = 0; it < trips; it++) {                                                                           \
            asm volatile(I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7  \
                         I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 OPS32); \
        }                                                                                                              \
        const long long t1 = __builtin_readcyclecounter();                                                             \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                    \
    }
#define DEF64(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(const unsigned *in, unsigned *out, long long *cyc, int trips)      \
    {                                                                                                                  \
        v2f q0 = {1.f, 2.f}, q1 = {3.f, 4.f}, q2 = {5.f, 6.f}, q3 = {7.f, 8.f}, q4 = q0, q5 = q1, q6 = q2, q7 = q3;     \
        v2f qa = {(float)in[1], (float)in[2]}, qb = {(float)in[3], 0.5f};                                               \
        unsigned sv = __builtin_amdgcn_readfirstlane(in[3]);                                                           \
        const long long t0 = __builtin_readcyclecounter();                                                             \
        for (int it = 0; it < trips; it++) {                                                                           \
            asm volatile(I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7  \
                         I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 OPS64); \
        }                                                                                                              \
        const long long t1 = __builtin_readcyclecounter();                                                             \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                \
        out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(q0[0] + q1[1] + q2[0] + q3[1] + q4[0] + q5[1] + q6[0] + q7[1]); \
    }

// one instruction applied to the eight chains
#define ALL8(NAME, PRE, POST) DEF32(NAME, PRE "%[r0]" POST, PRE "%[r1]" POST, PRE "%[r2]" POST, PRE "%[r3]" POST, PRE "%[r4]" POST, PRE "%[r5]" POST, PRE "%[r6]" POST, PRE "%[r7]" POST)
#define I(OP, N, TAIL) OP " %[r" #N "], %[r" #N "], " TAIL "\n"
#define D32(NAME, OP, TAIL) DEF32(NAME, I(OP, 0, TAIL), I(OP, 1, TAIL), I(OP, 2, TAIL), I(OP, 3, TAIL), I(OP, 4, TAIL), I(OP, 5, TAIL), I(OP, 6, TAIL), I(OP, 7, TAIL))
#define D64(NAME, OP, TAIL) DEF64(NAME, I(OP, 0, TAIL), I(OP, 1, TAIL), I(OP, 2, TAIL), I(OP, 3, TAIL), I(OP, 4, TAIL), I(OP, 5, TAIL), I(OP, 6, TAIL), I(OP, 7, TAIL))
// dst, a, dst forms (shift-reversed operands)
#define IR(OP, N, HEAD) OP " %[r" #N "], " HEAD ", %[r" #N "]\n"
#define D32R(NAME, OP, HEAD) DEF32(NAME, IR(OP, 0, HEAD), IR(OP, 1, HEAD), IR(OP, 2, HEAD), IR(OP, 3, HEAD), IR(OP, 4, HEAD), IR(OP, 5, HEAD), IR(OP, 6, HEAD), IR(OP, 7, HEAD))
// unary: dst, dst
#define IU(OP, N, TAIL) OP " %[r" #N "], %[r" #N "]" TAIL "\n"
#define D32U(NAME, OP, TAIL) DEF32(NAME, IU(OP, 0, TAIL), IU(OP, 1, TAIL), IU(OP, 2, TAIL), IU(OP, 3, TAIL), IU(OP, 4, TAIL), IU(OP, 5, TAIL), IU(OP, 6, TAIL), IU(OP, 7, TAIL))

D32(add_u32, "v_add_u32", "%[a]")
D32(sub_u32_clamp, "v_sub_u32", "%[a] clamp")
D32(max_u32, "v_max_u32", "%[a]")
D32(min3_u32, "v_min3_u32", "%[a], %[b]")
D32(med3_u32, "v_med3_u32", "%[a], %[b]")
D32(max3_f32, "v_max3_f32", "%[a], %[b]")
D32(and_or, "v_and_or_b32", "%[a], %[b]")
D32(lshl_or, "v_lshl_or_b32", "%[b], %[a]")
D32(lshl_add, "v_lshl_add_u32", "%[b], %[a]")
D32(add3, "v_add3_u32", "%[a], %[b]")
D32(bfe, "v_bfe_u32", "%[b], 8")
D32(perm, "v_perm_b32", "%[a], %[b]")
D32(alignbit, "v_alignbit_b32", "%[a], %[b]")
D32R(lshrrev, "v_lshrrev_b32", "%[b]")
D32(mad_u24, "v_mad_u32_u24", "%[a], %[b]")
D32(mul_lo, "v_mul_lo_u32", "%[a]")
D32(cndmask, "v_cndmask_b32", "%[a], vcc")
D32U(mov, "v_mov_b32", "")
D32U(mov_dpp_shr1, "v_mov_b32_dpp", " wave_shr:1 row_mask:0xf bank_mask:0xf")
D32U(mov_dpp_rowshr, "v_mov_b32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf")
D32(add_dpp, "v_add_u32_dpp", "%[a] row_shr:1 row_mask:0xf bank_mask:0xf")
D32(fma_f32, "v_fma_f32", "%[a], %[b]")
D32(add_f32, "v_add_f32", "%[a]")
D32(pk_add_u16, "v_pk_add_u16", "%[a]")
D32(pk_sub_u16_clamp, "v_pk_sub_u16", "%[a] clamp")
D32(pk_max_u16, "v_pk_max_u16", "%[a]")
D32(pk_min_u16, "v_pk_min_u16", "%[a]")
D32R(pk_lshrrev_b16, "v_pk_lshrrev_b16", "%[b]")
D32(pk_mad_u16, "v_pk_mad_u16", "%[a], %[b]")
D32(cvt_pk_u16_u32, "v_cvt_pk_u16_u32", "%[a]")
D32(mad_u16, "v_mad_u16", "%[a], %[b]")
D32(add_u16, "v_add_u16", "%[a]")
D32(sad_u16, "v_sad_u16", "%[a], %[b]")
D32(sad_u8, "v_sad_u8", "%[a], %[b]")
D32(dot2_u32_u16, "v_dot2_u32_u16", "%[a], %[b]")
D32(dot4_u32_u8, "v_dot4_u32_u8", "%[a], %[b]")
D64(pk_add_f32, "v_pk_add_f32", "%[a]")
D64(pk_fma_f32, "v_pk_fma_f32", "%[a], %[b]")
D64(pk_mul_f32, "v_pk_mul_f32", "%[a]")
D64(add_f64, "v_add_f64", "%[a]")
D64(fma_f64, "v_fma_f64", "%[a], %[b]")
D32(and_b32, "v_and_b32", "%[a]")
D32(or_b32, "v_or_b32", "%[a]")
D32(xor_b32, "v_xor_b32", "%[a]")
D32(min_u32, "v_min_u32", "%[a]")
D32(sub_u32, "v_sub_u32", "%[a]")
D32R(lshlrev, "v_lshlrev_b32", "%[b]")
D32(mul_f32, "v_mul_f32", "%[a]")
D32(max_f32, "v_max_f32", "%[a]")
D32(mul_u32_u24, "v_mul_u32_u24", "%[a]")
D32(bfi, "v_bfi_b32", "%[a], %[b]")
D32(pk_sub_i16, "v_pk_sub_i16", "%[a]")
D32(pk_mul_lo_u16, "v_pk_mul_lo_u16", "%[a]")
D32R(pk_lshlrev_b16, "v_pk_lshlrev_b16", "%[b]")
D32U(cvt_f32_u32, "v_cvt_f32_u32", "")
D32U(mbcnt_lo, "v_mbcnt_lo_u32_b32", ", 0")
// a select on a mask that nothing rewrites (the `cndmask` entry above selects on vcc right behind instructions that clobber it)
DEF32(cndmask_static, "v_cndmask_b32 %[r0], %[r0], %[a], s[20:21]\n", "v_cndmask_b32 %[r1], %[r1], %[a], s[20:21]\n", "v_cndmask_b32 %[r2], %[r2], %[a], s[20:21]\n",
      "v_cndmask_b32 %[r3], %[r3], %[a], s[20:21]\n", "v_cndmask_b32 %[r4], %[r4], %[a], s[20:21]\n", "v_cndmask_b32 %[r5], %[r5], %[a], s[20:21]\n",
      "v_cndmask_b32 %[r6], %[r6], %[a], s[20:21]\n", "v_cndmask_b32 %[r7], %[r7], %[a], s[20:21]\n")
// eight s_nop 0 / eight s_waitcnt that never wait, between nothing: what an issued no-op costs
DEF32(s_nop0, "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n", "s_nop 0\n")
// one vector add + one s_nop 0 / + one scalar add: does the other instruction take an issue slot of the SIMD?
DEF32(valu_nop_mix, "v_add_u32 %[r0], %[r0], %[a]\n", "s_nop 0\n", "v_add_u32 %[r2], %[r2], %[a]\n", "s_nop 0\n",
      "v_add_u32 %[r4], %[r4], %[a]\n", "s_nop 0\n", "v_add_u32 %[r6], %[r6], %[a]\n", "s_nop 0\n")
// v_cmp writing an SGPR pair + dependent cndmask would serialise; a lone compare into vcc
DEF32(cmp_lt_u32, "v_cmp_lt_u32 vcc, %[r0], %[a]\n", "v_cmp_lt_u32 vcc, %[r1], %[a]\n", "v_cmp_lt_u32 vcc, %[r2], %[a]\n", "v_cmp_lt_u32 vcc, %[r3], %[a]\n",
      "v_cmp_lt_u32 vcc, %[r4], %[a]\n", "v_cmp_lt_u32 vcc, %[r5], %[a]\n", "v_cmp_lt_u32 vcc, %[r6], %[a]\n", "v_cmp_lt_u32 vcc, %[r7], %[a]\n")
DEF32(readlane, "v_readlane_b32 s20, %[r0], 3\n", "v_readlane_b32 s21, %[r1], 3\n", "v_readlane_b32 s22, %[r2], 3\n", "v_readlane_b32 s23, %[r3], 3\n",
      "v_readlane_b32 s20, %[r4], 3\n", "v_readlane_b32 s21, %[r5], 3\n", "v_readlane_b32 s22, %[r6], 3\n", "v_readlane_b32 s23, %[r7], 3\n")
DEF32(salu_add, "s_add_u32 s20, s20, %[s]\n", "s_add_u32 s21, s21, %[s]\n", "s_add_u32 s22, s22, %[s]\n", "s_add_u32 s23, s23, %[s]\n",
      "s_add_u32 s24, s24, %[s]\n", "s_add_u32 s25, s25, %[s]\n", "s_add_u32 s26, s26, %[s]\n", "s_add_u32 s27, s27, %[s]\n")
DEF32(salu_bcnt64, "s_bcnt1_i32_b64 s20, vcc\n", "s_bcnt1_i32_b64 s21, vcc\n", "s_bcnt1_i32_b64 s22, vcc\n", "s_bcnt1_i32_b64 s23, vcc\n",
      "s_bcnt1_i32_b64 s24, vcc\n", "s_bcnt1_i32_b64 s25, vcc\n", "s_bcnt1_i32_b64 s26, vcc\n", "s_bcnt1_i32_b64 s27, vcc\n")
// half VALU half SALU, interleaved
DEF32(valu_salu_mix, "v_add_u32 %[r0], %[r0], %[a]\n", "s_add_u32 s20, s20, %[s]\n", "v_add_u32 %[r2], %[r2], %[a]\n", "s_add_u32 s21, s21, %[s]\n",
      "v_add_u32 %[r4], %[r4], %[a]\n", "s_add_u32 s22, s22, %[s]\n", "v_add_u32 %[r6], %[r6], %[a]\n", "s_add_u32 s23, s23, %[s]\n")

// waves 0 and 2 of a block run the vector-add stream, waves 1 and 3 the scalar-add stream: do streams of different waves share the
// SIMD's issue?  (a block's four waves sit on the four SIMDs; with 2+ blocks per CU every SIMD holds both kinds)
__global__ __launch_bounds__(256) void k_split_valu_salu(const unsigned *in, unsigned *out, long long *cyc, int trips)
{
    unsigned r0 = in[threadIdx.x], r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    unsigned a = in[64 + (threadIdx.x & 63)] | 1u, b = in[128 + (threadIdx.x & 63)] & 15u;
    unsigned sv = __builtin_amdgcn_readfirstlane(in[3]);
    const bool vec = (((threadIdx.x >> 6) + blockIdx.x) & 1) == 0;          // alternate by block too: every SIMD gets both kinds
    const long long t0 = __builtin_readcyclecounter();
    if (vec) {
        for (int it = 0; it < trips; it++)
            asm volatile(BODY(VADD) OPS32);
    } else {
        for (int it = 0; it < trips; it++)
            asm volatile(BODY(SADD) OPS32);
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

// LDS: bpermute and atomics (own kernels: need shared memory)
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(const unsigned *in, unsigned *out, long long *cyc, int trips)
{
    __shared__ unsigned sm[4 * 1024];
    unsigned *h = sm + (threadIdx.x >> 6) * 1024;
    for (int i = threadIdx.x & 63; i < 1024; i += 64) h[i] = 0;
    unsigned r0 = in[threadIdx.x], acc = 0;
    const int lane = threadIdx.x & 63;
    // MODE 0: ds_bpermute; 1: ds_add_u32 conflict-free (lane's own word); 2: ds_add_u32 random bins of 256; 3: all lanes one address;
    // 4: 8 lanes one address, rest own word; 5: ds_add_rtn_u32 own word; 6: ds_read_b32 own word; 7: ds_write_b32 own word
    unsigned idx = MODE == 2 ? (in[256 + threadIdx.x] & 255u) : (MODE == 3 ? 0u : (MODE == 4 ? (lane < 8 ? 0u : (unsigned)lane) : (unsigned)lane));
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < trips; it++) {
#pragma unroll
        for (int u = 0; u < 64; u++) {
            if (MODE == 0) r0 = (unsigned)__builtin_amdgcn_ds_bpermute((int)(((lane + 1) & 63) << 2), (int)r0);
            else if (MODE == 5) acc += atomicAdd(&h[idx], 1u);
            else if (MODE == 6) { acc += *(volatile unsigned *)&h[idx + (u & 7) * 64]; }
            else if (MODE == 7) { *(volatile unsigned *)&h[idx + (u & 7) * 64] = r0; }
            else {
                if (MODE == 2) idx = (idx * 5u + 1u) & 255u;
                __hip_atomic_fetch_add(&h[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = r0 + acc + h[lane];
}

static unsigned *d_in, *d_out;
static long long *d_cyc;

typedef void (*kern_t)(const unsigned *, unsigned *, long long *, int);

static double run(kern_t k, int wps, int trips)
{
    const int blocks = 256 * wps;
    hipMemset(d_cyc, 0, sizeof(long long) * blocks * 4);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_in, d_out, d_cyc, trips); hipDeviceSynchronize(); }
    long long *h = (long long *)malloc(sizeof(long long) * blocks * 4);
    hipMemcpy(h, d_cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    // median over waves (the block -> CU placement is not perfectly even)
    long long mx = 0;
    double sum = 0;
    for (int i = 0; i < blocks * 4; i++) { if (h[i] > mx) mx = h[i]; sum += (double)h[i]; }
    free(h);
    return sum / (blocks * 4) / ((double)trips * 64.0 * wps);
}

int main()
{
    hipMalloc(&d_in, 1 << 16);
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    hipMalloc(&d_cyc, sizeof(long long) * 256 * 8 * 4);
    unsigned hin[4096];
    for (int i = 0; i < 4096; i++) hin[i] = (unsigned)(i * 2654435761u) >> 7;
    hipMemcpy(d_in, hin, sizeof(hin), hipMemcpyHostToDevice);
    const int trips = 4000;
    struct { const char *name; kern_t k; } tab[] = {
#define E(N) {#N, k_##N}
        E(add_u32), E(sub_u32_clamp), E(max_u32), E(min3_u32), E(med3_u32), E(max3_f32), E(and_or), E(lshl_or), E(lshl_add), E(add3), E(bfe), E(perm),
        E(alignbit), E(lshrrev), E(mad_u24), E(mul_lo), E(cndmask), E(mov), E(mov_dpp_shr1), E(mov_dpp_rowshr), E(add_dpp), E(fma_f32), E(add_f32),
        E(pk_add_u16), E(pk_sub_u16_clamp), E(pk_max_u16), E(pk_min_u16), E(pk_lshrrev_b16), E(pk_mad_u16), E(cvt_pk_u16_u32), E(mad_u16), E(add_u16),
        E(sad_u16), E(sad_u8), E(dot2_u32_u16), E(dot4_u32_u8), E(pk_add_f32), E(pk_fma_f32), E(pk_mul_f32), E(add_f64), E(fma_f64), E(cmp_lt_u32),
        E(readlane), E(salu_add), E(salu_bcnt64), E(valu_salu_mix), E(and_b32), E(or_b32), E(xor_b32), E(min_u32), E(sub_u32), E(lshlrev),
        E(mul_f32), E(max_f32), E(mul_u32_u24), E(bfi), E(pk_sub_i16), E(pk_mul_lo_u16), E(pk_lshlrev_b16), E(cvt_f32_u32), E(mbcnt_lo),
        E(cndmask_static), E(s_nop0), E(valu_nop_mix), E(split_valu_salu),
        {"ds_bpermute", k_lds<0>}, {"ds_add_u32 own word", k_lds<1>}, {"ds_add_u32 256 random bins", k_lds<2>}, {"ds_add_u32 one address", k_lds<3>},
        {"ds_add_u32 8 lanes one address", k_lds<4>}, {"ds_add_rtn_u32 own word", k_lds<5>}, {"ds_read_b32", k_lds<6>}, {"ds_write_b32", k_lds<7>},
    };
    printf("%-32s %10s %10s %10s %10s   (cycles per wave-instruction per SIMD; waves per SIMD = 1, 2, 4, 8)\n", "instruction", "1", "2", "4", "8");
    for (auto &t : tab) {
        const int tr = (t.name[0] == 'd' && t.name[1] == 's') ? trips / 8 : trips;
        printf("%-32s %10.2f %10.2f %10.2f %10.2f\n", t.name, run(t.k, 1, tr), run(t.k, 2, tr), run(t.k, 4, tr), run(t.k, 8, tr));
        fflush(stdout);
    }
    return 0;
}
