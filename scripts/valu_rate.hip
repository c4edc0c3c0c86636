// valu_rate.hip -- what does one gfx950 SIMD sustain per vector instruction, by instruction kind and by the number
// of resident waves?  (the compositing kernels are VALU-issue-bound; this pins the denominator.)
//   hipcc -O3 --offload-arch=gfx950 scripts/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
// Every wave runs ITER x 64 independent instructions of one kind (8 accumulators, no dependency stalls) and
// stamps s_memtime around the loop; cycles per instruction per SIMD = wave cycles x resident waves / instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 256
#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0, cnt = 0;
    unsigned long long m64 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // kinds with scalar instructions: the whole loop is one asm statement (scalar values carried around a C loop through asm
    // outputs count as divergent and land in VGPRs); ITER x 64 instructions as elsewhere
#define SLOOP(body) "s_mov_b32 %[cnt], 256\n s_mov_b32 %[s0], 1\n s_mov_b32 %[s1], 2\n s_mov_b32 %[s2], 3\n s_mov_b32 %[s3], 4\n s_mov_b64 %[m], -1\n" \
                    "1:\n .rept 8\n" body ".endr\n s_sub_u32 %[cnt], %[cnt], 1\n s_cmp_lg_u32 %[cnt], 0\n s_cbranch_scc1 1b\n"
#define SOUT [cnt] "=&s"(cnt), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [m] "=&s"(m64)
    if (KIND == 9) {          // scalar ALU only: 4 independent s_add_u32 chains
        asm volatile(SLOOP("s_add_u32 %[s0], %[s0], 3\n s_add_u32 %[s1], %[s1], 5\n s_add_u32 %[s2], %[s2], 7\n s_add_u32 %[s3], %[s3], 9\n"
                           "s_add_u32 %[s0], %[s0], 3\n s_add_u32 %[s1], %[s1], 5\n s_add_u32 %[s2], %[s2], 7\n s_add_u32 %[s3], %[s3], 9\n")
                     : SOUT : : "scc");
    } else if (KIND == 10) {  // one scalar after every vector instruction
        asm volatile(SLOOP("v_fma_f32 %[a0], %[a0], %[b], %[c]\n s_add_u32 %[s0], %[s0], 3\n v_fma_f32 %[a1], %[a1], %[b], %[c]\n s_add_u32 %[s1], %[s1], 5\n"
                           "v_fma_f32 %[a2], %[a2], %[b], %[c]\n s_add_u32 %[s2], %[s2], 7\n v_fma_f32 %[a3], %[a3], %[b], %[c]\n s_add_u32 %[s3], %[s3], 9\n")
                     : SOUT, [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3) : [b] "v"(b), [c] "v"(c) : "scc");
    } else if (KIND == 11) {  // the forward compositing loop's mix: 5 vector to 3 scalar
        asm volatile(SLOOP("v_fma_f32 %[a0], %[a0], %[b], %[c]\n v_fma_f32 %[a1], %[a1], %[b], %[c]\n s_add_u32 %[s0], %[s0], 3\n v_fma_f32 %[a2], %[a2], %[b], %[c]\n"
                           "s_add_u32 %[s1], %[s1], 5\n v_fma_f32 %[a3], %[a3], %[b], %[c]\n v_fma_f32 %[a4], %[a4], %[b], %[c]\n s_add_u32 %[s2], %[s2], 7\n")
                     : SOUT, [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4) : [b] "v"(b), [c] "v"(c) : "scc");
    } else if (KIND == 12) {  // set-bit walk as the compositing kernels do it: s_ff1 + s_bitset0 + v_readlane of the found lane (dependent chain)
        asm volatile(SLOOP("s_ff1_i32_b64 %[s0], %[m]\n s_bitset0_b64 %[m], %[s0]\n v_readlane_b32 %[s1], %[a0], %[s0]\n s_ff1_i32_b64 %[s2], %[m]\n"
                           "s_bitset0_b64 %[m], %[s2]\n v_readlane_b32 %[s3], %[a0], %[s2]\n s_or_b64 %[m], %[m], 0xff\n s_add_u32 %[s1], %[s1], %[s3]\n")
                     : SOUT, [a0] "+v"(a0) : : "scc");
    } else
    for (int i = 0; i < ITER; i++) {
        if (KIND == 0) {          // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) {   // v_pk_fma_f32 (2 lanes-worth per instruction)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (KIND == 2) {   // v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) {   // v_cmp + v_cndmask pairs (counted as 2 instructions each)
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                              "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
        } else if (KIND == 4) {   // v_mul_f32 + v_add_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                              "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 5) {   // v_min_f32 / v_max_f32
            REP8(asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %9\n"
                              "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 6) {   // DPP add (row_shr:1)
            REP8(asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 7) {   // v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 8) {   // v_mfma_f32_16x16x4_f32 (4 independent accumulators)
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 m0 = {a0, a1, a2, a3}, m1 = {a4, a5, a6, a7}, m2 = m0, m3 = m1;
            REP8(m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m1, 0, 0, 0);
                 m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m2, 0, 0, 0); m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m3, 0, 0, 0);
                 m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m1, 0, 0, 0);
                 m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m2, 0, 0, 0); m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, m3, 0, 0, 0);)
            a0 += m0[0] + m1[1] + m2[2] + m3[3];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    a0 += p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + (float)(s0 + s1 + s2 + s3 + cnt) + (float)(unsigned)m64;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int waves_per_simd, double flop_per_inst) {
    const int blocks = 256 * waves_per_simd;     // 256-thread blocks = 4 waves = one per SIMD; waves_per_simd blocks per CU
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&cyc, (size_t)blocks * 4 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
    (void)hipDeviceSynchronize();
    const int reps = 20;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double inst = (double)ITER * 64;
    const double med = (double)h[h.size() / 2];
    // s_memtime ticks at a constant 100 MHz on gfx9? -> report both: ticks per instruction, and wall-derived
    const double wall_per_inst_ns = (double)ms * 1e6 / reps / inst / waves_per_simd;   // per SIMD (all waves of a SIMD interleave)
    const double tflops = flop_per_inst * 64 * inst * blocks * 4 / ((double)ms * 1e-3 / reps) / 1e12;
    printf("%-22s waves/SIMD=%d  memtime ticks/inst/wave=%.3f  wall ns/inst/SIMD=%.3f (= %.2f cyc @2.4GHz)  %.1f TFLOP/s\n", name,
           waves_per_simd, med / inst, wall_per_inst_ns, wall_per_inst_ns * 2.4, tflops);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32", w, 2);
    for (int w : {1, 2, 4, 8}) run<1>("v_pk_fma_f32", w, 4);
    for (int w : {1, 4}) run<4>("v_mul/v_add", w, 1);
    for (int w : {1, 4}) run<5>("v_min/v_max", w, 1);
    for (int w : {1, 4}) run<3>("v_cmp+v_cndmask", w, 1);
    for (int w : {1, 4}) run<2>("v_exp_f32", w, 1);
    for (int w : {1, 4}) run<7>("v_rcp_f32", w, 1);
    for (int w : {1, 4}) run<6>("v_add_f32_dpp", w, 1);
    for (int w : {1, 2, 4}) run<8>("v_mfma_16x16x4_f32", w, 2 * 16 * 16 * 4 / 64.0);
    // scalar instructions: "instructions" below counts scalar and vector alike (TFLOP/s column meaningless)
    for (int w : {1, 2, 4, 8}) run<9>("s_add_u32", w, 0);
    for (int w : {1, 2, 4, 8}) run<10>("v_fma,s_add 1:1", w, 1);
    for (int w : {1, 2, 4, 8}) run<11>("v_fma,s_add 5:3", w, 1.25);
    for (int w : {1, 2, 4, 8}) run<12>("ff1,bitset0,readlane", w, 0);
    return 0;
}
