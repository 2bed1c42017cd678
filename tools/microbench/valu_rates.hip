// Issue cost of the vector instructions the path-tracing kernels are made of, per wave64 instruction and SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/microbench/valu_rates.hip && ./valu_rates
// Every SIMD runs WAVES waves that each issue ITER x 8 independent instances of one instruction (8 accumulators, no
// dependency closer than 8 instructions); cycles = time x clock x 1024 SIMDs / (waves x ITER x 8).  The clock is read with
// s_memrealtime-free arithmetic: we report ns per instruction per SIMD and cycles at the clock measured by a v_fma loop
// known to take 4 cycles (MI355X_MICROARCH.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed, unsigned useed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = useed + threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 * 9, u5 = u0 * 11, u6 = u0 * 13, u7 = u0 * 15;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float b = seed * 0.5f + 1.0f;
    const unsigned ub = useed | 1u;
    const double db = b;
#define R8F(ins) asm volatile(ins : "+v"(a0) : "v"(b)); asm volatile(ins : "+v"(a1) : "v"(b)); asm volatile(ins : "+v"(a2) : "v"(b)); asm volatile(ins : "+v"(a3) : "v"(b)); \
                 asm volatile(ins : "+v"(a4) : "v"(b)); asm volatile(ins : "+v"(a5) : "v"(b)); asm volatile(ins : "+v"(a6) : "v"(b)); asm volatile(ins : "+v"(a7) : "v"(b));
#define R8FC(ins, ...) asm volatile(ins : "+v"(a0) : "v"(b) : __VA_ARGS__); asm volatile(ins : "+v"(a1) : "v"(b) : __VA_ARGS__); asm volatile(ins : "+v"(a2) : "v"(b) : __VA_ARGS__); asm volatile(ins : "+v"(a3) : "v"(b) : __VA_ARGS__); \
                 asm volatile(ins : "+v"(a4) : "v"(b) : __VA_ARGS__); asm volatile(ins : "+v"(a5) : "v"(b) : __VA_ARGS__); asm volatile(ins : "+v"(a6) : "v"(b) : __VA_ARGS__); asm volatile(ins : "+v"(a7) : "v"(b) : __VA_ARGS__);
#define R8U(ins) asm volatile(ins : "+v"(u0) : "v"(ub)); asm volatile(ins : "+v"(u1) : "v"(ub)); asm volatile(ins : "+v"(u2) : "v"(ub)); asm volatile(ins : "+v"(u3) : "v"(ub)); \
                 asm volatile(ins : "+v"(u4) : "v"(ub)); asm volatile(ins : "+v"(u5) : "v"(ub)); asm volatile(ins : "+v"(u6) : "v"(ub)); asm volatile(ins : "+v"(u7) : "v"(ub));
#define R8D(ins) asm volatile(ins : "+v"(d0) : "v"(db)); asm volatile(ins : "+v"(d1) : "v"(db)); asm volatile(ins : "+v"(d2) : "v"(db)); asm volatile(ins : "+v"(d3) : "v"(db)); \
                 asm volatile(ins : "+v"(d4) : "v"(db)); asm volatile(ins : "+v"(d5) : "v"(db)); asm volatile(ins : "+v"(d6) : "v"(db)); asm volatile(ins : "+v"(d7) : "v"(db));
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) { R8F("v_fma_f32 %0, %0, %1, %1") }
        else if (OP == 1) { R8F("v_mul_f32 %0, %0, %1") }
        else if (OP == 2) { R8U("v_mul_lo_u32 %0, %0, %1") }
        else if (OP == 3) { R8U("v_mul_u32_u24 %0, %0, %1") }
        else if (OP == 4) { R8U("v_mad_u32_u24 %0, %0, %1, %1") }
        else if (OP == 5) { R8U("v_xor_b32 %0, %0, %1") }
        else if (OP == 6) { R8U("v_lshrrev_b32 %0, 7, %0") }
        else if (OP == 7) { R8F("v_rcp_f32 %0, %0") }
        else if (OP == 8) { R8F("v_sqrt_f32 %0, %0") }
        else if (OP == 9) { R8F("v_div_fixup_f32 %0, %0, %1, %1") }
        else if (OP == 10) { R8F("v_max3_f32 %0, %0, %1, %1") }
        else if (OP == 11) { R8F("v_cndmask_b32 %0, %0, %1, vcc") }
        else if (OP == 12) { R8D("v_fma_f64 %0, %0, %1, %1") }
        else if (OP == 13) { R8D("v_mul_f64 %0, %0, %1") }
        else if (OP == 14) { R8D("v_add_f64 %0, %0, %1") }
        else if (OP == 15) { R8D("v_rcp_f64 %0, %0") }
        else if (OP == 16) { R8D("v_ldexp_f64 %0, %0, 3") }
        else if (OP == 17) { asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d0) : "v"(u0)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d1) : "v"(u1)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d2) : "v"(u2)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d3) : "v"(u3));
                           asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d4) : "v"(u4)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d5) : "v"(u5)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d6) : "v"(u6)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d7) : "v"(u7)); }
        else if (OP == 18) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a0) : "v"(d0)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a1) : "v"(d1)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a2) : "v"(d2)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a3) : "v"(d3));
                           asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a4) : "v"(d4)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a5) : "v"(d5)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a6) : "v"(d6)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a7) : "v"(d7)); }
        else if (OP == 19) { R8F("v_cvt_f32_u32 %0, %0") }
        else if (OP == 20) { asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 4\n v_readlane_b32 s22, %0, 5\n v_readlane_b32 s23, %0, 6\n v_readlane_b32 s24, %0, 7\n v_readlane_b32 s25, %0, 8\n v_readlane_b32 s26, %0, 9\n v_readlane_b32 s27, %0, 10" : : "v"(a0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); }
        else if (OP == 21) { asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a0) : "v"(b) : "vcc"); asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a1) : "v"(b) : "vcc"); asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a2) : "v"(b) : "vcc"); asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a3) : "v"(b) : "vcc");
                           asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a4) : "v"(b) : "vcc"); asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a5) : "v"(b) : "vcc"); asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a6) : "v"(b) : "vcc"); asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a7) : "v"(b) : "vcc"); }
        else if (OP == 22) { R8F("v_div_fmas_f32 %0, %0, %1, %1") }
        else if (OP == 23) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a0), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a1), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a2), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a3), "v"(b) : "vcc");
                           asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a4), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a5), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a6), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a7), "v"(b) : "vcc"); }
        else if (OP == 24) { R8F("v_mul_f32 %0, %0, s4") }
        else if (OP == 25) { R8F("v_mul_f32 %0, 0x40490fdb, %0") }
        else if (OP == 26) { asm volatile("v_mad_u64_u32 v[40:41], vcc, %0, %1, v[40:41]\n v_mad_u64_u32 v[42:43], vcc, %0, %1, v[42:43]\n v_mad_u64_u32 v[44:45], vcc, %0, %1, v[44:45]\n v_mad_u64_u32 v[46:47], vcc, %0, %1, v[46:47]\n"
                                         "v_mad_u64_u32 v[40:41], vcc, %0, %1, v[40:41]\n v_mad_u64_u32 v[42:43], vcc, %0, %1, v[42:43]\n v_mad_u64_u32 v[44:45], vcc, %0, %1, v[44:45]\n v_mad_u64_u32 v[46:47], vcc, %0, %1, v[46:47]"
                                         : : "v"(u0), "v"(ub) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc"); }
        else if (OP == 27) { R8U("v_mul_hi_u32 %0, %0, %1") }
        else if (OP == 28) { R8F("v_frexp_exp_i32_f32 %0, %0") }
        // round 4: packed FP32 (two results per lane), the two-operand maximum, selects, modifiers
        else if (OP == 29) { asm volatile("v_pk_fma_f32 v[40:41], v[40:41], v[56:57], v[56:57]\n v_pk_fma_f32 v[42:43], v[42:43], v[56:57], v[56:57]\n v_pk_fma_f32 v[44:45], v[44:45], v[56:57], v[56:57]\n v_pk_fma_f32 v[46:47], v[46:47], v[56:57], v[56:57]\n"
                                         "v_pk_fma_f32 v[48:49], v[48:49], v[56:57], v[56:57]\n v_pk_fma_f32 v[50:51], v[50:51], v[56:57], v[56:57]\n v_pk_fma_f32 v[52:53], v[52:53], v[56:57], v[56:57]\n v_pk_fma_f32 v[54:55], v[54:55], v[56:57], v[56:57]"
                                         : : : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"); }
        else if (OP == 30) { asm volatile("v_pk_mul_f32 v[40:41], v[40:41], v[56:57]\n v_pk_mul_f32 v[42:43], v[42:43], v[56:57]\n v_pk_mul_f32 v[44:45], v[44:45], v[56:57]\n v_pk_mul_f32 v[46:47], v[46:47], v[56:57]\n"
                                         "v_pk_mul_f32 v[48:49], v[48:49], v[56:57]\n v_pk_mul_f32 v[50:51], v[50:51], v[56:57]\n v_pk_mul_f32 v[52:53], v[52:53], v[56:57]\n v_pk_mul_f32 v[54:55], v[54:55], v[56:57]"
                                         : : : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"); }
        else if (OP == 31) { asm volatile("v_pk_fma_f32 v[40:41], v[40:41], v[56:57], v[56:57] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[42:43], v[42:43], v[56:57], v[56:57] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[44:45], v[44:45], v[56:57], v[56:57] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[46:47], v[46:47], v[56:57], v[56:57] op_sel_hi:[1,0,0]\n"
                                         "v_pk_fma_f32 v[48:49], v[48:49], v[56:57], v[56:57] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[50:51], v[50:51], v[56:57], v[56:57] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[52:53], v[52:53], v[56:57], v[56:57] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[54:55], v[54:55], v[56:57], v[56:57] op_sel_hi:[1,0,0]"
                                         : : : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"); }
        else if (OP == 32) { R8F("v_max_f32 %0, %0, %1") }
        else if (OP == 33) { R8F("v_add_f32 %0, %0, %1") }
        else if (OP == 34) { R8F("v_sub_f32 %0, 0x40490fdb, %0") }
        else if (OP == 35) { R8F("v_sub_f32_e64 %0, |%0|, s4") }
        else if (OP == 36) { R8F("v_cndmask_b32_e32 %0, %0, %1, vcc") }
        else if (OP == 37) { R8F("v_max3_f32 %0, |%0|, %1, %1") }
        else if (OP == 38) { R8F("v_fma_f32 %0, %0, %1, %0") }
        else if (OP == 39) { R8F("v_fmac_f32 %0, %1, %1") }
        else if (OP == 40) { R8F("v_and_or_b32 %0, %0, %1, %1") }
        else if (OP == 41) { R8F("v_med3_f32 %0, %0, %1, %1") }
        else if (OP == 42) { R8F("v_min_u32 %0, %0, %1") }
        else if (OP == 43) { R8FC("v_cndmask_b32_e64 %0, %0, %1, s[60:61]", "s60", "s61") }
        else if (OP == 44) { R8FC("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc", "vcc") }   // two instructions per instance
        else if (OP == 45) { R8FC("v_cmp_gt_f32 s[60:61], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[60:61]", "s60", "s61") }
        else if (OP == 46) { R8FC("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %0, %1, %0, vcc", "vcc") }   // three
        else if (OP == 47) { R8F("v_mov_b32 %0, %1") }
        else if (OP == 48) { R8FC("v_cndmask_b32_e32 %0, %1, %0, vcc", "vcc") }
        else if (OP == 49) { R8F("v_bfi_b32 %0, %0, %1, %1") }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <int OP>
static double run(float *d_out, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float *d_out;
    hipMalloc(&d_out, sizeof(float) * blocks * 256);
    const char *names[] = {"v_fma_f32", "v_mul_f32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_xor_b32", "v_lshrrev_b32", "v_rcp_f32", "v_sqrt_f32",
                           "v_div_fixup_f32", "v_max3_f32", "v_cndmask_b32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_ldexp_f64", "v_cvt_f64_u32",
                           "v_cvt_f32_f64", "v_cvt_f32_u32", "v_readlane_b32", "v_div_scale_f32", "v_div_fmas_f32", "v_cmp_gt_f32", "v_mul_f32 sgpr", "v_mul_f32 literal",
                           "v_mad_u64_u32", "v_mul_hi_u32", "v_frexp_exp_i32_f32",
                           "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_fma_f32 op_sel splat", "v_max_f32", "v_add_f32", "v_sub_f32 literal", "v_sub_f32 |v| sgpr", "v_cndmask_b32 e32",
                           "v_max3_f32 |v|", "v_fma_f32 2 regs", "v_fmac_f32", "v_and_or_b32", "v_med3_f32", "v_min_u32",
                           "v_cndmask_b32 e64 sgpr", "v_cmp+v_cndmask vcc (2)", "v_cmp+v_cndmask sgpr (2)", "v_cmp+2 v_cndmask (3)", "v_mov_b32", "v_cndmask_b32 e32 swapped", "v_bfi_b32"};
    double ms[50];
#define RUN(i) ms[i] = run<i>(d_out, blocks);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20)
    RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28) RUN(29) RUN(30) RUN(31) RUN(32) RUN(33) RUN(34) RUN(35) RUN(36) RUN(37) RUN(38) RUN(39) RUN(40) RUN(41) RUN(42) RUN(43) RUN(44) RUN(45) RUN(46) RUN(47) RUN(48) RUN(49)
    // per SIMD: (blocks * 4 waves / 1024 SIMDs) waves x ITER x 8 instructions
    const double per_simd = (double)blocks * 4 / 1024 * ITER * 8;
    const double ns_fma = ms[0] * 1e6 / per_simd;
    printf("{\"waves_per_simd\": %d, \"assumed_v_fma_f32_cycles\": 4, \"ns_per_v_fma_f32\": %.4f, \"implied_clock_GHz\": %.3f, \"cycles\": {", blocks * 4 / 1024, ns_fma, 4.0 / ns_fma);
    for (int i = 0; i < 50; i++) printf("%s\"%s\": %.2f", i ? ", " : "", names[i], ms[i] * 1e6 / per_simd / ns_fma * 4.0);
    printf("}}\n");
    return 0;
}
