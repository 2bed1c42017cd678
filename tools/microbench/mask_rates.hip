// What a lane mask costs a select on gfx950: v_cndmask_b32 reading VCC against reading an SGPR pair, by distance from the compare.
//   hipcc --offload-arch=gfx950 -O3 -o mask_rates tools/microbench/mask_rates.hip && ./mask_rates
// Same method as valu_rates.hip: 8 waves per SIMD, each issuing ITER x 8 independent instances of a short sequence; the figure is
// SIMD cycles per INSTANCE of the whole sequence (its instruction count is in the name), at the clock a v_fma_f32 loop implies.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float c0 = a0 * 2, c1 = a1 * 2, c2 = a2 * 2, c3 = a3 * 2, c4 = a4 * 2, c5 = a5 * 2, c6 = a6 * 2, c7 = a7 * 2;
    const float b = seed * 0.5f + 1.0f;
#define ONE(ins, a, c) asm volatile(ins : "+v"(a), "+v"(c) : "v"(b) : "vcc", "s60", "s61");
#define R8(ins) ONE(ins, a0, c0) ONE(ins, a1, c1) ONE(ins, a2, c2) ONE(ins, a3, c3) ONE(ins, a4, c4) ONE(ins, a5, c5) ONE(ins, a6, c6) ONE(ins, a7, c7)
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) { R8("v_fma_f32 %0, %0, %2, %2") }
        else if (OP == 1) { R8("v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc") }
        else if (OP == 2) { R8("v_cmp_gt_f32 vcc, %0, %2\n v_add_f32 %1, %1, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc") }
        else if (OP == 3) { R8("v_cmp_gt_f32 vcc, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc") }
        else if (OP == 4) { R8("v_cndmask_b32_e64 %0, %0, %2, vcc") }
        else if (OP == 5) { R8("v_cmp_gt_f32 s[60:61], %0, %2\n v_cndmask_b32_e64 %0, %0, %2, s[60:61]\n v_cndmask_b32_e64 %1, %1, %2, s[60:61]\n v_cndmask_b32_e64 %0, %2, %0, s[60:61]") }
        else if (OP == 6) { R8("v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc\n v_cndmask_b32_e32 %0, %2, %0, vcc") }
        else if (OP == 7) { R8("v_cmp_gt_f32 vcc, %0, %2\n s_mov_b64 s[60:61], vcc\n v_cndmask_b32_e64 %0, %0, %2, s[60:61]\n v_cndmask_b32_e64 %1, %1, %2, s[60:61]\n v_cndmask_b32_e64 %0, %2, %0, s[60:61]") }
        else if (OP == 8) { R8("v_cmp_gt_f32 s[60:61], %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_cndmask_b32_e64 %0, %0, %2, s[60:61]") }
        else if (OP == 9) { R8("v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_add_f32 %1, %1, %2\n v_cndmask_b32_e32 %0, %2, %0, vcc") }
        else if (OP == 10) { R8("v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2") }
        else if (OP == 11) { R8("v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %0, %2, %0, vcc") }
        else if (OP == 12) { R8("v_cmp_gt_f32 vcc, %0, %2\n v_cmp_lt_f32 s[60:61], %1, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %2, s[60:61]") }
        else if (OP == 13) { R8("v_cmp_gt_f32 vcc, %0, %2") }
        else if (OP == 14) { R8("v_cmp_gt_f32 s[60:61], %0, %2") }
        else if (OP == 15) { R8("v_cmp_gt_f32 vcc, %0, %2\n v_max_f32 %1, %1, %2\n v_max_f32 %1, %1, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc") }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <int OP>
static double run(float *d_out, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    const int blocks = 256 * 8;
    float *d_out;
    hipMalloc(&d_out, sizeof(float) * blocks * 256);
    const char *names[] = {"v_fma_f32 (1)", "cmp vcc + cndmask vcc (2)", "cmp vcc, add, cndmask vcc (3)", "cmp vcc, 3 add, cndmask vcc (5)", "cndmask e64 naming vcc (1)",
                           "cmp sgpr + 3 cndmask sgpr (4)", "cmp vcc + 3 cndmask vcc (4)", "cmp vcc, s_mov to sgpr, 3 cndmask sgpr (4 + 1 scalar)", "cmp sgpr, 3 add, cndmask sgpr (5)",
                           "cmp vcc, cndmask, add, cndmask (4)", "3 add (3)", "cmp vcc + 3 cndmask e64 naming vcc (4)", "cmp vcc, cmp sgpr, cndmask vcc, cndmask sgpr (4)",
                           "cmp vcc (1)", "cmp sgpr (1)", "cmp vcc, 2 max, cndmask vcc (4)"};
    const int N = 16;
    double ms[N];
#define RUN(i) ms[i] = run<i>(d_out, blocks);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
    const double per_simd = (double)blocks * 4 / 1024 * ITER * 8;
    const double ns_fma = ms[0] * 1e6 / per_simd;
    printf("{\"waves_per_simd\": %d, \"ns_per_v_fma_f32\": %.4f, \"cycles_per_instance\": {", blocks * 4 / 1024, ns_fma);
    for (int i = 0; i < N; i++) printf("%s\"%s\": %.2f", i ? ", " : "", names[i], ms[i] * 1e6 / per_simd / ns_fma * 4.0);
    printf("}}\n");
    return 0;
}
