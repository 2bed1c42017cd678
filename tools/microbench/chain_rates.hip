// Dependent chains: what a v_fma_f32 / v_pk_fma_f32 costs when every instruction needs the result of the one before it, by the number
// of waves a SIMD holds -- the exact-division refinement of the sweep (mul, fma, fma, fma, fma) is such a chain.
//   hipcc --offload-arch=gfx950 -O3 -o chain_rates tools/microbench/chain_rates.hip && ./chain_rates
// Figures: SIMD cycles per INSTRUCTION at the clock an independent v_fma_f32 loop implies (4 cycles each).
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 4096
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float b = seed * 0.5f + 1.0f;
    const v2f pb = {b, b};
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) {   // 8 independent fma
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a1) : "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a2) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a3) : "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a4) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a5) : "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a6) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a7) : "v"(b));
        } else if (OP == 1) {   // one chain of 8 fma
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(b));
        } else if (OP == 2) {   // two interleaved chains of fma (8 instructions)
#pragma unroll
            for (int j = 0; j < 4; j++) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a1) : "v"(b)); }
        } else if (OP == 3) {   // one chain of 8 pk_fma
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb));
        } else if (OP == 4) {   // two interleaved chains of pk_fma
#pragma unroll
            for (int j = 0; j < 4; j++) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p1) : "v"(pb)); }
        } else if (OP == 5) {   // 4 independent pk_fma chains (8 instructions)
#pragma unroll
            for (int j = 0; j < 2; j++) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p1) : "v"(pb));
                                          asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2) : "v"(pb)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p3) : "v"(pb)); }
        } else if (OP == 6) {   // one chain of mul (2-cycle class), 8
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(b));
        } else if (OP == 7) {   // one chain of pk_mul, 8
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(pb));
        } else if (OP == 8) {   // a pk_fma whose result feeds plain ops on its halves (8 instructions: pk, add lo, add hi, pk, ...)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(a0) : "v"(p0.x), "v"(b)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(a1) : "v"(p0.y), "v"(b));
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(a2) : "v"(a0), "v"(a1));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int OP>
static double run(float *d_out, int blocks)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *d_out;
    (void)hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256);
    const char *names[] = {"8 independent v_fma_f32", "one chain of v_fma_f32", "two chains of v_fma_f32", "one chain of v_pk_fma_f32", "two chains of v_pk_fma_f32",
                           "four chains of v_pk_fma_f32", "one chain of v_mul_f32", "one chain of v_pk_mul_f32", "pk_fma, add lo, add hi, max (per instruction)"};
    const int N = 9;
    printf("{");
    for (int w = 0; w < 4; w++) {
        const int waves = (w == 0) ? 8 : (w == 1) ? 4 : (w == 2) ? 2 : 1;
        const int blocks = 256 * waves;   // `waves` workgroups of 4 waves per CU = `waves` waves per SIMD
        double ms[N];
#define RUN(i) ms[i] = run<i>(d_out, blocks);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
        const double per_simd = (double)blocks * 4 / 1024 * ITER * 8;
        double ns_fma = ms[0] * 1e6 / per_simd;
        static double ref_ns = 0;
        if (w == 0) ref_ns = ns_fma;   // 8 waves of independent fma: 4 cycles each
        printf("%s\"%d waves per SIMD\": {", w ? ", " : "", waves);
        for (int i = 0; i < N; i++) printf("%s\"%s\": %.2f", i ? ", " : "", names[i], ms[i] * 1e6 / per_simd / ref_ns * 4.0);
        printf("}");
    }
    printf("}\n");
    return 0;
}
