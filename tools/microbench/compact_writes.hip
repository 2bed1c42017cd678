// What does queue compaction cost in WRITE bandwidth?  Every wave appends `len` float4 (a fraction of its lanes, like the
// continuing paths of a chunk) to a stream at a position it reserved -- the writes of k_shade.  Modes:
//   atomic     one returning atomicAdd per wave on ONE counter: adjacent ranges belong to waves of different XCDs (different L2s)
//   atomic8    one counter and one region per XCD (blockIdx % 8): adjacent ranges belong to waves of the same XCD
//   own        no atomic: a wave's ranges are consecutive in its own region (adjacent ranges: the same wave, one L2)
//   interleave no atomic: range = (iteration, wave) in wave order: adjacent ranges belong to neighbouring waves / workgroups
//   full       atomic, len = 64: every range is 1 KiB, line-aligned
//   hipcc --offload-arch=gfx950 -O3 -o compact_writes compact_writes.hip && ./compact_writes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_write(float4 *out, int *counters, int iters, int len, long long region)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    const float4 v = make_float4((float)gw, (float)lane, 1.0f, 2.0f);
    for (int it = 0; it < iters; it++) {
        long long base;
        if (MODE == 0 || MODE == 4) {
            int b = 0;
            if (lane == 0) b = atomicAdd(&counters[0], len);
            base = __builtin_amdgcn_readfirstlane(b);
        } else if (MODE == 1) {
            const int x = blockIdx.x & 7;
            int b = 0;
            if (lane == 0) b = atomicAdd(&counters[x * 32], len);
            base = (long long)x * region + __builtin_amdgcn_readfirstlane(b);
        } else if (MODE == 2) {
            base = (gw * iters + it) * len;
        } else {
            base = ((long long)it * nw + gw) * len;
        }
        if (lane < len) out[base + lane] = v;
    }
}

int main(int argc, char **argv)
{
    const int grid = 4096, iters = 256;
    const long long total = (long long)grid * 4 * iters * 64;   // float4 slots, enough for len = 64
    float4 *out; CK(hipMalloc(&out, total * 16));
    int *counters; CK(hipMalloc(&counters, 8 * 32 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char *name; int mode, len; } cases[] = {{"atomic len 40", 0, 40}, {"atomic8 len 40", 1, 40}, {"own len 40", 2, 40}, {"interleave len 40", 3, 40},
                                                          {"full len 64 atomic", 4, 64}, {"atomic len 24", 0, 24}, {"atomic8 len 24", 1, 24}, {"own len 24", 2, 24},
                                                          {"atomic len 56", 0, 56}, {"atomic8 len 56", 1, 56}};
    for (int rep = 0; rep < 2; rep++)
        for (auto &c : cases) {
            CK(hipMemset(counters, 0, 8 * 32 * 4));
            CK(hipDeviceSynchronize());
            const long long region = total / 8;
            CK(hipEventRecord(e0, 0));
            switch (c.mode) {
            case 0: hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, 0, out, counters, iters, c.len, region); break;
            case 1: hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, 0, out, counters, iters, c.len, region); break;
            case 2: hipLaunchKernelGGL(k_write<2>, dim3(grid), dim3(256), 0, 0, out, counters, iters, c.len, region); break;
            case 3: hipLaunchKernelGGL(k_write<3>, dim3(grid), dim3(256), 0, 0, out, counters, iters, c.len, region); break;
            default: hipLaunchKernelGGL(k_write<4>, dim3(grid), dim3(256), 0, 0, out, counters, iters, c.len, region); break;
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)grid * 4 * iters * c.len * 16;
            printf("{\"case\": \"%s\", \"ms\": %.3f, \"GB\": %.2f, \"TBps\": %.3f}\n", c.name, ms, bytes / 1e9, bytes / (ms * 1e-3) / 1e12);
        }
    return 0;
}
