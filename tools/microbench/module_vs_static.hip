// Does a kernel loaded with hipModuleLoadData run as fast as the same kernel linked into the binary?  (the
// per-scene build.)  Two kernels -- a 300-byte loop and a ~40 KB straight-line loop body -- built by hipcc into this program
// and by hiprtc at run time from the same text, launched both ways, timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o module_vs_static module_vs_static.hip -lhiprtc && ./module_vs_static
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define KSRC(...) #__VA_ARGS__
static const char *kText = KSRC(
template <int UNROLL>
__device__ __forceinline__ float body(float x, float a, float b)
{
    _Pragma("unroll")
    for (int i = 0; i < UNROLL; i++) { x = x * a + b; a = a + 1.0e-7f * (float)(i + 1); b = b - x * 1.0e-9f; }
    return x;
}
extern "C" __global__ void __launch_bounds__(256) k_small(float *out, int iters, float a, float b)
{
    float x = (float)threadIdx.x;
    for (int it = 0; it < iters * 512; it++) x = body<4>(x, a, b);
    if (x == 12345.678f) out[blockIdx.x] = x;
}
extern "C" __global__ void __launch_bounds__(256) k_big(float *out, int iters, float a, float b)
{
    float x = (float)threadIdx.x;
    for (int it = 0; it < iters; it++) x = body<2048>(x, a, b);
    if (x == 12345.678f) out[blockIdx.x] = x;
}
extern "C" __global__ void __launch_bounds__(256) k_mem(float *out, const float4 *in, int iters, long long n)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++)
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 12345.678f) out[blockIdx.x] = acc.y + acc.z + acc.w;
}

struct Big { float v[220]; };
extern "C" __global__ void __launch_bounds__(256) k_karg(float *out, Big big, int iters, int mask)
{
    float x = (float)threadIdx.x;
    for (int it = 0; it < iters * 4096; it++) {
        const int j = (it * 7 + (int)blockIdx.x) & mask;   // wave-uniform, data-dependent: a scalar load from the kernarg segment per step
        x = x * 0.999f + big.v[j];
    }
    if (x == 12345.678f) out[blockIdx.x] = x;
}
);
// the same text compiled into this program
template <int UNROLL>
__device__ __forceinline__ float body(float x, float a, float b)
{
#pragma unroll
    for (int i = 0; i < UNROLL; i++) { x = x * a + b; a = a + 1.0e-7f * (float)(i + 1); b = b - x * 1.0e-9f; }
    return x;
}
extern "C" __global__ void __launch_bounds__(256) k_small(float *out, int iters, float a, float b)
{
    float x = (float)threadIdx.x;
    for (int it = 0; it < iters * 512; it++) x = body<4>(x, a, b);
    if (x == 12345.678f) out[blockIdx.x] = x;
}
extern "C" __global__ void __launch_bounds__(256) k_big(float *out, int iters, float a, float b)
{
    float x = (float)threadIdx.x;
    for (int it = 0; it < iters; it++) x = body<2048>(x, a, b);
    if (x == 12345.678f) out[blockIdx.x] = x;
}
extern "C" __global__ void __launch_bounds__(256) k_mem(float *out, const float4 *in, int iters, long long n)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++)
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 12345.678f) out[blockIdx.x] = acc.y + acc.z + acc.w;
}


struct Big { float v[220]; };
extern "C" __global__ void __launch_bounds__(256) k_karg(float *out, Big big, int iters, int mask)
{
    float x = (float)threadIdx.x;
    for (int it = 0; it < iters * 4096; it++) {
        const int j = (it * 7 + (int)blockIdx.x) & mask;   // wave-uniform, data-dependent: a scalar load from the kernarg segment per step
        x = x * 0.999f + big.v[j];
    }
    if (x == 12345.678f) out[blockIdx.x] = x;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    hiprtcProgram prog;
    std::string src = kText;
    // the stringified text lost its _Pragma line structure only in spelling; hiprtc takes it as is
    if (hiprtcCreateProgram(&prog, src.c_str(), "m.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { fprintf(stderr, "create failed\n"); return 1; }
    const char *opts[] = {"--offload-arch=gfx950", "-O3"};
    if (hiprtcCompileProgram(prog, 2, opts) != HIPRTC_SUCCESS) {
        size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls); std::string log(ls, 0); hiprtcGetProgramLog(prog, &log[0]); fprintf(stderr, "compile failed:\n%s\n", log.c_str()); return 1;
    }
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs); std::vector<char> code(cs); hiprtcGetCode(prog, code.data());
    fprintf(stderr, "hiprtc code object: %zu bytes\n", cs);
    float *out; CK(hipMalloc(&out, 4096 * 4));
    const long long n = 64ll << 20;   // 1 GiB of float4
    float4 *in; CK(hipMalloc(&in, n * 16)); CK(hipMemset(in, 0, n * 16));
    hipModule_t mod; CK(hipModuleLoadData(&mod, code.data()));
    hipFunction_t f_small, f_big, f_mem, f_karg;
    CK(hipModuleGetFunction(&f_karg, mod, "k_karg"));
    Big big; for (int i = 0; i < 220; i++) big.v[i] = 1e-3f * i;
    int mask = 127, it_k = 4;
    CK(hipModuleGetFunction(&f_small, mod, "k_small")); CK(hipModuleGetFunction(&f_big, mod, "k_big")); CK(hipModuleGetFunction(&f_mem, mod, "k_mem"));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s; CK(hipStreamCreate(&s));
    int iters = 4; float a = 1.0000001f, b = 1e-6f;
    int it_mem = 2; long long nn = n;
    auto time = [&](const char *name, auto launch) -> int {
        launch(); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; r++) launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"%s\", \"ms_per_launch\": %.4f}\n", name, ms / 10);
        return 0;
    };
    for (int rep = 0; rep < 2; rep++) {
        time("static k_small", [&] { hipLaunchKernelGGL(k_small, dim3(4096), dim3(256), 0, s, out, iters, a, b); });
        time("module k_small", [&] { void *args[] = {&out, &iters, &a, &b}; (void)hipModuleLaunchKernel(f_small, 4096, 1, 1, 256, 1, 1, 0, s, args, nullptr); });
        time("static k_big", [&] { hipLaunchKernelGGL(k_big, dim3(4096), dim3(256), 0, s, out, iters, a, b); });
        time("module k_big", [&] { void *args[] = {&out, &iters, &a, &b}; (void)hipModuleLaunchKernel(f_big, 4096, 1, 1, 256, 1, 1, 0, s, args, nullptr); });
        time("static k_karg", [&] { hipLaunchKernelGGL(k_karg, dim3(4096), dim3(256), 0, s, out, big, it_k, mask); });
        time("module k_karg", [&] { void *args[] = {&out, &big, &it_k, &mask}; (void)hipModuleLaunchKernel(f_karg, 4096, 1, 1, 256, 1, 1, 0, s, args, nullptr); });
        time("static k_mem", [&] { hipLaunchKernelGGL(k_mem, dim3(4096), dim3(256), 0, s, out, (const float4 *)in, it_mem, nn); });
        time("module k_mem", [&] { void *args[] = {&out, &in, &it_mem, &nn}; (void)hipModuleLaunchKernel(f_mem, 4096, 1, 1, 256, 1, 1, 0, s, args, nullptr); });
    }
    return 0;
}
