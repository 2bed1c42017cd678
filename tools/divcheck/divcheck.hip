// Exhaustive / randomized check of pt_fdiv.h against the compiler's IEEE f32 division, on the GPU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/divcheck/divcheck.hip -o gpurun_out/divcheck && gpurun_out/divcheck
// Prints one JSON object; "mismatch" must be 0 in every test whose operands satisfy the documented precondition.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../pathtrace_amd/csrc/device/pt_fdiv.h"
using namespace ptd;

struct Res { unsigned long long tested, mismatch; unsigned ex_n, ex_d, ex_got, ex_want; };

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ inline bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__device__ inline void report(Res *r, float n, float d, float got, float want)
{
    if (atomicAdd(&r->mismatch, 1ull) == 0) { r->ex_n = __float_as_uint(n); r->ex_d = __float_as_uint(d); r->ex_got = __float_as_uint(got); r->ex_want = __float_as_uint(want); }
}
// test 0: every denominator bit pattern with biased exponent in [elo, ehi] (both signs), numerators {1, -1, 3, 0.7, 555.001, 1e-3}
__global__ void k_all_denominators(Res *r, int elo, int ehi)
{
    const float nums[6] = {1.0f, -1.0f, 3.0f, 0.7f, 555.001f, 1e-3f};
    unsigned long long cnt = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = (uint32_t)i;
        const int e = (u >> 23) & 255;
        if (e < elo || e > ehi) continue;
        const float d = __uint_as_float(u);
        const float rc = fdiv_rcp(d);
        for (int k = 0; k < 6; k++) {
            const float got = fdiv_q(nums[k], d, rc), want = nums[k] / d;
            cnt++;
            if (!same(got, want)) report(r, nums[k], d, got, want);
        }
    }
    atomicAdd(&r->tested, cnt);
}
// random operands: biased exponents uniform in [nlo, nhi] x [dlo, dhi], random mantissas and signs; mode 1: numerators
// built to sit within a few ulps of a rounding boundary of the quotient (n = RN(q * d) for a random q whose low
// mantissa bits are 0x000 / 0x7ff / 0x800)
__global__ void k_random(Res *r, uint32_t seed, int per_thread, int nlo, int nhi, int dlo, int dhi, int qlo, int qhi, int mode)
{
    uint32_t s = hash32(seed ^ (blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B9u);
    unsigned long long cnt = 0;
    for (int it = 0; it < per_thread; it++) {
        s = hash32(s + 0x632BE5ABu); const uint32_t a = s;
        s = hash32(s + 0x632BE5ABu); const uint32_t b = s;
        s = hash32(s + 0x632BE5ABu); const uint32_t c = s;
        const uint32_t de = dlo + (c % (uint32_t)(dhi - dlo + 1));
        const float d = __uint_as_float((b & 0x807fffffu) | (de << 23));
        float n;
        if (mode == 0) {
            const uint32_t ne = nlo + ((c >> 8) % (uint32_t)(nhi - nlo + 1));
            n = __uint_as_float((a & 0x807fffffu) | (ne << 23));
        } else {
            const uint32_t low = ((c >> 20) & 3u) == 0 ? 0x000u : ((((c >> 20) & 3u) == 1) ? 0x7ffu : (((c >> 20) & 3u) == 2 ? 0x800u : 0x001u));
            const float q = __uint_as_float(((a & 0x807ff000u) | low) | (127u << 23));
            n = q * d;
            n = __uint_as_float(__float_as_uint(n) + (int)((c >> 24) & 7u) - 3);   // a few ulps either side
        }
        const int ne_ = (__float_as_uint(n) >> 23) & 255, de_ = (__float_as_uint(d) >> 23) & 255;
        const int qe = ne_ - de_;   // quotient exponent within +-1
        if (qe < qlo || qe > qhi || ne_ < nlo || ne_ > nhi) continue;
        const float got = fdiv(n, d), want = n / d;
        cnt++;
        if (!same(got, want)) report(r, n, d, got, want);
    }
    atomicAdd(&r->tested, cnt);
}
// special operands through the fixup: zeros, infinities, NaNs, denormal numerators that are exactly zero-free
__global__ void k_special(Res *r)
{
    const uint32_t sp[12] = {0x00000000u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0xffc00001u,
                             0x3f800000u, 0xbf800000u, 0x40490fdbu, 0x44096000u, 0x3a83126fu, 0xc2c80000u};
    const int i = threadIdx.x / 12, j = threadIdx.x % 12;
    if (i >= 12) return;
    const float n = __uint_as_float(sp[i]), d = __uint_as_float(sp[j]);
    const float got = fdiv(n, d), want = n / d;
    atomicAdd(&r->tested, 1ull);
    if (!same(got, want)) report(r, n, d, got, want);
}

static void run(const char *name, Res *dr, bool last = false)
{
    Res h;
    hipDeviceSynchronize();
    hipMemcpy(&h, dr, sizeof h, hipMemcpyDeviceToHost);
    printf("  \"%s\": {\"tested\": %llu, \"mismatch\": %llu, \"example_n_d_got_want\": [\"0x%08x\", \"0x%08x\", \"0x%08x\", \"0x%08x\"]}%s\n",
           name, h.tested, h.mismatch, h.ex_n, h.ex_d, h.ex_got, h.ex_want, last ? "" : ",");
    hipMemset(dr, 0, sizeof h);
}

int main()
{
    Res *dr;
    hipMalloc(&dr, sizeof(Res));
    hipMemset(dr, 0, sizeof(Res));
    printf("{\n");
    // biased exponent e <-> 2^(e-127).  Precondition on d: [2^-125, 2^125] = biased [2, 252]
    k_all_denominators<<<4096, 256>>>(dr, 2, 252);            run("all_denominators_in_precondition", dr);
    k_all_denominators<<<4096, 256>>>(dr, 0, 1);              run("denominators_below_precondition(expected to differ)", dr);
    k_all_denominators<<<4096, 256>>>(dr, 253, 254);          run("denominators_above_precondition(expected to differ)", dr);
    // the traversal's proven range: n, d in [2^-63, 2^43]
    for (int rep = 0; rep < 4; rep++) k_random<<<8192, 256>>>(dr, 1234u + rep, 4096, 64, 170, 64, 170, -200, 200, 0);
    run("random_traversal_range", dr);
    for (int rep = 0; rep < 4; rep++) k_random<<<8192, 256>>>(dr, 99u + rep, 4096, 64, 170, 64, 170, -200, 200, 1);
    run("near_boundary_traversal_range", dr);
    // the whole precondition: |d| in [2^-125, 2^125], |n| >= 2^-101, quotient exponent in [-124, 125]
    for (int rep = 0; rep < 4; rep++) k_random<<<8192, 256>>>(dr, 777u + rep, 4096, 26, 254, 2, 252, -124, 125, 0);
    run("random_full_precondition", dr);
    for (int rep = 0; rep < 4; rep++) k_random<<<8192, 256>>>(dr, 4242u + rep, 4096, 26, 254, 2, 252, -124, 125, 1);
    run("near_boundary_full_precondition", dr);
    // outside: tiny numerators / denormal quotients (documents that the precondition is needed)
    k_random<<<8192, 256>>>(dr, 5u, 1024, 1, 25, 2, 252, -124, 125, 0);   run("numerators_below_2^-101(expected to differ)", dr);
    k_random<<<8192, 256>>>(dr, 6u, 1024, 26, 254, 2, 252, -160, -126, 0); run("denormal_quotients(expected to differ)", dr);
    k_special<<<1, 144>>>(dr);                                 run("special_operands_fixup", dr, true);
    printf("}\n");
    return 0;
}
