// gfx950 (MI355X / CDNA4) wavefront kernels for the per-pixel NEE path-tracing hot path.
//
//   generate -> [ extend -> shade -> connect ] x max_bounces -> accumulate
//
// The arithmetic of every stage follows the reference (file:line cited per function, paths relative to
// the reference repository) operation by operation: IEEE float/double +,-,*,/ and sqrt only, compiled
// with -ffp-contract=off and correctly-rounded division, so that a path's radiance is bit-identical to
// the CPU restatement in stream mode (oracle/pt_oracle.c; the oracle is never linked here).  The two
// differences from the reference's sequential program are by construction and documented in DESIGN.md:
// random numbers are hash(pixel, sample, dimension) instead of one global mt19937, and the four
// transcendental call sites use the short polynomials below instead of glibc's libm.
//
// Wave64 design notes:
//  * Traversal is a lock-step *sweep* of a flattened traversal program (pt_device.h DOp): every lane of a
//    wave executes the same op, so node boxes, instance matrices and primitive records are wave-uniform
//    and arrive through scalar loads into SGPRs; lanes that missed an enclosing box are masked off until
//    the op where that subtree ends.  The per-lane short stack of partial results lives in LDS,
//    [slot][lane] so that consecutive lanes hit consecutive banks.
//  * Queues are segmented (pt_device.h): compaction = one wave ballot + popcount prefix and ONE atomicAdd per wave on the
//    output segment's counter (lane 0, the base broadcast through an SGPR; no barrier, no LDS).  Queue order is therefore
//    not deterministic; per-path arithmetic does not depend on it, and k_accumulate adds a pixel's samples in sample order.
//  * All streams are float4 / float2 planes indexed by queue position: each wave-level load or store is
//    one fully coalesced 1 KiB / 512 B transaction.
#ifdef __HIPCC_RTC__
// hiprtc (the per-scene build of the sweep, pt_spec.cpp) has the HIP device builtins and fixed-width integers, but no C headers
#ifndef FLT_MAX
#define FLT_MAX 3.40282347e+38F
#endif
#ifndef INFINITY
#define INFINITY __builtin_huge_valf()
#endif
#ifndef NAN
#define NAN __builtin_nanf("")
#endif
#else
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#endif

#include "pt_device.h"
#include "pt_fdiv.h"
#ifndef __HIPCC_RTC__
#include "pt_spec.h"
#endif

namespace ptd {

// Per-scene build of the sweep (pt_context.cpp pt_spec_source; compiled with hiprtc at pt_create): PT_SPEC_HEADER names a
// header that holds the scene's fast program as a compile-time table -- PT_SPEC_N ops, kSpecW[PT_SPEC_N][32] = the DOp
// words -- so that world_hit_fast unrolls into straight-line code: op fetch, decode and dispatch fold away, every leaf
// keeps only the body of its own kind and transform shape.  The arithmetic per ray is the generic sweep's, statement
// for statement (same functions, same operands), so the result is the same bit for bit.
#ifdef PT_SPEC_HEADER
#include PT_SPEC_HEADER
#ifndef PT_SPEC_REINV
#define PT_SPEC_REINV 0   // measurement knob: form 1 / direction again for the parent-box check instead of keeping it in registers
#endif
#endif

#define PT_BLOCK 256
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 6     // waves per SIMD k_shade is compiled for (80 VGPRs)
#endif
#ifndef PT_SHADE_WAVES_TEX
#define PT_SHADE_WAVES_TEX 5   // the instantiations that evaluate textures: 96 VGPRs, 0 - 16 of them spilled instead of 36 - 41 (+1 % on textured_room; 4 waves, no spills: -2 %)
#endif
#ifndef PT_CONNECT_WAVES
#define PT_CONNECT_WAVES 6   // k_connect with two rays per sweep
#endif
#ifndef PT_CONNECT_WAVES_GA
#define PT_CONNECT_WAVES_GA 4   // the same with sphere / constant_medium leaves (108 VGPRs; at 96 it spills 12, at 80 19: with_volume 24.9 -> 25.5 Grays/s
                                // against 5 waves at the end of round 2, three_orbs / light_test equal)
#endif
#ifndef PT_CONNECT_PREFETCH
#define PT_CONNECT_PREFETCH 1   // k_connect: 1 = the radiance is requested as soon as the slot is known; 2 = and the next group's shadow
                                // records before the current group is traced (12 more VGPRs at two rays per sweep); 0 = neither
#endif
#ifndef PT_SKIP_NOOP_RADIANCE
#define PT_SKIP_NOOP_RADIANCE 1   // radiance updates that cannot change a bit are not performed (0: the A/B)
#endif
#define PT_LIGHT_ONE(P, F, k_) rect_light_sample(PlaneTag<P>{}, BoolTag<F>{}, (k_), base + NV + (k_) * (3u + NV), tl, lq.x0, lq.z0, lq.x1, lq.z1, lq.y)
#define PT_LIGHT_LOOP(P, F) for (uint32_t k = 0; k < L; k++) PT_LIGHT_ONE(P, F, k);
#ifndef PT_SHADE_FDIV
#define PT_SHADE_FDIV 1      // k_shade's rect-light sample loop divides on pt_fdiv.h's form behind range checks (0: IEEE sequences, the A/B)
#endif
#ifndef PT_PK_DIV
#define PT_PK_DIV 1          // world_hit_fast_rb: the two sides of a box axis divide on packed FP32 (0: one quotient at a time)
#endif
#ifndef PT_SYM_BOUNDS
#define PT_SYM_BOUNDS 1      // per-scene build: faces centred on the local origin test |xh| - x1 instead of two differences (0: the A/B)
#endif
#ifndef PT_CULL_B0
#define PT_CULL_B0 1           // per-scene build, k_extend of bounce 0: a wave of camera rays none of which can touch ANY box / medium leaf runs the program
                               // without those leaves (wave_skips_cullable); 0: the A/B.  Measured for the other launches too and not used there (below)
#endif
#ifndef PT_VOL_SHARED
#define PT_VOL_SHARED 1      // fast sweep, constant_medium on a box: both boundary queries from one evaluation of the six sides (0: box_hit_fast twice, the A/B)
#endif
#ifndef PT_VOL_LAZY_DRAW
#define PT_VOL_LAZY_DRAW 1   // ... and the free-flight draw only in waves where some ray crosses the medium (0: every wave draws, the A/B)
#endif
#ifndef PT_CONNECT_NOHOIST
#define PT_CONNECT_NOHOIST 0   // k_connect: 1 = do not keep the per-leaf origin terms of a hit across its groups of rays (the A/B)
#endif
#ifndef PT_SHADOW_WALLS
#define PT_SHADOW_WALLS 1   // per-scene k_connect: the rects that wall the scene in are proven unreachable instead of tested (0: the A/B)
#endif
#ifndef PT_FAST_RB
#define PT_FAST_RB 1         // scenes of rects and boxes: the fast sweep with box faces in the global fold (world_hit_fast_rb); 0: world_hit_fast
#endif
#ifndef PT_FUSE_GENERATE
#define PT_FUSE_GENERATE 1   // bounce 0 forms its camera rays itself, no k_generate launch (0: k_generate writes them, as before)
#endif
#define PT_PI_D 3.14159265358979323846
#define PT_PI_F 3.14159274f

// ------------------------------------------------------------------------------------------------
// vec3 (reference vec3.h:11-199); same association as the C++ operators
// ------------------------------------------------------------------------------------------------
struct v3 { float x, y, z; };
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define DEVI __device__ __forceinline__
DEVI v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
DEVI v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
DEVI v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
DEVI v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
DEVI v3 vscale(float t, v3 v) { return V(t * v.x, t * v.y, t * v.z); }
DEVI v3 vdivf(v3 v, float t) { return V(v.x / t, v.y / t, v.z / t); }
DEVI v3 vneg(v3 v) { return V(-v.x, -v.y, -v.z); }
DEVI float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEVI v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEVI float vsqlen(v3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
DEVI float vlen(v3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
DEVI v3 vunit(v3 v) { return vdivf(v, vlen(v)); }
// Stream stores and loads.  A record is written once and read once by a later kernel, after gigabytes of other traffic: marked
// non-temporal (the `nt` bit of global_load / global_store) it does not stay in the L2 behind its one use.  PT_NT_STORES: 1
// k_shade's continuation and shadow records, 2 k_extend's hit records and the bounce-0 radiance; PT_NT_LOADS: 1 k_shade's inputs,
// 2 k_extend's rays, 4 k_connect's shadow records (bit masks; 0 / 0 is the A/B).  Measured: k_shade's record stores alone
// +2.4 % (k_shade 15.6 -> 14.6 ms per 64 spp), everything +3.1 %.
#ifndef PT_NT_STORES
#define PT_NT_STORES 3
#endif
#ifndef PT_NT_LOADS
#define PT_NT_LOADS 7
#endif
typedef float pt_f4 __attribute__((ext_vector_type(4)));
typedef float pt_f2 __attribute__((ext_vector_type(2)));
template <int BIT> DEVI void st4(float4 *p, float4 v)
{
    if (PT_NT_STORES & BIT) { pt_f4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<pt_f4 *>(p)); }
    else *p = v;
}
template <int BIT> DEVI void st2(float2 *p, float2 v)
{
    if (PT_NT_STORES & BIT) { pt_f2 w = {v.x, v.y}; __builtin_nontemporal_store(w, reinterpret_cast<pt_f2 *>(p)); }
    else *p = v;
}
template <int BIT> DEVI float4 ld4(const float4 *p)
{
    if (PT_NT_LOADS & BIT) { const pt_f4 w = __builtin_nontemporal_load(reinterpret_cast<const pt_f4 *>(p)); return make_float4(w.x, w.y, w.z, w.w); }
    return *p;
}
template <int BIT> DEVI float2 ld2(const float2 *p)
{
    if (PT_NT_LOADS & BIT) { const pt_f2 w = __builtin_nontemporal_load(reinterpret_cast<const pt_f2 *>(p)); return make_float2(w.x, w.y); }
    return *p;
}
DEVI bool is_nanf(float x) { return !(x == x); }
DEVI bool v_is_nan(v3 v) { return is_nanf(v.x) || is_nanf(v.y) || is_nanf(v.z); }

// ------------------------------------------------------------------------------------------------
// stream RNG + portable math (definition shared with the CPU restatement's stream mode)
// ------------------------------------------------------------------------------------------------
DEVI uint32_t mix_lowbias32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
DEVI uint32_t stream_u32(uint32_t k0, uint32_t k1, uint32_t dim)
{
    uint32_t x = k0 + dim * 0x9E3779B9u;
    x ^= x >> 17; x *= 0xed5ad4bbU;
    x ^= k1;
    x ^= x >> 11; x *= 0xac4c1b51U;
    x ^= x >> 15; x *= 0x31848babU;
    x ^= x >> 14;
    return x;
}
DEVI double rnd(uint32_t k0, uint32_t k1, uint32_t dim) { return (double)stream_u32(k0, k1, dim) * (1.0 / 4294967296.0); }
// float(rnd(...)): the 32-bit integer is rounded to 24 bits exactly once on either route and the scaling by 2^-32
// is exact, so this equals the double route bit for bit without touching the f64 pipe
DEVI float rndf(uint32_t k0, uint32_t k1, uint32_t dim) { return (float)stream_u32(k0, k1, dim) * 2.3283064365386963e-10f; }

DEVI void ptm_sincos_2pi(float r, float &s, float &c)
{
    float t = r * 4.0f;
    float q = floorf(t + 0.5f);
    float f = t - q;
    float a = f * 1.57079637f;
    float a2 = a * a;
    float sp = a + a * a2 * (-0.16666667f + a2 * (0.0083333310f + a2 * (-0.00019840874f + a2 * 2.7525562e-06f)));
    float cp = 1.0f + a2 * (-0.5f + a2 * (0.041666638f + a2 * (-0.0013888378f + a2 * 2.4760495e-05f)));
    int qi = ((int)q) & 3;
    if (qi == 0) { s = sp; c = cp; }
    else if (qi == 1) { s = cp; c = -sp; }
    else if (qi == 2) { s = -sp; c = -cp; }
    else { s = -cp; c = sp; }
}
DEVI float ptm_cbrtf(float x)
{
    if (!(x > 0.0f)) return 0.0f;
    uint32_t u = __float_as_uint(x);
    u = u / 3u + 709921077u;
    float y = __uint_as_float(u);
#pragma unroll
    for (int i = 0; i < 3; i++) y = y - (y * y * y - x) / (3.0f * y * y);
    return y;
}
DEVI float ptm_logf(float x)
{
    if (!(x > 0.0f)) return -INFINITY;
    uint32_t u = __float_as_uint(x);
    int e = 0;
    if (u < 0x00800000u) { u = __float_as_uint(x * 8388608.0f); e = -23; }
    e += (int)(u >> 23) - 127;
    u = (u & 0x007fffffu) | 0x3f800000u;
    float m = __uint_as_float(u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float p = 2.0f * s * (1.0f + z * (0.33333334f + z * (0.2f + z * (0.14285715f + z * 0.11111111f))));
    float fe = (float)e;
    return fe * 0.693145752f + (fe * 1.42860677e-06f + p);
}

// sinf / atan2f / acosf of the texture row (checker_texture::sines texture.h:70-73, environment coordinates
// integrator.h:327-330): double +,-,*,/ and rint in a fixed order, the same statements as oracle/pt_oracle.c
DEVI double ptm_sin_reduced(double r, int q)
{
    const double r2 = r * r;
    const double sp = r + r * r2 * (-1.6666666666666666e-01 + r2 * (8.3333333333333332e-03 + r2 * (-1.9841269841269841e-04
                      + r2 * (2.7557319223985893e-06 + r2 * (-2.5052108385441720e-08 + r2 * 1.6059043836821613e-10)))));
    const double cp = 1.0 + r2 * (-0.5 + r2 * (4.1666666666666664e-02 + r2 * (-1.3888888888888889e-03 + r2 * (2.4801587301587302e-05
                      + r2 * (-2.7557319223985888e-07 + r2 * (2.0876756987868100e-09 + r2 * -1.1470745597729725e-11))))));
    return (q & 1) ? ((q & 2) ? -cp : cp) : ((q & 2) ? -sp : sp);
}
DEVI float ptm_sinf(float x)
{
    if (!(fabsf(x) < 1073741824.0f)) return x - x;
    const double xd = (double)x;
    const double k = rint(xd * 0.63661977236758138);
    const double r = ((xd - k * 1.5707963267341256) - k * 6.077100506303966e-11) - k * 2.0222662487959506e-21;
    return (float)ptm_sin_reduced(r, (int)k & 3);
}
DEVI double ptm_atan2_d(double y, double x)
{
    if (x != x || y != y) return x + y;
    const double ax = fabs(x), ay = fabs(y);
    const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    double a = (mx > 0.0) ? mn / mx : 0.0;
    double off = 0.0;
    if (a > 0.41421356237309503) { a = (a - 1.0) / (a + 1.0); off = 0.78539816339744828; }
    const double z = a * a;
    double p = 1.0 / 27.0;
    p = 1.0 / 25.0 - z * p; p = 1.0 / 23.0 - z * p; p = 1.0 / 21.0 - z * p; p = 1.0 / 19.0 - z * p; p = 1.0 / 17.0 - z * p;
    p = 1.0 / 15.0 - z * p; p = 1.0 / 13.0 - z * p; p = 1.0 / 11.0 - z * p; p = 1.0 / 9.0 - z * p; p = 1.0 / 7.0 - z * p;
    p = 1.0 / 5.0 - z * p; p = 1.0 / 3.0 - z * p; p = 1.0 - z * p;
    double t = off + a * p;
    if (ay > ax) t = 1.5707963267948966 - t;
    if (x < 0.0) t = 3.1415926535897931 - t;
    return (y < 0.0) ? -t : t;
}
DEVI float ptm_atan2f(float y, float x) { return (float)ptm_atan2_d((double)y, (double)x); }
DEVI float ptm_acosf(float x)
{
    const double xd = (double)x;
    return (float)ptm_atan2_d(sqrt((1.0 - xd) * (1.0 + xd)), xd);
}
// float -> int as the reference's x86 build converts (cvttss2si): truncation, INT_MIN for NaN and out-of-range values
DEVI int f2i_x86(float f) { return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (int)0x80000000; }

// stream-mode dimension layout (must match oracle/pt_oracle.c "stream-mode dimension layout")
#define DIM_JITTER_U 0u
#define DIM_JITTER_V 1u
#define DIM_LENS 2u
#define DIM_BOUNCE0 8u
DEVI uint32_t dims_per_bounce(const DScene &S) { return (uint32_t)S.n_vol + (uint32_t)S.light_samples * (3u + (uint32_t)S.n_vol) + 4u; }

// ------------------------------------------------------------------------------------------------
// transform3 application (reference transform3.h:56-68 through Eigen 3.2.10: t + ((a0*b0 + a1*b1) + a2*b2))
// ------------------------------------------------------------------------------------------------
DEVI v3 xf_point(const float *m, v3 p)
{
    return V(m[3] + ((m[0] * p.x + m[1] * p.y) + m[2] * p.z),
             m[7] + ((m[4] * p.x + m[5] * p.y) + m[6] * p.z),
             m[11] + ((m[8] * p.x + m[9] * p.y) + m[10] * p.z));
}
DEVI v3 xf_linear(const float *m, v3 v)
{
    return V((m[0] * v.x + m[1] * v.y) + m[2] * v.z,
             (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
             (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}
DEVI v3 xf_normal(const float *inv, v3 n)
{   // normalize((L^-1)^T n), Eigen's norm association x2 + (y2 + z2)   (transform3.h:60-63)
    float x = (inv[0] * n.x + inv[4] * n.y) + inv[8] * n.z;
    float y = (inv[1] * n.x + inv[5] * n.y) + inv[9] * n.z;
    float z = (inv[2] * n.x + inv[6] * n.y) + inv[10] * n.z;
    float nrm = sqrtf(x * x + (y * y + z * z));
    return V(x / nrm, y / nrm, z / nrm);
}

// ------------------------------------------------------------------------------------------------
// primitives (reference primitive.h, volume.h) -- "t only" versions for traversal; the winner's
// hit_record is rebuilt afterwards by finalize_hit()
// ------------------------------------------------------------------------------------------------
DEVI v3 shuffle(v3 v, int plane)
{   // primitive.h:104-121
    if (plane == 0) return V(v.x, v.z, v.y);
    if (plane == 2) return V(v.y, v.x, v.z);
    return v;
}
DEVI bool rect_hit_t(const DRect &q, v3 A, v3 B, float t0, float t1, float &t_out)
{   // primitive.h:186-206
    v3 o = shuffle(A, q.plane);
    v3 d = shuffle(B, q.plane);
    float t = (q.y - o.y) / d.y;
    if (t < t0 || t > t1) return false;
    float xh = o.x + t * d.x;
    float zh = o.z + t * d.z;
    if (xh < q.x0 || xh > q.x1 || zh < q.z0 || zh > q.z1) return false;
    t_out = t;
    return true;
}
DEVI v3 rect_normal(const DRect &q, v3 B)
{   // primitive.h:212-222 (two_sided is always true)
    v3 n = shuffle(V(0.0f, q.ny, 0.0f), q.plane);
    if (vdot(B, n) > 0) n = vneg(n);
    return n;
}
DEVI bool sphere_hit_t(const DPrim &s, v3 A, v3 B, float t_min, float t_max, float &t_out)
{   // primitive.h:64-95
    v3 oc = vsub(A, V(s.cx, s.cy, s.cz));
    float a = vdot(B, B);
    float b = vdot(oc, B);
    float c = vdot(oc, oc) - s.radius * s.radius;
    float disc = b * b - a * c;
    if (disc > 0) {
        float temp = (-b - sqrtf(disc)) / a;
        if (temp < t_max && temp > t_min) { t_out = temp; return true; }
        temp = (-b + sqrtf(disc)) / a;
        if (temp < t_max && temp > t_min) { t_out = temp; return true; }
    }
    return false;
}
// sphere::hit primitive.h:64-95 on the local ray, oc = origin - center, c = dot(oc, oc) - radius^2
DEVI bool sphere_t(v3 oc, float c, v3 Bl, float t_min, float t_max, float &t_out)
{
    const float a = vdot(Bl, Bl);
    const float b = vdot(oc, Bl);
    const float disc = b * b - a * c;
    if (disc > 0) {
        float temp = (-b - sqrtf(disc)) / a;
        if (temp < t_max && temp > t_min) { t_out = temp; return true; }
        temp = (-b + sqrtf(disc)) / a;
        if (temp < t_max && temp > t_min) { t_out = temp; return true; }
    }
    return false;
}
// constant_medium::hit volume.h:29-93 once both boundary hits are known (hit1 / hit2 with t1v, t2v): clamp, free-flight
// distance from one draw, inside test.  Returns the hit and its t.
DEVI bool medium_decide(bool hit, float t1v, float t2v, v3 Bl, float density, float u, float &t_out)
{
    const float T_MIN = 0.001f, T_MAX = FLT_MAX;
    t1v = (t1v < T_MIN) ? T_MIN : t1v;
    t2v = (t2v > T_MAX) ? T_MAX : t2v;
    hit = hit && !(t1v >= t2v);
    t1v = (t1v < 0) ? 0.0f : t1v;
    const float dlen = vlen(Bl);
    const float distance_inside = (t2v - t1v) * dlen;
    const float hit_distance = (-(1 / density)) * ptm_logf(u);
    t_out = t1v + hit_distance / dlen;
    return hit && (hit_distance < distance_inside);
}
// constant_medium::hit (volume.h:29-93) called with the range (t_min, t_max): the clamp, the draw u of this call, the inside test.
// medium_decide is this with the integrator's range; a medium that is the BOUNDARY of another medium is called with
// (-FLT_MAX, FLT_MAX) and (rec1.t + 0.0001, FLT_MAX) (volume.h:38-40).
DEVI bool medium_decide_in(bool hit, float t1v, float t2v, float t_min, float t_max, v3 Bl, float density, float u, float &t_out)
{
    t1v = (t1v < t_min) ? t_min : t1v;
    t2v = (t2v > t_max) ? t_max : t2v;
    hit = hit && !(t1v >= t2v);
    t1v = (t1v < 0) ? 0.0f : t1v;
    const float dlen = vlen(Bl);
    const float distance_inside = (t2v - t1v) * dlen;
    const float hit_distance = (-(1 / density)) * ptm_logf(u);
    t_out = t1v + hit_distance / dlen;
    return hit && (hit_distance < distance_inside);
}
// ---- leaf tests on NR rays that share one origin (NR = 1: extension ray; NR = light_samples: the shadow rays of
// one hit).  The local origin (ray::apply ray.h:20-24) and every numerator that depends only on it are computed
// once; each ray's own arithmetic is exactly the single-ray sequence.
//
// All per-lane decisions are kept in the VECTOR unit: a CU has one scalar ALU for its four SIMDs, and the first
// version of this sweep spent 0.83 scalar instructions per vector instruction on exec-mask branches and on and/or of
// compare results -- the scalar unit, not the VALU, set the pace.  The reference's chains of comparisons are
// therefore restated as one "excess" value e whose sign decides, with identical NaN behaviour:
//     (a < lo || a > hi)   <=>   fmaxf(lo - a, a - hi) > 0
// for every float a: a - b > 0 <=> a > b exactly (subnormals are kept, so a != b never subtracts to 0), infinities give
// +-inf with the right sign, and a NaN a makes both differences NaN, which fmaxf ignores -- exactly like the
// reference's comparisons, which are all false for NaN so that a NaN t or hit coordinate is NOT rejected.
template <int PLANE>   // 0 XY, 1 XZ, 2 YZ: which local component is the plane axis / x / z (primitive.h:104-121)
DEVI void rect_axes(v3 v, float &x, float &pl, float &z)
{
    if (PLANE == 0) { x = v.x; pl = v.z; z = v.y; }
    else if (PLANE == 2) { x = v.y; pl = v.x; z = v.z; }
    else { x = v.x; pl = v.y; z = v.z; }
}
// rect::hit primitive.h:186-206 with the shuffle resolved at compile time; num = y - o.y.
// Returns the excess: the reference rejects the hit iff excess > 0.  t_out is always the quotient.
template <int PLANE>
DEVI float rect_excess(float x0, float z0, float x1, float z1, float num, float ox, float oz, v3 Bl, float t0, float t1, float &t_out)
{
    float dx, dpl, dz;
    rect_axes<PLANE>(Bl, dx, dpl, dz);
    const float t = num / dpl;
    const float xh = ox + t * dx;
    const float zh = oz + t * dz;
    t_out = t;
    const float et = fmaxf(t0 - t, t - t1);                                   // t < t0 || t > t1
    const float ex = fmaxf(x0 - xh, xh - x1), ez = fmaxf(z0 - zh, zh - z1);   // xh < x0 || xh > x1 || zh < z0 || zh > z1
    return fmaxf(et, fmaxf(ex, ez));
}
// box::hit -> hittable_list::hit over the six sides in primitive.h:232-240 order; closest_so_far shrinks and a later
// side with an equal t replaces (hittable_list.h:27-35).  face = -1: no side hit.
DEVI void box_hit_shared(const float *p0, const float *p1, v3 Al, v3 Bl, float t0, float t1, float &t_out, int &face)
{
    float closest = t1, t;
    int f = -1;
    bool h;
    h = !(rect_excess<0>(p0[0], p0[1], p1[0], p1[1], p0[2] - Al.z, Al.x, Al.y, Bl, t0, closest, t) > 0.0f); closest = h ? t : closest; f = h ? 0 : f;
    h = !(rect_excess<0>(p0[0], p0[1], p1[0], p1[1], p1[2] - Al.z, Al.x, Al.y, Bl, t0, closest, t) > 0.0f); closest = h ? t : closest; f = h ? 1 : f;
    h = !(rect_excess<2>(p0[1], p0[2], p1[1], p1[2], p0[0] - Al.x, Al.y, Al.z, Bl, t0, closest, t) > 0.0f); closest = h ? t : closest; f = h ? 2 : f;
    h = !(rect_excess<2>(p0[1], p0[2], p1[1], p1[2], p1[0] - Al.x, Al.y, Al.z, Bl, t0, closest, t) > 0.0f); closest = h ? t : closest; f = h ? 3 : f;
    h = !(rect_excess<1>(p0[0], p0[2], p1[0], p1[2], p0[1] - Al.y, Al.x, Al.z, Bl, t0, closest, t) > 0.0f); closest = h ? t : closest; f = h ? 4 : f;
    h = !(rect_excess<1>(p0[0], p0[2], p1[0], p1[2], p1[1] - Al.y, Al.x, Al.z, Bl, t0, closest, t) > 0.0f); closest = h ? t : closest; f = h ? 5 : f;
    t_out = closest;
    face = f;
}

// ---- the same two leaf tests for the FAST sweep (world_hit_fast): every operand is finite and inside the precondition
// of pt_fdiv.h, so the quotient is fdiv_q -- the bits of num / dpl -- over r = fdiv_rcp(dpl), one refined reciprocal
// shared by the numerators over one denominator.
template <int PLANE>
DEVI float rect_excess_fast(float x0, float z0, float x1, float z1, float num, float ox, float oz, v3 Bl, float t0, float t1, float r, float &t_out)
{
    float dx, dpl, dz;
    rect_axes<PLANE>(Bl, dx, dpl, dz);
    const float t = fdiv_q(num, dpl, r);
    const float xh = ox + t * dx;
    const float zh = oz + t * dz;
    t_out = t;
    const float et = fmaxf(t0 - t, t - t1);
    const float ex = fmaxf(x0 - xh, xh - x1), ez = fmaxf(z0 - zh, zh - z1);
    return fmaxf(et, fmaxf(ex, ez));
}
DEVI void box_hit_fast(const float *p0, const float *p1, v3 Al, v3 Bl, float t0, float t1, float &t_out, int &face)
{
    float closest = t1, t;
    int f = -1;
    bool h;
    const float rz = fdiv_rcp(Bl.z);   // the two sides of an axis divide by the same local direction component
    h = !(rect_excess_fast<0>(p0[0], p0[1], p1[0], p1[1], p0[2] - Al.z, Al.x, Al.y, Bl, t0, closest, rz, t) > 0.0f); closest = h ? t : closest; f = h ? 0 : f;
    h = !(rect_excess_fast<0>(p0[0], p0[1], p1[0], p1[1], p1[2] - Al.z, Al.x, Al.y, Bl, t0, closest, rz, t) > 0.0f); closest = h ? t : closest; f = h ? 1 : f;
    const float rx = fdiv_rcp(Bl.x);
    h = !(rect_excess_fast<2>(p0[1], p0[2], p1[1], p1[2], p0[0] - Al.x, Al.y, Al.z, Bl, t0, closest, rx, t) > 0.0f); closest = h ? t : closest; f = h ? 2 : f;
    h = !(rect_excess_fast<2>(p0[1], p0[2], p1[1], p1[2], p1[0] - Al.x, Al.y, Al.z, Bl, t0, closest, rx, t) > 0.0f); closest = h ? t : closest; f = h ? 3 : f;
    const float ry = fdiv_rcp(Bl.y);
    h = !(rect_excess_fast<1>(p0[0], p0[2], p1[0], p1[2], p0[1] - Al.y, Al.x, Al.z, Bl, t0, closest, ry, t) > 0.0f); closest = h ? t : closest; f = h ? 4 : f;
    h = !(rect_excess_fast<1>(p0[0], p0[2], p1[0], p1[2], p1[1] - Al.y, Al.x, Al.z, Bl, t0, closest, ry, t) > 0.0f); closest = h ? t : closest; f = h ? 5 : f;
    t_out = closest;
    face = f;
}

// The general sweep's short stack of partial results, [slot][ray][thread]: in LDS (stride PT_BLOCK) while the tree needs at
// most PT_MAX_STACK slots, else in a global scratch area of the launching lane (DStreams::gstack; stride = threads of
// the largest grid).  One pointer type for both: the few waves that take the general sweep use flat loads / stores.
struct Stk { float2 *p; int stride; };
DEVI Stk stack_of(const DStreams &st, float2 *lds)
{
    Stk k;
    if (st.gstack) { k.p = st.gstack + ((size_t)blockIdx.x * PT_BLOCK + threadIdx.x); k.stride = st.gstack_stride; }
    else { k.p = lds + threadIdx.x; k.stride = PT_BLOCK; }
    return k;
}
// ------------------------------------------------------------------------------------------------
// World::hit (world.h:17-20 -> bvh.h:31-69 -> primitive.h:298-312) as a lock-step sweep over NR rays per lane
// that share the origin A.  Returns per ray id = -1 (miss) or instance*8 + face, and t.  `stk` points at this
// lane's column of the LDS short stack, laid out [slot][ray][PT_BLOCK] float2.
//
// Lane masking without branches.  skip[r] is the pc at which ray r becomes active again after missing a box
// (0 = active).  Invariants used below: (i) a masked ray has cur_id = -1 (set when it missed the box) and keeps it;
// (ii) its skip is the end of an ENCLOSING subtree, hence >= the end op_a of any node met while masked, so
// skip = max(skip, miss ? op_a : 0) leaves masked rays alone; (iii) stack slots written inside a skipped subtree are
// dead for that ray, so pushes need no mask.
// ------------------------------------------------------------------------------------------------
// GA ("all geometry"): the scene has sphere or constant_medium leaves.  Scenes of rects and boxes only (all BASELINE
// Cornell boxes but the volume one) run the GA = false instantiations, which carry neither the code nor the registers of
// those leaves.
//
// This is the GENERAL sweep: IEEE divisions, the tree of COMBINEs on the LDS short stack (NaN t and exact-t ties resolve
// by the tree's shape like bvh_node::hit), finite or non-finite operands.  Waves whose rays are all tame take
// world_hit_fast below instead (world_hit picks).
template <int NR, bool GA>
DEVI void world_hit_n(const DScene &S, bool lane_valid, v3 A, const v3 (&B)[NR], uint32_t k0, uint32_t k1,
                      const uint32_t (&vol_dim_base)[NR], Stk stk, float (&out_t)[NR], int (&out_id)[NR], bool all_finite)
{
    const float T_MIN = 0.001f, T_MAX = FLT_MAX;   // integrator.h:193,246
    v3 inv[NR];
    float cur_t[NR];
    int cur_id[NR], skip[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        inv[r] = V(1.0f / B[r].x, 1.0f / B[r].y, 1.0f / B[r].z);   // aabb.h:38, same value at every node
        cur_t[r] = 0.0f;
        cur_id[r] = -1;
        skip[r] = lane_valid ? 0 : 0x7fffffff;
    }
    const int n_ops = S.n_ops;
    // An op is two 64-byte scalar loads whose addresses depend on pc alone; the second half (primitive parameters) is only
    // waited for after the ray has been transformed.  Prefetching the next op's first half was tried and removed: carrying
    // it across the loop edge costs eight s_mov_b64 per op on the scalar unit, more than the latency the other waves hide
    // anyway (k_extend -32 % without it).
    for (int pc = 0; pc < n_ops; ++pc) {
        const i32x16 w0 = *reinterpret_cast<const i32x16 *>(&S.ops[pc]);
        const i32x16 w1 = *(reinterpret_cast<const i32x16 *>(&S.ops[pc]) + 1);
        const int kind = w0[0], op_a = w0[1], op_slot = w0[2], op_push = w0[3];
        const int op_id_base = op_a * 8;
#define OPF(i) __int_as_float((i) < 12 ? w0[4 + (i)] : w1[(i) - 12])
        if (op_push >= 0) {
#pragma unroll
            for (int r = 0; r < NR; r++) stk.p[(size_t)(op_push * NR + r) * stk.stride] = make_float2(cur_t[r], __int_as_float(cur_id[r]));
        }
        if (kind == OP_ENTER) {
            // aabb::hit aabb.h:34-53; (min - origin), (max - origin) are shared by the NR rays.
            // "t0 > tmin ? t0 : tmin" keeps tmin when t0 is NaN = fmaxf(tmin, t0); likewise fminf for tmax; they are never
            // NaN themselves.  The reference returns false as soon as tmax <= tmin after an axis; tmin only grows and tmax
            // only shrinks from axis to axis, so that is the case iff it holds after the last axis.
            const float dx0 = OPF(0) - A.x, dy0 = OPF(1) - A.y, dz0 = OPF(2) - A.z;
            const float dx1 = OPF(3) - A.x, dy1 = OPF(4) - A.y, dz1 = OPF(5) - A.z;
            bool any_in = false;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                float tmin = T_MIN, tmax = T_MAX;
                {
                    const float a = dx0 * inv[r].x, c = dx1 * inv[r].x;
                    const bool neg = inv[r].x < 0.0f;
                    tmin = fmaxf(tmin, neg ? c : a); tmax = fminf(tmax, neg ? a : c);
                }
                {
                    const float a = dy0 * inv[r].y, c = dy1 * inv[r].y;
                    const bool neg = inv[r].y < 0.0f;
                    tmin = fmaxf(tmin, neg ? c : a); tmax = fminf(tmax, neg ? a : c);
                }
                {
                    const float a = dz0 * inv[r].z, c = dz1 * inv[r].z;
                    const bool neg = inv[r].z < 0.0f;
                    tmin = fmaxf(tmin, neg ? c : a); tmax = fminf(tmax, neg ? a : c);
                }
                const bool miss = tmax <= tmin;
                cur_id[r] = miss ? -1 : cur_id[r];            // masked rays already hold -1
                skip[r] = max(skip[r], miss ? op_a : 0);      // masked rays: skip >= op_a already
                any_in |= (pc >= skip[r]);
            }
            // no ray of the wave is inside this subtree any more: jump to its end -- wave-uniform, one ballot
            if (!__any(any_in)) pc = op_a - 1;
        } else if (kind == OP_COMBINE) {   // bvh.h:36-66: left iff left.hit && (!right.hit || left.t < right.t)
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const float2 l = stk.p[(size_t)(op_slot * NR + r) * stk.stride];
                const int lid = (pc >= skip[r]) ? __float_as_int(l.y) : -1;   // a masked ray keeps its -1
                const int lt = (l.x < cur_t[r]) ? -1 : 0;
                const int take = (~(lid >> 31)) & ((cur_id[r] >> 31) | lt);    // all-ones iff take left
                cur_t[r] = take ? l.x : cur_t[r];
                cur_id[r] = take ? lid : cur_id[r];
            }
        } else {
            // instance::hit primitive.h:298-312: local origin once, local direction per ray
            const float m[12] = {OPF(0), OPF(1), OPF(2), OPF(3), OPF(4), OPF(5), OPF(6), OPF(7), OPF(8), OPF(9), OPF(10), OPF(11)};
            const float q0[3] = {OPF(12), OPF(13), OPF(14)}, q1[3] = {OPF(15), OPF(16), OPF(17)};
            // Pure translation (op_slot = 1): for FINITE ray components t + ((1*x + 0*y) + 0*z) == t + x and
            // (1*x + 0*y) + 0*z == x; only the sign of a zero component can differ, which no comparison below can see
            // (+-inf t is rejected either way, 0/0 is NaN either way).  A wave holding any non-finite ray (0*inf = NaN
            // would spread across components) takes the general path.
            const bool ident = ((op_slot & 15) == 1) && all_finite;
            const v3 Al = ident ? V(m[3] + A.x, m[7] + A.y, m[11] + A.z) : xf_point(m, A);
#define XF_DIR(b) (ident ? (b) : xf_linear(m, (b)))
            if (kind <= OP_LEAF_RECT_YZ) {   // the three rect alignments first: the most frequent leaf (rect::hit primitive.h:186-225)
                float ox, opl, oz;
                if (kind == OP_LEAF_RECT_XY) rect_axes<0>(Al, ox, opl, oz);
                else if (kind == OP_LEAF_RECT_YZ) rect_axes<2>(Al, ox, opl, oz);
                else rect_axes<1>(Al, ox, opl, oz);
                const float num = q1[1] - opl;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const v3 Bl = XF_DIR(B[r]);
                    float t, e;
                    if (kind == OP_LEAF_RECT_XY) e = rect_excess<0>(q0[0], q0[1], q0[2], q1[0], num, ox, oz, Bl, T_MIN, T_MAX, t);
                    else if (kind == OP_LEAF_RECT_YZ) e = rect_excess<2>(q0[0], q0[1], q0[2], q1[0], num, ox, oz, Bl, T_MIN, T_MAX, t);
                    else e = rect_excess<1>(q0[0], q0[1], q0[2], q1[0], num, ox, oz, Bl, T_MIN, T_MAX, t);
                    const bool hit = !(e > 0.0f) && (pc >= skip[r]);
                    cur_id[r] = hit ? op_id_base : -1;
                    cur_t[r] = hit ? t : cur_t[r];
                }
            } else if (kind == OP_LEAF_BOX) {
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const v3 Bl = XF_DIR(B[r]);
                    float t;
                    int face;
                    box_hit_shared(q0, q1, Al, Bl, T_MIN, T_MAX, t, face);
                    const bool hit = (face >= 0) && (pc >= skip[r]);
                    cur_id[r] = hit ? (op_id_base + face) : -1;
                    cur_t[r] = hit ? t : cur_t[r];
                }
            } else if (GA && kind == OP_LEAF_VOLBOX) {   // constant_medium::hit volume.h:29-93 with a box boundary
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const v3 Bl = XF_DIR(B[r]);
                    float t1v, t2v;
                    int f1, f2;
                    box_hit_shared(q0, q1, Al, Bl, -FLT_MAX, FLT_MAX, t1v, f1);
                    box_hit_shared(q0, q1, Al, Bl, (float)((double)t1v + 0.0001), FLT_MAX, t2v, f2);
                    bool hit = (f1 >= 0) && (f2 >= 0);
                    t1v = (t1v < T_MIN) ? T_MIN : t1v;
                    t2v = (t2v > T_MAX) ? T_MAX : t2v;
                    hit = hit && !(t1v >= t2v);
                    t1v = (t1v < 0) ? 0.0f : t1v;
                    const float dlen = vlen(Bl);
                    const float distance_inside = (t2v - t1v) * dlen;
                    const float u = rndf(k0, k1, vol_dim_base[r] + (uint32_t)w1[7]);
                    const float hit_distance = (-(1 / OPF(18))) * ptm_logf(u);
                    hit = hit && (hit_distance < distance_inside) && (pc >= skip[r]);
                    cur_id[r] = hit ? op_id_base : -1;
                    cur_t[r] = hit ? (t1v + hit_distance / dlen) : cur_t[r];
                }
            } else if (GA && kind == OP_LEAF_SPHERE) {   // sphere::hit primitive.h:64-95
                const v3 oc = vsub(Al, V(q0[0], q0[1], q0[2]));
                const float c = vdot(oc, oc) - q1[0] * q1[0];
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const v3 Bl = XF_DIR(B[r]);
                    const float a = vdot(Bl, Bl);
                    const float b = vdot(oc, Bl);
                    const float disc = b * b - a * c;
                    const float ta = (-b - sqrtf(disc)) / a, tb = (-b + sqrtf(disc)) / a;
                    const bool ha = (ta < T_MAX) && (ta > T_MIN), hb = (tb < T_MAX) && (tb > T_MIN);
                    const bool hit = (disc > 0) && (ha || hb) && (pc >= skip[r]);
                    cur_id[r] = hit ? op_id_base : -1;
                    cur_t[r] = hit ? (ha ? ta : tb) : cur_t[r];
                }
            } else if (GA && kind == OP_LEAF_VOLSPHERE) {   // constant_medium::hit volume.h:29-93 with a sphere boundary
                const v3 oc = vsub(Al, V(q0[0], q0[1], q0[2]));
                const float c = vdot(oc, oc) - q1[0] * q1[0];
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const v3 Bl = XF_DIR(B[r]);
                    float t1v = 0.0f, t2v = 0.0f, tv;
                    bool hit = sphere_t(oc, c, Bl, -FLT_MAX, FLT_MAX, t1v);
                    hit = hit && sphere_t(oc, c, Bl, (float)((double)t1v + 0.0001), FLT_MAX, t2v);
                    const float u = rndf(k0, k1, vol_dim_base[r] + (uint32_t)w1[7]);
                    hit = medium_decide(hit, t1v, t2v, Bl, OPF(18), u, tv) && (pc >= skip[r]);
                    cur_id[r] = hit ? op_id_base : -1;
                    cur_t[r] = hit ? tv : cur_t[r];
                }
            } else if (GA && kind == OP_LEAF_VOLVOL) {
                // constant_medium whose boundary is a constant_medium (volume.h:10: any hittable).  boundary->hit(r, -FLT_MAX,
                // FLT_MAX, rec1) and boundary->hit(r, rec1.t + 0.0001, FLT_MAX, rec2) are two scattering events of the INNER medium
                // (each: its own boundary hit twice -- the same two values both times -- its clamp to the caller's range, a draw
                // of its own), then the outer medium's clamp, draw and inside test.  The three draws of a traversal have
                // consecutive dimensions in the reference's order (pt_context.cpp volume_draws).
                const bool ibox = w1[9] == 1;
                const float dens_out = OPF(18), dens_in = OPF(20);
                const v3 oc = vsub(Al, V(q0[0], q0[1], q0[2]));
                const float cs = vdot(oc, oc) - q1[0] * q1[0];
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const v3 Bl = XF_DIR(B[r]);
                    float b1 = 0.0f, b2 = 0.0f;   // the innermost boundary's two hits: the same for both calls of the inner medium
                    bool bh;
                    if (ibox) {
                        int f1, f2;
                        box_hit_shared(q0, q1, Al, Bl, -FLT_MAX, FLT_MAX, b1, f1);
                        box_hit_shared(q0, q1, Al, Bl, (float)((double)b1 + 0.0001), FLT_MAX, b2, f2);
                        bh = (f1 >= 0) && (f2 >= 0);
                    } else {
                        bh = sphere_t(oc, cs, Bl, -FLT_MAX, FLT_MAX, b1);
                        bh = bh && sphere_t(oc, cs, Bl, (float)((double)b1 + 0.0001), FLT_MAX, b2);
                    }
                    const uint32_t d0 = vol_dim_base[r] + (uint32_t)w1[7];
                    float t1v = 0.0f, t2v = 0.0f, tv;
                    bool hit = medium_decide_in(bh, b1, b2, -FLT_MAX, FLT_MAX, Bl, dens_in, rndf(k0, k1, d0), t1v);
                    hit = hit && medium_decide_in(bh, b1, b2, (float)((double)t1v + 0.0001), FLT_MAX, Bl, dens_in, rndf(k0, k1, d0 + 1u), t2v);
                    hit = medium_decide(hit, t1v, t2v, Bl, dens_out, rndf(k0, k1, d0 + 2u), tv) && (pc >= skip[r]);
                    cur_id[r] = hit ? op_id_base : -1;
                    cur_t[r] = hit ? tv : cur_t[r];
                }
            } else {   // OP_LEAF_NONE (and kinds this instantiation does not carry): the leaf reports a miss
#pragma unroll
                for (int r = 0; r < NR; r++) cur_id[r] = -1;
            }
        }
#undef XF_DIR
#undef OPF
    }
#pragma unroll
    for (int r = 0; r < NR; r++) { out_t[r] = cur_t[r]; out_id[r] = cur_id[r]; }
}

// ------------------------------------------------------------------------------------------------
// The FAST sweep: the same World::hit for waves whose rays are all TAME (world_hit checks): every component of A and of
// the B[r] is finite, the B components are non-zero and within [2^-20, 2^20], the A components zero or within
// [2^-20, 2^20], and the program's leaf data are in the ranges DScene::tame stands for (linear part zero or
// [2^-44, 2^13] per entry, translation zero or [2^-44, 2^20], bounds zero or [2^-20, 2^20]; checked on the host).
// It returns the same (t, id) as world_hit_n bit for bit, with fewer instructions:
//
//  * Divisions are the unscaled exact sequence of pt_fdiv.h.  Precondition, by the ulp argument (a sum or difference of
//    floats that are zero or at least m in magnitude is zero or at least ulp(m)): a product m*x is zero or within
//    [2^-64, 2^33]; a local direction component (m0 x + m1 y) + m2 z is zero or within [2^-87, 2^35]; a local origin
//    component t + (..) zero or within [2^-87, 2^36]; a numerator plane - o zero or within [2^-87, 2^37]; so every
//    quotient is zero or within [2^-122, 2^124] and no residual underflows (|n| >= 2^-101).
//  * Node boxes (aabb::hit): with every B component finite and non-zero no slab product is NaN (0 * inf cannot occur),
//    so the reference's "swap if invD < 0" is min / max of the two products (multiplication by one finite inv is
//    monotone under rounding and node boxes are ordered, min <= max), and "tmax <= tmin after some axis" is the test
//    after the last axis (tmin only grows, tmax only shrinks).
//  * Transforms: DOp::slot = 2 / 3 / 4 marks an inverse whose linear part maps the x / y / z axis to itself (row and
//    column of that axis zero off the diagonal: every rotation about one axis, with any scaling) -- the products by
//    those exact zeros are skipped; for finite operands that changes at most the sign of a zero, which no comparison
//    of a leaf can see (the argument of the pure-translation shortcut, slot = 1).
//  * The tree of COMBINEs is a left fold over the leaves in program order: "left iff left.hit && (!right.hit ||
//    left.t < right.t)" is associative as long as every accepted t is ordered (closest hit, the later leaf on equal t),
//    so the short stack in LDS and the COMBINE ops are not needed.  A leaf accepted with a NaN t (0 / 0: a ray inside the
//    plane of a rect through its origin) makes the result depend on the tree's shape: the lane reports it (return
//    value) and the wave repeats the query with world_hit_n.
// ------------------------------------------------------------------------------------------------
template <int AXIS>
DEVI v3 xf_axis_linear(const float *m, v3 v)
{   // the axis component is mapped to itself; the other two mix
    if (AXIS == 0) return V(m[0] * v.x, m[5] * v.y + m[6] * v.z, m[9] * v.y + m[10] * v.z);
    if (AXIS == 1) return V(m[0] * v.x + m[2] * v.z, m[5] * v.y, m[8] * v.x + m[10] * v.z);
    return V(m[0] * v.x + m[1] * v.y, m[4] * v.x + m[5] * v.y, m[10] * v.z);
}
// local origin and the NR local directions for the shape `pat` of the inverse (one scalar branch chain per op)
template <int NR>
DEVI void xf_pat(int pat, const float *m, v3 A, const v3 (&B)[NR], v3 &Al, v3 (&Bl)[NR])
{
    if (pat == 1) {
        Al = V(m[3] + A.x, m[7] + A.y, m[11] + A.z);
#pragma unroll
        for (int r = 0; r < NR; r++) Bl[r] = B[r];
    } else if (pat == 2) {
        const v3 l = xf_axis_linear<0>(m, A);
        Al = V(m[3] + l.x, m[7] + l.y, m[11] + l.z);
#pragma unroll
        for (int r = 0; r < NR; r++) Bl[r] = xf_axis_linear<0>(m, B[r]);
    } else if (pat == 3) {
        const v3 l = xf_axis_linear<1>(m, A);
        Al = V(m[3] + l.x, m[7] + l.y, m[11] + l.z);
#pragma unroll
        for (int r = 0; r < NR; r++) Bl[r] = xf_axis_linear<1>(m, B[r]);
    } else if (pat == 4) {
        const v3 l = xf_axis_linear<2>(m, A);
        Al = V(m[3] + l.x, m[7] + l.y, m[11] + l.z);
#pragma unroll
        for (int r = 0; r < NR; r++) Bl[r] = xf_axis_linear<2>(m, B[r]);
    } else {
        Al = xf_point(m, A);
#pragma unroll
        for (int r = 0; r < NR; r++) Bl[r] = xf_linear(m, B[r]);
    }
}
// The fast program (DScene::ops + ops_fast_off .. + n_ops_fast): the same op list without the COMBINE ops, ENTER's `a`
// re-targeted.  All per-lane decisions stay in the vector unit (see the note on the scalar unit above): the state of a
// ray is (cur_t, cur_id, skipf) with cur_t = FLT_MAX while nothing is hit -- a first hit has t <= FLT_MAX and replaces
// it like the tie rule does -- and a leaf's verdict is the sign of ONE maximum:
//     take  <=>  !( max3(excess, t - cur_t, skipf - pc) > 0 )
// excess > 0: the leaf rejects the ray; t - cur_t > 0: the current hit is strictly closer (NaN when t is NaN: ignored,
// like the reference's "left.t < right.t" is false); skipf - pc > 0: the ray is masked until op skipf.
// `chk` turns NaN for good as soon as an accepted t is NaN (0 * NaN), which is what the lane reports.
template <int NR, bool GA>
DEVI bool world_hit_fast(const DScene &S, bool lane_valid, v3 A, const v3 (&B)[NR], uint32_t k0, uint32_t k1,
                         const uint32_t (&vol_dim_base)[NR], float (&out_t)[NR], int (&out_id)[NR])
{
    const float T_MIN = 0.001f, T_MAX = FLT_MAX;   // integrator.h:193,246
    v3 inv[NR];
    float cur_t[NR], skipf[NR], chk = 0.0f;
    int cur_id[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        inv[r] = V(fdiv(1.0f, B[r].x), fdiv(1.0f, B[r].y), fdiv(1.0f, B[r].z));   // aabb.h:38
        cur_t[r] = FLT_MAX;
        cur_id[r] = -1;
        skipf[r] = lane_valid ? 0.0f : 1e9f;
    }
#ifdef PT_SPEC_HEADER
#define OPW(i) kSpecW[pc][(i)]
#pragma unroll
    for (int pc = 0; pc < PT_SPEC_N; ++pc) {
#else
    const DOp *__restrict__ prog = S.ops + S.ops_fast_off;
    const int n_ops = S.n_ops_fast;
#define OPW(i) ((i) < 16 ? w0[(i)] : w1[(i) - 16])
    for (int pc = 0; pc < n_ops; ++pc) {
        const i32x16 w0 = *reinterpret_cast<const i32x16 *>(&prog[pc]);
        const i32x16 w1 = *(reinterpret_cast<const i32x16 *>(&prog[pc]) + 1);
#endif
        const int kind = OPW(0), op_a = OPW(1), pat = OPW(2) & 15;
        const bool op_ieee = (OPW(2) & 16) != 0;    // this leaf's data are outside the unscaled division's precondition
        const int op_id_base = op_a * 8;
        const float pcf = __int_as_float(OPW(3));   // (float)pc, stored by the host (pt_context.cpp)
#define OPF(i) __int_as_float(OPW(4 + (i)))
        if (kind == OP_ENTER) {
            const float dx0 = OPF(0) - A.x, dy0 = OPF(1) - A.y, dz0 = OPF(2) - A.z;
            const float dx1 = OPF(3) - A.x, dy1 = OPF(4) - A.y, dz1 = OPF(5) - A.z;
            const float endf = OPF(6);             // (float)op_a
            float in_max = 1.0f;   // min over the rays of (skipf - pc): <= 0 iff some ray is still inside
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const float ax = dx0 * inv[r].x, cx = dx1 * inv[r].x;
                const float ay = dy0 * inv[r].y, cy = dy1 * inv[r].y;
                const float az = dz0 * inv[r].z, cz = dz1 * inv[r].z;
                const float tmin = fmaxf(fmaxf(fmaxf(fminf(ax, cx), fminf(ay, cy)), fminf(az, cz)), T_MIN);
                const float tmax = fminf(fminf(fminf(fmaxf(ax, cx), fmaxf(ay, cy)), fmaxf(az, cz)), T_MAX);
                // max of two non-negative floats, taken on their bit patterns (same order, no NaN-quieting instruction)
                skipf[r] = __int_as_float(max(__float_as_int(skipf[r]), (tmax <= tmin) ? __float_as_int(endf) : 0));
                in_max = fminf(in_max, skipf[r] - pcf);
            }
#ifdef PT_SPEC_HEADER
            // straight-line code has no program counter to set: a subtree that no ray of the wave is inside is left by a
            // forward branch when it is the rest of the program (the root's box), else its ops run masked (skipf)
            if (op_a >= PT_SPEC_N && !__any(in_max <= 0.0f)) break;
#else
            if (!__any(in_max <= 0.0f)) pc = op_a - 1;   // no ray of the wave is inside this subtree: jump to its end
#endif
            continue;
        }
        const float m[12] = {OPF(0), OPF(1), OPF(2), OPF(3), OPF(4), OPF(5), OPF(6), OPF(7), OPF(8), OPF(9), OPF(10), OPF(11)};
        const float q0[3] = {OPF(12), OPF(13), OPF(14)}, q1[3] = {OPF(15), OPF(16), OPF(17)};
        v3 Al, Bl[NR];
        xf_pat<NR>(pat, m, A, B, Al, Bl);
#define FOLD(r, e_, t_, id_)                                                                              \
        {                                                                                                 \
            const bool take_ = !(fmaxf(fmaxf((e_), (t_) - cur_t[r]), skipf[r] - pcf) > 0.0f);             \
            cur_t[r] = take_ ? (t_) : cur_t[r];                                                           \
            cur_id[r] = take_ ? (id_) : cur_id[r];                                                        \
            chk = __builtin_fmaf(0.0f, cur_t[r], chk);                                                    \
        }
        if (kind <= OP_LEAF_RECT_YZ) {
            // one of three bodies per wave (kind is wave-uniform).  The empty volatile asm keeps hipcc from if-converting the
            // three into "compute all three quotients, select by kind" -- it did: 85 instead of 33 vector instructions
            float t[NR], e[NR];
#define RECT_BODY(PLANE, DPL)                                                                                              \
            {                                                                                                                \
                float ox, opl, oz;                                                                                           \
                rect_axes<PLANE>(Al, ox, opl, oz);                                                                           \
                const float num = q1[1] - opl;                                                                               \
                _Pragma("unroll") for (int r = 0; r < NR; r++)                                                               \
                    e[r] = rect_excess_fast<PLANE>(q0[0], q0[1], q0[2], q1[0], num, ox, oz, Bl[r], T_MIN, T_MAX, fdiv_rcp(Bl[r].DPL), t[r]); \
            }
#define RECT_BODY_IEEE(PLANE)                                                                                          \
            {                                                                                                                \
                float ox, opl, oz;                                                                                           \
                rect_axes<PLANE>(Al, ox, opl, oz);                                                                           \
                const float num = q1[1] - opl;                                                                               \
                _Pragma("unroll") for (int r = 0; r < NR; r++)                                                               \
                    e[r] = rect_excess<PLANE>(q0[0], q0[1], q0[2], q1[0], num, ox, oz, Bl[r], T_MIN, T_MAX, t[r]);             \
            }
            if (op_ieee) {
                if (kind == OP_LEAF_RECT_XY) { asm volatile("; rect xy ieee"); RECT_BODY_IEEE(0) }
                else if (kind == OP_LEAF_RECT_YZ) { asm volatile("; rect yz ieee"); RECT_BODY_IEEE(2) }
                else { asm volatile("; rect xz ieee"); RECT_BODY_IEEE(1) }
            }
            else if (kind == OP_LEAF_RECT_XY) { asm volatile("; rect xy"); RECT_BODY(0, z) }
            else if (kind == OP_LEAF_RECT_YZ) { asm volatile("; rect yz"); RECT_BODY(2, x) }
            else { asm volatile("; rect xz"); RECT_BODY(1, y) }
#undef RECT_BODY_IEEE
#undef RECT_BODY
#pragma unroll
            for (int r = 0; r < NR; r++) FOLD(r, e[r], t[r], op_id_base)
        } else if (kind == OP_LEAF_BOX) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                float t;
                int face;
                // two bodies, one per wave (op_ieee is wave-uniform): the empty asm keeps hipcc from computing both and selecting
                if (op_ieee) { asm volatile("; box ieee"); box_hit_shared(q0, q1, Al, Bl[r], T_MIN, T_MAX, t, face); }
                else { asm volatile("; box fast"); box_hit_fast(q0, q1, Al, Bl[r], T_MIN, T_MAX, t, face); }
                FOLD(r, (face >= 0) ? 0.0f : 1.0f, t, op_id_base + face)
            }
        } else if (GA && kind == OP_LEAF_VOLBOX) {   // constant_medium::hit volume.h:29-93 with a box boundary
#pragma unroll
            for (int r = 0; r < NR; r++) {
                float t1v, t2v;
                int f1, f2;
                if (op_ieee) {
                    asm volatile("; volbox ieee");
                    box_hit_shared(q0, q1, Al, Bl[r], -FLT_MAX, FLT_MAX, t1v, f1);
                    box_hit_shared(q0, q1, Al, Bl[r], (float)((double)t1v + 0.0001), FLT_MAX, t2v, f2);
                } else {
                    asm volatile("; volbox fast");
                    box_hit_fast(q0, q1, Al, Bl[r], -FLT_MAX, FLT_MAX, t1v, f1);
                    box_hit_fast(q0, q1, Al, Bl[r], (float)((double)t1v + 0.0001), FLT_MAX, t2v, f2);
                }
                bool hit = (f1 >= 0) && (f2 >= 0);
                chk = __builtin_fmaf(0.0f, t1v, chk);   // a NaN boundary t: let the general sweep decide
                chk = __builtin_fmaf(0.0f, t2v, chk);
                t1v = (t1v < T_MIN) ? T_MIN : t1v;
                t2v = (t2v > T_MAX) ? T_MAX : t2v;
                hit = hit && !(t1v >= t2v);
                t1v = (t1v < 0) ? 0.0f : t1v;
                const float dlen = vlen(Bl[r]);
                const float distance_inside = (t2v - t1v) * dlen;
                const float u = rndf(k0, k1, vol_dim_base[r] + (uint32_t)OPW(23));
                const float hit_distance = (-(1 / OPF(18))) * ptm_logf(u);
                hit = hit && (hit_distance < distance_inside);
                const float tv = t1v + hit_distance / dlen;
                FOLD(r, hit ? 0.0f : 1.0f, tv, op_id_base)
            }
        } else if (GA && kind == OP_LEAF_SPHERE) {   // sphere::hit primitive.h:64-95 (IEEE divisions: a = |Bl|^2 may leave the precondition)
            const v3 oc = vsub(Al, V(q0[0], q0[1], q0[2]));
            const float c = vdot(oc, oc) - q1[0] * q1[0];
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const float a = vdot(Bl[r], Bl[r]);
                const float b = vdot(oc, Bl[r]);
                const float disc = b * b - a * c;
                const float ta = (-b - sqrtf(disc)) / a, tb = (-b + sqrtf(disc)) / a;
                const bool ha = (ta < T_MAX) && (ta > T_MIN), hb = (tb < T_MAX) && (tb > T_MIN);
                const float ts = ha ? ta : tb;
                FOLD(r, ((disc > 0) && (ha || hb)) ? 0.0f : 1.0f, ts, op_id_base)
            }
        } else if (GA && kind == OP_LEAF_VOLSPHERE) {   // constant_medium::hit volume.h:29-93 with a sphere boundary
            const v3 oc = vsub(Al, V(q0[0], q0[1], q0[2]));
            const float c = vdot(oc, oc) - q1[0] * q1[0];
#pragma unroll
            for (int r = 0; r < NR; r++) {
                float t1v = 0.0f, t2v = 0.0f, tv;
                bool hit = sphere_t(oc, c, Bl[r], -FLT_MAX, FLT_MAX, t1v);
                hit = hit && sphere_t(oc, c, Bl[r], (float)((double)t1v + 0.0001), FLT_MAX, t2v);
                chk = __builtin_fmaf(0.0f, t1v, chk);
                chk = __builtin_fmaf(0.0f, t2v, chk);
                const float u = rndf(k0, k1, vol_dim_base[r] + (uint32_t)OPW(23));
                hit = medium_decide(hit, t1v, t2v, Bl[r], OPF(18), u, tv);
                FOLD(r, hit ? 0.0f : 1.0f, tv, op_id_base)
            }
        }
#undef FOLD
#undef OPF
#undef OPW
    }
    if (S.n_chain) {
        // flat program (pt_device.h DScene::chains): below the root no node box was tested on the way.  bvh_node::hit would
        // have dropped a leaf whose ancestor box the ray misses; the closest leaf over ALL leaves is the tree's answer iff
        // its own ancestors are hit (it is then also the closest of the leaves the tree reaches, ties by order included),
        // and its parent's box stands for all of them.  So only that one box is tested -- per lane, the slab arithmetic of
        // the ENTER ops -- and a lane whose winner fails it sends the wave to the general sweep.
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const float4 *ch = S.chains + (size_t)max(cur_id[r] >> 3, 0) * 2;
            const float4 lo = ch[0], hi = ch[1];
#if defined(PT_SPEC_HEADER) && PT_SPEC_REINV
            // the straight-line sweep is short of registers, not of issue slots: 1 / direction is formed again here (the same
            // three quotients) instead of living in registers across every leaf; the empty asm keeps hipcc from reusing the first
            float bx = B[r].x, by = B[r].y, bz = B[r].z;
            asm volatile("" : "+v"(bx), "+v"(by), "+v"(bz));
            const v3 inv_r = V(fdiv(1.0f, bx), fdiv(1.0f, by), fdiv(1.0f, bz));
#else
            const v3 inv_r = inv[r];
#endif
            const float ax = (lo.x - A.x) * inv_r.x, cx = (hi.x - A.x) * inv_r.x;
            const float ay = (lo.y - A.y) * inv_r.y, cy = (hi.y - A.y) * inv_r.y;
            const float az = (lo.z - A.z) * inv_r.z, cz = (hi.z - A.z) * inv_r.z;
            const float tmin = fmaxf(fmaxf(fmaxf(fminf(ax, cx), fminf(ay, cy)), fminf(az, cz)), T_MIN);
            const float tmax = fminf(fminf(fminf(fmaxf(ax, cx), fmaxf(ay, cy)), fmaxf(az, cz)), T_MAX);
            chk = ((tmax <= tmin) && cur_id[r] >= 0) ? NAN : chk;
        }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) { out_t[r] = (cur_id[r] >= 0) ? cur_t[r] : 0.0f; out_id[r] = cur_id[r]; }
    return is_nanf(chk);
}

// ------------------------------------------------------------------------------------------------
// The fast sweep, round 4 (rect and box leaves restated; sphere and constant_medium leaves as in world_hit_fast).  Same
// program, same preconditions and the same (t, id) as world_hit_fast above, bit for bit, with fewer and cheaper
// instructions per leaf face -- the sweep is bound by vector-instruction issue, and of the ~78 issue cycles a face cost
// (20 two-cycle FP32 operations, 8 four-cycle compares / selects / maxima) these go:
//
//  * Box faces join the GLOBAL fold.  box::hit is hittable_list::hit over six rects with a shrinking closest_so_far
//    (primitive.h:232-256, hittable_list.h:21-38), and bvh_node::hit then keeps the box's result iff it is closer than the
//    current one (bvh.h:36-66).  Both levels apply one rule -- the smallest t wins, the LATER candidate on equal t -- and that
//    rule picks the same element whether the sequence is folded in one level or in two (all accepted t are ordered: no NaN
//    reaches the fold, see below).  So a box face is folded like a rect leaf with id = instance * 8 + face, against the
//    ray's current (cur_t, cur_id) instead of a per-box (closest, face) pair followed by a fold of the box: six selects, a
//    compare and a maximum less per box and ray.
//  * No v_div_fixup, no NaN tracker.  v_div_fixup only patches zero / infinite / NaN operands and quotients outside the
//    normal range; with a finite numerator, a finite NON-ZERO denominator and the ranges of pt_fdiv.h's precondition it
//    returns its first operand (for a zero numerator the quotient is a zero whose sign cannot matter: t = +-0 < t_min is
//    rejected).  World-space direction components are non-zero (tame); a LOCAL direction component of a rotated leaf can
//    cancel to exactly zero: its hardware reciprocal is then infinite and the refined reciprocal NaN, and 0 * that reciprocal
//    turns the wave's tracker NaN -- one two-cycle operation per RECIPROCAL of a rotated leaf (not per quotient) -- and the wave
//    repeats the query with the general sweep, which is also what the old tracker did with the only NaN a leaf can produce
//    (0 / 0).  Without zero denominators no t is NaN or infinite.
//  * No "t > t_max" term: t_max is FLT_MAX (integrator.h:193, 246) and t is finite.
//  * No mask term in flat programs: a ray that missed the root's box (or an idle lane) starts with cur_t = -inf, and
//    "t - cur_t > 0" then rejects every face; tree-shaped programs (more than PT_FLAT_MAX_INSTANCES instances) keep the
//    skip position of world_hit_fast.
// A face then costs 16 two-cycle operations and 6 four-cycle ones (~61 cycles).  Leaves whose data are outside the
// unscaled division's precondition (DOp::slot bit 4) keep their IEEE divisions and the NaN tracker, like before.
// ------------------------------------------------------------------------------------------------
DEVI float fdiv_q_nofix(float n, float d, float r)
{   // fdiv_q (pt_fdiv.h) without the final v_div_fixup: the same quotient for finite n, finite non-zero d inside the precondition
    float q = n * r;
    float rem = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(rem, r, q);
}
// One rect face of a leaf (a rect, or one side of a box) against ray state (cur_t, cur_id): rect::hit primitive.h:186-206 with
// the closest-hit rule folded in.  PLANE as in rect_axes; num = plane - local origin's plane component; r = fdiv_rcp(dpl);
// skip <= 0: the ray is not masked (tree programs pass skipf - pc, flat programs a constant).
// the closest-hit rule on one rect face given its quotient t (see face_fold)
// rect::hit's bounds test on a hit point (xh, zh) (primitive.h:200-205): positive iff "xh < x0 || xh > x1 || zh < z0 || zh > z1";
// NaN coordinates compare false there and are ignored by the maxima here.
DEVI float side_excess(float x0, float z0, float x1, float z1, float xh, float zh)
{
    float ex, ez;
#if defined(PT_SPEC_HEADER) && PT_SYM_BOUNDS
    // Bounds symmetric about the local origin (x0 == -x1: every rect and box the scene format centres, scene_parser.h:131-170;
    // a compile-time fact of the per-scene build): max(x0 - xh, xh - x1) and |xh| - x1 have the same sign for every xh -- for
    // xh >= 0 the first term -(x1 + xh) is not positive and the second IS |xh| - x1, for xh < 0 the roles swap; NaN stays NaN
    // (ignored by the maximum either way) -- and only the sign of the maximum is used.  One subtraction per axis instead of two.
    if (x0 == -x1) ex = fabsf(xh) - x1; else ex = fmaxf(x0 - xh, xh - x1);
    if (z0 == -z1) ez = fabsf(zh) - z1; else ez = fmaxf(z0 - zh, zh - z1);
#else
    ex = fmaxf(x0 - xh, xh - x1); ez = fmaxf(z0 - zh, zh - z1);
#endif
    return fmaxf(ex, ez);
}
template <int PLANE, bool IEEE, bool SKIP>
DEVI void face_fold_t(float x0, float z0, float x1, float z1, float t, float ox, float oz, v3 Bl, float skip, int id,
                      float &cur_t, int &cur_id, float &chk)
{
    const float T_MIN = 0.001f;
    float dx, dpl, dz;
    rect_axes<PLANE>(Bl, dx, dpl, dz);
    const float xh = ox + t * dx;
    const float zh = oz + t * dz;
    // reject iff t < t_min || xh < x0 || xh > x1 || zh < z0 || zh > z1 (primitive.h:193-205; NaN compares false: not rejected) or the
    // current hit is strictly closer -- one maximum, decided by its sign
    float e = fmaxf(fmaxf(T_MIN - t, t - cur_t), side_excess(x0, z0, x1, z1, xh, zh));
    if (SKIP) e = fmaxf(e, skip);
    const bool take = !(e > 0.0f);
    cur_t = take ? t : cur_t;
    cur_id = take ? id : cur_id;
    // an accepted NaN t (0 / 0): the general sweep decides.  Only the ACCEPTED t feeds the tracker: cur_t itself is -inf for a lane
    // that missed the root box of a flat program, and 0 * -inf would send every wave at the scene's silhouette to the general sweep
    if (IEEE) chk = __builtin_fmaf(0.0f, take ? t : 0.0f, chk);
}
template <int PLANE, bool IEEE, bool SKIP>
DEVI void face_fold(float x0, float z0, float x1, float z1, float num, float ox, float oz, v3 Bl, float r, float skip, int id,
                    float &cur_t, int &cur_id, float &chk)
{
    float dx, dpl, dz;
    rect_axes<PLANE>(Bl, dx, dpl, dz);
    const float t = IEEE ? num / dpl : fdiv_q_nofix(num, dpl, r);
    face_fold_t<PLANE, IEEE, SKIP>(x0, z0, x1, z1, t, ox, oz, Bl, skip, id, cur_t, cur_id, chk);
}
// The two sides of a box axis divide their numerators by ONE denominator: both quotients at once on packed FP32
// (v_pk_mul_f32 / v_pk_fma_f32: two individually rounded IEEE operations per instruction -- a plain v_fma_f32 occupies the SIMD as
// long as the packed one, tools/microbench/valu_rates.hip -- the scalars d and r enter through op_sel, no moves).  Per
// component it is fdiv_q_nofix statement for statement.
typedef float pt_pk2 __attribute__((ext_vector_type(2)));
DEVI void fdiv_q2_nofix_dd(float n0, float n1, float d0, float d1, float r0, float r1, float &q0, float &q1)
{   // two independent quotients n0 / d0, n1 / d1 side by side (fdiv_q_nofix per component)
#if PT_PK_DIV
    pt_pk2 n; n.x = n0; n.y = n1;
    pt_pk2 dd; dd.x = d0; dd.y = d1;
    pt_pk2 rr; rr.x = r0; rr.y = r1;
    pt_pk2 q = n * rr;
    pt_pk2 rem = __builtin_elementwise_fma(-dd, q, n);
    q = __builtin_elementwise_fma(rem, rr, q);
    rem = __builtin_elementwise_fma(-dd, q, n);
    q = __builtin_elementwise_fma(rem, rr, q);
    q0 = q.x; q1 = q.y;
#else
    q0 = fdiv_q_nofix(n0, d0, r0); q1 = fdiv_q_nofix(n1, d1, r1);
#endif
}
DEVI void fdiv_q2_nofix(float n0, float n1, float d, float r, float &q0, float &q1)
{
#if PT_PK_DIV
    pt_pk2 n; n.x = n0; n.y = n1;
    pt_pk2 dd; dd.x = d; dd.y = d;
    pt_pk2 rr; rr.x = r; rr.y = r;
    pt_pk2 q = n * rr;
    pt_pk2 rem = __builtin_elementwise_fma(-dd, q, n);
    q = __builtin_elementwise_fma(rem, rr, q);
    rem = __builtin_elementwise_fma(-dd, q, n);
    q = __builtin_elementwise_fma(rem, rr, q);
    q0 = q.x; q1 = q.y;
#else
    q0 = fdiv_q_nofix(n0, d, r); q1 = fdiv_q_nofix(n1, d, r);
#endif
}
// Exact divisions of k_shade's light-sample loop without the IEEE scaffolding (PT_SHADE_FDIV).  hipcc expands every n / d into
// v_div_scale x2, v_rcp, four fma, a mul, v_div_fmas, v_div_fixup (~47 issue cycles) and shares NOTHING between quotients over
// one denominator (v_div_scale looks at both operands) -- nine such sequences per light sample, half the sample's cycles.
// pt_fdiv.h's division (refined reciprocal shared by the numerators of one denominator, two residual corrections, v_div_fixup
// for zero / infinite / NaN operands) returns the same correctly rounded quotient whenever no intermediate leaves the normal
// range; two quotients over one denominator run on packed FP32.  The loop therefore runs on it and CHECKS the ranges that
// make it exact as it goes (gmin / gmax in k_shade, per lane); a wave in which any lane left them repeats the loop with the
// IEEE divisions.
DEVI void fdiv_q2(float n0, float n1, float d, float r, float &q0, float &q1)
{   // two fdiv_q (pt_fdiv.h) over one denominator, statement for statement per component
    pt_pk2 n; n.x = n0; n.y = n1;
    pt_pk2 dd; dd.x = d; dd.y = d;
    pt_pk2 rr; rr.x = r; rr.y = r;
    pt_pk2 q = n * rr;
    pt_pk2 rem = __builtin_elementwise_fma(-dd, q, n);
    q = __builtin_elementwise_fma(rem, rr, q);
    rem = __builtin_elementwise_fma(-dd, q, n);
    q = __builtin_elementwise_fma(rem, rr, q);
    q0 = __builtin_amdgcn_div_fixupf(q.x, d, n0);
    q1 = __builtin_amdgcn_div_fixupf(q.y, d, n1);
}
DEVI v3 vdivf_fast(v3 v, float d)
{
    const float r = fdiv_rcp(d);
    v3 o;
    fdiv_q2(v.x, v.y, d, r, o.x, o.y);
    o.z = fdiv_q(v.z, d, r);
    return o;
}
template <bool B> struct BoolTag { static constexpr bool value = B; };
// NOCULL (per-scene build): the instantiation for waves none of whose rays can touch a cullable leaf (wave_skips_cullable): the same
// straight-line program without those leaves.
//
// SHADOW (per-scene build, k_connect): the rays are shadow rays -- origin a hit point, direction = light sample - origin, so the
// sample sits at t = 1 -- and all that is asked of the closest hit is whether it is an emitter (connect_contribution).  A rect that
// WALLS THE SCENE IN (DOp::slot bit 5, set by the host: every instance on one side of its plane, the lights strictly inside, no
// emitter itself) can only be crossed before t_min or beyond the sample, and its exact test is replaced by the proof that it is:
// t~ = (plane - origin) * rcp(direction) outside [2^-11, 1 + 2^-12] (t~ is within 3 ulp of the quotient the reference forms, which is
// then < t_min = 0.001 and rejected, primitive.h:193, or > 1 + 2^-13).  If afterwards the closest hit has t <= 1 + 2^-14 it is closer
// than every such wall and stays the closest hit with them; no hit, or a non-emitter, stays "no contribution" whatever a wall behind
// it would answer.  A ray that fails either test (t~ inside the interval or NaN: a grazing ray on the wall's own plane; a hit beyond
// the sample) sends its wave to the general sweep like a NaN does.  The sides of a box that wall the scene in (bits 8-13, BOX_AXIS
// below) are proven the same way.  cornell_box: five of its six rects and the two blocks' bottoms, 374 vector instructions per traced
// shadow ray instead of 458, k_connect -13 % (profiles/experiments/r05_ab_runs.json r05_walls*).
template <int NR, bool GA, bool NOCULL = false, bool SHADOW = false>
DEVI bool world_hit_fast_rb(const DScene &S, bool lane_valid, v3 A, const v3 (&B)[NR], uint32_t k0, uint32_t k1,
                            const uint32_t (&vol_dim_base)[NR], float (&out_t)[NR], int (&out_id)[NR])
{
    const float T_MIN = 0.001f, T_MAX = FLT_MAX;   // integrator.h:193,246
    v3 inv[NR];
    float cur_t[NR], skipf[NR], chk = 0.0f;
    int cur_id[NR];
    bool beyond = false;   // SHADOW: some wall's plane may be crossed inside (t_min, sample], or the closest hit lies beyond the sample
    constexpr float kWallLo = 0x1p-11f, kWallHi = 1.0f + 0x1p-12f, kWallMid = 0.5f * (kWallLo + kWallHi), kWallHalf = 0.5f * (kWallHi - kWallLo);
#if defined(PT_SPEC_HEADER) && defined(PT_SPEC_NWALL)
    constexpr bool kBoxWalls = SHADOW && PT_SHADOW_WALLS != 0;
#else
    constexpr bool kBoxWalls = false;
#endif
#pragma unroll
    for (int r = 0; r < NR; r++) {
        {   // aabb.h:38: 1 / direction, exact; tame components are finite and non-zero: no fix-up, x and y as a packed pair
            const float rx = fdiv_rcp(B[r].x), ry = fdiv_rcp(B[r].y), rz = fdiv_rcp(B[r].z);
            fdiv_q2_nofix_dd(1.0f, 1.0f, B[r].x, B[r].y, rx, ry, inv[r].x, inv[r].y);
            inv[r].z = fdiv_q_nofix(1.0f, B[r].z, rz);
        }
        cur_t[r] = lane_valid ? FLT_MAX : -INFINITY;
        cur_id[r] = -1;
        skipf[r] = 0.0f;
    }
#ifdef PT_SPEC_HEADER
#define OPW(i) kSpecW[pc][(i)]
    constexpr bool kFlat = PT_SPEC_FLAT != 0;   // one ENTER op (the root's): no skip positions
#pragma unroll
    for (int pc = 0; pc < PT_SPEC_N; ++pc) {
#else
    constexpr bool kFlat = false;               // decided per scene at run time: the skip term stays (one subtraction per face)
    const DOp *__restrict__ prog = S.ops + S.ops_fast_off;
    const int n_ops = S.n_ops_fast;
#define OPW(i) ((i) < 16 ? w0[(i)] : w1[(i) - 16])
    for (int pc = 0; pc < n_ops; ++pc) {
        const i32x16 w0 = *reinterpret_cast<const i32x16 *>(&prog[pc]);
        const i32x16 w1 = *(reinterpret_cast<const i32x16 *>(&prog[pc]) + 1);
#endif
        const int kind = OPW(0), op_a = OPW(1), pat = OPW(2) & 15;
        const bool op_ieee = (OPW(2) & 16) != 0;    // this leaf's data are outside the unscaled division's precondition
        const int op_id_base = op_a * 8;
        const float pcf = __int_as_float(OPW(3));   // (float)pc, stored by the host (pt_context.cpp)
#define OPF(i) __int_as_float(OPW(4 + (i)))
        if (NOCULL && kind != OP_ENTER && (OPW(2) & 80) == 64) continue;   // (bit 6 without bit 4: a cullable leaf; this wave touches none of them)
        if (kind == OP_ENTER) {   // aabb::hit aabb.h:34-53, as in world_hit_fast
            const float dx0 = OPF(0) - A.x, dy0 = OPF(1) - A.y, dz0 = OPF(2) - A.z;
            const float dx1 = OPF(3) - A.x, dy1 = OPF(4) - A.y, dz1 = OPF(5) - A.z;
            const float endf = OPF(6);             // (float)op_a
            float in_max = 1.0f;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const float ax = dx0 * inv[r].x, cx = dx1 * inv[r].x;
                const float ay = dy0 * inv[r].y, cy = dy1 * inv[r].y;
                const float az = dz0 * inv[r].z, cz = dz1 * inv[r].z;
                const float tmin = fmaxf(fmaxf(fmaxf(fminf(ax, cx), fminf(ay, cy)), fminf(az, cz)), T_MIN);
                const float tmax = fminf(fminf(fminf(fmaxf(ax, cx), fmaxf(ay, cy)), fmaxf(az, cz)), T_MAX);
                const bool miss = tmax <= tmin;
                if (kFlat) {
                    // the root's box: a ray that misses it hits nothing -- cur_t = -inf makes "t - cur_t > 0" reject every face
                    cur_t[r] = miss ? -INFINITY : cur_t[r];
                    in_max = fminf(in_max, (cur_t[r] > 0.0f) ? 0.0f : 1.0f);
                } else {
                    skipf[r] = __int_as_float(max(__float_as_int(skipf[r]), miss ? __float_as_int(endf) : 0));
                    in_max = fminf(in_max, (lane_valid ? skipf[r] : 1e9f) - pcf);
                }
            }
#ifdef PT_SPEC_HEADER
            if (op_a >= PT_SPEC_N && !__any(in_max <= 0.0f)) break;
#else
            if (!__any(in_max <= 0.0f)) pc = op_a - 1;   // no ray of the wave is inside this subtree: jump to its end
#endif
            continue;
        }
        const float m[12] = {OPF(0), OPF(1), OPF(2), OPF(3), OPF(4), OPF(5), OPF(6), OPF(7), OPF(8), OPF(9), OPF(10), OPF(11)};
        const float q0[3] = {OPF(12), OPF(13), OPF(14)}, q1[3] = {OPF(15), OPF(16), OPF(17)};
        v3 Al, Bl[NR];
        xf_pat<NR>(pat, m, A, B, Al, Bl);
        // a rotated leaf (pat != 1): a local direction component may have cancelled to exactly zero (world components cannot:
        // tame).  Its reciprocal is then infinite and the refined one NaN: 0 * rc makes the tracker NaN for good, one two-cycle
        // operation per reciprocal instead of the v_div_fixup of every quotient.
        const bool rot = pat != 1;
        float skip[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) skip[r] = kFlat ? 0.0f : skipf[r] - pcf;
        // one body per (kind, division) runs, kind and op_ieee being wave-uniform; the empty volatile asm keeps hipcc from
        // if-converting the bodies into "compute all, select" (it did: DESIGN.md 4.1)
#define RECT_LEAF(PLANE, DPL, IEEE_)                                                                                        \
        {                                                                                                                \
            float ox, opl, oz;                                                                                           \
            rect_axes<PLANE>(Al, ox, opl, oz);                                                                           \
            const float num = q1[1] - opl;                                                                               \
            _Pragma("unroll") for (int r = 0; r < NR; r++)                                                               \
            {                                                                                                            \
                const float rc = IEEE_ ? 0.0f : fdiv_rcp(Bl[r].DPL);                                                     \
                if (!IEEE_ && rot) chk = __builtin_fmaf(0.0f, rc, chk);                                                  \
                face_fold<PLANE, IEEE_, !kFlat>(q0[0], q0[1], q0[2], q1[0], num, ox, oz, Bl[r], rc, skip[r], op_id_base, cur_t[r], cur_id[r], chk); \
            }                                                                                                            \
        }
        // box::hit primitive.h:229-242: sides in the order XY(p0.z) XY(p1.z) YZ(p0.x) YZ(p1.x) XZ(p0.y) XZ(p1.y); the two sides of an
        // axis divide by the same local direction component
#define BOX_AXIS(IEEE_, PLANE, DPL, X0, Z0, X1, Z1, N0, N1, OX, OZ, F0)                                                    \
            {                                                                                                            \
                const float n0 = (N0), n1 = (N1);                                                                        \
                /* SHADOW: a side that walls the scene in (DOp::slot bits 8-13) is proven unreachable instead of tested, like a rect */ \
                const bool wall0 = kBoxWalls && !(IEEE_) && ((OPW(2) >> (8 + (F0))) & 1), wall1 = kBoxWalls && !(IEEE_) && ((OPW(2) >> (9 + (F0))) & 1); \
                _Pragma("unroll") for (int r = 0; r < NR; r++) {                                                         \
                    float t0, t1;                                                                                        \
                    float rcw = 0.0f;                                                                                    \
                    if (IEEE_) { t0 = n0 / Bl[r].DPL; t1 = n1 / Bl[r].DPL; }                                             \
                    else {                                                                                               \
                        const float rc = fdiv_rcp(Bl[r].DPL);                                                            \
                        rcw = rc;                                                                                        \
                        if (rot) chk = __builtin_fmaf(0.0f, rc, chk);                                                    \
                        if (wall0 && wall1) { t0 = 0.0f; t1 = 0.0f; }                                                    \
                        else fdiv_q2_nofix(n0, n1, Bl[r].DPL, rc, t0, t1);                                               \
                    }                                                                                                    \
                    if (wall0) beyond = beyond || !(fabsf(n0 * rcw - kWallMid) > kWallHalf);                              \
                    else face_fold_t<PLANE, IEEE_, !kFlat>(X0, Z0, X1, Z1, t0, OX, OZ, Bl[r], skip[r], op_id_base + (F0), cur_t[r], cur_id[r], chk);     \
                    if (wall1) beyond = beyond || !(fabsf(n1 * rcw - kWallMid) > kWallHalf);                              \
                    else face_fold_t<PLANE, IEEE_, !kFlat>(X0, Z0, X1, Z1, t1, OX, OZ, Bl[r], skip[r], op_id_base + (F0) + 1, cur_t[r], cur_id[r], chk); \
                }                                                                                                        \
            }
#define BOX_LEAF(IEEE_)                                                                                                  \
        {   /* axis by axis, both sides of an axis for every ray: two numerators live at a time */                       \
            BOX_AXIS(IEEE_, 0, z, q0[0], q0[1], q1[0], q1[1], q0[2] - Al.z, q1[2] - Al.z, Al.x, Al.y, 0)                  \
            BOX_AXIS(IEEE_, 2, x, q0[1], q0[2], q1[1], q1[2], q0[0] - Al.x, q1[0] - Al.x, Al.y, Al.z, 2)                  \
            BOX_AXIS(IEEE_, 1, y, q0[0], q0[2], q1[0], q1[2], q0[1] - Al.y, q1[1] - Al.y, Al.x, Al.z, 4)                  \
        }
#if defined(PT_SPEC_HEADER) && defined(PT_SPEC_NWALL)
#define WALL_LEAF(PLANE, DPL)                                                                                            \
        {                                                                                                                \
            float ox, opl, oz;                                                                                           \
            rect_axes<PLANE>(Al, ox, opl, oz);                                                                           \
            const float num = q1[1] - opl;                                                                               \
            _Pragma("unroll") for (int r = 0; r < NR; r++)                                                               \
            {                                                                                                            \
                const float tq = num * fdiv_rcp(Bl[r].DPL);                                                              \
                beyond = beyond || !(fabsf(tq - kWallMid) > kWallHalf);                                                  \
            }                                                                                                            \
        }
        if (SHADOW && PT_SHADOW_WALLS && (OPW(2) & 48) == 32 && kind >= OP_LEAF_RECT_XY && kind <= OP_LEAF_RECT_YZ) {
            if (kind == OP_LEAF_RECT_XY) { asm volatile("; wall xy"); WALL_LEAF(0, z) }
            else if (kind == OP_LEAF_RECT_YZ) { asm volatile("; wall yz"); WALL_LEAF(2, x) }
            else { asm volatile("; wall xz"); WALL_LEAF(1, y) }
            continue;
        }
#undef WALL_LEAF
#endif
        if (kind == OP_LEAF_RECT_XY) { if (op_ieee) { asm volatile("; rect xy ieee"); RECT_LEAF(0, z, true) } else { asm volatile("; rect xy"); RECT_LEAF(0, z, false) } }
        else if (kind == OP_LEAF_RECT_YZ) { if (op_ieee) { asm volatile("; rect yz ieee"); RECT_LEAF(2, x, true) } else { asm volatile("; rect yz"); RECT_LEAF(2, x, false) } }
        else if (kind == OP_LEAF_RECT_XZ) { if (op_ieee) { asm volatile("; rect xz ieee"); RECT_LEAF(1, y, true) } else { asm volatile("; rect xz"); RECT_LEAF(1, y, false) } }
        else if (kind == OP_LEAF_BOX) { if (op_ieee) { asm volatile("; box ieee"); BOX_LEAF(true) } else { asm volatile("; box fast"); BOX_LEAF(false) } }
        // Sphere and constant_medium leaves (GA scenes): world_hit_fast's arithmetic, folded with the same rule.  Their t can be NaN
        // (a NaN boundary hit, a / a with a = |Bl|^2 outside the precondition): the tracker follows every accepted t.
#define FOLD_GA(r, e_, t_, id_)                                                                                          \
        {                                                                                                                \
            float eg_ = fmaxf((e_), (t_) - cur_t[r]);                                                                    \
            if (!kFlat) eg_ = fmaxf(eg_, skip[r]);                                                                       \
            const bool take_ = !(eg_ > 0.0f);                                                                            \
            cur_t[r] = take_ ? (t_) : cur_t[r];                                                                          \
            cur_id[r] = take_ ? (id_) : cur_id[r];                                                                       \
            chk = __builtin_fmaf(0.0f, take_ ? (t_) : 0.0f, chk);   /* the accepted t only: cur_t is -inf for root-box misses (see face_fold_t) */ \
        }
        else if (GA && kind == OP_LEAF_VOLBOX) {   // constant_medium::hit volume.h:29-93 with a box boundary
#pragma unroll
            for (int r = 0; r < NR; r++) {
                float t1v, t2v;
                int f1, f2;
                if (op_ieee) {
                    asm volatile("; volbox ieee");
                    box_hit_shared(q0, q1, Al, Bl[r], -FLT_MAX, FLT_MAX, t1v, f1);
                    box_hit_shared(q0, q1, Al, Bl[r], (float)((double)t1v + 0.0001), FLT_MAX, t2v, f2);
                    chk = __builtin_fmaf(0.0f, t1v, chk);   // a NaN boundary t: let the general sweep decide
                    chk = __builtin_fmaf(0.0f, t2v, chk);
                } else if (!PT_VOL_SHARED) {
                    asm volatile("; volbox fast, two queries");
                    box_hit_fast(q0, q1, Al, Bl[r], -FLT_MAX, FLT_MAX, t1v, f1);
                    box_hit_fast(q0, q1, Al, Bl[r], (float)((double)t1v + 0.0001), FLT_MAX, t2v, f2);
                    chk = __builtin_fmaf(0.0f, t1v, chk);
                    chk = __builtin_fmaf(0.0f, t2v, chk);
                } else {
                    // Round 5: BOTH boundary queries of the medium (volume.h:38-40: boundary->hit(r, -FLT_MAX, FLT_MAX, rec1), then
                    // boundary->hit(r, rec1.t + 0.0001, FLT_MAX, rec2)) from ONE evaluation of the box's six sides.  The two calls
                    // compute the same six quotients and the same six hit points; they differ only in which sides the t range
                    // lets through.  hittable_list::hit keeps the smallest accepted t (primitive.h:243-246, hittable_list.h:21-38), and
                    // the medium wants the two t, not the sides: with tt[i] = the side's t where its hit point is inside the side's
                    // bounds and +inf elsewhere, rec1.t = min tt and rec2.t = min { tt[i] : !(tt[i] < rec1.t + 0.0001) }.  A local
                    // direction component that cancelled to exactly zero (rotated leaves only) would make a quotient infinite or NaN:
                    // 0 * its refined reciprocal turns the tracker NaN and the wave repeats the query on the general sweep, so every
                    // t here is finite and the sequential "t <= closest_so_far" fold IS the minimum.
                    asm volatile("; volbox fast");
                    float tt[6];
#define VOL_AXIS(PLANE, DPL, X0, Z0, X1, Z1, N0, N1, OX, OZ, I0)                                                          \
                    {                                                                                                    \
                        float dx_, dpl_, dz_, ta_, tb_;                                                                  \
                        rect_axes<PLANE>(Bl[r], dx_, dpl_, dz_);                                                         \
                        const float rc_ = fdiv_rcp(dpl_);                                                                \
                        if (rot) chk = __builtin_fmaf(0.0f, rc_, chk);                                                   \
                        fdiv_q2_nofix((N0), (N1), dpl_, rc_, ta_, tb_);                                                  \
                        const float xa_ = (OX) + ta_ * dx_, za_ = (OZ) + ta_ * dz_, xb_ = (OX) + tb_ * dx_, zb_ = (OZ) + tb_ * dz_; \
                        tt[(I0)] = (side_excess((X0), (Z0), (X1), (Z1), xa_, za_) > 0.0f) ? INFINITY : ta_;              \
                        tt[(I0) + 1] = (side_excess((X0), (Z0), (X1), (Z1), xb_, zb_) > 0.0f) ? INFINITY : tb_;          \
                    }
                    VOL_AXIS(0, z, q0[0], q0[1], q1[0], q1[1], q0[2] - Al.z, q1[2] - Al.z, Al.x, Al.y, 0)
                    VOL_AXIS(2, x, q0[1], q0[2], q1[1], q1[2], q0[0] - Al.x, q1[0] - Al.x, Al.y, Al.z, 2)
                    VOL_AXIS(1, y, q0[0], q0[2], q1[0], q1[2], q0[1] - Al.y, q1[1] - Al.y, Al.x, Al.z, 4)
#undef VOL_AXIS
                    const float m1 = fminf(fminf(fminf(tt[0], tt[1]), fminf(tt[2], tt[3])), fminf(tt[4], tt[5]));
                    f1 = (m1 < INFINITY) ? 0 : -1;
                    t1v = (m1 < INFINITY) ? m1 : FLT_MAX;                 // no side hit: closest_so_far stays t_max
                    const float lo2 = (float)((double)t1v + 0.0001);
                    float m2 = INFINITY;
#pragma unroll
                    for (int i = 0; i < 6; i++) m2 = fminf(m2, (tt[i] < lo2) ? INFINITY : tt[i]);
                    f2 = (m2 < INFINITY) ? 0 : -1;
                    t2v = (m2 < INFINITY) ? m2 : FLT_MAX;
                }
                bool hit = (f1 >= 0) && (f2 >= 0);
                t1v = (t1v < T_MIN) ? T_MIN : t1v;
                t2v = (t2v > T_MAX) ? T_MAX : t2v;
                hit = hit && !(t1v >= t2v);
                t1v = (t1v < 0) ? 0.0f : t1v;
                float tv = 0.0f;
                // the free-flight draw (volume.h:70) only where some ray of the wave crosses the medium at all: its result is used
                // by no other lane, and in stream mode a draw that is not made costs its dimension nothing
                if (!PT_VOL_LAZY_DRAW || __any(hit)) {
                    const float dlen = vlen(Bl[r]);
                    const float distance_inside = (t2v - t1v) * dlen;
                    const float u = rndf(k0, k1, vol_dim_base[r] + (uint32_t)OPW(23));
                    const float hit_distance = (-(1 / OPF(18))) * ptm_logf(u);
                    hit = hit && (hit_distance < distance_inside);
                    tv = t1v + hit_distance / dlen;
                }
                FOLD_GA(r, hit ? 0.0f : 1.0f, tv, op_id_base)
            }
        } else if (GA && kind == OP_LEAF_SPHERE) {   // sphere::hit primitive.h:64-95 (IEEE divisions: a = |Bl|^2 may leave the precondition)
            const v3 oc = vsub(Al, V(q0[0], q0[1], q0[2]));
            const float c = vdot(oc, oc) - q1[0] * q1[0];
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const float a = vdot(Bl[r], Bl[r]);
                const float b = vdot(oc, Bl[r]);
                const float disc = b * b - a * c;
                const float ta = (-b - sqrtf(disc)) / a, tb = (-b + sqrtf(disc)) / a;
                const bool ha = (ta < T_MAX) && (ta > T_MIN), hb = (tb < T_MAX) && (tb > T_MIN);
                const float ts = ha ? ta : tb;
                FOLD_GA(r, ((disc > 0) && (ha || hb)) ? 0.0f : 1.0f, ts, op_id_base)
            }
        } else if (GA && kind == OP_LEAF_VOLSPHERE) {   // constant_medium::hit volume.h:29-93 with a sphere boundary
            const v3 oc = vsub(Al, V(q0[0], q0[1], q0[2]));
            const float c = vdot(oc, oc) - q1[0] * q1[0];
#pragma unroll
            for (int r = 0; r < NR; r++) {
                float t1v = 0.0f, t2v = 0.0f, tv;
                bool hit = sphere_t(oc, c, Bl[r], -FLT_MAX, FLT_MAX, t1v);
                hit = hit && sphere_t(oc, c, Bl[r], (float)((double)t1v + 0.0001), FLT_MAX, t2v);
                chk = __builtin_fmaf(0.0f, t1v, chk);
                chk = __builtin_fmaf(0.0f, t2v, chk);
                const float u = rndf(k0, k1, vol_dim_base[r] + (uint32_t)OPW(23));
                hit = medium_decide(hit, t1v, t2v, Bl[r], OPF(18), u, tv);
                FOLD_GA(r, hit ? 0.0f : 1.0f, tv, op_id_base)
            }
        }
#undef FOLD_GA
        // OP_LEAF_NONE: never a hit
#undef BOX_LEAF
#undef BOX_AXIS
#undef RECT_LEAF
#undef OPF
#undef OPW
    }
    if (S.n_chain) {   // flat program: the winner's parent box, per lane (see world_hit_fast)
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const float4 *ch = S.chains + (size_t)max(cur_id[r] >> 3, 0) * 2;
            const float4 lo = ch[0], hi = ch[1];
            const v3 inv_r = inv[r];
            const float ax = (lo.x - A.x) * inv_r.x, cx = (hi.x - A.x) * inv_r.x;
            const float ay = (lo.y - A.y) * inv_r.y, cy = (hi.y - A.y) * inv_r.y;
            const float az = (lo.z - A.z) * inv_r.z, cz = (hi.z - A.z) * inv_r.z;
            const float tmin = fmaxf(fmaxf(fmaxf(fminf(ax, cx), fminf(ay, cy)), fminf(az, cz)), T_MIN);
            const float tmax = fminf(fminf(fminf(fmaxf(ax, cx), fmaxf(ay, cy)), fmaxf(az, cz)), T_MAX);
            chk = ((tmax <= tmin) && cur_id[r] >= 0) ? NAN : chk;
        }
    }
#if defined(PT_SPEC_HEADER) && defined(PT_SPEC_NWALL)
    if (SHADOW && PT_SHADOW_WALLS && PT_SPEC_NWALL > 0) {
#pragma unroll
        for (int r = 0; r < NR; r++) beyond = beyond || (cur_id[r] >= 0 && !(cur_t[r] <= 1.0f + 0x1p-14f));
    }
#endif
#pragma unroll
    for (int r = 0; r < NR; r++) { out_t[r] = (cur_id[r] >= 0) ? cur_t[r] : 0.0f; out_id[r] = cur_id[r]; }
    return is_nanf(chk) || beyond;
}

// ------------------------------------------------------------------------------------------------
// The WALK: World::hit for scenes of many instances (DScene::walk; the sweeps cost O(#ops) per ray whatever the ray
// hits).  Every lane descends the bvh_node tree on its own -- node boxes and leaf records are per-lane loads, the pending
// right children sit on a per-lane stack (the general sweep's short-stack area) -- in bvh_node::hit's order, left subtree
// before right, so the fold of world_hit_fast applies unchanged: closest hit, the later leaf on equal t, a leaf accepted
// with a NaN t reported for the general sweep.  Leaf arithmetic is world_hit_fast's (tame waves only: unscaled exact
// division, min / max slabs) with the plain 3x4 transform; "while-while": all lanes first descend to a leaf, then all
// evaluate their leaves, whatever kinds they are.
// ------------------------------------------------------------------------------------------------
DEVI bool world_hit_walk(const DScene &S, bool lane_valid, v3 A, v3 B, uint32_t k0, uint32_t k1, uint32_t vol_dim, Stk stk,
                         float &out_t, int &out_id)
{
    const float T_MIN = 0.001f, T_MAX = FLT_MAX;   // integrator.h:193,246
    const v3 inv = V(fdiv(1.0f, B.x), fdiv(1.0f, B.y), fdiv(1.0f, B.z));   // aabb.h:38
    const int NONE = (int)0x80000000;
    float cur_t = FLT_MAX, chk = 0.0f;
    int cur_id = -1, sp = 0;
    int cur = lane_valid ? 0 : NONE;   // >= 0: bvh node, < 0 and != NONE: leaf ~(index of its DOp), NONE: this lane is done
    for (;;) {
        while (__any(cur >= 0)) {
            if (cur >= 0) {   // aabb::hit aabb.h:34-53, the slab arithmetic of the fast sweep's ENTER
                const float4 lo = S.wnodes[2 * cur], hi = S.wnodes[2 * cur + 1];
                const float ax = (lo.x - A.x) * inv.x, cx = (hi.x - A.x) * inv.x;
                const float ay = (lo.y - A.y) * inv.y, cy = (hi.y - A.y) * inv.y;
                const float az = (lo.z - A.z) * inv.z, cz = (hi.z - A.z) * inv.z;
                const float tmin = fmaxf(fmaxf(fmaxf(fminf(ax, cx), fminf(ay, cy)), fminf(az, cz)), T_MIN);
                const float tmax = fminf(fminf(fminf(fmaxf(ax, cx), fmaxf(ay, cy)), fmaxf(az, cz)), T_MAX);
                if (tmax <= tmin) {
                    if (sp > 0) { sp--; cur = __float_as_int(stk.p[(size_t)sp * stk.stride].x); } else cur = NONE;
                } else {   // bvh.h:36-38: left->hit, then right->hit
                    stk.p[(size_t)sp * stk.stride] = make_float2(hi.w, 0.0f);
                    sp++;
                    cur = __float_as_int(lo.w);
                }
            }
        }
        if (!__any(cur != NONE)) break;
        if (cur != NONE) {
            const float4 *op = reinterpret_cast<const float4 *>(S.ops + ~cur);
            const float4 w0 = op[0], w1 = op[1], w2 = op[2], w3 = op[3], w4 = op[4], w5 = op[5];
            const int kind = __float_as_int(w0.x), inst = __float_as_int(w0.y);
            const bool ieee = (__float_as_int(w0.z) & 16) != 0;   // leaf data outside the unscaled division's precondition
            const float m[12] = {w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
            const v3 Al = xf_point(m, A);            // instance::hit primitive.h:298-312
            const v3 Bl = xf_linear(m, B);
            float t = 0.0f, e = 1.0f;
            int face = 0;
            if (kind <= OP_LEAF_RECT_YZ) {           // rect::hit primitive.h:186-225; g = x0 z0 x1 z1 y
                const int plane = kind - OP_LEAF_RECT_XY;
                const v3 o = shuffle(Al, plane), d = shuffle(Bl, plane);
                t = ieee ? (w5.x - o.y) / d.y : fdiv(w5.x - o.y, d.y);
                const float xh = o.x + t * d.x, zh = o.z + t * d.z;
                e = fmaxf(fmaxf(T_MIN - t, t - T_MAX), fmaxf(fmaxf(w4.x - xh, xh - w4.z), fmaxf(w4.y - zh, zh - w4.w)));
            } else if (kind == OP_LEAF_BOX) {
                const float p0[3] = {w4.x, w4.y, w4.z}, p1[3] = {w4.w, w5.x, w5.y};
                if (ieee) box_hit_shared(p0, p1, Al, Bl, T_MIN, T_MAX, t, face);
                else box_hit_fast(p0, p1, Al, Bl, T_MIN, T_MAX, t, face);
                e = (face >= 0) ? 0.0f : 1.0f;
            } else if (kind == OP_LEAF_VOLBOX) {     // constant_medium::hit volume.h:29-93 with a box boundary
                const float p0[3] = {w4.x, w4.y, w4.z}, p1[3] = {w4.w, w5.x, w5.y};
                float t1v, t2v;
                int f1, f2;
                if (ieee) {
                    box_hit_shared(p0, p1, Al, Bl, -FLT_MAX, FLT_MAX, t1v, f1);
                    box_hit_shared(p0, p1, Al, Bl, (float)((double)t1v + 0.0001), FLT_MAX, t2v, f2);
                } else {
                    box_hit_fast(p0, p1, Al, Bl, -FLT_MAX, FLT_MAX, t1v, f1);
                    box_hit_fast(p0, p1, Al, Bl, (float)((double)t1v + 0.0001), FLT_MAX, t2v, f2);
                }
                chk = __builtin_fmaf(0.0f, t1v, chk);
                chk = __builtin_fmaf(0.0f, t2v, chk);
                const float u = rndf(k0, k1, vol_dim + (uint32_t)__float_as_int(w5.w));
                e = medium_decide((f1 >= 0) && (f2 >= 0), t1v, t2v, Bl, w5.z, u, t) ? 0.0f : 1.0f;
                face = 0;
            } else if (kind == OP_LEAF_SPHERE || kind == OP_LEAF_VOLSPHERE) {   // g = center xyz, radius
                const v3 oc = vsub(Al, V(w4.x, w4.y, w4.z));
                const float c = vdot(oc, oc) - w4.w * w4.w;
                if (kind == OP_LEAF_SPHERE) {        // sphere::hit primitive.h:64-95
                    e = sphere_t(oc, c, Bl, T_MIN, T_MAX, t) ? 0.0f : 1.0f;
                } else {
                    float t1v = 0.0f, t2v = 0.0f;
                    bool hit = sphere_t(oc, c, Bl, -FLT_MAX, FLT_MAX, t1v);
                    hit = hit && sphere_t(oc, c, Bl, (float)((double)t1v + 0.0001), FLT_MAX, t2v);
                    chk = __builtin_fmaf(0.0f, t1v, chk);
                    chk = __builtin_fmaf(0.0f, t2v, chk);
                    const float u = rndf(k0, k1, vol_dim + (uint32_t)__float_as_int(w5.w));
                    e = medium_decide(hit, t1v, t2v, Bl, w5.z, u, t) ? 0.0f : 1.0f;
                }
            }                                        // OP_LEAF_NONE: e stays 1
            const bool take = !(fmaxf(e, t - cur_t) > 0.0f);
            cur_t = take ? t : cur_t;
            cur_id = take ? (inst * 8 + face) : cur_id;
            chk = __builtin_fmaf(0.0f, cur_t, chk);
            if (sp > 0) { sp--; cur = __float_as_int(stk.p[(size_t)sp * stk.stride].x); } else cur = NONE;
        }
    }
    out_t = (cur_id >= 0) ? cur_t : 0.0f;
    out_id = cur_id;
    return is_nanf(chk);
}

// Wave-level leaf cull (round 5, per-scene build of flat programs).  The node boxes of the Cornell scenes cull nothing per RAY (7.0 of 8
// instances tested) and the flat program dropped them -- but the rays of one WAVE are coherent at bounce 0: 85 % of the
// camera-ray waves and 63 % of the bounce-0 shadow sweeps of cornell_box 1080p touch neither block (software count over 40 rows).  For
// every cullable leaf (a box, or a medium on a box: bit 6 of DOp::slot, enlarged world box in g[10..15]) the slab test of the enlarged box
// on the exact reciprocals the sweep computes anyway; a wave in which NO ray touches ANY of them runs the instantiation without those
// leaves -- two straight-line programs behind one scalar branch, not a branch per leaf (a per-leaf skip broke the sweep's one basic
// block: k_connect +10 %, measured).  Used by k_extend's bounce-0 instantiation only (-4.5 % of k_extend): in k_connect the second
// program costs the five-wave build 12 more spilled VGPRs (+23 %), and at four waves the cull only buys back what the fourth wave
// loses (11.41 against 11.43 ms); at bounce 1 the waves are no longer coherent (profiles/experiments/r05_ab_runs.json, r05_cull*).  Conservative, so exact: a face the reference accepts has its computed hit point inside the
// face's bounds; with coordinates within 2^12 and scales within 2^+-4 that point is within 2^-5 (world units) of the exact line at the
// accepted t, the enlarged box is >= 0.25 further out, and the slab parameters' own error (2^-21 relative of |bound - origin| <= 2^13)
// is 2^-8: the exact line is inside all three slabs at that t with room to spare, so "no ray touches" is never wrong.  The one hit
// that needs no geometry -- t = 0 / 0 on a face whose plane holds the ray, accepted wherever the line runs -- needs a local
// direction component that cancelled to exactly zero in a rotated leaf: such a ray counts as touching.
#if defined(PT_SPEC_HEADER) && defined(PT_SPEC_NCULL)
template <int NR>
DEVI bool wave_skips_cullable(bool lane_valid, v3 A, const v3 (&B)[NR])
{
    if (PT_SPEC_NCULL == 0) return false;
    bool touch = !(fmaxf(fmaxf(fabsf(A.x), fabsf(A.y)), fabsf(A.z)) <= 4096.0f);   // the margin's argument needs the origin within 2^12
    v3 inv[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {   // aabb.h:38, the sweep's own exact reciprocals (the compiler shares them with the sweep's)
        const float rx = fdiv_rcp(B[r].x), ry = fdiv_rcp(B[r].y), rz = fdiv_rcp(B[r].z);
        fdiv_q2_nofix_dd(1.0f, 1.0f, B[r].x, B[r].y, rx, ry, inv[r].x, inv[r].y);
        inv[r].z = fdiv_q_nofix(1.0f, B[r].z, rz);
    }
#pragma unroll
    for (int pc = 0; pc < PT_SPEC_N; ++pc) {
        if (kSpecW[pc][0] < OP_LEAF_RECT_XY || (kSpecW[pc][2] & 80) != 64) continue;
#define CF(i) __int_as_float(kSpecW[pc][4 + (i)])
        const float lx = CF(22) - A.x, ly = CF(23) - A.y, lz = CF(24) - A.z, hx = CF(25) - A.x, hy = CF(26) - A.y, hz = CF(27) - A.z;
        const int pat = kSpecW[pc][2] & 15;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const float ax = lx * inv[r].x, cx = hx * inv[r].x, ay = ly * inv[r].y, cy = hy * inv[r].y, az = lz * inv[r].z, cz = hz * inv[r].z;
            const float tn = fmaxf(fmaxf(fmaxf(fminf(ax, cx), fminf(ay, cy)), fminf(az, cz)), 0.0f);
            const float tf = fminf(fminf(fmaxf(ax, cx), fmaxf(ay, cy)), fmaxf(az, cz));
            touch = touch || !(tf < tn);
            if (pat != 1) {   // a rotated leaf: an exactly zero local direction component (see above)
                const float m[12] = {CF(0), CF(1), CF(2), CF(3), CF(4), CF(5), CF(6), CF(7), CF(8), CF(9), CF(10), CF(11)};
                const v3 bl = (pat == 2) ? xf_axis_linear<0>(m, B[r]) : ((pat == 3) ? xf_axis_linear<1>(m, B[r]) : ((pat == 4) ? xf_axis_linear<2>(m, B[r]) : xf_linear(m, B[r])));
                touch = touch || (fminf(fminf(fabsf(bl.x), fabsf(bl.y)), fabsf(bl.z)) == 0.0f);
            }
        }
#undef CF
    }
    return !__any(touch && lane_valid);
}
#endif

// World::hit for NR rays of one origin: picks the sweep for this wave (wave-uniform, one scalar branch).
// CULL: this instantiation may use the wave-level leaf cull (k_extend of bounce 0: camera rays)
template <int NR, bool GA, bool WALK = false, bool CULL = false, bool SHADOW = false>
DEVI void world_hit(const DScene &S, bool lane_valid, v3 A, const v3 (&B)[NR], uint32_t k0, uint32_t k1,
                    const uint32_t (&vol_dim_base)[NR], Stk stk, float (&out_t)[NR], int (&out_id)[NR])
{
    // tame: A components zero or 2^-20 <= |x| <= 2^20, B components 2^-20 <= |x| <= 2^20 (non-zero); implies finite
    bool tame = fdiv_in_range(A.x, -20, 20) && fdiv_in_range(A.y, -20, 20) && fdiv_in_range(A.z, -20, 20);
    bool fin = isfinite(A.x) && isfinite(A.y) && isfinite(A.z);
#pragma unroll
    for (int r = 0; r < NR; r++) {
        tame = tame && fdiv_in_range_nz(B[r].x, -20, 20) && fdiv_in_range_nz(B[r].y, -20, 20) && fdiv_in_range_nz(B[r].z, -20, 20);
        fin = fin && isfinite(B[r].x) && isfinite(B[r].y) && isfinite(B[r].z);
    }
    bool general = true;
    if (S.tame && __all(tame || !lane_valid)) {
        if constexpr (WALK) {
            bool redo = false;
#pragma unroll
            for (int r = 0; r < NR; r++) redo = world_hit_walk(S, lane_valid, A, B[r], k0, k1, vol_dim_base[r], stk, out_t[r], out_id[r]) || redo;
            general = __any(redo && lane_valid);
        } else
        {
            // the round-4 form of the fast sweep (PT_FAST_RB=0: world_hit_fast, the A/B)
            bool redo;
            if constexpr (PT_FAST_RB) {
#if defined(PT_SPEC_HEADER) && defined(PT_SPEC_NCULL)
                // (tame waves only: every direction component finite and non-zero, so the reciprocals of the test are finite)
                if (CULL && PT_SPEC_NCULL > 0 && PT_SPEC_FLAT && wave_skips_cullable<NR>(lane_valid, A, B))
                    redo = world_hit_fast_rb<NR, GA, true>(S, lane_valid, A, B, k0, k1, vol_dim_base, out_t, out_id);
                else
#endif
                redo = world_hit_fast_rb<NR, GA, false, SHADOW>(S, lane_valid, A, B, k0, k1, vol_dim_base, out_t, out_id);
            }
            else redo = world_hit_fast<NR, GA>(S, lane_valid, A, B, k0, k1, vol_dim_base, out_t, out_id);
            general = __any(redo && lane_valid);
        }
#ifdef PT_DBG_NO_REDO
        general = false;
#endif
    }
    if (general) world_hit_n<NR, GA>(S, lane_valid, A, B, k0, k1, vol_dim_base, stk, out_t, out_id, __all(fin || !lane_valid));
}

// ------------------------------------------------------------------------------------------------
// textures (texture.h, image.h): per-lane table reads, off the fast path of the constant-texture scenes
// ------------------------------------------------------------------------------------------------
// perlin::noise texture.h:134-160 + perlin_interp :111-131 (the Hermite smoothing is applied twice, as written there)
DEVI float perlin_noise(const DScene &S, v3 p)
{
    float u = p.x - floorf(p.x);
    float v = p.y - floorf(p.y);
    float w = p.z - floorf(p.z);
    u = u * u * (3 - 2 * u);
    v = v * v * (3 - 2 * v);
    w = w * w * (3 - 2 * w);
    const int i = f2i_x86(floorf(p.x));
    const int j = f2i_x86(floorf(p.y));
    const int k = f2i_x86(floorf(p.z));
    const float uu = u * u * (3 - 2 * u);
    const float vv = v * v * (3 - 2 * v);
    const float ww = w * w * (3 - 2 * w);
    float accum = 0;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                const float4 c4 = S.ranvec[S.perm[(i + di) & 255] ^ S.perm[256 + ((j + dj) & 255)] ^ S.perm[512 + ((k + dk) & 255)]];
                const v3 weight_v = V(u - di, v - dj, w - dk);
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) *
                         vdot(V(c4.x, c4.y, c4.z), weight_v);
            }
    return accum;
}
// texture::value(u, v, p) and ::alpha(u, v, p); a checker picks the same child for both (texture.h:43-68)
DEVI void tex_eval(const DScene &S, int ti, float u, float v, v3 p, v3 &color, float &alpha)
{
    for (;;) {
        const DTex t = S.tex[ti];
        if (t.type == 1) {   // PT_TEX_CHECKER
            const float sx = ptm_sinf(t.scale * p.x), sy = ptm_sinf(t.scale * p.y), sz = ptm_sinf(t.scale * p.z);
            ti = (sx * sy * sz > 0) ? t.odd : t.even;
            continue;
        }
        if (t.type == 2) {   // PT_TEX_PERLIN: noise_texture::value texture.h:190-193 = vec3(1,1,1) * noise; alpha(): 1.0
            const float n = perlin_noise(S, vscale(t.scale, p));
            color = V(n, n, n);
            alpha = 1.0f;
            return;
        }
        if (t.type == 3) {   // PT_TEX_IMAGE: image.h:15-49
            v -= f2i_x86(v);
            if (v < 0) v += 1;
            u -= f2i_x86(u);
            if (u < 0) u += 1;
            int y = f2i_x86(v * t.height);
            int x = f2i_x86(u * t.width);
            if (y > t.height - 1) y = t.height - 1;   // u or v == 1.0f after the wrap: the reference reads out of bounds
            if (x > t.width - 1) x = t.width - 1;
            if (y < 0) y = 0;                          // NaN coordinates
            if (x < 0) x = 0;
            const float4 px = S.texels[t.texel0 + y * t.width + x];
            color = V(px.x, px.y, px.z);
            alpha = px.w;
            return;
        }
        color = V(t.r, t.g, t.b);
        alpha = t.a;
        return;
    }
}

// hit_record of the winning primitive: rec.p, rec.normal, rec.mat_ptr  (primitive.h:186-225, 298-312; volume.h:77-88)
struct HitInfo { v3 p, n; int mat; v3 pl; };
// rec.u, rec.v of a rect / box-face hit (primitive.h:206-207; v divides (zh - x0), sic), from the local hit point
DEVI void rect_uv(const DScene &S, int id, v3 pl, float &u, float &v)
{
    const DPrim &pr = S.prims[S.insts[id >> 3].prim];
    u = 0.0f; v = 0.0f;
    if (pr.type > 1) return;
    const DRect &q = pr.r[id & 7];
    const v3 h = shuffle(pl, q.plane);   // pl = o + t*d computed per component, exactly the xh / zh of rect::hit
    u = (h.x - q.x0) / (q.x1 - q.x0);
    v = (h.z - q.x0) / (q.z1 - q.z0);
}
DEVI HitInfo finalize_hit(const DScene &S, v3 A, v3 B, float t, int id, bool need_normal)
{
    const int ii = id >> 3, face = id & 7;
    const DInst &in = S.insts[ii];
    const DPrim &pr = S.prims[in.prim];
    v3 Al = xf_point(in.inv, A);
    v3 Bl = xf_linear(in.inv, B);
    v3 pl = vadd(Al, vscale(t, Bl));   // r.point_at_parameter(t) in local space
    v3 nl;
    HitInfo h;
    // rec.mat_ptr comes from one table load (hit_mat[face], resolved on the host).  Do NOT turn this back into a
    // per-type chain of loads through `pr`: hipcc 7.2 merges such loads into one load behind an address select
    // and materialises the last branch's address on only one predecessor path (seen in the gfx950 ISA; it
    // faulted on volume hits).
    h.mat = pr.hit_mat[face];
    if (pr.type <= 1) nl = rect_normal(pr.r[face], Bl);   // rect: face = 0
    else if (pr.type == 2) nl = vdivf(vsub(pl, V(pr.cx, pr.cy, pr.cz)), pr.radius);
    else nl = V(1.0f, 0.0f, 0.0f);
    h.p = xf_point(in.fwd, pl);
    h.n = need_normal ? xf_normal(in.inv, nl) : V(0.0f, 0.0f, 0.0f);
    h.pl = pl;
    return h;
}

// ------------------------------------------------------------------------------------------------
// lights and pdfs (primitive.h:151-175, 319-342; pdf.h; helpers.h:112-144; material.h)
// ------------------------------------------------------------------------------------------------
DEVI float prim_pdf_value(const DPrim &p, v3 o, v3 v)
{
    if (p.type == 0) {   // rect::pdf_value primitive.h:151-166
        const DRect &q = p.r[0];
        float t;
        if (rect_hit_t(q, o, v, 0.001f, FLT_MAX, t)) {
            float area = (q.x1 - q.x0) * (q.z1 - q.z0);
            float vl = vlen(v);
            float d2 = (t * vl) * (t * vl);
            float cosine = fabsf(vdot(v, rect_normal(q, v)) / vl);
            return d2 / (cosine * area);
        }
        return 0.0f;
    }
    if (p.type == 2) {   // sphere::pdf_value primitive.h:37-51
        float t;
        if (sphere_hit_t(p, o, v, 0.001f, FLT_MAX, t)) {
            float cos_theta_max = sqrtf(1 - p.radius * p.radius / vsqlen(vsub(V(p.cx, p.cy, p.cz), o)));
            float solid_angle = (float)(2 * PT_PI_D * (double)(1 - cos_theta_max));
            return 1 / solid_angle;
        }
        return 0.0f;
    }
    return 0.0f;   // hittable.h:27
}
DEVI float instance_pdf_value(const DScene &S, int ii, v3 o, v3 v)
{   // primitive.h:319-337
    const DInst &in = S.insts[ii];
    return prim_pdf_value(S.prims[in.prim], xf_point(in.inv, o), xf_linear(in.inv, v));
}
struct Onb { v3 u, v, w; };
DEVI Onb onb_from_w(v3 n)
{   // helpers.h:127-136
    Onb b;
    b.w = vunit(n);
    v3 a = (fabsf(b.w.x) > 0.9) ? V(0.0f, 1.0f, 0.0f) : V(1.0f, 0.0f, 0.0f);
    b.v = vunit(vcross(b.w, a));
    b.u = vcross(b.w, b.v);
    return b;
}
DEVI v3 onb_local(const Onb &b, v3 a) { return vadd(vadd(vscale(a.x, b.u), vscale(a.y, b.v)), vscale(a.z, b.w)); }

DEVI v3 prim_random(const DPrim &p, v3 o, uint32_t k0, uint32_t k1, uint32_t dim)
{
    if (p.type == 0) {   // rect::random primitive.h:168-175 (first draw -> z, second -> x)
        const DRect &q = p.r[0];
        double rz = rnd(k0, k1, dim + 0);
        double rx = rnd(k0, k1, dim + 1);
        float pz = (float)((double)q.z0 + rz * (double)(q.z1 - q.z0));
        float px = (float)((double)q.x0 + rx * (double)(q.x1 - q.x0));
        return vsub(shuffle(V(px, q.y, pz), q.plane), o);
    }
    if (p.type == 2) {   // sphere::random primitive.h:52-59 + random.h:45-55
        v3 direction = vsub(V(p.cx, p.cy, p.cz), o);
        float d2 = vsqlen(direction);
        Onb uvw = onb_from_w(direction);
        float r1 = rndf(k0, k1, dim + 0);
        float r2 = rndf(k0, k1, dim + 1);
        float z = 1 + r2 * (sqrtf(1 - p.radius * p.radius / d2) - 1);
        float s, c;
        ptm_sincos_2pi(r1, s, c);
        float x = c * sqrtf(1 - z * z);
        float y = s * sqrtf(1 - z * z);
        return onb_local(uvw, V(x, y, z));
    }
    return V(1.0f, 0.0f, 0.0f);   // hittable.h:28
}
DEVI v3 instance_random(const DScene &S, int ii, v3 o, uint32_t k0, uint32_t k1, uint32_t dim)
{   // primitive.h:338-342
    const DInst &in = S.insts[ii];
    return xf_linear(in.fwd, prim_random(S.prims[in.prim], xf_point(in.inv, o), k0, k1, dim));
}
DEVI float power_heuristic(float fPdf, float gPdf)
{   // helpers.h:138-144, nf = ng = 1, pow = 2
    float f = 1 * fPdf, g = 1 * gPdf;
    float fp = f * f;
    return fp / (fp + g * g);
}
DEVI float cosine_pdf_value(v3 normal, v3 direction)
{   // pdf.h:18-29
    float cosine = vdot(vunit(direction), vunit(normal));
    if (cosine > 0) {
        // The reference computes (float)((double)cosine / M_PI).  q = cosine * RN(1/pi) is within 3 ulp(double) of
        // d = RN(cosine / pi), so (float)q == (float)d unless a float rounding boundary (low 29 mantissa bits =
        // 0x10000000) lies within a few double ulps of q; only then (about 1 call in 10^7), or when the result could be
        // a float denormal, pay for the IEEE double division.
        const double q = (double)cosine * 0.31830988618379067154;
        const unsigned lo = (unsigned)__double2loint(q) & 0x1fffffffu;
        if (lo - 0x0ffffff0u <= 0x20u || !(cosine > 1e-30f)) return (float)((double)cosine / PT_PI_D);
        return (float)q;
    }
    return 0.0f;
}
DEVI float material_value(int type, v3 normal, v3 direction)
{
    if (type == 0 || type == 1) return cosine_pdf_value(normal, direction);   // material.h:66-69, 105-108
    if (type == 4) return (float)(1 / (4 * PT_PI_D));                        // pdf.h:41-44
    return 0.0f;                                                             // void_pdf
}
DEVI v3 random_in_unit_sphere(uint32_t k0, uint32_t k1, uint32_t dim)
{   // random.h:17-24 with cos(acos(x)) = x
    float su, cu;
    ptm_sincos_2pi(rndf(k0, k1, dim + 0), su, cu);
    float cv = (float)(2 * rnd(k0, k1, dim + 1) - 1);
    float sv2 = 1.0f - cv * cv;
    float sv = sqrtf(sv2 > 0.0f ? sv2 : 0.0f);
    float w = ptm_cbrtf(rndf(k0, k1, dim + 2));
    return V(cu * sv * w, cv * w, su * sv * w);
}
DEVI v3 material_generate(int type, v3 normal, uint32_t k0, uint32_t k1, uint32_t dim)
{
    if (type == 0 || type == 1) {   // cosine_pdf::generate pdf.h:30-33, random.h:36-44
        Onb uvw = onb_from_w(normal);
        float r1 = rndf(k0, k1, dim + 0);
        float r2 = rndf(k0, k1, dim + 1);
        float z = sqrtf(1 - r2);
        float s, c;
        ptm_sincos_2pi(r1, s, c);
        float x = c * sqrtf(r2);
        float y = s * sqrtf(r2);
        return onb_local(uvw, V(x, y, z));
    }
    // dielectric::generate (material.h:125-166) picks reflect / refract with one draw, but dielectric::value is 0, so
    // the path ends at scatter_pdf_s < 1e-7 (integrator.h:301-304) before the direction is used: nothing to compute
    if (type == 2) return V(0.0f, 0.0f, 0.0f);
    return random_in_unit_sphere(k0, k1, dim);
}
// albedo->value / emit->value and ->alpha at the hit (rec.u, rec.v, rec.p): the material's own constant texture, or
// its entry of the texture table
DEVI void material_texture(const DScene &S, const DMat &m, int id, const HitInfo &hi, v3 &color, float &alpha)
{
    color = V(m.r, m.g, m.b);
    alpha = m.alpha;
    if (m.tex >= 0) {
        float u = 0.0f, v = 0.0f;
        if (S.tex[m.tex].uses_uv) rect_uv(S, id, hi.pl, u, v);
        tex_eval(S, m.tex, u, v, hi.p, color, alpha);
    }
}
// TEX = false: the scene has no texture table entries in use (every BASELINE scene); the texture code is compiled out
// so that it costs those kernels neither registers nor occupancy
template <bool TEX>
DEVI v3 material_emitted(const DScene &S, const DMat &m, int id, const HitInfo &hi, v3 ray_dir)
{   // material.h:211-229 (others: material.h:21-24)
    if (m.type != 3) return V(0.0f, 0.0f, 0.0f);
    bool aligned = vdot(hi.n, ray_dir) > 0;
    if (!aligned || m.two_sided) {
        v3 c = V(m.r, m.g, m.b);
        float a = m.alpha;
        if (TEX) material_texture(S, m, id, hi, c, a);
        return vscale(a, vscale(m.power, c));
    }
    return V(0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------------------------------------
// workgroup helpers
// ------------------------------------------------------------------------------------------------
// batch-local pixel -> film pixel (i, j)
DEVI void batch_pixel(const DBatch &b, int pl, int &pi, int &pj)
{
    int x0 = b.x0, y0 = b.y0, w = b.w;
    if (b.n_tiles > 1) {
        int lo = 0, hi = b.n_tiles - 1;          // last tile with pix0 <= pl
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (b.tiles[mid].pix0 <= pl) lo = mid; else hi = mid - 1;
        }
        const DTile t = b.tiles[lo];
        x0 = t.x0; y0 = t.y0; w = t.w; pl -= t.pix0;
    }
    const int py = pl / w;
    pi = x0 + (pl - py * w);
    pj = y0 + py;
}
// The order in which a persistent workgroup visits the 256-lane chunks of a segmented queue: linear index c = chunk * n_seg
// + k (chunk-major: the live chunks of every segment are its first few), advanced by the grid stride without a division,
// and k -> segment (k * mul) mod n_seg with mul ~ n_seg / golden ratio, coprime to n_seg (pt_context.cpp seg_perm).  The
// multiplication scatters the segments a workgroup visits quasi-uniformly over the queue: in queue order a workgroup's
// segments are an arithmetic progression of stride gridDim.x, which beats against the period of the image in slot space
// (pixels per sample / segment size) -- at 2^18 pixels per batch every workgroup saw the SAME pixels of every sample, the
// ones over the background idled while the ones over the box did all the work (k_shade 18.0 ms against 10.9).
// How far a launch has to look.  A queue keeps its capacity from bounce to bounce (segments merge, their slots add up) while
// the live paths halve, so from the third bounce on almost every 256-lane chunk slot of a segment is empty, and a persistent
// workgroup that walks all of them pays a scalar load and a compare per slot: ~150 us per k_shade launch however few paths
// were left (half of the time of a batch's last five bounces).  The producer therefore records the largest live count of
// any of its output segments -- every wave keeps the largest end of the ranges it reserved in a scalar register and adds
// it with ONE fire-and-forget atomicMax per queue when it retires, into one of PT_QMAX_BANKS words picked by its workgroup
// (DStreams::qmax; a single word serialised a million atomics per launch: k_shade 16 -> 85 ms) -- and the consumers of that
// queue take the maximum over the banks and stop at the chunk index beyond which every segment is empty.
// PT_CHUNK_BOUND = 0: walk every slot (A/B knob).
#ifndef PT_CHUNK_BOUND
#define PT_CHUNK_BOUND 1
#endif
#define PT_QMAX_BOUNCES 64   // bounces that have words in DStreams::qmax; later ones walk every slot
#define PT_QMAX_BANKS 64
// qmax[(bounce * 2 + word) * PT_QMAX_BANKS + bank], word 0 = path queue, 1 = shadow queue
DEVI int chunk_limit(const int32_t *qmax, int word, int bounce_of_producer, int n_seg, int total_chunks)
{
#if PT_CHUNK_BOUND
    if (bounce_of_producer >= 0 && bounce_of_producer < PT_QMAX_BOUNCES) {
        const int32_t *w = qmax + (size_t)(bounce_of_producer * 2 + word) * PT_QMAX_BANKS;
        int live = 0;
#pragma unroll
        for (int k = 0; k < PT_QMAX_BANKS; k++) live = max(live, w[k]);   // wave-uniform addresses: four scalar loads
        const long long lim = (long long)((live + PT_BLOCK - 1) / PT_BLOCK) * n_seg;   // chunk-major: chunk = c / n_seg
        return lim < total_chunks ? (int)lim : total_chunks;
    }
#endif
    return total_chunks;
}
struct ChunkWalk {
    int chunk, k, seg, g_chunks, g_k, d_seg, n_seg;
    DEVI void init(int n_seg_, int mul)
    {
        n_seg = n_seg_;
        g_chunks = (int)gridDim.x / n_seg; g_k = (int)gridDim.x - g_chunks * n_seg;
        chunk = (int)blockIdx.x / n_seg; k = (int)blockIdx.x - chunk * n_seg;
        seg = (int)(((uint32_t)k * (uint32_t)mul) % (uint32_t)n_seg);       // k, g_k, mul < n_seg < 2^16 (seg_perm): 32-bit products
        d_seg = (int)(((uint32_t)g_k * (uint32_t)mul) % (uint32_t)n_seg);
    }
    DEVI void advance()
    {
        chunk += g_chunks; k += g_k; seg += d_seg;
        if (seg >= n_seg) seg -= n_seg;
        if (k >= n_seg) { k -= n_seg; chunk++; }
    }
};
enum { C_SAMPLES = 0, C_RAYS, C_EXT, C_EXT_HITS, C_SHADOW, C_MISS, C_RR, C_EMIT, C_PDF, C_LIMIT, C_N };
DEVI DCounters *counter_bank(DCounters *g) { return g + (blockIdx.x & (PT_COUNTER_BANKS - 1)); }
DEVI void flush_counters(unsigned int *sh_ctr, DCounters *g)
{
    __syncthreads();
    if (threadIdx.x < C_N) {
        unsigned long long v = sh_ctr[threadIdx.x];
        if (v) atomicAdd(((unsigned long long *)counter_bank(g)) + threadIdx.x, v);
    }
}

// ------------------------------------------------------------------------------------------------
// generate: renderer.h:648-649 jitter + camera::get_ray camera.h:38-47
// ------------------------------------------------------------------------------------------------
// The part of a camera path's record that k_generate does not store (bounce 0 only): k1 of the path in `slot`.
DEVI uint32_t bounce0_k1(const DScene &S, const DBatch &b, int slot)
{
    const uint32_t sample = (uint32_t)b.s0 + (uint32_t)slot / (uint32_t)b.npix;   // slot = s_local * npix + pixel, below 2^30
    return mix_lowbias32(sample ^ S.seed_k1);
}
// One camera sample: pixel jitter (renderer.h:648-649) + camera::get_ray (camera.h:38-47) for the path in `slot`
// (= s_local * npix + batch-local pixel), and its stream RNG keys.
struct CamRay { v3 A, B; uint32_t k0, k1; };
DEVI CamRay camera_ray(const DScene &S, const DBatch &b, int slot)
{
    const uint32_t s_local = (uint32_t)slot / (uint32_t)b.npix;   // slot < 2^30 (pt_create)
    const int pl = (int)((uint32_t)slot - s_local * (uint32_t)b.npix);
    int pi, pj;
    batch_pixel(b, pl, pi, pj);
    const uint32_t pixel = (uint32_t)(pj * S.width + pi);
    const uint32_t sample = (uint32_t)b.s0 + s_local;
    CamRay r;
    r.k0 = mix_lowbias32(pixel ^ S.seed_k0);
    r.k1 = mix_lowbias32(sample ^ S.seed_k1);
    const float u = (float)((double)pi + rnd(r.k0, r.k1, DIM_JITTER_U)) / (float)S.width;
    const float v = (float)((double)pj + rnd(r.k0, r.k1, DIM_JITTER_V)) / (float)S.height;
    const v3 cu = V(S.cam.u[0], S.cam.u[1], S.cam.u[2]), cv = V(S.cam.v[0], S.cam.v[1], S.cam.v[2]);
    v3 offset = V(0.0f, 0.0f, 0.0f);
    if (S.cam.lens_radius != 0.0f) {   // random_in_unit_disk random.h:27-34
        float su, cu2;
        ptm_sincos_2pi(rndf(r.k0, r.k1, DIM_LENS), su, cu2);
        const float rv = sqrtf(rndf(r.k0, r.k1, DIM_LENS + 1));
        const v3 rd = vscale(S.cam.lens_radius, V(cu2 * rv, su * rv, 0.0f));
        offset = vadd(vscale(rd.x, cu), vscale(rd.y, cv));
    }
    const v3 origin = V(S.cam.origin[0], S.cam.origin[1], S.cam.origin[2]);
    const v3 llc = V(S.cam.llc[0], S.cam.llc[1], S.cam.llc[2]);
    const v3 hor = V(S.cam.horizontal[0], S.cam.horizontal[1], S.cam.horizontal[2]);
    const v3 ver = V(S.cam.vertical[0], S.cam.vertical[1], S.cam.vertical[2]);
    r.A = vadd(origin, offset);
    r.B = vsub(vsub(vadd(vadd(llc, vscale(u, hor)), vscale(v, ver)), origin), offset);
    return r;
}
// live entries of input segment `seg`: at bounce 0 the queue IS the batch's slots in order (slot = position), so the count
// is arithmetic; later bounces read what the previous k_shade appended
template <bool B0>
DEVI int seg_live(const DQueue &q, const DBatch &b, int seg)
{
    if (B0) {
        const long long rem = b.n_paths - (long long)seg * b.seg_cap;
        return rem <= 0 ? 0 : (rem < b.seg_cap ? (int)rem : b.seg_cap);
    }
    return q.count[seg];
}
// PT_FUSE_GENERATE (default): camera rays are never stored.  k_extend and k_shade of bounce 0 form the ray of their slot
// themselves (camera_ray above, ~160 vector instructions) instead of reading a 32-byte record that a k_generate launch
// wrote: 96 bytes of HBM traffic less per camera sample (32 written, 32 + 32 read; 13 % of the pipeline's stream bytes)
// and one launch less per batch.  k_generate remains for max_bounces = 0 (no bounce kernel runs) and as the A/B.
__global__ __launch_bounds__(PT_BLOCK) void k_generate(DScene S, DStreams st, DBatch b)
{
    const int seg = blockIdx.x;
    const long long seg_base = (long long)seg * b.seg_cap;
    long long remaining = b.n_paths - seg_base;
    const int n = remaining <= 0 ? 0 : (remaining < b.seg_cap ? (int)remaining : b.seg_cap);
    DQueue q = st.q[0];
    for (int i = threadIdx.x; i < n; i += PT_BLOCK) {
        const long long slot = seg_base + i;
        const CamRay r = camera_ray(S, b, (int)slot);
        // A camera path's record is 32 bytes: beta = 1, attenuation = 0 and last_bsdf_pdf = -1 (integrator.h:183) are constants
        // and k1 follows from the slot (bounce0_k1); k0 rides in the pdf's place.
        q.r0[slot] = make_float4(r.A.x, r.A.y, r.A.z, __int_as_float((int)slot));
        q.r1[slot] = make_float4(r.B.x, r.B.y, r.B.z, __uint_as_float(r.k0));
        st.radiance[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (threadIdx.x == 0) {
        q.count[seg] = n;
        if (n) atomicAdd(&counter_bank(st.counters)->camera_samples, (unsigned long long)n);
        // max_bounces = 0: the integrator's loop never runs and every path ends at the depth limit with no radiance
        if (n && S.max_bounces == 0) atomicAdd(&counter_bank(st.counters)->term_bounce_limit, (unsigned long long)n);
    }
}

// ------------------------------------------------------------------------------------------------
// extend: closest hit of every live path's ray (integrator.h:192-193)
// ------------------------------------------------------------------------------------------------
// B0: the bounce-0 instantiation of a batch whose camera rays are not stored (PT_FUSE_GENERATE): it forms them itself.  The
// other instantiation carries none of that code (the camera's 19 scalars cost the bounce loop 45 spilled SGPRs when the
// choice was a run-time branch).
template <bool GA, bool WALK, bool B0>
__global__ __launch_bounds__(PT_BLOCK) void k_extend(DScene S, const DOp *__restrict__ t_ops, const DInst *__restrict__ t_insts, const DPrim *__restrict__ t_prims,
        const DMat *__restrict__ t_mats, const int32_t *__restrict__ t_lights, const float4 *__restrict__ t_emit, DStreams st, DBatch b, int qi, int bounce)
{
    // scene tables arrive as separate __restrict__ kernel arguments: only then can hipcc prove that the stream
    // stores below never clobber them and keep the wave-uniform table reads on the scalar unit (s_load)
    S.ops = t_ops; S.insts = t_insts; S.prims = t_prims; S.mats = t_mats; S.lights = t_lights; S.emit = t_emit;
    extern __shared__ float2 stack[];   // [stack_depth][1][PT_BLOCK]
    const Stk stk = stack_of(st, stack);
    // Persistent workgroups: a launch has at most a few thousand workgroups (dispatching a 256-thread workgroup costs
    // 3-15 ns of serial dispatcher time on MI355X, which dominated the thin late bounces when every chunk was its own
    // workgroup), and each strides over the 256-lane chunks of the segmented queue; empty chunks cost one scalar load.
    const int cps = b.seg_cap / PT_BLOCK;
    const int total_chunks = b.n_seg * cps;
    const DQueue q = st.q[qi];
    const uint32_t base_dim = DIM_BOUNCE0 + (uint32_t)bounce * dims_per_bounce(S);
    const bool has_vol = S.n_vol > 0;
    unsigned long long n_rays = 0;
    // the live count of a chunk's segment is fetched one chunk ahead: the scalar load is in flight while the current chunk is
    // processed, and an empty chunk (the tail of a segment) costs a compare instead of a memory round trip
    // (chunk, seg) of the next chunk advance by the grid stride without a division per chunk
    ChunkWalk nx;
    nx.init(b.n_seg, b.perm);
    // see chunk_limit.  Chunk 0 of every segment is visited whatever the bound says (chunk-major: its indices are [0, n_seg)):
    // that is where this bounce's output counters are zeroed, and a batch whose paths have all died (bound 0) must still
    // zero them -- else the consumers of a later bounce that trusts the counts again (bounce >= PT_QMAX_BOUNCES, k_tally)
    // would meet the last live bounce's values and walk its stale records a second time.
    const int end_bound = B0 ? total_chunks : chunk_limit(st.qmax, 0, bounce - 1, b.n_seg, total_chunks);
    const int end_chunks = max(end_bound, min(b.n_seg, total_chunks));
    int n_ahead = ((int)blockIdx.x < end_chunks) ? seg_live<B0>(q, b, nx.seg) : 0;
    for (int c = blockIdx.x; c < end_chunks; c += gridDim.x) {
        // chunk-major order: the live chunks of every segment are its first few, so they sit together at the front of
        // the index space and spread evenly over the workgroups (segment-major order would alias with the stride)
        const int chunk = nx.chunk, seg = nx.seg;
        const int n = (c < end_bound) ? n_ahead : 0;   // beyond the bound every segment is empty (its count word may be stale)
        nx.advance();
        if (c + (int)gridDim.x < end_chunks) n_ahead = seg_live<B0>(q, b, nx.seg);
        if (chunk == 0 && threadIdx.x == 0 && (seg & 1) == 0) {
            // zero the counters this bounce's shade will append to (segment g -> g >> 1; every output segment has an
            // even source).  The other path queue and the shadow queue are idle now: their last readers were the
            // previous bounce's kernels on this stream.
            const int so = b.n_seg_out == b.n_seg ? seg : (seg >> 1);
            st.q[qi ^ 1].count[so] = 0;
            st.sq.count[so] = 0;
            if (b.n_seg_out == b.n_seg && seg + 1 < b.n_seg) { st.q[qi ^ 1].count[seg + 1] = 0; st.sq.count[seg + 1] = 0; }
        }
        const int i0 = chunk * PT_BLOCK;
        if (i0 >= n) continue;
        const long long seg_base = (long long)seg * b.seg_cap;
        const int i = i0 + threadIdx.x;
        const bool valid = i < n;
        n_rays += (unsigned long long)((n - i0) < PT_BLOCK ? (n - i0) : PT_BLOCK);
        // whole wave beyond the end: nothing to do (wave-uniform exit keeps the sweep convergent)
        if ((i0 + (int)(threadIdx.x & ~63u)) >= n) continue;
        const long long pos = seg_base + (valid ? i : i0);
        float4 r0, r1;
        uint32_t k0 = 0, k1 = 0;
        if (B0) {   // the camera ray of the slot, formed here (position = slot at bounce 0)
            const CamRay cr = camera_ray(S, b, (int)pos);
            r0 = make_float4(cr.A.x, cr.A.y, cr.A.z, 0.0f);
            r1 = make_float4(cr.B.x, cr.B.y, cr.B.z, 0.0f);
            k0 = cr.k0; k1 = cr.k1;
        } else {
            r0 = ld4<2>(&q.r0[pos]); r1 = ld4<2>(&q.r1[pos]);
            if (has_vol) {
                if (bounce == 0) { k0 = __float_as_uint(r1.w); k1 = bounce0_k1(S, b, __float_as_int(r0.w)); }   // see k_generate
                else { k0 = __float_as_uint(q.s0[pos].w); k1 = __float_as_uint(q.s1[pos].w); }
            }
        }
        float t[1];
        int id[1];
        const v3 Bd[1] = {V(r1.x, r1.y, r1.z)};
        const uint32_t vd[1] = {base_dim};
        world_hit<1, GA, WALK, B0 && PT_CULL_B0 != 0>(S, valid, V(r0.x, r0.y, r0.z), Bd, k0, k1, vd, stk, t, id);
        if (valid) {
            // the hit stream carries the shading class of the face in bits 28-29 of the id (k_shade's sort key): one table
            // read here, at the end of a chunk, instead of a dependent one at the head of k_shade's
            int hid = id[0];
            if (hid >= 0) hid |= __float_as_int(S.faces[(size_t)hid * PT_FACE_F4].x) & (3 << 28);
            st2<2>(&st.hit[pos], make_float2(t[0], __int_as_float(hid)));
        }
    }
    if (threadIdx.x == 0 && n_rays) {
        atomicAdd(&counter_bank(st.counters)->rays, n_rays);
        atomicAdd(&counter_bank(st.counters)->ext_rays, n_rays);
        if (B0) atomicAdd(&counter_bank(st.counters)->camera_samples, n_rays);   // every camera sample is extended once
    }
}

// ------------------------------------------------------------------------------------------------
// shade: one bounce of NEEIterative::color between the two World::hit calls (integrator.h:193-336):
// material scatter + emission/MIS, light sampling -> shadow records, BSDF sampling + russian roulette ->
// continuation ray; survivors are compacted into the next path queue, shadow records into the shadow queue.
// ------------------------------------------------------------------------------------------------
// hit_record of the winning primitive for k_shade: rec.p, rec.normal, rec.mat_ptr (primitive.h:186-225, 298-312;
// volume.h:77-88) plus unit_vector(rec.normal) and the onb of it.  Everything that does not depend on where the face was
// hit comes from the per-face table (pt_device.h PT_FACE_F4), computed on the host with the same float operations; what
// is left per hit is the local hit point, rec.p = transform * p_local and the facing test.
struct ShadeHit { v3 p, n, nu, ou, ov, pl; int mat; };
template <int P> struct PlaneTag { static constexpr int value = P; };   // a rect's alignment as a compile-time value (k_shade's light samples)
DEVI ShadeHit shade_hit(const DScene &S, v3 A, v3 B, float t, int id)
{
    const DInst &in = S.insts[id >> 3];
    const float4 *fc = S.faces + (size_t)id * PT_FACE_F4;
    const float4 h0 = fc[0];
    const int head = __float_as_int(h0.x);
    const int ptype = (head >> 24) & 15;
    const v3 Al = xf_point(in.inv, A);
    const v3 Bl = xf_linear(in.inv, B);
    ShadeHit h;
    h.pl = vadd(Al, vscale(t, Bl));   // r.point_at_parameter(t) in local space
    h.p = xf_point(in.fwd, h.pl);
    h.mat = head & 0xfffff;
    if (ptype != 2) {
        // rect / box side: "if (dot(r.direction(), normal) > 0) normal = -normal" (primitive.h:214-222); volume: no test
        const bool flip = (ptype <= 1) && (vdot(Bl, V(h0.y, h0.z, h0.w)) > 0);
        const float4 *sd = fc + (flip ? 4 : 1);
        const float4 a = sd[0], b = sd[1], c = sd[2];
        h.n = V(a.x, a.y, a.z);
        h.nu = V(a.w, b.x, b.y);
        h.ou = V(b.z, b.w, c.x);
        h.ov = V(c.y, c.z, c.w);
    } else {   // sphere::hit primitive.h:76-78: normal = (p - center) / radius, per hit
        const DPrim &pr = S.prims[in.prim];
        const v3 nl = vdivf(vsub(h.pl, V(pr.cx, pr.cy, pr.cz)), pr.radius);
        h.n = xf_normal(in.inv, nl);
        const Onb o = onb_from_w(h.n);
        h.nu = o.w; h.ou = o.u; h.ov = o.v;
    }
    return h;
}
// cosine_pdf::value (pdf.h:18-29) given cosine = dot(unit_vector(direction), unit_vector(normal))
DEVI float cosine_pdf_of(float cosine)
{
    if (cosine > 0) {
        // The reference computes (float)((double)cosine / M_PI).  q = cosine * RN(1/pi) is within 3 ulp(double) of
        // d = RN(cosine / pi), so (float)q == (float)d unless a float rounding boundary (low 29 mantissa bits =
        // 0x10000000) lies within a few double ulps of q; only then (about 1 call in 10^7), or when the result could be
        // a float denormal, pay for the IEEE double division.
        const double q = (double)cosine * 0.31830988618379067154;
        const unsigned lo = (unsigned)__double2loint(q) & 0x1fffffffu;
        float r = (float)q;
        // one call in 10^7: a real branch (the empty asm keeps hipcc from computing the f64 division -- eleven f64 instructions
        // -- on every call and selecting, which it did)
        if (__builtin_expect(lo - 0x0ffffff0u <= 0x20u || !(cosine > 1e-30f), 0)) { asm volatile("; cosine_pdf: exact division"); r = (float)((double)cosine / PT_PI_D); }
        return r;
    }
    return 0.0f;
}
// material::value(r_in, rec, direction) (material.h:66-69, 105-108; pdf.h:41-44) given that cosine
DEVI float material_value_of(int type, float cosine)
{
    if (type == 0 || type == 1) return cosine_pdf_of(cosine);
    if (type == 4) return (float)(1 / (4 * PT_PI_D));
    return 0.0f;   // void_pdf
}

// LM: how the light of an NEE sample is found -- 1: the scene has one light (scalar records), 2: two lights (both records
// scalar, per-lane select), 0: any number (per-lane gathers).  Separate instantiations keep the registers of one mode out
// of the others.
template <bool TEX, int LM, bool STAGE, bool B0>   // B0: see k_extend
__global__ __launch_bounds__(PT_BLOCK, TEX ? PT_SHADE_WAVES_TEX : PT_SHADE_WAVES) void k_shade(DScene S, const DOp *__restrict__ t_ops, const DInst *__restrict__ t_insts, const DPrim *__restrict__ t_prims,
        const DMat *__restrict__ t_mats, const int32_t *__restrict__ t_lights, const float4 *__restrict__ t_emit, DStreams st, DBatch b, int qi, int bounce)
{
    // scene tables arrive as separate __restrict__ kernel arguments: only then can hipcc prove that the stream
    // stores below never clobber them and keep the wave-uniform table reads on the scalar unit (s_load)
    S.ops = t_ops; S.insts = t_insts; S.prims = t_prims; S.mats = t_mats; S.lights = t_lights; S.emit = t_emit;
    __shared__ int sh_key[3][PT_BLOCK / 64];
    __shared__ unsigned char sh_perm[PT_BLOCK];
    // staged shadow records (DBatch::stage_shadow): [light_samples][PT_BLOCK] float4 (direction, coef.x) then float2 (coef.yz),
    // every lane writes and reads its own column
    extern __shared__ float4 sh_stage[];
    const int cps = b.seg_cap / PT_BLOCK;
    const int total_chunks = b.n_seg * cps;
    const DQueue q = st.q[qi];
    const DQueue qo = st.q[qi ^ 1];
    const DShadowQueue sq = st.sq;
    const long long P = b.P;
    const uint32_t L = (uint32_t)S.light_samples, NV = (uint32_t)S.n_vol;
    const uint32_t D = NV + L * (3u + NV) + 4u;
    const uint32_t base = DIM_BOUNCE0 + (uint32_t)bounce * D;
    const uint32_t gb = base + NV + L * (3u + NV);
    const bool last_bounce = (bounce + 1 >= S.max_bounces);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // path counters: one popcount of a ballot per event and chunk, kept in scalar registers, flushed once per wave
    unsigned n_miss = 0, n_hit = 0, n_rr = 0, n_emit = 0, n_pdf = 0, n_limit = 0, n_dark = 0;
    constexpr bool stage = STAGE;   // DBatch::stage_shadow picked the instantiation (launch_shade)
    float4 *const st_d = sh_stage + threadIdx.x;
    float2 *const st_e = reinterpret_cast<float2 *>(sh_stage + (size_t)L * PT_BLOCK) + threadIdx.x;
    ChunkWalk nx;   // see k_extend
    nx.init(b.n_seg, b.perm);
    const int end_chunks = B0 ? total_chunks : chunk_limit(st.qmax, 0, bounce - 1, b.n_seg, total_chunks);   // see chunk_limit
    // this bounce's words of this workgroup's bank, and the largest range end this wave reserved in either queue (scalars)
    int32_t *const qmax_out = (PT_CHUNK_BOUND && bounce < PT_QMAX_BOUNCES) ? st.qmax + (size_t)(bounce * 2) * PT_QMAX_BANKS + (blockIdx.x & (PT_QMAX_BANKS - 1)) : nullptr;
    int wave_top[2] = {0, 0};
    int n_ahead = ((int)blockIdx.x < end_chunks) ? seg_live<B0>(q, b, nx.seg) : 0;   // one chunk ahead, see k_extend
    for (int c = blockIdx.x; c < end_chunks; c += gridDim.x) {   // persistent workgroups, chunk-major, see k_extend
        const int chunk = nx.chunk, seg = nx.seg;
        const int n = n_ahead;
        nx.advance();
        if (c + (int)gridDim.x < end_chunks) n_ahead = seg_live<B0>(q, b, nx.seg);
        const int i0 = chunk * PT_BLOCK;
        if (i0 >= n) continue;
        const long long seg_base = (long long)seg * b.seg_cap;
        const int seg_o = b.n_seg_out == b.n_seg ? seg : (seg >> 1);
        const long long seg_base_o = (long long)seg_o * b.seg_cap_out;
        // ---- material-sorted shading order inside the chunk (north_star: per-bounce material-sorted shading queues).
        // The 256 paths of a chunk are partitioned by what their hit needs -- 0: untextured lambertian / metal on a rect or
        // box (the long path), 1: every other scattering hit (textures, dielectric, fog, sphere normals), 2: emitters,
        // 3: misses and the unused tail -- with one counting sort in LDS (four ballots per wave, one exchange), so that
        // whole waves run one kind of work instead of every wave running all of them at partial occupancy.  Per-path
        // arithmetic does not depend on the lane that performs it, so the image does not change.
        int src = threadIdx.x;
        if (b.sort_shade) {
            int key = 3;
            if (i0 + (int)threadIdx.x < n) {
                const int hid = __float_as_int(st.hit[seg_base + i0 + threadIdx.x].y);
                if (hid >= 0) key = hid >> 28;   // k_extend put the face's class there
            }
            const unsigned long long m0 = __ballot(key == 0), m1 = __ballot(key == 1), m2 = __ballot(key == 2);
            if (lane == 0) { sh_key[0][wave] = __popcll(m0); sh_key[1][wave] = __popcll(m1); sh_key[2][wave] = __popcll(m2); }
            __syncthreads();
            int before[3] = {0, 0, 0}, total[3] = {0, 0, 0};
#pragma unroll
            for (int w = 0; w < PT_BLOCK / 64; w++)
#pragma unroll
                for (int k = 0; k < 3; k++) { const int cnt = sh_key[k][w]; if (w < wave) before[k] += cnt; total[k] += cnt; }
            const unsigned long long below = (1ull << lane) - 1ull;
            const int r0_ = __popcll(m0 & below), r1_ = __popcll(m1 & below), r2_ = __popcll(m2 & below);
            // class 3 takes what is left, in workgroup order: my index minus the lanes of classes 0..2 before me
            const int before_all = before[0] + before[1] + before[2] + r0_ + r1_ + r2_;
            int pos = total[0] + total[1] + total[2] + ((int)threadIdx.x - before_all);
            pos = (key == 2) ? total[0] + total[1] + before[2] + r2_ : pos;
            pos = (key == 1) ? total[0] + before[1] + r1_ : pos;
            pos = (key == 0) ? before[0] + r0_ : pos;
            sh_perm[pos] = (unsigned char)threadIdx.x;
            __syncthreads();
            src = sh_perm[threadIdx.x];
        }
        const int i = i0 + src;
        const bool valid = i < n;
        bool cont = false, shadow = false, pending = false;
        bool ev_miss = false, ev_rr = false, ev_emit = false, ev_pdf = false, ev_limit = false;
        v3 beta = V(0, 0, 0), att = V(0, 0, 0);      // beta BEFORE this bounce's update; attenuation AFTER scatter()
        v3 nA = V(0, 0, 0), nB = V(0, 0, 0), nbeta = V(0, 0, 0);
        v3 hp = V(0, 0, 0), hnu = V(0, 0, 0);
        float new_pdf = 0.0f;
        uint32_t k0 = 0, k1 = 0;
        int slot = 0, mat_type = 0;
        float4 rad0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // B0: this camera sample's radiance so far
        if (valid) {
            const long long pos = seg_base + i;
            float4 r0, r1, s0, s1;
            const float2 h = ld2<1>(&st.hit[pos]);
            float last_bsdf_pdf;
            if (B0) {            // a camera path: the ray of the slot is formed here, like k_extend formed it (position = slot at
                                 // bounce 0); beta = 1, attenuation = 0 and last_bsdf_pdf = -1 (integrator.h:183) are constants
                const CamRay cr = camera_ray(S, b, (int)pos);
                r0 = make_float4(cr.A.x, cr.A.y, cr.A.z, __int_as_float((int)pos));
                r1 = make_float4(cr.B.x, cr.B.y, cr.B.z, -1.0f);
                s0 = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(cr.k0));
                s1 = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(cr.k1));
                // nobody has written this sample's radiance yet: its sum starts at zero in a register (rad0) and is stored once,
                // below, with whatever this bounce adds -- k_generate's initialisation and this bounce's read-modify-writes in one store
            } else {
                r0 = ld4<1>(&q.r0[pos]); r1 = ld4<1>(&q.r1[pos]);
                if (bounce == 0) {   // a camera path: 32-byte record, the rest are constants (k_generate)
                    s0 = make_float4(1.0f, 1.0f, 1.0f, r1.w);
                    s1 = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(bounce0_k1(S, b, __float_as_int(r0.w))));
                    r1.w = -1.0f;
                } else { s0 = ld4<1>(&q.s0[pos]); s1 = ld4<1>(&q.s1[pos]); }
            }
            const v3 A = V(r0.x, r0.y, r0.z);
            const v3 B = V(r1.x, r1.y, r1.z);
            slot = __float_as_int(r0.w);
            last_bsdf_pdf = r1.w;
            beta = V(s0.x, s0.y, s0.z);
            att = V(s1.x, s1.y, s1.z);
            k0 = __float_as_uint(s0.w);
            k1 = __float_as_uint(s1.w);
            const int id = __float_as_int(h.y) < 0 ? -1 : (__float_as_int(h.y) & 0x0fffffff);   // bits 28-29: shading class
            if (id < 0) {
                // miss: sum += beta * world->value(u, v, unit_direction) (integrator.h:325-336, world.h:27-30); a constant
                // background ignores all three.  TAU is "2 * M_PI" unparenthesised (random.h:7), hence the form of u.
                v3 bg = V(S.bg[0], S.bg[1], S.bg[2]);
                if (TEX && S.bg_tex >= 0) {
                    const v3 ud = vunit(B);
                    float eu = 0.0f, ev = 0.0f, ea;
                    if (S.tex[S.bg_tex].uses_uv) {
                        eu = (float)((PT_PI_D + (double)ptm_atan2f(ud.y, ud.x)) / 2 * PT_PI_D);
                        ev = (float)((double)ptm_acosf(ud.z) / PT_PI_D);
                    }
                    tex_eval(S, S.bg_tex, eu, ev, ud, bg, ea);
                }
                const v3 add = vmul(beta, bg);
                if (B0) rad0 = make_float4(rad0.x + add.x, rad0.y + add.y, rad0.z + add.z, 0.0f);
                else if (!PT_SKIP_NOOP_RADIANCE || !(add.x == 0.0f && add.y == 0.0f && add.z == 0.0f)) {
                    // A black background (the Cornell scenes) adds +-0: the sum keeps its bits -- x + (+-0) = x for every x but -0,
                    // and a radiance component is never -0 (it starts as +0, and a float sum is -0 only when both terms are) --
                    // so the random 16-byte read and write of the slot (a 64-byte sector each way in HBM; a third of the paths that
                    // end at bounce >= 1 end here) are skipped.  NaN and inf * 0 are not equal to 0 and take the addition.
                    const float4 rad = st.radiance[slot];
                    st.radiance[slot] = make_float4(rad.x + add.x, rad.y + add.y, rad.z + add.z, 0.0f);
                }
                ev_miss = true;
            } else {
                const ShadeHit hi = shade_hit(S, A, B, h.x, id);
                // the face's material record sits in the face table: one round trip for everything a hit id leads to
                const DMat m = *reinterpret_cast<const DMat *>(S.faces + (size_t)id * PT_FACE_F4 + 8);
                mat_type = m.type;
                hp = hi.p;
                hnu = hi.nu;
                HitInfo hti;   // the texture / emission helpers' view of the record
                hti.p = hi.p; hti.n = hi.n; hti.mat = hi.mat; hti.pl = hi.pl;
                // scatter(): material.h:39-53 lambertian, :90-98 metal, :187-191 diffuse_light (attenuation keeps
                // its previous value), :252-261 isotropic.  A constant albedo / pi comes from the table.
                bool did_scatter = true;
                v3 albedo = V(m.r, m.g, m.b);
                v3 att_l = V(m.att[0], m.att[1], m.att[2]);
                if (TEX && m.tex >= 0 && m.type != 3) {   // albedo->value(rec.u, rec.v, rec.p)
                    float ta;
                    material_texture(S, m, id, hti, albedo, ta);
                    att_l = vdivf(albedo, PT_PI_F);
                }
                if (m.type == 0) {
                    if (vdot(B, hi.n) < 0) att = att_l;
                    else att = V(0.0f, 0.0f, 0.0f);
                } else if (m.type == 1) att = V(m.att[0], m.att[1], m.att[2]);   // metal: its own colour (material.h:94)
                else if (m.type == 2) att = V(1.0f, 1.0f, 1.0f);   // dielectric::scatter material.h:118-124
                else if (m.type == 3) did_scatter = false;
                else if (m.type == 4) att = albedo;
                const float cos_i = fabsf(vdot(vunit(B), hi.nu));
                const v3 hit_emission = material_emitted<TEX>(S, m, id, hti, B);
                if ((double)vsqlen(hit_emission) > 0.000001) {   // integrator.h:205-218
                    v3 add;
                    if (last_bsdf_pdf <= 0) add = vmul(beta, hit_emission);
                    else {
                        const float lp = instance_pdf_value(S, id >> 3, A, hi.p);   // a POSITION as direction
                        const float weight = power_heuristic(last_bsdf_pdf, lp);
                        add = vscale(weight, vmul(beta, hit_emission));
                    }
                    if (B0) rad0 = make_float4(rad0.x + add.x, rad0.y + add.y, rad0.z + add.z, 0.0f);
                    else {
                        const float4 rad = st.radiance[slot];
                        st.radiance[slot] = make_float4(rad.x + add.x, rad.y + add.y, rad.z + add.z, 0.0f);
                    }
                }
                shadow = true;
                if (did_scatter) {   // integrator.h:271-316
                    nA = vadd(hi.p, vscale(S.normal_offset, hi.n));
                    float scatter_pdf_s;
                    if (m.type == 0 || m.type == 1) {   // cosine_pdf::generate pdf.h:30-33, random.h:36-44 over the table's onb
                        const float r1 = rndf(k0, k1, gb + 0);
                        const float r2 = rndf(k0, k1, gb + 1);
                        const float z = sqrtf(1 - r2);
                        float sn, cs;
                        ptm_sincos_2pi(r1, sn, cs);
                        const float x = cs * sqrtf(r2);
                        const float y = sn * sqrtf(r2);
                        nB = vadd(vadd(vscale(x, hi.ou), vscale(y, hi.ov)), vscale(z, hi.nu));
                        scatter_pdf_s = cosine_pdf_of(vdot(vunit(nB), hi.nu));
                    } else {
                        nB = material_generate(m.type, hi.n, k0, k1, gb);
                        scatter_pdf_s = material_value_of(m.type, 0.0f);
                    }
                    const float pin = (beta.y < beta.z) ? beta.z : beta.y;   // std::max(a,b) = (a<b)?b:a
                    const float p = (beta.x < pin) ? pin : beta.x;
                    bool alive = true;
                    nbeta = beta;
                    if (S.russian_roulette && p <= 1 && 0.001 < (double)p) {
                        if (rnd(k0, k1, gb + 3) > (double)p) { alive = false; ev_rr = true; }
                        else { const float ip = 1 / p; nbeta = V(beta.x * ip, beta.y * ip, beta.z * ip); }
                    }
                    if (alive) {
                        if (S.only_direct) alive = false;
                        else if ((double)scatter_pdf_s < 0.0000001) { alive = false; ev_pdf = true; }
                        else {
                            nbeta = vmul(nbeta, vdivf(vscale(fabsf(cos_i), att), scatter_pdf_s));
                            new_pdf = scatter_pdf_s;
                            if (last_bounce) { alive = false; ev_limit = true; }
                        }
                    }
                    cont = alive;
                } else {
                    // emitter: sum += beta*hit_emission a second time (integrator.h:317-323).  connect applies
                    // it AFTER this bounce's light contribution so the float additions keep the reference's order.
                    const v3 add = vmul(beta, hit_emission);
                    st.pending[slot] = make_float4(add.x, add.y, add.z, 0.0f);
                    pending = true;
                    ev_emit = true;
                }
            }
        }
        // ---- compaction: continuation rays -> next path queue, shadow records -> shadow queue.  A wave reserves its range
        // in the output segment with one ballot and one atomicAdd (lane 0, broadcast through an SGPR): no barrier, the four
        // waves of the workgroup do not wait for each other.
        n_miss += __popcll(__ballot(ev_miss)); n_hit += __popcll(__ballot(shadow)); n_rr += __popcll(__ballot(ev_rr));
        n_emit += __popcll(__ballot(ev_emit)); n_pdf += __popcll(__ballot(ev_pdf)); n_limit += __popcll(__ballot(ev_limit));
        const unsigned long long below = (1ull << lane) - 1ull;
        auto reserve = [&](bool take, int32_t *counter, int word) -> long long {
            const unsigned long long m = __ballot(take);
            int base_s = 0;
            if (lane == 0 && m) base_s = atomicAdd(counter, __popcll(m));
            base_s = __builtin_amdgcn_readfirstlane(base_s);
            wave_top[word] = max(wave_top[word], m ? base_s + (int)__popcll(m) : 0);   // scalar (chunk_limit)
            return seg_base_o + base_s + __popcll(m & below);
        };
        {
            const long long oc = reserve(cont, &qo.count[seg_o], 0);
            if (cont) {
                st4<1>(&qo.r0[oc], make_float4(nA.x, nA.y, nA.z, __int_as_float(slot)));
                st4<1>(&qo.r1[oc], make_float4(nB.x, nB.y, nB.z, new_pdf));
                st4<1>(&qo.s0[oc], make_float4(nbeta.x, nbeta.y, nbeta.z, __uint_as_float(k0)));
                st4<1>(&qo.s1[oc], make_float4(att.x, att.y, att.z, __uint_as_float(k1)));
            }
        }
        // Staged (light_samples small enough for LDS): the samples go to LDS first and a hit whose samples cannot contribute
        // -- every coefficient is (+-0, +-0, +-0) or has a NaN: connect would add +-0 or drop it (integrator.h:252-262), the
        // sum does not change by a bit -- gets no record: its shadow rays are counted, not traced (11 % of the hits of
        // cornell_box: surfaces facing away from the light, hits on the light).  Otherwise the record is reserved first and
        // the samples are stored as they are made.
        bool lit = !stage;
        long long o = stage ? 0 : reserve(shadow, &sq.count[seg_o], 1);
        const bool wave_finite = __all(!shadow || (isfinite(hp.x) && isfinite(hp.y) && isfinite(hp.z)));
        if (shadow) {
            // light sampling, integrator.h:221-243: everything up to (not including) the shadow ray's World::hit
            const bool att_ok = (double)vlen(att) > 0.0001;   // integrator.h:248
            const v3 ab = vmul(att, beta);
            // the tail every sample shares: MIS weight, attenuation * beta * weight_l / light_pdf_l * dropoff
            // (* light_emission / pick_pdf in connect), the record
            // fast_c (BoolTag): the divisions run on pt_fdiv.h's form and gmin / gmax collect what would make that inexact.  Ranges
            // (zero, and for light_pdf_l / weight_l a non-finite value, are handled by v_div_fixup and allowed where noted):
            //   light_pdf_l within [2^-45, 2^20]  =>  fp = light_pdf_l^2 within [2^-90, 2^40]: a numerator >= 2^-101, fp + g^2 (g <= 1 / pi)
            //   within [2^-90, 2^41], weight_l = fp / (fp + g^2) >= 2^-87;
            //   weight_l within [2^-40, 1] and the components of ab within [2^-60, 2^30] (once per hit; throughput decays with every
            //   coloured bounce, so the lower limit is generous)  =>  weight_l * ab >= 2^-100 and its quotient by light_pdf_l within
            //   [2^-120, 2^75].
            // The checks cost no branch and no scalar instruction: every checked magnitude, scaled so that its lower limit is 1,
            // feeds a running minimum (gmin), scaled so that its upper limit is 1 a running maximum (gmax); the loop was exact iff
            // gmin >= 1 and gmax <= 1 at its end.  Zeros trip the minimum (rare: a wave then takes the IEEE loop); NaNs and
            // infinities pass through both (ignored by min / max) and need no check: v_div_fixup gives non-finite operands the
            // IEEE result.
            float gmin = 1.0f, gmax = 1.0f;
            float litacc = 0.0f;   // largest |c.x| + |c.y| + |c.z| over the samples staged so far
            auto emit_sample = [&](auto fast_c, const uint32_t k, v3 ldir, float cos_l, float light_pdf_l) {
                constexpr bool FAST = decltype(fast_c)::value;
                const float scatter_pdf_l = material_value_of(mat_type, cos_l);   // cosine_pdf's cosine IS cos_l (pdf.h:20)
                float weight_l;
                if (FAST) {   // power_heuristic helpers.h:138-144 with the division restated
                    const float fp = light_pdf_l * light_pdf_l;
                    weight_l = fdiv(fp, fp + scatter_pdf_l * scatter_pdf_l);
                    gmin = fminf(gmin, fminf(fabsf(light_pdf_l) * 0x1p45f, weight_l * 0x1p40f));
                    gmax = fmaxf(gmax, fmaxf(fabsf(light_pdf_l) * 0x1p-20f, weight_l));
                } else weight_l = power_heuristic(light_pdf_l, scatter_pdf_l);
                const float dropoff = cos_l > 0.0f ? cos_l : 0.0f;
                v3 c = vscale(weight_l, ab);
                c = FAST ? vdivf_fast(c, light_pdf_l) : vdivf(c, light_pdf_l);
                c = vscale(dropoff, c);
                if (!att_ok) c = V(NAN, NAN, NAN);   // contribution skipped: NaN is dropped by connect like integrator.h:255
                if (stage) {
                    st_d[k * PT_BLOCK] = make_float4(ldir.x, ldir.y, ldir.z, c.x);
                    st_e[k * PT_BLOCK] = make_float2(c.y, c.z);
                    // "no component NaN and some component non-zero" <=> |c.x| + |c.y| + |c.z| > 0 (the sum is NaN iff a component is;
                    // +inf counts as non-zero): the running maximum ignores NaN sums and is positive iff some sample can contribute
                    // (round 3: six compares and five mask operations per sample)
                    litacc = fmaxf(litacc, fabsf(c.x) + fabsf(c.y) + fabsf(c.z));
                } else {
                    sq.d[(long long)k * P + o] = make_float4(ldir.x, ldir.y, ldir.z, c.x);
                    sq.e[(long long)k * P + o] = make_float2(c.y, c.z);
                }
            };
            // A rect light under a pure translation (tl = the inverse's translation), sampled from a finite point: the pdf's
            // ray (ol, ldir) is the one rect::random just built, so rect::pdf_value (primitive.h:151-166) simplifies WITHOUT
            // changing a bit:
            //  * its t = (y - o.y) / d.y divides a float by itself -- d.y = y - o.y is the same subtraction -- which is
            //    exactly 1 for a finite non-zero value and NaN otherwise (0 / 0);
            //  * dot(v, normal) has two exact-zero terms: |dot| = |v.y| for the finite v here.
            // The rect's fields may be per-lane values (the light a lane drew) or wave-uniform; the alignment is wave-uniform and
            // a compile-time constant here (plane_c: std::integral_constant-like tag): as a run-time value its shuffles were
            // eight scalar branches with register moves per sample.
            // fast_c: see emit_sample.  Its own ranges: every component of ldir within [2^-20, 2^20] (not zero) => |ldir| within
            // [2^-20, 2^22], the four quotients by it within [2^-42, 2^40]; d2 = |ldir|^2 >= 2^-40, cosine * area within [2^-62, 2^40]
            // for an area within [2^-20, 2^40], light_pdf_l within [2^-80, 2^105] (and checked against [2^-45, 2^20] afterwards).
            auto rect_light_sample = [&](auto plane_c, auto fast_c, const uint32_t k, const uint32_t kb, v3 tl, float x0, float z0, float x1, float z1, float y) {
                constexpr int plane = decltype(plane_c)::value;
                constexpr bool FAST = decltype(fast_c)::value;
                const v3 ol = V(tl.x + hp.x, tl.y + hp.y, tl.z + hp.z);
                const v3 os = shuffle(ol, plane);
                const float area = (x1 - x0) * (z1 - z0);
                // rect::random primitive.h:168-175 (first draw -> z, second -> x): z0 + r * (z1 - z0) in double with r = u * 2^-32.
                // The scaling by 2^-32 is exact wherever it is applied, so (u * 2^-32) * w and u * (w * 2^-32) round the same real
                // product: the draw stays an integer and the (loop-invariant) width carries the scale -- one f64 operation less
                const double wz = (double)(z1 - z0) * (1.0 / 4294967296.0), wx = (double)(x1 - x0) * (1.0 / 4294967296.0);
                const float pz = (float)((double)z0 + (double)stream_u32(k0, k1, kb + 1) * wz);
                const float px = (float)((double)x0 + (double)stream_u32(k0, k1, kb + 2) * wx);
                const v3 ldir = vsub(shuffle(V(px, y, pz), plane), ol);
                const float vl = vlen(ldir);
                const v3 ds = shuffle(ldir, plane);
                float cos_l, cosine_f = 0.0f;
                if (FAST) {   // unit_vector(ldir) and |ds.y| / |ldir|: four quotients over one denominator, two packed pairs
                    const float rv = fdiv_rcp(vl);
                    v3 u;
                    fdiv_q2(ldir.x, ldir.y, vl, rv, u.x, u.y);
                    fdiv_q2(ldir.z, fabsf(ds.y), vl, rv, u.z, cosine_f);
                    cos_l = vdot(u, hnu);
                    // A hit ON the light's plane (every emitter hit: 3 % of the hits, some lane of most waves) has ds.y == 0 exactly:
                    // tq is NaN then and so is every coefficient whatever the quotients are (0 / |ldir| is exact anyway), so a
                    // zero plane component does not count against the lower limit
                    const float py = (ds.y == 0.0f) ? 1.0f : fabsf(ds.y);
                    const float lo = fminf(fminf(fabsf(ds.x), py), fabsf(ds.z)), hi = fmaxf(fmaxf(fabsf(ldir.x), fabsf(ldir.y)), fabsf(ldir.z));
                    gmin = fminf(gmin, lo * 0x1p20f);
                    gmax = fmaxf(gmax, hi * 0x1p-20f);
                } else cos_l = vdot(vdivf(ldir, vl), hnu);
                const float tq = (fabsf(ds.y) > 0.0f && fabsf(ds.y) < INFINITY) ? 1.0f : NAN;   // ds.y / ds.y
                const float xh = os.x + tq * ds.x, zh = os.z + tq * ds.z;
                float light_pdf_l = 0.0f;
                if (FAST && !B0) {   // (the bounce-0 instantiation has no register to spare for it: 18 spilled VGPRs)
                    // the quotient is formed for every lane (some lane of the wave needs it anyway) and kept where the reference's
                    // "xh < x0 || xh > x1 || zh < z0 || zh > z1" is false: one maximum decided by its sign, NaN coordinates pass
                    // (ignored by the maximum, false in the reference's comparisons) -- no branch, no mask arithmetic
                    const float d2 = (tq * vl) * (tq * vl);
                    const float lp = fdiv(d2, cosine_f * area);
                    const float eo = fmaxf(fmaxf(x0 - xh, xh - x1), fmaxf(z0 - zh, zh - z1));
                    light_pdf_l = (eo > 0.0f) ? 0.0f : lp;
                } else if (!(xh < x0 || xh > x1 || zh < z0 || zh > z1)) {   // t = 1 or NaN passes 0.001 .. FLT_MAX
                    const float d2 = (tq * vl) * (tq * vl);
                    const float cosine = FAST ? cosine_f : fabsf(ds.y) / vl;
                    light_pdf_l = FAST ? fdiv(d2, cosine * area) : d2 / (cosine * area);
                }
                emit_sample(fast_c, k, ldir, cos_l, light_pdf_l);
            };
            // one light (the common case): its index is wave-uniform, so its instance/primitive records are scalar
            // loads and the pick draw (always index 0) is not needed; several lights: per-lane gather.
            // tr: the light's transform is a pure translation and every hit point of this wave is finite (same exactness
            // argument as in world_hit_n): local point = translation + p, directions are unchanged
            auto light_sample = [&](const uint32_t k, const uint32_t kb, const DInst &lin, const DPrim &lpr, const bool tr) {
                const v3 ol = tr ? V(lin.inv[3] + hp.x, lin.inv[7] + hp.y, lin.inv[11] + hp.z) : xf_point(lin.inv, hp);
                const v3 dl = prim_random(lpr, ol, k0, k1, kb + 1);                       // instance::random primitive.h:338-342
                const v3 ldir = tr ? dl : xf_linear(lin.fwd, dl);
                const float cos_l = vdot(vunit(ldir), hnu);
                const float light_pdf_l = prim_pdf_value(lpr, ol, tr ? ldir : xf_linear(lin.inv, ldir));   // primitive.h:319-337
                emit_sample(BoolTag<false>{}, k, ldir, cos_l, light_pdf_l);
            };
            if (LM == 1) {
                const DInst &lin = S.insts[S.lights[0]];
                const DPrim &lpr = S.prims[lin.prim];
                const bool tr = lin.ident && wave_finite;
                if (tr && lpr.type == 0) {
                    const DRect &lq = lpr.r[0];
                    const v3 tl = V(lin.inv[3], lin.inv[7], lin.inv[11]);
                    // one loop per alignment (wave-uniform branch); the empty asm keeps the three bodies apart
                    // the loop on the cheap divisions first; a wave in which a lane left their ranges repeats it on the IEEE ones
                    bool redo = true;
                    const float area = (lq.x1 - lq.x0) * (lq.z1 - lq.z0);   // wave-uniform
                    if (PT_SHADE_FDIV && fdiv_in_range_nz(area, -20, 40)) {
                        gmin = fminf(fminf(ab.x, ab.y), ab.z) * 0x1p60f;   // attenuation * beta: non-negative components
                        gmax = fmaxf(fmaxf(ab.x, ab.y), ab.z) * 0x1p-30f;
                        gmax = fmaxf(gmax, 1.0f); gmin = fminf(gmin, 1.0f);
                        if (lq.plane == 0) { asm volatile("; rect light xy fast"); PT_LIGHT_LOOP(0, true) }
                        else if (lq.plane == 2) { asm volatile("; rect light yz fast"); PT_LIGHT_LOOP(2, true) }
                        else { asm volatile("; rect light xz fast"); PT_LIGHT_LOOP(1, true) }
                        // a lane whose attenuation test failed stores NaN coefficients whatever the loop computed (att_ok)
                        redo = __any(att_ok && (!(gmin >= 1.0f) || !(gmax <= 1.0f)));
#ifdef PT_DBG_FDIV_NOREDO   // timing-only build: never repeat (images may differ)
                        redo = false;
#endif
#ifdef PT_DBG_FDIV_COUNT    // debugging: how often a wave repeats the loop (counted into term_pdf's word: wrong counters on purpose)
                        if (redo && lane == 0) n_pdf += 1;
#endif
                        if (redo) { lit = !stage; litacc = 0.0f; }
                    }
                    if (redo) {
                        if (lq.plane == 0) { asm volatile("; rect light xy"); PT_LIGHT_LOOP(0, false) }
                        else if (lq.plane == 2) { asm volatile("; rect light yz"); PT_LIGHT_LOOP(2, false) }
                        else { asm volatile("; rect light xz"); PT_LIGHT_LOOP(1, false) }
                    }
                } else {
                    for (uint32_t k = 0; k < L; k++) light_sample(k, base + NV + k * (3u + NV), lin, lpr, tr);
                }
            } else if (LM == 2) {
                // two lights (BASELINE config 3): both records come in by scalar loads and every lane selects the fields of the
                // one it drew -- a few dozen v_cndmask instead of per-lane table gathers
                const DInst &ia = S.insts[S.lights[0]], &ib = S.insts[S.lights[1]];
                const DPrim &pa = S.prims[ia.prim], &pb = S.prims[ib.prim];
                // both rects of one alignment under pure translations (cornell_box_small_lights): the shortcut above with the
                // drawn light's eight fields selected per lane
                const bool both_tr = ia.ident && ib.ident && wave_finite && pa.type == 0 && pb.type == 0 && pa.r[0].plane == pb.r[0].plane;
                if (both_tr) {
                    const DRect &qa = pa.r[0], &qb = pb.r[0];
                    // the drawn light's fields per lane, then the one-light loop's body; on the cheap divisions first (see LM == 1)
                    auto both_loop = [&](auto fast_c) {
                        for (uint32_t k = 0; k < L; k++) {
                            const uint32_t kb = base + NV + k * (3u + NV);
                            const bool second = (int)(rnd(k0, k1, kb + 0) * 2.0) != 0;            // world.h:31-35
                            const v3 tl = V(second ? ib.inv[3] : ia.inv[3], second ? ib.inv[7] : ia.inv[7], second ? ib.inv[11] : ia.inv[11]);
                            const float x0 = second ? qb.x0 : qa.x0, z0 = second ? qb.z0 : qa.z0, x1 = second ? qb.x1 : qa.x1, z1 = second ? qb.z1 : qa.z1;
                            const float y = second ? qb.y : qa.y;
                            if (qa.plane == 0) { asm volatile("; rect lights xy"); rect_light_sample(PlaneTag<0>{}, fast_c, k, kb, tl, x0, z0, x1, z1, y); }
                            else if (qa.plane == 2) { asm volatile("; rect lights yz"); rect_light_sample(PlaneTag<2>{}, fast_c, k, kb, tl, x0, z0, x1, z1, y); }
                            else { asm volatile("; rect lights xz"); rect_light_sample(PlaneTag<1>{}, fast_c, k, kb, tl, x0, z0, x1, z1, y); }
                        }
                    };
                    const float area_a = (qa.x1 - qa.x0) * (qa.z1 - qa.z0), area_b = (qb.x1 - qb.x0) * (qb.z1 - qb.z0);   // wave-uniform
                    bool redo = true;
                    if (PT_SHADE_FDIV && fdiv_in_range_nz(area_a, -20, 40) && fdiv_in_range_nz(area_b, -20, 40)) {
                        gmin = fminf(fminf(fminf(ab.x, ab.y), ab.z) * 0x1p60f, 1.0f);
                        gmax = fmaxf(fmaxf(fmaxf(ab.x, ab.y), ab.z) * 0x1p-30f, 1.0f);
                        both_loop(BoolTag<true>{});
                        redo = __any(att_ok && (!(gmin >= 1.0f) || !(gmax <= 1.0f)));
                        if (redo) { lit = !stage; litacc = 0.0f; }
                    }
                    if (redo) both_loop(BoolTag<false>{});
                } else
                for (uint32_t k = 0; k < L; k++) {
                    const uint32_t kb = base + NV + k * (3u + NV);
                    const bool second = (int)(rnd(k0, k1, kb + 0) * 2.0) != 0;            // world.h:31-35
                    DInst lin;
                    DPrim lpr;
#pragma unroll
                    for (int i = 0; i < 12; i++) { lin.inv[i] = second ? ib.inv[i] : ia.inv[i]; lin.fwd[i] = second ? ib.fwd[i] : ia.fwd[i]; }
                    lpr.type = second ? pb.type : pa.type;
                    lpr.cx = second ? pb.cx : pa.cx; lpr.cy = second ? pb.cy : pa.cy; lpr.cz = second ? pb.cz : pa.cz;
                    lpr.radius = second ? pb.radius : pa.radius;
                    lpr.r[0].x0 = second ? pb.r[0].x0 : pa.r[0].x0; lpr.r[0].z0 = second ? pb.r[0].z0 : pa.r[0].z0;
                    lpr.r[0].x1 = second ? pb.r[0].x1 : pa.r[0].x1; lpr.r[0].z1 = second ? pb.r[0].z1 : pa.r[0].z1;
                    lpr.r[0].y = second ? pb.r[0].y : pa.r[0].y; lpr.r[0].plane = second ? pb.r[0].plane : pa.r[0].plane;
                    lpr.r[0].ny = second ? pb.r[0].ny : pa.r[0].ny;
                    light_sample(k, kb, lin, lpr, false);
                }
            } else {
                for (uint32_t k = 0; k < L; k++) {
                    const uint32_t kb = base + NV + k * (3u + NV);
                    const int idx = (int)(rnd(k0, k1, kb + 0) * (double)S.n_lights);   // world.h:31-35
                    const DInst &lin = S.insts[S.lights[idx]];
                    light_sample(k, kb, lin, S.prims[lin.prim], false);
                }
            }
            if (stage) lit = litacc > 0.0f;
        }
        if (stage) {
            o = reserve(shadow && lit, &sq.count[seg_o], 1);
            if (shadow && lit) {
                for (uint32_t k = 0; k < L; k++) {
                    st4<1>(&sq.d[(long long)k * P + o], st_d[k * PT_BLOCK]);
                    st2<1>(&sq.e[(long long)k * P + o], st_e[k * PT_BLOCK]);
                }
            }
            const bool dark = shadow && !lit;
            n_dark += __popcll(__ballot(dark));
            if (dark && pending) {   // no record for connect to add the second emitter addition after: nothing comes between
                const float4 pe = st.pending[slot];
                if (B0) rad0 = make_float4(rad0.x + pe.x, rad0.y + pe.y, rad0.z + pe.z, 0.0f);
                else {
                    const float4 rad = st.radiance[slot];
                    st.radiance[slot] = make_float4(rad.x + pe.x, rad.y + pe.y, rad.z + pe.z, 0.0f);
                }
            }
        }
        if (B0 && valid) st4<2>(&st.radiance[slot], rad0);   // the one radiance store of a camera sample at bounce 0 (coalesced: slot = position)
        if (shadow && lit) {
            st4<1>(&sq.p0[o], make_float4(hp.x, hp.y, hp.z, __int_as_float(slot | (pending ? (int)0x80000000 : 0))));
            if (NV) sq.key[o] = make_uint2(k0, k1);
        }
    }
    if (lane == 0) {
        if (qmax_out) {   // no result awaited
            if (wave_top[0]) atomicMax(qmax_out, wave_top[0]);
            if (wave_top[1]) atomicMax(qmax_out + PT_QMAX_BANKS, wave_top[1]);
        }
        DCounters *cb = counter_bank(st.counters);
        if (n_hit) atomicAdd(&cb->ext_hits, (unsigned long long)n_hit);
        if (n_miss) atomicAdd(&cb->term_miss, (unsigned long long)n_miss);
        if (n_rr) atomicAdd(&cb->term_rr, (unsigned long long)n_rr);
        if (n_emit) atomicAdd(&cb->term_emitter, (unsigned long long)n_emit);
        if (n_pdf) atomicAdd(&cb->term_pdf, (unsigned long long)n_pdf);
        if (n_limit) atomicAdd(&cb->term_bounce_limit, (unsigned long long)n_limit);
        if (n_dark) {   // the light_samples shadow rays of every hit without a record: counted like the reference counts them
            // (integrator.h:246-247) and, separately, as NOT traced (pt_counters::rays_traced leaves them out)
            atomicAdd(&cb->rays, (unsigned long long)n_dark * L);
            atomicAdd(&cb->shadow_rays, (unsigned long long)n_dark * L);
            atomicAdd(&cb->shadow_untraced, (unsigned long long)n_dark * L);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// connect: the light_samples shadow rays of one hit (integrator.h:244-268): closest hit, emitted() of
// whatever was hit, NaN contributions dropped, sum += light_contribution / light_samples,
// then the deferred second emitter addition.
// ------------------------------------------------------------------------------------------------
// emitted() of whatever a shadow ray hit, times the stored coefficient (integrator.h:252-262).  The emitted radiance
// comes from one table load by hit id; non-emitters contribute (coef*0)/pick = +-0 or NaN, i.e. nothing.
// v / (float)n for a small positive count n (lights, light samples).  For n a power of two the quotient is v * (1 / n)
// bit for bit -- both are the correctly rounded value of the same real number, 1 / n is exact -- which spares three IEEE
// divisions per call (one light, four light samples: every BASELINE configuration).
struct CountDiv { float d, inv; bool pow2; };
DEVI CountDiv count_div(int n) { CountDiv c; c.d = (float)n; c.inv = 1.0f / (float)n; c.pow2 = n > 0 && (n & (n - 1)) == 0; return c; }
DEVI v3 vdiv_count(v3 v, const CountDiv &c) { return c.pow2 ? vscale(c.inv, v) : vdivf(v, c.d); }
template <bool TEX>
DEVI void connect_contribution(const DScene &S, v3 hp, v3 ldir, float t, int id, v3 coef, const CountDiv &pick_pdf, v3 &lc)
{
    if (id < 0) return;
    const float4 e = S.emit[id];
    v3 le = V(e.x, e.y, e.z);
    if (e.w != 0.0f) {   // one-sided and / or textured light: needs the hit record (material.h:214-228)
        const DMat m = S.mats[S.prims[S.insts[id >> 3].prim].hit_mat[id & 7]];
        le = material_emitted<TEX>(S, m, id, finalize_hit(S, hp, ldir, t, id, true), ldir);
    }
    if (le.x == 0.0f && le.y == 0.0f && le.z == 0.0f) return;
    v3 c = vmul(coef, le);
    c = vdiv_count(c, pick_pdf);
    if (!v_is_nan(c)) lc = vadd(lc, c);
}

// NR rays of one hit (they share their origin) are traversed together; light_samples is a multiple of NR.
template <int NR, bool TEX, bool GA, bool WALK>
__global__ __launch_bounds__(PT_BLOCK, NR >= 4 ? 4 : (GA ? PT_CONNECT_WAVES_GA : PT_CONNECT_WAVES)) void k_connect(DScene S, const DOp *__restrict__ t_ops, const DInst *__restrict__ t_insts, const DPrim *__restrict__ t_prims,
        const DMat *__restrict__ t_mats, const int32_t *__restrict__ t_lights, const float4 *__restrict__ t_emit, DStreams st, DBatch b, int bounce)
{
    // scene tables arrive as separate __restrict__ kernel arguments: only then can hipcc prove that the stream
    // stores below never clobber them and keep the wave-uniform table reads on the scalar unit (s_load)
    S.ops = t_ops; S.insts = t_insts; S.prims = t_prims; S.mats = t_mats; S.lights = t_lights; S.emit = t_emit;
    extern __shared__ float2 stack[];   // [stack_depth][max(NR,1)][PT_BLOCK]
    const Stk stk = stack_of(st, stack);
    // the shadow queue was written by this bounce's shade in the OUTPUT segmentation
    const int cps = b.seg_cap_out / PT_BLOCK;
    const int total_chunks = b.n_seg_out * cps;
    const DShadowQueue sq = st.sq;
    const long long P = b.P;
    const uint32_t L = (uint32_t)S.light_samples, NV = (uint32_t)S.n_vol;
    const uint32_t D = NV + L * (3u + NV) + 4u;
    const uint32_t base = DIM_BOUNCE0 + (uint32_t)bounce * D;
    const CountDiv pick_pdf = count_div(S.n_lights);   // integrator.h:224
    const CountDiv n_samples = count_div(S.light_samples);
    unsigned long long n_rays = 0;
    ChunkWalk nx;   // see k_extend
    nx.init(b.n_seg_out, b.perm_out);
    const int end_chunks = chunk_limit(st.qmax, 1, bounce, b.n_seg_out, total_chunks);   // see chunk_limit
    int n_ahead = ((int)blockIdx.x < end_chunks) ? sq.count[nx.seg] : 0;   // one chunk ahead, see k_extend
    for (int c = blockIdx.x; c < end_chunks; c += gridDim.x) {   // persistent workgroups, chunk-major, see k_extend
        const int chunk = nx.chunk, seg = nx.seg;
        const int n = n_ahead;
        nx.advance();
        if (c + (int)gridDim.x < end_chunks) n_ahead = sq.count[nx.seg];
        const int i0 = chunk * PT_BLOCK;
        if (i0 >= n) continue;
        const long long seg_base = (long long)seg * b.seg_cap_out;
        const int i = i0 + threadIdx.x;
        const bool valid = i < n;
        n_rays += (unsigned long long)((n - i0) < PT_BLOCK ? (n - i0) : PT_BLOCK) * L;
        if ((i0 + (int)(threadIdx.x & ~63u)) >= n) continue;
        const long long pos = seg_base + (valid ? i : i0);
        // Memory round trips of a chunk, issued so that they overlap the sweeps instead of preceding them one by one (a wave of
        // this kernel was parked on s_waitcnt 41 % of its life): the records of the first group of rays and the hit point go
        // out together; the NEXT group's records are requested before the current group is traced; the radiance the result
        // is added to is requested as soon as the slot is known, two sweeps before it is needed.
        constexpr int R = NR > 0 ? NR : 1;
        float4 dn[R];
        float2 en[R];
#pragma unroll
        for (int k = 0; k < R; k++) { dn[k] = ld4<4>(&sq.d[(long long)k * P + pos]); en[k] = ld2<4>(&sq.e[(long long)k * P + pos]); }
        const float4 p0 = ld4<4>(&sq.p0[pos]);
        const v3 hp = V(p0.x, p0.y, p0.z);
        const int slotw = __float_as_int(p0.w);
        const int slot = slotw & 0x7fffffff;
        uint32_t k0 = 0, k1 = 0;
        if (NV) { const uint2 kk = sq.key[pos]; k0 = kk.x; k1 = kk.y; }
#if PT_CONNECT_PREFETCH >= 1
        const float4 rad = st.radiance[valid ? slot : 0];   // no other lane of this launch touches the slot: each hit has one record
#endif
        v3 lc = V(0.0f, 0.0f, 0.0f);
        for (uint32_t kg = 0; kg < L; kg += R) {   // light_samples is a multiple of R (launch_connect)
            v3 ldir[R], coef[R];
            uint32_t vd[R];
            float t[R];
            int id[R];
#pragma unroll
            for (int k = 0; k < R; k++) {
                ldir[k] = V(dn[k].x, dn[k].y, dn[k].z);
                coef[k] = V(dn[k].w, en[k].x, en[k].y);
                vd[k] = base + NV + (kg + (uint32_t)k) * (3u + NV) + 3u;
            }
#if PT_CONNECT_PREFETCH >= 2
            if (kg + R < L) {
#pragma unroll
                for (int k = 0; k < R; k++) { dn[k] = ld4<4>(&sq.d[(long long)(kg + R + k) * P + pos]); en[k] = ld2<4>(&sq.e[(long long)(kg + R + k) * P + pos]); }
            }
#endif
#if PT_CONNECT_NOHOIST   // the origin is re-read through an opaque move per group of rays: its per-leaf terms are formed again, not kept
            v3 hq = hp;
            asm volatile("" : "+v"(hq.x), "+v"(hq.y), "+v"(hq.z));
#else
            const v3 hq = hp;
#endif
#ifdef PT_DBG_CONNECT_NOSWEEP   // timing only (wrong images; PATHTRACE_HIP_SPEC_FLAGS=-DPT_DBG_CONNECT_NOSWEEP): k_connect without its
            // sweeps -- records, contribution, radiance -- measured 4.3 of 11.4 ms on cornell_box (profiles/experiments/r05_ab_runs.json r05_nosweep)
#pragma unroll
            for (int k = 0; k < R; k++) { t[k] = ldir[k].x + hp.x; id[k] = (t[k] == 12345.0f) ? 8 : -1; }
#else
            world_hit<R, GA, WALK, false, true>(S, valid, hq, ldir, k0, k1, vd, stk, t, id);   // shadow rays: see world_hit_fast_rb SHADOW
#endif
            if (valid) {
#pragma unroll
                for (int k = 0; k < R; k++) connect_contribution<TEX>(S, hp, ldir[k], t[k], id[k], coef[k], pick_pdf, lc);   // k order = integrator.h:221
            }
#if PT_CONNECT_PREFETCH < 2
            if (kg + R < L) {
#pragma unroll
                for (int k = 0; k < R; k++) { dn[k] = ld4<4>(&sq.d[(long long)(kg + R + k) * P + pos]); en[k] = ld2<4>(&sq.e[(long long)(kg + R + k) * P + pos]); }
            }
#endif
        }
        if (valid) {
#if PT_CONNECT_PREFETCH < 1
            const float4 rad = st.radiance[slot];
#endif
            v3 r = vadd(V(rad.x, rad.y, rad.z), vdiv_count(lc, n_samples));   // integrator.h:268
            if (slotw < 0) { const float4 pe = st.pending[slot]; r = vadd(r, V(pe.x, pe.y, pe.z)); }
            // every shadow ray blocked (or every coefficient +-0): the sum has the bits it had, the store is skipped
            if (!PT_SKIP_NOOP_RADIANCE || __float_as_int(r.x) != __float_as_int(rad.x) || __float_as_int(r.y) != __float_as_int(rad.y) || __float_as_int(r.z) != __float_as_int(rad.z))
                st.radiance[slot] = make_float4(r.x, r.y, r.z, 0.0f);
        }
    }
    if (threadIdx.x == 0 && n_rays) {
        atomicAdd(&counter_bank(st.counters)->rays, n_rays);
        atomicAdd(&counter_bank(st.counters)->shadow_rays, n_rays);
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate: framebuffer[j][i] += de_nan(col), samples in increasing order (renderer.h:670-682)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PT_BLOCK) void k_accumulate(DScene S, DStreams st, DBatch b)
{
    const int npix = b.npix;
    for (int pl = blockIdx.x * PT_BLOCK + threadIdx.x; pl < npix; pl += gridDim.x * PT_BLOCK) {
        int pi, pj;
        batch_pixel(b, pl, pi, pj);
        const long long fi = (long long)pj * S.width + pi;
        float4 acc = st.fb[fi];
        for (int s = 0; s < b.ns; s++) {
            const float4 r = st.radiance[(long long)s * npix + pl];
            acc.x += is_nanf(r.x) ? 0.0f : r.x;
            acc.y += is_nanf(r.y) ? 0.0f : r.y;
            acc.z += is_nanf(r.z) ? 0.0f : r.z;
        }
        st.fb[fi] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// trace: World::hit for caller-supplied rays (pt_trace_rays), same traversal as k_extend / k_connect
// ------------------------------------------------------------------------------------------------
template <int NR, bool GA, bool WALK>
__global__ __launch_bounds__(PT_BLOCK) void k_trace(DScene S, const DOp *__restrict__ t_ops, DStreams st, long long n, const float *__restrict__ org,
                                                    const float *__restrict__ dir, uint32_t k0, uint32_t k1, uint32_t vol_dim,
                                                    float *t_out, int *id_out)
{
    S.ops = t_ops;
    extern __shared__ float2 stack[];
    const Stk stk = stack_of(st, stack);
    for (long long base = (long long)blockIdx.x * PT_BLOCK; base < n; base += (long long)gridDim.x * PT_BLOCK) {   // bounded grid: the
        const long long i = base + threadIdx.x;                                                                   // global stack is per thread
        const bool valid = i < n;
        const long long j = valid ? i : 0;
        const v3 A = V(org[3 * j], org[3 * j + 1], org[3 * j + 2]);
        v3 B[NR];
        uint32_t vd[NR];
        float t[NR];
        int id[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            B[r] = V(dir[3 * (j * NR + r)], dir[3 * (j * NR + r) + 1], dir[3 * (j * NR + r) + 2]);
            vd[r] = vol_dim + (uint32_t)r * 16u;
        }
        world_hit<NR, GA, WALK>(S, valid, A, B, k0, k1, vd, stk, t, id);
        if (valid) {
#pragma unroll
            for (int r = 0; r < NR; r++) { t_out[j * NR + r] = t[r]; id_out[j * NR + r] = id[r]; }
        }
    }
}

#ifndef PT_SPEC_BUILD   // the per-scene module holds kernels only (pt_spec.cpp launches them through the module API)
// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------
// at most 4096 workgroups per launch (256 CUs x 8 resident x 2: measured better balanced than 2048), few enough to dispatch quickly
#ifndef PT_GRID_MAX
#define PT_GRID_MAX 4096   // (-DPT_GRID_MAX=... for measurements: 2048 is less balanced, 8192 dispatches more slowly)
#endif
static int persistent_grid(long long chunks) { return (int)(chunks < PT_GRID_MAX ? chunks : PT_GRID_MAX); }
void launch_generate(const DScene &S, const DStreams &st, const DBatch &b, hipStream_t s)
{
    hipLaunchKernelGGL(k_generate, dim3(b.n_seg), dim3(PT_BLOCK), 0, s, S, st, b);
}
void launch_extend(const DScene &S, const DStreams &st, const DBatch &b, int qi, int bounce, hipStream_t s, SpecJob *spec)
{
    const size_t lds = st.gstack ? 0 : (size_t)S.stack_depth * PT_BLOCK * sizeof(float2);
    const dim3 grid(persistent_grid(b.n_seg * (b.seg_cap / PT_BLOCK))), block(PT_BLOCK);
    // the scene's own build of the sweep when it is ready (pt_spec.cpp), else -- and on any launch error -- the generic kernel
    if (spec && !S.walk && spec_launch_extend(spec, PT_FUSE_GENERATE && bounce == 0, (int)grid.x, lds, s, S, st, b, qi, bounce) == 0) return;
#define PT_LAUNCH_EXTEND_B(GA, WALK, B0) hipLaunchKernelGGL((k_extend<GA, WALK, B0>), grid, block, lds, s, S, S.ops, S.insts, S.prims, S.mats, S.lights, S.emit, st, b, qi, bounce)
#define PT_LAUNCH_EXTEND(GA, WALK) do { if (PT_FUSE_GENERATE && bounce == 0) PT_LAUNCH_EXTEND_B(GA, WALK, true); else PT_LAUNCH_EXTEND_B(GA, WALK, false); } while (0)
    if (S.walk) PT_LAUNCH_EXTEND(true, true);
    else if (S.geom_all) PT_LAUNCH_EXTEND(true, false);
    else PT_LAUNCH_EXTEND(false, false);
#undef PT_LAUNCH_EXTEND_B
#undef PT_LAUNCH_EXTEND
}
void launch_shade(const DScene &S, const DStreams &st, const DBatch &b, int qi, int bounce, hipStream_t s)
{
    const dim3 grid(persistent_grid(b.n_seg * (b.seg_cap / PT_BLOCK))), block(PT_BLOCK);
    const size_t lds = b.stage_shadow ? (size_t)S.light_samples * PT_BLOCK * (sizeof(float4) + sizeof(float2)) : 0;
#define PT_LAUNCH_SHADE_B(TEX, LM, B0)                                                                                                   \
    do {                                                                                                                                \
        if (b.stage_shadow) hipLaunchKernelGGL((k_shade<TEX, LM, true, B0>), grid, block, lds, s, S, S.ops, S.insts, S.prims, S.mats, S.lights, S.emit, st, b, qi, bounce); \
        else hipLaunchKernelGGL((k_shade<TEX, LM, false, B0>), grid, block, 0, s, S, S.ops, S.insts, S.prims, S.mats, S.lights, S.emit, st, b, qi, bounce);                 \
    } while (0)
#define PT_LAUNCH_SHADE(TEX, LM) do { if (PT_FUSE_GENERATE && bounce == 0) PT_LAUNCH_SHADE_B(TEX, LM, true); else PT_LAUNCH_SHADE_B(TEX, LM, false); } while (0)
    const int lm = S.n_lights == 1 ? 1 : (S.n_lights == 2 ? 2 : 0);   // cornell_box_small_lights 1080p: 2 = 21.1, 0 = 20.8 Grays/s
    if (S.textured) { if (lm == 1) PT_LAUNCH_SHADE(true, 1); else if (lm == 2) PT_LAUNCH_SHADE(true, 2); else PT_LAUNCH_SHADE(true, 0); }
    else { if (lm == 1) PT_LAUNCH_SHADE(false, 1); else if (lm == 2) PT_LAUNCH_SHADE(false, 2); else PT_LAUNCH_SHADE(false, 0); }
#undef PT_LAUNCH_SHADE
#undef PT_LAUNCH_SHADE_B
}
void launch_connect(const DScene &S, const DStreams &st, const DBatch &b, int bounce, hipStream_t s, SpecJob *spec)
{
    const int L = S.light_samples;
    // rays of one hit traversed together: 2 when light_samples is even, else 1.  Measured on cornell_box 1080p: 2 rays
    // (80 VGPRs, 6 waves/SIMD) beat 4 rays (112 VGPRs, 4 waves/SIMD: more sharing, less latency hiding) and 1 ray.
    int nr = (L % 2 == 0) ? 2 : 1;
    if (S.walk) nr = 1;   // the walk takes its rays one at a time
    // The module's k_connect (straight-line sweep, 5 waves per SIMD) is worth 3 - 4 % of this kernel -- two rays per sweep already
    // share the op fetch of the generic loop.  PATHTRACE_HIP_SPEC=...,extend-only keeps shadow rays on the generic kernel (the A/B).
    const char *sc_env = getenv("PATHTRACE_HIP_SPEC");   // read per launch: the tests switch it
    const bool spec_connect = !(sc_env && strstr(sc_env, "extend-only"));
    if (!spec_connect) spec = nullptr;
    if (spec && !S.walk && L % spec_connect_nr(spec) == 0) nr = spec_connect_nr(spec);
    const size_t lds = st.gstack ? 0 : (size_t)S.stack_depth * (nr ? nr : 1) * PT_BLOCK * sizeof(float2);
    const dim3 grid(persistent_grid(b.n_seg_out * (b.seg_cap_out / PT_BLOCK))), block(PT_BLOCK);
    if (spec && !S.walk && nr == spec_connect_nr(spec) && spec_launch_connect(spec, (int)grid.x, lds, s, S, st, b, bounce) == 0) return;
#define PT_LAUNCH_CONNECT(NR, TEX, GA, WALK) hipLaunchKernelGGL((k_connect<NR, TEX, GA, WALK>), grid, block, lds, s, S, S.ops, S.insts, S.prims, S.mats, S.lights, S.emit, st, b, bounce)
#define PT_LAUNCH_CONNECT_NR(TEX, GA) { if (nr == 4) PT_LAUNCH_CONNECT(4, TEX, GA, false); else if (nr == 2) PT_LAUNCH_CONNECT(2, TEX, GA, false); else PT_LAUNCH_CONNECT(1, TEX, GA, false); }
    if (S.walk) { if (S.textured) PT_LAUNCH_CONNECT(1, true, true, true); else PT_LAUNCH_CONNECT(1, false, true, true); }
    else if (S.textured) { if (S.geom_all) PT_LAUNCH_CONNECT_NR(true, true) else PT_LAUNCH_CONNECT_NR(true, false) }
    else { if (S.geom_all) PT_LAUNCH_CONNECT_NR(false, true) else PT_LAUNCH_CONNECT_NR(false, false) }
#undef PT_LAUNCH_CONNECT_NR
#undef PT_LAUNCH_CONNECT
}
void launch_trace(const DScene &S, const DStreams &st, long long n, int nr, const float *org, const float *dir, uint32_t k0, uint32_t k1,
                  uint32_t vol_dim, float *t_out, int *id_out, hipStream_t s, SpecJob *spec)
{
    const int blocks = persistent_grid((n + PT_BLOCK - 1) / PT_BLOCK);
    const size_t lds = st.gstack ? 0 : (size_t)S.stack_depth * nr * PT_BLOCK * sizeof(float2);
    if (spec && !S.walk && spec_launch_trace(spec, nr, blocks, lds, s, S, st, n, org, dir, k0, k1, vol_dim, t_out, id_out) == 0) return;
#define PT_LAUNCH_TRACE(NR, GA, WALK) hipLaunchKernelGGL((k_trace<NR, GA, WALK>), dim3(blocks), dim3(PT_BLOCK), lds, s, S, S.ops, st, n, org, dir, k0, k1, vol_dim, t_out, id_out)
    if (S.walk) { if (nr == 4) PT_LAUNCH_TRACE(4, true, true); else if (nr == 2) PT_LAUNCH_TRACE(2, true, true); else PT_LAUNCH_TRACE(1, true, true); }
    else if (S.geom_all) { if (nr == 4) PT_LAUNCH_TRACE(4, true, false); else if (nr == 2) PT_LAUNCH_TRACE(2, true, false); else PT_LAUNCH_TRACE(1, true, false); }
    else { if (nr == 4) PT_LAUNCH_TRACE(4, false, false); else if (nr == 2) PT_LAUNCH_TRACE(2, false, false); else PT_LAUNCH_TRACE(1, false, false); }
#undef PT_LAUNCH_TRACE
}
int launch_grid_max() { return persistent_grid(1ll << 40); }

// ------------------------------------------------------------------------------------------------
// tile-cost planner (pt_measure_tile_costs): after k_shade of every bounce, the World::hit queries this bounce PERFORMS are
// added to the cost word of the image tile the path's pixel lies in -- one per live path (its extension ray, traced by
// k_extend) and light_samples per shadow record k_shade wrote (the rays k_connect is about to trace; hits that got no
// record cost no shadow rays).  One launch over the whole frame yields the traced-ray count of every tile; the render
// kernels are untouched (this kernel only runs while a planner call has set the cost table).
// ------------------------------------------------------------------------------------------------
DEVI int tile_of_slot(const DBatch &b, int slot)
{
    const int pl = slot % b.npix;
    if (b.n_tiles <= 1) return 0;
    int lo = 0, hi = b.n_tiles - 1;   // last tile with pix0 <= pl (batch_pixel's search)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.tiles[mid].pix0 <= pl) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(PT_BLOCK) void k_tally(DStreams st, DBatch b, int qi, int bounce, int light_samples, unsigned long long *__restrict__ cost)
{
    // What a tile costs, in the reference's own unit (pt_counters::rays, integrator.h:192, 246-247): one extension ray per path this
    // bounce extended and light_samples shadow rays per HIT -- also for the hits whose samples cannot contribute, which get no shadow
    // record here (their rays are counted, not traced): a tile's time follows its hits (k_shade works on every one of them), not
    // the records that survive.  Measured (round 5, eight ranks' tile lists on one GPU): rank time against this count +-0.1 ms of
    // 16, against the traced rays +-0.3.  Runs behind k_shade, in front of k_connect: st.hit holds this bounce's hit ids.
    const DQueue q = st.q[qi];
    const int cps = b.seg_cap / PT_BLOCK, total_chunks = b.n_seg * cps;
    // the kernels' own bound (chunk_limit): no segment of the path queue holds more than live_p entries -- a count word beyond it
    // is not trusted (k_extend zeroes the words of every bounce, also of a batch whose paths have all died)
    const int live_p = (PT_FUSE_GENERATE && bounce == 0) ? cps : chunk_limit(st.qmax, 0, bounce - 1, 1, cps);
    for (int c = blockIdx.x; c < total_chunks; c += gridDim.x) {           // the paths this bounce extended
        const int seg = c / cps, i = (c - seg * cps) * PT_BLOCK + (int)threadIdx.x;
        if (c - seg * cps >= live_p) continue;
        if (i >= ((PT_FUSE_GENERATE && bounce == 0) ? seg_live<true>(q, b, seg) : seg_live<false>(q, b, seg))) continue;
        const long long pos = (long long)seg * b.seg_cap + i;
        const int slot = (PT_FUSE_GENERATE && bounce == 0) ? (int)pos : __float_as_int(q.r0[pos].w);
        const bool hit = __float_as_int(st.hit[pos].y) >= 0;
        atomicAdd(cost + tile_of_slot(b, slot), hit ? 1ull + (unsigned long long)light_samples : 1ull);
    }
}
void launch_tally(const DScene &S, const DStreams &st, const DBatch &b, int qi, int bounce, unsigned long long *cost, hipStream_t s)
{
    const dim3 grid(persistent_grid(b.n_seg * (b.seg_cap / PT_BLOCK))), block(PT_BLOCK);
    hipLaunchKernelGGL(k_tally, grid, block, 0, s, st, b, qi, bounce, S.light_samples, cost);
}
int launch_fuses_generate() { return PT_FUSE_GENERATE; }
int launch_qmax_words() { return 2 * PT_QMAX_BOUNCES * PT_QMAX_BANKS; }

// ------------------------------------------------------------------------------------------------
// multi-GPU exchange (pt_multi.cpp): a device sends only the pixels of the tiles it owns.  rects = n x (x0, y0, x1, y1),
// pix0[k] = packed index of rect k's first pixel (pix0[n] = total), row-major inside a rect.
// ------------------------------------------------------------------------------------------------
DEVI long long packed_to_film(const int4 *__restrict__ rects, const int *__restrict__ pix0, int n, int width, int p)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pix0[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const int4 r = rects[lo];
    const int w = r.z - r.x, l = p - pix0[lo], py = l / w;
    return (long long)(r.y + py) * width + (r.x + (l - py * w));
}
__global__ __launch_bounds__(PT_BLOCK) void k_pack_tiles(float4 *__restrict__ packed, const float4 *__restrict__ fb, const int4 *__restrict__ rects,
                                                         const int *__restrict__ pix0, int n, int width, int total)
{
    for (int p = blockIdx.x * PT_BLOCK + threadIdx.x; p < total; p += gridDim.x * PT_BLOCK) packed[p] = fb[packed_to_film(rects, pix0, n, width, p)];
}
// dst += packed over the same rect list: disjoint ownership, so every film pixel receives one non-zero term in all
__global__ __launch_bounds__(PT_BLOCK) void k_unpack_add_tiles(float4 *__restrict__ fb, const float4 *__restrict__ packed, const int4 *__restrict__ rects,
                                                               const int *__restrict__ pix0, int n, int width, int total)
{
    for (int p = blockIdx.x * PT_BLOCK + threadIdx.x; p < total; p += gridDim.x * PT_BLOCK) {
        const long long f = packed_to_film(rects, pix0, n, width, p);
        const float4 a = fb[f], v = packed[p];
        fb[f] = make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w);
    }
}
void launch_pack_tiles(void *packed, const void *fb, const void *rects, const void *pix0, int n, int width, int total, hipStream_t s)
{
    if (total < 1) return;
    const int blocks = (int)((total + PT_BLOCK - 1) / PT_BLOCK < 4096 ? (total + PT_BLOCK - 1) / PT_BLOCK : 4096);
    hipLaunchKernelGGL(k_pack_tiles, dim3(blocks), dim3(PT_BLOCK), 0, s, (float4 *)packed, (const float4 *)fb, (const int4 *)rects, (const int *)pix0, n, width, total);
}
void launch_unpack_add_tiles(void *fb, const void *packed, const void *rects, const void *pix0, int n, int width, int total, hipStream_t s)
{
    if (total < 1) return;
    const int blocks = (int)((total + PT_BLOCK - 1) / PT_BLOCK < 4096 ? (total + PT_BLOCK - 1) / PT_BLOCK : 4096);
    hipLaunchKernelGGL(k_unpack_add_tiles, dim3(blocks), dim3(PT_BLOCK), 0, s, (float4 *)fb, (const float4 *)packed, (const int4 *)rects, (const int *)pix0, n, width, total);
}
// multi-GPU reduce on the root device (pt_multi.cpp): dst += src over RGBA framebuffers whose tiles are disjoint
__global__ __launch_bounds__(PT_BLOCK) void k_add_fb(float4 *__restrict__ dst, const float4 *__restrict__ src, long long n)
{
    for (long long i = (long long)blockIdx.x * PT_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * PT_BLOCK) {
        const float4 a = dst[i], b = src[i];
        dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
void launch_add_fb(void *dst_rgba, const void *src_rgba, long long n_pixels, hipStream_t s)
{
    long long blocks = (n_pixels + PT_BLOCK - 1) / PT_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_add_fb, dim3((int)blocks), dim3(PT_BLOCK), 0, s, (float4 *)dst_rgba, (const float4 *)src_rgba, n_pixels);
}
void launch_accumulate(const DScene &S, const DStreams &st, const DBatch &b, hipStream_t s)
{
    int npix = b.npix;
    int blocks = (npix + PT_BLOCK - 1) / PT_BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_accumulate, dim3(blocks), dim3(PT_BLOCK), 0, s, S, st, b);
}

#endif  // PT_SPEC_BUILD

}  // namespace ptd
