// Correctly rounded f32 division without the range scaling of the compiler's sequence.
//
// hipcc expands an IEEE `n / d` into  v_div_scale x2, v_rcp, 4 fma, mul, v_div_fmas, v_div_fixup  (about 54 issue
// cycles per wave64 on MI355X).  v_div_scale / v_div_fmas only move the operands into a range where no
// intermediate value under- or overflows (multiplications by powers of two, exact) and v_div_fixup patches the special
// operands (zeros, infinities, NaNs) with the right sign.  When no intermediate can leave the normal range the scaling is
// the identity and the arithmetic in between -- one Newton step on the hardware reciprocal, two residual corrections of
// the quotient -- is exactly what is restated here, instruction for instruction, so the result is the same correctly
// rounded quotient bit for bit; what is saved are the scaling instructions and, for several numerators over one
// denominator (vec3 / float, the two sides of a box axis), the reciprocal.  26 issue cycles for one quotient, 12 + 14 k
// for k numerators.
//
// Precondition (callers establish it, see `fdiv_tame*`): d is zero / non-finite (fixed up) or 2^-125 <= |d| <= 2^125;
// n is zero / non-finite or |n| >= 2^-101 (the residual n - d*q is a multiple of 2^-47 |n| and must be representable);
// the quotient is zero or within [2^-125, 2^126].  tools/divcheck/divcheck.hip compares this against `n / d` on the chip
// for every denominator bit pattern and ~10^11 operand pairs inside and on the edges of the precondition.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace ptd {

__device__ __forceinline__ float fdiv_rcp(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float fdiv_q(float n, float d, float r)
{
    float q = n * r;
    float rem = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(rem, r, q);
    return __builtin_amdgcn_div_fixupf(q, d, n);
}
__device__ __forceinline__ float fdiv(float n, float d) { return fdiv_q(n, d, fdiv_rcp(d)); }

// x is zero or 2^lo <= |x| <= 2^hi, by bit pattern (NaN and infinities fail)
__device__ __forceinline__ bool fdiv_in_range(float x, int lo, int hi)
{
    const unsigned u = __float_as_uint(x) & 0x7fffffffu;
    return u == 0u || (u - ((unsigned)(lo + 127) << 23)) <= (((unsigned)(hi - lo)) << 23);
}

// 2^lo <= |x| <= 2^hi (zero, NaN and infinities fail)
__device__ __forceinline__ bool fdiv_in_range_nz(float x, int lo, int hi)
{
    const unsigned u = __float_as_uint(x) & 0x7fffffffu;
    return (u - ((unsigned)(lo + 127) << 23)) <= (((unsigned)(hi - lo)) << 23);
}

}  // namespace ptd
