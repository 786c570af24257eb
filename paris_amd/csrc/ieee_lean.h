// Hand-expanded IEEE fp32 division and square root for operands far from the range limits, shared by the kernels that use them
// (bp_device.h: column_constants; filter_fused.hip: weighted) and by the device validators that prove them (validate.hip).
//
// hipcc expands a correctly rounded fp32 division into v_div_scale x 2, v_rcp, one Newton step on the reciprocal, two quotient
// corrections, v_div_fmas, v_div_fixup, and a correctly rounded sqrtf into v_sqrt_f32 plus a +-1 ulp residual test (with a 2^32
// pre-scaling for radicands below 2^-96). For ordinary operands the scalings are by 1 and the fix-ups are the identity, which
// leaves the sequences below. "Has the bits of the compiler's expansion" is a claim about THIS compiler: it is not taken on
// trust -- before a kernel may use these forms the host runs validate.hip's kernels, which compare them with the compiler's
// `/` and sqrtf for EVERY fp32 operand the launch can produce (src/openmp/backprojection.cpp:125,139, src/openmp/weighting.cpp:52
// are the operations reproduced), and falls back to the plain IEEE forms on any mismatch.
#ifndef PARIS_HIP_IEEE_LEAN_H_
#define PARIS_HIP_IEEE_LEAN_H_

#include <hip/hip_runtime.h>

namespace paris_lean
{
    // r = 1 / den refined once: the reciprocal both quotients of lean_div share
    __device__ __forceinline__ float refined_rcp(float den)
    {
        float r = __builtin_amdgcn_rcpf(den);
        return __builtin_fmaf(__builtin_fmaf(-den, r, 1.f), r, r);
    }

    // n / den given r = refined_rcp(den): q = n r; q += (n - den q) r; q += (n - den q) r
    __device__ __forceinline__ float div_with_rcp(float n, float den, float r)
    {
        float q = n * r;
        q = __builtin_fmaf(__builtin_fmaf(-den, q, n), r, q);
        return __builtin_fmaf(__builtin_fmaf(-den, q, n), r, q);
    }

    __device__ __forceinline__ float div_rn_safe_range(float n, float d)
    {
        return div_with_rcp(n, d, refined_rcp(d));
    }

    __device__ __forceinline__ float sqrt_rn_safe_range(float x)
    {
        float s = __builtin_amdgcn_sqrtf(x);
        const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
        const float r_lo = __builtin_fmaf(-lo, s, x), r_hi = __builtin_fmaf(-hi, s, x);
        s = r_lo <= 0.f ? lo : s;
        s = r_hi > 0.f ? hi : s;
        return s;
    }
}

#endif
