// ncc_arith.hpp -- fl(a / fl(sqrt(p))) of DisparityNCorr.cu:106 for operands in a checked range, without the
// range handling of the compiler's sqrtf / division (denormal scaling, v_div_scale / v_div_fmas / v_div_fixup,
// class tests): 14 VALU instructions instead of 28.
//
//   ncc_sqrt(p): v_rsq_f32, s = p r, h = r / 2, one residual step s + h (p - s^2).  Equal to the correctly rounded
//       sqrtf(p) for EVERY float with exponent in [-64, 96] -- checked exhaustively on the device
//       (tools/probes/ncc_arith_probe.hip, tests/test_ncc_arith_gpu.py).
//   ncc_div(a, q): the compiler's own sequence (v_rcp_f32, one Newton step, quotient, two remainder steps) with
//       the scaling taken out: v_div_scale is the identity and v_div_fmas a plain fma when no operand or
//       intermediate leaves the normal range, v_div_fixup only replaces results for zero / inf / NaN operands.
//       The probe compares 10^11 pairs (random and adversarial mantissas) with a / q as well.
//
// stereo.hip takes this path for a wave whose staged pixels are all 0 or of magnitude in [2^-8, 2^16] (8-bit,
// 16-bit and [0, 1]-normalised images all are) and whose window energies are 0 or in [2^-16, 2^39]: then
// p in {0} u [2^-32, 2^78], q in {0} u [2^-16, 2^39], |a| in {0} u [2^-62, 2^39] (sums of products that are
// multiples of 2^-62), quotients and remainders stay normal.  p = 0 (a window of zeros, a = 0) gives NaN on both
// paths -- the comparison that follows is false either way.
#pragma once
#include <hip/hip_runtime.h>

namespace micv {

__device__ __forceinline__ float ncc_sqrt(float p) {
    const float r = __builtin_amdgcn_rsqf(p);
    const float s = p * r, h = 0.5f * r;
    const float d = __builtin_fmaf(-s, s, p);
    return __builtin_fmaf(d, h, s);
}

__device__ __forceinline__ float ncc_div(float a, float q) {
    float r = __builtin_amdgcn_rcpf(q);
    const float e = __builtin_fmaf(-q, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float t = a * r;
    float m = __builtin_fmaf(-q, t, a);
    t = __builtin_fmaf(m, r, t);
    m = __builtin_fmaf(-q, t, a);
    return __builtin_fmaf(m, r, t);
}

// bit patterns of the range limits
constexpr unsigned NCC_PIX_LO = (127u - 8) << 23, NCC_PIX_HI = (127u + 16) << 23;
constexpr unsigned NCC_EN_LO = (127u - 16) << 23, NCC_EN_HI = (127u + 39) << 23;

// Range tracking without lane masks (compare results held in SGPR pairs spill): the largest magnitude and the
// smallest NONZERO magnitude seen, as bit patterns (0 - 1 wraps to the top and never wins the minimum; NaN and
// inf sort above every finite value).  Four VALU instructions per value.
struct NccRange {
    unsigned hi = 0u, lo = 0xffffffffu;  // max |v|, min (|v| - 1 ulp) over nonzero v
    __device__ __forceinline__ void add(float v) {
        const unsigned b = __float_as_uint(v) & 0x7fffffffu;
        hi = b > hi ? b : hi;
        const unsigned t = b - 1u;
        lo = t < lo ? t : lo;
    }
    __device__ __forceinline__ bool inside(unsigned lo_bits, unsigned hi_bits) const {
        return hi <= hi_bits && lo >= lo_bits - 1u;
    }
};

}  // namespace micv
