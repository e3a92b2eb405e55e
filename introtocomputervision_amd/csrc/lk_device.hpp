// lk_device.hpp -- device-side arithmetic shared by the generic and the fused LK kernels.
// Every function here is the single definition of one step of the arithmetic contract
// (DESIGN.md): the generic and fused paths call the same code, so they agree bit for bit.
#pragma once
#include "common.hpp"

namespace micv {

// Sobel 3x3 pair from a 3x3 neighbourhood I[row][col] (already border-resolved).
// cv::cuda separable filter = row pass (float result) then column pass, each an fmaf chain
// from +0.  d/dx: rows [-1,0,1], cols [s,2s,s];  d/dy: rows [s,2s,s], cols [-1,0,1].
// The [-1,0,1] chain fmaf(c,1,fmaf(b,0,fmaf(a,-1,0))) equals c - a for finite inputs.
__device__ __forceinline__ void sobel3(const float I[3][3], float s1, float s2, float &gx,
                                       float &gy) {
    float tx[3], ty[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        tx[j] = I[j][2] - I[j][0];
        ty[j] = fmaf(I[j][2], s1, fmaf(I[j][1], s2, I[j][0] * s1));
    }
    gx = fmaf(tx[2], s1, fmaf(tx[1], s2, tx[0] * s1));
    gy = ty[2] - ty[0];
}

// OpticalFlow.cpp:62-64: Ix = (nextIx + prevIx) / 2.f  (cv::addWeighted(a,.5,b,.5,0)),
// It = next - prev.
__device__ __forceinline__ float avg2(float a, float b) { return a * 0.5f + b * 0.5f; }

// OpticalFlow.cpp:85-103 with cv::determinant / cv::solve(DECOMP_LU) 2x2 CV_32F: double
// det2, threshold tau = 0.1, Cramer with d = 1/det in double, results cast to float.
//
// Instruction count matters here (f64 issues at 1.85 ns per wave-instruction, ~14 % of the level
// kernel's VALU time went to this function), so two identities are used, both exact:
//  * a product of two floats is exact in double (48-bit significand), hence a*b - c*d rounds once
//    whether it is written mul, mul, sub or mul, fma;
//  * the correctly rounded 1/det is computed by the same rcp + Newton + residual-correction chain
//    the compiler emits for a division, minus its v_div_scale / v_div_fmas / v_div_fixup wrapping:
//    that wrapping only acts on operands near the ends of the exponent range, and det here lies in
//    [0.1, 2^256) (or is NaN, which the chain propagates like the division does).
__device__ __forceinline__ double lk_exact_diff(float a, float b, float c, float d) {
    return fma((double)a, (double)b, -((double)c * (double)d));  // == a*b - c*d in double
}
//    An infinite det does occur (an infinite pixel, or window sums that overflow): the Newton steps
//    turn rcp(inf) = 0 into NaN where the division gives 0, so the chain ends in the division's own
//    v_div_fixup (one instruction: 1/inf = 0, NaN stays NaN, every other det passes through).
__device__ __forceinline__ double lk_rcp_normal(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);  // residual of the quotient 1 * r
    return __builtin_amdgcn_div_fixup(fma(e, r, r), x, 1.0);
}
__device__ __forceinline__ void lk_solve(float sxx, float sxy, float syy, float sxt, float syt,
                                         float &u, float &v) {
    const double det = lk_exact_diff(sxx, syy, sxy, sxy);
    u = 0.f;
    v = 0.f;
    if (!(det < 0.1) && det != 0.) {
        const double d = lk_rcp_normal(det);
        const float b0 = -sxt, b1 = -syt;
        u = (float)(lk_exact_diff(b0, syy, b1, sxy) * d);
        v = (float)(lk_exact_diff(b1, sxx, b0, sxy) * d);
    }
}

// cvRound(float) as the reference's x86-64 OpenCV 3.4.1 executes it inside cv::remap
// (OpticalFlow.cpp:119; _mm_cvtss_si32 / _mm_cvtps_epi32): round half to even into 32 bits, and
// INT_MIN ("integer indefinite") for a NaN and for every value that does not fit an int32.  The map
// cell is then (-2^26 -> saturate_cast<short> = -32768, fraction 0): all four taps are outside the
// image, the sample is the border constant.  v_cvt_i32_f32 alone saturates and turns NaN into 0.
__device__ __forceinline__ int cv_round_i32(float v) {
    const int r = __float2int_rn(v);
    return fabsf(v) < 2147483648.f ? r : (int)0x80000000;
}

// The same sample, with the four taps served from an LDS copy of `next` when they fall inside
// the staged window [nx0, nx0+NW) x [ny0, ny0+NH) (which must lie inside the image) and from
// global memory otherwise.  Coordinates, weights and the blend are identical to warp_sample.
//
// xy32 = (32 x, 32 y) of the pixel (exact floats: the caller counts in float), flow = (du, dv).
//  * cv::remap's fixed-point coordinate is cvRound((x + du) * 32).  (x + du) * 32 = fma(du, 32, 32 x)
//    bit for bit: scaling by 32 is exact, so both round the same real number once (over- and underflow
//    included: x is an integer, so x + du is subnormal only for x = 0, where both are exact).  One
//    packed fma for both coordinates instead of a packed add and a packed multiply.
//  * With r = rndne(t) (the float the conversion rounds to) and q = r / 32 (exact), the map cell is
//    floor(q) and the 5-bit fraction over 32 is q - floor(q): one v_cvt_flr_i32_f32 and one
//    v_fract_f32 per coordinate replace convert, shift, mask, convert back and scale (r03: 36 -> 30
//    VALU instructions per warped pixel).  Both are exact for every finite r (q has at most five
//    fraction bits); an infinite or NaN coordinate never gets that far (see the window test).
//  * FS = 32 for a flow given as such; FS = 64 when the caller hands over HALF the flow (pyrUp's value
//    before OpticalFlow.cpp:142's "* 2"): (2 a) * 32 = a * 64 exactly, so the doubling costs nothing
//    where only the warp needs it (the halo pixels of a tile).
// ASM_TAPS: the four window taps are read by hand-placed ds_read2_b32 + an explicit wait (tiles whose prev DMA is
// still in flight: a tap the compiler can see would wait for that DMA too, lk_fused.hip DEFER).
template <int NW, int NH, int FS = 32, bool ASM_TAPS = false>
__device__ __forceinline__ float warp_sample_staged(const float *__restrict__ N, int nx0, int ny0,
                                                    const float *__restrict__ src, int rows,
                                                    int cols, int stride,
                                                    float __attribute__((ext_vector_type(2))) xy32,
                                                    float __attribute__((ext_vector_type(2))) flow, int xs = 1) {
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    const v2f_ t = __builtin_elementwise_fma(flow, (v2f_){(float)FS, (float)FS}, xy32);
    const v2f_ r = {__builtin_rintf(t.x), __builtin_rintf(t.y)};  // v_rndne_f32
    const v2f_ q = r * (v2f_){0.03125f, 0.03125f};
    // The saturating floor-conversion serves the window test: a coordinate beyond the int range lands
    // far outside the window (INT_MAX or INT_MIN against a window a few dozen cells wide: the
    // subtraction may wrap, it cannot wrap INTO the window), and inside the window the cell is
    // cvRound(t) >> 5 exactly.  Only a NaN needs help -- it converts to 0, which may well be a window
    // cell -- and gets it from ONE ordered compare of both coordinates.
    int ix, iy;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix) : "v"(q.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy) : "v"(q.y));
    const bool ordered = !__builtin_isunordered(t.x, t.y);
    float ax1, ay1, v0, v1, v2, v3;
    // the 16-bit clamp of cv::remap's integer map only matters outside the window: the window test
    // runs on the unclamped cell, the clamp moves into the fallback
    const int lx = ix - nx0, ly = iy - ny0;
    if (ordered && (unsigned)lx < (unsigned)(NW - 1) && (unsigned)ly < (unsigned)(NH - 1)) {
        ax1 = __builtin_amdgcn_fractf(q.x);
        ay1 = __builtin_amdgcn_fractf(q.y);
        // cell index by one full-rate 24-bit multiply-add (a plain `ly * NW + lx` turns into the
        // quarter-rate v_mul_lo_u32, and LLVM folds __mul24 and shift pairs back into it), and the
        // address as ONE register the four taps hang off by immediate offsets (two ds_read2_b32;
        // left alone, the compiler materialises a base per tap row)
        typedef const __attribute__((address_space(3))) float lds_cfloat;
        int cell;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(cell) : "v"(ly), "s"(NW), "v"(lx));
        int nbase = (int)(size_t)(lds_cfloat *)N;  // the window's LDS address, kept in a scalar register
        asm("" : "+s"(nbase));
        lds_cfloat *p = (lds_cfloat *)(size_t)(nbase + 4 * cell);
        asm("" : "+v"(p));
        if constexpr (ASM_TAPS) {
            static_assert(NW + 1 < 256, "ds_read2_b32 offsets are 8 bits");
            v2f_ t01, t23;
            asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(t01) : "v"(p) : "memory");
            asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t23) : "v"(p), "n"(NW), "n"(NW + 1) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t01), "+v"(t23));
            v0 = t01.x;
            v1 = t01.y;
            v2 = t23.x;
            v3 = t23.y;
        } else {
            v0 = p[0];
            v1 = p[1];
            v2 = p[NW];
            v3 = p[NW + 1];
        }
    } else {
        // cvRound proper (INT_MIN for NaN and beyond the int range: every tap is then the border
        // constant), cell and fraction from its integer result as cv::remap forms them
        const int cx = cv_round_i32(t.x), cy = cv_round_i32(t.y);
        ax1 = (float)(cx & 31) * 0.03125f;
        ay1 = (float)(cy & 31) * 0.03125f;
        const int gx = clampi(cx >> 5, -32768, 32767), gy = clampi(cy >> 5, -32768, 32767);
        const bool x0 = (unsigned)gx < (unsigned)cols, x1 = (unsigned)(gx + 1) < (unsigned)cols;
        const bool y0 = (unsigned)gy < (unsigned)rows, y1 = (unsigned)(gy + 1) < (unsigned)rows;
        const float *p = src + (ptrdiff_t)gy * stride + (ptrdiff_t)gx * xs;  // xs: pixel stride (levels read from level 0)
        v0 = (x0 && y0) ? p[0] : 0.f;
        v1 = (x1 && y0) ? p[xs] : 0.f;
        v2 = (x0 && y1) ? p[stride] : 0.f;
        v3 = (x1 && y1) ? p[stride + xs] : 0.f;
    }
    // weights as aligned pairs, so the four weight products and the four tap products are two packed
    // multiplies each with a broadcast operand (the scalar form made the compiler shuffle halves)
    const v2f_ X = {1.f - ax1, ax1}, Y = {1.f - ay1, ay1};
    // r = v0*(ay0*ax0) + v1*(ay0*ax1) + v2*(ay1*ax0) + v3*(ay1*ax1), summed left to right
    const v2f_ w0 = (v2f_){Y.x, Y.x} * X, w1 = (v2f_){Y.y, Y.y} * X;
    const v2f_ p0 = (v2f_){v0, v1} * w0, p1 = (v2f_){v2, v3} * w1;
    float s = p0.x + p0.y;
    s = s + p1.x;
    s = s + p1.y;
    return s;
}

// lk::warp (OpticalFlow.cpp:111-119) for one pixel: map = (x + du, y + dv); cv::remap
// INTER_LINEAR with 1/32-pixel fixed-point coordinates, BORDER_CONSTANT(0).
__device__ __forceinline__ float warp_sample(const float *__restrict__ src, int rows, int cols,
                                             int stride, int x, int y, float du, float dv, int xs = 1) {
    const float mx = (float)x + du, my = (float)y + dv;
    const int sx = cv_round_i32(mx * 32.f), sy = cv_round_i32(my * 32.f);
    const int fx = sx & 31, fy = sy & 31;
    const int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
    const float ax1 = (float)fx * 0.03125f, ax0 = 1.f - ax1;
    const float ay1 = (float)fy * 0.03125f, ay0 = 1.f - ay1;
    const bool x0 = (unsigned)ix < (unsigned)cols, x1 = (unsigned)(ix + 1) < (unsigned)cols;
    const bool y0 = (unsigned)iy < (unsigned)rows, y1 = (unsigned)(iy + 1) < (unsigned)rows;
    const float *p = src + (ptrdiff_t)iy * stride + (ptrdiff_t)ix * xs;  // xs: pixel stride (levels read from level 0)
    const float v0 = (x0 && y0) ? p[0] : 0.f;
    const float v1 = (x1 && y0) ? p[xs] : 0.f;
    const float v2 = (x0 && y1) ? p[stride] : 0.f;
    const float v3 = (x1 && y1) ? p[stride + xs] : 0.f;
    float r = v0 * (ay0 * ax0);
    r = r + v1 * (ay0 * ax1);
    r = r + v2 * (ay1 * ax0);
    r = r + v3 * (ay1 * ax1);
    return r;
}

}  // namespace micv
