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
__device__ __forceinline__ void lk_solve(float sxx, float sxy, float syy, float sxt, float syt,
                                         float &u, float &v) {
    const double det = (double)sxx * (double)syy - (double)sxy * (double)sxy;
    u = 0.f;
    v = 0.f;
    if (!(det < 0.1) && det != 0.) {
        const double d = 1. / det;
        const float b0 = -sxt, b1 = -syt;
        u = (float)(((double)b0 * (double)syy - (double)b1 * (double)sxy) * d);
        v = (float)(((double)b1 * (double)sxx - (double)b0 * (double)sxy) * d);
    }
}

// The same sample, with the four taps served from an LDS copy of `next` when they fall inside
// the staged window [nx0, nx0+NW) x [ny0, ny0+NH) (which must lie inside the image) and from
// global memory otherwise.  Coordinates, weights and the blend are identical to warp_sample.
template <int NW, int NH>
__device__ __forceinline__ float warp_sample_staged(const float *__restrict__ N, int nx0, int ny0,
                                                    const float *__restrict__ src, int rows,
                                                    int cols, int stride, int x, int y, float du,
                                                    float dv) {
    const float mx = (float)x + du, my = (float)y + dv;
    const int sx = __float2int_rn(mx * 32.f), sy = __float2int_rn(my * 32.f);
    const int fx = sx & 31, fy = sy & 31;
    const int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
    const float ax1 = (float)fx * 0.03125f, ax0 = 1.f - ax1;
    const float ay1 = (float)fy * 0.03125f, ay0 = 1.f - ay1;
    float v0, v1, v2, v3;
    const int lx = ix - nx0, ly = iy - ny0;
    if ((unsigned)lx < (unsigned)(NW - 1) && (unsigned)ly < (unsigned)(NH - 1)) {
        // row offset by shifts where the window width allows (96 = 64 + 32: two full-rate ops instead
        // of the quarter-rate v_mul_lo_u32 a plain `ly * NW` turns into)
        int row_off;
        if (NW == 96) {
            int hi = ly << 6;
            asm("" : "+v"(hi));  // keeps LLVM from folding the two shifts back into one multiply
            row_off = hi + (ly << 5);
        } else {
            row_off = ly * NW;
        }
        const float *p = N + row_off + lx;
        v0 = p[0];
        v1 = p[1];
        v2 = p[NW];
        v3 = p[NW + 1];
    } else {
        const bool x0 = (unsigned)ix < (unsigned)cols, x1 = (unsigned)(ix + 1) < (unsigned)cols;
        const bool y0 = (unsigned)iy < (unsigned)rows, y1 = (unsigned)(iy + 1) < (unsigned)rows;
        const float *p = src + (ptrdiff_t)iy * stride + ix;
        v0 = (x0 && y0) ? p[0] : 0.f;
        v1 = (x1 && y0) ? p[1] : 0.f;
        v2 = (x0 && y1) ? p[stride] : 0.f;
        v3 = (x1 && y1) ? p[stride + 1] : 0.f;
    }
    float r = v0 * (ay0 * ax0);
    r = r + v1 * (ay0 * ax1);
    r = r + v2 * (ay1 * ax0);
    r = r + v3 * (ay1 * ax1);
    return r;
}

// lk::warp (OpticalFlow.cpp:111-119) for one pixel: map = (x + du, y + dv); cv::remap
// INTER_LINEAR with 1/32-pixel fixed-point coordinates, BORDER_CONSTANT(0).
__device__ __forceinline__ float warp_sample(const float *__restrict__ src, int rows, int cols,
                                             int stride, int x, int y, float du, float dv) {
    const float mx = (float)x + du, my = (float)y + dv;
    const int sx = __float2int_rn(mx * 32.f), sy = __float2int_rn(my * 32.f);
    const int fx = sx & 31, fy = sy & 31;
    const int ix = clampi(sx >> 5, -32768, 32767), iy = clampi(sy >> 5, -32768, 32767);
    const float ax1 = (float)fx * 0.03125f, ax0 = 1.f - ax1;
    const float ay1 = (float)fy * 0.03125f, ay0 = 1.f - ay1;
    const bool x0 = (unsigned)ix < (unsigned)cols, x1 = (unsigned)(ix + 1) < (unsigned)cols;
    const bool y0 = (unsigned)iy < (unsigned)rows, y1 = (unsigned)(iy + 1) < (unsigned)rows;
    const float *p = src + (ptrdiff_t)iy * stride + ix;
    const float v0 = (x0 && y0) ? p[0] : 0.f;
    const float v1 = (x1 && y0) ? p[1] : 0.f;
    const float v2 = (x0 && y1) ? p[stride] : 0.f;
    const float v3 = (x1 && y1) ? p[stride + 1] : 0.f;
    float r = v0 * (ay0 * ax0);
    r = r + v1 * (ay0 * ax1);
    r = r + v2 * (ay1 * ax0);
    r = r + v3 * (ay1 * ax1);
    return r;
}

}  // namespace micv
