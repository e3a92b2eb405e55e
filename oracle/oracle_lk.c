/*
 * oracle_lk.c -- CPU restatement of the ps5 Lucas-Kanade / pyramid path and the separable
 * filters it rests on.  TEST INFRASTRUCTURE ONLY; parity unpinned (see oracle.h).
 *
 * Build with -ffp-contract=off: every fused multiply-add below is an explicit fmaf();
 * everything written as a*b + c is an unfused multiply followed by an add.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AT(p, stride, y, x) ((p)[(size_t)(y) * (stride) + (size_t)(x)])

/* cv::borderInterpolate, BORDER_REFLECT_101 branch (OpenCV 3.4.1 modules/core/src/copy.cpp). */
int orc_reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0)
            p = -p;
        else
            p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* cv::getGaussianKernel(n, sigma, CV_32F), sigma > 0 (OpenCV 3.4.1 imgproc/src/smooth.cpp). */
void orc_gaussian_kernel(int n, double sigma, float *taps) {
    double scale2x = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = exp(scale2x * x * x);
        taps[i] = (float)t;
        sum += taps[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) taps[i] = (float)(taps[i] * sum);
}

void orc_sep_filter(const float *src, int rows, int cols, size_t sstride,
                    const float *krow, int nrow, const float *kcol, int ncol,
                    float *dst, size_t dstride) {
    float *tmp = (float *)malloc((size_t)rows * cols * sizeof(float));
    int ar = nrow / 2, ac = ncol / 2;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float acc = 0.f;
            for (int k = 0; k < nrow; k++)
                acc = fmaf(AT(src, sstride, y, orc_reflect101(x + k - ar, cols)), krow[k], acc);
            tmp[(size_t)y * cols + x] = acc;
        }
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float acc = 0.f;
            for (int k = 0; k < ncol; k++)
                acc = fmaf(tmp[(size_t)orc_reflect101(y + k - ac, rows) * cols + x], kcol[k], acc);
            AT(dst, dstride, y, x) = acc;
        }
    free(tmp);
}

/* ---- alternative accumulation orders, for BOUNDING the unpinned decisions (never the contract) ----
 * cv::GaussianBlur on CV_32F (OpticalFlow.cpp:73-77) runs OpenCV 3.4.1's CPU FilterEngine
 * (imgproc/src/filter.cpp): RowFilter<float,float,RowVec_32f> -- `s = kx[0]*S[0]; s += kx[k]*S[k]`,
 * k ascending -- then SymmColumnFilter<Cast<float,float>,SymmColumnVec_32f> for the symmetric Gaussian
 * taps -- `s = ky[c]*S[c]; s += ky[c+i]*(S[c+i] + S[c-i])`, i = 1..ksize/2, from the centre outwards.
 * Whether its multiply-adds are fused depends on the build (the SSE2 baseline is not; the AVX2/FMA3
 * dispatch of filter.avx2.cpp is): `fused` selects.  The contract (orc_sep_filter) is the left-to-right
 * fmaf chain in BOTH passes; its row pass equals this row pass with fused = 1 bit for bit, the column
 * pass differs in order either way.  Published algorithm restated; not verifiable here. */
void orc_sep_filter_cvcpu(const float *src, int rows, int cols, size_t sstride,
                          const float *krow, int nrow, const float *kcol, int ncol,
                          float *dst, size_t dstride, int fused) {
    float *tmp = (float *)malloc((size_t)rows * cols * sizeof(float));
    int ar = nrow / 2, ac = ncol / 2;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float acc = krow[0] * AT(src, sstride, y, orc_reflect101(x - ar, cols));
            for (int k = 1; k < nrow; k++) {
                float s = AT(src, sstride, y, orc_reflect101(x + k - ar, cols));
                acc = fused ? fmaf(s, krow[k], acc) : acc + krow[k] * s;
            }
            tmp[(size_t)y * cols + x] = acc;
        }
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float acc = kcol[ac] * tmp[(size_t)y * cols + x];
            for (int i = 1; i <= ac; i++) {
                float pair = tmp[(size_t)orc_reflect101(y + i, rows) * cols + x] +
                             tmp[(size_t)orc_reflect101(y - i, rows) * cols + x];
                acc = fused ? fmaf(pair, kcol[ac + i], acc) : acc + kcol[ac + i] * pair;
            }
            AT(dst, dstride, y, x) = acc;
        }
    free(tmp);
}

/* cv::getDerivKernels -> getSobelKernels (OpenCV 3.4.1 imgproc/src/deriv.cpp), integer taps. */
static int sobel_kernel_1d(int ksize, int order, float *out) {
    int ker[40];
    if (ksize == 1 && order > 0) ksize = 3;
    if (ksize > 31 || (ksize & 1) == 0) return -1;
    if (ksize == 1) {
        ker[0] = 1;
    } else if (ksize == 3) {
        if (order == 0) { ker[0] = 1; ker[1] = 2; ker[2] = 1; }
        else if (order == 1) { ker[0] = -1; ker[1] = 0; ker[2] = 1; }
        else { ker[0] = 1; ker[1] = -2; ker[2] = 1; }
    } else {
        ker[0] = 1;
        for (int i = 0; i < ksize; i++) ker[i + 1] = 0;
        for (int i = 0; i < ksize - order - 1; i++) {
            int oldval = ker[0];
            for (int j = 1; j <= ksize; j++) {
                int newval = ker[j] + ker[j - 1];
                ker[j - 1] = oldval;
                oldval = newval;
            }
        }
        for (int i = 0; i < order; i++) {
            int oldval = -ker[0];
            for (int j = 1; j <= ksize; j++) {
                int newval = ker[j - 1] - ker[j];
                ker[j - 1] = oldval;
                oldval = newval;
            }
        }
    }
    for (int i = 0; i < ksize; i++) out[i] = (float)ker[i];
    return ksize;
}

/* cv::cuda::createSobelFilter (OpenCV 3.4.1 cudafilters/src/filtering.cpp): the scale is
 * folded into the smoothing kernel: `if (dx == 0) kx *= scale; else ky *= scale;`. */
int orc_sobel(const float *src, int rows, int cols, size_t sstride, int ksize, float scale,
              float *gx, float *gy, size_t gstride) {
    float kx[32], ky[32];
    int nx, ny;
    /* d/dx: dx=1, dy=0 */
    nx = sobel_kernel_1d(ksize, 1, kx);
    ny = sobel_kernel_1d(ksize, 0, ky);
    if (nx < 0 || ny < 0) return -1;
    if (scale != 1.f) for (int i = 0; i < ny; i++) ky[i] *= scale;
    orc_sep_filter(src, rows, cols, sstride, kx, nx, ky, ny, gx, gstride);
    /* d/dy: dx=0, dy=1 */
    nx = sobel_kernel_1d(ksize, 0, kx);
    ny = sobel_kernel_1d(ksize, 1, ky);
    if (scale != 1.f) for (int i = 0; i < nx; i++) kx[i] *= scale;
    orc_sep_filter(src, rows, cols, sstride, kx, nx, ky, ny, gy, gstride);
    return 0;
}

/* lk::calcOpticalFlow, ps5_cpp/lib/OpticalFlow.cpp:41-104.  variant 0 = the contract; ORC_VAR_BLUR_CVCPU
 * (+ ORC_VAR_BLUR_FUSED) runs the five window sums in OpenCV's CPU filter order instead (above).  det_out
 * (optional, rows x cols doubles, dense) receives det(A) of every pixel: the distance to the
 * `det < 0.1` discontinuity (:82,95). */
int orc_lk_flow_ex(const float *prev, const float *next, int rows, int cols, size_t stride,
                   int win, int variant, float *u, float *v, size_t ostride, double *det_out) {
    if (win < 1 || (win & 1) == 0 || win > 255) return -1;
    size_t n = (size_t)rows * cols;
    float *buf = (float *)malloc(9 * n * sizeof(float));
    float *pIx = buf, *pIy = buf + n, *nIx = buf + 2 * n, *nIy = buf + 3 * n;
    float *Sxx = buf + 4 * n, *Sxy = buf + 5 * n, *Syy = buf + 6 * n, *Sxt = buf + 7 * n,
          *Syt = buf + 8 * n;
    const float scale = 1.f / 9.f; /* OpticalFlow.cpp:19 */
    orc_sobel(prev, rows, cols, stride, 3, scale, pIx, pIy, cols); /* :60 */
    orc_sobel(next, rows, cols, stride, 3, scale, nIx, nIy, cols); /* :61 */
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            size_t i = (size_t)y * cols + x;
            /* :62-63 (a+b)/2.f evaluates as cv::addWeighted(a,.5,b,.5,0) */
            float ix = nIx[i] * 0.5f + pIx[i] * 0.5f;
            float iy = nIy[i] * 0.5f + pIy[i] * 0.5f;
            float it = AT(next, stride, y, x) - AT(prev, stride, y, x); /* :64 */
            Sxx[i] = ix * ix; /* :66-70 */
            Sxy[i] = ix * iy;
            Syy[i] = iy * iy;
            Sxt[i] = ix * it;
            Syt[i] = iy * it;
        }
    /* :73-77 cv::GaussianBlur(S, S, Size(win,win), float(win)/3.f) */
    float g[256];
    orc_gaussian_kernel(win, (double)((float)win / 3.f), g);
    float *tmp = pIx; /* gradients are dead now */
    float *fields[5] = {Sxx, Sxy, Syy, Sxt, Syt};
    for (int f = 0; f < 5; f++) {
        if (variant & ORC_VAR_BLUR_CVCPU)
            orc_sep_filter_cvcpu(fields[f], rows, cols, cols, g, win, g, win, tmp, cols, (variant & ORC_VAR_BLUR_FUSED) != 0);
        else
            orc_sep_filter(fields[f], rows, cols, cols, g, win, g, win, tmp, cols);
        memcpy(fields[f], tmp, n * sizeof(float));
    }
    const double tau = 0.1; /* :82 */
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            size_t i = (size_t)y * cols + x;
            float a00 = Sxx[i], a01 = Sxy[i], a10 = Sxy[i], a11 = Syy[i];
            float b0 = -Sxt[i], b1 = -Syt[i];
            /* cv::determinant, 2x2 CV_32F: det2() in double (OpenCV 3.4.1 core/src/lapack.cpp) */
            double det = (double)a00 * a11 - (double)a01 * a10;
            if (det_out) det_out[i] = det;
            float uu = 0.f, vv = 0.f;
            if (!(det < tau)) {
                /* cv::solve, DECOMP_LU, 2x2 CV_32F fast path (lapack.cpp) */
                double d = det;
                if (d != 0.) {
                    d = 1. / d;
                    uu = (float)(((double)b0 * a11 - (double)b1 * a01) * d);
                    vv = (float)(((double)b1 * a00 - (double)b0 * a10) * d);
                }
            }
            AT(u, ostride, y, x) = uu;
            AT(v, ostride, y, x) = vv;
        }
    free(buf);
    return 0;
}

int orc_lk_flow(const float *prev, const float *next, int rows, int cols, size_t stride,
                int win, float *u, float *v, size_t ostride) {
    return orc_lk_flow_ex(prev, next, rows, cols, stride, win, 0, u, v, ostride, NULL);
}

static int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* cvRound(float) as OpenCV 3.4.1 executes it on the reference's x86-64 build (core/fast_math.hpp:
 * _mm_cvtss_si32; the vector body of RemapInvoker uses _mm_cvtps_epi32): round half to even in
 * 32 bits, and the "integer indefinite" value INT_MIN for a NaN and for every value whose rounded
 * result does not fit an int32.  Written out so that it does not depend on sizeof(long) or on what
 * lrintf() returns outside the int range.  With INT_MIN the map cell is (-2^26 -> saturate_cast<short>
 * = -32768, fraction 0): every tap is outside the image and cv::remap writes the border constant. */
int orc_cv_round(float v) {
    if (!(v > -2147483648.f && v < 2147483648.f)) return (int)(-2147483647 - 1);
    return (int)rintf(v); /* default rounding mode: to nearest even; the result fits an int */
}

/* cv::remap, CV_32FC1 maps, INTER_LINEAR, BORDER_CONSTANT(0)
 * (OpenCV 3.4.1 imgproc/src/imgwarp.cpp: RemapInvoker + remapBilinear<Cast<float,float>>). */
/* The frame position of pixel (0, 0) of the arrays orc_lk_warp works on (orc_lk_flow_pyr_at: the oracle on a CROP of a
 * large frame).  cv::remap's map is the float sum "pixel index + flow", so a warped value depends on where in the frame
 * the pixel sits (the sum rounds at the magnitude of the index): with an origin the map carries the frame's index and the
 * tap addresses are brought back into the crop after the fixed-point conversion.  (0, 0) everywhere else. */
static int g_org_y = 0, g_org_x = 0, g_base_org_y = 0, g_base_org_x = 0;

void orc_remap_linear(const float *src, int rows, int cols, size_t sstride,
                      const float *mapx, const float *mapy, size_t mstride,
                      float *dst, int drows, int dcols, size_t dstride) {
    for (int y = 0; y < drows; y++)
        for (int x = 0; x < dcols; x++) {
            int sx = orc_cv_round(AT(mapx, mstride, y, x) * 32.f); /* cvRound(v*INTER_TAB_SIZE) */
            int sy = orc_cv_round(AT(mapy, mstride, y, x) * 32.f);
            int fx = sx & 31, fy = sy & 31;
            int ix = sat_short(sx >> 5) - g_org_x, iy = sat_short(sy >> 5) - g_org_y;
            /* BilinearTab_f: 1-D taps {1 - k/32, k/32}, 2-D weight = vy*vx */
            float ax1 = fx * (1.f / 32.f), ax0 = 1.f - ax1;
            float ay1 = fy * (1.f / 32.f), ay0 = 1.f - ay1;
            float w0 = ay0 * ax0, w1 = ay0 * ax1, w2 = ay1 * ax0, w3 = ay1 * ax1;
            int x0ok = (unsigned)ix < (unsigned)cols, x1ok = (unsigned)(ix + 1) < (unsigned)cols;
            int y0ok = (unsigned)iy < (unsigned)rows, y1ok = (unsigned)(iy + 1) < (unsigned)rows;
            float v0 = (x0ok && y0ok) ? AT(src, sstride, iy, ix) : 0.f;
            float v1 = (x1ok && y0ok) ? AT(src, sstride, iy, ix + 1) : 0.f;
            float v2 = (x0ok && y1ok) ? AT(src, sstride, iy + 1, ix) : 0.f;
            float v3 = (x1ok && y1ok) ? AT(src, sstride, iy + 1, ix + 1) : 0.f;
            float r = v0 * w0;
            r = r + v1 * w1;
            r = r + v2 * w2;
            r = r + v3 * w3;
            AT(dst, dstride, y, x) = r;
        }
}

/* lk::warp, OpticalFlow.cpp:106-120. */
void orc_lk_warp(const float *src, const float *du, const float *dv, int rows, int cols,
                 size_t stride, float *dst) {
    size_t n = (size_t)rows * cols;
    float *mx = (float *)malloc(2 * n * sizeof(float)), *my = mx + n;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            mx[(size_t)y * cols + x] = (float)(x + g_org_x) + AT(du, stride, y, x); /* :113 */
            my[(size_t)y * cols + x] = (float)(y + g_org_y) + AT(dv, stride, y, x); /* :114 */
        }
    orc_remap_linear(src, rows, cols, stride, mx, my, cols, dst, rows, cols, stride); /* :119 */
    free(mx);
}

/* cv::resize INTER_LINEAR, CV_32F (OpenCV 3.4.1 imgproc/src/resize.cpp: resizeGeneric_,
 * HResizeLinear<float,float,float,1>, VResizeLinear<float,float,float,Cast>). */
void orc_resize_linear(const float *src, int srows, int scols, size_t sstride,
                       float *dst, int drows, int dcols, size_t dstride) {
    double scale_x = 1. / ((double)dcols / scols);
    double scale_y = 1. / ((double)drows / srows);
    int *xofs = (int *)malloc((size_t)dcols * sizeof(int));
    float *alpha = (float *)malloc((size_t)dcols * 2 * sizeof(float));
    float *row0 = (float *)malloc((size_t)dcols * 2 * sizeof(float)), *row1 = row0 + dcols;
    for (int dx = 0; dx < dcols; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= scols - 1) { fx = 0; sx = scols - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = 1.f - fx;
        alpha[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < drows; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        float b0 = 1.f - fy, b1 = fy;
        int y0 = sy < 0 ? 0 : (sy > srows - 1 ? srows - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 > srows - 1 ? srows - 1 : sy + 1);
        const float *S[2] = {src + (size_t)y0 * sstride, src + (size_t)y1 * sstride};
        float *R[2] = {row0, row1};
        for (int k = 0; k < 2; k++)
            for (int dx = 0; dx < dcols; dx++) {
                int sx = xofs[dx];
                if (sx + 1 >= scols) /* dx >= xmax: single tap times ONE */
                    R[k][dx] = S[k][sx] * 1.f;
                else
                    R[k][dx] = S[k][sx] * alpha[2 * dx] + S[k][sx + 1] * alpha[2 * dx + 1];
            }
        for (int dx = 0; dx < dcols; dx++)
            AT(dst, dstride, dy, dx) = row0[dx] * b0 + row1[dx] * b1;
    }
    free(xofs);
    free(alpha);
    free(row0);
}

/* pyr::pyrDown as written: Pyramids.cu:31 (kernel), :53 (dst dims), :65-66 (launched on d_src). */
void orc_pyr_down(const float *src, int rows, int cols, size_t sstride,
                  float *dst, size_t dstride) {
    int dr = rows / 2, dc = cols / 2;
    for (int y = 0; y < dr; y++)
        for (int x = 0; x < dc; x++)
            AT(dst, dstride, y, x) = AT(src, sstride, 2 * y + 1, 2 * x + 1);
}

/* pyr::pyrUp: Pyramids.cu:86-91 (replicate), :19,126-127 (separable [1,4,6,4,1]/16). */
void orc_pyr_up(const float *src, int rows, int cols, size_t sstride,
                float *dst, size_t dstride) {
    static const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    int ur = rows * 2, uc = cols * 2;
    float *up = (float *)malloc((size_t)ur * uc * sizeof(float));
    for (int y = 0; y < ur; y++)
        for (int x = 0; x < uc; x++) up[(size_t)y * uc + x] = AT(src, sstride, y / 2, x / 2);
    orc_sep_filter(up, ur, uc, uc, g5, 5, g5, 5, dst, dstride);
    free(up);
}

/* orc_lk_flow_pyr on a crop whose pixel (0, 0) is pixel (oy, ox) of the frame (both multiples of 2^(levels - 1), so the
 * crop's pyramid is a crop of the frame's): away from the crop's own borders the result is the frame's, bit for bit --
 * the size-independent check of tests/test_large_gpu.py.  Not re-entrant (file-scope origin). */
int orc_lk_flow_pyr_at(const float *prev, const float *next, int rows, int cols, size_t stride,
                       int win, int levels, int oy, int ox, float *u, float *v, size_t ostride) {
    if (levels < 1 || levels > 16 || oy < 0 || ox < 0 || (oy & ((1 << (levels - 1)) - 1)) || (ox & ((1 << (levels - 1)) - 1))) return -1;
    g_base_org_y = oy; g_base_org_x = ox;
    int rc = orc_lk_flow_pyr_ex(prev, next, rows, cols, stride, win, levels, 0, u, v, ostride, NULL);
    g_base_org_y = 0; g_base_org_x = 0;
    return rc;
}

/* lk::calcOpticalFlowPyr, OpticalFlow.cpp:122-167; `levels` replaces pyrDepth = 4 (:127). */
int orc_lk_flow_pyr(const float *prev, const float *next, int rows, int cols, size_t stride,
                    int win, int levels, float *u, float *v, size_t ostride) {
    return orc_lk_flow_pyr_ex(prev, next, rows, cols, stride, win, levels, 0, u, v, ostride, NULL);
}

/* The same with a `variant` of the window sums at every level (orc_lk_flow_ex); det0_out: det(A) of the
 * finest level's solve. */
int orc_lk_flow_pyr_ex(const float *prev, const float *next, int rows, int cols, size_t stride,
                       int win, int levels, int variant, float *u, float *v, size_t ostride, double *det0_out) {
    if (levels < 1 || levels > 16) return -1;
    float *pp[16], *np[16];
    int pr[16], pc[16];
    pr[0] = rows; pc[0] = cols;
    pp[0] = (float *)malloc((size_t)rows * cols * sizeof(float));
    np[0] = (float *)malloc((size_t)rows * cols * sizeof(float));
    for (int y = 0; y < rows; y++) {
        memcpy(pp[0] + (size_t)y * cols, prev + (size_t)y * stride, cols * sizeof(float));
        memcpy(np[0] + (size_t)y * cols, next + (size_t)y * stride, cols * sizeof(float));
    }
    for (int l = 1; l < levels; l++) { /* Pyramids.cpp:19-23 */
        pr[l] = pr[l - 1] / 2; pc[l] = pc[l - 1] / 2;
        if (pr[l] < 1 || pc[l] < 1) {
            for (int k = 0; k < l; k++) { free(pp[k]); free(np[k]); }
            return -2;
        }
        pp[l] = (float *)malloc((size_t)pr[l] * pc[l] * sizeof(float));
        np[l] = (float *)malloc((size_t)pr[l] * pc[l] * sizeof(float));
        orc_pyr_down(pp[l - 1], pr[l - 1], pc[l - 1], pc[l - 1], pp[l], pc[l]);
        orc_pyr_down(np[l - 1], pr[l - 1], pc[l - 1], pc[l - 1], np[l], pc[l]);
    }
    int dr = pr[levels - 1], dc = pc[levels - 1];
    float *du = (float *)calloc((size_t)dr * dc, sizeof(float)); /* :132-133 */
    float *dv = (float *)calloc((size_t)dr * dc, sizeof(float));
    int rc = 0;
    for (int level = 0; level < levels; level++) {
        int k = levels - level - 1;
        int R = pr[k], C = pc[k];
        if (level > 0) { /* :139-145 */
            float *tu = (float *)malloc((size_t)dr * 2 * dc * 2 * sizeof(float));
            float *tv = (float *)malloc((size_t)dr * 2 * dc * 2 * sizeof(float));
            orc_pyr_up(du, dr, dc, dc, tu, (size_t)dc * 2);
            orc_pyr_up(dv, dr, dc, dc, tv, (size_t)dc * 2);
            free(du); free(dv);
            du = tu; dv = tv;
            dr *= 2; dc *= 2;
            for (size_t i = 0; i < (size_t)dr * dc; i++) { du[i] = 2.f * du[i]; dv[i] = 2.f * dv[i]; }
        }
        if (dr != R || dc != C) { /* :148-151 */
            float *tu = (float *)malloc((size_t)R * C * sizeof(float));
            float *tv = (float *)malloc((size_t)R * C * sizeof(float));
            orc_resize_linear(du, dr, dc, dc, tu, R, C, C);
            orc_resize_linear(dv, dr, dc, dc, tv, R, C, C);
            free(du); free(dv);
            du = tu; dv = tv;
            dr = R; dc = C;
        }
        float *warped = (float *)malloc((size_t)R * C * sizeof(float));
        float *dx = (float *)malloc((size_t)R * C * sizeof(float));
        float *dy = (float *)malloc((size_t)R * C * sizeof(float));
        const int at = g_base_org_y | g_base_org_x;  /* (0 unless orc_lk_flow_pyr_at set them: then nothing is written here,
                                                       * so concurrent plain calls -- bench.py's all-cores baseline -- share no state) */
        if (at) { g_org_y = g_base_org_y >> k; g_org_x = g_base_org_x >> k; }
        orc_lk_warp(np[k], du, dv, R, C, C, warped);               /* :155 */
        if (at) { g_org_y = 0; g_org_x = 0; }
        rc = orc_lk_flow_ex(pp[k], warped, R, C, C, win, variant, dx, dy, C, k == 0 ? det0_out : NULL);  /* :159 */
        for (size_t i = 0; i < (size_t)R * C; i++) { du[i] = du[i] + dx[i]; dv[i] = dv[i] + dy[i]; } /* :161-162 */
        free(warped); free(dx); free(dy);
        if (rc) break;
    }
    if (!rc)
        for (int y = 0; y < rows; y++) {
            memcpy(u + (size_t)y * ostride, du + (size_t)y * cols, cols * sizeof(float));
            memcpy(v + (size_t)y * ostride, dv + (size_t)y * cols, cols * sizeof(float));
        }
    free(du); free(dv);
    for (int l = 0; l < levels; l++) { free(pp[l]); free(np[l]); }
    return rc;
}

/* cv::cvtColor(COLOR_RGB2GRAY) for CV_8UC3 (OpenCV 3.4.1 imgproc/src/color.cpp RGB2Gray<uchar>:
 * R2Y=4899, G2Y=9617, B2Y=1868, yuv_shift=14) then convertTo(CV_32F). Pyramids.cpp:10-15. */
/* The general colour entry of the path: pyr::makeGaussianPyramid (ps5_cpp/lib/Pyramids.cpp:9-15)
 * runs cv::cvtColor(COLOR_RGB2GRAY) on anything with more than one channel and then
 * convertTo(CV_32F); denseLKWrapper (ps5_cpp/src/Solution.cpp:48-56) does the same for the
 * single-level mode.  OpenCV 3.4 semantics restated (not verifiable here):
 *   CV_8U : (c0*4899 + c1*9617 + c2*1868 + 2^13) >> 14, 3 or 4 channels (alpha ignored);
 *   CV_32F: (c0*0.299f + c1*0.587f) + c2*0.114f, unfused, left to right;
 *   one channel: the conversion to float only.  depth: 0 = 8U, 5 = 32F (OpenCV's depth codes). */
int orc_to_gray_f32(const void *src, int rows, int cols, size_t sstride_bytes, int channels, int depth,
                    float *dst, size_t dstride) {
    if ((channels != 1 && channels != 3 && channels != 4) || (depth != 0 && depth != 5)) return -1;
    for (int y = 0; y < rows; y++) {
        const uint8_t *row = (const uint8_t *)src + (size_t)y * sstride_bytes;
        for (int x = 0; x < cols; x++) {
            float g;
            if (depth == 0) {
                const uint8_t *s = row + (size_t)channels * x;
                g = channels == 1 ? (float)s[0]
                                  : (float)((s[0] * 4899 + s[1] * 9617 + s[2] * 1868 + (1 << 13)) >> 14);
            } else {
                const float *s = (const float *)row + (size_t)channels * x;
                if (channels == 1) {
                    g = s[0];
                } else {
                    const float a = s[0] * 0.299f, b = s[1] * 0.587f, c = s[2] * 0.114f;
                    g = (a + b) + c;
                }
            }
            AT(dst, dstride, y, x) = g;
        }
    }
    return 0;
}

void orc_rgb8_to_gray_f32(const uint8_t *rgb, int rows, int cols, size_t sstride_bytes,
                          float *dst, size_t dstride) {
    for (int y = 0; y < rows; y++) {
        const uint8_t *s = rgb + (size_t)y * sstride_bytes;
        for (int x = 0; x < cols; x++) {
            int g = (s[3 * x] * 4899 + s[3 * x + 1] * 9617 + s[3 * x + 2] * 1868 + (1 << 13)) >> 14;
            AT(dst, dstride, y, x) = (float)g;
        }
    }
}
