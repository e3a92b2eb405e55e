/*
 * oracle_ps7.c -- CPU restatement of the ps7 motion-history kernels.
 * TEST INFRASTRUCTURE ONLY; parity unpinned (oracle.h).  Single-channel 8-bit frames.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AT(p, stride, y, x) ((p)[(size_t)(y) * (stride) + (size_t)(x)])

static uint8_t sat_u8_rn(float v) { /* saturate_cast<uchar>(float): round-half-even, clamp */
    long r = lrintf(v);
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

/* cv::cuda::createGaussianFilter(CV_8UC1, -1, ksize, sigma) -> separable filter with a CV_32F
 * buffer: row pass u8 -> float (fmaf chain), column pass float -> saturate_cast<uchar>;
 * BORDER_REFLECT_101 (MotionHistory.cpp:50-52). */
static void gauss_u8(const uint8_t *src, int rows, int cols, size_t stride, int kw, int kh, double sigma,
                     uint8_t *dst) {
    float k[64], ky[64]; /* cv::Size(kw, kh): kw taps along x, kh taps along y, one sigma */
    orc_gaussian_kernel(kw, sigma, k);
    orc_gaussian_kernel(kh, sigma, ky);
    float *buf = (float *)malloc((size_t)rows * cols * sizeof(float));
    int a = kw / 2, ay = kh / 2;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float acc = 0.f;
            for (int i = 0; i < kw; i++)
                acc = fmaf((float)AT(src, stride, y, orc_reflect101(x + i - a, cols)), k[i], acc);
            buf[(size_t)y * cols + x] = acc;
        }
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float acc = 0.f;
            for (int i = 0; i < kh; i++)
                acc = fmaf(buf[(size_t)orc_reflect101(y + i - ay, rows) * cols + x], ky[i], acc);
            dst[(size_t)y * cols + x] = sat_u8_rn(acc);
        }
    free(buf);
}

/* cv::getStructuringElement(MORPH_ELLIPSE, Size(7,7)): per row the half-width
 * dx = cvRound(c * sqrt((r*r - dy*dy) * inv_r2)). */
void orc_ellipse7(uint8_t m[7][7]) {
    const int r = 3, c = 3;
    const double inv_r2 = 1.0 / ((double)r * r);
    for (int i = 0; i < 7; i++) {
        int dy = i - r;
        int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
        int j1 = c - dx < 0 ? 0 : c - dx, j2 = c + dx + 1 > 7 ? 7 : c + dx + 1;
        for (int j = 0; j < 7; j++) m[i][j] = (j >= j1 && j < j2) ? 1 : 0;
    }
}

/* One pass of cv::cuda morphology (copyMakeBorder BORDER_REFLECT_101 + NPP erode/dilate with the
 * element anchored at its centre). */
static void morph7(const uint8_t *src, int rows, int cols, int dilate, uint8_t *dst) {
    uint8_t m[7][7];
    orc_ellipse7(m);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            int v = dilate ? 0 : 255;
            for (int i = 0; i < 7; i++)
                for (int j = 0; j < 7; j++) {
                    if (!m[i][j]) continue;
                    int s = src[(size_t)orc_reflect101(y + i - 3, rows) * cols + orc_reflect101(x + j - 3, cols)];
                    v = dilate ? (s > v ? s : v) : (s < v ? s : v);
                }
            dst[(size_t)y * cols + x] = (uint8_t)v;
        }
}

/* thresholdDifference / AbsThreshold<uint8_t>, MotionHistory.cu:17-48. */
void orc_mhi_threshold(const uint8_t *src, size_t n, double thresh, uint8_t *dst) {
    for (size_t i = 0; i < n; i++) {
        int val = src[i];
        dst[i] = ((double)val >= thresh || (double)(-val) >= thresh) ? 1 : 0;
    }
}

/* mhi::frameDifference, MotionHistory.cpp:26-77, single-channel CV_8U frames. */
int orc_mhi_frame_difference(const uint8_t *f1, const uint8_t *f2, int rows, int cols, size_t stride,
                             double thresh, int kw, int kh, double sigma, uint8_t *diff, size_t dstride) {
    if (kw < 1 || kw > 31 || (kw & 1) == 0 || kh < 1 || kh > 31 || (kh & 1) == 0 || !(sigma > 0)) return -1;
    size_t n = (size_t)rows * cols;
    uint8_t *b1 = (uint8_t *)malloc(4 * n), *b2 = b1 + n, *d = b1 + 2 * n, *t = b1 + 3 * n;
    gauss_u8(f1, rows, cols, stride, kw, kh, sigma, b1); /* :50-52 */
    gauss_u8(f2, rows, cols, stride, kw, kh, sigma, b2);
    for (size_t i = 0; i < n; i++) { /* cv::cuda::subtract on CV_8U saturates, :56 */
        int v = (int)b2[i] - (int)b1[i];
        d[i] = (uint8_t)(v < 0 ? 0 : v);
    }
    orc_mhi_threshold(d, n, thresh, t); /* :69 */
    morph7(t, rows, cols, 0, d);        /* MORPH_OPEN = erode ... */
    morph7(d, rows, cols, 1, t);        /* ... then dilate, :53-54,73 */
    for (int y = 0; y < rows; y++) memcpy(diff + (size_t)y * dstride, t + (size_t)y * cols, cols);
    free(b1);
    return 0;
}

/* mhi::calcMotionHistory -> motionHistoryKernel, MotionHistory.cu:52-66. */
void orc_mhi_update(uint8_t *history, size_t hstride, const uint8_t *mask, size_t mstride, int rows,
                    int cols, int tau) {
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            int h = AT(history, hstride, y, x);
            AT(history, hstride, y, x) =
                (uint8_t)(AT(mask, mstride, y, x) == 1 ? tau : (h - 1 > 0 ? h - 1 : 0));
        }
}

/* mhi::energyFromHistory, MotionHistory.cpp:98-105. */
void orc_mhi_energy(const uint8_t *mhi, size_t n, uint8_t *mei) {
    for (size_t i = 0; i < n; i++) mei[i] = mhi[i] > 0 ? 1 : 0;
}
