/*
 * oracle_canny.c -- CPU restatement of the edge front-end of ps1:
 * sol::generateEdge, ps1_cpp/src/Solution.cpp:21-47 = cv::cuda Gaussian blur (CV_8U) + Canny.
 * TEST INFRASTRUCTURE ONLY; parity unpinned (oracle.h).  The Canny definition is OpenCV 3.4's
 * (imgproc/src/canny.cpp; cv::cuda's kernel uses the same constants): 3x3 Sobel on the 8-bit image
 * with replicated border, L1 magnitude, non-maximum suppression along the gradient direction
 * quantised with the fixed-point tangents TG22 / TG67, double threshold, 8-connected hysteresis.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cv::cuda::createGaussianFilter on CV_8UC1 (same contract as oracle_ps7.c's blur). */
void orc_gauss_u8(const uint8_t *src, int rows, int cols, size_t stride, int ksize, double sigma,
                  uint8_t *dst, size_t dstride) {
    float k[64];
    if (ksize == 1) { /* a 1-tap kernel normalises to exactly 1 */
        for (int y = 0; y < rows; y++) memcpy(dst + (size_t)y * dstride, src + (size_t)y * stride, cols);
        return;
    }
    orc_gaussian_kernel(ksize, sigma, k);
    float *buf = (float *)malloc((size_t)rows * cols * sizeof(float));
    int a = ksize / 2;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float acc = 0.f;
            for (int i = 0; i < ksize; i++)
                acc = fmaf((float)src[(size_t)y * stride + orc_reflect101(x + i - a, cols)], k[i], acc);
            buf[(size_t)y * cols + x] = acc;
        }
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float acc = 0.f;
            for (int i = 0; i < ksize; i++)
                acc = fmaf(buf[(size_t)orc_reflect101(y + i - a, rows) * cols + x], k[i], acc);
            long r = lrintf(acc);
            dst[(size_t)y * dstride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    free(buf);
}

/* Canny on an 8-bit image; edges = 255 / 0.  low / high are compared with the integer L1 magnitude
 * after flooring, as cv::Canny does for L2gradient = false. */
int orc_canny(const uint8_t *src, int rows, int cols, size_t stride, double low_thresh,
              double high_thresh, uint8_t *edges, size_t estride) {
    if (low_thresh > high_thresh) { double t = low_thresh; low_thresh = high_thresh; high_thresh = t; }
    const int low = (int)floor(low_thresh), high = (int)floor(high_thresh);
    size_t n = (size_t)rows * cols;
    int *mag = (int *)malloc(n * sizeof(int));
    short *dx = (short *)malloc(n * sizeof(short)), *dy = (short *)malloc(n * sizeof(short));
    uint8_t *map = (uint8_t *)malloc(n); /* 0 = no edge, 1 = candidate, 2 = edge */
#define S(yy, xx) ((int)src[(size_t)clampi(yy, 0, rows - 1) * stride + clampi(xx, 0, cols - 1)])
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            int gx = (S(y - 1, x + 1) + 2 * S(y, x + 1) + S(y + 1, x + 1)) -
                     (S(y - 1, x - 1) + 2 * S(y, x - 1) + S(y + 1, x - 1));
            int gy = (S(y + 1, x - 1) + 2 * S(y + 1, x) + S(y + 1, x + 1)) -
                     (S(y - 1, x - 1) + 2 * S(y - 1, x) + S(y - 1, x + 1));
            dx[(size_t)y * cols + x] = (short)gx;
            dy[(size_t)y * cols + x] = (short)gy;
            mag[(size_t)y * cols + x] = abs(gx) + abs(gy);
        }
#undef S
#define M(yy, xx) (((yy) < 0 || (yy) >= rows || (xx) < 0 || (xx) >= cols) ? 0 : mag[(size_t)(yy) * cols + (xx)])
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            size_t i = (size_t)y * cols + x;
            int m = mag[i];
            map[i] = 0;
            if (m > low) {
                int xs = dx[i], ys = dy[i];
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * TG22;
                int is_max;
                if (ay < tg22x)
                    is_max = m > M(y, x - 1) && m >= M(y, x + 1);
                else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x)
                        is_max = m > M(y - 1, x) && m >= M(y + 1, x);
                    else {
                        int s = (xs ^ ys) < 0 ? -1 : 1;
                        is_max = m > M(y - 1, x - s) && m > M(y + 1, x + s);
                    }
                }
                if (is_max) map[i] = m > high ? 2 : 1;
            }
        }
#undef M
    /* hysteresis: everything 8-connected to a strong pixel through candidates */
    size_t *stack = (size_t *)malloc(n * sizeof(size_t)), sp = 0;
    for (size_t i = 0; i < n; i++)
        if (map[i] == 2) stack[sp++] = i;
    while (sp) {
        size_t i = stack[--sp];
        int y = (int)(i / cols), x = (int)(i % cols);
        for (int yy = y - 1; yy <= y + 1; yy++)
            for (int xx = x - 1; xx <= x + 1; xx++) {
                if (yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
                size_t j = (size_t)yy * cols + xx;
                if (map[j] == 1) { map[j] = 2; stack[sp++] = j; }
            }
    }
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) edges[(size_t)y * estride + x] = map[(size_t)y * cols + x] == 2 ? 255 : 0;
    free(mag); free(dx); free(dy); free(map); free(stack);
    return 0;
}

/* sol::generateEdge, Solution.cpp:21-47 on a single-channel 8-bit image. */
int orc_generate_edge(const uint8_t *src, int rows, int cols, size_t stride, int gauss_size,
                      double gauss_sigma, double low_thresh, double high_thresh, uint8_t *edges,
                      size_t estride) {
    if (gauss_size < 1 || gauss_size > 31 || (gauss_size & 1) == 0 || !(gauss_sigma > 0)) return -1;
    uint8_t *blur = (uint8_t *)malloc((size_t)rows * cols);
    orc_gauss_u8(src, rows, cols, stride, gauss_size, gauss_sigma, blur, cols);
    int rc = orc_canny(blur, rows, cols, cols, low_thresh, high_thresh, edges, estride);
    free(blur);
    return rc;
}
