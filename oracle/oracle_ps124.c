/*
 * oracle_ps124.c -- CPU restatement of the ps4 (Harris, SIFT-style keypoints), ps2 (window
 * stereo) and ps1 (Hough) kernels.  TEST INFRASTRUCTURE ONLY; parity unpinned (oracle.h).
 * Build with -ffp-contract=off (fused multiply-adds only where fmaf() is written).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define AT(p, stride, y, x) ((p)[(size_t)(y) * (stride) + (size_t)(x)])

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------ ps4: Harris ---- */

/* harris::getCornerResponse.  Window/border/weights as harris::cpu (Harris.cpp:61-76: clamped
 * coordinates, weights = outer product of getGaussianKernel(win, sigma, CV_32F) -- the float
 * product g[wy]*g[wx]), accumulation as harris::gpu (Harris.cu:36-43,85: M = fma(w, I, M) per
 * tap in (wy, wx) raster order; :87-91: trace, det, response in float, left to right). */
int orc_harris_response_ex(const float *gx, const float *gy, int rows, int cols, size_t stride,
                           int win, double sigma, float alpha, int mode, float *resp, size_t rstride) {
    if (win < 1 || (win & 1) == 0 || win > 63 || !(sigma > 0)) return -1;
    if (mode < 0 || mode > ORC_HARRIS_GPU_FMAD) return -1;
    float g[64];
    orc_gaussian_kernel(win, sigma, g);
    int r = win / 2;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float mxx = 0.f, mxy = 0.f, myy = 0.f;
            for (int wy = -r; wy <= r; wy++)
                for (int wx = -r; wx <= r; wx++) {
                    int yy = clampi(y + wy, 0, rows - 1), xx = clampi(x + wx, 0, cols - 1);
                    float ix = AT(gx, stride, yy, xx), iy = AT(gy, stride, yy, xx);
                    float w = g[wy + r] * g[wx + r]; /* gauss * gauss.t(), Harris.cpp:63 */
                    if (mode == ORC_HARRIS_CPU) {
                        /* Harris.cpp:81-87: gradVals holds float products; `secondMoment + weight * gradVals`
                         * is a MatExpr that cv::scaleAdd evaluates per element as a multiply then an add */
                        mxx = mxx + w * (ix * ix);
                        mxy = mxy + w * (ix * iy);
                        myy = myy + w * (iy * iy);
                    } else { /* Harris.cu:36-43,85: fma.rn.f32 weight, intensity, moment */
                        mxx = fmaf(w, ix * ix, mxx);
                        mxy = fmaf(w, ix * iy, mxy);
                        myy = fmaf(w, iy * iy, myy);
                    }
                }
            float trace = mxx + myy; /* cv::trace sums in double and is stored to float: the same value */
            float R;
            if (mode == ORC_HARRIS_CPU) {
                /* Harris.cpp:91-92: cv::determinant of a 2x2 CV_32F is det2 in double; `harrisScore * trace *
                 * trace` is float arithmetic; the difference is taken in double and stored to float */
                double det = (double)mxx * myy - (double)mxy * mxy;
                R = (float)(det - (double)(alpha * trace * trace));
            } else if (mode == ORC_HARRIS_GPU_FMAD) {
                /* Harris.cu:89-91 as nvcc's default -fmad=true would contract it: a*b - c*d -> fma(a, b, -(c*d)),
                 * det - (alpha*trace)*trace -> fma(-(alpha*trace), trace, det).  Bounding variant only. */
                float det = fmaf(mxx, myy, -(mxy * mxy));
                R = fmaf(-(alpha * trace), trace, det);
            } else {
                float det = mxx * myy - mxy * mxy;
                R = det - alpha * trace * trace;
            }
            AT(resp, rstride, y, x) = R;
        }
    return 0;
}

int orc_harris_response(const float *gx, const float *gy, int rows, int cols, size_t stride,
                        int win, double sigma, float alpha, float *resp, size_t rstride) {
    return orc_harris_response_ex(gx, gy, rows, cols, stride, win, sigma, alpha, ORC_HARRIS_GPU, resp, rstride);
}

/* harris::refineCorners, Harris.cpp:115-143: R >= threshold (double compare) and strictly
 * greater than every other pixel of the clamped (2d+1)^2 window; the reference's row skip
 * (:140) only skips pixels that cannot be maxima, so it does not change the result. */
int64_t orc_harris_refine(const float *resp, int rows, int cols, size_t stride,
                          double threshold, int min_distance,
                          float *corners, size_t cstride, int32_t *locs_yx, int64_t cap) {
    int64_t n = 0;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            float v = AT(resp, stride, y, x);
            int keep = 0;
            if ((double)v >= threshold) {
                keep = 1;
                for (int wy = -min_distance; wy <= min_distance && keep; wy++)
                    for (int wx = -min_distance; wx <= min_distance; wx++) {
                        int cy = clampi(y + wy, 0, rows - 1), cx = clampi(x + wx, 0, cols - 1);
                        if (cy == y && cx == x) continue;
                        if (v <= AT(resp, stride, cy, cx)) { keep = 0; break; }
                    }
            }
            AT(corners, cstride, y, x) = keep ? v : 0.f;
            /* Harris.cu:300-306 compacts pixels whose value is > 0; cpu:: pushes every kept
             * maximum.  They differ only for thresholds <= 0; we follow cpu::. */
            if (keep) {
                if (n < cap) { locs_yx[2 * n] = y; locs_yx[2 * n + 1] = x; }
                n++;
            }
        }
    return n;
}

void orc_sift_angles(const float *gx, const float *gy, int rows, int cols, size_t stride,
                     float *angles, size_t astride) {
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++)
            AT(angles, astride, y, x) = atan2f(AT(gy, stride, y, x), AT(gx, stride, y, x));
}

void orc_sift_keypoints(const float *gx, const float *gy, int rows, int cols, size_t stride,
                        const int32_t *locs_yx, int64_t n, float size, float *kp) {
    (void)rows; (void)cols;
    const float PI = 3.1415921636f; /* sic, Descriptors.cpp:5 */
    for (int64_t i = 0; i < n; i++) {
        int y = locs_yx[2 * i], x = locs_yx[2 * i + 1];
        float a = atan2f(AT(gy, stride, y, x), AT(gx, stride, y, x)) * 180.f / PI; /* :43 */
        kp[4 * i] = (float)x;  /* KeyPoint(x = corner.second, y = corner.first, size, angle), :45 */
        kp[4 * i + 1] = (float)y;
        kp[4 * i + 2] = size;
        kp[4 * i + 3] = a;
    }
}

/* ------------------------------------------------------------------ ps2: stereo ---- */

/* Clamp-to-edge fetch = the CUDA path's 2-D texture (DisparitySSD.cu:19-20, point, clamp). */
static float tex(const float *img, int rows, int cols, size_t stride, int x, int y) {
    return AT(img, stride, clampi(y, 0, rows - 1), clampi(x, 0, cols - 1));
}


/* The CUDA kernels' rolling column sums (ORC_STEREO_ROLLING): DisparitySSD.cu:27-141 and
 * DisparityNCorr.cu:28-175 restated strip by strip.  ncc = 0: SSD; ncc = 1: cross-correlation. */
#define ORC_STEREO_STRIP 40 /* ROWS_PER_THREAD, DisparitySSD.cu:17 / DisparityNCorr.cu:17 */
static int stereo_rolling(const float *left, const float *right, int rows, int cols, size_t stride,
                          int rad, int min_d, int max_d, int flags, int ncc, int8_t *disp, size_t dstride) {
    const int wcols = (flags & ORC_STEREO_COLS_2R) ? 2 * rad : 2 * rad + 1;
    const size_t n = (size_t)(cols + 2 * rad);
    float *cs = (float *)malloc(3 * n * sizeof(float));
    float *best = (float *)malloc((size_t)ORC_STEREO_STRIP * cols * sizeof(float));
    for (int y0 = 0; y0 < rows; y0 += ORC_STEREO_STRIP) {
        const int nr = rows - y0 < ORC_STEREO_STRIP ? rows - y0 : ORC_STEREO_STRIP;
        for (int j = 0; j < nr; j++)
            for (int x = 0; x < cols; x++) {
                best[(size_t)j * cols + x] = ncc ? 0.f : ((flags & ORC_STEREO_MIN_SSD_5E6) ? 5000000.f : INFINITY);
                AT(disp, dstride, y0 + j, x) = -1;
            }
        for (int d = min_d; d <= max_d; d++) {
            for (int j = 0; j < nr; j++) {
                const int y = y0 + j;
                for (int xc = -rad; xc < cols + rad; xc++) {
                    float *p = cs + (xc + rad), *aa = p + n, *bb = p + 2 * n;
                    if (j == 0) { /* :68-81 / :77-95: from 0, top -> bottom */
                        *p = 0.f; *aa = 0.f; *bb = 0.f;
                        for (int wy = -rad; wy <= rad; wy++) {
                            float a = tex(left, rows, cols, stride, xc, y + wy);
                            float b = tex(right, rows, cols, stride, xc + d, y + wy);
                            if (ncc) { *p += a * b; *aa += a * a; *bb += b * b; }
                            else { float diff = a - b; *p += diff * diff; }
                        }
                    } else { /* :101-109 / :120-134: minus the row that left, plus the row that entered */
                        float a = tex(left, rows, cols, stride, xc, y - 1 - rad);
                        float b = tex(right, rows, cols, stride, xc + d, y - 1 - rad);
                        if (ncc) { *p -= a * b; *aa -= a * a; *bb -= b * b; }
                        else { float diff = a - b; *p -= diff * diff; }
                        a = tex(left, rows, cols, stride, xc, y + rad);
                        b = tex(right, rows, cols, stride, xc + d, y + rad);
                        if (ncc) { *p += a * b; *aa += a * a; *bb += b * b; }
                        else { float diff = a - b; *p += diff * diff; }
                    }
                }
                for (int x = 0; x < cols; x++) {
                    float tot = 0.f, at = 0.f, ai = 0.f;
                    for (int i = 0; i < wcols; i++) { /* left -> right */
                        tot += cs[x + i];
                        if (ncc) { at += cs[n + x + i]; ai += cs[2 * n + x + i]; }
                    }
                    float *b = best + (size_t)j * cols + x;
                    if (ncc) {
                        float nc = tot / sqrtf(at * ai); /* DisparityNCorr.cu:164 */
                        if (nc > *b) { *b = nc; AT(disp, dstride, y, x) = (int8_t)d; }
                    } else if (tot < *b) { /* DisparitySSD.cu:133 */
                        *b = tot;
                        AT(disp, dstride, y, x) = (int8_t)d;
                    }
                }
            }
        }
    }
    free(cs);
    free(best);
    return 0;
}

int orc_disparity_ssd(const float *left, const float *right, int rows, int cols, size_t stride,
                      int rad, int min_d, int max_d, int flags, int8_t *disp, size_t dstride) {
    if (rad < 0 || rad > 31 || min_d > max_d || min_d < -128 || max_d > 127) return -1;
    if (flags & ORC_STEREO_ROLLING)
        return stereo_rolling(left, right, rows, cols, stride, rad, min_d, max_d, flags, 0, disp, dstride);
    int wcols = (flags & ORC_STEREO_COLS_2R) ? 2 * rad : 2 * rad + 1; /* DisparitySSD.cu:84 */
    float *colsum = (float *)malloc((size_t)(cols + 2 * rad) * sizeof(float));
    for (int y = 0; y < rows; y++) {
        float *best = (float *)malloc((size_t)cols * sizeof(float));
        for (int x = 0; x < cols; x++) {
            best[x] = (flags & ORC_STEREO_MIN_SSD_5E6) ? 5000000.f : INFINITY;
            AT(disp, dstride, y, x) = -1; /* DisparitySSD.cu:177 */
        }
        for (int d = min_d; d <= max_d; d++) { /* ascending, :56 */
            for (int xc = -rad; xc < cols + rad; xc++) { /* column sums, top -> bottom, :68-81 */
                float s = 0.f;
                for (int wy = -rad; wy <= rad; wy++) {
                    float diff = tex(left, rows, cols, stride, xc, y + wy) -
                                 tex(right, rows, cols, stride, xc + d, y + wy);
                    s += diff * diff;
                }
                colsum[xc + rad] = s;
            }
            for (int x = 0; x < cols; x++) {
                float ssd = 0.f;
                for (int i = 0; i < wcols; i++) ssd += colsum[x + i]; /* :84-86, left -> right */
                if (ssd < best[x]) { /* strict: lowest d wins ties, :88 */
                    best[x] = ssd;
                    AT(disp, dstride, y, x) = (int8_t)d;
                }
            }
        }
        free(best);
    }
    free(colsum);
    return 0;
}

/* serial::disparitySSD as written, DisparitySSD.cpp:35-61. */
int orc_disparity_ssd_serial(const float *left, const float *right, int rows, int cols,
                             size_t stride, int rad, int min_d, int max_d,
                             int8_t *disp, size_t dstride) {
    if (rad < 0 || rad > 31) return -1;
    int prow = rows + 2 * rad, pcol = cols + 2 * rad;
    /* copyMakeBorder(BORDER_REPLICATE) == clamped fetch at (y - rad, x - rad) */
    for (int y = rad; y < prow - rad; y++)
        for (int x = rad; x < pcol - rad; x++) {
            int bestCost = 99999999, bestDisparity = 0;
            int searchIndex = (int)fmax(0, x + min_d);
            int maxSearchIndex = (int)fmin(pcol - 1, x + max_d);
            for (; searchIndex <= maxSearchIndex; searchIndex++) {
                int sum = 0;
                for (int winY = -rad; winY <= rad; winY++)
                    for (int winX = -rad; winX <= rad; winX++) {
                        /* reads past the padded image (searchIndex + winX outside [0, pcol)) are
                         * undefined in the reference; clamped here */
                        int rx = clampi(searchIndex + winX, 0, pcol - 1);
                        float rawCost = tex(left, rows, cols, stride, x + winX - rad, y + winY - rad) -
                                        tex(right, rows, cols, stride, rx - rad, y + winY - rad);
                        sum += (int)round(rawCost * rawCost);
                    }
                if (sum < bestCost) { bestCost = sum; bestDisparity = searchIndex - x; }
            }
            AT(disp, dstride, y - rad, x - rad) = (int8_t)bestDisparity;
        }
    return 0;
}

int orc_disparity_ncorr(const float *left, const float *right, int rows, int cols, size_t stride,
                        int rad, int min_d, int max_d, int flags, int8_t *disp, size_t dstride) {
    if (rad < 0 || rad > 31 || min_d > max_d || min_d < -128 || max_d > 127) return -1;
    if (flags & ORC_STEREO_ROLLING)
        return stereo_rolling(left, right, rows, cols, stride, rad, min_d, max_d, flags, 1, disp, dstride);
    int wcols = (flags & ORC_STEREO_COLS_2R) ? 2 * rad : 2 * rad + 1; /* DisparityNCorr.cu:99 */
    size_t n = (size_t)(cols + 2 * rad);
    float *cs = (float *)malloc(3 * n * sizeof(float));
    float *best = (float *)malloc((size_t)cols * sizeof(float));
    for (int y = 0; y < rows; y++) {
        for (int x = 0; x < cols; x++) {
            best[x] = 0.f; /* DisparityNCorr.cu:16,212 */
            AT(disp, dstride, y, x) = -1;
        }
        for (int d = min_d; d <= max_d; d++) {
            for (int xc = -rad; xc < cols + rad; xc++) {
                float p = 0.f, aa = 0.f, bb = 0.f;
                for (int wy = -rad; wy <= rad; wy++) { /* :83-96 */
                    float a = tex(left, rows, cols, stride, xc, y + wy);
                    float b = tex(right, rows, cols, stride, xc + d, y + wy);
                    p += a * b;
                    aa += a * a;
                    bb += b * b;
                }
                cs[xc + rad] = p; cs[n + xc + rad] = aa; cs[2 * n + xc + rad] = bb;
            }
            for (int x = 0; x < cols; x++) {
                float nc = 0.f, at = 0.f, ai = 0.f;
                for (int i = 0; i < wcols; i++) { /* :99-103 */
                    nc += cs[x + i];
                    at += cs[n + x + i];
                    ai += cs[2 * n + x + i];
                }
                nc = nc / sqrtf(at * ai); /* :106 */
                if (nc > best[x]) { /* :108, first max wins */
                    best[x] = nc;
                    AT(disp, dstride, y, x) = (int8_t)d;
                }
            }
        }
    }
    free(cs);
    free(best);
    return 0;
}

/* ------------------------------------------------------------------- ps1: Hough ---- */

/* degToRad (Hough.cu:20-24): theta * PI / 180.f with PI the DOUBLE literal 3.14159265, result
 * returned as float.  cos/sin: the reference's __sincosf is a hardware approximation that
 * cannot be reproduced; the contract here is correctly-rounded-double libm, cast to float. */
void orc_hough_trig_table(float *cos360, float *sin360) {
    for (int i = 0; i < 360; i++) {
        int theta = i - 90;
        float rad = (float)((double)(float)theta * 3.14159265 / 180.f);
        cos360[i] = (float)cos((double)rad);
        sin360[i] = (float)sin((double)rad);
    }
}

void orc_hough_lines_dims(int rows, int cols, unsigned rho_bin, unsigned theta_bin,
                          int *rho_bins, int *theta_bins) {
    size_t maxDist = (size_t)ceil(sqrt((double)(rows * rows + cols * cols))); /* Hough.cu:258-259 */
    size_t rb = (size_t)ceilf((float)(2 * maxDist) / (float)rho_bin);
    size_t tb = (size_t)ceilf(180.f / (float)theta_bin);
    *rho_bins = (int)(rb < 1 ? 1 : rb);
    *theta_bins = (int)(tb < 1 ? 1 : tb);
}

int orc_hough_lines(const uint8_t *mask, int rows, int cols, size_t stride,
                    unsigned rho_bin, unsigned theta_bin, int32_t *acc) {
    if (rho_bin == 0 || theta_bin == 0) return -1;
    int rb, tb;
    orc_hough_lines_dims(rows, cols, rho_bin, theta_bin, &rb, &tb);
    float ct[360], st[360];
    orc_hough_trig_table(ct, st);
    size_t diag = (size_t)ceil(sqrt((double)(rows * rows + cols * cols)));
    memset(acc, 0, (size_t)rb * tb * sizeof(int32_t));
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            if (!mask[(size_t)y * stride + x]) continue; /* IsNonzero, Hough.cu:183-187 */
            for (int theta = -90; theta < 90; theta += (int)theta_bin) { /* :51 */
                float c = ct[theta + 90], s = st[theta + 90];
                float rho = roundf((float)x * c + (float)y * s) + (float)diag; /* :54 */
                int rhoBin = (int)roundf(rho / (float)rho_bin);                 /* :55 */
                int thetaBin = (int)roundf(((float)theta - -90.f) / (float)theta_bin); /* :56 */
                /* the reference does not range-check; out-of-range votes are dropped here */
                if (rhoBin >= 0 && rhoBin < rb && thetaBin >= 0 && thetaBin < tb)
                    acc[(size_t)rhoBin * tb + thetaBin] += 1;
            }
        }
    return 0;
}

/* float -> unsigned as CUDA converts it (cvt.rzi.u32.f32): truncate, negatives/NaN -> 0. */
static unsigned f2u_sat(float v) {
    if (!(v > 0.f)) return 0u;
    if (v >= 4294967296.f) return 0xFFFFFFFFu;
    return (unsigned)v;
}

int orc_hough_circles(const uint8_t *mask, int rows, int cols, size_t stride,
                      unsigned radius, int32_t *acc) {
    float ct[360], st[360];
    memset(acc, 0, (size_t)rows * cols * sizeof(int32_t)); /* the reference forgets to, :318 */
    for (int t = 0; t < 360; t++) { /* degToRad(theta), theta = 0..359, Hough.cu:85-86 */
        float rad = (float)((double)(float)t * 3.14159265 / 180.f);
        ct[t] = (float)cos((double)rad);
        st[t] = (float)sin((double)rad);
    }
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            if (!mask[(size_t)y * stride + x]) continue;
            for (int t = 0; t < 360; t++) {
                unsigned a = f2u_sat((float)x - (float)radius * ct[t]); /* :87 */
                unsigned b = f2u_sat((float)y - (float)radius * st[t]); /* :88 */
                if (a < (unsigned)cols && b < (unsigned)rows && a > 0 && b > 0) /* :91 */
                    acc[(size_t)b * cols + a] += 1;
            }
        }
    return 0;
}

typedef struct { int32_t votes; uint32_t idx; } peak_t;
static int peak_cmp(const void *pa, const void *pb) {
    const peak_t *a = (const peak_t *)pa, *b = (const peak_t *)pb;
    if (a->votes != b->votes) return a->votes > b->votes ? -1 : 1; /* votes descending */
    return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0);       /* stable: original order */
}

int64_t orc_hough_peaks(const int32_t *acc, int rows, int cols, unsigned num_peaks,
                        int threshold, uint32_t *peaks_rc) {
    peak_t *cand = (peak_t *)malloc((size_t)rows * cols * sizeof(peak_t));
    size_t n = 0;
    for (int ty = 0; ty < rows; ty++)
        for (int tx = 0; tx < cols; tx++) {
            int v = acc[(size_t)ty * cols + tx];
            int is_max = 1;
            /* Hough.cu:150-153: exclusive upper bounds -> only the up/left 2x2 block */
            int y1 = (rows - 1 < ty + 1) ? rows - 1 : ty + 1;
            int x1 = (cols - 1 < tx + 1) ? cols - 1 : tx + 1;
            for (int y = (ty - 1 > 0 ? ty - 1 : 0); y < y1; y++)
                for (int x = (tx - 1 > 0 ? tx - 1 : 0); x < x1; x++)
                    if (acc[(size_t)y * cols + x] > v) is_max = 0;
            if (is_max && v >= threshold) { /* MaskAndThreshold, :239-249 */
                cand[n].votes = v;
                cand[n].idx = (uint32_t)((size_t)ty * cols + tx);
                n++;
            }
        }
    qsort(cand, n, sizeof(peak_t), peak_cmp); /* thrust::stable_sort(greater), :402 */
    size_t take = n < num_peaks ? n : num_peaks;
    for (size_t i = 0; i < take; i++) {
        peaks_rc[2 * i] = cand[i].idx / (uint32_t)cols;
        peaks_rc[2 * i + 1] = cand[i].idx % (uint32_t)cols;
    }
    free(cand);
    return (int64_t)take;
}
