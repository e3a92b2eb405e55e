/* oracle_sift.c -- CPU restatement of the SIFT-style descriptor window (TEST INFRASTRUCTURE ONLY).
 *
 * Reference call site: Solution::siftHelper, ProblemSets/ps4_cpp/src/Solution.cpp:166-169
 *   auto sift = cv::xfeatures2d::SIFT::create();  sift->compute(img, keypoints, descriptors);
 * on the keypoints of sift::getKeypoints (ps4_cpp/lib/Descriptors.cpp:27-47: x = col, y = row,
 * size = 10, angle = atan2(Iy, Ix) in degrees, octave 0).
 *
 * PARITY UNPINNED.  cv::xfeatures2d::SIFT is third-party code (opencv_contrib 3.4.1, not in
 * /root/reference, not installable here) and it builds its own Gaussian scale space from the image
 * before it samples gradients.  What is restated here is the published per-keypoint algorithm of
 * that implementation (calcSIFTDescriptor in xfeatures2d/src/sift.cpp; Lowe, IJCV 2004 section 6):
 * 4 x 4 spatial bins x 8 orientation bins, window rotated by the keypoint angle, bin width
 * 3 * size / 2 pixels, Gaussian weight exp(-(r^2 + c^2) / 8) in bin units, trilinear distribution,
 * L2 normalise -> clamp at 0.2 -> renormalise to 512 -> saturate to 8 bits -- SAMPLED ON THE
 * GRADIENT FIELDS OF harris::getGradients (ps4_cpp/lib/Harris.cpp:14-41, SURVEY.md section 7 step 8: "own
 * 4x4x8 spec on the gradient fields") instead of on SIFT's internal pyramid.  Decisions that make
 * the result reproducible bit for bit on any IEEE-754 machine:
 *   - cos / sin of the keypoint angle and exp() of the Gaussian weight are fixed polynomial
 *     evaluations written out below (fmaf chains), not libm calls;
 *   - orientation = cv::fastAtan2's published polynomial (degrees), unfused;
 *   - the histogram is accumulated in 64-bit fixed point at 2^-40 of a bound on the window's gradient
 *     magnitudes (2 x the largest |gx| or |gy| in the window's bounding square), so the sum does
 *     not depend on the order of the samples;
 *   - the two 128-term norms are summed left to right in float.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define AT(p, stride, y, x) ((p)[(size_t)(y) * (stride) + (size_t)(x)])
#define SIFT_D 4
#define SIFT_N 8

/* sin and cos of `deg` degrees: quadrant by exact float arithmetic, Taylor polynomials of the
 * in-quadrant angle y in [0, pi/2) evaluated as fmaf chains. */
static void sincos_deg(float deg, float *s, float *c) {
    float t = deg / 360.f;
    t = t - floorf(t);           /* [0, 1] */
    const float x = t * 4.f;     /* quadrants, [0, 4] */
    int q = (int)x;
    const float f = x - (float)q;
    q &= 3;
    const float y = f * 1.57079632679489662f, y2 = y * y;
    float ps = -2.50521083854417188e-8f;                 /* -1/11! */
    ps = fmaf(ps, y2, 2.75573192239858907e-6f);          /*  1/9!  */
    ps = fmaf(ps, y2, -1.98412698412698413e-4f);         /* -1/7!  */
    ps = fmaf(ps, y2, 8.33333333333333333e-3f);          /*  1/5!  */
    ps = fmaf(ps, y2, -1.66666666666666667e-1f);         /* -1/3!  */
    ps = fmaf(ps, y2, 1.f);
    const float sy = ps * y;
    float pc = 2.08767569878680990e-9f;                  /*  1/12! */
    pc = fmaf(pc, y2, -2.75573192239858907e-7f);         /* -1/10! */
    pc = fmaf(pc, y2, 2.48015873015873016e-5f);          /*  1/8!  */
    pc = fmaf(pc, y2, -1.38888888888888889e-3f);         /* -1/6!  */
    pc = fmaf(pc, y2, 4.16666666666666667e-2f);          /*  1/4!  */
    pc = fmaf(pc, y2, -0.5f);
    pc = fmaf(pc, y2, 1.f);
    switch (q) {
        case 0: *s = sy; *c = pc; break;
        case 1: *s = pc; *c = -sy; break;
        case 2: *s = -sy; *c = -pc; break;
        default: *s = -pc; *c = sy; break;
    }
}

/* exp(w) for w <= 0: 2^k * P(f) with k = rint(w * log2 e), f the remainder in [-0.5, 0.5]. */
static float exp_neg(float w) {
    if (w < -80.f) return 0.f;
    const float t = w * 1.44269504088896341f;
    const float k = rintf(t);
    const float f = t - k;
    float p = 1.52527338040598403e-5f;                   /* ln2^7 / 7! */
    p = fmaf(p, f, 1.54035303933816099e-4f);             /* ln2^6 / 6! */
    p = fmaf(p, f, 1.33335581464284434e-3f);             /* ln2^5 / 5! */
    p = fmaf(p, f, 9.61812910762847716e-3f);             /* ln2^4 / 4! */
    p = fmaf(p, f, 5.55041086648215800e-2f);             /* ln2^3 / 3! */
    p = fmaf(p, f, 2.40226506959100712e-1f);             /* ln2^2 / 2! */
    p = fmaf(p, f, 6.93147180559945309e-1f);             /* ln2 */
    p = fmaf(p, f, 1.f);
    return ldexpf(p, (int)k);
}

/* cv::fastAtan2 (OpenCV 3.4 core/src/mathfuncs_core.cpp): degrees in [0, 360). */
static float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * 57.29577951308232f, p3 = -0.3258083974640975f * 57.29577951308232f,
                p5 = 0.1555786518463281f * 57.29577951308232f, p7 = -0.04432655554792128f * 57.29577951308232f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

typedef struct {
    float cos_t, sin_t, ori;
    int radius, px, py;
} sift_geom;

static void sift_geometry(const float *kp, int rows, int cols, sift_geom *g) {
    g->px = (int)lrintf(kp[0]);  /* Point pt(cvRound(ptf.x), cvRound(ptf.y)) */
    g->py = (int)lrintf(kp[1]);
    /* The descriptor runs in SIFT's y-up angle convention: ori = 360 - kpt.angle (SIFT::compute).
     * cv::KeyPoint::angle is meant to lie in [0, 360); sift::getKeypoints stores atan2 * 180 / PI in
     * (-180, 180] (Descriptors.cpp:43), for which OpenCV's orientation bin index runs out of range.
     * Decision: ori is reduced to [0, 360) first, so every keypoint angle is well defined. */
    float ori = 360.f - kp[3];
    ori = ori - 360.f * floorf(ori / 360.f);
    if (!(ori < 360.f)) ori = 0.f;
    g->ori = ori;
    const float scl = kp[2] * 0.5f, hist_width = 3.f * scl;  /* SIFT_DESCR_SCL_FCTR = 3 */
    const float rf = hist_width * 1.4142135623730951f * (float)(SIFT_D + 1) * 0.5f;
    const int diag = (int)lrint(sqrt((double)cols * cols + (double)rows * rows));
    int radius = rf < (float)diag ? (int)lrintf(rf) : diag; /* clipped to the image diagonal */
    if (radius > diag) radius = diag;
    if (radius < 0) radius = 0;
    g->radius = radius;
    float s, c;
    sincos_deg(ori, &s, &c);
    g->cos_t = c / hist_width;
    g->sin_t = s / hist_width;
}

/* One sample of the window: returns 0 when it does not contribute. */
static int sift_sample(const float *gx, const float *gy, int rows, int cols, size_t stride, const sift_geom *g,
                       int i, int j, float *rbin, float *cbin, float *dx, float *dy, float *wexp) {
    const float c_rot = (float)j * g->cos_t - (float)i * g->sin_t;
    const float r_rot = (float)j * g->sin_t + (float)i * g->cos_t;
    *rbin = r_rot + (float)(SIFT_D / 2) - 0.5f;
    *cbin = c_rot + (float)(SIFT_D / 2) - 0.5f;
    const int r = g->py + i, c = g->px + j;
    if (!(*rbin > -1.f && *rbin < (float)SIFT_D && *cbin > -1.f && *cbin < (float)SIFT_D && r > 0 &&
          r < rows - 1 && c > 0 && c < cols - 1))
        return 0;
    *dx = AT(gx, stride, r, c);
    *dy = -AT(gy, stride, r, c);  /* SIFT's dy is "up minus down"; the Sobel field is d/dy downwards */
    *wexp = (c_rot * c_rot + r_rot * r_rot) * (-1.f / ((float)(SIFT_D * SIFT_D) * 0.5f));
    return 1;
}

int orc_sift_descriptors(const float *gx, const float *gy, int rows, int cols, size_t stride,
                         const float *kp_xysa, int64_t n, float *desc, size_t dstride) {
    const int HS = (SIFT_D + 2) * (SIFT_D + 2) * (SIFT_N + 2);
    for (int64_t k = 0; k < n; k++) {
        const float *kp = kp_xysa + 4 * k;
        float *dst = desc + (size_t)k * dstride;
        for (int t = 0; t < SIFT_D * SIFT_D * SIFT_N; t++) dst[t] = 0.f;
        /* a keypoint without a positive finite size / finite position and angle: all-zero descriptor */
        if (!(kp[2] > 0.f) || !isfinite(kp[2]) || !isfinite(kp[0]) || !isfinite(kp[1]) || !isfinite(kp[3]) ||
            !(fabsf(kp[0]) < 1e9f) || !(fabsf(kp[1]) < 1e9f))
            continue;
        sift_geom g;
        sift_geometry(kp, rows, cols, &g);
        /* pass 1: a bound on every contribution fixes the scale of the fixed-point accumulators:
         * mag = sqrt(dx^2 + dy^2) <= 2 max(|dx|, |dy|), maximised over the window's bounding square
         * inside the image interior (a max: no order, no rotation, no square root) */
        float bound = 0.f;
        for (int i = -g.radius; i <= g.radius; i++)
            for (int j = -g.radius; j <= g.radius; j++) {
                const int r = g.py + i, c = g.px + j;
                if (!(r > 0 && r < rows - 1 && c > 0 && c < cols - 1)) continue;
                const float ax = fabsf(AT(gx, stride, r, c)), ay = fabsf(AT(gy, stride, r, c));
                const float m = ax > ay ? ax : ay;
                if (m > bound) bound = m;
            }
        bound = bound * 2.f;
        for (int t = 0; t < SIFT_D * SIFT_D * SIFT_N; t++) dst[t] = 0.f;
        if (!(bound > 0.f) || !isfinite(bound)) continue;  /* flat window: all-zero descriptor */
        int e;
        (void)frexpf(bound, &e);  /* bound < 2^e */
        int64_t hist[(SIFT_D + 2) * (SIFT_D + 2) * (SIFT_N + 2)];
        memset(hist, 0, sizeof(hist));
        (void)HS;
        for (int i = -g.radius; i <= g.radius; i++)
            for (int j = -g.radius; j <= g.radius; j++) {
                float rbin, cbin, dx, dy, w;
                if (!sift_sample(gx, gy, rows, cols, stride, &g, i, j, &rbin, &cbin, &dx, &dy, &w)) continue;
                const float mag = sqrtf(dx * dx + dy * dy) * exp_neg(w);
                float obin = (fast_atan2_deg(dy, dx) - g.ori) * ((float)SIFT_N / 360.f);
                const float r0f = floorf(rbin), c0f = floorf(cbin), o0f = floorf(obin);
                rbin -= r0f;
                cbin -= c0f;
                obin -= o0f;
                const int r0 = (int)r0f, c0 = (int)c0f;
                int o0 = (int)o0f;
                if (o0 < 0) o0 += SIFT_N;
                if (o0 >= SIFT_N) o0 -= SIFT_N;
                const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
                const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
                const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
                const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
                const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
                const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
                const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
                o0 = o0 < 0 ? 0 : (o0 > SIFT_N - 1 ? SIFT_N - 1 : o0); /* only a NaN sample gets here out of range */
                const int idx = ((r0 + 1) * (SIFT_D + 2) + c0 + 1) * (SIFT_N + 2) + o0;
#define FX(v) ((int64_t)llrintf(ldexpf((v), 40 - e)))
                hist[idx] += FX(v000);
                hist[idx + 1] += FX(v001);
                hist[idx + (SIFT_N + 2)] += FX(v010);
                hist[idx + (SIFT_N + 3)] += FX(v011);
                hist[idx + (SIFT_D + 2) * (SIFT_N + 2)] += FX(v100);
                hist[idx + (SIFT_D + 2) * (SIFT_N + 2) + 1] += FX(v101);
                hist[idx + (SIFT_D + 3) * (SIFT_N + 2)] += FX(v110);
                hist[idx + (SIFT_D + 3) * (SIFT_N + 2) + 1] += FX(v111);
#undef FX
            }
        /* finalize: the orientation histogram is circular; border spatial bins are dropped */
        for (int i = 0; i < SIFT_D; i++)
            for (int j = 0; j < SIFT_D; j++) {
                const int idx = ((i + 1) * (SIFT_D + 2) + (j + 1)) * (SIFT_N + 2);
                hist[idx] += hist[idx + SIFT_N];
                hist[idx + 1] += hist[idx + SIFT_N + 1];
                for (int o = 0; o < SIFT_N; o++)
                    dst[(i * SIFT_D + j) * SIFT_N + o] = ldexpf((float)hist[idx + o], e - 40);
            }
        const int len = SIFT_D * SIFT_D * SIFT_N;
        float nrm2 = 0.f;
        for (int t = 0; t < len; t++) nrm2 += dst[t] * dst[t];
        const float thr = sqrtf(nrm2) * 0.2f;  /* SIFT_DESCR_MAG_THR */
        nrm2 = 0.f;
        for (int t = 0; t < len; t++) {
            const float val = dst[t] < thr ? dst[t] : thr;
            dst[t] = val;
            nrm2 += val * val;
        }
        const float nrm = sqrtf(nrm2);
        const float scale = 512.f / (nrm > FLT_EPSILON ? nrm : FLT_EPSILON);  /* SIFT_INT_DESCR_FCTR */
        for (int t = 0; t < len; t++) {
            const float v = rintf(dst[t] * scale);  /* saturate_cast<uchar> */
            dst[t] = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
        }
    }
    return 0;
}
