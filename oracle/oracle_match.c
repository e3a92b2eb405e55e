/*
 * oracle_match.c -- CPU restatement of the descriptor matching step of ps4:
 * cv::BFMatcher::create() (NORM_L2, no cross-check) -> knnMatch(k = 2) -> Lowe ratio test
 * (ps4_cpp/src/Solution.cpp:172-184).  TEST INFRASTRUCTURE ONLY; parity unpinned (oracle.h): the
 * descriptors themselves come from cv::xfeatures2d::SIFT (third party), so tests use synthetic ones.
 */
#include "oracle.h"

#include <math.h>

/* L2 distance, decision of this repository: d2 = fmaf chain of (a-b)^2 over the dimensions in
 * order, distance = sqrtf(d2).  Best two per query by (distance, train index) ascending -- what a
 * strict `<` scan of the train set in index order keeps. */
void orc_bf_knn2(const float *query, int nq, size_t qstride, const float *train, int nt, size_t tstride,
                 int dim, int32_t *idx2, float *dist2) {
    for (int q = 0; q < nq; q++) {
        float d0 = INFINITY, d1 = INFINITY;
        int i0 = -1, i1 = -1;
        const float *a = query + (size_t)q * qstride;
        for (int t = 0; t < nt; t++) {
            const float *b = train + (size_t)t * tstride;
            float acc = 0.f;
            for (int k = 0; k < dim; k++) {
                float diff = a[k] - b[k];
                acc = fmaf(diff, diff, acc);
            }
            if (acc < d0) { d1 = d0; i1 = i0; d0 = acc; i0 = t; }
            else if (acc < d1) { d1 = acc; i1 = t; }
        }
        idx2[2 * q] = i0; idx2[2 * q + 1] = i1;
        dist2[2 * q] = sqrtf(d0); dist2[2 * q + 1] = sqrtf(d1);
    }
}

/* Solution.cpp:180-184: keep matchPair[0] when distance0 < 0.75 * distance1 (float vs double). */
int64_t orc_bf_ratio_filter(const int32_t *idx2, const float *dist2, int nq, double ratio,
                            int32_t *matches_qt, float *distances, int64_t cap) {
    int64_t n = 0;
    for (int q = 0; q < nq; q++)
        if ((double)dist2[2 * q] < ratio * (double)dist2[2 * q + 1]) {
            if (n < cap) { matches_qt[2 * n] = q; matches_qt[2 * n + 1] = idx2[2 * q]; distances[n] = dist2[2 * q]; }
            n++;
        }
    return n;
}
