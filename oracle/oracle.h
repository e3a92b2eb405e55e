/*
 * oracle.h -- CPU restatement (plain C) of the reference's per-pixel kernel path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped library (libmicv.so) never
 * links, loads or calls anything in this directory.
 *
 * PARITY STATUS: **parity unpinned**.  The reference (tanmaniac/IntroToComputerVision)
 * cannot be compiled in this image (every translation unit on the path includes OpenCV
 * 3.4.1 / CUDA headers, neither is installed; its yaml-cpp/spdlog submodules are empty),
 * and it ships no tests, golden vectors or known-answer fixtures.  Every function below
 * therefore restates (a) the reference's own source where the arithmetic is written
 * there, cited file:line, and (b) OpenCV 3.4.1's published algorithm where the reference
 * delegates to the library (Sobel, GaussianBlur, remap, resize, solve, determinant,
 * getGaussianKernel).  Floating-point accumulation ORDER inside those library calls is a
 * recorded decision of this repository (DESIGN.md "Arithmetic contract"), not something
 * that could be checked against the library here.  Independent cross-checks against
 * scipy.ndimage live in tests/test_oracle_crosscheck.py.
 *
 * All images are row-major float32 with an explicit row stride in ELEMENTS.
 * Reference paths are relative to /root/reference.
 */
#ifndef MICV_ORACLE_H
#define MICV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- shared building blocks (OpenCV 3.4.1 semantics, restated) ---- */

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101). */
int orc_reflect101(int p, int len);

/* cv::getGaussianKernel(n, sigma, CV_32F) for sigma > 0: taps exp(-(i-(n-1)/2)^2/(2 sigma^2))
 * evaluated in double, cast to float, summed in double, scaled by 1/sum, cast to float. */
void orc_gaussian_kernel(int n, double sigma, float *taps);

/* Separable correlation, row pass then column pass, float intermediate, every tap a
 * fused multiply-add into an accumulator that starts at +0 (taps visited left->right /
 * top->bottom), BORDER_REFLECT_101.  Models cv::cuda::createSeparableLinearFilter
 * (Pyramids.cu:49,126; OpticalFlow.cpp:20-21) and, by decision, cv::GaussianBlur. */
void orc_sep_filter(const float *src, int rows, int cols, size_t sstride,
                    const float *krow, int nrow, const float *kcol, int ncol,
                    float *dst, size_t dstride);

/* 3x3 Sobel pair as cv::cuda::createSobelFilter(CV_32F,-1,dx,dy,3,scale):
 * d/dx: row kernel [-1,0,1], column kernel [1,2,1]*scale;
 * d/dy: row kernel [1,2,1]*scale, column kernel [-1,0,1].
 * ksize in {1,3,5,7} follows cv::getDerivKernels (Harris.cpp:16,24-25). */
int orc_sobel(const float *src, int rows, int cols, size_t sstride, int ksize, float scale,
              float *gx, float *gy, size_t gstride);

/* ---- ps5: Lucas-Kanade + pyramids ---- */

/* lk::calcOpticalFlow, ps5_cpp/lib/OpticalFlow.cpp:41-104 (tau = 0.1). */
int orc_lk_flow(const float *prev, const float *next, int rows, int cols, size_t stride,
                int win, float *u, float *v, size_t ostride);

/* Alternative accumulation orders of the five window sums, used ONLY to bound how far a real OpenCV 3.4.1
 * run could be from the contract (tests/test_unpinned_bounds.py). */
#define ORC_VAR_BLUR_CVCPU 1 /* OpenCV's CPU FilterEngine order: row left->right, column folded from the centre */
#define ORC_VAR_BLUR_FUSED 2 /* ... with fused multiply-adds (an AVX2/FMA3 build) instead of mul + add */
void orc_sep_filter_cvcpu(const float *src, int rows, int cols, size_t sstride,
                          const float *krow, int nrow, const float *kcol, int ncol,
                          float *dst, size_t dstride, int fused);
int orc_lk_flow_ex(const float *prev, const float *next, int rows, int cols, size_t stride,
                   int win, int variant, float *u, float *v, size_t ostride, double *det_out);
/* the same on a crop at frame position (oy, ox): see oracle_lk.c */
int orc_lk_flow_pyr_at(const float *prev, const float *next, int rows, int cols, size_t stride,
                       int win, int levels, int oy, int ox, float *u, float *v, size_t ostride);
int orc_lk_flow_pyr_ex(const float *prev, const float *next, int rows, int cols, size_t stride,
                       int win, int levels, int variant, float *u, float *v, size_t ostride, double *det0_out);

/* cvRound(float) of the reference's x86-64 OpenCV 3.4.1 build: round half to even into 32 bits,
 * INT_MIN ("integer indefinite") for NaN and for values that do not fit an int32. */
int orc_cv_round(float v);

/* cv::remap(src,dst,mapx,mapy,INTER_LINEAR) with BORDER_CONSTANT(0): 1/32-pixel
 * fixed-point coordinates cvRound(v * 32) (orc_cv_round: a NaN / infinite / out-of-range map
 * entry gives the border constant), 4-tap float weights. */
void orc_remap_linear(const float *src, int rows, int cols, size_t sstride,
                      const float *mapx, const float *mapy, size_t mstride,
                      float *dst, int drows, int dcols, size_t dstride);

/* lk::warp, OpticalFlow.cpp:106-120. */
void orc_lk_warp(const float *src, const float *du, const float *dv, int rows, int cols,
                 size_t stride, float *dst);

/* cv::resize(src,dst,dsize) default INTER_LINEAR for CV_32F (OpticalFlow.cpp:149-150). */
void orc_resize_linear(const float *src, int srows, int scols, size_t sstride,
                       float *dst, int drows, int dcols, size_t dstride);

/* pyr::pyrDown AS WRITTEN (Pyramids.cu:21-32,53,65-66): dst(y,x) = src(2y+1,2x+1),
 * dst is (rows/2) x (cols/2); the 5-tap blur the reference computes is never read. */
void orc_pyr_down(const float *src, int rows, int cols, size_t sstride,
                  float *dst, size_t dstride);

/* pyr::pyrUp (Pyramids.cu:75-131): 2x nearest-neighbour replicate, then separable
 * [1,4,6,4,1]/16 blur with BORDER_REFLECT_101; dst is (2 rows) x (2 cols). */
void orc_pyr_up(const float *src, int rows, int cols, size_t sstride,
                float *dst, size_t dstride);

/* lk::calcOpticalFlowPyr, OpticalFlow.cpp:122-167, with the pyramid depth (hard-coded 4
 * at :127) as a parameter.  Inputs are single-channel float32 (makeGaussianPyramid's
 * colour/convert step is orc_rgb_to_gray_f32 below). */
int orc_lk_flow_pyr(const float *prev, const float *next, int rows, int cols, size_t stride,
                    int win, int levels, float *u, float *v, size_t ostride);

/* cv::cvtColor(COLOR_RGB2GRAY) on 8-bit 3-channel data followed by convertTo(CV_32F)
 * (Pyramids.cpp:10-15): fixed-point R*4899 + G*9617 + B*1868, +2^13, >>14. */
int orc_to_gray_f32(const void *src, int rows, int cols, size_t sstride_bytes, int channels, int depth,
                    float *dst, size_t dstride);
void orc_rgb8_to_gray_f32(const uint8_t *rgb, int rows, int cols, size_t sstride_bytes,
                          float *dst, size_t dstride);

/* ---- ps4: Harris + SIFT-style keypoint angles ---- */

/* harris::getCornerResponse.  Semantics of harris::cpu (Harris.cpp:43-97: clamped window,
 * Gaussian outer-product weights) with the accumulation written as harris::gpu does
 * (Harris.cu:36-43,76-91: fmaf chain in (wy,wx) raster order, all-float det/trace). */
int orc_harris_response(const float *gx, const float *gy, int rows, int cols, size_t stride,
                        int win, double sigma, float alpha, float *resp, size_t rstride);

/* The same with the arithmetic selected: ORC_HARRIS_GPU = the contract (harris::gpu, Harris.cu:36-43,85-91:
 * fma.rn accumulation, float det - alpha tr^2, unfused); ORC_HARRIS_CPU = harris::cpu as written
 * (Harris.cpp:78-92: unfused `secondMoment + weight * gradVals`, cv::determinant in double, the difference
 * rounded to float once); ORC_HARRIS_GPU_FMAD = the GPU kernel's last three lines as nvcc's default
 * contraction would compile them (a bounding variant, tests/test_unpinned_bounds.py). */
#define ORC_HARRIS_GPU 0
#define ORC_HARRIS_CPU 1
#define ORC_HARRIS_GPU_FMAD 2
int orc_harris_response_ex(const float *gx, const float *gy, int rows, int cols, size_t stride,
                           int win, double sigma, float alpha, int mode, float *resp, size_t rstride);

/* harris::{cpu,gpu}::refineCorners (Harris.cpp:99-147 / Harris.cu:173-329).  corners is
 * zero except at kept maxima; locs receives (y,x) pairs in row-major order.  Returns the
 * number of corners found (may exceed cap; only cap pairs are written). */
int64_t orc_harris_refine(const float *resp, int rows, int cols, size_t stride,
                          double threshold, int min_distance,
                          float *corners, size_t cstride, int32_t *locs_yx, int64_t cap);

/* sift::getAnglesFromGradients (Descriptors.cpp:7-25): atan2f(Iy, Ix). */
void orc_sift_angles(const float *gx, const float *gy, int rows, int cols, size_t stride,
                     float *angles, size_t astride);

/* sift::getKeypoints (Descriptors.cpp:27-47): angle_deg = atan2f(Iy,Ix)*180.f/3.1415921636f
 * per (y,x) corner; out is [n][4] = {x, y, size, angle}. */
void orc_sift_keypoints(const float *gx, const float *gy, int rows, int cols, size_t stride,
                        const int32_t *locs_yx, int64_t n, float size, float *kp_xysa);

/* The SIFT-style descriptor window at the keypoints above (call site ps4_cpp/src/Solution.cpp:166-169,
 * cv::xfeatures2d::SIFT::compute).  PARITY UNPINNED: OpenCV's SIFT is absent third-party code; this is
 * the published 4x4x8 algorithm sampled on the harris::getGradients fields with every transcendental
 * and the accumulation order fixed -- see oracle_sift.c.  desc is [n][128] (row pitch dstride floats),
 * values are the 8-bit quantised descriptor stored as float like cv::xfeatures2d::SIFT's CV_32F output. */
int orc_sift_descriptors(const float *gx, const float *gy, int rows, int cols, size_t stride,
                         const float *kp_xysa, int64_t n, float *desc, size_t dstride);

/* ---- ps2: window stereo ---- */

#define ORC_STEREO_COLS_2R     1  /* reproduce CUDA's `i < 2*windowRad` column count (DisparitySSD.cu:84) */
#define ORC_STEREO_MIN_SSD_5E6 2  /* keep -1 where best SSD >= 5e6 (DisparitySSD.cu:16,177-178) */
#define ORC_STEREO_ROLLING     8  /* column sums as the CUDA kernels keep them (DisparitySSD.cu:97-138,
                                     DisparityNCorr.cu:117-173): rows are cut into strips of
                                     ROWS_PER_THREAD = 40; the first row of a strip sums its 2r+1 terms
                                     top -> bottom from 0, every further row takes the previous row's
                                     column sum, subtracts the term that left the window and then adds
                                     the one that entered it (two roundings per row, float).  Equal to
                                     the fresh sums for integer-valued images, not for general f32. */

/* disparitySSD.  CUDA-path addressing (clamp-to-edge textures, every d in [minD,maxD]
 * evaluated, strict '<' so the lowest d wins ties; DisparitySSD.cu:54-140) with the window
 * corrected to (2r+1)^2 unless ORC_STEREO_COLS_2R.  Cost = column sums (top->bottom float
 * adds of float squares) then a left->right float sum of 2r+1 column sums. */
int orc_disparity_ssd(const float *left, const float *right, int rows, int cols, size_t stride,
                      int rad, int min_d, int max_d, int flags, int8_t *disp, size_t dstride);

/* serial::disparitySSD exactly as written (DisparitySSD.cpp:35-61), including the
 * clamped search range and per-term integer rounding; reads outside the padded image
 * (undefined in the reference) are clamped here. */
int orc_disparity_ssd_serial(const float *left, const float *right, int rows, int cols,
                             size_t stride, int rad, int min_d, int max_d,
                             int8_t *disp, size_t dstride);

/* disparityNCorr, CUDA-path semantics (DisparityNCorr.cu:60-174): score =
 * sum(ab)/sqrt(sum(a^2) sum(b^2)), running max initialised to 0, first max wins. */
int orc_disparity_ncorr(const float *left, const float *right, int rows, int cols, size_t stride,
                        int rad, int min_d, int max_d, int flags, int8_t *disp, size_t dstride);

/* ---- ps1: Hough ---- */

/* Dimensions of the line accumulator (Hough.cu:258-263). */
void orc_hough_lines_dims(int rows, int cols, unsigned rho_bin, unsigned theta_bin,
                          int *rho_bins, int *theta_bins);
/* cuda::houghLinesAccumulate (Hough.cu:35-59,251-290) with cos/sin taken from a
 * host-built float table (the reference's __sincosf is not reproducible). */
int orc_hough_lines(const uint8_t *mask, int rows, int cols, size_t stride,
                    unsigned rho_bin, unsigned theta_bin, int32_t *acc);
/* cuda::houghCirclesAccumulate (Hough.cu:70-95,311-346); acc is rows x cols, zeroed here. */
int orc_hough_circles(const uint8_t *mask, int rows, int cols, size_t stride,
                      unsigned radius, int32_t *acc);
/* cuda::findLocalMaxima (Hough.cu:137-162,366-415): up/left 2x2 "local max" rule,
 * votes >= threshold, stable sort by votes descending, first num_peaks. Returns count. */
int64_t orc_hough_peaks(const int32_t *acc, int rows, int cols, unsigned num_peaks,
                        int threshold, uint32_t *peaks_rc);
/* The float cos/sin table (degrees -90..269) shared by the oracle's Hough functions. */
void orc_hough_trig_table(float *cos360, float *sin360);

/* ---- ps1: edge front-end ---- */

/* cv::cuda Gaussian blur on CV_8U, Canny (aperture 3, L1 norm), sol::generateEdge
 * (ps1_cpp/src/Solution.cpp:21-47). */
void orc_gauss_u8(const uint8_t *src, int rows, int cols, size_t stride, int ksize, double sigma,
                  uint8_t *dst, size_t dstride);
int orc_canny(const uint8_t *src, int rows, int cols, size_t stride, double low_thresh,
              double high_thresh, uint8_t *edges, size_t estride);
int orc_generate_edge(const uint8_t *src, int rows, int cols, size_t stride, int gauss_size,
                      double gauss_sigma, double low_thresh, double high_thresh, uint8_t *edges,
                      size_t estride);

/* ---- ps4: descriptor matching ---- */

/* cv::BFMatcher (NORM_L2) knnMatch k = 2 and the ratio test, ps4_cpp/src/Solution.cpp:172-184. */
void orc_bf_knn2(const float *query, int nq, size_t qstride, const float *train, int nt, size_t tstride,
                 int dim, int32_t *idx2, float *dist2);
int64_t orc_bf_ratio_filter(const int32_t *idx2, const float *dist2, int nq, double ratio,
                            int32_t *matches_qt, float *distances, int64_t cap);

/* ---- ps7: motion history ---- */

/* thresholdDifference (MotionHistory.cu:17-48), mhi::frameDifference (MotionHistory.cpp:26-77,
 * single-channel CV_8U), mhi::calcMotionHistory (MotionHistory.cu:52-66), mhi::energyFromHistory
 * (MotionHistory.cpp:98-105). */
void orc_mhi_threshold(const uint8_t *src, size_t n, double thresh, uint8_t *dst);
int orc_mhi_frame_difference(const uint8_t *f1, const uint8_t *f2, int rows, int cols, size_t stride,
                             double thresh, int kw, int kh, double sigma, uint8_t *diff, size_t dstride);
void orc_mhi_update(uint8_t *history, size_t hstride, const uint8_t *mask, size_t mstride, int rows,
                    int cols, int tau);
void orc_mhi_energy(const uint8_t *mhi, size_t n, uint8_t *mei);
void orc_ellipse7(uint8_t m[7][7]);

#ifdef __cplusplus
}
#endif
#endif
