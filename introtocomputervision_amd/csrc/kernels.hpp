// kernels.hpp -- host-side launchers shared between translation units of libmicv.
#pragma once
#include "common.hpp"

namespace micv {

// Separable correlation passes over `nfields` planar fields (field f at base + f*field_elems).
// Row pass: dst(y,x) = chain_k fmaf(src(y, reflect101(x+k-n/2)), taps[k], acc), acc0 = +0.
int launch_filter_rows(hipStream_t s, const float *src, int sstride, size_t sfield, float *dst,
                       int dstride, size_t dfield, int rows, int cols, int nfields, const Taps &t);
// Column pass, same chain top->bottom.
int launch_filter_cols(hipStream_t s, const float *src, int sstride, size_t sfield, float *dst,
                       int dstride, size_t dfield, int rows, int cols, int nfields, const Taps &t);

// Sobel pair through the two generic passes. tmp: 2*rows*cols floats.
int sobel_dev(hipStream_t s, const float *src, int rows, int cols, int sstride, int ksize,
              float scale, float *gx, float *gy, int gstride, float *tmp, bool force_generic = false);

// dst(y,x) = src(2y+1, 2x+1), dst dims (rows/2, cols/2).
int launch_pyr_down(hipStream_t s, const float *src, int rows, int cols, int sstride, float *dst,
                    int dstride);
// 2x replicate + [1,4,6,4,1]/16 separable blur, result times `scale` (1 or 2, exact).
// tmp: rows * 2*cols floats.
int launch_pyr_up(hipStream_t s, const float *src, int rows, int cols, int sstride, float *dst,
                  int dstride, float scale, float *tmp);
// Builds pyramid levels of `batch` images in one launch (level l = direct decimation of the
// input). dst[l] == nullptr skips a level; image b of level l lands at dst[l] + b*rows_l*cols_l.
int launch_pyr_build(hipStream_t s, const float *src, size_t img_elems, int sstride, int rows,
                     int cols, int levels, float *const *dst, int batch);
// Two image sets (prev / next) in one launch; dst_a[l] == nullptr skips level l in both.
int launch_pyr_build2(hipStream_t s, const float *src_a, const float *src_b, size_t img_elems,
                      int sstride, int rows, int cols, int levels, float *const *dst_a,
                      float *const *dst_b, int batch, const int *row_lo = nullptr,
                      const int *row_hi = nullptr);
int launch_resize_linear(hipStream_t s, const float *src, int srows, int scols, int sstride,
                         float *dst, int drows, int dcols, int dstride);
// du = resize(2 * pyrUp(du_coarse), drows x dcols) for `batch` pairs and both fields, one launch.
int launch_flow_expand_resize(hipStream_t s, const float *src_u, const float *src_v, int fr, int fc,
                              size_t src_pair, float *dst_u, float *dst_v, int drows, int dcols,
                              size_t dst_pair, int batch);
int launch_warp(hipStream_t s, const float *src, int sstride, const float *du, const float *dv,
                int fstride, int rows, int cols, float *dst, int dstride, int batch = 1, size_t src_img = 0,
                size_t flow_img = 0, size_t dst_img = 0);
int launch_pyr_up_batch(hipStream_t s, const float *src_u, const float *src_v, int rows, int cols, size_t src_img,
                        float *dst_u, float *dst_v, size_t dst_img, float scale, int batch);

}  // namespace micv
