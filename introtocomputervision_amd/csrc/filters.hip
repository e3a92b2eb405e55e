// filters.hip -- generic separable filters (any tap count <= 64) and the Sobel pair.
//
// These are the "any window" building blocks: one thread per output pixel, taps in the
// kernarg block, inputs read straight from global memory (L2 serves the re-reads).  The
// metric path (win = 15 pyramidal LK) does not use them; it runs the LDS-tiled fused
// kernel in lk_fused.hip.  Both produce identical bits: each pass is the same fmaf chain.
#include "kernels.hpp"

namespace micv {

__global__ __launch_bounds__(256) void filter_rows_kernel(const float *__restrict__ src,
                                                           int sstride, size_t sfield,
                                                           float *__restrict__ dst, int dstride,
                                                           size_t dfield, int rows, int cols,
                                                           Taps t) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float *s = src + blockIdx.z * sfield + (size_t)y * sstride;
    const int a = t.n / 2;
    float acc = 0.f;
    if (x - a >= 0 && x - a + t.n <= cols) {
        for (int k = 0; k < t.n; k++) acc = fmaf(s[x + k - a], t.k[k], acc);
    } else {
        for (int k = 0; k < t.n; k++) acc = fmaf(s[reflect101(x + k - a, cols)], t.k[k], acc);
    }
    dst[blockIdx.z * dfield + (size_t)y * dstride + x] = acc;
}

__global__ __launch_bounds__(256) void filter_cols_kernel(const float *__restrict__ src,
                                                           int sstride, size_t sfield,
                                                           float *__restrict__ dst, int dstride,
                                                           size_t dfield, int rows, int cols,
                                                           Taps t) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float *s = src + blockIdx.z * sfield + x;
    const int a = t.n / 2;
    float acc = 0.f;
    if (y - a >= 0 && y - a + t.n <= rows) {
        for (int k = 0; k < t.n; k++) acc = fmaf(s[(size_t)(y + k - a) * sstride], t.k[k], acc);
    } else {
        for (int k = 0; k < t.n; k++)
            acc = fmaf(s[(size_t)reflect101(y + k - a, rows) * sstride], t.k[k], acc);
    }
    dst[blockIdx.z * dfield + (size_t)y * dstride + x] = acc;
}

int launch_filter_rows(hipStream_t s, const float *src, int sstride, size_t sfield, float *dst,
                       int dstride, size_t dfield, int rows, int cols, int nfields,
                       const Taps &t) {
    dim3 grid(cdiv(cols, 64), cdiv(rows, 4), nfields);
    filter_rows_kernel<<<grid, 256, 0, s>>>(src, sstride, sfield, dst, dstride, dfield, rows, cols,
                                            t);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int launch_filter_cols(hipStream_t s, const float *src, int sstride, size_t sfield, float *dst,
                       int dstride, size_t dfield, int rows, int cols, int nfields,
                       const Taps &t) {
    dim3 grid(cdiv(cols, 64), cdiv(rows, 4), nfields);
    filter_cols_kernel<<<grid, 256, 0, s>>>(src, sstride, sfield, dst, dstride, dfield, rows, cols,
                                            t);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// cv::cuda::createSobelFilter: `if (dx == 0) kx *= scale; else ky *= scale;` then a
// separable filter, row kernel kx, column kernel ky.
int sobel_dev(hipStream_t s, const float *src, int rows, int cols, int sstride, int ksize,
              float scale, float *gx, float *gy, int gstride, float *tmp) {
    Taps kx, ky;
    const size_t n = (size_t)rows * cols;
    // d/dx
    if (sobel_taps(ksize, 1, &kx) < 0 || sobel_taps(ksize, 0, &ky) < 0) {
        set_error("sobel: kernel size %d not supported (1,3,5,7..31 odd)", ksize);
        return MICV_EINVAL;
    }
    if (scale != 1.f)
        for (int i = 0; i < ky.n; i++) ky.k[i] *= scale;
    MICV_TRY(launch_filter_rows(s, src, sstride, 0, tmp, cols, 0, rows, cols, 1, kx));
    MICV_TRY(launch_filter_cols(s, tmp, cols, 0, gx, gstride, 0, rows, cols, 1, ky));
    // d/dy
    sobel_taps(ksize, 0, &kx);
    sobel_taps(ksize, 1, &ky);
    if (scale != 1.f)
        for (int i = 0; i < kx.n; i++) kx.k[i] *= scale;
    MICV_TRY(launch_filter_rows(s, src, sstride, 0, tmp + n, cols, 0, rows, cols, 1, kx));
    MICV_TRY(launch_filter_cols(s, tmp + n, cols, 0, gy, gstride, 0, rows, cols, 1, ky));
    return MICV_OK;
}

}  // namespace micv

using namespace micv;

extern "C" int micv_sobel_dev(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                              int ksize, float scale, float *gx, float *gy, size_t gstride,
                              micv_stream stream) {
    MICV_REQUIRE(ctx && src && gx && gy, "micv_sobel: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0, "micv_sobel: bad size %dx%d", rows, cols);
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(gstride, cols, 4),
                 "micv_sobel: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need((size_t)rows * cols * 2, 4), &scratch));
    return sobel_dev(static_cast<hipStream_t>(stream), src, rows, cols, (int)(sstride / 4), ksize,
                     scale, gx, gy, (int)(gstride / 4), static_cast<float *>(scratch));
}
