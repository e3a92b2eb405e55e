// host_api.hip -- `_host` flavours: host pointers in, host pointers out, synchronous.
// This is the behaviour of the reference's cv::Mat functions (upload -> kernel -> sync ->
// download inside every call, e.g. Harris.cu:118-158, Pyramids.cu:45-72); it is PCIe-bound
// by construction.  Device buffers are allocated per call like the reference's GpuMats.
#include <vector>

#include "common.hpp"

namespace micv {

// RAII device allocation; `ok()` reports failure through set_error.
struct DevBuf {
    void *p = nullptr;
    explicit DevBuf(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) p = nullptr;
    }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

#define MICV_ALLOC_OK(buf)                                        \
    do {                                                          \
        if (!(buf).p) {                                           \
            ::micv::set_error("device allocation failed");        \
            return MICV_ENOMEM;                                   \
        }                                                         \
    } while (0)

static int up2d(void *dst, const void *src, size_t sstride, size_t row_bytes, int rows,
                hipStream_t s) {
    MICV_HIP(hipMemcpy2DAsync(dst, row_bytes, src, sstride, row_bytes, rows, hipMemcpyHostToDevice,
                              s));
    return MICV_OK;
}
static int down2d(void *dst, size_t dstride, const void *src, size_t row_bytes, int rows,
                  hipStream_t s) {
    MICV_HIP(hipMemcpy2DAsync(dst, dstride, src, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost,
                              s));
    return MICV_OK;
}

}  // namespace micv

using namespace micv;

#define HOST_PROLOGUE(fn)                                    \
    MICV_REQUIRE(ctx != nullptr, fn ": ctx is null");        \
    MICV_HIP(hipSetDevice(ctx->device));                     \
    hipStream_t s = nullptr

extern "C" {

int micv_lk_flow_pyr_host(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                          size_t stride, int win, int levels, float *u, float *v, size_t ostride) {
    HOST_PROLOGUE("micv_lk_flow_pyr_host");
    MICV_REQUIRE(prev && next && u && v && rows > 0 && cols > 0, "micv_lk_flow_pyr_host: bad argument");
    MICV_REQUIRE(stride_ok(stride, cols, 4) && stride_ok(ostride, cols, 4),
                 "micv_lk_flow_pyr_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf dp(n), dn(n), du(n), dv(n);
    MICV_ALLOC_OK(dp); MICV_ALLOC_OK(dn); MICV_ALLOC_OK(du); MICV_ALLOC_OK(dv);
    MICV_TRY(up2d(dp.p, prev, stride, rb, rows, s));
    MICV_TRY(up2d(dn.p, next, stride, rb, rows, s));
    MICV_TRY(micv_lk_flow_pyr_dev(ctx, dp.as<float>(), dn.as<float>(), rows, cols, rb, win, levels,
                                  du.as<float>(), dv.as<float>(), rb, s));
    MICV_TRY(down2d(u, ostride, du.p, rb, rows, s));
    MICV_TRY(down2d(v, ostride, dv.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_lk_flow_host(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                      size_t stride, int win, float *u, float *v, size_t ostride) {
    HOST_PROLOGUE("micv_lk_flow_host");
    MICV_REQUIRE(prev && next && u && v && rows > 0 && cols > 0, "micv_lk_flow_host: bad argument");
    MICV_REQUIRE(stride_ok(stride, cols, 4) && stride_ok(ostride, cols, 4),
                 "micv_lk_flow_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf dp(n), dn(n), du(n), dv(n);
    MICV_ALLOC_OK(dp); MICV_ALLOC_OK(dn); MICV_ALLOC_OK(du); MICV_ALLOC_OK(dv);
    MICV_TRY(up2d(dp.p, prev, stride, rb, rows, s));
    MICV_TRY(up2d(dn.p, next, stride, rb, rows, s));
    MICV_TRY(micv_lk_flow_dev(ctx, dp.as<float>(), dn.as<float>(), rows, cols, rb, win,
                              du.as<float>(), dv.as<float>(), rb, s));
    MICV_TRY(down2d(u, ostride, du.p, rb, rows, s));
    MICV_TRY(down2d(v, ostride, dv.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_lk_warp_host(micv_ctx *ctx, const float *src, size_t sstride, const float *du,
                      const float *dv, size_t fstride, int rows, int cols, float *dst,
                      size_t dstride) {
    HOST_PROLOGUE("micv_lk_warp_host");
    MICV_REQUIRE(src && du && dv && dst && rows > 0 && cols > 0, "micv_lk_warp_host: bad argument");
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(fstride, cols, 4) &&
                     stride_ok(dstride, cols, 4),
                 "micv_lk_warp_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf ds(n), dU(n), dV(n), dd(n);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dU); MICV_ALLOC_OK(dV); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, src, sstride, rb, rows, s));
    MICV_TRY(up2d(dU.p, du, fstride, rb, rows, s));
    MICV_TRY(up2d(dV.p, dv, fstride, rb, rows, s));
    MICV_TRY(micv_lk_warp_dev(ctx, ds.as<float>(), rb, dU.as<float>(), dV.as<float>(), rb, rows,
                              cols, dd.as<float>(), rb, s));
    MICV_TRY(down2d(dst, dstride, dd.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_pyr_down_host(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                       float *dst, size_t dstride) {
    HOST_PROLOGUE("micv_pyr_down_host");
    MICV_REQUIRE(src && dst && rows > 0 && cols > 0, "micv_pyr_down_host: bad argument");
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(dstride, cols / 2, 4),
                 "micv_pyr_down_host: bad stride");
    const int dr = rows / 2, dc = cols / 2;
    DevBuf ds((size_t)rows * cols * 4), dd((size_t)dr * dc * 4);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, src, sstride, (size_t)cols * 4, rows, s));
    MICV_TRY(micv_pyr_down_dev(ctx, ds.as<float>(), rows, cols, (size_t)cols * 4, dd.as<float>(),
                               (size_t)dc * 4, s));
    if (dr > 0 && dc > 0) MICV_TRY(down2d(dst, dstride, dd.p, (size_t)dc * 4, dr, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_pyr_up_host(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                     float *dst, size_t dstride) {
    HOST_PROLOGUE("micv_pyr_up_host");
    MICV_REQUIRE(src && dst && rows > 0 && cols > 0, "micv_pyr_up_host: bad argument");
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(dstride, 2 * cols, 4),
                 "micv_pyr_up_host: bad stride");
    DevBuf ds((size_t)rows * cols * 4), dd((size_t)rows * cols * 16);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, src, sstride, (size_t)cols * 4, rows, s));
    MICV_TRY(micv_pyr_up_dev(ctx, ds.as<float>(), rows, cols, (size_t)cols * 4, dd.as<float>(),
                             (size_t)cols * 8, s));
    MICV_TRY(down2d(dst, dstride, dd.p, (size_t)cols * 8, 2 * rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_gaussian_pyramid_host(micv_ctx *ctx, const float *src, int rows, int cols,
                               size_t sstride, int levels, float *const *dst_levels) {
    HOST_PROLOGUE("micv_gaussian_pyramid_host");
    MICV_REQUIRE(src && dst_levels && rows > 0 && cols > 0, "micv_gaussian_pyramid_host: bad argument");
    MICV_REQUIRE(levels >= 1 && levels <= 16 && (rows >> (levels - 1)) > 0 &&
                     (cols >> (levels - 1)) > 0,
                 "micv_gaussian_pyramid_host: %d levels do not fit a %dx%d image", levels, rows,
                 cols);
    MICV_REQUIRE(stride_ok(sstride, cols, 4), "micv_gaussian_pyramid_host: bad stride");
    size_t total = 0, off[16];
    for (int l = 0; l < levels; l++) {
        off[l] = total;
        total += (((size_t)(rows >> l) * (cols >> l)) + 63) & ~size_t(63);
    }
    DevBuf ds((size_t)rows * cols * 4), dd(total * 4);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, src, sstride, (size_t)cols * 4, rows, s));
    float *lv[16];
    for (int l = 0; l < levels; l++) lv[l] = dd.as<float>() + off[l];
    MICV_TRY(micv_gaussian_pyramid_dev(ctx, ds.as<float>(), rows, cols, (size_t)cols * 4, levels, lv, s));
    for (int l = 0; l < levels; l++) {
        MICV_REQUIRE(dst_levels[l] != nullptr, "micv_gaussian_pyramid_host: dst_levels[%d] is null", l);
        MICV_HIP(hipMemcpyAsync(dst_levels[l], lv[l], (size_t)(rows >> l) * (cols >> l) * 4,
                                hipMemcpyDeviceToHost, s));
    }
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_sobel_host(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                    int ksize, float scale, float *gx, float *gy, size_t gstride) {
    HOST_PROLOGUE("micv_sobel_host");
    MICV_REQUIRE(src && gx && gy && rows > 0 && cols > 0, "micv_sobel_host: bad argument");
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(gstride, cols, 4),
                 "micv_sobel_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf ds(n), dx(n), dy(n);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dx); MICV_ALLOC_OK(dy);
    MICV_TRY(up2d(ds.p, src, sstride, rb, rows, s));
    MICV_TRY(micv_sobel_dev(ctx, ds.as<float>(), rows, cols, rb, ksize, scale, dx.as<float>(),
                            dy.as<float>(), rb, s));
    MICV_TRY(down2d(gx, gstride, dx.p, rb, rows, s));
    MICV_TRY(down2d(gy, gstride, dy.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

}  // extern "C"
