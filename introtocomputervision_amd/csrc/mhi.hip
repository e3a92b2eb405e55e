// mhi.hip -- ps7 motion-history path: frame differencing and the MHI
// update.  Byte kernels, HBM-bound by nature; the blur reuses the separable fmaf-chain contract.
//
//   mhi::frameDifference (MotionHistory.cpp:26-77), single-channel CV_8U frames:
//     Gaussian blur of both frames (cv::cuda separable filter: u8 -> float row pass -> column pass
//     -> saturate_cast<uchar>), saturating subtract f2 - f1, AbsThreshold -> {0,1}
//     (MotionHistory.cu:17-48), morphological OPEN with the 7x7 ellipse (erode, dilate; each pass
//     pads with BORDER_REFLECT_101 like cv::cuda's copyMakeBorder).
//   mhi::calcMotionHistory -> motionHistoryKernel (MotionHistory.cu:52-66).
#include <cmath>

#include "kernels.hpp"

namespace micv {

__device__ __forceinline__ uint8_t sat_u8_rn(float v) {
    const int r = __float2int_rn(v);  // round-half-even like saturate_cast<uchar>(float)
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

// Row pass of both frames at once: u8 -> float (blockIdx.z selects the frame).
__global__ __launch_bounds__(256) void mhi_blur_rows_kernel(const uint8_t *__restrict__ f1,
                                                             const uint8_t *__restrict__ f2,
                                                             size_t stride, int rows, int cols,
                                                             float *__restrict__ buf, Taps t) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const uint8_t *s = (blockIdx.z ? f2 : f1) + (size_t)y * stride;
    const int a = t.n / 2;
    float acc = 0.f;
    for (int k = 0; k < t.n; k++) acc = fmaf((float)s[reflect101(x + k - a, cols)], t.k[k], acc);
    buf[blockIdx.z * (size_t)rows * cols + (size_t)y * cols + x] = acc;
}

// Column pass of both frames, saturating subtract f2 - f1 (cv::cuda::subtract on CV_8U) and
// AbsThreshold, fused: writes the {0,1} mask.
__global__ __launch_bounds__(256) void mhi_blur_cols_diff_kernel(const float *__restrict__ buf,
                                                                  int rows, int cols, Taps t,
                                                                  double thresh,
                                                                  uint8_t *__restrict__ mask) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const size_t n = (size_t)rows * cols;
    const int a = t.n / 2;
    float a1 = 0.f, a2 = 0.f;
    for (int k = 0; k < t.n; k++) {
        const size_t o = (size_t)reflect101(y + k - a, rows) * cols + x;
        a1 = fmaf(buf[o], t.k[k], a1);
        a2 = fmaf(buf[n + o], t.k[k], a2);
    }
    const int d = (int)sat_u8_rn(a2) - (int)sat_u8_rn(a1);
    const int val = d < 0 ? 0 : d;
    mask[(size_t)y * cols + x] = ((double)val >= thresh || (double)(-val) >= thresh) ? 1 : 0;
}

// mhi::energyFromHistory (MotionHistory.cpp:98-105): any nonzero history value -> 1.
__global__ __launch_bounds__(256) void mhi_energy_kernel(const uint8_t *__restrict__ mhi, size_t sstride,
                                                          int rows, int cols, uint8_t *__restrict__ mei,
                                                          size_t dstride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    mei[(size_t)y * dstride + x] = mhi[(size_t)y * sstride + x] > 0 ? 1 : 0;
}

struct Ellipse7 {
    unsigned char m[7];  // bit j of m[i] = element (i, j)
};

// Erode / dilate with the 7x7 ellipse from an LDS tile: 64x16 outputs + 3-pixel halo (BORDER_REFLECT_101 resolved
// while loading), 37 taps per pixel served from LDS instead of global bytes.
__global__ __launch_bounds__(256) void mhi_morph7_tiled_kernel(const uint8_t *__restrict__ src, int rows,
                                                                int cols, int dilate, Ellipse7 e,
                                                                uint8_t *__restrict__ dst,
                                                                size_t dstride) {
    constexpr int TW = 64, TH = 16, RW = TW + 6, RH = TH + 6, PS = RW + 1;
    __shared__ int T[RH * PS];  // one value per word: byte-wide LDS reads would serialise on banks
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    {
        constexpr int NB = (RH * RW + 255) / 256;
        int v[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = threadIdx.x + k * 256 < RH * RW ? threadIdx.x + k * 256 : RH * RW - 1;
            const int ly = i / RW, lx = i - ly * RW;
            v[k] = src[(size_t)reflect101(y0 - 3 + ly, rows) * cols + reflect101(x0 - 3 + lx, cols)];
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = threadIdx.x + k * 256;
            if (i < RH * RW) T[(i / RW) * PS + (i % RW)] = v[k];
        }
    }
    __syncthreads();
    const int c = threadIdx.x & 63, x = x0 + c;
    if (x >= cols) return;
#pragma unroll
    for (int q = 0; q < TH / 4; q++) {
        const int ry = (threadIdx.x >> 6) * (TH / 4) + q, y = y0 + ry;
        if (y >= rows) break;
        int v = dilate ? 0 : 255;
#pragma unroll
        for (int i = 0; i < 7; i++) {
#pragma unroll
            for (int j = 0; j < 7; j++) {
                if (!((e.m[i] >> j) & 1)) continue;
                const int t = T[(ry + i) * PS + c + j];
                v = dilate ? (t > v ? t : v) : (t < v ? t : v);
            }
        }
        dst[(size_t)y * dstride + x] = (uint8_t)v;
    }
}

__global__ __launch_bounds__(256) void mhi_threshold_kernel(const uint8_t *__restrict__ src,
                                                             size_t sstride, int rows, int cols,
                                                             double thresh,
                                                             uint8_t *__restrict__ dst,
                                                             size_t dstride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const int val = src[(size_t)y * sstride + x];
    dst[(size_t)y * dstride + x] = ((double)val >= thresh || (double)(-val) >= thresh) ? 1 : 0;
}

__global__ __launch_bounds__(256) void mhi_update_kernel(uint8_t *__restrict__ hist, size_t hstride,
                                                          const uint8_t *__restrict__ mask,
                                                          size_t mstride, int rows, int cols,
                                                          int tau) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const int h = hist[(size_t)y * hstride + x];
    hist[(size_t)y * hstride + x] =
        (uint8_t)(mask[(size_t)y * mstride + x] == 1 ? tau : (h - 1 > 0 ? h - 1 : 0));  // MotionHistory.cu:63-65
}

// cv::getStructuringElement(MORPH_ELLIPSE, Size(7,7)).
static Ellipse7 ellipse7() {
    Ellipse7 e;
    const int r = 3, c = 3;
    const double inv_r2 = 1.0 / ((double)r * r);
    for (int i = 0; i < 7; i++) {
        const int dy = i - r;
        const int dx = (int)std::lrint(c * std::sqrt((r * r - dy * dy) * inv_r2));
        const int j1 = c - dx < 0 ? 0 : c - dx, j2 = c + dx + 1 > 7 ? 7 : c + dx + 1;
        e.m[i] = 0;
        for (int j = j1; j < j2; j++) e.m[i] |= (unsigned char)(1u << j);
    }
    return e;
}

}  // namespace micv

using namespace micv;

extern "C" {

int micv_mhi_frame_difference_dev(micv_ctx *ctx, const uint8_t *f1, const uint8_t *f2, int rows,
                                  int cols, size_t stride, double thresh, int blur_w, int blur_h,
                                  double blur_sigma, uint8_t *diff, size_t dstride,
                                  micv_stream stream) {
    MICV_REQUIRE(ctx && f1 && f2 && diff, "micv_mhi_frame_difference: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && stride >= (size_t)cols && dstride >= (size_t)cols,
                 "micv_mhi_frame_difference: bad size / stride");
    MICV_REQUIRE(blur_w >= 1 && blur_w <= 31 && (blur_w & 1) && blur_h >= 1 && blur_h <= 31 && (blur_h & 1) &&
                     blur_sigma > 0,
                 "micv_mhi_frame_difference: blur %dx%d / sigma %g not supported (odd sizes <= 31, sigma > 0)",
                 blur_w, blur_h, blur_sigma);
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)rows * cols;
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need(2 * n, 4) + 2 * Carver::need(n, 1), &scratch));
    Carver c(scratch);
    float *buf = c.take<float>(2 * n);
    uint8_t *m0 = c.take<uint8_t>(n), *m1 = c.take<uint8_t>(n);
    Taps t, ty;  // cv::Size(width, height): width taps along x, height taps along y
    gaussian_taps(blur_w, blur_sigma, &t);
    gaussian_taps(blur_h, blur_sigma, &ty);
    const dim3 grid(cdiv(cols, 64), cdiv(rows, 4));
    mhi_blur_rows_kernel<<<dim3(grid.x, grid.y, 2), 256, 0, s>>>(f1, f2, stride, rows, cols, buf, t);
    MICV_LAUNCH_CHECK();
    mhi_blur_cols_diff_kernel<<<grid, 256, 0, s>>>(buf, rows, cols, ty, thresh, m0);
    MICV_LAUNCH_CHECK();
    const Ellipse7 e = ellipse7();
    const dim3 mgrid(cdiv(cols, 64), cdiv(rows, 16));
    mhi_morph7_tiled_kernel<<<mgrid, 256, 0, s>>>(m0, rows, cols, 0, e, m1, (size_t)cols);  // erode
    MICV_LAUNCH_CHECK();
    mhi_morph7_tiled_kernel<<<mgrid, 256, 0, s>>>(m1, rows, cols, 1, e, diff, dstride);     // dilate
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_mhi_energy_dev(micv_ctx *ctx, const uint8_t *mhi, int rows, int cols, size_t sstride,
                        uint8_t *mei, size_t dstride, micv_stream stream) {
    MICV_REQUIRE(ctx && mhi && mei, "micv_mhi_energy: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && sstride >= (size_t)cols && dstride >= (size_t)cols,
                 "micv_mhi_energy: bad size / stride");
    MICV_HIP(hipSetDevice(ctx->device));
    mhi_energy_kernel<<<dim3(cdiv(cols, 64), cdiv(rows, 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        mhi, sstride, rows, cols, mei, dstride);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_mhi_threshold_dev(micv_ctx *ctx, const uint8_t *src, int rows, int cols, size_t sstride,
                           double thresh, uint8_t *dst, size_t dstride, micv_stream stream) {
    MICV_REQUIRE(ctx && src && dst, "micv_mhi_threshold: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && sstride >= (size_t)cols && dstride >= (size_t)cols,
                 "micv_mhi_threshold: bad size / stride");
    MICV_HIP(hipSetDevice(ctx->device));
    mhi_threshold_kernel<<<dim3(cdiv(cols, 64), cdiv(rows, 4)), 256, 0,
                           static_cast<hipStream_t>(stream)>>>(src, sstride, rows, cols, thresh, dst,
                                                               dstride);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_mhi_update_dev(micv_ctx *ctx, uint8_t *history, size_t hstride, const uint8_t *mask,
                        size_t mstride, int rows, int cols, int tau, micv_stream stream) {
    MICV_REQUIRE(ctx && history && mask, "micv_mhi_update: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && hstride >= (size_t)cols && mstride >= (size_t)cols,
                 "micv_mhi_update: bad size / stride");
    MICV_REQUIRE(tau > 0, "micv_mhi_update: tau must be > 0");  // MotionHistory.cpp:80
    MICV_HIP(hipSetDevice(ctx->device));
    mhi_update_kernel<<<dim3(cdiv(cols, 64), cdiv(rows, 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        history, hstride, mask, mstride, rows, cols, tau);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

}  // extern "C"
