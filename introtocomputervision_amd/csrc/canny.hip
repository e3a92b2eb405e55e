// canny.hip -- the edge front-end of ps1: sol::generateEdge,
// ps1_cpp/src/Solution.cpp:21-47 = cv::cuda Gaussian blur on CV_8U + Canny (aperture 3, L1 norm).
// Integer / byte work throughout; the only data-dependent part is the hysteresis, which runs
// tile-local flood fills in LDS and repeats them until no tile changes (the host reads one flag per
// round, as OpenCV's own CUDA Canny reads its queue counter).
#include "kernels.hpp"

namespace micv {

__global__ __launch_bounds__(256) void gauss_u8_rows_kernel(const uint8_t *__restrict__ src,
                                                             size_t stride, int rows, int cols,
                                                             float *__restrict__ buf, Taps t) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const uint8_t *s = src + (size_t)y * stride;
    const int a = t.n / 2;
    float acc = 0.f;
    for (int k = 0; k < t.n; k++) acc = fmaf((float)s[reflect101(x + k - a, cols)], t.k[k], acc);
    buf[(size_t)y * cols + x] = acc;
}

__global__ __launch_bounds__(256) void gauss_u8_cols_kernel(const float *__restrict__ buf, int rows,
                                                             int cols, Taps t,
                                                             uint8_t *__restrict__ dst,
                                                             size_t dstride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const int a = t.n / 2;
    float acc = 0.f;
    for (int k = 0; k < t.n; k++)
        acc = fmaf(buf[(size_t)reflect101(y + k - a, rows) * cols + x], t.k[k], acc);
    const int r = __float2int_rn(acc);
    dst[(size_t)y * dstride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

// 3x3 Sobel (replicated border) -> packed (dx, dy) int16 pair and L1 magnitude.
__global__ __launch_bounds__(256) void canny_grad_kernel(const uint8_t *__restrict__ src,
                                                          size_t stride, int rows, int cols,
                                                          int *__restrict__ mag,
                                                          int *__restrict__ dxy) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const uint8_t *r0 = src + (size_t)clampi(y - 1, 0, rows - 1) * stride;
    const uint8_t *r1 = src + (size_t)y * stride;
    const uint8_t *r2 = src + (size_t)clampi(y + 1, 0, rows - 1) * stride;
    const int xl = clampi(x - 1, 0, cols - 1), xr = clampi(x + 1, 0, cols - 1);
    const int gx = (r0[xr] + 2 * r1[xr] + r2[xr]) - (r0[xl] + 2 * r1[xl] + r2[xl]);
    const int gy = (r2[xl] + 2 * r2[x] + r2[xr]) - (r0[xl] + 2 * r0[x] + r0[xr]);
    const size_t i = (size_t)y * cols + x;
    mag[i] = abs(gx) + abs(gy);
    dxy[i] = (gx & 0xFFFF) | (gy << 16);
}

// Non-maximum suppression along the quantised gradient direction + double threshold.
__global__ __launch_bounds__(256) void canny_map_kernel(const int *__restrict__ mag,
                                                         const int *__restrict__ dxy, int rows,
                                                         int cols, int low, int high,
                                                         uint8_t *__restrict__ map) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    auto M = [&](int yy, int xx) -> int {
        return ((unsigned)yy >= (unsigned)rows || (unsigned)xx >= (unsigned)cols) ? 0 : mag[(size_t)yy * cols + xx];
    };
    const size_t i = (size_t)y * cols + x;
    const int m = mag[i];
    uint8_t out = 0;
    if (m > low) {
        const int p = dxy[i];
        const int xs = (short)(p & 0xFFFF), ys = p >> 16;
        const int ax = abs(xs), ay = abs(ys) << 15;
        const int TG22 = 13573;  // (int)(0.41421356237 * (1 << 15) + 0.5)
        const int tg22x = ax * TG22;
        bool is_max;
        if (ay < tg22x) {
            is_max = m > M(y, x - 1) && m >= M(y, x + 1);
        } else {
            const int tg67x = tg22x + (ax << 16);
            if (ay > tg67x) {
                is_max = m > M(y - 1, x) && m >= M(y + 1, x);
            } else {
                const int s = (xs ^ ys) < 0 ? -1 : 1;
                is_max = m > M(y - 1, x - s) && m > M(y + 1, x + s);
            }
        }
        if (is_max) out = m > high ? 2 : 1;
    }
    map[i] = out;
}

// One hysteresis round: every 32x32 tile (+1 halo) floods strong pixels (2) into 8-connected
// candidates (1) inside LDS until the tile is stable, then writes back; *changed is set when a
// tile promoted anything (its neighbours may need another round).
__global__ __launch_bounds__(256) void canny_hyst_kernel(uint8_t *__restrict__ map, int rows, int cols,
                                                          int *__restrict__ changed) {
    __shared__ uint8_t t[34][36];
    __shared__ int tile_changed, any;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    {   // all five loads per thread in flight at once (clamped addresses, zeroed outside the image)
        uint8_t v[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = threadIdx.x + 256 * k < 34 * 34 ? threadIdx.x + 256 * k : 34 * 34 - 1;
            const int ly = i / 34, lx = i - ly * 34;
            const int gy = y0 + ly - 1, gx = x0 + lx - 1;
            const uint8_t m = map[(size_t)clampi(gy, 0, rows - 1) * cols + clampi(gx, 0, cols - 1)];
            v[k] = ((unsigned)gy < (unsigned)rows && (unsigned)gx < (unsigned)cols) ? m : 0;
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = threadIdx.x + 256 * k;
            if (i < 34 * 34) t[i / 34][i % 34] = v[k];
        }
    }
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    {   // nothing to promote in a tile without candidates, or without a strong pixel in reach
        bool weak = false, strong = false;
        for (int i = threadIdx.x; i < 34 * 34; i += 256) {
            const int ly = i / 34, lx = i - ly * 34;
            const uint8_t m = t[ly][lx];
            weak |= m == 1 && ly >= 1 && ly <= 32 && lx >= 1 && lx <= 32;
            strong |= m == 2;
        }
        if (!__syncthreads_or(weak) || !__syncthreads_or(strong)) return;
    }
    for (;;) {
        if (threadIdx.x == 0) tile_changed = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * 32; i += 256) {
            const int ly = 1 + i / 32, lx = 1 + (i & 31);
            if (t[ly][lx] == 1) {
                const bool strong = t[ly - 1][lx - 1] == 2 || t[ly - 1][lx] == 2 || t[ly - 1][lx + 1] == 2 ||
                                    t[ly][lx - 1] == 2 || t[ly][lx + 1] == 2 || t[ly + 1][lx - 1] == 2 ||
                                    t[ly + 1][lx] == 2 || t[ly + 1][lx + 1] == 2;
                if (strong) {
                    t[ly][lx] = 2;  // monotone 1 -> 2: a racy read of the old value only delays it
                    tile_changed = 1;
                }
            }
        }
        __syncthreads();
        const int c = tile_changed;
        __syncthreads();
        if (!c) break;
        if (threadIdx.x == 0) any = 1;
    }
    __syncthreads();
    if (any) {
        for (int i = threadIdx.x; i < 32 * 32; i += 256) {
            const int ly = 1 + i / 32, lx = 1 + (i & 31);
            const int gy = y0 + ly - 1, gx = x0 + lx - 1;
            if (gy < rows && gx < cols) map[(size_t)gy * cols + gx] = t[ly][lx];
        }
        if (threadIdx.x == 0) *changed = 1;
    }
}

__global__ __launch_bounds__(256) void canny_edges_kernel(const uint8_t *__restrict__ map, int rows,
                                                           int cols, uint8_t *__restrict__ edges,
                                                           size_t estride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    edges[(size_t)y * estride + x] = map[(size_t)y * cols + x] == 2 ? 255 : 0;
}

}  // namespace micv

using namespace micv;

extern "C" int micv_generate_edge_dev(micv_ctx *ctx, const uint8_t *src, int rows, int cols,
                                      size_t stride, int gauss_size, double gauss_sigma,
                                      double low_thresh, double high_thresh, uint8_t *edges,
                                      size_t estride, micv_stream stream) {
    MICV_REQUIRE(ctx && src && edges, "micv_generate_edge: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && stride >= (size_t)cols && estride >= (size_t)cols,
                 "micv_generate_edge: bad size / stride");
    MICV_REQUIRE(gauss_size >= 1 && gauss_size <= 31 && (gauss_size & 1) && gauss_sigma > 0,
                 "micv_generate_edge: gaussian %d / sigma %g not supported (odd size <= 31, sigma > 0)",
                 gauss_size, gauss_sigma);
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)rows * cols;
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need(n, 4) * 3 + Carver::need(n, 1) * 2 + 256, &scratch));
    Carver c(scratch);
    float *buf = c.take<float>(n);
    int *mag = c.take<int>(n), *dxy = c.take<int>(n);
    uint8_t *blur = c.take<uint8_t>(n), *map = c.take<uint8_t>(n);
    int *changed = c.take<int>(1);
    const dim3 grid(cdiv(cols, 64), cdiv(rows, 4));
    const uint8_t *cin = src;
    size_t cstride = stride;
    if (gauss_size > 1) {  // a 1-tap Gaussian is the identity (Solution.cpp's problem-2 setting)
        Taps t;
        gaussian_taps(gauss_size, gauss_sigma, &t);
        gauss_u8_rows_kernel<<<grid, 256, 0, s>>>(src, stride, rows, cols, buf, t);
        MICV_LAUNCH_CHECK();
        gauss_u8_cols_kernel<<<grid, 256, 0, s>>>(buf, rows, cols, t, blur, (size_t)cols);
        MICV_LAUNCH_CHECK();
        cin = blur;
        cstride = cols;
    }
    double lo = low_thresh, hi = high_thresh;
    if (lo > hi) { const double tsw = lo; lo = hi; hi = tsw; }
    canny_grad_kernel<<<grid, 256, 0, s>>>(cin, cstride, rows, cols, mag, dxy);
    MICV_LAUNCH_CHECK();
    canny_map_kernel<<<grid, 256, 0, s>>>(mag, dxy, rows, cols, (int)std::floor(lo), (int)std::floor(hi), map);
    MICV_LAUNCH_CHECK();
    int *h_changed = static_cast<int *>(ctx->pinned);
    const int max_rounds = (int)(cdiv(cols, 32) * cdiv(rows, 32)) + 2;  // a path crosses each tile at most once per round
    // Data-dependent loop: rounds are enqueued four at a time and the host reads ONE flag per batch
    // (did the batch's last round still promote anything?).  A round after convergence changes
    // nothing, so overshooting is harmless; a host round trip per round was most of the time.
    constexpr int kBatch = 4;
    for (int round = 0; round < max_rounds; round += kBatch) {
        for (int b = 0; b < kBatch; b++) {
            MICV_HIP(hipMemsetAsync(changed, 0, 4, s));
            canny_hyst_kernel<<<dim3(cdiv(cols, 32), cdiv(rows, 32)), 256, 0, s>>>(map, rows, cols, changed);
            MICV_LAUNCH_CHECK();
        }
        MICV_HIP(hipMemcpyAsync(h_changed, changed, 4, hipMemcpyDeviceToHost, s));
        MICV_HIP(hipStreamSynchronize(s));
        if (!*h_changed) break;
    }
    canny_edges_kernel<<<grid, 256, 0, s>>>(map, rows, cols, edges, estride);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}
