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

// LDS-tiled forms of the two passes (the generic LK path runs 5 fields x 43 taps through them): the
// tile (+ n/2 halo, BORDER_REFLECT_101 resolved while loading, every load of the tile in flight at
// once) is staged once; a thread keeps a sliding window of the staged values in registers and runs 4
// (rows: four adjacent columns) / 8 (columns: eight adjacent rows) independent chains, taps four at a
// time -- one LDS read per 16 / 8 FMAs.  Same fmaf chain per output as the kernels above.
__global__ __launch_bounds__(256) void filter_rows_lds_kernel(const float *__restrict__ src,
                                                               int sstride, size_t sfield,
                                                               float *__restrict__ dst, int dstride,
                                                               size_t dfield, int rows, int cols,
                                                               Taps t) {
    constexpr int TW = 256, TR = 8;
    extern __shared__ float fl_lds[];
    // staged row: TW + n - 1 values, pitch rounded up to whole float4s plus one (the last chunk's second
    // ds_read_b128 reaches up to 7 floats past the thread's first column)
    const int a = t.n / 2, pw = ((TW + t.n - 1 + 3) & ~3) + 4;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TR;
    const float *sp = src + blockIdx.z * sfield;
    {
        // every load of the tile in flight at once (the kernel was bound by the latency of its staging,
        // two dependent batches per small workgroup); (row, column) of a thread's elements by stepping,
        // not by dividing by the run-time pitch
        constexpr int NB = (TR * (((TW + 62 + 3) & ~3) + 4) + 255) / 256;  // n <= 63
        float v[NB];
        int r = 0, c = threadIdx.x;
#pragma unroll
        for (int k = 0; k < NB; k++) {
            while (c >= pw) {
                c -= pw;
                r++;
            }
            const int rr = r < TR ? r : TR - 1;
            const int yy = y0 + rr < rows ? y0 + rr : rows - 1;
            v[k] = sp[(size_t)yy * sstride + reflect101(x0 - a + c, cols)];
            c += 256;
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = threadIdx.x + k * 256;
            if (i < TR * pw) fl_lds[i] = v[k];
        }
    }
    __syncthreads();
    // A thread finishes FOUR adjacent outputs of one row (thread = (row, group of 4 columns)): the taps
    // go four at a time, each chunk needs the 7 staged values [4c, 4c + 7) = the float4 it already holds
    // plus one new ds_read_b128, and does 16 FMAs -- one LDS instruction per 16 FMAs instead of one per
    // FMA (the kernel was LDS-issue bound).  Every output still runs its own fmaf chain over taps 0, 1, ...
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int q = threadIdx.x & 63;
#pragma unroll 1
    for (int r = threadIdx.x >> 6; r < TR; r += 4) {
    const f4 *lp4 = reinterpret_cast<const f4 *>(fl_lds + r * pw) + q;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    f4 cur = lp4[0];
    int c = 0;
    for (; c + 4 <= t.n; c += 4) {
        const f4 nxt = lp4[(c >> 2) + 1];
        const float w[8] = {cur.x, cur.y, cur.z, cur.w, nxt.x, nxt.y, nxt.z, nxt.w};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const float tap = t.k[c + kk];
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = fmaf(w[kk + j], tap, acc[j]);
        }
        cur = nxt;
    }
    if (c < t.n) {  // 1-3 left-over taps
        const f4 nxt = lp4[(c >> 2) + 1];
        const float w[8] = {cur.x, cur.y, cur.z, cur.w, nxt.x, nxt.y, nxt.z, nxt.w};
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            if (c + kk < t.n) {
                const float tap = t.k[c + kk];
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = fmaf(w[kk + j], tap, acc[j]);
            }
        }
    }
    if (y0 + r < rows) {
        float *o = dst + blockIdx.z * dfield + (size_t)(y0 + r) * dstride + x0 + 4 * q;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (x0 + 4 * q + j < cols) o[j] = acc[j];
    }
    }
}

__global__ __launch_bounds__(256) void filter_cols_lds_kernel(const float *__restrict__ src,
                                                               int sstride, size_t sfield,
                                                               float *__restrict__ dst, int dstride,
                                                               size_t dfield, int rows, int cols,
                                                               Taps t) {
    constexpr int TW = 64, TH = 32, RP = TH / 4;
    extern __shared__ float fl_lds[];
    const int a = t.n / 2, ph = TH + t.n - 1;  // staged rows
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const float *sp = src + blockIdx.z * sfield;
    const int c = threadIdx.x & 63, x = x0 + c < cols ? x0 + c : cols - 1;
    {
        // every load of the tile in flight at once: row r = (threadIdx.x >> 6) + 4 k of the staged column
        constexpr int NB = (32 + 62 + 3) / 4;  // n <= 63
        float v[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int r = 4 * k + (threadIdx.x >> 6);
            v[k] = sp[(size_t)reflect101(y0 - a + (r < ph ? r : ph - 1), rows) * sstride + x];
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int r = 4 * k + (threadIdx.x >> 6);
            if (r < ph) fl_lds[r * TW + c] = v[k];
        }
    }
    __syncthreads();
    if (x0 + c >= cols) return;
    const int rb = (threadIdx.x >> 6) * RP;
    float acc[RP];
#pragma unroll
    for (int j = 0; j < RP; j++) acc[j] = 0.f;
    // 8 vertically adjacent outputs per thread, taps four at a time: a chunk needs staged rows
    // [4c, 4c + 11) of the thread's column = 7 it already holds + 4 new LDS reads, and does 32 FMAs.
    const float *lp = fl_lds + rb * TW + c;
    float w[12];
#pragma unroll
    for (int i = 0; i < 7; i++) w[i] = lp[i * TW];
    int k0 = 0;
    for (; k0 + 4 <= t.n; k0 += 4) {
#pragma unroll
        for (int i = 7; i < 11; i++) w[i] = lp[(k0 + i) * TW];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const float tap = t.k[k0 + kk];
#pragma unroll
            for (int j = 0; j < RP; j++) acc[j] = fmaf(w[kk + j], tap, acc[j]);
        }
#pragma unroll
        for (int i = 0; i < 7; i++) w[i] = w[i + 4];
    }
    if (k0 < t.n) {  // 1-3 left-over taps: rows up to k0 + 2 + 7
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            if (k0 + kk < t.n) {
                w[7 + kk] = lp[(k0 + 7 + kk) * TW];
                const float tap = t.k[k0 + kk];
#pragma unroll
                for (int j = 0; j < RP; j++) acc[j] = fmaf(w[kk + j], tap, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RP; j++)
        if (y0 + rb + j < rows) dst[blockIdx.z * dfield + (size_t)(y0 + rb + j) * dstride + x0 + c] = acc[j];
}

int launch_filter_rows(hipStream_t s, const float *src, int sstride, size_t sfield, float *dst,
                       int dstride, size_t dfield, int rows, int cols, int nfields,
                       const Taps &t) {
    if (t.n >= 5) {
        filter_rows_lds_kernel<<<dim3(cdiv(cols, 256), cdiv(rows, 8), nfields), 256,
                                 (size_t)8 * (((256 + t.n - 1 + 3) & ~3) + 4) * sizeof(float), s>>>(
            src, sstride, sfield, dst, dstride, dfield, rows, cols, t);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    dim3 grid(cdiv(cols, 64), cdiv(rows, 4), nfields);
    filter_rows_kernel<<<grid, 256, 0, s>>>(src, sstride, sfield, dst, dstride, dfield, rows, cols,
                                            t);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int launch_filter_cols(hipStream_t s, const float *src, int sstride, size_t sfield, float *dst,
                       int dstride, size_t dfield, int rows, int cols, int nfields,
                       const Taps &t) {
    if (t.n >= 5) {
        filter_cols_lds_kernel<<<dim3(cdiv(cols, 64), cdiv(rows, 32), nfields), 256,
                                 (size_t)64 * (32 + t.n - 1) * sizeof(float), s>>>(
            src, sstride, sfield, dst, dstride, dfield, rows, cols, t);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    dim3 grid(cdiv(cols, 64), cdiv(rows, 4), nfields);
    filter_cols_kernel<<<grid, 256, 0, s>>>(src, sstride, sfield, dst, dstride, dfield, rows, cols,
                                            t);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// ---- fused Sobel pair, ksize 3 / 5 / 7 --------------------------------------------------------
// One pass over HBM (read the image once, write gx and gy: 12 B/px) instead of the four generic
// launches with two temporaries.  64x32 output tile per 256-thread workgroup: the source tile
// (+K/2 halo, BORDER_REFLECT_101 resolved while loading, every load in flight at once) goes to
// LDS, the row pass of both filters reads it once and keeps both results in LDS, the column
// pass finishes 8 rows per thread.  Each pass is the same fmaf chain from +0 as the generic
// kernels above, with the float row-pass result as the intermediate: identical bits.
template <int K>
struct SobelTaps {
    float row_dx[K], col_dx[K], row_dy[K], col_dy[K];
};

template <int K>
__global__ __launch_bounds__(256) void sobel_fused_kernel(const float *__restrict__ src, int sstride,
                                                           int rows, int cols, SobelTaps<K> t,
                                                           float *__restrict__ gx,
                                                           float *__restrict__ gy, int gstride) {
    constexpr int A = K / 2, TW = 64, TH = 32, RW = TW + 2 * A, RH = TH + 2 * A;
    constexpr int SS = RW | 1;  // odd pitch: the column-strided staging stores spread over banks
    __shared__ float S[RH * SS];
    __shared__ float DX[RH * TW], DY[RH * TW];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    {
        constexpr int NB = (RH * RW + 255) / 256;
        float v[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = threadIdx.x + k * 256 < RH * RW ? threadIdx.x + k * 256 : RH * RW - 1;
            const int ly = i / RW, lx = i - ly * RW;
            const int yy = reflect101(y0 - A + ly, rows), xx = reflect101(x0 - A + lx, cols);
            v[k] = src[(size_t)yy * sstride + xx];
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = threadIdx.x + k * 256;
            if (i < RH * RW) {
                const int ly = i / RW, lx = i - ly * RW;
                S[ly * SS + lx] = v[k];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RH * TW; i += 256) {
        const int r = i / TW, c = i - r * TW;
        const float *sp = S + r * SS + c;
        float ax = 0.f, ay = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const float v = sp[k];
            ax = fmaf(v, t.row_dx[k], ax);
            ay = fmaf(v, t.row_dy[k], ay);
        }
        DX[i] = ax;
        DY[i] = ay;
    }
    __syncthreads();
    const int c = threadIdx.x & 63, x = x0 + c;
    if (x >= cols) return;
    const int rb = (threadIdx.x >> 6) * (TH / 4);
    // marching register window over the column: K + 7 reads per plane for 8 outputs
    float wx[K + TH / 4 - 1], wy[K + TH / 4 - 1];
#pragma unroll
    for (int k = 0; k < K + TH / 4 - 1; k++) {
        wx[k] = DX[(rb + k) * TW + c];
        wy[k] = DY[(rb + k) * TW + c];
    }
#pragma unroll
    for (int j = 0; j < TH / 4; j++) {
        const int y = y0 + rb + j;
        if (y >= rows) break;
        float ax = 0.f, ay = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) {
            ax = fmaf(wx[j + k], t.col_dx[k], ax);
            ay = fmaf(wy[j + k], t.col_dy[k], ay);
        }
        gx[(size_t)y * gstride + x] = ax;
        gy[(size_t)y * gstride + x] = ay;
    }
}

template <int K>
static int launch_sobel_fused(hipStream_t s, const float *src, int rows, int cols, int sstride,
                              float scale, float *gx, float *gy, int gstride) {
    Taps d, m;
    sobel_taps(K, 1, &d);
    sobel_taps(K, 0, &m);
    SobelTaps<K> t;
    for (int i = 0; i < K; i++) {
        // d/dx: row kernel = derivative, column kernel = smoothing * scale;  d/dy: the transpose
        t.row_dx[i] = d.k[i];
        t.col_dx[i] = scale != 1.f ? m.k[i] * scale : m.k[i];
        t.row_dy[i] = scale != 1.f ? m.k[i] * scale : m.k[i];
        t.col_dy[i] = d.k[i];
    }
    sobel_fused_kernel<K><<<dim3(cdiv(cols, 64), cdiv(rows, 32)), 256, 0, s>>>(src, sstride, rows, cols, t,
                                                                            gx, gy, gstride);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// cv::cuda::createSobelFilter: `if (dx == 0) kx *= scale; else ky *= scale;` then a
// separable filter, row kernel kx, column kernel ky.
int sobel_dev(hipStream_t s, const float *src, int rows, int cols, int sstride, int ksize,
              float scale, float *gx, float *gy, int gstride, float *tmp, bool force_generic) {
    Taps kx, ky;
    const size_t n = (size_t)rows * cols;
    // d/dx
    if (sobel_taps(ksize, 1, &kx) < 0 || sobel_taps(ksize, 0, &ky) < 0) {
        set_error("sobel: kernel size %d not supported (1,3,5,7..31 odd)", ksize);
        return MICV_EINVAL;
    }
    if (!force_generic) {
        if (ksize == 3) return launch_sobel_fused<3>(s, src, rows, cols, sstride, scale, gx, gy, gstride);
        if (ksize == 5) return launch_sobel_fused<5>(s, src, rows, cols, sstride, scale, gx, gy, gstride);
        if (ksize == 7) return launch_sobel_fused<7>(s, src, rows, cols, sstride, scale, gx, gy, gstride);
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
                     scale, gx, gy, (int)(gstride / 4), static_cast<float *>(scratch),
                     ctx->opt[MICV_OPT_SOBEL_GENERIC] != 0);
}
