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

// ---- mhi::frameDifference on bit planes (r05) ----------------------------------------------------------------------
// Two launches, no float or byte temporaries in HBM (r04: four launches through an f32 plane pair and two byte planes,
// 0.072 ms at 1080p = 0.011 of the HBM roofline):
//   mhi_blur_diff_bits_kernel   both frames' Gaussian blur in one LDS tile (row pass of both frames -> LDS, column pass,
//                               saturate_cast<uchar>), saturating subtract, AbsThreshold; a wave is one 64-column row
//                               segment, so its BALLOT is the mask: one 64-bit word per (row, 64-column tile), 1/8 B
//                               per pixel instead of 9.
//   mhi_open7_bits_kernel       the 7x7 elliptical OPEN on those words: a wave owns 64 columns x 52 rows, lane = row, one
//                               128-bit register pair = the row with 6 columns either side; a horizontal erosion /
//                               dilation of half-width r is a shift-and-combine ladder (3 steps for r = 3), the vertical
//                               combination takes the neighbours' rows by wave shuffles -- no LDS, no barriers, 7 row
//                               terms per pass.  BORDER_REFLECT_101 (cv::cuda pads every pass): the mask is EXTENDED by
//                               reflection (rows: the lanes beyond the image load the reflected row; columns: edge tiles
//                               assemble their 76 bits one by one), and because the ellipse is symmetric in both axes the
//                               erosion of the extended mask IS the reflected extension of the eroded image, so the
//                               dilation's padding needs no second pass.
__global__ __launch_bounds__(256) void mhi_blur_diff_bits_kernel(const uint8_t *__restrict__ f1, const uint8_t *__restrict__ f2,
                                                                  size_t stride, int rows, int cols, Taps tx, Taps ty, double thresh,
                                                                  unsigned long long *__restrict__ words, int tiles_x) {
    constexpr int TW = 64, TH = 16;
    // dynamic LDS: the source tile of both frames as floats (every pixel converted once, the reflected border resolved
    // once), then the row-pass values of both frames
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int ax = tx.n / 2, ay = ty.n / 2, nrp = TH + 2 * ay, SW = TW + 2 * ax;
    float *S = sm, *RP = sm + 2 * nrp * SW;  // S[f][r][cx], RP[f][r][c]
    // the taps in LDS: a run-time index into the by-value Taps is a dependent scalar load per tap (measured: 22 us at
    // blur 5), 62 constant-index taps do not fit the scalar registers (62 spills, slower still)
    __shared__ float TX[32], TY[32];
    if (threadIdx.x < 32) {
        TX[threadIdx.x] = threadIdx.x < (unsigned)tx.n ? tx.k[threadIdx.x] : 0.f;
        TY[threadIdx.x] = threadIdx.x < (unsigned)ty.n ? ty.k[threadIdx.x] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a wave stages whole rows (64 consecutive bytes per load), FOUR rows' loads in flight at a time: a load -> convert
    // -> ds_write loop serialises on every round trip (18 of them per wave at blur 3 = the whole launch time)
    const int c1 = reflect101(x0 - ax + lane, cols), c2 = reflect101(x0 - ax + lane + 64 < x0 + TW + ax ? x0 - ax + lane + 64 : x0, cols);
    const bool second = lane + 64 < SW;
    for (int j0 = wave; j0 < 2 * nrp; j0 += 16) {
        uint8_t va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 4 * u < 2 * nrp ? j0 + 4 * u : 2 * nrp - 1;
            const int f = j >= nrp, r = j - f * nrp;
            const uint8_t *s = (f ? f2 : f1) + (size_t)reflect101(y0 - ay + r, rows) * stride;
            va[u] = s[c1];
            vb[u] = s[c2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + 4 * u;
            if (j < 2 * nrp) {
                S[j * SW + lane] = (float)va[u];
                if (second) S[j * SW + lane + 64] = (float)vb[u];
            }
        }
    }
    __syncthreads();
    const int c = lane, x = x0 + c;
    const int ntx = tx.n, nty = ty.n;
    for (int j = wave; j < 2 * nrp; j += 4) {
        const float *sp = S + j * SW + c;
        float acc = 0.f;
#pragma unroll 4
        for (int k = 0; k < ntx; k++) acc = fmaf(sp[k], TX[k], acc);
        RP[j * TW + c] = acc;
    }
    __syncthreads();
    // column pass, saturate, saturating subtract f2 - f1 (cv::cuda::subtract on CV_8U), AbsThreshold (MotionHistory.cu:17-48)
    for (int r = wave; r < TH; r += 4) {  // (wave-uniform)
        const int y = y0 + r;
        if (y >= rows) break;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll 4
        for (int k = 0; k < nty; k++) {
            a1 = fmaf(RP[(r + k) * TW + c], TY[k], a1);
            a2 = fmaf(RP[(nrp + r + k) * TW + c], TY[k], a2);
        }
        const int d = (int)sat_u8_rn(a2) - (int)sat_u8_rn(a1);
        const int val = d < 0 ? 0 : d;
        const bool on = x < cols && ((double)val >= thresh || (double)(-val) >= thresh);
        const unsigned long long m = __ballot(on);
        if (c == 0) words[(size_t)y * tiles_x + blockIdx.x] = m;
    }
}

struct U128 {
    unsigned long long lo, hi;
};
__device__ __forceinline__ U128 u_and(U128 a, U128 b) { return {a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ U128 u_or(U128 a, U128 b) { return {a.lo | b.lo, a.hi | b.hi}; }
template <int T>
__device__ __forceinline__ U128 u_shr(U128 a) { return {(a.lo >> T) | (a.hi << (64 - T)), a.hi >> T}; }  // bit k <- bit k + T
template <int T>
__device__ __forceinline__ U128 u_shl(U128 a) { return {a.lo << T, (a.hi << T) | (a.lo >> (64 - T))}; }  // bit k <- bit k - T
__device__ __forceinline__ U128 u_shfl(U128 a, int src_lane) {
    U128 r;
    r.lo = __shfl(a.lo, src_lane, 64);
    r.hi = __shfl(a.hi, src_lane, 64);
    return r;
}

// Horizontal pass of one ellipse row: bit k of H<r> = AND (OR) of bits k - r .. k + r.  The 7x7 ellipse of
// cv::getStructuringElement has rows of half-width 0, 2, 3, 3, 3, 2, 0 (ellipse7() below; checked by the host).
template <bool DILATE>
__device__ __forceinline__ void morph_rows(U128 e, U128 &h2, U128 &h3) {
    auto op = [](U128 a, U128 b) { return DILATE ? u_or(a, b) : u_and(a, b); };
    const U128 w2 = op(e, u_shr<1>(e));    // bits k, k + 1
    const U128 w4 = op(w2, u_shr<2>(w2));  // k .. k + 3
    const U128 w5 = op(w4, u_shr<1>(w4));  // k .. k + 4
    const U128 w7 = op(w4, u_shr<3>(w4));  // k .. k + 6
    h2 = u_shl<2>(w5);                     // centred: k - 2 .. k + 2
    h3 = u_shl<3>(w7);                     // k - 3 .. k + 3
}
// The vertical combination: row y of the result = rows y - 3 .. y + 3 through their ellipse rows.
template <bool DILATE>
__device__ __forceinline__ U128 morph7_bits(U128 e, int lane) {
    U128 h2, h3;
    morph_rows<DILATE>(e, h2, h3);
    auto op = [](U128 a, U128 b) { return DILATE ? u_or(a, b) : u_and(a, b); };
    U128 r = h3;
    r = op(r, u_shfl(h3, lane - 1));
    r = op(r, u_shfl(h3, lane + 1));
    r = op(r, u_shfl(h2, lane - 2));
    r = op(r, u_shfl(h2, lane + 2));
    r = op(r, u_shfl(e, lane - 3));
    r = op(r, u_shfl(e, lane + 3));
    return r;  // (lanes within 3 of the wave's ends hold wrapped rows: the caller uses the inner lanes only)
}

__global__ __launch_bounds__(256) void mhi_open7_bits_kernel(const unsigned long long *__restrict__ words, int tiles_x, int rows,
                                                              int cols, uint8_t *__restrict__ dst, size_t dstride) {
    constexpr int OUT_ROWS = 52;  // 64 lanes = 52 output rows + 6 either side (3 for the erosion, 3 for the dilation)
    __shared__ unsigned long long W[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = blockIdx.x, x0 = 64 * tx;
    const int yb = (blockIdx.y * 4 + wave) * OUT_ROWS;
    if (yb >= rows) return;  // (wave-uniform; no workgroup barrier below)
    const int y = yb + lane - 6;
    const unsigned long long *row = words + (size_t)reflect101(y, rows) * tiles_x;
    // the row with 6 columns either side: bit k <-> column x0 - 6 + k, k = 0 .. 75
    U128 e;
    if (x0 - 6 >= 0 && x0 + 70 <= cols) {
        const unsigned long long cw = row[tx], lw = tx > 0 ? row[tx - 1] : 0ull, rw = x0 + 64 < cols ? row[tx + 1] : 0ull;
        e.lo = (lw >> 58) | (cw << 6);
        e.hi = ((cw >> 58) | (rw << 6)) & 0xFFFull;
    } else {
        // edge tiles: the columns outside the image by reflection (also the centre word's columns past a ragged right
        // edge).  Their sources lie in this tile or a neighbour: three words loaded up front, no memory in the loop.
        const int t0 = tx > 0 ? tx - 1 : 0, t2 = tx + 1 < tiles_x ? tx + 1 : tiles_x - 1;
        const unsigned long long w0 = row[t0], w1 = row[tx], w2 = row[t2];
        e.lo = ((tx > 0 ? w0 : 0ull) >> 58) | (w1 << 6);
        e.hi = ((w1 >> 58) | ((tx + 1 < tiles_x ? w2 : 0ull) << 6)) & 0xFFFull;
        const int k_lo_end = x0 - 6 < 0 ? 6 - x0 : 0;                       // columns < 0: k in [0, k_lo_end)
        const int k_hi_begin = cols - (x0 - 6) < 76 ? cols - (x0 - 6) : 76;  // columns >= cols: k in [k_hi_begin, 76)
        for (int k = 0; k < 76; k++) {
            if (k >= k_lo_end && k < k_hi_begin) continue;
            const int xx = reflect101(x0 - 6 + k, cols), tw = xx >> 6;
            const unsigned long long wsel = tw == tx ? w1 : (tw == t0 ? w0 : w2);
            const unsigned long long bit = (wsel >> (xx & 63)) & 1ull;
            if (k < 64) e.lo = (e.lo & ~(1ull << k)) | (bit << k);
            else e.hi = (e.hi & ~(1ull << (k - 64))) | (bit << (k - 64));
        }
    }
    const U128 er = morph7_bits<false>(e, lane);   // valid: lanes 3 .. 60, bits 3 .. 72
    const U128 op = morph7_bits<true>(er, lane);   // valid: lanes 6 .. 57, bits 6 .. 69
    W[wave][lane] = (op.lo >> 6) | (op.hi << 58);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // bytes out: four rows per wave-instruction, a lane expands four mask bits into one dword (64 contiguous bytes per row)
    const int d4 = lane & 15, rg = lane >> 4;
    for (int it = 0; it < OUT_ROWS / 4; it++) {
        const int r = 4 * it + rg, yo = yb + r;
        if (yo >= rows) continue;
        const unsigned bits = (unsigned)(W[wave][r + 6] >> (4 * d4)) & 0xFu;
        const unsigned bytes = (bits * 0x00204081u) & 0x01010101u;
        const int xo = x0 + 4 * d4;
        uint8_t *o = dst + (size_t)yo * dstride + xo;
        if (xo + 3 < cols && ((reinterpret_cast<uintptr_t>(o) & 3) == 0)) {
            *reinterpret_cast<unsigned *>(o) = bytes;
        } else {
            for (int b = 0; b < 4; b++)
                if (xo + b < cols) o[b] = (uint8_t)((bytes >> (8 * b)) & 1u);
        }
    }
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
    const int tiles_x = cdiv(cols, 64);
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need((size_t)rows * tiles_x, 8), &scratch));
    unsigned long long *words = static_cast<unsigned long long *>(scratch);
    Taps t, ty;  // cv::Size(width, height): width taps along x, height taps along y
    gaussian_taps(blur_w, blur_sigma, &t);
    gaussian_taps(blur_h, blur_sigma, &ty);
    // the bit-plane open is written for the 7x7 ellipse's row half-widths 0, 2, 3, 3, 3, 2, 0
    {
        const Ellipse7 e = ellipse7();
        static const unsigned char want[7] = {0x08, 0x3E, 0x7F, 0x7F, 0x7F, 0x3E, 0x08};
        for (int i = 0; i < 7; i++)
            if (e.m[i] != want[i]) {
                set_error("micv_mhi_frame_difference: unexpected 7x7 ellipse row %d = %#x", i, e.m[i]);
                return MICV_EUNSUPPORTED;
            }
    }
    const size_t lds = (size_t)2 * (16 + 2 * (blur_h / 2)) * (64 + 2 * (blur_w / 2) + 64) * sizeof(float);  // <= 58 KB at 31 x 31
    mhi_blur_diff_bits_kernel<<<dim3(tiles_x, cdiv(rows, 16)), 256, lds, s>>>(f1, f2, stride, rows, cols, t, ty, thresh, words, tiles_x);
    MICV_LAUNCH_CHECK();
    mhi_open7_bits_kernel<<<dim3(tiles_x, cdiv(cdiv(rows, 52), 4)), 256, 0, s>>>(words, tiles_x, rows, cols, diff, dstride);
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
