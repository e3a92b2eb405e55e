// harris.hip -- ps4: Harris response (a9), threshold + NMS + ordered corner list (a10),
// SIFT-style keypoint angles (a11).
#include "compact.hpp"
#include "kernels.hpp"

namespace micv {

// ---- a9: corner response ---------------------------------------------------------------------
// Tile 64x16 outputs; Ix, Iy staged in LDS with a clamped (2r)-halo (Harris.cpp:73-76 /
// texture clamp on the CUDA path).  Accumulation exactly as Harris.cu:36-43,85: per tap
// M = fma(w, I, M) in (wy, wx) raster order, w = g[wy]*g[wx] (float), then :87-91 in float.
// R from the second-moment sums.  CPU = harris::cpu as written (Harris.cpp:91-92): cv::determinant of a 2x2 CV_32F
// matrix is det2 in double, `harrisScore * trace * trace` float arithmetic, the difference taken in double and
// stored to float (one rounding).  Otherwise harris::gpu (Harris.cu:87-91): float throughout, unfused.
template <bool CPU>
__device__ __forceinline__ float harris_r(float mxx, float mxy, float myy, float alpha) {
    const float trace = mxx + myy;
    if (CPU) {
        const double det = (double)mxx * (double)myy - (double)mxy * (double)mxy;
        return (float)(det - (double)(alpha * trace * trace));
    }
    const float det = mxx * myy - mxy * mxy;
    return det - alpha * trace * trace;
}

template <bool CPU>
__global__ __launch_bounds__(256) void harris_response_kernel(const float *__restrict__ gx,
                                                               const float *__restrict__ gy,
                                                               int gstride, int rows, int cols,
                                                               int r, Taps g, float alpha,
                                                               float *__restrict__ resp,
                                                               int rstride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TW = 64, TH = 16;
    const int RW = TW + 2 * r, RH = TH + 2 * r;
    float *sx = lds, *sy = lds + RW * RH;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    for (int i = threadIdx.x; i < RW * RH; i += 256) {
        const int ly = i / RW, lx = i - ly * RW;
        const int yy = clampi(y0 - r + ly, 0, rows - 1), xx = clampi(x0 - r + lx, 0, cols - 1);
        sx[i] = gx[(size_t)yy * gstride + xx];
        sy[i] = gy[(size_t)yy * gstride + xx];
    }
    __syncthreads();
    const int c = threadIdx.x & 63;
    const int x = x0 + c;
    for (int ry = threadIdx.x >> 6; ry < TH; ry += 4) {
        const int y = y0 + ry;
        if (x >= cols || y >= rows) continue;
        float mxx = 0.f, mxy = 0.f, myy = 0.f;
        for (int wy = 0; wy <= 2 * r; wy++) {
            const float gw = g.k[wy];
            const float *px = sx + (ry + wy) * RW + c, *py = sy + (ry + wy) * RW + c;
            for (int wx = 0; wx <= 2 * r; wx++) {
                const float ix = px[wx], iy = py[wx];
                const float w = gw * g.k[wx];
                if (CPU) {  // Harris.cpp:81-87: `secondMoment + weight * gradVals`, a multiply then an add
                    mxx = mxx + w * (ix * ix);
                    mxy = mxy + w * (ix * iy);
                    myy = myy + w * (iy * iy);
                } else {
                    mxx = fmaf(w, ix * ix, mxx);
                    mxy = fmaf(w, ix * iy, mxy);
                    myy = fmaf(w, iy * iy, myy);
                }
            }
        }
        resp[(size_t)y * rstride + x] = harris_r<CPU>(mxx, mxy, myy, alpha);
    }
}

// Windows 3 / 5 / 7 (the configuration uses 5): the three product fields Ix^2, IxIy, Iy^2 are
// formed ONCE per staged cell (not once per tap) and kept in LDS; a thread owns 4 adjacent outputs
// of one row and reads its (4 + 2r)-wide windows with ds_read_b128 (pitch 80 floats and the
// lane -> (row = lane & 3, group = lane >> 2) mapping make them conflict-free); the weight table
// w[wy][wx] = g[wy]*g[wx] (float product, as above) arrives in SGPRs.  Every accumulator is still
// the (wy, wx)-raster fmaf chain of Harris.cu:36-43, so the bits do not change.
//
// Outputs (0, 1) and (2, 3) share every weight, so each field's four chains are two v_pk_fma_f32
// chains: tap wx multiplies the window's cells (wx, wx + 1) / (wx + 2, wx + 3), which for even wx
// are the register pairs the b128 reads delivered and for odd wx one v_pk_mov_b32 apart (three
// odd pairs per field and window row): 39 VALU instructions per window row instead of 60.
// Interior tiles stage with aligned 16-byte loads (columns x0 - 4 .. x0 + 67, the four left of
// the halo dropped) and store float4; tiles touching the left / right edge, or unaligned images,
// take the clamped scalar loads.
template <int R>
struct HarrisW {
    float w[(2 * R + 1) * (2 * R + 1)];
};
typedef float hv4f __attribute__((ext_vector_type(4)));
typedef float hv2f __attribute__((ext_vector_type(2)));

// The window sums and R of a tile whose three product planes are in LDS (PS floats per row, LDS column = global
// column - (x0 - 4)): shared by the kernel that stages gradients from HBM and the one that forms them from the image.
template <int R, int TH, bool CPU>
__device__ __forceinline__ void harris_tile_accumulate(const float *XX, const float *XY, const float *YY, const HarrisW<R> &hw,
                                                       float alpha, float *__restrict__ resp, int rstride, int rows, int cols,
                                                       int x0, int y0, int vec_ok) {
    constexpr int W = 2 * R + 1, PS = 80, SH = 4 - R;
    constexpr int NV = (SH + 4 + 2 * R + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ry = 4 * wave + (lane & 3), c0 = 4 * (lane >> 2);
    // [field][output pair]
    hv2f acc[3][2];
#pragma unroll
    for (int f = 0; f < 3; f++) acc[f][0] = acc[f][1] = (hv2f){0.f, 0.f};
    const float *planes[3] = {XX, XY, YY};
#pragma unroll
    for (int wy = 0; wy < W; wy++) {
        const int o = (ry + wy) * PS + c0;
#pragma unroll
        for (int f = 0; f < 3; f++) {
            // LDS cells c0 .. c0 + 4 NV - 1 of the window row: even pairs E[k] = (2k, 2k + 1) as read, odd pairs
            // O[k] = (2k + 1, 2k + 2) one v_pk_mov_b32 away (the unused ones are dropped by the compiler)
            hv2f E[2 * NV], O[2 * NV - 1];
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const hv4f v = *reinterpret_cast<const hv4f *>(planes[f] + o + 4 * i);
                E[2 * i] = (hv2f){v.x, v.y};
                E[2 * i + 1] = (hv2f){v.z, v.w};
            }
#pragma unroll
            for (int k = 0; k + 1 < 2 * NV; k++)  // dst.lo = E[k].hi, dst.hi = E[k + 1].lo
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(O[k]) : "v"(E[k]), "v"(E[k + 1]));
#pragma unroll
            for (int wx = 0; wx < W; wx++) {
                const float w = hw.w[wy * W + wx];
                const hv2f w2 = (hv2f){w, w};
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int cell = SH + 2 * j + wx;
                    const hv2f d = (cell & 1) ? O[cell >> 1] : E[cell >> 1];
                    if (CPU)  // harris::cpu: the product rounded, then added (v_pk_mul_f32 + v_pk_add_f32; -ffp-contract=off)
                        acc[f][j] = acc[f][j] + w2 * d;
                    else
                        acc[f][j] = __builtin_elementwise_fma(w2, d, acc[f][j]);
                }
            }
        }
        // pin the accumulators here: the row's FMAs must finish before the next row's reads are
        // issued (the compiler otherwise hoists all the reads and holds every window register)
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                          "+v"(acc[2][1])
                     :: "memory");
    }
    const int y = y0 + ry;
    if (y >= rows) return;
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        out[j] = harris_r<CPU>(acc[0][j >> 1][j & 1], acc[1][j >> 1][j & 1], acc[2][j >> 1][j & 1], alpha);
    }
    const int x = x0 + c0;
    if (vec_ok && x + 3 < cols) {
        *reinterpret_cast<hv4f *>(resp + (size_t)y * rstride + x) = (hv4f){out[0], out[1], out[2], out[3]};
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (x + j < cols) resp[(size_t)y * rstride + x + j] = out[j];
    }
}

template <int R, int TH, bool CPU = false>
__global__ __launch_bounds__(16 * TH) void harris_response_tiled_kernel(
    const float *__restrict__ gx, const float *__restrict__ gy, int gstride, int rows, int cols,
    HarrisW<R> hw, float alpha, float *__restrict__ resp, int rstride, int vec_ok) {
    // LDS column = global column - (x0 - 4): the 72 columns x0 - 4 .. x0 + 67 are 18 aligned float4 of the
    // image, written whole (the window of output column c starts at LDS column c + 4 - R)
    constexpr int TW = 64, NT = 16 * TH, RH = TH + 2 * R, PS = 80, SH = 4 - R, V4 = (TW + 8) / 4;
    constexpr int NV = (SH + 4 + 2 * R + 3) / 4;  // float4 per window row, from the thread's own column group
    static_assert(R >= 1 && R <= 4 && 60 + 4 * NV <= PS, "window reads stay inside a plane row");
    __shared__ __attribute__((aligned(16))) float XX[RH * PS], XY[RH * PS], YY[RH * PS];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    constexpr int NB = (RH * V4 + NT - 1) / NT;
    hv4f vx[NB], vy[NB];
    if (vec_ok && x0 >= 4 && x0 + TW + 4 <= cols) {
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = threadIdx.x + k * NT < RH * V4 ? threadIdx.x + k * NT : RH * V4 - 1;
            const int ly = i / V4, m = i - ly * V4;
            const int yy = clampi(y0 - R + ly, 0, rows - 1);
            const size_t o = (size_t)yy * gstride + (x0 - 4 + 4 * m);
            vx[k] = *reinterpret_cast<const hv4f *>(gx + o);
            vy[k] = *reinterpret_cast<const hv4f *>(gy + o);
        }
    } else {
        // tiles at the left / right edge (and unaligned images): clamped scalar loads, same slots
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = threadIdx.x + k * NT < RH * V4 ? threadIdx.x + k * NT : RH * V4 - 1;
            const int ly = i / V4, m = i - ly * V4;
            const int yy = clampi(y0 - R + ly, 0, rows - 1);
            const float *rx = gx + (size_t)yy * gstride, *ry_ = gy + (size_t)yy * gstride;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int xx = clampi(x0 - 4 + 4 * m + e, 0, cols - 1);
                vx[k][e] = rx[xx];
                vy[k][e] = ry_[xx];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int i = threadIdx.x + k * NT;
        if (i < RH * V4) {
            const int ly = i / V4, m = i - ly * V4;
            *reinterpret_cast<hv4f *>(XX + ly * PS + 4 * m) = vx[k] * vx[k];
            *reinterpret_cast<hv4f *>(XY + ly * PS + 4 * m) = vx[k] * vy[k];
            *reinterpret_cast<hv4f *>(YY + ly * PS + 4 * m) = vy[k] * vy[k];
        }
    }
    __syncthreads();
    harris_tile_accumulate<R, TH, CPU>(XX, XY, YY, hw, alpha, resp, rstride, rows, cols, x0, y0, vec_ok);
}

template <int R, bool CPU>
static void launch_harris_tiled(hipStream_t s, const float *gx, const float *gy, int gstride, int rows,
                                int cols, const Taps &g, float alpha, float *resp, int rstride) {
    HarrisW<R> hw;
    for (int wy = 0; wy <= 2 * R; wy++)
        for (int wx = 0; wx <= 2 * R; wx++) hw.w[wy * (2 * R + 1) + wx] = g.k[wy] * g.k[wx];
    const int vec_ok = ((reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(resp)) & 15) == 0 &&
                       (gstride & 3) == 0 && (rstride & 3) == 0;
    // 64x32 tiles (512 threads) carry a sixth less halo per output; small images keep 64x16 for more blocks
    if ((size_t)cdiv(cols, 64) * cdiv(rows, 32) >= 1024)
        harris_response_tiled_kernel<R, 32, CPU><<<dim3(cdiv(cols, 64), cdiv(rows, 32)), 512, 0, s>>>(
            gx, gy, gstride, rows, cols, hw, alpha, resp, rstride, vec_ok);
    else
        harris_response_tiled_kernel<R, 16, CPU><<<dim3(cdiv(cols, 64), cdiv(rows, 16)), 256, 0, s>>>(
            gx, gy, gstride, rows, cols, hw, alpha, resp, rstride, vec_ok);
}

// ---- image -> R in one launch (r05; the ps4 caller's chain, ps4_cpp/src/Solution.cpp:77-124) -------------------
// harris::getGradients (3x3 Sobel, scale 1, BORDER_REFLECT_101; Harris.cpp:14-41) computed IN the response kernel's
// tile: the image tile (+R for the window, +1 for the Sobel) goes to LDS with the reflection resolved while loading,
// the Sobel pair runs as sobel_fused_kernel<3> runs it (row pass of both filters into LDS, column pass: the same fmaf
// chains from +0 with the float row-pass value in between -- same bits as micv_sobel_dev), the products Ix^2, IxIy,
// Iy^2 go straight into the three planes harris_tile_accumulate reads.  A plane cell outside the image is the
// gradient at the CLAMPED coordinate (Harris.cpp:73-76), i.e. the Sobel of the edge pixel -- formed here from the
// staged neighbourhood of that pixel, not by reflecting the gradient.  8 B per pixel of HBM traffic (image in, R out)
// instead of 12 + 12; the gradients are written as well when the caller wants them (sift::getKeypoints does).
struct Sobel3 {
    float row_dx[3], col_dx[3], row_dy[3], col_dy[3];
};

template <int R, int TH, bool CPU>
__global__ __launch_bounds__(16 * TH) void harris_image_response_kernel(const float *__restrict__ img, int istride, int rows, int cols,
                                                                         Sobel3 t, HarrisW<R> hw, float alpha, float *__restrict__ resp,
                                                                         int rstride, float *__restrict__ gxo, float *__restrict__ gyo,
                                                                         int gstride, int vec_ok, int img_vec_ok) {
    constexpr int TW = 64, NT = 16 * TH, RH = TH + 2 * R, PS = 80, PC = TW + 8;  // plane columns x0 - 4 .. x0 + 67
    // LDS: the three product planes and nothing else -- the same 34.5 KB (window 5, 64x32 tile) as the kernel that reads
    // gradients from HBM, i.e. the same four workgroups per CU (a staged image tile + row-pass planes took 56 KB = two
    // workgroups per CU and the launch 43-49 us at 4K; the image is read through L1 / L2 instead)
    __shared__ __attribute__((aligned(16))) float lds[3 * RH * PS];
    float *XX = lds, *XY = lds + RH * PS, *YY = lds + 2 * RH * PS;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const hv2f rdx0 = {t.row_dx[0], t.row_dx[0]}, rdx1 = {t.row_dx[1], t.row_dx[1]}, rdx2 = {t.row_dx[2], t.row_dx[2]};
    const hv2f rdy0 = {t.row_dy[0], t.row_dy[0]}, rdy1 = {t.row_dy[1], t.row_dy[1]}, rdy2 = {t.row_dy[2], t.row_dy[2]};
    const hv2f cdx0 = {t.col_dx[0], t.col_dx[0]}, cdx1 = {t.col_dx[1], t.col_dx[1]}, cdx2 = {t.col_dx[2], t.col_dx[2]};
    const hv2f cdy0 = {t.col_dy[0], t.col_dy[0]}, cdy1 = {t.col_dy[1], t.col_dy[1]}, cdy2 = {t.col_dy[2], t.col_dy[2]};
    const hv2f zero = {0.f, 0.f};
    // Marching jobs: two adjacent plane columns x SEG rows per thread, a three-row register window of row-pass values
    // (packed f32: both columns per instruction), products straight into the planes.  The row pass of both filters and
    // the column pass are sobel_fused_kernel<3>'s fmaf chains from +0 with the float row-pass value in between.
    constexpr int SEG = 3, NSEG = (RH + SEG - 1) / SEG, NJ = (PC / 2) * NSEG;
    // (rows: the jobs load SEG + 2 rows each, NSEG * SEG + 2 in all -- up to two more than the TH + 2 R + 2 the tile needs
    // when RH is not a multiple of SEG; the test covers what is LOADED, not what is used: ADVICE r5, an exact-size image
    // with rows % TH in 2..5 was read one or two rows past its end)
    const bool interior = img_vec_ok && x0 >= 8 && x0 + TW + 8 <= cols && y0 - R - 1 >= 0 && y0 - R + NSEG * SEG + 1 <= rows;
    for (int n = threadIdx.x; n < NJ; n += NT) {
        const int seg = n / (PC / 2), lx = 2 * (n - seg * (PC / 2));
        const int q0 = seg * SEG;
        // the job's image values: rows q0 - 1 .. q0 + SEG of the plane rows' centres, columns lx - 1 .. lx + 2 of the
        // pair's -- every load issued before the first use
        hv2f va[SEG + 2], vb[SEG + 2], vc[SEG + 2];
        if (interior) {
            const float *sp = img + (size_t)(y0 - R + q0 - 1) * istride + (x0 - 4 + lx - 1);  // (odd column: 4-byte aligned pairs)
#pragma unroll
            for (int k = 0; k < SEG + 2; k++) {
                const float *rp = sp + (size_t)k * istride;
                const float v0 = rp[0];
                const hv2f v12 = *reinterpret_cast<const hv2f *>(rp + 1);
                const float v3 = rp[3];
                va[k] = (hv2f){v0, v12.x};
                vb[k] = v12;
                vc[k] = (hv2f){v12.y, v3};
            }
        } else {
            // border tiles: a plane cell outside the image is the gradient at the CLAMPED coordinate, whose own 3x3
            // neighbourhood is read with BORDER_REFLECT_101 -- per column of the pair, since the two may clamp to the
            // same pixel.  Rows: plane row q's centre is clamp(y0 - R + q); window slot k serves rows q0 - 1 + k only
            // when no clamp intervenes, so every plane row takes its own three image rows below instead.
#pragma unroll
            for (int k = 0; k < SEG + 2; k++) va[k] = vb[k] = vc[k] = zero;
        }
        hv2f ax[3], ay[3];
        auto rowpass_v = [&](hv2f a_, hv2f b_, hv2f c_, int slot) {
            ax[slot] = __builtin_elementwise_fma(c_, rdx2, __builtin_elementwise_fma(b_, rdx1, __builtin_elementwise_fma(a_, rdx0, zero)));
            ay[slot] = __builtin_elementwise_fma(c_, rdy2, __builtin_elementwise_fma(b_, rdy1, __builtin_elementwise_fma(a_, rdy0, zero)));
        };
        auto finish = [&](int q, int s_top, int s_mid, int s_new) {
            const hv2f gxv = __builtin_elementwise_fma(ax[s_new], cdx2, __builtin_elementwise_fma(ax[s_mid], cdx1, __builtin_elementwise_fma(ax[s_top], cdx0, zero)));
            const hv2f gyv = __builtin_elementwise_fma(ay[s_new], cdy2, __builtin_elementwise_fma(ay[s_mid], cdy1, __builtin_elementwise_fma(ay[s_top], cdy0, zero)));
            *reinterpret_cast<hv2f *>(XX + q * PS + lx) = gxv * gxv;
            *reinterpret_cast<hv2f *>(XY + q * PS + lx) = gxv * gyv;
            *reinterpret_cast<hv2f *>(YY + q * PS + lx) = gyv * gyv;
            const int gy_ = y0 - R + q, gx_ = x0 - 4 + lx;
            if (gxo && q >= R && q < R + TH && lx >= 4 && lx < TW + 4 && gy_ < rows) {
                const size_t o = (size_t)gy_ * gstride + gx_;
                if (interior) {
                    *reinterpret_cast<hv2f *>(gxo + o) = gxv;
                    *reinterpret_cast<hv2f *>(gyo + o) = gyv;
                } else {
                    if (gx_ < cols) { gxo[o] = gxv.x; gyo[o] = gyv.x; }
                    if (gx_ + 1 < cols) { gxo[o + 1] = gxv.y; gyo[o + 1] = gyv.y; }
                }
            }
        };
        if (interior) {
            rowpass_v(va[0], vb[0], vc[0], 0);
            rowpass_v(va[1], vb[1], vc[1], 1);
#pragma unroll
            for (int j = 0; j < SEG; j++) {
                if (q0 + j < RH) {
                    rowpass_v(va[j + 2], vb[j + 2], vc[j + 2], (j + 2) % 3);
                    finish(q0 + j, j % 3, (j + 1) % 3, (j + 2) % 3);
                }
            }
        } else {
            const int cxa = clampi(x0 - 4 + lx, 0, cols - 1), cxb = clampi(x0 - 4 + lx + 1, 0, cols - 1);
            const int xa[3] = {reflect101(cxa - 1, cols), cxa, reflect101(cxa + 1, cols)};
            const int xb[3] = {reflect101(cxb - 1, cols), cxb, reflect101(cxb + 1, cols)};
            for (int j = 0; j < SEG; j++) {
                const int q = q0 + j;
                if (q >= RH) break;
                const int cy = clampi(y0 - R + q, 0, rows - 1);
                const int ys[3] = {reflect101(cy - 1, rows), cy, reflect101(cy + 1, rows)};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float *rp = img + (size_t)ys[k] * istride;
                    rowpass_v((hv2f){rp[xa[0]], rp[xb[0]]}, (hv2f){rp[xa[1]], rp[xb[1]]}, (hv2f){rp[xa[2]], rp[xb[2]]}, k);
                }
                finish(q, 0, 1, 2);
            }
        }
    }
    __syncthreads();
    harris_tile_accumulate<R, TH, CPU>(XX, XY, YY, hw, alpha, resp, rstride, rows, cols, x0, y0, vec_ok);
}

template <int R, bool CPU>
static void launch_harris_image(hipStream_t s, const float *img, int istride, int rows, int cols, const Taps &g, float alpha,
                                float *resp, int rstride, float *gx, float *gy, int gstride) {
    HarrisW<R> hw;
    for (int wy = 0; wy <= 2 * R; wy++)
        for (int wx = 0; wx <= 2 * R; wx++) hw.w[wy * (2 * R + 1) + wx] = g.k[wy] * g.k[wx];
    Taps d, m;
    sobel_taps(3, 1, &d);
    sobel_taps(3, 0, &m);
    Sobel3 t;
    for (int i = 0; i < 3; i++) {  // filters.hip, launch_sobel_fused with scale 1 (harris::getGradients)
        t.row_dx[i] = d.k[i];
        t.col_dx[i] = m.k[i];
        t.row_dy[i] = m.k[i];
        t.col_dy[i] = d.k[i];
    }
    const int vec_ok = ((reinterpret_cast<uintptr_t>(resp) | reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0 &&
                       (rstride & 3) == 0 && (gstride & 3) == 0;
    // the marching body stores gradient pairs (8 B) and loads the image by float4
    const int img_vec_ok = (reinterpret_cast<uintptr_t>(img) & 15) == 0 && (istride & 3) == 0 &&
                           ((reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(gy)) & 7) == 0 && (gstride & 1) == 0;
    if ((size_t)cdiv(cols, 64) * cdiv(rows, 32) >= 1024)
        harris_image_response_kernel<R, 32, CPU><<<dim3(cdiv(cols, 64), cdiv(rows, 32)), 512, 0, s>>>(
            img, istride, rows, cols, t, hw, alpha, resp, rstride, gx, gy, gstride, vec_ok, img_vec_ok);
    else
        harris_image_response_kernel<R, 16, CPU><<<dim3(cdiv(cols, 64), cdiv(rows, 16)), 256, 0, s>>>(
            img, istride, rows, cols, t, hw, alpha, resp, rstride, gx, gy, gstride, vec_ok, img_vec_ok);
}

// ---- a10: threshold + non-maximum suppression ----------------------------------------------
// One thread per pixel; only pixels with R >= threshold scan their (2d+1)^2 clamped window
// (strictly greater than every OTHER pixel, Harris.cpp:119-135).  flag marks kept maxima for
// the ordered compaction that follows.
__global__ __launch_bounds__(256) void harris_nms_kernel(const float *__restrict__ resp,
                                                          int rstride, int rows, int cols,
                                                          double threshold, int d,
                                                          float *__restrict__ corners, int cstride,
                                                          unsigned long long *__restrict__ rowmask, int tiles_x) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool in = x < cols && y < rows;
    bool keep = false;
    float v = 0.f;
    if (in) {
        v = resp[(size_t)y * rstride + x];
        keep = (double)v >= threshold;  // Harris.cpp:115 (double compare)
        if (keep) {
            for (int wy = -d; wy <= d && keep; wy++) {
                const int cy = clampi(y + wy, 0, rows - 1);
                const float *row = resp + (size_t)cy * rstride;
                for (int wx = -d; wx <= d; wx++) {
                    const int cx = clampi(x + wx, 0, cols - 1);
                    if (cy == y && cx == x) continue;
                    if (v <= row[cx]) {
                        keep = false;
                        break;
                    }
                }
            }
        }
        if (corners) corners[(size_t)y * cstride + x] = keep ? v : 0.f;
    }
    // a wave = 64 consecutive cells of one row: its ballot is that row segment's mask (compact.hpp)
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && y < rows) rowmask[(size_t)y * tiles_x + blockIdx.x] = m;
}

// LDS-tiled form for minDistance <= 16: "strictly greater than every other pixel of the clamped
// window" == "equals the window maximum and that maximum occurs once".  (max, multiplicity) is
// separable: a row pass over the staged tile, then a column pass -- 2(2d+1) steps per pixel instead
// of (2d+1)^2 divergent global reads.  Cells outside the image are staged as NaN, which neither
// wins a `>` nor matches a `==`, exactly like the clamped window never adding a new pixel; a NaN
// response inside the image is ignored by its neighbours for the same reason (`v <= NaN` is false
// in the scan above too).
// DT > 0: minDistance known at compile time (loops unroll, the LDS reads of a window are all in
// flight); DT == 0: any d <= 16 with rolled loops.
// (maximum, how often it occurs) of the four windows [j, j + 2D], j = 0..3, of 4 + 2D values; w = nullptr: every
// value counts once, else value k counts w[k] times (the column pass adds up the row pass's counts).  The windows
// share the values [3, 2D]: their maximum and its count are formed once; a window's own maximum is that one or one
// of its three other values, and when it is larger than the common maximum nothing of the common part equals it.
// fmaxf skips NaN and `==` is false for it, as in the plain scans; max and integer sums do not depend on the order.
template <int D>
__device__ __forceinline__ void nms_window4(const float (&v)[4 + 2 * D], const int *w, float (&m)[4], int (&n)[4]) {
    float mc = -INFINITY;
#pragma unroll
    for (int k = 3; k <= 2 * D; k++) mc = fmaxf(mc, v[k]);
    int nc = 0;
#pragma unroll
    for (int k = 3; k <= 2 * D; k++) nc += v[k] == mc ? (w ? w[k] : 1) : 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float mj = mc;
#pragma unroll
        for (int k = j; k <= j + 2 * D; k++)
            if (k < 3 || k > 2 * D) mj = fmaxf(mj, v[k]);
        int nj = mj == mc ? nc : 0;
#pragma unroll
        for (int k = j; k <= j + 2 * D; k++)
            if (k < 3 || k > 2 * D) nj += v[k] == mj ? (w ? w[k] : 1) : 0;
        m[j] = mj;
        n[j] = nj;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void harris_nms_tiled_kernel(const float *__restrict__ resp,
                                                                int rstride, int rows, int cols,
                                                                double threshold, int d_rt,
                                                                float *__restrict__ corners,
                                                                int cstride,
                                                                unsigned long long *__restrict__ rowmask, int tiles_x) {
    constexpr int TW = 64, TH = DT > 0 ? 32 : 16, DMAX = DT > 0 ? DT : 16;  // taller tiles re-fetch less halo
    const int d = DT > 0 ? DT : d_rt;
    __shared__ float T[(TH + 2 * DMAX) * (TW + 2 * DMAX + 1)];
    __shared__ __attribute__((aligned(16))) float RM[(TH + 2 * DMAX) * TW];
    // (counts as bytes -- a row window holds at most 2 d + 1 <= 33 equal cells: r06, 34 -> 26 KB of LDS for d = 5 = six
    // workgroups per CU instead of four, and one dword store for four cells' counts)
    __shared__ __attribute__((aligned(16))) unsigned char RN[(TH + 2 * DMAX) * TW];
    const int RW = TW + 2 * d, RH = TH + 2 * d, TS = RW | 1;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    // Staging: 128 lanes per region row (RW <= 96), two rows per pass of the workgroup, four
    // passes' loads in flight at once -- no division by the runtime pitch anywhere.
    {
        const int lx = threadIdx.x & 127, lr = threadIdx.x >> 7;
        const int xx = x0 - d + lx;
        const bool col_in = (unsigned)xx < (unsigned)cols;
        const int xc = clampi(xx, 0, cols - 1);
        if constexpr (DT > 0) {
            // every load of the tile in flight at once (r06: the kernel's time was the four dependent rounds of four loads)
            constexpr int NRB = (TH + 2 * DT + 7) / 8;
            float v[NRB][4];
#pragma unroll
            for (int b = 0; b < NRB; b++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                    v[b][k] = resp[(size_t)clampi(y0 - d + 8 * b + 2 * k + lr, 0, rows - 1) * rstride + xc];
#pragma unroll
            for (int b = 0; b < NRB; b++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int ly = 8 * b + 2 * k + lr, yy = y0 - d + ly;
                    if (ly < RH && lx < RW)
                        T[ly * TS + lx] = (col_in && (unsigned)yy < (unsigned)rows) ? v[b][k] : __builtin_nanf("");
                }
        } else
        for (int rb = 0; rb < RH; rb += 8) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                v[k] = resp[(size_t)clampi(y0 - d + rb + 2 * k + lr, 0, rows - 1) * rstride + xc];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ly = rb + 2 * k + lr, yy = y0 - d + ly;
                if (ly < RH && lx < RW)
                    T[ly * TS + lx] = (col_in && (unsigned)yy < (unsigned)rows) ? v[k] : __builtin_nanf("");
            }
        }
    }
    __syncthreads();
    if constexpr (DT > 0) {
        // Four adjacent windows at a time (nms_window4): 4 + 2d staged values serve four cells, and their common
        // part is scanned once -- 14 LDS reads and ~70 instructions per four cells at d = 5 instead of 44 and 132.
        // Row pass: job = (region row, group of four columns); lanes = 16 groups x 4 rows, conflict-free at the odd pitch.
        for (int i = threadIdx.x; i < RH * (TW / 4); i += 256) {
            const int r = i / (TW / 4), c4 = 4 * (i - r * (TW / 4));
            const float *tp = T + r * TS + c4;
            float v[4 + 2 * DT];
#pragma unroll
            for (int k = 0; k < 4 + 2 * DT; k++) v[k] = tp[k];
            float m[4];
            int n[4];
            nms_window4<DT>(v, nullptr, m, n);
            *reinterpret_cast<hv4f *>(RM + r * TW + c4) = (hv4f){m[0], m[1], m[2], m[3]};
            *reinterpret_cast<unsigned *>(RN + r * TW + c4) = (unsigned)n[0] | ((unsigned)n[1] << 8) | ((unsigned)n[2] << 16) | ((unsigned)n[3] << 24);
        }
        __syncthreads();
        // column pass: job = (column, group of four rows)
        const int c = threadIdx.x & 63, x = x0 + c;
        for (int rg = threadIdx.x >> 6; rg < TH / 4; rg += 4) {  // (wave-uniform: a wave is one 64-cell row segment)
            const int ry0 = 4 * rg;
            if (y0 + ry0 >= rows) break;
            float v[4 + 2 * DT];
            int w[4 + 2 * DT];
#pragma unroll
            for (int k = 0; k < 4 + 2 * DT; k++) {
                v[k] = RM[(ry0 + k) * TW + c];
                w[k] = RN[(ry0 + k) * TW + c];
            }
            float M[4];
            int N[4];
            nms_window4<DT>(v, w, M, N);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int y = y0 + ry0 + j;
                if (y < rows) {  // (wave-uniform)
                    const float pv = T[(ry0 + j + DT) * TS + c + DT];
                    const bool keep = x < cols && (double)pv >= threshold && pv == M[j] && N[j] == 1;
                    if (corners && x < cols) corners[(size_t)y * cstride + x] = keep ? pv : 0.f;
                    const unsigned long long m = __ballot(keep);  // the row segment's mask (compact.hpp)
                    if (c == 0) rowmask[(size_t)y * tiles_x + blockIdx.x] = m;
                }
            }
        }
        return;
    }
    // row pass: window maximum (fmaxf skips NaN), then how many cells equal it
    for (int i = threadIdx.x; i < RH * TW; i += 256) {
        const int r = i / TW, c = i - r * TW;
        const float *tp = T + r * TS + c;
        float m = -INFINITY;
        for (int k = 0; k <= 2 * d; k++) m = fmaxf(m, tp[k]);
        int n = 0;
        for (int k = 0; k <= 2 * d; k++) n += tp[k] == m ? 1 : 0;
        RM[i] = m;
        RN[i] = (unsigned char)n;
    }
    __syncthreads();
    const int c = threadIdx.x & 63, x = x0 + c;
    for (int ry = threadIdx.x >> 6; ry < TH; ry += 4) {  // (wave-uniform)
        const int y = y0 + ry;
        if (y >= rows) break;
        float M = -INFINITY;
        for (int k = 0; k <= 2 * d; k++) M = fmaxf(M, RM[(ry + k) * TW + c]);
        int N = 0;
        for (int k = 0; k <= 2 * d; k++) N += RM[(ry + k) * TW + c] == M ? RN[(ry + k) * TW + c] : 0;
        const float v = T[(ry + d) * TS + c + d];
        const bool keep = x < cols && (double)v >= threshold && v == M && N == 1;
        if (corners && x < cols) corners[(size_t)y * cstride + x] = keep ? v : 0.f;
        const unsigned long long m = __ballot(keep);
        if (c == 0) rowmask[(size_t)y * tiles_x + blockIdx.x] = m;
    }
}

// masks -> one flag byte per cell (the fallback when no one-launch state slot is free: the three-launch form below
// then runs on the flags)
__global__ __launch_bounds__(256) void masks_to_flags_kernel(const unsigned long long *__restrict__ masks, int rows, int cols,
                                                              int tiles_x, uint8_t *__restrict__ flag) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    flag[(size_t)y * cols + x] = (masks[(size_t)y * tiles_x + blockIdx.x] >> (threadIdx.x & 63)) & 1ull;
}

struct FlagPred {
    const uint8_t *flag;
    __device__ bool operator()(int64_t i) const { return flag[i] != 0; }
};
// mask word i = row i / tiles_x, columns 64 (i % tiles_x) ..: its set bit b is corner (y, x) -- Harris.cu:314-318
struct MaskYxEmit {
    int32_t *locs;
    int tiles_x;
    __device__ void operator()(int64_t pos, int64_t i, int bit) const {
        const int y = (int)(i / tiles_x);
        locs[2 * pos] = y;
        locs[2 * pos + 1] = 64 * (int)(i - (int64_t)y * tiles_x) + bit;
    }
};

// ---- a11 ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sift_angles_kernel(const float *__restrict__ gx,
                                                           const float *__restrict__ gy,
                                                           int gstride, int rows, int cols,
                                                           float *__restrict__ ang, int astride) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    ang[(size_t)y * astride + x] =
        atan2f(gy[(size_t)y * gstride + x], gx[(size_t)y * gstride + x]);  // Descriptors.cpp:22
}

__global__ void sift_keypoints_kernel(const float *__restrict__ gx, const float *__restrict__ gy,
                                      int gstride, const int32_t *__restrict__ locs, int64_t n,
                                      float size, float *__restrict__ kp) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int y = locs[2 * i], x = locs[2 * i + 1];
    const float PI = 3.1415921636f;  // sic, Descriptors.cpp:5
    const float a = atan2f(gy[(size_t)y * gstride + x], gx[(size_t)y * gstride + x]) * 180.f / PI;
    kp[4 * i] = (float)x;  // cv::KeyPoint(x = corner.second, y = corner.first, size, angle), :45
    kp[4 * i + 1] = (float)y;
    kp[4 * i + 2] = size;
    kp[4 * i + 3] = a;
}

}  // namespace micv

using namespace micv;

extern "C" {

int micv_harris_response_ex_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                                size_t gstride, int win, double sigma, float alpha, int flags, float *resp,
                                size_t rstride, micv_stream stream) {
    MICV_REQUIRE(ctx && gx && gy && resp, "micv_harris_response: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0, "micv_harris_response: bad size %dx%d", rows, cols);
    MICV_REQUIRE(win >= 1 && (win & 1) && win <= kMaxWin,
                 "micv_harris_response: window %d must be odd and <= %d", win, kMaxWin);
    MICV_REQUIRE(sigma > 0, "micv_harris_response: sigma must be > 0");
    MICV_REQUIRE(stride_ok(gstride, cols, 4) && stride_ok(rstride, cols, 4),
                 "micv_harris_response: bad stride");
    MICV_REQUIRE((flags & ~MICV_HARRIS_CPU) == 0, "micv_harris_response: unknown flags %#x", flags);
    MICV_HIP(hipSetDevice(ctx->device));
    Taps g;
    gaussian_taps(win, sigma, &g);  // cv::getGaussianKernel(win, sigma, CV_32F), Harris.cpp:61
    const int r = win / 2;
    const bool cpu = (flags & MICV_HARRIS_CPU) != 0;
    const bool force_generic = ctx->opt[MICV_OPT_HARRIS_GENERIC] != 0;
    if (!force_generic && r >= 1 && r <= 3) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int gs = (int)(gstride / 4), rs = (int)(rstride / 4);
        if (r == 1) (cpu ? launch_harris_tiled<1, true> : launch_harris_tiled<1, false>)(st, gx, gy, gs, rows, cols, g, alpha, resp, rs);
        if (r == 2) (cpu ? launch_harris_tiled<2, true> : launch_harris_tiled<2, false>)(st, gx, gy, gs, rows, cols, g, alpha, resp, rs);
        if (r == 3) (cpu ? launch_harris_tiled<3, true> : launch_harris_tiled<3, false>)(st, gx, gy, gs, rows, cols, g, alpha, resp, rs);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    const size_t lds = (size_t)(64 + 2 * r) * (16 + 2 * r) * 2 * sizeof(float);
    static thread_local int attr_dev = -1;
    if (attr_dev != ctx->device) {
        MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&harris_response_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&harris_response_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_dev = ctx->device;
    }
    const dim3 grid(cdiv(cols, 64), cdiv(rows, 16));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cpu)
        harris_response_kernel<true><<<grid, 256, lds, st>>>(gx, gy, (int)(gstride / 4), rows, cols, r, g, alpha, resp, (int)(rstride / 4));
    else
        harris_response_kernel<false><<<grid, 256, lds, st>>>(gx, gy, (int)(gstride / 4), rows, cols, r, g, alpha, resp, (int)(rstride / 4));
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_harris_response_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                             size_t gstride, int win, double sigma, float alpha, float *resp,
                             size_t rstride, micv_stream stream) {
    return micv_harris_response_ex_dev(ctx, gx, gy, rows, cols, gstride, win, sigma, alpha, 0, resp, rstride, stream);
}

}  // extern "C"

namespace micv {
static size_t harris_refine_scratch_bytes(int rows, int cols) {
    const int64_t n = (int64_t)rows * cols, nseg = (int64_t)rows * cdiv(cols, 64);
    return Carver::need(nseg, 8) + Carver::need(n, 1) + compact_scratch_bytes(n);
}

// harris::refineCorners on a response already on the device; `scratch` = harris_refine_scratch_bytes() of context
// scratch.  corners (the sparse map) may be null.
static int harris_refine_run(micv_ctx *ctx, hipStream_t s, const float *resp, int rows, int cols, size_t rstride, double threshold,
                             int min_distance, float *corners, size_t cstride, int32_t *locs_yx, int64_t cap, int64_t *count,
                             void *scratch) {
    const int64_t n = (int64_t)rows * cols;
    // The NMS kernels leave one 64-bit mask per (row, 64-column tile) -- a wave is exactly such a row segment, its
    // ballot the mask -- and the ordered list is a chained scan over those words (compact_masks_onepass_kernel):
    // 130 k words in 32 chunks at 4K, where the flag-byte form scanned 8.3 M bytes in three launches.
    const int tiles_x = cdiv(cols, 64);
    const int64_t nseg = (int64_t)rows * tiles_x;
    Carver c(scratch);
    unsigned long long *rowmask = c.take<unsigned long long>(nseg);
    const bool force_scan = ctx->opt[MICV_OPT_NMS_SCAN] != 0;
    if (min_distance <= 16 && !force_scan) {
#define MICV_NMS(DT)                                                                             \
    harris_nms_tiled_kernel<DT><<<dim3(tiles_x, cdiv(rows, (DT) > 0 ? 32 : 16)), 256, 0, s>>>(     \
        resp, (int)(rstride / 4), rows, cols,                                                     \
                                                     threshold, min_distance, corners,             \
                                                     (int)(cstride / 4), rowmask, tiles_x)
        switch (min_distance) {
            case 1: MICV_NMS(1); break;
            case 2: MICV_NMS(2); break;
            case 3: MICV_NMS(3); break;
            case 4: MICV_NMS(4); break;
            case 5: MICV_NMS(5); break;
            case 6: MICV_NMS(6); break;
            case 7: MICV_NMS(7); break;
            case 8: MICV_NMS(8); break;
            default: MICV_NMS(0); break;
        }
#undef MICV_NMS
    }
    else
        harris_nms_kernel<<<dim3(tiles_x, cdiv(rows, 4)), 256, 0, s>>>(
            resp, (int)(rstride / 4), rows, cols, threshold, min_distance, corners,
            (int)(cstride / 4), rowmask, tiles_x);
    MICV_LAUNCH_CHECK();
    const int nchunks = compact_masks_chunks(nseg);
    unsigned long long *status = nullptr;
    unsigned *counters = nullptr;
    if (ctx->opt[MICV_OPT_COMPACT_3PASS] <= 0 && ctx->compact_state(s, nchunks, &status, &counters) == MICV_OK) {
        launch_compact_masks(s, rowmask, MaskYxEmit{locs_yx, tiles_x}, nseg, status, counters, cap, count);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    // no state slot for this stream (or MICV_OPT_COMPACT_3PASS = 1): flag bytes and the count / scan / emit launches
    uint8_t *flag = c.take<uint8_t>(n);
    masks_to_flags_kernel<<<dim3(tiles_x, cdiv(rows, 4)), 256, 0, s>>>(rowmask, rows, cols, tiles_x, flag);
    MICV_LAUNCH_CHECK();
    // (y, x) written directly, Harris.cu:314-318 (Conv1Dto2D)
    return ordered_compact3(s, FlagPred{flag}, YxEmit{locs_yx, cols}, n, cap, count, c.base + c.off);
}
}  // namespace micv

extern "C" {

int micv_harris_refine_dev(micv_ctx *ctx, const float *resp, int rows, int cols, size_t rstride,
                           double threshold, int min_distance, float *corners, size_t cstride,
                           int32_t *locs_yx, int64_t cap, int64_t *count, micv_stream stream) {
    MICV_REQUIRE(ctx && resp && corners && count, "micv_harris_refine: null argument");
    MICV_REQUIRE(locs_yx || cap == 0, "micv_harris_refine: locs_yx is null");
    MICV_REQUIRE(rows > 0 && cols > 0 && (int64_t)rows * cols < ((int64_t)1 << 31),
                 "micv_harris_refine: bad size %dx%d", rows, cols);
    MICV_REQUIRE(min_distance >= 0 && cap >= 0, "micv_harris_refine: bad min_distance / cap");
    MICV_REQUIRE(stride_ok(rstride, cols, 4) && stride_ok(cstride, cols, 4),
                 "micv_harris_refine: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    void *scratch;
    MICV_TRY(ctx->reserve(harris_refine_scratch_bytes(rows, cols), &scratch));
    return harris_refine_run(ctx, static_cast<hipStream_t>(stream), resp, rows, cols, rstride, threshold, min_distance, corners, cstride,
                             locs_yx, cap, count, scratch);
}

int micv_harris_corners_dev(micv_ctx *ctx, const float *img, int rows, int cols, size_t stride, int sobel_ksize, int win,
                            double sigma, float alpha, int flags, double threshold, int min_distance, float *gx, float *gy,
                            size_t gstride, float *resp, size_t rstride, float *corners, size_t cstride, int32_t *locs_yx,
                            int64_t cap, int64_t *count, micv_stream stream) {
    MICV_REQUIRE(ctx && img && count, "micv_harris_corners: null argument");
    MICV_REQUIRE((gx == nullptr) == (gy == nullptr), "micv_harris_corners: give both gradient outputs or neither");
    MICV_REQUIRE(locs_yx || cap == 0, "micv_harris_corners: locs_yx is null");
    MICV_REQUIRE(rows > 0 && cols > 0 && (int64_t)rows * cols < ((int64_t)1 << 31), "micv_harris_corners: bad size %dx%d", rows, cols);
    MICV_REQUIRE(win >= 1 && (win & 1) && win <= kMaxWin, "micv_harris_corners: window %d must be odd and <= %d", win, kMaxWin);
    MICV_REQUIRE(sigma > 0 && min_distance >= 0 && cap >= 0, "micv_harris_corners: bad sigma / min_distance / cap");
    MICV_REQUIRE(stride_ok(stride, cols, 4) && (!gx || stride_ok(gstride, cols, 4)) && (!resp || stride_ok(rstride, cols, 4)) &&
                     (!corners || stride_ok(cstride, cols, 4)),
                 "micv_harris_corners: bad stride");
    MICV_REQUIRE((flags & ~MICV_HARRIS_CPU) == 0, "micv_harris_corners: unknown flags %#x", flags);
    Taps probe;
    MICV_REQUIRE(sobel_taps(sobel_ksize, 1, &probe) >= 0, "micv_harris_corners: Sobel size %d not supported", sobel_ksize);
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int r = win / 2;
    const bool cpu = (flags & MICV_HARRIS_CPU) != 0;
    const bool fused = sobel_ksize == 3 && r >= 1 && r <= 3 && !ctx->opt[MICV_OPT_HARRIS_GENERIC] && !ctx->opt[MICV_OPT_SOBEL_GENERIC];
    // one reservation for everything the chain needs: R when the caller does not keep it, the gradient planes and the
    // Sobel temporaries of the three-launch form, the list's scan state
    const size_t n = (size_t)rows * cols;
    const size_t need_r = resp ? 0 : Carver::need(n, 4);
    const size_t need_g = (fused || gx) ? 0 : 2 * Carver::need(n, 4);
    const size_t need_t = fused ? 0 : Carver::need(2 * n, 4);
    void *scratch;
    MICV_TRY(ctx->reserve(need_r + need_g + need_t + harris_refine_scratch_bytes(rows, cols), &scratch));
    Carver c(scratch);
    float *R_ = resp;
    size_t rs = rstride;
    if (!R_) {
        R_ = c.take<float>(n);
        rs = (size_t)cols * 4;
    }
    Taps g;
    gaussian_taps(win, sigma, &g);  // cv::getGaussianKernel(win, sigma, CV_32F), Harris.cpp:61
    if (fused) {
        const int is = (int)(stride / 4), gs = gx ? (int)(gstride / 4) : 0, rsi = (int)(rs / 4);
#define MICV_HI(RR)                                                                                               \
    (cpu ? launch_harris_image<RR, true> : launch_harris_image<RR, false>)(s, img, is, rows, cols, g, alpha, R_, rsi, gx, gy, gs)
        if (r == 1) MICV_HI(1);
        if (r == 2) MICV_HI(2);
        if (r == 3) MICV_HI(3);
#undef MICV_HI
        MICV_LAUNCH_CHECK();
    } else {
        // other Sobel sizes / windows: the three launches of the separate entry points, same scratch
        float *dgx = gx, *dgy = gy;
        size_t gsb = gstride;
        if (!dgx) {
            dgx = c.take<float>(n);
            dgy = c.take<float>(n);
            gsb = (size_t)cols * 4;
        }
        float *tmp = c.take<float>(2 * n);
        MICV_TRY(sobel_dev(s, img, rows, cols, (int)(stride / 4), sobel_ksize, 1.f, dgx, dgy, (int)(gsb / 4), tmp,
                           ctx->opt[MICV_OPT_SOBEL_GENERIC] != 0));
        // (micv_harris_response_ex_dev takes no scratch)
        MICV_TRY(micv_harris_response_ex_dev(ctx, dgx, dgy, rows, cols, gsb, win, sigma, alpha, flags, R_, rs, stream));
    }
    return harris_refine_run(ctx, s, R_, rows, cols, rs, threshold, min_distance, corners, cstride, locs_yx, cap, count, c.base + c.off);
}

int micv_sift_angles_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                         size_t gstride, float *angles, size_t astride, micv_stream stream) {
    MICV_REQUIRE(ctx && gx && gy && angles, "micv_sift_angles: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0, "micv_sift_angles: bad size %dx%d", rows, cols);
    MICV_REQUIRE(stride_ok(gstride, cols, 4) && stride_ok(astride, cols, 4),
                 "micv_sift_angles: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    sift_angles_kernel<<<dim3(cdiv(cols, 64), cdiv(rows, 4)), 256, 0,
                         static_cast<hipStream_t>(stream)>>>(gx, gy, (int)(gstride / 4), rows, cols,
                                                             angles, (int)(astride / 4));
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_sift_keypoints_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                            size_t gstride, const int32_t *locs_yx, int64_t n, float size,
                            float *kp_xysa, micv_stream stream) {
    MICV_REQUIRE(ctx && gx && gy, "micv_sift_keypoints: null argument");
    MICV_REQUIRE(n >= 0 && (n == 0 || (locs_yx && kp_xysa)), "micv_sift_keypoints: bad list");
    MICV_REQUIRE(rows > 0 && cols > 0 && stride_ok(gstride, cols, 4),
                 "micv_sift_keypoints: bad size / stride");
    MICV_HIP(hipSetDevice(ctx->device));
    if (n == 0) return MICV_OK;
    sift_keypoints_kernel<<<(unsigned)((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
        gx, gy, (int)(gstride / 4), locs_yx, n, size, kp_xysa);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

}  // extern "C"
