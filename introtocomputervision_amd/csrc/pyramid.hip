// pyramid.hip -- REDUCE / EXPAND, bilinear resize, grey conversion (ps5 pyramids, a5-a7).
#include "kernels.hpp"

namespace micv {

// pyr::pyrDown as the reference executes it (Pyramids.cu:31, launched on d_src :65-66):
// pure odd-index decimation.  Because every level is a decimation of the one above,
// level l is also a direct decimation of level 0:  L_l(y,x) = L_0(2^l y + 2^l - 1, ...).
// The multi-level kernel below uses that to build the whole pyramid in ONE launch.
__global__ __launch_bounds__(256) void pyr_down_kernel(const float *__restrict__ src, int sstride,
                                                        float *__restrict__ dst, int dstride,
                                                        int drows, int dcols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dcols || y >= drows) return;
    dst[(size_t)y * dstride + x] = src[(size_t)(2 * y + 1) * sstride + 2 * x + 1];
}

// The same with 16 bytes per lane (r04): a thread takes source row 2y+1, columns 8q .. 8q+7 as two float4 and
// stores the four odd ones as one float4 -- 2 KiB of contiguous source per wave and row instead of 64
// scattered dwords.  Needs 16-byte aligned rows on both sides; the last, partial quad of a row stores scalars.
typedef float pv4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void pyr_down_vec_kernel(const float *__restrict__ src, int sstride,
                                                            float *__restrict__ dst, int dstride,
                                                            int drows, int dcols, int scols) {
    const int q = blockIdx.x * 64 + (threadIdx.x & 63);  // quad of output columns 4q .. 4q+3
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (4 * q >= dcols || y >= drows) return;
    const float *sr = src + (size_t)(2 * y + 1) * sstride + 8 * q;
    float *dr = dst + (size_t)y * dstride + 4 * q;
    if (4 * q + 3 < dcols && 8 * q + 7 < scols) {
        const pv4f a = *reinterpret_cast<const pv4f *>(sr), b = *reinterpret_cast<const pv4f *>(sr + 4);
        *reinterpret_cast<pv4f *>(dr) = (pv4f){a.y, a.w, b.y, b.w};
    } else {
        for (int j = 0; j < 4 && 4 * q + j < dcols; j++) dr[j] = sr[2 * j + 1];
    }
}

struct PyrLevels {
    float *dst[16];
    int rows[16], cols[16];
    int tiles_before[17];  // prefix sum of 64x4-tiles per level (levels 1..n-1 and level 0 copy)
    int n;
    int chain;  // the lowest level built (>= 1): ITS tiles also emit every deeper level; 0 = every level has its own tiles
    // Row-sharded builds: level l is written for rows [row_lo[l], row_hi[l]) only, and the level-1
    // tiles start at tile row l1_tile0 (full builds: 0 .. rows, 0).
    int row_lo[16], row_hi[16];
    int l1_tile0;
};

// One block = 64x4 output pixels of some level; blockIdx.x indexes tiles over all levels,
// blockIdx.y = image in the batch.  Level 0 is a plain copy (the reference clones the input,
// Pyramids.cpp:9,18) and is only emitted when dst[0] != nullptr.
__global__ __launch_bounds__(256) void pyr_build_kernel(const float *__restrict__ src_a,
                                                         const float *__restrict__ src_b,
                                                         size_t img_elems, int sstride,
                                                         PyrLevels La, PyrLevels Lb) {
    // blockIdx.z selects the image set (prev / next pyramids are built by one launch)
    const float *__restrict__ src = blockIdx.z ? src_b : src_a;
    const PyrLevels &L = blockIdx.z ? Lb : La;
    int l = 0;
    const int bid = blockIdx.x;
    while (l + 1 < L.n && bid >= L.tiles_before[l + 1]) l++;
    if (L.dst[l] == nullptr) return;
    const int t = bid - L.tiles_before[l];
    const int tw = (L.cols[l] + 63) >> 6;
    const int tx = t % tw, ty = t / tw + (l == 1 ? L.l1_tile0 : 0);
    const int x = tx * 64 + (threadIdx.x & 63);
    const int y = ty * 4 + (threadIdx.x >> 6);
    if (x >= L.cols[l] || y >= L.rows[l]) return;
    const int sh = (1 << l) - 1;
    const float *s = src + blockIdx.y * img_elems;
    float *d = L.dst[l] + blockIdx.y * (size_t)L.rows[l] * L.cols[l];
    const float v = s[(size_t)((y << l) + sh) * sstride + (x << l) + sh];
    if (y >= L.row_lo[l] && y < L.row_hi[l]) d[(size_t)y * L.cols[l] + x] = v;
    // Every level is a decimation of the previous one at odd coordinates, so the tiles of the lowest level
    // built (level 1; level 2 when the level-1 build runs on a stream of its own, lk.hip) also emit the deeper
    // levels (which then have no tiles of their own): the source is read once.
    if (L.chain && l == L.chain) {
        int yy = y, xx = x, ll = l;
        while (ll + 1 < L.n && (yy & 1) && (xx & 1)) {
            yy >>= 1;
            xx >>= 1;
            ll++;
            if (L.dst[ll] && yy >= L.row_lo[ll] && yy < L.row_hi[ll] && xx < L.cols[ll])
                L.dst[ll][blockIdx.y * (size_t)L.rows[ll] * L.cols[ll] + (size_t)yy * L.cols[ll] + xx] = v;
        }
    }
}

// The same build for the common case (levels >= 2, 16-byte aligned rows, no level-0 copy, all rows):
// a thread loads FOUR source floats with one 16-byte load and keeps the two odd ones, so a wave
// reads 1 KiB of contiguous bytes per row instead of 64 scattered dwords (1080p x 8 pairs: 43 -> see
// DESIGN.md).  blockIdx.z = image set * batch + image.
__global__ __launch_bounds__(256) void pyr_build_vec_kernel(const float *__restrict__ src_a,
                                                             const float *__restrict__ src_b,
                                                             size_t img_elems, int sstride, int batch,
                                                             PyrLevels La, PyrLevels Lb) {
    const int set = blockIdx.z / batch, img = blockIdx.z - set * batch;
    const float *__restrict__ src = (set ? src_b : src_a) + img * img_elems;
    const PyrLevels &L = set ? Lb : La;
    const int x2 = blockIdx.x * 64 + (threadIdx.x & 63);  // pair of level-1 columns (2*x2, 2*x2 + 1)
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * x2 >= L.cols[1] || y >= L.rows[1]) return;
    // source row 2y+1, columns 4*x2 .. 4*x2+3: elements 1 and 3 are level-1 columns 2*x2, 2*x2+1
    const float4 v4 = *reinterpret_cast<const float4 *>(src + (size_t)(2 * y + 1) * sstride + 4 * x2);
    float *d1 = L.dst[1] + img * (size_t)L.rows[1] * L.cols[1] + (size_t)y * L.cols[1] + 2 * x2;
    if (2 * x2 + 1 < L.cols[1])
        *reinterpret_cast<float2 *>(d1) = make_float2(v4.y, v4.w);
    else
        d1[0] = v4.y;
    // deeper levels: level-1 pixel (y, x) with y and x odd is level-2 pixel (y/2, x/2), and so on;
    // of the two columns only the odd one (2*x2 + 1) can continue
    if ((y & 1) && 2 * x2 + 1 < L.cols[1]) {
        int yy = y, xx = 2 * x2 + 1, ll = 1;
        const float v = v4.w;
        while (ll + 1 < L.n && (yy & 1) && (xx & 1)) {
            yy >>= 1;
            xx >>= 1;
            ll++;
            if (L.dst[ll] && yy < L.rows[ll] && xx < L.cols[ll])
                L.dst[ll][img * (size_t)L.rows[ll] * L.cols[ll] + (size_t)yy * L.cols[ll] + xx] = v;
        }
    }
}

// pyr::pyrUp step 1+2a (Pyramids.cu:86-91 replicate, :126 row filter): the replicated image's
// rows 2yc and 2yc+1 are identical, so the row pass is evaluated once per COARSE row:
//   R(yc, x) = chain_k fmaf(src(yc, reflect101(x+k-2, 2w) / 2), g5[k], acc)
__global__ __launch_bounds__(256) void pyr_up_rows_kernel(const float *__restrict__ src,
                                                           int sstride, float *__restrict__ tmp,
                                                           int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int uc = 2 * cols;
    if (x >= uc || y >= rows) return;
    const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    const float *s = src + (size_t)y * sstride;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 5; k++) acc = fmaf(s[reflect101(x + k - 2, uc) >> 1], g5[k], acc);
    tmp[(size_t)y * uc + x] = acc;
}

// Step 2b: column pass over the replicated rows, then the caller's exact power-of-two scale
// (`du = 2 * du`, OpticalFlow.cpp:142,144).
__global__ __launch_bounds__(256) void pyr_up_cols_kernel(const float *__restrict__ tmp,
                                                           float *__restrict__ dst, int dstride,
                                                           int rows, int cols, float scale) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int uc = 2 * cols, ur = 2 * rows;
    if (x >= uc || y >= ur) return;
    const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 5; k++)
        acc = fmaf(tmp[(size_t)(reflect101(y + k - 2, ur) >> 1) * uc + x], g5[k], acc);
    dst[(size_t)y * dstride + x] = acc * scale;
}

// pyr::pyrUp in ONE launch (r04; the two kernels above stay for tiny images): a 128x32 tile of the fine
// image.  The (16 + 2) x (64 + 2) coarse block it depends on is staged in LDS (edge-clamped loads; BORDER_REFLECT_101
// on the FINE grid maps every out-of-range tap back inside the block, see the index arithmetic), the row pass
// runs once per coarse row and fine column into LDS -- the replicated rows 2m and 2m+1 are identical -- and the
// column pass takes four adjacent columns per thread from 16-byte LDS reads and stores 16 bytes.  No temporary in
// HBM: 1080p -> 2160p moves 8.3 + 33.2 MB instead of 8.3 + 2 x 16.6 + 16.6 + 33.2.  Same chains as the
// two-launch form (taps left -> right, then top -> bottom, fmaf from +0, then * scale): same bits.
constexpr int PU_W = 128, PU_H = 32, PU_CW = PU_W / 2 + 2, PU_CH = PU_H / 2 + 2, PU_CP = PU_CW + 2;
// blockIdx.z = 2 * image + field when a second field (src2 / dst2) is given, the image otherwise (launch_pyr_up_batch).
__global__ __launch_bounds__(256) void pyr_up_tiled_kernel(const float *__restrict__ src, int sstride,
                                                            float *__restrict__ dst, int dstride, int rows,
                                                            int cols, float scale, int vec_ok,
                                                            const float *__restrict__ src2 = nullptr, float *__restrict__ dst2 = nullptr,
                                                            size_t src_img = 0, size_t dst_img = 0) {
    {
        const unsigned z = blockIdx.z, img = src2 ? z >> 1 : z;
        if (src2 && (z & 1)) { src = src2; dst = dst2; }
        src += img * src_img;
        dst += img * dst_img;
    }
    __shared__ float Cs[PU_CH][PU_CP];                              // coarse rows r0-1 .., columns c0-1 ..
    __shared__ __attribute__((aligned(16))) float Rp[PU_CH][PU_W];  // row pass: coarse row x fine column
    const int tid = threadIdx.x;
    const int fx0 = blockIdx.x * PU_W, fy0 = blockIdx.y * PU_H;
    const int c0 = fx0 >> 1, r0 = fy0 >> 1, ur = 2 * rows, uc = 2 * cols;
    const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    for (int i = tid; i < PU_CH * PU_CW; i += 256) {
        const int ly = i / PU_CW, lx = i - ly * PU_CW;
        Cs[ly][lx] = src[(size_t)clampi(r0 - 1 + ly, 0, rows - 1) * sstride + clampi(c0 - 1 + lx, 0, cols - 1)];
    }
    __syncthreads();
    {
        const int t = tid & (PU_W - 1), x = fx0 + t;
        if (x < uc) {
            int ci[5];
#pragma unroll
            for (int k = 0; k < 5; k++) ci[k] = (reflect101(x + k - 2, uc) >> 1) - (c0 - 1);
            for (int i = tid >> 7; i < PU_CH; i += 2) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 5; k++) acc = fmaf(Cs[i][ci[k]], g5[k], acc);
                Rp[i][t] = acc;
            }
        }
    }
    __syncthreads();
    const int x4 = 4 * (tid & 31), x = fx0 + x4;
    if (x >= uc) return;
    constexpr int RPT = PU_H / 8;  // rows per thread: 8 row groups of 32 column quads
#pragma unroll
    for (int o = 0; o < RPT; o++) {
        const int y = fy0 + RPT * (tid >> 5) + o;
        if (y >= ur) break;
        pv4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int ri = (reflect101(y + k - 2, ur) >> 1) - (r0 - 1);
            const pv4f v = *reinterpret_cast<const pv4f *>(&Rp[ri][x4]);
            const pv4f gk = {g5[k], g5[k], g5[k], g5[k]};
            acc = __builtin_elementwise_fma(v, gk, acc);
        }
        acc = acc * (pv4f){scale, scale, scale, scale};
        float *d = dst + (size_t)y * dstride + x;
        if (vec_ok && x + 3 < uc) {
            __builtin_nontemporal_store(acc, reinterpret_cast<pv4f *>(d));  // written once, read by a later launch
        } else {
            for (int j = 0; j < 4 && x + j < uc; j++) d[j] = acc[j];
        }
    }
}

// cv::resize INTER_LINEAR for CV_32F: half-pixel centres, x taps zero-weighted at the edges,
// y taps clamped; horizontal pass per source row, then the vertical blend (unfused mul/add).
__global__ __launch_bounds__(256) void resize_linear_kernel(const float *__restrict__ src,
                                                             int srows, int scols, int sstride,
                                                             float *__restrict__ dst, int drows,
                                                             int dcols, int dstride, double scale_x,
                                                             double scale_y) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dcols || dy >= drows) return;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= scols - 1) { fx = 0.f; sx = scols - 1; }
    const float a0 = 1.f - fx, a1 = fx;
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    const float b0 = 1.f - fy, b1 = fy;
    const float *r0 = src + (size_t)clampi(sy, 0, srows - 1) * sstride;
    const float *r1 = src + (size_t)clampi(sy + 1, 0, srows - 1) * sstride;
    float h0, h1;
    if (sx + 1 >= scols) {
        h0 = r0[sx] * 1.f;
        h1 = r1[sx] * 1.f;
    } else {
        h0 = r0[sx] * a0 + r0[sx + 1] * a1;
        h1 = r1[sx] * a0 + r1[sx + 1] * a1;
    }
    dst[(size_t)dy * dstride + dx] = h0 * b0 + h1 * b1;
}

// The same with four adjacent outputs per thread and one 16-byte store (r04); the taps are scalar loads that the
// neighbouring lanes share through L1.  Same expressions, same bits.
__global__ __launch_bounds__(256) void resize_linear_vec_kernel(const float *__restrict__ src, int srows, int scols,
                                                                 int sstride, float *__restrict__ dst, int drows,
                                                                 int dcols, int dstride, double scale_x,
                                                                 double scale_y) {
    const int dx0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx0 >= dcols || dy >= drows) return;
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    const float b0 = 1.f - fy, b1 = fy;
    const float *r0 = src + (size_t)clampi(sy, 0, srows - 1) * sstride;
    const float *r1 = src + (size_t)clampi(sy + 1, 0, srows - 1) * sstride;
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int dx = dx0 + j < dcols ? dx0 + j : dcols - 1;
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= scols - 1) { fx = 0.f; sx = scols - 1; }
        const float a0 = 1.f - fx, a1 = fx;
        float h0, h1;
        if (sx + 1 >= scols) {
            h0 = r0[sx] * 1.f;
            h1 = r1[sx] * 1.f;
        } else {
            h0 = r0[sx] * a0 + r0[sx + 1] * a1;
            h1 = r1[sx] * a0 + r1[sx + 1] * a1;
        }
        out[j] = h0 * b0 + h1 * b1;
    }
    float *d = dst + (size_t)dy * dstride + dx0;
    if (dx0 + 3 < dcols) {
        *reinterpret_cast<pv4f *>(d) = (pv4f){out[0], out[1], out[2], out[3]};
    } else {
        for (int j = 0; j < 4 && dx0 + j < dcols; j++) d[j] = out[j];
    }
}

// OpticalFlow.cpp:139-151 for levels whose size is not twice the coarser one: du = 2*pyrUp(du),
// then cv::resize(du, du, level size).  One launch for the whole batch and both flow fields
// (blockIdx.z = 2*pair + field); every output evaluates the four expanded samples it blends
// from the coarse field directly (same fmaf chains as pyr_up_rows/cols_kernel, same blend as
// resize_linear_kernel), so no intermediate image is written.
__global__ __launch_bounds__(256) void flow_expand_resize_kernel(
    const float *__restrict__ src_u, const float *__restrict__ src_v, int fr, int fc,
    size_t src_pair, float *__restrict__ dst_u, float *__restrict__ dst_v, int drows, int dcols,
    size_t dst_pair, double scale_x, double scale_y) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dcols || dy >= drows) return;
    const int pair = blockIdx.z >> 1;
    const float *__restrict__ src = ((blockIdx.z & 1) ? src_v : src_u) + pair * src_pair;
    float *__restrict__ dst = ((blockIdx.z & 1) ? dst_v : dst_u) + pair * dst_pair;
    const int ur = 2 * fr, uc = 2 * fc;  // size of the expanded field
    const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    auto expanded = [&](int y, int x) -> float {  // 2 * pyrUp(src)(y, x)
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float *s = src + (size_t)(reflect101(y + k - 2, ur) >> 1) * fc;
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 5; j++) r = fmaf(s[reflect101(x + j - 2, uc) >> 1], g5[j], r);
            acc = fmaf(r, g5[k], acc);
        }
        return acc * 2.f;
    };
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= uc - 1) { fx = 0.f; sx = uc - 1; }
    const float a0 = 1.f - fx, a1 = fx;
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    const float b0 = 1.f - fy, b1 = fy;
    const int y0 = clampi(sy, 0, ur - 1), y1 = clampi(sy + 1, 0, ur - 1);
    float h0, h1;
    if (sx + 1 >= uc) {
        h0 = expanded(y0, sx) * 1.f;
        h1 = expanded(y1, sx) * 1.f;
    } else {
        h0 = expanded(y0, sx) * a0 + expanded(y0, sx + 1) * a1;
        h1 = expanded(y1, sx) * a0 + expanded(y1, sx + 1) * a1;
    }
    dst[(size_t)dy * dcols + dx] = h0 * b0 + h1 * b1;
}

// The same result with the expanded field shared through LDS: a 64x4 output tile touches at most
// TE_R x TE_C samples of 2*pyrUp(src); their row-pass values (one per COARSE row and column) and then
// the samples themselves are computed once per tile -- 10 FMAs per sample instead of 25 per use, 4 uses
// per output.  Same fmaf chains (row pass per coarse row, then the column taps), same blend: same bits.
// Used when the resize does not shrink by more than 2 (the tile's footprint is bounded).
constexpr int TE_R = 12, TE_C = 72, TE_CR = TE_R / 2 + 3;
__global__ __launch_bounds__(256) void flow_expand_resize_tiled_kernel(
    const float *__restrict__ src_u, const float *__restrict__ src_v, int fr, int fc,
    size_t src_pair, float *__restrict__ dst_u, float *__restrict__ dst_v, int drows, int dcols,
    size_t dst_pair, double scale_x, double scale_y) {
    __shared__ float Rp[TE_CR][TE_C];  // row pass of coarse row cr0 + i at expanded column ex0 + j
    __shared__ float Ex[TE_R][TE_C];   // 2 * pyrUp(src) at (ey0 + i, ex0 + j)
    const int tid = threadIdx.x;
    const int pair = blockIdx.z >> 1;
    const float *__restrict__ src = ((blockIdx.z & 1) ? src_v : src_u) + pair * src_pair;
    float *__restrict__ dst = ((blockIdx.z & 1) ? dst_v : dst_u) + pair * dst_pair;
    const int ur = 2 * fr, uc = 2 * fc;
    const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    // source coordinates of an output row / column, exactly as resize_linear_kernel computes them
    auto src_y = [&](int dy, float &fy) { fy = (float)((dy + 0.5) * scale_y - 0.5); const int sy = (int)floorf(fy); fy -= sy; return sy; };
    auto src_x = [&](int dx, float &fx) {
        fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= uc - 1) { fx = 0.f; sx = uc - 1; }
        return sx;
    };
    const int dx0 = blockIdx.x * 64, dy0 = blockIdx.y * 4;
    const int dx1 = min(dx0 + 63, dcols - 1), dy1 = min(dy0 + 3, drows - 1);
    float t;
    const int ey0 = clampi(src_y(dy0, t), 0, ur - 1), ey1 = clampi(src_y(dy1, t) + 1, 0, ur - 1);
    const int ex0 = src_x(dx0, t), ex1 = min(src_x(dx1, t) + 1, uc - 1);
    const int ner = ey1 - ey0 + 1, nec = ex1 - ex0 + 1;            // expanded samples the tile reads
    // coarse rows the column taps of those samples read: reflect101(y + k - 2, ur) >> 1
    int cr0 = fr, cr1 = -1;
    for (int y = ey0 - 2; y <= ey1 + 2; y++) {
        const int cr = reflect101(y, ur) >> 1;
        cr0 = cr < cr0 ? cr : cr0;
        cr1 = cr > cr1 ? cr : cr1;
    }
    const int ncr = cr1 - cr0 + 1;
    // ner <= 4 * scale_y + 2 <= 10, nec <= 64 * scale_x + 3 <= 71, ncr <= (ner + 4) / 2 + 1 <= 8: the
    // launcher admits only scales for which these bounds hold
    for (int i = tid; i < ncr * nec; i += 256) {
        const int ci = i / nec, j = i - ci * nec;
        const float *srow = src + (size_t)(cr0 + ci) * fc;
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < 5; q++) r = fmaf(srow[reflect101(ex0 + j + q - 2, uc) >> 1], g5[q], r);
        Rp[ci][j] = r;
    }
    __syncthreads();
    for (int i = tid; i < ner * nec; i += 256) {
        const int ei = i / nec, j = i - ei * nec;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 5; k++) acc = fmaf(Rp[(reflect101(ey0 + ei + k - 2, ur) >> 1) - cr0][j], g5[k], acc);
        Ex[ei][j] = acc * 2.f;
    }
    __syncthreads();
    const int dx = dx0 + (tid & 63), dy = dy0 + (tid >> 6);
    if (dx >= dcols || dy >= drows) return;
    float fx, fy;
    const int sx = src_x(dx, fx), sy = src_y(dy, fy);
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const int y0 = clampi(sy, 0, ur - 1) - ey0, y1 = clampi(sy + 1, 0, ur - 1) - ey0;
    float h0, h1;
    if (sx + 1 >= uc) {
        h0 = Ex[y0][sx - ex0] * 1.f;
        h1 = Ex[y1][sx - ex0] * 1.f;
    } else {
        h0 = Ex[y0][sx - ex0] * a0 + Ex[y0][sx + 1 - ex0] * a1;
        h1 = Ex[y1][sx - ex0] * a0 + Ex[y1][sx + 1 - ex0] * a1;
    }
    dst[(size_t)dy * dcols + dx] = h0 * b0 + h1 * b1;
}

int launch_flow_expand_resize(hipStream_t s, const float *src_u, const float *src_v, int fr, int fc,
                              size_t src_pair, float *dst_u, float *dst_v, int drows, int dcols,
                              size_t dst_pair, int batch) {
    const double scale_x = 1. / ((double)dcols / (2 * fc));
    const double scale_y = 1. / ((double)drows / (2 * fr));
    const dim3 grid(cdiv(dcols, 64), cdiv(drows, 4), 2 * batch);
    // tile footprint: 64 output columns span <= 64 * scale + 2 source columns, 4 rows <= 4 * scale + 2
    if (scale_x <= 1.05 && scale_y <= 2.0 && scale_x > 0 && scale_y > 0) {
        flow_expand_resize_tiled_kernel<<<grid, 256, 0, s>>>(src_u, src_v, fr, fc, src_pair, dst_u, dst_v, drows,
                                                             dcols, dst_pair, scale_x, scale_y);
    } else {
        flow_expand_resize_kernel<<<grid, 256, 0, s>>>(src_u, src_v, fr, fc, src_pair, dst_u, dst_v, drows, dcols,
                                                       dst_pair, scale_x, scale_y);
    }
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// cv::cvtColor(COLOR_RGB2GRAY) on CV_8UC3 (R2Y=4899, G2Y=9617, B2Y=1868, shift 14, round) then
// convertTo(CV_32F).
__global__ __launch_bounds__(256) void rgb8_to_gray_kernel(const uint8_t *__restrict__ rgb,
                                                            size_t sstride,
                                                            float *__restrict__ dst, int dstride,
                                                            int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const uint8_t *s = rgb + (size_t)y * sstride + 3 * x;
    const int g = (s[0] * 4899 + s[1] * 9617 + s[2] * 1868 + (1 << 13)) >> 14;
    dst[(size_t)y * dstride + x] = (float)g;
}

// The general form: 1, 3 or 4 interleaved channels of 8-bit or float samples -> grey f32.
//   8U : the fixed-point weights above (4-channel input ignores alpha), then convertTo(CV_32F);
//   32F: cv::cvtColor's float path, (c0*0.299f + c1*0.587f) + c2*0.114f, unfused;
//   1 channel: convertTo(CV_32F) only (exact for 8-bit, a copy for float).
template <typename T, int CN>
__global__ __launch_bounds__(256) void to_gray_kernel(const T *__restrict__ src, size_t sstride_bytes,
                                                       float *__restrict__ dst, int dstride, int rows,
                                                       int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const T *s = reinterpret_cast<const T *>(reinterpret_cast<const char *>(src) + (size_t)y * sstride_bytes) + CN * x;
    float g;
    if (CN == 1) {
        g = (float)s[0];
    } else if (sizeof(T) == 1) {
        g = (float)(((int)s[0] * 4899 + (int)s[1] * 9617 + (int)s[2] * 1868 + (1 << 13)) >> 14);
    } else {
        g = ((float)s[0] * 0.299f + (float)s[1] * 0.587f) + (float)s[2] * 0.114f;
    }
    dst[(size_t)y * dstride + x] = g;
}

// dst = a - b (dense a, b of pitch `cols`; dst strided).
__global__ __launch_bounds__(256) void sub_kernel(const float *__restrict__ a,
                                                   const float *__restrict__ b,
                                                   float *__restrict__ dst, int dstride, int rows,
                                                   int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    dst[(size_t)y * dstride + x] = a[(size_t)y * cols + x] - b[(size_t)y * cols + x];
}

int launch_pyr_down(hipStream_t s, const float *src, int rows, int cols, int sstride, float *dst,
                    int dstride) {
    const int dr = rows / 2, dc = cols / 2;
    if (dr == 0 || dc == 0) return MICV_OK;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 &&
                     (sstride & 3) == 0 && (dstride & 3) == 0 && dc >= 4;
    if (vec)
        pyr_down_vec_kernel<<<dim3(cdiv(dc, 256), cdiv(dr, 4)), 256, 0, s>>>(src, sstride, dst, dstride, dr, dc, cols);
    else
        pyr_down_kernel<<<dim3(cdiv(dc, 64), cdiv(dr, 4)), 256, 0, s>>>(src, sstride, dst, dstride, dr,
                                                                     dc);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int launch_pyr_up(hipStream_t s, const float *src, int rows, int cols, int sstride, float *dst,
                  int dstride, float scale, float *tmp) {
    // one LDS-tiled launch (no HBM temporary); 16-byte stores when the output rows allow them.  tmp == nullptr
    // always takes this form.
    if (2 * cols >= 8 || tmp == nullptr) {
        const int vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (dstride & 3) == 0;
        pyr_up_tiled_kernel<<<dim3(cdiv(2 * cols, PU_W), cdiv(2 * rows, PU_H)), 256, 0, s>>>(src, sstride, dst, dstride, rows,
                                                                                          cols, scale, vec_ok);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    pyr_up_rows_kernel<<<dim3(cdiv(2 * cols, 64), cdiv(rows, 4)), 256, 0, s>>>(src, sstride, tmp,
                                                                                rows, cols);
    MICV_LAUNCH_CHECK();
    pyr_up_cols_kernel<<<dim3(cdiv(2 * cols, 64), cdiv(2 * rows, 4)), 256, 0, s>>>(
        tmp, dst, dstride, rows, cols, scale);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// 2 * pyrUp of both flow fields of `batch` dense images in one launch (OpticalFlow.cpp:140-145 for a whole batch).
int launch_pyr_up_batch(hipStream_t s, const float *src_u, const float *src_v, int rows, int cols, size_t src_img,
                        float *dst_u, float *dst_v, size_t dst_img, float scale, int batch) {
    const int vec_ok = ((reinterpret_cast<uintptr_t>(dst_u) | reinterpret_cast<uintptr_t>(dst_v)) & 15) == 0 && ((2 * cols) & 3) == 0 &&
                       (dst_img & 3) == 0;
    pyr_up_tiled_kernel<<<dim3(cdiv(2 * cols, PU_W), cdiv(2 * rows, PU_H), 2 * batch), 256, 0, s>>>(
        src_u, cols, dst_u, 2 * cols, rows, cols, scale, vec_ok, src_v, dst_v, src_img, dst_img);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int launch_resize_linear(hipStream_t s, const float *src, int srows, int scols, int sstride,
                         float *dst, int drows, int dcols, int dstride) {
    const double scale_x = 1. / ((double)dcols / scols);
    const double scale_y = 1. / ((double)drows / srows);
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (dstride & 3) == 0 && dcols >= 8)
        resize_linear_vec_kernel<<<dim3(cdiv(dcols, 256), cdiv(drows, 4)), 256, 0, s>>>(
            src, srows, scols, sstride, dst, drows, dcols, dstride, scale_x, scale_y);
    else
        resize_linear_kernel<<<dim3(cdiv(dcols, 64), cdiv(drows, 4)), 256, 0, s>>>(
            src, srows, scols, sstride, dst, drows, dcols, dstride, scale_x, scale_y);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// Builds levels [first..n) of `batch` images in one launch. dst[l] holds the batch densely:
// image b of level l at dst[l] + b*rows_l*cols_l.
int launch_pyr_build2(hipStream_t s, const float *src_a, const float *src_b, size_t img_elems,
                      int sstride, int rows, int cols, int levels, float *const *dst_a,
                      float *const *dst_b, int batch, const int *row_lo, const int *row_hi) {
    PyrLevels L[2];
    int total = 0;
    // Row-restricted builds need the level-1 tiles (they emit the deeper levels): level-l row yy comes
    // from level-1 row 2^(l-1) (yy + 1) - 1.
    int l1_lo = 0, l1_hi = levels > 1 ? rows >> 1 : 0;
    const bool restricted = row_lo && row_hi && levels > 1 && dst_a[1] != nullptr;
    if (restricted) {
        l1_lo = rows;
        l1_hi = 0;
        for (int l = 1; l < levels; l++) {
            if (row_lo[l] >= row_hi[l]) continue;
            const int a = ((row_lo[l] + 1) << (l - 1)) - 1, b = (row_hi[l] << (l - 1));  // [a, b)
            l1_lo = a < l1_lo ? a : l1_lo;
            l1_hi = b > l1_hi ? b : l1_hi;
        }
        if (l1_lo < 0) l1_lo = 0;
        if (l1_hi > (rows >> 1)) l1_hi = rows >> 1;
        if (l1_lo >= l1_hi) return MICV_OK;
    }
    int chain_level = 0;
    for (int l = levels - 1; l >= 1; l--)
        if (dst_a[l]) chain_level = l;
    for (int k = 0; k < 2; k++) {
        float *const *dst = k ? dst_b : dst_a;
        L[k].n = levels;
        total = 0;
        L[k].l1_tile0 = restricted ? l1_lo / 4 : 0;
        for (int l = 0; l < levels; l++) {
            L[k].rows[l] = rows >> l;
            L[k].cols[l] = cols >> l;
            L[k].dst[l] = dst ? dst[l] : nullptr;
            L[k].tiles_before[l] = total;
            L[k].row_lo[l] = (row_lo && row_hi) ? row_lo[l] : 0;
            L[k].row_hi[l] = (row_lo && row_hi) ? row_hi[l] : L[k].rows[l];
            // both sets use the same tiling (a skipped level must be skipped in both); the tiles of the lowest
            // level present emit every deeper level
            L[k].chain = chain_level;
            if (dst_a[l] && !(chain_level && l > chain_level)) {
                if (restricted && l == 1)
                    total += cdiv(L[k].cols[l], 64) * (cdiv(l1_hi, 4) - l1_lo / 4);
                else
                    total += cdiv(L[k].cols[l], 64) * cdiv(L[k].rows[l], 4);
            }
        }
        L[k].tiles_before[levels] = total;
    }
    if (total == 0) return MICV_OK;
    const bool vec_ok = L[0].chain == 1 && !restricted && dst_a[0] == nullptr && (!dst_b || dst_b[0] == nullptr) &&
                        (sstride & 3) == 0 && (img_elems & 3) == 0 && (cols & 3) == 0 && (L[0].cols[1] & 1) == 0 &&
                        ((reinterpret_cast<uintptr_t>(src_a) | reinterpret_cast<uintptr_t>(src_b)) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(dst_a[1]) & 7) == 0 && (!dst_b || (reinterpret_cast<uintptr_t>(dst_b[1]) & 7) == 0) &&
                        (long)batch * (src_b ? 2 : 1) <= 65535;
    if (vec_ok) {
        pyr_build_vec_kernel<<<dim3(cdiv(L[0].cols[1], 128), cdiv(L[0].rows[1], 4), batch * (src_b ? 2 : 1)), 256, 0, s>>>(
            src_a, src_b, img_elems, sstride, batch, L[0], L[1]);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    pyr_build_kernel<<<dim3(total, batch, src_b ? 2 : 1), 256, 0, s>>>(src_a, src_b, img_elems, sstride,
                                                                      L[0], L[1]);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int launch_pyr_build(hipStream_t s, const float *src, size_t img_elems, int sstride, int rows,
                     int cols, int levels, float *const *dst, int batch) {
    return launch_pyr_build2(s, src, nullptr, img_elems, sstride, rows, cols, levels, dst, nullptr,
                             batch, nullptr, nullptr);
}

}  // namespace micv

using namespace micv;

extern "C" {

int micv_pyr_down_dev(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                      float *dst, size_t dstride, micv_stream stream) {
    MICV_REQUIRE(ctx && src && dst, "micv_pyr_down: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0, "micv_pyr_down: bad size %dx%d", rows, cols);
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(dstride, cols / 2, 4),
                 "micv_pyr_down: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    return launch_pyr_down(static_cast<hipStream_t>(stream), src, rows, cols, (int)(sstride / 4),
                           dst, (int)(dstride / 4));
}

int micv_pyr_up_dev(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                    float *dst, size_t dstride, micv_stream stream) {
    MICV_REQUIRE(ctx && src && dst, "micv_pyr_up: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && rows < (1 << 29) && cols < (1 << 29),
                 "micv_pyr_up: bad size %dx%d", rows, cols);
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(dstride, 2 * cols, 4),
                 "micv_pyr_up: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need((size_t)rows * cols * 2, 4), &scratch));
    return launch_pyr_up(static_cast<hipStream_t>(stream), src, rows, cols, (int)(sstride / 4),
                         dst, (int)(dstride / 4), 1.f, static_cast<float *>(scratch));
}

int micv_resize_linear_dev(micv_ctx *ctx, const float *src, int srows, int scols, size_t sstride,
                           float *dst, int drows, int dcols, size_t dstride,
                           micv_stream stream) {
    MICV_REQUIRE(ctx && src && dst, "micv_resize_linear: null argument");
    MICV_REQUIRE(srows > 0 && scols > 0 && drows > 0 && dcols > 0, "micv_resize_linear: bad size");
    MICV_REQUIRE(stride_ok(sstride, scols, 4) && stride_ok(dstride, dcols, 4),
                 "micv_resize_linear: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    return launch_resize_linear(static_cast<hipStream_t>(stream), src, srows, scols,
                                (int)(sstride / 4), dst, drows, dcols, (int)(dstride / 4));
}

int micv_gaussian_pyramid_dev(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                              int levels, float *const *dst_levels, micv_stream stream) {
    MICV_REQUIRE(ctx && src && dst_levels, "micv_gaussian_pyramid: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0, "micv_gaussian_pyramid: bad size %dx%d", rows, cols);
    MICV_REQUIRE(levels >= 1 && levels <= 16 && (rows >> (levels - 1)) > 0 &&
                     (cols >> (levels - 1)) > 0,
                 "micv_gaussian_pyramid: %d levels do not fit a %dx%d image", levels, rows, cols);
    MICV_REQUIRE(stride_ok(sstride, cols, 4), "micv_gaussian_pyramid: bad stride");
    for (int l = 0; l < levels; l++)
        MICV_REQUIRE(dst_levels[l] != nullptr, "micv_gaussian_pyramid: dst_levels[%d] is null", l);
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // level 0 is a copy of the input (Pyramids.cpp:9,18): a 2-D device copy; the levels below it then take the
    // 16-byte-per-lane build (13 -> ~6 us at 1080p; the one-pixel-per-thread kernel builds level 0 as well otherwise)
    if (levels > 1 && dst_levels[0] != src) {
        MICV_HIP(hipMemcpy2DAsync(dst_levels[0], (size_t)cols * 4, src, sstride, (size_t)cols * 4, rows, hipMemcpyDeviceToDevice, s));
        float *lv[16];
        for (int l = 0; l < levels; l++) lv[l] = dst_levels[l];
        lv[0] = nullptr;
        return launch_pyr_build(s, src, 0, (int)(sstride / 4), rows, cols, levels, lv, 1);
    }
    return launch_pyr_build(s, src, 0, (int)(sstride / 4), rows,
                            cols, levels, dst_levels, 1);
}

int micv_gaussian_pyramid_batch_dev(micv_ctx *ctx, const float *src, int batch, size_t image_stride,
                                    int rows, int cols, size_t sstride, int levels, float *const *dst_levels,
                                    const int *row_begin, const int *row_end, micv_stream stream) {
    MICV_REQUIRE(ctx && src && dst_levels, "micv_gaussian_pyramid_batch: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && batch >= 1 && batch <= 32767, "micv_gaussian_pyramid_batch: bad size");
    MICV_REQUIRE(levels >= 1 && levels <= 16 && (rows >> (levels - 1)) > 0 && (cols >> (levels - 1)) > 0,
                 "micv_gaussian_pyramid_batch: %d levels do not fit a %dx%d image", levels, rows, cols);
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && image_stride % 4 == 0 &&
                     (batch == 1 || image_stride >= sstride * (size_t)rows),
                 "micv_gaussian_pyramid_batch: bad stride");
    MICV_REQUIRE((row_begin == nullptr) == (row_end == nullptr), "micv_gaussian_pyramid_batch: give both row arrays or none");
    for (int l = 1; l < levels; l++)
        MICV_REQUIRE(dst_levels[l] != nullptr, "micv_gaussian_pyramid_batch: dst_levels[%d] is null", l);
    if (row_begin)
        for (int l = 0; l < levels; l++)
            MICV_REQUIRE(row_begin[l] >= 0 && row_begin[l] <= row_end[l] && row_end[l] <= (rows >> l),
                         "micv_gaussian_pyramid_batch: bad row range at level %d", l);
    MICV_HIP(hipSetDevice(ctx->device));
    return launch_pyr_build2(static_cast<hipStream_t>(stream), src, nullptr, image_stride / 4, (int)(sstride / 4),
                             rows, cols, levels, dst_levels, nullptr, batch, row_begin, row_end);
}

int micv_laplacian_pyramid_dev(micv_ctx *ctx, const float *src, int rows, int cols,
                               size_t sstride, int levels, float *const *dst_levels,
                               micv_stream stream) {
    MICV_REQUIRE(ctx && src && dst_levels, "micv_laplacian_pyramid: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0, "micv_laplacian_pyramid: bad size %dx%d", rows, cols);
    MICV_REQUIRE(levels >= 1 && levels <= 16 && (rows >> (levels - 1)) > 0 &&
                     (cols >> (levels - 1)) > 0,
                 "micv_laplacian_pyramid: %d levels do not fit a %dx%d image", levels, rows, cols);
    MICV_REQUIRE(stride_ok(sstride, cols, 4), "micv_laplacian_pyramid: bad stride");
    for (int l = 0; l < levels; l++)
        MICV_REQUIRE(dst_levels[l] != nullptr, "micv_laplacian_pyramid: dst_levels[%d] is null", l);
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scratch: the Gaussian levels, the expanded image, the resized image, pyrUp's row-pass buffer
    size_t need = 0;
    for (int l = 0; l < levels; l++) need += Carver::need((size_t)(rows >> l) * (cols >> l), 4);
    const size_t big = Carver::need((size_t)rows * cols, 4);
    need += 3 * big;
    void *scratch;
    MICV_TRY(ctx->reserve(need, &scratch));
    Carver c(scratch);
    float *G[16];
    for (int l = 0; l < levels; l++) G[l] = c.take<float>((size_t)(rows >> l) * (cols >> l));
    float *up = c.take<float>((size_t)rows * cols), *rs = c.take<float>((size_t)rows * cols);
    float *tmp = c.take<float>((size_t)rows * cols);
    MICV_TRY(launch_pyr_build(s, src, 0, (int)(sstride / 4), rows, cols, levels, G, 1));
    for (int i = 0; i + 1 < levels; i++) {
        const int r = rows >> i, cc = cols >> i, r1 = rows >> (i + 1), c1 = cols >> (i + 1);
        MICV_TRY(launch_pyr_up(s, G[i + 1], r1, c1, c1, up, 2 * c1, 1.f, tmp));  // Solution.cpp:191
        const float *next = up;
        if (2 * r1 < r || 2 * c1 < cc) {  // :194-196
            MICV_TRY(launch_resize_linear(s, up, 2 * r1, 2 * c1, 2 * c1, rs, r, cc, cc));
            next = rs;
        }
        sub_kernel<<<dim3(cdiv(cc, 64), cdiv(r, 4)), 256, 0, s>>>(G[i], next, dst_levels[i], cc, r, cc);
        MICV_LAUNCH_CHECK();
    }
    const int rl = rows >> (levels - 1), cl = cols >> (levels - 1);
    MICV_HIP(hipMemcpyAsync(dst_levels[levels - 1], G[levels - 1], (size_t)rl * cl * 4,
                            hipMemcpyDeviceToDevice, s));  // :199
    return MICV_OK;
}

int micv_rgb8_to_gray_f32_dev(micv_ctx *ctx, const uint8_t *rgb, int rows, int cols,
                              size_t sstride, float *dst, size_t dstride, micv_stream stream) {
    MICV_REQUIRE(ctx && rgb && dst, "micv_rgb8_to_gray_f32: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0, "micv_rgb8_to_gray_f32: bad size %dx%d", rows, cols);
    MICV_REQUIRE(sstride >= (size_t)cols * 3 && stride_ok(dstride, cols, 4),
                 "micv_rgb8_to_gray_f32: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    rgb8_to_gray_kernel<<<dim3(cdiv(cols, 64), cdiv(rows, 4)), 256, 0,
                          static_cast<hipStream_t>(stream)>>>(rgb, sstride, dst,
                                                              (int)(dstride / 4), rows, cols);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_to_gray_f32_dev(micv_ctx *ctx, const void *src, int rows, int cols, size_t sstride,
                         int channels, int depth, float *dst, size_t dstride, micv_stream stream) {
    MICV_REQUIRE(ctx && src && dst, "micv_to_gray_f32: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0, "micv_to_gray_f32: bad size %dx%d", rows, cols);
    MICV_REQUIRE(channels == 1 || channels == 3 || channels == 4,
                 "micv_to_gray_f32: %d channels (1, 3 or 4 as cv::cvtColor(COLOR_RGB2GRAY) takes them)", channels);
    MICV_REQUIRE(depth == MICV_DEPTH_8U || depth == MICV_DEPTH_32F,
                 "micv_to_gray_f32: depth %d (MICV_DEPTH_8U or MICV_DEPTH_32F)", depth);
    const size_t es = depth == MICV_DEPTH_8U ? 1 : 4;
    MICV_REQUIRE(sstride >= (size_t)cols * channels * es && sstride % es == 0 && stride_ok(dstride, cols, 4),
                 "micv_to_gray_f32: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    const dim3 grid(cdiv(cols, 64), cdiv(rows, 4));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ds = (int)(dstride / 4);
#define MICV_GRAY(T, CN) to_gray_kernel<T, CN><<<grid, 256, 0, s>>>(static_cast<const T *>(src), sstride, dst, ds, rows, cols)
    if (depth == MICV_DEPTH_8U) {
        if (channels == 1) MICV_GRAY(uint8_t, 1);
        if (channels == 3) MICV_GRAY(uint8_t, 3);
        if (channels == 4) MICV_GRAY(uint8_t, 4);
    } else {
        if (channels == 1) MICV_GRAY(float, 1);
        if (channels == 3) MICV_GRAY(float, 3);
        if (channels == 4) MICV_GRAY(float, 4);
    }
#undef MICV_GRAY
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

}  // extern "C"
