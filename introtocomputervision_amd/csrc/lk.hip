// lk.hip -- Lucas-Kanade: generic (any odd window <= 63) kernels, the pyramid driver and the
// C entry points for a1-a4.  The win = 15 metric path dispatches to lk_fused.hip.
#include <cstdlib>

#include "kernels.hpp"
#include "lk_device.hpp"
#include "lk_fused.hpp"

namespace micv {

// ---- generic single-level pieces (one thread per pixel) -------------------------------

// blockIdx.z of the generic kernels = the pair of a batch: element offsets of pair z's images, base flow and outputs
// (r05: one launch per step for the whole batch instead of one per pair -- window 43 on 8 x 1080p pairs was bound by the
// host issuing ~330 launches per call).  The product / sum planes of pair z are the five fields from 5 z on.
struct LkPairs {
    size_t prev = 0, next = 0, base = 0, out = 0;
};

// computeGradients x2 + OpticalFlow.cpp:62-70: the five product fields, planar in S.
__global__ __launch_bounds__(256) void lk_products_kernel(const float *__restrict__ prev,
                                                           int pstride,
                                                           const float *__restrict__ next,
                                                           int nstride, int rows, int cols,
                                                           float *__restrict__ S, size_t field, LkPairs pp) {
    prev += blockIdx.z * pp.prev; next += blockIdx.z * pp.next; S += blockIdx.z * 5 * field;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float s1 = 1.f / 9.f, s2 = 2.f * s1;  // OpticalFlow.cpp:19, ky = [1,2,1]*scale
    int ry[3], rx[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        ry[j] = reflect101(y + j - 1, rows);
        rx[j] = reflect101(x + j - 1, cols);
    }
    float P[3][3], N[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            P[j][i] = prev[(size_t)ry[j] * pstride + rx[i]];
            N[j][i] = next[(size_t)ry[j] * nstride + rx[i]];
        }
    float pgx, pgy, ngx, ngy;
    sobel3(P, s1, s2, pgx, pgy);
    sobel3(N, s1, s2, ngx, ngy);
    const float ix = avg2(ngx, pgx), iy = avg2(ngy, pgy), it = N[1][1] - P[1][1];
    const size_t i = (size_t)y * cols + x;
    S[i] = ix * ix;
    S[field + i] = ix * iy;
    S[2 * field + i] = iy * iy;
    S[3 * field + i] = ix * it;
    S[4 * field + i] = iy * it;
}

// OpticalFlow.cpp:85-103 (+ :161-162 when base != nullptr).
__global__ __launch_bounds__(256) void lk_solve_kernel(const float *__restrict__ S, size_t field,
                                                        int rows, int cols,
                                                        const float *__restrict__ base_u,
                                                        const float *__restrict__ base_v,
                                                        int bstride, float *__restrict__ u,
                                                        float *__restrict__ v, int ostride, LkPairs pp) {
    S += blockIdx.z * 5 * field; u += blockIdx.z * pp.out; v += blockIdx.z * pp.out;
    if (base_u) { base_u += blockIdx.z * pp.base; base_v += blockIdx.z * pp.base; }
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const size_t i = (size_t)y * cols + x;
    float uu, vv;
    lk_solve(S[i], S[field + i], S[2 * field + i], S[3 * field + i], S[4 * field + i], uu, vv);
    if (base_u) {
        uu = base_u[(size_t)y * bstride + x] + uu;
        vv = base_v[(size_t)y * bstride + x] + vv;
    }
    u[(size_t)y * ostride + x] = uu;
    v[(size_t)y * ostride + x] = vv;
}

// ---- generic level in two launches (any odd window 5 .. 63; config/ps5.yaml runs 43) ----------------------
// The four-launch form above moves the five product fields through HBM three times (write, row pass, column
// pass) and reads them a fourth time to solve.  Here the row pass starts from the images and the column pass
// ends in the solve: one intermediate (the five row-filtered fields), 57 + 57 MB instead of 58 + 83 + 83 + 58
// at 1080p.  Every value is produced by the same expressions in the same order (sobel3 / avg2 / the products
// of lk_products_kernel, the fmaf chains of filter_rows / filter_cols from +0, lk_solve): identical bits.
//
// Pass A: 256 x 8 outputs per workgroup.  Ix, Iy, It of the tile's cells (+ n/2 columns either side, columns
// BORDER_REFLECT_101 as the row filter would address the product fields) go to LDS, a thread producing the 8
// cells of one column from a 10 x 3 neighbourhood of each image held in registers (7.5 loads per cell instead
// of 18).  The row pass is filter_rows_lds_kernel's: four adjacent outputs per thread, taps four at a time, one
// ds_read_b128 per plane and chunk; the five products of the four new cells are formed from them in registers.
__global__ __launch_bounds__(256) void lk_products_rows_kernel(const float *__restrict__ prev, int pstride,
                                                                const float *__restrict__ next, int nstride, int rows,
                                                                int cols, float *__restrict__ T, size_t field, Taps t, LkPairs pp) {
    prev += blockIdx.z * pp.prev; next += blockIdx.z * pp.next; T += blockIdx.z * 5 * field;
    constexpr int TW = 256, TR = 8;
    extern __shared__ float lkg_lds[];
    const int a = t.n / 2, rw = TW + t.n - 1, pw = ((rw + 3) & ~3) + 4;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TR;
    float *LX = lkg_lds, *LY = LX + TR * pw, *LT = LY + TR * pw;
    const float s1 = 1.f / 9.f, s2 = 2.f * s1;  // OpticalFlow.cpp:19
    int ry[TR + 2];
#pragma unroll
    for (int j = 0; j < TR + 2; j++) ry[j] = reflect101(y0 - 1 + j, rows);
#pragma unroll 1
    for (int c = threadIdx.x; c < rw; c += 256) {
        const int xx = reflect101(x0 - a + c, cols);
        const int rx0 = reflect101(xx - 1, cols), rx2 = reflect101(xx + 1, cols);
        float P[TR + 2][3], N[TR + 2][3];
#pragma unroll
        for (int j = 0; j < TR + 2; j++) {
            const float *pr = prev + (size_t)ry[j] * pstride, *nr = next + (size_t)ry[j] * nstride;
            P[j][0] = pr[rx0]; P[j][1] = pr[xx]; P[j][2] = pr[rx2];
            N[j][0] = nr[rx0]; N[j][1] = nr[xx]; N[j][2] = nr[rx2];
        }
#pragma unroll
        for (int k = 0; k < TR; k++) {
            float pgx, pgy, ngx, ngy;
            sobel3(&P[k], s1, s2, pgx, pgy);
            sobel3(&N[k], s1, s2, ngx, ngy);
            LX[k * pw + c] = avg2(ngx, pgx);
            LY[k * pw + c] = avg2(ngy, pgy);
            LT[k * pw + c] = N[k + 1][1] - P[k + 1][1];
        }
    }
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int q = threadIdx.x & 63;
#pragma unroll 1
    for (int r = threadIdx.x >> 6; r < TR; r += 4) {
        const f4 *px = reinterpret_cast<const f4 *>(LX + r * pw) + q;
        const f4 *py = reinterpret_cast<const f4 *>(LY + r * pw) + q;
        const f4 *pt = reinterpret_cast<const f4 *>(LT + r * pw) + q;
        float acc[5][4];
#pragma unroll
        for (int f = 0; f < 5; f++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[f][j] = 0.f;
        // w[f][0..3]: products of the four cells the chunk starts at, w[f][4..7]: of the next four
        float w[5][8];
        auto products = [&](const f4 ix, const f4 iy, const f4 it, int o) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                w[0][o + e] = ix[e] * ix[e];
                w[1][o + e] = ix[e] * iy[e];
                w[2][o + e] = iy[e] * iy[e];
                w[3][o + e] = ix[e] * it[e];
                w[4][o + e] = iy[e] * it[e];
            }
        };
        products(px[0], py[0], pt[0], 0);
        int c = 0;
        for (; c + 4 <= t.n; c += 4) {
            const int i = (c >> 2) + 1;
            products(px[i], py[i], pt[i], 4);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const float tap = t.k[c + kk];
#pragma unroll
                for (int f = 0; f < 5; f++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[f][j] = fmaf(w[f][kk + j], tap, acc[f][j]);
            }
#pragma unroll
            for (int f = 0; f < 5; f++)
#pragma unroll
                for (int e = 0; e < 4; e++) w[f][e] = w[f][4 + e];
        }
        if (c < t.n) {  // 1-3 left-over taps
            const int i = (c >> 2) + 1;
            products(px[i], py[i], pt[i], 4);
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                if (c + kk < t.n) {
                    const float tap = t.k[c + kk];
#pragma unroll
                    for (int f = 0; f < 5; f++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[f][j] = fmaf(w[f][kk + j], tap, acc[f][j]);
                }
            }
        }
        if (y0 + r < rows) {
            const int x = x0 + 4 * q;
#pragma unroll
            for (int f = 0; f < 5; f++) {
                float *o = T + f * field + (size_t)(y0 + r) * cols + x;
                if (x + 3 < cols && (cols & 3) == 0) {
                    *reinterpret_cast<f4 *>(o) = (f4){acc[f][0], acc[f][1], acc[f][2], acc[f][3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (x + j < cols) o[j] = acc[f][j];
                }
            }
        }
    }
}

// Pass B: 64 x 32 outputs per workgroup, eight vertically adjacent outputs per thread (filter_cols_lds_kernel's
// job), the five fields one after the other through one LDS buffer -- the next field's rows are in flight while
// a field is summed -- then lk_solve (+ base flow) on the 40 sums in registers: u, v are the only stores.
__global__ __launch_bounds__(256) void lk_cols_solve_kernel(const float *__restrict__ T, size_t field, int rows, int cols,
                                                             Taps t, const float *__restrict__ base_u,
                                                             const float *__restrict__ base_v, int bstride,
                                                             float *__restrict__ u, float *__restrict__ v, int ostride, LkPairs pp) {
    T += blockIdx.z * 5 * field; u += blockIdx.z * pp.out; v += blockIdx.z * pp.out;
    if (base_u) { base_u += blockIdx.z * pp.base; base_v += blockIdx.z * pp.base; }
    constexpr int TW = 64, TH = 32, RP = TH / 4, NB = (32 + 62 + 3) / 4;  // n <= 63
    extern __shared__ float lkg_lds[];
    const int a = t.n / 2, ph = TH + t.n - 1;  // staged rows
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int c = threadIdx.x & 63, x = x0 + c < cols ? x0 + c : cols - 1;
    int roff[NB];  // (a field is below 2^31 elements: the callers' size checks)
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int r = 4 * k + (threadIdx.x >> 6);
        roff[k] = reflect101(y0 - a + (r < ph ? r : ph - 1), rows) * cols + x;
    }
    float st[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) st[k] = T[roff[k]];
    const int rb = (threadIdx.x >> 6) * RP;
    const float *lp = lkg_lds + rb * TW + c;
    float acc[5][RP];
#pragma unroll
    for (int f = 0; f < 5; f++) {
        if (f > 0) __syncthreads();  // field f - 1 has been summed by every wave
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int r = 4 * k + (threadIdx.x >> 6);
            if (r < ph) lkg_lds[r * TW + c] = st[k];
        }
        __syncthreads();
        if (f < 4) {
#pragma unroll
            for (int k = 0; k < NB; k++) st[k] = T[(f + 1) * field + roff[k]];
        }
#pragma unroll
        for (int j = 0; j < RP; j++) acc[f][j] = 0.f;
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
                for (int j = 0; j < RP; j++) acc[f][j] = fmaf(w[kk + j], tap, acc[f][j]);
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
                    for (int j = 0; j < RP; j++) acc[f][j] = fmaf(w[kk + j], tap, acc[f][j]);
                }
            }
        }
    }
    if (x0 + c >= cols) return;
#pragma unroll
    for (int j = 0; j < RP; j++) {
        const int y = y0 + rb + j;
        if (y < rows) {
            float uu, vv;
            lk_solve(acc[0][j], acc[1][j], acc[2][j], acc[3][j], acc[4][j], uu, vv);
            if (base_u) {
                uu = base_u[(size_t)y * bstride + x] + uu;
                vv = base_v[(size_t)y * bstride + x] + vv;
            }
            u[(size_t)y * ostride + x] = uu;
            v[(size_t)y * ostride + x] = vv;
        }
    }
}

// The same two passes with the tap count a template argument (instantiated for config/ps5.yaml's 43): fully
// unrolled, taps in SGPRs, and two outputs per v_pk_fma_f32.  Outputs (j, j + 1) share every tap and read the
// window values (k + j, k + j + 1) at tap k: the pair of values starting at an EVEN index is what a packed
// multiply (pass A) or a two-row ds_read2st64_b32 (pass B) delivers, the pair starting at an ODD index is one
// v_pk_mov_b32 (A) or one more two-row read (B) away.  Each half is still its output's own fmaf chain from +0.
typedef float lk_v2f __attribute__((ext_vector_type(2)));

template <int N>
__global__ __launch_bounds__(256) void lk_products_rows_pk_kernel(const float *__restrict__ prev, int pstride,
                                                                   const float *__restrict__ next, int nstride, int rows,
                                                                   int cols, float *__restrict__ T, size_t field, Taps t, LkPairs pp) {
    prev += blockIdx.z * pp.prev; next += blockIdx.z * pp.next; T += blockIdx.z * 5 * field;
    constexpr int TW = 256, TR = 8, A = N / 2, RW = TW + N - 1, PW = ((RW + 3) & ~3) + 4;
    constexpr int NV4 = (N + 3 + 3) / 4;  // float4 of a thread's window: values 0 .. N + 2
    static_assert(4 * 63 + 4 * NV4 <= PW, "window reads stay inside a staged row");
    extern __shared__ float lkg_lds[];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TR;
    float *LX = lkg_lds, *LY = LX + TR * PW, *LT = LY + TR * PW;
    const float s1 = 1.f / 9.f, s2 = 2.f * s1;  // OpticalFlow.cpp:19
    int ry[TR + 2];
#pragma unroll
    for (int j = 0; j < TR + 2; j++) ry[j] = reflect101(y0 - 1 + j, rows);
#pragma unroll 1
    for (int c = threadIdx.x; c < RW; c += 256) {
        const int xx = reflect101(x0 - A + c, cols);
        const int rx0 = reflect101(xx - 1, cols), rx2 = reflect101(xx + 1, cols);
        float P[TR + 2][3], Nx[TR + 2][3];
#pragma unroll
        for (int j = 0; j < TR + 2; j++) {
            const float *pr = prev + (size_t)ry[j] * pstride, *nr = next + (size_t)ry[j] * nstride;
            P[j][0] = pr[rx0]; P[j][1] = pr[xx]; P[j][2] = pr[rx2];
            Nx[j][0] = nr[rx0]; Nx[j][1] = nr[xx]; Nx[j][2] = nr[rx2];
        }
#pragma unroll
        for (int k = 0; k < TR; k++) {
            float pgx, pgy, ngx, ngy;
            sobel3(&P[k], s1, s2, pgx, pgy);
            sobel3(&Nx[k], s1, s2, ngx, ngy);
            LX[k * PW + c] = avg2(ngx, pgx);
            LY[k * PW + c] = avg2(ngy, pgy);
            LT[k * PW + c] = Nx[k + 1][1] - P[k + 1][1];
        }
    }
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int q = threadIdx.x & 63;
#pragma unroll 1
    for (int r = threadIdx.x >> 6; r < TR; r += 4) {
        const f4 *px = reinterpret_cast<const f4 *>(LX + r * PW) + q;
        const f4 *py = reinterpret_cast<const f4 *>(LY + r * PW) + q;
        const f4 *pt = reinterpret_cast<const f4 *>(LT + r * PW) + q;
        lk_v2f acc[5][2];
#pragma unroll
        for (int f = 0; f < 5; f++) acc[f][0] = acc[f][1] = (lk_v2f){0.f, 0.f};
        // E[f][m] = products of window values (2m, 2m + 1), O[f][m] = of (2m + 1, 2m + 2)
        lk_v2f E[5][2 * NV4], O[5][2 * NV4];
#pragma unroll
        for (int i = 0; i < NV4; i++) {
            const f4 ix = px[i], iy = py[i], it = pt[i];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const lk_v2f x2 = h ? (lk_v2f){ix.z, ix.w} : (lk_v2f){ix.x, ix.y};
                const lk_v2f y2 = h ? (lk_v2f){iy.z, iy.w} : (lk_v2f){iy.x, iy.y};
                const lk_v2f t2 = h ? (lk_v2f){it.z, it.w} : (lk_v2f){it.x, it.y};
                E[0][2 * i + h] = x2 * x2;
                E[1][2 * i + h] = x2 * y2;
                E[2][2 * i + h] = y2 * y2;
                E[3][2 * i + h] = x2 * t2;
                E[4][2 * i + h] = y2 * t2;
            }
#pragma unroll
            for (int f = 0; f < 5; f++) {
                if (i > 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(O[f][2 * i - 1]) : "v"(E[f][2 * i - 1]), "v"(E[f][2 * i]));
                asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(O[f][2 * i]) : "v"(E[f][2 * i]), "v"(E[f][2 * i + 1]));
            }
            // the taps whose last window value (k + 3) arrived with this float4
#pragma unroll
            for (int k = 4 * i - 3; k <= 4 * i; k++) {
                if (k >= 0 && k < N) {
                    const float tap = t.k[k];
                    const lk_v2f tap2 = (lk_v2f){tap, tap};
#pragma unroll
                    for (int f = 0; f < 5; f++)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int cell = k + 2 * h;
                            const lk_v2f d = (cell & 1) ? O[f][cell >> 1] : E[f][cell >> 1];
                            acc[f][h] = __builtin_elementwise_fma(d, tap2, acc[f][h]);
                        }
                }
            }
            // pin the accumulators: this float4's FMAs finish before the next one's reads are issued (the compiler
            // otherwise hoists every read of the unrolled window and runs out of registers)
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                              "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[4][0]), "+v"(acc[4][1])
                         :: "memory");
        }
        if (y0 + r < rows) {
            const int x = x0 + 4 * q;
#pragma unroll
            for (int f = 0; f < 5; f++) {
                float *o = T + f * field + (size_t)(y0 + r) * cols + x;
                const float out[4] = {acc[f][0].x, acc[f][0].y, acc[f][1].x, acc[f][1].y};
                if (x + 3 < cols && (cols & 3) == 0) {
                    *reinterpret_cast<f4 *>(o) = (f4){out[0], out[1], out[2], out[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (x + j < cols) o[j] = out[j];
                }
            }
        }
    }
}

// Pass B, packed: a thread owns two adjacent columns x 8 rows (64 x 32 outputs per wave-row of the 128 x 32
// tile), so a pair is simply the 8 staged bytes of its two columns -- one ds_read_b64 per window row, no pair
// assembly -- and every window row is read once into a register window that slides with the taps.  Tiles that
// lie inside the image (16-byte aligned rows) stage by LDS-DMA, field f + 1 into the second buffer while field f
// is summed: no staging registers, no ds_write.  Edge tiles load element by element, columns clamped.
template <int N, int TW = 128>
__global__ __launch_bounds__(256) void lk_cols_solve_pk_kernel(const float *__restrict__ T, size_t field, int rows, int cols,
                                                                Taps t, const float *__restrict__ base_u,
                                                                const float *__restrict__ base_v, int bstride,
                                                                float *__restrict__ u, float *__restrict__ v, int ostride, LkPairs pp) {
    T += blockIdx.z * 5 * field; u += blockIdx.z * pp.out; v += blockIdx.z * pp.out;
    if (base_u) { base_u += blockIdx.z * pp.base; base_v += blockIdx.z * pp.base; }
    // TW = 128: a wave is 64 column pairs x 8 rows, the tile 128 x 32.  TW = 64 (r05): a wave is 32 column pairs x two groups
    // of 8 rows, the tile 64 x 64 -- the column pass stages TH + N - 1 rows of every field for TH outputs, 2.3 x the planes'
    // bytes at 32 rows and window 43 and 1.66 x at 64, in the same LDS (two workgroups per CU either way).
    constexpr int LW = TW / 2, RG = 64 / LW;  // lanes per row of pairs, row groups per wave
    constexpr int TH = 32 * RG, RP = 8, A = N / 2, PH0 = TH + N - 1, V4 = TW / 4;
    constexpr int PH = PH0 + ((64 - (PH0 * V4) % 64) % 64) / V4;  // staged rows: whole waves of staging slots (the extra rows are never read)
    constexpr int NB = (PH * V4 + 255) / 256;
    static_assert((PH * V4) % 64 == 0 && ((64 - (PH0 * V4) % 64) % 64) % V4 == 0, "whole waves of staging slots");
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_cvoid;
    extern __shared__ float lkg_lds[];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int lane = threadIdx.x & 63, cl = lane % LW, rb = ((threadIdx.x >> 6) * RG + lane / LW) * RP;
    const bool dma = (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(T) & 15) == 0 && x0 + TW <= cols;  // uniform
    // staging slot i = float4 (i % V4) of staged row i / V4; its first element's offset inside a field (a field is
    // below 2^31 elements: the callers' size checks)
    int roff[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int i = threadIdx.x + 256 * k < PH * V4 ? threadIdx.x + 256 * k : PH * V4 - 1;
        const int ly = i / V4, m = i - ly * V4;
        roff[k] = reflect101(y0 - A + ly, rows) * cols + x0 + 4 * m;
    }
    const int slot0 = __builtin_amdgcn_readfirstlane(threadIdx.x - lane);
    auto stage = [&](const float *F, float *buf) {
        if (dma) {
#pragma unroll
            for (int k = 0; k < NB; k++)
                if (slot0 + 256 * k < PH * V4)
                    __builtin_amdgcn_global_load_lds((glb_cvoid *)(F + roff[k]), (lds_void *)(buf + 4 * (slot0 + 256 * k)), 16, 0, 0);
        } else {
#pragma unroll 1
            for (int k = 0; k < NB; k++) {
                const int i = threadIdx.x + 256 * k;
                if (i < PH * V4) {
                    const int xb = x0 + 4 * (i % V4), rowo = roff[k] - xb;
#pragma unroll
                    for (int e = 0; e < 4; e++) buf[4 * i + e] = F[rowo + (xb + e < cols ? xb + e : cols - 1)];
                }
            }
        }
    };
    stage(T, lkg_lds);
    lk_v2f acc[5][RP];
#pragma unroll
    for (int f = 0; f < 5; f++) {
        float *cur = lkg_lds + (f & 1) * (PH * TW), *nxt = lkg_lds + ((f + 1) & 1) * (PH * TW);
        __syncthreads();  // field f is staged (and field f - 1, in the other buffer, has been summed by every wave)
        if (f < 4) stage(T + (f + 1) * field, nxt);
        const float *lp = cur + rb * TW + 2 * cl;
#pragma unroll
        for (int j = 0; j < RP; j++) acc[f][j] = (lk_v2f){0.f, 0.f};
        lk_v2f W[N + RP - 1];
#pragma unroll
        for (int sidx = 0; sidx < RP - 1; sidx++) W[sidx] = *reinterpret_cast<const lk_v2f *>(lp + sidx * TW);
#pragma unroll
        for (int k = 0; k < N; k++) {
            W[k + RP - 1] = *reinterpret_cast<const lk_v2f *>(lp + (k + RP - 1) * TW);
            const float tap = t.k[k];
            const lk_v2f tap2 = (lk_v2f){tap, tap};
#pragma unroll
            for (int j = 0; j < RP; j++) acc[f][j] = __builtin_elementwise_fma(W[k + j], tap2, acc[f][j]);
            if ((k & 3) == 3)  // (as in pass A: keeps the unrolled window's reads from all being hoisted)
                asm volatile("" : "+v"(acc[f][0]), "+v"(acc[f][1]), "+v"(acc[f][2]), "+v"(acc[f][3]), "+v"(acc[f][4]),
                                  "+v"(acc[f][5]), "+v"(acc[f][6]), "+v"(acc[f][7])
                             :: "memory");
        }
    }
    // every sum in its register before any solve starts (the scheduler otherwise interleaves the double-precision
    // solves with the last field's chains)
#pragma unroll
    for (int f = 0; f < 5; f++)
        asm volatile("" : "+v"(acc[f][0]), "+v"(acc[f][1]), "+v"(acc[f][2]), "+v"(acc[f][3]), "+v"(acc[f][4]), "+v"(acc[f][5]),
                          "+v"(acc[f][6]), "+v"(acc[f][7])
                     :: "memory");
    const int x = x0 + 2 * cl;
    if (x >= cols) return;
    const bool pair_ok = x + 1 < cols && (ostride & 1) == 0 && (bstride & 1) == 0 &&
                         ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(base_u) |
                           reinterpret_cast<uintptr_t>(base_v)) & 7) == 0;
#pragma unroll
    for (int j = 0; j < RP; j++) {
        // (opaque, and behind the previous row's fence: the addresses and base-flow loads of all eight rows would
        // otherwise be formed ahead of the last field's sums -- 100+ registers)
        int y = y0 + rb + j;
        asm volatile("" : "+v"(y));
        if (y < rows) {
            lk_v2f uu, vv;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                float a_, b_;
                lk_solve(acc[0][j][e], acc[1][j][e], acc[2][j][e], acc[3][j][e], acc[4][j][e], a_, b_);
                uu[e] = a_;
                vv[e] = b_;
            }
            if (pair_ok) {
                if (base_u) {
                    uu = *reinterpret_cast<const lk_v2f *>(base_u + (size_t)y * bstride + x) + uu;
                    vv = *reinterpret_cast<const lk_v2f *>(base_v + (size_t)y * bstride + x) + vv;
                }
                *reinterpret_cast<lk_v2f *>(u + (size_t)y * ostride + x) = uu;
                *reinterpret_cast<lk_v2f *>(v + (size_t)y * ostride + x) = vv;
            } else {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    if (x + e < cols) {
                        float a_ = uu[e], b_ = vv[e];
                        if (base_u) {
                            a_ = base_u[(size_t)y * bstride + x + e] + a_;
                            b_ = base_v[(size_t)y * bstride + x + e] + b_;
                        }
                        u[(size_t)y * ostride + x + e] = a_;
                        v[(size_t)y * ostride + x + e] = b_;
                    }
                }
            }
        }
        // one row's solves at a time (unfenced, the 16 double-precision solves interleave and need 200+ registers)
        if (j + 1 < RP)
            asm volatile("" : "+v"(acc[0][j + 1]), "+v"(acc[1][j + 1]), "+v"(acc[2][j + 1]), "+v"(acc[3][j + 1]), "+v"(acc[4][j + 1])
                         :: "memory");
    }
}

__global__ __launch_bounds__(256) void lk_warp_kernel(const float *__restrict__ src, int sstride,
                                                       const float *__restrict__ du,
                                                       const float *__restrict__ dv, int fstride,
                                                       int rows, int cols,
                                                       float *__restrict__ dst, int dstride, size_t src_img, size_t flow_img,
                                                       size_t dst_img) {
    src += blockIdx.z * src_img; du += blockIdx.z * flow_img; dv += blockIdx.z * flow_img; dst += blockIdx.z * dst_img;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    dst[(size_t)y * dstride + x] = warp_sample(src, rows, cols, sstride, x, y,
                                               du[(size_t)y * fstride + x],
                                               dv[(size_t)y * fstride + x]);
}

// lk::warp, tiled (r04): a 64x16 tile whose `src` window (tile + 8 px margin, zeros outside the image = cv::remap's
// BORDER_CONSTANT) is staged in LDS with 16-byte loads; a thread takes four adjacent pixels -- float4 loads of du and dv,
// one float4 store -- and each sample's four bilinear taps come from the window (warp_sample_staged: the level
// kernel's sampler, the contract's arithmetic; a flow that leaves the margin takes its global-memory fallback).
// 8 B of flow + 4 B of output per pixel are 16-byte traffic and the taps cost no vector-memory instruction:
// 1080p 19.2 -> see DESIGN.md.  Needs 16-byte aligned rows everywhere and cols % 4 == 0 (a chunk never straddles
// the image edge); other shapes keep the one-pixel-per-thread kernel above.
constexpr int WT_W = 64, WT_H = 16, WT_M = 8, WT_NW = WT_W + 2 * WT_M + 4, WT_NH = WT_H + 2 * WT_M + 1;
typedef float wv4f __attribute__((ext_vector_type(4)));
typedef float wv2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void lk_warp_tiled_kernel(const float *__restrict__ src, int sstride,
                                                             const float *__restrict__ du,
                                                             const float *__restrict__ dv, int fstride,
                                                             int rows, int cols,
                                                             float *__restrict__ dst, int dstride, size_t src_img, size_t flow_img,
                                                             size_t dst_img) {
    src += blockIdx.z * src_img; du += blockIdx.z * flow_img; dv += blockIdx.z * flow_img; dst += blockIdx.z * dst_img;
    __shared__ __attribute__((aligned(16))) float N[WT_NH * WT_NW];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * WT_W, y0 = blockIdx.y * WT_H;
    const int nx0 = x0 - WT_M, ny0 = y0 - WT_M;
    constexpr int V4 = WT_NW / 4, NCH = WT_NH * V4, NB = (NCH + 255) / 256;
    // the thread's own flow quad first: its loads share one memory round trip with the window's
    const int x = x0 + 4 * (tid & 15), y = y0 + (tid >> 4);
    const bool mine = x < cols && y < rows;
    wv4f fu = {0.f, 0.f, 0.f, 0.f}, fv = fu;
    if (mine) {
        fu = *reinterpret_cast<const wv4f *>(du + (size_t)y * fstride + x);
        fv = *reinterpret_cast<const wv4f *>(dv + (size_t)y * fstride + x);
    }
    wv4f w[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) {  // all loads first, then the LDS writes
        const int i = tid + k * 256 < NCH ? tid + k * 256 : NCH - 1;
        const int ly = i / V4, lv = i - ly * V4;
        const int gy = ny0 + ly, gx = nx0 + 4 * lv;
        const bool in = (unsigned)gy < (unsigned)rows && (unsigned)gx < (unsigned)cols;  // cols % 4 == 0: whole chunk
        w[k] = (wv4f){0.f, 0.f, 0.f, 0.f};
        if (in) w[k] = *reinterpret_cast<const wv4f *>(src + (size_t)gy * sstride + gx);
    }
#pragma unroll
    for (int k = 0; k < NB; k++)
        if (tid + k * 256 < NCH) *reinterpret_cast<wv4f *>(N + 4 * (tid + k * 256)) = w[k];
    __syncthreads();
    if (!mine) return;
    wv4f out;
    const float yf32 = 32.f * (float)y;
#pragma unroll
    for (int j = 0; j < 4; j++)
        out[j] = warp_sample_staged<WT_NW, WT_NH, 32>(N, nx0, ny0, src, rows, cols, sstride,
                                                      (wv2f){32.f * (float)(x + j), yf32}, (wv2f){fu[j], fv[j]});
    *reinterpret_cast<wv4f *>(dst + (size_t)y * dstride + x) = out;
}

// The generic chain's two steps between levels as ONE launch (r05): base flow = 2 * pyrUp(coarse flow) for the tile's own
// pixels (pyr_up_tiled_kernel's chains: row taps left -> right over the coarse rows, column taps top -> bottom, fmaf from
// +0, then * 2 -- OpticalFlow.cpp:140-145), written to bu / bv for the solve that follows, and lk::warp of `next` by exactly
// those registers (lk_warp_tiled_kernel's body).  Saves the 8 B per pixel the warp launch read back, and a launch per level.
// blockIdx.z = pair.  Needs what the tiled warp needs (16-byte rows everywhere, cols % 4 == 0) and R == 2 fr, C == 2 fc.
__global__ __launch_bounds__(256) void lk_expand_warp_kernel(const float *__restrict__ cu, const float *__restrict__ cv, int fr, int fc,
                                                              size_t coarse_img, const float *__restrict__ src, int sstride,
                                                              size_t src_img, int rows, int cols, float *__restrict__ bu,
                                                              float *__restrict__ bv, float *__restrict__ dst, size_t fine_img) {
    constexpr int CW = WT_W / 2 + 2, CH = WT_H / 2 + 2;  // coarse block: rows r0 - 1 .., columns c0 - 1 ..
    __shared__ __attribute__((aligned(16))) float N[WT_NH * WT_NW];
    __shared__ float Cs[2][CH][CW + 1];
    __shared__ __attribute__((aligned(16))) float Rp[2][CH][WT_W];
    const unsigned pair = blockIdx.z;
    cu += pair * coarse_img; cv += pair * coarse_img; src += pair * src_img;
    bu += pair * fine_img; bv += pair * fine_img; dst += pair * fine_img;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * WT_W, y0 = blockIdx.y * WT_H;
    const int nx0 = x0 - WT_M, ny0 = y0 - WT_M;
    const int c0 = x0 >> 1, r0 = y0 >> 1;
    const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    // every global load first: the `next` window (as lk_warp_tiled_kernel) and the coarse block of both fields
    constexpr int V4 = WT_NW / 4, NCH = WT_NH * V4, NB = (NCH + 255) / 256;
    wv4f w[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int i = tid + k * 256 < NCH ? tid + k * 256 : NCH - 1;
        const int ly = i / V4, lv = i - ly * V4;
        const int gy = ny0 + ly, gx = nx0 + 4 * lv;
        const bool in = (unsigned)gy < (unsigned)rows && (unsigned)gx < (unsigned)cols;  // cols % 4 == 0: whole chunk
        w[k] = (wv4f){0.f, 0.f, 0.f, 0.f};
        if (in) w[k] = *reinterpret_cast<const wv4f *>(src + (size_t)gy * sstride + gx);
    }
    constexpr int NC = 2 * CH * CW, NCB = (NC + 255) / 256;
    float cval[NCB];
#pragma unroll
    for (int k = 0; k < NCB; k++) {
        const int i = tid + k * 256 < NC ? tid + k * 256 : NC - 1;
        const int f = i / (CH * CW), j = i - f * (CH * CW), ly = j / CW, lx = j - ly * CW;
        const float *cp = f ? cv : cu;
        cval[k] = cp[(size_t)clampi(r0 - 1 + ly, 0, fr - 1) * fc + clampi(c0 - 1 + lx, 0, fc - 1)];
    }
#pragma unroll
    for (int k = 0; k < NB; k++)
        if (tid + k * 256 < NCH) *reinterpret_cast<wv4f *>(N + 4 * (tid + k * 256)) = w[k];
#pragma unroll
    for (int k = 0; k < NCB; k++) {
        const int i = tid + k * 256;
        if (i < NC) {
            const int f = i / (CH * CW), j = i - f * (CH * CW), ly = j / CW, lx = j - ly * CW;
            Cs[f][ly][lx] = cval[k];
        }
    }
    __syncthreads();
    // row pass: (field, coarse row, fine column) -- 2 x CH x 64 values
    for (int i = tid; i < 2 * CH * WT_W; i += 256) {
        const int f = i / (CH * WT_W), j = i - f * (CH * WT_W), cr = j / WT_W, t = j - cr * WT_W;
        const int x = x0 + t;
        if (x >= cols) continue;  // (past the frame: no output reads it, and its taps would fall outside the staged block)
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 5; k++) acc = fmaf(Cs[f][cr][(reflect101(x + k - 2, cols) >> 1) - (c0 - 1)], g5[k], acc);
        Rp[f][cr][t] = acc;
    }
    __syncthreads();
    const int x = x0 + 4 * (tid & 15), y = y0 + (tid >> 4);
    if (x >= cols || y >= rows) return;
    wv4f fu = {0.f, 0.f, 0.f, 0.f}, fv = fu;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int ri = (reflect101(y + k - 2, rows) >> 1) - (r0 - 1);
        const wv4f gk = {g5[k], g5[k], g5[k], g5[k]};
        fu = __builtin_elementwise_fma(*reinterpret_cast<const wv4f *>(&Rp[0][ri][4 * (tid & 15)]), gk, fu);
        fv = __builtin_elementwise_fma(*reinterpret_cast<const wv4f *>(&Rp[1][ri][4 * (tid & 15)]), gk, fv);
    }
    fu = fu * (wv4f){2.f, 2.f, 2.f, 2.f};
    fv = fv * (wv4f){2.f, 2.f, 2.f, 2.f};
    *reinterpret_cast<wv4f *>(bu + (size_t)y * cols + x) = fu;
    *reinterpret_cast<wv4f *>(bv + (size_t)y * cols + x) = fv;
    wv4f out;
    const float yf32 = 32.f * (float)y;
#pragma unroll
    for (int j = 0; j < 4; j++)
        out[j] = warp_sample_staged<WT_NW, WT_NH, 32>(N, nx0, ny0, src, rows, cols, sstride,
                                                      (wv2f){32.f * (float)(x + j), yf32}, (wv2f){fu[j], fv[j]});
    *reinterpret_cast<wv4f *>(dst + (size_t)y * cols + x) = out;
}

// Launches it when the shapes allow; *done = false otherwise (the caller then runs the two launches).
static int launch_expand_warp(hipStream_t s, const float *cu, const float *cv, int fr, int fc, size_t coarse_img, const float *src,
                              int sstride, size_t src_img, int rows, int cols, float *bu, float *bv, float *dst, size_t fine_img,
                              int batch, bool *done) {
    *done = rows == 2 * fr && cols == 2 * fc && (cols & 3) == 0 && (sstride & 3) == 0 && ((src_img | fine_img) & 3) == 0 &&
            ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(bu) | reinterpret_cast<uintptr_t>(bv) |
              reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if (!*done) return MICV_OK;
    lk_expand_warp_kernel<<<dim3(cdiv(cols, WT_W), cdiv(rows, WT_H), batch), 256, 0, s>>>(cu, cv, fr, fc, coarse_img, src, sstride, src_img,
                                                                                       rows, cols, bu, bv, dst, fine_img);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// micv_flow_bound_check_dev: raises *flag when a flow value of the given rows exceeds `bound` in magnitude
// (or is not finite).  One atomic per wave that sees a violation.
__global__ __launch_bounds__(256) void flow_bound_kernel(const float *__restrict__ v, size_t pair_elems, int stride,
                                                         int cols, int row_begin, int nrows, float bound,
                                                         unsigned *__restrict__ flag) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    bool bad = false;
    if (x < cols && y < nrows) {
        const float t = v[blockIdx.z * pair_elems + (size_t)(row_begin + y) * stride + x];
        bad = !(fabsf(t) <= bound);  // true for NaN as well
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

int launch_warp(hipStream_t s, const float *src, int sstride, const float *du, const float *dv,
                int fstride, int rows, int cols, float *dst, int dstride, int batch, size_t src_img, size_t flow_img,
                size_t dst_img) {
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(dv) |
                       reinterpret_cast<uintptr_t>(dst)) & 15) == 0 &&
                     ((sstride | fstride | dstride | cols) & 3) == 0 && ((src_img | flow_img | dst_img) & 3) == 0;
    if (vec) {
        lk_warp_tiled_kernel<<<dim3(cdiv(cols, WT_W), cdiv(rows, WT_H), batch), 256, 0, s>>>(src, sstride, du, dv, fstride, rows, cols,
                                                                                          dst, dstride, src_img, flow_img, dst_img);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    lk_warp_kernel<<<dim3(cdiv(cols, 64), cdiv(rows, 4), batch), 256, 0, s>>>(src, sstride, du, dv, fstride,
                                                                               rows, cols, dst, dstride, src_img, flow_img, dst_img);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// Level sizes from which the two-launch forms win (tools/probes/generic_form_bench.py, one level per call, us):
//   window 43, unrolled:  540 x 960: 38 vs 32 (four launches)   1080 x 1920: 51 vs 79
//   window 9, run-time taps:  68 x 120: 31 vs 30   135 x 240: 23 vs 35   1080 x 1920: 37 vs 64
constexpr size_t kTwoLaunchMinPixels = 16 * 1024, kTwoLaunchMinPixelsUnrolled = 1024 * 1024;
// Scratch floats the generic level needs: 10 planar fields.
static size_t lk_generic_scratch(int rows, int cols) { return (size_t)rows * cols * 10; }

// MICV_OPT_LK_FORCE_GENERIC: 1 = generic kernels, form by size; 2 = always four launches; 3 = always two
static int lk_generic_form(const micv_ctx *ctx) {
    const int o = ctx->opt[MICV_OPT_LK_FORCE_GENERIC];
    return o == 2 ? 2 : (o == 3 ? 1 : 0);
}

// lk::calcOpticalFlow on (prev, next); output = base + flow when base_u != nullptr.
static int lk_level_generic(hipStream_t s, const float *prev, int pstride, const float *next,
                            int nstride, int rows, int cols, int win, const float *base_u,
                            const float *base_v, int bstride, float *u, float *v, int ostride,
                            float *scratch, int form = 0, int batch = 1, LkPairs pp = LkPairs()) {
    // scratch: 10 planar fields PER PAIR -- the five product fields of every pair first, then the five row sums of every
    // pair (so the four-launch form's filter launches see 5 x batch fields one field apart)
    const size_t n = (size_t)rows * cols;
    float *S = scratch, *T = scratch + 5 * n * batch;
    const unsigned zb = (unsigned)batch;
    Taps g;
    gaussian_taps(win, (double)((float)win / 3.f), &g);  // OpticalFlow.cpp:73
    // form: 0 = by size, 2 = always the four launches, 1 = always two (MICV_OPT_LK_FORCE_GENERIC 2 / 1 on a window
    // the fused kernels do not cover).  Small levels are latency-bound and the two long kernels lose there.
    // (the sizes are the LAUNCH's: a batch of small levels fills the GPU like one large level -- window 43, 8 x 1080p: 0.757 ->
    // 0.746 ms per call, 4 pairs 0.445 -> 0.437)
    const size_t launch_px = (size_t)rows * cols * batch;
    const bool two = form == 1 || (form == 0 && launch_px >= kTwoLaunchMinPixels);
    if (g.n == 43 && (form == 1 || (form == 0 && launch_px >= kTwoLaunchMinPixelsUnrolled))) {  // config/ps5.yaml:11
        constexpr int N = 43;
        const size_t lds_a = (size_t)3 * 8 * ((((256 + N - 1) + 3) & ~3) + 4) * sizeof(float);
        lk_products_rows_pk_kernel<N><<<dim3(cdiv(cols, 256), cdiv(rows, 8), zb), 256, lds_a, s>>>(prev, pstride, next, nstride,
                                                                                                   rows, cols, T, n, g, pp);
        MICV_LAUNCH_CHECK();
        static thread_local int attr_dev = -1;
        int dev = 0;
        MICV_HIP(hipGetDevice(&dev));
        if (attr_dev != dev) {  // two staging buffers: 54 KB of dynamic LDS
            MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lk_cols_solve_pk_kernel<N, 64>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_dev = dev;
        }
        // 64 x 64 tiles (r05; 128 x 32 before): 8 x 1080p 0.777 -> 0.761 ms per call, 4 pairs 0.451 -> 0.444, one pair the same
        lk_cols_solve_pk_kernel<N, 64><<<dim3(cdiv(cols, 64), cdiv(rows, 64), zb), 256, (size_t)2 * 64 * (64 + N - 1 + 2) * sizeof(float), s>>>(
            T, n, rows, cols, g, base_u, base_v, bstride, u, v, ostride, pp);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    if (two && g.n >= 5) {  // two launches: images -> row sums -> flow
        const size_t lds_a = (size_t)3 * 8 * ((((256 + g.n - 1) + 3) & ~3) + 4) * sizeof(float);
        lk_products_rows_kernel<<<dim3(cdiv(cols, 256), cdiv(rows, 8), zb), 256, lds_a, s>>>(prev, pstride, next, nstride, rows,
                                                                                             cols, T, n, g, pp);
        MICV_LAUNCH_CHECK();
        lk_cols_solve_kernel<<<dim3(cdiv(cols, 64), cdiv(rows, 32), zb), 256, (size_t)64 * (32 + g.n - 1) * sizeof(float), s>>>(
            T, n, rows, cols, g, base_u, base_v, bstride, u, v, ostride, pp);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    const dim3 grid(cdiv(cols, 64), cdiv(rows, 4), zb);
    lk_products_kernel<<<grid, 256, 0, s>>>(prev, pstride, next, nstride, rows, cols, S, n, pp);
    MICV_LAUNCH_CHECK();
    MICV_TRY(launch_filter_rows(s, S, cols, n, T, cols, n, rows, cols, 5 * batch, g));
    MICV_TRY(launch_filter_cols(s, T, cols, n, S, cols, n, rows, cols, 5 * batch, g));
    lk_solve_kernel<<<grid, 256, 0, s>>>(S, n, rows, cols, base_u, base_v, bstride, u, v, ostride, pp);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// ---- pyramid driver ---------------------------------------------------------------------

struct PyrPlan {
    int levels;
    int rows[16], cols[16];
    size_t lvl_off[16];   // offset (floats) of level l (l >= 1) inside one image's pyramid block
    size_t pyr_elems;     // floats per image for levels >= 1
};

static void make_plan(int rows, int cols, int levels, PyrPlan *p) {
    p->levels = levels;
    size_t off = 0;
    for (int l = 0; l < levels; l++) {
        p->rows[l] = rows >> l;
        p->cols[l] = cols >> l;
        p->lvl_off[l] = off;
        if (l >= 1) off += ((size_t)p->rows[l] * p->cols[l] + 63) & ~size_t(63);
    }
    p->pyr_elems = off;
}

// Everything one coarse-to-fine chain needs: a contiguous run of pairs, its stream, and its private
// flow ping-pong buffers.
struct LkChain {
    hipStream_t s;
    const float *prev, *next;  // level-0 images of the chain's first pair
    size_t pair_elems;
    int stride;
    const float *ppyr[16], *npyr[16];  // levels >= 1 of the chain's first pair (dense, pairs contiguous)
    int nb;                            // pairs in the chain
    float *fu[2], *fv[2];              // flow ping-pong (level-0 size per pair)
    float *u, *v;                      // outputs of the chain's first pair
    size_t opair_elems;
    int ostride;
    bool profile;                      // bracket level launches with events (group 0 only)
    int direct_from = 0;               // levels >= direct_from (>= 1) are read straight from level 0; 0 = none
    // MICV_OPT_LK_BUILD_OVERLAP: no build launch -- the top level reads level 0 itself and the launch of every level
    // k >= 2 carries the build of level k - 1 as extra workgroups (LkBuildJob, lk_fused.hpp)
    bool carry_build = false;
    float *pyr_a = nullptr, *pyr_b = nullptr;  // the arenas of the whole batch (carry_build: nb = batch)
    // MICV_OPT_LK_SPLIT (lk_split.hip): the chain's block of padded gradient planes, sized for its largest split level
    float *grad = nullptr;
    size_t grad_elems = 0;  // floats available (all pairs of the chain)
};

// Fused path: OpticalFlow.cpp:135-163 for all pairs of the chain, one launch per level.
static int lk_chain_fused(micv_ctx *ctx, const PyrPlan &plan, const LkChain &c, int win) {
    int cur = 0;         // ping-pong index holding the flow of the previous (coarser) level
    int fr = 0, fc = 0;  // its dims
    for (int level = 0; level < plan.levels; level++) {
        const int k = plan.levels - 1 - level;
        const int R = plan.rows[k], C = plan.cols[k];
        const bool last = (k == 0);
        const size_t lvl_elems = (size_t)R * C;
        LkLevelArgs a;
        a.rows = R; a.cols = C; a.batch = c.nb; a.win = win;
        a.prev = last ? c.prev : c.ppyr[k];
        a.next = last ? c.next : c.npyr[k];
        a.img_stride = last ? c.stride : C;
        a.img_pair = last ? c.pair_elems : lvl_elems;
        const bool direct = (c.direct_from > 0 && k >= c.direct_from) || (c.carry_build && k == plan.levels - 1);
        if (!last && direct) {
            // Pyramids.cu:31 applied k times: L_k(y, x) = L_0(2^k y + 2^k - 1, 2^k x + 2^k - 1) -- the level is
            // level 0 seen through a row stride of 2^k rows and a pixel stride of 2^k (lk_fused.hip, GATHER)
            const size_t o = ((size_t)1 << k) - 1;
            a.prev = c.prev + o * c.stride + o;
            a.next = c.next + o * c.stride + o;
            a.img_stride = c.stride << k;
            a.img_xstride = 1 << k;
            a.img_pair = c.pair_elems;
        }
        // this level's flow goes to the user's u/v at the finest level
        a.out_u = last ? c.u : c.fu[cur ^ 1];
        a.out_v = last ? c.v : c.fv[cur ^ 1];
        a.out_pair = last ? c.opair_elems : lvl_elems;
        a.out_stride = last ? c.ostride : C;
        a.add_base = 1;
        a.row_begin = 0; a.row_end = R;
        a.stamps = (last && c.profile) ? ctx->stamps : nullptr;  // phase stamps: level 0 only
        a.narrow = ctx->opt[MICV_OPT_LK_NARROW_TILES];
        a.ctx = ctx;
        a.max_chain = ctx->opt[MICV_OPT_LK_CHAIN];
        a.short_tiles = ctx->opt[MICV_OPT_LK_SHORT_TILES];
        a.stream_tiles = ctx->opt[MICV_OPT_LK_STREAM];
        a.tall_tiles = ctx->opt[MICV_OPT_LK_TALL_TILES];
        a.split = ctx->opt[MICV_OPT_LK_SPLIT];
        a.strip = ctx->opt[MICV_OPT_LK_STRIP];
        if (c.grad && !direct) {
            const LkGradGeom gg = lk_grad_geom(R, C, win);
            if (gg.pair_elems * c.nb <= c.grad_elems) {
                a.grad = c.grad;
                a.grad_pair = gg.pair_elems;
                a.grad_pitch = gg.pitch;
                a.grad_rows = gg.rows;
                a.grad_pad = gg.pad;
            }
        }
        bool out_in_cur = false;
        if (c.carry_build && k >= 2) {  // this launch also builds level k - 1, which the next launch reads
            LkBuildJob &j = a.job;
            j.level = k - 1;
            j.src_a = c.prev; j.src_b = c.next; j.img_elems = c.pair_elems; j.sstride = c.stride;
            j.batch = c.nb; j.rows = plan.rows[0]; j.cols = plan.cols[0];
            j.pyr_a = c.pyr_a; j.pyr_b = c.pyr_b;
            j.dst_off = plan.lvl_off[k - 1] * c.nb;
            j.units = lk_build_units(j.rows, j.cols, j.level, j.batch);
            j.blocks = j.units < 4096 ? j.units : 4096;
        }
        if (c.profile) MICV_TRY(ctx->prof_begin(k, c.s));
        if (level == 0) {
            a.mode = LK_FLOW_NONE;  // du = dv = 0 (:132-133)
            a.flow_u = a.flow_v = nullptr; a.flow_rows = a.flow_cols = 0; a.flow_pair = 0;
        } else if (2 * fr == R && 2 * fc == C) {
            a.mode = LK_FLOW_COARSE;  // pyrUp + x2 fused into the level kernel
            a.flow_u = c.fu[cur]; a.flow_v = c.fv[cur];
            a.flow_rows = fr; a.flow_cols = fc; a.flow_pair = (size_t)fr * fc;
        } else {
            // :139-151 odd sizes -> pyrUp, x2, cv::resize; one launch for all pairs, both fields
            float *full_u = c.fu[cur ^ 1], *full_v = c.fv[cur ^ 1];  // this level's base flow
            MICV_TRY(launch_flow_expand_resize(c.s, c.fu[cur], c.fv[cur], fr, fc, (size_t)fr * fc, full_u,
                                               full_v, R, C, lvl_elems, c.nb));
            a.mode = LK_FLOW_FULL;
            a.flow_u = full_u; a.flow_v = full_v;
            a.flow_rows = R; a.flow_cols = C; a.flow_pair = lvl_elems;
            // The tiled kernel reads the base flow of its halo pixels, which other workgroups own:
            // the output must NOT alias the base.  The coarse buffers are dead after the expand
            // launch, so the result goes there and the ping-pong index stays put.
            if (!last) {
                a.out_u = c.fu[cur]; a.out_v = c.fv[cur];
                out_in_cur = true;
            }
        }
        MICV_TRY(launch_lk_level_fused(c.s, a));
        if (c.profile) MICV_TRY(ctx->prof_end(k, c.s));
        if (!out_in_cur) cur ^= 1;
        fr = R;
        fc = C;
    }
    return MICV_OK;
}

// Generic path (any odd window): the same chain with the generic kernels, every step ONE launch for all pairs of the
// chain (r05; blockIdx.z = pair): a level is at most seven launches whatever the batch -- base flow (a memset, the batched
// 2 x pyrUp, or the fused path's expand + resize launch for odd sizes), lk::warp, and the two to four launches of the level.
static int lk_chain_generic(micv_ctx *ctx, const PyrPlan &plan, const LkChain &c, int win, float *warped,
                            float *gen) {
    hipStream_t s = c.s;
    int cur = 0, fr = 0, fc = 0;
    for (int level = 0; level < plan.levels; level++) {
        const int k = plan.levels - 1 - level;
        const int R = plan.rows[k], C = plan.cols[k];
        const bool last = (k == 0);
        const size_t lvl_elems = (size_t)R * C;
        if (c.profile) MICV_TRY(ctx->prof_begin(k, s));
        const float *pk = last ? c.prev : c.ppyr[k];
        const float *nk = last ? c.next : c.npyr[k];
        const int ist = last ? c.stride : C;
        const size_t img_pair = last ? c.pair_elems : lvl_elems;
        float *bu = c.fu[cur ^ 1], *bv = c.fv[cur ^ 1];  // this level's base flow, pairs lvl_elems apart
        bool warped_done = false;
        if (level == 0) {
            MICV_HIP(hipMemsetAsync(bu, 0, lvl_elems * 4 * c.nb, s));  // OpticalFlow.cpp:132-133
            MICV_HIP(hipMemsetAsync(bv, 0, lvl_elems * 4 * c.nb, s));
        } else if (2 * fr == R && 2 * fc == C) {
            // :140-145 and :155 in one launch when the shapes allow (16-byte rows), else the batched pyrUp and the warp below
            MICV_TRY(launch_expand_warp(s, c.fu[cur], c.fv[cur], fr, fc, (size_t)fr * fc, nk, ist, img_pair, R, C, bu, bv, warped, lvl_elems,
                                        c.nb, &warped_done));
            if (!warped_done)
                MICV_TRY(launch_pyr_up_batch(s, c.fu[cur], c.fv[cur], fr, fc, (size_t)fr * fc, bu, bv, lvl_elems, 2.f, c.nb));
        } else {  // :148-151, odd sizes: 2 x pyrUp then cv::resize, one launch (the fused path's)
            MICV_TRY(launch_flow_expand_resize(s, c.fu[cur], c.fv[cur], fr, fc, (size_t)fr * fc, bu, bv, R, C, lvl_elems, c.nb));
        }
        if (!warped_done) MICV_TRY(launch_warp(s, nk, ist, bu, bv, C, R, C, warped, C, c.nb, img_pair, lvl_elems, lvl_elems));  // :155
        // in place is fine here: the solve reads and writes the same pixel in one thread
        LkPairs pp;
        pp.prev = img_pair;
        pp.next = lvl_elems;  // the warped frames
        pp.base = lvl_elems;
        pp.out = last ? c.opair_elems : lvl_elems;
        MICV_TRY(lk_level_generic(s, pk, ist, warped, C, R, C, win, bu, bv, C, last ? c.u : bu, last ? c.v : bv,
                                  last ? c.ostride : C, gen, lk_generic_form(ctx), c.nb, pp));  // :159-162
        if (c.profile) MICV_TRY(ctx->prof_end(k, s));
        cur ^= 1;
        fr = R;
        fc = C;
    }
    return MICV_OK;
}

// lk::calcOpticalFlowPyr over a batch of pairs.
static int lk_pyr_batch(micv_ctx *ctx, hipStream_t s, const float *prev, const float *next,
                        int batch, size_t pair_elems, int rows, int cols, int stride, int win,
                        int levels, float *u, float *v, size_t opair_elems, int ostride,
                        bool allow_fused) {
    PyrPlan plan;
    make_plan(rows, cols, levels, &plan);
    const bool fused = allow_fused && lk_fused_supports(win);
    const size_t n0 = (size_t)rows * cols;
    // Scratch: pyramids (levels >= 1) of both images for the whole batch, two flow ping-pong pairs
    // at level-0 size per pair, and the generic path's temporaries.
    const size_t flow_elems = (n0 + 63) & ~size_t(63);
    size_t total = Carver::need(plan.pyr_elems * batch, 4) * 2 + Carver::need(flow_elems * batch, 4) * 4;
    // Generic path: every step is one launch for the whole batch (r05; r04 ran the pairs on four forked streams, one
    // launch per pair and step -- ~330 launches per call at 8 pairs and 5 levels, and the host was the limit): the warped
    // frames and the ten product / sum planes of every pair.
    // (at most kGenChunk pairs per launch: blockIdx.z carries 5 x pairs fields in the four-launch form)
    // ... and at most 1 GiB of these temporaries (11 level-0 planes per pair: 8 x 1080p take 0.73 GB and stay one chunk;
    // 64 x 1080p reserved 5.8 GB of the context's one scratch block for good)
    const size_t gen_pair_bytes = (n0 + lk_generic_scratch(rows, cols)) * sizeof(float);
    const size_t gen_fit = (size_t(1) << 30) / gen_pair_bytes;
    const int kGenChunk = gen_fit < 1 ? 1 : (gen_fit > 4096 ? 4096 : (int)gen_fit);
    const int gen_nb = batch < kGenChunk ? batch : kGenChunk;
    if (!fused) total += Carver::need(n0 * gen_nb, 4) + Carver::need(lk_generic_scratch(rows, cols) * gen_nb, 4);
    // MICV_OPT_LK_SPLIT: the padded gradient planes of the largest level some launch of the chain will split (the
    // levels run one after the other on the chain's stream and share the block)
    size_t grad_elems = 0;
    if (fused && lk_split_supports(win) && ctx->opt[MICV_OPT_LK_SPLIT] > 0)
        for (int l = 0; l + 1 < levels; l++)  // the coarsest level has no coarse flow
            if (plan.rows[l] == 2 * plan.rows[l + 1] && plan.cols[l] == 2 * plan.cols[l + 1] &&
                lk_split_wanted(plan.rows[l], plan.cols[l], batch, win, 1, ctx->opt[MICV_OPT_LK_SPLIT])) {
                const size_t e = lk_grad_geom(plan.rows[l], plan.cols[l], win).pair_elems * batch;
                grad_elems = e > grad_elems ? e : grad_elems;
            }
    total += Carver::need(grad_elems, 4);
    void *base;
    MICV_TRY(ctx->reserve(total, &base));
    Carver carve(base);
    float *grad = grad_elems ? carve.take<float>(grad_elems) : nullptr;
    float *ppyr = carve.take<float>(plan.pyr_elems * batch);
    float *npyr = carve.take<float>(plan.pyr_elems * batch);
    float *fu[2] = {carve.take<float>(flow_elems * batch), carve.take<float>(flow_elems * batch)};
    float *fv[2] = {carve.take<float>(flow_elems * batch), carve.take<float>(flow_elems * batch)};

    // Pyramids (Pyramids.cpp:19-23): every level is a direct decimation of level 0, so ONE launch
    // builds all levels of both images of every pair.  Level l of pair b sits at
    // pyr + lvl_off[l]*batch + b*rows_l*cols_l.
    // Fused path, MICV_OPT_LK_DIRECT_LEVELS = n > 0: every level >= n takes its images from level 0 itself and the
    // build launch only makes the levels below that (none for n = 1).  Measured on MI355X (r04, 8 x 1080p, A/B on one
    // box, profiles/r04/direct_levels_ab.txt): one pass at a time 0.327 -> 0.318 ms (the 20 us build launch goes,
    // the gather-staged levels take 61.6 / 27.9 / 27.5 / 12.4 us instead of 58.3 / 26.0 / 23.3 / 11.1: a dword LDS-DMA
    // moves 256 B where the 16-byte form moves 1 KiB, and a DMA instruction costs the issuing wave 60+ cycles
    // whatever it moves); with two passes in flight -- the bench's configuration -- 0.294-0.300 against 0.288 ms: the
    // bandwidth-bound build overlaps the other pass's compute-bound levels, the longer level kernels do not.
    // Off by default.
    int direct_from = 0;
    if (fused && levels > 1 && lk_fused_supports_direct_levels(win) && !ctx->opt[MICV_OPT_LK_NARROW_TILES] &&
        ctx->opt[MICV_OPT_LK_TALL_TILES] <= 0 && !ctx->opt[MICV_OPT_LK_STREAM]) {
        const int o = ctx->opt[MICV_OPT_LK_DIRECT_LEVELS];
        direct_from = o <= 0 ? 0 : o;  // off by default: measured below
        if ((long long)stride << (levels - 1) > 0x7fffffffLL) direct_from = 0;  // the row stride is an int
    }
    const int build_levels = direct_from > 0 ? (direct_from < levels ? direct_from : levels) : levels;
    // MICV_OPT_LK_BUILD_OVERLAP (r04, default on; window 15, >= 3 levels, 16-byte rows): NO build launch.  The chain
    // starts at the coarsest level with launches of a few workgroups whose time is latency, and 94 % of the build's
    // bytes are level 1, which nobody reads before the second-to-last launch.  So the top level reads level 0 itself
    // (the GATHER form), and the launch of every level k >= 2 carries the build of level k - 1 -- what the NEXT launch
    // reads -- as extra workgroups dispatched behind its tiles.  A side stream for
    // the level-1 build was tried first and LOST (one pass 0.3233 -> 0.3357 ms, single pair 0.0817 -> 0.0958: two
    // cross-stream event waits cost more than the build).
    // Measured on MI355X (A/B on one box, profiles/r04/build_overlap_ab.txt): single 1080p pair 0.0821 -> 0.0788 ms per
    // call; 8 pairs, one pass at a time, 0.3260 -> 0.3159 ms; 8 pairs with TWO passes in flight (the bench) 0.2842 ->
    // 0.2886 ms -- there the build launch (no LDS, 256 threads) already runs in the wave slots the other pass's level-0
    // tiles leave free, while carried build workgroups hold a level tile's LDS.  So the default (0) carries the build for
    // single pairs only -- the latency case; 1 carries it for any batch.
    int groups = 1;
    if (fused && batch >= 2) {
        // one group by default: with the r02 level kernels a second group buys nothing for a single pass
        // (0.4217 vs 0.4228 ms) and costs 2.5 % when passes overlap across contexts (bench.py --inflight 2)
        groups = ctx->opt[MICV_OPT_LK_STREAM_GROUPS] > 0 ? ctx->opt[MICV_OPT_LK_STREAM_GROUPS] : 1;
        groups = groups > 4 ? 4 : groups;
        if (groups > batch) groups = batch;
    }
    const bool carry_build = fused && groups == 1 && direct_from == 0 && levels >= 3 &&
                             (ctx->opt[MICV_OPT_LK_BUILD_OVERLAP] > 0 || (ctx->opt[MICV_OPT_LK_BUILD_OVERLAP] == 0 && batch == 1)) &&
                             lk_fused_supports_direct_levels(win) && !ctx->opt[MICV_OPT_LK_NARROW_TILES] &&
                             ctx->opt[MICV_OPT_LK_TALL_TILES] <= 0 && !ctx->opt[MICV_OPT_LK_STREAM] &&
                             (long long)stride << (levels - 1) <= 0x7fffffffLL && (stride & 3) == 0 && (pair_elems & 3) == 0 &&
                             (cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(prev) | reinterpret_cast<uintptr_t>(next)) & 15) == 0;
    if (build_levels > 1 && !carry_build) {
        float *pd[16], *nd[16];
        pd[0] = nd[0] = nullptr;
        for (int l = 1; l < levels; l++) {
            pd[l] = ppyr + plan.lvl_off[l] * batch;
            nd[l] = npyr + plan.lvl_off[l] * batch;
        }
        MICV_TRY(launch_pyr_build2(s, prev, next, pair_elems, stride, rows, cols, build_levels, pd, nd, batch));
    }
    auto make_chain = [&](hipStream_t cs, int b0, int nb, bool profile) {
        LkChain c;
        c.s = cs;
        c.prev = prev + b0 * pair_elems;
        c.next = next + b0 * pair_elems;
        c.pair_elems = pair_elems;
        c.stride = stride;
        for (int l = 1; l < levels; l++) {
            const size_t off = plan.lvl_off[l] * batch + (size_t)b0 * plan.rows[l] * plan.cols[l];
            c.ppyr[l] = ppyr + off;
            c.npyr[l] = npyr + off;
        }
        c.nb = nb;
        for (int i = 0; i < 2; i++) {
            c.fu[i] = fu[i] + flow_elems * b0;
            c.fv[i] = fv[i] + flow_elems * b0;
        }
        c.u = u + b0 * opair_elems;
        c.v = v + b0 * opair_elems;
        c.opair_elems = opair_elems;
        c.ostride = ostride;
        c.profile = profile;
        c.direct_from = direct_from;
        c.carry_build = carry_build;
        c.pyr_a = ppyr;
        c.pyr_b = npyr;
        if (grad) {  // a group's share of the block: pairs b0 .. b0 + nb of `batch`
            c.grad = grad + (grad_elems / batch) * b0;
            c.grad_elems = (grad_elems / batch) * nb;
        }
        return c;
    };

    if (!fused) {
        ctx->prof_pairs = gen_nb;
        float *warped = carve.take<float>(n0 * gen_nb);
        float *gen = carve.take<float>(lk_generic_scratch(rows, cols) * gen_nb);
        for (int b0 = 0; b0 < batch; b0 += kGenChunk)
            MICV_TRY(lk_chain_generic(ctx, plan, make_chain(s, b0, batch - b0 < kGenChunk ? batch - b0 : kGenChunk, b0 == 0), win, warped, gen));
        return MICV_OK;
    }

    // Fused path: the batch is split into groups of pairs whose chains run on forked HIP streams and
    // are joined on the caller's stream.  A chain's coarse levels are tiny, latency-bound launches;
    // beside another group's launches they fill CUs that would otherwise idle (0.713 -> 0.671 ms per
    // 8-pair step on MI355X).  Groups share nothing but the read-only pyramids.
    if (groups > 1) MICV_TRY(ctx->fork(s, groups - 1));
    int rc = MICV_OK;
    for (int g = 0; g < groups && rc == MICV_OK; g++) {
        const int b0 = (int)((long long)batch * g / groups);
        const int nb = (int)((long long)batch * (g + 1) / groups) - b0;
        if (g == 0) ctx->prof_pairs = nb;
        rc = lk_chain_fused(ctx, plan, make_chain(g == 0 ? s : ctx->aux_stream[g - 1], b0, nb, g == 0), win);
    }
    if (groups > 1) {  // always re-join, also after an error, so no stream is left dangling
        const int rcj = ctx->join(s, groups - 1);
        if (rc == MICV_OK) rc = rcj;
    }
    return rc;
}

}  // namespace micv

using namespace micv;

static int check_lk_args(const char *fn, const void *ctx, const void *a, const void *b,
                         const void *u, const void *v, int rows, int cols, size_t stride,
                         size_t ostride, int win) {
    MICV_REQUIRE(ctx && a && b && u && v, "%s: null argument", fn);
    MICV_REQUIRE(rows > 0 && cols > 0 && rows <= 32767 && cols <= 32767, "%s: bad size %dx%d", fn,
                 rows, cols);
    MICV_REQUIRE(stride_ok(stride, cols, 4) && stride_ok(ostride, cols, 4), "%s: bad stride", fn);
    MICV_REQUIRE(win >= 1 && win <= kMaxWin && (win & 1), "%s: window %d must be odd and <= %d", fn,
                 win, kMaxWin);
    return MICV_OK;
}

extern "C" {

int micv_lk_flow_pyr_batch_dev(micv_ctx *ctx, const float *prev, const float *next, int batch,
                               size_t pair_stride, int rows, int cols, size_t stride, int win,
                               int levels, float *u, float *v, size_t opair_stride,
                               size_t ostride, micv_stream stream) {
    MICV_TRY(check_lk_args("micv_lk_flow_pyr", ctx, prev, next, u, v, rows, cols, stride, ostride,
                           win));
    MICV_REQUIRE(batch >= 1 && batch <= 32767, "micv_lk_flow_pyr: bad batch %d", batch);
    MICV_REQUIRE(levels >= 1 && levels <= 16 && (rows >> (levels - 1)) > 0 &&
                     (cols >> (levels - 1)) > 0,
                 "micv_lk_flow_pyr: %d levels do not fit a %dx%d image", levels, rows, cols);
    MICV_REQUIRE(pair_stride % 4 == 0 && opair_stride % 4 == 0 &&
                     (batch == 1 || (pair_stride >= stride * (size_t)rows &&
                                     opair_stride >= ostride * (size_t)rows)),
                 "micv_lk_flow_pyr: bad pair stride");
    MICV_HIP(hipSetDevice(ctx->device));
    return lk_pyr_batch(ctx, static_cast<hipStream_t>(stream), prev, next, batch, pair_stride / 4,
                        rows, cols, (int)(stride / 4), win, levels, u, v, opair_stride / 4,
                        (int)(ostride / 4), !ctx->opt[MICV_OPT_LK_FORCE_GENERIC]);
}

int micv_lk_flow_pyr_dev(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                         size_t stride, int win, int levels, float *u, float *v, size_t ostride,
                         micv_stream stream) {
    return micv_lk_flow_pyr_batch_dev(ctx, prev, next, 1, 0, rows, cols, stride, win, levels, u, v,
                                      0, ostride, stream);
}

int micv_lk_flow_dev(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                     size_t stride, int win, float *u, float *v, size_t ostride,
                     micv_stream stream) {
    MICV_TRY(check_lk_args("micv_lk_flow", ctx, prev, next, u, v, rows, cols, stride, ostride, win));
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (lk_fused_supports(win) && !ctx->opt[MICV_OPT_LK_FORCE_GENERIC]) {
        LkLevelArgs a;
        a.rows = rows; a.cols = cols; a.batch = 1; a.win = win;
        a.prev = prev; a.next = next; a.img_stride = (int)(stride / 4); a.img_pair = 0;
        a.mode = LK_FLOW_NONE;
        a.flow_u = a.flow_v = nullptr; a.flow_rows = a.flow_cols = 0; a.flow_pair = 0;
        a.out_u = u; a.out_v = v; a.out_stride = (int)(ostride / 4); a.out_pair = 0;
        a.add_base = 0;
        a.row_begin = 0; a.row_end = rows;
        a.narrow = ctx->opt[MICV_OPT_LK_NARROW_TILES];
        a.ctx = ctx;
        a.max_chain = ctx->opt[MICV_OPT_LK_CHAIN];
        a.short_tiles = ctx->opt[MICV_OPT_LK_SHORT_TILES];
        a.stream_tiles = ctx->opt[MICV_OPT_LK_STREAM];
        a.tall_tiles = ctx->opt[MICV_OPT_LK_TALL_TILES];
        return launch_lk_level_fused(s, a);
    }
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need(lk_generic_scratch(rows, cols), 4), &scratch));
    return lk_level_generic(s, prev, (int)(stride / 4), next, (int)(stride / 4), rows, cols, win,
                            nullptr, nullptr, 0, u, v, (int)(ostride / 4),
                            static_cast<float *>(scratch), lk_generic_form(ctx));
}

int micv_lk_level_batch_dev(micv_ctx *ctx, const float *prev, const float *next, int batch,
                            size_t pair_stride, int rows, int cols, size_t stride, int win,
                            const float *flow_u, const float *flow_v, int flow_rows, int flow_cols,
                            size_t flow_pair_stride, int row_begin, int row_end, float *u, float *v,
                            size_t opair_stride, size_t ostride, micv_stream stream) {
    MICV_TRY(check_lk_args("micv_lk_level", ctx, prev, next, u, v, rows, cols, stride, ostride, win));
    MICV_REQUIRE(batch >= 1 && batch <= 32767, "micv_lk_level: bad batch %d", batch);
    MICV_REQUIRE((flow_u == nullptr) == (flow_v == nullptr), "micv_lk_level: give both flow fields or none");
    MICV_REQUIRE(!flow_u || (flow_rows > 0 && flow_cols > 0), "micv_lk_level: bad coarse flow size");
    MICV_REQUIRE(row_begin >= 0 && row_begin < row_end && row_end <= rows,
                 "micv_lk_level: bad row band [%d, %d)", row_begin, row_end);
    MICV_REQUIRE(pair_stride % 4 == 0 && opair_stride % 4 == 0 && flow_pair_stride % 4 == 0 &&
                     (batch == 1 || (pair_stride >= stride * (size_t)rows && opair_stride >= ostride * (size_t)rows &&
                                     (!flow_u || flow_pair_stride >= (size_t)flow_rows * flow_cols * 4))),
                 "micv_lk_level: bad pair stride");
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)rows * cols;
    const bool doubles = flow_u && 2 * flow_rows == rows && 2 * flow_cols == cols;
    const bool fused = lk_fused_supports(win) && !ctx->opt[MICV_OPT_LK_FORCE_GENERIC];
    const size_t pe = pair_stride / 4, ope = opair_stride / 4, fpe = flow_pair_stride / 4;
    if (fused) {
        // odd-sized levels: the expanded + resized base flow of every pair goes to scratch first
        float *bu = nullptr, *bv = nullptr;
        if (flow_u && !doubles) {
            void *scratch;
            MICV_TRY(ctx->reserve(Carver::need(n * batch, 4) * 2, &scratch));
            Carver carve(scratch);
            bu = carve.take<float>(n * batch);
            bv = carve.take<float>(n * batch);
        }
        LkLevelArgs a;
        a.rows = rows; a.cols = cols; a.batch = batch; a.win = win;
        a.prev = prev; a.next = next; a.img_stride = (int)(stride / 4); a.img_pair = pe;
        a.out_u = u; a.out_v = v; a.out_stride = (int)(ostride / 4); a.out_pair = ope;
        a.add_base = 1;
        a.row_begin = row_begin; a.row_end = row_end;
        a.narrow = ctx->opt[MICV_OPT_LK_NARROW_TILES];
        a.ctx = ctx;
        a.max_chain = ctx->opt[MICV_OPT_LK_CHAIN];
        a.short_tiles = ctx->opt[MICV_OPT_LK_SHORT_TILES];
        a.stream_tiles = ctx->opt[MICV_OPT_LK_STREAM];
        a.tall_tiles = ctx->opt[MICV_OPT_LK_TALL_TILES];
        a.flow_pair = 0;
        if (!flow_u) {
            a.mode = LK_FLOW_NONE;
            a.flow_u = a.flow_v = nullptr; a.flow_rows = a.flow_cols = 0;
        } else if (doubles) {
            a.mode = LK_FLOW_COARSE;
            a.flow_u = flow_u; a.flow_v = flow_v; a.flow_rows = flow_rows; a.flow_cols = flow_cols;
            a.flow_pair = fpe;
        } else {
            MICV_TRY(launch_flow_expand_resize(s, flow_u, flow_v, flow_rows, flow_cols, fpe, bu, bv, rows,
                                               cols, n, batch));
            a.mode = LK_FLOW_FULL;
            a.flow_u = bu; a.flow_v = bv; a.flow_rows = rows; a.flow_cols = cols;
            a.flow_pair = n;
        }
        return launch_lk_level_fused(s, a);
    }
    // generic kernels compute the whole level, pair by pair; rows outside the band are simply not needed
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need(n, 4) * 4 + Carver::need(lk_generic_scratch(rows, cols), 4), &scratch));
    Carver carve(scratch);
    float *bu = carve.take<float>(n), *bv = carve.take<float>(n);
    float *warped = carve.take<float>(n), *tmp = carve.take<float>(n);
    float *gen = carve.take<float>(lk_generic_scratch(rows, cols));
    for (int b = 0; b < batch; b++) {
        const float *pb = prev + b * pe, *nb = next + b * pe;
        const float *fub = flow_u ? flow_u + b * fpe : nullptr, *fvb = flow_v ? flow_v + b * fpe : nullptr;
        if (!flow_u) {
            MICV_HIP(hipMemsetAsync(bu, 0, n * 4, s));
            MICV_HIP(hipMemsetAsync(bv, 0, n * 4, s));
        } else if (doubles) {
            MICV_TRY(launch_pyr_up(s, fub, flow_rows, flow_cols, flow_cols, bu, cols, 2.f, tmp));
            MICV_TRY(launch_pyr_up(s, fvb, flow_rows, flow_cols, flow_cols, bv, cols, 2.f, tmp));
        } else {
            MICV_TRY(launch_flow_expand_resize(s, fub, fvb, flow_rows, flow_cols, 0, bu, bv, rows, cols, 0, 1));
        }
        MICV_TRY(launch_warp(s, nb, (int)(stride / 4), bu, bv, cols, rows, cols, warped, cols));
        MICV_TRY(lk_level_generic(s, pb, (int)(stride / 4), warped, cols, rows, cols, win, bu, bv, cols,
                                  u + b * ope, v + b * ope, (int)(ostride / 4), gen, lk_generic_form(ctx)));
    }
    return MICV_OK;
}

int micv_flow_bound_check_dev(micv_ctx *ctx, const float *v, int batch, size_t pair_stride, int rows, int cols,
                              size_t stride, int row_begin, int row_end, float bound, uint32_t *flag,
                              micv_stream stream) {
    MICV_REQUIRE(ctx && v && flag, "micv_flow_bound_check: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && batch >= 1 && batch <= 65535 && row_begin >= 0 && row_begin <= row_end && row_end <= rows,
                 "micv_flow_bound_check: bad size or row range");
    MICV_REQUIRE(stride_ok(stride, cols, 4) && pair_stride % 4 == 0 && (batch == 1 || pair_stride >= stride * (size_t)rows),
                 "micv_flow_bound_check: bad stride");
    MICV_REQUIRE(bound >= 0.f, "micv_flow_bound_check: negative bound");
    if (row_begin == row_end) return MICV_OK;
    MICV_HIP(hipSetDevice(ctx->device));
    flow_bound_kernel<<<dim3(cdiv(cols, 64), cdiv(row_end - row_begin, 4), batch), 256, 0, static_cast<hipStream_t>(stream)>>>(
        v, pair_stride / 4, (int)(stride / 4), cols, row_begin, row_end - row_begin, bound, flag);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_lk_level_kernel_name(micv_ctx *ctx, int win, int rows, int cols, int batch, char *buf, size_t cap) {
    MICV_REQUIRE(ctx && buf && cap >= 96, "micv_lk_level_kernel_name: null argument or a buffer under 96 bytes");
    MICV_REQUIRE(rows >= 2 && cols >= 2 && (rows & 1) == 0 && (cols & 1) == 0 && batch >= 1 && lk_fused_supports(win),
                 "micv_lk_level_kernel_name: even sizes, a batch and a window with a tiled kernel are expected");
    // the level launch of lk_chain_fused for a level with a doubling coarse flow, aligned dense frames -- with the
    // pointers never dereferenced: name_out makes launch_lk_level_fused stop where it would launch
    LkLevelArgs a;
    float *fake = reinterpret_cast<float *>(uintptr_t(4096));
    a.rows = rows; a.cols = cols; a.batch = batch; a.win = win;
    a.prev = a.next = fake; a.img_stride = cols; a.img_pair = (size_t)rows * cols;
    a.mode = LK_FLOW_COARSE;
    a.flow_u = a.flow_v = fake; a.flow_rows = rows / 2; a.flow_cols = cols / 2; a.flow_pair = (size_t)(rows / 2) * (cols / 2);
    a.out_u = a.out_v = fake; a.out_stride = cols; a.out_pair = (size_t)rows * cols;
    a.add_base = 1; a.row_begin = 0; a.row_end = rows;
    a.narrow = ctx->opt[MICV_OPT_LK_NARROW_TILES];
    a.ctx = ctx;
    a.max_chain = ctx->opt[MICV_OPT_LK_CHAIN];
    a.short_tiles = ctx->opt[MICV_OPT_LK_SHORT_TILES];
    a.stream_tiles = ctx->opt[MICV_OPT_LK_STREAM];
    a.tall_tiles = ctx->opt[MICV_OPT_LK_TALL_TILES];
    a.split = ctx->opt[MICV_OPT_LK_SPLIT];
    a.strip = ctx->opt[MICV_OPT_LK_STRIP];
    if (a.split > 0 && lk_split_supports(win)) {
        const LkGradGeom gg = lk_grad_geom(rows, cols, win);
        a.grad = fake; a.grad_pair = gg.pair_elems; a.grad_pitch = gg.pitch; a.grad_rows = gg.rows; a.grad_pad = gg.pad;
    }
    // (MICV_OPT_LK_DIRECT_LEVELS / MICV_OPT_LK_BUILD_OVERLAP change the staging of the COARSER levels only: level 0
    // always reads the frames themselves)
    buf[0] = 0;
    a.name_out = buf;
    a.name_cap = cap;
    MICV_HIP(hipSetDevice(ctx->device));
    return launch_lk_level_fused(nullptr, a);
}

int micv_lk_schedule_host(int rows, int cols, int batch, int win, int max_chain, int32_t *entries_xycp,
                          int64_t capacity, int64_t *count, int *tile_w, int *tile_h) {
    MICV_REQUIRE(count && tile_w && tile_h, "micv_lk_schedule_host: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && batch >= 1 && max_chain >= 1 && max_chain <= 32,
                 "micv_lk_schedule_host: bad argument");
    std::vector<int4> v;
    const int th = lk_schedule_host(rows, cols, batch, win, max_chain, &v);
    if (th == 0) {
        set_error("micv_lk_schedule_host: window %d has no scheduled launch", win);
        return MICV_EUNSUPPORTED;
    }
    *tile_w = 64;
    *tile_h = th;
    *count = (int64_t)v.size();
    if (entries_xycp)
        for (int64_t i = 0; i < (int64_t)v.size() && i < capacity; i++) {
            entries_xycp[4 * i] = v[i].x;
            entries_xycp[4 * i + 1] = v[i].y;
            entries_xycp[4 * i + 2] = v[i].z;
            entries_xycp[4 * i + 3] = v[i].w;
        }
    return MICV_OK;
}

int micv_lk_level_dev(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                      size_t stride, int win, const float *flow_u, const float *flow_v,
                      int flow_rows, int flow_cols, int row_begin, int row_end, float *u, float *v,
                      size_t ostride, micv_stream stream) {
    return micv_lk_level_batch_dev(ctx, prev, next, 1, 0, rows, cols, stride, win, flow_u, flow_v, flow_rows,
                                   flow_cols, 0, row_begin, row_end, u, v, 0, ostride, stream);
}

int micv_lk_warp_dev(micv_ctx *ctx, const float *src, size_t sstride, const float *du,
                     const float *dv, size_t fstride, int rows, int cols, float *dst,
                     size_t dstride, micv_stream stream) {
    MICV_REQUIRE(ctx && src && du && dv && dst, "micv_lk_warp: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && rows <= 32767 && cols <= 32767,
                 "micv_lk_warp: bad size %dx%d", rows, cols);
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(fstride, cols, 4) &&
                     stride_ok(dstride, cols, 4),
                 "micv_lk_warp: bad stride");
    MICV_REQUIRE(src != dst, "micv_lk_warp: src and dst must not alias");
    MICV_HIP(hipSetDevice(ctx->device));
    return launch_warp(static_cast<hipStream_t>(stream), src, (int)(sstride / 4), du, dv,
                       (int)(fstride / 4), rows, cols, dst, (int)(dstride / 4));
}

}  // extern "C"
