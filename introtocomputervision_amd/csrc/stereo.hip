// stereo.hip -- ps2 window stereo: SSD (a12) and normalised cross-correlation (a13).
//
// Cost definition (DESIGN.md "Arithmetic contract", following DisparitySSD.cu:68-86):
//   colsum(y, xc, d) = sum over wy = -r..r (top -> bottom, float adds) of the per-pixel term
//   cost(y, x, d)    = sum over the window's columns (left -> right, float adds) of colsum
// with clamp-to-edge addressing of both images, every d in [minD, maxD] evaluated in ascending
// order, strict compare (lowest d wins ties).
//
// Kernel shape: one wave64 owns a strip of 64 window-columns x 8 (or 10) output rows.  The strip of
// `right` that a chunk of 64 disparities slides over (128 columns x the strip's rows) is staged
// once per chunk in wave-private LDS -- the disparity loop then reads it at lane + (d - d0):
// consecutive lanes, conflict-free, no global load in the loop.  A lane walks down its column
// keeping the last 2r+1 per-pixel terms in registers (fully unrolled, so the ring is static),
// re-adds them in the contract's order for every output row, and the horizontal sum runs as a
// systolic chain of v_add_f32 with a DPP wave_shr:1 operand:
//   acc <- shift_right_one_lane(acc) + colsum     (2r steps, unrolled, rows interleaved)
// which is exactly the left -> right association.  Running best cost / disparity live in
// registers; only the int8 disparity is written (9 B/px algorithmic).  VALU-bound.  Per wave and
// disparity at r = 5, 8 rows: the terms and column sums of two rows per v_pk_* instruction (67 instead of
// 116 for the scalar form), 80 DPP adds, 24 compare / selects: ~171 instructions, C3 0.225 ms measured vs
// 0.173 ms at one VALU instruction per cycle and CU (54 of a wave's 64 lanes produce outputs).  Whether the
// window has 2r + 1 or 2r columns is a template argument of the search loop: as a run-time flag it was a
// branch per row, and the rows' DPP chains then ran one after the other behind s_nop wait states.
//
// NCC adds fl(acc / fl(sqrt(AT * E))) per (row, disparity): 28 instructions through the compiler's sqrtf and
// division, 14 through the short exact sequences of ncc_arith.hpp, which a wave takes (per chunk of disparities)
// when every pixel it staged is 0 or of magnitude in [2^-8, 2^16] -- tracked while staging, four integer
// instructions per staged value.  C3 size: 0.62 -> 0.40 ms (2.24x -> 1.59x SSD).
#include "stereo_float.hpp"
#include "stereo_exact.hpp"

namespace micv {

// NCC: the energy of the right-image window depends on (row, last window column) only, not on the
// pair (x, d) that selects it, so it is summed once per position here -- same terms, same order as
// the search loop would (column sums top -> bottom from +0, then the systolic left -> right chain) --
// instead of once per (pixel, disparity).  One wave = 64 consecutive positions x RPW rows; the first
// 2R lanes only feed the chain.
template <int R, int RPW>
__global__ __launch_bounds__(256) void stereo_energy_kernel(StereoArgs a, float *__restrict__ E) {
    constexpr int W = 2 * R + 1, STEPS = RPW + 2 * R, OUTW = 64 - 2 * R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ys = blockIdx.y * (4 * RPW) + wave * RPW;
    if (ys >= a.rows || a.skip()) return;
    const bool full = a.wcols == W;
    const int sp = a.s_lo - 2 * R + blockIdx.x * OUTW + lane;  // position of this lane's column
    const int xr = clampi(sp, 0, a.cols - 1);
    float ringB[W], rvs[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++)  // every row's load first: between the stores below they waited one by one
        rvs[s] = a.right[(size_t)clampi(ys - R + s, 0, a.rows - 1) * a.stride + xr];
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
        const float rv = rvs[s];
        ringB[s % W] = rv * rv;
        if (s >= 2 * R) {
            float csb = 0.f;
#pragma unroll
            for (int k = 0; k < W; k++) csb += ringB[(s - 2 * R + k) % W];
            const float accb = systolic_sum<W>(csb, full);
            const int y = ys + s - 2 * R, e = sp - a.s_lo;
            if (lane >= 2 * R && y < a.rows && e < a.e_width) E[(size_t)y * a.e_width + e] = accb;
        }
    }
}

template <int R, int MODE, int RPW, int ST_DCH = ST_DCH_DEFAULT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == ST_NCC ? 1 : 3, MODE == ST_NCC ? 8 : 3)))
void stereo_kernel(StereoArgs a) {
    extern __shared__ float st_lds[];
    // (one workgroup per tile: a grid of resident workgroups walking the tiles made the SSD form 25 % slower -- 0.214 ->
    // 0.268 ms at C3 -- and the launch that only finds the flag and leaves no shorter: 4.6 us either way, r06)
    if (a.skip()) return;
    stereo_tile<R, MODE, RPW, ST_DCH>(a, st_lds, blockIdx.x, blockIdx.y);
}

// MICV_STEREO_ROLLING: the column sums as the CUDA kernels keep them (DisparitySSD.cu:97-138,
// DisparityNCorr.cu:117-173).  Rows are cut into strips of ROWS_PER_THREAD = 40 (DisparitySSD.cu:17);
// the first row of a strip sums its 2r+1 terms top -> bottom from 0, every further row takes the
// previous row's column sum, subtracts the term that left the window and then adds the one that
// entered (two roundings per row).  The chain is serial down the strip, so a wave owns a whole strip
// of 64 window columns: per disparity it walks the 40 rows once with the running sums in registers;
// the strip's best costs / disparities sit in wave-private LDS (40 x 64 lanes), horizontal sums run as
// the same systolic DPP chain as above.  A compatibility mode: images are read straight from global
// memory (clamp-to-edge = the reference's textures), any radius up to 31.
constexpr int ST_STRIP = 40, ST_ROLL_WAVES = 2;

template <bool NCC>
__global__ __launch_bounds__(64 * ST_ROLL_WAVES) void stereo_rolling_kernel(StereoArgs a, int r) {
    __shared__ float s_best[ST_ROLL_WAVES][ST_STRIP][64];
    __shared__ signed char s_bestd[ST_ROLL_WAVES][ST_STRIP][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int outw = 64 - 2 * r;
    const int seg = blockIdx.x * ST_ROLL_WAVES + wave;  // this wave's run of output columns
    const int y0 = blockIdx.y * ST_STRIP;
    if (seg * outw >= a.cols || a.skip()) return;  // whole wave; waves never synchronise with each other
    const int nr = a.rows - y0 < ST_STRIP ? a.rows - y0 : ST_STRIP;
    const int xc = seg * outw - r + lane;  // window column of this lane (unclamped)
    const int xl = clampi(xc, 0, a.cols - 1);
    const int xo = (a.wcols == 2 * r + 1) ? xc - r : xc - r + 1;  // output whose window ENDS at this lane
    const bool lane_ok = lane >= a.wcols - 1 && xo >= seg * outw && xo < (seg + 1) * outw && xo < a.cols;
    for (int j = 0; j < nr; j++) {
        s_best[wave][j][lane] = a.init_best;
        s_bestd[wave][j][lane] = -1;
    }
    auto sum_cols = [&](float cs) {
        float acc = cs;
        for (int k = 1; k < a.wcols; k++) acc = dpp_shr1(acc) + cs;
        return acc;
    };
    for (int d = a.min_d; d <= a.max_d; d++) {
        const int xr = clampi(xc + d, 0, a.cols - 1);
        float p = 0.f, aa = 0.f, bb = 0.f;
        auto term = [&](int y, bool add) {
            const int yy = clampi(y, 0, a.rows - 1);
            const float l = a.left[(size_t)yy * a.stride + xl], rv = a.right[(size_t)yy * a.stride + xr];
            if (NCC) {
                const float t0 = l * rv, t1 = l * l, t2 = rv * rv;
                p = add ? p + t0 : p - t0;
                aa = add ? aa + t1 : aa - t1;
                bb = add ? bb + t2 : bb - t2;
            } else {
                const float diff = l - rv, sq = diff * diff;
                p = add ? p + sq : p - sq;
            }
        };
        for (int wy = -r; wy <= r; wy++) term(y0 + wy, true);
        for (int j = 0; j < nr; j++) {
            if (j > 0) {
                term(y0 + j - 1 - r, false);
                term(y0 + j + r, true);
            }
            const float tot = sum_cols(p);
            float score;
            bool better;
            if (NCC) {
                const float at = sum_cols(aa), ai = sum_cols(bb);
                score = tot / sqrtf(at * ai);  // DisparityNCorr.cu:164
                better = score > s_best[wave][j][lane];
            } else {
                score = tot;
                better = score < s_best[wave][j][lane];  // DisparitySSD.cu:133
            }
            if (better) {
                s_best[wave][j][lane] = score;
                s_bestd[wave][j][lane] = (signed char)d;
            }
        }
    }
    if (lane_ok)
        for (int j = 0; j < nr; j++) a.disp[(size_t)(y0 + j) * a.dstride + xo] = (int8_t)s_bestd[wave][j][lane];
}

// Any radius: one thread per pixel, same arithmetic order, no reuse.
template <int MODE>
__global__ __launch_bounds__(256) void stereo_generic_kernel(StereoArgs a, int r) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.cols || y >= a.rows || a.skip()) return;
    const int x_first = x - r;  // first window column
    float best = MODE == ST_SSD_SERIAL ? 0.f : a.init_best;
    int besti = 99999999, bestd = MODE == ST_SSD_SERIAL ? 0 : -1;
    const int d_lo = MODE == ST_SSD_SERIAL ? -(x + r) : a.min_d;
    const int d_hi = MODE == ST_SSD_SERIAL ? a.cols - 1 + r - x : a.max_d;
    for (int d = a.min_d; d <= a.max_d; d++) {
        float tot = 0.f, totA = 0.f, totB = 0.f;
        int toti = 0;
        for (int i = 0; i < a.wcols; i++) {
            const int xl = clampi(x_first + i, 0, a.cols - 1);
            const int xr = clampi(x_first + i + d, 0, a.cols - 1);
            float cs = 0.f, csA = 0.f, csB = 0.f;
            int csi = 0;
            for (int wy = -r; wy <= r; wy++) {
                const int yy = clampi(y + wy, 0, a.rows - 1);
                const float l = a.left[(size_t)yy * a.stride + xl];
                const float rv = a.right[(size_t)yy * a.stride + xr];
                if (MODE == ST_NCC) {
                    cs += l * rv;
                    csA += l * l;
                    csB += rv * rv;
                } else {
                    const float diff = l - rv;
                    if (MODE == ST_SSD_SERIAL) csi += (int)roundf(diff * diff);
                    else cs += diff * diff;
                }
            }
            tot += cs; totA += csA; totB += csB; toti += csi;
        }
        if (MODE == ST_NCC) {
            const float nc = tot / sqrtf(totA * totB);
            if (nc > best) { best = nc; bestd = d; }
        } else if (MODE == ST_SSD_SERIAL) {
            if (d >= d_lo && d <= d_hi && toti < besti) { besti = toti; bestd = d; }
        } else if (tot < best) {
            best = tot;
            bestd = d;
        }
    }
    a.disp[(size_t)y * a.dstride + x] = (int8_t)bestd;
}

// NCC's chunk of disparities: 64 (r05).  r03 chose 32 to keep 3-4 workgroups per CU beside the staged energy rows; with the
// prefetched strips the kernel runs two waves per SIMD whatever the chunk, and half as many stagings win: 0.326 -> 0.316 ms at C3.
#ifndef MICV_NCC_DCH
#define MICV_NCC_DCH 64
#endif
template <int MODE>
static int launch_stereo(hipStream_t s, const StereoArgs &a, int r, int force_rpw) {
    const bool ten = stereo_rows10(a.rows, a.cols, r, force_rpw);
#define MICV_ST_LAUNCH(RR, RPW)                                                                    \
    do {                                                                                           \
        if (MODE == ST_NCC) {                                                                      \
            /*  the staged strips of a 64-disparity chunk pass 64 KB of dynamic LDS for 2 RPW + 2 R > 32 */ \
            /* (radius 9, 10; radius 7, 8 at 10 rows): those instantiations keep the 32-disparity chunk */          \
            constexpr int DCH = (2 * RPW + 2 * RR) * (64 + MICV_NCC_DCH) * 16 > 65536 ? 32 : MICV_NCC_DCH;           \
            stereo_energy_kernel<RR, RPW><<<dim3(cdiv(a.e_width, 64 - 2 * RR), cdiv(a.rows, 4 * RPW)), 256, 0, s>>>( \
                a, const_cast<float *>(a.energy));                                                 \
            stereo_kernel<RR, MODE, RPW, DCH><<<dim3(cdiv(a.cols, 64 - 2 * RR), cdiv(a.rows, 4 * RPW)), 256, \
                                              4 * (2 * RPW + 2 * RR) * (64 + DCH) * sizeof(float), s>>>(a); \
        } else {                                                                                   \
            stereo_kernel<RR, MODE, RPW><<<dim3(cdiv(a.cols, 64 - 2 * RR), cdiv(a.rows, 4 * RPW)), 256, \
                                           4 * (RPW + 2 * RR) * (64 + ST_DCH_DEFAULT) * sizeof(float), s>>>(a); \
        }                                                                                          \
    } while (0)
#define MICV_ST_CASE(RR)                                                                      \
    case RR:                                                                                  \
        if (ten) MICV_ST_LAUNCH(RR, 10); else MICV_ST_LAUNCH(RR, 8);                          \
        break;
    switch (r) {
        MICV_ST_CASE(1) MICV_ST_CASE(2) MICV_ST_CASE(3) MICV_ST_CASE(4) MICV_ST_CASE(5)
        MICV_ST_CASE(6) MICV_ST_CASE(7) MICV_ST_CASE(8) MICV_ST_CASE(9) MICV_ST_CASE(10)
        default:
            stereo_generic_kernel<MODE><<<dim3(cdiv(a.cols, 64), cdiv(a.rows, 4)), 256, 0, s>>>(a, r);
    }
#undef MICV_ST_LAUNCH
#undef MICV_ST_CASE
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

static int stereo_common(const char *fn, micv_ctx *ctx, const float *left, const float *right,
                         int rows, int cols, size_t stride, int rad, int min_d, int max_d,
                         int flags, int8_t *disp, size_t dstride, micv_stream stream, bool ncc) {
    MICV_REQUIRE(ctx && left && right && disp, "%s: null argument", fn);
    MICV_REQUIRE(rows > 0 && cols > 0, "%s: bad size %dx%d", fn, rows, cols);
    MICV_REQUIRE(rad >= 0 && rad <= 31, "%s: window radius %d out of range 0..31", fn, rad);
    MICV_REQUIRE(min_d <= max_d && min_d >= -128 && max_d <= 127,
                 "%s: disparities [%d, %d] do not fit the int8 output (CV_8SC1)", fn, min_d, max_d);
    MICV_REQUIRE(stride_ok(stride, cols, 4) && dstride >= (size_t)cols, "%s: bad stride", fn);
    MICV_REQUIRE((flags & ~15) == 0, "%s: unknown flags 0x%x", fn, flags);
    MICV_REQUIRE(!(ncc && (flags & MICV_STEREO_SERIAL)), "%s: SERIAL applies to SSD only", fn);
    MICV_REQUIRE(!((flags & MICV_STEREO_SERIAL) && (flags & MICV_STEREO_ROLLING)),
                 "%s: SERIAL and ROLLING describe different reference functions", fn);
    MICV_REQUIRE(!((flags & MICV_STEREO_COLS_2R) && rad == 0), "%s: COLS_2R needs radius >= 1", fn);
    MICV_HIP(hipSetDevice(ctx->device));
    StereoArgs a;
    a.left = left; a.right = right; a.stride = (int)(stride / 4);
    a.rows = rows; a.cols = cols; a.min_d = min_d; a.max_d = max_d;
    a.wcols = (flags & MICV_STEREO_COLS_2R) ? 2 * rad : 2 * rad + 1;
    a.init_best = ncc ? 0.f : ((flags & MICV_STEREO_MIN_SSD_5E6) ? 5000000.f : INFINITY);
    a.disp = disp; a.dstride = (int)dstride;
    a.energy = nullptr; a.e_width = 0; a.s_lo = 0;
    a.fallback_flag = nullptr; a.epoch = 0;
    size_t energy_bytes = 0;
    if (ncc && rad >= 1 && rad <= 10 && !(flags & MICV_STEREO_ROLLING)) {
        // positions a window's last column can take: lanes reach from -R to past cols + R (whole
        // 64-lane strips), shifted by every disparity
        const int outw = 64 - 2 * rad;
        a.s_lo = -rad + min_d;
        a.e_width = (int)cdiv(cols, outw) * outw + 64 + (max_d - min_d);
        energy_bytes = Carver::need((size_t)rows * a.e_width, 4);
    }
    // 8-bit-valued images (every plain ps2 call, main.cpp:87-88): the exact-sum kernels go first and the kernels below
    // return at once unless the pack pre-pass found a pixel that is not an integer in 0..255 (no host round trip).
    const bool exact = ctx->opt[MICV_OPT_STEREO_EXACT] >= 0 && stereo_exact_covers(rad, flags, ncc);
    const size_t exact_bytes = exact ? stereo_exact_scratch(rows, cols, rad, min_d, max_d, a.wcols, flags, ctx->wave_slots(3)) : 0;
    void *scratch = nullptr;
    if (energy_bytes + exact_bytes) MICV_TRY(ctx->reserve(energy_bytes + exact_bytes, &scratch));
    if (energy_bytes) a.energy = static_cast<const float *>(scratch);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (exact) {
        unsigned *flag;
        MICV_TRY(ctx->stereo_flag_word(&flag));
        a.fallback_flag = flag;
        if (++ctx->stereo_epoch == 0) ++ctx->stereo_epoch;  // (0 is what the flag word holds before any call)
        a.epoch = ctx->stereo_epoch;
        // (the float search of the same call rides in the exact-sum launch as trailing workgroups: nothing more to launch)
        return stereo_exact_launch(s, static_cast<char *>(scratch) + energy_bytes, left, right, rows, cols, a.stride, rad, min_d,
                                   max_d, flags, a.wcols, disp, a.dstride, flag, a.epoch, ctx->wave_slots(3), a,
                                   stereo_rows10(rows, cols, rad, ctx->opt[MICV_OPT_STEREO_ROWS]));
    }
    if (flags & MICV_STEREO_ROLLING) {
        const dim3 grid(cdiv(cdiv(cols, 64 - 2 * rad), ST_ROLL_WAVES), cdiv(rows, ST_STRIP));
        if (ncc) stereo_rolling_kernel<true><<<grid, 64 * ST_ROLL_WAVES, 0, s>>>(a, rad);
        else stereo_rolling_kernel<false><<<grid, 64 * ST_ROLL_WAVES, 0, s>>>(a, rad);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    const int rpw = ctx->opt[MICV_OPT_STEREO_ROWS];
    if (ncc) return launch_stereo<ST_NCC>(s, a, rad, rpw);
    if (flags & MICV_STEREO_SERIAL) return launch_stereo<ST_SSD_SERIAL>(s, a, rad, rpw);
    return launch_stereo<ST_SSD>(s, a, rad, rpw);
}

}  // namespace micv

using namespace micv;

extern "C" {

int micv_disparity_ssd_dev(micv_ctx *ctx, const float *left, const float *right, int rows,
                           int cols, size_t stride, int window_rad, int min_disparity,
                           int max_disparity, int flags, int8_t *disp, size_t dstride,
                           micv_stream stream) {
    return stereo_common("micv_disparity_ssd", ctx, left, right, rows, cols, stride, window_rad,
                         min_disparity, max_disparity, flags, disp, dstride, stream, false);
}

int micv_disparity_ncorr_dev(micv_ctx *ctx, const float *left, const float *right, int rows,
                             int cols, size_t stride, int window_rad, int min_disparity,
                             int max_disparity, int flags, int8_t *disp, size_t dstride,
                             micv_stream stream) {
    return stereo_common("micv_disparity_ncorr", ctx, left, right, rows, cols, stride, window_rad,
                         min_disparity, max_disparity, flags, disp, dstride, stream, true);
}

}  // extern "C"
