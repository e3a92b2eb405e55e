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
#include <type_traits>

#include "kernels.hpp"
#include "ncc_arith.hpp"
#include "stereo_exact.hpp"

namespace micv {

enum { ST_SSD = 0, ST_SSD_SERIAL = 1, ST_NCC = 2 };

typedef float st_v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dpp_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ int dpp_shr1(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false);
}

// Horizontal window sum, left -> right: acc <- shift_right_one_lane(acc) + colsum, (wcols - 1)
// times.  wcols is W or W - 1 (COLS_2R); both chains are fully unrolled so the scheduler can
// interleave the chains of different rows (a DPP read of a just-written VGPR costs wait states).
template <int W, typename T>
__device__ __forceinline__ T systolic_sum(T cs, bool full) {
    T acc = cs;
#pragma unroll
    for (int k = 1; k < W - 1; k++) acc = dpp_shr1(acc) + cs;
    if (full && W > 1) acc = dpp_shr1(acc) + cs;
    return acc;
}

struct StereoArgs {
    const float *left, *right;
    int stride, rows, cols, min_d, max_d;
    int wcols;        // 2r+1, or 2r with MICV_STEREO_COLS_2R
    float init_best;  // +inf, or 5e6 with MICV_STEREO_MIN_SSD_5E6 (SSD); 0 for NCC
    int8_t *disp;
    int dstride;
    // NCC: window energy of `right`, E[y][s - s_lo] = sum over the window whose LAST column is
    // (unclamped) column s, every column clamped on its own -- written by stereo_energy_kernel
    const float *energy;
    int e_width, s_lo;
    // The exact-sum kernels of stereo_exact.hip were launched in front for this call: they did the work unless the
    // flag word holds `epoch` (an image is not 8-bit-valued) -- only then do the kernels of this file run.
    const unsigned *fallback_flag;
    unsigned epoch;
    __device__ bool skip() const { return fallback_flag && __builtin_nontemporal_load(fallback_flag) != epoch; }
};

// LDS budget of the staged right-image strip: DCH disparities per chunk -> SPAN columns per row.
constexpr int ST_DCH_DEFAULT = 64;

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

template <int R, int MODE, int RPW, int ST_DCH>
// Register budget: LDS allows three workgroups per CU = three waves per SIMD, so the SSD forms may take 170 VGPRs instead
// of the 128 the default heuristic aims at (r05 A/B on one box: 0.2246-0.2267 -> 0.2176-0.2208 ms at C3); NCC with its
// prefetched strips keeps the default budget -- it then takes 243 VGPRs = two waves per SIMD, 0.326 ms; capped at 170 it spills
// (0.389 ms), told "two waves" it allocates 179 and schedules worse (0.417 ms).
__device__ __forceinline__ void stereo_tile(const StereoArgs &a, float *st_lds, const int bx, const int by) {
    constexpr int W = 2 * R + 1, STEPS = RPW + 2 * R, OUTW = 64 - 2 * R, ST_SPAN = 64 + ST_DCH;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar row addressing
    const bool full = a.wcols == W;
    const int ys = by * (4 * RPW) + wave * RPW;
    if (ys >= a.rows) return;  // whole wave; waves never synchronise with each other
    constexpr int ESTEPS = MODE == ST_NCC ? RPW : 0;  // staged rows of the window-energy field
    float *Rs = st_lds + wave * ((STEPS + ESTEPS) * ST_SPAN);
    float *Es = Rs + STEPS * ST_SPAN;
    const int x_base = bx * OUTW - R;
    const int xc = x_base + lane;  // window column of this lane (unclamped)
    const int xl = clampi(xc, 0, a.cols - 1);
    // output pixel whose window ENDS at this lane
    const int xo = (a.wcols == W) ? xc - R : xc - R + 1;
    const bool lane_ok = lane >= a.wcols - 1 && xo >= bx * OUTW &&
                         xo < (bx + 1) * OUTW && xo < a.cols;

    float Lv[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
        const int yy = clampi(ys - R + s, 0, a.rows - 1);
        Lv[s] = a.left[(size_t)yy * a.stride + xl];
    }

    // NCC: operands in the checked range take the short exact sqrt / division (ncc_arith.hpp)
    NccRange pix_l;
    if (MODE == ST_NCC) {
#pragma unroll
        for (int s = 0; s < STEPS; s++) pix_l.add(Lv[s]);
    }

    using acc_t = typename std::conditional<MODE == ST_SSD_SERIAL, int, float>::type;
    acc_t best[RPW];
    int bestd[RPW];
#pragma unroll
    for (int j = 0; j < RPW; j++) {
        best[j] = MODE == ST_SSD_SERIAL ? (acc_t)99999999 : (acc_t)a.init_best;
        bestd[j] = MODE == ST_SSD_SERIAL ? 0 : -1;  // DisparitySSD.cpp:37-38 / .cu:177
    }
    // NCC: the template's own energy does not depend on d -- sum it once.
    float AT[RPW];
    if (MODE == ST_NCC) {
        float ring[W];
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            ring[s % W] = Lv[s] * Lv[s];
            if (s >= 2 * R) {
                float cs = 0.f;
#pragma unroll
                for (int k = 0; k < W; k++) cs += ring[(s - 2 * R + k) % W];
                AT[s - 2 * R] = systolic_sum<W>(cs, full);
            }
        }
    }
    // serial:: search range of this output pixel (DisparitySSD.cpp:42-43), padded coords.
    const int d_lo = MODE == ST_SSD_SERIAL ? -(xo + R) : a.min_d;
    const int d_hi = MODE == ST_SSD_SERIAL ? a.cols - 1 + R - xo : a.max_d;

    // The strip of `right` (and, for NCC, of the window-energy field) a chunk of disparities slides over is PREFETCHED:
    // its global loads are issued into registers before the previous chunk's search loop and written to LDS after it,
    // so the memory round trip runs under ~8 k instructions of arithmetic instead of in front of them (r05: the wave has
    // nothing else to overlap it with -- SQ_WAIT_ANY was 35 % of the NCC kernel's wave cycles, profiles/r05/stereo_ncc.txt).
    // NCC only (0.388 -> 0.326 ms at C3); SSD, whose chunks are twice as long and whose loads are a third of NCC's per
    // disparity, LOSES with it (0.2265 -> 0.265 ms: 40 more live registers), so it keeps staging in front of the loop.
#ifndef MICV_STEREO_PF
#define MICV_STEREO_PF 2
#endif
    constexpr int PF = MODE == ST_NCC ? MICV_STEREO_PF : 0;  // 0 = no prefetch, 1 = the right strip, 2 = + the energy strip
    static_assert(MODE != ST_NCC || PF == 2, "the NCC staging below reads the prefetched energy strip (pre_e)");
    constexpr int NH = (ST_SPAN + 63) / 64;
    float pre_r[PF >= 1 ? STEPS : 1][NH], pre_e[PF >= 2 && ESTEPS > 0 ? ESTEPS : 1][NH];
    auto load_r = [&](int d0, int s, int h) {
        const int yy = clampi(ys - R + s, 0, a.rows - 1), i = lane + 64 * h;
        return a.right[(size_t)yy * a.stride + clampi(x_base + d0 + (i < ST_SPAN ? i : ST_SPAN - 1), 0, a.cols - 1)];
    };
    auto load_e = [&](int d0, int j, int h) {
        const int i = lane + 64 * h;
        return a.energy[(size_t)(ys + j < a.rows ? ys + j : a.rows - 1) * a.e_width +
                        clampi(x_base + d0 + (i < ST_SPAN ? i : ST_SPAN - 1) - a.s_lo, 0, a.e_width - 1)];
    };
    auto prefetch = [&](int d0) {
        if constexpr (PF >= 1) {
#pragma unroll
            for (int s = 0; s < STEPS; s++)
#pragma unroll
                for (int h = 0; h < NH; h++) pre_r[s][h] = load_r(d0, s, h);
        }
        if constexpr (PF >= 2 && MODE == ST_NCC) {
#pragma unroll
            for (int j = 0; j < ESTEPS; j++)
#pragma unroll
                for (int h = 0; h < NH; h++) pre_e[j][h] = load_e(d0, j, h);
        }
    };
    if (PF > 0) prefetch(a.min_d);
    for (int d0 = a.min_d; d0 <= a.max_d; d0 += ST_DCH) {
        // Stage the strip of `right` this chunk of disparities slides over: column i of the strip
        // is image column clamp(x_base + d0 + i) (clamp-to-edge), rows as for Lv.  Every later
        // read is an LDS read at lane + (d - d0): consecutive lanes, conflict-free.
        __builtin_amdgcn_wave_barrier();  // the previous chunk's reads are done (in-order LDS)
        NccRange pix = pix_l, en;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
#pragma unroll
            for (int h = 0; h < NH; h++) {
                const int i = lane + 64 * h;
                if (ST_SPAN % 64 == 0 || i < ST_SPAN) {
                    const float v = PF >= 1 ? pre_r[PF >= 1 ? s : 0][h] : load_r(d0, s, h);
                    Rs[s * ST_SPAN + i] = v;
                    if (MODE == ST_NCC) pix.add(v);
                }
            }
        }
        if (MODE == ST_NCC) {
#pragma unroll
            for (int j = 0; j < RPW; j++) {
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const int i = lane + 64 * h;
                    if (ST_SPAN % 64 == 0 || i < ST_SPAN) {
                        const float v = pre_e[j][h];
                        Es[j * ST_SPAN + i] = v;
                        en.add(v);
                    }
                }
            }
        }
        if (PF > 0 && d0 + ST_DCH <= a.max_d) prefetch(d0 + ST_DCH);  // the next chunk's loads fly under this chunk's search
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int d1 = d0 + ST_DCH - 1 < a.max_d ? d0 + ST_DCH - 1 : a.max_d;
        // (both switches are wave-uniform and loop-invariant; as template arguments the loop body has no branch,
        // so the rows' DPP chains interleave instead of running one after the other behind wait states)
        auto search = [&](auto short_arith, auto full_window) {
        constexpr bool SHORT = decltype(short_arith)::value, FULLW = decltype(full_window)::value;
        for (int d = d0; d <= d1; d++) {
            const float *rcol = Rs + lane + (d - d0);
            const bool d_ok = MODE != ST_SSD_SERIAL || (d >= d_lo && d <= d_hi);
            if constexpr (MODE != ST_SSD_SERIAL) {
                // Two output rows (j, j + 1), j even, per v_pk_add_f32: their column sums add the terms of rows
                // j + k and j + 1 + k at step k, i.e. the pair X[m] = (term[m], term[m + 1]), m = j + k.  Even m is
                // the pair the packed subtract / multiply produced (E), odd m one v_pk_mov_b32 away (O).  Each
                // half is the same top -> bottom chain as before; the horizontal chains stay scalar (DPP).
                static_assert(STEPS % 2 == 0 && RPW % 2 == 0, "rows in pairs");
                st_v2f E[STEPS / 2], O[STEPS / 2];
#pragma unroll
                for (int t = 0; t < STEPS / 2; t++) {
                    const st_v2f rv = (st_v2f){rcol[(2 * t) * ST_SPAN], rcol[(2 * t + 1) * ST_SPAN]};
                    const st_v2f lv = (st_v2f){Lv[2 * t], Lv[2 * t + 1]};
                    if (MODE == ST_NCC) {
                        E[t] = lv * rv;
                    } else {
                        const st_v2f diff = lv - rv;
                        E[t] = diff * diff;
                    }
                    if (t > 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(O[t - 1]) : "v"(E[t - 1]), "v"(E[t]));
                    if (2 * t >= 2 * R) {
                        const int j = 2 * t - 2 * R;
                        st_v2f cs2 = E[j / 2];  // (no 0 + x: see the scalar loop below)
#pragma unroll
                        for (int k = 1; k < W; k++) cs2 += ((j + k) & 1) ? O[(j + k) / 2] : E[(j + k) / 2];
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const float acc = systolic_sum<W>(cs2[h], FULLW);
                            if (MODE == ST_NCC) {
                                const float accb = Es[(j + h) * ST_SPAN + lane + (d - d0)];  // window energy of `right`
                                const float pr = AT[j + h] * accb;
                                // DisparityNCorr.cu:106
                                const float nc = SHORT ? ncc_div(acc, ncc_sqrt(pr)) : acc / sqrtf(pr);
                                if (nc > (float)best[j + h]) {  // :108
                                    best[j + h] = (acc_t)nc;
                                    bestd[j + h] = d;
                                }
                            } else if (acc < (float)best[j + h]) {  // DisparitySSD.cu:88
                                best[j + h] = (acc_t)acc;
                                bestd[j + h] = d;
                            }
                        }
                    }
                }
                continue;
            }
            acc_t ring[W];
#pragma unroll
            for (int s = 0; s < STEPS; s++) {
                const float rv = rcol[s * ST_SPAN];
                if (MODE == ST_NCC) {
                    ring[s % W] = (acc_t)(Lv[s] * rv);
                } else {
                    const float diff = Lv[s] - rv;
                    const float sq = diff * diff;
                    ring[s % W] = MODE == ST_SSD_SERIAL ? (acc_t)(int)roundf(sq) : (acc_t)sq;
                }
                if (s >= 2 * R) {
                    const int j = s - 2 * R;
                    // 0 + x == x bit for bit when x is never -0 (x = diff^2, or an int): skip that add.  A product
                    // of NCC can be -0, but the column sum then differs (as -0 for +0) only when every term is -0,
                    // the window sum only when every column sum is: a correlation of -0 for +0 (or NaN both
                    // ways), and `nc > best` is false for either since best starts at 0 and only grows.
                    acc_t cs = ring[(s - 2 * R) % W];
#pragma unroll
                    for (int k = 1; k < W; k++) cs += ring[(s - 2 * R + k) % W];
                    const acc_t acc = systolic_sum<W>(cs, FULLW);
                    if (MODE == ST_NCC) {
                        const float accb = Es[j * ST_SPAN + lane + (d - d0)];  // window energy of `right`
                        const float pr = AT[j] * accb;
                        // DisparityNCorr.cu:106
                        const float nc = SHORT ? ncc_div((float)acc, ncc_sqrt(pr)) : (float)acc / sqrtf(pr);
                        if (nc > (float)best[j]) {                          // :108
                            best[j] = (acc_t)nc;
                            bestd[j] = d;
                        }
                    } else if (d_ok && acc < best[j]) {  // DisparitySSD.cu:88 / .cpp:54
                        best[j] = acc;
                        bestd[j] = d;
                    }
                }
            }
        }
        };
        auto search_w = [&](auto short_arith) {
            if (full)
                search(short_arith, std::true_type{});
            else
                search(short_arith, std::false_type{});
        };
        if constexpr (MODE == ST_NCC) {
            // wave-uniform: every staged operand of this chunk is in the checked range
            if (__builtin_amdgcn_ballot_w64(!(pix.inside(NCC_PIX_LO, NCC_PIX_HI) && en.inside(NCC_EN_LO, NCC_EN_HI))) == 0)
                search_w(std::true_type{});
            else
                search_w(std::false_type{});
        } else {
            search_w(std::false_type{});
        }
    }
    if (lane_ok) {
#pragma unroll
        for (int j = 0; j < RPW; j++)
            if (ys + j < a.rows) a.disp[(size_t)(ys + j) * a.dstride + xo] = (int8_t)bestd[j];
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
    // Rows per wave: 8, or 10 when that lets the whole grid be resident at once (4 waves/SIMD on
    // 256 CUs = 4096 wave slots; 1080p r=5: 3888 waves instead of 4860 = one round, no tail).
    const long waves8 = (long)cdiv(a.cols, 64 - 2 * (r < 1 ? 1 : r)) * cdiv(a.rows, 8);
    const long waves10 = (long)cdiv(a.cols, 64 - 2 * (r < 1 ? 1 : r)) * cdiv(a.rows, 10);
    const bool ten = force_rpw ? force_rpw == 10 : (waves8 > 4096 && (waves10 + 4095) / 4096 < (waves8 + 4095) / 4096);
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
        MICV_TRY(stereo_exact_launch(s, static_cast<char *>(scratch) + energy_bytes, left, right, rows, cols, a.stride, rad,
                                     min_d, max_d, flags, a.wcols, disp, a.dstride, flag, a.epoch, ctx->wave_slots(3)));
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
