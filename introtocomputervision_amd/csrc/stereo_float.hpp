// stereo_float.hpp -- the float stereo search's tile body (SSD / serial:: / NCC on general f32 images), shared between
// stereo.hip (its own launch) and stereo_exact.hip (the exact-sum launch carries the float tiles of the same call as
// trailing workgroups, so that an 8-bit-valued pair costs no third launch).  See stereo.hip for the kernel's description.
#pragma once
#include <type_traits>

#include "kernels.hpp"
#include "ncc_arith.hpp"

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

// Rows per wave of the float search: 8, or 10 when that lets the whole grid be resident at once (4 waves/SIMD on
// 256 CUs = 4096 wave slots; 1080p r=5: 3888 waves instead of 4860 = one round, no tail).
inline bool stereo_rows10(int rows, int cols, int r, int force_rpw) {
    const long waves8 = (long)cdiv(cols, 64 - 2 * (r < 1 ? 1 : r)) * cdiv(rows, 8);
    const long waves10 = (long)cdiv(cols, 64 - 2 * (r < 1 ? 1 : r)) * cdiv(rows, 10);
    return force_rpw ? force_rpw == 10 : (waves8 > 4096 && (waves10 + 4095) / 4096 < (waves8 + 4095) / 4096);
}

// LDS budget of the staged right-image strip: DCH disparities per chunk -> SPAN columns per row.
constexpr int ST_DCH_DEFAULT = 64;

#ifndef MICV_STEREO_PF
#define MICV_STEREO_PF 2
#endif

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

}  // namespace micv
