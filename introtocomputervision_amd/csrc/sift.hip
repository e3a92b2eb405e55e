// sift.hip -- the SIFT-style descriptor window (gfx950): 4 x 4 spatial x 8 orientation bins around
// each keypoint, sampled on the harris::getGradients fields.
//
// Reference call site: Solution::siftHelper, ps4_cpp/src/Solution.cpp:166-169
// (cv::xfeatures2d::SIFT::compute on the keypoints of sift::getKeypoints, Descriptors.cpp:27-47).
// OpenCV's SIFT is third-party code that is not part of the reference tree; DESIGN.md section 2 states
// the arithmetic this kernel and the CPU checker agree on bit for bit: the published
// calcSIFTDescriptor algorithm (window rotated by the keypoint angle, bin width 3 * size / 2 px,
// Gaussian weight, trilinear distribution, normalise -> clamp 0.2 -> renormalise to 512 -> 8 bits)
// with fixed polynomial cos / sin / exp, cv::fastAtan2's polynomial, and a histogram accumulated in
// 64-bit fixed point (2^-40 of a bound on the window's gradient magnitudes) so that the result does
// not depend on the order in which the samples arrive.
//
// One wave64 per keypoint, four keypoints per workgroup.  The window (107 x 107 samples for the
// reference's size-10 keypoints) is swept twice by the wave's lanes with coalesced row reads
// (second sweep served by L1/L2): sweep 1 finds the magnitude bound (a plain max of |gx|, |gy| over
// the bounding square) by a wave max-reduction, sweep
// 2 scatters the eight trilinear shares of every sample into the wave's LDS histogram with native
// 64-bit LDS atomics.  The 128-term norms are summed left to right by one lane (the contract).
#include <cfloat>

#include "common.hpp"

namespace micv {

constexpr int SD = 4, SN = 8;
// The wave's histogram holds the 4 x 4 cells the descriptor keeps, nine orientation slots each (slot 8 = the upper
// share of bin 7, folded onto bin 0 at the end): shares that fall on the ring of cells around the 4 x 4 grid are dropped
// by the published algorithm when it copies the histogram out, so they are not added in the first place (36 % of all
// shares), and a copy is 1 152 bytes instead of the 2 880 of the full 6 x 6 x 10 array.
constexpr int SHIST = SD * SD * (SN + 1);
// Private copies of the histogram per wave, picked by the lane's position within its 4x4 block: lanes
// that hit the same bin in the same instruction serialise in the LDS atomic unit (flat regions and
// straight edges send a whole block to one bin), and COPIES copies divide that by COPIES.
#ifndef MICV_SIFT_COPIES
#define MICV_SIFT_COPIES 4
#endif
constexpr int COPIES = MICV_SIFT_COPIES;

// sin / cos of `deg` degrees: quadrant by float arithmetic, Taylor polynomials as fmaf chains.
__device__ __forceinline__ void sincos_deg(float deg, float &s, float &c) {
    float t = deg / 360.f;
    t = t - floorf(t);
    const float x = t * 4.f;
    int q = (int)x;
    const float f = x - (float)q;
    q &= 3;
    const float y = f * 1.57079632679489662f, y2 = y * y;
    float ps = -2.50521083854417188e-8f;
    ps = fmaf(ps, y2, 2.75573192239858907e-6f);
    ps = fmaf(ps, y2, -1.98412698412698413e-4f);
    ps = fmaf(ps, y2, 8.33333333333333333e-3f);
    ps = fmaf(ps, y2, -1.66666666666666667e-1f);
    ps = fmaf(ps, y2, 1.f);
    const float sy = ps * y;
    float pc = 2.08767569878680990e-9f;
    pc = fmaf(pc, y2, -2.75573192239858907e-7f);
    pc = fmaf(pc, y2, 2.48015873015873016e-5f);
    pc = fmaf(pc, y2, -1.38888888888888889e-3f);
    pc = fmaf(pc, y2, 4.16666666666666667e-2f);
    pc = fmaf(pc, y2, -0.5f);
    pc = fmaf(pc, y2, 1.f);
    switch (q) {
        case 0: s = sy; c = pc; break;
        case 1: s = pc; c = -sy; break;
        case 2: s = -sy; c = -pc; break;
        default: s = -pc; c = sy; break;
    }
}

// exp(w), w <= 0: 2^k * P(f), k = rint(w log2 e).
__device__ __forceinline__ float exp_neg(float w) {
    if (w < -80.f) return 0.f;
    const float t = w * 1.44269504088896341f;
    const float k = rintf(t);
    const float f = t - k;
    float p = 1.52527338040598403e-5f;
    p = fmaf(p, f, 1.54035303933816099e-4f);
    p = fmaf(p, f, 1.33335581464284434e-3f);
    p = fmaf(p, f, 9.61812910762847716e-3f);
    p = fmaf(p, f, 5.55041086648215800e-2f);
    p = fmaf(p, f, 2.40226506959100712e-1f);
    p = fmaf(p, f, 6.93147180559945309e-1f);
    p = fmaf(p, f, 1.f);
    return ldexpf(p, (int)k);
}

// cv::fastAtan2's polynomial, degrees in [0, 360).
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * 57.29577951308232f, p3 = -0.3258083974640975f * 57.29577951308232f,
                p5 = 0.1555786518463281f * 57.29577951308232f, p7 = -0.04432655554792128f * 57.29577951308232f;
    const float ax = fabsf(x), ay = fabsf(y);
    // one quotient for both octants (the operands are selected, not the results: a divergent branch would run the IEEE
    // division and the polynomial twice per wave) -- the same operations on the same values as the two-branch form
    const bool flat = ax >= ay;
    const float c = (flat ? ay : ax) / ((flat ? ax : ay) + (float)DBL_EPSILON);
    const float c2 = c * c;
    const float p = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    float a = flat ? p : 90.f - p;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

struct SiftGeom {
    float cos_t, sin_t, ori;
    int radius, px, py;
    bool valid;
};

__device__ __forceinline__ SiftGeom sift_geometry(const float *__restrict__ kp, int rows, int cols) {
    SiftGeom g;
    const float x = kp[0], y = kp[1], size = kp[2], angle = kp[3];
    g.valid = size > 0.f && isfinite(size) && isfinite(x) && isfinite(y) && isfinite(angle) &&
              fabsf(x) < 1e9f && fabsf(y) < 1e9f;
    g.px = g.valid ? (int)lrintf(x) : 0;
    g.py = g.valid ? (int)lrintf(y) : 0;
    float ori = 360.f - angle;
    ori = ori - 360.f * floorf(ori / 360.f);  // keypoint angles outside [0, 360) are reduced first
    if (!(ori < 360.f)) ori = 0.f;
    g.ori = ori;
    const float hist_width = 3.f * (size * 0.5f);
    const float rf = hist_width * 1.4142135623730951f * (float)(SD + 1) * 0.5f;
    const int diag = (int)lrint(sqrt((double)cols * cols + (double)rows * rows));
    int radius = rf < (float)diag ? (int)lrintf(rf) : diag;
    g.radius = radius < 0 ? 0 : (radius > diag ? diag : radius);
    float s, c;
    sincos_deg(ori, s, c);
    g.cos_t = c / hist_width;
    g.sin_t = s / hist_width;
    return g;
}

// Window geometry of sample (i, j): rotated bin coordinates, the Gaussian exponent, and whether the
// sample contributes at all (inside the 4x4 grid's reach and the image interior).
__device__ __forceinline__ bool sift_test(int rows, int cols, const SiftGeom &g, int i, int j, float &rbin,
                                          float &cbin, float &wexp) {
    const float c_rot = (float)j * g.cos_t - (float)i * g.sin_t;
    const float r_rot = (float)j * g.sin_t + (float)i * g.cos_t;
    rbin = r_rot + (float)(SD / 2) - 0.5f;
    cbin = c_rot + (float)(SD / 2) - 0.5f;
    const int r = g.py + i, c = g.px + j;
    wexp = (c_rot * c_rot + r_rot * r_rot) * (-1.f / ((float)(SD * SD) * 0.5f));
    return rbin > -1.f && rbin < (float)SD && cbin > -1.f && cbin < (float)SD && r > 0 && r < rows - 1 && c > 0 &&
           c < cols - 1;
}

// Waves of one keypoint meet at a workgroup barrier; a keypoint that owns a single wave only needs
// its own LDS traffic ordered.
template <int WPK>
__device__ __forceinline__ void keypoint_sync() {
    if constexpr (WPK > 1) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// WPK waves per keypoint: 1 when there are enough keypoints to fill the chip (4 keypoints per
// workgroup), 2 in between, 4 when there are few (one keypoint per workgroup, the window's passes dealt round-robin
// to the waves, each wave adding into its own histogram copies) -- a lone wave takes ~0.13 ms for a
// 107x107 window, which is all the latency a small keypoint list would ever see.
template <int WPK>
__global__ __launch_bounds__(256) void sift_descriptor_kernel(const float *__restrict__ gx,
                                                               const float *__restrict__ gy, int gstride,
                                                               int rows, int cols,
                                                               const float *__restrict__ kps, long long n,
                                                               float *__restrict__ desc, int dstride) {
    constexpr int G = 4 / WPK;  // keypoints per workgroup
    constexpr int LEN = SD * SD * SN;
    __shared__ unsigned long long hist_all[4][COPIES * SHIST];
    // per wave: the queue of sweep 2; after it (and a keypoint_sync) the first G of these are the keypoints' float rows
    __shared__ unsigned scratch_all[4][LEN];
    __shared__ float bound_all[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = wave / WPK, sub = wave - grp * WPK;
    long long k = (long long)blockIdx.x * G + grp;
    if constexpr (WPK == 1) {
        if (k >= n) return;  // whole waves leave, there is no workgroup barrier below
    } else {
        k = k < n ? k : n - 1;  // WPK 2, odd n: the last workgroup's second pair of waves repeats the last keypoint (same bytes)
    }
    unsigned long long *hist = hist_all[wave];
    float *dst = reinterpret_cast<float *>(scratch_all[grp]);
    unsigned *queue = scratch_all[wave];
    float *out = desc + (size_t)k * dstride;
    const SiftGeom g = sift_geometry(kps + 4 * k, rows, cols);
    for (int t = lane; t < COPIES * SHIST; t += 64) hist[t] = 0ull;
    const int side = 2 * g.radius + 1;

    // sweep 1: a bound on every contribution, 2 * max(|gx|, |gy|) over the window's bounding square
    // inside the image interior (no rotation, no square root: ~8 instructions per sample)
    float bound = 0.f;
    {
        // a wave takes whole rows of the square: lane = column (two columns per lane when the side exceeds 64, more in a
        // loop), four rows' loads in flight per step -- the row-major "sample s of side^2" walk cost an index carry and
        // two dependent loads per sample (3.6 k of the kernel's 30 k instructions per keypoint at size 10)
        const int c_lo = g.px - g.radius > 1 ? g.px - g.radius : 1, c_hi = g.px + g.radius < cols - 2 ? g.px + g.radius : cols - 2;
        const int r_lo = g.py - g.radius > 1 ? g.py - g.radius : 1, r_hi = g.py + g.radius < rows - 2 ? g.py + g.radius : rows - 2;
        if (g.valid && c_lo <= c_hi) {
            for (int r = r_lo + 4 * sub; r <= r_hi; r += 4 * WPK) {
                for (int c = c_lo + lane; c <= c_hi; c += 64) {
                    float ax[4], ay[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int rr = r + k <= r_hi ? r + k : r_hi;  // (a repeated row does not change a max)
                        ax[k] = fabsf(gx[(size_t)rr * gstride + c]);
                        ay[k] = fabsf(gy[(size_t)rr * gstride + c]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float m = ax[k] > ay[k] ? ax[k] : ay[k];
                        bound = m > bound ? m : bound;  // NaN gradients never raise the bound
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float other = __shfl_xor(bound, o, 64);
        bound = other > bound ? other : bound;
    }
    if constexpr (WPK > 1) {
        bound_all[wave] = bound;
        __syncthreads();
#pragma unroll
        for (int w2 = 0; w2 < WPK; w2++) bound = bound_all[grp * WPK + w2] > bound ? bound_all[grp * WPK + w2] : bound;
    }
    bound = bound * 2.f;
    __builtin_amdgcn_wave_barrier();
    // flat (or empty, or invalid) window: all-zero descriptor -- the same decision in every wave of the keypoint; the
    // keypoint's waves still meet the other keypoint's at the workgroup barriers below when WPK > 1
    const bool flat = !(bound > 0.f) || !isfinite(bound);
    if (flat) {
        for (int t = 64 * sub + lane; t < LEN; t += 64 * WPK) out[t] = 0.f;
        if constexpr (WPK == 1) return;
    }
    int e = 0;
    if (!flat) (void)frexpf(bound, &e);  // bound < 2^e
    const int sh = 40 - e;
    const double fx_scale = ldexp(1.0, sh);

    // sweep 2: eight trilinear shares per sample, fixed-point adds into the wave's histogram
    // (the wave's zero fill of its own copies is ordered before its own atomics: same wave, in order)
    // A lone wave is bound by the latency of its gradient loads (one dependent L2 / HBM round trip per
    // sample would cost ~1 us each), so the samples go four passes at a time: the geometry tests and the
    // eight unconditional loads of a batch first (rejected samples read element 0), the arithmetic
    // and the LDS atomics after them.
    // The wave visits the bounding square in 4x4-pixel blocks, four blocks per pass (the sums are
    // order-independent): about half of the square lies outside the rotated window, and only blocks that can hold an
    // accepted sample are visited at all.
    {
        constexpr int U = 4;
        constexpr int CAND = 128;  // candidate blocks per round, two per lane; the survivors' (bi, bj) queue in LDS
        const unsigned nb = (unsigned)(side + 3) >> 2;
        const unsigned nblk = g.valid && !flat ? nb * nb : 0u;  // (nb <= 46 341 for the 65 535-pixel sides the entry point admits)
        const int ly = (lane >> 2) & 3, lx = lane & 3, quarter = lane >> 4;
        // Blocks that cannot hold an accepted sample are dropped BEFORE their per-sample tests (r05): rbin and cbin are
        // monotone in i and in j separately (a float product of a fixed factor is monotone, so is a float sum in each
        // operand), so over a block they lie between their values at the block's four corners; if that range misses
        // (-1, SD) for either coordinate no sample of the block passes sift_test.  One lane = one candidate block.
        // Exact, not a heuristic: nothing that contributes is skipped.  With 4x4 blocks 86 % of the visited lanes hold an
        // accepted sample of a size-10 window (8x8 blocks, one per pass: 63 %).
        for (unsigned base = 0; base < nblk; base += CAND) {
            int count = 0;
#pragma unroll
            for (int q = 0; q < CAND / 64; q++) {
                const unsigned cand = base + q * 64 + lane;
                bool maybe = false;
                unsigned packed = 0;
                if (cand < nblk) {
                    const unsigned bi_l = cand / nb, bj_l = cand - bi_l * nb;
                    const int i0 = -g.radius + 4 * (int)bi_l, j0 = -g.radius + 4 * (int)bj_l;
                    const int i1 = i0 + 3 < g.radius ? i0 + 3 : g.radius, j1 = j0 + 3 < g.radius ? j0 + 3 : g.radius;
                    float rmin = INFINITY, rmax = -INFINITY, cmin = INFINITY, cmax = -INFINITY;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int i = (k & 2) ? i1 : i0, j = (k & 1) ? j1 : j0;
                        const float c_rot = (float)j * g.cos_t - (float)i * g.sin_t;  // sift_test's own expressions
                        const float r_rot = (float)j * g.sin_t + (float)i * g.cos_t;
                        const float rb = r_rot + (float)(SD / 2) - 0.5f, cb = c_rot + (float)(SD / 2) - 0.5f;
                        rmin = fminf(rmin, rb); rmax = fmaxf(rmax, rb);
                        cmin = fminf(cmin, cb); cmax = fmaxf(cmax, cb);
                    }
                    // (NaN geometry cannot occur: g.valid; the comparisons are the negation of sift_test's, on the extremes)
                    maybe = rmax > -1.f && rmin < (float)SD && cmax > -1.f && cmin < (float)SD;
                    packed = bi_l << 16 | bj_l;
                }
                const unsigned long long m = __ballot(maybe);
                if (maybe)
                    queue[count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = packed;
                count += __popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the wave's own queue writes before its reads
            __builtin_amdgcn_wave_barrier();
            // this wave's share of the passes (four queued blocks each): every WPK-th one, U at a time
            const int npass = (count + 3) >> 2;
            for (int p0 = sub; p0 < npass; p0 += U * WPK) {
            float rbin[U], cbin[U], dx[U], dy[U], w[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                ok[u] = false;
                const int pp = p0 + u * WPK;
                if (pp >= npass) continue;  // the same in every lane
                const int slot = 4 * pp + quarter;
                const unsigned packed = queue[slot < count ? slot : 0];
                const int i = -g.radius + 4 * (int)(packed >> 16) + ly, j = -g.radius + 4 * (int)(packed & 0xffffu) + lx;
                ok[u] = slot < count && i <= g.radius && j <= g.radius &&
                        sift_test(rows, cols, g, i, j, rbin[u], cbin[u], w[u]);
                const size_t off = ok[u] ? (size_t)(g.py + i) * gstride + (g.px + j) : 0;
                dx[u] = gx[off];
                dy[u] = gy[off];  // (negated below: a negation here would wait for the load inside the batch)
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!ok[u]) continue;
                const float dyu = -dy[u];  // SIFT's dy is "up minus down"
                const float mag = sqrtf(dx[u] * dx[u] + dyu * dyu) * exp_neg(w[u]);
                float obin = (fast_atan2_deg(dyu, dx[u]) - g.ori) * ((float)SN / 360.f);
                float rb = rbin[u], cb = cbin[u];
                const float r0f = floorf(rb), c0f = floorf(cb), o0f = floorf(obin);
                rb -= r0f;
                cb -= c0f;
                obin -= o0f;
                const int r0 = (int)r0f, c0 = (int)c0f;
                int o0 = (int)o0f;
                if (o0 < 0) o0 += SN;
                if (o0 >= SN) o0 -= SN;
                const float v_r1 = mag * rb, v_r0 = mag - v_r1;
                const float v_rc11 = v_r1 * cb, v_rc10 = v_r1 - v_rc11;
                const float v_rc01 = v_r0 * cb, v_rc00 = v_r0 - v_rc01;
                const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
                const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
                const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
                const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
                // NaN samples (a NaN gradient) have no defined bin: o0 is clamped so the adds stay inside
                // the histogram; what they add is llrintf(NaN) as the host's cvtss2si returns it, INT64_MIN, for
                // each of the eight shares (every share of a NaN magnitude is NaN)
                o0 = o0 < 0 ? 0 : (o0 > SN - 1 ? SN - 1 : o0);
                // r0, c0 in [-1, SD - 1]: the shares on row r0 (column c0) belong to the grid when r0 (c0) >= 0, those on
                // row r0 + 1 (column c0 + 1) when it is <= SD - 1
                const bool r_lo = r0 >= 0, r_hi = r0 < SD - 1, c_lo = c0 >= 0, c_hi = c0 < SD - 1;
                const int idx = (r0 * SD + c0) * (SN + 1) + o0 + (lane & (COPIES - 1)) * SHIST;
                constexpr int DC = SN + 1, DR = SD * (SN + 1);
                if (mag != mag) {
                    const unsigned long long ind = 0x8000000000000000ull;
                    if (r_lo && c_lo) { atomicAdd(&hist[idx], ind); atomicAdd(&hist[idx + 1], ind); }
                    if (r_lo && c_hi) { atomicAdd(&hist[idx + DC], ind); atomicAdd(&hist[idx + DC + 1], ind); }
                    if (r_hi && c_lo) { atomicAdd(&hist[idx + DR], ind); atomicAdd(&hist[idx + DR + 1], ind); }
                    if (r_hi && c_hi) { atomicAdd(&hist[idx + DR + DC], ind); atomicAdd(&hist[idx + DR + DC + 1], ind); }
                    continue;
                }
                // llrintf(ldexpf(v, sh)) for 0 <= v * 2^sh < 2^41: one double fma onto 2^52 rounds to the
                // nearest-even integer and leaves it in the low mantissa bits
#define MICV_FX(v) ((unsigned long long)__double_as_longlong(fma((double)(v), fx_scale, 4503599627370496.0)) & 0x000FFFFFFFFFFFFFull)
                if (r_lo && c_lo) { atomicAdd(&hist[idx], MICV_FX(v000)); atomicAdd(&hist[idx + 1], MICV_FX(v001)); }
                if (r_lo && c_hi) { atomicAdd(&hist[idx + DC], MICV_FX(v010)); atomicAdd(&hist[idx + DC + 1], MICV_FX(v011)); }
                if (r_hi && c_lo) { atomicAdd(&hist[idx + DR], MICV_FX(v100)); atomicAdd(&hist[idx + DR + 1], MICV_FX(v101)); }
                if (r_hi && c_hi) { atomicAdd(&hist[idx + DR + DC], MICV_FX(v110)); atomicAdd(&hist[idx + DR + DC + 1], MICV_FX(v111)); }
#undef MICV_FX
            }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // queue reads before the next round's writes
            __builtin_amdgcn_wave_barrier();
        }
    }
    keypoint_sync<WPK>();

    // finalize: circular orientation axis, spatial border bins dropped, back to float
    const unsigned long long *hist0 = hist_all[grp * WPK];  // the keypoint's WPK x COPIES copies are contiguous
    for (int t = 64 * sub + lane; t < LEN; t += 64 * WPK) {
        const int cell = t / SN, o = t - cell * SN;
        const int ci = cell / SD, cj = cell - ci * SD;
        const int idx = (ci * SD + cj) * (SN + 1);
        long long h = 0;
#pragma unroll
        for (int cp = 0; cp < COPIES * WPK; cp++) {
            h += (long long)hist0[cp * SHIST + idx + o];
            if (o == 0) h += (long long)hist0[cp * SHIST + idx + SN];  // the orientation axis is circular
        }
        dst[t] = ldexpf((float)h, e - 40);
    }
    keypoint_sync<WPK>();
    if (flat) return;  // (after the last barrier)
    // the two 128-term norms, left to right in float (every lane computes them: uniform, no broadcast)
    float nrm2 = 0.f;
    for (int t = 0; t < LEN; t++) nrm2 += dst[t] * dst[t];
    const float thr = sqrtf(nrm2) * 0.2f;  // SIFT_DESCR_MAG_THR
    nrm2 = 0.f;
    for (int t = 0; t < LEN; t++) {
        const float d0 = dst[t];
        const float val = d0 < thr ? d0 : thr;
        nrm2 += val * val;
    }
    const float nrm = sqrtf(nrm2);
    const float scale = 512.f / (nrm > FLT_EPSILON ? nrm : FLT_EPSILON);  // SIFT_INT_DESCR_FCTR
    for (int t = 64 * sub + lane; t < LEN; t += 64 * WPK) {
        const float d0 = dst[t];
        const float val = d0 < thr ? d0 : thr;
        const float v = rintf(val * scale);  // saturate_cast<uchar>
        out[t] = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    }
}

}  // namespace micv

using namespace micv;

extern "C" int micv_sift_descriptors_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                                         size_t gstride, const float *kp_xysa, int64_t n, float *desc,
                                         size_t dstride, micv_stream stream) {
    MICV_REQUIRE(ctx && gx && gy, "micv_sift_descriptors: null argument");
    MICV_REQUIRE(n >= 0 && (n == 0 || (kp_xysa && desc)), "micv_sift_descriptors: bad keypoint list");
    MICV_REQUIRE(rows > 0 && cols > 0 && stride_ok(gstride, cols, 4), "micv_sift_descriptors: bad size / stride");
    MICV_REQUIRE(dstride % 4 == 0 && dstride >= 128 * 4 && dstride / 4 < ((size_t)1 << 30),
                 "micv_sift_descriptors: descriptor rows are 128 floats");
    MICV_REQUIRE(n < ((int64_t)1 << 31) * 4, "micv_sift_descriptors: too many keypoints");
    MICV_REQUIRE(rows <= 65535 && cols <= 65535, "micv_sift_descriptors: images up to 65 535 pixels on a side");
    MICV_HIP(hipSetDevice(ctx->device));
    if (n == 0) return MICV_OK;
    // fewer keypoints than two waves per SIMD: spend four waves on each
    // Waves per keypoint by the list's length (r05, 4K / 1080p checkerboards, size-10 keypoints, ms at 4 / 2 / 1 waves):
    // 1 222 keypoints 0.075 / 0.085 / 0.119; 2 006: 0.112 / 0.106 / 0.123; 3 476: 0.171 / 0.164 / 0.179; 5 035: 0.231 /
    // 0.218 / 0.220; 9 176: 0.390 / 0.360 / 0.351 -- a short list needs the waves, a long one pays for the repeated set-up.
    const int wpk = n < 1800 ? 4 : (n < 6144 ? 2 : 1);
    if (wpk == 4)
        sift_descriptor_kernel<4><<<(unsigned)n, 256, 0, static_cast<hipStream_t>(stream)>>>(
            gx, gy, (int)(gstride / 4), rows, cols, kp_xysa, (long long)n, desc, (int)(dstride / 4));
    else if (wpk == 2)
        sift_descriptor_kernel<2><<<(unsigned)((n + 1) / 2), 256, 0, static_cast<hipStream_t>(stream)>>>(
            gx, gy, (int)(gstride / 4), rows, cols, kp_xysa, (long long)n, desc, (int)(dstride / 4));
    else
        sift_descriptor_kernel<1><<<(unsigned)((n + 3) / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(
            gx, gy, (int)(gstride / 4), rows, cols, kp_xysa, (long long)n, desc, (int)(dstride / 4));
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}
