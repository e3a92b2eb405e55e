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
// 64-bit fixed point (2^-40 of the window's largest gradient magnitude) so that the result does
// not depend on the order in which the samples arrive.
//
// One wave64 per keypoint, four keypoints per workgroup.  The window (107 x 107 samples for the
// reference's size-10 keypoints) is swept twice by the wave's lanes with coalesced row reads
// (second sweep served by L1/L2): sweep 1 finds the magnitude bound by a wave max-reduction, sweep
// 2 scatters the eight trilinear shares of every sample into the wave's LDS histogram with native
// 64-bit LDS atomics.  The 128-term norms are summed left to right by one lane (the contract).
#include <cfloat>

#include "common.hpp"

namespace micv {

constexpr int SD = 4, SN = 8;
constexpr int SHIST = (SD + 2) * (SD + 2) * (SN + 2);

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
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
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

__device__ __forceinline__ bool sift_sample(const float *__restrict__ gx, const float *__restrict__ gy,
                                            int gstride, int rows, int cols, const SiftGeom &g, int i, int j,
                                            float &rbin, float &cbin, float &dx, float &dy, float &wexp) {
    const float c_rot = (float)j * g.cos_t - (float)i * g.sin_t;
    const float r_rot = (float)j * g.sin_t + (float)i * g.cos_t;
    rbin = r_rot + (float)(SD / 2) - 0.5f;
    cbin = c_rot + (float)(SD / 2) - 0.5f;
    const int r = g.py + i, c = g.px + j;
    if (!(rbin > -1.f && rbin < (float)SD && cbin > -1.f && cbin < (float)SD && r > 0 && r < rows - 1 &&
          c > 0 && c < cols - 1))
        return false;
    dx = gx[(size_t)r * gstride + c];
    dy = -gy[(size_t)r * gstride + c];  // SIFT's dy is "up minus down"
    wexp = (c_rot * c_rot + r_rot * r_rot) * (-1.f / ((float)(SD * SD) * 0.5f));
    return true;
}

__global__ __launch_bounds__(256) void sift_descriptor_kernel(const float *__restrict__ gx,
                                                               const float *__restrict__ gy, int gstride,
                                                               int rows, int cols,
                                                               const float *__restrict__ kps, long long n,
                                                               float *__restrict__ desc, int dstride) {
    __shared__ unsigned long long hist_all[4][SHIST];
    __shared__ float dst_all[4][SD * SD * SN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long k = (long long)blockIdx.x * 4 + wave;
    if (k >= n) return;  // whole waves leave; no workgroup barrier below
    unsigned long long *hist = hist_all[wave];
    float *dst = dst_all[wave];
    float *out = desc + (size_t)k * dstride;
    const SiftGeom g = sift_geometry(kps + 4 * k, rows, cols);
    for (int t = lane; t < SHIST; t += 64) hist[t] = 0ull;
    const int side = 2 * g.radius + 1;
    const long long total = g.valid ? (long long)side * side : 0;

    // sweep 1: largest gradient magnitude among the contributing samples
    float bound = 0.f;
    {
        int i = -g.radius, j = -g.radius + lane;
        for (long long s = lane; s < total; s += 64) {
            while (j > g.radius) {
                j -= side;
                i++;
            }
            float rbin, cbin, dx, dy, w;
            if (sift_sample(gx, gy, gstride, rows, cols, g, i, j, rbin, cbin, dx, dy, w)) {
                const float mag = sqrtf(dx * dx + dy * dy);
                bound = mag > bound ? mag : bound;  // NaN magnitudes never raise the bound
            }
            j += 64;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float other = __shfl_xor(bound, o, 64);
        bound = other > bound ? other : bound;
    }
    __builtin_amdgcn_wave_barrier();
    if (!(bound > 0.f) || !isfinite(bound)) {  // flat (or empty, or invalid) window: all-zero descriptor
        for (int t = lane; t < SD * SD * SN; t += 64) out[t] = 0.f;
        return;
    }
    int e;
    (void)frexpf(bound, &e);  // bound < 2^e
    const int sh = 40 - e;

    // sweep 2: eight trilinear shares per sample, fixed-point adds into the wave's histogram
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the zero fill above is ordered first
    __builtin_amdgcn_wave_barrier();
    {
        int i = -g.radius, j = -g.radius + lane;
        for (long long s = lane; s < total; s += 64) {
            while (j > g.radius) {
                j -= side;
                i++;
            }
            float rbin, cbin, dx, dy, w;
            if (sift_sample(gx, gy, gstride, rows, cols, g, i, j, rbin, cbin, dx, dy, w)) {
                const float mag = sqrtf(dx * dx + dy * dy) * exp_neg(w);
                float obin = (fast_atan2_deg(dy, dx) - g.ori) * ((float)SN / 360.f);
                const float r0f = floorf(rbin), c0f = floorf(cbin), o0f = floorf(obin);
                rbin -= r0f;
                cbin -= c0f;
                obin -= o0f;
                const int r0 = (int)r0f, c0 = (int)c0f;
                int o0 = (int)o0f;
                if (o0 < 0) o0 += SN;
                if (o0 >= SN) o0 -= SN;
                const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
                const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
                const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
                const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
                const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
                const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
                const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
                // NaN samples (a NaN gradient) have no defined bin: o0 is clamped so the adds stay inside
                // the histogram; what they add is llrint(NaN), as on the host
                o0 = o0 < 0 ? 0 : (o0 > SN - 1 ? SN - 1 : o0);
                const int idx = ((r0 + 1) * (SD + 2) + c0 + 1) * (SN + 2) + o0;
#define MICV_FX(v) ((unsigned long long)llrintf(ldexpf((v), sh)))
                atomicAdd(&hist[idx], MICV_FX(v000));
                atomicAdd(&hist[idx + 1], MICV_FX(v001));
                atomicAdd(&hist[idx + (SN + 2)], MICV_FX(v010));
                atomicAdd(&hist[idx + (SN + 3)], MICV_FX(v011));
                atomicAdd(&hist[idx + (SD + 2) * (SN + 2)], MICV_FX(v100));
                atomicAdd(&hist[idx + (SD + 2) * (SN + 2) + 1], MICV_FX(v101));
                atomicAdd(&hist[idx + (SD + 3) * (SN + 2)], MICV_FX(v110));
                atomicAdd(&hist[idx + (SD + 3) * (SN + 2) + 1], MICV_FX(v111));
#undef MICV_FX
            }
            j += 64;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // finalize: circular orientation axis, spatial border bins dropped, back to float
    for (int t = lane; t < SD * SD * SN; t += 64) {
        const int cell = t / SN, o = t - cell * SN;
        const int ci = cell / SD, cj = cell - ci * SD;
        const int idx = ((ci + 1) * (SD + 2) + (cj + 1)) * (SN + 2);
        long long h = (long long)hist[idx + o];
        if (o < 2) h += (long long)hist[idx + SN + o];
        dst[t] = ldexpf((float)h, e - 40);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the two 128-term norms, left to right in float (every lane computes them: uniform, no broadcast)
    constexpr int LEN = SD * SD * SN;
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
    for (int t = lane; t < LEN; t += 64) {
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
    MICV_HIP(hipSetDevice(ctx->device));
    if (n == 0) return MICV_OK;
    sift_descriptor_kernel<<<(unsigned)((n + 3) / 4), 256, 0, static_cast<hipStream_t>(stream)>>>(
        gx, gy, (int)(gstride / 4), rows, cols, kp_xysa, (long long)n, desc, (int)(dstride / 4));
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}
