// match.hip -- descriptor matching of ps4: brute-force 2-nearest
// neighbours under L2 and the Lowe ratio test (cv::BFMatcher::knnMatch(k=2) + `d0 < 0.75 d1`,
// ps4_cpp/src/Solution.cpp:172-184).
//
// N x M x dim squared differences look like a GEMM, and ||a-b||^2 = ||a||^2 + ||b||^2 - 2ab would
// put it on MFMA -- but that changes every low-order bit (cancellation) and with it the ranking of
// near-ties.  The contract is the direct form, one fmaf chain per (query, train) pair in dimension
// order, so this is an LDS-tiled VALU kernel: a workgroup owns 64 queries, streams the train set
// through LDS in tiles of 64 rows, and every thread carries 16 independent chains (one query x 16
// train rows), reading its query element from a padded LDS image and the train element as a
// wave-wide broadcast.
#include "compact.hpp"
#include "kernels.hpp"

namespace micv {

struct Top2 {
    float d0, d1;
    int i0, i1;
    __device__ void init() { d0 = d1 = INFINITY; i0 = i1 = -1; }
    // order by (distance, index): what a strict `<` scan in index order keeps
    __device__ void push(float d, int i) {
        if (d < d0 || (d == d0 && (unsigned)i < (unsigned)i0)) {
            d1 = d0; i1 = i0; d0 = d; i0 = i;
        } else if (d < d1 || (d == d1 && (unsigned)i < (unsigned)i1)) {
            d1 = d; i1 = i;
        }
    }
};

constexpr int kQT = 64, kTT = 64, kDC = 32;  // queries / train rows per tile, dimensions per chunk

__global__ __launch_bounds__(256) void bf_knn2_kernel(const float *__restrict__ query, int nq,
                                                       int qstride, const float *__restrict__ train,
                                                       int nt, int tstride, int dim,
                                                       int32_t *__restrict__ idx2,
                                                       float *__restrict__ dist2) {
    __shared__ float Q[kQT][kDC + 1];   // +1: lanes read a column, one row per lane
    __shared__ float T[kTT][kDC];
    __shared__ float md[4][kQT][2];
    __shared__ int mi[4][kQT][2];
    const int tid = threadIdx.x, q = tid & 63, grp = tid >> 6;
    const int q0 = blockIdx.x * kQT;
    Top2 best;
    best.init();
    for (int t0 = 0; t0 < nt; t0 += kTT) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; j++) acc[j] = 0.f;
        for (int k0 = 0; k0 < dim; k0 += kDC) {  // chains continue across chunks in dimension order
            __syncthreads();
            for (int i = tid; i < kQT * kDC; i += 256) {
                const int r = i / kDC, c = i - r * kDC;
                Q[r][c] = (q0 + r < nq && k0 + c < dim) ? query[(size_t)(q0 + r) * qstride + k0 + c] : 0.f;
                T[r][c] = (t0 + r < nt && k0 + c < dim) ? train[(size_t)(t0 + r) * tstride + k0 + c] : 0.f;
            }
            __syncthreads();
            const int kn = dim - k0 < kDC ? dim - k0 : kDC;
            for (int k = 0; k < kn; k++) {
                const float a = Q[q][k];
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const float diff = a - T[grp * 16 + j][k];
                    acc[j] = fmaf(diff, diff, acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int t = t0 + grp * 16 + j;
            if (t < nt) best.push(acc[j], t);
        }
    }
    md[grp][q][0] = best.d0; md[grp][q][1] = best.d1;
    mi[grp][q][0] = best.i0; mi[grp][q][1] = best.i1;
    __syncthreads();
    if (grp == 0 && q0 + q < nq) {
        Top2 m;
        m.init();
        for (int g = 0; g < 4; g++)
            for (int s = 0; s < 2; s++)
                if (mi[g][q][s] >= 0) m.push(md[g][q][s], mi[g][q][s]);
        idx2[2 * (q0 + q)] = m.i0;
        idx2[2 * (q0 + q) + 1] = m.i1;
        dist2[2 * (q0 + q)] = sqrtf(m.d0);
        dist2[2 * (q0 + q) + 1] = sqrtf(m.d1);
    }
}

struct RatioPred {
    const float *dist2;
    double ratio;
    __device__ bool operator()(int64_t q) const {
        return (double)dist2[2 * q] < ratio * (double)dist2[2 * q + 1];  // Solution.cpp:181
    }
};

__global__ void bf_emit_kernel(const int32_t *__restrict__ sel, const int64_t *__restrict__ count,
                               int64_t cap, const int32_t *__restrict__ idx2,
                               const float *__restrict__ dist2, int32_t *__restrict__ matches,
                               float *__restrict__ distances) {
    const int64_t n = *count < cap ? *count : cap;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int q = sel[i];
        matches[2 * i] = q;
        matches[2 * i + 1] = idx2[2 * q];
        distances[i] = dist2[2 * q];
    }
}

}  // namespace micv

using namespace micv;

extern "C" {

int micv_bf_knn2_dev(micv_ctx *ctx, const float *query, int nq, size_t qstride, const float *train,
                     int nt, size_t tstride, int dim, int32_t *idx2, float *dist2,
                     micv_stream stream) {
    MICV_REQUIRE(ctx && query && train && idx2 && dist2, "micv_bf_knn2: null argument");
    MICV_REQUIRE(nq > 0 && nt >= 2 && dim > 0, "micv_bf_knn2: need nq > 0, nt >= 2, dim > 0");
    MICV_REQUIRE(stride_ok(qstride, dim, 4) && stride_ok(tstride, dim, 4), "micv_bf_knn2: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    bf_knn2_kernel<<<cdiv(nq, kQT), 256, 0, static_cast<hipStream_t>(stream)>>>(
        query, nq, (int)(qstride / 4), train, nt, (int)(tstride / 4), dim, idx2, dist2);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_bf_ratio_filter_dev(micv_ctx *ctx, const int32_t *idx2, const float *dist2, int nq,
                             double ratio, int32_t *matches_qt, float *distances, int64_t cap,
                             int64_t *count, micv_stream stream) {
    MICV_REQUIRE(ctx && idx2 && dist2 && count, "micv_bf_ratio_filter: null argument");
    MICV_REQUIRE(nq > 0 && cap >= 0 && (cap == 0 || (matches_qt && distances)),
                 "micv_bf_ratio_filter: bad size / outputs");
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need((size_t)cap + 1, 4) + compact_scratch_bytes(nq), &scratch));
    Carver c(scratch);
    int32_t *sel = c.take<int32_t>((size_t)cap + 1);
    MICV_TRY(ordered_compact(s, RatioPred{dist2, ratio}, nq, sel, cap, count, c.base + c.off));
    if (cap > 0) {
        bf_emit_kernel<<<64, 256, 0, s>>>(sel, count, cap, idx2, dist2, matches_qt, distances);
        MICV_LAUNCH_CHECK();
    }
    return MICV_OK;
}

}  // extern "C"
