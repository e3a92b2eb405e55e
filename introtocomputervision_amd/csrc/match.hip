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

// Tile: 64 queries x 128 train rows per pass of a 256-thread workgroup, a thread owns 4 queries x 8
// train rows (32 chains).  Both operands are staged TRANSPOSED ([k][row]) so that a thread's 4
// query values and 8 train values of one dimension k are ds_read_b128 reads, and adjacent train
// rows sit in adjacent registers: the chain step  d = a - t; acc = fma(d, d, acc)  runs as
// v_pk_add_f32 / v_pk_fma_f32 on (t_j, t_j+1) pairs -- two chains per instruction, each still its
// own IEEE fmaf chain in dimension order.  The train set is cut into slices over blockIdx.y so
// that small query sets still fill the GPU; bf_merge_kernel folds the slices' top-2 in slice
// (= index) order.
typedef float mf2 __attribute__((ext_vector_type(2)));
typedef float mf4 __attribute__((ext_vector_type(4)));
constexpr int kQT = 64, kTT = 128, kDC = 32;

__global__ __launch_bounds__(256) void bf_knn2_kernel(const float *__restrict__ query, int nq,
                                                       int qstride, const float *__restrict__ train,
                                                       int nt, int tstride, int dim, int slice_rows,
                                                       int32_t *__restrict__ part_i,
                                                       float *__restrict__ part_d, int vec_ok) {
    __shared__ __attribute__((aligned(16))) float Qt[kDC][kQT];
    __shared__ __attribute__((aligned(16))) float Tt[kDC][kTT];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int q0 = blockIdx.x * kQT;
    const int ts0 = blockIdx.y * slice_rows, ts1 = ts0 + slice_rows < nt ? ts0 + slice_rows : nt;
    Top2 best[4];
#pragma unroll
    for (int i = 0; i < 4; i++) best[i].init();
    for (int t0 = ts0; t0 < ts1; t0 += kTT) {
        mf2 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = (mf2)(0.f);
        // The next chunk's operands are PREFETCHED into registers (24 floats per thread) before the current chunk's chains
        // run and written to LDS after them: the global round trip flies under 32 x 32 packed instructions instead of in
        // front of them (r05; the same move as the stereo NCC kernel).
        float pq[8], pt[16];
        // (a lane reads 8 / 16 consecutive floats of ITS OWN row, so every load instruction touches 64 different cache
        // lines: as 16-byte loads -- rows and pitches aligned, dim a multiple of 4 -- that is 6 instructions per chunk
        // instead of 24, and the address unit stops being the co-limiter of the chains; r05)
        auto prefetch = [&](int k0) {
            {
                const int r = tid & 63, kq = (tid >> 6) * 8;
                const bool rin = q0 + r < nq;
                const float *src = query + (size_t)(rin ? q0 + r : 0) * qstride + k0 + kq;
                if (vec_ok) {
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const mf4 v = (rin && k0 + kq + 4 * i < dim) ? *reinterpret_cast<const mf4 *>(src + 4 * i) : (mf4)(0.f);
                        pq[4 * i] = v.x; pq[4 * i + 1] = v.y; pq[4 * i + 2] = v.z; pq[4 * i + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) pq[i] = (rin && k0 + kq + i < dim) ? src[i] : 0.f;
                }
            }
            {
                const int r = tid & 127, kh = (tid >> 7) * 16;
                const bool rin = t0 + r < ts1;
                const float *src = train + (size_t)(rin ? t0 + r : 0) * tstride + k0 + kh;
                if (vec_ok) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const mf4 v = (rin && k0 + kh + 4 * i < dim) ? *reinterpret_cast<const mf4 *>(src + 4 * i) : (mf4)(0.f);
                        pt[4 * i] = v.x; pt[4 * i + 1] = v.y; pt[4 * i + 2] = v.z; pt[4 * i + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i++) pt[i] = (rin && k0 + kh + i < dim) ? src[i] : 0.f;
                }
            }
        };
        prefetch(0);
        for (int k0 = 0; k0 < dim; k0 += kDC) {  // chains continue across chunks in dimension order
            __syncthreads();
            {   // stage, transposing: lane = row, so the LDS stores of one k are contiguous
                const int r = tid & 63, kq = (tid >> 6) * 8;
#pragma unroll
                for (int i = 0; i < 8; i++) Qt[kq + i][r] = pq[i];
            }
            {
                const int r = tid & 127, kh = (tid >> 7) * 16;
#pragma unroll
                for (int i = 0; i < 16; i++) Tt[kh + i][r] = pt[i];
            }
            __syncthreads();
            if (k0 + kDC < dim) prefetch(k0 + kDC);
            const int kn = dim - k0 < kDC ? dim - k0 : kDC;
#pragma unroll 4
            for (int k = 0; k < kn; k++) {
                const mf4 a = *reinterpret_cast<const mf4 *>(&Qt[k][4 * tx]);
                const mf4 b0 = *reinterpret_cast<const mf4 *>(&Tt[k][8 * ty]);
                const mf4 b1 = *reinterpret_cast<const mf4 *>(&Tt[k][8 * ty + 4]);
                const mf2 t[4] = {b0.xy, b0.zw, b1.xy, b1.zw};
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const mf2 d = (mf2)(av[i]) - t[j];
                        acc[i][j] = __builtin_elementwise_fma(d, d, acc[i][j]);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int t = t0 + 8 * ty + 2 * j;
                if (t < ts1) best[i].push(acc[i][j].x, t);
                if (t + 1 < ts1) best[i].push(acc[i][j].y, t + 1);
            }
    }
    // fold the 16 train-row groups of each query (ascending ty = ascending index)
    __shared__ float md[16][kQT][2];
    __shared__ int mi[16][kQT][2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        md[ty][4 * tx + i][0] = best[i].d0; md[ty][4 * tx + i][1] = best[i].d1;
        mi[ty][4 * tx + i][0] = best[i].i0; mi[ty][4 * tx + i][1] = best[i].i1;
    }
    __syncthreads();
    if (tid < kQT && q0 + tid < nq) {
        Top2 m;
        m.init();
        for (int g = 0; g < 16; g++)
            for (int sidx = 0; sidx < 2; sidx++)
                if (mi[g][tid][sidx] >= 0) m.push(md[g][tid][sidx], mi[g][tid][sidx]);
        const size_t o = ((size_t)blockIdx.y * nq + q0 + tid) * 2;
        part_i[o] = m.i0; part_i[o + 1] = m.i1;
        part_d[o] = m.d0; part_d[o + 1] = m.d1;
    }
}

// Top-2 over the slices' partial top-2 lists (slice order = index order), then the square roots.
__global__ __launch_bounds__(256) void bf_merge_kernel(const int32_t *__restrict__ part_i,
                                                        const float *__restrict__ part_d, int nq,
                                                        int slices, int32_t *__restrict__ idx2,
                                                        float *__restrict__ dist2) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    Top2 m;
    m.init();
    for (int s = 0; s < slices; s++)
        for (int k = 0; k < 2; k++) {
            const size_t o = ((size_t)s * nq + q) * 2 + k;
            if (part_i[o] >= 0) m.push(part_d[o], part_i[o]);
        }
    idx2[2 * q] = m.i0;
    idx2[2 * q + 1] = m.i1;
    dist2[2 * q] = sqrtf(m.d0);
    dist2[2 * q + 1] = sqrtf(m.d1);
}

struct RatioPred {
    const float *dist2;
    double ratio;
    // (center / test: the one-launch compaction loads a thread's 16 pairs before it decides the first, compact.hpp)
    __device__ float2 center(int64_t q) const { return make_float2(dist2[2 * q], dist2[2 * q + 1]); }  // (caller's pointer: 4-byte aligned)
    __device__ bool test(int64_t, float2 d) const { return (double)d.x < ratio * (double)d.y; }  // Solution.cpp:181
    __device__ bool operator()(int64_t q) const { return test(q, center(q)); }
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
    hipStream_t s = static_cast<hipStream_t>(stream);
    // slices of the train set: enough workgroups for 2 - 4 per CU, whole 128-row passes per slice
    const int qblocks = (int)cdiv(nq, kQT), passes = (int)cdiv(nt, kTT);
    // (r05: four workgroups per CU once every workgroup still has eight passes to run -- 8192 x 8192: 0.438 -> 0.389 ms;
    // shorter jobs keep two per CU: 5035 x 5035 0.202 against 0.210, 500 x 20000 0.127 against 0.140)
    const int want_wgs = (long long)qblocks * passes >= 8192 ? 1024 : 512;
    int slices = (want_wgs + qblocks - 1) / qblocks;
    if (slices > passes) slices = passes;
    if (slices < 1) slices = 1;
    const int slice_rows = (int)cdiv(passes, slices) * kTT;
    slices = (int)cdiv(nt, slice_rows);
    void *scratch;
    MICV_TRY(ctx->reserve(2 * Carver::need((size_t)slices * nq * 2, 4), &scratch));
    Carver c(scratch);
    int32_t *part_i = c.take<int32_t>((size_t)slices * nq * 2);
    float *part_d = c.take<float>((size_t)slices * nq * 2);
    const int vec_ok = ((reinterpret_cast<uintptr_t>(query) | reinterpret_cast<uintptr_t>(train) | qstride | tstride) & 15) == 0 &&
                       (dim & 3) == 0;
    bf_knn2_kernel<<<dim3(qblocks, slices), 256, 0, s>>>(query, nq, (int)(qstride / 4), train, nt,
                                                         (int)(tstride / 4), dim, slice_rows, part_i,
                                                         part_d, vec_ok);
    MICV_LAUNCH_CHECK();
    bf_merge_kernel<<<cdiv(nq, 256), 256, 0, s>>>(part_i, part_d, nq, slices, idx2, dist2);
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
    MICV_TRY(ordered_compact(ctx, s, RatioPred{dist2, ratio}, IndexEmit{sel}, nq, cap, count, c.base + c.off));
    if (cap > 0) {
        bf_emit_kernel<<<64, 256, 0, s>>>(sel, count, cap, idx2, dist2, matches_qt, distances);
        MICV_LAUNCH_CHECK();
    }
    return MICV_OK;
}

}  // extern "C"
