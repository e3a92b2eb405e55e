// compact.hpp -- ordered stream compaction (replaces the reference's thrust::copy_if,
// Harris.cu:300-306, Hough.cu:226-227).  Emits, in ASCENDING index order, the linear indices
// i in [0, n) for which pred(i) holds -- row-major order for images, which is the order the
// reference's corner lists and point lists have.
//
// Three launches, no host synchronisation:
//   count : every 1024-element chunk counts its hits (wave ballots)          -> chunk_count[]
//   scan  : one workgroup scans the chunk counts                              -> chunk_off[], total
//   emit  : every chunk recomputes pred, ranks its hits and writes them at chunk_off + rank
#pragma once
#include "common.hpp"

namespace micv {

constexpr int kChunk = 1024;  // elements per workgroup (256 threads x 4)

template <typename Pred>
__global__ __launch_bounds__(256) void compact_count_kernel(Pred pred, int64_t n,
                                                             int *__restrict__ chunk_count) {
    __shared__ int wsum[4];
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const bool hit = i < n && pred(i);
        c += __popcll(__ballot(hit));
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Exclusive scan of nchunks counts by a single 1024-thread workgroup; total -> *count.
__global__ void compact_scan_kernel(const int *__restrict__ chunk_count,
                                    int64_t *__restrict__ chunk_off, int nchunks,
                                    int64_t *__restrict__ count);

template <typename Pred>
__global__ __launch_bounds__(256) void compact_emit_kernel(Pred pred, int64_t n,
                                                            const int64_t *__restrict__ chunk_off,
                                                            int32_t *__restrict__ out,
                                                            int64_t cap) {
    __shared__ int wcount[16];
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool hit[4];
    int before[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        hit[j] = i < n && pred(i);
        const unsigned long long m = __ballot(hit[j]);
        before[j] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[j * 4 + wave] = __popcll(m);
    }
    __syncthreads();
    const int64_t off = chunk_off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!hit[j]) continue;
        int prior = 0;  // hits in earlier (j', wave') slots: slot order == index order
        for (int s = 0; s < j * 4 + wave; s++) prior += wcount[s];
        const int64_t pos = off + prior + before[j];
        if (pos < cap) out[pos] = (int32_t)(base + j * 256 + threadIdx.x);
    }
}

inline size_t compact_scratch_bytes(int64_t n) {
    const int64_t nchunks = (n + kChunk - 1) / kChunk;
    return Carver::need(nchunks, 4) + Carver::need(nchunks, 8);
}

int launch_compact_scan(hipStream_t s, const int *chunk_count, int64_t *chunk_off, int nchunks,
                        int64_t *count);

// out: device int32[cap]; count: device int64 (total hits, may exceed cap).
template <typename Pred>
int ordered_compact(hipStream_t s, Pred pred, int64_t n, int32_t *out, int64_t cap, int64_t *count,
                    void *scratch) {
    if (n >= (int64_t)1 << 31) {
        set_error("ordered_compact: %lld elements exceed int32 indices", (long long)n);
        return MICV_EINVAL;
    }
    const int nchunks = (int)((n + kChunk - 1) / kChunk);
    Carver c(scratch);
    int *chunk_count = c.take<int>(nchunks);
    int64_t *chunk_off = c.take<int64_t>(nchunks);
    if (nchunks == 0) {
        MICV_HIP(hipMemsetAsync(count, 0, 8, s));
        return MICV_OK;
    }
    compact_count_kernel<<<nchunks, 256, 0, s>>>(pred, n, chunk_count);
    MICV_LAUNCH_CHECK();
    MICV_TRY(launch_compact_scan(s, chunk_count, chunk_off, nchunks, count));
    compact_emit_kernel<<<nchunks, 256, 0, s>>>(pred, n, chunk_off, out, cap);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

}  // namespace micv
