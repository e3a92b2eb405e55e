// compact.hip -- the non-template half of the ordered stream compaction (compact.hpp): the
// single-workgroup exclusive scan of per-chunk hit counts.
#include "compact.hpp"

namespace micv {

__global__ __launch_bounds__(1024) void compact_scan_kernel(const int *__restrict__ chunk_count,
                                                             int64_t *__restrict__ chunk_off,
                                                             int nchunks,
                                                             int64_t *__restrict__ count) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nchunks; base += 1024) {
        const int i = base + threadIdx.x;
        const int64_t v = i < nchunks ? chunk_count[i] : 0;
        int64_t incl = v;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int64_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const int64_t carry = carry_s;
        if (i < nchunks) chunk_off[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = carry_s;
}

int launch_compact_scan(hipStream_t s, const int *chunk_count, int64_t *chunk_off, int nchunks,
                        int64_t *count) {
    compact_scan_kernel<<<1, 1024, 0, s>>>(chunk_count, chunk_off, nchunks, count);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

}  // namespace micv
