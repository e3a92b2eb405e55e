// compact.hpp -- ordered stream compaction (replaces the reference's thrust::copy_if,
// Harris.cu:300-306, Hough.cu:226-227).  Emits, in ASCENDING index order, the linear indices
// i in [0, n) for which pred(i) holds -- row-major order for images, which is the order the
// reference's corner lists and point lists have.
//
// One launch (compact_onepass_kernel), no host synchronisation: a chained scan with decoupled look-back.  Every
// 4096-element chunk takes a ticket (its position in index order = the order workgroups started, so all its
// predecessors are running or done), counts its hits, publishes the count, finds its offset by looking back over
// its predecessors' published counts / running totals (the whole workgroup, 256 chunks per step), publishes its own
// running total and writes its hits.  pred is evaluated once per element.  The status words live in a per-stream,
// context-owned buffer that is all zero between launches: the last chunk to finish zeroes it again.
// Without such a buffer (more than 16 streams on one context) the three-launch form runs instead:
//   count : every 1024-element chunk counts its hits (wave ballots)          -> chunk_count[]
//   scan  : one workgroup scans the chunk counts                              -> chunk_off[], total
//   emit  : every chunk recomputes pred, ranks its hits and writes them at chunk_off + rank
#pragma once
#include <type_traits>
#include <utility>
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

template <typename Pred, typename Emit>
__global__ __launch_bounds__(256) void compact_emit_kernel(Pred pred, Emit emit, int64_t n,
                                                            const int64_t *__restrict__ chunk_off,
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
        if (pos < cap) emit(pos, base + j * 256 + threadIdx.x);
    }
}

// What a hit becomes in the output: its linear index (default) ...
struct IndexEmit {
    int32_t *out;
    __device__ void operator()(int64_t pos, int64_t i) const { out[pos] = (int32_t)i; }
};
// ... or its (row, column), Harris.cu:314-318 (Conv1Dto2D)
struct YxEmit {
    int32_t *locs;
    int cols;
    __device__ void operator()(int64_t pos, int64_t i) const {
        const int y = (int)(i / cols);
        locs[2 * pos] = y;
        locs[2 * pos + 1] = (int)(i - (int64_t)y * cols);
    }
};

constexpr unsigned long long kCompactAgg = 1ull << 32, kCompactInc = 2ull << 32;  // status = flag | 32-bit count
constexpr int kChunk1 = 4096;  // elements per workgroup of the one-launch form (256 threads x 16)

// 4096 elements per workgroup and a look-back by the whole workgroup (256 predecessors per step).
// The shared state and the two steps every one-launch kernel runs between counting and writing its hits.
struct CompactShared {
    int wcount[64];  // hits of slot (j, wave), then their exclusive prefix: slot order == index order
    unsigned s_chunk, s_last;
    int s_total, s_first[4];
    long long s_part[4];
};

// After wcount[] holds the chunk's 64 slot counts (and a barrier): turns them into exclusive prefixes, publishes the
// chunk's count, finds the chunk's offset by the look-back, publishes its running total (and the grand total from
// the last chunk).  Returns the offset of the chunk's first hit; every thread of the workgroup calls it.
__device__ __forceinline__ long long compact_chunk_offset(CompactShared &sh, int chunk, int nchunks,
                                                          unsigned long long *__restrict__ status, int64_t *__restrict__ count) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const int c = sh.wcount[lane];
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        sh.wcount[lane] = incl - c;
        if (lane == 63) {
            sh.s_total = incl;
            __hip_atomic_store(&status[chunk], (chunk == 0 ? kCompactInc : kCompactAgg) | (unsigned)incl, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    long long excl = 0;
    for (int p = chunk - 1; p >= 0; p -= 256) {  // (uniform: every thread sees the same shared flags)
        const int q = p - (int)threadIdx.x;      // thread t looks at predecessor p - t
        unsigned long long st = kCompactInc;     // before chunk 0: a running total of 0
        if (q >= 0) {
            do {
                st = __hip_atomic_load(&status[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((st >> 32) == 0);
        }
        const unsigned long long inc = __ballot((st & kCompactInc) != 0);
        if (lane == 0) sh.s_first[wave] = inc ? __ffsll((long long)inc) - 1 : 64;  // nearest running total in this wave
        __syncthreads();
        int fw = 4;  // first wave (nearest 64 predecessors first) that saw a running total
#pragma unroll
        for (int w = 3; w >= 0; w--)
            if (sh.s_first[w] < 64) fw = w;
        long long v = (wave < fw || (wave == fw && lane <= sh.s_first[wave])) ? (long long)(unsigned)st : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0) sh.s_part[wave] = v;
        __syncthreads();
        excl += sh.s_part[0] + sh.s_part[1] + sh.s_part[2] + sh.s_part[3];
        __syncthreads();  // s_first / s_part are rewritten by the next step
        if (fw < 4) break;
    }
    __syncthreads();  // wcount's prefix and s_total (chunk 0 takes no step above)
    if (threadIdx.x == 0) {
        if (chunk > 0)
            __hip_atomic_store(&status[chunk], kCompactInc | (unsigned long long)(excl + sh.s_total), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        if (chunk == nchunks - 1) *count = excl + sh.s_total;
    }
    return excl;
}

// The last chunk out leaves the state as it found it: all zero.  Thread 0 is the one that stored this chunk's status
// words (sc1 stores): it waits for their acknowledgement before it counts the chunk out, so no such store can still
// be on its way when the last chunk zeroes the words (a wait, not a fence -- an agent-scope release would
// write back the whole L2 once per chunk).
__device__ __forceinline__ void compact_chunk_exit(CompactShared &sh, int nchunks, unsigned long long *__restrict__ status,
                                                   unsigned *__restrict__ counters) {
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh.s_last = atomicAdd(&counters[1], 1u) == (unsigned)(nchunks - 1);
    }
    __syncthreads();
    if (sh.s_last) {
        for (int i = threadIdx.x; i < nchunks; i += 256) status[i] = 0;
        if (threadIdx.x == 0) {
            counters[0] = 0;
            counters[1] = 0;
        }
    }
}

// Pred::center(i) / Pred::test(i, value) when the predicate offers them (see PeakPred), Pred::operator()(i) otherwise.
template <typename P, typename = void>
struct CompactHasCenter : std::false_type {};
template <typename P>
struct CompactHasCenter<P, std::void_t<decltype(std::declval<const P &>().center((int64_t)0))>> : std::true_type {};
template <typename P>
__device__ __forceinline__ auto compact_center(const P &p, int64_t i) {
    if constexpr (CompactHasCenter<P>::value) return p.center(i);
    else return 0;
}

template <typename Pred, typename Emit>
__global__ __launch_bounds__(256) void compact_onepass_kernel(Pred pred, Emit emit, int64_t n, int nchunks,
                                                               unsigned long long *__restrict__ status,
                                                               unsigned *__restrict__ counters, int64_t cap,
                                                               int64_t *__restrict__ count) {
    constexpr int J = kChunk1 / 256;
    static_assert(J * 4 == 64, "one (j, wave) slot per lane of the scanning wave");
    __shared__ CompactShared sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sh.s_chunk = atomicAdd(&counters[0], 1u);
    __syncthreads();
    const int chunk = (int)sh.s_chunk;
    const int64_t base = (int64_t)chunk * kChunk1;
    unsigned hits = 0;
    int before[J];
    // predicates that split into "load the element" and "decide" have all J loads in flight before the first decision
    [[maybe_unused]] decltype(compact_center(pred, (int64_t)0)) ctr[J];
    if constexpr (CompactHasCenter<Pred>::value) {
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int64_t i = base + j * 256 + threadIdx.x;
            ctr[j] = pred.center(i < n ? i : n - 1);
        }
    }
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        bool hit;
        if constexpr (CompactHasCenter<Pred>::value) hit = i < n && pred.test(i, ctr[j]);
        else hit = i < n && pred(i);
        const unsigned long long m = __ballot(hit);
        before[j] = __popcll(m & ((1ull << lane) - 1ull));
        hits |= (unsigned)hit << j;
        if (lane == 0) sh.wcount[j * 4 + wave] = __popcll(m);
    }
    __syncthreads();
    const long long excl = compact_chunk_offset(sh, chunk, nchunks, status, count);
#pragma unroll
    for (int j = 0; j < J; j++) {
        if (!((hits >> j) & 1)) continue;
        const int64_t pos = excl + sh.wcount[j * 4 + wave] + before[j];
        if (pos < cap) emit(pos, base + j * 256 + threadIdx.x);
    }
    compact_chunk_exit(sh, nchunks, status, counters);
}

// The same chained scan over 64-bit MASKS (r04): element i stands for 64 consecutive cells of an image row and
// contributes popcount(masks[i]) hits -- its set bits in ascending order.  The NMS kernels write one mask per
// (row, 64-column tile) by wave ballot, so the ordered corner list of a 4K frame is a scan over 130 k words in 32
// chunks instead of 8.3 M flag bytes in three launches.  emit(pos, i, bit).
// J words per thread = 256 J words per chunk: 16 for long lists; 4 keeps the launch wide when the words are few (32 k words
// of a 1080p edge mask are 8 chunks of 4096 -- eight workgroups on the whole GPU -- but 32 of 1024).
template <typename Emit, int J = kChunk1 / 256>
__global__ __launch_bounds__(256) void compact_masks_onepass_kernel(const unsigned long long *__restrict__ masks, Emit emit,
                                                                     int64_t n, int nchunks,
                                                                     unsigned long long *__restrict__ status,
                                                                     unsigned *__restrict__ counters, int64_t cap,
                                                                     int64_t *__restrict__ count) {
    static_assert(J >= 1 && J <= 16, "one (j, wave) slot per lane of the scanning wave");
    constexpr int CH = 256 * J;
    __shared__ CompactShared sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sh.s_chunk = atomicAdd(&counters[0], 1u);
    __syncthreads();
    const int chunk = (int)sh.s_chunk;
    const int64_t base = (int64_t)chunk * CH;
    unsigned long long m[J];
    int before[J];
    if (J < 16 && threadIdx.x >= 4 * J && threadIdx.x < 64) sh.wcount[threadIdx.x] = 0;  // the slots this chunk size leaves empty
#pragma unroll
    for (int j = 0; j < J; j++) {  // all loads in flight first
        const int64_t i = base + j * 256 + threadIdx.x;
        m[j] = i < n ? masks[i] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int c = __popcll(m[j]);
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        before[j] = incl - c;
        if (lane == 63) sh.wcount[j * 4 + wave] = incl;
    }
    __syncthreads();
    const long long excl = compact_chunk_offset(sh, chunk, nchunks, status, count);
#pragma unroll
    for (int j = 0; j < J; j++) {
        unsigned long long mm = m[j];
        int64_t pos = excl + sh.wcount[j * 4 + wave] + before[j];
        // Words with a few set bits (strict maxima are sparse) are emitted by their own lane; a word with many (a
        // horizontal run of an edge mask: up to 64 serial trips for one lane while the wave waits) is emitted by the
        // whole wave, one bit per lane (r05: the 1080p edge list 22 -> see profiles/r05/hough_chain.txt).
        const bool dense = __popcll(mm) > 4;
        unsigned long long todo = __ballot(dense);
        if (dense) mm = 0;
        while (mm) {
            const int bit = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            if (pos < cap) emit(pos, base + j * 256 + threadIdx.x, bit);
            pos++;
        }
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;  // wave-uniform
            todo &= todo - 1;
            const unsigned long long w = __shfl(m[j], l);
            const int64_t p = __shfl(pos, l) + __popcll(w & ((1ull << lane) - 1ull));
            if (((w >> lane) & 1ull) && p < cap) emit(p, base + j * 256 + (threadIdx.x - lane) + l, lane);
        }
    }
    compact_chunk_exit(sh, nchunks, status, counters);
}

// Launches the mask scan with the chunk size that suits the list (above); `nchunks_out` chunks of state are needed.
inline int compact_masks_chunks(int64_t nwords) { return (int)((nwords + (nwords <= 256 * 1024 ? 1024 : kChunk1) - 1) / (nwords <= 256 * 1024 ? 1024 : kChunk1)); }
template <typename Emit>
inline void launch_compact_masks(hipStream_t s, const unsigned long long *masks, Emit emit, int64_t nwords, unsigned long long *status,
                                 unsigned *counters, int64_t cap, int64_t *count) {
    const int nchunks = compact_masks_chunks(nwords);
    if (nwords <= 256 * 1024)
        compact_masks_onepass_kernel<Emit, 4><<<nchunks, 256, 0, s>>>(masks, emit, nwords, nchunks, status, counters, cap, count);
    else
        compact_masks_onepass_kernel<Emit, 16><<<nchunks, 256, 0, s>>>(masks, emit, nwords, nchunks, status, counters, cap, count);
}

inline size_t compact_scratch_bytes(int64_t n) {
    const int64_t nchunks = (n + kChunk - 1) / kChunk;
    return Carver::need(nchunks, 4) + Carver::need(nchunks, 8);
}

int launch_compact_scan(hipStream_t s, const int *chunk_count, int64_t *chunk_off, int nchunks,
                        int64_t *count);

// count: device int64 (total hits, may exceed cap: only the first cap are written).
template <typename Pred, typename Emit>
int ordered_compact3(hipStream_t s, Pred pred, Emit emit, int64_t n, int64_t cap, int64_t *count, void *scratch) {
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
    compact_emit_kernel<<<nchunks, 256, 0, s>>>(pred, emit, n, chunk_off, cap);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// The one-launch form when the context has a state buffer for this stream and the list is short enough (below;
// MICV_OPT_COMPACT_3PASS forces either form), the three launches otherwise.  `emit` decides what a hit becomes.
template <typename Pred, typename Emit>
int ordered_compact(micv_ctx *ctx, hipStream_t s, Pred pred, Emit emit, int64_t n, int64_t cap, int64_t *count,
                    void *scratch) {
    if (n >= (int64_t)1 << 31) {
        set_error("ordered_compact: %lld elements exceed int32 indices", (long long)n);
        return MICV_EINVAL;
    }
    const int nchunks = (int)((n + kChunk1 - 1) / kChunk1);
    unsigned long long *status = nullptr;
    unsigned *counters = nullptr;
    // Up to 256 chunks (1 M elements) every chunk reaches chunk 0 in ONE look-back step -- it just adds up its
    // predecessors' counts -- and the launch costs less than three (VGA corner list 36 -> 27 us).  Beyond that the
    // walk back to the spreading running totals takes several dependent steps for the first round of workgroups and
    // measured slower than the three launches (4K: 0.146 vs 0.104 ms; 1080p edge points: 46 vs 34 us), unless forced
    // (MICV_OPT_COMPACT_3PASS = -1, tests).
    const int opt = ctx->opt[MICV_OPT_COMPACT_3PASS];
    if (nchunks > 0 && opt <= 0 && (nchunks <= 256 || opt < 0) && ctx->compact_state(s, nchunks, &status, &counters) == MICV_OK) {
        compact_onepass_kernel<<<nchunks, 256, 0, s>>>(pred, emit, n, nchunks, status, counters, cap, count);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    return ordered_compact3(s, pred, emit, n, cap, count, scratch);
}

}  // namespace micv
