// common.hpp -- shared host/device helpers for libmicv (gfx950 only).
//
// Arithmetic contract: the library is compiled with -ffp-contract=off, so a
// fused multiply-add happens exactly where the source says fmaf(); `a * b + c` is an
// unfused multiply followed by an add.  That is what lets the HIP path match the CPU
// oracle bit for bit.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <utility>
#include <vector>
#include <cstdint>
#include <cstdio>

#include "../../include/mi_cv.h"

namespace micv {

void set_error(const char *fmt, ...);

#define MICV_HIP(expr)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ::micv::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                              __FILE__, __LINE__);                                       \
            return e_ == hipErrorOutOfMemory ? MICV_ENOMEM : MICV_EHIP;                  \
        }                                                                                \
    } while (0)

#define MICV_REQUIRE(cond, ...)                                                          \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            ::micv::set_error(__VA_ARGS__);                                              \
            return MICV_EINVAL;                                                          \
        }                                                                                \
    } while (0)

#define MICV_TRY(expr)                                                                   \
    do {                                                                                 \
        int rc_ = (expr);                                                                \
        if (rc_ != MICV_OK) return rc_;                                                  \
    } while (0)

// Launch check: catches bad launch configurations without synchronising.
#define MICV_LAUNCH_CHECK() MICV_HIP(hipGetLastError())

constexpr int kMaxTaps = 64;  // separable kernels are passed by value in the kernarg block
constexpr int kMaxWin = 63;

struct Taps {
    float k[kMaxTaps];
    int n;
};

// A strided single-channel image (stride in ELEMENTS).
template <typename T>
struct Img {
    T *p;
    int rows, cols;
    int stride;
    __host__ __device__ T &at(int y, int x) const { return p[(size_t)y * stride + x]; }
};
using ImgF = Img<float>;
using ImgCF = Img<const float>;

static inline unsigned cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }

// cv::borderInterpolate(p, len, BORDER_REFLECT_101).
__host__ __device__ inline int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}
__host__ __device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Host: cv::getGaussianKernel(n, sigma, CV_32F) for sigma > 0.
void gaussian_taps(int n, double sigma, Taps *out);
// Host: cv::getDerivKernels integer taps for one direction; returns tap count or -1.
int sobel_taps(int ksize, int order, Taps *out);

}  // namespace micv

// The context: device ordinal + one growable scratch arena.
struct micv_ctx {
    int device = 0;
    int opt[MICV_OPT_COUNT] = {0};  // micv_ctx_set_option
    void *arena = nullptr;
    size_t arena_bytes = 0;
    void *pinned = nullptr;  // small pinned staging block for counts
    // Per-launch timing of the pyramid-level kernels (micv_profile_*): event pairs per level.
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof[16];
    bool prof_open[16] = {false};
    static constexpr size_t kMaxProfPairs = 4096;  // per level; recording stops there until a reset
    unsigned long long *stamps = nullptr;  // 16 device counters (micv_profile_lk_phases)
    int prof_pairs = 0;                    // frame pairs per profiled level launch
    // Device blocks of the host-pointer entry points (host_api.hip), kept between calls so that a
    // repeated call of the same shape does no hipMalloc / hipFree.  Freed with the context.
    struct IoBlock {
        void *p;
        size_t bytes;
        bool busy;
    };
    std::vector<IoBlock> io_cache;
    // Tile-chain schedules of the fused LK level kernel (lk_fused.hip), one per launch shape.
    struct LkSched {
        int rows, cols, batch, r, max_chain, th;
        void *dev;
        int nblocks;
    };
    std::vector<LkSched> lk_sched;
    // Ticket counters of the streamed LK level launch: 16 slots of 16 words (8 per-XCD counters, the
    // count of workgroups that have left, padding), zero when idle (the launch resets its own slot);
    // one slot per stream the context has launched on (context.hip).
    unsigned *lk_tickets = nullptr;
    static constexpr int kLkTicketSlots = 16;
    hipStream_t lk_ticket_stream[kLkTicketSlots] = {};
    int lk_ticket_used = 0;
    int lk_ticket_slot(hipStream_t stream, unsigned **out);
    // State of the one-launch ordered compaction (compact.hpp): per stream, all zero between launches
    struct CompactSlot {
        hipStream_t stream = nullptr;
        void *buf = nullptr;  // 16 B of counters, then one status word per chunk
        int chunks = 0;
    };
    CompactSlot compact_slots[kLkTicketSlots];
    int compact_used = 0;
    int compact_state(hipStream_t stream, int nchunks, unsigned long long **status, unsigned **counters);
    // Eligibility word of the exact-sum stereo path (stereo_exact.hip): the pack pre-pass of call number `stereo_epoch`
    // stores that number when an image is not 8-bit-valued; zero-filled once, never reset (epochs only grow).
    unsigned *stereo_flag = nullptr;
    unsigned stereo_epoch = 0;
    int stereo_flag_word(unsigned **out);
    int cu_count = 0;  // multiProcessorCount, read once
    int wave_slots(int waves_per_simd);
    // Hough trig tables (hough.hip), uploaded once per context: [0] theta = -90.., [1] theta = 0..
    void *trig_tables[2] = {nullptr, nullptr};
    void *io_acquire(size_t bytes);
    void io_release(void *p);
    // Up to 3 auxiliary streams for group-parallel pyramid chains (lk.hip); fork makes them wait
    // for everything enqueued on `s` so far, join makes `s` wait for them.
    hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    int fork(hipStream_t s, int n);
    int join(hipStream_t s, int n);
    int prof_begin(int level, hipStream_t s);
    int prof_end(int level, hipStream_t s);
    // Returns scratch of at least `bytes` (256-B aligned). Growing synchronises the device.
    int reserve(size_t bytes, void **out);
};

namespace micv {
// Bump allocator over ctx scratch: compute total first, reserve once, then carve.
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base(static_cast<char *>(b)) {}
    template <typename T>
    T *take(size_t count) {
        T *p = reinterpret_cast<T *>(base + off);
        off += (count * sizeof(T) + 255) & ~size_t(255);
        return p;
    }
    static size_t need(size_t count, size_t elem) { return (count * elem + 255) & ~size_t(255); }
};

inline bool stride_ok(size_t stride_bytes, int cols, size_t elem) {
    return stride_bytes % elem == 0 && stride_bytes / elem >= (size_t)cols &&
           stride_bytes / elem < (size_t)1 << 30;
}
}  // namespace micv
