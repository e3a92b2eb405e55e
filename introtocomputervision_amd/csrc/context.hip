// context.hip -- context, errors, timer, warm-up (the common/ helpers of the reference, a17).
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

namespace micv {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// cv::getGaussianKernel(n, sigma, CV_32F), sigma > 0: taps evaluated in double, cast to
// float, summed in double, normalised by 1/sum in double, cast to float.
void gaussian_taps(int n, double sigma, Taps *out) {
    const double scale2x = -0.5 / (sigma * sigma);
    double sum = 0.0;
    out->n = n;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        out->k[i] = static_cast<float>(std::exp(scale2x * x * x));
        sum += out->k[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) out->k[i] = static_cast<float>(out->k[i] * sum);
}

// cv::getDerivKernels (normalize=false): binomial smoothing passes then `order` differences.
int sobel_taps(int ksize, int order, Taps *out) {
    if (ksize == 1 && order > 0) ksize = 3;
    if (ksize < 1 || ksize > 31 || (ksize & 1) == 0 || order < 0 || order > 2) return -1;
    int ker[40] = {0};
    if (ksize == 1) {
        ker[0] = 1;
    } else if (ksize == 3) {
        static const int k3[3][3] = {{1, 2, 1}, {-1, 0, 1}, {1, -2, 1}};
        for (int i = 0; i < 3; i++) ker[i] = k3[order][i];
    } else {
        ker[0] = 1;
        for (int pass = 0; pass < ksize - order - 1; pass++) {
            int carry = ker[0];
            for (int j = 1; j <= ksize; j++) {
                const int next = ker[j] + ker[j - 1];
                ker[j - 1] = carry;
                carry = next;
            }
        }
        for (int pass = 0; pass < order; pass++) {
            int carry = -ker[0];
            for (int j = 1; j <= ksize; j++) {
                const int next = ker[j - 1] - ker[j];
                ker[j - 1] = carry;
                carry = next;
            }
        }
    }
    out->n = ksize;
    for (int i = 0; i < ksize; i++) out->k[i] = static_cast<float>(ker[i]);
    return ksize;
}

// common::warmup's kernel (CudaWarmup.cu:5-12): a little arithmetic, no memory traffic.
__global__ void warmup_kernel() {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a = 1.f * tid, b = 2.f * tid;
    asm volatile("" ::"v"(a), "v"(b));
}

}  // namespace micv

int micv_ctx::reserve(size_t bytes, void **out) {
    if (bytes > arena_bytes) {
        MICV_HIP(hipSetDevice(device));
        MICV_HIP(hipDeviceSynchronize());
        if (arena) MICV_HIP(hipFree(arena));
        arena = nullptr;
        arena_bytes = 0;
        const size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
        MICV_HIP(hipMalloc(&arena, want));
        arena_bytes = want;
    }
    *out = arena;
    return MICV_OK;
}

int micv_ctx::stereo_flag_word(unsigned **out) {
    if (!stereo_flag) {
        void *p = nullptr;
        MICV_HIP(hipSetDevice(device));
        MICV_HIP(hipMalloc(&p, 256));
        hipError_t e = hipMemset(p, 0, 256);
        if (e == hipSuccess) e = hipDeviceSynchronize();  // (null-stream fill: finish it before any stream reads the word)
        if (e != hipSuccess) {
            (void)hipFree(p);
            MICV_HIP(e);
        }
        stereo_flag = static_cast<unsigned *>(p);
    }
    *out = stereo_flag;
    return MICV_OK;
}

int micv_ctx::wave_slots(int waves_per_simd) {
    if (!cu_count) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
        cu_count = n;
    }
    return cu_count * 4 * waves_per_simd;
}

int micv_ctx::prof_begin(int level, hipStream_t s) {
    if (!profile) return MICV_OK;
    if (prof[level].size() >= kMaxProfPairs) {  // bounded: stop recording, keep what there is
        prof_open[level] = false;
        return MICV_OK;
    }
    hipEvent_t a, b;
    MICV_HIP(hipEventCreate(&a));
    hipError_t e = hipEventCreate(&b);
    if (e != hipSuccess) {
        (void)hipEventDestroy(a);
        MICV_HIP(e);
    }
    prof[level].emplace_back(a, b);
    prof_open[level] = true;
    MICV_HIP(hipEventRecord(a, s));
    return MICV_OK;
}
int micv_ctx::fork(hipStream_t s, int n) {
    if (!ev_fork) MICV_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    MICV_HIP(hipEventRecord(ev_fork, s));
    for (int i = 0; i < n; i++) {
        if (!aux_stream[i]) {
            MICV_HIP(hipStreamCreateWithFlags(&aux_stream[i], hipStreamNonBlocking));
            MICV_HIP(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
        }
        MICV_HIP(hipStreamWaitEvent(aux_stream[i], ev_fork, 0));
    }
    return MICV_OK;
}
int micv_ctx::join(hipStream_t s, int n) {
    for (int i = 0; i < n; i++) {
        MICV_HIP(hipEventRecord(ev_join[i], aux_stream[i]));
        MICV_HIP(hipStreamWaitEvent(s, ev_join[i], 0));
    }
    return MICV_OK;
}
int micv_ctx::prof_end(int level, hipStream_t s) {
    if (!profile || !prof_open[level]) return MICV_OK;
    prof_open[level] = false;
    MICV_HIP(hipEventRecord(prof[level].back().second, s));
    return MICV_OK;
}

struct micv_timer {
    hipEvent_t start, stop;
};

void *micv_ctx::io_acquire(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    IoBlock *best = nullptr;
    for (auto &b : io_cache)
        if (!b.busy && b.bytes >= bytes && b.bytes <= 2 * bytes && (!best || b.bytes < best->bytes)) best = &b;
    if (best) {
        best->busy = true;
        return best->p;
    }
    if (io_cache.size() >= 24) {  // shapes keep changing: drop what is idle
        for (size_t i = 0; i < io_cache.size();) {
            if (!io_cache[i].busy) {
                (void)hipFree(io_cache[i].p);
                io_cache.erase(io_cache.begin() + i);
            } else {
                i++;
            }
        }
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    io_cache.push_back({p, bytes, true});
    return p;
}

void micv_ctx::io_release(void *p) {
    for (auto &b : io_cache)
        if (b.p == p) b.busy = false;
}

extern "C" {

const char *micv_version(void) { return "micv 0.1 (gfx950)"; }
const char *micv_last_error(void) { return micv::g_err; }

int micv_ctx_create(int device, micv_ctx **out) {
    MICV_REQUIRE(out != nullptr, "micv_ctx_create: out is null");
    int n = 0;
    MICV_HIP(hipGetDeviceCount(&n));
    MICV_REQUIRE(device >= 0 && device < n, "micv_ctx_create: device %d out of range (%d visible)",
                 device, n);
    MICV_HIP(hipSetDevice(device));
    micv_ctx *c = new (std::nothrow) micv_ctx();
    if (!c) {
        micv::set_error("micv_ctx_create: host allocation failed");
        return MICV_ENOMEM;
    }
    c->device = device;
    hipError_t e = hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault);
    if (e != hipSuccess) {
        micv::set_error("hipHostMalloc failed: %s", hipGetErrorString(e));
        delete c;
        return MICV_EHIP;
    }
    *out = c;
    return MICV_OK;
}

void micv_ctx_destroy(micv_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();  // launches in flight may still use the arena, the schedules or the ticket slots
    (void)micv_profile_reset(ctx);
    if (ctx->stamps) (void)hipFree(ctx->stamps);
    for (int i = 0; i < 3; i++) {
        if (ctx->aux_stream[i]) (void)hipStreamDestroy(ctx->aux_stream[i]);
        if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (auto &b : ctx->io_cache) (void)hipFree(b.p);
    for (auto &e : ctx->lk_sched) (void)hipFree(e.dev);
    if (ctx->lk_tickets) (void)hipFree(ctx->lk_tickets);
    if (ctx->stereo_flag) (void)hipFree(ctx->stereo_flag);
    for (auto &c : ctx->compact_slots)
        if (c.buf) (void)hipFree(c.buf);
    for (void *t : ctx->trig_tables)
        if (t) (void)hipFree(t);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    delete ctx;
}

// Ticket counters of the streamed level launch.  One slot PER STREAM (round-robin slots could put
// two launches that run at the same time on different streams onto one slot, where they would take tickets
// off each other): launches on one stream are ordered, so they may share their slot -- the last workgroup
// out of a launch zeroes it for the next.  A context that meets more than kLkTicketSlots different streams
// gets MICV_EUNSUPPORTED and the caller takes the plain launch.
int micv_ctx::lk_ticket_slot(hipStream_t stream, unsigned **out) {
    constexpr int kWords = 16;
    if (!lk_tickets) {
        void *p = nullptr;
        MICV_HIP(hipMalloc(&p, kLkTicketSlots * kWords * sizeof(unsigned)));
        hipError_t e = hipMemset(p, 0, kLkTicketSlots * kWords * sizeof(unsigned));
        // (the fill runs on the null stream, which non-blocking streams do not wait for: finish it before any
        // slot is handed out)
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            (void)hipFree(p);
            MICV_HIP(e);
        }
        lk_tickets = static_cast<unsigned *>(p);
    }
    int slot = -1;
    for (int i = 0; i < lk_ticket_used; i++)
        if (lk_ticket_stream[i] == stream) slot = i;
    if (slot < 0) {
        if (lk_ticket_used >= kLkTicketSlots) return MICV_EUNSUPPORTED;
        slot = lk_ticket_used++;
        lk_ticket_stream[slot] = stream;
    }
    *out = lk_tickets + (size_t)slot * kWords;
    return MICV_OK;
}

// One slot per stream, as for the tickets above: launches on one stream are ordered and each leaves the state
// zeroed.  Growing a slot frees the old buffer, which waits for the device (hipFree), so no launch still uses it.
int micv_ctx::compact_state(hipStream_t stream, int nchunks, unsigned long long **status, unsigned **counters) {
    int slot = -1;
    for (int i = 0; i < compact_used; i++)
        if (compact_slots[i].stream == stream) slot = i;
    if (slot < 0) {
        if (compact_used >= kLkTicketSlots) return MICV_EUNSUPPORTED;
        slot = compact_used++;
        compact_slots[slot].stream = stream;
    }
    CompactSlot &c = compact_slots[slot];
    if (c.chunks < nchunks) {
        const int want = nchunks < 1024 ? 1024 : nchunks + nchunks / 2;
        void *p = nullptr;
        MICV_HIP(hipMalloc(&p, 16 + (size_t)want * 8));
        // on the caller's stream: hipMemset runs on the null stream, which a non-blocking stream does not wait for --
        // the launch that follows could start first and have its state wiped under it
        hipError_t e = hipMemsetAsync(p, 0, 16 + (size_t)want * 8, stream);
        if (e != hipSuccess) {
            (void)hipFree(p);
            MICV_HIP(e);
        }
        if (c.buf) (void)hipFree(c.buf);
        c.buf = p;
        c.chunks = want;
    }
    *counters = static_cast<unsigned *>(c.buf);
    *status = reinterpret_cast<unsigned long long *>(static_cast<char *>(c.buf) + 16);
    return MICV_OK;
}

size_t micv_ctx_scratch_bytes(const micv_ctx *ctx) { return ctx ? ctx->arena_bytes : 0; }

int micv_device_malloc(micv_ctx *ctx, size_t bytes, void **out) {
    MICV_REQUIRE(ctx && out, "micv_device_malloc: null argument");
    MICV_HIP(hipSetDevice(ctx->device));
    MICV_HIP(hipMalloc(out, bytes ? bytes : 1));
    return MICV_OK;
}
void micv_device_free(micv_ctx *ctx, void *p) {
    if (!p) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipFree(p);
}
int micv_memcpy2d_h2d(micv_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch,
                      size_t width_bytes, int rows) {
    MICV_REQUIRE(ctx && dst && src && rows > 0 && width_bytes > 0 && dpitch >= width_bytes && spitch >= width_bytes,
                 "micv_memcpy2d_h2d: bad argument");
    MICV_HIP(hipSetDevice(ctx->device));
    MICV_HIP(hipMemcpy2D(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyHostToDevice));
    return MICV_OK;
}
int micv_memcpy2d_d2h(micv_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch,
                      size_t width_bytes, int rows) {
    MICV_REQUIRE(ctx && dst && src && rows > 0 && width_bytes > 0 && dpitch >= width_bytes && spitch >= width_bytes,
                 "micv_memcpy2d_d2h: bad argument");
    MICV_HIP(hipSetDevice(ctx->device));
    MICV_HIP(hipMemcpy2D(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyDeviceToHost));
    return MICV_OK;
}

int micv_ctx_set_option(micv_ctx *ctx, int option, int value) {
    MICV_REQUIRE(ctx != nullptr, "micv_ctx_set_option: ctx is null");
    MICV_REQUIRE(option >= 1 && option < MICV_OPT_COUNT, "micv_ctx_set_option: unknown option %d", option);
    if (option == MICV_OPT_LK_STREAM_GROUPS)
        MICV_REQUIRE(value >= 0 && value <= 4, "micv_ctx_set_option: stream groups must be 0..4");
    if (option == MICV_OPT_STEREO_ROWS)
        MICV_REQUIRE(value == 0 || value == 8 || value == 10, "micv_ctx_set_option: stereo rows must be 0, 8 or 10");
    if (option == MICV_OPT_LK_SHORT_TILES)
        MICV_REQUIRE(value >= -1 && value <= (1 << 20), "micv_ctx_set_option: short-tile limit must be -1..2^20");
    if (option == MICV_OPT_LK_CHAIN)
        MICV_REQUIRE(value >= -1 && value <= 32, "micv_ctx_set_option: chain length must be -1..32");
    if (option == MICV_OPT_LK_STREAM)
        MICV_REQUIRE(value >= 0 && value <= 1, "micv_ctx_set_option: streamed launch must be 0 or 1");
    if (option == MICV_OPT_LK_TALL_TILES)
        MICV_REQUIRE(value >= -1 && value <= 3, "micv_ctx_set_option: tall tiles must be -1..3");
    if (option == MICV_OPT_COMPACT_3PASS)
        MICV_REQUIRE(value >= -1 && value <= 1, "micv_ctx_set_option: compaction form must be -1, 0 or 1");
    if (option == MICV_OPT_LK_DIRECT_LEVELS)
        MICV_REQUIRE(value >= 0 && value <= 15, "micv_ctx_set_option: direct levels must be 0..15");
    if (option == MICV_OPT_LK_BUILD_OVERLAP)
        MICV_REQUIRE(value >= -1 && value <= 1, "micv_ctx_set_option: build overlap must be -1 (never), 0 (single pairs) or 1 (every batch)");
    if (option == MICV_OPT_LK_SPLIT)
        MICV_REQUIRE(value >= 0 && value <= 3, "micv_ctx_set_option: split launch must be 0 (never) .. 3");
    if (option == MICV_OPT_LK_STRIP)
        MICV_REQUIRE(value >= 0 && value <= 8192 + 4096 && (value & 8191) <= 4096, "micv_ctx_set_option: strip segments must be 0 (off) .. 4096 blocks of 16 rows (+ 8192: launches of 4096 tiles or more only)");
    if (option == MICV_OPT_STEREO_EXACT)
        MICV_REQUIRE(value >= -1 && value <= 0, "micv_ctx_set_option: exact-sum stereo must be 0 (automatic) or -1 (never)");
    ctx->opt[option] = value;
    return MICV_OK;
}
int micv_ctx_get_option(const micv_ctx *ctx, int option, int *value) {
    MICV_REQUIRE(ctx && value, "micv_ctx_get_option: null argument");
    MICV_REQUIRE(option >= 1 && option < MICV_OPT_COUNT, "micv_ctx_get_option: unknown option %d", option);
    *value = ctx->opt[option];
    return MICV_OK;
}

int micv_profile_enable(micv_ctx *ctx, int on) {
    MICV_REQUIRE(ctx != nullptr, "micv_profile_enable: ctx is null");
    ctx->profile = on != 0;
    return MICV_OK;
}

int micv_profile_lk_phases(micv_ctx *ctx, int enable, uint64_t *ticks16) {
    MICV_REQUIRE(ctx != nullptr, "micv_profile_lk_phases: ctx is null");
#ifndef MICV_DIAG
    if (enable) {
        micv::set_error("micv_profile_lk_phases: phase stamps need a -DMICV_DIAG build of libmicv.so");
        return MICV_EUNSUPPORTED;
    }
#endif
    MICV_HIP(hipSetDevice(ctx->device));
    if (ticks16) {
        for (int i = 0; i < 16; i++) ticks16[i] = 0;
        if (ctx->stamps) {
            MICV_HIP(hipDeviceSynchronize());
            MICV_HIP(hipMemcpy(ticks16, ctx->stamps, 16 * 8, hipMemcpyDeviceToHost));
        }
    }
    if (enable && !ctx->stamps) MICV_HIP(hipMalloc(reinterpret_cast<void **>(&ctx->stamps), 16 * 8));
    if (!enable && ctx->stamps) {
        MICV_HIP(hipDeviceSynchronize());
        MICV_HIP(hipFree(ctx->stamps));
        ctx->stamps = nullptr;
    }
    if (ctx->stamps) MICV_HIP(hipMemset(ctx->stamps, 0, 16 * 8));
    return MICV_OK;
}

int micv_profile_reset(micv_ctx *ctx) {
    MICV_REQUIRE(ctx != nullptr, "micv_profile_reset: ctx is null");
    for (auto &v : ctx->prof) {
        for (auto &e : v) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        v.clear();
    }
    return MICV_OK;
}

int micv_profile_lk_pairs(micv_ctx *ctx, int *pairs_per_launch) {
    MICV_REQUIRE(ctx && pairs_per_launch, "micv_profile_lk_pairs: null argument");
    *pairs_per_launch = ctx->prof_pairs;
    return MICV_OK;
}

int micv_profile_lk_level(micv_ctx *ctx, int level, double *total_ms, int64_t *launches) {
    MICV_REQUIRE(ctx && total_ms && launches, "micv_profile_lk_level: null argument");
    MICV_REQUIRE(level >= 0 && level < 16, "micv_profile_lk_level: bad level %d", level);
    double sum = 0.0;
    for (auto &e : ctx->prof[level]) {
        MICV_HIP(hipEventSynchronize(e.second));
        float ms = 0.f;
        MICV_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)ctx->prof[level].size();
    return MICV_OK;
}

int micv_warmup(micv_ctx *ctx, micv_stream stream) {
    MICV_REQUIRE(ctx != nullptr, "micv_warmup: ctx is null");
    MICV_HIP(hipSetDevice(ctx->device));
    micv::warmup_kernel<<<dim3(10), dim3(64), 0, static_cast<hipStream_t>(stream)>>>();
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

size_t micv_div_round_up(size_t num, size_t denom) {
    // Utils.h:12-15 computes this in float on purpose-or-not; keep its rounding behaviour.
    const float q = std::ceil(static_cast<float>(num) / static_cast<float>(denom));
    return std::max<size_t>(1, static_cast<size_t>(q));
}

int micv_timer_create(micv_timer **out) {
    MICV_REQUIRE(out != nullptr, "micv_timer_create: out is null");
    micv_timer *t = new (std::nothrow) micv_timer();
    if (!t) {
        micv::set_error("micv_timer_create: host allocation failed");
        return MICV_ENOMEM;
    }
    hipError_t e = hipEventCreate(&t->start);
    if (e == hipSuccess) {
        e = hipEventCreate(&t->stop);
        if (e != hipSuccess) (void)hipEventDestroy(t->start);
    }
    if (e != hipSuccess) {
        micv::set_error("micv_timer_create: hipEventCreate failed: %s", hipGetErrorString(e));
        delete t;
        return MICV_EHIP;
    }
    *out = t;
    return MICV_OK;
}
int micv_timer_start(micv_timer *t, micv_stream stream) {
    MICV_REQUIRE(t != nullptr, "micv_timer_start: timer is null");
    MICV_HIP(hipEventRecord(t->start, static_cast<hipStream_t>(stream)));
    return MICV_OK;
}
int micv_timer_stop(micv_timer *t, micv_stream stream) {
    MICV_REQUIRE(t != nullptr, "micv_timer_stop: timer is null");
    MICV_HIP(hipEventRecord(t->stop, static_cast<hipStream_t>(stream)));
    MICV_HIP(hipEventSynchronize(t->stop));
    return MICV_OK;
}
int micv_timer_elapsed_ms(micv_timer *t, float *ms) {
    MICV_REQUIRE(t != nullptr && ms != nullptr, "micv_timer_elapsed_ms: null argument");
    MICV_HIP(hipEventElapsedTime(ms, t->start, t->stop));
    return MICV_OK;
}
void micv_timer_destroy(micv_timer *t) {
    if (!t) return;
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
}

}  // extern "C"
