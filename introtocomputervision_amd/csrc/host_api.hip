// host_api.hip -- `_host` flavours: host pointers in, host pointers out, synchronous.
// This is the behaviour of the reference's cv::Mat functions (upload -> kernel -> sync ->
// download inside every call, e.g. Harris.cu:118-158, Pyramids.cu:45-72); it is PCIe-bound
// by construction.  Device buffers are allocated per call like the reference's GpuMats.
#include <atomic>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"

namespace micv {

// RAII device allocation; `ok()` reports failure through set_error.
// Device block of a host-pointer call, taken from the context's cache (no hipMalloc / hipFree on a
// repeated call of the same shape).  `io_ctx` is set by HOST_PROLOGUE.
static thread_local micv_ctx *io_ctx = nullptr;
struct DevBuf {
    void *p = nullptr;
    micv_ctx *owner;
    explicit DevBuf(size_t bytes) : owner(io_ctx) { p = owner ? owner->io_acquire(bytes) : nullptr; }
    ~DevBuf() {
        if (p) owner->io_release(p);
    }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

#define MICV_ALLOC_OK(buf)                                        \
    do {                                                          \
        if (!(buf).p) {                                           \
            ::micv::set_error("device allocation failed");        \
            return MICV_ENOMEM;                                   \
        }                                                         \
    } while (0)

// A continuous cv::Mat (step == row bytes: what cv::Mat::create gives) goes as ONE linear copy; only a
// pitched view (an ROI) needs the 2-D form.  Pageable host memory is fine: the runtime pins the pages
// for the transfer and reaches the same 53 GB/s as hipHostMalloc'd memory on this platform
// (tools/probes/pcie_probe.py), so there is no staging ring to copy through.
static int up2d(void *dst, const void *src, size_t sstride, size_t row_bytes, int rows,
                hipStream_t s) {
    if (sstride == row_bytes || rows == 1)
        MICV_HIP(hipMemcpyAsync(dst, src, row_bytes * (size_t)rows, hipMemcpyHostToDevice, s));
    else
        MICV_HIP(hipMemcpy2DAsync(dst, row_bytes, src, sstride, row_bytes, rows, hipMemcpyHostToDevice, s));
    return MICV_OK;
}
static int down2d(void *dst, size_t dstride, const void *src, size_t row_bytes, int rows,
                  hipStream_t s) {
    if (dstride == row_bytes || rows == 1)
        MICV_HIP(hipMemcpyAsync(dst, src, row_bytes * (size_t)rows, hipMemcpyDeviceToHost, s));
    else
        MICV_HIP(hipMemcpy2DAsync(dst, dstride, src, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, s));
    return MICV_OK;
}

}  // namespace micv

using namespace micv;

// Every `_host` function enqueues asynchronous copies; whichever way it returns (an error in a
// later step included) the stream is drained first, so no D2H copy into the caller's buffer is
// still in flight and no cached device block is handed out again while something uses it.
//
// Kernel-timing log lines: the reference brackets its kernels with a GpuTimer and
// logs "<kernel> execution took {} ms" (Harris.cu:144-155, DisparitySSD.cu:192-203, Hough.cu:277-289,
// Pyramids.cu:61-69).  With a sink registered (micv_set_kernel_log) the `_host` entry point of each of
// those functions records an event pair around its device call and, after the final synchronisation,
// hands (the reference's kernel name, milliseconds) to the sink; without one nothing is recorded.
static std::atomic<micv_kernel_log_fn> g_log_fn{nullptr};
static std::atomic<void *> g_log_user{nullptr};

struct HostSync {
    hipStream_t s;
    const char *name = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    void begin(const char *kernel) {
        if (!g_log_fn.load(std::memory_order_relaxed)) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
        name = kernel;
        (void)hipEventRecord(e0, s);
    }
    void end() {
        if (name) (void)hipEventRecord(e1, s);
    }
    ~HostSync() {
        (void)hipStreamSynchronize(s);
        if (name) {
            float ms = 0.f;
            const micv_kernel_log_fn fn = g_log_fn.load(std::memory_order_relaxed);
            if (fn && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) fn(name, ms, g_log_user.load(std::memory_order_relaxed));
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

// The device call of a `_host` function, bracketed for the kernel log (a no-op without a sink).
#define MICV_TIMED(kernel, call)       \
    do {                               \
        host_sync_.begin(kernel);      \
        const int rc_timed_ = (call);  \
        host_sync_.end();              \
        if (rc_timed_ != MICV_OK) return rc_timed_; \
    } while (0)

#define HOST_PROLOGUE(fn)                                    \
    MICV_REQUIRE(ctx != nullptr, fn ": ctx is null");        \
    MICV_HIP(hipSetDevice(ctx->device));                     \
    ::micv::io_ctx = ctx;                                    \
    hipStream_t s = nullptr;                                 \
    HostSync host_sync_{s}

extern "C" {

int micv_set_kernel_log(micv_kernel_log_fn fn, void *user) {
    g_log_user.store(user, std::memory_order_relaxed);
    g_log_fn.store(fn, std::memory_order_release);
    return MICV_OK;
}

int micv_lk_flow_pyr_host(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                          size_t stride, int win, int levels, float *u, float *v, size_t ostride) {
    HOST_PROLOGUE("micv_lk_flow_pyr_host");
    MICV_REQUIRE(prev && next && u && v && rows > 0 && cols > 0, "micv_lk_flow_pyr_host: bad argument");
    MICV_REQUIRE(stride_ok(stride, cols, 4) && stride_ok(ostride, cols, 4),
                 "micv_lk_flow_pyr_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf dp(n), dn(n), du(n), dv(n);
    MICV_ALLOC_OK(dp); MICV_ALLOC_OK(dn); MICV_ALLOC_OK(du); MICV_ALLOC_OK(dv);
    MICV_TRY(up2d(dp.p, prev, stride, rb, rows, s));
    MICV_TRY(up2d(dn.p, next, stride, rb, rows, s));
    MICV_TRY(micv_lk_flow_pyr_dev(ctx, dp.as<float>(), dn.as<float>(), rows, cols, rb, win, levels,
                                  du.as<float>(), dv.as<float>(), rb, s));
    MICV_TRY(down2d(u, ostride, du.p, rb, rows, s));
    MICV_TRY(down2d(v, ostride, dv.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

/* lk::calcOpticalFlowPyr on frames as the ps5 driver hands them over (denseLKWrapper passes the
 * COLOUR frames, Solution.cpp:63; makeGaussianPyramid converts, Pyramids.cpp:9-15): one upload of
 * the interleaved frames, grey conversion on the device, then the pyramid chain. */
int micv_lk_flow_pyr_frames_host(micv_ctx *ctx, const void *prev, const void *next, int rows, int cols,
                                 size_t stride, int channels, int depth, int win, int levels, float *u,
                                 float *v, size_t ostride) {
    HOST_PROLOGUE("micv_lk_flow_pyr_frames_host");
    MICV_REQUIRE(prev && next && u && v && rows > 0 && cols > 0, "micv_lk_flow_pyr_frames_host: bad argument");
    MICV_REQUIRE((channels == 1 || channels == 3 || channels == 4) &&
                     (depth == MICV_DEPTH_8U || depth == MICV_DEPTH_32F),
                 "micv_lk_flow_pyr_frames_host: frames must be 1/3/4-channel 8U or 32F");
    const size_t es = depth == MICV_DEPTH_8U ? 1 : 4, srb = (size_t)cols * channels * es;
    MICV_REQUIRE(stride >= srb && stride_ok(ostride, cols, 4), "micv_lk_flow_pyr_frames_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf cp(srb * rows), cn(srb * rows), dp(n), dn(n), du(n), dv(n);
    MICV_ALLOC_OK(cp); MICV_ALLOC_OK(cn); MICV_ALLOC_OK(dp); MICV_ALLOC_OK(dn); MICV_ALLOC_OK(du); MICV_ALLOC_OK(dv);
    MICV_TRY(up2d(cp.p, prev, stride, srb, rows, s));
    MICV_TRY(up2d(cn.p, next, stride, srb, rows, s));
    MICV_TRY(micv_to_gray_f32_dev(ctx, cp.p, rows, cols, srb, channels, depth, dp.as<float>(), rb, s));
    MICV_TRY(micv_to_gray_f32_dev(ctx, cn.p, rows, cols, srb, channels, depth, dn.as<float>(), rb, s));
    MICV_TRY(micv_lk_flow_pyr_dev(ctx, dp.as<float>(), dn.as<float>(), rows, cols, rb, win, levels,
                                  du.as<float>(), dv.as<float>(), rb, s));
    MICV_TRY(down2d(u, ostride, du.p, rb, rows, s));
    MICV_TRY(down2d(v, ostride, dv.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

/* lk::calcOpticalFlowPyr over a SEQUENCE of frames -- pairs (0, 1), (1, 2), ... as the ps5 driver walks a directory
 * (ps5_cpp/lib/Config.cpp:17-46, src/Solution.cpp:255-285: frame t is `next` of one call and `prev` of the following one).
 * A per-pair `_host` call moves four images over PCIe one after the other around its kernels (0.745 ms per 1080p pair,
 * r05).  Here every frame is uploaded ONCE, and the legs run side by side:
 *   upload     frame t + 2 -> raw block -> grey f32 (device conversion), ring of three frames; the calling thread
 *   compute    pair t + 1: the pyramid chain (micv_lk_flow_pyr_dev) on its own stream, ring of three flow outputs
 *   download   pair t: u, v -> the caller's buffers, a thread and a stream of their own
 * The caller's images are pageable memory, and a copy from / to pageable memory occupies the thread that issues it
 * (streams alone do not overlap such copies: tools/probes/pcie_probe.py) -- hence the download thread.  What r06 tried
 * instead and measured slower per 1080p f32 pair (profiles/r06/host_sequence.md): registering the caller's images in place
 * (hipHostRegister: a fresh registration costs 0.07 ms per image and serialises with the transfers in flight, 0.71 ms),
 * pinned staging rings with host copies on three threads (a host copy into / out of pinned memory runs at 30 GB/s here,
 * 0.27 ms per image: 0.65 ms).  Same bits as the per-pair calls. */
int micv_lk_flow_seq_host(micv_ctx *ctx, const void *const *frames, int nframes, int rows, int cols, size_t stride,
                          int channels, int depth, int win, int levels, float *const *u, float *const *v,
                          size_t ostride) {
    MICV_REQUIRE(ctx != nullptr, "micv_lk_flow_seq_host: ctx is null");
    MICV_REQUIRE(frames && u && v && nframes >= 2 && rows > 0 && cols > 0, "micv_lk_flow_seq_host: bad argument");
    MICV_REQUIRE((channels == 1 || channels == 3 || channels == 4) && (depth == MICV_DEPTH_8U || depth == MICV_DEPTH_32F),
                 "micv_lk_flow_seq_host: frames must be 1/3/4-channel 8U or 32F");
    const size_t es = depth == MICV_DEPTH_8U ? 1 : 4, srb = (size_t)cols * channels * es;
    MICV_REQUIRE(stride >= srb && stride_ok(ostride, cols, 4), "micv_lk_flow_seq_host: bad stride");
    for (int t = 0; t < nframes; t++) MICV_REQUIRE(frames[t] != nullptr, "micv_lk_flow_seq_host: frame %d is null", t);
    for (int t = 0; t + 1 < nframes; t++) MICV_REQUIRE(u[t] && v[t], "micv_lk_flow_seq_host: output %d is null", t);
    MICV_HIP(hipSetDevice(ctx->device));
    ::micv::io_ctx = ctx;
    const bool convert = !(channels == 1 && depth == MICV_DEPTH_32F);
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    constexpr int RING = 3;
    DevBuf raw(convert ? srb * rows : 256), g0(n), g1(n), g2(n), u0(n), u1(n), u2(n), v0(n), v1(n), v2(n);
    MICV_ALLOC_OK(raw); MICV_ALLOC_OK(g0); MICV_ALLOC_OK(g1); MICV_ALLOC_OK(g2);
    MICV_ALLOC_OK(u0); MICV_ALLOC_OK(u1); MICV_ALLOC_OK(u2); MICV_ALLOC_OK(v0); MICV_ALLOC_OK(v1); MICV_ALLOC_OK(v2);
    float *grey[RING] = {g0.as<float>(), g1.as<float>(), g2.as<float>()};
    float *du[RING] = {u0.as<float>(), u1.as<float>(), u2.as<float>()}, *dv[RING] = {v0.as<float>(), v1.as<float>(), v2.as<float>()};
    const int npairs = nframes - 1;

    struct Scope {  // released on every way out, after everything enqueued has finished
        hipStream_t up = nullptr, run = nullptr, down[2] = {nullptr, nullptr};
        std::vector<hipEvent_t> ev;
        ~Scope() {
            for (hipStream_t st : {up, run, down[0], down[1]})
                if (st) (void)hipStreamSynchronize(st);
            for (hipEvent_t e : ev) (void)hipEventDestroy(e);
            for (hipStream_t st : {up, run, down[0], down[1]})
                if (st) (void)hipStreamDestroy(st);
        }
    } st;
    MICV_HIP(hipStreamCreateWithFlags(&st.up, hipStreamNonBlocking));
    MICV_HIP(hipStreamCreateWithFlags(&st.run, hipStreamNonBlocking));
    MICV_HIP(hipStreamCreateWithFlags(&st.down[0], hipStreamNonBlocking));
    MICV_HIP(hipStreamCreateWithFlags(&st.down[1], hipStreamNonBlocking));
    st.ev.reserve((size_t)nframes + (size_t)npairs);
    auto new_event = [&](hipEvent_t *e) -> int {
        MICV_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        st.ev.push_back(*e);
        return MICV_OK;
    };
    std::vector<hipEvent_t> ev_up(nframes), ev_pair(npairs);
    for (auto &e : ev_up) MICV_TRY(new_event(&e));
    for (auto &e : ev_pair) MICV_TRY(new_event(&e));

    // The download threads: pair p's field is copied out once its chain has finished.  `enqueued` / `done[]` order them
    // against the calling thread: a flow block is written again only after its previous content has reached the caller.
    std::mutex mu;
    std::condition_variable cv;
    int enqueued = 0, done[2] = {0, 0};
    bool stop = false, failed = false;
    char down_err[256] = "";
    auto download = [&](int which) {
        (void)hipSetDevice(ctx->device);
        for (int p = 0; p < npairs; p++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return enqueued > p || stop; });
                if (enqueued <= p) return;  // the calling thread gave up
            }
            int rc = hipStreamWaitEvent(st.down[which], ev_pair[p], 0) == hipSuccess ? MICV_OK : MICV_EHIP;
            if (rc == MICV_OK) rc = down2d(u[p], ostride, du[p % RING], rb, rows, st.down[which]);
            if (rc == MICV_OK) rc = down2d(v[p], ostride, dv[p % RING], rb, rows, st.down[which]);
            if (rc == MICV_OK && hipStreamSynchronize(st.down[which]) != hipSuccess) rc = MICV_EHIP;
            std::lock_guard<std::mutex> lk(mu);
            if (rc != MICV_OK && !failed) {
                failed = true;
                snprintf(down_err, sizeof(down_err), "micv_lk_flow_seq_host: download of pair %d failed: %s", p, micv_last_error());
            }
            done[0] = done[1] = p + 1;
            cv.notify_all();
        }
    };
    // (ONE thread for both fields: two -- a stream and a thread per field -- measured slower, 0.48-0.52 against 0.45 ms
    // per pair: copies to pageable memory do not overlap each other either)
    std::thread tu;
    try {
        tu = std::thread(download, 0);
    } catch (const std::exception &e) {  // (no thread to be had: nothing has been enqueued for it yet)
        set_error("micv_lk_flow_seq_host: cannot start the download thread: %s", e.what());
        return MICV_EHIP;
    }
    struct Joiner {  // (declared after Scope: runs first -- the thread is gone before its stream is)
        std::thread &a; std::mutex &mu; std::condition_variable &cv; bool &stop;
        ~Joiner() {
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            if (a.joinable()) a.join();
        }
    } joiner{tu, mu, cv, stop};

    for (int t = 0; t < nframes; t++) {
        // frame t takes the grey block frame t - 3 had: pairs t - 4 and t - 3 read that one
        if (t >= RING) MICV_HIP(hipStreamWaitEvent(st.up, ev_pair[t - RING], 0));
        if (convert) {
            MICV_TRY(up2d(raw.p, frames[t], stride, srb, rows, st.up));
            MICV_TRY(micv_to_gray_f32_dev(ctx, raw.p, rows, cols, srb, channels, depth, grey[t % RING], rb, st.up));
        } else {
            MICV_TRY(up2d(grey[t % RING], frames[t], stride, rb, rows, st.up));
        }
        MICV_HIP(hipEventRecord(ev_up[t], st.up));
        if (t == 0) continue;
        const int p = t - 1;
        if (p >= RING) {  // the flow blocks pair p - 3 used must have reached the caller
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return (done[0] >= p - RING + 1 && done[1] >= p - RING + 1) || failed; });
            if (failed) break;
        }
        MICV_HIP(hipStreamWaitEvent(st.run, ev_up[t], 0));  // (frame t - 1: earlier on the same streams)
        MICV_TRY(micv_lk_flow_pyr_dev(ctx, grey[p % RING], grey[t % RING], rows, cols, rb, win, levels, du[p % RING],
                                      dv[p % RING], rb, st.run));
        MICV_HIP(hipEventRecord(ev_pair[p], st.run));
        {
            std::lock_guard<std::mutex> lk(mu);
            enqueued = p + 1;
        }
        cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return (done[0] >= enqueued && done[1] >= enqueued) || failed; });
        if (failed) {
            set_error("%s", down_err);
            return MICV_EHIP;
        }
        if (enqueued < npairs) return MICV_EHIP;  // (unreachable: every early way out returns above)
    }
    return MICV_OK;
}

int micv_to_gray_f32_host(micv_ctx *ctx, const void *src, int rows, int cols, size_t sstride, int channels,
                          int depth, float *dst, size_t dstride) {
    HOST_PROLOGUE("micv_to_gray_f32_host");
    MICV_REQUIRE(src && dst && rows > 0 && cols > 0, "micv_to_gray_f32_host: bad argument");
    MICV_REQUIRE((channels == 1 || channels == 3 || channels == 4) &&
                     (depth == MICV_DEPTH_8U || depth == MICV_DEPTH_32F),
                 "micv_to_gray_f32_host: source must be 1/3/4-channel 8U or 32F");
    const size_t es = depth == MICV_DEPTH_8U ? 1 : 4, srb = (size_t)cols * channels * es;
    MICV_REQUIRE(sstride >= srb && stride_ok(dstride, cols, 4), "micv_to_gray_f32_host: bad stride");
    const size_t rb = (size_t)cols * 4;
    DevBuf cs(srb * rows), dd(rb * rows);
    MICV_ALLOC_OK(cs); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(cs.p, src, sstride, srb, rows, s));
    MICV_TRY(micv_to_gray_f32_dev(ctx, cs.p, rows, cols, srb, channels, depth, dd.as<float>(), rb, s));
    MICV_TRY(down2d(dst, dstride, dd.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_lk_flow_host(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                      size_t stride, int win, float *u, float *v, size_t ostride) {
    HOST_PROLOGUE("micv_lk_flow_host");
    MICV_REQUIRE(prev && next && u && v && rows > 0 && cols > 0, "micv_lk_flow_host: bad argument");
    MICV_REQUIRE(stride_ok(stride, cols, 4) && stride_ok(ostride, cols, 4),
                 "micv_lk_flow_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf dp(n), dn(n), du(n), dv(n);
    MICV_ALLOC_OK(dp); MICV_ALLOC_OK(dn); MICV_ALLOC_OK(du); MICV_ALLOC_OK(dv);
    MICV_TRY(up2d(dp.p, prev, stride, rb, rows, s));
    MICV_TRY(up2d(dn.p, next, stride, rb, rows, s));
    MICV_TRY(micv_lk_flow_dev(ctx, dp.as<float>(), dn.as<float>(), rows, cols, rb, win,
                              du.as<float>(), dv.as<float>(), rb, s));
    MICV_TRY(down2d(u, ostride, du.p, rb, rows, s));
    MICV_TRY(down2d(v, ostride, dv.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_lk_warp_host(micv_ctx *ctx, const float *src, size_t sstride, const float *du,
                      const float *dv, size_t fstride, int rows, int cols, float *dst,
                      size_t dstride) {
    HOST_PROLOGUE("micv_lk_warp_host");
    MICV_REQUIRE(src && du && dv && dst && rows > 0 && cols > 0, "micv_lk_warp_host: bad argument");
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(fstride, cols, 4) &&
                     stride_ok(dstride, cols, 4),
                 "micv_lk_warp_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf ds(n), dU(n), dV(n), dd(n);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dU); MICV_ALLOC_OK(dV); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, src, sstride, rb, rows, s));
    MICV_TRY(up2d(dU.p, du, fstride, rb, rows, s));
    MICV_TRY(up2d(dV.p, dv, fstride, rb, rows, s));
    MICV_TRY(micv_lk_warp_dev(ctx, ds.as<float>(), rb, dU.as<float>(), dV.as<float>(), rb, rows,
                              cols, dd.as<float>(), rb, s));
    MICV_TRY(down2d(dst, dstride, dd.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_pyr_down_host(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                       float *dst, size_t dstride) {
    HOST_PROLOGUE("micv_pyr_down_host");
    MICV_REQUIRE(src && dst && rows > 0 && cols > 0, "micv_pyr_down_host: bad argument");
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(dstride, cols / 2, 4),
                 "micv_pyr_down_host: bad stride");
    const int dr = rows / 2, dc = cols / 2;
    DevBuf ds((size_t)rows * cols * 4), dd((size_t)dr * dc * 4);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, src, sstride, (size_t)cols * 4, rows, s));
    MICV_TIMED("pyrDownsampleKernel", micv_pyr_down_dev(ctx, ds.as<float>(), rows, cols, (size_t)cols * 4, dd.as<float>(),
                               (size_t)dc * 4, s));
    if (dr > 0 && dc > 0) MICV_TRY(down2d(dst, dstride, dd.p, (size_t)dc * 4, dr, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_pyr_up_host(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                     float *dst, size_t dstride) {
    HOST_PROLOGUE("micv_pyr_up_host");
    MICV_REQUIRE(src && dst && rows > 0 && cols > 0, "micv_pyr_up_host: bad argument");
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(dstride, 2 * cols, 4),
                 "micv_pyr_up_host: bad stride");
    DevBuf ds((size_t)rows * cols * 4), dd((size_t)rows * cols * 16);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, src, sstride, (size_t)cols * 4, rows, s));
    MICV_TIMED("pyrUpsampleKernel", micv_pyr_up_dev(ctx, ds.as<float>(), rows, cols, (size_t)cols * 4, dd.as<float>(),
                             (size_t)cols * 8, s));
    MICV_TRY(down2d(dst, dstride, dd.p, (size_t)cols * 8, 2 * rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_gaussian_pyramid_host(micv_ctx *ctx, const float *src, int rows, int cols,
                               size_t sstride, int levels, float *const *dst_levels) {
    HOST_PROLOGUE("micv_gaussian_pyramid_host");
    MICV_REQUIRE(src && dst_levels && rows > 0 && cols > 0, "micv_gaussian_pyramid_host: bad argument");
    MICV_REQUIRE(levels >= 1 && levels <= 16 && (rows >> (levels - 1)) > 0 &&
                     (cols >> (levels - 1)) > 0,
                 "micv_gaussian_pyramid_host: %d levels do not fit a %dx%d image", levels, rows,
                 cols);
    MICV_REQUIRE(stride_ok(sstride, cols, 4), "micv_gaussian_pyramid_host: bad stride");
    size_t total = 0, off[16];
    for (int l = 0; l < levels; l++) {
        off[l] = total;
        total += (((size_t)(rows >> l) * (cols >> l)) + 63) & ~size_t(63);
    }
    DevBuf ds((size_t)rows * cols * 4), dd(total * 4);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, src, sstride, (size_t)cols * 4, rows, s));
    float *lv[16];
    for (int l = 0; l < levels; l++) lv[l] = dd.as<float>() + off[l];
    MICV_TRY(micv_gaussian_pyramid_dev(ctx, ds.as<float>(), rows, cols, (size_t)cols * 4, levels, lv, s));
    for (int l = 0; l < levels; l++) {
        MICV_REQUIRE(dst_levels[l] != nullptr, "micv_gaussian_pyramid_host: dst_levels[%d] is null", l);
        MICV_HIP(hipMemcpyAsync(dst_levels[l], lv[l], (size_t)(rows >> l) * (cols >> l) * 4,
                                hipMemcpyDeviceToHost, s));
    }
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_sobel_host(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                    int ksize, float scale, float *gx, float *gy, size_t gstride) {
    HOST_PROLOGUE("micv_sobel_host");
    MICV_REQUIRE(src && gx && gy && rows > 0 && cols > 0, "micv_sobel_host: bad argument");
    MICV_REQUIRE(stride_ok(sstride, cols, 4) && stride_ok(gstride, cols, 4),
                 "micv_sobel_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf ds(n), dx(n), dy(n);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dx); MICV_ALLOC_OK(dy);
    MICV_TRY(up2d(ds.p, src, sstride, rb, rows, s));
    MICV_TRY(micv_sobel_dev(ctx, ds.as<float>(), rows, cols, rb, ksize, scale, dx.as<float>(),
                            dy.as<float>(), rb, s));
    MICV_TRY(down2d(gx, gstride, dx.p, rb, rows, s));
    MICV_TRY(down2d(gy, gstride, dy.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

}  // extern "C"

// ---- ps4 / ps2 / ps1 host flavours -----------------------------------------------------------
extern "C" {

int micv_harris_response_ex_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                                 size_t gstride, int win, double sigma, float alpha, int flags, float *resp,
                                 size_t rstride) {
    HOST_PROLOGUE("micv_harris_response_host");
    MICV_REQUIRE(gx && gy && resp && rows > 0 && cols > 0, "micv_harris_response_host: bad argument");
    MICV_REQUIRE(stride_ok(gstride, cols, 4) && stride_ok(rstride, cols, 4),
                 "micv_harris_response_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf dx(n), dy(n), dr(n);
    MICV_ALLOC_OK(dx); MICV_ALLOC_OK(dy); MICV_ALLOC_OK(dr);
    MICV_TRY(up2d(dx.p, gx, gstride, rb, rows, s));
    MICV_TRY(up2d(dy.p, gy, gstride, rb, rows, s));
    MICV_TIMED("cornerResponseKernel", micv_harris_response_ex_dev(ctx, dx.as<float>(), dy.as<float>(), rows, cols, rb, win,
                                      sigma, alpha, flags, dr.as<float>(), rb, s));
    MICV_TRY(down2d(resp, rstride, dr.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_harris_response_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                              size_t gstride, int win, double sigma, float alpha, float *resp,
                              size_t rstride) {
    return micv_harris_response_ex_host(ctx, gx, gy, rows, cols, gstride, win, sigma, alpha, 0, resp, rstride);
}

int micv_harris_refine_host(micv_ctx *ctx, const float *resp, int rows, int cols, size_t rstride,
                            double threshold, int min_distance, float *corners, size_t cstride,
                            int32_t *locs_yx, int64_t cap, int64_t *count) {
    HOST_PROLOGUE("micv_harris_refine_host");
    MICV_REQUIRE(resp && corners && count && rows > 0 && cols > 0 && cap >= 0,
                 "micv_harris_refine_host: bad argument");
    MICV_REQUIRE(stride_ok(rstride, cols, 4) && stride_ok(cstride, cols, 4),
                 "micv_harris_refine_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf dr(n), dc(n), dl((size_t)cap * 8), dn(8);
    MICV_ALLOC_OK(dr); MICV_ALLOC_OK(dc); MICV_ALLOC_OK(dl); MICV_ALLOC_OK(dn);
    MICV_TRY(up2d(dr.p, resp, rstride, rb, rows, s));
    MICV_TIMED("refineCornersKernel", micv_harris_refine_dev(ctx, dr.as<float>(), rows, cols, rb, threshold, min_distance,
                                    dc.as<float>(), rb, dl.as<int32_t>(), cap, dn.as<int64_t>(), s));
    MICV_TRY(down2d(corners, cstride, dc.p, rb, rows, s));
    MICV_HIP(hipMemcpyAsync(count, dn.p, 8, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipStreamSynchronize(s));
    const int64_t take = *count < cap ? *count : cap;
    if (take > 0) MICV_HIP(hipMemcpy(locs_yx, dl.p, (size_t)take * 8, hipMemcpyDeviceToHost));
    return MICV_OK;
}

int micv_harris_corners_host(micv_ctx *ctx, const float *img, int rows, int cols, size_t stride, int sobel_ksize, int win,
                             double sigma, float alpha, int flags, double threshold, int min_distance, float *gx, float *gy,
                             size_t gstride, float *resp, size_t rstride, float *corners, size_t cstride, int32_t *locs_yx,
                             int64_t cap, int64_t *count) {
    HOST_PROLOGUE("micv_harris_corners_host");
    MICV_REQUIRE(img && count && rows > 0 && cols > 0 && cap >= 0 && (locs_yx || cap == 0), "micv_harris_corners_host: bad argument");
    MICV_REQUIRE((gx == nullptr) == (gy == nullptr), "micv_harris_corners_host: give both gradient outputs or neither");
    MICV_REQUIRE(stride_ok(stride, cols, 4) && (!gx || stride_ok(gstride, cols, 4)) && (!resp || stride_ok(rstride, cols, 4)) &&
                     (!corners || stride_ok(cstride, cols, 4)),
                 "micv_harris_corners_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    // one upload, every requested field downloaded: what three _host calls move three times
    DevBuf di(n), dgx(gx ? n : 0), dgy(gy ? n : 0), dr(resp ? n : 0), dc(corners ? n : 0), dl((size_t)cap * 8), dn(8);
    MICV_ALLOC_OK(di); MICV_ALLOC_OK(dl); MICV_ALLOC_OK(dn);
    if (gx) { MICV_ALLOC_OK(dgx); MICV_ALLOC_OK(dgy); }
    if (resp) MICV_ALLOC_OK(dr);
    if (corners) MICV_ALLOC_OK(dc);
    MICV_TRY(up2d(di.p, img, stride, rb, rows, s));
    MICV_TIMED("harrisCornersChain",
               micv_harris_corners_dev(ctx, di.as<float>(), rows, cols, rb, sobel_ksize, win, sigma, alpha, flags, threshold, min_distance,
                                       gx ? dgx.as<float>() : nullptr, gy ? dgy.as<float>() : nullptr, rb, resp ? dr.as<float>() : nullptr, rb,
                                       corners ? dc.as<float>() : nullptr, rb, dl.as<int32_t>(), cap, dn.as<int64_t>(), s));
    if (gx) {
        MICV_TRY(down2d(gx, gstride, dgx.p, rb, rows, s));
        MICV_TRY(down2d(gy, gstride, dgy.p, rb, rows, s));
    }
    if (resp) MICV_TRY(down2d(resp, rstride, dr.p, rb, rows, s));
    if (corners) MICV_TRY(down2d(corners, cstride, dc.p, rb, rows, s));
    MICV_HIP(hipMemcpyAsync(count, dn.p, 8, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipStreamSynchronize(s));
    const int64_t take = *count < cap ? *count : cap;
    if (take > 0) MICV_HIP(hipMemcpy(locs_yx, dl.p, (size_t)take * 8, hipMemcpyDeviceToHost));
    return MICV_OK;
}

int micv_sift_angles_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                          size_t gstride, float *angles, size_t astride) {
    HOST_PROLOGUE("micv_sift_angles_host");
    MICV_REQUIRE(gx && gy && angles && rows > 0 && cols > 0, "micv_sift_angles_host: bad argument");
    MICV_REQUIRE(stride_ok(gstride, cols, 4) && stride_ok(astride, cols, 4),
                 "micv_sift_angles_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf dx(n), dy(n), da(n);
    MICV_ALLOC_OK(dx); MICV_ALLOC_OK(dy); MICV_ALLOC_OK(da);
    MICV_TRY(up2d(dx.p, gx, gstride, rb, rows, s));
    MICV_TRY(up2d(dy.p, gy, gstride, rb, rows, s));
    MICV_TRY(micv_sift_angles_dev(ctx, dx.as<float>(), dy.as<float>(), rows, cols, rb,
                                  da.as<float>(), rb, s));
    MICV_TRY(down2d(angles, astride, da.p, rb, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_sift_keypoints_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                             size_t gstride, const int32_t *locs_yx, int64_t n, float size,
                             float *kp_xysa) {
    HOST_PROLOGUE("micv_sift_keypoints_host");
    MICV_REQUIRE(gx && gy && rows > 0 && cols > 0 && n >= 0 && (n == 0 || (locs_yx && kp_xysa)),
                 "micv_sift_keypoints_host: bad argument");
    MICV_REQUIRE(stride_ok(gstride, cols, 4), "micv_sift_keypoints_host: bad stride");
    for (int64_t i = 0; i < n; i++)
        MICV_REQUIRE((unsigned)locs_yx[2 * i] < (unsigned)rows && (unsigned)locs_yx[2 * i + 1] < (unsigned)cols,
                     "micv_sift_keypoints_host: corner %lld outside the image", (long long)i);
    if (n == 0) return MICV_OK;
    const size_t rb = (size_t)cols * 4, bytes = rb * rows;
    DevBuf dx(bytes), dy(bytes), dl((size_t)n * 8), dk((size_t)n * 16);
    MICV_ALLOC_OK(dx); MICV_ALLOC_OK(dy); MICV_ALLOC_OK(dl); MICV_ALLOC_OK(dk);
    MICV_TRY(up2d(dx.p, gx, gstride, rb, rows, s));
    MICV_TRY(up2d(dy.p, gy, gstride, rb, rows, s));
    MICV_HIP(hipMemcpyAsync(dl.p, locs_yx, (size_t)n * 8, hipMemcpyHostToDevice, s));
    MICV_TRY(micv_sift_keypoints_dev(ctx, dx.as<float>(), dy.as<float>(), rows, cols, rb,
                                     dl.as<int32_t>(), n, size, dk.as<float>(), s));
    MICV_HIP(hipMemcpyAsync(kp_xysa, dk.p, (size_t)n * 16, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_sift_descriptors_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                               size_t gstride, const float *kp_xysa, int64_t n, float *desc,
                               size_t dstride) {
    HOST_PROLOGUE("micv_sift_descriptors_host");
    MICV_REQUIRE(gx && gy && rows > 0 && cols > 0 && n >= 0 && (n == 0 || (kp_xysa && desc)),
                 "micv_sift_descriptors_host: bad argument");
    MICV_REQUIRE(stride_ok(gstride, cols, 4) && dstride % 4 == 0 && dstride >= 512,
                 "micv_sift_descriptors_host: bad stride");
    if (n == 0) return MICV_OK;
    const size_t rb = (size_t)cols * 4, bytes = rb * rows;
    DevBuf dx(bytes), dy(bytes), dk((size_t)n * 16), dd((size_t)n * 512);
    MICV_ALLOC_OK(dx); MICV_ALLOC_OK(dy); MICV_ALLOC_OK(dk); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(dx.p, gx, gstride, rb, rows, s));
    MICV_TRY(up2d(dy.p, gy, gstride, rb, rows, s));
    MICV_HIP(hipMemcpyAsync(dk.p, kp_xysa, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MICV_TRY(micv_sift_descriptors_dev(ctx, dx.as<float>(), dy.as<float>(), rows, cols, rb, dk.as<float>(), n,
                                       dd.as<float>(), 512, s));
    MICV_TRY(down2d(desc, dstride, dd.p, 512, (int)n, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

static int stereo_host(bool ncc, micv_ctx *ctx, const float *left, const float *right, int rows,
                       int cols, size_t stride, int rad, int min_d, int max_d, int flags,
                       int8_t *disp, size_t dstride) {
    HOST_PROLOGUE("micv_disparity_host");
    MICV_REQUIRE(left && right && disp && rows > 0 && cols > 0, "micv_disparity_host: bad argument");
    MICV_REQUIRE(stride_ok(stride, cols, 4) && dstride >= (size_t)cols, "micv_disparity_host: bad stride");
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    DevBuf dl(n), dr(n), dd((size_t)rows * cols);
    MICV_ALLOC_OK(dl); MICV_ALLOC_OK(dr); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(dl.p, left, stride, rb, rows, s));
    MICV_TRY(up2d(dr.p, right, stride, rb, rows, s));
    if (ncc)
        MICV_TIMED("disparityNCorrKernel", micv_disparity_ncorr_dev(ctx, dl.as<float>(), dr.as<float>(), rows, cols, rb, rad,
                                          min_d, max_d, flags, dd.as<int8_t>(), cols, s));
    else
        MICV_TIMED("disparitySSDKernel", micv_disparity_ssd_dev(ctx, dl.as<float>(), dr.as<float>(), rows, cols, rb, rad,
                                        min_d, max_d, flags, dd.as<int8_t>(), cols, s));
    MICV_TRY(down2d(disp, dstride, dd.p, (size_t)cols, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_disparity_ssd_host(micv_ctx *ctx, const float *left, const float *right, int rows,
                            int cols, size_t stride, int window_rad, int min_disparity,
                            int max_disparity, int flags, int8_t *disp, size_t dstride) {
    return stereo_host(false, ctx, left, right, rows, cols, stride, window_rad, min_disparity,
                       max_disparity, flags, disp, dstride);
}
int micv_disparity_ncorr_host(micv_ctx *ctx, const float *left, const float *right, int rows,
                              int cols, size_t stride, int window_rad, int min_disparity,
                              int max_disparity, int flags, int8_t *disp, size_t dstride) {
    return stereo_host(true, ctx, left, right, rows, cols, stride, window_rad, min_disparity,
                       max_disparity, flags, disp, dstride);
}

int micv_hough_lines_host(micv_ctx *ctx, const uint8_t *mask, int rows, int cols, size_t mstride,
                          unsigned rho_bin, unsigned theta_bin, int32_t *acc) {
    HOST_PROLOGUE("micv_hough_lines_host");
    MICV_REQUIRE(mask && acc && mstride >= (size_t)cols, "micv_hough_lines_host: bad argument");
    int rb, tb;
    MICV_TRY(micv_hough_lines_dims(rows, cols, rho_bin, theta_bin, &rb, &tb));
    DevBuf dm((size_t)rows * cols), da((size_t)rb * tb * 4);
    MICV_ALLOC_OK(dm); MICV_ALLOC_OK(da);
    MICV_TRY(up2d(dm.p, mask, mstride, (size_t)cols, rows, s));
    MICV_TIMED("houghLinesAccumulateKernel", micv_hough_lines_dev(ctx, dm.as<uint8_t>(), rows, cols, cols, rho_bin, theta_bin,
                                  da.as<int32_t>(), s));
    MICV_HIP(hipMemcpyAsync(acc, da.p, (size_t)rb * tb * 4, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_hough_circles_host(micv_ctx *ctx, const uint8_t *mask, int rows, int cols,
                            size_t mstride, unsigned radius, int32_t *acc) {
    HOST_PROLOGUE("micv_hough_circles_host");
    MICV_REQUIRE(mask && acc && rows > 0 && cols > 0 && mstride >= (size_t)cols,
                 "micv_hough_circles_host: bad argument");
    DevBuf dm((size_t)rows * cols), da((size_t)rows * cols * 4);
    MICV_ALLOC_OK(dm); MICV_ALLOC_OK(da);
    MICV_TRY(up2d(dm.p, mask, mstride, (size_t)cols, rows, s));
    MICV_TIMED("houghCirclesAccumulateKernel", micv_hough_circles_dev(ctx, dm.as<uint8_t>(), rows, cols, cols, radius,
                                    da.as<int32_t>(), s));
    MICV_HIP(hipMemcpyAsync(acc, da.p, (size_t)rows * cols * 4, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_hough_peaks_host(micv_ctx *ctx, const int32_t *acc, int rows, int cols,
                          unsigned num_peaks, int threshold, uint32_t *peaks_rc, int64_t *count) {
    HOST_PROLOGUE("micv_hough_peaks_host");
    MICV_REQUIRE(acc && count && rows > 0 && cols > 0 && (peaks_rc || num_peaks == 0),
                 "micv_hough_peaks_host: bad argument");
    DevBuf da((size_t)rows * cols * 4), dp((size_t)num_peaks * 8 + 8), dn(8);
    MICV_ALLOC_OK(da); MICV_ALLOC_OK(dp); MICV_ALLOC_OK(dn);
    MICV_HIP(hipMemcpyAsync(da.p, acc, (size_t)rows * cols * 4, hipMemcpyHostToDevice, s));
    MICV_TIMED("findLocalMaximaKernel", micv_hough_peaks_dev(ctx, da.as<int32_t>(), rows, cols, num_peaks, threshold,
                                  dp.as<uint32_t>(), dn.as<int64_t>(), s));
    MICV_HIP(hipMemcpyAsync(count, dn.p, 8, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipStreamSynchronize(s));
    if (*count > 0) MICV_HIP(hipMemcpy(peaks_rc, dp.p, (size_t)*count * 8, hipMemcpyDeviceToHost));
    return MICV_OK;
}

// ---- host-pointer flavours of the "next" rows --------------------------

int micv_generate_edge_host(micv_ctx *ctx, const uint8_t *src, int rows, int cols, size_t stride,
                            int gauss_size, double gauss_sigma, double low_thresh,
                            double high_thresh, uint8_t *edges, size_t estride) {
    HOST_PROLOGUE("micv_generate_edge_host");
    MICV_REQUIRE(src && edges && rows > 0 && cols > 0 && stride >= (size_t)cols && estride >= (size_t)cols,
                 "micv_generate_edge_host: bad argument");
    DevBuf ds((size_t)rows * cols), de((size_t)rows * cols);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(de);
    MICV_TRY(up2d(ds.p, src, stride, (size_t)cols, rows, s));
    MICV_TRY(micv_generate_edge_dev(ctx, ds.as<uint8_t>(), rows, cols, cols, gauss_size, gauss_sigma,
                                    low_thresh, high_thresh, de.as<uint8_t>(), cols, s));
    MICV_TRY(down2d(edges, estride, de.p, (size_t)cols, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_bf_knn2_host(micv_ctx *ctx, const float *query, int nq, size_t qstride, const float *train,
                      int nt, size_t tstride, int dim, int32_t *idx2, float *dist2) {
    HOST_PROLOGUE("micv_bf_knn2_host");
    MICV_REQUIRE(query && train && idx2 && dist2 && nq > 0 && nt >= 2 && dim > 0 &&
                     stride_ok(qstride, dim, 4) && stride_ok(tstride, dim, 4),
                 "micv_bf_knn2_host: bad argument");
    const size_t rb = (size_t)dim * 4;
    DevBuf dq(rb * nq), dt(rb * nt), di((size_t)nq * 8), dd((size_t)nq * 8);
    MICV_ALLOC_OK(dq); MICV_ALLOC_OK(dt); MICV_ALLOC_OK(di); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(dq.p, query, qstride, rb, nq, s));
    MICV_TRY(up2d(dt.p, train, tstride, rb, nt, s));
    MICV_TRY(micv_bf_knn2_dev(ctx, dq.as<float>(), nq, rb, dt.as<float>(), nt, rb, dim, di.as<int32_t>(),
                              dd.as<float>(), s));
    MICV_HIP(hipMemcpyAsync(idx2, di.p, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipMemcpyAsync(dist2, dd.p, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_bf_ratio_filter_host(micv_ctx *ctx, const int32_t *idx2, const float *dist2, int nq,
                              double ratio, int32_t *matches_qt, float *distances, int64_t cap,
                              int64_t *count) {
    HOST_PROLOGUE("micv_bf_ratio_filter_host");
    MICV_REQUIRE(idx2 && dist2 && count && nq > 0 && cap >= 0 && (cap == 0 || (matches_qt && distances)),
                 "micv_bf_ratio_filter_host: bad argument");
    DevBuf di((size_t)nq * 8), dd((size_t)nq * 8), dm((size_t)cap * 8 + 8), dl((size_t)cap * 4 + 8), dn(8);
    MICV_ALLOC_OK(di); MICV_ALLOC_OK(dd); MICV_ALLOC_OK(dm); MICV_ALLOC_OK(dl); MICV_ALLOC_OK(dn);
    MICV_HIP(hipMemcpyAsync(di.p, idx2, (size_t)nq * 8, hipMemcpyHostToDevice, s));
    MICV_HIP(hipMemcpyAsync(dd.p, dist2, (size_t)nq * 8, hipMemcpyHostToDevice, s));
    MICV_TRY(micv_bf_ratio_filter_dev(ctx, di.as<int32_t>(), dd.as<float>(), nq, ratio, dm.as<int32_t>(),
                                      dl.as<float>(), cap, dn.as<int64_t>(), s));
    MICV_HIP(hipMemcpyAsync(count, dn.p, 8, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipStreamSynchronize(s));
    const int64_t n = *count < cap ? *count : cap;
    if (n > 0) {
        MICV_HIP(hipMemcpy(matches_qt, dm.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        MICV_HIP(hipMemcpy(distances, dl.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    return MICV_OK;
}

int micv_mhi_frame_difference_host(micv_ctx *ctx, const uint8_t *f1, const uint8_t *f2, int rows,
                                   int cols, size_t stride, double thresh, int blur_w, int blur_h,
                                   double blur_sigma, uint8_t *diff, size_t dstride) {
    HOST_PROLOGUE("micv_mhi_frame_difference_host");
    MICV_REQUIRE(f1 && f2 && diff && rows > 0 && cols > 0 && stride >= (size_t)cols && dstride >= (size_t)cols,
                 "micv_mhi_frame_difference_host: bad argument");
    const size_t n = (size_t)rows * cols;
    DevBuf d1(n), d2(n), dd(n);
    MICV_ALLOC_OK(d1); MICV_ALLOC_OK(d2); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(d1.p, f1, stride, (size_t)cols, rows, s));
    MICV_TRY(up2d(d2.p, f2, stride, (size_t)cols, rows, s));
    MICV_TRY(micv_mhi_frame_difference_dev(ctx, d1.as<uint8_t>(), d2.as<uint8_t>(), rows, cols, cols, thresh,
                                           blur_w, blur_h, blur_sigma, dd.as<uint8_t>(), cols, s));
    MICV_TRY(down2d(diff, dstride, dd.p, (size_t)cols, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_mhi_energy_host(micv_ctx *ctx, const uint8_t *mhi, int rows, int cols, size_t sstride,
                         uint8_t *mei, size_t dstride) {
    HOST_PROLOGUE("micv_mhi_energy_host");
    MICV_REQUIRE(mhi && mei && rows > 0 && cols > 0 && sstride >= (size_t)cols && dstride >= (size_t)cols,
                 "micv_mhi_energy_host: bad argument");
    const size_t n = (size_t)rows * cols;
    DevBuf ds(n), dd(n);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, mhi, sstride, (size_t)cols, rows, s));
    MICV_TRY(micv_mhi_energy_dev(ctx, ds.as<uint8_t>(), rows, cols, cols, dd.as<uint8_t>(), cols, s));
    MICV_TRY(down2d(mei, dstride, dd.p, (size_t)cols, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_mhi_threshold_host(micv_ctx *ctx, const uint8_t *src, int rows, int cols, size_t sstride,
                            double thresh, uint8_t *dst, size_t dstride) {
    HOST_PROLOGUE("micv_mhi_threshold_host");
    MICV_REQUIRE(src && dst && rows > 0 && cols > 0 && sstride >= (size_t)cols && dstride >= (size_t)cols,
                 "micv_mhi_threshold_host: bad argument");
    const size_t n = (size_t)rows * cols;
    DevBuf ds(n), dd(n);
    MICV_ALLOC_OK(ds); MICV_ALLOC_OK(dd);
    MICV_TRY(up2d(ds.p, src, sstride, (size_t)cols, rows, s));
    MICV_TRY(micv_mhi_threshold_dev(ctx, ds.as<uint8_t>(), rows, cols, cols, thresh, dd.as<uint8_t>(), cols, s));
    MICV_TRY(down2d(dst, dstride, dd.p, (size_t)cols, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

int micv_mhi_update_host(micv_ctx *ctx, uint8_t *history, size_t hstride, const uint8_t *mask,
                         size_t mstride, int rows, int cols, int tau) {
    HOST_PROLOGUE("micv_mhi_update_host");
    MICV_REQUIRE(history && mask && rows > 0 && cols > 0 && hstride >= (size_t)cols && mstride >= (size_t)cols,
                 "micv_mhi_update_host: bad argument");
    const size_t n = (size_t)rows * cols;
    DevBuf dh(n), dm(n);
    MICV_ALLOC_OK(dh); MICV_ALLOC_OK(dm);
    MICV_TRY(up2d(dh.p, history, hstride, (size_t)cols, rows, s));
    MICV_TRY(up2d(dm.p, mask, mstride, (size_t)cols, rows, s));
    MICV_TRY(micv_mhi_update_dev(ctx, dh.as<uint8_t>(), cols, dm.as<uint8_t>(), cols, rows, cols, tau, s));
    MICV_TRY(down2d(history, hstride, dh.p, (size_t)cols, rows, s));
    MICV_HIP(hipStreamSynchronize(s));
    return MICV_OK;
}

}  // extern "C"
