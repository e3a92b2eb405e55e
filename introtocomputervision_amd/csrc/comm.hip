// comm.hip -- the multi-GPU part of the C ABI: a communicator on
// RCCL, one frame pair split by rows over the ranks with the coarse-flow halo exchanged point to point per
// pyramid level, and the int32 all-reduce of the sharded Hough accumulator -- so that the reference's C++
// caller (ps5_cpp/src/Solution.cpp:60-64 -> lk::calcOpticalFlowPyr) can shard without Python.
//
// RCCL is resolved at run time (dlopen "librccl.so.1"): libmicv.so keeps linking only libamdhip64, a process
// that never creates a communicator never loads RCCL, and a process that already holds an RCCL (PyTorch ships
// one under the same soname) shares that copy instead of initialising a second one.
//
// The plan (RowPlan) is the one of introtocomputervision_amd/shard.py (RowShardPlan): the coarsest level is cut
// evenly, finer levels double the cuts, the last rank absorbs odd remainders; a band needs
// halo_rows(win) = (win/2 + 2)/2 + 3 coarse-flow rows beyond itself.  Every rank holds the whole frames (the
// images are static inputs; the flow is the only dynamic exchange).  Results equal the unsharded call bit for
// bit: the band launches are micv_lk_level_batch_dev, i.e. the same kernels on a row range.
#include <dlfcn.h>
// RCCL is dlopen'ed, so its header is only a source of types: a ROCm install without the RCCL development files still
// builds libmicv.so from the handful of declarations below -- the NCCL 2.x ABI these entry points use, fixed
// since NCCL 2.0 (-DMICV_NO_RCCL_HEADER forces this branch; tests/test_capi_and_host.py compiles it).  With the header
// present the static_asserts hold the two in step.
#if defined(__has_include) && !defined(MICV_NO_RCCL_HEADER)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define MICV_HAVE_RCCL_H 1
#endif
#endif
#ifdef MICV_HAVE_RCCL_H
static_assert(sizeof(ncclUniqueId) == 128 && (int)ncclSuccess == 0 && (int)ncclInt32 == 2 && (int)ncclFloat32 == 7 && (int)ncclSum == 0,
              "the local declarations below state the same ABI");
#else
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;  // other values are only ever handed to ncclGetErrorString
typedef enum { ncclInt32 = 2, ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

#include <cstring>
#include <mutex>
#include <vector>

#include "kernels.hpp"

namespace micv {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static const Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (!r.h) return;
#define MICV_SYM(field, sym)                                            \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, sym)); \
    if (!r.field) {                                                     \
        r.h = nullptr;                                                  \
        return;                                                         \
    }
        MICV_SYM(GetUniqueId, "ncclGetUniqueId")
        MICV_SYM(CommInitRank, "ncclCommInitRank")
        MICV_SYM(CommDestroy, "ncclCommDestroy")
        MICV_SYM(CommCount, "ncclCommCount")
        MICV_SYM(CommUserRank, "ncclCommUserRank")
        MICV_SYM(GroupStart, "ncclGroupStart")
        MICV_SYM(GroupEnd, "ncclGroupEnd")
        MICV_SYM(Send, "ncclSend")
        MICV_SYM(Recv, "ncclRecv")
        MICV_SYM(AllReduce, "ncclAllReduce")
        MICV_SYM(Broadcast, "ncclBroadcast")
        MICV_SYM(GetErrorString, "ncclGetErrorString")
#undef MICV_SYM
    });
    return r.h ? &r : nullptr;
}

#define MICV_NCCL(call)                                                                              \
    do {                                                                                             \
        const ncclResult_t nr_ = (call);                                                             \
        if (nr_ != ncclSuccess) {                                                                    \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, rccl()->GetErrorString(nr_)); \
            return MICV_EHIP;                                                                        \
        }                                                                                            \
    } while (0)

// shard.py RowShardPlan, restated.
struct RowPlan {
    int levels = 0, world = 0, halo = 0;
    int rows[16], cols[16];
    std::vector<int> cuts[16];  // cuts[l][g] .. cuts[l][g + 1] = band of rank g at level l
    struct Xfer {
        int src, dst, r0, r1;
    };
    bool build(int r, int c, int L, int W, int win) {
        levels = L;
        world = W;
        halo = (win / 2 + 2) / 2 + 3;
        for (int l = 0; l < L; l++) {
            rows[l] = r >> l;
            cols[l] = c >> l;
        }
        const int top = rows[L - 1];
        if (top < W) return false;
        cuts[L - 1].resize(W + 1);
        for (int g = 0; g < W; g++) cuts[L - 1][g] = (int)((long long)g * top / W);
        cuts[L - 1][W] = top;
        for (int l = L - 2; l >= 0; l--) {
            cuts[l].resize(W + 1);
            for (int g = 0; g < W; g++) cuts[l][g] = 2 * cuts[l + 1][g];
            cuts[l][W] = rows[l];
        }
        return true;
    }
    void band(int l, int g, int *a, int *b) const {
        *a = cuts[l][g];
        *b = cuts[l][g + 1];
    }
    void needed(int l, int g, int *n0, int *n1) const {
        int a, b;
        band(l, g, &a, &b);
        *n0 = a - halo > 0 ? a - halo : 0;
        *n1 = b + halo < rows[l] ? b + halo : rows[l];
    }
    std::vector<Xfer> transfers(int l) const {
        std::vector<Xfer> out;
        for (int dst = 0; dst < world; dst++) {
            int n0, n1;
            needed(l, dst, &n0, &n1);
            for (int src = 0; src < world; src++) {
                if (src == dst) continue;
                int a, b;
                band(l, src, &a, &b);
                const int r0 = a > n0 ? a : n0, r1 = b < n1 ? b : n1;
                if (r0 < r1) out.push_back({src, dst, r0, r1});
            }
        }
        return out;
    }
};

}  // namespace micv

using namespace micv;

// Opaque in mi_cv.h.
struct micv_comm {
    ncclComm_t comm = nullptr;
    bool owned = false;
    int rank = 0, world = 1, device = 0;
    // device memory of the row-shard driver: pyramids of both image sets, one flow block per level, exchange
    // slabs.  One allocation, grown when a call needs more (never while a launch of this communicator may still
    // read it: growth synchronises the device first).
    void *mem = nullptr;
    size_t mem_bytes = 0;
    int reserve(size_t bytes, void **out) {
        if (bytes > mem_bytes) {
            MICV_HIP(hipDeviceSynchronize());
            if (mem) (void)hipFree(mem);
            mem = nullptr;
            mem_bytes = 0;
            MICV_HIP(hipMalloc(&mem, bytes));
            mem_bytes = bytes;
        }
        *out = mem;
        return MICV_OK;
    }
};

extern "C" {

int micv_comm_unique_id(void *id128) {
    MICV_REQUIRE(id128 != nullptr, "micv_comm_unique_id: null argument");
    const Rccl *r = rccl();
    if (!r) {
        set_error("micv_comm_unique_id: librccl.so.1 not found (RCCL is loaded at run time)");
        return MICV_EUNSUPPORTED;
    }
    static_assert(sizeof(ncclUniqueId) == MICV_COMM_ID_BYTES, "mi_cv.h states the size of an RCCL unique id");
    ncclUniqueId id;
    MICV_NCCL(r->GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return MICV_OK;
}

int micv_comm_create(micv_ctx *ctx, void *nccl_comm, const void *unique_id128, int rank, int world, micv_comm **out) {
    MICV_REQUIRE(ctx && out, "micv_comm_create: null argument");
    MICV_REQUIRE((nccl_comm != nullptr) != (unique_id128 != nullptr),
                 "micv_comm_create: give an existing ncclComm_t OR a unique id (exactly one)");
    MICV_REQUIRE(world >= 1 && rank >= 0 && rank < world, "micv_comm_create: rank %d of %d", rank, world);
    const Rccl *r = rccl();
    if (!r) {
        set_error("micv_comm_create: librccl.so.1 not found (RCCL is loaded at run time)");
        return MICV_EUNSUPPORTED;
    }
    MICV_HIP(hipSetDevice(ctx->device));
    micv_comm *c = new micv_comm;
    c->rank = rank;
    c->world = world;
    c->device = ctx->device;
    if (nccl_comm) {
        c->comm = static_cast<ncclComm_t>(nccl_comm);  // borrowed: the caller destroys it
        int n = 0, me = -1;
        if (r->CommCount(c->comm, &n) != ncclSuccess || r->CommUserRank(c->comm, &me) != ncclSuccess || n != world || me != rank) {
            delete c;
            set_error("micv_comm_create: the communicator has rank %d of %d, not %d of %d", me, n, rank, world);
            return MICV_EINVAL;
        }
    } else {
        ncclUniqueId id;
        memcpy(&id, unique_id128, sizeof(id));
        const ncclResult_t nr = r->CommInitRank(&c->comm, world, id, rank);
        if (nr != ncclSuccess) {
            delete c;
            set_error("micv_comm_create: ncclCommInitRank -> %s", r->GetErrorString(nr));
            return MICV_EHIP;
        }
        c->owned = true;
    }
    *out = c;
    return MICV_OK;
}

int micv_comm_destroy(micv_comm *c) {
    if (!c) return MICV_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->mem) (void)hipFree(c->mem);
    if (c->owned && c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    delete c;
    return MICV_OK;
}

int micv_comm_rank(const micv_comm *c, int *rank, int *world) {
    MICV_REQUIRE(c && rank && world, "micv_comm_rank: null argument");
    *rank = c->rank;
    *world = c->world;
    return MICV_OK;
}

int micv_rowshard_band(int rows, int cols, int levels, int world, int win, int rank, int level, int *row_begin,
                       int *row_end, int *need_begin, int *need_end) {
    MICV_REQUIRE(row_begin && row_end, "micv_rowshard_band: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && levels >= 1 && levels <= 16 && (rows >> (levels - 1)) > 0 && (cols >> (levels - 1)) > 0,
                 "micv_rowshard_band: %d levels do not fit a %dx%d image", levels, rows, cols);
    MICV_REQUIRE(world >= 1 && rank >= 0 && rank < world && level >= 0 && level < levels && win >= 1 && (win & 1),
                 "micv_rowshard_band: bad rank / level / window");
    RowPlan p;
    if (!p.build(rows, cols, levels, world, win)) {
        set_error("micv_rowshard_band: %d ranks cannot split the %d-row coarsest level", world, rows >> (levels - 1));
        return MICV_EINVAL;
    }
    p.band(level, rank, row_begin, row_end);
    if (need_begin && need_end) p.needed(level, rank, need_begin, need_end);
    return MICV_OK;
}

int micv_allreduce_sum_i32_dev(micv_ctx *ctx, micv_comm *comm, int32_t *buf, size_t count, micv_stream stream) {
    MICV_REQUIRE(ctx && comm && buf, "micv_allreduce_sum_i32: null argument");
    MICV_REQUIRE(comm->device == ctx->device, "micv_allreduce_sum_i32: communicator and context are on different devices");
    MICV_HIP(hipSetDevice(ctx->device));
    if (count == 0) return MICV_OK;
    MICV_NCCL(rccl()->AllReduce(buf, buf, count, ncclInt32, ncclSum, comm->comm, static_cast<hipStream_t>(stream)));
    return MICV_OK;
}

}  // extern "C"

namespace micv {
// micv_comm_selftest: fill / verify kernels (the check runs on the device; one counter comes back)
__global__ void comm_stamp_kernel(int32_t *buf, int n, int rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = rank * 1000003 + i;
}
__global__ void comm_reduce_fill_kernel(int32_t *buf, int n, int rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = (rank + 1) * (i % 1021 + 1);
}
// bad[0]: ring cells that are not `from`'s stamp; bad[1]: all-reduce cells that are not the sum over the ranks
__global__ void comm_verify_kernel(const int32_t *ring, int n_ring, int from, const int32_t *red, int n_red, int world, unsigned *bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_ring && ring[i] != from * 1000003 + i) atomicAdd(&bad[0], 1u);
    if (i < n_red && red[i] != (world * (world + 1) / 2) * (i % 1021 + 1)) atomicAdd(&bad[1], 1u);
}
}  // namespace micv

extern "C" {

int micv_comm_selftest(micv_ctx *ctx, micv_comm *comm, micv_stream stream) {
    MICV_REQUIRE(ctx && comm, "micv_comm_selftest: null argument");
    MICV_REQUIRE(comm->device == ctx->device, "micv_comm_selftest: communicator and context are on different devices");
    const Rccl *r = rccl();
    if (!r) {
        set_error("micv_comm_selftest: librccl.so.1 not found");
        return MICV_EUNSUPPORTED;
    }
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int N = 1 << 16, NR = 1 << 12;  // 256 KB ring slab (a halo slab of a 1080p batch), 16 KB all-reduce
    void *base = nullptr;
    MICV_TRY(comm->reserve((size_t)(2 * N + NR) * 4 + 256, &base));
    int32_t *snd = static_cast<int32_t *>(base), *rcv = snd + N, *red = rcv + N;
    unsigned *bad = reinterpret_cast<unsigned *>(red + NR);
    const int me = comm->rank, W = comm->world, to = (me + 1) % W, from = (me + W - 1) % W;
    MICV_HIP(hipMemsetAsync(rcv, 0xFF, (size_t)N * 4, s));
    MICV_HIP(hipMemsetAsync(bad, 0, 8, s));
    comm_stamp_kernel<<<N / 256, 256, 0, s>>>(snd, N, me);
    comm_reduce_fill_kernel<<<NR / 256, 256, 0, s>>>(red, NR, me);
    MICV_LAUNCH_CHECK();
    if (W > 1) {
        // the exchange pattern of the row-shard driver: one group, a send and a receive per neighbour, on the launch stream
        MICV_NCCL(r->GroupStart());
        const ncclResult_t n1 = r->Send(snd, N, ncclInt32, to, comm->comm, s);
        const ncclResult_t n2 = n1 == ncclSuccess ? r->Recv(rcv, N, ncclInt32, from, comm->comm, s) : n1;
        const ncclResult_t n3 = r->GroupEnd();
        if (n2 != ncclSuccess || n3 != ncclSuccess) {
            set_error("micv_comm_selftest: rank %d: ring ncclSend -> %d / ncclRecv <- %d failed: %s", me, to, from,
                      r->GetErrorString(n2 != ncclSuccess ? n2 : n3));
            return MICV_EHIP;
        }
    } else {
        MICV_HIP(hipMemcpyAsync(rcv, snd, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
    }
    {
        const ncclResult_t nr = r->AllReduce(red, red, NR, ncclInt32, ncclSum, comm->comm, s);
        if (nr != ncclSuccess) {
            set_error("micv_comm_selftest: rank %d: ncclAllReduce failed: %s", me, r->GetErrorString(nr));
            return MICV_EHIP;
        }
    }
    comm_verify_kernel<<<N / 256, 256, 0, s>>>(rcv, N, from, red, NR, W, bad);
    MICV_LAUNCH_CHECK();
    unsigned host_bad[2] = {~0u, ~0u};
    MICV_HIP(hipMemcpyAsync(host_bad, bad, 8, hipMemcpyDeviceToHost, s));
    MICV_HIP(hipStreamSynchronize(s));
    if (host_bad[0] || host_bad[1]) {
        set_error("micv_comm_selftest: rank %d of %d: %u of %d ring cells from rank %d and %u of %d all-reduce cells are wrong -- "
                  "the fabric or the stream ordering of RCCL calls is broken; results of sharded calls cannot be trusted",
                  me, W, host_bad[0], N, from, host_bad[1], NR);
        return MICV_EHIP;
    }
    return MICV_OK;
}

int micv_hough_lines_rowshard_dev(micv_ctx *ctx, micv_comm *comm, const uint8_t *mask_band, int band_rows, int cols,
                                  size_t mstride, int row0, int rows, unsigned rho_bin, unsigned theta_bin,
                                  int32_t *acc, micv_stream stream) {
    MICV_REQUIRE(ctx && comm && acc, "micv_hough_lines_rowshard: null argument");
    MICV_REQUIRE(comm->device == ctx->device, "micv_hough_lines_rowshard: communicator and context are on different devices");
    int rb = 0, tb = 0;
    MICV_TRY(micv_hough_lines_dims(rows, cols, rho_bin, theta_bin, &rb, &tb));
    // this rank's edge points vote into its private full-size accumulator; integer sums over the ranks are the
    // unsharded accumulator bit for bit, in any order
    MICV_TRY(micv_hough_lines_band_dev(ctx, mask_band, band_rows, cols, mstride, row0, rows, rho_bin, theta_bin, acc, stream));
    return micv_allreduce_sum_i32_dev(ctx, comm, acc, (size_t)rb * tb, stream);
}

// ---- the row-shard driver ------------------------------------------------------------------------------------
// One rank's share of a row-sharded lk::calcOpticalFlowPyr, cut into the steps between which the transport runs:
// build() -- pyramids; per level, coarsest first: pack(l + 1) -> [transport] -> unpack(l + 1) -> launch(l).
// The transport is RCCL point-to-point (micv_lk_flow_pyr_rowshard_dev) or, for `world` virtual ranks in one process
// on one device, row copies between the ranks' slabs (micv_lk_flow_pyr_rowshard_virtual_dev: the same plan, the
// same packing, the same band launches -- what the tests run at world sizes a one-GPU box cannot give RCCL).
}  // extern "C"

namespace micv {

struct RowShardRun {
    micv_ctx *ctx;
    hipStream_t s;
    RowPlan plan;
    int me, batch, win, levels;
    const float *prev, *next;
    size_t pair_stride, stride;
    float *u, *v;
    size_t opair_stride, ostride;
    float *mem = nullptr;
    size_t lvl_elems[16], pyr_off[16], flow_off[16], total = 0;
    std::vector<RowPlan::Xfer> xf[16];
    std::vector<size_t> slab_off[16];

    // memory: pyramids (levels >= 1, both image sets, whole frames), flow [B][2][rows_l][cols_l] per level >= 1, one
    // slab per transfer this rank takes part in.  Returns the floats needed.
    size_t layout() {
        total = 0;
        for (int l = 1; l < levels; l++) {
            lvl_elems[l] = (size_t)plan.rows[l] * plan.cols[l];
            pyr_off[l] = total;
            total += pyr_block(l) * 2;  // prev block then next block
        }
        for (int l = 1; l < levels; l++) {
            flow_off[l] = total;
            total += Carver::need(lvl_elems[l] * batch * 2 * 4, 1) / 4;
        }
        for (int l = 1; l < levels; l++) {
            xf[l].clear();
            slab_off[l].clear();
            for (const auto &t : plan.transfers(l))
                if (t.src == me || t.dst == me) {
                    xf[l].push_back(t);
                    slab_off[l].push_back(total);
                    total += Carver::need(slab_floats(l, t) * 4, 1) / 4;
                }
        }
        return total + 64;
    }
    size_t pyr_block(int l) const { return Carver::need(lvl_elems[l] * batch * 4, 1) / 4; }
    size_t slab_floats(int l, const RowPlan::Xfer &t) const { return (size_t)(t.r1 - t.r0) * plan.cols[l] * batch * 2; }
    float *slab(int l, size_t i) const { return mem + slab_off[l][i]; }

    int build() {
        if (levels <= 1) return MICV_OK;
        float *pd[16], *nd[16];
        pd[0] = nd[0] = nullptr;
        for (int l = 1; l < levels; l++) {
            pd[l] = mem + pyr_off[l];
            nd[l] = pd[l] + pyr_block(l);
        }
        return launch_pyr_build2(s, prev, next, pair_stride / 4, (int)(stride / 4), plan.rows[0], plan.cols[0], levels, pd, nd, batch);
    }
    // rows [r0, r1) of every (pair, field) plane of level l's flow block <-> a dense slab: one 2-D copy
    int pack(int l) {
        const size_t plane = lvl_elems[l] * 4;
        for (size_t i = 0; i < xf[l].size(); i++) {
            const auto &t = xf[l][i];
            if (t.src != me) continue;
            const size_t w = (size_t)(t.r1 - t.r0) * plan.cols[l] * 4;
            MICV_HIP(hipMemcpy2DAsync(slab(l, i), w, mem + flow_off[l] + (size_t)t.r0 * plan.cols[l], plane, w, 2 * (size_t)batch,
                                      hipMemcpyDeviceToDevice, s));
        }
        return MICV_OK;
    }
    int unpack(int l) {
        const size_t plane = lvl_elems[l] * 4;
        for (size_t i = 0; i < xf[l].size(); i++) {
            const auto &t = xf[l][i];
            if (t.dst != me) continue;
            const size_t w = (size_t)(t.r1 - t.r0) * plan.cols[l] * 4;
            MICV_HIP(hipMemcpy2DAsync(mem + flow_off[l] + (size_t)t.r0 * plan.cols[l], plane, slab(l, i), w, w, 2 * (size_t)batch,
                                      hipMemcpyDeviceToDevice, s));
        }
        return MICV_OK;
    }
    // the band of level l for all pairs (micv_lk_level_batch_dev: the unsharded kernels on a row range)
    int launch(int l) {
        const int R = plan.rows[l], C = plan.cols[l];
        int a, b;
        plan.band(l, me, &a, &b);
        if (a >= b) return MICV_OK;
        const float *pl = l == 0 ? prev : mem + pyr_off[l];
        const float *nl = l == 0 ? next : mem + pyr_off[l] + pyr_block(l);
        const size_t ps = l == 0 ? pair_stride : lvl_elems[l] * 4, st = l == 0 ? stride : (size_t)C * 4;
        const float *fu = nullptr, *fv = nullptr;
        int fr = 0, fc = 0;
        size_t fps = 0;
        if (l < levels - 1) {
            fr = plan.rows[l + 1];
            fc = plan.cols[l + 1];
            fu = mem + flow_off[l + 1];
            fv = fu + (size_t)fr * fc;
            fps = 2 * (size_t)fr * fc * 4;
        }
        float *ou = u, *ov = v;
        size_t ops = opair_stride, ost = ostride;
        if (l > 0) {
            ou = mem + flow_off[l];
            ov = ou + lvl_elems[l];
            ops = 2 * lvl_elems[l] * 4;
            ost = (size_t)C * 4;
        }
        return micv_lk_level_batch_dev(ctx, pl, nl, batch, ps, R, C, st, win, fu, fv, fr, fc, fps, a, b, ou, ov, ops, ost, s);
    }
};

static int rowshard_check_args(const char *fn, micv_ctx *ctx, const float *prev, const float *next, int batch, size_t pair_stride,
                               int rows, int cols, size_t stride, int win, int levels, float *u, float *v, size_t opair_stride,
                               size_t ostride) {
    MICV_REQUIRE(ctx && prev && next && u && v, "%s: null argument", fn);
    MICV_REQUIRE(rows > 0 && cols > 0 && rows <= 32767 && cols <= 32767, "%s: bad size %dx%d", fn, rows, cols);
    MICV_REQUIRE(stride_ok(stride, cols, 4) && stride_ok(ostride, cols, 4), "%s: bad stride", fn);
    MICV_REQUIRE(win >= 1 && win <= kMaxWin && (win & 1), "%s: window %d must be odd and <= %d", fn, win, kMaxWin);
    MICV_REQUIRE(batch >= 1 && batch <= 32767, "%s: bad batch %d", fn, batch);
    MICV_REQUIRE(levels >= 1 && levels <= 16 && (rows >> (levels - 1)) > 0 && (cols >> (levels - 1)) > 0,
                 "%s: %d levels do not fit a %dx%d image", fn, levels, rows, cols);
    MICV_REQUIRE(pair_stride % 4 == 0 && opair_stride % 4 == 0 &&
                     (batch == 1 || (pair_stride >= stride * (size_t)rows && opair_stride >= ostride * (size_t)rows)),
                 "%s: bad pair stride", fn);
    return MICV_OK;
}

}  // namespace micv

extern "C" {

int micv_lk_flow_pyr_rowshard_dev(micv_ctx *ctx, micv_comm *comm, const float *prev, const float *next, int batch,
                                  size_t pair_stride, int rows, int cols, size_t stride, int win, int levels,
                                  float *u, float *v, size_t opair_stride, size_t ostride, micv_stream stream) {
    MICV_REQUIRE(comm != nullptr, "micv_lk_flow_pyr_rowshard: null communicator");
    MICV_TRY(rowshard_check_args("micv_lk_flow_pyr_rowshard", ctx, prev, next, batch, pair_stride, rows, cols, stride, win, levels,
                                 u, v, opair_stride, ostride));
    MICV_REQUIRE(comm->device == ctx->device, "micv_lk_flow_pyr_rowshard: communicator and context are on different devices");
    RowShardRun run;
    if (!run.plan.build(rows, cols, levels, comm->world, win)) {
        set_error("micv_lk_flow_pyr_rowshard: %d ranks cannot split the %d-row coarsest level", comm->world, rows >> (levels - 1));
        return MICV_EINVAL;
    }
    MICV_HIP(hipSetDevice(ctx->device));
    run.ctx = ctx; run.s = static_cast<hipStream_t>(stream); run.me = comm->rank; run.batch = batch; run.win = win; run.levels = levels;
    run.prev = prev; run.next = next; run.pair_stride = pair_stride; run.stride = stride;
    run.u = u; run.v = v; run.opair_stride = opair_stride; run.ostride = ostride;
    void *base = nullptr;
    MICV_TRY(comm->reserve(run.layout() * 4, &base));
    run.mem = static_cast<float *>(base);
    MICV_TRY(run.build());
    const Rccl *r = rccl();
    for (int l = levels - 1; l >= 0; l--) {
        const int cl = l + 1;
        if (l < levels - 1 && !run.xf[cl].empty()) {
            // halo rows of the coarse flow: pack, ONE grouped send / receive on the launch stream, unpack
            MICV_TRY(run.pack(cl));
            MICV_NCCL(r->GroupStart());
            for (size_t i = 0; i < run.xf[cl].size(); i++) {
                const auto &t = run.xf[cl][i];
                const size_t n = run.slab_floats(cl, t);
                const ncclResult_t nr = t.src == run.me ? r->Send(run.slab(cl, i), n, ncclFloat32, t.dst, comm->comm, run.s)
                                                        : r->Recv(run.slab(cl, i), n, ncclFloat32, t.src, comm->comm, run.s);
                if (nr != ncclSuccess) {
                    (void)r->GroupEnd();
                    set_error("micv_lk_flow_pyr_rowshard: halo exchange of level %d -> %s", cl, r->GetErrorString(nr));
                    return MICV_EHIP;
                }
            }
            MICV_NCCL(r->GroupEnd());
            MICV_TRY(run.unpack(cl));
        }
        MICV_TRY(run.launch(l));
    }
    return MICV_OK;
}

int micv_lk_flow_pyr_rowshard_virtual_dev(micv_ctx *ctx, int world, const float *prev, const float *next, int batch,
                                          size_t pair_stride, int rows, int cols, size_t stride, int win, int levels,
                                          float *u, float *v, size_t opair_stride, size_t ostride, int poison,
                                          micv_stream stream) {
    MICV_TRY(rowshard_check_args("micv_lk_flow_pyr_rowshard_virtual", ctx, prev, next, batch, pair_stride, rows, cols, stride, win,
                                 levels, u, v, opair_stride, ostride));
    MICV_REQUIRE(world >= 1 && world <= 64, "micv_lk_flow_pyr_rowshard_virtual: %d ranks", world);
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<RowShardRun> runs(world);
    std::vector<void *> blocks(world, nullptr);
    auto free_all = [&]() {
        (void)hipStreamSynchronize(s);
        for (void *p : blocks)
            if (p) (void)hipFree(p);
    };
    int rc = MICV_OK;
    for (int g = 0; g < world && rc == MICV_OK; g++) {
        RowShardRun &run = runs[g];
        if (!run.plan.build(rows, cols, levels, world, win)) {
            set_error("micv_lk_flow_pyr_rowshard_virtual: %d ranks cannot split the %d-row coarsest level", world, rows >> (levels - 1));
            rc = MICV_EINVAL;
            break;
        }
        run.ctx = ctx; run.s = s; run.me = g; run.batch = batch; run.win = win; run.levels = levels;
        run.prev = prev; run.next = next; run.pair_stride = pair_stride; run.stride = stride;
        run.u = u; run.v = v; run.opair_stride = opair_stride; run.ostride = ostride;
        const size_t bytes = run.layout() * 4;
        if (hipMalloc(&blocks[g], bytes) != hipSuccess) {
            set_error("micv_lk_flow_pyr_rowshard_virtual: out of device memory");
            rc = MICV_ENOMEM;
            break;
        }
        run.mem = static_cast<float *>(blocks[g]);
        // poison: every byte of the rank's private block is 0xFF (a NaN in every float) before anything is built, so a
        // flow row the rank neither computed nor received shows up in the result
        if (poison && hipMemsetAsync(blocks[g], 0xFF, bytes, s) != hipSuccess) rc = MICV_EHIP;
        if (rc == MICV_OK) rc = run.build();
    }
    for (int l = levels - 1; l >= 0 && rc == MICV_OK; l--) {
        const int cl = l + 1;
        if (l < levels - 1) {
            for (int g = 0; g < world && rc == MICV_OK; g++) rc = runs[g].pack(cl);
            // the transport: the sender's slab of transfer (src, dst, r0, r1) into the receiver's slab of the same transfer
            for (int g = 0; g < world && rc == MICV_OK; g++) {
                RowShardRun &dstr = runs[g];
                for (size_t i = 0; i < dstr.xf[cl].size() && rc == MICV_OK; i++) {
                    const auto &t = dstr.xf[cl][i];
                    if (t.dst != g) continue;
                    RowShardRun &srcr = runs[t.src];
                    size_t j = 0;
                    for (; j < srcr.xf[cl].size(); j++) {
                        const auto &q = srcr.xf[cl][j];
                        if (q.src == t.src && q.dst == t.dst && q.r0 == t.r0 && q.r1 == t.r1) break;
                    }
                    if (j == srcr.xf[cl].size()) {
                        set_error("micv_lk_flow_pyr_rowshard_virtual: rank %d has no send for rank %d's receive", t.src, g);
                        rc = MICV_EINVAL;
                        break;
                    }
                    if (hipMemcpyAsync(dstr.slab(cl, i), srcr.slab(cl, j), dstr.slab_floats(cl, t) * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
                        rc = MICV_EHIP;
                }
            }
            for (int g = 0; g < world && rc == MICV_OK; g++) rc = runs[g].unpack(cl);
        }
        for (int g = 0; g < world && rc == MICV_OK; g++) rc = runs[g].launch(l);
    }
    free_all();
    return rc;
}

int micv_lk_flow_pyr_rowshard_host(micv_ctx *ctx, micv_comm *comm, const float *prev, const float *next, int rows, int cols,
                                   size_t stride, int win, int levels, float *u, float *v, size_t ostride) {
    MICV_REQUIRE(ctx && comm && prev && next && u && v, "micv_lk_flow_pyr_rowshard_host: null argument");
    MICV_REQUIRE(rows > 0 && cols > 0 && stride_ok(stride, cols, 4) && stride_ok(ostride, cols, 4),
                 "micv_lk_flow_pyr_rowshard_host: bad size / stride");
    MICV_HIP(hipSetDevice(ctx->device));
    const size_t rb = (size_t)cols * 4, n = rb * rows;
    // frames and full-size outputs on the device: the context's cached blocks, as every other `_host` entry point
    // (host_api.hip) -- a repeated call of the same shape does no hipMalloc / hipFree (this one did four per
    // call).  The communicator's own block is in use by the driver.
    MICV_REQUIRE(comm->device == ctx->device, "micv_lk_flow_pyr_rowshard_host: communicator and context are on different devices");
    float *dp = static_cast<float *>(ctx->io_acquire(n)), *dn = static_cast<float *>(ctx->io_acquire(n));
    float *du = static_cast<float *>(ctx->io_acquire(n)), *dv = static_cast<float *>(ctx->io_acquire(n));
    hipStream_t s = nullptr;  // the null stream: every call of this entry point is synchronous, like the reference's
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);  // nothing may still use a block that goes back to the cache
        for (float *p : {dp, dn, du, dv})
            if (p) ctx->io_release(p);
    };
    if (!dp || !dn || !du || !dv) {
        cleanup();
        set_error("micv_lk_flow_pyr_rowshard_host: device allocation failed");
        return MICV_ENOMEM;
    }
#define MICV_RS(expr)                 \
    do {                              \
        const int rc_ = (expr);       \
        if (rc_ != MICV_OK) {         \
            cleanup();                \
            return rc_;               \
        }                             \
    } while (0)
#define MICV_RS_HIP(expr)                                                          \
    do {                                                                           \
        const hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                                    \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            cleanup();                                                             \
            return MICV_EHIP;                                                      \
        }                                                                          \
    } while (0)
    MICV_RS_HIP(hipMemcpy2DAsync(dp, rb, prev, stride, rb, rows, hipMemcpyHostToDevice, s));
    MICV_RS_HIP(hipMemcpy2DAsync(dn, rb, next, stride, rb, rows, hipMemcpyHostToDevice, s));
    MICV_RS(micv_lk_flow_pyr_rowshard_dev(ctx, comm, dp, dn, 1, 0, rows, cols, rb, win, levels, du, dv, 0, rb, s));
    // every rank's band to every rank: the cv::Mat caller gets whole fields back (lk::calcOpticalFlowPyr's contract)
    if (comm->world > 1) {
        RowPlan plan;
        plan.build(rows, cols, levels, comm->world, win);
        const Rccl *r = rccl();
        ncclResult_t nr = r->GroupStart();
        for (int g = 0; g < comm->world && nr == ncclSuccess; g++) {
            int a, b;
            plan.band(0, g, &a, &b);
            const size_t cnt = (size_t)(b - a) * cols;
            if (cnt == 0) continue;
            nr = r->Broadcast(du + (size_t)a * cols, du + (size_t)a * cols, cnt, ncclFloat32, g, comm->comm, s);
            if (nr == ncclSuccess) nr = r->Broadcast(dv + (size_t)a * cols, dv + (size_t)a * cols, cnt, ncclFloat32, g, comm->comm, s);
        }
        const ncclResult_t ne = r->GroupEnd();
        if (nr != ncclSuccess || ne != ncclSuccess) {
            set_error("micv_lk_flow_pyr_rowshard_host: gathering the bands -> %s", r->GetErrorString(nr != ncclSuccess ? nr : ne));
            cleanup();
            return MICV_EHIP;
        }
    }
    MICV_RS_HIP(hipMemcpy2DAsync(u, ostride, du, rb, rb, rows, hipMemcpyDeviceToHost, s));
    MICV_RS_HIP(hipMemcpy2DAsync(v, ostride, dv, rb, rb, rows, hipMemcpyDeviceToHost, s));
    MICV_RS_HIP(hipStreamSynchronize(s));
#undef MICV_RS
#undef MICV_RS_HIP
    cleanup();
    return MICV_OK;
}

}  // extern "C"
