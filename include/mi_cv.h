/*
 * mi_cv.h -- C ABI of the MI355X-native per-pixel CV kernel library (libmicv.so).
 *
 * Drop-in boundary for the cv::Mat-in / cv::Mat-out functions of
 * tanmaniac/IntroToComputerVision (paths below are relative to that repository).
 * Every entry point replaces one reference function; the header-only C++ shim in
 * introtocomputervision_amd/shim/micv_shim.hpp puts the reference's namespaces and
 * signatures (lk::, pyr::, harris::, sift::, cuda::, serial::) on top of these calls.
 *
 * Conventions
 *   - plain pointers + sizes; no C++ / torch / OpenCV types.
 *   - images are row-major, single channel; `*_stride` is the row pitch in BYTES
 *     (cv::Mat::step), must be a multiple of the element size.
 *   - `_dev` functions take DEVICE pointers, enqueue on `stream` (a hipStream_t, NULL =
 *     default stream) and do not synchronise; `_host` functions take HOST pointers, do
 *     H2D / D2H themselves and return after the result is in host memory (this is the
 *     behaviour of the reference's functions, which upload/download inside every call).
 *   - return value: MICV_OK (0) or a negative MICV_E* code; micv_last_error() returns a
 *     thread-local message.  Nothing calls exit() (the reference's checkCudaErrors does,
 *     common/include/common/CudaCommon.cuh:13-22).
 *   - a micv_ctx owns the device ordinal and a scratch arena that is reused between
 *     calls; use one ctx per host thread / stream (the reference's functions are not
 *     re-entrant either: global texture references, Harris.cu:28-31).
 */
#ifndef MI_CV_H
#define MI_CV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MICV_OK            0
#define MICV_EINVAL       -1   /* bad argument (size, stride, window, null pointer) */
#define MICV_EHIP         -2   /* HIP runtime error (message has hipGetErrorString) */
#define MICV_ENOMEM       -3   /* device allocation failed */
#define MICV_EUNSUPPORTED -4   /* valid in the reference but not implemented here */

typedef struct micv_ctx micv_ctx;
typedef void *micv_stream; /* hipStream_t */
typedef struct micv_comm micv_comm; /* a communicator over RCCL (multi-GPU entry points, below) */

const char *micv_version(void);
const char *micv_last_error(void);

int micv_ctx_create(int device, micv_ctx **out);
void micv_ctx_destroy(micv_ctx *ctx);
/* Bytes of device scratch currently held by the context. */
size_t micv_ctx_scratch_bytes(const micv_ctx *ctx);
/* Execution options of a context.  None of them changes a result: they select between kernels
 * that produce identical bits (tests compare the alternatives) or how a batch is scheduled.
 * The library reads NO environment variables. */
#define MICV_OPT_LK_STREAM_GROUPS  1 /* stream groups a batch is split into: 0 = default (1), 1..4 */
#define MICV_OPT_LK_FORCE_GENERIC  2 /* LK through the generic kernels: 1 = two or four launches per level by size (as for windows the fused kernels do not cover), 2 = always four, 3 = always two */
#define MICV_OPT_LK_NARROW_TILES   3 /* win-15 level kernel: 256-thread tiles instead of 512 */
#define MICV_OPT_SOBEL_GENERIC     4 /* Sobel through the generic row / column passes */
#define MICV_OPT_HARRIS_GENERIC    5 /* Harris response: one-thread-per-pixel kernel */
#define MICV_OPT_NMS_SCAN          6 /* Harris NMS: scanning kernel instead of the separable one */
#define MICV_OPT_STEREO_ROWS       7 /* rows per stereo strip: 0 = automatic, 8 or 10 */
#define MICV_OPT_LK_CHAIN          8 /* fused LK tile chains: 0 = pairs of tiles only for launches a little over one or two rounds of workgroups (default), 1 = never, n = longest chain (<= 32), -1 = schedule only */
#define MICV_OPT_LK_SHORT_TILES    9 /* win-15 level kernel, 64x16 tiles: 0 = up to 512 tiles (one round), n = up to n, -1 = never */
#define MICV_OPT_LK_STREAM         10 /* level kernel as a persistent grid that stages the next tile ahead: 1 = on, 0 = off (default; measured slower) */
#define MICV_OPT_LK_TALL_TILES     11 /* 1024-thread tiles, one workgroup per CU, for big launches: window 15 on 64x64 tiles only with 1 (measured slower); window 21 on 64x32 tiles by default (0 or 1; measured faster); 2 = window 15 on 32x64 tiles, 512 threads, two workgroups per CU; 3 = window 15 on 64x32 tiles, 1024 threads, eight waves per SIMD (r04 experiments, both measured slower: DESIGN.md section 5); -1 = never */
#define MICV_OPT_COMPACT_3PASS     12 /* ordered lists (corners, edge points, peak candidates, matches): 0 = one-launch chained scan up to 1 M elements, count / scan / emit launches beyond; 1 = always three launches; -1 = always one */
#define MICV_OPT_LK_DIRECT_LEVELS  13 /* fused LK: pyramid levels >= n read straight from level 0 with a pixel stride (n = 1: no pyramid-build launch): 0 = off (default: measured faster one pass at a time, slower with two passes in flight), n = 1..15 */
#define MICV_OPT_LK_BUILD_OVERLAP  14 /* fused LK, window 15, >= 3 levels: no pyramid-build launch -- the top level reads level 0 itself and the launch of every level k >= 2 carries the build of level k - 1 as extra workgroups behind its tiles: 0 = for single pairs only (default: the latency case; with several passes in flight a batch is faster with the build launch), -1 = never, 1 = for every batch */
#define MICV_OPT_LK_SPLIT          15 /* fused LK, window 15: a level launch SPLIT in two (r05, lk_split.hip) -- a pre-pass (pyrUp + warp + Sobel once per pixel, Ix / Iy / It into padded planes in HBM) plus a streaming window-sum kernel that carries its row-pass rows from block to block: 0 = never (default: measured 28-42 % slower than the fused launch, HBM traffic 2.4-3x), 1 = every whole-frame launch of at least 64 x 64 with a doubling coarse flow, 2 = 1 with the base flow through u, v, 3 = the pre-pass leaves only the warped image and the second half is the fused kernel in its no-flow mode */
#define MICV_OPT_LK_STRIP          16 /* fused LK, window 15: the INTERIOR tiles of a level launch as vertical strips that a workgroup streams down in blocks of 16 rows, carrying the last 14 row-pass rows of the five product fields and two warped rows from block to block (lk_strip.hpp) -- no vertical halo recomputation; the border tiles of the same launch stay tiles: 0 = off (default: 15 % fewer instructions, the same time -- DESIGN.md section 5), n > 0 = on with row segments of n blocks (16 = 256 rows); + 8192 = only for launches of 4096 tiles or more */
#define MICV_OPT_STEREO_EXACT      17 /* disparitySSD on 8-bit-valued images (integers 0..255 in both f32 images, radius <= 7; serial:: semantics radius <= 5): 0 = the exact-sum kernels (stereo_exact.hip: integer dot products, sliding window sums, lanes = disparities), chosen on the device per call by a pre-pass that tests every pixel (default); -1 = always the float kernels that add every window in the contract's order.  Same disparities either way: all sums are integers below 2^24, exact in f32 in any order.  (disparityNCorr always takes the float kernels.) */
#define MICV_OPT_COUNT            18
int micv_ctx_set_option(micv_ctx *ctx, int option, int value);
int micv_ctx_get_option(const micv_ctx *ctx, int option, int *value);

/* Device memory for callers that keep images resident between calls (the role cv::cuda::GpuMat's
 * allocator plays for the reference's GpuMat overloads, ps1_cpp/src/Hough.h:22-25,48-51,73-75).
 * Copies are blocking, like GpuMat::upload / download without a Stream; pitches in bytes. */
int micv_device_malloc(micv_ctx *ctx, size_t bytes, void **out);
void micv_device_free(micv_ctx *ctx, void *p);
int micv_memcpy2d_h2d(micv_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch,
                      size_t width_bytes, int rows);
int micv_memcpy2d_d2h(micv_ctx *ctx, void *dst, size_t dpitch, const void *src, size_t spitch,
                      size_t width_bytes, int rows);

/* ---------------------------------------------------------------- common/ (a17) ---- */
/* Kernel-timing log lines.  The reference brackets each kernel with a GpuTimer and logs
 * "<kernel> execution took {} ms" through spdlog (ps4_cpp/lib/Harris.cu:144-155,290;
 * ps2_cpp/lib/DisparitySSD.cu:192-203, DisparityNCorr.cu:236-247; ps1_cpp/src/Hough.cu:289,345,391;
 * ps5_cpp/lib/Pyramids.cu:69,123).  With a sink registered here (process-wide; NULL removes it), the
 * `_host` entry points of exactly those functions time their device call with an event pair and call
 * fn(kernel, ms, user) after their final synchronisation, `kernel` being the reference's kernel name
 * ("cornerResponseKernel", "refineCornersKernel", "disparitySSDKernel", "disparityNCorrKernel",
 * "houghLinesAccumulateKernel", "houghCirclesAccumulateKernel", "findLocalMaximaKernel",
 * "pyrDownsampleKernel", "pyrUpsampleKernel").  The shim formats the reference's lines from it
 * (micv_shim::log_kernel_times_to).  Costs nothing when no sink is set. */
typedef void (*micv_kernel_log_fn)(const char *kernel, float ms, void *user);
int micv_set_kernel_log(micv_kernel_log_fn fn, void *user);

/* common::warmup, common/src/CudaWarmup.cu:5-19 (10 blocks x 64 threads). */
int micv_warmup(micv_ctx *ctx, micv_stream stream);
/* common::divRoundUp, common/include/common/Utils.h:12-15: max(1, ceil(float(n)/float(d))). */
size_t micv_div_round_up(size_t num, size_t denom);
/* GpuTimer, common/include/common/GpuTimer.h:5-22 (event pair; stop() synchronises). */
typedef struct micv_timer micv_timer;
int micv_timer_create(micv_timer **out);
int micv_timer_start(micv_timer *t, micv_stream stream);
int micv_timer_stop(micv_timer *t, micv_stream stream);
int micv_timer_elapsed_ms(micv_timer *t, float *ms);
void micv_timer_destroy(micv_timer *t);

/* Per-launch timing of the pyramid-level kernels: the reference wraps every kernel launch in
 * a GpuTimer and logs "<kernel> took {} ms" (Pyramids.cu:61-69, Harris.cu:144-155); this is the
 * same measurement without the log.  While enabled, micv_lk_flow_pyr*_dev brackets each
 * pyramid level's launch(es) with an event pair on the caller's stream.
 * micv_profile_lk_level synchronises on those events and returns their summed duration. */
int micv_profile_enable(micv_ctx *ctx, int on);
int micv_profile_reset(micv_ctx *ctx);
int micv_profile_lk_level(micv_ctx *ctx, int level, double *total_ms, int64_t *launches);
/* Frame pairs covered by each profiled level launch: micv_lk_flow_pyr_batch_dev splits a batch
 * into groups that run on separate streams (MICV_OPT_LK_STREAM_GROUPS, default 1 = no split); the events bracket
 * the launches of the first group. */
int micv_profile_lk_pairs(micv_ctx *ctx, int *pairs_per_launch);
/* In-kernel phase stamps of the fused LK level kernel.  Compiled in only with -DMICV_DIAG (a
 * diagnostic build for timing studies; the default build returns MICV_EUNSUPPORTED when asked to
 * enable them): while enabled, wave 0 of every workgroup adds the s_memtime ticks it spent
 * in each phase to 16 device counters ([0..5] interior tiles, [8..13] border tiles: stage, pyrUp
 * rows, warp, gradients, window sums, solve).  Reads the counters into ticks16 (may be NULL),
 * then enables/disables and zeroes them.  Synchronises the device. */
int micv_profile_lk_phases(micv_ctx *ctx, int enable, uint64_t *ticks16);

/* --------------------------------------------------------- ps5: LK + pyramids ------ */

/* lk::calcOpticalFlowPyr, ps5_cpp/lib/OpticalFlow.cpp:122-167 (a1).  `levels` replaces
 * the hard-coded pyrDepth = 4 (:127).  Inputs are single-channel f32 (grey); u, v are
 * rows x cols f32.  win odd, 1..63. */
int micv_lk_flow_pyr_dev(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                         size_t stride, int win, int levels, float *u, float *v, size_t ostride,
                         micv_stream stream);
int micv_lk_flow_pyr_host(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                          size_t stride, int win, int levels, float *u, float *v, size_t ostride);
/* The same over `batch` independent frame pairs in one set of launches.  Pair i lives at
 * prev + i*pair_stride (bytes), likewise next / u / v with opair_stride. */
int micv_lk_flow_pyr_batch_dev(micv_ctx *ctx, const float *prev, const float *next, int batch,
                               size_t pair_stride, int rows, int cols, size_t stride, int win,
                               int levels, float *u, float *v, size_t opair_stride,
                               size_t ostride, micv_stream stream);

/* Row-sharded execution with a DECLARED bound on the vertical flow: a rank that holds
 * only band + margin rows of `next` must know when a flow value would make lk::warp read beyond them.
 * Sets bit 0 of *flag (device memory, zeroed by the caller) when |v| > bound or v is not finite anywhere in
 * rows [row_begin, row_end) of the `batch` fields (field i at v + i*pair_stride bytes); asynchronous on
 * `stream`, no host synchronisation.  introtocomputervision_amd/shard.py reruns the step with the whole
 * frame when the flag comes back set. */
int micv_flow_bound_check_dev(micv_ctx *ctx, const float *v, int batch, size_t pair_stride, int rows, int cols,
                              size_t stride, int row_begin, int row_end, float bound, uint32_t *flag,
                              micv_stream stream);

/* ------------------------------------------------- multi-GPU: one process per GPU, RCCL -------- */
/* SURVEY.md section 8e.  The reference has no multi-GPU code; its caller (`denseLKWrapper`,
 * ps5_cpp/src/Solution.cpp:60-64) is what these entry points let shard: every rank (one process per GPU) holds the
 * whole frames and computes a band of rows; the only dynamic exchange is the coarse-flow halo per pyramid level --
 * (win/2 + 2)/2 + 3 rows per neighbour -- sent point to point (ncclSend / ncclRecv in one group, on the launch
 * stream).  RCCL is loaded at run time (dlopen "librccl.so.1": libmicv.so itself links only libamdhip64, and a
 * process that already holds an RCCL shares it); without it these calls return MICV_EUNSUPPORTED.
 *
 * A communicator wraps an existing ncclComm_t (borrowed, e.g. the application's own) or is created from an RCCL
 * unique id: rank 0 calls micv_comm_unique_id, ships the MICV_COMM_ID_BYTES bytes to the other ranks by any means
 * (MPI, a file, torch.distributed), every rank calls micv_comm_create(ctx, NULL, id, rank, world, &comm) -- a
 * collective call, like ncclCommInitRank. */
#define MICV_COMM_ID_BYTES 128
int micv_comm_unique_id(void *id128);
int micv_comm_create(micv_ctx *ctx, void *nccl_comm, const void *unique_id128, int rank, int world, micv_comm **out);
int micv_comm_destroy(micv_comm *comm);
int micv_comm_rank(const micv_comm *comm, int *rank, int *world);
/* A communicator owns ONE device block (pyramids, per-level flow, exchange slabs) that every sharded call carves anew:
 * use a communicator from ONE stream at a time (calls on one stream queue up correctly; two streams would race on the
 * block silently), and with the context of the device it was created on (checked: MICV_EINVAL otherwise). */
/* Fabric check to run once before timing or trusting sharded results (collective, synchronises `stream`): a ring of
 * grouped ncclSend / ncclRecv of a rank-stamped 256 KB slab -- the row-shard driver's exchange pattern -- and one int32
 * sum all-reduce, both verified ON THE DEVICE.  MICV_OK, or MICV_EHIP with micv_last_error() naming the rank, the peer
 * and the step: a fabric or ordering failure then reads as such, not as a parity mismatch of the flow. */
int micv_comm_selftest(micv_ctx *ctx, micv_comm *comm, micv_stream stream);
/* The row plan, host only: rows [row_begin, row_end) of pyramid level `level` that `rank` of `world` computes
 * (the coarsest level is cut evenly, finer levels double the cuts), and optionally the rows of that level it
 * needs to compute the next finer one (its band + halo). */
int micv_rowshard_band(int rows, int cols, int levels, int world, int win, int rank, int level, int *row_begin,
                       int *row_end, int *need_begin, int *need_end);
/* lk::calcOpticalFlowPyr (OpticalFlow.cpp:122-167) of `batch` pairs, every pair split by rows over the ranks of
 * `comm`: this rank writes rows micv_rowshard_band(..., level 0) of u / v (full-size buffers) and nothing else.
 * prev / next: the whole frames on every rank.  Same bits as micv_lk_flow_pyr_batch_dev.  Collective: every rank
 * calls it with the same sizes.  Asynchronous on `stream`. */
int micv_lk_flow_pyr_rowshard_dev(micv_ctx *ctx, micv_comm *comm, const float *prev, const float *next, int batch,
                                  size_t pair_stride, int rows, int cols, size_t stride, int win, int levels,
                                  float *u, float *v, size_t opair_stride, size_t ostride, micv_stream stream);
/* The same plan, packing and band launches for `world` VIRTUAL ranks in this process on one device, the exchange
 * done by copies between the ranks' private slabs: whole u / v come back (every rank writes its band).  What a
 * one-GPU box can run at world sizes > 1; `poison` != 0 fills every rank's private memory with NaNs first, so a halo
 * row that was neither computed nor received shows up in the result.  Synchronises `stream` before it returns. */
int micv_lk_flow_pyr_rowshard_virtual_dev(micv_ctx *ctx, int world, const float *prev, const float *next, int batch,
                                          size_t pair_stride, int rows, int cols, size_t stride, int win, int levels,
                                          float *u, float *v, size_t opair_stride, size_t ostride, int poison,
                                          micv_stream stream);
/* The cv::Mat caller's form: host frames in, the WHOLE flow fields out on every rank (the bands are gathered
 * with one broadcast per rank and field); synchronous. */
int micv_lk_flow_pyr_rowshard_host(micv_ctx *ctx, micv_comm *comm, const float *prev, const float *next, int rows,
                                   int cols, size_t stride, int win, int levels, float *u, float *v, size_t ostride);
/* cuda::houghLinesAccumulate (Hough.cu:251-309) with the edge points sharded by rows: micv_hough_lines_band_dev
 * into this rank's private accumulator, then ONE int32 sum all-reduce in place -- the only real collective of the
 * whole path; integer sums make it bit-exact in any order.  acc: rho_bins x theta_bins on every rank. */
int micv_hough_lines_rowshard_dev(micv_ctx *ctx, micv_comm *comm, const uint8_t *mask_band, int band_rows, int cols,
                                  size_t mstride, int row0, int rows, unsigned rho_bin, unsigned theta_bin,
                                  int32_t *acc, micv_stream stream);
int micv_allreduce_sum_i32_dev(micv_ctx *ctx, micv_comm *comm, int32_t *buf, size_t count, micv_stream stream);

/* Diagnostic, host only (no device call): the work list the optional chain / streamed launches of the
 * level kernel walk (MICV_OPT_LK_CHAIN, MICV_OPT_LK_STREAM) for a rows x cols level of `batch` pairs.
 * Entry i = (tile x, first tile y, tiles in the chain, pair), 0 tiles = padding; tiles are
 * tile_w x tile_h pixels.  *count = entries of the list (also when entries_xycp is NULL or smaller).
 * Every tile of every pair must appear in exactly one entry -- tests/test_capi_and_host.py checks it. */
/* Diagnostic, no launch: the name (as rocprofv3 prints it, without the `micv::` prefix and the argument list) of the
 * kernel instantiation(s) the fused path launches for a rows x cols pyramid level of `batch` pairs with a doubling
 * coarse flow under this context's options -- answered by the launch dispatch itself, so that tools filtering profiler
 * rows by kernel name cannot drift from it.  cap >= 96. */
int micv_lk_level_kernel_name(micv_ctx *ctx, int win, int rows, int cols, int batch, char *buf, size_t cap);
int micv_lk_schedule_host(int rows, int cols, int batch, int win, int max_chain, int32_t *entries_xycp,
                          int64_t capacity, int64_t *count, int *tile_w, int *tile_h);

/* One iteration of the coarse-to-fine loop of lk::calcOpticalFlowPyr (OpticalFlow.cpp:135-163) on
 * one pyramid level, restricted to output rows [row_begin, row_end): the building block of
 * row-sharded execution (a rank owns a band of rows of every level and exchanges only coarse-flow
 * halo rows with its neighbours; introtocomputervision_amd/shard.py).  prev/next are the level's
 * images (rows x cols); flow_u/flow_v the coarser level's flow (flow_rows x flow_cols, dense pitch)
 * or NULL at the coarsest level.  Output rows outside the band may or may not be written. */
int micv_lk_level_dev(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                      size_t stride, int win, const float *flow_u, const float *flow_v,
                      int flow_rows, int flow_cols, int row_begin, int row_end, float *u, float *v,
                      size_t ostride, micv_stream stream);
/* The same over `batch` pairs in one launch per level (pair i at prev + i*pair_stride bytes, its
 * coarse flow at flow_u + i*flow_pair_stride, its output at u + i*opair_stride): what a rank of a
 * row-sharded batch runs per level. */
int micv_lk_level_batch_dev(micv_ctx *ctx, const float *prev, const float *next, int batch,
                            size_t pair_stride, int rows, int cols, size_t stride, int win,
                            const float *flow_u, const float *flow_v, int flow_rows, int flow_cols,
                            size_t flow_pair_stride, int row_begin, int row_end, float *u, float *v,
                            size_t opair_stride, size_t ostride, micv_stream stream);

/* lk::calcOpticalFlow, OpticalFlow.cpp:41-104 (a2; includes computeGradients :12-39, a3). */
int micv_lk_flow_dev(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                     size_t stride, int win, float *u, float *v, size_t ostride,
                     micv_stream stream);
int micv_lk_flow_host(micv_ctx *ctx, const float *prev, const float *next, int rows, int cols,
                      size_t stride, int win, float *u, float *v, size_t ostride);

/* lk::warp, OpticalFlow.cpp:106-120 (a4).  src/du/dv/dst are rows x cols f32. */
int micv_lk_warp_dev(micv_ctx *ctx, const float *src, size_t sstride, const float *du,
                     const float *dv, size_t fstride, int rows, int cols, float *dst,
                     size_t dstride, micv_stream stream);
int micv_lk_warp_host(micv_ctx *ctx, const float *src, size_t sstride, const float *du,
                      const float *dv, size_t fstride, int rows, int cols, float *dst,
                      size_t dstride);

/* pyr::pyrDown, ps5_cpp/lib/Pyramids.cu:34-73 (a5): dst is (rows/2) x (cols/2). */
int micv_pyr_down_dev(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                      float *dst, size_t dstride, micv_stream stream);
int micv_pyr_down_host(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                       float *dst, size_t dstride);
/* pyr::pyrUp, Pyramids.cu:94-131 (a6): dst is (2 rows) x (2 cols). */
int micv_pyr_up_dev(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                    float *dst, size_t dstride, micv_stream stream);
int micv_pyr_up_host(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                     float *dst, size_t dstride);
/* pyr::makeGaussianPyramid, ps5_cpp/lib/Pyramids.cpp:5-26 (a7).  Level 0 is a copy of the
 * (grey f32) input; level l is (rows>>l) x (cols>>l), written densely (pitch = cols_l*4)
 * at dst_levels[l].  All levels are produced by one launch. */
int micv_gaussian_pyramid_dev(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                              int levels, float *const *dst_levels, micv_stream stream);
int micv_gaussian_pyramid_host(micv_ctx *ctx, const float *src, int rows, int cols,
                               size_t sstride, int levels, float *const *dst_levels);
/* The Laplacian pyramid of sol::runProblem2, ps5_cpp/src/Solution.cpp:187-200 (SURVEY.md §8f row
 * N4): L_i = G_i - resize_if_smaller(pyrUp(G_{i+1})), L_{levels-1} = G_{levels-1}; level l is
 * (rows>>l) x (cols>>l), written densely at dst_levels[l].  Device-resident throughout. */
int micv_laplacian_pyramid_dev(micv_ctx *ctx, const float *src, int rows, int cols,
                               size_t sstride, int levels, float *const *dst_levels,
                               micv_stream stream);
/* The same for `batch` images in one launch (image i at src + i*image_stride bytes; level l of image i
 * at dst_levels[l] + i*rows_l*cols_l floats, dense; dst_levels[0] may be NULL = no level-0 copy).
 * row_begin / row_end (both NULL = everything): level l is written for rows
 * [row_begin[l], row_end[l]) only -- a rank of a row-sharded run builds just the rows its band and
 * halos touch. */
int micv_gaussian_pyramid_batch_dev(micv_ctx *ctx, const float *src, int batch, size_t image_stride,
                                    int rows, int cols, size_t sstride, int levels, float *const *dst_levels,
                                    const int *row_begin, const int *row_end, micv_stream stream);
/* cv::cvtColor(COLOR_RGB2GRAY) + convertTo(CV_32F) for 8-bit 3-channel input
 * (Pyramids.cpp:10-15). */
int micv_rgb8_to_gray_f32_dev(micv_ctx *ctx, const uint8_t *rgb, int rows, int cols,
                              size_t sstride, float *dst, size_t dstride, micv_stream stream);
/* The general form of the same step, as pyr::makeGaussianPyramid (Pyramids.cpp:9-15) and
 * denseLKWrapper (ps5_cpp/src/Solution.cpp:48-56) apply it to whatever cv::imread returned:
 * `channels` 1 (convertTo only), 3 or 4 interleaved samples per pixel (cv::cvtColor(COLOR_RGB2GRAY)
 * accepts both; alpha is ignored), `depth` MICV_DEPTH_8U (fixed-point weights 4899/9617/1868 >> 14,
 * rounded) or MICV_DEPTH_32F ((c0*0.299f + c1*0.587f) + c2*0.114f, unfused).  sstride in bytes. */
#define MICV_DEPTH_8U  0 /* CV_8U  */
#define MICV_DEPTH_32F 5 /* CV_32F */
int micv_to_gray_f32_dev(micv_ctx *ctx, const void *src, int rows, int cols, size_t sstride,
                         int channels, int depth, float *dst, size_t dstride, micv_stream stream);
int micv_to_gray_f32_host(micv_ctx *ctx, const void *src, int rows, int cols, size_t sstride,
                          int channels, int depth, float *dst, size_t dstride);
/* lk::calcOpticalFlowPyr on the frames as the unchanged ps5 caller passes them: denseLKWrapper hands
 * the COLOUR frames to lk::calcOpticalFlowPyr (ps5_cpp/src/Solution.cpp:63) and
 * makeGaussianPyramid converts them (Pyramids.cpp:9-15).  One upload of the interleaved frames,
 * grey conversion on the device, then micv_lk_flow_pyr_dev. */
int micv_lk_flow_pyr_frames_host(micv_ctx *ctx, const void *prev, const void *next, int rows, int cols,
                                 size_t stride, int channels, int depth, int win, int levels, float *u,
                                 float *v, size_t ostride);
/* lk::calcOpticalFlowPyr over a sequence of frames: pairs (0, 1), (1, 2), ..., (nframes - 2, nframes - 1) -- the way
 * the ps5 driver walks the frames of a directory (ps5_cpp/lib/Config.cpp:17-46, src/Solution.cpp:255-285; frame t is
 * `next` of one lk::calcOpticalFlowPyr call and `prev` of the following one).  frames[t]: nframes host images of one
 * format (as micv_lk_flow_pyr_frames_host); u[p], v[p]: nframes - 1 host outputs, rows x cols f32, ostride bytes.
 * Every frame crosses PCIe once; upload of frame t + 2, the chain of pair t + 1 and the download of pair t run at
 * the same time (uploads on the calling thread, the chains on their own stream, downloads on two helper threads: a
 * copy from / to the caller's pageable memory occupies the thread that issues it).  Byte-identical to nframes - 1 calls
 * of micv_lk_flow_pyr_frames_host.  Blocking; one call at a time per context, like every other entry point. */
int micv_lk_flow_seq_host(micv_ctx *ctx, const void *const *frames, int nframes, int rows, int cols, size_t stride,
                          int channels, int depth, int win, int levels, float *const *u, float *const *v,
                          size_t ostride);
/* cv::resize(..., INTER_LINEAR) on f32 as used at OpticalFlow.cpp:149-150. */
int micv_resize_linear_dev(micv_ctx *ctx, const float *src, int srows, int scols, size_t sstride,
                           float *dst, int drows, int dcols, size_t dstride, micv_stream stream);

/* ------------------------------------------------------------- ps4: Harris --------- */

/* harris::getGradients, ps4_cpp/lib/Harris.cpp:14-41 (a8) and computeGradients,
 * OpticalFlow.cpp:12-39 (a3): Sobel pair, ksize in {1,3,5,7}, scale folded into the
 * smoothing taps (1 for Harris, 1/9 for LK). */
int micv_sobel_dev(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride, int ksize,
                   float scale, float *gx, float *gy, size_t gstride, micv_stream stream);
int micv_sobel_host(micv_ctx *ctx, const float *src, int rows, int cols, size_t sstride,
                    int ksize, float scale, float *gx, float *gy, size_t gstride);

/* harris::{cpu,gpu}::getCornerResponse, Harris.cpp:43-97 / Harris.cu:96-159 (a9). */
int micv_harris_response_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                             size_t gstride, int win, double sigma, float alpha, float *resp,
                             size_t rstride, micv_stream stream);
int micv_harris_response_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                              size_t gstride, int win, double sigma, float alpha, float *resp,
                              size_t rstride);
/* The same with the arithmetic selected.  flags = 0: harris::gpu (Harris.cu:36-43,85-91: `fma.rn` accumulation in
 * (wy, wx) raster order, float `det - alpha tr^2`) -- the default of both functions above, because the reference's
 * configuration sets use_gpu: true (config/ps4.yaml:16).  MICV_HARRIS_CPU: harris::cpu::getCornerResponse as
 * written (ps4_cpp/lib/Harris.cpp:78-92): `secondMoment + weight * gradVals` is a multiply then an add per element,
 * cv::determinant is taken in double, `harrisScore * trace * trace` in float, and the difference is rounded to
 * float once.  Same window, weights, clamping and raster order. */
#define MICV_HARRIS_CPU 1
int micv_harris_response_ex_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                                size_t gstride, int win, double sigma, float alpha, int flags, float *resp,
                                size_t rstride, micv_stream stream);
int micv_harris_response_ex_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                                 size_t gstride, int win, double sigma, float alpha, int flags, float *resp,
                                 size_t rstride);
/* harris::{cpu,gpu}::refineCorners, Harris.cpp:99-147 / Harris.cu:243-329 (a10).
 * corners: rows x cols f32, zero except kept maxima.  locs_yx: capacity `cap` (y,x) int32
 * pairs, filled in row-major order; *count receives the number found (may exceed cap).
 * The _dev flavour leaves count/locs in device memory. */
int micv_harris_refine_dev(micv_ctx *ctx, const float *resp, int rows, int cols, size_t rstride,
                           double threshold, int min_distance, float *corners, size_t cstride,
                           int32_t *locs_yx, int64_t cap, int64_t *count, micv_stream stream);
int micv_harris_refine_host(micv_ctx *ctx, const float *resp, int rows, int cols, size_t rstride,
                            double threshold, int min_distance, float *corners, size_t cstride,
                            int32_t *locs_yx, int64_t cap, int64_t *count);

/* The ps4 caller's chain as ONE device call (harrisHelper, ps4_cpp/src/Solution.cpp:77-124: harris::getGradients ->
 * getCornerResponse -> refineCorners): image -> [gradients] -> R -> ordered corner list.  With a 3x3 Sobel and a window
 * of 3 / 5 / 7 the gradients are formed inside the response kernel's LDS tile (image in, R out: 8 B per pixel instead of
 * 12 + 12, two launches instead of three); other sizes run the three launches of the separate entry points.  Same bits
 * as micv_sobel_dev(scale 1) + micv_harris_response_ex_dev + micv_harris_refine_dev either way.
 * Optional outputs (NULL = not wanted): gx / gy (both or neither; sift::getKeypoints reads them), resp (R; context
 * scratch otherwise), corners (the sparse map harris::refineCorners also returns).  locs_yx / cap / count as
 * micv_harris_refine_dev; flags as micv_harris_response_ex_dev. */
int micv_harris_corners_dev(micv_ctx *ctx, const float *img, int rows, int cols, size_t stride, int sobel_ksize, int win,
                            double sigma, float alpha, int flags, double threshold, int min_distance, float *gx, float *gy,
                            size_t gstride, float *resp, size_t rstride, float *corners, size_t cstride, int32_t *locs_yx,
                            int64_t cap, int64_t *count, micv_stream stream);
int micv_harris_corners_host(micv_ctx *ctx, const float *img, int rows, int cols, size_t stride, int sobel_ksize, int win,
                             double sigma, float alpha, int flags, double threshold, int min_distance, float *gx, float *gy,
                             size_t gstride, float *resp, size_t rstride, float *corners, size_t cstride, int32_t *locs_yx,
                             int64_t cap, int64_t *count);

/* sift::getAnglesFromGradients, ps4_cpp/lib/Descriptors.cpp:7-25 (a11). */
int micv_sift_angles_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                         size_t gstride, float *angles, size_t astride, micv_stream stream);
int micv_sift_angles_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                          size_t gstride, float *angles, size_t astride);
/* sift::getKeypoints, Descriptors.cpp:27-47 (a11): kp_xysa is [n][4] = x, y, size, angle. */
int micv_sift_keypoints_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                            size_t gstride, const int32_t *locs_yx, int64_t n, float size,
                            float *kp_xysa, micv_stream stream);
int micv_sift_keypoints_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                             size_t gstride, const int32_t *locs_yx, int64_t n, float size,
                             float *kp_xysa);

/* The SIFT-style descriptor window at those keypoints: the step Solution::siftHelper runs next
 * (ps4_cpp/src/Solution.cpp:166-169, cv::xfeatures2d::SIFT::compute).  OpenCV's SIFT is third-party
 * code outside the reference tree (parity unpinned); this is its published per-keypoint algorithm
 * -- 4 x 4 spatial x 8 orientation bins, window rotated by the keypoint angle, bin width
 * 3 * size / 2 px, Gaussian weight, trilinear distribution, normalise -> clamp 0.2 -> renormalise to
 * 512 -> 8-bit values stored as float -- sampled on the harris::getGradients fields, with the
 * arithmetic fixed as DESIGN.md section 2 states (bit-exact between library and checker).
 * kp_xysa: n x {x, y, size, angle_deg} as micv_sift_keypoints writes them; desc: n rows of 128
 * floats, dstride bytes apart.  Keypoints without a positive finite size get an all-zero row. */
int micv_sift_descriptors_dev(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                              size_t gstride, const float *kp_xysa, int64_t n, float *desc,
                              size_t dstride, micv_stream stream);
int micv_sift_descriptors_host(micv_ctx *ctx, const float *gx, const float *gy, int rows, int cols,
                               size_t gstride, const float *kp_xysa, int64_t n, float *desc,
                               size_t dstride);

/* ------------------------------------------------------------- ps2: stereo --------- */

#define MICV_STEREO_COLS_2R     1 /* window of (2r+1) rows x 2r columns, DisparitySSD.cu:84 */
#define MICV_STEREO_MIN_SSD_5E6 2 /* leave -1 where best SSD >= 5e6, DisparitySSD.cu:16 */
#define MICV_STEREO_SERIAL      4 /* serial::disparitySSD as written (DisparitySSD.cpp:35-61):
                                     per-term round() into an int sum, search clamped to the padded
                                     image, best = (99999999, 0) initially.  SSD only. */
#define MICV_STEREO_ROLLING     8 /* column sums as the CUDA kernels keep them (DisparitySSD.cu:97-138,
                                     DisparityNCorr.cu:117-173): strips of ROWS_PER_THREAD = 40 rows;
                                     a strip's first row sums its 2r+1 terms top -> bottom from 0, each
                                     further row subtracts the term that left the window from the
                                     previous row's sum, then adds the one that entered.  Same result
                                     as the fresh sums on integer-valued images; on general f32 the
                                     two round differently.  Runs a slower, strip-serial kernel. */

/* cuda::disparitySSD / serial::disparitySSD, ps2_cpp/lib/DisparitySSD.cu:143-207 and
 * DisparitySSD.cpp:9-62 (a12).  disp is rows x cols int8 (CV_8SC1), dstride in bytes.
 * flags = 0: CUDA-path semantics with the window corrected to (2r+1)^2 columns (clamp-to-edge
 * addressing, every d in [min,max] tried in ascending order, strict '<').
 * flags = COLS_2R | MIN_SSD_5E6 | ROLLING: the CUDA kernel as written, source order of operations,
 * no contraction (COLS_2R | MIN_SSD_5E6 alone keeps its window and threshold but sums every row's
 * columns afresh -- identical on integer-valued images).  flags = SERIAL: the CPU function exactly as
 * written. */
int micv_disparity_ssd_dev(micv_ctx *ctx, const float *left, const float *right, int rows,
                           int cols, size_t stride, int window_rad, int min_disparity,
                           int max_disparity, int flags, int8_t *disp, size_t dstride,
                           micv_stream stream);
int micv_disparity_ssd_host(micv_ctx *ctx, const float *left, const float *right, int rows,
                            int cols, size_t stride, int window_rad, int min_disparity,
                            int max_disparity, int flags, int8_t *disp, size_t dstride);
/* cuda::disparityNCorr, ps2_cpp/lib/DisparityNCorr.cu:177-251 (a13). */
int micv_disparity_ncorr_dev(micv_ctx *ctx, const float *left, const float *right, int rows,
                             int cols, size_t stride, int window_rad, int min_disparity,
                             int max_disparity, int flags, int8_t *disp, size_t dstride,
                             micv_stream stream);
int micv_disparity_ncorr_host(micv_ctx *ctx, const float *left, const float *right, int rows,
                              int cols, size_t stride, int window_rad, int min_disparity,
                              int max_disparity, int flags, int8_t *disp, size_t dstride);

/* -------------------------------------------------------------- ps1: Hough --------- */

/* Accumulator shape of cuda::houghLinesAccumulate, ps1_cpp/src/Hough.cu:258-263. */
int micv_hough_lines_dims(int rows, int cols, unsigned rho_bin, unsigned theta_bin,
                          int *rho_bins, int *theta_bins);
/* cuda::houghLinesAccumulate, Hough.cu:251-309 (a14): mask u8 -> acc i32 [rho_bins x theta_bins]
 * (dense, zeroed here). */
int micv_hough_lines_dev(micv_ctx *ctx, const uint8_t *mask, int rows, int cols, size_t mstride,
                         unsigned rho_bin, unsigned theta_bin, int32_t *acc, micv_stream stream);
int micv_hough_lines_host(micv_ctx *ctx, const uint8_t *mask, int rows, int cols, size_t mstride,
                          unsigned rho_bin, unsigned theta_bin, int32_t *acc);
/* cuda::houghCirclesAccumulate, Hough.cu:311-364 (a15): acc i32 [rows x cols] (dense, zeroed
 * here -- the reference forgets to, Hough.cu:318). */
int micv_hough_circles_dev(micv_ctx *ctx, const uint8_t *mask, int rows, int cols, size_t mstride,
                           unsigned radius, int32_t *acc, micv_stream stream);
int micv_hough_circles_host(micv_ctx *ctx, const uint8_t *mask, int rows, int cols,
                            size_t mstride, unsigned radius, int32_t *acc);
/* Row-sharded forms: `mask` points at row `row0` of a `rows`-row image and holds
 * `band_rows` rows; votes of those edge points go into a full-size (zeroed here) accumulator.
 * Integer sums of the per-shard accumulators (RCCL all-reduce) equal the unsharded accumulator
 * bit for bit.  The unsharded entry points are these with row0 = 0, band_rows = rows. */
int micv_hough_lines_band_dev(micv_ctx *ctx, const uint8_t *mask, int band_rows, int cols,
                              size_t mstride, int row0, int rows, unsigned rho_bin,
                              unsigned theta_bin, int32_t *acc, micv_stream stream);
int micv_hough_circles_band_dev(micv_ctx *ctx, const uint8_t *mask, int band_rows, int cols,
                                size_t mstride, int row0, int rows, unsigned radius, int32_t *acc,
                                micv_stream stream);
/* cuda::findLocalMaxima, Hough.cu:366-426 (a16): peaks_rc receives up to num_peaks (row,col)
 * pairs ordered by votes descending (stable); *count = number written. */
int micv_hough_peaks_dev(micv_ctx *ctx, const int32_t *acc, int rows, int cols,
                         unsigned num_peaks, int threshold, uint32_t *peaks_rc, int64_t *count,
                         micv_stream stream);
int micv_hough_peaks_host(micv_ctx *ctx, const int32_t *acc, int rows, int cols,
                          unsigned num_peaks, int threshold, uint32_t *peaks_rc, int64_t *count);

/* ------------------------------ ps1: edge front-end ------------ */

/* sol::generateEdge, ps1_cpp/src/Solution.cpp:21-47, on a single-channel 8-bit image: Gaussian
 * blur (cv::cuda::createGaussianFilter, odd size <= 31; size 1 = identity) then Canny with Sobel
 * aperture 3 and the L1 gradient norm; edges = 255 / 0.  The hysteresis pass reads one device
 * flag per round, so this entry point SYNCHRONISES `stream` (OpenCV's CUDA Canny does the same
 * with its queue counter). */
int micv_generate_edge_dev(micv_ctx *ctx, const uint8_t *src, int rows, int cols, size_t stride,
                           int gauss_size, double gauss_sigma, double low_thresh, double high_thresh,
                           uint8_t *edges, size_t estride, micv_stream stream);
int micv_generate_edge_host(micv_ctx *ctx, const uint8_t *src, int rows, int cols, size_t stride,
                            int gauss_size, double gauss_sigma, double low_thresh,
                            double high_thresh, uint8_t *edges, size_t estride);

/* --------------------------- ps4: descriptor matching ----------- */

/* cv::BFMatcher::create()->knnMatch(query, train, matches, 2), ps4_cpp/src/Solution.cpp:172-179:
 * NORM_L2, no cross-check.  query is nq x dim, train nt x dim (f32, strides in bytes, nt >= 2).
 * idx2 [nq][2] = train indices of the nearest and second nearest, dist2 [nq][2] their distances,
 * ordered by (distance, index). */
int micv_bf_knn2_dev(micv_ctx *ctx, const float *query, int nq, size_t qstride, const float *train,
                     int nt, size_t tstride, int dim, int32_t *idx2, float *dist2,
                     micv_stream stream);
int micv_bf_knn2_host(micv_ctx *ctx, const float *query, int nq, size_t qstride, const float *train,
                      int nt, size_t tstride, int dim, int32_t *idx2, float *dist2);
/* The ratio test of Solution.cpp:180-184: keep query q when dist0 < ratio * dist1.  matches_qt
 * [cap][2] = (queryIdx, trainIdx) in query order, distances [cap]; *count (device) = number kept. */
int micv_bf_ratio_filter_dev(micv_ctx *ctx, const int32_t *idx2, const float *dist2, int nq,
                             double ratio, int32_t *matches_qt, float *distances, int64_t cap,
                             int64_t *count, micv_stream stream);
int micv_bf_ratio_filter_host(micv_ctx *ctx, const int32_t *idx2, const float *dist2, int nq,
                              double ratio, int32_t *matches_qt, float *distances, int64_t cap,
                              int64_t *count);

/* ----------------------------------- ps7: motion history ----- */

/* mhi::frameDifference, ps7_cpp/lib/MotionHistory.cpp:26-77, for single-channel CV_8U frames:
 * Gaussian blur with the reference's `const cv::Size& blurSize` (MotionHistory.h:14: blur_w taps
 * along x, blur_h along y, each odd <= 31; blur_sigma > 0 for both directions as
 * cv::cuda::createGaussianFilter(type, -1, ksize, sigma1) does), saturating f2 - f1,
 * AbsThreshold -> {0,1}, 7x7 elliptical morphological open.  diff is rows x cols u8. */
int micv_mhi_frame_difference_dev(micv_ctx *ctx, const uint8_t *f1, const uint8_t *f2, int rows,
                                  int cols, size_t stride, double thresh, int blur_w, int blur_h,
                                  double blur_sigma, uint8_t *diff, size_t dstride,
                                  micv_stream stream);
int micv_mhi_frame_difference_host(micv_ctx *ctx, const uint8_t *f1, const uint8_t *f2, int rows,
                                   int cols, size_t stride, double thresh, int blur_w, int blur_h,
                                   double blur_sigma, uint8_t *diff, size_t dstride);
/* mhi::energyFromHistory, MotionHistory.cpp:98-105: mei = mhi > 0 ? 1 : 0. */
int micv_mhi_energy_dev(micv_ctx *ctx, const uint8_t *mhi, int rows, int cols, size_t sstride,
                        uint8_t *mei, size_t dstride, micv_stream stream);
int micv_mhi_energy_host(micv_ctx *ctx, const uint8_t *mhi, int rows, int cols, size_t sstride,
                         uint8_t *mei, size_t dstride);
/* thresholdDifference / AbsThreshold<uint8_t>, ps7_cpp/lib/MotionHistory.cu:17-48. */
int micv_mhi_threshold_dev(micv_ctx *ctx, const uint8_t *src, int rows, int cols, size_t sstride,
                           double thresh, uint8_t *dst, size_t dstride, micv_stream stream);
int micv_mhi_threshold_host(micv_ctx *ctx, const uint8_t *src, int rows, int cols, size_t sstride,
                            double thresh, uint8_t *dst, size_t dstride);
/* mhi::calcMotionHistory -> motionHistoryKernel, MotionHistory.cu:52-83: in place,
 * history = mask == 1 ? tau : max(history - 1, 0). */
int micv_mhi_update_dev(micv_ctx *ctx, uint8_t *history, size_t hstride, const uint8_t *mask,
                        size_t mstride, int rows, int cols, int tau, micv_stream stream);
int micv_mhi_update_host(micv_ctx *ctx, uint8_t *history, size_t hstride, const uint8_t *mask,
                         size_t mstride, int rows, int cols, int tau);

#ifdef __cplusplus
}
#endif
#endif /* MI_CV_H */
