// lk_fused.hpp -- interface of the LDS-tiled fused Lucas-Kanade level kernel.
#pragma once
#include "common.hpp"

#include <vector>

namespace micv {

enum LkFlowMode {
    LK_FLOW_NONE = 0,    // coarsest level / lk::calcOpticalFlow: base flow is zero, no warp
    LK_FLOW_COARSE = 1,  // base = 2 * pyrUp(coarse flow), computed inside the kernel
    LK_FLOW_FULL = 2,    // base flow given at this level's resolution (odd-size levels)
};

// Pyramid-build work carried by a level launch (r04, MICV_OPT_LK_BUILD_OVERLAP): `blocks` extra workgroups behind the
// launch's tiles copy the pyramid level the NEXT launch of the chain reads (Pyramids.cu:31 -- every level is level 0
// at odd coordinates: L_l(y, x) = L_0(2^l (y + 1) - 1, 2^l (x + 1) - 1)), so the build is not a launch in front of the
// latency-bound coarse levels.  A unit = 32 rows x 64 lanes of level `level` (level 1: 128 columns, a 16-byte load keeps
// two pixels).  Layout = PyrPlan: level l of pair b at pyr + lvl_off[l] * batch + b * rows_l * cols_l; `dst_off` is
// lvl_off[level] * batch.
struct LkBuildJob {
    int blocks = 0;                // extra workgroups (0 = none); units beyond are taken in a stride loop
    int first_block = 0;           // 1-D (chain) launches: the first extra block index
    int units = 0;
    int level = 0;                 // the pyramid level built (>= 1)
    const float *src_a = nullptr, *src_b = nullptr;  // level 0 of the two image sets (prev / next), pairs img_elems apart
    size_t img_elems = 0;
    int sstride = 0, batch = 0, rows = 0, cols = 0;  // level-0 geometry
    float *pyr_a = nullptr, *pyr_b = nullptr;        // the two pyramid arenas (levels >= 1)
    size_t dst_off = 0;
};

struct LkLevelArgs {
    const float *prev, *next;  // level images of pair 0
    int img_stride;            // elements between rows
    int img_xstride = 1;       // elements between pixels: 2^k when pyramid level k is read straight from level 0
    size_t img_pair;           // elements between consecutive pairs
    int rows, cols, batch, win;
    int mode;
    const float *flow_u, *flow_v;  // coarse (mode 1: flow_rows x flow_cols, dense) or full (mode 2)
    int flow_rows, flow_cols;
    size_t flow_pair;
    float *out_u, *out_v;
    int out_stride;
    size_t out_pair;
    int add_base;  // 1: out = base + flow (OpticalFlow.cpp:161-162); 0: out = flow (:100-101)
    // Output rows [row_begin, row_end) only (row-sharded execution); 0 / rows = everything.
    int row_begin = 0, row_end = 0;
    int y_shift = 0;  // set by the launcher: tile rows start at tile_y * TH + y_shift (band launches)
    int narrow = 0;  // MICV_OPT_LK_NARROW_TILES: 256-thread form of the win-15 kernel
    // Tile chains (lk_fused.hip): ctx owns the cached schedules; max_chain = MICV_OPT_LK_CHAIN
    // (0 and 1 = off, n > 1 = longest chain, -1 = the schedule's order with single tiles).  Host side only.
    micv_ctx *ctx = nullptr;
    int max_chain = 0;
    int short_tiles = 0;  // MICV_OPT_LK_SHORT_TILES: 0 = automatic (64x16 tiles under one round), -1 = never
    int tall_tiles = 0;    // MICV_OPT_LK_TALL_TILES: 1 = 64x64 tiles / 1024 threads for big win-15 launches
    int stream_tiles = 0;  // MICV_OPT_LK_STREAM: 1 = persistent grid with staging ahead (lk_level_stream_kernel), 0 = off
    // -DMICV_DIAG builds only (the default build compiles neither in):
    //  * stamps (micv_profile_lk_phases): when non-null, wave 0 of every workgroup adds the
    //    s_memtime ticks it spent in each phase to stamps[phase];
    //  * stop_after (env MICV_LK_STOP=k, read by the diagnostic build only): leave the kernel after
    //    phase k, so PMC counters can be attributed to phases by differencing runs -- garbage output.
    unsigned long long *stamps = nullptr;
    int stop_after = -1;
    LkBuildJob job;
    // Split launch (r05, MICV_OPT_LK_SPLIT; lk_split.hip): when `grad` is set and the launch qualifies, a pre-pass writes
    // Ix, Iy, It of every pixel into padded planes -- pixel (y, x) of pair p at grad[p * grad_pair + (y + grad_pad) *
    // 3 * grad_pitch + plane * grad_pitch + x + grad_pad], the grad_pad cells around the image = BORDER_REFLECT_101
    // copies -- and the base flow into out_u / out_v; the streaming sums kernel does the rest.
    float *grad = nullptr;
    size_t grad_pair = 0;
    int grad_pitch = 0, grad_rows = 0, grad_pad = 0;
    int split = 0;  // MICV_OPT_LK_SPLIT
    int strip = 0;  // MICV_OPT_LK_STRIP: 0 = off, n > 0 = the interior as streamed strips, segments of n blocks of 16 rows
    int pre_base = 0;  // pre-pass tiles store the base flow into out_u / out_v (LkSumsArgs::base == 1)
    // Diagnostic (micv_lk_level_kernel_name): when set, launch_lk_level_fused launches NOTHING and writes the name of the
    // kernel instantiation it would have launched -- the dispatch has one definition, so tools that filter profiler rows
    // by kernel name cannot drift from it
    char *name_out = nullptr;
    size_t name_cap = 0;
    int pre_warp = 0;  // variant A': pre-pass tiles store the warped image (dense plane at grad) and the base flow only
    int base_rmw = 0;  // no-flow mode: out_u / out_v hold a base flow to add (second half of an A' split launch)
};

// Geometry of the padded gradient planes of a rows x cols level for window `win` (host side; floats per pair).
struct LkGradGeom {
    int pad, pitch, rows;
    size_t pair_elems;
};
inline LkGradGeom lk_grad_geom(int rows, int cols, int win) {
    LkGradGeom g;
    g.pad = win / 2;
    g.pitch = ((cols + 63) / 64) * 64 + ((2 * g.pad + 2 + 3) & ~3);  // a strip reads 64 + 2 pad (+ 2) columns, 16-byte chunks
    g.rows = ((rows + 15) & ~15) + 2 * g.pad;                      // blocks of 16 rows + the window's rows below the last
    g.pair_elems = (size_t)g.rows * 3 * g.pitch;
    return g;
}
bool lk_split_supports(int win);
// does launch_lk_level_fused split this launch (given a.grad)?  Host side; also used to size the scratch arena.
bool lk_split_wanted(int rows, int cols, int batch, int win, int mode, int split_opt);

// The streaming sums kernel (lk_split.hip)
struct LkSumsArgs {
    const float *grad;
    size_t grad_pair;
    int gpitch;
    int rows, cols, batch;
    int strips, segs, seg_rows;  // work items = batch * segs * strips
    float *out_u, *out_v;
    int out_stride;
    size_t out_pair;
    // out = base + flow (OpticalFlow.cpp:161-162): 0 = no base, 1 = out_u / out_v hold it (the pre-pass wrote it),
    // 2 = 2 * pyrUp of the coarse flow, recomputed for the kernel's own pixels
    int base;
    const float *flow_u, *flow_v;
    int flow_rows, flow_cols;
    size_t flow_pair;
};
void lk_sums_partition(int rows, int cols, int batch, int slots, int *strips, int *segs, int *seg_rows);
int launch_lk_sums(hipStream_t s, const LkSumsArgs &a, int win);

// units of a build job (host side): level `level` of `batch` pairs, both image sets
inline int lk_build_units(int rows, int cols, int level, int batch) {
    const int r = rows >> level, c = cols >> level;
    return ((c + (level == 1 ? 127 : 63)) / (level == 1 ? 128 : 64)) * ((r + 31) / 32) * 2 * batch;
}

bool lk_fused_supports(int win);
bool lk_fused_supports_direct_levels(int win);  // MICV_OPT_LK_DIRECT_LEVELS has kernels for this window
// Host-only: the (tile x, first tile y, count, pair) entries of the chain / streamed launch schedule,
// 8 per round (one per XCD, count 0 = padding).  Returns the tile height, 0 for windows without one.
int lk_schedule_host(int rows, int cols, int batch, int win, int max_chain, std::vector<int4> *out);
int launch_lk_level_fused(hipStream_t s, const LkLevelArgs &a);

}  // namespace micv
