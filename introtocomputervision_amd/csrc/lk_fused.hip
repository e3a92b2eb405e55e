// lk_fused.hip -- one pyramid level of Lucas-Kanade as ONE LDS-tiled kernel (gfx950).
//
// Per 64x32 output tile (512 threads = 8 wave64 at window 15, 256 at the other windows; 79.5 KB of
// LDS = 2 workgroups per CU; 64x16 tiles for launches of at most one tile per CU):
//   phase 0  stage in LDS by LDS-DMA: the prev tile (+R+1 halo), a window of `next` (tile + halo +
//            8 px margin); the coarse flow block goes in as (u, v) pairs
//   phase 2  "marching" jobs (one column x 4 rows each): pyrUp of the coarse flow, both fields as
//            the lanes of packed f32 ops (row taps, column taps, x2 -- Pyramids.cu:126-127,
//            OpticalFlow.cpp:142), then lk::warp of `next` with the 4 bilinear taps served from the
//            LDS window (global fallback for flows that leave it)        -> warped tile in LDS
//   phase 3  Sobel pairs of prev / warped with a 3-row register window, two adjacent columns per
//            job (packed f32), Ix Iy It -> LDS, the three planes interleaved by row
//   phase 4  five Gaussian-weighted window sums in two sweeps (xx,xy,yy then xt,yt): products
//            formed on the fly, separable (2R+1)-tap row pass (4 outputs per thread from
//            ds_read_b128 windows, every FMA packed with skewed output pairs: row_taps_skew) into
//            XOR-swizzled LDS row buffers, column pass (4 outputs per thread) in registers
//   phase 5  2x2 solve in double, add the base flow, store du / dv
// HBM traffic per level pixel: read prev (4 B) + next (4 B) + coarse flow (2 B), write du, dv
// (8 B).  Nothing else leaves the CU.
//
// Border tiles (tile, halo or pyrUp support touching the image edge) run a slower body for
// phases 0-3 (reflect101 / bounds checks on every neighbour, separate pyrUp row-pass phase,
// global gathers), then fill the out-of-image cells of the gradient planes by reflection so
// that phase 4 is the same straight-line code for every tile.
// Optional: lk_level_chain_kernel walks vertically adjacent tiles in one workgroup and carries the
// last 2R gradient rows over in LDS (MICV_OPT_LK_CHAIN; measured in DESIGN.md section 5).
// All arithmetic goes through lk_device.hpp / the fmaf chains below, identical to the
// generic kernels in lk.hip and to the CPU oracle: every body produces the same bits.
#include "lk_fused.hpp"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "lk_device.hpp"
#include "lk_window.hpp"

namespace micv {


// ---- the tile body ---------------------------------------------------------------------------

// LDS-DMA of a dense block: `nrows` rows of 4 * V4 floats, global row pitch `istride`, into a dense
// LDS image at `dst` (global_load_lds_dwordx4: 16 B per lane straight into LDS, no VGPR round trip,
// no ds_write; every transfer in flight at once).  Element i of the block lives at float4 slot i and
// one wave-instruction covers slots [i0, i0 + 64), every lane busy.  A thread's slots are NT apart:
// (row, float4) of the next one follows from the previous by constant steps and one carry, so only
// the first costs a divide and a 64-bit multiply (a flat index per transfer ran 139 VALU instructions
// per wave on addresses; whole rows per instruction need fewer still but 23 % more, partly empty,
// transfers: measured slower).  `src` points at the block's first element; rows must be 16-B aligned.
template <int NT, int V4, int NROWS>
__device__ __forceinline__ void dma_rows(const float *__restrict__ src, int istride, float *dst, int tid) {
    constexpr int NP = (NROWS * V4 + NT - 1) / NT, A = NT / V4, B = NT % V4;  // slot + NT = (row + A, float4 + B)
    const int lane = tid & 63;
    const int slot0 = __builtin_amdgcn_readfirstlane(tid - lane);  // the wave's first slot of a pass
    int ly = tid / V4, lv = tid - ly * V4;
    size_t goff = (size_t)ly * istride + 4 * lv;
    const size_t step = (size_t)A * istride + 4 * B, carry = (size_t)istride - 4 * V4;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        if (k == NP - 1 ? (ly < NROWS) : true)
            __builtin_amdgcn_global_load_lds((glb_cvoid *)(src + goff), (lds_void *)(dst + 4 * (slot0 + k * NT)), 16,
                                             0, 0);
        if (k + 1 < NP) {
            lv += B;
            const bool c = lv >= V4;
            lv -= c ? V4 : 0;
            ly += A + (c ? 1 : 0);
            goff += step + (c ? carry : 0);
        }
    }
}

// The same block for a border tile: its top-left element is image pixel (gx0, gy0), possibly outside the
// rows x cols image `img`.  16-byte chunks inside the image go by LDS-DMA; chunks outside it are
// zero-filled (ZERO: the `next` window, whose out-of-image cells must read as lk::warp's constant
// border) or left alone (the prev tile: nothing reads them before the reflected ring fill).  Needs
// gx0 and cols to be multiples of 4, so that no chunk straddles the image edge.
template <int NT, int V4, int NROWS, bool ZERO>
__device__ __forceinline__ void dma_rows_clipped(const float *__restrict__ img, int istride, int rows, int cols,
                                                 int gx0, int gy0, float *dst, int tid) {
    constexpr int NP = (NROWS * V4 + NT - 1) / NT, A = NT / V4, B = NT % V4;
    const int lane = tid & 63;
    const int slot0 = __builtin_amdgcn_readfirstlane(tid - lane);
    int ly = tid / V4, lv = tid - ly * V4;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        if (k == NP - 1 ? (ly < NROWS) : true) {
            const int gy = gy0 + ly, gx = gx0 + 4 * lv;
            const bool in = (unsigned)gy < (unsigned)rows && (unsigned)gx < (unsigned)cols;
            if (in)
                __builtin_amdgcn_global_load_lds((glb_cvoid *)(img + (size_t)gy * istride + gx),
                                                 (lds_void *)(dst + 4 * (slot0 + k * NT)), 16, 0, 0);
            else if (ZERO)
                *reinterpret_cast<v4f *>(dst + 4 * (tid + k * NT)) = (v4f){0.f, 0.f, 0.f, 0.f};
        }
        if (k + 1 < NP) {
            lv += B;
            const bool c = lv >= V4;
            lv -= c ? V4 : 0;
            ly += A + (c ? 1 : 0);
        }
    }
}

// ---- pyramid levels read straight from level 0 (r04) ---------------------------------------------------
// Every pyramid level is a decimation of level 0 (Pyramids.cu:31: L_k(y, x) = L_0(2^k y + 2^k - 1, 2^k x + 2^k - 1)),
// so a level kernel can take its images from level 0 with a row stride of 2^k rows and a PIXEL stride of
// xs = 2^k floats instead of from a pyramid some other launch built (that launch was 20 us of a 330 us step
// at the roofline, i.e. only removable).  The staging then is a gather: global_load_lds_dword gives every lane
// its own global address and puts lane l's dword at (wave's LDS base) + 4 l, so one wave-instruction fills 64
// consecutive floats of the dense LDS image from 64 strided pixels -- still no VGPR round trip, no ds_write,
// every transfer in flight at once; four times the transfers of the 16-byte form and no alignment rules.
// Decomposition: one wave-instruction = (one row, one 64-float part of it), so the row is wave-uniform -- its
// address is scalar arithmetic -- and the lane's part, column * xs, never changes: no vector instruction per
// transfer at all (the dense "64 consecutive slots" split of dma_rows needs five per transfer to walk
// (row, column) with a carry; at four times the transfers that was +6 % instructions on level 1, measured
// +3.5 us of 58).  A row of RWC in (64, 128] floats is two parts, the second partly filled.
template <int NT, int RWC, int NROWS>
__device__ __forceinline__ void dma_gather(const float *__restrict__ src, int rstride, int xs, float *dst, int tid) {
    constexpr int PARTS = (RWC + 63) / 64, NWV = NT / 64, RSTEP = NWV / PARTS, NP = (NROWS + RSTEP - 1) / RSTEP;
    static_assert(NWV % PARTS == 0 && RSTEP >= 1, "a wave keeps its part of the row");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = wave % PARTS, col = 64 * part + lane;
    const unsigned voff = (unsigned)(col * xs);
    if (col < RWC) {
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int row = wave / PARTS + RSTEP * k;  // wave-uniform
            if (k == NP - 1 ? row < NROWS : true)
                __builtin_amdgcn_global_load_lds((glb_cvoid *)(src + (size_t)row * rstride + voff),
                                                 (lds_void *)(dst + row * RWC + 64 * part), 4, 0, 0);
        }
    }
}

// The same for a border tile: the block's top-left element is level pixel (gx0, gy0), possibly outside the
// rows x cols level image whose pixel (0, 0) is img[0].  Pixels inside the image go by LDS-DMA, the others are
// zero-filled (ZERO: the `next` window -- lk::warp's constant border) or left alone (the prev tile).
template <int NT, int RWC, int NROWS, bool ZERO>
__device__ __forceinline__ void dma_gather_clipped(const float *__restrict__ img, int rstride, int xs, int rows,
                                                   int cols, int gx0, int gy0, float *dst, int tid) {
    constexpr int PARTS = (RWC + 63) / 64, NWV = NT / 64, RSTEP = NWV / PARTS, NP = (NROWS + RSTEP - 1) / RSTEP;
    static_assert(NWV % PARTS == 0 && RSTEP >= 1, "a wave keeps its part of the row");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = wave % PARTS, col = 64 * part + lane;
    const int gx = gx0 + col;
    const bool col_in = (unsigned)gx < (unsigned)cols;
    const unsigned voff = col_in ? (unsigned)(gx * xs) : 0u;
    if (col < RWC) {
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int row = wave / PARTS + RSTEP * k;  // wave-uniform
            if (k == NP - 1 ? row < NROWS : true) {
                const int gy = gy0 + row;
                const bool row_in = (unsigned)gy < (unsigned)rows;  // wave-uniform
                if (row_in && col_in)
                    __builtin_amdgcn_global_load_lds((glb_cvoid *)(img + (size_t)gy * rstride + voff),
                                                     (lds_void *)(dst + row * RWC + 64 * part), 4, 0, 0);
                else if (ZERO)
                    dst[row * RWC + col] = 0.f;
            }
        }
    }
}

// Streamed tiles (lk_level_stream_kernel): the coarse flow block goes in by LDS-DMA too.  LDS layout:
// [u row | v row] per coarse row, CWP = CW rounded up to whole float4s each, so that pyrUp reads a
// (u, v) pair with one ds_read2_b32; a wave-instruction moves as many consecutive half-rows as fit
// in 64 lanes (16 B per lane, 11 transfers per tile; one dword transfer per half-row, 54 per tile,
// measured the same).  Interior tiles only: every row and column of the block lies inside the coarse
// image (no clamping); rows start at any 4-byte address, which the 16-byte DMA accepts.
template <typename C>
__device__ __forceinline__ void dma_coarse(const LkLevelArgs &a, int pair, int cx0, int cy0, float *Cf, int tid) {
    constexpr int Q = C::CWP / 4, HPI = 64 / Q, HALF_ROWS = 2 * C::CH, NI = (HALF_ROWS + HPI - 1) / HPI;
    constexpr int NWV = C::NT / 64;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hl = lane / Q, q = lane - hl * Q;  // half-row within the instruction, float4 within the half-row
    const int fc = a.flow_cols;
    const float *__restrict__ fu = a.flow_u + pair * a.flow_pair + (size_t)cy0 * fc + cx0 + 4 * q;
    const float *__restrict__ fv = a.flow_v + pair * a.flow_pair + (size_t)cy0 * fc + cx0 + 4 * q;
#pragma unroll
    for (int k = 0; k < (NI + NWV - 1) / NWV; k++) {
        const int it = wave + NWV * k;
        const int hr = it * HPI + hl;  // half-row: coarse row hr / 2, field hr % 2
        if (it < NI && hl < HPI && hr < HALF_ROWS)
            __builtin_amdgcn_global_load_lds((glb_cvoid *)(((hr & 1) ? fv : fu) + (size_t)(hr >> 1) * fc),
                                             (lds_void *)(Cf + it * HPI * C::CWP), 16, 0, 0);
    }
}

#include "lk_strip.hpp"  // the streaming interior body (r05): needs dma_rows / dma_coarse above

// What a streamed tile hands back to the loop that runs it: the entry it took the next ticket for,
// and whether that tile's `next` window and coarse block are already on their way into LDS.
struct LkStreamLink {
    const int4 *sched;  // per-XCD lists, entry t of XCD x at sched[8 * t + x]
    int per_xcd, xcd;
    unsigned *counter;  // this XCD's ticket counter
    int *slot;          // LDS word the ticket travels through
    int4 next;          // .z == 0: none
    bool next_staged;
    // The tile's results stay in registers until the barrier that ends it has passed: a store issued
    // before that barrier would have to be acknowledged by memory before the workgroup may go on (the
    // barrier waits for every outstanding vector-memory operation, stores included).
    float ou[8], ov[8];
};

template <typename C>
__device__ __forceinline__ bool lk_tile_interior(const LkLevelArgs &a, int tile_x, int tile_y) {
    constexpr int E = C::M > 2 ? C::M : 2;
    const int rx0 = tile_x * C::TW - C::H, ry0 = tile_y * C::TH - C::H;
    return rx0 - E >= 0 && rx0 + C::RW + E <= a.cols && ry0 - E >= 0 && ry0 + C::RH + E <= a.rows;
}

// The `next` window and the coarse block of an interior tile, by LDS-DMA into the staging area
// (what phase 0 stages besides the prev tile); no wait, no barrier.
template <typename C>
__device__ __forceinline__ void lk_stage_ahead(const LkLevelArgs &a, float *lds, int tile_x, int tile_y, int pair,
                                               int tid) {
    float *Xs = lds + C::IMG_F;
    const int rx0 = tile_x * C::TW - C::H, ry0 = tile_y * C::TH - C::H;
    const int cx0 = (rx0 - 2 > 0 ? rx0 - 2 : 0) >> 1, cy0 = (ry0 - 2 > 0 ? ry0 - 2 : 0) >> 1;
    dma_coarse<C>(a, pair, cx0, cy0, Xs, tid);
    const float *next = a.next + pair * a.img_pair;
    dma_rows<C::NT, C::NW / 4, C::NH>(next + (size_t)(ry0 - C::M) * a.img_stride + rx0 - C::M, a.img_stride,
                                      Xs + C::CS_F, tid);
}

// CARRY: this tile is not the first of its chain -- gradient rows [0, QC) are already in LDS (the
// tile above left them there), phases 0-3 cover region rows [LYC, RH) only.  MORE: another tile of
// the chain follows -- the last QC gradient rows are moved to the front of the gradient area once
// the row passes are done with them.
// STREAM: the tile runs in lk_level_stream_kernel's loop (its window and coarse block were staged ahead).
// GATHER: the level's images are read from pyramid level 0 with a pixel stride (a.img_xstride; above).
// PRE (r05, lk_split.hip): the tile is a PRE-PASS tile -- phases 0-3 as always, then the gradient cells of its own
// pixels go to the padded planes in HBM and the base flow to u, v; the window sums and the solve run in the streaming
// sums kernel.
template <int R, int MODE, bool INT, int NTV, bool CARRY = false, int THV = 32, bool STREAM = false,
          bool IN_LOOP = CARRY || STREAM, bool GATHER = false, bool PARTIAL_COLS = false, int TWV = 64, bool PRE = false>
__device__ __forceinline__ void lk_tile(const LkLevelArgs &a, const TapsN<2 * R + 1> &g,
                                        float *lds, int tile_x, int tile_y, int pair, bool more = false,
                                        LkStreamLink *link = nullptr) {
    using C = LkCfg<R, NTV, THV, TWV>;
    constexpr int RPT = C::RPT;
    constexpr int TW = C::TW, TH = C::TH, H = C::H, RW = C::RW, RH = C::RH, PS = C::PS;
    static_assert(TWV == 64 || (!CARRY && !STREAM && !GATHER), "32-wide tiles: the plain kernel only");
    constexpr int GW = C::GW, GH = C::GH, GS = C::GS, GP = C::GP, CW = C::CW, NT = C::NT;
    constexpr int M = C::M, NW = C::NW;
    static_assert(!CARRY || (INT && C::FAST && MODE == LK_FLOW_COARSE && C::CHAIN_OK), "carry tiles: interior, coarse flow");
    static_assert(!STREAM || (INT && C::FAST && MODE == LK_FLOW_COARSE && !CARRY && C::RW % 4 == 0 && C::NW % 4 == 0 && !GATHER),
                  "streamed tiles: interior, coarse flow, LDS-DMA staging");
    static_assert(!(STREAM || (INT && C::FAST && MODE == LK_FLOW_COARSE && !CARRY)) || C::CS_F + C::NW * C::NH <= C::X_F,
                  "interior tiles: DMA-staged coarse block + next window fit the gradient area");
    constexpr int LY0 = CARRY ? C::LYC : 0;            // first region row this tile stages / warps
    constexpr int Q0 = CARRY ? C::QC : 0;              // first gradient row this tile computes
    constexpr int CH = CARRY ? C::CHC : C::CH;         // coarse block rows
    constexpr int NH = CARRY ? C::NHC : C::NH;         // `next` window rows
    // interior tiles take the coarse block by LDS-DMA, rows of u and v interleaved (dma_coarse): no
    // registers, no ds_write, no per-element index arithmetic (carry tiles keep the (u, v)-pair layout)
    constexpr bool CDMA = STREAM || (INT && C::FAST && MODE == LK_FLOW_COARSE && !CARRY);
    constexpr int CBF = CDMA ? C::CS_F : (CARRY ? C::CC_F : C::C_F);  // floats of the coarse block (both fields)
    constexpr bool STAGED = INT && C::FAST && MODE != LK_FLOW_NONE;  // next window in LDS (interior tiles: by DMA)
    // DEFER (r04, -DMICV_LK_DEFER_PREV): the prev tile is not read before phase 3, so its DMA is issued LAST and the
    // barrier that opens phase 2 waits only for the coarse block and the `next` window (s_waitcnt vmcnt(n) with n = this
    // wave's prev transfers); the prev tile lands while the march runs and is awaited at the barrier that ends it.
    // hipcc makes every LDS access it can see wait for ALL outstanding LDS-DMA (it has no alias information for them),
    // so phase 2 of such tiles touches LDS through inline asm only: the coarse block (already), the warp's taps and
    // its stores.  Interior, 16-byte-DMA tiles with a coarse flow only.
#ifdef MICV_LK_DEFER_PREV
    constexpr bool DEFER = INT && C::FAST && MODE == LK_FLOW_COARSE && !CARRY && !STREAM && !GATHER;
#else
    constexpr bool DEFER = false;
#endif
    // Border tiles take the marching body too when the image is at least 4 x 4 (one reflection per tap):
    // their `next` window is staged with zeros outside the image (= lk::warp's constant border), their
    // coarse block with replicated edges from a possibly negative origin, and pyrUp's reflected taps differ
    // from the interior pattern in one place only: the first / last image column and row (see march).
    const bool fastb = !INT && C::FAST && MODE != LK_FLOW_NONE && a.rows >= 4 && a.cols >= 4;
    float *P = lds;
    float *Wp = lds + RH * PS;
    float *X = lds + C::IMG_F;
    float *Xs = X + (CARRY ? C::CARRY_F : 0);     // staging area (behind the carried rows)
    // coarse flow block, (u, v) interleaved: both fields go through pyrUp as the two lanes of packed f32 ops
    v2f *Cuv = reinterpret_cast<v2f *>(Xs);
    float *Ru = Xs + CBF, *Rv = Ru + CH * RW;     // border tiles
    float *Nx = Xs + CBF;                         // interior tiles (aliases Ru/Rv)
    float *Gx = X, *Gy = X + GP, *Gt = X + 2 * GP;
    float *rb0 = lds, *rb1 = lds + GH * C::RBS, *rb2 = lds + 2 * GH * C::RBS;  // alias P / Wp

    // Tiles that run inside a loop (chains, the streamed launch): an opaque copy of the thread index
    // keeps the compiler from hoisting every lane-derived address out of that loop (which costs ~70
    // spilled VGPRs).
    int tid_ = threadIdx.x;
    if (IN_LOOP) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int rows = a.rows, cols = a.cols;
    const int x0 = tile_x * TW, y0 = tile_y * TH + a.y_shift;  // y_shift: band launches tile from row_begin
    const int rx0 = x0 - H, ry0 = y0 - H;
    const float *__restrict__ prev = a.prev + pair * a.img_pair;
    const float *__restrict__ next = a.next + pair * a.img_pair;
    const int istride = a.img_stride;
    const int xs = GATHER ? a.img_xstride : 1;  // pixel stride of prev / next (floats)
    const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};  // Pyramids.cu:19
#ifdef MICV_DIAG
    // wave-uniform on purpose: the running stamp then lives in SGPRs, not in a (spilled) VGPR pair
    const bool stamp_wave = a.stamps != nullptr && __builtin_amdgcn_readfirstlane(tid) < 64;
    unsigned long long t_prev = 0;
    if (stamp_wave) t_prev = __builtin_amdgcn_s_memtime();
#define MICV_STAMP(k)                                                        \
    if (stamp_wave) {                                                        \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();       \
        if (tid == 0) atomicAdd(&a.stamps[(k) + (INT ? 0 : 8)], t_now - t_prev); \
        t_prev = t_now;                                                      \
    }                                                                        \
    if (a.stop_after == (k)) return;
#define MICV_STOP(k) if (a.stop_after == (k)) return;
#else
#define MICV_STAMP(k)
#define MICV_STOP(k)
#endif

    // ---- phase 0: stage prev (+ next / its window) and the coarse flow block -----------------
    // Interior tiles issue ALL their global loads into registers first and write LDS afterwards,
    // so the loads overlap each other (a load -> ds_write loop would serialise on every wait).
    int cx0 = 0, cy0 = 0;
    constexpr int NC = CDMA ? 1 : (CH * CW + NT - 1) / NT;
    float rcu[NC], rcv[NC];
    if (CDMA) {
        // streamed tiles: the coarse block and the `next` window were staged ahead (lk_stage_ahead), a
        // barrier ago
        cx0 = (rx0 - 2 > 0 ? rx0 - 2 : 0) >> 1;
        cy0 = (ry0 - 2 > 0 ? ry0 - 2 : 0) >> 1;
        if (!STREAM) dma_coarse<C>(a, pair, cx0, cy0, Xs, tid);
    } else if (MODE == LK_FLOW_COARSE) {
        const float *__restrict__ fu = a.flow_u + pair * a.flow_pair;
        const float *__restrict__ fv = a.flow_v + pair * a.flow_pair;
        const int fr = a.flow_rows, fc = a.flow_cols;
        cx0 = (rx0 - 2 > 0 ? rx0 - 2 : 0) >> 1;
        // carry tiles: the coarse block starts at the rows the tile's own first output row needs
        cy0 = (ry0 + (CARRY ? H : 0) - 2 > 0 ? ry0 + (CARRY ? H : 0) - 2 : 0) >> 1;
        if (fastb) {  // the block keeps its place relative to the region: columns / rows before 0 replicate
            cx0 = (rx0 - 2) >> 1;
            cy0 = (ry0 - 2) >> 1;
        }
#pragma unroll
        for (int k = 0; k < NC; k++) {
            // unconditional loads from clamped (always valid) addresses: no branch, so the
            // compiler keeps every load in flight instead of waiting inside each predicated block
            const int i = tid + k * NT < CH * CW ? tid + k * NT : CH * CW - 1;
            const int cy = i / CW, cx = i - cy * CW;
            const int yy = clampi(cy0 + cy, 0, fr - 1), xx = clampi(cx0 + cx, 0, fc - 1);
            rcu[k] = fu[(size_t)yy * fc + xx];
            rcv[k] = fv[(size_t)yy * fc + xx];
        }
    }
    // (the host launches streamed tiles only when this holds)
    const bool vec_ok = STREAM || GATHER ||
                        (INT && (istride & 3) == 0 && ((a.img_pair & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(a.prev) | reinterpret_cast<uintptr_t>(a.next)) & 15) == 0);
    // 16-byte rows; the last wave of a transfer may be partial and carry tiles start mid-wave (lanes
    // beyond the range are masked off, the wave's LDS base stays uniform)
    constexpr bool DMA_OK = (RW % 4 == 0) && (NW % 4 == 0);
    bool deferred = false;  // (uniform) this tile's prev DMA is still in flight when phase 2 starts
    if (INT && DMA_OK && vec_ok) {
        // LDS-DMA (dma_rows): the LDS images are dense (P: 80-float rows, window: 96-float rows); carry
        // tiles stage region rows [LYC, RH) only
        constexpr int V = RW / 4;
        if constexpr (GATHER) {
            // the `next` window first: phase 2 needs it, the prev tile is not read before phase 3
            if (STAGED)
                dma_gather<NT, NW, NH>(next + (ptrdiff_t)(ry0 - M + LY0) * istride + (ptrdiff_t)(rx0 - M) * xs, istride, xs, Nx, tid);
            dma_gather<NT, RW, RH - LY0>(prev + (size_t)(ry0 + LY0) * istride + (size_t)rx0 * xs, istride, xs, P + LY0 * RW, tid);
            if (MODE == LK_FLOW_NONE)
                dma_gather<NT, RW, RH - LY0>(next + (size_t)(ry0 + LY0) * istride + (size_t)rx0 * xs, istride, xs, Wp + LY0 * RW, tid);
        } else if constexpr (DEFER) {
            dma_rows<NT, NW / 4, NH>(next + (size_t)(ry0 - M + LY0) * istride + rx0 - M, istride, Nx, tid);
            dma_rows<NT, V, RH - LY0>(prev + (size_t)(ry0 + LY0) * istride + rx0, istride, P + LY0 * RW, tid);
            deferred = true;
        } else {
        dma_rows<NT, V, RH - LY0>(prev + (size_t)(ry0 + LY0) * istride + rx0, istride, P + LY0 * RW, tid);
        if (MODE == LK_FLOW_NONE)
            dma_rows<NT, V, RH - LY0>(next + (size_t)(ry0 + LY0) * istride + rx0, istride, Wp + LY0 * RW, tid);
        if (STAGED && !STREAM)
            dma_rows<NT, NW / 4, NH>(next + (size_t)(ry0 - M + LY0) * istride + rx0 - M, istride, Nx, tid);
        }
        if (MODE == LK_FLOW_COARSE && !CDMA) {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int i = tid + k * NT;
                if (i < CH * CW) {
                    Cuv[i] = (v2f){rcu[k], rcv[k]};
                }
            }
        }
    } else {
        if (MODE == LK_FLOW_COARSE && !CDMA) {
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int i = tid + k * NT;
                if (i < CH * CW) {
                    Cuv[i] = (v2f){rcu[k], rcv[k]};
                }
            }
        }
        // Border tiles whose 16-byte chunks cannot straddle the image edge take the DMA as well
        // (in-image chunks only; the `next` window's other chunks are zero-filled)
        const bool clip_dma = fastb && (GATHER || ((cols & 3) == 0 && (istride & 3) == 0 && ((a.img_pair & 3) == 0) &&
                              ((reinterpret_cast<uintptr_t>(a.prev) | reinterpret_cast<uintptr_t>(a.next)) & 15) == 0));
        if (GATHER && clip_dma) {
            dma_gather_clipped<NT, NW, NH, true>(next, istride, xs, rows, cols, rx0 - M, ry0 - M, Nx, tid);
            dma_gather_clipped<NT, RW, RH, false>(prev, istride, xs, rows, cols, rx0, ry0, P, tid);
        } else if (clip_dma) {
            dma_rows_clipped<NT, RW / 4, RH, false>(prev, istride, rows, cols, rx0, ry0, P, tid);
            dma_rows_clipped<NT, NW / 4, NH, true>(next, istride, rows, cols, rx0 - M, ry0 - M, Nx, tid);
        } else
        // Batched, unconditional loads from clamped (always valid) addresses, stored afterwards:
        // every load of the tile is in flight at once.  Cells outside the image receive edge
        // replicas that nothing reads (the gradient phase reflects what it needs).
        {
            constexpr int NB = (RH * RW + NT - 1) / NT;
            float rp[NB], rn[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int i = tid + k * NT < RH * RW ? tid + k * NT : RH * RW - 1;
                const int ly = i / RW, lx = i - ly * RW;
                const int gy = clampi(ry0 + ly, 0, rows - 1), gx = clampi(rx0 + lx, 0, cols - 1);
                rp[k] = prev[(size_t)gy * istride + (size_t)gx * xs];
                if (MODE == LK_FLOW_NONE) rn[k] = next[(size_t)gy * istride + (size_t)gx * xs];
            }
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int i = tid + k * NT;
                if (i < RH * RW) {
                    P[i] = rp[k];
                    if (MODE == LK_FLOW_NONE) Wp[i] = rn[k];
                }
            }
        }
        if (STAGED) {
            for (int i = tid; i < NH * NW; i += NT) {
                const int ly = i / NW, lx = i - ly * NW;
                Nx[i] = next[(ptrdiff_t)(ry0 - M + LY0 + ly) * istride + (ptrdiff_t)(rx0 - M + lx) * xs];
            }
        }
        if (fastb && !clip_dma) {
            // the `next` window of a border tile: zeros outside the image, so that the staged warp's four
            // taps are exactly cv::remap's BORDER_CONSTANT(0) taps.  Batched like the prev tile above;
            // a thread's elements are NT apart = (A rows, B columns) with one carry.
            constexpr int NBN = (NH * NW + NT - 1) / NT, HB = (NBN + 1) / 2, A = NT / NW, B = NT % NW;
            int ly = tid / NW, lx = tid - ly * NW;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                float rn[HB];
                int ly2 = ly, lx2 = lx;
#pragma unroll
                for (int k = 0; k < HB; k++) {
                    const int gy = ry0 - M + ly2, gx = rx0 - M + lx2;
                    const bool ok = (unsigned)gy < (unsigned)rows && (unsigned)gx < (unsigned)cols;
                    const float val = next[(size_t)clampi(gy, 0, rows - 1) * istride + (size_t)clampi(gx, 0, cols - 1) * xs];
                    rn[k] = ok ? val : 0.f;
                    lx2 += B;
                    const bool c = lx2 >= NW;
                    lx2 -= c ? NW : 0;
                    ly2 += A + (c ? 1 : 0);
                }
#pragma unroll
                for (int k = 0; k < HB; k++) {
                    const int i = tid + (half * HB + k) * NT;
                    if (i < NH * NW) Nx[i] = rn[k];
                }
                ly = ly2;
                lx = lx2;
            }
        }
    }
    float base_u[RPT], base_v[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) base_u[j] = base_v[j] = 0.f;
    if (MODE == LK_FLOW_NONE && a.base_rmw) {
        // second half of an A' split launch: `next` is the warped image, u / v hold the base flow (loads issued now, used
        // after the solve)
        const int c_ = threadIdx.x & (TW - 1), r0_ = RPT * (threadIdx.x / TW);
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const int gy = y0 + r0_ + j, gx = x0 + c_;
            if (gy < rows && gx < cols) {
                base_u[j] = a.out_u[pair * a.out_pair + (size_t)gy * a.out_stride + gx];
                base_v[j] = a.out_v[pair * a.out_pair + (size_t)gy * a.out_stride + gx];
            }
        }
    }
    // streamed tiles: what phase 2 reads was staged before the previous barrier, and the prev tile's
    // DMA (issued just now) is only needed by phase 3, a barrier further on
    if constexpr (DEFER) {
        if (deferred) {
            // this wave's prev transfers: every pass but the last is whole; the last one exists for the waves whose
            // first slot lies inside the block (dma_rows: a wave's slots are 64 consecutive ones)
            constexpr int V = RW / 4, TOT = (RH - LY0) * V, NPP = (TOT + NT - 1) / NT;
            static_assert(NPP >= 1 && NPP <= 8, "prev transfers per wave");
            const bool last = __builtin_amdgcn_readfirstlane(tid) + (NPP - 1) * NT < TOT;
            if (last) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPP) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPP - 1) : "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
    } else if (MODE != LK_FLOW_NONE && !STREAM) __syncthreads();
    MICV_STAMP(0)

    if (MODE != LK_FLOW_NONE) {
        // ---- phase 2: base flow at every region pixel, warp `next` --------------------------
        if (C::FAST && (INT || fastb)) {
            // Marching job: column lx, 8 region rows from ly0 (global row even).
            // pyrUp = 2x replicate + [1,4,6,4,1]/16 rows then columns.  Row taps of column gx read
            // coarse columns {m-1,m-1,m,m,m+1} (gx = 2m) or {m-1,m,m,m+1,m+1} (gx = 2m+1); column
            // taps of rows 2m / 2m+1 read the same pattern of coarse rows: six row-pass values per
            // field, built from 3 coarse columns each, serve all 8 rows.
            // origin of the staged window, as scalars (the compiler otherwise re-derives rx0 - M per pixel)
            int nx0s = rx0 - M, ny0s = ry0 - M + LY0;
            asm("" : "+s"(nx0s), "+s"(ny0s));
            auto march = [&](int lx, int ly0, float *bu8, float *bv8) {
                const int gx = rx0 + lx, gy0 = ry0 + ly0;
                // 32 x and 32 y of the job's first pixel, as floats (exact; rows advance by adding 32)
                const float xf32 = 32.f * (float)gx, yf32 = 32.f * (float)gy0;
                // one address register for the job's four stores (rows at immediate offsets)
                typedef __attribute__((address_space(3))) float lds_float;
                lds_float *wrow = (lds_float *)(Wp + ly0 * PS + lx);
                asm("" : "+v"(wrow));
                v2f ruv[RPT / 2 + 2];  // row-pass values of both fields, (u, v) per coarse row
                if (MODE == LK_FLOW_COARSE) {
                    const int cyb = ((gy0 >> 1) - 1) - cy0;
                    const int ccb = ((gx >> 1) - 1) - cx0;
                    // taps 1 and 3 read coarse column (0 or 1) + parity: two more LDS reads off a
                    // second base register instead of four v_cndmask per coarse row
                    const int odd = gx & 1;
                    // Border tiles: BORDER_REFLECT_101 on the fine grid (Pyramids.cu:126-127) against the
                    // replicated edges of the coarse block.  Writing out the five taps of fine columns 0, 1,
                    // cols-2, cols-1 (cols = 2 fc): replication already gives the reflected pattern except
                    // tap 0 of column 0 (coarse 1, not 0) and tap 4 of column cols-1 (coarse fc-2, not fc-1);
                    // the same holds for rows.  Those are c2 / c0 of this job's own three columns.
                    const bool first_col = !INT && gx == 0, last_col = !INT && gx == cols - 1;
                    constexpr int NR = RPT / 2 + 2;
                    v2f cc[CDMA ? 5 * NR : 1];
                    if constexpr (CDMA)  // rows of u and v interleaved (dma_coarse): a pair = one ds_read2_b32
                        coarse_block_load<C>(cc, Xs + cyb * (2 * C::CWP) + ccb, odd, std::make_integer_sequence<int, NR>{});
#pragma unroll
                    for (int i = 0; i < NR; i++) {
                        v2f c0, c1, c2, ca, cb;
                        if constexpr (CDMA) {
                            c0 = cc[5 * i], c1 = cc[5 * i + 1], c2 = cc[5 * i + 2], ca = cc[5 * i + 3], cb = cc[5 * i + 4];
                        } else {
                            const v2f *c = Cuv + (cyb + i) * CW + ccb;
                            c0 = c[0];
                            c1 = c[1];
                            c2 = c[2];
                            ca = c[odd];
                            cb = c[1 + odd];
                        }
                        const v2f t0 = first_col ? c2 : c0, t4 = last_col ? c0 : c2;
                        v2f t = t0 * (v2f){g5[0], g5[0]};
                        t = __builtin_elementwise_fma(ca, (v2f){g5[1], g5[1]}, t);
                        t = __builtin_elementwise_fma(c1, (v2f){g5[2], g5[2]}, t);
                        t = __builtin_elementwise_fma(cb, (v2f){g5[3], g5[3]}, t);
                        ruv[i] = __builtin_elementwise_fma(t4, (v2f){g5[4], g5[4]}, t);
                    }
                }
                // FULL mode reads the (already expanded + resized) base flow from global memory:
                // issue all the rows' loads up front, the per-row scheduling fence below would
                // otherwise serialise them behind each row's warp
                float fu_[RPT], fv_[RPT];
                if (MODE == LK_FLOW_FULL) {
#pragma unroll
                    for (int j = 0; j < RPT; j++) {
                        // border tiles: region pixels outside the image read a valid address (their flow is never used)
                        const size_t fo = pair * a.flow_pair + (size_t)(INT ? gy0 + j : clampi(gy0 + j, 0, rows - 1)) * a.flow_cols +
                                          (INT ? gx : clampi(gx, 0, cols - 1));
                        fu_[j] = a.flow_u[fo];
                        fv_[j] = a.flow_v[fo];
                    }
                }
#pragma unroll
                for (int p = 0; p < RPT / 2; p++) {
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        const int j = 2 * p + o;
                        float bu, bv;
                        v2f half_uv;  // MODE COARSE: pyrUp's value before the "* 2" of OpticalFlow.cpp:142,144
                        if (MODE == LK_FLOW_COARSE) {
                            const int i1 = o ? p + 1 : p, i3 = o ? p + 2 : p + 1;
                            // border tiles: tap 0 of image row 0 and tap 4 of image row rows-1 (see above).
                            // The two candidates go through an opaque copy: LLVM otherwise rewrites the
                            // select of two array elements as an indexed read of ruv[] = 16 v_cndmask each.
                            v2f ra = ruv[p], rc = ruv[p + 2];
                            if (!INT) asm("" : "+v"(ra), "+v"(rc));
                            const v2f r0 = (!INT && gy0 + j == 0) ? rc : ra;
                            const v2f r4 = (!INT && gy0 + j == rows - 1) ? ra : rc;
                            v2f auv = r0 * (v2f){g5[0], g5[0]};
                            auv = __builtin_elementwise_fma(ruv[i1], (v2f){g5[1], g5[1]}, auv);
                            auv = __builtin_elementwise_fma(ruv[p + 1], (v2f){g5[2], g5[2]}, auv);
                            auv = __builtin_elementwise_fma(ruv[i3], (v2f){g5[3], g5[3]}, auv);
                            auv = __builtin_elementwise_fma(r4, (v2f){g5[4], g5[4]}, auv);
                            half_uv = auv;
                            auv = auv * (v2f){2.f, 2.f};  // OpticalFlow.cpp:142,144
                            bu = auv.x;
                            bv = auv.y;
                        } else {
                            bu = fu_[j];
                            bv = fv_[j];
                            half_uv = (v2f){bu, bv};
                        }
                        bu8[j] = bu;
                        bv8[j] = bv;
                        // warp right away: the flow pair's live range ends here.  Carry tiles need
                        // the warped image from region row LYC on (the base flow of every own row)
                        if constexpr (DEFER) {
                            // (both forms compiled: `deferred` is uniform, and false for unaligned images)
                            if (deferred) {
                                const float wv = warp_sample_staged<NW, NH, 64, true>(
                                    Nx, nx0s, ny0s, next, rows, cols, istride, (v2f){xf32, yf32 + 32.f * (float)j}, half_uv, xs);
                                asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(wrow), "v"(wv), "n"(4 * j * PS) : "memory");
                            } else {
                                wrow[j * PS] = warp_sample_staged<NW, NH, 64>(
                                    Nx, nx0s, ny0s, next, rows, cols, istride, (v2f){xf32, yf32 + 32.f * (float)j}, half_uv, xs);
                            }
                        } else if (!CARRY || ly0 + j >= LY0)
                            wrow[j * PS] = warp_sample_staged<NW, NH, MODE == LK_FLOW_COARSE ? 64 : 32>(
                                Nx, nx0s, ny0s, next, rows, cols, istride, (v2f){xf32, yf32 + 32.f * (float)j}, half_uv, xs);
                        // 512-thread tiles run at a 128-VGPR budget: keep the rows from interleaving
                        if (NT >= 512) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            march(H + (tid & (TW - 1)), H + RPT * (tid / TW), base_u, base_v);  // own outputs
            // halo jobs: top / bottom bands (H/RPT segments x RW columns each), left / right bands
            // (H columns each x TH/RPT segments)
            // carry tiles: no top band, and of the side bands only the segments that reach row LYC
            constexpr int BS = H / RPT, NB = CARRY ? BS : 2 * BS;
            constexpr int SEG0 = CARRY ? (LY0 - H) / RPT : 0, NJ = NB * RW + 2 * H * (TH / RPT - SEG0);
            static_assert(!CARRY || (LY0 >= H && LY0 < H + TH), "carry rows start inside the tile's own rows");
            for (int n = tid; n < (PRE && a.pre_warp ? 0 : NJ); n += NT) {  // (warp-only pre-pass tiles: own pixels only)
                int lx, ly0;
                if (n < NB * RW) {
                    const int band = n / RW + (CARRY ? BS : 0);
                    lx = n - (band - (CARRY ? BS : 0)) * RW;
                    ly0 = band < BS ? band * RPT : H + TH + (band - BS) * RPT;
                } else {
                    const int m = n - NB * RW;
                    const int seg = m / (2 * H) + SEG0, cc = m - (seg - SEG0) * (2 * H);
                    lx = cc < H ? cc : TW + cc;
                    ly0 = H + RPT * seg;
                }
                float tu[RPT], tv[RPT];
                march(lx, ly0, tu, tv);
            }
        } else {
            if (MODE == LK_FLOW_COARSE) {
                // ---- phase 1 (border tiles): pyrUp row pass, once per coarse row -------------
                const int fr = a.flow_rows;
                for (int i = tid; i < CH * RW; i += NT) {
                    const int cy = i / RW, lx = i - cy * RW;
                    const int gx = rx0 + lx;
                    if ((INT || (unsigned)gx < (unsigned)cols) && cy0 + cy < fr) {
                        float au = 0.f, av = 0.f;
#pragma unroll
                        for (int k = 0; k < 5; k++) {
                            const int sc = ((INT ? gx + k - 2 : reflect101(gx + k - 2, cols)) >> 1) - cx0;
                            const v2f cuv = Cuv[cy * CW + sc];
                            au = fmaf(cuv.x, g5[k], au);
                            av = fmaf(cuv.y, g5[k], av);
                        }
                        Ru[i] = au;
                        Rv[i] = av;
                    }
                }
                __syncthreads();
                MICV_STAMP(1)
            }
            auto do_px = [&](int ly, int lx, float &bu, float &bv) {
                const int gy = ry0 + ly, gx = rx0 + lx;
                if (!INT && ((unsigned)gy >= (unsigned)rows || (unsigned)gx >= (unsigned)cols)) return;
                if (MODE == LK_FLOW_COARSE) {
                    float au = 0.f, av = 0.f;
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const int rr = ((INT ? gy + k - 2 : reflect101(gy + k - 2, rows)) >> 1) - cy0;
                        au = fmaf(Ru[rr * RW + lx], g5[k], au);
                        av = fmaf(Rv[rr * RW + lx], g5[k], av);
                    }
                    bu = au * 2.f;  // OpticalFlow.cpp:142,144
                    bv = av * 2.f;
                } else {
                    bu = a.flow_u[pair * a.flow_pair + (size_t)gy * a.flow_cols + gx];
                    bv = a.flow_v[pair * a.flow_pair + (size_t)gy * a.flow_cols + gx];
                }
                Wp[ly * PS + lx] = warp_sample(next, rows, cols, istride, gx, gy, bu, bv, xs);
            };
            {
                const int c = tid & (TW - 1), grp = tid / TW;
#pragma unroll
                for (int j = 0; j < RPT; j++) do_px(H + RPT * grp + j, H + c, base_u[j], base_v[j]);
            }
            constexpr int NHALO = 2 * H * RW + TH * 2 * H;
            for (int n = tid; n < NHALO; n += NT) {
                int ly, lx;
                if (n < 2 * H * RW) {
                    ly = n / RW;
                    lx = n - ly * RW;
                    if (ly >= H) ly += TH;
                } else {
                    const int m = n - 2 * H * RW;
                    const int rr = m / (2 * H), cc = m - rr * (2 * H);
                    ly = H + rr;
                    lx = cc < H ? cc : TW + cc;
                }
                float du_, dv_;
                do_px(ly, lx, du_, dv_);
            }
        }
    }
    if constexpr (DEFER) {
        if (deferred) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the asm stores above, the prev DMA
    }
    __syncthreads();
    MICV_STAMP(2)
    if constexpr (PRE) {
        if (a.pre_warp) {
            // ---- warp-only pre-pass (variant A' of the split, experiment): the warped image of the tile's own pixels
            // -> a dense plane at a.grad (cols floats per row), the base flow -> u, v
            const int c = tid & (TW - 1), r0 = RPT * (tid / TW);
            const int gx = x0 + c;
            float *__restrict__ wb = a.grad + pair * a.grad_pair;
            float *__restrict__ ou = a.out_u + pair * a.out_pair;
            float *__restrict__ ov = a.out_v + pair * a.out_pair;
            if (INT || gx < cols) {
#pragma unroll
                for (int j = 0; j < RPT; j++) {
                    const int gy = y0 + r0 + j;
                    if (INT || gy < rows) {
                        wb[(size_t)gy * cols + gx] = Wp[(H + r0 + j) * PS + H + c];
                        ou[(size_t)gy * a.out_stride + gx] = base_u[j];
                        ov[(size_t)gy * a.out_stride + gx] = base_v[j];
                    }
                }
            }
            return;
        }
    }
    // streamed tiles: one thread takes the ticket of the tile that follows; the answer is needed after
    // sweep B's row pass and travels through LDS two barriers before that
    int ticket = 0;
    if (STREAM && tid == 0) ticket = (int)atomicAdd(link->counter, 1u);

    // ---- phase 3: gradients --------------------------------------------------------------
    {
        const float s1 = 1.f / 9.f, s2 = 2.f * s1;  // OpticalFlow.cpp:19
        if (!INT) {
            // BORDER_REFLECT_101 of the Sobel source: fill the one-pixel ring just outside the
            // image (inside this region) of prev and of the warped image with the reflected
            // values, so the marching code below serves border tiles too.  Gradient cells outside
            // the image come out as garbage here and are overwritten by the reflected fill.
            // The ring is at most two region rows (gy = -1, gy = rows) and two region columns
            // (gx = -1, gx = cols): walk those 2 RW + 2 RH cells, not the whole region.
            for (int n = tid; n < 2 * RW + 2 * RH; n += NT) {
                int ly, lx;
                if (n < 2 * RW) {
                    ly = (n < RW ? -1 : rows) - ry0;
                    lx = n < RW ? n : n - RW;
                } else {
                    const int m = n - 2 * RW;
                    lx = (m < RH ? -1 : cols) - rx0;
                    ly = m < RH ? m : m - RH;
                }
                if ((unsigned)ly >= (unsigned)RH || (unsigned)lx >= (unsigned)RW) continue;
                const int gy = ry0 + ly, gx = rx0 + lx;
                if (gy >= -1 && gy <= rows && gx >= -1 && gx <= cols) {
                    const int sy = reflect101(gy, rows) - ry0, sx = reflect101(gx, cols) - rx0;
                    if ((unsigned)sy < (unsigned)RH && (unsigned)sx < (unsigned)RW) {
                        P[ly * PS + lx] = P[sy * PS + sx];
                        Wp[ly * PS + lx] = Wp[sy * PS + sx];
                    }
                }
            }
            __syncthreads();
        }
        {
            // Marching job: gradient column qx, a run of gradient rows.  Row pass of the Sobel
            // pair (tx = right - left, ty = [s,2s,s]) is computed once per image row and kept in
            // a 3-row register window; the column pass finishes one output per step.
            // carry tiles compute 2R fewer rows: shorter segments keep every thread's job short
            // 512-thread tiles: 4-row segments put 468 jobs on the 512 threads; 8-row segments were 234 jobs
            // on four of the eight waves (9 % fewer instructions, twice the critical path: measured
            // 226 -> 221 us for the level-0 launch; 6 rows: 224)
            constexpr int SEG = CARRY ? 6 : (TH == 16 ? 5 : (NT >= 512 ? 4 : 16)), NSEG = (GH - Q0 + SEG - 1) / SEG;
            // A job covers TWO adjacent gradient columns: the pair rides the two lanes of packed f32
            // instructions (v_pk_add / v_pk_mul / v_pk_fma), so the Sobel arithmetic of both images costs
            // half the VALU instructions per pixel, and every LDS access is a two-element one.
            static_assert(GW % 2 == 0, "gradient columns in pairs");
            constexpr int GW2 = GW / 2;
            const v2f s1v = {s1, s1}, s2v = {s2, s2}, halfv = {0.5f, 0.5f};
            for (int n = tid; n < GW2 * NSEG; n += NT) {
                const int seg = n / GW2, qx = 2 * (n - seg * GW2);
                const int q0 = Q0 + seg * SEG, q1 = q0 + SEG < GH ? q0 + SEG : GH;
                const int lx = qx + (H - R);
                v2f ptx[3], pty[3], wtx[3], wty[3], pcc[3], wcc[3];
                auto rowpass = [&](int ly, int slot) {
                    const float *pr = P + ly * PS + lx, *wr = Wp + ly * PS + lx;
                    v2f pa, pb, pc, wa, wb, wc;
                    if constexpr (((H - R) & 1) == 1 && (PS & 1) == 0) {
                        // lx is odd (qx even, H - R odd): (pr[-1], pr[0]) and (pr[1], pr[2]) are 8-byte
                        // aligned pairs -> two ds_read_b64 per image row, lanes 8 bytes apart = one
                        // conflict-free sweep of the bank row.  As scalars the compiler emits ds_read2_b32
                        // at a 2-float lane stride: 2-way conflicts in its 32-bank mode (r02: 26 % of the
                        // phase's LDS cycles were conflict cycles).
                        const v2f p01 = *reinterpret_cast<const v2f *>(pr - 1), p23 = *reinterpret_cast<const v2f *>(pr + 1);
                        const v2f w01 = *reinterpret_cast<const v2f *>(wr - 1), w23 = *reinterpret_cast<const v2f *>(wr + 1);
                        pa = p01; pb = (v2f){p01.y, p23.x}; pc = p23;
                        wa = w01; wb = (v2f){w01.y, w23.x}; wc = w23;
                    } else {
                        pa = (v2f){pr[-1], pr[0]}; pb = (v2f){pr[0], pr[1]}; pc = (v2f){pr[1], pr[2]};
                        wa = (v2f){wr[-1], wr[0]}; wb = (v2f){wr[0], wr[1]}; wc = (v2f){wr[1], wr[2]};
                    }
                    ptx[slot] = pc - pa;
                    pty[slot] = __builtin_elementwise_fma(pc, s1v, __builtin_elementwise_fma(pb, s2v, pa * s1v));
                    wtx[slot] = wc - wa;
                    wty[slot] = __builtin_elementwise_fma(wc, s1v, __builtin_elementwise_fma(wb, s2v, wa * s1v));
                    pcc[slot] = pb;
                    wcc[slot] = wb;
                };
                const int lyb = q0 + (H - R);  // image row of gradient row q0
                rowpass(lyb - 1, 0);
                rowpass(lyb, 1);
                // three outputs per trip so the window slots stay compile-time constants; short
                // segments are unrolled altogether
                constexpr int TRIP = SEG <= 6 ? SEG : 3;
#pragma unroll 1
                for (int qy = q0; qy < q1; qy += TRIP) {
#pragma unroll
                    for (int t = 0; t < TRIP; t++) {
                        if (qy + t < q1) {
                            const int s_new = (t + 2) % 3, s_top = t % 3, s_mid = (t + 1) % 3;
                            rowpass(qy + t + (H - R) + 1, s_new);
                            const v2f pgx = __builtin_elementwise_fma(
                                ptx[s_new], s1v, __builtin_elementwise_fma(ptx[s_mid], s2v, ptx[s_top] * s1v));
                            const v2f pgy = pty[s_new] - pty[s_top];
                            const v2f ngx = __builtin_elementwise_fma(
                                wtx[s_new], s1v, __builtin_elementwise_fma(wtx[s_mid], s2v, wtx[s_top] * s1v));
                            const v2f ngy = wty[s_new] - wty[s_top];
                            // OpticalFlow.cpp:62-64: avg2(next, prev) = next * .5f + prev * .5f, It = next - prev
                            *reinterpret_cast<v2f *>(Gx + (qy + t) * GS + qx) = ngx * halfv + pgx * halfv;
                            *reinterpret_cast<v2f *>(Gy + (qy + t) * GS + qx) = ngy * halfv + pgy * halfv;
                            *reinterpret_cast<v2f *>(Gt + (qy + t) * GS + qx) = wcc[s_mid] - pcc[s_mid];
                        }
                    }
                }
            }
        }
        if (!INT) {
            // BORDER_REFLECT_101 of the window sums (cv::GaussianBlur default border) = the
            // product fields at reflected positions: fill the out-of-image cells once, so the
            // window sums below run the same straight-line code as interior tiles.  Cells whose
            // source lies outside this tile feed only outputs that are outside the image.
            __syncthreads();
            // Walk only the cells outside the image: whole gradient rows above / below it, then the
            // left / right columns of the rows in between.
            auto fill = [&](int qy, int qx) {
                const int gy = y0 - R + qy, gx = x0 - R + qx;
                const int sy = reflect101(gy, rows) - (y0 - R), sx = reflect101(gx, cols) - (x0 - R);
                const bool ok = (unsigned)sy < (unsigned)GH && (unsigned)sx < (unsigned)GW;
                Gx[qy * GS + qx] = ok ? Gx[sy * GS + sx] : 0.f;
                Gy[qy * GS + qx] = ok ? Gy[sy * GS + sx] : 0.f;
                Gt[qy * GS + qx] = ok ? Gt[sy * GS + sx] : 0.f;
            };
            const int rt = clampi(-(y0 - R), 0, GH), rb = clampi(rows - (y0 - R), rt, GH);
            const int cl = clampi(-(x0 - R), 0, GW), cr = clampi(cols - (x0 - R), cl, GW);
            const int n_full = (rt + GH - rb) * GW, side = cl + GW - cr;
            for (int i = tid; i < n_full; i += NT) {
                const int r = i / GW, qx = i - r * GW;
                fill(r < rt ? r : rb + (r - rt), qx);
            }
            if (side > 0) {
                for (int i = tid; i < (rb - rt) * 16; i += NT) {
                    const int qy = rt + (i >> 4);
                    for (int cc = i & 15; cc < side; cc += 16) fill(qy, cc < cl ? cc : cr + (cc - cl));
                }
            }
        }
    }
    __syncthreads();
    MICV_STAMP(3)

    if constexpr (PRE) {
        // ---- pre-pass: the gradient cells of the tile's own pixels -> padded planes; base flow -> u, v -------------
        static_assert(MODE == LK_FLOW_COARSE && !CARRY && !STREAM && TWV == 64, "pre-pass tiles: coarse flow, plain launch");
        const int c = tid & (TW - 1), r0 = RPT * (tid / TW);
        const int gx = x0 + c, pad = a.grad_pad, gp = a.grad_pitch;
        const size_t rowp = 3 * (size_t)gp;
        float *__restrict__ gb = a.grad + pair * a.grad_pair;
        float *__restrict__ ou = a.out_u + pair * a.out_pair;
        float *__restrict__ ov = a.out_v + pair * a.out_pair;
        if (INT || gx < cols) {
            // BORDER_REFLECT_101 of the window sums (OpticalFlow.cpp:73-77): a cell up to `pad` outside the image is the
            // cell at the reflected position -- the pixel that IS that position writes it (border tiles only; rows and
            // cols are at least 2 pad + 2, so a pixel has at most one mirror image per axis)
            int px2 = -1;
            if (!INT) {
                if (gx >= 1 && gx <= pad) px2 = pad - gx;
                else if (gx >= cols - 1 - pad && gx <= cols - 2) px2 = pad + 2 * (cols - 1) - gx;
            }
#pragma unroll
            for (int j = 0; j < RPT; j++) {
                const int gy = y0 + r0 + j;
                if (INT || gy < rows) {
                    const int q = r0 + j + R, qx = c + R;
                    const float ix = Gx[q * GS + qx], iy = Gy[q * GS + qx], it = Gt[q * GS + qx];
                    auto put = [&](int py, int px) {
                        float *d = gb + (size_t)py * rowp + px;
                        d[0] = ix;
                        d[gp] = iy;
                        d[2 * gp] = it;
                    };
                    put(gy + pad, gx + pad);
                    if (!INT) {
                        int py2 = -1;
                        if (gy >= 1 && gy <= pad) py2 = pad - gy;
                        else if (gy >= rows - 1 - pad && gy <= rows - 2) py2 = pad + 2 * (rows - 1) - gy;
                        if (py2 >= 0) put(py2, gx + pad);
                        if (px2 >= 0) put(gy + pad, px2);
                        if (py2 >= 0 && px2 >= 0) put(py2, px2);
                    }
                    if (a.pre_base) {
                        ou[(size_t)gy * a.out_stride + gx] = base_u[j];
                        ov[(size_t)gy * a.out_stride + gx] = base_v[j];
                    }
                }
            }
        }
        return;
    }

    // ---- phase 4: five window sums, two sweeps ----------------------------------------------
    const int c = tid & (TW - 1), r0 = RPT * (tid / TW);
    float Sxx[RPT], Sxy[RPT], Syy[RPT], Sxt[RPT], Syt[RPT];
    constexpr int RPW = 64 / (TW / 4);     // gradient rows a wave's 64 row-pass jobs cover (4 outputs per job)
    constexpr int RPI = RPW * (NT / 64);  // gradient rows one row-pass iteration covers
    constexpr int SWEEP_UNROLL = NT >= 512 ? 1 : 4;  // 512 threads: rolled sweeps keep the 128-VGPR budget
    // 64x16 tiles: one trip per sweep.  As straight-line code the scheduler overlaps the sweeps with the
    // column passes around them and spills; a trip count the compiler cannot see (a.batch is never
    // negative) keeps each sweep a rolled loop = its own scheduling region, as at 64x32.
    constexpr int SWEEP_TRIPS = (GH + RPI - 1) / RPI;
    const int sweep_trips = SWEEP_TRIPS == 1 ? 1 + (a.batch < 0 ? 1 : 0) : SWEEP_TRIPS;
    {
        const int lane = tid & 63, wave = tid >> 6;
        // lane -> (row within the wave's rows, group of four columns).  64-wide tiles: 4 rows x 16 groups, row in the
        // low bits.  32-wide tiles: 8 rows x 8 groups with a QUARTER wave = 4 rows x 4 groups (bits g1 g0 r1 r0 | g2 r2
        // from the top): the 16 ds_read_b128 / ds_write_b128 of a quarter then fall on 16 different 16-byte slots
        // (plane rows are 36 = 4 (mod 16) slots apart; row buffers: rb_off32).
        const int grp = TW == 64 ? lane >> 2 : ((lane & 3) | ((lane >> 2) & 4)), c0 = 4 * grp;
        const int lrow = TW == 64 ? (lane & 3) : (((lane >> 2) & 3) | ((lane >> 3) & 4));
        // sweep A: Ix^2, Ix*Iy, Iy^2  (windows of Ix, Iy read once)
#pragma unroll SWEEP_UNROLL
        for (int it = 0; it < sweep_trips; it++) {
            const int qy = it * RPI + wave * RPW + lrow;
            if (qy < GH) {
                v2f wx[2 * C::WV], wy[2 * C::WV];
                load_window_pairs<C>(Gx, qy, c0, wx);
                load_window_pairs<C>(Gy, qy, c0, wy);
                const int o = TW == 64 ? rb_off(qy, grp) : rb_off32(qy, grp);
                row_taps_skew<C>(wx, wx, g, rb0 + o);
                // 64x16 tiles: the sweep is a single (peeled) trip; keep its three products from
                // interleaving so the body stays inside the 128-VGPR budget
                if (TH == 16) __builtin_amdgcn_sched_barrier(0);
                row_taps_skew<C>(wx, wy, g, rb1 + o);
                if (TH == 16) __builtin_amdgcn_sched_barrier(0);
                row_taps_skew<C>(wy, wy, g, rb2 + o);
            }
        }
        if (STREAM && tid == 0) *link->slot = ticket;
        __syncthreads();
        MICV_STOP(41)
        int cls[TW == 64 ? 4 : 8];  // the column pass's address registers (all five fields)
        if constexpr (TW == 64) {
            col_bases<C>(rb0, c, r0, cls);
            col_pass<C, 0, PARTIAL_COLS>(cls, Sxx, g);
            col_pass<C, 1, PARTIAL_COLS>(cls, Sxy, g);
            col_pass<C, 2, PARTIAL_COLS>(cls, Syy, g);
        } else {
            col_bases32<C>(rb0, c, r0, cls);
            col_pass32<C, 0, PARTIAL_COLS>(cls, Sxx, g);
            col_pass32<C, 1, PARTIAL_COLS>(cls, Sxy, g);
            col_pass32<C, 2, PARTIAL_COLS>(cls, Syy, g);
        }
        // 64x16 tiles: finish the three column chains here instead of letting them sink below the
        // barrier into sweep B (their 48 loaded values would be spilled there)
        if (TH == 16) __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        MICV_STOP(42)
        // sweep B: Ix*It, Iy*It
#pragma unroll SWEEP_UNROLL
        for (int it = 0; it < sweep_trips; it++) {
            const int qy = it * RPI + wave * RPW + lrow;
            if (qy < GH) {
                v2f wx[2 * C::WV], wt[2 * C::WV];
                load_window_pairs<C>(Gx, qy, c0, wx);
                load_window_pairs<C>(Gt, qy, c0, wt);
                const int o = TW == 64 ? rb_off(qy, grp) : rb_off32(qy, grp);
                row_taps_skew<C>(wx, wt, g, rb0 + o);
                // 128-VGPR budget at 512 threads: the Iy window reuses the Ix window's registers
                if (NT >= 512) __builtin_amdgcn_sched_barrier(0);
                load_window_pairs<C>(Gy, qy, c0, wx);
                row_taps_skew<C>(wx, wt, g, rb1 + o);
            }
        }
        __syncthreads();
        MICV_STOP(43)
        if constexpr (STREAM) {
            // the gradient planes are dead: the next tile's `next` window and coarse block go in over
            // them while this tile runs its last column passes and the solve
            const int t = __builtin_amdgcn_readfirstlane(*link->slot);
            link->next = make_int4(0, 0, 0, 0);
            link->next_staged = false;
            if (t < link->per_xcd) {
                const int4 e2 = link->sched[8 * t + link->xcd];
                link->next = e2;
                if (e2.z > 0 && lk_tile_interior<C>(a, e2.x, e2.y)) {
                    lk_stage_ahead<C>(a, lds, e2.x, e2.y, e2.w, tid);
                    link->next_staged = true;
                }
            }
        }
        if (INT && C::CHAIN_OK && MODE == LK_FLOW_COARSE && more) {
            // the row passes are done with the gradient planes: hand the last QC rows (all three
            // planes, one contiguous block) to the next tile of the chain as its first QC rows
            const v4f *src = reinterpret_cast<const v4f *>(X + TH * GS);
            v4f *dst = reinterpret_cast<v4f *>(X);
            for (int i = tid; i < C::CARRY_F / 4; i += NT) dst[i] = src[i];
        }
        if constexpr (TW == 64) {
            col_pass<C, 0, PARTIAL_COLS>(cls, Sxt, g);
            col_pass<C, 1, PARTIAL_COLS>(cls, Syt, g);
        } else {
            col_pass32<C, 0, PARTIAL_COLS>(cls, Sxt, g);
            col_pass32<C, 1, PARTIAL_COLS>(cls, Syt, g);
        }
    }
    MICV_STAMP(4)

    // ---- phase 5: solve + store ----------------------------------------------------------------
    float *__restrict__ ou = a.out_u + pair * a.out_pair;
    float *__restrict__ ov = a.out_v + pair * a.out_pair;
    const int gx = x0 + c;
    if (INT || gx < cols) {
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const int gy = y0 + r0 + j;
            if (gy >= a.row_begin && gy < a.row_end) {
                float uu, vv;
                lk_solve(Sxx[j], Sxy[j], Syy[j], Sxt[j], Syt[j], uu, vv);
                if (a.add_base) {
                    uu = base_u[j] + uu;
                    vv = base_v[j] + vv;
                }
                if constexpr (STREAM) {
                    link->ou[j] = uu;
                    link->ov[j] = vv;
                } else {
                    ou[(size_t)gy * a.out_stride + gx] = uu;
                    ov[(size_t)gy * a.out_stride + gx] = vv;
                }
            }
        }
    }
    MICV_STAMP(5)
#undef MICV_STAMP
#undef MICV_STOP
}

// Workgroup -> tile order.  (1) XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so workgroup b
// and b+8 share an L2; each XCD gets a contiguous run of row-major tiles, so neighbouring tiles
// (which share halo rows of prev / next / coarse flow) hit the same L2.  (2) Border tiles first:
// they take the bounds-checked body and run ~2x longer than interior tiles; dealt in plain
// row-major order the bottom image row would be the launch's tail.  Each XCD therefore runs
// its share of the border tiles first and then its contiguous run of interior tiles.  The
// interior tiles form a rectangle [ix0, ix1) x [iy0, iy1) (the predicate below is separable);
// the mapping is a bijection for ANY rectangle inside the grid, so a rectangle that disagreed
// with the predicate would cost speed only.
template <typename C>
__device__ __forceinline__ void lk_tile_of(const LkLevelArgs &a, int bidx, int &tile_x, int &tile_y) {
    constexpr int E = C::M > 2 ? C::M : 2;
    const int nb = gridDim.x, per = nb >> 3, rem = nb & 7;
    const int xcd = bidx & 7, idx = bidx >> 3;
    const int tiles_x = (a.cols + C::TW - 1) / C::TW;
    const int tiles_y = nb / tiles_x, ty_base = a.row_begin / C::TH;
    {
        constexpr int FX = C::TW + C::H + E, FY = C::TH + C::H + E;
        int ix0 = (C::H + E + C::TW - 1) / C::TW, ix1 = a.cols >= FX ? (a.cols - FX) / C::TW + 1 : 0;
        // tile rows start at ty * TH + y_shift (band launches): interior <=> y_shift + ty*TH - H - E >= 0 and
        // y_shift + ty*TH + FY <= rows
        int iy0 = (C::H + E - a.y_shift + C::TH - 1) / C::TH - ty_base;
        int iy1 = (a.rows >= FY + a.y_shift ? (a.rows - FY - a.y_shift) / C::TH + 1 : 0) - ty_base;
        ix1 = ix1 < tiles_x ? ix1 : tiles_x;
        iy0 = iy0 > 0 ? iy0 : 0;
        iy1 = iy1 < tiles_y ? iy1 : tiles_y;
        const int iw = ix1 > ix0 ? ix1 - ix0 : 0, ih = iy1 > iy0 ? iy1 - iy0 : 0;
        const int ni = iw * ih, nbd = nb - ni;
        const int t_plain = xcd * per + (xcd < rem ? xcd : rem) + idx;
        if (ni * 2 >= nb && nb >= 64) {
            // this XCD: border tiles [b0, b1) of the border list, then interior tiles from i0 on
            const int b0 = (int)(((long long)xcd * nbd) >> 3), b1 = (int)(((long long)(xcd + 1) * nbd) >> 3);
            if (idx < b1 - b0) {
                int b = b0 + idx;
                const int n_top = iy0 * tiles_x, n_bot = (tiles_y - iy1) * tiles_x, side = tiles_x - iw;
                if (b < n_top) {
                    tile_y = b / tiles_x;
                    tile_x = b - tile_y * tiles_x;
                } else if (b < n_top + n_bot) {
                    b -= n_top;
                    tile_y = b / tiles_x;
                    tile_x = b - tile_y * tiles_x;
                    tile_y += iy1;
                } else {
                    b -= n_top + n_bot;
                    const int r = b / side, c = b - r * side;
                    tile_y = iy0 + r;
                    tile_x = c < ix0 ? c : c + iw;
                }
            } else {
                const int i = t_plain - b1;  // = interior tiles of lower XCDs + (idx - own border count)
                const int r = i / iw;
                tile_y = iy0 + r;
                tile_x = ix0 + (i - r * iw);
            }
        } else {
            tile_y = t_plain / tiles_x;
            tile_x = t_plain - tile_y * tiles_x;
        }
        tile_y += ty_base;
    }
}

// The extra workgroups of a launch that carries a pyramid-build job (LkBuildJob, lk_fused.hpp).  Plain copies: every
// thread issues the loads of all its rows first, then stores.
template <int NT>
__device__ __forceinline__ void lk_build_block(const LkBuildJob &j, int block, int tid) {
    constexpr int RPT = 32 / (NT / 64);  // rows per thread of a 32-row unit
    static_assert(NT % 64 == 0 && 32 % (NT / 64) == 0, "a unit is 32 rows x 64 lanes");
    const int lane = tid & 63, trow = (tid >> 6) * RPT;
    const int l = j.level, rl = j.rows >> l, cl = j.cols >> l;
    const int bx = l == 1 ? (cl + 127) / 128 : (cl + 63) / 64, per_img = bx * ((rl + 31) / 32);
    for (int u = block; u < j.units; u += j.blocks) {
        const int img = u / per_img, r = u - img * per_img;
        const int set = img / j.batch, b = img - set * j.batch;
        const int byi = r / bx, bxi = r - byi * bx;
        const int y0 = byi * 32 + trow;
        const float *src = (set ? j.src_b : j.src_a) + b * j.img_elems;
        float *dst = (set ? j.pyr_b : j.pyr_a) + j.dst_off + (size_t)b * rl * cl;
        if (l == 1) {
            // the float4 at level-0 column 4 * x2 of row 2y + 1 holds level-1 columns 2 * x2 and 2 * x2 + 1 (cols % 4 == 0)
            const int x2 = bxi * 64 + lane;
            if (2 * x2 >= cl) continue;
            float4 q[RPT];
#pragma unroll
            for (int i = 0; i < RPT; i++)
                if (y0 + i < rl) q[i] = *reinterpret_cast<const float4 *>(src + (size_t)(2 * (y0 + i) + 1) * j.sstride + 4 * x2);
#pragma unroll
            for (int i = 0; i < RPT; i++)
                if (y0 + i < rl) *reinterpret_cast<float2 *>(dst + (size_t)(y0 + i) * cl + 2 * x2) = make_float2(q[i].y, q[i].w);
        } else {
            const int x = bxi * 64 + lane, sh = (1 << l) - 1;
            if (x >= cl) continue;
            float q[RPT];
#pragma unroll
            for (int i = 0; i < RPT; i++)
                if (y0 + i < rl) q[i] = src[(size_t)(((y0 + i) << l) + sh) * j.sstride + (x << l) + sh];
#pragma unroll
            for (int i = 0; i < RPT; i++)
                if (y0 + i < rl) dst[(size_t)(y0 + i) * cl + x] = q[i];
        }
    }
}

// (r05: an exact amdgpu_waves_per_eu(4, 4) hint -- LDS allows no fifth wave per SIMD -- changes nothing here: same 82 VGPRs,
// same 197.8 us; it is worth 4 % on the stereo SSD kernel, stereo.hip)
template <int R, int MODE, int NTV, int THV = 32, bool GATHER = false, int TWV = 64>
__global__ __launch_bounds__(NTV, lk_waves_per_simd_rt(R, NTV, THV)) void lk_level_kernel(LkLevelArgs a, TapsN<2 * R + 1> g) {
    using C = LkCfg<R, NTV, THV, TWV>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int E = C::M > 2 ? C::M : 2;
    if (blockIdx.y >= a.batch) {  // the launch carries a pyramid-build job: rows of the grid behind the pairs
        lk_build_block<NTV>(a.job, (blockIdx.y - a.batch) * gridDim.x + blockIdx.x, threadIdx.x);
        return;
    }
    int tile_x, tile_y;
    lk_tile_of<C>(a, blockIdx.x, tile_x, tile_y);
    const int rx0 = tile_x * C::TW - C::H, ry0 = tile_y * C::TH + a.y_shift - C::H;
    // interior: tile + halo + the staged `next` margin (>= the pyrUp support) inside the image
    const bool interior = rx0 - E >= 0 && rx0 + C::RW + E <= a.cols && ry0 - E >= 0 &&
                          ry0 + C::RH + E <= a.rows;
    if (interior)
        lk_tile<R, MODE, true, NTV, false, THV, false, false, GATHER, true, TWV>(a, g, lds, tile_x, tile_y, blockIdx.y);
    else
        lk_tile<R, MODE, false, NTV, false, THV, false, false, GATHER, true, TWV>(a, g, lds, tile_x, tile_y, blockIdx.y);
}

// Pre-pass of a split launch (r05): the tiles of lk_level_kernel with the halo of a 7-tap window -- R = 3: image halo 4,
// i.e. the Sobel ring plus whole 16-byte chunks -- running phases 0-3 and leaving Ix, Iy, It and the base flow in HBM.
template <int R, int NTV, int THV = 32>
__global__ __launch_bounds__(NTV, lk_waves_per_simd(NTV)) void lk_grad_kernel(LkLevelArgs a) {
    using C = LkCfg<R, NTV, THV>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int E = C::M > 2 ? C::M : 2;
    int tile_x, tile_y;
    lk_tile_of<C>(a, blockIdx.x, tile_x, tile_y);
    const int rx0 = tile_x * C::TW - C::H, ry0 = tile_y * C::TH - C::H;
    const bool interior = rx0 - E >= 0 && rx0 + C::RW + E <= a.cols && ry0 - E >= 0 && ry0 + C::RH + E <= a.rows;
    TapsN<2 * R + 1> g = {};  // the pre-pass sums no windows
    if (interior)
        lk_tile<R, LK_FLOW_COARSE, true, NTV, false, THV, false, false, false, false, 64, true>(a, g, lds, tile_x, tile_y, blockIdx.y);
    else
        lk_tile<R, LK_FLOW_COARSE, false, NTV, false, THV, false, false, false, false, 64, true>(a, g, lds, tile_x, tile_y, blockIdx.y);
}

// Strip launch (r05, lk_strip.hpp): the interior tiles of the level as streamed strips, the border tiles as tiles, one
// launch.  Workgroups [0, items) take the strip items -- numbered strip-fastest, dealt to the XCDs in contiguous runs
// (neighbouring strips share 32 of their 96 window columns) -- the rest take the border tiles (top rows, bottom rows,
// side columns); the long items are dispatched first and the short border tiles fill in behind them.
struct LkStripPlan {
    int items, strips, segs, seg_rows;  // per launch: items = batch * segs * strips
    int ix0, iw, iy0, ih;               // the interior tile rectangle (64x32 tiles)
    int tiles_x, tiles_y, border;       // border tiles per pair
};

template <int R, int NTV>
__global__ __launch_bounds__(NTV, lk_waves_per_simd(NTV)) void lk_level_strip_kernel(LkLevelArgs a, TapsN<2 * R + 1> g, LkStripPlan p) {
    using C = LkCfg<R, NTV>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int b = blockIdx.x;
    if (b < p.items) {
        const int per = p.items >> 3, rem = p.items & 7;
        const int xcd = b & 7, idx = b >> 3;
        // (when items is not a multiple of 8 the last run is short: workgroups past it on their XCD take the tail)
        int item = xcd * per + (xcd < rem ? xcd : rem) + idx;
        if (idx >= per + (xcd < rem ? 1 : 0)) return;  // cannot happen: the grid has exactly `items` strip workgroups
        const int strip = item % p.strips, t = item / p.strips;
        const int seg = t % p.segs, pair = t / p.segs;
        const int x0 = (p.ix0 + strip) * C::TW;
        const int s0 = p.iy0 * C::TH + seg * p.seg_rows;
        const int end = (p.iy0 + p.ih) * C::TH;
        const int s1 = s0 + p.seg_rows < end ? s0 + p.seg_rows : end;
        lk_strip<R>(a, g, lds, x0, s0, s1, pair);
        return;
    }
    b -= p.items;
    const int pair = b / p.border;
    b -= pair * p.border;
    int tile_x, tile_y;
    const int iy1 = p.iy0 + p.ih, n_top = p.iy0 * p.tiles_x, n_bot = (p.tiles_y - iy1) * p.tiles_x, side = p.tiles_x - p.iw;
    if (b < n_top) {
        tile_y = b / p.tiles_x;
        tile_x = b - tile_y * p.tiles_x;
    } else if (b < n_top + n_bot) {
        b -= n_top;
        tile_y = b / p.tiles_x;
        tile_x = b - tile_y * p.tiles_x;
        tile_y += iy1;
    } else {
        b -= n_top + n_bot;
        const int r = b / side, c = b - r * side;
        tile_y = p.iy0 + r;
        tile_x = c < p.ix0 ? c : c + p.iw;
    }
    lk_tile<R, LK_FLOW_COARSE, false, NTV>(a, g, lds, tile_x, tile_y, pair);
}

// Chain launch: workgroup b runs the `count` vertically adjacent tiles of sched[b] (tile_x, first
// tile_y, count, pair), carrying the gradient rows from one tile to the next.  A 1-D grid: the
// host-built schedule already contains the batch, the XCD-aware placement and the order of issue.
template <int R, int NTV, bool GATHER = false>
__global__ __launch_bounds__(NTV, lk_waves_per_simd(NTV)) void lk_level_chain_kernel(LkLevelArgs a, TapsN<2 * R + 1> g,
                                                                        const int4 *__restrict__ sched) {
    using C = LkCfg<R, NTV>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int E = C::M > 2 ? C::M : 2;
    if (a.job.blocks && (int)blockIdx.x >= a.job.first_block) {  // the workgroups behind the schedule: a pyramid-build job
        lk_build_block<NTV>(a.job, blockIdx.x - a.job.first_block, threadIdx.x);
        return;
    }
    const int4 e = sched[blockIdx.x];
    if (e.z <= 0) return;  // padding entry
    const int rx0 = e.x * C::TW - C::H, ry0 = e.y * C::TH - C::H;
    const bool interior = rx0 - E >= 0 && rx0 + C::RW + E <= a.cols && ry0 - E >= 0 &&
                          ry0 + C::RH + E <= a.rows;
    if (!interior) {  // border tiles are never chained
        lk_tile<R, LK_FLOW_COARSE, false, NTV, false, 32, false, false, GATHER>(a, g, lds, e.x, e.y, e.w);
        return;
    }
    lk_tile<R, LK_FLOW_COARSE, true, NTV, false, 32, false, false, GATHER>(a, g, lds, e.x, e.y, e.w, e.z > 1);
    for (int t = 1; t < e.z; t++) {
        __syncthreads();  // the tile above is done with the row buffers / staged images this one overwrites
        lk_tile<R, LK_FLOW_COARSE, true, NTV, true, 32, false, true, GATHER>(a, g, lds, e.x, e.y + t, e.w, t + 1 < e.z);
    }
}

// Streamed launch: a persistent grid (two workgroups per CU) whose workgroups take tiles off their
// XCD's list by ticket (the schedule of the chain launch with single tiles: every pair's border
// tiles first, then contiguous runs of interior tiles per XCD).  An interior tile's staging is
// taken off its critical path: its `next` window and coarse block are DMA'd into the gradient area
// of the tile BEFORE it while that one runs its last column passes and the solve, and its prev tile
// is DMA'd at its own start but not awaited until the warp phase is over (in-kernel stamps of the
// plain launch: a tile spends 32 % of its life waiting for phase 0).  tickets[0..7] = per-XCD
// counters, tickets[8] = workgroups that have left; the last one out zeroes them for the next launch.
template <int R, int NTV, int THV = 32>
__global__ __launch_bounds__(NTV, lk_waves_per_simd(NTV)) void lk_level_stream_kernel(LkLevelArgs a, TapsN<2 * R + 1> g,
                                                                         const int4 *__restrict__ sched, int per_xcd,
                                                                         unsigned *__restrict__ tickets) {
    using C = LkCfg<R, NTV, THV>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int slot;
    const int tid = threadIdx.x;
    LkStreamLink link;
    link.sched = sched;
    link.per_xcd = per_xcd;
    link.xcd = blockIdx.x & 7;
    link.counter = tickets + link.xcd;
    link.slot = &slot;
    link.next = make_int4(0, 0, 0, 0);
    link.next_staged = false;
    if (tid == 0) slot = (int)atomicAdd(link.counter, 1u);
    __syncthreads();
    int t = __builtin_amdgcn_readfirstlane(slot);
    int4 e = t < per_xcd ? sched[8 * t + link.xcd] : make_int4(0, 0, 0, 0);
    bool staged = false;
    int4 pending = make_int4(0, 0, 0, 0);  // the streamed tile whose results are still in registers
    auto flush = [&]() {
        if (pending.z > 0) {
            float *__restrict__ ou = a.out_u + pending.w * a.out_pair;
            float *__restrict__ ov = a.out_v + pending.w * a.out_pair;
            const int gx = pending.x * C::TW + (tid & (C::TW - 1));
            const int gy0 = pending.y * C::TH + C::RPT * (tid / C::TW);
#pragma unroll
            for (int j = 0; j < C::RPT; j++) {
                ou[(size_t)(gy0 + j) * a.out_stride + gx] = link.ou[j];
                ov[(size_t)(gy0 + j) * a.out_stride + gx] = link.ov[j];
            }
            pending.z = 0;
        }
    };
    while (e.z > 0) {  // a padding entry ends the list
        if (lk_tile_interior<C>(a, e.x, e.y)) {
            if (!staged) {  // the first tile, or the one after a border tile
                lk_stage_ahead<C>(a, lds, e.x, e.y, e.w, tid);
                __syncthreads();
            }
            flush();  // the previous tile's stores ride behind this tile's first phases
            lk_tile<R, LK_FLOW_COARSE, true, NTV, false, THV, true>(a, g, lds, e.x, e.y, e.w, false, &link);
            pending = e;
            e = link.next;
            staged = link.next_staged;
        } else {
            flush();
            int tk = 0;
            if (tid == 0) tk = (int)atomicAdd(link.counter, 1u);
            lk_tile<R, LK_FLOW_COARSE, false, NTV, false, THV, false, true>(a, g, lds, e.x, e.y, e.w);
            if (tid == 0) slot = tk;
            __syncthreads();
            t = __builtin_amdgcn_readfirstlane(slot);
            e = t < per_xcd ? sched[8 * t + link.xcd] : make_int4(0, 0, 0, 0);
            staged = false;
        }
        __syncthreads();  // the tile is done with the row buffers; what was staged ahead has landed
    }
    flush();
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(tickets + 8, 1u) == gridDim.x - 1) {
            for (int x = 0; x < 8; x++) tickets[x] = 0;
            tickets[8] = 0;
            __threadfence();
        }
    }
}

// Host: the chain schedule of one level launch.  Interior tiles of every column are cut into chains
// of decreasing length (max_chain first, halving towards the bottom of the column), border tiles
// stay single.  Order of issue = expected duration, longest first (list scheduling: the short items
// at the end fill the slots the long ones leave); inside a class neighbouring columns are
// neighbours in the list, and every class is dealt to the 8 XCDs in contiguous runs (workgroup b
// lands on XCD b % 8), so chains that share halo columns run at the same time on the same L2.
template <typename C>
static void build_chain_schedule(int rows, int cols, int batch, int max_chain, std::vector<int4> *out) {
    constexpr int E = C::M > 2 ? C::M : 2;
    const int tiles_x = cdiv(cols, C::TW), tiles_y = cdiv(rows, C::TH);
    auto interior = [&](int tx, int ty) {
        const int rx0 = tx * C::TW - C::H, ry0 = ty * C::TH - C::H;
        return rx0 - E >= 0 && rx0 + C::RW + E <= cols && ry0 - E >= 0 && ry0 + C::RH + E <= rows;
    };
    // classes by chain length (index = length), plus the border singles
    std::vector<std::vector<int4>> cls(max_chain + 1);
    std::vector<int4> border;
    for (int p = 0; p < batch; p++) {
        for (int ty = 0; ty < tiles_y; ty++)
            for (int tx = 0; tx < tiles_x; tx++)
                if (!interior(tx, ty)) border.push_back(make_int4(tx, ty, 1, p));
        // interior rows of a column are contiguous (the predicate is separable); cut them per segment so
        // that one segment's chains of all columns are neighbours in the class list.  The row range comes
        // from the y half of the predicate alone: probing a column (r02 probed tiles_x / 2) fails when that
        // column is not x-interior -- cols 194..206 have tiles_x = 4 and only column 1 interior -- and the
        // interior tiles then landed in neither list.
        auto interior_y = [&](int ty) {
            const int ry0 = ty * C::TH - C::H;
            return ry0 - E >= 0 && ry0 + C::RH + E <= rows;
        };
        int iy0 = 0;
        while (iy0 < tiles_y && !interior_y(iy0)) iy0++;
        int iy1 = iy0;
        while (iy1 < tiles_y && interior_y(iy1)) iy1++;
        int y = iy0;
        while (y < iy1) {
            const int rem = iy1 - y;
            int len = rem / 2 > max_chain ? max_chain : rem / 2;
            if (len < 1) len = 1;
            if (rem <= max_chain && rem <= 2) len = 1;
            for (int tx = 0; tx < tiles_x; tx++)
                if (interior(tx, y)) cls[len].push_back(make_int4(tx, y, len, p));
            y += len;
        }
    }
    std::vector<int4> per_xcd[8];
    auto deal = [&](const std::vector<int4> &v) {
        const size_t n = v.size();
        for (int x = 0; x < 8; x++)
            for (size_t i = n * x / 8; i < n * (x + 1) / 8; i++) per_xcd[x].push_back(v[i]);
    };
    for (int len = max_chain; len >= 2; len--) deal(cls[len]);
    deal(border);  // ~1.5 tile times each: after the chains, before the interior singles
    if (max_chain >= 1) deal(cls[1]);
    size_t longest = 0;
    for (auto &v : per_xcd) longest = v.size() > longest ? v.size() : longest;
    out->clear();
    for (size_t i = 0; i < longest; i++)
        for (int x = 0; x < 8; x++) out->push_back(i < per_xcd[x].size() ? per_xcd[x][i] : make_int4(0, 0, 0, 0));
}

bool lk_fused_supports(int win) { return win == 15 || win == 7 || win == 21 || win == 11; }
bool lk_split_supports(int win) { return win == 15; }

// MICV_OPT_LK_SPLIT: 0 = never (default: every form measured 28-42 % SLOWER than the fused launch on MI355X, r05,
// profiles/r05/split_ab.txt), 1 = every launch that can (whole frames of at least 64 x 64 with a doubling coarse flow;
// gradient planes, base flow recomputed by the sums kernel), 2 = 1 with the base flow through u, v, 3 = variant A' (the
// pre-pass leaves the warped image only; second half = the fused kernel in its no-flow mode).
bool lk_split_wanted(int rows, int cols, int batch, int win, int mode, int split_opt) {
    (void)batch;
    return split_opt > 0 && lk_split_supports(win) && mode == LK_FLOW_COARSE && rows >= 64 && cols >= 64 && (cols & 3) == 0;
}

// Host-only view of the schedule the chain / streamed launches walk (micv_lk_schedule_host): lets a
// CPU test check that every (tile x, tile y, pair) is covered exactly once.
int lk_schedule_host(int rows, int cols, int batch, int win, int max_chain, std::vector<int4> *out) {
    switch (win) {
        case 15: build_chain_schedule<LkCfg<7, 512>>(rows, cols, batch, max_chain, out); return 32;
        case 11: build_chain_schedule<LkCfg<5, 512>>(rows, cols, batch, max_chain, out); return 32;
        case 7: build_chain_schedule<LkCfg<3, 512>>(rows, cols, batch, max_chain, out); return 32;
        default: return 0;
    }
}

// The cached device copy of a launch shape's schedule (built on first use).
template <typename C>
static int get_schedule(const LkLevelArgs &a, int max_chain, const int4 **sched, int *nblocks) {
    for (auto &e : a.ctx->lk_sched)
        if (e.rows == a.rows && e.cols == a.cols && e.batch == a.batch && e.r == C::R && e.max_chain == max_chain &&
            e.th == C::TH) {
            *sched = static_cast<const int4 *>(e.dev);
            *nblocks = e.nblocks;
            return MICV_OK;
        }
    std::vector<int4> host;
    build_chain_schedule<C>(a.rows, a.cols, a.batch, max_chain, &host);
    void *dev = nullptr;
    MICV_HIP(hipMalloc(&dev, host.size() * sizeof(int4)));
    hipError_t ce = hipMemcpy(dev, host.data(), host.size() * sizeof(int4), hipMemcpyHostToDevice);
    if (ce != hipSuccess) {
        (void)hipFree(dev);
        MICV_HIP(ce);
    }
    if (a.ctx->lk_sched.size() >= 32) {  // shapes keep changing: start over
        MICV_HIP(hipDeviceSynchronize());  // a launch in flight may still read one of them
        for (auto &e : a.ctx->lk_sched) (void)hipFree(e.dev);
        a.ctx->lk_sched.clear();
    }
    a.ctx->lk_sched.push_back({a.rows, a.cols, a.batch, C::R, max_chain, C::TH, dev, (int)host.size()});
    *sched = static_cast<const int4 *>(dev);
    *nblocks = (int)host.size();
    return MICV_OK;
}

template <int R, int NTV, int THV = 32, bool GATHER = false, int TWV = 64>
static int launch_r(hipStream_t s, const LkLevelArgs &a) {
    using C = LkCfg<R, NTV, THV, TWV>;
    if constexpr (!GATHER && R == 7 && NTV == 512 && TWV == 64) {
        // pyramid level k read straight from level 0 (img_xstride = 2^k): the gather-staging instantiations
        // (window 15, 512-thread tiles only -- the option is off by default and every instantiation costs build time)
        if (a.img_xstride != 1) return launch_r<R, NTV, THV, true>(s, a);
    }
    if (!GATHER && a.img_xstride != 1) {
        set_error("lk fused: levels read from level 0 (pixel stride %d) are built for window 15 only", a.img_xstride);
        return MICV_EUNSUPPORTED;
    }
    static TapsN<2 * R + 1> taps;
    static std::once_flag once;
    std::call_once(once, [] {
        Taps t;
        gaussian_taps(2 * R + 1, (double)((float)(2 * R + 1) / 3.f), &t);  // OpticalFlow.cpp:73
        for (int i = 0; i < 2 * R + 1; i++) taps.k[i] = t.k[i];
    });
    // Dynamic LDS above 64 KB needs the attribute; set it on every device we launch on.
    if (!a.name_out) {
        static thread_local int done_dev = -1;
        int dev = 0;
        MICV_HIP(hipGetDevice(&dev));
        if (done_dev != dev) {
            if constexpr (TWV == 64) {  // (32-wide tiles exist for the coarse-flow mode only)
                MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lk_level_kernel<R, 0, NTV, THV, GATHER, TWV>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)C::LDS_BYTES));
                MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lk_level_kernel<R, 2, NTV, THV, GATHER, TWV>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)C::LDS_BYTES));
            }
            MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lk_level_kernel<R, 1, NTV, THV, GATHER, TWV>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::LDS_BYTES));
            done_dev = dev;
        }
    }
    if (a.row_begin < 0 || a.row_end > a.rows || a.row_begin >= a.row_end) {
        set_error("lk fused: bad row band [%d, %d) for %d rows", a.row_begin, a.row_end, a.rows);
        return MICV_EINVAL;
    }
    if constexpr (!GATHER && TWV == 64 && C::FAST && C::RW % 4 == 0 && C::NW % 4 == 0 && ((THV == 32 && NTV == 512) || (THV == 64 && NTV == 1024))) {
        // Streamed launch (MICV_OPT_LK_STREAM = 1; off by default): whole frames with a coarse flow whose
        // images the LDS-DMA can address (16-byte rows).  Measured on MI355X (8 x 1080p, tools/stream_bench.py,
        // one box): level-0 launch 0.244 ms against 0.220 ms for the plain grid -- the loop's staging
        // address arithmetic costs 11 % more VALU instructions (109.5 M against 98.4 M) and hiding the
        // staging latency buys nothing back: with two workgroups per CU the other workgroup already
        // covers it.  Kept as an option (bit-exact, tested), not a default.
        const long tiles = (long)cdiv(a.cols, C::TW) * cdiv(a.rows, C::TH) * a.batch;
        const bool dma_ok = (a.img_stride & 3) == 0 && (a.img_pair & 3) == 0 &&
                            ((reinterpret_cast<uintptr_t>(a.prev) | reinterpret_cast<uintptr_t>(a.next)) & 15) == 0;
        if (a.mode == LK_FLOW_COARSE && a.ctx && a.stream_tiles > 0 &&
            a.max_chain <= 1 && a.max_chain >= 0 && dma_ok && a.row_begin == 0 && a.row_end == a.rows &&
            a.rows == 2 * a.flow_rows && a.cols == 2 * a.flow_cols && a.stamps == nullptr && a.stop_after < 0) {
            if (a.name_out) {
                snprintf(a.name_out, a.name_cap, "lk_level_stream_kernel<%d, %d, %d>", R, NTV, THV);
                return MICV_OK;
            }
            const int4 *sched = nullptr;
            int nblocks = 0;
            MICV_TRY(get_schedule<C>(a, 1, &sched, &nblocks));
            unsigned *tickets = nullptr;
            const int trc = a.ctx->lk_ticket_slot(s, &tickets);
            if (trc != MICV_OK && trc != MICV_EUNSUPPORTED) return trc;
            if (trc == MICV_OK) {
            static thread_local int stream_dev = -1;
            static thread_local int n_cu = 0;
            int dev = 0;
            MICV_HIP(hipGetDevice(&dev));
            if (stream_dev != dev) {
                MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lk_level_stream_kernel<R, NTV, THV>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
                MICV_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
                stream_dev = dev;
            }
            const long per_cu = NTV >= 1024 ? 1 : 2;  // resident workgroups per CU
            long want = per_cu * n_cu < tiles ? per_cu * n_cu : tiles;
            const int grid = (int)((want + 7) / 8) * 8;
            lk_level_stream_kernel<R, NTV, THV><<<grid, C::NT, C::LDS_BYTES, s>>>(a, taps, sched, nblocks / 8, tickets);
            MICV_LAUNCH_CHECK();
            return MICV_OK;
            }  // no ticket slot left for this stream: the plain launch below
        }
    }
    if constexpr (C::CHAIN_OK) {
        if (a.mode == LK_FLOW_COARSE && a.ctx && a.max_chain != 1 && a.row_begin == 0 && a.row_end == a.rows &&
            a.rows == 2 * a.flow_rows && a.cols == 2 * a.flow_cols) {
            // Measured on MI355X (8 x 1080p, tools/chain_bench.py): pairs of tiles are the best chain
            // length (level 0: 0.243 -> 0.233 ms; longer chains lose to the coarser work items), and the
            // schedule alone (all pairs' border tiles first, one 1-D grid) is worth as much again on the
            // levels that fill the GPU only once or twice.  Launches below one round stay on the plain grid.
            const long tiles = (long)cdiv(a.cols, C::TW) * cdiv(a.rows, C::TH) * a.batch;
            // A single 1080p pair (1020 tiles = two exact rounds) is slower with chains (0.041 -> 0.045 ms):
            // chains from four rounds on; between one and four rounds a batch still gains from the
            // schedule's order (every pair's border tiles first), with single tiles.
            // With the shuffle-free packed row pass (r02) the chain kernel no longer beats the plain one
            // (A/B on one box, bench.py: 41.6 vs 41.6 Gpix/s, sustained 0.372 vs 0.368 ms): chains are
            // an option (MICV_OPT_LK_CHAIN > 1), not the default.
            // r03b, on the kernel as it is now (tools/level_bench.py, 8 / 16 pairs): chains of two still lose on
            // level 0 and level 1 (0.203 vs 0.199, 0.059 vs 0.058 ms) but win where the tile count is a little over
            // a whole number of rounds -- level 2 of 8 pairs, 576 tiles on 512 slots: 0.0286 -> 0.0251 ms (288
            // two-tile workgroups in one round instead of two rounds of single tiles); 16 pairs, 1152 tiles:
            // 0.0421 -> 0.0382.  MICV_OPT_LK_CHAIN = 0 takes them exactly there; 1 = never.
            int max_chain = a.max_chain > 1 ? a.max_chain : 1;
            if (a.max_chain == 0) {
                static thread_local int slots_dev = -1;
                static thread_local long slots = 512;
                int dev = 0;
                MICV_HIP(hipGetDevice(&dev));
                if (slots_dev != dev) {
                    int n_cu = 256;
                    MICV_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
                    slots = 2L * n_cu;  // two 512-thread workgroups per CU
                    slots_dev = dev;
                }
                const long rounds = tiles / slots, rem = tiles % slots;
                if (rounds >= 1 && rounds <= 2 && rem > 0 && rem <= slots / 4) max_chain = 2;
            }
            if (max_chain > 32) max_chain = 32;
            const bool sched_only = a.max_chain < 0;
            if (sched_only) max_chain = 1;  // the schedule kernel with single tiles only
            if ((max_chain > 1 || sched_only) && a.name_out) {
                snprintf(a.name_out, a.name_cap, "lk_level_chain_kernel<%d, %d, %s>", R, NTV, GATHER ? "true" : "false");
                return MICV_OK;
            }
            if (max_chain > 1 || sched_only) {
                const int4 *sched = nullptr;
                int nblocks = 0;
                MICV_TRY(get_schedule<C>(a, max_chain, &sched, &nblocks));
                {
                    static thread_local int chain_dev = -1;
                    int dev = 0;
                    MICV_HIP(hipGetDevice(&dev));
                    if (chain_dev != dev) {
                        MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lk_level_chain_kernel<R, NTV, GATHER>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
                        chain_dev = dev;
                    }
                }
                LkLevelArgs c = a;
                c.job.first_block = nblocks;
                lk_level_chain_kernel<R, NTV, GATHER><<<nblocks + c.job.blocks, C::NT, C::LDS_BYTES, s>>>(c, taps, sched);
                MICV_LAUNCH_CHECK();
                return MICV_OK;
            }
        }
    }
    // Band launches (row-sharded execution) tile from row_begin, not from the multiple of TH below it: a
    // 135-row band is 5 tile rows instead of 6.  pyrUp pairs fine rows (2m, 2m+1) per coarse row, so the
    // modes with a base flow need an even origin (the row-shard plan's cuts are even on those levels).
    if (a.name_out) {
        snprintf(a.name_out, a.name_cap, "lk_level_kernel<%d, %d, %d, %d, %s, %d>", R, a.mode, NTV, THV, GATHER ? "true" : "false", TWV);
        return MICV_OK;
    }
    LkLevelArgs b = a;
    b.y_shift = (a.row_begin > 0 && (a.mode == LK_FLOW_NONE || (a.row_begin & 1) == 0)) ? a.row_begin % C::TH : 0;
    const int tile_rows = b.y_shift ? cdiv(a.row_end - a.row_begin, C::TH) : cdiv(a.row_end, C::TH) - a.row_begin / C::TH;
    dim3 grid(cdiv(a.cols, C::TW) * tile_rows, a.batch);
    if (b.job.blocks > 0) {  // the build job's workgroups: whole rows of the grid behind the pairs (dispatched after the tiles)
        long extra_y = cdiv(b.job.blocks, (int)grid.x);
        if (a.batch + extra_y > 65535) extra_y = 65535 - a.batch;
        b.job.blocks = (int)(extra_y * grid.x);  // every extra workgroup works; units beyond go round in the stride loop
        grid.y = a.batch + (unsigned)extra_y;
    }
    switch (a.mode) {
        case LK_FLOW_NONE:
            if constexpr (TWV == 64) lk_level_kernel<R, 0, NTV, THV, GATHER, TWV><<<grid, C::NT, C::LDS_BYTES, s>>>(b, taps);
            break;
        case LK_FLOW_COARSE:
            if (a.rows != 2 * a.flow_rows || a.cols != 2 * a.flow_cols) {
                set_error("lk fused: coarse flow %dx%d does not double to %dx%d", a.flow_rows,
                          a.flow_cols, a.rows, a.cols);
                return MICV_EINVAL;
            }
            lk_level_kernel<R, 1, NTV, THV, GATHER, TWV><<<grid, C::NT, C::LDS_BYTES, s>>>(b, taps);
            break;
        case LK_FLOW_FULL:
            if constexpr (TWV == 64) lk_level_kernel<R, 2, NTV, THV, GATHER, TWV><<<grid, C::NT, C::LDS_BYTES, s>>>(b, taps);
            break;
        default:
            set_error("lk fused: bad mode %d", a.mode);
            return MICV_EINVAL;
    }
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// The strip launch (MICV_OPT_LK_STRIP): whole frames with a doubling coarse flow whose images the LDS-DMA can address, and
// at least one interior tile row and column.  blocks = rows of a segment / 16 (0 = the default).
template <int R>
static int launch_strip(hipStream_t s, const LkLevelArgs &a, int blocks) {
    using C = LkCfg<R, 512>;
    using SC = StripCfg<R>;
    static_assert(SC::LDS_BYTES <= C::LDS_BYTES, "a strip workgroup fits the tile body's LDS");
    constexpr int E = C::M > 2 ? C::M : 2, FX = C::TW + C::H + E, FY = C::TH + C::H + E;
    LkStripPlan p;
    p.tiles_x = cdiv(a.cols, C::TW);
    p.tiles_y = cdiv(a.rows, C::TH);
    p.ix0 = (C::H + E + C::TW - 1) / C::TW;
    int ix1 = a.cols >= FX ? (a.cols - FX) / C::TW + 1 : 0;
    p.iy0 = (C::H + E + C::TH - 1) / C::TH;
    int iy1 = a.rows >= FY ? (a.rows - FY) / C::TH + 1 : 0;
    ix1 = ix1 < p.tiles_x ? ix1 : p.tiles_x;
    iy1 = iy1 < p.tiles_y ? iy1 : p.tiles_y;
    p.iw = ix1 - p.ix0;
    p.ih = iy1 - p.iy0;
    if (p.iw < 1 || p.ih < 1) return MICV_EUNSUPPORTED;
    // the warm-up block above a segment reads 16 rows above the first interior tile row: inside the image by the interior
    // predicate (tile + halo + margin = 16 rows) -- and the strip body assumes R + 1 = H
    if (p.iy0 * C::TH < SC::B + SC::M) return MICV_EUNSUPPORTED;
    p.strips = p.iw;
    p.seg_rows = (blocks > 0 ? blocks : 16) * SC::B;
    p.segs = cdiv(p.ih * C::TH, p.seg_rows);
    p.items = p.strips * p.segs * a.batch;
    p.border = p.tiles_x * p.tiles_y - p.iw * p.ih;
    static TapsN<2 * R + 1> taps;
    static std::once_flag once;
    std::call_once(once, [] {
        Taps t;
        gaussian_taps(2 * R + 1, (double)((float)(2 * R + 1) / 3.f), &t);  // OpticalFlow.cpp:73
        for (int i = 0; i < 2 * R + 1; i++) taps.k[i] = t.k[i];
    });
    static thread_local int done_dev = -1;
    int dev = 0;
    MICV_HIP(hipGetDevice(&dev));
    if (done_dev != dev) {
        MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lk_level_strip_kernel<R, 512>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        done_dev = dev;
    }
    LkLevelArgs b = a;
    b.y_shift = 0;
    b.job.blocks = 0;
    const long grid = (long)p.items + (long)p.border * a.batch;
    lk_level_strip_kernel<R, 512><<<(unsigned)grid, 512, C::LDS_BYTES, s>>>(b, taps, p);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

// The split launch: pre-pass (phases 0-3 once per pixel + 1-px... see lk_grad_kernel) then the streaming sums kernel.
template <int RS>
static int launch_split(hipStream_t s, const LkLevelArgs &a) {
    using CP = LkCfg<3, 512, 32>;
    static thread_local int done_dev = -1;
    static thread_local int n_cu = 256;
    int dev = 0;
    MICV_HIP(hipGetDevice(&dev));
    if (done_dev != dev) {
        MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lk_grad_kernel<3, 512, 32>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)CP::LDS_BYTES));
        MICV_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        done_dev = dev;
    }
    LkLevelArgs b = a;
    b.y_shift = 0;
    b.job.blocks = 0;
    b.pre_base = a.split == 2;  // experiment: the base flow through u, v instead of recomputed by the sums kernel
    dim3 grid(cdiv(a.cols, CP::TW) * cdiv(a.rows, CP::TH), a.batch);
    if (a.split == 3 && a.img_stride == a.cols && a.img_pair == (size_t)a.rows * a.cols) {
        // Variant A' (experiment): the pre-pass leaves only the warped image (4 B per pixel) and the base flow; the second
        // half is the fused level kernel in its no-flow mode on (prev, warped), adding the base from u, v.
        b.pre_warp = 1;
        b.grad_pair = (size_t)a.rows * a.cols;
        lk_grad_kernel<3, 512, 32><<<grid, CP::NT, CP::LDS_BYTES, s>>>(b);
        MICV_LAUNCH_CHECK();
        LkLevelArgs m = a;
        m.grad = nullptr;
        m.next = a.grad;
        m.mode = LK_FLOW_NONE;
        m.flow_u = m.flow_v = nullptr;
        m.flow_rows = m.flow_cols = 0;
        m.base_rmw = 1;
        m.job.blocks = 0;
        return launch_r<7, 512>(s, m);
    }
    lk_grad_kernel<3, 512, 32><<<grid, CP::NT, CP::LDS_BYTES, s>>>(b);
    MICV_LAUNCH_CHECK();
    LkSumsArgs m;
    m.grad = a.grad;
    m.grad_pair = a.grad_pair;
    m.gpitch = a.grad_pitch;
    m.rows = a.rows; m.cols = a.cols; m.batch = a.batch;
    lk_sums_partition(a.rows, a.cols, a.batch, 3 * n_cu, &m.strips, &m.segs, &m.seg_rows);
    m.out_u = a.out_u; m.out_v = a.out_v; m.out_stride = a.out_stride; m.out_pair = a.out_pair;
    m.base = a.add_base ? (b.pre_base ? 1 : 2) : 0;
    m.flow_u = a.flow_u; m.flow_v = a.flow_v; m.flow_rows = a.flow_rows; m.flow_cols = a.flow_cols; m.flow_pair = a.flow_pair;
    return launch_lk_sums(s, m, 2 * RS + 1);
}
bool lk_fused_supports_direct_levels(int win) { return win == 15; }

int launch_lk_level_fused(hipStream_t s, const LkLevelArgs &a_in) {
    LkLevelArgs a = a_in;
#ifdef MICV_DIAG
    static const int stop = [] { const char *e = getenv("MICV_LK_STOP"); return e ? atoi(e) : -1; }();
    a.stop_after = stop;
#else
    a.stamps = nullptr;
#endif
    // Strip launch (r05, MICV_OPT_LK_STRIP; off by default: -15 % instructions, the same time -- DESIGN.md section 5): the
    // interior as streamed strips, window 15
    if (a.strip > 0 && a.win == 15 && a.row_begin == 0 && a.row_end == a.rows && a.img_xstride == 1 && a.job.blocks == 0 &&
        a.stamps == nullptr && a.stop_after < 0 && a.mode == LK_FLOW_COARSE && a.rows == 2 * a.flow_rows && a.cols == 2 * a.flow_cols &&
        (a.img_stride & 3) == 0 && (a.img_pair & 3) == 0 && !a.narrow &&
        ((reinterpret_cast<uintptr_t>(a.prev) | reinterpret_cast<uintptr_t>(a.next)) & 15) == 0 &&
        a.strip > 0 && (!(a.strip & 8192) || (long)cdiv(a.cols, 64) * cdiv(a.rows, 32) * a.batch >= 4096)) {
        if (a.name_out) {
            snprintf(a.name_out, a.name_cap, "lk_level_strip_kernel<7, 512>");
            return MICV_OK;
        }
        const int rc = launch_strip<7>(s, a, a.strip & 8191);
        if (rc != MICV_EUNSUPPORTED) return rc;  // (no interior rectangle: the tile launch below)
    }
    // Split launch (r05): pre-pass + streaming sums, for whole-frame launches with a doubling coarse flow whose caller
    // provided the gradient planes
    if (a.grad && a.row_begin == 0 && a.row_end == a.rows && a.img_xstride == 1 && a.job.blocks == 0 && a.stamps == nullptr &&
        a.stop_after < 0 && a.mode == LK_FLOW_COARSE && a.rows == 2 * a.flow_rows && a.cols == 2 * a.flow_cols &&
        (a.img_stride & 3) == 0 && (a.img_pair & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(a.prev) | reinterpret_cast<uintptr_t>(a.next)) & 15) == 0 &&
        lk_split_wanted(a.rows, a.cols, a.batch, a.win, a.mode, a.split)) {
        const LkGradGeom gg = lk_grad_geom(a.rows, a.cols, a.win);
        if (gg.pad == a.grad_pad && gg.pitch == a.grad_pitch && gg.rows <= a.grad_rows && a.win == 15) {
            if (a.name_out) {
                snprintf(a.name_out, a.name_cap, a.split == 3 ? "lk_grad_kernel<3, 512, 32> + lk_level_kernel<7, 0, 512, 32, false, 64>"
                                                              : "lk_grad_kernel<3, 512, 32> + lk_sums_stream_kernel<7>");
                return MICV_OK;
            }
            return launch_split<7>(s, a);
        }
    }
    switch (a.win) {
        case 15: {
            // 512 threads per tile (4 waves per SIMD at 2 workgroups per CU) vs 256 (2 waves per SIMD)
            // Measured on MI355X (8 pairs of 1080p): 512 threads 0.377 ms per level-0 launch vs 0.406 ms,
            // and the latency-bound coarse levels gain more.  MICV_OPT_LK_NARROW_TILES selects the narrow form.
            if (a.narrow) return launch_r<7, 256>(s, a);
            // a launch of at most one round of 64x16 tiles costs one tile's latency: half-height tiles shorten
            // it (batch 1, 1080p: levels 1-4, 0.111 -> 0.096 ms per call)
            const long short_tiles = (long)cdiv(a.cols, 64) * (cdiv(a.row_end > 0 ? a.row_end : a.rows, 16)) * a.batch;
            const long short_limit = a.short_tiles > 0 ? a.short_tiles : 512;  // one round at two workgroups per CU
            if (short_tiles <= short_limit && a.short_tiles >= 0) return launch_r<7, 512, 16>(s, a);
            // MICV_OPT_LK_TALL_TILES: 64x64 tiles, 1024 threads, one workgroup per CU (131.6 KB of LDS): the
            // structural cut of the halo overhead (phases 0-3 on 80x80 for 64x64 = 1.56x instead of 1.875x,
            // row pass 78 rows for 64 = 1.22x instead of 1.44x).  Launches of at least four rounds only.
            if (a.tall_tiles == 1 && (long)cdiv(a.cols, 64) * cdiv(a.rows, 64) * a.batch >= 1024) return launch_r<7, 1024, 64>(s, a);
            // MICV_OPT_LK_TALL_TILES = 2: 32x64 tiles, 512 threads, two workgroups per CU (r04; LkCfg).  Launches of
            // at least two rounds, whole frames with a doubling coarse flow (the throughput levels).
            // MICV_OPT_LK_TALL_TILES = 3 (r04 experiment): 64x32 tiles worked by 1024 threads, two workgroups per CU = eight
            // waves per SIMD at a 64-VGPR budget (the occupancy lever that needs no third workgroup)
            if (a.tall_tiles == 3 && (long)cdiv(a.cols, 64) * cdiv(a.rows, 32) * a.batch >= 1024) return launch_r<7, 1024, 32>(s, a);
            if (a.tall_tiles == 2 && a.mode == LK_FLOW_COARSE && a.img_xstride == 1 && a.row_begin == 0 && a.row_end == a.rows &&
                (long)cdiv(a.cols, 32) * cdiv(a.rows, 64) * a.batch >= 1024)
                return launch_r<7, 512, 64, false, 32>(s, a);
            return launch_r<7, 512>(s, a);
        }
        case 7: return a.narrow ? launch_r<3, 256>(s, a) : launch_r<3, 512>(s, a);  // 512 threads: the staged / marching body
        case 21:  // the reference's default winSize (OpticalFlow.h:9,18): halo 12, 64x16 tiles fit two workgroups per CU
            // (64x32 tiles need 92 KB of LDS = one workgroup per CU: measured 17.6 Gpix/s against 18.6)
            if (a.narrow) return launch_r<10, 256>(s, a);
            // Launches of at least two rounds: 64x32 tiles worked by 1024 threads (2 output rows per thread, 92 KB
            // of LDS = one workgroup of 16 waves per CU) instead of two 512-thread 64x16 tiles per CU: the halo
            // overhead falls from 3.4x to 2.4x (phases 0-3) and from 2.25x to 1.6x (row pass), which at this
            // window outweighs what one workgroup per CU loses at its barriers (r03, 4 x 1080p: level 0
            // 236 -> 223 us; the same trade LOSES at window 15, see MICV_OPT_LK_TALL_TILES).  -1 = never.
            // r04: 64x64 tiles, 1024 threads (149 KB of LDS, still one workgroup of 16 waves per CU): region 1.89x and
            // row pass 1.31x instead of 2.4x / 1.6x at the same occupancy.  MICV_OPT_LK_TALL_TILES = 1 keeps 64x32.
            if (a.tall_tiles == 0 && (long)cdiv(a.cols, 64) * cdiv(a.rows, 64) * a.batch >= 256) return launch_r<10, 1024, 64>(s, a);
            if (a.tall_tiles >= 0 && (long)cdiv(a.cols, 64) * cdiv(a.rows, 32) * a.batch >= 512) return launch_r<10, 1024, 32>(s, a);
            return launch_r<10, 512, 16>(s, a);
        case 11: return a.narrow ? launch_r<5, 256>(s, a) : launch_r<5, 512>(s, a);
        default:
            set_error("lk fused: window %d has no tiled instantiation", a.win);
            return MICV_EUNSUPPORTED;
    }
}

}  // namespace micv
