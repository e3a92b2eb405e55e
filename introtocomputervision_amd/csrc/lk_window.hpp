// lk_window.hpp -- the window-sum machinery of the LK level kernels (phase 4 of lk_fused.hip, the whole of
// lk_split.hip): tile geometry (LkCfg), the skewed packed row / column chains, the hand-placed column-pass loads.
// Device code shared by the fused level kernel and the split (pre-pass + streaming sums) form, so that both run
// the same fmaf chains: same bits.
#pragma once
#include "lk_fused.hpp"

#include <utility>

#include "lk_device.hpp"

namespace micv {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;

template <int N>
struct TapsN {
    float k[N];
};

// TH_ = 32 is the throughput tile; TH_ = 16 (512 threads, 2 output rows per thread) halves every
// thread's share of every phase: the form for launches that do not fill the GPU once, where the
// time of a level is one tile's latency.
// TW_ = 32 (r04): the 32x64 tile -- the same 3 840-pixel region and LDS as 64x32, but the row pass runs 78 gradient
// rows for 64 output rows (1.22x) instead of 46 for 32 (1.44x): 10 wave-jobs per sweep instead of 12.
template <int R_, int NT_ = 256, int TH_ = 32, int TW_ = 64>
struct LkCfg {
    static constexpr int R = R_;
    static constexpr int W = 2 * R + 1;
    static constexpr int TW = TW_, TH = TH_, NT = NT_;
    static_assert(TW == 64 || (TW == 32 && R_ == 7 && NT_ == 512 && TH_ == 64), "32-wide tiles: window 15, 512 threads, 64 rows");
    // image halo: Sobel + window = R + 1, rounded up to a multiple of 4 so that region rows are whole
    // 16-byte chunks (LDS-DMA) and the halo bands split into whole marching jobs at every window
    static constexpr int H = (R + 1 + 3) & ~3;
    static constexpr int RW = TW + 2 * H, RH = TH + 2 * H;  // image region
    static constexpr int PS = RW;                           // LDS row stride of P / Wp
    static constexpr int GW = TW + 2 * R, GH = TH + 2 * R;  // gradient region
    // The three gradient planes are interleaved by ROW: row q of Ix, Iy, It sit side by side
    // (GP floats each, 16-B aligned), GS floats per row.  Rows [0, k) of all three planes are then one
    // contiguous block at the start of the area -- what a tile chain carries over (below).  With
    // GP/4 = 4 (mod 16) (GP = 80 for R = 7) GS/4 = 12 = -4 (mod 16): the row pass's ds_read_b128
    // pattern (4 rows x 16 groups per wave) is conflict-free, as it is with separate planes.
    static constexpr int GP = (GW + 3) & ~3;
    static constexpr int GS = 3 * GP;
    static constexpr int WV = (4 + 2 * R + 3) / 4;          // float4 loads per row-pass window
    static_assert(4 * (TW / 4 - 1) + 4 * WV <= GP, "row-pass window reads stay inside a plane row");
    static constexpr int RBS = TW;                          // row-buffer floats per row (XOR-swizzled chunks; rb_off)
    static_assert(TW == 64 || (TH + 2 * R_) % 2 == 0, "32-wide row buffers hold two rows per 64-float unit");
    static constexpr int CW = RW / 2 + 3, CH = RH / 2 + 3;  // coarse flow block
    static constexpr int M = 8;                             // margin of the staged `next` window
    static constexpr int NW = RW + 2 * M, NH = RH + 2 * M;
    static constexpr int RPT = TH / (NT / TW);  // output rows per thread: 8 (256 threads) or 4 (512)
    static constexpr int ROWBUF_F = 3 * GH * RBS;
    static constexpr int IMG_F = (2 * RH * PS > ROWBUF_F ? 2 * RH * PS : ROWBUF_F);
    static constexpr int C_F = (2 * CH * CW + 3) & ~3;         // coarse block, both fields, 16-B padded
    // streamed tiles: rows of u and v interleaved, each padded to whole float4s (dma_coarse)
    static constexpr int CWP = (CW + 3) & ~3, CS_F = 2 * CH * CWP;
    static constexpr int FLOW_F = C_F + 2 * CH * RW;           // border tiles: C + R
    static constexpr int STAGE_F = C_F + NW * NH;              // interior tiles: C + next window
    static constexpr int GRAD_F = GH * GS;
    // Tile chains (vertically adjacent tiles run by one workgroup): the lower tile keeps the upper
    // tile's last QC gradient rows (its own first QC rows) and computes phases 0-3 only for region
    // rows [LYC, RH).  CARRY_F floats at the start of the gradient area hold them; the lower tile's
    // staging (coarse block + `next` window, both QC.. rows shorter) goes behind them.
    static constexpr int QC = GH - TH;                       // carried gradient rows (2R)
    static constexpr int LYC = QC + (H - R) - 1;             // first region row a carry tile needs
    static constexpr int CARRY_F = QC * GS;
    static constexpr int CHC = (RH - H) / 2 + 3;             // coarse rows of a carry tile: base flow from its own first row on
    static constexpr int NHC = RH - LYC + 2 * M;             // `next` window rows of a carry tile
    static constexpr int CC_F = (2 * CHC * CW + 3) & ~3;
    static constexpr bool CHAIN_OK = (RH * (RW / 4)) % 64 == 0 && NT_ == 512 && TH_ == 32 && (LYC % 2 == 0) &&
                                     CARRY_F + CC_F + NW * NHC <= (GRAD_F > STAGE_F ? GRAD_F : STAGE_F);
    static constexpr int X_F = (FLOW_F > GRAD_F ? FLOW_F : GRAD_F) > STAGE_F
                                   ? (FLOW_F > GRAD_F ? FLOW_F : GRAD_F)
                                   : STAGE_F;
    static_assert(RPT == 8 || RPT == 4 || RPT == 2, "256 or 512 threads per 64x32 tile, 512 per 64x16 tile");
    static constexpr int LDS_FLOATS = IMG_F + X_F;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * 4;
    // The marching body of phase 2 is written for 8-row segments and 16-B rows.
    static constexpr bool FAST = (H % RPT == 0) && (RW % 4 == 0);
};

// Second argument of __launch_bounds__ (waves per SIMD the register allocation must allow): 512-thread
// tiles run two workgroups per CU, the 1024-thread 64x64 tile one -- four waves per SIMD either way.
constexpr int lk_waves_per_simd(int nt) { return nt >= 1024 ? 4 : nt / 128; }
// The plain kernel's bound also knows the window and the tile: window 15 on a 64x32 tile worked by 1024 threads (r04
// experiment, MICV_OPT_LK_TALL_TILES = 3) keeps the 80 KB of LDS, so TWO such workgroups fit a CU = 8 waves per SIMD,
// and the register allocation must allow that (64 VGPRs).
constexpr int lk_waves_per_simd_rt(int r, int nt, int th) { return (r == 7 && nt == 1024 && th == 32) ? 8 : lk_waves_per_simd(nt); }

// ---- phase 4 building blocks ------------------------------------------------------------------

// Row-buffer addressing: row q, 16-byte chunk ch lives at chunk (ch ^ 2*(q&3)).  The row
// pass's b128 stores (4 rows x 2 chunks per 8-lane group) and the column pass's b32 loads are
// both bank-conflict-free with this layout at a 64-float row pitch.
__device__ __forceinline__ int rb_off(int q, int chunk) { return q * 64 + 4 * (chunk ^ (2 * (q & 3))); }
// 32-wide tiles: two rows share a 64-float unit, row q in half (q ^ (q >> 2)) & 1 -- rows q and q + 4 (the two row
// groups a wave's lanes 0-31 / 32-63 read in the column pass) then sit in different halves = different banks -- and
// the 8 chunks of a row are XOR-ed with 4 ((q >> 1) & 1), which keeps the row pass's b128 stores (a quarter wave =
// 4 rows x 4 chunks) on 16 different 16-byte slots.  Rows q and q + 8 differ in the unit only.
__device__ __forceinline__ int rb_off32(int q, int chunk) {
    return (q >> 1) * 64 + 32 * ((q ^ (q >> 2)) & 1) + 4 * (chunk ^ (4 * ((q >> 1) & 1)));
}

template <typename C>
__device__ __forceinline__ void load_window(const float *__restrict__ A, int qy, int c0,
                                            float (&w)[4 * C::WV]) {
    // Native vector type on purpose: HIP's float4 is a struct, its copy decays into scalar loads
    // that the SLP vectoriser re-pairs as <2 x float> align 4 -> ds_read2_b64, whose 32-bank,
    // 16-contiguous-lane banking makes this access pattern 2-way conflicted (measured: 52 M of the
    // launch's 57 M LDS conflict cycles).  <4 x float> align 16 -> ds_read_b128, conflict-free here.
    const v4f *a4 = reinterpret_cast<const v4f *>(A + qy * C::GS + c0);
#pragma unroll
    for (int i = 0; i < C::WV; i++) {
        const v4f v = a4[i];
        w[4 * i + 0] = v.x;
        w[4 * i + 1] = v.y;
        w[4 * i + 2] = v.z;
        w[4 * i + 3] = v.w;
    }
}

// Four adjacent outputs of the (2R+1)-tap row pass of the product field a*b.
template <typename C>
__device__ __forceinline__ void row_taps(const float (&a)[4 * C::WV], const float (&b)[4 * C::WV],
                                         const TapsN<C::W> &g, float *__restrict__ out) {
    float p[4 + 2 * C::R];
#pragma unroll
    for (int i = 0; i < 4 + 2 * C::R; i++) p[i] = a[i] * b[i];
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < C::W; k++) acc = fmaf(p[j + k], g.k[k], acc);
        o[j] = acc;
    }
    *reinterpret_cast<float4 *>(out) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- the row pass with every FMA packed and no operand shuffles ---------------------------------
// v_pk_fma_f32 wants its operands as aligned register pairs.  Pairing outputs (j, j+1) tap by tap
// needs (p[j+k], p[j+k+1]), which is an aligned pair only for even j+k: the compiler fills the odd
// half with ~54 v_mov / v_pk_mov and 9 duplicate v_pk_mul per 4-output job.  Skewing the pair by one
// tap removes them: at step s output j takes tap s and output j+1 takes tap s-1, so BOTH lanes
// multiply the SAME product p[j+s] (op_sel broadcasts one half of an aligned pair) and the taps come
// as the pair (g[s-1], g[s]) from SGPRs.  Every output still runs its own fmaf chain over taps
// 0..2R in order, so the bits are those of row_taps().  Steps 0 and 2R+1 touch one lane only.
typedef float v2f __attribute__((ext_vector_type(2)));

template <int SEL>
__device__ __forceinline__ void pk_fma_skew(v2f &acc, v2f p, v2f gpair) {
    // lanes: acc.x += p[SEL] * gpair.y (tap s);  acc.y += p[SEL] * gpair.x (tap s - 1)
    if (SEL == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "v"(p), "s"(gpair));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(p), "s"(gpair));
}

// Where value I of a chain's window lives: register pair PAIR, half HALF.  The skewed step picks one half of
// one aligned pair by op_sel, so ANY fixed placement works -- the chains are indifferent to which two values
// share a register pair.  SlotAdjacent: (I, I+1) together, what a ds_read_b128 of a row delivers (row pass).
// SlotStride4: (I, I+4) together, what ONE ds_read2st64_b32 delivers from the XOR-swizzled row buffers
// (rows q and q+4 share a swizzle class, so one base register + two immediate offsets reach both): the
// column pass then needs no v_mov to assemble its pairs (r03: 33 of the 219 instructions of a column pass
// were such moves).
struct SlotAdjacent {
    static constexpr int pair(int i) { return i >> 1; }
    static constexpr int half(int i) { return i & 1; }
};
struct SlotStride4 {
    static constexpr int pair(int i) { return (i >> 3) * 4 + (i & 3); }
    static constexpr int half(int i) { return (i >> 2) & 1; }
};

template <typename SL, int W, int NP, int J, int S>
__device__ __forceinline__ void skew_step(v2f &acc, const v2f (&P)[NP], const TapsN<W> &g) {
    constexpr int I = J + S;  // index of the value both lanes use at this step
    constexpr int PI = SL::pair(I), H = SL::half(I);
    static_assert(PI < NP, "skewed chain reads inside its window");
    const float pv = H ? P[PI].y : P[PI].x;
    if (S == 0) {
        acc.x = fmaf(pv, g.k[0], 0.f);  // output J, tap 0; output J+1 has not started
        acc.y = 0.f;
    } else if (S == W) {
        acc.y = fmaf(pv, g.k[W - 1], acc.y);  // output J+1, last tap; output J is complete
    } else {
        const v2f gp = {g.k[S - 1], g.k[S]};
        pk_fma_skew<H>(acc, P[PI], gp);
    }
}

template <typename SL, int W, int NP, int J, int... S>
__device__ __forceinline__ void skew_chain(v2f &acc, const v2f (&P)[NP], const TapsN<W> &g,
                                           std::integer_sequence<int, S...>) {
    (skew_step<SL, W, NP, J, S>(acc, P, g), ...);
}

// Four adjacent outputs of the row pass of the product field a*b, windows given as aligned pairs.
template <typename C>
__device__ __forceinline__ void row_taps_skew(const v2f (&a)[C::WV * 2], const v2f (&b)[C::WV * 2],
                                              const TapsN<C::W> &g, float *__restrict__ out) {
    v2f P[C::WV * 2];
#pragma unroll
    for (int i = 0; i < C::WV * 2; i++) P[i] = a[i] * b[i];  // v_pk_mul_f32, each product once
    v2f acc0, acc1;
    skew_chain<SlotAdjacent, C::W, C::WV * 2, 0>(acc0, P, g, std::make_integer_sequence<int, C::W + 1>{});
    skew_chain<SlotAdjacent, C::W, C::WV * 2, 2>(acc1, P, g, std::make_integer_sequence<int, C::W + 1>{});
    *reinterpret_cast<float4 *>(out) = make_float4(acc0.x, acc0.y, acc1.x, acc1.y);
}

template <typename C>
__device__ __forceinline__ void load_window_pairs(const float *__restrict__ A, int qy, int c0,
                                                  v2f (&w)[C::WV * 2]) {
    const v4f *a4 = reinterpret_cast<const v4f *>(A + qy * C::GS + c0);
    // four outputs read W + 3 values: when that leaves the last pair unused (windows 15 and 43) the last load is a
    // b64.  As a b128 its two dead registers were handed to the address arithmetic that follows the loads, and that
    // write-after-write made hipcc drain all ten loads (lgkmcnt(0)) before the first product (r04).
    constexpr int USED_PAIRS = (C::W + 3 + 1) / 2;
#pragma unroll
    for (int i = 0; i < C::WV; i++) {
        if (2 * i + 1 < USED_PAIRS) {
            const v4f v = a4[i];
            w[2 * i + 0] = (v2f){v.x, v.y};
            w[2 * i + 1] = (v2f){v.z, v.w};
        } else {
            w[2 * i + 0] = *reinterpret_cast<const v2f *>(a4 + i);
            w[2 * i + 1] = (v2f){0.f, 0.f};
        }
    }
}

// Column pass: thread (c, r0) produces RPT vertically adjacent window sums from one row buffer.
// The outputs go in pairs (j, j + 1) through the same skewed packed chain as the row pass (r03): at
// step s output j takes tap s and output j + 1 tap s - 1, both on the staged value v[j + s], so every
// FMA of the pass is one lane of a v_pk_fma_f32 and each output still runs its own chain over taps
// 0..2R in order -- the bits of the scalar loop it replaces (4 x 15 v_fma_f32 -> 2 x 16 packed).
template <typename C, int... PJ>
__device__ __forceinline__ void col_pairs(const v2f (&V)[((C::RPT + 2 * C::R + 7) / 8) * 4], float (&S)[C::RPT],
                                          const TapsN<C::W> &g, std::integer_sequence<int, PJ...>) {
    constexpr int NP = ((C::RPT + 2 * C::R + 7) / 8) * 4;
    v2f acc[C::RPT / 2];
    (skew_chain<SlotStride4, C::W, NP, 2 * PJ>(acc[PJ], V, g, std::make_integer_sequence<int, C::W + 1>{}), ...);
#pragma unroll
    for (int i = 0; i < C::RPT / 2; i++) {
        S[2 * i] = acc[i].x;
        S[2 * i + 1] = acc[i].y;
    }
}

// The staged values of one column pass, read with hand-placed ds_read2st64_b32: pair p = (row I0, row I0 + 4)
// of field F's row buffer, I0 = 8 (p / 4) + p % 4.  Rows of one swizzle class (q mod 4) share the lane part of
// their address, and a row is exactly one 64-dword unit of the instruction's offsets, so FOUR address
// registers (cls[k], computed once per tile for all five fields) and two immediates reach every cell of all
// three row buffers -- 9 loads per field, no address arithmetic, no v_mov.  (Left to the compiler the loads
// of consecutive fields get merged across row buffers -- same cell, 46 rows apart -- and every chain value
// then costs a v_mov to reach its pair: 33 of a pass's 219 instructions, r03.)  The loads are asm the
// compiler does not count: the wait statement names every destination, so nothing reads them early, and
// the "memory" clobber keeps them behind the barrier and the row-pass stores they depend on.
template <typename C, int F, int P>
__device__ __forceinline__ void col_load_pair(v2f &dst, const int (&cls)[4]) {
    constexpr int NV = C::RPT + 2 * C::R;
    constexpr int I0 = 8 * (P / 4) + (P & 3), I1 = I0 + 4, O0 = I0 + F * C::GH, O1 = I1 + F * C::GH;
    static_assert(C::RBS == 64 && O1 < 256, "one row = one 64-dword unit; offsets are 8 bits");
    // A pair whose second row lies past the window (I1 >= NV) is loaded whole all the same: its upper half is
    // never used by a chain, the address stays inside the kernel's LDS (at most 2 GH + TH + 2R + 4 rows from
    // rb0), and the alternative -- a single ds_read_b32 into a float that is then packed into the pair -- would
    // make the compiler copy the asm's destination BEFORE the wait below (hipcc counts an asm output as written
    // at the end of the statement): stale data, timing-dependent.
    if constexpr (I0 >= NV) {
        dst = (v2f){0.f, 0.f};
    } else {
        static_assert((2 * C::GH + C::TH + 2 * C::R + 8) * C::RBS <= C::LDS_FLOATS, "the spare half-pair reads stay inside LDS");
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(cls[P & 3]), "i"(O0), "i"(O1) : "memory");
    }
}

template <typename C, int F, int... PS>
__device__ __forceinline__ void col_load_all(v2f (&V)[sizeof...(PS)], const int (&cls)[4], std::integer_sequence<int, PS...>) {
    (col_load_pair<C, F, PS>(V[PS], cls), ...);
}

// s_waitcnt for hand-placed LDS loads, naming their destinations (four per statement; a wait on a drained
// counter costs nothing), so the compiler places every use -- and every copy -- of them behind it.
template <int N, int M>
__device__ __forceinline__ void lds_wait_n(v2f (&V)[M]) {
    static_assert(N <= M);
#pragma unroll
    for (int i = 0; i + 4 <= N; i += 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[i]), "+v"(V[i + 1]), "+v"(V[i + 2]), "+v"(V[i + 3]));
    constexpr int T = N & ~3;
    if constexpr (N % 4 == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[T]), "+v"(V[T + 1]), "+v"(V[T + 2]));
    if constexpr (N % 4 == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[T]), "+v"(V[T + 1]));
    if constexpr (N % 4 == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[T]));
}

// The march's coarse block (dma_coarse layout: rows of u and v interleaved, CWP floats each): the (u, v) pair
// of coarse column k is ds_read2_b32 offset0:k offset1:k + CWP.  Left to itself the compiler pairs ADJACENT
// columns instead -- (u[0], u[1]) and (v[0], v[1]) -- and transposes with three v_mov per coarse row.  Same
// discipline as the column pass: whole pairs, "memory" clobber, one wait naming every destination.  Rows come
// three to a base register (offsets are 8 bits: 2 * 2 CWP + CWP + 2 < 256).
template <typename C, int I>
__device__ __forceinline__ void coarse_row_load(v2f *cc, const int (&a)[2], const int (&b)[2]) {
    constexpr int CWP = C::CWP, O = (I % 3) * 2 * CWP;
    static_assert(O + CWP + 2 < 256 && I < 6, "offsets are 8 bits, two base registers");
#define MICV_RD2(dst, addr, o0, o1) \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "i"(o0), "i"(o1) : "memory")
    MICV_RD2(cc[5 * I + 0], a[I / 3], O, O + CWP);
    MICV_RD2(cc[5 * I + 1], a[I / 3], O + 1, O + CWP + 1);
    MICV_RD2(cc[5 * I + 2], a[I / 3], O + 2, O + CWP + 2);
    MICV_RD2(cc[5 * I + 3], b[I / 3], O, O + CWP);
    MICV_RD2(cc[5 * I + 4], b[I / 3], O + 1, O + CWP + 1);
#undef MICV_RD2
}

template <typename C, int... IS>
__device__ __forceinline__ void coarse_block_load(v2f (&cc)[5 * sizeof...(IS)], const float *c, int odd,
                                                  std::integer_sequence<int, IS...>) {
    typedef const __attribute__((address_space(3))) float lds_cfloat;
    constexpr int NR = sizeof...(IS);
    int a[2], b[2];
    a[0] = (int)(size_t)(lds_cfloat *)c;
    b[0] = a[0] + 4 * odd;
    a[1] = NR > 3 ? a[0] + 4 * 3 * 2 * C::CWP : a[0];
    b[1] = NR > 3 ? b[0] + 4 * 3 * 2 * C::CWP : b[0];
    (coarse_row_load<C, IS>(cc, a, b), ...);
    lds_wait_n<5 * NR>(cc);
}

// cls[k] = LDS byte address, in row buffer 0, of column c's cell in row r0 -- with the chunk swizzle of the
// rows r0 + I, I = k (mod 4): rb_off() swizzles by the row's own index mod 4, and r0 is a multiple of RPT
// only (RPT = 2: 64x16 tiles and the 1024-thread window-21 tiles), so the class of staged row I is
// (r0 + k) & 3, not k.  Row r0 + I is then cls[I & 3] + I rows, the rows being the immediates of the loads.
template <typename C>
__device__ __forceinline__ void col_bases(const float *rb0, int c, int r0, int (&cls)[4]) {
    typedef const __attribute__((address_space(3))) float lds_cfloat;
    const int base = (int)(size_t)(lds_cfloat *)rb0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        cls[k] = base + 4 * (r0 * C::RBS + 4 * ((c >> 2) ^ (2 * ((r0 + k) & 3))) + (c & 3));
}

// [A, B) as an integer_sequence
template <int A, int... I>
constexpr std::integer_sequence<int, (A + I)...> offset_seq(std::integer_sequence<int, I...>) { return {}; }
template <int A, int B>
using range_seq = decltype(offset_seq<A>(std::make_integer_sequence<int, B - A>{}));

// s_waitcnt lgkmcnt(N) naming up to four destinations (N = LDS operations that may still be outstanding)
template <int N>
__device__ __forceinline__ void lds_wait_upto4(v2f &a, v2f &b, v2f &c, v2f &d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

#ifndef MICV_LK_COL_FULLWAIT
// Column pass with PARTIAL waits (r04; -DMICV_LK_COL_FULLWAIT builds the single-wait form for A/B:
// level-0 launch 199.0 -> 197.1 us in every one of four interleaved rounds, profiles/r04/lk_ab.txt): the loads return in order, pairs 0-3 hold
// staged rows 0..7, pairs 4-7 rows 8..15, pairs 8-9 rows 16..21.  The first chain's first eight steps need rows
// 0..7 only, so they start when four loads have landed instead of ten; the rest follows the second and third wait.
// Window 15 at four outputs per thread only (the 64x32 tile); every output still runs its chain over taps 0..2R in
// order -- same bits.
template <typename C, int F>
__device__ __forceinline__ void col_pass_partial(const int (&cls)[4], float (&S)[C::RPT], const TapsN<C::W> &g) {
    constexpr int NV = C::RPT + 2 * C::R, NP = ((NV + 7) / 8) * 4, W = C::W;
    static_assert(C::RPT == 4 && C::R == 7 && NP == 12, "written for the 64x32 window-15 tile");
    v2f V[NP];
    // scalar loads share the counter and return out of order: none may be outstanding while partial counts are used
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    col_load_all<C, F>(V, cls, std::make_integer_sequence<int, NP>{});  // 10 loads issued (pairs 10, 11 are constants)
    v2f acc0, acc1;
    lds_wait_upto4<6>(V[0], V[1], V[2], V[3]);
    skew_chain<SlotStride4, W, NP, 0>(acc0, V, g, range_seq<0, 8>{});
    // the waits are volatile but the FMAs are not: without a fence the scheduler sinks the first chain's steps below
    // the later waits (ISA: lgkmcnt(6), four FMAs, lgkmcnt(2), lgkmcnt(0), then everything else; with the fences the
    // level-0 launch is another 0.65 % shorter in every round of the A/B, profiles/r04/lk_ab.txt)
    __builtin_amdgcn_sched_barrier(0);
    lds_wait_upto4<2>(V[4], V[5], V[6], V[7]);
    skew_chain<SlotStride4, W, NP, 0>(acc0, V, g, range_seq<8, W + 1>{});
    skew_chain<SlotStride4, W, NP, 2>(acc1, V, g, range_seq<0, 14>{});
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[8]), "+v"(V[9]));
    skew_chain<SlotStride4, W, NP, 2>(acc1, V, g, range_seq<14, W + 1>{});
    S[0] = acc0.x;
    S[1] = acc0.y;
    S[2] = acc1.x;
    S[3] = acc1.y;
}
#endif

// PARTIAL (lk_tile's PARTIAL_COLS, set by lk_level_kernel only): only where the compiler keeps scalar loads out of the
// counted window -- the plain kernel; in the chain /
// streamed kernels it re-loads the taps there under SGPR pressure (tools/audit_asm_loads.py finds such loads and
// fails the build's test), and a scalar load returning out of order would break a partial count.
template <typename C, int F, bool PARTIAL = false>
__device__ __forceinline__ void col_pass(const int (&cls)[4], float (&S)[C::RPT], const TapsN<C::W> &g) {
#ifndef MICV_LK_COL_FULLWAIT
    if constexpr (PARTIAL && C::RPT == 4 && C::R == 7) {
        col_pass_partial<C, F>(cls, S, g);
        return;
    }
#endif
    constexpr int NV = C::RPT + 2 * C::R, NP = ((NV + 7) / 8) * 4;
    static_assert(C::RPT % 2 == 0, "column outputs in pairs");
    // pairs wholly past the window (a suffix of V) are constants: keep them out of the wait, which would make
    // the compiler materialise them in registers
    constexpr int NPL = NP - (8 * ((NP - 1) / 4) + ((NP - 1) & 3) >= NV) - (8 * ((NP - 2) / 4) + ((NP - 2) & 3) >= NV) -
                        (8 * ((NP - 3) / 4) + ((NP - 3) & 3) >= NV);
    v2f V[NP];
    col_load_all<C, F>(V, cls, std::make_integer_sequence<int, NP>{});
    lds_wait_n<NPL>(V);
    col_pairs<C>(V, S, g, std::make_integer_sequence<int, C::RPT / 2>{});
}

// ---- the column pass of the 32-wide tile ---------------------------------------------------------------------
// Thread (c, r0 = 4 k) reads staged rows r0 + I, I = 0..17.  rb_off32: unit (r0 + I) >> 1 = 2 k + (I >> 1), and the
// half and the chunk swizzle depend on I & 7 (and on k's parity) only -- so rows I and I + 8 share an address
// register and differ by four units: ONE ds_read2st64_b32 delivers the pair (I, I + 8), eight of them rows 0..15;
// rows 16 and 17 come as two more (both halves the same row).  Eight address registers per thread and tile.
struct SlotStride8 {
    static constexpr int pair(int i) { return i < 16 ? (i & 7) : 8 + (i - 16); }
    static constexpr int half(int i) { return i < 16 ? (i >> 3) : 0; }
};

template <typename C>
__device__ __forceinline__ void col_bases32(const float *rb0, int c, int r0, int (&cls)[8]) {
    typedef const __attribute__((address_space(3))) float lds_cfloat;
    const int base = (int)(size_t)(lds_cfloat *)rb0;
    const int k = r0 >> 2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int half = ((i & 1) ^ ((k + (i >> 2)) & 1));
        cls[i] = base + 4 * ((r0 >> 1) * 64 + 32 * half + 4 * ((c >> 2) ^ (4 * ((i >> 1) & 1))) + (c & 3));
    }
}

template <typename C, int F, bool PARTIAL>
__device__ __forceinline__ void col_pass32(const int (&cls)[8], float (&S)[C::RPT], const TapsN<C::W> &g) {
    constexpr int W = C::W, NP = 10, FU = F * (C::GH / 2);  // field offset in 64-float units
    static_assert(C::RPT == 4 && C::R == 7 && C::TW == 32 && FU + 8 + 4 < 256, "32x64 tile, window 15");
    v2f V[NP];
    if (PARTIAL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // no scalar load may share the counted window
#define MICV_RD(P, I0, I1) \
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(V[P]) : "v"(cls[(I0) & 7]), "n"(FU + ((I0) >> 1)), "n"(FU + ((I1) >> 1)) : "memory")
    MICV_RD(0, 0, 8); MICV_RD(1, 1, 9); MICV_RD(2, 2, 10); MICV_RD(3, 3, 11);
    MICV_RD(4, 4, 12); MICV_RD(5, 5, 13); MICV_RD(6, 6, 14); MICV_RD(7, 7, 15);
    MICV_RD(8, 16, 16); MICV_RD(9, 17, 17);
#undef MICV_RD
    v2f acc0, acc1;
    if (PARTIAL) {
        // rows 0..15 have landed when two loads are still out: the first chain whole, the second up to its row 15
        lds_wait_upto4<2>(V[0], V[1], V[2], V[3]);
        lds_wait_upto4<2>(V[4], V[5], V[6], V[7]);
        skew_chain<SlotStride8, W, NP, 0>(acc0, V, g, std::make_integer_sequence<int, W + 1>{});
        skew_chain<SlotStride8, W, NP, 2>(acc1, V, g, range_seq<0, 14>{});
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[8]), "+v"(V[9]));
        skew_chain<SlotStride8, W, NP, 2>(acc1, V, g, range_seq<14, W + 1>{});
    } else {
        lds_wait_n<NP>(V);
        skew_chain<SlotStride8, W, NP, 0>(acc0, V, g, std::make_integer_sequence<int, W + 1>{});
        skew_chain<SlotStride8, W, NP, 2>(acc1, V, g, std::make_integer_sequence<int, W + 1>{});
    }
    S[0] = acc0.x;
    S[1] = acc0.y;
    S[2] = acc1.x;
    S[3] = acc1.y;
}

}  // namespace micv
