// lk_strip.hpp -- the INTERIOR of a pyramid level as vertical strips that a workgroup streams down (r05).
//
// The fused 64x32 tile (lk_fused.hip) recomputes pyrUp + warp + Sobel on a region 1.875x its outputs and runs the row
// pass of the window sums on 46 rows for 32.  Every other tile shape lost to LDS capacity or barriers, and cutting the
// launch in two lost to HBM.  This form keeps everything in ONE kernel and removes the vertical
// halo instead: a workgroup owns a 64-column strip of a row segment and walks down it in blocks of 16 rows, CARRYING
//   * the last 2R row-pass rows of all five product fields (17.9 KB; moved to the front of the field buffers), and
//   * the last two rows of the warped image (the Sobel ring of the next block),
// so that per block of 16 x 64 outputs it warps 16 x 80 pixels (1.25x), differentiates 16 x 78 (1.22x) and row-passes
// 16 rows (1.0x).  Only INTERIOR tiles are streamed -- every pixel, halo, margin and coarse tap of a strip lies inside
// the image, so this body has no border case at all; the border tiles of the same launch run the tile body
// (lk_level_strip_kernel in lk_fused.hip).  Same chains as the tile body: pyrUp / warp (the march), Sobel pairs,
// skewed packed row and column passes, double-precision solve -- same bits, tests/test_lk_gpu.py compares.
//
// Included by lk_fused.hip INSIDE namespace micv, after its LDS-DMA helpers (dma_rows, dma_coarse).
#pragma once

template <int R_>
struct StripCfg {
    static constexpr int R = R_, W = 2 * R + 1;
    static constexpr int TW = 64, B = 16, NT = 512, TH = B;
    static constexpr int RPT = B / (NT / TW);  // 2 output rows per thread in the column pass and the solve
    static constexpr int MR = 4;               // rows of a marching job (pyrUp pairs fine rows; 4 | B)
    static constexpr int H = (R + 1 + 3) & ~3;  // column halo of the image region (Sobel + window, whole float4s)
    static constexpr int RW = TW + 2 * H, PS = RW;
    static constexpr int GW = TW + 2 * R, GP = (GW + 3) & ~3, GS = 3 * GP;
    static constexpr int WV = (4 + 2 * R + 3) / 4;
    static_assert(4 * (TW / 4 - 1) + 4 * WV <= GP, "row-pass window reads stay inside a plane row");
    static constexpr int QC = 2 * R, GH = B + QC, RBS = 64, NF = 5;
    static_assert(QC <= B && B % MR == 0 && H - R == 1, "written for R + 1 = H (windows 7, 15)");
    static constexpr int PR = B + 2;            // rows of the prev / warped buffers: one Sobel ring row either side
    static constexpr int M = 8;                 // margin of the staged `next` window
    static constexpr int NW = RW + 2 * M, NH = B + 2 * M;
    static constexpr int CH = (B + 8) / 2 + 2;  // coarse rows: fine rows y .. y + B + 7 (own outputs and the new warped rows)
    static constexpr int CW = RW / 2 + 2, CWP = (CW + 3) & ~3, CS_F = 2 * CH * CWP;
    // LDS (floats): five field buffers | prev rows | warped rows | gradient block (aliases the `next` window) | 2 coarse blocks
    static constexpr int FIELD_F = NF * GH * RBS;
    static constexpr int P_F = PR * PS, W_F = PR * PS;
    static constexpr int G_F = B * GS, N_F = NW * NH;
    static constexpr int X_F = G_F > N_F ? G_F : N_F;
    static constexpr int LDS_FLOATS = FIELD_F + P_F + W_F + X_F + 2 * CS_F;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * 4;
    static_assert(((NF - 1) * GH + (B - RPT) + RPT + 2 * R + 8) * RBS <= LDS_FLOATS, "spare half-pair reads stay inside LDS");
};

// One block's staging, by LDS-DMA: the `next` window (rows y .. y + NH - 1 of the level = the new warped rows y + 8 ..
// y + 23 with the 8-row margin), the prev rows y + 6 .. y + 23, the coarse block.  No wait, no barrier.
template <typename C>
__device__ __forceinline__ void strip_stage(const LkLevelArgs &a, const float *__restrict__ prev, const float *__restrict__ next,
                                            int pair, int x0, int y, float *P, float *Nx, float *Cf, int tid) {
    dma_rows<C::NT, C::NW / 4, C::NH>(next + (size_t)y * a.img_stride + (x0 - C::H - C::M), a.img_stride, Nx, tid);
    dma_rows<C::NT, C::RW / 4, C::PR>(prev + (size_t)(y + C::R - 1) * a.img_stride + (x0 - C::H), a.img_stride, P, tid);
    dma_coarse<C>(a, pair, (x0 - C::H) / 2 - 1, y / 2 - 1, Cf, tid);
}

// pyrUp of the coarse flow at column gx, MRV fine rows from gy0 (even), both fields as the lanes of packed ops -- the
// interior case of the tile body's march (lk_fused.hip): row taps of fine column gx read coarse columns
// {m-1, m-1, m, m, m+1} (gx = 2m) or {m-1, m, m, m+1, m+1}; the column taps likewise.  huv[j] = the value before the
// "* 2" of OpticalFlow.cpp:142,144 (the warp takes it with a factor 64), buv[j] = after.
template <typename C, int MRV>
__device__ __forceinline__ void strip_pyrup(const float *Cf, int cx0, int cy0, int gx, int gy0, v2f (&huv)[MRV], v2f (&buv)[MRV]) {
    constexpr int NR = MRV / 2 + 2;
    const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};  // Pyramids.cu:19
    const int cyb = ((gy0 >> 1) - 1) - cy0, ccb = ((gx >> 1) - 1) - cx0, odd = gx & 1;
    v2f cc[5 * NR];
    coarse_block_load<C>(cc, Cf + cyb * (2 * C::CWP) + ccb, odd, std::make_integer_sequence<int, NR>{});
    v2f ruv[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const v2f c0 = cc[5 * i], c1 = cc[5 * i + 1], c2 = cc[5 * i + 2], ca = cc[5 * i + 3], cb = cc[5 * i + 4];
        v2f t = c0 * (v2f){g5[0], g5[0]};
        t = __builtin_elementwise_fma(ca, (v2f){g5[1], g5[1]}, t);
        t = __builtin_elementwise_fma(c1, (v2f){g5[2], g5[2]}, t);
        t = __builtin_elementwise_fma(cb, (v2f){g5[3], g5[3]}, t);
        ruv[i] = __builtin_elementwise_fma(c2, (v2f){g5[4], g5[4]}, t);
    }
#pragma unroll
    for (int p = 0; p < MRV / 2; p++) {
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const int i1 = o ? p + 1 : p, i3 = o ? p + 2 : p + 1;
            v2f auv = ruv[p] * (v2f){g5[0], g5[0]};
            auv = __builtin_elementwise_fma(ruv[i1], (v2f){g5[1], g5[1]}, auv);
            auv = __builtin_elementwise_fma(ruv[p + 1], (v2f){g5[2], g5[2]}, auv);
            auv = __builtin_elementwise_fma(ruv[i3], (v2f){g5[3], g5[3]}, auv);
            auv = __builtin_elementwise_fma(ruv[p + 2], (v2f){g5[4], g5[4]}, auv);
            huv[2 * p + o] = auv;
            buv[2 * p + o] = auv * (v2f){2.f, 2.f};  // OpticalFlow.cpp:142,144
        }
    }
}

// One block: new warped rows, gradient block, row pass of the five fields into field rows QC .. QC + B - 1.  Entered
// with the block's staging in flight; leaves with the next block's staging in flight (if `stage_next`).
// cbuf: which coarse buffer holds this block's coarse flow.
template <int R>
__device__ __forceinline__ void strip_block_front(const LkLevelArgs &a, const TapsN<2 * R + 1> &g, float *lds,
                                                  const float *__restrict__ prev, const float *__restrict__ next, int pair,
                                                  int x0, int y, int cbuf, bool stage_next, bool do_move, int tid) {
    using C = StripCfg<R>;
    float *F0 = lds, *P = F0 + C::FIELD_F, *Wb = P + C::P_F, *X = Wb + C::W_F, *Cf0 = X + C::X_F;
    float *Nx = X, *Gb = X, *Cf = Cf0 + cbuf * C::CS_F;
    constexpr int RW = C::RW, PS = C::PS, H = C::H, M = C::M, NW = C::NW, B = C::B, MR = C::MR;
    // ---- the march: warped rows y + 8 .. y + 23 (buffer rows 2 .. 17), 4-row jobs ------------------------------------
    {
        int nx0s = x0 - H - M, ny0s = y;
        asm("" : "+s"(nx0s), "+s"(ny0s));
        const int cx0 = (x0 - H) / 2 - 1, cy0 = y / 2 - 1;
        // The march has jobs for five of the eight waves; the other three carry the previous block's last 2R row-pass rows
        // of the five fields to the front of their buffers meanwhile (nobody reads rows 0 .. 2R - 1 before this block's
        // column pass, nobody writes rows 2R .. before its row pass) -- the move is off the critical path and the
        // barrier that used to precede it is gone.
        constexpr int NJ = RW * (B / MR);
        static_assert(NJ <= 5 * 64 && C::NT == 512, "march jobs on waves 0-4, the carry on waves 5-7");
        if (do_move && tid >= 5 * 64) {
            constexpr int MV4 = C::QC * C::RBS / 4, FS4 = C::GH * C::RBS / 4;
            v4f *f4 = reinterpret_cast<v4f *>(F0);
            for (int i = tid - 5 * 64; i < C::NF * MV4; i += C::NT - 5 * 64) {
                const int f = i / MV4, k = i - f * MV4;
                f4[f * FS4 + k] = f4[f * FS4 + B * C::RBS / 4 + k];
            }
        }
        for (int n = tid; n < NJ; n += C::NT) {
            const int rg = n / RW, lx = n - rg * RW;
            const int gx = x0 - H + lx, gy0 = y + (R + 1) + MR * rg;
            v2f huv[MR], buv[MR];
            strip_pyrup<C, MR>(Cf, cx0, cy0, gx, gy0, huv, buv);
            const float xf32 = 32.f * (float)gx, yf32 = 32.f * (float)gy0;
            typedef __attribute__((address_space(3))) float lds_float;
            lds_float *wrow = (lds_float *)(Wb + (2 + MR * rg) * PS + lx);
            asm("" : "+v"(wrow));
#pragma unroll
            for (int j = 0; j < MR; j++) {
                wrow[j * PS] = warp_sample_staged<NW, C::NH, 64>(Nx, nx0s, ny0s, next, a.rows, a.cols, a.img_stride,
                                                                 (v2f){xf32, yf32 + 32.f * (float)j}, huv[j]);
                __builtin_amdgcn_sched_barrier(0);  // (as the tile body: keep the rows from interleaving; without: the same time)
            }
        }
    }
    __syncthreads();
    // ---- gradients: block rows 0 .. 15 = level rows y + 7 .. y + 22; row q's centre is buffer row q + 1 ----------------
    {
        const float s1 = 1.f / 9.f, s2 = 2.f * s1;  // OpticalFlow.cpp:19
        constexpr int SEG = 2, GW2 = C::GW / 2, NSEG = B / SEG;  // (4-row segments: 156 jobs on 2.4 waves, measured 2 % slower)
        const v2f s1v = {s1, s1}, s2v = {s2, s2}, halfv = {0.5f, 0.5f};
        float *Gx = Gb, *Gy = Gb + C::GP, *Gt = Gb + 2 * C::GP;
        for (int n = tid; n < GW2 * NSEG; n += C::NT) {
            const int seg = n / GW2, qx = 2 * (n - seg * GW2);
            const int q0 = seg * SEG, lx = qx + (H - R);
            v2f ptx[3], pty[3], wtx[3], wty[3], pcc[3], wcc[3];
            auto rowpass = [&](int br, int slot) {
                const float *pr = P + br * PS + lx, *wr = Wb + br * PS + lx;
                // lx is odd (H - R = 1): (pr[-1], pr[0]) and (pr[1], pr[2]) are aligned pairs
                const v2f p01 = *reinterpret_cast<const v2f *>(pr - 1), p23 = *reinterpret_cast<const v2f *>(pr + 1);
                const v2f w01 = *reinterpret_cast<const v2f *>(wr - 1), w23 = *reinterpret_cast<const v2f *>(wr + 1);
                const v2f pa = p01, pb = (v2f){p01.y, p23.x}, pc = p23;
                const v2f wa = w01, wb = (v2f){w01.y, w23.x}, wc = w23;
                ptx[slot] = pc - pa;
                pty[slot] = __builtin_elementwise_fma(pc, s1v, __builtin_elementwise_fma(pb, s2v, pa * s1v));
                wtx[slot] = wc - wa;
                wty[slot] = __builtin_elementwise_fma(wc, s1v, __builtin_elementwise_fma(wb, s2v, wa * s1v));
                pcc[slot] = pb;
                wcc[slot] = wb;
            };
            rowpass(q0, 0);      // buffer row q0 = the row above block row q0
            rowpass(q0 + 1, 1);
#pragma unroll
            for (int t = 0; t < SEG; t++) {
                const int s_new = (t + 2) % 3, s_top = t % 3, s_mid = (t + 1) % 3;
                rowpass(q0 + t + 2, s_new);
                const v2f pgx = __builtin_elementwise_fma(ptx[s_new], s1v, __builtin_elementwise_fma(ptx[s_mid], s2v, ptx[s_top] * s1v));
                const v2f pgy = pty[s_new] - pty[s_top];
                const v2f ngx = __builtin_elementwise_fma(wtx[s_new], s1v, __builtin_elementwise_fma(wtx[s_mid], s2v, wtx[s_top] * s1v));
                const v2f ngy = wty[s_new] - wty[s_top];
                // OpticalFlow.cpp:62-64: avg2(next, prev) = next * .5f + prev * .5f, It = next - prev
                *reinterpret_cast<v2f *>(Gx + (q0 + t) * C::GS + qx) = ngx * halfv + pgx * halfv;
                *reinterpret_cast<v2f *>(Gy + (q0 + t) * C::GS + qx) = ngy * halfv + pgy * halfv;
                *reinterpret_cast<v2f *>(Gt + (q0 + t) * C::GS + qx) = wcc[s_mid] - pcc[s_mid];
            }
        }
    }
    __syncthreads();
    // ---- row pass of the five product fields, 16 rows -> field rows QC .. QC + 15; the two halves of the workgroup
    // split the fields (xx, xy, yy | xt, yt).  The threads that have nothing to do in the second half's shorter job carry
    // the last two warped rows to the top of the buffer (the next block's Sobel ring).
    {
        const int half = tid >> 8, j = tid & 255, lane = j & 63, w4 = j >> 6;
        const int grp = lane >> 2, c0 = 4 * grp, q = 4 * w4 + (lane & 3);
        const float *Gx = Gb, *Gy = Gb + C::GP, *Gt = Gb + 2 * C::GP;
        float *o = F0 + rb_off(C::QC + q, grp);
        constexpr int FS = C::GH * C::RBS;
        v2f wx[2 * C::WV], wy[2 * C::WV];
        load_window_pairs<C>(Gx, q, c0, wx);
        load_window_pairs<C>(Gy, q, c0, wy);
        if (half == 0) {
            row_taps_skew<C>(wx, wx, g, o);
            row_taps_skew<C>(wx, wy, g, o + FS);
            row_taps_skew<C>(wy, wy, g, o + 2 * FS);
        } else {
            v2f wt[2 * C::WV];
            load_window_pairs<C>(Gt, q, c0, wt);
            row_taps_skew<C>(wx, wt, g, o + 3 * FS);
            row_taps_skew<C>(wy, wt, g, o + 4 * FS);
            if (j < 2 * (PS / 4)) {  // warped rows B, B + 1 -> rows 0, 1 (nobody reads either before the next march)
                reinterpret_cast<v4f *>(Wb)[j] = reinterpret_cast<const v4f *>(Wb + B * PS)[j];
            }
        }
    }
    __syncthreads();
    // the gradient block, the prev rows and the other coarse buffer are dead: the next block's staging goes in
    // (issuing it at the block's start instead -- a timing experiment with wrong results -- is no faster: 206.6 against
    // 197.8 us; the staging round trip is not what the block waits for)
    if (stage_next) strip_stage<C>(a, prev, next, pair, x0, y + B, P, Nx, Cf0 + (cbuf ^ 1) * C::CS_F, tid);
}

// A strip segment: output rows [s0, s1) (multiples of 16, inside the interior tile rows) of columns x0 .. x0 + 63.
template <int R>
__device__ __forceinline__ void lk_strip(const LkLevelArgs &a, const TapsN<2 * R + 1> &g, float *lds, int x0, int s0, int s1, int pair) {
    using C = StripCfg<R>;
    constexpr int B = C::B, RPT = C::RPT;
    const int tid = threadIdx.x;
    float *F0 = lds, *P = F0 + C::FIELD_F, *X = P + C::P_F + C::W_F, *Cf0 = X + C::X_F;
    const float *__restrict__ prev = a.prev + pair * a.img_pair;
    const float *__restrict__ next = a.next + pair * a.img_pair;
    float *__restrict__ ou = a.out_u + pair * a.out_pair;
    float *__restrict__ ov = a.out_v + pair * a.out_pair;
    const int c = tid & 63, r0 = RPT * (tid >> 6), gx = x0 + c;
    int cls[4];
    col_bases<C>(F0, c, r0, cls);

    // warm-up: the block above the segment, front half only -- it leaves field rows s0 - 9 .. s0 + 6 in buffer rows
    // QC .. QC + 15 (the first two from an unset Sobel ring: never read) and the warped rows s0 + 6, s0 + 7 carried
    int cbuf = 0;
    strip_stage<C>(a, prev, next, pair, x0, s0 - B, P, X, Cf0, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    strip_block_front<R>(a, g, lds, prev, next, pair, x0, s0 - B, cbuf, true, false, tid);
    cbuf ^= 1;  // (its field rows B .. B + QC - 1 move to the front during the first block's march)

    float ru[RPT], rv[RPT];
    int ry = -1;
    auto flush = [&]() {
        if (ry >= 0) {
#pragma unroll
            for (int j = 0; j < RPT; j++) {
                ou[(size_t)(ry + r0 + j) * a.out_stride + gx] = ru[j];
                ov[(size_t)(ry + r0 + j) * a.out_stride + gx] = rv[j];
            }
        }
    };
#pragma unroll 1
    for (int y = s0; y < s1; y += B) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's staging has landed (every wave's own part)
        __syncthreads();                                   // ... all of it; the carried rows are in place
        flush();  // the previous block's results drain under this block's arithmetic
        strip_block_front<R>(a, g, lds, prev, next, pair, x0, y, cbuf, y + B < s1, true, tid);
        float S[C::NF][RPT];
        col_pass<C, 0>(cls, S[0], g);
        col_pass<C, 1>(cls, S[1], g);
        col_pass<C, 2>(cls, S[2], g);
        col_pass<C, 3>(cls, S[3], g);
        col_pass<C, 4>(cls, S[4], g);
        // the base flow of the own pixels: pyrUp again from this block's coarse buffer (11 instructions per pixel; keeping
        // what the march computed for rows y + 8 .. would need 8 KB more LDS)
        v2f huv[RPT], buv[RPT];
        if (a.add_base) strip_pyrup<C, RPT>(Cf0 + cbuf * C::CS_F, (x0 - C::H) / 2 - 1, y / 2 - 1, gx, y + r0, huv, buv);
        // (no barrier here: the next writer of the field buffers is the carry in the next block's march, behind the
        // barrier at the loop's top; the coarse buffer this pyrUp read is restaged two row passes from now)
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            float uu, vv;
            lk_solve(S[0][j], S[1][j], S[2][j], S[3][j], S[4][j], uu, vv);
            ru[j] = a.add_base ? buv[j].x + uu : uu;  // OpticalFlow.cpp:161-162
            rv[j] = a.add_base ? buv[j].y + vv : vv;
        }
        ry = y;
        cbuf ^= 1;
    }
    flush();
}

