// lk_split.hip -- the second half of a SPLIT level launch (r05): the five Gaussian-weighted window sums and the
// 2x2 solve of lk::calcOpticalFlow (OpticalFlow.cpp:66-103) as a STREAMING kernel over gradient planes that a
// pre-pass (lk_fused.hip, lk_grad_kernel) left in HBM.
//
// Why: the fused level kernel recomputes pyrUp + warp + Sobel on a 1.875x halo'd region per 64x32 tile and runs its
// row pass on 46 gradient rows for 32 output rows (1.44x); every tile shape that shrinks those factors was measured
// and lost to LDS capacity.  HBM has 5x headroom, so the launch is cut at the gradient planes:
//   pre-pass   Ix, Iy, It once per pixel (tile + 1 px), written to a padded plane set -- the R cells around the
//              image hold the BORDER_REFLECT_101 copies cv::GaussianBlur would read (OpticalFlow.cpp:73-77), so this
//              kernel has no border case at all -- and the base flow 2 * pyrUp (OpticalFlow.cpp:140-145) into u, v;
//   this       a workgroup owns a 64-column strip of a row segment and walks down it in blocks of 16 rows: LDS-DMA
//              of the block's 16 new gradient rows (3 planes x 80 floats), ONE row pass of all five product fields
//              (each gradient window loaded once; 16 rows for 16 output rows: 1.0x, after a 14-row warm-up per
//              segment), column pass + double-precision solve as in the fused kernel, u += du.  The last 2R rows of
//              the row-pass outputs are carried to the next block in LDS (moved to the front: 17.9 KB that the fused
//              kernel's 80 KB never had room for).
// 53.76 KB of LDS = three 256-thread workgroups per CU.  Same fmaf chains as the fused kernel (lk_window.hpp): same
// bits; tests/test_lk_gpu.py compares the two paths and the oracle.
#include "lk_window.hpp"

#include <mutex>

namespace micv {

template <int R_>
struct SumsCfg {
    static constexpr int R = R_, W = 2 * R + 1;
    static constexpr int TW = 64, B = 16, NT = 256, RPT = 4, TH = B;
    static_assert((NT / TW) * RPT == B, "a block is one column-pass job per thread");
    static constexpr int GW = TW + 2 * R;
    static constexpr int GP = (GW + 3) & ~3, GS = 3 * GP;  // a block row = [Ix | Iy | It], GP floats each
    static constexpr int WV = (4 + 2 * R + 3) / 4;
    static_assert(4 * (TW / 4 - 1) + 4 * WV <= GP, "row-pass window reads stay inside a plane row");
    static constexpr int QC = 2 * R;   // row-pass rows carried from block to block
    static constexpr int GH = B + QC;  // rows of one field's buffer (the name the column-pass helpers use)
    static constexpr int RBS = 64;
    static constexpr int NF = 5;  // Ix^2, IxIy, Iy^2, IxIt, IyIt
    static constexpr int FIELD_F = NF * GH * RBS;
    static constexpr int G_F = B * GS;
    // field buffers first: the column pass's spare half-pair reads (col_load_pair) run a few rows past a buffer,
    // into the next one or into the gradient block -- inside the kernel's LDS either way
    static constexpr int LDS_FLOATS = FIELD_F + G_F;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOATS * 4;
    static_assert(((NF - 1) * GH + (B - RPT) + RPT + 2 * R + 8) * RBS <= LDS_FLOATS, "spare half-pair reads stay inside LDS");
    static_assert(QC <= B, "the warm-up is one short block");
};

// LDS-DMA of `nrows` rows of the padded gradient planes into the dense block image (row q = [Ix | Iy | It], GP floats
// each = GS / 4 float4 slots): slot i of the block lives at float4 i, one wave-instruction covers 64 consecutive
// slots.  `src` points at plane 0 of the first row, first column of the strip; rows are 3 * gpitch floats apart, the
// planes of a row gpitch apart.
template <typename C>
__device__ __forceinline__ void dma_grad_rows(const float *__restrict__ src, int gpitch, int nrows, float *dst, int tid) {
    constexpr int V4 = C::GS / 4, P4 = C::GP / 4, NT = C::NT, NP = (C::B * V4 + NT - 1) / NT;
    constexpr int A = NT / V4, Bs = NT % V4;  // slot + NT = (row + A, float4 + Bs), one carry
    const int lane = tid & 63;
    const int slot0 = __builtin_amdgcn_readfirstlane(tid - lane);
    int row = tid / V4, rem = tid - row * V4;
    const size_t rowp = 3 * (size_t)gpitch;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        if (row < nrows) {
            const int plane = rem / P4, c4 = rem - plane * P4;
            __builtin_amdgcn_global_load_lds((glb_cvoid *)(src + row * rowp + (size_t)plane * gpitch + 4 * c4),
                                             (lds_void *)(dst + 4 * (slot0 + k * NT)), 16, 0, 0);
        }
        rem += Bs;
        const bool cy = rem >= V4;
        rem -= cy ? V4 : 0;
        row += A + (cy ? 1 : 0);
    }
}

// Row pass of all five product fields for block rows [0, nrows): the job of a thread is one row x four adjacent outputs;
// the three gradient windows are loaded once and feed five skewed packed chains pairs (lk_window.hpp).  Outputs go
// to rows buf0 .. buf0 + nrows - 1 of the field buffers (XOR-swizzled chunks, rb_off).
template <typename C>
__device__ __forceinline__ void sums_row_pass(const float *Gb, float *F0, int nrows, int buf0, const TapsN<C::W> &g, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = lane >> 2, c0 = 4 * grp, q = 4 * wave + (lane & 3);
    if (q < nrows) {
        v2f wx[2 * C::WV], wy[2 * C::WV], wt[2 * C::WV];
        load_window_pairs<C>(Gb, q, c0, wx);
        load_window_pairs<C>(Gb + C::GP, q, c0, wy);
        load_window_pairs<C>(Gb + 2 * C::GP, q, c0, wt);
        float *o = F0 + rb_off(buf0 + q, grp);
        constexpr int FS = C::GH * C::RBS;
        row_taps_skew<C>(wx, wx, g, o);
        row_taps_skew<C>(wx, wy, g, o + FS);
        row_taps_skew<C>(wy, wy, g, o + 2 * FS);
        row_taps_skew<C>(wx, wt, g, o + 3 * FS);
        row_taps_skew<C>(wy, wt, g, o + 4 * FS);
    }
}

// The base flow of a thread's own pixels -- column gx, rows gy0 .. gy0 + 3 (gy0 even): 2 * pyr::pyrUp of the coarse
// flow (OpticalFlow.cpp:140-145; Pyramids.cu:126-127: 2x replicate, [1,4,6,4,1]/16 rows then columns,
// BORDER_REFLECT_101) with the fused kernel's chains (lk_fused.hip, march): row taps of fine column gx read coarse
// columns {m-1, m-1, m, m, m+1} (gx = 2m) or {m-1, m, m, m+1, m+1} (gx = 2m + 1), rows likewise; replicated edges give
// the reflected pattern except tap 0 of column / row 0 and tap 4 of the last one.  Coarse values come straight from
// global memory (L1 / L2: neighbouring lanes share them), clamped indices = replicated edges.
template <int RPT>
__device__ __forceinline__ void pyrup2_own(const float *__restrict__ fu, const float *__restrict__ fv, int fr, int fc, int gx,
                                           int gy0, int rows, int cols, float (&bu)[RPT], float (&bv)[RPT]) {
    constexpr int NR = RPT / 2 + 2;
    const float g5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};  // Pyramids.cu:19
    const int cyb = (gy0 >> 1) - 1, ccb = (gx >> 1) - 1;
    const bool odd = gx & 1, first_col = gx == 0, last_col = gx == cols - 1;
    const int x0 = clampi(ccb, 0, fc - 1), x1 = clampi(ccb + 1, 0, fc - 1), x2 = clampi(ccb + 2, 0, fc - 1);
    v2f c[NR][3];
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const size_t ro = (size_t)clampi(cyb + i, 0, fr - 1) * fc;
        c[i][0] = (v2f){fu[ro + x0], fv[ro + x0]};
        c[i][1] = (v2f){fu[ro + x1], fv[ro + x1]};
        c[i][2] = (v2f){fu[ro + x2], fv[ro + x2]};
    }
    v2f ruv[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const v2f c0 = c[i][0], c1 = c[i][1], c2 = c[i][2];
        const v2f ca = odd ? c1 : c0, cb = odd ? c2 : c1;
        const v2f t0 = first_col ? c2 : c0, t4 = last_col ? c0 : c2;
        v2f t = t0 * (v2f){g5[0], g5[0]};
        t = __builtin_elementwise_fma(ca, (v2f){g5[1], g5[1]}, t);
        t = __builtin_elementwise_fma(c1, (v2f){g5[2], g5[2]}, t);
        t = __builtin_elementwise_fma(cb, (v2f){g5[3], g5[3]}, t);
        ruv[i] = __builtin_elementwise_fma(t4, (v2f){g5[4], g5[4]}, t);
    }
#pragma unroll
    for (int p = 0; p < RPT / 2; p++) {
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const int j = 2 * p + o;
            const int i1 = o ? p + 1 : p, i3 = o ? p + 2 : p + 1;
            const v2f r0 = (gy0 + j == 0) ? ruv[p + 2] : ruv[p];
            const v2f r4 = (gy0 + j == rows - 1) ? ruv[p] : ruv[p + 2];
            v2f auv = r0 * (v2f){g5[0], g5[0]};
            auv = __builtin_elementwise_fma(ruv[i1], (v2f){g5[1], g5[1]}, auv);
            auv = __builtin_elementwise_fma(ruv[p + 1], (v2f){g5[2], g5[2]}, auv);
            auv = __builtin_elementwise_fma(ruv[i3], (v2f){g5[3], g5[3]}, auv);
            auv = __builtin_elementwise_fma(r4, (v2f){g5[4], g5[4]}, auv);
            auv = auv * (v2f){2.f, 2.f};  // OpticalFlow.cpp:142,144
            bu[j] = auv.x;
            bv[j] = auv.y;
        }
    }
}

template <int R>
__global__ __launch_bounds__(256, 3) void lk_sums_stream_kernel(LkSumsArgs a, TapsN<2 * R + 1> g) {
    using C = SumsCfg<R>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *F0 = lds;               // five field buffers of GH rows
    float *Gb = lds + C::FIELD_F;  // the gradient block
    const int tid = threadIdx.x;
    // XCD-aware item order: workgroup b runs on XCD b % 8; each XCD takes a contiguous run of items, and items are
    // numbered strip-fastest, so the strips that share gradient columns (14 of 78) meet in one L2.
    const int nb = gridDim.x, per = nb >> 3, rem = nb & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int item = xcd * per + (xcd < rem ? xcd : rem) + idx;
    const int strip = item % a.strips, t = item / a.strips;
    const int seg = t % a.segs, pair = t / a.segs;
    const int x0 = strip * C::TW, s0 = seg * a.seg_rows;
    const int s1 = s0 + a.seg_rows < a.rows ? s0 + a.seg_rows : a.rows;
    // padded plane coordinates: image pixel (y, x) is cell (y + R, x + R); the strip's first gradient column x0 - R is
    // padded column x0 (16-byte aligned: gpitch and x0 are multiples of 4)
    const float *__restrict__ gp = a.grad + pair * a.grad_pair + x0;
    const size_t rowp = 3 * (size_t)a.gpitch;
    float *__restrict__ ou = a.out_u + pair * a.out_pair;
    float *__restrict__ ov = a.out_v + pair * a.out_pair;

    const int c = tid & (C::TW - 1), r0 = C::RPT * (tid / C::TW);
    const int gx = x0 + c;
    int cls[4];
    col_bases<C>(F0, c, r0, cls);

    // warm-up: row-pass rows s0 - R .. s0 + R - 1 (padded rows s0 .. s0 + 2R - 1) -> buffer rows 0 .. 2R - 1
    dma_grad_rows<C>(gp + (size_t)s0 * rowp, a.gpitch, C::QC, Gb, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    sums_row_pass<C>(Gb, F0, C::QC, 0, g, tid);
    __syncthreads();
    // the first block's gradient rows
    dma_grad_rows<C>(gp + (size_t)(s0 + C::QC) * rowp, a.gpitch, C::B, Gb, tid);

    // Results stay in registers until the NEXT block's gradient rows have been awaited: that wait is vmcnt(0), and a
    // store issued just before it would be a memory round trip in front of every row pass (the wave may not go on
    // before the store is acknowledged).  Issued right behind the wait they drain under the block's arithmetic.
    float ru[C::RPT], rv[C::RPT];
    int ry = -1;  // first row of the block whose results are pending
    auto flush = [&]() {
        if (ry >= 0 && gx < a.cols) {
#pragma unroll
            for (int j = 0; j < C::RPT; j++) {
                const int gy = ry + r0 + j;
                if (gy < s1) {
                    ou[(size_t)gy * a.out_stride + gx] = ru[j];
                    ov[(size_t)gy * a.out_stride + gx] = rv[j];
                }
            }
        }
    };
#pragma unroll 1
    for (int y = s0; y < s1; y += C::B) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's gradient rows have landed (every wave's own part)
        __syncthreads();                                   // ... all of them; the carried rows are in place
        flush();
        // base flow of the block's own pixels: recomputed from the coarse flow (base == 2; 2 B per pixel from L2) or
        // read back from u, v where the pre-pass left it (base == 1); loads issued here, used after the solve
        float bu[C::RPT], bv[C::RPT];
        if (a.base == 2) {
            pyrup2_own<C::RPT>(a.flow_u + pair * a.flow_pair, a.flow_v + pair * a.flow_pair, a.flow_rows, a.flow_cols,
                               gx < a.cols ? gx : a.cols - 1, y + r0, a.rows, a.cols, bu, bv);
        } else {
#pragma unroll
            for (int j = 0; j < C::RPT; j++) {
                const int gy = y + r0 + j;
                const bool ok = a.base == 1 && gx < a.cols && gy < s1;
                bu[j] = ok ? ou[(size_t)gy * a.out_stride + gx] : 0.f;
                bv[j] = ok ? ov[(size_t)gy * a.out_stride + gx] : 0.f;
            }
        }
        sums_row_pass<C>(Gb, F0, C::B, C::QC, g, tid);     // padded rows y + 2R .. -> buffer rows 2R .. 2R + B - 1
        __syncthreads();
        float S[C::NF][C::RPT];
        col_pass<C, 0, true>(cls, S[0], g);
        col_pass<C, 1, true>(cls, S[1], g);
        col_pass<C, 2, true>(cls, S[2], g);
        col_pass<C, 3, true>(cls, S[3], g);
        col_pass<C, 4, true>(cls, S[4], g);
        __syncthreads();  // every column pass has read its rows
        if (y + C::B < s1) {
            // carry: buffer rows B .. B + 2R - 1 become rows 0 .. 2R - 1 (B is a multiple of 4: same swizzle class)
            constexpr int MV4 = C::QC * C::RBS / 4, FS4 = C::GH * C::RBS / 4;
            v4f *f4 = reinterpret_cast<v4f *>(F0);
            for (int i = tid; i < C::NF * MV4; i += C::NT) {
                const int f = i / MV4, k = i - f * MV4;
                f4[f * FS4 + k] = f4[f * FS4 + C::B * C::RBS / 4 + k];
            }
            // the next block's gradient rows (the block is dead since the barrier behind the row pass)
            dma_grad_rows<C>(gp + (size_t)(y + C::B + C::QC) * rowp, a.gpitch, C::B, Gb, tid);
        }
#pragma unroll
        for (int j = 0; j < C::RPT; j++) {
            float uu, vv;
            lk_solve(S[0][j], S[1][j], S[2][j], S[3][j], S[4][j], uu, vv);
            ru[j] = a.base ? bu[j] + uu : uu;  // OpticalFlow.cpp:161-162
            rv[j] = a.base ? bv[j] + vv : vv;
        }
        ry = y;
    }
    flush();
}

// Work items of the streaming launch: strips of 64 columns cut into row segments so that all items together are about
// one round of the GPU's workgroup slots (three per CU); every segment pays a 2R-row warm-up, so fewer, longer
// segments are cheaper as long as the slots are filled.
void lk_sums_partition(int rows, int cols, int batch, int slots, int *strips, int *segs, int *seg_rows) {
    const int st = cdiv(cols, 64);
    const long total = (long)st * batch;
    long k = total >= slots ? 1 : slots / total;
    int sr = (cdiv(rows, (int)k) + 15) & ~15;
    if (sr < 32) sr = 32;
    *strips = st;
    *seg_rows = sr;
    *segs = cdiv(rows, sr);
}

template <int R>
static int launch_sums_r(hipStream_t s, const LkSumsArgs &a) {
    using C = SumsCfg<R>;
    static TapsN<2 * R + 1> taps;
    static std::once_flag once;
    std::call_once(once, [] {
        Taps t;
        gaussian_taps(2 * R + 1, (double)((float)(2 * R + 1) / 3.f), &t);  // OpticalFlow.cpp:73
        for (int i = 0; i < 2 * R + 1; i++) taps.k[i] = t.k[i];
    });
    {
        static thread_local int done_dev = -1;
        int dev = 0;
        MICV_HIP(hipGetDevice(&dev));
        if (done_dev != dev) {
            MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lk_sums_stream_kernel<R>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
            done_dev = dev;
        }
    }
    const long items = (long)a.strips * a.segs * a.batch;
    lk_sums_stream_kernel<R><<<(unsigned)items, C::NT, C::LDS_BYTES, s>>>(a, taps);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int launch_lk_sums(hipStream_t s, const LkSumsArgs &a, int win) {
    switch (win) {
        case 15: return launch_sums_r<7>(s, a);
        default:
            set_error("lk split: window %d has no streaming sums kernel", win);
            return MICV_EUNSUPPORTED;
    }
}

}  // namespace micv
