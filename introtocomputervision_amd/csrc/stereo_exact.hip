// stereo_exact.hip -- ps2 window stereo on 8-bit-valued images (a12 / a13), the exact-sum kernels.
//
// Every plain ps2 call hands over 8-bit images converted to CV_32F (ps2_cpp/src/main.cpp:87-88,115-117), and
// serial::disparitySSD sums integers by definition (DisparitySSD.cpp:45-51).  When both images hold integers in
// [0, 255] and the window has at most 15 x 15 taps, every partial sum of the contract (stereo.hip: column sums top
// -> bottom, window sum left -> right, f32) is an integer below 2^24: each float add is exact and the ORDER of the
// additions does not matter.  The cost can then be formed any way that gives the same integers:
//
//   SSD(y, x, d) = A(y, x) + B(y, x + d) - 2 C(y, x, d)
//     A = window sum of left^2, B = window sum of right^2 (one value per window POSITION, formed once per position
//     by stereo_energy8_kernel -- the same sharing DisparityNCorr's energy field already uses), C = window sum of
//     left * right.  A does not depend on d, so arg min_d SSD = arg max_d (2 C - B).
//
// Kernel shape (stereo_exact_kernel): LANES ARE DISPARITIES.  One wave64 owns 8 output rows x X output columns and
// 64 consecutive disparities; it walks along x.  The left pixel of a column is then the same for all 64 lanes --
// a SCALAR operand: the pre-pass packs four rows of a column into one dword (bytes), the wave fetches the packed
// left words of the column with s_load (no VALU, no LDS), and the packed right words at column x + d0 + lane from
// its LDS strip (consecutive lanes, conflict-free).  v_dot4_u32_u8 multiplies four rows and accumulates in one
// instruction, so the 8 column sums of 2r + 1 rows cost 18 instructions (r = 5; rows outside a window are masked
// in the scalar operand, shared runs of full words are formed once).  The window sum slides along x IN the lane:
//   C += cs(x + 2r);  [use];  C -= cs(x)       (a ring of 2r column sums per row, statically indexed)
// -- no cross-lane traffic at all until the arg max, which is the one thing the lanes of a pixel share.  The key
//   key = (2 C - B(x + d)) * 64 - (x_rel + lane)    = v_lshl_add_u32(C, 7, T[x_rel + lane]), T from the LDS strip
// orders by cost, then by lane (lowest disparity wins ties: the contract's strict '<'), in one instruction; the
// 8 keys of a column (8 rows) are max-reduced TRANSPOSED: three merge levels (row_ror:8 in a DPP row,
// v_permlane16_swap, v_permlane32_swap) fold 8 registers into one whose lane group 8j..8j+7 belongs to row j, three
// more DPP steps finish the reduction inside the groups.  Eight columns' results collect in one register (lane
// 8j + c = row j, column c) and are decoded, compared with the previous 64 disparities' best and stored together.
//
// Per (pixel, 64 disparities): 2.25 v_dot4 + 2 add/sub + 1 key + ~2.4 reduction instructions, against 27 lane
// instructions per (pixel, disparity) of the float kernel that re-adds every window in the contract's order.
//
// Which images qualify is decided ON THE DEVICE, without a host round trip: the pack pre-pass tests every pixel
// (integer, 0..255) and writes the launch's epoch into the context's flag word on the first failure; this kernel
// returns at once when the flag holds its epoch, the float kernels of stereo.hip (launched behind it) when it does
// not.  MICV_OPT_STEREO_EXACT = -1 never takes this path.
#include <utility>

#include "kernels.hpp"
#include "stereo_exact.hpp"

namespace micv {

namespace {

typedef const __attribute__((address_space(4))) uint32_t *sx_cptr;  // scalar (constant-address-space) loads

constexpr int SX_Y = 8;        // output rows per wave
constexpr int SX_LW = 8;       // packed left words per column in the plan (32 B: one s_load_dwordx8)
constexpr int SX_INVALID = (int)0x80000400;  // serial:: positions outside the padded image (T table)

__host__ __device__ constexpr int sx_groups(int R) { return (SX_Y + 2 * R + 3) / 4; }

template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int sx_dpp(int old, int v) {
    // every lane written: no `old` operand, so the move folds into the v_max that consumes it (v_max_i32_dpp)
    if constexpr (ROW_MASK == 0xF && BANK_MASK == 0xF) return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ int sx_max(int a, int b) { return a > b ? a : b; }

// ---- pre-pass 1: pack rows into bytes, test eligibility ----------------------------------------------------------
// Strip s = output rows 8 s .. 8 s + 7; its NR = 8 + 2 R window rows (clamped to the image) go four to a dword:
// word g, byte b = row 8 s - R + 4 g + b.  Left: plan[s][column][8 words]; right: pack[s][g][column].
template <int R>
__global__ __launch_bounds__(256) void stereo_pack_kernel(StereoExactArgs a) {
    constexpr int NR = SX_Y + 2 * R, NG = sx_groups(R);
    const int c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (c >= a.cols) return;
    uint32_t lw[SX_LW], rw[NG];
#pragma unroll
    for (int g = 0; g < SX_LW; g++) lw[g] = 0;
#pragma unroll
    for (int g = 0; g < NG; g++) rw[g] = 0;
    bool ok = true;
    float lv[NR], rv[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int yy = clampi(s * SX_Y - R + k, 0, a.rows - 1);
        lv[k] = a.left[(size_t)yy * a.stride + c];
        rv[k] = a.right[(size_t)yy * a.stride + c];
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
        // (unsigned) of a float saturates, NaN -> 0: the comparison back rejects everything that is not 0..255 exactly
        // (-0 passes and packs as 0: its squares, products and sums are the same numbers)
        const uint32_t ul = (uint32_t)lv[k], ur = (uint32_t)rv[k];
        ok = ok && (float)ul == lv[k] && ul <= 255u && (float)ur == rv[k] && ur <= 255u;
        lw[k >> 2] |= (ul & 255u) << (8 * (k & 3));
        rw[k >> 2] |= (ur & 255u) << (8 * (k & 3));
    }
    uint4 *lp = reinterpret_cast<uint4 *>(a.lplan + ((size_t)s * a.cols + c) * SX_LW);
    lp[0] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    lp[1] = make_uint4(lw[4], lw[5], lw[6], lw[7]);
#pragma unroll
    for (int g = 0; g < NG; g++) a.rpack[((size_t)s * NG + g) * a.colsP + c] = rw[g];
    if (!ok) *a.flag = a.epoch;  // every failing thread stores the same word
}

// Column sum of rows J .. J + 2 R (strip-local) of the byte-wise product of two packed columns.  The run of FULL
// words is a chain in a fixed order, so the runs shared by neighbouring J are one computation (CSE); head and tail
// words are masked in the operand `l` (a scalar in the search kernel: s_and, free).
template <int R, int J, int NG>
__device__ __forceinline__ uint32_t sx_colsum(const uint32_t (&l)[NG], const uint32_t (&r)[NG]) {
    constexpr int k0 = J, k1 = J + 2 * R;
    constexpr int g0 = k0 / 4, g1 = k1 / 4;
    constexpr int f0 = (k0 + 3) / 4, f1 = (k1 + 1) / 4 - 1;  // words that lie wholly inside the window
    uint32_t acc = 0;
#pragma unroll
    for (int g = f0; g <= f1; g++) acc = __builtin_amdgcn_udot4(l[g], r[g], acc, false);
    if (g0 < f0) acc = __builtin_amdgcn_udot4(l[g0] & (0xFFFFFFFFu << (8 * (k0 & 3))), r[g0], acc, false);
    if (g1 > f1) acc = __builtin_amdgcn_udot4(l[g1] & (0xFFFFFFFFu >> (8 * (3 - (k1 & 3)))), r[g1], acc, false);
    return acc;
}

template <int R, int NG, size_t... J>
__device__ __forceinline__ void sx_colsums(const uint32_t (&l)[NG], const uint32_t (&r)[NG], uint32_t (&cs)[SX_Y],
                                           std::index_sequence<J...>) {
    ((cs[J] = sx_colsum<R, (int)J, NG>(l, r)), ...);
}

// ---- pre-pass 2: window energies ----------------------------------------------------------------------------------
// blockIdx.z = 0: A(y, x) = window sum of left^2 at output x (window columns x - R .. x - R + wcols - 1, each clamped
// on its own); 1: B(y, p) the same of right at position p = x + d, p in [min_d, cols - 1 + max_d].
template <int R>
__global__ __launch_bounds__(256) void stereo_energy8_kernel(StereoExactArgs a) {
    constexpr int NG = sx_groups(R), WMAX = 2 * R + 1;
    __shared__ uint32_t cs2[SX_Y][256 + WMAX];
    const bool right = blockIdx.z == 1;
    const int npos = right ? a.nB : a.cols, pmin = right ? a.min_d : 0;
    const int p0 = blockIdx.x * 256, s = blockIdx.y;
    if (p0 >= npos) return;
    for (int u = threadIdx.x; u < 256 + a.wcols - 1; u += 256) {
        const int q = clampi(pmin + p0 + u - R, 0, a.cols - 1);
        uint32_t w[NG];
        if (right) {
#pragma unroll
            for (int g = 0; g < NG; g++) w[g] = a.rpack[((size_t)s * NG + g) * a.colsP + q];
        } else {
#pragma unroll
            for (int g = 0; g < NG; g++) w[g] = a.lplan[((size_t)s * a.cols + q) * SX_LW + g];
        }
        uint32_t cs[SX_Y];
        sx_colsums<R, NG>(w, w, cs, std::make_index_sequence<SX_Y>{});
#pragma unroll
        for (int j = 0; j < SX_Y; j++) cs2[j][u] = cs[j];
    }
    __syncthreads();
    const int p = p0 + threadIdx.x;
    if (p >= npos) return;
    int32_t *out = right ? a.B : a.A;
#pragma unroll
    for (int j = 0; j < SX_Y; j++) {
        uint32_t e = 0;
        for (int i = 0; i < a.wcols; i++) e += cs2[j][threadIdx.x + i];
        const int y = s * SX_Y + j;
        if (y < a.rows) out[(size_t)y * npos + p] = (int32_t)e;
    }
}

// ---- the search ------------------------------------------------------------------------------------------------
// Transposed max-reduction of the 8 keys of a column: lane group 8 j .. 8 j + 7 of the result holds row j's maximum.
__device__ __forceinline__ int sx_merge8(int a, int b) {  // partner 8 lanes away inside a DPP row; a -> lanes with bit 3 clear
    const int ma = sx_max(a, sx_dpp<0x128>(a, a));        // row_ror:8
    const int mb = sx_max(b, sx_dpp<0x128>(b, b));
    return sx_dpp<0xE4, 0xF, 0xC>(ma, mb);                // banks 2, 3 (lanes 8..15 of each row) take b's
}
__device__ __forceinline__ int sx_merge16(int a, int b) {  // a -> even rows of 16 lanes
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    return sx_max((int)r[0], (int)r[1]);
}
__device__ __forceinline__ int sx_merge32(int a, int b) {  // a -> lanes 0..31
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return sx_max((int)r[0], (int)r[1]);
}
__device__ __forceinline__ int sx_reduce8(const int (&k)[SX_Y]) {
    const int a0 = sx_merge8(k[0], k[1]), a1 = sx_merge8(k[2], k[3]), a2 = sx_merge8(k[4], k[5]), a3 = sx_merge8(k[6], k[7]);
    int x = sx_merge32(sx_merge16(a0, a1), sx_merge16(a2, a3));
    x = sx_max(x, sx_dpp<0xB1>(x, x));   // quad_perm:[1,0,3,2]
    x = sx_max(x, sx_dpp<0x4E>(x, x));   // quad_perm:[2,3,0,1]
    x = sx_max(x, sx_dpp<0x141>(x, x));  // row_half_mirror: the other quad of the 8-lane group
    return x;
}

template <int XMAX>
struct SxLayout {  // one wave's LDS, in dwords
    static constexpr int P_MAX = 14;
    static constexpr int XP = XMAX + P_MAX;            // output columns incl. the unroll's overrun
    static constexpr int RSTR = XP + P_MAX + 64;       // right strip: window columns + 63 disparities
    static constexpr int TSTR = XP + 64;               // key table: output columns + 63 disparities
    static constexpr int NB = (XP + 7) / 8;            // batches of 8 output columns
    __host__ __device__ static constexpr int words(int NG) { return NG * RSTR + SX_Y * TSTR + NB * 80; }  // per batch: 64 scores, 64 disparity bytes
};

template <int R, int WC, bool SERIAL, int XMAX>
__global__ __launch_bounds__(256) void stereo_exact_kernel(StereoExactArgs a) {
    constexpr int NG = sx_groups(R), P = WC - 1;  // P: columns of the window that stay when it moves on = ring length
    using Lay = SxLayout<XMAX>;
    static_assert(P <= Lay::P_MAX && P >= 1, "window");
    extern __shared__ uint32_t sx_lds[];
    if (__builtin_nontemporal_load(a.flag) == a.epoch) return;  // not 8-bit-valued: the float kernel does this call
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.ntiles) return;  // whole wave; waves never synchronise with each other
    const int s = tile / a.nxs, ys = s * SX_Y, x0 = (tile - s * a.nxs) * a.X;
    uint32_t *Rs = sx_lds + wave * Lay::words(NG);
    int32_t *Ts = reinterpret_cast<int32_t *>(Rs + NG * Lay::RSTR);
    int32_t *res = Ts + SX_Y * Lay::TSTR;
    const sx_cptr lrow = (sx_cptr)(a.lplan + (size_t)s * a.cols * SX_LW);
    const int nouter = (a.X + P - 1) / P;  // the column loop runs nouter * P output columns (overrun: masked)
    const int nchunks = (a.max_d - a.min_d) / 64 + 1;

    for (int chunk = 0; chunk < nchunks; chunk++) {
        const int d0 = a.min_d + 64 * chunk;
        const int nvalid = a.max_d - d0 + 1 < 64 ? a.max_d - d0 + 1 : 64;
        const int dl = lane < nvalid ? lane : nvalid - 1;  // lanes past max_d repeat the last disparity (same key)
        __builtin_amdgcn_wave_barrier();  // the previous chunk's reads are done (in-order LDS)
        // the strip of packed `right` this chunk slides over: entry i = column clamp(x0 - R + d0 + i)
        const int nrs = nouter * P + P + 63, nts = nouter * P + 63;
#pragma unroll
        for (int g = 0; g < NG; g++)
            for (int i = lane; i < nrs; i += 64)
                Rs[g * Lay::RSTR + i] = a.rpack[((size_t)s * NG + g) * a.colsP + clampi(x0 - R + d0 + i, 0, a.cols - 1)];
        // key table: entry i of row j belongs to position p = x0 + d0 + i (output column x0 + x_rel seen by lane i - x_rel)
#pragma unroll
        for (int j = 0; j < SX_Y; j++) {
            const int y = ys + j < a.rows ? ys + j : a.rows - 1;
            for (int i = lane; i < nts; i += 64) {
                const int p = x0 + d0 + i;
                const int pi = p - a.min_d < a.nB ? p - a.min_d : a.nB - 1;
                int t = -(a.B[(size_t)y * a.nB + pi] << 6) - i;
                if (SERIAL && (p < -R || p > a.cols - 1 + R)) t = SX_INVALID - i;  // DisparitySSD.cpp:42-43
                Ts[j * Lay::TSTR + i] = t;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const uint32_t *rp = Rs + dl;
        const int32_t *tp = Ts + dl;
        uint32_t C[SX_Y], ring[P][SX_Y];
#pragma unroll
        for (int j = 0; j < SX_Y; j++) C[j] = 0;
        // c_rel: window column, 0 = x0 - R (uniform); c_loc: the same column counted from where rp points
        auto column = [&](int c_rel, int c_loc, uint32_t (&cs)[SX_Y]) {
            const int xc = clampi(x0 - R + c_rel, 0, a.cols - 1);
            uint32_t lw[NG], rw[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) {
                lw[g] = lrow[xc * SX_LW + g];
                rw[g] = rp[g * Lay::RSTR + c_loc];
            }
            sx_colsums<R, NG>(lw, rw, cs, std::make_index_sequence<SX_Y>{});
        };
#pragma unroll
        for (int k = 0; k < P; k++) {  // the first 2R (2R - 1) columns of the first window
            uint32_t cs[SX_Y];
            column(k, k, cs);
#pragma unroll
            for (int j = 0; j < SX_Y; j++) {
                C[j] += cs[j];
                ring[k][j] = cs[j];
            }
        }
        int cur = 0;
        for (int base = 0; base < nouter * P; base += P) {
#pragma unroll
            for (int m = 0; m < P; m++) {
                const int x_rel = base + m;  // output column x0 + x_rel; its window: columns x_rel .. x_rel + P
                uint32_t cs[SX_Y];
                column(x_rel + P, m + P, cs);
                int key[SX_Y];
#pragma unroll
                for (int j = 0; j < SX_Y; j++) {
                    C[j] += cs[j];
                    key[j] = (int)(C[j] << 7) + tp[j * Lay::TSTR + m];
                    C[j] -= ring[m][j];
                    ring[m][j] = cs[j];
                }
                const int colres = sx_reduce8(key);
                cur = (lane & 7) == (x_rel & 7) ? colres : cur;
                if ((x_rel & 7) == 7) {
                    // lane 8 j + c: row j, output column 8 b + c of this strip
                    const int b = x_rel >> 3, xr = 8 * b + (lane & 7), j = lane >> 3;
                    const int dsel = ((-cur) - xr) & 63;
                    int score = (cur + xr + dsel) >> 6;  // 2 C - B, exact
                    int d = d0 + dsel;
                    if (chunk > 0) {
                        const int ps = res[b * 80 + lane], pd = reinterpret_cast<const int8_t *>(res + b * 80 + 64)[lane];
                        if (!(score > ps)) {  // the lower disparity wins ties
                            score = ps;
                            d = pd;
                        }
                    }
                    if (chunk + 1 < nchunks) {
                        res[b * 80 + lane] = score;
                        reinterpret_cast<int8_t *>(res + b * 80 + 64)[lane] = (int8_t)d;
                    } else {
                        const int y = ys + j, x = x0 + xr;
                        if (xr < a.X && x < a.cols && y < a.rows) {
                            if (SERIAL) {
                                if (score < -12000000) d = 0;  // no position inside the padded image: DisparitySSD.cpp:37
                            } else if (a.min_ssd_5e6) {
                                const int ssd = a.A[(size_t)y * a.cols + x] - score;
                                if (!(ssd < 5000000)) d = -1;  // DisparitySSD.cu:16,177
                            }
                            a.disp[(size_t)y * a.dstride + x] = (int8_t)d;
                        }
                    }
                }
            }
            rp += P;
            tp += P;
        }
    }
}

template <int R, int WC, bool SERIAL>
static int launch_search(hipStream_t s, const StereoExactArgs &a) {
    constexpr int XMAX = 128;
    const size_t lds = 4 * SxLayout<XMAX>::words(sx_groups(R)) * sizeof(uint32_t);
    auto k = stereo_exact_kernel<R, WC, SERIAL, XMAX>;
    static bool attr_set[16] = {false};  // per device: the launch needs more than 64 KB of dynamic LDS
    int dev = 0;
    MICV_HIP(hipGetDevice(&dev));
    if (dev < 16 && !attr_set[dev]) {
        MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    } else if (dev >= 16) {
        MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    k<<<cdiv(a.ntiles, 4), 256, lds, s>>>(a);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

template <int R>
static int launch_r(hipStream_t s, const StereoExactArgs &a, bool serial) {
    const int nstrips = cdiv(a.rows, SX_Y);
    stereo_pack_kernel<R><<<dim3(cdiv(a.cols, 256), nstrips), 256, 0, s>>>(a);
    stereo_energy8_kernel<R><<<dim3(cdiv(a.nB > a.cols ? a.nB : a.cols, 256), nstrips, 2), 256, 0, s>>>(a);
    MICV_LAUNCH_CHECK();
    const bool full = a.wcols == 2 * R + 1;
    if (serial) return full ? launch_search<R, 2 * R + 1, true>(s, a) : MICV_EUNSUPPORTED;
    if (full) return launch_search<R, 2 * R + 1, false>(s, a);
    if constexpr (R >= 1) return launch_search<R, 2 * R, false>(s, a);
    return MICV_EUNSUPPORTED;
}

}  // namespace

bool stereo_exact_covers(int rad, int flags, bool ncc) {
    if (ncc) return false;
    if (rad < 1 || rad > 7) return false;                       // (2r+1)^2 * 255^2 < 2^24
    if ((flags & MICV_STEREO_SERIAL) && rad > 5) return false;  // the invalid-position keys need a spare bit
    if ((flags & MICV_STEREO_COLS_2R) && rad < 2) return false;
    return true;
}

size_t stereo_exact_scratch(int rows, int cols, int rad, int min_d, int max_d) {
    const int nstrips = cdiv(rows, SX_Y), colsP = (cols + 63) & ~63, nB = cols + (max_d - min_d);
    return Carver::need((size_t)nstrips * cols * SX_LW, 4) + Carver::need((size_t)nstrips * sx_groups(rad) * colsP, 4) +
           Carver::need((size_t)rows * cols, 4) + Carver::need((size_t)rows * nB, 4);
}

int stereo_exact_launch(hipStream_t s, void *scratch, const float *left, const float *right, int rows, int cols,
                        int stride, int rad, int min_d, int max_d, int flags, int wcols, int8_t *disp, int dstride,
                        unsigned *flag, unsigned epoch, int wave_slots) {
    StereoExactArgs a;
    const int nstrips = cdiv(rows, SX_Y);
    a.left = left; a.right = right; a.stride = stride; a.rows = rows; a.cols = cols;
    a.min_d = min_d; a.max_d = max_d; a.wcols = wcols;
    a.colsP = (cols + 63) & ~63;
    a.nB = cols + (max_d - min_d);
    Carver cv(scratch);
    a.lplan = cv.take<uint32_t>((size_t)nstrips * cols * SX_LW);
    a.rpack = cv.take<uint32_t>((size_t)nstrips * sx_groups(rad) * a.colsP);
    a.A = cv.take<int32_t>((size_t)rows * cols);
    a.B = cv.take<int32_t>((size_t)rows * a.nB);
    a.flag = flag; a.epoch = epoch;
    a.disp = disp; a.dstride = dstride;
    a.min_ssd_5e6 = (flags & MICV_STEREO_MIN_SSD_5E6) ? 1 : 0;
    // Output columns per wave: a wave's work is X + wcols - 1 columns; the launch takes ceil(waves / slots) rounds of
    // the chip's resident waves.  Pick the strip count with the least (rounds x columns).
    const int XMAX = 128;
    int best_nxs = cdiv(cols, XMAX);
    long best_cost = -1;
    for (int nxs = cdiv(cols, XMAX); nxs <= 4 * (int)cdiv(cols, XMAX) && nxs <= cols; nxs++) {
        const int X = ((int)cdiv(cols, nxs) + 7) & ~7;
        if (X > XMAX) continue;
        const long waves = (long)nstrips * cdiv(cols, X);
        const long cost = ((waves + wave_slots - 1) / wave_slots) * (X + wcols - 1 + 12);  // + staging, in column units
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_nxs = cdiv(cols, X);
            a.X = X;
        }
    }
    if (best_cost < 0) a.X = XMAX, best_nxs = cdiv(cols, XMAX);
    a.nxs = best_nxs;
    a.ntiles = nstrips * a.nxs;
    const bool serial = flags & MICV_STEREO_SERIAL;
    switch (rad) {
        case 1: return launch_r<1>(s, a, serial);
        case 2: return launch_r<2>(s, a, serial);
        case 3: return launch_r<3>(s, a, serial);
        case 4: return launch_r<4>(s, a, serial);
        case 5: return launch_r<5>(s, a, serial);
        case 6: return launch_r<6>(s, a, serial);
        case 7: return launch_r<7>(s, a, serial);
    }
    return MICV_EUNSUPPORTED;
}

}  // namespace micv
