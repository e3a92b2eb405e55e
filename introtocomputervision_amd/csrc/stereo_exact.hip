// stereo_exact.hip -- ps2 window stereo on 8-bit-valued images (a12: disparitySSD), the exact-sum kernels.
//
// Every plain ps2 call hands over 8-bit images converted to CV_32F (ps2_cpp/src/main.cpp:87-88,115-117), and
// serial::disparitySSD sums integers by definition (DisparitySSD.cpp:45-51).  When both images hold integers in
// [0, 255] and the window has at most 15 x 15 taps, every partial sum of the contract (stereo.hip: column sums top
// -> bottom, window sum left -> right, f32) is an integer below 2^24: each float add is exact and the ORDER of the
// additions does not matter.  The cost can then be formed any way that gives the same integers:
//
//   SSD(y, x, d) = A(y, x) + B(y, x + d) - 2 C(y, x, d)
//     A = window sum of left^2, B = window sum of right^2 (one value per window POSITION, formed once per position
//     by the pre-pass -- the same sharing DisparityNCorr's energy field already uses), C = window sum of
//     left * right.  A does not depend on d, so arg min_d SSD = arg max_d (2 C - B).
//
// Kernel shape (stereo_exact_kernel): LANES ARE DISPARITIES.  One wave64 owns 8 output rows x X output columns and
// 64 consecutive disparities; it walks along x.  The left pixel of a column is then the same for all 64 lanes --
// a SCALAR operand: the pre-pass packs four rows of a column into one dword (bytes), the wave fetches the packed
// left words of the column with s_load (no VALU, no LDS), and the packed right words at column x + d0 + lane from
// its LDS strip (consecutive lanes, conflict-free).  v_dot4_u32_u8 multiplies four rows and accumulates in one
// instruction, so the 8 column sums of 2r + 1 rows cost 18 instructions (r = 5; rows outside a window are masked
// in the scalar operand, shared runs of full words are formed once).  The window sum slides along x IN the lane:
//   C += cs(x + 2r);  [use];  C -= cs(x)       (a ring of 2r + 1 column sums per row, statically indexed)
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
// (integer, 0..255) and writes the launch's epoch into the context's flag word on the first failure; the search launch
// holds the exact-sum tiles and, behind them, the float kernel's tiles of the same call (stereo_float.hpp) -- whichever
// the flag does not select leave at once.  MICV_OPT_STEREO_EXACT = -1 never takes this path.
#include <utility>

#include "kernels.hpp"
#include "stereo_exact.hpp"
#include "stereo_float.hpp"

namespace micv {

namespace {

typedef const __attribute__((address_space(4))) uint32_t *sx_cptr;  // scalar (constant-address-space) loads
typedef uint32_t sx_u32x8 __attribute__((ext_vector_type(8)));

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
#ifdef MICV_SX_NOMASK  // timing experiment only (wrong sums): what the scalar masks cost
    if (g0 < f0) acc = __builtin_amdgcn_udot4(l[(g0 + (k0 & 3)) % NG], r[g0], acc, false);
    if (g1 > f1) acc = __builtin_amdgcn_udot4(l[(g1 + 1 + (k1 & 3)) % NG], r[g1], acc, false);
#else
    if (g0 < f0) acc = __builtin_amdgcn_udot4(l[g0] & (0xFFFFFFFFu << (8 * (k0 & 3))), r[g0], acc, false);
    if (g1 > f1) acc = __builtin_amdgcn_udot4(l[g1] & (0xFFFFFFFFu >> (8 * (3 - (k1 & 3)))), r[g1], acc, false);
#endif
    return acc;
}

template <int R, int NG, size_t... J>
__device__ __forceinline__ void sx_colsums(const uint32_t (&l)[NG], const uint32_t (&r)[NG], uint32_t (&cs)[SX_Y],
                                           std::index_sequence<J...>) {
    ((cs[J] = sx_colsum<R, (int)J, NG>(l, r)), ...);
}

// ---- the pre-pass: pack rows into bytes, test eligibility, window energies -------------------------------------------
// Strip s = output rows 8 s .. 8 s + 7; its NR = 8 + 2 R window rows (clamped to the image) go four to a dword:
// word g, byte b = row 8 s - R + 4 g + b.  Left: plan[s][column + R][8 words], columns -R .. (clamped copies: the
// search kernel walks it with a plain pointer); right: pack[s][g][column], columns 0 .. cols - 1.
// Energies: B(y, p) = window sum of right^2 at position p = x + d (window columns p - R .. p - R + wcols - 1, each
// clamped on its own), p in [min_d, cols - 1 + max_d]; A(y, x) the same of left at x (only MIN_SSD_5E6 reads it).
// One workgroup = 256 consecutive columns q of one strip, of which it owns the first 256 - (wcols - 1).
template <int R>
__global__ __launch_bounds__(256) void stereo_prep_kernel(StereoExactArgs a) {
    constexpr int NR = SX_Y + 2 * R, NG = sx_groups(R), WMAX = 2 * R + 1;
    __shared__ uint32_t csB[SX_Y][256 + WMAX], csA[SX_Y][256 + WMAX];
    // ONE pass: a workgroup owns 256 - (wcols - 1) columns and its last wcols - 1 threads take the columns the windows that
    // start in them reach into (with 256 owned columns those few threads ran a second pass alone, every load of it
    // exposed: r06, 17.2 -> see profiles/r06/stereo_exact.md)
    const int nown = 256 - (a.wcols - 1);
    const int s = blockIdx.y, q0 = a.qlo + blockIdx.x * nown;
    const bool want_a = a.min_ssd_5e6 != 0;
    bool ok = true;
    {
        const int u = threadIdx.x;
        const int q = q0 + u, qc = clampi(q, 0, a.cols - 1);
        const bool own = u < nown;                                                  // this workgroup stores column q
        const bool need_l = (own && q + R >= 0 && q + R < a.lcols) || want_a;
        uint32_t lw[SX_LW], rw[NG];
#pragma unroll
        for (int g = 0; g < SX_LW; g++) lw[g] = 0;
#pragma unroll
        for (int g = 0; g < NG; g++) rw[g] = 0;
        float lv[NR], rv[NR];
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int yy = clampi(s * SX_Y - R + k, 0, a.rows - 1);
            rv[k] = a.right[(size_t)yy * a.stride + qc];
            lv[k] = need_l ? a.left[(size_t)yy * a.stride + qc] : 0.f;
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
        if (own && q + R >= 0 && q + R < a.lcols) {
            uint4 *lp = reinterpret_cast<uint4 *>(a.lplan + ((size_t)s * a.lcols + q + R) * SX_LW);
            lp[0] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
            lp[1] = make_uint4(lw[4], lw[5], lw[6], lw[7]);
        }
        if (own && q >= 0 && q < a.cols) {
#pragma unroll
            for (int g = 0; g < NG; g++) a.rpack[((size_t)s * NG + g) * a.colsP + q] = rw[g];
        }
        uint32_t cs[SX_Y];
        sx_colsums<R, NG>(rw, rw, cs, std::make_index_sequence<SX_Y>{});
#pragma unroll
        for (int j = 0; j < SX_Y; j++) csB[j][u] = cs[j];
        if (want_a) {
            uint32_t l5[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) l5[g] = lw[g];
            sx_colsums<R, NG>(l5, l5, cs, std::make_index_sequence<SX_Y>{});
#pragma unroll
            for (int j = 0; j < SX_Y; j++) csA[j][u] = cs[j];
        }
    }
    if (!ok) *a.flag = a.epoch;  // every failing thread stores the same word
    __syncthreads();
    const int p = q0 + (int)threadIdx.x + R;  // the window that starts at column q0 + t
    const bool mine = (int)threadIdx.x < nown;
    const bool b_ok = mine && p >= a.min_d && p - a.min_d < a.nB, a_ok = mine && want_a && p >= 0 && p < a.cols;
    if (!b_ok && !a_ok) return;
#pragma unroll
    for (int j = 0; j < SX_Y; j++) {
        const int y = s * SX_Y + j;
        if (y >= a.rows) break;
        uint32_t eb = 0, ea = 0;
        for (int i = 0; i < a.wcols; i++) eb += csB[j][threadIdx.x + i];
        if (b_ok) a.B[(size_t)y * a.nB + (p - a.min_d)] = (int32_t)eb;
        if (a_ok) {
            for (int i = 0; i < a.wcols; i++) ea += csA[j][threadIdx.x + i];
            a.A[(size_t)y * a.cols + p] = (int32_t)ea;
        }
    }
}

// ---- the search ------------------------------------------------------------------------------------------------
// Transposed max-reduction of the 8 keys of a column: lane group 8 j .. 8 j + 7 of the result holds row j's maximum.
// First level, partner 8 lanes away inside a DPP row: out[q] takes k[2q] on lanes with bit 3 clear, k[2q + 1] on the
// others (banks 2, 3 of every row).  The masked halves are one asm block: the compiler does not look inside asm for
// the two wait states a DPP read (in front) or a v_permlane read (behind) of a freshly written VGPR needs -- the
// s_nop at both ends are those.
__device__ __forceinline__ void sx_merge8x4(const int (&k)[SX_Y], int (&m)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) m[q] = sx_max(k[2 * q], sx_dpp<0x128>(k[2 * q], k[2 * q]));  // row_ror:8
    asm("s_nop 1\n\t"
        "v_max_i32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_max_i32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_max_i32_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_max_i32_dpp %3, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1"
        : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3])
        : "v"(k[1]), "v"(k[3]), "v"(k[5]), "v"(k[7]));
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
    int m[4];
    sx_merge8x4(k, m);
    int x = sx_merge32(sx_merge16(m[0], m[1]), sx_merge16(m[2], m[3]));
    x = sx_max(x, sx_dpp<0xB1>(x, x));   // quad_perm:[1,0,3,2]
    x = sx_max(x, sx_dpp<0x4E>(x, x));   // quad_perm:[2,3,0,1]
    x = sx_max(x, sx_dpp<0x141>(x, x));  // row_half_mirror: the other quad of the 8-lane group
    return x;
}

// One wave's LDS (dwords), position-major so that a column's words sit within a read's immediate offset:
//   right strip  entry i = NG words (stride padded odd: conflict-free), i = window column + lane
//   key table    entry i = 8 rows + 1 pad word,                        i = output column + lane
//   results      one key per lane and batch of 8 output columns (two chunks of disparities), or score + byte (more)
struct SxLds {
    int rs, npos_r, npos_t, nbatch, res_stride;
    __host__ __device__ int r_words() const { return npos_r * rs; }
    __host__ __device__ int t_words() const { return npos_t * 9; }
    __host__ __device__ int words() const { return r_words() + t_words() + nbatch * res_stride; }
};
__host__ __device__ inline SxLds sx_lds_layout(int NG, int X, int WC, int nchunks) {
    SxLds l;
    const int nouter = (X + WC - 1) / WC;
    l.rs = NG | 1;
    l.npos_r = nouter * WC + WC - 1 + 64;  // window columns of nouter * WC outputs, + 63 disparities, + 1
    l.npos_t = nouter * WC + 64;
    l.nbatch = (nouter * WC + 7) / 8;
    l.res_stride = nchunks <= 2 ? 64 : 80;
    return l;
}

#ifndef MICV_SX_WAVES
#define MICV_SX_WAVES 3
#endif
enum { SX_SSD = 0, SX_SERIAL = 1 };

// The launch holds BOTH searches of the call: workgroups [0, nblk_exact) are exact-sum tiles, the ones behind them the float
// kernel's tiles (stereo_float.hpp, RPW rows per wave) -- whichever the flag word does not select leave at once.  (As a
// separate launch behind this one the float kernel's empty pass cost 4.7 us per call on 8-bit-valued pairs.)
template <int R, int WC, int MODE, int RPW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MICV_SX_WAVES, MICV_SX_WAVES)))
void stereo_exact_kernel(StereoExactArgs a, StereoArgs f) {
    constexpr bool SERIAL = MODE == SX_SERIAL;
    constexpr int NG = sx_groups(R), P = WC - 1, RS = NG | 1, TS = 9;
    extern __shared__ uint32_t sx_lds[];
    const bool float_images = __builtin_nontemporal_load(a.flag) == a.epoch;  // a pixel was not an integer in 0..255
    if ((int)blockIdx.x >= a.nblk_exact) {
        if (!float_images) return;
        const int t = (int)blockIdx.x - a.nblk_exact;
        stereo_tile<R, SERIAL ? ST_SSD_SERIAL : ST_SSD, RPW, ST_DCH_DEFAULT>(f, reinterpret_cast<float *>(sx_lds), t % a.fgx, t / a.fgx);
        return;
    }
    if (float_images) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.ntiles) return;  // whole wave; waves never synchronise with each other
    const int s = tile / a.nxs, ys = s * SX_Y, x0 = (tile - s * a.nxs) * a.X;
    const int nchunks = (a.max_d - a.min_d) / 64 + 1;
    const SxLds lay = sx_lds_layout(NG, a.X, WC, nchunks);
    uint32_t *Rs = sx_lds + wave * lay.words();
    int32_t *Ts = reinterpret_cast<int32_t *>(Rs + lay.r_words());
    int32_t *res = Ts + lay.t_words();
    const int nouter = (a.X + WC - 1) / WC;  // the column loop runs nouter * WC output columns (overrun: masked)

    for (int chunk = 0; chunk < nchunks; chunk++) {
        const int d0 = a.min_d + 64 * chunk;
        const int nvalid = a.max_d - d0 + 1 < 64 ? a.max_d - d0 + 1 : 64;
        const int dl = lane < nvalid ? lane : nvalid - 1;  // lanes past max_d repeat the last disparity (same key)
        __builtin_amdgcn_wave_barrier();  // the previous chunk's reads are done (in-order LDS)
        // Stage: every load of the chunk is issued before the first is waited for.
        // right strip: entry i = packed column clamp(x0 - R + d0 + i)
        {
            constexpr int QR = 4;  // 4 x 64 entries >= npos_r (X <= 128)
            uint32_t v[QR][NG];
#pragma unroll
            for (int q = 0; q < QR; q++) {
                const int i = lane + 64 * q, c = clampi(x0 - R + d0 + i, 0, a.cols - 1);
#pragma unroll
                for (int g = 0; g < NG; g++) v[q][g] = a.rpack[((size_t)s * NG + g) * a.colsP + c];
            }
            // key table: entry i of row j = position p = x0 + d0 + i (output column x0 + x_rel seen by lane i - x_rel)
            constexpr int QT = 4;
            int32_t t[QT][SX_Y];
#pragma unroll
            for (int q = 0; q < QT; q++) {
                const int i = lane + 64 * q, p = x0 + d0 + i;
                const int pi = p - a.min_d < a.nB ? p - a.min_d : a.nB - 1;
#pragma unroll
                for (int j = 0; j < SX_Y; j++) {
                    const int y = ys + j < a.rows ? ys + j : a.rows - 1;
                    t[q][j] = a.B[(size_t)y * a.nB + pi];
                }
            }
#pragma unroll
            for (int q = 0; q < QR; q++) {
                const int i = lane + 64 * q;
                if (i < lay.npos_r) {
#pragma unroll
                    for (int g = 0; g < NG; g++) Rs[i * RS + g] = v[q][g];
                }
            }
#pragma unroll
            for (int q = 0; q < QT; q++) {
                const int i = lane + 64 * q, p = x0 + d0 + i;
                if (i < lay.npos_t) {
#pragma unroll
                    for (int j = 0; j < SX_Y; j++) {
                        int k = -(t[q][j] << 6) - i;
                        if (SERIAL && (p < -R || p > a.cols - 1 + R)) k = SX_INVALID - i;  // DisparitySSD.cpp:42-43
                        Ts[i * TS + j] = k;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const uint32_t *rp = Rs + dl * RS;
        const int32_t *tp = Ts + dl * TS;
        sx_cptr lp = (sx_cptr)(a.lplan + ((size_t)s * a.lcols + x0) * SX_LW);  // record of window column 0 = image column x0 - R
        // ring: the column sums of the window's columns, slot = window column mod WC; Cp = their sum without the
        // column that enters next
        uint32_t Cp[SX_Y], ring[WC][SX_Y];
#pragma unroll
        for (int j = 0; j < SX_Y; j++) Cp[j] = 0;
        auto column = [&](int c_loc, uint32_t (&cs)[SX_Y]) {  // c_loc: window column counted from where lp / rp point
            uint32_t lw[NG], rw[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) {
                lw[g] = lp[c_loc * SX_LW + g];
                rw[g] = rp[c_loc * RS + g];
            }
            sx_colsums<R, NG>(lw, rw, cs, std::make_index_sequence<SX_Y>{});
        };
#pragma unroll
        for (int k = 0; k < P; k++) {  // the first 2R (2R - 1) columns of the first window
            column(k, ring[k]);
#pragma unroll
            for (int j = 0; j < SX_Y; j++) Cp[j] += ring[k][j];
        }
        int cur = 0;
        // The packed left words of a column are fetched ONE COLUMN AHEAD, by hand: a scalar load shares its counter with
        // the LDS reads and returns out of order, so the compiler waits for everything (lgkmcnt(0)) at the first LDS use
        // behind one -- issued at the top of its own column the load's whole latency was exposed, once per column
        // (r06: 35 % of the wave's cycles).  Here it is issued behind the column's last LDS use (tied to a key) and
        // flies under the reduction; the wait sits in front of the next column's first dot product.  (The compiler's
        // own counted waits stay correct with one more operation in flight: they only wait longer.)
        // The packed right words come one column ahead too (LDS, issued with the column's key-table reads at its top).
        sx_u32x8 lraw;
        asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(lraw) : "s"(lp), "n"(P * SX_LW * 4));
        uint32_t rnext[NG];
        int tnext[SX_Y];
#pragma unroll
        for (int g = 0; g < NG; g++) rnext[g] = rp[P * RS + g];
#pragma unroll
        for (int j = 0; j < SX_Y; j++) tnext[j] = tp[j];
        for (int base = 0; base < nouter * WC; base += WC) {
#pragma unroll
            for (int m = 0; m < WC; m++) {
                const int x_rel = base + m;  // output column x0 + x_rel; its window: window columns x_rel .. x_rel + P
                uint32_t (&cs)[SX_Y] = ring[(m + P) % WC];  // the slot of the column that left one step ago
                int tk[SX_Y];
                {
                    uint32_t lw[NG], rw[NG];
                    sx_u32x8 lcur = lraw;
#pragma unroll
                    for (int g = 0; g < NG; g++) rw[g] = rnext[g];
                    // the wait: this column's left words (scalar) and right words (LDS), both asked for a column ago
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lcur), "+v"(rw[0]), "+v"(tnext[0]));
#pragma unroll
                    for (int j = 0; j < SX_Y; j++) tk[j] = tnext[j];
#pragma unroll
                    for (int g = 0; g < NG; g++) rnext[g] = rp[(m + 1 + P) * RS + g];
#pragma unroll
                    for (int j = 0; j < SX_Y; j++) tnext[j] = tp[(m + 1) * TS + j];
#pragma unroll
                    for (int g = 0; g < NG; g++) lw[g] = lcur[g];
                    sx_colsums<R, NG>(lw, rw, cs, std::make_index_sequence<SX_Y>{});
                }
                int key[SX_Y];
#pragma unroll
                for (int j = 0; j < SX_Y; j++) {
                    const uint32_t C = Cp[j] + cs[j];
                    key[j] = (int)(C << 7) + tk[j];
                    Cp[j] = C - ring[m][j];
                }
                // next column's record (the last step of the unrolled body reads past the pointer bump below)
                // (tied to the keys both ways: behind the last one's LDS operand, in front of the reduction that reads them)
                asm volatile("s_load_dwordx8 %0, %2, %3" : "=s"(lraw), "+v"(key[0]) : "s"(lp), "n"((m + 1 + P) * SX_LW * 4), "v"(key[SX_Y - 1]));
                const int colres = sx_reduce8(key);
                const unsigned long long sel = 0x0101010101010101ull << (x_rel & 7);  // lanes 8 j + (x_rel & 7)
                asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(cur) : "v"(colres), "s"(sel));
                if ((x_rel & 7) == 7) {
                    // lane 8 j + c: row j, output column 8 b + c of this strip
                    const int b = x_rel >> 3, xr = 8 * b + (lane & 7), j = lane >> 3;
                    int32_t *rb = res + b * lay.res_stride;
                    const int y = ys + j, x = x0 + xr;
                    const bool inside = xr < a.X && x < a.cols && y < a.rows;
                    auto decode = [&](int k, int dbase, int &score, int &d) {
                        const int dsel = ((-k) - xr) & 63;
                        score = (k + xr + dsel) >> 6;  // 2 C - B, exact
                        d = dbase + dsel;
                    };
                    int score, d;
                    decode(cur, d0, score, d);
                    if (chunk > 0) {
                        int ps, pd;
                        if (nchunks <= 2) {
                            decode(rb[lane], d0 - 64, ps, pd);
                        } else {
                            ps = rb[lane];
                            pd = reinterpret_cast<const int8_t *>(rb + 64)[lane];
                        }
                        if (!(score > ps)) {  // the lower disparity wins ties
                            score = ps;
                            d = pd;
                        }
                    }
                    if (chunk + 1 < nchunks) {
                        if (nchunks <= 2) {
                            rb[lane] = cur;
                        } else {
                            rb[lane] = score;
                            reinterpret_cast<int8_t *>(rb + 64)[lane] = (int8_t)d;
                        }
                    } else if (inside) {
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
            rp += WC * RS;
            tp += WC * TS;
            lp += WC * SX_LW;
        }
        // the last column's look-ahead load is still in flight: its registers are free for reuse only once it has landed
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lraw));
    }
}

template <int R, int WC, int MODE, int RPW>
static int launch_search(hipStream_t s, StereoExactArgs a, const StereoArgs &f) {
    const int nchunks = (a.max_d - a.min_d) / 64 + 1;
    const size_t lds_exact = 4 * (size_t)sx_lds_layout(sx_groups(R), a.X, WC, nchunks).words() * sizeof(uint32_t);
    const size_t lds_float = 4 * (size_t)(RPW + 2 * R) * (64 + ST_DCH_DEFAULT) * sizeof(float);
    const size_t lds = lds_exact > lds_float ? lds_exact : lds_float;
    a.nblk_exact = cdiv(a.ntiles, 4);
    a.fgx = cdiv(a.cols, 64 - 2 * R);
    const int nblk_float = a.fgx * (int)cdiv(a.rows, 4 * RPW);
    auto k = stereo_exact_kernel<R, WC, MODE, RPW>;
    static size_t attr_set[16] = {0};  // per device: the launch may need more than 64 KB of dynamic LDS
    int dev = 0;
    MICV_HIP(hipGetDevice(&dev));
    if (dev >= 16 || attr_set[dev] < lds) {
        MICV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev < 16) attr_set[dev] = lds;
    }
    k<<<a.nblk_exact + nblk_float, 256, lds, s>>>(a, f);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

template <int R, int RPW>
static int launch_r(hipStream_t s, const StereoExactArgs &a, const StereoArgs &f, bool serial) {
    const int nstrips = cdiv(a.rows, SX_Y);
    stereo_prep_kernel<R><<<dim3(cdiv(a.qhi - a.qlo, 256 - (a.wcols - 1)), nstrips), 256, 0, s>>>(a);
    MICV_LAUNCH_CHECK();
    const bool full = a.wcols == 2 * R + 1;
    if (serial) return full ? launch_search<R, 2 * R + 1, SX_SERIAL, RPW>(s, a, f) : MICV_EUNSUPPORTED;
    if (full) return launch_search<R, 2 * R + 1, SX_SSD, RPW>(s, a, f);
    return launch_search<R, 2 * R, SX_SSD, RPW>(s, a, f);
}

}  // namespace

bool stereo_exact_covers(int rad, int flags, bool ncc) {
    // disparityNCorr stays with the float kernels: its score's own roundings decide between candidates wherever the image is
    // nearly flat (the C3 pair: a quarter of the pixels), and an approximate-score search with an exact route for those
    // pixels -- built and bit-exact in r06 -- was slower than the float kernel on flat AND on textured 1080p pairs
    // (0.88 / 0.60 ms against 0.316; profiles/r06/stereo_exact.md)
    if (ncc) return false;
    // ROLLING is a compatibility mode with a kernel of its own (on 8-bit-valued images it equals the fresh sums, so nothing is
    // lost in results; its float fallback is not the tile the exact-sum launch carries)
    if (flags & MICV_STEREO_ROLLING) return false;
    if (rad < 1 || rad > 7) return false;                       // (2r+1)^2 * 255^2 < 2^24
    if ((flags & MICV_STEREO_SERIAL) && rad > 5) return false;  // the invalid-position keys need a spare bit
    if ((flags & MICV_STEREO_COLS_2R) && rad < 2) return false;
    return true;
}

// Output columns per wave: a wave's work is X + wcols - 1 columns (+ staging); the launch takes ceil(waves / slots)
// rounds of the chip's resident waves.  Picks the strip count with the least (rounds x columns).
static void sx_tiling(int rows, int cols, int rad, int wcols, int nchunks, int wave_slots3, int *X_out, int *nxs_out) {
    const int nstrips = cdiv(rows, SX_Y), XMAX = 128;
    long best_cost = -1;
    int bx = XMAX, bn = cdiv(cols, XMAX);
    for (int nxs = cdiv(cols, XMAX); nxs <= 6 * (int)cdiv(cols, XMAX) && nxs <= cols; nxs++) {
        const int X = ((int)cdiv(cols, nxs) + 7) & ~7;
        if (X > XMAX) continue;
        const long waves = (long)nstrips * cdiv(cols, X);
        // three waves per SIMD while a wave's LDS stays within a twelfth of the CU's 160 KB, else two
        const size_t lds = (size_t)sx_lds_layout(sx_groups(rad), X, wcols, nchunks).words() * 4;
        long slots = lds * 12 <= 160 * 1024 ? wave_slots3 : wave_slots3 / 3 * 2;
        if (MICV_SX_WAVES == 4) {
            if (lds * 16 > 160 * 1024) continue;
            slots = wave_slots3 / 3 * 4;
        }
        if (MICV_SX_WAVES == 2) slots = wave_slots3 / 3 * 2;
        // fewer waves per SIMD hide less: price a two-wave launch's column 15 % higher
        const long cost = ((waves + slots - 1) / slots) * (X + wcols - 1 + 10) * (slots >= wave_slots3 ? 100 : 115);
        if (best_cost < 0 || cost < best_cost) best_cost = cost, bx = X, bn = cdiv(cols, X);
    }
    *X_out = bx;
    *nxs_out = bn;
}

static void sx_geometry(StereoExactArgs &a, int rad, int wave_slots3) {
    const int nchunks = (a.max_d - a.min_d) / 64 + 1;
    sx_tiling(a.rows, a.cols, rad, a.wcols, nchunks, wave_slots3, &a.X, &a.nxs);
    a.ntiles = (int)cdiv(a.rows, SX_Y) * a.nxs;
    a.colsP = (a.cols + 63) & ~63;
    a.nB = a.cols + (a.max_d - a.min_d);
    a.lcols = a.nxs * a.X + 4 * rad + 8 + 2 * (2 * rad + 1);  // window columns of every strip incl. the loop's overrun
    // columns the pre-pass visits: the plan's (from -R), the pack's, and the window starts of every position p - R
    a.qlo = a.min_d - rad < -rad ? a.min_d - rad : -rad;
    const int qh1 = a.lcols - rad, qh2 = a.cols + a.max_d - rad;
    a.qhi = qh1 > qh2 ? qh1 : qh2;
    if (a.qhi < a.cols) a.qhi = a.cols;
}

size_t stereo_exact_scratch(int rows, int cols, int rad, int min_d, int max_d, int wcols, int flags, int wave_slots3) {
    StereoExactArgs a;
    a.rows = rows; a.cols = cols; a.min_d = min_d; a.max_d = max_d; a.wcols = wcols;
    sx_geometry(a, rad, wave_slots3);
    const int nstrips = cdiv(rows, SX_Y);
    return Carver::need((size_t)nstrips * a.lcols * SX_LW, 4) + Carver::need((size_t)nstrips * sx_groups(rad) * a.colsP, 4) +
           ((flags & MICV_STEREO_MIN_SSD_5E6) ? Carver::need((size_t)rows * cols, 4) : 0) + Carver::need((size_t)rows * a.nB, 4);
}

int stereo_exact_launch(hipStream_t s, void *scratch, const float *left, const float *right, int rows, int cols,
                        int stride, int rad, int min_d, int max_d, int flags, int wcols, int8_t *disp, int dstride,
                        unsigned *flag, unsigned epoch, int wave_slots3, const StereoArgs &f, bool rows10) {
    StereoExactArgs a;
    const int nstrips = cdiv(rows, SX_Y);
    a.left = left; a.right = right; a.stride = stride; a.rows = rows; a.cols = cols;
    a.min_d = min_d; a.max_d = max_d; a.wcols = wcols;
    sx_geometry(a, rad, wave_slots3);
    Carver cv(scratch);
    a.lplan = cv.take<uint32_t>((size_t)nstrips * a.lcols * SX_LW);
    a.rpack = cv.take<uint32_t>((size_t)nstrips * sx_groups(rad) * a.colsP);
    a.A = (flags & MICV_STEREO_MIN_SSD_5E6) ? cv.take<int32_t>((size_t)rows * cols) : nullptr;  // only the 5e6 threshold reads it
    a.B = cv.take<int32_t>((size_t)rows * a.nB);
    a.flag = flag; a.epoch = epoch;
    a.disp = disp; a.dstride = dstride;
    a.min_ssd_5e6 = (flags & MICV_STEREO_MIN_SSD_5E6) ? 1 : 0;
    const bool serial = flags & MICV_STEREO_SERIAL;
#define MICV_SX_CASE(RR) \
    case RR: return rows10 ? launch_r<RR, 10>(s, a, f, serial) : launch_r<RR, 8>(s, a, f, serial);
    switch (rad) {
        MICV_SX_CASE(1) MICV_SX_CASE(2) MICV_SX_CASE(3) MICV_SX_CASE(4) MICV_SX_CASE(5) MICV_SX_CASE(6) MICV_SX_CASE(7)
    }
#undef MICV_SX_CASE
    return MICV_EUNSUPPORTED;
}

}  // namespace micv
