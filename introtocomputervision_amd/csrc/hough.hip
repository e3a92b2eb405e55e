// hough.hip -- ps1: Hough line / circle accumulators (a14, a15) and peak finding (a16).
//
//  * edge mask -> ordered (row-major) point list: compact.hpp (replaces K1 + thrust::copy_if,
//    Hough.cu:173-232, without the two full-image uint2 scratch vectors)
//  * lines: one workgroup per (theta bin, point chunk) votes into a rho histogram that lives
//    in LDS (<= 16 K bins) and is flushed once with global integer atomics -- the reference
//    issues one global atomic per (point, theta) (Hough.cu:57).  Integer sums: order-free.
//  * trig: host-built float tables (correctly rounded double cos/sin of the float radian,
//    cast to float) replace the reference's __sincosf (Hough.cu:53), which is a hardware
//    approximation and not reproducible off NVIDIA parts.
//  * peaks: up/left 2x2 "local max" flags -> ordered candidate list -> the top num_peaks by
//    (votes desc, index asc) -- the order thrust::stable_sort(greater) produces (Hough.cu:402).
#include <cmath>

#include "compact.hpp"
#include "kernels.hpp"

namespace micv {

struct MaskPred {
    const uint8_t *mask;
    int cols;
    size_t stride;
    __device__ bool operator()(int64_t i) const {
        const int y = (int)(i / cols), x = (int)(i - (int64_t)y * cols);
        return mask[(size_t)y * stride + x] != 0;  // IsNonzero, Hough.cu:183-187
    }
};

// The ordered point list by row-segment masks (r04, as the corner list in harris.hip): a wave reads one 64-pixel
// segment of a mask row, its ballot is that segment's 64-bit word, and the list is a chained scan over the words
// (compact_masks_onepass_kernel) -- 32 k words at 1080p where the byte form scanned 2 M flags in three launches.
__global__ __launch_bounds__(256) void mask_words_kernel(const uint8_t *__restrict__ mask, size_t stride, int rows, int cols,
                                                          int tiles_x, unsigned long long *__restrict__ words) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= rows) return;  // (wave-uniform)
    const bool on = x < cols && mask[(size_t)y * stride + x] != 0;  // IsNonzero, Hough.cu:183-187
    const unsigned long long w = __ballot(on);
    if ((threadIdx.x & 63) == 0) words[(size_t)y * tiles_x + blockIdx.x] = w;
}
// word i = row i / tiles_x, columns 64 (i % tiles_x) ..: bit b is the point with linear index y * cols + x
struct MaskIndexEmit {
    int32_t *out;
    int tiles_x, cols;
    __device__ void operator()(int64_t pos, int64_t i, int bit) const {
        const int y = (int)(i / tiles_x);
        out[pos] = y * cols + 64 * (int)(i - (int64_t)y * tiles_x) + bit;
    }
};

struct TrigTable {
    float c[360], s[360];
};

// degToRad (Hough.cu:20-24): float theta * double PI / 180.f -> float.
static TrigTable make_trig(int theta0) {
    TrigTable t;
    for (int i = 0; i < 360; i++) {
        const float rad = (float)((double)(float)(theta0 + i) * 3.14159265 / 180.f);
        t.c[i] = (float)std::cos((double)rad);
        t.s[i] = (float)std::sin((double)rad);
    }
    return t;
}

// One workgroup per theta: the whole rho column of that angle is a histogram in LDS (dynamic
// LDS = rho_bins ints), every edge point votes into it with an LDS atomic, and the column is then
// written once with plain stores -- zeros included, so neither global atomics nor a memset of the
// accumulator are needed (a chunked grid does not aggregate: with ~1 vote per bin per chunk its
// flush was as many global atomics as there are votes).
__global__ __launch_bounds__(1024) void hough_lines_kernel(const int32_t *__restrict__ pts,
                                                            const int64_t *__restrict__ npts_p,
                                                            int cols, int row0,
                                                            const float *__restrict__ ct,
                                                            const float *__restrict__ st,
                                                            float diag, unsigned rho_bin,
                                                            unsigned theta_bin, int rho_bins,
                                                            int theta_bins,
                                                            int32_t *__restrict__ acc) {
    extern __shared__ int hist[];
    const int tb = blockIdx.x;
    const int theta = -90 + tb * (int)theta_bin;  // Hough.cu:51
    for (int i = threadIdx.x; i < rho_bins; i += 1024) hist[i] = 0;
    __syncthreads();
    const float c = ct[theta + 90], s = st[theta + 90];
    const int thetaBin = (int)roundf(((float)theta - -90.f) / (float)theta_bin);  // :56
    const int64_t npts = *npts_p;
    for (int64_t i = threadIdx.x; i < npts; i += 1024) {
        const int32_t p = pts[i];
        const int yl = p / cols, x = p - yl * cols, y = yl + row0;
        const float rho = roundf((float)x * c + (float)y * s) + diag;  // :54
        const int rhoBin = (int)roundf(rho / (float)rho_bin);            // :55
        if ((unsigned)rhoBin < (unsigned)rho_bins) atomicAdd(&hist[rhoBin], 1);
    }
    __syncthreads();
    if ((unsigned)thetaBin >= (unsigned)theta_bins) return;
    for (int i = threadIdx.x; i < rho_bins; i += 1024) acc[(size_t)i * theta_bins + thetaBin] = hist[i];
}

// Fallback for accumulators whose rho axis does not fit LDS: global atomics per vote.
__global__ __launch_bounds__(256) void hough_lines_global_kernel(
    const int32_t *__restrict__ pts, const int64_t *__restrict__ npts_p, int cols, int row0,
    const float *__restrict__ ct, const float *__restrict__ st, float diag, unsigned rho_bin,
    unsigned theta_bin, int rho_bins, int theta_bins, int32_t *__restrict__ acc) {
    const int64_t npts = *npts_p;
    const int tb = blockIdx.y;
    const int theta = -90 + tb * (int)theta_bin;
    const float c = ct[theta + 90], s = st[theta + 90];
    const int thetaBin = (int)roundf(((float)theta - -90.f) / (float)theta_bin);
    if ((unsigned)thetaBin >= (unsigned)theta_bins) return;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < npts; i += (int64_t)gridDim.x * 256) {
        const int32_t p = pts[i];
        const int yl = p / cols, x = p - yl * cols, y = yl + row0;
        const float rho = roundf((float)x * c + (float)y * s) + diag;
        const int rhoBin = (int)roundf(rho / (float)rho_bin);
        if ((unsigned)rhoBin < (unsigned)rho_bins)
            atomicAdd(&acc[(size_t)rhoBin * theta_bins + thetaBin], 1);
    }
}

// float -> unsigned the way the reference's device code converts it (saturating truncation).
__device__ __forceinline__ unsigned f2u_sat(float v) {
    if (!(v > 0.f)) return 0u;
    if (v >= 4294967296.f) return 0xFFFFFFFFu;
    return (unsigned)v;
}

// Votes of Hough.cu:85-93, gathered per accumulator tile: a 64x32 tile of the accumulator lives in LDS, the
// workgroup walks the edge points that can reach it (the point list is in row-major order, so the
// rows [b0 - r - 1, b0 + 32 + r + 1] are one contiguous range found by binary search; columns are
// filtered into an LDS list), every thread owns one or two angles, votes are LDS atomics, and the
// tile is written once with plain stores -- no global atomics and no memset of the accumulator.
// Integer counts: identical to the scatter form in any order.
__global__ __launch_bounds__(1024) void hough_circles_tiled_kernel(
    const int32_t *__restrict__ pts, const int64_t *__restrict__ npts_p, int rows, int cols, int row0,
    const float *__restrict__ ct, const float *__restrict__ st, float radius, int reach,
    int32_t *__restrict__ acc) {
    // 16 waves per tile: the launch lasts as long as its heaviest tile's vote loop (r05: 512 threads 54.6 us, 256: 86.9, against 41.5)
    constexpr int TA = 64, TB = 32, CH = 2048, NT = 1024;
    __shared__ int hist[TA * TB];
    __shared__ int list[4 * CH];  // (point, quadrant of the angle) entries: up to four per point
    __shared__ int nlist;
    __shared__ long long range[2];
    const int tid = threadIdx.x;
    const int a0 = blockIdx.x * TA, b0 = blockIdx.y * TB;
    for (int i = tid; i < TA * TB; i += NT) hist[i] = 0;
    if (tid < 128) {
        // first point with local row >= yl (wave 0: lowest row that reaches the tile; wave 1: one past the highest).
        // A 64-ary search by the whole wave: three dependent loads for 19 k points where the one-lane bisection took
        // fourteen -- at ~0.8 us each that chain was most of a tile's time (r05: 44 -> see profiles/r05/hough_chain.txt).
        const int wv = tid >> 6, ln = tid & 63;
        const long long yl = wv == 0 ? (long long)b0 - reach - row0 : (long long)b0 + TB + reach - row0;
        const long long key = yl <= 0 ? 0 : yl * cols;
        long long lo = 0, hi = *npts_p;  // the answer is in [lo, hi]
        while (hi - lo > 64) {
            const long long stride = (hi - lo + 63) >> 6;
            const long long at = lo + (ln + 1) * stride - 1;  // the last element of this lane's segment
            const bool below = at < hi && (long long)pts[at] < key;
            const int c = __popcll(__ballot(below));  // whole segments below the key (the predicate is monotone)
            lo += c * stride;
            hi = lo + stride < hi ? lo + stride : hi;
        }
        const bool below = lo + ln < hi && (long long)pts[lo + ln] < key;
        lo += __popcll(__ballot(below));
        if (ln == 0) range[wv] = lo;
    }
    __shared__ float tc[360], ts[360];
    for (int i = tid; i < 360; i += NT) {
        tc[i] = ct[i];
        ts[i] = st[i];
    }
    // cells of this tile a vote may land in (Hough.cu:89: 0 < a < cols, 0 < b < rows)
    const unsigned amin = a0 > 1 ? a0 : 1, bmin = b0 > 1 ? b0 : 1;
    const unsigned amax = a0 + TA < cols ? a0 + TA : cols, bmax = b0 + TB < rows ? b0 + TB : rows;
    const unsigned aw = amax > amin ? amax - amin : 0, bw = bmax > bmin ? bmax - bmin : 0;
    __syncthreads();
    const long long lo = range[0], hi = range[1];
    for (long long base = lo; base < hi; base += CH) {
        if (tid == 0) nlist = 0;
        __syncthreads();
        const long long end = base + CH < hi ? base + CH : hi;
        for (long long i = base + tid; i < end; i += NT) {
            const int p = pts[i];
            const int yl = p / cols, x = p - yl * cols;
            if (x >= a0 - reach && x < a0 + TA + reach) {
                // Only the quadrants of the circle that can reach the tile are listed (r05): at angle t the centre lies at
                // (x - r cos t, y - r sin t), so over t in [90 q, 90 q + 90) it stays on one side of the point in each axis --
                // left of it (cos >= 0: q = 0, 3; the truncated column is in [x - reach, x]) or right of it (q = 1, 2:
                // [x - 1, x + reach], the - 1 for table entries that are a rounding error off zero), likewise above
                // (q = 0, 1) or below.  A quadrant is listed unless the tile lies wholly on the other side; the exact
                // test of every vote below is unchanged, so the filter only has to be conservative.  It drops about half
                // of the (point, angle) pairs a tile used to evaluate.
                const int y = yl + row0;
                const bool lft = x >= a0, rgt = x <= a0 + TA, up = y >= b0, dwn = y <= b0 + TB;
                const int e = (y << 15) | x;  // (rows, cols <= 32767: 15 bits each, the quadrant on top)
                if (lft && up) list[atomicAdd(&nlist, 1)] = e;
                if (rgt && up) list[atomicAdd(&nlist, 1)] = e | (1 << 30);
                if (rgt && dwn) list[atomicAdd(&nlist, 1)] = e | (2 << 30);
                if (lft && dwn) list[atomicAdd(&nlist, 1)] = e | (3 << 30);
            }
        }
        __syncthreads();
        // (point, angle) pairs dealt flat over the workgroup: every lane busy, one vote per trip
        const int nv = nlist * 90;
        for (int w = tid; w < nv; w += NT) {
            const int k = w / 90;
            const unsigned e = (unsigned)list[k];
            const int t = (w - k * 90) + 90 * (int)(e >> 30);
            const float fx = (float)(e & 0x7FFF), fy = (float)((e >> 15) & 0x7FFF);
            // v_cvt_u32_f32 saturates (negative and NaN -> 0, >= 2^32 -> 0xFFFFFFFF): it IS f2u_sat
            unsigned a, b;
            const float va = fx - radius * tc[t], vb = fy - radius * ts[t];
            asm("v_cvt_u32_f32 %0, %1" : "=v"(a) : "v"(va));
            asm("v_cvt_u32_f32 %0, %1" : "=v"(b) : "v"(vb));
            const unsigned la = a - amin, lb = b - bmin;
            if (la < aw && lb < bw) atomicAdd(&hist[(b - (unsigned)b0) * TA + (a - (unsigned)a0)], 1);
        }
        __syncthreads();
    }
    for (int i = tid; i < TA * TB; i += NT) {
        const int a = a0 + (i & (TA - 1)), b = b0 + i / TA;
        if (a < cols && b < rows) acc[(size_t)b * cols + a] = hist[i];
    }
}

// Hough.cu:148-157 + MaskAndThreshold :239-249.
struct PeakPred {
    const int32_t *acc;
    int rows, cols, threshold;
    // Two steps for the one-launch compaction: a workgroup's 16 centre loads per thread go out together, the rare
    // cells at or above the threshold then look at their neighbours (as one call the 16 loads ran one after the other,
    // each behind the previous cell's branch: 15.3 -> see profiles/r05/hough_chain.txt).
    __device__ int center(int64_t i) const { return acc[i]; }
    __device__ bool test(int64_t i, int v) const {
        if (v < threshold) return false;
        const uint32_t ii = (uint32_t)i;  // (the entry point admits fewer than 2^31 cells: a 32-bit quotient)
        const int ty = (int)(ii / (uint32_t)cols), tx = (int)(ii - (uint32_t)ty * (uint32_t)cols);
        const int y1 = rows - 1 < ty + 1 ? rows - 1 : ty + 1;  // exclusive bounds, as written
        const int x1 = cols - 1 < tx + 1 ? cols - 1 : tx + 1;
        for (int y = ty - 1 > 0 ? ty - 1 : 0; y < y1; y++)
            for (int x = tx - 1 > 0 ? tx - 1 : 0; x < x1; x++)
                if (acc[(size_t)y * cols + x] > v) return false;
        return true;
    }
    __device__ bool operator()(int64_t i) const { return test(i, acc[i]); }
};

// Selection round k: the largest key strictly below the key chosen in round k-1.
// key = (votes biased to unsigned) << 32 | (0xFFFFFFFF - index): larger = more votes, then
// smaller index -- the stable descending order.
__device__ __forceinline__ unsigned long long peak_key(int votes, uint32_t idx) {
    return ((unsigned long long)((uint32_t)votes ^ 0x80000000u) << 32) | (0xFFFFFFFFu - idx);
}

__global__ __launch_bounds__(256) void peak_select_kernel(const int32_t *__restrict__ acc,
                                                           const int32_t *__restrict__ cand,
                                                           const int64_t *__restrict__ ncand_p,
                                                           int64_t cap,
                                                           unsigned long long *__restrict__ sel,
                                                           int k) {
    __shared__ unsigned long long wmax[4];
    const int64_t n = *ncand_p < cap ? *ncand_p : cap;
    const unsigned long long bound = k == 0 ? ~0ull : sel[k - 1];
    unsigned long long m = 0;
    if (k == 0 || bound != 0) {
        for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const uint32_t idx = (uint32_t)cand[i];
            const unsigned long long key = peak_key(acc[idx], idx);
            if (key < bound && key > m) m = key;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(m, d);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) m = wmax[w] > m ? wmax[w] : m;
        if (m) atomicMax(&sel[k], m);
    }
}

// Unsigned max over the wave by DPP (quad permutes, the two row mirrors, row_bcast15 / row_bcast31; lane 63 ends up
// with the total): six dependent VALU steps instead of six LDS-crossbar round trips of __shfl_xor -- the selection rounds
// are one wave's serial chain, so the latency of the reduction is the kernel's time.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define MICV_DPP_MAX(ctrl, rmask)                                                                          \
    {                                                                                                      \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xF, false); \
        v = o > v ? o : v;                                                                                 \
    }
    MICV_DPP_MAX(0xB1, 0xF)   // quad_perm [1,0,3,2]
    MICV_DPP_MAX(0x4E, 0xF)   // quad_perm [2,3,0,1]
    MICV_DPP_MAX(0x141, 0xF)  // row_half_mirror
    MICV_DPP_MAX(0x140, 0xF)  // row_mirror: every lane of a row of 16 holds the row's max
    MICV_DPP_MAX(0x142, 0xA)  // row_bcast15 into rows 1 and 3
    MICV_DPP_MAX(0x143, 0xC)  // row_bcast31 into rows 2 and 3
#undef MICV_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// The largest of the wave's 64-bit keys (every lane returns it): the high words first, then the low words of the
// lanes that hold that high word -- the lexicographic order of (hi, lo) is the order of the keys.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long m) {
    const unsigned hi = (unsigned)(m >> 32), lo = (unsigned)m;
    const unsigned H = wave_max_u32(hi);
    const unsigned L = wave_max_u32(hi == H ? lo : 0u);
    return ((unsigned long long)H << 32) | L;
}

// All selection rounds in ONE launch by a single 1024-thread workgroup (num_peaks <= 64): a round
// per launch costs more in launch latency than the scan itself when the candidate list is the
// usual few hundred entries.
// Up to 4096 candidates (r05): the keys are distinct, so the K largest of the list are among the K largest of each
// wave's 256-key slice -- every wave selects its own top K from registers (K rounds of a wave reduction, no workgroup
// barrier), the 16 x K survivors go to LDS, and wave 0 runs the K rounds again over them: two barriers in all instead
// of two per round (top-10 of a few hundred candidates: 16 -> 6 us).  Longer lists rescan the candidates every round.
__global__ __launch_bounds__(1024) void peak_select_all_kernel(
    const int32_t *__restrict__ acc, const int32_t *__restrict__ cand,
    const int64_t *__restrict__ ncand_p, int64_t cap, unsigned num_peaks, int cols,
    uint32_t *__restrict__ peaks_rc, int64_t *__restrict__ count) {
    constexpr int KEEP = 4096, KMAX = 64;
    __shared__ unsigned long long top[16 * KMAX];
    __shared__ unsigned long long wmax[16];
    const int64_t n = *ncand_p < cap ? *ncand_p : cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (n <= KEEP) {
        unsigned long long key[4];
        uint32_t idx[4];
        int votes[4];
        // the four candidate loads together, then the four accumulator loads (clamped indices: no branch between them)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t i = wave * 256 + j * 64 + lane;
            idx[j] = n > 0 ? (uint32_t)cand[i < n ? i : n - 1] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) votes[j] = acc[idx[j]];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t i = wave * 256 + j * 64 + lane;
            key[j] = i < n ? peak_key(votes[j], idx[j]) : 0ull;  // (a real key is never 0: its low word is 2^32 - 1 - index)
        }
        unsigned long long bound = ~0ull;
        for (unsigned k = 0; k < num_peaks; k++) {  // the wave's own top K, descending; 0 once it runs out
            unsigned long long m = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) m = key[j] < bound && key[j] > m ? key[j] : m;
            m = wave_max_u64(m);
            if (lane == 0) top[wave * KMAX + k] = m;
            bound = m ? m : 1ull;  // (nothing is below 1)
        }
        __syncthreads();
        if (wave != 0) return;
        // wave 0: K rounds over the 16 x K survivors, K / 4 per lane (wave w's k-th key sits at top[w * KMAX + k])
        bound = ~0ull;
        int64_t found = 0;
        for (unsigned k = 0; k < num_peaks; k++) {
            unsigned long long m = 0;
            for (unsigned t = lane; t < 16 * num_peaks; t += 64) {
                const unsigned long long c = top[(t / num_peaks) * KMAX + t % num_peaks];
                m = c < bound && c > m ? c : m;
            }
            m = wave_max_u64(m);
            if (m == 0) break;
            if (lane == 0) {
                const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(m & 0xFFFFFFFFull);
                peaks_rc[2 * found] = idx / (uint32_t)cols;      // rho = row
                peaks_rc[2 * found + 1] = idx % (uint32_t)cols;  // theta = col
            }
            bound = m;
            found++;
        }
        if (lane == 0) *count = found;
        return;
    }
    unsigned long long bound = ~0ull;
    int64_t found = 0;
    for (unsigned k = 0; k < num_peaks; k++) {
        unsigned long long m = 0;
        for (int64_t i = threadIdx.x; i < n; i += 1024) {
            const uint32_t idx = (uint32_t)cand[i];
            const unsigned long long key = peak_key(acc[idx], idx);
            if (key < bound && key > m) m = key;
        }
        m = wave_max_u64(m);
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        m = wmax[0];
#pragma unroll
        for (int w = 1; w < 16; w++) m = wmax[w] > m ? wmax[w] : m;
        __syncthreads();  // wmax is rewritten next round
        if (m == 0) break;
        if (threadIdx.x == 0) {
            const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(m & 0xFFFFFFFFull);
            peaks_rc[2 * found] = idx / (uint32_t)cols;      // rho = row
            peaks_rc[2 * found + 1] = idx % (uint32_t)cols;  // theta = col
        }
        bound = m;
        found++;
    }
    if (threadIdx.x == 0) *count = found;
}

__global__ void peak_emit_kernel(const unsigned long long *__restrict__ sel, unsigned num_peaks,
                                 int cols, uint32_t *__restrict__ peaks_rc,
                                 int64_t *__restrict__ count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t n = 0;
    for (unsigned k = 0; k < num_peaks; k++) {
        if (sel[k] == 0) break;
        const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(sel[k] & 0xFFFFFFFFull);
        peaks_rc[2 * n] = idx / (uint32_t)cols;      // rho = row
        peaks_rc[2 * n + 1] = idx % (uint32_t)cols;  // theta = col
        n++;
    }
    *count = n;
}

static size_t hough_diag(int rows, int cols) {
    return (size_t)std::ceil(std::sqrt((double)(rows * rows + cols * cols)));  // Hough.cu:258-259
}

}  // namespace micv

using namespace micv;

extern "C" {

int micv_hough_lines_dims(int rows, int cols, unsigned rho_bin, unsigned theta_bin, int *rho_bins,
                          int *theta_bins) {
    MICV_REQUIRE(rows > 0 && cols > 0 && rows <= 32767 && cols <= 32767 && rho_bin > 0 &&
                     theta_bin > 0 && rho_bins && theta_bins,
                 "micv_hough_lines_dims: bad argument");
    const size_t max_dist = hough_diag(rows, cols);
    const size_t rb = (size_t)std::ceil((float)(2 * max_dist) / (float)rho_bin);  // :260
    const size_t tb = (size_t)std::ceil(180.f / (float)theta_bin);                // :261-262
    *rho_bins = (int)(rb < 1 ? 1 : rb);
    *theta_bins = (int)(tb < 1 ? 1 : tb);
    return MICV_OK;
}

// Device copy of a trig table, uploaded on the context's first use (a pageable H2D copy per call
// costs more than the vote kernel).
static int trig_on_device(micv_ctx *ctx, int which, int theta0, const TrigTable **out) {
    if (!ctx->trig_tables[which]) {
        const TrigTable ht = make_trig(theta0);
        void *p = nullptr;
        MICV_HIP(hipMalloc(&p, sizeof(TrigTable)));
        if (hipMemcpy(p, &ht, sizeof(ht), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(p);
            set_error("hough: trig table upload failed");
            return MICV_EHIP;
        }
        ctx->trig_tables[which] = p;
    }
    *out = static_cast<const TrigTable *>(ctx->trig_tables[which]);
    return MICV_OK;
}

static int hough_points(micv_ctx *ctx, hipStream_t s, const uint8_t *mask, int rows, int cols,
                        size_t mstride, size_t extra_bytes, int32_t **pts, int64_t **npts,
                        char **extra) {
    const int64_t n = (int64_t)rows * cols;
    const int tiles_x = cdiv(cols, 64);
    const int64_t nwords = (int64_t)rows * tiles_x;
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need(n, 4) + Carver::need(1, 8) + Carver::need(nwords, 8) + compact_scratch_bytes(n) +
                              ((extra_bytes + 255) & ~size_t(255)),
                          &scratch));
    Carver c(scratch);
    *pts = c.take<int32_t>(n);
    *npts = c.take<int64_t>(1);
    *extra = c.take<char>(extra_bytes);
    unsigned long long *words = c.take<unsigned long long>(nwords);
    const int nchunks = compact_masks_chunks(nwords);
    unsigned long long *status = nullptr;
    unsigned *counters = nullptr;
    if (ctx->opt[MICV_OPT_COMPACT_3PASS] <= 0 && ctx->compact_state(s, nchunks, &status, &counters) == MICV_OK) {
        mask_words_kernel<<<dim3(tiles_x, cdiv(rows, 4)), 256, 0, s>>>(mask, mstride, rows, cols, tiles_x, words);
        MICV_LAUNCH_CHECK();
        launch_compact_masks(s, words, MaskIndexEmit{*pts, tiles_x, cols}, nwords, status, counters, n, *npts);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    return ordered_compact3(s, MaskPred{mask, cols, mstride}, IndexEmit{*pts}, n, n, *npts, c.base + c.off);
}

int micv_hough_lines_dev(micv_ctx *ctx, const uint8_t *mask, int rows, int cols, size_t mstride,
                         unsigned rho_bin, unsigned theta_bin, int32_t *acc, micv_stream stream) {
    return micv_hough_lines_band_dev(ctx, mask, rows, cols, mstride, 0, rows, rho_bin, theta_bin, acc,
                                     stream);
}

int micv_hough_lines_band_dev(micv_ctx *ctx, const uint8_t *mask, int band_rows, int cols,
                              size_t mstride, int row0, int rows, unsigned rho_bin,
                              unsigned theta_bin, int32_t *acc, micv_stream stream) {
    MICV_REQUIRE(ctx && mask && acc, "micv_hough_lines: null argument");
    MICV_REQUIRE(mstride >= (size_t)cols, "micv_hough_lines: bad stride");
    MICV_REQUIRE(band_rows > 0 && row0 >= 0 && row0 + band_rows <= rows,
                 "micv_hough_lines: band [%d, %d) outside the %d-row image", row0, row0 + band_rows, rows);
    int rb, tb;
    MICV_TRY(micv_hough_lines_dims(rows, cols, rho_bin, theta_bin, &rb, &tb));
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int32_t *pts;
    int64_t *npts;
    char *extra;
    MICV_TRY(hough_points(ctx, s, mask, band_rows, cols, mstride, 0, &pts, &npts, &extra));
    const TrigTable *dt;
    MICV_TRY(trig_on_device(ctx, 0, -90, &dt));  // theta = -90..89, Hough.cu:51
    const float diag = (float)hough_diag(rows, cols);
    // theta loop `for (theta = -90; theta < 90; theta += bin)` has ceil(180/bin) iterations = tb
    const int n_theta = (int)((180 + theta_bin - 1) / theta_bin);
    if ((size_t)rb * 4 <= 64 * 1024 && n_theta == tb) {
        hough_lines_kernel<<<n_theta, 1024, (size_t)rb * 4, s>>>(
            pts, npts, cols, row0, dt->c, dt->s, diag, rho_bin, theta_bin, rb, tb, acc);
    } else {
        MICV_HIP(hipMemsetAsync(acc, 0, (size_t)rb * tb * sizeof(int32_t), s));
        hough_lines_global_kernel<<<dim3(64, n_theta), 256, 0, s>>>(
            pts, npts, cols, row0, dt->c, dt->s, diag, rho_bin, theta_bin, rb, tb, acc);
    }
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_hough_circles_dev(micv_ctx *ctx, const uint8_t *mask, int rows, int cols, size_t mstride,
                           unsigned radius, int32_t *acc, micv_stream stream) {
    return micv_hough_circles_band_dev(ctx, mask, rows, cols, mstride, 0, rows, radius, acc, stream);
}

int micv_hough_circles_band_dev(micv_ctx *ctx, const uint8_t *mask, int band_rows, int cols,
                                size_t mstride, int row0, int rows, unsigned radius, int32_t *acc,
                                micv_stream stream) {
    MICV_REQUIRE(ctx && mask && acc, "micv_hough_circles: null argument");
    MICV_REQUIRE(band_rows > 0 && row0 >= 0 && row0 + band_rows <= rows,
                 "micv_hough_circles: band [%d, %d) outside the %d-row image", row0, row0 + band_rows, rows);
    MICV_REQUIRE(rows > 0 && cols > 0 && rows <= 32767 && cols <= 32767,
                 "micv_hough_circles: bad size %dx%d", rows, cols);
    MICV_REQUIRE(mstride >= (size_t)cols, "micv_hough_circles: bad stride");
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int32_t *pts;
    int64_t *npts;
    char *extra;
    MICV_TRY(hough_points(ctx, s, mask, band_rows, cols, mstride, 0, &pts, &npts, &extra));
    const TrigTable *dt;
    MICV_TRY(trig_on_device(ctx, 1, 0, &dt));  // theta = 0..359, Hough.cu:85
    // every accumulator cell is written by its tile (zeros included; the reference forgets to
    // clear, Hough.cu:318)
    const int reach = (int)std::ceil((double)radius) + 1;
    hough_circles_tiled_kernel<<<dim3(cdiv(cols, 64), cdiv(rows, 32)), 1024, 0, s>>>(
        pts, npts, rows, cols, row0, dt->c, dt->s, (float)radius, reach, acc);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

int micv_hough_peaks_dev(micv_ctx *ctx, const int32_t *acc, int rows, int cols,
                         unsigned num_peaks, int threshold, uint32_t *peaks_rc, int64_t *count,
                         micv_stream stream) {
    MICV_REQUIRE(ctx && acc && count, "micv_hough_peaks: null argument");
    MICV_REQUIRE(peaks_rc || num_peaks == 0, "micv_hough_peaks: peaks_rc is null");
    MICV_REQUIRE(rows > 0 && cols > 0 && (int64_t)rows * cols < ((int64_t)1 << 31),
                 "micv_hough_peaks: bad size %dx%d", rows, cols);
    MICV_REQUIRE(num_peaks <= 4096, "micv_hough_peaks: num_peaks %u > 4096 not supported",
                 num_peaks);
    MICV_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)rows * cols;
    void *scratch;
    MICV_TRY(ctx->reserve(Carver::need(n, 4) + Carver::need(1, 8) +
                              Carver::need((size_t)num_peaks + 1, 8) + compact_scratch_bytes(n),
                          &scratch));
    Carver c(scratch);
    int32_t *cand = c.take<int32_t>(n);
    int64_t *ncand = c.take<int64_t>(1);
    unsigned long long *sel = c.take<unsigned long long>((size_t)num_peaks + 1);
    MICV_TRY(ordered_compact(ctx, s, PeakPred{acc, rows, cols, threshold}, IndexEmit{cand}, n, n, ncand, c.base + c.off));
    if (num_peaks <= 64) {
        peak_select_all_kernel<<<1, 1024, 0, s>>>(acc, cand, ncand, n, num_peaks, cols, peaks_rc, count);
        MICV_LAUNCH_CHECK();
        return MICV_OK;
    }
    MICV_HIP(hipMemsetAsync(sel, 0, ((size_t)num_peaks + 1) * 8, s));
    for (unsigned k = 0; k < num_peaks; k++) {
        peak_select_kernel<<<64, 256, 0, s>>>(acc, cand, ncand, n, sel, (int)k);
        MICV_LAUNCH_CHECK();
    }
    peak_emit_kernel<<<1, 64, 0, s>>>(sel, num_peaks, cols, peaks_rc, count);
    MICV_LAUNCH_CHECK();
    return MICV_OK;
}

}  // extern "C"
