// stereo_exact.hpp -- the exact-sum stereo path for 8-bit-valued images (stereo_exact.hip), as stereo.hip sees it.
#pragma once
#include "common.hpp"
#include "stereo_float.hpp"

namespace micv {

struct StereoExactArgs {
    const float *left, *right;  // the caller's f32 images (read by the pack pre-pass only)
    int stride, rows, cols, min_d, max_d, wcols;
    uint32_t *lplan;  // [strip][column + R][8]: packed rows of `left`, one 32-byte record per column (scalar loads)
    int lcols;        // records per strip: columns -R .. lcols - R - 1, clamped copies outside the image
    int qlo, qhi;     // columns the pre-pass visits
    uint32_t *rpack;  // [strip][word][colsP]: packed rows of `right`
    int colsP;
    int32_t *A;  // [rows][cols]  window energy of left at output x (MIN_SSD_5E6 only, else null)
    int32_t *B;  // [rows][nB]    window energy of right at position p = x + d, p - min_d in [0, nB)
    int nB;
    unsigned *flag;  // holds `epoch` once a pixel of this call was found not to be an integer in 0..255
    unsigned epoch;
    int8_t *disp;
    int dstride;
    int X, nxs, ntiles;  // output columns per wave, strips per row of strips, waves
    int min_ssd_5e6;
    int nblk_exact, fgx;  // workgroups of exact-sum tiles in the launch; the float tiles behind them: fgx strips of columns per row of tiles
};

// Whether the exact path has a kernel for this call at all (radius, flags); the images decide on the device.
bool stereo_exact_covers(int rad, int flags, bool ncc);
size_t stereo_exact_scratch(int rows, int cols, int rad, int min_d, int max_d, int wcols, int flags, int wave_slots3);
// Enqueues pre-pass + search (wave_slots3: waves the device holds at three per SIMD).  The search launch carries the
// float kernel's tiles for the same call (`f`, 8 or 10 rows per wave) as trailing workgroups: nothing else to launch.  `scratch` holds stereo_exact_scratch() bytes.
int stereo_exact_launch(hipStream_t s, void *scratch, const float *left, const float *right, int rows, int cols,
                        int stride, int rad, int min_d, int max_d, int flags, int wcols, int8_t *disp, int dstride,
                        unsigned *flag, unsigned epoch, int wave_slots3, const StereoArgs &f, bool rows10);

}  // namespace micv
