"""ctypes binding of libmicv.so (the C ABI in include/mi_cv.h).

Fails loudly (ImportError) when the HIP library is missing: there is no CPU fallback in the
product path.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``bash introtocomputervision_amd/csrc/build.sh``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MICV_LIB (python side only) points the binding at another build of the same ABI, e.g. the
# -DMICV_DIAG flavour tools/phase_pmc.sh makes; the library itself reads no environment variables.
LIB_PATH = os.environ.get("MICV_LIB") or os.path.join(_HERE, "libmicv.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP extension has not been built "
        "(run `bash introtocomputervision_amd/csrc/build.sh`). There is no CPU fallback."
    )

lib = C.CDLL(LIB_PATH)

c_float_p = C.POINTER(C.c_float)
vp = C.c_void_p
sz = C.c_size_t
i32 = C.c_int
i64 = C.c_int64
u32 = C.c_uint
f32 = C.c_float
f64 = C.c_double

OK, EINVAL, EHIP, ENOMEM, EUNSUPPORTED = 0, -1, -2, -3, -4
STEREO_COLS_2R, STEREO_MIN_SSD_5E6, STEREO_SERIAL, STEREO_ROLLING = 1, 2, 4, 8
# micv_ctx_set_option (include/mi_cv.h): none of these changes a result
(OPT_LK_STREAM_GROUPS, OPT_LK_FORCE_GENERIC, OPT_LK_NARROW_TILES, OPT_SOBEL_GENERIC, OPT_HARRIS_GENERIC,
 OPT_NMS_SCAN, OPT_STEREO_ROWS, OPT_LK_CHAIN, OPT_LK_SHORT_TILES, OPT_LK_STREAM, OPT_LK_TALL_TILES,
 OPT_COMPACT_3PASS, OPT_LK_DIRECT_LEVELS, OPT_LK_BUILD_OVERLAP, OPT_LK_SPLIT, OPT_LK_STRIP,
 OPT_STEREO_EXACT) = range(1, 18)


MICV_COMM_ID_BYTES = 128  # mi_cv.h


class MicvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"micv error {code}: {msg}")
        self.code = code


# name -> (restype, argtypes).  Every symbol include/mi_cv.h declares appears here; the
# CPU test-suite checks the two lists against each other.
SIGNATURES = {
    "micv_version": (C.c_char_p, []),
    "micv_last_error": (C.c_char_p, []),
    "micv_ctx_create": (i32, [i32, C.POINTER(vp)]),
    "micv_ctx_destroy": (None, [vp]),
    "micv_ctx_scratch_bytes": (sz, [vp]),
    "micv_device_malloc": (i32, [vp, sz, C.POINTER(vp)]),
    "micv_device_free": (None, [vp, vp]),
    "micv_memcpy2d_h2d": (i32, [vp, vp, sz, vp, sz, sz, i32]),
    "micv_memcpy2d_d2h": (i32, [vp, vp, sz, vp, sz, sz, i32]),
    "micv_ctx_set_option": (i32, [vp, i32, i32]),
    "micv_ctx_get_option": (i32, [vp, i32, C.POINTER(i32)]),
    "micv_profile_enable": (i32, [vp, i32]),
    "micv_profile_reset": (i32, [vp]),
    "micv_profile_lk_level": (i32, [vp, i32, C.POINTER(f64), C.POINTER(i64)]),
    "micv_profile_lk_pairs": (i32, [vp, C.POINTER(i32)]),
    "micv_profile_lk_phases": (i32, [vp, i32, vp]),
    "micv_set_kernel_log": (i32, [vp, vp]),
    "micv_warmup": (i32, [vp, vp]),
    "micv_div_round_up": (sz, [sz, sz]),
    "micv_timer_create": (i32, [C.POINTER(vp)]),
    "micv_timer_start": (i32, [vp, vp]),
    "micv_timer_stop": (i32, [vp, vp]),
    "micv_timer_elapsed_ms": (i32, [vp, C.POINTER(f32)]),
    "micv_timer_destroy": (None, [vp]),
    # ps5
    "micv_lk_flow_pyr_dev": (i32, [vp, vp, vp, i32, i32, sz, i32, i32, vp, vp, sz, vp]),
    "micv_lk_flow_pyr_host": (i32, [vp, vp, vp, i32, i32, sz, i32, i32, vp, vp, sz]),
    "micv_lk_flow_pyr_batch_dev": (i32, [vp, vp, vp, i32, sz, i32, i32, sz, i32, i32, vp, vp, sz, sz, vp]),
    "micv_flow_bound_check_dev": (i32, [vp, vp, i32, sz, i32, i32, sz, i32, i32, f32, vp, vp]),
    "micv_lk_schedule_host": (i32, [i32, i32, i32, i32, i32, vp, i64, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]),
    "micv_lk_level_dev": (i32, [vp, vp, vp, i32, i32, sz, i32, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "micv_lk_level_batch_dev": (i32, [vp, vp, vp, i32, sz, i32, i32, sz, i32, vp, vp, i32, i32, sz, i32, i32, vp, vp, sz, sz, vp]),
    "micv_gaussian_pyramid_batch_dev": (i32, [vp, vp, i32, sz, i32, i32, sz, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), vp]),
    "micv_lk_flow_dev": (i32, [vp, vp, vp, i32, i32, sz, i32, vp, vp, sz, vp]),
    "micv_lk_flow_host": (i32, [vp, vp, vp, i32, i32, sz, i32, vp, vp, sz]),
    "micv_lk_warp_dev": (i32, [vp, vp, sz, vp, vp, sz, i32, i32, vp, sz, vp]),
    "micv_lk_warp_host": (i32, [vp, vp, sz, vp, vp, sz, i32, i32, vp, sz]),
    "micv_pyr_down_dev": (i32, [vp, vp, i32, i32, sz, vp, sz, vp]),
    "micv_pyr_down_host": (i32, [vp, vp, i32, i32, sz, vp, sz]),
    "micv_pyr_up_dev": (i32, [vp, vp, i32, i32, sz, vp, sz, vp]),
    "micv_pyr_up_host": (i32, [vp, vp, i32, i32, sz, vp, sz]),
    "micv_gaussian_pyramid_dev": (i32, [vp, vp, i32, i32, sz, i32, C.POINTER(vp), vp]),
    "micv_gaussian_pyramid_host": (i32, [vp, vp, i32, i32, sz, i32, C.POINTER(vp)]),
    "micv_laplacian_pyramid_dev": (i32, [vp, vp, i32, i32, sz, i32, C.POINTER(vp), vp]),
    "micv_rgb8_to_gray_f32_dev": (i32, [vp, vp, i32, i32, sz, vp, sz, vp]),
    "micv_to_gray_f32_dev": (i32, [vp, vp, i32, i32, sz, i32, i32, vp, sz, vp]),
    "micv_to_gray_f32_host": (i32, [vp, vp, i32, i32, sz, i32, i32, vp, sz]),
    "micv_lk_flow_pyr_frames_host": (i32, [vp, vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, vp, sz]),
    "micv_lk_flow_seq_host": (i32, [vp, C.POINTER(vp), i32, i32, i32, sz, i32, i32, i32, i32, C.POINTER(vp), C.POINTER(vp), sz]),
    "micv_resize_linear_dev": (i32, [vp, vp, i32, i32, sz, vp, i32, i32, sz, vp]),
    # ps4
    "micv_sobel_dev": (i32, [vp, vp, i32, i32, sz, i32, f32, vp, vp, sz, vp]),
    "micv_sobel_host": (i32, [vp, vp, i32, i32, sz, i32, f32, vp, vp, sz]),
    "micv_harris_response_dev": (i32, [vp, vp, vp, i32, i32, sz, i32, f64, f32, vp, sz, vp]),
    "micv_harris_response_host": (i32, [vp, vp, vp, i32, i32, sz, i32, f64, f32, vp, sz]),
    "micv_comm_unique_id": (i32, [vp]),
    "micv_comm_create": (i32, [vp, vp, vp, i32, i32, C.POINTER(vp)]),
    "micv_comm_destroy": (i32, [vp]),
    "micv_comm_rank": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
    "micv_comm_selftest": (i32, [vp, vp, vp]),
    "micv_lk_level_kernel_name": (i32, [vp, i32, i32, i32, i32, C.c_char_p, sz]),
    "micv_rowshard_band": (i32, [i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "micv_lk_flow_pyr_rowshard_dev": (i32, [vp, vp, vp, vp, i32, sz, i32, i32, sz, i32, i32, vp, vp, sz, sz, vp]),
    "micv_lk_flow_pyr_rowshard_virtual_dev": (i32, [vp, i32, vp, vp, i32, sz, i32, i32, sz, i32, i32, vp, vp, sz, sz, i32, vp]),
    "micv_lk_flow_pyr_rowshard_host": (i32, [vp, vp, vp, vp, i32, i32, sz, i32, i32, vp, vp, sz]),
    "micv_hough_lines_rowshard_dev": (i32, [vp, vp, vp, i32, i32, sz, i32, i32, C.c_uint, C.c_uint, vp, vp]),
    "micv_allreduce_sum_i32_dev": (i32, [vp, vp, vp, sz, vp]),
    "micv_harris_response_ex_dev": (i32, [vp, vp, vp, i32, i32, sz, i32, f64, f32, i32, vp, sz, vp]),
    "micv_harris_response_ex_host": (i32, [vp, vp, vp, i32, i32, sz, i32, f64, f32, i32, vp, sz]),
    "micv_harris_refine_dev": (i32, [vp, vp, i32, i32, sz, f64, i32, vp, sz, vp, i64, vp, vp]),
    "micv_harris_refine_host": (i32, [vp, vp, i32, i32, sz, f64, i32, vp, sz, vp, i64, vp]),
    "micv_harris_corners_dev": (i32, [vp, vp, i32, i32, sz, i32, i32, f64, f32, i32, f64, i32, vp, vp, sz, vp, sz, vp, sz, vp, i64, vp, vp]),
    "micv_harris_corners_host": (i32, [vp, vp, i32, i32, sz, i32, i32, f64, f32, i32, f64, i32, vp, vp, sz, vp, sz, vp, sz, vp, i64, vp]),
    "micv_sift_angles_dev": (i32, [vp, vp, vp, i32, i32, sz, vp, sz, vp]),
    "micv_sift_angles_host": (i32, [vp, vp, vp, i32, i32, sz, vp, sz]),
    "micv_sift_keypoints_dev": (i32, [vp, vp, vp, i32, i32, sz, vp, i64, f32, vp, vp]),
    "micv_sift_keypoints_host": (i32, [vp, vp, vp, i32, i32, sz, vp, i64, f32, vp]),
    "micv_sift_descriptors_dev": (i32, [vp, vp, vp, i32, i32, sz, vp, i64, vp, sz, vp]),
    "micv_sift_descriptors_host": (i32, [vp, vp, vp, i32, i32, sz, vp, i64, vp, sz]),
    # ps2
    "micv_disparity_ssd_dev": (i32, [vp, vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, sz, vp]),
    "micv_disparity_ssd_host": (i32, [vp, vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, sz]),
    "micv_disparity_ncorr_dev": (i32, [vp, vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, sz, vp]),
    "micv_disparity_ncorr_host": (i32, [vp, vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, sz]),
    # ps1
    "micv_hough_lines_dims": (i32, [i32, i32, u32, u32, C.POINTER(i32), C.POINTER(i32)]),
    "micv_hough_lines_dev": (i32, [vp, vp, i32, i32, sz, u32, u32, vp, vp]),
    "micv_hough_lines_host": (i32, [vp, vp, i32, i32, sz, u32, u32, vp]),
    "micv_hough_circles_dev": (i32, [vp, vp, i32, i32, sz, u32, vp, vp]),
    "micv_hough_circles_host": (i32, [vp, vp, i32, i32, sz, u32, vp]),
    "micv_hough_lines_band_dev": (i32, [vp, vp, i32, i32, sz, i32, i32, u32, u32, vp, vp]),
    "micv_hough_circles_band_dev": (i32, [vp, vp, i32, i32, sz, i32, i32, u32, vp, vp]),
    "micv_hough_peaks_dev": (i32, [vp, vp, i32, i32, u32, i32, vp, vp, vp]),
    "micv_hough_peaks_host": (i32, [vp, vp, i32, i32, u32, i32, vp, vp]),
    # ps1 edge front-end
    "micv_generate_edge_dev": (i32, [vp, vp, i32, i32, sz, i32, f64, f64, f64, vp, sz, vp]),
    # ps4 matching
    "micv_bf_knn2_dev": (i32, [vp, vp, i32, sz, vp, i32, sz, i32, vp, vp, vp]),
    "micv_bf_ratio_filter_dev": (i32, [vp, vp, vp, i32, f64, vp, vp, i64, vp, vp]),
    # ps7
    "micv_mhi_frame_difference_dev": (i32, [vp, vp, vp, i32, i32, sz, f64, i32, i32, f64, vp, sz, vp]),
    "micv_mhi_energy_dev": (i32, [vp, vp, i32, i32, sz, vp, sz, vp]),
    "micv_mhi_energy_host": (i32, [vp, vp, i32, i32, sz, vp, sz]),
    "micv_mhi_threshold_dev": (i32, [vp, vp, i32, i32, sz, f64, vp, sz, vp]),
    "micv_mhi_update_dev": (i32, [vp, vp, sz, vp, sz, i32, i32, i32, vp]),
    # host-pointer flavours of the "next" rows
    "micv_generate_edge_host": (i32, [vp, vp, i32, i32, sz, i32, f64, f64, f64, vp, sz]),
    "micv_bf_knn2_host": (i32, [vp, vp, i32, sz, vp, i32, sz, i32, vp, vp]),
    "micv_bf_ratio_filter_host": (i32, [vp, vp, vp, i32, f64, vp, vp, i64, vp]),
    "micv_mhi_frame_difference_host": (i32, [vp, vp, vp, i32, i32, sz, f64, i32, i32, f64, vp, sz]),
    "micv_mhi_threshold_host": (i32, [vp, vp, i32, i32, sz, f64, vp, sz]),
    "micv_mhi_update_host": (i32, [vp, vp, sz, vp, sz, i32, i32, i32]),
}

MISSING = []
for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        MISSING.append(_name)
        continue
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return lib.micv_last_error().decode()


def check(rc):
    if rc != OK:
        raise MicvError(rc, last_error())


class Context:
    """Owns a micv_ctx (device ordinal + scratch arena). One per host thread / stream."""

    def __init__(self, device=0):
        h = vp()
        check(lib.micv_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("micv Context used after close()")
        return self._h

    def scratch_bytes(self):
        return int(lib.micv_ctx_scratch_bytes(self._h))

    def set_option(self, option, value):
        check(lib.micv_ctx_set_option(self._h, int(option), int(value)))

    def get_option(self, option):
        v = i32()
        check(lib.micv_ctx_get_option(self._h, int(option), C.byref(v)))
        return v.value

    def set_lk_groups(self, n):
        """Stream groups a batch of pairs is split into (0 = library default)."""
        self.set_option(OPT_LK_STREAM_GROUPS, n)

    def lk_level_kernel_name(self, win=15, rows=1080, cols=1920, batch=8):
        """The level-kernel instantiation launch_lk_level_fused picks for a rows x cols level of `batch` pairs with
        a doubling coarse flow (mode 1) under this context's options -- asked of the dispatch itself
        (micv_lk_level_kernel_name: same code path as a launch, nothing launched)."""
        buf = C.create_string_buffer(160)
        check(lib.micv_lk_level_kernel_name(self._h, int(win), int(rows), int(cols), int(batch), buf, 160))
        return buf.value.decode()

    def profile(self, on=True):
        check(lib.micv_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        check(lib.micv_profile_reset(self._h))

    def profile_lk_level(self, level):
        """(total_ms, launches) of pyramid level `level` since the last reset."""
        ms, n = f64(), i64()
        check(lib.micv_profile_lk_level(self._h, int(level), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_lk_pairs(self):
        n = i32()
        check(lib.micv_profile_lk_pairs(self._h, C.byref(n)))
        return n.value

    def profile_lk_phases(self, enable):
        """Read + reset the 16 in-kernel phase counters, then enable/disable stamping."""
        buf = (C.c_uint64 * 16)()
        check(lib.micv_profile_lk_phases(self._h, 1 if enable else 0, buf))
        return list(buf)

    def warmup(self, stream=None):
        check(lib.micv_warmup(self._h, stream))

    def close(self):
        if self._h:
            lib.micv_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


KERNEL_LOG_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_float, vp)
_kernel_log_keepalive = []


def set_kernel_log(fn):
    """fn(kernel_name: str, ms: float) for every timed `_host` call, or None to remove the sink."""
    if fn is None:
        check(lib.micv_set_kernel_log(None, None))
        return
    cb = KERNEL_LOG_FN(lambda name, ms, user: fn(name.decode(), float(ms)))
    _kernel_log_keepalive.append(cb)  # the library keeps the pointer: the thunk must outlive it
    check(lib.micv_set_kernel_log(C.cast(cb, vp), None))


class Timer:
    """GpuTimer (common/include/common/GpuTimer.h): hipEvent pair on a stream."""

    def __init__(self):
        h = vp()
        check(lib.micv_timer_create(C.byref(h)))
        self._h = h

    def start(self, stream=None):
        check(lib.micv_timer_start(self._h, stream))

    def stop(self, stream=None):
        check(lib.micv_timer_stop(self._h, stream))

    def elapsed_ms(self):
        ms = f32()
        check(lib.micv_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self._h:
                lib.micv_timer_destroy(self._h)
                self._h = None
        except Exception:
            pass
