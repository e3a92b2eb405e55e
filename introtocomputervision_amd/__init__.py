"""MI355X-native per-pixel CV kernel path of tanmaniac/IntroToComputerVision.

The product is ``libmicv.so`` (hand-written HIP for gfx950 behind the C ABI declared in
``include/mi_cv.h``).  This package is the thin Python host side used by the tests and
``bench.py``: a ctypes binding plus modules named after the reference's namespaces
(``lk``, ``pyr``, ``harris``, ``sift``, ``stereo``, ``hough``).  There is no CPU fallback:
importing ``_capi`` fails loudly when the library has not been built.
"""
__version__ = "0.1.0"
