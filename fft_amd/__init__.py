"""fft_amd -- MI355X-native implementation of the CMSIS-DSP FFT hot path of fy10000000/FFT.

The product is the C-ABI shared library ``fft_amd/lib/libarmfft_mi355.so`` (hand-written HIP for
gfx950, declared in ``include/arm_fft_mi355.h``).  This package only loads it through ctypes and
mirrors the reference's init/execute interface (same names, argument order and error behaviour:
Include/dsp/transform_functions.h of the reference) for tests and benchmarks.  There is no CPU
fallback: loading fails loudly when the library has not been built.
"""
from .capi import *  # noqa: F401,F403
from .capi import __all__  # noqa: F401
