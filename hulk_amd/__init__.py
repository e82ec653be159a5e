"""hulk_amd — MI355X-native implementation of HULK's `sketch` hot path.

Only what the path needs lives here:
  csrc/        HIP kernels + the C ABI (libhulkhip.so, declared in include/hulk_hip.h)
  _lib.py      ctypes binding (fails loudly when the library or a gfx950 GPU is missing)
  sketcher.py  host mirror of the reference's boss / HistoSketch interface
  sketchio.py  the reference's JSON sketch container (byte-compatible writer, loader)
  synth.py     counter-based synthetic read generator used by bench.py and the tests
"""
from ._lib import HulkError, LIB_PATH  # noqa: F401
from .sketcher import GpuSketcher, HistoSketch, spectrum_size  # noqa: F401

__all__ = ["GpuSketcher", "HistoSketch", "HulkError", "spectrum_size", "LIB_PATH"]
