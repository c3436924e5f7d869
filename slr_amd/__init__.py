"""slr_amd — MI355X-native unidirectional path tracer behind libSLR's renderer interface.

The product is the C-ABI shared library slr_amd/csrc/libslrhip.so (include/slrhip.h) plus the
C++ `Renderer`-shaped adapter in slr_amd/csrc/host/.  This Python package is the harness:
ctypes binding, synthetic scene builders, and the multi-GPU driver used by bench.py.
"""
from . import abi, scenes  # noqa: F401
from .binding import Context, HipLibraryMissing, load_library  # noqa: F401
