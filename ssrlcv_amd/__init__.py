"""ssrlcv_amd -- MI355X-native (gfx950) SIFT -> match -> triangulate hot path of uga-ssrl/SSRLCV.

The product is the C-ABI library `libssrlcv_hip.so` (include/ssrlcv_hip.h) plus the C++ host mirror of the
reference API in ssrlcv_amd/host/.  This Python package is plumbing for tests and bench.py: it loads the
library with ctypes and passes raw device pointers of torch tensors.  It never falls back to a CPU path: if
the HIP library is missing, importing `ssrlcv_amd.capi` raises.
"""
__version__ = "0.1.0"
