"""hosnerf_amd -- MI355X-native (gfx950) implementation of HOSNeRF's per-ray hot path.

Host code is Python over a C-ABI HIP library (`hosnerf_amd/lib/libhosrender.so`, built by
`__graft_entry__.build()`), loaded with ctypes.  There is no CPU fallback: every op raises
`hosnerf_amd._lib.HosLibraryError` if the library or a GPU is missing.
"""
__version__ = "0.1.0"
