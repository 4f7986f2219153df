"""chatterbox_amd -- MI355X-native (gfx950) inference path for the Chatterbox TTS family.

Python host code (this package) sequences hand-written HIP kernels in libcbx_hip.so (C ABI: include/cbx.h).
Importing `chatterbox_amd.ops` (or anything above it) requires the built library; there is no CPU fallback.
"""
__version__ = "0.1.0"
