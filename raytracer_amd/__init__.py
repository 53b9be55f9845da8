"""MI355X-native ray-intersection and photon-gather engine (host-side Python mirror).

The compute lives in librt_engine.so (hand-written HIP for gfx950, C ABI in
include/rt_engine.h); `engine` binds it, `workloads` generates synthetic inputs.
"""
