"""karios_amd -- MI355X-native image-matching hot path for KARIOS.

`karios_amd.matcher` mirrors `karios.matcher` (KLT, klt_tracker, ZNCCService,
LargeOffsetMatcher); `karios_amd.ops` exposes the individual GPU operators;
`karios_amd.parallel` shards tiles over the GPUs of a node.  All numeric work runs in
`libkarios_hip.so` (hand-written HIP for gfx950) -- there is no CPU fallback.
"""
__version__ = "0.1.0"
