"""karios_amd -- MI355X-native image-matching hot path for KARIOS.

`karios_amd.matcher` mirrors `karios.matcher` (KLT, klt_tracker, ZNCCService,
LargeOffsetMatcher); `karios_amd.ops` exposes the individual GPU operators;
`karios_amd.parallel` shards tiles over the GPUs of a node.  All numeric work runs in
`libkarios_hip.so` (hand-written HIP for gfx950) -- there is no CPU fallback.
"""
__version__ = "0.2.0"


def pinned_empty(shape, dtype, ctx=None):
    """numpy array over page-locked host memory (see `karios_amd._lib.pinned_empty`): rasters read into such arrays upload
    asynchronously and at full PCIe rate."""
    from ._lib import pinned_empty as _pe
    return _pe(shape, dtype, ctx)
