"""cer-mvs_amd: MI355X-native (gfx950) depth-inference hot path of CER-MVS.

Public names mirror the reference's (core/raft.py, core/corr.py, core/update.py,
alt_cuda_corr) so an ``inference.py``-style driver switches by changing its imports.
Importing the package never touches the GPU; the HIP library ``csrc/libcermvs.so`` is
loaded on first use and its absence is a hard error (there is no CPU fallback).
"""
__version__ = "0.1.0"

_LAZY = {
    "RAFT": ("raft", "RAFT"),
    "CorrBlock": ("corr", "CorrBlock"),
    "UpdateBlock": ("update", "UpdateBlock"),
    "ConvGRU": ("update", "ConvGRU"),
    "BasicEncoder": ("extractor", "BasicEncoder"),
    "alt_cuda_corr": ("alt_cuda_corr", None),
    "inference": ("inference", "inference"),
    "DepthMapPipeline": ("pipeline", "DepthMapPipeline"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        m = importlib.import_module("cer_mvs_amd." + mod)
        return m if attr is None else getattr(m, attr)
    raise AttributeError(name)
