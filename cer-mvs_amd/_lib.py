"""ctypes binding of csrc/libcermvs.so (the C ABI in include/cer_mvs.h).

There is deliberately NO fallback: if the shared library is missing, cannot be loaded, or
reports no HIP device when an op is called, a RuntimeError is raised."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CER_MVS_LIB") or os.path.join(_HERE, "csrc", "libcermvs.so")
ABI_VERSION = 1070
CONV_MAX_SRC = 4
EPI_LINEAR, EPI_RELU, EPI_GATES, EPI_GRU, EPI_DELTA = 0, 1, 2, 3, 4
EPI_OUT_SPLIT, EPI_AUX_SPLIT = 0x100, 0x200       # cer_mvs.h: split32 activation layout flags, or-ed into `epi`
EPI_CORR_FP6 = 0x800                               # cer_conv3x3_s16: ... on its FP6 (e2m3, block-scaled) form: half the matrix-pipe passes again (round 6)
EPI_CORR_FP8 = 0x400                               # cer_conv3x3_s16: correction terms of the tensor sources on the fp8 matrix instruction
# cer_mvs.h, "s16" convs: log2 scales of the split16 activation classes (|x| <= 1: hidden state, r*h; ReLU outputs; generated
# disparity features)
S16_UNIT, S16_RELU, S16_DISP = 14, 4, 6
S16_FRAG16, S16_ACC32, S16_F32X8 = 0, 1, 2        # cer_s16_layout_f32 layouts

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_long
_F = _c.c_float
_D = _c.c_double


class ConvInputs(ctypes.Structure):
    _fields_ = [("src", _P * CONV_MAX_SRC), ("ch", _I * CONV_MAX_SRC), ("kind", _I * CONV_MAX_SRC), ("nsrc", _I)]


COPY_MAX_SEG = 4


class CopySegments(ctypes.Structure):
    _fields_ = [("src", _P * COPY_MAX_SEG), ("dst", _P * COPY_MAX_SEG), ("n", _L * COPY_MAX_SEG)]


_SIGNATURES = {
    "cer_abi_version": (_I, []),
    "cer_error_string": (_c.c_char_p, [_I]),
    "cer_device_count": (_I, []),
    "cer_alt_corr_forward_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cer_alt_corr_backward_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cer_alt_corr_bwd_tuples_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cer_alt_corr_bwd_reduce_f32": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "cer_cost_build_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _D, _I, _I, _I, _I, _F, _P]),
    "cer_cost_build_algo": (_I, [_I]),
    "cer_overflow_flag": (_I, [_P]),
    "cer_f16_scan_overflow": (_I, [_P, _L, _P, _I, _P]),
    "cer_feat_split_f16": (_I, [_P, _P, _L, _L, _I, _P, _P]),
    "cer_cost_lines_workspace": (_L, [_I, _I, _I, _I]),
    "cer_cost_lines_views_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _D, _I, _I, _I, _P]),
    "cer_cost_lines_reduce_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _D, _I, _I, _I, _F, _P]),
    "cer_cost_lines_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _D, _I, _I, _I, _I, _F, _I, _P]),
    "cer_pyramid_f32": (_I, [_P, _L, _I, _I, _I, _F, _P]),
    "cer_corr_lookup_f32": (_I, [_P, _P, _P, _L, _P, _I, _L, _I, _I, _D, _I, _I, _I, _P]),
    "cer_corr_encode_f32": (_I, [_P, _P, _P, _P, _I, _I, _L, _I, _P]),
    "cer_lookup_encode_f32": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _D, _I, _I, _I, _I, _I, _I, _P, _I, _F, _I, _P]),
    "cer_conv3x3_packed_size": (_L, [_I, _I]),
    "cer_conv3x3_pack_f32": (_I, [_P, _P, _I, _I, _c.POINTER(_I), _c.POINTER(_I), _I]),
    "cer_conv3x3_f32": (_I, [_c.POINTER(ConvInputs), _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cer_conv3x3_f16x3_packed_size": (_L, [_I, _I]),
    "cer_conv3x3_f16x3_pack": (_I, [_P, _P, _I, _I, _c.POINTER(_I), _c.POINTER(_I), _I]),
    "cer_conv3x3_f16x3_collapsed_size": (_L, [_I, _c.POINTER(_I), _c.POINTER(_I), _I]),
    "cer_conv3x3_f16x3_pack_collapsed": (_I, [_P, _P, _I, _I, _c.POINTER(_I), _c.POINTER(_I), _I]),
    "cer_conv3x3_f16x3": (_I, [_c.POINTER(ConvInputs), _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cer_delta_proj_packed_size": (_L, [_I]),
    "cer_delta_proj_pack": (_I, [_P, _P, _I]),
    "cer_delta_sum_f32": (_I, [_P, _I, _F, _P, _P, _P, _I, _I, _P]),
    "cer_delta_tail_f32": (_I, [_P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "cer_enc_stem_tiles": (_I, [_I, _I]),
    "cer_enc_stem_s16_packed_size": (_L, []),
    "cer_enc_stem_s16_tiles": (_I, [_I, _I]),
    "cer_enc_stem_s16_pack": (_I, [_P, _P, _P]),
    "cer_enc_stem_s16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cer_enc_stem_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cer_enc_conv_packed_size": (_L, [_I, _I, _I]),
    "cer_enc_conv_pack": (_I, [_P, _P, _I, _I, _I]),
    "cer_enc_conv_pack_f6": (_I, [_P, _P, _I, _I, _I]),
    "cer_enc_conv_tiles": (_I, [_I, _I, _I, _I, _I]),
    "cer_enc_conv_f16x3": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "cer_enc_stats_reduce_f32": (_I, [_P, _P, _I, _I, _I, _L, _F, _P]),
    "cer_enc_merge_f32": (_I, [_P, _P, _P, _P, _P, _I, _L, _I, _I, _P]),
    "cer_enc_pc_supported": (_I, [_I, _I, _I, _I, _I]),
    "cer_enc_pc_tiles": (_I, [_I, _I, _I, _I, _I]),
    "cer_enc_pc_conv": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "cer_plane_stats_f32": (_I, [_P, _P, _L, _L, _F, _P]),
    "cer_norm_act_f32": (_I, [_P, _P, _P, _P, _P, _L, _L, _I, _P]),
    "cer_nchw_to_nhwc_f32": (_I, [_P, _P, _I, _L, _F, _P]),
    "cer_nhwc_to_nchw_f32": (_I, [_P, _P, _I, _L, _F, _P]),
    "cer_nchw_to_nhwc_border_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "cer_copy_segments_f32": (_I, [_c.POINTER(CopySegments), _P]),
    "cer_split32_f32": (_I, [_P, _P, _L, _I, _I, _P]),
    "cer_conv3x3_s16_packed_size": (_L, [_I, _c.POINTER(_I), _c.POINTER(_I), _I, _I]),
    "cer_conv3x3_s16_scale": (_I, [_P, _I, _I, _c.POINTER(_I), _c.POINTER(_I), _c.POINTER(_I), _I]),
    "cer_conv3x3_s16_pack": (_I, [_P, _P, _I, _I, _c.POINTER(_I), _c.POINTER(_I), _c.POINTER(_I), _I, _I, _I]),
    "cer_conv3x3_s16_edge_size": (_L, [_I]),
    "cer_conv3x3_s16_edge_pack": (_I, [_P, _P, _I, _I, _c.POINTER(_I), _c.POINTER(_I), _c.POINTER(_I), _I, _I]),
    "cer_conv3x3_s16": (_I, [_c.POINTER(ConvInputs), _c.POINTER(_I), _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cer_delta_proj_s16_packed_size": (_L, [_I]),
    "cer_delta_proj_s16_pack": (_I, [_P, _P, _I, _c.POINTER(_I)]),
    "cer_s16_padded_pixels": (_L, [_I, _I]),
    "cer_s16_layout_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cer_s16_rows_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cer_multires_merge_f32": (_I, [_P, _I, _I, _P, _I, _I, _D, _P, _P]),
    "cer_resize_linear_f32": (_I, [_P, _I, _I, _P, _I, _I, _P]),
    "cer_geo_consistency_f32": (_I, [_P, _P, _P, _I, _I, _I, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
}

# include/cer_mvs_variants.h: exported by csrc/variants/libcermvs_optin.so only (round 4's opt-in kernel forms; CER_MVS_LIB selects the library)
_VARIANT_SIGNATURES = {
    "cer_cost_lines_form": (_I, [_I]),
    "cer_cost_lines_stats": (_I, [_P, _I]),
    "cer_conv3x3_s16_pc": (_I, [_I]),
}

_lib = None
_recorder = None


class LaunchPlan:
    """The raw C-ABI calls of one pass over fixed buffers (e.g. one GRU iteration), recorded while they execute and
    replayable without the Python-side argument marshalling (~40 us -> ~3 us of host time per launch).  Valid while
    every tensor the recorded pointers refer to is alive and in place - the caller keeps them (``keep``)."""

    def __init__(self, keep=()):
        self.calls = []
        self.keep = list(keep)

    def replay(self):
        for name, fn, args in self.calls:
            rc = fn(*args)
            if rc != 0:
                check(rc, name)


class _Recording:
    def __init__(self, lib, plan):
        self._lib, self._plan = lib, plan

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        calls = self._plan.calls

        def call(*args):
            calls.append((name, fn, args))
            return fn(*args)
        return call


class recording:
    """``with recording(plan): ...`` - every library call made through load() inside the block is appended to plan."""

    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        global _recorder
        if isinstance(_recorder, _Recording):
            raise RuntimeError("cer-mvs_amd: nested launch recording")
        self._outer = _recorder                    # (a ``timing`` context stays active underneath: its wrappers get recorded)
        _recorder = _Recording(load(), self.plan)
        return self.plan

    def __exit__(self, *exc):
        global _recorder
        _recorder = self._outer
        return False


class _Timing:
    """Wraps the library so that every stream-taking call (the last argument of every launching entry point) is bracketed by
    HIP events recorded on that stream - bench.py's instrumented pass.  ``records``: list of (name, args, e0, e1)."""

    def __init__(self, lib, records):
        self._lib, self._records = lib, records

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        sig = _SIGNATURES.get(name)
        if sig is None or not sig[1] or sig[1][-1] is not _P or name.endswith("_pack"):
            return fn
        records = self._records

        def call(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            records.append((name, args, e0, e1))
            return rc
        return call


class timing:
    """``with timing(records): ...`` - every launching library call inside the block is timed with HIP events."""

    def __init__(self, records):
        self.records = records

    def __enter__(self):
        global _recorder
        if _recorder is not None:
            raise RuntimeError("cer-mvs_amd: timing inside a launch recording")
        _recorder = _Timing(load(), self.records)
        return self.records

    def __exit__(self, *exc):
        global _recorder
        _recorder = None
        return False


def exported_symbols():
    """Names every build of the library must export (mirrors include/cer_mvs.h)."""
    return sorted(_SIGNATURES)


def load():
    """Load (once) and return the ctypes library; never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _recorder if _recorder is not None else _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"cer-mvs_amd: HIP library not found at {LIB_PATH}. Build it with `make -C cer-mvs_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host's ROCm install
        raise RuntimeError(f"cer-mvs_amd: cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"cer-mvs_amd: {LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _VARIANT_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if lib.cer_abi_version() != ABI_VERSION:
        raise RuntimeError(f"cer-mvs_amd: ABI mismatch (library {lib.cer_abi_version()}, python {ABI_VERSION}); rebuild")
    _lib = lib
    return lib


def has_variant_forms():
    """True when the loaded library is the variant build that carries round 4's opt-in kernel forms (include/cer_mvs_variants.h)."""
    lib = load()
    return all(hasattr(lib, n) for n in _VARIANT_SIGNATURES)


def check(rc, what):
    if rc != 0:
        msg = load().cer_error_string(rc).decode()
        raise RuntimeError(f"cer-mvs_amd: {what} failed with code {rc}: {msg}")


def dev_ptr(t, name, dtype=torch.float32):
    """Device pointer of a dense CUDA tensor - the reference's CHECK_INPUT (correlation.cpp:19-21)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}")
    return ctypes.c_void_p(t.data_ptr())


def cur_stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
