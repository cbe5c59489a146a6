"""ctypes binding of libmovenet_hip.so (the C ABI in include/movenet_hip.h).

There is deliberately NO fallback: if the library is missing or fails to load,
every entry point raises ``NativeLibraryError`` -- the product never routes
through PyTorch ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

# MOVENET_HIP_LIB selects another build of the SAME library (the diagnostic twin with
# in-kernel stamps); it is not a fallback mechanism.
LIB_PATH = os.environ.get("MOVENET_HIP_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "lib", "libmovenet_hip.so")

MVN_OK = 0
MVN_ERR_BAD_DIMS = -1
MVN_ERR_BAD_ARG = -2
MVN_ERR_TOO_SHORT = -3
MVN_ERR_LAUNCH = -4
MVN_ERR_UNSUPPORTED = -5

GEN_AUTO, GEN_GENERIC, GEN_STREAM, GEN_PIPE, GEN_PIPE_F16, GEN_FOLD = 0, 1, 2, 3, 4, 5
BWD_FORM_GENERIC, BWD_FORM_HALVES, BWD_FORM_ONE = 1, 2, 3  # mvn_last_backward_form (include/movenet_hip.h)
PIPE_VARIANTS = (GEN_PIPE, GEN_PIPE_F16, GEN_FOLD)  # variants with a hand-off status word


class NativeLibraryError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [
        ("layer_size", C.c_int32),
        ("stack_size", C.c_int32),
        ("input_channels", C.c_int32),
        ("residual_channels", C.c_int32),
        ("skip_channels", C.c_int32),
    ]


_PP = C.POINTER(C.c_void_p)


class Params(C.Structure):
    _fields_ = [
        ("causal_w", C.c_void_p),
        ("filter_w", _PP), ("gate_w", _PP),
        ("residual_w", _PP), ("residual_b", _PP),
        ("skip_w", _PP), ("skip_b", _PP),
        ("ctx_filter_w", _PP), ("ctx_filter_b", _PP),
        ("ctx_gate_w", _PP), ("ctx_gate_b", _PP),
        ("head1_w", C.c_void_p), ("head1_b", C.c_void_p),
        ("head2_w", C.c_void_p), ("head2_b", C.c_void_p),
    ]


_FP = C.POINTER(C.c_void_p)


class FwdBuffers(C.Structure):
    _fields_ = [("acts", C.c_void_p), ("th", C.c_void_p), ("sg", C.c_void_p), ("z", C.c_void_p),
                ("skip", C.c_void_p), ("a1", C.c_void_p), ("ctx", C.c_void_p), ("ctx_ld", C.c_int32),
                ("dense_audio", C.c_void_p), ("dense_ld", C.c_int32)]


class ParamGrads(C.Structure):
    _fields_ = [
        ("causal_w", C.c_void_p),
        ("filter_w", _PP), ("gate_w", _PP),
        ("residual_w", _PP), ("residual_b", _PP),
        ("skip_w", _PP), ("skip_b", _PP),
        ("head1_w", C.c_void_p), ("head1_b", C.c_void_p),
        ("head2_w", C.c_void_p), ("head2_b", C.c_void_p),
        ("ctx_filter_w", _PP), ("ctx_filter_b", _PP),
        ("ctx_gate_w", _PP), ("ctx_gate_b", _PP),
    ]


class BwdBuffers(C.Structure):
    _fields_ = [("dx_a", C.c_void_p), ("dx_b", C.c_void_p), ("dfg", C.c_void_p),
                ("dskip", C.c_void_p), ("da1", C.c_void_p), ("dlogit", C.c_void_p),
                ("dctx", C.c_void_p)]


class VideoParams(C.Structure):  # also used for mvn_video_grads (same layout)
    _fields_ = [("conv_w", C.c_void_p), ("conv_b", C.c_void_p),
                ("up_w", C.c_void_p * 3), ("up_b", C.c_void_p * 3)]


# name -> (restype, argtypes); tests/test_capi.py checks the header against this
SIGNATURES = {
    "mvn_abi_version": (C.c_int, []),
    "mvn_reload_switches": (C.c_int, []),
    "mvn_last_error": (C.c_char_p, []),
    "mvn_receptive_fields": (C.c_int, [C.POINTER(Dims)]),
    "mvn_output_size": (C.c_int, [C.POINTER(Dims), C.c_int]),
    "mvn_gen_variant": (C.c_int, [C.POINTER(Dims), C.c_int, C.c_int]),
    "mvn_gen_launch_pipelines": (C.c_int, [C.POINTER(Dims), C.c_int, C.c_int]),
    "mvn_gen_launch_is_cooperative": (C.c_int, []),
    "mvn_gen_weights_floats": (C.c_size_t, [C.POINTER(Dims), C.c_int]),
    "mvn_gen_state_floats": (C.c_size_t, [C.POINTER(Dims), C.c_int]),
    "mvn_gen_status_offset": (C.c_size_t, [C.POINTER(Dims), C.c_int]),
    "mvn_gen_pack_weights": (C.c_int, [C.POINTER(Dims), C.c_int, C.POINTER(Params), C.c_void_p,
                                       C.c_void_p]),
    "mvn_generate": (C.c_int, [C.POINTER(Dims), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                               C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mvn_transpose_context": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "mvn_padded_len": (C.c_int, [C.c_int]),
    "mvn_forward": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), C.c_void_p, C.c_int, C.c_int,
                              C.c_int, C.POINTER(FwdBuffers), C.c_void_p, C.c_int, C.c_int,
                              C.c_int, C.c_void_p]),
    "mvn_forward_f16": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(FwdBuffers), C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p]),
    "mvn_backward": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(ParamGrads),
                               C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(FwdBuffers),
                               C.POINTER(BwdBuffers), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                               C.c_void_p]),
    "mvn_last_backward_form": (C.c_int, []),
    "mvn_upsample_video": (C.c_int, [C.POINTER(Dims), C.POINTER(VideoParams), C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_void_p]),
    "mvn_upsample_video_scratch_floats": (C.c_size_t, [C.POINTER(Dims), C.c_int, C.c_int]),
    "mvn_upsample_video_backward": (C.c_int, [C.POINTER(Dims), C.POINTER(VideoParams),
                                              C.POINTER(VideoParams), C.c_void_p, C.c_int, C.c_int,
                                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mvn_gen_prime_from_forward": (C.c_int, [C.POINTER(Dims), C.POINTER(FwdBuffers), C.c_int,
                                             C.c_int, C.c_void_p, C.c_void_p]),
    "mvn_ce_parts": (C.c_int, [C.c_int, C.c_int]),
    "mvn_ce_on_probs_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "mvn_ce_on_probs_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvn_softmax_ce_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "mvn_softmax_ce_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]),
    "mvn_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                 C.POINTER(C.c_size_t), C.c_int, C.c_void_p]),
    "mvn_mu_law_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "mvn_mu_law_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "mvn_onehot_to_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "mvn_index_to_onehot": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "mvn_publish_words": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the library; raises NativeLibraryError loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m movenet_amd.csrc.build` "
            "(or __graft_entry__.build()).  movenet_amd has no CPU/PyTorch fallback."
        )
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the box
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if handle.mvn_abi_version() != 2:
        raise NativeLibraryError("libmovenet_hip.so ABI version mismatch")
    _lib = handle
    return handle


def last_error() -> str:
    return lib().mvn_last_error().decode(errors="replace")


def check(rc: int, what: str) -> int:
    """Map a negative status to the exception type the reference would raise."""
    if rc >= 0:
        return rc
    msg = f"{what}: {last_error()} (status {rc})"
    if rc == MVN_ERR_TOO_SHORT:
        raise ValueError(last_error())  # movenet/wavenet.py:141-146
    if rc in (MVN_ERR_BAD_DIMS, MVN_ERR_BAD_ARG, MVN_ERR_UNSUPPORTED):
        raise ValueError(msg)
    raise RuntimeError(msg)


def make_dims(layer_size: int, stack_size: int, input_channels: int, residual_channels: int,
              skip_channels: int) -> Dims:
    return Dims(layer_size, stack_size, input_channels, residual_channels, skip_channels)


def ptr_array(ptrs: Sequence[int]):
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return arr, C.cast(arr, _PP)
