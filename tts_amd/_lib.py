"""ctypes binding of libglvoc.so (the C ABI declared in include/glvoc.h).

The library is the only compute path of this package: if it cannot be loaded
the import of any operator fails loudly; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GLVOC_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("GLVOC_LIB") or os.path.join(_HERE, "csrc", "libglvoc.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); mirrors include/glvoc.h declaration by declaration
SIGNATURES = {
    "glv_version": (ctypes.c_int, []),
    "glv_plan_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int]),
    "glv_plan_destroy": (None, [ctypes.c_void_p]),
    "glv_plan_workspace_bytes": (ctypes.c_int64, [ctypes.c_void_p]),
    "glv_plan_reserve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_void_p]),
    "glv_plan_set_utterance_ids": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                                  ctypes.c_int, ctypes.c_void_p]),
    "glv_plan_set_inverse_basis": (ctypes.c_int, [ctypes.c_void_p, c_float_p]),
    "glv_mel_invert": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                      ctypes.c_void_p]),
    "glv_plan_set_mel_basis": (ctypes.c_int, [ctypes.c_void_p, c_float_p]),
    "glv_mel_invert_nnls": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                           ctypes.c_void_p, ctypes.c_void_p]),
    "glv_griffinlim": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_uint64, c_int_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_void_p, ctypes.c_void_p]),
    "glv_mel_to_audio": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_void_p, ctypes.c_uint64, c_int_p, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "glv_peak_normalize": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int_p, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]),
    "glv_peak_normalize_noalloc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                  ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]),
    "glv_peak_reduce": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int_p, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "glv_peak_scale": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int_p, ctypes.c_int,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]),
    "glv_shard_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "glv_shard_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]),
    "glv_shard_halo_samples": (ctypes.c_int, [ctypes.c_void_p]),
    "glv_shard_halo_pack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p]),
    "glv_shard_halo_unpack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p]),
    "glv_shard_halo_views": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_void_p)] * 4),
    "glv_shard_output_range": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                              ctypes.POINTER(ctypes.c_int64)]),
    "glv_shard_end": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "glv_halo_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "glv_halo_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int]),
    "glv_halo_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "glv_halo_attach": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "glv_shard_exchange": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "glv_shard_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]),
    "glv_melspectrogram": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, c_int_p, c_int_p,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "glv_power_to_db": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int_p, ctypes.c_int,
                                       ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "glv_stft": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, c_int_p, c_int_p,
                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "glv_istft": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int_p, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "glv_plan_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]),
    "glv_plan_last_loop_ms": (ctypes.c_int, [ctypes.c_void_p, c_float_p]),
    "glv_plan_status": (ctypes.c_int, [ctypes.c_void_p]),
    "glv_plan_persistent_capacity": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "glv_plan_last_loop_launches": (ctypes.c_int, [ctypes.c_void_p]),
    "glv_plan_last_loop_groups": (ctypes.c_int, [ctypes.c_void_p]),
    "glv_shard_last_run_form": (ctypes.c_int, [ctypes.c_void_p]),
    "glv_plan_drawn_phase": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "glv_last_error": (ctypes.c_char_p, []),
}


class GlvError(RuntimeError):
    """A libglvoc call returned a negative glv_status."""


_lib = None


def load():
    """Loads libglvoc.so once and attaches the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). "
            "tts_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().glv_last_error()
        raise GlvError(f"libglvoc status {rc}: {msg.decode() if msg else ''}")
