"""ctypes binding of libmentflow_hip.so (C ABI declared in include/mentflow_hip.h).

The product has exactly one compute path: the gfx950 library built by ``__graft_entry__.build()`` into
``mentflow_amd/csrc/libmentflow_hip.so``.  If it is missing, or a tensor is not resident on the GPU, the ops
raise — there is no CPU fallback.  (``use_library`` exists so the test-suite can point the same Python layer at
the host-emulated kernel build under tests/emu; nothing in the package calls it.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "csrc", "libmentflow_hip.so")

_lib: Optional[C.CDLL] = None
_device_type = "cuda"

_i64, _i32, _f32, _ptr = C.c_int64, C.c_int, C.c_float, C.c_void_p

# name -> (restype, argtypes); mirrors include/mentflow_hip.h line by line
PROTOTYPES = {
    "mf_abi_version": (_i32, []),
    "mf_last_error": (C.c_char_p, []),
    "mf_is_emulation": (_i32, []),
    "mf_prof_enable": (_i32, [_i32]),
    "mf_prof_report": (_i32, [_i32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mf_gather_f32": (_i32, [_ptr, _ptr, _ptr, _i64, _i32, _ptr]),
    "mf_flow_image_floats": (_i64, [_i32, _i32]),
    "mf_flow_rqs_deriv_slot": (_i32, [_i32]),
    "mf_flow_set_bwd_variant": (_i32, [_i32]),
    "mf_flow_bwd_scratch_floats": (_i64, [_i64, _i32, _i32, _ptr]),
    "mf_flow_rqs_layer_fwd": (_i32, [_ptr, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i32, _ptr]),
    "mf_flow_rqs_act_level": (_i32, [_i32, _i32, _i32, _ptr]),
    "mf_flow_rqs_act_floats": (_i64, [_i64, _i32, _i32, _i32, _i32]),
    "mf_flow_rqs_layer_fwd_save": (_i32, [_ptr, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i32, _ptr, _i64, _i32,
                                          _ptr]),
    "mf_flow_rqs_layer_bwd_saved": (_i32, [_ptr, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr,
                                           _i64, _i32, _ptr]),
    "mf_flow_bwd_slab_rows": (_i32, [_i64, _i32, _i32, _ptr]),
    "mf_flow_rqs_layer_bwd": (_i32, [_ptr, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr, _i64,
                                     _ptr]),
    "mf_flow_grad_reduce": (_i32, [_ptr, _i32, _i32, _i64, _ptr, _ptr, _i64, _ptr]),
    "mf_flow_rqs_layer_inv": (_i32, [_ptr, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr]),
    "mf_flow_affine_layer_inv": (_i32, [_ptr, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr]),
    "mf_flow_affine_image_floats": (_i64, [_i32, _i32]),
    "mf_flow_affine_bwd_scratch_floats": (_i64, [_i64, _i32]),
    "mf_flow_affine_layer_fwd": (_i32, [_ptr, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i32, _ptr]),
    "mf_flow_affine_bwd_slab_rows": (_i32, [_i64]),
    "mf_flow_affine_layer_bwd": (_i32, [_ptr, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr, _i64,
                                        _ptr]),
    "mf_flow_wide_limits": (_i32, [_ptr, _ptr, _ptr]),
    "mf_flow_wide_image_floats": (_i64, [_i32, _i32]),
    "mf_flow_wide_grad_floats": (_i64, [_i32, _i32]),
    "mf_flow_wide_layer_fwd": (_i32, [_ptr, _i32, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i32, _ptr]),
    "mf_flow_wide_bwd_scratch_floats": (_i64, [_i64, _i32, _i32, _i32]),
    "mf_flow_wide_bwd_slab_rows": (_i32, [_i64]),
    "mf_flow_wide_layer_bwd": (_i32, [_ptr, _i32, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr,
                                      _i64, _ptr]),
    "mf_flow_wide_layer_inv": (_i32, [_ptr, _i32, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr]),
    "mf_flow_wide_act_floats": (_i64, [_i64, _i32, _i32, _i32]),
    "mf_flow_wide_layer_fwd_save": (_i32, [_ptr, _i32, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i32, _ptr, _i64, _ptr]),
    "mf_flow_wide_layer_bwd_saved": (_i32, [_ptr, _i32, _i32, _i32, _i32, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _ptr,
                                            _i64, _ptr, _i64, _ptr]),
    "mf_proj_kde_ws_bytes": (_i64, [_i32, _i32]),
    "mf_proj_kde1d_fwd": (_i32, [_ptr, _i64, _i32, _ptr, _i32, _ptr, _i32, _f32, _i32, _ptr, _ptr, _ptr]),
    "mf_proj_kde1d_bwd": (_i32, [_ptr, _i64, _i32, _ptr, _i32, _ptr, _i32, _f32, _i32, _ptr, _ptr, _i32, _ptr]),
    "mf_proj_kde2d_fwd": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _i32, _ptr, _i32, _f32, _i32, _ptr, _i32, _f32, _i32,
                                 _ptr, _ptr, _ptr]),
    "mf_proj_kde2d_bwd": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _i32, _ptr, _i32, _f32, _i32, _ptr, _i32, _f32, _i32,
                                 _ptr, _ptr, _i32, _ptr]),
    "mf_multipole_kick_fwd": (_i32, [_ptr, _i64, _i32, _i32, _f32, _i32, _ptr, _ptr]),
    "mf_multipole_kick_bwd": (_i32, [_ptr, _i64, _i32, _i32, _f32, _i32, _ptr, _ptr, _ptr]),
    "mf_proj_hist1d_counts": (_i32, [_ptr, _i64, _i32, _ptr, _i32, _ptr, _i32, _ptr, _ptr]),
    "mf_proj_hist2d_counts": (_i32, [_ptr, _i64, _i32, _ptr, _ptr, _i32, _ptr, _i32, _ptr, _i32, _ptr, _ptr]),
    "mf_hist_norm_discrepancy_fwd": (_i32, [_ptr, _i32, _i32, _i32, _f32, _f32, _f32, _ptr, _i32, _f32, _f32, _ptr, _ptr,
                                            _ptr]),
    "mf_hist_norm_discrepancy_bwd": (_i32, [_ptr, _i32, _i32, _i32, _f32, _f32, _f32, _ptr, _i32, _f32, _f32, _ptr, _ptr,
                                            _ptr, _ptr]),
    "mf_mc_entropy_sums": (_i32, [_ptr, _ptr, _i64, _i32, _ptr, _ptr, _ptr]),
    "mf_scale_rows": (_i32, [_ptr, _i64, _i32, _ptr, _f32, _ptr, _i32, _ptr]),
}


class LibraryError(RuntimeError):
    pass


def _bind(lib: C.CDLL) -> C.CDLL:
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:  # pragma: no cover
            raise LibraryError(f"{lib._name} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    return lib


def use_library(path: str) -> None:
    """Load a specific build of the C-ABI library (tests only)."""
    global _lib, _device_type
    lib = _bind(C.CDLL(path))
    if lib.mf_abi_version() != 5:
        raise LibraryError(f"ABI version mismatch in {path}")
    _lib = lib
    _device_type = "cpu" if lib.mf_is_emulation() else "cuda"


def get_lib() -> C.CDLL:
    if _lib is None:
        if not os.path.exists(DEFAULT_PATH):
            raise LibraryError(
                f"{DEFAULT_PATH} not found: build the gfx950 kernels first "
                "(python -c 'import __graft_entry__ as g; g.build()'). mentflow_amd has no CPU fallback."
            )
        use_library(DEFAULT_PATH)
    return _lib


def set_flow_bwd_variant(variant=None) -> None:
    """None: default (MENTFLOW_BWD_FUSED read once, unset = fused); False: two-kernel backward; True: fused backward."""
    call("mf_flow_set_bwd_variant", -1 if variant is None else int(bool(variant)))


def device_type() -> str:
    get_lib()
    return _device_type


def call(name: str, *args) -> None:
    lib = get_lib()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {lib.mf_last_error().decode()}")


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous tensor resident where the loaded library computes (the GPU)."""
    if t is None:
        return None
    if t.device.type != device_type():
        raise RuntimeError(
            f"mentflow_amd kernels run on the GPU only: got a tensor on '{t.device}' "
            f"(library computes on '{device_type()}'); there is no CPU fallback"
        )
    if not t.is_contiguous():
        raise RuntimeError("mentflow_amd kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def stream_ptr(t: torch.Tensor):
    if t.device.type == "cuda":
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


PROF_KERNELS = {"flow_layer_fwd": 0, "flow_layer_bwd": 1, "outer_accum": 2, "kde1d_fwd": 3, "kde1d_bwd": 4,
                "kde2d_fwd": 5, "kde2d_bwd": 6}


def prof_enable(on: bool) -> None:
    get_lib().mf_prof_enable(int(on))


def prof_report() -> dict:
    """{kernel: (total_ms, launches)} of the launches recorded since prof_enable(True)."""
    lib = get_lib()
    out = {}
    for name, kid in PROF_KERNELS.items():
        ms, cnt = C.c_double(0.0), C.c_int64(0)
        if lib.mf_prof_report(kid, C.byref(ms), C.byref(cnt)) != 0:
            raise RuntimeError(lib.mf_last_error().decode())
        out[name] = (ms.value, cnt.value)
    return out
