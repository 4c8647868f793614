"""ctypes binding of libspx.so (the C ABI declared in include/spx.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the operator layer raises.
torch is imported first on purpose: PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7);
loading libspx.so afterwards makes the dynamic linker bind our kernels to THAT runtime instance, so the
hipStream_t handles we get from torch.cuda.current_stream() are valid inside the library.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL — see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPX_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libspx.so")   # SPX_LIB_PATH: dev override (A/B of two builds)

SPX_ABI_VERSION = 2
SPX_MAX_KVOL = 32

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_sz = ctypes.c_size_t
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)

#: name -> (restype, argtypes).  Mirrors include/spx.h one to one (tests/test_abi.py checks the header).
SIGNATURES = {
    "spx_strerror": (ctypes.c_char_p, [_int]),
    "spx_abi_version": (_int, []),
    "spx_voxelize_ws_bytes": (_sz, [_i64, _int, _int]),
    "spx_read_status": (_int, [_vp, _vp]),
    "spx_voxelize": (_int, [_vp, _i64, _int, _int, _int, _int, _int, _int, _f32p, _f32p, _i32p, _int, _int, _vp, _vp,
                            _vp, _vp, _vp, _i64, _int, _vp, _vp, _sz, _vp]),
    "spx_dynamic_voxelize_ws_bytes": (_sz, [_i64, _int, _i32p, _i64]),
    "spx_dynamic_voxelize": (_int, [_vp, _i64, _int, _int, _int, _int, _f32p, _f32p, _i32p, _int, _vp, _vp, _vp, _vp, _i64,
                                    _vp, _sz, _vp]),
    "spx_voxel_query": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i32p, _int, ctypes.c_float, _i32p, _vp, _vp, _vp]),
    "spx_voxel_query_dilated": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i32p, _int, ctypes.c_float, ctypes.c_float, _i32p,
                                       _i32p, _vp, _vp, _vp, _vp]),
    "spx_mean_vfe": (_int, [_vp, _vp, _i64, _vp, _int, _int, _vp, _vp]),
    "spx_subm_rulebook_ws_bytes": (_sz, [_i64]),
    "spx_subm_rulebook": (_int, [_vp, _i64, _vp, _int, _i32p, _i32p, _i32p, _vp, _i64, _vp, _int, _vp, _vp, _sz, _vp]),
    "spx_conv_out_cap": (_i64, [_i64, _int, _i32p, _i32p, _i32p]),
    "spx_conv_rulebook_ws_bytes": (_sz, [_i64, _int, _i32p]),
    "spx_conv_rulebook": (_int, [_vp, _i64, _vp, _int, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _vp, _vp, _vp, _vp,
                                 _vp, _i64, _i32p, _i32p, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spx_pack_weight": (_int, [_vp, _int, _int, _int, _int, _vp, _vp]),
    "spx_pack_weight_batched": (_int, [_vp, _int, _i64, _vp]),
    "spx_conv_gemm": (_int, [_vp, _int, _vp, _int, _int, _int, _vp, _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp]),
    "spx_conv_plan_bytes": (_sz, [_i64]),
    "spx_conv_plan": (_int, [_vp, _i64, _int, _i64, _vp, _vp, _vp]),
    "spx_conv_gemm_balanced_ws_bytes": (_sz, [_int, _i64]),
    "spx_conv_group_ws_bytes": (_sz, [_i64]),
    "spx_conv_group": (_int, [_vp, _i64, _int, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spx_conv_gemm_balanced": (_int, [_vp, _int, _vp, _int, _int, _int, _vp, _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp, _vp,
                                      _vp, _sz, _vp]),
    "spx_conv_ring_plan_bytes": (_sz, [_i64]),
    "spx_conv_ring_stat_rows": (_int, []),
    "spx_conv_ring_tiles_per_wave": (_int, [_int]),
    "spx_conv_ring_plan": (_int, [_vp, _i64, _int, _i64, _vp, _vp, _vp]),
    "spx_conv_gemm_ring": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _vp, _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp, _vp,
                                  _vp, _vp, _vp]),
    "spx_conv_wgrad_ws_bytes": (_sz, [_int, _int, _int, _i64]),
    "spx_conv_wgrad_counts_bytes": (_sz, [_int, _i64]),
    "spx_conv_wgrad_counts": (_int, [_vp, _i64, _int, _i64, _vp, _vp, _vp]),
    "spx_conv_wgrad": (_int, [_vp, _int, _vp, _int, _int, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spx_densify": (_int, [_vp, _vp, _i64, _vp, _int, _int, _i32p, _int, _vp, _vp]),
    "spx_densify_bwd": (_int, [_vp, _vp, _i64, _vp, _int, _int, _i32p, _int, _vp, _vp]),
    "spx_boxes_iou_bev": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _vp]),
    "spx_nms_ws_bytes": (_sz, [_i64]),
    "spx_nms_bev": (_int, [_vp, _i64, ctypes.c_float, _int, _vp, _vp, _vp, _sz, _vp]),
    "spx_assign_targets_ws_bytes": (_sz, [_int, _int, _int]),
    "spx_bn_relu_ws_bytes": (_sz, [_int]),
    "spx_bn_relu_fwd": (_int, [_vp, _i64, _vp, _int, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _int, _vp, _vp,
                               _vp, _vp, _sz, _vp]),
    "spx_bn_relu_bwd": (_int, [_vp, _vp, _i64, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "spx_bn_add_relu_fwd": (_int, [_vp, _vp, _i64, _vp, _int, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _int,
                                   _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "spx_bn_add_relu_bwd": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _sz,
                                   _vp]),
    "spx_bn_apply": (_int, [_vp, _vp, _i64, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp, _i64, _vp]),
    "spx_wino_weight_floats": (_i64, [_int, _int]),
    "spx_wino_weight": (_int, [_vp, _i64, _i64, _i64, _i64, _int, _int, _int, _vp, _vp]),
    "spx_conv2d_wino": (_int, [_vp, _i64, _vp, _int, _int, _int, _int, _int, _vp, _vp, _int, _vp, _i64, _vp, _vp]),
    "spx_wino_stat_rows": (_i64, [_int, _int, _int]),
    "spx_bn_relu_fwd_from_sums": (_int, [_vp, _i64, _vp, _int, _vp, _i64, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float,
                                         _int, _vp, _i64, _vp, _vp, _vp]),
    "spx_wino_wgrad_ws_bytes": (_sz, [_int, _int]),
    "spx_conv2d_wino_wgrad": (_int, [_vp, _i64, _vp, _i64, _int, _int, _int, _int, _int, _vp, _i64, _i64, _i64, _i64, _vp, _sz,
                                     _vp]),
    "spx_anchor_loss_ws_bytes": (_sz, [_int, _i64]),
    "spx_anchor_loss": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _int, _int, ctypes.c_float, ctypes.c_float,
                               ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _sz,
                               _vp]),
    "spx_assign_targets": (_int, [_vp, _int, _i64, _int, _vp, _int, _int, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                  _vp]),
}

_lib = None


class SpxError(RuntimeError):
    pass


def load():
    """Load libspx.so and declare every prototype.  Raises (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpxError(
            "libspx.so not found at %s — build it with `make -C %s` or __graft_entry__.build(); "
            "there is no CPU / eager fallback for the sparse-conv hot path." % (LIB_PATH, os.path.dirname(LIB_PATH)))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    if lib.spx_abi_version() != SPX_ABI_VERSION:
        raise SpxError("libspx.so ABI version %d != binding version %d" % (lib.spx_abi_version(), SPX_ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise SpxError("%s failed: %s (code %d)" % (what, load().spx_strerror(rc).decode(), rc))


def i3(v):
    """host int32[3] argument"""
    a = (ctypes.c_int32 * 3)(*[int(x) for x in v])
    return a


def f_arr(v):
    a = (ctypes.c_float * len(v))(*[float(x) for x in v])
    return a
