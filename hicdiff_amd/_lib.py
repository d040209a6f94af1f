"""ctypes binding of libhicdiff_hip.so (the C ABI declared in include/hicdiff_hip.h).

There is no fallback: if the shared library is missing or a symbol is absent, importing the
product path raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C hicdiff_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HICDIFF_HIP_LIB: another build of the same library (experiment builds, `make TAG=_x EXTRA=-D...` -> libhicdiff_hip_x.so); never a fallback
LIB_PATH = os.environ.get("HICDIFF_HIP_LIB") or os.path.join(_HERE, "libhicdiff_hip.so")

HD_OK, HD_EINVAL, HD_ENOWEIGHT, HD_EHIP, HD_ENOMEM, HD_ESTATE = 0, -1, -2, -3, -4, -5
HD_ARCH_UNET, HD_ARCH_HICEDRN = 0, 1
HD_T_INT64, HD_T_FLOAT32 = 0, 1
HD_PRECISION_F32, HD_PRECISION_BF16X3, HD_PRECISION_F16W2, HD_PRECISION_F16W1 = 0, 1, 2, 3
HD_ARITH_DEFAULT, HD_ARITH_F16W2, HD_ARITH_F16W1, HD_ARITH_F16W2_LOW = 0, 1, 2, 3      # hd_ddpm_coef.arith
HD_TRAIN_PREC_BF16 = 2
HD_PROFILE_MAX_ROWS = 96


class HdArchDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("dim", C.c_int32), ("n_mults", C.c_int32), ("mults", C.c_int32 * 8),
        ("channels", C.c_int32), ("self_condition", C.c_int32), ("sr3", C.c_int32), ("groups", C.c_int32),
        ("number_resnet", C.c_int32), ("reserved", C.c_int32 * 4),
    ]


class HdNamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


HD_ABI_VERSION = 3      # include/hicdiff_hip.h; checked against hd_abi_version() in load()


class _Prefixed(C.Structure):
    """Size-prefixed struct of the ABI: `struct_bytes` (first field) is filled in on construction."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_bytes = C.sizeof(type(self))


class HdDdpmCoef(_Prefixed):
    _fields_ = [
        ("struct_bytes", C.c_uint32),
        ("sqrt_recip_alphas_cumprod", C.c_float), ("sqrt_recipm1_alphas_cumprod", C.c_float),
        ("posterior_mean_coef1", C.c_float), ("posterior_mean_coef2", C.c_float),
        ("sigma", C.c_float), ("time_value", C.c_float), ("eps_coef", C.c_float), ("arith", C.c_uint32),
    ]


class HdDdrmCoef(_Prefixed):
    _fields_ = [
        ("struct_bytes", C.c_uint32),
        ("sqrt_at", C.c_float), ("sqrt_1m_at", C.c_float), ("sqrt_at_next", C.c_float), ("sigma_next", C.c_float),
        ("sigma_0", C.c_float), ("etaA", C.c_float), ("etaB", C.c_float), ("etaC", C.c_float), ("time_value", C.c_float),
        ("skip_network", C.c_uint32),
    ]


class HdProfileRow(C.Structure):
    _fields_ = [("kernel", C.c_char_p), ("launches", C.c_longlong), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


_P = C.c_void_p
# symbol -> (restype, argtypes); exactly the declarations of include/hicdiff_hip.h
SYMBOLS = {
    "hd_create": (C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(HdArchDesc)]),
    "hd_destroy": (None, [_P]),
    "hd_last_error": (C.c_char_p, [_P]),
    "hd_version": (C.c_char_p, []),
    "hd_abi_version": (C.c_int, []),
    "hd_load_weights": (C.c_int, [_P, C.POINTER(HdNamedTensor), C.c_int, _P]),
    "hd_reserve": (C.c_int, [_P, C.c_int, C.c_int]),
    "hd_workspace_bytes": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "hd_eps_forward": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "hd_ddpm_step": (C.c_int, [_P, _P, _P, _P, C.POINTER(HdDdpmCoef), _P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, _P]),
    "hd_ddrm_step": (C.c_int, [_P, _P, _P, _P, C.POINTER(HdDdrmCoef), _P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, _P]),
    "hd_q_sample": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "hd_loss_per_sample": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, C.c_int, _P]),
    "hd_profile_enable": (C.c_int, [C.c_int]),
    "hd_profile_read": (C.c_int, [C.POINTER(HdProfileRow), C.c_int]),
    "hd_set_precision": (C.c_int, [_P, C.c_int]),
    "hd_set_graphs": (C.c_int, [_P, C.c_int]),
    "hd_set_chains": (C.c_int, [_P, C.c_int]),
    "hd_chains_for": (C.c_int, [_P, C.c_int, C.c_int]),
    "hd_chain_begin": (C.c_int, [_P, _P]),
    "hd_chain_end": (C.c_int, [_P, _P]),
    "hd_randn": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, _P]),
    "hd_tile_metrics": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "hd_train_create": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int]),
    "hd_train_destroy": (None, [_P]),
    "hd_train_last_error": (C.c_char_p, [_P]),
    "hd_train_set_precision": (C.c_int, [_P, C.c_int]),
    "hd_train_set_objective": (C.c_int, [_P, C.c_int]),
    "hd_train_set_loss_weights": (C.c_int, [_P, _P]),
    "hd_train_param_count": (C.c_int, [_P, _P]),
    "hd_train_param_slot": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "hd_train_stage_count": (C.c_int, [_P]),
    "hd_train_slot_stage": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    "hd_train_stage_wait": (C.c_int, [_P, C.c_int, _P]),
    "hd_train_loss_backward": (C.c_int, [_P] * 6 + [C.c_int] + [_P] * 3 + [C.c_int, _P, _P]),
    "hd_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_longlong, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, _P]),
    "hd_ddrm_x0": (C.c_int, [_P, _P, C.c_float, C.c_float, _P, C.c_size_t, _P]),
    "hd_ddrm_general_update": (C.c_int, [_P] * 4 + [C.c_int] + [_P] * 3 + [C.POINTER(HdDdrmCoef), _P, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                          C.c_uint32, _P]),
    "hd_gather_cols": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "hd_kvec_matmul": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, _P]),
    "hd_sandwich_matmul": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "hd_dense_matmul": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "hd_fwht": (C.c_int, [_P, C.c_int, C.c_int, C.c_float, _P]),
    "hd_split_pieces": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "hd_stitch_pieces": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HiCDiff hot path has no CPU/PyTorch fallback. "
            "Build it with `make -C hicdiff_amd/csrc` (hipcc --offload-arch=gfx950).")
    # torch ships its own libamdhip64; load it FIRST so this library binds to the same HIP runtime
    # (streams and device pointers are shared with torch). Loading /opt/rocm's copy before torch
    # leaves two runtimes in the process and hipSetDevice fails.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype, fn.argtypes = res, args
    if lib.hd_abi_version() != HD_ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} implements ABI revision {lib.hd_abi_version()}, this binding is written for {HD_ABI_VERSION}: rebuild "
                           "the library (make -C hicdiff_amd/csrc)")
    _lib = lib
    return lib


class HdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hicdiff_hip error {code}: {msg}")
        self.code = code
