"""ctypes binding of libtactilesr_hip.so (the C ABI declared in include/tactilesr_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, the
product raises.  Nothing here imports from ``oracle/``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_void_p, c_float, c_longlong, c_double

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSR_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libtactilesr_hip.so")   # override: kernel A/B experiments
ABI_VERSION = 23

_P, _I, _F, _L = c_void_p, c_int, c_float, c_longlong

# name -> argtypes; must list every symbol include/tactilesr_hip.h declares
SIGNATURES = {
    "tsr_abi_version": [],
    "tsr_build_flags": [],
    "tsr_pack_conv_weight": [_P, _P, _I, _I, _I, _P],
    "tsr_conv2d_fwd": [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsr_conv_weight_bf16s_elems": [_I, _I, _I, _I],
    "tsr_pack_conv_weight_bf16s": [_P, _P, _I, _I, _I, _I, _P],
    "tsr_conv2d_fwd_bf16s": [_P, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsr_stem_fwd": [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "tsr_pack_conv_weight_f16s": [_P, _P, _I, _I, _I, _F, _P],
    "tsr_pack_conv_weight_f16s_dev": [_P, _P, _I, _I, _I, _P, _P],
    "tsr_pack_conv_weight_dgrad_f16s_dev": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "tsr_conv2d_fwd_f16s": [_P, _I, _I, _I, _P, _I, _I, _F, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsr_conv2d_fwd_f16s_fuse1x1": [_P, _I, _I, _I, _P, _I, _F, _P, _P, _P, _P, _I, _P, _F, _P, _P, _I, _I, _P, _I, _I,
                                    _I, _I, _I, _I, _P],
    "tsr_pair_channel_perm": [_P],
    "tsr_conv_weight_pair_elems": [_I],
    "tsr_pack_conv_weight_pair_f16s": [_P, _P, _P, _I, _F, _P, _P],
    "tsr_conv2d_fwd_f16s_pair": [_P, _I, _I, _I, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsr_head_fwd": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _P],
    "tsr_conv2d_fwd_b16": [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsr_conv_weight_b16k_elems": [_I, _I, _I],
    "tsr_pack_conv_weight_b16k": [_P, _P, _I, _I, _I, _P],
    "tsr_pack_w2_b16k": [_P, _P, _P],
    "tsr_conv2d_ex_dgrad_b16k": [_I, _I, _I],
    "tsr_pack_conv_weight_dgrad_b16k": [_P, _P, _I, _I, _I, _I, _I, _P],
    "tsr_conv_weight_b16k_pair_elems": [_I],
    "tsr_pack_conv_weight_b16k_pair": [_P, _P, _I, _P],
    "tsr_conv2d_fwd_b16k": [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsr_conv2d_fwd_b16k_fuse1x1": [_P, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsr_conv2d_fwd_b16k_pair": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsr_stem_fwd_b16": [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "tsr_head_fwd_b16": [_P, _I, _I, _P, _P, _I, _I, _I, _I, _P],
    "tsr_conv2d_ex": [_P, _P],
    "tsr_conv2d_slab_entries": [_I, _I, _I],
    "tsr_conv2d_slab_entries_ex": [_I, _I, _I, _I, _I, _I],
    "tsr_pack_conv_weight_dgrad": [_P, _P, _I, _I, _I, _I, _I, _P],
    "tsr_pack_conv_weight_dgrad_bf16s": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "tsr_pack_conv_weight_dgrad_f16s": [_P, _P, _I, _I, _I, _I, _I, _F, _P],
    "tsr_conv2d_wgrad": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P],
    "tsr_conv2d_wgrad_bf16s": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "tsr_conv2d_wgrad_splits": [_I, _I, _I, _I, _I, _I, _I],
    "tsr_conv2d_wgrad_wgs_per_split": [_I, _I, _I, _I],
    "tsr_reduce_splits": [_P, _P, _L, _I, _F, _P],
    "tsr_bn_stats_finalize": [_P, _P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P],
    "tsr_cb16_stats_entries": [_I, _I],
    "tsr_cb16_stats": [_P, _I, _I, _I, _I, _P, _P, _P],
    "tsr_bn_bwd_finalize": [_P, _I, _I, ctypes.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "tsr_bn_bwd_apply": [_P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P],
    "tsr_stem_wgrad": [_P, _I, _I, _I, _I, _I, _P, _I, _I, _P, _I, _I, _P],
    "tsr_head_bwd": [_P, _P, _P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P, _P],
    "tsr_cb16_stats_b16": [_P, _I, _I, _I, _I, _P, _P, _P],
    "tsr_bn_bwd_apply_b16": [_P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P],
    "tsr_bn_relu_b16": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P],
    "tsr_conv2d_wgrad_b16k": [_I, _I, _I],
    "tsr_conv2d_ex_fwd1x1_b16k": [_I, _I],
    "tsr_stem_wgrad_b16": [_P, _I, _I, _I, _I, _I, _P, _I, _I, _P, _I, _I, _P],
    "tsr_head_bwd_b16": [_P, _P, _P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "tsr_target_prep": [_P, _P, _F, _I, _I, _I, _I, _I, _P],
    "tsr_mse_fwd_bwd": [_P, _P, _P, _P, _L, _F, _P, _P],
    "tsr_adam_l2_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P],
    "tsr_adam_l2_multi": [_P, _I, _F, c_double, c_double, _F, _F, _I, _P],
    "tsr_psnr_ssim": [_P, _P, _I, _I, c_double, c_double, c_double, c_double, _P, _P, _P],
    "tpsf_forward": [_P, _P, _P, _P, _P, _I, _P],
    "tpsf_backward": [_P, _P, _P, _P, _P, _P, _I, _P],
    "tsr_sgemm": [_P, _L, _L, _P, _L, _L, _P, _P, _I, _I, _I, _I, _P],
    "tsr_sgemm_splitk": [_P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _I, _P],
    "tsr_sgemm_masked": [_P, _L, _L, _P, _L, _L, _P, _P, _I, _I, _I, _P],
    "tsr_sgemm_splitk_strided": [_P, _L, _L, _P, _L, _L, _P, _L, _I, _I, _I, _I, _P],
    "tsr_colsum_splitk": [_P, _P, _L, _I, _I, _I, _P],
    "tsr_act_bwd": [_P, _P, _L, _I, _P],
    "tsr_nchw_to_cb16": [_P, _P, _I, _I, _I, _I, _I, _P],
    "tsr_cb16_to_nchw": [_P, _P, _I, _I, _I, _I, _I, _P],
}

_lib = None

# Bumped by every kernel that rewrites parameters behind autograd's back (the fused optimizer step): modules key
# their packed-weight caches on it, so no torch-private version-counter API is needed.
_param_epoch = [0]


def param_epoch() -> int:
    return _param_epoch[0]


def bump_param_epoch() -> None:
    _param_epoch[0] += 1


class TactileSRHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TactileSRHipError(
            f"{LIB_PATH} not found: build it with `python -m tactilesr_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = c_longlong if name.endswith("_elems") else c_int
    v = lib.tsr_abi_version()
    if v != ABI_VERSION:
        raise TactileSRHipError(f"libtactilesr_hip.so ABI {v} != expected {ABI_VERSION}: rebuild")
    flags = lib.tsr_build_flags()
    if flags != 0:
        # an experimental variant (tools/build_variant.py): never loaded by accident, never silent
        if os.environ.get("TSR_ALLOW_VARIANT") != "1":
            raise TactileSRHipError(f"{LIB_PATH} is an experimental variant build (tsr_build_flags() = {flags}); set "
                                    "TSR_ALLOW_VARIANT=1 to load it for a kernel A/B measurement")
        import warnings
        warnings.warn(f"tactilesr_amd: loaded an experimental variant library ({LIB_PATH}, build flags {flags}); "
                      "its results are NOT parity-checked")
    _lib = lib
    return lib


def build_flags() -> int:
    """0 = the shipped library, non-zero = an experimental variant (see tsr_build_flags in include/tactilesr_hip.h)."""
    return int(load().tsr_build_flags())


def ptr(t) -> c_void_p:
    """Device pointer of a tensor (None -> NULL).  Tensors must be CUDA fp32/int contiguous."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise TactileSRHipError("tactilesr_amd kernels need CUDA (ROCm) tensors; got a CPU tensor "
                                "(there is no CPU fallback)")
    if not t.is_contiguous():
        raise TactileSRHipError("non-contiguous tensor passed to a HIP kernel")
    return c_void_p(t.data_ptr())


def stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


_ERR = {1: "bad argument", 2: "kernel launch failure"}


def call(name: str, *args) -> None:
    st = getattr(load(), name)(*args)
    if st != 0:
        raise TactileSRHipError(f"{name} failed: status {st} ({_ERR.get(st, 'unknown')})")
