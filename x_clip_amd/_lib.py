"""ctypes binding of libxclip_hip.so (C ABI: include/xclip.h).

The product has exactly one compute backend: the gfx950 HIP library built in-tree by `python -m x_clip_amd.build`.
If it is missing, or a tensor is not on a GPU, the ops raise -- there is no CPU / PyTorch fallback.

`_use_library_for_tests` exists only so the CPU test-suite can point the same Python code at
tests/emu/libxclip_emu.so (the identical kernel sources compiled for the host against a wave64 emulator).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

import torch  # noqa: F401  (must be imported first: the .so binds to torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxclip_hip.so")

_lib = None
_is_emulator = False

P, I, L, F, U = c_void_p, c_int, c_int64, c_float, c_uint64

_SIGNATURES = {
    "xclip_abi_version": (c_int, []),
    "xclip_last_error": (c_char_p, []),
    "xclip_build_info": (c_char_p, []),
    "xclip_clock_sample": (c_int, [P, L, P]),
    "xclip_layernorm_fwd": (c_int, [P, L, P, P, P, L, L, P, P, L, L, F, I, I, P]),
    "xclip_layernorm_bwd_workspace_bytes": (c_int64, [L, L]),
    "xclip_layernorm_bwd": (c_int, [P, P, L, P, P, P, P, P, L, P, P, L, L, L, I, I, P]),
    "xclip_l2norm_fwd": (c_int, [P, P, P, L, L, I, P]),
    "xclip_l2norm_bwd": (c_int, [P, P, P, P, L, L, I, P]),
    "xclip_text_embed_fwd": (c_int, [P, P, P, P, P, L, L, L, L, P, I, P]),
    "xclip_text_embed_bwd": (c_int, [P, P, P, P, P, L, L, L, L, I, I, P]),
    "xclip_patchify": (c_int, [P, P, P, L, L, L, L, L, L, L, I, P]),
    "xclip_token_mean_fwd": (c_int, [P, L, P, L, L, L, I, P]),
    "xclip_token_mean_bwd": (c_int, [P, P, L, P, L, L, L, I, P]),
    "xclip_copy_rows": (c_int, [P, L, P, L, L, L, I, P]),
    "xclip_add": (c_int, [P, P, P, L, I, P]),
    "xclip_rows_scatter_add_workspace_bytes": (c_int64, [L, L]),
    "xclip_rows_scatter_add": (c_int, [P, L, P, P, P, L, L, P, L, I, P]),
    "xclip_scatter_add_sorted": (c_int, [P, L, P, P, P, L, L, L, L, L, L, I, P]),
    "xclip_sort_ids_workspace_bytes": (L, [L]),
    "xclip_sort_ids": (c_int, [P, L, L, P, P, P, L, P]),
    "xclip_cast_from_f32": (c_int, [P, P, L, F, I, P]),
    "xclip_gemm_workspace_bytes": (c_int64, [L, L, L, I]),
    "xclip_gemm_small_limit": (c_int64, [L]),
    "xclip_gemm": (c_int, [I, I, P, L, P, L, P, L, L, L, L, F, P, P, L, P, P, L, P, L, I, P]),
    "xclip_ffn_dgrad_geglu_ok": (c_int, [L, L, L, I]),
    "xclip_ffn_dgrad_geglu_workspace_bytes": (c_int64, [L, L, L]),
    "xclip_ffn_dgrad_geglu": (c_int, [P, L, P, L, P, L, P, P, P, P, L, P, L, P, L, P, P, L, L, L, L, I, P]),
    "xclip_ffn_wgamma": (c_int, [P, L, P, P, L, L, I, P]),
    "xclip_layernorm_bwd_ffnstats": (c_int, [P, P, L, P, P, P, P, P, L, P, P, L, L, L, P, L, P, P, P, F, P, I, P]),
    "xclip_ffn_dgrad_geglu_rowc": (c_int, [P, L, P, L, P, L, P, P, P, L, P, P, L, L, L, L, I, P]),
    "xclip_gemm_batched": (c_int, [I, I, P, L, L, P, L, L, P, L, L, L, L, L, L, F, I, P]),
    "xclip_rowdot": (c_int, [P, L, P, L, P, L, L, I, P]),
    "xclip_attention_fwd": (c_int, [P, P, P, P, L, L, L, L, F, I, F, U, I, P]),
    "xclip_attention_bwd": (c_int, [P, P, P, P, P, P, P, L, L, L, L, F, I, F, U, I, P]),
    "xclip_attention_pool_fwd": (c_int, [P, P, P, P, P, L, L, L, L, F, L, I, P]),
    "xclip_attention_pool_bwd": (c_int, [P, P, P, P, P, P, P, P, L, L, L, L, F, L, I, P]),
    "xclip_dropout": (c_int, [P, P, L, F, U, I, P]),
    "xclip_filip_reduce": (c_int, [P, L, P, P, P, P, L, P, P, P, L, L, L, L, L, L, I, P]),
    "xclip_filip_fused_ok": (c_int, [L, L, L, I]),
    "xclip_filip_fused_workspace_bytes": (c_int64, [L, L, L, L]),
    "xclip_filip_fused_fwd": (c_int, [P, P, P, P, P, P, L, P, P, P, P, L, L, L, L, L, L, L, L, I, P]),
    "xclip_filip_route": (c_int, [P, L, P, P, P, P, L, P, P, P, L, L, L, L, L, L, I, P]),
    "xclip_rowlse": (c_int, [P, L, L, L, L, I, F, P, P, P]),
    "xclip_rowgrad": (c_int, [P, L, P, L, L, L, I, F, P, P, L, P, P]),
    "xclip_simreg_diff": (c_int, [P, L, P, L, P, L, L, L, L, P, I, P]),
    "xclip_rotary": (c_int, [P, L, L, L, L, L, L, P, I, I, P]),
    "xclip_layernorm_chain_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, L, L, F, I, P]),
    "xclip_layernorm_chain_bwd_workspace_bytes": (c_int64, [L, L]),
    "xclip_layernorm_chain_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, L, L, I, P]),
    "xclip_gather_rows": (c_int, [P, L, P, P, L, L, I, P]),
    "xclip_cross_entropy_fwd": (c_int, [P, L, P, L, L, P, P, I, P]),
    "xclip_cross_entropy_bwd": (c_int, [P, L, P, P, P, L, L, I, P]),
    "xclip_batchnorm_workspace_bytes": (c_int64, [L, L]),
    "xclip_batchnorm_fwd": (c_int, [P, P, P, P, P, P, P, P, F, F, I, I, L, L, P, L, I, P]),
    "xclip_batchnorm_bwd": (c_int, [P, P, P, P, P, P, P, P, P, I, I, L, L, P, L, I, P]),
    "xclip_neg_cosine_fwd": (c_int, [P, P, L, L, F, P, P, P, P, I, P]),
    "xclip_neg_cosine_bwd": (c_int, [P, P, P, P, P, P, F, P, L, L, I, P]),
    "xclip_dwconv4s2_workspace_bytes": (c_int64, [L, L, L, I]),
    "xclip_dwconv4s2_fwd": (c_int, [P, P, P, L, L, L, I, P]),
    "xclip_dwconv4s2_bwd": (c_int, [P, P, P, P, P, P, L, L, L, L, I, P]),
    "xclip_simloss_workspace_bytes": (c_int64, [L, L]),
    "xclip_simloss_partial": (c_int, [P, P, L, L, L, F, P, L, I, P, L, L, P, I, P]),
    "xclip_simloss_combine": (c_int, [P, L, L, P, P, P, F, P]),
    "xclip_simloss_fwd": (c_int, [P, P, L, L, L, F, P, L, I, F, P, P, P, P, I, P]),
    "xclip_simloss_grad": (c_int, [P, P, L, L, L, F, P, L, I, F, F, F, P, I, P, P, P, L, P, I, P]),
}
EXPORTS = tuple(_SIGNATURES)
ABI_VERSION = 23


def _bind(path: str):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.xclip_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.xclip_abi_version()} != {ABI_VERSION} -- rebuild (python -m x_clip_amd.build)")
    return lib


def lib():
    """The loaded HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"x_clip_amd: {LIB_PATH} is missing -- build the gfx950 kernel library first "
                f"(`python -m x_clip_amd.build`). There is no CPU/PyTorch fallback.")
        _lib = _bind(LIB_PATH)
    return _lib


def is_emulator() -> bool:
    return _is_emulator


def _use_library_for_tests(path):
    """TEST HOOK: load another build of the same C ABI (the host emulator) or reset with None."""
    global _lib, _is_emulator
    if path is None:
        _lib, _is_emulator = None, False
    else:
        _lib, _is_emulator = _bind(path), True


def use_measurement_build():
    """tools/ only: switch this process to libxclip_hip_measure.so (`python -m x_clip_amd.build --measure`), the build that carries the
    XCLIP_GEMM / XCLIP_*_ABL / ... environment switches of the A/B and ablation runs.  Never called by the product or the tests' parity
    cases: several of those switches return garbage by design."""
    global _lib, _is_emulator
    path = os.path.join(_HERE, "libxclip_hip_measure.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: python -m x_clip_amd.build --measure")
    _lib, _is_emulator = _bind(path), False


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed ({code}): {lib().xclip_last_error().decode()}")
