"""ctypes binding of libptts_hip.so (include/ptts.h). The product path has NO CPU fallback: if the HIP
library is missing or a call fails, this module raises (`NativeLibraryError` / `ValueError`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

PTTS_F32, PTTS_BF16 = 0, 1
PTTS_OK, PTTS_E_INVALID, PTTS_E_HIP, PTTS_E_MISSING, PTTS_E_CAPACITY, PTTS_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptts_hip.so")


class NativeLibraryError(RuntimeError):
    pass


class PttsConfig(C.Structure):
    _fields_ = [
        ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("ffn_dim", C.c_int32),
        ("num_codebooks", C.c_int32), ("vocab_size", C.c_int32), ("max_positions", C.c_int32), ("rope", C.c_int32),
        ("rope_theta", C.c_float), ("pad_token_id", C.c_int32), ("eos_token_id", C.c_int32), ("bos_token_id", C.c_int32),
        ("dtype", C.c_int32), ("max_batch", C.c_int32), ("max_ctx", C.c_int32), ("max_enc", C.c_int32),
        ("max_prompt", C.c_int32), ("device", C.c_int32), ("num_kv_heads", C.c_int32), ("num_cross_kv_heads", C.c_int32),
        ("weights_fp8", C.c_int32), ("kv_fp8", C.c_int32),
    ]


class PttsGenParams(C.Structure):
    _fields_ = [
        ("max_length", C.c_int32), ("min_new_tokens", C.c_int32), ("do_sample", C.c_int32), ("temperature", C.c_float),
        ("top_k", C.c_int32), ("top_p", C.c_float), ("use_eos_gate", C.c_int32), ("seed", C.c_uint64),
    ]


class PttsT5Config(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("d_model", C.c_int32), ("d_kv", C.c_int32), ("d_ff", C.c_int32), ("num_layers", C.c_int32),
        ("num_heads", C.c_int32), ("rel_buckets", C.c_int32), ("rel_max_distance", C.c_int32), ("layer_norm_eps", C.c_float),
        ("dtype", C.c_int32), ("max_batch", C.c_int32), ("max_len", C.c_int32), ("device", C.c_int32),
    ]


class PttsDacConfig(C.Structure):
    _fields_ = [
        ("num_codebooks", C.c_int32), ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32), ("latent_dim", C.c_int32),
        ("decoder_dim", C.c_int32), ("num_rates", C.c_int32), ("rates", C.c_int32 * 8), ("compute_dtype", C.c_int32),
        ("max_batch", C.c_int32), ("max_frames", C.c_int32), ("device", C.c_int32), ("encoder_dim", C.c_int32),
    ]


ABI_VERSION = 8  # PTTS_ABI_VERSION in include/ptts.h

# every symbol include/ptts.h declares: name -> (restype, argtypes)
_VP, _I32, _I64P = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
SYMBOLS = {
    "ptts_last_error": (C.c_char_p, []),
    "ptts_abi_version": (C.c_int, []),
    "ptts_engine_create": (C.c_int, [C.POINTER(PttsConfig), C.POINTER(_VP)]),
    "ptts_engine_destroy": (None, [_VP]),
    "ptts_load_weight": (C.c_int, [_VP, C.c_char_p, _VP, _I32, _I64P, _I32, _VP]),
    "ptts_load_weight_fp8": (C.c_int, [_VP, C.c_char_p, _VP, _VP, _I64P, _I32, _VP]),
    "ptts_weights_ready": (C.c_int, [_VP]),
    "ptts_set_gen_params": (C.c_int, [_VP, C.POINTER(PttsGenParams)]),
    "ptts_prefill": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP]),
    "ptts_first_token_sync": (C.c_int, [_VP]),
    "ptts_first_token_times": (C.c_int, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "ptts_decode_steps": (C.c_int, [_VP, _I32, _VP]),
    "ptts_state": (C.c_int, [_VP, C.POINTER(_I32), C.POINTER(_I32), _VP]),
    "ptts_ids": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_I32)]),
    "ptts_step_forward": (C.c_int, [_VP, _VP]),
    "ptts_logits": (C.c_int, [_VP, C.POINTER(_VP)]),
    "ptts_push_tokens": (C.c_int, [_VP, _VP, _VP, _VP]),
    "ptts_debug_hidden": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_I32)]),
    "ptts_debug_graph_nodes": (C.c_int, [_VP, C.POINTER(_I32)]),
    "ptts_set_audio_prefix": (C.c_int, [_VP, _VP, _I32, _I32, _VP]),
    "ptts_dac_create": (C.c_int, [C.POINTER(PttsDacConfig), C.POINTER(_VP)]),
    "ptts_dac_destroy": (None, [_VP]),
    "ptts_dac_load_weight": (C.c_int, [_VP, C.c_char_p, _VP, _I64P, _I32, _VP]),
    "ptts_dac_weights_ready": (C.c_int, [_VP]),
    "ptts_dac_decode": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP]),
    "ptts_dac_compact_codes": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _VP]),
    "ptts_dac_decode_ragged": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _VP]),
    "ptts_dac_decode_chunk": (C.c_int, [_VP, _VP, C.c_int64, _I32, _I32, _I32, _VP, C.c_int64, _I32, _I32, _VP]),
    "ptts_dac_encode": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    "ptts_dac_debug_latents": (C.c_int, [_VP, C.POINTER(_VP)]),
    "ptts_t5_create": (C.c_int, [C.POINTER(PttsT5Config), C.POINTER(_VP)]),
    "ptts_t5_destroy": (None, [_VP]),
    "ptts_t5_load_weight": (C.c_int, [_VP, C.c_char_p, _VP, _I32, _I64P, _I32, _VP]),
    "ptts_t5_weights_ready": (C.c_int, [_VP]),
    "ptts_t5_encode": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP, _VP]),
    "ptts_t5_relative_bucket": (C.c_int32, [_I32, _I32, _I32]),
    "ptts_t5_debug_graph_nodes": (C.c_int, [_VP, C.POINTER(_I32)]),
    "ptts_dac_debug_decode_upto": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _VP, C.POINTER(_VP), C.POINTER(_I32), C.POINTER(_VP), C.POINTER(_I32), C.POINTER(_I32)]),
}

_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Loads libptts_hip.so (built in-tree by ``__graft_entry__.build()``) and binds every ptts.h symbol."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # torch bundles its own HIP runtime (soname libamdhip64.so.7); importing it first makes this library bind to
    # the SAME runtime instance, so torch streams / device pointers are valid on both sides of the C ABI.
    import torch  # noqa: F401

    p = path or os.environ.get("PTTS_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise NativeLibraryError(
            f"{p} not found: the HIP library is required (there is no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    try:
        lib = C.CDLL(p)
    except OSError as e:  # e.g. libamdhip64 missing
        raise NativeLibraryError(f"failed to load {p}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ptts_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: library {lib.ptts_abi_version()} != binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


_hip: Optional[C.CDLL] = None


def hip_runtime() -> C.CDLL:
    """The HIP runtime instance shared with torch (resolved by soname: already loaded by `import torch`)."""
    global _hip
    if _hip is None:
        import torch  # noqa: F401

        _hip = C.CDLL("libamdhip64.so.7")
        _hip.hipMemcpyAsync.restype = C.c_int
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _hip.hipMemcpy2DAsync.restype = C.c_int
        _hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    return _hip


def check(rc: int, what: str = "") -> None:
    """Maps ptts error classes onto the reference's Python exceptions (SURVEY.md §8(b) 'Errors')."""
    if rc == PTTS_OK:
        return
    msg = load_library().ptts_last_error().decode("utf-8", "replace")
    full = f"{what}: {msg}" if what else msg
    if rc in (PTTS_E_INVALID, PTTS_E_CAPACITY, PTTS_E_MISSING):
        raise ValueError(full)
    if rc == PTTS_E_UNSUPPORTED:
        raise NotImplementedError(full)
    raise NativeLibraryError(full)
