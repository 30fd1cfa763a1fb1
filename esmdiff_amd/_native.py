"""ctypes binding of libesmdiff_hip.so (include/esmdiff_hip.h).  There is NO fallback: if the library is
missing or no gfx950 device is present, every entry point raises."""
from __future__ import annotations

import ctypes
from pathlib import Path

import os

# ESMDIFF_LIB: another build of the same library (ablation / A-B variants made by scratch/build_variant.py); never a fallback
_LIB_PATH = Path(os.environ.get("ESMDIFF_LIB") or Path(__file__).resolve().parent / "lib" / "libesmdiff_hip.so")
_lib = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i64p = ctypes.POINTER(ctypes.c_int64)

EPI_BF16, EPI_RESID_F32, EPI_SWIGLU_BF16, EPI_BIAS_GELU_BF16, EPI_BIAS_F32 = range(5)
DT_F32, DT_BF16 = 0, 1
PRECISION = {"bf16": 0, "f32": 1, "f32_split": 2, "f16": 3}       # esmdiff_precision
F32EPI_STORE, F32EPI_BIAS_GELU, F32EPI_RESID_DIV = range(3)
SECTIONS = ["embed", "layernorm", "gemm_qkv", "qk_norm_rope", "attention", "gemm_out", "gemm_ffn_up",
            "gemm_ffn_down", "head", "sampler"]


class Config(ctypes.Structure):
    _fields_ = [("d_model", ctypes.c_int32), ("n_heads", ctypes.c_int32), ("n_layers", ctypes.c_int32),
                ("ffn_hidden", ctypes.c_int32), ("vocab_out", ctypes.c_int32), ("freq_dim", ctypes.c_int32),
                ("max_batch", ctypes.c_int32), ("max_len", ctypes.c_int32), ("residue_scale", ctypes.c_float),
                ("time_conditioning", ctypes.c_int32), ("precision", ctypes.c_int32), ("head_precision", ctypes.c_int32)]


class Weight(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("dtype", ctypes.c_int32),
                ("ndim", ctypes.c_int32), ("shape", ctypes.c_int64 * 4)]


# esmdiff_sample_step as a numpy record (one per sample; uploaded as raw bytes)
SAMPLE_STEP_DTYPE = [("sample_index", "<u8"), ("move_chance_t", "<f4"), ("move_chance_s", "<f4"), ("step", "<i4"), ("final", "<i4")]
# esmdiff_gibbs_sample_step
GIBBS_STEP_DTYPE = [("sample_index", "<u8"), ("step", "<i4"), ("n_unmask", "<i4")]


class Rng(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("sample_offset", ctypes.c_uint64)]


EXPORTS = [
    "esmdiff_abi_version", "esmdiff_engine_create", "esmdiff_engine_destroy", "esmdiff_last_error",
    "esmdiff_forward_logits", "esmdiff_ddpm_step", "esmdiff_ddpm_sample", "esmdiff_gibbs_step",
    "esmdiff_gibbs_sample", "esmdiff_gemm_bf16", "esmdiff_branch_linear_layernorm",
    "esmdiff_layernorm_bf16", "esmdiff_attention_bf16", "esmdiff_set_profiling",
    "esmdiff_get_profile", "esmdiff_set_frames", "esmdiff_gemm_bf16_ws", "esmdiff_decoder_create",
    "esmdiff_decoder_decode", "esmdiff_metrics_js_pwd", "esmdiff_metrics_js_rg", "esmdiff_metrics_validity",
    "esmdiff_metrics_bonding_validity", "esmdiff_metrics_pwd", "esmdiff_metrics_js_columns", "esmdiff_encoder_create", "esmdiff_encoder_destroy", "esmdiff_encoder_last_error",
    "esmdiff_encoder_encode", "esmdiff_gemm_f32", "esmdiff_set_step0_sharing", "esmdiff_get_counters",
    "esmdiff_set_gibbs_options", "esmdiff_split_rows", "esmdiff_split_weight", "esmdiff_gemm_split",
    "esmdiff_get_embeddings", "esmdiff_set_final_skip", "esmdiff_gemm_f16", "esmdiff_ddpm_step_margin", "esmdiff_forward_logits_sigmas", "esmdiff_set_small_batch_splitk",
    "esmdiff_ddpm_step_rows", "esmdiff_logit_error_stats", "esmdiff_get_build_info", "esmdiff_describe_plan", "esmdiff_set_option",
    "esmdiff_get_sequence_logits", "esmdiff_gibbs_step_rows", "esmdiff_shared_forward_batch",
]
OPT_STREAMS, OPT_DUAL_MIN_TOKENS = 1, 2      # esmdiff_option


def lib_path() -> Path:
    return _LIB_PATH


def lib():
    """Load the shared library (building is __graft_entry__.build()'s / `python -m esmdiff_amd.build`'s job)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: PyTorch-ROCm ships its own libamdhip64 / HSA runtime, and the tensors handed to this library live in that
    # runtime's context.  libesmdiff_hip.so only names "libamdhip64.so.7"; loaded after torch it binds to the copy torch
    # already mapped (one runtime per process), loaded before torch it would pull in /opt/rocm's and the process would hold
    # two runtimes — the second one then reports no devices (seen with `python __graft_entry__.py --smoke`, where build()
    # loaded this library before smoke() imported torch).
    import torch  # noqa: F401
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"{_LIB_PATH} not found: build it with `python -m esmdiff_amd.build` (needs hipcc). "
            "esmdiff_amd has no CPU fallback.")
    L = ctypes.CDLL(str(_LIB_PATH))
    vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
    L.esmdiff_abi_version.restype = ctypes.c_int
    L.esmdiff_engine_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(Weight), i32, i32,
                                        ctypes.POINTER(vp)]
    L.esmdiff_engine_destroy.argtypes = [vp]
    L.esmdiff_engine_destroy.restype = None
    L.esmdiff_last_error.argtypes = [vp]
    L.esmdiff_last_error.restype = ctypes.c_char_p
    L.esmdiff_forward_logits.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.esmdiff_forward_logits_sigmas.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.esmdiff_forward_logits_sigmas.restype = ctypes.c_int
    L.esmdiff_ddpm_step.argtypes = [vp, vp, vp, i32, f32, f32, i32, vp, ctypes.POINTER(Rng), i32, i32, i32, vp]
    L.esmdiff_ddpm_step_margin.argtypes = [vp, vp, vp, i32, f32, f32, i32, ctypes.POINTER(Rng), i32, i32, i32, f32, vp, vp]
    L.esmdiff_ddpm_step_margin.restype = ctypes.c_int
    L.esmdiff_ddpm_step_rows.argtypes = [vp, vp, vp, i32, vp, ctypes.c_uint64, i32, i32, f32, f32, vp, vp, vp]
    L.esmdiff_logit_error_stats.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32, vp, vp]
    L.esmdiff_gibbs_step_rows.argtypes = [vp, vp, vp, vp, i32, f32, f32, vp, ctypes.c_uint64, i32, i32, f32, f32, vp, vp, vp]
    L.esmdiff_get_build_info.argtypes = [ctypes.c_char_p, i32]
    L.esmdiff_describe_plan.argtypes = [vp, i32, i32, ctypes.c_char_p, i32]
    L.esmdiff_set_option.argtypes = [vp, i32, ctypes.c_int64]
    L.esmdiff_ddpm_sample.argtypes = [vp, vp, vp, i32, i32, i32, c_f32p, c_f32p, c_f32p, ctypes.POINTER(Rng), vp]
    L.esmdiff_gibbs_step.argtypes = [vp, vp, vp, vp, i32, f32, f32, vp, vp, ctypes.POINTER(Rng), i32, i32, i32, vp]
    L.esmdiff_gibbs_sample.argtypes = [vp, vp, vp, i32, i32, i32, f32, f32, ctypes.POINTER(i32), ctypes.POINTER(Rng), vp]
    L.esmdiff_gemm_bf16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]
    L.esmdiff_gemm_f16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]
    if hasattr(L, "esmdiff_gemm_bf16_timed"):      # -DED_DEBUG builds only (scratch/ A-B scripts)
        L.esmdiff_gemm_bf16_timed.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, i32, c_f32p, vp]
        L.esmdiff_gemm_bf16_timed.restype = ctypes.c_int
        L.esmdiff_debug_graph_ab.argtypes = [vp, vp, vp, i32, i32, i32, c_f32p, c_f32p]
        L.esmdiff_debug_graph_ab.restype = ctypes.c_int
    L.esmdiff_branch_linear_layernorm.argtypes = [vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, ctypes.POINTER(i32), vp]
    L.esmdiff_layernorm_bf16.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.esmdiff_attention_bf16.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    L.esmdiff_set_profiling.argtypes = [vp, i32]
    L.esmdiff_get_profile.argtypes = [vp, c_f32p, ctypes.POINTER(i32)]
    L.esmdiff_set_frames.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.esmdiff_decoder_create.argtypes = L.esmdiff_engine_create.argtypes
    L.esmdiff_decoder_decode.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]
    f64, f64p = ctypes.c_double, ctypes.POINTER(ctypes.c_double)
    L.esmdiff_metrics_js_pwd.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, f64p, vp]
    L.esmdiff_metrics_js_rg.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, i32, f64p, vp]
    L.esmdiff_metrics_pwd.argtypes = [vp, i32, i32, i32, vp, vp]
    L.esmdiff_metrics_js_columns.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, i32, f64p, vp]
    L.esmdiff_metrics_validity.argtypes = [vp, i32, i32, f64, f64, i32, f64p, vp]
    L.esmdiff_metrics_bonding_validity.argtypes = [vp, i32, vp, i32, i32, f64p, vp]
    L.esmdiff_encoder_create.argtypes = [i32] * 9 + [ctypes.POINTER(Weight), i32, i32, ctypes.POINTER(vp)]
    L.esmdiff_encoder_destroy.argtypes = [vp]
    L.esmdiff_encoder_destroy.restype = None
    L.esmdiff_encoder_last_error.argtypes = [vp]
    L.esmdiff_encoder_last_error.restype = ctypes.c_char_p
    L.esmdiff_encoder_encode.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.esmdiff_set_gibbs_options.argtypes = [vp, i32, ctypes.POINTER(i32), i32]
    L.esmdiff_set_step0_sharing.argtypes = [vp, i32]
    L.esmdiff_shared_forward_batch.argtypes = [vp, i32, i32]
    L.esmdiff_set_final_skip.argtypes = [vp, i32]
    L.esmdiff_set_small_batch_splitk.argtypes = [vp, i32]
    L.esmdiff_get_counters.argtypes = [vp, c_i64p, c_i64p, i32]
    L.esmdiff_gemm_f32.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]
    L.esmdiff_get_embeddings.argtypes = [vp, vp, i32, i32, vp]
    L.esmdiff_get_sequence_logits.argtypes = [vp, vp, i32, i32, i32, vp]
    L.esmdiff_split_rows.argtypes = [vp, i32, vp, vp, i32, i32, vp]
    L.esmdiff_split_weight.argtypes = [vp, vp, i32, i32, i32, c_f32p]
    L.esmdiff_gemm_split.argtypes = [vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, f32, i32, vp]
    L.esmdiff_gemm_bf16_ws.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]
    for n in EXPORTS:
        if n not in ("esmdiff_engine_destroy", "esmdiff_last_error", "esmdiff_encoder_destroy", "esmdiff_encoder_last_error"):
            getattr(L, n).restype = ctypes.c_int
    if L.esmdiff_abi_version() != 8:
        raise RuntimeError("libesmdiff_hip.so ABI version mismatch")
    _lib = L
    return L


def build_info() -> str:
    """esmdiff_get_build_info: ABI, arch, whether this is a -DED_DEBUG build (the only kind that reads ESMDIFF_* tuning switches)."""
    buf = ctypes.create_string_buffer(1024)
    lib().esmdiff_get_build_info(buf, 1024)
    return buf.value.decode()


def check(code: int, eng=None):
    if code != 0:
        msg = lib().esmdiff_last_error(eng)
        raise RuntimeError(f"libesmdiff_hip error {code}: {msg.decode() if msg else '?'}")
