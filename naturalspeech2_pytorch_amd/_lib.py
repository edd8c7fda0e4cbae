"""ctypes binding of libns2hip.so (C ABI: include/ns2hip.h).

The HIP library is the product: importing this module FAILS LOUDLY when the shared object is missing or does
not export the full ABI — there is no CPU / PyTorch fallback for the hot path.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NS2_LIB", os.path.join(_HERE, "libns2hip.so"))   # NS2_LIB: experiment builds (tools/)


class Ns2Error(RuntimeError):
    pass


class ModelConfig(ctypes.Structure):
    _fields_ = [(n, c_int) for n in (
        "dim", "depth", "dim_head", "heads", "ff_mult", "wavenet_layers", "wavenet_stacks", "dim_cond_mult",
        "condition_on_prompt", "dim_prompt", "num_latents_m", "resampler_depth", "precision")]


P = c_void_p
I = c_int
F = c_float
L = c_int64


class AttnBwdArgs(ctypes.Structure):
    """ns2_attn_bwd_args (include/ns2hip.h), field for field"""
    _fields_ = [("q_hi", P), ("q_lo", P), ("ldq", I), ("q_col0", I),
                ("k_hi", P), ("k_lo", P), ("ldk", I), ("k_col0", I),
                ("v_hi", P), ("v_lo", P), ("ldv", I), ("v_col0", I),
                ("do_hi", P), ("do_lo", P), ("lddo", I),
                ("lse", P), ("delta", P),
                ("dq", P), ("lddq", I), ("dq_col0", I),
                ("dk", P), ("lddk", I), ("dk_col0", I),
                ("dv", P), ("lddv", I), ("dv_col0", I),
                ("B", I), ("H", I), ("Nq", I), ("Nk", I), ("scale", F),
                ("gp_hi", P), ("gp_lo", P), ("gp_ld", I), ("gp_precision", I), ("gp_q", I), ("gp_kv", I)]


class RepackPart(ctypes.Structure):
    """ns2_repack_part (include/ns2hip.h), field for field"""
    _fields_ = [("w", P), ("src", P), ("sr", L), ("sc", L), ("st", L), ("row0", I), ("rows", I), ("col0", I), ("cols", I)]


# name -> (restype, argtypes); mirrors include/ns2hip.h line by line
SIGNATURES = {
    "ns2_last_error": (c_char_p, []),
    "ns2_version": (I, []),
    "ns2_debug_force_gemm": (I, [I]),
    "ns2_splitk_scratch_bytes": (L, []),
    "ns2_debug_splitk_plan": (I, [I, I, I, I, I, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "ns2_debug_lend_splitk_scratch": (I, [P, L]),
    "ns2_weight_pack": (I, [P, I, I, I, I, P, I, POINTER(c_void_p), P]),
    "ns2_weight_free": (None, [P]),
    "ns2_split_f32": (I, [P, I, I, I, P, P, I, I, P]),
    "ns2_join_f32": (I, [P, P, I, P, I, L, I, I, P]),
    "ns2_linear_f32": (I, [P, P, P, I, I, I, I, I, P, P, I, P, I, I, I, I, P]),
    "ns2_linear_split": (I, [P, P, P, I, I, I, I, I, P, P, P, I, I, I, I, P]),
    "ns2_linear_split_as": (I, [P, P, P, I, I, I, I, I, P, P, P, I, I, I, I, I, P]),
    "ns2_linear_geglu": (I, [P, P, P, I, I, P, P, P, I, I, P]),
    "ns2_geglu_pack_bias": (I, [P, I, P, I, P]),
    "ns2_linear_qkv": (I, [P, P, P, I, I, I, I, P, P, I, P, P, I, I, P]),
    "ns2_wavenet_block": (I, [P, P, P, I, I, I, I, P, P, P, I, P, P, I, I, P]),
    "ns2_attention": (I, [P, P, I, I, P, P, I, I, P, P, I, P, P, I, I, I, I, I, F, P, I, P]),
    "ns2_attention_hd": (I, [P, P, I, I, P, P, I, I, P, P, I, P, P, I, I, I, I, I, F, P, I, I, P]),
    "ns2_rmsnorm": (I, [P, I, I, I, I, P, P, I, P, P, I, P, I, I, P]),
    "ns2_skinny_linear_workspace_bytes": (L, [I, I, I]),
    "ns2_skinny_linear": (I, [P, I, P, P, P, I, I, I, I, I, P, L, P]),
    "ns2_time_embed": (I, [P, P, P, P, P, P, I, I, I, I, P, L, P]),
    "ns2_transpose_f32": (I, [P, I, I, I, P, P]),
    "ns2_embedding": (I, [P, P, P, L, I, L, P]),
    "ns2_ddim_step": (I, [P, P, P, P, P, I, L, I, I, F, P]),
    "ns2_cfg_mix": (I, [P, P, P, L, F, P]),
    "ns2_seanet_prep": (I, [P, I, I, P, I, I, L, I, I, I, I, P, P, I, I, P]),
    "ns2_seanet_prep2": (I, [P, I, I, I, L, I, I, P, P, I, I, I, P, P, I, I, I, I, P]),
    "ns2_seanet_conv_narrow": (I, [P, L, I, I, L, I, I, I, I, P, P, P, L, P]),
    "ns2_seanet_resblock_narrow": (I, [P, L, I, I, L, I, P, P, P, P, P, P, L, P]),
    "ns2_seanet_unpad": (I, [P, L, I, P, L, I, L, I, P]),
    "ns2_lstm_state_floats": (L, [I, I]),
    "ns2_lstm_layer": (I, [P, L, P, P, P, L, P, L, P, L, I, L, I, P]),
    "ns2_lstm2_state_floats": (L, []),
    "ns2_lstm2": (I, [P, L, P, P, P, P, P, P, P, L, P, L, P, L, I, L, P]),
    "ns2_lstm_abort_count": (I, [I, POINTER(c_int64)]),
    "ns2_debug_lstm_inject_abort": (I, [I]),
    "ns2_saturation_count": (I, [I, POINTER(c_int64)]),
    "ns2_saturation_peek_async": (I, [P, P]),
    "ns2_rvq_prepare": (I, [P, P, I, I, I, P]),
    "ns2_rvq_encode": (I, [P, P, P, P, P, P, P, I, I, I, I, F, P]),
    "ns2_rvq_decode": (I, [P, P, P, I, I, I, I, P]),
    "ns2_model_create": (I, [POINTER(ModelConfig), POINTER(c_void_p)]),
    "ns2_model_set_param": (I, [P, c_char_p, P, I, POINTER(c_int64)]),
    "ns2_model_finalize": (I, [P, P]),
    "ns2_model_workspace_bytes": (L, [P, I, I, I, I]),
    "ns2_model_cond_bytes": (L, [P, I, I, I, I]),
    "ns2_model_prepare_cond": (I, [P, P, I, P, I, I, I, I, P, P, L, P]),
    "ns2_model_cond_stack": (I, [P, P, P, I, I, I, P, P]),
    "ns2_model_forward": (I, [P, P, P, P, I, P, I, I, P, L, P]),
    "ns2_model_table_cols": (I, [P]),
    "ns2_model_time_table_workspace_bytes": (L, [P, I]),
    "ns2_model_time_table": (I, [P, P, I, I, P, P, L, P]),
    "ns2_model_forward_row": (I, [P, P, P, P, I, P, I, I, P, L, P]),
    "ns2_model_param_count": (I, [P]),
    "ns2_model_param_checksum": (I, [P, P, I, P]),
    "ns2_model_debug_tap": (I, [P, c_char_p, P, L]),
    "ns2_model_profile_begin": (I, [P, ctypes.c_uint]),
    "ns2_model_profile_end": (I, [P, POINTER(ctypes.c_double), POINTER(c_int64)]),
    "ns2_model_destroy": (None, [P]),
    # ---- training: the backward pass
    "ns2_weight_update": (I, [P, P, P, P]),
    "ns2_weight_tile_conv3": (I, [P, P]),
    "ns2_weight_tile_linear": (I, [P, P]),
    "ns2_weight_tile_wavenet": (I, [P, P]),
    "ns2_conv3_input_ld": (I, [I]),
    "ns2_saturation_peek_train_async": (I, [P, P]),
    "ns2_grad_prep_slices": (L, [I, L]),
    "ns2_grad_prep": (I, [P, L, I, I, I, I, P, P, I, P, P, L, I, I, P, I, P]),
    "ns2_planes_transpose": (I, [P, P, I, I, I, I, I, I, P, P, L, I, I, I, P]),
    "ns2_reduce_slices": (I, [P, L, I, L, P, I, P]),
    "ns2_weights_repack_table_bytes": (L, [I]),
    "ns2_weights_repack_build": (I, [P, I, P, L, POINTER(c_int64), P]),
    "ns2_weights_repack": (I, [P, I, L, P]),
    "ns2_weights_retile": (I, [P, I, P]),
    "ns2_wgrad_workspace_bytes": (L, [I, I, L]),
    "ns2_wgrad": (I, [P, P, P, P, L, I, I, I, I, P, P, L, I, P]),
    "ns2_wgrad_rows_preferred": (I, [I, I, L]),
    "ns2_wgrad_rows": (I, [P, P, I, P, P, I, L, I, I, I, I, I, I, P, P, L, I, P]),
    "ns2_film_gate_fwd": (I, [P, L, P, I, I, L, I, P, L, P]),
    "ns2_film_gate_slices": (I, [I]),
    "ns2_film_gate_bwd": (I, [P, L, P, L, P, I, I, I, I, P, L, P, P]),
    "ns2_geglu_fwd": (I, [P, L, L, I, P, P, I, I, P]),
    "ns2_geglu_bwd": (I, [P, L, P, L, L, I, P, L, P]),
    "ns2_rmsnorm_bwd_slices": (I, [I]),
    "ns2_rmsnorm_bwd": (I, [P, L, P, L, P, P, I, I, I, I, P, P, L, P, P, P]),
    "ns2_attention_lse": (I, [P, P, I, I, P, P, I, I, P, P, I, P, P, I, I, I, I, I, F, P, I, I, P]),
    "ns2_attention_delta": (I, [P, L, P, P, I, I, I, I, P, I, P]),
    "ns2_attention_bwd": (I, [POINTER(AttnBwdArgs), P]),
}

NS2_UNAVAILABLE = 1          # include/ns2hip.h: "this fast path does not apply here" (not an error)

_lib = None


def load():
    """Load libns2hip.so and bind every symbol of the ABI; raises Ns2Error if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Ns2Error(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                       f"(or naturalspeech2_pytorch_amd/csrc/build.sh); there is no fallback path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise Ns2Error(f"libns2hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ns2_last_error()
        raise Ns2Error(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def header_symbols():
    """Symbols declared in include/ns2hip.h (used by the CPU test that the .so exports the whole ABI)."""
    import re
    hdr = os.path.join(_HERE, "..", "include", "ns2hip.h")
    return sorted(set(re.findall(r"\b(ns2_[a-z0-9_]+)\(", open(hdr).read())))
