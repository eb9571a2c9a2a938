"""ctypes binding of ``libseedstory_hip.so`` (the C ABI declared in ``include/seedstory_hip.h``).

There is deliberately NO fallback: if the shared library is missing or a call fails, an
exception is raised.  The product path never computes on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEEDSTORY_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libseedstory_hip.so"))

SS_F32, SS_BF16, SS_F16 = 0, 1, 2
SS_FILTER_BILINEAR, SS_FILTER_BICUBIC = 0, 1
EPI_NONE, EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_SILU_MUL, EPI_GEGLU_PAIR = 0, 1, 2, 4, 8, 16

vp, i64, i32, f32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
i32p = C.POINTER(C.c_int32)


class SSError(RuntimeError):
    pass


class LlamaConfig(C.Structure):
    _fields_ = [("hidden", i32), ("n_heads", i32), ("n_layers", i32), ("inter", i32), ("vocab", i32),
                ("max_pos", i32), ("rms_eps", f32), ("dtype", i32), ("cache_cap", i32), ("max_new", i32),
                ("n_img_ids", i32), ("eos_id", i32), ("n_seq", i32)]


class LlamaLayerWeights(C.Structure):
    _fields_ = [("wqkv", vp), ("wo", vp), ("wgu", vp), ("wdown", vp), ("ln1", vp), ("ln2", vp)]


class LlamaWeights(C.Structure):
    _fields_ = [("embed", vp), ("lm_head", vp), ("final_norm", vp), ("rope_cos", vp), ("rope_sin", vp),
                ("layers", C.POINTER(LlamaLayerWeights))]


class ResamplerWeights(C.Structure):
    _fields_ = [("q_in", vp), ("pos_kv", vp), ("kv_proj", vp), ("ln_kv_w", vp), ("ln_kv_b", vp), ("in_w", vp),
                ("in_b", vp), ("out_w", vp), ("out_b", vp), ("nq", i32), ("embed", i32), ("n_heads", i32),
                ("kv_dim", i32), ("l_kv", i32), ("ln_eps", f32)]


class VitLayerWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "in_w", "in_b", "out_w", "out_b", "fc_w",
                                  "fc_b", "proj_w", "proj_b")]


class VitWeights(C.Structure):
    _fields_ = [("conv_w", vp), ("pos", vp), ("ln_pre_w", vp), ("ln_pre_b", vp),
                ("layers", C.POINTER(VitLayerWeights)), ("width", i32), ("n_layers", i32), ("n_heads", i32),
                ("mlp_width", i32), ("patch", i32), ("image", i32), ("kpad", i32), ("ln_eps", f32)]


# name -> (restype, argtypes); every symbol declared in include/seedstory_hip.h
PROTOTYPES = {
    "ss_last_error": (C.c_char_p, []),
    "ss_abi_version": (C.c_int, []),
    "ss_device_info": (C.c_int, [i32p]),
    "ss_rowstats": (C.c_int, [vp, i64, i64, i64, f32, vp, vp, C.c_int, vp]),
    "ss_gemm_lnfold": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "ss_gemm_rowstat": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, i64, C.c_int, vp, C.c_int, vp]),
    "ss_gemm_rowpart_strips": (i64, [i64, i64, i64, C.c_int]),
    "ss_gemm_rowpart": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, i64, C.c_int, vp, C.c_int, vp]),
    "ss_gemm_lnfold_part": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, vp, i64, i64, f32, vp, vp, C.c_int, C.c_int, vp]),
    "ss_rowstat_finalize": (C.c_int, [vp, i64, i64, f32, vp, vp, vp]),
    "ss_gemm_splitk_workspace_bytes": (C.c_size_t, [i64, i64, i64]),
    "ss_gemm_splitk": (C.c_int, [vp, vp, vp, i64, i64, i64, vp, vp, vp, C.c_size_t, C.c_int, vp]),
    "ss_quantize_rows_fp8": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp, f32, C.c_int, vp]),
    "ss_gemm_fp8": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, i64, vp, vp, i64, C.c_int, vp]),
    "ss_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "ss_context_info": (C.c_int, [vp, C.POINTER(i64)]),
    "ss_destroy": (None, [vp]),
    "ss_rccl_unique_id": (C.c_int, [vp]),
    "ss_rccl_init": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "ss_rccl_destroy": (None, [vp]),
    "ss_rccl_send": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp]),
    "ss_rccl_recv": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp]),
    "ss_rccl_bcast": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp]),
    "ss_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "ss_get_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "ss_rmsnorm": (C.c_int, [vp, vp, vp, i64, i64, f32, C.c_int, vp]),
    "ss_layernorm": (C.c_int, [vp, vp, vp, vp, i64, i64, f32, C.c_int, vp]),
    "ss_add_bcast": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, C.c_int, vp]),
    "ss_silu_mul": (C.c_int, [vp, vp, i64, i64, C.c_int, vp]),
    "ss_gather_rows": (C.c_int, [vp, vp, vp, i64, i64, C.c_int, vp]),
    "ss_scatter_rows": (C.c_int, [vp, vp, vp, i64, i64, C.c_int, vp]),
    "ss_im2col_patch": (C.c_int, [vp, vp, i64, i64, i64, i64, C.c_int, vp]),
    "ss_l2normalize_dim1": (C.c_int, [vp, vp, i64, i64, i64, C.c_int, vp]),
    "ss_rope_kv_append": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, C.c_int, vp]),
    "ss_attention": (C.c_int, [vp, vp, vp, vp] + [i64] * 17 + [f32, C.c_int, C.c_int, vp]),
    "ss_attention_ragged": (C.c_int, [vp, vp, vp, vp] + [i64] * 3 + [vp] + [i64] * 13 + [f32, C.c_int, C.c_int, vp]),
    "ss_attn_decode_workspace_bytes": (sz, [i64, i64]),
    "ss_attn_decode": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, i64, C.c_int, vp]),
    "ss_gemm": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, i64, C.c_int, C.c_int, vp]),
    "ss_gemm_tune_workspace_bytes": (sz, [i64, i64, i64, C.c_int]),
    "ss_gemm_tune": (C.c_int, [i64, i64, i64, C.c_int, C.c_int, vp, sz, vp, C.POINTER(f32)]),
    "ss_conv3x3_tune_workspace_bytes": (sz, [i64, i64, i64, i64, i64, i64, i64, C.c_int]),
    "ss_conv3x3_tune": (C.c_int, [i64, i64, i64, i64, i64, i64, i64, C.c_int, vp, sz, vp, C.POINTER(f32)]),
    "ss_tune_lookup": (C.c_int, [i64, i64, i64, i64, i64, i64, i64, i64, C.c_int, i32p]),
    "ss_tune_export": (i64, [i32p, i64]),
    "ss_tune_import": (C.c_int, [i32p, i64]),
    "ss_tune_clear": (C.c_int, []),
    "ss_gemv": (C.c_int, [vp, vp, vp, i64, i64, vp, f32, vp, vp, C.c_int, C.c_int, vp]),
    "ss_gemv_batched": (C.c_int, [vp, vp, vp, i64, i64, i64, vp, f32, vp, vp, C.c_int, C.c_int, vp]),
    "ss_imgproc_argmax": (C.c_int, [vp, i64, vp, vp, i64, vp, C.c_int, vp]),
    "ss_llama_workspace_bytes": (sz, [C.POINTER(LlamaConfig), i64]),
    "ss_llama_create": (C.c_int, [C.POINTER(LlamaConfig), C.POINTER(LlamaWeights), vp, sz, i64, i32p,
                                  C.POINTER(vp)]),
    "ss_llama_destroy": (None, [vp]),
    "ss_llama_select": (C.c_int, [vp, i32]),
    "ss_llama_set_stop_id": (C.c_int, [vp, i32]),
    "ss_llama_buffer": (vp, [vp, C.c_int]),
    "ss_llama_set_lengths": (C.c_int, [vp, i64, i64, vp]),
    "ss_llama_get_lengths": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64)]),
    "ss_llama_kv_gather": (C.c_int, [vp, vp, i64, vp]),
    "ss_llama_prefill": (C.c_int, [vp, vp, i64, vp, vp, vp]),
    "ss_llama_prefill_batch": (C.c_int, [vp, vp, C.POINTER(i64), vp, vp]),
    "ss_llama_generate": (C.c_int, [vp, i64, i32, i32p, i64, C.POINTER(i64), vp]),
    "ss_llama_generate_batch": (C.c_int, [vp, i64, i32p, i32p, i64, C.POINTER(i64), i32p, C.POINTER(i64), vp]),
    "ss_llama_profile_decode": (C.c_int, [vp, i64, C.POINTER(f32), C.POINTER(C.c_double), vp]),
    "ss_resampler_workspace_bytes": (sz, [C.POINTER(ResamplerWeights), i64, C.c_int]),
    "ss_resampler_forward": (C.c_int, [C.POINTER(ResamplerWeights), vp, vp, i64, vp, sz, C.c_int, vp]),
    "ss_vit_workspace_bytes": (sz, [C.POINTER(VitWeights), i64, C.c_int]),
    "ss_vit_forward": (C.c_int, [C.POINTER(VitWeights), vp, vp, i64, vp, sz, C.c_int, vp]),
    "ss_vit_blocks": (C.c_int, [C.POINTER(VitWeights), vp, i64, i64, i64, i64, vp, sz, C.c_int, vp]),
    "ss_conv3x3": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, vp, vp, i64, vp, C.c_int, vp]),
    "ss_loss_workspace_bytes": (C.c_size_t, [i64]),
    "ss_cross_entropy_rows": (C.c_int, [vp, i64, vp, i64, i64, i64, vp, vp, C.c_int, vp]),
    "ss_cosine_rows": (C.c_int, [vp, vp, i64, i64, vp, C.c_int, vp]),
    "ss_masked_mean": (C.c_int, [vp, vp, i64, vp, vp, vp]),
    "ss_mse": (C.c_int, [vp, vp, i64, vp, vp, C.c_int, vp]),
    "ss_groupnorm_workspace_bytes": (C.c_size_t, [i64, i64, i64, i64, C.c_int]),
    "ss_groupnorm": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, i64, i64, f32, C.c_int, C.c_int, vp]),
    "ss_geglu": (C.c_int, [vp, vp, i64, i64, C.c_int, vp]),
    "ss_unary": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp]),
    "ss_softmax_rows": (C.c_int, [vp, i64, i64, f32, C.c_int, vp]),
    "ss_transpose": (C.c_int, [vp, vp, i64, i64, C.c_int, vp]),
    "ss_concat_channels": (C.c_int, [vp, vp, vp, i64, i64, i64, C.c_int, vp]),
    "ss_layout_nchw_nhwc": (C.c_int, [vp, vp, i64, i64, i64, i64, C.c_int, C.c_int, vp]),
    "ss_euler_scale_dup": (C.c_int, [vp, vp, i64, f32, C.c_int, vp]),
    "ss_euler_cfg_step": (C.c_int, [vp, vp, i64, f32, f32, f32, C.c_int, vp]),
    "ss_resample_ksize": (C.c_int, [i64, i64, C.c_int]),
    "ss_resample_coeffs": (C.c_int, [i64, i64, C.c_int, i32p, i32p]),
    "ss_image_preprocess": (C.c_int, [vp, i64, i64, vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, C.c_int, vp, vp, C.c_int,
                                      i64, i64, vp, C.POINTER(f32), C.POINTER(f32), C.c_int, vp]),
    "ss_image_to_u8": (C.c_int, [vp, vp, i64, i64, C.c_int, vp]),
    "ss_debug_tr_probe": (C.c_int, [vp, vp, vp, vp]),
}

_lib = None


def lib():
    """The loaded library; raises if it was not built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SSError("libseedstory_hip.so not found at %s — build it first (seed-story_amd/csrc/Makefile); "
                          "there is no CPU fallback" % LIB_PATH)
        # torch ships its own libamdhip64.so.7 (+ HSA runtime); it must be the copy this process binds
        # to, or the two runtimes disagree about the device ("no ROCm-capable device").  Import torch
        # first so the library's DT_NEEDED libamdhip64.so.7 resolves to the already-loaded one.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.ss_abi_version() != 1:
            raise SSError("ABI version mismatch")
        _lib = l
        from . import tune
        tune.load_default_table()
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ss_last_error()
        raise SSError("%s failed (%d): %s" % (what or "seedstory call", rc, msg.decode() if msg else "?"))


_tuning_gen = 0     # bumped by every set_tuning: captured hipGraphs replay the arithmetic / kernels chosen at capture time


def set_tuning(key: str, value: int):
    global _tuning_gen
    check(lib().ss_set_tuning(key.encode(), int(value)), "ss_set_tuning")
    _tuning_gen += 1


def tuning_generation() -> int:
    """Counter of ``set_tuning`` calls: part of the key of every captured graph on the Python side (toggling
    ``gemm_f32_split`` / ``attn_ver`` after a first render used to replay the old arithmetic)."""
    return _tuning_gen


def get_tuning(key: str, default: int = 0) -> int:
    return int(lib().ss_get_tuning(key.encode(), int(default)))
