"""ctypes binding of liblatte_amd.so — the ONLY compute backend of this package.

There is no CPU / PyTorch fallback: if the HIP library is missing or no MI355X is visible the
product path raises.  `import torch` must precede loading the library so that it binds to the HIP
runtime torch already loaded (one runtime per process -> torch data_ptr()s and streams are valid).
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported before the engine library, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblatte_amd.so")

c_int, c_i64, c_f32, c_void, c_char = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_char_p
c_u64 = ctypes.c_uint64


class LatteError(RuntimeError):
    pass


class ModelConfig(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("input_size", "patch_size", "in_channels", "hidden_size", "depth", "num_heads",
                                     "mlp_hidden", "num_frames", "num_classes", "learn_sigma", "extras",
                                     "compute_dtype")]


class T2VConfig(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("num_attention_heads", "attention_head_dim", "in_channels", "out_channels", "num_layers",
                                     "sample_size", "patch_size", "cross_attention_dim", "caption_channels", "video_length",
                                     "max_text_tokens", "compute_dtype")]


DTYPES = {"bf16": 0, "bfloat16": 0, "f16": 1, "fp16": 1, "float16": 1}

# name -> (restype, argtypes); mirrors include/latte_amd.h and include/latte_amd_debug.h
PROTOTYPES = {
    "latte_last_error": (c_char, []),
    "latte_version": (c_char, []),
    "latte_schedule_create": (c_int, [c_int, c_char, c_char, ctypes.POINTER(c_void)]),
    "latte_schedule_destroy": (None, [c_void]),
    "latte_schedule_set_model_types": (c_int, [c_void, c_int, c_int, c_int]),
    "latte_schedule_num_timesteps": (c_int, [c_void]),
    "latte_schedule_timestep_map": (c_int, [c_void, c_void, c_int]),
    "latte_schedule_table": (c_int, [c_void, c_char, c_void, c_int]),
    "latte_engine_create": (c_int, [ctypes.POINTER(ModelConfig), c_int, ctypes.POINTER(c_void)]),
    "latte_engine_destroy": (None, [c_void]),
    "latte_engine_set_option": (c_int, [c_void, c_char, c_i64]),
    "latte_engine_get_option": (c_int, [c_void, c_char, ctypes.POINTER(c_i64)]),
    "latte_engine_load_tensor": (c_int, [c_void, c_char, c_void, c_i64, c_int, c_void]),
    "latte_engine_check_weights": (c_int, [c_void]),
    "latte_engine_num_keys": (c_int, [c_void]),
    "latte_engine_key": (c_char, [c_void, c_int]),
    "latte_engine_temb_table": (c_int, [c_void, c_void, c_void, c_void]),
    "latte_engine_set_temb_table": (c_int, [c_void, c_void, c_void, c_void]),
    "latte_engine_set_text_embedding": (c_int, [c_void, c_void, c_int, c_void]),
    "latte_forward": (c_int, [c_void, c_void, c_void, c_void, c_int, c_void, c_void]),
    "latte_forward_with_cfg": (c_int, [c_void, c_void, c_void, c_void, c_int, c_f32, c_void, c_void]),
    "latte_sampler_step_ex": (c_int, [c_void, c_int, c_int, c_f32, c_int, c_void, c_void, c_void, c_void, c_void, c_int, c_int,
                                      c_int, c_int, c_int, c_void, c_void, c_void]),
    "latte_sampler_step": (c_int, [c_void, c_int, c_int, c_f32, c_int, c_void, c_void, c_void, c_int, c_int, c_int,
                                   c_int, c_void, c_void, c_void]),
    "latte_sample_loop": (c_int, [c_void, c_void, c_int, c_f32, c_int, c_f32, c_void, c_void, c_int, c_int, c_int,
                                  c_void, c_void, c_void, c_void]),
    "latte_sample_loop_ex": (c_int, [c_void, c_void, c_int, c_f32, c_int, c_int, c_f32, c_void, c_void, c_int, c_int, c_int,
                                     c_void, c_void, c_void, c_void]),
    "latte_q_sample": (c_int, [c_void, c_void, c_void, c_void, c_int, c_i64, c_void, c_void]),
    "latte_training_workspace_floats": (c_i64, [c_int, c_i64]),
    "latte_training_losses": (c_int, [c_void, c_int, c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_void,
                                      c_i64, c_void, c_void, c_void, c_void]),
    "latte_trainer_create": (c_int, [ctypes.POINTER(ModelConfig), c_int, ctypes.POINTER(c_void)]),
    "latte_trainer_destroy": (None, [c_void]),
    "latte_trainer_num_params": (c_int, [c_void]),
    "latte_trainer_param_key": (c_char, [c_void, c_int]),
    "latte_trainer_param_offset": (c_i64, [c_void, c_int]),
    "latte_trainer_param_numel": (c_i64, [c_void, c_int]),
    "latte_trainer_total_numel": (c_i64, [c_void]),
    "latte_trainer_bind": (c_int, [c_void, c_void, c_void, c_void, c_void, c_void]),
    "latte_trainer_set_frozen": (c_int, [c_void, c_void, c_void, c_int, c_void]),
    "latte_trainer_sync_weights": (c_int, [c_void, c_void]),
    "latte_trainer_forward_backward": (c_int, [c_void, c_void, c_int, c_void, c_void, c_void, c_void, c_int, c_void, c_void, c_void]),
    "latte_trainer_begin": (c_int, [c_void, c_void, c_int, c_void, c_void, c_void, c_void, c_int, c_void, c_void, c_void]),
    "latte_trainer_num_stages": (c_int, [c_void]),
    "latte_trainer_stage_range": (c_int, [c_void, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "latte_trainer_backward_stage": (c_int, [c_void, c_int, c_void]),
    "latte_trainer_optimizer_step": (c_int, [c_void, c_f32, c_f32, c_f32, c_f32, c_f32, c_int, c_f32, c_int, c_f32, c_void, c_void]),
    "latte_trainer_set_option": (c_int, [c_void, c_char, ctypes.c_double]),
    "latte_trainer_scaler_state": (c_int, [c_void, ctypes.POINTER(ctypes.c_double)]),
    "latte_profile_forward": (c_int, [c_void, c_void, c_void, c_void, c_int, c_void, c_void, c_void, c_int, c_void]),
    "latte_profile_forward_ex": (c_int, [c_void, c_void, c_void, c_void, c_int, c_int, c_void, c_void, c_void, c_int, c_void]),
    "latte_t2v_create": (c_int, [ctypes.POINTER(T2VConfig), c_int, ctypes.POINTER(c_void)]),
    "latte_t2v_destroy": (None, [c_void]),
    "latte_t2v_num_keys": (c_int, [c_void]),
    "latte_t2v_key": (c_char, [c_void, c_int]),
    "latte_t2v_load_tensor": (c_int, [c_void, c_char, c_void, c_i64, c_int, c_void]),
    "latte_t2v_check_weights": (c_int, [c_void]),
    "latte_t2v_set_option": (c_int, [c_void, c_char, c_i64]),
    "latte_t2v_forward": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_void, c_void]),
    "latte_t2v_set_text": (c_int, [c_void, c_void, c_void, c_int, c_int, c_void]),
    "latte_t2v_guided_ddim_loop": (c_int, [c_void, c_void, c_int, c_int, c_void, c_void, c_void, c_f32, c_int, c_void]),
    "latte_bench_gemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_f32), c_void]),
    "latte_vae_create": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_void)]),
    "latte_vae_create_temporal": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_void)]),
    "latte_vae_destroy": (None, [c_void]),
    "latte_vae_num_keys": (c_int, [c_void]),
    "latte_vae_key": (c_char, [c_void, c_int]),
    "latte_vae_load_tensor": (c_int, [c_void, c_char, c_void, c_i64, c_int, c_void]),
    "latte_vae_check_weights": (c_int, [c_void]),
    "latte_vae_decode": (c_int, [c_void, c_void, c_int, c_f32, c_int, c_void, c_void]),
    "latte_vae_profile_decode": (c_int, [c_void, c_void, c_int, c_f32, c_int, c_void, c_void, c_void, c_int, c_void]),
    # test hooks
    "latte_debug_gemm_lo8": (c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_void]),
    "latte_debug_attention_split8": (c_int, [c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_int, c_void]),
    "latte_debug_pack_w4": (c_int, [c_void, c_void, c_void, c_int, c_int, c_int, c_void]),
    "latte_debug_ln_modulate_split4": (c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_void]),
    "latte_debug_gemm_lo4": (c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_void]),
    "latte_debug_pack_w8": (c_int, [c_void, c_void, c_i64, c_int, c_void]),
    "latte_debug_ln_modulate_split8": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_void]),
    "latte_debug_qkv_attention_split8": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                 c_void]),
    "latte_debug_gemm": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_void]),
    "latte_debug_gemm_gelu": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_void]),
    "latte_debug_gemm_choice": (c_int, [c_int, c_int, c_int, c_int]),
    "latte_debug_qkv_attention_fusable": (c_int, [c_int, c_int, c_int, c_int, c_int, c_i64]),
    "latte_debug_gemm_tn_plan": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int)]),
    "latte_debug_attention": (c_int, [c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_int,
                                      c_void]),
    "latte_debug_qkv_attention": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_void]),
    "latte_debug_qkv_attention_trace": (c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int,
                                                c_int, c_int, c_int, c_void]),
    "latte_debug_attention_bwd": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64,
                                          c_int, c_void]),
    "latte_debug_gemm_tn": (c_int, [c_void, c_void, c_void, c_void, c_i64, c_int, c_int, c_int, c_int, c_void]),
    "latte_debug_gemm_tn_colsum": (c_int, [c_void, c_void, c_void, c_void, c_void, c_i64, c_int, c_int, c_int, c_int, c_void]),
    "latte_debug_ln_modulate": (c_int, [c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_void, c_int,
                                        c_int, c_int, c_void]),
    "latte_debug_convert": (c_int, [c_void, c_void, c_i64, c_int, c_void]),
    "latte_debug_fill_normal": (c_int, [c_void, c_i64, c_u64, c_u64, c_void]),
    "latte_debug_tr16_probe": (c_int, [c_void, c_void]),
    "latte_debug_set_choice": (c_int, [c_char, c_int]),
    "latte_debug_dma_probe": (c_int, [c_void, c_void, c_int, c_int, c_int, c_void]),
    "latte_debug_conv3x3": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_void]),
    "latte_debug_vae_trace": (c_int, [c_void, c_void, c_int, c_f32, c_int, c_void, c_void, c_void, c_void]),
    "latte_debug_conv3rows_f32": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_void]),
    "latte_debug_conv3x3_f32": (c_int, [c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void]),
    "latte_debug_groupnorm_f32": (c_int, [c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_void]),
    "latte_debug_groupnorm": (c_int, [c_void, c_void, c_void, c_void, c_int, c_int, c_int, c_int, c_int, c_void]),
}

_lib = None


def load_library():
    """Load liblatte_amd.so (raises LatteError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("LATTE_AMD_LIB") or LIB_PATH     # LATTE_AMD_LIB: the measurement build (latte_amd/build.py), tools only
    if not os.path.exists(path):
        raise LatteError(
            f"{path} not found: build it with `python -m latte_amd.build` (hipcc, gfx950). "
            "latte_amd has no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue  # reported by tests/test_abi.py; entry points of later build stages
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def require_gpu():
    if not torch.cuda.is_available():
        raise LatteError("latte_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; there is no CPU fallback")


def check(rc):
    if rc != 0:
        msg = load_library().latte_last_error()
        raise LatteError((msg or b"unknown error").decode() + f" [code {rc}]")


def stream_ptr():
    return c_void(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void(t.data_ptr()) if t is not None else c_void(None)
