"""ctypes binding of libtld_hip.so (the C ABI declared in include/tld_hip.h).

The library is built in-tree by ``csrc/Makefile`` (``__graft_entry__.build()``).  There is no
fallback: if the shared object is missing or a call fails, a RuntimeError carrying
``tld_last_error()`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TLD_LIB", os.path.join(_HERE, "libtld_hip.so"))   # TLD_LIB: A/B-testing builds only

DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2

KERNEL_CLASSES = ("gemm_qkv", "gemm_up", "gemm_down", "attention", "cross_row", "dwconv_gelu", "layernorm",
                  "embed", "tail", "update", "conditioning")

# every symbol include/tld_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "tld_engine_create", "tld_engine_load_tensor", "tld_engine_finalize_weights", "tld_denoiser_forward",
    "tld_sample", "tld_engine_set_gemm_dtype", "tld_engine_set_low_latency", "tld_debug_gemm_splitk", "tld_debug_quant_mx8", "tld_debug_quant_mx8_host", "tld_debug_gemm_mx8",
    "tld_engine_set_debug", "tld_engine_read_stage", "tld_debug_gemm_bf16", "tld_debug_gemm_bench",
    "tld_engine_set_profile", "tld_engine_profile_reserve", "tld_engine_get_profile", "tld_engine_weight_bytes", "tld_engine_destroy",
    "tld_vae_create", "tld_vae_load_tensor", "tld_vae_finalize_weights", "tld_vae_decode", "tld_vae_set_debug",
    "tld_vae_read_stage", "tld_vae_set_profile", "tld_vae_get_profile", "tld_debug_conv3x3", "tld_vae_weight_bytes",
    "tld_vae_destroy",
    "tld_clip_create", "tld_clip_load_tensor", "tld_clip_finalize_weights", "tld_clip_encode_text", "tld_clip_read_buffer", "tld_clip_weight_bytes",
    "tld_clip_destroy",
    "tld_train_create", "tld_train_param_count", "tld_train_tensor_count", "tld_train_param_layout", "tld_train_set_angular_speeds", "tld_train_bind",
    "tld_train_refresh_weights", "tld_train_forward_backward", "tld_train_forward_backward_cb", "tld_train_adam_ema", "tld_debug_attention_bwd", "tld_debug_wgrad", "tld_debug_attention_fwd", "tld_debug_dwconv_gelu", "tld_train_destroy",
    "tld_last_error",
)

VAE_KERNEL_CLASSES = ("conv3x3", "gemm", "groupnorm", "other")


# void (*tld_grad_ready_fn)(void* user, int64_t offset, int64_t numel)   (include/tld_hip.h: tld_train_forward_backward_cb)
GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)


class TldConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("image_size", "noise_embed_dims", "patch_size", "embed_dim", "n_layers",
                                         "text_emb_size", "n_channels", "mlp_multiplier", "max_batch",
                                         "device_id")]


class TldVaeConfig(C.Structure):
    _fields_ = [("latent_channels", C.c_int32), ("out_channels", C.c_int32), ("n_blocks", C.c_int32),
                ("block_out_channels", C.c_int32 * 4), ("layers_per_block", C.c_int32), ("norm_num_groups", C.c_int32),
                ("mid_block_attention", C.c_int32), ("use_post_quant_conv", C.c_int32), ("latent_size", C.c_int32),
                ("max_batch", C.c_int32), ("device_id", C.c_int32)]


class TldClipConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab_size", "context_length", "width", "heads", "layers", "embed_dim", "max_batch",
                                         "device_id")]


_lib = None


def build(verbose: bool = False) -> None:
    """Compile libtld_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode:
        print(r.stdout)
    if r.returncode:
        raise RuntimeError("building libtld_hip.so failed")


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C transformer_latent_diffusion_amd/csrc`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
    L.tld_last_error.restype = C.c_char_p
    L.tld_engine_create.argtypes = [C.POINTER(TldConfig), C.POINTER(vp)]
    L.tld_engine_load_tensor.argtypes = [vp, C.c_char_p, vp, i64p, i32, i32]
    L.tld_engine_finalize_weights.argtypes = [vp]
    L.tld_denoiser_forward.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    L.tld_sample.argtypes = [vp, vp, vp, C.POINTER(C.c_float), i32, C.c_float, C.c_float, C.c_float, vp, i32,
                             vp, vp, vp]
    L.tld_engine_set_gemm_dtype.argtypes = [vp, i32]
    L.tld_engine_set_low_latency.argtypes = [vp, i32]
    L.tld_debug_gemm_splitk.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    L.tld_debug_quant_mx8.argtypes = [vp, vp, vp, i32, i32, vp]
    L.tld_debug_quant_mx8_host.argtypes = [C.POINTER(C.c_float), i32, i32, vp, vp]
    L.tld_debug_gemm_mx8.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.tld_engine_set_debug.argtypes = [vp, i32]
    L.tld_engine_read_stage.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_int64]
    L.tld_debug_gemm_bf16.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.tld_debug_gemm_bench.argtypes = [i32, i32, i32, i32, i32, i32, C.POINTER(C.c_double)]
    L.tld_engine_set_profile.argtypes = [vp, C.c_uint32]
    L.tld_engine_profile_reserve.argtypes = [vp, i32, C.c_int64]
    L.tld_engine_get_profile.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.tld_engine_weight_bytes.argtypes = [vp]
    L.tld_engine_weight_bytes.restype = C.c_int64
    L.tld_engine_destroy.argtypes = [vp]
    L.tld_vae_create.argtypes = [C.POINTER(TldVaeConfig), C.POINTER(vp)]
    L.tld_vae_load_tensor.argtypes = [vp, C.c_char_p, vp, i64p, i32, i32]
    L.tld_vae_finalize_weights.argtypes = [vp]
    L.tld_vae_decode.argtypes = [vp, vp, vp, i32, i32, vp]
    L.tld_vae_set_debug.argtypes = [vp, i32]
    L.tld_vae_read_stage.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_int64, i64p]
    L.tld_vae_set_profile.argtypes = [vp, i32]
    L.tld_vae_get_profile.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.tld_debug_conv3x3.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.tld_vae_weight_bytes.argtypes = [vp]
    L.tld_vae_weight_bytes.restype = C.c_int64
    L.tld_vae_destroy.argtypes = [vp]
    L.tld_clip_create.argtypes = [C.POINTER(TldClipConfig), C.POINTER(vp)]
    L.tld_clip_load_tensor.argtypes = [vp, C.c_char_p, vp, i64p, i32, i32]
    L.tld_clip_finalize_weights.argtypes = [vp]
    L.tld_clip_encode_text.argtypes = [vp, vp, vp, vp, i32, vp]
    L.tld_clip_read_buffer.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_int64]
    L.tld_clip_weight_bytes.argtypes = [vp]
    L.tld_clip_weight_bytes.restype = C.c_int64
    L.tld_clip_destroy.argtypes = [vp]
    f32 = C.c_float
    L.tld_train_create.argtypes = [C.POINTER(TldConfig), C.POINTER(vp)]
    L.tld_train_param_count.argtypes = [vp]
    L.tld_train_param_count.restype = C.c_int64
    L.tld_train_tensor_count.argtypes = [vp]
    L.tld_train_tensor_count.restype = C.c_int32
    L.tld_train_param_layout.argtypes = [vp, i32, C.c_char_p, i32, i64p, i64p]
    L.tld_train_set_angular_speeds.argtypes = [vp, C.POINTER(C.c_float), i32]
    L.tld_train_bind.argtypes = [vp, vp, vp]
    L.tld_train_refresh_weights.argtypes = [vp, vp]
    L.tld_train_forward_backward.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp]
    L.tld_train_forward_backward_cb.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, GRAD_READY_FN, vp]
    L.tld_train_adam_ema.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, f32, f32, f32, f32, i32, f32, f32, vp]
    L.tld_debug_attention_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.tld_debug_wgrad.argtypes = [vp, vp, vp, vp, C.c_int64, i32, i32, i32, vp]
    L.tld_debug_dwconv_gelu.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, i32, i32, i32, vp]
    L.tld_debug_attention_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, C.POINTER(C.c_float), vp]
    L.tld_train_destroy.argtypes = [vp]
    for name in ABI_SYMBOLS:
        if "TLD_LIB" in os.environ and not hasattr(L, name):     # an older A/B build: symbols added since are simply absent (tests/test_abi.py checks the real library)
            continue
        if name not in ("tld_last_error", "tld_engine_weight_bytes", "tld_vae_weight_bytes", "tld_clip_weight_bytes", "tld_train_param_count",
                        "tld_train_tensor_count"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().tld_last_error()
        raise RuntimeError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
