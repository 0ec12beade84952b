"""ctypes binding of libpaintmind_hip.so (the C ABI declared in include/pmhip.h).

There is deliberately NO fallback: if the shared library is missing or a symbol cannot be resolved
the import of the compute path raises, and every op raises when handed a non-ROCm tensor.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpaintmind_hip.so")

PMHIP_OK = 0
F32, BF16 = 0, 1
PART_Q, PART_K, PART_V = 0, 1, 2
ABI_VERSION = 10

vp = C.c_void_p
i32 = C.c_int
f32 = C.c_float
u64 = C.c_uint64
u32 = C.c_uint32
i64 = C.c_int64


class LayerWeights(C.Structure):
    """mirror of pmhip_layer_weights"""
    _fields_ = [(n, vp) for n in (
        "ln1_g", "ln1_b", "wqkv", "wo", "bo", "lnx_g", "lnx_b", "wqkv2", "wo2", "bo2",
        "ln2_g", "ln2_b", "w12p", "b12p", "w3p", "b3",
        "wqkv_f", "qkv_c", "qkv_d", "wqkv2_f", "qkv2_c", "qkv2_d", "w12p_f", "w12_c", "w12_d")] + \
        [("bo_mean", C.c_float), ("bo2_mean", C.c_float), ("b3_mean", C.c_float)]


class TowerCfg(C.Structure):
    _fields_ = [("dim", i32), ("depth", i32), ("heads", i32), ("hidden_pad", i32), ("dim_head", i32)]


class VqganCfg(C.Structure):
    _fields_ = [("image_size", i32), ("patch_size", i32), ("channels", i32), ("n_embed", i32),
                ("embed_dim", i32), ("beta", f32), ("enc", TowerCfg), ("dec", TowerCfg)]


class VqganWeights(C.Structure):
    _fields_ = [("patch_w", vp), ("enc_pos", vp), ("pre_g", vp), ("pre_b", vp),
                ("enc_layers", C.POINTER(LayerWeights)), ("prevq_w", vp), ("prevq_b", vp),
                ("codebook_n", vp), ("codebook_sq", vp), ("postq_w", vp), ("postq_b", vp),
                ("dec_pos", vp), ("dec_layers", C.POINTER(LayerWeights)), ("dn_g", vp), ("dn_b", vp),
                ("proj_w", vp), ("proj_b", vp)]


class S2Cfg(C.Structure):
    _fields_ = [("tokens", i32), ("embed_dim", i32), ("n_embed", i32), ("context_dim", i32),
                ("context_dim_pad", i32), ("tower", TowerCfg)]


class S2Weights(C.Structure):
    _fields_ = [("tok_table", vp), ("tokproj_w", vp), ("tokproj_b", vp), ("pos", vp), ("ctxproj_w", vp),
                ("layers", C.POINTER(LayerWeights)), ("norm_g", vp), ("norm_b", vp), ("logits_w", vp),
                ("logits_b", vp), ("logits_wf", vp), ("logits_c", vp), ("logits_d", vp)]


class LnFold(C.Structure):
    """mirror of pmhip_lnfold"""
    _fields_ = [("coef", vp), ("c", vp), ("d", vp), ("parts", vp), ("nparts", i32), ("eps", f32)]


# name -> (restype, argtypes); every symbol include/pmhip.h declares
PROTOTYPES = {
    "pmhip_abi_version": (i32, []),
    "pmhip_last_error": (C.c_char_p, []),
    "pmhip_device_info": (i32, [i32, C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32]),
    "pmhip_gemm": (i32, [i32, vp, i32, vp, i32, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp]),
    "pmhip_gemm_swiglu": (i32, [i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "pmhip_gemm_heads": (i32, [i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32),
                               C.POINTER(vp), f32, vp]),
    "pmhip_gemm_hilo": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, vp]),
    "pmhip_gemm_hilo_stats": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp]),
    "pmhip_gemm_hilo_center": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, C.c_float, vp, i32, vp]),
    "pmhip_unshift_hilo": (i32, [vp, vp, vp, i32, i32, vp]),
    "pmhip_ln_coef_parts": (i32, [vp, i32, f32, vp, i32, vp]),
    "pmhip_layernorm_hilo": (i32, [vp, vp, vp, vp, f32, vp, i32, i32, i32, vp]),
    "pmhip_layernorm_to_hilo": (i32, [vp, vp, vp, f32, vp, vp, i32, i32, vp]),
    "pmhip_split_hilo": (i32, [vp, vp, vp, i32, i32, vp]),
    "pmhip_join_hilo": (i32, [vp, vp, vp, i32, i32, vp]),
    "pmhip_ln_coef": (i32, [vp, f32, vp, i32, i32, vp]),
    "pmhip_gemm_ln": (i32, [i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, C.POINTER(LnFold), vp]),
    "pmhip_gemm_softmax_stats": (i32, [i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, C.POINTER(LnFold), vp, vp]),
    "pmhip_gemm_swiglu_ln": (i32, [i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, C.POINTER(LnFold), vp]),
    "pmhip_gemm_heads_ln": (i32, [i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32),
                                  C.POINTER(vp), f32, C.POINTER(LnFold), vp]),
    "pmhip_lnfold_supported": (i32, [i32, i32, i32, i32, i32]),
    "pmhip_attention": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "pmhip_attention_fallbacks": (i32, [C.POINTER(C.c_ulonglong), i32]),
    "pmhip_gemm_heads_dh": (i32, [i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32),
                                  C.POINTER(vp), f32, vp, vp]),
    "pmhip_attention_dh": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "pmhip_layernorm": (i32, [vp, vp, vp, f32, vp, i32, i32, i32, vp]),
    "pmhip_patchify": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "pmhip_unpatchify_clamp": (i32, [vp, vp, i32, i32, i32, i32, i32, f32, f32, vp]),
    "pmhip_convert_pad": (i32, [vp, i32, vp, i32, i32, i32, vp]),
    "pmhip_add_rows": (i32, [vp, vp, i32, vp, i32, i32, vp]),
    "pmhip_guidance_combine": (i32, [vp, vp, C.c_float, vp, C.c_size_t, vp]),
    "pmhip_guidance_combine_stats": (i32, [vp, vp, C.c_float, vp, C.c_size_t, vp, vp]),
    "pmhip_embed_rows": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "pmhip_random_mask": (i32, [vp, vp, vp, i32, vp, vp, i32, i32, i32, vp]),
    "pmhip_masked_ce": (i32, [vp, i32, vp, vp, f32, vp, vp, i32, i32, vp]),
    "pmhip_vq_prepare": (i32, [vp, vp, vp, i32, i32, vp]),
    "pmhip_vq_scratch_bytes": (C.c_size_t, [i32, i32]),
    "pmhip_vq_quantize": (i32, [vp, vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, vp]),
    "pmhip_sample_rows": (i32, [vp, i32, vp, i64, i32, f32, vp, u64, u32, u64, vp, vp, vp, i32, i32, vp]),
    "pmhip_sample_rows_stats": (i32, [vp, i32, vp, vp, i64, i32, f32, vp, u64, u32, u64, vp, vp, vp, i32, i32, vp]),
    "pmhip_remask": (i32, [vp, vp, i32, i64, i32, i32, vp]),
    "pmhip_vqgan_create": (i32, [C.POINTER(vp), i32, i32, C.POINTER(VqganCfg), C.POINTER(VqganWeights)]),
    "pmhip_vqgan_destroy": (None, [vp]),
    "pmhip_vqgan_encode": (i32, [vp, vp, i32, vp, vp, vp, vp]),
    "pmhip_vqgan_decode": (i32, [vp, vp, i32, vp, vp]),
    "pmhip_vqgan_decode_indices": (i32, [vp, vp, i32, vp, vp]),
    "pmhip_vqgan_encoder_forward": (i32, [vp, vp, i32, vp, vp]),
    "pmhip_vqgan_decoder_forward": (i32, [vp, vp, i32, vp, vp]),
    "pmhip_s2_create": (i32, [C.POINTER(vp), i32, i32, C.POINTER(S2Cfg), C.POINTER(S2Weights)]),
    "pmhip_s2_destroy": (None, [vp]),
    "pmhip_s2_forward": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "pmhip_pipeline_sample": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, u64, u32, u64, vp, vp, vp, vp]),
    "pmhip_pipeline_generate": (i32, [vp, vp, vp, vp, i32, i32, i32, C.POINTER(f32), C.POINTER(i32),
                                      C.POINTER(C.c_ubyte), i32, u64, u64, vp, i32, vp, vp, C.c_size_t, vp]),
    "pmhip_pipeline_sample_guided": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, u64, u32, u64, vp, vp, vp, f32, vp]),
    "pmhip_pipeline_generate_guided": (i32, [vp, vp, vp, vp, i32, i32, i32, C.POINTER(f32), C.POINTER(i32),
                                             C.POINTER(C.c_ubyte), i32, u64, u64, vp, i32, vp, vp, C.c_size_t, vp, f32]),
    "pmhip_s2_switches": (i32, [vp]),
    "pmhip_vqgan_switches": (i32, [vp]),
    "pmhip_timing_enable": (i32, [i32]),
    "pmhip_timing_reset": (i32, []),
    "pmhip_timing_get": (i32, [C.c_char_p, C.POINTER(i32), C.POINTER(C.c_double)]),
}

_lib = None


class PmhipError(RuntimeError):
    pass


def load():
    """Load the library (once) and bind every prototype.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PmhipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` (or paintmind_amd/csrc/build.sh). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.pmhip_abi_version()
    if got != ABI_VERSION:
        raise PmhipError(f"libpaintmind_hip.so ABI version {got} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != PMHIP_OK:
        msg = load().pmhip_last_error()
        raise PmhipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
