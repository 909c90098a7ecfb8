"""ctypes binding of libunimp_hip.so (C ABI declared in include/unimp_hip.h).

There is NO fallback: if the shared library is missing, or a tensor is not on a HIP device, the call
raises.  PyTorch is used only to own device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunimp_hip.so")

c_p, c_i, c_l, c_f = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [("A", c_p), ("B", c_p), ("C", c_p), ("M", c_i), ("N", c_i), ("K", c_i),
                ("lda", c_l), ("ldb", c_l), ("ldc", c_l), ("a_kstrided", c_i), ("b_kstrided", c_i),
                ("bias", c_p), ("res", c_p), ("ldres", c_l), ("aux", c_p), ("ldaux", c_l),
                ("pre", c_p), ("ldpre", c_l), ("gate", c_p), ("alpha", c_f), ("act", c_i), ("dact", c_i),
                ("out_f32", c_i), ("accumulate", c_i), ("pre_deriv", c_i),
                ("rope_rot", c_i), ("rope_hd", c_i), ("rope_period", c_i), ("rope_span", c_i), ("rope_L", c_i), ("rope_log2_base", c_f),
                ("rope_pos", c_p), ("ln_gamma", c_p), ("ln_beta", c_p), ("ln_eps", c_f)]


class DecodeStepDesc(C.Structure):
    """include/unimp_hip.h unimp_decode_step_desc"""
    _fields_ = [("qkv", c_p), ("row_stride", c_l), ("head_stride", c_l), ("q_off", c_i), ("k_off", c_i), ("v_off", c_i),
                ("rows", c_i), ("heads", c_i), ("hd", c_i), ("rot", c_i), ("cos_rows", c_p), ("sin_rows", c_p),
                ("kcache", c_p), ("vcache", c_p), ("c_row_stride", c_l), ("c_slot_stride", c_l), ("c_head_stride", c_l), ("capacity", c_i),
                ("pos_idx", c_p), ("scale", C.c_float), ("alibi_slopes", c_p), ("out", c_p), ("o_row_stride", c_l), ("o_head_stride", c_l),
                ("workspace", c_p), ("arrived", c_p), ("group_mode", c_i), ("group", c_i), ("shared_len", c_p)]


class AttnDesc(C.Structure):
    _fields_ = [("q", c_p), ("k", c_p), ("v", c_p), ("o", c_p), ("lse", c_p)] + \
               [(n, c_l) for n in ("q_bs", "q_ss", "q_hs", "k_bs", "k_ss", "k_hs", "v_bs", "v_ss", "v_hs",
                                   "o_bs", "o_ss", "o_hs")] + \
               [("B", c_i), ("H", c_i), ("Sq", c_i), ("Sk", c_i), ("D", c_i), ("scale", c_f), ("mask_mode", c_i),
                ("kv_len", c_p), ("seg", c_p), ("seg_len", c_i),
                ("d_o", c_p), ("dq", c_p), ("dk", c_p), ("dv", c_p), ("delta", c_p)] + \
               [(n, c_l) for n in ("do_bs", "do_ss", "do_hs", "dq_bs", "dq_ss", "dq_hs", "dk_bs", "dk_ss", "dk_hs",
                                   "dv_bs", "dv_ss", "dv_hs")] + [("alibi_slopes", c_p), ("rope_cos", c_p), ("rope_sin", c_p), ("rope_half", c_i), ("rope_log2_base", c_f),
                                                    ("q_row_off", c_p), ("q_len", c_p), ("k_row_off", c_p), ("flags", c_i)]


class MxGemmDesc(C.Structure):
    _fields_ = [("A", c_p), ("B", c_p), ("scale_a", c_p), ("scale_b", c_p), ("C", c_p),
                ("bias", c_p), ("res", c_p), ("aux", c_p), ("pre", c_p)] + \
               [(n, c_l) for n in ("lda", "ldb", "ldsa", "ldsb", "ldc", "ldres", "ldaux", "ldpre")] + \
               [("M", c_i), ("N", c_i), ("K", c_i), ("act", c_i), ("deriv_u8", c_i), ("reserved0", c_i), ("scale_c", c_p), ("ldsc", c_l)]


# name -> argtypes (every entry point returns int status)
_SIGS = {
    "unimp_gemm_bf16": [C.POINTER(GemmDesc), c_p],
    "unimp_gemm_bf16_variant": [C.POINTER(GemmDesc), c_i, c_p],
    "unimp_gemm_bf16_splitk": [C.POINTER(GemmDesc), c_i, c_p, c_p],
    "unimp_gemm_set_skinny2": [c_i],
    "unimp_gemm_skinny_ln_ok": [c_i, c_i],
    "unimp_layernorm_fwd": [c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_p],
    "unimp_layernorm_fwd_mx": [c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_f, c_i, c_p],
    "unimp_layernorm_bwd": [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_i,
                            c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "unimp_rope_halfsplit": [c_p, c_l, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p],
    "unimp_rope_halfsplit_pos": [c_p, c_l, c_l, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p],
    "unimp_decode_rope_append": [c_p, c_l, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_p, c_p],
    "unimp_kv_reorder_beams": [c_p, c_l, c_i, c_l, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p],
    "unimp_attn_decode_step": [C.POINTER(DecodeStepDesc), c_p],
    "unimp_attn_decode_step_slots": [c_i, c_i, c_i, c_i],
    "unimp_attn_fwd": [C.POINTER(AttnDesc), c_p],
    "unimp_attn_bwd": [C.POINTER(AttnDesc), c_p],
    "unimp_attn_decode": [C.POINTER(AttnDesc), c_p, c_i, c_p],
    "unimp_attn_decode_grouped": [C.POINTER(AttnDesc), c_p, c_i, c_i, c_p, c_p],
    "unimp_attn_decode_splits": [c_i, c_i, c_i],
    "unimp_embedding_fwd": [c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p],
    "unimp_embedding_bwd": [c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p],
    "unimp_embedding_bwd_sorted": [c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_i, c_i, c_i, c_p],
    "unimp_embedding_bwd_sorted_scratch": [c_i, c_i],
    "unimp_vit_patchify": [c_p, c_i, c_p, c_l, c_i, c_i, c_i, c_i, c_p],
    "unimp_vit_assemble": [c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "unimp_marker": [c_i, c_p],
    "unimp_gather_rows": [c_p, c_l, c_p, c_p, c_l, c_i, c_i, c_p],
    "unimp_add_bf16": [c_p, c_p, c_p, c_l, c_p],
    "unimp_cast_f32_to_bf16": [c_p, c_p, c_l, c_f, c_p],
    "unimp_swiglu_fwd": [c_p, c_l, c_p, c_l, c_i, c_i, c_p],
    "unimp_swiglu_bwd": [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_p],
    "unimp_dot_bf16": [c_p, c_p, c_l, c_p, c_p],
    "unimp_prefetch": [c_p, c_l, c_i, c_p, c_p],
    "unimp_beam_topk": [c_p, c_i, c_l, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "unimp_beam_topk_scratch": [c_i],
    "unimp_gemm_skinny_rows": [c_i, c_i, c_p],
    "unimp_bcast_rows": [c_p, c_p, c_l, c_i, c_i, c_i, c_p],
    "unimp_reduce_rows_periodic": [c_p, c_l, c_p, c_i, c_i, c_i, c_p],
    "unimp_label_mask": [c_p, c_p, c_p, c_i, c_i, c_l, c_l, c_l, c_l, c_p],
    "unimp_attn_set_generation": [c_i],
    "unimp_attn_get_generation": [],
    "unimp_attn_set_vit_tail": [c_i],
    "unimp_attn_set_dkv3": [c_i],
    "unimp_attn_last_dkv": [],
    "unimp_pack_b_bf16": [c_p, c_l, c_i, c_i, c_i, c_p, c_p],
    "unimp_mx_quantize": [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_p],
    "unimp_gemm_mxfp8": [C.POINTER(MxGemmDesc), c_p],
    "unimp_focal_ce_fwd": [c_p, c_l, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "unimp_focal_ce_bwd": [c_p, c_l, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "unimp_focal_ce_bwd_rows": [c_p, c_l, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_i, c_p],
    "unimp_sumsq_bf16": [c_p, c_l, c_p, c_p],
    "unimp_adamw_flat": [c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_f, c_f, c_f, c_f, c_f, c_i, c_p, c_f, c_f, c_i, c_p],
    "unimp_image_resize_normalize": [c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_p],
}

ABI_VERSION = 8          # must equal UNIMP_ABI_VERSION of include/unimp_hip.h the library was built from

_lib = None


class UnimpHipError(RuntimeError):
    pass


def lib():
    """Load libunimp_hip.so once.  Fails loudly -- there is no CPU or eager fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C unimp_amd/csrc`).")
        import torch  # noqa: F401  -- load PyTorch's bundled HIP runtime FIRST so libunimp_hip.so binds to the same
        #                          libamdhip64 (a second runtime from /opt/rocm would see no device and other streams)
        L = C.CDLL(LIB_PATH)
        L.unimp_last_error.restype = C.c_char_p
        L.unimp_abi_version.restype = c_i
        L.unimp_pack_b_bytes.restype, L.unimp_pack_b_bytes.argtypes = c_l, [c_i, c_i]
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes, fn.restype = args, c_i
        L.unimp_embedding_bwd_sorted_scratch.restype = c_l
        L.unimp_beam_topk_scratch.restype = c_l
        if L.unimp_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {L.unimp_abi_version()}, this package binds version {ABI_VERSION} -- a stale build; "
                              "rebuild with `make -C unimp_amd/csrc`")
        # the descriptors are passed by pointer: a layout the library and these ctypes mirrors disagree on would bind silently
        L.unimp_struct_size.restype, L.unimp_struct_size.argtypes = c_i, [c_i]
        for which, st in ((0, GemmDesc), (1, AttnDesc), (3, MxGemmDesc), (4, DecodeStepDesc)):
            if L.unimp_struct_size(which) != C.sizeof(st):
                raise ImportError(f"{LIB_PATH}: sizeof descriptor {which} is {L.unimp_struct_size(which)} in the library, "
                                  f"{C.sizeof(st)} in {st.__name__} -- stale build")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise UnimpHipError(f"{what} failed (code {rc}): {lib().unimp_last_error().decode()}")


def declared_symbols():
    return ["unimp_abi_version", "unimp_struct_size", "unimp_last_error", "unimp_set_error", "unimp_check_launch", "unimp_pack_b_bytes"] + list(_SIGS)
