"""ctypes binding of the gfx950 C-ABI library (include/corrif.h) - host mirror of the kernels.

This is the ONLY way the product reaches the device kernels.  There is no CPU or PyTorch
fall-back: if `libcorrif_gfx950.so` is missing the import of the library raises, and every
entry point raises RuntimeError on a non-zero C-ABI status.

PyTorch is used for device memory (torch.empty), the current HIP stream and autograd plumbing only.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CORRIF_LIB") or os.path.join(_HERE, "libcorrif_gfx950.so")      # CORRIF_LIB: A/B builds side by side
_lib = None

i32, i64, u64, f32 = C.c_int32, C.c_int64, C.c_uint64, C.c_float
ptr = C.c_void_p


class Geom(C.Structure):
    _fields_ = [("is_gemm", i32), ("Rd", i32), ("Rh", i32), ("Rw", i32), ("Sd", i32), ("Sh", i32), ("Sw", i32),
                ("kd", i32), ("kh", i32), ("kw", i32), ("mul_d", i32), ("mul_h", i32), ("mul_w", i32),
                ("off_d", i32), ("off_h", i32), ("off_w", i32), ("div_d", i32), ("div_h", i32), ("div_w", i32),
                ("dir", i32), ("clamp", i32), ("ntaps", i32), ("src_batch_pitch", i64)]


class Gemm(C.Structure):
    _fields_ = [("A", ptr), ("lda", i64), ("Cs", i32), ("B", ptr), ("ldb", i64), ("b_layout", i32), ("C", ptr), ("ldc", i64),
                ("bias", ptr), ("addend", ptr), ("ld_add", i64), ("M", i32), ("N", i32), ("K", i32), ("act", i32),
                ("Z", i32), ("Zi", i32), ("sA_o", i64), ("sA_i", i64), ("sB_o", i64), ("sB_i", i64), ("sC_o", i64), ("sC_i", i64),
                ("g", Geom), ("ntap_sel", i32), ("tap_sel", C.c_int8 * 28),
                ("out_map", i32), ("OD", i32), ("OH", i32), ("OW", i32), ("om_d", i32), ("om_h", i32), ("om_w", i32),
                ("oo_d", i32), ("oo_h", i32), ("oo_w", i32),
                ("stats_part", ptr), ("stats_rows_per_group", i64), ("stats_relu", i32), ("ws", ptr), ("no_split", i32),
                ("addend2", ptr), ("ld_add2", i64),
                ("bstats_x", ptr), ("bstats_ldx", i64), ("bstats_y", ptr), ("bstats_ldy", i64), ("bstats_mean", ptr), ("bstats_rstd", ptr),
                ("zs_bias", i64), ("zs_add", i64), ("zs_add2", i64), ("zs_stats", i64), ("zs_bsx", i64), ("zs_bsy", i64), ("zs_bsstat", i64),
                ("f32_mfma", i32)]


class Wgrad(C.Structure):
    _fields_ = [("A", ptr), ("lda", i64), ("B", ptr), ("ldb", i64), ("Cs", i32), ("C", ptr), ("ldc", i64), ("ws", ptr),
                ("R", i32), ("M", i32), ("N", i32), ("splits", i32), ("Z", i32), ("Zi", i32),
                ("sA_o", i64), ("sA_i", i64), ("sB_o", i64), ("sB_i", i64), ("sC_o", i64), ("sC_i", i64), ("g", Geom), ("f32_mfma", i32)]


class Conv3Patch(C.Structure):
    _fields_ = [("X", ptr), ("ldx", i64), ("Wp", ptr), ("Y", ptr), ("ldy", i64), ("bias", ptr),
                ("B", i32), ("Sd", i32), ("Sh", i32), ("Sw", i32), ("Od", i32), ("Oh", i32), ("Ow", i32),
                ("Ci", i32), ("Co", i32), ("pad", i32), ("clamp", i32), ("cc", i32), ("fold", i32),
                ("stats_part", ptr), ("stats_chunks", i32), ("stats_relu", i32), ("add_src", ptr), ("ld_add", i64), ("add_Ds", i32)]


class Conv3PatchWgrad(C.Structure):
    _fields_ = [("X", ptr), ("ldx", i64), ("DY", ptr), ("lddy", i64), ("dW", ptr), ("ws", ptr),
                ("B", i32), ("Sd", i32), ("Sh", i32), ("Sw", i32), ("Od", i32), ("Oh", i32), ("Ow", i32),
                ("Ci", i32), ("Co", i32), ("pad", i32), ("clamp", i32)]


_SIGS = {
    "corrif_abi_version": (i32, []),
    "corrif_build_arch": (C.c_char_p, []),
    "corrif_gemm_fwd": (i32, [C.POINTER(Gemm), ptr]),
    "corrif_gemm_fwd_workspace": (C.c_size_t, [C.POINTER(Gemm)]),
    "corrif_gemm_fwd_is_split": (i32, [C.POINTER(Gemm)]),
    "corrif_wgrad_is_split": (i32, [C.POINTER(Wgrad)]),
    "corrif_wgrad": (i32, [C.POINTER(Wgrad), ptr]),
    "corrif_wgrad_workspace": (C.c_size_t, [C.POINTER(Wgrad)]),
    "corrif_wgrad_plan": (i32, [i32, i32, i32, i32]),
    "corrif_conv3_patch": (i32, [C.POINTER(Conv3Patch), ptr]),
    "corrif_conv3_patch_cc": (i32, [i32, i32]),
    "corrif_conv3_patch_stats_chunks": (i32, [i32, i32, i32, i32]),
    "corrif_conv3_patch_wgrad": (i32, [C.POINTER(Conv3PatchWgrad), ptr]),
    "corrif_conv3_patch_wgrad_workspace": (C.c_size_t, [i32, i32]),
    "corrif_conv3_patch_wgrad_slots": (i32, [i32, i32]),
    "corrif_conv1x1_small_supported": (i32, [i32, i32]),
    "corrif_conv1x1_small_fwd": (i32, [ptr, i64, ptr, i32, ptr, ptr, i64, i64, i32, i32, ptr]),
    "corrif_conv1x1_small_wgrad": (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i32, i32, ptr]),
    "corrif_conv1x1_small_workspace": (C.c_size_t, [i64, i32, i32]),
    "corrif_stem_supported": (i32, [i32] * 10),
    "corrif_stem_fwd": (i32, [ptr, i64, ptr, ptr, i64, i32, i32, i32, i32, ptr]),
    "corrif_stem_wgrad_workspace": (C.c_size_t, []),
    "corrif_stem_wgrad": (i32, [ptr, i64, ptr, i64, ptr, ptr, i32, i32, i32, i32, ptr]),
    "corrif_slab_reduce": (i32, [ptr, ptr, i64, i32, ptr]),
    "corrif_col_sum": (i32, [ptr, i64, i64, i32, ptr, ptr, ptr]),
    "corrif_col_sum_workspace": (C.c_size_t, [i64, i32]),
    "corrif_weight_repack": (i32, [ptr, ptr, i32, i32, i32, i32, i64, ptr]),
    "corrif_norm_stats": (i32, [ptr, i64, i64, i32, i32, i32, f32, ptr, ptr, ptr, ptr, f32, ptr, ptr]),
    "corrif_norm_stats_finalize": (i32, [ptr, i32, i32, i32, i64, f32, ptr, ptr, ptr, ptr, f32, ptr]),
    "corrif_norm_eval_rstd": (i32, [ptr, f32, ptr, i32, ptr]),
    "corrif_norm_apply": (i32, [ptr, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, i64, i64, i32, i32, i32, ptr]),
    "corrif_norm_bwd": (i32, [ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr, ptr, i64, i32, i32, i32, i32, ptr, ptr]),
    "corrif_norm_bwd_pre": (i32, [ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr, ptr, i64, i32, i32, ptr, i32, ptr, ptr]),
    "corrif_norm_workspace": (C.c_size_t, [i64, i32, i32]),
    "corrif_norm_workspace_g": (C.c_size_t, [i64, i32, i32]),
    "corrif_conv3_patch_stats_supported": (i32, [i32, i32]),
    "corrif_norm_stats_g": (i32, [ptr, i64, i64, i32, i32, i32, f32, ptr, ptr, ptr, ptr, f32, ptr, ptr]),
    "corrif_norm_stats_finalize_g": (i32, [ptr, i32, i32, i32, i64, f32, ptr, ptr, ptr, ptr, f32, ptr]),
    "corrif_norm_apply_g": (i32, [ptr, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, i64, i64, i32, i32, i32, i64, ptr]),
    "corrif_norm_bwd_g": (i32, [ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr, ptr, i64, i32, i32, i32, i32, i64, ptr, ptr]),
    "corrif_norm_bwd_pre_g": (i32, [ptr, i64, ptr, i64, ptr, i64, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr, ptr, i64, i32, i32, i32, i64, ptr, i32, ptr, ptr]),
    "corrif_col_sum_g": (i32, [ptr, i64, i64, i32, i32, ptr, ptr, ptr]),
    "corrif_stack_groups": (i32, [ptr, i32, ptr, i64, ptr]),
    "corrif_gather_multi": (i32, [ptr, ptr, ptr, i32, ptr, ptr]),
    "corrif_layernorm_fwd": (i32, [ptr, ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr, i64, i32, f32, ptr]),
    "corrif_layernorm_bwd": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i32, ptr]),
    "corrif_layernorm_workspace": (C.c_size_t, [i64, i32]),
    "corrif_maxpool133_fwd": (i32, [ptr, ptr, ptr, i32, i32, i32, i32, i32, ptr]),
    "corrif_maxpool133_bwd": (i32, [ptr, ptr, ptr, i32, i32, i32, i32, i32, ptr]),
    "corrif_trilinear_fwd": (i32, [ptr, i64, ptr, i64] + [i32] * 8 + [ptr]),
    "corrif_trilinear_bwd": (i32, [ptr, i64, ptr, i64] + [i32] * 8 + [ptr]),
    "corrif_trilinear_bwd_sep_workspace": (i64, [i32] * 8),
    "corrif_trilinear_bwd_sep": (i32, [ptr, ptr, ptr] + [i32] * 8 + [ptr]),
    "corrif_nearest_fwd": (i32, [ptr, i64, ptr, i64] + [i32] * 8 + [ptr]),
    "corrif_nearest_bwd": (i32, [ptr, i64, ptr, i64] + [i32] * 8 + [ptr]),
    "corrif_pad_fold": (i32, [ptr, ptr, i64, i32, i32, i32, i32, i32, ptr]),
    "corrif_softmax_rows": (i32, [ptr, i64, i32, f32, ptr]),
    "corrif_softmax_rows_bwd": (i32, [ptr, ptr, i64, i32, f32, ptr]),
    "corrif_softmax_dropout_rows": (i32, [ptr, ptr, i64, i32, f32, f32, u64, u64, ptr]),
    "corrif_softmax_dropout_rows_bwd": (i32, [ptr, ptr, i64, i32, f32, f32, u64, u64, ptr]),
    "corrif_flash_attn_supported": (i32, [i32, i32]),
    "corrif_flash_attn_mask_bytes": (C.c_size_t, [i32, i32, i32, f32]),
    "corrif_flash_attn_fwd": (i32, [ptr, ptr, ptr, ptr, i32, i32, i32, f32, f32, u64, u64, ptr]),
    "corrif_flash_attn_bwd": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, f32, f32, ptr]),
    "corrif_dropout": (i32, [ptr, ptr, i64, f32, u64, u64, ptr]),
    "corrif_depth_bcast_add": (i32, [ptr, i64, ptr, i64, i32, i32, i32, i32, i32, ptr]),
    "corrif_depth_class_reduce": (i32, [ptr, i64, ptr, i64, i32, i32, i32, i32, i32, ptr]),
    "corrif_add": (i32, [ptr, ptr, ptr, i64, ptr]),
    "corrif_add_bcast_rows": (i32, [ptr, ptr, i64, ptr, i64, ptr]),
    "corrif_gelu_fwd": (i32, [ptr, ptr, i64, ptr]),
    "corrif_gelu_bwd": (i32, [ptr, ptr, ptr, i64, ptr]),
    "corrif_relu_bwd": (i32, [ptr, ptr, ptr, i64, ptr]),
    "corrif_scale_dev": (i32, [ptr, ptr, ptr, i64, ptr]),
    "corrif_scale": (i32, [ptr, ptr, i64, f32, ptr]),
    "corrif_fill": (i32, [ptr, i64, f32, ptr]),
    "corrif_copy2d": (i32, [ptr, i64, ptr, i64, i64, i32, i32, ptr]),
    "corrif_sum_groups": (i32, [ptr, ptr, i64, i32, ptr]),
    "corrif_intercorr_fwd": (i32, [ptr, ptr, ptr, i64, ptr, ptr, ptr, i64, i32, i32, i32, ptr]),
    "corrif_intercorr_bwd": (i32, [ptr, ptr, ptr, i64, ptr, ptr, ptr, i64, ptr, ptr, ptr, i32, i32, i32, ptr]),
    "corrif_head_fwd": (i32, [ptr, ptr, ptr, ptr, i32, i32, ptr]),
    "corrif_head_bwd": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, ptr]),
    "corrif_head_workspace": (C.c_size_t, [i32, i32]),
    "corrif_bce_logits_mean": (i32, [ptr, ptr, i64, ptr, ptr, ptr, ptr]),
    "corrif_bce_workspace": (C.c_size_t, [i64]),
    "corrif_jaccard": (i32, [ptr, ptr, i64, f32, ptr, ptr, ptr]),
    "corrif_jaccard_workspace": (C.c_size_t, [i64]),
    "corrif_prep_means": (i32, [ptr, ptr, ptr, i32, i32, ptr, ptr, ptr]),
    "corrif_prep_stack": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, ptr]),
    "corrif_prep_workspace": (C.c_size_t, [i32, i32]),
    "corrif_adam_multi": (i32, [ptr, ptr, ptr, i32, f32, f32, f32, f32, f32, i32, ptr]),
    "corrif_adam_step": (i32, [ptr, ptr, ptr, ptr, i64, f32, f32, f32, f32, f32, i32, ptr]),
}
EXPORTS = tuple(_SIGS)
_ERR = {-1: "CORRIF_EINVAL (bad argument)", -2: "CORRIF_EUNSUPPORTED", -3: "CORRIF_ELAUNCH (hipGetLastError)"}


def lib():
    """Load (once) the in-tree HIP library; fail loudly when it is missing - there is no fall-back path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("corrif: %s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU / PyTorch fall-back for the MMVit4 hot path" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if l.corrif_abi_version() != 7:
            raise RuntimeError("corrif: ABI version mismatch")
        _lib = l
    return _lib


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(status, what):
    if status != 0:
        raise RuntimeError("corrif: %s failed with %s" % (what, _ERR.get(status, status)))


def P(t):
    """device address of a tensor as a plain int (0 = NULL); ints pass through"""
    if t is None:
        return 0
    return t if isinstance(t, int) else t.data_ptr()


def ws_bytes(n, device):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


def gemm_geom():
    g = Geom()
    g.is_gemm = 1
    g.dir = 1
    g.div_d = g.div_h = g.div_w = 1
    g.kd = g.kh = g.kw = 1
    g.ntaps = 1
    return g


def conv_geom(R, S, k, stride, pad, transposed=False, clamp=False, ntaps=None, src_batch_pitch=0):
    """R: rows grid (d,h,w); S: source grid; forward: R = output grid, S = input grid.
    transposed: R = input-gradient grid, S = output-gradient grid."""
    g = Geom()
    g.is_gemm = 0
    g.Rd, g.Rh, g.Rw = R
    g.Sd, g.Sh, g.Sw = S
    g.kd, g.kh, g.kw = k
    if not transposed:
        g.mul_d, g.mul_h, g.mul_w = stride
        g.off_d, g.off_h, g.off_w = (-pad[0], -pad[1], -pad[2])
        g.div_d = g.div_h = g.div_w = 1
        g.dir = 1
    else:
        g.mul_d = g.mul_h = g.mul_w = 1
        g.off_d, g.off_h, g.off_w = pad
        g.div_d, g.div_h, g.div_w = stride
        g.dir = -1
    g.clamp = 1 if clamp else 0
    g.ntaps = k[0] * k[1] * k[2] if ntaps is None else ntaps
    g.src_batch_pitch = src_batch_pitch
    return g
