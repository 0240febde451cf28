"""ctypes binding of include/aitk_mi355.h.  The product path has NO fallback: if the HIP library is missing
or a call returns non-zero, a RuntimeError is raised."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AITK_LIB_PATH", os.path.join(_HERE, "libaitk_mi355.so"))  # override only for A/B of two builds
_lib = None

vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", vp), ("lda", i64), ("a_seg_rows", i32), ("_pad0", i32), ("a_seg_stride", i64),
        ("B", vp), ("ldb", i64),
        ("A2", vp), ("lda2", i64),
        ("B2", vp), ("ldb2", i64),
        ("C", vp), ("ldc", i64), ("c_seg_rows", i32), ("_pad1", i32), ("c_seg_stride", i64),
        ("bias", vp),
        ("aux_out", vp), ("ld_aux_out", i64),
        ("aux_in", vp), ("ld_aux_in", i64),
        ("gate", vp), ("ld_gate", i64), ("gate_rows", i32),
        ("M", i32), ("N", i32), ("K", i32), ("K2", i32),
        ("flags", i32), ("stage_mode", i32), ("tile_mode", i32), ("conv_mode", i32),
        ("conv_H", i32), ("conv_W", i32), ("conv_Cin", i32), ("conv_Wo", i32), ("conv_HoWo", i32), ("conv_stride", i32),
        ("conv_pad_t", i32), ("conv_pad_l", i32), ("conv_t3d", i32),
        ("zero_page", vp),
        ("b_scale", vp), ("b_scale_mode", i32), ("_pad4", i32),
        ("col_scale", vp),
        ("a_scale", vp),
        ("t_partial", vp), ("t_p", vp), ("t_p_lo", vp), ("t_ldp", i64), ("t_tile0", i32), ("t_rank", i32),
    ]


class QuantRowsArgs(C.Structure):
    _fields_ = [("X", vp), ("ldx", i64), ("seg_rows", i32), ("_pad0", i32), ("seg_stride", i64), ("col_mul", vp),
                ("Q", vp), ("ldq", i64), ("row_scale", vp), ("M", i32), ("K", i32)]


class WgradSrc2(C.Structure):
    _fields_ = [("G2", vp), ("ldg2", i64), ("split_col", i32), ("act", i32)]


class LoraDownArgs(C.Structure):
    _fields_ = [
        ("X", vp), ("ldx", i64), ("x_seg_rows", i32), ("_pad0", i32), ("x_seg_stride", i64),
        ("P", vp), ("ldp", i64),
        ("T", vp), ("ldt", i64),
        ("mult", vp), ("scale", C.c_float), ("rows_per_batch", i32),
        ("M", i32), ("K", i32), ("R", i32), ("split_rp", i32),
        ("P_lo", vp),
        ("tmask", vp), ("tmask_rows_per_batch", i32), ("_pad1", i32),
    ]


class LoraWgradArgs(C.Structure):
    _fields_ = [
        ("S", vp), ("lds", i64),
        ("G", vp), ("ldg", i64), ("g_seg_rows", i32), ("split_rp", i32), ("g_seg_stride", i64),
        ("partial", vp),
        ("out", vp), ("out_stride_r", i64), ("out_stride_l", i64),
        ("accumulate", i32),
        ("M", i32), ("R", i32), ("L", i32),
    ]


class LnModArgs(C.Structure):
    _fields_ = [
        ("x", vp), ("ldx", i64),
        ("shift", vp), ("scale", vp), ("ld_mod", i64),
        ("out", vp), ("ld_out", i64),
        ("mean", vp), ("rstd", vp),
        ("eps", C.c_float), ("rows_per_batch", i32), ("M", i32), ("C", i32),
    ]


class LnModBwdArgs(C.Structure):
    _fields_ = [
        ("dxn", vp), ("ld_dxn", i64),
        ("x", vp), ("ldx", i64),
        ("mean", vp), ("rstd", vp),
        ("scale", vp), ("ld_mod", i64),
        ("dres", vp), ("ld_dres", i64),
        ("dx", vp), ("ld_dx", i64),
        ("partial", vp),
        ("S", i32), ("B", i32), ("C", i32), ("_pad", i32),
    ]


class GateBwdArgs(C.Structure):
    _fields_ = [
        ("dx", vp), ("ld_dx", i64),
        ("y", vp), ("ld_y", i64),
        ("gate", vp), ("ld_gate", i64),
        ("dy", vp), ("ld_dy", i64),
        ("partial", vp),
        ("S", i32), ("B", i32), ("C", i32), ("_pad", i32),
    ]


class ColsumFinishArgs(C.Structure):
    _fields_ = [
        ("partial", vp),
        ("out0", vp), ("out1", vp), ("ld_out", i64),
        ("B", i32), ("nchunk", i32), ("V", i32), ("C", i32),
    ]


class QkvJob(C.Structure):
    _fields_ = [("src", vp), ("ld_src", i64), ("dst", vp), ("ld_dst", i64), ("weight", vp), ("raw", vp), ("ld_raw", i64)]


class QkvPostArgs(C.Structure):
    _fields_ = [
        ("job", QkvJob * 3),
        ("cos", vp), ("sin", vp),
        ("eps", C.c_float), ("njobs", i32),
        ("B", i32), ("H", i32), ("D", i32), ("S_src", i32), ("S_dst", i32), ("s_off", i32),
    ]


class EwArgs(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("a", vp), ("lda", i64), ("y", vp), ("ldy", i64),
                ("rows", i32), ("C", i32), ("op", i32), ("alpha", C.c_float), ("a_rows_per_batch", i32), ("_pad", i32)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("Q", vp), ("K", vp), ("V", vp), ("ldq", i64), ("ldk", i64), ("ldv", i64),
        ("O", vp), ("ldo", i64),
        ("LSE", vp),
        ("dO", vp), ("lddo", i64),
        ("dQ", vp), ("dK", vp), ("dV", vp), ("lddq", i64), ("lddk", i64), ("lddv", i64),
        ("delta", vp),
        ("scale", C.c_float), ("B", i32), ("H", i32), ("S", i32), ("D", i32), ("Skv", i32), ("Dv", i32), ("hstride", i32),
        ("dS", vp), ("ds_mode", i32), ("_pad_ds", i32),
    ]


class GemvArgs(C.Structure):
    _fields_ = [
        ("X", vp), ("ldx", i64), ("W", vp), ("ldw", i64), ("bias", vp),
        ("T", vp), ("ldt", i64), ("Bl", vp), ("ldbl", i64),
        ("out", vp), ("ldo", i64),
        ("Bm", i32), ("N", i32), ("K", i32), ("R", i32), ("accumulate", i32), ("cols_per_group", i32),
        ("col_scale", vp),
    ]


class NoisePackArgs(C.Structure):
    _fields_ = [("latents", vp), ("noise", vp), ("t", vp), ("noisy", vp), ("target", vp),
                ("B", i32), ("C", i32), ("H", i32), ("W", i32)]


class MseArgs(C.Structure):
    _fields_ = [("pred", vp), ("target", vp), ("weight", vp), ("dpred", vp), ("partial", vp),
                ("loss_per_sample", vp), ("loss", vp), ("n_per_sample", i64), ("B", i32), ("feat", i32), ("mask", vp),
                ("loss_type", i32), ("huber_c", C.c_float), ("max_loss", C.c_float), ("_pad_guard", i32), ("guard", vp)]


class AdamWArgs(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("ema", vp),
                ("norm_partial", vp), ("norm_partial2", vp), ("norm_out", vp), ("n", i64)] + [
        (k, C.c_float) for k in ("lr", "beta1", "beta2", "eps", "weight_decay", "bias_correction1",
                                 "bias_correction2_sqrt", "max_norm", "ema_decay", "grad_scale", "ema_feedback", "param_multiplier")] + [
        ("guard", vp), ("n_micro", i32), ("_pad_micro", i32), ("beta1_d", C.c_double), ("beta2_d", C.c_double)]


class GroupNormArgs(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("gamma", vp), ("beta", vp), ("partial", vp), ("stats", vp),
                ("eps", C.c_float), ("silu", i32), ("B", i32), ("HW", i32), ("C", i32), ("G", i32), ("stats_out", vp)]


class GroupNormBwdArgs(C.Structure):
    _fields_ = [("dy", vp), ("ld_dy", i64), ("x", vp), ("ldx", i64), ("gamma", vp), ("beta", vp), ("stats", vp),
                ("dres", vp), ("ld_dres", i64), ("dx", vp), ("ld_dx", i64), ("partial", vp), ("red", vp),
                ("silu", i32), ("B", i32), ("HW", i32), ("C", i32), ("G", i32), ("_pad", i32)]


class DdpmNoiseArgs(C.Structure):
    _fields_ = [("latents", vp), ("noise", vp), ("a", vp), ("s", vp), ("noisy", vp), ("target", vp),
                ("B", i32), ("C", i32), ("HW", i32), ("Cp", i32), ("mode", i32), ("_pad", i32)]


class RmsFullArgs(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("g", vp), ("ldg", i64), ("y", vp), ("ldy", i64), ("weight", vp), ("cos", vp), ("sin", vp),
                ("eps", C.c_float), ("S", i32), ("M", i64), ("C", i32), ("_pad", i32)]


class DoraColscaleArgs(C.Structure):
    _fields_ = [("w2", vp), ("tw", vp), ("ldtw", i64), ("up", vp), ("gram", vp), ("mag", vp), ("c", vp),
                ("s", C.c_float), ("N", i32), ("R", i32), ("_pad", i32)]


class DoraBwdArgs(C.Structure):
    _fields_ = [("dy", vp), ("ld_dy", i64), ("y", vp), ("ld_y", i64), ("c", vp), ("bias", vp), ("mag", vp),
                ("dz", vp), ("ld_dz", i64), ("dmag", vp), ("partial", vp), ("M", i32), ("N", i32)]


class KronApplyArgs(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("x_seg_stride", i64), ("A", vp), ("B", vp), ("out", vp), ("ldo", i64),
                ("out_seg_stride", i64), ("x_seg_rows", i32), ("out_seg_rows", i32), ("M", i32), ("a_in", i32), ("b_in", i32),
                ("a_out", i32), ("b_out", i32), ("transpose_out", i32), ("accumulate", i32), ("col0", i32), ("ncols", i32),
                ("scale", C.c_float)]


class ShadowDesc(C.Structure):
    _fields_ = [("src_off", i64), ("d0", i64), ("d1", i64), ("d2", i64), ("rows", i32), ("cols", i32), ("kind", i32), ("aux", i32)]


EPI_BIAS, EPI_ACCUM, EPI_GELU, EPI_DGELU, EPI_GATE_RES, EPI_BIAS_ROW, EPI_ADD_AUX, EPI_COL_SCALE, EPI_SPLIT_SLAB, EPI_EMIT_T = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512

_STRUCTS = {0: GemmArgs, 1: LoraDownArgs, 2: LoraWgradArgs, 3: LnModArgs, 4: LnModBwdArgs, 5: GateBwdArgs,
            6: ColsumFinishArgs, 7: QkvPostArgs, 8: EwArgs, 9: AttnArgs, 10: GemvArgs, 11: NoisePackArgs,
            12: MseArgs, 13: AdamWArgs, 14: ShadowDesc, 15: GroupNormArgs, 16: RmsFullArgs, 17: DoraColscaleArgs, 18: DoraBwdArgs,
            19: KronApplyArgs, 20: GroupNormBwdArgs, 21: DdpmNoiseArgs, 22: QuantRowsArgs, 23: WgradSrc2}


ABI_VERSION = 12  # AITK_ABI_VERSION of include/aitk_mi355.h this mirror was written against


def lib():
    """Load libaitk_mi355.so (import torch first so its bundled HIP runtime is the one in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MI355X path has no CPU/PyTorch fallback)"
        )
    import torch  # noqa: F401  (loads libamdhip64 from torch/lib before our library resolves it)

    L = C.CDLL(LIB_PATH)
    L.aitk_abi_version.restype = C.c_int
    if L.aitk_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} is ABI version {L.aitk_abi_version()}, this package was written against {ABI_VERSION}: rebuild the library")
    L.aitk_sizeof.restype = C.c_int
    L.aitk_sizeof.argtypes = [i32]
    for which, st in _STRUCTS.items():
        n = L.aitk_sizeof(which)
        if n != C.sizeof(st):
            raise RuntimeError(f"ABI mismatch: struct {st.__name__} is {C.sizeof(st)} B in Python, {n} B in C")
    L.aitk_lora_wgrad_workspace_bytes.restype = C.c_int64
    L.aitk_lora_wgrad_workspace_bytes.argtypes = [i32, i32, i32]
    L.aitk_lora_bwd_fused_workspace_bytes.restype = C.c_int64
    L.aitk_lora_bwd_fused_workspace_bytes.argtypes = [i32, i32, i32]
    L.aitk_lora_bwd_fused.argtypes = [vp, vp, vp, vp]
    L.aitk_rows_per_block.restype = i32
    L.aitk_mse_workspace_bytes.restype = C.c_int64
    L.aitk_mse_workspace_bytes.argtypes = [i32, i64]
    L.aitk_ema_update.argtypes = [vp, vp, i64, C.c_float, C.c_float, C.c_float, vp]
    L.aitk_adamw_workspace_bytes.restype = C.c_int64
    L.aitk_adamw_workspace_bytes.argtypes = [i64]
    L.aitk_lora_refresh_shadows.argtypes = [vp, vp, vp, i32, vp]
    L.aitk_slab_rescale.argtypes = [vp, i64, i32, i32, vp, i32, vp, i32, vp]
    L.aitk_gemm_nt_grouped.argtypes = [vp, vp, vp]
    L.aitk_lora_down_raw.argtypes = [vp, vp, vp]
    L.aitk_lora_t_finish.argtypes = [vp, vp, i32, vp]
    L.aitk_lora_down_ksplit_workspace_bytes.restype = C.c_int64
    L.aitk_lora_down_ksplit_workspace_bytes.argtypes = [i32, i32, i32]
    L.aitk_lora_down_ksplit.argtypes = [vp, vp, i32, vp]
    L.aitk_lora_wgrad_main.argtypes = [vp, vp, vp]
    L.aitk_lora_wgrad_finish_multi.argtypes = [vp, i32, vp]
    L.aitk_lokr_lowrank_grad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.aitk_grad_compress_bf16.argtypes = [vp, vp, i64, vp]
    L.aitk_grad_expand_bf16.argtypes = [vp, vp, i64, vp]
    L.aitk_groupnorm_workspace_bytes.restype = C.c_int64
    L.aitk_groupnorm_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.aitk_softmax_rows.argtypes = [vp, i64, i32, i32, C.c_float, vp]
    L.aitk_kron_merge.argtypes = [vp, i64, vp, vp, i32, i32, i32, i32, C.c_float, vp]
    L.aitk_image_to_nhwc8.argtypes = [vp, vp, i32, i32, i32, vp]
    L.aitk_image_resize_to_nhwc8.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.aitk_latent_sample.argtypes = [vp, i64, vp, vp, i32, i32, i32, C.c_float, C.c_float, vp]
    L.aitk_latent_sample_affine.argtypes = [vp, i64, vp, vp, i32, i32, i32, vp, vp, vp]
    L.aitk_rmsnorm_rows.argtypes = [vp, i64, vp, i64, vp, i64, i32, C.c_float, i32, vp]
    L.aitk_timestep_embed.argtypes = [vp, vp, i32, i32, C.c_float, vp]
    L.aitk_copy2d.argtypes = [vp, i64, vp, i64, i64, i64, vp]
    L.aitk_groupnorm_bwd_workspace_bytes.restype = C.c_int64
    L.aitk_groupnorm_bwd_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.aitk_geglu_fwd.argtypes = [vp, i64, vp, i64, i64, i32, vp]
    L.aitk_geglu_bwd.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, vp]
    L.aitk_resample2x.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.aitk_pad_nhwc.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.aitk_copy_heads.argtypes = [vp, i64, vp, i64, i64, i32, i32, i32, vp]
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc} ({'AITK_ERR' if rc < 0 else 'hipError'})")


def stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
