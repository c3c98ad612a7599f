"""ctypes binding of libspann3r_hip.so (include/spann3r_hip.h).

This is the whole host<->device boundary: plain pointers, sizes and a stream handle.  There is no
CPU fallback anywhere in the product path: if the library is missing or a call fails, we raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SP3_LIB_PATH") or os.path.join(_HERE, "libspann3r_hip.so")   # override: kernel-variant experiments

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
EPI_PLAIN, EPI_ROPE_VT, EPI_PIXSHUF, EPI_PARTIAL = 0, 1, 2, 3
LOAD_PLAIN, LOAD_CONV3X3, LOAD_SOFTMAX = 0, 1, 2


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("A2", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
        ("res1", C.c_void_p), ("res2", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32),
        ("K1", C.c_int32),
        ("lda", C.c_int64), ("lda2", C.c_int64), ("ldw", C.c_int64), ("ldc", C.c_int64),
        ("ldr1", C.c_int64), ("ldr2", C.c_int64),
        ("strideA", C.c_int64), ("strideW", C.c_int64), ("strideC", C.c_int64),
        ("alpha", C.c_float), ("wdtype", C.c_int32), ("act", C.c_int32), ("out_bf16", C.c_int32),
        ("relu_in", C.c_int32), ("loader", C.c_int32),
        ("conv_H", C.c_int32), ("conv_W", C.c_int32), ("conv_C", C.c_int32),
        ("conv_OH", C.c_int32), ("conv_OW", C.c_int32), ("conv_stride", C.c_int32),
        ("epi", C.c_int32),
        ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("pos", C.c_void_p),
        ("rope_cols", C.c_int32), ("vt", C.c_void_p), ("tokens", C.c_int32), ("heads", C.c_int32),
        ("vt_ld", C.c_int64),
        ("ps_k", C.c_int32), ("ps_H", C.c_int32), ("ps_W", C.c_int32), ("ps_C", C.c_int32),
        ("tile", C.c_int32), ("a_bf16", C.c_int32), ("splitk", C.c_int32), ("a_packed", C.c_int32),
        ("ln_stats", C.c_void_p), ("ln_s", C.c_void_p), ("ln_nt", C.c_int32), ("ln_C", C.c_int32), ("ln_eps", C.c_float),
        ("stats_out", C.c_void_p), ("c2", C.c_void_p), ("qkv_packed", C.c_int32), ("out_packed", C.c_int32), ("w_packed", C.c_int32),
        ("sb_A2", C.c_int64), ("sb_bias", C.c_int64), ("sb_ln_stats", C.c_int64), ("sb_ln_s", C.c_int64),
        ("sb_stats_out", C.c_int64), ("sb_c2", C.c_int64), ("sb_vt", C.c_int64),
        ("trace", C.c_void_p),
        ("sm_stats_out", C.c_void_p), ("sm_stats", C.c_void_p), ("sm_nt", C.c_int32), ("sm_thresh", C.c_float), ("sm_zout", C.c_void_p),
        ("f32x3", C.c_int32), ("res_bf16", C.c_int32), ("dyn_n", C.c_void_p),
    ]


class HeadPart(C.Structure):
    """sp3_head_part (include/spann3r_hip.h)"""
    _fields_ = [
        ("src", C.c_void_p), ("s_b", C.c_int64), ("s_n", C.c_int64), ("s_h", C.c_int64),
        ("dst", C.c_void_p), ("d_b", C.c_int64), ("d_n", C.c_int64), ("d_h", C.c_int64),
        ("dstT", C.c_void_p), ("pos", C.c_void_p), ("N", C.c_int32), ("fwd", C.c_float), ("ldT", C.c_int64),
    ]


class AttnBwdDesc(C.Structure):
    """sp3_attn_bwd_desc (include/spann3r_hip.h)"""
    _fields_ = ([(n, C.c_void_p) for n in ("q", "k", "v", "dout")]
                + [(n, C.c_int64) for n in ("sq", "ldq", "sk", "ldk", "sv", "ldv", "sdo", "lddo")]
                + [(n, C.c_void_p) for n in ("qT", "kT", "doT")] + [("ldTq", C.c_int64), ("ldTk", C.c_int64)]
                + [("lse", C.c_void_p), ("D", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p)]
                + [(n, C.c_int64) for n in ("sdq", "lddq", "sdk", "lddk", "sdv", "lddv")]
                + [("B", C.c_int32), ("heads", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("scale", C.c_float), ("bf16_products", C.c_int32)])


class AttnQProjDesc(C.Structure):
    """sp3_attn_qproj_desc (include/spann3r_hip.h)"""
    _fields_ = [
        ("x_packed", C.c_void_p), ("x_group_stride", C.c_int64), ("ln_stats", C.c_void_p), ("stats_group_stride", C.c_int64),
        ("w_packed", C.c_void_p), ("w_group_stride", C.c_int64), ("ln_s", C.c_void_p), ("bias", C.c_void_p), ("vec_group_stride", C.c_int64),
        ("pos", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("ln_eps", C.c_float), ("D", C.c_int32),
        ("kp", C.c_void_p), ("k_cols", C.c_int32), ("k_col0", C.c_int32), ("npad_k", C.c_int32), ("vtp", C.c_void_p),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("out_bf16", C.c_int32), ("out_packed", C.c_int32),
        ("B", C.c_int32), ("heads", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("scale", C.c_float),
        ("o_group", C.c_int32), ("o_group_rows", C.c_int32),
    ]


class ReduceLnDesc(C.Structure):
    _fields_ = [
        ("partial", C.c_void_p), ("split_stride", C.c_int64), ("bias", C.c_void_p), ("res", C.c_void_p),
        ("ldres", C.c_int64), ("x_out", C.c_void_p), ("ldx", C.c_int64),
        ("g1", C.c_void_p), ("b1", C.c_void_p), ("out1", C.c_void_p), ("ld1", C.c_int64), ("out1_bf16", C.c_int32),
        ("g2", C.c_void_p), ("b2", C.c_void_p), ("out2", C.c_void_p), ("ld2", C.c_int64), ("out2_bf16", C.c_int32),
        ("eps", C.c_float), ("splits", C.c_int32), ("rows", C.c_int32), ("C", C.c_int32),
        ("out1_packed", C.c_int32), ("out2_packed", C.c_int32),
            ("act", C.c_int32), ("res2", C.c_void_p), ("ldres2", C.c_int64),
    ]


class BankWriteDesc(C.Structure):
    _fields_ = [
        ("feat_k", C.c_void_p), ("feat_v", C.c_void_p), ("k_raw", C.c_void_p), ("v_raw", C.c_void_p),
        ("k_hat", C.c_void_p), ("v_hat_t", C.c_void_p), ("s_bank", C.c_void_p), ("b_bank", C.c_void_p),
        ("gamma_k", C.c_void_p), ("beta_k", C.c_void_p), ("gamma_v", C.c_void_p), ("beta_v", C.c_void_p),
        ("gamma_q", C.c_void_p), ("beta_q", C.c_void_p),
        ("eps", C.c_float), ("alpha", C.c_float),
        ("M", C.c_int32), ("P", C.c_int32), ("C", C.c_int32), ("cap", C.c_int32), ("wdtype", C.c_int32), ("state", C.c_void_p),
    ]


_lib = None

_PROTOS = {
    "sp3_gemm": [C.POINTER(GemmDesc), C.c_void_p],
    "sp3_gemm2": [C.POINTER(GemmDesc), C.POINTER(GemmDesc), C.c_void_p],
    "sp3_gemm_plan": [C.POINTER(GemmDesc)],
    "sp3_layernorm": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_int,
                      C.c_int, C.c_int, C.c_void_p],
    "sp3_reduce_ln": [C.POINTER(ReduceLnDesc), C.c_void_p],
    "sp3_layernorm_dual": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                           C.c_int, C.c_int, C.c_void_p],
    "sp3_layernorm_packed": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int,
                             C.c_int, C.c_int, C.c_void_p],
    "sp3_attention_ex": [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                         C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                         C.c_void_p],
    "sp3_layernorm_t": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_int,
                        C.c_int, C.c_int, C.c_void_p],
    "sp3_rope_2d": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                    C.c_void_p, C.c_float, C.c_float, C.c_void_p],
    "sp3_attention": [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                      C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p],
    "sp3_attention_packed": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                             C.c_void_p],
    "sp3_attention_packed_qproj": [C.POINTER(AttnQProjDesc), C.c_void_p],
    "sp3_softmax_thresh": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float,
                           C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p],
    "sp3_softmax_pack": [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                         C.c_void_p],
    "sp3_colsum_packed": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "sp3_colsum_softmax": [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "sp3_bank_write": [C.POINTER(BankWriteDesc), C.c_void_p],
    "sp3_pack_stats": [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "sp3_gather_packed_rows": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_gather_packed_cols": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_colsum_accum": [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "sp3_cos_sim": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_mem_append": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "sp3_bank_state_set": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "sp3_prob_merge": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "sp3_colsum_prob": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_cos_sim_state": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_prune_select": [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p],
    "sp3_gather_rows": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_gather_cols": [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int,
                        C.c_int, C.c_void_p],
    "sp3_gather_1d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "sp3_im2col_patch": [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                         C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "sp3_conv3x3_tile": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_upsample2x": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_head_final": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                       C.c_void_p],
    "sp3_upsample2x_bf16": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_head_final_bf16": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p],
    "sp3_fill_f32": [C.c_void_p, C.c_float, C.c_int64, C.c_void_p],
    "sp3_cast_f32_to_bf16": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "sp3_copy2d_f32": [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p],
    "sp3_spin": [C.c_int64, C.c_void_p, C.c_void_p],
    "sp3_conf_loss_forward": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_conf_loss_backward": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_transpose": [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p],
    "sp3_transpose_batched": [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_transpose_pad": [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_softmax_bwd_pad": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p],
    "sp3_head_shuffle": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "sp3_attention_train_fwd": [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p],
    "sp3_attention_train_bwd": [C.c_void_p, C.c_void_p],
    "sp3_gelu": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "sp3_gelu_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "sp3_im2col3x3": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_col2im3x3": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_relu": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "sp3_relu_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "sp3_upsample2x_bwd": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "sp3_postprocess": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "sp3_postprocess_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "sp3_adamw": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                  C.c_float, C.c_void_p],
    "sp3_mul": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "sp3_pack_bf16": [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_pack_bf16_conv3x3": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_pack_bf16_act": [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "sp3_pack_bf16_colsum": [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "sp3_colsum_rows": [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "sp3_sumsq_partial": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    "sp3_clip_coef": [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_adamw_flat": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                       C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_softmax_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_void_p],
    "sp3_layernorm_bwd": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p],
    "sp3_focal_weiszfeld": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p],
    "sp3_conf_filter": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_copy_multi": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "sp3_pnp_dlt_accum": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                          C.c_void_p, C.c_void_p],
    "sp3_pnp_gn_accum": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p],
    "sp3_pnp_score": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p],
    "sp3_ssi_loss_forward": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                             C.c_void_p],
    "sp3_preprocess_image": [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                             C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.c_void_p, C.c_void_p],
}
EXPORTS = sorted(list(_PROTOS) + ["sp3_last_error", "sp3_version", "sp3_conf_loss_ws_bytes", "sp3_ssi_loss_ws_bytes", "sp3_sumsq_blocks", "sp3_colsum_rows_ws"])


def load():
    """dlopen the in-tree library; raises (never falls back) if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libspann3r_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the hot path)")
    lib = C.CDLL(LIB_PATH)
    lib.sp3_last_error.restype = C.c_char_p
    lib.sp3_last_error.argtypes = []
    lib.sp3_version.restype = C.c_int
    lib.sp3_conf_loss_ws_bytes.restype = C.c_int64
    lib.sp3_conf_loss_ws_bytes.argtypes = [C.c_int, C.c_int]
    lib.sp3_ssi_loss_ws_bytes.restype = C.c_int64
    lib.sp3_ssi_loss_ws_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.sp3_sumsq_blocks.restype = C.c_int64
    lib.sp3_sumsq_blocks.argtypes = [C.c_int64]
    lib.sp3_colsum_rows_ws.restype = C.c_int64
    lib.sp3_colsum_rows_ws.argtypes = [C.c_int, C.c_int]
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().sp3_last_error().decode(errors="replace")
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The caller's CURRENT HIP stream (also the capturing stream under torch.cuda.graph).  Every launch asks for it, so it goes
    through torch's raw C accessors where they exist: torch.cuda.current_stream() builds a Stream object and resolves the device
    index in Python -- 8 us per call, a third of the host time of an eagerly launched step (tools/cold_start.py --profile)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()
