"""ctypes binding of ``libtacotron2_amd.so`` (C ABI: ``include/tacotron2_amd.h``).

The engine has exactly one compute path: the HIP kernels behind this module.
There is no CPU or eager-PyTorch fallback — if the library is missing or a
tensor is not a device tensor the call fails loudly.  PyTorch is used for
device memory and streams only: every wrapper takes ``torch`` tensors, checks
dtype/layout, and hands raw device pointers plus the current HIP stream to the
library.

Build the library with ``python -m tacotron2_amd.build`` (hipcc, gfx950).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2AMD_LIB", os.path.join(_HERE, "lib", "libtacotron2_amd.so"))

TORCH_OPS_PATH = os.path.join(_HERE, "lib", "libtacotron2_amd_torch.so")

ATT_DIM = 128
LOC_FILTERS = 32
LOC_KERNEL = 31
LOC_TAPS = 62
ATT_SLICES = 4

_f32p = C.c_void_p
_i64 = C.c_longlong


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", _f32p), ("B", _f32p), ("C", _f32p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", _i64), ("ldb", _i64), ("ldc", _i64),
        ("a_kcontig", C.c_int), ("b_kcontig", C.c_int),
        ("batch", C.c_int),
        ("strideA", _i64), ("strideB", _i64), ("strideC", _i64),
        ("splitk", C.c_int), ("strideSplitC", _i64),
        ("accumulate", C.c_int),
        ("bias", _f32p), ("act", C.c_int),
        ("keep", C.c_void_p), ("ldkeep", _i64), ("keep_scale", C.c_float),
        ("convA_T", C.c_int), ("convA_C", C.c_int), ("convA_pad", C.c_int), ("convA_sign", C.c_int),
        ("convB_T", C.c_int), ("convB_C", C.c_int), ("convB_pad", C.c_int),
        ("precision", C.c_int),
    ]


class Gemm16Desc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", _f32p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", _i64), ("ldb", _i64), ("ldc", _i64),
        ("splitk", C.c_int), ("strideSplitC", _i64), ("accumulate", C.c_int), ("bias", _f32p),
        ("win_T", C.c_int), ("win_Tp", C.c_int),
    ]


class Seg(C.Structure):
    _fields_ = [("p", _f32p), ("ld", _i64), ("width", C.c_int)]


class LstmStep(C.Structure):
    _fields_ = [
        ("x", Seg * 3), ("nseg", C.c_int),
        ("W", _f32p), ("Ktot", C.c_int), ("H", C.c_int), ("B", C.c_int),
        ("gin", _f32p), ("ld_gin", _i64),
        ("bias", _f32p),
        ("c_prev", _f32p), ("ld_cprev", _i64),
        ("gates_out", _f32p), ("ld_gates", _i64),
        ("c_out", _f32p), ("ld_c", _i64),
        ("h_out", _f32p), ("ld_h", _i64),
        ("keep", C.c_void_p), ("ld_keep", _i64), ("keep_scale", C.c_float),
        ("lens", C.c_void_p), ("t", C.c_int), ("tag", C.c_int),
        ("bf16", C.c_int), ("h16_out", C.c_void_p), ("ld_h16", _i64),
    ]


class SkinnyGemm(C.Structure):
    _fields_ = [
        ("x", Seg * 3), ("nseg", C.c_int),
        ("W", _f32p), ("Ktot", C.c_int), ("N", C.c_int), ("B", C.c_int),
        ("Y", _f32p), ("ldy", _i64), ("nsplit", C.c_int), ("split_stride", _i64), ("tag", C.c_int),
        ("bf16", C.c_int),
        ("bias", _f32p), ("act", C.c_int), ("keep", C.c_void_p), ("ld_keep", _i64), ("keep_scale", C.c_float),
        ("Y16", C.c_void_p), ("ldy16", _i64),
        ("stop_active", C.c_void_p), ("stop_lengths", C.c_void_p), ("stop_done", C.c_void_p),
        ("stop_col", C.c_int), ("stop_t", C.c_int), ("stop_max_steps", C.c_int), ("stop_threshold", C.c_float),
    ]


class Addend(C.Structure):
    _fields_ = [("p", _f32p), ("ld", _i64), ("nsplit", C.c_int), ("split_stride", _i64)]


class LstmBwd(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("H", C.c_int),
        ("dh", Addend * 3),
        ("gates", _f32p), ("ld_gates", _i64),
        ("c_prev", _f32p), ("ld_cprev", _i64),
        ("c", _f32p), ("ld_c", _i64),
        ("keep", C.c_void_p), ("ld_keep", _i64), ("keep_scale", C.c_float),
        ("dc", _f32p), ("ld_dc", _i64),
        ("dgates", _f32p), ("ld_dgates", _i64),
        ("lens", C.c_void_p), ("t", C.c_int),
        ("dgates16", C.c_void_p), ("ld_dgates16", _i64),
        ("dgates16_x3", C.c_int),
    ]


class AttnFwd(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("Ti", C.c_int), ("E", C.c_int), ("Hq", C.c_int),
        ("h", _f32p), ("ld_h", _i64),
        ("Wq", _f32p), ("U", _f32p), ("v", _f32p), ("pm", _f32p), ("memory", _f32p),
        ("lens", C.c_void_p),
        ("w_prev", _f32p), ("ld_wprev", _i64),
        ("cum", _f32p), ("cum_save", _f32p),
        ("w_out", _f32p), ("ld_wout", _i64),
        ("ctx_out", _f32p), ("ld_ctx", _i64),
        ("q_out", _f32p), ("ld_q", _i64),
        ("active", C.c_void_p),
        ("ws", _f32p),
        ("ctx16_out", C.c_void_p), ("ld_ctx16", _i64),
        ("loc_split_bf16", C.c_int),
        ("memory16", C.c_void_p),
        ("Wq16", C.c_void_p),
        ("ws_floats", _i64),
        ("ctx16_x3", C.c_int),
    ]


class AttnBwd(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("Ti", C.c_int), ("E", C.c_int), ("Hq", C.c_int),
        ("dctx", Addend * 3),
        ("dctx_total", _f32p), ("ld_dctx_total", _i64),
        ("d_w_extra", _f32p), ("ld_dwextra", _i64),
        ("q", _f32p), ("ld_q", _i64),
        ("Wq", _f32p), ("U", _f32p), ("v", _f32p), ("pm", _f32p), ("memory", _f32p),
        ("lens", C.c_void_p),
        ("w", _f32p), ("ld_w", _i64),
        ("w_prev", _f32p), ("ld_wprev", _i64),
        ("cum_before", _f32p),
        ("dwin_part", _f32p), ("dcum_acc", _f32p),
        ("d_pm", _f32p), ("dU_acc", _f32p), ("dv_acc", _f32p),
        ("dq_out", _f32p), ("ld_dq", _i64),
        ("dh_out", _f32p), ("ld_dh", _i64), ("dh_split_stride", _i64),
        ("ws", _f32p),
        ("bf16", C.c_int),
        ("memory16", C.c_void_p),
        ("cell_q", C.POINTER(LstmBwd)), ("cell_x", C.POINTER(LstmBwd)),
        ("ws_floats", _i64),
        ("Wq16", C.c_void_p),
    ]


class DecTrain(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("Ti", C.c_int), ("To", C.c_int), ("E", C.c_int), ("Ha", C.c_int), ("Hd", C.c_int),
        ("Wa_rec", _f32p), ("Wd_cat", _f32p), ("bias_d", _f32p), ("Wq", _f32p), ("U", _f32p), ("v", _f32p),
        ("GA", _f32p), ("memory", _f32p), ("pm", _f32p), ("lens", C.c_void_p),
        ("keep_att", C.c_void_p), ("keep_dec", C.c_void_p),
        ("scale_att", C.c_float), ("scale_dec", C.c_float),
        ("HA", _f32p), ("CA", _f32p), ("GD", _f32p), ("HD", _f32p), ("CD", _f32p), ("CTX", _f32p),
        ("Q", _f32p), ("ALIGN", _f32p), ("CUM", _f32p), ("cum_work", _f32p), ("attn_ws", _f32p),
        ("bf16", C.c_int), ("Wa_rec16", C.c_void_p), ("Wd_cat16", C.c_void_p),
        ("HA16", C.c_void_p), ("HD16", C.c_void_p), ("CTX16", C.c_void_p),
        ("memory16", C.c_void_p), ("Wq16", C.c_void_p),
    ]


class DecTrainBwd(C.Structure):
    _fields_ = [
        ("f", DecTrain),
        ("Wa_recT", _f32p), ("Wd_catT", _f32p), ("DHC", _f32p), ("d_align", _f32p),
        ("nsplit", C.c_int),
        ("DGA", _f32p), ("DGD", _f32p), ("DCTX", _f32p), ("DQ", _f32p), ("d_pm", _f32p),
        ("dU_acc", _f32p), ("dv_acc", _f32p),
        ("dXd", _f32p), ("dXa", _f32p), ("dc_a", _f32p), ("dc_d", _f32p),
        ("dwin_part", _f32p), ("dcum_acc", _f32p), ("dq_h", _f32p),
        ("Wa_recT16", C.c_void_p), ("Wd_catT16", C.c_void_p), ("DGA16", C.c_void_p), ("DGD16", C.c_void_p),
        ("dg16_step_a", _i64), ("dg16_step_d", _i64),
        ("dXd_ring", C.c_int),
    ]


class LstmSeq(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("T", C.c_int), ("H", C.c_int), ("reverse", C.c_int),
        ("Whh", _f32p), ("WhhT", _f32p), ("GX", _f32p),
        ("out", _f32p), ("ld_out", _i64),
        ("C", _f32p), ("lens", C.c_void_p),
        ("dout", _f32p), ("ld_dout", _i64),
        ("DG", _f32p), ("dX", _f32p), ("dc", _f32p),
        ("dx_splits", C.c_int),
    ]


class SmallLinear(C.Structure):
    _fields_ = [
        ("X", _f32p), ("ldx", _i64), ("W", _f32p), ("ldw", _i64), ("bias", _f32p),
        ("Y", _f32p), ("ldy", _i64),
        ("B", C.c_int), ("N", C.c_int), ("K", C.c_int), ("act", C.c_int),
        ("keep", C.c_void_p), ("ldkeep", _i64), ("keep_scale", C.c_float),
    ]


class DecInfer(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("Ti", C.c_int), ("E", C.c_int), ("Ha", C.c_int), ("Hd", C.c_int),
        ("P", C.c_int), ("C", C.c_int),
        ("t0", C.c_int), ("n_steps", C.c_int), ("max_steps", C.c_int),
        ("gate_threshold", C.c_float),
        ("W1", _f32p), ("W2", _f32p), ("Wa_cat", _f32p), ("bias_a", _f32p), ("Wd_cat", _f32p),
        ("bias_d", _f32p), ("Wq", _f32p), ("U", _f32p), ("v", _f32p), ("Wpg", _f32p), ("bias_pg", _f32p),
        ("memory", _f32p), ("pm", _f32p), ("lens", C.c_void_p), ("keep_prenet", C.c_void_p),
        ("h_a", _f32p), ("c_a", _f32p), ("c_d", _f32p), ("hc", _f32p), ("cum", _f32p),
        ("x_prenet", _f32p), ("gates", _f32p), ("zero_frame", _f32p), ("attn_ws", _f32p),
        ("PG", _f32p), ("ALIGN", _f32p), ("out_lengths", C.c_void_p), ("active", C.c_void_p),
        ("done_count", C.c_void_p),
        ("bf16", C.c_int), ("Wa_cat16", C.c_void_p), ("Wd_cat16", C.c_void_p),
        ("x_prenet16", C.c_void_p), ("h_a16", C.c_void_p), ("hc16", C.c_void_p),
        ("Wf", _f32p), ("bias_f", _f32p), ("memory16", C.c_void_p), ("Wq16", C.c_void_p),
        ("Wf16", C.c_void_p), ("Wpg16", C.c_void_p), ("W2_16", C.c_void_p), ("x_prenet1_16", C.c_void_p),
        ("attn_ws_floats", _i64),
    ]


class DecPersist(C.Structure):
    _fields_ = [
        ("Ti", C.c_int), ("E", C.c_int), ("H", C.c_int), ("P", C.c_int), ("C", C.c_int),
        ("max_steps", C.c_int), ("gate_threshold", C.c_float),
        ("Wa16", C.c_void_p), ("Wd16", C.c_void_p), ("bias_a", _f32p), ("bias_d", _f32p),
        ("Wq", _f32p), ("U", _f32p), ("v", _f32p), ("Wf", _f32p), ("bias_f", _f32p), ("W2", _f32p),
        ("memory", _f32p), ("pm", _f32p), ("keep_prenet", C.c_void_p),
        ("PG", _f32p), ("ALIGN", _f32p), ("out_length", C.c_void_p), ("status", C.c_void_p),
        ("steps_done", C.c_void_p), ("mailbox", C.c_void_p), ("trace", _f32p), ("timing", C.c_void_p),
        ("weights_f32", C.c_int),
    ]


PERSIST_TIMEOUT = 7
MAX_TENSORS = 64


class TensorList(C.Structure):
    _fields_ = [("param", C.c_void_p * MAX_TENSORS), ("grad", C.c_void_p * MAX_TENSORS),
                ("exp_avg", C.c_void_p * MAX_TENSORS), ("exp_avg_sq", C.c_void_p * MAX_TENSORS),
                ("numel", C.c_longlong * MAX_TENSORS), ("first_block", C.c_int * MAX_TENSORS), ("count", C.c_int)]


class AdamHyper(C.Structure):
    _fields_ = [("step_size", C.c_float), ("bc2_sqrt", C.c_float), ("one_minus_beta1", C.c_float),
                ("beta2", C.c_float), ("one_minus_beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float)]


_STRUCTS = [GemmDesc, Seg, LstmStep, SkinnyGemm, Addend, LstmBwd, AttnFwd, AttnBwd, DecTrain,
            DecTrainBwd, LstmSeq, DecInfer, SmallLinear, TensorList, AdamHyper, DecPersist, Gemm16Desc]

# every exported symbol of include/tacotron2_amd.h
SYMBOLS = [
    "t2amd_abi_version", "t2amd_source_sha1", "t2amd_last_error", "t2amd_struct_sizes", "t2amd_set_validate_only", "t2amd_profile_enable", "t2amd_profile_read", "t2amd_profile_event_overhead",
    "t2amd_gemm_f32", "t2amd_gemm_tile_size", "t2amd_splitk_reduce_f32", "t2amd_splitk_reduce2d_f32", "t2amd_gemm16_tn", "t2amd_gemm16_kk", "t2amd_gemm16_kk_group", "t2amd_transpose_cast_bf16", "t2amd_cast_halo_bf16", "t2amd_pack_conv_bf16",
    "t2amd_bn_stats_f32", "t2amd_bn_eval_invstd_f32", "t2amd_bn_act_fwd_f32", "t2amd_bn_act_bwd_f32", "t2amd_bn_act_bwd_img_f32", "t2amd_colsum_bf16", "t2amd_bn_act_fwd_img_f32",
    "t2amd_colsum_f32",
    "t2amd_embedding_fwd_f32", "t2amd_embedding_bwd_f32", "t2amd_philox_keep_mask", "t2amd_fill_f32",
    "t2amd_copy2d_f32", "t2amd_cast_bf16_f32", "t2amd_split_bf16x3_f32", "t2amd_transpose_f32", "t2amd_frames_to_time_major_f32",
    "t2amd_split_projection_f32", "t2amd_finalize_outputs_f32", "t2amd_grads_to_channel_last_f32",
    "t2amd_gather_dout_f32", "t2amd_relu_dropout_bwd_f32",
    "t2amd_lstm_step_fwd_f32", "t2amd_skinny_gemm_f32", "t2amd_lstm_pointwise_bwd_f32",
    "t2amd_lstm_step_fwd2_f32", "t2amd_skinny_gemm2_f32", "t2amd_lstm_pointwise_bwd2_f32",
    "t2amd_fold_location_f32", "t2amd_unfold_location_grads_f32",
    "t2amd_attention_step_fwd_f32", "t2amd_attention_step_bwd_f32",
    "t2amd_decoder_train_fwd_loop_f32", "t2amd_decoder_train_bwd_loop_f32",
    "t2amd_decoder_train_fwd_persistent_flag_bytes", "t2amd_decoder_train_fwd_persistent_supported", "t2amd_decoder_train_fwd_persistent_f32",
    "t2amd_lstm_seq_fwd_f32", "t2amd_lstm_seq_bwd_f32", "t2amd_decoder_infer_steps_f32",
    "t2amd_set_decoder_streams", "t2amd_set_bptt_cell_fold", "t2amd_get_bptt_cell_fold", "t2amd_attn_bwd_ws_floats",
    "t2amd_set_small_batch_max", "t2amd_get_small_batch_max", "t2amd_dec_infer_uses_tiles",
    "t2amd_set_attn_bwd_granules", "t2amd_attn_handoff_timeouts", "t2amd_set_attn_bwd_fused", "t2amd_attn_fwd_ws_floats", "t2amd_set_attn_fwd_fused", "t2amd_lstm_step_small_f32", "t2amd_linear_small_f32",
    "t2amd_lstm_seq_fwd2_f32", "t2amd_lstm_seq_bwd2_f32",
    "t2amd_lstm_seq_persistent_mailbox_bytes", "t2amd_lstm_seq_persistent_supported", "t2amd_lstm_seq_fwd2_persistent_f32",
    "t2amd_lstm_seq_batch_persistent_flag_bytes", "t2amd_lstm_seq_batch_persistent_supported", "t2amd_lstm_seq_fwd2_batch_persistent_f32", "t2amd_encoder_handoff_timeouts",
    "t2amd_lstm_seq_bwd2_batch_persistent_supported", "t2amd_lstm_seq_bwd2_batch_persistent_f32",
    "t2amd_reflect_pad_f32", "t2amd_reflect_index", "t2amd_stft_magnitude_f32", "t2amd_mel_log_compress_f32",
    "t2amd_optim_chunk", "t2amd_grad_norm_f32", "t2amd_adam_step_f32",
    "t2amd_decoder_persist_mailbox_bytes", "t2amd_decoder_persist_supported", "t2amd_decoder_infer_persistent_f32",
    "t2amd_loss_workspace_doubles", "t2amd_tacotron2_loss_fwd_f32", "t2amd_tacotron2_loss_bwd_f32",
]

_P, _I, _L, _F, _UL = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_ulonglong


def _argtypes():
    pt = C.POINTER
    return {
        "t2amd_gemm_f32": [pt(GemmDesc), _P],
        "t2amd_gemm_tile_size": [_I, _I, _I, _I, _I, _I],
        "t2amd_splitk_reduce_f32": [_P, _I, _L, _P, _L, _I, _I, _I, _P],
        "t2amd_splitk_reduce2d_f32": [_P, _I, _L, _P, _I, _I, _L, _I, _P],
        "t2amd_gemm16_tn": [pt(Gemm16Desc), _P],
        "t2amd_gemm16_kk": [pt(Gemm16Desc), _P],
        "t2amd_gemm16_kk_group": [pt(Gemm16Desc), _I, _P],
        "t2amd_transpose_cast_bf16": [_P, _I, _L, _P, _L, _I, _I, _I, _P],
        "t2amd_cast_halo_bf16": [_P, _L, _P, _L, _I, _I, _I, _P],
        "t2amd_pack_conv_bf16": [_P, _P, _I, _I, _I, _I, _I, _P],
        "t2amd_bn_stats_f32": [_P, _L, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P],
        "t2amd_bn_eval_invstd_f32": [_P, _P, _I, _F, _P],
        "t2amd_bn_act_fwd_f32": [_P, _L, _P, _L, _I, _I, _P, _P, _P, _P, _I, _P, _L, _F, _P, _I, _P],
        "t2amd_bn_act_fwd_img_f32": [_P, _L, _P, _L, _I, _I, _P, _P, _P, _P, _I, _P, _L, _F, _P, _I, _I, _P],
        "t2amd_bn_act_bwd_f32": [_P, _L, _P, _L, _P, _L, _I, _I, _P, _P, _P, _I, _P, _L, _F, _P, _P, _P, _P],
        "t2amd_bn_act_bwd_img_f32": [_P, _L, _P, _L, _P, _L, _I, _I, _P, _P, _P, _I, _P, _L, _F, _P, _P, _P, _P, _I, _I, _P, _I, _P],
        "t2amd_colsum_f32": [_P, _L, _I, _I, _P, _P, _I, _P],
        "t2amd_colsum_bf16": [_P, _L, _I, _I, _P, _P, _I, _P],
        "t2amd_embedding_fwd_f32": [_P, _P, _P, _L, _I, _I, _P],
        "t2amd_embedding_bwd_f32": [_P, _P, _P, _P, _L, _I, _I, _P],
        "t2amd_philox_keep_mask": [_P, _L, _F, _UL, _UL, _P],
        "t2amd_fill_f32": [_P, _L, _F, _P],
        "t2amd_cast_bf16_f32": [_P, _P, _L, _P],
        "t2amd_split_bf16x3_f32": [_P, _L, _P, _L, _L, _I, _P],
        "t2amd_copy2d_f32": [_P, _L, _P, _L, _P, _L, _I, _I, _P],
        "t2amd_transpose_f32": [_P, _L, _P, _L, _I, _I, _I, _L, _L, _P],
        "t2amd_frames_to_time_major_f32": [_P, _P, _I, _I, _I, _P],
        "t2amd_split_projection_f32": [_P, _P, _P, _P, _I, _I, _I, _P],
        "t2amd_finalize_outputs_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
        "t2amd_grads_to_channel_last_f32": [_P, _P, _P, _P, _I, _I, _I, _P],
        "t2amd_gather_dout_f32": [_P, _P, _P, _I, _I, _I, _P],
        "t2amd_relu_dropout_bwd_f32": [_P, _P, _F, _L, _P],
        "t2amd_lstm_step_fwd_f32": [pt(LstmStep), _P],
        "t2amd_skinny_gemm_f32": [pt(SkinnyGemm), _P],
        "t2amd_lstm_pointwise_bwd_f32": [pt(LstmBwd), _P],
        "t2amd_lstm_step_fwd2_f32": [pt(LstmStep), pt(LstmStep), _P],
        "t2amd_skinny_gemm2_f32": [pt(SkinnyGemm), pt(SkinnyGemm), _P],
        "t2amd_lstm_pointwise_bwd2_f32": [pt(LstmBwd), pt(LstmBwd), _P],
        "t2amd_fold_location_f32": [_P, _P, _P, _P],
        "t2amd_unfold_location_grads_f32": [_P, _P, _I, _P, _P, _P, _P, _P, _P],
        "t2amd_attention_step_fwd_f32": [pt(AttnFwd), _P],
        "t2amd_attention_step_bwd_f32": [pt(AttnBwd), _P],
        "t2amd_decoder_train_fwd_loop_f32": [pt(DecTrain), _P],
        "t2amd_decoder_train_fwd_persistent_flag_bytes": [_I, _I],
        "t2amd_decoder_train_fwd_persistent_supported": [pt(DecTrain), _I],
        "t2amd_decoder_train_fwd_persistent_f32": [pt(DecTrain), _P, _P, _P, _P],
        "t2amd_decoder_train_bwd_loop_f32": [pt(DecTrainBwd), _P],
        "t2amd_lstm_seq_fwd_f32": [pt(LstmSeq), _P],
        "t2amd_lstm_seq_bwd_f32": [pt(LstmSeq), _P],
        "t2amd_lstm_seq_fwd2_f32": [pt(LstmSeq), pt(LstmSeq), _P],
        "t2amd_lstm_seq_bwd2_f32": [pt(LstmSeq), pt(LstmSeq), _P],
        "t2amd_lstm_seq_persistent_mailbox_bytes": [_I, _I],
        "t2amd_lstm_seq_persistent_supported": [pt(LstmSeq)],
        "t2amd_lstm_seq_fwd2_persistent_f32": [pt(LstmSeq), pt(LstmSeq), _P, _P, _P],
        "t2amd_lstm_seq_batch_persistent_flag_bytes": [_I, _I, _I],
        "t2amd_lstm_seq_batch_persistent_supported": [pt(LstmSeq), _I, _I],
        "t2amd_lstm_seq_fwd2_batch_persistent_f32": [pt(LstmSeq), pt(LstmSeq), _P, _P, _P, _P],
        "t2amd_lstm_seq_bwd2_batch_persistent_supported": [pt(LstmSeq), _I, _I],
        "t2amd_lstm_seq_bwd2_batch_persistent_f32": [pt(LstmSeq), pt(LstmSeq), _P, _P, _P, _P],
        "t2amd_encoder_handoff_timeouts": [_I],
        "t2amd_decoder_infer_steps_f32": [pt(DecInfer), _P],
        "t2amd_struct_sizes": [pt(C.c_int), _I],
        "t2amd_set_validate_only": [_I],
        "t2amd_set_decoder_streams": [_I],
        "t2amd_set_bptt_cell_fold": [_I],
        "t2amd_get_bptt_cell_fold": [],
        "t2amd_set_small_batch_max": [_I],
        "t2amd_get_small_batch_max": [_I],
        "t2amd_dec_infer_uses_tiles": [_I, _I, _I, _I, _I, _I],
        "t2amd_attn_bwd_ws_floats": [_I, _I],
        "t2amd_set_attn_bwd_granules": [_I],
        "t2amd_attn_handoff_timeouts": [_I],
        "t2amd_set_attn_bwd_fused": [_I],
        "t2amd_attn_fwd_ws_floats": [_I, _I],
        "t2amd_set_attn_fwd_fused": [_I],
        "t2amd_lstm_step_small_f32": [pt(LstmStep), _P],
        "t2amd_linear_small_f32": [pt(SmallLinear), _P],
        "t2amd_profile_enable": [_I, _I],
        "t2amd_profile_read": [pt(C.c_float), pt(C.c_int)],
        "t2amd_profile_event_overhead": [pt(C.c_float)],
        "t2amd_reflect_pad_f32": [_P, _L, _P, _L, _I, _I, _I, _I, _P],
        "t2amd_stft_magnitude_f32": [_P, _L, _P, _L, _L, _I, _I, _P],
        "t2amd_mel_log_compress_f32": [_P, _L, _P, _I, _I, _I, _F, _P],
        "t2amd_optim_chunk": [],
        "t2amd_grad_norm_f32": [pt(TensorList), _F, _P, _P, _P],
        "t2amd_adam_step_f32": [pt(TensorList), pt(AdamHyper), _P, _P],
        "t2amd_decoder_persist_mailbox_bytes": [_I, _I, _I, _I],
        "t2amd_decoder_persist_supported": [pt(DecPersist)],
        "t2amd_decoder_infer_persistent_f32": [pt(DecPersist), _P],
        "t2amd_loss_workspace_doubles": [],
        "t2amd_tacotron2_loss_fwd_f32": [_P, _P, _P, _L, _P, _P, _L, _P, _P, _P],
        "t2amd_tacotron2_loss_bwd_f32": [_P, _P, _P, _L, _P, _P, _L, _P, _P, _P, _P, _P],
    }


_lib = None


class NativeError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (once).  Raises NativeError when it is missing —
    there is no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "tacotron2_amd: %s not found. The MI355X engine has no fallback path; build it with "
            "`python -m tacotron2_amd.build` (needs hipcc)." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name in SYMBOLS:
        if not hasattr(lib, name):
            raise NativeError("tacotron2_amd: symbol %s missing from %s" % (name, LIB_PATH))
    for name, at in _argtypes().items():
        fn = getattr(lib, name)
        fn.argtypes = at
        fn.restype = C.c_int
    lib.t2amd_last_error.restype = C.c_char_p
    lib.t2amd_reflect_index.argtypes = [_L, _L]
    lib.t2amd_reflect_index.restype = C.c_longlong
    lib.t2amd_abi_version.restype = C.c_int
    lib.t2amd_attn_bwd_ws_floats.restype = C.c_longlong
    lib.t2amd_attn_fwd_ws_floats.restype = C.c_longlong
    lib.t2amd_decoder_persist_mailbox_bytes.restype = C.c_longlong
    lib.t2amd_lstm_seq_persistent_mailbox_bytes.restype = C.c_longlong
    lib.t2amd_lstm_seq_batch_persistent_flag_bytes.restype = C.c_longlong
    lib.t2amd_decoder_train_fwd_persistent_flag_bytes.restype = C.c_longlong
    if lib.t2amd_abi_version() != 1:
        raise NativeError("tacotron2_amd: ABI version mismatch")
    # the binary must be THESE sources' binary (by content: a snapshot copy or a checkout resets mtimes).  An explicitly
    # selected library (T2AMD_LIB: instrumented / A-B variants built by tacotron2_amd.build) carries the hash of the same
    # tree; T2AMD_ALLOW_STALE_LIB=1 is the escape hatch for a deliberately older binary.
    lib.t2amd_source_sha1.restype = C.c_char_p
    lib.t2amd_source_sha1.argtypes = []
    built = (lib.t2amd_source_sha1() or b"").decode()
    if os.path.isdir(os.path.join(_HERE, "csrc")) and os.environ.get("T2AMD_ALLOW_STALE_LIB", "0") != "1":
        from . import build as _build
        try:
            want = _build.source_sha1()
        except OSError as e:             # e.g. installed as package data without include/: nothing to compare against
            want = None
            if os.environ.get("T2AMD_REQUIRE_SOURCE_HASH", "0") == "1":
                raise NativeError("tacotron2_amd: cannot hash the kernel sources beside %s (%s)" % (LIB_PATH, e))
            # csrc/ is here but not everything the hash covers: say that the stale-library check did NOT happen --
            # silently loading whatever .so lies there would defeat "binary and sources are provably the same tree"
            import sys
            print("tacotron2_amd: warning: the kernel sources beside %s could not be hashed (%s); the library (built from %s) "
                  "is loaded WITHOUT the stale-library check (T2AMD_REQUIRE_SOURCE_HASH=1 makes this an error)"
                  % (LIB_PATH, e, built or "<unstamped sources>"), file=sys.stderr, flush=True)
        if want is not None and built != want:
            raise NativeError("tacotron2_amd: %s was built from other sources (binary %s, sources %s): run "
                              "`python -m tacotron2_amd.build` (or set T2AMD_ALLOW_STALE_LIB=1 to use it anyway)"
                              % (LIB_PATH, built or "<unstamped>", want))
    sizes = (C.c_int * 32)()
    n = lib.t2amd_struct_sizes(sizes, 32)
    if n != len(_STRUCTS):
        raise NativeError("tacotron2_amd: struct count mismatch (%d vs %d)" % (n, len(_STRUCTS)))
    for i, st in enumerate(_STRUCTS):
        if C.sizeof(st) != sizes[i]:
            raise NativeError("tacotron2_amd: sizeof(%s) = %d in Python, %d in the library"
                              % (st.__name__, C.sizeof(st), sizes[i]))
    _lib = lib
    return lib


def library_sha1():
    """Source hash compiled into the loaded library (``t2amd_source_sha1``)."""
    return (load().t2amd_source_sha1() or b"").decode()


_validate_only = False


def cast_bf16(src, dst):
    """dst (torch.bfloat16, contiguous) = bf16(src) with round-to-nearest-even."""
    _fullc(src), _fullc(dst)
    if src.numel() != dst.numel() or dst.dtype != torch.bfloat16:
        raise NativeError("cast_bf16: dst must be a bfloat16 tensor of the same size")
    _check(load().t2amd_cast_bf16_f32(ptr(src), ptr(dst, torch.bfloat16), src.numel(), _stream()), "t2amd_cast_bf16_f32")


def split_bf16x3(src, dst):
    """dst (torch.bfloat16, [rows][2 K], contiguous) = the split-bf16 operand image of src (f32 [rows][K], contiguous, K % 16 == 0):
    per 16 k, 16 hi = bf16(x) then 16 lo = bf16(x - hi) -- what t2amd_lstm_step.bf16 == 3 / the 'bf16x3' mode multiplies."""
    _fullc(src), _fullc(dst)
    K = src.shape[-1]
    rows = src.numel() // max(K, 1)
    if dst.dtype != torch.bfloat16 or dst.numel() != 2 * src.numel() or K % 16 != 0:
        raise NativeError("split_bf16x3: dst must be a bfloat16 tensor of twice the size, K %% 16 == 0 (src %s, dst %s)"
                          % (tuple(src.shape), tuple(dst.shape)))
    _check(load().t2amd_split_bf16x3_f32(ptr(src), _i64(K), ptr(dst, torch.bfloat16), _i64(K), _i64(rows), K, _stream()),
           "t2amd_split_bf16x3_f32")


def decoder_persist_mailbox_bytes(Ti, E, H, P):
    return int(load().t2amd_decoder_persist_mailbox_bytes(int(Ti), int(E), int(H), int(P)))


def decoder_persist_supported(desc):
    """None when the persistent single-utterance decode kernel can run this geometry, else the reason."""
    lib = load()
    if lib.t2amd_decoder_persist_supported(C.byref(desc)) == 0:
        return None
    msg = lib.t2amd_last_error()
    return msg.decode() if msg else "unsupported"


def decoder_infer_persistent(desc, reads=None, writes=None):
    """The whole free-running decode loop of ONE utterance as one persistent launch (csrc/decode_persist.hip)."""
    _loop("decoder_infer_persistent", "t2amd_decoder_infer_persistent_f32", [desc], reads, writes)


def tacotron2_loss_fwd(mel, post, tgt, gate, gate_tgt, ws, out4):
    for t in (mel, post, tgt, gate, gate_tgt, out4):
        _fullc(t)
    if mel.numel() != post.numel() or mel.numel() != tgt.numel() or gate.numel() != gate_tgt.numel():
        raise NativeError("tacotron2_loss: shape mismatch between outputs and targets")
    _check(load().t2amd_tacotron2_loss_fwd_f32(ptr(mel), ptr(post), ptr(tgt), mel.numel(), ptr(gate), ptr(gate_tgt),
                                               gate.numel(), ptr(ws, torch.float64), ptr(out4), _stream()),
           "t2amd_tacotron2_loss_fwd_f32")


def tacotron2_loss_bwd(mel, post, tgt, gate, gate_tgt, upstream, d_mel, d_post, d_gate):
    for t in (mel, post, tgt, gate, gate_tgt, upstream, d_mel, d_post, d_gate):
        _fullc(t)
    _check(load().t2amd_tacotron2_loss_bwd_f32(ptr(mel), ptr(post), ptr(tgt), mel.numel(), ptr(gate), ptr(gate_tgt),
                                               gate.numel(), ptr(upstream), ptr(d_mel), ptr(d_post), ptr(d_gate), _stream()),
           "t2amd_tacotron2_loss_bwd_f32")


def loss_workspace_doubles():
    return int(load().t2amd_loss_workspace_doubles())


_decoder_streams = 1


def set_decoder_streams(n):
    """1 (default): single stream, fused launches; 2: decoder-LSTM chain of the training loops on a side stream."""
    global _decoder_streams
    _check(load().t2amd_set_decoder_streams(int(n)), "t2amd_set_decoder_streams")
    _decoder_streams = int(n)


def decoder_streams():
    return _decoder_streams


def set_bptt_cell_fold(on):
    """1: the LSTM cell backwards of a decoder BPTT step run inside the step's attention-backward launch (5 dependent
    launches per time step instead of 6); 0: as a launch of their own.  Bit-identical gradients either way."""
    _check(load().t2amd_set_bptt_cell_fold(1 if on else 0), "t2amd_set_bptt_cell_fold")


def get_bptt_cell_fold():
    return int(load().t2amd_get_bptt_cell_fold())


def set_small_batch_max(n):
    """Free-running decoder: the largest batch the matrix-vector kernels serve (0 .. 8; -1 = by operand mode: 3 rows with bf16
    operands, 4 otherwise); above it the 64-row MFMA tiles."""
    _check(load().t2amd_set_small_batch_max(int(n)), "t2amd_set_small_batch_max")


def small_batch_max(bf16=0):
    """The boundary in force for operand mode ``bf16`` (t2amd_dec_infer.bf16: 0 f32, 1 bf16, 3 split-bf16)."""
    return int(load().t2amd_get_small_batch_max(int(bf16)))


def dec_infer_uses_tiles(B, bf16, E, Ha, Hd, P):
    """True when the free-running decoder's launch chain runs B rows in operand mode ``bf16`` on the 64-row tiles."""
    return bool(load().t2amd_dec_infer_uses_tiles(int(B), int(bf16), int(E), int(Ha), int(Hd), int(P)))


def small_batch_max_setting():
    """The raw setting (-1 = by operand mode), for save / restore around a test."""
    a, b = small_batch_max(0), small_batch_max(1)
    return -1 if a != b else a


def set_validate_only(on):
    """Argument-check mode for CPU tests: host logic runs, kernels are not launched, tensors may
    live on the CPU and outputs stay uninitialised.  Never a compute path."""
    global _validate_only
    load().t2amd_set_validate_only(1 if on else 0)
    _validate_only = bool(on)


def validate_only():
    return _validate_only


def profile_enable(tag, max_launches):
    """Bracket every LSTM-step launch with role `tag` (1 attention LSTM, 2 decoder LSTM) by HIP events."""
    _check(load().t2amd_profile_enable(tag, max_launches), "t2amd_profile_enable")


def profile_event_overhead():
    """ms of an empty event bracket on the launch stream (call before profile_read)."""
    ms = C.c_float(0)
    _check(load().t2amd_profile_event_overhead(C.byref(ms)), "t2amd_profile_event_overhead")
    return ms.value


def profile_read():
    """-> (total_ms, count) of the bracketed launches; disables profiling."""
    ms, n = C.c_float(0), C.c_int(0)
    _check(load().t2amd_profile_read(C.byref(ms), C.byref(n)), "t2amd_profile_read")
    return ms.value, n.value


def _check(rc, what):
    if rc != 0:
        msg = _lib.t2amd_last_error()
        raise NativeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def _stream():
    if _validate_only:
        return None
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32):
    """Raw device pointer of a tensor (None -> NULL), with dtype/device checks."""
    if t is None:
        return None
    if not t.is_cuda and not _validate_only:
        raise NativeError("tacotron2_amd: expected a device (HIP) tensor, got a %s tensor; the engine "
                          "has no CPU path" % t.device)
    if dtype is not None and t.dtype != dtype:
        raise NativeError("tacotron2_amd: expected %s, got %s" % (dtype, t.dtype))
    return C.c_void_p(t.data_ptr())


def _mat(t, dtype=torch.float32):
    """2-D view with unit inner stride -> (ptr, ld, rows, cols)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise NativeError("expected a 2-D tensor with contiguous rows, got shape %s stride %s"
                          % (tuple(t.shape), t.stride()))
    ld = t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
    return ptr(t, dtype), ld, t.shape[0], t.shape[1]


def _fullc(t):
    if not t.is_contiguous():
        raise NativeError("expected a contiguous tensor, got shape %s stride %s" % (tuple(t.shape), t.stride()))
    return t


def scale_for(p):
    """fp32 1/(1-p), formed like ATen's dropout (noise.div_(1-p))."""
    return float(torch.tensor(1.0, dtype=torch.float32) / torch.tensor(1.0 - p, dtype=torch.float32))


# ----------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------
def gemm_tile_size(M, N, precision, nz=1, a_km=False, b_kn=False):
    """Tile edge (128/256) the library will use for this product launched as nz = batch*splitk slices."""
    return int(load().t2amd_gemm_tile_size(int(M), int(N), int(precision), int(nz), 0 if a_km else 1, 0 if b_kn else 1))


def gemm(Cm, A, B, a_km=False, b_kn=False, accumulate=False, bias=None, act=0, keep=None,
         keep_scale=1.0, convA=None, convB=None, batch=1, strides=(0, 0, 0), splitk=1, partials=None,
         fast=False):
    """Cm[M,N] (+)= act(A.B + bias)*keep.

    A is a view [M,K] (default) or [K,M] when ``a_km``;  B is a view [N,K] (default, the
    nn.Linear weight layout) or [K,N] when ``b_kn``.  ``convA=(T,C,pad,sign)`` / ``convB=(T,C,pad)``
    switch on implicit-convolution addressing (see the header).  With ``splitk>1`` the partial
    results go to ``partials`` ([splitk, M*N] contiguous) and Cm is not written."""
    lib = load()
    d = GemmDesc()
    pa, lda, r0, c0 = _mat(A)
    pb, ldb, r1, c1 = _mat(B)
    pc, ldc, M, N = _mat(Cm)
    Ka, Ma = (r0, c0) if a_km else (c0, r0)
    Kb, Nb = (r1, c1) if b_kn else (c1, r1)
    if convA is not None:
        # A is the plain activation matrix [rows, C]; the virtual K is taps*C, taken from B
        if a_km or Ka != convA[1] or Kb % convA[1] != 0:
            raise NativeError("gemm: convA needs A=[rows,C] and K(B) a multiple of C")
        K = Kb
    else:
        K = Ka
    if convB is not None:
        # B is the plain activation matrix [rows, C] (K = rows); the virtual N is taps*C, taken from C
        if not b_kn or Nb != convB[1] or N % convB[1] != 0:
            raise NativeError("gemm: convB needs B=[rows,C] (b_kn) and N(C) a multiple of C")
        Nb = N
    if Ma != M or Nb != N or Kb != K:
        raise NativeError("gemm: shape mismatch C=%s A=%s B=%s (a_km=%s b_kn=%s K=%d)"
                          % (tuple(Cm.shape), tuple(A.shape), tuple(B.shape), a_km, b_kn, K))
    d.A, d.B, d.C = pa, pb, pc
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.a_kcontig = 0 if a_km else 1
    d.b_kcontig = 0 if b_kn else 1
    d.batch = batch
    d.strideA, d.strideB, d.strideC = strides
    d.splitk = splitk
    d.precision = int(fast)          # False/0 exact f32, True/1 split-bf16 x3, 2 plain bf16
    d.accumulate = 1 if accumulate else 0
    d.bias = ptr(bias)
    d.act = act
    if keep is not None:
        if keep.dim() != 2 or tuple(keep.shape) != (M, N) or keep.stride(1) != 1:
            raise NativeError("gemm: keep mask must be a [M,N] uint8 view, got %s" % (tuple(keep.shape),))
        d.keep, d.ldkeep, d.keep_scale = ptr(keep, torch.uint8), keep.stride(0), keep_scale
    if convA is not None:
        d.convA_T, d.convA_C, d.convA_pad, d.convA_sign = convA
    if convB is not None:
        d.convB_T, d.convB_C, d.convB_pad = convB
    if splitk > 1:
        if partials is None or partials.numel() < splitk * M * N or not partials.is_contiguous():
            raise NativeError("gemm: split-K needs a contiguous partials buffer")
        d.C = ptr(partials)
        d.ldc = N
        d.strideSplitC = M * N
    _check(lib.t2amd_gemm_f32(C.byref(d), _stream()), "t2amd_gemm_f32")


def gemm16_tn(Cm, A16, B16, K=None, splitk=1, partials=None, accumulate=False, bias=None):
    """Cm[M,N] (f32) = A16[M,K] . B16[N,K]^T, bf16 K-contiguous operands (csrc/gemm16.hip).  ``K`` defaults to the operands'
    width (a multiple of 64); with ``splitk`` > 1 the partial products go to ``partials`` (splitk, M*N)."""
    lib = load()
    d = Gemm16Desc()
    pa, lda, M, Ka = _mat(A16, torch.bfloat16)
    pb, ldb, N, Kb = _mat(B16, torch.bfloat16)
    K = Ka if K is None else K
    if K > Ka or K > Kb or Cm.shape[0] != M or Cm.shape[1] != N:
        raise NativeError("gemm16_tn: shape mismatch A=%s B=%s C=%s K=%d" % (tuple(A16.shape), tuple(B16.shape), tuple(Cm.shape), K))
    d.A, d.B, d.M, d.N, d.K, d.lda, d.ldb = pa, pb, M, N, K, lda, ldb
    d.splitk = splitk
    if splitk > 1:
        if partials is None or partials.numel() < splitk * M * N:
            raise NativeError("gemm16_tn: split-K needs a (splitk, M*N) partials buffer")
        d.C, d.ldc, d.strideSplitC = ptr(partials), N, M * N
    else:
        d.C, d.ldc = _mat(Cm)[:2]
    d.accumulate = 1 if accumulate else 0
    d.bias = ptr(bias)
    _check(lib.t2amd_gemm16_tn(C.byref(d), _stream()), "t2amd_gemm16_tn")


def conv16(Cm, img16, W16, B, T, pad, bias=None, accumulate=False):
    """1-d convolution over channel-last rows on the bf16-resident product: Cm[(b T + t), co] (+)= sum_{tap, ci}
    img16[b (T + 2 pad) + t + tap][ci] * W16[co][tap Ci + ci] (+ bias).  ``img16`` is cast_halo_bf16's image
    ((B (T + 2 pad) + 2 pad) rows of Ci), ``W16`` the bf16 weights packed [Co][k Ci]."""
    lib = load()
    d = Gemm16Desc()
    Ci = img16.shape[1]
    Tp = T + 2 * pad
    pw, ldw, Co, K = _mat(W16, torch.bfloat16)
    # K may be the window length k Ci rounded up to the 64-deep k-steps (zero weight columns, pack_conv_bf16): the overhang of
    # the last real window, row (B - 1) Tp + T - 1, must stay inside the image's trailing 2 pad halo rows
    klen = (2 * pad + 1) * Ci
    if img16.shape[0] < B * Tp + 2 * pad or K < klen or K - klen > 2 * pad * Ci or Cm.shape[0] != B * T or Cm.shape[1] != Co:
        raise NativeError("conv16: shape mismatch img=%s W=%s C=%s" % (tuple(img16.shape), tuple(W16.shape), tuple(Cm.shape)))
    d.A, d.lda = ptr(_fullc(img16), torch.bfloat16), Ci
    d.B, d.ldb = pw, ldw
    d.C, d.ldc = _mat(Cm)[:2]
    d.M, d.N, d.K = (B - 1) * Tp + T, Co, K             # window rows up to the last real output (the rest would be discarded)
    d.splitk, d.accumulate, d.bias = 1, 1 if accumulate else 0, ptr(bias)
    d.win_T, d.win_Tp = T, Tp
    _check(lib.t2amd_gemm16_tn(C.byref(d), _stream()), "t2amd_gemm16_tn")


def _kk_desc(Cm, A16, B16, K, M=None, N=None, lda=None, ldb=None, splitk=1, partials=None, accumulate=False, bias=None):
    d = Gemm16Desc()
    pa, lda0, ra, wa = _mat(A16, torch.bfloat16)
    pb, ldb0, rb, wb = _mat(B16, torch.bfloat16)
    M = wa if M is None else M
    N = wb if N is None else N
    lda = lda0 if lda is None else lda
    ldb = ldb0 if ldb is None else ldb
    # the elements a product touches must lie inside the tensors handed in (contiguous storage from their first element)
    span_a = (ra - 1) * lda0 + wa
    span_b = (rb - 1) * ldb0 + wb
    if K < 1 or (K - 1) * lda + M > span_a or (K - 1) * ldb + N > span_b or Cm.shape[0] != M or Cm.shape[1] != N:
        raise NativeError("gemm16_kk: shape mismatch A=%s B=%s C=%s M=%d N=%d K=%d lda=%d ldb=%d"
                          % (tuple(A16.shape), tuple(B16.shape), tuple(Cm.shape), M, N, K, lda, ldb))
    d.A, d.B, d.M, d.N, d.K, d.lda, d.ldb = pa, pb, M, N, K, lda, ldb
    d.splitk = splitk
    if splitk > 1:
        if partials is None or partials.numel() < splitk * M * N:
            raise NativeError("gemm16_kk: split-K needs a (splitk, M*N) partials buffer")
        d.C, d.ldc, d.strideSplitC = ptr(partials), N, M * N
    else:
        d.C, d.ldc = _mat(Cm)[:2]
    d.accumulate = 1 if accumulate else 0
    d.bias = ptr(bias)
    return d


def gemm16_kk(Cm, A16, B16, K, M=None, N=None, lda=None, ldb=None, splitk=1, partials=None, accumulate=False, bias=None):
    """Cm[M,N] (f32) = sum_k A16[k, m] . B16[k, n]: bf16 K-MAJOR operands (csrc/gemm16.hip, transposing LDS reads) -- the
    layout of the time loops' slabs and of channel-last images.  ``A16`` / ``B16`` are 2-D bf16 tensors whose rows are the k
    index; ``M`` / ``N`` / ``lda`` / ``ldb`` default to their widths and row strides.  A row stride SMALLER than the width
    expresses overlapping rows (B[k][n] = img[k Ci + n], n < taps Ci: the windows of a convolution); the tensor must then
    hold the last row's reach.  With ``splitk`` > 1 the partial products go to ``partials`` (splitk, M*N)."""
    d = _kk_desc(Cm, A16, B16, K, M, N, lda, ldb, splitk, partials, accumulate, bias)
    _check(load().t2amd_gemm16_kk(C.byref(d), _stream()), "t2amd_gemm16_kk")


def gemm16_kk_group(problems):
    """Up to four K-major products in one launch (same M, same splitk): ``problems`` is a list of keyword dictionaries of
    gemm16_kk.  Their workgroups run side by side on the same rows of A (read once): the input blocks of an LSTM's dW."""
    if not 1 <= len(problems) <= 4:
        raise NativeError("gemm16_kk_group: 1..4 problems, got %d" % len(problems))
    arr = (Gemm16Desc * len(problems))(*[_kk_desc(**kw) for kw in problems])
    # through the dispatcher when the registration library is there (torch.ops.tacotron2_amd.wgrad_gemm16), else ctypes
    reads = [kw[k_] for kw in problems for k_ in ('A16', 'B16')]
    writes = [kw['partials'] if kw.get('partials') is not None else kw['Cm'] for kw in problems]
    if not _via_ops("wgrad_gemm16", [arr], reads, writes):
        _check(load().t2amd_gemm16_kk_group(arr, len(problems), _stream()), "t2amd_gemm16_kk_group")


def pack_conv_bf16(W, reversed=False):
    """bf16 image of a Conv1d weight (Co, Ci, k) for the window mode: [Co][Kp] rows (tap, ci) for the forward, or, with
    ``reversed``, [Ci][Kp] rows (k - 1 - tap, co) for the data gradient; Kp = the row length rounded up to 64, zero behind it."""
    Co, Ci, k = W.shape
    rows, inner = (Ci, Co) if reversed else (Co, Ci)
    Kp = (k * inner + 63) // 64 * 64
    out = torch.empty(rows, Kp, dtype=torch.bfloat16, device=W.device)
    _check(load().t2amd_pack_conv_bf16(ptr(_fullc(W)), ptr(out, torch.bfloat16), Co, Ci, k, Kp, 1 if reversed else 0, _stream()),
           "t2amd_pack_conv_bf16")
    return out


def cast_halo_bf16(src, dst, T, pad):
    """dst[(b (T + 2 pad) + pad + t), c] (bf16) = src[(b T + t), c]; every row of the image -- (rows / T) (T + 2 pad) + 2 pad of
    them -- is written, the halo rows with zeros (dst need not be initialised)."""
    ps, lds, rows, Cc = _mat(src)
    if dst.dtype != torch.bfloat16 or dst.shape[1] != Cc or dst.shape[0] < (rows // T) * (T + 2 * pad) + 2 * pad:
        raise NativeError("cast_halo_bf16: shape mismatch src=%s dst=%s" % (tuple(src.shape), tuple(dst.shape)))
    _check(load().t2amd_cast_halo_bf16(ps, _i64(lds), ptr(_fullc(dst), torch.bfloat16), _i64(rows), Cc, T, pad, _stream()),
           "t2amd_cast_halo_bf16")


def transpose_cast_bf16(src, dst, rows_padded=None):
    """dst[c, r] (bf16) = src[r, c] (f32 or bf16); columns rows..rows_padded-1 of dst are zeroed."""
    lib = load()
    is16 = src.dtype == torch.bfloat16
    ps, lds, rows, cols = _mat(src, torch.bfloat16 if is16 else torch.float32)
    pd, ldd, c2, rp = _mat(dst, torch.bfloat16)
    rows_padded = rp if rows_padded is None else rows_padded
    if c2 != cols or rows_padded < rows or rows_padded > rp:
        raise NativeError("transpose_cast_bf16: shape mismatch src=%s dst=%s" % (tuple(src.shape), tuple(dst.shape)))
    _check(lib.t2amd_transpose_cast_bf16(ps, 1 if is16 else 0, _i64(lds), pd, _i64(ldd), rows, cols, rows_padded, _stream()),
           "t2amd_transpose_cast_bf16")


def splitk_reduce(partials, nsplit, out, accumulate=False, perm_taps=0, perm_ci=0):
    lib = load()
    n = out.numel()
    _fullc(out)
    _check(lib.t2amd_splitk_reduce_f32(ptr(partials), nsplit, _i64(n), ptr(out), _i64(n),
                                       1 if accumulate else 0, perm_taps, perm_ci, _stream()),
           "t2amd_splitk_reduce_f32")


def splitk_reduce2d(partials, nsplit, out, accumulate=False):
    """out[r, c] (+)= sum_s partials[s][r cols + c]; ``out`` may be a column block of a wider matrix."""
    po, ldo, rows, cols = _mat(out)
    _check(load().t2amd_splitk_reduce2d_f32(ptr(partials), nsplit, _i64(rows * cols), po, rows, cols, _i64(ldo),
                                            1 if accumulate else 0, _stream()), "t2amd_splitk_reduce2d_f32")


# ----------------------------------------------------------------------------
# normalisation / elementwise
# ----------------------------------------------------------------------------
def bn_stats(x, ws, mean, invstd, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    lib = load()
    px, ldx, M, N = _mat(x)
    _check(lib.t2amd_bn_stats_f32(px, _i64(ldx), M, N, ptr(ws, torch.float64), ptr(mean), ptr(invstd),
                                  ptr(running_mean), ptr(running_var), C.c_float(momentum), C.c_float(eps),
                                  _stream()), "t2amd_bn_stats_f32")


def bn_eval_invstd(running_var, invstd, eps=1e-5):
    lib = load()
    _check(lib.t2amd_bn_eval_invstd_f32(ptr(running_var), ptr(invstd), running_var.numel(), C.c_float(eps),
                                        _stream()), "t2amd_bn_eval_invstd_f32")


def bn_act_fwd(x, y, mean, invstd, gamma, beta, act, keep=None, keep_scale=1.0, lens=None, T=0):
    lib = load()
    px, ldx, M, N = _mat(x)
    py, ldy, M2, N2 = _mat(y)
    assert (M, N) == (M2, N2)
    pk, ldk = (None, 0)
    if keep is not None:
        assert tuple(keep.shape) == (M, N) and keep.stride(1) == 1
        pk, ldk = ptr(keep, torch.uint8), keep.stride(0)
    _check(lib.t2amd_bn_act_fwd_f32(px, _i64(ldx), py, _i64(ldy), M, N, ptr(mean), ptr(invstd), ptr(gamma),
                                    ptr(beta), act, pk, _i64(ldk), C.c_float(keep_scale),
                                    ptr(lens, torch.int32), T, _stream()), "t2amd_bn_act_fwd_f32")


def bn_act_fwd_img(x, y, mean, invstd, gamma, beta, act, keep, keep_scale, y_img, T, pad):
    """bn_act_fwd with y leaving also as the bf16 halo image ``y_img`` ([M/T (T + 2 pad) + 2 pad][N]) the next convolution reads."""
    lib = load()
    px, ldx, M, N = _mat(x)
    py, ldy, M2, N2 = _mat(y)
    assert (M, N) == (M2, N2)
    pk, ldk = (None, 0)
    if keep is not None:
        assert tuple(keep.shape) == (M, N) and keep.stride(1) == 1
        pk, ldk = ptr(keep, torch.uint8), keep.stride(0)
    assert y_img.is_contiguous() and y_img.dtype == torch.bfloat16 and tuple(y_img.shape) == ((M // T) * (T + 2 * pad) + 2 * pad, N)
    _check(lib.t2amd_bn_act_fwd_img_f32(px, _i64(ldx), py, _i64(ldy), M, N, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), act, pk,
                                        _i64(ldk), C.c_float(keep_scale), ptr(y_img, torch.bfloat16), int(T), int(pad), _stream()),
           "t2amd_bn_act_fwd_img_f32")


def bn_act_bwd(dy, y, x, mean, invstd, gamma, act, keep, keep_scale, ws, dgamma, dbeta):
    lib = load()
    pd, ldd, M, N = _mat(dy)
    py, ldy, _, _ = _mat(y)
    px, ldx, _, _ = _mat(x)
    pk, ldk = (None, 0)
    if keep is not None:
        assert tuple(keep.shape) == (M, N) and keep.stride(1) == 1
        pk, ldk = ptr(keep, torch.uint8), keep.stride(0)
    _check(lib.t2amd_bn_act_bwd_f32(pd, _i64(ldd), py, _i64(ldy), px, _i64(ldx), M, N, ptr(mean), ptr(invstd),
                                    ptr(gamma), act, pk, _i64(ldk), C.c_float(keep_scale),
                                    ptr(ws, torch.float64), ptr(dgamma), ptr(dbeta), _stream()),
           "t2amd_bn_act_bwd_f32")


def bn_act_bwd_img(dy, y, x, mean, invstd, gamma, act, keep, keep_scale, ws, dgamma, dbeta, dx_img, T, pad, dbias, keep_f32=False):
    """bn_act_bwd with stage 2's output leaving as the bf16 halo image ``dx_img`` ([M/T (T + 2 pad) + 2 pad][N]) and as the bias
    gradient ``dbias`` [N] (t2amd_bn_act_bwd_img_f32); ``keep_f32``: also as the f32 slab in ``dy``."""
    lib = load()
    pd, ldd, M, N = _mat(dy)
    py, ldy, _, _ = _mat(y)
    px, ldx, _, _ = _mat(x)
    pk, ldk = (None, 0)
    if keep is not None:
        assert tuple(keep.shape) == (M, N) and keep.stride(1) == 1
        pk, ldk = ptr(keep, torch.uint8), keep.stride(0)
    assert dx_img.is_contiguous() and dx_img.dtype == torch.bfloat16 and tuple(dx_img.shape) == ((M // T) * (T + 2 * pad) + 2 * pad, N)
    assert dbias.numel() == N
    _check(lib.t2amd_bn_act_bwd_img_f32(pd, _i64(ldd), py, _i64(ldy), px, _i64(ldx), M, N, ptr(mean), ptr(invstd),
                                        ptr(gamma), act, pk, _i64(ldk), C.c_float(keep_scale),
                                        ptr(ws, torch.float64), ptr(dgamma), ptr(dbeta), ptr(dx_img, torch.bfloat16), int(T), int(pad),
                                        ptr(dbias), 1 if keep_f32 else 0, _stream()),
           "t2amd_bn_act_bwd_img_f32")


def colsum(x, ws, out, accumulate=False):
    lib = load()
    px, ldx, M, N = _mat(x)
    assert out.numel() == N
    _check(lib.t2amd_colsum_f32(px, _i64(ldx), M, N, ptr(ws, torch.float64), ptr(out), 1 if accumulate else 0,
                                _stream()), "t2amd_colsum_f32")


def colsum16(x16, ws, out, accumulate=False):
    """Column sums of a bf16 slab [M][N] (row stride in elements) into f32 ``out`` [N]."""
    lib = load()
    assert x16.dtype == torch.bfloat16 and x16.dim() == 2 and x16.stride(1) == 1 and out.numel() == x16.shape[1]
    _check(lib.t2amd_colsum_bf16(ptr(x16, torch.bfloat16), _i64(x16.stride(0)), x16.shape[0], x16.shape[1], ptr(ws, torch.float64),
                                 ptr(out), 1 if accumulate else 0, _stream()), "t2amd_colsum_bf16")


def embedding_fwd(ids, table, out):
    lib = load()
    rows = ids.numel()
    _check(lib.t2amd_embedding_fwd_f32(ptr(_fullc(ids), torch.int64), ptr(_fullc(table)), ptr(_fullc(out)),
                                       _i64(rows), table.shape[1], table.shape[0], _stream()),
           "t2amd_embedding_fwd_f32")


def embedding_bwd(ids, dout, dtable, ws=None):
    lib = load()
    rows = ids.numel()
    _check(lib.t2amd_embedding_bwd_f32(ptr(_fullc(ids), torch.int64), ptr(_fullc(dout)), ptr(_fullc(dtable)),
                                       None, _i64(rows), dtable.shape[1], dtable.shape[0], _stream()),
           "t2amd_embedding_bwd_f32")


def philox_keep_mask(out, p, seed, offset=0):
    lib = load()
    _check(lib.t2amd_philox_keep_mask(ptr(_fullc(out), torch.uint8), _i64(out.numel()), C.c_float(p),
                                      C.c_ulonglong(seed), C.c_ulonglong(offset), _stream()),
           "t2amd_philox_keep_mask")


def fill(t, v):
    lib = load()
    _check(lib.t2amd_fill_f32(ptr(_fullc(t)), _i64(t.numel()), C.c_float(v), _stream()), "t2amd_fill_f32")


def copy2d(dst, src, src2=None):
    lib = load()
    pd, ldd, R, Cc = _mat(dst)
    ps, lds, R1, C1 = _mat(src)
    assert (R, Cc) == (R1, C1), (dst.shape, src.shape)
    p2, ld2 = (None, 0)
    if src2 is not None:
        p2, ld2, R2, C2 = _mat(src2)
        assert (R, Cc) == (R2, C2)
    _check(lib.t2amd_copy2d_f32(ps, _i64(lds), p2, _i64(ld2), pd, _i64(ldd), R, Cc, _stream()), "t2amd_copy2d_f32")


def transpose(dst, src, batch=1, sstride=0, dstride=0):
    """dst[c][r] = src[r][c] for 2-D views (per batch item)."""
    lib = load()
    ps, lds, R, Cc = _mat(src)
    pd, ldd, R1, C1 = _mat(dst)
    assert (R1, C1) == (Cc, R), (dst.shape, src.shape)
    _check(lib.t2amd_transpose_f32(ps, _i64(lds), pd, _i64(ldd), R, Cc, batch, _i64(sstride), _i64(dstride),
                                   _stream()), "t2amd_transpose_f32")


def frames_to_time_major(mels, x0):
    lib = load()
    B, Cm, To = mels.shape
    assert tuple(x0.shape) == (To, B, Cm)
    _check(lib.t2amd_frames_to_time_major_f32(ptr(_fullc(mels)), ptr(_fullc(x0)), B, Cm, To, _stream()),
           "t2amd_frames_to_time_major_f32")


def split_projection(pg, mel_cl, gate, out_lens):
    lib = load()
    B, To, Cm = mel_cl.shape
    assert pg.numel() == To * B * (Cm + 1)
    _check(lib.t2amd_split_projection_f32(ptr(_fullc(pg)), ptr(_fullc(mel_cl)), ptr(_fullc(gate)),
                                          ptr(out_lens, torch.int32), B, Cm, To, _stream()),
           "t2amd_split_projection_f32")


def finalize_outputs(mel_cl, post_cl, mel, mel_post, out_lens):
    lib = load()
    B, To, Cm = mel_cl.shape
    _check(lib.t2amd_finalize_outputs_f32(ptr(_fullc(mel_cl)), ptr(post_cl), ptr(_fullc(mel)), ptr(mel_post),
                                          ptr(out_lens, torch.int32), B, Cm, To, _stream()),
           "t2amd_finalize_outputs_f32")


def grads_to_channel_last(dmel, dmel_post, dmel_cl, dpost_cl):
    lib = load()
    B, To, Cm = dmel_cl.shape
    _check(lib.t2amd_grads_to_channel_last_f32(ptr(dmel), ptr(dmel_post), ptr(_fullc(dmel_cl)),
                                               ptr(_fullc(dpost_cl)), B, Cm, To, _stream()),
           "t2amd_grads_to_channel_last_f32")


def gather_dout(dmel_cl, dgate, dout):
    lib = load()
    B, To, Cm = dmel_cl.shape
    _check(lib.t2amd_gather_dout_f32(ptr(_fullc(dmel_cl)), ptr(dgate), ptr(_fullc(dout)), B, Cm, To, _stream()),
           "t2amd_gather_dout_f32")


def relu_dropout_bwd(dy, y, scale):
    lib = load()
    assert dy.shape == y.shape
    _check(lib.t2amd_relu_dropout_bwd_f32(ptr(_fullc(dy)), ptr(_fullc(y)), C.c_float(scale), _i64(dy.numel()),
                                          _stream()), "t2amd_relu_dropout_bwd_f32")


# ----------------------------------------------------------------------------
# recurrent / attention single steps (used by the unit tests; the loops call them in C)
# ----------------------------------------------------------------------------
def _seg(t, width, dtype=torch.float32, x3=False):
    """``x3``: t is a split-bf16 image (split_bf16x3: 2 bf16 per k); width and ld count k, as for f32."""
    s = Seg()
    if t is None:
        s.p, s.ld, s.width = None, width, width
    else:
        p, ld, _, cols = _mat(t, dtype)
        if x3:
            if cols != 2 * width or ld % 2:
                raise NativeError("split-bf16 segment: expected %d columns (2 per k), got %d (ld %d)" % (2 * width, cols, ld))
            ld //= 2
        else:
            assert cols == width
        s.p, s.ld, s.width = p, ld, width
    return s


def lstm_step_fwd(xs, widths, W, H, B, gates_out, c_out, h_out, gin=None, bias=None, c_prev=None,
                  keep=None, keep_scale=1.0, lens=None, t=0, small=False, bf16=False, h16_out=None):
    """``bf16=True``: xs and W are torch.bfloat16 (MFMA operands), everything else stays f32.  ``bf16=3``: xs, W and h16_out are
    split-bf16 images (split_bf16x3: bfloat16 tensors with two columns per k) -- the 'bf16x3' mode of the wide tile."""
    lib = load()
    a = LstmStep()
    a.nseg = len(xs)
    x3 = bf16 == 3 and bf16 is not True
    odt = torch.bfloat16 if bf16 else torch.float32
    for i, (x, w) in enumerate(zip(xs, widths)):
        a.x[i] = _seg(x, w, odt, x3)
    a.W = ptr(_fullc(W), odt)
    a.bf16 = 3 if x3 else (1 if bf16 else 0)
    if h16_out is not None:
        a.h16_out, a.ld_h16 = _mat(h16_out, torch.bfloat16)[:2]
        if x3:
            a.ld_h16 //= 2
    a.Ktot, a.H, a.B = sum(widths), H, B
    if gin is not None:
        a.gin, a.ld_gin = _mat(gin)[:2]
    a.bias = ptr(bias)
    if c_prev is not None:
        a.c_prev, a.ld_cprev = _mat(c_prev)[:2]
    a.gates_out, a.ld_gates = _mat(gates_out)[:2]
    a.c_out, a.ld_c = _mat(c_out)[:2]
    a.h_out, a.ld_h = _mat(h_out)[:2]
    if keep is not None:
        a.keep, a.ld_keep, a.keep_scale = ptr(keep, torch.uint8), keep.stride(0), keep_scale
    a.lens = ptr(lens, torch.int32)
    a.t = t
    if small:
        _check(lib.t2amd_lstm_step_small_f32(C.byref(a), _stream()), "t2amd_lstm_step_small_f32")
    else:
        _check(lib.t2amd_lstm_step_fwd_f32(C.byref(a), _stream()), "t2amd_lstm_step_fwd_f32")


def linear_small(X, W, Y, bias=None, act=0, keep=None, keep_scale=1.0):
    """Y[B,N] = act(X[B,K] . W[N,K]^T + bias) * keep, B <= 8 (matrix-vector kernel)."""
    lib = load()
    a = SmallLinear()
    a.X, a.ldx, Bx, K = _mat(X)
    a.W, a.ldw, N, Kw = _mat(W)
    a.Y, a.ldy, By, Ny = _mat(Y)
    if K != Kw or N != Ny or Bx != By:
        raise NativeError("linear_small: shape mismatch")
    a.B, a.N, a.K, a.act = Bx, N, K, act
    a.bias = ptr(bias)
    if keep is not None:
        a.keep, a.ldkeep, a.keep_scale = ptr(keep, torch.uint8), keep.stride(0), keep_scale
    _check(lib.t2amd_linear_small_f32(C.byref(a), _stream()), "t2amd_linear_small_f32")


def skinny_gemm(xs, widths, W, N, B, Y, nsplit=1, bf16=False, bias=None, act=0, keep=None, keep_scale=1.0):
    """Y[nsplit, B, N] = [xs...] . W[N, K]^T   (``bf16=3``: xs and W are split-bf16 images, see lstm_step_fwd)"""
    lib = load()
    a = SkinnyGemm()
    a.nseg = len(xs)
    x3 = bf16 == 3 and bf16 is not True
    odt = torch.bfloat16 if bf16 else torch.float32
    for i, (x, w) in enumerate(zip(xs, widths)):
        a.x[i] = _seg(x, w, odt, x3)
    a.W = ptr(_fullc(W), odt)
    a.bf16 = 3 if x3 else (1 if bf16 else 0)
    a.Ktot, a.N, a.B = sum(widths), N, B
    _fullc(Y)
    a.Y, a.ldy, a.nsplit, a.split_stride = ptr(Y), N, nsplit, B * N
    if bias is not None:
        a.bias = ptr(_fullc(bias))
    a.act = int(act)
    if keep is not None:
        a.keep, a.ld_keep, a.keep_scale = ptr(_fullc(keep), torch.uint8), keep.stride(0), keep_scale
    _check(lib.t2amd_skinny_gemm_f32(C.byref(a), _stream()), "t2amd_skinny_gemm_f32")


def _addend(t, nsplit=1, split_stride=0):
    a = Addend()
    if t is None:
        a.p, a.ld, a.nsplit, a.split_stride = None, 0, 1, 0
    else:
        p, ld, _, _ = _mat(t)
        a.p, a.ld, a.nsplit, a.split_stride = p, ld, nsplit, split_stride
    return a


def lstm_bwd_desc(B, H, dh_list, gates, c_prev, c, keep, keep_scale, dc, dgates, lens=None, t=0, dgates16=None, x3=False):
    """One cell-backward descriptor.  dh_list entries: a tensor, None, or (tensor, nsplit, split_stride) for an addend
    made of partial slabs."""
    a = LstmBwd()
    a.B, a.H = B, H
    for i in range(3):
        d = dh_list[i] if i < len(dh_list) else None
        a.dh[i] = _addend(*d) if isinstance(d, tuple) else _addend(d)
    a.gates, a.ld_gates = _mat(gates)[:2]
    if c_prev is not None:
        a.c_prev, a.ld_cprev = _mat(c_prev)[:2]
    a.c, a.ld_c = _mat(c)[:2]
    if keep is not None:
        a.keep, a.ld_keep, a.keep_scale = ptr(keep, torch.uint8), keep.stride(0), keep_scale
    a.dc, a.ld_dc = _mat(dc)[:2]
    a.dgates, a.ld_dgates = _mat(dgates)[:2]
    a.lens = ptr(lens, torch.int32)
    a.t = t
    if dgates16 is not None:
        a.dgates16, a.ld_dgates16 = ptr(dgates16, torch.bfloat16), dgates16.stride(0)
        if x3:                           # split-bf16 image of the gate gradients: [B][2 * 4H] bf16, ld in k
            a.ld_dgates16, a.dgates16_x3 = dgates16.stride(0) // 2, 1
    return a


def lstm_pointwise_bwd2(a, b=None):
    """The stand-alone launch for one or two descriptors of lstm_bwd_desc()."""
    _check(load().t2amd_lstm_pointwise_bwd2_f32(C.byref(a), C.byref(b) if b is not None else None, _stream()),
           "t2amd_lstm_pointwise_bwd2_f32")


def lstm_pointwise_bwd(B, H, dh_list, gates, c_prev, c, keep, keep_scale, dc, dgates, lens=None, t=0):
    lib = load()
    a = LstmBwd()
    a.B, a.H = B, H
    for i in range(3):
        a.dh[i] = _addend(dh_list[i] if i < len(dh_list) else None)
    a.gates, a.ld_gates = _mat(gates)[:2]
    if c_prev is not None:
        a.c_prev, a.ld_cprev = _mat(c_prev)[:2]
    a.c, a.ld_c = _mat(c)[:2]
    if keep is not None:
        a.keep, a.ld_keep, a.keep_scale = ptr(keep, torch.uint8), keep.stride(0), keep_scale
    a.dc, a.ld_dc = _mat(dc)[:2]
    a.dgates, a.ld_dgates = _mat(dgates)[:2]
    a.lens = ptr(lens, torch.int32)
    a.t = t
    _check(lib.t2amd_lstm_pointwise_bwd_f32(C.byref(a), _stream()), "t2amd_lstm_pointwise_bwd_f32")


def fold_location(wdense, wconv, U):
    """U buffer: 128*62 floats."""
    lib = load()
    assert U.numel() == ATT_DIM * LOC_TAPS
    assert tuple(wdense.shape) == (ATT_DIM, LOC_FILTERS) and tuple(wconv.shape) == (LOC_FILTERS, 2, LOC_KERNEL)
    _check(lib.t2amd_fold_location_f32(ptr(_fullc(wdense)), ptr(_fullc(wconv)), ptr(_fullc(U)), _stream()),
           "t2amd_fold_location_f32")


def unfold_location_grads(dU_acc, dv_acc, nb, wdense, wconv, dwdense, dwconv, dv):
    lib = load()
    _check(lib.t2amd_unfold_location_grads_f32(ptr(_fullc(dU_acc)), ptr(_fullc(dv_acc)), nb, ptr(_fullc(wdense)),
                                               ptr(_fullc(wconv)), ptr(_fullc(dwdense)), ptr(_fullc(dwconv)),
                                               ptr(_fullc(dv)), _stream()), "t2amd_unfold_location_grads_f32")


def attn_fwd_ws_floats(B, Ti):
    """partial energies + the granule block of the one-launch form (t2amd_attn_fwd_ws_floats; a multiple of 4)."""
    return int(load().t2amd_attn_fwd_ws_floats(int(B), int(Ti)))


_attn_fused_sel = {"fwd": -1, "bwd": -1}     # what this process last selected (-1: the library's default / environment)


def set_attn_fwd_fused(on):
    """K_e and K_c of an attention step as one launch (1), two launches (0), or the library default (-1)."""
    _check(load().t2amd_set_attn_fwd_fused(int(on)), "t2amd_set_attn_fwd_fused")
    _attn_fused_sel["fwd"] = int(on)


def get_attn_fwd_fused():
    """The last selection made through set_attn_fwd_fused (-1: none; the library default)."""
    return _attn_fused_sel["fwd"]


def get_attn_bwd_fused():
    return _attn_fused_sel["bwd"]


def attn_bwd_ws_floats(B, Ti):
    """dw slab + slice partials + the token blocks of the two hand-offs + the granule block (t2amd_attn_bwd_ws_floats)."""
    return int(load().t2amd_attn_bwd_ws_floats(int(B), int(Ti)))


def set_attn_bwd_granules(on):
    """First hand-off of the one-launch attention backward: 1 granules, 0 drained stores + token, -1 library default."""
    _check(load().t2amd_set_attn_bwd_granules(int(on)), "t2amd_set_attn_bwd_granules")


def attn_handoff_timeouts(reset=True):
    """Abandoned in-launch hand-offs of the one-launch attention forms since the last reset (synchronises)."""
    n = load().t2amd_attn_handoff_timeouts(1 if reset else 0)
    if n < 0:
        raise NativeError("t2amd_attn_handoff_timeouts failed")
    return n


def set_attn_bwd_fused(on):
    """1: the attention backward of a step as one launch (default), 0: K_b1 / K_b2 as separate launches, -1: default."""
    _check(load().t2amd_set_attn_bwd_fused(int(on)), "t2amd_set_attn_bwd_fused")
    _attn_fused_sel["bwd"] = int(on)


def attention_step_fwd(h, Wq, U, v, pm, memory, lens, w_prev, cum, cum_save, w_out, ctx_out, q_out, ws, active=None,
                       bf16=False, memory16=None, Wq16=None, ctx_x3_out=None):
    """``ctx_x3_out``: a [B][2 E] bfloat16 tensor that receives the split-bf16 image of the context ('bf16x3' mode)."""
    lib = load()
    a = AttnFwd()
    B, Ti, E = memory.shape
    a.B, a.Ti, a.E, a.Hq = B, Ti, E, h.shape[1]
    a.h, a.ld_h = _mat(h)[:2]
    a.Wq, a.U, a.v, a.pm, a.memory = ptr(_fullc(Wq)), ptr(U), ptr(v), ptr(_fullc(pm)), ptr(_fullc(memory))
    a.lens = ptr(lens, torch.int32)
    if w_prev is not None:
        a.w_prev, a.ld_wprev = _mat(w_prev)[:2]
    a.cum = ptr(_fullc(cum))
    a.cum_save = ptr(cum_save)
    a.w_out, a.ld_wout = _mat(w_out)[:2]
    a.ctx_out, a.ld_ctx = _mat(ctx_out)[:2]
    if q_out is not None:
        a.q_out, a.ld_q = _mat(q_out)[:2]
    a.active = ptr(active, torch.uint8)
    if ws.numel() < attn_fwd_ws_floats(B, Ti):
        raise NativeError("attention_step_fwd: workspace too small")
    a.ws = ptr(_fullc(ws))
    a.ws_floats = ws.numel()
    a.loc_split_bf16 = 1 if bf16 else 0
    if memory16 is not None:
        a.memory16 = ptr(_fullc(memory16), torch.bfloat16)
    if Wq16 is not None:
        a.Wq16 = ptr(_fullc(Wq16), torch.bfloat16)
    if ctx_x3_out is not None:
        if tuple(ctx_x3_out.shape) != (B, 2 * E) or E % 16:
            raise NativeError("attention_step_fwd: ctx_x3_out must be [B][2 E] bfloat16, E %% 16 == 0")
        a.ctx16_out, a.ld_ctx16, a.ctx16_x3 = ptr(_fullc(ctx_x3_out), torch.bfloat16), E, 1
    _check(lib.t2amd_attention_step_fwd_f32(C.byref(a), _stream()), "t2amd_attention_step_fwd_f32")


def attention_step_bwd(dctx_list, dctx_total, d_w_extra, q, Wq, U, v, pm, memory, lens, w, w_prev, cum_before,
                       dwin_part, dcum_acc, d_pm, dU_acc, dv_acc, dq_out, dh_parts, ws, bf16=False, memory16=None,
                       cell_q=None, cell_x=None, Wq16=None):
    """dwin_part: (ATT_SLICES, B, 2, Ti) in/out; dcum_acc: (B, Ti) in/out; dh_parts: (ATT_SLICES, B, Hq) out.
    cell_q / cell_x: descriptors of lstm_bwd_desc() to run inside this launch (t2amd_attn_bwd.cell_q / cell_x)."""
    lib = load()
    a = AttnBwd()
    B, Ti, E = memory.shape
    a.B, a.Ti, a.E, a.Hq = B, Ti, E, Wq.shape[1]
    for i in range(3):
        a.dctx[i] = _addend(dctx_list[i] if i < len(dctx_list) else None)
    a.dctx_total, a.ld_dctx_total = _mat(dctx_total)[:2]
    if d_w_extra is not None:
        a.d_w_extra, a.ld_dwextra = _mat(d_w_extra)[:2]
    a.q, a.ld_q = _mat(q)[:2]
    a.Wq, a.U, a.v, a.pm, a.memory = ptr(_fullc(Wq)), ptr(U), ptr(v), ptr(_fullc(pm)), ptr(_fullc(memory))
    a.lens = ptr(lens, torch.int32)
    a.w, a.ld_w = _mat(w)[:2]
    if w_prev is not None:
        a.w_prev, a.ld_wprev = _mat(w_prev)[:2]
    a.cum_before = ptr(_fullc(cum_before))
    if tuple(dwin_part.shape) != (ATT_SLICES, B, 2, Ti) or tuple(dh_parts.shape) != (ATT_SLICES, B, Wq.shape[1]):
        raise NativeError("attention_step_bwd: dwin_part / dh_parts have the wrong shape")
    a.dwin_part, a.dcum_acc = ptr(_fullc(dwin_part)), ptr(_fullc(dcum_acc))
    a.d_pm, a.dU_acc, a.dv_acc = ptr(_fullc(d_pm)), ptr(_fullc(dU_acc)), ptr(_fullc(dv_acc))
    a.dq_out, a.ld_dq = _mat(dq_out)[:2]
    a.dh_out, a.ld_dh, a.dh_split_stride = ptr(_fullc(dh_parts)), Wq.shape[1], B * Wq.shape[1]
    if ws.numel() < attn_bwd_ws_floats(B, Ti):
        raise NativeError("attention_step_bwd: workspace too small")
    a.ws = ptr(_fullc(ws))
    a.ws_floats = ws.numel()
    a.bf16 = 1 if bf16 else 0
    if memory16 is not None:
        a.memory16 = ptr(_fullc(memory16), torch.bfloat16)
    if cell_q is not None:
        a.cell_q = C.pointer(cell_q)
    if cell_x is not None:
        a.cell_x = C.pointer(cell_x)
    if Wq16 is not None:
        a.Wq16 = ptr(_fullc(Wq16), torch.bfloat16)
    _check(lib.t2amd_attention_step_bwd_f32(C.byref(a), _stream()), "t2amd_attention_step_bwd_f32")


# ----------------------------------------------------------------------------
# loops
# ----------------------------------------------------------------------------
# The loop-level entry points are also registered with the PyTorch dispatcher (csrc/torch_ops.cpp, TORCH_LIBRARY over
# the same C ABI: torch.ops.tacotron2_amd.*).  When that library is present the engine's loops go through it -- the
# descriptor travels as a CPU uint8 tensor, the tensors it points into as the op's `reads` / `writes` lists -- otherwise
# (and for the instrumented T2AMD_LIB builds, the validate-only CPU runs, and T2AMD_TORCH_OPS=0) through ctypes.
_torch_ops = None


def torch_ops():
    """torch.ops.tacotron2_amd, or None when the registration library is absent / disabled."""
    global _torch_ops
    if _torch_ops is None:
        ok = (os.environ.get("T2AMD_TORCH_OPS", "1") != "0" and "T2AMD_LIB" not in os.environ
              and os.path.exists(TORCH_OPS_PATH))
        if ok:
            load()                                    # the C ABI library first: the ops library resolves against it
            torch.ops.load_library(TORCH_OPS_PATH)
            _torch_ops = torch.ops.tacotron2_amd
        else:
            _torch_ops = False
    return _torch_ops or None


def _desc_tensor(desc):
    return torch.frombuffer(desc, dtype=torch.uint8)          # zero-copy view of the ctypes struct


def _dev_tensors(ts):
    return [t for t in ts if t is not None and torch.is_tensor(t) and t.is_cuda]


def _via_ops(name, descs, reads, writes):
    """Run loop `name` through the dispatcher when possible; False -> the caller takes the ctypes route."""
    ops = None if _validate_only else torch_ops()
    if ops is None or reads is None or writes is None:
        return False
    w = _dev_tensors(writes)
    if not w:
        return False
    getattr(ops, name)(*[_desc_tensor(d) for d in descs], _dev_tensors(reads), w)
    return True


last_loop_route = None          # 'torch.ops' / 'ctypes': which way the most recent loop-level call went (tests)


def _loop(name, cname, descs, reads, writes):
    global last_loop_route
    if _via_ops(name, descs, reads, writes):
        last_loop_route = 'torch.ops'
        return
    last_loop_route = 'ctypes'
    _check(getattr(load(), cname)(*[C.byref(d) for d in descs], _stream()), cname)


def decoder_train_fwd_loop(desc, reads=None, writes=None):
    _loop("decoder_train_fwd", "t2amd_decoder_train_fwd_loop_f32", [desc], reads, writes)


def decoder_train_bwd_loop(desc, reads=None, writes=None):
    _loop("decoder_train_bwd", "t2amd_decoder_train_bwd_loop_f32", [desc], reads, writes)


def lstm_seq_fwd(desc):
    lib = load()
    _check(lib.t2amd_lstm_seq_fwd_f32(C.byref(desc), _stream()), "t2amd_lstm_seq_fwd_f32")


def lstm_seq_fwd2(d0, d1, reads=None, writes=None):
    _loop("encoder_lstm_fwd", "t2amd_lstm_seq_fwd2_f32", [d0, d1], reads, writes)


def lstm_seq_persistent_supported(desc):
    """None when the encoder bi-LSTM of this geometry can run as one persistent launch, else the reason."""
    lib = load()
    if lib.t2amd_lstm_seq_persistent_supported(C.byref(desc)) == 0:
        return None
    msg = lib.t2amd_last_error()
    return msg.decode() if msg else "unsupported"


def lstm_seq_fwd2_persistent(d0, d1, mailbox, status):
    """Both directions of a single-utterance encoder bi-LSTM as one persistent launch (csrc/decode_persist.hip)."""
    need = load().t2amd_lstm_seq_persistent_mailbox_bytes(d0.H, 2)
    if mailbox.numel() * mailbox.element_size() < need:
        raise NativeError("lstm_seq_fwd2_persistent: mailbox of %d bytes, %d needed" % (mailbox.numel() * mailbox.element_size(), need))
    _check(load().t2amd_lstm_seq_fwd2_persistent_f32(C.byref(d0), C.byref(d1), ptr(mailbox, torch.int64), ptr(status, torch.int32),
                                                     _stream()), "t2amd_lstm_seq_fwd2_persistent_f32")


def decoder_train_fwd_persistent_supported(desc, cus):
    """None when the teacher-forced decoder loop can run as ONE persistent launch on a device of `cus` CUs, else the reason."""
    lib = load()
    if lib.t2amd_decoder_train_fwd_persistent_supported(C.byref(desc), int(cus)) == 0:
        return None
    msg = lib.t2amd_last_error()
    return msg.decode() if msg else "unsupported"


def decoder_train_fwd_persistent_flag_words(B, Ha):
    return int(load().t2amd_decoder_train_fwd_persistent_flag_bytes(int(B), int(Ha))) // 4


def decoder_train_fwd_persistent(desc, flags, status, poison=None):
    """reference model.py:405-411 (the teacher-forced loop) as one persistent launch (csrc/attention.hip,
    dec_train_fwd_persistent_kernel): bit-identical to decoder_train_fwd_loop.  ``poison``: an f32 tensor whose first element
    becomes NaN if a workgroup gave up (for callers that do not read ``status`` back)."""
    lib = load()
    need = lib.t2amd_decoder_train_fwd_persistent_flag_bytes(desc.B, desc.Ha)
    if flags.dtype != torch.int32 or flags.numel() * 4 < need:
        raise NativeError("decoder_train_fwd_persistent: int32 flags of %d bytes needed" % need)
    _check(lib.t2amd_decoder_train_fwd_persistent_f32(C.byref(desc), C.c_void_p(flags.data_ptr()), ptr(status, torch.int32),
                                                      ptr(poison) if poison is not None else None, _stream()),
           "t2amd_decoder_train_fwd_persistent_f32")


def lstm_seq_batch_persistent_supported(desc, ndir, cus):
    """None when the encoder bi-LSTM of this batch can run as one persistent launch on a device of `cus` CUs, else the reason."""
    lib = load()
    if lib.t2amd_lstm_seq_batch_persistent_supported(C.byref(desc), int(ndir), int(cus)) == 0:
        return None
    msg = lib.t2amd_last_error()
    return msg.decode() if msg else "unsupported"


def lstm_seq_batch_persistent_flag_words(B, H, ndir=2):
    return int(load().t2amd_lstm_seq_batch_persistent_flag_bytes(int(B), int(H), int(ndir))) // 4


def lstm_seq_fwd2_batch_persistent(d0, d1, flags, status, poison=None):
    """Both directions of the encoder bi-LSTM of a BATCH as one persistent launch (csrc/decode_persist.hip).  ``poison``: an
    f32 tensor whose first element becomes NaN if a workgroup gave up (for callers that do not read ``status`` back)."""
    lib = load()
    need = lib.t2amd_lstm_seq_batch_persistent_flag_bytes(d0.B, d0.H, 2)
    if flags.numel() * flags.element_size() < need:
        raise NativeError("lstm_seq_fwd2_batch_persistent: flags of %d bytes, %d needed" % (flags.numel() * flags.element_size(), need))
    _check(lib.t2amd_lstm_seq_fwd2_batch_persistent_f32(C.byref(d0), C.byref(d1), C.c_void_p(flags.data_ptr()), ptr(status, torch.int32),
                                                        ptr(poison) if poison is not None else None, _stream()),
           "t2amd_lstm_seq_fwd2_batch_persistent_f32")


def lstm_seq_bwd2_batch_persistent_supported(desc, ndir, cus):
    """None when the encoder bi-LSTM's BPTT of this batch can run as one persistent launch on a device of `cus` CUs, else the reason."""
    lib = load()
    if lib.t2amd_lstm_seq_bwd2_batch_persistent_supported(C.byref(desc), int(ndir), int(cus)) == 0:
        return None
    msg = lib.t2amd_last_error()
    return msg.decode() if msg else "unsupported"


def lstm_seq_bwd2_batch_persistent(d0, d1, flags, status, poison=None):
    """BPTT of both directions of the encoder bi-LSTM of a BATCH as one persistent launch (csrc/decode_persist.hip,
    encoder_bilstm_batch_persistent_bwd_kernel) instead of 2 T dependent launches (lstm_seq_bwd2)."""
    lib = load()
    need = lib.t2amd_lstm_seq_batch_persistent_flag_bytes(d0.B, d0.H, 2)
    if flags.numel() * flags.element_size() < need:
        raise NativeError("lstm_seq_bwd2_batch_persistent: flags of %d bytes, %d needed" % (flags.numel() * flags.element_size(), need))
    _check(lib.t2amd_lstm_seq_bwd2_batch_persistent_f32(C.byref(d0), C.byref(d1), C.c_void_p(flags.data_ptr()), ptr(status, torch.int32),
                                                        ptr(poison) if poison is not None else None, _stream()),
           "t2amd_lstm_seq_bwd2_batch_persistent_f32")


def encoder_handoff_timeouts(reset=True):
    """Give-ups of the batched persistent encoder launch since the last reset (synchronises)."""
    n = load().t2amd_encoder_handoff_timeouts(1 if reset else 0)
    if n < 0:
        raise NativeError("t2amd_encoder_handoff_timeouts failed")
    return n


def lstm_seq_bwd2(d0, d1, reads=None, writes=None):
    _loop("encoder_lstm_bwd", "t2amd_lstm_seq_bwd2_f32", [d0, d1], reads, writes)


def lstm_seq_bwd(desc):
    lib = load()
    _check(lib.t2amd_lstm_seq_bwd_f32(C.byref(desc), _stream()), "t2amd_lstm_seq_bwd_f32")


def decoder_infer_steps(desc, reads=None, writes=None):
    _loop("decoder_infer_steps", "t2amd_decoder_infer_steps_f32", [desc], reads, writes)


# ----------------------------------------------------------------------------
# mel front end (csrc/audio.hip)
# ----------------------------------------------------------------------------
def reflect_index(i, T):
    """Host copy of the kernel's reflect rule (no GPU needed)."""
    return int(load().t2amd_reflect_index(int(i), int(T)))


def reflect_pad(y, out, pad):
    """out[b][:T+2*pad] = reflect-padded y[b]; the rest of each out row is zeroed.  y (B,T), out (B,Tout)."""
    py, ldy, B, T = _mat(y)
    po, ldo, B1, Tout = _mat(out)
    if B1 != B:
        raise NativeError("reflect_pad: batch mismatch %s vs %s" % (tuple(y.shape), tuple(out.shape)))
    _check(load().t2amd_reflect_pad_f32(py, _i64(ldy), po, _i64(ldo), B, T, int(pad), Tout, _stream()),
           "t2amd_reflect_pad_f32")


def stft_magnitude(spec, mag, F):
    """mag[r][:F] = |spec[r][:F] + i spec[r][F:2F]|, mag[r][F:] = 0.  Chunked to the grid limit."""
    ps, lds, R, C2 = _mat(spec)
    pm, ldm, R1, Fpad = _mat(mag)
    if R1 != R or C2 < 2 * F or Fpad < F:
        raise NativeError("stft_magnitude: shape mismatch spec=%s mag=%s F=%d" % (tuple(spec.shape), tuple(mag.shape), F))
    lib = load()
    for r0 in range(0, R, 65535):
        rows = min(65535, R - r0)
        _check(lib.t2amd_stft_magnitude_f32(ptr(spec[r0:]), _i64(lds), ptr(mag[r0:]), _i64(ldm), _i64(rows), int(F),
                                            Fpad, _stream()), "t2amd_stft_magnitude_f32")


def mel_log_compress(mel, out, clip):
    """out (B, n_mel, n) = log(max(mel (B*n, n_mel), clip)) transposed per utterance."""
    pm, ld, R, n_mel = _mat(mel)
    B, n_mel1, n = out.shape
    if n_mel1 != n_mel or B * n != R:
        raise NativeError("mel_log_compress: shape mismatch mel=%s out=%s" % (tuple(mel.shape), tuple(out.shape)))
    _check(load().t2amd_mel_log_compress_f32(pm, _i64(ld), ptr(_fullc(out)), B, n, n_mel, C.c_float(clip), _stream()),
           "t2amd_mel_log_compress_f32")


# ----------------------------------------------------------------------------
# optimiser step (csrc/optim.hip)
# ----------------------------------------------------------------------------
def _raw(t):
    v = ptr(t)
    return None if v is None else v.value


def tensor_list(grads, params=None, exp_avgs=None, exp_avg_sqs=None):
    """-> (TensorList, total_blocks).  Every tensor must be contiguous f32 (views at any offset are fine)."""
    n = len(grads)
    if not 0 < n <= MAX_TENSORS:
        raise NativeError("tensor_list: 1..%d tensors per call, got %d" % (MAX_TENSORS, n))
    chunk = int(load().t2amd_optim_chunk())
    L = TensorList()
    blocks = 0
    for i in range(n):
        g = _fullc(grads[i])
        L.grad[i] = _raw(g)
        L.numel[i] = g.numel()
        L.first_block[i] = blocks
        blocks += (g.numel() + chunk - 1) // chunk
        for field, seq in (("param", params), ("exp_avg", exp_avgs), ("exp_avg_sq", exp_avg_sqs)):
            if seq is not None:
                t = _fullc(seq[i])
                if t.numel() != g.numel():
                    raise NativeError("tensor_list: %s[%d] has %d elements, gradient %d" % (field, i, t.numel(), g.numel()))
                getattr(L, field)[i] = _raw(t)
    L.count = n
    return L, blocks


def grad_norm(L, blocks, max_norm, ws, out):
    """out[0] = global L2 norm of the gradients of L, out[1] = clip coefficient.  ws: float64, >= blocks."""
    if ws.dtype != torch.float64 or ws.numel() < blocks or out.numel() < 2:
        raise NativeError("grad_norm: workspace too small")
    _check(load().t2amd_grad_norm_f32(C.byref(L), C.c_float(max_norm), ptr(ws, torch.float64), ptr(out), _stream()),
           "t2amd_grad_norm_f32")


def adam_step(L, hyper, norm_and_coef=None):
    _check(load().t2amd_adam_step_f32(C.byref(L), C.byref(hyper), ptr(norm_and_coef) if norm_and_coef is not None else None,
                                      _stream()), "t2amd_adam_step_f32")
