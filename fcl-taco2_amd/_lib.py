"""ctypes binding of libfcl_hip.so (include/fcl_hip.h).  Fails loudly when the library is missing.

This is the only place the package touches the shared library.  There is no CPU fallback: if the HIP
extension cannot be loaded, importing any compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FCL_LIB") or os.path.join(_HERE, "libfcl_hip.so")  # (FCL_LIB: developer A/B builds of the same ABI)

# name -> (restype, argtypes); mirrors include/fcl_hip.h one to one
_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t


class DecoderWeights(C.Structure):
    _fields_ = [("c", _I), ("p", _I), ("u", _I), ("odim", _I)] + [
        (n, _P) for n in ("prenet_w0", "prenet_b0", "prenet_w1", "prenet_b1", "w0_att", "w0_pre", "w0_pos", "w0_hh", "b0",
                          "w1_ih", "w1_hh", "b1", "wf_h", "wf_att")
    ] + [("zoneout_rate", _F), ("prenet_dropout", _F)] + [
        (n + sfx, _P) for n in ("prenet_w0", "prenet_w1", "w0_pre", "w0_hh", "w1_ih", "w1_hh", "wf_h") for sfx in ("_hi", "_lo")] + [
        (n + "_p", _P) for n in ("w0_att", "wf_att", "w0_pre", "w0_hh", "w1_ih", "w1_hh")] + [("out_act", _I), ("wf_h_ff", _P), ("prenet_w0_ff", _P), ("prenet_w1_ff", _P),
                                                                                           ("w0_pre_ff", _P), ("w0_hh_ff", _P), ("w1_ih_ff", _P), ("w1_hh_ff", _P), ("stream", _P),
                                                                                           ("prenet_layers", _I), ("dlayers", _I), ("prenet_w2", _P), ("prenet_b2", _P), ("w2_ih", _P),
                                                                                           ("w2_hh", _P), ("b2", _P), ("reduction_factor", _I)]


class DecoderIO(C.Structure):
    _fields_ = [("n", _I), ("lmax", _I), ("att_c", _P), ("dur", _P), ("live_rows_host", _P), ("frame_off", _P),
                ("teacher_ys", _P), ("dropout_mode", _I), ("prenet_keep", _P), ("seed", C.c_uint32), ("seed_dev", _P), ("before", _P),
                ("tap_prenet", _P), ("tap_lstm0", _P), ("tap_lstm1", _P), ("workspace", _P), ("workspace_bytes", _Z), ("att_c_p", _P), ("before_p", _P),
                ("live_rows", _P), ("status", _P), ("tail_from", _I)]


class RowMaps(C.Structure):  # fcl_row_maps_t
    _fields_ = [("b", _I), ("n", _I), ("lmax_cap", _I), ("frames_cap", _I), ("t_max", _I)] + [
        (n, _P) for n in ("row_src", "utt_row0", "pad", "dur_i64", "dur_i32", "src_rows", "dur_sorted", "frame_off", "order", "live_rows", "utt_frame0",
                          "frame_lo", "frame_hi", "totals", "status", "scratch")]


class GemmTerm(C.Structure):
    _fields_ = [("A", _P), ("W", _P), ("lda", _I), ("ldw", _I), ("K", _I), ("shift", _I), ("Whi", _P), ("Wlo", _P), ("Ap", _P), ("Wp", _P),
                ("lda_p", _I), ("ldw_p", _I), ("a_chunk_stride", C.c_int64), ("Wff", _P)]


class LstmStep(C.Structure):
    _fields_ = [("term", GemmTerm * 3), ("nterms", _I), ("M", _I), ("U", _I), ("G", _P), ("g_row_mul", C.c_longlong), ("g_row_add", C.c_longlong),
                ("bias", _P), ("rank1_w", _P), ("dur", _P), ("step", _I), ("h_in", _P), ("h_out", _P), ("c", _P), ("zoneout", _F),
                ("zone_keep_h", _P), ("zone_keep_c", _P), ("row_len", _P), ("out2", _P), ("out2_row_base", _P), ("out2_row_mul", C.c_longlong),
                ("out2_row_add", C.c_longlong), ("ld2", _I), ("out2_col_off", _I), ("h_out_p", _P), ("ld_hp", _I), ("save_gates", _P), ("save_c_new", _P), ("save_c_old", _P),
                ("save_h_old", _P), ("m_dev", _P)]


class DecoderTrain(C.Structure):  # fcl_decoder_train_t
    _fields_ = [("n", _I), ("lmax", _I), ("u", _I), ("p", _I), ("live_rows_host", _P), ("p1d", _P), ("g0", _P), ("w0_pre", _P), ("w0_hh", _P),
                ("w0_pos", _P), ("dur", _P), ("w1_ih", _P), ("w1_hh", _P), ("b1", _P), ("zoneout", _F), ("zk_h0", _P), ("zk_c0", _P), ("zk_h1", _P),
                ("zk_c1", _P), ("s0", _P * 4), ("s1", _P * 4), ("h0_all", _P), ("h1_all", _P), ("workspace", _P), ("workspace_bytes", _Z),
                ("p1d_p", _P), ("w0_pre_p", _P), ("w0_hh_p", _P), ("w1_ih_p", _P), ("w1_hh_p", _P)]


class DecoderBptt(C.Structure):  # fcl_decoder_bptt_t
    _fields_ = [("n", _I), ("lmax", _I), ("u", _I), ("live_rows_host", _P), ("s0", _P * 3), ("s1", _P * 3), ("zoneout", _F), ("zk_h0", _P),
                ("zk_c0", _P), ("zk_h1", _P), ("zk_c1", _P), ("dh1_all", _P), ("dh0_all", _P), ("w1_ih_t", _P), ("w1_hh_t", _P), ("w0_hh_t", _P),
                ("dg0_all", _P), ("dg1_all", _P), ("workspace", _P), ("workspace_bytes", _Z), ("w1_ih_t_p", _P), ("w1_hh_t_p", _P), ("w0_hh_t_p", _P),
                ("dg0_all_p", _P), ("dg1_all_p", _P), ("w1_cat_t", _P), ("w1_cat_t_p", _P)]


class BilstmTrain(C.Structure):  # fcl_bilstm_train_t
    _fields_ = [("b", _I), ("t", _I), ("h", _I), ("lens", _P), ("gx", _P * 2), ("w_hh", _P * 2), ("out", _P), ("s", (_P * 4) * 2), ("workspace", _P),
                ("workspace_bytes", _Z), ("status", _P)]


class BilstmBptt(C.Structure):  # fcl_bilstm_bptt_t
    _fields_ = [("b", _I), ("t", _I), ("h", _I), ("lens", _P), ("s", (_P * 3) * 2), ("d_out", _P), ("ld_dout", _I), ("w_hh_t", _P * 2), ("dg", _P * 2),
                ("workspace", _P), ("workspace_bytes", _Z), ("status", _P)]


GEMM_F32, GEMM_BF16 = 0, 1


class Derive(C.Structure):  # fcl_derive_t
    _fields_ = [("src", _P), ("src2", _P), ("dst", _P), ("dst_p", _P), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("sa", C.c_int32),
                ("sb", C.c_int32), ("sc", C.c_int32), ("first_block", C.c_int32), ("reserved", C.c_int32)]


ABI_VERSION = 422  # FCL_ABI_VERSION of include/fcl_hip.h


class PwgLayer(C.Structure):  # fcl_pwg_layer_t
    _fields_ = [("m", C.c_int64), ("r", C.c_int32), ("aux", C.c_int32), ("ksize", C.c_int32), ("dilation", C.c_int32), ("first_layer", C.c_int32)] + [
        (n, _P) for n in ("seg_lo", "seg_hi", "x", "xp", "cp", "w_conv_p", "b_conv", "w_aux_p", "w_os_p", "b_os", "skips", "z", "gp", "o", "xp_out",
                          "kp", "pt_a", "pt_b")] + [("ld_pt", C.c_int32), ("hop", C.c_int32)]


class BernoulliSite(C.Structure):  # fcl_bernoulli_site_t
    _fields_ = [("out", _P), ("n", C.c_int64), ("p_one", _F), ("seed", C.c_uint32)]


BERNOULLI_MAX_SITES = 16


class LossTerm(C.Structure):  # fcl_loss_term_t
    _fields_ = [(n, _P) for n in ("a", "b", "b2", "valid", "valid2", "da", "da_planes", "sums", "sums2")] + [
        ("m", C.c_int32), ("c", C.c_int32), ("b_log", C.c_int32), ("b_log_offset", _F), ("w_l1", _F), ("w_mse", _F), ("count", C.c_double),
        ("w_l1_2", _F), ("w_mse_2", _F), ("count2", C.c_double)]


LOSS_MAX_TERMS, SUM_ROWS_MAX = 12, 6


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 56), ("launches", _I), ("ms", C.c_double), ("flops", C.c_double), ("rows", C.c_double), ("fill_bytes", C.c_double)]


TE_MAX_SITES, TE_MAX_LOSSES = 48, 48
TE_TEACHER, TE_KD_TEACHER, TE_STUDENT = 0, 1, 2


class TeConfig(C.Structure):  # fcl_te_config_t
    _fields_ = ([(n, C.c_int32) for n in ("role", "idim", "odim", "embed_dim", "econv_layers", "econv_chans", "econv_filts", "eunits", "dunits", "prenet_units",
                                          "postnet_layers", "postnet_chans", "postnet_filts", "dp_layers", "dp_chans", "dp_kernel", "vp_layers", "vp_chans",
                                          "vp_kernel", "ve_kernel")]
                + [(n, _F) for n in ("dropout_rate", "zoneout_rate", "dp_dropout", "vp_dropout", "ve_dropout")]
                + [(n, C.c_int32) for n in ("use_masking", "t_embed_dim", "t_econv_chans", "t_eunits", "t_prenet_units", "t_dunits", "t_postnet_chans", "share_proj",
                                            "distill_output", "distill_encoder", "distill_decoder", "distill_prosody", "accum_grad")]
                + [("seed", C.c_uint32), ("site_tag", C.c_uint32 * TE_MAX_SITES), ("dw_planes_min", C.c_int32), ("pred_stream", C.c_int32),
                   ("late_losses", C.c_int32)])


class TeBatch(C.Structure):  # fcl_te_batch_t
    _fields_ = ([(n, C.c_int32) for n in ("B", "T", "L", "N", "F", "lmax")]
                + [(n, _P) for n in ("xs", "ys", "f0", "energy", "ds", "lens", "e_lo", "e_hi", "f_lo", "f_hi", "src_sorted", "row_of_enc", "cell_frame", "frame_cell",
                                     "prev_frame", "cell_row", "dur", "perm_tb", "cell_row_i64", "enc_pad", "enc_valid", "frame_valid", "cell_valid", "pos4",
                                     "live_rows_host")]
                + [("n_enc", C.c_double), ("n_frames", C.c_double)])


class TeKnowledge(C.Structure):  # fcl_te_knowledge_t
    _fields_ = [("after", _P), ("before", _P), ("enc", _P * 5), ("dec", _P * 8), ("pro", _P * 5), ("dec_cell_major", C.c_int32)]


SIGNATURES = {
    "fcl_te_create": (_I, [C.POINTER(TeConfig), C.POINTER(_P)]),
    "fcl_te_destroy": (None, [_P]),
    "fcl_te_site_name": (C.c_char_p, [_I]),
    "fcl_te_loss_name": (C.c_char_p, [_I]),
    "fcl_te_bind_param": (_I, [_P, C.c_char_p, _P, _P, C.c_int64]),
    "fcl_te_bind_buffer": (_I, [_P, C.c_char_p, _P]),
    "fcl_te_finalize": (_I, [_P, _P]),
    "fcl_te_params_changed": (_I, [_P]),
    "fcl_te_side_stream": (_P, [_P]),
    "fcl_te_place_streams": (_I, [_P, _P, C.POINTER(C.c_int)]),
    "fcl_te_knowledge": (_I, [_P, C.POINTER(TeBatch), C.c_uint32, C.POINTER(TeKnowledge), _P]),
    "fcl_te_forward_backward": (_I, [_P, C.POINTER(TeBatch), C.POINTER(TeKnowledge), C.c_uint32, _P, _P, _P]),
    "fcl_te_backward_stage": (_I, [_P, _I, _P]),
    "fcl_te_join": (_I, [_P, _P]),
    "fcl_te_last_launches": (C.c_int64, [_P]),
    "fcl_te_phase_ms": (_I, [_P, _P]),
    "fcl_te_arena_bytes": (C.c_int64, [_P]),
    "fcl_last_error": (C.c_char_p, []),
    "fcl_version": (_I, []),
    "fcl_debug_ptr": (_P, []),
    "fcl_set_gemm_mode": (_I, [_I]),
    "fcl_get_gemm_mode": (_I, []),
    "fcl_pack_conv1d_weight": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "fcl_fold_batchnorm": (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _P]),
    "fcl_copy2d": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "fcl_add2d": (_I, [_P, _I, _P, _I, _I, _I, _F, _P, _P]),
    "fcl_frag_bf16_elems": (_Z, [_I, _I]),
    "fcl_pack_frag_bf16": (_I, [_P, _I, _I, _P, _P, _P]),
    "fcl_pack_frag_f32": (_I, [_P, _I, _I, _P, _P]),
    "fcl_add_vec": (_I, [_P, _P, _P, _I, _P]),
    "fcl_u32_add": (_I, [_P, C.c_uint32, _P]),
    "fcl_planes_elems": (_Z, [_I, _I]),
    "fcl_pack_planes": (_I, [_P, _I, _I, _I, _P, _P]),
    "fcl_linear_planes_fwd": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "fcl_linear_planes_mse_fwd": (_I, [_P, _I, _P, _P, _I, _P, C.c_double, _P, _I, _P, _P, _I, _I, _I, _P]),
    "fcl_conv1d_planes_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fcl_conv1d_planes_rows_fwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "fcl_embedding_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "fcl_linear_fwd": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fcl_conv1d_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fcl_layernorm_fwd": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _P]),
    "fcl_layernorm_group_fwd": (_I, [_P, _I, C.c_int64, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fcl_conv1d_planes_bn_fwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P]),
    "fcl_conv1d_planes_group_fwd": (_I, [_P, _I, C.c_int64, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fcl_duration_round_fwd": (_I, [_P, _P, _I, _I, _F, _P, _P]),
    "fcl_variance_embed_add_fwd": (_I, [_P] * 12 + [_I, _I, _I, _P]),
    "fcl_position_table_fwd": (_I, [_P, _P, _I, _I, _P]),
    "fcl_concat_spk_fwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fcl_kaldi_ark_append": (C.c_longlong, [_I, C.c_longlong, _I, _P, _P, _P, _I, _P]),
    "fcl_host_device_ptr": (_P, [_P]),
    "fcl_feed_copy": (_I, [_P, _P, _Z, _P, _P, _P, _P]),
    "fcl_row_maps_build": (_I, [C.POINTER(RowMaps), _P]),
    "fcl_gather_rows_fwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "fcl_bilstm_workspace_bytes": (_Z, [_I, _I, _I]),
    "fcl_bilstm_fwd": (_I, [_P] * 13 + [_I, _I, _I, _I, _I, _P, _Z, _P, _P, _P]),
    "fcl_decoder_loop_workspace_bytes": (_Z, [C.POINTER(DecoderWeights), _I]),
    "fcl_decoder_stream_bytes": (_Z, [C.POINTER(DecoderWeights)]),
    "fcl_decoder_stream_pack": (_I, [C.POINTER(DecoderWeights), _P, _Z, _P]),
    "fcl_decoder_loop_fwd": (_I, [C.POINTER(DecoderWeights), C.POINTER(DecoderIO), _P]),
    "fcl_masked_l1_mse_fwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _F, _P, _P]),
    "fcl_gemm_tn_fwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "fcl_gemm_tn_taps_fwd": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _Z, _P, _P, _P]),
    "fcl_colsum_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fcl_colsum2_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fcl_conv1d_in1_dw": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fcl_act_fwd": (_I, [_P, _P, _F, _P, _P, _I, _Z, _I, _P]),
    "fcl_unpack_conv1d_grad": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "fcl_act_bwd": (_I, [_P, _P, _P, _F, _P, _P, _I, _Z, _I, _P]),
    "fcl_l1_mse_grad": (_I, [_P, _P, _P, _I, _I, _I, _F, _F, _F, C.c_double, _P, _I, _P]),
    "fcl_l1_mse_loss_grad": (_I, [_P, _P, _P, _I, _I, _I, _F, _F, _F, C.c_double, _P, _I, _P, _P, _P]),
    "fcl_layernorm_bwd": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _P]),
    "fcl_bn_stats_fwd": (_I, [_P, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    "fcl_bn_stats_ws_fwd": (_I, [_P, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    "fcl_bn_act_fwd": (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "fcl_bn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "fcl_scale": (_I, [_P, _Z, _F, _P]),
    "fcl_bernoulli_u8": (_I, [_P, _Z, _F, C.c_uint32, _P, _P]),
    "fcl_bernoulli_batch": (_I, [C.POINTER(BernoulliSite), _I, _P]),
    "fcl_lstm_cell_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _F, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P]),
    "fcl_scatter_add_rows": (_I, [_P, _P, _P, _I, _I, C.c_int64, _P]),
    "fcl_transpose2d": (_I, [_P, _P, _I, _I, _P]),
    "fcl_pack_planes_t": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "fcl_gemm_tn_planes": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _Z, _P]),
    "fcl_pwg_upsample_stage": (_I, [_P, _P, _P, C.c_int64, _I, _I, _P, _P, _P, _I, _I, _P]),
    "fcl_pwg_noise": (_I, [_P, C.c_int64, C.c_uint32, _P]),
    "fcl_pwg_aux_coeff": (_I, [_P, C.c_int64, _I, C.c_int64, _P, _P]),
    "fcl_pwg_first_conv": (_I, [_P, _P, _P, _P, _P, C.c_int64, _I, _I, _P]),
    "fcl_pwg_layer_fwd": (_I, [_P, _P]),
    "fcl_pwg_last_fwd": (_I, [_P, _F, _P, _P, _P, _F, _P, _P, _P, C.c_int64, _I, _P]),
    "fcl_derive_blocks": (_I, [_I, _I, _I]),
    "fcl_derive_batch": (_I, [_P, _I, _I, _P]),
    "fcl_sumsq_accum": (_I, [_P, _Z, _P, _P]),
    "fcl_adam_step": (_I, [_P, _P, _P, _P, _Z, _P, _F, _F, _F, _F, _F, _P, _P, _P]),
    "fcl_adam_step_wd": (_I, [_P, _P, _P, _P, _Z, _P, _F, _F, _F, _F, _F, _F, _P, _P, _P]),
    "fcl_lstm_step_fwd": (_I, [C.POINTER(LstmStep), _P]),
    "fcl_decoder_train_workspace_bytes": (_Z, [_I, _I]),
    "fcl_decoder_train_fwd": (_I, [C.POINTER(DecoderTrain), _P]),
    "fcl_decoder_bptt": (_I, [C.POINTER(DecoderBptt), _P]),
    "fcl_bilstm_train_workspace_bytes": (_Z, [_I, _I]),
    "fcl_bilstm_train_fwd": (_I, [C.POINTER(BilstmTrain), _P]),
    "fcl_bilstm_bptt": (_I, [C.POINTER(BilstmBptt), _P]),
    "fcl_loss_terms_batch": (_I, [C.POINTER(LossTerm), _I, _P]),
    "fcl_sum_rows": (_I, [C.POINTER(C.c_void_p), _I, _P, _P, _P, _I, _I, _P]),
    "fcl_bn_bwd_sums": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "fcl_act_bwd_sum": (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _Z, _I, _P]),
    "fcl_gather_rows_sum_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "fcl_linear2_fwd": (_I, [_P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "fcl_stream_create_cus": (_I, [_I, C.POINTER(C.c_void_p)]),
    "fcl_stream_destroy": (_I, [_P]),
    "fcl_streams_share_pipe": (_I, [_P, _P, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "fcl_stream_create_apart": (_I, [C.POINTER(C.c_void_p), _I, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "fcl_prof_enable": (_I, [_I]),
    "fcl_prof_collect": (_I, [C.POINTER(ProfEntry), _I]),
}

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
DROP_NONE, DROP_MASK, DROP_RNG = 0, 1, 2
STATUS_GROUP_TIMEOUT, STATUS_ZERO_DURATION, STATUS_LMAX_CAP, STATUS_FRAMES_CAP, STATUS_ROWS_CAP = 1, 2, 4, 8, 16  # FCL_STATUS_* bits of a device status word

_lib = None


class FclError(RuntimeError):
    pass


def load():
    """Load libfcl_hip.so and bind every symbol of include/fcl_hip.h.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FclError(
            "fcl-taco2_amd: %s not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fcl-taco2_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # torch first: device memory and streams come from torch, so libfcl_hip.so must bind to the HIP runtime torch has loaded.  Loading the .so
    # before torch pulls in /opt/rocm's libamdhip64 and torch then finds "no ROCm-capable device" in the same process (seen with build(); smoke()).
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.fcl_version() != ABI_VERSION:  # a stale in-tree build: the struct layouts mirrored here would not match
        raise FclError("fcl-taco2_amd: %s has ABI revision %d, this package mirrors %d — rebuild it (`make -C fcl-taco2_amd/csrc`)"
                       % (LIB_PATH, lib.fcl_version(), ABI_VERSION))
    _lib = lib
    return lib


def prof_enable(on):
    check(load().fcl_prof_enable(int(on)))


def prof_collect():
    """{kernel name: dict(launches, ms, flops, rows)} for the launches since prof_enable(True)."""
    buf = (ProfEntry * 256)()
    n = load().fcl_prof_collect(buf, 256)
    if n < 0:
        check(n)
    return {buf[i].name.decode(): dict(launches=buf[i].launches, ms=buf[i].ms, flops=buf[i].flops, rows=buf[i].rows, fill_bytes=buf[i].fill_bytes) for i in range(n)}


def check(rc):
    if rc != 0:
        msg = load().fcl_last_error()
        raise FclError("libfcl_hip error %d: %s" % (rc, msg.decode("utf-8", "replace") if msg else "?"))
