"""ctypes binding of libdsdneo_hip.so (include/ddn_hip.h, include/ddn_mbe.h) — the shape of stub a host application adds.

This module is plumbing for tests/ and bench.py: it loads the in-tree shared library and declares the C-ABI
prototypes.  It never falls back to a CPU implementation: if the library (or a GPU) is missing the calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(_HERE, "..", ".."))
LIB_PATH = os.environ.get("DDN_LIB_PATH") or os.path.join(ROOT, "dsd-neo_amd", "libdsdneo_hip.so")  # DDN_LIB_PATH: timing-experiment builds (tools/build_variant.sh)

DDN_OK, DDN_EINVAL, DDN_ENODEV, DDN_ENOMEM, DDN_EHIP, DDN_ERANGE = 0, -1, -2, -3, -4, -5
LPF_WIDE, LPF_6K25, LPF_12K5, LPF_PROVOICE, LPF_P25_C4FM, LPF_P25_CQPSK = range(6)
IN_CU8, IN_CF32 = 0, 1


class FrontEndConfig(C.Structure):
    _fields_ = [
        ("n_channels", C.c_int),
        ("sample_rate_hz", C.c_int),
        ("symbol_rate_hz", C.c_int),
        ("levels", C.c_int),
        ("lpf_profile", C.c_int),
        ("input_format", C.c_int),
        ("block_len", C.c_int),
        ("squelch_level", C.c_float),
    ]


class Cu8Moments(C.Structure):  # == dsd_input_level_cu8_moments (include/ddn_hip.h)
    _fields_ = [("count", C.c_uint64), ("sum", C.c_uint64), ("sum_sq", C.c_uint64), ("clipped", C.c_uint64),
                ("min_sample", C.c_uint8), ("max_sample", C.c_uint8)]


class ModeFlags(C.Structure):  # == ddn_mode_flags
    _fields_ = [(k, C.c_int) for k in ("p25p1", "p25p2", "provoice", "dmr", "nxdn48", "nxdn96", "x2tdma", "ysf", "dstar", "dpmr", "m17",
                                       "mod_qpsk", "analog_only")]


class ModeResult(C.Structure):  # == ddn_mode_result
    _fields_ = [(k, C.c_int) for k in ("output_kind", "symbol_rate_hz", "symbol_levels", "lpf_profile", "channel_lpf_enable",
                                       "cqpsk_enable", "ted_enabled", "samples_per_symbol")]


class DemodState(C.Structure):  # == struct demod_state of include/ddn_demod_adapter.h
    _fields_ = [("result", C.c_float * (16 * 16384)), ("lowpassed", C.POINTER(C.c_float)), ("lp_len", C.c_int), ("result_len", C.c_int),
                ("rate_in", C.c_int), ("rate_out", C.c_int), ("output_kind", C.c_int), ("symbol_rate_hz", C.c_int),
                ("symbol_levels", C.c_int), ("channel_lpf_enable", C.c_int), ("channel_lpf_profile", C.c_int),
                ("channel_squelch_level", C.c_float), ("cqpsk_enable", C.c_int), ("ted_enabled", C.c_int), ("ted_sps", C.c_int),
                ("ted_gain", C.c_float), ("ddn_adapter", C.c_void_p)]


class FskModemState(C.Structure):
    _fields_ = [
        ("cfg_sample_rate_hz", C.c_int),
        ("cfg_symbol_rate_hz", C.c_int),
        ("cfg_levels", C.c_int),
        ("cfg_channel_profile", C.c_int),
        ("prev_i", C.c_float),
        ("prev_q", C.c_float),
        ("have_prev", C.c_int),
        ("dc_est", C.c_float),
        ("discriminator_peak_est", C.c_float),
    ]


# every symbol include/ddn_hip.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "ddn_last_error": (C.c_char_p, []),
    "ddn_version": (C.c_char_p, []),
    "ddn_batch_create": (C.c_int, [C.POINTER(FrontEndConfig), C.POINTER(C.c_void_p)]),
    "ddn_batch_destroy": (None, [C.c_void_p]),
    "ddn_batch_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_batch_get_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ddn_front_end_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_front_end_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_batch_set_channels_per_workgroup": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_batch_set_segments": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_front_end_run_segments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_batch_get_fsk_state": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ddn_batch_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_batch_get_timing": (C.c_int, [C.c_void_p, C.c_void_p]),
    "simd_fir_complex_apply": (None, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "simd_hb_decim2_complex": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "simd_hb_decim2_real": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "simd_fir_get_impl_name": (C.c_char_p, []),
    "widen_u8_to_f32_bias127": (None, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "widen_u8_to_f32_bias127_moments": (None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "widen_rotate90_u8_to_f32_bias127_phase": (C.c_uint32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "widen_rotate90_u8_to_f32_bias127_phase_moments": (C.c_uint32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "ddn_fec_p25_12_soft_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_12_soft_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_r34_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_r34_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_fec_nxdn_conv_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p]),
    "ddn_fec_nxdn_conv_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int]),
    "ddn_fec_viterbi_k5_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p]),
    "ddn_fec_viterbi_k5_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                          C.c_void_p]),
    "ddn_slicer_batch_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ddn_slicer_batch_destroy": (None, [C.c_void_p]),
    "ddn_slicer_batch_reset": (C.c_int, [C.c_void_p]),
    "ddn_p25_slicer_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_p25_slicer_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_slicer_batch_get_thresholds": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ddn_p25_matched_filter_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_p25_matched_filter_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_batch_set_decimation": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_cqpsk_batch_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ddn_cqpsk_batch_destroy": (None, [C.c_void_p]),
    "ddn_cqpsk_batch_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_cqpsk_max_symbols": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "ddn_cqpsk_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_cqpsk_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_cqpsk_get_state": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ddn_cq_rx_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ddn_cq_rx_destroy": (None, [C.c_void_p]),
    "ddn_cq_rx_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_cq_rx_set_events": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ddn_cq_rx_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                               C.c_void_p]),
    "ddn_cq_rx_get_state": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ddn_p25_rx_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ddn_p25_rx_destroy": (None, [C.c_void_p]),
    "ddn_p25_rx_reset": (C.c_int, [C.c_void_p]),
    "ddn_p25_rx_set_lock_symbols": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25_rx_set_channels_per_wave": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_p25_rx_set_filter_in_loop": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_p25_rx_set_debug_flags": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_p25_rx_set_handlers": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ddn_p25_rx_debug_counters": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ddn_p25_rx_set_events": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ddn_p25_rx_run_host_ev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_p25_rx_set_event_data": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25_rx_max_symbols": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "ddn_p25_rx_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                 C.c_void_p]),
    "ddn_p25_rx_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_size_t]),
    "ddn_p25_rx_get_thresholds": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ddn_ted_batch_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_void_p)]),
    "ddn_ted_batch_destroy": (None, [C.c_void_p]),
    "ddn_ted_batch_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_ted_batch_set_block_len": (C.c_int, [C.c_void_p, C.c_size_t]),
    "ddn_gardner_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_gardner_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_ted_batch_get_state": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ddn_p25p1_nid_decode_batch": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_p25p1_nid_decode_host": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_size_t, C.c_void_p]),
    "ddn_p25p1_nid_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_ubyte, C.c_uint8, C.c_int, C.c_void_p]),
    "ddn_p25p1_imbe_deinterleave_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p1_imbe_deinterleave_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_batch_set_iq_conditioning": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
    "ddn_p25p1_layout_nid": (C.c_int, [C.c_void_p]),
    "ddn_p25p1_layout_trellis_block": (C.c_int, [C.c_int, C.c_void_p]),
    "ddn_p25p1_layout_ldu_words": (C.c_int, [C.c_int, C.c_void_p]),
    "ddn_p25p1_layout_ldu_imbe": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25p1_framer_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ddn_p25p1_framer_destroy": (None, [C.c_void_p]),
    "ddn_p25p1_framer_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_p25p1_framer_get_syncs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p1_framer_gather_nid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 6),
    "ddn_p25p1_framer_gather_trellis_block": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
                                              + [C.c_void_p] * 4),
    "ddn_p25p1_framer_gather_r34_block": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
                                          + [C.c_void_p] * 4),
    "ddn_p25p1_framer_gather_ldu_words": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
                                          + [C.c_void_p] * 4),
    "ddn_p25p1_framer_pack_ldu_rs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p1_layout_hdu": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25p1_framer_gather_hdu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 6),
    "ddn_p25p1_framer_pack_hdu_rs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p1_layout_tdulc": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25p1_framer_gather_tdulc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 6),
    "ddn_p25p1_framer_pack_tdulc_rs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p1_layout_ldu_lsd": (C.c_int, [C.c_void_p]),
    "ddn_p25p1_framer_gather_lsd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 4),
    "ddn_fec_p25_crc16_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_crc16_host": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "crc16_lb_bridge": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_fec_p25_lsd_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_lsd_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "p25_lsd_fec_16x8": (C.c_int, [C.c_void_p]),
    "p25_lsd_fec_16x8_soft": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25p1_framer_imbe_index": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_iq_capture_read_info": (C.c_int, [C.c_char_p, C.c_void_p]),
    "ddn_iq_capture_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ddn_iq_capture_close": (None, [C.c_void_p]),
    "ddn_iq_capture_get_info": (C.c_void_p, [C.c_void_p]),
    "ddn_iq_capture_get_events": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "ddn_iq_capture_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ddn_iq_capture_rewind": (C.c_int, [C.c_void_p]),
    "ddn_iq_effective_bytes": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "ddn_iq_load_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]),
    "ddn_iq_free": (None, [C.c_void_p]),
    "ddn_stream_set_create": (C.c_int, [C.c_int, C.c_size_t, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_void_p)]),
    "ddn_stream_set_destroy": (None, [C.c_void_p]),
    "ddn_stream_set_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "ddn_stream_set_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "ddn_stream_set_set_power": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "ddn_stream_set_bump_generation": (C.c_int, [C.c_void_p]),
    "ddn_stream_set_close": (None, [C.c_void_p]),
    "ddn_hooks_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "ddn_hooks_return_pwr": (C.c_double, [C.c_void_p]),
    "ddn_hooks_output_rate_hz": (C.c_uint, [C.c_void_p]),
    "ddn_hooks_output_kind": (C.c_int, [C.c_void_p]),
    "ddn_hooks_symbol_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ddn_hooks_stream_generation": (C.c_uint32, [C.c_void_p]),
    "ddn_resampler_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ddn_resampler_destroy": (None, [C.c_void_p]),
    "ddn_resampler_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_resampler_out_len": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "ddn_resampler_get_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "ddn_resampler_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_resampler_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "dsd_resampler_process_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "ddn_fec_hamming_10_6_3_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_hamming_10_6_3_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_fec_golay24_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_golay24_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_rs_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_rs_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_fec_rs28_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_rs28_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ddn_audio_s16_state_init": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_audio_s16_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_audio_s16_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_bptc_128x77_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_bptc_128x77_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_bptc_16x2_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_bptc_16x2_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "BPTC_128x77_Extract_Data": (C.c_uint32, [C.c_void_p, C.c_void_p]),
    "BPTC_16x2_Extract_Data": (C.c_uint32, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "ddn_fec_isch_lookup_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_isch_lookup_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "isch_lookup": (C.c_int, [C.c_uint64]),
    "isch_lookup_soft": (C.c_int, [C.c_uint64, C.c_void_p]),
    "ez_rs28_ess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ez_rs28_facch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ez_rs28_sacch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ddn_fec_golay24_soft_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "ddn_fec_golay24_soft_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_hamming_10_6_3_soft_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_hamming_10_6_3_soft_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_rs_soft_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                            C.c_void_p]),
    "ddn_fec_p25_rs_soft_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "p25p1_rs_24_12_13_soft_reliability": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "p25p1_rs_24_16_9_soft_reliability": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "p25p1_rs_36_20_17_soft_reliability": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "check_and_fix_golay_24_6_soft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "check_and_fix_golay_24_12_soft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hamming_10_6_3_soft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "check_and_fix_golay_24_6": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "check_and_fix_golay_24_12": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "check_and_fix_reedsolomon_24_12_13": (C.c_int, [C.c_void_p, C.c_void_p]),
    "check_and_fix_reedsolomon_24_16_9": (C.c_int, [C.c_void_p, C.c_void_p]),
    "check_and_fix_redsolomon_36_20_17": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hamming_10_6_3_decode": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_12_soft_list_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_12_soft_list_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "p25_12_soft_llr_list": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "p25_12_soft_llr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_r34_list_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_r34_list_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "dmr_r34_viterbi_decode_list": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dmr_r34_viterbi_decode": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dmr_r34_viterbi_decode_soft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "viterbi_decode": (C.c_uint32, [C.c_void_p, C.c_void_p, C.c_uint16]),
    "viterbi_decode_punctured": (C.c_uint32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint16, C.c_uint16]),
    "viterbi_decode_bit": (None, [C.c_uint16, C.c_uint16, C.c_size_t]),
    "viterbi_chainback": (C.c_uint32, [C.c_void_p, C.c_size_t, C.c_uint16]),
    "viterbi_reset": (None, []),
    "dsd_fsk_modem_discriminator_process": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "CNXDNConvolution_init": (None, []),
    "CNXDNConvolution_start": (None, []),
    "CNXDNConvolution_decode": (None, [C.c_uint8, C.c_uint8]),
    "CNXDNConvolution_decode_soft": (None, [C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8]),
    "CNXDNConvolution_chainback": (None, [C.c_void_p, C.c_uint]),
    "ddn_fsk_modem_discriminator_process": (C.c_int, [C.POINTER(FskModemState), C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
}

# ---- include/ddn_mbe.h ------------------------------------------------------------------------------------------------
class MbeParms(C.Structure):
    _fields_ = [("w0", C.c_float), ("L", C.c_int), ("K", C.c_int), ("Vl", C.c_int * 57), ("Ml", C.c_float * 57),
                ("log2Ml", C.c_float * 57), ("PHIl", C.c_float * 57), ("PSIl", C.c_float * 57), ("gamma", C.c_float),
                ("un", C.c_int), ("repeat", C.c_int)]


class MbeProcessResult(C.Structure):
    _fields_ = [("flags", C.c_uint), ("c0_errors", C.c_int), ("c4_errors", C.c_int), ("total_errors", C.c_int),
                ("protected_errors", C.c_int)]


class MbeTables(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("synthetic", C.c_uint32),
                ("imbe_gain_b2", C.c_float * 64), ("imbe_gain_step", C.c_float * 11), ("imbe_gain_sigma", C.c_float * 5),
                ("imbe_hoc_step", C.c_float * 11), ("imbe_hoc_sigma", C.c_float * 9),
                ("imbe_bits", (C.c_uint8 * 58) * 48), ("imbe_bit_order", ((C.c_uint8 * 2) * 88) * 48),
                ("ambe_f0", C.c_float * 120), ("ambe_L", C.c_uint8 * 120), ("ambe_vuv", (C.c_uint8 * 8) * 32),
                ("ambe_dg", C.c_float * 32), ("ambe_prba24", (C.c_float * 3) * 512), ("ambe_prba58", (C.c_float * 4) * 128),
                ("ambe_hoc5", (C.c_float * 4) * 32), ("ambe_hoc6", (C.c_float * 4) * 16), ("ambe_hoc7", (C.c_float * 4) * 16),
                ("ambe_hoc8", (C.c_float * 4) * 8), ("ambe_blocks", (C.c_uint8 * 4) * 57)]


MBE_IMBE, MBE_AMBE = 0, 1
_PP = C.POINTER(MbeParms)
PROTOTYPES.update({
    "ddn_fec_block_code_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_block_code_host": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_fec_bptc_196x96_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_bptc_196x96_host": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_trellis_decode_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "ddn_fec_trellis_decode_host": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_int]),
    "trellis_decode": (None, [C.c_void_p, C.c_void_p, C.c_int]),
    "ddn_fec_rs_12_9_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_rs_12_9_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "Hamming_7_4_init": (None, []), "Hamming_12_8_init": (None, []), "Hamming_13_9_init": (None, []),
    "Hamming_15_11_init": (None, []), "Hamming_16_11_4_init": (None, []), "Golay_20_8_init": (None, []),
    "Golay_24_12_init": (None, []), "QR_16_7_6_init": (None, []), "InitAllFecFunction": (None, []),
    "Hamming_7_4_decode": (C.c_bool, [C.c_void_p]),
    "Hamming_12_8_decode": (C.c_bool, [C.c_void_p, C.c_void_p, C.c_int]),
    "Hamming_13_9_decode": (C.c_bool, [C.c_void_p, C.c_void_p, C.c_int]),
    "Hamming_15_11_decode": (C.c_bool, [C.c_void_p, C.c_void_p, C.c_int]),
    "Hamming_16_11_4_decode": (C.c_bool, [C.c_void_p, C.c_void_p, C.c_int]),
    "Golay_20_8_decode": (C.c_bool, [C.c_void_p]), "Golay_24_12_decode": (C.c_bool, [C.c_void_p]),
    "QR_16_7_6_decode": (C.c_bool, [C.c_void_p]),
    "BPTCDeInterleaveDMRData": (None, [C.c_void_p, C.c_void_p]),
    "BPTC_196x96_Extract_Data": (C.c_uint32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rs_12_9_calc_syndrome": (None, [C.c_void_p, C.c_void_p]),
    "rs_12_9_check_syndrome": (C.c_uint8, [C.c_void_p]),
    "rs_12_9_correct_errors": (C.c_uint8, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25_rx_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_p25_rx_get_timing": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25_rx_get_timing_avg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_mbe_default_tables": (C.c_int, [C.POINTER(MbeTables)]),
    "ddn_mbe_validate_tables": (C.c_int, [C.POINTER(MbeTables)]),
    "ddn_mbe_frame_decode_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_mbe_result_skip_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_p25p1_framer_voice_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_mbe_batch_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ddn_mbe_batch_destroy": (None, [C.c_void_p]),
    "ddn_mbe_batch_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_mbe_batch_set_tables": (C.c_int, [C.c_void_p, C.POINTER(MbeTables)]),
    "ddn_mbe_dropin_set_tables": (C.c_int, [C.c_void_p]),
    "ddn_mbe_batch_set_p25p1_tail_rule": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_mbe_synth_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_mbe_batch_get_state": (C.c_int, [C.c_void_p, C.c_int, _PP, _PP, _PP]),
    "ddn_mbe_batch_set_state": (C.c_int, [C.c_void_p, C.c_int, _PP, _PP, _PP]),
    "ddn_mbe_batch_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_mbe_batch_get_timing": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mbe_initMbeParms": (None, [_PP, _PP, _PP]),
    "mbe_initProcessResult": (None, [C.POINTER(MbeProcessResult)]),
    "mbe_synthesizeSilencef": (None, [C.c_void_p]),
    "mbe_formatProcessResult": (None, [C.c_char_p, C.c_size_t, C.POINTER(MbeProcessResult)]),
    "mbe_decodeImbe7200x4400Frame": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(MbeProcessResult)]),
    "mbe_decodeImbe7200x4400SoftFrame": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(MbeProcessResult)]),
    "mbe_decodeAmbe3600x2450Frame": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(MbeProcessResult)]),
    "mbe_decodeAmbe3600x2450SoftFrame": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(MbeProcessResult)]),
    "mbe_processImbe4400Dataf": (C.c_int, [C.c_void_p, C.POINTER(MbeProcessResult), C.c_void_p, _PP, _PP, _PP]),
    "mbe_processAmbe2450Dataf": (C.c_int, [C.c_void_p, C.POINTER(MbeProcessResult), C.c_void_p, _PP, _PP, _PP]),
    "mbe_processAmbe2400Dataf": (C.c_int, [C.c_void_p, C.POINTER(MbeProcessResult), C.c_void_p, _PP, _PP, _PP]),
    "mbe_decodeImbe7100x4400Frame": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(MbeProcessResult)]),
    "mbe_processAmbe3600x2450Framef": (C.c_int, [C.c_void_p, C.POINTER(MbeProcessResult), C.c_void_p, C.c_void_p, _PP, _PP, _PP]),
    "mbe_processAmbe3600x2450SoftFramef": (C.c_int, [C.c_void_p, C.POINTER(MbeProcessResult), C.c_void_p, C.c_void_p, _PP, _PP, _PP]),
    "mbe_processAmbe3600x2400Framef": (C.c_int, [C.c_void_p, C.POINTER(MbeProcessResult), C.c_void_p, C.c_void_p, _PP, _PP, _PP]),
    "mbe_floattoshort": (None, [C.c_void_p, C.c_void_p]),
    "mbe_versionString": (C.c_char_p, []),
})


_lib = None


class Fsk4RxConfig(C.Structure):  # == ddn_fsk4_rx_config (include/ddn_fsk4.h)
    _fields_ = [("n_channels", C.c_int), ("out_rate_hz", C.c_int), ("protocol", C.c_int), ("rf_mod", C.c_int),
                ("inverted", C.c_int), ("use_matched_filter", C.c_int), ("lock_symbols", C.c_int * 4)]


FSK4_DMR, FSK4_NXDN48, FSK4_PRE = 1, 2, 90
FSK4_NXDN96 = 3
FSK4_M17 = 4
FSK4_YSF = 5
PROTOTYPES.update({
    "ddn_fsk4_rx_create": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_fsk4_rx_destroy": (None, [C.c_void_p]),
    "ddn_fsk4_rx_reset": (C.c_int, [C.c_void_p]),
    "ddn_fsk4_rx_set_handlers": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_fsk4_rx_set_events": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ddn_fsk4_rx_events_host_arm": (C.c_int, [C.c_void_p, C.c_size_t]),
    "ddn_fsk4_rx_events_host_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fsk4_rx_max_symbols": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "ddn_fsk4_rx_max_syncs": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "ddn_fsk4_rx_set_lock_symbols": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_fsk4_rx_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 4 + [C.c_size_t] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]),
    "ddn_fsk4_rx_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 4 + [C.c_size_t] + [C.c_void_p] * 5 + [C.c_size_t]),
    "ddn_fsk4_rx_get_thresholds": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ddn_fsk4_rx_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_fsk4_rx_set_channels_per_wave": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_fsk4_rx_set_debug_flags": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_fsk4_rx_set_sync_thresholds": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_m17_lsf_decode_batch": (C.c_int, [C.c_void_p, C.c_size_t] + [C.c_void_p] * 5 + [C.c_int, C.c_size_t] + [C.c_void_p] * 4),
    "ddn_m17_str_decode_batch": (C.c_int, [C.c_void_p, C.c_size_t] + [C.c_void_p] * 4 + [C.c_int, C.c_size_t] + [C.c_void_p] * 5),
    "ddn_m17_lich_assemble_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t] + [C.c_void_p] * 9),
    "ddn_ysf_fich_decode_batch": (C.c_int, [C.c_void_p, C.c_size_t] + [C.c_void_p] * 3 + [C.c_int, C.c_size_t] + [C.c_void_p] * 4),
    "ddn_ysf_payload_decode_batch": (C.c_int, [C.c_void_p, C.c_size_t] + [C.c_void_p] * 3 + [C.c_int, C.c_size_t] + [C.c_void_p] * 12),
    "ddn_fsk4_rx_get_timing": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_mode_config": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "full_demod": (None, [C.c_void_p]),
    "op25_gardner_cc": (None, [C.c_void_p]),
    "ddn_demod_state_release": (None, [C.c_void_p]),
    "ddn_audio_agf_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_audio_agf_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "ddn_agf_frame": (C.c_int, [C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "ddn_symbol_capture_write": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "ddn_wav_write_s16": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_float]),
    "ddn_ambe2450_deinterleave_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_nxdn_voice_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_dmr_voice_burst_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_nxdn_frame_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t] + [C.c_void_p] * 7),
    "ddn_nxdn_crc_check_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_dmr_burst_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int]
                             + [C.c_void_p] * 5),
})


class P25ChainConfig(C.Structure):  # == ddn_p25_chain_config (include/ddn_chain.h)
    _fields_ = [("n_channels", C.c_int), ("samples_per_call", C.c_int), ("block_len", C.c_int), ("input_format", C.c_int),
                ("vocoder", C.c_int), ("max_frames", C.c_int), ("max_ldu", C.c_int), ("max_events", C.c_int),
                ("carry_symbols", C.c_int), ("modulation", C.c_int), ("sample_rate_hz", C.c_int), ("snr_cqpsk_db", C.c_float),
                ("d2h_blit", C.c_int)]


class P25ChainResults(C.Structure):  # == ddn_p25_chain_results
    _fields_ = [("stride_symbols", C.c_size_t), ("d_records10", C.c_void_p), ("d_flags", C.c_void_p), ("d_counts", C.c_void_p),
                ("d_new", C.c_void_p), ("d_events", C.c_void_p), ("d_n_events", C.c_void_p), ("d_event_data", C.c_void_p),
                ("d_n_syncs", C.c_void_p), ("d_dropped_syncs", C.c_void_p),
                ("d_sync_pos", C.c_void_p), ("d_nid4", C.c_void_p), ("d_tsbk", C.c_void_p), ("d_tsbk_crc", C.c_void_p),
                ("d_ldu_words", C.c_void_p * 2), ("d_ldu_rs_data", C.c_void_p * 2), ("d_ldu_rs_status", C.c_void_p * 2),
                ("d_lsd_bits", C.c_void_p), ("d_lsd_ok", C.c_void_p), ("d_hdu_rs_data", C.c_void_p), ("d_hdu_rs_status", C.c_void_p),
                ("d_tdulc_rs_data", C.c_void_p), ("d_tdulc_rs_status", C.c_void_p),
                ("pdu_per_channel", C.c_int), ("pdu_blocks", C.c_int), ("d_n_pdu", C.c_void_p), ("d_pdu_slot", C.c_void_p),
                ("d_pdu_header", C.c_void_p), ("d_pdu_info", C.c_void_p), ("d_pdu_blocks", C.c_void_p), ("d_pdu_block_valid", C.c_void_p), ("d_pdu_blocks18", C.c_void_p), ("d_pdu_crc9_ok", C.c_void_p),
                ("d_n_ldu", C.c_void_p),
                ("d_imbe_bits", C.c_void_p), ("d_imbe_result", C.c_void_p), ("d_pcm", C.c_void_p), ("d_synth_result", C.c_void_p)]


class P25ChainHostOut(C.Structure):  # == ddn_p25_chain_host_out
    _fields_ = [("records10", C.c_void_p), ("flags", C.c_void_p), ("counts", C.c_void_p), ("events", C.c_void_p),
                ("n_events", C.c_void_p), ("event_data", C.c_void_p), ("nid4", C.c_void_p), ("tsbk", C.c_void_p), ("pcm", C.c_void_p),
                ("records2", C.c_void_p), ("pcm_dense", C.c_void_p), ("pcm_slot", C.c_void_p), ("pcm_count", C.c_void_p),
                ("pcm_dense_frames", C.c_int64)]


PROTOTYPES.update({
    "ddn_mbe_tables_save_file": (C.c_int, [C.c_char_p, C.c_void_p]),
    "ddn_mbe_tables_load_file": (C.c_int, [C.c_char_p, C.c_void_p]),
    "ddn_mbe_batch_load_tables_file": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ddn_mbe_batch_tables_synthetic": (C.c_int, [C.c_void_p]),
    "ddn_p25p1_framer_device_dropped": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_mbf34_list_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_mbf34_list_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "p25_mbf34_decode_soft_list": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ddn_p25_chain_create": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25_chain_destroy": (None, [C.c_void_p]),
    "ddn_p25_chain_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25_chain_run_pipelined": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25_chain_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25_chain_flush": (C.c_int, [C.c_void_p]),
    "ddn_p25_chain_wait": (C.c_int, [C.c_void_p]),
    "ddn_p25_chain_get_results": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25_chain_stride_symbols": (C.c_size_t, [C.c_void_p]),
    "ddn_p25_chain_frame_slots": (C.c_int, [C.c_void_p]),
    "ddn_p25_chain_max_ldu": (C.c_int, [C.c_void_p]),
    "ddn_p25_chain_max_events": (C.c_int, [C.c_void_p]),
    "ddn_p25_chain_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_p25_chain_set_first_channel": (C.c_int, [C.c_void_p, C.c_int]),
    "ddn_mbe_batch_set_first_stream": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "ddn_p25_chain_get_stage_ms": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25_chain_front_end": (C.c_void_p, [C.c_void_p]),
    "ddn_p25_chain_rx": (C.c_void_p, [C.c_void_p]),
    "ddn_p25_chain_mbe": (C.c_void_p, [C.c_void_p]),
    "ddn_p25p1_framer_device_syncs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fec_p25_tsbk_select_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_device_alloc": (C.c_int, [C.c_size_t, C.c_void_p]),
    "ddn_device_free": (None, [C.c_void_p]),
    "ddn_device_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ddn_device_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ddn_host_alloc_pinned": (C.c_int, [C.c_size_t, C.c_void_p]),
    "ddn_host_free_pinned": (None, [C.c_void_p]),
})


class Fsk4ChainConfig(C.Structure):  # == ddn_fsk4_chain_config
    _fields_ = [("n_channels", C.c_int), ("samples_per_call", C.c_int), ("block_len", C.c_int), ("input_format", C.c_int),
                ("protocol", C.c_int), ("rf_mod", C.c_int), ("inverted", C.c_int), ("handlers", C.c_int), ("vocoder", C.c_int)]


class Fsk4ChainResults(C.Structure):  # == ddn_fsk4_chain_results
    _fields_ = [("stride_symbols", C.c_size_t), ("carry_symbols", C.c_size_t), ("max_syncs", C.c_size_t), ("voice_slots", C.c_int)] + [
        (k, C.c_void_p) for k in ("d_records10", "d_flags", "d_payload2", "d_new", "d_counts", "d_n_sync", "d_dropped_syncs", "d_sync_pos", "d_sync_pat", "d_pre",
                                  "d_valid", "d_dmr_slot_type", "d_dmr_slot_type_ok", "d_dmr_pdu96", "d_dmr_bptc_errs", "d_nxdn_lich",
                                  "d_nxdn_sacch", "d_nxdn_sacch_ok", "d_nxdn_sacch_hard", "d_nxdn_sacch_hard_ok", "d_nxdn_facch",
                                  "d_nxdn_facch_ok", "d_nxdn_voice_skip", "d_nxdn_ambe_bits", "d_nxdn_pcm")] + [
        ("dmr_voice_bursts", C.c_int)] + [(k, C.c_void_p) for k in (
            "d_dmr_n_voice", "d_dmr_voice_start", "d_dmr_voice_pre", "d_dmr_voice_skip", "d_dmr_ambe_frames", "d_dmr_ambe_bits",
            "d_dmr_ambe_result", "d_dmr_pcm", "d_events", "d_n_events")] + [("max_events", C.c_int), ("dmr_data_bursts", C.c_int)] + [
        (k, C.c_void_p) for k in ("d_dmr_n_data", "d_dmr_data_start", "d_dmr_data_slot", "d_dmr_data_type", "d_dmr_data_info196",
                                  "d_dmr_data_bits96", "d_dmr_data_bytes12", "d_dmr_data_errs", "d_dmr_data_crc", "d_dmr_r34_unconfirmed",
                                  "d_dmr_r34_confirmed", "d_dmr_r34_confirmed_crc", "d_dmr_r34_pool", "d_dmr_r34_pool_n")] + [
        ("dmr_emb_lcs", C.c_int)] + [(k, C.c_void_p) for k in ("d_dmr_n_emb", "d_dmr_emb_pos", "d_dmr_emb_lc77", "d_dmr_emb_errs", "d_dmr_emb_ok")] + [
        (k, C.c_void_p) for k in ("d_sync_thr5", "d_m17_lsf30", "d_m17_lsf_status", "d_m17_lsf_cost", "d_m17_lich6", "d_m17_lich_cnt",
                                  "d_m17_fn_payload18", "d_m17_str_status", "d_m17_lich_lsf30", "d_m17_lich_status", "d_ysf_fich4",
                                  "d_ysf_fich_status", "d_ysf_fich_cost", "d_ysf_info2", "d_ysf_dch40", "d_ysf_dch_status2",
                                  "d_ysf_dch_cost2", "d_ysf_ambe49x5", "d_ysf_errs2x5", "d_ysf_frames184x5", "d_ysf_n_frames")] + [("ysf_voice_frames", C.c_int)] + [
        (k, C.c_void_p) for k in ("d_ysf_n_voice", "d_ysf_voice_slot", "d_ysf_voice_result", "d_ysf_pcm", "d_ysf_voice_skip", "d_ysf_imbe_n_voice",
                                  "d_ysf_imbe_voice_slot", "d_ysf_imbe_voice_skip", "d_ysf_imbe_voice_result", "d_ysf_imbe_pcm")]


class MixedChainConfig(C.Structure):  # == ddn_mixed_chain_config
    _fields_ = [("n_p25", C.c_int), ("n_dmr", C.c_int), ("n_nxdn48", C.c_int), ("samples_per_call", C.c_int), ("block_len", C.c_int),
                ("input_format", C.c_int), ("vocoder", C.c_int), ("overlap", C.c_int)]


PROTOTYPES.update({
    "ddn_p25_chain_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_p25_chain_d2h_route": (C.c_int, [C.c_void_p]),
    "ddn_p25p2_chain_create": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ddn_p25p2_chain_destroy": (None, [C.c_void_p]),
    "ddn_p25p2_chain_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_chain_flush": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25p2_chain_get_results": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_p25p2_mac_crc_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_mac_crc_host": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_ess_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "ddn_p25p2_ess_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_voice_frames_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_scramble_bits_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_descramble_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "p25p2_generate_scramble_bits": (None, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]),
    "ddn_p25p2_burst_fields_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_groups_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_sync_cut_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_groups_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_sync_cut_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_burst_fields_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_xcch_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_p25p2_xcch_host": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fsk4_chain_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_fsk4_chain_create": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_fsk4_chain_destroy": (None, [C.c_void_p]),
    "ddn_fsk4_chain_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_fsk4_chain_get_results": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_fsk4_chain_flush": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_fsk4_chain_front_end": (C.c_void_p, [C.c_void_p]),
    "ddn_fsk4_chain_rx": (C.c_void_p, [C.c_void_p]),
    "ddn_mixed_chain_create": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_mixed_chain_destroy": (None, [C.c_void_p]),
    "ddn_mixed_chain_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_mixed_chain_wait": (C.c_int, [C.c_void_p]),
    "ddn_mixed_chain_part": (C.c_void_p, [C.c_void_p, C.c_int]),
    "ddn_mixed_partition": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
})


def mixed_partition(n_p25, n_dmr, n_nxdn48, rank, world):
    """ddn_mixed_partition -> ((first, count) of P25, DMR, NXDN48 for this rank)"""
    f, c = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    _check(lib().ddn_mixed_partition(n_p25, n_dmr, n_nxdn48, rank, world, f, c), "ddn_mixed_partition")
    return [(f[k], c[k]) for k in range(3)]


class Fsk4ChainC:
    """ddn_fsk4_chain (include/ddn_chain.h): the DMR / NXDN48 path as one C object; fetch() copies a result array to the host"""

    def __init__(self, n_channels, samples_per_call, protocol, rf_mod=0, inverted=0, block_len=8192, handlers=1, vocoder=1, handle=None):
        import numpy as np
        self.np = np
        self.own = handle is None
        if handle is None:
            cfg = Fsk4ChainConfig(n_channels, samples_per_call, block_len, 0, protocol, rf_mod, inverted, handlers, vocoder)
            self.h = C.c_void_p()
            _check(lib().ddn_fsk4_chain_create(C.byref(cfg), C.byref(self.h)), "ddn_fsk4_chain_create")
        else:
            self.h = C.c_void_p(handle)
        self.B, self.n = n_channels, samples_per_call

    def close(self):
        if self.h and self.own:
            lib().ddn_fsk4_chain_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, d_iq_ptr, stream=None):
        _check(lib().ddn_fsk4_chain_run(self.h, d_iq_ptr, stream), "ddn_fsk4_chain_run")

    def flush(self, stream=None):
        _check(lib().ddn_fsk4_chain_flush(self.h, stream), "ddn_fsk4_chain_flush")

    def results(self):
        r = Fsk4ChainResults()
        _check(lib().ddn_fsk4_chain_get_results(self.h, C.byref(r)), "ddn_fsk4_chain_get_results")
        return r

    @property
    def rx(self):
        return lib().ddn_fsk4_chain_rx(self.h)

    def fetch(self, ptr, dtype, shape):
        a = self.np.zeros(shape, dtype)
        _check(lib().ddn_device_download(a.ctypes.data, ptr, a.nbytes), "ddn_device_download")
        return a


class MixedChainC:
    """ddn_mixed_chain: P25 Phase 1 + DMR + NXDN48 channel groups of one GPU (BASELINE configs[3])"""

    def __init__(self, n_p25, n_dmr, n_nxdn48, samples_per_call, block_len=8192, vocoder=1, overlap=0):
        cfg = MixedChainConfig(n_p25, n_dmr, n_nxdn48, samples_per_call, block_len, 0, vocoder, overlap)
        self.h = C.c_void_p()
        _check(lib().ddn_mixed_chain_create(C.byref(cfg), C.byref(self.h)), "ddn_mixed_chain_create")
        self.counts = (n_p25, n_dmr, n_nxdn48)
        self.n = samples_per_call

    def close(self):
        if self.h:
            lib().ddn_mixed_chain_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, p25_ptr, dmr_ptr, nxdn_ptr):
        _check(lib().ddn_mixed_chain_run(self.h, p25_ptr, dmr_ptr, nxdn_ptr), "ddn_mixed_chain_run")

    def wait(self):
        _check(lib().ddn_mixed_chain_wait(self.h), "ddn_mixed_chain_wait")

    def part(self, which):
        """the group's chain object as a non-owning view (0 P25ChainC is not wrapped: use the handle with the ddn_p25_chain_* calls)"""
        h = lib().ddn_mixed_chain_part(self.h, which)
        if not h or which == 0:
            return h
        return Fsk4ChainC(self.counts[which], self.n, 0, handle=h)


class NodeConfig(C.Structure):  # == ddn_node_config (include/ddn_node.h)
    _fields_ = [("n_channels", C.c_int), ("samples_per_call", C.c_int), ("block_len", C.c_int), ("input_format", C.c_int),
                ("vocoder", C.c_int), ("modulation", C.c_int), ("n_devices", C.c_int),
                ("kind", C.c_int), ("n_dmr", C.c_int), ("n_nxdn48", C.c_int), ("overlap", C.c_int),
                ("fsk4", C.c_void_p), ("p25p2", C.c_void_p), ("p25p2_seed44", C.c_void_p)]


NODE_P25, NODE_MIXED, NODE_FSK4, NODE_P25P2 = 0, 1, 2, 3
NODE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)

PROTOTYPES.update({
    "ddn_node_partition": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_node_create": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_node_destroy": (None, [C.c_void_p]),
    "ddn_node_parts": (C.c_int, [C.c_void_p]),
    "ddn_node_part_info": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_node_chain": (C.c_void_p, [C.c_void_p, C.c_int]),
    "ddn_node_kind_of": (C.c_int, [C.c_void_p]),
    "ddn_node_chain_object": (C.c_void_p, [C.c_void_p, C.c_int]),
    "ddn_node_part_groups": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_node_on_part": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ddn_node_run_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ddn_node_run_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddn_node_wait": (C.c_int, [C.c_void_p]),
    "ddn_node_flush": (C.c_int, [C.c_void_p]),
    "ddn_node_device_alloc": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "ddn_node_device_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ddn_node_device_download": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ddn_node_device_free": (None, [C.c_void_p, C.c_int, C.c_void_p]),
})


def node_partition(n_channels, rank, world):
    """ddn_node_partition: the block partition every multi-device layer here uses (== ddn_shard.channel_range)"""
    a, b = C.c_int(), C.c_int()
    _check(lib().ddn_node_partition(n_channels, rank, world, C.byref(a), C.byref(b)), "ddn_node_partition")
    return a.value, b.value


class NodeC:
    """ddn_node (include/ddn_node.h): one chain object + one host thread per device, the channel index block-partitioned.  kind =
    NODE_MIXED: n_channels is the P25 group, n_dmr / n_nxdn48 the other two (BASELINE configs[3]); NODE_FSK4 / NODE_P25P2: `chain_cfg` =
    the ctypes configuration structure of that chain object"""

    def __init__(self, n_channels, samples_per_call, block_len=8192, vocoder=1, input_format=0, modulation=0, n_devices=0,
                 kind=0, n_dmr=0, n_nxdn48=0, overlap=0, chain_cfg=None, seed44=None):
        self._keep = (chain_cfg, seed44)
        cfg = NodeConfig(n_channels, samples_per_call, block_len, input_format, vocoder, modulation, n_devices, kind, n_dmr, n_nxdn48,
                         overlap, C.addressof(chain_cfg) if (kind == NODE_FSK4 and chain_cfg is not None) else None,
                         C.addressof(chain_cfg) if (kind == NODE_P25P2 and chain_cfg is not None) else None,
                         seed44.ctypes.data if seed44 is not None else None)
        self.kind = kind
        self.h = C.c_void_p()
        _check(lib().ddn_node_create(C.byref(cfg), C.byref(self.h)), "ddn_node_create")
        self.parts = lib().ddn_node_parts(self.h)
        self.info = []
        for p in range(self.parts):
            d, f, n = C.c_int(), C.c_int(), C.c_int()
            _check(lib().ddn_node_part_info(self.h, p, C.byref(d), C.byref(f), C.byref(n)), "ddn_node_part_info")
            self.info.append((d.value, f.value, n.value))

    def close(self):
        if self.h:
            lib().ddn_node_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def chain(self, part):
        return lib().ddn_node_chain(self.h, part)

    def chain_object(self, part):
        return lib().ddn_node_chain_object(self.h, part)

    def groups(self, part):
        """[(first, count)] of the part's P25 / DMR / NXDN48 blocks (ddn_node_part_groups)"""
        f, n = (C.c_int32 * 3)(), (C.c_int32 * 3)()
        _check(lib().ddn_node_part_groups(self.h, part, f, n), "ddn_node_part_groups")
        return [(f[g], n[g]) for g in range(3)]

    def on_part(self, part, fn, arg=None):
        """fn(chain object pointer, arg) -> int on the part's host thread (ddn_node_on_part)"""
        cb = NODE_FN(lambda chain, a: int(fn(chain, a)))
        return lib().ddn_node_on_part(self.h, part, cb, arg)

    def run_host(self, h_iq_ptr, outs=None):
        arr = (P25ChainHostOut * self.parts)(*outs) if outs is not None else None
        self._outs = arr                                         # the worker threads read the array during the call only
        _check(lib().ddn_node_run_host(self.h, h_iq_ptr, arr), "ddn_node_run_host")

    def run_device(self, d_ptrs):
        """one device pointer per part; NODE_MIXED: three per part (P25, DMR, NXDN48; None where the part has no such channel)"""
        k = 3 if self.kind == NODE_MIXED else 1
        arr = (C.c_void_p * (k * self.parts))(*d_ptrs)
        _check(lib().ddn_node_run_device(self.h, arr), "ddn_node_run_device")

    def wait(self):
        _check(lib().ddn_node_wait(self.h), "ddn_node_wait")

    def flush(self):
        _check(lib().ddn_node_flush(self.h), "ddn_node_flush")


class P25ChainC:
    """ddn_p25_chain (include/ddn_chain.h): the whole P25 Phase 1 path as one C object.  Thin ctypes view: run() / run_pipelined()
    take a device pointer; fetch(name, dtype, shape) copies one of the result arrays of the last call to the host."""

    def __init__(self, n_channels, samples_per_call, block_len=8192, vocoder=1, max_frames=0, max_ldu=0, max_events=0,
                 carry_symbols=0, input_format=0, modulation=0, sample_rate_hz=0, snr_cqpsk_db=0.0, d2h_blit=0):
        import numpy as np
        self.np = np
        cfg = P25ChainConfig(n_channels, samples_per_call, block_len, input_format, vocoder, max_frames, max_ldu, max_events,
                             carry_symbols, modulation, sample_rate_hz, snr_cqpsk_db, d2h_blit)
        self.h = C.c_void_p()
        _check(lib().ddn_p25_chain_create(C.byref(cfg), C.byref(self.h)), "ddn_p25_chain_create")
        self.B, self.n = n_channels, samples_per_call
        self.stride = lib().ddn_p25_chain_stride_symbols(self.h)
        self.F = lib().ddn_p25_chain_frame_slots(self.h)
        self.Fv = lib().ddn_p25_chain_max_ldu(self.h)
        self.E = lib().ddn_p25_chain_max_events(self.h)
        self.T = carry_symbols or 960

    def close(self):
        if self.h:
            lib().ddn_p25_chain_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, d_iq_ptr, stream=None):
        _check(lib().ddn_p25_chain_run(self.h, d_iq_ptr, stream), "ddn_p25_chain_run")

    def run_pipelined(self, d_iq_ptr):
        _check(lib().ddn_p25_chain_run_pipelined(self.h, d_iq_ptr), "ddn_p25_chain_run_pipelined")

    def run_host(self, h_iq_ptr, out=None):
        _check(lib().ddn_p25_chain_run_host(self.h, h_iq_ptr, C.byref(out) if out is not None else None), "ddn_p25_chain_run_host")

    def flush(self):
        _check(lib().ddn_p25_chain_flush(self.h), "ddn_p25_chain_flush")

    def wait(self):
        _check(lib().ddn_p25_chain_wait(self.h), "ddn_p25_chain_wait")

    def results(self):
        r = P25ChainResults()
        _check(lib().ddn_p25_chain_get_results(self.h, C.byref(r)), "ddn_p25_chain_get_results")
        return r

    def set_timing(self, on):
        _check(lib().ddn_p25_chain_set_timing(self.h, 1 if on else 0), "ddn_p25_chain_set_timing")

    def stage_ms(self):
        t = self.np.zeros(4, self.np.float32)
        _check(lib().ddn_p25_chain_get_stage_ms(self.h, t.ctypes.data), "ddn_p25_chain_get_stage_ms")
        return t

    @property
    def rx(self):
        return lib().ddn_p25_chain_rx(self.h)

    @property
    def fe(self):
        return lib().ddn_p25_chain_front_end(self.h)

    @property
    def mbe(self):
        return lib().ddn_p25_chain_mbe(self.h)

    def fetch(self, ptr, dtype, shape):
        np = self.np
        a = np.zeros(shape, dtype)
        _check(lib().ddn_device_download(a.ctypes.data, ptr, a.nbytes), "ddn_device_download")
        return a


class Fsk4Rx:
    """ddn_fsk4_rx batch object (host-buffer convenience wrapper used by the tests)"""

    def __init__(self, n_channels, protocol, rf_mod=0, inverted=0, use_matched_filter=1, lock=None, out_rate=48000,
                 handlers=False, max_events=2048):
        """handlers=True: the reference's handlers decide the in-frame length; run_host() then also returns events / n_events"""
        import numpy as np
        self.np = np
        cfg = Fsk4RxConfig(n_channels, out_rate, protocol, rf_mod, inverted, use_matched_filter)
        for k in range(4):
            cfg.lock_symbols[k] = (lock or [0, 0, 0, 0])[k]
        self.h = C.c_void_p()
        _check(lib().ddn_fsk4_rx_create(C.byref(cfg), C.byref(self.h)), "ddn_fsk4_rx_create")
        self.B = n_channels
        self.handlers, self.max_events = bool(handlers), max_events
        if handlers:
            _check(lib().ddn_fsk4_rx_set_handlers(self.h, 1), "ddn_fsk4_rx_set_handlers")
            _check(lib().ddn_fsk4_rx_events_host_arm(self.h, max_events), "ddn_fsk4_rx_events_host_arm")

    def close(self):
        if self.h:
            lib().ddn_fsk4_rx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_host(self, disc):
        np = self.np
        disc = np.ascontiguousarray(disc, np.float32)
        B, n = disc.shape
        assert B == self.B
        l = lib()
        ms, my = l.ddn_fsk4_rx_max_symbols(self.h, n), l.ddn_fsk4_rx_max_syncs(self.h, n)
        rec, fl, pay = np.zeros((B, ms, 10), np.uint8), np.zeros((B, ms), np.uint8), np.zeros((B, ms, 2), np.uint8)
        cnt, ns = np.zeros(B, np.int32), np.zeros(B, np.int32)
        spos, spat = np.zeros((B, my), np.int32), np.zeros((B, my), np.uint8)
        pre, prel = np.zeros((B, my, FSK4_PRE), np.uint8), np.zeros((B, my, FSK4_PRE), np.uint8)
        _check(l.ddn_fsk4_rx_run_host(self.h, disc.ctypes.data, n, rec.ctypes.data, fl.ctypes.data, pay.ctypes.data, cnt.ctypes.data, ms,
                                      spos.ctypes.data, spat.ctypes.data, pre.ctypes.data, prel.ctypes.data, ns.ctypes.data, my),
               "ddn_fsk4_rx_run_host")
        out = dict(rec=rec, fl=fl, pay=pay, cnt=cnt, sync_pos=spos, sync_pat=spat, pre=pre, pre_rel=prel, n_sync=ns)
        if self.handlers:
            ev, nev = np.zeros((B, self.max_events, 4), np.int32), np.zeros(B, np.int32)
            _check(l.ddn_fsk4_rx_events_host_read(self.h, ev.ctypes.data, nev.ctypes.data), "ddn_fsk4_rx_events_host_read")
            out["events"], out["n_events"] = ev, nev
        return out

    def thresholds(self, ch):
        t = self.np.zeros(7, self.np.float32)
        _check(lib().ddn_fsk4_rx_get_thresholds(self.h, ch, t.ctypes.data), "ddn_fsk4_rx_get_thresholds")
        return t


def lib():
    """Load libdsdneo_hip.so (RTLD_LOCAL) and bind prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is None:
        try:
            if os.environ.get("DDN_NO_TORCH"):
                raise ImportError("torch import skipped on request")
            # When PyTorch is in the process it must load ITS HIP runtime first: two different libamdhip64
            # images in one process leave the second one without a device ("No HIP GPUs are available").
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class DdnError(RuntimeError):
    pass


def _check(rc, what):
    if rc < 0:
        raise DdnError("%s failed: rc=%d %s" % (what, rc, lib().ddn_last_error().decode()))
    return rc


class Batch:
    """RAII wrapper over ddn_batch (B channels of the FSK front end on the current HIP device)."""

    def __init__(self, n_channels, sample_rate_hz=48000, symbol_rate_hz=4800, levels=4, lpf_profile=LPF_P25_C4FM,
                 input_format=IN_CU8, block_len=8192, squelch_level=0.0):
        self.cfg = FrontEndConfig(n_channels, sample_rate_hz, symbol_rate_hz, levels, lpf_profile, input_format,
                                  block_len, squelch_level)
        self.h = C.c_void_p()
        _check(lib().ddn_batch_create(C.byref(self.cfg), C.byref(self.h)), "ddn_batch_create")

    def close(self):
        if self.h:
            lib().ddn_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, stream=None):
        _check(lib().ddn_batch_reset(self.h, stream), "ddn_batch_reset")

    def taps(self):
        import numpy as np
        t = np.zeros(160, np.float32)
        n = _check(lib().ddn_batch_get_taps(self.h, t.ctypes.data, 160), "ddn_batch_get_taps")
        return t[:n].copy()

    def run_device(self, d_iq_ptr, n, d_out_ptr, stream=None):
        _check(lib().ddn_front_end_run(self.h, d_iq_ptr, n, d_out_ptr, stream), "ddn_front_end_run")

    def set_decimation(self, passes):
        _check(lib().ddn_batch_set_decimation(self.h, passes), "ddn_batch_set_decimation")
        self.passes = passes

    def set_iq_conditioning(self, dc_enable=0, dc_shift=11, bal_enable=0, bal_thr=0.0, bal_ema_a=0.0):
        _check(lib().ddn_batch_set_iq_conditioning(self.h, dc_enable, dc_shift, bal_enable, bal_thr, bal_ema_a),
               "ddn_batch_set_iq_conditioning")

    def run_host(self, iq, n):
        """iq: numpy [B, n, 2] uint8 or float32 (channel-major).  Returns float32 [B, n >> passes]."""
        import numpy as np
        iq = np.ascontiguousarray(iq)
        out = np.empty((self.cfg.n_channels, n >> getattr(self, "passes", 0)), np.float32)
        _check(lib().ddn_front_end_run_host(self.h, iq.ctypes.data, n, out.ctypes.data), "ddn_front_end_run_host")
        return out

    def fsk_state(self, ch):
        import numpy as np
        s = np.zeros(5, np.float32)
        _check(lib().ddn_batch_get_fsk_state(self.h, ch, s.ctypes.data), "ddn_batch_get_fsk_state")
        return s

    def set_timing(self, on):
        _check(lib().ddn_batch_set_timing(self.h, 1 if on else 0), "ddn_batch_set_timing")

    def timing(self):
        import numpy as np
        t = np.zeros(3, np.float32)
        _check(lib().ddn_batch_get_timing(self.h, t.ctypes.data), "ddn_batch_get_timing")
        return t


class P25RxConfig(C.Structure):
    """ddn_p25_rx_config (include/ddn_hip.h)."""
    _fields_ = [("n_channels", C.c_int), ("out_rate_hz", C.c_int), ("sym_rate_hz", C.c_int),
                ("lock_symbols", C.c_int), ("use_matched_filter", C.c_int)]


class P25Rx:
    """Batched fixed-protocol P25p1 receive loop (ddn_p25_rx_*), host-buffer convenience wrapper."""

    def __init__(self, n_channels, out_rate=48000, sym_rate=4800, lock_symbols=840, use_matched_filter=1,
                 channels_per_wave=0, handlers=False, max_events=0, filter_in_loop=False, debug_flags=0):
        """handlers=True: the reference's per-DUID handlers decide the in-frame length (lock_symbols unused); run() then
        also leaves .events int32 [B, max_events, 4] / .n_events int32 [B] of the call.  filter_in_loop: ddn_p25_rx_set_filter_in_loop"""
        import numpy as np
        self.np = np
        self.B = n_channels
        cfg = P25RxConfig(n_channels, out_rate, sym_rate, max(lock_symbols, 0), use_matched_filter)
        self.h = C.c_void_p()
        rc = lib().ddn_p25_rx_create(C.byref(cfg), C.byref(self.h))
        _check(-abs(rc), "ddn_p25_rx_create")
        if channels_per_wave:
            assert lib().ddn_p25_rx_set_channels_per_wave(self.h, channels_per_wave) == 0
        self.handlers = bool(handlers)
        self.max_events = max_events or 4096
        if handlers:
            assert lib().ddn_p25_rx_set_handlers(self.h, 1, 64) == 0
        if filter_in_loop:
            assert lib().ddn_p25_rx_set_filter_in_loop(self.h, 1) == 0
        if debug_flags:
            assert lib().ddn_p25_rx_set_debug_flags(self.h, debug_flags) == 0

    def run(self, disc):
        """disc float32 [B, n] -> (records uint8 [B, max_sym, 10], flags uint8 [B, max_sym], counts int32 [B])."""
        np = self.np
        disc = np.ascontiguousarray(disc, np.float32)
        n = disc.shape[1]
        ms = lib().ddn_p25_rx_max_symbols(self.h, n)
        rec = np.zeros((self.B, ms, 10), np.uint8)
        fl = np.zeros((self.B, ms), np.uint8)
        cnt = np.zeros(self.B, np.int32)
        if self.handlers:
            self.events = np.zeros((self.B, self.max_events, 4), np.int32)
            self.n_events = np.zeros(self.B, np.int32)
            self.event_data = np.zeros((self.B, self.max_events, 4), np.int32)
            rc = lib().ddn_p25_rx_run_host_ev(self.h, disc.ctypes.data, n, rec.ctypes.data, fl.ctypes.data, cnt.ctypes.data, ms, self.events.ctypes.data,
                   self.n_events.ctypes.data, self.max_events, self.event_data.ctypes.data)
        else:
            rc = lib().ddn_p25_rx_run_host(self.h, disc.ctypes.data, n, rec.ctypes.data, fl.ctypes.data, cnt.ctypes.data, ms)
        _check(-abs(rc), "ddn_p25_rx_run_host")
        return rec, fl, cnt

    def thresholds(self, ch):
        t = self.np.zeros(7, self.np.float32)
        assert lib().ddn_p25_rx_get_thresholds(self.h, ch, t.ctypes.data) == 0
        return t

    def __del__(self):
        try:
            lib().ddn_p25_rx_destroy(self.h)
        except Exception:
            pass


class CqpskConfig(C.Structure):
    """ddn_cqpsk_config (include/ddn_hip.h)."""
    _fields_ = [("n_channels", C.c_int), ("sample_rate_hz", C.c_int), ("symbol_rate_hz", C.c_int),
                ("lpf_profile", C.c_int), ("lpf_enable", C.c_int), ("input_format", C.c_int), ("block_len", C.c_int),
                ("ted_gain", C.c_float)]


class CqpskBatch:
    """Batched CQPSK front end (ddn_cqpsk_*), host-buffer convenience wrapper."""

    def __init__(self, n_channels, rate=24000, sym_rate=4800, profile=5, lpf_enable=1, input_format=IN_CF32,
                 block_len=4096, ted_gain=0.0):
        self.B = n_channels
        self.cfg = CqpskConfig(n_channels, rate, sym_rate, profile, lpf_enable, input_format, block_len, ted_gain)
        self.h = C.c_void_p()
        _check(-abs(lib().ddn_cqpsk_batch_create(C.byref(self.cfg), C.byref(self.h))), "ddn_cqpsk_batch_create")

    def run(self, iq):
        """iq [B, n, 2] -> (symbols float32 [B, stride], counts int32 [B])."""
        import numpy as np
        iq = np.ascontiguousarray(iq)
        n = iq.shape[1]
        stride = lib().ddn_cqpsk_max_symbols(self.h, n)
        sym = np.zeros((self.B, stride), np.float32)
        cnt = np.zeros(self.B, np.int32)
        _check(-abs(lib().ddn_cqpsk_run_host(self.h, iq.ctypes.data, n, sym.ctypes.data, stride, cnt.ctypes.data)),
               "ddn_cqpsk_run_host")
        return sym, cnt

    def state(self, ch):
        import numpy as np
        s = np.zeros(8, np.float32)
        _check(lib().ddn_cqpsk_get_state(self.h, ch, s.ctypes.data), "ddn_cqpsk_get_state")
        return s

    def __del__(self):
        try:
            lib().ddn_cqpsk_batch_destroy(self.h)
        except Exception:
            pass


CQ_P25P1, CQ_P25P2 = 0, 1


class P25P2ChainConfig(C.Structure):  # == ddn_p25p2_chain_config (include/ddn_chain.h)
    _fields_ = [("n_channels", C.c_int), ("samples_per_call", C.c_int), ("block_len", C.c_int), ("input_format", C.c_int),
                ("sample_rate_hz", C.c_int), ("vocoder", C.c_int), ("max_groups", C.c_int), ("snr_cqpsk_db", C.c_float)]


class P25P2ChainResults(C.Structure):  # == ddn_p25p2_chain_results
    _fields_ = [("stride_symbols", C.c_size_t), ("carry_symbols", C.c_int), ("max_groups", C.c_int), ("voice_frames", C.c_int)] + [
        (k, C.c_void_p) for k in ("d_records10", "d_flags", "d_new", "d_counts", "d_n_groups", "d_group_pos", "d_dropped_syncs", "d_info",
                                  "d_payload", "d_ambe_fr", "d_ambe_rel", "d_ess", "d_voice_src", "d_voice_count", "d_voice_bits",
                                  "d_voice_result", "d_pcm")]


class P25P2ChainC:
    """ddn_p25p2_chain: I/Q of B Phase 2 channels -> MAC PDUs + PCM of both logical channels, one C call per batch of samples"""

    def __init__(self, seeds44, samples_per_call, block_len=8192, input_format=0, vocoder=1, max_groups=0):
        import numpy as np
        self.np = np
        self.B, self.n = len(seeds44), samples_per_call
        cfg = P25P2ChainConfig(self.B, samples_per_call, block_len, input_format, 0, vocoder, max_groups, 0.0)
        seeds = np.ascontiguousarray(seeds44, np.uint64)
        self.h = C.c_void_p()
        _check(lib().ddn_p25p2_chain_create(C.byref(cfg), seeds.ctypes.data, C.byref(self.h)), "ddn_p25p2_chain_create")

    def run(self, d_iq_ptr, stream=None):
        _check(lib().ddn_p25p2_chain_run(self.h, d_iq_ptr, stream), "ddn_p25p2_chain_run")

    def flush(self, stream=None):
        _check(lib().ddn_p25p2_chain_flush(self.h, stream), "ddn_p25p2_chain_flush")

    def results(self):
        r = P25P2ChainResults()
        _check(lib().ddn_p25p2_chain_get_results(self.h, C.byref(r)), "ddn_p25p2_chain_get_results")
        return r

    def fetch(self, ptr, dtype, shape):
        np = self.np
        a = np.zeros(shape, dtype)
        _check(lib().ddn_device_download(a.ctypes.data, ptr, a.nbytes), "ddn_device_download")
        return a

    def close(self):
        if self.h:
            lib().ddn_p25p2_chain_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CqRxConfig(C.Structure):  # == ddn_cq_rx_config
    _fields_ = [("n_channels", C.c_int), ("protocol", C.c_int), ("lock_symbols", C.c_int), ("nid_erasure_threshold", C.c_int),
                ("snr_cqpsk_db", C.c_float)]


class CqRx:
    """ddn_cq_rx batch object with device buffers of its own (host-buffer convenience wrapper used by the tests):
    run(symbols f32 [B][n], counts) -> (rec u8 [B][n][10], flags u8 [B][n], counts i32 [B], events [B] lists of (pos, kind, a, b, data4))"""

    def __init__(self, n_channels, protocol=CQ_P25P1, lock_symbols=0, snr_db=0.0, max_events=1024):
        self.B, self.E = n_channels, max_events
        cfg = CqRxConfig(n_channels, protocol, lock_symbols, 0, snr_db)
        self.h = C.c_void_p()
        _check(lib().ddn_cq_rx_create(C.byref(cfg), C.byref(self.h)), "ddn_cq_rx_create")
        self.d_ev, self.d_nev, self.d_evd = (C.c_void_p() for _ in range(3))
        l = lib()
        _check(l.ddn_device_alloc(n_channels * max_events * 16, C.byref(self.d_ev)), "alloc")
        _check(l.ddn_device_alloc(n_channels * 4, C.byref(self.d_nev)), "alloc")
        _check(l.ddn_device_alloc(n_channels * max_events * 16, C.byref(self.d_evd)), "alloc")
        _check(l.ddn_cq_rx_set_events(self.h, self.d_ev, self.d_nev, self.d_evd, max_events), "ddn_cq_rx_set_events")

    def run(self, sym, counts=None):
        import numpy as np
        l = lib()
        sym = np.ascontiguousarray(sym, np.float32)
        B, n = sym.shape
        assert B == self.B
        d = {k: C.c_void_p() for k in ("sym", "cin", "rec", "fl", "cnt")}
        _check(l.ddn_device_alloc(max(sym.nbytes, 4), C.byref(d["sym"])), "alloc")
        _check(l.ddn_device_alloc(B * 4, C.byref(d["cin"])), "alloc")
        _check(l.ddn_device_alloc(max(B * n * 10, 4), C.byref(d["rec"])), "alloc")
        _check(l.ddn_device_alloc(max(B * n, 4), C.byref(d["fl"])), "alloc")
        _check(l.ddn_device_alloc(B * 4, C.byref(d["cnt"])), "alloc")
        if sym.nbytes:
            _check(l.ddn_device_upload(d["sym"], sym.ctypes.data, sym.nbytes), "upload")
        cin = None
        if counts is not None:
            c = np.ascontiguousarray(counts, np.int32)
            _check(l.ddn_device_upload(d["cin"], c.ctypes.data, c.nbytes), "upload")
            cin = d["cin"]
        _check(l.ddn_cq_rx_run(self.h, d["sym"], cin, n, n, d["rec"], d["fl"], d["cnt"], n, None), "ddn_cq_rx_run")
        rec, fl, cnt = np.zeros((B, n, 10), np.uint8), np.zeros((B, n), np.uint8), np.zeros(B, np.int32)
        ev, nev, evd = np.zeros((B, self.E, 4), np.int32), np.zeros(B, np.int32), np.zeros((B, self.E, 4), np.int32)
        for a, p in ((rec, d["rec"]), (fl, d["fl"]), (cnt, d["cnt"]), (ev, self.d_ev), (nev, self.d_nev), (evd, self.d_evd)):
            if a.nbytes:
                _check(l.ddn_device_download(a.ctypes.data, p, a.nbytes), "download")
        for p in d.values():
            l.ddn_device_free(p)
        events = [[(int(ev[c, k, 0]), int(ev[c, k, 1]), int(ev[c, k, 2]), int(ev[c, k, 3]), evd[c, k].copy()) for k in range(min(int(nev[c]), self.E))]
                  for c in range(B)]
        return rec, fl, cnt, events

    def state(self, ch):
        import numpy as np
        s = np.zeros(8, np.float32)
        _check(lib().ddn_cq_rx_get_state(self.h, ch, s.ctypes.data), "ddn_cq_rx_get_state")
        return s

    def close(self):
        if self.h:
            l = lib()
            l.ddn_cq_rx_destroy(self.h)
            for p in (self.d_ev, self.d_nev, self.d_evd):
                l.ddn_device_free(p)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


P25P2_SEQ_STATE_BYTES = 592   # sizeof(ddn_p25p2_seq_state): offset, fourv[2], reserved, ess_b u8 [2][96], ess_b_llr i16 [2][96]


class P25P2Groups:
    """ddn_p25p2_groups_batch with its carried per-channel state on the device (torch tensors for memory only):
    run(bits u8 [C][G][1400], llr i16 [C][G][1400]) -> (info i32 [C][G][4][8], payload u8 [..][180], ambe_fr, ambe_rel u8 [..][4][4][24],
    ess u8 [..][96]) as numpy arrays"""

    def __init__(self, seeds44, threshold=64):
        import torch
        self.torch = torch
        self.C = len(seeds44)
        self.seed = torch.tensor([int(v) for v in seeds44], dtype=torch.int64, device="cuda")
        self.state = torch.zeros((self.C, P25P2_SEQ_STATE_BYTES), dtype=torch.uint8, device="cuda")
        self.threshold = threshold

    def run_stream(self, dibits, llr2, cursor=None, max_groups=8):
        """dibits u8 [C][n], llr2 i16 [C][n][2] -> (n_groups [C], group_pos [C][max_groups], cursor_out [C], run()'s arrays):
        ddn_p25p2_sync_cut_batch feeding ddn_p25p2_groups_batch on the device"""
        import numpy as np
        torch = self.torch
        Cn, n = dibits.shape
        td = torch.from_numpy(np.ascontiguousarray(dibits, np.uint8)).cuda()
        tl = torch.from_numpy(np.ascontiguousarray(llr2, np.int16)).cuda()
        cur = None if cursor is None else torch.tensor([int(v) for v in cursor], dtype=torch.int32, device="cuda")
        ng = torch.zeros(Cn, dtype=torch.int32, device="cuda")
        gp = torch.full((Cn, max_groups), -1, dtype=torch.int32, device="cuda")
        co = torch.zeros(Cn, dtype=torch.int32, device="cuda")
        gb = torch.zeros((Cn, max_groups, 1400), dtype=torch.uint8, device="cuda")
        gl = torch.zeros((Cn, max_groups, 1400), dtype=torch.int16, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        _check(lib().ddn_p25p2_sync_cut_batch(td.data_ptr(), tl.data_ptr(), Cn, n, n, None if cur is None else cur.data_ptr(), max_groups,
                                              ng.data_ptr(), gp.data_ptr(), co.data_ptr(), gb.data_ptr(), gl.data_ptr(), st),
               "ddn_p25p2_sync_cut_batch")
        res = self.run(gb, gl, groups_of=ng)
        return ng.cpu().numpy(), gp.cpu().numpy(), co.cpu().numpy(), res, gb.cpu().numpy(), gl.cpu().numpy()

    def run(self, bits, llr, groups_of=None):
        import numpy as np
        torch = self.torch
        Cn, G = bits.shape[0], bits.shape[1]
        assert Cn == self.C and tuple(bits.shape) == tuple(llr.shape) == (Cn, G, 1400)
        tb = bits if torch.is_tensor(bits) else torch.from_numpy(np.ascontiguousarray(bits, np.uint8)).cuda()
        tl = llr if torch.is_tensor(llr) else torch.from_numpy(np.ascontiguousarray(llr, np.int16)).cuda()
        n = Cn * G * 4
        info = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
        pay = torch.zeros((n, 180), dtype=torch.uint8, device="cuda")
        fr = torch.zeros((n, 4, 4, 24), dtype=torch.uint8, device="cuda")
        rel = torch.zeros((n, 4, 4, 24), dtype=torch.uint8, device="cuda")
        ess = torch.zeros((n, 96), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        _check(lib().ddn_p25p2_groups_batch(tb.data_ptr(), tl.data_ptr(), Cn, G, None if groups_of is None else groups_of.data_ptr(),
                                            self.seed.data_ptr(), self.state.data_ptr(), self.threshold,
                                            info.data_ptr(), pay.data_ptr(), fr.data_ptr(), rel.data_ptr(), ess.data_ptr(), st),
               "ddn_p25p2_groups_batch")
        torch.cuda.synchronize()
        shp = (Cn, G, 4)
        return (info.cpu().numpy().reshape(shp + (8,)), pay.cpu().numpy().reshape(shp + (180,)), fr.cpu().numpy().reshape(shp + (4, 4, 24)),
                rel.cpu().numpy().reshape(shp + (4, 4, 24)), ess.cpu().numpy().reshape(shp + (96,)))
