"""ctypes binding of libsmilehip.so's C ABI (include/smilehip.h).

Python host mirror of the boundary: names and argument meaning follow the
reference's component options (cFramer / cVectorPreemphasis / cWindower /
cTransformFFT / cMelspec / cMfcc / cDeltaRegression). There is no CPU path in
here: if the HIP library is missing or no gfx950 device is present every
compute entry point raises SmileHipError.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsmilehip.so")

WINFUNC = {"rec": 0, "rect": 0, "han": 1, "hann": 1, "ham": 2, "hamming": 2, "gau": 3,
           "gauss": 3, "sin": 4, "sine": 4, "tri": 5, "bar": 6, "bartlett": 6, "lac": 7}


class SmileHipError(RuntimeError):
    pass


class LldConfig(C.Structure):
    """smilehip_lld_config"""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("sample_rate", C.c_double), ("frame_size_sec", C.c_double), ("frame_step_sec", C.c_double),
        ("preemph", C.c_int32), ("preemph_k", C.c_float), ("preemph_de", C.c_int32),
        ("win_func", C.c_int32), ("win_sigma", C.c_double), ("win_gain", C.c_double),
        ("win_offset", C.c_double),
        ("zero_pad_symmetric", C.c_int32),
        ("n_bands", C.c_int32), ("lofreq", C.c_float), ("hifreq", C.c_float),
        ("use_power", C.c_int32), ("mel_htk_compatible", C.c_int32),
        ("first_mfcc", C.c_int32), ("last_mfcc", C.c_int32), ("cep_lifter", C.c_float),
        ("mfcc_htk_compatible", C.c_int32), ("melfloor", C.c_float),
        ("n_delta", C.c_int32), ("delta_win", C.c_int32),
        ("chain_kind", C.c_int32), ("plp_lp_order", C.c_int32), ("plp_compression", C.c_float), ("pitch_max", C.c_double), ("voicing_cutoff", C.c_double),
        ("sma_win", C.c_int32),
        ("force_frame_size", C.c_int64), ("force_fft_frame_size_sec", C.c_double),
        ("stage_mask", C.c_uint32),
        ("pitch_min", C.c_double), ("shs_n_harmonics", C.c_int32), ("shs_compression", C.c_float),
        ("f0_min_energy", C.c_float), ("append_log_energy", C.c_int32), ("cms", C.c_int32), ("jitter_broken_thresh", C.c_int32),
        ("vit_buffer_len", C.c_int32), ("jitter_search_range", C.c_double), ("formant_max_freq", C.c_double), ("specscale_min_f", C.c_double), ("shs_n_candidates", C.c_int32), ("shs_old_peak_algo", C.c_int32),
        ("spectral_band_lo", C.c_int32 * 2), ("spectral_band_hi", C.c_int32 * 2), ("specscale_off", C.c_int32),
    ]


STAGE_WINDOW, STAGE_FFT, STAGE_MEL, STAGE_MFCC, STAGE_ALL = 1, 2, 4, 8, 15


class FuncSpec(C.Structure):
    """smilehip_func_spec (include/smilehip.h), field for field: one cFunctionals instance."""
    _fields_ = [
        ("n_fam", C.c_int32), ("fam", C.c_int32 * 12), ("non_zero_functs", C.c_int32), ("reserved0", C.c_int32),
        ("period", C.c_double),
        ("ext_mask", C.c_uint32), ("ext_norm", C.c_int32),
        ("means_mask", C.c_uint32), ("means_norm", C.c_int32),
        ("mom_mask", C.c_uint32), ("mom_stddev_norm", C.c_int32), ("mom_ratio_limit", C.c_int32), ("reserved1", C.c_int32),
        ("reg_mask", C.c_uint32), ("reg_centroid_norm", C.c_int32), ("reg_norm_coeff", C.c_int32),
        ("reg_norm_inputs", C.c_int32), ("reg_centroid_abs", C.c_int32), ("reg_centroid_limit", C.c_int32),
        ("reg_ratio_limit", C.c_int32), ("reg_old_buggy_qerr", C.c_int32),
        ("pct_mask", C.c_uint32), ("pct_interp", C.c_int32), ("n_pctl", C.c_int32), ("n_range", C.c_int32),
        ("pctl", C.c_double * 8), ("range_a", C.c_int32 * 8), ("range_b", C.c_int32 * 8),
        ("times_mask", C.c_uint32), ("times_norm", C.c_int32), ("times_buggy_sec_norm", C.c_int32), ("reserved2", C.c_int32),
        ("seg_mask", C.c_uint32), ("seg_norm", C.c_int32), ("seg_algo", C.c_int32), ("seg_max_num", C.c_int32),
        ("seg_min_lng", C.c_int32), ("seg_auto_min_lng", C.c_int32), ("seg_pause_min_lng", C.c_int32),
        ("seg_x_is_rel", C.c_int32), ("seg_n_thresholds", C.c_int32), ("seg_ravg_lng", C.c_int32),
        ("seg_x", C.c_float), ("seg_thresholds", C.c_float * 8), ("seg_range_rel_threshold", C.c_float),
        ("lpc_gain", C.c_int32), ("lpc_coeffs", C.c_int32), ("lpc_first", C.c_int32), ("lpc_order", C.c_int32),
        ("pk_mask", C.c_uint32), ("pk_norm", C.c_int32), ("pk_ratio_limit", C.c_int32), ("pk_dyn_rel", C.c_int32),
        ("pk_use_abs", C.c_int32), ("reserved5", C.c_int32),
        ("pk_rel_thresh", C.c_float), ("pk_abs_thresh", C.c_float),
        ("ons_mask", C.c_uint32), ("ons_norm", C.c_int32), ("ons_use_abs", C.c_int32), ("reserved6", C.c_int32),
        ("ons_thr_on", C.c_float), ("ons_thr_off", C.c_float),
        ("pko_mask", C.c_uint32), ("pko_norm", C.c_int32),
        ("crs_mask", C.c_uint32), ("dct_first", C.c_int32), ("dct_last", C.c_int32), ("n_samples", C.c_int32),
        ("sample_pos", C.c_double * 8),
        ("n_quot", C.c_int32), ("quot_a", C.c_int32 * 8), ("quot_b", C.c_int32 * 8),
        ("n_ul", C.c_int32), ("n_dl", C.c_int32), ("reserved7", C.c_int32),
        ("ul", C.c_double * 8), ("dl", C.c_double * 8),
        ("mod_win_frames", C.c_int32), ("mod_step_frames", C.c_int32), ("mod_n_bins", C.c_int32), ("mod_win_func", C.c_int32),
        ("mod_remove_nz_mean", C.c_int32), ("reserved8", C.c_int32),
        ("mod_min_freq", C.c_double), ("mod_max_freq", C.c_double),
    ]


class SpectralOpts(C.Structure):
    """smilehip_spectral_opts (include/smilehip.h): cSpectral's general option set"""
    _fields_ = [("n_bands", C.c_int32), ("band_lo", C.c_int32 * 16), ("band_hi", C.c_int32 * 16), ("n_rolloff", C.c_int32),
                ("rolloff", C.c_double * 16)] + [(k, C.c_int32) for k in ("flux", "centroid", "max_pos", "min_pos", "entropy", "variance",
                                                                          "skewness", "kurtosis", "slope", "sharpness", "harmonicity", "flatness",
                                                                          "log_flatness", "spec_diff", "spec_pos_diff", "flux_centroid",
                                                                          "flux_at_flux_centroid", "standard_deviation", "n_slopes")] + [
                   ("slope_lo", C.c_int32 * 16), ("slope_hi", C.c_int32 * 16)]


def spectral_opts(bands, rolloff=(0.25, 0.5, 0.75, 0.9), slopes=(), **flags):
    o = SpectralOpts()
    o.n_slopes = len(slopes)
    for i, (a, b) in enumerate(slopes):
        o.slope_lo[i], o.slope_hi[i] = a, b
    o.n_bands = len(bands)
    for i, (a, b) in enumerate(bands):
        o.band_lo[i], o.band_hi[i] = a, b
    o.n_rolloff = len(rolloff)
    for i, r in enumerate(rolloff):
        o.rolloff[i] = r
    for k, v in flags.items():
        setattr(o, k, int(v))
    return o


class Geometry(C.Structure):
    """smilehip_geometry"""
    _fields_ = [("frame_size", C.c_int64), ("frame_step", C.c_int64), ("fft_size", C.c_int64),
                ("n_bins", C.c_int64), ("n_static", C.c_int32), ("n_out", C.c_int32),
                ("frame_period", C.c_double), ("fft_frame_size_sec", C.c_double)]


# every symbol include/smilehip.h declares: (restype, argtypes)
_vp, _i32, _i64, _f32, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
SYMBOLS = {
    "smilehip_version": (C.c_int, []),
    "smilehip_last_error": (C.c_char_p, []),
    "smilehip_init": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "smilehip_shutdown": (None, [_vp]),
    "smilehip_device_name": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "smilehip_config_is09_lld": (None, [C.POINTER(LldConfig)]),
    "smilehip_config_compare16_ab": (None, [C.POINTER(LldConfig)]),
    "smilehip_config_compare16_f0": (None, [C.POINTER(LldConfig)]),
    "smilehip_config_htk_variant": (C.c_int, [C.POINTER(LldConfig), C.c_char_p]),
    "smilehip_config_compare16": (None, [C.POINTER(LldConfig)]),
    "smilehip_specscale_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_pitchshs_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_batch_f0_taps": (C.c_int, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "smilehip_config_plp_0_d_a": (None, [C.POINTER(LldConfig)]),
    "smilehip_batch_total_rows": (_i64, [_vp]),
    "smilehip_batch_delta_fused": (C.c_int, [_vp]),
    "smilehip_functionals_is09_mask": (C.c_uint32, []),
    "smilehip_functionals_count": (C.c_int, [C.c_uint32]),
    "smilehip_functionals_matrix": (C.c_int, [_vp, _vp, _i64, _i64, C.c_int32, C.c_uint32, _vp, _vp]),
    "smilehip_batch_func_rows": (C.c_int, [_vp, _vp]),
    "smilehip_batch_functionals": (C.c_int, [_vp, _vp, _vp, _i64, C.c_uint32, _vp, _i64, _vp]),
    "smilehip_functionals_compare16_count": (C.c_int, []),
    "smilehip_batch_functionals_compare16": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "smilehip_batch_compare_b_extra": (C.c_int, [_vp, _vp]),
    "smilehip_funcspec_count": (C.c_int, [_vp]),
    "smilehip_funcspec_compare16": (C.c_int, [C.c_char_p, _vp]),
    "smilehip_funcspec_is13_compare": (C.c_int, [C.c_char_p, _vp]),
    "smilehip_batch_functionals_is13_compare": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "smilehip_config_is13_compare": (None, [_vp]),
    "smilehip_config_egemapsv02": (None, [C.POINTER(LldConfig)]),
    "smilehip_config_egemapsv01a": (None, [C.POINTER(LldConfig)]),
    "smilehip_funcspec_egemaps": (C.c_int, [C.c_char_p, _vp]),
    "smilehip_functionals_egemaps_count": (C.c_int, []),
    "smilehip_batch_functionals_egemaps": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "smilehip_batch_egemaps_taps": (C.c_int, [_vp] + [C.POINTER(_vp)] * 9 + [_vp]),
    "smilehip_spectral_gemaps_frames": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, _vp, _i64, _i64, _vp]),
    "smilehip_specresample_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_lpc_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_formantlpc_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_harmonics_frames": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_valbased_select_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _f32, _i32, _i32, _i32, _i32, _f32, _vp, _i64, _vp, _vp]),
    "smilehip_jitter_stream_create": (C.c_int, [_vp, _dbl, _i64, _i64, _dbl, _dbl, _i32, C.POINTER(_vp)]),
    "smilehip_jitter_stream_push": (C.c_int, [_vp, _f32, _vp, _i64, _i64, _vp, _vp, _vp]),
    "smilehip_jitter_stream_set_time_offset": (C.c_int, [_vp, _i64]),
    "smilehip_jitter_stream_destroy": (C.c_int, [_vp]),
    "smilehip_viterbi_stream_create": (C.c_int, [_vp, _i32, _f32, _vp, C.POINTER(_vp)]),
    "smilehip_viterbi_stream_push": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32]),
    "smilehip_viterbi_stream_set_candidates": (C.c_int, [_vp, _i32]),
    "smilehip_viterbi_stream_flush": (C.c_int, [_vp, _vp, _vp, _vp, _i32]),
    "smilehip_viterbi_stream_destroy": (C.c_int, [_vp]),
    "smilehip_funcspec_matrix": (C.c_int, [_vp, _vp, _vp, _i64, _i64, C.c_int32, _vp, _vp]),
    "smilehip_batch_funcspec": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.c_int32, C.c_int32, C.c_int32, _vp, _i64, _vp, _i64, _vp]),
    "smilehip_lld_run": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "smilehip_lld_run_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "smilehip_lld_run_host": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "smilehip_alloc": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    "smilehip_free": (C.c_int, [_vp, _vp]),
    "smilehip_alloc_host": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    "smilehip_free_host": (C.c_int, [_vp, _vp]),
    "smilehip_htk_rows_be": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "smilehip_copy_to_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp]),
    "smilehip_copy_to_host": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp]),
    "smilehip_alloc_cache": (C.c_int, [_vp, C.c_uint64]),
    "smilehip_stream_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "smilehip_stream_destroy": (C.c_int, [_vp, _vp]),
    "smilehip_event_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "smilehip_event_destroy": (C.c_int, [_vp, _vp]),
    "smilehip_event_record": (C.c_int, [_vp, _vp, _vp]),
    "smilehip_stream_wait_event": (C.c_int, [_vp, _vp, _vp]),
    "smilehip_kernel_timing": (C.c_int, [C.c_int]),
    "smilehip_kernel_timing_report": (C.c_int64, [_vp, _i64]),
    "smilehip_copy_to_device_2d": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint64, C.c_uint64, _vp]),
    "smilehip_copy_to_host_2d": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint64, C.c_uint64, _vp]),
    "smilehip_host_register": (C.c_int, [_vp, _vp, C.c_uint64]),
    "smilehip_host_unregister": (C.c_int, [_vp, _vp]),
    "smilehip_frame_rows": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "smilehip_window_op_block": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i32, C.c_int, C.c_int, C.c_int, _vp]),
    "smilehip_delta_segments_block": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, C.c_int, C.c_int, _vp, _vp]),
    "smilehip_pitchacf_contour_frames": (C.c_int, [_vp, _vp, _vp, _dbl, _dbl, _vp, _vp, _i64, _vp]),
    "smilehip_viterbi_stream_push_frames": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32]),
    "smilehip_jitter_stream_push_frames": (C.c_int, [_vp, _vp, _i32, _vp, _i64, _i64, _vp, _vp, _vp]),
    "smilehip_stream_synchronize": (C.c_int, [_vp, _vp]),
    "smilehip_config_mfcc12_0_d_a": (None, [C.POINTER(LldConfig)]),
    "smilehip_plan_create": (C.c_int, [_vp, C.POINTER(LldConfig), C.POINTER(_vp)]),
    "smilehip_plan_create_host_only": (C.c_int, [C.POINTER(LldConfig), C.POINTER(_vp)]),
    "smilehip_plan_destroy": (None, [_vp]),
    "smilehip_plan_geometry": (C.c_int, [_vp, C.POINTER(Geometry)]),
    "smilehip_num_frames": (_i64, [_vp, _i64]),
    "smilehip_frame_time": (_dbl, [_vp, _i64]),
    "smilehip_row_time": (_dbl, [_vp, _i64, _i64]),
    "smilehip_plan_get_window": (_i64, [_vp, _vp, _i64]),
    "smilehip_plan_get_mel_weights": (_i64, [_vp, _vp, _i64]),
    "smilehip_plan_get_mel_chanmap": (_i64, [_vp, _vp, _i64]),
    "smilehip_plan_get_dct": (_i64, [_vp, _vp, _i64]),
    "smilehip_plan_get_lifter": (_i64, [_vp, _vp, _i64]),
    "smilehip_batch_f0_pending": (C.c_int, [_vp, _vp]),
    "smilehip_batch_create": (C.c_int, [_vp, _vp, _i32, C.POINTER(_vp)]),
    "smilehip_batch_destroy": (None, [_vp]),
    "smilehip_batch_total_frames": (_i64, [_vp]),
    "smilehip_batch_frame_offsets": (C.c_int, [_vp, _vp]),
    "smilehip_mfcc_run": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "smilehip_mfcc_run_host": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "smilehip_plan_set_timing": (C.c_int, [_vp, C.c_int]),
    "smilehip_plan_last_timing": (C.c_int, [_vp, C.POINTER(_f32), C.POINTER(_f32)]),
    "smilehip_sumsq_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "smilehip_zcr_count_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "smilehip_acf_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "smilehip_pitchacf_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _dbl, _dbl, _vp, _vp, _vp]),
    "smilehip_pitchacf_contour_step": (C.c_int, [_vp, _vp, _vp, _dbl, _dbl, _vp, _vp, _vp]),
    "smilehip_spectral_frames": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, _vp, _i64, _i64, _vp]),
    "smilehip_spectral_opts_count": (C.c_int, [_vp]),
    "smilehip_spectral_op_create": (C.c_int, [_vp, _vp, _i64, C.c_double, _vp]),
    "smilehip_spectral_op_n_out": (C.c_int, [_vp]),
    "smilehip_spectral_op_frames": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, _vp, _i64, _i64, _vp]),
    "smilehip_spectral_op_destroy": (C.c_int, [_vp]),
    "smilehip_plp_audspec_frames": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, C.c_float, C.c_float, C.c_int, _vp, _vp, _vp, _i64,
                                             _i64, _vp]),
    "smilehip_plp_cc_frames": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, C.c_float, C.c_float, C.c_int, _vp, _vp, _vp, _i64, _i64,
                                        _vp]),
    "smilehip_mfcc_inverse_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, C.c_int, _vp]),
    "smilehip_plp_stage_frames": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, C.c_float, C.c_float, C.c_int, _vp, C.c_int, _vp, _i64, _i64,
                                           _vp]),
    "smilehip_window_op_row": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp]),
    "smilehip_window_op_row_ex": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp, _vp]),
    "smilehip_delta_op_row": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp, _vp]),
    "smilehip_pcm_convert": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _i64, _vp, _vp]),
    "smilehip_pcm_convert_float": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _i64, _vp, _vp]),
    "smilehip_pcm16_to_float": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "smilehip_preemphasis_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _f32, C.c_int, _vp]),
    "smilehip_window_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_rfft_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_fftmag_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_irfft_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_fftmagphase_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _f32, _f32, _vp, _i64, _i64, _vp]),
    "smilehip_melspec_table_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _i64, _i64, _vp]),
    "smilehip_melspec_inverse_table_frames": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _i64, _i64, _vp]),
    "smilehip_pitchacf_zcr_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _dbl, _dbl, _vp, _vp]),
    "smilehip_mzcr_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _vp]),
    "smilehip_melspec_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_mfcc_frames": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "smilehip_delta_chain": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "smilehip_specresample_geometry": (C.c_int, [_i64, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _vp]),
    "smilehip_specresample_tables": (C.c_int, [_i64, _i64, _i64, _dbl, _vp, _vp]),
    "smilehip_specresample_table_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp]),
    "smilehip_lpc_acf_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp]),
    "smilehip_lsp_frames": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _i64, _vp]),
    "smilehip_intensity_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp]),
    "smilehip_vecop_frames": (C.c_int, [_vp, _i32, _f32, _f32, _vp, _i64, _i32, _vp, _i64, _i64, _vp]),
    "smilehip_pitch_smoother_rows": (C.c_int, [_vp, _i32, _f32, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _i64, _vp, _i32, _vp, _i64, _vp, _vp]),
}

_lib = None


def load():
    """Load libsmilehip.so and bind every declared symbol. Raises if the
    library has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        path = os.environ.get("SMILEHIP_LIB", LIB_PATH)     # developer override (instrumented builds)
        if not os.path.exists(path):
            raise SmileHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C opensmile_amd/csrc). libsmilehip has no CPU fallback.")
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise SmileHipError(f"smilehip error {rc}: {load().smilehip_last_error().decode()}")


def plp_0_d_a_config():
    c = LldConfig()
    load().smilehip_config_plp_0_d_a(C.byref(c))
    return c


def compare16_ab_config():
    c = LldConfig()
    load().smilehip_config_compare16_ab(C.byref(c))
    return c


def htk_variant_config(name):
    """One of the eight files of config/mfcc and config/plp: MFCC12_{0,E}_D_A[_Z], PLP_{0,E}_D_A[_Z]."""
    c = LldConfig()
    _check(load().smilehip_config_htk_variant(C.byref(c), name.encode()))
    return c


def compare16_config():
    c = LldConfig()
    load().smilehip_config_compare16(C.byref(c))
    return c


def is13_compare_config():
    c = LldConfig()
    load().smilehip_config_is13_compare(C.byref(c))
    return c


def egemapsv02_config():
    """config/egemaps/v02/eGeMAPSv02.conf (BASELINE config 5): the 25-column LLD level."""
    c = LldConfig()
    load().smilehip_config_egemapsv02(C.byref(c))
    return c


def egemapsv01a_config():
    """The eGeMAPSv02 graph with the v01a files' option values (GeMAPSv01a.conf / eGeMAPSv01a.conf write column subsets of it)."""
    c = LldConfig()
    load().smilehip_config_egemapsv01a(C.byref(c))
    return c


def funcspec_egemaps(instance):
    """One of eGeMAPSv02's cFunctionals instances: F0, Loudness, MVZ, MVV, MU, numPeaks, segF0, segF0pause, leq."""
    s = FuncSpec()
    _check(load().smilehip_funcspec_egemaps(instance.encode(), C.byref(s)))
    return s


def rows_op_host(ctx, fn, handle, inputs, out_cols, pre=(), post=()):
    """Run a rows-in / rows-out operator on host matrices: inputs = list of 2-D float32 arrays (each uploaded, passed as
    pointer + leading dimension), output n x out_cols. pre / post: extra scalar arguments before / after."""
    L = load()
    inputs = [np.ascontiguousarray(a, dtype=np.float32) for a in inputs]
    n = inputs[0].shape[0]
    out = np.zeros((n, out_cols), np.float32)
    ptrs = []
    d_out = _vp()
    try:
        for a in inputs:
            d = _vp()
            _check(L.smilehip_alloc(ctx._h, max(a.nbytes, 4), C.byref(d)))
            ptrs.append(d)
            if a.nbytes:
                _check(L.smilehip_copy_to_device(ctx._h, d, a.ctypes.data, a.nbytes, None))
        _check(L.smilehip_alloc(ctx._h, max(out.nbytes, 4), C.byref(d_out)))
        args = [handle] + list(pre)
        for a, d in zip(inputs, ptrs):
            args += [d] + ([a.shape[1]] if a.ndim == 2 else [])
        args += [d_out, out_cols, n] + list(post) + [None]
        _check(fn(*args))
        _check(L.smilehip_stream_synchronize(ctx._h, None))
        if out.nbytes:
            _check(L.smilehip_copy_to_host(ctx._h, out.ctypes.data, d_out, out.nbytes, None))
    finally:
        for d in ptrs:
            L.smilehip_free(ctx._h, d)
        L.smilehip_free(ctx._h, d_out)
    return out


def specresample_host(plan, spec):
    return rows_op_host(plan.ctx, load().smilehip_specresample_frames, plan._h, [spec], 220)


def lpc_host(plan, x):
    return rows_op_host(plan.ctx, load().smilehip_lpc_frames, plan._h, [x], 11)


def formantlpc_host(plan, lpc):
    return rows_op_host(plan.ctx, load().smilehip_formantlpc_frames, plan._h, [lpc], 10)


def harmonics_host(plan, f0, formants, mag):
    f0 = np.ascontiguousarray(f0, dtype=np.float32).reshape(-1)
    return rows_op_host(plan.ctx, load().smilehip_harmonics_frames, plan._h, [f0, formants, mag], 6)


def spectral_gemaps_host(plan, mag):
    """The frames of one stream (rows of mag, K = 257) -> n x 5."""
    L = load()
    mag = np.ascontiguousarray(mag, dtype=np.float32)
    n, K = mag.shape
    out = np.zeros((n, 5), np.float32)
    d_in, d_out, d_st = _vp(), _vp(), _vp()
    ctx = plan.ctx
    _check(L.smilehip_alloc(ctx._h, max(mag.nbytes, 4), C.byref(d_in)))
    _check(L.smilehip_alloc(ctx._h, max(out.nbytes, 4), C.byref(d_out)))
    _check(L.smilehip_alloc(ctx._h, K * 4, C.byref(d_st)))
    try:
        if mag.nbytes:
            _check(L.smilehip_copy_to_device(ctx._h, d_in, mag.ctypes.data, mag.nbytes, None))
        half = n // 2                       # two calls: the state buffer carries the flux across them
        _check(L.smilehip_spectral_gemaps_frames(plan._h, d_in, K, d_st, 1, d_out, 5, half, None))
        _check(L.smilehip_spectral_gemaps_frames(plan._h, d_in.value + half * K * 4, K, d_st, 1 if half == 0 else 0,
                                                 d_out.value + half * 5 * 4, 5, n - half, None))
        _check(L.smilehip_stream_synchronize(ctx._h, None))
        if out.nbytes:
            _check(L.smilehip_copy_to_host(ctx._h, out.ctypes.data, d_out, out.nbytes, None))
    finally:
        L.smilehip_free(ctx._h, d_in)
        L.smilehip_free(ctx._h, d_out)
        L.smilehip_free(ctx._h, d_st)
    return out


def funcspec_is13_compare(instance):
    s = FuncSpec()
    _check(load().smilehip_funcspec_is13_compare(instance.encode(), C.byref(s)))
    return s


def compare16_f0_config():
    c = LldConfig()
    load().smilehip_config_compare16_f0(C.byref(c))
    return c


def funcspec_compare16(instance):
    """One of ComParE_2016's six cFunctionals instances: "A", "B", "F0", "Nz", "LLD", "Delta"."""
    s = FuncSpec()
    _check(load().smilehip_funcspec_compare16(instance.encode(), C.byref(s)))
    return s


def funcspec_count(spec):
    n = load().smilehip_funcspec_count(C.byref(spec))
    if n < 0:
        _check(n)
    return n


def funcspec_matrix_host(ctx, spec, x):
    """rows x cols host matrix -> cols x count(spec) functionals through smilehip_funcspec_matrix."""
    L = load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, cols = x.shape
    per = funcspec_count(spec)
    out = np.zeros((cols, per), np.float32)
    d_x, d_out = _vp(), _vp()
    _check(L.smilehip_alloc(ctx._h, max(x.nbytes, 4), C.byref(d_x)))
    _check(L.smilehip_alloc(ctx._h, max(out.nbytes, 4), C.byref(d_out)))
    try:
        _check(L.smilehip_copy_to_device(ctx._h, d_x, x.ctypes.data, x.nbytes, None))
        _check(L.smilehip_funcspec_matrix(ctx._h, C.byref(spec), d_x, cols, rows, cols, d_out, None))
        _check(L.smilehip_stream_synchronize(ctx._h, None))
        _check(L.smilehip_copy_to_host(ctx._h, out.ctypes.data, d_out, out.nbytes, None))
    finally:
        L.smilehip_free(ctx._h, d_x)
        L.smilehip_free(ctx._h, d_out)
    return out


def is09_lld_config():
    c = LldConfig()
    load().smilehip_config_is09_lld(C.byref(c))
    return c


def mfcc12_0_d_a_config():
    c = LldConfig()
    load().smilehip_config_mfcc12_0_d_a(C.byref(c))
    return c


class Context:
    def __init__(self, device=0):
        self._h = _vp()
        _check(load().smilehip_init(device, C.byref(self._h)))

    def name(self):
        buf = C.create_string_buffer(256)
        _check(load().smilehip_device_name(self._h, buf, 256))
        return buf.value.decode()

    def close(self):
        if self._h:
            load().smilehip_shutdown(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """Configured chain (tables resident on the device)."""

    def __init__(self, ctx, cfg=None):
        self.ctx = ctx
        self.cfg = cfg if cfg is not None else mfcc12_0_d_a_config()
        self._h = _vp()
        if ctx is None:      # host-only: tables + geometry, compute entry points refuse it
            _check(load().smilehip_plan_create_host_only(C.byref(self.cfg), C.byref(self._h)))
        else:
            _check(load().smilehip_plan_create(ctx._h, C.byref(self.cfg), C.byref(self._h)))
        g = Geometry()
        _check(load().smilehip_plan_geometry(self._h, C.byref(g)))
        self.geometry = g

    def num_frames(self, n_samples):
        return int(load().smilehip_num_frames(self._h, n_samples))

    def frame_time(self, t):
        return float(load().smilehip_frame_time(self._h, t))

    def _table(self, fn, dtype):
        n = fn(self._h, None, 0)
        a = np.zeros(n, dtype)
        fn(self._h, a.ctypes.data, n)
        return a

    def window(self):
        return self._table(load().smilehip_plan_get_window, np.float32)

    def mel_weights(self):
        return self._table(load().smilehip_plan_get_mel_weights, np.float32)

    def mel_chanmap(self):
        return self._table(load().smilehip_plan_get_mel_chanmap, np.int32)

    def dct(self):
        g = self.geometry
        return self._table(load().smilehip_plan_get_dct, np.float32).reshape(g.n_static, self.cfg.n_bands)

    def lifter(self):
        return self._table(load().smilehip_plan_get_lifter, np.float32)

    def set_timing(self, on=True):
        _check(load().smilehip_plan_set_timing(self._h, 1 if on else 0))

    def last_timing(self):
        a, b = _f32(), _f32()
        _check(load().smilehip_plan_last_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self._h:
            load().smilehip_plan_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """A packed set of utterances (sample offsets) prepared for a plan."""

    def __init__(self, plan, sample_offsets):
        self.plan = plan
        off = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        assert off.ndim == 1 and len(off) >= 1
        self.n_utt = len(off) - 1
        self.sample_offsets = off
        self._h = _vp()
        _check(load().smilehip_batch_create(plan._h, off.ctypes.data, self.n_utt, C.byref(self._h)))
        self.total_frames = int(load().smilehip_batch_total_frames(self._h))
        self.total_rows = int(load().smilehip_batch_total_rows(self._h))
        self.delta_fused = bool(load().smilehip_batch_delta_fused(self._h))
        fo = np.zeros(self.n_utt + 1, np.int64)
        _check(load().smilehip_batch_frame_offsets(self._h, fo.ctypes.data))
        self.frame_offsets = fo

    def run_device(self, d_pcm_ptr, d_out_ptr, ld_out, stream=None):
        """Device pointers (ints). Asynchronous on `stream` (a hipStream_t as int)."""
        _check(load().smilehip_lld_run(self.plan._h, self._h, d_pcm_ptr, d_out_ptr, ld_out, stream))

    def run_device_f32(self, d_pcm_f32_ptr, d_out_ptr, ld_out, stream=None):
        """The same on float samples (the output of smilehip_pcm_convert / smilehip_pcm16_to_float)."""
        _check(load().smilehip_lld_run_f32(self.plan._h, self._h, d_pcm_f32_ptr, d_out_ptr, ld_out, stream))

    def run_host(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        out = np.zeros((self.total_rows, self.plan.geometry.n_out), np.float32)
        _check(load().smilehip_lld_run_host(self.plan._h, self._h, pcm.ctypes.data, len(pcm),
                                            out.ctypes.data))
        return out

    def f0_run_host_taps(self, pcm):
        """F0 chain plans: run on host data and also return the intermediate levels
        {hps: total_frames x n_bins, shs: total_frames x 21, e60: total_frames} next to the output matrix."""
        L = load()
        ctx = self.plan.ctx._h
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        K = self.plan.geometry.n_bins
        nf = self.total_frames
        out = np.zeros((self.total_rows, self.plan.geometry.n_out), np.float32)
        taps = {"hps": np.zeros((nf, K), np.float32), "shs": np.zeros((nf, 21), np.float32), "e60": np.zeros((nf, 1), np.float32)}
        d_pcm, d_out, d_hps = _vp(), _vp(), _vp()
        _check(L.smilehip_alloc(ctx, max(pcm.nbytes, 4), C.byref(d_pcm)))
        _check(L.smilehip_alloc(ctx, max(out.nbytes, 4), C.byref(d_out)))
        _check(L.smilehip_alloc(ctx, max(taps["hps"].nbytes, 4), C.byref(d_hps)))
        try:
            if pcm.nbytes:
                _check(L.smilehip_copy_to_device(ctx, d_pcm, pcm.ctypes.data, pcm.nbytes, None))
            d_shs, d_e60 = _vp(), _vp()
            _check(L.smilehip_batch_f0_taps(self._h, d_hps, C.byref(d_shs), C.byref(d_e60)))
            _check(L.smilehip_lld_run(self.plan._h, self._h, d_pcm, d_out, self.plan.geometry.n_out, None))
            _check(L.smilehip_stream_synchronize(ctx, None))
            if out.nbytes:
                _check(L.smilehip_copy_to_host(ctx, out.ctypes.data, d_out, out.nbytes, None))
            if nf:
                _check(L.smilehip_copy_to_host(ctx, taps["hps"].ctypes.data, d_hps, taps["hps"].nbytes, None))
                _check(L.smilehip_copy_to_host(ctx, taps["shs"].ctypes.data, d_shs, taps["shs"].nbytes, None))
                _check(L.smilehip_copy_to_host(ctx, taps["e60"].ctypes.data, d_e60, taps["e60"].nbytes, None))
            _check(L.smilehip_batch_f0_taps(self._h, None, None, None))
        finally:
            L.smilehip_free(ctx, d_pcm)
            L.smilehip_free(ctx, d_out)
            L.smilehip_free(ctx, d_hps)
        return out, taps

    def func_rows(self):
        """LLD rows each utterance's functionals summarise (IS09 chain plans)."""
        r = np.zeros(self.n_utt, np.int64)
        _check(load().smilehip_batch_func_rows(self._h, r.ctypes.data))
        return r

    def functionals_device(self, d_lld_ptr, ld_lld, d_func_ptr, ld_func, mask=None, stream=None):
        L = load()
        mask = L.smilehip_functionals_is09_mask() if mask is None else mask
        _check(L.smilehip_batch_functionals(self.plan._h, self._h, d_lld_ptr, ld_lld, mask, d_func_ptr, ld_func, stream))

    def functionals_host(self, lld, mask=None):
        """lld: the matrix run_host returned -> n_utt x (n_out * count(mask))."""
        L = load()
        mask = L.smilehip_functionals_is09_mask() if mask is None else mask
        per = L.smilehip_functionals_count(mask)
        n_out = self.plan.geometry.n_out
        lld = np.ascontiguousarray(lld, dtype=np.float32)
        assert lld.shape == (self.total_rows, n_out)
        out = np.zeros((self.n_utt, n_out * per), np.float32)
        ctx = self.plan.ctx._h
        d_lld, d_out = _vp(), _vp()
        _check(L.smilehip_alloc(ctx, max(lld.nbytes, 4), C.byref(d_lld)))
        _check(L.smilehip_alloc(ctx, max(out.nbytes, 4), C.byref(d_out)))
        try:
            if lld.nbytes:
                _check(L.smilehip_copy_to_device(ctx, d_lld, lld.ctypes.data, lld.nbytes, None))
            _check(L.smilehip_batch_functionals(self.plan._h, self._h, d_lld, n_out, mask, d_out, n_out * per, None))
            _check(L.smilehip_stream_synchronize(ctx, None))
            if out.nbytes:
                _check(L.smilehip_copy_to_host(ctx, out.ctypes.data, d_out, out.nbytes, None))
        finally:
            L.smilehip_free(ctx, d_lld)
            L.smilehip_free(ctx, d_out)
        return out

    def run_host_with_functionals16(self, pcm):
        """ComParE whole-level plans: smilehip_lld_run + smilehip_batch_functionals_compare16 on the same device matrix.
        Returns (lld [total_rows x 130], func [n_utt x 6373], b_extra [n_utt x 110], pending [n_utt])."""
        L = load()
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        n_out = self.plan.geometry.n_out
        nf = L.smilehip_functionals_compare16_count()
        lld = np.zeros((self.total_rows, n_out), np.float32)
        func = np.zeros((self.n_utt, nf), np.float32)
        ex = np.zeros((self.n_utt, 110), np.float32)
        ctx = self.plan.ctx._h
        d_pcm, d_lld, d_func = _vp(), _vp(), _vp()
        _check(L.smilehip_alloc(ctx, max(pcm.nbytes, 4), C.byref(d_pcm)))
        _check(L.smilehip_alloc(ctx, max(lld.nbytes, 4), C.byref(d_lld)))
        _check(L.smilehip_alloc(ctx, max(func.nbytes, 4), C.byref(d_func)))
        try:
            if pcm.nbytes:
                _check(L.smilehip_copy_to_device(ctx, d_pcm, pcm.ctypes.data, pcm.nbytes, None))
            _check(L.smilehip_lld_run(self.plan._h, self._h, d_pcm, d_lld, n_out, None))
            fn = L.smilehip_batch_functionals_is13_compare if self.plan.cfg.jitter_broken_thresh else L.smilehip_batch_functionals_compare16
            _check(fn(self.plan._h, self._h, d_lld, n_out, d_func, nf, None))
            _check(L.smilehip_stream_synchronize(ctx, None))
            if lld.nbytes:
                _check(L.smilehip_copy_to_host(ctx, lld.ctypes.data, d_lld, lld.nbytes, None))
            if func.nbytes:
                _check(L.smilehip_copy_to_host(ctx, func.ctypes.data, d_func, func.nbytes, None))
            d_ex = _vp()
            _check(L.smilehip_batch_compare_b_extra(self._h, C.byref(d_ex)))
            if ex.nbytes:
                _check(L.smilehip_copy_to_host(ctx, ex.ctypes.data, d_ex, ex.nbytes, None))
        finally:
            L.smilehip_free(ctx, d_pcm)
            L.smilehip_free(ctx, d_lld)
            L.smilehip_free(ctx, d_func)
        return lld, func, ex

    def run_host_egemaps(self, pcm, functionals=True, taps=False):
        """eGeMAPSv02 plans: smilehip_lld_run [+ smilehip_batch_functionals_egemaps] on host data. Returns
        (lld [total_rows x 25], func [n_utt x 88] or None[, taps: dict of the per-frame scratch levels])."""
        L = load()
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        n_out = self.plan.geometry.n_out
        lld = np.zeros((self.total_rows, n_out), np.float32)
        func = np.zeros((self.n_utt, 88), np.float32) if functionals else None
        ctx = self.plan.ctx._h
        d_pcm, d_lld, d_func = _vp(), _vp(), _vp()
        _check(L.smilehip_alloc(ctx, max(pcm.nbytes, 4), C.byref(d_pcm)))
        _check(L.smilehip_alloc(ctx, max(lld.nbytes, 4), C.byref(d_lld)))
        _check(L.smilehip_alloc(ctx, max(self.n_utt * 88 * 4, 4), C.byref(d_func)))
        out_taps = {}
        try:
            if pcm.nbytes:
                _check(L.smilehip_copy_to_device(ctx, d_pcm, pcm.ctypes.data, pcm.nbytes, None))
            _check(L.smilehip_lld_run(self.plan._h, self._h, d_pcm, d_lld, n_out, None))
            if functionals:
                _check(L.smilehip_batch_functionals_egemaps(self.plan._h, self._h, d_func, 88, None))
            _check(L.smilehip_stream_synchronize(ctx, None))
            if lld.nbytes:
                _check(L.smilehip_copy_to_host(ctx, lld.ctypes.data, d_lld, lld.nbytes, None))
            if functionals and func.nbytes:
                _check(L.smilehip_copy_to_host(ctx, func.ctypes.data, d_func, func.nbytes, None))
            if taps:
                ptrs = [_vp() for _ in range(9)]
                fo60 = np.zeros(self.n_utt + 1, np.int64)
                _check(L.smilehip_batch_egemaps_taps(self._h, *[C.byref(q) for q in ptrs], fo60.ctypes.data))
                nf, nf60 = self.total_frames, int(fo60[-1])
                T20 = np.diff(self.frame_offsets_frames())
                T60 = np.diff(fo60)
                nfin = int(np.where(T60 >= 1, T20 + 1, 0).sum())
                shapes = [("raw20", (nf, 12)), ("lpc", (nf, 12)), ("formants", (nf, 10)), ("pitch3", (nf60, 3)), ("jit4", (nf60, 4)),
                          ("shim_db", (nf60, 1)), ("harm6", (nf60, 6)), ("func_in", (nfin, 36))]
                for (name, shp), q in zip(shapes, ptrs):
                    a = np.zeros(shp, np.float32)
                    if a.nbytes:
                        _check(L.smilehip_copy_to_host(ctx, a.ctypes.data, q, a.nbytes, None))
                    out_taps[name] = a
                pend = np.zeros(self.n_utt, np.int32)
                if pend.nbytes:
                    _check(L.smilehip_copy_to_host(ctx, pend.ctypes.data, ptrs[8], pend.nbytes, None))
                out_taps["pending"] = pend
                out_taps["frame_off60"] = fo60
                out_taps["fin_off"] = np.concatenate([[0], np.cumsum(np.where(T60 >= 1, T20 + 1, 0))]).astype(np.int64)
        finally:
            L.smilehip_free(ctx, d_pcm)
            L.smilehip_free(ctx, d_lld)
            L.smilehip_free(ctx, d_func)
        return (lld, func, out_taps) if taps else (lld, func)

    def frame_offsets_frames(self):
        """Frame (not row) offsets per utterance: frames of the plan's own framer."""
        T = [self.plan.num_frames(int(self.sample_offsets[u + 1] - self.sample_offsets[u])) for u in range(self.n_utt)]
        return np.concatenate([[0], np.cumsum(T)]).astype(np.int64)

    def funcspec_host(self, lld, spec, col_first, n_cols, rows_cut, extra=None):
        """lld: the matrix run_host returned -> n_utt x (n_cols * count(spec)) through smilehip_batch_funcspec;
        extra: optional n_utt x n_cols matrix, one more row per utterance."""
        L = load()
        per = funcspec_count(spec)
        lld = np.ascontiguousarray(lld, dtype=np.float32)
        ld = lld.shape[1]
        assert lld.shape[0] == self.total_rows
        out = np.zeros((self.n_utt, n_cols * per), np.float32)
        ctx = self.plan.ctx._h
        d_lld, d_out, d_ex = _vp(), _vp(), _vp()
        _check(L.smilehip_alloc(ctx, max(lld.nbytes, 4), C.byref(d_lld)))
        _check(L.smilehip_alloc(ctx, max(out.nbytes, 4), C.byref(d_out)))
        try:
            if extra is not None:
                extra = np.ascontiguousarray(extra, dtype=np.float32)
                assert extra.shape == (self.n_utt, n_cols)
                _check(L.smilehip_alloc(ctx, max(extra.nbytes, 4), C.byref(d_ex)))
                _check(L.smilehip_copy_to_device(ctx, d_ex, extra.ctypes.data, extra.nbytes, None))
            if lld.nbytes:
                _check(L.smilehip_copy_to_device(ctx, d_lld, lld.ctypes.data, lld.nbytes, None))
            _check(L.smilehip_batch_funcspec(self.plan._h, self._h, C.byref(spec), d_lld, ld, col_first, n_cols, rows_cut,
                                             d_ex if extra is not None else None, n_cols, d_out, n_cols * per, None))
            _check(L.smilehip_stream_synchronize(ctx, None))
            if out.nbytes:
                _check(L.smilehip_copy_to_host(ctx, out.ctypes.data, d_out, out.nbytes, None))
        finally:
            L.smilehip_free(ctx, d_lld)
            L.smilehip_free(ctx, d_out)
            if extra is not None:
                L.smilehip_free(ctx, d_ex)
        return out

    def close(self):
        if self._h:
            load().smilehip_batch_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def extract_mfcc(pcm_list, cfg=None, device=0):
    """One-call helper: list of int16 arrays -> list of (T x n_out) matrices."""
    ctx = Context(device)
    plan = Plan(ctx, cfg)
    lens = [len(p) for p in pcm_list]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([np.asarray(p, np.int16) for p in pcm_list]) if pcm_list else np.zeros(0, np.int16)
    b = Batch(plan, off)
    out = b.run_host(pcm)
    res = [out[b.frame_offsets[i]:b.frame_offsets[i + 1]] for i in range(len(pcm_list))]
    b.close()
    plan.close()
    ctx.close()
    return res


# ------------------------------------------------ per-component entry points
# Device pointers are plain ints (e.g. torch.Tensor.data_ptr()); ld = leading
# dimension in floats. Asynchronous on `stream`.
def pcm_convert_host(ctx, raw, n_bps, n_bits, n_chan, mixdown=True):
    """raw: bytes / uint8 array of interleaved PCM -> float32 array (n,) or (n, n_chan)."""
    L = load()
    raw = np.ascontiguousarray(np.frombuffer(bytes(raw), dtype=np.uint8))
    n = len(raw) // (n_bps * n_chan)
    out = np.zeros(n if mixdown else (n, n_chan), np.float32)
    d_in, d_out = _vp(), _vp()
    _check(L.smilehip_alloc(ctx._h, max(raw.nbytes, 4), C.byref(d_in)))
    _check(L.smilehip_alloc(ctx._h, max(out.nbytes, 4), C.byref(d_out)))
    try:
        if raw.nbytes:
            _check(L.smilehip_copy_to_device(ctx._h, d_in, raw.ctypes.data, raw.nbytes, None))
        _check(L.smilehip_pcm_convert(ctx._h, d_in, n_bps, n_bits, n_chan, int(mixdown), n, d_out, None))
        _check(L.smilehip_stream_synchronize(ctx._h, None))
        if out.nbytes:
            _check(L.smilehip_copy_to_host(ctx._h, out.ctypes.data, d_out, out.nbytes, None))
    finally:
        L.smilehip_free(ctx._h, d_in)
        L.smilehip_free(ctx._h, d_out)
    return out


def kernel_timing(enable=True):
    """HIP events around every batch-chain launch from now on (records cleared); process-wide (smilehip_kernel_timing)"""
    _check(load().smilehip_kernel_timing(1 if enable else 0))


def kernel_timing_report():
    """{kernel name (template arguments stripped): (launches, summed ms)} of the launches since kernel_timing(True); call after
    synchronising (smilehip_kernel_timing_report)"""
    L = load()
    n = 1 << 16
    while True:
        buf = C.create_string_buffer(n)
        got = L.smilehip_kernel_timing_report(buf, n)
        if got >= 0:
            break
        n = -got + 1
    out = {}
    for line in buf.value.decode().splitlines():
        name, launches, ms = line.split("\t")
        name = name.split("<")[0].strip()
        a = out.get(name, (0, 0.0))
        out[name] = (a[0] + int(launches), a[1] + float(ms))
    return out


def pcm16_to_float(ctx, d_pcm, n, d_out, stream=None):
    _check(load().smilehip_pcm16_to_float(ctx._h, d_pcm, n, d_out, stream))


def preemphasis_frames(ctx, d_src, ld_src, d_dst, ld_dst, n_frames, N, k, de=0, stream=None):
    _check(load().smilehip_preemphasis_frames(ctx._h, d_src, ld_src, d_dst, ld_dst, n_frames, N, k, de, stream))


def window_frames(plan, d_src, ld_src, d_dst, ld_dst, n_frames, stream=None):
    _check(load().smilehip_window_frames(plan._h, d_src, ld_src, d_dst, ld_dst, n_frames, stream))


def rfft_frames(plan, d_src, ld_src, d_dst, ld_dst, n_frames, stream=None):
    _check(load().smilehip_rfft_frames(plan._h, d_src, ld_src, d_dst, ld_dst, n_frames, stream))


def fftmag_frames(plan, d_src, ld_src, d_dst, ld_dst, n_frames, stream=None):
    _check(load().smilehip_fftmag_frames(plan._h, d_src, ld_src, d_dst, ld_dst, n_frames, stream))


def melspec_frames(plan, d_src, ld_src, d_dst, ld_dst, n_frames, stream=None):
    _check(load().smilehip_melspec_frames(plan._h, d_src, ld_src, d_dst, ld_dst, n_frames, stream))


def mfcc_frames(plan, d_src, ld_src, d_dst, ld_dst, n_frames, stream=None):
    _check(load().smilehip_mfcc_frames(plan._h, d_src, ld_src, d_dst, ld_dst, n_frames, stream))


def delta_chain(plan, batch, d_io, ld, D, W, n_orders, stream=None):
    _check(load().smilehip_delta_chain(plan._h, batch._h, d_io, ld, D, W, n_orders, stream))


class ViterbiStream:
    """cPitchSmootherViterbi as a stream (smilehip_viterbi_stream_*): push one frame of six (F0, voicing) candidates, get the
    frames that became decided -- (frame index, state) pairs; flush() at end of input."""

    def __init__(self, ctx, buffer_len=30, voicing_cutoff=0.7, weights=(2.0, 10.0, 10.0, 10.0, 4.0, 1.0)):
        self._h = _vp()
        w = (C.c_double * 6)(*weights)
        _check(load().smilehip_viterbi_stream_create(ctx._h, buffer_len, voicing_cutoff, w, C.byref(self._h)))

    def _out(self, fn, *head):
        n = C.c_int32(0)
        fr = (C.c_int32 * 128)()
        st = (C.c_int32 * 128)()
        _check(fn(self._h, *head, C.byref(n), fr, st, 128))
        return [(fr[i], st[i]) for i in range(n.value)]

    def push(self, cand_f0, cand_voicing):
        f = np.ascontiguousarray(cand_f0, dtype=np.float32)
        v = np.ascontiguousarray(cand_voicing, dtype=np.float32)
        assert f.size == 6 and v.size == 6
        return self._out(load().smilehip_viterbi_stream_push, f.ctypes.data, v.ctypes.data)

    def flush(self):
        return self._out(load().smilehip_viterbi_stream_flush)

    def close(self):
        if self._h:
            load().smilehip_viterbi_stream_destroy(self._h)
            self._h = None
