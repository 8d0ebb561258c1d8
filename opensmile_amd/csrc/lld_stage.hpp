// Launchers of the per-component kernels (lld_stage_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lld_params.hpp"

namespace smilehip {
hipError_t stage_pcm16(const int16_t *pcm, int64_t n, float *out, hipStream_t s);
hipError_t stage_pcm_convert_float(const float *buf, int n_chan, int mixdown, int64_t n, float *out, hipStream_t s);
hipError_t stage_pcm_convert(const void *buf, int n_bps, int n_bits, int n_chan, int mixdown, int64_t n, float *out,
                             hipStream_t s);
hipError_t stage_preemph(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int64_t N, float k,
                         int de, hipStream_t s);
hipError_t stage_window(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int64_t N,
                        const float *w, float off, hipStream_t s);
hipError_t stage_rfft_oo(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int N, int Nfft,
                         int pad_left, const OouraTab &T, hipStream_t s);
hipError_t stage_rfft(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int N, int Nfft,
                      int pad_left, const float2 *twh, const float2 *twf, hipStream_t s);
hipError_t stage_fftmag(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int Nfft, hipStream_t s);
hipError_t stage_melspec(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int K, int n_bands,
                         int use_power, const float *coef, const int32_t *rng, float scale, hipStream_t s);
hipError_t stage_mfcc(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int n_bands, int n_mfcc,
                      const float *rows, const float *gain, float melfloor, float log_floor, hipStream_t s);
// second set (lld_stage2_kernels.hip): R9, R10, R12, R13 one component at a time
hipError_t stage_sumsq(const float *src, int64_t lds, int64_t N, int64_t nF, double *out, hipStream_t s);
hipError_t stage_zcr_count(const float *src, int64_t lds, int64_t N, int64_t nF, int32_t *out, hipStream_t s);
// lld_stage4_kernels.hip: the components the other INTERSPEECH sets add (cIntensity, cLsp, cPitchSmoother, cVectorOperation) and
// cSpecResample / cLpc for any geometry
hipError_t stage_intensity(const float *src, int64_t lds, int n_sum, double w0, double w1, double win_sum, int flags, float *dst,
                           int64_t ldd, int64_t nF, hipStream_t s);
hipError_t stage_lsp(const float *lpc, int64_t lds, int p, float *dst, int64_t ldd, int64_t nF, hipStream_t s);
hipError_t stage_vecop(int op, float aux, float logfloor, const float *src, int64_t lds, int n_cols, float *dst, int64_t ldd, int64_t nF,
                       hipStream_t s);
hipError_t stage_pitch_smoother(int n_cand, float voicing_cutoff, int octave_correction, int post_simple, int flags, const float *src,
                                int64_t lds, const int64_t *row_off, int n_streams, int64_t n_single, void *state, int resume,
                                float *dst, int64_t ldd, int64_t *written, hipStream_t s);
hipError_t stage_specresample_g(const float *src, int64_t lds, int K, int I, int kMax, const float *cost, const float *sint, float *dst,
                                int64_t ldd, int64_t nF, hipStream_t s);
hipError_t stage_lpc_g(const float *x, int64_t lds, int n, int p, float *dst, int64_t ldd, int64_t nF, hipStream_t s);
// lld_stage3_kernels.hip: the components' other option sets
hipError_t stage_htk_rows_be(const float *src, int64_t n, uint32_t *dst, hipStream_t s);
struct OouraTab;
hipError_t stage_irfft_oo(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int Nfft, const OouraTab &T, hipStream_t s);
hipError_t stage_fftmagphase(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int Nfft, int flags, float dBpnorm,
                             float mindBp, hipStream_t s);
hipError_t stage_melspec_table(const float *src, int64_t lds, int K, int nB, int dense, const float *coef, const int32_t *chanmap, int nLoF,
                               int nHiF, int use_power, float htk_scale, float *dst, int64_t ldd, int64_t nF, hipStream_t s);
hipError_t stage_melspec_inverse_table(const float *src, int64_t lds, int n_src, int K, const float *coef, const int32_t *chanmap, int nLoF,
                                       int nHiF, int use_power, float htk_div, float *dst, int64_t ldd, int64_t nF, hipStream_t s);
hipError_t stage_pitchacf_zcr(const float *src, int64_t lds, int64_t nF, int n, int skip, double *zcr, hipStream_t s);
hipError_t stage_mzcr(const float *src, int64_t lds, int N, int64_t nF, int flags, float *dst, int64_t ldd, hipStream_t s);
hipError_t stage_valbased(const float *src, int64_t lds, int N, int64_t nF, int idx, float threshold, int invert, int allow_equal,
                          int zerovec, int remove_idx, float output_val, float *dst, int64_t ldd, int32_t *keep, hipStream_t s);
hipError_t stage_acf(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int K, int n_out, int use_power,
                     int cepstrum, int norm_output, int abs_cepstrum, const float2 *tw_half, const float2 *tw_full,
                     const OouraTab &OO, hipStream_t s);
hipError_t stage_pitchacf(const float *src, int64_t lds, int64_t nF, int n, double fs_sec, double max_pitch, double *voicing,
                          int32_t *max_idx, hipStream_t s);
struct PlpConsts { float melfloor, compression, iir, fir[5]; };
hipError_t stage_spectral(const float *src, int64_t lds, float *state, bool first, float *dst, int64_t ldd, int64_t nF, int K,
                          const SpectralConsts &C, hipStream_t s);
// cSpectral, general option set (lld_spectral_general.hip): what the kernel needs of smilehip_spectral_opts, ready to use
struct SpectralGeneral {
  int32_t K;
  double frame_size_sec;
  int32_t n_bands, band_iL[16], band_iR[16];            // edge bins and their weights (spectral.cpp:779-826)
  double band_wL[16], band_wR[16];
  int32_t n_rolloff;
  double rolloff[16];
  int32_t flux, centroid, max_pos, min_pos, entropy, variance, skewness, kurtosis, slope, sharpness, harmonicity, flatness, log_flatness;
  double slope_Sf, slope_S2f;                           // sums of frq and frq^2 over bins 1 .. K-1 (:1405-1412)
  const double *sharp_w;                                // [K - 1] sharpness weights of bins 1 .. K-1 (:1440-1455)
  // round 6: the rest of the linear-spectrum branch
  int32_t spec_diff, spec_pos_diff, flux_centroid, flux_at_flux_centroid, standard_deviation, n_out;
  int32_t n_slopes, sl_iL[16], sl_iR[16];               // slopes[]: edge bins, their weights, idxR - idxL (:872-943)
  double sl_wL[16], sl_wR[16], sl_Nind[16];
};
hipError_t stage_mfcc_inverse(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int n_bands, int first, int last, int htk,
                              int do_log, const float *rows, const float *lifter, hipStream_t s);
hipError_t stage_spectral_general(const SpectralGeneral &G, const float *mag, int64_t ld_src, float *state, int first, float *dst,
                                  int64_t ld_dst, int64_t n_frames, hipStream_t s);
hipError_t stage_plp(const float *src, int64_t lds, int n_bands, const float *eql, const PlpConsts &Q, int rasta, float *state,
                     float *dst, int64_t ldd, int64_t nF, hipStream_t s);
hipError_t stage_plp_cc(const float *src, int64_t lds, int n_bands, const float *eql, float melfloor, float compression,
                        int order, const float *costab, const float *sintab, float *dst, int64_t ldd, int64_t nF, hipStream_t s,
                        int out_stage = 3);               // 3: cepstra, 2: LP coefficients, 1: autocorrelation (cPlp's partial modes)
hipError_t stage_window_op(const float *x, float *y, int64_t nT, int kind, int W, float norm, hipStream_t s);
hipError_t stage_window_op_block(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t nT, int nC, int op, int W, float norm,
                                 int flags, hipStream_t s);
hipError_t stage_delta_seg_block(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t n_ticks, int bs, int nC, int W, float *d_norm,
                                 int flags, hipStream_t s);
hipError_t stage_frame_rows(const float *samples, int64_t N, int64_t step, int64_t nF, float *dst, int64_t ldd, hipStream_t s);
hipError_t stage_delta_op(const float *x, float *y, int64_t nT, int W, float norm, int flags, float *d_norm_io, hipStream_t s);
hipError_t stage_window_op_seq(const float *x, float *y, int64_t nT, int kind, int W, float *d_norm, hipStream_t s);
}  // namespace smilehip
