/* TEST INFRASTRUCTURE ONLY -- option block of the general functionals restatement
 * (lld_oracle_funcspec.c). One cFunctionals instance = an ordered list of families
 * (functionalsEnabled) plus each family's options; bit i of a family mask enables the family's
 * value i in the reference's own enab[] index order, which is also its output order. */
#ifndef LLD_ORACLE_FUNCSPEC_H
#define LLD_ORACLE_FUNCSPEC_H
#include <stdint.h>

enum {
  LLDO_FAM_EXTREMES = 0, LLDO_FAM_MEANS, LLDO_FAM_MOMENTS, LLDO_FAM_REGRESSION, LLDO_FAM_PERCENTILES,
  LLDO_FAM_TIMES, LLDO_FAM_SEGMENTS, LLDO_FAM_LPC, LLDO_FAM_PEAKS2, LLDO_FAM_ONSET, LLDO_FAM_PEAKS, LLDO_FAM_CROSSINGS, LLDO_FAM_DCT, LLDO_FAM_SAMPLES, LLDO_FAM_MODULATION, LLDO_FAM_COUNT
};
enum { LLDO_NORM_SEGMENT = 0, LLDO_NORM_SECOND = 1, LLDO_NORM_FRAME = 2 };   /* functionalComponent.hpp:27-33 */
/* functionalSegments.cpp:118-155; ltX / gtX / geqX / leqX are parsed there but fall to the switch's default (:872-874): delta */
enum { LLDO_SEG_RELTH = 0, LLDO_SEG_NONX = 1, LLDO_SEG_EQX = 2, LLDO_SEG_MRELTH = 3, LLDO_SEG_ABSTH = 4, LLDO_SEG_NARELTH = 5,
       LLDO_SEG_NAMRELTH = 6, LLDO_SEG_NAABSTH = 7, LLDO_SEG_DELTA = 8, LLDO_SEG_DELTA2 = 9, LLDO_SEG_CHX = 10 };

typedef struct lldo_func_spec {
  int32_t n_fam;
  int32_t fam[12];              /* functionalsEnabled, in order */
  int32_t non_zero_functs;      /* cFunctionals.nonZeroFuncts: 0, 1 (x != 0), 2 (x > 0) */
  int32_t reserved0;
  double period;                /* period of the input level in seconds (getInputPeriod) */
  /* Extremes: max min range maxpos minpos amean maxameandist minameandist */
  uint32_t ext_mask; int32_t ext_norm;
  /* Means: amean absmean qmean nzamean nzabsmean nzqmean nzgmean nnz flatness posamean negamean posqmean
   * posrqmean negqmean negrqmean rqmean nzrqmean */
  uint32_t means_mask; int32_t means_norm;
  /* Moments: variance stddev skewness kurtosis amean stddevNorm; mom_stddev_norm = the option's value (1 abs, 2) */
  uint32_t mom_mask; int32_t mom_stddev_norm; int32_t mom_ratio_limit; int32_t reserved1;
  /* Regression: linregc1 linregc2 linregerrA linregerrQ qregc1 qregc2 qregc3 qregerrA qregerrQ centroid qregls
   * qregrs qregx0 qregy0 qregyr qregy0nn qregc3nn qregyrnn */
  uint32_t reg_mask; int32_t reg_centroid_norm, reg_norm_coeff, reg_norm_inputs, reg_centroid_abs,
      reg_centroid_limit, reg_ratio_limit, reg_old_buggy_qerr;
  /* Percentiles: quartile1..3 iqr12 iqr23 iqr13, then percentile[], pctlrange[] */
  uint32_t pct_mask; int32_t pct_interp, n_pctl, n_range;
  double pctl[8]; int32_t range_a[8], range_b[8];
  /* Times: upleveltime25 downleveltime25 ..50 ..75 ..90 risetime falltime leftctime rightctime duration */
  uint32_t times_mask; int32_t times_norm, times_buggy_sec_norm, reserved2;
  /* Segments: numSegments meanSegLen maxSegLen minSegLen segLenStddev */
  uint32_t seg_mask; int32_t seg_norm, seg_algo, seg_max_num, seg_min_lng, seg_auto_min_lng, seg_pause_min_lng,
      seg_x_is_rel, seg_n_thresholds, seg_ravg_lng;        /* ravgLng (delta, delt2): <= 0 = Nin / (maxNumSeg / 2) */
  float seg_x; float seg_thresholds[8]; float seg_range_rel_threshold;
  /* Lpc: lpGain, lpc[first..order) */
  int32_t lpc_gain, lpc_coeffs, lpc_first, lpc_order;
  /* Peaks2: the 32 values of functionalPeaks2.cpp:60-67 */
  uint32_t pk_mask; int32_t pk_norm, pk_ratio_limit, pk_dyn_rel, pk_use_abs, reserved5;
  float pk_rel_thresh, pk_abs_thresh;
  /* Onset (functionalOnset.cpp:21-29): onsetPos offsetPos numOnsets numOffsets onsetRate */
  uint32_t ons_mask; int32_t ons_norm, ons_use_abs, reserved6;
  float ons_thr_on, ons_thr_off;
  /* Peaks (the older peak picker, functionalPeaks.cpp:20-30; overlapFlag = 1): numPeaks meanPeakDist peakMean peakMeanMeanDist
   * peakDistStddev */
  uint32_t pko_mask; int32_t pko_norm;
  /* Crossings (functionalCrossings.cpp:21-28): zcr mcr amean; DCT (functionalDCT.cpp): coefficients first .. last; Samples
   * (functionalSamples.cpp): the contour's values at n_samples relative positions in [0, 1] */
  uint32_t crs_mask; int32_t dct_first, dct_last, n_samples;
  double sample_pos[8];
  /* Percentiles.pctlquotient[] (functionalPercentiles.cpp:179-232, :402-411): percentile a over percentile b through the soft
   * limiter (50, 100). The reference forms them only when pctlrange[] is not empty -- zeros otherwise -- and only if n_pctl > 0. */
  int32_t n_quot, quot_a[8], quot_b[8];
  /* Times.upleveltime[] / downleveltime[] (functionalTimes.cpp:129-165, :347-364): share of the contour above / not above
   * level * range + min */
  int32_t n_ul, n_dl, reserved7;
  double ul[8], dl[8];
  /* Modulation (lld_oracle_modspec.c): window / step in values, number of bins, LLDO_WIN_* window, removeNonZeroMean, axis */
  int32_t mod_win_frames, mod_step_frames, mod_n_bins, mod_win_func, mod_remove_nz_mean, reserved8;
  double mod_min_freq, mod_max_freq;
} lldo_func_spec;

/* values per input column; < 0 for an unusable spec */
int lldo_funcspec_count(const lldo_func_spec *s);
/* x: rows x cols (leading dimension ld) -> out: cols x count, element-major (column c's values at [c*count, ..)).
 * Returns count; 0 if rows <= 0 (the reference then emits no vector at all). */
int lldo_funcspec_apply(const lldo_func_spec *s, const float *x, int64_t ld, int64_t rows, int cols, float *out);

#endif
