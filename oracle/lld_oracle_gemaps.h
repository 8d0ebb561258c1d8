/*
 * lld_oracle_gemaps.h -- CPU ORACLE, eGeMAPSv02 part. TEST INFRASTRUCTURE ONLY (see lld_oracle.h).
 */
#ifndef LLD_ORACLE_GEMAPS_H
#define LLD_ORACLE_GEMAPS_H
#include <stdint.h>
#include "lld_oracle_funcspec.h"

#ifdef __cplusplus
extern "C" {
#endif

/* cSpectral with the GeMAPS option sets (spectral.cpp:586-1254) */
typedef struct { long K, lo, hi; double *frq; float *prev; int have_prev; float spec_floor, log_spec_floor; } lldo_gspec;
void lldo_gspec_init(lldo_gspec *s, long K, double frame_size_sec);
void lldo_gspec_reset(lldo_gspec *s);
void lldo_gspec_free(lldo_gspec *s);
void lldo_gspec_frame(lldo_gspec *s, const float *mag, float *dst5);
float lldo_energy2(const float *x, long N);

/* cSpecResample -> cLpc -> cFormantLpc */
typedef struct { long K, I, kMax; double target_fs; float *costable, *sintable; } lldo_specresample;
int  lldo_specresample_init(lldo_specresample *r, long n_in, double fs_sec, double last_fs_sec, double base_period,
                            double target_fs);
void lldo_specresample_free(lldo_specresample *r);
void lldo_specresample_frame(const lldo_specresample *r, const float *in, float *out);
void lldo_lpc_acf(const float *x, long n, int p, float *lpc);
void lldo_formant_lpc(const float *lpc, int n_lpc, int nf, double T, double min_f, double max_f, double *roots, float *dst);

/* cHarmonics with [gemapsv01b_harmonics]'s options (harmonics.cpp:743-1031) */
void lldo_harmonics_frame(float F0, const float *formants, int n_formants, const float *mag, long K, double fs_sec, float *dst6);
/* cPitchSmootherViterbi with a given bufferLength; cPitchJitter with a given searchRangeRel and the shimmerLocalDB output
 * (lld_oracle_f0.c) */
void lldo_pitch_viterbi_ex(const float *shs, long T, float voicing_cutoff, int buflen, float *out2, int *states, long *pending);
void lldo_pitch_jitter_ex(const float *wave, long n, const float *f0, long T, long N, long H, double sample_rate,
                          double frame_step_sec, double search_range_rel, float *out4, float *shimmer_db);

/* every per-frame level of eGeMAPSv02's LLD graph for one utterance; arrays are frame-major */
typedef struct {
  long T20, T60, P;          /* frames of the 20 ms / 60 ms framers; frames the Viterbi smoother had not decided at the end */
  float *loudness;           /* T20      gemapsv01b_loudness */
  float *lspec;              /* T20 x 4  gemapsv01b_logSpectral [slope0-500, slope500-1500, alphaRatio, hammarberg] */
  float *flux;               /* T20      egemapsv02_logSpectral_flux */
  float *mfcc;               /* T20 x 4  egemapsv02_mfcc */
  float *energy2;            /* T20      egemapsv02_energyRMS */
  float *formants;           /* T20 x 10 gemapsv01b_formants [5 frequencies | 5 bandwidths] */
  float *pitch;              /* T60 x 3  gemapsv01b_logPitch [F0final, F0finalLog, voicingFinalUnclipped] */
  float *jitter;             /* T60 x 2  gemapsv01b_jitterShimmer [jitterLocal, shimmerLocalDB] */
  float *harm;               /* T60 x 6  gemapsv01b_harmonics */
  float *shs, *e60;          /* T60 x 21 gemapsv01b_pitchShsG60, T60 gemapsv01b_e60 */
} lldo_egemaps_lv;
long lldo_egemaps_levels(const int16_t *pcm, long n_samples, lldo_egemaps_lv *L);
void lldo_egemaps_levels_free(lldo_egemaps_lv *L);

/* the smoothed levels (cContourSmoother outputs) the LLD sinks and the functionals read */
typedef struct {
  long T20, T60, P;
  float *E;                  /* (T20+1) x 10 egemapsv02_lldsetE_smo */
  float *F;                  /* (T60+1) x 15 egemapsv02_lldsetF_smo */
  float *logf0;              /* (T60+1)      gemapsv01b_lld_single_logF0_smo */
  float *loud;               /* (T20+1)      gemapsv01b_loudness_smo */
  float *NoZ;                /* (T20+1) x 5  egemapsv02_lldSetNoF0AndLoudnessZ_smo */
  float *NoNz;               /* (T60+1) x 14 egemapsv02_lldSetNoF0AndLoudnessNz_smo */
  float *specV;              /* (T60+1) x 9  egemapsv02_lldSetSpectralNz_smo */
  float *specU;              /* (T60+1) x 5  egemapsv02_lldSetSpectralZ_smo */
} lldo_egemaps_smo;
void lldo_egemaps_smooth(const lldo_egemaps_lv *L, lldo_egemaps_smo *S);
void lldo_egemaps_smo_free(lldo_egemaps_smo *S);
long lldo_egemaps_lld_chain(const int16_t *pcm, long n_samples, float *out25);
int  lldo_funcspec_egemaps(const char *inst, lldo_func_spec *s);
int  lldo_egemaps_func_from_levels(const lldo_egemaps_lv *L, const lldo_egemaps_smo *S, float *out88);
int  lldo_egemaps_func(const int16_t *pcm, long n_samples, float *out88);

#ifdef __cplusplus
}
#endif
#endif
