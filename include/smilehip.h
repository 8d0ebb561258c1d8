/*
 * smilehip.h -- C ABI of libsmilehip.so: openSMILE's per-frame LLD hot path
 * (framing -> pre-emphasis -> window -> real FFT -> magnitude -> mel bank ->
 * log/DCT/lifter -> delta regression) as hand-written HIP kernels for
 * MI355X (gfx950).
 *
 * This is the drop-in boundary: plain C, pointers and sizes only, no C++ or
 * torch types. The openSMILE-side adapter (opensmile_amd/plugin/, a
 * plugins/NAME.so registering components under the built-in names, see
 * INTEGRATION.md) and the Python host mirror (opensmile_amd/capi.py) both bind
 * exactly these symbols.
 *
 * Every entry point names the reference interface it replaces
 * (paths relative to the audeering/opensmile v3.0.2 tree).
 *
 * Conventions
 *  - All functions return SMILEHIP_OK (0) or a negative smilehip_status; they
 *    never throw. smilehip_last_error() gives the message for the calling
 *    thread's last failure (the C++ component turns it into COMP_ERR,
 *    src/include/core/exceptions.hpp:137).
 *  - Pointers named d_* are DEVICE pointers (HIP), h_* are host pointers.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream). All
 *    kernels are enqueued on it; nothing synchronises unless stated.
 *  - Frame matrices are frame-major ("column-major by frame" in the reference's
 *    words, cMatrix data[n + t*N], src/include/core/dataMemoryLevel.hpp:152-217):
 *    element n of frame t is at base[t*ld + n].
 *  - FLOAT_DMEM = float (src/include/core/smileTypes.h:28).
 *  - Concurrency: a context, a plan and a batch each own device scratch (functionals scratch and auxiliary streams in the
 *    context; side stream, fork / join events and tables in the plan; per-batch scratch matrices) that calls reuse. Calls on
 *    the SAME context / plan / batch must therefore be issued from one host thread at a time and on one stream at a time
 *    (finish or synchronise before switching streams); different batches of one plan may run concurrently only through
 *    entry points that do not use the plan's side stream (the MFCC / PLP / IS09 chains, the per-component operators), and
 *    the functionals calls of one context serialise on its scratch. One context per host thread is the simple rule.
 */
#ifndef SMILEHIP_H
#define SMILEHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMILEHIP_VERSION 0x000100

typedef enum smilehip_status {
  SMILEHIP_OK = 0,
  SMILEHIP_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
  SMILEHIP_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture */
  SMILEHIP_ERR_HIP = -3,         /* a HIP runtime call failed */
  SMILEHIP_ERR_NOMEM = -4
} smilehip_status;

/* winFunc ids; names as cWindower's `winFunc` option accepts them
 * (src/dspcore/windower.cpp:26-58, winFuncToInt in smileUtil.c) */
typedef enum smilehip_winfunc {
  SMILEHIP_WIN_RECT = 0, SMILEHIP_WIN_HANN = 1, SMILEHIP_WIN_HAMM = 2,
  SMILEHIP_WIN_GAUSS = 3, SMILEHIP_WIN_SINE = 4, SMILEHIP_WIN_TRI = 5,
  SMILEHIP_WIN_BARTLETT = 6, SMILEHIP_WIN_LANCZOS = 7
} smilehip_winfunc;

/* Parameters of the MFCC chain. Each group carries exactly the options the
 * corresponding reference component reads in myFetchConfig(). */
typedef struct smilehip_lld_config {
  uint32_t struct_size;        /* = sizeof(smilehip_lld_config) */
  /* cWaveSource (src/iocore/waveSource.cpp:190) + cFramer
   * (src/core/winToVecProcessor.cpp:435-456): frameMode=fixed,
   * frameCenterSpecial=left, noPostEOIprocessing=1 */
  double sample_rate;
  double frame_size_sec;
  double frame_step_sec;
  /* cVectorPreemphasis (src/dspcore/vectorPreemphasis.cpp:55,89-107);
   * preemph=0: component not in the chain */
  int32_t preemph;
  float   preemph_k;
  int32_t preemph_de;
  /* cWindower (src/dspcore/windower.cpp:66-118) */
  int32_t win_func;
  double  win_sigma;
  double  win_gain;
  double  win_offset;
  /* cTransformFFT (src/dspcore/transformFft.cpp:55-64) */
  int32_t zero_pad_symmetric;
  /* cMelspec (src/lldcore/melspec.cpp:66-136), specScale=mel, bwMethod=lr */
  int32_t n_bands;
  float   lofreq;
  float   hifreq;
  int32_t use_power;
  int32_t mel_htk_compatible;
  /* cMfcc (src/lldcore/mfcc.cpp:64-96) */
  int32_t first_mfcc;
  int32_t last_mfcc;
  float   cep_lifter;
  int32_t mfcc_htk_compatible;
  float   melfloor;
  /* cDeltaRegression x n_delta (src/dspcore/deltaRegression.cpp:58-91):
   * 0 = static only, 1 = +delta, 2 = +delta+accel (cVectorConcat order) */
  int32_t n_delta;
  int32_t delta_win;
  /* --- single-component plans (the plugin's per-component overrides) ---
   * A component sitting in the middle of a graph knows its input size and the
   * level's frameSizeSec, not the wave source's sample rate: when > 0 these
   * replace the values derived from sample_rate/frame_size_sec. stage_mask
   * selects which tables the plan builds (0 = whole chain). */
  /* --- LLD set --- chain_kind selects what smilehip_lld_run computes:
   *   SMILEHIP_CHAIN_MFCC: [mfcc | delta | accel]           (config/mfcc/MFCC12_0_D_A.conf)
   *   SMILEHIP_CHAIN_IS09: IS09_emotion's LLD level as its LLD sinks see it
   *     (config/is09-13/IS09_emotion_core.lld.conf.inc): 16 columns
   *     [pcm_RMSenergy | mfcc firstMfcc..lastMfcc (12) | pcm_zcr | voiceProb | F0]
   *     smoothed by cContourSmoother(smaWin) followed by their 16 cDeltaRegression
   *     columns; T+smaWin/2 rows per utterance (the SMA's end-of-input frame is kept).
   *     Uses cEnergy rms (src/lldcore/energy.cpp:152-168) on the windowed frame, cMZcr zcr
   *     (src/lldcore/mzcr.cpp:109-126) on the raw frame, two cAcf instances
   *     (src/dspcore/acf.cpp:249-349) and cPitchACF (src/lldcore/pitchACF.cpp:137-247).
   *   SMILEHIP_CHAIN_PLP: config/plp/PLP_0_D_A.conf: the MFCC chain with cPlp in cMfcc's place
   *     (src/lldcore/plp.cpp:416-593 with doAud = doIDFT = doLP = doLpToCeps = 1, htkcompatible = 1):
   *     [plp c1..c_lpOrder, c0 | delta | accel]; cep_lifter is cPlp's cepLifter.
   *   SMILEHIP_CHAIN_COMPARE_AB: ComParE_2016's LLD groups A and B as its LLD sinks see them
   *     (config/compare16/ComParE_2016_core.lld.conf.inc; columns 6..64 and 71..129 of the
   *     130-column lld;lld_de file): 59 columns
   *     [audspec_lengthL1norm | audspecRasta_lengthL1norm | pcm_RMSenergy | pcm_zcr (60 ms frame) |
   *      audSpec_Rfilt[26] | cSpectral x 15 | mfcc 1..14] after cContourSmoother(3), then their 59
   *     cDeltaRegression columns; T60+1 rows per utterance (T60 = frames of the 60 ms framer).
   *     cPlp as auditory spectrum with and without newRASTA (src/lldcore/plp.cpp:416-593),
   *     cSpectral with [is13_spectral]'s options (src/lldcore/spectral.cpp:586-1560),
   *     cVectorOperation ll1 (src/other/vectorOperation.cpp:475-481). The F0 group (SHS pitch,
   *     Viterbi, jitter) is out of scope. Utterances with T60 < 4 yield no rows. */
  int32_t  chain_kind;
  int32_t  plp_lp_order;                /* SMILEHIP_CHAIN_PLP: cPlp lpOrder (outputs lpOrder+1 cepstra, c0 last) */
  float    plp_compression;             /* cPlp compression */
  double   pitch_max;                   /* cPitchACF maxPitch */
  double   voicing_cutoff;              /* cPitchACF voicingCutoff */
  int32_t  sma_win;                     /* cContourSmoother smaWin (odd) */
  int64_t  force_frame_size;            /* N  */
  double   force_fft_frame_size_sec;    /* frameSizeSec of the spectrum level (cMelspec::configureField, melspec.cpp:150-173) */
  uint32_t stage_mask;                  /* SMILEHIP_STAGE_* bits */
  /* SMILEHIP_CHAIN_COMPARE_F0 (pitch_max = cPitchShs maxPitch, voicing_cutoff = its voicingCutoff) */
  double   pitch_min;                   /* cPitchBase minPitch */
  int32_t  shs_n_harmonics;             /* cPitchShs nHarmonics (<= 16) */
  float    shs_compression;             /* cPitchShs compressionFactor */
  float    f0_min_energy;               /* cValbasedSelector threshold on the 60 ms frame's RMS energy */
  /* SMILEHIP_CHAIN_MFCC / _PLP: the other files of config/mfcc and config/plp (MFCC12_E_D_A, *_Z, PLP_E_D_A, ...) */
  int32_t  append_log_energy;           /* [energy:cEnergy] log = 1, htkcompatible = 1 on the raw frame (src/lldcore/energy.cpp:152-185),
                                           appended to the static block ([cat:cVectorConcat]); the deltas cover it */
  int32_t  cms;                         /* [cms:cFullinputMean] (src/dspcore/fullinputMean.cpp, multiLoopMode = 0, meanNorm = amean):
                                           the utterance's mean is subtracted from the static cepstra (not from the energy
                                           column; the deltas come from the un-normalised level) */
  int32_t  jitter_broken_thresh;        /* [is13_pitchJitter] useBrokenJitterThresh (src/lld/pitchJitter.cpp:801-809): the period
                                           is accepted if its peak correlation exceeds the frame's running minimum (float)
                                           instead of minCC = 0.5 -- IS13_ComParE.conf; 0 in ComParE_2016.conf */
  /* F0 chains: cPitchSmootherViterbi bufferLength (src/lld/pitchSmootherViterbi.cpp:241; 0 = 30 as in ComParE_2016, 40 in
   * GeMAPS; <= 128) and cPitchJitter searchRangeRel (src/lld/pitchJitter.cpp:632-637; 0 = 0.25 as in ComParE_2016, 0.1 in GeMAPS) */
  int32_t  vit_buffer_len;
  double   jitter_search_range;
  /* SMILEHIP_CHAIN_EGEMAPS: cFormantLpc maxF (src/lld/formantLpc.cpp:224-231; 0 = 5450 as in GeMAPSv01b / eGeMAPSv02, 5500 in the
   * v01a files) */
  double   formant_max_freq;
  /* F0 chains: cSpecScale minF (src/dsp/specScale.cpp:228-300; 0 = 25 as in ComParE_2016 / GeMAPS, 20 in IS10_paraling .. IS12) */
  double   specscale_min_f;
  /* F0 chains: cPitchShs nCandidates (0 = 6 as in ComParE_2016 / GeMAPS; 1 .. 6: the 21-value rows keep six slots per field, the
   * unused ones zero) and greedyPeakAlgo = 0 (src/lld/pitchShs.cpp:286-302: IS11_speaker_state, IS12_speaker_trait) */
  int32_t  shs_n_candidates;
  int32_t  shs_old_peak_algo;
  /* cSpectral bands[0], bands[1] in Hz (src/lldcore/spectral.cpp:779-853; all four 0 = 250-650 and 1000-4000 as in ComParE_2016;
   * IS11_speaker_state has 25-650) */
  int32_t  spectral_band_lo[2], spectral_band_hi[2];
  /* F0 chains / smilehip_specscale_frames: cSpecScale's three post-processing switches, as "off" bits (0 = all three on, what the
   * shipped F0 chains set): 1 = specEnhance 0, 2 = specSmooth 0, 4 = auditoryWeighting 0 (src/dsp/specScale.cpp:326-353; without the
   * weighting the spline's values pass as they are, negative ones too) -- emobase2010, IS10_paraling_compat */
  int32_t  specscale_off;
} smilehip_lld_config;

#define SMILEHIP_CHAIN_MFCC 0
#define SMILEHIP_CHAIN_IS09 1
#define SMILEHIP_CHAIN_COMPARE_AB 2
#define SMILEHIP_CHAIN_PLP 3
#define SMILEHIP_CHAIN_COMPARE_F0 4
#define SMILEHIP_CHAIN_COMPARE 5
/* SMILEHIP_CHAIN_EGEMAPS: the LLD level of config/egemaps/v02/eGeMAPSv02.conf (with config/gemaps/v01b/GeMAPSv01b_core.lld.conf.inc;
 * BASELINE config 5), 25 columns, T60 + 1 rows per utterance (T60 = frames of the 60 ms framer; none if the input is shorter):
 *   Loudness, alphaRatio, hammarbergIndex, slope0-500, slope500-1500, spectralFlux, mfcc1..4 (_sma3: [egemapsv02_smoE]) |
 *   F0semitoneFrom27.5Hz, jitterLocal, shimmerLocaldB, HNRdBACF, logRelF0-H1-H2, logRelF0-H1-A3, F1frequency, F1bandwidth,
 *   F1amplitudeLogRelF0, F2..., F3... (_sma3nz: [egemapsv02_smoFnz])
 * from cSpectral with the GeMAPS option sets (src/lldcore/spectral.cpp:586-1254), cPlp as auditory spectrum + cVectorOperation
 * ll1, cMfcc 1..4, cSpecResample -> cLpc -> cFormantLpc (src/dsp/specResample.cpp:175, src/lld/lpc.cpp:171,
 * src/lld/formantLpc.cpp:192), the F0 group (cSpecScale, cPitchShs, cPitchSmootherViterbi with F0finalLog, cValbasedSelector),
 * cPitchJitter (jitterLocal, shimmerLocalDB), cHarmonics (src/lld/harmonics.cpp:743), the voiced / unvoiced cValbasedSelector
 * gates and cContourSmoother with the graph's end-of-input rules. */
#define SMILEHIP_CHAIN_EGEMAPS 6

#define SMILEHIP_STAGE_WINDOW 1u
#define SMILEHIP_STAGE_FFT    2u
#define SMILEHIP_STAGE_MEL    4u
#define SMILEHIP_STAGE_MFCC   8u
#define SMILEHIP_STAGE_ALL    15u
#define SMILEHIP_STAGE_SPECTRAL 16u   /* cSpectral constants for a K-bin spectrum (single-component plans) */

/* Integer geometry derived from the config -- the part of the contract that
 * must be bit-exact. */
typedef struct smilehip_geometry {
  int64_t frame_size;      /* N  = frameSizeFrames */
  int64_t frame_step;      /* H  = frameStepFrames */
  int64_t fft_size;        /* Nfft */
  int64_t n_bins;          /* K = Nfft/2+1 */
  int32_t n_static;        /* lastMfcc-firstMfcc+1 */
  int32_t n_out;           /* n_static*(1+n_delta): columns of the output */
  double  frame_period;    /* seconds between frames: level period after cFramer */
  double  fft_frame_size_sec; /* frameSizeSec after cTransformFFT (transformFft.cpp:66-96) */
} smilehip_geometry;

typedef struct smilehip_context smilehip_context;
typedef struct smilehip_plan smilehip_plan;
typedef struct smilehip_batch smilehip_batch;

/* ------------------------------------------------------------ life cycle */
int  smilehip_version(void);
/* thread-local message of the last failing call (never NULL) */
const char *smilehip_last_error(void);
/* Opens HIP device `device`; fails (SMILEHIP_ERR_NO_DEVICE) unless it is a
 * gfx950 part. There is NO CPU fallback anywhere in this library. */
int  smilehip_init(int device, smilehip_context **ctx);
void smilehip_shutdown(smilehip_context *ctx);
int  smilehip_device_name(smilehip_context *ctx, char *buf, int buflen);

/* fills c with the LLD part of config/is09-13/IS09_emotion.conf (chain_kind = IS09) */
void smilehip_config_is09_lld(smilehip_lld_config *c);

/* fills c with config/plp/PLP_0_D_A.conf (chain_kind = PLP): 6 PLP cepstra + delta + accel */
void smilehip_config_plp_0_d_a(smilehip_lld_config *c);
/* fills c with groups A+B of config/compare16/ComParE_2016.conf (chain_kind = COMPARE_AB) */
void smilehip_config_compare16_ab(smilehip_lld_config *c);
/* fills c with one of the eight files of config/mfcc and config/plp, by name without the extension: MFCC12_0_D_A,
 * MFCC12_E_D_A, MFCC12_0_D_A_Z, MFCC12_E_D_A_Z, PLP_0_D_A, PLP_E_D_A, PLP_0_D_A_Z, PLP_E_D_A_Z (E: cepstra 1.. + HTK log
 * energy; Z: cepstral mean subtraction, and the default symmetric zero padding those files leave in place).
 * Returns SMILEHIP_ERR_INVALID for any other name. */
int smilehip_config_htk_variant(smilehip_lld_config *c, const char *name);
/* fills c with the F0 group of config/compare16/ComParE_2016.conf up to is13_pitchG60 (chain_kind = COMPARE_F0) */
void smilehip_config_compare16_f0(smilehip_lld_config *c);
/* fills c with the whole LLD level of config/compare16/ComParE_2016.conf (chain_kind = COMPARE, 130 columns) */
void smilehip_config_compare16(smilehip_lld_config *c);
/* ... of config/is09-13/IS13_ComParE.conf: the same graph with zeroPadSymmetric = 0 in both cTransformFFT instances and
 * useBrokenJitterThresh = 1 (IS13_ComParE_core.lld.conf.inc vs ComParE_2016_core.lld.conf.inc) */
void smilehip_config_is13_compare(smilehip_lld_config *c);

/* fills c with the LLD level of config/egemaps/v02/eGeMAPSv02.conf (chain_kind = EGEMAPS, 25 columns) */
void smilehip_config_egemapsv02(smilehip_lld_config *c);
/* ... with the option values of the v01a files (config/gemaps/v01a/GeMAPSv01a_core.lld.conf.inc: zeroPadSymmetric = 0 in both
 * cTransformFFT instances :34,67, useBrokenJitterThresh = 1 :199, cFormantLpc maxF = 5500 :281; everything else is the v01b /
 * v02 graph): GeMAPSv01a.conf and eGeMAPSv01a.conf write column subsets of this level and of its 88 functionals, the same
 * subsets GeMAPSv01b.conf / eGeMAPSv01b.conf take of eGeMAPSv02's */
void smilehip_config_egemapsv01a(smilehip_lld_config *c);

/* F0 group, per component, on an F0 chain plan (smilehip_config_compare16_f0; the plan's spectrum geometry -- n_bins and
 * the level's frameSizeSec, force_fft_frame_size_sec -- must be the input level's):
 * cSpecScale::processVector (src/dsp/specScale.cpp:305-357) with scale=octave, sourceScale=lin, interpMethod=spline,
 * minF=25, maxF=-1, nPointsTarget=0, specSmooth=specEnhance=auditoryWeighting=1: n_bins magnitudes -> n_bins values */
int smilehip_specscale_frames(smilehip_plan *plan, const float *d_mag, int64_t ld_src, float *d_dst, int64_t ld_dst,
                              int64_t n_frames, void *stream);
/* cPitchBase::processVector around cPitchShs::pitchDetect (src/lldcore/pitchBase.cpp:187-310, src/lld/pitchShs.cpp:214-347;
 * nCandidates=6, scores=voicing=1, F0raw=voicingClip=1, greedyPeakAlgo=1, octaveCorrection=0): n_bins octave-scale values ->
 * 21 values [nCandidates | F0Cand[6] | candVoicing[6] | candScores[6] | F0raw | voicingClip] */
int smilehip_pitchshs_frames(smilehip_plan *plan, const float *d_hps, int64_t ld_src, float *d_dst, int64_t ld_dst,
                             int64_t n_frames, void *stream);

/* F0 chain taps (tests / diagnostics): device pointers to the per-frame scratch the last smilehip_lld_run of this
 * batch filled -- candidates [total_frames x 21] = level is13_pitchShsG60 (nCandidates | F0Cand[6] | candVoicing[6] |
 * candScores[6] | F0raw | voicingClip), energies [total_frames] = level is13_e60 -- and an optional caller-owned
 * destination [total_frames x n_bins] for level is13_hpsG60 (NULL switches the tap off). */
int smilehip_batch_f0_taps(smilehip_batch *b, float *d_hps_dst, const float **d_shs, const float **d_e60);
/* Frames per utterance that cPitchSmootherViterbi had not decided when the input ended (they are flushed at the end of input:
 * src/lld/pitchSmootherViterbi.cpp:451-570), [n_utt] device ints filled by the last smilehip_lld_run of an F0 / ComParE / eGeMAPS
 * chain batch. Every end-of-input rule of the smoothed levels and the row counts the F0-group functionals see follow from it. */
int smilehip_batch_f0_pending(smilehip_batch *b, const int32_t **d_pending);

/* ---- functionals over the LLD level ---------------------------------
 * cFunctionals in frameMode=full (src/functionals/functionals.cpp:284-330, one output vector
 * per utterance) with cFunctionalExtremes (functionalExtremes.cpp:92-134, norm=frame),
 * the linear part of cFunctionalRegression (functionalRegression.cpp:140-425, normInputs =
 * normRegCoeff = doRatioLimit = 0) and cFunctionalMoments (functionalMoments.cpp:88-165).
 * Bit i of the mask enables value i; values keep the reference's order (functionalsEnabled =
 * Extremes;Regression;Moments), the output vector is element-major: column c's values sit at
 * [c*count, (c+1)*count) like the reference's func level (functionals.cpp:233-262). */
#define SMILEHIP_FUNC_MAX          (1u << 0)
#define SMILEHIP_FUNC_MIN          (1u << 1)
#define SMILEHIP_FUNC_RANGE        (1u << 2)
#define SMILEHIP_FUNC_MAXPOS       (1u << 3)
#define SMILEHIP_FUNC_MINPOS       (1u << 4)
#define SMILEHIP_FUNC_AMEAN        (1u << 5)
#define SMILEHIP_FUNC_MAXAMEANDIST (1u << 6)
#define SMILEHIP_FUNC_MINAMEANDIST (1u << 7)
#define SMILEHIP_FUNC_LINREGC1     (1u << 8)
#define SMILEHIP_FUNC_LINREGC2     (1u << 9)
#define SMILEHIP_FUNC_LINREGERRA   (1u << 10)
#define SMILEHIP_FUNC_LINREGERRQ   (1u << 11)
#define SMILEHIP_FUNC_VARIANCE     (1u << 12)
#define SMILEHIP_FUNC_STDDEV       (1u << 13)
#define SMILEHIP_FUNC_SKEWNESS     (1u << 14)
#define SMILEHIP_FUNC_KURTOSIS     (1u << 15)
#define SMILEHIP_FUNC_AMEAN_M      (1u << 16)
#define SMILEHIP_FUNC_ALL          0x1ffffu
/* the 12 functionals of config/is09-13/IS09_emotion_core.func.conf.inc */
uint32_t smilehip_functionals_is09_mask(void);
/* values per LLD column for a mask; < 0 on unknown bits */
int smilehip_functionals_count(uint32_t mask);
/* Number of LLD rows each utterance's functionals summarise, rows[n_utt] (host). In full mode
 * cFunctionals consumes what its input levels hold at its first end-of-input tick
 * (src/core/winToVecProcessor.cpp:504-528, 868-1098): for the IS09 chain (SMA(3) -> delta(2))
 * that is max(1, T-2) of the T+1 rows the LLD sinks get; 0 for utterances without a frame. */
int smilehip_batch_func_rows(const smilehip_batch *batch, int64_t *rows);
/* d_lld: the matrix smilehip_lld_run produced for this batch (IS09 chain plans only);
 * d_func: n_utt x ld_func, n_out * count(mask) values per utterance (zeros for utterances
 * without a frame, where the reference writes no instance). Asynchronous on `stream`. */
int smilehip_batch_functionals(smilehip_plan *plan, smilehip_batch *batch, const float *d_lld, int64_t ld_lld,
                               uint32_t mask, float *d_func, int64_t ld_func, void *stream);

/* The same functionals over ONE matrix (rows x cols, leading dimension ld_x), all rows: what
 * cFunctionals::doProcess computes for one input row of length `rows` (cols = 1) or for a block of them;
 * d_out receives cols * count(mask) floats, element-major. */
int smilehip_functionals_matrix(smilehip_context *ctx, const float *d_x, int64_t ld_x, int64_t rows, int32_t cols,
                                uint32_t mask, float *d_out, void *stream);

/* ---- general functionals: any cFunctionals instance -------------------
 * One instance = the ordered family list `functionalsEnabled` (src/functionals/functionals.cpp:179-222) plus the
 * options of each family. Bit i of a family mask enables the family's value i in the reference's own enab[] order,
 * which is also its output order:
 *   Extremes    (functionalExtremes.cpp:91-134)    max min range maxPos minPos amean maxameandist minameandist
 *   Means       (functionalMeans.cpp:115-259)      amean absmean qmean nzamean nzabsmean nzqmean nzgmean nnz flatness
 *                                                  posamean negamean posqmean posrqmean negqmean negrqmean rqmean nzrqmean
 *   Moments     (functionalMoments.cpp:88-165)     variance stddev skewness kurtosis amean stddevNorm
 *   Regression  (functionalRegression.cpp:142-425) linregc1 linregc2 linregerrA linregerrQ qregc1 qregc2 qregc3 qregerrA
 *                                                  qregerrQ centroid qregls qregrs qregx0 qregy0 qregyr qregy0nn qregc3nn qregyrnn
 *   Percentiles (functionalPercentiles.cpp:312-417) quartile1..3 iqr1-2 iqr2-3 iqr1-3, percentile[], pctlrange[]
 *   Times       (functionalTimes.cpp:213-367)      up/downleveltime25,50,75,90 risetime falltime leftctime rightctime duration
 *   Segments    (functionalSegments.cpp:309-367, 656-799, 801-958; relTh, nonX and eqX) numSegments meanSegLen maxSegLen
 *                                                  minSegLen segLenStddev
 *   Lpc         (functionalLpc.cpp:95-119)         lpgain, lpc[first..order)
 *   Peaks2      (functionalPeaks2.cpp:316-905)     its 32 values in the order of functionalPeaks2.cpp:60-67
 *   Onset       (functionalOnset.cpp:83-151)       onsetPos offsetPos numOnsets numOffsets onsetRate
 *   Peaks       (functionalPeaks.cpp:98-214, the older peak picker; overlapFlag = 1) numPeaks meanPeakDist peakMean
 *                                                  peakMeanMeanDist peakDistStddev
 *   Crossings   (functionalCrossings.cpp:66-97)    zcr mcr amean
 *   DCT         (functionalDCT.cpp:84-137)         dct[first .. last]: the table entry (float)cos(pi i / N (m + 0.5)) is formed on the device
 *   Samples     (functionalSamples.cpp:100-117)    the contour at n_samples (<= 8) relative positions
 * Time norms: 0 = segment, 1 = second, 2 = frame (functionalComponent.hpp:27-33) -- the value AFTER the reference's
 * precedence rule (the family's own `norm` if set, else cFunctionals.masterTimeNorm, else the family default).
 * Not restated: Segments.useOldBuggyChX / growDynSegBuffer, Peaks2.noClearPeakList / debug outputs, Times.useRobustPercentileRange,
 * Modulation over the whole contour (stftWinSize 0) -- a spec cannot express them. */
enum {
  SMILEHIP_FAM_EXTREMES = 0, SMILEHIP_FAM_MEANS, SMILEHIP_FAM_MOMENTS, SMILEHIP_FAM_REGRESSION, SMILEHIP_FAM_PERCENTILES,
  SMILEHIP_FAM_TIMES, SMILEHIP_FAM_SEGMENTS, SMILEHIP_FAM_LPC, SMILEHIP_FAM_PEAKS2, SMILEHIP_FAM_ONSET, SMILEHIP_FAM_PEAKS, SMILEHIP_FAM_CROSSINGS, SMILEHIP_FAM_DCT,
  SMILEHIP_FAM_SAMPLES, SMILEHIP_FAM_MODULATION, SMILEHIP_FAM_COUNT
};
enum { SMILEHIP_NORM_SEGMENT = 0, SMILEHIP_NORM_SECOND = 1, SMILEHIP_NORM_FRAME = 2 };
/* Segments.segmentationAlgorithm (functionalSegments.cpp:118-155). The reference parses ltX / gtX / geqX / leqX but has no
 * code for them: its switch falls to the delta method (:872-874) -- pass SMILEHIP_SEG_DELTA for those. absTh never reads its
 * thresholds (:179-180): it finds no segment, here as there. */
enum { SMILEHIP_SEG_RELTH = 0, SMILEHIP_SEG_NONX = 1, SMILEHIP_SEG_EQX = 2, SMILEHIP_SEG_MRELTH = 3, SMILEHIP_SEG_ABSTH = 4,
       SMILEHIP_SEG_NARELTH = 5, SMILEHIP_SEG_NAMRELTH = 6, SMILEHIP_SEG_NAABSTH = 7, SMILEHIP_SEG_DELTA = 8,
       SMILEHIP_SEG_DELTA2 = 9, SMILEHIP_SEG_CHX = 10 };

typedef struct smilehip_func_spec {
  int32_t n_fam;
  int32_t fam[12];              /* functionalsEnabled, in order */
  int32_t non_zero_functs;      /* cFunctionals.nonZeroFuncts: 0, 1 (x != 0), 2 (x > 0) */
  int32_t reserved0;
  double period;                /* period of the input level in seconds */
  uint32_t ext_mask; int32_t ext_norm;
  uint32_t means_mask; int32_t means_norm;
  uint32_t mom_mask; int32_t mom_stddev_norm; int32_t mom_ratio_limit; int32_t reserved1;
  uint32_t reg_mask; int32_t reg_centroid_norm, reg_norm_coeff, reg_norm_inputs, reg_centroid_abs,
      reg_centroid_limit, reg_ratio_limit, reg_old_buggy_qerr;
  uint32_t pct_mask; int32_t pct_interp, n_pctl, n_range;
  double pctl[8]; int32_t range_a[8], range_b[8];
  uint32_t times_mask; int32_t times_norm, times_buggy_sec_norm, reserved2;
  uint32_t seg_mask; int32_t seg_norm, seg_algo, seg_max_num, seg_min_lng, seg_auto_min_lng, seg_pause_min_lng,
      seg_x_is_rel, seg_n_thresholds, seg_ravg_lng;        /* ravgLng of delta / delt2; <= 0: Nin / (maxNumSeg / 2) */
  float seg_x; float seg_thresholds[8]; float seg_range_rel_threshold;   /* rangeRelThreshold of delta / delt2 */
  int32_t lpc_gain, lpc_coeffs, lpc_first, lpc_order;      /* order <= 16 */
  uint32_t pk_mask; int32_t pk_norm, pk_ratio_limit, pk_dyn_rel, pk_use_abs, reserved5;
  float pk_rel_thresh, pk_abs_thresh;
  uint32_t ons_mask; int32_t ons_norm, ons_use_abs, reserved6;   /* Onset: thresholdOnset / thresholdOffset (= threshold unless set) */
  float ons_thr_on, ons_thr_off;
  uint32_t pko_mask; int32_t pko_norm;                             /* Peaks */
  uint32_t crs_mask; int32_t dct_first, dct_last, n_samples;       /* Crossings; DCT firstCoeff .. lastCoeff; Samples */
  double sample_pos[8];                                            /* Samples.samplepos[], clipped to [0, 1] */
  /* Percentiles.pctlquotient[] (functionalPercentiles.cpp:179-232, :402-411): percentile quot_a over percentile quot_b through
   * the soft limiter (50, 100); an index < 0 gives 0. The reference forms the quotients only when pctlrange[] is not empty
   * (zeros otherwise) and only with n_pctl > 0 -- so does this. */
  int32_t n_quot, quot_a[8], quot_b[8];
  /* Times.upleveltime[] / downleveltime[] (functionalTimes.cpp:129-165, :347-364): the share of the contour above / not above
   * level * range + min, levels clipped to [0, 1]. (Times.useRobustPercentileRange is not built: a spec cannot ask for it.) */
  int32_t n_ul, n_dl, reserved7;
  double ul[8], dl[8];
  /* Modulation (cFunctionalModulation, functionalModulation.cpp:452-566): mod_n_bins ModulationSpec values per contour -- the
   * magnitude spectra of windows of mod_win_frames values every mod_step_frames (window function SMILEHIP_WIN_* of the window's
   * own length, zero padding to the next power of two, the reference's rdft), mapped by a natural cubic spline onto mod_min_freq +
   * i (mod_max_freq - mod_min_freq) / mod_n_bins Hz and averaged. Built for windows of 33 .. 1024 values (a contour needs >= 34
   * rows) and <= 128 bins; stftWinSize 0 (one transform over the whole contour) is not built. */
  /* Limits: 49 <= mod_win_frames <= 1024 (refused otherwise: the reference's transforms of fewer than 64 points -- windows of
   * fewer than 33 values -- are not built); a contour of fewer than 34 values has no window of 33 values and gets NaN in every
   * Modulation bin (the reference computes it with a 4 .. 32-point transform): callers that may see such contours check the row
   * count first, as the plugin's cFunctionals override does. */
  int32_t mod_win_frames, mod_step_frames, mod_n_bins, mod_win_func, mod_remove_nz_mean, reserved8;
  double mod_min_freq, mod_max_freq;
} smilehip_func_spec;

/* values per input column; < 0 (and smilehip_last_error) for a spec this library cannot run */
int smilehip_funcspec_count(const smilehip_func_spec *spec);
/* The six cFunctionals instances of config/compare16/ComParE_2016_core.func.conf.inc: "A", "B", "F0", "Nz", "LLD",
 * "Delta" ([is13_functionalsA] ... [is13_functionalsDelta]); period = 0.01 s. */
int smilehip_funcspec_compare16(const char *instance, smilehip_func_spec *spec);
/* The same six instances as config/is09-13/IS13_ComParE_core.func.conf.inc configures them: no soft limiting of ratio
 * features (Moments / Regression / Peaks2 doRatioLimit = 0, centroidRatioLimit = 0), centroid of the signed values,
 * normInputs = 0, normRegCoeff = 0. */
int smilehip_funcspec_is13_compare(const char *instance, smilehip_func_spec *spec);
/* cFunctionals::doProcess for every column of ONE matrix (rows x cols, leading dimension ld_x, all rows): d_out
 * receives cols * count(spec) floats, element-major (column c's values at [c*count, (c+1)*count)). Scratch is owned
 * by the context and grown on demand (growing synchronises the device once). Asynchronous on `stream`; calls that use
 * the same context's scratch must be issued in stream order from one host thread at a time -- use one context per
 * host thread otherwise. */
int smilehip_funcspec_matrix(smilehip_context *ctx, const smilehip_func_spec *spec, const float *d_x, int64_t ld_x,
                             int64_t rows, int32_t cols, float *d_out, void *stream);
/* The same over a batch's LLD matrix: for each utterance, columns [col_first, col_first + n_cols) of its rows
 * 0 .. rows_u - rows_cut - 1 (at least one row if the utterance has any), optionally followed by ONE more row taken from
 * d_extra[u * ld_extra + (0 .. n_cols)] (NULL: none). d_func: n_utt x ld_func, n_cols * count(spec) values per
 * utterance (zeros for utterances without rows). */
int smilehip_batch_funcspec(smilehip_plan *plan, smilehip_batch *batch, const smilehip_func_spec *spec,
                            const float *d_lld, int64_t ld_lld, int32_t col_first, int32_t n_cols, int32_t rows_cut,
                            const float *d_extra, int64_t ld_extra, float *d_func, int64_t ld_func, void *stream);

/* The whole functionals level of config/compare16/ComParE_2016.conf: 6373 values per utterance in the reference's own
 * order ([functionals] concatenates is13_functionalsA, B, Nz, F0, LLD, Delta; names: smilehip_host func_names_compare16).
 * d_lld: the 130-column matrix smilehip_lld_run produced for THIS batch on a smilehip_config_compare16 plan (the run also
 * leaves what the functionals need beyond that matrix: the Viterbi smoother's undecided-frame counts and row T60+1 of
 * group B's levels). Rows each instance summarises -- decided in the reference by its first end-of-input tick
 * (winToVecProcessor.cpp:504-528, 868-1098), measured against the binary -- with T = rows - 1 and P = frames the
 * smoother had not decided (0 if P >= T): A T-2, B T+2, Nz T-P-2, F0 T-P, LLD T, Delta T-2 (at least 1).
 * d_func: n_utt x ld_func (zeros for utterances without rows, where the reference writes no instance). */
int smilehip_functionals_compare16_count(void);
int smilehip_batch_functionals_compare16(smilehip_plan *plan, smilehip_batch *batch, const float *d_lld, int64_t ld_lld,
                                         float *d_func, int64_t ld_func, void *stream);
/* The functionals level of config/is09-13/IS13_ComParE.conf (same 6373 elements and names, the instances' options of
 * smilehip_funcspec_is13_compare) on a smilehip_config_is13_compare plan. */
int smilehip_batch_functionals_is13_compare(smilehip_plan *plan, smilehip_batch *batch, const float *d_lld, int64_t ld_lld,
                                            float *d_func, int64_t ld_func, void *stream);
/* Row T60+1 of group B's sma / delta levels, [n_utt x 110] device floats filled by the last smilehip_lld_run of a
 * ComParE chain batch (tests / callers that run single instances through smilehip_batch_funcspec). */
int smilehip_batch_compare_b_extra(smilehip_batch *batch, const float **d_extra);

/* ---- eGeMAPSv02 functionals (config/gemaps/v01b/GeMAPSv01b_core.func.conf.inc + config/egemaps/v02/eGeMAPSv02_core.func.conf.inc)
 * The cFunctionals instances as specs of the general engine: "F0" / "Loudness" ([gemapsv01b_functionalsF0] / [..Loudness]),
 * "MVZ" ([egemapsv02_functionalsMVR]), "MVV" ([egemapsv02_functionalsMVRVoiced]), "MU" ([egemapsv02_functionalsMeanUV]), "numPeaks"
 * ([gemapsv01b_temporalLoudness]), "segF0" / "segF0pause" ([gemapsv01b_temporalF0] / [..F0p]), "leq" ([egemapsv02_leqLin]). */
int smilehip_funcspec_egemaps(const char *instance, smilehip_func_spec *spec);
/* The whole functionals level: 88 values per utterance in [funcconcat]'s order (gemapsv01b_functionalsF0 (10),
 * gemapsv01b_functionalsLoudness (10), egemapsv02_functionalsMeanStddevZ (10), ..MeanStddevVoiced (46), ..MeanUnvoiced (5),
 * gemapsv01b_temporalSet (6), egemapsv02_leq (1, after cVectorOperation dBp); names: smilehip_host func_names_egemaps).
 * Uses what the last smilehip_lld_run of THIS batch (a smilehip_config_egemapsv02 plan) left on the device: the smoothed
 * levels the instances read, which hold more rows than the LLD matrix. Rows each instance summarises -- decided in the reference
 * by its first end-of-input tick (winToVecProcessor.cpp:504-528, 868-1098), measured against the binary -- with T20 / T60 the
 * frames of the 20 ms / 60 ms framers and P the frames the Viterbi smoother had not decided at the end of input: instances on
 * 20 ms levels T20; instances that follow the Viterbi smoother max(1, T60 - P); MeanStddevVoiced, which also waits for
 * cPitchJitter, T60 - P, or T60 when P = T60. d_func: n_utt x ld_func (zeros for utterances without a 60 ms frame, where the
 * reference writes no instance). */
int smilehip_functionals_egemaps_count(void);
int smilehip_batch_functionals_egemaps(smilehip_plan *plan, smilehip_batch *batch, float *d_func, int64_t ld_func, void *stream);
/* eGeMAPS chain taps (tests / diagnostics): device pointers to the per-frame scratch the last smilehip_lld_run of this batch
 * filled. raw20 [frames20 x 12]: loudness | slope0-500, slope500-1500, alphaRatio, hammarbergIndex | flux | mfcc1..4 | energy2 | 0;
 * lpc [frames20 x 12]: 11 LP coefficients; formants [frames20 x 10]: 5 frequencies | 5 bandwidths; pitch3 [frames60 x 3]: F0final,
 * F0finalLog, voicingFinalUnclipped (level gemapsv01b_logPitch); jit4 [frames60 x 4]: jitterLocal in column 0; shim_db [frames60];
 * harm6 [frames60 x 6]: level gemapsv01b_harmonics; func_in [(T20+1 rows per utterance with a 60 ms frame) x 36]: the levels the
 * functionals read; pending [n_utt]. Any pointer may be NULL. h_frame_off60[n_utt+1] (host, may be NULL): 60 ms frame offsets. */
int smilehip_batch_egemaps_taps(smilehip_batch *batch, const float **d_raw20, const float **d_lpc, const float **d_formants,
                                const float **d_pitch3, const float **d_jit4, const float **d_shim_db, const float **d_harm6,
                                const float **d_func_in, const int32_t **d_pending, int64_t *h_frame_off60);

/* Plain device-memory plumbing for hosts that do not link the HIP runtime
 * themselves (the openSMILE plugin is compiled with the host g++ only). */
int  smilehip_alloc(smilehip_context *ctx, uint64_t bytes, void **d_ptr);
int  smilehip_free(smilehip_context *ctx, void *d_ptr);
int  smilehip_copy_to_device(smilehip_context *ctx, void *d_dst, const void *h_src, uint64_t bytes, void *stream);
int  smilehip_copy_to_host(smilehip_context *ctx, void *h_dst, const void *d_src, uint64_t bytes, void *stream);
int  smilehip_stream_synchronize(smilehip_context *ctx, void *stream);
/* Block cache for the device memory of batches and plans: up to `bytes` of the blocks the library frees are kept and handed out
 * again for allocations of exactly their size (0, the default: off, and the kept blocks are freed). For hosts that create and
 * destroy one batch per chunk of files while the next chunk's copies and kernels are already enqueued on another stream:
 * hipFree synchronises the whole device, i.e. would wait for them. With the cache on, smilehip_batch_destroy no longer implies
 * that synchronisation -- the caller destroys a batch only after the stream its work ran on has drained. Process-wide. */
int  smilehip_alloc_cache(smilehip_context *ctx, uint64_t bytes);
/* A stream of the caller's own (non-blocking with respect to the null stream): what a host that overlaps the copies and kernels of
 * consecutive chunks needs and cannot create without linking the HIP runtime (smilextract_hip: chunk k's copy-out beside chunk
 * k + 1's copy-in and kernels). Every entry point with a `stream` argument takes it. */
int  smilehip_stream_create(smilehip_context *ctx, void **stream);
int  smilehip_stream_destroy(smilehip_context *ctx, void *stream);
/* ... and the ordering between two such streams: an event (no timing) recorded on one, awaited by the other -- "chunk k + 1's
 * kernels start when chunk k's kernels are done" (the library's context-wide scratch, e.g. the functionals', serves one run at a
 * time) while the copies either side run on. */
int  smilehip_event_create(smilehip_context *ctx, void **event);
int  smilehip_event_destroy(smilehip_context *ctx, void *event);
int  smilehip_event_record(smilehip_context *ctx, void *event, void *stream);
int  smilehip_stream_wait_event(smilehip_context *ctx, void *stream, void *event);
/* The same copies for `rows` rows of `width_bytes` bytes with a row pitch on either side: a block of frames of a dataMemory level
 * (cMatrix::data, [frames x N] floats: src/core/dataMemoryLevel.cpp:1530-1582 setMatrix / :1651-1740 getMatrix) against a device
 * block of another width. What the plugin's block-per-tick overrides move per tick (one call per level instead of one per frame). */
int  smilehip_copy_to_device_2d(smilehip_context *ctx, void *d_dst, uint64_t d_pitch, const void *h_src, uint64_t h_pitch,
                                uint64_t width_bytes, uint64_t rows, void *stream);
int  smilehip_copy_to_host_2d(smilehip_context *ctx, void *h_dst, uint64_t h_pitch, const void *d_src, uint64_t d_pitch,
                              uint64_t width_bytes, uint64_t rows, void *stream);
/* Page-locked host memory for the staging buffers of a file-to-file host (the copies above run at the link's rate only from
 * such memory; what cWaveSource's read buffer and the sinks' write buffers are to the reference: src/iocore/waveSource.cpp:240-294,
 * src/iocore/htkSink.cpp:183-202). */
int  smilehip_alloc_host(smilehip_context *ctx, uint64_t bytes, void **h_ptr);
int  smilehip_free_host(smilehip_context *ctx, void *h_ptr);
/* ... and page-locking of memory the caller owns already (the plugin's block matrices are cMatrix objects of the data memory,
 * src/core/dataMemoryLevel.cpp:459-472: their storage is the reference's own allocation). */
int  smilehip_host_register(smilehip_context *ctx, void *h_ptr, uint64_t bytes);
int  smilehip_host_unregister(smilehip_context *ctx, void *h_ptr);
/* cHtkSink::myTick's byte order (src/iocore/htkSink.cpp:183-202: every value of a vector through smileHtk_SwapFloat on a
 * little-endian host) done on the device: d_dst[i] = big-endian image of d_src[i], n values; d_dst may equal d_src. The sink
 * then writes a file's rows with one write(). */
int  smilehip_htk_rows_be(smilehip_context *ctx, const float *d_src, int64_t n, void *d_dst, void *stream);

/* fills c with config/mfcc/MFCC12_0_D_A.conf's values */
void smilehip_config_mfcc12_0_d_a(smilehip_lld_config *c);

/* Replaces the configure/finalise phase of the chain's components
 * (cDataProcessor::myFinaliseInstance, src/core/dataProcessor.cpp:548-596 ->
 * cWindower::precomputeWinFunc windower.cpp:159, cMelspec::computeFilters
 * melspec.cpp:184, cMfcc::initTables mfcc.cpp:136): builds all tables on the
 * host with the reference's double->float rounding and uploads them once. */
int  smilehip_plan_create(smilehip_context *ctx, const smilehip_lld_config *cfg, smilehip_plan **plan);
/* Tables and geometry only, no device needed (table-level parity tests, config
 * validation on a host without a GPU). Every compute entry point refuses such
 * a plan with SMILEHIP_ERR_NO_DEVICE. */
int  smilehip_plan_create_host_only(const smilehip_lld_config *cfg, smilehip_plan **plan);
void smilehip_plan_destroy(smilehip_plan *plan);
int  smilehip_plan_geometry(const smilehip_plan *plan, smilehip_geometry *g);
/* cWinToVecProcessor framing rule: T = floor((S-N)/H)+1, 0 if S<N
 * (src/core/winToVecProcessor.cpp:872-877) */
int64_t smilehip_num_frames(const smilehip_plan *plan, int64_t n_samples);
/* frame t's time stamp = t*H/fs (cMatrix::squashTimeMeta,
 * src/core/dataMemoryLevel.cpp:617-626) and vIdx = t */
double  smilehip_frame_time(const smilehip_plan *plan, int64_t t);
/* Time stamp (TimeMetaInfo::time, what cCsvSink/cArffSink print as frameTime) of output row
 * `row` of an utterance with n_frames frames. MFCC chain: row * frame period. IS09 / ComParE
 * chains: the rows a window processor emits at end of input carry the time of the last real
 * frame, i.e. min(row, n_frames-1) * period -- except for n_frames == 1, where the binary
 * stamps the extra row with `period` (measured: tests/golden/files, test_host_io.py). eGeMAPS chain: row * period (the
 * level's time stamps come from egemapsv02_lldsetE_smo, whose rows 0 .. T60 are regular frames of the 20 ms chain). */
double  smilehip_row_time(const smilehip_plan *plan, int64_t n_frames, int64_t row);
/* host copies of the generated tables (for table-level parity tests); each
 * returns the element count or a negative status. out may be NULL. */
int64_t smilehip_plan_get_window(const smilehip_plan *plan, float *out, int64_t cap);
int64_t smilehip_plan_get_mel_weights(const smilehip_plan *plan, float *out, int64_t cap);
int64_t smilehip_plan_get_mel_chanmap(const smilehip_plan *plan, int32_t *out, int64_t cap);
int64_t smilehip_plan_get_dct(const smilehip_plan *plan, float *out, int64_t cap);
int64_t smilehip_plan_get_lifter(const smilehip_plan *plan, float *out, int64_t cap);

/* ------------------------------------------------------- batched hot path */
/* A batch = a set of utterances packed back to back in one int16 buffer.
 * h_sample_offsets[n_utt+1]: utterance u occupies samples
 * [off[u], off[u+1]) of the PCM buffer. The batch object owns the device-side
 * work tables (utterance/frame offsets, tile table) and may be reused for any
 * PCM buffer with the same layout. This replaces the reference's per-tick
 * control flow (cComponentManager::tick, componentManager.cpp:1233-1261): the
 * batch dimension the GPU needs does not exist there. */
int  smilehip_batch_create(smilehip_plan *plan, const int64_t *h_sample_offsets, int32_t n_utt,
                           smilehip_batch **batch);
void smilehip_batch_destroy(smilehip_batch *batch);
int64_t smilehip_batch_total_frames(const smilehip_batch *batch);
/* rows of the output matrix (= frames for the MFCC chain; frames + 1 per non-empty
 * utterance for the IS09 chain, whose smoother emits one end-of-input frame) */
int64_t smilehip_batch_total_rows(const smilehip_batch *batch);
/* 1 if the batch of an MFCC / PLP chain (fast kernel, two regression stages of window 2, no log energy / mean subtraction)
 * holds the tiles of the delta-fused frame kernel: smilehip_mfcc_run then computes cDeltaRegression's two stages
 * (deltaRegression.cpp:144-152) inside the frame kernel whenever the PCM pointer is dword-aligned and every utterance starts
 * at an even sample, and the window-chain kernel runs only for utterances of <= 16 frames. 0: the window chain runs. */
int  smilehip_batch_delta_fused(const smilehip_batch *batch);
/* h_row_offsets[n_utt+1]: row range of utterance u in the output matrix */
int  smilehip_batch_frame_offsets(const smilehip_batch *batch, int64_t *h_row_offsets);

/* The fused chain: R0 smilePcm_convertSamples (smileUtil.c:2527-2535) ->
 * R1 cFramer -> R2 cVectorPreemphasis::processVector -> R3
 * cWindower::processVector -> R4 cTransformFFT::processVector -> R5
 * cFFTmagphase::processVector -> R6 cMelspec::processVector -> R7
 * cMfcc::processVector -> R13 cDeltaRegression x n_delta + cVectorConcat.
 * d_pcm: int16 mono samples (device). d_out: total_frames x ld_out floats,
 * ld_out >= n_out; row = frame, columns [static | delta | accel].
 * Asynchronous on `stream`. */
int  smilehip_mfcc_run(smilehip_plan *plan, smilehip_batch *batch, const int16_t *d_pcm,
                       float *d_out, int64_t ld_out, void *stream);

/* The LLD set selected by chain_kind (smilehip_mfcc_run for SMILEHIP_CHAIN_MFCC plans;
 * for SMILEHIP_CHAIN_IS09: R0-R7 + R9 + R10 + R12 per frame, cPitchACF's smoother per
 * utterance, then cContourSmoother + cDeltaRegression with the reference's end-of-input
 * rules). d_out: total_rows x ld_out, ld_out >= n_out. Asynchronous on `stream`. */
int  smilehip_lld_run(smilehip_plan *plan, smilehip_batch *batch, const int16_t *d_pcm,
                      float *d_out, int64_t ld_out, void *stream);
/* The same chains on samples that are floats already: d_pcm_f32 = the batch's packed utterances after R0, i.e. the output of
 * smilehip_pcm_convert for ANY sample format / channel count (8 / 16 / 24 / 32-bit, mono mix-down of N channels: what
 * cWaveSource hands to the graph, waveSource.cpp:240-294 + smileUtil.c:2500-2627) or of smilehip_pcm16_to_float. Every
 * reference-order chain reads them in place of the int16 PCM -- same bits downstream; the MFCC / PLP chains run on the
 * reference-order kernel (the fast kernel's loader is int16-specific). */
int  smilehip_lld_run_f32(smilehip_plan *plan, smilehip_batch *batch, const float *d_pcm_f32,
                          float *d_out, int64_t ld_out, void *stream);
/* host-buffer convenience of the same (H2D, run, D2H, synchronises) */
int  smilehip_lld_run_host(smilehip_plan *plan, smilehip_batch *batch, const int16_t *h_pcm,
                           int64_t n_samples, float *h_out);

/* Convenience for hosts that hold plain memory (the plugin / batch driver):
 * H2D, run, D2H, synchronises. h_out must hold total_frames*n_out floats. */
int  smilehip_mfcc_run_host(smilehip_plan *plan, smilehip_batch *batch, const int16_t *h_pcm,
                            int64_t n_samples, float *h_out);

/* HIP-event timing of smilehip_mfcc_run on this plan, per kernel (ms):
 * smilehip_plan_set_timing(plan,1) resets the counters and makes every run
 * record events on its stream before/after each launch;
 * smilehip_plan_last_timing returns the AVERAGE launch duration over the runs
 * recorded since (at most the last 128) -- call it after synchronising. */
int  smilehip_plan_set_timing(smilehip_plan *plan, int enable);
int  smilehip_plan_last_timing(smilehip_plan *plan, float *ms_main, float *ms_delta);
/* Live timing of EVERY kernel the batch chains launch (bench.py's roofline objects of configs 3-5 name the kernel with the largest
 * share of a step from it): smilehip_kernel_timing(1) clears the records and brackets every launch from then on with two HIP events
 * on the stream the kernel is launched on; after the caller has synchronised, smilehip_kernel_timing_report writes one line per
 * kernel name -- "name<TAB>launches<TAB>summed ms" -- into buf and returns the text's length (negative: buflen too small, the
 * length needed negated). smilehip_kernel_timing(0) switches it off. Process-wide. */
int  smilehip_kernel_timing(int enable);
int64_t smilehip_kernel_timing_report(char *buf, int64_t buflen);

/* ------------------------------------- per-component batched entry points */
/* Same arithmetic, one reference component at a time, over n_frames frames;
 * used by the plugin in per-component mode and by the stage-level parity
 * tests. src/dst are device pointers, frame-major with leading dimensions. */

/* R0: smilePcm_convertSamples, 16-bit mono (smileUtil.c:2527-2535) */
int smilehip_pcm16_to_float(smilehip_context *ctx, const int16_t *d_pcm, int64_t n, float *d_out, void *stream);
/* R0, all sample formats of smilePcm_convertSamples (smileUtil.c:2500-2627): n_bps bytes per
 * sample (1, 2, 3, 4; n_bits = 24 or 32 selects packed-24 vs int32 when n_bps == 4), n_chan
 * interleaved channels, n sample frames. mono_mixdown != 0 (cWaveSource's default): n floats,
 * (sum_c s_c / n_chan) / full_scale; else n * n_chan floats, interleaved as in the input.
 * Full scales: 127, 32767, 32767*256, 2147483647. The fused batch path takes 16-bit mono. */
int smilehip_pcm_convert(smilehip_context *ctx, const void *d_raw, int n_bps, int n_bits, int n_chan, int mono_mixdown,
                         int64_t n, float *d_out, void *stream);
/* The same for IEEE-float wave files (sample type 3, 32 bit): smilePcm_convertFloatSamples, src/smileutil/smileUtil.c:2629-2690, which
 * smilePcm_readSamples picks for such files (smileUtil.c:2718-2722; cWaveSource::readData, waveSource.cpp:334-341). d_raw: n * n_chan floats, interleaved. mono_mixdown: the channels are
 * added to a float zero in channel order and the sum is divided by the channel count (one channel goes through the same two
 * operations); otherwise the samples are copied. */
int smilehip_pcm_convert_float(smilehip_context *ctx, const float *d_raw, int n_chan, int mono_mixdown, int64_t n, float *d_out,
                               void *stream);
/* R4 with inverse = 1 (src/dspcore/transformFft.cpp:196-216): rows of fft_size packed spectrum values (a[0] = X[0], a[1] = X[N/2],
 * a[2k] = Re, a[2k+1] = -Im, Ooura's packing) -> rdft(N, -1) -> fft_size samples, each times (FLOAT_DMEM)2 / N. The reference's bits. */
int smilehip_irfft_frames(smilehip_plan *plan, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst, int64_t n_frames,
                          void *stream);
/* R5, every output mode of cFFTmagphase::processVector (src/dspcore/fftmagphase.cpp:215-287). flags: 1 magnitude, 2 phase,
 * 4 normalise, 8 power, 16 dBpsd (with dbp_norm = dBpnorm and min_dbp = mindBp, :129-131). Rows of nfft packed values ->
 * [magnitude field (nfft/2 + 1) | phase field (nfft/2 + 1)], whichever were asked for (magnitude + phase together = joinMagphase). */
#define SMILEHIP_MAGPHASE_MAGNITUDE 1
#define SMILEHIP_MAGPHASE_PHASE 2
#define SMILEHIP_MAGPHASE_NORMALISE 4
#define SMILEHIP_MAGPHASE_POWER 8
#define SMILEHIP_MAGPHASE_DBPSD 16
int smilehip_fftmagphase_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t nfft, int32_t flags, float dbp_norm,
                                float min_dbp, float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
/* R6 with the filter tables given: cMelspec::processVector, forward (src/lldcore/melspec.cpp:519-570) for ANY bank
 * cMelspec::computeFilters builds (:186-451: spectral scales mel / bark / bark_schroed / bark_speex / semitone / log / lin, bwMethod,
 * HFCC and custom-bandwidth banks). dense = 0: d_coef[K] rising-slope weights, d_chanmap[K] band of a bin minus 1 (-3 = unused), the
 * standard bank; dense = 1: d_coef[n_bands x K], d_chanmap[2 n_bands] first / last bin of a band. n_lo / n_hi: nLoF / nHiF (bins).
 * htk_scale: 1, 32767 or 32767^2 (:556-567). The float sums run bin after bin as in the reference. */
int smilehip_melspec_table_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t K, int32_t n_bands, int32_t dense,
                                  const float *d_coef, const int32_t *d_chanmap, int32_t n_lo, int32_t n_hi, int32_t use_power,
                                  float htk_scale, float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
/* R6 inverse: cMelspec::processVector with inverse = 1 (src/lldcore/melspec.cpp:466-516): rows of n_src mel bands -> rows of K spectrum
 * bins ("nBands" of an inverse instance is K, :38,191-194) through the standard bank's tables as cMelspec::computeFilters builds them
 * with the roles swapped (d_coef[K], d_chanmap[K], n_lo / n_hi = nLoF / nHiF). htk_div: 1, 32767 or 32767^2 -- the bands are DIVIDED by
 * it first (:468-479). use_power: the square root of the positive sums, 0 otherwise (:508-514). The HFCC / custom-bandwidth banks are
 * refused by the reference itself in this direction (:487-490). */
int smilehip_melspec_inverse_table_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int32_t n_src, int64_t K,
                                          const float *d_coef, const int32_t *d_chanmap, int32_t n_lo, int32_t n_hi, int32_t use_power,
                                          float htk_div, float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
/* R10, for cPitchACF's voiceQual output: the second result of voicingProb (src/lldcore/pitchACF.cpp:249-283), the zero- or
 * mean-crossing rate of the ACF (rows of n ACF values; fs_sec / max_pitch as in smilehip_pitchacf_frames), one double per row.
 * The HNR outputs (computeHNR / _dB / _lin, :310-361) are three scalar expressions on acf[0] and acf[max_idx]. */
int smilehip_pitchacf_zcr_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t n, int64_t n_frames, double fs_sec,
                                 double max_pitch, double *d_zcr, void *stream);
/* R12, every output of cMZcr::processVector (src/lldcore/mzcr.cpp:108-150). flags: 1 zcr, 2 mcr, 4 amax, 8 maxmin, 16 dc; the row
 * holds the selected values in that order (maxmin: max, min), as floats. */
#define SMILEHIP_MZCR_ZCR 1
#define SMILEHIP_MZCR_MCR 2
#define SMILEHIP_MZCR_AMAX 4
#define SMILEHIP_MZCR_MAXMIN 8
#define SMILEHIP_MZCR_DC 16
int smilehip_mzcr_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N, int64_t n_frames, int32_t flags,
                         float *d_dst, int64_t ld_dst, void *stream);
/* R1: the framing cWinToVecProcessor::myTick + cFramer::doProcess do one frame per tick (src/core/winToVecProcessor.cpp:983
 * getNextMatrix with step / length from setupSequentialMatrixReading, :1037-1052 the row copy; src/dspcore/framer.cpp doProcess),
 * for n_frames frames of ONE channel at once: d_dst[f * ld_dst + i] = d_samples[f * frame_step + i], i < frame_size. d_samples holds
 * (n_frames - 1) * frame_step + frame_size floats. */
int smilehip_frame_rows(smilehip_context *ctx, const float *d_samples, int64_t frame_size, int64_t frame_step, int64_t n_frames,
                        float *d_dst, int64_t ld_dst, void *stream);
/* R2: cVectorPreemphasis::processVector (vectorPreemphasis.cpp:89-107) */
int smilehip_preemphasis_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, float *d_dst,
                                int64_t ld_dst, int64_t n_frames, int64_t N, float k, int de, void *stream);
/* R3: cWindower::processVector (windower.cpp:221-229); window of the plan */
int smilehip_window_frames(smilehip_plan *plan, const float *d_src, int64_t ld_src, float *d_dst,
                           int64_t ld_dst, int64_t n_frames, void *stream);
/* R4: cTransformFFT::processVector forward (transformFft.cpp:165-223):
 * N -> Nfft floats in Ooura's packed layout (fftsg.c:103-135) */
int smilehip_rfft_frames(smilehip_plan *plan, const float *d_src, int64_t ld_src, float *d_dst,
                         int64_t ld_dst, int64_t n_frames, void *stream);
/* R5: cFFTmagphase::processVector magnitude branch (fftmagphase.cpp:215-221) */
int smilehip_fftmag_frames(smilehip_plan *plan, const float *d_src, int64_t ld_src, float *d_dst,
                           int64_t ld_dst, int64_t n_frames, void *stream);
/* R6: cMelspec::processVector (melspec.cpp:519-570) */
int smilehip_melspec_frames(smilehip_plan *plan, const float *d_src, int64_t ld_src, float *d_dst,
                            int64_t ld_dst, int64_t n_frames, void *stream);
/* R7: cMfcc::processVector (mfcc.cpp:239-273) */
int smilehip_mfcc_frames(smilehip_plan *plan, const float *d_src, int64_t ld_src, float *d_dst,
                         int64_t ld_dst, int64_t n_frames, void *stream);
/* cMfcc::processVector with inverse = 1 (src/lldcore/mfcc.cpp:184-235): cepstra back to a mel spectrum -- inverse liftering, the
 * transposed cosine table with c0 halved, sqrt(2 / nBands), exp() when do_log. The plan is one whose n_bands is the component's
 * nBands option (the bands to CREATE) and whose first_mfcc / last_mfcc / cep_lifter / mfcc_htk_compatible are the instance's; the
 * input rows hold last - first + 1 coefficients in the forward component's output order. */
int  smilehip_mfcc_inverse_frames(smilehip_plan *plan, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                  int64_t n_frames, int do_log, void *stream);
/* R12: the accumulation of cEnergy::processVector (energy.cpp:152-161): d_out[f] = sum_n x[n]*x[n] with the
 * float product added to a double; rms / squared / log variants are one host expression on d. */
int smilehip_sumsq_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N, int64_t n_frames,
                          double *d_out, void *stream);
/* R12: the zero-crossing count of cMZcr::processVector (mzcr.cpp:117-124); zcr = count / N */
int smilehip_zcr_count_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N, int64_t n_frames,
                              int32_t *d_out, void *stream);
/* R9: cAcf::processVector, forward path (acf.cpp:249-349) on a plan whose FFT size is 2*(K-1): input K
 * magnitudes per frame, output the first n_out (<= 2*(K-1)) lags. use_power squares the input (:252-259),
 * cepstrum = 1 takes log(x+1) first (:288-305), cepstrum = 2 is oldCompatCepstrum (:275-286: log(x) of the inner bins, DC and
 * Nyquist as they are); cosLifterCepstrum is not covered,
 * norm_output divides by K (:321-325), abs_cepstrum takes |.| of the cepstrum (:327-331; the ACF is
 * always |.|, :343). */
int smilehip_acf_frames(smilehip_plan *plan, const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst, int64_t n_out,
                        int64_t n_frames, int use_power, int cepstrum, int norm_output, int abs_cepstrum, void *stream);
/* R10: the per-frame analysis of cPitchACF::processVector (pitchACF.cpp:137-192, voicingProb :249-284,
 * pitchPeak :286-310). Input per frame: [acf(n) | cepstrum(n)]; output the voicing probability (double) and
 * the index of the cepstral pitch peak (0 = none). F0 = 1/(idx*Tsamp), the voicing cut-off and the causal
 * contour (:189-243) follow in smilehip_pitchacf_contour_step. */
int smilehip_pitchacf_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t n, int64_t n_frames,
                             double fs_sec, double max_pitch, double *d_voicing, int32_t *d_max_idx, void *stream);
/* R10, the rest of cPitchACF::processVector (pitchACF.cpp:185-243) for ONE stream, one frame per call, on the device:
 * F0raw = 1 / (idx * t_samp) in float, the voicing cut-off, one step of the causal F0 contour and the F0 envelope -- the device
 * function the batch chain's lld_pitch_smooth runs over whole utterances (csrc/lld_pitch_contour.hpp). d_voicing / d_max_idx:
 * what smilehip_pitchacf_frames wrote for this frame; t_samp = fsSec / (2 n); d_state: 8 words, zeroed before the stream's
 * first frame, carries the contour across calls; d_out4: F0 (contour), F0raw, F0env, 0. */
int smilehip_pitchacf_contour_step(smilehip_context *ctx, const double *d_voicing, const int32_t *d_max_idx, double t_samp,
                                   double voicing_cutoff, float *d_state, float *d_out4, void *stream);
/* The same for n_frames consecutive frames of the stream in one call (a block tick of the plugin): d_voicing / d_max_idx hold
 * n_frames values, d_out4 n_frames rows of four. */
int smilehip_pitchacf_contour_frames(smilehip_context *ctx, const double *d_voicing, const int32_t *d_max_idx, double t_samp,
                                     double voicing_cutoff, float *d_state, float *d_out4, int64_t n_frames, void *stream);
/* R11: cSpectral::processVector with ComParE_2016's option set (spectral.cpp:586-1560; bands 250-650 and
 * 1000-4000, roll-off .25/.5/.75/.9, flux, centroid, entropy, variance, skewness, kurtosis, slope, sharpness,
 * harmonicity; squareInput = 1, freqRange 0-0, oldSlopeScale = 1): 15 values per frame, in the reference's
 * output order. The frames of ONE stream, in order: d_state (K floats) carries the previous frame's magnitudes
 * across calls (the flux; `first` != 0 marks the stream's first frame, whose flux is 0). Plan: K = 129 / 257 / 513, built
 * with SMILEHIP_STAGE_SPECTRAL (or a ComParE chain plan). */
int smilehip_spectral_frames(smilehip_plan *plan, const float *d_mag, int64_t ld_src, float *d_state, int first,
                             float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
/* R11, every option set of the shipped configuration files: cSpectral::processVector (spectral.cpp:586-1560; options :31-63, output
 * order = the order of the fields below) with squareInput = 1 on a linear magnitude spectrum, freqRange 0-0, normBandEnergies = 0,
 * useLogSpectrum = 0, buggyRollOff = 0, oldSlopeScale = 1. An operator object holds what follows from the options and the frequency axis
 * (frq[i] = i / frame_size_sec, i < K: the axis cTransformFFT attaches, transformFft.cpp:102-117). Frames of ONE stream in order:
 * d_state (K floats; needed when flux is on) carries the last frame's magnitudes from call to call, `first` != 0 marks the stream's
 * first frame (a single 0 for the flux, :1132-1136). One thread per frame, every sum the reference's own sequential chain. */
typedef struct smilehip_spectral_opts {
  int32_t n_bands;              /* bands[]: lo-hi in Hz, <= 16 */
  int32_t band_lo[16], band_hi[16];
  int32_t n_rolloff;            /* rollOff[]: <= 16 */
  double  rolloff[16];
  int32_t flux, centroid, max_pos, min_pos, entropy, variance, skewness, kurtosis, slope, sharpness, harmonicity;
  int32_t flatness, log_flatness;      /* flatness: one output; log_flatness: its logarithm instead (:1515-1545) */
  /* round 6 -- the rest of the linear-spectrum branch (no shipped file uses them): specDiff, specPosDiff, fluxCentroid,
   * fluxAtFluxCentroid (:1124-1254: with flux they share ONE zero on a field's first frame, the values behind it move up and the
   * vector's last slots stay zero, as in the reference), standardDeviation (:1370-1376), slopes[] on the linear axis (:872-985,
   * oldSlopeScale = 1). Output order: bands, slopes, rollOff, specDiff, specPosDiff, flux, fluxCentroid, fluxAtFluxCentroid,
   * centroid, maxPos, minPos, entropy, standardDeviation, variance, skewness, kurtosis, slope, sharpness, harmonicity, flatness. */
  int32_t spec_diff, spec_pos_diff, flux_centroid, flux_at_flux_centroid, standard_deviation;
  int32_t n_slopes;             /* slopes[]: lo-hi in Hz, <= 16 */
  int32_t slope_lo[16], slope_hi[16];
} smilehip_spectral_opts;
typedef struct smilehip_spectral_op smilehip_spectral_op;
int smilehip_spectral_opts_count(const smilehip_spectral_opts *opts);      /* outputs per frame, -1 if out of range */
int smilehip_spectral_op_create(smilehip_context *ctx, const smilehip_spectral_opts *opts, int64_t K, double frame_size_sec,
                                smilehip_spectral_op **op);
int smilehip_spectral_op_n_out(const smilehip_spectral_op *op);
int smilehip_spectral_op_frames(smilehip_spectral_op *op, const float *d_mag, int64_t ld_src, float *d_state, int first,
                                float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
int smilehip_spectral_op_destroy(smilehip_spectral_op *op);
/* R8: cPlp::processVector as auditory spectrum (doAud = 1, doIDFT = doLP = 0; plp.cpp:416-593). d_eql: the
 * equal-loudness weights of the bands (their logs when new_rasta, plp.cpp:335-357), as cPlp::initTables
 * derives them from the input level's band-centre metadata. new_rasta: rasta_coef (host) = {iir, fir[0..4]}
 * (plp.cpp:369-399) and d_state = 4*n_bands + 1 floats, zeroed before a stream's first frame, carries the
 * filter taps and the frame counter across calls. new_rasta == 2: the older RASTA form (:447-466; same coefficients, d_eql = the
 * logs as well): d_state = 6*n_bands + 2 floats (a five-frame ring and the IIR value per band, frame counter, ring position). */
int smilehip_plp_audspec_frames(smilehip_context *ctx, const float *d_mel, int64_t ld_src, int n_bands, const float *d_eql,
                                float melfloor, float compression, int new_rasta, const float *rasta_coef, float *d_state,
                                float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
/* R8: cPlp::processVector producing PLP cepstra (doAud = doIDFT = doLP = doLpToCeps = 1, htkcompatible = 1,
 * firstCC = 0; plp.cpp:499-583): n_bands mel values in, lp_order + 1 cepstra out (c1..c_lpOrder, c0).
 * d_eql: HTK equal-loudness weights of the bands; d_cos: the (lp_order+1) x (n_bands+2) IDFT cosine table and
 * d_sin: the lp_order+1 lifter values of cPlp::initTables (plp.cpp:288-334), all device arrays. */
int smilehip_plp_cc_frames(smilehip_context *ctx, const float *d_mel, int64_t ld_src, int n_bands, const float *d_eql,
                           float melfloor, float compression, int lp_order, const float *d_cos, const float *d_sin,
                           float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
/* R8: cPlp's partial modes in HTK mode (plp.cpp:573-583): the same chain cut short -- out_stage 1: the autocorrelation the IDFT leaves
 * (doIDFT = 1, doLP = 0: lp_order + 1 values per frame), out_stage 2: the LP coefficients of the Durbin recursion (doLP = 1,
 * doLpToCeps = 0: lp_order values). d_cos as for smilehip_plp_cc_frames. */
int smilehip_plp_stage_frames(smilehip_context *ctx, const float *d_mel, int64_t ld_src, int n_bands, const float *d_eql,
                              float melfloor, float compression, int lp_order, const float *d_cos, int out_stage,
                              float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
/* R13: cDeltaRegression::processBuffer (kind 0, deltaRegression.cpp:144-152, norm = 2*sum i^2) and
 * cContourSmoother::processBuffer (kind 1, contourSmoother.cpp:106-114, smaWin = 2W+1) on one row of a
 * cWindowProcessor block: d_x points at sample 0 of the row and is valid on [-W, n_t + W). */
int smilehip_window_op_row(smilehip_context *ctx, const float *d_x, float *d_y, int64_t n_t, int kind, int W, void *stream);
/* The option variants of the same two components, one row: kind 2 = cContourSmoother with noZeroSma (contourSmoother.cpp:91-104);
 * kind 3 = cDeltaRegression with onlyInSegments (deltaRegression.cpp:121-137). The reference's onlyInSegments branch adds i^2 to
 * its `norm` member for every pair it uses and never resets it: d_norm_io (one device float, initialised by the caller to
 * 2 * sum i^2 -- deltaRegression.cpp:77-79) is that member, read and written by every call in the instance's processing order.
 * kinds 0 / 1 forward to smilehip_window_op_row. */
int smilehip_window_op_row_ex(smilehip_context *ctx, const float *d_x, float *d_y, int64_t n_t, int kind, int W, float *d_norm_io,
                              void *stream);
/* R13, every option of cDeltaRegression::processBuffer (src/dspcore/deltaRegression.cpp:104-170) on one row: deltawin W >= 0 (0: the
 * simple difference x[n] - x[n-1], :141-153; the row is then valid on [-1, n_t)), flags SMILEHIP_DELTA_RELATIVE (relativeDelta:
 * delta / |prior|, 0 where prior is 0, :104-111), _HALFWAVE (halfWaveRect), _ABS (absOutput; halfWaveRect wins, :158-166),
 * _SEGMENTS (onlyInSegments, with d_norm_io as in smilehip_window_op_row_ex). */
#define SMILEHIP_DELTA_RELATIVE 1
#define SMILEHIP_DELTA_HALFWAVE 2
#define SMILEHIP_DELTA_ABS 4
#define SMILEHIP_DELTA_SEGMENTS 8
int smilehip_delta_op_row(smilehip_context *ctx, const float *d_x, float *d_y, int64_t n_t, int W, int flags, float *d_norm_io,
                          void *stream);
/* R13 on a whole cWindowProcessor block at once (src/core/windowProcessor.cpp:171-236 walks the block's element rows one after the
 * other; every output depends on its own window only, so the rows and frames of a block are independent): the matrices are
 * FRAME-major as the data memory holds them (cMatrix::data[el + t * N]). d_x points at frame 0 of the block and is valid on frames
 * [-max(W,1), n_t + W); d_y gets n_t frames. op 0: cDeltaRegression (deltaRegression.cpp:104-170) with delta_flags
 * SMILEHIP_DELTA_RELATIVE | _HALFWAVE | _ABS (not _SEGMENTS: its divisor is carried from value to value in processing order, a
 * block has no such order) and W >= 0; op 1: cContourSmoother (contourSmoother.cpp:106-114), op 2: with noZeroSma (:91-104), W >= 1.
 * Same expressions as the row operators above: same bits. */
int smilehip_window_op_block(smilehip_context *ctx, const float *d_x, int64_t ld_x, float *d_y, int64_t ld_y, int64_t n_t,
                             int32_t n_cols, int op, int W, int delta_flags, void *stream);
/* cDeltaRegression with onlyInSegments (deltaRegression.cpp:121-137) on the same kind of block: n_ticks ticks of tick_frames frames
 * (the component's blocksize) each. Its divisor grows with every pair of values it uses, in the order the reference's ticks visit
 * them (tick, then element, then the tick's frames): the block is walked in that order by one thread; d_norm_io as in
 * smilehip_window_op_row_ex. */
int smilehip_delta_segments_block(smilehip_context *ctx, const float *d_x, int64_t ld_x, float *d_y, int64_t ld_y, int64_t n_ticks,
                                  int32_t tick_frames, int32_t n_cols, int W, int delta_flags, float *d_norm_io, void *stream);
/* ---- GeMAPS / eGeMAPS components, per component, on an eGeMAPS plan (smilehip_config_egemapsv02: 16 kHz, 20 ms Hamming frames
 * -> 512-point spectrum; 60 ms frames -> 1024-point spectrum). Rows in, rows out, like the operators above. */
/* cSpectral::processVector (src/lldcore/spectral.cpp:586-1254) with the GeMAPS option sets -- squareInput = 1, useLogSpectrum = 1,
 * specFloor 1e-7, freqRange 0-5000: [gemapsv01b_logSpectral]'s four outputs in its own order (logSpectralSlopeOfBand0-500,
 * ..500-1500, alphaRatioDB, hammarbergIndexDB) followed by [egemapsv02_logSpectral_flux]'s spectralFlux: 5 values per frame of
 * K = 257 magnitudes. The frames of ONE stream in order; d_state (K floats) carries the previous frame's magnitudes across
 * calls, `first` != 0 marks the stream's first frame (flux 0). */
int smilehip_spectral_gemaps_frames(smilehip_plan *plan, const float *d_mag, int64_t ld_src, float *d_state, int first,
                                    float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
/* cSpecResample::processVector (src/dsp/specResample.cpp:175-185, smileDsp_irdft src/smileutil/smileUtil.c:1800-1820) for
 * [gemapsv01b_resampLpc] (targetFs 11000): the Nfft-value complex spectrum of a 20 ms frame (Ooura packing, level
 * gemapsv01b_fftcH25) -> 220 samples at 11 kHz; every output's float sum in the reference's order. */
int smilehip_specresample_frames(smilehip_plan *plan, const float *d_spec, int64_t ld_src, float *d_dst, int64_t ld_dst,
                                 int64_t n_frames, void *stream);
/* cLpc::processVector (src/lld/lpc.cpp:171-213) with method = acf, p = 11, saveLPCoeff only: smileDsp_autoCorr + Durbin
 * (smileUtil.c:1560-1630) on 220 samples -> 11 coefficients */
int smilehip_lpc_frames(smilehip_plan *plan, const float *d_x, int64_t ld_src, float *d_lpc, int64_t ld_dst, int64_t n_frames,
                        void *stream);
/* ---- the components the other INTERSPEECH sets of config/is09-13 add (IS10_paraling, IS11_speaker_state, IS12_speaker_trait) ----
 * cSpecResample for ANY geometry. smilehip_specresample_geometry: cSpecResample::setupNewNames (src/dsp/specResample.cpp:117-172) for
 * n_in packed spectrum values of a level with frameSizeSec fs_sec, lastFrameSizeSec last_fs_sec (the frame before zero padding),
 * basePeriod base_period and the option targetFs -> output samples, kMax (antiAlias = 1) and the tables' denominator nd.
 * smilehip_specresample_tables: smileDsp_initIrdft (src/smileutil/smileUtil.c:1752-1786), k_max / 2 * n_out floats each (host).
 * smilehip_specresample_table_frames: smileDsp_irdft (:1800-1820) with those tables on the device. */
int smilehip_specresample_geometry(int64_t n_in, double fs_sec, double last_fs_sec, double base_period, double target_fs,
                                   int64_t *n_out, int64_t *k_max, double *nd);
int smilehip_specresample_tables(int64_t n_in, int64_t n_out, int64_t k_max, double nd, float *cos_table, float *sin_table);
int smilehip_specresample_table_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t n_in, int64_t n_out,
                                       int64_t k_max, const float *d_cos, const float *d_sin, float *d_dst, int64_t ld_dst,
                                       int64_t n_frames, void *stream);
/* cLpc::processVector (src/lld/lpc.cpp:171-213) with method = acf, saveLPCoeff only, any frame length n and order p <= 32:
 * smileDsp_autoCorr + smileDsp_calcLpcAcf (smileUtil.c:1560-1630) */
int smilehip_lpc_acf_frames(smilehip_context *ctx, const float *d_x, int64_t ld_src, int64_t n, int32_t p, float *d_lpc,
                            int64_t ld_dst, int64_t n_frames, void *stream);
/* cLsp::processVector (src/lld/lsp.cpp:289-312): p LP coefficients -> p line spectral frequencies (the Speex-derived lpc_to_lsp,
 * :144-269, with the C library's acosf) */
int smilehip_lsp_frames(smilehip_context *ctx, const float *d_lpc, int64_t ld_src, int32_t p, float *d_dst, int64_t ld_dst,
                        int64_t n_frames, void *stream);
/* cIntensity::processVector (src/lldcore/intensity.cpp:125-145) on rows of N samples. flags: 1 intensity, 2 loudness, outputs in
 * that order. As in the reference the Hamming-weighted sum runs over MIN(N, number of outputs) samples (:134). */
int smilehip_intensity_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N, int32_t flags, float *d_dst,
                              int64_t ld_dst, int64_t n_frames, void *stream);
/* cVectorOperation::processVector, the element-wise operations (src/other/vectorOperation.cpp:360-435, 508-527): add / mul (param1),
 * log, lgA (base param1), sqr, ee, abs, dBp, dBv; logfloor <= 0 selects the component's default 1e-12 -- and the vector-to-scalar
 * operations sum, ssm, ll1, ll2 (:461-490: one float accumulation over the row, ONE output value per row). */
enum { SMILEHIP_VOP_ADD = 0, SMILEHIP_VOP_MUL, SMILEHIP_VOP_LOG, SMILEHIP_VOP_LOGA, SMILEHIP_VOP_SQRT, SMILEHIP_VOP_E, SMILEHIP_VOP_ABS,
       SMILEHIP_VOP_DB_POW, SMILEHIP_VOP_DB_MAG, SMILEHIP_VOP_X_SUM, SMILEHIP_VOP_X_SUMSQ, SMILEHIP_VOP_X_L1, SMILEHIP_VOP_X_L2 };
int smilehip_vecop_frames(smilehip_context *ctx, int32_t op, float param1, float logfloor, const float *d_src, int64_t ld_src,
                          int32_t n_cols, float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);
/* cPitchSmoother::processVector (src/lldcore/pitchSmoother.cpp:236-425) over the frames of n_streams streams: medianFilter0 = 0,
 * postSmoothingMethod none (post_simple = 0) or simple (1), octaveCorrection as given. Rows of d_src = [F0Cand | candVoicing |
 * candScore] (3 n_cand values: cPitchShs' output without its nCandidates element); stream u = rows [d_row_off[u], d_row_off[u+1])
 * (d_row_off = NULL: one stream of n_rows_single rows). flags: 1 F0final, 2 F0finEnv, 4 voicingFinalClipped, 8 voicingFinalUnclipped
 * (output order). The rows the component writes go to d_dst from row d_row_off[u] on -- with simple post smoothing one fewer than it
 * reads (the first frame yields none, :331); d_written[u] (optional) = that count. d_state (optional, 32 bytes per stream): the
 * carried state, read when resume != 0 and written back (frame-by-frame callers). */
int smilehip_pitch_smoother_rows(smilehip_context *ctx, int32_t n_cand, float voicing_cutoff, int32_t octave_correction,
                                 int32_t post_simple, int32_t flags, const float *d_src, int64_t ld_src, const int64_t *d_row_off,
                                 int32_t n_streams, int64_t n_rows_single, void *d_state, int32_t resume, float *d_dst, int64_t ld_dst,
                                 int64_t *d_written, void *stream);
/* cFormantLpc::processVector (src/lld/formantLpc.cpp:192-290): nFormants = 5, saveFormants = saveBandwidths = 1, minF 50,
 * maxF 5450, no median filter / octave correction; roots of the LP polynomial by the reference's balanced companion-matrix
 * QR iteration (src/smileutil/zerosolve.cpp): 11 coefficients -> [5 frequencies | 5 bandwidths] */
int smilehip_formantlpc_frames(smilehip_plan *plan, const float *d_lpc, int64_t ld_src, float *d_dst, int64_t ld_dst,
                               int64_t n_frames, void *stream);
/* cHarmonics::processVector (src/lld/harmonics.cpp:743-1031) with [gemapsv01b_harmonics]'s options: per row F0 (Hz; 0 =
 * unvoiced), the 10 values of level gemapsv01b_formants and the 513 magnitudes of the 60 ms frame -> 6 values
 * [HarmonicsToNoiseRatioACFLogdB, HarmonicDifferenceLogRelH1-H2, ..H1-A3, FormantAmplitudeByMaxHarmonicLogRelF0[1..3]] */
int smilehip_harmonics_frames(smilehip_plan *plan, const float *d_f0, const float *d_formants, int64_t ld_formants,
                              const float *d_mag, int64_t ld_mag, float *d_dst, int64_t ld_dst, int64_t n_frames, void *stream);

/* R13: n_orders chained cDeltaRegression::processBuffer (deltaRegression.cpp:
 * 113-170) with the reference's end-of-input semantics, per utterance of the
 * batch; reads columns [0,D) of d_io rows, writes columns [D, D*(1+n_orders)). */
int smilehip_delta_chain(smilehip_plan *plan, smilehip_batch *batch, float *d_io, int64_t ld,
                         int32_t D, int32_t W, int32_t n_orders, void *stream);

/* cValbasedSelector::myTick (src/other/valbasedSelector.cpp:139-247) for a block of frames, fixed threshold (adaptiveThreshold = 0):
 * element min(idx, N-1) of a frame against `threshold` -- val > threshold (invert: <; allow_equal: also ==) -- decides:
 * d_keep[t] = 1: the frame is handed on (d_dst row = the frame, without element idx if remove_idx); 2: zeroVec, the row is
 * output_val everywhere; 0: the frame is dropped (no row is to be written). d_dst rows have N (remove_idx: N-1) values. */
int smilehip_valbased_select_frames(smilehip_context *ctx, const float *d_src, int64_t ld_src, int64_t N, int64_t n_frames,
                                    int32_t idx, float threshold, int32_t invert, int32_t allow_equal, int32_t zero_vec,
                                    int32_t remove_idx, float output_val, float *d_dst, int64_t ld_dst, int32_t *d_keep, void *stream);

/* ---- cPitchSmootherViterbi as a stream (src/lld/pitchSmootherViterbi.cpp:451-564: myTick pushes one frame of candidates
 * into cSmileViterbi::addFrame, :80-216, and writes every frame that has become decided; the rest at end of input through
 * flushTrellis). What the plugin's tick-level override binds: one utterance, one frame per call, the trellis (7 states =
 * 6 candidates + unvoiced, `buffer_len` frames of paths) lives on the device between the calls. weights6 = wLocal, wTvv,
 * wTvvd, wTvuv, wThr, wRange as cSmileViterbiPitchSmooth uses them (note its setWeights stores wTvv into wTvvd,
 * pitchSmootherViterbi.hpp:291-299 -- pass the effective values). Host pointers throughout.
 * push / flush return the frames that became decided by this call, in order: frames[i] = frame index, states[i] = the state
 * chosen (0..5 = candidate, 6 = unvoiced); F0 / voicing of a decided frame are the caller's own candidate values. */
typedef struct smilehip_viterbi_stream smilehip_viterbi_stream;
int smilehip_viterbi_stream_create(smilehip_context *ctx, int32_t buffer_len, float voicing_cutoff, const double *weights6,
                                   smilehip_viterbi_stream **out);
/* nCandidates of the pitch detector feeding the smoother (default 6; 1 .. 6: states = candidates + "unvoiced"). Before the first push. */
int smilehip_viterbi_stream_set_candidates(smilehip_viterbi_stream *s, int32_t n_candidates);
int smilehip_viterbi_stream_push(smilehip_viterbi_stream *s, const float *cand_f0, const float *cand_voicing,
                                 int32_t *n_decided, int32_t *frames, int32_t *states, int32_t cap);
/* n_frames consecutive frames in one call (a block tick of the plugin): cand_f0 / cand_voicing are n_frames rows of `ld` floats
 * (the first n_candidates of a row are read); the decisions of all the steps come back in the order the single pushes would have
 * reported them. */
int smilehip_viterbi_stream_push_frames(smilehip_viterbi_stream *s, const float *cand_f0, const float *cand_voicing, int64_t ld,
                                        int32_t n_frames, int32_t *n_decided, int32_t *frames, int32_t *states, int32_t cap);
int smilehip_viterbi_stream_flush(smilehip_viterbi_stream *s, int32_t *n_decided, int32_t *frames, int32_t *states, int32_t cap);
int smilehip_viterbi_stream_destroy(smilehip_viterbi_stream *s);

/* ---- cPitchJitter as a stream (src/lld/pitchJitter.cpp:591-1084, myTick: one F0 frame per tick, waveform matching around the
 * pitch periods in the wave samples [lastIdx, lastIdx + toRead), with the read position, the left-over samples and the last
 * period / difference / jitter / shimmer values carried from frame to frame). What the plugin's tick-level override binds: the
 * carried state lives on the device; the caller reads the wave samples the reference would read (its own prologue, :604-668, from
 * last_idx / last_mis returned by the previous push) and hands them over as 16-bit PCM with their absolute start index.
 * n_pcm = 0: the samples could not be read (the reference's NULL matrix): the read position still advances, out5 is not valid.
 * out5 = jitterLocal, jitterDDP, shimmerLocal, logHNR, shimmerLocalDB of this frame. sample_period = 1 / sample rate;
 * frame_size / frame_step: the F0 frames in samples; frame_step_sec: their period as the level reports it. */
typedef struct smilehip_jitter_stream smilehip_jitter_stream;
int smilehip_jitter_stream_create(smilehip_context *ctx, double sample_period, int64_t frame_size, int64_t frame_step,
                                  double frame_step_sec, double search_range_rel, int32_t broken_jitter_thresh,
                                  smilehip_jitter_stream **out);
/* The first F0 frame's time stamp in frames (tmeta->time / frame_step_sec): 0 behind cPitchSmootherViterbi, 1 behind cPitchSmoother
 * with simple post smoothing (its first output carries the second frame's time meta data). Before the first push. */
int smilehip_jitter_stream_set_time_offset(smilehip_jitter_stream *s, int64_t frames);
int smilehip_jitter_stream_push(smilehip_jitter_stream *s, float f0, const int16_t *h_pcm, int64_t pcm_start, int64_t n_pcm,
                                float *out5, int64_t *last_idx, int64_t *last_mis);
/* n_frames consecutive F0 frames in one call (a block tick of the plugin). h_pcm / pcm_start / n_pcm: the samples that arrived since
 * the last call (every sample of the stream is kept on the device); pcm_start + n_pcm is what exists of the stream now -- a frame
 * whose periods reach beyond it is the reference's "no matrix" case (pitchJitter.cpp:660-665). out5: n_frames rows of five. */
int smilehip_jitter_stream_push_frames(smilehip_jitter_stream *s, const float *f0, int32_t n_frames, const int16_t *h_pcm,
                                       int64_t pcm_start, int64_t n_pcm, float *out5, int64_t *last_idx, int64_t *last_mis);
int smilehip_jitter_stream_destroy(smilehip_jitter_stream *s);

#ifdef __cplusplus
}
#endif
#endif /* SMILEHIP_H */
