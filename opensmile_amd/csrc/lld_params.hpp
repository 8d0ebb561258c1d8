// Kernel parameter block shared by the host API (smilehip_*.cpp) and the device
// code (lld_kernels.hip). Plain data, passed by value at launch.
#pragma once
#include "lld_device.hpp"
#include "lld_ooura.hpp"
#include "lld_ooura_wave.hpp"
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/smilehip.h"

namespace smilehip {

// One tile of the fast kernel: up to fast512_tile_frames() consecutive frames of one utterance,
// everything a wave needs in one 24-byte scalar load (no dependent loads through the utterance tables).
struct TileRec {
  int64_t samp0;      // absolute index (in the packed PCM) of the first frame's first sample
  int64_t row0;       // output row of the first frame
  int32_t n_frames;   // frames of the tile
  int32_t pad;
};

// A tile of the fast kernel's delta-fused form (lld_mfcc512<..., DELTA = true>): passes of four frames from frame p0 of its
// utterance on. The tile's own frames are [t0, t1); p0 = t0 - 4 inside an utterance (one pass ahead: the regression of the tile's
// first frames reaches four frames back), and one pass follows the tile's last frame (the regression of its last frames
// reaches four frames ahead; at the utterance's end the same pass drains the two regression stages).
struct FTileRec {
  int64_t samp0;       // absolute index (in the packed PCM) of frame p0's first sample
  int64_t row0;        // output row of frame p0
  int32_t n_frames;    // 4 x passes
  int32_t live_n;      // frames p0 + r with r >= live_n lie behind the utterance's last frame (T - p0)
  int32_t e0, e1;      // the rows of frames p0 + r, e0 <= r < e1, are this tile's to write
  int32_t lo;          // -p0: the utterance's first frame as a relative index (the lower index clamp)
  int32_t delta_on;    // 0: static coefficients only (utterances of <= short_T frames: lld_chain_short finishes them)
};

struct LldParams {
  // batch
  const int16_t *pcm;          // packed utterances
  const float *pcm_f32;        // the same samples as floats (R0 done by smilehip_pcm_convert), or nullptr: then pcm is read
  int64_t pcm_total;           // samples in the packed buffer (= samp_off[n_utt])
  const int64_t *samp_off;     // [n_utt+1] sample offsets (device)
  const int64_t *frame_off;    // [n_utt+1] output row offsets (device)
  const int32_t *frame_utt;    // optional [total_frames]: the utterance of a frame (else: binary search in frame_off)
  const int32_t *tile_utt;     // [n_tiles] utterance of tile
  const int32_t *tile_t0;      // [n_tiles] first frame of tile (within utterance)
  const TileRec *tile_rec;     // [n_tiles] the same tiles, resolved (fast kernel)
  int32_t n_utt;
  int32_t n_tiles;
  const FTileRec *ftile_rec;   // [n_ftiles] tiles of the delta-fused fast kernel (null: not built for this batch)
  int32_t n_ftiles;
  int64_t total_frames;
  float *out;                  // [total_frames x ld_out]
  int64_t ld_out;
  // geometry
  int32_t N, H, Nfft, K, pad_left;
  // R2 / R3
  int32_t preemph, de;
  float k, one_minus_k, win_offset;
  const float *window;         // [N]
  // R4: complex FFT of length M = Nfft/2 + real untangle
  const float2 *tw_half;       // [M/2]  e^{-2 pi i j / M}
  const float2 *tw_full;       // [M/2+1] e^{-2 pi i k / Nfft}
  OouraTab oo;                 // reference-order transform (lld_ooura.hpp); oo.tw == nullptr: the radix-2 order above
  // R6
  const float *mel_coef;       // [K]
  const int32_t *mel_rng;      // [4*n_bands] rise_lo, rise_hi, fall_lo, fall_hi per band
  float mel_scale;
  int32_t use_power, n_bands;
  // R7
  const float *dct_rows;       // [n_mfcc x n_bands], row = output position
  const float *dct_gain;       // [n_mfcc]
  int32_t n_mfcc;
  float melfloor, log_floor;
  // R8 PLP-CC instead of R7 (SMILEHIP_CHAIN_PLP): cPlp after the mel bank
  int32_t plp;                 // 0 = cMfcc, 1 = cPlp (n_mfcc = lp_order + 1 outputs)
  int32_t plp_order;
  float plp_compression;
  const float *plp_eql;        // [n_bands] HTK equal-loudness weights at the band centres
  const float *plp_cos;        // [(lp_order+1) x (n_bands+2)] IDFT cosine table
  const float *plp_sin;        // [lp_order+1] lifter table
};

// the batch's samples as the kernels read them (lld_device.hpp: PcmIn)
__device__ __forceinline__ PcmIn pcm_in(const LldParams &P) { PcmIn r; r.s = P.pcm; r.f = P.pcm_f32; return r; }

// Device-side tables of the fast Nfft=512 kernel (lld_mfcc512.hip)
struct Fast512Tables {
  const float2 *tw256;        // [256] w256^(j*k1), index k1*16+j
  const float2 *tw512;        // [256] e^{-2 pi i k/512}
  const float2 *win;          // [MP*16] window pairs (x 1/32767), index m*16+j
  const float4 *melw;         // [U*16*2] weights of unit (step, lane): two float4
  const uint32_t *melo;       // [U*16] byte offset of the unit's octet in the frame's power buffer
  const float *dct28;         // [16 x 28] DCT rows padded to 28
  const int32_t *lane_bands;  // [16] per lane: bit i = unit i belongs to the first band | first band << 8 | second band << 16
  int32_t mel_units;          // units per lane
  float mel_scale;
  const float *plp_eql;       // PLP chain: [32] equal-loudness weights (dct28 then holds the IDFT cosine rows)
  const float *plp_sin;       // [16] lifter table
};

// Extra parameters of the IS09 LLD frame kernel (lld_is09.hip)
struct Is09Params {
  float *raw16;              // [total_frames x 16] pre-smoothing LLD columns
  float fsSec;               // (float) frameSizeSec of the ACF level (pitchACF.cpp:113-116)
  double maxPitch;
  double voicingCutoff;
};

// Parameters of the ComParE A+B kernels (lld_compare.hip)
struct CompareParams {
  const int32_t *run_utt;    // [n_runs] runs of run_frames consecutive 20 ms frames: utterance
  const int32_t *run_t0;     // [n_runs] first frame
  int32_t run_frames;        // frames per run (0: 8). A run pays one extra transform (the flux needs the frame before it): the batch picks the longest runs that still fill the device
  float *rawA;               // [total_frames20 x 4]  audspecSum, audspecRastaSum, rms, zcr (zcr valid for t < T60)
  float *rawB;               // [total_frames20 x 55] audSpec_Rfilt[26], spectral[15], mfcc[14]
  float *mel1;               // [total_frames20 x 26] un-filtered mel power spectrum (input of the RASTA pass)
  const float *eql;          // [26] equal-loudness weights (linear), plp.cpp:335-357
  const float *eql_log;      // [26] their logs (RASTA instance: doLog forced)
  const double *sharp_w;     // [K-1] sharpness weights bark * g(bark) per bin 1..K-1
  float plp_melfloor, compression, rasta_iir;
  float rasta_fir[5];
  double fsSec;              // frameSizeSec of the magnitude level
  int32_t N60;               // samples of a 60 ms frame
  int64_t max_utt_samples;   // the batch's longest utterance (the quad form keeps 32-bit row state: launch_compare)
  // band-energy edges (spectral.cpp:779-853), resolved on the host: first/last bin and their weights
  int32_t band_iL[2], band_iR[2];
  double band_wL[2], band_wR[2];
  double slope_Sf, slope_S2f;   // sums of f and f^2 over bins 1..K-1 (spectral.cpp:1399-1427)
};

// ComParE / GeMAPS F0 group (lld_f0.hip): 60 ms frames -> cSpecScale -> cPitchShs -> Viterbi -> selector
struct F0Params {
  int32_t N, H, Nfft, K, pad_left;  // 60 ms framing @ 16 kHz: 960, 160, 1024, 513, 32
  const float *window;              // [N]
  const float2 *tw_half;            // [M/2], M = Nfft/2
  const float2 *tw_full;            // [M/2+1]
  OouraTab oo;                      // reference-order transform (lld_ooura.hpp); oo.tw == nullptr: the radix-2 order above
  // cSpecScale: natural cubic spline over the octave-scaled bin positions (smileUtilSpline.c:139-212); the
  // decomposition part of the tridiagonal sweep does not depend on the data and is precomputed:
  const double *sp_rec;             // [K x 4] per bin i: sigma_i, p_i = 1/(sigma_i*dec_{i-1}+2), dec_i = (sigma_i-1)*p_i, 0
  const double *sp_d1, *sp_d2;      // [K] (x[i+1]-x[i])(x[i+1]-x[i-1]), (x[i]-x[i-1])(x[i+1]-x[i-1])
  const int32_t *ip_k;              // [K] lower source bin of target point i (smileMath_csplint_init, :296-342)
  const double *ip_co;              // [K x 3] a, c, d of target point i
  const double *audw;               // [K] auditory weighting (specScale.cpp:279-287)
  const double *ip_rec;             // [K x 4] a, c, d, audw of target point i as one record (lld_f0_sweep's scalar loads)
  const int32_t *ip_cnt;            // [ceil(K / 16) x 16] target points whose lower source bin is k (zero beyond K - 2)
  const double *sw_rec;             // [K x 8] per bin: sigma, p, dec, d1, RN(1 / d1), d2, RN(1 / d2), 0 (lld_f0_sweep: one record)
  // cPitchShs
  int32_t n_harm;
  int32_t shift[16];                // shift[i-2] for harmonic i = 2..n_harm
  float scale[16];                  // compressionFactor^(i-1)
  float Fmint, Fstept;
  double log_base;
  double min_pitch, max_pitch;
  float voicing_cutoff;
  float min_energy;
  double jit_Tw;                    // cPitchJitter: sample period of the wave level, 1.0 / sampleRate
  int32_t jit_broken_thresh;        // useBrokenJitterThresh (pitchJitter.cpp:801-809)
  double jit_step_sec;              // period of the F0 level (frameStep)
  double vit_w[6];                  // cPitchSmootherViterbi: wLocal, wTvv, wTvvd, wTvuv, wThr, wRange (wTuu is never used)
  // per-frame results between the kernels
  float *shs;                       // [total_frames x 21] nCand | F0Cand[6] | candVoicing[6] | candScores[6] | F0raw | voicingClip
  float *e60;                       // [total_frames] RMS energy of the windowed frame
  float *hps_tap;                   // optional [total_frames x K] level is13_hpsG60, or null
  float *mag_keep;                  // optional [total_frames x mag_ld]: the magnitude spectra (level fftmagG60) for the readers behind the
  int64_t mag_ld;                   //   pitch decision (cHarmonics reads the same level: lld_gemaps_harm), or null
  // per-component operators (mode 1: cSpecScale rows -> hps_tap, mode 2: cPitchShs rows -> shs)
  int32_t mode;
  double *ab;                       // chain mode, between the three frame kernels (lld_f0.hip, F0Scratch): the enhanced magnitudes
                                    // and the octave-scale spectrum as floats in blocks [64-frame tile][16-bin block][frame (64)][bin in
                                    // block (16)], the sweep's checkpoints [tile][block][frame], the frames' sums of squares
  int64_t ab_rows;                  // rows the scratch holds (a multiple of 64)
  int32_t tile0, n_tiles_chunk;     // the tiles [tile0, tile0 + n_tiles_chunk) this launch works on
  int64_t n_rows;
  const float *in_rows;
  int64_t ld_in, ld_tap, ld_shs;
  int32_t *pending;                 // optional [n_utt]: frames the Viterbi pass had not decided at the end of input
  int32_t vit_buf;                  // cPitchSmootherViterbi bufferLength (30 ComParE, 40 GeMAPS; <= 128)
  int32_t vit_log_out;              // 1: rows [F0final, F0finalLog, voicingFinalUnclipped] (GeMAPS), 0: [F0final, voicing]
  double jit_search_range;          // cPitchJitter searchRangeRel (0.25 ComParE, 0.1 GeMAPS)
  int32_t n_cand;                   // cPitchShs nCandidates (1 .. 6; the rows keep six slots, the unused ones are zero)
  int32_t old_peaks;                // cPitchShs greedyPeakAlgo = 0: only peaks above every earlier peak become candidates (pitchShs.cpp:286-302)
  int64_t jit_t_shift;              // frames: F0 frame t carries the time stamp of frame t + shift (1 behind cPitchSmoother, which delays its
                                    // values by one frame but hands on the time meta data of the frame it was called with; 0 otherwise)
  float *jit_shim_db;               // optional [total_frames]: shimmerLocalDB = 20 log10(shimmerLocal + 1)
  double *jit_stream;               // stream mode (one frame per launch, smilehip_jitter_stream_*): [0] lastIdx [1] lastMis [2] next frame
                                    // [3] lastT0 [4] lastDiff [5] lastJitterLocal [6] lastJitterDDP [7] lastShimmerLocal; null = whole utterances
  const int32_t *jit_item_utt, *jit_item_t0;   // lld_jitter_runs' work items (utterance, first frame) of 64 consecutive frames each, ordered
  int32_t n_jit_items;              //   by first frame, then utterance (smilehip_batch: d_jit_utt / d_jit_t0); null = one workgroup per utterance
  int32_t *jit_redo;                // [n_utt], zero between runs: utterances lld_jitter_runs hands back to the per-utterance kernel
  int32_t scale_off;                // cSpecScale: 1 = no peak enhancement, 2 = no smoothing, 4 = no auditory weighting (per-component operator; the chains run with 0)
  int32_t jit_cap;                  // samples of wave per frame the kernels hold in LDS (jitter_wave_capacity; 0 = the general capacity)
  int32_t *jit_ctl;                 // [2], zero between runs: lld_jitter_runs' item counter and count of finished workgroups
};

// eGeMAPSv02 / GeMAPSv01b LLD level (lld_gemaps.hip): per-frame scratch and constants
struct GemapsParams {
  // ---- 20 ms chain ----
  const int32_t *run_utt, *run_t0;  // runs of run_frames consecutive 20 ms frames (the flux needs the previous frame)
  int32_t run_frames;               // frames per run (0: 8), see CompareParams
  float *raw20;                     // [total_frames20 x 12] loudness | slope0-500, slope500-1500, alphaRatio, hammarberg | flux |
                                    //                        mfcc1..4 | energy2 | 0
  float *spec220;                   // [total_frames20 x 220] what cSpecResample reads of the complex spectrum: (Re, Im Ooura) of
                                    //                        bins 1..109, then the DC value, one pad
  const float *eql;                 // [26] equal-loudness weights (plp.cpp:335-357)
  float plp_melfloor, compression;
  double fsSec;                     // frameSizeSec of the 20 ms spectrum level
  int32_t sl_iL[2], sl_iR[2];       // cSpectral slopes 0-500 / 500-1500: edge bins, weights, Nind (spectral.cpp:872-946)
  double sl_wL[2], sl_wR[2], sl_Nind[2];
  int32_t rng_lo, rng_hi;           // freqRange 0-5000 in bins (flux)
  int32_t ar_n1, ar_n2;             // alpha ratio: bins [0, n1) have f < 1000 Hz, bins [n1, n2) 1000 <= f <= 5000 Hz (f = i / fsSec)
  float spec_floor, log_spec_floor, log_spec_factor;
  // ---- cSpecResample -> cLpc -> cFormantLpc ----
  const float *rs_cos, *rs_sin;     // [109 x 220] smileDsp_initIrdft's tables, transposed: [k/2 - 1][i]
  float rs_norm;                    // smileDsp_irdft's divisor: (FLOAT_DMEM)(K / 2), K = the Nfft values of the complex spectrum (256 at 16 kHz)
  float *lpc;                       // [total_frames20 x 12] 11 LP coefficients + pad
  float *formants;                  // [total_frames20 x 10] 5 frequencies | 5 bandwidths
  int64_t total_frames20;
  double fm_T, fm_min, fm_max;      // sample period of the resampled signal, formant search range
  // ---- 60 ms chain ----
  double fsSec60;                   // frameSizeSec of the 60 ms spectrum level (frequency axis of cHarmonics)
  const float *pitch3;              // [total_frames60 x 3] F0final, F0finalLog, voicing (after the energy gate)
  const float *jit4;                // [total_frames60 x 4] cPitchJitter's outputs, jitterLocal in column 0
  const float *shim_db;             // [total_frames60] shimmerLocalDB
  float *harm6;                     // [total_frames60 x 6] cHarmonics' outputs
  const float *mag60;               // [total_frames60 x mag60_ld] the 60 ms magnitude spectra as lld_f0_spec kept them (F0Params::mag_keep), or
  int64_t mag60_ld;                 //   null: lld_gemaps_harm transforms the frames again
  const int64_t *frame_off60;       // [n_utt+1]
  const TileRec *tile60;            // tiles of <= 8 consecutive 60 ms frames
  int32_t n_tiles60;
  const int32_t *pending;           // [n_utt] frames the Viterbi pass had not decided at the end of input
  int32_t *harm_ctl;                // [2], zero between runs: lld_gemaps_harm's tile counter and count of finished waves (null: static tiles)
  // ---- smoothed levels ----
  float *func_in;                   // [fin_off[n_utt] x 36], T20+1 rows per utterance that has a 60 ms frame:
                                    //   loudness_sma3, flux_sma3, mfcc1..4_sma3 (6) | F0semitone_sma3nz (1) | lldSetNoF0AndLoudnessNz_smo (14) |
                                    //   lldSetSpectralNz_smo (9) | lldSetSpectralZ_smo (5) | energy2 of the raw 20 ms frame (T20 rows)
  const int64_t *fin_off;           // [n_utt+1] row offsets of func_in
  int32_t *pending_j;               // [n_utt] P if P < T60 else 0: rows the jitter-gated functionals leave out
  // ---- per-component operators (the plugin's overrides, stage-level tests): rows instead of the batch scratch
  int32_t op_mode;                  // lld_gemaps_lpc: 0 fused (spec220 -> lpc), 1 cSpecResample only (Ooura-packed spectra -> 220
                                    //   samples), 2 cLpc only (220 samples -> 11 coefficients); lld_gemaps_harm: 1 = rows
  const float *op_in;               // input rows
  int64_t op_ld_in;
  float *op_out;                    // output rows
  int64_t op_ld_out;
  int64_t op_rows;
  int64_t lpc_ld, fm_ld;            // leading dimensions of lpc / formants (12 / 10 in the batch scratch)
  const float *op_f0;               // cHarmonics rows: F0 per row; formants = `formants` with fm_ld; magnitudes = op_in
};

// Constants of cSpectral for one spectrum geometry (host-resolved in smilehip_plan.cpp)
struct SpectralConsts {
  double fsSec;               // frameSizeSec of the magnitude level: frq[i] = i / fsSec (transformFft.cpp:102-117)
  const double *sharp_w;      // [K-1] bark(f) * g(bark(f)) for bins 1..K-1 (spectral.cpp:1440-1455)
  int32_t band_iL[2], band_iR[2];
  double band_wL[2], band_wR[2];
  double slope_Sf, slope_S2f;
  const void *log_tab;        // kLogTab's copy in LDS (128 double2) where the kernel staged one, else null: the table in global memory
};

// General functionals (lld_funcspec.hip): one cFunctionals instance over columns [col_first, col_first + n_cols) of
// the utterances' rows (or of ONE matrix if single_rows >= 0).
// Tables of the Modulation family (lld_funcspec.hip: fs_modulation; built by smilehip_funcspec.cpp: mod_prepare). size[s] serves
// the transform length n = 64 << s: the reference-order transform's tables, the spline's per-knot constants over the axis
// i / (T n) (smileMath_cspline_init, smileUtilSpline.c:138-153) and, per output bin, the knot below it and the three
// coefficients of smileMath_csplint_init (:295-342); ok = 0: the bins do not lie on that axis.
struct ModSizeTab {
  OouraTab oo;
  const double *sigma, *d1, *d2, *co;
  const int32_t *k;
  int32_t ok, pad;
};
struct ModTables {
  const float *win;            // the window function of every length N <= mod_win_frames, length N at N (N - 1) / 2
  ModSizeTab size[5];
};

struct FsParams {
  smilehip_func_spec spec;
  const float *x;
  int64_t ld_x;
  int32_t col_first, n_cols;
  const int64_t *row_off;    // [n_utt+1]; unused if single_rows >= 0
  int64_t single_rows;
  int32_t rows_cut;          // rows used per utterance = max(1, rows - rows_cut - p) (0 if the utterance has none)
  const int32_t *pending;    // optional [n_utt]: p = pending[u] if smaller than the utterance's rows - 1, else 0
  int32_t per;               // values per column
  const float *extra;        // optional: one more row per utterance, [n_utt x ld_extra]
  int64_t ld_extra;
  // scratch, all indexed by scratch row = row_off[u] + u (+ t) and column: [scratch_rows x n_cols]
  float *nz;                 // compacted columns (nonZeroFuncts)
  unsigned char *alive;      // Peaks2
  float *sorted;             // Percentiles beyond the LDS capacity: 2 * scratch_rows * n_cols floats
  float *st_min, *st_max, *st_mean;   // [n_utt x n_cols]
  int32_t *st_n;
  float *out;                // [n_utt x ld_out]
  int64_t ld_out;
  int64_t max_rows;          // upper bound of the rows of any column (host knowledge): decides whether the workgroup sort is launched
  ModTables mod;             // Modulation family only
};

// R13: chain of window processors (cDeltaRegression / cContourSmoother) over the
// rows of each utterance. Level 0 = the input block x (T rows, D columns);
// stage s (kind 0 = delta regression with deltawin W, 1 = simple moving average
// with smaWin = 2W+1) maps level s to level s+1 which has W more rows.
// The output keeps row_off[u+1]-row_off[u] rows per utterance (what the reference's
// multi-level readers keep: the shortest level that is written out).
struct ChainParams {
  const int64_t *frame_off;    // [n_utt+1] input rows (frames) per utterance
  const int64_t *row_off;      // [n_utt+1] output rows per utterance
  const int32_t *tile_utt;     // [n_tiles] output-row tiles: utterance
  const int32_t *tile_t0;      // [n_tiles] first output row of the tile
  int32_t n_tiles;
  int32_t n_utt;
  int32_t copy_col;            // >= 0: also copy level 0 (the input block) to this output column; -1: no copy
  const float *x;              // level 0
  int64_t ld_x;
  float *out;
  int64_t ld_out;
  int32_t D;
  int32_t n_stages;            // 1 or 2
  int32_t kind[2];
  int32_t W[2];
  int32_t out_col[2];          // column offset of level s+1 in `out`
  int32_t short_T;             // utterances with T <= short_T go to the tick-accurate path
  const int32_t *short_utts;
  int32_t n_short;
};

}  // namespace smilehip
