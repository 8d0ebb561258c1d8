// eGeMAPSv02 / GeMAPSv01b LLD level (BASELINE config 5, config/egemaps/v02/eGeMAPSv02.conf with
// config/gemaps/v01b/GeMAPSv01b_core.lld.conf.inc), the kernels beside the F0 group of lld_f0.hip:
//   lld_gemaps_frame20   one wave per run of 8 consecutive 20 ms frames (four runs per workgroup, no workgroup barrier after
//                        the table staging): Hamming window, FFT 512, magnitude; one mel bank in two scalings ->
//                        cPlp auditory spectrum + its ll1 mean (loudness), MFCC 1..4; cSpectral's GeMAPS option sets
//                        (log-spectrum slopes 0-500 / 500-1500, alpha ratio, Hammarberg index, flux over 0-5000 Hz);
//                        cEnergy energy2 of the raw frame; the 110 complex bins cSpecResample reads go to scratch
//   lld_gemaps_lpc       cSpecResample (smileDsp_irdft: a 220 x 219 real matrix applied to every frame's spectrum) as a
//                        register-tiled product -- one wave = 8 frames x 220 outputs, four outputs per lane, every
//                        output's float sum in the reference's own order -- then cLpc: autocorrelation (one thread per
//                        (frame, lag), sequential float sums) and Durbin's recursion
//   lld_gemaps_formants  cFormantLpc: one thread per frame, roots of the LP polynomial by the reference's balanced
//                        companion-matrix QR iteration in double (zerosolve.cpp), formant frequencies / bandwidths
//   lld_gemaps_harm      cHarmonics, one wave per voiced 60 ms frame: gauss window, FFT 1024, magnitude; ACF of the squared
//                        magnitudes by the inverse real FFT -> harmonics-to-noise ratio at the F0 lag; the 100 harmonic
//                        peaks (one lane per harmonic), H1-H2, H1-A3, formant amplitudes F1..F3 relative to F0
//   lld_gemaps_tail      per utterance: cDataSelector picks, the voiced / unvoiced cValbasedSelector gates, the nine
//                        cContourSmoother instances with the graph's end-of-input rules -> the 25-column LLD level and the
//                        levels the functionals read
// Reference-order arithmetic where the result is order-sensitive (float sums of cSpecResample / cLpc, the QR iteration);
// sums the reference keeps in double are formed as double tree sums. Parity first; bounds in DESIGN.md.
#include <hip/hip_runtime.h>
#include "kernel_timing.hpp"

#include <cstring>

#include "lld_blocks.hpp"
#include "lld_fft.hpp"
#include "lld_blocks_compare.hpp"
#include "lld_gemaps_quad.hpp"
#include "lld_device.hpp"
#include "lld_launch.hpp"
#include "lld_params.hpp"

namespace smilehip {

// Development instrumentation (tools/ubench/variant_any.sh gemaps <name> -DSMILEHIP_PHASE_TIMING): s_memtime at the phase
// boundaries of lld_gemaps_harm, summed over all waves. Not compiled into the product.
#ifdef SMILEHIP_PHASE_TIMING
__device__ unsigned long long g_phase_gm[16];
#define GPHASE_DECL unsigned long long gph_acc[8] = {0}; unsigned long long gph_last = __builtin_amdgcn_s_memtime();
#define GPHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); gph_acc[i] += t_ - gph_last; gph_last = t_; } while (0)
#define GPHASE_FLUSH do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_phase_gm[i_], gph_acc[i_]); } while (0)
// the 20 ms frame kernel's phases go to slots 8 .. 15
#define GPHASE20_FLUSH do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_phase_gm[8 + i_], gph_acc[i_]); } while (0)
}  // namespace smilehip
extern "C" int smilehip_debug_phase_gm(unsigned long long *out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(smilehip::g_phase_gm), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(smilehip::g_phase_gm), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
namespace smilehip {
#else
#define GPHASE_DECL
#define GPHASE(i)
#define GPHASE_FLUSH
#define GPHASE20_FLUSH
#endif

namespace {
constexpr int kRun = 8;            // 20 ms frames per run (same runs as the ComParE A+B kernel)
constexpr int kRsI = 220;          // resampled samples per frame (cSpecResample: 20 ms at 11 kHz)
constexpr int kRsB = 109;          // complex bins the slow inverse DFT uses (kMax/2 - 1)
constexpr int kLpcP = 11;          // cLpc p
constexpr int kLpcTile = 32;       // frames per workgroup of the resampling / LPC kernel (8 per wave)
constexpr int kNH = 100;           // cHarmonics nHarmonics

// smileMath_quadFrom3pts (smileUtil.c:1009-1033)
__device__ __forceinline__ double quad3(double x1, double y1, double x2, double y2, double x3, double y3, double &y) {
  const double den = x1 * x1 * x2 + x2 * x2 * x3 + x3 * x3 * x1 - x3 * x3 * x2 - x2 * x2 * x1 - x1 * x1 * x3;
  if (den != 0.0) {
    const double a = (y1 * x2 + y2 * x3 + y3 * x1 - y3 * x2 - y2 * x1 - y1 * x3) / den;
    const double b = (x1 * x1 * y2 + x2 * x2 * y3 + x3 * x3 * y1 - x3 * x3 * y2 - x2 * x2 * y1 - x1 * x1 * y3) / den;
    const double c = (x1 * x1 * x2 * y3 + x2 * x2 * x3 * y1 + x3 * x3 * x1 * y2 - x3 * x3 * x2 * y1 - x2 * x2 * x1 * y3 - x1 * x1 * x3 * y2) / den;
    if (a != 0.0) {
      const double x = -b / (2.0 * a);
      y = c - a * x * x;
      return x;
    }
  }
  if (y1 > y2 && y1 > y3) { y = y1; return x1; }
  if (y2 > y1 && y2 > y3) { y = y2; return x2; }
  if (y3 > y1 && y3 > y2) { y = y3; return x3; }
  y = y1;
  return x1;
}
}  // namespace

// cSpectral::processVector with the GeMAPS option sets, one wave per frame: [gemapsv01b_logSpectral] (slopes 0-500 and
// 500-1500 Hz of the log spectrum, alpha ratio, Hammarberg index) and [egemapsv02_logSpectral_flux] (flux over freqRange
// 0-5000 Hz); squareInput = 1, useLogSpectrum = 1, specFloor 1e-7. mg / pw: magnitudes and powers of the K bins, prev: the
// previous frame's magnitudes (ignored when first), lg: 64 floats of LDS scratch. dst5 (lane 0 writes): slope0-500,
// slope500-1500, alphaRatioDB, hammarbergIndexDB, spectralFlux.
__device__ __forceinline__ void gemaps_spectral_wave(const float *mg, const float *pw, const float *prev, bool first, float *lg,
                                                     const GemapsParams &G, int K, int lane, float *dst5) {
  const double F0 = 1.0 / G.fsSec;                       // frq[i] = F0 * i (transformFft.cpp:102-117)
  // log power spectrum of the bins the two slopes cover (spectral.cpp:689-716): factor 10/ln 10 as float, floor at specFloor^2
  {
    const float p = pw[lane];
    lg[lane] = (p <= G.spec_floor) ? G.log_spec_floor : G.log_spec_factor * glibc_logf(p);
  }
  WaveG::sync();
  // band slopes of the log spectrum (spectral.cpp:872-992), frequency axis given: four double sums per band
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int iL = G.sl_iL[b], iR = G.sl_iR[b];
    const double wL = G.sl_wL[b], wR = G.sl_wR[b];
    double v[4] = {0.0, 0.0, 0.0, 0.0};                // Sf, S2f, sumA, sumB
    const int j = iL + lane;
    if (j <= iR) {
      const double f = F0 * (double)j, l = (double)lg[j];
      if (j == iL) { const double fw = f * wL; v[0] = fw; v[1] = fw * fw; v[2] = fw * l; v[3] = wL * l; }
      else if (j == iR) { const double fw = f * wR; v[0] = fw; v[1] = fw * fw; v[2] = fw * l; v[3] = wR * l; }
      else { v[0] = f; v[1] = f * f; v[2] = f * l; v[3] = l; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = WaveG::sum(v[q], nullptr);
    if (lane == 0) {
      const double Nind = G.sl_Nind[b];
      const double deno = (Nind * v[1] - v[0] * v[0]);
      double slope = 0.0;
      if (deno != 0.0) slope = (Nind * v[2] - v[0] * v[3]) / deno;
      dst5[b] = (float)slope;                       // oldSlopeScale = 0
    }
  }
  // alpha ratio (:995-1037) and Hammarberg index (:1039-1089) over the bins up to 5000 Hz, flux (:1124-1254) over freqRange
  {
    double fl = 0.0;
    float m02 = 0.0f, m25 = 0.0f;
    for (int j = lane; j < K; j += 64) {
      const double f = F0 * (double)j;
      if (f > 5000.0) break;
      const float p = pw[j];
      if (f < 2000.0) m02 = p > m02 ? p : m02; else m25 = p > m25 ? p : m25;
    }
    // sum01 / sum15 are FLOAT_DMEM accumulators (:997-1022): one lane adds the powers bin after bin, the two bands as two
    // interleaved chains (a band that has ended adds +0: s + 0 = s)
    // (the band edges n1 / n2 come from the plan: finding them with two loops of double products on one lane, every frame, was part of
    // the 39 % this function took of the 20 ms kernel; the two chains run side by side on lanes 0 and 1)
    float s01 = 0.0f, s15 = 0.0f;
    if (lane < 2) {
      const float sb = seq_sum_f32(pw, lane == 0 ? 0 : G.ar_n1, lane == 0 ? G.ar_n1 : G.ar_n2);   // (measured alternative: every lane walking the bins with v_readlane, 2 % slower)
      s01 = sb;
    }
    s15 = __shfl(s01, 1);
    for (int j = G.rng_lo + lane; j <= G.rng_hi; j += 64) {
      const double myB = (double)mg[j] - (double)prev[j];
      fl += myB * myB;
    }
    fl = WaveG::sum(fl, nullptr);
    {
      auto fmx = [](int a, int b) { return __int_as_float(b) > __int_as_float(a) ? b : a; };
      m02 = __int_as_float(__builtin_amdgcn_readfirstlane(wave_tree_i(__float_as_int(m02), fmx)));
      m25 = __int_as_float(__builtin_amdgcn_readfirstlane(wave_tree_i(__float_as_int(m25), fmx)));
    }
    if (lane == 0) {
      const float sum01 = s01, sum15 = s15;
      float a = 0.0f, h = 0.0f;
      if (sum01 > 0.0f) {
        if (sum15 > G.spec_floor) a = (float)(10.0 * (double)glibc_logf(sum15 / sum01) / log(10.0));
        else a = (float)(10.0 * (double)(glibc_logf(G.spec_floor) - glibc_logf(sum01)) / log(10.0));
      }
      if (m25 > 0.0f) {
        if (m02 > G.spec_floor) h = (float)(10.0 * (double)glibc_logf(m02 / m25) / log(10.0));
        else h = (float)(10.0 * (double)(glibc_logf(G.spec_floor) - glibc_logf(m25)) / log(10.0));
      }
      dst5[2] = a;
      dst5[3] = h;
      const int nBins = G.rng_hi - G.rng_lo + 1;
      const double flux = (nBins > 0) ? fl / (double)nBins : 0.0;
      dst5[4] = (!first && flux > 0.0) ? (float)sqrt(flux) : 0.0f;      // first frame of a stream: 0 (:1132-1136)
    }
  }
}

// ------------------------------------------------------------------------------------------------ 20 ms frames
// LDS: shared coef[Kpad] | rng[128] | dct[16 x 32]; per wave z[fft_pairs(M)] pairs | mg[Kpad] | pw[Kpad] | prev[Kpad] |
// lg[64] | mel[32] | aud[32] | lmel[32]
__global__ void __launch_bounds__(256) lld_gemaps_frame20(LldParams P, GemapsParams G, int n_runs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int M = P.Nfft >> 1, K = P.K;
  const int Kpad = (K + 3) & ~3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *s_coef = smem;
  int32_t *s_rng = reinterpret_cast<int32_t *>(s_coef + Kpad);
  float *s_dct = reinterpret_cast<float *>(s_rng + 128);
  for (int i = threadIdx.x; i < K; i += blockDim.x) s_coef[i] = P.mel_coef[i];
  for (int i = threadIdx.x; i < 4 * P.n_bands; i += blockDim.x) s_rng[i] = P.mel_rng[i];
  for (int i = threadIdx.x; i < P.n_mfcc * P.n_bands; i += blockDim.x) s_dct[i] = P.dct_rows[i];
  const OouraTab OO = oo_stage_tables(P.oo, s_dct + 16 * 32, threadIdx.x, blockDim.x);   // reference-order FFT tables (or none)
  __syncthreads();                                       // the only workgroup barrier
  const int run = blockIdx.x * 4 + wave;
  if (run >= n_runs) return;
  const int zf = 2 * fft_pairs(M) > 2 * Kpad ? 2 * fft_pairs(M) : 2 * Kpad;     // the transform's pairs, then the mel terms (two rows of Kpad floats)
  const int per_wave = zf + 3 * Kpad + 64 + 96;
  float2 *z = reinterpret_cast<float2 *>(s_dct + 16 * 32 + oo_table_floats(P.oo) + wave * per_wave);   // the transform's (re, im) pairs
  const int zpad = fft_pad(M);
  float *mg = reinterpret_cast<float *>(z) + zf;
  float *yv = mg;                                        // the raw frame lives in mg | pw (N <= 2 M < 2 Kpad) until the transform has read it
  float *pw = mg + Kpad;
  float *prev = pw + Kpad;
  float *lg = prev + Kpad;
  float *melv = lg + 64;
  float *aud = melv + 32;
  float *lmel = aud + 32;
  int logM = 0;
  while ((1 << logM) < M) ++logM;

  const int u = G.run_utt[run];
  const int t0 = G.run_t0[run];
  const int64_t f0 = P.frame_off[u];
  const int T20 = (int)(P.frame_off[u + 1] - f0);
  const int16_t *xu = P.pcm + P.samp_off[u];
  const float *xuf = P.pcm_f32 ? P.pcm_f32 + P.samp_off[u] : nullptr;    // float input (smilehip_lld_run_f32): read at the frame, no prefetch
  const int run_len = G.run_frames > 0 ? G.run_frames : kRun;
  const int t_last = (t0 + run_len < T20) ? t0 + run_len : T20;
  const int lane_in = lane;
  // the raw samples of frame t + 1 are asked for while frame t is processed (see lld_compare_frame_wave)
  int16_t pre[8];
  auto prefetch = [&](const int16_t *xx, int ln) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int n = ln + 64 * q; pre[q] = (n < P.N) ? xx[n] : (int16_t)0; }
  };
  const bool ahead = !xuf && P.N <= 512;                 // (longer frames -- 20 ms above 25.6 kHz -- are read at the frame like float input)
  if (ahead) prefetch(xu + (int64_t)(t0 > 0 ? t0 - 1 : 0) * P.H, lane);
  GPHASE_DECL
  for (int t = (t0 > 0 ? t0 - 1 : 0); t < t_last; ++t) {
    int lane = lane_in;                                  // opaque per frame (see lld_compare_frame_wave): nothing that depends on
    asm volatile("" : "+v"(lane));                       // the lane only is kept in registers across the frame loop
    const bool warm = t < t0;
    const int16_t *x = xu + (int64_t)t * P.H;
    float *raw = G.raw20 + (f0 + t) * 12;
    if (ahead) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int n = lane + 64 * q; if (n < P.N) yv[n] = pcm16_to_float(pre[q]); }
      if (t + 1 < t_last) prefetch(x + P.H, lane);
    } else {
      for (int n = lane; n < P.N; n += 64) yv[n] = xuf ? xuf[(int64_t)t * P.H + n] : pcm16_to_float(x[n]);
    }
    WaveG::sync();
    // cEnergy energy2 of the raw frame (energy.cpp:152-170): float squares added in double
    double e2 = 0.0;
    if (!warm) {
      for (int n = lane; n < P.N; n += 64) { const float tmp = yv[n]; e2 += tmp * tmp; }
      e2 = WaveG::sum(e2, nullptr);
    }
    GPHASE(0);   // frame into LDS, energy2
    const auto load_pair = [&](int i) {
      const int n0 = 2 * i - P.pad_left, n1 = n0 + 1;
      return make_float2((n0 >= 0 && n0 < P.N) ? yv[n0] * P.window[n0] + P.win_offset : 0.0f,
                         (n1 >= 0 && n1 < P.N) ? yv[n1] * P.window[n1] + P.win_offset : 0.0f);
    };
    if (OO.tw) oo_wave_forward(z, OO, lane, load_pair);  // the reference's rdft network, register form (lld_ooura_wave.hpp)
    else wave_cfft(z, M, P.tw_half, lane, load_pair);
    GPHASE(1);   // FFT
    float *spec = G.spec220 + (f0 + t) * kRsI;
    for (int k = lane; k <= M; k += 64) {
      const float2 X = OO.tw ? oo_wave_bin(z, OO, k) : wave_untangle(z, M, zpad, k, P.tw_full);
      const float m = bin_magnitude(X, k == 0 || k == M);
      mg[k] = m;
      pw[k] = m * m;                                     // squareInput (spectral.cpp:677-684) == melspec usePower
      if (!warm) {
        // what cSpecResample reads of the complex level (Ooura packing, fftsg.c:103-135): a[0], then a[2k], a[2k+1] = -Im
        // stored as (Re, Im) pairs of bins 1..109, then a[0], then one pad
        if (k == 0) { spec[2 * kRsB] = X.x; spec[2 * kRsB + 1] = 0.0f; }
        else if (k <= kRsB) { spec[2 * k - 2] = X.x; spec[2 * k - 1] = -X.y; }
      }
    }
    WaveG::sync();
    GPHASE(2);   // magnitudes, spectrum rows for cSpecResample
    if (warm) {
      for (int k = lane; k < K; k += 64) prev[k] = mg[k];
      WaveG::sync();
      continue;
    }
    // R6 once, two scalings: [gemapsv01b_melspec1] (htk = 0) feeds cPlp, [egemapsv02_melspecMfcc] (htk = 1) feeds cMfcc
    float *mt_a = reinterpret_cast<float *>(z), *mt_r = mt_a + Kpad;      // (z: free behind the transform; lld_device.hpp: mel_terms_fill)
    mel_terms_fill<WaveG>(pw, s_coef, K, mt_a, mt_r);
    WaveG::sync();
    if (lane < P.n_bands) {
      const int b = lane;
      const float acc = mel_band_from_terms(mt_a, mt_r, s_rng, b, 1.0f);
      melv[b] = acc;
      lmel[b] = log_mel(acc * P.mel_scale, P.melfloor, P.log_floor);
      aud[b] = plp_aud_band(acc, G.plp_melfloor, G.eql[b], G.compression);     // [gemapsv01b_audspec], plp.cpp:499-507
    }
    WaveG::sync();
    if (lane < P.n_mfcc) raw[6 + lane] = dct_coeff(lmel, s_dct + lane * P.n_bands, P.n_bands, P.dct_gain[lane]);   // R7
    if (lane == 32) {                                    // [gemapsv01b_audspecSum] ll1, vectorOperation.cpp:475-481
      const float d = seq_sum_f32(aud, P.n_bands);
      raw[0] = d / (float)P.n_bands;
    }
    if (lane == 0) raw[10] = (float)(e2 / (double)P.N) * 1.0f + 0.0f;
    GPHASE(3);   // mel, auditory spectrum, MFCC
    gemaps_spectral_wave(mg, pw, prev, t == 0, lg, G, K, lane, raw + 1);
    if (lane == 0) raw[11] = 0.0f;
    WaveG::sync();
    for (int k = lane; k < K; k += 64) prev[k] = mg[k];
    WaveG::sync();
    GPHASE(4);   // GeMAPS spectral descriptors
  }
  GPHASE20_FLUSH;
}

// Sixteen lanes per frame, four runs per wave (lld_gemaps_quad.hpp): the shipped geometry. Workgroups of four waves, three per CU.
namespace {
constexpr int kGmQuadWaves = 4;
inline size_t gemaps_quad_lds_floats(const OouraTab &oo) {
  return (size_t)gq::kTableFloats + (size_t)((oo_table_floats(oo) + 3) & ~3) + (size_t)kGmQuadWaves * 4 * gq::kRowFloats;
}
}  // namespace
template <int PAD>
__global__ void __launch_bounds__(kGmQuadWaves * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) lld_gemaps_frame20_quad(LldParams P, GemapsParams G, int n_runs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *s_win = smem;
  float *s_coef = s_win + gq::kN;
  int32_t *s_rng = reinterpret_cast<int32_t *>(s_coef + 260);
  float *s_dct = reinterpret_cast<float *>(s_rng + 128);
  constexpr int NT = kGmQuadWaves * 64;
  for (int i = threadIdx.x; i < gq::kN; i += NT) s_win[i] = P.window[i];
  for (int i = threadIdx.x; i < gq::kK; i += NT) s_coef[i] = P.mel_coef[i];
  for (int i = threadIdx.x; i < 4 * gq::kBands; i += NT) s_rng[i] = P.mel_rng[i];
  for (int i = threadIdx.x; i < gq::kMfcc * gq::kBands; i += NT) s_dct[i] = P.dct_rows[i];
  const OouraTab OO = oo_stage_tables<true>(P.oo, smem + gq::kTableFloats, threadIdx.x, NT);
  __syncthreads();                                       // the only workgroup barrier
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int first_run = (blockIdx.x * kGmQuadWaves + wave) * 4;
  if (first_run >= n_runs) return;
  float *fmem = smem + gq::kTableFloats + ((oo_table_floats(P.oo) + 3) & ~3) + wave * 4 * gq::kRowFloats;
  gemaps_frame20_quad_body<PAD>(P, G, n_runs, first_run, s_win, s_coef, s_rng, s_dct, OO, fmem);
}

// ------------------------------------------------------------------------------------------------ cSpecResample + cLpc
// smileDsp_irdft (smileUtil.c:1800-1820): out[i] = in[0]; for k = 2, 4, ..: out[i] += in[k] * cos[k/2][i]; out[i] += in[k+1] *
// sin[k/2][i]; out[i] /= (K/2) -- float, in this order, products rounded before they are added. One wave owns 8 frames of the
// tile and all 220 outputs (lane l: outputs l, l+64, l+128, l+192): a table element is loaded once per 8 frames, a spectrum
// element once per 4 outputs (LDS broadcast).
__global__ void __launch_bounds__(256) lld_gemaps_lpc(GemapsParams G) {
  // Row stride 236 floats = 12 mod 32 banks: the autocorrelation's lanes are (frame f, lag) pairs reading x[i - lag] of row f, bank
  // 12 f + i - lag -- the 32 (f, lag) pairs of a half wave then sit on 32 different banks (with the 224 of before every row
  // started on bank 0 and six rows read the same bank at different addresses: 2.5 conflict cycles per LDS instruction,
  // profiles/r03_pmc_egemaps.txt)
  constexpr int kXsRow = kRsI + 16;
  static_assert(kXsRow % 32 == 12 && kXsRow % 4 == 0, "bank layout of the autocorrelation reads");
  __shared__ __attribute__((aligned(16))) float xs[kLpcTile][kXsRow];        // spectra, then the resampled signals
  __shared__ float racf[kLpcTile][kLpcP + 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t g0 = (int64_t)blockIdx.x * kLpcTile;
  const int64_t n_all = G.op_mode ? G.op_rows : G.total_frames20;
  const int n_fr = (int)((n_all - g0 < kLpcTile) ? n_all - g0 : kLpcTile);
  for (int i = threadIdx.x; i < kLpcTile * kRsI; i += 256) {
    const int f = i / kRsI, c = i - f * kRsI;
    float v = 0.0f;
    if (f < n_fr) {
      if (G.op_mode == 0) v = G.spec220[(g0 + f) * kRsI + c];
      else if (G.op_mode == 2) v = G.op_in[(g0 + f) * G.op_ld_in + c];                       // the resampled signal itself
      else {                                                                                  // Ooura-packed spectrum row
        const float *a = G.op_in + (g0 + f) * G.op_ld_in;
        v = (c < 2 * kRsB) ? a[c + 2] : (c == 2 * kRsB ? a[0] : 0.0f);
      }
    }
    xs[f][c] = v;
  }
  __syncthreads();
  if (G.op_mode != 2) {
    constexpr int FW = kLpcTile / 4;                     // frames per wave
    const int fw = wave * FW;
    float acc[FW][4];
#pragma unroll
    for (int f = 0; f < FW; ++f) {
      const float dc = xs[fw + f][2 * kRsB];
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[f][o] = dc;
    }
    const bool o3 = lane + 192 < kRsI;
    for (int b = 0; b < kRsB; ++b) {
      float c[4], s[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int i = lane + 64 * o;
        c[o] = (o < 3 || o3) ? G.rs_cos[b * kRsI + i] : 0.0f;
        s[o] = (o < 3 || o3) ? G.rs_sin[b * kRsI + i] : 0.0f;
      }
#pragma unroll
      for (int f = 0; f < FW; ++f) {
        const float2 z = *reinterpret_cast<const float2 *>(&xs[fw + f][2 * b]);       // (Re, Im Ooura) of bin b+1
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          acc[f][o] += z.x * c[o];
          acc[f][o] += z.y * s[o];
        }
      }
    }
    WaveG::sync();                                       // this wave's rows are only read by this wave up to here
#pragma unroll
    for (int f = 0; f < FW; ++f)
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < 3 || o3) xs[fw + f][lane + 64 * o] = acc[f][o] / G.rs_norm;   // /= (FLOAT_DMEM)(K/2), K = 512 inputs at 16 kHz
  }
  __syncthreads();
  if (G.op_mode == 1) {                                  // cSpecResample alone: the resampled frames are the output
    for (int i = threadIdx.x; i < n_fr * kRsI; i += 256) {
      const int f = i / kRsI, c = i - f * kRsI;
      G.op_out[(g0 + f) * G.op_ld_out + c] = xs[f][c];
    }
    return;
  }
  // smileDsp_autoCorr (smileUtil.c:1560-1570): r[lag] = sum_{i >= lag} x[i] * x[i-lag], float, ascending i
  for (int task = threadIdx.x; task < kLpcTile * (kLpcP + 1); task += 256) {
    const int f = task / (kLpcP + 1), lag = task - f * (kLpcP + 1);
    const float *x = xs[f];
    float r = 0.0f;
    for (int i = lag; i < kRsI; ++i) r += x[i] * x[i - lag];
    racf[f][lag] = r;
  }
  __syncthreads();
  // smileDsp_calcLpcAcf (Durbin, smileUtil.c:1572-1630); cLpc::processVector with saveLPCoeff only (lpc.cpp:171-213)
  if (threadIdx.x < n_fr) {
    const float *r = racf[threadIdx.x];
    float a[kLpcP + 1];
#pragma unroll
    for (int i = 0; i <= kLpcP; ++i) a[i] = 0.0f;
    if (!((r[0] == 0.0f) || (r[0] == -0.0f))) {
      float e = r[0];
#pragma unroll
      for (int m = 1; m <= kLpcP; m++) {
        float sum = (float)1.0 * r[m];
#pragma unroll
        for (int i = 1; i < m; i++) sum += a[i - 1] * r[m - i];
        const float k_m = ((float)-1.0 / e) * sum;
        a[m - 1] = k_m;
#pragma unroll
        for (int i = 1; i <= m / 2; i++) {
          const float x = a[i - 1];
          a[i - 1] += k_m * a[m - i - 1];
          if ((i < (m / 2)) || ((m & 1) == 1)) a[m - i - 1] += k_m * x;
        }
        e *= ((float)1.0 - k_m * k_m);
        if (e == 0.0f) {
          for (int i = m; i < kLpcP; i++) a[i] = 0.0f;
          break;
        }
      }
    }
    float *o = G.lpc + (g0 + threadIdx.x) * G.lpc_ld;
#pragma unroll
    for (int i = 0; i < kLpcP; ++i) o[i] = a[i];
    if (G.lpc_ld > kLpcP) o[kLpcP] = 0.0f;
  }
}

// ------------------------------------------------------------------------------------------------ cFormantLpc
// zerosolve.cpp restated for one thread: companion matrix of the monic polynomial, balancing, Francis QR steps; h is the
// thread's nc x nc matrix (row-major), 1-based accessors like the reference's MATF. The matrices of a wave's 64 frames
// live in LDS, element-major: element e of lane l at double index e * 64 + l -- the bank of an access depends on the lane
// only, so the 64 lanes never conflict however their indices diverge (as private arrays they were 1.2 KB of scratch per
// lane: 60 GB of HBM traffic per million frames, profiles/r02_pmc_egemaps.txt).
namespace {
constexpr int kNC = kLpcP;
#define GM_MATC(m, i, j) ((m)[((i) * kNC + (j)) * 64])
#define GM_MATF(m, i, j) ((m)[(((i) - 1) * kNC + ((j) - 1)) * 64])
#define GM_ROOT(r, i) ((r)[(i) * 64])
#define GM_EPS 2.2204460492503131e-16

// Register form (round 2, third version): the matrix lives in the lane's registers. Every loop of the reference over matrix
// indices is unrolled over its full static range 1..nc and its body predicated with the reference's runtime bounds, so that
// every matrix access has compile-time indices; a body whose predicate is false for all 64 lanes is skipped by a wave-uniform
// branch. Each executed operation is the reference's own, in its order: the roots are bit-identical to the serial code's.
struct ZsMat {
  double a[kNC][kNC];
};
#define ZM(m, i, j) ((m).a[(i) - 1][(j) - 1])             // 1-based like the reference's MATF; i, j compile-time after unrolling

__device__ __forceinline__ void zs_balance_reg(ZsMat &m) {   // zerosolveBalanceCmatrix, zerosolve.cpp:22-84
  const double radix = 2.0, radix2 = 4.0;
  bool converged = false;
  while (__any(!converged)) {
    const bool active = !converged;                        // lanes that are done go through the sweep without effect
    converged = true;
#pragma unroll
    for (int i = 0; i < kNC; i++) {
      double nrow, ncol;
      if (i != kNC - 1) ncol = fabs(m.a[i + 1][i]);
      else {
        ncol = 0.0;
#pragma unroll
        for (int j = 0; j < kNC - 1; j++) ncol += fabs(m.a[j][kNC - 1]);
      }
      if (i == 0) nrow = fabs(m.a[0][kNC - 1]);
      else if (i == kNC - 1) nrow = fabs(m.a[i][i - 1]);
      else nrow = (fabs(m.a[i][i - 1]) + fabs(m.a[i][kNC - 1]));
      if (!active || ncol == 0.0 || nrow == 0.0) continue;
      double t2 = 1.0, t1 = nrow / radix;
      const double t3 = ncol + nrow;
      while (ncol < t1) { t2 *= radix; ncol *= radix2; }
      t1 = nrow * radix;
      while (ncol > t1) { t2 /= radix; ncol /= radix2; }
      if ((nrow + ncol) < 0.95 * t3 * t2) {
        converged = false;
        t1 = 1.0 / t2;
        if (i == 0) m.a[0][kNC - 1] *= t1;
        else { m.a[i][i - 1] *= t1; m.a[i][kNC - 1] *= t1; }
        if (i == kNC - 1) {
#pragma unroll
          for (int j = 0; j < kNC; j++) m.a[j][i] *= t2;
        } else m.a[i + 1][i] *= t2;
      }
    }
  }
}

// zerosolveQRhelper, zerosolve.cpp:100-283. root: the lane's (re, im) pairs in LDS, element e at root[e * 64]
__device__ __forceinline__ void zs_qr_reg(ZsMat &h, double *root) {
  int N = kNC, nit = 0;
  double t = 0.0;
  bool live = true;                                        // false: all roots found, or given up after 70 iterations
  while (__any(live)) {
    // ---- e: the last small sub-diagonal element (e = N .. 2; 1 if none)
    int e = 1;
    {
      bool found = false;
#pragma unroll
      for (int ec = kNC; ec >= 2; ec--) {
        if (ec <= N && !found) {
          const double a1 = fabs(ZM(h, ec, ec - 1)), a2 = fabs(ZM(h, ec - 1, ec - 1)), a3 = fabs(ZM(h, ec, ec));
          if (a1 <= GM_EPS * (a2 + a3)) { found = true; e = ec; }
        }
      }
    }
    double x = 0.0, y = 0.0, w = 0.0;
#pragma unroll
    for (int c = 1; c <= kNC; c++) {
      if (c == N) x = ZM(h, c, c);
      if (c >= 2) { if (c == N) { y = ZM(h, c - 1, c - 1); w = ZM(h, c - 1, c) * ZM(h, c, c - 1); } }
    }
    if (live && e == N) {                                  // one real root
      root[(2 * (N - 1)) * 64] = x + t; root[(2 * (N - 1) + 1) * 64] = 0;
      N--;
      nit = 0;
      if (N == 0) live = false;
      continue;
    }
    if (live && e == N - 1) {                              // a pair
      double p = (y - x) / 2;
      const double q = p * p + w;
      double yy = sqrt(fabs(q));
      const double xx = x + t;
      if (q > 0) {
        if (p < 0) yy = -yy;
        yy += p;
        root[(2 * (N - 1)) * 64] = xx - w / yy; root[(2 * (N - 1) + 1) * 64] = 0;
        root[(2 * (N - 2)) * 64] = xx + yy; root[(2 * (N - 2) + 1) * 64] = 0;
      } else {
        root[(2 * (N - 1)) * 64] = xx + p; root[(2 * (N - 1) + 1) * 64] = -yy;
        root[(2 * (N - 2)) * 64] = xx + p; root[(2 * (N - 2) + 1) * 64] = yy;
      }
      N -= 2;
      nit = 0;
      if (N == 0) live = false;
      continue;
    }
    if (live && nit == 70) live = false;
    if (!live) continue;                                   // (lanes that are done idle until the wave's last lane is)
    if (nit % 10 == 0 && nit > 0) {                        // exceptional shift
      t += x;
#pragma unroll
      for (int i = 1; i <= kNC; i++) if (i <= N) ZM(h, i, i) -= x;
      double sN = 0.0, sN1 = 0.0;
#pragma unroll
      for (int c = 3; c <= kNC; c++) if (c == N) { sN = ZM(h, c, c - 1); sN1 = ZM(h, c - 1, c - 2); }
      const double s = fabs(sN) + fabs(sN1);
      y = 3.0 / 4.0 * s;
      x = y;
      w = -0.4375 * s * s;
    }
    nit++;
    // ---- m: where the double-shift step starts (m = N-2 .. e)
    int m = e;
    double p = 0, q = 0, r = 0;
    {
      bool done = false;
#pragma unroll
      for (int mc = kNC - 2; mc >= 1; mc--) {
        if (mc <= N - 2 && mc >= e && !done) {
          const double z = ZM(h, mc, mc);
          double rr = x - z, ss = y - z;
          p = ZM(h, mc, mc + 1) + (rr * ss - w) / ZM(h, mc + 1, mc);
          q = ZM(h, mc + 1, mc + 1) - z - rr - ss;
          r = ZM(h, mc + 2, mc + 1);
          ss = fabs(p) + fabs(q) + fabs(r);
          p /= ss; q /= ss; r /= ss;
          m = mc;
          if (mc == e) done = true;
          else if (mc >= 2) {
            const double a1 = fabs(ZM(h, mc, mc - 1)), a2 = fabs(ZM(h, mc - 1, mc - 1)), a3 = fabs(ZM(h, mc + 1, mc + 1));
            if (a1 * (fabs(q) + fabs(r)) <= GM_EPS * fabs(p) * (a2 + a3)) done = true;
          }
        }
      }
    }
#pragma unroll
    for (int i = 3; i <= kNC; i++) if (i >= m + 2 && i <= N) ZM(h, i, i - 2) = 0;
#pragma unroll
    for (int i = 4; i <= kNC; i++) if (i >= m + 3 && i <= N) ZM(h, i, i - 3) = 0;
    // ---- the double QR step over k = m .. N-1
    double s = 0, z = 0;
#pragma unroll
    for (int k = 1; k <= kNC - 1; k++) {
      const bool on = k >= m && k <= N - 1;
      if (!__any(on)) continue;
      const bool notlast = (k != N - 1);
      bool go = on;
      if (k >= 2) {
        if (on && k != m) {
          p = ZM(h, k, k - 1);
          q = ZM(h, k + 1, k - 1);
          r = 0.0;
          if (k + 2 <= kNC) { if (notlast) r = ZM(h, k + 2, k - 1); }
          x = fabs(p) + fabs(q) + fabs(r);
          if (x == 0) go = false;
          else { p /= x; q /= x; r /= x; }
        }
      }
      if (go) {
        s = sqrt(p * p + q * q + r * r);
        if (p < 0) s = -s;
        if (k >= 2) {
          if (k != m) ZM(h, k, k - 1) = -s * x;
          else if (e != m) ZM(h, k, k - 1) *= -1;
        }
        p += s;
        z = r / s; y = q / s; x = p / s;
        r /= p; q /= p;
      }
#pragma unroll
      for (int j = k; j <= kNC; j++) {                     // rows k, k+1, k+2 over the columns j = k .. N
        if (go && j <= N) {
          double pp = ZM(h, k, j) + q * ZM(h, k + 1, j);
          if (k + 2 <= kNC) { if (notlast) { pp += r * ZM(h, k + 2, j); ZM(h, k + 2, j) -= pp * z; } }
          ZM(h, k + 1, j) -= pp * y;
          ZM(h, k, j) -= pp * x;
        }
      }
#pragma unroll
      for (int i = 1; i <= (k + 3 < kNC ? k + 3 : kNC); i++) {   // columns k, k+1, k+2 over the rows i = e .. min(k+3, N)
        if (go && i >= e && i <= N) {
          double pp = x * ZM(h, i, k) + y * ZM(h, i, k + 1);
          if (k + 2 <= kNC) { if (notlast) { pp += z * ZM(h, i, k + 2); ZM(h, i, k + 2) -= pp * r; } }
          ZM(h, i, k + 1) -= pp * q;
          ZM(h, i, k) -= pp;
        }
      }
    }
  }
}
}  // namespace

// cFormantLpc::processVector (formantLpc.cpp:192-290), nFormants = 5, saveFormants = saveBandwidths = 1, no median filter /
// octave correction. One thread per frame. When the QR iteration does not converge the reference goes on with what its
// roots member held before (the previous frame's values); a frame here starts from zeros instead (not observed on speech).
__global__ void __launch_bounds__(64) lld_gemaps_formants(GemapsParams G) {
  const int64_t g = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (g >= (G.op_mode ? G.op_rows : G.total_frames20)) return;
  const float *lp = G.lpc + g * G.lpc_ld;
  __shared__ double fm_roots[2 * kNC * 64];                              // the lanes' roots, element-major
  double *roots = fm_roots + threadIdx.x;
  double fc[5], bc[5];
  for (int i = 0; i < 2 * kNC; ++i) GM_ROOT(roots, i) = 0.0;
  ZsMat mat;
#pragma unroll
  for (int i = 0; i < kNC; i++)
#pragma unroll
    for (int j = 0; j < kNC; j++) mat.a[i][j] = 0.0;
#pragma unroll
  for (int i = 1; i < kNC; i++) mat.a[i][i - 1] = 1.0;                       // zerosolveSetCmatrix, zerosolve.cpp:86-98
#pragma unroll
  for (int i = 0; i < kNC; i++) mat.a[i][kNC - 1] = -(double)(-lp[kNC - i - 1]) / 1.0;   // a[i] = -lpc[n-1-i], a[n] = 1
  zs_balance_reg(mat);
  zs_qr_reg(mat, roots);
  for (int i = 0; i < kNC; i++) {                                            // smileMath_complexIntoUnitCircle, smileUtil.c:992-1003
    const double re = GM_ROOT(roots, 2 * i), im = GM_ROOT(roots, 2 * i + 1);
    if (sqrt(re * re + im * im) > 1.0) {
      const double c = re, d = -im;                                          // 1 / conj(root), smileMath_complexDiv :951-977
      double R = 0, I = 0;
      if (fabs(c) >= fabs(d)) {
        if (c != 0.0) { const double r = d / c, den = c + r * d; if (den != 0.0) { R = (1.0 + 0.0 * r) / den; I = (0.0 - r * 1.0) / den; } }
      } else {
        if (d != 0.0) { const double r = c / d, den = d + r * c; if (den != 0.0) { R = (1.0 * r + 0.0) / den; I = (0.0 * r - 1.0) / den; } }
      }
      GM_ROOT(roots, 2 * i) = R; GM_ROOT(roots, 2 * i + 1) = I;
    }
  }
  int n_found = 0;                                                           // smileDsp_lpcrootsToFormants, smileUtil.c:2019-2053
  {
    const double spPi = G.fm_T * M_PI, spPi2 = spPi * 2.0;
    double fHigh = G.fm_max;
    if ((fHigh < G.fm_min) || (fHigh > 1.0 / G.fm_T)) fHigh = 0.5 / G.fm_T - G.fm_min;
    for (int i = 0; i < kNC; i++) {
      const double re = GM_ROOT(roots, 2 * i), im = GM_ROOT(roots, 2 * i + 1);
      if (im < 0) continue;
      const double f = fabs(atan2(im, re)) / spPi2;
      if ((f >= G.fm_min) && (f <= fHigh)) {
        const double b = -log(sqrt(re * re + im * im)) / spPi;
#pragma unroll
        for (int k = 0; k < 5; ++k) if (k == n_found) { bc[k] = b; fc[k] = f; }   // (compile-time indices: the arrays stay in registers)
        n_found++;
        if (n_found >= 5) break;
      }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) if (i >= n_found) { fc[i] = 0.0; bc[i] = 0.0; }
  }
  int nz = 5;                                                                // ascending order, formantLpc.cpp:270-289
#pragma unroll
  for (int k = 4; k >= 0; --k) if (fc[k] == 0.0) nz = k;                      // the first zero
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int j = i + 1; j < 5; j++)
      if (j < nz && fc[j] < fc[i]) {
        double t = fc[j]; fc[j] = fc[i]; fc[i] = t;
        t = bc[j]; bc[j] = bc[i]; bc[i] = t;
      }
  float *o = G.formants + g * G.fm_ld;
  for (int i = 0; i < 5; i++) { o[i] = (float)fc[i]; o[5 + i] = (float)bc[i]; }
}
#undef GM_MATC
#undef GM_MATF
#undef GM_ROOT
#undef ZM
#undef GM_EPS

// ------------------------------------------------------------------------------------------------ cHarmonics
// cHarmonics::processVector (harmonics.cpp:743-1031) with [gemapsv01b_harmonics]'s options. One wave per tile of <= 8
// consecutive 60 ms frames; unvoiced frames (F0final == 0) only write the constants the reference emits.
// The 60 ms magnitude spectrum is the level cSpecScale reads (gemapsv01b_fftmagG60): the batch keeps lld_f0_spec's magnitudes
// (G.mag60, 2 KB per frame: 58 ms less here for 14 ms more there per 37 M frames) when they fit; without them -- batches near the
// memory limit, the per-component operator's rows come in by G.op_in -- the frame is windowed and transformed here.
// LDS: shared win[NP] | twh[256] | twf[260]; per wave z[576 pairs] (later hbin[128] | hfi[128] | hmag[128] | hlr[128]) |
// mg[516] | acf[516]
namespace {
// Geometry of the 60 ms spectrum by sample rate (as in lld_f0.hip): FFT 512 (8 kHz), 1024 (11.025 / 16 kHz: the tuned case),
// 2048 (22.05 .. 32 kHz), 4096 (44.1 / 48 kHz)
template <int LOGM>
struct HarmG {
  static constexpr int kHM = 1 << LOGM, kHK = kHM + 1, kHKP = kHM + 4;
  static constexpr size_t kTwBytes = (size_t)12 * kHM;    // twh | twf (4128 B for M = 512) or the reference-order tables (<= 12 M, lld_ooura.hpp)
  // M = 512: 8.7 KB of LDS per wave + 7.9 KB of tables per workgroup: two workgroups = 16 waves per CU
  static constexpr int kWaves = LOGM <= 9 ? 8 : (LOGM == 10 ? 4 : 2);
  static constexpr int kZ = LOGM == 9 ? WaveFft<9>::kZ : kHM + 64;   // (re, im) pairs of the transform's buffer (>= 256: the harmonics' arrays live there)
  static constexpr int kMC = LOGM <= 9 ? kHM : -1;        // register-resident transform for M = 256 / 512, in place in LDS above
};
__device__ __forceinline__ int harm_is_peak(const float *x, int N, int n) {  // cHarmonics::isPeak, :369-390
  if (n >= N || n < 0) return 0;
  if (n + 1 < N) {
    if (n > 0) { if (x[n] > x[n - 1] && x[n] > x[n + 1]) return 1; }
    else { if (x[0] > x[1]) return 1; }
  } else {
    if (n > 0) { if (x[n] > x[n - 1]) return 1; }
  }
  return 0;
}
// freqToBin (:403-415) on the linear axis frq[i] = Fb * i: the first bin above freq, or its lower neighbour if that one
// is closer; 0 if there is none. The reference's search start never lies above that bin (see the call sites), so the
// result does not depend on it.
__device__ __forceinline__ int harm_freq_to_bin(double Fb, float freq, int kHK) {
  const double f = (double)freq;
  int s = (int)(f / Fb);
  while (Fb * (double)s <= f) s++;
  while (s > 0 && Fb * (double)(s - 1) > f) s--;
  if (s >= kHK) return 0;
  if (s < 1) s = 1;
  return (Fb * (double)s - f > f - Fb * (double)(s - 1)) ? s - 1 : s;
}
}  // namespace

template <int LOGM>
__global__ void __launch_bounds__(HarmG<LOGM>::kWaves * 64) lld_gemaps_harm(LldParams P, F0Params Q, GemapsParams G) {
  using HG = HarmG<LOGM>;
  constexpr int kHM = HG::kHM, kHK = HG::kHK, kHKP = HG::kHKP, kHarmWaves = HG::kWaves, kMC = HG::kMC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NP = (Q.N + 3) & ~3;
  float *c_win = reinterpret_cast<float *>(smem_h);
  float2 *c_twh = reinterpret_cast<float2 *>(c_win + NP);
  float2 *c_twf = c_twh + kHM / 2;
  for (int i = threadIdx.x; i < Q.N; i += blockDim.x) c_win[i] = Q.window[i];
  OouraTab OO = OouraTab{};                              // reference-order transform: its tables take the place of twh | twf
  if (Q.oo.tw) OO = oo_stage_tables(Q.oo, reinterpret_cast<float *>(c_twh), threadIdx.x, blockDim.x);
  else {
    for (int i = threadIdx.x; i < kHM / 2; i += blockDim.x) c_twh[i] = Q.tw_half[i];
    for (int i = threadIdx.x; i <= kHM / 2; i += blockDim.x) c_twf[i] = Q.tw_full[i];
  }
  __syncthreads();                                       // the only workgroup barrier
  using Fft = WaveFft<9>;                                // kHM == 512, own-order A/B build: fused passes on (re, im) pairs, lld_fft.hpp
  constexpr int per_wave = 2 * HG::kZ + 2 * kHKP;
  float2 *z = reinterpret_cast<float2 *>(reinterpret_cast<unsigned char *>(c_twh) + HG::kTwBytes) + (size_t)wave * (per_wave / 2);
  float *mg = reinterpret_cast<float *>(z + HG::kZ);
  float *acf = mg + kHKP;
  int *hbin = reinterpret_cast<int *>(z);                // the harmonics' arrays live in the transform's buffer (dead after the ACF)
  float *hfi = reinterpret_cast<float *>(hbin + 128);
  float *hmag = hfi + 128;
  float *hlr = hmag + 128;
  const double Fb = 1.0 / G.fsSec60;                     // frequency axis of the 60 ms spectrum: frq[i] = Fb * i
  const int tile_stride = (int)gridDim.x * kHarmWaves;
  GPHASE_DECL
  const int lane_in = lane;
  const bool rows_mode = G.op_mode == 1;                 // per-component operator: F0, formants and magnitudes given per row
  const int n_tiles = rows_mode ? (int)((G.op_rows + 7) / 8) : G.n_tiles60;
  // Tiles cost what their voiced frames cost: with the batch's counter (G.harm_ctl: [0] next tile, [1] waves that have
  // finished -- the last one zeroes both) the waves take the next tile when they are free; without it (per-component
  // operator) tile = wave index + k * waves.
  int tile = blockIdx.x * kHarmWaves + wave;
  const auto next_tile = [&]() {
    if (!G.harm_ctl) return tile + tile_stride;
    int t = 0;
    if (lane_in == 0) t = atomicAdd(&G.harm_ctl[0], 1);
    return __builtin_amdgcn_readfirstlane(t);
  };
  if (G.harm_ctl) tile = next_tile();
  for (; tile < n_tiles; tile = next_tile()) {
    int64_t samp0 = 0, row0 = (int64_t)tile * 8, r20 = (int64_t)tile * 8;
    int n_fr = (int)((G.op_rows - row0 < 8) ? G.op_rows - row0 : 8);
    if (!rows_mode) {
      samp0 = G.tile60[tile].samp0; row0 = G.tile60[tile].row0; n_fr = G.tile60[tile].n_frames;
      // the 20 ms frame with the same start sample: row0 - frame_off60[u] + frame_off20[u]; resolved through the utterance
      int lo = 0, hi = P.n_utt;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (G.frame_off60[mid] <= row0) lo = mid; else hi = mid;
      }
      r20 = row0 - G.frame_off60[lo] + P.frame_off[lo];
    }
    // The kept spectra (G.mag60): a frame's nine loads per lane are issued one frame ahead -- while the frame before it is worked on --
    // for voiced frames only.
    constexpr int kMagPer = (kHM + 64) / 64;
    float mv[kMagPer];
    const auto fetch_mag = [&](int tfn) {
      if (!G.mag60 || rows_mode || tfn >= n_fr) return;
      const int64_t gn = row0 + tfn;
      if (!(G.pitch3[gn * 3] > 0.0f)) return;
      const float *mi = G.mag60 + gn * G.mag60_ld;
#pragma unroll
      for (int m = 0; m < kMagPer; ++m) {                // (branch-free: a load behind a lane condition brings its own branch and wait with it)
        const int k = lane_in + 64 * m;
        const float v = mi[k <= kHM ? k : 0];
        mv[m] = k <= kHM ? v : 0.0f;
      }
    };
#pragma unroll
    for (int m = 0; m < kMagPer; ++m) mv[m] = 0.0f;
    fetch_mag(0);
    for (int tf = 0; tf < n_fr; ++tf) {
      int lane = lane_in;                                // opaque per frame (see lld_compare_frame_wave)
      asm volatile("" : "+v"(lane));
      const int64_t g = row0 + tf;
      const float F0 = rows_mode ? G.op_f0[g] : G.pitch3[g * 3];
      float *o = rows_mode ? G.op_out + g * G.op_ld_out : G.harm6 + g * 6;
      if (!(F0 > 0.0f)) {
        if (lane == 0) {                                 // :790-810 (no ACF peak), :1005-1025
          o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f;
          o[3] = (float)-201.0; o[4] = (float)-201.0; o[5] = (float)-201.0;   // logRelValueFloorUnvoiced
        }
        fetch_mag(tf + 1);
        continue;
      }
      if (rows_mode) {
        const float *mi = G.op_in + g * G.op_ld_in;
        for (int k = lane; k <= kHM; k += 64) mg[k] = mi[k];
      } else if (G.mag60) {                              // the level as the pitch chain's lld_f0_spec kept it (same frame, window, transform)
#pragma unroll
        for (int m = 0; m < kMagPer; ++m) { const int k = lane_in + 64 * m; if (k <= kHM) mg[k] = mv[m]; }
        fetch_mag(tf + 1);
      } else {
        const PcmIn x = pcm_in(P) + (samp0 + (int64_t)tf * Q.H);
        const auto load_pair = [&](int i) {
          const int n0 = 2 * i - Q.pad_left, n1 = n0 + 1;
          const bool v0 = n0 >= 0 && n0 < Q.N, v1 = n1 >= 0 && n1 < Q.N;      // (branch-free: see lld_f0.hip's load_pair)
          const int c0 = v0 ? n0 : 0, c1 = v1 ? n1 : 0;
          const float a = x[c0] * c_win[c0], b = x[c1] * c_win[c1];
          return make_float2(v0 ? a : 0.0f, v1 ? b : 0.0f);
        };
        if constexpr (LOGM == 9) {
          if (OO.tw) oo_wave_forward<kMC>(z, OO, lane, load_pair);
          else Fft::forward(z, c_twh, lane, load_pair);
          for (int k = lane; k <= kHM; k += 64)
            mg[k] = bin_magnitude(OO.tw ? oo_wave_bin<kMC>(z, OO, k) : fft_untangle<Fft>(z, k, c_twf), k == 0 || k == kHM);
        } else {
          oo_wave_forward<kMC>(z, OO, lane, load_pair);
          for (int k = lane; k <= kHM; k += 64) mg[k] = bin_magnitude(oo_wave_bin<kMC>(z, OO, k), k == 0 || k == kHM);
        }
      }
      WaveG::sync();
      GPHASE(0);   // load, window, FFT, magnitudes
      // computeAcf (:590-630): inverse real FFT of the squared magnitudes, |.| / nBins, lags 0 .. nBins-1
      if (LOGM != 9 || OO.tw) {                          // rdft(N, -1) on the packed squares (harmonics.cpp:609-627)
        oo_wave_inverse<kMC>(z, OO, lane, [&](int e) {
          if (e == 0) { const float m0 = mg[0], m1 = mg[kHM]; return make_float2(m0 * m0, m1 * m1); }
          const float m = mg[e];
          return make_float2(m * m, 0.0f);
        });
        for (int k = lane; k <= kHM; k += 64) acf[k] = fabsf(oo_wave_inverse_out<kMC>(z, OO, k)) / (float)kHK;
        WaveG::sync();
      } else if constexpr (LOGM == 9) {
        const int n = 2 * kHM;
        Fft::forward(z, c_twh, lane, [&](int i) {        // the squared magnitudes, formed as they are asked for
          const int n0 = 2 * i, n1 = 2 * i + 1;
          const float m0 = mg[n0 <= kHM ? n0 : n - n0], m1 = mg[n1 <= kHM ? n1 : n - n1];
          return make_float2(m0 * m0, m1 * m1);
        });
        for (int k = lane; k <= kHM; k += 64) {
          const float a = 0.5f * fft_untangle<Fft>(z, k, c_twf).x;
          acf[k] = fabsf(a) / (float)kHK;
        }
        WaveG::sync();
      }
      GPHASE(1);   // ACF
      // HNR at the ACF peak closest to the F0 lag (freqToAcfBinLin :393-401, getClosestPeak :632-665, computeAcfHnr_dB :690-712)
      float hnr_db = 0.0f;
      {
        const double fs = Fb * (double)(kHK - 1) * 2.0;
        long idx = (long)(int)floor(fs / F0);
        long refined = 0;
        if (idx > 0) {
          if (harm_is_peak(acf, kHK, (int)idx)) refined = idx;
          else {
            long off = 1;
            bool found = false;
            while (idx - off > 0 || idx + off < kHK - 1) {
              if (idx - off > 0 && harm_is_peak(acf, kHK, (int)(idx - off))) { refined = idx - off; found = true; break; }
              if (idx + off < kHK - 1 && harm_is_peak(acf, kHK, (int)(idx + off))) { refined = idx + off; found = true; break; }
              off++;
            }
            if (!found) {
              const long ic = idx < kHK ? idx : kHK - 1;   // (F0 >= minPitch keeps the lag inside the array)
              const float xi = acf[ic];
              if (acf[0] > xi && acf[kHK - 1] <= xi) refined = 0;
              else if (acf[0] <= xi && acf[kHK - 1] > xi) refined = kHK - 1;
              else if (acf[0] > xi && acf[kHK - 1] > xi) refined = (idx < kHK / 2) ? 0 : kHK - 1;
              else refined = idx;
            }
          }
        }
        if (refined > 0 && refined < kHK) {
          double hnr = acf[0] - acf[refined], ret;
          if (hnr == 0.0) hnr = 10e10; else hnr = acf[refined] / hnr;
          if (hnr > 10e10) ret = 10.0 * log10(10e10);
          else if (hnr < 10e-10) ret = 10.0 * log10(10e-10);
          else ret = 10.0 * log10(hnr);
          hnr_db = (float)ret;
        }
      }
      GPHASE(2);   // HNR peak search
      // findHarmonicPeaks, frequency-axis branch (:478-546): harmonic i = lane, lane + 64
      const int firstBin = harm_freq_to_bin(Fb, 0.5f * F0, kHK);
      for (int i = lane; i < 128; i += 64) {
        int bin = -1;
        float fi = 0.0f, mag = 0.0f, mi = 0.0f;
        if (i < kNH) {
          const int candBin = harm_freq_to_bin(Fb, (float)(i + 1) * F0, kHK);
          int peakBin = -1;
          if (harm_is_peak(mg, kHK, candBin)) peakBin = candBin;
          else {
            int cl = candBin - 1, cr = candBin + 1;
            const int lower = harm_freq_to_bin(Fb, ((float)i + 0.5f) * F0, kHK);
            const int upper = harm_freq_to_bin(Fb, ((float)i + 1.5f) * F0, kHK);
            while ((cl >= lower || cr <= upper) && peakBin == -1) {
              if (cr <= upper) { if (harm_is_peak(mg, kHK, cr)) { peakBin = cr; break; } cr++; }
              if (cl >= lower) { if (harm_is_peak(mg, kHK, cl)) { peakBin = cl; break; } cl--; }
            }
          }
          if (peakBin >= firstBin && peakBin < kHK - 1) {
            bin = peakBin;
            mag = mg[peakBin];
            double m2 = 0.0;
            fi = (float)quad3(Fb * (double)(peakBin - 1), (double)mg[peakBin - 1], Fb * (double)peakBin, (double)mg[peakBin],
                              Fb * (double)(peakBin + 1), (double)mg[peakBin + 1], m2);
            mi = (float)m2;
          } else {
            bin = candBin;
          }
        }
        hbin[i] = bin; hfi[i] = fi; hmag[i] = mag; hlr[i] = mi;      // hlr: interpolated magnitude for now
      }
      WaveG::sync();
      GPHASE(3);   // harmonic peaks
      // postProcessHarmonics(…, true) (:550-588): log magnitudes relative to harmonic 0 (log10 of a float: log10f), then the
      // duplicate removal, which is sequential (an entry is compared with its predecessor AFTER that one was cleared)
      {
        const float m0 = hmag[0];
        const bool logRel = !(m0 == 0.0f);
        // log10f as glibc computes it (glibc_float.hpp)
        const float lm0 = logRel ? glibc_log10f(m0) : 0.0f;
        for (int i = lane; i < kNH; i += 64) {
          float v;
          if (i == 0) v = 0.0f;
          else if (!logRel) v = -201.0f;
          else if (hlr[i] > 0.0f) {
            const double tmp = (double)glibc_log10f(hlr[i]);
            v = (float)(20.0 * (tmp - (double)lm0));
            if (v < -200.0f) v = -200.0f;
          } else v = -200.0f;
          hlr[i] = v;
        }
        // Duplicate removal: entry i is cleared if its bin equals its predecessor's -- the predecessor's AFTER its own
        // clearing (then 0). cleared[i] = cleared[i-1] ? (bin[i] == 0) : (bin[i] == bin[i-1]): a two-state recurrence over two
        // bit masks, run on the scalar unit (the sequential loop over LDS it replaces was a quarter of the kernel's time).
        static_assert(kNH <= 128, "two 64-bit masks");
        const int b0 = hbin[lane], b0p = lane > 0 ? hbin[lane - 1] : -1;
        const int b1 = hbin[64 + lane], b1p = hbin[63 + lane];
        const unsigned long long E0 = __ballot(lane > 0 && b0 == b0p), Z0 = __ballot(b0 == 0);
        const unsigned long long E1 = __ballot(64 + lane < kNH && b1 == b1p), Z1 = __ballot(64 + lane < kNH && b1 == 0);
        unsigned long long C0 = 0ull, C1 = 0ull;
        {
          bool c = false;
          for (int i = 1; i < 64; ++i) { c = ((c ? Z0 : E0) >> i) & 1ull; C0 |= (unsigned long long)c << i; }
          for (int i = 0; i < kNH - 64; ++i) { c = ((c ? Z1 : E1) >> i) & 1ull; C1 |= (unsigned long long)c << i; }
        }
        WaveG::sync();
        if ((C0 >> lane) & 1ull) { hbin[lane] = 0; hfi[lane] = 0.0f; hmag[lane] = 0.0f; hlr[lane] = -201.0f; }
        if ((C1 >> lane) & 1ull) { hbin[64 + lane] = 0; hfi[64 + lane] = 0.0f; hmag[64 + lane] = 0.0f; hlr[64 + lane] = -201.0f; }
        WaveG::sync();
      }
      GPHASE(4);   // log magnitudes + duplicate removal
      // getFormantAmplitudeIndices (:714-740): the strongest harmonic within 0.8 .. 1.2 of the formant frequency
      int fa[3];
      const float *fm = G.formants + (r20 + tf) * G.fm_ld;
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const float fl = 0.8f * fm[f], fr = 1.2f * fm[f];
        float bm = 0.0f;
        int bi = 1 << 30;
        for (int h = lane; h < kNH; h += 64) {
          const float v = hfi[h];
          if (v >= fl && v <= fr && hmag[h] > bm) { bm = hmag[h]; bi = h; }      // ascending h: the first maximum
        }
        wave_argmax_f(bm, bi);
        fa[f] = (bi == (1 << 30) || !(bm > 0.0f)) ? -1 : bi;
      }
      if (lane == 0) {
        o[0] = hnr_db;
        float v = hlr[1] - hlr[2];                        // H1-H2 (:876-900)
        v = v < -201.0f ? -201.0f : (v > 201.0f ? 201.0f : v);
        o[1] = v;
        v = (fa[2] >= 0) ? hlr[1] - hlr[fa[2]] : (float)(hlr[1] - 201.0f);       // H1-A3 (:903-917 when A3 has no harmonic)
        v = v < -201.0f ? -201.0f : (v > 201.0f ? 201.0f : v);
        o[2] = v;
#pragma unroll
        for (int f = 0; f < 3; ++f) o[3 + f] = (fa[f] >= 0) ? hlr[fa[f]] : 0.0f;   // :957-973
      }
      WaveG::sync();
      GPHASE(5);   // formant amplitudes + output
    }
  }
  if (G.harm_ctl && lane_in == 0 && atomicAdd(&G.harm_ctl[1], 1) == (int)gridDim.x * kHarmWaves - 1) {
    G.harm_ctl[0] = 0;
    G.harm_ctl[1] = 0;
  }
  GPHASE_FLUSH;
}

// ------------------------------------------------------------------------------------------------ selectors + smoothers
// One workgroup per utterance. Rows: the LLD level keeps T60 + 1 (what both lldsetE_smo and lldsetF_smo hold); the levels
// the functionals read go to func_in with T20 + 1 rows per utterance (none for utterances without a 60 ms frame). End-of-input rules of the nine cContourSmoother
// instances as measured against the binary (oracle/lld_oracle_gemaps.c, smooth_level): levels that follow the Viterbi
// smoother only run in lockstep with it when nothing was decided before the end (P == T60: row n sees frames 0 .. n, row 0
// is x[0]); levels that also wait for cPitchJitter see themselves clipped at frame T60 - P - 1 in rows n <= T60 - P.
namespace {
enum { kEoiNone = 0, kEoiLockstep = 1, kEoiJitter = 2 };
struct Sma {
  int T, P, kind, n;
  __device__ __forceinline__ int clip() const {
    if (kind == kEoiLockstep && P >= T) return n < T - 1 ? n : T - 1;
    if (kind == kEoiJitter && T > 1 && P < T && n <= T - P) return T - P - 1;
    return T - 1;
  }
  __device__ __forceinline__ bool first_exact() const { return n == 0 && ((kind == kEoiLockstep && P >= T) || (kind == kEoiJitter && T == 1)); }
};
// cContourSmoother::processBuffer (contourSmoother.cpp:85-118), smaWin = 3; X(i): frame i of the level, clamped
template <class F>
__device__ __forceinline__ float sma3(const Sma &s, bool nz, F X) {
  int c = s.clip();
  if (c < 0) c = 0;
  auto I = [&](int i) { return i < 0 ? 0 : (i > c ? c : i); };
  // (the three values asked for together -- the clamped indices are always rows of the level --, then the cases: a load behind a
  //  condition brings its own branch and wait with it)
  const float a = X(I(s.n)), l = X(I(s.n - 1)), r = X(I(s.n + 1));
  if (s.first_exact()) return a;
  if (nz) {
    if (a != 0.0f) {
      int N = 1;
      float v = a;
      if (l != 0.0f) { v += l; N++; }
      if (r != 0.0f) { v += r; N++; }
      return v / (float)N;
    }
    return 0.0f;
  }
  float v = a;
  v += l;
  v += r;
  return v / (float)3;
}
}  // namespace

__global__ void __launch_bounds__(256) lld_gemaps_tail(const int64_t *frame_off20, const int64_t *row_off, int n_utt, GemapsParams G,
                                                      float *out, int64_t ld) {
  const int u = blockIdx.x;
  if (u >= n_utt) return;
  const int64_t f20 = frame_off20[u], f60 = G.frame_off60[u];
  const int T20 = (int)(frame_off20[u + 1] - f20), T = (int)(G.frame_off60[u + 1] - f60);
  const int P = G.pending ? G.pending[u] : 0;
  if (threadIdx.x == 0) G.pending_j[u] = (P < T) ? P : 0;
  if (T < 1) return;                                     // no 60 ms frame: no LLD rows, no functionals (func_in holds no rows)
  float *fin = G.func_in + G.fin_off[u] * 36;
  const float *raw = G.raw20 + f20 * 12, *fm = G.formants + f20 * 10;
  const float *p3 = G.pitch3 + f60 * 3, *j4 = G.jit4 + f60 * 4, *sdb = G.shim_db + f60, *h6 = G.harm6 + f60 * 6;
  float *o = out + row_off[u] * ld;
  // (row, column) items: columns 0..9 E (T20+1 rows, the first T+1 also go to the LLD level), 10..24 F, 25 logF0, 26..39 NoNz,
  // 40..48 specV, 49..53 specU (T+1 rows each)
  const int rowsE = T20 + 1, rowsF = T + 1;
  // column maps
  const int eRaw[10] = {0, 3, 4, 1, 2, 5, 6, 7, 8, 9};   // loudness, alphaRatio, hammarberg, slope0-500, slope500-1500, flux, mfcc1..4
  const int spRaw[9] = {3, 4, 1, 2, 5, 6, 7, 8, 9};      // the spectral set of the voiced / unvoiced selectors
  for (int64_t it = threadIdx.x; it < (int64_t)rowsE * 10; it += 256) {
    const int n = (int)(it / 10), c = (int)(it - (int64_t)n * 10);
    Sma s{T20, 0, kEoiNone, n};
    const int rc = eRaw[c];
    const float v = sma3(s, false, [&](int i) { return raw[(int64_t)i * 12 + rc]; });
    if (n < rowsF) o[(int64_t)n * ld + c] = v;
    if (c == 0) fin[(int64_t)n * 36] = v;                // loudness_sma3
    else if (c >= 5) fin[(int64_t)n * 36 + (c - 4)] = v; // flux, mfcc1..4 _sma3
  }
  for (int64_t it = threadIdx.x; it < (int64_t)rowsF * 44; it += 256) {
    const int n = (int)(it / 44), c = (int)(it - (int64_t)n * 44);
    // (a column's source as one pointer and one stride, picked once per item: the three rows' loads are then plain and together)
    const auto src_of = [&](int q, const float *&ptr, int &stride) {     // q: 0 F0finalLog | 1 jitter | 2 shimmer dB | 3..5 harm6[0..2] | 6.. F(k+1) f, bw, amp
      if (q == 0) { ptr = p3 + 1; stride = 3; }
      else if (q == 1) { ptr = j4; stride = 4; }
      else if (q == 2) { ptr = sdb; stride = 1; }
      else if (q < 6) { ptr = h6 + (q - 3); stride = 6; }
      else {
        const int k = (q - 6) / 3, w = (q - 6) - 3 * k;
        if (w == 2) { ptr = h6 + 3 + k; stride = 6; }
        else { ptr = fm + (w == 0 ? k : 5 + k); stride = 10; }
      }
    };
    if (c < 15) {                                        // [egemapsv02_lldSetSelectorF] -> [egemapsv02_smoFnz]
      Sma s{T, P, kEoiJitter, n};
      const float *ptr; int stride;
      src_of(c, ptr, stride);
      const float v = sma3(s, true, [&](int i) -> float { return ptr[(int64_t)i * stride]; });
      o[(int64_t)n * ld + 10 + c] = v;
    } else if (c == 15) {                                // [gemapsv01b_lldSetSelectorLogF0] -> [gemapsv01b_smoF0]
      Sma s{T, P, kEoiLockstep, n};
      fin[(int64_t)n * 36 + 6] = sma3(s, true, [&](int i) { return p3[(int64_t)i * 3 + 1]; });
    } else if (c < 30) {                                 // [gemapsv01b_formantVoiced] + [egemapsv02_lldSetSelectorNoF0LoudnNz]
      const int d = c - 16;
      Sma s{T, P, kEoiJitter, n};
      const float *ptr; int stride;
      src_of(d + 1, ptr, stride);                        // (d = 0 jitter ... = the selector F's column d + 1)
      const bool gated = d >= 5 && ((d - 5) % 3) != 2;   // formant frequency / bandwidth: zero in unvoiced frames
      fin[(int64_t)n * 36 + 7 + d] = sma3(s, true, [&](int i) -> float {
        const float x = ptr[(int64_t)i * stride], f0l = p3[(int64_t)i * 3 + 1];
        return (!gated || f0l > (float)0.000001) ? x : 0.0f;
      });
    } else if (c < 39) {                                 // [egemapsv02_logSpectralVoiced] + SelectorSpectralNz
      const int d = c - 30;
      Sma s{T, P, kEoiLockstep, n};
      fin[(int64_t)n * 36 + 21 + d] = sma3(s, true, [&](int i) -> float {
        const float x = raw[(int64_t)i * 12 + spRaw[d]], f0l = p3[(int64_t)i * 3 + 1];
        return (f0l > (float)0.000001) ? x : 0.0f;
      });
    } else {                                             // [egemapsv02_logSpectralUnvoiced] + SelectorSpectralZ
      const int d = c - 39;
      Sma s{T, P, kEoiLockstep, n};
      fin[(int64_t)n * 36 + 30 + d] = sma3(s, true, [&](int i) -> float {
        const float x = raw[(int64_t)i * 12 + spRaw[d]], f0l = p3[(int64_t)i * 3 + 1];
        return (f0l < (float)0.000001) ? x : 0.0f;
      });
    }
  }
  // rows of func_in beyond T (the 60 ms levels are shorter than the 20 ms ones): zero
  for (int64_t it = threadIdx.x; it < (int64_t)(rowsE - rowsF) * 29; it += 256) {
    const int n = rowsF + (int)(it / 29), c = 6 + (int)(it % 29);           // columns 6..34
    fin[(int64_t)n * 36 + c] = 0.0f;
  }
  for (int n = threadIdx.x; n < rowsE; n += 256) fin[(int64_t)n * 36 + 35] = (n < T20) ? raw[(int64_t)n * 12 + 10] : 0.0f;   // energy2
}

// [egemapsv02_leq] cVectorOperation dBp (vectorOperation.cpp:507-516) on one value per utterance, in place
__global__ void lld_gemaps_dbp(float *x, int64_t ld, int n_utt, const int64_t *row_off) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_utt) return;
  if (row_off[u + 1] - row_off[u] <= 0) return;
  const float factor = (float)(10.0 / log(10.0)), logfloor = (float)0.000000000001;
  const float v = x[u * ld];
  x[u * ld] = factor * glibc_logf(v > logfloor ? v : logfloor);
}

// ------------------------------------------------------------------------------------------------ per-component: cSpectral
// The frames of ONE stream in order (the flux needs the previous frame's magnitudes; `state` carries them across calls,
// like smilehip_spectral_frames). One wave.
__global__ void __launch_bounds__(64) lld_gemaps_spectral_rows(const float *src, int64_t lds, float *state, int first, float *dst,
                                                              int64_t ldd, int64_t nF, int K, GemapsParams G) {
  __shared__ __attribute__((aligned(16))) float mg[516], pw[516], prev[516], lg[64];     // K <= 513 (20 ms frames up to 48 kHz)
  const int lane = threadIdx.x;
  for (int k = lane; k < K; k += 64) prev[k] = first ? 0.0f : state[k];
  WaveG::sync();
  for (int64_t f = 0; f < nF; ++f) {
    const float *m = src + f * lds;
    for (int k = lane; k < K; k += 64) { const float v = m[k]; mg[k] = v; pw[k] = v * v; }
    WaveG::sync();
    gemaps_spectral_wave(mg, pw, prev, first && f == 0, lg, G, K, lane, dst + f * ldd);
    WaveG::sync();
    for (int k = lane; k < K; k += 64) prev[k] = mg[k];
    WaveG::sync();
  }
  for (int k = lane; k < K; k += 64) state[k] = prev[k];
}

// ------------------------------------------------------------------------------------------------ launchers
hipError_t launch_gemaps_spectral_rows(const float *src, int64_t lds, float *state, bool first, float *dst, int64_t ldd, int64_t nF,
                                       int K, const GemapsParams &G, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  if (K > 516) return hipErrorInvalidValue;
  SMILEHIP_KLAUNCH(lld_gemaps_spectral_rows, dim3(1), dim3(64), 0, s, src, lds, state, first ? 1 : 0, dst, ldd, nF, K, G);
  return hipGetLastError();
}

// op_mode 1 (cSpecResample rows) / 2 (cLpc rows) of the resampling kernel; cFormantLpc rows
hipError_t launch_gemaps_lpc_rows(const GemapsParams &G, hipStream_t s) {
  if (G.op_rows <= 0) return hipSuccess;
  SMILEHIP_KLAUNCH(lld_gemaps_lpc, dim3((unsigned)((G.op_rows + kLpcTile - 1) / kLpcTile)), dim3(256), 0, s, G);
  return hipGetLastError();
}
static hipError_t launch_formants(const GemapsParams &G, int64_t rows, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  SMILEHIP_KLAUNCH(lld_gemaps_formants, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, G);
  return hipGetLastError();
}
hipError_t launch_gemaps_formant_rows(const GemapsParams &G, hipStream_t s) {
  if (G.op_rows <= 0) return hipSuccess;
  return launch_formants(G, G.op_rows, s);
}

hipError_t launch_gemaps_frames(const LldParams &P, const GemapsParams &G, int n_runs, hipStream_t s) {
  if (n_runs <= 0) return hipSuccess;
  if ((P.Nfft != 256 && P.Nfft != 512 && P.Nfft != 1024) || P.N > P.Nfft || P.n_mfcc > 16 || P.n_bands > 32) return hipErrorInvalidValue;   // 20 ms at 8 .. 48 kHz
  const int M = P.Nfft / 2;
  const int Kpad = (P.K + 3) & ~3;
  const size_t lds = sizeof(float) * (size_t)(Kpad + 128 + 16 * 32 + oo_table_floats(P.oo) + 4 * ((2 * fft_pairs(M) > 2 * Kpad ? 2 * fft_pairs(M) : 2 * Kpad) + 3 * Kpad + 64 + 96));
  // sixteen lanes per frame for the shipped geometry (lld_gemaps_quad.hpp); SMILEHIP_GEMAPS_WAVE=1: the wave-per-frame form (A/B switch)
  const bool quad_ok = P.oo.tw && P.N == 320 && P.H == 160 && P.Nfft == 512 && (P.pad_left == 0 || P.pad_left == 96) && P.K == 257 && P.n_bands == 26 && P.n_mfcc == 4 &&
                       P.pcm && !P.pcm_f32 && P.total_frames < (int64_t(1) << 31) && G.sl_iL[0] >= 0 && G.sl_iR[0] <= 63 && G.sl_iL[1] >= 0 &&
                       G.sl_iR[1] <= 63 && G.ar_n1 >= 0 && G.ar_n1 <= G.ar_n2 && G.ar_n2 <= 257 && G.rng_lo >= 0 && !getenv("SMILEHIP_GEMAPS_WAVE");
  if (quad_ok) {
    const size_t qlds = sizeof(float) * gemaps_quad_lds_floats(P.oo);
    const void *qfn = P.pad_left ? reinterpret_cast<const void *>(&lld_gemaps_frame20_quad<96>) : reinterpret_cast<const void *>(&lld_gemaps_frame20_quad<0>);
    hipError_t eq = hipFuncSetAttribute(qfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qlds);
    if (eq != hipSuccess) return eq;
    const int per_wg = kGmQuadWaves * 4;
    const dim3 qgrid((unsigned)((n_runs + per_wg - 1) / per_wg)), qblock(kGmQuadWaves * 64);
    if (P.pad_left) SMILEHIP_KLAUNCH(lld_gemaps_frame20_quad<96>, qgrid, qblock, qlds, s, P, G, n_runs);
    else SMILEHIP_KLAUNCH(lld_gemaps_frame20_quad<0>, qgrid, qblock, qlds, s, P, G, n_runs);
  } else {
    SMILEHIP_KLAUNCH(lld_gemaps_frame20, dim3((unsigned)((n_runs + 3) / 4)), dim3(256), lds, s, P, G, n_runs);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (G.total_frames20 <= 0) return hipSuccess;
  SMILEHIP_KLAUNCH(lld_gemaps_lpc, dim3((unsigned)((G.total_frames20 + kLpcTile - 1) / kLpcTile)), dim3(256), 0, s, G);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_formants(G, G.total_frames20, s);
}

namespace {
template <int LOGM>
hipError_t launch_harm_g(const LldParams &P, const F0Params &Q, const GemapsParams &G, int64_t n_tiles, int max_blocks, hipStream_t s) {
  using HG = HarmG<LOGM>;
  const int NP = (Q.N + 3) & ~3;
  if (sizeof(float) * (size_t)oo_table_floats(Q.oo) > HG::kTwBytes) return hipErrorInvalidValue;
  if (LOGM != 9 && !Q.oo.tw) return hipErrorInvalidValue;             // the own-order A/B build exists for FFT 1024 only
  const size_t lds = sizeof(float) * (size_t)NP + HG::kTwBytes + sizeof(float) * HG::kWaves * (size_t)(2 * HG::kZ + 2 * HG::kHKP);
  const void *fn = reinterpret_cast<const void *>(&lld_gemaps_harm<LOGM>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  unsigned grid = (unsigned)((n_tiles + HG::kWaves - 1) / HG::kWaves);
  if (grid > (unsigned)(2 * max_blocks)) grid = (unsigned)(2 * max_blocks);   // M = 512: 78 KB of LDS per workgroup, two per CU
  // (the tile counter starts from zero at every launch, whatever an earlier launch left behind)
  if (G.harm_ctl && (e = hipMemsetAsync(G.harm_ctl, 0, 2 * sizeof(int32_t), s)) != hipSuccess) return e;
  SMILEHIP_KLAUNCH(lld_gemaps_harm<LOGM>, dim3(grid), dim3(HG::kWaves * 64), lds, s, P, Q, G);
  return hipGetLastError();
}
}  // namespace
hipError_t launch_gemaps_harm(const LldParams &P, const F0Params &Q, const GemapsParams &G, int max_blocks, hipStream_t s) {
  const int64_t n_tiles = G.op_mode == 1 ? (G.op_rows + 7) / 8 : G.n_tiles60;
  if (n_tiles <= 0) return hipSuccess;
  if (Q.K != Q.Nfft / 2 + 1) return hipErrorInvalidValue;
  switch (Q.Nfft) {                                      // 60 ms frames at 8 .. 48 kHz
    case 512: return launch_harm_g<8>(P, Q, G, n_tiles, max_blocks, s);
    case 1024: return launch_harm_g<9>(P, Q, G, n_tiles, max_blocks, s);
    case 2048: return launch_harm_g<10>(P, Q, G, n_tiles, max_blocks, s);
    case 4096: return launch_harm_g<11>(P, Q, G, n_tiles, max_blocks, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemaps_tail(const int64_t *d_frame_off20, const int64_t *d_row_off, int n_utt, const GemapsParams &G, float *d_out,
                              int64_t ld_out, hipStream_t s) {
  if (n_utt <= 0) return hipSuccess;
  SMILEHIP_KLAUNCH(lld_gemaps_tail, dim3((unsigned)n_utt), dim3(256), 0, s, d_frame_off20, d_row_off, n_utt, G, d_out, ld_out);
  return hipGetLastError();
}

hipError_t launch_gemaps_dbp(float *d_x, int64_t ld, int n_utt, const int64_t *d_row_off, hipStream_t s) {
  if (n_utt <= 0) return hipSuccess;
  SMILEHIP_KLAUNCH(lld_gemaps_dbp, dim3((unsigned)((n_utt + 255) / 256)), dim3(256), 0, s, d_x, ld, n_utt, d_row_off);
  return hipGetLastError();
}

int gemaps_lpc_tile() { return kLpcTile; }

}  // namespace smilehip
