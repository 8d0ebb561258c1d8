// ComParE_2016 LLD groups A and B on the hot path (SURVEY.md 8a rows R8 cPlp as auditory
// spectrum incl. RASTA, R11 cSpectral with ComParE's option set, R12 cEnergy / cMZcr on the
// 20 ms / 60 ms frames), reference-order kernels (parity first, not tuned):
//   lld_compare_frame   one workgroup per run of 8 consecutive 20 ms frames: window, FFT,
//                       magnitude; mel (one bank, two scalings) -> audspec + its mean,
//                       MFCC 1..14; the 15 spectral descriptors, each evaluated by one lane
//                       in the reference's own loop order (four waves work on different
//                       descriptors concurrently); RMS energy of the raw frame; ZCR of the
//                       60 ms frame; the un-filtered mel spectrum goes to scratch for RASTA
//   lld_compare_rasta   one wave per utterance: log -> RASTA IIR over frames (sequential by
//                       nature) -> equal loudness + compression in the log domain -> exp,
//                       and the mean over bands
//   lld_compare_groupA  SMA + delta of group A, whose levels have different lengths
// Group B's SMA + delta run through the generic window chain (lld_kernels.hip).
#include <hip/hip_runtime.h>
#include "kernel_timing.hpp"

#include <cstdlib>

#include "lld_blocks.hpp"
#include "lld_fft.hpp"
#include "lld_blocks_compare.hpp"
#include "lld_compare_quad.hpp"
#include "lld_device.hpp"
#include "lld_launch.hpp"
#include "lld_params.hpp"

namespace smilehip {

namespace {
constexpr int kRun = 8;          // frames per run when the caller names none (the flux needs the previous frame's magnitudes)

}  // namespace

// Development instrumentation (-DSMILEHIP_PHASE_TIMING in a private build, tools/ubench): s_memtime at the phase
// boundaries of the frame loop, summed over all workgroups by thread 0. Not compiled into the product.
#ifdef SMILEHIP_PHASE_TIMING
__device__ unsigned long long g_phase_cmp[16];
#define PHASE_DECL unsigned long long ph_acc[8] = {0}; unsigned long long ph_last = __builtin_amdgcn_s_memtime();
#define PHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += t_ - ph_last; ph_last = t_; } while (0)
#define PHASE_FLUSH do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_phase_cmp[i_], ph_acc[i_]); } while (0)
extern "C" int smilehip_debug_phase_cmp(unsigned long long *out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cmp), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cmp), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define PHASE_DECL
#define PHASE(i)
#define PHASE_FLUSH
#endif

// LDS: yv[N] | re[M] | im[M] | mg[K] | pw[K] | prev[K] | mel[32] | aud[32] | lmel[32] | double red[64] | double cum[256] | peaks
// Nfft = 512 only: thread i of the 256 owns bin i+1 in the descriptor section.
__global__ void __launch_bounds__(256) lld_compare_frame(LldParams P, CompareParams Q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int M = P.Nfft >> 1, K = P.K;
  const int Npad = (P.N + 3) & ~3, Kpad = (K + 3) & ~3;
  float *yv = smem;
  float *re = yv + Npad;
  float *im = re + M;
  float *mg = im + M;
  float *pw = mg + Kpad;
  float *prev = pw + Kpad;
  float *melv = prev + Kpad;
  float *aud = melv + 32;
  float *lmel = aud + 32;
  double *red = reinterpret_cast<double *>(lmel + 32);
  double *cum = red + 64;
  float *pk_val = reinterpret_cast<float *>(cum + 256);   // [4] last peak of each wave
  int *pk_has = reinterpret_cast<int *>(pk_val + 4);      // [4]
  // tables of the mel / MFCC section, staged once per workgroup (the few threads that walk them would otherwise wait
  // on global memory at every step): mel weights [Kpad], band ranges [4 x 32], DCT rows [16 x 32]
  float *s_coef = reinterpret_cast<float *>(pk_has + 4);
  int32_t *s_rng = reinterpret_cast<int32_t *>(s_coef + Kpad);
  float *s_dct = reinterpret_cast<float *>(s_rng + 128);
  for (int i = threadIdx.x; i < K; i += blockDim.x) s_coef[i] = P.mel_coef[i];
  for (int i = threadIdx.x; i < 4 * P.n_bands; i += blockDim.x) s_rng[i] = P.mel_rng[i];
  for (int i = threadIdx.x; i < P.n_mfcc * P.n_bands; i += blockDim.x) s_dct[i] = P.dct_rows[i];
  int logM = 0;
  while ((1 << logM) < M) ++logM;

  const int u = Q.run_utt[blockIdx.x];
  const int t0 = Q.run_t0[blockIdx.x];
  const int64_t f0 = P.frame_off[u];
  const int T20 = (int)(P.frame_off[u + 1] - f0);
  const int64_t s_utt = P.samp_off[u];
  const int64_t utt_len = P.samp_off[u + 1] - s_utt;
  const int T60 = (utt_len >= Q.N60) ? (int)((utt_len - Q.N60) / P.H + 1) : 0;
  const PcmIn xu = pcm_in(P) + s_utt;
  SpectralConsts SC;
  SC.fsSec = Q.fsSec;
  SC.sharp_w = Q.sharp_w;
  for (int i = 0; i < 2; ++i) {
    SC.band_iL[i] = Q.band_iL[i]; SC.band_iR[i] = Q.band_iR[i];
    SC.band_wL[i] = Q.band_wL[i]; SC.band_wR[i] = Q.band_wR[i];
  }
  SC.slope_Sf = Q.slope_Sf;
  SC.slope_S2f = Q.slope_S2f;
  SC.log_tab = nullptr;
  const int run_len = Q.run_frames > 0 ? Q.run_frames : kRun;
  const int t_last = (t0 + run_len < T20) ? t0 + run_len : T20;
  PHASE_DECL

  // frames t0-1 (magnitudes only, for the flux) .. t_last-1
  for (int t = (t0 > 0 ? t0 - 1 : 0); t < t_last; ++t) {
    const bool warm = t < t0;
    const PcmIn x = xu + (int64_t)t * P.H;
    float *rawA = Q.rawA + (f0 + t) * 4;
    float *rawB = Q.rawB + (f0 + t) * 55;
    // R3 (no pre-emphasis in this chain) + R12 energy of the RAW frame ([is13_energy] reads is13_frame25)
    for (int n = threadIdx.x; n < P.N; n += blockDim.x) yv[n] = x[n];
    __syncthreads();
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
      const int n0 = 2 * i - P.pad_left, n1 = n0 + 1;
      const float v0 = (n0 >= 0 && n0 < P.N) ? yv[n0] * P.window[n0] + P.win_offset : 0.0f;
      const float v1 = (n1 >= 0 && n1 < P.N) ? yv[n1] * P.window[n1] + P.win_offset : 0.0f;
      if (P.oo.tw) {
        reinterpret_cast<float2 *>(re)[i] = make_float2(v0, v1);
      } else {
        const int r = (int)(__brev((unsigned)i) >> (32 - logM));
        re[r] = v0;
        im[r] = v1;
      }
    }
    __syncthreads();
    PHASE(0);   // load + window
    if (P.oo.tw) ooura_levels<BlockG, false>(reinterpret_cast<float2 *>(re), P.oo);   // the reference's rdft network
    else block_cfft_radix2(re, im, M, P.tw_half);
    PHASE(1);   // FFT
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
      const float2 X = P.oo.tw ? ooura_bin(reinterpret_cast<const float2 *>(re), P.oo, k) : untangle_bin(re, im, M, k, P.tw_full);
      const float m = bin_magnitude(X, k == 0 || k == M);
      mg[k] = m;
      pw[k] = m * m;                                    // squareInput (spectral.cpp:676-683) == melspec usePower
    }
    __syncthreads();
    PHASE(2);   // magnitudes
    if (warm) {
      for (int k = threadIdx.x; k < K; k += blockDim.x) prev[k] = mg[k];
      __syncthreads();
      continue;
    }
    // R6 once, two scalings: [is13_melspec1] (htk=0) feeds cPlp, [is13_melspecMfcc] (htk=1) feeds cMfcc
    for (int b = threadIdx.x; b < P.n_bands; b += blockDim.x) {
      const float acc = mel_band_exact(pw, s_coef, s_rng, b, 1.0f);
      melv[b] = acc;
      // log mel spectrum for the RASTA pass (doLog, plp.cpp:434-439; double log = correctly rounded logf)
      Q.mel1[(f0 + t) * 26 + b] = glibc_logf(acc < Q.plp_melfloor ? Q.plp_melfloor : acc);   // plp.cpp:434-439: logf
      lmel[b] = log_mel(acc * P.mel_scale, P.melfloor, P.log_floor);
      // R8 cPlp without RASTA: melfloor, equal loudness, power-law compression (plp.cpp:499-507)
      aud[b] = plp_aud_band(acc, Q.plp_melfloor, Q.eql[b], Q.compression);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < P.n_mfcc; r += blockDim.x)
      rawB[41 + r] = dct_coeff(lmel, s_dct + r * P.n_bands, P.n_bands, P.dct_gain[r]);   // R7
    if (threadIdx.x == 192) {                           // cVectorOperation ll1, vectorOperation.cpp:475-481
      const float d = seq_sum_f32(aud, P.n_bands);
      rawA[0] = d / (float)P.n_bands;
    }
    PHASE(3);   // mel, auditory spectrum, MFCC
    // R12: energy of the raw 20 ms frame (energy.cpp:152-168) and ZCR of the 60 ms frame (mzcr.cpp:117-124)
    {
      const int tid = threadIdx.x;
      double v0[2] = {0.0, 0.0};
      for (int n = tid; n < P.N; n += 256) { const float tmp = yv[n]; v0[0] += tmp * tmp; }
      if (t < T60)
        for (int i = 1 + tid; i < Q.N60 - 1; i += 256) {
          const float a = x[i - 1], b = x[i], c = x[i + 1];
          if (((a * c <= 0.0f) && (b == 0.0f)) || (a * b < 0.0f)) v0[1] += 1.0;
        }
      block_sum_n<2>(v0, red);
      if (tid == 0) {
        rawA[2] = (float)sqrt(v0[0] / (float)P.N) * 1.0f + 0.0f;
        if (t < T60) rawA[3] = (float)v0[1] / (float)Q.N60;
      }
    }
    PHASE(4);   // energy + ZCR
    // R11: the 15 spectral descriptors, block-parallel (lld_blocks_compare.hpp)
    spectral_frame(mg, pw, prev, t == 0, SC, K, red, cum, pk_val, pk_has, rawB + 26);
    for (int k = threadIdx.x; k < K; k += blockDim.x) prev[k] = mg[k];
    __syncthreads();
    PHASE(5);   // spectral descriptors
  }
  PHASE_FLUSH;
}

// The same frame pipeline with ONE wave per frame and four independent runs per workgroup: no workgroup barriers after
// the table staging, the thinly parallel sections (26 mel bands, 14 cepstra, scalar tails) of four frames overlap. Every sum
// keeps the block kernel's order (lld_blocks_compare.hpp), so the two kernels give bit-identical rows.
// LDS: shared coef[Kpad] | rng[128] | dct[16 x 32]; per wave z[fft_pairs(M)] pairs | mg[Kpad] | pw[Kpad] | prev[Kpad] |
// mel[32] | aud[32] | lmel[32]
// the wave's z region: the transform's pairs, then the mel terms (two rows of Kpad floats), then the descriptors' chains
__host__ __device__ inline int compare_z_floats(int M, int Kpad) { return 2 * fft_pairs(M) > 2 * Kpad ? 2 * fft_pairs(M) : 2 * Kpad; }
// TUNED: the shipped geometry -- 16 kHz, 20 ms frames (N = 320, hop 160, FFT 512, symmetric padding: 96 zeros in front), the 60 ms
// window of the zero-crossing rate 960 samples, reference-order transform tables present -- as compile-time facts: the range tests of
// the loads and the own-order transform's code path go, the staged tables are LDS pointers on every path (no flat loads)
template <int W, bool TUNED = false>                   // K = 64 W + 1 bins: W = 4 at 16 kHz (FFT 512), 2 / 8 for FFT 256 / 1024
__device__ __forceinline__ void compare_frame_wave_body(const LldParams &P, const CompareParams &Q_in, int n_runs, float *smem) {
  struct Geo { int N, H, pad_left, N60; };
  const Geo Gq = TUNED ? Geo{320, 160, 96, 960} : Geo{P.N, P.H, P.pad_left, Q_in.N60};
  const CompareParams &Q = Q_in;
  const int M = TUNED ? 256 : (P.Nfft >> 1), K = TUNED ? 257 : P.K;
  const int Kpad = (K + 3) & ~3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double2 *s_log = reinterpret_cast<double2 *>(smem);     // the logarithm's table (first: 16-byte aligned), 512 floats
  float *s_coef = smem + 512;
  int32_t *s_rng = reinterpret_cast<int32_t *>(s_coef + Kpad);
  float *s_dct = reinterpret_cast<float *>(s_rng + 128);
  for (int i = threadIdx.x; i < 128; i += blockDim.x) s_log[i] = kLogTab[i];
  for (int i = threadIdx.x; i < K; i += blockDim.x) s_coef[i] = P.mel_coef[i];
  for (int i = threadIdx.x; i < 4 * P.n_bands; i += blockDim.x) s_rng[i] = P.mel_rng[i];
  for (int i = threadIdx.x; i < P.n_mfcc * P.n_bands; i += blockDim.x) s_dct[i] = P.dct_rows[i];
  const OouraTab OO = oo_stage_tables<TUNED>(P.oo, s_dct + 16 * 32, threadIdx.x, blockDim.x);   // reference-order FFT tables (or none)
  __syncthreads();                                       // the only workgroup barrier
  const int run = blockIdx.x * 4 + wave;
  if (run >= n_runs) return;
  const int rawpad = (Gq.N60 + 3) & ~3;
  const int per_wave = compare_z_floats(M, Kpad) + 3 * Kpad + 96 + rawpad;
  float2 *z = reinterpret_cast<float2 *>(s_dct + 16 * 32 + oo_table_floats(P.oo) + wave * per_wave);   // the transform's (re, im) pairs
  const int zpad = fft_pad(M);
  float *mg = reinterpret_cast<float *>(z) + compare_z_floats(M, Kpad);
  float *yv = mg;                                        // the raw frame lives in mg | pw (N <= 2 M < 2 Kpad) until the transform has read it
  float *pw = mg + Kpad;
  float *prev = pw + Kpad;
  float *melv = prev + Kpad;
  float *aud = melv + 32;
  float *lmel = aud + 32;
  // the frame's raw samples: the 60 ms span the zero-crossing rate covers (its first N samples are the 20 ms frame), staged ONCE per frame
  // with coalesced loads. (Round 3, phase timing: the energy + ZCR section read every sample three times from global memory through
  // the int16 / float accessor -- 45 dependent loads per lane and frame, 28 % of the kernel.)
  float *raw = lmel + 32;
  (void)yv;
  int logM = 0;
  while ((1 << logM) < M) ++logM;

  const int u = Q.run_utt[run];
  const int t0 = Q.run_t0[run];
  const int64_t f0 = P.frame_off[u];
  const int T20 = (int)(P.frame_off[u + 1] - f0);
  const int64_t s_utt = P.samp_off[u];
  const int64_t utt_len = P.samp_off[u + 1] - s_utt;
  const int T60 = (utt_len >= Gq.N60) ? (int)((utt_len - Gq.N60) / Gq.H + 1) : 0;
  const PcmIn xu = pcm_in(P) + s_utt;
  SpectralConsts SC;
  SC.fsSec = Q.fsSec;
  SC.sharp_w = Q.sharp_w;
  for (int i = 0; i < 2; ++i) {
    SC.band_iL[i] = Q.band_iL[i]; SC.band_iR[i] = Q.band_iR[i];
    SC.band_wL[i] = Q.band_wL[i]; SC.band_wR[i] = Q.band_wR[i];
  }
  SC.slope_Sf = Q.slope_Sf;
  SC.slope_S2f = Q.slope_S2f;
  SC.log_tab = s_log;
  const int run_len = Q.run_frames > 0 ? Q.run_frames : kRun;
  const int t_last = (t0 + run_len < T20) ? t0 + run_len : T20;
  const int lane_in = lane;
  PHASE_DECL
  const int t_begin = t0 > 0 ? t0 - 1 : 0;
  int rbase = 0;
  int zc_count = 0, zc_front = 0;                        // this lane's crossings inside the window / inside its first hop
  bool zc_have = false;
  for (int t = t_begin; t < t_last; ++t) {
    // an opaque copy of the lane index per frame: otherwise everything below that depends on the lane only (bit-reversed
    // FFT addresses, table addresses, range tests) is hoisted out of the frame loop and kept in ~150 VGPRs across it --
    // two waves per SIMD instead of four
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const bool warm = t < t0;
    const PcmIn x = xu + (int64_t)t * Gq.H;
    float *rawA = Q.rawA + (f0 + t) * 4;
    float *rawB = Q.rawB + (f0 + t) * 55;
    // (asking for frame t + 1's samples here, a frame ahead, was measured: -2 % for this kernel alone, +24 % for the whole
    // ComParE level of a small batch, where the kernel runs beside the jitter pass -- not kept)
    // `raw` is a ring over the 60 ms window: consecutive frames of the run share all but one hop of it, so only the H new samples are
    // loaded (the whole window at the run's first frame); sample n of THIS frame's window sits at raw[(n + rbase) mod N60]. Samples past
    // the end of the utterance (the last frames, where only the 20 ms frame still exists) are not read.
    const int64_t left = utt_len - (int64_t)t * Gq.H;
    const int lim = left < (int64_t)Gq.N60 ? (int)left : Gq.N60;
    if (t == t_begin) {
      rbase = 0;
      for (int n = lane; n < lim; n += 64) raw[n] = x[n];
    } else {
      rbase += Gq.H;
      if (rbase >= Gq.N60) rbase -= Gq.N60;
      for (int n = Gq.N60 - Gq.H + lane; n < lim; n += 64) { int k = n + rbase; if (k >= Gq.N60) k -= Gq.N60; raw[k] = x[n]; }
    }
    const auto R = [&](int n) { int k = n + rbase; if (k >= Gq.N60) k -= Gq.N60; return raw[k]; };
    WaveG::sync();
    PHASE(0);
    const auto load_pair = [&](int i) {
      const int n0 = 2 * i - Gq.pad_left, n1 = n0 + 1;
      return make_float2((n0 >= 0 && n0 < Gq.N) ? R(n0) * P.window[n0] + P.win_offset : 0.0f,
                         (n1 >= 0 && n1 < Gq.N) ? R(n1) * P.window[n1] + P.win_offset : 0.0f);
    };
    if (TUNED || OO.tw) oo_wave_forward<TUNED ? 256 : 0>(z, OO, lane, load_pair);  // the reference's rdft network, register form (lld_ooura_wave.hpp)
    else wave_cfft(z, M, P.tw_half, lane, load_pair);
    PHASE(1);
    for (int k = lane; k <= M; k += 64) {
      const float m = bin_magnitude((TUNED || OO.tw) ? oo_wave_bin<TUNED ? 256 : 0>(z, OO, k) : wave_untangle(z, M, zpad, k, P.tw_full), k == 0 || k == M);
      mg[k] = m;
      pw[k] = m * m;
    }
    WaveG::sync();
    PHASE(2);
    if (warm) {
      for (int k = lane; k < K; k += 64) prev[k] = mg[k];
      WaveG::sync();
      continue;
    }
    float *mt_a = reinterpret_cast<float *>(z), *mt_r = mt_a + Kpad;      // (z: free between the transform and the descriptors' chains)
    mel_terms_fill<WaveG>(pw, s_coef, K, mt_a, mt_r);
    WaveG::sync();
    if (lane < P.n_bands) {
      const int b = lane;
      const float acc = mel_band_from_terms(mt_a, mt_r, s_rng, b, 1.0f);
      melv[b] = acc;
      Q.mel1[(f0 + t) * 26 + b] = glibc_logf(acc < Q.plp_melfloor ? Q.plp_melfloor : acc);   // plp.cpp:434-439: logf
      lmel[b] = log_mel(acc * P.mel_scale, P.melfloor, P.log_floor);
      aud[b] = plp_aud_band(acc, Q.plp_melfloor, Q.eql[b], Q.compression);
    }
    WaveG::sync();
    if (lane < P.n_mfcc) rawB[41 + lane] = dct_coeff(lmel, s_dct + lane * P.n_bands, P.n_bands, P.dct_gain[lane]);   // R7
    if (lane == 32) {                                    // cVectorOperation ll1, vectorOperation.cpp:475-481
      const float d = seq_sum_f32(aud, P.n_bands);
      rawA[0] = d / (float)P.n_bands;
    }
    PHASE(3);
    // R12 in the block kernel's summation order: lane l holds the partial sums of its threads l, l+64, l+128, l+192
    {
      double v0[4][2];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int tid = lane + 64 * w;
        v0[w][0] = 0.0; v0[w][1] = 0.0;
        for (int n = tid; n < Gq.N; n += 256) { const float tmp = R(n); v0[w][0] += tmp * tmp; }
      }
      // Zero crossings of the 60 ms window (mzcr.cpp:117-124): a count, so any order of the positions gives the reference's sum.
      // Consecutive frames of the run share all but one hop of the window: the lane that owns the ABSOLUTE position p (p mod 64)
      // keeps its count, subtracts the crossings of the hop that leaves (the window's first H positions, counted one frame
      // earlier from that frame's own window: zc_front) and adds those of the hop that enters (the last H positions).
      if (t < T60) {
        const auto cross = [&](int i) {
          const float a = R(i - 1), b = R(i), c = R(i + 1);
          return (((a * c <= 0.0f) && (b == 0.0f)) || (a * b < 0.0f)) ? 1 : 0;
        };
        const auto count = [&](int lo, int hi) {           // positions lo .. hi (relative to the frame) that this lane owns
          const int off = (int)(((int64_t)t * Gq.H) & 63);
          int c = 0;
          for (int i = lo + ((lane - off - lo) & 63); i <= hi; i += 64) c += cross(i);
          return c;
        };
        const int last = Gq.N60 - 2;
        if (zc_have && 2 * Gq.H < last) zc_count += count(last - Gq.H + 1, last) - zc_front;
        else zc_count = count(1, last);
        zc_front = count(1, Gq.H);
        zc_have = true;
        v0[0][1] = (double)zc_count;
      }
      double tot[2];
      wave_sum4<2>(v0, tot);
      if (lane == 0) {
        rawA[2] = (float)sqrt(tot[0] / (float)Gq.N) * 1.0f + 0.0f;
        if (t < T60) rawA[3] = (float)tot[1] / (float)Gq.N60;
      }
    }
    PHASE(4);
    spectral_frame_wave<W>(mg, pw, prev, t == 0, SC, K, reinterpret_cast<float *>(z), rawB + 26);   // z: free between two transforms
    WaveG::sync();
    for (int k = lane; k < K; k += 64) prev[k] = mg[k];
    WaveG::sync();
    PHASE(5);
  }
  PHASE_FLUSH;
}

// Two builds of the same body. Three waves per SIMD (167 VGPRs, 4 spilled): ComParE A+B alone 12.8 -> 10.1 ms per 1000 x 10 s,
// one GPU's share of config 4 514 -> 473 ms. Beside the jitter pass of a SMALL batch (the whole level, fewer utterances than the
// device has wave slots) the denser kernel takes the slots the jitter chains need and the level gets slower (44.7 -> 47.8 ms per
// 1000 x 10 s): there the two-wave build runs.
__global__ void __launch_bounds__(256) lld_compare_frame_wave(LldParams P, CompareParams Q, int n_runs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  compare_frame_wave_body<4>(P, Q, n_runs, smem);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) lld_compare_frame_wave3(LldParams P, CompareParams Q, int n_runs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  compare_frame_wave_body<4>(P, Q, n_runs, smem);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) lld_compare_frame_wave3t(LldParams P, CompareParams Q, int n_runs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  compare_frame_wave_body<4, true>(P, Q, n_runs, smem);
}
// Sixteen lanes per frame, four runs per wave (lld_compare_quad.hpp): the shipped geometry. Workgroups of four waves; LDS:
// the tables once per workgroup, 2.4 KB per run -- three workgroups per CU.
namespace {
constexpr int kCmpQuadWaves = 4;
inline size_t compare_quad_lds_floats(const OouraTab &oo) {
  return (size_t)cq::kTableFloats + (size_t)((oo_table_floats(oo) + 3) & ~3) + (size_t)kCmpQuadWaves * 4 * cq::kRowFloats;
}
}  // namespace
__global__ void __launch_bounds__(kCmpQuadWaves * 64) __attribute__((amdgpu_waves_per_eu(CQ_WAVES, CQ_WAVES))) lld_compare_frame_quad(LldParams P, CompareParams Q, int n_runs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double2 *s_log = reinterpret_cast<double2 *>(smem);
  double *s_sharp = reinterpret_cast<double *>(smem + 512);
  float *s_win = smem + 1024;
  float *s_coef = s_win + cq::kN;
  int32_t *s_rng = reinterpret_cast<int32_t *>(s_coef + 260);
  float *s_dct = reinterpret_cast<float *>(s_rng + 128);
  constexpr int NT = kCmpQuadWaves * 64;
  for (int i = threadIdx.x; i < 128; i += NT) s_log[i] = kLogTab[i];
  for (int i = threadIdx.x; i < cq::kK - 1; i += NT) s_sharp[i] = Q.sharp_w[i];
  for (int i = threadIdx.x; i < cq::kN; i += NT) s_win[i] = P.window[i];
  for (int i = threadIdx.x; i < cq::kK; i += NT) s_coef[i] = P.mel_coef[i];
  for (int i = threadIdx.x; i < 4 * cq::kBands; i += NT) s_rng[i] = P.mel_rng[i];
  for (int i = threadIdx.x; i < cq::kMfcc * cq::kBands; i += NT) s_dct[i] = P.dct_rows[i];
  const OouraTab OO = oo_stage_tables<true>(P.oo, smem + cq::kTableFloats, threadIdx.x, NT);
  __syncthreads();                                       // the only workgroup barrier
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int first_run = (blockIdx.x * kCmpQuadWaves + wave) * 4;
  if (first_run >= n_runs) return;
  float *fmem = smem + cq::kTableFloats + ((oo_table_floats(P.oo) + 3) & ~3) + wave * 4 * cq::kRowFloats;
  compare_frame_quad_body(P, Q, n_runs, first_run, s_win, s_coef, s_rng, s_dct, s_log, s_sharp, OO, fmem);
}

// the other spectrum sizes of 20 ms frames: FFT 256 (8 / 11.025 kHz), FFT 1024 (32 / 44.1 / 48 kHz)
template <int W>
__global__ void __launch_bounds__(256) lld_compare_frame_wave_g(LldParams P, CompareParams Q, int n_runs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  compare_frame_wave_body<W>(P, Q, n_runs, smem);
}

// R8 with newRASTA (plp.cpp:434-439, 468-485, 490-497, 512-517): log -> 4-tap FIR + 1-pole IIR
// per band over the frames of an utterance -> + log equal loudness, x compression -> exp.
// One wave per utterance, lane = band. Writes audSpec_Rfilt[26] (group B) and its ll1 mean (group A).
__global__ void __launch_bounds__(64) lld_compare_rasta(const int64_t *frame_off, int n_utt, CompareParams Q) {
  const int u = blockIdx.x;
  if (u >= n_utt) return;
  const int b = threadIdx.x;
  const int bb = b < 26 ? b : 25;
  const int64_t f0 = frame_off[u];
  const int T = (int)(frame_off[u + 1] - f0);
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  int init = 0;
  constexpr int kDepth = 8;                 // frames loaded ahead of the recursion
  __shared__ float row[kDepth][32];
  const float eq = Q.eql_log[bb];
  for (int tc = 0; tc < T; tc += kDepth) {
    float xs[kDepth];
#pragma unroll
    for (int i = 0; i < kDepth; ++i) {
      const int t = (tc + i < T) ? tc + i : T - 1;
      xs[i] = Q.mel1[(f0 + t) * 26 + bb];
    }
#pragma unroll
    for (int i = 0; i < kDepth; ++i) {
      float x = xs[i];
      const float out = Q.rasta_fir[0] * x + b0;
      b0 = Q.rasta_fir[1] * x + b1 + (float)(init >= 5) * Q.rasta_iir * out;
      b1 = Q.rasta_fir[2] * x + b2;
      b2 = Q.rasta_fir[3] * x + b3;
      b3 = Q.rasta_fir[4] * x;
      x = (init >= 5) ? out : 0.0f;
      if (init < 5) init++;
      x += eq;
      x *= Q.compression;
      const float y = glibc_expf(x);                      // plp.cpp:512-517: exp() on a float is expf
      if (b < 26) {
        row[i][b] = y;
        if (tc + i < T) Q.rawB[(f0 + tc + i) * 55 + b] = y;
      }
    }
    __syncthreads();
    if (b < kDepth && tc + b < T) {         // ll1 of each frame of the chunk, one lane per frame
      float d = 0.0f;
      for (int i = 0; i < 26; i++) d += row[b][i];
      Q.rawA[(f0 + tc + b) * 4 + 1] = d / 26.0f;
    }
    __syncthreads();
  }
}

// Group A: [is13_smoA] reads levels of lengths (T20, T20, T20, T60); a multi-level reader serves
// a block only if every level can, and each level pads with its own last frame
// (dataReader.cpp:446-522, dataMemoryLevel.cpp:1699-1708): T60+1 SMA frames; [is13_deA] is the
// delta of that level. One thread per (output row, column).
__device__ __forceinline__ float sma_a(const float *la, int t, int d, int len) {
  int i0 = t, im = t - 1, ip = t + 1;
  if (im < 0) im = 0;
  if (i0 > len - 1) i0 = len - 1;
  if (im > len - 1) im = len - 1;
  if (ip > len - 1) ip = len - 1;
  float y = la[i0 * 4 + d];
  y += la[im * 4 + d];
  y += la[ip * 4 + d];
  return y / 3.0f;
}
__global__ void __launch_bounds__(256) lld_compare_groupA(const int64_t *frame_off, const int64_t *row_off, int n_utt,
                                                         int64_t total_rows, CompareParams Q, float *out, int64_t ld, int de_col) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid >> 2;
  const int d = (int)(gid & 3);
  if (row >= total_rows) return;
  int lo = 0, hi = n_utt;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (row_off[mid] <= row) lo = mid; else hi = mid;
  }
  const int rows = (int)(row_off[lo + 1] - row_off[lo]);
  const int t = (int)(row - row_off[lo]);
  const int T20 = (int)(frame_off[lo + 1] - frame_off[lo]);
  const int T60 = rows - 1;
  const int len = (d == 3) ? T60 : T20;
  const float *la = Q.rawA + frame_off[lo] * 4;
  out[row * ld + d] = sma_a(la, t, d, len);
  // delta (deltaRegression.cpp:144-152) of the SMA level (rows frames, replicated at both ends)
  float num = 0.0f;
  for (int i = 1; i <= 2; ++i) {
    int a = t - i, b = t + i;
    a = a < 0 ? 0 : a;
    b = b > rows - 1 ? rows - 1 : b;
    num += (float)i * (sma_a(la, b, d, len) - sma_a(la, a, d, len));
  }
  out[row * ld + de_col + d] = num / 10.0f;
}

// Row T60+1 of group B's own levels (55 sma values, then 55 deltas) per utterance: [is13_functionalsB] reads
// lldB_smo;lldB_smo_de, which are longer than the T60+1 rows the LLD sinks keep, and summarises T60+2 of their rows
// (tests/test_oracle_pin_funcspec.py). Same float expressions as the window chain (contourSmoother.cpp:104-111,
// deltaRegression.cpp:144-152), indices clamped to each level's own range.
__global__ void __launch_bounds__(64) lld_compare_b_extra(const int64_t *frame_off, const int64_t *row_off, int n_utt,
                                                          const float *rawB, float *out110) {
  const int u = blockIdx.x, d = threadIdx.x;
  if (u >= n_utt || d >= 55) return;
  float *o = out110 + (int64_t)u * 110;
  const int64_t f0 = frame_off[u];
  const int T20 = (int)(frame_off[u + 1] - f0);
  const int rows = (int)(row_off[u + 1] - row_off[u]);
  if (rows <= 0 || T20 <= 0) { o[d] = 0.0f; o[55 + d] = 0.0f; return; }
  const float *x = rawB + f0 * 55 + d;
  auto raw = [&](int t) { t = t < 0 ? 0 : (t > T20 - 1 ? T20 - 1 : t); return x[(int64_t)t * 55]; };
  auto smo = [&](int t) {                       // level of T20+1 rows
    t = t < 0 ? 0 : (t > T20 ? T20 : t);
    float acc = raw(t);
    acc += raw(t - 1);
    acc += raw(t + 1);
    return acc / 3.0f;
  };
  const int c = rows;                           // = T60 + 1
  o[d] = smo(c);
  float norm = 0.0f;
  for (int i = 1; i <= 2; ++i) norm += (float)i * (float)i;
  norm *= 2.0;
  float num = 0.0f;
  for (int k = 1; k <= 2; ++k) {
    const float delta = smo(c + k) - smo(c - k);
    num += (float)k * delta;
  }
  o[55 + d] = num / norm;
}

hipError_t launch_compare_b_extra(const int64_t *d_frame_off, const int64_t *d_row_off, int n_utt, const float *rawB, float *out110,
                                  hipStream_t s) {
  if (n_utt <= 0) return hipSuccess;
  SMILEHIP_KLAUNCH(lld_compare_b_extra, dim3((unsigned)n_utt), dim3(64), 0, s, d_frame_off, d_row_off, n_utt, rawB, out110);
  return hipGetLastError();
}

hipError_t launch_compare(const LldParams &P, const CompareParams &Q, int n_runs, const int64_t *d_row_off,
                          int64_t total_rows, float *d_out, int64_t ld_out, int de_col, hipStream_t s) {
  if (n_runs <= 0) return hipSuccess;
  const int M = P.Nfft / 2;
  const int Npad = (P.N + 3) & ~3, Kpad = (P.K + 3) & ~3;
  if ((P.Nfft != 256 && P.Nfft != 512 && P.Nfft != 1024) || P.K != M + 1 || P.N > P.Nfft) return hipErrorInvalidValue;   // 20 ms at 8 .. 48 kHz
  static const bool use_block = getenv("SMILEHIP_COMPARE_BLOCK") != nullptr;      // the one-workgroup-per-run kernel (A/B checks, FFT 512)
  if (P.Nfft != 512) {
    const size_t lds = sizeof(float) * (size_t)(512 + Kpad + 128 + 16 * 32 + oo_table_floats(P.oo) + 4 * (compare_z_floats(M, Kpad) + 3 * Kpad + 96 + ((Q.N60 + 3) & ~3)));
    if (!P.oo.tw) return hipErrorInvalidValue;            // (the own-order A/B transform exists for the tuned geometry only)
    const void *fn = P.Nfft == 256 ? reinterpret_cast<const void *>(&lld_compare_frame_wave_g<2>) : reinterpret_cast<const void *>(&lld_compare_frame_wave_g<8>);
    hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
    if (P.Nfft == 256) SMILEHIP_KLAUNCH(lld_compare_frame_wave_g<2>, dim3((unsigned)((n_runs + 3) / 4)), dim3(256), lds, s, P, Q, n_runs);
    else SMILEHIP_KLAUNCH(lld_compare_frame_wave_g<8>, dim3((unsigned)((n_runs + 3) / 4)), dim3(256), lds, s, P, Q, n_runs);
  } else if (use_block) {
    const size_t lds = sizeof(float) * (size_t)(Npad + 2 * M + 3 * Kpad + 96) + sizeof(double) * (64 + 256) + 32 +
                       sizeof(float) * (size_t)(Kpad + 128 + 16 * 32);
    SMILEHIP_KLAUNCH(lld_compare_frame, dim3((unsigned)n_runs), dim3(256), lds, s, P, Q);
  } else {
    const size_t lds = sizeof(float) * (size_t)(512 + Kpad + 128 + 16 * 32 + oo_table_floats(P.oo) + 4 * (compare_z_floats(M, Kpad) + 3 * Kpad + 96 + ((Q.N60 + 3) & ~3)));
    if (P.N > Q.N60) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
      hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&lld_compare_frame_wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (ea == hipSuccess) ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&lld_compare_frame_wave3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (ea == hipSuccess) ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&lld_compare_frame_wave3t), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (ea != hipSuccess) return ea;
    }
    static const char *force = getenv("SMILEHIP_COMPARE_WAVES");              // "2" / "3": A/B checks
    const bool beside_small_jitter_pass = force ? force[0] == '2' : (de_col == 65 && P.n_utt < 2048);   // (65: the whole ComParE level, see above)
    // (the quad form first: it is short enough not to crowd out a small batch's jitter pass, the reason for the two-wave build below)
    // ONE statement of when the sixteen-lanes-per-frame form runs (the condition used to be written twice and the two
    // copies differed -- with SMILEHIP_COMPARE_WAVE=1 on a small batch the launch went to wave3t instead of the two-wave kernel)
    const bool tuned_geo = P.oo.tw && P.N == 320 && P.H == 160 && P.pad_left == 96 && Q.N60 == 960 && P.K == 257;
    const bool use_quad = tuned_geo && P.n_bands == 26 && P.n_mfcc == 14 && P.pcm && !P.pcm_f32 && !(force && force[0] == '2') &&
                          P.total_frames < (int64_t(1) << 31) && Q.max_utt_samples < (int64_t(1) << 31) &&
                          Q.band_iL[0] >= 0 && Q.band_iL[0] < Q.band_iR[0] && Q.band_iR[0] <= 256 &&
                          Q.band_iL[1] >= 0 && Q.band_iL[1] < Q.band_iR[1] && Q.band_iR[1] <= 256 &&
                          !getenv("SMILEHIP_COMPARE_GENERAL") && !getenv("SMILEHIP_COMPARE_WAVE");
    // (the quad form first: it is short enough not to crowd out a small batch's jitter pass, the reason for the two-wave build)
    if (beside_small_jitter_pass && !use_quad) SMILEHIP_KLAUNCH(lld_compare_frame_wave, dim3((unsigned)((n_runs + 3) / 4)), dim3(256), lds, s, P, Q, n_runs);
    else if (use_quad) {
      // sixteen lanes per frame (lld_compare_quad.hpp); SMILEHIP_COMPARE_WAVE=1: the wave-per-frame form (A/B switch)
      const size_t qlds = sizeof(float) * compare_quad_lds_floats(P.oo);
      hipError_t eq = hipFuncSetAttribute(reinterpret_cast<const void *>(&lld_compare_frame_quad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)qlds);
      if (eq != hipSuccess) return eq;
      const int per_wg = kCmpQuadWaves * 4;
      SMILEHIP_KLAUNCH(lld_compare_frame_quad, dim3((unsigned)((n_runs + per_wg - 1) / per_wg)), dim3(kCmpQuadWaves * 64), qlds, s, P, Q, n_runs);
    } else if (tuned_geo && !getenv("SMILEHIP_COMPARE_GENERAL"))
      SMILEHIP_KLAUNCH(lld_compare_frame_wave3t, dim3((unsigned)((n_runs + 3) / 4)), dim3(256), lds, s, P, Q, n_runs);
    else SMILEHIP_KLAUNCH(lld_compare_frame_wave3, dim3((unsigned)((n_runs + 3) / 4)), dim3(256), lds, s, P, Q, n_runs);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  SMILEHIP_KLAUNCH(lld_compare_rasta, dim3((unsigned)P.n_utt), dim3(64), 0, s, P.frame_off, P.n_utt, Q);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (total_rows > 0)
    SMILEHIP_KLAUNCH(lld_compare_groupA, dim3((unsigned)((total_rows * 4 + 255) / 256)), dim3(256), 0, s, P.frame_off,
                       d_row_off, P.n_utt, total_rows, Q, d_out, ld_out, de_col);
  return hipGetLastError();
}

// Frames per run for a batch of `total_frames` 20 ms frames: a run costs one transform more than its frames (the warm-up frame),
// 12.5 % at 8; the longest of 8 / 16 / 32 / 64 that still leaves >= 65 536 runs (21 per wave slot of the device).
int compare_run_frames(int64_t total_frames) {
  int L = kRun;
  while (L < 64 && total_frames / (2 * L) >= 65536) L *= 2;
  return L;
}

}  // namespace smilehip
