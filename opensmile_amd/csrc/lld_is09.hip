// IS09_emotion LLD set (SURVEY.md 8a rows R9 cAcf, R10 cPitchACF, R12 cEnergy / cMZcr on
// top of the MFCC chain), reference-order kernels: one 256-thread workgroup per frame
// computes the 16 pre-smoothing LLD columns
//   [RMS energy | mfcc 1..12 | zcr | voicing probability | pitch candidate]
// a per-utterance scan applies cPitchACF's causal contour smoother, and the generic
// window chain (lld_kernels.hip) adds SMA + delta. Correctness first: this path is
// not yet tuned (three LDS FFTs per frame).
#include <hip/hip_runtime.h>
#include "kernel_timing.hpp"

#include <cstdlib>
#include <cstring>

#include <cmath>
#include <type_traits>
#include <vector>

#include "lld_blocks.hpp"
#include "lld_fft.hpp"
#include "lld_ooura_quad.hpp"
#include "lld_device.hpp"
#include "lld_launch.hpp"
#include "lld_params.hpp"
#include "lld_pitch_contour.hpp"

namespace smilehip {

// Development instrumentation (tools/ubench/variant_any.sh is09 phaseis09 -DSMILEHIP_PHASE_TIMING): s_memtime at the phase boundaries of
// the frame body, summed over all waves by lane 0. Not compiled into the product. (Wave RESIDENCE, not issue slots: with six waves per
// SIMD a phase that waits for memory shows large here and costs little -- round 3 learnt that by turning the kernel into persistent
// waves with a register prefetch: 256 VGPRs, one wave per SIMD, 3 x slower.)
#ifdef SMILEHIP_PHASE_TIMING
__device__ unsigned long long g_phase_is09[16];
#define IPHASE_DECL unsigned long long iph_last = __builtin_amdgcn_s_memtime(); unsigned long long iph_acc[8] = {0};
#define IPHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); iph_acc[i] += t_ - iph_last; iph_last = t_; } while (0)
#define IPHASE_FLUSH do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_phase_is09[i_], iph_acc[i_]); } while (0)
}  // namespace smilehip
extern "C" int smilehip_debug_phase_is09(unsigned long long *out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(smilehip::g_phase_is09), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(smilehip::g_phase_is09), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
namespace smilehip {
#else
#define IPHASE_DECL
#define IPHASE(i)
#define IPHASE_FLUSH
#endif

// the tables a frame reads: in global memory for the workgroup kernel, staged in LDS once per workgroup for the wave kernel
// (143 global loads per frame and wave before: the kernel waited on them)
struct Is09Tbl {
  const float *window;
  const float2 *tw_half, *tw_full;
  const float *mel_coef;
  const int32_t *mel_rng;
  const float *dct_rows;
  OouraTab oo;
  const double2 *log_tab;                                  // kLogTab's copy in LDS (the quad form), else null
};

// LDS, workgroup form: xr[N] | yv[N] | re[M] | im[M] | mg[K+3] | sp[K+3] | acf[M] | cep[M] | lmel[32] | scr (4 doubles)
// wave form: xr[N] | z[fft_pairs(M)] pairs | mg[K+3] (yv first) | sp[K+3] (yv, later cep) | acf[M] | lmel[32] | scr
// MC: M = Nfft / 2 when known at compile time (256: the 25 ms / 16 kHz geometry of the shipped configs), 0 = run-time
template <class G, int MC = 0>
__device__ __forceinline__ void is09_frame_body(const LldParams &P, const Is09Params &Q, const Is09Tbl &T, int64_t row, float *smem) {
  const int M = MC > 0 ? MC : (P.Nfft >> 1);
  const int Npad = (P.N + 3) & ~3;
  constexpr bool kWave = std::is_same<G, WaveG>::value;
  const int Kpad = MC > 0 ? ((MC + 1 + 3) & ~3) : ((P.K + 3) & ~3);
  const int zpad = fft_pad(M);
  float *xr = smem;
  float *re = kWave ? xr + Npad : xr + 2 * Npad;         // wave form: fft_pairs(M) (re, im) pairs, lld_fft.hpp
  float *im = re + M;
  float2 *z = reinterpret_cast<float2 *>(re);
  float *mg = re + 2 * fft_pairs(M);
  float *sp = mg + Kpad;
  float *yv = kWave ? mg : xr + Npad;                    // wave form: the windowed frame lives in mg | sp until the transform has read it
  float *acf = sp + Kpad;
  float *cep = kWave ? sp : acf + M;                     // wave form: the cepstrum's lags replace its own input (read by then)
  float *lmel = kWave ? acf + M : cep + M;
  double *scr = reinterpret_cast<double *>(lmel + 32);
  int *iscr = reinterpret_cast<int *>(scr + 4);

  int lo = 0, hi = P.n_utt;
  if (P.frame_utt) lo = P.frame_utt[row];                // (one load instead of log2(n_utt) dependent ones per frame)
  else
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (P.frame_off[mid] <= row) lo = mid; else hi = mid;
    }
  const int64_t t = row - P.frame_off[lo];
  const PcmIn x = pcm_in(P) + (P.samp_off[lo] + t * (int64_t)P.H);
  float *out = Q.raw16 + row * 16;
  int logM = 0;
  while ((1 << logM) < M) ++logM;

  IPHASE_DECL
  for (int n = G::tid(); n < P.N; n += G::size()) xr[n] = x[n];                      // R0 (or already done: float input)
  G::sync();
  IPHASE(0);   // utterance lookup + frame load

  // R12 cMZcr::processVector, zcr (mzcr.cpp:117-124): on the RAW frames
  {
    int cnt = 0;
    for (int i = 1 + G::tid(); i < P.N - 1; i += G::size())
      if (((xr[i - 1] * xr[i + 1] <= 0.0f) && (xr[i] == 0.0f)) || (xr[i - 1] * xr[i] < 0.0f)) ++cnt;
    const int total = G::sum_i(cnt, iscr);
    if (G::tid() == 0) out[13] = (float)total / (float)P.N;
  }
  IPHASE(1);   // ZCR
  // R2 + R3, then R12 cEnergy rms on the WINDOWED frame (energy.cpp:152-168)
  double e2 = 0.0;
  for (int n = G::tid(); n < P.N; n += G::size()) {
    float y = xr[n];
    if (P.preemph) y = (n == 0) ? P.one_minus_k * xr[0] : (P.de ? (xr[n] + P.k * xr[n - 1]) : (xr[n] - P.k * xr[n - 1]));
    y = y * T.window[n] + P.win_offset;
    yv[n] = y;
    const float sq = y * y;
    e2 += (double)sq;
  }
  {
    const double d = G::sum(e2, scr);
    if (G::tid() == 0) out[0] = (float)sqrt(d / (float)P.N) * 1.0f + 0.0f;
  }
  IPHASE(2);   // pre-emphasis, window, RMS energy
  // R4 forward real FFT
  if (T.oo.tw) {                                         // the reference's operation order (lld_ooura.hpp / lld_ooura_wave.hpp)
    const auto load_pair = [&](int i) {
      const int n0 = 2 * i - P.pad_left, n1 = n0 + 1;
      return make_float2((n0 >= 0 && n0 < P.N) ? yv[n0] : 0.0f, (n1 >= 0 && n1 < P.N) ? yv[n1] : 0.0f);
    };
    if constexpr (kWave) {
      oo_wave_forward<MC>(z, T.oo, G::tid(), load_pair);
      for (int k = G::tid(); k <= M; k += 64) mg[k] = bin_magnitude(oo_wave_bin<MC>(z, T.oo, k), k == 0 || k == M);  // R5
    } else {
      ooura_forward<G>(z, T.oo, load_pair);
      for (int k = G::tid(); k <= M; k += G::size()) mg[k] = bin_magnitude(ooura_bin(z, T.oo, k), k == 0 || k == M);   // R5
    }
  } else if constexpr (kWave) {
    wave_cfft(z, M, T.tw_half, G::tid(), [&](int i) {
      const int n0 = 2 * i - P.pad_left, n1 = n0 + 1;
      return make_float2((n0 >= 0 && n0 < P.N) ? yv[n0] : 0.0f, (n1 >= 0 && n1 < P.N) ? yv[n1] : 0.0f);
    });
    for (int k = G::tid(); k <= M; k += 64)
      mg[k] = bin_magnitude(wave_untangle(z, M, zpad, k, T.tw_full), k == 0 || k == M);   // R5
  } else {
    for (int i = G::tid(); i < M; i += G::size()) {
      const int n0 = 2 * i - P.pad_left, n1 = n0 + 1;
      const int r = (int)(__brev((unsigned)i) >> (32 - logM));
      re[r] = (n0 >= 0 && n0 < P.N) ? yv[n0] : 0.0f;
      im[r] = (n1 >= 0 && n1 < P.N) ? yv[n1] : 0.0f;
    }
    G::sync();
    group_cfft_radix2<G>(re, im, M, T.tw_half);
    for (int k = G::tid(); k <= M; k += G::size())
      mg[k] = bin_magnitude(untangle_bin(re, im, M, k, T.tw_full), k == 0 || k == M);     // R5
  }
  G::sync();
  IPHASE(3);   // forward transform + magnitudes
  // R6 / R7: mel (usePower per config) -> log -> DCT
  for (int k = G::tid(); k <= M; k += G::size()) sp[k] = P.use_power ? mg[k] * mg[k] : mg[k];
  G::sync();
  for (int b = G::tid(); b < P.n_bands; b += G::size())
    lmel[b] = log_mel(mel_band_exact(sp, T.mel_coef, T.mel_rng, b, P.mel_scale), P.melfloor, P.log_floor);
  G::sync();
  for (int r = G::tid(); r < P.n_mfcc; r += G::size())
    out[1 + r] = dct_coeff(lmel, T.dct_rows + r * P.n_bands, P.n_bands, P.dct_gain[r]);
  G::sync();

  IPHASE(4);   // mel, log, DCT
  // R9 cAcf (acf.cpp:249-349): ACF of the power spectrum, then the cepstrum instance
  for (int k = G::tid(); k <= M; k += G::size()) sp[k] = mg[k] * mg[k];                // usePower=1 (:252-259)
  G::sync();
  if (T.oo.tw) { if constexpr (kWave) oo_wave_irfft_even<MC>(sp, z, T.oo, acf, (float)P.K, true, G::tid()); else oo_irfft_even<G>(sp, z, T.oo, acf, (float)P.K, true); }
  else if constexpr (kWave) wave_irfft_even(sp, z, M, T.tw_half, T.tw_full, acf, (float)P.K, true, G::tid());
  else group_irfft_even<G>(sp, re, im, M, logM, T.tw_half, T.tw_full, acf, (float)P.K, true);
  for (int k = G::tid(); k <= M; k += G::size()) {
    const float p = mg[k] * mg[k];
    sp[k] = (p > 0.0f) ? (float)log_d((double)p + 1.0) : 0.0f;                              // :288-305
  }
  G::sync();
  if (T.oo.tw) { if constexpr (kWave) oo_wave_irfft_even<MC>(sp, z, T.oo, cep, (float)P.K, false, G::tid()); else oo_irfft_even<G>(sp, z, T.oo, cep, (float)P.K, false); }
  else if constexpr (kWave) wave_irfft_even(sp, z, M, T.tw_half, T.tw_full, cep, (float)P.K, false, G::tid());
  else group_irfft_even<G>(sp, re, im, M, logM, T.tw_half, T.tw_full, cep, (float)P.K, false);

  IPHASE(5);   // ACF + cepstrum: two inverse transforms, 257 double logs
  // R10 cPitchACF::processVector, per-frame part (pitchACF.cpp:137-192)
  double voicing, Tsamp;
  int max_idx;
  group_pitchacf_frame<G>(acf, cep, M, Q.fsSec, Q.maxPitch, scr, iscr, voicing, max_idx, Tsamp);
  if (G::tid() == 0) {
    long maxIdx = max_idx;
    float pitch = 0.0f;
    if (maxIdx > 0) pitch = 1.0f / ((float)maxIdx * (float)Tsamp);
    if (voicing < Q.voicingCutoff) pitch = 0.0f;
    out[14] = (float)voicing;
    out[15] = pitch;                                    // smoothed in place by lld_pitch_smooth
  }
  IPHASE(6);   // cPitchACF
  IPHASE_FLUSH;
}

// one workgroup per frame (any FFT size the LDS holds)
__global__ void __launch_bounds__(256) lld_is09_frame(LldParams P, Is09Params Q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Is09Tbl T = {P.window, P.tw_half, P.tw_full, P.mel_coef, P.mel_rng, P.dct_rows, P.oo};
  is09_frame_body<BlockG>(P, Q, T, (int64_t)blockIdx.x, smem);
}

// floats of LDS the wave kernel's shared tables take: window | tw_half | tw_full | mel_coef | mel_rng | dct_rows
__host__ __device__ inline int is09_table_floats(int N, int M, int K) {
  return ((N + 3) & ~3) + M + (M + 4) + ((K + 3) & ~3) + 128 + 16 * 32;
}

// one WAVE per frame, four frames per workgroup, no barriers after the table staging
__global__ void __launch_bounds__(256) lld_is09_frame_wave(LldParams P, Is09Params Q, int wave_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int M = P.Nfft >> 1, Npad = (P.N + 3) & ~3, Kpad = (P.K + 3) & ~3;
  float *s_win = smem;
  float2 *s_twh = reinterpret_cast<float2 *>(s_win + Npad);
  float2 *s_twf = s_twh + M / 2;
  float *s_coef = reinterpret_cast<float *>(s_twf + M / 2 + 2);
  int32_t *s_rng = reinterpret_cast<int32_t *>(s_coef + Kpad);
  float *s_dct = reinterpret_cast<float *>(s_rng + 128);
  for (int i = threadIdx.x; i < P.N; i += 256) s_win[i] = P.window[i];
  for (int i = threadIdx.x; i < M / 2; i += 256) s_twh[i] = P.tw_half[i];
  for (int i = threadIdx.x; i <= M / 2; i += 256) s_twf[i] = P.tw_full[i];
  for (int i = threadIdx.x; i < P.K; i += 256) s_coef[i] = P.mel_coef[i];
  for (int i = threadIdx.x; i < 4 * P.n_bands; i += 256) s_rng[i] = P.mel_rng[i];
  for (int i = threadIdx.x; i < P.n_mfcc * P.n_bands; i += 256) s_dct[i] = P.dct_rows[i];
  const OouraTab s_oo = oo_stage_tables(P.oo, smem + is09_table_floats(P.N, M, P.K), threadIdx.x, 256);
  __syncthreads();                                       // the only workgroup barrier
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P.total_frames) return;
  const Is09Tbl T = {s_win, s_twh, s_twf, s_coef, s_rng, s_dct, s_oo};
  float *wave_mem = smem + is09_table_floats(P.N, M, P.K) + oo_table_floats(P.oo) + (threadIdx.x >> 6) * wave_floats;
  if (M == 256) is09_frame_body<WaveG, 256>(P, Q, T, row, wave_mem);
  else is09_frame_body<WaveG>(P, Q, T, row, wave_mem);
}

// ---- SIXTEEN LANES per frame, four frames per wave (16 kHz / 25 ms: M = 256 with the reference-order tables).
// The wave-per-frame form is bound by the vector ALU's issue rate at 3 455 instructions per frame: the 64 butterflies of a
// transform level are one per lane and of mixed type / kind (the wave walks every variant), the reductions are six steps
// deep, and the narrow phases -- 26 mel bands, 12 cepstra, one lane's scalar tail -- leave most of a wave idle. Here a DPP
// row of 16 lanes owns a frame: the transform is lld_ooura_quad.hpp's schedule of the same network, every reduction a
// four-step butterfly of row rotations, the narrow phases are shared by four frames, and every loop over samples / bins /
// lags has a compile-time trip count (16 or 17 per lane; 32 predicated rounds over the frame's samples), so that a phase's
// LDS reads are in flight together. Same operations on the same operands as the wave form: bit-identical outputs
// (tests/test_gpu_is09.py::test_is09_quad_form_equals_wave_form_bit_for_bit).
// LDS per frame: zx[2 x 272] only -- the raw frame, then the windowed frame, the transform's transposition / its 256 result
// pairs, the mel input (power spectrum) and the log mel bands, the transposition buffers of the two inverse transforms. The
// magnitudes, the ACF and the cepstrum stay in registers (17 + 16 + 16 per lane): the inverse transforms take their real input
// from the registers (lld_ooura_quad.hpp: oo_quad_inverse_real), the pitch phase reads its neighbours with row rotations.
// 2.2 KB per frame; workgroups of four waves (16 frames + 8 KB of tables = 43 KB), three per CU = three waves per SIMD, persistent.
#ifndef SMILEHIP_IS09_QUAD_WAVES
#define SMILEHIP_IS09_QUAD_WAVES 3                         // waves per SIMD the register budget is set for (experiments: 2)
#endif
#ifndef SMILEHIP_IS09_PREFETCH_EARLY
#define SMILEHIP_IS09_PREFETCH_EARLY 0
#endif
#ifndef SMILEHIP_IS09_LOG_GROUP
#define SMILEHIP_IS09_LOG_GROUP 3
#endif
namespace {
constexpr int kQuadWaves = 4;
constexpr int kQuadKpad = 260;
constexpr int kQuadLmel = 2 * 257;                          // the log mel bands' place in the frame's buffer: behind the two rows of 257 mel terms
constexpr int kQuadFrameFloats = (2 * kQuadZPairs > kQuadLmel + 34 ? 2 * kQuadZPairs : kQuadLmel + 34);      // (548: the transform needs 544)
static_assert(kQuadFrameFloats % 4 == 0, "16-byte multiples per frame buffer");
// the tables the quad form reads: window | mel_coef | mel_rng | dct_rows (the reference-order transform's follow)
// log table (128 double2, first: 16-byte aligned) | window | mel_coef | mel_rng | dct_rows
__host__ __device__ inline int is09_quad_table_floats(int N) { return 512 + ((N + 3) & ~3) + kQuadKpad + 128 + 16 * 32; }

__device__ __forceinline__ float quad_ror1(float x) { return __int_as_float(row_ror_i<1>(__float_as_int(x))); }     // lane j <- j - 1
__device__ __forceinline__ float quad_rol1(float x) { return __int_as_float(row_ror_i<15>(__float_as_int(x))); }    // lane j <- j + 1
__device__ __forceinline__ float quad_lane(float x, int lane64, int src) {                                           // lane `src` of the row
  return __int_as_float(__builtin_amdgcn_ds_bpermute(4 * ((lane64 & 48) | src), __float_as_int(x)));
}

// group_pitchacf_frame (lld_blocks.hpp) for a row of 16 lanes and n = 256 lags held in registers, in two parts so that the ACF
// is dead before the cepstrum's transform starts: a[it] = acf[j + 16 it], c[it] = cep[j + 16 it]. The same selections, trip
// counts fixed.
__device__ __forceinline__ int quad_pitch_preskip(double fsSec, double maxPitch) {
  const double Tsamp = fsSec / (double)(2 * 256);
  return (maxPitch <= 0.0) ? 0 : (int)(1.0 / (maxPitch * Tsamp));
}
__device__ __forceinline__ double quad_pitch_voicing(const float (&a)[16], int preskip, int lane64) {
  const int j = lane64 & 15;
  double vmax = quad_lane(a[15], lane64, 15);              // acf[n - 1]
  float prev = 0.0f;                                       // lane 15's acf[16 (it - 1) + 15] as lane 0 sees it
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int i = j + 16 * it;
    const float ra = quad_ror1(a[it]);
    const float am = (j == 0) ? prev : ra;                 // acf[i - 1]
    prev = ra;
    if (i >= 1 && i >= preskip && ((double)a[it] > vmax) && (am < a[it])) vmax = a[it];
  }
  vmax = QuadG::max(vmax, nullptr);
  const float a0 = quad_lane(a[0], lane64, 0);
  return (a0 > 0.0f) ? vmax / (double)a0 : 0.0;
}
__device__ __forceinline__ int quad_pitch_cep_peak(const float (&c)[16], int preskip, int lane64) {
  constexpr int n = 256;
  const int j = lane64 & 15;
  const int skip = preskip + 1;
  double csum = 0.0, cmax = quad_lane(c[15], lane64, 15);  // cep[n - 1]
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int i = j + 16 * it;
    const double b = c[it];
    csum += fabs(b);
    if (i >= skip && b > cmax) cmax = b;
  }
  csum = QuadG::sum(csum, nullptr) / n;
  cmax = QuadG::max(cmax, nullptr);
  const double thr = (cmax + csum) * 0.6;
  int first = 1 << 30;
  float next = 0.0f;                                       // lane 0's cep[16 (it + 1)] as lane 15 sees it
#pragma unroll
  for (int it = 15; it >= 0; --it) {                       // (descending: the lane's smallest qualifying index stays)
    const int i = j + 16 * it;
    const float lc = quad_rol1(c[it]);
    const float cp = (j == 15) ? next : lc;                // cep[i + 1]
    next = lc;
    const float rc = quad_ror1(c[it]);
    const float rcp = (it > 0) ? quad_ror1(c[it > 0 ? it - 1 : 0]) : 0.0f;
    const float cm = (j == 0) ? rcp : rc;                  // cep[i - 1]
    if (i >= skip + 1 && i < n - 1)
      if ((double)c[it] > thr && (cm < c[it]) && (c[it] > cp)) first = i;
  }
  first = QuadG::min_i(first, nullptr);
  return (first == (1 << 30)) ? 0 : first;
}

// The samples of frame `row` that lane j of the frame's row of lanes keeps: x[j + 16 it], it < NIT (R0 done at the load)
// FULL: N = 16 NIT exactly and no zero padding in front of the frame (25 ms at 16 kHz, zeroPadSymmetric = 0: the shipped
// IS09 / emobase files) -- every "n < N" below is then a compile-time fact and the predicated regions around the LDS accesses go
// (in two steps: the frame's first sample -- two dependent look-ups -- is asked for a whole pass before the samples are)
__device__ __forceinline__ int64_t is09_quad_base(const LldParams &P, int64_t row) {
  const int lo = P.frame_utt[row];
  const int64_t t = row - P.frame_off[lo];
  return P.samp_off[lo] + t * (int64_t)P.H;
}
template <int NIT, bool FULL>
__device__ __forceinline__ void is09_quad_fetch(const LldParams &P, int64_t base0, float (&R)[NIT], int j) {
  const int64_t base = base0 + j;
  const int N = FULL ? 16 * NIT : P.N;
  if (P.pcm_f32) {                                         // (uniform: one kind of load per instance of the loop)
    const float *x = P.pcm_f32 + base;
#pragma unroll
    for (int it = 0; it < NIT; ++it) R[it] = (FULL || j + 16 * it < N) ? x[16 * it] : 0.0f;
  } else {
    const int16_t *x = P.pcm + base;
#pragma unroll
    for (int it = 0; it < NIT; ++it) R[it] = (FULL || j + 16 * it < N) ? pcm16_to_float(x[16 * it]) : 0.0f;
  }
}

// One pass: the four frames whose samples R holds. Before the last phase (which needs few registers) the samples of the
// wave's next pass are requested into R again (next_row < 0: none), so that their latency is the pitch phase's.
template <int NIT, bool FULL>
__device__ __forceinline__ void is09_quad_body(const LldParams &P, const Is09Params &Q, const Is09Tbl &T, int64_t row, float *fmem,
                                               float (&R)[NIT], int64_t next_row) {
  constexpr int M = 256;
  // The thread index is made opaque per pass: what depends on the lane alone (addresses, the masks of the unrolled loops) is
  // loop-invariant over the wave's passes, and the compiler would carry all of it across the loop (44 registers, see f0_shs)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane64 = tid & 63, j = tid & 15;
  fmem += ((lane64 >> 4)) * kQuadFrameFloats;              // (fmem: the wave's four frame buffers)
  float *xr = fmem;                                        // raw frame, then the windowed frame
  float2 *z = reinterpret_cast<float2 *>(fmem);
  float *sp = fmem;                                        // mel input, once the spectrum has been read out of z
  float *lmel = fmem + kQuadLmel;
  const int N = FULL ? 16 * NIT : P.N;
  const int pad_left = FULL ? 0 : P.pad_left;
  float *out = Q.raw16 + row * 16;
  // where the NEXT pass's frame starts: requested now, used by the fetch at the end of this pass (two dependent global loads:
  // asked for at the fetch itself they are two memory round trips during which the wave -- one of three on its SIMD -- waits)
  // (the FULL instance has the two registers to spare; the general ones ask at the fetch)
  const int64_t next_base = (FULL && next_row >= 0) ? is09_quad_base(P, next_row) : 0;
  IPHASE_DECL
#pragma unroll
  for (int it = 0; it < NIT; ++it) { const int n = j + 16 * it; if (FULL || n < N) xr[n] = R[it]; }
  QuadG::sync();
  IPHASE(0);   // utterance lookup + frame load
  // R12 cMZcr::processVector, zcr (mzcr.cpp:117-124): on the RAW frames
  {
    int cnt = 0;
#pragma unroll 8
    for (int it = 0; it < NIT; ++it) {
      const int i = 1 + j + 16 * it;
      if (i < N - 1)                                       // (FULL: false in lanes 14, 15 of the last round only)
        if (((xr[i - 1] * xr[i + 1] <= 0.0f) && (xr[i] == 0.0f)) || (xr[i - 1] * xr[i] < 0.0f)) ++cnt;
    }
    const int total = QuadG::sum_i(cnt, nullptr);
    if (j == 0) out[13] = (float)total / (float)N;
  }
  IPHASE(1);   // ZCR
  // R2 + R3, then R12 cEnergy rms on the WINDOWED frame (energy.cpp:152-168): the lane's samples in R again, then in place
  double e2 = 0.0;
#pragma unroll 8
  for (int it = 0; it < NIT; ++it) {
    const int n = j + 16 * it;
    if (FULL || n < N) {
      float y = xr[n];
      if (P.preemph) y = (n == 0) ? P.one_minus_k * xr[0] : (P.de ? (xr[n] + P.k * xr[n - 1]) : (xr[n] - P.k * xr[n - 1]));
      y = y * T.window[n] + P.win_offset;
      R[it] = y;
      const float sq = y * y;
      e2 += (double)sq;
    }
  }
  {
    const double d = QuadG::sum(e2, nullptr);
    if (j == 0) out[0] = (float)sqrt(d / (float)N) * 1.0f + 0.0f;
  }
  QuadG::sync();                                           // (every lane has read its raw samples)
#pragma unroll
  for (int it = 0; it < NIT; ++it) { const int n = j + 16 * it; if (FULL || n < N) xr[n] = R[it]; }
  QuadG::sync();
  IPHASE(2);   // pre-emphasis, window, RMS energy
  // R4 forward real FFT in the reference's operation order (all of the frame is in registers before the transposition writes
  // z = the same buffer), R5 magnitudes: mv[m] = |X[j + 16 m]|, kept in registers for the two inverse transforms
  oo_quad_forward(z, T.oo, lane64, [&](int i) {
    const int n0 = 2 * i - pad_left, n1 = n0 + 1;
    return make_float2((n0 >= 0 && n0 < N) ? xr[n0] : 0.0f, (n1 >= 0 && n1 < N) ? xr[n1] : 0.0f);
  });
  // (bins j + 16 m, m < 16, are inner bins -- sqrt(re^2 + im^2) -- except bin 0 (lane row j = 0, m = 0); m = 16 exists for j = 0 only
  //  and is the edge bin M: |re|. The sixteen square roots go through sqrt_rn_batch, lld_device.hpp.)
  float mv[17];
  float edge0 = 0.0f;
#pragma unroll
  for (int m = 0; m < 17; ++m) {
    const int k = j + 16 * m;
    const float2 X = oo_wave_bin<256>(z, T.oo, k <= M ? k : 0);
    mv[m] = (m < 16) ? X.x * X.x + X.y * X.y : ((k <= M) ? fabsf(X.x) : 0.0f);
    if (m == 0) edge0 = fabsf(X.x);
    if (m % 6 == 5) __builtin_amdgcn_sched_barrier(0);     // (six bins' loads in flight at a time: all 17 at once spill)
  }
  if (j == 0) mv[0] = 1.0f;                                // (bin 0 takes |re|: no root of it is used)
  sqrt_rn_batch(reinterpret_cast<float (&)[16]>(mv));
  if (j == 0) mv[0] = edge0;
  QuadG::sync();                                           // (z has been read: the mel terms take its place)
  // R6's terms per bin, straight from the registers (lld_device.hpp: mel_band_from_terms): a[k] = p w, r[k] = p - p w
  {
    float *mt_a = sp, *mt_r = sp + (M + 1);
#pragma unroll
    for (int m = 0; m < 17; ++m) {
      const int k = j + 16 * m;
      if (k <= M) {
        const float pk = P.use_power ? mv[m] * mv[m] : mv[m], ak = pk * T.mel_coef[k];
        mt_a[k] = ak;
        mt_r[k] = pk - ak;
      }
    }
  }
  QuadG::sync();
  IPHASE(3);   // forward transform + magnitudes
  // R6 / R7: mel (usePower per config) -> log -> DCT
  // a lane takes band j and band n_bands - 1 - j: the bands widen with their index, and a lane's two sums are walked one after
  // the other -- narrow + wide is about the same length in every lane (band j and j + 16: lane 9 walked 8 + 60 terms, lane 0 4)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int half = P.n_bands >> 1;                       // (n_bands <= 32: launch_is09)
    const int b = h == 0 ? j : P.n_bands - 1 - j;
    const bool mine = j < half || (h == 0 && j == half && (P.n_bands & 1));   // (an odd count's middle band: once)
    if (mine) lmel[b] = log_mel(mel_band_from_terms(sp, sp + (M + 1), T.mel_rng, b, P.mel_scale), P.melfloor, P.log_floor);
  }
  QuadG::sync();
  if (j < P.n_mfcc) out[1 + j] = dct_coeff(lmel, T.dct_rows + j * P.n_bands, P.n_bands, P.dct_gain[j]);
  QuadG::sync();
  IPHASE(4);   // mel, log, DCT
  // R9 cAcf (acf.cpp:249-349): ACF of the power spectrum (usePower = 1, :252-259), and what cPitchACF::processVector
  // (pitchACF.cpp:137-192) takes from it ...
  const int preskip = quad_pitch_preskip(Q.fsSec, Q.maxPitch);
  const double Tsamp = Q.fsSec / (double)(2 * M);
  double voicing;
  {
    float pw[17], acf[16];
#pragma unroll
    for (int m = 0; m < 17; ++m) pw[m] = mv[m] * mv[m];
    oo_quad_irfft_even_real(z, T.oo, acf, (float)P.K, true, lane64, pw);
    voicing = quad_pitch_voicing(acf, preskip, lane64);
  }
  __builtin_amdgcn_sched_barrier(0);
#if SMILEHIP_IS09_PREFETCH_EARLY
  if (next_row >= 0) is09_quad_fetch<NIT, FULL>(P, FULL ? next_base : is09_quad_base(P, next_row), R, j);   // (the next pass's samples: their latency is the logarithms' and the second transform's)
  __builtin_amdgcn_sched_barrier(0);
#endif
  // ... then the cepstrum instance: log(P + 1) (:288-305)
  int max_idx;
  {
    float pw[17], cep[16];
#pragma unroll
    for (int m = 0; m < 17; ++m) {
      const float p = mv[m] * mv[m];
      pw[m] = (p > 0.0f) ? (float)log_d<true>((double)p + 1.0, T.log_tab) : 0.0f;   // (1 + p >= 1: never the library's path)
      if (m % SMILEHIP_IS09_LOG_GROUP == SMILEHIP_IS09_LOG_GROUP - 1) __builtin_amdgcn_sched_barrier(0);   // (a few logarithms' intermediates at a time: all 17 at once spill)
    }
    oo_quad_irfft_even_real(z, T.oo, cep, (float)P.K, false, lane64, pw);
    IPHASE(5);   // ACF + cepstrum: two inverse transforms, 257 double logs
#if !SMILEHIP_IS09_PREFETCH_EARLY
    if (next_row >= 0) is09_quad_fetch<NIT, FULL>(P, FULL ? next_base : is09_quad_base(P, next_row), R, j);
#endif
    max_idx = quad_pitch_cep_peak(cep, preskip, lane64);
  }
  if (j == 0) {
    long maxIdx = max_idx;
    float pitch = 0.0f;
    if (maxIdx > 0) pitch = 1.0f / ((float)maxIdx * (float)Tsamp);
    if (voicing < Q.voicingCutoff) pitch = 0.0f;
    out[14] = (float)voicing;
    out[15] = pitch;                                    // smoothed in place by lld_pitch_smooth
  }
  QuadG::sync();
  IPHASE(6);   // cPitchACF
  IPHASE_FLUSH;
}
}  // namespace

// persistent workgroups; rows past the end of the batch repeat the last frame (same values to the same cells)
template <int NIT, bool FULL>                            // samples per lane: N <= 16 NIT (FULL: N = 16 NIT, pad_left = 0)
__global__ void __launch_bounds__(kQuadWaves * 64) __attribute__((amdgpu_waves_per_eu(SMILEHIP_IS09_QUAD_WAVES, SMILEHIP_IS09_QUAD_WAVES))) lld_is09_frame_quad(LldParams P, Is09Params Q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Npad = (P.N + 3) & ~3;
  double2 *s_log = reinterpret_cast<double2 *>(smem);
  float *s_win = smem + 512;
  float *s_coef = s_win + Npad;
  int32_t *s_rng = reinterpret_cast<int32_t *>(s_coef + kQuadKpad);
  float *s_dct = reinterpret_cast<float *>(s_rng + 128);
  for (int i = threadIdx.x; i < 128; i += kQuadWaves * 64) s_log[i] = kLogTab[i];
  for (int i = threadIdx.x; i < P.N; i += kQuadWaves * 64) s_win[i] = P.window[i];
  for (int i = threadIdx.x; i < P.K; i += kQuadWaves * 64) s_coef[i] = P.mel_coef[i];
  for (int i = threadIdx.x; i < 4 * P.n_bands; i += kQuadWaves * 64) s_rng[i] = P.mel_rng[i];
  for (int i = threadIdx.x; i < P.n_mfcc * P.n_bands; i += kQuadWaves * 64) s_dct[i] = P.dct_rows[i];
  const OouraTab s_oo = oo_stage_tables<true>(P.oo, smem + is09_quad_table_floats(P.N), threadIdx.x, kQuadWaves * 64);   // (launch_is09: P.oo.tw != null)
  __syncthreads();                                       // the only workgroup barrier
  const Is09Tbl T = {s_win, nullptr, nullptr, s_coef, s_rng, s_dct, s_oo, s_log};
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = (threadIdx.x & 63) >> 4;
  float *fmem = smem + is09_quad_table_floats(P.N) + oo_table_floats(P.oo) + (wave * 4) * kQuadFrameFloats;
  const int64_t stride = (int64_t)gridDim.x * (4 * kQuadWaves), last = P.total_frames - 1;
  int64_t row0 = ((int64_t)blockIdx.x * kQuadWaves + wave) * 4;
  if (row0 > last) return;
  float R[NIT];
  is09_quad_fetch<NIT, FULL>(P, is09_quad_base(P, row0 + g < last ? row0 + g : last), R, (int)(threadIdx.x & 15));
  for (; row0 <= last; row0 += stride) {
    const int64_t nxt = row0 + stride;
    is09_quad_body<NIT, FULL>(P, Q, T, row0 + g < last ? row0 + g : last, fmem, R, nxt <= last ? (nxt + g < last ? nxt + g : last) : (int64_t)-1);
  }
}

// R10, sequential part: cPitchACF's causal contour (lld_pitch_contour.hpp), state per utterance. One thread per utterance.
__global__ void __launch_bounds__(64) lld_pitch_smooth(const int64_t *frame_off, int n_utt, float *raw16) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_utt) return;
  const int64_t f0 = frame_off[u];
  const int64_t T = frame_off[u + 1] - f0;
  PitchContour S = {};
  for (int64_t t = 0; t < T; ++t) {
    float *cell = raw16 + (f0 + t) * 16 + 15;
    *cell = pitch_contour_step(S, *cell);
  }
}

// The same contour for ONE stream, a frame per launch (the plugin's cPitchACF override): F0 = 1 / (idx * Tsamp) in float
// (pitchACF.cpp:185-192), the voicing cut-off, one step of the contour. out4: F0 (contour), F0raw, F0env, 0.
__global__ void __launch_bounds__(64) lld_pitch_contour_step(const double *voicing, const int32_t *max_idx, double Tsamp, double cutoff,
                                                             PitchContour *state, float *out4) {
  if (threadIdx.x != 0) return;
  PitchContour S = *state;
  const int idx = *max_idx;
  float raw = 0.0f;
  if (idx > 0) raw = 1.0f / ((float)idx * (float)Tsamp);
  const float p = (*voicing < cutoff) ? 0.0f : raw;
  out4[0] = pitch_contour_step(S, p);
  out4[1] = raw;
  out4[2] = S.env;
  out4[3] = 0.0f;
  *state = S;
}
// ... and the frames of one stream that a block tick of the plugin hands over at once: the same step, frame after frame, in one launch
__global__ void __launch_bounds__(64) lld_pitch_contour_frames(const double *voicing, const int32_t *max_idx, double Tsamp, double cutoff,
                                                               PitchContour *state, float *out4, int64_t n_frames) {
  if (threadIdx.x != 0) return;
  PitchContour S = *state;
  for (int64_t t = 0; t < n_frames; ++t) {
    const int idx = max_idx[t];
    float raw = 0.0f;
    if (idx > 0) raw = 1.0f / ((float)idx * (float)Tsamp);
    const float p = (voicing[t] < cutoff) ? 0.0f : raw;
    out4[4 * t + 0] = pitch_contour_step(S, p);
    out4[4 * t + 1] = raw;
    out4[4 * t + 2] = S.env;
    out4[4 * t + 3] = 0.0f;
  }
  *state = S;
}
hipError_t launch_pitch_contour_frames(const double *d_voicing, const int32_t *d_max_idx, double Tsamp, double cutoff, float *d_state,
                                       float *d_out4, int64_t n_frames, hipStream_t s) {
  if (n_frames > 0)
    SMILEHIP_KLAUNCH(lld_pitch_contour_frames, dim3(1), dim3(64), 0, s, d_voicing, d_max_idx, Tsamp, cutoff,
                       reinterpret_cast<PitchContour *>(d_state), d_out4, n_frames);
  return hipGetLastError();
}
hipError_t launch_pitch_contour_step(const double *d_voicing, const int32_t *d_max_idx, double Tsamp, double cutoff, float *d_state,
                                     float *d_out4, hipStream_t s) {
  SMILEHIP_KLAUNCH(lld_pitch_contour_step, dim3(1), dim3(64), 0, s, d_voicing, d_max_idx, Tsamp, cutoff,
                     reinterpret_cast<PitchContour *>(d_state), d_out4);
  return hipGetLastError();
}

hipError_t launch_is09(const LldParams &P, const Is09Params &Q, hipStream_t s) {
  if (P.total_frames <= 0) return hipSuccess;
  const int M = P.Nfft / 2;
  const int Npad = (P.N + 3) & ~3, Kpad = (P.K + 3) & ~3;
  const size_t lds = sizeof(float) * (size_t)(2 * Npad + 2 * fft_pairs(M) + 2 * Kpad + 2 * M + 32) + 4 * sizeof(double) + 8 * sizeof(int);
  // wave form: no yv, no cep of their own
  const size_t lds_wave = sizeof(float) * (size_t)(Npad + 2 * fft_pairs(M) + 2 * Kpad + M + 32) + 4 * sizeof(double) + 8 * sizeof(int);
  hipError_t e;
  const size_t tbl_floats = (size_t)is09_table_floats(P.N, M, P.K) + (size_t)oo_table_floats(P.oo);
  const size_t quad_bytes = sizeof(float) * ((size_t)is09_quad_table_floats(P.N) + (size_t)oo_table_floats(P.oo) +
                                             (size_t)kQuadWaves * 4 * (size_t)kQuadFrameFloats);
  // SMILEHIP_IS09 (A/B switch): "wave" = one wave per frame, "block" = one workgroup per frame (the forms before the quad one)
  const char *form = getenv("SMILEHIP_IS09");
  if (P.oo.tw && M == 256 && P.K == 257 && P.N <= 512 && P.frame_utt && P.n_bands <= 32 && P.n_mfcc <= 16 && quad_bytes <= 160 * 1024 && !form) {
    const bool n25 = P.N <= 400;                           // 25 ms at 16 kHz: 25 samples per lane
    const bool full = P.N == 400 && P.pad_left == 0;       // ... exactly 400 and the padding behind the frame: the shipped files
    const void *qfn = full ? reinterpret_cast<const void *>(&lld_is09_frame_quad<25, true>)
                           : (n25 ? reinterpret_cast<const void *>(&lld_is09_frame_quad<25, false>) : reinterpret_cast<const void *>(&lld_is09_frame_quad<32, false>));
    e = hipFuncSetAttribute(qfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_bytes);
    if (e != hipSuccess) return e;
    const int n_cu = current_device_cus();
    int64_t grid = (P.total_frames + 4 * kQuadWaves - 1) / (4 * kQuadWaves);
    const int64_t cap = (int64_t)(n_cu > 0 ? n_cu : 256) * (int64_t)(160 * 1024 / quad_bytes);
    if (grid > cap) grid = cap;
    if (full) SMILEHIP_KLAUNCH((lld_is09_frame_quad<25, true>), dim3((unsigned)grid), dim3(kQuadWaves * 64), quad_bytes, s, P, Q);
    else if (n25) SMILEHIP_KLAUNCH((lld_is09_frame_quad<25, false>), dim3((unsigned)grid), dim3(kQuadWaves * 64), quad_bytes, s, P, Q);
    else SMILEHIP_KLAUNCH((lld_is09_frame_quad<32, false>), dim3((unsigned)grid), dim3(kQuadWaves * 64), quad_bytes, s, P, Q);
  } else if (P.n_bands <= 32 && P.n_mfcc <= 16 && 4 * lds_wave + 4 * tbl_floats <= 64 * 1024 &&
      !(form && !strcmp(form, "block"))) {                             // a wave per frame: four frames per workgroup
    const int wave_floats = (int)((lds_wave + 15) / 16) * 4;
    const size_t total = sizeof(float) * (tbl_floats + 4 * (size_t)wave_floats);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lld_is09_frame_wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)total);
    if (e != hipSuccess) return e;
    SMILEHIP_KLAUNCH(lld_is09_frame_wave, dim3((unsigned)((P.total_frames + 3) / 4)), dim3(256), total, s, P, Q, wave_floats);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lld_is09_frame), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    SMILEHIP_KLAUNCH(lld_is09_frame, dim3((unsigned)P.total_frames), dim3(256), lds, s, P, Q);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  SMILEHIP_KLAUNCH(lld_pitch_smooth, dim3((unsigned)((P.n_utt + 63) / 64)), dim3(64), 0, s, P.frame_off, P.n_utt, Q.raw16);
  return hipGetLastError();
}

}  // namespace smilehip
