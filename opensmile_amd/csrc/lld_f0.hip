// F0 group of ComParE_2016 / GeMAPS, reference-order kernels:
//   lld_f0_frame    persistent workgroups of 4 waves, every table in LDS for the workgroup's life; a wave works on
//                   3 consecutive 60 ms frames at a time:
//                   per frame, whole wave: gauss window + RMS energy, FFT, magnitude; cSpecScale's peak enhancement
//                   (local maxima as ballots, "a maximum within two bins" as mask shifts) and (1,2,1)/4 smoothing;
//                   the parallel half of the natural cubic spline onto the octave axis (6*ut);
//                   once per 3 frames, one lane per frame: the spline's tridiagonal sweep -- a recurrence in double
//                   whose rounding order is part of the result. Its data-independent half is precomputed on the
//                   host (F0Params::sp_rec), the data-dependent half (2 x K dependent steps, 5 double operations
//                   per bin) is software-pipelined eight steps per round;
//                   per frame: spline evaluation + auditory weighting; cPitchShs: sub-harmonic summation, the six
//                   best local maxima (greedy insertion == top six by (score descending, bin ascending)),
//                   the reference's sequential double mean (one lane per frame again), parabolic refinement,
//                   voicing, the range filter and best-first reordering of cPitchBase.
//                   Result: the 21 values of level is13_pitchShsG60 and the frame energy.
//                   Measured (1000 x 10 s, 995 000 frames): 32 ms; variants 6 waves x 2 frames 35 ms, 8 x 1 48 ms,
//                   12 x 1 43 ms: the frames-per-wave of the serial phases matters most, LDS (8.4 KB per frame in
//                   flight + 47 KB of tables) limits it. Next step if this chain becomes a headline: the sweeps as
//                   a thread-per-frame kernel over global scratch (64 chains per wave).
//   lld_f0_viterbi  one wave per utterance: cPitchSmootherViterbi's incremental Viterbi pass (7 states, 30-frame
//                   path buffer, decisions emitted where all paths agree or forced when the buffer is full),
//                   lane = (state now, state before) pair; then cValbasedSelector's energy gate. 2.6 ms.
#include <hip/hip_runtime.h>

#include "lld_blocks.hpp"
#include "lld_device.hpp"
#include "lld_launch.hpp"
#include "lld_params.hpp"

namespace smilehip {

namespace {
constexpr int kWaves = 4;        // waves per persistent workgroup
constexpr int kW = 3;            // frames a wave holds at a time: one lane each in the serial phases
constexpr int kTileFrames = 8;   // consecutive frames of one utterance per work item
constexpr int kNC = 6;           // candidates (nCandidates of [is13_shs])
constexpr int kVB = 30;          // bufferLength of [is13_pitchSmoothViterbi]
constexpr int kNS = kNC + 1;     // Viterbi states: candidates + "unvoiced"
// the frame kernel is written for the 60 ms / 16 kHz geometry of ComParE / GeMAPS: FFT 1024
constexpr int kNfftF0 = 1024, kM = 512, kK = 513, kKP = 516, kPer = 9;   // complex points, bins, padded bins, bins per lane

// LDS tables shared by the workgroup for its whole life (every table the per-frame code reads)
struct F0Tbl {
  const double2 *sp;     // [kKP] (sigma_i, p_i)
  const double *dec, *d1, *d2, *a, *c, *d, *audw;   // [kKP] each
  const int *k;          // [kKP]
  const float *win;      // [NP]
  const float2 *twh;     // [kM/2]
  const float2 *twf;     // [kM/2+4]
};
__host__ __device__ inline size_t f0_shared_bytes(int N) {
  const size_t np = (size_t)((N + 3) & ~3);
  return (size_t)kKP * 16 + (size_t)kKP * 8 * 7 + (size_t)kKP * 4 + np * 4 + (size_t)(kM / 2) * 8 + (size_t)(kM / 2 + 4) * 8;
}
// one frame's LDS region: A[kKP] B[kKP] doubles | ci[8] ints (ci[7]: number of candidates) | cf[3][8] floats | double
constexpr size_t kFrameBytes = (size_t)kKP * 16 + 8 * 4 + 24 * 4 + 16;   // + the frame's sum of squares

// smileMath_quadFrom3pts (smileUtil.c:1009-1033)
__device__ __forceinline__ double quad_vertex(double x1, double y1, double x2, double y2, double x3, double y3, double &y) {
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

// cSmileViterbiPitchSmooth::getFweight (pitchSmootherViterbi.hpp:167-197)
__device__ __forceinline__ double f_weight(float f) {
  if (f > 0.0 && f < 100.0) return -(1.0 / 100.0) * f + 1.0;
  else if (f >= 100.0 && f < 350.0) return 0.0;
  else if (f >= 350.0 && f < 600.0) return ((f - 350.0) / 250.0);
  else if (f >= 600.0) return 1.2;
  else if (f <= 0) return 2.0;
  return 0.0;
}
}  // namespace

// Development instrumentation (tools/ubench/variant_f0.sh builds a private copy with -DSMILEHIP_PHASE_TIMING):
// s_memtime at the phase boundaries, summed over all waves. Not compiled into the product.
#ifdef SMILEHIP_PHASE_TIMING
__device__ unsigned long long g_phase_f0[16];
#define PHASE_DECL unsigned long long ph_acc[12] = {0}; unsigned long long ph_last = __builtin_amdgcn_s_memtime();
#define PHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[i] += t_ - ph_last; ph_last = t_; } while (0)
#define PHASE_FLUSH do { if (lane == 0) for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_phase_f0[i_], ph_acc[i_]); } while (0)
extern "C" int smilehip_debug_phase_f0(unsigned long long *out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_f0), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_f0), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#else
#define PHASE_DECL
#define PHASE(i)
#define PHASE_FLUSH
#endif

// ---- per-frame phases. Lane l owns bins i = l + 64 m, m = 0..8 (bin 512: lane 0); all loops over m are unrolled
// so that a phase's LDS reads are issued together. A, B: the frame's two double arrays.
#define F0_FOR_BINS(m, i) _Pragma("unroll") for (int m = 0, i = lane; m < kPer; ++m, i += 64)

// window + energy, FFT, magnitude, cSpecScale's enhancement and smoothing, and the parallel half of the spline:
// leaves y (smoothed spectrum) in A and 6*ut in B. Returns the frame's sum of squares.
__device__ __forceinline__ double f0_spectrum(const F0Tbl &T, const F0Params &Q, const int16_t *x, int lane, double *A,
                                              double *B) {
  float *re = reinterpret_cast<float *>(A), *im = re + kM;
  // R0 + R3 (gauss) + R12 energy of the windowed frame ([is13_energy60], energy.cpp:152-168)
  double esum = 0.0;
#pragma unroll
  for (int m = 0; m < kM / 64; ++m) {
    const int i = lane + 64 * m;
    const int n0 = 2 * i - Q.pad_left, n1 = n0 + 1;
    float a = 0.0f, b = 0.0f;
    if (n0 >= 0 && n0 < Q.N) { a = pcm16_to_float(x[n0]) * T.win[n0]; const float sq = a * a; esum += (double)sq; }
    if (n1 >= 0 && n1 < Q.N) { b = pcm16_to_float(x[n1]) * T.win[n1]; const float sq = b * b; esum += (double)sq; }
    const int r = (int)(__brev((unsigned)i) >> (32 - 9));
    re[r] = a;
    im[r] = b;
  }
  esum = WaveG::sum(esum, nullptr);
  WaveG::sync();
  group_cfft_radix2<WaveG>(re, im, kM, T.twh);
  double mg[kPer];
  F0_FOR_BINS(m, k) {
    mg[m] = 0.0;
    if (k < kK) mg[m] = (double)bin_magnitude(untangle_bin(re, im, kM, k, T.twf), k == 0 || k == kM);
  }
  F0_FOR_BINS(m, k) if (k < kK) B[k] = mg[m];
  WaveG::sync();
  // smileDsp_specEnhanceSHS (smileUtil.c:1965-2001): between consecutive local maxima everything further than two
  // bins from both is zeroed; nothing before the first / after the last maximum. With exactly one maximum the
  // reference reads its zero-initialised list: everything from bin 3 on is zeroed. The maxima of 64 consecutive
  // bins are one ballot; "a maximum within two bins" is shifts of those masks (scalar work).
  double lf[kPer], rt[kPer];
  F0_FOR_BINS(m, i) {
    lf[m] = (i >= 1 && i < kK) ? B[i - 1] : 0.0;
    rt[m] = (i < kK - 1) ? B[i + 1] : 0.0;
  }
  unsigned long long pm[kPer];
  F0_FOR_BINS(m, i) {
    bool isp = false;
    if (i < kK) {
      if (i == 0) isp = mg[m] > rt[m];
      else if (i == kK - 1) isp = mg[m] > lf[m];
      else isp = (mg[m] > lf[m]) && (mg[m] >= rt[m]);
    }
    pm[m] = __ballot(isp);
  }
  int cnt = 0, first = 1 << 30, last = -1;
#pragma unroll
  for (int m = 0; m < kPer; ++m) {
    cnt += __popcll(pm[m]);
    if (pm[m]) {
      if (first == (1 << 30)) first = 64 * m + __ffsll((long long)pm[m]) - 1;
      last = 64 * m + 63 - __clzll((long long)pm[m]);
    }
  }
  WaveG::sync();                                         // every lane has its neighbours: B may be overwritten
  F0_FOR_BINS(m, i) {
    unsigned long long sp = pm[m] | (pm[m] << 1) | (pm[m] << 2) | (pm[m] >> 1) | (pm[m] >> 2);
    if (m > 0) sp |= ((pm[m - 1] >> 63) * 3ull) | ((pm[m - 1] >> 62) & 1ull);
    if (m < kPer - 1) sp |= ((pm[m + 1] & 1ull) ? (3ull << 62) : 0ull) | ((pm[m + 1] & 2ull) ? (1ull << 63) : 0ull);
    const bool near = (sp >> lane) & 1ull;
    bool z = false;
    if (cnt >= 2) z = (i > first) && (i < last) && !near;
    else if (cnt == 1) z = i >= 3;
    if (z) mg[m] = 0.0;
    if (i < kK) B[i] = mg[m];
  }
  WaveG::sync();
  // smileDsp_specSmoothSHS (smileUtil.c:2004-2014); y = A (the FFT buffers are dead)
  F0_FOR_BINS(m, i) {
    lf[m] = (i >= 1 && i < kK) ? B[i - 1] : 0.0;
    rt[m] = (i < kK - 1) ? B[i + 1] : 0.0;
  }
  F0_FOR_BINS(m, i) {
    mg[m] = (i < kK - 1) ? (lf[m] + 2.0 * mg[m] + rt[m]) / 4.0 : mg[m];
    if (i < kK) A[i] = mg[m];
  }
  WaveG::sync();
  // smileMath_cspline (smileUtilSpline.c:157-212), data-dependent half, parallel part: 6*ut
  F0_FOR_BINS(m, i) {
    lf[m] = (i >= 1 && i < kK) ? A[i - 1] : 0.0;
    rt[m] = (i < kK - 1) ? A[i + 1] : 0.0;
  }
  F0_FOR_BINS(m, i)
    if (i < kK) B[i] = (i >= 1 && i < kK - 1) ? 6.0 * ((rt[m] - mg[m]) / T.d1[i] - (mg[m] - lf[m]) / T.d2[i]) : 0.0;
  WaveG::sync();
  return esum;
}

// the spline's two recurrences for this lane's frame, eight steps per round; the next round's operands are loaded
// before the current round's dependent chain runs (the loads are off the chain)
__device__ __forceinline__ void f0_spline_serial(const F0Tbl &T, double *B) {
  constexpr int R = 8;
  constexpr int nfw = (kK - 2) / R;                      // full rounds of the forward sweep (bins 1 .. kK-2)
  double up = 0.0;
  double cc[R], nc[R];
  double2 sp[R], ns[R];
#pragma unroll
  for (int q = 0; q < R; ++q) { cc[q] = B[1 + q]; sp[q] = T.sp[1 + q]; }
  for (int r = 0; r < nfw; ++r) {
    const int i = 1 + r * R;
    if (r + 1 < nfw) {
#pragma unroll
      for (int q = 0; q < R; ++q) { nc[q] = B[i + R + q]; ns[q] = T.sp[i + R + q]; }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) { up = sp[q].y * (cc[q] - sp[q].x * up); cc[q] = up; }
#pragma unroll
    for (int q = 0; q < R; ++q) B[i + q] = cc[q];
#pragma unroll
    for (int q = 0; q < R; ++q) { cc[q] = nc[q]; sp[q] = ns[q]; }
  }
  for (int i = 1 + nfw * R; i < kK - 1; ++i) { const double2 s1 = T.sp[i]; up = s1.y * (B[i] - s1.x * up); B[i] = up; }
  double yn = 0.0;                                       // y2[K-1] of the natural spline
  B[kK - 1] = 0.0;
  constexpr int nbw = (kK - 1) / R;                      // full rounds of the backward sweep (bins kK-2 .. 0)
  double dd[R], nd[R];
#pragma unroll
  for (int q = 0; q < R; ++q) { cc[q] = B[kK - 2 - q]; dd[q] = T.dec[kK - 2 - q]; }
  for (int r = 0; r < nbw; ++r) {
    const int j = kK - 2 - r * R;
    if (r + 1 < nbw) {
#pragma unroll
      for (int q = 0; q < R; ++q) { nc[q] = B[j - R - q]; nd[q] = T.dec[j - R - q]; }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) { yn = dd[q] * yn + cc[q]; cc[q] = yn; }
#pragma unroll
    for (int q = 0; q < R; ++q) B[j - q] = cc[q];
#pragma unroll
    for (int q = 0; q < R; ++q) { cc[q] = nc[q]; dd[q] = nd[q]; }
  }
  for (int j = kK - 2 - nbw * R; j >= 0; --j) { yn = T.dec[j] * yn + B[j]; B[j] = yn; }
}

// smileMath_csplint + auditory weighting (specScale.cpp:340-353), then cPitchShs::pitchDetect's summation and
// peak picking (pitchShs.cpp:226-283). Leaves hps | SS (floats) in A, SS as doubles in B, the candidate bins in ci.
// Returns the number of candidates.
__device__ __forceinline__ int f0_shs(const F0Tbl &T, const F0Params &Q, int lane, int64_t g, double *A, double *B, int *ci) {
  float *hps = reinterpret_cast<float *>(A), *SS = hps + kKP;
  float hv[kPer];
  F0_FOR_BINS(m, i) {
    hv[m] = 0.0f;
    if (i < kK) {
      const int k = T.k[i];
      const double a = T.a[i], b = 1.0 - a, c = T.c[i], d = T.d[i];
      const double o = a * A[k] + b * A[k + 1] + c * B[k] + d * B[k + 1];
      float v = (float)o;
      v = (v > 0.0f) ? (float)((double)v * T.audw[i]) : 0.0f;
      hv[m] = v;
      if (Q.hps_tap) Q.hps_tap[g * kK + i] = v;
    }
  }
  WaveG::sync();
  F0_FOR_BINS(m, i) if (i < kK) hps[i] = hv[m];
  WaveG::sync();
  // SS[j] = (in[j] + sum_h in[j + shift_h] * scale_h) / nHarmonics, terms in harmonic order
#pragma unroll
  for (int h = 0; h < 16; ++h)
    if (h < Q.n_harm - 1) {
      float v[kPer];
      F0_FOR_BINS(m, j) { const int jj = j + Q.shift[h]; v[m] = hps[jj < kK ? jj : kK - 1]; }
      F0_FOR_BINS(m, j) { const float s2 = hv[m] + v[m] * Q.scale[h]; hv[m] = (j + Q.shift[h] < kK) ? s2 : hv[m]; }
    }
  F0_FOR_BINS(m, j) {
    float s = hv[m] / (float)Q.n_harm;
    if (s < 0) s = 0.0f;
    hv[m] = s;
  }
  F0_FOR_BINS(m, j) if (j < kK) { SS[j] = hv[m]; B[j] = (double)hv[m]; }
  WaveG::sync();
  // local maxima, then the six best: greedy insertion (:262-283) keeps (score descending, bin ascending)
  float lf[kPer], rt[kPer];
  F0_FOR_BINS(m, j) {
    lf[m] = (j >= 1 && j < kK) ? SS[j - 1] : 0.0f;
    rt[m] = (j < kK - 1) ? SS[j + 1] : 0.0f;
  }
  unsigned open = 0;
  F0_FOR_BINS(m, j) if (j >= 1 && j < kK - 1 && lf[m] < hv[m] && hv[m] > rt[m]) open |= 1u << m;
  int n_found = 0;
#pragma unroll
  for (int r = 0; r < kNC; ++r) {
    float bv = -1.0f;
    int bi = 1 << 30;
    F0_FOR_BINS(m, j) if (((open >> m) & 1u) && hv[m] > bv) { bv = hv[m]; bi = j; }      // ascending j: first maximum
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (bi != (1 << 30)) {
      if (lane == 0) ci[r] = bi;
      if ((bi & 63) == lane) open &= ~(1u << (bi >> 6));
      n_found = r + 1;
    }
  }
  return n_found;
}

// the reference's sequential double sum of the summation spectrum (pitchShs.cpp:259-301) for this lane's frame
__device__ __forceinline__ double f0_mean_serial(const double *B) {
  constexpr int R = 16;
  constexpr int nr = (kK - 2) / R;
  double mean = B[0];
  double cc[R], nc[R];
#pragma unroll
  for (int q = 0; q < R; ++q) cc[q] = B[1 + q];
  for (int r = 0; r < nr; ++r) {
    if (r + 1 < nr) {
#pragma unroll
      for (int q = 0; q < R; ++q) nc[q] = B[1 + (r + 1) * R + q];
    }
#pragma unroll
    for (int q = 0; q < R; ++q) mean += cc[q];
#pragma unroll
    for (int q = 0; q < R; ++q) cc[q] = nc[q];
  }
  for (int i = 1 + nr * R; i < kK - 1; ++i) mean += B[i];
  return (mean + B[kK - 1]) / (double)kK;
}

// candidate refinement (pitchShs.cpp:304-325), cPitchBase::processVector's range filter, best-first reordering and
// output vector (pitchBase.cpp:212-300), the frame energy
__device__ __forceinline__ void f0_candidates(const F0Params &Q, int lane, int64_t g, const double *A, const int *ci, float *cf,
                                              int n_found, double mean, double esum) {
  const float *SS = reinterpret_cast<const float *>(A) + kKP;
  if (lane < 8) {
    float f0c = 0.0f, cv = 0.0f, cs = 0.0f;
    if (lane < n_found) {
      const int j = ci[lane];
      const float fj = (float)j;
      const float f1 = fj * Q.Fstept + Q.Fmint;
      const float f2 = (fj + (float)1.0) * Q.Fstept + Q.Fmint;
      const float f0 = (fj - (float)1.0) * Q.Fstept + Q.Fmint;
      double sc = 0.0;
      const double fx = quad_vertex((double)f0, (double)SS[j - 1], (double)f1, (double)SS[j], (double)f2, (double)SS[j + 1], sc);
      f0c = (float)exp(fx * Q.log_base);
      cs = (float)sc;
      cv = (sc > 0.0 && sc > mean) ? (float)(1.0 - mean / sc) : 0.0f;
    }
    cf[lane] = f0c; cf[8 + lane] = cv; cf[16 + lane] = cs;
  }
  WaveG::sync();
  if (lane == 0) {
    float *f0c = cf, *cv = cf + 8, *cs = cf + 16;
    int n = n_found;
    if (n > 0) {
      for (int c = 0; c < kNC && n > 0; c++) {
        if ((double)f0c[c] > Q.max_pitch || (double)f0c[c] < Q.min_pitch) {
          const float orig = f0c[c];
          int j;
          for (j = c + 1; j < kNC; j++) { f0c[j - 1] = f0c[j]; cv[j - 1] = cv[j]; cs[j - 1] = cs[j]; }
          f0c[j - 1] = 0; cv[j - 1] = 0; cs[j - 1] = 0;
          if (orig > 0.0f) { n--; c--; }
        }
      }
    }
    int best = 0;
    float mx = cs[0];
    for (int c = 1; c < kNC; c++) if (cs[c] > mx) { mx = cs[c]; best = c; }
    if (best > 0) {
      float tmp;
      tmp = f0c[0]; f0c[0] = f0c[best]; f0c[best] = tmp;
      tmp = cv[0]; cv[0] = cv[best]; cv[best] = tmp;
      tmp = cs[0]; cs[0] = cs[best]; cs[best] = tmp;
    }
    float *o = Q.shs + g * 21;
    o[0] = (float)n;
    for (int c = 0; c < kNC; c++) { o[1 + c] = f0c[c]; o[1 + kNC + c] = cv[c]; o[1 + 2 * kNC + c] = cs[c]; }
    o[19] = (cv[0] <= Q.voicing_cutoff) ? 0.0f : f0c[0];
    o[20] = (cv[0] <= Q.voicing_cutoff) ? 0.0f : cv[0];
    Q.e60[g] = (float)sqrt(esum / (float)Q.N) * 1.0f + 0.0f;
  }
  WaveG::sync();
}

__global__ void __launch_bounds__(kWaves * 64) lld_f0_frame(LldParams P, F0Params Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f0[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NP = (Q.N + 3) & ~3;
  double2 *c_sp = reinterpret_cast<double2 *>(smem_f0);
  double *c_dec = reinterpret_cast<double *>(c_sp + kKP);
  double *c_d1 = c_dec + kKP, *c_d2 = c_d1 + kKP, *c_a = c_d2 + kKP, *c_c = c_a + kKP, *c_d = c_c + kKP, *c_audw = c_d + kKP;
  int *c_k = reinterpret_cast<int *>(c_audw + kKP);
  float *c_win = reinterpret_cast<float *>(c_k + kKP);
  float2 *c_twh = reinterpret_cast<float2 *>(c_win + NP);
  float2 *c_twf = c_twh + kM / 2;
  for (int i = threadIdx.x; i < kK; i += blockDim.x) {
    c_sp[i] = make_double2(Q.sp_rec[4 * i], Q.sp_rec[4 * i + 1]);
    c_dec[i] = Q.sp_rec[4 * i + 2];
    c_d1[i] = Q.sp_d1[i]; c_d2[i] = Q.sp_d2[i];
    c_a[i] = Q.ip_co[3 * i]; c_c[i] = Q.ip_co[3 * i + 1]; c_d[i] = Q.ip_co[3 * i + 2];
    c_audw[i] = Q.audw[i];
    c_k[i] = Q.ip_k[i];
  }
  for (int i = threadIdx.x; i < Q.N; i += blockDim.x) c_win[i] = Q.window[i];
  for (int i = threadIdx.x; i < kM / 2; i += blockDim.x) c_twh[i] = Q.tw_half[i];
  for (int i = threadIdx.x; i <= kM / 2; i += blockDim.x) c_twf[i] = Q.tw_full[i];
  __syncthreads();                                       // the only workgroup barrier
  F0Tbl T;
  T.sp = c_sp; T.dec = c_dec; T.d1 = c_d1; T.d2 = c_d2; T.a = c_a; T.c = c_c; T.d = c_d; T.audw = c_audw;
  T.k = c_k; T.win = c_win; T.twh = c_twh; T.twf = c_twf;
  unsigned char *base = smem_f0 + f0_shared_bytes(Q.N) + (size_t)wave * (kW * kFrameBytes);
  PHASE_DECL
  // persistent waves: work item = up to kTileFrames consecutive frames of one utterance (TileRec), kW at a time
  const int tile_stride = __builtin_amdgcn_readfirstlane((int)gridDim.x) * kWaves;
  for (int tile = blockIdx.x * kWaves + wave; tile < P.n_tiles; tile += tile_stride) {
    const int64_t samp0 = P.tile_rec[tile].samp0, row0 = P.tile_rec[tile].row0;
    const int n_fr = P.tile_rec[tile].n_frames;
    for (int tf = 0; tf < n_fr; tf += kW) {
      const int n_act = (n_fr - tf < kW) ? n_fr - tf : kW;
#pragma unroll 1
      for (int w = 0; w < n_act; ++w) {
        double *A = reinterpret_cast<double *>(base + w * kFrameBytes);
        const double es = f0_spectrum(T, Q, P.pcm + samp0 + (int64_t)(tf + w) * Q.H, lane, A, A + kKP);
        if (lane == 0) *reinterpret_cast<double *>(reinterpret_cast<int *>(A + 2 * kKP) + 8 + 24) = es;
      }
      PHASE(0);   // load .. 6*ut
      if (lane < n_act) f0_spline_serial(T, reinterpret_cast<double *>(base + lane * kFrameBytes) + kKP);
      WaveG::sync();
      PHASE(1);   // recurrences
#pragma unroll 1
      for (int w = 0; w < n_act; ++w) {
        double *A = reinterpret_cast<double *>(base + w * kFrameBytes);
        int *ci = reinterpret_cast<int *>(A + 2 * kKP);
        const int nf = f0_shs(T, Q, lane, row0 + tf + w, A, A + kKP, ci);
        if (lane == 0) ci[7] = nf;
      }
      WaveG::sync();
      PHASE(2);   // interpolation, summation, top six
      double mean = 0.0;
      if (lane < n_act) mean = f0_mean_serial(reinterpret_cast<double *>(base + lane * kFrameBytes) + kKP);
      PHASE(3);   // mean
#pragma unroll 1
      for (int w = 0; w < n_act; ++w) {
        double *A = reinterpret_cast<double *>(base + w * kFrameBytes);
        int *ci = reinterpret_cast<int *>(A + 2 * kKP);
        const double es = *reinterpret_cast<double *>(ci + 8 + 24);
        f0_candidates(Q, lane, row0 + tf + w, A, ci, reinterpret_cast<float *>(ci + 8), ci[7], __shfl(mean, w), es);
      }
      PHASE(4);   // candidates + output
    }
  }
  PHASE_FLUSH;
}
#undef F0_FOR_BINS

// cSmileViterbi::addFrame / flushTrellis / getNextOutputFrame with cSmileViterbiPitchSmooth's costs
// (pitchSmootherViterbi.cpp:80-216, .hpp:105-125,158-300), driven as cPitchSmootherViterbi::myTick drives them
// (:451-570): every decided frame is written at once. Quirks kept: setWeights stores tvv into wTvvd (.hpp:294);
// `i == j == nStates-1` never holds, so unvoiced->unvoiced costs the final "return 1.0" (.hpp:229-252);
// lastChange is one variable shared by all transitions in evaluation order (i outer, j inner) and across frames.
__global__ void __launch_bounds__(64) lld_f0_viterbi(const int64_t *frame_off, int n_utt, F0Params Q, float *out, int64_t ld) {
  const int u = blockIdx.x;
  if (u >= n_utt) return;
  const int64_t fo = frame_off[u];
  const int T = (int)(frame_off[u + 1] - fo);
  if (T <= 0) return;
  __shared__ int paths[2][kNS * kVB];
  __shared__ double cost[kNS];
  __shared__ int msel[kNS];
  const int lane = threadIdx.x;
  const bool valid = lane < kNS * kNS;
  const int si = valid ? lane / kNS : 0, sj = valid ? lane % kNS : 0;
  const double wLocal = Q.vit_w[0], wTvv = Q.vit_w[1], wTvvd = Q.vit_w[2], wTvuv = Q.vit_w[3], wThr = Q.vit_w[4],
               wRange = Q.vit_w[5];
  const float thr = Q.voicing_cutoff;
  const float *S = Q.shs + fo * 21;
  double lastChange = 1.0;
  int pathBuf = 0;
  int pathIdx = 0, convIdx = -1;

  auto emit = [&](int n, int s) {
    const int64_t row = fo + n;
    const float *fr = S + (int64_t)n * 21;
    float f = (s < kNC) ? fr[1 + s] : 0.0f;
    float vp = (s < kNC) ? fr[1 + kNC + s] : fr[1 + kNC];
    if (!(Q.e60[row] > Q.min_energy)) { f = 0.0f; vp = 0.0f; }     // cValbasedSelector, zeroVec
    out[row * ld] = f;
    out[row * ld + 1] = vp;
  };

  for (int t = 0; t < T; ++t) {
    const float *cur = S + (int64_t)t * 21;
    double lc = 0.0;                                       // localCost of state `lane`
    if (lane < kNC) {
      double pv = (double)cur[1 + kNC + lane], tc = 0.0;
      if (pv < 0.01) pv = 0.01;
      if (pv > 1.00) pv = 1.00;
      if (pv < thr) tc = wThr;
      lc = (-log(pv) + tc) * wLocal + f_weight(cur[1 + lane]) * wRange;
    } else if (lane == kNC) {
      double flag = 0.0;
      for (int c = 0; c < kNC; ++c) if (cur[1 + kNC + c] >= thr) { flag = wThr; break; }
      if (flag == 0.0 && 0.0f >= thr) flag = wThr;         // frame[13] of the reference's buffer is 0
      lc = wLocal * flag;
    }
    if (t == 0) {
      if (lane < kNS) { cost[lane] = lc; paths[0][lane * kVB] = lane; }
      __syncthreads();
    } else {
      const float *prev = cur - 21;
      const bool vv = valid && si < kNC && sj < kNC;
      float fa = 0.0f, fb2 = 0.0f;
      if (vv) { fa = prev[1 + sj]; fb2 = cur[1 + si]; }
      const bool zero = vv && (fa == 0 || fb2 == 0);
      const bool modr = vv && !zero;
      const bool mod0 = valid && ((si == kNC) != (sj == kNC));
      const double r = modr ? log((double)(fb2 / fa)) : 0.0;
      const unsigned long long mask = __ballot(modr || mod0);
      const unsigned long long lower = mask & ((1ull << lane) - 1ull);
      const int src = lower ? 63 - __clzll((long long)lower) : 0;
      const double rprev = __shfl(r, src);
      const double lastc = lower ? rprev : lastChange;
      double c = 1.0;
      if (vv) c = zero ? 999.0 : wTvv * fabs(r) + wTvvd * fabs(r - lastc);
      else if (mod0) c = wTvuv;
      if (mask) lastChange = __shfl(r, 63 - __clzll((long long)mask));
      const double tot = valid ? c + cost[sj] : 0.0;
      const int gb = si * kNS;
      double mc = __shfl(tot, gb);
      int ms = 0;
#pragma unroll
      for (int jj = 1; jj < kNS; ++jj) {
        const double v = __shfl(tot, gb + jj);
        if (v < mc) { mc = v; ms = jj; }
      }
      const double lci = __shfl(lc, si);
      __syncthreads();
      if (valid && sj == 0) { cost[si] = mc + lci; msel[si] = ms; }
      __syncthreads();
      const int nb = pathBuf ^ 1;
      for (int idx = lane; idx < kNS * kVB; idx += 64) {
        const int ii = idx / kVB, n = idx - ii * kVB;
        paths[nb][idx] = paths[pathBuf][msel[ii] * kVB + n];
      }
      __syncthreads();
      if (lane < kNS) paths[nb][lane * kVB + pathIdx % kVB] = lane;
      __syncthreads();
      pathBuf = nb;
    }
    pathIdx++;
    const int *Pp = paths[pathBuf];
    if (pathIdx - convIdx > kVB) {                         // forced decision for the oldest open frame
      int ms = 0;
      for (int i = 1; i < kNS; i++) if (cost[i] < cost[ms]) ms = i;
      convIdx++;
      if (lane == 0) emit(convIdx, Pp[ms * kVB + convIdx % kVB]);
    } else {                                               // decide up to where all paths agree
      const int n = convIdx + 1 + lane;
      bool match = false;
      int xs = 0;
      if (n < pathIdx) {
        xs = Pp[n % kVB];
        match = true;
        for (int i = 1; i < kNS; i++) if (Pp[i * kVB + n % kVB] != xs) match = false;
      }
      const unsigned long long mm = __ballot(match);
      const int nlead = __ffsll((long long)~mm) - 1;       // lanes >= 31 never match: ~mm != 0
      if (lane < nlead) emit(n, xs);
      convIdx += nlead;
    }
    __syncthreads();
  }
  // flushTrellis at end of input
  {
    int ms = 0;
    for (int i = 1; i < kNS; i++) if (cost[i] < cost[ms]) ms = i;
    const int *Pp = paths[pathBuf];
    for (int n = convIdx + 1 + lane; n < pathIdx; n += 64) emit(n, Pp[ms * kVB + n % kVB]);
  }
}

int f0_tile_frames() { return kTileFrames; }

hipError_t launch_f0(const LldParams &P, const F0Params &Q, int max_blocks, float *d_out, int64_t ld_out, hipStream_t s) {
  if (P.total_frames <= 0) return hipSuccess;
  if (Q.Nfft != kNfftF0 || Q.K != kK || Q.n_harm > 17) return hipErrorInvalidValue;    // 60 ms @ 16 kHz geometry only
  const size_t lds = f0_shared_bytes(Q.N) + kFrameBytes * kW * kWaves;
  const void *fn = reinterpret_cast<const void *>(&lld_f0_frame);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  unsigned grid = (unsigned)((P.n_tiles + kWaves - 1) / kWaves);
  if (grid > (unsigned)max_blocks) grid = (unsigned)max_blocks;         // persistent: one workgroup per CU
  hipLaunchKernelGGL(lld_f0_frame, dim3(grid), dim3(kWaves * 64), lds, s, P, Q);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lld_f0_viterbi, dim3((unsigned)P.n_utt), dim3(64), 0, s, P.frame_off, P.n_utt, Q, d_out, ld_out);
  return hipGetLastError();
}

}  // namespace smilehip
