// Per-component operators for the option sets of SURVEY.md 8(a)'s components that no shipped BASELINE config uses but the
// components have (missing 3): cTransformFFT with inverse = 1, every output mode of cFFTmagphase (normalise, power,
// dBpsd, phase, joinMagphase), cMZcr's mcr / amax / maxmin / dc. Same conventions as lld_stage_kernels.hip: frame-major rows with
// leading dimensions, the reference's float expressions in the reference's order.
#include <hip/hip_runtime.h>

#include "glibc_float.hpp"
#include "lld_blocks.hpp"
#include "lld_blocks_compare.hpp"
#include "lld_ooura.hpp"
#include "lld_stage.hpp"

namespace smilehip {

namespace {
struct BlockG3 {
  __device__ static __forceinline__ int tid() { return threadIdx.x; }
  __device__ static __forceinline__ int size() { return blockDim.x; }
  __device__ static __forceinline__ void sync() { __syncthreads(); }
};
inline unsigned nblk3(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }
}  // namespace

// cTransformFFT::processVector with inverse = 1 (transformFft.cpp:196-216): x = the Nsrc input values (no padding rule: a shorter
// input leaves the rest of x as it was -- refused by the caller), rdft(Ndst, -1, x), dst[i] = x[i] * (2 / Ndst) as floats.
__global__ void __launch_bounds__(256) k_irfft_oo(const float *src, int64_t lds, float *dst, int64_t ldd, const OouraTab T) {
  extern __shared__ __attribute__((aligned(16))) float smem3[];
  float2 *z = reinterpret_cast<float2 *>(smem3);
  const int M = T.M;
  const float *a = src + (int64_t)blockIdx.x * lds;
  ooura_inverse<BlockG3>(z, T, [&](int e) { return make_float2(a[2 * e], a[2 * e + 1]); });
  float *o = dst + (int64_t)blockIdx.x * ldd;
  const float norm = (float)2.0 / (float)(2 * M);
  for (int i = threadIdx.x; i < 2 * M; i += blockDim.x) o[i] = ooura_inverse_out(z, T, i) * norm;
}

// cFFTmagphase::processVector (fftmagphase.cpp:215-287). flags: 1 magnitude, 2 phase, 4 normalise, 8 power, 16 dBpsd.
// Output: the magnitude field (K = Nfft/2 + 1 values) if asked for, directly followed by the phase field if asked for (what
// joinMagphase = 1 produces; the reference's separate-fields mode for both at once is broken upstream, :268-276, and is not offered).
// log10 / atan2 on FLOAT_DMEM arguments are log10f / atan2f: glibc's algorithms (glibc_float.hpp), the reference's bits.
__global__ void k_fftmagphase(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int Nfft, int flags,
                              float dBpnorm, float mindBp) {
  const int K = Nfft / 2 + 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nF * K) return;
  const int64_t f = i / K;
  const int k = (int)(i - f * K);
  const float *a = src + f * lds;
  float *o = dst + f * ldd;
  const bool edge = (k == 0 || k == K - 1);
  const float re = (k == 0) ? a[0] : (k == K - 1 ? a[1] : a[2 * k]);
  const float im = edge ? 0.0f : a[2 * k + 1];
  const float fN = (float)Nfft;
  int off = 0;
  if (flags & 1) {
    const bool normalise = flags & 4, power = flags & 8, dBpsd = flags & 16;
    float m;
    if (!dBpsd && !normalise && !power) m = edge ? fabsf(re) : sqrtf(re * re + im * im);
    else if (!dBpsd && normalise && !power) m = ((float)1.0 / fN) * (edge ? fabsf(re) : sqrtf(re * re + im * im));
    else if (!dBpsd && normalise && power) {
      if (k == 0) { m = ((float)1.0 / fN) * re; m *= m; }                 // (:233-234: no fabs on bin 0; squared anyway)
      else if (k == K - 1) { m = ((float)1.0 / fN) * fabsf(re); m *= m; }
      else m = ((float)1.0 / (fN * fN)) * (re * re + im * im);
    } else if (!dBpsd && !normalise && power) {
      if (edge) { m = fabsf(re); m *= m; }
      else m = re * re + im * im;
    } else {                                                            // dBpsd (:251-257)
      float v;
      if (edge) v = dBpnorm + (float)20.0 * glibc_log10f(((float)1.0 / fN) * fabsf(re));
      else v = dBpnorm + (float)10.0 * glibc_log10f(((float)1.0 / (fN * fN)) * (re * re + im * im));
      m = v > mindBp ? v : mindBp;                                       // MAX(mindBp, v)
    }
    o[k] = m;
    off = K;
  }
  if (flags & 2) o[off + k] = edge ? ((re >= 0) ? (float)0 : (float)M_PI) : glibc_atan2f(im, re);
}

// cMZcr::processVector (mzcr.cpp:108-150). flags: 1 zcr, 2 mcr, 4 amax, 8 maxmin, 16 dc; outputs in that order (maxmin: max, min).
// `mean` is a FLOAT_DMEM accumulator over src[0 .. N-2] (one float addition after the other: lane 0 walks the frame's LDS copy),
// nmc starts at 4.0 as in the reference. One wave per frame.
__global__ void __launch_bounds__(64) k_mzcr(const float *src, int64_t lds, int N, int flags, float *dst, int64_t ldd) {
  extern __shared__ __attribute__((aligned(16))) float s_x[];
  const int lane = threadIdx.x;
  const float *x = src + (int64_t)blockIdx.x * lds;
  for (int i = lane; i < N; i += 64) s_x[i] = x[i];
  __syncthreads();
  float mean = 0.0f;
  if (flags & (1 | 2 | 16)) {
    if (lane == 0) {
      mean = s_x[0];
      for (int i = 1; i < N - 1; ++i) mean += s_x[i];
      mean /= (float)N;
    }
    mean = __shfl(mean, 0);
  }
  int nz = 0, nm = 0;
  for (int i = 1 + lane; i < N - 1; i += 64) {
    const float a = s_x[i - 1], b = s_x[i], c = s_x[i + 1];
    if (((a * c <= 0.0f) && (b == 0.0f)) || (a * b < 0.0f)) nz++;
    if (flags & 2) {
      const float am = a - mean, bm = b - mean, cm = c - mean;
      if (((am * cm <= 0.0f) && (bm == 0.0f)) || (am * bm < 0.0f)) nm++;
    }
  }
  float mx = s_x[0], mn = s_x[0];
  for (int i = 1 + lane; i < N; i += 64) { const float v = s_x[i]; if (v < mn) mn = v; if (v > mx) mx = v; }
  for (int of = 32; of > 0; of >>= 1) {
    nz += __shfl_down(nz, of); nm += __shfl_down(nm, of);
    const float omx = __shfl_down(mx, of), omn = __shfl_down(mn, of);
    if (omx > mx) mx = omx;
    if (omn < mn) mn = omn;
  }
  if (lane == 0) {
    float *o = dst + (int64_t)blockIdx.x * ldd;
    int n = 0;
    if (flags & 1) o[n++] = (float)nz / (float)N;                        // nzc counts in float (exact below 2^24), / Nsrc
    if (flags & 2) o[n++] = ((float)4.0 + (float)nm) / (float)N;
    if (flags & 4) o[n++] = (fabsf(mn) > fabsf(mx)) ? fabsf(mn) : fabsf(mx);
    if (flags & 8) { o[n++] = mx; o[n++] = mn; }
    if (flags & 16) o[n++] = mean;
  }
}

// cPitchACF::voicingProb's second result (pitchACF.cpp:249-283): the zero- or mean-crossing rate of the ACF, whichever count is
// greater, over n -- what the voiceQual output needs. `mean` is a double accumulator in index order, a[skip] counted twice as in
// the reference (:257-266); lane 0 walks it, the counts are wave-parallel. One wave per frame.
__global__ void __launch_bounds__(64) k_pitchacf_zcr(const float *src, int64_t lds, int n, int skip, double *zcr_out) {
  const float *a = src + (int64_t)blockIdx.x * lds;
  const int lane = threadIdx.x;
  double mean = 0.0;
  if (lane == 0) {
    mean = a[skip];
    for (int i = (skip > 1 ? skip : 1); i < n; ++i) mean += a[i];
    mean /= (double)(n - skip + 1);
  }
  mean = __shfl(mean, 0);
  int zc = 0, mc = 0;
  for (int i = 1 + lane; i < n; i += 64) {
    if (a[i - 1] * a[i] < 0) zc++;
    if (((double)a[i - 1] - mean) * ((double)a[i] - mean) < 0) mc++;
  }
  for (int of = 32; of > 0; of >>= 1) { zc += __shfl_down(zc, of); mc += __shfl_down(mc, of); }
  if (lane == 0) zcr_out[blockIdx.x] = (mc > zc) ? (double)mc / (double)n : (double)zc / (double)n;
}
// cMelspec::processVector, forward (melspec.cpp:519-570), with the filter tables GIVEN -- whatever spectral scale (mel, bark, bark_schroed,
// bark_speex, semitone, log, lin), bandwidth method or bank type cMelspec::computeFilters (:186-451) built them for:
//   dense == 0: the standard bank: coef[K] = the rising-slope weight of bin n, chanmap[K] = its band - 1 (-3: unused). Band m gets
//               (float)((double)x[n] coef[n]) from the bins mapped to it and x[n] - that from the bins mapped to m - 1, bin after bin;
//   dense == 1: HFCC / custom-bandwidth banks: coef[n_bands x K], chanmap[2 n_bands] = first / last bin of a band;
//               dst[m] += (float)((double)x[n] (double)coef[m K + n]).
// x = src or src^2 (usePower); htk_scale multiplies the band afterwards (32767 or 32767^2, :556-567; 1 = none). Lane = band, every band
// walks its bins in ascending order: the float sums are the reference's.
__global__ void __launch_bounds__(64) k_melspec_table(const float *src, int64_t lds, int K, int nB, int dense, const float *coef,
                                                      const int32_t *chanmap, int nLoF, int nHiF, int use_power, float htk_scale,
                                                      float *dst, int64_t ldd) {
  extern __shared__ __attribute__((aligned(16))) float s_p[];
  const float *a = src + (int64_t)blockIdx.x * lds;
  for (int n = threadIdx.x; n < K; n += 64) { const float v = a[n]; s_p[n] = use_power ? v * v : v; }
  __syncthreads();
  for (int m = threadIdx.x; m < nB; m += 64) {
    float acc = 0.0f;
    if (dense) {
      const int n1 = chanmap[2 * m], n2 = chanmap[2 * m + 1];
      for (int n = (n1 > nLoF ? n1 : nLoF); n <= n2 && n < nHiF; ++n) acc += (float)((double)s_p[n] * (double)coef[(int64_t)m * K + n]);
    } else {
      for (int n = nLoF; n < nHiF; ++n) {
        const int c = chanmap[n];
        if (c == m) acc += (float)((double)s_p[n] * (double)coef[n]);
        else if (c == m - 1 && c > -2) acc += s_p[n] - (float)((double)s_p[n] * (double)coef[n]);
      }
    }
    if (htk_scale != 1.0f) acc *= htk_scale;
    dst[(int64_t)blockIdx.x * ldd + m] = acc;
  }
}
hipError_t stage_melspec_table(const float *src, int64_t lds, int K, int nB, int dense, const float *coef, const int32_t *chanmap, int nLoF,
                               int nHiF, int use_power, float htk_scale, float *dst, int64_t ldd, int64_t nF, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  const size_t bytes = sizeof(float) * (size_t)((K + 3) & ~3);
  if (bytes > 48 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_melspec_table, dim3((unsigned)nF), dim3(64), bytes, s, src, lds, K, nB, dense, coef, chanmap, nLoF, nHiF, use_power,
                     htk_scale, dst, ldd);
  return hipGetLastError();
}

// cMelspec::processVector, inverse = 1 (src/lldcore/melspec.cpp:466-516): n_src bands -> K spectrum bins through the standard bank's
// tables (coef[K], chanmap[K] as computeFilters builds them with the roles swapped, :186-199). Lane = bin:
//   x[m] = src[m] / htk_div (32767 or 32767^2, :468-479; 1 = none);  m = chanmap[n]
//   dst[n] = x[m] coef[n] (+ x[m + 1] (1 - coef[n]) unless m is the last band) for nLoF <= n < min(K, nHiF) and m >= 0, else 0
//   usePower: dst[n] = dst[n] > 0 ? sqrt(dst[n]) : 0
__global__ void __launch_bounds__(64) k_melspec_inverse_table(const float *src, int64_t lds, int n_src, int K, const float *coef,
                                                              const int32_t *chanmap, int nLoF, int nHiF, int use_power, float htk_div,
                                                              float *dst, int64_t ldd) {
  extern __shared__ __attribute__((aligned(16))) float s_x[];
  const float *a = src + (int64_t)blockIdx.x * lds;
  for (int m = threadIdx.x; m < n_src; m += 64) s_x[m] = (htk_div != 1.0f) ? a[m] / htk_div : a[m];
  __syncthreads();
  const int hi = K < nHiF ? K : nHiF;
  for (int n = threadIdx.x; n < K; n += 64) {
    float acc = 0.0f;
    if (n >= nLoF && n < hi) {
      const int m = chanmap[n];
      if (m > -1 && m < n_src) {
        acc += s_x[m] * coef[n];
        if (m < n_src - 1) acc += s_x[m + 1] * (1.0f - coef[n]);
      }
    }
    if (use_power) acc = (acc > 0.0f) ? sqrtf(acc) : 0.0f;
    dst[(int64_t)blockIdx.x * ldd + n] = acc;
  }
}
hipError_t stage_melspec_inverse_table(const float *src, int64_t lds, int n_src, int K, const float *coef, const int32_t *chanmap, int nLoF,
                                       int nHiF, int use_power, float htk_div, float *dst, int64_t ldd, int64_t nF, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  const size_t bytes = sizeof(float) * (size_t)((n_src + 3) & ~3);
  if (bytes > 48 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_melspec_inverse_table, dim3((unsigned)nF), dim3(64), bytes, s, src, lds, n_src, K, coef, chanmap, nLoF, nHiF,
                     use_power, htk_div, dst, ldd);
  return hipGetLastError();
}

hipError_t stage_pitchacf_zcr(const float *src, int64_t lds, int64_t nF, int n, int skip, double *zcr, hipStream_t s) {
  if (nF > 0) hipLaunchKernelGGL(k_pitchacf_zcr, dim3((unsigned)nF), dim3(64), 0, s, src, lds, n, skip, zcr);
  return hipGetLastError();
}

hipError_t stage_irfft_oo(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int Nfft, const OouraTab &T, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  const size_t bytes = sizeof(float) * (size_t)Nfft;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_irfft_oo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_irfft_oo, dim3((unsigned)nF), dim3(256), bytes, s, src, lds, dst, ldd, T);
  return hipGetLastError();
}
hipError_t stage_fftmagphase(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int Nfft, int flags, float dBpnorm,
                             float mindBp, hipStream_t s) {
  const int64_t n = nF * (Nfft / 2 + 1);
  if (n > 0) hipLaunchKernelGGL(k_fftmagphase, dim3(nblk3(n, 256)), dim3(256), 0, s, src, lds, dst, ldd, nF, Nfft, flags, dBpnorm, mindBp);
  return hipGetLastError();
}
hipError_t stage_mzcr(const float *src, int64_t lds, int N, int64_t nF, int flags, float *dst, int64_t ldd, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  const size_t bytes = sizeof(float) * (size_t)((N + 3) & ~3);
  if (bytes > 150 * 1024) return hipErrorInvalidValue;
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mzcr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_mzcr, dim3((unsigned)nF), dim3(64), bytes, s, src, lds, N, flags, dst, ldd);
  return hipGetLastError();
}

}  // namespace smilehip
