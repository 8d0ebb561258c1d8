// Block-level (256-thread workgroup) building blocks shared by the reference-order kernels:
// reductions, the inverse real FFT cAcf needs, and cPitchACF's per-frame analysis.
#pragma once
#include <hip/hip_runtime.h>

#include "lld_device.hpp"

namespace smilehip {

// block-wide reductions over 256 threads through LDS scratch (all threads get the result)
__device__ __forceinline__ double block_sum(double v, double *scr) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
  __syncthreads();
  return scr[0] + scr[1] + scr[2] + scr[3];
}
__device__ __forceinline__ double block_max(double v, double *scr) {
  for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = w > v ? w : v; }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
  __syncthreads();
  double m = scr[0];
  for (int i = 1; i < 4; ++i) m = scr[i] > m ? scr[i] : m;
  return m;
}
__device__ __forceinline__ int block_sum_i(int v, int *scr) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
  __syncthreads();
  return scr[0] + scr[1] + scr[2] + scr[3];
}
__device__ __forceinline__ int block_min_i(int v, int *scr) {
  for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = w < v ? w : v; }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
  __syncthreads();
  int m = scr[0];
  for (int i = 1; i < 4; ++i) m = scr[i] < m ? scr[i] : m;
  return m;
}

// Inverse of the packed real FFT for a purely real spectrum R[0..M] (what cAcf feeds
// Ooura's rdft(n,-1), fftsg.c:103-135):  a[k] = R0/2 + R_M (-1)^k / 2 + sum_j R_j cos(2 pi jk/n).
// Computed as half the forward DFT of the even extension s[j] = s[n-j] = R_j, through the
// same half-length complex FFT + untangle the forward transform uses.
__device__ void irfft_even(const float *R, float *re, float *im, int M, int logM, const float2 *tw_half,
                           const float2 *tw_full, float *out, float inv_norm, bool take_abs) {
  const int n = 2 * M;
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    const int n0 = 2 * i, n1 = 2 * i + 1;
    const float v0 = R[n0 <= M ? n0 : n - n0];
    const float v1 = R[n1 <= M ? n1 : n - n1];
    const int r = (int)(__brev((unsigned)i) >> (32 - logM));
    re[r] = v0;
    im[r] = v1;
  }
  __syncthreads();
  block_cfft_radix2(re, im, M, tw_half);
  for (int k = threadIdx.x; k < M; k += blockDim.x) {
    const float a = 0.5f * untangle_bin(re, im, M, k, tw_full).x;
    const float v = a / inv_norm;                       // acf.cpp:321-325: (FLOAT_DMEM)data / (FLOAT_DMEM)Nsrc
    out[k] = take_abs ? fabsf(v) : v;
  }
  __syncthreads();
}


// R10 cPitchACF::processVector, per-frame part (pitchACF.cpp:137-192): voicing probability from
// the ACF (voicingProb, :249-284) and the index of the first cepstral peak above
// 0.6 * (max + mean|.|) (pitchPeak, :286-310). acf / cep: n values each, in LDS or global.
// Every thread returns the same (voicing, maxIdx).
__device__ __forceinline__ void pitchacf_frame(const float *acf, const float *cep, int n, double fsSec, double maxPitch,
                                               double *scr, int *iscr, double &voicing, int &max_idx, double &Tsamp_out) {
  const double Nd = (double)(2 * n);
  const double Tsamp = fsSec / Nd;
  Tsamp_out = Tsamp;
  const int preskip = (maxPitch <= 0.0) ? 0 : (int)(1.0 / (maxPitch * Tsamp));
  double vmax = acf[n - 1];
  for (int i = 1 + threadIdx.x; i < n; i += blockDim.x)
    if (i >= preskip && (acf[i] > vmax) && (acf[i - 1] < acf[i])) vmax = acf[i];
  // (the reference's running-max test "a[i] > max" only ever raises max, so the result is the
  //  maximum over the qualifying set; taking it in parallel gives the same value)
  vmax = block_max(vmax, scr);
  voicing = (acf[0] > 0.0f) ? vmax / (double)acf[0] : 0.0;
  const int skip = preskip + 1;
  double csum = 0.0, cmax = cep[n - 1];
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double b = cep[i];
    csum += fabs(b);
    if (i >= skip && b > cmax) cmax = b;
  }
  csum = block_sum(csum, scr) / n;
  cmax = block_max(cmax, scr);
  const double thr = (cmax + csum) * 0.6;
  int first = 1 << 30;
  for (int i = skip + 1 + threadIdx.x; i < n - 1; i += blockDim.x)
    if ((double)cep[i] > thr && (cep[i - 1] < cep[i]) && (cep[i] > cep[i + 1])) { first = i; break; }
  first = block_min_i(first, iscr);
  max_idx = (first == (1 << 30)) ? 0 : first;
}

}  // namespace smilehip
