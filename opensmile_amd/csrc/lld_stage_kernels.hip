// Per-component batched kernels: one reference component at a time over a
// block of frames (plugin per-component mode, stage-level parity tests).
// Same device functions as the fused kernels, so results are identical.
#include <hip/hip_runtime.h>

#include "lld_blocks.hpp"
#include "lld_device.hpp"
#include "lld_ooura.hpp"
#include "lld_stage.hpp"

namespace smilehip {

__global__ void k_pcm16_to_float(const int16_t *pcm, int64_t n, float *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pcm16_to_float(pcm[i]);
}

// cWinToVecProcessor::myTick + cFramer::doProcess (winToVecProcessor.cpp:983, :1037-1052; framer.cpp): frame f = samples [f * step, f * step + N)
__global__ void k_frame_rows(const float *samples, int64_t N, int64_t step, int64_t nF, float *dst, int64_t ldd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nF * N) return;
  const int64_t f = i / N, n = i - f * N;
  dst[f * ldd + n] = samples[f * step + n];
}

// cVectorPreemphasis::processVector (vectorPreemphasis.cpp:89-107)
__global__ void k_preemphasis(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int64_t N,
                              float k, float omk, int de) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nF * N) return;
  const int64_t f = i / N, n = i - f * N;
  const float *x = src + f * lds;
  float y;
  if (n == 0) y = omk * x[0];
  else y = de ? (x[n] + k * x[n - 1]) : (x[n] - k * x[n - 1]);
  dst[f * ldd + n] = y;
}

// cWindower::processVector (windower.cpp:221-229)
__global__ void k_window(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int64_t N,
                         const float *w, float offset) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nF * N) return;
  const int64_t f = i / N, n = i - f * N;
  dst[f * ldd + n] = src[f * lds + n] * w[n] + offset;
}

// cTransformFFT::processVector forward (transformFft.cpp:165-223), packed as
// Ooura's rdft does (fftsg.c:103-135): a[0]=Re X0, a[1]=Re X[n/2],
// a[2k]=Re Xk, a[2k+1]=+sum x sin(2 pi jk/n) = -Im of the standard DFT.
__global__ void __launch_bounds__(256) k_rfft(const float *src, int64_t lds, float *dst, int64_t ldd, int N,
                                              int Nfft, int pad_left, const float2 *tw_half,
                                              const float2 *tw_full) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int M = Nfft >> 1;
  float *re = smem, *im = smem + M;
  int logM = 0;
  while ((1 << logM) < M) ++logM;
  const float *x = src + (int64_t)blockIdx.x * lds;
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    const int n0 = 2 * i - pad_left, n1 = n0 + 1;
    const float v0 = (n0 >= 0 && n0 < N) ? x[n0] : 0.0f;
    const float v1 = (n1 >= 0 && n1 < N) ? x[n1] : 0.0f;
    const int r = (int)(__brev((unsigned)i) >> (32 - logM));
    re[r] = v0;
    im[r] = v1;
  }
  __syncthreads();
  block_cfft_radix2(re, im, M, tw_half);
  float *o = dst + (int64_t)blockIdx.x * ldd;
  for (int k = threadIdx.x; k <= M; k += blockDim.x) {
    const float2 X = untangle_bin(re, im, M, k, tw_full);
    if (k == 0) o[0] = X.x;
    else if (k == M) o[1] = X.x;
    else { o[2 * k] = X.x; o[2 * k + 1] = -X.y; }
  }
}

// The same operator on the reference-order transform (lld_ooura.hpp): bit-identical to rdft() of fftsg.c:322-363.
__global__ void __launch_bounds__(256) k_rfft_oo(const float *src, int64_t lds, float *dst, int64_t ldd, int N,
                                                 int pad_left, const OouraTab T) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2 *z = reinterpret_cast<float2 *>(smem);
  const int M = T.M;
  const float *x = src + (int64_t)blockIdx.x * lds;
  ooura_forward<BlockG>(z, T, [&](int i) {
    const int n0 = 2 * i - pad_left, n1 = n0 + 1;
    return make_float2((n0 >= 0 && n0 < N) ? x[n0] : 0.0f, (n1 >= 0 && n1 < N) ? x[n1] : 0.0f);
  });
  float *o = dst + (int64_t)blockIdx.x * ldd;
  for (int k = threadIdx.x; k <= M; k += blockDim.x) {
    const float2 X = ooura_bin(z, T, k);
    if (k == 0) o[0] = X.x;
    else if (k == M) o[1] = X.x;
    else { o[2 * k] = X.x; o[2 * k + 1] = -X.y; }
  }
}

// FFT 512 / 1024: one wave per frame, the register form of the same network (lld_ooura_wave.hpp), four frames per workgroup
__global__ void __launch_bounds__(256) k_rfft_oo_wave(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int N,
                                                      int pad_left, const OouraTab T) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t f = (int64_t)blockIdx.x * 4 + wave;
  if (f >= nF) return;
  const int M = T.M;
  float2 *z = reinterpret_cast<float2 *>(smem) + (size_t)wave * M;
  const float *x = src + f * lds;
  oo_wave_forward(z, T, lane, [&](int i) {
    const int n0 = 2 * i - pad_left, n1 = n0 + 1;
    return make_float2((n0 >= 0 && n0 < N) ? x[n0] : 0.0f, (n1 >= 0 && n1 < N) ? x[n1] : 0.0f);
  });
  float *o = dst + f * ldd;
  for (int k = lane; k <= M; k += 64) {
    const float2 X = oo_wave_bin(z, T, k);
    if (k == 0) o[0] = X.x;
    else if (k == M) o[1] = X.x;
    else { o[2 * k] = X.x; o[2 * k + 1] = -X.y; }
  }
}

// cFFTmagphase::processVector, magnitude branch (fftmagphase.cpp:215-221)
__global__ void k_fftmag(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int Nfft) {
  const int K = Nfft / 2 + 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nF * K) return;
  const int64_t f = i / K;
  const int k = (int)(i - f * K);
  const float *a = src + f * lds;
  float m;
  if (k == 0) m = fabsf(a[0]);
  else if (k == K - 1) m = fabsf(a[1]);
  else m = sqrtf(a[2 * k] * a[2 * k] + a[2 * k + 1] * a[2 * k + 1]);
  dst[f * ldd + k] = m;
}

// cMelspec::processVector (melspec.cpp:519-570); one workgroup per frame so
// the squared spectrum is staged once in LDS.
__global__ void __launch_bounds__(256) k_melspec(const float *src, int64_t lds, float *dst, int64_t ldd, int K,
                                                 int n_bands, int use_power, const float *coef,
                                                 const int32_t *rng, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const float *m = src + (int64_t)blockIdx.x * lds;
  for (int k = threadIdx.x; k < K; k += blockDim.x) smem[k] = use_power ? m[k] * m[k] : m[k];
  __syncthreads();
  for (int b = threadIdx.x; b < n_bands; b += blockDim.x)
    dst[(int64_t)blockIdx.x * ldd + b] = mel_band_exact(smem, coef, rng, b, scale);
}

// cMfcc::processVector (mfcc.cpp:239-273)
__global__ void __launch_bounds__(64) k_mfcc(const float *src, int64_t lds, float *dst, int64_t ldd, int n_bands,
                                             int n_mfcc, const float *rows, const float *gain, float melfloor,
                                             float log_floor) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const float *m = src + (int64_t)blockIdx.x * lds;
  for (int b = threadIdx.x; b < n_bands; b += blockDim.x) smem[b] = log_mel(m[b], melfloor, log_floor);
  __syncthreads();
  for (int r = threadIdx.x; r < n_mfcc; r += blockDim.x)
    dst[(int64_t)blockIdx.x * ldd + r] = dct_coeff(smem, rows + r * n_bands, n_bands, gain[r]);
}

// cMfcc::processVector with inverse = 1 (mfcc.cpp:184-235): one thread per (frame, band). Coefficient i = first + i0 sits at input
// position pos(i) -- HTK mode with firstMfcc = 0: c0 last -- which is also its row of `rows` (rows are by OUTPUT position of the
// forward transform); lifter[i0] is the reference's sintable. The sum runs over i in ascending order, a float chain, every product
// ((c x cos) x correction) x factor as written there.
struct MfccInverse { int first, last, n_bands, htk, do_log; float factor; float lifter[64]; };
__global__ void __launch_bounds__(64) k_mfcc_inverse(const float *src, int64_t lds, float *dst, int64_t ldd, const float *rows, MfccInverse Q) {
  const float *c = src + (int64_t)blockIdx.x * lds;
  for (int m = threadIdx.x; m < Q.n_bands; m += blockDim.x) {
    float acc = 0.0f;
    for (int i = Q.first; i <= Q.last; ++i) {
      const int i0 = i - Q.first;
      const int pos = (Q.htk && Q.first == 0) ? (i == 0 ? Q.last : i0 - 1) : i0;
      const float v = c[pos] / Q.lifter[i0];
      const float corr = (i == 0) ? 0.5f : 1.0f;
      acc += v * rows[pos * Q.n_bands + m] * corr * Q.factor;
    }
    dst[(int64_t)blockIdx.x * ldd + m] = Q.do_log ? glibc_expf(acc) : acc;
  }
}
hipError_t stage_mfcc_inverse(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int n_bands, int first, int last, int htk,
                              int do_log, const float *rows, const float *lifter, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  MfccInverse Q;
  Q.first = first; Q.last = last; Q.n_bands = n_bands; Q.htk = htk; Q.do_log = do_log;
  Q.factor = (float)sqrt((double)2.0 / (double)(n_bands));
  for (int i = 0; i < 64; ++i) Q.lifter[i] = (i <= last - first) ? lifter[i] : 1.0f;
  hipLaunchKernelGGL(k_mfcc_inverse, dim3((unsigned)nF), dim3(64), 0, s, src, lds, dst, ldd, rows, Q);
  return hipGetLastError();
}

// R0, every sample format of smilePcm_convertSamples (smileUtil.c:2500-2627): one thread per
// output sample (mixdown) or per (sample, channel). The divisions are IEEE float divisions in
// the reference's order: (sum / nChan) / full-scale.
__device__ __forceinline__ float pcm_sample(const unsigned char *buf, int64_t idx, int n_bps, int n_bits) {
  switch (n_bps) {
    case 1: return (float)reinterpret_cast<const int8_t *>(buf)[idx];
    case 2: return (float)reinterpret_cast<const int16_t *>(buf)[idx];
    case 3: {
      uint32_t is = 0;
      is |= (uint32_t)(buf[idx * 3]) << 8;
      is |= (uint32_t)(buf[idx * 3 + 1]) << 16;
      is |= (uint32_t)(buf[idx * 3 + 2]) << 24;
      return (float)((int32_t)is >> 8);
    }
    default: {
      const int32_t v = reinterpret_cast<const int32_t *>(buf)[idx];
      return (n_bits == 24) ? (float)(v & 0xFFFFFF) : (float)v;
    }
  }
}
__device__ __forceinline__ float pcm_full_scale(int n_bps, int n_bits) {
  if (n_bps == 1) return (float)127.0;
  if (n_bps == 2) return (float)32767.0;
  if (n_bps == 3 || n_bits == 24) return (float)(32767.0 * 256.0);
  return (float)2147483647.0;
}
__global__ void k_pcm_convert(const unsigned char *buf, int n_bps, int n_bits, int n_chan, int mixdown, int64_t n,
                              float *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float fs = pcm_full_scale(n_bps, n_bits);
  if (mixdown) {
    if (i >= n) return;
    float tmp = 0.0f;
    for (int c = 0; c < n_chan; c++) tmp += pcm_sample(buf, i * n_chan + c, n_bps, n_bits);
    out[i] = (tmp / (float)n_chan) / fs;
  } else {
    if (i >= n * n_chan) return;
    out[i] = pcm_sample(buf, i, n_bps, n_bits) / fs;
  }
}

// smilePcm_convertFloatSamples (smileUtil.c:2629-2690): 32-bit IEEE float samples; the mono mix-down adds the channels to a
// float zero in channel order and divides by the channel count (a single channel goes through the same two operations)
__global__ void k_pcm_convert_float(const float *buf, int n_chan, int mixdown, int64_t n, float *out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (mixdown) {
    if (i >= n) return;
    float tmp = 0.0f;
    for (int c = 0; c < n_chan; c++) tmp += buf[i * n_chan + c];
    out[i] = tmp / (float)n_chan;
  } else {
    if (i >= n * n_chan) return;
    out[i] = buf[i];
  }
}

static inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

// cHtkSink's byte order (htkSink.cpp:183-202), 16 bytes per thread where the pointers allow
__global__ void k_htk_rows_be(const uint32_t *src, int64_t n, uint32_t *dst, int vec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    if (4 * i + 3 < n) {
      uint4 v = reinterpret_cast<const uint4 *>(src)[i];
      v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y); v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
      reinterpret_cast<uint4 *>(dst)[i] = v;
    } else {
      for (int64_t k = 4 * i; k < n; ++k) dst[k] = __builtin_bswap32(src[k]);
    }
  } else if (i < n) {
    dst[i] = __builtin_bswap32(src[i]);
  }
}
hipError_t stage_htk_rows_be(const float *src, int64_t n, uint32_t *dst, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const int64_t work = vec ? (n + 3) / 4 : n;
  hipLaunchKernelGGL(k_htk_rows_be, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(src), n, dst, vec ? 1 : 0);
  return hipGetLastError();
}
hipError_t stage_pcm16(const int16_t *pcm, int64_t n, float *out, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_pcm16_to_float, dim3(nblk(n, 256)), dim3(256), 0, s, pcm, n, out);
  return hipGetLastError();
}
hipError_t stage_pcm_convert(const void *buf, int n_bps, int n_bits, int n_chan, int mixdown, int64_t n, float *out,
                             hipStream_t s) {
  const int64_t work = mixdown ? n : n * n_chan;
  if (work > 0)
    hipLaunchKernelGGL(k_pcm_convert, dim3(nblk(work, 256)), dim3(256), 0, s, reinterpret_cast<const unsigned char *>(buf),
                       n_bps, n_bits, n_chan, mixdown, n, out);
  return hipGetLastError();
}
hipError_t stage_pcm_convert_float(const float *buf, int n_chan, int mixdown, int64_t n, float *out, hipStream_t s) {
  const int64_t work = mixdown ? n : n * n_chan;
  if (work > 0) hipLaunchKernelGGL(k_pcm_convert_float, dim3(nblk(work, 256)), dim3(256), 0, s, buf, n_chan, mixdown, n, out);
  return hipGetLastError();
}
hipError_t stage_frame_rows(const float *samples, int64_t N, int64_t step, int64_t nF, float *dst, int64_t ldd, hipStream_t s) {
  if (nF * N > 0) hipLaunchKernelGGL(k_frame_rows, dim3(nblk(nF * N, 256)), dim3(256), 0, s, samples, N, step, nF, dst, ldd);
  return hipGetLastError();
}
hipError_t stage_preemph(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int64_t N, float k,
                         int de, hipStream_t s) {
  if (nF * N > 0)
    hipLaunchKernelGGL(k_preemphasis, dim3(nblk(nF * N, 256)), dim3(256), 0, s, src, lds, dst, ldd, nF, N, k,
                       1 - k, de);
  return hipGetLastError();
}
hipError_t stage_window(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int64_t N,
                        const float *w, float off, hipStream_t s) {
  if (nF * N > 0)
    hipLaunchKernelGGL(k_window, dim3(nblk(nF * N, 256)), dim3(256), 0, s, src, lds, dst, ldd, nF, N, w, off);
  return hipGetLastError();
}
hipError_t stage_rfft_oo(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int N, int Nfft,
                         int pad_left, const OouraTab &T, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  if (T.M == 256 || T.M == 512)
    hipLaunchKernelGGL(k_rfft_oo_wave, dim3((unsigned)((nF + 3) / 4)), dim3(256), 4 * sizeof(float) * (size_t)Nfft, s, src, lds,
                       dst, ldd, nF, N, pad_left, T);
  else
    hipLaunchKernelGGL(k_rfft_oo, dim3((unsigned)nF), dim3(256), sizeof(float) * (size_t)Nfft, s, src, lds, dst, ldd,
                       N, pad_left, T);
  return hipGetLastError();
}
hipError_t stage_rfft(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int N, int Nfft,
                      int pad_left, const float2 *twh, const float2 *twf, hipStream_t s) {
  if (nF > 0)
    hipLaunchKernelGGL(k_rfft, dim3((unsigned)nF), dim3(256), sizeof(float) * (size_t)Nfft, s, src, lds, dst, ldd,
                       N, Nfft, pad_left, twh, twf);
  return hipGetLastError();
}
hipError_t stage_fftmag(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int Nfft,
                        hipStream_t s) {
  const int64_t n = nF * (Nfft / 2 + 1);
  if (n > 0) hipLaunchKernelGGL(k_fftmag, dim3(nblk(n, 256)), dim3(256), 0, s, src, lds, dst, ldd, nF, Nfft);
  return hipGetLastError();
}
hipError_t stage_melspec(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int K, int n_bands,
                         int use_power, const float *coef, const int32_t *rng, float scale, hipStream_t s) {
  if (nF > 0)
    hipLaunchKernelGGL(k_melspec, dim3((unsigned)nF), dim3(256), sizeof(float) * (size_t)(K + 4), s, src, lds, dst,
                       ldd, K, n_bands, use_power, coef, rng, scale);
  return hipGetLastError();
}
hipError_t stage_mfcc(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int n_bands, int n_mfcc,
                      const float *rows, const float *gain, float melfloor, float log_floor, hipStream_t s) {
  if (nF > 0)
    hipLaunchKernelGGL(k_mfcc, dim3((unsigned)nF), dim3(64), sizeof(float) * (size_t)(n_bands + 4), s, src, lds,
                       dst, ldd, n_bands, n_mfcc, rows, gain, melfloor, log_floor);
  return hipGetLastError();
}

}  // namespace smilehip
