// Per-component kernels, second set: the batched operators
// behind the plugin's cEnergy / cMZcr / cAcf / cPitchACF / cDeltaRegression / cContourSmoother
// overrides. One 256-thread workgroup per frame (or per row); they share their device
// functions with the fused IS09 kernel (lld_blocks.hpp), so plugin and fused results agree.
#include <hip/hip_runtime.h>

#include "lld_blocks.hpp"
#include "lld_blocks_compare.hpp"
#include "lld_device.hpp"
#include "lld_stage.hpp"

namespace smilehip {

// R12 cEnergy::processVector (energy.cpp:152-168): d = sum of float squares, accumulated in double
__global__ void __launch_bounds__(256) k_sumsq(const float *src, int64_t lds, int64_t N, double *out) {
  __shared__ double scr[4];
  const float *x = src + (int64_t)blockIdx.x * lds;
  double d = 0.0;
  for (int64_t n = threadIdx.x; n < N; n += blockDim.x) { const float t = x[n]; d += t * t; }
  d = block_sum(d, scr);
  if (threadIdx.x == 0) out[blockIdx.x] = d;
}

// R12 cMZcr::processVector, zero crossings (mzcr.cpp:117-124): the count; the caller divides by N
__global__ void __launch_bounds__(256) k_zcr_count(const float *src, int64_t lds, int64_t N, int32_t *out) {
  __shared__ int iscr[4];
  const float *x = src + (int64_t)blockIdx.x * lds;
  int cnt = 0;
  for (int64_t i = 1 + threadIdx.x; i < N - 1; i += blockDim.x)
    if (((x[i - 1] * x[i + 1] <= 0.0f) && (x[i] == 0.0f)) || (x[i - 1] * x[i] < 0.0f)) ++cnt;
  cnt = block_sum_i(cnt, iscr);
  if (threadIdx.x == 0) out[blockIdx.x] = cnt;
}

// R9 cAcf::processVector, forward path (acf.cpp:249-349). LDS: sp[K+3] | re[M] | im[M] | res[M]
__global__ void __launch_bounds__(256) k_acf(const float *src, int64_t lds, float *dst, int64_t ldd, int K, int n_out,
                                             int use_power, int cepstrum, int norm_output, int abs_cepstrum,
                                             const float2 *tw_half, const float2 *tw_full, const OouraTab OO) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int M = K - 1;
  float *sp = smem;
  float *re = sp + ((K + 3) & ~3);
  float *im = re + M;
  float *res = im + M;
  int logM = 0;
  while ((1 << logM) < M) ++logM;
  const float *m = src + (int64_t)blockIdx.x * lds;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float p = m[k];
    if (use_power) p = p * p;                                                              // :252-259
    if (cepstrum == 2) { if (k != 0 && k != K - 1) p = (p > 0.0f) ? (float)log((double)p) : 0.0f; }   // oldCompatCepstrum, :275-286: DC and Nyquist as they are
    else if (cepstrum) p = (p > 0.0f) ? (float)log((double)p + 1.0) : 0.0f;                     // :288-305
    sp[k] = p;
  }
  __syncthreads();
  if (OO.tw) oo_irfft_even<BlockG>(sp, reinterpret_cast<float2 *>(re), OO, res, norm_output ? (float)K : 1.0f, cepstrum ? abs_cepstrum != 0 : true);
  else irfft_even(sp, re, im, M, logM, tw_half, tw_full, res, norm_output ? (float)K : 1.0f, cepstrum ? abs_cepstrum != 0 : true);
  for (int k = threadIdx.x; k < n_out; k += blockDim.x) dst[(int64_t)blockIdx.x * ldd + k] = res[k];
}

// the same operator for K = 257 / 513: one wave per frame on the register form of the inverse network, four frames per workgroup.
// LDS per wave: sp[Kpad] | z[M pairs]
__global__ void __launch_bounds__(256) k_acf_oo_wave(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int K,
                                                     int n_out, int use_power, int cepstrum, int norm_output, int abs_cepstrum,
                                                     const OouraTab OO) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t f = (int64_t)blockIdx.x * 4 + wave;
  if (f >= nF) return;
  const int M = K - 1, Kpad = (K + 3) & ~3;
  float *sp = smem + (size_t)wave * (Kpad + 2 * M);
  float2 *z = reinterpret_cast<float2 *>(sp + Kpad);
  const float *m = src + f * lds;
  for (int k = lane; k < K; k += 64) {
    float p = m[k];
    if (use_power) p = p * p;                                                              // :252-259
    if (cepstrum == 2) { if (k != 0 && k != K - 1) p = (p > 0.0f) ? (float)log((double)p) : 0.0f; }   // oldCompatCepstrum, :275-286: DC and Nyquist as they are
    else if (cepstrum) p = (p > 0.0f) ? (float)log((double)p + 1.0) : 0.0f;                     // :288-305
    sp[k] = p;
  }
  oo_wave_sync();
  oo_wave_inverse(z, OO, lane, [&](int e) { return e == 0 ? make_float2(sp[0], sp[M]) : make_float2(sp[e], 0.0f); });
  const float inv_norm = norm_output ? (float)K : 1.0f;
  const bool take_abs = cepstrum ? abs_cepstrum != 0 : true;
  for (int i = lane; i < n_out; i += 64) {
    const float v = oo_wave_inverse_out(z, OO, i) / inv_norm;
    dst[f * ldd + i] = take_abs ? fabsf(v) : v;
  }
}

// R10 cPitchACF::processVector, per-frame analysis (pitchACF.cpp:137-192): src = [acf(n) | cepstrum(n)]
__global__ void __launch_bounds__(256) k_pitchacf(const float *src, int64_t lds, int n, double fs_sec, double max_pitch,
                                                  double *voicing, int32_t *max_idx) {
  __shared__ double scr[4];
  __shared__ int iscr[4];
  const float *a = src + (int64_t)blockIdx.x * lds;
  double v, Tsamp;
  int mi;
  pitchacf_frame(a, a + n, n, fs_sec, max_pitch, scr, iscr, v, mi, Tsamp);
  if (threadIdx.x == 0) { voicing[blockIdx.x] = v; max_idx[blockIdx.x] = mi; }
}

// R13 cDeltaRegression / cContourSmoother::processBuffer on one row: x points at index 0 and is
// valid on [-W, nT+W) (deltaRegression.cpp:144-152, contourSmoother.cpp:106-114)
__global__ void k_window_op(const float *x, float *y, int64_t nT, int kind, int W, float norm) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nT) return;
  if (kind == 0) {
    float num = 0.0f;
    for (int i = 1; i <= W; ++i) num += (float)i * (x[n + i] - x[n - i]);
    y[n] = num / norm;
  } else {
    float v = x[n];
    for (int w = 1; w <= W; ++w) { v += x[n - w]; v += x[n + w]; }
    y[n] = v / (float)(2 * W + 1);
  }
}

// The option variants of the two window processors, one row: kind 2 = cContourSmoother with noZeroSma (contourSmoother.cpp:91-104:
// a zero stays zero, the mean runs over the non-zero neighbours); kind 3 = cDeltaRegression with onlyInSegments
// (deltaRegression.cpp:121-137): pairs with a zero / NaN member are skipped -- and the reference ADDS i^2 to its norm member for
// every pair it uses, frame after frame, field after field (the member is never reset): the divisor is state carried through
// every row the instance ever processes. One thread walks the row in order; *norm_io is that member, in and out.
__global__ void k_window_op_seq(const float *x, float *y, int64_t nT, int kind, int W, float *norm_io) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (kind == 2) {
    for (int64_t n = 0; n < nT; ++n) {
      if (x[n] != 0.0f) {
        long N = 1;
        float v = x[n];
        for (int w = 1; w <= W; ++w) {
          if (x[n - w] != 0.0f) { v += x[n - w]; N++; }
          if (x[n + w] != 0.0f) { v += x[n + w]; N++; }
        }
        y[n] = v / (float)N;
      } else {
        y[n] = 0.0f;
      }
    }
    return;
  }
  float norm = *norm_io;
  for (int64_t n = 0; n < nT; ++n) {
    float num = 0.0f;
    for (int i = 1; i <= W; ++i) {
      const float a = x[n + i], b = x[n - i];
      if (!(a == 0.0f || a != a || b == 0.0f || b != b)) {
        num += (float)i * (a - b);
        norm += (float)i * (float)i;
      }
    }
    y[n] = (norm != 0.0f) ? num / norm : 0.0f;
  }
  *norm_io = norm;
}

// cDeltaRegression::processBuffer with its option variants (deltaRegression.cpp:104-170), one row: flags 1 = relativeDelta
// (computeDelta :104-111: delta / |prior|, 0 where prior is 0), 2 = halfWaveRect, 4 = absOutput (:158-166; halfWaveRect wins),
// 8 = onlyInSegments (pairs with a zero / NaN member skipped; the norm member grows with every pair used, see k_window_op_seq).
// W = 0: the simple difference x[n] - x[n-1] (:141-153). Without onlyInSegments every output is independent.
__device__ __forceinline__ float delta_of(float prior, float later, int relative) {
  float delta = later - prior;
  if (relative) delta = (prior != 0.0f) ? delta / fabsf(prior) : 0.0f;
  return delta;
}
__device__ __forceinline__ float delta_post(float y, int flags) {
  if (flags & 2) return y < 0.0f ? 0.0f : y;
  if (flags & 4) return y < 0.0f ? -y : y;
  return y;
}
__device__ __forceinline__ bool no_value(float v) { return v == 0.0f || v != v; }    // isNoValue, deltaRegression.hpp
__global__ void k_delta_op(const float *x, float *y, int64_t nT, int W, float norm, int flags) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nT) return;
  float v;
  if (W > 0) {
    float num = 0.0f;
    for (int i = 1; i <= W; ++i) num += (float)i * delta_of(x[n - i], x[n + i], flags & 1);
    v = num / norm;
  } else {
    v = delta_of(x[n - 1], x[n], flags & 1);
  }
  y[n] = delta_post(v, flags);
}
__global__ void k_delta_op_seq(const float *x, float *y, int64_t nT, int W, float *norm_io, int flags) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (W > 0) {
    float norm = *norm_io;
    for (int64_t n = 0; n < nT; ++n) {
      float num = 0.0f;
      for (int i = 1; i <= W; ++i) {
        const float a = x[n + i], b = x[n - i];
        if (!(no_value(a) || no_value(b))) {
          num += (float)i * delta_of(b, a, flags & 1);
          norm += (float)i * (float)i;
        }
      }
      y[n] = delta_post((norm != 0.0f) ? num / norm : 0.0f, flags);
    }
    *norm_io = norm;
  } else {
    for (int64_t n = 0; n < nT; ++n)
      y[n] = delta_post((no_value(x[n]) || no_value(x[n - 1])) ? 0.0f : delta_of(x[n - 1], x[n], flags & 1), flags);
  }
}

// The same three window processors on a whole block, frame-major (x[(n + i) * ldx + c]): one thread per (frame, element). Identical
// expressions to k_window_op / k_window_op_seq (kind 2) / k_delta_op, value for value.
__global__ void k_window_op_block(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t nT, int nC, int op, int W, float norm,
                                  int flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nT * nC) return;
  const int64_t n = i / nC;
  const int c = (int)(i - n * nC);
  const float *xc = x + c;
  float out;
  if (op == 0) {
    float v;
    if (W > 0) {
      float num = 0.0f;
      for (int k = 1; k <= W; ++k) num += (float)k * delta_of(xc[(n - k) * ldx], xc[(n + k) * ldx], flags & 1);
      v = num / norm;
    } else {
      v = delta_of(xc[(n - 1) * ldx], xc[n * ldx], flags & 1);
    }
    out = delta_post(v, flags);
  } else if (op == 1) {
    float v = xc[n * ldx];
    for (int w = 1; w <= W; ++w) { v += xc[(n - w) * ldx]; v += xc[(n + w) * ldx]; }
    out = v / (float)(2 * W + 1);
  } else {
    const float x0 = xc[n * ldx];
    if (x0 != 0.0f) {
      long N = 1;
      float v = x0;
      for (int w = 1; w <= W; ++w) {
        const float a = xc[(n - w) * ldx], b = xc[(n + w) * ldx];
        if (a != 0.0f) { v += a; N++; }
        if (b != 0.0f) { v += b; N++; }
      }
      out = v / (float)N;
    } else {
      out = 0.0f;
    }
  }
  y[n * ldy + c] = out;
}

// cDeltaRegression with onlyInSegments on a block of n_ticks * bs frames: the divisor is carried from value to value in the order the
// reference's ticks visit them -- tick after tick (bs frames each), within a tick element after element, within an element the
// tick's frames (windowProcessor.cpp:188-200) -- so ONE thread walks the block in exactly that order (k_delta_op_seq's expressions).
__global__ void k_delta_seg_block(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t n_ticks, int bs, int nC, int W,
                                  float *norm_io, int flags) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float norm = *norm_io;
  for (int64_t j = 0; j < n_ticks; ++j)
    for (int c = 0; c < nC; ++c)
      for (int t = 0; t < bs; ++t) {
        const int64_t n = j * bs + t;
        const float *xc = x + c;
        float out;
        if (W > 0) {
          float num = 0.0f;
          for (int i = 1; i <= W; ++i) {
            const float a = xc[(n + i) * ldx], b = xc[(n - i) * ldx];
            if (!(no_value(a) || no_value(b))) {
              num += (float)i * delta_of(b, a, flags & 1);
              norm += (float)i * (float)i;
            }
          }
          out = delta_post((norm != 0.0f) ? num / norm : 0.0f, flags);
        } else {
          const float a = xc[n * ldx], b = xc[(n - 1) * ldx];
          out = delta_post((no_value(a) || no_value(b)) ? 0.0f : delta_of(b, a, flags & 1), flags);
        }
        y[n * ldy + c] = out;
      }
  *norm_io = norm;
}

// R11 cSpectral::processVector, ComParE option set: the frames of one stream in order (the flux needs the
// previous frame's magnitudes; `state` carries them across calls). One workgroup, K = 257.
__global__ void __launch_bounds__(256) k_spectral(const float *src, int64_t lds, float *state, int first, float *dst,
                                                  int64_t ldd, int64_t nF, int K, SpectralConsts C) {
  __shared__ __attribute__((aligned(16))) float mg[260], pw[260], prev[260];
  __shared__ double red[64], cum[256];
  __shared__ float pk_val[4];
  __shared__ int pk_has[4];
  for (int k = threadIdx.x; k < K; k += blockDim.x) prev[k] = first ? 0.0f : state[k];
  __syncthreads();
  for (int64_t f = 0; f < nF; ++f) {
    const float *m = src + f * lds;
    for (int k = threadIdx.x; k < K; k += blockDim.x) { const float v = m[k]; mg[k] = v; pw[k] = v * v; }   // squareInput (:676-683)
    __syncthreads();
    spectral_frame(mg, pw, prev, first && f == 0, C, K, red, cum, pk_val, pk_has, dst + f * ldd);
    for (int k = threadIdx.x; k < K; k += blockDim.x) prev[k] = mg[k];
    __syncthreads();
  }
  for (int k = threadIdx.x; k < K; k += blockDim.x) state[k] = prev[k];
}

// The same operator for the other spectrum sizes of 20 ms frames (K = 129: FFT 256 at 8 / 11.025 kHz, K = 513: FFT 1024 at 32 .. 48 kHz):
// one wave, the wave form of the descriptors (spectral_frame_wave<W>: the fused ComParE kernel's own code, bit-identical to the block form).
template <int W>
__global__ void __launch_bounds__(64) k_spectral_w(const float *src, int64_t lds, float *state, int first, float *dst, int64_t ldd,
                                                  int64_t nF, SpectralConsts C) {
  constexpr int K = 64 * W + 1, KP = K + 3;
  __shared__ __attribute__((aligned(16))) float mg[KP], pw[KP], prev[KP], chain[128 * W];
  for (int k = threadIdx.x; k < K; k += 64) prev[k] = first ? 0.0f : state[k];
  WaveG::sync();
  for (int64_t f = 0; f < nF; ++f) {
    const float *m = src + f * lds;
    for (int k = threadIdx.x; k < K; k += 64) { const float v = m[k]; mg[k] = v; pw[k] = v * v; }
    WaveG::sync();
    spectral_frame_wave<W>(mg, pw, prev, first && f == 0, C, K, chain, dst + f * ldd);
    WaveG::sync();
    for (int k = threadIdx.x; k < K; k += 64) prev[k] = mg[k];
    WaveG::sync();
  }
  for (int k = threadIdx.x; k < K; k += 64) state[k] = prev[k];
}

// R8 cPlp::processVector as auditory spectrum (doAud = 1, no IDFT / LP), with or without newRASTA: the frames of
// one stream in order, lane = band. state: 4 filter taps per band + the frame counter (as a float) at [4*nB].
// rasta == 2: the older RASTA form (plp.cpp:447-466): a five-frame ring of the log band values, FIR over it, one IIR value per band;
// state: [6 x nB] (ring[5] | iir) + the frame counter at [6*nB] + the ring position at [6*nB + 1].
__global__ void __launch_bounds__(64) k_plp(const float *src, int64_t lds, int nB, const float *eql, PlpConsts Q, int rasta,
                                            float *state, float *dst, int64_t ldd, int64_t nF) {
  const int b = threadIdx.x;
  if (b >= nB) return;
  const float e = eql[b];
  if (!rasta) {
    for (int64_t f = 0; f < nF; ++f) dst[f * ldd + b] = plp_aud_band(src[f * lds + b], Q.melfloor, e, Q.compression);
    return;
  }
  if (rasta == 2) {
    float *ring = state + 6 * b;                          // (read and written in place: one frame at a time in the plugin's use)
    float iirv = ring[5];
    int init = (int)state[6 * nB], ptr = (int)state[6 * nB + 1];
    __syncthreads();                                     // every band has read the counters before band 0 rewrites them
    for (int64_t f = 0; f < nF; ++f) {
      const float v = src[f * lds + b];
      const float x = glibc_logf(v < Q.melfloor ? Q.melfloor : v);
      ring[ptr] = x;
      float sum = Q.fir[0] * x;
      for (int m = 1; m < 5; m++) sum += Q.fir[m] * ring[(5 - m + ptr) % 5];
      sum += Q.iir * iirv;
      iirv = sum;
      float y = (init >= 5) ? sum : 0.0f;
      y += e;                                            // log equal loudness, compression, exp (:490-497, :512-517)
      y *= Q.compression;
      dst[f * ldd + b] = glibc_expf(y);
      if (init < 5) init++;
      ptr = (ptr + 1) % 5;
    }
    ring[5] = iirv;
    if (b == 0) { state[6 * nB] = (float)init; state[6 * nB + 1] = (float)ptr; }
    return;
  }
  float st[4] = {state[4 * b], state[4 * b + 1], state[4 * b + 2], state[4 * b + 3]};
  int init = (int)state[4 * nB];
  for (int64_t f = 0; f < nF; ++f) {
    const float v = src[f * lds + b];
    const float x = glibc_logf(v < Q.melfloor ? Q.melfloor : v);          // doLog, plp.cpp:434-439
    dst[f * ldd + b] = plp_rasta_band(x, st, init, Q.fir, Q.iir, e, Q.compression);
    if (init < 5) init++;
  }
  for (int i = 0; i < 4; ++i) state[4 * b + i] = st[i];
  if (b == 0) state[4 * nB] = (float)init;
}

// R8 cPlp::processVector with doAud = doIDFT = doLP = doLpToCeps = 1, htkcompatible = 1 (PLP cepstra, plp.cpp:499-583):
// one 64-thread workgroup per frame. LDS: aud[n_bands] | acf[16]
// out_stage: 3 = cepstra (order + 1 values), 2 = LP coefficients (order), 1 = the autocorrelation (order + 1)
__global__ void __launch_bounds__(64) k_plp_cc(const float *src, int64_t lds, int nB, const float *eql, float melfloor,
                                               float compression, int order, const float *costab, const float *sintab,
                                               float *dst, int64_t ldd, int out_stage) {
  __shared__ float aud[64], acf[16];
  const float *m = src + (int64_t)blockIdx.x * lds;
  const int b = threadIdx.x;
  if (b < nB) {
    float v = m[b];
    if (v < melfloor) v = melfloor;
    v *= eql[b];
    aud[b] = (float)pow((double)v, (double)compression);
  }
  __syncthreads();
  if (b <= order) acf[b] = plp_acf_lag(aud, costab + b * (nB + 2), nB);
  __syncthreads();
  if (out_stage == 1) {
    if (b <= order) dst[(int64_t)blockIdx.x * ldd + b] = acf[b];
    return;
  }
  if (b == 0) {
    float o[16];
    plp_cc_serial(acf, order, sintab, o, out_stage == 2);
    const int n = out_stage == 2 ? order : order + 1;
    for (int r = 0; r < n; ++r) dst[(int64_t)blockIdx.x * ldd + r] = o[r];
  }
}

static inline unsigned nblk2(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

// cValbasedSelector::myTick (valbasedSelector.cpp:139-247) for a block of frames: element idx against the threshold decides
// whether a frame is handed on (keep = 1), replaced by a constant vector (zeroVec: keep = 2) or dropped (keep = 0); removeIdx
// takes the tested element out of the output vector. One thread per output element.
__global__ void __launch_bounds__(256) k_valbased(const float *src, int64_t lds, int N, int idx, float threshold, int invert,
                                                  int allow_equal, int zerovec, int remove_idx, float output_val, float *dst,
                                                  int64_t ldd, int32_t *keep, int64_t nF) {
  const int64_t f = blockIdx.x;
  if (f >= nF) return;
  const float *x = src + f * lds;
  const int i = idx >= N ? N - 1 : idx;
  const float val = x[i];
  const bool copy = ((!invert) && (val > threshold)) || (invert && (val < threshold)) || (allow_equal && (val == threshold));
  const int n_out = remove_idx ? N - 1 : N;
  for (int j = threadIdx.x; j < n_out; j += blockDim.x) {
    const int sj = (remove_idx && j >= i) ? j + 1 : j;
    dst[f * ldd + j] = copy ? x[sj] : output_val;
  }
  if (threadIdx.x == 0) keep[f] = copy ? 1 : (zerovec ? 2 : 0);
}

hipError_t stage_valbased(const float *src, int64_t lds, int N, int64_t nF, int idx, float threshold, int invert, int allow_equal,
                          int zerovec, int remove_idx, float output_val, float *dst, int64_t ldd, int32_t *keep, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_valbased, dim3((unsigned)nF), dim3(N > 128 ? 256 : 64), 0, s, src, lds, N, idx, threshold, invert, allow_equal, zerovec,
                     remove_idx, output_val, dst, ldd, keep, nF);
  return hipGetLastError();
}

hipError_t stage_sumsq(const float *src, int64_t lds, int64_t N, int64_t nF, double *out, hipStream_t s) {
  if (nF > 0) hipLaunchKernelGGL(k_sumsq, dim3((unsigned)nF), dim3(256), 0, s, src, lds, N, out);
  return hipGetLastError();
}
hipError_t stage_zcr_count(const float *src, int64_t lds, int64_t N, int64_t nF, int32_t *out, hipStream_t s) {
  if (nF > 0) hipLaunchKernelGGL(k_zcr_count, dim3((unsigned)nF), dim3(256), 0, s, src, lds, N, out);
  return hipGetLastError();
}
hipError_t stage_acf(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t nF, int K, int n_out, int use_power,
                     int cepstrum, int norm_output, int abs_cepstrum, const float2 *tw_half, const float2 *tw_full,
                     const OouraTab &OO, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  const int M = K - 1;
  if (OO.tw && (M == 256 || M == 512)) {
    const size_t lb = 4 * sizeof(float) * (size_t)(((K + 3) & ~3) + 2 * M);
    hipLaunchKernelGGL(k_acf_oo_wave, dim3((unsigned)((nF + 3) / 4)), dim3(256), lb, s, src, lds, dst, ldd, nF, K, n_out, use_power,
                       cepstrum, norm_output, abs_cepstrum, OO);
    return hipGetLastError();
  }
  const size_t lds_bytes = sizeof(float) * (size_t)(((K + 3) & ~3) + 3 * M);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_acf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_acf, dim3((unsigned)nF), dim3(256), lds_bytes, s, src, lds, dst, ldd, K, n_out, use_power, cepstrum,
                     norm_output, abs_cepstrum, tw_half, tw_full, OO);
  return hipGetLastError();
}
hipError_t stage_pitchacf(const float *src, int64_t lds, int64_t nF, int n, double fs_sec, double max_pitch, double *voicing,
                          int32_t *max_idx, hipStream_t s) {
  if (nF > 0) hipLaunchKernelGGL(k_pitchacf, dim3((unsigned)nF), dim3(256), 0, s, src, lds, n, fs_sec, max_pitch, voicing, max_idx);
  return hipGetLastError();
}
hipError_t stage_spectral(const float *src, int64_t lds, float *state, bool first, float *dst, int64_t ldd, int64_t nF, int K,
                          const SpectralConsts &C, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  if (K == 257) hipLaunchKernelGGL(k_spectral, dim3(1), dim3(256), 0, s, src, lds, state, first ? 1 : 0, dst, ldd, nF, K, C);
  else if (K == 129) hipLaunchKernelGGL(k_spectral_w<2>, dim3(1), dim3(64), 0, s, src, lds, state, first ? 1 : 0, dst, ldd, nF, C);
  else if (K == 513) hipLaunchKernelGGL(k_spectral_w<8>, dim3(1), dim3(64), 0, s, src, lds, state, first ? 1 : 0, dst, ldd, nF, C);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t stage_plp(const float *src, int64_t lds, int n_bands, const float *eql, const PlpConsts &Q, int rasta, float *state,
                     float *dst, int64_t ldd, int64_t nF, hipStream_t s) {
  if (nF > 0) hipLaunchKernelGGL(k_plp, dim3(1), dim3(64), 0, s, src, lds, n_bands, eql, Q, rasta, state, dst, ldd, nF);
  return hipGetLastError();
}
hipError_t stage_plp_cc(const float *src, int64_t lds, int n_bands, const float *eql, float melfloor, float compression,
                        int order, const float *costab, const float *sintab, float *dst, int64_t ldd, int64_t nF, hipStream_t s, int out_stage) {
  if (nF > 0)
    hipLaunchKernelGGL(k_plp_cc, dim3((unsigned)nF), dim3(64), 0, s, src, lds, n_bands, eql, melfloor, compression, order, costab,
                       sintab, dst, ldd, out_stage);
  return hipGetLastError();
}
hipError_t stage_window_op_seq(const float *x, float *y, int64_t nT, int kind, int W, float *d_norm, hipStream_t s) {
  if (nT > 0) hipLaunchKernelGGL(k_window_op_seq, dim3(1), dim3(64), 0, s, x, y, nT, kind, W, d_norm);
  return hipGetLastError();
}
hipError_t stage_delta_op(const float *x, float *y, int64_t nT, int W, float norm, int flags, float *d_norm_io, hipStream_t s) {
  if (nT <= 0) return hipSuccess;
  if (flags & 8) hipLaunchKernelGGL(k_delta_op_seq, dim3(1), dim3(64), 0, s, x, y, nT, W, d_norm_io, flags);
  else hipLaunchKernelGGL(k_delta_op, dim3(nblk2(nT, 256)), dim3(256), 0, s, x, y, nT, W, norm, flags);
  return hipGetLastError();
}
hipError_t stage_delta_seg_block(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t n_ticks, int bs, int nC, int W, float *d_norm,
                                 int flags, hipStream_t s) {
  if (n_ticks > 0 && nC > 0) hipLaunchKernelGGL(k_delta_seg_block, dim3(1), dim3(64), 0, s, x, ldx, y, ldy, n_ticks, bs, nC, W, d_norm, flags);
  return hipGetLastError();
}
hipError_t stage_window_op_block(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t nT, int nC, int op, int W, float norm,
                                 int flags, hipStream_t s) {
  if (nT > 0 && nC > 0)
    hipLaunchKernelGGL(k_window_op_block, dim3(nblk2(nT * nC, 256)), dim3(256), 0, s, x, ldx, y, ldy, nT, nC, op, W, norm, flags);
  return hipGetLastError();
}
hipError_t stage_window_op(const float *x, float *y, int64_t nT, int kind, int W, float norm, hipStream_t s) {
  if (nT > 0) hipLaunchKernelGGL(k_window_op, dim3(nblk2(nT, 256)), dim3(256), 0, s, x, y, nT, kind, W, norm);
  return hipGetLastError();
}

}  // namespace smilehip
