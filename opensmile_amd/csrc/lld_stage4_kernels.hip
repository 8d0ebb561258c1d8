// Per-component operators for the components the other INTERSPEECH sets of config/is09-13 (IS10_paraling, IS11_speaker_state,
// IS12_speaker_trait) add to the five BASELINE configs' graphs: cIntensity, cLsp, cPitchSmoother, cVectorOperation, and cSpecResample
// / cLpc for ANY geometry (lld_gemaps.hip has the two fused for eGeMAPS' 512 -> 220 samples, p = 11). The per-frame recipes are
// sequential float / double chains whose order is the result (lld_is10_ops.hpp, checked on the host against the real binary), so
// the kernels here are layouts around them: one thread per frame (cLsp: ~3 k dependent float operations per frame, no parallel
// form in the reference's rounding), one thread per stream (cPitchSmoother: a state machine over the frames), one lane per lag
// (cLpc's autocorrelation: a float accumulation in sample order per lag), one thread per output sample (cSpecResample's direct
// inverse DFT: a float accumulation over the bins per sample).
#include <hip/hip_runtime.h>

#include "lld_is10_ops.hpp"
#include "lld_stage.hpp"

namespace smilehip {

namespace {
inline unsigned nblk4(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }
}  // namespace

// cIntensity::processVector (intensity.cpp:125-145): w0, w1 = the first two values of the Hamming window (all the reference's
// clamped summation ever reads: n_sum = number of outputs <= 2), win_sum = the whole window's sum.
__global__ void __launch_bounds__(256) k_intensity(const float *src, int64_t lds, int n_sum, double w0, double w1, double win_sum, int flags,
                                                   float *dst, int64_t ldd, int64_t nF) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nF) return;
  const double win[2] = {w0, w1};
  float x[2] = {0.0f, 0.0f}, o[2];
  for (int i = 0; i < n_sum; ++i) x[i] = src[f * lds + i];
  const int n = is10::intensity_frame(x, n_sum, win, win_sum, flags, o);
  for (int i = 0; i < n; ++i) dst[f * ldd + i] = o[i];
}

// cLsp::processVector (lsp.cpp:289-312), one thread per frame
__global__ void __launch_bounds__(64) k_lsp(const float *lpc, int64_t lds, int p, float *dst, int64_t ldd, int64_t nF) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nF) return;
  float a[is10::kLspMaxOrder], o[is10::kLspMaxOrder];
  for (int i = 0; i < p; ++i) a[i] = lpc[f * lds + i];
  is10::lsp_frame(a, p, o);
  for (int i = 0; i < p; ++i) dst[f * ldd + i] = o[i];
}

// cVectorOperation::processVector, element-wise operations (vectorOperation.cpp:360-435, 508-527)
__global__ void __launch_bounds__(256) k_vecop(int op, float aux, float logfloor, const float *src, int64_t lds, int n_cols, float *dst,
                                               int64_t ldd, int64_t nF) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nF * n_cols) return;
  const int64_t f = i / n_cols;
  const int c = (int)(i - f * n_cols);
  dst[f * ldd + c] = is10::vecop(op, aux, logfloor, src[f * lds + c]);
}

// the vector-to-scalar operations (sum, ssm, ll1, ll2; vectorOperation.cpp:461-490): one float accumulation per frame, thread per frame
__global__ void __launch_bounds__(256) k_vecop_reduce(int op, const float *src, int64_t lds, int n_cols, float *dst, int64_t ldd, int64_t nF) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nF) return;
  dst[f * ldd] = is10::vecop_reduce(op, src + f * lds, n_cols);
}

// cPitchSmoother::processVector (pitchSmoother.cpp:236-425) over the frames of a stream, one thread per stream. Stream u reads
// rows [row_off[u], row_off[u+1]) of src (row_off == nullptr: ONE stream of n_single rows) and writes the rows the component
// writes one after the other from row row_off[u] of dst on: rows - 1 of them with simple post smoothing (the component delays by
// one frame and emits nothing for the first), rows otherwise; written[u] = that count. state (optional): the stream's carried state,
// read when resume != 0, always written back -- the plugin pushes one frame per call.
__global__ void __launch_bounds__(64) k_pitch_smoother(is10::PitchSmootherOpts o, const float *src, int64_t lds, const int64_t *row_off,
                                                       int n_streams, int64_t n_single, is10::PitchSmootherState *state, int resume,
                                                       float *dst, int64_t ldd, int64_t *written) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_streams) return;
  const int64_t r0 = row_off ? row_off[u] : 0, r1 = row_off ? row_off[u + 1] : n_single;
  is10::PitchSmootherState s;
  if (state && resume) s = state[u];
  else is10::pitch_smoother_reset(s);
  int64_t w = 0;
  for (int64_t r = r0; r < r1; ++r) {
    float o4[4];
    const int n = is10::pitch_smoother_frame(o, s, src + r * lds, 1, o4);
    if (n > 0) {
      for (int i = 0; i < n; ++i) dst[(r0 + w) * ldd + i] = o4[i];
      ++w;
    }
  }
  if (state) state[u] = s;
  if (written) written[u] = w;
}

// cSpecResample::processVector -> smileDsp_irdft (smileUtil.c:1800-1820): out[i] = (in[0] (+ in[1] cos[K/2] if I >= K) + sum over the
// bin pairs of in[k] cos + in[k+1] sin) / (K/2) as ONE float accumulation in index order per output sample. The tables hold
// kMax/2 entries per output sample from index -1 on (smileDsp_initIrdft, :1752-1786). Block = one frame (its spectrum in LDS),
// thread = output samples i, i + 256, ...
__global__ void __launch_bounds__(256) k_specresample_g(const float *src, int64_t lds, int K, int I, int kMax, const float *cost,
                                                        const float *sint, float *dst, int64_t ldd) {
  extern __shared__ __attribute__((aligned(16))) float s_in4[];
  const float *in = src + (int64_t)blockIdx.x * lds;
  for (int k = threadIdx.x; k < K; k += blockDim.x) s_in4[k] = in[k];
  __syncthreads();
  const int h = kMax / 2;
  const float div = (float)(K / 2);
  for (int i = threadIdx.x; i < I; i += blockDim.x) {
    const float *c = cost + (int64_t)i * h - 1, *s = sint + (int64_t)i * h - 1;
    float acc = s_in4[0];
    if (I >= K) acc += s_in4[1] * c[K / 2];
    int k = 2;
    for (; k + 8 <= kMax; k += 8) {                    // table values four pairs ahead of the dependent additions
      float cv[4], sv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { cv[q] = c[(k >> 1) + q]; sv[q] = s[(k >> 1) + q]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc += s_in4[k + 2 * q] * cv[q];
        acc += s_in4[k + 2 * q + 1] * sv[q];
      }
    }
    for (; k < kMax; k += 2) {
      acc += s_in4[k] * c[k >> 1];
      acc += s_in4[k + 1] * s[k >> 1];
    }
    dst[(int64_t)blockIdx.x * ldd + i] = acc / div;
  }
}

// The same for FOUR frames per block: a table value is loaded once and used for four accumulations (one per frame, each its own
// chain in the reference's order); the four spectra sit interleaved in LDS so that one ds_read_b128 broadcasts bin k of all four.
// Batches; the single-frame form above serves one frame at a time (the plugin) and spectra too long for the LDS budget.
__global__ void __launch_bounds__(1024) k_specresample_g4(const float *src, int64_t lds, int K, int I, int kMax, const float *cost,
                                                         const float *sint, float *dst, int64_t ldd, int64_t nF) {
  extern __shared__ __attribute__((aligned(16))) float4 s_in4v[];
  const int64_t f0 = (int64_t)blockIdx.x * 4;
  const int nf = (int)((nF - f0) < 4 ? (nF - f0) : 4);
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    v.x = src[f0 * lds + k];
    if (nf > 1) v.y = src[(f0 + 1) * lds + k];
    if (nf > 2) v.z = src[(f0 + 2) * lds + k];
    if (nf > 3) v.w = src[(f0 + 3) * lds + k];
    s_in4v[k] = v;
  }
  __syncthreads();
  const int h = kMax / 2;
  const float div = (float)(K / 2);
  for (int i = threadIdx.x; i < I; i += blockDim.x) {
    const float *c = cost + (int64_t)i * h - 1, *s = sint + (int64_t)i * h - 1;
    float4 acc = s_in4v[0];
    if (I >= K) {
      const float4 b = s_in4v[1];
      const float cn = c[K / 2];
      acc.x += b.x * cn; acc.y += b.y * cn; acc.z += b.z * cn; acc.w += b.w * cn;
    }
    for (int k = 2; k < kMax; k += 2) {
      const float cv = c[k >> 1], sv = s[k >> 1];
      const float4 a = s_in4v[k], b = s_in4v[k + 1];
      acc.x += a.x * cv; acc.x += b.x * sv;
      acc.y += a.y * cv; acc.y += b.y * sv;
      acc.z += a.z * cv; acc.z += b.z * sv;
      acc.w += a.w * cv; acc.w += b.w * sv;
    }
    dst[f0 * ldd + i] = acc.x / div;
    if (nf > 1) dst[(f0 + 1) * ldd + i] = acc.y / div;
    if (nf > 2) dst[(f0 + 2) * ldd + i] = acc.z / div;
    if (nf > 3) dst[(f0 + 3) * ldd + i] = acc.w / div;
  }
}

// cLpc::processVector with method = acf, saveLPCoeff only (lpc.cpp:171-213): smileDsp_autoCorr (smileUtil.c:1560-1570: r[lag] = sum
// over i = lag .. n-1 of x[i] x[i-lag], a float accumulation in sample order) on lanes 0 .. p, then Durbin's recursion
// (smileDsp_calcLpcAcf, :1572-1630) on lane 0. One wave per frame, p <= 32.
__global__ void __launch_bounds__(64) k_lpc_g(const float *x, int64_t lds, int n, int p, float *dst, int64_t ldd) {
  extern __shared__ __attribute__((aligned(16))) float s_x4[];
  __shared__ float s_r[64];
  const int lane = threadIdx.x;
  const float *xi = x + (int64_t)blockIdx.x * lds;
  for (int i = lane; i < n; i += 64) s_x4[i] = xi[i];
  __syncthreads();
  if (lane <= p) {
    float r = 0.0f;
    for (int i = lane; i < n; ++i) r += s_x4[i] * s_x4[i - lane];
    s_r[lane] = r;
  }
  __syncthreads();
  if (lane != 0) return;
  float lpc[is10::kLspMaxOrder];
  float *o = dst + (int64_t)blockIdx.x * ldd;
  if (s_r[0] == 0.0f) {
    for (int i = 0; i < p; ++i) o[i] = 0.0f;
    return;
  }
  for (int i = 0; i < p; ++i) lpc[i] = 0.0f;
  float e = s_r[0];
  for (int m = 1; m <= p; ++m) {
    float sum = 1.0f * s_r[m];
    for (int i = 1; i < m; ++i) sum += lpc[i - 1] * s_r[m - i];
    const float k_m = (-1.0f / e) * sum;
    lpc[m - 1] = k_m;
    for (int i = 1; i <= m / 2; ++i) {
      const float xx = lpc[i - 1];
      lpc[i - 1] += k_m * lpc[m - i - 1];
      if ((i < (m / 2)) || ((m & 1) == 1)) lpc[m - i - 1] += k_m * xx;
    }
    e *= (1.0f - k_m * k_m);
    if (e == 0.0f) {
      for (int i = m; i < p; ++i) lpc[i] = 0.0f;
      break;
    }
  }
  for (int i = 0; i < p; ++i) o[i] = lpc[i];
}

// ---------------------------------------------------------------------------------------------------------------- launchers
hipError_t stage_intensity(const float *src, int64_t lds, int n_sum, double w0, double w1, double win_sum, int flags, float *dst,
                           int64_t ldd, int64_t nF, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_intensity, dim3(nblk4(nF, 256)), dim3(256), 0, s, src, lds, n_sum, w0, w1, win_sum, flags, dst, ldd, nF);
  return hipGetLastError();
}

hipError_t stage_lsp(const float *lpc, int64_t lds, int p, float *dst, int64_t ldd, int64_t nF, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  if (p < 2 || p > is10::kLspMaxOrder) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_lsp, dim3(nblk4(nF, 64)), dim3(64), 0, s, lpc, lds, p, dst, ldd, nF);
  return hipGetLastError();
}

hipError_t stage_vecop(int op, float aux, float logfloor, const float *src, int64_t lds, int n_cols, float *dst, int64_t ldd, int64_t nF,
                       hipStream_t s) {
  if (nF <= 0 || n_cols <= 0) return hipSuccess;
  if (op >= is10::kVopXSum && op < is10::kVopXCount) {
    hipLaunchKernelGGL(k_vecop_reduce, dim3(nblk4(nF, 256)), dim3(256), 0, s, op, src, lds, n_cols, dst, ldd, nF);
    return hipGetLastError();
  }
  if (op < 0 || op >= is10::kVopCount) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_vecop, dim3(nblk4(nF * n_cols, 256)), dim3(256), 0, s, op, aux, logfloor, src, lds, n_cols, dst, ldd, nF);
  return hipGetLastError();
}

hipError_t stage_pitch_smoother(int n_cand, float voicing_cutoff, int octave_correction, int post_simple, int flags, const float *src,
                                int64_t lds, const int64_t *row_off, int n_streams, int64_t n_single, void *state, int resume,
                                float *dst, int64_t ldd, int64_t *written, hipStream_t s) {
  if (n_streams <= 0) return hipSuccess;
  if (n_cand < 1 || n_cand > is10::kSmootherMaxCand || !(flags & 15) || (flags & ~15)) return hipErrorInvalidValue;
  is10::PitchSmootherOpts o{n_cand, octave_correction, post_simple, flags, voicing_cutoff};
  hipLaunchKernelGGL(k_pitch_smoother, dim3(nblk4(n_streams, 64)), dim3(64), 0, s, o, src, lds, row_off, n_streams, n_single,
                     reinterpret_cast<is10::PitchSmootherState *>(state), resume, dst, ldd, written);
  return hipGetLastError();
}

hipError_t stage_specresample_g(const float *src, int64_t lds, int K, int I, int kMax, const float *cost, const float *sint, float *dst,
                                int64_t ldd, int64_t nF, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  const size_t bytes = sizeof(float) * (size_t)((K + 3) & ~3);
  if (bytes > 60 * 1024) return hipErrorInvalidValue;
  if (nF >= 4 && bytes * 4 <= 48 * 1024) {
    const int threads = I >= 1024 ? 1024 : ((I + 63) & ~63);        // one pass over the output samples where a block can hold them (275 -> 320 threads)
    hipLaunchKernelGGL(k_specresample_g4, dim3((unsigned)((nF + 3) / 4)), dim3(threads), bytes * 4, s, src, lds, K, I, kMax, cost, sint, dst, ldd, nF);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_specresample_g, dim3((unsigned)nF), dim3(256), bytes, s, src, lds, K, I, kMax, cost, sint, dst, ldd);
  return hipGetLastError();
}

hipError_t stage_lpc_g(const float *x, int64_t lds, int n, int p, float *dst, int64_t ldd, int64_t nF, hipStream_t s) {
  if (nF <= 0) return hipSuccess;
  if (p < 1 || p > is10::kLspMaxOrder || n <= p) return hipErrorInvalidValue;
  const size_t bytes = sizeof(float) * (size_t)((n + 3) & ~3);
  if (bytes > 60 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_lpc_g, dim3((unsigned)nF), dim3(64), bytes, s, x, lds, n, p, dst, ldd);
  return hipGetLastError();
}

}  // namespace smilehip
