// HIP kernels for the openSMILE LLD hot path on gfx950 (CDNA4).
//
// Compiled with -ffp-contract=off: every a*b+c below rounds twice exactly like
// the reference's x86-64 build (no FMA without -march=native); FMA is used
// only where it is written explicitly (fmaf in the FFT, whose round-off
// cannot match Ooura's split-radix order anyway).
//
// Kernels in this file
//   lld_mfcc_generic   one workgroup per frame, any power-of-two Nfft <= 8192:
//                      the reference-order ("exact") path used for odd
//                      geometries (44.1 kHz / 60 ms frames) and as the
//                      in-library cross-check of the fast kernel.
//   lld_delta_*        R13 delta/accel tail incl. the end-of-input rules.
// The fast fused kernel for Nfft = 512 lives in lld_mfcc512.hip.
#include <hip/hip_runtime.h>

#include "lld_device.hpp"
#include "lld_launch.hpp"
#include "lld_params.hpp"

namespace smilehip {

// ---------------------------------------------------------------------------
// generic path: one 256-thread workgroup per frame
// ---------------------------------------------------------------------------
// LDS: re[M] | im[M] | p[K] | lmel[n_bands]   (M = Nfft/2)
__global__ void __launch_bounds__(256) lld_mfcc_generic(LldParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int M = P.Nfft >> 1;
  float *re = smem;
  float *im = smem + M;
  float *pw = smem + 2 * M;
  float *lmel = pw + P.K + 1;

  const int64_t row = blockIdx.x;
  // utterance of this row: largest u with frame_off[u] <= row
  int lo = 0, hi = P.n_utt;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (P.frame_off[mid] <= row) lo = mid; else hi = mid;
  }
  const int u = lo;
  const int64_t t = row - P.frame_off[u];
  const int16_t *x = P.pcm + P.samp_off[u] + t * (int64_t)P.H;

  int logM = 0;
  while ((1 << logM) < M) ++logM;

  // R0..R3 + zero padding + bit-reversed load of z[i] = y[2i] + i*y[2i+1]
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = 2 * i + h - P.pad_left;     // sample index within the frame
      float y = 0.0f;
      if (n >= 0 && n < P.N) {
        const float s = pcm16_to_float(x[n]);                       // R0
        if (P.preemph) {                                            // R2
          if (n == 0) y = P.one_minus_k * s;
          else {
            const float sp = pcm16_to_float(x[n - 1]);
            y = P.de ? (s + P.k * sp) : (s - P.k * sp);
          }
        } else y = s;
        y = y * P.window[n] + P.win_offset;                         // R3
      }
      v[h] = y;
    }
    const int r = (int)(__brev((unsigned)i) >> (32 - logM));
    re[r] = v[0];
    im[r] = v[1];
  }
  __syncthreads();

  // R4: radix-2 DIT, complex length M
  block_cfft_radix2(re, im, M, P.tw_half);

  // real-FFT untangle + R5 magnitude (+ R6's squaring, melspec.cpp:520-527)
  for (int k = threadIdx.x; k <= M; k += blockDim.x) {
    const float mag = bin_magnitude(untangle_bin(re, im, M, k, P.tw_full), k == 0 || k == M);
    pw[k] = P.use_power ? mag * mag : mag;
  }
  __syncthreads();

  // R6 (melspec.cpp:544-570), one thread per band, the reference's bin order
  for (int b = threadIdx.x; b < P.n_bands; b += blockDim.x)
    lmel[b] = mel_band_exact(pw, P.mel_coef, P.mel_rng, b, P.mel_scale);
  __syncthreads();
  // log floor, mfcc.cpp:239-243
  for (int b = threadIdx.x; b < P.n_bands; b += blockDim.x)
    lmel[b] = log_mel(lmel[b], P.melfloor, P.log_floor);
  __syncthreads();
  // R7 DCT + lifter (mfcc.cpp:251-273)
  for (int r = threadIdx.x; r < P.n_mfcc; r += blockDim.x) {
    P.out[row * P.ld_out + r] = dct_coeff(lmel, P.dct_rows + r * P.n_bands, P.n_bands, P.dct_gain[r]);
  }
}

// ---------------------------------------------------------------------------
// R13: delta regression chain
// ---------------------------------------------------------------------------
// One thread per (row, column). Closed form valid for utterances longer than
// short_T frames: order-1 output d[t], t in [0, T+W), uses x with indices
// clamped to [0, T-1] (first/last frame replication at the level edges,
// dataMemoryLevel.cpp:1687-1712); order-2 output a[t], t < T, uses d on
// [0, T+W) with only the low clamp ever active. d is recomputed on the fly
// with the identical float expression, so it is bit-identical to a stored d.
__device__ __forceinline__ float delta1(const float *x, int64_t ld, int64_t T, int64_t t, int W, float norm) {
  float num = 0.0f;
  for (int i = 1; i <= W; ++i) {
    int64_t a = t - i, b = t + i;
    a = a < 0 ? 0 : (a > T - 1 ? T - 1 : a);
    b = b > T - 1 ? T - 1 : b;
    const float delta = x[b * ld] - x[a * ld];
    num += (float)i * delta;
  }
  return num / norm;
}

// Tiled form: one workgroup per tile of kDeltaTile consecutive frames of one
// utterance. The static block (with a halo of 2W frames, indices clamped to
// [0, T-1] = first/last-frame replication) is staged in LDS, the order-1 level
// d[t] is formed for t in [t0-W, t0+nF+W) (d[t<0] := d[0]: the accel
// component's own left padding), then the order-2 level. All float
// expressions are those of delta1() above, so results are bit-identical to the
// one-thread-per-element form.
constexpr int kDeltaTile = 128;
constexpr int kDeltaMaxD = 16;
constexpr int kDeltaMaxW = 4;

__global__ void __launch_bounds__(256) lld_delta_tiled(DeltaParams P) {
  __shared__ float xs[(kDeltaTile + 4 * kDeltaMaxW) * kDeltaMaxD];
  __shared__ float ds[(kDeltaTile + 2 * kDeltaMaxW) * kDeltaMaxD];
  const int u = P.tile_utt[blockIdx.x];
  const int t0 = P.tile_t0[blockIdx.x];
  const int64_t r0 = P.frame_off[u];
  const int T = (int)(P.frame_off[u + 1] - r0);
  if (T <= P.short_T) return;                 // handled by lld_delta_short
  const int D = P.D, W = P.W;
  const int nF = (T - t0 < kDeltaTile) ? T - t0 : kDeltaTile;
  const float *x = P.io + r0 * P.ld;
  // stage xs[i][d] = x[clamp(t0 - 2W + i)][d]
  const int nX = nF + 4 * W;
  for (int idx = threadIdx.x; idx < nX * D; idx += blockDim.x) {
    const int i = idx / D, d = idx - i * D;
    int tt = t0 - 2 * W + i;
    tt = tt < 0 ? 0 : (tt > T - 1 ? T - 1 : tt);
    xs[i * D + d] = x[(int64_t)tt * P.ld + d];
  }
  __syncthreads();
  // order 1: ds[i][d] = d[max(t0 - W + i, 0)]
  const int nD = nF + 2 * W;
  for (int idx = threadIdx.x; idx < nD * D; idx += blockDim.x) {
    const int i = idx / D, d = idx - i * D;
    int t = t0 - W + i;
    t = t < 0 ? 0 : t;
    const int c = t - (t0 - 2 * W);             // centre position in xs
    float num = 0.0f;
    for (int k = 1; k <= W; ++k) {
      const float delta = xs[(c + k) * D + d] - xs[(c - k) * D + d];
      num += (float)k * delta;
    }
    ds[i * D + d] = num / P.norm;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < nF * D; idx += blockDim.x) {
    const int f = idx / D, d = idx - f * D;
    float *o = P.io + (r0 + t0 + f) * P.ld + d;
    const int c = f + W;                         // position of frame t0+f in ds
    o[D] = ds[c * D + d];
    if (P.n_orders >= 2) {
      float num = 0.0f;
      for (int k = 1; k <= W; ++k) {
        const float delta = ds[(c + k) * D + d] - ds[(c - k) * D + d];
        num += (float)k * delta;
      }
      o[2 * D] = num / P.norm;
    }
  }
}

// Tick-accurate path for very short utterances (T <= short_T): the reference's
// components run in lockstep, one frame per tick, and cDataMemoryLevel::
// getMatrix reads never-written (zero) slots in its left-padding branch
// (dataMemoryLevel.cpp:1687-1698) -- see DESIGN.md "R13 end-of-input". One
// thread per (short utterance, column) replays that loop.
constexpr int kShortMaxW = kDeltaMaxW;
constexpr int kShortMaxOrders = 2;
constexpr int kShortCap = 4 * kShortMaxW + kShortMaxW * kShortMaxOrders + 2 * kShortMaxW + 2;

__global__ void __launch_bounds__(64) lld_delta_short(DeltaParams P) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int si = gid / P.D;
  const int d = gid - si * P.D;
  if (si >= P.n_short) return;
  const int u = P.short_utts[si];
  const int64_t r0 = P.frame_off[u];
  const int T = (int)(P.frame_off[u + 1] - r0);
  if (T <= 0) return;
  const int W = P.W;
  float lv[kShortMaxOrders + 1][kShortCap];
  int curW[kShortMaxOrders + 1];
  bool done[kShortMaxOrders + 1];
  for (int o = 0; o <= kShortMaxOrders; ++o) {
    curW[o] = 0; done[o] = false;
    for (int i = 0; i < kShortCap; ++i) lv[o][i] = 0.0f;
  }
  for (int t = 0; t < T; ++t) lv[0][t] = P.io[(r0 + t) * P.ld + d];
  curW[0] = T;
  for (int eoi = 0; eoi <= 1; ++eoi) {
    bool progress = true;
    while (progress) {
      progress = false;
      for (int o = 1; o <= P.n_orders; ++o) {
        if (done[o]) continue;
        const float *in = lv[o - 1];
        const int wIn = curW[o - 1];
        const int t = curW[o];
        const int vOld = t - W, vEnd = t + W + 1;
        const int v = vOld < 0 ? 0 : vOld;
        int padEnd = 0;
        if (vEnd > wIn) {
          if (!eoi) continue;
          padEnd = vEnd - wIn;
          if (padEnd >= vEnd - v) { done[o] = true; continue; }
        }
        if (!(v < wIn)) continue;
        if (t >= kShortCap) { done[o] = true; continue; }
        float num = 0.0f;
        for (int i = 1; i <= W; ++i) {
          // element at block index W+i / W-i, assembled as getMatrix does
          float hiV, loV;
          {
            const int idx = t + i;     // absolute frame index of the later sample
            if (vOld < 0) hiV = in[idx];                                  // raw read, may be past wIn (zeros)
            else if (padEnd > 0) hiV = in[idx < wIn ? idx : wIn - 1];     // replicate last written frame
            else hiV = in[idx];
          }
          {
            const int idx = t - i;
            if (idx < 0) loV = in[0];                                     // replicate first frame
            else if (vOld >= 0 && padEnd > 0) loV = in[idx < wIn ? idx : wIn - 1];
            else loV = in[idx];
          }
          const float delta = hiV - loV;
          num += (float)i * delta;
        }
        lv[o][t] = num / P.norm;
        curW[o] = t + 1;
        progress = true;
      }
    }
  }
  for (int o = 1; o <= P.n_orders; ++o)
    for (int t = 0; t < T; ++t) P.io[(r0 + t) * P.ld + o * P.D + d] = lv[o][t];
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_mfcc_generic(const LldParams &P, hipStream_t s) {
  const int M = P.Nfft / 2;
  const size_t lds = sizeof(float) * (size_t)(2 * M + P.K + 1 + P.n_bands + 8);
  hipLaunchKernelGGL(lld_mfcc_generic, dim3((unsigned)P.total_frames), dim3(256), lds, s, P);
  return hipGetLastError();
}

int delta_tile_frames() { return kDeltaTile; }

hipError_t launch_delta(const DeltaParams &P, hipStream_t s) {
  if (P.D > kDeltaMaxD || P.W > kDeltaMaxW || P.W < 1) return hipErrorInvalidValue;
  if (P.n_dtiles > 0) hipLaunchKernelGGL(lld_delta_tiled, dim3((unsigned)P.n_dtiles), dim3(256), 0, s, P);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (P.n_short > 0) {
    const int m = P.n_short * P.D;
    hipLaunchKernelGGL(lld_delta_short, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s, P);
    e = hipGetLastError();
  }
  return e;
}

}  // namespace smilehip
