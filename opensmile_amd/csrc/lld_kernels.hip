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
#include "kernel_timing.hpp"

#include "lld_device.hpp"
#include "lld_blocks.hpp"
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
  const PcmIn x = pcm_in(P) + (P.samp_off[u] + t * (int64_t)P.H);

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
        const float s = x[n];                                       // R0 (or already done: float input)
        if (P.preemph) {                                            // R2
          if (n == 0) y = P.one_minus_k * s;
          else {
            const float sp = x[n - 1];
            y = P.de ? (s + P.k * sp) : (s - P.k * sp);
          }
        } else y = s;
        y = y * P.window[n] + P.win_offset;                         // R3
      }
      v[h] = y;
    }
    if (P.oo.tw) {                                         // reference-order transform: natural order, (re, im) pairs
      reinterpret_cast<float2 *>(smem)[i] = make_float2(v[0], v[1]);
    } else {
      const int r = (int)(__brev((unsigned)i) >> (32 - logM));
      re[r] = v[0];
      im[r] = v[1];
    }
  }
  __syncthreads();

  // R4: the reference's rdft network (lld_ooura.hpp), or the radix-2 DIT of round 2 (SMILEHIP_FFT=radix2)
  if (P.oo.tw) ooura_levels<BlockG, false>(reinterpret_cast<float2 *>(smem), P.oo);
  else block_cfft_radix2(re, im, M, P.tw_half);

  // real-FFT untangle + R5 magnitude (+ R6's squaring, melspec.cpp:520-527)
  for (int k = threadIdx.x; k <= M; k += blockDim.x) {
    const float2 X = P.oo.tw ? ooura_bin(reinterpret_cast<const float2 *>(smem), P.oo, k) : untangle_bin(re, im, M, k, P.tw_full);
    const float mag = bin_magnitude(X, k == 0 || k == M);
    pw[k] = P.use_power ? mag * mag : mag;
  }
  __syncthreads();

  // R6 (melspec.cpp:544-570), one thread per band, the reference's bin order
  for (int b = threadIdx.x; b < P.n_bands; b += blockDim.x)
    lmel[b] = mel_band_exact(pw, P.mel_coef, P.mel_rng, b, P.mel_scale);
  __syncthreads();
  if (P.plp) {
    // R8 cPlp with IDFT / LP / cepstra (plp.cpp:499-583): auditory spectrum, autocorrelation, Durbin, cepstra, lifter
    float *acf = lmel + P.n_bands;                      // 16 floats of slack behind the bands
    for (int b = threadIdx.x; b < P.n_bands; b += blockDim.x) {
      float v = lmel[b];
      if (v < P.melfloor) v = P.melfloor;
      v *= P.plp_eql[b];
      lmel[b] = (float)pow((double)v, (double)P.plp_compression);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= P.plp_order; i += blockDim.x) acf[i] = plp_acf_lag(lmel, P.plp_cos + i * (P.n_bands + 2), P.n_bands);
    __syncthreads();
    if (threadIdx.x == 0) {
      float o[16];
      plp_cc_serial(acf, P.plp_order, P.plp_sin, o);
      for (int r = 0; r < P.n_mfcc; ++r) P.out[row * P.ld_out + r] = o[r];      // firstCC = 1 drops c0 (the last one)
    }
    return;
  }
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
// R13: window-processor chain (cDeltaRegression, cContourSmoother)
// ---------------------------------------------------------------------------
// One stage applied at the centre position c of a staged level (stride D):
//   kind 0: cDeltaRegression::processBuffer (deltaRegression.cpp:144-152, norm :77-79)
//   kind 1: cContourSmoother::processBuffer (contourSmoother.cpp:104-111)
__device__ __forceinline__ float chain_op(const float *lv, int c, int D, int d, int kind, int W) {
  if (kind == 0) {
    float norm = 0.0f;
    for (int i = 1; i <= W; ++i) norm += (float)i * (float)i;
    norm *= 2.0;
    float num = 0.0f;
    for (int k = 1; k <= W; ++k) {
      const float delta = lv[(c + k) * D + d] - lv[(c - k) * D + d];
      num += (float)k * delta;
    }
    return num / norm;
  }
  float acc = lv[c * D + d];
  for (int k = 1; k <= W; ++k) { acc += lv[(c - k) * D + d]; acc += lv[(c + k) * D + d]; }
  return acc / (float)(2 * W + 1);
}

// Tiled closed form, valid for utterances longer than short_T frames: level s+1 at
// index t uses level s at t-W..t+W with indices clamped to the level's own range
// (first/last-frame replication, dataMemoryLevel.cpp:1687-1712); level s has
// T + W_1 + .. + W_s rows, all of which the next stage consumes as real data.
// One workgroup per tile of kChainTile output rows; levels are staged in LDS with
// the halo the later stages need. Float expressions are the reference's.
constexpr int kChainTile = 128;
constexpr int kChainMaxD = 16;
constexpr int kChainMaxW = 4;

__global__ void __launch_bounds__(256) lld_chain_tiled(ChainParams P) {
  constexpr int S = kChainMaxD;               // LDS row stride; thread = (row lane r, column d): no integer divisions
  __shared__ float l0[(kChainTile + 4 * kChainMaxW) * S];
  __shared__ float l1[(kChainTile + 2 * kChainMaxW) * S];
  // Natural tile order: workgroup b takes tile b. An XCD-grouped order (tile (b % 8) * ceil(n/8) + b / 8, so that
  // neighbouring tiles share one L2) was measured 19 % SLOWER for this streaming kernel (0.063 vs 0.053 ms): the eight
  // XCDs then stream eight distant regions instead of one, and the shared halo is only 6 % of a tile.
  // Threads walking the tile's input / output elements in memory order (whole cache lines per store instead of 52-byte row
  // pieces, group and column recovered per element) was measured too: 0.094 vs 0.056 ms -- the divergent per-element
  // selection costs more than the partial lines, which L2 merges anyway.
  const int tile = (int)blockIdx.x;
  if (tile >= P.n_tiles) return;
  const int u = P.tile_utt[tile];
  const int t0 = P.tile_t0[tile];
  const int64_t f0 = P.frame_off[u];
  const int T = (int)(P.frame_off[u + 1] - f0);
  if (T <= P.short_T) return;                 // handled by lld_chain_short
  const int rows = (int)(P.row_off[u + 1] - P.row_off[u]);
  const int c0 = blockIdx.y * kChainMaxD;     // column block
  const int D = (P.D - c0 < kChainMaxD) ? P.D - c0 : kChainMaxD;
  const int W1 = P.W[0], W2 = (P.n_stages > 1) ? P.W[1] : 0;
  const int nF = (rows - t0 < kChainTile) ? rows - t0 : kChainTile;
  const int L1 = T + W1;                      // rows of level 1
  const float *x = P.x + f0 * P.ld_x + c0;
  const int d = threadIdx.x & (S - 1), r = threadIdx.x / S;
  constexpr int R = 256 / S;
  const bool on = d < D;
  // level 0 at positions [t0 - W1 - W2, t0 + nF + W1 + W2), clamped to [0, T-1]
  const int n0 = nF + 2 * (W1 + W2);
  if (on)
    for (int i = r; i < n0; i += R) {
      int tt = t0 - W1 - W2 + i;
      tt = tt < 0 ? 0 : (tt > T - 1 ? T - 1 : tt);
      l0[i * S + d] = x[(int64_t)tt * P.ld_x + d];
    }
  __syncthreads();
  // level 1 at positions [t0 - W2, t0 + nF + W2), index clamped to [0, L1-1]
  const int n1 = nF + 2 * W2;
  if (on)
    for (int i = r; i < n1; i += R) {
      int t = t0 - W2 + i;
      t = t < 0 ? 0 : (t > L1 - 1 ? L1 - 1 : t);
      l1[i * S + d] = chain_op(l0, t - (t0 - W1 - W2), S, d, P.kind[0], W1);
    }
  __syncthreads();
  float *o = P.out + (P.row_off[u] + t0) * P.ld_out;
  if (on)
    for (int f = r; f < nF; f += R) {
      float *orow = o + (int64_t)f * P.ld_out + c0 + d;
      if (P.copy_col >= 0) orow[P.copy_col] = l0[(f + W1 + W2) * S + d];      // rows == frames for chains that copy
      orow[P.out_col[0]] = l1[(f + W2) * S + d];
      if (P.n_stages > 1) orow[P.out_col[1]] = chain_op(l1, f + W2, S, d, P.kind[1], W2);
    }
}

// Tick-accurate path for very short utterances (T <= short_T): the reference's
// components run in lockstep, one frame per tick, and cDataMemoryLevel::getMatrix
// reads never-written (zero) slots in its left-padding branch
// (dataMemoryLevel.cpp:1687-1698) -- see DESIGN.md "R13 end-of-input". One thread
// per (short utterance, column) replays that loop.
constexpr int kShortMaxT = 16;
constexpr int kShortCap = kShortMaxT + 6 * kChainMaxW + 4;

__global__ void __launch_bounds__(64) lld_chain_short(ChainParams P) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int si = gid / P.D;
  const int d = gid - si * P.D;
  if (si >= P.n_short) return;
  const int u = P.short_utts[si];
  const int64_t f0 = P.frame_off[u];
  const int T = (int)(P.frame_off[u + 1] - f0);
  if (T <= 0) return;
  float lv[3][kShortCap];
  int curW[3];
  bool done[3];
  for (int o = 0; o < 3; ++o) {
    curW[o] = 0; done[o] = false;
    for (int i = 0; i < kShortCap; ++i) lv[o][i] = 0.0f;
  }
  for (int t = 0; t < T; ++t) lv[0][t] = P.x[(f0 + t) * P.ld_x + d];
  curW[0] = T;
  for (int eoi = 0; eoi <= 1; ++eoi) {
    bool progress = true;
    while (progress) {
      progress = false;
      for (int o = 1; o <= P.n_stages; ++o) {
        if (done[o]) continue;
        const int W = P.W[o - 1];
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
        if (t >= kShortCap - 1) { done[o] = true; continue; }
        // block [t-W, t+W] assembled as getMatrix does
        float blk[2 * kChainMaxW + 1];
        for (int k = -W; k <= W; ++k) {
          const int idx = t + k;
          float val;
          if (idx < 0) val = in[0];                                        // replicate first frame
          else if (vOld < 0) val = in[idx];                                // raw read, may be past wIn (zeros)
          else if (padEnd > 0) val = in[idx < wIn ? idx : wIn - 1];        // replicate last written frame
          else val = in[idx];
          blk[k + W] = val;
        }
        lv[o][t] = chain_op(blk, W, 1, 0, P.kind[o - 1], W);
        curW[o] = t + 1;
        progress = true;
      }
    }
  }
  const int rows = (int)(P.row_off[u + 1] - P.row_off[u]);
  for (int o = 1; o <= P.n_stages; ++o)
    for (int t = 0; t < rows && t < kShortCap; ++t) P.out[(P.row_off[u] + t) * P.ld_out + P.out_col[o - 1] + d] = lv[o][t];
  if (P.copy_col >= 0)
    for (int t = 0; t < T && t < rows; ++t) P.out[(P.row_off[u] + t) * P.ld_out + P.copy_col + d] = lv[0][t];
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_mfcc_generic(const LldParams &P, hipStream_t s) {
  const int M = P.Nfft / 2;
  const size_t lds = sizeof(float) * (size_t)(2 * M + P.K + 1 + P.n_bands + 24);
  SMILEHIP_KLAUNCH(lld_mfcc_generic, dim3((unsigned)P.total_frames), dim3(256), lds, s, P);
  return hipGetLastError();
}

// [energy:cEnergy] of the E variants (src/lldcore/energy.cpp:152-185 with log = 1, htkcompatible = 1) on the RAW frame:
// d = sum (float)(x*x) accumulated in double in sample order, times 32767^2, floored at 1, (float)log(d) -- the
// reference's own summation order, so the column is bit-exact. One wave per tile of output rows (the window-chain
// kernel's tiles), 32 frames at a time: the squares of the frames' common span are staged in LDS (coalesced PCM reads,
// one pad word per hop so that the 32 lanes' strided reads fall into different banks), then lane f sums frame f.
constexpr int kEnFrames = 32;
__global__ void __launch_bounds__(64) lld_log_energy(LldParams P, const int32_t *tile_utt, const int32_t *tile_t0, int tile_rows,
                                                     float *dst, int64_t ld, int col) {
  extern __shared__ float s_sq[];
  const int u = tile_utt[blockIdx.x];
  const int t_first = tile_t0[blockIdx.x];
  const int64_t f0 = P.frame_off[u];
  const int T = (int)(P.frame_off[u + 1] - f0);
  const int t_end = (t_first + tile_rows < T) ? t_first + tile_rows : T;
  const int lane = threadIdx.x;
  for (int tc = t_first; tc < t_end; tc += kEnFrames) {
    const int nf = (t_end - tc < kEnFrames) ? t_end - tc : kEnFrames;
    const int span = (nf - 1) * P.H + P.N;
    __syncthreads();
    const int pitch = P.H + 1;                            // one pad word per hop: lane-strided reads hit different banks
    // the PCM of the span in 16-byte vectors (8 samples) from the 16-byte boundary below its first sample; vectors that
    // would reach past the end of the packed buffer are read sample by sample
    const float rH = 1.0f / (float)P.H;                   // hop of sample i: floor((i + 0.5) / H), exact for these sizes
    const int64_t first = P.samp_off[u] + (int64_t)tc * P.H;          // absolute sample index of the span's start
    const int64_t a16 = first & ~(int64_t)7;
    const int par = (int)(first - a16);
    const int nvec = (par + span + 7) >> 3;
    for (int k0 = lane; k0 < nvec; k0 += 64 * 4) {
      int16_t r[4][8];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int k = k0 + 64 * m;
        if (k < nvec) {
          const int64_t s8 = a16 + 8 * (int64_t)k;
          if (P.pcm_f32) {
            // float input: the samples are read where they are used, below
          } else if (s8 + 8 <= P.pcm_total) {
            const uint4 v = *reinterpret_cast<const uint4 *>(P.pcm + s8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) r[m][q] = (int16_t)(w[q >> 1] >> (16 * (q & 1)));
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[m][q] = (s8 + q < P.pcm_total) ? P.pcm[s8 + q] : (int16_t)0;
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int k = k0 + 64 * m;
        if (k < nvec) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int i = 8 * k + q - par;
            if (i >= 0 && i < span) {
              const int h = (int)(((float)i + 0.5f) * rH);
              const float v = P.pcm_f32 ? P.pcm_f32[a16 + 8 * (int64_t)k + q] : pcm16_to_float(r[m][q]);
              s_sq[h * pitch + (i - h * P.H)] = v * v;
            }
          }
        }
      }
    }
    __syncthreads();
    if (lane < nf) {
      double d = 0.0;
      for (int h = 0, done = 0; done < P.N; ++h) {        // frame `lane` = hops lane, lane+1, ... of the span
        const int len = (P.N - done < P.H) ? P.N - done : P.H;
        const float *row = s_sq + (lane + h) * pitch;
        int j = 0;
        for (; j + 8 <= len; j += 8) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = row[j + q];
#pragma unroll
          for (int q = 0; q < 8; ++q) d += (double)v[q];
        }
        for (; j < len; ++j) d += (double)row[j];
        done += len;
      }
      d *= 32767.0 * 32767.0;
      if (d <= 1.0) d = 1.0;
      dst[(f0 + tc + lane) * ld + col] = (float)log(d) * 1.0f + 0.0f;
    }
  }
}

// [cms:cFullinputMean] of the Z variants (src/dspcore/fullinputMean.cpp, multiLoopMode = 0, meanNorm = amean): per column
// the float sum of all frames in order (first frame, then += ...), divided by (float)T, subtracted from every frame.
// x: the un-normalised static block (ld_x), out: the output rows (ld_out); columns 0 .. n_cols-1. One wave per utterance.
__global__ void __launch_bounds__(64) lld_cms(const int64_t *frame_off, int n_utt, const float *x, int64_t ld_x, float *out,
                                              int64_t ld_out, int n_cols) {
  const int u = blockIdx.x;
  if (u >= n_utt) return;
  const int64_t f0 = frame_off[u];
  const int T = (int)(frame_off[u + 1] - f0);
  if (T <= 0) return;
  const int lane = threadIdx.x;
  __shared__ float mean[64];
  if (lane < n_cols) {
    const float *p = x + f0 * ld_x + lane;
    float m = p[0];
    int t = 1;
    for (; t + 8 <= T; t += 8) {                          // eight rows' loads in flight, the sum stays sequential
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(t + q) * ld_x];
#pragma unroll
      for (int q = 0; q < 8; ++q) m += v[q];
    }
    for (; t < T; ++t) m += p[(int64_t)t * ld_x];
    mean[lane] = m / (float)T;
  }
  __syncthreads();
  for (int64_t e = lane; e < (int64_t)T * n_cols; e += 64) {
    const int64_t t = e / n_cols;
    const int c = (int)(e - t * n_cols);
    out[(f0 + t) * ld_out + c] = x[(f0 + t) * ld_x + c] - mean[c];
  }
}

hipError_t launch_log_energy(const LldParams &P, const int32_t *d_tile_utt, const int32_t *d_tile_t0, int n_tiles, float *dst,
                             int64_t ld, int col, hipStream_t s) {
  if (P.total_frames <= 0 || n_tiles <= 0) return hipSuccess;
  const int span = (kEnFrames - 1) * P.H + P.N;
  const size_t lds = sizeof(float) * (size_t)((span / (P.H > 0 ? P.H : 1) + 2) * (P.H + 1));
  if (lds > 150 * 1024) return hipErrorInvalidValue;     // frame geometry far outside the speech configs (25 ms / 10 ms at 48 kHz: 67 KB)
  if (lds > 48 * 1024) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&lld_log_energy), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;
  }
  SMILEHIP_KLAUNCH(lld_log_energy, dim3((unsigned)n_tiles), dim3(64), lds, s, P, d_tile_utt, d_tile_t0, kChainTile, dst, ld, col);
  return hipGetLastError();
}

hipError_t launch_cms(const int64_t *d_frame_off, int n_utt, const float *x, int64_t ld_x, float *out, int64_t ld_out, int n_cols,
                      hipStream_t s) {
  if (n_utt <= 0 || n_cols <= 0) return hipSuccess;
  if (n_cols > 64) return hipErrorInvalidValue;
  SMILEHIP_KLAUNCH(lld_cms, dim3((unsigned)n_utt), dim3(64), 0, s, d_frame_off, n_utt, x, ld_x, out, ld_out, n_cols);
  return hipGetLastError();
}

int chain_tile_rows() { return kChainTile; }
int chain_short_max() { return kShortMaxT; }

hipError_t launch_chain(const ChainParams &P, hipStream_t s) {
  if (P.D < 1 || P.D > 4 * kChainMaxD || P.n_stages < 1 || P.n_stages > 2) return hipErrorInvalidValue;
  for (int i = 0; i < P.n_stages; ++i)
    if (P.W[i] < 1 || P.W[i] > kChainMaxW) return hipErrorInvalidValue;
  if (P.short_T > kShortMaxT) return hipErrorInvalidValue;
  if (P.n_tiles > 0)
    SMILEHIP_KLAUNCH(lld_chain_tiled, dim3((unsigned)P.n_tiles, (unsigned)((P.D + kChainMaxD - 1) / kChainMaxD)), dim3(256), 0, s, P);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (P.n_short > 0) {
    const int m = P.n_short * P.D;
    SMILEHIP_KLAUNCH(lld_chain_short, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s, P);
    e = hipGetLastError();
  }
  return e;
}

}  // namespace smilehip
