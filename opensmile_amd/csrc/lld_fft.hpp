// The half-length complex FFT of the wave-per-frame kernels, fused form (round 2).
//
// group_cfft_radix2 (lld_blocks.hpp) runs log2 M radix-2 stages in place in LDS: per stage and butterfly four b32 reads and
// four b32 writes, 2-way bank conflicts in the five stages with half < 32, one wave barrier per stage -- 288 LDS
// instructions per lane for M = 512, and the kernels that use it sit at ~50 % LDS-pipe busy with 20-45 % of it conflicts
// (profiles/r02_pmc_*.txt). WaveFft keeps the SAME butterflies -- the same twiddle table entries, the same fmaf / multiply /
// add per butterfly in the same order, so every output is bit-identical to the radix-2 form -- and changes only where the
// operands live between stages:
//   * complex points are (re, im) pairs: one b64 access instead of two b32;
//   * three (M = 512: 8 points per lane, passes with halves 1-2-4 | 8-16-32 | 64-128-256) or two (M = 256: 4 points per
//     lane, halves 1-2 | 4-8 | 16-32 | 64-128) consecutive stages run on one lane's registers between an LDS read and an
//     LDS write; the first pass takes its operands straight from the caller's loader (the bit reversal is the choice of
//     which inputs a lane asks for), so the scatter into bit-reversed order is gone too;
//   * between passes the points sit in a padded order chosen so that the write of one pass and the read of the next are
//     both conflict-free for b64 (32 lanes cover the 32 bank pairs): pos1 / pos2 below.
// M = 512: 40 b64 LDS instructions per lane and three wave barriers instead of 288 b32 and nine.
// One wave per transform (LDS operations of a wave execute in order, so a pass may write a different order than it read).
#pragma once
#include <hip/hip_runtime.h>

namespace smilehip {

// one butterfly of group_cfft_radix2: (p0, p1) <- (p0 + w p1, p0 - w p1), the operations in its order
__device__ __forceinline__ void fft_bfly(float2 &p0, float2 &p1, const float2 w) {
  const float xr = p1.x, xi = p1.y;
  const float tr = fmaf(xr, w.x, -xi * w.y);
  const float ti = fmaf(xr, w.y, xi * w.x);
  const float ar = p0.x, ai = p0.y;
  p1.x = ar - tr; p1.y = ai - ti;
  p0.x = ar + tr; p0.y = ai + ti;
}

// Three consecutive stages (halves H0, 2 H0, 4 H0) on the eight points base + H0 * m, m = 0..7; jb = base & (H0 - 1).
template <int M, int H0>
__device__ __forceinline__ void fft_pass8(float2 (&v)[8], int jb, const float2 *tw) {
  {
    const float2 w = tw[jb * (M / (2 * H0))];
    fft_bfly(v[0], v[1], w); fft_bfly(v[2], v[3], w); fft_bfly(v[4], v[5], w); fft_bfly(v[6], v[7], w);
  }
  {
    const float2 w0 = tw[jb * (M / (4 * H0))], w1 = tw[(jb + H0) * (M / (4 * H0))];
    fft_bfly(v[0], v[2], w0); fft_bfly(v[1], v[3], w1); fft_bfly(v[4], v[6], w0); fft_bfly(v[5], v[7], w1);
  }
  {
#pragma unroll
    for (int k = 0; k < 4; ++k) fft_bfly(v[k], v[k + 4], tw[(jb + k * H0) * (M / (8 * H0))]);
  }
}
// Two consecutive stages (halves H0, 2 H0) on the four points base + H0 * m, m = 0..3
template <int M, int H0>
__device__ __forceinline__ void fft_pass4(float2 (&v)[4], int jb, const float2 *tw) {
  {
    const float2 w = tw[jb * (M / (2 * H0))];
    fft_bfly(v[0], v[1], w); fft_bfly(v[2], v[3], w);
  }
  {
    const float2 w0 = tw[jb * (M / (4 * H0))], w1 = tw[(jb + H0) * (M / (4 * H0))];
    fft_bfly(v[0], v[2], w0); fft_bfly(v[1], v[3], w1);
  }
}

__device__ __forceinline__ void fft_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int LOGM> struct WaveFft;

// M = 512 (FFT 1024): z holds kZ = 576 (re, im) pairs.
template <> struct WaveFft<9> {
  static constexpr int M = 512, kZ = 576;
  __device__ static __forceinline__ int pos(int e) { return e + ((e >> 6) << 3); }   // where Z[e] sits after forward()
  // load(n): the n-th complex input (x[2n], x[2n+1]) in natural order; every n in [0, M) is asked for exactly once
  template <class Load>
  __device__ static __forceinline__ void forward(float2 *z, const float2 *tw_half, int lane, Load load) {
    float2 v[8];
    {   // stages 1-2-4 on points 8 lane + k: point e of the bit-reversed order is input brev9(e)
      const int rg = (int)(__brev((unsigned)lane) >> 26);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[((k & 1) << 2) | (k & 2) | (k >> 2)] = load(rg + 64 * k);
      fft_pass8<M, 1>(v, 0, tw_half);
#pragma unroll
      for (int k = 0; k < 8; ++k) z[9 * lane + k] = v[k];                        // e + (e >> 3)
    }
    fft_wave_sync();
    {   // stages 8-16-32 on points jb + 8 m + 64 hi
      const int jb = lane & 7, hi = lane >> 3;
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = z[jb + 72 * hi + 9 * m];
      fft_pass8<M, 8>(v, jb, tw_half);
#pragma unroll
      for (int m = 0; m < 8; ++m) z[jb + 72 * hi + 8 * m] = v[m];                // e + 8 (e >> 6)
    }
    fft_wave_sync();
    {   // stages 64-128-256 on points lane + 64 m
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = z[lane + 72 * m];
      fft_pass8<M, 64>(v, lane, tw_half);
#pragma unroll
      for (int m = 0; m < 8; ++m) z[lane + 72 * m] = v[m];
    }
    fft_wave_sync();
  }
};

// M = 256 (FFT 512): z holds kZ = 320 pairs.
template <> struct WaveFft<8> {
  static constexpr int M = 256, kZ = 320;
  __device__ static __forceinline__ int pos(int e) { return e + ((e >> 6) << 4); }
  template <class Load>
  __device__ static __forceinline__ void forward(float2 *z, const float2 *tw_half, int lane, Load load) {
    float2 v[4];
    {   // stages 1-2 on points 4 lane + k
      const int rg = (int)(__brev((unsigned)lane) >> 26);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[((k & 1) << 1) | (k >> 1)] = load(rg + 64 * k);
      fft_pass4<M, 1>(v, 0, tw_half);
#pragma unroll
      for (int k = 0; k < 4; ++k) z[5 * lane + k] = v[k];                        // e + (e >> 2)
    }
    fft_wave_sync();
    {   // stages 4-8 on points jb + 4 m + 16 hi
      const int jb = lane & 3, hi = lane >> 2;
#pragma unroll
      for (int m = 0; m < 4; ++m) v[m] = z[jb + 20 * hi + 5 * m];
      fft_pass4<M, 4>(v, jb, tw_half);
#pragma unroll
      for (int m = 0; m < 4; ++m) z[jb + 20 * hi + 4 * m] = v[m];                // e + 4 (e >> 4)
    }
    fft_wave_sync();
    {   // stages 16-32 on points jb + 16 m + 64 hi
      const int jb = lane & 15, hi = lane >> 4;
#pragma unroll
      for (int m = 0; m < 4; ++m) v[m] = z[jb + 80 * hi + 20 * m];
      fft_pass4<M, 16>(v, jb, tw_half);
#pragma unroll
      for (int m = 0; m < 4; ++m) z[jb + 80 * hi + 16 * m] = v[m];               // e + 16 (e >> 6)
    }
    fft_wave_sync();
    {   // stages 64-128 on points lane + 64 m
#pragma unroll
      for (int m = 0; m < 4; ++m) v[m] = z[lane + 80 * m];
      fft_pass4<M, 64>(v, lane, tw_half);
#pragma unroll
      for (int m = 0; m < 4; ++m) z[lane + 80 * m] = v[m];
    }
    fft_wave_sync();
  }
};

// untangle_bin (lld_device.hpp) on the pair layout: the same arithmetic, Z[e] at z[F::pos(e)]
template <class F>
__device__ __forceinline__ float2 fft_untangle(const float2 *z, int k, const float2 *tw_full) {
  constexpr int M = F::M;
  if (k == 0) { const float2 a = z[0]; return make_float2(a.x + a.y, 0.0f); }
  if (k == M) { const float2 a = z[0]; return make_float2(a.x - a.y, 0.0f); }
  const float2 p = z[F::pos(k)], q = z[F::pos(M - k)];
  const float a = p.x, b = p.y, c = q.x, d = q.y;
  const float2 w = (k <= (M >> 1)) ? tw_full[k] : make_float2(-tw_full[M - k].x, tw_full[M - k].y);
  const float sr = a + c, si = b - d, dr = a - c, di = b + d;
  return make_float2(0.5f * fmaf(w.x, di, fmaf(w.y, dr, sr)), 0.5f * fmaf(w.y, di, fmaf(-w.x, dr, si)));
}

// ---- run-time M (the kernels whose frame geometry is a parameter): the fused forms for M = 512 / 256, otherwise the
// in-place radix-2 stages on pairs in natural order. fft_pairs(M): pairs of LDS the transform needs; fft_pad(M): Z[e] sits at
// z[e + (e >> 6) * fft_pad(M)].
__host__ __device__ inline int fft_pairs(int M) { return M == 512 ? WaveFft<9>::kZ : (M == 256 ? WaveFft<8>::kZ : M); }
__host__ __device__ inline int fft_pad(int M) { return M == 512 ? 8 : (M == 256 ? 16 : 0); }

template <class Load>
__device__ __forceinline__ void wave_cfft(float2 *z, int M, const float2 *tw_half, int lane, Load load) {
  if (M == 512) { WaveFft<9>::forward(z, tw_half, lane, load); return; }
  if (M == 256) { WaveFft<8>::forward(z, tw_half, lane, load); return; }
  int logM = 0;
  while ((1 << logM) < M) ++logM;
  for (int i = lane; i < M; i += 64) z[(int)(__brev((unsigned)i) >> (32 - logM))] = load(i);
  fft_wave_sync();
  for (int len = 2; len <= M; len <<= 1) {
    const int half = len >> 1, tstep = M / len;
    for (int b = lane; b < (M >> 1); b += 64) {
      const int j = b & (half - 1);
      const int i0 = ((b - j) << 1) + j;
      float2 p0 = z[i0], p1 = z[i0 + half];
      fft_bfly(p0, p1, tw_half[j * tstep]);
      z[i0 + half] = p1;
      z[i0] = p0;
    }
    fft_wave_sync();
  }
}

__device__ __forceinline__ float2 wave_untangle(const float2 *z, int M, int pad, int k, const float2 *tw_full) {
  if (k == 0) { const float2 a = z[0]; return make_float2(a.x + a.y, 0.0f); }
  if (k == M) { const float2 a = z[0]; return make_float2(a.x - a.y, 0.0f); }
  const int k2 = M - k;
  const float2 p = z[k + (k >> 6) * pad], q = z[k2 + (k2 >> 6) * pad];
  const float a = p.x, b = p.y, c = q.x, d = q.y;
  const float2 w = (k <= (M >> 1)) ? tw_full[k] : make_float2(-tw_full[M - k].x, tw_full[M - k].y);
  const float sr = a + c, si = b - d, dr = a - c, di = b + d;
  return make_float2(0.5f * fmaf(w.x, di, fmaf(w.y, dr, sr)), 0.5f * fmaf(w.y, di, fmaf(-w.x, dr, si)));
}

// group_irfft_even (lld_blocks.hpp) on the fused transform: out[k] = (Re X[k] / 2) / inv_norm (or its magnitude), k < M, of
// the even extension of the real spectrum R[0..M]
__device__ __forceinline__ void wave_irfft_even(const float *R, float2 *z, int M, const float2 *tw_half, const float2 *tw_full,
                                                float *out, float inv_norm, bool take_abs, int lane) {
  const int n = 2 * M, pad = fft_pad(M);
  wave_cfft(z, M, tw_half, lane, [&](int i) {
    const int n0 = 2 * i, n1 = 2 * i + 1;
    return make_float2(R[n0 <= M ? n0 : n - n0], R[n1 <= M ? n1 : n - n1]);
  });
  for (int k = lane; k < M; k += 64) {
    const float a = 0.5f * wave_untangle(z, M, pad, k, tw_full).x;
    const float v = a / inv_norm;                       // acf.cpp:321-325: (FLOAT_DMEM)data / (FLOAT_DMEM)Nsrc
    out[k] = take_abs ? fabsf(v) : v;
  }
  fft_wave_sync();
}

}  // namespace smilehip
