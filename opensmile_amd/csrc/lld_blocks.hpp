// Building blocks shared by the reference-order kernels: reductions, the half-length complex FFT,
// the inverse real FFT cAcf needs, and cPitchACF's per-frame analysis. Every function is written
// against a "group" policy G -- the set of threads that cooperates on one frame:
//   BlockG  the whole 256-thread workgroup (barriers + LDS scratch for the reductions)
//   WaveG   one 64-lane wave (no barriers at all: a wave's LDS operations execute in order, the
//           reductions are cross-lane shuffles), so four frames proceed independently per workgroup
#pragma once
#include <hip/hip_runtime.h>

#include "lld_device.hpp"

namespace smilehip {

struct BlockG {
  __device__ static __forceinline__ int tid() { return threadIdx.x; }
  __device__ static __forceinline__ int size() { return blockDim.x; }
  __device__ static __forceinline__ void sync() { __syncthreads(); }
  __device__ static __forceinline__ double sum(double v, double *scr) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    return scr[0] + scr[1] + scr[2] + scr[3];
  }
  __device__ static __forceinline__ double max(double v, double *scr) {
    for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = w > v ? w : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    double m = scr[0];
    for (int i = 1; i < 4; ++i) m = scr[i] > m ? scr[i] : m;
    return m;
  }
  __device__ static __forceinline__ int sum_i(int v, int *scr) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    return scr[0] + scr[1] + scr[2] + scr[3];
  }
  __device__ static __forceinline__ int min_i(int v, int *scr) {
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = w < v ? w : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    int m = scr[0];
    for (int i = 1; i < 4; ++i) m = scr[i] < m ? scr[i] : m;
    return m;
  }
};

// Lane l receives lane l + off's value for the lanes a shuffle-down reduction tree needs (l < off): two gfx950 lane swaps
// and four DPP row shifts -- vector-ALU moves, where __shfl_down / __shfl_xor compile to ds_bpermute (the LDS pipe, ~10x
// the latency). Lane 0 ends with the value of `for (off = 32; off; off >>= 1) x += __shfl_down(x, off)`, which is also what
// every lane of the xor butterfly ends with (IEEE addition is commutative, the two trees are mirror images).
template <int OFF>
__device__ __forceinline__ int wave_down_i(int v) {
  if constexpr (OFF == 32) { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return r[1]; }
  else if constexpr (OFF == 16) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return r[1]; }
  else return __builtin_amdgcn_update_dpp(0, v, 0x100 + OFF, 0xf, 0xf, true);      // row_shl:OFF
}
template <int OFF>
__device__ __forceinline__ double wave_down_d(double x) {
  return __hiloint2double(wave_down_i<OFF>(__double2hiint(x)), wave_down_i<OFF>(__double2loint(x)));
}
template <class Op>
__device__ __forceinline__ double wave_tree_d(double x, Op op) {       // lane 0: the shuffle-down tree of op
  x = op(x, wave_down_d<32>(x)); x = op(x, wave_down_d<16>(x)); x = op(x, wave_down_d<8>(x));
  x = op(x, wave_down_d<4>(x)); x = op(x, wave_down_d<2>(x)); x = op(x, wave_down_d<1>(x));
  return x;
}
template <class Op>
__device__ __forceinline__ int wave_tree_i(int x, Op op) {
  x = op(x, wave_down_i<32>(x)); x = op(x, wave_down_i<16>(x)); x = op(x, wave_down_i<8>(x));
  x = op(x, wave_down_i<4>(x)); x = op(x, wave_down_i<2>(x)); x = op(x, wave_down_i<1>(x));
  return x;
}
__device__ __forceinline__ double wave_first_d(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// the greatest value of the wave, among equals the smallest index (commutative: any tree gives the same winner); every lane
// returns the winner
template <int OFF>
__device__ __forceinline__ void wave_argmax_step_f(float &bv, int &bi) {
  const float ov = __int_as_float(wave_down_i<OFF>(__float_as_int(bv)));
  const int oi = wave_down_i<OFF>(bi);
  if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}
__device__ __forceinline__ void wave_argmax_f(float &bv, int &bi) {
  wave_argmax_step_f<32>(bv, bi); wave_argmax_step_f<16>(bv, bi); wave_argmax_step_f<8>(bv, bi);
  wave_argmax_step_f<4>(bv, bi); wave_argmax_step_f<2>(bv, bi); wave_argmax_step_f<1>(bv, bi);
  bv = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bv)));
  bi = __builtin_amdgcn_readfirstlane(bi);
}

struct WaveG {                                           // (all 64 lanes call these together)
  __device__ static __forceinline__ int tid() { return threadIdx.x & 63; }
  __device__ static __forceinline__ int size() { return 64; }
  __device__ static __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __device__ static __forceinline__ double sum(double v, double *) {
    return wave_first_d(wave_tree_d(v, [](double a, double b) { return a + b; }));
  }
  __device__ static __forceinline__ double max(double v, double *) {
    return wave_first_d(wave_tree_d(v, [](double a, double b) { return b > a ? b : a; }));
  }
  __device__ static __forceinline__ int sum_i(int v, int *) {
    return __builtin_amdgcn_readfirstlane(wave_tree_i(v, [](int a, int b) { return a + b; }));
  }
  __device__ static __forceinline__ int min_i(int v, int *) {
    return __builtin_amdgcn_readfirstlane(wave_tree_i(v, [](int a, int b) { return b < a ? b : a; }));
  }
};

// Sixteen consecutive lanes (a DPP row) work on one frame, four frames per wave: the reductions are butterflies of row
// rotations (row_ror 8, 4, 2, 1 -- vector-ALU moves), after which every lane of the row holds the row's result. (Sums of doubles:
// another association than WaveG's tree; like that one it differs from the reference's sequential sum by nothing that
// survives the rounding to float that follows every use.)
template <int ROR>
__device__ __forceinline__ int row_ror_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + ROR, 0xf, 0xf, true); }
template <int ROR>
__device__ __forceinline__ double row_ror_d(double x) { return __hiloint2double(row_ror_i<ROR>(__double2hiint(x)), row_ror_i<ROR>(__double2loint(x))); }
struct QuadG {                                           // (all 64 lanes call these together)
  __device__ static __forceinline__ int tid() { return threadIdx.x & 15; }
  __device__ static __forceinline__ int size() { return 16; }
  __device__ static __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __device__ static __forceinline__ double sum(double v, double *) {
    v += row_ror_d<8>(v); v += row_ror_d<4>(v); v += row_ror_d<2>(v); v += row_ror_d<1>(v);
    return v;
  }
  __device__ static __forceinline__ double max(double v, double *) {
    double w;
    w = row_ror_d<8>(v); v = w > v ? w : v; w = row_ror_d<4>(v); v = w > v ? w : v;
    w = row_ror_d<2>(v); v = w > v ? w : v; w = row_ror_d<1>(v); v = w > v ? w : v;
    return v;
  }
  __device__ static __forceinline__ int sum_i(int v, int *) {
    v += row_ror_i<8>(v); v += row_ror_i<4>(v); v += row_ror_i<2>(v); v += row_ror_i<1>(v);
    return v;
  }
  __device__ static __forceinline__ int min_i(int v, int *) {
    int w;
    w = row_ror_i<8>(v); v = w < v ? w : v; w = row_ror_i<4>(v); v = w < v ? w : v;
    w = row_ror_i<2>(v); v = w < v ? w : v; w = row_ror_i<1>(v); v = w < v ? w : v;
    return v;
  }
};

// workgroup-wide shorthands used by the per-component kernels
__device__ __forceinline__ double block_sum(double v, double *scr) { return BlockG::sum(v, scr); }
__device__ __forceinline__ double block_max(double v, double *scr) { return BlockG::max(v, scr); }
__device__ __forceinline__ int block_sum_i(int v, int *scr) { return BlockG::sum_i(v, scr); }
__device__ __forceinline__ int block_min_i(int v, int *scr) { return BlockG::min_i(v, scr); }

// R4 core: in-place radix-2 DIT complex FFT of length M in LDS (re/im loaded in bit-reversed order)
template <class G>
__device__ __forceinline__ void group_cfft_radix2(float *re, float *im, int M, const float2 *tw_half) {
  for (int len = 2; len <= M; len <<= 1) {
    const int half = len >> 1;
    const int tstep = M / len;
    for (int b = G::tid(); b < (M >> 1); b += G::size()) {
      const int j = b & (half - 1);
      const int i0 = ((b - j) << 1) + j;
      const int i1 = i0 + half;
      const float2 w = tw_half[j * tstep];
      const float xr = re[i1], xi = im[i1];
      const float tr = fmaf(xr, w.x, -xi * w.y);
      const float ti = fmaf(xr, w.y, xi * w.x);
      const float ar = re[i0], ai = im[i0];
      re[i1] = ar - tr; im[i1] = ai - ti;
      re[i0] = ar + tr; im[i0] = ai + ti;
    }
    G::sync();
  }
}

// Inverse of the packed real FFT for a purely real spectrum R[0..M] (what cAcf feeds
// Ooura's rdft(n,-1), fftsg.c:103-135):  a[k] = R0/2 + R_M (-1)^k / 2 + sum_j R_j cos(2 pi jk/n).
// Computed as half the forward DFT of the even extension s[j] = s[n-j] = R_j, through the
// same half-length complex FFT + untangle the forward transform uses.
template <class G>
__device__ __forceinline__ void group_irfft_even(const float *R, float *re, float *im, int M, int logM, const float2 *tw_half,
                                                 const float2 *tw_full, float *out, float inv_norm, bool take_abs) {
  const int n = 2 * M;
  for (int i = G::tid(); i < M; i += G::size()) {
    const int n0 = 2 * i, n1 = 2 * i + 1;
    const float v0 = R[n0 <= M ? n0 : n - n0];
    const float v1 = R[n1 <= M ? n1 : n - n1];
    const int r = (int)(__brev((unsigned)i) >> (32 - logM));
    re[r] = v0;
    im[r] = v1;
  }
  G::sync();
  group_cfft_radix2<G>(re, im, M, tw_half);
  for (int k = G::tid(); k < M; k += G::size()) {
    const float a = 0.5f * untangle_bin(re, im, M, k, tw_full).x;
    const float v = a / inv_norm;                       // acf.cpp:321-325: (FLOAT_DMEM)data / (FLOAT_DMEM)Nsrc
    out[k] = take_abs ? fabsf(v) : v;
  }
  G::sync();
}
__device__ __forceinline__ void irfft_even(const float *R, float *re, float *im, int M, int logM, const float2 *tw_half,
                                           const float2 *tw_full, float *out, float inv_norm, bool take_abs) {
  group_irfft_even<BlockG>(R, re, im, M, logM, tw_half, tw_full, out, inv_norm, take_abs);
}

// R10 cPitchACF::processVector, per-frame part (pitchACF.cpp:137-192): voicing probability from
// the ACF (voicingProb, :249-284) and the index of the first cepstral peak above
// 0.6 * (max + mean|.|) (pitchPeak, :286-310). acf / cep: n values each, in LDS or global.
// Every thread of the group returns the same (voicing, maxIdx).
template <class G>
__device__ __forceinline__ void group_pitchacf_frame(const float *acf, const float *cep, int n, double fsSec, double maxPitch,
                                                     double *scr, int *iscr, double &voicing, int &max_idx, double &Tsamp_out) {
  const double Nd = (double)(2 * n);
  const double Tsamp = fsSec / Nd;
  Tsamp_out = Tsamp;
  const int preskip = (maxPitch <= 0.0) ? 0 : (int)(1.0 / (maxPitch * Tsamp));
  double vmax = acf[n - 1];
  for (int i = 1 + G::tid(); i < n; i += G::size())
    if (i >= preskip && (acf[i] > vmax) && (acf[i - 1] < acf[i])) vmax = acf[i];
  // (the reference's running-max test "a[i] > max" only ever raises max, so the result is the
  //  maximum over the qualifying set; taking it in parallel gives the same value)
  vmax = G::max(vmax, scr);
  voicing = (acf[0] > 0.0f) ? vmax / (double)acf[0] : 0.0;
  const int skip = preskip + 1;
  double csum = 0.0, cmax = cep[n - 1];
  for (int i = G::tid(); i < n; i += G::size()) {
    const double b = cep[i];
    csum += fabs(b);
    if (i >= skip && b > cmax) cmax = b;
  }
  csum = G::sum(csum, scr) / n;
  cmax = G::max(cmax, scr);
  const double thr = (cmax + csum) * 0.6;
  int first = 1 << 30;
  for (int i = skip + 1 + G::tid(); i < n - 1; i += G::size())
    if ((double)cep[i] > thr && (cep[i - 1] < cep[i]) && (cep[i] > cep[i + 1])) { first = i; break; }
  first = G::min_i(first, iscr);
  max_idx = (first == (1 << 30)) ? 0 : first;
}
__device__ __forceinline__ void pitchacf_frame(const float *acf, const float *cep, int n, double fsSec, double maxPitch,
                                               double *scr, int *iscr, double &voicing, int &max_idx, double &Tsamp_out) {
  group_pitchacf_frame<BlockG>(acf, cep, n, fsSec, maxPitch, scr, iscr, voicing, max_idx, Tsamp_out);
}

}  // namespace smilehip
