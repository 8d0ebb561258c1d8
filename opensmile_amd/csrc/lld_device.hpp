// Device helpers shared by the LLD kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "glibc_float.hpp"

namespace smilehip {

// R0: smilePcm_convertSamples, 16-bit mono (smileUtil.c:2527-2535):
// (float)s / 32767.0f, IEEE correctly rounded. Implemented as a reciprocal
// multiply plus one exact-residual correction (Markstein): q0 = RN(s*r),
// e = fma(-q0, 32767, s) (exact), q = RN(q0 + e*r). Verified against the
// correctly-rounded quotient for all 65536 inputs by tests/test_gpu_stages.py
// (and on the host by tests/test_host_logic.py with the same fmaf sequence).
__device__ __forceinline__ float pcm16_to_float(int16_t s) {
  const float a = (float)s;
  const float r = 1.0f / 32767.0f;           // constant-folded, correctly rounded
  const float q0 = a * r;
  const float e = fmaf(-q0, 32767.0f, a);
  return fmaf(e, r, q0);
}

// The chains' input: 16-bit mono PCM (the fused path's native format; R0 happens at the load) or float samples that
// smilehip_pcm_convert produced from any other sample format / channel count (R0 done: smilehip_lld_run_f32)
struct PcmIn {
  const int16_t *s;
  const float *f;
  __device__ __forceinline__ float operator[](int64_t n) const { return f ? f[n] : pcm16_to_float(s[n]); }
  __device__ __forceinline__ PcmIn operator+(int64_t o) const { PcmIn r; r.s = s + o; r.f = f ? f + o : nullptr; return r; }
};

// the same accessor for kernels instantiated for 16-bit input alone (no pointer test at the loads, two registers less)
// (and for 32-bit float input alone: the generic accessor's pointer test sits in front of EVERY load, and a load behind a branch waits alone)
struct PcmF32In {
  const float *f;
  __device__ __forceinline__ float operator[](int64_t n) const { return f[n]; }
  __device__ __forceinline__ PcmF32In operator+(int64_t o) const { PcmF32In r; r.f = f + o; return r; }
};
struct Pcm16In {
  const int16_t *s;
  __device__ __forceinline__ float operator[](int64_t n) const { return pcm16_to_float(s[n]); }
  __device__ __forceinline__ Pcm16In operator+(int64_t o) const { Pcm16In r; r.s = s + o; return r; }
};

// R6: one mel band in the reference's accumulation order
// (cMelspec::processVector, melspec.cpp:544-553): ascending bins, first the
// rising-slope run (p - p*w), then the falling-slope run (p*w). The reference
// forms p*w as (float)((double)p*(double)w): a 24x24-bit product is exact in
// double, so rounding it to float equals the float product.
__device__ __forceinline__ float mel_band_exact(const float *p, const float *coef, const int32_t *rng,
                                                int b, float scale) {
  const int rl = rng[4 * b + 0], rh = rng[4 * b + 1], fl = rng[4 * b + 2], fh = rng[4 * b + 3];
  float acc = 0.0f;
  for (int n = rl; n < rh; ++n) {
    const float a = p[n] * coef[n];
    acc += p[n] - a;
  }
  for (int n = fl; n < fh; ++n) {
    const float a = p[n] * coef[n];
    acc += a;
  }
  return acc * scale;
}

// The same band sum from precomputed terms: every lane of the group forms a[n] = p[n] * coef[n] and r[n] = p[n] - a[n] for its
// bins (mel_terms_fill), the band's lane then only adds -- four loads in flight per step instead of a load round trip per
// term (the sums of the widest bands are chains of ~60 dependent additions; with the loads inside the chain the 26 busy lanes
// kept a wave for a fifth of the ComParE frame kernel's time). Same products, same differences, same order: the same bits.
template <class G>
__device__ __forceinline__ void mel_terms_fill(const float *p, const float *coef, int K, float *a, float *r) {
  for (int n = G::tid(); n < K; n += G::size()) {
    const float pn = p[n], an = pn * coef[n];
    a[n] = an;
    r[n] = pn - an;
  }
}
__device__ __forceinline__ float mel_band_from_terms(const float *a, const float *r, const int32_t *rng, int b, float scale) {
  const int rl = rng[4 * b + 0], rh = rng[4 * b + 1], fl = rng[4 * b + 2], fh = rng[4 * b + 3];
  float acc = 0.0f;
  int n = rl;
  for (; n + 4 <= rh; n += 4) { const float t0 = r[n], t1 = r[n + 1], t2 = r[n + 2], t3 = r[n + 3]; acc += t0; acc += t1; acc += t2; acc += t3; }
  for (; n < rh; ++n) acc += r[n];
  n = fl;
  for (; n + 4 <= fh; n += 4) { const float t0 = a[n], t1 = a[n + 1], t2 = a[n + 2], t3 = a[n + 3]; acc += t0; acc += t1; acc += t2; acc += t3; }
  for (; n < fh; ++n) acc += a[n];
  return acc * scale;
}

// R4 core: in-place radix-2 DIT complex FFT of length M in LDS (re/im already
// loaded in bit-reversed order), executed by the whole workgroup.
__device__ __forceinline__ void block_cfft_radix2(float *re, float *im, int M, const float2 *tw_half) {
  for (int len = 2; len <= M; len <<= 1) {
    const int half = len >> 1;
    const int tstep = M / len;
    for (int b = threadIdx.x; b < (M >> 1); b += blockDim.x) {
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
    __syncthreads();
  }
}

// Real-FFT untangle of bin k (0 <= k <= M) from the half-length complex FFT Z
// of z[i] = x[2i] + i x[2i+1]:
//   X[k] = 1/2 [ (Z[k] + conj Z[M-k]) - i w^k (Z[k] - conj Z[M-k]) ],  w = e^{-2 pi i/(2M)}
// Returns the standard DFT X[k] = sum x[n] e^{-2 pi i nk/Nfft}.
__device__ __forceinline__ float2 untangle_bin(const float *re, const float *im, int M, int k,
                                               const float2 *tw_full) {
  if (k == 0) return make_float2(re[0] + im[0], 0.0f);
  if (k == M) return make_float2(re[0] - im[0], 0.0f);
  const float a = re[k], b = im[k], c = re[M - k], d = im[M - k];
  const float2 w = (k <= (M >> 1)) ? tw_full[k] : make_float2(-tw_full[M - k].x, tw_full[M - k].y);
  const float sr = a + c, si = b - d, dr = a - c, di = b + d;
  return make_float2(0.5f * fmaf(w.x, di, fmaf(w.y, dr, sr)), 0.5f * fmaf(w.y, di, fmaf(-w.x, dr, si)));
}

// R5: cFFTmagphase magnitude (fftmagphase.cpp:215-221). sqrtf() compiles to the
// correctly rounded sequence (v_sqrt_f32 + two FMA fix-ups); __fsqrt_rn is the
// bare 1-ulp instruction on gfx950 and must not be used where bit-parity counts.
__device__ __forceinline__ float bin_magnitude(float2 X, bool edge) {
  return edge ? fabsf(X.x) : sqrtf(X.x * X.x + X.y * X.y);
}
// sqrtf for an argument in [2^-96, infinity): v_sqrt_f32 and the library sequence's two residual tests, without its scaling of small
// arguments and its zero / infinity test (16 -> 9 instructions). Equal to sqrtf for every such float
// (tools/ubench/sqrt_f32_normal_check.hip: all 1.88e9 of them on the device).
__device__ __forceinline__ float sqrt_rn_normal(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  s = (0.0f >= rm) ? sm : s;
  s = (0.0f < rp) ? sp : s;
  return s;
}
// v[m] = sqrtf(v[m]) for a lane's batch: the lean form when every value of every lane of the wave lies in [2^-96, infinity), the
// library's otherwise (a zero, a tiny or a non-finite value anywhere: one wave-uniform test per batch). Callers put 1.0f where they
// have no value.
template <int N>
__device__ __forceinline__ void sqrt_rn_batch(float (&v)[N]) {
  bool odd = false;
#pragma unroll
  for (int m = 0; m < N; ++m) odd |= (__float_as_uint(v[m]) - 0x0f800000u) >= (0x7f800000u - 0x0f800000u);
  if (__builtin_amdgcn_ballot_w64(odd) != 0) {
#pragma unroll
    for (int m = 0; m < N; ++m) v[m] = sqrtf(v[m]);
  } else {
#pragma unroll
    for (int m = 0; m < N; ++m) v[m] = sqrt_rn_normal(v[m]);
  }
}

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

// Natural logarithm in double for the per-bin / per-band logarithms of the frame kernels (the device library's log is ~95
// instructions; the cepstrum instance of IS09 takes 257 of them per frame, ComParE's spectral entropy 256). Table + polynomial
// after the scheme of glibc's log.c (S. Nagy): x = 2^k z, z in [0.6875, 1.375) cut into 128 sub-intervals, z = c (1 + r) with
// |r| < 2^-7, log x = k ln2 + log c + log1p(r); the table (tools/gen_log_table.py) holds c with log c within 2^-64 of a
// double; log1p by its series to r^9 (truncation < 2^-73). Error <= 1 ulp, <= 1.5 where k ln2 + log c straddles a binade boundary
// (tests/test_gpu_fft.py measures it against long double) -- the class of the device library's own log, so "(float)log(double)" rounds like the reference's libm except for
// the same one-in-2^28 near-ties. Zero, negative, subnormal, infinite and NaN arguments go to the library function.
#include "log_table.inc"
__device__ const double2 kLogTab[128] = {SMILEHIP_LOG_TABLE};
// kNormalOrInf: the caller guarantees a positive normal number or +inf (e.g. 1 + a power); tab: the table (kLogTab, or a copy of it
// in LDS -- the table lookup is the one memory access of a logarithm, 257 of them per IS09 frame)
template <bool kNormalOrInf = false>
__device__ __forceinline__ double log_d(double x, const double2 *tab = kLogTab) {
  const unsigned long long ix = (unsigned long long)__double_as_longlong(x);
  if constexpr (kNormalOrInf) {
    if (ix == 0x7ff0000000000000ull) return x;           // log(+inf) = +inf
  } else {
    if (!(ix - 0x0010000000000000ull < 0x7fe0000000000000ull)) return log(x);
  }
  const unsigned long long tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127);
  const long long k = (long long)tmp >> 52;
  const double z = __longlong_as_double((long long)(ix - (tmp & (0xfffull << 52))));
  const double2 t = tab[i];
  const double kd = (double)k;
  const double r = fma(z, t.x, -1.0);
  const double w = kd * SMILEHIP_LN2HI + t.y;           // the product is exact (42-bit constant, |k| < 2^11)
  const double hi = w + r;
  const double lo = w - hi + r + kd * SMILEHIP_LN2LO;
  double q = 1.0 / 9.0;
  q = fma(q, r, -1.0 / 8.0); q = fma(q, r, 1.0 / 7.0); q = fma(q, r, -1.0 / 6.0); q = fma(q, r, 1.0 / 5.0);
  q = fma(q, r, -1.0 / 4.0); q = fma(q, r, 1.0 / 3.0); q = fma(q, r, -0.5);
  return lo + (r * r) * q + hi;
}

// R7: log floor (mfcc.cpp:239-243). The reference's logf (glibc) is correctly
// rounded in practice; the device logf (v_log_f32 based) is ~1 ulp. The
// reference-order kernels therefore take the double-precision log and round
// once; the fast kernel uses log_mel_fast.
__device__ __forceinline__ float log_mel(float v, float melfloor, float log_floor) {
  return (v < melfloor) ? log_floor : glibc_logf(v);     // mfcc.cpp:239-243: log() on a float is logf (glibc_float.hpp)
}
__device__ __forceinline__ float log_mel_fast(float v, float melfloor, float log_floor) {
#ifndef SMILEHIP_MFCC512_LIBRARY_LOGF
  // (round 6) v_log_f32 x ln 2: two instructions instead of the library logf's ~10 -- v >= melfloor > 0 is a normal number, the
  // library's denormal scaling and extended-precision product are dead weight in a kernel whose time is its instruction count
  // (868 -> 849 per pass). ~1.5 ulp of the logarithm instead of < 1; the distance to the reference did not move (per-frame-scaled
  // 1.0412e-6, max abs 1.4877e-4 before and after).
  return (v < melfloor) ? log_floor : __builtin_amdgcn_logf(v) * 0.693147182464599609375f;
#else
  return (v < melfloor) ? log_floor : logf(v);
#endif
}

// R7: one cepstral coefficient, sequential band order (mfcc.cpp:251-273)
__device__ __forceinline__ float dct_coeff(const float *lmel, const float *row_c, int n_bands, float gain) {
  float acc = 0.0f;
  int m = 0;
  for (; m + 4 <= n_bands; m += 4) {                       // (the four products' loads in flight together; the additions in band order)
    const float t0 = lmel[m] * row_c[m], t1 = lmel[m + 1] * row_c[m + 1], t2 = lmel[m + 2] * row_c[m + 2], t3 = lmel[m + 3] * row_c[m + 3];
    acc += t0; acc += t1; acc += t2; acc += t3;
  }
  for (; m < n_bands; ++m) acc += lmel[m] * row_c[m];
  return acc * gain;
}
// a float chain over n values in index order (cVectorOperation's vector sums and the like), four loads in flight per step
__device__ __forceinline__ float seq_sum_f32(const float *v, int n) {
  float d = 0.0f;
  int i = 0;
  for (; i + 4 <= n; i += 4) { const float t0 = v[i], t1 = v[i + 1], t2 = v[i + 2], t3 = v[i + 3]; d += t0; d += t1; d += t2; d += t3; }
  for (; i < n; ++i) d += v[i];
  return d;
}

// R8, PLP-CC branch of cPlp::processVector (plp.cpp:522-583, htkcompatible = 1, firstCC = 0):
// autocorrelation by the IDFT cosine table (double accumulate) for lag i ...
__device__ __forceinline__ float plp_acf_lag(const float *aud, const float *cosrow, int n_bands) {
  const int nFreq = n_bands + 2;
  double tmp = (double)cosrow[0] * (double)aud[0];
  int m;
  for (m = 1; m < nFreq - 1; m++) tmp += (double)cosrow[m] * (double)aud[m - 1];
  tmp += (double)cosrow[m] * (double)aud[nFreq - 3];
  return (float)(tmp / (2.0 * (nFreq - 1)));
}
// ... then Durbin (smileDsp_calcLpcAcf, smileUtil.c:1572-1630), LP -> cepstra (smileDsp_lpToCeps, :1532-1556) and the
// lifter, all in the reference's float sequence; p <= 15. out: c1..cp, c0 (HTK order).
// lp_only: stop behind the Durbin recursion and hand out the p LP coefficients (cPlp with doLP = 1, doLpToCeps = 0: plp.cpp:573-577)
__device__ __forceinline__ void plp_cc_serial(const float *r, int p, const float *sintable, float *out, bool lp_only = false) {
  float a[16], ceps[16];
  for (int i = 0; i < 16; ++i) { a[i] = 0.0f; ceps[i] = 0.0f; }
  float gain = 0.0f;
  if (!((r[0] == 0.0f) || (r[0] == -0.0f))) {
    float e = r[0];
    for (int m = 1; m <= p; m++) {
      float sum = (float)1.0 * r[m];
      for (int i = 1; i < m; i++) sum += a[i - 1] * r[m - i];
      const float k_m = ((float)-1.0 / e) * sum;
      a[m - 1] = k_m;
      for (int i = 1; i <= m / 2; i++) {
        const float x = a[i - 1];
        a[i - 1] += k_m * a[m - i - 1];
        if ((i < (m / 2)) || ((m & 1) == 1)) a[m - i - 1] += k_m * x;
      }
      e *= ((float)1.0 - k_m * k_m);
      if (e == 0.0f) { for (int i = m; i < p; i++) a[i] = 0.0f; break; }
    }
    gain = e;
  }
  if (lp_only) {
    for (int i = 0; i < p; i++) out[i] = a[i];
    return;
  }
  if (gain <= 0) gain = (float)1.0;
  for (int n = 1; n <= p; n++) {
    double sum = 0;
    for (int i = 1; i < n; i++) sum += (n - i) * a[i - 1] * ceps[n - i - 1];
    ceps[n - 1] = -(a[n - 1] + (float)(sum / (double)n));
  }
  ceps[p] = (float)(-log(1.0 / (double)gain));            // zeroth coefficient goes last
  for (int i = 0; i <= p; i++) {
    const int i1 = (i == p) ? 0 : i + 1;
    out[i] = ceps[i] * sintable[i1];
  }
}

}  // namespace smilehip
