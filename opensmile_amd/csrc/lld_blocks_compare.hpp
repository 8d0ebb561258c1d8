// Block-level (256 threads, Nfft = 512) building blocks of the ComParE descriptors, shared by the
// fused kernel (lld_compare.hip) and the per-component operators (lld_stage2_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "lld_params.hpp"

namespace smilehip {

// sums of NV doubles over the 256 threads of the block (4 waves); every thread gets the totals.
// red: 4*NV doubles of LDS. The order differs from the reference's sequential loops, the
// accumulator type (double) does not.
template <int NV>
__device__ __forceinline__ void block_sum_n(double (&v)[NV], double *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_down(v[i], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = ((red[i] + red[NV + i]) + red[2 * NV + i]) + red[3 * NV + i];
  __syncthreads();
}

// Two float sums in index order, one addition after the other -- what the reference's FLOAT_DMEM accumulators do (sharpness
// spectral.cpp:1435-1471, harmonicity :1485-1499, alpha ratio :997-1022). A float sum cannot be re-associated without changing
// its bits, so ONE thread walks the terms (the two chains interleave: two independent dependency chains); n a multiple of 4,
// a and b 16-byte aligned.
__device__ __forceinline__ void seq_sum2_f32(const float *a, const float *b, int n, float &s0, float &s1) {
  float x = 0.0f, y = 0.0f;
  float4 u = *reinterpret_cast<const float4 *>(a), v = *reinterpret_cast<const float4 *>(b);
  for (int i = 4; i <= n; i += 4) {
    float4 un = u, vn = v;                               // the next block's loads are in flight while this block's adds run
    if (i < n) { un = *reinterpret_cast<const float4 *>(a + i); vn = *reinterpret_cast<const float4 *>(b + i); }
    x += u.x; y += v.x;
    x += u.y; y += v.y;
    x += u.z; y += v.z;
    x += u.w; y += v.w;
    u = un; v = vn;
  }
  s0 = x; s1 = y;
}

// sum of x[lo .. hi) in index order, one float addition after the other; the loads run a few elements ahead of the adds
__device__ __forceinline__ float seq_sum_f32(const float *x, int lo, int hi) {
  float s = 0.0f;
  int i = lo;
  for (; i < hi && (i & 3); ++i) s += x[i];
  if (i + 4 <= hi) {
    float4 u = *reinterpret_cast<const float4 *>(x + i);
    for (i += 4; i + 4 <= hi; i += 4) {
      const float4 un = *reinterpret_cast<const float4 *>(x + i);
      s += u.x; s += u.y; s += u.z; s += u.w;
      u = un;
    }
    s += u.x; s += u.y; s += u.z; s += u.w;
  }
  for (; i < hi; ++i) s += x[i];
  return s;
}

// R11 cSpectral::processVector with ComParE_2016's option set ([is13_spectral]: bands 250-650 and
// 1000-4000, roll-off .25/.5/.75/.9, flux, centroid, entropy, variance, skewness, kurtosis, slope,
// sharpness, harmonicity; squareInput = 1, freqRange 0-0, oldSlopeScale = 1; spectral.cpp:586-1560).
// Thread i of 256 owns bin j = i+1. mg / pw: magnitudes and powers of the K = 257 bins, prev: the
// previous frame's magnitudes (ignored when first). red: 64 doubles, cum: 256 doubles, pk_val[4],
// pk_has[4] of LDS scratch. Writes sp[0..14]; thread 0 returns valid frameSum-based values; ends
// with a block barrier. The sums are double like the reference's, combined in a fixed tree order.
__device__ __forceinline__ void spectral_frame(const float *mg, const float *pw, const float *prev, bool first,
                                               const SpectralConsts &C, int K, double *red, double *cum, float *pk_val,
                                               int *pk_has, float *sp) {
  const double F0 = 1.0 / C.fsSec;
  const int lo = 1, hi = K - 1, nBins = K - 1;          // freqRange 0-0 (spectral.cpp:625-627)
  const int tid = threadIdx.x, j = tid + 1;
  const int lane = tid & 63, wave = tid >> 6;
  const float pf = pw[j];
  const double p = (double)pf, fj = F0 * j;
  double v1[6];
  v1[0] = p;                                            // frame energy (:762-767), centroid denominator
  v1[1] = fj * p;                                       // centroid numerator (:1256-1330)
  v1[2] = 0.0;                                          // (sharpness: a float chain, below)
  { const double myB = (double)mg[j] - (double)prev[j]; v1[3] = first ? 0.0 : myB * myB; }   // flux (:1124-1254)
#pragma unroll
  for (int b = 0; b < 2; ++b) {                         // band energies (:779-853), edges resolved on the host
    auto part = [&](int k) {
      const double pk = (double)pw[k];
      double c = 0.0;
      if (k == C.band_iL[b]) c += pk * C.band_wL[b];
      if (k > C.band_iL[b] && k < C.band_iR[b]) c += pk;
      if (k == C.band_iR[b]) c += pk * C.band_wR[b];
      return c;
    };
    v1[4 + b] = part(j) + (tid == 0 ? part(0) : 0.0);
  }
  block_sum_n<6>(v1, red);
  const double frameSum = v1[0], sumA = v1[1];
  float ctr = 0.0f;
  if (frameSum != 0.0) ctr = (float)(sumA / frameSum);
  // roll-off (:1102-1122): inclusive prefix of the power in double, first bin whose prefix reaches the share
  {
    double c = p;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(c, off, 64); if (lane >= off) c += o; }
    if (lane == 63) red[wave] = c;
    __syncthreads();
    for (int w = 0; w < wave; ++w) c += red[w];
    cum[tid] = c;
    __syncthreads();
    const double before = tid ? cum[tid - 1] : -1.0;
    const double rollOff[4] = {0.25, 0.50, 0.75, 0.90};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double th = rollOff[i] * frameSum;
      if (c >= th && (tid == 0 || !(before >= th))) sp[2 + i] = (float)(F0 * j);
    }
  }
  // harmonicity (:1484-1513): alternating peaks/valleys, distance to the previous one
  float hc = 0.0f;
  {
    bool flag = false;
    if (j >= lo + 2 && j < hi - 1)
      flag = (pw[j - 2] < pf && pw[j - 1] < pf && pf > pw[j + 1] && pf > pw[j + 2]) ||
             (pw[j - 2] > pf && pw[j - 1] > pf && pf < pw[j + 1] && pf < pw[j + 2]);
    const unsigned long long mask = __ballot(flag);
    const unsigned long long lower = mask & ((1ull << lane) - 1ull);
    const int src = lower ? 63 - __clzll(lower) : 0;
    const float prevw = __shfl(pf, src, 64);
    if (mask && lane == 63 - __clzll(mask)) pk_val[wave] = pf;
    if (lane == 0) pk_has[wave] = mask != 0ull;
    __syncthreads();
    if (flag) {
      if (lower) hc = fabsf(pf - prevw);
      else
        for (int w = wave - 1; w >= 0; --w)
          if (pk_has[w]) { hc = fabsf(pf - pk_val[w]); break; }
    }
  }
  double v2[5];
  {                                                     // entropy (smileStat_entropy, smileUtil.c:2079-2124; powers: min = 0)
    const double entropy_floor = 0.0000001;
    double dn = frameSum;
    if (dn < (float)entropy_floor) dn = (float)entropy_floor;
    double v = p;
    if (v <= entropy_floor) v = entropy_floor;
    const double ln = v / dn;
    v2[0] = (ln > 0.0) ? ln * log_d(ln) / log(2.0) : 0.0;
    const double t1 = fj - (double)ctr;                 // moments (:1338-1397)
    double m = t1 * t1 * p;
    v2[1] = m; m *= t1; v2[2] = m; v2[3] = m * t1;
    v2[4] = 0.0;                                        // (harmonicity: a float chain, below)
  }
  // sharpness and harmonicity accumulate in FLOAT_DMEM, bin after bin: the terms go to LDS (cum is free after the roll-off:
  // the barrier of the harmonicity section lies behind its last read), thread 0 adds them in order
  float *chain = reinterpret_cast<float *>(cum);
  chain[tid] = (float)(C.sharp_w[tid] * p);             // (FLOAT_DMEM)(sharpnessWeights[j - lo] * (double)srcP[j]), :1455 / :1469
  chain[256 + tid] = hc;                                // |srcLP[j] - lastPeak| of a flagged bin, +0 elsewhere (s + 0 = s), :1493
  block_sum_n<5>(v2, red);
  if (tid == 0) {
    float sumAA_seq, ptp_seq;
    seq_sum2_f32(chain, chain + 256, 256, sumAA_seq, ptp_seq);
    sp[0] = (float)(v1[4] / (double)nBins);
    sp[1] = (float)(v1[5] / (double)nBins);
    float c2 = 0.0f;
    const float sumAA = sumAA_seq;
    if (frameSum != 0.0) c2 = (float)(sumAA / frameSum);
    sp[13] = (float)(0.11 * c2);
    const double flux = v1[3] / (double)nBins;
    sp[6] = (!first && flux > 0.0) ? (float)sqrt(flux) : 0.0f;
    sp[7] = ctr;
    sp[8] = (float)(-v2[0]);
    const double sumB = frameSum;
    const double sigma2 = (sumB != 0.0) ? v2[1] / sumB : 0.0;
    sp[9] = (float)sigma2;
    sp[10] = (sigma2 <= 0.0) ? 0.0f : (float)(v2[2] / (sumB * sigma2 * sqrt(sigma2)));
    sp[11] = (sigma2 == 0.0) ? 0.0f : (float)(v2[3] / (sumB * sigma2 * sigma2));
    const double Nind = (double)nBins;
    const double deno = (Nind * C.slope_S2f - C.slope_Sf * C.slope_Sf);
    double slope = 0.0;
    if (deno != 0.0) slope = (Nind * sumA - C.slope_Sf * sumB) / deno;
    sp[12] = (float)(slope * (Nind - 1.0));              // oldSlopeScale = 1
    float ptpSum = ptp_seq;
    ptpSum /= 2.0f;
    ptpSum /= (float)nBins;
    sp[14] = ptpSum;
  }
  __syncthreads();
}

// ---- the same sums with ONE wave per frame: lane l plays the four threads l, l+64, l+128, l+192 of the block version
// (index w = the block version's wave), so every reduction keeps the block version's tree: shuffle-down inside each group
// of 64, then ((s0 + s1) + s2) + s3 -- the results are bit-identical to block_sum_n / spectral_frame.
template <int NV>
__device__ __forceinline__ void wave_sum4(const double (&v)[4][NV], double (&tot)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double t = ((v[0][i] + v[1][i]) + v[2][i]) + v[3][i];      // (the lane's four terms first, then one tree: see wave_sumw)
    tot[i] = wave_first_d(wave_tree_d(t, [](double a, double b) { return a + b; }));
  }
}

// spectral_frame for one wave, K = 64 W + 1 bins (W = 4: the 512-point spectrum of 20 ms frames at 16 kHz, the tuned case; W = 2 / 8:
// the 256- / 1024-point spectra of other sample rates): no LDS scratch, no barriers. mg / pw / prev as above; sp[0..14] written
// by lane 0 (roll-off points by the lane that owns the crossing bin).
// chain: 128 W floats of LDS scratch (16-byte aligned) for the two float chains.
// (The sums are double accumulators of float-derived terms whose results are rounded to float: the reference adds bin after bin, any
// other order differs from it by a few ulps of the DOUBLE and gives the same float except with probability ~1e-9 per value. Round 3:
// the lane's W terms are added first and ONE wave tree follows per quantity -- a quarter of the reductions of the block form's
// per-group trees, which cost ~1200 of the kernel's ~4000 instruction slots per frame.)
template <int NV, int W>
__device__ __forceinline__ void wave_sumw(const double (&v)[W][NV], double (&tot)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double t = v[0][i];
#pragma unroll
    for (int w = 1; w < W; ++w) t += v[w][i];
    tot[i] = wave_first_d(wave_tree_d(t, [](double a, double b) { return a + b; }));   // the shuffle-down tree, lld_blocks.hpp
  }
}
template <int W>
__device__ __forceinline__ void spectral_frame_wave(const float *mg, const float *pw, const float *prev, bool first,
                                                    const SpectralConsts &C, int K, float *chain, float *sp) {
  const double F0 = 1.0 / C.fsSec;
  const int lo = 1, hi = K - 1, nBins = K - 1;
  const int lane = threadIdx.x & 63;
  float pf[W], tsh[W];
  double p[W], fj[W], v1[W][6];
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const int tid = lane + 64 * w, j = tid + 1;
    pf[w] = pw[j];
    p[w] = (double)pf[w];
    fj[w] = F0 * j;
    v1[w][0] = p[w];
    v1[w][1] = fj[w] * p[w];
    v1[w][2] = 0.0;
    tsh[w] = (float)(C.sharp_w[tid] * p[w]);            // sharpness term, :1455 / :1469
    { const double myB = (double)mg[j] - (double)prev[j]; v1[w][3] = first ? 0.0 : myB * myB; }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      auto part = [&](int k) {
        const double pk = (double)pw[k];
        double c = 0.0;
        if (k == C.band_iL[b]) c += pk * C.band_wL[b];
        if (k > C.band_iL[b] && k < C.band_iR[b]) c += pk;
        if (k == C.band_iR[b]) c += pk * C.band_wR[b];
        return c;
      };
      v1[w][4 + b] = part(j) + (tid == 0 ? part(0) : 0.0);
    }
  }
  double t1v[6];
  wave_sumw<6, W>(v1, t1v);
  const double frameSum = t1v[0], sumA = t1v[1];
  float ctr = 0.0f;
  if (frameSum != 0.0) ctr = (float)(sumA / frameSum);
  // roll-off: inclusive prefix in the block version's order (scan inside each group of 64, then + the earlier groups' totals)
  {
    double c[W], red[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      double x = p[w];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
      red[w] = __shfl(x, 63, 64);
      c[w] = x;
    }
#pragma unroll
    for (int w = 1; w < W; ++w)
#pragma unroll
      for (int w2 = 0; w2 < w; ++w2) c[w] += red[w2];
    const double rollOff[4] = {0.25, 0.50, 0.75, 0.90};
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int tid = lane + 64 * w, j = tid + 1;
      const double up1 = __shfl_up(c[w], 1, 64);
      const double prev63 = __shfl(c[w > 0 ? w - 1 : 0], 63, 64);
      const double before = (lane == 0) ? (w == 0 ? -1.0 : prev63) : up1;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const double th = rollOff[i] * frameSum;
        if (c[w] >= th && (tid == 0 || !(before >= th))) sp[2 + i] = (float)(F0 * j);
      }
    }
  }
  // harmonicity: alternating peaks/valleys, distance to the previous one
  float hc[W];
  {
    float pk_val[W];
    bool pk_has[W];
    bool flag[W];
    unsigned long long lower[W];
    float prevw[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int j = lane + 64 * w + 1;
      flag[w] = false;
      if (j >= lo + 2 && j < hi - 1)
        flag[w] = (pw[j - 2] < pf[w] && pw[j - 1] < pf[w] && pf[w] > pw[j + 1] && pf[w] > pw[j + 2]) ||
                  (pw[j - 2] > pf[w] && pw[j - 1] > pf[w] && pf[w] < pw[j + 1] && pf[w] < pw[j + 2]);
      const unsigned long long mask = __ballot(flag[w]);
      lower[w] = mask & ((1ull << lane) - 1ull);
      const int src = lower[w] ? 63 - __clzll((long long)lower[w]) : 0;
      prevw[w] = __shfl(pf[w], src, 64);
      pk_has[w] = mask != 0ull;
      pk_val[w] = __shfl(pf[w], mask ? 63 - __clzll((long long)mask) : 0, 64);
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
      hc[w] = 0.0f;
      if (flag[w]) {
        if (lower[w]) hc[w] = fabsf(pf[w] - prevw[w]);
        else {
          bool found = false;
#pragma unroll
          for (int w2 = W - 1; w2 >= 0; --w2)
            if (w2 < w && !found && pk_has[w2]) { hc[w] = fabsf(pf[w] - pk_val[w2]); found = true; }
        }
      }
    }
  }
  double v2[W][5];
#pragma unroll
  for (int w = 0; w < W; ++w) {
    const double entropy_floor = 0.0000001;
    double dn = frameSum;
    if (dn < (float)entropy_floor) dn = (float)entropy_floor;
    double v = p[w];
    if (v <= entropy_floor) v = entropy_floor;
    const double ln = v / dn;
    // (ln > 0: a quotient of two values >= 1e-7 that are floats or sums of floats -- a positive normal double, or +inf)
    v2[w][0] = (ln > 0.0) ? ln * log_d<true>(ln, C.log_tab ? static_cast<const double2 *>(C.log_tab) : kLogTab) / log(2.0) : 0.0;
    const double t1 = fj[w] - (double)ctr;
    double m = t1 * t1 * p[w];
    v2[w][1] = m; m *= t1; v2[w][2] = m; v2[w][3] = m * t1;
    v2[w][4] = 0.0;
  }
  double t2v[5];
  wave_sumw<5, W>(v2, t2v);
  // sharpness and harmonicity accumulate in FLOAT_DMEM, bin after bin (:1435-1471, :1485-1499): the terms go to LDS, lane 0
  // adds them in order. (Measured alternative: every lane walking the chains with 512 unrolled v_readlane + v_add was 17 %
  // slower for the whole kernel -- issue slots and registers -- than one lane reading float4s one block ahead.)
#pragma unroll
  for (int w = 0; w < W; ++w) {
    chain[lane + 64 * w] = tsh[w];
    chain[64 * W + lane + 64 * w] = hc[w];                 // |srcLP[j] - lastPeak| of a flagged bin, +0 elsewhere (s + 0 = s)
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    float sumAA_seq, ptp_seq;
    seq_sum2_f32(chain, chain + 64 * W, 64 * W, sumAA_seq, ptp_seq);
    sp[0] = (float)(t1v[4] / (double)nBins);
    sp[1] = (float)(t1v[5] / (double)nBins);
    float c2 = 0.0f;
    const float sumAA = sumAA_seq;
    if (frameSum != 0.0) c2 = (float)(sumAA / frameSum);
    sp[13] = (float)(0.11 * c2);
    const double flux = t1v[3] / (double)nBins;
    sp[6] = (!first && flux > 0.0) ? (float)sqrt(flux) : 0.0f;
    sp[7] = ctr;
    sp[8] = (float)(-t2v[0]);
    const double sumB = frameSum;
    const double sigma2 = (sumB != 0.0) ? t2v[1] / sumB : 0.0;
    sp[9] = (float)sigma2;
    sp[10] = (sigma2 <= 0.0) ? 0.0f : (float)(t2v[2] / (sumB * sigma2 * sqrt(sigma2)));
    sp[11] = (sigma2 == 0.0) ? 0.0f : (float)(t2v[3] / (sumB * sigma2 * sigma2));
    const double Nind = (double)nBins;
    const double deno = (Nind * C.slope_S2f - C.slope_Sf * C.slope_Sf);
    double slope = 0.0;
    if (deno != 0.0) slope = (Nind * sumA - C.slope_Sf * sumB) / deno;
    sp[12] = (float)(slope * (Nind - 1.0));
    float ptpSum = ptp_seq;
    ptpSum /= 2.0f;
    ptpSum /= (float)nBins;
    sp[14] = ptpSum;
  }
}

// R8 cPlp as auditory spectrum, one band (plp.cpp:416-593 with doAud = 1, doIDFT = doLP = 0).
// Without RASTA: melfloor, x equal loudness, power-law compression through double pow (:499-507).
__device__ __forceinline__ float plp_aud_band(float mel, float melfloor, float eql, float compression) {
  float v = mel < melfloor ? melfloor : mel;
  v *= eql;
  return (float)pow((double)v, (double)compression);
}
// newRASTA in the log domain (:434-439 doLog, :468-485 filter, :490-497 log equal loudness and compression,
// :512-517 exp): x = log of the floored band, st = the band's 4 filter taps, init = frames seen (capped at 5)
__device__ __forceinline__ float plp_rasta_band(float x, float (&st)[4], int init, const float *fir, float iir, float eql_log,
                                                float compression) {
  const float out = fir[0] * x + st[0];
  st[0] = fir[1] * x + st[1] + (float)(init >= 5) * iir * out;
  st[1] = fir[2] * x + st[2];
  st[2] = fir[3] * x + st[3];
  st[3] = fir[4] * x;
  x = (init >= 5) ? out : 0.0f;
  x += eql_log;
  x *= compression;
  return glibc_expf(x);                                 // plp.cpp:512-517: exp() on a float is expf
}

}  // namespace smilehip
