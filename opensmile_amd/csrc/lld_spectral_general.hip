// R11, the whole option space of the shipped configuration files: cSpectral::processVector (src/lldcore/spectral.cpp:586-1560)
// with any number of bands[] (<= 16), slopes[] (<= 16) and rollOff[] points (<= 16) and specDiff, specPosDiff, flux, fluxCentroid,
// fluxAtFluxCentroid, centroid, maxPos, minPos, entropy, standardDeviation, variance, skewness, kurtosis, slope, sharpness, harmonicity,
// flatness (or its logarithm) each optional, in the reference's output order; squareInput = 1, a linear magnitude
// spectrum with the frequency axis cTransformFFT attaches (frq[i] = i / frameSizeSec), freqRange 0-0, normBandEnergies = 0,
// useLogSpectrum = 0, buggyRollOff = 0, oldSlopeScale = 1 -- what avec2011 / avec2013, emo_large and the MediaEval files ask for
// (ComParE_2016's and GeMAPS' sets have their own wave-parallel kernels, lld_blocks_compare.hpp / lld_gemaps.hip).
// One THREAD per frame: every accumulator is the reference's own sequential chain (double or FLOAT_DMEM as there), the frames of
// a launch run side by side. The spectrum is walked twice: sums that need nothing but the bins (frame energy, centroid
// numerator, bands, flux, extremes), then the ones that need those (roll-off points, entropy, moments, sharpness, harmonicity);
// an accumulator's order never depends on which loop it sits in.
#include <hip/hip_runtime.h>

#include "lld_device.hpp"
#include "lld_stage.hpp"

namespace smilehip {

__global__ void __launch_bounds__(64) lld_spectral_general(SpectralGeneral G, const float *mag, int64_t ld_src, const float *state, int first,
                                                          float *dst, int64_t ld_dst, int64_t n_frames) {
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= n_frames) return;
  const float *src = mag + t * ld_src;
  const float *prev = t > 0 ? mag + (t - 1) * ld_src : state;    // the frame before (the flux): the launch's, or the stream's
  const bool have_prev = t > 0 || !first;
  float *o = dst + t * ld_dst;
  const int Nsrc = G.K;
  const int lo = 1, hi = Nsrc - 1;                               // specRange 0-0 => bins 1 .. Nsrc-1 (:625-627)
  const int nBins = hi - lo + 1;
  const double F0 = 1.0 / G.frame_size_sec;                      // frq[i] = F0 * i (transformFft.cpp:102-117)
  // ---- first walk
  double frameSum = 0.0, sumA = 0.0, fluxA = 0.0;                // :762-767 (== sumB, :1093-1097, == the entropy's dn), :1262-1266, :1196-1203
  double fluxAf = 0.0, sdiff = 0.0, spdiff = 0.0;                // :1177-1190 (sum of myB^2 frq), :1143-1168 (specDiff / specPosDiff)
  const bool flux_family = G.spec_pos_diff || G.spec_diff || G.flux || G.flux_centroid || G.flux_at_flux_centroid;
  double band[16];
  for (int b = 0; b < 16; ++b) band[b] = 0.0;
  double sl_Sf[16], sl_S2f[16], sl_A[16], sl_B[16];              // slopes[]: :944-962
  for (int b = 0; b < 16; ++b) { sl_Sf[b] = 0.0; sl_S2f[b] = 0.0; sl_A[b] = 0.0; sl_B[b] = 0.0; }
  int maP = lo, miP = lo;
  float vmax = 0.0f, vmin = 0.0f;
  for (int j = 0; j < Nsrc; ++j) {
    const float m = src[j];
    const float p = m * m;                                       // :676-683
    for (int b = 0; b < G.n_bands; ++b) {                        // :832-836: the left edge bin weighted, the bins between, the right edge bin weighted
      if (j == G.band_iL[b]) band[b] = (double)p * G.band_wL[b];
      else if (j > G.band_iL[b] && j < G.band_iR[b]) band[b] += (double)p;
      if (j == G.band_iR[b]) band[b] += (double)p * G.band_wR[b];
    }
    for (int b = 0; b < G.n_slopes; ++b) {                       // :944-962: the same three-part walk, four sums
      const double fj = F0 * (double)j;
      if (j == G.sl_iL[b]) {
        sl_Sf[b] = fj * G.sl_wL[b];
        sl_S2f[b] = sl_Sf[b] * sl_Sf[b];
        sl_A[b] = fj * G.sl_wL[b] * (double)p;
        sl_B[b] = G.sl_wL[b] * (double)p;
      } else if (j > G.sl_iL[b] && j < G.sl_iR[b]) {
        sl_S2f[b] += fj * fj;
        sl_Sf[b] += fj;
        sl_A[b] += fj * (double)p;
        sl_B[b] += (double)p;
      }
      if (j == G.sl_iR[b]) {
        sl_S2f[b] += fj * G.sl_wR[b] * fj * G.sl_wR[b];
        sl_Sf[b] += fj * G.sl_wR[b];
        sl_A[b] += fj * G.sl_wR[b] * (double)p;
        sl_B[b] += G.sl_wR[b] * (double)p;
      }
    }
    if (j >= lo) {
      frameSum += p;
      sumA += (F0 * (double)j) * (double)p;
      if (have_prev && flux_family) {
        const double d = ((double)m / 1.0 - (double)prev[j] / 1.0);
        if (G.flux || G.flux_centroid) fluxA += d * d;
        if (G.flux_centroid) fluxAf += d * d * (F0 * (double)j);
        if (G.spec_diff || G.spec_pos_diff) {                    // (the reference subtracts the two FLOAT_DMEM values as floats here)
          const double myd = (double)(m - prev[j]);
          if (G.spec_diff) sdiff += myd * myd;
          if (G.spec_pos_diff && myd > 0.0) spdiff += myd * myd;
        }
      }
      if (j == lo) { vmax = p; vmin = p; }                       // :1314-1330 (the last bin is not looked at)
      else if (j < hi) {
        if (p < vmin) { vmin = p; miP = j; }
        if (p > vmax) { vmax = p; maP = j; }
      }
    }
  }
  int n = 0;
  for (int b = 0; b < G.n_bands; ++b) o[n++] = (float)(band[b] / (double)nBins);     // :853 (normBandEnergies = 0)
  for (int b = 0; b < G.n_slopes; ++b) {                         // :963-980 (oldSlopeScale = 1)
    const double Nind = G.sl_Nind[b];
    const double deno = (Nind * sl_S2f[b] - sl_Sf[b] * sl_Sf[b]);
    double slope = 0.0;
    if (deno != 0.0) slope = (Nind * sl_A[b] - sl_Sf[b] * sl_B[b]) / deno;
    o[n++] = (float)(slope * (Nind - 1.0));
  }
  const double sumB = frameSum;
  float ctr = 0.0f;
  const bool need_ctr = G.centroid || G.standard_deviation || G.variance || G.skewness || G.kurtosis || G.slope;
  if (need_ctr && sumB != 0.0) ctr = (float)(sumA / sumB);       // :1256-1311
  // ---- second walk
  float ro[16];
  for (int i = 0; i < 16; ++i) ro[i] = 0.0f;
  double sumC = 0.0, ent = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0;
  float sumAA = 0.0f, ptpSum = 0.0f, lastPeak = -99.0f, gmean = 0.0f;
  int nGm = 0;
  const double entropy_floor = 0.0000001;
  double dn = frameSum;                                          // smileStat_entropy (smileUtil.c:2079-2124) on powers: min = 0
  if (dn < (float)entropy_floor) dn = (float)entropy_floor;
  const double l2 = log(2.0);
  const double u = ctr;
  float w0 = 0.0f, w1 = 0.0f, w2 = 0.0f, w3 = 0.0f;             // the powers of bins j-2 .. j+1 (harmonicity's window)
  w2 = src[lo] * src[lo];
  w3 = (lo + 1 < Nsrc) ? src[lo + 1] * src[lo + 1] : 0.0f;
  for (int j = lo; j <= hi; ++j) {
    const float p = w2;
    const float w4 = (j + 2 < Nsrc) ? src[j + 2] * src[j + 2] : 0.0f;
    sumC += (double)p;                                           // :1102-1117
    for (int i = 0; i < G.n_rolloff; ++i)
      if ((ro[i] == 0.0f) && (sumC >= G.rolloff[i] * frameSum)) ro[i] = (float)(F0 * (double)j);
    if (G.entropy) {
      double v = p;
      if (v <= entropy_floor) v = entropy_floor;
      const double ln = v / dn;
      if (ln > 0.0) ent += ln * log_d(ln) / l2;
    }
    if (G.standard_deviation || G.variance || G.skewness || G.kurtosis) {   // :1338-1397
      const double t1 = (F0 * (double)j - u);
      double m = t1 * t1 * (double)p;
      m2 += m; m *= t1; m3 += m; m4 += m * t1;
    }
    if (G.sharpness) sumAA += (float)(G.sharp_w[j - lo] * (double)p);                 // :1455 / :1469
    if (G.harmonicity && j >= lo + 2 && j < hi - 1) {            // :1484-1513
      if ((w0 < p && w1 < p && p > w3 && p > w4) || (w0 > p && w1 > p && p < w3 && p < w4)) {
        if (lastPeak != -99.0f) ptpSum += fabsf(p - lastPeak);
        lastPeak = p;
      }
    }
    if (G.flatness && sumB != 0.0 && p != 0.0f) { gmean += glibc_logf(fabsf(p)); nGm++; }     // :1521-1526: log() on a FLOAT_DMEM is logf
    w0 = w1; w1 = w2; w2 = w3; w3 = w4;
  }
  for (int i = 0; i < G.n_rolloff; ++i) o[n++] = ro[i];
  if (flux_family) {                                             // :1124-1254
    if (!have_prev) o[n++] = 0.0f;                               // a field's first frame: ONE zero for the whole family (:1136)
    else {
      const double nR = (double)(hi - lo + 1);
      if (G.spec_diff) { const double d = sdiff / nR; o[n++] = (d > 0.0) ? (float)sqrt(d) : 0.0f; }
      if (G.spec_pos_diff) { const double d = spdiff / nR; o[n++] = (d > 0.0) ? (float)sqrt(d) : 0.0f; }
      if (G.flux) {
        const double flux = (nBins > 0) ? fluxA / (double)nBins : 0.0;
        o[n++] = (flux > 0.0) ? (float)sqrt(flux) : 0.0f;
      }
      if (G.flux_centroid || G.flux_at_flux_centroid) {
        const double fluxCentr = (fluxA > 0.0) ? fluxAf / fluxA : 0.0;
        if (G.flux_centroid) o[n++] = (float)fluxCentr;
        if (G.flux_at_flux_centroid) {                           // :1209-1247: the flux of the five bins around the centroid's bin
          int bin = hi;
          for (int j = lo; j <= hi; ++j) if (F0 * (double)j >= fluxCentr) { bin = j; break; }
          int start = bin - 2, end = bin + 2;
          if (start < lo) start = lo;
          if (end > hi) end = hi;
          double myF = 0.0;
          for (int j = start; j <= end; ++j) { const double d = ((double)src[j] / 1.0 - (double)prev[j] / 1.0); myF += d * d; }
          if (end - start + 1 > 0) myF /= (double)(end - start + 1); else myF = 0.0;
          o[n++] = (float)myF;
        }
      }
    }
  }
  if (G.centroid) o[n++] = ctr;
  if (G.max_pos) o[n++] = (float)(F0 * (double)maP);
  if (G.min_pos) o[n++] = (float)(F0 * (double)miP);
  if (G.entropy) o[n++] = (float)(-ent);
  if (G.standard_deviation || G.variance || G.skewness || G.kurtosis) {
    const double sigma2 = (sumB != 0.0) ? m2 / sumB : 0.0;
    if (G.standard_deviation) o[n++] = (sigma2 > 0.0) ? (float)sqrt(sigma2) : 0.0f;
    if (G.variance) o[n++] = (float)sigma2;
    if (G.skewness) o[n++] = (sigma2 <= 0.0) ? 0.0f : (float)(m3 / (sumB * sigma2 * sqrt(sigma2)));
    if (G.kurtosis) o[n++] = (sigma2 == 0.0) ? 0.0f : (float)(m4 / (sumB * sigma2 * sigma2));
  }
  if (G.slope) {                                                 // :1399-1427 (oldSlopeScale = 1)
    const double Nind = (double)nBins;
    const double deno = (Nind * G.slope_S2f - G.slope_Sf * G.slope_Sf);
    double slope = 0.0;
    if (deno != 0.0) slope = (Nind * sumA - G.slope_Sf * sumB) / deno;
    o[n++] = (float)(slope * (Nind - 1.0));
  }
  if (G.sharpness) {
    float c2 = 0.0f;
    if (frameSum != 0.0) c2 = (float)(sumAA / frameSum);
    o[n++] = (float)(0.11 * c2);
  }
  if (G.harmonicity) {
    ptpSum /= 2.0;
    ptpSum /= (float)nBins;
    o[n++] = ptpSum;
  }
  if (G.flatness) {                                              // :1515-1545: the geometric mean over the arithmetic one
    float sf = 0.0f;
    if (sumB != 0.0) {
      if (nGm > 0) gmean /= (float)nGm;
      gmean = glibc_expf(gmean);
      sf = gmean / (float)fabs(sumB / (double)nBins);
    }
    o[n++] = G.log_flatness ? ((sf > 0.0f) ? glibc_logf(sf) : 0.0f) : sf;
  }
  while (n < G.n_out) o[n++] = 0.0f;                             // (a field's first frame with more than one of the flux family on: the
                                                                 // reference's vector keeps its calloc'd zeros in the last slots)
}

// the last frame's magnitudes become the stream's state (the next launch's flux)
__global__ void lld_spectral_keep(const float *row, float *state, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K) state[i] = row[i];
}

hipError_t stage_spectral_general(const SpectralGeneral &G, const float *mag, int64_t ld_src, float *state, int first, float *dst,
                                  int64_t ld_dst, int64_t n_frames, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  hipLaunchKernelGGL(lld_spectral_general, dim3((unsigned)((n_frames + 63) / 64)), dim3(64), 0, s, G, mag, ld_src, state, first, dst, ld_dst,
                     n_frames);
  if ((G.flux || G.spec_diff || G.spec_pos_diff || G.flux_centroid || G.flux_at_flux_centroid) && state)
    hipLaunchKernelGGL(lld_spectral_keep, dim3((unsigned)((G.K + 255) / 256)), dim3(256), 0, s, mag + (n_frames - 1) * ld_src, state, G.K);
  return hipGetLastError();
}

}  // namespace smilehip
