/*
 * lld_oracle_gemaps.c -- CPU ORACLE, part 6. TEST INFRASTRUCTURE ONLY (see lld_oracle.h).
 *
 * BASELINE.json config 5: config/egemaps/v02/eGeMAPSv02.conf (which includes config/gemaps/v01b/GeMAPSv01b_core.*.conf.inc),
 * the 25-column LLD level and the 88 functionals. Restated here, every function citing the reference lines it follows:
 *   cSpectral with the GeMAPS option sets (src/lldcore/spectral.cpp:586-1254): log-spectrum band slopes, alpha ratio,
 *     Hammarberg index, spectral flux over 0-5000 Hz
 *   cEnergy energy2 (src/lldcore/energy.cpp:152-170)
 *   cSpecResample -> cLpc -> cFormantLpc (src/dsp/specResample.cpp:117-185, smileUtil.c:1752-1820 (slow inverse DFT),
 *     src/lld/lpc.cpp:154-166 + smileUtil.c:1560-1630, src/lld/formantLpc.cpp:192-290, src/smileutil/zerosolve.cpp,
 *     smileUtil.c:992-1003, 2019-2053)
 *   cHarmonics (src/lld/harmonics.cpp:369-1031): ACF harmonics-to-noise ratio, H1-H2, H1-A3, formant amplitudes
 *   cPitchSmootherViterbi with bufferLength 40 and F0finalLog (src/lld/pitchSmootherViterbi.cpp:451-570)
 *   cPitchJitter with GeMAPS' option set (jitterLocal, shimmerLocalDB; src/lld/pitchJitter.cpp:591-1064)
 *   cValbasedSelector gates, cDataSelector column picks, cContourSmoother (with and without noZeroSma) and the
 *   end-of-input behaviour of the tick loop for this graph.
 * Everything the reference does in double is done in double, in the reference's order; `log(x)` of a float argument is
 * the float overload (the reference is C++ including <math.h>: logf), of a double argument the double one.
 *
 * Pinning (tests/test_oracle_pin_gemaps.py): with the reference's own rdft plugged in, every level is compared with the
 * level of the same name of the real SMILExtract (HTK taps, oracle/conf/egemaps_taps.conf).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "lld_oracle.h"
#include "lld_oracle_gemaps.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ cSpectral, GeMAPS option sets */
/* [gemapsv01b_logSpectral]: slopes 0-500, 500-1500, alphaRatio, hammarbergIndex; [egemapsv02_logSpectral_flux]: flux.
 * Both: squareInput = 1, useLogSpectrum = 1, normBandEnergies = 1, freqRange 0-5000, oldSlopeScale = 0, specFloor 1e-7;
 * the constructor leaves requireMagSpec = requireLogSpec = requirePowerSpec = true (spectral.cpp:86), so the log
 * spectrum is taken of the POWER spectrum with factor 10/ln 10 (:689-716). */
void lldo_gspec_init(lldo_gspec *s, long K, double frame_size_sec)
{
  memset(s, 0, sizeof(*s));
  s->K = K;
  s->frq = (double *)malloc(sizeof(double) * (size_t)K);
  const double F0 = 1.0 / frame_size_sec;                          /* transformFft.cpp:102-117 */
  for (long i = 0; i < K; i++) s->frq[i] = F0 * (double)i;
  /* freqRange 0-5000 -> bins (spectral.cpp:625-644) */
  long lo = -1, hi = -1;
  for (long i = 0; i < K; i++) {
    if ((double)0 >= s->frq[i]) lo = i;
    if ((double)5000 > s->frq[i]) hi = i;
  }
  if (hi == -1 || hi >= K) hi = K - 1;
  if (lo < 0) lo = 0;
  s->lo = lo; s->hi = hi;
  s->prev = (float *)calloc((size_t)K, sizeof(float));
  s->have_prev = 0;
  float specFloor = (float)0.0000001;                              /* :228-235 */
  specFloor = specFloor * specFloor;
  s->spec_floor = specFloor;
  s->log_spec_floor = (float)(10.0 * (double)logf(specFloor) / log(10.0));
}
void lldo_gspec_reset(lldo_gspec *s) { s->have_prev = 0; }
void lldo_gspec_free(lldo_gspec *s) { free(s->frq); free(s->prev); memset(s, 0, sizeof(*s)); }

/* slope of the log spectrum in [lo_hz, hi_hz] with a frequency axis (spectral.cpp:872-992) */
static float band_slope(const lldo_gspec *s, const float *srcLP, long lo_hz, long hi_hz)
{
  const long Nsrc = s->K, nScale = s->K;
  const double *frq = s->frq;
  long ii;
  double idxL, wghtL, idxR, wghtR;
  for (ii = 0; ii < nScale; ii++) if (frq[ii] > (double)lo_hz) break;
  if ((ii < nScale) && (ii > 0)) wghtL = (frq[ii] - (double)lo_hz) / (frq[ii] - frq[ii - 1]); else wghtL = 1.0;
  idxL = (double)ii - 1.0;
  if (idxL < 0) idxL = 0;
  if (idxL >= Nsrc) idxL = Nsrc;
  if (wghtL == 0.0) wghtL = 1.0;
  for (ii = 0; ii < nScale; ii++) if (frq[ii] >= (float)hi_hz) break;
  if ((ii < nScale) && (ii > 0)) wghtR = ((double)hi_hz - frq[ii - 1]) / (frq[ii] - frq[ii - 1]); else wghtR = 1.0;
  if ((ii < nScale) && (frq[ii] == (float)hi_hz)) idxR = (double)ii; else idxR = (double)ii - 1.0;
  if (idxR >= Nsrc) idxR = Nsrc - 1;
  if (wghtR == 0.0) wghtR = 1.0;
  long iL = (long)floor(idxL), iR = (long)floor(idxR);
  if (iL >= Nsrc) { iL = iR = Nsrc - 1; wghtR = 0.0; wghtL = 0.0; }
  if (iR >= Nsrc) { iR = Nsrc - 1; wghtR = 1.0; }
  if (iL < 0) iL = 0;
  if (iR < 0) iR = 0;
  const double Nind = idxR - idxL;
  double Sf = (double)frq[iL] * wghtL;
  double S2f = Sf * Sf;
  double sumA = (double)frq[iL] * wghtL * (double)srcLP[iL];
  double sumB = wghtL * srcLP[iL];
  for (ii = iL + 1; ii < iR && ii < nScale; ii++) {
    S2f += (double)frq[ii] * (double)frq[ii];
    Sf += (double)frq[ii];
    sumA += (double)frq[ii] * (double)srcLP[ii];
    sumB += (double)srcLP[ii];
  }
  S2f += (double)frq[iR] * wghtR * (double)frq[iR] * wghtR;
  Sf += (double)frq[iR] * wghtR;
  sumA += (double)frq[iR] * wghtR * (double)srcLP[iR];
  sumB += wghtR * (double)srcLP[iR];
  const double deno = (Nind * S2f - Sf * Sf);
  double slope = 0.0;
  if (deno != 0.0) slope = (Nind * sumA - Sf * sumB) / deno;
  return (float)slope;                                             /* oldSlopeScale = 0 */
}

/* src: K magnitudes. dst5: [logSpectralSlopeOfBand0-500, logSpectralSlopeOfBand500-1500, alphaRatioDB,
 * hammarbergIndexDB | spectralFlux] -- the 4 outputs of [gemapsv01b_logSpectral] in its own order, then the one of
 * [egemapsv02_logSpectral_flux] (first frame of a stream: 0, spectral.cpp:1132-1136). */
void lldo_gspec_frame(lldo_gspec *s, const float *src, float *dst5)
{
  const long K = s->K;
  float *srcP = (float *)malloc(sizeof(float) * (size_t)K * 2);
  float *srcL = srcP + K;
  const float logSpecFactor = (float)(10.0 / log(10.0));          /* :690 */
  for (long i = 0; i < K; i++) srcP[i] = src[i] * src[i];          /* :677-684 */
  for (long i = 0; i < K; i++) {                                   /* :707-714 */
    if (srcP[i] <= s->spec_floor) srcL[i] = s->log_spec_floor;
    else srcL[i] = logSpecFactor * logf(srcP[i]);
  }
  dst5[0] = band_slope(s, srcL, 0, 500);
  dst5[1] = band_slope(s, srcL, 500, 1500);
  {                                                                /* alpha ratio, :995-1037 */
    float sum01 = 0.0f, sum15 = 0.0f;
    for (long j = 0; j < K; j++) {
      if (s->frq[j] > 5000.0) break;
      if (s->frq[j] < 1000.0) sum01 += srcP[j]; else sum15 += srcP[j];
    }
    if (sum01 > 0.0) {
      if (sum15 > s->spec_floor) dst5[2] = (float)(10.0 * logf(sum15 / sum01) / log(10.0));
      else dst5[2] = (float)(10.0 * (logf(s->spec_floor) - logf(sum01)) / log(10.0));
    } else dst5[2] = 0.0f;
  }
  {                                                                /* Hammarberg index, :1039-1089 */
    float max02 = 0.0f, max25 = 0.0f;
    for (long j = 0; j < K; j++) {
      if (s->frq[j] > 5000.0) break;
      if (s->frq[j] < 2000.0) { if (srcP[j] > max02) max02 = srcP[j]; }
      else { if (srcP[j] > max25) max25 = srcP[j]; }
    }
    if (max25 > 0.0) {
      if (max02 > s->spec_floor) dst5[3] = (float)(10.0 * logf(max02 / max25) / log(10.0));
      else dst5[3] = (float)(10.0 * (logf(s->spec_floor) - logf(max25)) / log(10.0));
    } else dst5[3] = 0.0f;
  }
  {                                                                /* flux of the magnitudes, :1124-1254 */
    const long lo = s->lo, hi = s->hi, nBins = hi - lo + 1;
    if (!s->have_prev) {
      dst5[4] = 0.0f;
      s->have_prev = 1;
    } else {
      double myA = 0.0;
      for (long j = lo; j <= hi; j++) {
        const double myB = ((double)src[j] / 1.0 - (double)s->prev[j - lo] / 1.0);
        myA += myB * myB;
      }
      const double flux = (nBins > 0) ? myA / (double)nBins : 0.0;
      dst5[4] = (flux > 0.0) ? (float)sqrt(flux) : 0.0f;
    }
    for (long j = lo; j <= hi; j++) s->prev[j - lo] = src[j];
  }
  free(srcP);
}

/* ------------------------------------------------------------------ cEnergy energy2 */
/* [egemapsv02_energyRMS] rms = 0, energy2 = 1, log = 0 on the raw 20 ms frame (energy.cpp:152-170) */
float lldo_energy2(const float *x, long N)
{
  double d = 0.0;
  for (long i = 0; i < N; i++) { const float t = x[i]; d += t * t; }
  return (float)(d / (double)N) * 1.0f + 0.0f;
}

/* ------------------------------------------------------------------ cSpecResample */
/* setupNewNames (specResample.cpp:117-172) + smileDsp_initIrdft (smileUtil.c:1752-1786). n_in = Nfft values of the
 * complex spectrum (Ooura packing), fs_sec = frameSizeSec of the spectrum level (Nfft / rate), last_fs_sec = the frame
 * size before zero padding, base_period = 1 / sample rate. */
int lldo_specresample_init(lldo_specresample *r, long n_in, double fs_sec, double last_fs_sec, double base_period,
                           double target_fs)
{
  memset(r, 0, sizeof(*r));
  const double sr = 1.0 / base_period;
  double ratio = target_fs / sr, nd;
  long n_out;
  if ((fs_sec != last_fs_sec) && (last_fs_sec != 0.0) && (last_fs_sec != base_period)) {
    const double nout0 = round((double)n_in * ratio * last_fs_sec / fs_sec);
    const double new_ratio = nout0 / ((double)n_in * (last_fs_sec / fs_sec));
    n_out = (long)nout0;
    if (new_ratio != ratio) { target_fs = sr * new_ratio; ratio = new_ratio; }
    nd = (double)n_in * ratio;
  } else {
    const double nout0 = round((double)n_in * ratio);
    const double new_ratio = nout0 / (double)n_in;
    n_out = (long)nout0;
    if (new_ratio != ratio) { target_fs = sr * new_ratio; ratio = new_ratio; }
    nd = nout0;
  }
  r->K = n_in; r->I = n_out; r->target_fs = target_fs;
  r->kMax = n_in > n_out ? n_out : n_in;
  if (r->kMax & 1) r->kMax--;
  const long h = r->kMax / 2;
  r->costable = (float *)calloc((size_t)(h * n_out + 1), sizeof(float));
  r->sintable = (float *)calloc((size_t)(h * n_out + 1), sizeof(float));
  const double pi2 = 2.0 * M_PI;
  for (long i = 0; i < n_out; i++) {
    const long i_n = i * h - 1;
    if (n_out >= n_in) r->costable[i_n + n_in / 2] = (float)cos((pi2 * (double)((n_in / 2) * i)) / nd);
    for (long k = 2; k < r->kMax; k += 2) {
      const double kn = pi2 * (double)(k / 2 * i) / nd;
      r->costable[i_n + k / 2] = (float)cos(kn);
      r->sintable[i_n + k / 2] = (float)sin(kn);
    }
  }
  return 1;
}
void lldo_specresample_free(lldo_specresample *r) { free(r->costable); free(r->sintable); memset(r, 0, sizeof(*r)); }

/* smileDsp_irdft (smileUtil.c:1800-1820): float accumulation in index order */
void lldo_specresample_frame(const lldo_specresample *r, const float *in, float *out)
{
  const long h = r->kMax / 2;
  const float *costable = r->costable - 1, *sintable = r->sintable - 1;
  for (long i = 0; i < r->I; i++) {
    out[i] = in[0];
    if (r->I >= r->K) out[i] += in[1] * costable[r->K / 2];
    for (long k = 2; k < r->kMax; k += 2) {
      const long k2 = k >> 1;
      out[i] += in[k] * costable[k2];
      out[i] += in[k + 1] * sintable[k2];
    }
    out[i] /= (float)(r->K / 2);
    costable += h;
    sintable += h;
  }
}

/* ------------------------------------------------------------------ cLpc, method = acf */
/* smileDsp_autoCorr (smileUtil.c:1560-1570) + smileDsp_calcLpcAcf (:1572-1630); lpc: p coefficients
 * (cLpc::processVector with saveLPCoeff = 1 only, lpc.cpp:171-213). The coefficient array persists between frames in
 * the reference; every element is rewritten before it is read, or zeroed when r[0] == 0. */
void lldo_lpc_acf(const float *x, long n, int p, float *lpc)
{
  float r[64];
  int lag = p + 1;
  while (lag) {
    r[--lag] = 0.0f;
    for (long i = lag; i < n; i++) r[lag] += x[i] * x[i - lag];
  }
  if ((r[0] == 0.0) || (r[0] == -0.0)) { for (int i = 0; i < p; i++) lpc[i] = 0.0f; return; }
  float e = r[0];
  for (int m = 1; m <= p; m++) {
    float sum = (float)1.0 * r[m];
    for (int i = 1; i < m; i++) sum += lpc[i - 1] * r[m - i];
    const float k_m = ((float)-1.0 / e) * sum;
    lpc[m - 1] = k_m;
    for (int i = 1; i <= m / 2; i++) {
      const float xx = lpc[i - 1];
      lpc[i - 1] += k_m * lpc[m - i - 1];
      if ((i < (m / 2)) || ((m & 1) == 1)) lpc[m - i - 1] += k_m * xx;
    }
    e *= ((float)1.0 - k_m * k_m);
    if (e == 0.0) { for (int i = m; i < p; i++) lpc[i] = 0.0f; break; }
  }
}

/* ------------------------------------------------------------------ polynomial roots (src/smileutil/zerosolve.cpp) */
#define MATC(m, i, j, n) ((m)[(i) * (n) + (j)])
#define MATF(m, i, j, n) ((m)[((i) - 1) * (n) + ((j) - 1)])
#define ZS_EPS 2.2204460492503131e-16                              /* ZEROSOLVER_DBL_EPSILON, zerosolve.h */

static void zs_set_matrix(const double *a, long nc, double *m)     /* zerosolveSetCmatrix, :86-98 */
{
  for (long i = 0; i < nc; i++) for (long j = 0; j < nc; j++) MATC(m, i, j, nc) = 0.0;
  for (long i = 1; i < nc; i++) MATC(m, i, i - 1, nc) = 1.0;
  for (long i = 0; i < nc; i++) MATC(m, i, nc - 1, nc) = -a[i] / a[nc];
}

static void zs_balance(double *m, long nc)                         /* zerosolveBalanceCmatrix, :22-84 */
{
  const double radix = 2.0, radix2 = 4.0;
  int converged = 0;
  double nrow = 0, ncol = 0;
  while (!converged) {
    double t1, t2, t3;
    converged = 1;
    for (long i = 0; i < nc; i++) {
      if (i != nc - 1) ncol = fabs(MATC(m, i + 1, i, nc));
      else { ncol = 0.0; for (long j = 0; j < nc - 1; j++) ncol += fabs(MATC(m, j, nc - 1, nc)); }
      if (i == 0) nrow = fabs(MATC(m, 0, nc - 1, nc));
      else if (i == nc - 1) nrow = fabs(MATC(m, i, i - 1, nc));
      else nrow = (fabs(MATC(m, i, i - 1, nc)) + fabs(MATC(m, i, nc - 1, nc)));
      if (ncol == 0.0 || nrow == 0.0) continue;
      t2 = 1.0; t1 = nrow / radix; t3 = ncol + nrow;
      while (ncol < t1) { t2 *= radix; ncol *= radix2; }
      t1 = nrow * radix;
      while (ncol > t1) { t2 /= radix; ncol /= radix2; }
      if ((nrow + ncol) < 0.95 * t3 * t2) {
        converged = 0;
        t1 = 1.0 / t2;
        if (i == 0) MATC(m, 0, nc - 1, nc) *= t1;
        else { MATC(m, i, i - 1, nc) *= t1; MATC(m, i, nc - 1, nc) *= t1; }
        if (i == nc - 1) { for (long j = 0; j < nc; j++) MATC(m, j, i, nc) *= t2; }
        else MATC(m, i + 1, i, nc) *= t2;
      }
    }
  }
}

static int zs_qr(double *h, long nc, double *root)                 /* zerosolveQRhelper, :100-283 */
{
  long i, j, k, m = 0, e, nit = 0, N = nc;
  double w, s, x, y, z, p = 0, q = 0, r = 0, t = 0.0;
  int notlast;
  if (N == 0) return 1;
  for (;;) {
    for (e = N; e >= 2; e--) {
      const double a1 = fabs(MATF(h, e, e - 1, nc)), a2 = fabs(MATF(h, e - 1, e - 1, nc)), a3 = fabs(MATF(h, e, e, nc));
      if (a1 <= ZS_EPS * (a2 + a3)) break;
    }
    x = MATF(h, N, N, nc);
    if (e == N) {
      root[2 * (N - 1)] = x + t; root[2 * (N - 1) + 1] = 0;
      N--;
      if (N == 0) return 1;
      nit = 0;
      continue;
    }
    y = MATF(h, N - 1, N - 1, nc);
    w = MATF(h, N - 1, N, nc) * MATF(h, N, N - 1, nc);
    if (e == N - 1) {
      p = (y - x) / 2;
      q = p * p + w;
      y = sqrt(fabs(q));
      x += t;
      if (q > 0) {
        if (p < 0) y = -y;
        y += p;
        root[2 * (N - 1)] = x - w / y; root[2 * (N - 1) + 1] = 0;
        root[2 * (N - 2)] = x + y; root[2 * (N - 2) + 1] = 0;
      } else {
        root[2 * (N - 1)] = x + p; root[2 * (N - 1) + 1] = -y;
        root[2 * (N - 2)] = x + p; root[2 * (N - 2) + 1] = y;
      }
      N -= 2;
      if (N == 0) return 1;
      nit = 0;
      continue;
    }
    if (nit == 70) return 0;
    if (nit % 10 == 0 && nit > 0) {
      t += x;
      for (i = 1; i <= N; i++) MATF(h, i, i, nc) -= x;
      s = fabs(MATF(h, N, N - 1, nc)) + fabs(MATF(h, N - 1, N - 2, nc));
      y = 3.0 / 4.0 * s;
      x = y;
      w = -0.4375 * s * s;
    }
    nit++;
    for (m = N - 2; m >= e; m--) {
      z = MATF(h, m, m, nc);
      r = x - z;
      s = y - z;
      p = MATF(h, m, m + 1, nc) + (r * s - w) / MATF(h, m + 1, m, nc);
      q = MATF(h, m + 1, m + 1, nc) - z - r - s;
      r = MATF(h, m + 2, m + 1, nc);
      s = fabs(p) + fabs(q) + fabs(r);
      p /= s; q /= s; r /= s;
      if (m == e) break;
      const double a1 = fabs(MATF(h, m, m - 1, nc)), a2 = fabs(MATF(h, m - 1, m - 1, nc)), a3 = fabs(MATF(h, m + 1, m + 1, nc));
      if (a1 * (fabs(q) + fabs(r)) <= ZS_EPS * fabs(p) * (a2 + a3)) break;
    }
    for (i = m + 2; i <= N; i++) MATF(h, i, i - 2, nc) = 0;
    for (i = m + 3; i <= N; i++) MATF(h, i, i - 3, nc) = 0;
    for (k = m; k <= N - 1; k++) {
      notlast = (k != N - 1);
      if (k != m) {
        p = MATF(h, k, k - 1, nc);
        q = MATF(h, k + 1, k - 1, nc);
        r = notlast ? MATF(h, k + 2, k - 1, nc) : 0.0;
        x = fabs(p) + fabs(q) + fabs(r);
        if (x == 0) continue;
        p /= x; q /= x; r /= x;
      }
      s = sqrt(p * p + q * q + r * r);
      if (p < 0) s = -s;
      if (k != m) MATF(h, k, k - 1, nc) = -s * x;
      else if (e != m) MATF(h, k, k - 1, nc) *= -1;
      p += s;
      z = r / s; y = q / s; x = p / s;
      r /= p; q /= p;
      for (j = k; j <= N; j++) {
        p = MATF(h, k, j, nc) + q * MATF(h, k + 1, j, nc);
        if (notlast) { p += r * MATF(h, k + 2, j, nc); MATF(h, k + 2, j, nc) -= p * z; }
        MATF(h, k + 1, j, nc) -= p * y;
        MATF(h, k, j, nc) -= p * x;
      }
      j = (k + 3 < N) ? k + 3 : N;
      for (i = e; i <= j; i++) {
        p = x * MATF(h, i, k, nc) + y * MATF(h, i, k + 1, nc);
        if (notlast) { p += z * MATF(h, i, k + 2, nc); MATF(h, i, k + 2, nc) -= p * r; }
        MATF(h, i, k + 1, nc) -= p * q;
        MATF(h, i, k, nc) -= p;
      }
    }
  }
}

/* ------------------------------------------------------------------ cFormantLpc */
/* processVector (formantLpc.cpp:192-290) with saveFormants = saveBandwidths = 1, nFormants = nf, medianFilter =
 * octaveCorrection = 0: lpc (n_lpc coefficients) -> dst [nf frequencies | nf bandwidths]; T = sample period of the
 * resampled signal (cSpecResample::configureWriter sets basePeriod = 1 / targetFs). When the QR iteration does not
 * converge the reference goes on with whatever the roots array holds (zerosolve.cpp:337-341): `roots` carries it over
 * from frame to frame like the reference's member does. */
void lldo_formant_lpc(const float *lpc_in, int n_lpc, int nf, double T, double min_f, double max_f, double *roots,
                      float *dst)
{
  double a[64], mat[64 * 64], fc[32], bc[32];
  for (int i = 0; i < n_lpc; i++) a[i] = -lpc_in[n_lpc - i - 1];
  a[n_lpc] = 1.0;
  zs_set_matrix(a, n_lpc, mat);
  zs_balance(mat, n_lpc);
  zs_qr(mat, n_lpc, roots);
  for (int i = 0; i < n_lpc; i++) {                                /* smileMath_complexIntoUnitCircle, smileUtil.c:992-1003 */
    const double re = roots[2 * i], im = roots[2 * i + 1];
    if (sqrt(re * re + im * im) > 1.0) {
      /* smileMath_complexDiv(1, 0, re, -im), smileUtil.c:963-990 */
      const double c = re, d = -im;
      double r, den, e, f;
      if (fabs(c) >= fabs(d)) {
        if (c == 0.0) { e = 0; f = 0; }
        else { r = d / c; den = c + r * d; e = (1.0 + r * 0.0) / den; f = (0.0 - r * 1.0) / den; }
      } else {
        r = c / d; den = d + r * c; e = (1.0 * r + 0.0) / den; f = (0.0 * r - 1.0) / den;
      }
      roots[2 * i] = e; roots[2 * i + 1] = f;
    }
  }
  /* smileDsp_lpcrootsToFormants (smileUtil.c:2019-2053) */
  int n_found = 0;
  {
    const double spPi = T * M_PI, spPi2 = spPi * 2.0;
    double fHigh = max_f;
    if ((fHigh < min_f) || (fHigh > 1.0 / T)) fHigh = 0.5 / T - min_f;
    for (int i = 0; i < n_lpc; i++) {
      const double re = roots[2 * i], im = roots[2 * i + 1];
      if (im < 0) continue;
      const double f = fabs(atan2(im, re)) / spPi2;
      if ((f >= min_f) && (f <= fHigh)) {
        bc[n_found] = -log(sqrt(re * re + im * im)) / spPi;
        fc[n_found] = f;
        n_found++;
        if (n_found >= nf) break;
      }
    }
    for (int i = n_found; i < nf; i++) { fc[i] = 0.0; bc[i] = 0.0; }
  }
  int nz = 0;                                                      /* sort ascending, formantLpc.cpp:270-289 */
  for (nz = 0; nz < nf; nz++) if (fc[nz] == 0.0) break;
  for (int i = 0; i < nz; i++)
    for (int j = i + 1; j < nz; j++)
      if (fc[j] < fc[i]) {
        double t = fc[j]; fc[j] = fc[i]; fc[i] = t;
        t = bc[j]; bc[j] = bc[i]; bc[i] = t;
      }
  for (int i = 0; i < nf; i++) { dst[i] = (float)fc[i]; dst[nf + i] = (float)bc[i]; }
}

/* ------------------------------------------------------------------ cHarmonics */
/* smileMath_quadFrom3pts (smileUtil.c:1009-1033) */
static double quad3(double x1, double y1, double x2, double y2, double x3, double y3, double *y)
{
  const double den = x1 * x1 * x2 + x2 * x2 * x3 + x3 * x3 * x1 - x3 * x3 * x2 - x2 * x2 * x1 - x1 * x1 * x3;
  if (den != 0.0) {
    const double a = (y1 * x2 + y2 * x3 + y3 * x1 - y3 * x2 - y2 * x1 - y1 * x3) / den;
    const double b = (x1 * x1 * y2 + x2 * x2 * y3 + x3 * x3 * y1 - x3 * x3 * y2 - x2 * x2 * y1 - x1 * x1 * y3) / den;
    const double c = (x1 * x1 * x2 * y3 + x2 * x2 * x3 * y1 + x3 * x3 * x1 * y2 - x3 * x3 * x2 * y1 - x2 * x2 * x1 * y3 - x1 * x1 * x3 * y2) / den;
    if (a != 0.0) {
      const double x = -b / (2.0 * a);
      *y = c - a * x * x;
      return x;
    }
  }
  if (y1 > y2 && y1 > y3) { *y = y1; return x1; }
  if (y2 > y1 && y2 > y3) { *y = y2; return x2; }
  if (y3 > y1 && y3 > y2) { *y = y3; return x3; }
  *y = y1;
  return x1;
}

static int is_peak(const float *x, long N, long n)                  /* cHarmonics::isPeak, harmonics.cpp:369-390 */
{
  if (n >= N || n < 0) return 0;
  if (n + 1 < N) {
    if (n > 0) { if (x[n] > x[n - 1] && x[n] > x[n + 1]) return 1; }
    else { if (x[0] > x[1]) return 1; }
  } else {
    if (n > 0) { if (x[n] > x[n - 1]) return 1; }
  }
  return 0;
}

static int freq_to_bin(const double *frq, long nFrq, float freq, int start)   /* freqToBin, :403-415 */
{
  for (; start < nFrq; start++) {
    if (frq[start] > freq) {
      if (frq[start] - freq > freq - frq[start - 1]) return start - 1;
      return start;
    }
  }
  return 0;
}

static long closest_peak(const float *x, long N, long idx)          /* getClosestPeak, :632-665 */
{
  if (is_peak(x, N, idx)) return idx;
  long o = 1;
  while (idx - o > 0 || idx + o < N - 1) {
    if (idx - o > 0) { if (is_peak(x, N, idx - o)) return idx - o; }
    if (idx + o < N - 1) { if (is_peak(x, N, idx + o)) return idx + o; }
    o++;
  }
  if (x[0] > x[idx] && x[N - 1] <= x[idx]) return 0;
  else if (x[0] <= x[idx] && x[N - 1] > x[idx]) return N - 1;
  else if (x[0] > x[idx] && x[N - 1] > x[idx]) return (idx < N / 2) ? 0 : N - 1;
  return idx;
}

typedef struct { int bin; float freqExpected, freqFromBin, freqInterpolated, magnitude, magnitudeInterpolated, magnitudeLogRelF0; } harm_t;

/* cHarmonics::processVector (harmonics.cpp:743-1031) with [gemapsv01b_harmonics]'s options: nHarmonics = 100,
 * harmonicDifferences = H1-H2; H1-A3 (log), formant amplitudes 1..3 (log, relative to F0), computeAcfHnrLogdB = 1.
 * F0: element F0final of level gemapsv01b_logPitch (Hz); formants: [5 frequencies | 5 bandwidths] of level
 * gemapsv01b_formants; mag: the K = 513 magnitudes of the 60 ms frame (frequency axis i * F0bin, fs_sec = Nfft / rate).
 * dst6: [HarmonicsToNoiseRatioACFLogdB, HarmonicDifferenceLogRelH1-H2, HarmonicDifferenceLogRelH1-A3,
 *        FormantAmplitudeByMaxHarmonicLogRelF0[1..3]]. */
void lldo_harmonics_frame(float F0, const float *formants, int n_formants, const float *mag, long K, double fs_sec, float *dst6)
{
  enum { NH = 100 };
  double *frq = (double *)malloc(sizeof(double) * (size_t)K);
  const double F0bin = 1.0 / fs_sec;
  for (long i = 0; i < K; i++) frq[i] = F0bin * (double)i;
  long n = 0;
  /* HNR from the ACF of the squared magnitudes (computeAcf :590-630, computeAcfHnr_dB :690-712) */
  {
    const double fs = frq[K - 1] * 2.0;
    const long F0acfBin = (F0 > 0.0) ? (long)(int)floor(fs / F0) : 0;           /* freqToAcfBinLin, :393-401 */
    const long N = (K - 1) * 2;
    float *d = (float *)malloc(sizeof(float) * (size_t)N);
    float *acf = (float *)malloc(sizeof(float) * (size_t)K);
    d[0] = mag[0] * mag[0];
    d[1] = mag[K - 1] * mag[K - 1];
    for (long i = 2; i < N - 1; i += 2) { d[i] = mag[i >> 1] * mag[i >> 1]; d[i + 1] = 0.0f; }
    lldo_irfft_packed_real(d, N);
    for (long i = 0; (i < N) && (i < K); i++) acf[i] = (float)fabs(d[i]) / (float)K;
    long refined = 0;
    if (F0acfBin > 0) refined = closest_peak(acf, K, F0acfBin);
    float v = 0.0f;
    if (refined > 0) {
      double hnr = acf[0] - acf[refined], ret;
      if (hnr == 0.0) hnr = 10e10; else hnr = acf[refined] / hnr;
      if (hnr > 10e10) ret = 10.0 * log10(10e10);
      else if (hnr < 10e-10) ret = 10.0 * log10(10e-10);
      else ret = 10.0 * log10(hnr);
      v = (float)ret;
    }
    dst6[n++] = v;
    free(d); free(acf);
  }
  if (F0 > 0.0) {
    harm_t h[NH];
    memset(h, 0, sizeof(h));
    /* findHarmonicPeaks, frequency-axis branch (:478-546) */
    int lastBin = freq_to_bin(frq, K, 0.5f * F0, 1);
    const int firstBin = freq_to_bin(frq, K, 0.5f * F0, lastBin);
    for (int i = 0; i < NH; i++) {
      const int candBin = freq_to_bin(frq, K, (float)(i + 1) * F0, lastBin);
      int peakBin = -1;
      if (candBin >= K) {
        h[i].freqExpected = 0.0f; h[i].magnitudeLogRelF0 = -201.0f; h[i].bin = -1; h[i].freqFromBin = 0.0f;
        h[i].freqInterpolated = 0.0f; h[i].magnitude = 0.0f; h[i].magnitudeInterpolated = 0.0f;
        continue;
      }
      if (is_peak(mag, K, candBin)) peakBin = candBin;
      else {
        int cl = candBin - 1, cr = candBin + 1;
        const int lower = freq_to_bin(frq, K, ((float)i + 0.5f) * F0, lastBin);
        const int upper = freq_to_bin(frq, K, ((float)i + 1.5f) * F0, candBin);
        while ((cl >= lower || cr <= upper) && peakBin == -1) {
          if (cr <= upper) { if (is_peak(mag, K, cr)) { peakBin = cr; break; } cr++; }
          if (cl >= lower) { if (is_peak(mag, K, cl)) { peakBin = cl; break; } cl--; }
        }
      }
      h[i].freqExpected = (float)(i + 1) * F0;
      h[i].magnitudeLogRelF0 = -201.0f;
      if (peakBin >= firstBin && peakBin < K - 1) {
        h[i].bin = peakBin;
        h[i].freqFromBin = (float)frq[peakBin];
        h[i].magnitude = mag[peakBin];
        double mi = 0.0;
        h[i].freqInterpolated = (float)quad3(frq[peakBin - 1], (double)mag[peakBin - 1], frq[peakBin], (double)mag[peakBin],
                                             frq[peakBin + 1], (double)mag[peakBin + 1], &mi);
        h[i].magnitudeInterpolated = (float)mi;
      } else {
        h[i].bin = candBin; h[i].freqFromBin = 0.0f; h[i].freqInterpolated = 0.0f; h[i].magnitude = 0.0f;
        h[i].magnitudeInterpolated = 0.0f;
      }
      lastBin = candBin;
    }
    /* postProcessHarmonics(…, true), :550-588; log10 of a float is log10f */
    {
      int logRel = 1;
      float magF0 = h[0].magnitude;
      if (magF0 == 0.0) logRel = 0; else magF0 = log10f(magF0);
      h[0].magnitudeLogRelF0 = 0.0f;
      for (int i = 1; i < NH; i++) {
        if (logRel) {
          if (h[i].magnitudeInterpolated > 0.0) {
            const double tmp = log10f(h[i].magnitudeInterpolated);
            h[i].magnitudeLogRelF0 = (float)(20.0 * (tmp - magF0));
            if (h[i].magnitudeLogRelF0 < -200.0) h[i].magnitudeLogRelF0 = -200.0f;
          } else h[i].magnitudeLogRelF0 = -200.0f;
        } else h[i].magnitudeLogRelF0 = -201.0f;
        if (h[i].bin == h[i - 1].bin) {
          h[i].bin = 0; h[i].freqFromBin = 0.0f; h[i].freqInterpolated = 0.0f; h[i].freqExpected = 0.0f;
          h[i].magnitude = 0.0f; h[i].magnitudeInterpolated = 0.0f; h[i].magnitudeLogRelF0 = -201.0f;
        }
      }
    }
    /* getFormantAmplitudeIndices, :714-740 */
    int fa[16];
    for (int f = 0; f < n_formants; f++) {
      const float fl = 0.8f * formants[f], fr = 1.2f * formants[f];
      int mi = -1;
      float mm = 0.0f;
      for (int k = 0; k < NH; k++)
        if (h[k].freqInterpolated >= fl && h[k].freqInterpolated <= fr)
          if (h[k].magnitude > mm) { mi = k; mm = h[k].magnitude; }
      fa[f] = mi;
    }
    /* harmonic differences H1-H2, H1-A3 (:876-956) */
    const int h1[2] = {1, 1}, h2[2] = {2, fa[2]};
    for (int i = 0; i < 2; i++) {
      float v;
      if (h1[i] >= 0 && h2[i] >= 0 && h1[i] < NH && h2[i] < NH) v = h[h1[i]].magnitudeLogRelF0 - h[h2[i]].magnitudeLogRelF0;
      else v = (float)(h[h1[i]].magnitudeLogRelF0 - 201.0f);
      if (v < (float)-201.0) v = (float)-201.0;
      if (v > (float)201.0) v = (float)201.0;
      dst6[n++] = v;
    }
    for (int i = 1; i <= 3; i++) dst6[n++] = (fa[i - 1] >= 0) ? h[fa[i - 1]].magnitudeLogRelF0 : 0.0f;   /* :957-990 */
  } else {
    dst6[n++] = 0.0f; dst6[n++] = 0.0f;                              /* :1005-1014 */
    for (int i = 1; i <= 3; i++) dst6[n++] = (float)-201.0;          /* logRelValueFloorUnvoiced, :1021-1025 */
  }
  free(frq);
}

/* ------------------------------------------------------------------ the frame-level part of the graph */
static float vec_ll1(const float *src, int N)                       /* cVectorOperation ll1, vectorOperation.cpp:475-481 */
{
  float d = 0.0f;
  for (int i = 0; i < N; i++) d += src[i];
  if (N > 0) d /= (float)N;
  return d;
}

void lldo_egemaps_levels_free(lldo_egemaps_lv *L)
{
  free(L->loudness); free(L->lspec); free(L->flux); free(L->mfcc); free(L->energy2); free(L->formants);
  free(L->pitch); free(L->jitter); free(L->harm); free(L->shs); free(L->e60);
  memset(L, 0, sizeof(*L));
}

/* Every per-frame level of config/gemaps/v01b/GeMAPSv01b_core.lld.conf.inc + config/egemaps/v02/eGeMAPSv02_core.lld.conf.inc
 * for one utterance (16 kHz): 20 ms Hamming frames (T20) and 60 ms Gauss frames (T60), step 10 ms. Returns T60. */
/* GeMAPSv01a / eGeMAPSv01a instead of v01b / v02: the same graph with three option values of openSMILE 2.2 kept for compatibility
 * (config/gemaps/v01a/GeMAPSv01a_core.lld.conf.inc:34,67 zeroPadSymmetric = 0 in both cTransformFFT instances, :199
 * useBrokenJitterThresh = 1, :281 maxF = 5500.0 in cFormantLpc) */
static int g_lldo_gemaps_v01a = 0;
void lldo_gemaps_set_v01a(int on) { g_lldo_gemaps_v01a = on ? 1 : 0; }
extern int g_lldo_is13;     /* lld_oracle_compare.c; read by the jitter restatement for useBrokenJitterThresh */

long lldo_egemaps_levels(const int16_t *pcm, long n_samples, lldo_egemaps_lv *L)
{
  memset(L, 0, sizeof(*L));
  lldo_mfcc_cfg c;
  lldo_default_mfcc12_cfg(&c);
  c.frame_size_sec = 0.020; c.preemph_enable = 0; c.zero_pad_symmetric = 1;
  c.lofreq = 20.0f; c.first_mfcc = 1; c.last_mfcc = 4; c.n_delta = 0;
  lldo_geom g, g60;
  lldo_geometry(&c, &g);
  lldo_mfcc_cfg c60 = c;
  c60.frame_size_sec = 0.060;
  lldo_geometry(&c60, &g60);
  const long T20 = lldo_num_frames(n_samples, g.N, g.H), T60 = lldo_num_frames(n_samples, g60.N, g60.H);
  L->T20 = T20 > 0 ? T20 : 0; L->T60 = T60 > 0 ? T60 : 0;
  if (T20 <= 0) return 0;
  float *x = (float *)malloc(sizeof(float) * (size_t)n_samples);
  lldo_pcm16_to_float(pcm, n_samples, x);
  /* ---- 20 ms chain: gemapsv01b_frame25 -> winH25 -> fftcH25 -> fftmagH25 */
  double *w = (double *)malloc(sizeof(double) * (size_t)g.N);
  lldo_window_table(LLDO_WIN_HAMM, g.N, 0.4, 1.0, w);
  lldo_mel mel1, mel2; lldo_dct dct;
  lldo_mel_init(&mel1, g.K, g.frame_size_sec_fft, 26, 20.0f, 8000.0f, 1, 0);   /* [gemapsv01b_melspec1] htk = 0 */
  lldo_mel_init(&mel2, g.K, g.frame_size_sec_fft, 26, 20.0f, 8000.0f, 1, 1);   /* [egemapsv02_melspecMfcc] htk = 1 */
  lldo_mfcc_init(&dct, 26, 1, 4, 22.0f, 1, c.melfloor);
  double band_hz[26];
  for (int m = 1; m <= 26; m++) band_hz[m - 1] = 700.0 * (exp((double)mel1.cfs[m] / 1127.0) - 1.0);   /* melspec.cpp:408-412 */
  lldo_plp plp;
  lldo_plp_init(&plp, 26, band_hz, 0, c.frame_step_sec);
  lldo_gspec gs;
  lldo_gspec_init(&gs, g.K, g.frame_size_sec_fft);
  lldo_specresample rs;
  lldo_specresample_init(&rs, g.Nfft, g.frame_size_sec_fft, (double)g.N / c.sample_rate, 1.0 / c.sample_rate, 11000.0);
  float *fr = (float *)malloc(sizeof(float) * (size_t)g60.N);
  float *sp = (float *)malloc(sizeof(float) * (size_t)g60.Nfft);
  float *mg = (float *)malloc(sizeof(float) * (size_t)g60.K);
  float *res = (float *)malloc(sizeof(float) * (size_t)rs.I);
  L->loudness = (float *)calloc((size_t)T20, sizeof(float));
  L->lspec = (float *)calloc((size_t)T20 * 4, sizeof(float));
  L->flux = (float *)calloc((size_t)T20, sizeof(float));
  L->mfcc = (float *)calloc((size_t)T20 * 4, sizeof(float));
  L->energy2 = (float *)calloc((size_t)T20, sizeof(float));
  L->formants = (float *)calloc((size_t)T20 * 10, sizeof(float));
  float mb[26], aud[26], s5[5], lpc[11];
  double roots[22];
  memset(roots, 0, sizeof(roots));
  for (long t = 0; t < T20; t++) {
    const float *src = x + t * g.H;
    L->energy2[t] = lldo_energy2(src, g.N);                          /* [egemapsv02_energyRMS] on gemapsv01b_frame25 */
    lldo_window_apply(src, fr, g.N, w, 0.0);
    lldo_rfft_frame(fr, g.N, sp, g.Nfft, g_lldo_gemaps_v01a ? 0 : 1);
    lldo_fftmag(sp, g.Nfft, mg);
    lldo_melspec(&mel1, mg, mb);
    lldo_plp_audspec(&plp, mb, aud);                                 /* [gemapsv01b_audspec] */
    L->loudness[t] = vec_ll1(aud, 26);                               /* [gemapsv01b_audspecSum] ll1 */
    lldo_gspec_frame(&gs, mg, s5);
    memcpy(L->lspec + 4 * t, s5, sizeof(float) * 4);
    L->flux[t] = s5[4];
    lldo_melspec(&mel2, mg, mb);
    lldo_mfcc(&dct, mb, L->mfcc + 4 * t);
    lldo_specresample_frame(&rs, sp, res);                           /* [gemapsv01b_resampLpc] on the complex spectrum */
    lldo_lpc_acf(res, rs.I, 11, lpc);                                /* [gemapsv01b_lpc] p = 11 */
    lldo_formant_lpc(lpc, 11, 5, 1.0 / rs.target_fs, 50.0, g_lldo_gemaps_v01a ? 5500.0 : 5450.0, roots, L->formants + 10 * t);
  }
  lldo_mel_free(&mel1); lldo_mel_free(&mel2); lldo_mfcc_free(&dct); lldo_plp_free(&plp); lldo_gspec_free(&gs);
  lldo_specresample_free(&rs);
  free(w);
  /* ---- 60 ms chain: frame60 -> gauss -> fft -> mag -> cSpecScale -> cPitchShs -> Viterbi -> gate; harmonics */
  if (T60 > 0) {
    const long T = T60;
    w = (double *)malloc(sizeof(double) * (size_t)g60.N);
    lldo_window_table(LLDO_WIN_GAUSS, g60.N, 0.4, 1.0, w);
    lldo_specscale ss;
    lldo_shs sh;
    lldo_specscale_init(&ss, g60.K, g60.frame_size_sec_fft);
    lldo_shs_init(&sh, &ss);
    sh.min_pitch = 55.0; sh.max_pitch = 1000.0;                      /* [gemapsv01b_shs] */
    float *hp = (float *)malloc(sizeof(float) * (size_t)g60.K);
    float *mags = (float *)malloc(sizeof(float) * (size_t)g60.K * (size_t)T);
    L->shs = (float *)calloc((size_t)T * 21, sizeof(float));
    L->e60 = (float *)calloc((size_t)T, sizeof(float));
    L->pitch = (float *)calloc((size_t)T * 3, sizeof(float));
    L->jitter = (float *)calloc((size_t)T * 2, sizeof(float));
    L->harm = (float *)calloc((size_t)T * 6, sizeof(float));
    for (long t = 0; t < T; t++) {
      lldo_window_apply(x + t * g60.H, fr, g60.N, w, 0.0);
      L->e60[t] = lldo_energy_rms(fr, g60.N);                        /* [gemapsv01b_energy60] on winG60 */
      lldo_rfft_frame(fr, g60.N, sp, g60.Nfft, g_lldo_gemaps_v01a ? 0 : 1);
      lldo_fftmag(sp, g60.Nfft, mags + t * g60.K);
      lldo_specscale_frame(&ss, mags + t * g60.K, hp);
      lldo_pitch_shs(&sh, hp, L->shs + 21 * t, NULL);
    }
    float *v2 = (float *)malloc(sizeof(float) * 2 * (size_t)T);
    lldo_pitch_viterbi_ex(L->shs, T, sh.voicing_cutoff, 40, v2, NULL, &L->P);    /* bufferLength = 40 */
    float *f0 = (float *)malloc(sizeof(float) * (size_t)T);
    for (long t = 0; t < T; t++) {
      float f = v2[2 * t], s = 0.0f, vp = v2[2 * t + 1];
      /* F0finalLog (pitchSmootherViterbi.cpp:497-505): semitones above 27.5 Hz, float arithmetic throughout */
      if (f > 29.136) s = (float)12.0 * logf(f / (float)27.5) / logf((float)2.0);
      else if (f > 0.0) s = 1.0f;
      if (!(L->e60[t] > (float)0.001)) { f = 0.0f; s = 0.0f; vp = 0.0f; }         /* [gemapsv01b_volmerge] */
      L->pitch[3 * t] = f; L->pitch[3 * t + 1] = s; L->pitch[3 * t + 2] = vp;
      f0[t] = f;
    }
    float *j4 = (float *)malloc(sizeof(float) * 4 * (size_t)T), *sdb = (float *)malloc(sizeof(float) * (size_t)T);
    { const int keep = g_lldo_is13;
      if (g_lldo_gemaps_v01a) g_lldo_is13 = 1;                        /* useBrokenJitterThresh = 1 */
      lldo_pitch_jitter_ex(x, n_samples, f0, T, g60.N, g60.H, c.sample_rate, c.frame_step_sec, 0.1, j4, sdb);
      g_lldo_is13 = keep; }
    for (long t = 0; t < T; t++) {
      L->jitter[2 * t] = j4[4 * t]; L->jitter[2 * t + 1] = sdb[t];
      lldo_harmonics_frame(f0[t], L->formants + 10 * t, 5, mags + t * g60.K, g60.K, g60.frame_size_sec_fft, L->harm + 6 * t);
    }
    free(v2); free(f0); free(j4); free(sdb); free(hp); free(mags); free(w);
    lldo_specscale_free(&ss);
  }
  free(x); free(fr); free(sp); free(mg); free(res);
  return L->T60;
}

/* ------------------------------------------------------------------ selectors, gates, smoothers: the LLD level */
/* cContourSmoother::processBuffer (contourSmoother.cpp:85-118), smaWin = 3, over rows 0 .. rows-1 of a level holding T
 * frames of D values; clip[n] = the last frame row n can see (later ones are replaced by it: cDataMemoryLevel::getMatrix
 * pads with the last frame it holds, dataMemoryLevel.cpp:1699-1708), frames before the first are replaced by frame 0. */
static void sma3(const float *x, long T, int D, long rows, const long *clip, int nz, float *y)
{
  for (long n = 0; n < rows; n++) {
    long c = clip ? clip[n] : T - 1;
    if (c > T - 1) c = T - 1;
    if (c < 0) c = 0;
#define IX(i) ((i) < 0 ? 0 : ((i) > c ? c : (i)))
    const float *a = x + IX(n) * D, *l = x + IX(n - 1) * D, *r = x + IX(n + 1) * D;
#undef IX
    for (int d = 0; d < D; d++) {
      float v;
      if (nz) {
        if (a[d] != 0.0) {
          long N = 1;
          v = a[d];
          if (l[d] != 0.0) { v += l[d]; N++; }
          if (r[d] != 0.0) { v += r[d]; N++; }
          v /= (float)N;
        } else v = 0.0f;
      } else {
        v = a[d];
        v += l[d];
        v += r[d];
        v /= (float)3;
      }
      y[n * D + d] = v;
    }
  }
}

/* End-of-input behaviour of the nine cContourSmoother instances, measured against the real binary (taps of
 * oracle/conf/egemaps_taps.conf; T = T60 = 1 .. 995, P = frames the Viterbi smoother had not decided when the input
 * ended) and explained by the tick loop (componentManager.cpp:1416-1560; every smoother has noPostEOIprocessing = 0):
 *  - levels fed by 20 ms frames only (lldsetE, loudness, lldSetNoF0AndLoudnessZ): T20 + 1 rows, no effect.
 *  - levels that follow the Viterbi smoother but not cPitchJitter (lld_single_logF0, lldSetSpectralNz, lldSetSpectralZ):
 *    no effect when P < T (the smoother runs one frame behind its input); when P == T nothing was decided before the end
 *    of input, the P frames are then released one per tick and the smoother follows in lockstep in end-of-input mode:
 *    row n sees frames 0 .. n only, and row 0 reads the never-written slot of frame 1 as zero (getMatrix's left-padding
 *    branch, dataMemoryLevel.cpp:1687-1698), i.e. row 0 = x[0].
 *  - levels that also wait for cPitchJitter (lldsetF, lldSetNoF0AndLoudnessNz; cPitchJitter does not run in
 *    end-of-input ticks, pitchJitter.cpp:593): at the first end of input they hold T - P frames: rows n <= T - P see the
 *    level clipped at frame T - P - 1; P == T: no effect (the smoother has nothing to read until the next normal phase);
 *    T == 1: row 0 = x[0] as above. */
enum { EOI_NONE = 0, EOI_LOCKSTEP = 1, EOI_JITTER = 2 };
static void smooth_level(const float *x, long T, int D, long P, int nz, int eoi_kind, float *y)
{
  const long rows = T + 1;
  long *clip = (long *)malloc(sizeof(long) * (size_t)rows);
  int first_exact = 0;
  for (long n = 0; n < rows; n++) clip[n] = T - 1;
  if (eoi_kind == EOI_LOCKSTEP && P >= T) {
    for (long n = 0; n < rows; n++) clip[n] = n;
    first_exact = 1;
  } else if (eoi_kind == EOI_JITTER) {
    if (T == 1) first_exact = 1;
    else if (P < T) for (long n = 0; n <= T - P && n < rows; n++) clip[n] = T - P - 1;
  }
  sma3(x, T, D, rows, clip, nz, y);
  if (first_exact) memcpy(y, x, sizeof(float) * (size_t)D);
  free(clip);
}

void lldo_egemaps_smo_free(lldo_egemaps_smo *S)
{
  free(S->E); free(S->F); free(S->logf0); free(S->loud); free(S->NoZ); free(S->NoNz); free(S->specV); free(S->specU);
  memset(S, 0, sizeof(*S));
}

/* The smoothed levels of the graph from the per-frame levels: cDataSelector picks (dataSelector.cpp), cValbasedSelector
 * gates on F0finalLog with threshold 1e-6 (valbasedSelector.cpp:139-237; `invert` for the unvoiced set), then the
 * smoothers. Requires T60 >= 1. */
void lldo_egemaps_smooth(const lldo_egemaps_lv *L, lldo_egemaps_smo *S)
{
  memset(S, 0, sizeof(*S));
  const long T20 = L->T20, T = L->T60, P = L->P;
  if (T < 1) return;
  S->T20 = T20; S->T60 = T; S->P = P;
  float *E = (float *)malloc(sizeof(float) * 10 * (size_t)T20), *NoZ = (float *)malloc(sizeof(float) * 5 * (size_t)T20);
  float *F = (float *)malloc(sizeof(float) * 15 * (size_t)T), *lf = (float *)malloc(sizeof(float) * (size_t)T);
  float *NoNz = (float *)malloc(sizeof(float) * 14 * (size_t)T), *sV = (float *)malloc(sizeof(float) * 9 * (size_t)T);
  float *sU = (float *)malloc(sizeof(float) * 5 * (size_t)T);
  for (long t = 0; t < T20; t++) {
    const float *ls = L->lspec + 4 * t, *mf = L->mfcc + 4 * t;
    float *e = E + 10 * t;                                           /* [egemapsv02_lldSetSelectorE] */
    e[0] = L->loudness[t]; e[1] = ls[2]; e[2] = ls[3]; e[3] = ls[0]; e[4] = ls[1]; e[5] = L->flux[t];
    memcpy(e + 6, mf, sizeof(float) * 4);
    NoZ[5 * t] = L->flux[t];                                         /* [egemapsv02_lldSetSelectorNoF0LoudnZ] */
    memcpy(NoZ + 5 * t + 1, mf, sizeof(float) * 4);
  }
  for (long t = 0; t < T; t++) {
    const float *p = L->pitch + 3 * t, *j = L->jitter + 2 * t, *h = L->harm + 6 * t, *fm = L->formants + 10 * t;
    const float *ls = L->lspec + 4 * t, *mf = L->mfcc + 4 * t;
    const float lg = p[1];
    lf[t] = lg;                                                      /* [gemapsv01b_lldSetSelectorLogF0] */
    float *f = F + 15 * t;                                           /* [egemapsv02_lldSetSelectorF] */
    f[0] = lg; f[1] = j[0]; f[2] = j[1]; f[3] = h[0]; f[4] = h[1]; f[5] = h[2];
    for (int k = 0; k < 3; k++) { f[6 + 3 * k] = fm[k]; f[7 + 3 * k] = fm[5 + k]; f[8 + 3 * k] = h[3 + k]; }
    const int voiced = lg > (float)0.000001, unvoiced = lg < (float)0.000001;
    float *z = NoNz + 14 * t;                                        /* [gemapsv01b_formantVoiced] + [egemapsv02_lldSetSelectorNoF0LoudnNz] */
    z[0] = j[0]; z[1] = j[1]; z[2] = h[0]; z[3] = h[1]; z[4] = h[2];
    for (int k = 0; k < 3; k++) {
      z[5 + 3 * k] = voiced ? fm[k] : 0.0f; z[6 + 3 * k] = voiced ? fm[5 + k] : 0.0f; z[7 + 3 * k] = h[3 + k];
    }
    const float sp[9] = {ls[2], ls[3], ls[0], ls[1], L->flux[t], mf[0], mf[1], mf[2], mf[3]};
    for (int k = 0; k < 9; k++) sV[9 * t + k] = voiced ? sp[k] : 0.0f;     /* [egemapsv02_logSpectralVoiced] + SelectorSpectralNz */
    for (int k = 0; k < 5; k++) sU[5 * t + k] = unvoiced ? sp[k] : 0.0f;   /* [egemapsv02_logSpectralUnvoiced] + SelectorSpectralZ */
  }
  S->E = (float *)malloc(sizeof(float) * 10 * (size_t)(T20 + 1));
  S->loud = (float *)malloc(sizeof(float) * (size_t)(T20 + 1));
  S->NoZ = (float *)malloc(sizeof(float) * 5 * (size_t)(T20 + 1));
  S->F = (float *)malloc(sizeof(float) * 15 * (size_t)(T + 1));
  S->logf0 = (float *)malloc(sizeof(float) * (size_t)(T + 1));
  S->NoNz = (float *)malloc(sizeof(float) * 14 * (size_t)(T + 1));
  S->specV = (float *)malloc(sizeof(float) * 9 * (size_t)(T + 1));
  S->specU = (float *)malloc(sizeof(float) * 5 * (size_t)(T + 1));
  smooth_level(E, T20, 10, 0, 0, EOI_NONE, S->E);                    /* [egemapsv02_smoE] */
  smooth_level(L->loudness, T20, 1, 0, 0, EOI_NONE, S->loud);        /* [gemapsv01b_smoLoudness] */
  smooth_level(NoZ, T20, 5, 0, 0, EOI_NONE, S->NoZ);                 /* [egemapsv02_smoNoFLZ] */
  smooth_level(F, T, 15, P, 1, EOI_JITTER, S->F);                    /* [egemapsv02_smoFnz] */
  smooth_level(NoNz, T, 14, P, 1, EOI_JITTER, S->NoNz);              /* [egemapsv02_smoNoF0andLoudnNz] */
  smooth_level(lf, T, 1, P, 1, EOI_LOCKSTEP, S->logf0);              /* [gemapsv01b_smoF0] */
  smooth_level(sV, T, 9, P, 1, EOI_LOCKSTEP, S->specV);              /* [egemapsv02_smoSpectralNz] */
  smooth_level(sU, T, 5, P, 1, EOI_LOCKSTEP, S->specU);              /* [egemapsv02_smoSpectralZ] */
  free(E); free(NoZ); free(F); free(lf); free(NoNz); free(sV); free(sU);
}

/* The LLD level of eGeMAPSv02.conf ([lldconcat]: egemapsv02_lldsetE_smo; egemapsv02_lldsetF_smo), 25 columns:
 *   Loudness, alphaRatio, hammarbergIndex, slope0-500, slope500-1500, spectralFlux, mfcc1..4 (_sma3) |
 *   F0semitoneFrom27.5Hz, jitterLocal, shimmerLocaldB, HNRdBACF, logRelF0-H1-H2, logRelF0-H1-A3, F1frequency, F1bandwidth,
 *   F1amplitudeLogRelF0, F2..., F3... (_sma3nz)
 * rows = T60 + 1 (the rows both levels hold); 0 if the input has no 60 ms frame. out25 == NULL: query. */
long lldo_egemaps_lld_chain(const int16_t *pcm, long n_samples, float *out25)
{
  const long T60 = lldo_num_frames(n_samples, lround(0.060 * lldo_get_sample_rate()), lround(0.010 * lldo_get_sample_rate()));
  if (T60 < 1) return 0;
  if (!out25) return T60 + 1;
  lldo_egemaps_lv L;
  lldo_egemaps_smo S;
  lldo_egemaps_levels(pcm, n_samples, &L);
  lldo_egemaps_smooth(&L, &S);
  for (long r = 0; r <= T60; r++) {
    memcpy(out25 + 25 * r, S.E + 10 * r, sizeof(float) * 10);
    memcpy(out25 + 25 * r + 10, S.F + 15 * r, sizeof(float) * 15);
  }
  lldo_egemaps_smo_free(&S);
  lldo_egemaps_levels_free(&L);
  return T60 + 1;
}

/* ------------------------------------------------------------------ the functionals level (88 values) */
/* The cFunctionals instances of GeMAPSv01b_core.func.conf.inc and eGeMAPSv02_core.func.conf.inc as specs of the general
 * restatement (lld_oracle_funcspec.c): "F0" / "Loudness" ([gemapsv01b_functionalsF0] / [..Loudness]: Moments amean +
 * stddevNorm, Percentiles 20/50/80 + range 0-2, Peaks2 rising / falling slope mean + stddev in seconds), "MVZ"
 * ([egemapsv02_functionalsMVR]), "MVV" ([egemapsv02_functionalsMVRVoiced]), "MU" ([egemapsv02_functionalsMeanUV]), "numPeaks"
 * ([gemapsv01b_temporalLoudness]), "segF0" / "segF0pause" ([gemapsv01b_temporalF0] / [..F0p]: Segments nonX / eqX, X = 0,
 * maxNumSeg 1000, seconds), "leq" ([egemapsv02_leqLin]: Means amean). Returns 0 for an unknown name. */
int lldo_funcspec_egemaps(const char *inst, lldo_func_spec *s)
{
  memset(s, 0, sizeof(*s));
  s->period = 0.01;
  s->ext_norm = s->means_norm = s->times_norm = s->seg_norm = s->pk_norm = s->reg_centroid_norm = LLDO_NORM_SEGMENT;
  s->seg_max_num = 20; s->seg_min_lng = 3; s->seg_pause_min_lng = 2; s->lpc_order = 5;
  if (!strcmp(inst, "F0") || !strcmp(inst, "Loudness")) {
    s->n_fam = 3; s->fam[0] = LLDO_FAM_MOMENTS; s->fam[1] = LLDO_FAM_PERCENTILES; s->fam[2] = LLDO_FAM_PEAKS2;
    s->non_zero_functs = !strcmp(inst, "F0") ? 1 : 0;
    s->mom_mask = (1u << 4) | (1u << 5); s->mom_stddev_norm = 2;
    s->pct_interp = 1; s->n_pctl = 3; s->pctl[0] = 0.20; s->pctl[1] = 0.50; s->pctl[2] = 0.80;
    s->n_range = 1; s->range_a[0] = 0; s->range_b[0] = 2;
    s->pk_mask = (1u << 22) | (1u << 25) | (1u << 26) | (1u << 29);
    s->pk_norm = LLDO_NORM_SECOND; s->pk_rel_thresh = (float)0.1;
  } else if (!strcmp(inst, "MVZ") || !strcmp(inst, "MVV")) {
    s->n_fam = 1; s->fam[0] = LLDO_FAM_MOMENTS;
    s->non_zero_functs = !strcmp(inst, "MVV") ? 1 : 0;
    s->mom_mask = (1u << 4) | (1u << 5); s->mom_stddev_norm = 2;
  } else if (!strcmp(inst, "MU")) {
    s->n_fam = 1; s->fam[0] = LLDO_FAM_MOMENTS; s->non_zero_functs = 1; s->mom_mask = 1u << 4;
  } else if (!strcmp(inst, "numPeaks")) {
    s->n_fam = 1; s->fam[0] = LLDO_FAM_PEAKS2; s->pk_mask = 1u; s->pk_norm = LLDO_NORM_SECOND; s->pk_rel_thresh = (float)0.1;
    s->pk_ratio_limit = 1;
  } else if (!strcmp(inst, "segF0") || !strcmp(inst, "segF0pause")) {
    const int pause = !strcmp(inst, "segF0pause");
    s->n_fam = 1; s->fam[0] = LLDO_FAM_SEGMENTS;
    s->seg_mask = pause ? ((1u << 1) | (1u << 4)) : ((1u << 0) | (1u << 1) | (1u << 4));
    s->seg_norm = LLDO_NORM_SECOND; s->seg_algo = pause ? LLDO_SEG_EQX : LLDO_SEG_NONX; s->seg_max_num = 1000;
    s->seg_min_lng = 3; s->seg_auto_min_lng = 1; s->seg_pause_min_lng = 2; s->seg_x = 0.0f;
  } else if (!strcmp(inst, "leq")) {
    s->n_fam = 1; s->fam[0] = LLDO_FAM_MEANS; s->means_mask = 1u;
  } else return 0;
  return 1;
}

/* Rows each instance summarises (its first end-of-input tick decides, winToVecProcessor.cpp:504-528, 868-1098;
 * measured against the binary): instances on 20 ms levels T20 of the T20 + 1 rows; instances that follow the Viterbi
 * smoother max(1, T60 - P); [egemapsv02_functionalsMVRVoiced], which also waits for cPitchJitter, T60 - P, or all T60
 * when P == T60. out88 in [funcconcat]'s order: F0 (10), Loudness (10), MeanStddevZ (10), MeanStddevVoiced (46),
 * MeanUnvoiced (5), temporalSet (6: loudnessPeaksPerSec, VoicedSegmentsPerSec, MeanVoicedSegmentLengthSec,
 * StddevVoicedSegmentLengthSec, MeanUnvoicedSegmentLength, StddevUnvoicedSegmentLength), equivalentSoundLevel_dBp (1).
 * Returns 1, or 0 when the input has no 60 ms frame (the reference then writes no functionals vector). */
int lldo_egemaps_func_from_levels(const lldo_egemaps_lv *L, const lldo_egemaps_smo *S, float *out88)
{
  const long T20 = L->T20, T = L->T60, P = L->P;
  if (T < 1) return 0;
  const long rV = (T - P) > 1 ? (T - P) : 1, rJ = (P >= T) ? T : T - P;
  lldo_func_spec sp;
  float *o = out88, tmp[8];
  lldo_funcspec_egemaps("F0", &sp); lldo_funcspec_apply(&sp, S->logf0, 1, rV, 1, o); o += 10;
  lldo_funcspec_egemaps("Loudness", &sp); lldo_funcspec_apply(&sp, S->loud, 1, T20, 1, o); o += 10;
  lldo_funcspec_egemaps("MVZ", &sp); lldo_funcspec_apply(&sp, S->NoZ, 5, T20, 5, o); o += 10;
  lldo_funcspec_egemaps("MVV", &sp);
  lldo_funcspec_apply(&sp, S->NoNz, 14, rJ, 14, o); o += 28;
  lldo_funcspec_apply(&sp, S->specV, 9, rJ, 9, o); o += 18;
  lldo_funcspec_egemaps("MU", &sp); lldo_funcspec_apply(&sp, S->specU, 5, rV, 5, o); o += 5;
  lldo_funcspec_egemaps("numPeaks", &sp); lldo_funcspec_apply(&sp, S->loud, 1, T20, 1, o); o += 1;
  lldo_funcspec_egemaps("segF0", &sp); lldo_funcspec_apply(&sp, S->logf0, 1, rV, 1, o); o += 3;
  lldo_funcspec_egemaps("segF0pause", &sp); lldo_funcspec_apply(&sp, S->logf0, 1, rV, 1, o); o += 2;
  lldo_funcspec_egemaps("leq", &sp); lldo_funcspec_apply(&sp, L->energy2, 1, T20, 1, tmp);
  {                                                                  /* [egemapsv02_leq] cVectorOperation dBp, vectorOperation.cpp:507-516 */
    const float factor = (float)(10.0 / log(10.0)), logfloor = (float)0.000000000001;
    *o++ = (tmp[0] > logfloor) ? factor * logf(tmp[0]) : factor * logf(logfloor);
  }
  return 1;
}

int lldo_egemaps_func(const int16_t *pcm, long n_samples, float *out88)
{
  const long T60 = lldo_num_frames(n_samples, lround(0.060 * lldo_get_sample_rate()), lround(0.010 * lldo_get_sample_rate()));
  if (T60 < 1) return 0;
  lldo_egemaps_lv L;
  lldo_egemaps_smo S;
  lldo_egemaps_levels(pcm, n_samples, &L);
  lldo_egemaps_smooth(&L, &S);
  const int r = lldo_egemaps_func_from_levels(&L, &S, out88);
  lldo_egemaps_smo_free(&S);
  lldo_egemaps_levels_free(&L);
  return r;
}
