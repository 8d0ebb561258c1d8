/*
 * lld_oracle_compare.c -- CPU ORACLE, part 3. TEST INFRASTRUCTURE ONLY (see lld_oracle.h).
 *
 * Restatement of the ComParE_2016 LLD groups A and B that lie on the hot path
 * (SURVEY.md 8a rows R8 cPlp as auditory spectrum incl. RASTA, R11 cSpectral with
 * ComParE's option set, R12 cEnergy/cMZcr on the 20 ms / 60 ms frames, cVectorOperation
 * ll1) and of the SMA -> delta tail feeding the LLD sinks. The F0 group (SHS pitch,
 * Viterbi, jitter/shimmer) is out of scope.
 * Citations are relative to the reference root.
 */
#include "lld_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---------------------------------------------------------------------- R11 */
/* Bark scale (Traunmueller), smileDsp_specScaleTransfFwd SPECTSCALE_BARK, smileUtil.c:1123-1137 */
static double hz_to_bark(double x)
{
  if (x > 0) {
    double zz = (26.81 / (1.0 + 1960.0 / x)) - 0.53;
    if (zz < 2) return (0.85 * zz + 0.3);
    else if (zz > 20.1) return (1.22 * zz - 0.22 * 20.1);
    else return zz;
  }
  return 0.0;
}
/* smileDsp_getSharpnessWeightG for a Bark argument, smileUtil.c:1063-1078 */
static double sharp_g(double bark)
{
  if (bark <= 16.0) return 1.0;
  return pow((bark - 16.0) / 4.0, 1.5849625) + 1.0;
}

/* smileStat_entropy, smileUtil.c:2079-2124 */
static float stat_entropy(const float *vals, long N)
{
  const double entropy_floor = 0.0000001;
  double e = 0.0, dn = 0.0;
  float min = 0.0;
  double l2 = (double)log(2.0);
  long i;
  for (i = 0; i < N; i++) { dn += (double)vals[i]; if (vals[i] < min) min = vals[i]; }
  if (min < 0.0) {
    double mf = entropy_floor + min;
    for (i = 0; i < N; i++) { if (vals[i] <= mf) dn += mf - vals[i]; dn -= (double)min; }
  } else {
    min = 0.0;
  }
  if (dn < (float)entropy_floor) dn = (float)entropy_floor;
  for (i = 0; i < N; i++) {
    double v = vals[i] - min, ln;
    if (v <= entropy_floor) v = entropy_floor;
    ln = v / dn;
    if (ln > 0.0) e += ln * (double)log(ln) / l2;
  }
  return (float)(-e);
}

void lldo_spectral_init(lldo_spectral *s, long K, double frame_size_sec)
{
  s->K = K;
  s->fsSec = frame_size_sec;
  s->prev = (float *)calloc((size_t)K, sizeof(float));
  s->have_prev = 0;
  s->frq = (double *)malloc(sizeof(double) * (size_t)K);
  /* frequency axis written by cTransformFFT::generateSpectralVectorInfo (transformFft.cpp:102-117)
   * and copied through cFFTmagphase: frq[i] = F0 * i, F0 = 1/frameSizeSecOut */
  double F0 = 1.0 / frame_size_sec;
  for (long i = 0; i < K; i++) s->frq[i] = F0 * (double)i;
  s->sharp = 0;
}
void lldo_spectral_reset(lldo_spectral *s) { s->have_prev = 0; }
void lldo_spectral_free(lldo_spectral *s) { free(s->prev); free(s->frq); free(s->sharp); }

/* band edge mapping of cSpectral::processVector with a frequency axis, spectral.cpp:779-826 */
static double band_energy(const float *srcP, const double *frq, long Nsrc, long lo, long hi, long nBins)
{
  long ii;
  double wghtL, wghtR, idxL, idxR;
  for (ii = 0; ii < Nsrc; ii++) if (frq[ii] > (double)lo) break;
  if ((ii < Nsrc) && (ii > 0)) wghtL = (frq[ii] - (double)lo) / (frq[ii] - frq[ii - 1]); else wghtL = 1.0;
  idxL = (double)ii - 1.0;
  if (idxL < 0) idxL = 0;
  if (idxL >= Nsrc) idxL = Nsrc;
  if (wghtL == 0.0) wghtL = 1.0;
  for (ii = 0; ii < Nsrc; ii++) if (frq[ii] >= (float)hi) break;
  if ((ii < Nsrc) && (ii > 0)) wghtR = ((double)hi - frq[ii - 1]) / (frq[ii] - frq[ii - 1]); else wghtR = 1.0;
  if ((ii < Nsrc) && (frq[ii] == (float)hi)) idxR = (double)ii; else idxR = (double)ii - 1.0;
  if (idxR >= Nsrc) idxR = Nsrc - 1;
  if (wghtR == 0.0) wghtR = 1.0;
  long iL = (long)floor(idxL), iR = (long)floor(idxR), j;
  if (iL >= Nsrc) { iL = iR = Nsrc - 1; wghtR = 0.0; wghtL = 0.0; }
  if (iR >= Nsrc) { iR = Nsrc - 1; wghtR = 1.0; }
  if (iL < 0) iL = 0;
  if (iR < 0) iR = 0;
  double sum = (double)srcP[iL] * wghtL;                                   /* :832-836 */
  for (j = iL + 1; j < iR; j++) sum += (double)srcP[j];
  sum += (double)srcP[iR] * wghtR;
  return sum / (double)nBins;                                              /* :853 (normBandEnergies=0, lin spectrum) */
}

/* outputs per frame of an option set, in the order cSpectral::processVector writes them (spectral.cpp:770-1545) */
int lldo_spectral_count(const lldo_spectral_opts *o)
{
  return o->n_bands + o->n_slopes + o->n_rolloff + (o->spec_diff != 0) + (o->spec_pos_diff != 0) + (o->flux != 0) + (o->flux_centroid != 0) +
         (o->flux_at_flux_centroid != 0) + (o->centroid != 0) + (o->max_pos != 0) + (o->min_pos != 0) + (o->entropy != 0) +
         (o->standard_deviation != 0) + (o->variance != 0) + (o->skewness != 0) + (o->kurtosis != 0) + (o->slope != 0) +
         (o->sharpness != 0) + (o->harmonicity != 0) + (o->flatness != 0);
}

/* slopes[i] = lo-hi: the slope of the (linear power) spectrum over the band, spectral.cpp:872-985, frequency axis given
 * (nScale >= Nsrc), oldSlopeScale = 1 */
static float band_slope(const float *srcLP, const double *frq, long Nsrc, long lo, long hi)
{
  long ii;
  double wghtL, wghtR, idxL, idxR;
  for (ii = 0; ii < Nsrc; ii++) if (frq[ii] > (double)lo) break;                             /* :886-896 */
  if ((ii < Nsrc) && (ii > 0)) wghtL = (frq[ii] - (double)lo) / (frq[ii] - frq[ii - 1]); else wghtL = 1.0;
  idxL = (double)ii - 1.0;
  if (idxL < 0) idxL = 0;
  if (idxL >= Nsrc) idxL = Nsrc;
  if (wghtL == 0.0) wghtL = 1.0;
  for (ii = 0; ii < Nsrc; ii++) if (frq[ii] >= (float)hi) break;                             /* :912-926 */
  if ((ii < Nsrc) && (ii > 0)) wghtR = ((double)hi - frq[ii - 1]) / (frq[ii] - frq[ii - 1]); else wghtR = 1.0;
  if ((ii < Nsrc) && (frq[ii] == (float)hi)) idxR = (double)ii; else idxR = (double)ii - 1.0;
  if (idxR >= Nsrc) idxR = Nsrc - 1;
  if (wghtR == 0.0) wghtR = 1.0;
  long iL = (long)floor(idxL), iR = (long)floor(idxR);
  if (iL >= Nsrc) { iL = iR = Nsrc - 1; wghtR = 0.0; wghtL = 0.0; }
  if (iR >= Nsrc) { iR = Nsrc - 1; wghtR = 1.0; }
  if (iL < 0) iL = 0;
  if (iR < 0) iR = 0;
  double Sf, S2f, sumA, sumB, Nind = idxR - idxL;                                           /* :944-962 */
  Sf = (double)frq[iL] * wghtL;
  S2f = Sf * Sf;
  sumA = (double)frq[iL] * wghtL * (double)srcLP[iL];
  sumB = wghtL * srcLP[iL];
  for (ii = iL + 1; ii < iR && ii < Nsrc; ii++) {
    S2f += (double)frq[ii] * (double)frq[ii];
    Sf += (double)frq[ii];
    sumA += (double)frq[ii] * (double)srcLP[ii];
    sumB += (double)srcLP[ii];
  }
  S2f += (double)frq[iR] * wghtR * (double)frq[iR] * wghtR;
  Sf += (double)frq[iR] * wghtR;
  sumA += (double)frq[iR] * wghtR * (double)srcLP[iR];
  sumB += wghtR * (double)srcLP[iR];
  double deno = (Nind * S2f - Sf * Sf), slope = 0.0;
  if (deno != 0.0) slope = (Nind * sumA - Sf * sumB) / deno;
  return (float)(slope * (Nind - 1.0));                                                     /* oldSlopeScale = 1 (:979) */
}

/* cSpectral::processVector (spectral.cpp:586-1560) for the descriptor sets of the shipped configurations: any number of bands[]
 * (<= 16), four rollOff points, and flux, centroid, maxPos, minPos, entropy, variance, skewness, kurtosis, slope, sharpness,
 * harmonicity, flatness (logFlatness) each optional, in the reference's output order; defaults squareInput=1, normBandEnergies=0, useLogSpectrum=0,
 * buggyRollOff=0, oldSlopeScale=1, freqRange 0-0. src: magnitude spectrum (K). Returns the number of values written. */
int lldo_spectral_general(lldo_spectral *s, const lldo_spectral_opts *o, const float *src, float *dst)
{
  const long Nsrc = s->K;
  const double *frq = s->frq;
  long i, j, n = 0;
  const long lo = 1, hi = Nsrc - 1;                  /* specRange 0-0 => bins 1 .. Nsrc-1 (:625-627) */
  const long nBins = hi - lo + 1;
  float *srcP = (float *)malloc(sizeof(float) * (size_t)Nsrc);
  for (i = 0; i < Nsrc; i++) srcP[i] = src[i] * src[i];                    /* :676-683 */
  const float *srcM = src, *srcLP = srcP;
  double frameSum = 0.0;
  for (i = lo; i <= hi; i++) frameSum += srcP[i];                          /* :762-767 */
  for (int b = 0; b < o->n_bands; b++) dst[n++] = (float)band_energy(srcP, frq, Nsrc, o->band_lo[b], o->band_hi[b], nBins);
  for (int b = 0; b < o->n_slopes; b++) dst[n++] = band_slope(srcLP, frq, Nsrc, o->slope_lo[b], o->slope_hi[b]);
  double sumB = 0.0, sumC = 0.0;
  for (j = lo; j <= hi; j++) sumB += (double)srcLP[j];                     /* :1093-1097 */
  float ro[16];
  for (i = 0; i < o->n_rolloff; i++) ro[i] = 0;
  for (j = lo; j <= hi; j++) {                                             /* :1102-1117 */
    sumC += (double)srcP[j];
    for (i = 0; i < o->n_rolloff; i++)
      if ((ro[i] == 0.0) && (sumC >= o->rolloff[i] * frameSum)) ro[i] = (float)frq[j];
  }
  for (i = 0; i < o->n_rolloff; i++) dst[n++] = ro[i];
  if (o->spec_pos_diff || o->spec_diff || o->flux || o->flux_centroid || o->flux_at_flux_centroid) {   /* :1124-1254 */
    if (!s->have_prev) {
      dst[n++] = 0.0f;                               /* first frame of a field: ONE zero, however many of the five are on (:1136) */
      s->have_prev = 1;
    } else {
      const float *magP = s->prev;
      double myA = 0.0, myAf = 0.0, d = 0.0, dp = 0.0;
      if (o->spec_diff) {
        for (j = lo; j <= hi; j++) { double myd = (srcM[j] - magP[j - lo]); d += myd * myd; }
        d /= (double)(hi - lo + 1);
        dst[n++] = (d > 0.0) ? (float)sqrt(d) : 0.0f;
      }
      if (o->spec_pos_diff) {
        for (j = lo; j <= hi; j++) { double myd = (srcM[j] - magP[j - lo]); if (myd > 0.0) dp += myd * myd; }
        dp /= (double)(hi - lo + 1);
        dst[n++] = (dp > 0.0) ? (float)sqrt(dp) : 0.0f;
      }
      if (o->flux || o->flux_centroid)
        for (j = lo; j <= hi; j++) {
          double myB = ((double)srcM[j] / 1.0 - (double)magP[j - lo] / 1.0);
          myA += myB * myB;
        }
      if (o->flux_centroid)                          /* (nScale > specRangeUpperBin, frq given: :1178-1182) */
        for (j = lo; j <= hi; j++) {
          double myB = ((double)srcM[j] / 1.0 - (double)magP[j - lo] / 1.0);
          myAf += myB * myB * frq[j];
        }
      if (o->flux) {
        double flux = (nBins > 0) ? myA / (double)nBins : 0.0;
        dst[n++] = (flux > 0.0) ? (float)sqrt(flux) : 0.0f;
      }
      if (o->flux_centroid || o->flux_at_flux_centroid) {
        double fluxCentr = (myA > 0.0) ? myAf / myA : 0.0;
        if (o->flux_centroid) dst[n++] = (float)fluxCentr;
        if (o->flux_at_flux_centroid) {              /* :1209-1247 */
          long bin = hi;
          for (j = lo; j <= hi; j++) if (frq[j] >= fluxCentr) { bin = j; break; }
          long start = bin - 2, end = bin + 2;
          if (start < lo) start = lo;
          if (end > hi) end = hi;
          double myF = 0.0;
          for (j = start; j <= end; j++) {
            double myB = ((double)srcM[j] / 1.0 - (double)magP[j - lo] / 1.0);
            myF += myB * myB;
          }
          if (end - start + 1 > 0) myF /= (double)(end - start + 1); else myF = 0.0;
          dst[n++] = (float)myF;
        }
      }
    }
    for (j = lo; j <= hi; j++) s->prev[j - lo] = srcM[j];
  }
  /* centroid, :1256-1311: computed whenever a moment or the slope needs it */
  float ctr = 0.0f;
  double sumA = 0.0;
  if (o->centroid || o->standard_deviation || o->variance || o->skewness || o->kurtosis || o->slope) {
    for (j = lo; j <= hi; j++) sumA += (double)frq[j] * (double)srcLP[j];
    if (sumB != 0.0) ctr = (float)(sumA / sumB);
    if (o->centroid) dst[n++] = ctr;
  }
  if (o->max_pos || o->min_pos) {                                          /* :1314-1330 (the last bin is not looked at) */
    long maP = lo, miP = lo;
    float max = srcLP[lo], min = srcLP[lo];
    for (j = lo + 1; j < hi; j++) {
      if (srcLP[j] < min) { min = srcLP[j]; miP = j; }
      if (srcLP[j] > max) { max = srcLP[j]; maP = j; }
    }
    if (o->max_pos) dst[n++] = (float)frq[maP];
    if (o->min_pos) dst[n++] = (float)frq[miP];
  }
  if (o->entropy) dst[n++] = stat_entropy(srcLP + lo, hi - lo + 1);        /* :1332-1336 */
  if (o->standard_deviation || o->variance || o->skewness || o->kurtosis) { /* moments, :1338-1397 */
    double u = ctr, m2 = 0.0, m3 = 0.0, m4 = 0.0;
    for (i = lo; i <= hi; i++) {
      double t1 = ((double)frq[i] - u);
      double m = t1 * t1 * (double)srcLP[i];
      m2 += m; m *= t1; m3 += m; m4 += m * t1;
    }
    double sigma2 = (sumB != 0.0) ? m2 / sumB : 0.0;
    if (o->standard_deviation) dst[n++] = (sigma2 > 0.0) ? (float)(sqrt(sigma2)) : 0.0f;
    if (o->variance) dst[n++] = (float)sigma2;
    if (o->skewness) dst[n++] = (sigma2 <= 0.0) ? 0.0f : (float)(m3 / (sumB * sigma2 * sqrt(sigma2)));
    if (o->kurtosis) dst[n++] = (sigma2 == 0.0) ? 0.0f : (float)(m4 / (sumB * sigma2 * sigma2));
  }
  if (o->slope) {                                                          /* :1399-1427 (oldSlopeScale = 1) */
    double Sf = 0.0, S2f = 0.0, Nind = (double)nBins;
    for (i = lo; i <= hi && i < Nsrc; i++) { S2f += (double)frq[i] * (double)frq[i]; Sf += (double)frq[i]; }
    double deno = (Nind * S2f - Sf * Sf), slope = 0.0;
    if (deno != 0.0) slope = (Nind * sumA - Sf * sumB) / deno;
    dst[n++] = (float)(slope * (Nind - 1.0));
  }
  if (o->sharpness) {                                                      /* :1429-1482 (frequency axis given, linear scale) */
    if (!s->sharp) {
      s->sharp = (double *)malloc(sizeof(double) * (size_t)(hi - lo + 1));
      for (j = lo; j <= hi; j++) {
        double f = hz_to_bark((double)frq[j]);      /* lin -> (inv lin = identity) -> bark */
        s->sharp[j - lo] = f * sharp_g(f);
      }
    }
    float sumAA = 0.0f, c2 = 0.0f;
    for (j = lo; j <= hi; j++) sumAA += (float)(s->sharp[j - lo] * (double)srcP[j]);
    if (frameSum != 0.0) c2 = (float)(sumAA / frameSum);
    dst[n++] = (float)(0.11 * c2);
  }
  if (o->harmonicity) {                                                    /* :1484-1513 */
    float ptpSum = 0.0f, lastPeak = -99.0f;
    for (j = lo + 2; j < hi - 1; j++) {
      if ((srcLP[j - 2] < srcLP[j] && srcLP[j - 1] < srcLP[j] && srcLP[j] > srcLP[j + 1] && srcLP[j] > srcLP[j + 2]) ||
          (srcLP[j - 2] > srcLP[j] && srcLP[j - 1] > srcLP[j] && srcLP[j] < srcLP[j + 1] && srcLP[j] < srcLP[j + 2])) {
        if (lastPeak != -99.0) ptpSum += fabs(srcLP[j] - lastPeak);
        lastPeak = srcLP[j];
      }
    }
    ptpSum /= 2.0;
    ptpSum /= (float)nBins;
    dst[n++] = ptpSum;
  }
  if (o->flatness) {                                                       /* :1515-1545: FLOAT_DMEM chain of logf, expf of its mean */
    float sf = 0.0f, gmean = 0.0f;
    int nGm = 0;
    if (sumB != 0.0) {
      for (j = lo; j <= hi; j++)
        if (srcLP[j] != 0.0) { gmean += logf(fabsf(srcLP[j])); nGm++; }
      if (nGm > 0) gmean /= (float)nGm;
      gmean = expf(gmean);
      sf = gmean / (float)fabs(sumB / (double)nBins);
    }
    if (o->log_flatness) dst[n++] = (sf > 0.0) ? (float)logf(sf) : 0.0f;
    else dst[n++] = sf;
  }
  free(srcP);
  /* a field's first frame writes one value for the whole flux family: the values behind it move up and the vector's last slots keep
   * what cVector's calloc put there (dataMemoryLevel.cpp:386-398) */
  const long total = lldo_spectral_count(o);
  while (n < total) dst[n++] = 0.0f;
  return (int)n;
}

/* [is13_spectral]'s options (ComParE_2016_core.lld.conf.inc): bands 250-650, 1000-4000; rollOff .25 .50 .75 .90; flux, centroid,
 * entropy, variance, skewness, kurtosis, slope, sharpness, harmonicity: 15 values */
void lldo_spectral_compare(lldo_spectral *s, const float *src, float *dst)
{
  lldo_spectral_opts o;
  memset(&o, 0, sizeof(o));
  o.n_bands = 2; o.band_lo[0] = 250; o.band_hi[0] = 650; o.band_lo[1] = 1000; o.band_hi[1] = 4000;
  o.n_rolloff = 4; o.rolloff[0] = 0.25; o.rolloff[1] = 0.50; o.rolloff[2] = 0.75; o.rolloff[3] = 0.90;
  o.flux = o.centroid = o.entropy = o.variance = o.skewness = o.kurtosis = o.slope = o.sharpness = o.harmonicity = 1;
  (void)lldo_spectral_general(s, &o, src, dst);
}

/* ---------------------------------------------------------------------- R8 */
/* smileDsp_equalLoudnessWeight, smileUtil.c:1041-1054 */
static double eql_weight(double frequency)
{
  double w = 2.0 * M_PI * frequency;
  double w2 = w * w;
  double c = w2 + 6300000.0;
  if (c > 0.0) return (1e32 * ((w2 + 56.8e6) * w2 * w2) / (c * c * (w2 + 0.38e9) * (w2 * w2 * w2 * w + 1.7e31)));
  return 0.0;
}

/* cPlp::initTables for doAud (+ newRASTA), src/lldcore/plp.cpp:335-402. band_hz: the
 * band-centre metadata cMelspec writes (melspec.cpp:408-412); T: level period (0.01). */
void lldo_plp_init(lldo_plp *p, int n_bands, const double *band_hz, int new_rasta, double T)
{
  p->n_bands = n_bands; p->new_rasta = new_rasta;
  p->melfloor = (float)0.00000000093;
  p->compression = (float)0.33;
  p->eql = (float *)malloc(sizeof(float) * (size_t)n_bands);
  /* RASTA / newRASTA force doLog = doInvLog = 1 (plp.cpp:168-175): the equal-loudness
   * curve is then kept as its log (:349-352) */
  for (int i = 0; i < n_bands; i++) {
    p->eql[i] = (float)eql_weight((double)band_hz[i]);
    if (new_rasta) p->eql[i] = logf(p->eql[i]);
  }
  float lower = (float)1.0, upper = (float)29.0;
  p->rasta_iir = (float)(1.0 - sin(2.0 * M_PI * lower * T));
  float om = (float)cos(2.0 * M_PI * upper * T);
  float norm = (float)sqrt(10.0 * (32.0 * om * om + 8.0));
  p->rasta_fir[0] = (float)(2.0 / norm);
  p->rasta_fir[1] = (float)(-4.0 * om / norm);
  p->rasta_fir[2] = 0.0;
  p->rasta_fir[3] = -p->rasta_fir[1];
  p->rasta_fir[4] = -p->rasta_fir[0];
  p->buf = (float *)calloc((size_t)n_bands * 4, sizeof(float));
  p->init = 0;
}
void lldo_plp_reset(lldo_plp *p) { memset(p->buf, 0, sizeof(float) * (size_t)p->n_bands * 4); p->init = 0; }
void lldo_plp_free(lldo_plp *p) { free(p->eql); free(p->buf); }

/* cPlp::processVector with doLog=0, doAud=1, doInvLog=0, doIDFT=0 ([is13_audspec],
 * [is13_audspecRasta]), plp.cpp:416-593 */
void lldo_plp_audspec(lldo_plp *p, const float *src, float *dst)
{
  int i, N = p->n_bands;
  float *x = (float *)malloc(sizeof(float) * (size_t)N);
  if (p->new_rasta) {                                                       /* doLog, :434-439 */
    for (i = 0; i < N; i++) {
      if (src[i] < p->melfloor) x[i] = logf(p->melfloor);
      else x[i] = (float)logf(src[i]);
    }
  } else {
    for (i = 0; i < N; i++) x[i] = src[i];
  }
  if (p->new_rasta) {                                                       /* :468-485 */
    for (i = 0; i < N; i++) {
      float *b = p->buf + i * 4;
      float out = p->rasta_fir[0] * x[i] + b[0];
      b[0] = p->rasta_fir[1] * x[i] + b[1] + (p->init >= 5) * p->rasta_iir * out;
      b[1] = p->rasta_fir[2] * x[i] + b[2];
      b[2] = p->rasta_fir[3] * x[i] + b[3];
      b[3] = p->rasta_fir[4] * x[i];
      if (p->init >= 5) x[i] = out; else x[i] = 0;
    }
    if (p->init < 5) p->init++;
  }
  if (p->new_rasta) {                                                       /* doAud in the log domain, :490-497 */
    for (i = 0; i < N; i++) x[i] += p->eql[i];
    for (i = 0; i < N; i++) x[i] *= p->compression;
    for (i = 0; i < N; i++) dst[i] = expf(x[i]);                            /* doInvLog, :512-517 */
  } else {
    for (i = 0; i < N; i++) {                                               /* :499-503 */
      if (x[i] < p->melfloor) x[i] = p->melfloor;
      x[i] *= p->eql[i];
    }
    for (i = 0; i < N; i++) dst[i] = (float)pow((double)x[i], (double)p->compression);   /* :505-507 */
  }
  free(x);
}

/* cVectorOperation ll1, src/other/vectorOperation.cpp:475-481 */
static float vec_ll1(const float *src, int N)
{
  float d = 0.0f;
  for (int i = 0; i < N; i++) d += src[i];
  if (N > 0) d /= (float)N;
  return d;
}

/* ------------------------------------------------------------------- chain */
/* ComParE_2016 LLD groups A (4) and B (55) -> SMA(3) -> delta(2), as the LLD sinks see
 * them inside lld;lld_de (columns 6..64 and 71..129 of the 130-column file):
 *   A: audspec_lengthL1norm, audspecRasta_lengthL1norm, pcm_RMSenergy (20 ms raw frame),
 *      pcm_zcr (60 ms raw frame)
 *   B: audSpec_Rfilt[26], spectral[15], mfcc[1..14]
 * Level lengths differ: the 20 ms levels hold T20 frames, the zcr level T60 = T20 - 4.
 * A multi-level reader serves a block only if EVERY level can (dataReader.cpp:446-522)
 * and each level pads with ITS OWN last frame (dataMemoryLevel.cpp:1699-1708), so
 *   - [is13_smoA] emits T60+1 frames; near the end the 20 ms columns still see real
 *     frames while the zcr column sees its replicated last frame;
 *   - [is13_smoB] (20 ms levels only) emits T20+1 frames, of which the sinks keep the
 *     first T60+1: group B is free of end-of-input effects;
 *   - the deltas are taken of those levels; the LLD sinks keep rows = T60+1.
 * out: rows x 118 ([A|B] sma, then their deltas). Requires T60 >= 4 (the lockstep quirk
 * of very short inputs is not restated for this multi-length graph); returns 0 otherwise. */
/* Row T60+1 of group B's own levels (55 sma values, then 55 deltas): [is13_functionalsB] reads lldB_smo;lldB_smo_de,
 * which hold more rows than the T60+1 the LLD sinks keep, and summarises T20-2 = T60+2 of them. Set a destination
 * before calling lldo_compare_ab_chain (NULL switches it off). */
/* config/is09-13/IS13_ComParE.conf instead of compare16/ComParE_2016.conf: the same graph with zeroPadSymmetric = 0 in
 * both cTransformFFT instances and useBrokenJitterThresh = 1 in cPitchJitter */
int g_lldo_is13 = 0;
void lldo_compare_set_is13(int on) { g_lldo_is13 = on ? 1 : 0; }

static float *g_b_extra = NULL;
void lldo_compare_set_b_extra(float *dst110) { g_b_extra = dst110; }

long lldo_compare_ab_chain(const int16_t *pcm, long n_samples, float *out, float *raw59)
{
  lldo_mfcc_cfg c;
  lldo_default_mfcc12_cfg(&c);
  c.frame_size_sec = 0.020; c.preemph_enable = 0; c.zero_pad_symmetric = g_lldo_is13 ? 0 : 1;
  c.lofreq = 20.0f; c.first_mfcc = 1; c.last_mfcc = 14; c.n_delta = 0;
  lldo_geom g;
  lldo_geometry(&c, &g);
  const long N60 = lround(0.060 * c.sample_rate);
  long T20 = lldo_num_frames(n_samples, g.N, g.H);
  long T60 = lldo_num_frames(n_samples, N60, g.H);
  if (T60 < 4) return 0;
  const long rows = T60 + 1;
  if (!out) return rows;
  const int DA = 4, DB = 55, D = 59;
  float *x = (float *)malloc(sizeof(float) * (size_t)n_samples);
  lldo_pcm16_to_float(pcm, n_samples, x);
  double *w = (double *)malloc(sizeof(double) * (size_t)g.N);
  lldo_window_table(LLDO_WIN_HAMM, g.N, 0.4, 1.0, w);
  lldo_mel mel1, mel2; lldo_dct dct;
  lldo_mel_init(&mel1, g.K, g.frame_size_sec_fft, 26, 20.0f, 8000.0f, 1, 0);   /* [is13_melspec1]  htk = 0 */
  lldo_mel_init(&mel2, g.K, g.frame_size_sec_fft, 26, 20.0f, 8000.0f, 1, 1);   /* [is13_melspecMfcc] htk = 1 */
  lldo_mfcc_init(&dct, 26, 1, 14, 22.0f, 1, c.melfloor);
  double band_hz[26];
  for (int m = 1; m <= 26; m++) band_hz[m - 1] = 700.0 * (exp((double)mel1.cfs[m] / 1127.0) - 1.0);   /* melspec.cpp:408-412 */
  lldo_plp plp, plpr;
  lldo_plp_init(&plp, 26, band_hz, 0, c.frame_step_sec);
  lldo_plp_init(&plpr, 26, band_hz, 1, c.frame_step_sec);
  lldo_spectral spec;
  lldo_spectral_init(&spec, g.K, g.frame_size_sec_fft);
  float *fr = (float *)malloc(sizeof(float) * (size_t)g.N);
  float *sp = (float *)malloc(sizeof(float) * (size_t)g.Nfft);
  float *mg = (float *)malloc(sizeof(float) * (size_t)g.K);
  float mb[26], aud[26];
  float *la = (float *)calloc((size_t)T20 * DA, sizeof(float));     /* group A, column 3 valid for t < T60 */
  float *lb = (float *)calloc((size_t)T20 * DB, sizeof(float));
  for (long t = 0; t < T20; t++) {
    const float *src = x + t * g.H;
    float *ra = la + t * DA, *rb = lb + t * DB;
    lldo_window_apply(src, fr, g.N, w, 0.0);
    lldo_rfft_frame(fr, g.N, sp, g.Nfft, c.zero_pad_symmetric);
    lldo_fftmag(sp, g.Nfft, mg);
    lldo_melspec(&mel1, mg, mb);
    lldo_plp_audspec(&plp, mb, aud);
    ra[0] = vec_ll1(aud, 26);                                   /* audspec_lengthL1norm */
    lldo_plp_audspec(&plpr, mb, rb);                            /* audSpec_Rfilt[26] */
    ra[1] = vec_ll1(rb, 26);                                    /* audspecRasta_lengthL1norm */
    ra[2] = lldo_energy_rms(src, g.N);                          /* [is13_energy] on is13_frame25 (raw) */
    if (t < T60) ra[3] = lldo_zcr(src, N60);                    /* [is13_mzcr] on is13_frame60 (raw, 960 samples) */
    lldo_spectral_compare(&spec, mg, rb + 26);
    lldo_melspec(&mel2, mg, mb);
    lldo_mfcc(&dct, mb, rb + 41);
  }
  if (raw59)
    for (long t = 0; t < T60; t++) {
      memcpy(raw59 + t * D, la + t * DA, sizeof(float) * DA);
      memcpy(raw59 + t * D + DA, lb + t * DB, sizeof(float) * DB);
    }
  /* group A: SMA over levels of lengths (T20, T20, T20, T60), then delta of that level */
  const long len[4] = {T20, T20, T20, T60};
  float *sa = (float *)malloc(sizeof(float) * (size_t)rows * DA);
  for (long t = 0; t < rows; t++)
    for (int d = 0; d < DA; d++) {
      long i0 = t, im = t - 1, ip = t + 1;
      if (im < 0) im = 0;
      if (i0 > len[d] - 1) i0 = len[d] - 1;
      if (im > len[d] - 1) im = len[d] - 1;
      if (ip > len[d] - 1) ip = len[d] - 1;
      float y = la[i0 * DA + d];                                  /* contourSmoother.cpp:104-111 */
      y += la[im * DA + d];
      y += la[ip * DA + d];
      y /= (float)3;
      sa[t * DA + d] = y;
    }
  float *da = (float *)malloc(sizeof(float) * (size_t)(rows + 2) * DA);
  { int kind = 0, Wv = 2; float *lv = da; lldo_window_chain(sa, rows, DA, 1, &kind, &Wv, &lv); }
  /* group B: uniform lengths */
  int kind[2] = {1, 0}, Wv[2] = {1, 2};
  float *lvb[2];
  lvb[0] = (float *)malloc(sizeof(float) * (size_t)(T20 + 1) * DB);
  lvb[1] = (float *)malloc(sizeof(float) * (size_t)(T20 + 3) * DB);
  lldo_window_chain(lb, T20, DB, 2, kind, Wv, lvb);
  if (g_b_extra) {
    memcpy(g_b_extra, lvb[0] + rows * DB, sizeof(float) * DB);            /* rows = T60+1 < T20+1 */
    memcpy(g_b_extra + DB, lvb[1] + rows * DB, sizeof(float) * DB);
  }
  for (long t = 0; t < rows; t++) {
    float *o = out + t * 2 * D;
    memcpy(o, sa + t * DA, sizeof(float) * DA);
    memcpy(o + DA, lvb[0] + t * DB, sizeof(float) * DB);
    memcpy(o + D, da + t * DA, sizeof(float) * DA);
    memcpy(o + D + DA, lvb[1] + t * DB, sizeof(float) * DB);
  }
  free(sa); free(da); free(lvb[0]); free(lvb[1]);
  free(x); free(w); free(fr); free(sp); free(mg); free(la); free(lb);
  lldo_mel_free(&mel1); lldo_mel_free(&mel2); lldo_mfcc_free(&dct);
  lldo_plp_free(&plp); lldo_plp_free(&plpr); lldo_spectral_free(&spec);
  return rows;
}

/* ------------------------------------------------------------- R8, PLP-CC branch */
/* cPlp::processVector with doAud = doIDFT = doLP = doLpToCeps = 1, htkcompatible = 1, no log / RASTA
 * (config/plp/PLP_0_D_A.conf [plp:cPlp]; src/lldcore/plp.cpp:416-593, tables :276-334):
 * mel bands -> floor 1.0, HTK equal-loudness weights, pow(x, compression) -> IDFT by cosine table
 * (double accumulate) -> Durbin (smileDsp_calcLpcAcf, smileUtil.c:1572-1630) -> cepstra
 * (smileDsp_lpToCeps, :1532-1556) -> lifter; output order c1..c_lpOrder, c0 (HTK). */
static double eql_weight_htk(double frequency)          /* smileDsp_equalLoudnessWeight_htk, smileUtil.c:1056-1062 */
{
  double f2 = (frequency * frequency);
  double fs = f2 / (f2 + 1.6e5);
  return fs * fs * ((f2 + 1.44e6) / (f2 + 9.61e6));
}

static void calc_lpc_acf(const float *r, float *a, int p, float *gain)
{
  int i, m;
  float e, k_m;
  if ((r[0] == 0.0) || (r[0] == -0.0)) { for (i = 0; i < p; i++) a[i] = 0.0; return; }   /* gain stays as the caller set it */
  e = r[0];
  for (m = 1; m <= p; m++) {
    float sum = (float)1.0 * r[m];
    for (i = 1; i < m; i++) sum += a[i - 1] * r[m - i];
    k_m = ((float)-1.0 / e) * sum;
    a[m - 1] = k_m;
    for (i = 1; i <= m / 2; i++) {
      float x = a[i - 1];
      a[i - 1] += k_m * a[m - i - 1];
      if ((i < (m / 2)) || ((m & 1) == 1)) a[m - i - 1] += k_m * x;
    }
    e *= ((float)1.0 - k_m * k_m);
    if (e == 0.0) { for (i = m; i < p; i++) a[i] = 0.0; break; }
  }
  *gain = e;
}

static float lp_to_ceps(const float *lp, int nLp, float lpGain, float *ceps, int firstCC, int lastCC)
{
  if (firstCC < 1) firstCC = 1;
  if (lastCC > nLp) lastCC = nLp;
  for (int n = firstCC; n <= lastCC; n++) {
    double sum = 0;
    for (int i = 1; i < n; i++) sum += (n - i) * lp[i - 1] * ceps[n - i - 1];
    ceps[n - firstCC] = -(lp[n - firstCC] + (float)(sum / (double)n));
  }
  if (lpGain <= 0.0) lpGain = (float)1.0;
  return (float)(-log(1.0 / (double)lpGain));
}

/* band_hz: the band-centre metadata of the mel level; mel: n_bands values; out: lp_order + 1 values */
/* stage 1: the autocorrelation (doIDFT = 1, doLP = 0: lp_order + 1 values, plp.cpp:579-583); stage 2: the LP coefficients (doLP = 1,
 * doLpToCeps = 0: lp_order values, :573-577); stage 3: the cepstra (the HTK PLP-CC mode) */
void lldo_plp_stage(const float *mel, int n_bands, const double *band_hz, int lp_order, float compression, int cep_lifter_i, int stage,
                    float *out);
void lldo_plp_cc(const float *mel, int n_bands, const double *band_hz, int lp_order, float compression, int cep_lifter_i,
                 float *out)
{
  lldo_plp_stage(mel, n_bands, band_hz, lp_order, compression, cep_lifter_i, 3, out);
}
void lldo_plp_stage(const float *mel, int n_bands, const double *band_hz, int lp_order, float compression, int cep_lifter_i, int stage,
                    float *out)
{
  const float melfloor = 1.0f;                            /* htkcompatible forces melfloor = 1.0, plp.cpp:150-160 */
  const int nFreq = n_bands + 2, nAuto = lp_order + 1, nCeps = lp_order + 1, firstCC = 0, lastCC = lp_order;
  float *src = (float *)malloc(sizeof(float) * (size_t)n_bands);
  float *costable = (float *)malloc(sizeof(float) * (size_t)nAuto * (size_t)nFreq);
  float acf[32], lpc[32], ceps[32], sintable[32];
  const float cepLifter = (float)cep_lifter_i;
  int i, m;
  /* initTables, plp.cpp:288-334 */
  float a = (float)M_PI / (float)(nFreq - 1);
  for (i = 0; i < nAuto; i++) {
    int ib = i * nFreq;
    costable[ib] = 1.0;
    for (m = 1; m < (nFreq - 1); m++) costable[m + ib] = (float)(2.0 * cos(a * (double)i * (double)m));
    costable[m + ib] = (float)(cos(a * (double)i * (double)m));
  }
  for (i = firstCC; i <= lastCC; i++) {
    if (cepLifter > 0.0) sintable[i - firstCC] = ((float)1.0 + cepLifter / (float)2.0 * sinf((float)M_PI * ((float)(i)) / cepLifter));
    else sintable[i - firstCC] = 1.0;
  }
  /* doAud, linear domain (:499-507) */
  for (i = 0; i < n_bands; i++) {
    float v = mel[i];
    if (v < melfloor) v = melfloor;
    v *= (float)eql_weight_htk(band_hz[i]);
    src[i] = (float)pow((double)v, (double)compression);
  }
  /* IDFT (:522-532) */
  for (i = 0; i < nAuto; i++) {
    double tmp = (double)costable[i * nFreq] * (double)src[0];
    for (m = 1; m < nFreq - 1; m++) tmp += (double)costable[m + i * nFreq] * (double)src[m - 1];
    tmp += (double)costable[m + i * nFreq] * (double)src[nFreq - 3];
    acf[i] = (float)(tmp / (2.0 * (nFreq - 1)));
  }
  if (stage == 1) { for (i = 0; i < nAuto; i++) out[i] = acf[i]; free(src); free(costable); return; }
  float lpGain = 0.0f;
  for (i = 0; i < 32; i++) { lpc[i] = 0.0f; ceps[i] = 0.0f; }
  calc_lpc_acf(acf, lpc, lp_order, &lpGain);
  if (stage == 2) { for (i = 0; i < lp_order; i++) out[i] = lpc[i]; free(src); free(costable); return; }
  if (lpGain <= 0) lpGain = (float)1.0;
  float zeroth = lp_to_ceps(lpc, lp_order, lpGain, ceps, firstCC, lastCC);
  ceps[nCeps - 1] = zeroth;                               /* htkcompatible && firstCC == 0 */
  for (i = firstCC; i <= lastCC; i++) {
    int i0 = i - firstCC, i1 = (i == lastCC) ? 0 : i0 + 1;
    out[i0] = (cepLifter > 0.0) ? ceps[i0] * sintable[i1] : ceps[i0];
  }
  free(src); free(costable);
}

/* config/plp/PLP_0_D_A.conf: the MFCC12_0_D_A front end (R0-R6) -> cPlp -> delta -> accel; T x 18 */
long lldo_plp_chain(const int16_t *pcm, long n_samples, float *out)
{
  lldo_mfcc_cfg c;
  lldo_default_mfcc12_cfg(&c);
  lldo_geom g;
  lldo_geometry(&c, &g);
  const long T = lldo_num_frames(n_samples, g.N, g.H);
  if (!out || T <= 0) return T;
  const int D = 6;
  c.n_delta = 0;
  float *mfcc = (float *)malloc(sizeof(float) * (size_t)T * 13);
  float *mel = (float *)malloc(sizeof(float) * (size_t)T * 26);
  lldo_mfcc_chain(&c, pcm, n_samples, mfcc, NULL, NULL, NULL, mel);
  lldo_mel mb;
  lldo_mel_init(&mb, g.K, g.frame_size_sec_fft, 26, c.lofreq, c.hifreq, c.use_power, c.mel_htk_compatible);
  double band_hz[26];
  for (int m = 1; m <= 26; m++) band_hz[m - 1] = 700.0 * (exp((double)mb.cfs[m] / 1127.0) - 1.0);   /* melspec.cpp:408-412 */
  float *st = (float *)malloc(sizeof(float) * (size_t)T * D);
  for (long t = 0; t < T; t++) lldo_plp_cc(mel + t * 26, 26, band_hz, 5, (float)0.33, 22, st + t * D);
  /* R13: [static | delta | accel], orders stored order-major by lldo_delta_chain */
  float *tmp = (float *)malloc(sizeof(float) * (size_t)T * D * 2);
  lldo_delta_chain(st, T, D, 2, 2, tmp);
  for (long t = 0; t < T; t++) {
    memcpy(out + t * 3 * D, st + t * D, sizeof(float) * D);
    for (int o = 1; o <= 2; o++) memcpy(out + t * 3 * D + o * D, tmp + ((size_t)(o - 1) * (size_t)T + (size_t)t) * D, sizeof(float) * D);
  }
  lldo_mel_free(&mb);
  free(mfcc); free(mel); free(st); free(tmp);
  return T;
}

/* the [plp:cPlp] level of config/plp/PLP_0_D_A.conf with the chain cut after `stage` (doLP = 0 | doLpToCeps = 0 | as shipped):
 * T x (6 | 5 | 6) */
long lldo_plp_static_stage(const int16_t *pcm, long n_samples, int stage, float *out)
{
  lldo_mfcc_cfg c;
  lldo_default_mfcc12_cfg(&c);
  lldo_geom g;
  lldo_geometry(&c, &g);
  const long T = lldo_num_frames(n_samples, g.N, g.H);
  if (!out || T <= 0) return T;
  const int D = stage == 2 ? 5 : 6;
  c.n_delta = 0;
  float *mfcc = (float *)malloc(sizeof(float) * (size_t)T * 13);
  float *mel = (float *)malloc(sizeof(float) * (size_t)T * 26);
  lldo_mfcc_chain(&c, pcm, n_samples, mfcc, NULL, NULL, NULL, mel);
  lldo_mel mb;
  lldo_mel_init(&mb, g.K, g.frame_size_sec_fft, 26, c.lofreq, c.hifreq, c.use_power, c.mel_htk_compatible);
  double band_hz[26];
  for (int m = 1; m <= 26; m++) band_hz[m - 1] = 700.0 * (exp((double)mb.cfs[m] / 1127.0) - 1.0);   /* melspec.cpp:408-412 */
  for (long t = 0; t < T; t++) lldo_plp_stage(mel + t * 26, 26, band_hz, 5, (float)0.33, 22, stage, out + t * D);
  lldo_mel_free(&mb);
  free(mfcc); free(mel);
  return T;
}

/* cMfcc with inverse = 1 (mfcc.cpp:184-235): cepstra back to a (log) mel spectrum -- inverse liftering, the transposed cosine table
 * (initTables with blocksize = nBands, :142-170) with the 0th coefficient halved, sqrt(2 / nBands), exp() when doLog. src: the
 * lastMfcc - firstMfcc + 1 coefficients in the forward component's output order (HTK mode with firstMfcc = 0: c1 .. cN, c0). */
void lldo_mfcc_inverse(const float *src, int first, int last, int n_bands, float cep_lifter, int htk, int do_log, float *dst)
{
  const int nMfcc = last - first + 1;
  float costable[64 * 64], sintable[64], tmp[64];
  int i, m;
  double fnM = (double)n_bands;
  for (i = first; i <= last; i++) {
    double fi = (double)i;
    for (m = 0; m < n_bands; m++) costable[m + (i - first) * n_bands] = (float)cos((double)M_PI * (fi / fnM) * ((double)(m) + (double)0.5));
  }
  for (i = first; i <= last; i++)
    sintable[i - first] = (cep_lifter > 0.0) ? ((float)1.0 + cep_lifter / (float)2.0 * sinf((float)M_PI * ((float)(i)) / cep_lifter)) : (float)1.0;   /* sin() on a FLOAT_DMEM in C++ is the float overload */
  float factor = (float)sqrt((double)2.0 / (double)(n_bands));
  for (i = first; i <= last; i++) {
    int i0 = i - first, srcIdx = i0;
    if (htk && (first == 0)) srcIdx = (i == 0) ? last : i0 - 1;
    tmp[srcIdx] = src[srcIdx] / (sintable[i0]);
  }
  for (m = 0; m < n_bands; m++) {
    float *outc = dst + m;
    *outc = 0.0;
    for (i = first; i <= last; i++) {
      int i0 = i - first, srcIdx = i0;
      if (htk && (first == 0)) srcIdx = (i == 0) ? last : i0 - 1;
      float correctionfactor = 1.0;
      if (i == 0) correctionfactor = (float)(0.5f);
      *outc += tmp[srcIdx] * costable[m + i0 * n_bands] * correctionfactor * factor;
    }
    if (do_log) *outc = (float)(expf(*outc));            /* exp() on a FLOAT_DMEM in C++ is the float overload */
  }
  (void)nMfcc;
}

/* ------------------------------------------------------------- the other configs of config/mfcc and config/plp */
/* MFCC12_E_D_A, MFCC12_0_D_A_Z, MFCC12_E_D_A_Z, PLP_E_D_A, PLP_0_D_A_Z, PLP_E_D_A_Z next to the _0_D_A pair:
 *   E: cMfcc firstMfcc = 1 (cPlp firstCC = 1) and a cEnergy column appended to the static block -- log energy of the RAW
 *      frame, HTK style (src/lldcore/energy.cpp:152-185 with log = 1, htkcompatible = 1: log(max(1, 32767^2 sum x^2)));
 *      the deltas are taken of [cepstra | energy] ([cat:cVectorConcat] -> ft0 / delta1e, delta2e: same numbers);
 *   Z: cFullinputMean on the cepstra (src/dspcore/fullinputMean.cpp, multiLoopMode = 0, meanNorm = amean): the float
 *      sum of all frames in order, divided by (float)T, subtracted from the static cepstra only; the deltas come from
 *      the un-normalised level, the energy column is not normalised.
 * Column order of every variant: [cepstra (mean-normalised with Z) | E] [their deltas] [their accelerations]. */
long lldo_htk_variant_chain(int plp, int energy, int cms, const int16_t *pcm, long n_samples, float *out)
{
  lldo_mfcc_cfg c;
  lldo_default_mfcc12_cfg(&c);
  if (cms) c.zero_pad_symmetric = 1;          /* the _Z files do not set [fft] zeroPadSymmetric = 0: the default (1) applies */
  lldo_geom g;
  lldo_geometry(&c, &g);
  const long T = lldo_num_frames(n_samples, g.N, g.H);
  const int first = energy ? 1 : 0;
  const int Dc = plp ? (5 + 1 - first) : (12 + 1 - first);    /* cepstra */
  const int D = Dc + (energy ? 1 : 0);
  if (!out || T <= 0) return T;
  float *st = (float *)malloc(sizeof(float) * (size_t)T * D);
  float *x = (float *)malloc(sizeof(float) * (size_t)n_samples);
  lldo_pcm16_to_float(pcm, n_samples, x);
  if (!plp) {
    c.first_mfcc = first; c.last_mfcc = 12; c.n_delta = 0;
    float *cep = (float *)malloc(sizeof(float) * (size_t)T * Dc);
    lldo_mfcc_chain(&c, pcm, n_samples, cep, NULL, NULL, NULL, NULL);
    for (long t = 0; t < T; t++) memcpy(st + t * D, cep + t * Dc, sizeof(float) * Dc);
    free(cep);
  } else {
    c.n_delta = 0;
    float *mfcc = (float *)malloc(sizeof(float) * (size_t)T * 13);
    float *mel = (float *)malloc(sizeof(float) * (size_t)T * 26);
    lldo_mfcc_chain(&c, pcm, n_samples, mfcc, NULL, NULL, NULL, mel);
    lldo_mel mb;
    lldo_mel_init(&mb, g.K, g.frame_size_sec_fft, 26, c.lofreq, c.hifreq, c.use_power, c.mel_htk_compatible);
    double band_hz[26];
    for (int m = 1; m <= 26; m++) band_hz[m - 1] = 700.0 * (exp((double)mb.cfs[m] / 1127.0) - 1.0);
    float cc6[6];
    for (long t = 0; t < T; t++) {
      lldo_plp_cc(mel + t * 26, 26, band_hz, 5, (float)0.33, 22, cc6);      /* c1..c5, c0 */
      memcpy(st + t * D, cc6, sizeof(float) * Dc);                            /* firstCC = 1 drops c0 (the last one) */
    }
    lldo_mel_free(&mb);
    free(mfcc); free(mel);
  }
  if (energy)
    for (long t = 0; t < T; t++) {
      const float *src = x + t * g.H;
      double d = 0.0;
      for (long i = 0; i < g.N; i++) { float tmp = src[i]; d += tmp * tmp; }
      d *= 32767.0 * 32767.0;
      if (d <= 1.0) d = 1.0;
      st[t * D + Dc] = (float)log(d) * 1.0f + 0.0f;
    }
  float *tmp = (float *)malloc(sizeof(float) * (size_t)T * D * 2);
  lldo_delta_chain(st, T, D, 2, 2, tmp);
  float mean[16];
  if (cms) {
    for (int i = 0; i < Dc; i++) mean[i] = st[i];
    for (long t = 1; t < T; t++)
      for (int i = 0; i < Dc; i++) mean[i] += st[t * D + i];
    const float nM = (float)T;
    for (int i = 0; i < Dc; i++) mean[i] /= nM;
  }
  for (long t = 0; t < T; t++) {
    float *o = out + t * 3 * D;
    memcpy(o, st + t * D, sizeof(float) * D);
    if (cms) for (int i = 0; i < Dc; i++) o[i] -= mean[i];
    for (int k = 1; k <= 2; k++) memcpy(o + k * D, tmp + ((size_t)(k - 1) * (size_t)T + (size_t)t) * D, sizeof(float) * D);
  }
  free(st); free(x); free(tmp);
  return T;
}

/* cMelspec with inverse = 1 (src/lldcore/melspec.cpp:466-516; the bank's tables :186-199,217-239,393-447 with the roles swapped:
 * `n_out` spectrum bins to create -- the option nBands --, `n_src` mel bands coming in). Standard (mel, bwMethod lr) bank only: the
 * reference refuses the HFCC / custom-bandwidth banks here (:487-490). tab (optional, 2 + 2 n_out floats' worth): receives nLoF, nHiF,
 * then the n_out weights and the n_out channel numbers (as floats), i.e. what the table-driven device entry is fed with. */
void lldo_melspec_inverse(const float *src, int n_src, int n_out, double frame_size_sec, float lofreq, float hifreq, int use_power, int htk,
                          float *dst, float *tab)
{
  lldo_mel m;
  long n;
  float *s = (float *)malloc(sizeof(float) * (size_t)n_src);
  memset(dst, 0, sizeof(float) * (size_t)n_out);
  if (!lldo_mel_init(&m, n_out, frame_size_sec, n_src, lofreq, hifreq, use_power, htk)) { free(s); return; }
  for (n = 0; n < n_src; n++) {                                  /* :468-479 */
    if (htk) s[n] = use_power ? src[n] / (float)(32767.0 * 32767.0) : src[n] / (float)32767.0;
    else s[n] = src[n];
  }
  for (n = m.nLo; n < (n_out < m.nHi ? n_out : m.nHi); n++) {    /* :492-505 */
    const long mm = m.chan_map[n];
    if (mm > -1) {
      float a = s[mm] * m.coef[n];
      dst[n] += a;
      if (mm < n_src - 1) {
        a = s[mm + 1] * ((float)1.0 - m.coef[n]);
        dst[n] += a;
      }
    }
  }
  if (use_power)                                                 /* :508-514 (sqrt of a float: the C++ overload) */
    for (n = 0; n < n_out; n++) dst[n] = (dst[n] > 0.0) ? sqrtf(dst[n]) : (float)0.0;
  if (tab) {
    tab[0] = (float)m.nLo; tab[1] = (float)m.nHi;
    for (n = 0; n < n_out; n++) { tab[2 + n] = m.coef[n]; tab[2 + n_out + n] = (float)m.chan_map[n]; }
  }
  lldo_mel_free(&m);
  free(s);
}
