/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's functionals stage for an arbitrary
 * cFunctionals instance: the nine families the ComParE_2016 feature set uses, in the
 * reference's own operation order and accumulator types (FLOAT_DMEM = float; where the reference mixes float
 * and double the same promotions are written out here). Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use this file.
 *
 *   cFunctionals::doProcess          src/functionals/functionals.cpp:320-389 (nonZeroFuncts, sort, min/max/mean)
 *   cFunctionalExtremes::process     src/functionals/functionalExtremes.cpp:91-134
 *   cFunctionalMeans::process        src/functionals/functionalMeans.cpp:115-259
 *   cFunctionalMoments::process      src/functionals/functionalMoments.cpp:88-165
 *   cFunctionalRegression::process   src/functionals/functionalRegression.cpp:142-425
 *   cFunctionalPercentiles::process  src/functionals/functionalPercentiles.cpp:312-417 (+ getInterpPctl 292-310)
 *   cFunctionalTimes::process        src/functionals/functionalTimes.cpp:213-367
 *   cFunctionalSegments::process     src/functionals/functionalSegments.cpp:309-367 (relTh), 656-725 (nonX), 728-799 (eqX), 801-958
 *   cFunctionalLpc::process          src/functionals/functionalLpc.cpp:95-119, smileUtil.c:1560-1630
 *   cFunctionalPeaks2::process       src/functionals/functionalPeaks2.cpp:316-905
 *   smileMath_ratioLimit             src/smileutil/smileUtil.c:586-613
 * Pinned against the functionals level of the real binary on ComParE_2016 (tests/test_oracle_pin_funcspec.py). */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "lld_oracle_funcspec.h"
#include "lld_oracle_modspec.h"

static int popc(uint32_t v) { int n = 0; while (v) { n += (int)(v & 1u); v >>= 1; } return n; }
#define BIT(m, i) (((m) >> (i)) & 1u)

static int fam_count(const lldo_func_spec *s, int fam)
{
  switch (fam) {
    case LLDO_FAM_EXTREMES: return popc(s->ext_mask & 0xffu);
    case LLDO_FAM_MEANS: return popc(s->means_mask & 0x1ffffu);
    case LLDO_FAM_MOMENTS: return popc(s->mom_mask & 0x3fu);
    case LLDO_FAM_REGRESSION: return popc(s->reg_mask & 0x3ffffu);
    case LLDO_FAM_PERCENTILES: return popc(s->pct_mask & 0x3fu) + s->n_pctl + (s->n_pctl > 0 ? s->n_range + s->n_quot : 0);
    case LLDO_FAM_TIMES: return popc(s->times_mask & 0x1fffu) + s->n_ul + s->n_dl;
    case LLDO_FAM_SEGMENTS: return popc(s->seg_mask & 0x1fu);
    case LLDO_FAM_LPC: return (s->lpc_gain ? 1 : 0) + (s->lpc_coeffs ? s->lpc_order - s->lpc_first : 0);
    case LLDO_FAM_PEAKS2: return popc(s->pk_mask);
    case LLDO_FAM_ONSET: return popc(s->ons_mask & 0x1fu);
    case LLDO_FAM_PEAKS: return popc(s->pko_mask & 0x1fu);
    case LLDO_FAM_CROSSINGS: return popc(s->crs_mask & 0x7u);
    case LLDO_FAM_DCT: return s->dct_last >= s->dct_first && s->dct_first >= 0 ? s->dct_last - s->dct_first + 1 : -1;
    case LLDO_FAM_SAMPLES: return s->n_samples >= 1 && s->n_samples <= 8 ? s->n_samples : -1;
    case LLDO_FAM_MODULATION: return s->mod_n_bins >= 1 && s->mod_n_bins <= 128 && s->mod_win_frames >= 33 && s->mod_win_frames <= 1024 && s->mod_step_frames >= 1 ? s->mod_n_bins : -1;
  }
  return -1;
}

int lldo_funcspec_count(const lldo_func_spec *s)
{
  int n = 0;
  if (s->n_fam < 0 || s->n_fam > 12) return -1;
  for (int i = 0; i < s->n_fam; i++) {
    const int c = fam_count(s, s->fam[i]);
    if (c < 0) return -1;
    n += c;
  }
  return n;
}

/* smileUtil.c:586-613 */
static float logistic_f(float x)
{
  const float lim = (float)log(FLT_MAX);
  if (x > lim) return 1.0f;
  else if (x < -lim) return 0.0f;
  return (float)(1.0 / (1.0 + exp(-x)));
}
static float tanh_f(float x) { return 2.0f * logistic_f(2.0f * x) - 1.0f; }
static float ratio_limit(float x, float limit1, float excess)
{
  if (x > limit1) {
    float y = tanh_f((float)((sqrt(x - limit1 + 1.0) - 1.0) / (excess * 0.5))) * excess + limit1;
    return y;
  } else if (x < -limit1) {
    float y = tanh_f((float)(-(sqrt(-1.0 * (x + limit1) + 1.0) - 1.0) / (excess * 0.5))) * excess - limit1;
    return y;
  }
  return x;
}

static int f_extremes(const lldo_func_spec *s, const float *in, float min, float max, float mean, float *out, long Nin)
{
  long minpos = -1, maxpos = -1;
  for (long i = 0; i < Nin; i++) {
    if ((in[i] == max) && (maxpos == -1)) maxpos = i;
    if ((in[i] == min) && (minpos == -1)) minpos = i;
  }
  float maxposD = (float)maxpos, minposD = (float)minpos;
  if (s->ext_norm == LLDO_NORM_SEGMENT) {
    maxposD /= (float)Nin;
    minposD /= (float)Nin;
  } else if (s->ext_norm == LLDO_NORM_SECOND) {
    const float T = (float)s->period;
    if (T != 0.0f) { maxposD *= T; minposD *= T; }
  }
  const uint32_t m = s->ext_mask;
  int n = 0;
  if (BIT(m, 0)) out[n++] = max;
  if (BIT(m, 1)) out[n++] = min;
  if (BIT(m, 2)) out[n++] = max - min;
  if (BIT(m, 3)) out[n++] = maxposD;
  if (BIT(m, 4)) out[n++] = minposD;
  if (BIT(m, 5)) out[n++] = mean;
  if (BIT(m, 6)) out[n++] = max - mean;
  if (BIT(m, 7)) out[n++] = mean - min;
  return n;
}

static int f_means(const lldo_func_spec *s, const float *in, float mean, float *out, long Nin)
{
  double tmp = (double)in[0];
  double fa = fabs(tmp);
  double absmean = fa, qmean = tmp * tmp;
  long nnz;
  double nzamean, nzabsmean, nzqmean, nzgmean;
  double posamean = 0.0, negamean = 0.0, posqmean = 0.0, negqmean = 0.0;
  long nPos = 0, nNeg = 0;
  if (tmp != 0.0) {
    nzamean = tmp; nzabsmean = fa; nzqmean = tmp * tmp; nzgmean = log(fa); nnz = 1;
    if (tmp > 0) { posamean += tmp; posqmean += tmp * tmp; nPos++; }
    else { negamean += tmp; negqmean += tmp * tmp; nNeg++; }
  } else {
    nzamean = nzabsmean = nzqmean = nzgmean = 0.0; nnz = 0;
  }
  for (long i = 1; i < Nin; i++) {
    tmp = (double)in[i];
    fa = fabs(tmp);
    absmean += fa;
    if (tmp > 0) { posamean += tmp; nPos++; }
    if (tmp < 0) { negamean += tmp; nNeg++; }
    const double t0 = tmp;
    if (tmp != 0.0) {
      nzamean += tmp;
      nzabsmean += fa;
      nzgmean += log(fa);
      tmp *= tmp;
      nzqmean += tmp;
      nnz++;
      if (t0 > 0) posqmean += tmp;
      if (t0 < 0) negqmean += tmp;
      qmean += tmp;
    }
  }
  tmp = (double)Nin;
  absmean = absmean / tmp;
  qmean = qmean / tmp;
  if (nnz > 0) {
    tmp = (double)nnz;
    nzamean /= tmp; nzabsmean /= tmp; nzqmean /= tmp; nzgmean /= tmp;
    nzgmean = exp(nzgmean);
  }
  if (nPos > 0) { posamean /= (double)nPos; posqmean /= (double)nPos; }
  if (nNeg > 0) { negamean /= (double)nNeg; negqmean /= (double)nNeg; }
  const uint32_t m = s->means_mask;
  int n = 0;
  if (BIT(m, 0)) out[n++] = (float)mean;
  if (BIT(m, 1)) out[n++] = (float)absmean;
  if (BIT(m, 2)) out[n++] = (float)qmean;
  if (BIT(m, 3)) out[n++] = (float)nzamean;
  if (BIT(m, 4)) out[n++] = (float)nzabsmean;
  if (BIT(m, 5)) out[n++] = (float)nzqmean;
  if (BIT(m, 6)) out[n++] = (float)nzgmean;
  if (BIT(m, 7)) {
    if (s->means_norm == LLDO_NORM_FRAME) out[n++] = (float)nnz;
    else if (s->means_norm == LLDO_NORM_SEGMENT) out[n++] = (float)nnz / (float)Nin;
    else out[n++] = (float)nnz / (float)s->period;
  }
  if (BIT(m, 8)) {
    if (absmean != 0.0) out[n++] = (float)(nzgmean / absmean);
    else out[n++] = 1.0f;
  }
  if (BIT(m, 9)) out[n++] = (float)posamean;
  if (BIT(m, 10)) out[n++] = (float)negamean;
  if (BIT(m, 11)) out[n++] = (float)posqmean;
  if (BIT(m, 12)) out[n++] = (float)sqrt(posqmean);
  if (BIT(m, 13)) out[n++] = (float)negqmean;
  if (BIT(m, 14)) out[n++] = (float)sqrt(negqmean);
  if (BIT(m, 15)) out[n++] = (float)sqrt(qmean);
  if (BIT(m, 16)) out[n++] = (float)sqrt(nzqmean);
  return n;
}

static int f_moments(const lldo_func_spec *s, const float *in, float mean, float *out, long Nin)
{
  double m2 = 0.0, m3 = 0.0, m4 = 0.0;
  const double Nind = (double)Nin, meanD = (double)mean;
  for (long i = 0; i < Nin; i++) {
    const double tmp = ((double)in[i] - meanD);
    double tmp2 = tmp * tmp;
    m2 += tmp2;
    tmp2 *= tmp;
    m3 += tmp2;
    m4 += tmp2 * tmp;
  }
  m2 /= Nind;
  const uint32_t m = s->mom_mask;
  int n = 0;
  if (BIT(m, 0)) out[n++] = (float)m2;
  const double sqm2 = sqrt(m2);
  if (BIT(m, 1)) out[n++] = (m2 > 0.0) ? (float)sqm2 : 0.0f;
  if (BIT(m, 2)) out[n++] = (m2 > 0.0) ? (float)(m3 / (Nind * m2 * sqm2)) : 0.0f;
  if (BIT(m, 3)) out[n++] = (m2 > 0.0) ? (float)(m4 / (Nind * m2 * m2)) : 0.0f;
  if (BIT(m, 4)) out[n++] = (float)mean;
  if (BIT(m, 5)) {
    if (m2 > 0.0) {
      float meanLocal = (s->mom_stddev_norm == 1) ? (float)fabs(mean) : mean;
      if (s->mom_ratio_limit) {
        if (meanLocal != 0.0f) {
          const double v = ratio_limit((float)(sqm2 / (double)meanLocal), 10.0f, 20.0f);
          out[n++] = (float)v;
        } else out[n++] = 20.0f;
      } else {
        double mean1 = (double)meanLocal;
        if (mean1 == 0.0) mean1 = 1.0;
        out[n++] = (float)(sqm2 / mean1);
      }
    } else out[n++] = 0.0f;
  }
  return n;
}

static int f_regression(const lldo_func_spec *s, const float *in, float min, float max, float mean, float *out, long Nin)
{
  const double Nind = (double)Nin;
  double range = max - min, rangeInv;
  if (range <= 0.0) { range = 1.0; rangeInv = 0.0; } else rangeInv = 1.0 / range;
  const int enQreg = (s->reg_mask & 0x3fff0u) != 0;
  double num = 0.0, numAbs = 0.0, num2 = 0.0, num2Abs = 0.0, tmp = 0.0, ii = 0.0, asumAbs = 0.0;
  const double asum = mean * Nind;
  if (s->reg_centroid_abs) {
    for (long i = 0; i < Nin; i++) {
      asumAbs += (double)fabs(in[i]);
      tmp = (double)(fabs(in[i])) * ii;
      numAbs += tmp;
      tmp *= ii;
      num2Abs += tmp;
      tmp = (double)in[i] * ii;
      num += tmp;
      tmp *= ii;
      ii += 1.0;
      num2 += tmp;
    }
  } else {
    for (long i = 0; i < Nin; i++) {
      tmp = (double)in[i] * ii;
      num += tmp;
      tmp *= ii;
      ii += 1.0;
      num2 += tmp;
    }
  }
  (void)num2Abs;
  double centroid;
  if (s->reg_centroid_abs) centroid = (asumAbs != 0.0) ? numAbs / asumAbs : 0.0;
  else centroid = (asum != 0.0) ? num / asum : 0.0;
  if (s->reg_centroid_limit) centroid = (double)ratio_limit((float)centroid, (float)Nind, (float)Nind);
  if (s->reg_centroid_norm == LLDO_NORM_SECOND) centroid *= s->period;
  else if (s->reg_centroid_norm == LLDO_NORM_SEGMENT) centroid /= Nind;

  double m = 0.0, t = 0.0, leq = 0.0, lea = 0.0, a = 0.0, b = 0.0, c = 0.0, qeq = 0.0, qea = 0.0;
  if (Nin > 1) {
    const double NNm1 = (Nind) * (Nind - (double)1.0);
    const double S1 = NNm1 / (double)2.0;
    const double S2 = NNm1 * ((double)2.0 * Nind - (double)1.0) / (double)6.0;
    const double S1dS2 = S1 / S2;
    const double d = (Nind - S1 * S1dS2);
    if (d == 0.0) t = 0.0; else t = (asum - num * S1dS2) / d;
    m = (num - t * S1) / S2;
    const double S3 = S1 * S1;
    const double Nind1 = Nind - (double)1.0;
    const double S4 = S2 * ((double)3.0 * (Nind1 * Nind1 + Nind1) - (double)1.0) / (double)5.0;
    if (enQreg) {
      const double S3S3 = S3 * S3, S2S2 = S2 * S2, S1S2 = S1 * S2, S1S1 = S3;
      const double det = S4 * S2 * Nind + (double)2.0 * S3 * S1S2 - S2S2 * S2 - S3S3 * Nind - S1S1 * S4;
      if (det != 0.0) {
        a = ((S2 * Nind - S1S1) * num2 + (S1S2 - S3 * Nind) * num + (S3 * S1 - S2S2) * asum) / det;
        b = ((S1S2 - S3 * Nind) * num2 + (S4 * Nind - S2S2) * num + (S3 * S2 - S4 * S1) * asum) / det;
        c = ((S3 * S1 - S2S2) * num2 + (S3 * S2 - S4 * S1) * num + (S4 * S2 - S3S3) * asum) / det;
      } else { a = 0.0; b = 0.0; c = 0.0; }
    }
  } else {
    m = 0; t = c = in[0]; a = 0.0; b = 0.0;
  }
  ii = 0.0;
  for (long i = 0; i < Nin; i++) {
    double e = (double)in[i] - (m * ii + t);
    if (s->reg_norm_inputs) e *= rangeInv;
    lea += fabs(e);
    ii += 1.0;
    leq += e * e;
  }
  double rs = 0.0, ls = 0.0, x0 = 0.0, y0 = 0.0, yr = 0.0, yrnn = 0.0, c3nn = 0.0, y0nn = 0.0;
  if (enQreg) {
    ii = 0.0;
    for (long i = 0; i < Nin; i++) {
      double e = (double)in[i] - (a * ii * ii + b * ii + c);
      if (s->reg_norm_inputs) e *= rangeInv;
      qea += fabs(e);
      ii += 1.0;
      qeq += e * e;
    }
    x0 = b / (-2.0 * a);
    if (x0 < -1.0 * Nind) x0 = -Nind;
    if (x0 > Nind) x0 = Nind;
    if (!isfinite(x0)) x0 = Nind;
    y0 = c - b * b / (4.0 * a);
    if (!isfinite(y0)) y0 = 0.0;
    y0nn = y0;
    yrnn = yr = a * (Nind - 1.0) * (Nind - 1.0) + b * (Nind - 1.0) + c;
    if (!isfinite(yr)) { yr = 0.0; yrnn = 0.0; }
    c3nn = c;
  }
  double NOneSec = 1.0;
  if (s->reg_norm_coeff == 2) NOneSec = 1.0 / s->period;
  if (s->reg_ratio_limit) {
    m = ratio_limit((float)m, (float)(range / 10.0), (float)(range / 10.0 + 0.01));
    a = ratio_limit((float)a, (float)(sqrt(range / 10.0)), (float)(sqrt(range / 10.0) + 0.01));
    b = ratio_limit((float)b, (float)(range / 10.0), (float)(range / 10.0 + 0.01));
  }
  if (s->reg_norm_coeff == 1) {
    m *= Nind - 1.0;
    a *= (Nind - 1.0) * (Nind - 1.0);
    b *= Nind - 1.0;
    if (Nind != 1.0) x0 /= Nind - 1.0; else x0 = 0.0;
  } else if (s->reg_norm_coeff == 2) {
    m *= NOneSec;
    a *= NOneSec * NOneSec;
    b *= NOneSec;
    if (NOneSec != 1.0) x0 /= NOneSec; else x0 = 0.0;
  }
  if (s->reg_norm_inputs) {
    m *= rangeInv;
    t = (t - min) * rangeInv;
    a *= rangeInv;
    b *= rangeInv;
    c = (c - min) * rangeInv;
    y0 = (y0 - min) * rangeInv;
    yr = (yr - min) * rangeInv;
  }
  if (enQreg) {
    if (x0 > 0) ls = (y0 - c) / x0;
    if (s->reg_norm_coeff == 1) {
      if (x0 < 1.0) rs = (yr - y0) / (1.0 - x0);
    } else if (s->reg_norm_coeff == 2) {
      const double len_t = (Nind - 1.0) / NOneSec;
      if (x0 < len_t) rs = (yr - y0) / (len_t - x0);
    } else {
      if (x0 < Nind - 1.0) rs = (yr - y0) / (Nind - 1.0 - x0);
    }
  }
  if (!isfinite(m)) m = 0.0;
  if (!isfinite(t)) t = 0.0;
  if (!isfinite(lea / Nind)) lea = 0.0;
  if (!isfinite(leq / Nind)) leq = 0.0;
  if (!isfinite(a)) a = 0.0;
  if (!isfinite(b)) b = 0.0;
  if (!isfinite(c)) { c = 0.0; c3nn = 0.0; }
  if (!isfinite(ls)) ls = 0.0;
  if (!isfinite(rs)) rs = 0.0;
  if (!isfinite(qea / Nind)) qea = 0.0;
  if (!isfinite(qeq / Nind)) qeq = 0.0;
  if (!isfinite(centroid)) centroid = 0.0;
  const uint32_t k = s->reg_mask;
  int n = 0;
  if (BIT(k, 0)) out[n++] = (float)m;
  if (BIT(k, 1)) out[n++] = (float)t;
  if (BIT(k, 2)) out[n++] = (float)(lea / Nind);
  if (BIT(k, 3)) out[n++] = (float)(leq / Nind);
  if (BIT(k, 4)) out[n++] = (float)a;
  if (BIT(k, 5)) out[n++] = (float)b;
  if (BIT(k, 6)) out[n++] = (float)c;
  if (!s->reg_old_buggy_qerr) {
    if (BIT(k, 7)) out[n++] = (float)(qea / Nind);
    if (BIT(k, 8)) out[n++] = (float)(qeq / Nind);
  } else {
    if (BIT(k, 7)) out[n++] = (float)(qea);
    if (BIT(k, 8)) out[n++] = (float)(qeq);
  }
  if (BIT(k, 9)) out[n++] = (float)centroid;
  if (BIT(k, 10)) out[n++] = (float)ls;
  if (BIT(k, 11)) out[n++] = (float)rs;
  if (BIT(k, 12)) out[n++] = (float)x0;
  if (BIT(k, 13)) out[n++] = (float)y0;
  if (BIT(k, 14)) out[n++] = (float)yr;
  if (BIT(k, 15)) out[n++] = (float)y0nn;
  if (BIT(k, 16)) out[n++] = (float)c3nn;
  if (BIT(k, 17)) out[n++] = (float)yrnn;
  return n;
}

static float interp_pctl(double p, const float *sorted, long N)
{
  const double idx = p * (double)(N - 1);
  long i1 = (long)floor(idx), i2 = (long)ceil(idx);
  if (i1 < 0) i1 = 0;
  if (i2 < 0) i2 = 0;
  if (i1 >= N) i1 = N - 1;
  if (i2 >= N) i2 = N - 1;
  if (i1 != i2) {
    const double w1 = idx - (double)i1, w2 = (double)i2 - idx;
    return sorted[i1] * (float)w2 + sorted[i2] * (float)w1;
  }
  return sorted[i1];
}
static long pctl_idx(double p, long N)
{
  long r = (long)round(p * (double)(N - 1));
  if (r < 0) return 0;
  if (r >= N) return N - 1;
  return r;
}

static int f_percentiles(const lldo_func_spec *s, const float *sorted, float *out, long Nin)
{
  float q1, q2, q3;
  if (s->pct_interp) {
    q1 = interp_pctl(0.25, sorted, Nin); q2 = interp_pctl(0.50, sorted, Nin); q3 = interp_pctl(0.75, sorted, Nin);
  } else {
    q1 = sorted[pctl_idx(0.25, Nin)]; q2 = sorted[pctl_idx(0.50, Nin)]; q3 = sorted[pctl_idx(0.75, Nin)];
  }
  const uint32_t m = s->pct_mask;
  int n = 0;
  if (BIT(m, 0)) out[n++] = q1;
  if (BIT(m, 1)) out[n++] = q2;
  if (BIT(m, 2)) out[n++] = q3;
  if (BIT(m, 3)) out[n++] = q2 - q1;
  if (BIT(m, 4)) out[n++] = q3 - q2;
  if (BIT(m, 5)) out[n++] = q3 - q1;
  if (s->n_pctl > 0) {
    const int n0 = n;
    for (int i = 0; i < s->n_pctl; i++)
      out[n++] = s->pct_interp ? interp_pctl(s->pctl[i], sorted, Nin) : sorted[pctl_idx(s->pctl[i], Nin)];
    for (int i = 0; i < s->n_range; i++) {
      const float v = (float)fabs(out[n0 + s->range_b[i]] - out[n0 + s->range_a[i]]);
      out[n++] = v;
    }
    /* :402-411 -- under the test of the RANGE switch, and on the numerator being non-zero; without ranges cFunctionals zero-fills
     * the declared values (functionals.cpp:372-375) */
    for (int i = 0; i < s->n_quot; i++) {
      float v = 0.0f;
      if (s->n_range > 0 && s->quot_a[i] >= 0 && s->quot_b[i] >= 0 && out[n0 + s->quot_a[i]] != 0.0)
        v = ratio_limit(out[n0 + s->quot_a[i]] / out[n0 + s->quot_b[i]], 50.0f, 100.0f);
      out[n++] = v;
    }
  }
  return n;
}

static int f_times(const lldo_func_spec *s, const float *in, float min, float max, float *out, long Nin)
{
  const float Nind = (float)Nin;
  float Norm = Nind, Norm1 = Nind - 1.0f, Norm2 = Nind - 2.0f;
  float T = 1.0f;
  if (s->times_norm == LLDO_NORM_SECOND) {
    T = (float)s->period;
    if (T != 0.0f) {
      if (s->times_buggy_sec_norm) { Norm /= T; Norm1 /= T; Norm2 /= T; }
      else { Norm = 1.0f / T; Norm1 /= Nind * T; Norm2 /= Nind * T; }
    }
  }
  if (s->times_norm == LLDO_NORM_FRAME) { Norm = 1.0f; Norm1 /= Nind; Norm2 /= Nind; }
  const float range = max - min;
  const float l25 = 0.25f * range + min, l50 = 0.50f * range + min, l75 = 0.75f * range + min, l90 = 0.90f * range + min;
  long n25 = 0, n50 = 0, n75 = 0, n90 = 0, nR = 0, nF = 0, nLC = 0, nRC = 0;
  for (long i = 0; i < Nin; i++) {
    if (in[i] <= l25) n25++;
    if (in[i] <= l50) n50++;
    if (in[i] <= l75) n75++;
    if (in[i] <= l90) n90++;
  }
  for (long i = 1; i < Nin; i++) {
    if (in[i - 1] < in[i]) nR++;
    else if (in[i - 1] > in[i]) nF++;
  }
  for (long i = 1; i < Nin - 1; i++) {
    const float a1 = in[i] - in[i - 1], a2 = in[i + 1] - in[i];
    if (a2 < a1) nRC++;
    else if (a1 < a2) nLC++;
  }
  const uint32_t m = s->times_mask;
  int n = 0;
  if (BIT(m, 0)) out[n++] = ((float)(Nin - n25)) / Norm;
  if (BIT(m, 1)) out[n++] = ((float)(n25)) / Norm;
  if (BIT(m, 2)) out[n++] = ((float)(Nin - n50)) / Norm;
  if (BIT(m, 3)) out[n++] = ((float)(n50)) / Norm;
  if (BIT(m, 4)) out[n++] = ((float)(Nin - n75)) / Norm;
  if (BIT(m, 5)) out[n++] = ((float)(n75)) / Norm;
  if (BIT(m, 6)) out[n++] = ((float)(Nin - n90)) / Norm;
  if (BIT(m, 7)) out[n++] = ((float)(n90)) / Norm;
  if (Norm1 != 0.0f) {
    if (BIT(m, 8)) out[n++] = ((float)nR) / Norm1;
    if (BIT(m, 9)) out[n++] = ((float)nF) / Norm1;
  } else {
    if (BIT(m, 8)) out[n++] = 0.0f;
    if (BIT(m, 9)) out[n++] = 0.0f;
  }
  if (Norm2 != 0.0f) {
    if (BIT(m, 10)) out[n++] = ((float)nLC) / Norm2;
    if (BIT(m, 11)) out[n++] = ((float)nRC) / Norm2;
  } else {
    if (BIT(m, 10)) out[n++] = 0.0f;
    if (BIT(m, 11)) out[n++] = 0.0f;
  }
  if (BIT(m, 12)) out[n++] = (s->times_norm == LLDO_NORM_SECOND) ? ((float)(Nin) * T) : (float)Nin;
  for (int j2 = 0; j2 < s->n_ul; j2++) {                   /* second pass, user defined times :347-364 */
    const float lX = (float)(s->ul[j2] * range + min);
    long nX = 0;
    for (long i = 0; i < Nin; i++) if (in[i] > lX) nX++;
    out[n++] = ((float)nX) / Norm;
  }
  for (int j2 = 0; j2 < s->n_dl; j2++) {
    const float lX = (float)(s->dl[j2] * range + min);
    long nX = 0;
    for (long i = 0; i < Nin; i++) if (in[i] <= lX) nX++;
    out[n++] = ((float)nX) / Norm;
  }
  return n;
}

typedef struct { long n, sum, maxl, minl; long *lens; long cap; } seg_acc;
static void seg_add(seg_acc *r, long i, long last)
{
  const long len = i - last;
  if (r->n < r->cap) {
    r->sum += len;
    r->lens[r->n] = len;
    r->n++;
    if (len > r->maxl) r->maxl = len;
    if ((r->minl == 0) || (len < r->minl)) r->minl = len;
  }
}

static int f_segments(const lldo_func_spec *s, const float *in, float min, float max, float amean, float *out, long Nin)
{
  seg_acc r = {0, 0, 0, 0, NULL, s->seg_max_num};
  r.lens = (long *)calloc((size_t)s->seg_max_num + 1, sizeof(long));
  const float range = max - min;
  const int algo = s->seg_algo;
  long segMinLng = s->seg_min_lng;
  if (algo != LLDO_SEG_NONX && algo != LLDO_SEG_EQX && algo != LLDO_SEG_CHX && s->seg_auto_min_lng) {
    segMinLng = Nin / s->seg_max_num - 1;                /* :232-235, :328-331, :388-391 */
    if (segMinLng < 2) segMinLng = 2;
  }
  if (algo == LLDO_SEG_DELTA || algo == LLDO_SEG_DELTA2) {          /* process_SegDelta :227-264, process_SegDelta2 :266-307 */
    const float segThresh = range * s->seg_range_rel_threshold;
    const long ravgLng = (s->seg_ravg_lng > 0) ? s->seg_ravg_lng : Nin / (s->seg_max_num / 2);
    long lastSeg = -segMinLng / 2;
    const int d2 = algo == LLDO_SEG_DELTA2;
    float ravg = d2 ? in[0] : 0.0f, raLast = 0.0f;
    for (long i = d2 ? 1 : 0; i < Nin; i++) {
      ravg += in[i];
      if (i >= ravgLng) ravg -= in[i - ravgLng];
      const float cur = (float)((i + 1 < ravgLng) ? (i + 1) : ravgLng);
      const float ra = ravg / cur;
      const int hit = d2 ? ((in[i - 1] - raLast <= segThresh) && (in[i] - ra > segThresh)) : (in[i] - ra > segThresh);
      if (hit && (i - lastSeg > segMinLng)) { seg_add(&r, i, lastSeg); lastSeg = i; }
      raLast = ra;
    }
  } else if (algo == LLDO_SEG_RELTH || algo == LLDO_SEG_MRELTH || algo == LLDO_SEG_ABSTH || algo == LLDO_SEG_NARELTH ||
             algo == LLDO_SEG_NAMRELTH || algo == LLDO_SEG_NAABSTH) {
    /* process_SegThresh :309-367 (three-frame running average) and process_SegThreshNoavg :369-413 (the contour itself).
     * absTh never reads its thresholds (:179-180 leaves it out): no crossing, no segment. */
    float th[8];
    int nth = (algo == LLDO_SEG_ABSTH) ? 0 : s->seg_n_thresholds;
    for (int j = 0; j < nth; j++) {
      if (algo == LLDO_SEG_RELTH || algo == LLDO_SEG_NARELTH) th[j] = min + range * s->seg_thresholds[j];
      else if (algo == LLDO_SEG_MRELTH || algo == LLDO_SEG_NAMRELTH) th[j] = amean * s->seg_thresholds[j];
      else th[j] = s->seg_thresholds[j];
    }
    long lastSeg = -segMinLng / 2;
    if (algo == LLDO_SEG_RELTH || algo == LLDO_SEG_MRELTH || algo == LLDO_SEG_ABSTH) {
      const long ravgLng = 3;
      float ravg = 0.0f, raLast = 0.0f;
      for (long i = 0; i < Nin; i++) {
        ravg += in[i];
        if (i >= ravgLng) ravg -= in[i - ravgLng];
        const float cur = (float)((i + 1 < ravgLng) ? (i + 1) : ravgLng);
        const float ra = ravg / cur;
        int cross = 0;
        for (int j = 0; j < nth; j++)
          if ((ra > th[j] && raLast <= th[j]) || (ra < th[j] && raLast >= th[j])) cross = 1;
        raLast = ra;
        if (cross && (i - lastSeg > segMinLng)) { seg_add(&r, i, lastSeg); lastSeg = i; }
      }
    } else {
      for (long i = 1; i < Nin; i++) {
        int cross = 0;
        for (int j = 0; j < nth; j++)
          if ((in[i] > th[j] && in[i - 1] <= th[j]) || (in[i] < th[j] && in[i - 1] >= th[j])) cross = 1;
        if (cross && (i - lastSeg > segMinLng)) { seg_add(&r, i, lastSeg); lastSeg = i; }
      }
    }
  } else if (algo == LLDO_SEG_CHX) {                     /* process_SegChX :560-653: segments and pauses alike */
    const float X = s->seg_x_is_rel ? (min + range * s->seg_x) : s->seg_x;
    long segStartIndex = 0, segEndIndex = 0;
    int inSeg = 0, segStart = 0, segEnd = 0;
    for (long i = 0; i < Nin; i++) {
      if (in[i] != X) {
        if (inSeg == 1) {
          segEnd = 0;
          segStart++;
          if (segStart >= s->seg_min_lng) { inSeg = 2; seg_add(&r, segStartIndex - 1, segEndIndex); segStart = 0; }
        } else if (inSeg == 0) {
          segStart++;
          segStartIndex = i;
          inSeg = 1;
        } else if (inSeg == 2) {
          segEnd = 0;
        } else if (inSeg == 3) {
          segStart++;
          if (segStart >= s->seg_min_lng) { inSeg = 2; segEnd = 0; segStart = 0; }
        }
      }
      if (in[i] == X) {
        if (inSeg == 3) {
          segStart = 0;
          segEnd++;
          if (segEnd >= s->seg_min_lng) { inSeg = 0; seg_add(&r, segEndIndex - 1, segStartIndex); segEnd = 0; }
        } else if (inSeg == 2) {
          segEnd++;
          segEndIndex = i;
          inSeg = 3;
        } else if (inSeg == 0) {
          segStart = 0;
        } else if (inSeg == 1) {
          segEnd++;
          if (segEnd >= s->seg_pause_min_lng) { inSeg = 0; segEnd = 0; segStart = 0; }
        }
      }
    }
    if (inSeg == 2) seg_add(&r, segEndIndex - 1, segStartIndex);
    else if (inSeg == 0) seg_add(&r, segStartIndex - 1, segEndIndex);
  } else {                                               /* nonX; eqX (:728-799) is its mirror image */
    const float X = s->seg_x_is_rel ? (min + range * s->seg_x) : s->seg_x;
    const int eq = s->seg_algo == LLDO_SEG_EQX;
    long startIdx = 0, i;
    int inSeg = 0, segStart = 0, segEnd = 0;
    for (i = 0; i < Nin; i++) {
      if (eq ? (in[i] == X) : (in[i] != X)) {
        if (inSeg == 1) {
          segEnd = 0;
          segStart++;
          if (segStart >= s->seg_min_lng) { segStart = 0; inSeg = 2; }
        } else if (inSeg == 0) {
          segStart++;
          startIdx = i;
          inSeg = 1;
        } else if (inSeg == 2) {
          segEnd = 0;
        }
      }
      if (eq ? (in[i] != X) : (in[i] == X)) {
        if (inSeg == 2) {
          segStart = 0;
          segEnd++;
          if (segEnd >= s->seg_pause_min_lng) {
            inSeg = 0;
            seg_add(&r, i - segEnd, startIdx);
            segEnd = 0;
          }
        } else if (inSeg == 1) {
          segEnd++;
          if (segEnd >= s->seg_pause_min_lng) { inSeg = 0; segEnd = 0; segStart = 0; }
        }
      }
    }
    if (inSeg == 2) {
      segEnd++;
      seg_add(&r, i - segEnd, startIdx);
    }
  }
  float lenDev = 0.0f, mean;
  if (r.n > 1) mean = (float)r.sum / ((float)r.n); else mean = (float)r.sum;
  for (long i = 0; i < r.n; i++) lenDev += ((float)r.lens[i] - mean) * ((float)r.lens[i] - mean);
  if (r.n > 1) { lenDev /= (float)r.n; lenDev = (float)sqrt(lenDev); } else lenDev = 0.0f;
  free(r.lens);
  const uint32_t m = s->seg_mask;
  int n = 0;
  if (BIT(m, 0)) {
    if (s->seg_norm == LLDO_NORM_SECOND) {
      const float T = (float)s->period;
      float Norm = 1.0f;
      if (T != 0.0f) Norm = T;
      Norm *= (float)Nin;
      out[n++] = (float)r.n / Norm;
    } else if (s->seg_norm == LLDO_NORM_SEGMENT) out[n++] = (float)r.n / (float)(s->seg_max_num);
    else out[n++] = (float)r.n;
  }
  if (s->seg_norm == LLDO_NORM_SEGMENT) {
    if (BIT(m, 1)) out[n++] = mean / (float)(Nin);
    if (BIT(m, 2)) out[n++] = (float)r.maxl / (float)(Nin);
    if (BIT(m, 3)) out[n++] = (float)r.minl / (float)(Nin);
    if (BIT(m, 4)) out[n++] = lenDev / (float)(Nin);
  } else if (s->seg_norm == LLDO_NORM_FRAME) {
    if (BIT(m, 1)) out[n++] = mean;
    if (BIT(m, 2)) out[n++] = (float)r.maxl;
    if (BIT(m, 3)) out[n++] = (float)r.minl;
    if (BIT(m, 4)) out[n++] = lenDev;
  } else {
    const float T = (float)s->period;
    float Norm = 1.0f;
    if (T != 0.0f) Norm = T;
    if (BIT(m, 1)) out[n++] = mean * Norm;
    if (BIT(m, 2)) out[n++] = (float)r.maxl * Norm;
    if (BIT(m, 3)) out[n++] = (float)r.minl * Norm;
    if (BIT(m, 4)) out[n++] = lenDev * Norm;
  }
  return n;
}

static int f_lpc(const lldo_func_spec *s, const float *in, float *out, long Nin)
{
  const int p = s->lpc_order;
  float acf[34], a[34];
  float gain = 0.0f;
  memset(a, 0, sizeof a);          /* the reference's lpc[] persists between calls; it is fully written below */
  /* smileDsp_autoCorr (n is an int there) */
  {
    int lag = p + 1;
    const int n = (int)Nin;
    while (lag) {
      acf[--lag] = 0.0f;
      for (int i = lag; i < n; i++) acf[lag] += in[i] * in[i - lag];
    }
  }
  /* smileDsp_calcLpcAcf */
  if ((acf[0] == 0.0f) || (acf[0] == -0.0f)) {
    for (int i = 0; i < p; i++) a[i] = 0.0f;
  } else {
    float e = acf[0];
    for (int m = 1; m <= p; m++) {
      float sum = 1.0f * acf[m];
      for (int i = 1; i < m; i++) sum += a[i - 1] * acf[m - i];
      const float k_m = (-1.0f / e) * sum;
      a[m - 1] = k_m;
      for (int i = 1; i <= m / 2; i++) {
        const float x = a[i - 1];
        a[i - 1] += k_m * a[m - i - 1];
        if ((i < (m / 2)) || ((m & 1) == 1)) a[m - i - 1] += k_m * x;
      }
      e *= (1.0f - k_m * k_m);
      if (e == 0.0f) {
        for (int i = m; i <= p; i++) a[i] = 0.0f;
        break;
      }
    }
    gain = e;
  }
  int n = 0;
  if (s->lpc_gain) out[n++] = gain / (float)Nin;
  if (s->lpc_coeffs)
    for (int i = s->lpc_first; i < p; i++) out[n++] = a[i];
  return n;
}

/* ---- Peaks2: the reference's doubly linked list = an array in insertion order + alive flags */
typedef struct { int type; float y; long x; int alive; } mm_el;

static int pk_below(const lldo_func_spec *s, float absThresh, float diff, float base)
{
  if (s->pk_dyn_rel) {
    if (base == 0.0f) return diff != 0.0f ? 1 : 0;
    if (fabs(diff / base) < s->pk_rel_thresh) return 1;
    return 0;
  }
  return diff < absThresh ? 1 : 0;
}
static float pk_rl(const lldo_func_spec *s, float x) { return s->pk_ratio_limit ? ratio_limit(x, 10.0f, 10.0f) : x; }
static float pk_rlmax(const lldo_func_spec *s, float alt) { return s->pk_ratio_limit ? 20.0f : alt; }
static float pk_rlu(const lldo_func_spec *s, float x)
{
  if (s->pk_ratio_limit) {
    if (x > 1.0f) return 1.0f;
    if (x < -1.0f) return -1.0f;
  }
  return x;
}
#define FMIN2(a, b) ((a) < (b) ? (a) : (b))

static int f_peaks2(const lldo_func_spec *s, const float *in, float min, float max, float mean, float *out, long Nin)
{
  const float range = max - min;
  const float absThresh = s->pk_use_abs ? s->pk_abs_thresh : s->pk_rel_thresh * range;
  mm_el *L = (mm_el *)malloc(sizeof(mm_el) * (size_t)(Nin > 0 ? Nin : 1));
  long nl = 0;
  for (long i = 2; i < Nin - 2; i++) {
    if (in[i] > in[i - 1] && in[i] > in[i + 1]) { L[nl].type = 1; L[nl].y = in[i]; L[nl].x = i; L[nl].alive = 1; nl++; }
    else if (in[i] < in[i - 1] && in[i] < in[i + 1]) { L[nl].type = 0; L[nl].y = in[i]; L[nl].x = i; L[nl].alive = 1; nl++; }
  }
  /* pass 1: minimum rise / fall */
  float lastVal = in[0], lastMin = in[0], lastMax = in[0];
  int maxFlag = 0, minFlag = 0;
  long lastMaxPtr = -1, lastMinPtr = -1;
  (void)maxFlag;
  for (long e = 0; e < nl; e++) {
    mm_el *el = &L[e];
    if (el->type == 1) {
      if (pk_below(s, absThresh, (float)fabs(el->y - lastVal), FMIN2(el->y, lastVal))) {
        if (pk_below(s, absThresh, el->y - lastMin, lastMin)) {
          el->alive = 0;
        } else {
          if (el->y > lastMax * 1.05) {
            if (lastMaxPtr != -1) L[lastMaxPtr].alive = 0;      /* may be el itself */
            lastMax = el->y;
            lastMaxPtr = e;
          } else {
            if (minFlag) { lastMax = el->y; lastMaxPtr = e; }
            else el->alive = 0;
          }
          maxFlag = 1; minFlag = 0;
        }
      } else {
        maxFlag = 1; minFlag = 0;
        lastMax = el->y;
        lastMaxPtr = e;
      }
    } else {
      if (!pk_below(s, absThresh, (float)fabs(el->y - lastVal), FMIN2(el->y, lastVal))) {
        minFlag = 1; maxFlag = 0;
        lastMin = el->y;
        lastMinPtr = e;
      }
    }
    lastVal = el->y;
  }
  /* pass 2: minima too close below the last maximum */
  lastMax = in[0];
  for (long e = 0; e < nl; e++) {
    if (!L[e].alive) continue;
    if (L[e].type == 0) {
      if (pk_below(s, absThresh, lastMax - L[e].y, L[e].y)) L[e].alive = 0;
    } else lastMax = L[e].y;
  }
  /* pass 3: alternation of minima and maxima */
  lastMax = in[0]; lastMin = in[0];
  minFlag = 0; maxFlag = 0;
  int init = 1;
  for (long e = 0; e < nl; e++) {
    if (!L[e].alive) continue;
    if (L[e].type == 0) {
      if (!minFlag || init) { lastMin = L[e].y; lastMinPtr = e; minFlag = 1; init = 0; }
      else {
        if (L[e].y >= lastMin) L[e].alive = 0;
        else if (lastMinPtr != e) { L[lastMinPtr].alive = 0; lastMinPtr = e; lastMin = L[e].y; }
      }
    } else {
      if (minFlag || init) { lastMax = L[e].y; lastMaxPtr = e; minFlag = 0; init = 0; }
      else {
        if (L[e].y <= lastMax) L[e].alive = 0;
        else if (lastMaxPtr != e) { L[lastMaxPtr].alive = 0; lastMaxPtr = e; lastMax = L[e].y; }
      }
    }
  }
  float peakMax = 0.0f, peakMin = 0.0f, peakDist = 0.0f, peakDiff = 0.0f, peakStddevDist = 0.0f, peakStddevDiff = 0.0f;
  float peakMean = 0.0f, minMax = 0.0f, minMin = 0.0f, minDist = 0.0f, minDiff = 0.0f, minStddevDist = 0.0f;
  float minStddevDiff = 0.0f, minMean = 0.0f;
  long nPeakDist = 0, nPeaks = 0, nMinDist = 0, nMins = 0;
  lastMaxPtr = -1; lastMinPtr = -1;
  for (long e = 0; e < nl; e++) {
    if (!L[e].alive) continue;
    if (L[e].type == 0) {
      if (lastMinPtr == -1) { lastMinPtr = e; minMin = L[e].y; minMax = L[e].y; }
      else {
        nMinDist++;
        minDist += (float)(L[e].x - L[lastMinPtr].x);
        minDiff += (float)fabs(L[e].y - L[lastMinPtr].y);
        if (minMin > L[e].y) minMin = L[e].y;
        if (minMax < L[e].y) minMax = L[e].y;
        lastMinPtr = e;
      }
      minMean += L[e].y;
      nMins++;
    } else {
      if (lastMaxPtr == -1) { lastMaxPtr = e; peakMin = L[e].y; peakMax = L[e].y; }
      else {
        nPeakDist++;
        peakDist += (float)(L[e].x - L[lastMaxPtr].x);
        peakDiff += (float)fabs(L[e].y - L[lastMaxPtr].y);
        if (peakMin > L[e].y) peakMin = L[e].y;
        if (peakMax < L[e].y) peakMax = L[e].y;
        lastMaxPtr = e;
      }
      peakMean += L[e].y;
      nPeaks++;
    }
  }
  if (nPeaks > 1) {
    peakMean /= (float)nPeaks;
    if (nPeakDist > 1) { peakDist /= (float)nPeakDist; peakDiff /= (float)nPeakDist; }
  }
  if (nMins > 0) {
    minMean /= (float)nMins;
    if (nMinDist > 1) { minDist /= (float)nMinDist; minDiff /= (float)nMinDist; }
  }
  lastMaxPtr = -1; lastMinPtr = -1;
  for (long e = 0; e < nl; e++) {
    if (!L[e].alive) continue;
    if (L[e].type == 0) {
      if (lastMinPtr == -1) lastMinPtr = e;
      else {
        const float dx = (float)(L[e].x - L[lastMinPtr].x), dy = (float)fabs(L[e].y - L[lastMinPtr].y);
        minStddevDist += (dx - minDist) * (dx - minDist);
        minStddevDiff += (dy - minDiff) * (dy - minDiff);
        lastMinPtr = e;
      }
    } else {
      if (lastMaxPtr == -1) lastMaxPtr = e;
      else {
        /* the reference measures against the last MINIMUM here (functionalPeaks2.cpp:594-598); a maximum
         * that is not the first one always has a minimum before it after the alternation pass */
        const long q = lastMinPtr;
        const float dx = (float)(L[e].x - L[q].x), dy = (float)fabs(L[e].y - L[q].y);
        peakStddevDist += (dx - peakDist) * (dx - peakDist);
        peakStddevDiff += (dy - peakDiff) * (dy - peakDiff);
        lastMaxPtr = e;
      }
    }
  }
  if (nPeakDist > 1) { peakStddevDist /= (float)nPeakDist; peakStddevDiff /= (float)nPeakDist; }
  peakStddevDist = (peakStddevDist > 0.0f) ? (float)sqrt(peakStddevDist) : 0.0f;
  peakStddevDiff = (peakStddevDiff > 0.0f) ? (float)sqrt(peakStddevDiff) : 0.0f;
  if (nMinDist > 1) { minStddevDist /= (float)nMinDist; minStddevDiff /= (float)nMinDist; }
  minStddevDist = (minStddevDist > 0.0f) ? (float)sqrt(minStddevDist) : 0.0f;
  minStddevDiff = (minStddevDiff > 0.0f) ? (float)sqrt(minStddevDiff) : 0.0f;

  float meanRisingSlope = 0.0f, meanFallingSlope = 0.0f, minRisingSlope = 0.0f, maxRisingSlope = 0.0f;
  float minFallingSlope = 0.0f, maxFallingSlope = 0.0f, stddevRisingSlope = 0.0f, stddevFallingSlope = 0.0f;
  int nRising = 0, nFalling = 0, lastIsMax = -1;
  if (s->pk_mask & 0xffc00000u) {
    const float T = (float)s->period;
    long lastMaxPos = 0, lastMinPos = 0;
    lastMax = in[0]; lastMin = in[0];
    for (long e = 0; e < nl; e++) {
      if (!L[e].alive) continue;
      if (L[e].type == 0) {
        lastMin = L[e].y; lastMinPos = L[e].x;
        if (lastMinPos - lastMaxPos > 0) {
          const float slope = (lastMax - lastMin) / ((float)(lastMinPos - lastMaxPos) * T);
          meanFallingSlope += slope;
          if (nFalling == 0) { minFallingSlope = slope; maxFallingSlope = slope; }
          else {
            if (slope < minFallingSlope) minFallingSlope = slope;
            if (slope > maxFallingSlope) maxFallingSlope = slope;
          }
          nFalling++; lastIsMax = 0;
        }
      } else {
        lastMax = L[e].y; lastMaxPos = L[e].x;
        if (lastMaxPos - lastMinPos > 0) {
          const float slope = (lastMax - lastMin) / ((float)(lastMaxPos - lastMinPos) * T);
          meanRisingSlope += slope;
          if (nRising == 0) { minRisingSlope = slope; maxRisingSlope = slope; }
          else {
            if (slope < minRisingSlope) minRisingSlope = slope;
            if (slope > maxRisingSlope) maxRisingSlope = slope;
          }
          nRising++; lastIsMax = 1;
        }
      }
    }
    if (lastIsMax == 1) {
      if (Nin - 1 - lastMaxPos > 0) {
        const float slope = (in[Nin - 1] - lastMax) / ((float)(Nin - 1 - lastMaxPos) * T);
        meanFallingSlope += slope;
        if (nFalling == 0) { minFallingSlope = slope; maxFallingSlope = slope; }
        else {
          if (slope < minFallingSlope) minFallingSlope = slope;
          if (slope > maxFallingSlope) maxFallingSlope = slope;
        }
        nFalling++;
      }
    } else if (lastIsMax == 0) {
      if (Nin - 1 - lastMinPos > 0) {
        const float slope = (in[Nin - 1] - lastMin) / ((float)(Nin - 1 - lastMinPos) * T);
        meanRisingSlope += slope;
        if (nRising == 0) { minRisingSlope = slope; maxRisingSlope = slope; }
        else {
          if (slope < minRisingSlope) minRisingSlope = slope;
          if (slope > maxRisingSlope) maxRisingSlope = slope;
        }
        nRising++;
      }
    } else {
      const float slope = (in[Nin - 1] - in[0]) / (float)Nin;
      if (slope > 0) { meanRisingSlope = maxRisingSlope = minRisingSlope = slope; nRising = 1; }
      else if (slope < 0) { meanFallingSlope = maxFallingSlope = minFallingSlope = slope; nFalling = 1; }
    }
    if (nRising > 1) meanRisingSlope /= (float)nRising;
    if (nFalling > 1) meanFallingSlope /= (float)nFalling;
    lastMax = in[0]; lastMaxPos = 0; lastMin = in[0]; lastMinPos = 0;
    for (long e = 0; e < nl; e++) {
      if (!L[e].alive) continue;
      if (L[e].type == 0) {
        lastMin = L[e].y; lastMinPos = L[e].x;
        if (lastMinPos - lastMaxPos > 0) {
          const float slope = (lastMax - lastMin) / ((float)(lastMinPos - lastMaxPos) * T);
          stddevFallingSlope += (slope - meanFallingSlope) * (slope - meanFallingSlope);
        }
      } else {
        lastMax = L[e].y; lastMaxPos = L[e].x;
        if (lastMaxPos - lastMinPos) {
          const float slope = (lastMax - lastMin) / ((float)(lastMaxPos - lastMinPos) * T);
          stddevRisingSlope += (slope - meanRisingSlope) * (slope - meanRisingSlope);
        }
      }
    }
    if (nRising > 1) stddevRisingSlope /= (float)nRising;
    if (nFalling > 1) stddevFallingSlope /= (float)nFalling;
    stddevRisingSlope = (stddevRisingSlope > 0.0f) ? (float)sqrt(stddevRisingSlope) : 0.0f;
    stddevFallingSlope = (stddevFallingSlope > 0.0f) ? (float)sqrt(stddevFallingSlope) : 0.0f;
  }
  free(L);
  if (s->pk_norm == LLDO_NORM_SECOND) {
    const float T = (float)s->period;
    peakDist *= T; peakStddevDist *= T; minDist *= T; minStddevDist *= T;
  } else if (s->pk_norm == LLDO_NORM_SEGMENT) {
    peakDist /= (float)Nin; peakStddevDist /= (float)Nin; minDist /= (float)Nin; minStddevDist /= (float)Nin;
  }
  const uint32_t m = s->pk_mask;
  int n = 0;
  if (BIT(m, 0)) out[n++] = (s->pk_norm == LLDO_NORM_SECOND) ? ((float)nPeaks) / ((float)Nin * (float)s->period) : (float)nPeaks;
  if (BIT(m, 1)) out[n++] = peakDist;
  if (BIT(m, 2)) out[n++] = 0.0f;
  if (BIT(m, 3)) out[n++] = peakStddevDist;
  if (BIT(m, 4)) out[n++] = peakMax - peakMin;
  if (BIT(m, 5)) out[n++] = (range != 0.0f) ? pk_rlu(s, (float)fabs((peakMax - peakMin) / range)) : peakMax - peakMin;
  if (BIT(m, 6)) out[n++] = peakMean;
  if (BIT(m, 7)) out[n++] = peakMean - mean;
  if (BIT(m, 8)) out[n++] = (mean != 0.0f) ? pk_rl(s, peakMean / mean) : pk_rlmax(s, peakMean);
  if (BIT(m, 9)) out[n++] = peakDiff;
  if (BIT(m, 10)) out[n++] = (range != 0.0f) ? pk_rlu(s, peakDiff / range) : peakDiff;
  if (BIT(m, 11)) out[n++] = peakStddevDiff;
  if (BIT(m, 12)) out[n++] = (range != 0.0f) ? pk_rlu(s, peakStddevDiff / range) : peakStddevDiff;
  if (BIT(m, 13)) out[n++] = minMax - minMin;
  if (BIT(m, 14)) out[n++] = (range != 0.0f) ? pk_rlu(s, (float)fabs((minMax - minMin) / range)) : minMax - minMin;
  if (BIT(m, 15)) out[n++] = minMean;
  if (BIT(m, 16)) out[n++] = mean - minMean;
  if (BIT(m, 17)) out[n++] = (mean != 0.0f) ? pk_rl(s, minMean / mean) : pk_rlmax(s, minMean);
  if (BIT(m, 18)) out[n++] = minDiff;
  if (BIT(m, 19)) out[n++] = (range != 0.0f) ? pk_rlu(s, minDiff / range) : minDiff;
  if (BIT(m, 20)) out[n++] = minStddevDiff;
  if (BIT(m, 21)) out[n++] = (range != 0.0f) ? pk_rlu(s, minStddevDiff / range) : minStddevDiff;
  if (BIT(m, 22)) out[n++] = meanRisingSlope;
  if (BIT(m, 23)) out[n++] = maxRisingSlope;
  if (BIT(m, 24)) out[n++] = minRisingSlope;
  if (BIT(m, 25)) out[n++] = stddevRisingSlope;
  if (BIT(m, 26)) out[n++] = meanFallingSlope;
  if (BIT(m, 27)) out[n++] = maxFallingSlope;
  if (BIT(m, 28)) out[n++] = minFallingSlope;
  if (BIT(m, 29)) out[n++] = stddevFallingSlope;
  if (BIT(m, 30)) out[n++] = (meanFallingSlope > 0.0f) ? pk_rl(s, stddevFallingSlope / meanFallingSlope) : 0.0f;
  if (BIT(m, 31)) out[n++] = (meanRisingSlope > 0.0f) ? pk_rl(s, stddevRisingSlope / meanRisingSlope) : 0.0f;
  return n;
}

static int cmp_float(const void *a, const void *b)
{
  const float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}

/* ------------------------------------------------------------------ Onset (functionalOnset.cpp:83-151) */
static int f_onset(const lldo_func_spec *s, const float *in, float *out, long Nin)
{
  long onsetPos = -1, offsetPos = -1, nOnsets = 0, nOffsets = 0;
  int oo = 0;
  if (in[0] > s->ons_thr_on) oo = 1;
  for (long i = 1; i < Nin; i++) {
    float cur;
    if (s->ons_use_abs) cur = (float)fabs(in[i]);
    else cur = in[i];
    if (cur > s->ons_thr_on) {
      if (oo == 0) { nOnsets++; if (onsetPos == -1) onsetPos = i; oo = 1; }
    }
    if (cur <= s->ons_thr_off) {
      if (oo == 1) { nOffsets++; offsetPos = i; oo = 0; }
    }
  }
  if (offsetPos == -1) offsetPos = Nin - 1;
  if (onsetPos == -1) onsetPos = 0;
  const uint32_t m = s->ons_mask;
  int n = 0;
  if (s->ons_norm == LLDO_NORM_SEGMENT) {
    if (BIT(m, 0)) out[n++] = (float)onsetPos / (float)(Nin);
    if (BIT(m, 1)) out[n++] = (float)offsetPos / (float)(Nin);
  } else if (s->ons_norm == LLDO_NORM_SECOND) {
    const float T = (float)s->period;
    if (BIT(m, 0)) out[n++] = (float)onsetPos * T;
    if (BIT(m, 1)) out[n++] = (float)offsetPos * T;
  } else {
    if (BIT(m, 0)) out[n++] = (float)onsetPos;
    if (BIT(m, 1)) out[n++] = (float)offsetPos;
  }
  if (BIT(m, 2)) out[n++] = (float)nOnsets;
  if (BIT(m, 3)) out[n++] = (float)nOffsets;
  if (BIT(m, 4)) { const float T = (float)s->period; out[n++] = (float)nOnsets / ((float)Nin * T); }
  return n;
}

/* ------------------------------------------------------------------ Peaks (functionalPeaks.cpp:98-214), overlapFlag = 1 */
static int f_peaks_old(const lldo_func_spec *s, const float *in, float *out, long Nin)
{
  float max = *in, min = *in, mean = *in;
  float peakDist = (float)0.0, peakMean = (float)0.0;
  long nPeakDist = 0, nPeaks = 0;
  float lastMin = (float)0.0, lastMax = (float)0.0;
  long curmaxPos = 0, lastmaxPos = -1;
  int peakflag = 0;
  long *peakdists = (long *)calloc((size_t)(Nin + 2), sizeof(long));
  long i;
  for (i = 1; i < Nin; i++) {
    if (in[i] < min) min = in[i];
    if (in[i] > max) max = in[i];
    mean += in[i];
  }
  mean /= (float)Nin;
  const float range = max - min;
  float lastlastVal = in[0], lastVal = Nin > 1 ? in[1] : 0.0f;       /* (the reference reads in[1] whatever Nin is) */
  for (i = 2; i < Nin; i++) {
    if ((lastlastVal < lastVal) && (lastVal > in[i])) {
      if (!peakflag) lastMax = in[i];
      else { if (in[i] > lastMax) { lastMax = in[i]; curmaxPos = i; } }
      if (lastMax - lastMin > 0.11 * range) { peakflag = 1; curmaxPos = i; }
    } else {
      if ((lastlastVal > lastVal) && (lastVal < in[i])) lastMin = in[i];
    }
    if ((peakflag) && ((in[i] < lastMax - 0.09 * range) || (i == Nin - 1))) {
      nPeaks++;
      peakMean += lastMax;
      if (lastmaxPos >= 0) {
        const float dist = (float)(curmaxPos - lastmaxPos);
        peakDist += dist;
        peakdists[nPeakDist] = (long)dist;
        nPeakDist++;
      }
      lastmaxPos = curmaxPos;
      peakflag = 0;
    }
    lastlastVal = lastVal;
    lastVal = in[i];
  }
  float stddev = 0.0;
  if (nPeakDist > 0.0) {
    peakDist /= (float)nPeakDist;
    for (i = 0; i < nPeakDist; i++) stddev += (peakdists[i] - peakDist) * (peakdists[i] - peakDist);
    stddev /= (float)nPeakDist;
    stddev = sqrtf(stddev);
  } else {
    peakDist = (float)(Nin + 1);
    stddev = 0.0;
  }
  free(peakdists);
  const uint32_t m = s->pko_mask;
  int n = 0;
  if (BIT(m, 0)) out[n++] = (float)nPeaks;
  if (s->pko_norm == LLDO_NORM_SECOND) { peakDist *= (float)s->period; stddev *= (float)s->period; }
  else if (s->pko_norm == LLDO_NORM_SEGMENT) { peakDist /= (float)Nin; stddev /= (float)Nin; }
  if (BIT(m, 1)) out[n++] = peakDist;
  if (nPeaks > 0.0) peakMean /= (float)nPeaks;
  else peakMean = (float)0.0;
  if (BIT(m, 2)) out[n++] = peakMean;
  if (BIT(m, 3)) out[n++] = peakMean - mean;
  if (BIT(m, 4)) out[n++] = stddev;
  return n;
}

/* ------------------------------------------------------------------ Crossings (functionalCrossings.cpp:66-97) */
static int f_crossings(const lldo_func_spec *s, const float *in, float *out, long Nin)
{
  double amean = 0.0;
  long zcr = 0, mcr = 0, i;
  const uint32_t m = s->crs_mask;
  if (BIT(m, 1) || BIT(m, 2)) {
    amean = (double)*in;
    for (i = 1; i < Nin; i++) amean += in[i];
    amean /= (double)Nin;
  }
  for (i = 1; i < Nin - 1; i++) {
    in++;
    if (((*(in - 1) * *(in + 1) <= 0.0) && (*(in) == 0.0)) || (*(in - 1) * *(in) < 0.0)) zcr++;
    if (BIT(m, 1))
      if ((((*(in - 1) - amean) * (*(in + 1) - amean) <= 0.0) && ((*(in) - amean) == 0.0)) || ((*(in - 1) - amean) * (*(in) - amean) < 0.0)) mcr++;
  }
  int n = 0;
  if (BIT(m, 0)) out[n++] = (float)((double)zcr / (double)Nin);
  if (BIT(m, 1)) out[n++] = (float)((double)mcr / (double)Nin);
  if (BIT(m, 2)) out[n++] = (float)amean;
  return n;
}

/* ------------------------------------------------------------------ DCT (functionalDCT.cpp:84-137): table entry and sum in FLOAT_DMEM */
static int f_dct(const lldo_func_spec *s, const float *in, float *out, long Nin)
{
  const int nCo = s->dct_last - s->dct_first + 1;
  const float factor = (float)sqrt((double)2.0 / (double)(Nin));
  for (int i = 0; i < nCo; i++) {
    out[i] = 0.0;
    for (long m = 0; m < Nin; m++)
      out[i] += in[m] * (float)cos(M_PI * (double)(i + s->dct_first) / (double)(Nin) * ((float)(m) + 0.5));
    out[i] *= factor;
    if (!isfinite(out[i])) out[i] = 0.0;
  }
  return nCo;
}

/* ------------------------------------------------------------------ Samples (functionalSamples.cpp:100-117) */
static int f_samples(const lldo_func_spec *s, const float *in, float *out, long Nin)
{
  const float Nind = (float)Nin;
  for (int k = 0; k < s->n_samples; k++) {
    const int si = (int)((Nind - 1.0) * s->sample_pos[k]);
    out[k] = in[si];
  }
  return s->n_samples;
}

int lldo_funcspec_apply(const lldo_func_spec *s, const float *x, int64_t ld, int64_t rows, int cols, float *out)
{
  const int per = lldo_funcspec_count(s);
  if (per < 0) return -1;
  if (rows <= 0) return 0;
  int need_sorted = 0;
  for (int i = 0; i < s->n_fam; i++) need_sorted |= (s->fam[i] == LLDO_FAM_PERCENTILES);
  float *col = (float *)malloc(sizeof(float) * (size_t)rows);
  float *sorted = (float *)malloc(sizeof(float) * (size_t)rows);
  for (int c = 0; c < cols; c++) {
    long NN = 0;
    if (s->non_zero_functs == 2) {
      for (int64_t t = 0; t < rows; t++) if (x[t * ld + c] > 0.0f) col[NN++] = x[t * ld + c];
    } else if (s->non_zero_functs) {
      for (int64_t t = 0; t < rows; t++) if (x[t * ld + c] != 0.0f) col[NN++] = x[t * ld + c];
    } else {
      for (int64_t t = 0; t < rows; t++) col[NN++] = x[t * ld + c];
    }
    float *o = out + (size_t)c * (size_t)per;
    if (NN <= 0) {                          /* every family returns 0 values -> the driver zero-fills */
      for (int i = 0; i < per; i++) o[i] = 0.0f;
      continue;
    }
    if (need_sorted) {
      memcpy(sorted, col, sizeof(float) * (size_t)NN);
      qsort(sorted, (size_t)NN, sizeof(float), cmp_float);
    }
    float min = col[0], max = col[0];
    double mean = col[0];
    for (long i = 1; i < NN; i++) {
      if (col[i] < min) min = col[i];
      if (col[i] > max) max = col[i];
      mean += (double)col[i];
    }
    mean /= (double)NN;
    const float meanf = (float)mean;
    for (int i = 0; i < s->n_fam; i++) {
      const int want = fam_count(s, s->fam[i]);
      int got = 0;
      switch (s->fam[i]) {
        case LLDO_FAM_EXTREMES: got = f_extremes(s, col, min, max, meanf, o, NN); break;
        case LLDO_FAM_MEANS: got = f_means(s, col, meanf, o, NN); break;
        case LLDO_FAM_MOMENTS: got = f_moments(s, col, meanf, o, NN); break;
        case LLDO_FAM_REGRESSION: got = f_regression(s, col, min, max, meanf, o, NN); break;
        case LLDO_FAM_PERCENTILES: got = f_percentiles(s, sorted, o, NN); break;
        case LLDO_FAM_TIMES: got = f_times(s, col, min, max, o, NN); break;
        case LLDO_FAM_SEGMENTS: got = f_segments(s, col, min, max, mean, o, NN); break;
        case LLDO_FAM_LPC: got = f_lpc(s, col, o, NN); break;
        case LLDO_FAM_PEAKS2: got = f_peaks2(s, col, min, max, meanf, o, NN); break;
        case LLDO_FAM_ONSET: got = f_onset(s, col, o, NN); break;
        case LLDO_FAM_PEAKS: got = f_peaks_old(s, col, o, NN); break;
        case LLDO_FAM_CROSSINGS: got = f_crossings(s, col, o, NN); break;
        case LLDO_FAM_DCT: got = f_dct(s, col, o, NN); break;
        case LLDO_FAM_SAMPLES: got = f_samples(s, col, o, NN); break;
        case LLDO_FAM_MODULATION: {                          /* lld_oracle_modspec.c; an input it does not cover gives NaNs */
          lldo_modspec_cfg mc;
          memset(&mc, 0, sizeof(mc));
          mc.period = s->period; mc.min_freq = s->mod_min_freq; mc.max_freq = s->mod_max_freq;
          mc.win_frames = s->mod_win_frames; mc.step_frames = s->mod_step_frames; mc.n_bins = s->mod_n_bins;
          mc.win_func = s->mod_win_func; mc.remove_nz_mean = s->mod_remove_nz_mean;
          if (!lldo_modspec_apply(&mc, col, NN, o)) for (int q = 0; q < s->mod_n_bins; q++) o[q] = NAN;
          got = s->mod_n_bins;
          break; }
      }
      for (int j = got; j < want; j++) o[j] = 0.0f;
      o += want;
    }
  }
  free(col);
  free(sorted);
  return per;
}
