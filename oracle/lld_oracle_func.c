/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's functionals stage for the
 * option set of config/is09-13/IS09_emotion_core.func.conf.inc:
 * cFunctionals in frameMode=full over the LLD level, Extremes / Regression (linear) / Moments.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 *
 * Follows, in the reference's own operation order and accumulator types:
 *   cFunctionals::doProcess          src/functionals/functionals.cpp:284-330 (min, max, mean)
 *   cFunctionalExtremes::process     src/functionals/functionalExtremes.cpp:92-134 (norm = frame)
 *   cFunctionalRegression::process   src/functionals/functionalRegression.cpp:140-425
 *                                    (normInputs = normRegCoeff = doRatioLimit = 0, linear part)
 *   cFunctionalMoments::process      src/functionals/functionalMoments.cpp:88-165
 * Output order = functionalsEnabled order (Extremes;Regression;Moments), element-major
 * (functionals.cpp:233-262, winToVecProcessor output = Mu values per input element).
 * Pinned bit-exact against the real binary's func level (tests/test_oracle_pin_func.py). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include "lld_oracle.h"

int lldo_functionals_count(uint32_t mask)
{
  int n = 0;
  for (int i = 0; i < LLDO_FUNC_NBITS; i++) n += (mask >> i) & 1u;
  return n;
}

/* x: rows x cols (leading dimension ld); out: cols x count(mask). Returns values per column,
 * 0 if rows <= 0 (the reference then emits no vector at all, functionals.cpp:289-292). */
int lldo_functionals(const float *x, int64_t ld, int64_t rows, int cols, uint32_t mask, float *out)
{
  const int per = lldo_functionals_count(mask);
  if (rows <= 0) return 0;
  float *col = (float *)malloc(sizeof(float) * (size_t)rows);
  for (int c = 0; c < cols; c++) {
    for (int64_t t = 0; t < rows; t++) col[t] = x[t * ld + c];
    const long NN = (long)rows;
    /* functionals.cpp:309-318 */
    const float *p = col;
    float min = *p, max = *p;
    double mean = *p;
    const float *pE = col + NN;
    while (++p < pE) {
      if (*p < min) min = *p;
      if (*p > max) max = *p;
      mean += (double)*p;
    }
    mean /= (double)NN;
    const float meanf = (float)mean;
    float *o = out + (size_t)c * (size_t)per;
    int n = 0;
    /* ---- Extremes */
    {
      long minpos = -1, maxpos = -1;
      for (long i = 0; i < NN; i++) {
        if ((col[i] == max) && (maxpos == -1)) maxpos = i;
        if ((col[i] == min) && (minpos == -1)) minpos = i;
      }
      const float maxposD = (float)maxpos, minposD = (float)minpos;
      if (mask & LLDO_FUNC_MAX) o[n++] = max;
      if (mask & LLDO_FUNC_MIN) o[n++] = min;
      if (mask & LLDO_FUNC_RANGE) o[n++] = max - min;
      if (mask & LLDO_FUNC_MAXPOS) o[n++] = maxposD;
      if (mask & LLDO_FUNC_MINPOS) o[n++] = minposD;
      if (mask & LLDO_FUNC_AMEAN) o[n++] = meanf;
      if (mask & LLDO_FUNC_MAXAMEANDIST) o[n++] = max - meanf;
      if (mask & LLDO_FUNC_MINAMEANDIST) o[n++] = meanf - min;
    }
    /* ---- Regression (linear) */
    if (mask & (LLDO_FUNC_LINREGC1 | LLDO_FUNC_LINREGC2 | LLDO_FUNC_LINREGERRA | LLDO_FUNC_LINREGERRQ)) {
      const double Nind = (double)NN;
      double num = 0.0, num2 = 0.0, tmp, ii = 0.0;
      const double asum = (double)meanf * Nind;
      for (long i = 0; i < NN; i++) {
        tmp = (double)col[i] * ii;
        num += tmp;
        tmp *= ii;
        ii += 1.0;
        num2 += tmp;
      }
      (void)num2;
      double m = 0.0, t = 0.0, leq = 0.0, lea = 0.0;
      if (NN > 1) {
        const double NNm1 = (Nind) * (Nind - (double)1.0);
        const double S1 = NNm1 / (double)2.0;
        const double S2 = NNm1 * ((double)2.0 * Nind - (double)1.0) / (double)6.0;
        const double S1dS2 = S1 / S2;
        const double d = (Nind - S1 * S1dS2);
        if (d == 0.0) t = 0.0;
        else t = (asum - num * S1dS2) / d;
        m = (num - t * S1) / S2;
      } else {
        m = 0; t = col[0];
      }
      ii = 0.0;
      for (long i = 0; i < NN; i++) {
        const double e = (double)col[i] - (m * ii + t);
        lea += fabs(e);
        ii += 1.0;
        leq += e * e;
      }
      if (!isfinite(m)) m = 0.0;
      if (!isfinite(t)) t = 0.0;
      if (!isfinite(lea / Nind)) lea = 0.0;
      if (!isfinite(leq / Nind)) leq = 0.0;
      if (mask & LLDO_FUNC_LINREGC1) o[n++] = (float)m;
      if (mask & LLDO_FUNC_LINREGC2) o[n++] = (float)t;
      if (mask & LLDO_FUNC_LINREGERRA) o[n++] = (float)(lea / Nind);
      if (mask & LLDO_FUNC_LINREGERRQ) o[n++] = (float)(leq / Nind);
    }
    /* ---- Moments */
    if (mask & (LLDO_FUNC_VARIANCE | LLDO_FUNC_STDDEV | LLDO_FUNC_SKEWNESS | LLDO_FUNC_KURTOSIS | LLDO_FUNC_AMEAN_M)) {
      double m2 = 0.0, m3 = 0.0, m4 = 0.0;
      const double Nind = (double)NN;
      const double meanD = (double)meanf;
      for (long i = 0; i < NN; i++) {
        const double tmp = ((double)col[i] - meanD);
        double tmp2 = tmp * tmp;
        m2 += tmp2;
        tmp2 *= tmp;
        m3 += tmp2;
        m4 += tmp2 * tmp;
      }
      m2 /= Nind;
      if (mask & LLDO_FUNC_VARIANCE) o[n++] = (float)m2;
      const double sqm2 = sqrt(m2);
      if (mask & LLDO_FUNC_STDDEV) o[n++] = (m2 > 0.0) ? (float)sqm2 : 0.0f;
      if (mask & LLDO_FUNC_SKEWNESS) o[n++] = (m2 > 0.0) ? (float)(m3 / (Nind * m2 * sqm2)) : 0.0f;
      if (mask & LLDO_FUNC_KURTOSIS) o[n++] = (m2 > 0.0) ? (float)(m4 / (Nind * m2 * m2)) : 0.0f;
      if (mask & LLDO_FUNC_AMEAN_M) o[n++] = meanf;
    }
  }
  free(col);
  return per;
}
