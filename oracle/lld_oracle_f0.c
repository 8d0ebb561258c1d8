/*
 * lld_oracle_f0.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY (see lld_oracle.h).
 *
 * the F0 group of ComParE_2016 / GeMAPS,
 *   is13_frame60 -> gauss window -> FFT 1024 -> magnitude
 *     -> cSpecScale (octave axis by cubic-spline resampling, peak enhancement, smoothing, auditory weighting)
 *     -> cPitchShs  (sub-harmonic summation, six candidates, greedy peak picking; base class cPitchBase)
 *     -> cPitchSmootherViterbi (incremental Viterbi over 6 candidates + "unvoiced", 30-frame buffer)
 *     -> cValbasedSelector (frames with 60 ms RMS energy <= 0.001 are zeroed)
 * restated from config/compare16/ComParE_2016_core.lld.conf.inc:11-38,76-164 and the sources cited at
 * each function. Everything the reference does in double is done in double, in the reference's order.
 *
 * Pinning (tests/test_oracle_pin_f0.py): with the reference's own rdft plugged in, every level of this
 * chain is compared with the real SMILExtract's level of the same name (HTK taps, oracle/conf/compare_f0_taps.conf).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "lld_oracle.h"
extern int g_lldo_is13;     /* lld_oracle_compare.c: IS13_ComParE.conf instead of ComParE_2016.conf */

/* ------------------------------------------------------------------ cSpecScale */
/* smileDsp_specScaleTransfFwd, SPECTSCALE_LOG branch (smileUtil.c:1100-1105) */
static double to_log_scale(double x, double base) { return x > 0 ? log(x) / log(base) : 0.0; }

/* cSpecScale::dataProcessorCustomFinalise (specScale.cpp:228-300) for scale=octave, sourceScale=lin, minF=25,
 * maxF=-1, nPointsTarget=0 (-> K), auditoryWeighting=1; spline caches: smileMath_cspline_init
 * (smileUtilSpline.c:139-155), smileMath_csplint_init (:296-342) */
int lldo_specscale_init(lldo_specscale *s, long K, double frame_size_sec_level) { return lldo_specscale_init_ex(s, K, frame_size_sec_level, 25.0); }

/* the same with cSpecScale.minF given (25 in ComParE_2016 / GeMAPS, 20 in IS10_paraling) */
int lldo_specscale_init_ex(lldo_specscale *s, long K, double frame_size_sec_level, double min_f)
{
  memset(s, 0, sizeof(*s));
  s->K = K;
  const double fsSec = (double)(float)frame_size_sec_level;       /* specScale.cpp:186 */
  const double deltaF = 1.0 / fsSec;                              /* :207 */
  const double base = 2.0;
  double minF = min_f, maxF = -1.0;
  const double samplF = deltaF * (double)(K - 1);                 /* :235-238 */
  if (maxF <= minF || maxF > samplF) maxF = samplF;
  const double fmin_t = to_log_scale(minF, base), fmax_t = to_log_scale(maxF, base);
  const double deltaF_t = (fmax_t - fmin_t) / (double)(K - 1);
  s->ft = (double *)malloc(sizeof(double) * (size_t)K);
  s->sigma = (double *)calloc((size_t)K, sizeof(double));
  s->d1 = (double *)calloc((size_t)K, sizeof(double));
  s->d2 = (double *)calloc((size_t)K, sizeof(double));
  s->k = (long *)malloc(sizeof(long) * (size_t)K);
  s->co = (double *)malloc(sizeof(double) * 3 * (size_t)K);
  s->audw = (double *)malloc(sizeof(double) * (size_t)K);
  for (long i = 1; i < K; i++) s->ft[i] = to_log_scale((double)i * deltaF, base);
  s->ft[0] = 2.0 * s->ft[1] - s->ft[2];                           /* :253 */
  const double *x = s->ft;
  for (long i = 1; i < K - 1; i++) {
    s->sigma[i] = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    s->d1[i] = (x[i + 1] - x[i]) * (x[i + 1] - x[i - 1]);
    s->d2[i] = (x[i] - x[i - 1]) * (x[i + 1] - x[i - 1]);
  }
  long hi = 1;
  for (long i = 0; i < K; i++) {
    const double xt = fmin_t + (double)i * deltaF_t;
    if (i == 0 && xt < x[0]) return 0;
    while (hi < K && x[hi] < xt) hi++;
    if (hi == K) return 0;                                        /* the reference disables its output here */
    const long lo = hi - 1;
    s->k[i] = lo;
    const double range = x[hi] - x[lo];
    if (range == 0.0) return 0;
    const double a = (x[hi] - xt) / range, b = 1.0 - a, r2 = range * range / 6.0;
    s->co[3 * i] = a;
    s->co[3 * i + 1] = (a * a * a - a) * r2;
    s->co[3 * i + 2] = (b * b * b - b) * r2;
  }
  const double nOct = log(maxF / minF) / log(2.0);
  const double nPPO = (double)K / nOct;                            /* :281 */
  const double atan_s = nPPO * (log(65.0 / 50.0) / log(2.0)) - 1.0;
  for (long i = 0; i < K; i++) s->audw[i] = 0.5 + atan(3.0 * ((double)i + 1 - atan_s) / nPPO) / M_PI;
  s->meta[0] = (float)minF; s->meta[1] = (float)maxF; s->meta[2] = (float)nOct; s->meta[3] = (float)nPPO;   /* :289-299 */
  s->meta[4] = (float)fmin_t; s->meta[5] = (float)fmax_t; s->meta[6] = 0.0f; s->meta[7] = (float)base;
  return 1;
}

void lldo_specscale_free(lldo_specscale *s)
{
  free(s->ft); free(s->sigma); free(s->d1); free(s->d2); free(s->k); free(s->co); free(s->audw);
  memset(s, 0, sizeof(*s));
}

/* smileDsp_specEnhanceSHS (smileUtil.c:1965-2001): everything further than 2 bins from a local maximum is
 * zeroed BETWEEN consecutive maxima (not before the first / after the last). With exactly one maximum the
 * reference indexes posmax[1] of its zero-initialised list, i.e. treats it as a maximum at 0. */
static void enhance_peaks(double *a, long n, long *pos)
{
  long m = 0;
  if (n < 2) return;
  pos[0] = pos[1] = 0;
  if (a[0] > a[1]) pos[m++] = 0;
  for (long i = 1; i < n - 1; i++)
    if (a[i] > a[i - 1] && a[i] >= a[i + 1]) pos[m++] = i;
  if (a[n - 1] > a[n - 2]) pos[m++] = n - 1;
  if (m == 1) {
    const long p1 = 0;                                             /* calloc'ed posmax[1] */
    for (long j = 0; j <= p1 - 3; j++) a[j] = 0;
    for (long j = p1 + 3; j < n; j++) a[j] = 0;
  } else {
    for (long i = 1; i < m; i++)
      for (long j = pos[i - 1] + 3; j <= pos[i] - 3; j++) a[j] = 0;
  }
}

/* smileDsp_specSmoothSHS (smileUtil.c:2004-2014): (1,2,1)/4 over the ORIGINAL neighbours, last bin untouched */
static void smooth_121(double *a, long n)
{
  double prev = 0.0;
  for (long i = 0; i < n - 1; i++) {
    const double cur = a[i];
    a[i] = (prev + 2.0 * cur + a[i + 1]) / 4.0;
    prev = cur;
  }
}

/* cSpecScale::processVector (specScale.cpp:305-357); natural spline by smileMath_cspline
 * (smileUtilSpline.c:157-212), evaluation by smileMath_csplint (:344-357) */
void lldo_specscale_frame(const lldo_specscale *s, const float *mag, float *dst) { lldo_specscale_frame_ex(s, 7, mag, dst); }

/* flags: 1 specEnhance, 2 specSmooth, 4 auditoryWeighting (without it the spline's values pass as they are, negative ones too) */
void lldo_specscale_frame_ex(const lldo_specscale *s, int flags, const float *mag, float *dst)
{
  const long K = s->K;
  double *y = (double *)malloc(sizeof(double) * (size_t)K * 3);
  double *y2 = y + K, *u = y2 + K;
  long *pos = (long *)malloc(sizeof(long) * (size_t)(K / 2 + 3));
  for (long i = 0; i < K; i++) y[i] = (double)mag[i];
  if (flags & 1) enhance_peaks(y, K, pos);
  if (flags & 2) smooth_121(y, K);
  u[0] = 0.0; y2[0] = 0.0;
  for (long i = 1; i < K - 1; i++) {
    const double sg = s->sigma[i];
    const double p = 1.0 / (sg * y2[i - 1] + 2.0);
    y2[i] = (sg - 1.0) * p;
    const double ut = (y[i + 1] - y[i]) / s->d1[i] - (y[i] - y[i - 1]) / s->d2[i];
    u[i] = p * (6.0 * ut - sg * u[i - 1]);
  }
  y2[K - 1] = (0.0 - 0.0 * u[K - 2]) / (0.0 * y2[K - 2] + 1.0);
  for (long j = K - 2; j >= 0; j--) y2[j] = y2[j] * y2[j + 1] + u[j];
  for (long i = 0; i < K; i++) {
    const double a = s->co[3 * i], b = 1.0 - a, c = s->co[3 * i + 1], d = s->co[3 * i + 2];
    const long k = s->k[i];
    const double o = a * y[k] + b * y[k + 1] + c * y2[k] + d * y2[k + 1];
    float v = (float)o;
    if (flags & 4) { if (v > 0.0) v = (float)((double)v * s->audw[i]); else v = 0.0f; }
    dst[i] = v;
  }
  free(y); free(pos);
}

/* ------------------------------------------------------------------ cPitchShs / cPitchBase */
/* cPitchShs::setupNewNames (pitchShs.cpp:178-204) reading the level meta data of cSpecScale */
void lldo_shs_init(lldo_shs *h, const lldo_specscale *s)
{
  memset(h, 0, sizeof(*h));
  h->N = s->K;
  const float fmin = s->meta[0], fmint = s->meta[4], fmaxt = s->meta[5];
  h->n_octaves = s->meta[2];
  h->points_per_octave = s->meta[3];
  h->base = exp(log((double)fmin) / (double)fmint);
  if (fabs(h->base - 2.0) < 0.00001) h->base = 2.0;
  h->Fmint = fmint;
  h->Fstept = (fmaxt - fmint) / (float)(h->N - 1);
  h->n_harmonics = 15; h->compression = (float)0.85; h->n_cand = 6; h->old_peaks = 0;
  h->min_pitch = 52.0; h->max_pitch = 620.0; h->voicing_cutoff = (float)0.7;
}

/* smileMath_quadFrom3pts (smileUtil.c:1009-1033) */
static double quad_vertex(double x1, double y1, double x2, double y2, double x3, double y3, double *y)
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

/* One frame: cPitchBase::processVector (pitchBase.cpp:187-310) around cPitchShs::pitchDetect
 * (pitchShs.cpp:214-347, greedyPeakAlgo=1, octaveCorrection=0). dst: 21 values
 * [nCandidates | F0Cand[6] | candVoicing[6] | candScores[6] | F0raw | voicingClip].
 * ss_out (optional, N floats) receives the summation spectrum. */
void lldo_pitch_shs(const lldo_shs *h, const float *in, float *dst, float *ss_out)
{
  const long N = h->N;
  const int NC = h->n_cand;
  float f0c[LLDO_SHS_MAX_CAND], cv[LLDO_SHS_MAX_CAND], cs[LLDO_SHS_MAX_CAND];
  for (int i = 0; i < NC; i++) f0c[i] = cv[i] = cs[i] = 0.0f;
  float *SS = (float *)malloc(sizeof(float) * (size_t)N);
  for (long j = 0; j < N; j++) SS[j] = in[j];
  float scale = h->compression;
  for (int i = 2; i < h->n_harmonics + 1; i++) {
    const long shift = (long)floor((double)h->points_per_octave * (log((double)i) / log(2.0)));
    for (long j = shift; j < N; j++) SS[j - shift] += in[j] * scale;
    scale *= h->compression;
  }
  for (long j = 0; j < N; j++) {
    SS[j] /= (float)h->n_harmonics;
    if (SS[j] < 0) SS[j] = 0.0f;
  }
  if (ss_out) memcpy(ss_out, SS, sizeof(float) * (size_t)N);
  int n_found = 0;
  double mean = (double)SS[0];
  long i;
  for (i = 1; i < N - 1; i++) {
    if (h->old_peaks) {                                    /* greedyPeakAlgo = 0 (:286-302): only a new maximum enters, at the front */
      if ((SS[i - 1] < SS[i]) && (SS[i] > SS[i + 1]) && ((SS[i] > cs[0]) || (cs[0] == 0.0))) {
        for (int j = NC - 1; j > 0; j--) { cs[j] = cs[j - 1]; f0c[j] = f0c[j - 1]; }
        f0c[0] = (float)i;
        cs[0] = SS[i];
        if (n_found < NC) n_found++;
      }
    } else if (SS[i - 1] < SS[i] && SS[i] > SS[i + 1]) {
      for (int j = 0; j < NC; j++) {
        if (cs[j] == 0.0 || cs[j] < SS[i]) {
          for (int jj = NC - 1; jj > j; jj--) { cs[jj] = cs[jj - 1]; f0c[jj] = f0c[jj - 1]; }
          f0c[j] = (float)i;
          cs[j] = SS[i];
          if (n_found < NC) n_found++;
          break;
        }
      }
    }
    mean += (double)SS[i];
  }
  mean = (mean + (double)SS[i]) / (double)N;
  for (int c = 0; c < n_found; c++) {
    const long j = (long)f0c[c];
    const float f1 = f0c[c] * h->Fstept + h->Fmint;
    const float f2 = (f0c[c] + (float)1.0) * h->Fstept + h->Fmint;
    const float f0 = (f0c[c] - (float)1.0) * h->Fstept + h->Fmint;
    double sc = 0;
    const double fx = quad_vertex((double)f0, (double)SS[j - 1], (double)f1, (double)SS[j], (double)f2, (double)SS[j + 1], &sc);
    f0c[c] = (float)exp(fx * log(h->base));
    cs[c] = (float)sc;
    cv[c] = (sc > 0.0 && sc > mean) ? (float)(1.0 - mean / sc) : 0.0f;
  }
  free(SS);
  /* candidates outside [minPitch, maxPitch] are removed, the rest moves up (pitchBase.cpp:212-229) */
  int n = n_found;
  if (n > 0) {
    for (int c = 0; c < NC && n > 0; c++) {
      if ((double)f0c[c] > h->max_pitch || (double)f0c[c] < h->min_pitch) {
        const float orig = f0c[c];
        int j;
        for (j = c + 1; j < NC; j++) { f0c[j - 1] = f0c[j]; cv[j - 1] = cv[j]; cs[j - 1] = cs[j]; }
        f0c[j - 1] = 0; cv[j - 1] = 0; cs[j - 1] = 0;
        if (orig > 0.0) { n--; c--; }
      }
    }
  }
  /* best-scored candidate first (:238-258) */
  int best = 0;
  float mx = cs[0];
  for (int c = 1; c < NC; c++) if (cs[c] > mx) { mx = cs[c]; best = c; }
  if (best > 0) {
    float t;
    t = f0c[0]; f0c[0] = f0c[best]; f0c[best] = t;
    t = cv[0]; cv[0] = cv[best]; cv[best] = t;
    t = cs[0]; cs[0] = cs[best]; cs[best] = t;
  }
  float *o = dst;
  *o++ = (float)n;
  for (int c = 0; c < NC; c++) *o++ = f0c[c];
  for (int c = 0; c < NC; c++) *o++ = cv[c];
  for (int c = 0; c < NC; c++) *o++ = cs[c];
  *o++ = (cv[0] <= h->voicing_cutoff) ? 0.0f : f0c[0];           /* F0raw */
  *o++ = (cv[0] <= h->voicing_cutoff) ? 0.0f : cv[0];            /* voicingClip */
}

/* ------------------------------------------------------------------ cPitchSmootherViterbi */
/* cSmileViterbiPitchSmooth (pitchSmootherViterbi.hpp:158-300) driven by cSmileViterbi::addFrame / flushTrellis /
 * getNextOutputFrame (pitchSmootherViterbi.cpp:80-216, .hpp:105-125) exactly as cPitchSmootherViterbi::myTick
 * (:451-570) drives them: one addFrame per input frame, every decided frame is read at once, flush at end of input.
 * Weights as ComParE sets them AFTER setWeights' own assignment wTvvd = tvv (.hpp:291-299). Frames of 2*NC+4 floats:
 * [f0, voicing] x NC, then data[F0rawI=0] (= nCandidates), 0, 0, vIdx (:471-486 with the unset field indices). */
typedef struct {
  int nS, fsz, L;          /* L = bufferLength */
  float thresh;
  double wLocal, wTvv, wTvvd, wTvuv, wThr, wRange;
  double lastChange;
  long wr, rd, pathIdx, convIdx;
  int pathBuf;
  float *buf, *prev;
  int *paths[2], *best;
  double *cost, *costNew;
} viterbi_t;

static double f_weight(float f)
{
  if (f > 0.0 && f < 100.0) return -(1.0 / 100.0) * f + 1.0;
  else if (f >= 100.0 && f < 350.0) return 0.0;
  else if (f >= 350.0 && f < 600.0) return ((f - 350.0) / 250.0);
  else if (f >= 600.0) return 1.2;
  else if (f <= 0) return 2.0;
  return 0.0;
}

static double local_cost(const viterbi_t *v, int i, const float *fr)
{
  double pv = (double)fr[i * 2 + 1], thr = 0.0;
  if (pv < 0.01) pv = 0.01;
  if (pv > 1.00) pv = 1.00;
  if (pv < v->thresh) thr = v->wThr;
  if (i < v->nS - 1) return (-log(pv) + thr) * v->wLocal + f_weight(fr[i * 2]) * v->wRange;
  double flag = 0.0;
  for (int j = 0; j < v->nS; j++) if (fr[j * 2 + 1] >= v->thresh) { flag = v->wThr; break; }
  return v->wLocal * flag;
}

/* i: state in the current frame, j: state in the previous frame. `i == j == nStates-1` of the reference compares
 * (i == j) with 6 and never holds, so u->u falls through to the final "return 1.0". */
static double trans_cost(viterbi_t *v, int i, int j, const float *prev, const float *cur)
{
  const int U = v->nS - 1;
  if (i < U && j < U) {
    const float f0 = prev[j * 2], f1 = cur[i * 2];
    if (f0 == 0 || f1 == 0) return 999.0;
    const double r = log((double)(f1 / f0));
    const double x = v->wTvv * fabs(r) + v->wTvvd * fabs(r - v->lastChange);
    v->lastChange = r;
    return x;
  }
  if ((i == U && j < U) || (i < U && j == U)) { v->lastChange = 0.0; return v->wTvuv; }
  return 1.0;
}

static void vit_add(viterbi_t *v, const float *frame)
{
  const int nS = v->nS;
  float *b = v->buf + (v->wr % v->L) * v->fsz;
  memcpy(b, frame, sizeof(float) * (size_t)v->fsz);
  v->wr++;
  const float *a = v->prev;
  v->prev = b;
  if (v->pathIdx == 0 || a == NULL) {
    v->pathIdx = 0; v->convIdx = -1;
    for (int i = 0; i < nS; i++) { v->cost[i] = local_cost(v, i, b); v->paths[v->pathBuf][i * v->L] = i; }
  } else {
    const int nb = (v->pathBuf + 1) % 2;
    for (int i = 0; i < nS; i++) {
      int ms = 0;
      double mc = trans_cost(v, i, 0, a, b) + v->cost[0];
      for (int j = 1; j < nS; j++) {
        const double c = trans_cost(v, i, j, a, b) + v->cost[j];
        if (c < mc) { ms = j; mc = c; }
      }
      v->costNew[i] = mc + local_cost(v, i, b);
      memcpy(v->paths[nb] + i * v->L, v->paths[v->pathBuf] + ms * v->L, v->L * sizeof(int));
      v->paths[nb][i * v->L + v->pathIdx % v->L] = i;
    }
    double *t = v->cost; v->cost = v->costNew; v->costNew = t;
    v->pathBuf = nb;
  }
  v->pathIdx++;
  const int *P = v->paths[v->pathBuf];
  if (v->pathIdx - v->convIdx > v->L) {                        /* forced decision for the oldest open frame */
    int ms = 0;
    for (int i = 1; i < nS; i++) if (v->cost[i] < v->cost[ms]) ms = i;
    v->convIdx++;
    v->best[v->convIdx % v->L] = P[ms * v->L + v->convIdx % v->L];
  } else {                                                        /* decide up to where all paths agree */
    for (long n = v->convIdx + 1; n < v->pathIdx; n++) {
      const int x = P[n % v->L];
      int match = 1;
      for (int i = 1; i < nS; i++) if (x != P[i * v->L + n % v->L]) { match = 0; break; }
      if (!match) break;
      v->convIdx++;
      v->best[v->convIdx % v->L] = x;
    }
  }
}

static void vit_flush(viterbi_t *v)
{
  int ms = 0;
  for (int i = 1; i < v->nS; i++) if (v->cost[i] < v->cost[ms]) ms = i;
  const int *P = v->paths[v->pathBuf];
  for (long i = v->convIdx + 1; i < v->pathIdx; i++) {
    v->convIdx++;
    v->best[v->convIdx % v->L] = P[ms * v->L + v->convIdx % v->L];
  }
}

/* shs: T x 21 rows of lldo_pitch_shs; out: T x 2 [F0final, voicingFinalUnclipped]; states (optional): T ints;
 * pending (optional): number of frames that were still undecided at the end of input (decided by flushTrellis) */
void lldo_pitch_viterbi_ex(const float *shs, long T, float voicing_cutoff, int buflen, float *out, int *states, long *pending)
{
  const int VIT_BUF = buflen;
  enum { NC = 6, FSZ = NC * 2 + 4 };
  viterbi_t v;
  memset(&v, 0, sizeof(v));
  v.nS = NC + 1; v.fsz = FSZ; v.L = buflen; v.thresh = voicing_cutoff;
  v.wLocal = 2.0; v.wTvv = 10.0; v.wTvvd = 10.0; v.wTvuv = 10.0; v.wThr = 4.0; v.wRange = 1.0;
  v.lastChange = 1.0;
  v.convIdx = -1;
  v.buf = (float *)malloc(sizeof(float) * FSZ * VIT_BUF);
  v.paths[0] = (int *)malloc(sizeof(int) * v.nS * VIT_BUF);
  v.paths[1] = (int *)malloc(sizeof(int) * v.nS * VIT_BUF);
  v.best = (int *)malloc(sizeof(int) * v.nS * VIT_BUF);
  v.cost = (double *)calloc((size_t)v.nS, sizeof(double));
  v.costNew = (double *)calloc((size_t)v.nS, sizeof(double));
  long n_out = 0;
  for (long t = 0; t <= T; t++) {
    if (t < T) {
      const float *r = shs + t * 21;
      float fr[FSZ];
      for (int i = 0; i < NC; i++) { fr[2 * i] = r[1 + i]; fr[2 * i + 1] = r[1 + NC + i]; }
      fr[2 * NC] = r[0]; fr[2 * NC + 1] = 0.0f; fr[2 * NC + 2] = 0.0f; fr[2 * NC + 3] = (float)t;
      vit_add(&v, fr);
    } else {
      if (pending) *pending = (T > 0) ? v.pathIdx - (v.convIdx + 1) : 0;
      if (T > 0) vit_flush(&v);
    }
    while (v.convIdx + 1 - v.rd > 0) {
      const int s = v.best[v.rd % VIT_BUF];
      const float *b = v.buf + (v.rd % VIT_BUF) * FSZ;
      out[2 * n_out] = (s < NC) ? b[2 * s] : 0.0f;
      out[2 * n_out + 1] = (s < NC) ? b[2 * s + 1] : b[1];
      if (states) states[n_out] = s;
      n_out++;
      v.rd++;
    }
  }
  free(v.buf); free(v.paths[0]); free(v.paths[1]); free(v.best); free(v.cost); free(v.costNew);
}

void lldo_pitch_viterbi(const float *shs, long T, float voicing_cutoff, float *out, int *states, long *pending)
{
  lldo_pitch_viterbi_ex(shs, T, voicing_cutoff, 30, out, states, pending);      /* [is13_pitchSmooth] bufferLength = 30 */
}

/* ------------------------------------------------------------------ the group for one utterance */
/* out: T60 x 2 = level is13_pitchG60 [F0final, voicingFinalUnclipped] after cValbasedSelector
 * (valbasedSelector.cpp:139-237: idx 0 = RMS energy of the windowed 60 ms frame, threshold 0.001, zeroVec).
 * Optional taps: hps T60 x 513 (is13_hpsG60), shs T60 x 21 (is13_pitchShsG60), vit T60 x 2 (is13_pitchG60_viterbi),
 * e60 T60 (is13_e60). Returns T60 (out == NULL: query). */
long lldo_compare_f0_chain(const int16_t *pcm, long n_samples, float *out, float *tap_hps, float *tap_shs,
                           float *tap_vit, float *tap_e60)
{
  lldo_mfcc_cfg c;
  lldo_default_mfcc12_cfg(&c);
  c.frame_size_sec = 0.060; c.preemph_enable = 0; c.zero_pad_symmetric = g_lldo_is13 ? 0 : 1;
  lldo_geom g;
  lldo_geometry(&c, &g);
  const long T = lldo_num_frames(n_samples, g.N, g.H);
  if (!out) return T;
  if (T <= 0) return 0;
  float *x = (float *)malloc(sizeof(float) * (size_t)n_samples);
  lldo_pcm16_to_float(pcm, n_samples, x);
  double *w = (double *)malloc(sizeof(double) * (size_t)g.N);
  lldo_window_table(LLDO_WIN_GAUSS, g.N, 0.4, 1.0, w);
  lldo_specscale ss;
  lldo_shs sh;
  if (!lldo_specscale_init(&ss, g.K, g.frame_size_sec_fft)) { free(x); free(w); return -1; }
  lldo_shs_init(&sh, &ss);
  float *fr = (float *)malloc(sizeof(float) * (size_t)g.N);
  float *sp = (float *)malloc(sizeof(float) * (size_t)g.Nfft);
  float *mg = (float *)malloc(sizeof(float) * (size_t)g.K);
  float *hp = (float *)malloc(sizeof(float) * (size_t)g.K);
  float *shs = (float *)malloc(sizeof(float) * 21 * (size_t)T);
  float *e60 = (float *)malloc(sizeof(float) * (size_t)T);
  for (long t = 0; t < T; t++) {
    lldo_window_apply(x + t * g.H, fr, g.N, w, 0.0);
    e60[t] = lldo_energy_rms(fr, g.N);                            /* [is13_energy60] on is13_winG60 */
    lldo_rfft_frame(fr, g.N, sp, g.Nfft, c.zero_pad_symmetric);
    lldo_fftmag(sp, g.Nfft, mg);
    lldo_specscale_frame(&ss, mg, hp);
    if (tap_hps) memcpy(tap_hps + t * g.K, hp, sizeof(float) * (size_t)g.K);
    lldo_pitch_shs(&sh, hp, shs + t * 21, NULL);
  }
  lldo_pitch_viterbi(shs, T, sh.voicing_cutoff, out, NULL, NULL);
  if (tap_shs) memcpy(tap_shs, shs, sizeof(float) * 21 * (size_t)T);
  if (tap_vit) memcpy(tap_vit, out, sizeof(float) * 2 * (size_t)T);
  if (tap_e60) memcpy(tap_e60, e60, sizeof(float) * (size_t)T);
  for (long t = 0; t < T; t++)
    if (!(e60[t] > (float)0.001)) { out[2 * t] = 0.0f; out[2 * t + 1] = 0.0f; }
  free(x); free(w); free(fr); free(sp); free(mg); free(hp); free(shs); free(e60);
  lldo_specscale_free(&ss);
  return T;
}

/* ------------------------------------------------------------------ cPitchJitter */
/* cPitchJitter::myTick (src/lld/pitchJitter.cpp:591-1064) for [is13_pitchJitter]: searchRangeRel 0.25, minNumPeriods 2,
 * minCC 0.5, useBrokenJitterThresh 0, usePeakToPeakPeriodLength 0, shimmerUseRmsAmplitude 0, lgHNRfloor -100;
 * outputs jitterLocal, jitterDDP, shimmerLocal, logHNR. One call per F0 frame in order; the state (read position in the
 * wave, left-over samples, last period / difference, last jitter / shimmer values) carries over between frames.
 * Time meta of frame t as the framer produces it from a wave level without stored time stamps
 * (dataMemoryLevel.cpp:617-626,1226-1245; waveSource.cpp:196): time = (tH)*Tw, lengthSec = ((tH+N-1)*Tw - (tH)*Tw) + Tw,
 * framePeriod = Tw, period = frameStep -- lenF = ceil(lengthSec/framePeriod) is N or N+1 depending on rounding. */
static double cross_corr(const float *x, const float *y, long N)          /* crossCorr, :331-418 */
{
  double cc = 0.0, mx = 0.0, my = 0.0, nx = 0, ny = 0;
  for (long i = 0; i < N; i++) { mx += x[i]; my += y[i]; }
  mx /= (double)N;
  my /= (double)N;
  for (long i = 0; i < N; i++) {
    cc += (x[i] - mx) * (y[i] - my);
    nx += (x[i] - mx) * (x[i] - mx);
    ny += (y[i] - my) * (y[i] - my);
  }
  cc /= sqrt(nx) * sqrt(ny);
  return cc;
}

/* amplitudeDiff (:422-459); the parabolic peak positions are only used with usePeakToPeakPeriodLength */
static float amplitude_diff(const float *x, long Nx, const float *y, long Ny, float *a0, float *a1)
{
  float max0 = x[1], min0 = x[1];
  for (long i = 1; i < Nx - 1; i++) { if (x[i] > max0) max0 = x[i]; if (x[i] < min0) min0 = x[i]; }
  float max1 = y[1], min1 = y[1];
  for (long i = 1; i < Ny - 1; i++) { if (y[i] > max1) max1 = y[i]; if (y[i] < min1) min1 = y[i]; }
  *a0 = max0 - min0;
  *a1 = max1 - min1;
  return fabsf((max0 - min0) - (max1 - min1));
}

/* smileMath_quadFrom3pts with the curvature output unused */
static double quad_vertex_y(double x1, double y1, double x2, double y2, double x3, double y3, double *y)
{
  return quad_vertex(x1, y1, x2, y2, x3, y3, y);
}

/* wave: the utterance as floats (n samples); f0: T values (level is13_pitchG60, column F0final);
 * out: T x 4 [jitterLocal, jitterDDP, shimmerLocal, logHNR]. N, H: frame size / step in samples. */
/* frames: the F0 value of frame t carries the time stamp of frame t + shift (1 behind cPitchSmoother with simple post smoothing, which
 * delays its values by one frame and hands on the time meta data of the frame it was called with; 0 behind the Viterbi smoother) */
static long g_jit_t_shift = 0;
void lldo_set_jitter_time_shift(long frames) { g_jit_t_shift = frames; }

static void pitch_jitter_impl(const float *wave, long n, const float *f0, long T, long N, long H, double sample_rate,
                              double frame_step_sec, double searchRangeRel, float *out, float *shimmer_db)
{
  const double Tw = 1.0 / sample_rate;                              /* waveSource.cpp:190 */
  const int minNumPeriods = 2;
  float threshCC = (float)0.5;
  const float lgHNRfloor = (float)-100.0;
  long lastIdx = 0, lastMis = 0;
  float lastT0 = 0.0f, lastDiff = 0.0f, lastJitterLocal = 0.0f, lastJitterDDP = 0.0f, lastShimmerLocal = 0.0f;
  for (long t = 0; t < T; t++) {
    const float F0 = f0[t];
    const long tt = t + g_jit_t_shift;
    const double time = (double)(tt * H) * Tw;
    const double lengthSec = ((double)(tt * H + N - 1) * Tw - (double)(tt * H) * Tw) + Tw;
    const long lenF = (long)ceil(lengthSec / Tw);
    const long startVidx = (long)round(time / Tw);
    const long ppLen = (long)ceil(frame_step_sec / Tw);
    long toRead0 = ppLen + lastMis, toRead = toRead0;
    double Tf = 0.0;
    long T0f = 0, T0minF = 0, T0maxF = 0, two_pp = 0;
    if (F0 > 0.0) {
      const double T0 = 1.0 / F0;
      Tf = T0 / Tw;
      T0f = (long)round(Tf);
      T0minF = (long)floor((1.0 - searchRangeRel) * Tf);
      T0maxF = (long)ceil((1.0 + searchRangeRel) * Tf);
      two_pp = minNumPeriods * T0maxF + minNumPeriods;
      if (toRead < two_pp) toRead = two_pp;
    }
    long maxRead = lastMis + lenF;
    if (toRead > maxRead) toRead = maxRead;
    if (startVidx - lastMis != lastIdx) {
      lastIdx = startVidx;
      if (toRead > lenF) toRead = lenF;
      if (maxRead > lenF) maxRead = lenF;
    }
    float *o = out + 4 * t;
    if (lastIdx + toRead > n) {                                     /* no pcm data: the frame yields no output row */
      lastIdx += toRead0;                                           /* (cannot happen for complete frames) */
      o[0] = o[1] = o[2] = o[3] = 0.0f;
      continue;
    }
    const float *d = wave + lastIdx;
    const long nT = toRead;
    float nPeriodsLocal = 0, nPeriodsDDP = 0, nPeriods = 0, avgPeriod = 0.0f, JitterDDP = 0.0f, JitterLocal = 0.0f;
    float avgAmp = 0.0f, avgAmpDiff = 0.0f, lgHNR = 0.0f;
    long start = 0, lastPeriod = 0;
    if (F0 > 0.0) {
      int numPeriods = 0;
      float minCC = (float)-2.0;                                    /* :696, per frame */
      long *periodBuffer = (long *)calloc(1, sizeof(long) * (size_t)((T0f > 0 ? maxRead / T0minF + 3 : maxRead + 2) + 2));
      float *avgWf = (float *)calloc(1, sizeof(float) * (size_t)(T0f + 1));
      double *cc = (double *)calloc(1, sizeof(double) * (size_t)((int)(T0maxF - T0minF) + 1));
      long pp = 0;
      while (start < nT - 2 * T0maxF - 1) {
        for (long tf = T0minF; tf <= T0maxF; tf++) cc[tf - T0minF] = cross_corr(d + start, d + start + tf, tf);
        long maxI = -1;
        double mx = cc[T0f - T0minF];
        for (long i = 1; i < T0maxF - T0minF - 1; i++) {
          if (cc[i - 1] < cc[i] && cc[i] > cc[i + 1]) {
            if (maxI == -1) { maxI = i; mx = cc[i]; }
            else if (cc[i] > mx) { maxI = i; mx = cc[i]; }
          }
        }
        pp = (maxI == -1) ? T0f : T0minF + maxI;
        const long os = start;
        if (maxI >= 0) {
          start += pp;
          float a0 = 0.0f, a1 = 0.0f;
          const float ad = amplitude_diff(d + os, pp, d + start, pp, &a0, &a1);
          periodBuffer[numPeriods++] = os;
          for (long i = 0; i < T0f; i++) avgWf[i] += d[os + i];
          double ccI = 0.0;
          const double maxId = fabs(((double)T0minF + quad_vertex_y((double)(maxI - 1), cc[maxI - 1], (double)maxI, cc[maxI],
                                                                    (double)(maxI + 1), cc[maxI + 1], &ccI))) * Tw;
          if (minCC == (float)-2.0 || minCC > (float)ccI) minCC = (float)ccI;          /* :793-796 */
          if (g_lldo_is13) threshCC = minCC;                            /* useBrokenJitterThresh, :801-807 */
          if (ccI > threshCC) {
            const float period = (float)maxId;
            avgPeriod += period;
            nPeriods += 1.0;
            if (lastT0 > 0.0) {
              const float diff = fabsf(lastT0 - period);
              JitterLocal += diff;
              nPeriodsLocal += 1.0;
              if (lastDiff > 0.0) { JitterDDP += fabsf(lastDiff - diff); nPeriodsDDP += 1.0; }
              lastDiff = diff;
            }
            lastT0 = period;
            avgAmp += (a0 + a1) / (float)2.0;
            avgAmpDiff += ad;
          }
        } else {
          start += T0f;
        }
        if (start < toRead0 - 1) lastPeriod = start;
      }
      periodBuffer[numPeriods++] = start;
      float Eh = 0.0f;
      for (long i = 0; i < T0f && start + i < nT; i++) {
        avgWf[i] += d[start + i];
        avgWf[i] /= (float)numPeriods;
        if (i > 2 && i < T0f - 2) Eh += avgWf[i] * avgWf[i];
      }
      if (T0f - 4 > 0) Eh /= (float)(T0f - 4);
      Eh = sqrtf(Eh);
      float En = 0.0f;
      long nEn = 0;
      if (pp > 0) periodBuffer[numPeriods] = start + pp;
      for (int i = 0; i < numPeriods; i++) {
        long k = 2;
        const long lim = (periodBuffer[i + 1] < periodBuffer[i] + T0f ? periodBuffer[i + 1] : periodBuffer[i] + T0f) - 2;
        for (long j = periodBuffer[i] + 2; j < lim; j++) {
          const float delta = d[j] - avgWf[k++];
          En += delta * delta;
          nEn++;
        }
      }
      if (nEn > 0) En /= (float)nEn;
      En = sqrtf(En);
      if (En > 0.0) {
        const float HNR = Eh / En;
        if (HNR > 0.0) lgHNR = (float)(20.0 * log((double)HNR) / log(10.0));
        else lgHNR = lgHNRfloor;
      }
      lastMis = toRead0 - lastPeriod;
      free(cc); free(periodBuffer); free(avgWf);
    } else {
      lastPeriod = toRead0;
      lastMis = 0;
      lastT0 = 0.0f; lastDiff = 0.0f;
      lastJitterDDP = 0.0f; lastJitterLocal = 0.0f; lastShimmerLocal = 0.0f;
      lgHNR = lgHNRfloor;
    }
    lastIdx += lastPeriod;
    /* output vector (:906-1040) */
    if (nPeriods > 0.0 && nPeriodsLocal > 0.0 && F0 > 0.0) {
      JitterLocal /= nPeriodsLocal;
      lastJitterLocal = JitterLocal / (avgPeriod / nPeriods);
    }
    if (nPeriods > 0.0 && nPeriodsLocal > 0.0 && F0 > 0.0) {
      if (lastJitterLocal > 1.0) lastJitterLocal = 1.0;
      o[0] = lastJitterLocal;
    } else if (nPeriods == 0.0 && F0 > 0.0) {
      if (lastJitterLocal > 1.0) lastJitterLocal = 1.0;
      o[0] = lastJitterLocal;
    } else o[0] = 0.0f;
    if (nPeriods > 0.0 && nPeriodsDDP > 0.0 && F0 > 0.0) {
      JitterDDP /= nPeriodsDDP;
      lastJitterDDP = JitterDDP / (avgPeriod / nPeriods);
    }
    if (nPeriods > 0.0 && nPeriodsDDP > 0.0 && F0 > 0.0) {
      if (lastJitterDDP > 1.0) lastJitterDDP = 1.0;
      o[1] = lastJitterDDP;
    } else if (nPeriods == 0.0 && F0 > 0.0) {
      if (lastJitterDDP > 1.0) lastJitterDDP = 1.0;
      o[1] = lastJitterDDP;
    } else o[1] = 0.0f;
    if (nPeriods > 0.0 && F0 > 0.0) {
      if (avgAmp > 0.0) lastShimmerLocal = avgAmpDiff / avgAmp;
      else lastShimmerLocal = 0.0f;
    }
    if (nPeriods > 0.0 && F0 > 0.0) {
      if (lastShimmerLocal > 1.0) lastShimmerLocal = 1.0;
      o[2] = lastShimmerLocal;
    } else if (nPeriods == 0.0 && F0 > 0.0) {
      if (lastShimmerLocal > 1.0) lastShimmerLocal = 1.0;
      o[2] = lastShimmerLocal;
    } else o[2] = 0.0f;
    if (shimmer_db) {                                                /* shimmerLocalDB (:1000-1030): smileDsp_amplitudeRatioToDB */
      const int voiced_out = (nPeriods > 0.0 && F0 > 0.0) || (nPeriods == 0.0 && F0 > 0.0);
      if (voiced_out) {
        const double a = lastShimmerLocal + 1.0;
        shimmer_db[t] = (float)((a > 10e-50) ? 20.0 * log(a) / log(10.0) : -1000.0);
      } else shimmer_db[t] = 0.0f;
    }
    if (lgHNR < lgHNRfloor) lgHNR = lgHNRfloor;
    o[3] = lgHNR;
  }
}

void lldo_pitch_jitter(const float *wave, long n, const float *f0, long T, long N, long H, double sample_rate,
                       double frame_step_sec, float *out)
{
  pitch_jitter_impl(wave, n, f0, T, N, H, sample_rate, frame_step_sec, 0.25, out, NULL);       /* [is13_pitchJitter] */
}

/* GeMAPS' option set ([gemapsv01b_pitchJitter]: searchRangeRel 0.1, jitterLocal + shimmerLocalDB):
 * out4 as above, shimmer_db[T] = 20 log10(shimmerLocal + 1) */
void lldo_pitch_jitter_ex(const float *wave, long n, const float *f0, long T, long N, long H, double sample_rate,
                          double frame_step_sec, double search_range_rel, float *out4, float *shimmer_db)
{
  pitch_jitter_impl(wave, n, f0, T, N, H, sample_rate, frame_step_sec, search_range_rel, out4, shimmer_db);
}

/* ------------------------------------------------------------------ [is13_smoNz] + [is13_deNz] */
/* The F0 group's LLD columns as the LLD sinks see them: T60+1 rows x 12 =
 *   [F0final_sma, voicingFinalUnclipped_sma, jitterLocal_sma, jitterDDP_sma, shimmerLocal_sma, logHNR_sma | their deltas]
 * cContourSmoother with noZeroSma (contourSmoother.cpp:85-100) over the two levels is13_pitchG60 / is13_jitterShimmer,
 * then cDeltaRegression with onlyInSegments = zeroSegBound = 1 (deltaRegression.cpp:113-135), whose `norm` member
 * keeps growing by i^2 for every valid pair: rows in order, within a row columns in order (blocksize 1,
 * windowProcessor.cpp:164-229).
 * End of input [measured against the binary, T60 = 4..995, P = 1..7]: cPitchJitter does not run during end-of-input
 * ticks (pitchJitter.cpp:593), and the Viterbi smoother releases its last P frames only then. The tick loop alternates
 * end-of-input and normal phases (componentManager.cpp:1416-1560), each window processor padding with the last frame
 * it can see while at least one real frame is in its window:
 *   - smoNz: rows n <= T-P see the jitter level clipped at frame T-P-1 (the pitch level is complete);
 *   - deNz: rows n <= T-P+2 see the smoothed level clipped at row T-P, row T-P+3 at row T-1, later rows all of it;
 *   - P == T (nothing decided before the end): no clipping.
 * Requires T60 >= 4 like the A+B groups (returns 0 rows otherwise). */
long lldo_compare_f0_lld(const int16_t *pcm, long n_samples, float *out12)
{
  const double rate = lldo_get_sample_rate();
  const long N = lround(0.060 * rate), H = lround(0.010 * rate);
  const long T = lldo_num_frames(n_samples, N, H);
  if (T < 4) return 0;
  const long rows = T + 1;
  if (!out12) return rows;
  float *x6 = (float *)malloc(sizeof(float) * 6 * (size_t)T);
  float *p2 = (float *)malloc(sizeof(float) * 2 * (size_t)T);
  float *shs = (float *)malloc(sizeof(float) * 21 * (size_t)T);
  float *e60 = (float *)malloc(sizeof(float) * (size_t)T);
  float *wave = (float *)malloc(sizeof(float) * (size_t)n_samples);
  float *j4 = (float *)malloc(sizeof(float) * 4 * (size_t)T);
  float *f0 = (float *)malloc(sizeof(float) * (size_t)T);
  lldo_compare_f0_chain(pcm, n_samples, p2, NULL, shs, NULL, e60);
  long P = 0;
  { float *tmp = (float *)malloc(sizeof(float) * 2 * (size_t)T);
    lldo_pitch_viterbi(shs, T, (float)0.7, tmp, NULL, &P);
    free(tmp); }
  lldo_pcm16_to_float(pcm, n_samples, wave);
  for (long t = 0; t < T; t++) f0[t] = p2[2 * t];
  lldo_pitch_jitter(wave, n_samples, f0, T, N, H, rate, 0.010, j4);
  for (long t = 0; t < T; t++) {
    x6[6 * t] = p2[2 * t]; x6[6 * t + 1] = p2[2 * t + 1];
    for (int d = 0; d < 4; d++) x6[6 * t + 2 + d] = j4[4 * t + d];
  }
  float *z = out12;                                    /* smoothed level, row stride 12 */
  for (long n = 0; n < rows; n++)
    for (int d = 0; d < 6; d++) {
      long clip = T - 1;
      if (d >= 2 && n <= T - P && P < T) clip = T - P - 1;
      if (clip < 0) clip = 0;
#define XC(i) x6[6 * ((i) < 0 ? 0 : ((i) > clip ? clip : (i))) + d]
      const float c = XC(n);
      float y = 0.0f;
      if (c != 0.0) {
        long cnt = 1;
        y = c;
        if (XC(n - 1) != 0.0) { y += XC(n - 1); cnt++; }
        if (XC(n + 1) != 0.0) { y += XC(n + 1); cnt++; }
        y /= (float)cnt;
      }
#undef XC
      z[12 * n + d] = y;
    }
  float norm = 0.0f;
  for (int i = 1; i <= 2; i++) norm += (float)i * (float)i;
  norm *= 2.0;
  for (long n = 0; n < rows; n++) {
    long clip = T;
    if (P < T) clip = (n <= T - P + 2) ? T - P : ((n == T - P + 3) ? T - 1 : T);
    if (clip < 0) clip = 0;
    if (clip > rows - 1) clip = rows - 1;
    for (int d = 0; d < 6; d++) {
      float num = 0.0f;
      for (int i = 1; i <= 2; i++) {
        long ia = n - i, ib = n + i;
        if (ia < 0) ia = 0;
        if (ia > clip) ia = clip;
        if (ib > clip) ib = clip;
        const float a = z[12 * ia + d], b = z[12 * ib + d];
        if (!(a == 0.0 || b == 0.0 || isnan(a) || isnan(b))) {
          const float delta = b - a;
          num += (float)i * delta;
          norm += (float)i * (float)i;
        }
      }
      z[12 * n + 6 + d] = (norm != 0.0) ? num / norm : 0.0f;
    }
  }
  free(x6); free(p2); free(shs); free(e60); free(wave); free(j4); free(f0);
  return rows;
}

/* The whole LLD level of ComParE_2016 (lld;lld_de of config/compare16/ComParE_2016.conf, 130 columns):
 * [F0 group (6) | group A (4) | group B (55) | the same 65 columns' deltas]; rows = T60 + 1 */
long lldo_compare_lld_chain(const int16_t *pcm, long n_samples, float *out130)
{
  const long rows = lldo_compare_ab_chain(pcm, n_samples, NULL, NULL);
  if (rows <= 0) return 0;
  if (!out130) return rows;
  float *ab = (float *)malloc(sizeof(float) * 118 * (size_t)rows);
  float *f = (float *)malloc(sizeof(float) * 12 * (size_t)rows);
  lldo_compare_ab_chain(pcm, n_samples, ab, NULL);
  const long r2 = lldo_compare_f0_lld(pcm, n_samples, f);
  if (r2 != rows) { free(ab); free(f); return -1; }
  for (long r = 0; r < rows; r++) {
    float *o = out130 + 130 * r;
    memcpy(o, f + 12 * r, sizeof(float) * 6);
    memcpy(o + 6, ab + 118 * r, sizeof(float) * 59);
    memcpy(o + 65, f + 12 * r + 6, sizeof(float) * 6);
    memcpy(o + 71, ab + 118 * r + 59, sizeof(float) * 59);
  }
  free(ab); free(f);
  return rows;
}
