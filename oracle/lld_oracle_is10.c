/* TEST INFRASTRUCTURE ONLY -- never linked into the product (see oracle/Makefile).
 *
 * Plain-C restatement of the components the INTERSPEECH 2010-2012 sets (config/is09-13/IS10_paraling.conf,
 * IS11_speaker_state.conf, IS12_speaker_trait.conf) add to the ones the other files of this directory restate:
 *   cIntensity         src/lldcore/intensity.cpp:91-145
 *   cLsp               src/lld/lsp.cpp:112-312 (the Speex-derived lpc_to_lsp)
 *   cPitchSmoother     src/lldcore/pitchSmoother.cpp:236-425 (medianFilter0 = 0; post smoothing none / simple)
 *   cVectorOperation   src/other/vectorOperation.cpp:284-530 (the element-wise operations)
 * Pinned against the real binary's levels by tests/test_oracle_pin_is10.py. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lld_oracle.h"

/* ------------------------------------------------------------------ cIntensity */
/* setupNamesForField (intensity.cpp:91-112): Hamming window as doubles, its sum in index order. */
void lldo_intensity_window(long n, double *win, double *win_sum)
{
  const double NN = (double)n;
  long j = 0;
  for (double i = 0.0; i < NN; i += 1.0) win[j++] = 0.54 - 0.46 * cos((2.0 * M_PI * i) / (NN - 1.0));   /* smileUtil.c:1291-1303 */
  double s = 0.0;
  for (j = 0; j < n; j++) s += win[j];
  *win_sum = s <= 0.0 ? 1.0 : s;
}

/* processVector (intensity.cpp:125-145). n_dst = the number of outputs of the field (intensity + loudness): the
 * reference clamps the summation length to MIN(Nsrc, MIN(nWin, Ndst)) -- with one or two outputs it sums one or two samples. */
int lldo_intensity_frame(const float *src, long n_src, const double *win, long n_win, double win_sum, int intensity, int loudness,
                         float *dst)
{
  const long n_dst = (intensity ? 1 : 0) + (loudness ? 1 : 0);
  long safe = n_win < n_dst ? n_win : n_dst;
  if (n_src < safe) safe = n_src;
  double Im = 0.0;
  for (long i = 0; i < safe; i++) Im += win[i] * (double)src[i] * (double)src[i];
  Im /= win_sum;
  int n = 0;
  if (intensity) dst[n++] = (float)Im;
  if (loudness) dst[n++] = (float)pow(Im / 0.000001, 0.3);
  return n;
}

/* ------------------------------------------------------------------ cLsp */
static float cheb_poly_eva(const float *coef, float x, int m)             /* lsp.cpp:112-128 */
{
  float b0 = 0, b1 = 0, tmp;
  x *= 2;
  for (int k = m; k > 0; k--) {
    tmp = b0;
    b0 = x * b0 - b1 + coef[m - k];
    b1 = tmp;
  }
  return (-b1 + (float)0.5 * x * b0 + coef[m]);
}

static int lpc_to_lsp(const float *a, int lpcrdr, float *freq, int nb, float delta)   /* lsp.cpp:144-269 */
{
  float P[40], Q[40];
  float temp_xr, xl, xr, xm = 0, psuml, psumr, psumm, temp_psumr;
  int roots = 0, flag;
  const int m = lpcrdr / 2;
  memset(P, 0, sizeof(P));
  memset(Q, 0, sizeof(Q));
  P[0] = 1.f;
  Q[0] = 1.f;
  for (int i = 0; i < m; i++) {
    P[i + 1] = (a[i] + a[lpcrdr - 1 - i]) - P[i];
    Q[i + 1] = (a[i] - a[lpcrdr - 1 - i]) + Q[i];
  }
  for (int i = 0; i < m; i++) {
    P[i] = 2 * P[i];
    Q[i] = 2 * Q[i];
  }
  xr = 0;
  xl = 1.0;
  for (int j = 0; j < lpcrdr; j++) {
    const float *pt = (j & 1) ? Q : P;
    psuml = cheb_poly_eva(pt, xl, m);
    flag = 1;
    while (flag && (xr >= -1.0)) {
      float dd = delta * ((float)1.0 - (float)0.9 * xl * xl);
      if (fabs(psuml) < .2) dd *= (float)0.5;
      xr = xl - dd;
      psumr = cheb_poly_eva(pt, xr, m);
      temp_psumr = psumr;
      temp_xr = xr;
      if ((psumr * psuml) < 0.0) {
        roots++;
        psumm = psuml;
        for (int k = 0; k <= nb; k++) {
          xm = (float)0.5 * (xl + xr);
          psumm = cheb_poly_eva(pt, xm, m);
          if (!((psumm * psuml) < 0.0)) {
            psuml = psumm;
            xl = xm;
          } else {
            psumr = psumm;
            xr = xm;
          }
        }
        if (xm > 1.0) xm = 1.0;
        else if (xm < -1.0) xm = -1.0;
        freq[j] = acosf(xm);                     /* C++: acos(FLOAT_DMEM) is the float overload */
        xl = xm;
        flag = 0;
      } else {
        psuml = temp_psumr;
        xl = temp_xr;
      }
    }
  }
  (void)psumr;
  return roots;
}

/* processVector (lsp.cpp:289-312); dst is the frame's output row as the reference's writer hands it over (zeroed) */
void lldo_lsp_frame(const float *lpc, int n_lpc, float *dst)
{
  for (int i = 0; i < n_lpc; i++) dst[i] = 0.0f;
  int roots = lpc_to_lsp(lpc, n_lpc, dst, 10, (float).2);
  if (roots != n_lpc) {
    roots = lpc_to_lsp(lpc, n_lpc, dst, 10, (float).05);
    if (roots != n_lpc)
      for (int i = roots; i < n_lpc; i++) dst[i] = 0.0;
  }
}

/* ------------------------------------------------------------------ cPitchSmoother */
void lldo_pitch_smoother_init(lldo_pitch_smoother *s, int n_cand, float voicing_cutoff, int octave_correction, int post_simple,
                              int flags)
{
  memset(s, 0, sizeof(*s));
  s->n_cand = n_cand;
  s->voicing_cutoff = voicing_cutoff;
  s->octave_correction = octave_correction;
  s->post_simple = post_simple;
  s->flags = flags;
  s->first_frame = 1;
}

int lldo_pitch_smoother_width(int flags)
{
  int n = 0;
  for (int b = 0; b < 4; b++) n += (flags >> b) & 1;
  return n;
}

/* processVector (pitchSmoother.cpp:236-425) for one frame: src = [F0Cand (n) | candVoicing (n) | candScore (n)] of cPitchShs.
 * Returns the number of values written (0: no output for this frame -- the first frame with simple post smoothing). */
int lldo_pitch_smoother_frame(lldo_pitch_smoother *s, const float *src, float *dst)
{
  float f0cand[16], candVoice[16], candScore[16];
  const int c = s->n_cand;
  int n = 0;
  for (int j = 0; j < c; j++) {
    f0cand[j] = src[j];
    candVoice[j] = src[c + j];
    candScore[j] = src[2 * c + j];
  }
  if (s->octave_correction) {
    int cand0ismin = 1, minC = -1;
    float vpMin = 0.0;
    for (int i = 1; i < c; i++) {
      if ((f0cand[i] > 0.0) && (f0cand[i] < f0cand[0])) {
        if ((candVoice[i] > 0.9 * candVoice[0]) && (candVoice[i] > vpMin)) {
          vpMin = candVoice[i];
          minC = i;
        }
        cand0ismin = 0;
      }
    }
    if (!cand0ismin) {
      if (minC >= 0) {
        float t = f0cand[0]; f0cand[0] = f0cand[minC]; f0cand[minC] = t;
        t = candVoice[0]; candVoice[0] = candVoice[minC]; candVoice[minC] = t;
        t = candScore[0]; candScore[0] = candScore[minC]; candScore[minC] = t;
      }
    } else {
      int halfed = 0, j = 0;
      while ((!halfed) && j < c - 1) {
        for (int i = j + 1; i < c; i++) {
          if ((f0cand[i] > 0.0) && (f0cand[j] > 0.0)) {
            float k = fabs(f0cand[i] - f0cand[j]) * (float)2.0 / f0cand[0];
            k = (float)fabs(k - 1.0);
            if (k < 0.1) {
              f0cand[0] /= (float)2.0;
              halfed = 1;
              break;
            }
          }
        }
        j++;
      }
    }
  }
  float voiceC1 = candVoice[0];
  if (s->flags & 3) {
    float pitch, pitchOut;
    if (candVoice[0] > s->voicing_cutoff) pitch = f0cand[0];
    else pitch = 0.0;
    if (s->post_simple) {
      if (s->first_frame) { s->first_frame = 0; return 0; }
      voiceC1 = s->last_voice;
      s->last_voice = candVoice[0];
      if ((s->last_final == 0.0) && (pitch > 0.0)) s->ons_flag = 1;
      if ((s->last_final > 0.0) && (pitch == 0.0) && (s->ons_flag == 0)) s->ons_flag = -1;
      if ((s->last_final > 0.0) && (pitch > 0.0)) s->ons_flag = 0;
      if ((s->last_final == 0.0) && (pitch == 0.0)) s->ons_flag = 0;
      if ((pitch == 0.0) && (s->ons_flag == 1)) s->last_final = 0.0;
      else if ((pitch > 0.0) && (s->ons_flag == -1)) s->last_final = pitch;
      int doubling = 0, halfing = 0;
      if ((s->last_final > 0.0) && (pitch > 0.0)) {
        const float factor = s->last_final / pitch;
        if (factor > 1.2) halfing = 1;
        else if (factor < 0.8) doubling = 1;
      }
      if ((doubling) && (s->ons_flag_o == -1)) s->last_final = pitch;
      else if ((halfing) && (s->ons_flag_o == 1)) s->last_final = pitch;
      if (doubling) s->ons_flag_o = 1;
      if (halfing && (s->ons_flag == 0)) s->ons_flag_o = -1;
      if (!(halfing || doubling)) s->ons_flag_o = 0;
      pitchOut = s->last_final;
      s->last_final = pitch;
    } else {
      pitchOut = pitch;
    }
    if (s->flags & 1) dst[n++] = pitchOut;
    if (s->flags & 2) {
      if (pitchOut > 0.0) {
        if (s->pitch_env == 0.0) s->pitch_env = pitchOut;
        else s->pitch_env = (float)0.75 * s->pitch_env + (float)0.25 * pitchOut;
      }
      dst[n++] = s->pitch_env;
    }
  }
  if (s->flags & 4) dst[n++] = (voiceC1 > s->voicing_cutoff) ? voiceC1 : 0.0f;
  if (s->flags & 8) dst[n++] = voiceC1;
  return n;
}

/* ------------------------------------------------------------------ cVectorOperation, element-wise operations */
/* vectorOperation.cpp:360-435, 508-527; std::log / std::exp / std::sqrt on FLOAT_DMEM are the float overloads */
int lldo_vecop(int op, float param1, float logfloor, const float *src, float *dst, long n)
{
  float f;
  switch (op) {
    case LLDO_VOP_ADD: for (long i = 0; i < n; i++) dst[i] = src[i] + param1; return 0;
    case LLDO_VOP_MUL: for (long i = 0; i < n; i++) dst[i] = src[i] * param1; return 0;
    case LLDO_VOP_LOG: for (long i = 0; i < n; i++) dst[i] = src[i] > logfloor ? logf(src[i]) : logf(logfloor); return 0;
    case LLDO_VOP_LOGA:
      f = logf(param1);
      for (long i = 0; i < n; i++) dst[i] = (src[i] > logfloor ? logf(src[i]) : logf(logfloor)) / f;
      return 0;
    case LLDO_VOP_SQRT: for (long i = 0; i < n; i++) dst[i] = src[i] > 0.0 ? sqrtf(src[i]) : 0.0f; return 0;
    case LLDO_VOP_E: for (long i = 0; i < n; i++) dst[i] = expf(src[i]); return 0;
    case LLDO_VOP_ABS: for (long i = 0; i < n; i++) dst[i] = (float)fabs(src[i]); return 0;
    case LLDO_VOP_DB_POW:
      f = (float)(10.0 / log(10.0));
      for (long i = 0; i < n; i++) dst[i] = f * (src[i] > logfloor ? logf(src[i]) : logf(logfloor));
      return 0;
    case LLDO_VOP_DB_MAG:
      f = (float)(20.0 / log(10.0));
      for (long i = 0; i < n; i++) dst[i] = f * (src[i] > logfloor ? logf(src[i]) : logf(logfloor));
      return 0;
  }
  return -1;
}

/* the vector-to-scalar operations (vectorOperation.cpp:461-490): sum, ssm, ll1, ll2 */
float lldo_vecop_reduce(int op, const float *src, long n)
{
  float d = 0.0;
  switch (op) {
    case LLDO_VOP_X_SUM: for (long i = 0; i < n; i++) d += src[i]; break;
    case LLDO_VOP_X_SUMSQ: for (long i = 0; i < n; i++) d += src[i] * src[i]; break;
    case LLDO_VOP_X_L1:
      for (long i = 0; i < n; i++) d += src[i];
      if (n > 0) d /= (float)n;
      break;
    case LLDO_VOP_X_L2:
      for (long i = 0; i < n; i++) d += src[i] * src[i];
      if (d > 0.0) d = sqrtf(d);
      if (n > 0) d /= (float)n;
      break;
  }
  return d;
}
