/*
 * lld_oracle_is09.c -- CPU ORACLE, part 2. TEST INFRASTRUCTURE ONLY (see lld_oracle.h).
 *
 * Restatement of the reference components that IS09_emotion's LLD chain adds to
 * the MFCC chain (SURVEY.md 8a rows R9, R10, R12 and the SMA half of R13):
 * cAcf (ACF + cepstrum), cPitchACF, cEnergy, cMZcr, cContourSmoother, and the
 * tick-accurate window-processor chain [SMA -> delta] feeding the LLD sinks.
 * Citations are relative to the reference root.
 */
#include "lld_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

extern void lldo_irfft_symmetric(float *data, long n);   /* defined below; uses the rdft hook if set */

/* ---------------------------------------------------------------------- R12 */
/* cEnergy::processVector, rms branch, src/lldcore/energy.cpp:152-168
 * (escaleRms = 1, ebiasRms = 0 are the defaults IS09 keeps) */
float lldo_energy_rms(const float *x, long N)
{
  double d = 0.0;
  for (long i = 0; i < N; i++) { float t = x[i]; d += t * t; }
  return (float)sqrt(d / (float)N) * 1.0f + 0.0f;
}

/* cMZcr::processVector, zcr only, src/lldcore/mzcr.cpp:109-126 */
float lldo_zcr(const float *src, long N)
{
  float nzc = 0.0f;
  for (long i = 1; i < N - 1; i++) {
    if (((src[i - 1] * src[i + 1] <= 0.0) && (src[i] == 0.0)) || (src[i - 1] * src[i] < 0.0)) nzc += 1.0;
  }
  nzc /= (float)N;
  return nzc;
}

/* ---------------------------------------------------------------------- R9 */
static lldo_rfft_fn g_hook2 = 0;
void lldo_set_rfft_hook2(lldo_rfft_fn fn) { g_hook2 = fn; }

/* Inverse of the packed real FFT as Ooura's rdft(n,-1,a) defines it
 * (src/dspcore/fftsg.c:103-135):
 *   a[k] = R0/2 + R[n/2] cos(pi k)/2 + sum_{j=1}^{n/2-1} R[j] cos(2 pi jk/n) + I[j] sin(2 pi jk/n)
 * cAcf only feeds it purely real spectra (I = 0). Built-in version: forward
 * FFT of the symmetric extension (own radix-2 code path of lldo_rfft_frame). */
void lldo_irfft_packed_real(float *a, long n)
{
  if (g_hook2) {
    int *ip = (int *)calloc(1, sizeof(int) * (size_t)(n + 2));
    float *w = (float *)calloc(1, sizeof(float) * (size_t)((n * 5) / 4 + 2));
    g_hook2((int)n, -1, a, ip, w);
    free(ip); free(w);
    return;
  }
  if (n >= 64 && lldo_ooura_rdft((int)n, -1, a) == 0) return;   /* the reference's operation order */
  /* s[j] = R_j (j <= n/2), s[n-j] = R_j: DFT(s)[k] = 2 a[k] */
  float *s = (float *)malloc(sizeof(float) * (size_t)n);
  float *f = (float *)malloc(sizeof(float) * (size_t)n);
  s[0] = a[0]; s[n / 2] = a[1];
  for (long j = 1; j < n / 2; j++) { s[j] = a[2 * j]; s[n - j] = a[2 * j]; }
  lldo_rfft_frame(s, n, f, n, 0);            /* packed forward spectrum of s (real parts are what we need) */
  a[0] = 0.5f * f[0];
  for (long k = 1; k < n / 2; k++) a[k] = 0.5f * f[2 * k];
  a[n / 2] = 0.5f * f[1];
  for (long k = n / 2 + 1; k < n; k++) a[k] = a[n - k];
  free(s); free(f);
}

/* cAcf::processVector, forward direction, src/dspcore/acf.cpp:249-349 with the
 * defaults IS09 keeps: usePower=1, symmetricData=1 (Ndst = Nsrc-1),
 * acfCepsNormOutput=1, absCepstrum=0, cosLifterCepstrum=0, oldCompatCepstrum=0. */
void lldo_acf(const float *mag, long K, int cepstrum, float *dst)
{
  long N = (K - 1) * 2, Ndst = K - 1, i;
  float *src = (float *)malloc(sizeof(float) * (size_t)K);
  float *data = (float *)malloc(sizeof(float) * (size_t)N);
  for (i = 0; i < K; i++) src[i] = mag[i] * mag[i];                     /* :252-259 */
  if (cepstrum) {                                                        /* :288-305 */
    data[0] = (src[0] > 0.0) ? (float)(log(src[0] + 1.0)) : 0.0f;
    data[1] = (src[K - 1] > 0.0) ? (float)(log(src[K - 1] + 1.0)) : 0.0f;
    for (i = 2; i < N - 1; i += 2) {
      data[i] = (src[i >> 1] > 0.0) ? (float)log(src[i >> 1] + 1.0) : 0.0f;
      data[i + 1] = 0.0f;
    }
  } else {                                                               /* :308-314 */
    data[0] = src[0];
    data[1] = src[K - 1];
    for (i = 2; i < N - 1; i += 2) { data[i] = src[i >> 1]; data[i + 1] = 0.0f; }
  }
  lldo_irfft_packed_real(data, N);                                       /* :317 */
  for (i = 0; (i < N) && (i < Ndst); i++) data[i] = (float)data[i] / (float)K;   /* :321-325 */
  if (cepstrum) for (i = 0; (i < N) && (i < Ndst); i++) dst[i] = data[i];          /* :337-341 */
  else          for (i = 0; (i < N) && (i < Ndst); i++) dst[i] = (float)fabs(data[i]);  /* :343 */
  free(src); free(data);
}

/* ---------------------------------------------------------------------- R10 */
/* cPitchACF::voicingProb, src/lldcore/pitchACF.cpp:249-284 */
static double voicing_prob(const float *a, int n, int skip, double *Zcr)
{
  int zcr = 0, mcr = 0;
  double mean, max;
  max = a[n - 1];
  mean = a[skip];
  for (int i = 1; i < n; i++) {
    if (a[i - 1] * a[i] < 0) zcr++;
    if (i >= skip) {
      if ((a[i] > max) && (a[i - 1] < a[i])) max = a[i];
      mean += a[i];
    }
  }
  mean /= (double)(n - skip + 1);
  for (int i = 1; i < n; i++) if ((a[i - 1] - mean) * (a[i] - mean) < 0) mcr++;
  if (Zcr) *Zcr = (mcr > zcr) ? (double)mcr / (double)n : (double)zcr / (double)n;
  if (a[0] > 0) return max / a[0];
  return 0.0;
}

/* cPitchACF::pitchPeak, src/lldcore/pitchACF.cpp:286-310 */
static long pitch_peak(const float *a, long n, long skip)
{
  double max, buf, sum = 0.0;
  max = a[n - 1];
  for (int i = (int)n - 1; i >= 0; i--) {
    buf = a[i];
    sum += fabs(buf);
    if (i >= skip) if (buf > max) max = buf;
  }
  sum /= n;
  for (int i = (int)skip + 1; i < n - 1; i++) {
    if (a[i] > (max + sum) * 0.6) {
      if ((a[i - 1] < a[i]) && (a[i] > a[i + 1])) return i;
    }
  }
  return 0;
}

void lldo_pitch_state_init(lldo_pitch_state *s)
{
  s->lastPitch = 0.0f; s->lastlastPitch = 0.0f; s->glMeanPitch = 0.0f; s->onsFlag = 0; s->pitchEnv = 0.0f;
}

/* cPitchACF::processVector (voiceProb + F0 outputs), src/lldcore/pitchACF.cpp:137-247.
 * acf, ceps: the two 256-value inputs (reader order is09_acf;is09_cepstrum).
 * fsSec = (float) frameSizeSec of the first input level (:113-116). */
void lldo_pitch_acf(const float *acf, const float *ceps, long Nhalf, float fsSec, double maxPitch,
                    double voicingCutoff, lldo_pitch_state *st, float *voiceProb, float *F0,
                    float *raw_pitch_after_cutoff)
{
  long Nsrc = 2 * Nhalf;
  long N = (int)floor(Nsrc / 2.0);
  double Nd = (double)(Nsrc);
  double Tsamp = fsSec / Nd;
  int preskip = (maxPitch <= 0.0) ? 0 : (int)(1.0 / (maxPitch * Tsamp));
  double acfZcr = 0.0;
  double voicing = voicing_prob(acf, (int)N, preskip, &acfZcr);
  long maxIdx = pitch_peak(ceps, N, preskip + 1);
  *voiceProb = (float)voicing;
  float pitch = 0.0f;
  if (maxIdx > 0) pitch = (float)1.0 / ((float)(maxIdx) * (float)Tsamp);
  if (voicing < voicingCutoff) { maxIdx = 0; pitch = 0.0f; }
  if (raw_pitch_after_cutoff) *raw_pitch_after_cutoff = pitch;

  /* the causal contour smoother, :199-243 */
  if ((st->lastPitch == 0.0) && (pitch > 0.0)) st->onsFlag = 1;
  if ((st->lastPitch > 0.0) && (pitch == 0.0) && (st->onsFlag == 0)) st->onsFlag = -1;
  if ((st->lastPitch > 0.0) && (pitch > 0.0)) st->onsFlag = 0;
  if ((st->lastPitch == 0.0) && (pitch == 0.0)) st->onsFlag = 0;
  if ((pitch == 0.0) && (st->onsFlag == 1)) { st->lastPitch = 0.0f; }
  float oPitch = pitch;
  float tol = (float)0.4;
  float alpha = (float)0.3;
  if (pitch > 0.0) {
    if (st->glMeanPitch == 0.0) st->glMeanPitch = pitch;
    if (!((pitch < ((float)1.0 + tol) * st->glMeanPitch) && (pitch > ((float)1.0 - tol) * st->glMeanPitch))) {
      pitch = st->glMeanPitch;
      alpha /= (float)3.0;
    }
    if (st->onsFlag && (st->lastPitch > pitch)) st->lastPitch *= (float)0.85;
  }
  if ((pitch > 0.0) && (st->onsFlag == -1)) { st->lastPitch = pitch; }
  if (oPitch > (float)0.0) st->glMeanPitch = ((float)1.0 - alpha) * st->glMeanPitch + alpha * oPitch;
  float out;
  if ((st->lastlastPitch != (float)0.0) && (st->lastPitch != 0.0)) out = (float)0.5 * (st->lastlastPitch + st->lastPitch);
  else out = st->lastPitch;
  *F0 = out;
  st->lastlastPitch = st->lastPitch;
  st->lastPitch = pitch;
}

/* ------------------------------------------------------------- R13 (general) */
/* Tick-accurate chain of window processors (cWindowProcessor, blocksize 1):
 * stage s has kind[s] (0 = cDeltaRegression with deltawin W, 1 =
 * cContourSmoother with smaWin = 2W+1, 2 = the same with noZeroSma, 3 = cDeltaRegression with onlyInSegments; at most 15 stages) and W[s] = pre = post. Same control flow
 * as lldo_delta_chain (see there for the reference lines). levels_out[s]
 * receives level s+1 (must hold T + sum_{i<=s} W[i] frames of D floats);
 * returns nothing -- frame counts are T + cumulative W. */
void lldo_window_chain(const float *x, long T, long D, int n_stages, const int *kind, const int *Wv, float **levels_out)
{
  if (T <= 0 || n_stages <= 0) return;
  int o, i;
  long cap = T + 2;
  for (o = 0; o < n_stages; o++) cap += 3 * Wv[o];
  float **lv = (float **)calloc((size_t)n_stages + 1, sizeof(float *));
  long *curW = (long *)calloc((size_t)n_stages + 1, sizeof(long));
  char *done = (char *)calloc((size_t)n_stages + 1, 1);
  float seg_norm[16] = {0};
  char seg_norm_set[16] = {0};
  for (o = 0; o <= n_stages; o++) lv[o] = (float *)calloc((size_t)cap * (size_t)D, sizeof(float));
  memcpy(lv[0], x, sizeof(float) * (size_t)T * (size_t)D);
  curW[0] = T;
  int Wmax = 0;
  for (o = 0; o < n_stages; o++) if (Wv[o] > Wmax) Wmax = Wv[o];
  float *blk = (float *)malloc(sizeof(float) * (size_t)(2 * Wmax + 1) * (size_t)D);
  for (int eoi = 0; eoi <= 1; eoi++) {
    int progress = 1;
    while (progress) {
      progress = 0;
      for (o = 1; o <= n_stages; o++) {
        if (done[o]) continue;
        const int W = Wv[o - 1];
        const float *in = lv[o - 1];
        long wIn = curW[o - 1];
        long t = curW[o];
        long vOld = t - W, vEnd = t + W + 1;
        long v = vOld < 0 ? 0 : vOld;
        long padEnd = 0;
        if (vEnd > wIn) {
          if (!eoi) continue;
          padEnd = vEnd - wIn;
          if (padEnd >= vEnd - v) { done[o] = 1; continue; }
        }
        if (!(v < wIn)) continue;
        if (curW[o] >= cap) { done[o] = 1; continue; }
        long j;
        if (vOld < 0) {
          long i0 = -vOld;
          for (j = 0; j < i0; j++) memcpy(blk + j * D, in, sizeof(float) * (size_t)D);
          for (j = 0; j < vEnd; j++) memcpy(blk + (j + i0) * D, in + j * D, sizeof(float) * (size_t)D);
        } else if (padEnd > 0) {
          long n = (vEnd - v) - padEnd;
          for (j = 0; j < n; j++) memcpy(blk + j * D, in + (v + j) * D, sizeof(float) * (size_t)D);
          for (; j < vEnd - v; j++) memcpy(blk + j * D, in + (v + n - 1) * D, sizeof(float) * (size_t)D);
        } else {
          for (j = 0; j < vEnd - v; j++) memcpy(blk + j * D, in + (v + j) * D, sizeof(float) * (size_t)D);
        }
        float *y = lv[o] + curW[o] * D;
        if (kind[o - 1] == 0) {            /* cDeltaRegression::processBuffer, deltaRegression.cpp:144-152 */
          float norm = 0.0f;
          for (i = 1; i <= W; i++) norm += (float)i * (float)i;
          norm *= 2.0;
          for (long d = 0; d < D; d++) {
            float num = 0.0f;
            for (i = 1; i <= W; i++) num += (float)i * (blk[(W + i) * D + d] - blk[(W - i) * D + d]);
            y[d] = num / norm;
          }
        } else if (kind[o - 1] == 3) {     /* cDeltaRegression with onlyInSegments (zero = no value), deltaRegression.cpp:121-137:
                                              `norm` is a member that grows by i^2 for every pair it uses -- rows in order, columns in order */
          if (!seg_norm_set[o]) {
            seg_norm[o] = 0.0f;
            for (i = 1; i <= W; i++) seg_norm[o] += (float)i * (float)i;
            seg_norm[o] *= 2.0;
            seg_norm_set[o] = 1;
          }
          for (long d = 0; d < D; d++) {
            float num = 0.0f;
            for (i = 1; i <= W; i++) {
              const float a = blk[(W - i) * D + d], b = blk[(W + i) * D + d];
              if (!(b == 0.0 || b != b || a == 0.0 || a != a)) {
                num += (float)i * (b - a);
                seg_norm[o] += (float)i * (float)i;
              }
            }
            y[d] = (seg_norm[o] != 0.0) ? num / seg_norm[o] : 0.0f;
          }
        } else if (kind[o - 1] == 2) {     /* cContourSmoother with noZeroSma, contourSmoother.cpp:91-104 */
          for (long d = 0; d < D; d++) {
            const float c0 = blk[W * D + d];
            if (c0 != 0.0) {
              long N = 1;
              float acc = c0;
              for (i = 1; i <= W; i++) {
                if (blk[(W - i) * D + d] != 0.0) { acc += blk[(W - i) * D + d]; N++; }
                if (blk[(W + i) * D + d] != 0.0) { acc += blk[(W + i) * D + d]; N++; }
              }
              y[d] = acc / (float)N;
            } else {
              y[d] = 0.0f;
            }
          }
        } else {                           /* cContourSmoother::processBuffer, contourSmoother.cpp:104-111 */
          int smaWin = 2 * W + 1;
          for (long d = 0; d < D; d++) {
            float acc = blk[W * D + d];
            for (i = 1; i <= W; i++) { acc += blk[(W - i) * D + d]; acc += blk[(W + i) * D + d]; }
            acc /= (float)smaWin;
            y[d] = acc;
          }
        }
        curW[o]++;
        progress = 1;
      }
    }
  }
  long cum = T;
  for (o = 1; o <= n_stages; o++) {
    cum += Wv[o - 1];
    memcpy(levels_out[o - 1], lv[o], sizeof(float) * (size_t)cum * (size_t)D);
  }
  for (o = 0; o <= n_stages; o++) free(lv[o]);
  free(lv); free(curW); free(done); free(blk);
}

/* ------------------------------------------------------------------- chain */
/* config/is09-13/IS09_emotion_core.lld.conf.inc: LLD columns
 *   [pcm_RMSenergy | mfcc 1..12 | pcm_zcr | voiceProb | F0] -> SMA(3) = level is09_lld,
 *   delta(2) of that = is09_lld_de; the LLD sinks read lld;lld_de, so the file
 * has T+1 rows of 32 columns. out: (T+1) x 32; returns T+1 (0 if T == 0).
 * raw16 (optional, T x 16): the 16 columns before SMA. */
long lldo_is09_chain(const int16_t *pcm, long n_samples, float *out, float *raw16)
{
  lldo_mfcc_cfg c;
  lldo_default_mfcc12_cfg(&c);
  c.use_power = 0; c.first_mfcc = 1; c.last_mfcc = 12; c.n_delta = 0;
  lldo_geom g;
  lldo_geometry(&c, &g);
  long T = lldo_num_frames(n_samples, g.N, g.H);
  if (!out || T <= 0) return T > 0 ? T + 1 : 0;
  const int D = 16;
  float *x = (float *)malloc(sizeof(float) * (size_t)n_samples);
  lldo_pcm16_to_float(pcm, n_samples, x);
  double *w = (double *)malloc(sizeof(double) * (size_t)g.N);
  lldo_window_table(c.win_func, g.N, c.win_sigma, c.win_gain, w);
  lldo_mel mel; lldo_dct dct;
  lldo_mel_init(&mel, g.K, g.frame_size_sec_fft, c.n_bands, c.lofreq, c.hifreq, c.use_power, c.mel_htk_compatible);
  lldo_mfcc_init(&dct, c.n_bands, c.first_mfcc, c.last_mfcc, c.cep_lifter, c.mfcc_htk_compatible, c.melfloor);
  float *fr = (float *)malloc(sizeof(float) * (size_t)g.N);
  float *sp = (float *)malloc(sizeof(float) * (size_t)g.Nfft);
  float *mg = (float *)malloc(sizeof(float) * (size_t)g.K);
  float *mb = (float *)malloc(sizeof(float) * (size_t)c.n_bands);
  float *acf = (float *)malloc(sizeof(float) * (size_t)(g.K - 1));
  float *cep = (float *)malloc(sizeof(float) * (size_t)(g.K - 1));
  float *lld = (float *)calloc((size_t)T * D, sizeof(float));
  lldo_pitch_state st;
  lldo_pitch_state_init(&st);
  float fsSec = (float)g.frame_size_sec_fft;      /* level frameSizeSec of is09_acf (inherited from is09_fftmag) */
  for (long t = 0; t < T; t++) {
    const float *src = x + t * g.H;
    float *row = lld + t * D;
    row[13] = lldo_zcr(src, g.N);                                   /* is09_mzcr reads is09_frames */
    lldo_preemphasis(src, fr, g.N, c.preemph_k, c.preemph_de);
    lldo_window_apply(fr, fr, g.N, w, c.win_offset);
    row[0] = lldo_energy_rms(fr, g.N);                              /* is09_energy reads is09_winframe */
    lldo_rfft_frame(fr, g.N, sp, g.Nfft, c.zero_pad_symmetric);
    lldo_fftmag(sp, g.Nfft, mg);
    lldo_melspec(&mel, mg, mb);
    lldo_mfcc(&dct, mb, row + 1);
    lldo_acf(mg, g.K, 0, acf);
    lldo_acf(mg, g.K, 1, cep);
    lldo_pitch_acf(acf, cep, g.K - 1, fsSec, 500.0, 0.55, &st, &row[14], &row[15], 0);
  }
  if (raw16) memcpy(raw16, lld, sizeof(float) * (size_t)T * D);
  int kind[2] = {1, 0}, Wv[2] = {1, 2};
  float *lv[2];
  lv[0] = (float *)malloc(sizeof(float) * (size_t)(T + 1) * D);
  lv[1] = (float *)malloc(sizeof(float) * (size_t)(T + 3) * D);
  lldo_window_chain(lld, T, D, 2, kind, Wv, lv);
  for (long t = 0; t < T + 1; t++) {
    memcpy(out + t * 2 * D, lv[0] + t * D, sizeof(float) * D);
    memcpy(out + t * 2 * D + D, lv[1] + t * D, sizeof(float) * D);
  }
  free(lv[0]); free(lv[1]);
  free(x); free(w); free(fr); free(sp); free(mg); free(mb); free(acf); free(cep); free(lld);
  lldo_mel_free(&mel); lldo_mfcc_free(&dct);
  return T + 1;
}
