/*
 * lld_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY (see lld_oracle.h).
 *
 * Restates, function by function, the arithmetic of the reference's LLD hot
 * path. Citations are relative to the reference root (audeering/opensmile
 * v3.0.2). Compile with -ffp-contract=off (oracle/Makefile does).
 */
#include "lld_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ config */

/* The sample rate of the input the chain functions below work on (cWaveSource reads it from the file; every frame size of
 * the shipped configs is given in seconds, so the same conf runs at any rate). 16 kHz unless a test sets another one. */
double g_lldo_rate = 16000.0;
void lldo_set_sample_rate(double rate) { g_lldo_rate = rate > 0.0 ? rate : 16000.0; }
double lldo_get_sample_rate(void) { return g_lldo_rate; }

/* config/mfcc/MFCC12_0_D_A.conf:52-131 */
void lldo_default_mfcc12_cfg(lldo_mfcc_cfg *c)
{
  memset(c, 0, sizeof(*c));
  c->sample_rate = g_lldo_rate;
  c->frame_size_sec = 0.0250;
  c->frame_step_sec = 0.010;
  c->preemph_enable = 1; c->preemph_k = (float)0.97; c->preemph_de = 0;
  c->win_func = LLDO_WIN_HAMM; c->win_sigma = 0.4; c->win_gain = 1.0; c->win_offset = 0.0;
  c->zero_pad_symmetric = 0;
  c->n_bands = 26; c->lofreq = 0.0f; c->hifreq = 8000.0f; c->use_power = 1;
  c->mel_htk_compatible = 1;
  c->first_mfcc = 0; c->last_mfcc = 12; c->cep_lifter = 22.0f;
  c->mfcc_htk_compatible = 1; c->melfloor = 0.00000001f;
  c->n_delta = 2; c->delta_win = 2;
}

/* smileMath_ceilToNextPowOf2, src/smileutil/smileUtil.c:726-731 */
static long ceil_pow2(long x)
{
  long y = 1;
  while (y < x) y *= 2;
  return y;
}

/* R1. cWinToVecProcessor::configureWriter, src/core/winToVecProcessor.cpp:435-456:
 *   frameSizeFrames = round(frameSize / T), frameStepFrames = round(frameStep / T)
 * with T = 1.0/(double)sampleRate (src/iocore/waveSource.cpp:190);
 * the level keeps frameSizeSec = frameSize as configured (:562-568).
 * cTransformFFT::configureWriter rescales it by Nfft/N
 * (src/dspcore/transformFft.cpp:66-96) and pads to the next power of two,
 * minimum 4 (:119-137). */
void lldo_geometry(const lldo_mfcc_cfg *c, lldo_geom *g)
{
  double T = 1.0 / (double)(long)c->sample_rate;
  g->N = (long)round(c->frame_size_sec / T);
  double step = c->frame_step_sec;
  if (step == 0.0) step = c->frame_size_sec;
  g->H = (long)round(step / T);
  if (g->H == 0) g->H = g->N;
  long nfft = g->N;
  double fss = c->frame_size_sec;
  if (nfft & (nfft - 1)) {
    nfft = ceil_pow2(g->N);
    fss *= (double)nfft / (double)g->N;
  }
  if (nfft < 4) nfft = 4;
  g->Nfft = nfft;
  g->K = nfft / 2 + 1;
  g->frame_size_sec_fft = fss;
}

/* R1. Only complete frames are emitted when noPostEOIprocessing=1
 * (src/core/winToVecProcessor.cpp:872-877): frame t covers samples
 * [t*H, t*H+N) (frameMode=fixed, frameCenterSpecial=left => pre=0). */
long lldo_num_frames(long n_samples, long N, long H)
{
  if (n_samples < N) return 0;
  return (n_samples - N) / H + 1;
}

/* ---------------------------------------------------------------------- R0 */
/* smilePcm_convertSamples, 16-bit mono-mixdown branch,
 * src/smileutil/smileUtil.c:2527-2535: (tmp/(float)nChan)/(float)32767.0 */
/* R0, every sample format: smilePcm_convertSamples, smileUtil.c:2500-2627. n sample frames of
 * n_chan interleaved channels; mixdown -> n floats, else n*n_chan floats. Returns n, 0 on an
 * unknown format. */
static float pcm_sample(const unsigned char *buf, long idx, int n_bps, int n_bits)
{
  switch (n_bps) {
    case 1: return (float)((const int8_t *)buf)[idx];
    case 2: return (float)((const int16_t *)buf)[idx];
    case 3: {
      uint32_t is = 0;
      is |= (uint32_t)(buf[idx * 3]) << 8;
      is |= (uint32_t)(buf[idx * 3 + 1]) << 16;
      is |= (uint32_t)(buf[idx * 3 + 2]) << 24;
      return (float)((int32_t)is >> 8);
    }
    default: {
      const int32_t v = ((const int32_t *)buf)[idx];
      return (n_bits == 24) ? (float)(v & 0xFFFFFF) : (float)v;
    }
  }
}
long lldo_pcm_convert(const void *buf, int n_bps, int n_bits, int n_chan, int mixdown, long n, float *out)
{
  float fs;
  if (n_bps == 1) fs = (float)127.0;
  else if (n_bps == 2) fs = (float)32767.0;
  else if (n_bps == 3 || (n_bps == 4 && n_bits == 24)) fs = (float)(32767.0 * 256.0);
  else if (n_bps == 4 && n_bits == 32) fs = (float)2147483647.0;
  else return 0;
  const unsigned char *b = (const unsigned char *)buf;
  if (mixdown) {
    for (long i = 0; i < n; i++) {
      float tmp = 0.0f;
      for (int c = 0; c < n_chan; c++) tmp += pcm_sample(b, i * n_chan + c, n_bps, n_bits);
      out[i] = (tmp / (float)n_chan) / fs;
    }
  } else {
    for (long i = 0; i < n * n_chan; i++) out[i] = pcm_sample(b, i, n_bps, n_bits) / fs;
  }
  return n;
}

void lldo_pcm16_to_float(const int16_t *pcm, long n, float *out)
{
  for (long i = 0; i < n; i++) {
    float tmp = 0.0f;
    tmp += (float)pcm[i];
    out[i] = (tmp / (float)1) / (float)32767.0;
  }
}

/* ---------------------------------------------------------------------- R2 */
/* cVectorPreemphasis::processVector, src/dspcore/vectorPreemphasis.cpp:89-107 */
void lldo_preemphasis(const float *src, float *dst, long N, float k, int de)
{
  /* dst may alias src in the reference; we compute from a saved copy of the
   * previous input sample so aliasing is harmless. */
  float prev = src[0];
  dst[0] = (1 - k) * src[0];
  for (long n = 1; n < N; n++) {
    float cur = src[n];
    if (de) dst[n] = cur + k * prev;
    else    dst[n] = cur - k * prev;
    prev = cur;
  }
}

/* ---------------------------------------------------------------------- R3 */
/* smileDsp_win*, src/smileutil/smileUtil.c:1218-1350; gain applied as in
 * cWindower::precomputeWinFunc, src/dspcore/windower.cpp:159-218
 * (squareRoot / fade / xshift are not restated: unused by the configs) */
int lldo_window_table(int func, long N, double sigma, double gain, double *w)
{
  double NN = (double)N;
  long n;
  double i;
  switch (func) {
  case LLDO_WIN_RECT:
    for (n = 0; n < N; n++) w[n] = 1.0;
    break;
  case LLDO_WIN_HANN:      /* :1277-1289 */
    for (n = 0, i = 0.0; n < N; n++, i += 1.0)
      w[n] = 0.5 * (1.0 - cos((2.0 * M_PI * i) / (NN - 1.0)));
    break;
  case LLDO_WIN_HAMM:      /* :1291-1304 */
    for (n = 0, i = 0.0; n < N; n++, i += 1.0)
      w[n] = 0.54 - 0.46 * cos((2.0 * M_PI * i) / (NN - 1.0));
    break;
  case LLDO_WIN_SINE:      /* :1306-1318 */
    for (n = 0, i = 0.0; n < N; n++, i += 1.0)
      w[n] = sin((1.0 * M_PI * i) / (NN - 1.0));
    break;
  case LLDO_WIN_GAUSS: {   /* :1334-1350 */
    if (sigma <= 0.0) sigma = 0.01;
    if (sigma > 0.5) sigma = 0.5;
    for (n = 0, i = 0.0; n < N; n++, i += 1.0) {
      double tmp = (i - (NN - 1.0) / 2.0) / (sigma * (NN - 1.0) / 2.0);
      w[n] = exp(-0.5 * (tmp * tmp));
    }
    break; }
  case LLDO_WIN_TRI:       /* :1231-1246 */
    for (n = 0; n < N / 2; n++) w[n] = 2.0 * (double)(n + 1) / (double)N;
    for (n = N / 2; n < N; n++) w[n] = 2.0 * (double)(N - n) / (double)N;
    break;
  case LLDO_WIN_BARTLETT:  /* :1261-1275 */
    for (n = 0; n < N / 2; n++) w[n] = 2.0 * (double)(n) / (double)(N - 1);
    for (n = N / 2; n < N; n++) w[n] = 2.0 * (double)(N - 1 - n) / (double)(N - 1);
    break;
  case LLDO_WIN_LANCZOS:   /* :1320-1332, smileDsp_lcSinc :1208-1213 */
    for (n = 0, i = 0.0; n < N; n++, i += 1.0) {
      double y = M_PI * ((2.0 * i) / (NN - 1.0) - 1.0);
      w[n] = sin(y) / (y);
    }
    break;
  default:
    return 0;
  }
  if (gain != 1.0) for (n = 0; n < N; n++) w[n] *= gain;
  return 1;
}

/* cWindower::processVector, src/dspcore/windower.cpp:221-229:
 *   dst = src * (FLOAT_DMEM)w + (FLOAT_DMEM)offset   (w is double, cast per use) */
void lldo_window_apply(const float *src, float *dst, long N, const double *w, double offset)
{
  for (long n = 0; n < N; n++) dst[n] = src[n] * (float)w[n] + (float)offset;
}

/* ---------------------------------------------------------------------- R4 */
static lldo_rfft_fn g_rfft_hook = 0;
void lldo_set_rfft_hook(lldo_rfft_fn fn) { g_rfft_hook = fn; }

/* Built-in float32 real FFT producing the reference's packed layout
 *   a[0]=Re X0, a[1]=Re X[n/2], a[2k]=Re Xk, a[2k+1]=+sum x sin(2 pi jk/n)
 * (Ooura's sign convention, src/dspcore/fftsg.c:103-135). This is a plain
 * iterative radix-2 complex FFT of the full length (own code, not Ooura's
 * split-radix): its round-off differs from the reference's by O(1e-7)
 * relative to the spectrum's largest bin. */
static void own_rfft_packed(float *a, long n)
{
  float *re = (float *)malloc(sizeof(float) * 2 * (size_t)n);
  float *im = re + n;
  long i, j, len;
  /* bit reversal load */
  int bits = 0;
  while ((1L << bits) < n) bits++;
  for (i = 0; i < n; i++) {
    long r = 0;
    for (j = 0; j < bits; j++) if (i & (1L << j)) r |= 1L << (bits - 1 - j);
    re[r] = a[i]; im[r] = 0.0f;
  }
  for (len = 2; len <= n; len <<= 1) {
    long half = len >> 1;
    for (j = 0; j < half; j++) {
      double ang = -2.0 * M_PI * (double)j / (double)len;
      float wr = (float)cos(ang), wi = (float)sin(ang);
      for (i = j; i < n; i += len) {
        float xr = re[i + half], xi = im[i + half];
        float tr = xr * wr - xi * wi;
        float ti = xr * wi + xi * wr;
        re[i + half] = re[i] - tr; im[i + half] = im[i] - ti;
        re[i] = re[i] + tr;        im[i] = im[i] + ti;
      }
    }
  }
  a[0] = re[0];
  a[1] = re[n / 2];
  for (i = 1; i < n / 2; i++) { a[2 * i] = re[i]; a[2 * i + 1] = -im[i]; }
  free(re);
}

/* cTransformFFT::processVector (forward), src/dspcore/transformFft.cpp:165-223 */
void lldo_rfft_frame(const float *src, long Nsrc, float *dst, long Nfft, int zero_pad_symmetric)
{
  long i;
  if (zero_pad_symmetric) {                       /* :177-187 */
    long padlen2 = (Nfft - Nsrc) / 2;
    for (i = 0; i < padlen2; i++) dst[i] = 0;
    for (i = 0; i < Nsrc; i++) dst[i + padlen2] = src[i];
    for (i = Nsrc + padlen2; i < Nfft; i++) dst[i] = 0;
  } else {                                        /* :188-195 */
    for (i = 0; i < Nsrc; i++) dst[i] = src[i];
    for (i = Nsrc; i < Nfft; i++) dst[i] = 0;
  }
  if (g_rfft_hook) {
    /* work areas sized as in transformFft.cpp:201-208 */
    int *ip = (int *)calloc(1, sizeof(int) * (3 + (size_t)ceil(sqrt((float)Nfft))));
    float *w = (float *)calloc(1, sizeof(float) * (size_t)(Nfft / 2 + 1));
    g_rfft_hook((int)Nfft, 1, dst, ip, w);
    free(ip); free(w);
  } else if (Nfft >= 64) {
    lldo_ooura_rdft((int)Nfft, 1, dst);   /* the reference's operation order, lld_oracle_fft.c */
  } else {
    own_rfft_packed(dst, Nfft);           /* n < 64: own radix-2 order (no BASELINE config gets here) */
  }
}

/* ---------------------------------------------------------------------- R5 */
/* cFFTmagphase::processVector, magnitude branch,
 * src/dspcore/fftmagphase.cpp:215-221 */
void lldo_fftmag(const float *p, long Nfft, float *mag)
{
  mag[0] = fabsf(p[0]);
  for (long n = 2; n < Nfft; n += 2)
    mag[n / 2] = sqrtf(p[n] * p[n] + p[n + 1] * p[n + 1]);
  mag[Nfft / 2] = fabsf(p[1]);
}

/* ---------------------------------------------------------------------- R6 */
/* smileDsp_specScaleTransfFwd, SPECTSCALE_MEL, src/smileutil/smileUtil.c:1138-1141 */
static double hz_to_mel(double x)
{
  if (x > 0.0) return 1127.0 * log(1.0 + x / 700.0);
  return 0.0;
}
/* cMelspec::NtoFmel, src/include/lldcore/melspec.hpp:119-122 */
static float n_to_fmel(long n, float F0)
{
  return (float)hz_to_mel((double)(((float)n) * F0));
}

/* cMelspec::computeFilters, standard bank,
 * src/lldcore/melspec.cpp:184-238 (limits) and :391-449 (centres, channel
 * map, weights). htkcompatible forces the mel scale (:127-131). */
int lldo_mel_init(lldo_mel *m, long K, double frame_size_sec, int n_bands,
                  float lofreq, float hifreq, int use_power, int htk)
{
  long blocksize = K, n;
  int mm;
  if (blocksize < n_bands) return 0;
  m->K = K; m->n_bands = n_bands; m->use_power = use_power; m->htk = htk;
  m->coef = (float *)calloc(1, sizeof(float) * (size_t)blocksize);
  m->chan_map = (long *)malloc(sizeof(long) * (size_t)blocksize);
  m->cfs = (float *)malloc(sizeof(float) * (size_t)(n_bands + 2));

  float N = (float)((blocksize - 1) * 2);
  float F0 = (float)(1.0 / frame_size_sec);
  float Fs = (float)(N / frame_size_sec);
  float M = (float)n_bands;
  if ((lofreq < 0.0) || (lofreq > Fs / 2.0) || (lofreq > hifreq)) lofreq = 0.0;
  if ((hifreq < lofreq) || (hifreq > Fs / 2.0) || (hifreq <= 0.0)) hifreq = Fs / (float)2.0;
  float LoF = (float)hz_to_mel((double)lofreq);
  float HiF = (float)hz_to_mel((double)hifreq);
  /* FtoN, melspec.hpp:107-110 */
  long nLo = (long)round((double)(lofreq / F0));
  long nHi = (long)round((double)(hifreq / F0));
  if (nLo > blocksize) nLo = blocksize;
  if (nHi > blocksize) nHi = blocksize;
  if (nLo < 0) nLo = 0;
  if (nHi < 0) nHi = 0;
  m->nLo = nLo; m->nHi = nHi;

  float mBandw = (HiF - LoF) / (M + (float)1.0);                 /* :395 */
  for (mm = 0; mm <= n_bands + 1; mm++) m->cfs[mm] = LoF + (float)mm * mBandw;

  mm = 0;                                                          /* :428-438 */
  for (n = 0; n < blocksize; n++) {
    if ((n <= nLo) || (n >= nHi)) m->chan_map[n] = -3;
    else {
      while (m->cfs[mm] < n_to_fmel(n, F0)) {
        if (mm > n_bands) break;
        mm++;
      }
      m->chan_map[n] = mm - 2;
    }
  }
  mm = 0;                                                          /* :441-447 */
  for (n = nLo; n < nHi; n++) {
    float nM = n_to_fmel(n, F0);
    while ((nM > m->cfs[mm + 1]) && (mm <= n_bands)) mm++;
    m->coef[n] = (m->cfs[mm + 1] - nM) / (m->cfs[mm + 1] - m->cfs[mm]);
  }
  return 1;
}

void lldo_mel_free(lldo_mel *m) { free(m->coef); free(m->chan_map); free(m->cfs); }

/* cMelspec::processVector (forward), src/lldcore/melspec.cpp:519-570 */
void lldo_melspec(const lldo_mel *m, const float *mag, float *out)
{
  long n;
  int b;
  float *p = (float *)malloc(sizeof(float) * (size_t)m->K);
  if (m->use_power) for (n = 0; n < m->K; n++) p[n] = mag[n] * mag[n];     /* :520-527 */
  else              for (n = 0; n < m->K; n++) p[n] = mag[n];
  for (b = 0; b < m->n_bands; b++) out[b] = 0.0f;
  for (n = m->nLo; n < m->nHi; n++) {                                      /* :544-553 */
    long ch = m->chan_map[n];
    double a = (double)p[n] * (double)m->coef[n];
    if (ch > -2) {
      if (ch > -1) out[ch] += (float)a;
      if (ch < m->n_bands - 1) out[ch + 1] += p[n] - (float)a;
    }
  }
  if (m->htk) {                                                            /* :559-570 */
    for (b = 0; b < m->n_bands; b++) {
      if (m->use_power) out[b] *= (float)(32767.0 * 32767.0);
      else              out[b] *= (float)32767.0;
    }
  }
  free(p);
}

/* ---------------------------------------------------------------------- R7 */
/* cMfcc::initTables, src/lldcore/mfcc.cpp:136-170; melfloor forced to 1.0
 * when htkcompatible (:88-91) */
int lldo_mfcc_init(lldo_dct *d, int n_bands, int first, int last, float cep_lifter, int htk, float melfloor)
{
  int i, m;
  d->n_bands = n_bands; d->first = first; d->last = last; d->htk = htk;
  d->n_mfcc = last - first + 1;
  d->melfloor = htk ? 1.0f : melfloor;
  d->costable = (float *)malloc(sizeof(float) * (size_t)n_bands * (size_t)d->n_mfcc);
  d->sintable = (float *)malloc(sizeof(float) * (size_t)d->n_mfcc);
  double fnM = (double)n_bands;
  for (i = first; i <= last; i++) {
    double fi = (double)i;
    for (m = 0; m < n_bands; m++)
      d->costable[m + (i - first) * n_bands] =
          (float)cos((double)M_PI * (fi / fnM) * ((double)(m) + (double)0.5));
  }
  if (cep_lifter > 0.0) {
    for (i = first; i <= last; i++)
      d->sintable[i - first] = ((float)1.0 + cep_lifter / (float)2.0 *
                                sinf((float)M_PI * ((float)(i)) / cep_lifter));
  } else {
    for (i = first; i <= last; i++) d->sintable[i - first] = 1.0f;
  }
  return 1;
}

void lldo_mfcc_free(lldo_dct *d) { free(d->costable); free(d->sintable); }

/* cMfcc::processVector (forward), src/lldcore/mfcc.cpp:239-273.
 * HTK order (:255-258): with htkcompatible && firstMfcc==0 the output is
 * c1..c_last followed by c0. */
void lldo_mfcc(const lldo_dct *d, const float *mel, float *out)
{
  int i, m, N = d->n_bands;
  float *l = (float *)malloc(sizeof(float) * (size_t)N);
  for (i = 0; i < N; i++) {
    if (mel[i] < d->melfloor) l[i] = logf(d->melfloor);
    else l[i] = (float)logf(mel[i]);
  }
  float factor = (float)sqrt((double)2.0 / (double)(N));
  for (i = d->first; i <= d->last; i++) {
    int i0 = i - d->first;
    float *outc = out + i0;
    if (d->htk && (d->first == 0)) {
      if (i == d->last) i0 = 0;
      else i0 += 1;
    }
    *outc = 0.0f;
    for (m = 0; m < N; m++) *outc += l[m] * d->costable[m + i0 * N];
    *outc *= d->sintable[i0] * factor;
  }
  free(l);
}

/* --------------------------------------------------------------------- R13 */
/* cDeltaRegression::processBuffer, src/dspcore/deltaRegression.cpp:113-170
 * (norm :77-79), fed by cWindowProcessor (src/core/windowProcessor.cpp:82-118,
 * 164-229: blocks [t-W, t+1+W)) through cDataMemoryLevel::getMatrix's padding
 * (src/core/dataMemoryLevel.cpp:1687-1712: indices <0 replicate the first
 * frame, indices past the end replicate the last frame at end-of-input, and
 * validateIdxRangeR :1022-1029 refuses a block that is ALL padding). Hence a
 * level holding T frames yields T+W output frames. */
long lldo_delta_regression(const float *x, long T, long D, int W, float *y)
{
  long t, d;
  int i;
  float norm = 0.0f;
  if (T <= 0) return 0;
  for (i = 1; i <= W; i++) norm += (float)i * (float)i;
  norm *= 2.0;
  for (t = 0; t < T + W; t++) {
    for (d = 0; d < D; d++) {
      float num = 0.0f;
      for (i = 1; i <= W; i++) {
        long a = t - i, b = t + i;
        if (a < 0) a = 0;
        if (a > T - 1) a = T - 1;
        if (b > T - 1) b = T - 1;
        float delta = x[b * D + d] - x[a * D + d];
        num += (float)i * delta;
      }
      y[t * D + d] = num / norm;
    }
  }
  return T + W;
}

/* Tick-accurate restatement of a chain of n_orders cDeltaRegression
 * components (delta, accel, ...) as the reference's tick loop runs them
 * (src/core/componentManager.cpp:1416-1590: every component ticks once per
 * loop iteration in instance order; when nothing ran, EOI is set and the loop
 * continues). Each window processor emits ONE frame per tick (blocksize=1)
 * from the block [t-W, t+W+1) of its input level:
 *   - before EOI the block must be fully written (validateIdxRangeR,
 *     src/core/dataMemoryLevel.cpp:1005-1049);
 *   - at EOI a block reaching past the written data is padded, unless it is
 *     ALL padding (:1022-1026) -- then the component stops;
 *   - getMatrix (:1687-1712) replicates the first frame for indices < 0; in
 *     THAT branch (t < W) it reads indices up to t+W raw from the buffer even
 *     past the write pointer (zero-initialised, never-written slots), while
 *     for t >= W it replicates the last WRITTEN frame (:1699-1708).
 * For T >= 4 this reduces to lldo_delta_regression() applied order by order;
 * for T <= 3 the lockstep timing and the raw-read branch change the values
 * (verified against the real binary, tests/test_oracle_pin.py).
 * x: T x D. out: n_orders blocks of T x D (order-major), i.e. the rows that
 * survive cVectorConcat. */
void lldo_delta_chain(const float *x, long T, long D, int W, int n_orders, float *out)
{
  if (T <= 0 || n_orders <= 0) return;
  int o, i;
  long cap = T + (long)W * n_orders + 2 * W + 2;     /* frames per level buffer */
  float **lv = (float **)calloc((size_t)n_orders + 1, sizeof(float *));
  long *curW = (long *)calloc((size_t)n_orders + 1, sizeof(long));
  long *curR = (long *)calloc((size_t)n_orders + 1, sizeof(long));   /* next frame of component o */
  char *done = (char *)calloc((size_t)n_orders + 1, 1);
  float norm = 0.0f;
  for (i = 1; i <= W; i++) norm += (float)i * (float)i;
  norm *= 2.0;
  lv[0] = (float *)calloc((size_t)cap * (size_t)D, sizeof(float));
  memcpy(lv[0], x, sizeof(float) * (size_t)T * (size_t)D);
  curW[0] = T;
  for (o = 1; o <= n_orders; o++) lv[o] = (float *)calloc((size_t)cap * (size_t)D, sizeof(float));
  float *blk = (float *)malloc(sizeof(float) * (size_t)(2 * W + 1) * (size_t)D);

  for (int eoi = 0; eoi <= 1; eoi++) {
    int progress = 1;
    while (progress) {
      progress = 0;
      for (o = 1; o <= n_orders; o++) {
        if (done[o]) continue;
        const float *in = lv[o - 1];
        long wIn = curW[o - 1];
        long t = curR[o];
        long vOld = t - W, vEnd = t + W + 1;
        long v = vOld < 0 ? 0 : vOld;
        long padEnd = 0;
        if (vEnd > wIn) {
          if (!eoi) continue;
          padEnd = vEnd - wIn;
          if (padEnd >= vEnd - v) { done[o] = 1; continue; }
        }
        if (!(v < wIn)) continue;                      /* OOR_right: nothing to read yet */
        if (curW[o] >= cap) { done[o] = 1; continue; }
        /* assemble the block exactly as getMatrix does */
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
        /* cDeltaRegression::processBuffer, deltaRegression.cpp:144-152 */
        float *y = lv[o] + curW[o] * D;
        for (long d = 0; d < D; d++) {
          float num = 0.0f;
          for (i = 1; i <= W; i++) {
            float delta = blk[(W + i) * D + d] - blk[(W - i) * D + d];
            num += (float)i * delta;
          }
          y[d] = num / norm;
        }
        curW[o]++; curR[o]++;
        progress = 1;
      }
    }
  }
  for (o = 1; o <= n_orders; o++)
    memcpy(out + (size_t)(o - 1) * (size_t)T * (size_t)D, lv[o], sizeof(float) * (size_t)T * (size_t)D);
  for (o = 0; o <= n_orders; o++) free(lv[o]);
  free(lv); free(curW); free(curR); free(done); free(blk);
}

/* ------------------------------------------------------------------- chain */
long lldo_mfcc_chain(const lldo_mfcc_cfg *c, const int16_t *pcm, long n_samples, float *out,
                     float *tap_win, float *tap_fft, float *tap_mag, float *tap_mel)
{
  lldo_geom g;
  lldo_geometry(c, &g);
  long T = lldo_num_frames(n_samples, g.N, g.H);
  if (!out || T <= 0) return T;

  int D = c->last_mfcc - c->first_mfcc + 1;
  int Dtot = D * (1 + c->n_delta);
  float *x = (float *)malloc(sizeof(float) * (size_t)(n_samples > 0 ? n_samples : 1));
  lldo_pcm16_to_float(pcm, n_samples, x);

  double *w = (double *)malloc(sizeof(double) * (size_t)g.N);
  lldo_window_table(c->win_func, g.N, c->win_sigma, c->win_gain, w);
  lldo_mel mel; lldo_dct dct;
  lldo_mel_init(&mel, g.K, g.frame_size_sec_fft, c->n_bands, c->lofreq, c->hifreq,
                c->use_power, c->mel_htk_compatible);
  lldo_mfcc_init(&dct, c->n_bands, c->first_mfcc, c->last_mfcc, c->cep_lifter,
                 c->mfcc_htk_compatible, c->melfloor);

  float *fr = (float *)malloc(sizeof(float) * (size_t)g.N);
  float *sp = (float *)malloc(sizeof(float) * (size_t)g.Nfft);
  float *mg = (float *)malloc(sizeof(float) * (size_t)g.K);
  float *mb = (float *)malloc(sizeof(float) * (size_t)c->n_bands);
  float *cep = (float *)malloc(sizeof(float) * (size_t)T * (size_t)D);

  for (long t = 0; t < T; t++) {
    const float *src = x + t * g.H;                 /* cFramer::doProcess = memcpy, framer.cpp:60-66 */
    if (c->preemph_enable) lldo_preemphasis(src, fr, g.N, c->preemph_k, c->preemph_de);
    else memcpy(fr, src, sizeof(float) * (size_t)g.N);
    lldo_window_apply(fr, fr, g.N, w, c->win_offset);
    if (tap_win) memcpy(tap_win + t * g.N, fr, sizeof(float) * (size_t)g.N);
    lldo_rfft_frame(fr, g.N, sp, g.Nfft, c->zero_pad_symmetric);
    if (tap_fft) memcpy(tap_fft + t * g.Nfft, sp, sizeof(float) * (size_t)g.Nfft);
    lldo_fftmag(sp, g.Nfft, mg);
    if (tap_mag) memcpy(tap_mag + t * g.K, mg, sizeof(float) * (size_t)g.K);
    lldo_melspec(&mel, mg, mb);
    if (tap_mel) memcpy(tap_mel + t * c->n_bands, mb, sizeof(float) * (size_t)c->n_bands);
    lldo_mfcc(&dct, mb, cep + t * D);
  }

  /* R13: tick-accurate delta / accel chain, truncated to T rows by
   * cVectorConcat (src/other/vectorConcat.cpp:44-49,
   * src/core/dataReader.cpp:446-522). */
  for (long t = 0; t < T; t++) memcpy(out + t * Dtot, cep + t * D, sizeof(float) * (size_t)D);
  if (c->n_delta > 0) {
    float *de = (float *)malloc(sizeof(float) * (size_t)T * (size_t)D * (size_t)c->n_delta);
    lldo_delta_chain(cep, T, D, c->delta_win, c->n_delta, de);
    for (int o = 1; o <= c->n_delta; o++)
      for (long t = 0; t < T; t++)
        memcpy(out + t * Dtot + o * D, de + ((size_t)(o - 1) * (size_t)T + (size_t)t) * (size_t)D,
               sizeof(float) * (size_t)D);
    free(de);
  }

  free(x); free(w); free(fr); free(sp); free(mg); free(mb); free(cep);
  lldo_mel_free(&mel); lldo_mfcc_free(&dct);
  return T;
}

/* table export for table-level parity tests (tests/test_host_logic.py) */
void lldo_export_tables(const lldo_mfcc_cfg *c, float *win, float *melcoef, int *chan, float *costable, float *lifter)
{
  lldo_geom g;
  lldo_geometry(c, &g);
  double *w = (double *)malloc(sizeof(double) * (size_t)g.N);
  lldo_window_table(c->win_func, g.N, c->win_sigma, c->win_gain, w);
  for (long n = 0; n < g.N; n++) win[n] = (float)w[n];
  free(w);
  lldo_mel mel; lldo_dct dct;
  lldo_mel_init(&mel, g.K, g.frame_size_sec_fft, c->n_bands, c->lofreq, c->hifreq, c->use_power, c->mel_htk_compatible);
  for (long n = 0; n < g.K; n++) { melcoef[n] = mel.coef[n]; chan[n] = (int)mel.chan_map[n]; }
  lldo_mfcc_init(&dct, c->n_bands, c->first_mfcc, c->last_mfcc, c->cep_lifter, c->mfcc_htk_compatible, c->melfloor);
  memcpy(costable, dct.costable, sizeof(float) * (size_t)dct.n_mfcc * (size_t)c->n_bands);
  memcpy(lifter, dct.sintable, sizeof(float) * (size_t)dct.n_mfcc);
  lldo_mel_free(&mel); lldo_mfcc_free(&dct);
}
