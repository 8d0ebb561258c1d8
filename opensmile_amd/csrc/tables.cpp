// Host-side table generation (see tables.hpp). Each function names the
// reference lines whose precision recipe it follows.
#include "tables.hpp"

#include <cmath>

namespace smilehip {

static int64_t next_pow2(int64_t x) {
  int64_t y = 1;
  while (y < x) y <<= 1;
  return y;
}

// Framing integers: src/core/winToVecProcessor.cpp:435-456 (round(frameSize/T)),
// T from src/iocore/waveSource.cpp:190; FFT length and the rescaled
// frameSizeSec: src/dspcore/transformFft.cpp:66-96,119-137.
int make_geometry(const smilehip_lld_config &c, Geometry &g) {
  if (!(c.sample_rate >= 1.0) || !(c.frame_size_sec > 0.0) || c.frame_step_sec < 0.0)
    return SMILEHIP_ERR_INVALID;
  const double T = 1.0 / static_cast<double>(static_cast<long>(c.sample_rate));
  g.period = T;
  g.N = std::lround(c.frame_size_sec / T);
  const double step = (c.frame_step_sec == 0.0) ? c.frame_size_sec : c.frame_step_sec;
  g.H = std::lround(step / T);
  if (c.force_frame_size > 0) g.N = c.force_frame_size;   // single-component plan: size of the input field
  if (g.H == 0) g.H = g.N;
  if (g.N < 1) return SMILEHIP_ERR_INVALID;
  g.frame_period = step;
  int64_t nfft = g.N;
  double fss = c.frame_size_sec;
  if ((nfft & (nfft - 1)) != 0) {
    nfft = next_pow2(g.N);
    fss *= static_cast<double>(nfft) / static_cast<double>(g.N);
  }
  if (nfft < 4) nfft = 4;
  g.Nfft = nfft;
  g.K = nfft / 2 + 1;
  g.fft_frame_size_sec = (c.force_fft_frame_size_sec > 0.0) ? c.force_fft_frame_size_sec : fss;
  return SMILEHIP_OK;
}

// Window shapes: src/smileutil/smileUtil.c:1218-1350 evaluated in double, gain
// folded in double (src/dspcore/windower.cpp:193-197), then the per-use
// (FLOAT_DMEM) cast of windower.cpp:226 applied once here.
int make_window(const smilehip_lld_config &c, int64_t N, std::vector<float> &w) {
  std::vector<double> d(static_cast<size_t>(N));
  const double NN = static_cast<double>(N);
  const double pi = M_PI;
  switch (c.win_func) {
    case SMILEHIP_WIN_RECT:
      for (auto &v : d) v = 1.0;
      break;
    case SMILEHIP_WIN_HANN:
      for (int64_t n = 0; n < N; ++n) d[n] = 0.5 * (1.0 - std::cos((2.0 * pi * double(n)) / (NN - 1.0)));
      break;
    case SMILEHIP_WIN_HAMM:
      for (int64_t n = 0; n < N; ++n) d[n] = 0.54 - 0.46 * std::cos((2.0 * pi * double(n)) / (NN - 1.0));
      break;
    case SMILEHIP_WIN_SINE:
      for (int64_t n = 0; n < N; ++n) d[n] = std::sin((1.0 * pi * double(n)) / (NN - 1.0));
      break;
    case SMILEHIP_WIN_GAUSS: {
      double sigma = c.win_sigma;
      if (sigma <= 0.0) sigma = 0.01;
      if (sigma > 0.5) sigma = 0.5;
      for (int64_t n = 0; n < N; ++n) {
        const double t = (double(n) - (NN - 1.0) / 2.0) / (sigma * (NN - 1.0) / 2.0);
        d[n] = std::exp(-0.5 * (t * t));
      }
      break;
    }
    case SMILEHIP_WIN_TRI:
      for (int64_t n = 0; n < N / 2; ++n) d[n] = 2.0 * double(n + 1) / double(N);
      for (int64_t n = N / 2; n < N; ++n) d[n] = 2.0 * double(N - n) / double(N);
      break;
    case SMILEHIP_WIN_BARTLETT:
      for (int64_t n = 0; n < N / 2; ++n) d[n] = 2.0 * double(n) / double(N - 1);
      for (int64_t n = N / 2; n < N; ++n) d[n] = 2.0 * double(N - 1 - n) / double(N - 1);
      break;
    case SMILEHIP_WIN_LANCZOS:
      for (int64_t n = 0; n < N; ++n) {
        const double y = pi * ((2.0 * double(n)) / (NN - 1.0) - 1.0);
        d[n] = std::sin(y) / y;
      }
      break;
    default:
      return SMILEHIP_ERR_INVALID;
  }
  if (c.win_gain != 1.0)
    for (auto &v : d) v *= c.win_gain;
  w.resize(static_cast<size_t>(N));
  for (int64_t n = 0; n < N; ++n) w[n] = static_cast<float>(d[n]);
  return SMILEHIP_OK;
}

// Mel axis: smileDsp_specScaleTransfFwd, SPECTSCALE_MEL (smileUtil.c:1138-1141)
static double to_mel(double hz) { return hz > 0.0 ? 1127.0 * std::log(1.0 + hz / 700.0) : 0.0; }
// cMelspec::NtoFmel (src/include/lldcore/melspec.hpp:119-122): float product, double mel, float result
static float bin_to_mel(int64_t n, float F0) { return static_cast<float>(to_mel(double(float(n) * F0))); }

// HTK-style triangular bank stored as one weight per bin + channel map:
// cMelspec::computeFilters, src/lldcore/melspec.cpp:184-238 and :391-449.
int make_mel(const smilehip_lld_config &c, const Geometry &g, MelBank &m) {
  const int64_t K = g.K;
  const int nB = c.n_bands;
  if (nB < 1 || K < nB) return SMILEHIP_ERR_INVALID;
  m.n_bands = nB;
  m.coef.assign(static_cast<size_t>(K), 0.0f);
  m.chan.assign(static_cast<size_t>(K), -3);
  m.centres.assign(static_cast<size_t>(nB + 2), 0.0f);

  const float Nf = static_cast<float>((K - 1) * 2);
  const float F0 = static_cast<float>(1.0 / g.fft_frame_size_sec);
  const float Fs = static_cast<float>(Nf / g.fft_frame_size_sec);
  const float M = static_cast<float>(nB);
  float lo = c.lofreq, hi = c.hifreq;
  if ((lo < 0.0) || (lo > Fs / 2.0) || (lo > hi)) lo = 0.0;
  if ((hi < lo) || (hi > Fs / 2.0) || (hi <= 0.0)) hi = Fs / 2.0f;
  const float LoF = static_cast<float>(to_mel(double(lo)));
  const float HiF = static_cast<float>(to_mel(double(hi)));
  int64_t nLo = std::lround(double(lo / F0));
  int64_t nHi = std::lround(double(hi / F0));
  if (nLo > K) nLo = K;
  if (nHi > K) nHi = K;
  if (nLo < 0) nLo = 0;
  if (nHi < 0) nHi = 0;
  m.nLo = nLo;
  m.nHi = nHi;

  const float bw = (HiF - LoF) / (M + 1.0f);
  for (int b = 0; b <= nB + 1; ++b) m.centres[b] = LoF + float(b) * bw;

  int mm = 0;
  for (int64_t n = 0; n < K; ++n) {
    if ((n <= nLo) || (n >= nHi)) {
      m.chan[n] = -3;
    } else {
      while (m.centres[mm] < bin_to_mel(n, F0)) {
        if (mm > nB) break;
        ++mm;
      }
      m.chan[n] = mm - 2;
    }
  }
  mm = 0;
  for (int64_t n = nLo; n < nHi; ++n) {
    const float nM = bin_to_mel(n, F0);
    while ((nM > m.centres[mm + 1]) && (mm <= nB)) ++mm;
    m.coef[n] = (m.centres[mm + 1] - nM) / (m.centres[mm + 1] - m.centres[mm]);
  }

  // Per-band bin ranges. processVector (melspec.cpp:544-553) walks the bins in
  // ascending order and adds p*w to band chan[n] (if > -1) and p - p*w to band
  // chan[n]+1 (if chan[n] > -2 and < nB-1); chan[] is non-decreasing, so each
  // band receives first a run of "rising" bins then a run of "falling" bins.
  m.rise_lo.assign(nB, 0); m.rise_hi.assign(nB, 0);
  m.fall_lo.assign(nB, 0); m.fall_hi.assign(nB, 0);
  for (int b = 0; b < nB; ++b) {
    int64_t rl = -1, rh = -1, fl = -1, fh = -1;
    for (int64_t n = nLo; n < nHi; ++n) {
      const int ch = m.chan[n];
      if (ch <= -2) continue;
      if (ch == b - 1 && ch < nB - 1) { if (rl < 0) rl = n; rh = n + 1; }
      if (ch == b && ch > -1)         { if (fl < 0) fl = n; fh = n + 1; }
    }
    if (rl < 0) rl = rh = 0;
    if (fl < 0) fl = fh = 0;
    // contiguity is what the device loops rely on
    for (int64_t n = rl; n < rh; ++n) if (m.chan[n] != b - 1) return SMILEHIP_ERR_INVALID;
    for (int64_t n = fl; n < fh; ++n) if (m.chan[n] != b) return SMILEHIP_ERR_INVALID;
    if (rh > rl && fh > fl && fl < rh) return SMILEHIP_ERR_INVALID;
    m.rise_lo[b] = int32_t(rl); m.rise_hi[b] = int32_t(rh);
    m.fall_lo[b] = int32_t(fl); m.fall_hi[b] = int32_t(fh);
  }
  // HTK sample scaling, melspec.cpp:559-570
  m.scale = 1.0f;
  if (c.mel_htk_compatible) m.scale = c.use_power ? float(32767.0 * 32767.0) : float(32767.0);
  return SMILEHIP_OK;
}

// DCT-II rows + lifter: cMfcc::initTables, src/lldcore/mfcc.cpp:136-170; output
// ordering and the single float product lifter*factor: mfcc.cpp:246-273.
int make_dct(const smilehip_lld_config &c, DctTables &d) {
  const int nB = c.n_bands;
  d.n_bands = nB;
  d.first = c.first_mfcc;
  d.last = c.last_mfcc;
  d.n_mfcc = d.last - d.first + 1;
  if (d.n_mfcc < 1 || d.first < 0) return SMILEHIP_ERR_INVALID;
  const bool htk = c.mfcc_htk_compatible != 0;
  d.melfloor = htk ? 1.0f : c.melfloor;              // mfcc.cpp:88-91
  d.log_floor = std::log(d.melfloor);                // mfcc.cpp:241
  std::vector<float> costable(size_t(nB) * size_t(d.n_mfcc));
  const double fnM = double(nB);
  for (int i = d.first; i <= d.last; ++i) {
    const double fi = double(i);
    for (int mI = 0; mI < nB; ++mI)
      costable[size_t(mI) + size_t(i - d.first) * nB] =
          float(std::cos(double(M_PI) * (fi / fnM) * (double(mI) + 0.5)));
  }
  d.lifter.assign(d.n_mfcc, 1.0f);
  if (c.cep_lifter > 0.0f) {
    for (int i = d.first; i <= d.last; ++i)
      d.lifter[i - d.first] = 1.0f + c.cep_lifter / 2.0f * std::sin(float(M_PI) * float(i) / c.cep_lifter);
  }
  const float factor = float(std::sqrt(2.0 / double(nB)));
  d.cos_rows.assign(size_t(nB) * size_t(d.n_mfcc), 0.0f);
  d.gain.assign(d.n_mfcc, 0.0f);
  for (int i = d.first; i <= d.last; ++i) {
    const int r = i - d.first;       // output position
    int i0 = r;
    if (htk && d.first == 0) i0 = (i == d.last) ? 0 : r + 1;   // c0 goes last
    for (int mI = 0; mI < nB; ++mI) d.cos_rows[size_t(r) * nB + mI] = costable[size_t(mI) + size_t(i0) * nB];
    d.gain[r] = d.lifter[i0] * factor;
  }
  return SMILEHIP_OK;
}

// deltaRegression.cpp:77-79
float delta_norm(int W) {
  float norm = 0.0f;
  for (int i = 1; i <= W; ++i) norm += float(i) * float(i);
  norm *= 2.0;
  return norm;
}

}  // namespace smilehip

namespace smilehip {

// cSpecScale::dataProcessorCustomFinalise (src/dsp/specScale.cpp:228-300), smileMath_cspline_init /
// smileMath_csplint_init (src/smileutil/smileUtilSpline.c:139-155, 296-342), the level meta data cPitchShs reads
// in setupNewNames (src/lld/pitchShs.cpp:178-204) and its per-harmonic shifts (:235-243). All in double, rounded
// where the reference rounds (the meta data travels as FLOAT_DMEM).
int make_f0_tables(int64_t K, double fft_frame_size_sec, int n_harmonics, float compression, double min_f, F0Host &h) {
  if (K < 4 || n_harmonics < 1 || n_harmonics > 17 || !(min_f > 0.0)) return SMILEHIP_ERR_INVALID;
  const double fsSec = (double)(float)fft_frame_size_sec;
  const double deltaF = 1.0 / fsSec;
  const double minF = min_f;
  const double maxF = deltaF * (double)(K - 1);
  const double l2 = std::log(2.0);
  const double fmin_t = std::log(minF) / l2, fmax_t = std::log(maxF) / l2;
  const double step_t = (fmax_t - fmin_t) / (double)(K - 1);
  std::vector<double> x(static_cast<size_t>(K));
  for (int64_t i = 1; i < K; ++i) x[i] = std::log((double)i * deltaF) / l2;
  x[0] = 2.0 * x[1] - x[2];
  h.sp_rec.assign(size_t(K) * 4, 0.0);
  h.sp_d1.assign(size_t(K), 1.0);
  h.sp_d2.assign(size_t(K), 1.0);
  double dec_prev = 0.0;                                  // y2[0] = 0 (natural boundary)
  for (int64_t i = 1; i < K - 1; ++i) {
    const double sigma = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    h.sp_d1[i] = (x[i + 1] - x[i]) * (x[i + 1] - x[i - 1]);
    h.sp_d2[i] = (x[i] - x[i - 1]) * (x[i + 1] - x[i - 1]);
    const double p = 1.0 / (sigma * dec_prev + 2.0);
    const double dec = (sigma - 1.0) * p;
    h.sp_rec[4 * i + 0] = sigma;
    h.sp_rec[4 * i + 1] = p;
    h.sp_rec[4 * i + 2] = dec;
    dec_prev = dec;
  }
  h.ip_k.assign(size_t(K), 0);
  h.ip_co.assign(size_t(K) * 3, 0.0);
  int64_t hi = 1;
  for (int64_t i = 0; i < K; ++i) {
    const double xt = fmin_t + (double)i * step_t;
    if (i == 0 && xt < x[0]) return SMILEHIP_ERR_INVALID;
    while (hi < K && x[hi] < xt) hi++;
    if (hi == K) return SMILEHIP_ERR_INVALID;             // the reference's csplint_init fails here too
    const int64_t lo = hi - 1;
    const double range = x[hi] - x[lo];
    if (range == 0.0) return SMILEHIP_ERR_INVALID;
    const double a = (x[hi] - xt) / range, b = 1.0 - a, r2 = range * range / 6.0;
    h.ip_k[i] = (int32_t)lo;
    h.ip_co[3 * i + 0] = a;
    h.ip_co[3 * i + 1] = (a * a * a - a) * r2;
    h.ip_co[3 * i + 2] = (b * b * b - b) * r2;
  }
  const double nOct = std::log(maxF / minF) / l2;
  const double nPPO = (double)K / nOct;
  const double atan_s = nPPO * (std::log(65.0 / 50.0) / l2) - 1.0;
  h.audw.assign(size_t(K), 0.0);
  for (int64_t i = 0; i < K; ++i) h.audw[i] = 0.5 + std::atan(3.0 * ((double)i + 1 - atan_s) / nPPO) / M_PI;
  // the same constants as the thread-per-frame sweep reads them: one record per target point, and how many target
  // points sit above each source bin (the search above only moves `hi` up: ip_k is non-decreasing)
  h.ip_rec.assign(size_t(K) * 4, 0.0);
  h.ip_cnt.assign(size_t((K + 15) / 16 * 16), 0);
  h.sw_rec.assign(size_t(K) * 8, 0.0);
  for (int64_t i = 0; i < K; ++i) {
    h.sw_rec[8 * i + 0] = h.sp_rec[4 * i + 0]; h.sw_rec[8 * i + 1] = h.sp_rec[4 * i + 1]; h.sw_rec[8 * i + 2] = h.sp_rec[4 * i + 2];
    h.sw_rec[8 * i + 3] = h.sp_d1[i]; h.sw_rec[8 * i + 4] = 1.0 / h.sp_d1[i];
    h.sw_rec[8 * i + 5] = h.sp_d2[i]; h.sw_rec[8 * i + 6] = 1.0 / h.sp_d2[i];
  }
  for (int64_t i = 0; i < K; ++i) {
    h.ip_rec[4 * i + 0] = h.ip_co[3 * i + 0];
    h.ip_rec[4 * i + 1] = h.ip_co[3 * i + 1];
    h.ip_rec[4 * i + 2] = h.ip_co[3 * i + 2];
    h.ip_rec[4 * i + 3] = h.audw[i];
    if (i > 0 && h.ip_k[i] < h.ip_k[i - 1]) return SMILEHIP_ERR_INVALID;
    h.ip_cnt[h.ip_k[i]] += 1;
  }
  // what cPitchShs sees: FLOAT_DMEM meta data
  const float m_fmin = (float)minF, m_ppo = (float)nPPO, m_fmint = (float)fmin_t, m_fmaxt = (float)fmax_t;
  double base = std::exp(std::log((double)m_fmin) / (double)m_fmint);
  if (std::fabs(base - 2.0) < 0.00001) base = 2.0;
  h.log_base = std::log(base);
  h.Fmint = m_fmint;
  h.Fstept = (m_fmaxt - m_fmint) / (float)(K - 1);
  h.n_harm = n_harmonics;
  float sc = compression;
  for (int i = 2; i < n_harmonics + 1; ++i) {
    h.shift[i - 2] = (int32_t)std::floor((double)m_ppo * (std::log((double)i) / l2));
    h.scale[i - 2] = sc;
    sc *= compression;
  }
  return SMILEHIP_OK;
}

}  // namespace smilehip
