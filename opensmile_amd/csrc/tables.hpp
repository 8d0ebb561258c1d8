// Host-side table generation for the LLD chain (product code; independent of
// oracle/). Every table is computed with the reference's own precision
// recipe -- double math, one rounding to float where the reference rounds --
// so that the device kernels consume bit-identical constants.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/smilehip.h"

namespace smilehip {

struct Geometry {
  int64_t N = 0, H = 0, Nfft = 0, K = 0;
  double period = 0.0;          // sample period T = 1/fs
  double frame_period = 0.0;    // H*T as the framer's level period (= frameStep)
  double fft_frame_size_sec = 0.0;
};

struct MelBank {
  int n_bands = 0;
  int64_t nLo = 0, nHi = 0;
  std::vector<float> coef;        // one weight per bin (falling slope of band chan[n])
  std::vector<int32_t> chan;      // band index of the falling slope, -3 = unused bin
  std::vector<float> centres;     // n_bands+2 band edges/centres on the mel axis
  // per-band bin ranges derived from chan[] (device-side iteration order):
  // band b sums bins [rise_lo[b], rise_hi[b]) with (p - p*w) then
  // [fall_lo[b], fall_hi[b]) with p*w -- the reference's bin-ascending order.
  std::vector<int32_t> rise_lo, rise_hi, fall_lo, fall_hi;
  float scale = 1.0f;             // HTK sample-value scaling applied after the sums
};

struct DctTables {
  int n_bands = 0, first = 0, last = 0, n_mfcc = 0;
  std::vector<float> cos_rows;    // n_mfcc rows of n_bands, row r = OUTPUT position r
  std::vector<float> gain;        // per output position: lifter * sqrt(2/nB) (one float product)
  std::vector<float> lifter;      // raw lifter table (sintable), reference order
  float melfloor = 0.0f, log_floor = 0.0f;
};

// ComParE / GeMAPS F0 group: cSpecScale (scale=octave, interpMethod=spline, specSmooth=specEnhance=
// auditoryWeighting=1, minF=25, maxF=-1, nPointsTarget=0) and cPitchShs constants for one spectrum geometry
struct F0Host {
  std::vector<double> sp_rec, sp_d1, sp_d2, ip_co, audw;
  std::vector<int32_t> ip_k;
  std::vector<double> ip_rec;     // [K x 4] a, c, d, auditory weight of target point i: one 32-byte record (lld_f0_sweep)
  std::vector<int32_t> ip_cnt;    // [ceil(K / 16) x 16] target points whose lower source bin is k (ip_k is non-decreasing: they are consecutive)
  std::vector<double> sw_rec;     // [K x 8] sigma, p, dec, d1, RN(1 / d1), d2, RN(1 / d2), 0
  int32_t n_harm = 15;
  int32_t shift[16] = {0};
  float scale[16] = {0};
  float Fmint = 0.f, Fstept = 0.f;
  double log_base = 0.0;
};
// K bins of a spectrum level whose frameSizeSec is fft_frame_size_sec; 0 on success
int  make_f0_tables(int64_t K, double fft_frame_size_sec, int n_harmonics, float compression, double min_f, F0Host &h);

int  make_geometry(const smilehip_lld_config &c, Geometry &g);
int  make_window(const smilehip_lld_config &c, int64_t N, std::vector<float> &w);
int  make_mel(const smilehip_lld_config &c, const Geometry &g, MelBank &m);
int  make_dct(const smilehip_lld_config &c, DctTables &d);
float delta_norm(int W);

}  // namespace smilehip
