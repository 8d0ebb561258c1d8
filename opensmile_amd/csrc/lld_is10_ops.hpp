// Per-frame bodies of the components the INTERSPEECH 2010-2012 sets (config/is09-13/IS10_paraling.conf, IS11_speaker_state.conf,
// IS12_speaker_trait.conf) add to the built ones: cIntensity, cLsp, cPitchSmoother, cVectorOperation. Every one of them is a short
// sequential float / double recipe per frame (or per stream) whose rounding order IS the result, so the kernels of
// lld_stage4_kernels.hip run them one thread per frame (per stream) exactly as written here. The functions are
// __host__ __device__: tests/test_is10_ops_host.py compiles this header with g++ and holds it against the real binary's levels
// bit for bit without a GPU; the GPU tests then only have to show that the kernels call them on the right rows.
#pragma once
#include <cmath>
#include <cstdint>

#include "glibc_float.hpp"

namespace smilehip {
namespace is10 {

// ---- cIntensity::processVector (src/lldcore/intensity.cpp:125-145) --------------------------------------------------------
// The reference sums MIN(Nsrc, MIN(nWin, Ndst)) terms, Ndst = the number of OUTPUTS (1 or 2): `n_sum` is that number. win = the
// Hamming window as doubles (smileUtil.c:1291-1303), win_sum its sum in index order (intensity.cpp:99-104) -- host tables.
// flags: 1 intensity, 2 loudness. Returns the number of values written.
GLF_HD int intensity_frame(const float *src, int n_sum, const double *win, double win_sum, int flags, float *dst) {
  double Im = 0.0;
  for (int i = 0; i < n_sum; ++i) Im += win[i] * (double)src[i] * (double)src[i];
  Im /= win_sum;
  int n = 0;
  if (flags & 1) dst[n++] = (float)Im;
  if (flags & 2) dst[n++] = (float)pow(Im / 0.000001, 0.3);     // double pow, rounded to float (libm's and the device's agree after the rounding)
  return n;
}

// ---- cLsp (src/lld/lsp.cpp:112-312): the Speex-derived lpc_to_lsp on FLOAT_DMEM ------------------------------------------
constexpr int kLspMaxOrder = 32;

GLF_HD float cheb_poly_eva(const float *coef, float x, int m) {        // lsp.cpp:112-128
  float b0 = 0.0f, b1 = 0.0f;
  x *= 2.0f;
  for (int k = m; k > 0; --k) {
    const float tmp = b0;
    b0 = x * b0 - b1 + coef[m - k];
    b1 = tmp;
  }
  return (-b1 + 0.5f * x * b0 + coef[m]);
}

GLF_HD int lpc_to_lsp(const float *a, int lpcrdr, float *freq, int nb, float delta) {      // lsp.cpp:144-269
  float P[kLspMaxOrder / 2 + 2], Q[kLspMaxOrder / 2 + 2];
  const int m = lpcrdr / 2;
  for (int i = 0; i <= m; ++i) { P[i] = 0.0f; Q[i] = 0.0f; }
  P[0] = 1.0f;
  Q[0] = 1.0f;
  for (int i = 0; i < m; ++i) {
    P[i + 1] = (a[i] + a[lpcrdr - 1 - i]) - P[i];
    Q[i + 1] = (a[i] - a[lpcrdr - 1 - i]) + Q[i];
  }
  for (int i = 0; i < m; ++i) { P[i] = 2.0f * P[i]; Q[i] = 2.0f * Q[i]; }
  int roots = 0;
  float xr = 0.0f, xl = 1.0f, xm = 0.0f;
  for (int j = 0; j < lpcrdr; ++j) {
    const float *pt = (j & 1) ? Q : P;
    float psuml = cheb_poly_eva(pt, xl, m);
    bool searching = true;
    while (searching && (xr >= -1.0f)) {
      float dd = delta * (1.0f - 0.9f * xl * xl);
      if ((double)__builtin_fabsf(psuml) < .2) dd *= 0.5f;            // fabs(float) < .2: a double comparison
      xr = xl - dd;
      float psumr = cheb_poly_eva(pt, xr, m);
      const float temp_psumr = psumr, temp_xr = xr;
      if ((psumr * psuml) < 0.0f) {
        roots++;
        for (int k = 0; k <= nb; ++k) {
          xm = 0.5f * (xl + xr);
          const float psumm = cheb_poly_eva(pt, xm, m);
          if (!((psumm * psuml) < 0.0f)) { psuml = psumm; xl = xm; }
          else { psumr = psumm; xr = xm; }
        }
        if (xm > 1.0f) xm = 1.0f;
        else if (xm < -1.0f) xm = -1.0f;
        freq[j] = glibc_acosf(xm);                                     // acos(FLOAT_DMEM): the C++ float overload
        xl = xm;
        searching = false;
      } else {
        psuml = temp_psumr;
        xl = temp_xr;
      }
    }
  }
  return roots;
}

// processVector (lsp.cpp:289-312); the output row is zero where no root was written (the writer's fresh vector)
GLF_HD void lsp_frame(const float *lpc, int n_lpc, float *dst) {
  for (int i = 0; i < n_lpc; ++i) dst[i] = 0.0f;
  int roots = lpc_to_lsp(lpc, n_lpc, dst, 10, 0.2f);
  if (roots != n_lpc) {
    roots = lpc_to_lsp(lpc, n_lpc, dst, 10, 0.05f);
    if (roots != n_lpc)
      for (int i = roots; i < n_lpc; ++i) dst[i] = 0.0f;
  }
}

// ---- cPitchSmoother::processVector (src/lldcore/pitchSmoother.cpp:236-425) -------------------------------------------------
// medianFilter0 = 0; postSmoothingMethod none / simple; the outputs F0final (1), F0finEnv (2), voicingFinalClipped (4),
// voicingFinalUnclipped (8) in that order. State of one stream:
struct PitchSmootherState {
  int32_t first_frame, ons_flag, ons_flag_o, reserved;
  float last_voice, last_final, pitch_env, reserved1;
};
struct PitchSmootherOpts {
  int32_t n_cand, octave_correction, post_simple, flags;
  float voicing_cutoff;
};
constexpr int kSmootherMaxCand = 16;

GLF_HD void pitch_smoother_reset(PitchSmootherState &s) {
  s.first_frame = 1; s.ons_flag = 0; s.ons_flag_o = 0; s.reserved = 0;
  s.last_voice = 0.0f; s.last_final = 0.0f; s.pitch_env = 0.0f; s.reserved1 = 0.0f;
}

// src = [F0Cand (n) | candVoicing (n) | candScore (n)] with element stride `st`. Returns the number of values written: 0 for the
// first frame of a stream with simple post smoothing (the component delays its output by one frame).
GLF_HD int pitch_smoother_frame(const PitchSmootherOpts &o, PitchSmootherState &s, const float *src, int64_t st, float *dst) {
  float f0cand[kSmootherMaxCand], candVoice[kSmootherMaxCand];
  const int c = o.n_cand;
  for (int j = 0; j < c; ++j) {
    f0cand[j] = src[(int64_t)j * st];
    candVoice[j] = src[(int64_t)(c + j) * st];
  }
  if (o.octave_correction) {
    bool cand0ismin = true;
    int minC = -1;
    float vpMin = 0.0f;
    for (int i = 1; i < c; ++i) {
      if ((f0cand[i] > 0.0f) && (f0cand[i] < f0cand[0])) {
        if (((double)candVoice[i] > 0.9 * (double)candVoice[0]) && (candVoice[i] > vpMin)) { vpMin = candVoice[i]; minC = i; }
        cand0ismin = false;
      }
    }
    if (!cand0ismin) {
      if (minC >= 0) {
        float t = f0cand[0]; f0cand[0] = f0cand[minC]; f0cand[minC] = t;
        t = candVoice[0]; candVoice[0] = candVoice[minC]; candVoice[minC] = t;
      }
    } else {
      bool halfed = false;
      int j = 0;
      while ((!halfed) && j < c - 1) {
        for (int i = j + 1; i < c; ++i) {
          if ((f0cand[i] > 0.0f) && (f0cand[j] > 0.0f)) {
            float k = __builtin_fabsf(f0cand[i] - f0cand[j]) * 2.0f / f0cand[0];     // fabs(FLOAT_DMEM): the float overload
            k = (float)__builtin_fabs((double)k - 1.0);
            if ((double)k < 0.1) { f0cand[0] /= 2.0f; halfed = true; break; }
          }
        }
        j++;
      }
    }
  }
  int n = 0;
  float voiceC1 = candVoice[0];
  if (o.flags & 3) {
    const float pitch = (candVoice[0] > o.voicing_cutoff) ? f0cand[0] : 0.0f;
    float pitchOut;
    if (o.post_simple) {
      if (s.first_frame) { s.first_frame = 0; return 0; }
      voiceC1 = s.last_voice;
      s.last_voice = candVoice[0];
      if ((s.last_final == 0.0f) && (pitch > 0.0f)) s.ons_flag = 1;
      if ((s.last_final > 0.0f) && (pitch == 0.0f) && (s.ons_flag == 0)) s.ons_flag = -1;
      if ((s.last_final > 0.0f) && (pitch > 0.0f)) s.ons_flag = 0;
      if ((s.last_final == 0.0f) && (pitch == 0.0f)) s.ons_flag = 0;
      if ((pitch == 0.0f) && (s.ons_flag == 1)) s.last_final = 0.0f;
      else if ((pitch > 0.0f) && (s.ons_flag == -1)) s.last_final = pitch;
      bool doubling = false, halfing = false;
      if ((s.last_final > 0.0f) && (pitch > 0.0f)) {
        const float factor = s.last_final / pitch;
        if ((double)factor > 1.2) halfing = true;
        else if ((double)factor < 0.8) doubling = true;
      }
      if (doubling && (s.ons_flag_o == -1)) s.last_final = pitch;
      else if (halfing && (s.ons_flag_o == 1)) s.last_final = pitch;
      if (doubling) s.ons_flag_o = 1;
      if (halfing && (s.ons_flag == 0)) s.ons_flag_o = -1;
      if (!(halfing || doubling)) s.ons_flag_o = 0;
      pitchOut = s.last_final;
      s.last_final = pitch;
    } else {
      pitchOut = pitch;
    }
    if (o.flags & 1) dst[n++] = pitchOut;
    if (o.flags & 2) {
      if (pitchOut > 0.0f) {
        if (s.pitch_env == 0.0f) s.pitch_env = pitchOut;
        else s.pitch_env = 0.75f * s.pitch_env + 0.25f * pitchOut;
      }
      dst[n++] = s.pitch_env;
    }
  }
  if (o.flags & 4) dst[n++] = (voiceC1 > o.voicing_cutoff) ? voiceC1 : 0.0f;
  if (o.flags & 8) dst[n++] = voiceC1;
  return n;
}

// ---- cVectorOperation::processVector, the element-wise operations (src/other/vectorOperation.cpp:360-435, 508-527) ---------
// std::log / std::exp / std::sqrt on FLOAT_DMEM are the float functions of the C library. `aux` is the operation's constant:
// param1 (add, mul), logf(param1) (lgA), (float)(10 / log(10)) (dBp), (float)(20 / log(10)) (dBv) -- computed by the host.
enum { kVopAdd = 0, kVopMul, kVopLog, kVopLogA, kVopSqrt, kVopE, kVopAbs, kVopDbPow, kVopDbMag, kVopCount,
       kVopXSum = kVopCount, kVopXSumSq, kVopXL1, kVopXL2, kVopXCount };

GLF_HD float vecop(int op, float aux, float logfloor, float x) {
  switch (op) {
    case kVopAdd: return x + aux;
    case kVopMul: return x * aux;
    case kVopLog: return x > logfloor ? glibc_logf(x) : glibc_logf(logfloor);
    case kVopLogA: return (x > logfloor ? glibc_logf(x) : glibc_logf(logfloor)) / aux;
    case kVopSqrt: return x > 0.0f ? __builtin_sqrtf(x) : 0.0f;
    case kVopE: return glibc_expf(x);
    case kVopAbs: return __builtin_fabsf(x);
    case kVopDbPow:
    case kVopDbMag: return aux * (x > logfloor ? glibc_logf(x) : glibc_logf(logfloor));
  }
  return x;
}

// The vector-to-scalar operations (vectorOperation.cpp:461-490): sum, ssm, ll1, ll2 -- one FLOAT_DMEM accumulation in index order
// (the reference accumulates in dst[0]); sqrt on FLOAT_DMEM is sqrtf.
GLF_HD float vecop_reduce(int op, const float *src, int64_t n) {
  float acc = 0.0f;
  if (op == kVopXSum || op == kVopXL1) {
    for (int64_t i = 0; i < n; ++i) acc += src[i];
  } else {
    for (int64_t i = 0; i < n; ++i) acc += src[i] * src[i];
  }
  if (op == kVopXL2 && acc > 0.0f) acc = __builtin_sqrtf(acc);
  if ((op == kVopXL1 || op == kVopXL2) && n > 0) acc /= (float)n;
  return acc;
}

}  // namespace is10
}  // namespace smilehip
