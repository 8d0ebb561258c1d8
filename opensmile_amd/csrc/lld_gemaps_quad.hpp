// eGeMAPS / GeMAPS 20 ms frame kernel (lld_gemaps.hip: lld_gemaps_frame20) with SIXTEEN LANES per frame, four runs per wave -- the
// layout of lld_compare_quad.hpp (see there: a row of 16 lanes owns a run of consecutive frames, bin k = j + 16 m in register m of
// lane j, the previous frame's magnitudes stay in the row's registers for the flux). Per frame: energy2 of the raw frame, the
// reference-order transform, the complex bins cSpecResample reads (bins 0 .. 109 -> spec220), one mel bank in two scalings ->
// auditory spectrum sum + MFCC 1 .. 4, cSpectral's GeMAPS sets (two slopes of the log spectrum, alpha ratio, Hammarberg index,
// flux over 0 .. 5 kHz). Float chains in the reference's order, double sums in the row's association (lld_blocks.hpp: QuadG).
// The shipped geometry only: 16 kHz, N = 320, hop 160, FFT 512 (96 zeros in front of the frame, or none), 26 bands, 4 cepstra, int16 input.
#pragma once
#include <hip/hip_runtime.h>

#include "lld_compare_quad.hpp"

namespace smilehip {

namespace gq {
constexpr int kN = 320, kH = 160, kM = 256, kK = 257, kBands = 26, kMfcc = 4, kRsB = 109, kRsI = 220;
constexpr int kRowFloats = 2 * kQuadZPairs + 64 + 64;      // z | lmel[32] | aud[32] | lg[64]
constexpr int kTableFloats = kN + 260 + 128 + 16 * 32;     // window[320] | mel coef[260] | band ranges[128] | DCT rows
}  // namespace gq

template <int PAD>                                         // zeros in front of the frame: 96 (zeroPadSymmetric = 1: eGeMAPSv02 / GeMAPSv01b) or 0 (the v01a files)
__device__ __forceinline__ void gemaps_frame20_quad_body(const LldParams &P, const GemapsParams &G, int n_runs, int first_run, const float *s_win,
                                                         const float *s_coef, const int32_t *s_rng, const float *s_dct, const OouraTab &OO,
                                                         float *fmem) {
  using namespace gq;
  const int run_raw = first_run + (int)((threadIdx.x & 63) >> 4);
  const bool have_run = run_raw < n_runs;
  const int run = have_run ? run_raw : n_runs - 1;
  const int u = G.run_utt[run], t0 = G.run_t0[run];
  const int f0 = (int)P.frame_off[u];
  const int T20 = (int)(P.frame_off[u + 1] - P.frame_off[u]);
  const int16_t *xu = P.pcm + P.samp_off[u];
  const int run_len = G.run_frames > 0 ? G.run_frames : 8;
  const int t_last = (t0 + run_len < T20) ? t0 + run_len : T20;
  const int t_begin = t0 > 0 ? t0 - 1 : 0;
  const int n_pass = __builtin_amdgcn_readfirstlane(wave_tree_i(t_last - t_begin, [](int a, int b) { return b > a ? b : a; }));
  const double F0 = 1.0 / G.fsSec;
  float mvp[17];
#pragma unroll
  for (int m = 0; m < 17; ++m) mvp[m] = 0.0f;

  for (int it = 0; it < n_pass; ++it) {
    int tid = threadIdx.x;                                 // opaque per pass (lld_compare_quad.hpp)
    asm volatile("" : "+v"(tid));
    const int lane64 = tid & 63, j = tid & 15, g = lane64 >> 4, row_base4 = 4 * (lane64 & 48);
    float *rowm = fmem + g * kRowFloats;
    float2 *z = reinterpret_cast<float2 *>(rowm);
    float *zf = rowm;
    float *lmel = rowm + 2 * kQuadZPairs, *aud = lmel + 32, *lg = aud + 32;
    const int t_raw = t_begin + it;
    const bool live = have_run && t_raw < t_last;
    const int t = t_raw < t_last ? t_raw : t_last - 1;
    const bool warm = t < t0;
    const bool store = live && !warm;
    float *raw = G.raw20 + (int64_t)(f0 + t) * 12;
    const int16_t *x = xu + (int64_t)t * kH;      // (64-bit: t * kH in int overflows on an utterance of more than 2^31 samples)
    // ---- samples: element i = 16 r + j holds samples 2 i - PAD, 2 i - PAD + 1 (ten registers of the sixteen)
    static_assert(PAD % 32 == 0 && PAD >= 0 && PAD <= 192, "whole registers of padding");
    float2 v[16];
    double e2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r >= PAD / 32 && r < PAD / 32 + 10) {
        const int n0 = 32 * (r - PAD / 32) + 2 * j;
        const float x0 = pcm16_to_float(x[n0]), x1 = pcm16_to_float(x[n0 + 1]);
        { const float q0 = x0 * x0; e2 += (double)q0; const float q1 = x1 * x1; e2 += (double)q1; }      // cEnergy energy2 of the raw frame (energy.cpp:152-170)
        const float2 w = *reinterpret_cast<const float2 *>(s_win + n0);
        v[r] = make_float2(x0 * w.x + P.win_offset, x1 * w.y + P.win_offset);
      } else {
        v[r] = make_float2(0.0f, 0.0f);
      }
    }
    e2 = QuadG::sum(e2, nullptr);
    // ---- transform, complex bins for cSpecResample, magnitudes
    oo_quad256<false>(v, OO, z, lane64);
    oo_quad_store(v, z, lane64);
    float mv[17];
    float edge0 = 0.0f;
    float *spec = G.spec220 + (int64_t)(f0 + t) * kRsI;
#pragma unroll
    for (int m = 0; m < 17; ++m) {
      const int k = j + 16 * m;
      const float2 X = (k <= kM) ? oo_wave_bin<256>(z, OO, k <= kM ? k : 0) : make_float2(0.0f, 0.0f);
      mv[m] = (m < 16) ? X.x * X.x + X.y * X.y : ((k <= kM) ? fabsf(X.x) : 0.0f);     // (the roots: sqrt_rn_batch below)
      if (m == 0) edge0 = fabsf(X.x);
      // what cSpecResample reads of the complex level (Ooura packing, fftsg.c:103-135): (Re, -Im) of bins 1 .. 109, then a[0], one pad
      if (m < 7 && store) {
        if (k == 0) *reinterpret_cast<float2 *>(spec + 2 * kRsB) = make_float2(X.x, 0.0f);
        else if (k <= kRsB) *reinterpret_cast<float2 *>(spec + 2 * k - 2) = make_float2(X.x, -X.y);
      }
      if (m % 6 == 5) __builtin_amdgcn_sched_barrier(0);
    }
    if (j == 0) mv[0] = 1.0f;                              // (bin 0 and bin M take |re|; k > M is lanes j > 0 at m = 16 only)
    sqrt_rn_batch(reinterpret_cast<float (&)[16]>(mv));
    if (j == 0) mv[0] = edge0;
    QuadG::sync();
    // flux (:1124-1254) over freqRange's bins while the previous frame's magnitudes are here
    double fl = 0.0;
#pragma unroll
    for (int m = 0; m < 17; ++m) {
      const int k = j + 16 * m;
      const double d = (double)mv[m] - (double)mvp[m];
      if (k >= G.rng_lo && k <= G.rng_hi && k <= kM) fl += d * d;
      mvp[m] = mv[m];
    }
    if (!warm) {
      // ---- R6 once, two scalings: [gemapsv01b_melspec1] feeds cPlp, [egemapsv02_melspecMfcc] feeds cMfcc
      {
        float *mt_a = zf, *mt_r = zf + kK;
#pragma unroll
        for (int m = 0; m < 17; ++m) {
          const int k = j + 16 * m;
          if (k <= kM) {
            const float pk = mv[m] * mv[m], ak = pk * s_coef[k];
            mt_a[k] = ak;
            mt_r[k] = pk - ak;
          }
        }
      }
      QuadG::sync();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int b = h == 0 ? j : kBands - 1 - j;
        if (j < kBands / 2) {
          const float acc = mel_band_from_terms(zf, zf + kK, s_rng, b, 1.0f);
          lmel[b] = log_mel(acc * P.mel_scale, P.melfloor, P.log_floor);
          aud[b] = plp_aud_band(acc, G.plp_melfloor, G.eql[b], G.compression);     // [gemapsv01b_audspec], plp.cpp:499-507
        }
      }
      QuadG::sync();
      if (j < kMfcc) { const float c = dct_coeff(lmel, s_dct + j * kBands, kBands, P.dct_gain[j]); if (store) raw[6 + j] = c; }     // R7
      if (j == 15) { const float d = seq_sum_f32(aud, kBands); if (store) raw[0] = d / (float)kBands; }     // [gemapsv01b_audspecSum] ll1
      if (store && j == 14) { raw[10] = (float)(e2 / (double)kN) * 1.0f + 0.0f; raw[11] = 0.0f; }
      QuadG::sync();                                       // (the terms have been read: the powers take their place)
      // ---- cSpectral, GeMAPS sets (lld_gemaps.hip: gemaps_spectral_wave)
      const bool first = t == 0;
#pragma unroll
      for (int m = 0; m < 17; ++m) { const int k = j + 16 * m; if (k <= kM) zf[k] = mv[m] * mv[m]; }
      // log power spectrum of bins 0 .. 63 (spectral.cpp:689-716)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float p = mv[m] * mv[m];
        lg[j + 16 * m] = (p <= G.spec_floor) ? G.log_spec_floor : G.log_spec_factor * glibc_logf(p);
      }
      QuadG::sync();
      float slope[2] = {0.0f, 0.0f};
#pragma unroll
      for (int b = 0; b < 2; ++b) {                        // band slopes of the log spectrum (spectral.cpp:872-992)
        const int iL = G.sl_iL[b], iR = G.sl_iR[b];
        const double wL = G.sl_wL[b], wR = G.sl_wR[b];
        double s[4] = {0.0, 0.0, 0.0, 0.0};              // Sf, S2f, sumA, sumB
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int k = j + 16 * m;
          if (k >= iL && k <= iR) {
            const double f = F0 * (double)k, l = (double)lg[k];
            if (k == iL) { const double fw = f * wL; s[0] += fw; s[1] += fw * fw; s[2] += fw * l; s[3] += wL * l; }
            else if (k == iR) { const double fw = f * wR; s[0] += fw; s[1] += fw * fw; s[2] += fw * l; s[3] += wR * l; }
            else { s[0] += f; s[1] += f * f; s[2] += f * l; s[3] += l; }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = QuadG::sum(s[q], nullptr);
        const double Nind = G.sl_Nind[b];
        const double deno = (Nind * s[1] - s[0] * s[0]);
        double sl = 0.0;
        if (deno != 0.0) sl = (Nind * s[2] - s[0] * s[3]) / deno;
        slope[b] = (float)sl;                              // oldSlopeScale = 0
      }
      // alpha ratio (:995-1037) and Hammarberg index (:1039-1089) over the bins up to 5000 Hz
      float m02 = 0.0f, m25 = 0.0f;
#pragma unroll
      for (int m = 0; m < 17; ++m) {
        const int k = j + 16 * m;
        const double f = F0 * (double)k;
        if (k <= kM && !(f > 5000.0)) {
          const float p = mv[m] * mv[m];
          if (f < 2000.0) m02 = p > m02 ? p : m02; else m25 = p > m25 ? p : m25;
        }
      }
      {
        float w;
        w = __int_as_float(row_ror_i<8>(__float_as_int(m02))); m02 = w > m02 ? w : m02; w = __int_as_float(row_ror_i<4>(__float_as_int(m02))); m02 = w > m02 ? w : m02;
        w = __int_as_float(row_ror_i<2>(__float_as_int(m02))); m02 = w > m02 ? w : m02; w = __int_as_float(row_ror_i<1>(__float_as_int(m02))); m02 = w > m02 ? w : m02;
        w = __int_as_float(row_ror_i<8>(__float_as_int(m25))); m25 = w > m25 ? w : m25; w = __int_as_float(row_ror_i<4>(__float_as_int(m25))); m25 = w > m25 ? w : m25;
        w = __int_as_float(row_ror_i<2>(__float_as_int(m25))); m25 = w > m25 ? w : m25; w = __int_as_float(row_ror_i<1>(__float_as_int(m25))); m25 = w > m25 ? w : m25;
      }
      // sum01 / sum15 are FLOAT_DMEM accumulators (:997-1022): lanes 0 and 1 add the powers of their band bin after bin
      float chain = 0.0f;
      if (j < 2) chain = seq_sum_f32(zf, j == 0 ? 0 : G.ar_n1, j == 0 ? G.ar_n1 : G.ar_n2);
      const float sum01 = cq::lane_f(chain, row_base4, 0), sum15 = cq::lane_f(chain, row_base4, 1);
      const double flux_sum = QuadG::sum(first ? 0.0 : fl, nullptr);
      if (store && j == 0) {
        float a = 0.0f, h = 0.0f;
        if (sum01 > 0.0f) {
          if (sum15 > G.spec_floor) a = (float)(10.0 * (double)glibc_logf(sum15 / sum01) / log(10.0));
          else a = (float)(10.0 * (double)(glibc_logf(G.spec_floor) - glibc_logf(sum01)) / log(10.0));
        }
        if (m25 > 0.0f) {
          if (m02 > G.spec_floor) h = (float)(10.0 * (double)glibc_logf(m02 / m25) / log(10.0));
          else h = (float)(10.0 * (double)(glibc_logf(G.spec_floor) - glibc_logf(m25)) / log(10.0));
        }
        raw[1] = slope[0];
        raw[2] = slope[1];
        raw[3] = a;
        raw[4] = h;
        const int nBins = G.rng_hi - G.rng_lo + 1;
        const double flux = (nBins > 0) ? flux_sum / (double)nBins : 0.0;
        raw[5] = (!first && flux > 0.0) ? (float)sqrt(flux) : 0.0f;      // first frame of a stream: 0 (:1132-1136)
      }
      QuadG::sync();
    }
  }
}

}  // namespace smilehip
