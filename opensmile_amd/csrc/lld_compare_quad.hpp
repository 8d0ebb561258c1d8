// ComParE_2016 groups A + B (lld_compare.hip) with SIXTEEN LANES per 20 ms frame, four frames per wave -- the layout that took
// the IS09 frame kernel from 3 455 to ~1 450 vector instructions per frame (lld_is09.hip, lld_ooura_quad.hpp), for the kernel
// whose narrow phases weigh most: 26 mel bands / 14 cepstra / one lane's float chains kept a whole wave busy per frame, the
// transform walked every butterfly variant with most lanes masked off, every reduction was six steps deep.
//
// A DPP row of 16 lanes owns ONE RUN of consecutive frames (CompareParams::run_*), a wave four runs, side by side and
// independent of each other: what a frame needs of its predecessor -- the magnitudes (spectral flux), the zero-crossing counts of
// the 60 ms window's hops -- stays in the row's registers from pass to pass; nothing crosses a row. Bin k = j + 16 m lives in
// register m of lane j (m <= 16; bin 256 in lane 0).
//   samples     the frame's 320 samples straight into the transform's registers (lane j: pairs 2 (16 r + j) - 96), the window
//               applied on the way; RMS energy from the same registers
//   ZCR         the 60 ms window's crossings as six hop-aligned segment counts kept in a register FIFO: a pass counts the 160
//               positions that entered (11 samples per lane, neighbours by row rotation); integers: any order is the reference's sum
//   transform   oo_quad256 (the reference's rdft network, 16 points per lane), magnitudes in registers
//   mel / PLP   terms to LDS, a lane takes band j and band 25 - j (narrow + wide), DCT on 14 lanes, the auditory sum on lane 15
//   descriptors sums over bins: per lane, then four row rotations; roll-off: a row scan per register + running offset; harmonicity's
//               "previous flagged bin" from row ballots; the two FLOAT_DMEM chains by lanes 0 and 1 of the row, one chain each
// Sums of doubles are associated otherwise than in the wave form (as that one's are otherwise than the block form's and the
// reference's bin-after-bin loops): every one of them is rounded to float or compared with a threshold afterwards, and differs
// from the sequential sum only where the double's last bits decide that rounding (lld_blocks_compare.hpp; ~1e-9 per value --
// the bench's accuracy object counts the cells). Float chains keep the reference's order operation for operation.
// The shipped geometry only (16 kHz: N = 320, hop 160, FFT 512, 96 zeros in front, 60 ms = 960 samples, 26 bands, 14 cepstra,
// int16 input, reference-order tables): launch_compare checks it.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "lld_blocks.hpp"
#include "lld_blocks_compare.hpp"
#include "lld_ooura_quad.hpp"

// (development aid: tools/dev/phase_insts.sh lld_compare.hip <kernel> QPHASE compiles these as assembler comments and counts the
// instructions between them)
#ifndef QPHASE
#define QPHASE(i)
#endif
#ifndef CQ_BINS_GROUP
#define CQ_BINS_GROUP 3                                    // bins of the entropy / moments loop whose operations may interleave
#endif
#ifndef CQ_WAVES
#define CQ_WAVES 3                                         // waves per SIMD the register budget is set for
#endif

namespace smilehip {

namespace cq {
constexpr int kN = 320, kH = 160, kPad = 96, kN60 = 960, kM = 256, kK = 257, kBands = 26, kMfcc = 14;
constexpr int kRowFloats = 2 * kQuadZPairs + 64 + 16;      // z (544 floats: transform, then mel terms / powers / chains) | lmel[32] | aud[32] | the row's state[16]
// shared tables: log table (128 double2 = 512 floats, first: 16-byte aligned) | sharpness weights (256 doubles = 512 floats) |
// window[320] | mel coef[260] | band ranges[128] | DCT rows[16 x 32]
constexpr int kTableFloats = 512 + 512 + kN + 260 + 128 + 16 * 32;

__device__ __forceinline__ int ror_i(int v, int n) {       // (compile-time n at every call site)
  switch (n) {
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, true);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, true);
    case 14: return __builtin_amdgcn_update_dpp(0, v, 0x12e, 0xf, 0xf, true);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x12f, 0xf, 0xf, true);
  }
}
__device__ __forceinline__ float rol1(float x) { return __int_as_float(ror_i(__float_as_int(x), 15)); }   // lane j <- j + 1
__device__ __forceinline__ float rol2(float x) { return __int_as_float(ror_i(__float_as_int(x), 14)); }   // lane j <- j + 2
template <int N>
__device__ __forceinline__ double shr0_d(double x) {       // lane j <- lane j - N of its row, zero below the row's first lane
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x110 + N, 0xf, 0xf, true);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x110 + N, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_f(float x, int row_base4, int src) {      // lane `src` of the row (row_base4 = 4 x the row's first lane)
  return __int_as_float(__builtin_amdgcn_ds_bpermute(row_base4 + 4 * src, __float_as_int(x)));
}
__device__ __forceinline__ double lane_d(double x, int row_base4, int src) {
  const int hi = __builtin_amdgcn_ds_bpermute(row_base4 + 4 * src, __double2hiint(x));
  const int lo = __builtin_amdgcn_ds_bpermute(row_base4 + 4 * src, __double2loint(x));
  return __hiloint2double(hi, lo);
}
// RN(a / b) from y = RN(1 / b) (the division itself, once per divisor): q0 = RN(a y) is within two ulps, one residual step makes
// it faithful, the second returns the correctly rounded quotient (Markstein; lld_f0.hip: f0_div_by, tests/test_exact_sum_claims.py)
// -- five full-rate operations instead of the ~25 of the division sequence, for operands well inside the normal range
__device__ __forceinline__ double div_by(double a, double b, double y) {
  const double q0 = a * y;
  const double r0 = __builtin_fma(-q0, b, a);
  const double q1 = __builtin_fma(r0, y, q0);
  const double r1 = __builtin_fma(-q1, b, a);
  return __builtin_fma(r1, y, q1);
}
// x, behind a wall the common-subexpression pass does not see through: the bins' powers as doubles are used by three loops a long
// way apart, and the compiler would rather keep all seventeen (34 registers, spilled) than square and convert again
__device__ __forceinline__ float fresh(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double fresh(double x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int fresh(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int cross(float a, float b, float c) {               // mzcr.cpp:117-124
  return (((a * c <= 0.0f) && (b == 0.0f)) || (a * b < 0.0f)) ? 1 : 0;
}
// The 160 positions q = base + 1 + k, k = 0 .. 159 (utterance-relative sample indices; cross(q) reads samples q - 1, q, q + 1):
// `two` = the crossings at k = 0, 1, `rest` = those at k = 2 .. 159, summed over the row (every lane returns the totals). Samples
// outside [0, len) read as 0 (the caller does not use the counts they touch).
__device__ __forceinline__ void hop_counts(const int16_t *xu, int base, int len, int j, int &two, int &rest) {
  float b[11];
#pragma unroll
  for (int m = 0; m < 11; ++m) {
    const int i = base + j + 16 * m;
    const bool in = i >= 0 && i < len;                     // (branch-free: a load behind the condition waits alone, eleven times per hop)
    const float v = pcm16_to_float(xu[in ? i : 0]);
    b[m] = in ? v : 0.0f;
  }
  int c2 = 0, cr = 0;
#pragma unroll
  for (int m = 0; m < 10; ++m) {                           // position k = j + 16 m: samples b(k), b(k + 1), b(k + 2), b(i) = sample base + i
    const float n1a = rol1(b[m]), n1b = rol1(b[m + 1]), n2a = rol2(b[m]), n2b = rol2(b[m + 1]);
    const float s1 = (j < 15) ? n1a : n1b, s2 = (j < 14) ? n2a : n2b;
    const int c = cross(b[m], s1, s2);
    if (m == 0 && j < 2) c2 += c; else cr += c;
  }
  two = QuadG::sum_i(c2, nullptr);
  rest = QuadG::sum_i(cr, nullptr);
}
}  // namespace cq

// One wave: four runs (one per row of 16 lanes). smem: the workgroup's tables (staged by the kernel), fmem: this wave's 4 x kRowFloats.
__device__ __forceinline__ void compare_frame_quad_body(const LldParams &P, const CompareParams &Q, int n_runs, int first_run, const float *s_win,
                                                        const float *s_coef, const int32_t *s_rng, const float *s_dct, const double2 *s_log,
                                                        const double *s_sharp, const OouraTab &OO, float *fmem) {
  using namespace cq;
  const int run_raw = first_run + (int)((threadIdx.x & 63) >> 4);
  const bool have_run = run_raw < n_runs;
  const int run = have_run ? run_raw : n_runs - 1;        // (a row without a run repeats the last one, stores off)
  const int u = Q.run_utt[run], t0 = Q.run_t0[run];
  // (32-bit row state: launch_compare admits batches of < 2^31 frames and utterances of < 2^31 samples)
  const int f0 = (int)P.frame_off[u];
  const int T20 = (int)(P.frame_off[u + 1] - P.frame_off[u]);
  const int16_t *xu = P.pcm + P.samp_off[u];
  const int utt_len = (int)(P.samp_off[u + 1] - P.samp_off[u]);
  const int T60 = (utt_len >= kN60) ? (utt_len - kN60) / kH + 1 : 0;
  const int run_len = Q.run_frames > 0 ? Q.run_frames : 8;
  const int t_last = (t0 + run_len < T20) ? t0 + run_len : T20;
  const int t_begin = t0 > 0 ? t0 - 1 : 0;
  const int n_pass = __builtin_amdgcn_readfirstlane(wave_tree_i(t_last - t_begin, [](int a, int b) { return b > a ? b : a; }));
  const double F0 = 1.0 / Q.fsSec;
  // The row's own bookkeeping (its run's frame range, the utterance's samples, the zero-crossing segments) lives in the row's LDS
  // area between the passes: twelve registers less across the transform (they were spilled)
  {
    int *st = reinterpret_cast<int *>(fmem + ((threadIdx.x & 63) >> 4) * kRowFloats + 2 * kQuadZPairs + 64);
    if ((threadIdx.x & 15) == 0) {
      st[0] = f0; st[1] = utt_len; st[2] = T60; st[3] = t0; st[4] = t_last; st[5] = t_begin;
      st[6] = (int)(unsigned)(reinterpret_cast<uintptr_t>(xu) & 0xffffffffu); st[7] = (int)(unsigned)(reinterpret_cast<uintptr_t>(xu) >> 32);
      st[8] = 0; st[9] = 0; st[10] = 0; st[11] = 0; st[12] = 0; st[13] = 0; st[14] = have_run ? 1 : 0;
    }
    QuadG::sync();
  }
  float mvp[17];                                           // the previous frame's magnitudes (flux)
#pragma unroll
  for (int m = 0; m < 17; ++m) mvp[m] = 0.0f;

  for (int it = 0; it < n_pass; ++it) {
    // The thread index is made opaque PER PASS: what depends on the lane alone (LDS addresses of the transform's transposition and
    // read-out, the masks of the unrolled loops) is loop-invariant over the passes, and the compiler would carry all of it across
    // the loop (~40 registers, spilled: lld_is09.hip's is09_quad_body)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane64 = tid & 63, j = tid & 15, g = lane64 >> 4, row_base4 = 4 * (lane64 & 48);
    float *rowm = fmem + g * kRowFloats;
    float2 *z = reinterpret_cast<float2 *>(rowm);
    float *zf = rowm;
    float *lmel = rowm + 2 * kQuadZPairs, *aud = lmel + 32;
    int *st = reinterpret_cast<int *>(aud + 32);
    const int f0 = st[0], utt_len = st[1], T60 = st[2], t0 = st[3], t_last = st[4], t_begin = st[5];
    const int16_t *xu = reinterpret_cast<const int16_t *>((uintptr_t)(unsigned)st[6] | ((uintptr_t)(unsigned)st[7] << 32));
    const bool have_run = st[14] != 0;
    const int t_raw = t_begin + it;
    const bool live = have_run && t_raw < t_last;
    const int t = t_raw < t_last ? t_raw : t_last - 1;
    const bool warm = t < t0;
    const bool store = live && !warm;
    float *rawA = Q.rawA + (int64_t)(f0 + t) * 4;
    float *rawB = Q.rawB + (int64_t)(f0 + t) * 55;
    const int16_t *x = xu + t * kH;
    // ---- the frame's samples, in the transform's layout: element i = 16 r + j holds samples 2 i - 96, 2 i - 95 (r = 3 .. 12)
    float2 v[16];
    double e2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r >= 3 && r <= 12) {
        const int n0 = 32 * (r - 3) + 2 * j;
        const float x0 = pcm16_to_float(x[n0]), x1 = pcm16_to_float(x[n0 + 1]);
        { const float q0 = x0 * x0; e2 += (double)q0; const float q1 = x1 * x1; e2 += (double)q1; }      // R12 cEnergy on the RAW frame (energy.cpp:152-168)
        const float2 w = *reinterpret_cast<const float2 *>(s_win + n0);
        v[r] = make_float2(x0 * w.x + P.win_offset, x1 * w.y + P.win_offset);                            // R3 (no pre-emphasis in this chain)
      } else {
        v[r] = make_float2(0.0f, 0.0f);
      }
    }
    {
      const double tot = QuadG::sum(e2, nullptr);
      if (store && j == 0) rawA[2] = (float)sqrt(tot / (float)kN) * 1.0f + 0.0f;
    }
    QPHASE(0);
    // ---- R12 cMZcr on the 60 ms window (mzcr.cpp:117-124): positions t H + 1 .. t H + 958 = segments S_t .. S_{t+4} (S_m = the
    // positions m H + 1 .. m H + H) + the first 158 positions of S_{t+5}
    if (!warm) {
      const int tH = t * kH;
      int zS0 = st[8], zS1 = st[9], zS2 = st[10], zS3 = st[11], zP = st[12];     // S_t .. S_{t+3} of this frame, the open segment's partial count
      const bool z_have = st[13] != 0;
      if (!z_have) {                                       // the run's first frame: the whole window, hop by hop
        int S4 = 0, prev_rest = 0, rest = 0;
#pragma unroll 1
        for (int h = 0; h < 6; ++h) {
          int two;
          hop_counts(xu, tH + h * kH - 2, utt_len, j, two, rest);
          if (h > 0) { zS0 = zS1; zS1 = zS2; zS2 = zS3; zS3 = S4; S4 = prev_rest + two; }      // S_{t+h-1} is complete
          prev_rest = rest;
        }
        if (store && j == 0 && t < T60) rawA[3] = (float)(double)(zS0 + zS1 + zS2 + zS3 + S4 + rest) / (float)kN60;
        zS0 = zS1; zS1 = zS2; zS2 = zS3; zS3 = S4; zP = rest;
      } else {
        int two, rest;
        hop_counts(xu, tH + 5 * kH - 2, utt_len, j, two, rest);
        const int S4 = zP + two;
        if (store && j == 0 && t < T60) rawA[3] = (float)(double)(zS0 + zS1 + zS2 + zS3 + S4 + rest) / (float)kN60;
        zS0 = zS1; zS1 = zS2; zS2 = zS3; zS3 = S4; zP = rest;
      }
      if (j == 0) { st[8] = zS0; st[9] = zS1; st[10] = zS2; st[11] = zS3; st[12] = zP; st[13] = 1; }
    }
    QPHASE(1);
    // ---- R4 forward transform in the reference's operation order, R5 magnitudes (registers)
    oo_quad256<false>(v, OO, z, lane64);
    oo_quad_store(v, z, lane64);
    // (bins j + 16 m, m < 16: sqrt(re^2 + im^2) through sqrt_rn_batch (lld_device.hpp), bin 0 (j = 0, m = 0) and bin M (j = 0, m = 16): |re|)
    float mv[17];
    float edge0 = 0.0f;
#pragma unroll
    for (int m = 0; m < 17; ++m) {
      const int k = j + 16 * m;
      const float2 X = oo_wave_bin<256>(z, OO, k <= kM ? k : 0);
      mv[m] = (m < 16) ? X.x * X.x + X.y * X.y : ((k <= kM) ? fabsf(X.x) : 0.0f);
      if (m == 0) edge0 = fabsf(X.x);
      if (m % 6 == 5) __builtin_amdgcn_sched_barrier(0);
    }
    if (j == 0) mv[0] = 1.0f;
    sqrt_rn_batch(reinterpret_cast<float (&)[16]>(mv));
    if (j == 0) mv[0] = edge0;
    QuadG::sync();                                         // (z has been read)
    QPHASE(2);
    // spectral flux's sum (:1124-1254) while the previous frame's magnitudes are still here; then this frame's take their place
    double s3 = 0.0;
#pragma unroll
    for (int m = 0; m < 17; ++m) {
      const int k = j + 16 * m;
      const double d = (double)mv[m] - (double)mvp[m];
      if (k >= 1 && k <= kM) s3 += d * d;
      mvp[m] = mv[m];
    }
    if (!warm) {
      // ---- R6 mel terms (power spectrum x one bank), R8 auditory spectrum, R7 MFCC 1 .. 14
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
          if (store) Q.mel1[(int64_t)(f0 + t) * 26 + b] = glibc_logf(acc < Q.plp_melfloor ? Q.plp_melfloor : acc);     // plp.cpp:434-439: logf
          lmel[b] = log_mel(acc * P.mel_scale, P.melfloor, P.log_floor);
          aud[b] = plp_aud_band(acc, Q.plp_melfloor, Q.eql[b], Q.compression);
        }
      }
      QuadG::sync();
      if (j < kMfcc) { const float c = dct_coeff(lmel, s_dct + j * kBands, kBands, P.dct_gain[j]); if (store) rawB[41 + j] = c; }   // R7
      if (j == 15) { const float d = seq_sum_f32(aud, kBands); if (store) rawA[0] = d / (float)kBands; }     // cVectorOperation ll1, vectorOperation.cpp:475-481
      QuadG::sync();                                       // (the terms have been read: the powers take their place)
      QPHASE(3);
      // ---- R11 cSpectral (spectral.cpp:586-1560, ComParE's option set)
      const bool first = t == 0;
#pragma unroll
      for (int m = 0; m < 17; ++m) { const int k = j + 16 * m; if (k <= kM) zf[k] = mv[m] * mv[m]; }
      QuadG::sync();
      double s0 = 0.0, s1 = 0.0, s4 = 0.0, s5 = 0.0;
      if (first) s3 = 0.0;
#pragma unroll
      for (int m = 0; m < 17; ++m) {
        const int k = j + 16 * m;
        const double p = (double)(mv[m] * mv[m]);
        if (k >= 1 && k <= kM) {
          s0 += p;
          s1 += (F0 * k) * p;
        }
        // band energies (:779-853): the bins strictly inside a band here, its two edge bins (weighted) once per row below
        if (k > Q.band_iL[0] && k < Q.band_iR[0]) s4 += p;
        if (k > Q.band_iL[1] && k < Q.band_iR[1]) s5 += p;
      }
      const double frameSum = QuadG::sum(s0, nullptr), sumA = QuadG::sum(s1, nullptr), fluxS = QuadG::sum(s3, nullptr);
      double bandE0 = QuadG::sum(s4, nullptr), bandE1 = QuadG::sum(s5, nullptr);
      bandE0 += (double)zf[Q.band_iL[0]] * Q.band_wL[0]; bandE0 += (double)zf[Q.band_iR[0]] * Q.band_wR[0];      // (launch_compare: iL < iR <= 256)
      bandE1 += (double)zf[Q.band_iL[1]] * Q.band_wL[1]; bandE1 += (double)zf[Q.band_iR[1]] * Q.band_wR[1];
      float ctr = 0.0f;
      if (frameSum != 0.0) ctr = (float)(sumA / frameSum);
      // (what is final already is written now: the sums it comes from need not live to the end of the pass)
      if (store && j == 0) {
        const int nBins = kK - 1;
        float *sp = rawB + 26;
        sp[0] = (float)(bandE0 / (double)nBins);
        sp[1] = (float)(bandE1 / (double)nBins);
        const double flux = fluxS / (double)nBins;
        sp[6] = (!first && flux > 0.0) ? (float)sqrt(flux) : 0.0f;
        sp[7] = ctr;
        const double Nind = (double)nBins;
        const double deno = (Nind * Q.slope_S2f - Q.slope_Sf * Q.slope_Sf);
        double slope = 0.0;
        if (deno != 0.0) slope = (Nind * sumA - Q.slope_Sf * frameSum) / deno;
        sp[12] = (float)(slope * (Nind - 1.0));
      }
      QPHASE(4);
      // roll-off (:1102-1122): inclusive prefix of the powers of bins 1 .. 256, first bin whose prefix reaches the share
      {
        const double rollOff[4] = {0.25, 0.50, 0.75, 0.90};
        double off = 0.0;
        int kro[4] = {0, 0, 0, 0};                           // the bin at which share i is reached (one bin of the row meets the test)
#pragma unroll
        for (int m = 0; m < 17; ++m) {
          const int k = j + 16 * m;
          const float mq = fresh(mv[m]);
          double c = (k >= 1 && k <= kM) ? (double)(mq * mq) : 0.0;
          c += shr0_d<1>(c); c += shr0_d<2>(c); c += shr0_d<4>(c); c += shr0_d<8>(c);
          const double tot = lane_d(c, row_base4, 15);
          c += off;
          const double up = shr0_d<1>(c);
          const double before = (j == 0) ? off : up;      // the prefix at bin k - 1
          if (k >= 1 && k <= kM) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const double th = rollOff[i] * frameSum;
              if (c >= th && (k == 1 || !(before >= th))) kro[i] = k;
            }
          }
          off += tot;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = QuadG::sum_i(kro[i], nullptr);
          if (store && j == i) rawB[26 + 2 + i] = (float)(F0 * kk);
        }
      }
      QPHASE(5);
      // harmonicity (:1484-1513): alternating peaks / valleys, distance to the previous flagged bin
      {
        float hc[17];
        float carry = 0.0f;
        bool has_carry = false;
#pragma unroll
        for (int m = 0; m < 17; ++m) {
          const int k = j + 16 * m;
          const float pf = mv[m] * mv[m];
          bool flag = false;
          if (k >= 3 && k < kM - 1) {
            const float a2 = zf[k - 2], a1 = zf[k - 1], b1 = zf[k + 1], b2 = zf[k + 2];
            flag = (a2 < pf && a1 < pf && pf > b1 && pf > b2) || (a2 > pf && a1 > pf && pf < b1 && pf < b2);
          }
          const unsigned long long mask = __ballot(flag);
          const unsigned rowmask = (unsigned)(mask >> (lane64 & 48)) & 0xffffu;
          const unsigned lower = rowmask & ((1u << j) - 1u);
          const float prev_in = lane_f(pf, row_base4, lower ? 31 - __clz(lower) : 0);
          const float last_in = lane_f(pf, row_base4, rowmask ? 31 - __clz(rowmask) : 0);
          float h = 0.0f;
          if (flag) {
            if (lower) h = fabsf(pf - prev_in);
            else if (has_carry) h = fabsf(pf - carry);
          }
          hc[m] = h;
          if (rowmask) { carry = last_in; has_carry = true; }
        }
        QuadG::sync();                                     // (the powers have been read: the two chains' terms take their place)
#pragma unroll
        for (int m = 0; m < 17; ++m) { const int k = j + 16 * m; if (k >= 1 && k <= kM) zf[256 + k - 1] = hc[m]; }
      }
      QPHASE(6);
      // the per-bin terms of entropy, variance, skewness, kurtosis, and sharpness' chain terms
      double e0 = 0.0, e1 = 0.0, e2m = 0.0, e3 = 0.0;
      {
        const double entropy_floor = 0.0000001;
        double dn = frameSum;
        if (dn < (float)entropy_floor) dn = (float)entropy_floor;
        // the frame's two divisors -- its power sum and log 2 -- by their reciprocals (div_by). Its range holds by construction:
        // int16 samples are at most 1 in magnitude, a 320-sample frame's bins at most 320, the power sum at most 2.7e7 (and at
        // least the floor 1e-7); quotients and x log x stay between 1e-15 and 1e15
        const double inv_dn = 1.0 / dn;
        constexpr double kLog2 = 0.693147180559945286226764, kInvLog2 = 1.0 / kLog2;
        const double F0e = fresh(F0);                        // (the bins' frequencies F0 k again, not seventeen doubles kept from the sums' loop)
#pragma unroll
        for (int m = 0; m < 17; ++m) {
          const int k = j + 16 * m;
          const float me = fresh(mv[m]);
          const double p = (double)(me * me);
          if (k >= 1 && k <= kM) {
            zf[k - 1] = (float)(s_sharp[k - 1] * p);         // :1455 / :1469
            double vv = p;
            if (vv <= entropy_floor) vv = entropy_floor;
            const double ln = div_by(vv, dn, inv_dn);
            const double xl = ln * log_d<true>(ln, s_log);
            e0 += (ln > 0.0) ? div_by(xl, kLog2, kInvLog2) : 0.0;
            const double t1 = F0e * fresh(k) - (double)ctr;
            double mm = t1 * t1 * p;
            e1 += mm; mm *= t1; e2m += mm; e3 += mm * t1;
          }
          if (m % CQ_BINS_GROUP == CQ_BINS_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
      const double ent = QuadG::sum(e0, nullptr), mom2 = QuadG::sum(e1, nullptr), mom3 = QuadG::sum(e2m, nullptr), mom4 = QuadG::sum(e3, nullptr);
      QPHASE(7);
      // the two FLOAT_DMEM chains (:1435-1471 sharpness, :1485-1499 harmonicity): terms of bins 1 .. 256 in order, lane 0 adds the
      // first chain, lane 1 the second
      QuadG::sync();
      float chain = 0.0f;
      if (j < 2) chain = seq_sum_f32(zf + 256 * j, 0, 256);
      const float sumAA = lane_f(chain, row_base4, 0), ptp = lane_f(chain, row_base4, 1);
      if (store && j == 0) {
        const int nBins = kK - 1;
        float *sp = rawB + 26;
        float c2 = 0.0f;
        if (frameSum != 0.0) c2 = (float)(sumAA / frameSum);
        sp[13] = (float)(0.11 * c2);
        sp[8] = (float)(-ent);
        const double sumB = frameSum;
        const double sigma2 = (sumB != 0.0) ? mom2 / sumB : 0.0;
        sp[9] = (float)sigma2;
        sp[10] = (sigma2 <= 0.0) ? 0.0f : (float)(mom3 / (sumB * sigma2 * sqrt(sigma2)));
        sp[11] = (sigma2 == 0.0) ? 0.0f : (float)(mom4 / (sumB * sigma2 * sigma2));
        float ptpSum = ptp;
        ptpSum /= 2.0f;
        ptpSum /= (float)nBins;
        sp[14] = ptpSum;
      }
      QuadG::sync();
    }
  }
}

}  // namespace smilehip
