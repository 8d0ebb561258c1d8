// Sixteen lanes per transform, four transforms per wave: the reference-order network of lld_ooura.hpp for M = 256 (FFT 512)
// with sixteen points per lane.
//
// lld_ooura_wave.hpp gives a whole wave to one transform: four points per lane, and the 64 butterflies of a level -- one
// per lane -- are of mixed type and kind, so the wave walks through every variant of the butterfly (type 1 generic / c = 0 /
// c = q/2, type 2 generic / c = 0) with most lanes masked off: ~2.5 x the instructions of one butterfly per level, plus six
// register <-> lane transpositions. Here a lane owns 16 points, a 16-lane row one transform, and the wave runs the same
// network for four frames at once:
//   L0 (q = 64) and L1 (q = 16) pair index bits (e7 e6) and (e5 e4): with e = 16 r + j (register r, lane j of the row) both
//      pairs are register bits -- four butterflies per lane and level, the node (hence the type) a compile-time constant of the
//      register index, the two special kinds confined to lane 0 / lane 8 of a row;
//   one 16 x 16 transposition of (re, im) pairs through the row's own LDS buffer (b64 stores and loads, lines 17 pairs apart):
//      e = 16 j + r;
//   L2 (q = 4) and L3 (q = 1) pair (e3 e2) and (e1 e0), register bits again: c -- hence the kind -- is a compile-time
//      constant, the node's type depends on the lane (both types are walked, each with its one kind).
// The butterflies are oo_bf1 / oo_bf2 of lld_ooura.hpp through oo_level_bf of lld_ooura_wave.hpp, the tables the same: same
// operations, same operands, same bits as the wave form and the in-place LDS form (tests/test_gpu_ooura.py).
// Output: register r of lane j holds spectrum index F = 16 bitrev4(r) + bitrev4(j); oo_quad_store writes it to z[oo_pos(F)] of
// the same buffer, where oo_wave_bin<256> / oo_wave_inverse_out<256> read it.
#pragma once
#include <hip/hip_runtime.h>

#include "lld_ooura_wave.hpp"

// OO_QUAD_BF_FENCE: a scheduling fence behind every butterfly (two butterflies' operands and twiddles in flight instead of four)
#ifndef OO_QUAD_BF_FENCE
#define OO_QUAD_BF_FENCE __builtin_amdgcn_sched_barrier(0)
#endif

namespace smilehip {

constexpr int kQuadXRow = 17;                               // pairs per line of a row's 16 x 16 transposition: lane j reads line j --
                                                            // 17 pairs apart, sixteen different bank pairs
constexpr int kQuadZPairs = 16 * kQuadXRow;                 // a row's buffer: 272 (re, im) pairs -- the transposition, then the 256 results

__device__ __forceinline__ int oo_brev4(int x) { return (int)(__brev((unsigned)x) >> 28); }

// The four radix-4 levels on the 16 points of a lane. v[r] = point 16 r + j on entry (j = lane & 15). zx: the ROW's buffer
// (kQuadZPairs pairs; dead before the call, free again after it).
template <bool BWD>
__device__ __forceinline__ void oo_quad256(float2 (&v)[16], const OouraTab &T, float2 *zx, int lane) {
  const int j = lane & 15;
  // L0, q = 64: node 0, butterfly s works on e = c + {0, 64, 128, 192}, c = 16 s + j -> registers s, s + 4, s + 8, s + 12
#pragma unroll
  for (int s = 0; s < 4; ++s) { oo_level_bf<BWD>(v[s], v[4 + s], v[8 + s], v[12 + s], T, 0, 64, 0, 0u, 16 * s + j); OO_QUAD_BF_FENCE; }
  // L1, q = 16: node k = (e7 e6), c = j: e = 64 k + j + {0, 16, 32, 48} -> registers 4 k + {0, 1, 2, 3}
#pragma unroll
  for (int k = 0; k < 4; ++k) { oo_level_bf<false>(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3], T, 1, 16, 64, (unsigned)k, j); OO_QUAD_BF_FENCE; }
  // 16 x 16 transposition inside the row: element (register r, lane j) -> (register j, lane r): e = 16 j + r
  {
    float2 *xw = zx + j;
    const float2 *xr = zx + j * kQuadXRow;
#pragma unroll
    for (int r = 0; r < 16; ++r) xw[r * kQuadXRow] = v[r];
    oo_wave_sync();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = xr[r];
    oo_wave_sync();
  }
  // L2, q = 4: node j (e7 .. e4), butterfly c works on e = 16 j + c + {0, 4, 8, 12} -> registers c, c + 4, c + 8, c + 12
#pragma unroll
  for (int c = 0; c < 4; ++c) { oo_level_bf<false>(v[c], v[4 + c], v[8 + c], v[12 + c], T, 2, 4, 112, (unsigned)j, c); OO_QUAD_BF_FENCE; }
  // L3, q = 1: node 4 j + m, e = 16 j + 4 m + {0, 1, 2, 3} -> registers 4 m + {0, 1, 2, 3}
#pragma unroll
  for (int m = 0; m < 4; ++m) { oo_level_bf<false>(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3], T, 3, 1, 0, (unsigned)(4 * j + m), 0); OO_QUAD_BF_FENCE; }
}

// v[r] (spectrum index 16 bitrev4(r) + bitrev4(j)) -> z[oo_pos(F)], z = the row's buffer. Ends with a wave sync.
// LOW_HALF: only spectrum indices F < 128 (the even registers) -- all that oo_quad_inverse_outs reads for output samples
// 0 .. 255; the other half of the last level's results is then dead code
template <bool LOW_HALF = false>
__device__ __forceinline__ void oo_quad_store(const float2 (&v)[16], float2 *z, int lane) {
  const int fj = oo_brev4(lane & 15);
  // oo_pos(16 R + fj) = 16 R + (fj ^ (R & 3)): four lane-dependent addresses, the rest immediate offsets
  float2 *zc[4] = {z + fj, z + (fj ^ 1), z + (fj ^ 2), z + (fj ^ 3)};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    constexpr int kRev[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    if (!LOW_HALF || kRev[r] < 8) zc[kRev[r] & 3][16 * kRev[r]] = v[r];
  }
  oo_wave_sync();
}
// output sample j + 16 it of the inverse transform (oo_wave_inverse_out<256>): pair 8 it + (j >> 1) at oo_pos = 8 it + ((j >> 1) ^ c),
// c = (it >> 1) & 3 -- again four lane-dependent addresses
__device__ __forceinline__ void oo_quad_inverse_outs(const float2 *z, int lane, float (&o)[16]) {
  const int jh = (lane & 15) >> 1;
  const bool odd = lane & 1;
  const float2 *zc[4] = {z + jh, z + (jh ^ 1), z + (jh ^ 2), z + (jh ^ 3)};
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const float2 a = zc[(it >> 1) & 3][8 * it];
    o[it] = odd ? -a.y : a.x;
  }
}

// forward transform of 512 reals: load(i) = (x[2i], x[2i + 1]); afterwards oo_wave_bin<256>(z, T, k) is bin k
template <class Load>
__device__ __forceinline__ void oo_quad_forward(float2 *z, const OouraTab &T, int lane, Load load) {
  float2 v[16];
  const int j = lane & 15;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = load(16 * r + j);
  oo_quad256<false>(v, T, z, lane);
  oo_quad_store(v, z, lane);
}

// inverse transform rdft(512, -1): load(e) = (a[2e], a[2e + 1]) of the packed input; afterwards oo_wave_inverse_out<256>(z, T, i)
// is output sample i. The element of the array after rdft :350-351 and rftbsub :3266-3288 is computed where it is needed (see
// oo_wave_inverse).
template <class Load>
__device__ __forceinline__ void oo_quad_inverse(float2 *z, const OouraTab &T, int lane, Load load) {
  constexpr int M = 256;
  const auto pre = [&](int e) {
    if (e == 0) {
      float2 a = load(0);
      a.y = 0.5f * (a.x - a.y);
      a.x -= a.y;
      return a;
    }
    if (2 * e == M) return load(e);
    const int jj = (2 * e < M) ? e : M - e;
    const float2 aj = load(jj), ak = load(M - jj);
    const float2 wk = T.rft[jj];
    const float xr = aj.x - ak.x, xi = aj.y + ak.y;
    const float yr = wk.x * xr + wk.y * xi, yi = wk.x * xi - wk.y * xr;
    return (2 * e < M) ? make_float2(aj.x - yr, aj.y - yi) : make_float2(ak.x + yr, ak.y - yi);
  };
  float2 v[16];
  const int j = lane & 15;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    v[r] = pre(16 * r + j);
    if (r % 4 == 3) __builtin_amdgcn_sched_barrier(0);     // (four elements' loads in flight at a time)
  }
  oo_quad256<true>(v, T, z, lane);
  oo_quad_store(v, z, lane);
}

// cAcf's use of the inverse transform (see oo_irfft_even): load(e) = the packed input pair e (its source must not alias z, it may
// be `out`: every input is read before the first output is written); lags 0 .. 255 to out
template <class Load>
__device__ __forceinline__ void oo_quad_irfft_even(float2 *z, const OouraTab &T, float *out, float inv_norm, bool take_abs, int lane,
                                                   Load load) {
  constexpr int M = 256;
  oo_quad_inverse(z, T, lane, load);
  float o[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) o[it] = oo_wave_inverse_out<M>(z, T, (lane & 15) + 16 * it);
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const float v = o[it] / inv_norm;
    out[(lane & 15) + 16 * it] = take_abs ? fabsf(v) : v;
  }
  oo_wave_sync();
}

// ---- a REAL packed spectrum held in registers: P[m] = R[j + 16 m] (m <= 16; P[16] = R[256] counts in lane 0 only), the input
// cAcf gives rdft(512, -1): a[0] = R[0], a[1] = R[M], a[2e] = R[e], a[2e + 1] = 0. The element e = 16 r + j of the array after
// rdft :350-351 / rftbsub :3266-3288 needs R[e] (the lane's own register r) and R[M - e]: lane (16 - j) & 15's register 15 - r
// (lane 0: its own register 16 - r) -- two DPP moves instead of a round trip through LDS.
__device__ __forceinline__ float oo_quad_mirror1(float x) {       // lane j <- lane (16 - j) & 15 of its row
  int v = __float_as_int(x);
  v = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);   // row_mirror: lane j <- 15 - j
  v = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, true);   // row_ror:1:  lane j <- j - 1
  return __int_as_float(v);
}
__device__ __forceinline__ void oo_quad_inverse_real(float2 *z, const OouraTab &T, int lane, const float (&P)[17]) {
  constexpr int M = 256;
  const int j = lane & 15;
  float2 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int e = 16 * r + j;
    const float own = P[r];
    const float t = oo_quad_mirror1(P[15 - r]);
    const float mir = (j == 0) ? P[16 - r] : t;                  // R[M - e]
    const bool lower = 2 * e < M;
    const int jj = lower ? e : M - e;
    const float2 aj = make_float2(lower ? own : mir, 0.0f), ak = make_float2(lower ? mir : own, 0.0f);
    const float2 wk = T.rft[(2 * e == M) ? 0 : jj];
    const float xr = aj.x - ak.x, xi = aj.y + ak.y;
    const float yr = wk.x * xr + wk.y * xi, yi = wk.x * xi - wk.y * xr;
    float2 res = lower ? make_float2(aj.x - yr, aj.y - yi) : make_float2(ak.x + yr, ak.y - yi);
    if (r == 0) {                                                // e == 0 in lane 0
      float2 a = make_float2(own, mir);
      a.y = 0.5f * (a.x - a.y);
      a.x -= a.y;
      if (j == 0) res = a;
    }
    if (r == 8 && j == 0) res = make_float2(own, 0.0f);          // 2 e == M
    v[r] = res;
    if (r % 4 == 3) __builtin_amdgcn_sched_barrier(0);           // (four elements' loads in flight at a time)
  }
  oo_quad256<true>(v, T, z, lane);
  oo_quad_store<true>(v, z, lane);
}
// cAcf on that input: lags j + 16 it, it < 16, of the lane's row to out[it] (registers)
__device__ __forceinline__ void oo_quad_irfft_even_real(float2 *z, const OouraTab &T, float (&out)[16], float inv_norm, bool take_abs,
                                                        int lane, const float (&P)[17]) {
  // (M = 256: the sizes below are written out)
  oo_quad_inverse_real(z, T, lane, P);
  oo_quad_inverse_outs(z, lane, out);
  // RN(a / b) for the pass's one divisor: y = RN(1 / b) by the division itself, then Markstein's sequence (see f0_div_by in
  // lld_f0.hip: q0 = RN(a y) within two ulps, one residual step makes it faithful, the second returns the correctly rounded
  // quotient) -- five operations instead of the division's eleven, for operands well inside the normal range; zeros (whose
  // sign the sequence would lose), tiny, huge and non-finite values take the division.
  const float y = 1.0f / inv_norm;
  bool plain = false;
#pragma unroll
  for (int it = 0; it < 16; ++it) { const float m = fabsf(out[it]); plain |= !(m > 0x1p-60f && m < 0x1p60f); }
  const bool norm_ok = inv_norm > 0x1p-30f && inv_norm < 0x1p30f;
  if (!norm_ok || __builtin_amdgcn_ballot_w64(plain) != 0) {
#pragma unroll
    for (int it = 0; it < 16; ++it) out[it] = out[it] / inv_norm;
  } else {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const float a = out[it];
      const float q0 = a * y;
      const float r0 = __builtin_fmaf(-q0, inv_norm, a);
      const float q1 = __builtin_fmaf(r0, y, q0);
      const float r1 = __builtin_fmaf(-q1, inv_norm, a);
      out[it] = __builtin_fmaf(r1, y, q1);
    }
  }
  if (take_abs) {
#pragma unroll
    for (int it = 0; it < 16; ++it) out[it] = fabsf(out[it]);
  }
  oo_wave_sync();
}

}  // namespace smilehip
