// One wave per transform, register-resident: the reference-order network of lld_ooura.hpp for M = 256 (FFT 512: four points
// per lane) and M = 512 (FFT 1024: eight points per lane) without any LDS traffic between the levels.
//
// A radix-4 level needs the four points of a butterfly -- they differ in two bits of the point index e -- in one lane's
// registers. The 8 (9) index bits are split into "register bits" (which of the lane's 4 (8) points) and "lane bits" (which
// lane); between two levels the next level's two bits are swapped into the register side by 2 x 2 transpositions between a
// register bit and a lane bit: one v_permlane32_swap / v_permlane16_swap per register pair for lane bits 5 / 4, two
// bank-masked DPP row rotations for lane bits 3 / 2, a quad permutation + three selects for lane bits 1 / 0 -- vector-ALU moves
// only. (The in-place LDS form costs 8 ds_read_b64 + 8 ds_write_b64 per level and lane with 4-way bank conflicts in the levels
// with q <= 4, plus 8..16-way conflicts in the bit-reversed reads of the accessor; the LDS pipe is what bounds the frame kernels.)
// The butterflies are oo_bf1 / oo_bf2 / oo_leaf8_* of lld_ooura.hpp, the tables the same: same operations, same operands,
// same bits.
//
//   M = 256: regs (e7 e6) | lanes e5..e0 -> L0 (q 64) -> swap regs <-> lane bits 5 4 -> L1 (q 16) -> swap <-> 3 2 -> L2 (q 4)
//            -> swap <-> 1 0 -> L3 (q 1): lane = node, e = 4 lane + r
//   M = 512: regs (e8 e7 e6) | lanes e5..e0 -> L0 (q 128, two butterflies) -> swap e8 <-> lane bit 5 -> L1 (q 32) -> swap
//            e7 <-> 4, e6 <-> 3 -> L2 (q 8) -> swap e5 <-> 2, e4 <-> 1, e3 <-> 0 -> 8-point leaf: lane = leaf, e = 8 lane + r
//
// Output: point e holds spectrum index F = bitrev(e); the lane writes its 4 (8) values to z[oo_pos(F)], oo_pos(F) = F ^ ((F >> 4)
// & 3): sixteen consecutive lanes then hit sixteen different bank pairs (ds_write_b64), and readers that walk F with consecutive
// lanes stay conflict-free (the xor permutes within aligned groups of four).
#pragma once
#include <hip/hip_runtime.h>

#include "lld_ooura.hpp"

namespace smilehip {

__device__ __forceinline__ int oo_pos(int F) { return F ^ ((F >> 4) & 3); }

// 2 x 2 transposition between a register bit and lane bit B: a = the value with register bit 0, b = with register bit 1.
// Afterwards, in a lane whose bit B is 0: a = own a, b = partner's a; in a lane whose bit B is 1: a = partner's b, b = own b.
template <int B>
__device__ __forceinline__ void oo_xpose(float &a, float &b, int lane) {
  const int ia = __float_as_int(a), ib = __float_as_int(b);
  if constexpr (B == 5) {
    const auto r = __builtin_amdgcn_permlane32_swap(ia, ib, false, false);   // a[32..63] <-> b[0..31]
    a = __int_as_float(r[0]); b = __int_as_float(r[1]);
  } else if constexpr (B == 4) {
    const auto r = __builtin_amdgcn_permlane16_swap(ia, ib, false, false);   // odd rows of a <-> even rows of b
    a = __int_as_float(r[0]); b = __int_as_float(r[1]);
  } else if constexpr (B == 3) {
    // row_ror:8 = lane ^ 8 within a row; bank_mask picks the lanes that take the partner's value
    const int na = __builtin_amdgcn_update_dpp(ia, ib, 0x128, 0xf, 0xc, false);
    const int nb = __builtin_amdgcn_update_dpp(ib, ia, 0x128, 0xf, 0x3, false);
    a = __int_as_float(na); b = __int_as_float(nb);
  } else if constexpr (B == 2) {
    // lanes with bit 2 set (banks 1, 3) read lane - 4 (row_ror:4), the others lane + 4 (row_ror:12)
    const int na = __builtin_amdgcn_update_dpp(ia, ib, 0x124, 0xf, 0xa, false);
    const int nb = __builtin_amdgcn_update_dpp(ib, ia, 0x12c, 0xf, 0x5, false);
    a = __int_as_float(na); b = __int_as_float(nb);
  } else {
    const bool hi = (lane >> B) & 1;
    const int t = hi ? ia : ib;
    const int r = (B == 1) ? __builtin_amdgcn_update_dpp(0, t, 0x4e, 0xf, 0xf, true)     // quad_perm [2,3,0,1]
                           : __builtin_amdgcn_update_dpp(0, t, 0xb1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    a = __int_as_float(hi ? r : ia);
    b = __int_as_float(hi ? ib : r);
  }
}
template <int B>
__device__ __forceinline__ void oo_xpose2(float2 &a, float2 &b, int lane) {
  oo_xpose<B>(a.x, b.x, lane);
  oo_xpose<B>(a.y, b.y, lane);
}

// one butterfly of a level with quarter q (node type, kind, flags from (node, c) as in ooura_levels)
// off1: the level's type-1 table (records), its type-2 table follows at off1 + q
template <bool BWD0>
__device__ __forceinline__ void oo_level_bf(float2 &p0, float2 &p1, float2 &p2, float2 &p3, const OouraTab &T, int level, int q,
                                            int off1, unsigned node, int c) {
  const int type = oo_node_type(node, level);
  bool negA3 = false;
  if (q == 1 && level > 0) {
    const unsigned g = node & 3u;
    negA3 = (oo_node_type(node >> 2, level - 1) == 1) ? (g == 3u) : (g >= 2u);
  }
  if (type == 1) {
    const int kind = (c == 0) ? 0 : ((2 * c == q) ? 2 : 1);
    const float4 tw = (q > 1) ? T.tw[off1 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    oo_bf1<BWD0>(p0, p1, p2, p3, kind, tw, T.wn4r, negA3);
  } else {
    const int kind = (c == 0) ? 0 : 1;
    float4 ta = make_float4(0.f, 0.f, 0.f, 0.f), tb = ta;
    if (q > 1) { ta = T.tw[off1 + q + 2 * c]; tb = T.tw[off1 + q + 2 * c + 1]; }
    const bool swap23 = (2 * c == q) || (q == 4 && c < 3);
    oo_bf2(p0, p1, p2, p3, kind, ta, tb, T.wn4r, swap23, negA3);
  }
}

// ---- M = 256. in(e): point e of the level input (natural order); the result goes to z[oo_pos(bitrev8(e))].
template <bool BWD, class In>
__device__ __forceinline__ void oo_wave256(float2 *z, const OouraTab &T, int lane, In in) {
  float2 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = in(lane + 64 * k);
  oo_level_bf<BWD>(v[0], v[1], v[2], v[3], T, 0, 64, 0, 0u, lane);
  oo_xpose2<5>(v[0], v[2], lane); oo_xpose2<5>(v[1], v[3], lane);
  oo_xpose2<4>(v[0], v[1], lane); oo_xpose2<4>(v[2], v[3], lane);
  oo_level_bf<false>(v[0], v[1], v[2], v[3], T, 1, 16, 64, (unsigned)lane >> 4, lane & 15);
  oo_xpose2<3>(v[0], v[2], lane); oo_xpose2<3>(v[1], v[3], lane);
  oo_xpose2<2>(v[0], v[1], lane); oo_xpose2<2>(v[2], v[3], lane);
  oo_level_bf<false>(v[0], v[1], v[2], v[3], T, 2, 4, 112, (unsigned)lane >> 2, lane & 3);
  oo_xpose2<1>(v[0], v[2], lane); oo_xpose2<1>(v[1], v[3], lane);
  oo_xpose2<0>(v[0], v[1], lane); oo_xpose2<0>(v[2], v[3], lane);
  oo_level_bf<false>(v[0], v[1], v[2], v[3], T, 3, 1, 0, (unsigned)lane, 0);
  const int fl = (int)(__brev((unsigned)lane) >> 26);       // bitrev8(4 lane + r) = 64 bitrev2(r) + bitrev6(lane)
  z[oo_pos(fl)] = v[0];
  z[oo_pos(fl + 128)] = v[1];
  z[oo_pos(fl + 64)] = v[2];
  z[oo_pos(fl + 192)] = v[3];
}

// ---- M = 512. The result goes to z[oo_pos(bitrev9(e))].
template <bool BWD, class In>
__device__ __forceinline__ void oo_wave512(float2 *z, const OouraTab &T, int lane, In in) {
  float2 v[8];                                             // v[4 e8 + 2 e7 + e6]
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = in(lane + 64 * m);
  // L0, q = 128: k = (e8 e7), c = 64 e6 + lane
  oo_level_bf<BWD>(v[0], v[2], v[4], v[6], T, 0, 128, 0, 0u, lane);
  oo_level_bf<BWD>(v[1], v[3], v[5], v[7], T, 0, 128, 0, 0u, lane + 64);
  // register bit 2 (e8) <-> lane bit 5 (e5): v[4 e5 + 2 e7 + e6]
#pragma unroll
  for (int r = 0; r < 4; ++r) oo_xpose2<5>(v[r], v[r + 4], lane);
  // L1, q = 32: k = 2 e6 + e5, node = 2 e8 + e7 (e8 = lane bit 5), c = lane & 31
  {
    const unsigned n8 = ((unsigned)lane >> 5) << 1;
    oo_level_bf<false>(v[0], v[4], v[1], v[5], T, 1, 32, 128, n8, lane & 31);
    oo_level_bf<false>(v[2], v[6], v[3], v[7], T, 1, 32, 128, n8 + 1u, lane & 31);
  }
  // register bit 1 (e7) <-> lane bit 4 (e4); register bit 0 (e6) <-> lane bit 3 (e3): v[4 e5 + 2 e4 + e3]
#pragma unroll
  for (int r = 0; r < 8; ++r) if (!(r & 2)) oo_xpose2<4>(v[r], v[r + 2], lane);
#pragma unroll
  for (int r = 0; r < 8; ++r) if (!(r & 1)) oo_xpose2<3>(v[r], v[r + 1], lane);
  // L2, q = 8: k = 2 e4 + e3, node = 8 e8 + 4 e7 + 2 e6 + e5 = 2 (lane >> 3) + e5, c = lane & 7
  {
    const unsigned nb = ((unsigned)lane >> 3) << 1;
    oo_level_bf<false>(v[0], v[1], v[2], v[3], T, 2, 8, 224, nb, lane & 7);
    oo_level_bf<false>(v[4], v[5], v[6], v[7], T, 2, 8, 224, nb + 1u, lane & 7);
  }
  // register bits (e5 e4 e3) <-> lane bits 2 1 0 (e2 e1 e0): v[4 e2 + 2 e1 + e0], lane = (e8 .. e3) = leaf
#pragma unroll
  for (int r = 0; r < 4; ++r) oo_xpose2<2>(v[r], v[r + 4], lane);
#pragma unroll
  for (int r = 0; r < 8; ++r) if (!(r & 2)) oo_xpose2<1>(v[r], v[r + 2], lane);
#pragma unroll
  for (int r = 0; r < 8; ++r) if (!(r & 1)) oo_xpose2<0>(v[r], v[r + 1], lane);
  if (oo_node_type((unsigned)lane, 3) == 1) oo_leaf8_t1(v, T.wn4r);
  else oo_leaf8_t2(v, T.wn4r, T.wk1r, T.wk1i);
  const int fl = (int)(__brev((unsigned)lane) >> 26);       // bitrev9(8 lane + r) = 64 bitrev3(r) + bitrev6(lane)
  z[oo_pos(fl)] = v[0];
  z[oo_pos(fl + 256)] = v[1];
  z[oo_pos(fl + 128)] = v[2];
  z[oo_pos(fl + 384)] = v[3];
  z[oo_pos(fl + 64)] = v[4];
  z[oo_pos(fl + 320)] = v[5];
  z[oo_pos(fl + 192)] = v[6];
  z[oo_pos(fl + 448)] = v[7];
}

// ---- the wave-level interface of the chains: register form for M = 256 / 512, the in-place LDS form otherwise.
// MC: M when the caller knows it at compile time (256 / 512: only that form is generated), 0 = decided at run time,
// -1 = the in-place LDS form only (any other length known at compile time)
template <int MC = 0>
__device__ __forceinline__ bool oo_wave_nat(const OouraTab &T) { return MC < 0 ? false : (MC ? (MC == 256 || MC == 512) : (T.M == 256 || T.M == 512)); }
template <int MC = 0>
__device__ __forceinline__ float2 oo_wave_at(const float2 *z, const OouraTab &T, int F) {
  return oo_wave_nat<MC>(T) ? z[oo_pos(F)] : z[oo_rev(F, T.logM)];
}
__device__ __forceinline__ void oo_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
struct OoWaveG {                                           // the one-wave group of lld_blocks.hpp, restated here to keep this header free-standing
  __device__ static __forceinline__ int tid() { return threadIdx.x & 63; }
  __device__ static __forceinline__ int size() { return 64; }
  __device__ static __forceinline__ void sync() { oo_wave_sync(); }
};

// forward transform of 2M reals: load(i) = (x[2i], x[2i + 1]); z needs M pairs. Ends with a wave sync.
template <int MC = 0, class Load>
__device__ __forceinline__ void oo_wave_forward(float2 *z, const OouraTab &T, int lane, Load load) {
  if constexpr (MC < 0) ooura_forward<OoWaveG>(z, T, load);
  else if constexpr (MC == 256) { oo_wave256<false>(z, T, lane, load); oo_wave_sync(); }
  else if constexpr (MC == 512) { oo_wave512<false>(z, T, lane, load); oo_wave_sync(); }
  else if (T.M == 256) { oo_wave256<false>(z, T, lane, load); oo_wave_sync(); }
  else if (T.M == 512) { oo_wave512<false>(z, T, lane, load); oo_wave_sync(); }
  else ooura_forward<OoWaveG>(z, T, load);
}
// bin k (0 <= k <= M) as the standard DFT value (see ooura_bin)
template <int MC = 0>
__device__ __forceinline__ float2 oo_wave_bin(const float2 *z, const OouraTab &T, int k) {
  const int M = MC > 0 ? MC : T.M;
  if (k == 0) { const float2 a = oo_wave_at<MC>(z, T, 0); return make_float2(a.x + a.y, 0.0f); }
  if (k == M) { const float2 a = oo_wave_at<MC>(z, T, 0); return make_float2(a.x - a.y, 0.0f); }
  if (2 * k == M) { const float2 a = oo_wave_at<MC>(z, T, k); return make_float2(a.x, -a.y); }
  const int j = (2 * k < M) ? k : M - k;
  const float2 aj = oo_wave_at<MC>(z, T, j), ak = oo_wave_at<MC>(z, T, M - j);
  const float2 wk = T.rft[j];
  const float xr = aj.x - ak.x, xi = aj.y + ak.y;
  const float yr = wk.x * xr - wk.y * xi, yi = wk.x * xi + wk.y * xr;
  if (2 * k < M) return make_float2(aj.x - yr, -(aj.y - yi));
  return make_float2(ak.x + yr, -(ak.y - yi));
}

// inverse transform rdft(2M, -1): load(e) = (a[2e], a[2e + 1]) of the packed input. Output sample i: oo_wave_inverse_out.
template <int MC = 0, class Load>
__device__ __forceinline__ void oo_wave_inverse(float2 *z, const OouraTab &T, int lane, Load load) {
  if constexpr (MC < 0) { ooura_inverse<OoWaveG>(z, T, load); return; }
  if constexpr (MC == 0) { if (!oo_wave_nat(T)) { ooura_inverse<OoWaveG>(z, T, load); return; } }
  const int M = MC > 0 ? MC : T.M;
  // the element of the array after rdft :350-351 and rftbsub :3266-3288, computed where it is needed (the pair's other
  // member is computed by another lane: the same operations on the same operands, the same bits)
  const auto pre = [&](int e) {
    if (e == 0) {
      float2 a = load(0);
      a.y = 0.5f * (a.x - a.y);
      a.x -= a.y;
      return a;
    }
    if (2 * e == M) return load(e);
    const int j = (2 * e < M) ? e : M - e;
    const float2 aj = load(j), ak = load(M - j);
    const float2 wk = T.rft[j];
    const float xr = aj.x - ak.x, xi = aj.y + ak.y;
    const float yr = wk.x * xr + wk.y * xi, yi = wk.x * xi - wk.y * xr;
    return (2 * e < M) ? make_float2(aj.x - yr, aj.y - yi) : make_float2(ak.x + yr, ak.y - yi);
  };
  if constexpr (MC == 256) oo_wave256<true>(z, T, lane, pre);
  else if constexpr (MC == 512) oo_wave512<true>(z, T, lane, pre);
  else if (M == 256) oo_wave256<true>(z, T, lane, pre);
  else oo_wave512<true>(z, T, lane, pre);
  oo_wave_sync();
}
template <int MC = 0>
__device__ __forceinline__ float oo_wave_inverse_out(const float2 *z, const OouraTab &T, int i) {
  const float2 a = oo_wave_at<MC>(z, T, i >> 1);
  return (i & 1) ? -a.y : a.x;                               // bitrev2conj
}
// cAcf's use of the inverse transform (see oo_irfft_even); R must not alias z
template <int MC = 0>
__device__ __forceinline__ void oo_wave_irfft_even(const float *R, float2 *z, const OouraTab &T, float *out, float inv_norm,
                                                   bool take_abs, int lane) {
  const int M = MC > 0 ? MC : T.M;
  oo_wave_inverse<MC>(z, T, lane, [&](int e) { return e == 0 ? make_float2(R[0], R[M]) : make_float2(R[e], 0.0f); });
  for (int i = lane; i < M; i += 64) {
    const float v = oo_wave_inverse_out<MC>(z, T, i) / inv_norm;
    out[i] = take_abs ? fabsf(v) : v;
  }
  oo_wave_sync();
}

}  // namespace smilehip
