// The reference-order real FFT on the device: rdft() of src/dspcore/fftsg.c:322-363 as a parallel schedule of the SAME
// operation network (see ooura_tables.hpp for its shape). Every float add / sub / mul below is one the reference executes,
// with the same operands in the same order (the library is built with -ffp-contract=off: no FMA where the reference has
// none), so forward and inverse transforms are bit-identical to the reference's -- zero signs included -- for every input.
// The fast MFCC kernel (lld_mfcc512.hip) keeps its own transform (3e-7 of the frame's largest bin, inside the tolerance);
// every reference-order chain and the per-component operators use this one.
//
// Layout: the half-length complex array lives in LDS as (re, im) pairs, z[e] = point e. A group G (one wave, or the whole
// workgroup) runs the radix-4 levels in place: butterfly b of a level with quarter q works on node b / q, points
// node * 4q + c + {0, q, 2q, 3q}, c = b % q. After the levels point p holds the spectrum value of index bitrev(p); the
// accessors below read through the bit reversal and apply rftfsub (forward) or the conjugation of bitrv2conj (inverse).
#pragma once
#include <hip/hip_runtime.h>

namespace smilehip {

constexpr int kOouraLevels = 7;

struct OouraTab {                 // device view of OouraHost (ooura_tables.hpp)
  const float4 *tw;               // level tables
  const float2 *rft;              // (wkr, wki), k = 0 .. M/2 - 1
  int M, logM, nlev, leaf8;
  int n_tw;                       // records (float4) in tw
  float wn4r, wk1r, wk1i;
};
// Table layout (ooura_tables.cpp): level 0 (quarter q0 = M/4): q0 type-1 records at 0; every further level with q > 1: q type-1
// records, then 2q type-2 records. (No offset arrays in the struct: a dynamically indexed member would put it into scratch.)

// type of node `node` of level `level` (ooura_tables.hpp): trailing child-3 digits inherit, child 1 is type 2
__device__ __forceinline__ int oo_node_type(unsigned node, int level) {
  for (int i = 0; i < level; ++i) {
    const unsigned d = node & 3u;
    if (d != 3u) return d == 1u ? 2 : 1;
    node >>= 2;
  }
  return 1;
}

// type-1 butterfly (cftmdl1 / cftf1st / cftf161; BWD: cftb1st). kind 0: c == 0, 1: generic, 2: c == q/2.
template <bool BWD>
__device__ __forceinline__ void oo_bf1(float2 &p0, float2 &p1, float2 &p2, float2 &p3, int kind, const float4 tw, float wn4r,
                                       bool negA3) {
  float x0r, x0i, x1r, x1i, x2r, x2i, x3r, x3i, tr, ti, ur, ui;
  if constexpr (!BWD) {
    x0r = p0.x + p2.x; x0i = p0.y + p2.y;
    x1r = p0.x - p2.x; x1i = p0.y - p2.y;
  } else {
    x0r = p0.x + p2.x; x0i = -p0.y - p2.y;
    x1r = p0.x - p2.x; x1i = -p0.y + p2.y;
  }
  {
    const float sr = p1.x + p3.x, si = p1.y + p3.y, dr = p1.x - p3.x, di = p1.y - p3.y;
    x2r = negA3 ? dr : sr; x2i = negA3 ? di : si;
    x3r = negA3 ? sr : dr; x3i = negA3 ? si : di;
  }
  if constexpr (!BWD) {
    p0.x = x0r + x2r; p0.y = x0i + x2i;
    p1.x = x0r - x2r; p1.y = x0i - x2i;
    tr = x1r - x3i; ti = x1i + x3r;
    ur = x1r + x3i; ui = x1i - x3r;
  } else {
    p0.x = x0r + x2r; p0.y = x0i - x2i;
    p1.x = x0r - x2r; p1.y = x0i + x2i;
    tr = x1r + x3i; ti = x1i + x3r;
    ur = x1r - x3i; ui = x1i - x3r;
  }
  if (kind == 1) {
    p2.x = tw.x * tr - tw.y * ti;
    p2.y = tw.x * ti + tw.y * tr;
    p3.x = tw.z * ur + tw.w * ui;
    p3.y = tw.z * ui - tw.w * ur;
  } else if (kind == 0) {
    p2.x = tr; p2.y = ti;
    p3.x = ur; p3.y = ui;
  } else {
    p2.x = wn4r * (tr - ti);
    p2.y = wn4r * (ti + tr);
    p3.x = -wn4r * (ur + ui);
    p3.y = -wn4r * (ui - ur);
  }
}

// type-2 butterfly (cftmdl2 / cftf162). kind 0: c == 0, 1: generic with (a, b) = ta, (c, d) = tb.
__device__ __forceinline__ void oo_bf2(float2 &p0, float2 &p1, float2 &p2, float2 &p3, int kind, const float4 ta, const float4 tb,
                                       float wn4r, bool swap23, bool negA3) {
  const float x0r = p0.x - p2.y, x0i = p0.y + p2.x;
  const float x1r = p0.x + p2.y, x1i = p0.y - p2.x;
  float x2r, x2i, x3r, x3i;
  {
    const float mr = p1.x - p3.y, pi = p1.y + p3.x, pr = p1.x + p3.y, mi = p1.y - p3.x;
    x2r = negA3 ? pr : mr; x2i = negA3 ? mi : pi;
    x3r = negA3 ? mr : pr; x3i = negA3 ? pi : mi;
  }
  if (kind == 0) {
    float y0r = wn4r * (x2r - x2i), y0i = wn4r * (x2i + x2r);
    p0.x = x0r + y0r; p0.y = x0i + y0i;
    p1.x = x0r - y0r; p1.y = x0i - y0i;
    y0r = wn4r * (x3r - x3i); y0i = wn4r * (x3i + x3r);
    p2.x = x1r - y0i; p2.y = x1i + y0r;
    p3.x = x1r + y0i; p3.y = x1i - y0r;
  } else {
    float y0r = ta.x * x0r - ta.y * x0i, y0i = ta.x * x0i + ta.y * x0r;
    float y2r = ta.z * x2r - ta.w * x2i, y2i = ta.z * x2i + ta.w * x2r;
    p0.x = y0r + y2r; p0.y = y0i + y2i;
    p1.x = y0r - y2r; p1.y = y0i - y2i;
    y0r = tb.x * x1r + tb.y * x1i; y0i = tb.x * x1i - tb.y * x1r;
    y2r = tb.z * x3r + tb.w * x3i; y2i = tb.z * x3i - tb.w * x3r;
    const float sr = y0r + y2r, si = y0i + y2i, dr = y0r - y2r, di = y0i - y2i;
    p2.x = swap23 ? dr : sr; p2.y = swap23 ? di : si;
    p3.x = swap23 ? sr : dr; p3.y = swap23 ? si : di;
  }
}

// the 8-point leaves: cftf081 :3048-3107 (type 1), cftf082 :3110-3179 (type 2); a[k] = point k
__device__ __forceinline__ void oo_leaf8_t1(float2 (&a)[8], float wn4r) {
  float x0r = a[0].x + a[4].x, x0i = a[0].y + a[4].y;
  float x1r = a[0].x - a[4].x, x1i = a[0].y - a[4].y;
  float x2r = a[2].x + a[6].x, x2i = a[2].y + a[6].y;
  float x3r = a[2].x - a[6].x, x3i = a[2].y - a[6].y;
  const float y0r = x0r + x2r, y0i = x0i + x2i;
  const float y2r = x0r - x2r, y2i = x0i - x2i;
  const float y1r = x1r - x3i, y1i = x1i + x3r;
  const float y3r = x1r + x3i, y3i = x1i - x3r;
  x0r = a[1].x + a[5].x; x0i = a[1].y + a[5].y;
  x1r = a[1].x - a[5].x; x1i = a[1].y - a[5].y;
  x2r = a[3].x + a[7].x; x2i = a[3].y + a[7].y;
  x3r = a[3].x - a[7].x; x3i = a[3].y - a[7].y;
  const float y4r = x0r + x2r, y4i = x0i + x2i;
  const float y6r = x0r - x2r, y6i = x0i - x2i;
  x0r = x1r - x3i; x0i = x1i + x3r;
  x2r = x1r + x3i; x2i = x1i - x3r;
  const float y5r = wn4r * (x0r - x0i), y5i = wn4r * (x0r + x0i);
  const float y7r = wn4r * (x2r - x2i), y7i = wn4r * (x2r + x2i);
  a[4].x = y1r + y5r; a[4].y = y1i + y5i;
  a[5].x = y1r - y5r; a[5].y = y1i - y5i;
  a[6].x = y3r - y7i; a[6].y = y3i + y7r;
  a[7].x = y3r + y7i; a[7].y = y3i - y7r;
  a[0].x = y0r + y4r; a[0].y = y0i + y4i;
  a[1].x = y0r - y4r; a[1].y = y0i - y4i;
  a[2].x = y2r - y6i; a[2].y = y2i + y6r;
  a[3].x = y2r + y6i; a[3].y = y2i - y6r;
}

__device__ __forceinline__ void oo_leaf8_t2(float2 (&a)[8], float wn4r, float wk1r, float wk1i) {
  const float y0r = a[0].x - a[4].y, y0i = a[0].y + a[4].x;
  const float y1r = a[0].x + a[4].y, y1i = a[0].y - a[4].x;
  float x0r = a[2].x - a[6].y, x0i = a[2].y + a[6].x;
  const float y2r = wn4r * (x0r - x0i), y2i = wn4r * (x0i + x0r);
  x0r = a[2].x + a[6].y; x0i = a[2].y - a[6].x;
  const float y3r = wn4r * (x0r - x0i), y3i = wn4r * (x0i + x0r);
  x0r = a[1].x - a[5].y; x0i = a[1].y + a[5].x;
  const float y4r = wk1r * x0r - wk1i * x0i, y4i = wk1r * x0i + wk1i * x0r;
  x0r = a[1].x + a[5].y; x0i = a[1].y - a[5].x;
  const float y5r = wk1i * x0r - wk1r * x0i, y5i = wk1i * x0i + wk1r * x0r;
  x0r = a[3].x - a[7].y; x0i = a[3].y + a[7].x;
  const float y6r = wk1i * x0r - wk1r * x0i, y6i = wk1i * x0i + wk1r * x0r;
  x0r = a[3].x + a[7].y; x0i = a[3].y - a[7].x;
  const float y7r = wk1r * x0r - wk1i * x0i, y7i = wk1r * x0i + wk1i * x0r;
  x0r = y0r + y2r; x0i = y0i + y2i;
  float x1r = y4r + y6r, x1i = y4i + y6i;
  a[0].x = x0r + x1r; a[0].y = x0i + x1i;
  a[1].x = x0r - x1r; a[1].y = x0i - x1i;
  x0r = y0r - y2r; x0i = y0i - y2i;
  x1r = y4r - y6r; x1i = y4i - y6i;
  a[2].x = x0r - x1i; a[2].y = x0i + x1r;
  a[3].x = x0r + x1i; a[3].y = x0i - x1r;
  x0r = y1r - y3i; x0i = y1i + y3r;
  x1r = y5r - y7r; x1i = y5i - y7i;
  a[4].x = x0r + x1r; a[4].y = x0i + x1i;
  a[5].x = x0r - x1r; a[5].y = x0i - x1i;
  x0r = y1r + y3i; x0i = y1i - y3r;
  x1r = y5r + y7r; x1i = y5i + y7i;
  a[6].x = x0r - x1i; a[6].y = x0i + x1r;
  a[7].x = x0r + x1i; a[7].y = x0i - x1r;
}

// the butterfly levels of cftfsub (BWD = false) / cftbsub (true), in place on z[0..M); ends with a G::sync()
template <class G, bool BWD>
__device__ __forceinline__ void ooura_levels(float2 *z, const OouraTab &T) {
  const int M = T.M;
  int level = 0, lq = T.logM - 2, off = 0;
  for (; level < T.nlev; ++level, lq -= 2) {
    const int q = 1 << lq;
    const float4 *t1 = T.tw + off;
    const float4 *t2 = t1 + q;
    off += level == 0 ? q : 3 * q;
    for (int b = G::tid(); b < (M >> 2); b += G::size()) {
      const unsigned node = (unsigned)b >> lq;
      const int c = b & (q - 1);
      const int type = oo_node_type(node, level);
      bool negA3 = false;
      if (q == 1 && level > 0) {                            // the last level inside cftf161 / cftf162
        const unsigned g = node & 3u;
        negA3 = (oo_node_type(node >> 2, level - 1) == 1) ? (g == 3u) : (g >= 2u);
      }
      float2 *p = z + ((size_t)node << (lq + 2)) + c;
      float2 p0 = p[0], p1 = p[q], p2 = p[2 * q], p3 = p[3 * q];
      if (type == 1) {
        const int kind = (c == 0) ? 0 : ((2 * c == q) ? 2 : 1);
        const float4 tw = (kind == 1) ? t1[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (BWD && level == 0) oo_bf1<true>(p0, p1, p2, p3, kind, tw, T.wn4r, negA3);
        else oo_bf1<false>(p0, p1, p2, p3, kind, tw, T.wn4r, negA3);
      } else {
        const int kind = (c == 0) ? 0 : 1;
        float4 ta = make_float4(0.f, 0.f, 0.f, 0.f), tb = ta;
        if (kind) { ta = t2[2 * c]; tb = t2[2 * c + 1]; }
        const bool swap23 = (2 * c == q) || (q == 4 && c < 3);
        oo_bf2(p0, p1, p2, p3, kind, ta, tb, T.wn4r, swap23, negA3);
      }
      p[0] = p0; p[q] = p1; p[2 * q] = p2; p[3 * q] = p3;
    }
    G::sync();
  }
  if (T.leaf8) {
    for (int leaf = G::tid(); leaf < (M >> 3); leaf += G::size()) {
      float2 a[8];
      float2 *p = z + 8 * leaf;
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = p[k];
      if (oo_node_type((unsigned)leaf, level) == 1) oo_leaf8_t1(a, T.wn4r);
      else oo_leaf8_t2(a, T.wn4r, T.wk1r, T.wk1i);
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k] = a[k];
    }
    G::sync();
  }
}

__device__ __forceinline__ int oo_rev(int e, int logM) { return (int)(__brev((unsigned)e) >> (32 - logM)); }

// Forward transform of the 2M reals x[0..2M): load(i) = (x[2i], x[2i + 1]), every i in [0, M) asked for exactly once.
template <class G, class Load>
__device__ __forceinline__ void ooura_forward(float2 *z, const OouraTab &T, Load load) {
  for (int i = G::tid(); i < T.M; i += G::size()) z[i] = load(i);
  G::sync();
  ooura_levels<G, false>(z, T);
}

// Bin k (0 <= k <= M) of the forward transform, as the STANDARD DFT value X[k] = sum x[n] e^{-2 pi i nk / 2M}: the packed
// output of rdft is a[2k] = X.x, a[2k + 1] = -X.y (fftsg.c:103-117), a[0] = X[0].x, a[1] = X[M].x. rftfsub :3241-3263 and
// the a[0] / a[1] step of rdft :346-348 are applied here, per bin.
__device__ __forceinline__ float2 ooura_bin(const float2 *z, const OouraTab &T, int k) {
  const int M = T.M;
  if (k == 0) { const float2 a = z[0]; return make_float2(a.x + a.y, 0.0f); }
  if (k == M) { const float2 a = z[0]; return make_float2(a.x - a.y, 0.0f); }
  if (2 * k == M) { const float2 a = z[oo_rev(k, T.logM)]; return make_float2(a.x, -a.y); }
  const int j = (2 * k < M) ? k : M - k;                     // the loop index of rftfsub (complex), its partner is M - j
  const float2 aj = z[oo_rev(j, T.logM)], ak = z[oo_rev(M - j, T.logM)];
  const float2 wk = T.rft[j];
  const float xr = aj.x - ak.x, xi = aj.y + ak.y;
  const float yr = wk.x * xr - wk.y * xi, yi = wk.x * xi + wk.y * xr;
  if (2 * k < M) return make_float2(aj.x - yr, -(aj.y - yi));
  return make_float2(ak.x + yr, -(ak.y - yi));
}

// Inverse transform rdft(2M, -1, a): load(e) = (a[2e], a[2e + 1]) of the packed input (load(0) = (a[0], a[1])). Afterwards
// ooura_inverse_out(z, T, i) is output sample i, 0 <= i < 2M.
template <class G, class Load>
__device__ __forceinline__ void ooura_inverse(float2 *z, const OouraTab &T, Load load) {
  const int M = T.M;
  for (int j = G::tid(); j <= (M >> 1); j += G::size()) {
    if (j == 0) {                                            // rdft :350-351
      float2 a = load(0);
      a.y = 0.5f * (a.x - a.y);
      a.x -= a.y;
      z[0] = a;
    } else if (2 * j == M) {
      z[j] = load(j);
    } else {                                                 // rftbsub :3266-3288
      float2 aj = load(j), ak = load(M - j);
      const float2 wk = T.rft[j];
      const float xr = aj.x - ak.x, xi = aj.y + ak.y;
      const float yr = wk.x * xr + wk.y * xi, yi = wk.x * xi - wk.y * xr;
      aj.x -= yr; aj.y -= yi;
      ak.x += yr; ak.y -= yi;
      z[j] = aj;
      z[M - j] = ak;
    }
  }
  G::sync();
  ooura_levels<G, true>(z, T);
}
__device__ __forceinline__ float ooura_inverse_out(const float2 *z, const OouraTab &T, int i) {
  const float2 a = z[oo_rev(i >> 1, T.logM)];
  return (i & 1) ? -a.y : a.x;                               // bitrv2conj
}


// cAcf's use of the inverse transform (acf.cpp:308-343): the packed spectrum (R[0], R[M], R[1], 0, R[2], 0, ...) of the real
// values R[0..M], rdft(2M, -1), lags 0 .. M-1 divided by inv_norm (acfCepsNormOutput: Nsrc = M + 1; otherwise 1), |.| for the ACF
template <class G>
__device__ __forceinline__ void oo_irfft_even(const float *R, float2 *z, const OouraTab &T, float *out, float inv_norm,
                                              bool take_abs) {
  const int M = T.M;
  ooura_inverse<G>(z, T, [&](int e) { return e == 0 ? make_float2(R[0], R[M]) : make_float2(R[e], 0.0f); });
  for (int i = G::tid(); i < M; i += G::size()) {
    const float v = ooura_inverse_out(z, T, i) / inv_norm;     // acf.cpp:321-325: (FLOAT_DMEM)data / (FLOAT_DMEM)Nsrc
    out[i] = take_abs ? fabsf(v) : v;
  }
  G::sync();
}

// floats of LDS a staged copy of the tables takes, and the staging itself (all threads of the workgroup; the caller
// synchronises): the returned view points into lds
__host__ __device__ inline int oo_table_floats(const OouraTab &T) { return T.tw ? 4 * T.n_tw + T.M : 0; }
// ALWAYS: the caller knows the tables exist (its launch condition) -- without the early return both table pointers are LDS
// pointers on every path, and the compiler addresses them as such (ds_read with immediate offsets); with it, S.rft is "global or
// LDS" and every read of it becomes a flat load through a 64-bit address
template <bool ALWAYS = false>
__device__ __forceinline__ OouraTab oo_stage_tables(const OouraTab &T, float *lds, int tid, int nthreads) {
  OouraTab S = T;
  if (!ALWAYS && !T.tw) return S;
  float4 *tw = reinterpret_cast<float4 *>(lds);
  float2 *rft = reinterpret_cast<float2 *>(lds + 4 * T.n_tw);
  for (int i = tid; i < T.n_tw; i += nthreads) tw[i] = T.tw[i];
  for (int i = tid; i < (T.M >> 1); i += nthreads) rft[i] = T.rft[i];
  S.tw = tw;
  S.rft = rft;
  return S;
}

}  // namespace smilehip
