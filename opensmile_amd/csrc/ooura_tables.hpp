// Host-side tables of the reference-order real FFT (lld_ooura.hpp): product code, independent of oracle/.
//
// The reference's transform is rdft() of src/dspcore/fftsg.c:322-363. Unrolled, it is a fixed network of float add / sub /
// mul: radix-4 levels over a tree of nodes (type 1 = what cftmdl1 :2441-2548 does, type 2 = cftmdl2 :2551-2682; the root
// level is cftf1st :1801-2005 / cftb1st :2007-2211 with interpolated odd twiddles; children of a type-1 node are of types
// (1, 2, 1, 1), of a type-2 node (1, 2, 1, 2); the last levels are cftf161/162 :2706-3045 or the 8-point leaves cftf081/082
// :3048-3179), then a bit reversal (bitrv2 :913 / bitrv2conj :1260) and rftfsub :3241-3263 / rftbsub :3266-3288. A parallel
// schedule of the same network gives the same bits; what the device needs from the host is, per level and butterfly, the
// twiddle VALUES the reference's code would have in its registers -- they come out of makewt :660-719 / makect :741-760
// (float arithmetic on libm results) and of cftf1st's interpolation, restated in ooura_tables.cpp.
#pragma once
#include <cstdint>
#include <vector>

namespace smilehip {

constexpr int kOouraMaxLevels = 7;

struct OouraHost {
  int n = 0;            // real length, 64 ... 8192
  int M = 0;            // complex length n/2
  int logM = 0;
  int nlev = 0;         // radix-4 levels through LDS (the 8-point leaves not counted)
  int leaf8 = 0;        // 1: M = 2 * 4^k, the network ends in 8-point leaves
  // tw: 4 floats per record. Level l, quarter q = M >> (2l + 2): type-1 table = q records (w1r, w1i, w3r, w3i) at off1[l];
  // type-2 table = 2q records (ar, ai, br, bi), (cr, ci, dr, di) at off2[l] (-1: the level has no such table).
  std::vector<float> tw;
  int off1[kOouraMaxLevels], off2[kOouraMaxLevels];
  std::vector<float> rft;   // (wkr, wki) of rftfsub / rftbsub for k = 0 .. M/2 - 1 (entry 0 unused)
  float wn4r = 0.f, wk1r = 0.f, wk1i = 0.f;   // w[1]; the 8-point type-2 leaf's twiddle (&w[nw - 8])[2..3]
  std::vector<float> w, c;  // the raw makewt / makect tables (tests compare them with the reference's)
};

// 0 on success, -1 if n is not a power of two in [64, 8192]
int make_ooura(int n, OouraHost &h);

}  // namespace smilehip
