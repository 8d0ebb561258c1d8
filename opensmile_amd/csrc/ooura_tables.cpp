// See ooura_tables.hpp. Every float below is produced by the same sequence of float / double operations as the reference
// produces it with (fftsg.c:660-760, :1801-1850), so that the device multiplies by bit-identical constants.
#include "ooura_tables.hpp"

#include <cmath>
#include <cstring>

namespace smilehip {
namespace {

// makewt(nw, ip, w), fftsg.c:660-719 (ip is only used by the reference's own bit reversal)
void build_w(int nw, std::vector<float> &w) {
  w.assign(static_cast<size_t>(nw) + 8, 0.0f);
  if (nw <= 2) return;
  int nwh = nw >> 1;
  const float delta = static_cast<float>(std::atan(1.0)) / nwh;
  const float wn4r = static_cast<float>(std::cos(static_cast<double>(delta * nwh)));
  w[0] = 1.0f;
  w[1] = wn4r;
  if (nwh == 4) {
    w[2] = static_cast<float>(std::cos(static_cast<double>(delta * 2)));
    w[3] = static_cast<float>(std::sin(static_cast<double>(delta * 2)));
  } else if (nwh > 4) {
    w[2] = static_cast<float>(0.5 / std::cos(static_cast<double>(delta * 2)));
    w[3] = static_cast<float>(0.5 / std::cos(static_cast<double>(delta * 6)));
    for (int j = 4; j < nwh; j += 4) {
      w[j] = static_cast<float>(std::cos(static_cast<double>(delta * j)));
      w[j + 1] = static_cast<float>(std::sin(static_cast<double>(delta * j)));
      w[j + 2] = static_cast<float>(std::cos(static_cast<double>(3 * delta * j)));
      w[j + 3] = static_cast<float>(-std::sin(static_cast<double>(3 * delta * j)));
    }
  }
  int nw0 = 0;
  while (nwh > 2) {
    const int nw1 = nw0 + nwh;
    nwh >>= 1;
    w[nw1] = 1.0f;
    w[nw1 + 1] = wn4r;
    if (nwh == 4) {
      w[nw1 + 2] = w[nw0 + 4];
      w[nw1 + 3] = w[nw0 + 5];
    } else if (nwh > 4) {
      w[nw1 + 2] = 0.5f / w[nw0 + 4];
      w[nw1 + 3] = 0.5f / w[nw0 + 6];
      for (int j = 4; j < nwh; j += 4) {
        w[nw1 + j] = w[nw0 + 2 * j];
        w[nw1 + j + 1] = w[nw0 + 2 * j + 1];
        w[nw1 + j + 2] = w[nw0 + 2 * j + 2];
        w[nw1 + j + 3] = w[nw0 + 2 * j + 3];
      }
    }
    nw0 = nw1;
  }
}

// makect(nc, ip, c), fftsg.c:741-760
void build_c(int nc, std::vector<float> &c) {
  c.assign(static_cast<size_t>(nc) + 8, 0.0f);
  if (nc <= 1) return;
  const int nch = nc >> 1;
  const float delta = static_cast<float>(std::atan(1.0)) / nch;
  c[0] = static_cast<float>(std::cos(static_cast<double>(delta * nch)));
  c[nch] = 0.5f * c[0];
  for (int j = 1; j < nch; ++j) {
    c[j] = static_cast<float>(0.5 * std::cos(static_cast<double>(delta * j)));
    c[nc - j] = static_cast<float>(0.5 * std::sin(static_cast<double>(delta * j)));
  }
}

struct Rec1 { float w1r, w1i, w3r, w3i; };
struct Rec2 { float ar, ai, br, bi, cr, ci, dr, di; };

Rec1 mirror(const Rec1 &e) { return Rec1{e.w1i, e.w1r, e.w3i, e.w3r}; }

}  // namespace

int make_ooura(int n, OouraHost &h) {
  if (n < 64 || n > 8192 || (n & (n - 1))) return -1;
  h = OouraHost();
  h.n = n;
  h.M = n / 2;
  while ((1 << h.logM) < h.M) ++h.logM;
  const int nw = n / 4, M = h.M;
  build_w(nw, h.w);
  build_c(n / 4, h.c);
  const std::vector<float> &w = h.w;
  h.wn4r = w[1];
  h.wk1r = w[nw - 8 + 2];
  h.wk1i = w[nw - 8 + 3];
  for (int l = 0; l < kOouraMaxLevels; ++l) h.off1[l] = h.off2[l] = -1;

  int level = 0;
  for (int q = M / 4; q >= 1; q >>= 2, ++level) {
    if (q == 2) { h.leaf8 = 1; break; }
    h.nlev = level + 1;
    if (q == 1) break;                                        // no twiddles in the last level of a 16-point leaf
    // ---- type 1
    std::vector<Rec1> t1(static_cast<size_t>(q), Rec1{0, 0, 0, 0});
    if (level == 0) {
      // cftf1st: even c reads the table, odd c interpolates between its neighbours; c > q/2 mirrors q - c
      const float wn4r = w[1], csc1 = w[2], csc3 = w[3];
      auto entry = [&](int c) -> Rec1 {                       // table entry of an even c (0 and q/2 included)
        if (c == 0) return Rec1{1.0f, 0.0f, 1.0f, 0.0f};
        if (2 * c == q) return Rec1{wn4r, wn4r, -wn4r, -wn4r};
        return Rec1{w[2 * c], w[2 * c + 1], w[2 * c + 2], w[2 * c + 3]};
      };
      for (int c = 1; 2 * c < q; ++c) {
        Rec1 e;
        if (c & 1) {
          const Rec1 p = entry(c - 1), x = entry(c + 1);
          e = Rec1{csc1 * (p.w1r + x.w1r), csc1 * (p.w1i + x.w1i), csc3 * (p.w3r + x.w3r), csc3 * (p.w3i + x.w3i)};
        } else {
          e = entry(c);
        }
        t1[c] = e;
        t1[q - c] = mirror(e);
      }
    } else if (q == 4) {
      // cftf161, first half (w = &w[nw - 8])
      const float wk1r = w[nw - 8 + 2], wk1i = w[nw - 8 + 3];
      t1[1] = Rec1{wk1r, wk1i, wk1i, -wk1r};
      t1[3] = Rec1{wk1i, wk1r, wk1r, -wk1i};
    } else {
      // cftmdl1(8q, a, &w[nw - 4q])
      const float *W = w.data() + nw - 4 * q;
      for (int c = 1; 2 * c < q; ++c) {
        const Rec1 e{W[4 * c], W[4 * c + 1], W[4 * c + 2], W[4 * c + 3]};
        t1[c] = e;
        t1[q - c] = mirror(e);
      }
    }
    h.off1[level] = static_cast<int>(h.tw.size() / 4);
    for (const Rec1 &e : t1) { h.tw.push_back(e.w1r); h.tw.push_back(e.w1i); h.tw.push_back(e.w3r); h.tw.push_back(e.w3i); }
    // ---- type 2 (never at the root)
    if (level == 0) continue;
    std::vector<Rec2> t2(static_cast<size_t>(q), Rec2{0, 0, 0, 0, 0, 0, 0, 0});
    if (q == 4) {
      // cftf162, first half (w = &w[nw - 32])
      const float *W = w.data() + nw - 32;
      const float wk1r = W[4], wk1i = W[5], wk3r = W[6], wk3i = -W[7], wk2r = W[8], wk2i = W[9];
      t2[1] = Rec2{wk1r, wk1i, wk3i, wk3r, wk3r, -wk3i, wk1r, wk1i};
      t2[2] = Rec2{wk2r, wk2i, wk2i, wk2r, wk2i, -wk2r, wk2r, -wk2i};
      t2[3] = Rec2{wk3r, wk3i, wk1i, wk1r, wk1i, wk1r, wk3i, -wk3r};
    } else {
      // cftmdl2(8q, a, &w[nw - 8q])
      const float *W = w.data() + nw - 8 * q;
      for (int c = 1; 2 * c < q; ++c) {
        const int kr = 4 * q - 4 * c;
        const float wk1r = W[4 * c], wk1i = W[4 * c + 1], wk3r = W[4 * c + 2], wk3i = W[4 * c + 3];
        const float wd1i = W[kr], wd1r = W[kr + 1], wd3i = W[kr + 2], wd3r = W[kr + 3];
        t2[c] = Rec2{wk1r, wk1i, wd1r, wd1i, wk3r, wk3i, wd3r, wd3i};
        t2[q - c] = Rec2{wd1i, wd1r, wk1i, wk1r, wd3i, wd3r, wk3i, wk3r};
      }
      const float wk1r = W[2 * q], wk1i = W[2 * q + 1];
      t2[q / 2] = Rec2{wk1r, wk1i, wk1i, wk1r, wk1i, -wk1r, wk1r, -wk1i};
    }
    h.off2[level] = static_cast<int>(h.tw.size() / 4);
    for (const Rec2 &e : t2) {
      const float v[8] = {e.ar, e.ai, e.br, e.bi, e.cr, e.ci, e.dr, e.di};
      h.tw.insert(h.tw.end(), v, v + 8);
    }
  }
  if (h.tw.empty()) h.tw.assign(4, 0.0f);
  // rftfsub / rftbsub (ks = 1): wkr = 0.5 - c[nc - k], wki = c[k]
  h.rft.assign(static_cast<size_t>(M), 0.0f);
  const int nc = n / 4;
  for (int k = 1; k < M / 2; ++k) {
    h.rft[2 * k] = 0.5f - h.c[nc - k];
    h.rft[2 * k + 1] = h.c[k];
  }
  return 0;
}

}  // namespace smilehip
