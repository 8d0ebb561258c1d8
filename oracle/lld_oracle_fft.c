/* oracle/lld_oracle_fft.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The reference's real FFT, rdft() of src/dspcore/fftsg.c:322-363 (Ooura's split-radix package, float), restated as what it
 * is once the call tree is unrolled: a fixed, data-independent network of IEEE float add / sub / mul (the reference is built
 * without FMA contraction), i.e.
 *
 *   forward  = [ radix-4 levels over a tree of nodes ] -> bit reversal -> rftfsub -> (a0, a1) fix-up
 *   inverse  = (a0, a1) fix-up -> rftbsub -> [ same levels, first one in its conjugating form ] -> bit reversal + conj
 *
 * The tree (derived from cftfsub :817-862, cftrec4 :2319-2338, cfttree :2341-2373, cftleaf :2376-2438, cftfx41 :2685-2703):
 * the half-length complex array (Nc = n/2 points) is the root node; a node of 4q points is split by one radix-4 level into
 * four children of q points at offsets 0, q, 2q, 3q. Nodes come in two types: type 1 does what cftmdl1 :2441-2548 does
 * (butterfly, then twiddle), type 2 what cftmdl2 :2551-2682 does (twiddle inside a rotated butterfly). The root is type 1 (its
 * level is cftf1st :1801-2005 / cftb1st :2007-2211, whose odd twiddles are interpolated); the children of a type-1 node are
 * of types (1, 2, 1, 1), those of a type-2 node (1, 2, 1, 2). The last levels are what cftf161 :2706-2862 / cftf162
 * :2865-3045 (16 points = levels q = 4 and q = 1) or cftf081 :3048-3107 / cftf082 :3110-3179 (8 points) spell out.
 *
 * Every butterfly below is written in the reference's expression form (operand order included), so that the result is
 * bit-identical -- including the sign of zeros -- to rdft() for every input. tests/test_ooura_fft.py pins this file against
 * the REAL rdft (oracle/_ref/libref_dsp.so, compiled from fftsg.c) for n = 64 ... 8192, both directions, on random, sparse
 * and signed-zero inputs. The oracle's chains use this transform when no reference hook is installed, so that on a box
 * without oracle/_ref the oracle still has the reference's bits.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "lld_oracle.h"

typedef struct oo_plan {
  int n, nc, nw;
  float *w;            /* makewt table (fftsg.c:660-719), nw = n/4 entries */
  float *c;            /* makect table (fftsg.c:741-760), n/4 entries */
  int *rev;            /* bit reversal of the complex index */
  struct oo_plan *next;
} oo_plan;

static oo_plan *g_plans = 0;

/* makewt, fftsg.c:660-719: float arithmetic where the reference has float operands, libm on the promoted value */
static void oo_makewt(int nw, float *w)
{
  int j, nwh, nw0, nw1;
  float delta, wn4r, wk1r, wk1i, wk3r, wk3i;
  if (nw <= 2) return;
  nwh = nw >> 1;
  delta = (float)atan(1.0) / nwh;
  wn4r = (float)cos(delta * nwh);
  w[0] = 1; w[1] = wn4r;
  if (nwh == 4) {
    w[2] = (float)cos(delta * 2);
    w[3] = (float)sin(delta * 2);
  } else if (nwh > 4) {
    w[2] = (float)(0.5 / cos(delta * 2));
    w[3] = (float)(0.5 / cos(delta * 6));
    for (j = 4; j < nwh; j += 4) {
      w[j] = (float)cos(delta * j);
      w[j + 1] = (float)sin(delta * j);
      w[j + 2] = (float)cos(3 * delta * j);
      w[j + 3] = (float)(-sin(3 * delta * j));
    }
  }
  nw0 = 0;
  while (nwh > 2) {                       /* the sub-tables are copies of every second entry of the table above */
    nw1 = nw0 + nwh;
    nwh >>= 1;
    w[nw1] = 1; w[nw1 + 1] = wn4r;
    if (nwh == 4) {
      w[nw1 + 2] = w[nw0 + 4];
      w[nw1 + 3] = w[nw0 + 5];
    } else if (nwh > 4) {
      wk1r = w[nw0 + 4]; wk3r = w[nw0 + 6];
      w[nw1 + 2] = (float)0.5 / wk1r;
      w[nw1 + 3] = (float)0.5 / wk3r;
      for (j = 4; j < nwh; j += 4) {
        wk1r = w[nw0 + 2 * j]; wk1i = w[nw0 + 2 * j + 1];
        wk3r = w[nw0 + 2 * j + 2]; wk3i = w[nw0 + 2 * j + 3];
        w[nw1 + j] = wk1r; w[nw1 + j + 1] = wk1i;
        w[nw1 + j + 2] = wk3r; w[nw1 + j + 3] = wk3i;
      }
    }
    nw0 = nw1;
  }
}

/* makect, fftsg.c:741-760 */
static void oo_makect(int nc, float *c)
{
  int j, nch;
  float delta;
  if (nc <= 1) return;
  nch = nc >> 1;
  delta = (float)atan(1.0) / nch;
  c[0] = (float)cos(delta * nch);
  c[nch] = (float)0.5 * c[0];
  for (j = 1; j < nch; j++) {
    c[j] = (float)(0.5 * cos(delta * j));
    c[nc - j] = (float)(0.5 * sin(delta * j));
  }
}

static oo_plan *oo_get_plan(int n)
{
  oo_plan *p;
  for (p = g_plans; p; p = p->next) if (p->n == n) return p;
  p = (oo_plan *)calloc(1, sizeof(*p));
  p->n = n; p->nc = n / 2; p->nw = n / 4;
  p->w = (float *)calloc((size_t)p->nw + 8, sizeof(float));
  p->c = (float *)calloc((size_t)n / 4 + 8, sizeof(float));
  oo_makewt(p->nw, p->w);
  oo_makect(n / 4, p->c);
  {
    int bits = 0, i, j;
    while ((1 << bits) < p->nc) bits++;
    p->rev = (int *)malloc(sizeof(int) * (size_t)p->nc);
    for (i = 0; i < p->nc; i++) {
      int r = 0;
      for (j = 0; j < bits; j++) if (i & (1 << j)) r |= 1 << (bits - 1 - j);
      p->rev[i] = r;
    }
  }
  p->next = g_plans; g_plans = p;
  return p;
}

/* type of node `node` (index among the 4^level nodes of its level): strip trailing child-3 digits (they inherit), then
 * child 1 is type 2, children 0 and 2 are type 1; the root is type 1 */
static int oo_node_type(unsigned node, int level)
{
  int i;
  for (i = 0; i < level; i++) {
    unsigned d = node & 3u;
    if (d != 3u) return d == 1u ? 2 : 1;
    node >>= 2;
  }
  return 1;
}

enum { K_TRIVIAL = 0, K_GENERIC = 1, K_MID = 2 };

/* type-1 butterfly on the points p0..p3 (float pairs). tw = (w1r, w1i, w3r, w3i). bwd: the cftb1st form (root level of the
 * inverse). swapA3: the consumer forms of cftf161/162's last level where the fourth input enters with the opposite sign. */
static void oo_bf1(float *p0, float *p1, float *p2, float *p3, int kind, const float *tw, float wn4r, int bwd, int negA3)
{
  float x0r, x0i, x1r, x1i, x2r, x2i, x3r, x3i, tr, ti, ur, ui;
  if (!bwd) {
    x0r = p0[0] + p2[0]; x0i = p0[1] + p2[1];
    x1r = p0[0] - p2[0]; x1i = p0[1] - p2[1];
  } else {
    x0r = p0[0] + p2[0]; x0i = -p0[1] - p2[1];
    x1r = p0[0] - p2[0]; x1i = -p0[1] + p2[1];
  }
  if (!negA3) {
    x2r = p1[0] + p3[0]; x2i = p1[1] + p3[1];
    x3r = p1[0] - p3[0]; x3i = p1[1] - p3[1];
  } else {
    x2r = p1[0] - p3[0]; x2i = p1[1] - p3[1];
    x3r = p1[0] + p3[0]; x3i = p1[1] + p3[1];
  }
  if (!bwd) {
    p0[0] = x0r + x2r; p0[1] = x0i + x2i;
    p1[0] = x0r - x2r; p1[1] = x0i - x2i;
    tr = x1r - x3i; ti = x1i + x3r;
    ur = x1r + x3i; ui = x1i - x3r;
  } else {
    p0[0] = x0r + x2r; p0[1] = x0i - x2i;
    p1[0] = x0r - x2r; p1[1] = x0i + x2i;
    tr = x1r + x3i; ti = x1i + x3r;
    ur = x1r - x3i; ui = x1i - x3r;
  }
  if (kind == K_TRIVIAL) {
    p2[0] = tr; p2[1] = ti;
    p3[0] = ur; p3[1] = ui;
  } else if (kind == K_GENERIC) {
    p2[0] = tw[0] * tr - tw[1] * ti;
    p2[1] = tw[0] * ti + tw[1] * tr;
    p3[0] = tw[2] * ur + tw[3] * ui;
    p3[1] = tw[2] * ui - tw[3] * ur;
  } else {
    p2[0] = wn4r * (tr - ti);
    p2[1] = wn4r * (ti + tr);
    p3[0] = -wn4r * (ur + ui);
    p3[1] = -wn4r * (ui - ur);
  }
}

/* type-2 butterfly. tw = (ar, ai, br, bi, cr, ci, dr, di): a, b multiply x0, x2 in the generic form, c, d multiply x1, x3
 * in the conjugate form. swap23: the (+, -) pair of the last two outputs goes to (p3, p2) instead of (p2, p3). */
static void oo_bf2(float *p0, float *p1, float *p2, float *p3, int kind, const float *tw, float wn4r, int swap23, int negA3)
{
  float x0r, x0i, x1r, x1i, x2r, x2i, x3r, x3i, y0r, y0i, y2r, y2i;
  x0r = p0[0] - p2[1]; x0i = p0[1] + p2[0];
  x1r = p0[0] + p2[1]; x1i = p0[1] - p2[0];
  if (!negA3) {
    x2r = p1[0] - p3[1]; x2i = p1[1] + p3[0];
    x3r = p1[0] + p3[1]; x3i = p1[1] - p3[0];
  } else {
    x2r = p1[0] + p3[1]; x2i = p1[1] - p3[0];
    x3r = p1[0] - p3[1]; x3i = p1[1] + p3[0];
  }
  if (kind == K_TRIVIAL) {               /* the j = 0 butterfly of cftmdl2 */
    y0r = wn4r * (x2r - x2i); y0i = wn4r * (x2i + x2r);
    p0[0] = x0r + y0r; p0[1] = x0i + y0i;
    p1[0] = x0r - y0r; p1[1] = x0i - y0i;
    y0r = wn4r * (x3r - x3i); y0i = wn4r * (x3i + x3r);
    p2[0] = x1r - y0i; p2[1] = x1i + y0r;
    p3[0] = x1r + y0i; p3[1] = x1i - y0r;
  } else {
    y0r = tw[0] * x0r - tw[1] * x0i; y0i = tw[0] * x0i + tw[1] * x0r;
    y2r = tw[2] * x2r - tw[3] * x2i; y2i = tw[2] * x2i + tw[3] * x2r;
    p0[0] = y0r + y2r; p0[1] = y0i + y2i;
    p1[0] = y0r - y2r; p1[1] = y0i - y2i;
    y0r = tw[4] * x1r + tw[5] * x1i; y0i = tw[4] * x1i - tw[5] * x1r;
    y2r = tw[6] * x3r + tw[7] * x3i; y2i = tw[6] * x3i - tw[7] * x3r;
    if (!swap23) {
      p2[0] = y0r + y2r; p2[1] = y0i + y2i;
      p3[0] = y0r - y2r; p3[1] = y0i - y2i;
    } else {
      p2[0] = y0r - y2r; p2[1] = y0i - y2i;
      p3[0] = y0r + y2r; p3[1] = y0i + y2i;
    }
  }
}

/* twiddles of butterfly c of a level with quarter q. top: the root level (cftf1st). Returns the kind; *swap23 for type 2. */
static int oo_twiddle(const oo_plan *P, int top, int type, int q, int c, float *tw, int *swap23)
{
  const float *w = P->w;
  const int nw = P->nw;
  *swap23 = 0;
  if (c == 0) return K_TRIVIAL;
  if (type == 1) {
    if (2 * c == q) return K_MID;
    if (top) {                                              /* cftf1st: table entries for even c, interpolated for odd c */
      const float wn4r = w[1], csc1 = w[2], csc3 = w[3];
      const int cc = c < q - c ? c : q - c;
      float e[4];
      if (cc & 1) {
        float pr[4], nx[4];
        if (cc - 1 == 0) { pr[0] = 1; pr[1] = 0; pr[2] = 1; pr[3] = 0; }
        else memcpy(pr, w + 2 * (cc - 1), sizeof(pr));
        if (2 * (cc + 1) == q) { nx[0] = wn4r; nx[1] = wn4r; nx[2] = -wn4r; nx[3] = -wn4r; }
        else memcpy(nx, w + 2 * (cc + 1), sizeof(nx));
        e[0] = csc1 * (pr[0] + nx[0]);
        e[1] = csc1 * (pr[1] + nx[1]);
        e[2] = csc3 * (pr[2] + nx[2]);
        e[3] = csc3 * (pr[3] + nx[3]);
      } else {
        memcpy(e, w + 2 * cc, sizeof(e));
      }
      if (cc == c) { tw[0] = e[0]; tw[1] = e[1]; tw[2] = e[2]; tw[3] = e[3]; }
      else         { tw[0] = e[1]; tw[1] = e[0]; tw[2] = e[3]; tw[3] = e[2]; }
      return K_GENERIC;
    }
    if (q == 4) {                                           /* cftf161, first half: w = &w[nw - 8] */
      const float wk1r = w[nw - 8 + 2], wk1i = w[nw - 8 + 3];
      if (c == 1) { tw[0] = wk1r; tw[1] = wk1i; tw[2] = wk1i; tw[3] = -wk1r; }
      else        { tw[0] = wk1i; tw[1] = wk1r; tw[2] = wk1r; tw[3] = -wk1i; }
      return K_GENERIC;
    }
    {                                                       /* cftmdl1(8q, a, &w[nw - 4q]) */
      const float *W = w + nw - 4 * q;
      const int cc = c < q - c ? c : q - c;
      if (cc == c) { tw[0] = W[4 * cc]; tw[1] = W[4 * cc + 1]; tw[2] = W[4 * cc + 2]; tw[3] = W[4 * cc + 3]; }
      else         { tw[0] = W[4 * cc + 1]; tw[1] = W[4 * cc]; tw[2] = W[4 * cc + 3]; tw[3] = W[4 * cc + 2]; }
      return K_GENERIC;
    }
  }
  /* type 2 */
  if (q == 4) {                                             /* cftf162, first half: w = &w[nw - 32] */
    const float *W = w + nw - 32;
    const float wk1r = W[4], wk1i = W[5], wk3r = W[6], wk3i = -W[7], wk2r = W[8], wk2i = W[9];
    if (c == 1) {
      tw[0] = wk1r; tw[1] = wk1i; tw[2] = wk3i; tw[3] = wk3r;
      tw[4] = wk3r; tw[5] = -wk3i; tw[6] = wk1r; tw[7] = wk1i; *swap23 = 1;
    } else if (c == 2) {
      tw[0] = wk2r; tw[1] = wk2i; tw[2] = wk2i; tw[3] = wk2r;
      tw[4] = wk2i; tw[5] = -wk2r; tw[6] = wk2r; tw[7] = -wk2i; *swap23 = 1;
    } else {
      tw[0] = wk3r; tw[1] = wk3i; tw[2] = wk1i; tw[3] = wk1r;
      tw[4] = wk1i; tw[5] = wk1r; tw[6] = wk3i; tw[7] = -wk3r;
    }
    return K_GENERIC;
  }
  {                                                         /* cftmdl2(8q, a, &w[nw - 8q]) */
    const float *W = w + nw - 8 * q;
    if (2 * c == q) {
      const float wk1r = W[2 * q], wk1i = W[2 * q + 1];
      tw[0] = wk1r; tw[1] = wk1i; tw[2] = wk1i; tw[3] = wk1r;
      tw[4] = wk1i; tw[5] = -wk1r; tw[6] = wk1r; tw[7] = -wk1i; *swap23 = 1;
      return K_GENERIC;
    }
    {
      const int cc = c < q - c ? c : q - c;
      const int kr = 4 * q - 4 * cc;
      const float wk1r = W[4 * cc], wk1i = W[4 * cc + 1], wk3r = W[4 * cc + 2], wk3i = W[4 * cc + 3];
      const float wd1i = W[kr], wd1r = W[kr + 1], wd3i = W[kr + 2], wd3r = W[kr + 3];
      if (cc == c) {
        tw[0] = wk1r; tw[1] = wk1i; tw[2] = wd1r; tw[3] = wd1i;
        tw[4] = wk3r; tw[5] = wk3i; tw[6] = wd3r; tw[7] = wd3i;
      } else {
        tw[0] = wd1i; tw[1] = wd1r; tw[2] = wk1i; tw[3] = wk1r;
        tw[4] = wd3i; tw[5] = wd3r; tw[6] = wk3i; tw[7] = wk3r;
      }
      return K_GENERIC;
    }
  }
}

/* the 8-point leaves, cftf081 :3048-3107 (type 1) and cftf082 :3110-3179 (type 2), on float pairs a[0..15] */
static void oo_leaf8_t1(float *a, float wn4r)
{
  float x0r, x0i, x1r, x1i, x2r, x2i, x3r, x3i;
  float y0r, y0i, y1r, y1i, y2r, y2i, y3r, y3i, y4r, y4i, y5r, y5i, y6r, y6i, y7r, y7i;
  x0r = a[0] + a[8]; x0i = a[1] + a[9];
  x1r = a[0] - a[8]; x1i = a[1] - a[9];
  x2r = a[4] + a[12]; x2i = a[5] + a[13];
  x3r = a[4] - a[12]; x3i = a[5] - a[13];
  y0r = x0r + x2r; y0i = x0i + x2i;
  y2r = x0r - x2r; y2i = x0i - x2i;
  y1r = x1r - x3i; y1i = x1i + x3r;
  y3r = x1r + x3i; y3i = x1i - x3r;
  x0r = a[2] + a[10]; x0i = a[3] + a[11];
  x1r = a[2] - a[10]; x1i = a[3] - a[11];
  x2r = a[6] + a[14]; x2i = a[7] + a[15];
  x3r = a[6] - a[14]; x3i = a[7] - a[15];
  y4r = x0r + x2r; y4i = x0i + x2i;
  y6r = x0r - x2r; y6i = x0i - x2i;
  x0r = x1r - x3i; x0i = x1i + x3r;
  x2r = x1r + x3i; x2i = x1i - x3r;
  y5r = wn4r * (x0r - x0i); y5i = wn4r * (x0r + x0i);
  y7r = wn4r * (x2r - x2i); y7i = wn4r * (x2r + x2i);
  a[8] = y1r + y5r; a[9] = y1i + y5i;
  a[10] = y1r - y5r; a[11] = y1i - y5i;
  a[12] = y3r - y7i; a[13] = y3i + y7r;
  a[14] = y3r + y7i; a[15] = y3i - y7r;
  a[0] = y0r + y4r; a[1] = y0i + y4i;
  a[2] = y0r - y4r; a[3] = y0i - y4i;
  a[4] = y2r - y6i; a[5] = y2i + y6r;
  a[6] = y2r + y6i; a[7] = y2i - y6r;
}

static void oo_leaf8_t2(float *a, float wn4r, float wk1r, float wk1i)
{
  float x0r, x0i, x1r, x1i;
  float y0r, y0i, y1r, y1i, y2r, y2i, y3r, y3i, y4r, y4i, y5r, y5i, y6r, y6i, y7r, y7i;
  y0r = a[0] - a[9]; y0i = a[1] + a[8];
  y1r = a[0] + a[9]; y1i = a[1] - a[8];
  x0r = a[4] - a[13]; x0i = a[5] + a[12];
  y2r = wn4r * (x0r - x0i); y2i = wn4r * (x0i + x0r);
  x0r = a[4] + a[13]; x0i = a[5] - a[12];
  y3r = wn4r * (x0r - x0i); y3i = wn4r * (x0i + x0r);
  x0r = a[2] - a[11]; x0i = a[3] + a[10];
  y4r = wk1r * x0r - wk1i * x0i; y4i = wk1r * x0i + wk1i * x0r;
  x0r = a[2] + a[11]; x0i = a[3] - a[10];
  y5r = wk1i * x0r - wk1r * x0i; y5i = wk1i * x0i + wk1r * x0r;
  x0r = a[6] - a[15]; x0i = a[7] + a[14];
  y6r = wk1i * x0r - wk1r * x0i; y6i = wk1i * x0i + wk1r * x0r;
  x0r = a[6] + a[15]; x0i = a[7] - a[14];
  y7r = wk1r * x0r - wk1i * x0i; y7i = wk1r * x0i + wk1i * x0r;
  x0r = y0r + y2r; x0i = y0i + y2i;
  x1r = y4r + y6r; x1i = y4i + y6i;
  a[0] = x0r + x1r; a[1] = x0i + x1i;
  a[2] = x0r - x1r; a[3] = x0i - x1i;
  x0r = y0r - y2r; x0i = y0i - y2i;
  x1r = y4r - y6r; x1i = y4i - y6i;
  a[4] = x0r - x1i; a[5] = x0i + x1r;
  a[6] = x0r + x1i; a[7] = x0i - x1r;
  x0r = y1r - y3i; x0i = y1i + y3r;
  x1r = y5r - y7r; x1i = y5i - y7i;
  a[8] = x0r + x1r; a[9] = x0i + x1i;
  a[10] = x0r - x1r; a[11] = x0i - x1i;
  x0r = y1r + y3i; x0i = y1i - y3r;
  x1r = y5r + y7r; x1i = y5i + y7i;
  a[12] = x0r - x1i; a[13] = x0i + x1r;
  a[14] = x0r + x1i; a[15] = x0i - x1r;
}

/* the butterfly levels of cftfsub (bwd = 0) / cftbsub (bwd = 1) for n > 32, without the final permutation */
static void oo_levels(const oo_plan *P, float *a, int bwd)
{
  const int nc = P->nc;
  const float wn4r = P->w[1];
  int level = 0, q;
  float tw[8];
  for (q = nc / 4; q >= 1; q >>= 2, level++) {
    const int nodes = nc / (4 * q);
    int node, c;
    if (q == 2) {                                            /* nc = 2 * 4^k: 8-point leaves */
      const float wk1r = P->w[P->nw - 8 + 2], wk1i = P->w[P->nw - 8 + 3];
      for (node = 0; node < nc / 8; node++) {
        if (oo_node_type((unsigned)node, level) == 1) oo_leaf8_t1(a + 16 * node, wn4r);
        else oo_leaf8_t2(a + 16 * node, wn4r, wk1r, wk1i);
      }
      return;
    }
    for (node = 0; node < nodes; node++) {
      const int type = oo_node_type((unsigned)node, level);
      float *b = a + 2 * (size_t)(4 * q) * node;
      int negA3 = 0;
      if (q == 1 && level > 0) {                             /* last level inside cftf161 / cftf162 */
        const int g = node & 3, ptype = oo_node_type((unsigned)node >> 2, level - 1);
        negA3 = (ptype == 1) ? (g == 3) : (g >= 2);
      }
      for (c = 0; c < q; c++) {
        int swap23;
        const int kind = oo_twiddle(P, level == 0, type, q, c, tw, &swap23);
        float *p0 = b + 2 * c, *p1 = p0 + 2 * q, *p2 = p1 + 2 * q, *p3 = p2 + 2 * q;
        if (type == 1) oo_bf1(p0, p1, p2, p3, kind, tw, wn4r, bwd && level == 0, negA3);
        else oo_bf2(p0, p1, p2, p3, kind, tw, wn4r, swap23, negA3);
      }
    }
  }
}

/* rdft(n, isgn, a, ip, w), fftsg.c:322-363, for n = 64 ... (power of two); in place on a[0..n) */
int lldo_ooura_rdft(int n, int isgn, float *a)
{
  const oo_plan *P;
  float *t;
  int j, m;
  if (n < 64 || (n & (n - 1))) return -1;
  P = oo_get_plan(n);
  m = n >> 1;
  t = (float *)malloc(sizeof(float) * (size_t)n);
  if (isgn >= 0) {
    oo_levels(P, a, 0);
    for (j = 0; j < P->nc; j++) { t[2 * P->rev[j]] = a[2 * j]; t[2 * P->rev[j] + 1] = a[2 * j + 1]; }   /* bitrv2 */
    memcpy(a, t, sizeof(float) * (size_t)n);
    for (j = 2; j < m; j += 2) {                              /* rftfsub :3241-3263 (ks = 1: nc = n / 4) */
      const int k = n - j, kk = j >> 1;
      const float wkr = (float)0.5 - P->c[n / 4 - kk], wki = P->c[kk];
      const float xr = a[j] - a[k], xi = a[j + 1] + a[k + 1];
      const float yr = wkr * xr - wki * xi, yi = wkr * xi + wki * xr;
      a[j] -= yr; a[j + 1] -= yi;
      a[k] += yr; a[k + 1] -= yi;
    }
    {
      const float xi = a[0] - a[1];
      a[0] += a[1];
      a[1] = xi;
    }
  } else {
    a[1] = (float)0.5 * (a[0] - a[1]);
    a[0] -= a[1];
    for (j = 2; j < m; j += 2) {                              /* rftbsub :3266-3288 */
      const int k = n - j, kk = j >> 1;
      const float wkr = (float)0.5 - P->c[n / 4 - kk], wki = P->c[kk];
      const float xr = a[j] - a[k], xi = a[j + 1] + a[k + 1];
      const float yr = wkr * xr + wki * xi, yi = wkr * xi - wki * xr;
      a[j] -= yr; a[j + 1] -= yi;
      a[k] += yr; a[k + 1] -= yi;
    }
    oo_levels(P, a, 1);
    for (j = 0; j < P->nc; j++) { t[2 * P->rev[j]] = a[2 * j]; t[2 * P->rev[j] + 1] = -a[2 * j + 1]; }  /* bitrv2conj */
    memcpy(a, t, sizeof(float) * (size_t)n);
  }
  free(t);
  return 0;
}

/* debug taps for the product's table tests: the makewt / makect tables of length n */
int lldo_ooura_tables(int n, float *w_out, float *c_out)
{
  const oo_plan *P;
  if (n < 64 || (n & (n - 1))) return -1;
  P = oo_get_plan(n);
  memcpy(w_out, P->w, sizeof(float) * (size_t)P->nw);
  memcpy(c_out, P->c, sizeof(float) * (size_t)(n / 4));
  return 0;
}
