// Functionals over each utterance's LLD matrix: cFunctionals in
// frameMode=full with cFunctionalExtremes / cFunctionalRegression (linear part) /
// cFunctionalMoments -- the option set of config/is09-13/IS09_emotion_core.func.conf.inc.
//   cFunctionals::doProcess          src/functionals/functionals.cpp:284-330
//   cFunctionalExtremes::process     src/functionals/functionalExtremes.cpp:92-134
//   cFunctionalRegression::process   src/functionals/functionalRegression.cpp:140-425
//   cFunctionalMoments::process      src/functionals/functionalMoments.cpp:88-165
// One workgroup per utterance: 32 columns x 8 row lanes, rows read coalesced (a row of 32
// floats = one 128-byte line). Two sweeps over the rows (the second one hits L2): sweep 1
// = min / max / first positions / sum / sum(i*x) in double, sweep 2 = central moments and
// the regression residual, which need the mean and the regression line. The accumulators are
// double like the reference's; partial sums are combined across the 8 row lanes in a fixed
// order (the reference sums sequentially: the result differs by double round-off only, i.e.
// almost never after the cast to float).
#include <hip/hip_runtime.h>

#include "lld_launch.hpp"
#include "lld_params.hpp"

namespace smilehip {

namespace {
constexpr int kCW = 32;      // columns per sweep
constexpr int kRL = 8;       // row lanes
}  // namespace

__global__ void __launch_bounds__(kCW * kRL) lld_functionals(FuncParams P) {
  const int u = blockIdx.x;
  const int c_in = threadIdx.x % kCW, r = threadIdx.x / kCW;
  const int64_t row0 = P.single_rows >= 0 ? 0 : P.row_off[u];
  const int64_t lld_rows = P.single_rows >= 0 ? P.single_rows : P.row_off[u + 1] - row0;
  // rows the reference's functionals see (see FuncParams::rows_cut)
  int64_t NN = lld_rows - (P.single_rows >= 0 ? 0 : P.rows_cut);
  if (NN < 1) NN = lld_rows > 0 ? 1 : 0;
  float *orow = P.out + (int64_t)u * P.ld_out;
  const int per = __popc(P.mask);
  if (NN <= 0) {                                        // no frame at all: the reference emits no vector; zeros here
    for (int i = threadIdx.x; i < per * P.n_cols; i += blockDim.x) orow[i] = 0.0f;
    return;
  }
  __shared__ double s_d[6][kRL][kCW];
  __shared__ float s_f[2][kRL][kCW];
  __shared__ int s_i[2][kRL][kCW];
  const float *x = P.x + row0 * P.ld_x;
  const double Nind = (double)NN;

  for (int c0 = 0; c0 < P.n_cols; c0 += kCW) {
    const int c = c0 + c_in;
    const bool on = c < P.n_cols;
    // ---- sweep 1
    float vmin = 0.f, vmax = 0.f;
    int pmin = 0x7fffffff, pmax = 0x7fffffff;
    double sum = 0.0, num = 0.0;
    if (on && r < NN) {
      vmin = vmax = x[(int64_t)r * P.ld_x + c];
      pmin = pmax = r;
    }
    if (on)
      for (int64_t t = r; t < NN; t += kRL) {
        const float v = x[t * P.ld_x + c];
        if (v < vmin) { vmin = v; pmin = (int)t; }
        if (v > vmax) { vmax = v; pmax = (int)t; }
        sum += (double)v;
        num += (double)v * (double)t;
      }
    s_f[0][r][c_in] = vmin; s_f[1][r][c_in] = vmax;
    s_i[0][r][c_in] = pmin; s_i[1][r][c_in] = pmax;
    s_d[0][r][c_in] = sum; s_d[1][r][c_in] = num;
    __syncthreads();
    // every thread of a column recomputes the column's totals (same order, same result)
    float mn = s_f[0][0][c_in], mx = s_f[1][0][c_in];
    int pmn = s_i[0][0][c_in], pmx = s_i[1][0][c_in];
    sum = s_d[0][0][c_in]; num = s_d[1][0][c_in];
    for (int k = 1; k < kRL; ++k) {
      if (s_i[0][k][c_in] != 0x7fffffff) {             // lane k saw at least one row
        const float a = s_f[0][k][c_in], b = s_f[1][k][c_in];
        const int pa = s_i[0][k][c_in], pb = s_i[1][k][c_in];
        if (a < mn || (a == mn && pa < pmn)) { mn = a; pmn = pa; }
        if (b > mx || (b == mx && pb < pmx)) { mx = b; pmx = pb; }
      }
      sum += s_d[0][k][c_in];
      num += s_d[1][k][c_in];
    }
    const double mean = sum / Nind;                      // functionals.cpp:312-318
    const float meanf = (float)mean;
    // linear regression (functionalRegression.cpp:218-246)
    double m = 0.0, tt = 0.0;
    const double asum = (double)meanf * Nind;
    if (NN > 1) {
      const double NNm1 = (Nind) * (Nind - 1.0);
      const double S1 = NNm1 / 2.0;
      const double S2 = NNm1 * (2.0 * Nind - 1.0) / 6.0;
      const double S1dS2 = S1 / S2;
      const double d = (Nind - S1 * S1dS2);
      if (d == 0.0) tt = 0.0;
      else tt = (asum - num * S1dS2) / d;
      m = (num - tt * S1) / S2;
    } else {
      m = 0.0;
      tt = on ? (double)x[c] : 0.0;
    }
    __syncthreads();
    // ---- sweep 2
    double m2 = 0.0, m3 = 0.0, m4 = 0.0, leq = 0.0, lea = 0.0;
    const double meanD = (double)meanf;
    if (on)
      for (int64_t t = r; t < NN; t += kRL) {
        const double v = (double)x[t * P.ld_x + c];
        const double tmp = v - meanD;
        double tmp2 = tmp * tmp;
        m2 += tmp2;
        tmp2 *= tmp;
        m3 += tmp2;
        m4 += tmp2 * tmp;
        const double e = v - (m * (double)t + tt);
        lea += fabs(e);
        leq += e * e;
      }
    s_d[0][r][c_in] = m2; s_d[1][r][c_in] = m3; s_d[2][r][c_in] = m4; s_d[3][r][c_in] = leq; s_d[4][r][c_in] = lea;
    __syncthreads();
    if (r == 0 && on) {
      m2 = m3 = m4 = leq = lea = 0.0;
      for (int k = 0; k < kRL; ++k) {
        m2 += s_d[0][k][c_in]; m3 += s_d[1][k][c_in]; m4 += s_d[2][k][c_in];
        leq += s_d[3][k][c_in]; lea += s_d[4][k][c_in];
      }
      float *o = orow + (int64_t)c * per;
      int n = 0;
      const uint32_t mask = P.mask;
      if (mask & SMILEHIP_FUNC_MAX) o[n++] = mx;
      if (mask & SMILEHIP_FUNC_MIN) o[n++] = mn;
      if (mask & SMILEHIP_FUNC_RANGE) o[n++] = mx - mn;
      if (mask & SMILEHIP_FUNC_MAXPOS) o[n++] = (float)pmx;
      if (mask & SMILEHIP_FUNC_MINPOS) o[n++] = (float)pmn;
      if (mask & SMILEHIP_FUNC_AMEAN) o[n++] = meanf;
      if (mask & SMILEHIP_FUNC_MAXAMEANDIST) o[n++] = mx - meanf;
      if (mask & SMILEHIP_FUNC_MINAMEANDIST) o[n++] = meanf - mn;
      if (!isfinite(m)) m = 0.0;
      if (!isfinite(tt)) tt = 0.0;
      if (!isfinite(lea / Nind)) lea = 0.0;
      if (!isfinite(leq / Nind)) leq = 0.0;
      if (mask & SMILEHIP_FUNC_LINREGC1) o[n++] = (float)m;
      if (mask & SMILEHIP_FUNC_LINREGC2) o[n++] = (float)tt;
      if (mask & SMILEHIP_FUNC_LINREGERRA) o[n++] = (float)(lea / Nind);
      if (mask & SMILEHIP_FUNC_LINREGERRQ) o[n++] = (float)(leq / Nind);
      m2 /= Nind;
      const double sqm2 = sqrt(m2);
      if (mask & SMILEHIP_FUNC_VARIANCE) o[n++] = (float)m2;
      if (mask & SMILEHIP_FUNC_STDDEV) o[n++] = (m2 > 0.0) ? (float)sqm2 : 0.0f;
      if (mask & SMILEHIP_FUNC_SKEWNESS) o[n++] = (m2 > 0.0) ? (float)(m3 / (Nind * m2 * sqm2)) : 0.0f;
      if (mask & SMILEHIP_FUNC_KURTOSIS) o[n++] = (m2 > 0.0) ? (float)(m4 / (Nind * m2 * m2)) : 0.0f;
      if (mask & SMILEHIP_FUNC_AMEAN_M) o[n++] = meanf;
    }
    __syncthreads();
  }
}

hipError_t launch_functionals(const FuncParams &P, int n_utt, hipStream_t s) {
  if (n_utt <= 0) return hipSuccess;
  hipLaunchKernelGGL(lld_functionals, dim3((unsigned)n_utt), dim3(kCW * kRL), 0, s, P);
  return hipGetLastError();
}

}  // namespace smilehip
