// General functionals: any cFunctionals instance -- the nine families the ComParE_2016 set
// uses -- over the columns of LLD matrices.
//   cFunctionals::doProcess          src/functionals/functionals.cpp:320-389
//   cFunctionalExtremes / Means / Moments / Regression / Percentiles / Times / Segments / Lpc / Peaks2 ::process
//                                    src/functionals/functional*.cpp (line ranges in include/smilehip.h)
// Mapping: ONE THREAD PER (utterance, column). Most of these functionals are order-sensitive scans (double sums in
// index order, state machines over the contour, a linked list of extrema that is pruned in several passes), so a
// column is walked sequentially by its thread exactly as the reference walks it -- results agree bit for bit up to
// libm (log / exp) -- and parallelism comes from the 10^5 (utterance, column) pairs of a batch. A wave holds 64
// neighbouring columns of one utterance, so every step of a scan is one coalesced row segment (256 B) that stays in
// L2 across the passes. The only stage with a different shape is the sort Percentiles needs: one workgroup per
// (utterance, column), bitonic network in LDS.
//   fs_stats      nonZeroFuncts compaction, min / max / mean, N     (functionals.cpp:326-363)
//   fs_family<F>  one launch per enabled family
//   fs_percentiles sort + percentile read-out
// Peaks2's list of local extrema is not materialised: an element is identified by its row index and only an
// "alive" byte per row is kept; each pruning pass re-derives the extrema from three neighbouring samples.
#include <hip/hip_runtime.h>
#include "kernel_timing.hpp"

#include <cfloat>
#include <cmath>

#include "lld_launch.hpp"
#include "lld_params.hpp"
#include "lld_ooura.hpp"

namespace smilehip {

namespace {

constexpr int kColsPerBlock = 64;
constexpr int kSortThreads = 256;
constexpr int kSortLds = 8192;
constexpr int kWaveSortMaxDecl = 1024;   // contours up to this length are sorted by one wave (fs_percentiles_wave)

#define FS_BIT(m, i) (((m) >> (i)) & 1u)

struct Col {                       // one thread's view of its column
  const float *p;                  // element t at p[t * ld]
  int64_t ld;
  int64_t n_main;                  // rows served by p; row n_main (if N > n_main) comes from *pe
  const float *pe;
  int64_t N;
  __device__ __forceinline__ float operator[](int64_t t) const { return t < n_main ? p[t * ld] : *pe; }
};

// Sequential walk over rows [a, b) of a column with the loads batched eight ahead: a scan is a chain of dependent
// steps, and one wave per SIMD cannot hide a memory round trip per step -- eight independent loads are in flight
// before the first value is consumed. f(t, value) is called in index order.
template <typename F>
__device__ __forceinline__ void for_rows(const Col &in, int64_t a, int64_t b, F f) {
  int64_t t = a;
  for (; t + 8 <= b; t += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = in[t + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) f(t + k, v[k]);
  }
  for (; t < b; ++t) f(t, in[t]);
}

struct Where {                     // what a thread works on
  int u, c;                        // utterance, column inside the instance
  int64_t srow0;                   // first scratch row of the utterance
  int64_t rows;                    // rows before nonZeroFuncts
  bool on;
};

__device__ __forceinline__ void utt_rows(const FsParams &P, Where &w) {
  if (P.single_rows >= 0) {
    w.srow0 = 0;
    w.rows = P.single_rows;
    return;
  }
  const int64_t r0 = P.row_off[w.u], lld = P.row_off[w.u + 1] - r0;
  int64_t n = lld - P.rows_cut;
  if (P.pending) {
    const int64_t p = P.pending[w.u];
    if (p < lld - 1) n -= p;
  }
  if (n < 1) n = lld > 0 ? 1 : 0;
  if (n > 0 && P.extra) n += 1;
  w.srow0 = r0 + w.u;
  w.rows = n;
}

__device__ __forceinline__ Where locate(const FsParams &P) {
  Where w;
  const int groups = (P.n_cols + kColsPerBlock - 1) / kColsPerBlock;
  w.u = blockIdx.x / groups;
  w.c = (blockIdx.x % groups) * kColsPerBlock + threadIdx.x;
  w.on = w.c < P.n_cols;
  utt_rows(P, w);
  return w;
}

__device__ __forceinline__ Col raw_col(const FsParams &P, const Where &w) {
  Col x;
  const int64_t r0 = P.single_rows >= 0 ? 0 : P.row_off[w.u];
  x.p = P.x + r0 * P.ld_x + P.col_first + w.c;
  x.ld = P.ld_x;
  x.N = w.rows;
  x.n_main = (P.single_rows < 0 && P.extra) ? w.rows - 1 : w.rows;
  x.pe = P.extra ? P.extra + (int64_t)w.u * P.ld_extra + w.c : x.p;
  return x;
}

// the column the families see: the compacted copy if nonZeroFuncts, else the input itself
__device__ __forceinline__ Col data_col(const FsParams &P, const Where &w) {
  if (!P.spec.non_zero_functs) return raw_col(P, w);
  Col x;
  x.p = P.nz + w.srow0 * P.n_cols + w.c;
  x.ld = P.n_cols;
  x.N = x.n_main = P.st_n[(int64_t)w.u * P.n_cols + w.c];
  x.pe = x.p;
  return x;
}

// smileMath_ratioLimit and friends (smileUtil.c:586-613), float / double promotions as written there
__device__ float fs_logistic(float x) {
  const float lim = (float)log((double)FLT_MAX);
  if (x > lim) return 1.0f;
  else if (x < -lim) return 0.0f;
  return (float)(1.0 / (1.0 + exp(-(double)x)));
}
__device__ float fs_tanh(float x) { return 2.0f * fs_logistic(2.0f * x) - 1.0f; }
__device__ float fs_ratio_limit(float x, float limit1, float excess) {
  if (x > limit1) {
    return fs_tanh((float)((sqrt((double)(x - limit1) + 1.0) - 1.0) / ((double)excess * 0.5))) * excess + limit1;
  } else if (x < -limit1) {
    return fs_tanh((float)(-(sqrt(-1.0 * (double)(x + limit1) + 1.0) - 1.0) / ((double)excess * 0.5))) * excess - limit1;
  }
  return x;
}

}  // namespace

// ------------------------------------------------------------------ stats
__global__ void __launch_bounds__(kColsPerBlock) fs_stats(FsParams P) {
  const Where w = locate(P);
  if (!w.on) return;
  const Col r = raw_col(P, w);
  const int64_t si = (int64_t)w.u * P.n_cols + w.c;
  int64_t NN = r.N;
  if (P.spec.non_zero_functs) {
    float *d = P.nz + w.srow0 * P.n_cols + w.c;
    NN = 0;
    const bool pos = P.spec.non_zero_functs == 2;
    for_rows(r, 0, r.N, [&](int64_t, float v) {
      if (pos ? (v > 0.0f) : (v != 0.0f)) d[(NN++) * P.n_cols] = v;
    });
  }
  P.st_n[si] = (int32_t)NN;
  if (NN <= 0) {
    P.st_min[si] = P.st_max[si] = P.st_mean[si] = 0.0f;
    float *o = P.out + (int64_t)w.u * P.ld_out + (int64_t)w.c * P.per;      // every family yields nothing: zeros
    for (int i = 0; i < P.per; ++i) o[i] = 0.0f;
    return;
  }
  const Col x = data_col(P, w);
  float mn = x[0], mx = mn;
  double mean = mn;
  for_rows(x, 1, NN, [&](int64_t, float v) {
    if (v < mn) mn = v;
    if (v > mx) mx = v;
    mean += (double)v;
  });
  mean /= (double)NN;
  P.st_min[si] = mn; P.st_max[si] = mx; P.st_mean[si] = (float)mean;
}

// ------------------------------------------------------------------ families
namespace {

__device__ int f_extremes(const smilehip_func_spec &s, const Col &in, float min, float max, float mean, float *out) {
  const int64_t Nin = in.N;
  int64_t minpos = -1, maxpos = -1;
  for_rows(in, 0, Nin, [&](int64_t i, float v) {
    if ((v == max) && (maxpos == -1)) maxpos = i;
    if ((v == min) && (minpos == -1)) minpos = i;
  });
  float maxposD = (float)maxpos, minposD = (float)minpos;
  if (s.ext_norm == SMILEHIP_NORM_SEGMENT) {
    maxposD /= (float)Nin;
    minposD /= (float)Nin;
  } else if (s.ext_norm == SMILEHIP_NORM_SECOND) {
    const float T = (float)s.period;
    if (T != 0.0f) { maxposD *= T; minposD *= T; }
  }
  const uint32_t m = s.ext_mask;
  int n = 0;
  if (FS_BIT(m, 0)) out[n++] = max;
  if (FS_BIT(m, 1)) out[n++] = min;
  if (FS_BIT(m, 2)) out[n++] = max - min;
  if (FS_BIT(m, 3)) out[n++] = maxposD;
  if (FS_BIT(m, 4)) out[n++] = minposD;
  if (FS_BIT(m, 5)) out[n++] = mean;
  if (FS_BIT(m, 6)) out[n++] = max - mean;
  if (FS_BIT(m, 7)) out[n++] = mean - min;
  return n;
}

__device__ int f_means(const smilehip_func_spec &s, const Col &in, float mean, float *out) {
  const int64_t Nin = in.N;
  const uint32_t m = s.means_mask;
  const bool need_log = (m & ((1u << 6) | (1u << 8))) != 0;      // nzgmean / flatness
  double tmp = (double)in[0];
  double fa = fabs(tmp);
  double absmean = fa, qmean = tmp * tmp;
  int64_t nnz;
  double nzamean, nzabsmean, nzqmean, nzgmean;
  double posamean = 0.0, negamean = 0.0, posqmean = 0.0, negqmean = 0.0;
  int64_t nPos = 0, nNeg = 0;
  if (tmp != 0.0) {
    nzamean = tmp; nzabsmean = fa; nzqmean = tmp * tmp; nzgmean = need_log ? log(fa) : 0.0; nnz = 1;
    if (tmp > 0) { posamean += tmp; posqmean += tmp * tmp; nPos++; }
    else { negamean += tmp; negqmean += tmp * tmp; nNeg++; }
  } else {
    nzamean = nzabsmean = nzqmean = nzgmean = 0.0; nnz = 0;
  }
  for_rows(in, 1, Nin, [&](int64_t, float vin) {
    tmp = (double)vin;
    fa = fabs(tmp);
    absmean += fa;
    if (tmp > 0) { posamean += tmp; nPos++; }
    if (tmp < 0) { negamean += tmp; nNeg++; }
    const double t0 = tmp;
    if (tmp != 0.0) {
      nzamean += tmp;
      nzabsmean += fa;
      if (need_log) nzgmean += log(fa);
      tmp *= tmp;
      nzqmean += tmp;
      nnz++;
      if (t0 > 0) posqmean += tmp;
      if (t0 < 0) negqmean += tmp;
      qmean += tmp;
    }
  });
  tmp = (double)Nin;
  absmean = absmean / tmp;
  qmean = qmean / tmp;
  if (nnz > 0) {
    tmp = (double)nnz;
    nzamean /= tmp; nzabsmean /= tmp; nzqmean /= tmp; nzgmean /= tmp;
    nzgmean = exp(nzgmean);
  }
  if (nPos > 0) { posamean /= (double)nPos; posqmean /= (double)nPos; }
  if (nNeg > 0) { negamean /= (double)nNeg; negqmean /= (double)nNeg; }
  int n = 0;
  if (FS_BIT(m, 0)) out[n++] = (float)mean;
  if (FS_BIT(m, 1)) out[n++] = (float)absmean;
  if (FS_BIT(m, 2)) out[n++] = (float)qmean;
  if (FS_BIT(m, 3)) out[n++] = (float)nzamean;
  if (FS_BIT(m, 4)) out[n++] = (float)nzabsmean;
  if (FS_BIT(m, 5)) out[n++] = (float)nzqmean;
  if (FS_BIT(m, 6)) out[n++] = (float)nzgmean;
  if (FS_BIT(m, 7)) {
    if (s.means_norm == SMILEHIP_NORM_FRAME) out[n++] = (float)nnz;
    else if (s.means_norm == SMILEHIP_NORM_SEGMENT) out[n++] = (float)nnz / (float)Nin;
    else out[n++] = (float)nnz / (float)s.period;
  }
  if (FS_BIT(m, 8)) out[n++] = (absmean != 0.0) ? (float)(nzgmean / absmean) : 1.0f;
  if (FS_BIT(m, 9)) out[n++] = (float)posamean;
  if (FS_BIT(m, 10)) out[n++] = (float)negamean;
  if (FS_BIT(m, 11)) out[n++] = (float)posqmean;
  if (FS_BIT(m, 12)) out[n++] = (float)sqrt(posqmean);
  if (FS_BIT(m, 13)) out[n++] = (float)negqmean;
  if (FS_BIT(m, 14)) out[n++] = (float)sqrt(negqmean);
  if (FS_BIT(m, 15)) out[n++] = (float)sqrt(qmean);
  if (FS_BIT(m, 16)) out[n++] = (float)sqrt(nzqmean);
  return n;
}

__device__ int f_moments(const smilehip_func_spec &s, const Col &in, float mean, float *out) {
  const int64_t Nin = in.N;
  double m2 = 0.0, m3 = 0.0, m4 = 0.0;
  const double Nind = (double)Nin, meanD = (double)mean;
  for_rows(in, 0, Nin, [&](int64_t, float v) {
    const double tmp = ((double)v - meanD);
    double tmp2 = tmp * tmp;
    m2 += tmp2;
    tmp2 *= tmp;
    m3 += tmp2;
    m4 += tmp2 * tmp;
  });
  m2 /= Nind;
  const uint32_t m = s.mom_mask;
  int n = 0;
  if (FS_BIT(m, 0)) out[n++] = (float)m2;
  const double sqm2 = sqrt(m2);
  if (FS_BIT(m, 1)) out[n++] = (m2 > 0.0) ? (float)sqm2 : 0.0f;
  if (FS_BIT(m, 2)) out[n++] = (m2 > 0.0) ? (float)(m3 / (Nind * m2 * sqm2)) : 0.0f;
  if (FS_BIT(m, 3)) out[n++] = (m2 > 0.0) ? (float)(m4 / (Nind * m2 * m2)) : 0.0f;
  if (FS_BIT(m, 4)) out[n++] = (float)mean;
  if (FS_BIT(m, 5)) {
    if (m2 > 0.0) {
      const float meanLocal = (s.mom_stddev_norm == 1) ? fabsf(mean) : mean;
      if (s.mom_ratio_limit) {
        if (meanLocal != 0.0f) out[n++] = fs_ratio_limit((float)(sqm2 / (double)meanLocal), 10.0f, 20.0f);
        else out[n++] = 20.0f;
      } else {
        double mean1 = (double)meanLocal;
        if (mean1 == 0.0) mean1 = 1.0;
        out[n++] = (float)(sqm2 / mean1);
      }
    } else out[n++] = 0.0f;
  }
  return n;
}

__device__ int f_regression(const smilehip_func_spec &s, const Col &in, float min, float max, float mean, float *out) {
  const int64_t Nin = in.N;
  const double Nind = (double)Nin;
  double range = (double)(max - min), rangeInv;
  if (range <= 0.0) { range = 1.0; rangeInv = 0.0; } else rangeInv = 1.0 / range;
  const bool enQreg = (s.reg_mask & 0x3fff0u) != 0;
  double num = 0.0, numAbs = 0.0, num2 = 0.0, tmp = 0.0, ii = 0.0, asumAbs = 0.0;
  const double asum = (double)mean * Nind;
  for_rows(in, 0, Nin, [&](int64_t, float v) {
    if (s.reg_centroid_abs) {
      asumAbs += (double)fabsf(v);
      numAbs += (double)fabsf(v) * ii;
    }
    tmp = (double)v * ii;
    num += tmp;
    tmp *= ii;
    ii += 1.0;
    num2 += tmp;
  });
  double centroid;
  if (s.reg_centroid_abs) centroid = (asumAbs != 0.0) ? numAbs / asumAbs : 0.0;
  else centroid = (asum != 0.0) ? num / asum : 0.0;
  if (s.reg_centroid_limit) centroid = (double)fs_ratio_limit((float)centroid, (float)Nind, (float)Nind);
  if (s.reg_centroid_norm == SMILEHIP_NORM_SECOND) centroid *= s.period;
  else if (s.reg_centroid_norm == SMILEHIP_NORM_SEGMENT) centroid /= Nind;

  double m = 0.0, t = 0.0, leq = 0.0, lea = 0.0, a = 0.0, b = 0.0, c = 0.0, qeq = 0.0, qea = 0.0;
  if (Nin > 1) {
    const double NNm1 = (Nind) * (Nind - 1.0);
    const double S1 = NNm1 / 2.0;
    const double S2 = NNm1 * (2.0 * Nind - 1.0) / 6.0;
    const double S1dS2 = S1 / S2;
    const double d = (Nind - S1 * S1dS2);
    if (d == 0.0) t = 0.0; else t = (asum - num * S1dS2) / d;
    m = (num - t * S1) / S2;
    const double S3 = S1 * S1;
    const double Nind1 = Nind - 1.0;
    const double S4 = S2 * (3.0 * (Nind1 * Nind1 + Nind1) - 1.0) / 5.0;
    if (enQreg) {
      const double S3S3 = S3 * S3, S2S2 = S2 * S2, S1S2 = S1 * S2, S1S1 = S3;
      const double det = S4 * S2 * Nind + 2.0 * S3 * S1S2 - S2S2 * S2 - S3S3 * Nind - S1S1 * S4;
      if (det != 0.0) {
        a = ((S2 * Nind - S1S1) * num2 + (S1S2 - S3 * Nind) * num + (S3 * S1 - S2S2) * asum) / det;
        b = ((S1S2 - S3 * Nind) * num2 + (S4 * Nind - S2S2) * num + (S3 * S2 - S4 * S1) * asum) / det;
        c = ((S3 * S1 - S2S2) * num2 + (S3 * S2 - S4 * S1) * num + (S4 * S2 - S3S3) * asum) / det;
      }
    }
  } else {
    m = 0; t = c = (double)in[0];
  }
  ii = 0.0;
  for_rows(in, 0, Nin, [&](int64_t, float vin) {   // both residual sweeps share one walk; each sum keeps its own order
    const double v = (double)vin;
    double e = v - (m * ii + t);
    if (s.reg_norm_inputs) e *= rangeInv;
    lea += fabs(e);
    leq += e * e;
    if (enQreg) {
      double q = v - (a * ii * ii + b * ii + c);
      if (s.reg_norm_inputs) q *= rangeInv;
      qea += fabs(q);
      qeq += q * q;
    }
    ii += 1.0;
  });
  double rs = 0.0, ls = 0.0, x0 = 0.0, y0 = 0.0, yr = 0.0, yrnn = 0.0, c3nn = 0.0, y0nn = 0.0;
  if (enQreg) {
    x0 = b / (-2.0 * a);
    if (x0 < -1.0 * Nind) x0 = -Nind;
    if (x0 > Nind) x0 = Nind;
    if (!isfinite(x0)) x0 = Nind;
    y0 = c - b * b / (4.0 * a);
    if (!isfinite(y0)) y0 = 0.0;
    y0nn = y0;
    yrnn = yr = a * (Nind - 1.0) * (Nind - 1.0) + b * (Nind - 1.0) + c;
    if (!isfinite(yr)) { yr = 0.0; yrnn = 0.0; }
    c3nn = c;
  }
  double NOneSec = 1.0;
  if (s.reg_norm_coeff == 2) NOneSec = 1.0 / s.period;
  if (s.reg_ratio_limit) {
    m = fs_ratio_limit((float)m, (float)(range / 10.0), (float)(range / 10.0 + 0.01));
    a = fs_ratio_limit((float)a, (float)(sqrt(range / 10.0)), (float)(sqrt(range / 10.0) + 0.01));
    b = fs_ratio_limit((float)b, (float)(range / 10.0), (float)(range / 10.0 + 0.01));
  }
  if (s.reg_norm_coeff == 1) {
    m *= Nind - 1.0;
    a *= (Nind - 1.0) * (Nind - 1.0);
    b *= Nind - 1.0;
    if (Nind != 1.0) x0 /= Nind - 1.0; else x0 = 0.0;
  } else if (s.reg_norm_coeff == 2) {
    m *= NOneSec;
    a *= NOneSec * NOneSec;
    b *= NOneSec;
    if (NOneSec != 1.0) x0 /= NOneSec; else x0 = 0.0;
  }
  if (s.reg_norm_inputs) {
    m *= rangeInv;
    t = (t - (double)min) * rangeInv;
    a *= rangeInv;
    b *= rangeInv;
    c = (c - (double)min) * rangeInv;
    y0 = (y0 - (double)min) * rangeInv;
    yr = (yr - (double)min) * rangeInv;
  }
  if (enQreg) {
    if (x0 > 0) ls = (y0 - c) / x0;
    if (s.reg_norm_coeff == 1) {
      if (x0 < 1.0) rs = (yr - y0) / (1.0 - x0);
    } else if (s.reg_norm_coeff == 2) {
      const double len_t = (Nind - 1.0) / NOneSec;
      if (x0 < len_t) rs = (yr - y0) / (len_t - x0);
    } else {
      if (x0 < Nind - 1.0) rs = (yr - y0) / (Nind - 1.0 - x0);
    }
  }
  if (!isfinite(m)) m = 0.0;
  if (!isfinite(t)) t = 0.0;
  if (!isfinite(lea / Nind)) lea = 0.0;
  if (!isfinite(leq / Nind)) leq = 0.0;
  if (!isfinite(a)) a = 0.0;
  if (!isfinite(b)) b = 0.0;
  if (!isfinite(c)) { c = 0.0; c3nn = 0.0; }
  if (!isfinite(ls)) ls = 0.0;
  if (!isfinite(rs)) rs = 0.0;
  if (!isfinite(qea / Nind)) qea = 0.0;
  if (!isfinite(qeq / Nind)) qeq = 0.0;
  if (!isfinite(centroid)) centroid = 0.0;
  const uint32_t k = s.reg_mask;
  int n = 0;
  if (FS_BIT(k, 0)) out[n++] = (float)m;
  if (FS_BIT(k, 1)) out[n++] = (float)t;
  if (FS_BIT(k, 2)) out[n++] = (float)(lea / Nind);
  if (FS_BIT(k, 3)) out[n++] = (float)(leq / Nind);
  if (FS_BIT(k, 4)) out[n++] = (float)a;
  if (FS_BIT(k, 5)) out[n++] = (float)b;
  if (FS_BIT(k, 6)) out[n++] = (float)c;
  if (!s.reg_old_buggy_qerr) {
    if (FS_BIT(k, 7)) out[n++] = (float)(qea / Nind);
    if (FS_BIT(k, 8)) out[n++] = (float)(qeq / Nind);
  } else {
    if (FS_BIT(k, 7)) out[n++] = (float)(qea);
    if (FS_BIT(k, 8)) out[n++] = (float)(qeq);
  }
  if (FS_BIT(k, 9)) out[n++] = (float)centroid;
  if (FS_BIT(k, 10)) out[n++] = (float)ls;
  if (FS_BIT(k, 11)) out[n++] = (float)rs;
  if (FS_BIT(k, 12)) out[n++] = (float)x0;
  if (FS_BIT(k, 13)) out[n++] = (float)y0;
  if (FS_BIT(k, 14)) out[n++] = (float)yr;
  if (FS_BIT(k, 15)) out[n++] = (float)y0nn;
  if (FS_BIT(k, 16)) out[n++] = (float)c3nn;
  if (FS_BIT(k, 17)) out[n++] = (float)yrnn;
  return n;
}

__device__ int f_times(const smilehip_func_spec &s, const Col &in, float min, float max, float *out) {
  const int64_t Nin = in.N;
  const float Nind = (float)Nin;
  float Norm = Nind, Norm1 = Nind - 1.0f, Norm2 = Nind - 2.0f;
  float T = 1.0f;
  if (s.times_norm == SMILEHIP_NORM_SECOND) {
    T = (float)s.period;
    if (T != 0.0f) {
      if (s.times_buggy_sec_norm) { Norm /= T; Norm1 /= T; Norm2 /= T; }
      else { Norm = 1.0f / T; Norm1 /= Nind * T; Norm2 /= Nind * T; }
    }
  }
  if (s.times_norm == SMILEHIP_NORM_FRAME) { Norm = 1.0f; Norm1 /= Nind; Norm2 /= Nind; }
  const float range = max - min;
  const float l25 = 0.25f * range + min, l50 = 0.50f * range + min, l75 = 0.75f * range + min, l90 = 0.90f * range + min;
  int64_t n25 = 0, n50 = 0, n75 = 0, n90 = 0, nR = 0, nF = 0, nLC = 0, nRC = 0;
  // all counts are integers: one fused walk; sample i is classified when sample i+1 arrives (curvature looks ahead)
  float pm = 0.0f, p0 = in[0];
  auto step = [&](int64_t i, float pn, bool has_next) {
    if (p0 <= l25) n25++;
    if (p0 <= l50) n50++;
    if (p0 <= l75) n75++;
    if (p0 <= l90) n90++;
    if (i >= 1) {
      if (pm < p0) nR++;
      else if (pm > p0) nF++;
      if (has_next) {
        const float a1 = p0 - pm, a2 = pn - p0;
        if (a2 < a1) nRC++;
        else if (a1 < a2) nLC++;
      }
    }
    pm = p0; p0 = pn;
  };
  for_rows(in, 1, Nin, [&](int64_t j, float v) { step(j - 1, v, true); });
  step(Nin - 1, 0.0f, false);
  const uint32_t m = s.times_mask;
  int n = 0;
  if (FS_BIT(m, 0)) out[n++] = ((float)(Nin - n25)) / Norm;
  if (FS_BIT(m, 1)) out[n++] = ((float)(n25)) / Norm;
  if (FS_BIT(m, 2)) out[n++] = ((float)(Nin - n50)) / Norm;
  if (FS_BIT(m, 3)) out[n++] = ((float)(n50)) / Norm;
  if (FS_BIT(m, 4)) out[n++] = ((float)(Nin - n75)) / Norm;
  if (FS_BIT(m, 5)) out[n++] = ((float)(n75)) / Norm;
  if (FS_BIT(m, 6)) out[n++] = ((float)(Nin - n90)) / Norm;
  if (FS_BIT(m, 7)) out[n++] = ((float)(n90)) / Norm;
  if (FS_BIT(m, 8)) out[n++] = (Norm1 != 0.0f) ? ((float)nR) / Norm1 : 0.0f;
  if (FS_BIT(m, 9)) out[n++] = (Norm1 != 0.0f) ? ((float)nF) / Norm1 : 0.0f;
  if (FS_BIT(m, 10)) out[n++] = (Norm2 != 0.0f) ? ((float)nLC) / Norm2 : 0.0f;
  if (FS_BIT(m, 11)) out[n++] = (Norm2 != 0.0f) ? ((float)nRC) / Norm2 : 0.0f;
  if (FS_BIT(m, 12)) out[n++] = (s.times_norm == SMILEHIP_NORM_SECOND) ? ((float)(Nin) * T) : (float)Nin;
  // second pass, user defined times (:347-364): one walk for all levels
  if (s.n_ul + s.n_dl > 0) {
    float lu[8], ld[8];
    int64_t cu[8], cd[8];
    for (int j = 0; j < 8; ++j) {
      lu[j] = (j < s.n_ul) ? (float)(s.ul[j] * (double)range + (double)min) : 0.0f;
      ld[j] = (j < s.n_dl) ? (float)(s.dl[j] * (double)range + (double)min) : 0.0f;
      cu[j] = cd[j] = 0;
    }
    for_rows(in, 0, Nin, [&](int64_t, float v) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < s.n_ul && v > lu[j]) cu[j]++;
        if (j < s.n_dl && v <= ld[j]) cd[j]++;
      }
    });
    for (int j = 0; j < s.n_ul; ++j) out[n++] = ((float)cu[j]) / Norm;
    for (int j = 0; j < s.n_dl; ++j) out[n++] = ((float)cd[j]) / Norm;
  }
  return n;
}

// Onset (functionalOnset.cpp:83-151): a two-state walk over the contour -- an onset when the value rises above thresholdOnset
// from the "off" state, an offset when it is at or below thresholdOffset in the "on" state (both can fire on one sample)
__device__ int f_onset(const smilehip_func_spec &s, const Col &in, float *out) {
  const int64_t Nin = in.N;
  int64_t onsetPos = -1, offsetPos = -1, nOnsets = 0, nOffsets = 0;
  int oo = (in[0] > s.ons_thr_on) ? 1 : 0;                       // the first sample is compared as it is, also with useAbsVal
  const bool use_abs = s.ons_use_abs != 0;
  const float th_on = s.ons_thr_on, th_off = s.ons_thr_off;
  for_rows(in, 1, Nin, [&](int64_t i, float v) {
    const float cur = use_abs ? fabsf(v) : v;
    if (cur > th_on) {
      if (oo == 0) { nOnsets++; if (onsetPos == -1) onsetPos = i; oo = 1; }
    }
    if (cur <= th_off) {
      if (oo == 1) { nOffsets++; offsetPos = i; oo = 0; }
    }
  });
  if (offsetPos == -1) offsetPos = Nin - 1;
  if (onsetPos == -1) onsetPos = 0;
  const uint32_t m = s.ons_mask;
  int n = 0;
  if (s.ons_norm == SMILEHIP_NORM_SEGMENT) {
    if (FS_BIT(m, 0)) out[n++] = (float)onsetPos / (float)(Nin);
    if (FS_BIT(m, 1)) out[n++] = (float)offsetPos / (float)(Nin);
  } else if (s.ons_norm == SMILEHIP_NORM_SECOND) {
    const float T = (float)s.period;
    if (FS_BIT(m, 0)) out[n++] = (float)onsetPos * T;
    if (FS_BIT(m, 1)) out[n++] = (float)offsetPos * T;
  } else {
    if (FS_BIT(m, 0)) out[n++] = (float)onsetPos;
    if (FS_BIT(m, 1)) out[n++] = (float)offsetPos;
  }
  if (FS_BIT(m, 2)) out[n++] = (float)nOnsets;
  if (FS_BIT(m, 3)) out[n++] = (float)nOffsets;
  if (FS_BIT(m, 4)) { const float T = (float)s.period; out[n++] = (float)nOnsets / ((float)Nin * T); }
  return n;
}

// Peaks (functionalPeaks.cpp:98-214, overlapFlag = 1): the older peak picker. A local maximum arms a peak once it stands more than
// 0.11 range above the last local minimum (comparisons in double, as written there); the peak is counted when the contour falls more
// than 0.09 range below it or ends. The walk runs twice -- count / sums first, then the squared deviations of the peak distances from
// their mean in the same order (instead of the reference's list of distances).
struct PkoAcc { int64_t nPeaks, nDist; float peakMean, distSum, mean, dev; bool second; };
__device__ void pko_walk(const Col &in, float range, PkoAcc &r) {
  const int64_t Nin = in.N;
  float lastMin = 0.0f, lastMax = 0.0f;
  int64_t curmaxPos = 0, lastmaxPos = -1;
  bool peakflag = false;
  float lastlastVal = in[0], lastVal = Nin > 1 ? in[1] : 0.0f;
  for_rows(in, 2, Nin, [&](int64_t i, float v) {
    if ((lastlastVal < lastVal) && (lastVal > v)) {
      if (!peakflag) lastMax = v;
      else if (v > lastMax) { lastMax = v; curmaxPos = i; }
      if ((double)(lastMax - lastMin) > 0.11 * (double)range) { peakflag = true; curmaxPos = i; }
    } else if ((lastlastVal > lastVal) && (lastVal < v)) {
      lastMin = v;
    }
    if (peakflag && (((double)v < (double)lastMax - 0.09 * (double)range) || (i == Nin - 1))) {
      if (!r.second) { r.nPeaks++; r.peakMean += lastMax; }
      if (lastmaxPos >= 0) {
        const float dist = (float)(curmaxPos - lastmaxPos);
        if (!r.second) { r.distSum += dist; r.nDist++; }
        else { const float d = (float)(int64_t)dist - r.mean; r.dev += d * d; }
      }
      lastmaxPos = curmaxPos;
      peakflag = false;
    }
    lastlastVal = lastVal;
    lastVal = v;
  });
}

__device__ int f_peaks_old(const smilehip_func_spec &s, const Col &in, float min, float max, float *out) {
  const int64_t Nin = in.N;
  float mean = in[0];                                    // the family's own mean: one float chain from in[0] on (:119-124)
  for_rows(in, 1, Nin, [&](int64_t, float v) { mean += v; });
  mean /= (float)Nin;
  const float range = max - min;
  PkoAcc r{0, 0, 0.0f, 0.0f, 0.0f, 0.0f, false};
  pko_walk(in, range, r);
  float peakDist = r.distSum, stddev = 0.0f;
  if (r.nDist > 0) {
    peakDist /= (float)r.nDist;
    r.second = true;
    r.mean = peakDist;
    pko_walk(in, range, r);
    stddev = r.dev / (float)r.nDist;
    stddev = sqrtf(stddev);
  } else {
    peakDist = (float)(Nin + 1);
  }
  const uint32_t m = s.pko_mask;
  int n = 0;
  if (FS_BIT(m, 0)) out[n++] = (float)r.nPeaks;
  if (s.pko_norm == SMILEHIP_NORM_SECOND) { peakDist *= (float)s.period; stddev *= (float)s.period; }
  else if (s.pko_norm == SMILEHIP_NORM_SEGMENT) { peakDist /= (float)Nin; stddev /= (float)Nin; }
  if (FS_BIT(m, 1)) out[n++] = peakDist;
  float peakMean = r.nPeaks > 0 ? r.peakMean / (float)r.nPeaks : 0.0f;
  if (FS_BIT(m, 2)) out[n++] = peakMean;
  if (FS_BIT(m, 3)) out[n++] = peakMean - mean;
  if (FS_BIT(m, 4)) out[n++] = stddev;
  return n;
}

// Crossings (functionalCrossings.cpp:66-97): zero- and mean-crossing rate of the contour; the mean is a double chain in index order,
// the mean-crossing products are double (float - double)
__device__ int f_crossings(const smilehip_func_spec &s, const Col &in, float *out) {
  const int64_t Nin = in.N;
  const uint32_t m = s.crs_mask;
  double amean = 0.0;
  if (FS_BIT(m, 1) || FS_BIT(m, 2)) {
    amean = (double)in[0];
    for_rows(in, 1, Nin, [&](int64_t, float v) { amean += (double)v; });
    amean /= (double)Nin;
  }
  int64_t zcr = 0, mcr = 0;
  const bool want_m = FS_BIT(m, 1) != 0;
  float pm = in[0], p0 = Nin > 1 ? in[1] : 0.0f;
  for_rows(in, 2, Nin, [&](int64_t, float pn) {            // sample i = p0 with its neighbours pm, pn (i = 1 .. Nin - 2)
    if (((pm * pn <= 0.0f) && (p0 == 0.0f)) || (pm * p0 < 0.0f)) zcr++;
    if (want_m) {
      const double a = (double)pm - amean, b = (double)p0 - amean, c = (double)pn - amean;
      if (((a * c <= 0.0) && (b == 0.0)) || (a * b < 0.0)) mcr++;
    }
    pm = p0; p0 = pn;
  });
  int n = 0;
  if (FS_BIT(m, 0)) out[n++] = (float)((double)zcr / (double)Nin);
  if (FS_BIT(m, 1)) out[n++] = (float)((double)mcr / (double)Nin);
  if (FS_BIT(m, 2)) out[n++] = (float)amean;
  return n;
}

// DCT (functionalDCT.cpp:84-137): out[i] = factor * sum over m of in[m] * (FLOAT_DMEM)cos(pi i / N * (m + 0.5)) as ONE float chain per
// coefficient; the reference keeps the cosines in a table per contour length, here they are formed where they are used (double cos,
// rounded to float -- the table's values)
__device__ int f_dct(const smilehip_func_spec &s, const Col &in, float *out) {
  const int64_t Nin = in.N;
  const int nCo = s.dct_last - s.dct_first + 1;
  const float factor = (float)sqrt(2.0 / (double)Nin);
  for (int i = 0; i < nCo; ++i) {
    const double w = M_PI * (double)(i + s.dct_first) / (double)Nin;
    float acc = 0.0f;
    for_rows(in, 0, Nin, [&](int64_t m, float v) { acc += v * (float)cos(w * ((double)(float)m + 0.5)); });
    acc *= factor;
    out[i] = isfinite(acc) ? acc : 0.0f;
  }
  return nCo;
}

// Samples (functionalSamples.cpp:100-117): the contour at relative positions
__device__ int f_samples(const smilehip_func_spec &s, const Col &in, float *out) {
  const float Nind = (float)in.N;
  for (int k = 0; k < s.n_samples; ++k) {
    const int si = (int)(((double)Nind - 1.0) * s.sample_pos[k]);
    out[k] = in[si];
  }
  return s.n_samples;
}

// Segments: the segmentation runs twice -- first to get count / sum / extremes of the segment lengths, then again to
// accumulate the squared deviations from the mean in the same order (instead of keeping the reference's segLens[])
struct SegAcc {
  int64_t n, sum, maxl, minl, cap;
  bool second;
  float mean, dev;
  __device__ void add(int64_t i, int64_t last) {
    const int64_t len = i - last;
    if (n < cap) {
      if (!second) {
        sum += len;
        if (len > maxl) maxl = len;
        if ((minl == 0) || (len < minl)) minl = len;
      } else {
        dev += ((float)len - mean) * ((float)len - mean);
      }
      n++;
    }
  }
};

// The walks of functionalSegments.cpp: process_SegDelta :227-264, process_SegDelta2 :266-307, process_SegThresh :309-367,
// process_SegThreshNoavg :369-413, process_SegChX :560-653, process_SegNonX / process_SegEqX :656-799
__device__ void seg_walk(const smilehip_func_spec &s, const Col &in, float min, float range, float amean, SegAcc &r) {
  const int64_t Nin = in.N;
  const int algo = s.seg_algo;
  int64_t segMinLng = s.seg_min_lng;
  if (algo != SMILEHIP_SEG_NONX && algo != SMILEHIP_SEG_EQX && algo != SMILEHIP_SEG_CHX && s.seg_auto_min_lng) {
    segMinLng = Nin / s.seg_max_num - 1;
    if (segMinLng < 2) segMinLng = 2;
  }
  if (algo == SMILEHIP_SEG_DELTA || algo == SMILEHIP_SEG_DELTA2) {
    const float segThresh = range * s.seg_range_rel_threshold;
    const int64_t L = (s.seg_ravg_lng > 0) ? s.seg_ravg_lng : Nin / (s.seg_max_num / 2);
    int64_t lastSeg = -segMinLng / 2;
    const bool d2 = algo == SMILEHIP_SEG_DELTA2;
    float ravg = 0.0f, raLast = 0.0f, prev = 0.0f;
    for_rows(in, 0, Nin, [&](int64_t i, float v) {
      if (d2 && i == 0) { ravg = v; prev = v; return; }    // delt2 starts its average at in[0] and its loop at 1
      ravg += v;
      if (i >= L) ravg -= in[i - L];
      const float ra = ravg / (float)((i + 1 < L) ? (i + 1) : L);
      const bool hit = d2 ? ((prev - raLast <= segThresh) && (v - ra > segThresh)) : (v - ra > segThresh);
      if (hit && (i - lastSeg > segMinLng)) { r.add(i, lastSeg); lastSeg = i; }
      raLast = ra; prev = v;
    });
  } else if (algo == SMILEHIP_SEG_RELTH || algo == SMILEHIP_SEG_MRELTH || algo == SMILEHIP_SEG_ABSTH ||
             algo == SMILEHIP_SEG_NARELTH || algo == SMILEHIP_SEG_NAMRELTH || algo == SMILEHIP_SEG_NAABSTH) {
    const bool rel = algo == SMILEHIP_SEG_RELTH || algo == SMILEHIP_SEG_NARELTH;
    const bool mrel = algo == SMILEHIP_SEG_MRELTH || algo == SMILEHIP_SEG_NAMRELTH;
    const bool avg = algo == SMILEHIP_SEG_RELTH || algo == SMILEHIP_SEG_MRELTH || algo == SMILEHIP_SEG_ABSTH;
    const int nth = (algo == SMILEHIP_SEG_ABSTH) ? 0 : s.seg_n_thresholds;     // absTh: :179-180 never reads the array
    float th[8];
    for (int j = 0; j < 8; ++j)
      th[j] = (j < nth) ? (rel ? min + range * s.seg_thresholds[j] : mrel ? amean * s.seg_thresholds[j] : s.seg_thresholds[j]) : 0.0f;
    int64_t lastSeg = -segMinLng / 2;
    float ravg = 0.0f, raLast = 0.0f;
    float h1 = 0.0f, h2 = 0.0f, h3 = 0.0f;           // in[i-1], in[i-2], in[i-3]
    for_rows(in, 0, Nin, [&](int64_t i, float v) {
      float ra, last;
      if (avg) {
        ravg += v;
        if (i >= 3) ravg -= h3;
        ra = ravg / (float)((i + 1 < 3) ? (i + 1) : 3);
        last = raLast;
        raLast = ra;
      } else {
        ra = v; last = h1;
      }
      bool cross = false;
      if (avg || i >= 1)
        for (int j = 0; j < 8; ++j)
          if (j < nth && ((ra > th[j] && last <= th[j]) || (ra < th[j] && last >= th[j]))) cross = true;
      if (cross && (i - lastSeg > segMinLng)) { r.add(i, lastSeg); lastSeg = i; }
      h3 = h2; h2 = h1; h1 = v;
    });
  } else if (algo == SMILEHIP_SEG_CHX) {
    const float X = s.seg_x_is_rel ? (min + range * s.seg_x) : s.seg_x;
    int64_t segStartIndex = 0, segEndIndex = 0;
    int inSeg = 0, segStart = 0, segEnd = 0;
    for_rows(in, 0, Nin, [&](int64_t i, float v) {
      if (v != X) {
        if (inSeg == 1) {
          segEnd = 0;
          segStart++;
          if (segStart >= s.seg_min_lng) { inSeg = 2; r.add(segStartIndex - 1, segEndIndex); segStart = 0; }
        } else if (inSeg == 0) {
          segStart++;
          segStartIndex = i;
          inSeg = 1;
        } else if (inSeg == 2) {
          segEnd = 0;
        } else if (inSeg == 3) {
          segStart++;
          if (segStart >= s.seg_min_lng) { inSeg = 2; segEnd = 0; segStart = 0; }
        }
      }
      if (v == X) {
        if (inSeg == 3) {
          segStart = 0;
          segEnd++;
          if (segEnd >= s.seg_min_lng) { inSeg = 0; r.add(segEndIndex - 1, segStartIndex); segEnd = 0; }
        } else if (inSeg == 2) {
          segEnd++;
          segEndIndex = i;
          inSeg = 3;
        } else if (inSeg == 0) {
          segStart = 0;
        } else if (inSeg == 1) {
          segEnd++;
          if (segEnd >= s.seg_pause_min_lng) { inSeg = 0; segEnd = 0; segStart = 0; }
        }
      }
    });
    if (inSeg == 2) r.add(segEndIndex - 1, segStartIndex);
    else if (inSeg == 0) r.add(segStartIndex - 1, segEndIndex);
  } else {
    const float X = s.seg_x_is_rel ? (min + range * s.seg_x) : s.seg_x;
    const bool eq = s.seg_algo == SMILEHIP_SEG_EQX;      // eqX (:728-799) is nonX with the two tests swapped
    int64_t startIdx = 0;
    const int64_t i = Nin;                       // the index after the walk, as the reference's loop leaves it
    int inSeg = 0, segStart = 0, segEnd = 0;
    for_rows(in, 0, Nin, [&](int64_t i, float v) {
      if (eq ? (v == X) : (v != X)) {
        if (inSeg == 1) {
          segEnd = 0;
          segStart++;
          if (segStart >= s.seg_min_lng) { segStart = 0; inSeg = 2; }
        } else if (inSeg == 0) {
          segStart++;
          startIdx = i;
          inSeg = 1;
        } else if (inSeg == 2) {
          segEnd = 0;
        }
      }
      if (eq ? (v != X) : (v == X)) {
        if (inSeg == 2) {
          segStart = 0;
          segEnd++;
          if (segEnd >= s.seg_pause_min_lng) {
            inSeg = 0;
            r.add(i - segEnd, startIdx);
            segEnd = 0;
          }
        } else if (inSeg == 1) {
          segEnd++;
          if (segEnd >= s.seg_pause_min_lng) { inSeg = 0; segEnd = 0; segStart = 0; }
        }
      }
    });
    if (inSeg == 2) {
      segEnd++;
      r.add(i - segEnd, startIdx);
    }
  }
}

__device__ int f_segments(const smilehip_func_spec &s, const Col &in, float min, float max, float amean, float *out) {
  const int64_t Nin = in.N;
  const float range = max - min;
  SegAcc r;
  r.n = r.sum = r.maxl = r.minl = 0; r.cap = s.seg_max_num; r.second = false; r.mean = r.dev = 0.0f;
  seg_walk(s, in, min, range, amean, r);
  const int64_t nSeg = r.n;
  float mean = (nSeg > 1) ? (float)r.sum / ((float)nSeg) : (float)r.sum;
  float lenDev = 0.0f;
  if (nSeg > 1 && FS_BIT(s.seg_mask, 4)) {
    r.second = true; r.n = 0; r.mean = mean; r.dev = 0.0f;
    seg_walk(s, in, min, range, amean, r);
    lenDev = r.dev / (float)nSeg;
    lenDev = (float)sqrt((double)lenDev);
  }
  const uint32_t m = s.seg_mask;
  int n = 0;
  if (FS_BIT(m, 0)) {
    if (s.seg_norm == SMILEHIP_NORM_SECOND) {
      const float T = (float)s.period;
      float Norm = 1.0f;
      if (T != 0.0f) Norm = T;
      Norm *= (float)Nin;
      out[n++] = (float)nSeg / Norm;
    } else if (s.seg_norm == SMILEHIP_NORM_SEGMENT) out[n++] = (float)nSeg / (float)(s.seg_max_num);
    else out[n++] = (float)nSeg;
  }
  if (s.seg_norm == SMILEHIP_NORM_SEGMENT) {
    if (FS_BIT(m, 1)) out[n++] = mean / (float)(Nin);
    if (FS_BIT(m, 2)) out[n++] = (float)r.maxl / (float)(Nin);
    if (FS_BIT(m, 3)) out[n++] = (float)r.minl / (float)(Nin);
    if (FS_BIT(m, 4)) out[n++] = lenDev / (float)(Nin);
  } else if (s.seg_norm == SMILEHIP_NORM_FRAME) {
    if (FS_BIT(m, 1)) out[n++] = mean;
    if (FS_BIT(m, 2)) out[n++] = (float)r.maxl;
    if (FS_BIT(m, 3)) out[n++] = (float)r.minl;
    if (FS_BIT(m, 4)) out[n++] = lenDev;
  } else {
    const float T = (float)s.period;
    float Norm = 1.0f;
    if (T != 0.0f) Norm = T;
    if (FS_BIT(m, 1)) out[n++] = mean * Norm;
    if (FS_BIT(m, 2)) out[n++] = (float)r.maxl * Norm;
    if (FS_BIT(m, 3)) out[n++] = (float)r.minl * Norm;
    if (FS_BIT(m, 4)) out[n++] = lenDev * Norm;
  }
  return n;
}

// Lpc: the reference fills acf[lag] with one sequential float sum per lag (lags p .. 0); here all lags advance in one
// walk over the contour with a register delay line -- each sum still adds its products in index order.
template <int P>
__device__ int f_lpc_p(const smilehip_func_spec &s, const Col &in, float *out) {
  const int64_t Nin = in.N;
  float acf[P + 1], d[P + 1], a[P + 1];
#pragma unroll
  for (int k = 0; k <= P; ++k) { acf[k] = 0.0f; d[k] = 0.0f; a[k] = 0.0f; }
  const int n32 = (int)Nin;                              // smileDsp_autoCorr takes an int
  for_rows(in, 0, (int64_t)n32, [&](int64_t i, float v) {
#pragma unroll
    for (int k = P; k >= 1; --k) d[k] = d[k - 1];
    d[0] = v;
#pragma unroll
    for (int k = 0; k <= P; ++k)
      if (i >= k) acf[k] += d[0] * d[k];
  });
  float gain = 0.0f;
  if (!(acf[0] == 0.0f)) {
    float e = acf[0];
#pragma unroll
    for (int m = 1; m <= P; ++m) {
      float sum = 1.0f * acf[m];
#pragma unroll
      for (int i = 1; i < m; ++i) sum += a[i - 1] * acf[m - i];
      const float k_m = (-1.0f / e) * sum;
      a[m - 1] = k_m;
#pragma unroll
      for (int i = 1; i <= m / 2; ++i) {
        const float x = a[i - 1];
        a[i - 1] += k_m * a[m - i - 1];
        if ((i < (m / 2)) || ((m & 1) == 1)) a[m - i - 1] += k_m * x;
      }
      e *= (1.0f - k_m * k_m);
      if (e == 0.0f) {
#pragma unroll
        for (int i = 0; i <= P; ++i)
          if (i >= m) a[i] = 0.0f;
        break;
      }
    }
    gain = e;
  }
  int n = 0;
  if (s.lpc_gain) out[n++] = gain / (float)Nin;
  if (s.lpc_coeffs) {
#pragma unroll
    for (int i = 0; i < P; ++i)
      if (i >= s.lpc_first) out[n++] = a[i];
  }
  return n;
}

__device__ int f_lpc(const smilehip_func_spec &s, const Col &in, float *out) {
  switch (s.lpc_order) {
    case 1: return f_lpc_p<1>(s, in, out);
    case 2: return f_lpc_p<2>(s, in, out);
    case 3: return f_lpc_p<3>(s, in, out);
    case 4: return f_lpc_p<4>(s, in, out);
    case 5: return f_lpc_p<5>(s, in, out);
    case 6: return f_lpc_p<6>(s, in, out);
    case 7: return f_lpc_p<7>(s, in, out);
    case 8: return f_lpc_p<8>(s, in, out);
    case 10: return f_lpc_p<10>(s, in, out);
    case 12: return f_lpc_p<12>(s, in, out);
    case 16: return f_lpc_p<16>(s, in, out);
  }
  return 0;
}

// ---- Peaks2
__device__ __forceinline__ bool pk_below(const smilehip_func_spec &s, float absThresh, float diff, float base) {
  if (s.pk_dyn_rel) {
    if (base == 0.0f) return diff != 0.0f;
    return fabs((double)(diff / base)) < (double)s.pk_rel_thresh;
  }
  return diff < absThresh;
}
__device__ __forceinline__ float pk_rl(const smilehip_func_spec &s, float x) { return s.pk_ratio_limit ? fs_ratio_limit(x, 10.0f, 10.0f) : x; }
__device__ __forceinline__ float pk_rlmax(const smilehip_func_spec &s, float alt) { return s.pk_ratio_limit ? 20.0f : alt; }
__device__ __forceinline__ float pk_rlu(const smilehip_func_spec &s, float x) {
  if (s.pk_ratio_limit) {
    if (x > 1.0f) return 1.0f;
    if (x < -1.0f) return -1.0f;
  }
  return x;
}

// walks the local extrema of the contour in index order: calls f(i, y, is_max) for rows 2 .. N-3 that are strict
// local maxima / minima (functionalPeaks2.cpp:343-349) and, if `only_alive`, still alive
template <typename F>
__device__ __forceinline__ void pk_for_each(const Col &in, const unsigned char *alive, int64_t ald, bool only_alive, F f) {
  const int64_t N = in.N;
  if (N < 5) return;
  float a = in[1], b = in[2];
  auto step = [&](int64_t i, float c, bool live) {
    const bool mx = b > a && b > c, mi = b < a && b < c;
    if ((mx || mi) && live) f(i, b, mx);
    a = b; b = c;
  };
  // the alive bytes of the rows ahead are fetched with the samples: a pass only ever clears the flag of the row it is
  // at or of an earlier one, so a prefetched flag cannot be stale
  int64_t j = 3;                                 // j = i + 1
  for (; j + 8 <= N - 1; j += 8) {
    float v[8];
    unsigned char al[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = in[j + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) al[k] = only_alive ? alive[(j + k - 1) * ald] : (unsigned char)1;
#pragma unroll
    for (int k = 0; k < 8; ++k) step(j + k - 1, v[k], al[k] != 0);
  }
  for (; j < N - 1; ++j) step(j - 1, in[j], !only_alive || alive[(j - 1) * ald] != 0);
}

__device__ int f_peaks2(const smilehip_func_spec &s, const Col &in, float min, float max, float mean, unsigned char *alive,
                        int64_t ald, float *out) {
  const int64_t Nin = in.N;
  const float range = max - min;
  const float absThresh = s.pk_use_abs ? s.pk_abs_thresh : s.pk_rel_thresh * range;
  const float in0 = in[0], inL = in[Nin - 1];
  // pass 1: minimum rise / fall (functionalPeaks2.cpp:352-404); an element = its row index
  {
    float lastVal = in0, lastMin = in0, lastMax = in0;
    bool minFlag = false;
    int64_t lastMaxPtr = -1;
    pk_for_each(in, alive, ald, false, [&](int64_t i, float y, bool is_max) {
      unsigned char keep = 1;
      if (is_max) {
        if (pk_below(s, absThresh, (float)fabs((double)(y - lastVal)), fminf(y, lastVal))) {
          if (pk_below(s, absThresh, y - lastMin, lastMin)) {
            keep = 0;
          } else {
            if ((double)y > (double)lastMax * 1.05) {
              if (lastMaxPtr != -1) alive[lastMaxPtr * ald] = 0;
              lastMax = y;
              lastMaxPtr = i;
            } else {
              if (minFlag) { lastMax = y; lastMaxPtr = i; }
              else keep = 0;
            }
            minFlag = false;
          }
        } else {
          minFlag = false;
          lastMax = y;
          lastMaxPtr = i;
        }
      } else {
        if (!pk_below(s, absThresh, (float)fabs((double)(y - lastVal)), fminf(y, lastVal))) {
          minFlag = true;
          lastMin = y;
        }
      }
      lastVal = y;
      alive[i * ald] = keep;
    });
  }
  // pass 2: minima too close below the last maximum (:407-421)
  {
    float lastMax = in0;
    pk_for_each(in, alive, ald, true, [&](int64_t i, float y, bool is_max) {
      if (!is_max) {
        if (pk_below(s, absThresh, lastMax - y, y)) alive[i * ald] = 0;
      } else lastMax = y;
    });
  }
  // pass 3: alternation (:424-470)
  {
    float lastMax = in0, lastMin = in0;
    bool minFlag = false, init = true;
    int64_t lastMaxPtr = -1, lastMinPtr = -1;
    pk_for_each(in, alive, ald, true, [&](int64_t i, float y, bool is_max) {
      if (!is_max) {
        if (!minFlag || init) { lastMin = y; lastMinPtr = i; minFlag = true; init = false; }
        else {
          if (y >= lastMin) alive[i * ald] = 0;
          else { alive[lastMinPtr * ald] = 0; lastMinPtr = i; lastMin = y; }
        }
      } else {
        if (minFlag || init) { lastMax = y; lastMaxPtr = i; minFlag = false; init = false; }
        else {
          if (y <= lastMax) alive[i * ald] = 0;
          else { alive[lastMaxPtr * ald] = 0; lastMaxPtr = i; lastMax = y; }
        }
      }
    });
  }
  // statistics of the surviving extrema (:474-560) and the slopes between them (:640-745). The reference takes four
  // walks over its list (sums, squared deviations, slopes, squared slope deviations); the sums of the first and third
  // are independent of each other, as are those of the second and fourth: two walks here, every accumulator in its own
  // order.
  float peakMax = 0.0f, peakMin = 0.0f, peakDist = 0.0f, peakDiff = 0.0f, peakStddevDist = 0.0f, peakStddevDiff = 0.0f;
  float peakMean = 0.0f, minMax = 0.0f, minMin = 0.0f, minDist = 0.0f, minDiff = 0.0f, minStddevDist = 0.0f;
  float minStddevDiff = 0.0f, minMean = 0.0f;
  int64_t nPeakDist = 0, nPeaks = 0, nMinDist = 0, nMins = 0;
  float meanRisingSlope = 0.0f, meanFallingSlope = 0.0f, minRisingSlope = 0.0f, maxRisingSlope = 0.0f;
  float minFallingSlope = 0.0f, maxFallingSlope = 0.0f, stddevRisingSlope = 0.0f, stddevFallingSlope = 0.0f;
  int nRising = 0, nFalling = 0, lastIsMax = -1;
  const bool slopes = (s.pk_mask & 0xffc00000u) != 0;
  const float T = (float)s.period;
  auto fall = [&](float slope) {
    meanFallingSlope += slope;
    if (nFalling == 0) { minFallingSlope = slope; maxFallingSlope = slope; }
    else {
      if (slope < minFallingSlope) minFallingSlope = slope;
      if (slope > maxFallingSlope) maxFallingSlope = slope;
    }
    nFalling++;
  };
  auto rise = [&](float slope) {
    meanRisingSlope += slope;
    if (nRising == 0) { minRisingSlope = slope; maxRisingSlope = slope; }
    else {
      if (slope < minRisingSlope) minRisingSlope = slope;
      if (slope > maxRisingSlope) maxRisingSlope = slope;
    }
    nRising++;
  };
  {
    int64_t lmx = -1, lmn = -1;
    float lmy = 0.0f, lny = 0.0f;
    float lastMax = in0, lastMin = in0;
    int64_t lastMaxPos = 0, lastMinPos = 0;
    pk_for_each(in, alive, ald, true, [&](int64_t i, float y, bool is_max) {
      if (!is_max) {
        if (lmn == -1) { minMin = y; minMax = y; }
        else {
          nMinDist++;
          minDist += (float)(i - lmn);
          minDiff += (float)fabs((double)(y - lny));
          if (minMin > y) minMin = y;
          if (minMax < y) minMax = y;
        }
        lmn = i; lny = y;
        minMean += y;
        nMins++;
        if (slopes) {
          lastMin = y; lastMinPos = i;
          if (lastMinPos - lastMaxPos > 0) { fall((lastMax - lastMin) / ((float)(lastMinPos - lastMaxPos) * T)); lastIsMax = 0; }
        }
      } else {
        if (lmx == -1) { peakMin = y; peakMax = y; }
        else {
          nPeakDist++;
          peakDist += (float)(i - lmx);
          peakDiff += (float)fabs((double)(y - lmy));
          if (peakMin > y) peakMin = y;
          if (peakMax < y) peakMax = y;
        }
        lmx = i; lmy = y;
        peakMean += y;
        nPeaks++;
        if (slopes) {
          lastMax = y; lastMaxPos = i;
          if (lastMaxPos - lastMinPos > 0) { rise((lastMax - lastMin) / ((float)(lastMaxPos - lastMinPos) * T)); lastIsMax = 1; }
        }
      }
    });
    if (slopes) {
      if (lastIsMax == 1) {
        if (Nin - 1 - lastMaxPos > 0) fall((inL - lastMax) / ((float)(Nin - 1 - lastMaxPos) * T));
      } else if (lastIsMax == 0) {
        if (Nin - 1 - lastMinPos > 0) rise((inL - lastMin) / ((float)(Nin - 1 - lastMinPos) * T));
      } else {
        const float slope = (inL - in0) / (float)Nin;
        if (slope > 0) { meanRisingSlope = maxRisingSlope = minRisingSlope = slope; nRising = 1; }
        else if (slope < 0) { meanFallingSlope = maxFallingSlope = minFallingSlope = slope; nFalling = 1; }
      }
      if (nRising > 1) meanRisingSlope /= (float)nRising;
      if (nFalling > 1) meanFallingSlope /= (float)nFalling;
    }
  }
  if (nPeaks > 1) {
    peakMean /= (float)nPeaks;
    if (nPeakDist > 1) { peakDist /= (float)nPeakDist; peakDiff /= (float)nPeakDist; }
  }
  if (nMins > 0) {
    minMean /= (float)nMins;
    if (nMinDist > 1) { minDist /= (float)nMinDist; minDiff /= (float)nMinDist; }
  }
  {
    bool haveMax = false;
    int64_t lmn = -1;
    float lny = 0.0f;
    float lastMax = in0, lastMin = in0;
    int64_t lastMaxPos = 0, lastMinPos = 0;
    pk_for_each(in, alive, ald, true, [&](int64_t i, float y, bool is_max) {
      if (!is_max) {
        if (lmn != -1) {
          const float dx = (float)(i - lmn), dy = (float)fabs((double)(y - lny));
          minStddevDist += (dx - minDist) * (dx - minDist);
          minStddevDiff += (dy - minDiff) * (dy - minDiff);
        }
        lmn = i; lny = y;
        if (slopes) {
          lastMin = y; lastMinPos = i;
          if (lastMinPos - lastMaxPos > 0) {
            const float slope = (lastMax - lastMin) / ((float)(lastMinPos - lastMaxPos) * T);
            stddevFallingSlope += (slope - meanFallingSlope) * (slope - meanFallingSlope);
          }
        }
      } else {
        if (!haveMax) haveMax = true;
        else if (lmn != -1) {                    // measured against the last MINIMUM, as the reference does (:594-598)
          const float dx = (float)(i - lmn), dy = (float)fabs((double)(y - lny));
          peakStddevDist += (dx - peakDist) * (dx - peakDist);
          peakStddevDiff += (dy - peakDiff) * (dy - peakDiff);
        }
        if (slopes) {
          lastMax = y; lastMaxPos = i;
          if (lastMaxPos - lastMinPos) {
            const float slope = (lastMax - lastMin) / ((float)(lastMaxPos - lastMinPos) * T);
            stddevRisingSlope += (slope - meanRisingSlope) * (slope - meanRisingSlope);
          }
        }
      }
    });
  }
  if (nPeakDist > 1) { peakStddevDist /= (float)nPeakDist; peakStddevDiff /= (float)nPeakDist; }
  peakStddevDist = (peakStddevDist > 0.0f) ? (float)sqrt((double)peakStddevDist) : 0.0f;
  peakStddevDiff = (peakStddevDiff > 0.0f) ? (float)sqrt((double)peakStddevDiff) : 0.0f;
  if (nMinDist > 1) { minStddevDist /= (float)nMinDist; minStddevDiff /= (float)nMinDist; }
  minStddevDist = (minStddevDist > 0.0f) ? (float)sqrt((double)minStddevDist) : 0.0f;
  minStddevDiff = (minStddevDiff > 0.0f) ? (float)sqrt((double)minStddevDiff) : 0.0f;
  if (slopes) {
    if (nRising > 1) stddevRisingSlope /= (float)nRising;
    if (nFalling > 1) stddevFallingSlope /= (float)nFalling;
    stddevRisingSlope = (stddevRisingSlope > 0.0f) ? (float)sqrt((double)stddevRisingSlope) : 0.0f;
    stddevFallingSlope = (stddevFallingSlope > 0.0f) ? (float)sqrt((double)stddevFallingSlope) : 0.0f;
  }
  if (s.pk_norm == SMILEHIP_NORM_SECOND) {
    const float T = (float)s.period;
    peakDist *= T; peakStddevDist *= T; minDist *= T; minStddevDist *= T;
  } else if (s.pk_norm == SMILEHIP_NORM_SEGMENT) {
    peakDist /= (float)Nin; peakStddevDist /= (float)Nin; minDist /= (float)Nin; minStddevDist /= (float)Nin;
  }
  const uint32_t m = s.pk_mask;
  int n = 0;
  if (FS_BIT(m, 0)) out[n++] = (s.pk_norm == SMILEHIP_NORM_SECOND) ? ((float)nPeaks) / ((float)Nin * (float)s.period) : (float)nPeaks;
  if (FS_BIT(m, 1)) out[n++] = peakDist;
  if (FS_BIT(m, 2)) out[n++] = 0.0f;
  if (FS_BIT(m, 3)) out[n++] = peakStddevDist;
  if (FS_BIT(m, 4)) out[n++] = peakMax - peakMin;
  if (FS_BIT(m, 5)) out[n++] = (range != 0.0f) ? pk_rlu(s, (float)fabs((double)((peakMax - peakMin) / range))) : peakMax - peakMin;
  if (FS_BIT(m, 6)) out[n++] = peakMean;
  if (FS_BIT(m, 7)) out[n++] = peakMean - mean;
  if (FS_BIT(m, 8)) out[n++] = (mean != 0.0f) ? pk_rl(s, peakMean / mean) : pk_rlmax(s, peakMean);
  if (FS_BIT(m, 9)) out[n++] = peakDiff;
  if (FS_BIT(m, 10)) out[n++] = (range != 0.0f) ? pk_rlu(s, peakDiff / range) : peakDiff;
  if (FS_BIT(m, 11)) out[n++] = peakStddevDiff;
  if (FS_BIT(m, 12)) out[n++] = (range != 0.0f) ? pk_rlu(s, peakStddevDiff / range) : peakStddevDiff;
  if (FS_BIT(m, 13)) out[n++] = minMax - minMin;
  if (FS_BIT(m, 14)) out[n++] = (range != 0.0f) ? pk_rlu(s, (float)fabs((double)((minMax - minMin) / range))) : minMax - minMin;
  if (FS_BIT(m, 15)) out[n++] = minMean;
  if (FS_BIT(m, 16)) out[n++] = mean - minMean;
  if (FS_BIT(m, 17)) out[n++] = (mean != 0.0f) ? pk_rl(s, minMean / mean) : pk_rlmax(s, minMean);
  if (FS_BIT(m, 18)) out[n++] = minDiff;
  if (FS_BIT(m, 19)) out[n++] = (range != 0.0f) ? pk_rlu(s, minDiff / range) : minDiff;
  if (FS_BIT(m, 20)) out[n++] = minStddevDiff;
  if (FS_BIT(m, 21)) out[n++] = (range != 0.0f) ? pk_rlu(s, minStddevDiff / range) : minStddevDiff;
  if (FS_BIT(m, 22)) out[n++] = meanRisingSlope;
  if (FS_BIT(m, 23)) out[n++] = maxRisingSlope;
  if (FS_BIT(m, 24)) out[n++] = minRisingSlope;
  if (FS_BIT(m, 25)) out[n++] = stddevRisingSlope;
  if (FS_BIT(m, 26)) out[n++] = meanFallingSlope;
  if (FS_BIT(m, 27)) out[n++] = maxFallingSlope;
  if (FS_BIT(m, 28)) out[n++] = minFallingSlope;
  if (FS_BIT(m, 29)) out[n++] = stddevFallingSlope;
  if (FS_BIT(m, 30)) out[n++] = (meanFallingSlope > 0.0f) ? pk_rl(s, stddevFallingSlope / meanFallingSlope) : 0.0f;
  if (FS_BIT(m, 31)) out[n++] = (meanRisingSlope > 0.0f) ? pk_rl(s, stddevRisingSlope / meanRisingSlope) : 0.0f;
  return n;
}

}  // namespace

template <int FAM>
__global__ void __launch_bounds__(kColsPerBlock) fs_family(FsParams P, int out_off, int want) {
  const Where w = locate(P);
  if (!w.on) return;
  const int64_t si = (int64_t)w.u * P.n_cols + w.c;
  if (P.st_n[si] <= 0) return;                          // zero-filled by fs_stats
  const Col x = data_col(P, w);
  const float mn = P.st_min[si], mx = P.st_max[si], mean = P.st_mean[si];
  float *o = P.out + (int64_t)w.u * P.ld_out + (int64_t)w.c * P.per + out_off;
  int got = 0;
  if (FAM == SMILEHIP_FAM_EXTREMES) got = f_extremes(P.spec, x, mn, mx, mean, o);
  if (FAM == SMILEHIP_FAM_MEANS) got = f_means(P.spec, x, mean, o);
  if (FAM == SMILEHIP_FAM_MOMENTS) got = f_moments(P.spec, x, mean, o);
  if (FAM == SMILEHIP_FAM_REGRESSION) got = f_regression(P.spec, x, mn, mx, mean, o);
  if (FAM == SMILEHIP_FAM_TIMES) got = f_times(P.spec, x, mn, mx, o);
  if (FAM == SMILEHIP_FAM_SEGMENTS) got = f_segments(P.spec, x, mn, mx, mean, o);
  if (FAM == SMILEHIP_FAM_LPC) got = f_lpc(P.spec, x, o);
  if (FAM == SMILEHIP_FAM_ONSET) got = f_onset(P.spec, x, o);
  if (FAM == SMILEHIP_FAM_PEAKS) got = f_peaks_old(P.spec, x, mn, mx, o);
  if (FAM == SMILEHIP_FAM_CROSSINGS) got = f_crossings(P.spec, x, o);
  if (FAM == SMILEHIP_FAM_DCT) got = f_dct(P.spec, x, o);
  if (FAM == SMILEHIP_FAM_SAMPLES) got = f_samples(P.spec, x, o);
  if (FAM == SMILEHIP_FAM_PEAKS2)
    got = f_peaks2(P.spec, x, mn, mx, mean, P.alive + w.srow0 * P.n_cols + w.c, P.n_cols, o);
  for (int j = got; j < want; ++j) o[j] = 0.0f;
}

// ------------------------------------------------------------------ percentiles
namespace {

__device__ float interp_pctl(double p, const float *sorted, int64_t N) {
  const double idx = p * (double)(N - 1);
  int64_t i1 = (int64_t)floor(idx), i2 = (int64_t)ceil(idx);
  if (i1 < 0) i1 = 0;
  if (i2 < 0) i2 = 0;
  if (i1 >= N) i1 = N - 1;
  if (i2 >= N) i2 = N - 1;
  if (i1 != i2) {
    const double w1 = idx - (double)i1, w2 = (double)i2 - idx;
    return sorted[i1] * (float)w2 + sorted[i2] * (float)w1;
  }
  return sorted[i1];
}
__device__ int64_t pctl_idx(double p, int64_t N) {
  int64_t r = (int64_t)round(p * (double)(N - 1));
  if (r < 0) return 0;
  if (r >= N) return N - 1;
  return r;
}

__device__ void bitonic_sort(float *a, int n2) {       // n2 = power of two, whole workgroup
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const float x = a[i], y = a[l];
          const bool up = (i & k) == 0;
          if (up ? (x > y) : (x < y)) { a[i] = y; a[l] = x; }
        }
      }
      __syncthreads();
    }
}

}  // namespace

__device__ void pctl_readout(const FsParams &P, const Where &w, const float *a, int64_t N, int out_off) {
  const smilehip_func_spec &s = P.spec;
  float *out = P.out + (int64_t)w.u * P.ld_out + (int64_t)w.c * P.per + out_off;
  float q1, q2, q3;
  if (s.pct_interp) { q1 = interp_pctl(0.25, a, N); q2 = interp_pctl(0.50, a, N); q3 = interp_pctl(0.75, a, N); }
  else { q1 = a[pctl_idx(0.25, N)]; q2 = a[pctl_idx(0.50, N)]; q3 = a[pctl_idx(0.75, N)]; }
  const uint32_t m = s.pct_mask;
  int n = 0;
  if (FS_BIT(m, 0)) out[n++] = q1;
  if (FS_BIT(m, 1)) out[n++] = q2;
  if (FS_BIT(m, 2)) out[n++] = q3;
  if (FS_BIT(m, 3)) out[n++] = q2 - q1;
  if (FS_BIT(m, 4)) out[n++] = q3 - q2;
  if (FS_BIT(m, 5)) out[n++] = q3 - q1;
  if (s.n_pctl > 0) {
    const int n0 = n;
    for (int i = 0; i < s.n_pctl; ++i) out[n++] = s.pct_interp ? interp_pctl(s.pctl[i], a, N) : a[pctl_idx(s.pctl[i], N)];
    for (int i = 0; i < s.n_range; ++i) {
      const float v = (float)fabs((double)(out[n0 + s.range_b[i]] - out[n0 + s.range_a[i]]));
      out[n++] = v;
    }
    // pctlquotient (:402-411): under the test of the RANGE switch and of the numerator; without ranges cFunctionals zero-fills
    for (int i = 0; i < s.n_quot; ++i) {
      float v = 0.0f;
      if (s.n_range > 0 && out[n0 + s.quot_a[i]] != 0.0f) v = fs_ratio_limit(out[n0 + s.quot_a[i]] / out[n0 + s.quot_b[i]], 50.0f, 100.0f);
      out[n++] = v;
    }
  }
}

__global__ void __launch_bounds__(kSortThreads) fs_percentiles(FsParams P, int out_off) {
  __shared__ float lds[kSortLds];
  Where w;
  w.u = blockIdx.x / P.n_cols;
  w.c = blockIdx.x % P.n_cols;
  w.on = true;
  utt_rows(P, w);
  const int64_t si = (int64_t)w.u * P.n_cols + w.c;
  const int64_t N = P.st_n[si];
  if (N <= 0 || N <= kWaveSortMaxDecl) return;           // (up to 1024 rows: fs_percentiles_wave)
  const Col x = data_col(P, w);
  int n2 = 1;
  while (n2 < N) n2 <<= 1;
  // global scratch beyond the LDS capacity: the utterance's region holds 2 * (rows + 1) * n_cols floats; the column slots
  // are spaced by the power of two covering the utterance's rows BEFORE nonZeroFuncts -- the columns' own sizes differ
  // with nonZeroFuncts, and slots spaced by them would overlap
  int64_t slot = 1;
  while (slot < w.rows) slot <<= 1;
  float *a = (n2 <= kSortLds) ? lds : P.sorted + 2 * w.srow0 * P.n_cols + (int64_t)w.c * slot;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) a[i] = (i < N) ? x[i] : INFINITY;
  __syncthreads();
  bitonic_sort(a, n2);
  if (threadIdx.x != 0) return;
  pctl_readout(P, w, a, N, out_off);
}

// ---- one WAVE per (utterance, column) for N <= 1024 (a 10 s contour): the SAME bitonic network as bitonic_sort above -- the
// same compare-exchanges on the same element pairs, so the sorted order is identical down to the placement of equal values
// -- with element e in lane e / R, register e % R (R = n2 / 64): strides below R are register pairs, strides from R up are
// lane exchanges (DPP quad permutes and row rotations, v_permlane16_swap / v_permlane32_swap) -- no LDS, no barrier, where
// the workgroup form spends 55 barrier-separated LDS passes of 256 threads on one contour and holds 32 KB of LDS for it.
template <int D>
__device__ __forceinline__ float lane_xor_f(float x, int lane) {           // lane l receives lane l ^ D
  const int v = __float_as_int(x);
  int r;
  if constexpr (D == 1) r = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
  else if constexpr (D == 2) r = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  else if constexpr (D == 4) {
    const int a = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, true);                  // row_ror:4: lane i <- i - 4
    const int b = __builtin_amdgcn_update_dpp(0, v, 0x12C, 0xf, 0xf, true);                  // row_ror:12: lane i <- i + 4
    r = (lane & 4) ? a : b;
  } else if constexpr (D == 8) r = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);  // row_ror:8
  else if constexpr (D == 16) { const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false); r = (lane & 16) ? p[0] : p[1]; }
  else { const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false); r = (lane & 32) ? p[0] : p[1]; }
  return __int_as_float(r);
}

template <int LOG2N>
__device__ __forceinline__ void wave_bitonic_sort(const Col &x, int64_t N, int lane, float *dst) {
  constexpr int N2 = 1 << LOG2N, R = N2 >= 64 ? N2 / 64 : 1;
  float v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { const int e = lane * R + r; v[r] = (e < N) ? x[e] : INFINITY; }
#pragma unroll
  for (int k = 2; k <= N2; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < R) {                                       // both elements in this lane: registers r and r | j
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r & j) continue;
          const bool up = ((lane * R + r) & k) == 0;
          const float a = v[r], b = v[r | j];
          const bool sw = up ? (a > b) : (a < b);
          v[r] = sw ? b : a;
          v[r | j] = sw ? a : b;
        }
      } else {                                           // the partner element is register r of lane ^ (j / R)
        const bool lower = (lane & (j / R)) == 0;
        const bool up = ((lane * R) & k) == 0;           // (k >= 2 R here: the bit is a lane bit)
        const bool want_gt = lower == up;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float a = v[r];
          float b;
          switch (j / R) {
            case 1: b = lane_xor_f<1>(a, lane); break;
            case 2: b = lane_xor_f<2>(a, lane); break;
            case 4: b = lane_xor_f<4>(a, lane); break;
            case 8: b = lane_xor_f<8>(a, lane); break;
            case 16: b = lane_xor_f<16>(a, lane); break;
            default: b = lane_xor_f<32>(a, lane); break;
          }
          const bool sw = want_gt ? (a > b) : (a < b);
          v[r] = sw ? b : a;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) dst[lane * R + r] = v[r];
}

constexpr int kWaveSortMax = 1024;
__global__ void __launch_bounds__(256) fs_percentiles_wave(FsParams P, int out_off, int n_items) {
  __shared__ float lds_all[4 * kWaveSortMax];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  Where w;
  w.u = item / P.n_cols;
  w.c = item % P.n_cols;
  w.on = true;
  utt_rows(P, w);
  const int64_t N = P.st_n[(int64_t)w.u * P.n_cols + w.c];
  if (N <= 0 || N > kWaveSortMax) return;                // longer contours: fs_percentiles
  const Col x = data_col(P, w);
  float *a = lds_all + wave * kWaveSortMax;
  int lg = 1;
  while ((1 << lg) < N) ++lg;
  switch (lg) {
    case 1: wave_bitonic_sort<1>(x, N, lane, a); break;
    case 2: wave_bitonic_sort<2>(x, N, lane, a); break;
    case 3: wave_bitonic_sort<3>(x, N, lane, a); break;
    case 4: wave_bitonic_sort<4>(x, N, lane, a); break;
    case 5: wave_bitonic_sort<5>(x, N, lane, a); break;
    case 6: wave_bitonic_sort<6>(x, N, lane, a); break;
    case 7: wave_bitonic_sort<7>(x, N, lane, a); break;
    case 8: wave_bitonic_sort<8>(x, N, lane, a); break;
    case 9: wave_bitonic_sort<9>(x, N, lane, a); break;
    default: wave_bitonic_sort<10>(x, N, lane, a); break;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) pctl_readout(P, w, a, N, out_off);
}

// ------------------------------------------------------------------ launch
int fs_sort_lds_rows() { return kSortLds; }

// ------------------------------------------------------------------ Modulation (cFunctionalModulation, functionalModulation.cpp)
// One wave per (utterance, column): the windows of the contour in turn -- window function of the window's own length (host table),
// zero padding to the next power of two, the reference-order transform (lld_ooura.hpp, the in-place LDS form), magnitudes, the
// natural cubic spline through them (two sequential recurrences in double on lane 0: smileMath_cspline, smileUtilSpline.c:155-211,
// with the per-knot constants of mod_prepare), its values at the output bins (smileMath_csplint :344-357), added in window order;
// the average of the windows that count (computeModSpecSTFTavg :452-478). A contour whose first window has fewer than 33 values,
// or whose bins lie off a window's frequency axis, gives NaNs (the reference transforms down to 4 points; not built).
namespace {
struct ModG {
  __device__ static __forceinline__ int tid() { return threadIdx.x & 63; }
  __device__ static __forceinline__ int size() { return 64; }
  __device__ static __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};
}  // namespace

__global__ void __launch_bounds__(64) fs_modulation(FsParams P, int out_off) {
  __shared__ float2 z[512];
  __shared__ double yv[516], y2[516], uu[516];
  __shared__ float acc[128];
  Where w;
  w.u = blockIdx.x / P.n_cols;
  w.c = blockIdx.x % P.n_cols;
  w.on = true;
  utt_rows(P, w);
  const smilehip_func_spec &s = P.spec;
  const int64_t Nin = P.st_n[(int64_t)w.u * P.n_cols + w.c];
  if (Nin <= 0) return;
  const Col x = data_col(P, w);
  const int lane = threadIdx.x & 63, nb = s.mod_n_bins, W = s.mod_win_frames, step = s.mod_step_frames;
  float *out = P.out + (int64_t)w.u * P.ld_out + (int64_t)w.c * P.per + out_off;
  float mean = 0.0f;
  if (s.mod_remove_nz_mean) {                            // :512-541 -- a FLOAT_DMEM chain; every lane walks it (uniform loads)
    int64_t n_mean = 0;
    for_rows(x, 0, Nin, [&](int64_t, float v) { if (v != 0.0f) { mean += v; n_mean++; } });
    if (n_mean > 0) mean /= (float)n_mean;
  }
  const auto val = [&](int64_t t) {
    const float v = x[t];
    return s.mod_remove_nz_mean ? ((v != 0.0f) ? v - mean : 0.0f) : v;
  };
  for (int i = lane; i < nb; i += 64) acc[i] = 0.0f;
  ModG::sync();
  int n_spec = 0;
  bool bad = false;
  for (int64_t n = 0; n < Nin; n += step) {
    const int64_t N = (W < Nin - n - 1) ? W : Nin - n - 1;
    if (!(N > 2 * W / 3 || n_spec == 0)) continue;
    if (N < 33) { bad = true; break; }
    int si = 0;
    while ((64 << si) < N) ++si;
    ModSizeTab T;
    switch (si) { case 0: T = P.mod.size[0]; break; case 1: T = P.mod.size[1]; break; case 2: T = P.mod.size[2]; break;
                  case 3: T = P.mod.size[3]; break; default: T = P.mod.size[4]; break; }
    if (!T.ok) { bad = true; break; }
    const int M = 32 << si, Nmag = M + 1;
    const float *wN = P.mod.win + N * (N - 1) / 2;
    ooura_forward<ModG>(z, T.oo, [&](int i) {
      const int64_t k0 = 2 * (int64_t)i, k1 = k0 + 1;
      return make_float2(k0 < N ? val(n + k0) * wN[k0] : 0.0f, k1 < N ? val(n + k1) * wN[k1] : 0.0f);
    });
    for (int k = lane; k <= M; k += 64) {                // computeMagnitudes :224-232
      const float2 X = ooura_bin(z, T.oo, k);
      yv[k] = (double)((k == 0 || k == M) ? fabsf(X.x) : (float)sqrt((double)(X.x * X.x + X.y * X.y)));
    }
    ModG::sync();
    if (lane == 0) {
      uu[0] = 0.0; y2[0] = 0.0;
      for (int i = 1; i < Nmag - 1; ++i) {
        const double sg = T.sigma[i];
        const double p = 1.0 / (sg * y2[i - 1] + 2.0);
        y2[i] = (sg - 1.0) * p;
        const double ut = (yv[i + 1] - yv[i]) / T.d1[i] - (yv[i] - yv[i - 1]) / T.d2[i];
        uu[i] = p * (6.0 * ut - sg * uu[i - 1]);
      }
      y2[Nmag - 1] = (0.0 - 0.0 * uu[Nmag - 2]) / (0.0 * y2[Nmag - 2] + 1.0);
      for (int j = Nmag - 2; j >= 0; --j) y2[j] = y2[j] * y2[j + 1] + uu[j];
    }
    ModG::sync();
    for (int i = lane; i < nb; i += 64) {
      const double a = T.co[3 * i], b = 1.0 - a, c = T.co[3 * i + 1], d = T.co[3 * i + 2];
      const int k = T.k[i];
      acc[i] += (float)(a * yv[k] + b * yv[k + 1] + c * y2[k] + d * y2[k + 1]);
    }
    ModG::sync();
    n_spec++;
  }
  for (int i = lane; i < nb; i += 64)
    out[i] = bad ? __int_as_float(0x7fc00000) : (n_spec > 0 ? acc[i] / (float)n_spec : acc[i]);
}

hipError_t launch_funcspec(const FsParams &P, int n_utt, const int *fam_off, const int *fam_want, hipStream_t s) {
  if (n_utt <= 0 || P.n_cols <= 0) return hipSuccess;
  const int groups = (P.n_cols + kColsPerBlock - 1) / kColsPerBlock;
  const dim3 grid((unsigned)(n_utt * groups)), block(kColsPerBlock);
  SMILEHIP_KLAUNCH(fs_stats, grid, block, 0, s, P);
  for (int i = 0; i < P.spec.n_fam; ++i) {
    const int off = fam_off[i], want = fam_want[i];
    switch (P.spec.fam[i]) {
      case SMILEHIP_FAM_EXTREMES: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_EXTREMES>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_MEANS: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_MEANS>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_MOMENTS: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_MOMENTS>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_REGRESSION: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_REGRESSION>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_TIMES: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_TIMES>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_SEGMENTS: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_SEGMENTS>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_LPC: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_LPC>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_PEAKS2: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_PEAKS2>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_ONSET: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_ONSET>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_PEAKS: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_PEAKS>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_CROSSINGS: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_CROSSINGS>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_DCT: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_DCT>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_SAMPLES: SMILEHIP_KLAUNCH(fs_family<SMILEHIP_FAM_SAMPLES>, grid, block, 0, s, P, off, want); break;
      case SMILEHIP_FAM_MODULATION:
        SMILEHIP_KLAUNCH(fs_modulation, dim3((unsigned)(n_utt * P.n_cols)), dim3(64), 0, s, P, off);
        break;
      case SMILEHIP_FAM_PERCENTILES:
        SMILEHIP_KLAUNCH(fs_percentiles_wave, dim3((unsigned)((n_utt * P.n_cols + 3) / 4)), dim3(256), 0, s, P, off, n_utt * P.n_cols);
        if (P.max_rows > kWaveSortMax)                   // some contour may be longer than one wave sorts
          SMILEHIP_KLAUNCH(fs_percentiles, dim3((unsigned)(n_utt * P.n_cols)), dim3(kSortThreads), 0, s, P, off);
        break;
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}

}  // namespace smilehip
