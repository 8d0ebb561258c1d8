// libsmilehip, C ABI part 5: general functionals -- any cFunctionals instance over LLD matrices
// (include/smilehip.h, "general functionals"; kernels in lld_funcspec.hip).
#include "smilehip_internal.hpp"

namespace {

int popc(uint32_t v) { return __builtin_popcount(v); }

// values a family contributes; < 0 + message if the spec cannot be run
int family_count(const smilehip_func_spec &s, int fam) {
  switch (fam) {
    case SMILEHIP_FAM_EXTREMES:
      if (s.ext_mask & ~0xffu) return fail(SMILEHIP_ERR_INVALID, "Extremes: unknown bits in mask 0x%x", s.ext_mask);
      return popc(s.ext_mask);
    case SMILEHIP_FAM_MEANS:
      if (s.means_mask & ~0x1ffffu) return fail(SMILEHIP_ERR_INVALID, "Means: unknown bits in mask 0x%x", s.means_mask);
      return popc(s.means_mask);
    case SMILEHIP_FAM_MOMENTS:
      if (s.mom_mask & ~0x3fu) return fail(SMILEHIP_ERR_INVALID, "Moments: unknown bits in mask 0x%x", s.mom_mask);
      if ((s.mom_mask & 0x20u) && s.mom_stddev_norm != 1 && s.mom_stddev_norm != 2)
        return fail(SMILEHIP_ERR_INVALID, "Moments.stddevNorm must be 1 or 2 when its value is enabled");
      return popc(s.mom_mask);
    case SMILEHIP_FAM_REGRESSION:
      if (s.reg_mask & ~0x3ffffu) return fail(SMILEHIP_ERR_INVALID, "Regression: unknown bits in mask 0x%x", s.reg_mask);
      if (s.reg_norm_coeff < 0 || s.reg_norm_coeff > 2) return fail(SMILEHIP_ERR_INVALID, "Regression.normRegCoeff %d not in 0..2", s.reg_norm_coeff);
      return popc(s.reg_mask);
    case SMILEHIP_FAM_PERCENTILES:
      if (s.pct_mask & ~0x3fu) return fail(SMILEHIP_ERR_INVALID, "Percentiles: unknown bits in mask 0x%x", s.pct_mask);
      if (s.n_pctl < 0 || s.n_pctl > 8 || s.n_range < 0 || s.n_range > 8)
        return fail(SMILEHIP_ERR_INVALID, "Percentiles: at most 8 percentile[] and 8 pctlrange[] entries");
      if (s.n_pctl == 0 && s.n_range > 0) return fail(SMILEHIP_ERR_INVALID, "Percentiles: pctlrange[] without percentile[]");
      for (int i = 0; i < s.n_pctl; ++i)
        if (!(s.pctl[i] >= 0.0 && s.pctl[i] <= 1.0)) return fail(SMILEHIP_ERR_INVALID, "Percentiles: percentile[%d] = %g not in [0, 1]", i, s.pctl[i]);
      for (int i = 0; i < s.n_range; ++i)
        if (s.range_a[i] < 0 || s.range_a[i] >= s.n_pctl || s.range_b[i] < 0 || s.range_b[i] >= s.n_pctl || s.range_a[i] == s.range_b[i])
          return fail(SMILEHIP_ERR_INVALID, "Percentiles: pctlrange[%d] = %d-%d is not a pair of distinct percentile[] indices", i, s.range_a[i], s.range_b[i]);
      if (s.n_quot < 0 || s.n_quot > 8) return fail(SMILEHIP_ERR_INVALID, "Percentiles: at most 8 pctlquotient[] entries");
      if (s.n_pctl == 0 && s.n_quot > 0) return fail(SMILEHIP_ERR_INVALID, "Percentiles: pctlquotient[] without percentile[]");
      for (int i = 0; i < s.n_quot; ++i)
        if (s.quot_a[i] < 0 || s.quot_a[i] >= s.n_pctl || s.quot_b[i] < 0 || s.quot_b[i] >= s.n_pctl)
          return fail(SMILEHIP_ERR_INVALID, "Percentiles: pctlquotient[%d] = %d-%d is not a pair of percentile[] indices", i, s.quot_a[i], s.quot_b[i]);
      return popc(s.pct_mask) + s.n_pctl + s.n_range + s.n_quot;
    case SMILEHIP_FAM_TIMES:
      if (s.times_mask & ~0x1fffu) return fail(SMILEHIP_ERR_INVALID, "Times: unknown bits in mask 0x%x", s.times_mask);
      if (s.n_ul < 0 || s.n_ul > 8 || s.n_dl < 0 || s.n_dl > 8) return fail(SMILEHIP_ERR_INVALID, "Times: at most 8 upleveltime[] and 8 downleveltime[] entries");
      for (int i = 0; i < s.n_ul; ++i)
        if (!(s.ul[i] >= 0.0 && s.ul[i] <= 1.0)) return fail(SMILEHIP_ERR_INVALID, "Times: upleveltime[%d] = %g not in [0, 1]", i, s.ul[i]);
      for (int i = 0; i < s.n_dl; ++i)
        if (!(s.dl[i] >= 0.0 && s.dl[i] <= 1.0)) return fail(SMILEHIP_ERR_INVALID, "Times: downleveltime[%d] = %g not in [0, 1]", i, s.dl[i]);
      return popc(s.times_mask) + s.n_ul + s.n_dl;
    case SMILEHIP_FAM_SEGMENTS:
      if (s.seg_mask & ~0x1fu) return fail(SMILEHIP_ERR_INVALID, "Segments: unknown bits in mask 0x%x", s.seg_mask);
      if (s.seg_algo < SMILEHIP_SEG_RELTH || s.seg_algo > SMILEHIP_SEG_CHX)
        return fail(SMILEHIP_ERR_INVALID, "Segments: unknown segmentationAlgorithm %d", s.seg_algo);
      if (s.seg_max_num < 1) return fail(SMILEHIP_ERR_INVALID, "Segments.maxNumSeg must be >= 1");
      if ((s.seg_algo == SMILEHIP_SEG_DELTA || s.seg_algo == SMILEHIP_SEG_DELTA2) && s.seg_ravg_lng <= 0 && s.seg_max_num < 2)
        return fail(SMILEHIP_ERR_INVALID, "Segments: delta / delt2 without ravgLng need maxNumSeg >= 2 (Nin / (maxNumSeg / 2))");
      if (s.seg_n_thresholds < 0 || s.seg_n_thresholds > 8) return fail(SMILEHIP_ERR_INVALID, "Segments: at most 8 thresholds");
      if (s.seg_min_lng < 1 || s.seg_pause_min_lng < 1) return fail(SMILEHIP_ERR_INVALID, "Segments: segMinLng and pauseMinLng must be >= 1");
      return popc(s.seg_mask);
    case SMILEHIP_FAM_LPC: {
      const int o = s.lpc_order;
      const bool built = (o >= 1 && o <= 8) || o == 10 || o == 12 || o == 16;
      if (!built) return fail(SMILEHIP_ERR_INVALID, "Lpc.order %d is not built (1..8, 10, 12, 16 are)", o);
      if (s.lpc_first < 0 || s.lpc_first >= o) return fail(SMILEHIP_ERR_INVALID, "Lpc.firstCoeff %d not in 0..order-1", s.lpc_first);
      return (s.lpc_gain ? 1 : 0) + (s.lpc_coeffs ? o - s.lpc_first : 0);
    }
    case SMILEHIP_FAM_PEAKS2:
      return popc(s.pk_mask);
    case SMILEHIP_FAM_ONSET:
      if (s.ons_mask & ~0x1fu) return fail(SMILEHIP_ERR_INVALID, "Onset: unknown bits in mask 0x%x", s.ons_mask);
      return popc(s.ons_mask);
    case SMILEHIP_FAM_PEAKS:
      if (s.pko_mask & ~0x1fu) return fail(SMILEHIP_ERR_INVALID, "Peaks: unknown bits in mask 0x%x", s.pko_mask);
      return popc(s.pko_mask);
    case SMILEHIP_FAM_CROSSINGS:
      if (s.crs_mask & ~0x7u) return fail(SMILEHIP_ERR_INVALID, "Crossings: unknown bits in mask 0x%x", s.crs_mask);
      return popc(s.crs_mask);
    case SMILEHIP_FAM_DCT:
      if (s.dct_first < 0 || s.dct_last < s.dct_first || s.dct_last - s.dct_first >= 64)
        return fail(SMILEHIP_ERR_INVALID, "DCT: coefficients %d .. %d (0 <= first <= last, at most 64)", s.dct_first, s.dct_last);
      return s.dct_last - s.dct_first + 1;
    case SMILEHIP_FAM_SAMPLES:
      if (s.n_samples < 1 || s.n_samples > 8) return fail(SMILEHIP_ERR_INVALID, "Samples: 1 .. 8 positions");
      for (int i = 0; i < s.n_samples; ++i)
        if (!(s.sample_pos[i] >= 0.0 && s.sample_pos[i] <= 1.0)) return fail(SMILEHIP_ERR_INVALID, "Samples: samplepos[%d] = %g not in [0, 1]", i, s.sample_pos[i]);
      return s.n_samples;
    case SMILEHIP_FAM_MODULATION:
      // A window of W values is followed by tail windows of down to 2 W / 3 + 1 values; the transforms built are those of 33 .. 1024
      // values (64 .. 1024 points). The reference goes down to 4 points: with W < 49 a tail window would need one of those -- refused
      // here rather than answered with NaN. (A CONTOUR of fewer than 34 values has no window of 33: its bins are NaN, documented in
      // include/smilehip.h; the plugin refuses such rows by name.)
      if (s.mod_win_frames < 49 || s.mod_win_frames > 1024)
        return fail(SMILEHIP_ERR_INVALID, "Modulation: windows of 49 .. 1024 values are built (stftWinSize %d; 0 = the whole contour is not; below 49 the "
                    "tail windows need transforms of fewer than 64 points)", s.mod_win_frames);
      if (s.mod_step_frames < 1) return fail(SMILEHIP_ERR_INVALID, "Modulation: stftWinStep must be >= 1");
      if (s.mod_n_bins < 1 || s.mod_n_bins > 128) return fail(SMILEHIP_ERR_INVALID, "Modulation: 1 .. 128 bins (%d)", s.mod_n_bins);
      if (s.mod_win_func < SMILEHIP_WIN_RECT || s.mod_win_func > SMILEHIP_WIN_LANCZOS || s.mod_win_func == SMILEHIP_WIN_GAUSS)
        return fail(SMILEHIP_ERR_INVALID, "Modulation: window function %d (the reference has rectangle, Hann, Hamming, sine, triangle, Bartlett, Lanczos here)", s.mod_win_func);
      if (!(s.mod_max_freq > s.mod_min_freq) || !(s.mod_min_freq >= 0.0)) return fail(SMILEHIP_ERR_INVALID, "Modulation: frequency axis %g .. %g Hz", s.mod_min_freq, s.mod_max_freq);
      return s.mod_n_bins;
  }
  return fail(SMILEHIP_ERR_INVALID, "unknown functional family %d", fam);
}

int norm_ok(int v) { return v == SMILEHIP_NORM_SEGMENT || v == SMILEHIP_NORM_SECOND || v == SMILEHIP_NORM_FRAME; }

int spec_layout(const smilehip_func_spec *s, int *fam_off, int *fam_want) {
  if (!s) return fail(SMILEHIP_ERR_INVALID, "null functionals spec");
  if (s->n_fam < 1 || s->n_fam > 12) return fail(SMILEHIP_ERR_INVALID, "functionalsEnabled: %d families (1..12)", s->n_fam);
  if (s->non_zero_functs < 0 || s->non_zero_functs > 2) return fail(SMILEHIP_ERR_INVALID, "nonZeroFuncts %d not in 0..2", s->non_zero_functs);
  if (!norm_ok(s->ext_norm) || !norm_ok(s->means_norm) || !norm_ok(s->times_norm) || !norm_ok(s->seg_norm) || !norm_ok(s->pk_norm) ||
      !norm_ok(s->reg_centroid_norm) || !norm_ok(s->ons_norm) || !norm_ok(s->pko_norm))
    return fail(SMILEHIP_ERR_INVALID, "time norm must be 0 (segment), 1 (second) or 2 (frame)");
  if (!(s->period > 0.0)) return fail(SMILEHIP_ERR_INVALID, "functionals spec: period must be > 0");
  int n = 0;
  for (int i = 0; i < s->n_fam; ++i) {
    const int c = family_count(*s, s->fam[i]);
    if (c < 0) return c;
    if (fam_off) fam_off[i] = n;
    if (fam_want) fam_want[i] = c;
    n += c;
  }
  if (n < 1) return fail(SMILEHIP_ERR_INVALID, "functionals spec enables no value");
  return n;
}

size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// scratch of one launch set (scratch_rows x n_cols cells + per-utterance stats): its size, and its carving at `base`
struct FsNeed { bool nz, alive, sorted; int64_t rows; int n_utt; };
size_t fs_bytes(const FsParams &P, const FsNeed &n) {
  const size_t cells = size_t(n.rows) * size_t(P.n_cols), st = size_t(n.n_utt) * size_t(P.n_cols);
  return (n.nz ? align256(cells * 4) : 0) + (n.alive ? align256(cells) : 0) + (n.sorted ? align256(cells * 8) : 0) +
         4 * align256(st * 4);
}
void fs_carve(char *b, FsParams &P, const FsNeed &n) {
  const size_t cells = size_t(n.rows) * size_t(P.n_cols), st = size_t(n.n_utt) * size_t(P.n_cols);
  size_t off = 0;
  P.nz = n.nz ? reinterpret_cast<float *>(b + off) : nullptr; off += n.nz ? align256(cells * 4) : 0;
  P.alive = n.alive ? reinterpret_cast<unsigned char *>(b + off) : nullptr; off += n.alive ? align256(cells) : 0;
  P.sorted = n.sorted ? reinterpret_cast<float *>(b + off) : nullptr; off += n.sorted ? align256(cells * 8) : 0;
  P.st_min = reinterpret_cast<float *>(b + off);
  P.st_max = reinterpret_cast<float *>(b + off + align256(st * 4));
  P.st_mean = reinterpret_cast<float *>(b + off + 2 * align256(st * 4));
  P.st_n = reinterpret_cast<int32_t *>(b + off + 3 * align256(st * 4));
}
int fs_reserve(smilehip_context *ctx, size_t bytes, hipStream_t stream) {
  if (bytes <= ctx->fs_cap) return SMILEHIP_OK;
  // earlier launches may still read the old buffer
  HIP_TRY(hipStreamSynchronize(stream));
  HIP_TRY(hipDeviceSynchronize());
  if (ctx->fs_scratch) (void)hipFree(ctx->fs_scratch);
  ctx->fs_scratch = nullptr;
  ctx->fs_cap = 0;
  const size_t want = bytes + bytes / 4;
  HIP_TRY(hipMalloc(&ctx->fs_scratch, want));
  ctx->fs_cap = want;
  return SMILEHIP_OK;
}

bool has_family(const smilehip_func_spec &s, int fam) {
  for (int i = 0; i < s.n_fam; ++i)
    if (s.fam[i] == fam) return true;
  return false;
}

FsNeed spec_need(const FsParams &P, int n_utt, int64_t scratch_rows, int64_t max_rows) {
  FsNeed n;
  n.nz = P.spec.non_zero_functs != 0;
  n.alive = has_family(P.spec, SMILEHIP_FAM_PEAKS2);
  n.sorted = has_family(P.spec, SMILEHIP_FAM_PERCENTILES) && max_rows > fs_sort_lds_rows();
  n.rows = scratch_rows;
  n.n_utt = n_utt;
  return n;
}

int launch_spec(FsParams &P, int n_utt, hipStream_t stream) {
  int fam_off[12], fam_want[12];
  const int per = spec_layout(&P.spec, fam_off, fam_want);
  if (per < 0) return per;
  P.per = per;
  hipError_t e = launch_funcspec(P, n_utt, fam_off, fam_want, stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "functionals kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

// ---- Modulation: the device tables of one option set (ModTables, lld_params.hpp), kept in the context
}  // namespace
struct ModEntry {                                       // one option set's tables
  int32_t ki[4] = {0, 0, 0, 0};
  double kd[3] = {0, 0, 0};
  DevBuf<float> d_win;
  OouraDev oo[5];
  DevBuf<double> d_dbl;
  DevBuf<int32_t> d_k;
  ModTables tabs{};
};
// The context's cache: a few option sets side by side (callers alternate between instances: tables of a set in use are never
// rebuilt under a launch that reads them), most recently used first, behind a lock (a context may be shared by threads).
struct ModCache {
  std::mutex mu;
  std::vector<std::unique_ptr<ModEntry>> sets;
  // sets that left the cache: their tables may still be read by a launch another thread has not enqueued yet (it copied the
  // pointers out under the lock and launches after releasing it), so they are kept until many of them have piled up -- by then
  // every launch that could name them is long enqueued -- and freed behind a device synchronisation
  std::vector<std::unique_ptr<ModEntry>> retired;
};
void mod_cache_free(ModCache *m) { delete m; }
namespace {
constexpr size_t kModCacheSets = 8;
int mod_prepare(smilehip_context *ctx, const smilehip_func_spec &s, ModTables &out) {
  static std::mutex create_mu;
  {
    std::lock_guard<std::mutex> g(create_mu);
    if (!ctx->mod) ctx->mod = new (std::nothrow) ModCache();
  }
  ModCache *cache = ctx->mod;
  if (!cache) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  std::lock_guard<std::mutex> lock(cache->mu);
  const int32_t ki[4] = {s.mod_win_frames, s.mod_n_bins, s.mod_win_func, 0};
  const double kd[3] = {s.period, s.mod_min_freq, s.mod_max_freq};
  for (size_t i = 0; i < cache->sets.size(); ++i) {
    ModEntry *e = cache->sets[i].get();
    if (e->tabs.win && !std::memcmp(ki, e->ki, sizeof(ki)) && !std::memcmp(kd, e->kd, sizeof(kd))) {
      out = e->tabs;
      if (i) std::rotate(cache->sets.begin(), cache->sets.begin() + i, cache->sets.begin() + i + 1);
      return SMILEHIP_OK;
    }
  }
  if (cache->sets.size() >= kModCacheSets) {             // the least recently used set leaves the cache (ModCache::retired)
    cache->retired.push_back(std::move(cache->sets.back()));
    cache->sets.pop_back();
    if (cache->retired.size() > 64) {
      HIP_TRY(hipDeviceSynchronize());
      cache->retired.clear();
    }
  }
  std::unique_ptr<ModEntry> fresh(new (std::nothrow) ModEntry());
  ModEntry *m = fresh.get();
  if (!m) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  const int W = s.mod_win_frames, nb = s.mod_n_bins;
  // the window function of every length a window can have (cSmileUtilWindowedMagnitudeSpectrum::allocateWinFunc: the values of
  // smileDsp_win* in double, stored as FLOAT_DMEM), lengths below 33 never reach the transform
  std::vector<float> win(size_t(W) * size_t(W + 1) / 2, 0.0f), w1;
  smilehip_lld_config wc;
  std::memset(&wc, 0, sizeof(wc));
  wc.win_func = s.mod_win_func; wc.win_sigma = 0.4; wc.win_gain = 1.0;
  for (int N = 33; N <= W; ++N) {
    int rc = make_window(wc, N, w1);
    if (rc) return fail(rc, "Modulation: window function %d", s.mod_win_func);
    std::memcpy(win.data() + size_t(N) * size_t(N - 1) / 2, w1.data(), size_t(N) * sizeof(float));
  }
  int rc = m->d_win.upload(win);
  if (rc) return rc;
  std::vector<double> dbl;
  std::vector<int32_t> kk;
  size_t off_d[5][2] = {}, off_k[5] = {};
  int ok[5] = {0, 0, 0, 0, 0};
  const double T = (double)(float)s.period;             // the constructor's T is the FLOAT_DMEM period (functionalModulation.cpp:485)
  for (int si = 0; si < 5; ++si) {
    const int n = 64 << si;
    if (si > 0 && (64 << (si - 1)) >= W) break;                               // no window reaches this length (n / 2 >= W)
    if ((rc = m->oo[si].build(n, true))) return rc;
    const int Nmag = n / 2 + 1;
    const double dfm = (T == 0.0) ? 0.0 : 1.0 / (T * (double)n);
    std::vector<double> x((size_t)Nmag);
    for (int i = 0; i < Nmag; ++i) x[i] = (double)i * dfm;
    off_d[si][0] = dbl.size();
    dbl.resize(dbl.size() + 3 * (size_t)Nmag, 0.0);
    double *sigma = dbl.data() + off_d[si][0], *d1 = sigma + Nmag, *d2 = d1 + Nmag;
    for (int i = 1; i < Nmag - 1; ++i) {                                      // smileMath_cspline_init, smileUtilSpline.c:138-153
      sigma[i] = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
      d1[i] = (x[i + 1] - x[i]) * (x[i + 1] - x[i - 1]);
      d2[i] = (x[i] - x[i - 1]) * (x[i + 1] - x[i - 1]);
    }
    off_d[si][1] = dbl.size();
    dbl.resize(dbl.size() + 3 * (size_t)nb, 0.0);
    off_k[si] = kk.size();
    kk.resize(kk.size() + (size_t)nb, 0);
    double *co = dbl.data() + off_d[si][1];
    int32_t *kp = kk.data() + off_k[si];
    const double dF = (s.mod_max_freq - s.mod_min_freq) / (double)nb;         // functionalModulation.cpp:311: over Nout
    bool good = !(s.mod_min_freq < x[0] || s.mod_min_freq + (double)(nb - 1) * dF > x[Nmag - 1]);
    int hi = 1;
    for (int i = 0; i < nb && good; ++i) {                                    // smileMath_csplint_init, :295-342
      const double xt = s.mod_min_freq + (double)i * dF;
      while (hi < Nmag && x[hi] < xt) hi++;
      if (hi == Nmag) { good = false; break; }
      const int lo = hi - 1;
      const double range = x[hi] - x[lo];
      if (range == 0.0) { good = false; break; }
      const double a = (x[hi] - xt) / range, bq = 1.0 - a, r2 = range * range / 6.0;
      kp[i] = lo;
      co[3 * i] = a;
      co[3 * i + 1] = (a * a * a - a) * r2;
      co[3 * i + 2] = (bq * bq * bq - bq) * r2;
    }
    ok[si] = good ? 1 : 0;
  }
  if ((rc = m->d_dbl.upload(dbl)) || (rc = m->d_k.upload(kk))) return rc;
  ModTables tb{};
  tb.win = m->d_win.p;
  for (int si = 0; si < 5; ++si) {
    if (!m->oo[si].d_tw.p) continue;
    const int Nmag = (64 << si) / 2 + 1;
    tb.size[si].oo = m->oo[si].tab();
    tb.size[si].sigma = m->d_dbl.p + off_d[si][0];
    tb.size[si].d1 = tb.size[si].sigma + Nmag;
    tb.size[si].d2 = tb.size[si].d1 + Nmag;
    tb.size[si].co = m->d_dbl.p + off_d[si][1];
    tb.size[si].k = m->d_k.p + off_k[si];
    tb.size[si].ok = ok[si];
  }
  m->tabs = tb;
  std::memcpy(m->ki, ki, sizeof(ki));
  std::memcpy(m->kd, kd, sizeof(kd));
  out = tb;
  cache->sets.insert(cache->sets.begin(), std::move(fresh));
  return SMILEHIP_OK;
}

int run_spec(smilehip_context *ctx, FsParams &P, int n_utt, int64_t scratch_rows, int64_t max_rows, hipStream_t stream) {
  const int per = spec_layout(&P.spec, nullptr, nullptr);
  if (per < 0) return per;
  const FsNeed need = spec_need(P, n_utt, scratch_rows, max_rows);
  int rc = fs_reserve(ctx, fs_bytes(P, need), stream);
  if (rc) return rc;
  fs_carve(static_cast<char *>(ctx->fs_scratch), P, need);
  P.max_rows = max_rows;
  std::memset(&P.mod, 0, sizeof(P.mod));
  if (has_family(P.spec, SMILEHIP_FAM_MODULATION) && (rc = mod_prepare(ctx, P.spec, P.mod))) return rc;
  return launch_spec(P, n_utt, stream);
}

}  // namespace

int smilehip_funcspec_from_mask(uint32_t mask, double period, smilehip_func_spec *s) {
  if (!s || (mask & ~SMILEHIP_FUNC_ALL) || !mask) return fail(SMILEHIP_ERR_INVALID, "invalid functionals mask 0x%x", mask);
  std::memset(s, 0, sizeof(*s));
  s->period = period;
  s->ext_norm = SMILEHIP_NORM_FRAME;                      // positions in frames (Extremes.norm = frame)
  s->means_norm = s->times_norm = s->seg_norm = s->pk_norm = SMILEHIP_NORM_FRAME;
  s->reg_centroid_norm = SMILEHIP_NORM_SEGMENT;
  s->seg_max_num = 20; s->seg_min_lng = 3; s->seg_pause_min_lng = 2; s->lpc_order = 5;
  s->ext_mask = mask & 0xffu;
  s->reg_mask = (mask >> 8) & 0xfu;                       // linregc1 linregc2 linregerrA linregerrQ
  s->reg_old_buggy_qerr = 1;                              // the option's default; no quadratic value is enabled
  s->mom_mask = (mask >> 12) & 0x1fu;                     // variance stddev skewness kurtosis amean
  if (s->ext_mask) s->fam[s->n_fam++] = SMILEHIP_FAM_EXTREMES;
  if (s->reg_mask) s->fam[s->n_fam++] = SMILEHIP_FAM_REGRESSION;
  if (s->mom_mask) s->fam[s->n_fam++] = SMILEHIP_FAM_MOMENTS;
  return SMILEHIP_OK;
}

extern "C" int smilehip_funcspec_count(const smilehip_func_spec *spec) { return spec_layout(spec, nullptr, nullptr); }

// config/compare16/ComParE_2016_core.func.conf.inc
extern "C" int smilehip_funcspec_compare16(const char *instance, smilehip_func_spec *s) {
  if (!instance || !s) return fail(SMILEHIP_ERR_INVALID, "smilehip_funcspec_compare16: null argument");
  std::memset(s, 0, sizeof(*s));
  s->period = 0.01;
  const std::string k(instance);
  auto fams = [&](std::initializer_list<int> l) {
    s->n_fam = 0;
    for (int f : l) s->fam[s->n_fam++] = f;
  };
  auto extremes = [&] { s->ext_mask = (1u << 2) | (1u << 3) | (1u << 4); s->ext_norm = SMILEHIP_NORM_SEGMENT; };       // range maxPos minPos; masterTimeNorm
  auto percentiles = [&] {
    s->pct_mask = 0x3f; s->pct_interp = 1; s->n_pctl = 2; s->pctl[0] = 0.01; s->pctl[1] = 0.99;
    s->n_range = 1; s->range_a[0] = 0; s->range_b[0] = 1;
  };
  auto moments = [&] { s->mom_mask = (1u << 1) | (1u << 2) | (1u << 3); s->mom_ratio_limit = 1; };
  auto times = [&] {
    s->times_mask = (1u << 0) | (1u << 2) | (1u << 4) | (1u << 6) | (1u << 8) | (1u << 10);
    s->times_norm = SMILEHIP_NORM_SEGMENT;
  };
  auto lpc = [&] { s->lpc_gain = 1; s->lpc_coeffs = 1; s->lpc_first = 0; s->lpc_order = 5; };
  auto segments = [&](int algo) {
    s->seg_mask = 0x1e; s->seg_norm = SMILEHIP_NORM_SECOND; s->seg_algo = algo; s->seg_max_num = 100;
    s->seg_min_lng = 3; s->seg_auto_min_lng = 1; s->seg_pause_min_lng = 2;
    if (algo == SMILEHIP_SEG_RELTH) { s->seg_n_thresholds = 2; s->seg_thresholds[0] = 0.25f; s->seg_thresholds[1] = 0.75f; }
  };
  auto regression = [&](int norm_coeff) {
    s->reg_mask = (1u << 0) | (1u << 1) | (1u << 3) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 8) | (1u << 9);
    s->reg_centroid_norm = SMILEHIP_NORM_SEGMENT; s->reg_norm_coeff = norm_coeff;
    s->reg_norm_inputs = s->reg_centroid_abs = s->reg_centroid_limit = s->reg_ratio_limit = 1;
  };
  auto peaks2 = [&] {
    s->pk_mask = (1u << 1) | (1u << 3) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 7) | (1u << 8) | (1u << 14) | (1u << 22) |
                 (1u << 25) | (1u << 26) | (1u << 29);
    s->pk_norm = SMILEHIP_NORM_SECOND; s->pk_ratio_limit = 1; s->pk_rel_thresh = 0.1f;
  };
  // norms a family does not use still have to be valid
  s->means_norm = SMILEHIP_NORM_FRAME; s->seg_norm = SMILEHIP_NORM_SEGMENT; s->pk_norm = SMILEHIP_NORM_FRAME;
  if (k == "A" || k == "B") {
    fams({SMILEHIP_FAM_EXTREMES, SMILEHIP_FAM_PERCENTILES, SMILEHIP_FAM_MOMENTS, SMILEHIP_FAM_SEGMENTS, SMILEHIP_FAM_TIMES, SMILEHIP_FAM_LPC});
    extremes(); percentiles(); moments(); segments(SMILEHIP_SEG_RELTH); times(); lpc();
  } else if (k == "F0") {
    fams({SMILEHIP_FAM_MEANS, SMILEHIP_FAM_SEGMENTS});
    s->means_mask = 1u << 7; s->means_norm = SMILEHIP_NORM_SEGMENT;
    segments(SMILEHIP_SEG_NONX);
    s->seg_x = 0.0f;
  } else if (k == "Nz") {
    fams({SMILEHIP_FAM_MEANS, SMILEHIP_FAM_EXTREMES, SMILEHIP_FAM_REGRESSION, SMILEHIP_FAM_PERCENTILES, SMILEHIP_FAM_MOMENTS,
          SMILEHIP_FAM_TIMES, SMILEHIP_FAM_LPC});
    s->non_zero_functs = 1;
    s->means_mask = (1u << 0) | (1u << 8) | (1u << 9) | (1u << 15); s->means_norm = SMILEHIP_NORM_FRAME;
    extremes(); regression(0); percentiles(); moments(); times(); lpc();
  } else if (k == "LLD") {
    fams({SMILEHIP_FAM_MEANS, SMILEHIP_FAM_PEAKS2, SMILEHIP_FAM_REGRESSION});
    s->means_mask = (1u << 0) | (1u << 8) | (1u << 15);
    peaks2(); regression(2);
  } else if (k == "Delta") {
    fams({SMILEHIP_FAM_MEANS, SMILEHIP_FAM_PEAKS2});
    s->means_mask = (1u << 8) | (1u << 9) | (1u << 15);
    peaks2();
  } else {
    return fail(SMILEHIP_ERR_INVALID, "unknown ComParE_2016 functionals instance '%s' (A, B, F0, Nz, LLD, Delta)", instance);
  }
  return SMILEHIP_OK;
}

// config/is09-13/IS13_ComParE_core.func.conf.inc differs from ComParE_2016_core.func.conf.inc in these options only
extern "C" int smilehip_funcspec_is13_compare(const char *instance, smilehip_func_spec *s) {
  int rc = smilehip_funcspec_compare16(instance, s);
  if (rc) return rc;
  s->mom_ratio_limit = 0;
  s->reg_centroid_abs = s->reg_centroid_limit = s->reg_ratio_limit = s->reg_norm_inputs = 0;
  s->reg_norm_coeff = 0;
  s->pk_ratio_limit = 0;
  return SMILEHIP_OK;
}

extern "C" int smilehip_funcspec_matrix(smilehip_context *ctx, const smilehip_func_spec *spec, const float *d_x, int64_t ld_x,
                                        int64_t rows, int32_t cols, float *d_out, void *stream) {
  if (!ctx || !spec || rows < 1 || cols < 1 || ld_x < cols || !d_x || !d_out)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_funcspec_matrix: bad argument");
  FsParams P;
  std::memset(&P, 0, sizeof(P));
  P.spec = *spec;
  P.x = d_x;
  P.ld_x = ld_x;
  P.col_first = 0;
  P.n_cols = cols;
  P.single_rows = rows;
  P.out = d_out;
  const int per = spec_layout(spec, nullptr, nullptr);
  if (per < 0) return per;
  P.ld_out = (int64_t)cols * per;
  return run_spec(ctx, P, 1, rows, rows, (hipStream_t)stream);
}

extern "C" int smilehip_batch_funcspec(smilehip_plan *plan, smilehip_batch *b, const smilehip_func_spec *spec, const float *d_lld,
                                       int64_t ld_lld, int32_t col_first, int32_t n_cols, int32_t rows_cut, const float *d_extra,
                                       int64_t ld_extra, float *d_func, int64_t ld_func, void *stream) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_funcspec: plan/batch mismatch");
  if (!plan->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "smilehip_batch_funcspec: host-only plan");
  const int per = spec_layout(spec, nullptr, nullptr);
  if (per < 0) return per;
  if (col_first < 0 || n_cols < 1 || col_first + n_cols > ld_lld)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_funcspec: columns [%d, %d) outside the matrix (ld %lld)", col_first,
                col_first + n_cols, (long long)ld_lld);
  if (rows_cut < 0) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_funcspec: rows_cut must be >= 0");
  if (ld_func < (int64_t)n_cols * per)
    return fail(SMILEHIP_ERR_INVALID, "ld_func %lld < %d values per utterance", (long long)ld_func, n_cols * per);
  if (d_extra && ld_extra < n_cols) return fail(SMILEHIP_ERR_INVALID, "ld_extra %lld < n_cols %d", (long long)ld_extra, n_cols);
  if (!d_func || (!d_lld && b->total_rows > 0)) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_funcspec: null device pointer");
  if (b->n_utt == 0) return SMILEHIP_OK;
  int64_t max_rows = 0;
  for (int u = 0; u < b->n_utt; ++u) max_rows = std::max(max_rows, b->h_row_off[u + 1] - b->h_row_off[u] + 1);
  FsParams P;
  std::memset(&P, 0, sizeof(P));
  P.spec = *spec;
  P.x = d_lld;
  P.ld_x = ld_lld;
  P.col_first = col_first;
  P.n_cols = n_cols;
  P.row_off = b->d_row_off.p;
  P.single_rows = -1;
  P.rows_cut = rows_cut;
  P.extra = d_extra;
  P.ld_extra = ld_extra;
  P.out = d_func;
  P.ld_out = ld_func;
  return run_spec(plan->ctx, P, b->n_utt, b->total_rows + b->n_utt, max_rows, (hipStream_t)stream);
}

// ---- the whole functionals level of ComParE_2016 (6373 values per utterance)
namespace {
struct Part { const char *inst; int col_first, n_cols, rows_cut; bool pending; int extra_off; };   // extra_off < 0: no extra row
// writer levels in the order [functionals] concatenates them (ComParE_2016.conf); row rules measured against the
// binary (tests/test_oracle_pin_funcspec.py): T = rows - 1
const Part kCompare16Parts[] = {
    {"A", 6, 4, 3, false, -1},   {"A", 71, 4, 3, false, -1},       // lldA_smo ; lldA_smo_de        T-2
    {"B", 10, 55, 0, false, 0},  {"B", 75, 55, 0, false, 55},      // lldB_smo ; lldB_smo_de        T+2 (extra row)
    {"Nz", 0, 6, 3, true, -1},   {"Nz", 65, 6, 3, true, -1},       // lld_nzsmo ; lld_nzsmo_de      T-P-2
    {"F0", 0, 1, 1, true, -1},                                     // lld_f0_nzsmo                  T-P
    {"LLD", 6, 59, 1, false, -1},                                  // lldA_smo ; lldB_smo           T
    {"Delta", 71, 59, 3, false, -1},                               // lldA_smo_de ; lldB_smo_de     T-2
};
}  // namespace

extern "C" int smilehip_functionals_compare16_count(void) { return 6373; }

extern "C" int smilehip_batch_compare_b_extra(smilehip_batch *b, const float **d_extra) {
  if (!b || !d_extra) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_compare_b_extra: null argument");
  if (!b->d_b_extra.p) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_compare_b_extra: not a ComParE chain batch");
  *d_extra = b->d_b_extra.p;
  return SMILEHIP_OK;
}

static int functionals_compare_level(smilehip_plan *plan, smilehip_batch *b, const float *d_lld, int64_t ld_lld, float *d_func,
                                     int64_t ld_func, void *stream, bool is13);

extern "C" int smilehip_batch_functionals_compare16(smilehip_plan *plan, smilehip_batch *b, const float *d_lld, int64_t ld_lld,
                                                    float *d_func, int64_t ld_func, void *stream) {
  return functionals_compare_level(plan, b, d_lld, ld_lld, d_func, ld_func, stream, false);
}

extern "C" int smilehip_batch_functionals_is13_compare(smilehip_plan *plan, smilehip_batch *b, const float *d_lld, int64_t ld_lld,
                                                       float *d_func, int64_t ld_func, void *stream) {
  return functionals_compare_level(plan, b, d_lld, ld_lld, d_func, ld_func, stream, true);
}

static int functionals_compare_level(smilehip_plan *plan, smilehip_batch *b, const float *d_lld, int64_t ld_lld, float *d_func,
                                     int64_t ld_func, void *stream, bool is13) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals_compare16: plan/batch mismatch");
  if (plan->cfg.chain_kind != SMILEHIP_CHAIN_COMPARE)
    return fail(SMILEHIP_ERR_INVALID, "the ComParE_2016 functionals are defined for the whole-level chain (smilehip_config_compare16)");
  if (!plan->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "smilehip_batch_functionals_compare16: host-only plan");
  if (ld_lld < 130 || ld_func < 6373) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals_compare16: ld_lld >= 130 and ld_func >= 6373 required");
  if (!d_func || (!d_lld && b->total_rows > 0)) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals_compare16: null device pointer");
  if (b->n_utt == 0) return SMILEHIP_OK;
  if (!b->f0_batch || !b->f0_batch->d_pending.p || !b->d_b_extra.p)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals_compare16: run smilehip_lld_run on this batch first");
  int64_t max_rows = 0;
  for (int u = 0; u < b->n_utt; ++u) max_rows = std::max(max_rows, b->h_row_off[u + 1] - b->h_row_off[u] + 1);
  // The nine launch sets are independent (own output columns, own scratch slice) and each is latency-bound with about
  // one wave per SIMD: they run side by side on the context's auxiliary streams, forked from and joined to `stream`.
  constexpr int kParts = sizeof(kCompare16Parts) / sizeof(kCompare16Parts[0]);
  FsParams Ps[kParts];
  FsNeed needs[kParts];
  size_t offs[kParts], total = 0;
  int off = 0;
  for (int i = 0; i < kParts; ++i) {
    const Part &part = kCompare16Parts[i];
    FsParams &P = Ps[i];
    std::memset(&P, 0, sizeof(P));
    int rc = is13 ? smilehip_funcspec_is13_compare(part.inst, &P.spec) : smilehip_funcspec_compare16(part.inst, &P.spec);
    if (rc) return rc;
    const int per = spec_layout(&P.spec, nullptr, nullptr);
    if (per < 0) return per;
    P.x = d_lld;
    P.ld_x = ld_lld;
    P.col_first = part.col_first;
    P.n_cols = part.n_cols;
    P.row_off = b->d_row_off.p;
    P.single_rows = -1;
    P.rows_cut = part.rows_cut;
    P.pending = part.pending ? b->f0_batch->d_pending.p : nullptr;
    if (part.extra_off >= 0) {
      P.extra = b->d_b_extra.p + part.extra_off;
      P.ld_extra = 110;
    }
    P.out = d_func + off;
    P.ld_out = ld_func;
    needs[i] = spec_need(P, b->n_utt, b->total_rows + b->n_utt, max_rows);
    offs[i] = total;
    total += fs_bytes(P, needs[i]);
    off += per * part.n_cols;
  }
  hipStream_t main = (hipStream_t)stream;
  int rc = fs_reserve(plan->ctx, total, main);
  if (rc) return rc;
  smilehip_context *ctx = plan->ctx;
  if (!ctx->fs_streams_ready) {
    for (int k = 0; k < smilehip_context::kFsStreams; ++k) {
      HIP_TRY(hipStreamCreateWithFlags(&ctx->fs_stream[k], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&ctx->fs_done[k], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&ctx->fs_fork, hipEventDisableTiming));
    ctx->fs_streams_ready = true;
  }
  HIP_TRY(hipEventRecord(ctx->fs_fork, main));
  for (int k = 0; k < smilehip_context::kFsStreams; ++k) HIP_TRY(hipStreamWaitEvent(ctx->fs_stream[k], ctx->fs_fork, 0));
  // longest sets first, spread over main + auxiliary streams
  static const int order[kParts] = {7, 8, 2, 3, 4, 5, 0, 1, 6};
  for (int j = 0; j < kParts; ++j) {
    const int i = order[j];
    const int lane = j % (smilehip_context::kFsStreams + 1);
    hipStream_t st = lane == 0 ? main : ctx->fs_stream[lane - 1];
    fs_carve(static_cast<char *>(ctx->fs_scratch) + offs[i], Ps[i], needs[i]);
    Ps[i].max_rows = max_rows;
    rc = launch_spec(Ps[i], b->n_utt, st);
    if (rc) return rc;
  }
  for (int k = 0; k < smilehip_context::kFsStreams; ++k) {
    HIP_TRY(hipEventRecord(ctx->fs_done[k], ctx->fs_stream[k]));
    HIP_TRY(hipStreamWaitEvent(main, ctx->fs_done[k], 0));
  }
  if (off != 6373) return fail(SMILEHIP_ERR_INVALID, "internal: ComParE_2016 functionals layout adds up to %d", off);
  return SMILEHIP_OK;
}

// ---- eGeMAPSv02: the functionals level (88 values per utterance)
// config/gemaps/v01b/GeMAPSv01b_core.func.conf.inc + config/egemaps/v02/eGeMAPSv02_core.func.conf.inc
extern "C" int smilehip_funcspec_egemaps(const char *instance, smilehip_func_spec *s) {
  if (!instance || !s) return fail(SMILEHIP_ERR_INVALID, "smilehip_funcspec_egemaps: null argument");
  std::memset(s, 0, sizeof(*s));
  s->period = 0.01;
  s->ext_norm = s->means_norm = s->times_norm = s->seg_norm = s->pk_norm = s->reg_centroid_norm = SMILEHIP_NORM_SEGMENT;
  s->seg_max_num = 20; s->seg_min_lng = 3; s->seg_pause_min_lng = 2; s->lpc_order = 5;
  const std::string k(instance);
  if (k == "F0" || k == "Loudness") {
    // Moments amean + stddevNorm (= 2); Percentiles 20 / 50 / 80 + range 0-2, interpolated; Peaks2 mean / stddev of the rising
    // and falling slopes, in seconds, relThresh 0.1, doRatioLimit = 0; nonZeroFuncts = 1 for F0
    s->n_fam = 3; s->fam[0] = SMILEHIP_FAM_MOMENTS; s->fam[1] = SMILEHIP_FAM_PERCENTILES; s->fam[2] = SMILEHIP_FAM_PEAKS2;
    s->non_zero_functs = (k == "F0") ? 1 : 0;
    s->mom_mask = (1u << 4) | (1u << 5); s->mom_stddev_norm = 2;
    s->pct_interp = 1; s->n_pctl = 3; s->pctl[0] = 0.20; s->pctl[1] = 0.50; s->pctl[2] = 0.80;
    s->n_range = 1; s->range_a[0] = 0; s->range_b[0] = 2;
    s->pk_mask = (1u << 22) | (1u << 25) | (1u << 26) | (1u << 29);
    s->pk_norm = SMILEHIP_NORM_SECOND; s->pk_rel_thresh = 0.1f;
  } else if (k == "MVZ" || k == "MVV") {
    s->n_fam = 1; s->fam[0] = SMILEHIP_FAM_MOMENTS;
    s->non_zero_functs = (k == "MVV") ? 1 : 0;
    s->mom_mask = (1u << 4) | (1u << 5); s->mom_stddev_norm = 2;
  } else if (k == "MU") {
    s->n_fam = 1; s->fam[0] = SMILEHIP_FAM_MOMENTS; s->non_zero_functs = 1; s->mom_mask = 1u << 4;
  } else if (k == "numPeaks") {
    s->n_fam = 1; s->fam[0] = SMILEHIP_FAM_PEAKS2; s->pk_mask = 1u; s->pk_norm = SMILEHIP_NORM_SECOND; s->pk_rel_thresh = 0.1f;
    s->pk_ratio_limit = 1;
  } else if (k == "segF0" || k == "segF0pause") {
    const bool pause = k == "segF0pause";
    s->n_fam = 1; s->fam[0] = SMILEHIP_FAM_SEGMENTS;
    s->seg_mask = pause ? ((1u << 1) | (1u << 4)) : ((1u << 0) | (1u << 1) | (1u << 4));
    s->seg_norm = SMILEHIP_NORM_SECOND; s->seg_algo = pause ? SMILEHIP_SEG_EQX : SMILEHIP_SEG_NONX; s->seg_max_num = 1000;
    s->seg_min_lng = 3; s->seg_auto_min_lng = 1; s->seg_pause_min_lng = 2; s->seg_x = 0.0f;
  } else if (k == "leq") {
    s->n_fam = 1; s->fam[0] = SMILEHIP_FAM_MEANS; s->means_mask = 1u;
  } else {
    return fail(SMILEHIP_ERR_INVALID, "unknown eGeMAPSv02 functionals instance '%s' (F0, Loudness, MVZ, MVV, MU, numPeaks, segF0, "
                "segF0pause, leq)", instance);
  }
  return SMILEHIP_OK;
}

extern "C" int smilehip_functionals_egemaps_count(void) { return 88; }

namespace {
// columns of the batch's func_in matrix (lld_params.hpp, GemapsParams::func_in); pend: 0 none, 1 the Viterbi smoother's
// undecided frames P, 2 P if P < T60 else 0. Rows of func_in = T20 + 1 = T60 + 5.
struct EgPart { const char *inst; int col_first, n_cols, rows_cut, pend; };
const EgPart kEgemapsParts[] = {
    {"F0", 6, 1, 5, 1},          // gemapsv01b_lld_single_logF0_smo                                max(1, T60 - P)
    {"Loudness", 0, 1, 1, 0},    // gemapsv01b_loudness_smo                                        T20
    {"MVZ", 1, 5, 1, 0},         // egemapsv02_lldSetNoF0AndLoudnessZ_smo                          T20
    {"MVV", 7, 23, 5, 2},        // egemapsv02_lldSetNoF0AndLoudnessNz_smo; ..lldSetSpectralNz_smo T60 - P, T60 if P = T60
    {"MU", 30, 5, 5, 1},         // egemapsv02_lldSetSpectralZ_smo                                 max(1, T60 - P)
    {"numPeaks", 0, 1, 1, 0},    // gemapsv01b_temporalSet: loudnessPeaksPerSec
    {"segF0", 6, 1, 5, 1},       //   VoicedSegmentsPerSec, MeanVoicedSegmentLengthSec, StddevVoicedSegmentLengthSec
    {"segF0pause", 6, 1, 5, 1},  //   MeanUnvoicedSegmentLength, StddevUnvoicedSegmentLength
    {"leq", 35, 1, 1, 0},        // egemapsv02_leqLin (amean of energy2), cVectorOperation dBp applied afterwards
};
}  // namespace

extern "C" int smilehip_batch_functionals_egemaps(smilehip_plan *plan, smilehip_batch *b, float *d_func, int64_t ld_func, void *stream) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals_egemaps: plan/batch mismatch");
  if (plan->cfg.chain_kind != SMILEHIP_CHAIN_EGEMAPS)
    return fail(SMILEHIP_ERR_INVALID, "the eGeMAPSv02 functionals are defined for smilehip_config_egemapsv02 plans");
  if (!plan->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "smilehip_batch_functionals_egemaps: host-only plan");
  if (ld_func < 88) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals_egemaps: ld_func >= 88 required");
  if (!d_func) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals_egemaps: null device pointer");
  if (b->n_utt == 0) return SMILEHIP_OK;
  if (!b->gm_ran || !b->f0_batch) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals_egemaps: run smilehip_lld_run on this batch first");
  hipStream_t main = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(d_func, 0, size_t(b->n_utt) * size_t(ld_func) * sizeof(float), main));   // utterances without a 60 ms frame
  const int64_t total_rows = b->h_fin_off[b->n_utt];
  if (total_rows == 0) return SMILEHIP_OK;
  int64_t max_rows = 0;
  for (int u = 0; u < b->n_utt; ++u) max_rows = std::max(max_rows, b->h_fin_off[u + 1] - b->h_fin_off[u] + 1);
  constexpr int kParts = sizeof(kEgemapsParts) / sizeof(kEgemapsParts[0]);
  FsParams Ps[kParts];
  FsNeed needs[kParts];
  size_t offs[kParts], total = 0;
  int off = 0;
  for (int i = 0; i < kParts; ++i) {
    const EgPart &part = kEgemapsParts[i];
    FsParams &P = Ps[i];
    std::memset(&P, 0, sizeof(P));
    int rc = smilehip_funcspec_egemaps(part.inst, &P.spec);
    if (rc) return rc;
    const int per = spec_layout(&P.spec, nullptr, nullptr);
    if (per < 0) return per;
    P.x = b->d_func_in.p;
    P.ld_x = 36;
    P.col_first = part.col_first;
    P.n_cols = part.n_cols;
    P.row_off = b->d_fin_off.p;
    P.single_rows = -1;
    P.rows_cut = part.rows_cut;
    P.pending = part.pend == 1 ? b->f0_batch->d_pending.p : (part.pend == 2 ? b->d_pending_j.p : nullptr);
    P.out = d_func + off;
    P.ld_out = ld_func;
    needs[i] = spec_need(P, b->n_utt, total_rows + b->n_utt, max_rows);
    offs[i] = total;
    total += fs_bytes(P, needs[i]);
    off += per * part.n_cols;
  }
  if (off != 88) return fail(SMILEHIP_ERR_INVALID, "internal: eGeMAPSv02 functionals layout adds up to %d", off);
  int rc = fs_reserve(plan->ctx, total, main);
  if (rc) return rc;
  for (int i = 0; i < kParts; ++i) {
    fs_carve(static_cast<char *>(plan->ctx->fs_scratch) + offs[i], Ps[i], needs[i]);
    Ps[i].max_rows = max_rows;
    if ((rc = launch_spec(Ps[i], b->n_utt, main))) return rc;
  }
  hipError_t e = launch_gemaps_dbp(d_func + 87, ld_func, b->n_utt, b->d_fin_off.p, main);   // [egemapsv02_leq] dBp
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "dBp kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}
