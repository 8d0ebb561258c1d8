// libsmilehip, C ABI part 3: batches (packed utterances), the fused chain runs, functionals, timing.
#include "smilehip_internal.hpp"

// ------------------------------------------------------------------ batch
smilehip_batch::~smilehip_batch() { delete f0_batch; }

static inline bool compare_ab_like(const smilehip_plan *p) {
  return p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB || p->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE;
}
static inline bool is_egemaps(const smilehip_plan *p) { return p->cfg.chain_kind == SMILEHIP_CHAIN_EGEMAPS; }

extern "C" int smilehip_batch_create(smilehip_plan *plan, const int64_t *h_off, int32_t n_utt, smilehip_batch **out) {
  if (!plan || !out || n_utt < 0 || (n_utt > 0 && !h_off)) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_create: bad argument");
  *out = nullptr;
  if (!plan->ctx) return fail(SMILEHIP_ERR_NO_DEVICE, "host-only plan: no device attached (tables only)");
  if (plan->stage_mask != SMILEHIP_STAGE_ALL) return fail(SMILEHIP_ERR_INVALID, "single-component plan cannot run the fused chain");
  HIP_TRY(hipSetDevice(plan->ctx->device));
  auto *b = new (std::nothrow) smilehip_batch();
  if (!b) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  b->plan = plan;
  b->n_utt = n_utt;
  b->h_samp_off.assign(h_off, h_off + (n_utt ? n_utt + 1 : 0));
  if (n_utt == 0) b->h_samp_off.assign(1, 0);
  b->h_frame_off.assign(size_t(n_utt) + 1, 0);
  b->h_row_off.assign(size_t(n_utt) + 1, 0);
  const int short_T = chain_short_max();
  const int row_extra = plan_row_extra(plan);
  std::vector<int32_t> tile_utt, tile_t0, dtile_utt, dtile_t0, run_utt, run_t0;
  std::vector<TileRec> tile_rec;
  const int64_t dtile = chain_tile_rows();
  const int64_t tile_frames = plan->use_fast ? fast512_tile_frames()
                              : (plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_F0 ? f0_tile_frames() : (int64_t(1) << 40));
  {                                                      // the run length of the 20 ms frame kernels (lld_compare.hip / lld_gemaps.hip)
    int64_t total_T = 0;
    for (int32_t u = 0; u < n_utt; ++u) { const int64_t len = h_off[u + 1] - h_off[u]; if (len > 0) total_T += smilehip_num_frames(plan, len); }
    b->run_frames = compare_run_frames(total_T);
    if (const char *rf = getenv("SMILEHIP_RUN_FRAMES")) { const int v = atoi(rf); if (v >= 1 && v <= 4096) b->run_frames = v; }   // A/B switch
  }
  for (int32_t u = 0; u < n_utt; ++u) {
    const int64_t len = h_off[u + 1] - h_off[u];
    if (len < 0) {
      delete b;
      return fail(SMILEHIP_ERR_INVALID, "sample offsets must be non-decreasing (utterance %d)", u);
    }
    const int64_t T = smilehip_num_frames(plan, len);
    int64_t rows = T > 0 ? T + row_extra : 0;
    if (compare_ab_like(plan)) {
      // rows = T60 + 1 where T60 = frames of the 60 ms framer ([is13_frame60]); none if T60 < 4
      const int64_t N60 = std::lround(0.060 / plan->geo.period);
      const int64_t T60 = (len >= N60) ? (len - N60) / plan->geo.H + 1 : 0;
      rows = (T60 >= 4) ? T60 + 1 : 0;
      for (int64_t t0 = 0; t0 < T; t0 += b->run_frames) {
        run_utt.push_back(u);
        run_t0.push_back((int32_t)t0);
      }
    }
    if (is_egemaps(plan)) {
      // rows = T60 + 1 (what both egemapsv02_lldsetE_smo and egemapsv02_lldsetF_smo hold); none without a 60 ms frame
      const int64_t N60 = std::lround(0.060 / plan->geo.period);
      const int64_t T60 = (len >= N60) ? (len - N60) / plan->geo.H + 1 : 0;
      rows = (T60 >= 1) ? T60 + 1 : 0;
      for (int64_t t0 = 0; t0 < T; t0 += b->run_frames) {
        run_utt.push_back(u);
        run_t0.push_back((int32_t)t0);
      }
      if (b->h_fin_off.empty()) b->h_fin_off.assign(size_t(n_utt) + 1, 0);
      b->h_fin_off[u + 1] = b->h_fin_off[u] + (T60 >= 1 ? T + 1 : 0);
    }
    b->h_frame_off[u + 1] = b->h_frame_off[u] + T;
    b->h_row_off[u + 1] = b->h_row_off[u] + rows;
    if (T > 0 && T <= short_T) b->h_short.push_back(u);
    if (T > 0 && (h_off[u] & 1)) b->all_even = false;
    for (int64_t t0 = 0; t0 < T; t0 += tile_frames) {
      tile_utt.push_back(u);
      tile_t0.push_back((int32_t)t0);
      TileRec r;
      r.samp0 = h_off[u] + t0 * plan->geo.H;
      r.row0 = b->h_frame_off[u] + t0;
      r.n_frames = (int32_t)std::min<int64_t>(tile_frames, T - t0);
      r.pad = 0;
      tile_rec.push_back(r);
    }
    for (int64_t t0 = 0; t0 < rows; t0 += dtile) {
      dtile_utt.push_back(u);
      dtile_t0.push_back((int32_t)t0);
    }
  }
  b->total_frames = b->h_frame_off[n_utt];
  b->total_rows = b->h_row_off[n_utt];
  b->n_tiles = (int32_t)tile_utt.size();
  b->n_dtiles = (int32_t)dtile_utt.size();
  int rc;
  if ((rc = b->d_samp_off.upload(b->h_samp_off)) || (rc = b->d_frame_off.upload(b->h_frame_off)) ||
      (rc = b->d_row_off.upload(b->h_row_off)) ||
      (rc = b->d_tile_utt.upload(tile_utt)) || (rc = b->d_tile_t0.upload(tile_t0)) || (rc = b->d_tile_rec.upload(tile_rec)) ||
      (rc = b->d_dtile_utt.upload(dtile_utt)) || (rc = b->d_dtile_t0.upload(dtile_t0)) ||
      (rc = b->d_short.upload(b->h_short))) {
    delete b;
    return rc;
  }
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_IS09 && b->total_frames > 0) {
    std::vector<int32_t> fu((size_t)b->total_frames);
    for (int32_t u = 0; u < n_utt; ++u) std::fill(fu.begin() + b->h_frame_off[u], fu.begin() + b->h_frame_off[u + 1], u);
    if ((rc = b->d_frame_utt.upload(fu))) { delete b; return rc; }
  }
  if (compare_ab_like(plan)) {
    b->n_runs = (int32_t)run_utt.size();
    if ((rc = b->d_run_utt.upload(run_utt)) || (rc = b->d_run_t0.upload(run_t0))) {
      delete b;
      return rc;
    }
    const size_t nf = size_t(b->total_frames ? b->total_frames : 1);
    if (smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_rawA.p), nf * 4 * sizeof(float)) != hipSuccess ||
        smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_rawB.p), nf * 55 * sizeof(float)) != hipSuccess ||
        smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_mel1.p), nf * 26 * sizeof(float)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the ComParE scratch matrices failed");
    }
    b->d_rawA.n = nf * 4; b->d_rawB.n = nf * 55; b->d_mel1.n = nf * 26;
    b->d_b_extra.n = size_t(n_utt ? n_utt : 1) * 110;
    if (smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_b_extra.p), b->d_b_extra.n * sizeof(float)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the ComParE scratch matrices failed");
    }
    (void)hipMemset(b->d_rawA.p, 0, nf * 4 * sizeof(float));
  }
  if (is_egemaps(plan)) {
    b->n_runs = (int32_t)run_utt.size();
    if (b->h_fin_off.empty()) b->h_fin_off.assign(size_t(n_utt) + 1, 0);
    if ((rc = b->d_run_utt.upload(run_utt)) || (rc = b->d_run_t0.upload(run_t0)) || (rc = b->d_fin_off.upload(b->h_fin_off)) ||
        (rc = smilehip_batch_create(plan->f0_plan, h_off, n_utt, &b->f0_batch))) {
      delete b;
      return rc;
    }
    const size_t nf = size_t(b->total_frames ? b->total_frames : 1);
    const size_t nf60 = size_t(b->f0_batch->total_frames ? b->f0_batch->total_frames : 1);
    const size_t nfin = size_t(b->h_fin_off[n_utt] ? b->h_fin_off[n_utt] : 1);
    auto alloc = [&](DevBuf<float> &d, size_t n) {
      d.release();
      if (smilehip::dev_malloc(reinterpret_cast<void **>(&d.p), n * sizeof(float)) != hipSuccess) return false;
      d.n = n;
      return true;
    };
    if (!alloc(b->d_raw20, nf * 12) || !alloc(b->d_spec220, nf * 220) || !alloc(b->d_lpc, nf * 12) || !alloc(b->d_formants, nf * 10) ||
        !alloc(b->d_pitch3, nf60 * 3) || !alloc(b->d_jit4, nf60 * 4) || !alloc(b->d_shim, nf60) || !alloc(b->d_harm6, nf60 * 6) ||
        !alloc(b->d_func_in, nfin * 36)) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the eGeMAPS scratch matrices failed (%.1f GB needed)",
                  double(nf * 254 + nf60 * 14 + nfin * 36) * 4e-9);
    }
    std::vector<int32_t> zp(size_t(n_utt ? n_utt : 1), 0);
    if ((rc = b->d_pending_j.upload(zp)) || (rc = b->d_harm_ctl.upload(std::vector<int32_t>(2, 0)))) {
      delete b;
      return rc;
    }
    // cHarmonics reads gemapsv01b_fftmagG60, the level cSpecScale reads: lld_f0_spec keeps its magnitudes (2 KB per 60 ms frame) and
    // lld_gemaps_harm starts from them instead of windowing and transforming every voiced frame a second time. Kept only while it
    // is a modest part of what is free (the caller's output matrices come after this); SMILEHIP_HARM_KEEP_MAG=0 / =1 forces it.
    if (b->f0_batch->total_frames > 0) {
      smilehip_batch *fb = b->f0_batch;
      const int64_t ld = (plan->f0_plan->geo.K + 3) & ~int64_t(3);
      const size_t need = size_t(fb->total_frames) * size_t(ld) * sizeof(float);
      size_t free_b = 0, total_b = 0;
      const char *env = getenv("SMILEHIP_HARM_KEEP_MAG");
      bool keep = hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= free_b / 2;
      if (env && env[0] == '0') keep = false;
      if (env && env[0] == '1') keep = true;
      if (keep && smilehip::dev_malloc(reinterpret_cast<void **>(&fb->d_mag_keep.p), need) == hipSuccess) {
        fb->d_mag_keep.n = need / sizeof(float);
        fb->mag_ld = ld;
      } else {
        fb->d_mag_keep.p = nullptr;
        (void)hipGetLastError();
      }
    }
  }
  // The fast kernel with the two regression stages inside (lld_mfcc512<..., DELTA>): tiles as long as the batch allows -- a tile
  // pays one pass of four frames before it (inside an utterance) and one behind it. L = the tile length whose estimate
  // ceil(tiles / wave slots) x (L / 4 + 2) passes is smallest; an utterance is cut into equal parts of at most L frames
  // (multiples of four: a frame's lane group is its index mod 4).
  if (plan->use_fast && (plan->cfg.chain_kind == SMILEHIP_CHAIN_MFCC || plan->cfg.chain_kind == SMILEHIP_CHAIN_PLP) &&
      plan->cfg.n_delta == 2 && plan->cfg.delta_win == 2 && !plan->cfg.append_log_energy && !plan->cfg.cms && plan->ctx &&
      b->total_frames > 0 && b->all_even && !getenv("SMILEHIP_NO_FUSED_DELTA")) {     // (all_even: the dword loads of the aligned instance)
    const int64_t slots = std::max<int64_t>(1, plan->fast.max_blocks) * 8;
    const auto parts_of = [&](int64_t T, int64_t L) { return (T + L - 1) / L; };
    int64_t bestL = 32;
    double best = 1e300;
    // (the count of distinct utterance lengths, not the count of utterances, is what the 505 candidate lengths are tried on)
    std::map<int64_t, int64_t> hist;
    for (int32_t u = 0; u < n_utt; ++u) {
      const int64_t T = b->h_frame_off[u + 1] - b->h_frame_off[u];
      if (T > 0) hist[T]++;
    }
    for (int64_t L = 32; L <= 2048; L += 4) {
      int64_t n = 0;
      for (const auto &h : hist) n += h.second * ((h.first <= short_T) ? 1 : parts_of(h.first, L));
      const double cost = double((n + slots - 1) / slots) * double(L / 4 + 2);
      if (cost <= best) { best = cost; bestL = L; }
    }
    std::vector<FTileRec> ft;
    for (int32_t u = 0; u < n_utt; ++u) {
      const int64_t T = b->h_frame_off[u + 1] - b->h_frame_off[u];
      if (T <= 0) continue;
      const int64_t parts = (T <= short_T) ? 1 : parts_of(T, bestL);
      const int64_t len = (((T + parts - 1) / parts) + 3) & ~int64_t(3);
      for (int64_t t0 = 0; t0 < T; t0 += len) {
        const int64_t t1 = std::min<int64_t>(T, t0 + len), p0 = t0 > 0 ? t0 - 4 : 0;
        FTileRec r;
        r.samp0 = h_off[u] + p0 * plan->geo.H;
        r.row0 = b->h_frame_off[u] + p0;
        const int64_t last = (t1 + 3) & ~int64_t(3);      // first frame of the last pass: the one behind the tile's last frame (rows are written one pass late)
        r.n_frames = (int32_t)(last - p0 + 4);
        r.live_n = (int32_t)(T - p0);
        r.e0 = (int32_t)(t0 - p0);
        r.e1 = (int32_t)(t1 - p0);
        r.lo = (int32_t)(-p0);
        r.delta_on = T > short_T;
        ft.push_back(r);
      }
    }
    std::stable_sort(ft.begin(), ft.end(), [](const FTileRec &a, const FTileRec &c) { return a.n_frames > c.n_frames; });   // long tiles first
    b->n_ftiles = (int32_t)ft.size();
    if ((rc = b->d_ftile_rec.upload(ft))) { delete b; return rc; }
  }
  if ((plan->cfg.chain_kind == SMILEHIP_CHAIN_MFCC || plan->cfg.chain_kind == SMILEHIP_CHAIN_PLP) && plan->cfg.n_delta > 0 &&
      plan->ctx && b->total_frames > 0 && b->n_ftiles == 0) {
    const size_t n = size_t(b->total_frames) * size_t(plan_n_static(plan));
    if (smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_static.p), n * sizeof(float)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the static-block scratch failed");
    }
    b->d_static.n = n;
  }
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_F0) {
    const size_t nf = size_t(b->total_frames ? b->total_frames : 1);
    if (smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_shs.p), nf * 21 * sizeof(float)) != hipSuccess ||
        smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_e60.p), nf * sizeof(float)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the F0 scratch matrices failed");
    }
    b->d_shs.n = nf * 21; b->d_e60.n = nf;
    const size_t nab = (size_t)std::max<int64_t>(f0_scratch_doubles(b->n_tiles, (int)plan->geo.K), 1);
    if (smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_f0_ab.p), nab * sizeof(double)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the F0 row scratch (%zu MB) failed", nab * sizeof(double) >> 20);
    }
    b->d_f0_ab.n = nab;
    // SMILEHIP_F0_PIPE=1 (round 6, measured and NOT the default): a second set of rows for the chunk pipeline (F0Pipe). Bit-identical
    // (tests/test_gpu_f0_pipe.py) and no faster: config 4 243.4 ms piped against 241.0 ms chunk after chunk, config 5 1087 against
    // 1081 -- side by side the three kernels stretch (spec 39 -> 102 ms, cand 36 -> 45, sweep 26 -> 38 summed) by what they gain;
    // the idle issue slots their counters show are not slots another kernel's waves can use.
    {
      static const bool pipe_on = [] { const char *e = getenv("SMILEHIP_F0_PIPE"); return e && e[0] == '1'; }();
      if (pipe_on && b->n_tiles > f0_chunk_tiles()) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 4 > nab * sizeof(double) &&
            smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_f0_ab2.p), nab * sizeof(double)) == hipSuccess)
          b->d_f0_ab2.n = nab;
        else
          (void)hipGetLastError();
      }
    }
    std::vector<int32_t> zp(size_t(n_utt ? n_utt : 1), 0);
    if ((rc = b->d_pending.upload(zp)) || (rc = b->d_jit_redo.upload(zp))) {
      delete b;
      return rc;
    }
    // cPitchJitter's work items: 64 consecutive frames of one utterance each, all first chunks, then all second chunks, ...
    // (the chains that begin in a chunk can run to the utterance's end: the longest possible ones are launched first)
    std::vector<int32_t> ju, jt;
    int64_t maxT = 0;
    for (int32_t u = 0; u < n_utt; ++u) maxT = std::max(maxT, b->h_frame_off[u + 1] - b->h_frame_off[u]);
    for (int64_t t0 = 0; t0 < maxT; t0 += jitter_chunk_frames())
      for (int32_t u = 0; u < n_utt; ++u)
        if (t0 < b->h_frame_off[u + 1] - b->h_frame_off[u]) { ju.push_back(u); jt.push_back((int32_t)t0); }
    if ((rc = b->d_jit_utt.upload(ju)) || (rc = b->d_jit_t0.upload(jt)) || (rc = b->d_jit_ctl.upload(std::vector<int32_t>(2, 0)))) {
      delete b;
      return rc;
    }
  }
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE) {
    if ((rc = smilehip_batch_create(plan->f0_plan, h_off, n_utt, &b->f0_batch))) {
      delete b;
      return rc;
    }
    const size_t nf = size_t(b->f0_batch->total_frames ? b->f0_batch->total_frames : 1);
    if (smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_pitch2.p), nf * 2 * sizeof(float)) != hipSuccess ||
        smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_jit4.p), nf * 4 * sizeof(float)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the F0 group's scratch matrices failed");
    }
    b->d_pitch2.n = nf * 2; b->d_jit4.n = nf * 4;
  }
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_IS09) {
    std::vector<float> zero;   // allocate only
    b->d_raw16.release();
    b->d_raw16.n = size_t(b->total_frames) * 16;
    if (smilehip::dev_malloc(reinterpret_cast<void **>(&b->d_raw16.p), (b->d_raw16.n ? b->d_raw16.n : 1) * sizeof(float)) != hipSuccess) {
      delete b;
      return fail(SMILEHIP_ERR_HIP, "hipMalloc of the IS09 scratch matrix failed");
    }
  }
  *out = b;
  return SMILEHIP_OK;
}

extern "C" void smilehip_batch_destroy(smilehip_batch *b) { delete b; }
extern "C" int64_t smilehip_batch_total_frames(const smilehip_batch *b) { return b ? b->total_frames : 0; }
extern "C" int64_t smilehip_batch_total_rows(const smilehip_batch *b) { return b ? b->total_rows : 0; }
extern "C" int smilehip_batch_delta_fused(const smilehip_batch *b) { return b && b->n_ftiles > 0 ? 1 : 0; }
extern "C" int smilehip_batch_frame_offsets(const smilehip_batch *b, int64_t *o) {
  if (!b || !o) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_frame_offsets: null argument");
  std::memcpy(o, b->h_row_off.data(), b->h_row_off.size() * sizeof(int64_t));
  return SMILEHIP_OK;
}

// -------------------------------------------------------------------- run
static bool serial_forced() {                              // SMILEHIP_SERIAL=1: every kernel alone on the device (kernel tables)
  static const bool v = [] { const char *e = getenv("SMILEHIP_SERIAL"); return e && e[0] != '0'; }();
  return v;
}
static bool serial_streams(int64_t total_frames) {
  static const int forced = [] { const char *e = getenv("SMILEHIP_SERIAL"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
  return forced >= 0 ? forced != 0 : total_frames >= (int64_t)4000000;
}
static void fill_params(const smilehip_plan *p, const smilehip_batch *b, const int16_t *d_pcm, float *d_out,
                        int64_t ld, LldParams &P) {
  std::memset(&P, 0, sizeof(P));
  P.pcm = d_pcm;
  P.pcm_f32 = b->run_pcm_f32;
  P.pcm_total = b->h_samp_off.back();
  P.samp_off = b->d_samp_off.p;
  P.frame_off = b->d_frame_off.p;
  P.frame_utt = b->d_frame_utt.p;
  P.tile_utt = b->d_tile_utt.p;
  P.tile_t0 = b->d_tile_t0.p;
  P.tile_rec = b->d_tile_rec.p;
  P.ftile_rec = b->d_ftile_rec.p;
  P.n_ftiles = b->n_ftiles;
  P.n_utt = b->n_utt;
  P.n_tiles = b->n_tiles;
  P.total_frames = b->total_frames;
  P.out = d_out;
  P.ld_out = ld;
  P.N = (int32_t)p->geo.N;
  P.H = (int32_t)p->geo.H;
  P.Nfft = (int32_t)p->geo.Nfft;
  P.K = (int32_t)p->geo.K;
  P.pad_left = p->cfg.zero_pad_symmetric ? (int32_t)((p->geo.Nfft - p->geo.N) / 2) : 0;
  P.preemph = p->cfg.preemph;
  P.de = p->cfg.preemph_de;
  P.k = p->cfg.preemph_k;
  P.one_minus_k = 1 - p->cfg.preemph_k;     // (1-k) in float, vectorPreemphasis.cpp:94
  P.win_offset = (float)p->cfg.win_offset;
  P.window = p->d_window.p;
  P.tw_half = p->d_tw_half.p;
  P.tw_full = p->d_tw_full.p;
  P.oo = p->fft_radix2 ? OouraTab{} : p->oo.tab();
  P.mel_coef = p->d_mel_coef.p;
  P.mel_rng = p->d_mel_rng.p;
  P.mel_scale = p->mel.scale;
  P.use_power = p->cfg.use_power;
  P.n_bands = p->mel.n_bands;
  P.dct_rows = p->d_dct_rows.p;
  P.dct_gain = p->d_dct_gain.p;
  P.n_mfcc = p->dct.n_mfcc;
  P.melfloor = p->dct.melfloor;
  P.log_floor = p->dct.log_floor;
  P.plp = p->cfg.chain_kind == SMILEHIP_CHAIN_PLP;
  P.plp_order = p->cfg.plp_lp_order;
  P.plp_compression = p->cfg.plp_compression;
  P.plp_eql = p->d_plp_eql.p;
  P.plp_cos = p->d_plp_cos.p;
  P.plp_sin = p->d_plp_sin.p;
}

// R13 for a batch whose rows == frames: level 0 = x (leading dimension ld_x); writes [copy of x at copy_col (if >= 0) |
// order 1 at D | order 2 at 2D] into out
static int delta_chain_from(smilehip_plan *plan, smilehip_batch *b, const float *d_x, int64_t ld_x, int copy_col, float *d_io,
                            int64_t ld, int32_t D, int32_t W, int32_t n_orders, void *stream) {
  if (!plan || !b || !d_io) return fail(SMILEHIP_ERR_INVALID, "smilehip_delta_chain: null argument");
  if (n_orders < 1 || n_orders > 2 || W < 1 || W > 4 || D < 1 || D > 64 || ld < (int64_t)D * (1 + n_orders))   // D: column blocks of 16 on blockIdx.y
    return fail(SMILEHIP_ERR_INVALID, "smilehip_delta_chain: unsupported D=%d W=%d orders=%d ld=%lld", D, W, n_orders, (long long)ld);
  if (b->total_frames == 0) return SMILEHIP_OK;
  // rows == frames is what this entry point assumes (d_io holds the static block)
  if (b->total_rows != b->total_frames) return fail(SMILEHIP_ERR_INVALID, "smilehip_delta_chain: batch belongs to a chain with extra rows");
  ChainParams Q;
  std::memset(&Q, 0, sizeof(Q));
  Q.frame_off = b->d_frame_off.p;
  Q.row_off = b->d_row_off.p;
  Q.tile_utt = b->d_dtile_utt.p;
  Q.tile_t0 = b->d_dtile_t0.p;
  Q.n_tiles = b->n_dtiles;
  Q.n_utt = b->n_utt;
  Q.x = d_x;
  Q.ld_x = ld_x;
  Q.copy_col = copy_col;
  Q.out = d_io;
  Q.ld_out = ld;
  Q.D = D;
  Q.n_stages = n_orders;
  Q.kind[0] = Q.kind[1] = 0;
  Q.W[0] = Q.W[1] = W;
  Q.out_col[0] = D;
  Q.out_col[1] = 2 * D;
  Q.short_T = chain_short_max();
  Q.short_utts = b->d_short.p;
  Q.n_short = (int32_t)b->h_short.size();
  hipError_t e = launch_chain(Q, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "window-chain kernel launch failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_delta_chain(smilehip_plan *plan, smilehip_batch *b, float *d_io, int64_t ld, int32_t D,
                                    int32_t W, int32_t n_orders, void *stream) {
  return delta_chain_from(plan, b, d_io, ld, -1, d_io, ld, D, W, n_orders, stream);
}

extern "C" int smilehip_mfcc_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out,
                                 int64_t ld_out, void *stream) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_run: plan/batch mismatch");
  if (plan->cfg.chain_kind != SMILEHIP_CHAIN_MFCC && plan->cfg.chain_kind != SMILEHIP_CHAIN_PLP)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_run: plan is not an MFCC / PLP chain (use smilehip_lld_run)");
  const int n_static = plan_n_static(plan), n_cep = plan->dct.n_mfcc;      // cepstra [+ log energy]
  const int n_out = n_static * (1 + plan->cfg.n_delta);
  if (ld_out < n_out) return fail(SMILEHIP_ERR_INVALID, "ld_out %lld < n_out %d", (long long)ld_out, n_out);
  if (b->total_frames == 0) return SMILEHIP_OK;
  if (!d_pcm || !d_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_run: null device pointer");
  hipStream_t s = (hipStream_t)stream;
  LldParams P;
  fill_params(plan, b, d_pcm, d_out, ld_out, P);
  // With deltas to follow, the static block goes to a compact [frames x n_mfcc] scratch: the frame kernel
  // then writes whole lines, the window-chain kernel reads 1/3 of what it would read from 39-float rows,
  // and writes every output row in one piece (static | delta | accel).
  const bool aligned_in = b->all_even && ((reinterpret_cast<uintptr_t>(d_pcm) & 3) == 0);
  const bool fused_delta = plan->use_fast && !b->run_pcm_f32 && b->n_ftiles > 0 && aligned_in;   // the regression stages inside the frame kernel
  const bool compact = !fused_delta && plan->cfg.n_delta > 0 && b->d_static.p != nullptr;
  if (compact) {
    P.out = b->d_static.p;
    P.ld_out = n_static;
  }
  hipEvent_t *ev = plan->ev[plan->n_timed % smilehip_plan::kRing];
  if (plan->timing) {
    for (int i = 0; i < 3; ++i)
      if (!ev[i]) HIP_TRY(hipEventCreate(&ev[i]));
    HIP_TRY(hipEventRecord(ev[0], s));
  }
  hipError_t e;
  if (plan->use_fast && !b->run_pcm_f32) {               // (float input: the reference-order kernel; the fast one converts int16 itself)
    Fast512Tables F;
    F.tw256 = plan->d_tw256.p;
    F.tw512 = plan->d_tw512.p;
    F.win = plan->d_fwin.p;
    F.melw = plan->d_melw.p;
    F.melo = plan->d_melo.p;
    F.dct28 = plan->d_dct28.p;
    F.lane_bands = plan->d_lane_bands.p;
    F.mel_units = plan->fast.mel_units;
    F.mel_scale = plan->fast.mel_scale;
    F.plp_eql = plan->d_plp_eql.p;
    F.plp_sin = plan->d_plp_sin.p;
    const bool aligned = aligned_in;
    e = launch_mfcc512(P, F, plan->fast, aligned, fused_delta, s);
  } else {
    e = launch_mfcc_generic(P, s);
  }
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "mfcc kernel launch failed: %s", hipGetErrorString(e));
  if (plan->cfg.append_log_energy) {                   // E variants: [energy:cEnergy] into the static block's last column
    e = launch_log_energy(P, b->d_dtile_utt.p, b->d_dtile_t0.p, b->n_dtiles, P.out, P.ld_out, n_cep, s);
    if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "log-energy kernel launch failed: %s", hipGetErrorString(e));
  }
  if (plan->timing) HIP_TRY(hipEventRecord(ev[1], s));
  if (fused_delta) {                                   // what is left: the tick-accurate path of the very short utterances, in place
    if (!b->h_short.empty()) {
      ChainParams Q;
      std::memset(&Q, 0, sizeof(Q));
      Q.frame_off = b->d_frame_off.p;
      Q.row_off = b->d_row_off.p;
      Q.n_tiles = 0;
      Q.n_utt = b->n_utt;
      Q.x = d_out;
      Q.ld_x = ld_out;
      Q.copy_col = -1;
      Q.out = d_out;
      Q.ld_out = ld_out;
      Q.D = n_static;
      Q.n_stages = 2;
      Q.kind[0] = Q.kind[1] = 0;
      Q.W[0] = Q.W[1] = plan->cfg.delta_win;
      Q.out_col[0] = n_static;
      Q.out_col[1] = 2 * n_static;
      Q.short_T = chain_short_max();
      Q.short_utts = b->d_short.p;
      Q.n_short = (int32_t)b->h_short.size();
      e = launch_chain(Q, s);
      if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "window-chain kernel launch failed: %s", hipGetErrorString(e));
    }
  } else if (plan->cfg.n_delta > 0) {
    int rc = compact ? delta_chain_from(plan, b, b->d_static.p, n_static, 0, d_out, ld_out, n_static,
                                        plan->cfg.delta_win, plan->cfg.n_delta, stream)
                     : smilehip_delta_chain(plan, b, d_out, ld_out, n_static, plan->cfg.delta_win, plan->cfg.n_delta, stream);
    if (rc) return rc;
  }
  if (plan->cfg.cms) {                                 // Z variants: [cms:cFullinputMean] on the cepstra of the output rows
    e = launch_cms(b->d_frame_off.p, b->n_utt, P.out, P.ld_out, d_out, ld_out, n_cep, s);   // P.out: the un-normalised block
    if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "mean-subtraction kernel launch failed: %s", hipGetErrorString(e));
  }
  if (plan->timing) {
    HIP_TRY(hipEventRecord(ev[2], s));
    plan->n_timed++;
  }
  return SMILEHIP_OK;
}

// HIP-event marks of the timing ring for the chains run through smilehip_lld_run: 0 / 1 bracket the chain's dominant frame
// kernel(s) ON THE STREAM THEY ARE LAUNCHED ON, 2 closes the run on the caller's stream (smilehip_plan_last_timing)
static int timing_mark(smilehip_plan *plan, int which, hipStream_t s) {
  if (!plan->timing) return SMILEHIP_OK;
  hipEvent_t *ev = plan->ev[plan->n_timed % smilehip_plan::kRing];
  if (!ev[which]) HIP_TRY(hipEventCreate(&ev[which]));
  HIP_TRY(hipEventRecord(ev[which], s));
  if (which == 2) plan->n_timed++;
  return SMILEHIP_OK;
}

// IS09 LLD set: frame kernel -> pitch smoother -> SMA + delta chain
static int is09_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out, int64_t ld_out, void *stream) {
  const int n_out = plan_n_out(plan);
  if (ld_out < n_out) return fail(SMILEHIP_ERR_INVALID, "ld_out %lld < n_out %d", (long long)ld_out, n_out);
  if (b->total_frames == 0) return SMILEHIP_OK;
  if (!d_pcm || !d_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run: null device pointer");
  hipStream_t s = (hipStream_t)stream;
  LldParams P;
  fill_params(plan, b, d_pcm, d_out, ld_out, P);
  Is09Params I;
  I.raw16 = b->d_raw16.p;
  I.fsSec = (float)plan->geo.fft_frame_size_sec;
  I.maxPitch = plan->cfg.pitch_max;
  I.voicingCutoff = plan->cfg.voicing_cutoff;
  if (I.voicingCutoff > 1.0) I.voicingCutoff = 1.0;       // pitchACF.cpp:96-98
  if (I.voicingCutoff < 0.0) I.voicingCutoff = 0.0;
  if (I.maxPitch < 0.0) I.maxPitch = 0.0;
  int trc;
  if ((trc = timing_mark(plan, 0, s))) return trc;
  hipError_t e = launch_is09(P, I, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "IS09 kernel launch failed: %s", hipGetErrorString(e));
  if ((trc = timing_mark(plan, 1, s))) return trc;
  ChainParams Q;
  std::memset(&Q, 0, sizeof(Q));
  Q.frame_off = b->d_frame_off.p;
  Q.row_off = b->d_row_off.p;
  Q.tile_utt = b->d_dtile_utt.p;
  Q.tile_t0 = b->d_dtile_t0.p;
  Q.n_tiles = b->n_dtiles;
  Q.n_utt = b->n_utt;
  Q.x = b->d_raw16.p;
  Q.ld_x = 16;
  Q.copy_col = -1;
  Q.out = d_out;
  Q.ld_out = ld_out;
  Q.D = 16;
  Q.n_stages = 2;
  Q.kind[0] = 1; Q.W[0] = plan->cfg.sma_win / 2;          // cContourSmoother
  Q.kind[1] = 0; Q.W[1] = plan->cfg.delta_win;            // cDeltaRegression
  Q.out_col[0] = 0;
  Q.out_col[1] = 16;
  Q.short_T = chain_short_max();
  Q.short_utts = b->d_short.p;
  Q.n_short = (int32_t)b->h_short.size();
  e = launch_chain(Q, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "window-chain kernel launch failed: %s", hipGetErrorString(e));
  return timing_mark(plan, 2, s);
}

// ComParE groups A+B: frame kernel -> RASTA scan -> group A (multi-length SMA+delta) + group B chain
static int compare_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out, int64_t ld_out, void *stream,
                       int de_col = 59) {
  const int n_out = plan_n_out(plan);
  if (ld_out < n_out) return fail(SMILEHIP_ERR_INVALID, "ld_out %lld < n_out %d", (long long)ld_out, n_out);
  if (b->total_frames == 0 || b->total_rows == 0) return SMILEHIP_OK;
  if (!d_pcm || !d_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run: null device pointer");
  hipStream_t s = (hipStream_t)stream;
  LldParams P;
  fill_params(plan, b, d_pcm, d_out, ld_out, P);
  CompareParams Q;
  std::memset(&Q, 0, sizeof(Q));
  Q.run_utt = b->d_run_utt.p;
  Q.run_t0 = b->d_run_t0.p;
  Q.run_frames = b->run_frames;
  Q.rawA = b->d_rawA.p;
  Q.rawB = b->d_rawB.p;
  Q.mel1 = b->d_mel1.p;
  Q.eql = plan->d_eql.p;
  Q.eql_log = plan->d_eql_log.p;
  Q.sharp_w = plan->d_sharp.p;
  Q.plp_melfloor = 0.00000000093f;     // cPlp melfloor default (plp.cpp:66), htkcompatible = 0
  Q.compression = 0.33f;
  Q.rasta_iir = plan->rasta_iir;
  for (int i = 0; i < 5; ++i) Q.rasta_fir[i] = plan->rasta_fir[i];
  Q.fsSec = plan->geo.fft_frame_size_sec;
  Q.N60 = (int32_t)std::lround(0.060 / plan->geo.period);
  Q.max_utt_samples = 0;
  for (size_t u2 = 0; u2 + 1 < b->h_samp_off.size(); ++u2) Q.max_utt_samples = std::max<int64_t>(Q.max_utt_samples, b->h_samp_off[u2 + 1] - b->h_samp_off[u2]);
  for (int i = 0; i < 2; ++i) {
    Q.band_iL[i] = plan->band_iL[i]; Q.band_iR[i] = plan->band_iR[i];
    Q.band_wL[i] = plan->band_wL[i]; Q.band_wR[i] = plan->band_wR[i];
  }
  Q.slope_Sf = plan->slope_Sf;
  Q.slope_S2f = plan->slope_S2f;
  int trc;
  if ((trc = timing_mark(plan, 0, s))) return trc;
  hipError_t e = launch_compare(P, Q, b->n_runs, b->d_row_off.p, b->total_rows, d_out, ld_out, de_col, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "ComParE kernel launch failed: %s", hipGetErrorString(e));
  if ((trc = timing_mark(plan, 1, s))) return trc;
  ChainParams C;
  std::memset(&C, 0, sizeof(C));
  C.frame_off = b->d_frame_off.p;
  C.row_off = b->d_row_off.p;
  C.tile_utt = b->d_dtile_utt.p;
  C.tile_t0 = b->d_dtile_t0.p;
  C.n_tiles = b->n_dtiles;
  C.n_utt = b->n_utt;
  C.x = b->d_rawB.p;
  C.ld_x = 55;
  C.copy_col = -1;
  C.out = d_out;
  C.ld_out = ld_out;
  C.D = 55;
  C.n_stages = 2;
  C.kind[0] = 1; C.W[0] = 1;
  C.kind[1] = 0; C.W[1] = 2;
  C.out_col[0] = 4;
  C.out_col[1] = de_col + 4;
  C.short_T = chain_short_max();
  C.short_utts = b->d_short.p;
  C.n_short = (int32_t)b->h_short.size();
  e = launch_chain(C, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "window-chain kernel launch failed: %s", hipGetErrorString(e));
  e = launch_compare_b_extra(b->d_frame_off.p, b->d_row_off.p, b->n_utt, b->d_rawB.p, b->d_b_extra.p, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "group-B extra-row kernel launch failed: %s", hipGetErrorString(e));
  return de_col == 59 ? timing_mark(plan, 2, s) : SMILEHIP_OK;     // (inside the whole-level chain the caller closes the run)
}

void fill_f0_params(const smilehip_plan *plan, F0Params &Q) {
  std::memset(&Q, 0, sizeof(Q));
  Q.N = (int32_t)plan->geo.N; Q.H = (int32_t)plan->geo.H; Q.Nfft = (int32_t)plan->geo.Nfft; Q.K = (int32_t)plan->geo.K;
  Q.pad_left = plan->cfg.zero_pad_symmetric ? (int32_t)((plan->geo.Nfft - plan->geo.N) / 2) : 0;
  Q.window = plan->d_window.p;
  Q.tw_half = plan->d_tw_half.p;
  Q.tw_full = plan->d_tw_full.p;
  Q.oo = plan->fft_radix2 ? OouraTab{} : plan->oo.tab();
  Q.sp_rec = plan->d_f0_rec.p; Q.sp_d1 = plan->d_f0_d1.p; Q.sp_d2 = plan->d_f0_d2.p;
  Q.ip_k = plan->d_f0_k.p; Q.ip_co = plan->d_f0_co.p; Q.audw = plan->d_f0_audw.p;
  Q.ip_rec = plan->d_f0_iprec.p; Q.ip_cnt = plan->d_f0_ipcnt.p; Q.sw_rec = plan->d_f0_swrec.p;
  Q.n_harm = plan->f0.n_harm;
  for (int i = 0; i < 16; ++i) { Q.shift[i] = plan->f0.shift[i]; Q.scale[i] = plan->f0.scale[i]; }
  Q.Fmint = plan->f0.Fmint; Q.Fstept = plan->f0.Fstept;
  Q.log_base = plan->f0.log_base;
  Q.min_pitch = plan->cfg.pitch_min; Q.max_pitch = plan->cfg.pitch_max;
  Q.voicing_cutoff = (float)plan->cfg.voicing_cutoff;
  Q.min_energy = plan->cfg.f0_min_energy;
  Q.jit_Tw = plan->geo.period;                 // 1.0 / (double)sampleRate, waveSource.cpp:190
  Q.jit_broken_thresh = plan->cfg.jitter_broken_thresh;
  Q.jit_step_sec = plan->cfg.frame_step_sec;
  Q.ld_tap = plan->geo.K;
  Q.ld_shs = 21;
  // [is13_pitchSmoothViterbi]: wLocal 2, wTvv 10, wTvvd 5, wTvuv 10, wThr 4, wRange 1 -- but
  // cSmileViterbiPitchSmooth::setWeights stores tvv into wTvvd (pitchSmootherViterbi.hpp:291-299): 10
  Q.vit_w[0] = 2.0; Q.vit_w[1] = 10.0; Q.vit_w[2] = 10.0; Q.vit_w[3] = 10.0; Q.vit_w[4] = 4.0; Q.vit_w[5] = 1.0;
  Q.vit_buf = plan->cfg.vit_buffer_len > 0 ? plan->cfg.vit_buffer_len : 30;
  Q.jit_search_range = plan->cfg.jitter_search_range > 0.0 ? plan->cfg.jitter_search_range : 0.25;
  Q.n_cand = plan->cfg.shs_n_candidates > 0 ? plan->cfg.shs_n_candidates : 6;
  Q.old_peaks = plan->cfg.shs_old_peak_algo ? 1 : 0;
  Q.scale_off = plan->cfg.specscale_off & 7;
}

// cPitchJitter's work items and redo marks of an F0-group batch (lld_jitter.hip)
static void set_jitter_items(const smilehip_batch *fb, F0Params &Q) {
  Q.jit_item_utt = fb->d_jit_utt.p;
  Q.jit_item_t0 = fb->d_jit_t0.p;
  Q.n_jit_items = (int32_t)fb->d_jit_utt.n;
  Q.jit_redo = fb->d_jit_redo.p;
  Q.jit_ctl = fb->d_jit_ctl.p;
  // (the chain's F0 values are cPitchShs candidates: inside [minPitch, maxPitch], pitchBase.cpp:211-222)
  Q.jit_cap = jitter_wave_capacity(Q.jit_Tw, Q.N, Q.min_pitch, Q.jit_search_range);
}

// log_out: rows [F0final, F0finalLog, voicingFinalUnclipped] (the eGeMAPS sub-chain, ld_out >= 3) instead of [F0final, voicing]
static int f0_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out, int64_t ld_out, void *stream,
                  bool log_out = false, hipEvent_t frames_done = nullptr) {
  if (ld_out < (log_out ? 3 : 2)) return fail(SMILEHIP_ERR_INVALID, "ld_out %lld too small", (long long)ld_out);
  if (b->total_frames == 0) return SMILEHIP_OK;
  if (!d_pcm || !d_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run: null device pointer");
  if (plan->cfg.specscale_off & 7)
    return fail(SMILEHIP_ERR_INVALID, "the fused F0 chain runs cSpecScale with enhancement, smoothing and weighting on (specscale_off is for smilehip_specscale_frames)");
  LldParams P;
  fill_params(plan, b, d_pcm, d_out, ld_out, P);
  F0Params Q;
  fill_f0_params(plan, Q);
  Q.shs = b->d_shs.p;
  Q.e60 = b->d_e60.p;
  Q.ab = b->d_f0_ab.p;
  Q.ab_rows = f0_scratch_rows(b->n_tiles);
  Q.hps_tap = b->d_hps_tap;
  Q.mag_keep = b->d_mag_keep.p;
  Q.mag_ld = b->mag_ld;
  Q.pending = b->d_pending.p;
  Q.vit_log_out = log_out ? 1 : 0;
  int trc;
  if ((trc = timing_mark(plan, 0, (hipStream_t)stream))) return trc;       // (a sub-chain's plan never has timing switched on)
  const F0Pipe *pipe = nullptr;
  if (b->d_f0_ab2.p && !serial_forced()) {
    F0Pipe &fp = plan->f0_pipe;
    if (!fp.spec) {
      HIP_TRY(hipStreamCreateWithFlags(&fp.spec, hipStreamNonBlocking));
      HIP_TRY(hipStreamCreateWithFlags(&fp.sweep, hipStreamNonBlocking));
      for (hipEvent_t *ev_ : {&fp.start, &fp.spec_done[0], &fp.spec_done[1], &fp.sweep_done[0], &fp.sweep_done[1], &fp.cand_done[0], &fp.cand_done[1]})
        HIP_TRY(hipEventCreateWithFlags(ev_, hipEventDisableTiming));
    }
    fp.ab2 = b->d_f0_ab2.p;
    pipe = &fp;
  }
  hipError_t e = launch_f0(P, Q, plan->ctx->prop.multiProcessorCount, d_out, ld_out, (hipStream_t)stream, frames_done, pipe);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "F0 kernel launch failed: %s", hipGetErrorString(e));
  if ((trc = timing_mark(plan, 1, (hipStream_t)stream))) return trc;
  return timing_mark(plan, 2, (hipStream_t)stream);
}

// SMILEHIP_CHAIN_COMPARE: groups A+B into columns 6..64 / 71..129, the 60 ms sub-chain (F0 contour into scratch), then
// cPitchJitter and the F0 group's 12 LLD columns into 0..5 / 65..70
static int compare_full_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out, int64_t ld_out, void *stream) {
  if (ld_out < 130) return fail(SMILEHIP_ERR_INVALID, "ld_out %lld < 130", (long long)ld_out);
  if (b->total_rows == 0) return SMILEHIP_OK;
  if (!d_pcm || !d_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run: null device pointer");
  // The two groups share nothing but the PCM. The F0 group's frame kernels fill the device; its Viterbi and jitter passes
  // are one wave per utterance: groups A+B start on the plan's side stream when the frame kernels are done and run beside
  // those (both groups filling the device at once only took turns).
  hipStream_t s = (hipStream_t)stream;
  smilehip_batch *fb = b->f0_batch;
  int rc = f0_run(plan->f0_plan, fb, d_pcm, b->d_pitch2.p, 2, stream, false, plan->ev_fork);
  if (rc) return rc;
  if (fb->total_frames == 0) HIP_TRY(hipEventRecord(plan->ev_fork, s));
  // Small batches: the one-wave-per-utterance passes leave most of the device idle (500 x 10 s: 32.8 -> 28.5 ms with the side stream).
  // Batches that fill the device on their own: rounds 3-5 measured one stream a hair faster (config 4 315 -> 313 ms); with round 6's
  // kernels the side stream is (217.5 / 218.1 / 218.9 ms on one stream, 216.6 / 217.7 / 217.5 beside: groups A+B fill the tails of the
  // Viterbi and jitter passes), so this chain always forks. The eGeMAPS chain keeps the threshold (config 5: 1012.5 on one stream, 1028.9
  // forked, 1029 with the 20 ms chain alone forked beside the Viterbi pass: profiles/r06_streams_ab.txt). SMILEHIP_SERIAL=1 forces one stream (a kernel trace then shows each kernel alone).
  const bool serial = serial_forced();
  hipStream_t side = serial ? s : plan->side_stream;
  HIP_TRY(hipStreamWaitEvent(side, plan->ev_fork, 0));
  if ((rc = compare_run(plan, b, d_pcm, d_out + 6, ld_out, side, 65))) return rc;
  HIP_TRY(hipEventRecord(plan->ev_join, side));
  LldParams P;
  fill_params(plan->f0_plan, fb, d_pcm, d_out, ld_out, P);
  F0Params Q;
  fill_f0_params(plan->f0_plan, Q);
  Q.pending = fb->d_pending.p;
  set_jitter_items(fb, Q);
  hipError_t e = launch_f0_lld(P, Q, b->d_row_off.p, b->d_pitch2.p, b->d_jit4.p, d_out, ld_out, 0, 65, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "F0 LLD kernel launch failed: %s", hipGetErrorString(e));
  HIP_TRY(hipStreamWaitEvent(s, plan->ev_join, 0));
  return timing_mark(plan, 2, s);
}

// SMILEHIP_CHAIN_EGEMAPS: the 20 ms kernels on the plan's side stream, the F0 group on the caller's; then cPitchJitter and
// cHarmonics (which need the decided F0 contour, cHarmonics also the formants), then the selectors / smoothers
void gemaps_plan_consts(const smilehip_plan *plan, GemapsParams &G) {
  std::memset(&G, 0, sizeof(G));
  G.eql = plan->d_eql.p;
  G.plp_melfloor = 0.00000000093f;     // cPlp melfloor default (plp.cpp:66), htkcompatible = 0
  G.compression = 0.33f;
  G.fsSec = plan->geo.fft_frame_size_sec;
  for (int i = 0; i < 2; ++i) {
    G.sl_iL[i] = plan->gm_sl_iL[i]; G.sl_iR[i] = plan->gm_sl_iR[i];
    G.sl_wL[i] = plan->gm_sl_wL[i]; G.sl_wR[i] = plan->gm_sl_wR[i]; G.sl_Nind[i] = plan->gm_sl_Nind[i];
  }
  G.rng_lo = plan->gm_rng_lo; G.rng_hi = plan->gm_rng_hi;
  G.ar_n1 = plan->gm_ar_n1; G.ar_n2 = plan->gm_ar_n2;
  G.spec_floor = plan->gm_spec_floor; G.log_spec_floor = plan->gm_log_spec_floor; G.log_spec_factor = plan->gm_log_spec_factor;
  G.rs_cos = plan->d_rs_cos.p; G.rs_sin = plan->d_rs_sin.p;
  G.rs_norm = (float)(plan->geo.Nfft / 2);
  G.fm_T = 1.0 / plan->gm_target_fs;   // cSpecResample::configureWriter: basePeriod = 1 / targetFs
  G.fm_min = 50.0;                     // [gemapsv01b_formantLpc] minF; maxF 5450 (v01b / v02) or 5500 (v01a)
  G.fm_max = plan->cfg.formant_max_freq > 0.0 ? plan->cfg.formant_max_freq : 5450.0;
  G.fsSec60 = plan->f0_plan->geo.fft_frame_size_sec;
  G.lpc_ld = 12; G.fm_ld = 10;
}

static void fill_gemaps_params(const smilehip_plan *plan, const smilehip_batch *b, GemapsParams &G) {
  gemaps_plan_consts(plan, G);
  const smilehip_batch *fb = b->f0_batch;
  G.run_utt = b->d_run_utt.p; G.run_t0 = b->d_run_t0.p; G.run_frames = b->run_frames;
  G.raw20 = b->d_raw20.p; G.spec220 = b->d_spec220.p;
  G.lpc = b->d_lpc.p; G.formants = b->d_formants.p;
  G.total_frames20 = b->total_frames;
  G.pitch3 = b->d_pitch3.p; G.jit4 = b->d_jit4.p; G.shim_db = b->d_shim.p; G.harm6 = b->d_harm6.p;
  G.frame_off60 = fb->d_frame_off.p;
  G.tile60 = fb->d_tile_rec.p;
  G.n_tiles60 = fb->n_tiles;
  G.pending = fb->d_pending.p;
  G.harm_ctl = b->d_harm_ctl.p;
  G.mag60 = fb->d_mag_keep.p;
  G.mag60_ld = fb->mag_ld;
  G.func_in = b->d_func_in.p;
  G.fin_off = b->d_fin_off.p;
  G.pending_j = b->d_pending_j.p;
}

static int egemaps_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out, int64_t ld_out, void *stream) {
  if (ld_out < 25) return fail(SMILEHIP_ERR_INVALID, "ld_out %lld < 25", (long long)ld_out);
  b->gm_ran = false;
  if (b->total_frames == 0) { b->gm_ran = true; return SMILEHIP_OK; }
  if (!d_pcm || (!d_out && b->total_rows > 0)) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run: null device pointer");
  hipStream_t s = (hipStream_t)stream;
  smilehip_batch *fb = b->f0_batch;
  LldParams P;
  fill_params(plan, b, d_pcm, d_out, ld_out, P);
  GemapsParams G;
  fill_gemaps_params(plan, b, G);
  // The F0 group's frame kernels fill the device, its Viterbi and jitter passes are one wave per utterance: the 20 ms chain
  // (frame kernel, resampling + LPC, formant roots) starts on the side stream when the frame kernels are done
  hipError_t e = hipSuccess;
  const bool serial = serial_streams(b->total_frames);  // (see compare_full_run)
  hipStream_t side = serial ? s : plan->side_stream, bg = serial ? s : plan->bg_stream;
  if (fb->total_frames > 0) {
    int rc = f0_run(plan->f0_plan, fb, d_pcm, b->d_pitch3.p, 3, stream, true, plan->ev_fork);   // SHS candidates -> Viterbi -> energy gate
    if (rc) return rc;
  } else {
    HIP_TRY(hipEventRecord(plan->ev_fork, s));
  }
  HIP_TRY(hipStreamWaitEvent(side, plan->ev_fork, 0));
  int trc;
  if ((trc = timing_mark(plan, 0, side))) return trc;
  e = launch_gemaps_frames(P, G, b->n_runs, side);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "eGeMAPS 20 ms kernels: launch failed: %s", hipGetErrorString(e));
  if ((trc = timing_mark(plan, 1, side))) return trc;
  HIP_TRY(hipEventRecord(plan->ev_join, side));
  if (fb->total_frames > 0) {
    LldParams P60;
    fill_params(plan->f0_plan, fb, d_pcm, nullptr, 0, P60);
    F0Params Q;
    fill_f0_params(plan->f0_plan, Q);
    Q.jit_shim_db = b->d_shim.p;
    set_jitter_items(fb, Q);
    // cPitchJitter is one wave per utterance and latency-bound (a fifth of the VALU issue slots): it runs on the plan's
    // lowest-priority stream beside the 20 ms chain and cHarmonics instead of holding the wave slots they need
    HIP_TRY(hipEventRecord(plan->ev_bg_fork, s));
    HIP_TRY(hipStreamWaitEvent(bg, plan->ev_bg_fork, 0));
    e = launch_f0_jitter(P60, Q, b->d_pitch3.p, 3, b->d_jit4.p, bg);
    if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "jitter kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(plan->ev_bg_join, bg));
    HIP_TRY(hipStreamWaitEvent(s, plan->ev_join, 0));                          // cHarmonics reads the formants
    e = launch_gemaps_harm(P, Q, G, plan->ctx->prop.multiProcessorCount, s);
    if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "harmonics kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipStreamWaitEvent(s, plan->ev_bg_join, 0));
  } else {
    HIP_TRY(hipStreamWaitEvent(s, plan->ev_join, 0));
  }
  e = launch_gemaps_tail(b->d_frame_off.p, b->d_row_off.p, b->n_utt, G, d_out, ld_out, s);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "eGeMAPS tail kernel launch failed: %s", hipGetErrorString(e));
  b->gm_ran = true;
  return timing_mark(plan, 2, s);
}

extern "C" int smilehip_batch_f0_pending(smilehip_batch *b, const int32_t **d_pending) {
  if (!b || !d_pending) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_f0_pending: null argument");
  const smilehip_batch *fb = b->f0_batch ? b->f0_batch : b;
  if (!fb->d_pending.p) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_f0_pending: not an F0 / ComParE / eGeMAPS chain batch");
  *d_pending = fb->d_pending.p;
  return SMILEHIP_OK;
}

extern "C" int smilehip_batch_egemaps_taps(smilehip_batch *b, const float **d_raw20, const float **d_lpc, const float **d_formants,
                                           const float **d_pitch3, const float **d_jit4, const float **d_shim_db, const float **d_harm6,
                                           const float **d_func_in, const int32_t **d_pending, int64_t *h_frame_off60) {
  if (!b || !is_egemaps(b->plan)) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_egemaps_taps: not an eGeMAPS chain batch");
  if (d_raw20) *d_raw20 = b->d_raw20.p;
  if (d_lpc) *d_lpc = b->d_lpc.p;
  if (d_formants) *d_formants = b->d_formants.p;
  if (d_pitch3) *d_pitch3 = b->d_pitch3.p;
  if (d_jit4) *d_jit4 = b->d_jit4.p;
  if (d_shim_db) *d_shim_db = b->d_shim.p;
  if (d_harm6) *d_harm6 = b->d_harm6.p;
  if (d_func_in) *d_func_in = b->d_func_in.p;
  if (d_pending) *d_pending = b->f0_batch->d_pending.p;
  if (h_frame_off60) std::memcpy(h_frame_off60, b->f0_batch->h_frame_off.data(), b->f0_batch->h_frame_off.size() * sizeof(int64_t));
  return SMILEHIP_OK;
}

// Test/diagnostic taps of the F0 chain: device pointers to the per-frame scratch of the LAST run
// (candidates: total_frames x 21, level is13_pitchShsG60; energy: total_frames, level is13_e60) and an optional
// destination for the octave-scaled spectra (total_frames x K, level is13_hpsG60; pass NULL to switch it off).
extern "C" int smilehip_batch_f0_taps(smilehip_batch *b, float *d_hps_dst, const float **d_shs, const float **d_e60) {
  if (!b || b->plan->cfg.chain_kind != SMILEHIP_CHAIN_COMPARE_F0)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_f0_taps: not an F0 chain batch");
  b->d_hps_tap = d_hps_dst;
  if (d_shs) *d_shs = b->d_shs.p;
  if (d_e60) *d_e60 = b->d_e60.p;
  return SMILEHIP_OK;
}

// ------------------------------------------------------------- functionals
extern "C" uint32_t smilehip_functionals_is09_mask(void) {
  return SMILEHIP_FUNC_MAX | SMILEHIP_FUNC_MIN | SMILEHIP_FUNC_RANGE | SMILEHIP_FUNC_MAXPOS | SMILEHIP_FUNC_MINPOS |
         SMILEHIP_FUNC_AMEAN | SMILEHIP_FUNC_LINREGC1 | SMILEHIP_FUNC_LINREGC2 | SMILEHIP_FUNC_LINREGERRQ |
         SMILEHIP_FUNC_STDDEV | SMILEHIP_FUNC_SKEWNESS | SMILEHIP_FUNC_KURTOSIS;
}

extern "C" int smilehip_functionals_count(uint32_t mask) {
  if (mask & ~SMILEHIP_FUNC_ALL) return -1;
  return __builtin_popcount(mask);
}

static const int kIs09FuncRowsCut = 3;      // rows = T+1; functionals see max(1, T-2)

extern "C" int smilehip_batch_func_rows(const smilehip_batch *b, int64_t *rows) {
  if (!b || !rows) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_func_rows: null argument");
  if (b->plan->cfg.chain_kind != SMILEHIP_CHAIN_IS09)
    return fail(SMILEHIP_ERR_INVALID, "functionals are defined for IS09 chain plans only");
  for (int32_t u = 0; u < b->n_utt; ++u) {
    const int64_t r = b->h_row_off[u + 1] - b->h_row_off[u];
    rows[u] = r > 0 ? std::max<int64_t>(1, r - kIs09FuncRowsCut) : 0;
  }
  return SMILEHIP_OK;
}

extern "C" int smilehip_batch_functionals(smilehip_plan *plan, smilehip_batch *b, const float *d_lld, int64_t ld_lld,
                                          uint32_t mask, float *d_func, int64_t ld_func, void *stream) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals: plan/batch mismatch");
  if (plan->cfg.chain_kind != SMILEHIP_CHAIN_IS09)
    return fail(SMILEHIP_ERR_INVALID, "functionals are defined for IS09 chain plans only");
  const int per = smilehip_functionals_count(mask);
  if (per <= 0) return fail(SMILEHIP_ERR_INVALID, "invalid functionals mask 0x%x", mask);
  const int n_cols = plan_n_out(plan);
  if (ld_lld < n_cols || ld_func < (int64_t)n_cols * per)
    return fail(SMILEHIP_ERR_INVALID, "leading dimensions too small (ld_lld %lld, ld_func %lld)", (long long)ld_lld,
                (long long)ld_func);
  if (b->n_utt == 0) return SMILEHIP_OK;
  if (!d_func || (!d_lld && b->total_rows > 0)) return fail(SMILEHIP_ERR_INVALID, "smilehip_batch_functionals: null device pointer");
  // the mask is a cFunctionals instance with functionalsEnabled = Extremes;Regression;Moments, frame-normalised
  // positions and the linear regression only: one spec of the general engine (lld_funcspec.hip), which walks every
  // contour in the reference's order
  smilehip_func_spec spec;
  int rc = smilehip_funcspec_from_mask(mask, plan->geo.frame_period > 0.0 ? plan->geo.frame_period : 0.01, &spec);
  if (rc) return rc;
  return smilehip_batch_funcspec(plan, b, &spec, d_lld, ld_lld, 0, n_cols, kIs09FuncRowsCut, nullptr, 0, d_func, ld_func, stream);
}

extern "C" int smilehip_functionals_matrix(smilehip_context *ctx, const float *d_x, int64_t ld_x, int64_t rows, int32_t cols,
                                           uint32_t mask, float *d_out, void *stream) {
  const int per = smilehip_functionals_count(mask);
  if (!ctx || per <= 0 || rows < 1 || cols < 1 || ld_x < cols || !d_x || !d_out)
    return fail(SMILEHIP_ERR_INVALID, "smilehip_functionals_matrix: bad argument");
  smilehip_func_spec spec;
  int rc = smilehip_funcspec_from_mask(mask, 0.01, &spec);
  if (rc) return rc;
  return smilehip_funcspec_matrix(ctx, &spec, d_x, ld_x, rows, cols, d_out, stream);
}

extern "C" int smilehip_lld_run(smilehip_plan *plan, smilehip_batch *b, const int16_t *d_pcm, float *d_out, int64_t ld_out,
                                void *stream) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run: plan/batch mismatch");
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_MFCC || plan->cfg.chain_kind == SMILEHIP_CHAIN_PLP)
    return smilehip_mfcc_run(plan, b, d_pcm, d_out, ld_out, stream);
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_AB) return compare_run(plan, b, d_pcm, d_out, ld_out, stream);
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE_F0) return f0_run(plan, b, d_pcm, d_out, ld_out, stream);
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_COMPARE) return compare_full_run(plan, b, d_pcm, d_out, ld_out, stream);
  if (plan->cfg.chain_kind == SMILEHIP_CHAIN_EGEMAPS) return egemaps_run(plan, b, d_pcm, d_out, ld_out, stream);
  return is09_run(plan, b, d_pcm, d_out, ld_out, stream);
}

extern "C" int smilehip_lld_run_f32(smilehip_plan *plan, smilehip_batch *b, const float *d_pcm_f32, float *d_out, int64_t ld_out,
                                    void *stream) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run_f32: plan/batch mismatch");
  if (!d_pcm_f32 && b->total_frames > 0) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run_f32: null device pointer");
  b->run_pcm_f32 = d_pcm_f32;
  if (b->f0_batch) b->f0_batch->run_pcm_f32 = d_pcm_f32;
  // the int16 pointer is never dereferenced while run_pcm_f32 is set (PcmIn, lld_device.hpp); it only has to be non-null
  const int rc = smilehip_lld_run(plan, b, reinterpret_cast<const int16_t *>(d_pcm_f32), d_out, ld_out, stream);
  b->run_pcm_f32 = nullptr;
  if (b->f0_batch) b->f0_batch->run_pcm_f32 = nullptr;
  return rc;
}

extern "C" int smilehip_lld_run_host(smilehip_plan *plan, smilehip_batch *b, const int16_t *h_pcm, int64_t n_samples,
                                     float *h_out) {
  if (!plan || !b || b->plan != plan) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run_host: plan/batch mismatch");
  if (n_samples < b->h_samp_off.back()) return fail(SMILEHIP_ERR_INVALID, "PCM buffer shorter than the batch layout");
  if (b->total_rows == 0) return SMILEHIP_OK;
  if (!h_pcm || !h_out) return fail(SMILEHIP_ERR_INVALID, "smilehip_lld_run_host: null pointer");
  HIP_TRY(hipSetDevice(plan->ctx->device));
  const int n_out = plan_n_out(plan);
  int16_t *d_pcm = nullptr;
  float *d_out = nullptr;
  HIP_TRY(hipMalloc((void **)&d_pcm, size_t(n_samples) * sizeof(int16_t)));
  hipError_t e = hipMalloc((void **)&d_out, size_t(b->total_rows) * n_out * sizeof(float));
  if (e != hipSuccess) {
    (void)hipFree(d_pcm);
    return fail(SMILEHIP_ERR_HIP, "hipMalloc(out) failed: %s", hipGetErrorString(e));
  }
  int rc = SMILEHIP_OK;
  e = hipMemcpy(d_pcm, h_pcm, size_t(n_samples) * sizeof(int16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = smilehip_lld_run(plan, b, d_pcm, d_out, n_out, nullptr);
    if (rc == SMILEHIP_OK) e = hipMemcpy(h_out, d_out, size_t(b->total_rows) * n_out * sizeof(float), hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_pcm);
  (void)hipFree(d_out);
  if (rc) return rc;
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "HIP copy failed: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}

extern "C" int smilehip_mfcc_run_host(smilehip_plan *plan, smilehip_batch *b, const int16_t *h_pcm, int64_t n_samples,
                                      float *h_out) {
  if (!plan || plan->cfg.chain_kind != SMILEHIP_CHAIN_MFCC) return fail(SMILEHIP_ERR_INVALID, "smilehip_mfcc_run_host: plan is not an MFCC chain");
  return smilehip_lld_run_host(plan, b, h_pcm, n_samples, h_out);
}

extern "C" int smilehip_plan_set_timing(smilehip_plan *plan, int enable) {
  if (!plan) return fail(SMILEHIP_ERR_INVALID, "null plan");
  plan->timing = enable != 0;
  plan->n_timed = 0;
  return SMILEHIP_OK;
}

// Average over the runs recorded since set_timing (at most the last kRing).
// The caller must have synchronised the stream.
extern "C" int smilehip_plan_last_timing(smilehip_plan *plan, float *ms_main, float *ms_delta) {
  if (!plan || plan->n_timed <= 0) return fail(SMILEHIP_ERR_INVALID, "no timing recorded");
  const int64_t n = plan->n_timed < smilehip_plan::kRing ? plan->n_timed : smilehip_plan::kRing;
  double sa = 0.0, sd = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    float a = 0.f, d = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, plan->ev[i][0], plan->ev[i][1]));
    HIP_TRY(hipEventElapsedTime(&d, plan->ev[i][1], plan->ev[i][2]));
    sa += a;
    sd += d;
  }
  if (ms_main) *ms_main = float(sa / double(n));
  if (ms_delta) *ms_delta = float(sd / double(n));
  return SMILEHIP_OK;
}
