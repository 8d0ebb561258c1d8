// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): R2 - R7: cVectorPreemphasis, cWindower, cTransformFFT, cFFTmagphase, cMelspec, cMfcc
// ---------------------------------------------------------------- overrides
// a small device buffer of raw bytes (results that are not float frames)
struct DevBytes {
  void *d = nullptr;
  uint64_t cap = 0;
  void *ensure(uint64_t bytes) {
    if (bytes > cap) {
      if (d) smilehip_free(context(), d);
      if (smilehip_alloc(context(), bytes, &d)) COMP_ERR("libsmilehip: %s", smilehip_last_error());
      cap = bytes;
    }
    return d;
  }
  void down(void *h, uint64_t bytes) {
    if (smilehip_copy_to_host(context(), h, d, bytes, nullptr) || smilehip_stream_synchronize(context(), nullptr))
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
  }
  ~DevBytes() { if (g_ctx && d) smilehip_free(g_ctx, d); }
};

// R2  cVectorPreemphasis::processVector  (src/dspcore/vectorPreemphasis.cpp:89-107)
class cHipVectorPreemphasis : public BlockVP<cVectorPreemphasis> {
  FrameIO io_;
  bool cpu_warned_ = false;
  float k_ = 0.f;
  int de_ = 0;
  bool ready_ = false;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (!ready_) {
      double f = isSet("f") ? getDouble("f") : -1.0;
      k_ = (FLOAT_DMEM)getDouble("k");
      if (f >= 0.0) k_ = (FLOAT_DMEM)exp(-2.0 * M_PI * f * getBasePeriod());   // vectorPreemphasis.cpp:78-86
      de_ = getInt("de");
      ready_ = true;
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_preemphasis_frames(context(), io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, Ndst, k_, de_, nullptr));
    io_.down(dst, Ndst);
    g_frames[0] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipVectorPreemphasis(const char *n) : BlockVP<cVectorPreemphasis>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipVectorPreemphasis(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// one plan per input field configuration
template <int NPLANS = 8>
struct PlanSet {
  smilehip_plan *p[NPLANS];
  PlanSet() { for (auto &x : p) x = nullptr; }
  ~PlanSet() { for (auto &x : p) if (x) smilehip_plan_destroy(x); }
  smilehip_plan *&at(int i) {
    if (i < 0 || i >= NPLANS) COMP_ERR("libsmilehip plugin: more than %d differently sized fields", NPLANS);
    return p[i];
  }
};

// R3  cWindower::processVector  (src/dspcore/windower.cpp:221-229)
class cHipWindower : public BlockVP<cWindower> {
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      if (getDouble("fade") > 0.0 || getInt("squareRoot") || getDouble("xshift") != 0.0)
        COMP_ERR("libsmilehip plugin: cWindower options fade/squareRoot/xshift are not supported on the HIP path");
      smilehip_lld_config c = base_config(Nsrc, SMILEHIP_STAGE_WINDOW);
      c.win_func = winfunc_id(getStr("winFunc"));
      if (c.win_func < 0) COMP_ERR("libsmilehip plugin: window function '%s' not supported on the HIP path", getStr("winFunc"));
      c.win_sigma = getDouble("sigma");
      c.win_gain = getDouble("gain");
      c.win_offset = getDouble("offset");
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_window_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[1] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipWindower(const char *n) : BlockVP<cWindower>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipWindower(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R4  cTransformFFT::processVector, forward  (src/dspcore/transformFft.cpp:165-223)
class cHipTransformFFT : public BlockVP<cTransformFFT> {
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (getInt("inverse")) {                             // rdft(N, -1) and the 2 / N scaling (transformFft.cpp:196-216)
      if (Nsrc != Ndst || Ndst < 64 || Ndst > 8192 || (Ndst & (Ndst - 1))) {
        HIP_FALLTHROUGH(2, "cTransformFFT inverse = 1: only whole packed spectra of 64 .. 8192 values are built");
        return cTransformFFT::processVector(src, dst, Nsrc, Ndst, idxi);
      }
      smilehip_plan *&pli = plans_.at(getFconf(idxi));
      if (!pli) {
        smilehip_lld_config c = base_config(Nsrc, SMILEHIP_STAGE_FFT);
        check(smilehip_plan_create(context(), &c, &pli));
      }
      io_.ensure(Nsrc, Ndst);
      io_.up(src, Nsrc);
      check(smilehip_irfft_frames(pli, io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
      io_.down(dst, Ndst);
      g_frames[2] += g_blk.n;
      return 1;
    }
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      smilehip_lld_config c = base_config(Nsrc, SMILEHIP_STAGE_FFT);
      c.zero_pad_symmetric = getInt("zeroPadSymmetric");
      check(smilehip_plan_create(context(), &c, &pl));
      smilehip_geometry g;
      check(smilehip_plan_geometry(pl, &g));
      if (g.fft_size != Ndst) COMP_ERR("libsmilehip plugin: FFT size mismatch (%ld vs %ld)", (long)g.fft_size, Ndst);
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_rfft_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[2] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipTransformFFT(const char *n) : BlockVP<cTransformFFT>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipTransformFFT(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R5  cFFTmagphase::processVector, magnitude branch  (src/dspcore/fftmagphase.cpp:215-221)
class cHipFFTmagphase : public BlockVP<cFFTmagphase> {
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
  int plain_ = -1, modes_ = 0;
  bool other_ok_ = false;
  float dbp_norm_ = 0.0f, min_dbp_ = 0.0f;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (plain_ < 0) {
      plain_ = (!getInt("inverse") && getInt("magnitude") && !getInt("phase") && !getInt("normalise") &&
                !getInt("power") && !getInt("dBpsd")) ? 1 : 0;
      // every other output mode of :215-287 but the two the reference itself cannot produce properly: inverse = 1 (mag / phase ->
      // complex) and magnitude + phase as separate fields (:268-276, "this check is wrong" upstream)
      const int mag = getInt("magnitude"), ph = getInt("phase");
      modes_ = (mag ? SMILEHIP_MAGPHASE_MAGNITUDE : 0) | (ph ? SMILEHIP_MAGPHASE_PHASE : 0) |
               (getInt("normalise") ? SMILEHIP_MAGPHASE_NORMALISE : 0) | (getInt("power") ? SMILEHIP_MAGPHASE_POWER : 0) |
               (getInt("dBpsd") ? SMILEHIP_MAGPHASE_DBPSD : 0);
      other_ok_ = !getInt("inverse") && (mag || ph) && !(mag && ph && !getInt("joinMagphase"));
      dbp_norm_ = dBpnorm;                               // the members cFFTmagphase::myFetchConfig filled (:93-98: mindBp >= dBpnorm - 120 enforced)
      min_dbp_ = mindBp;
    }
    if (!plain_ && other_ok_) {
      const long K = Nsrc / 2 + 1;
      const long n_out = ((modes_ & 1) ? K : 0) + ((modes_ & 2) ? K : 0);
      if (n_out <= Ndst && Nsrc >= 4 && !(Nsrc & 1)) {
        io_.ensure(Nsrc, Ndst);
        io_.up(src, Nsrc);
        check(smilehip_fftmagphase_frames(context(), io_.d_in, Nsrc, Nsrc, modes_, dbp_norm_, min_dbp_, io_.d_out, Ndst, g_blk.n, nullptr));
        io_.down(dst, n_out);
        g_frames[3] += g_blk.n;
        return 1;
      }
    }
    if (!plain_) { HIP_FALLTHROUGH(3, "cFFTmagphase: inverse = 1 and magnitude + phase as separate fields are not built"); return cFFTmagphase::processVector(src, dst, Nsrc, Ndst, idxi); }
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      smilehip_lld_config c = base_config(Nsrc, SMILEHIP_STAGE_FFT);
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_fftmag_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[3] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipFFTmagphase(const char *n) : BlockVP<cFFTmagphase>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipFFTmagphase(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R6  cMelspec::processVector  (src/lldcore/melspec.cpp:519-570)
class cHipMelspec : public BlockVP<cMelspec> {
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
  int plain_ = -1;
  DevBytes tab_coef_[8], tab_map_[8];
  bool tab_ready_[8] = {false, false, false, false, false, false, false, false};
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (plain_ < 0) {
      const char *bw = getStr("bwMethod");
      const char *sc = getStr("specScale");
      const bool mel = getInt("htkcompatible") || (sc && !strcasecmp(sc, "mel"));
      plain_ = (!getInt("inverse") && mel && bw && !strncasecmp(bw, "lr", 2)) ? 1 : 0;
    }
    if (!plain_ && !getInt("inverse")) {
      // any other bank cMelspec::computeFilters built (other spectral scales, bwMethod, HFCC, custom bandwidth): the component's own
      // tables go to the device once, the frames through the table-driven operator
      const int fc = getFconf(idxi);
      if (fc >= 0 && fc < 8 && filterCoeffs_ && chanMap_ && filterCoeffs_[fc] && chanMap_[fc] && Ndst == nBands_ && Nsrc <= 8193) {
        const bool dense = hfcc_ || customBandwidth_;
        if (!tab_ready_[fc]) {
          const size_t nc = dense ? (size_t)nBands_ * (size_t)Nsrc : (size_t)Nsrc, nm = dense ? (size_t)2 * nBands_ : (size_t)Nsrc;
          std::vector<float> cf(filterCoeffs_[fc], filterCoeffs_[fc] + nc);
          std::vector<int32_t> cm(nm);
          for (size_t i = 0; i < nm; ++i) cm[i] = (int32_t)chanMap_[fc][i];
          void *d_c = tab_coef_[fc].ensure(sizeof(float) * nc);
          void *d_m = tab_map_[fc].ensure(sizeof(int32_t) * nm);
          if (smilehip_copy_to_device(context(), d_c, cf.data(), sizeof(float) * nc, nullptr) ||
              smilehip_copy_to_device(context(), d_m, cm.data(), sizeof(int32_t) * nm, nullptr))
            COMP_ERR("libsmilehip: %s", smilehip_last_error());
          tab_ready_[fc] = true;
        }
        const float scale = htkcompatible_ ? (usePower_ ? (FLOAT_DMEM)(32767.0 * 32767.0) : (FLOAT_DMEM)32767.0) : 1.0f;
        io_.ensure(Nsrc, Ndst);
        io_.up(src, Nsrc);
        check(smilehip_melspec_table_frames(context(), io_.d_in, Nsrc, Nsrc, nBands_, dense ? 1 : 0, (const float *)tab_coef_[fc].d,
                                            (const int32_t *)tab_map_[fc].d, (int32_t)nLoF_[fc], (int32_t)nHiF_[fc], usePower_, scale,
                                            io_.d_out, Ndst, g_blk.n, nullptr));
        io_.down(dst, Ndst);
        g_frames[4] += g_blk.n;
        return 1;
      }
    }
    if (getInt("inverse")) {
      // inverse = 1 (melspec.cpp:466-516): Nsrc bands -> nBands_ spectrum bins through the component's own tables (computeFilters with the
      // roles swapped, :191-194). The HFCC / custom-bandwidth banks have no inverse in the reference either (:487-490): its own message.
      const int fc = getFconf(idxi);
      const bool dense = hfcc_ || customBandwidth_;
      if (!dense && fc >= 0 && fc < 8 && filterCoeffs_ && chanMap_ && filterCoeffs_[fc] && chanMap_[fc] && Ndst == nBands_ && Ndst >= 2 &&
          Ndst <= 8193 && Nsrc <= 4096) {
        if (!tab_ready_[fc]) {
          const size_t nc = (size_t)Ndst;
          std::vector<float> cf(filterCoeffs_[fc], filterCoeffs_[fc] + nc);
          std::vector<int32_t> cm(nc);
          for (size_t i = 0; i < nc; ++i) cm[i] = (int32_t)chanMap_[fc][i];
          void *d_c = tab_coef_[fc].ensure(sizeof(float) * nc);
          void *d_m = tab_map_[fc].ensure(sizeof(int32_t) * nc);
          if (smilehip_copy_to_device(context(), d_c, cf.data(), sizeof(float) * nc, nullptr) ||
              smilehip_copy_to_device(context(), d_m, cm.data(), sizeof(int32_t) * nc, nullptr))
            COMP_ERR("libsmilehip: %s", smilehip_last_error());
          tab_ready_[fc] = true;
        }
        const float div = htkcompatible_ ? (usePower_ ? (FLOAT_DMEM)(32767.0 * 32767.0) : (FLOAT_DMEM)32767.0) : 1.0f;
        io_.ensure(Nsrc, Ndst);
        io_.up(src, Nsrc);
        check(smilehip_melspec_inverse_table_frames(context(), io_.d_in, Nsrc, (int32_t)Nsrc, Ndst, (const float *)tab_coef_[fc].d,
                                                    (const int32_t *)tab_map_[fc].d, (int32_t)nLoF_[fc], (int32_t)nHiF_[fc], usePower_, div,
                                                    io_.d_out, Ndst, g_blk.n, nullptr));
        io_.down(dst, Ndst);
        g_frames[4] += g_blk.n;
        return 1;
      }
    }
    if (!plain_) { HIP_FALLTHROUGH(4, "cMelspec: a bank without an inverse (HFCC / custom bandwidth)"); return cMelspec::processVector(src, dst, Nsrc, Ndst, idxi); }
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      // frame size of the input spectrum, cMelspec::configureField (melspec.cpp:150-173)
      const sDmLevelConfig *lc = reader_->getLevelConfig();
      double fss = isSet("overrideFrameSizeSec") ? getDouble("overrideFrameSizeSec")
                                                 : (lc->frameSizeSec > 0.0 ? lc->frameSizeSec : lc->lastFrameSizeSec);
      smilehip_lld_config c = base_config((Nsrc - 1) * 2, SMILEHIP_STAGE_MEL);
      c.force_fft_frame_size_sec = fss;
      c.n_bands = getInt("nBands");
      c.lofreq = (FLOAT_DMEM)getDouble("lofreq");
      c.hifreq = (FLOAT_DMEM)getDouble("hifreq");
      c.use_power = getInt("usePower");
      c.mel_htk_compatible = getInt("htkcompatible");
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_melspec_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[4] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipMelspec(const char *n) : BlockVP<cMelspec>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipMelspec(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R7  cMfcc::processVector, forward  (src/lldcore/mfcc.cpp:239-273)
class cHipMfcc : public BlockVP<cMfcc> {
  int fused_ = -1;
  const FusedLevel *fcols_ = nullptr;
  long fnext_ = 0;
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
  cMatrix *fblock_ = nullptr;
 protected:
  // fused chain, tick-level hand-out (plugin_shared.hpp: FusedChain::tick_write): this component's level gets its rows a block
  // per tick, whatever its reader holds (nothing: the wave source idles)
  eTickResult myTick(long long t) override {
    if (fused_ < 0) { g_fused.init(); fcols_ = g_fused.static_level(getStr("writer.dmLevel")); fused_ = fcols_ ? 1 : 0; }
    if (fused_ && g_fused.tick_mode && !fcols_->cols.empty()) {
      if (isEOI()) return TICK_INACTIVE;
      return g_fused.tick_write(*fcols_, writer_, fnext_, fblock_, blocksizeW_);
    }
    return BlockVP<cMfcc>::myTick(t);
  }
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    const bool inv = getInt("inverse") != 0;               // round 6: cepstra back to nBands mel bands (mfcc.cpp:184-235)
    if ((!inv && !getInt("doLog")) || (inv && (Ndst < 2 || Ndst > 64 || Nsrc > 64)))
      { HIP_FALLTHROUGH(5, "cMfcc: doLog = 0 (forward) is not built"); return cMfcc::processVector(src, dst, Nsrc, Ndst, idxi); }
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      smilehip_lld_config c = base_config(512, SMILEHIP_STAGE_MFCC);
      c.n_bands = inv ? (int)Ndst : (int)Nsrc;
      c.first_mfcc = getInt("firstMfcc");
      c.last_mfcc = getInt("lastMfcc");
      if (!isSet("lastMfcc") && isSet("nMfcc")) c.last_mfcc = c.first_mfcc + getInt("nMfcc") - 1;   // mfcc.cpp:77-82
      c.cep_lifter = (FLOAT_DMEM)getDouble("cepLifter");
      c.mfcc_htk_compatible = getInt("htkcompatible");
      c.melfloor = (FLOAT_DMEM)getDouble("melfloor");
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    if (inv) check(smilehip_mfcc_inverse_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, getInt("doLog") ? 1 : 0, nullptr));
    else
    check(smilehip_mfcc_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[5] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipMfcc(const char *n) : BlockVP<cMfcc>(n) {}
  ~cHipMfcc() override { delete fblock_; }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipMfcc(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};
