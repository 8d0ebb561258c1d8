// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): R11, R8: cSpectral, cPlp
// R11  cSpectral::processVector with ComParE_2016's option set  (src/lldcore/spectral.cpp:586-1560)
class cHipSpectral : public BlockVP<cSpectral> {
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
  DevBytes prev_[8];
  bool seen_[8] = {false, false, false, false, false, false, false, false};
  int plain_ = -1, gemaps_ = -1;
  smilehip_plan *gm_plan_ = nullptr;
  int band_lo_[2] = {250, 1000}, band_hi_[2] = {650, 4000};
  bool sel_[3] = {true, true, true};
  // the general option set (smilehip_spectral_op_*: any bands / rollOff points, every descriptor optional): one operator per field
  int general_ = -1, gen_n_out_ = 0;
  smilehip_spectral_opts gen_opts_;
  smilehip_spectral_op *gen_op_[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool array_is(const char *name, int n, const char *const *vals) {
    if (getArraySize(name) != n) return false;
    for (int i = 0; i < n; ++i) {
      const char *v = getStr_f(myvprint("%s[%i]", name, i));
      if (!v || strcmp(v, vals[i]) != 0) return false;
    }
    return true;
  }
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (plain_ < 0) {
      bool ok = getArraySize("bands") == 2 && getArraySize("rollOff") == 4 && getArraySize("slopes") <= 0;
      for (int b = 0; ok && b < 2; ++b) {                  // bands[b] = "lo-hi" in Hz, integers (spectral.cpp:163-190)
        const char *v = getStr_f(myvprint("bands[%i]", b));
        int lo = -1, hi = -1, used = 0;
        ok = v && sscanf(v, "%d-%d%n", &lo, &hi, &used) == 2 && v[used] == 0 && lo >= 0 && hi > lo;
        band_lo_[b] = lo; band_hi_[b] = hi;
      }
      static const double ro[4] = {0.25, 0.50, 0.75, 0.90};
      for (int i = 0; ok && i < 4; ++i) ok = getDouble_f(myvprint("rollOff[%i]", i)) == ro[i];
      // the optional outputs of the fifteen (their values do not enter the others: the centroid is computed whenever a moment or the
      // slope is on, spectral.cpp:1262): column 7, 13, 14 of the device row
      sel_[0] = getInt("centroid") != 0; sel_[1] = getInt("sharpness") != 0; sel_[2] = getInt("harmonicity") != 0;
      static const char *const on[] = {"squareInput", "flux", "entropy", "variance", "skewness", "kurtosis", "slope", "oldSlopeScale"};
      static const char *const off[] = {"normBandEnergies", "specDiff", "specPosDiff", "fluxCentroid", "fluxAtFluxCentroid", "maxPos",
                                        "minPos", "standardDeviation", "alphaRatio", "hammarbergIndex", "tonality", "flatness",
                                        "logFlatness", "buggyRollOff", "useLogSpectrum"};
      for (const char *o : on) ok = ok && getInt(o) != 0;
      for (const char *o : off) ok = ok && getInt(o) == 0;
      const char *fr = getStr("freqRange");
      ok = ok && fr && !strcmp(fr, "0-0");
      plain_ = ok ? 1 : 0;
    }
    const int fc = getFconf(idxi);
    if (gemaps_ < 0) {
      // the two GeMAPS option sets (GeMAPSv01b_core.lld.conf.inc [gemapsv01b_logSpectral], eGeMAPSv02_core.lld.conf.inc
      // [egemapsv02_logSpectral_flux]): log-spectrum slopes 0-500 / 500-1500 + alpha ratio + Hammarberg index, or flux alone
      static const char *const slopes[2] = {"0-500", "500-1500"};
      static const char *const off[] = {"specDiff", "specPosDiff", "fluxCentroid", "fluxAtFluxCentroid", "centroid", "maxPos", "minPos",
                                        "entropy", "standardDeviation", "variance", "skewness", "kurtosis", "slope", "sharpness",
                                        "tonality", "harmonicity", "flatness", "logFlatness", "buggyRollOff", "oldSlopeScale"};
      bool ok = getInt("squareInput") == 1 && getInt("useLogSpectrum") == 1 && getInt("normBandEnergies") == 1 &&
                getArraySize("bands") <= 0 && getArraySize("rollOff") <= 0 && getDouble("specFloor") == 0.0000001;
      for (const char *o : off) ok = ok && getInt(o) == 0;
      const char *fr = getStr("freqRange");
      ok = ok && fr && !strcmp(fr, "0-5000");
      gemaps_ = 0;
      if (ok && array_is("slopes", 2, slopes) && getInt("alphaRatio") == 1 && getInt("hammarbergIndex") == 1 && getInt("flux") == 0)
        gemaps_ = 1;                                     // 4 outputs
      else if (ok && getArraySize("slopes") <= 0 && getInt("alphaRatio") == 0 && getInt("hammarbergIndex") == 0 && getInt("flux") == 1)
        gemaps_ = 2;                                     // 1 output
    }
    if (gemaps_ > 0 && (Nsrc == 129 || Nsrc == 257 || Nsrc == 513) && Ndst == (gemaps_ == 1 ? 4 : 1) && fc >= 0 && fc < 8) {
      if (!gm_plan_) {
        const sDmLevelConfig *lc = reader_->getLevelConfig();
        smilehip_lld_config c;
        smilehip_config_egemapsv02(&c);
        c.sample_rate = std::round(2.0 * (double)(Nsrc - 1) / lc->frameSizeSec);       // the spectrum level's frameSizeSec = Nfft / rate
        if (!(c.sample_rate >= 7999.0 && c.sample_rate <= 48001.0))
          COMP_ERR("libsmilehip plugin: cSpectral (GeMAPS options): the HIP path is built for 20 ms frames at 8 .. 48 kHz (this level: %ld bins, %g s)", Nsrc, lc->frameSizeSec);
        check(smilehip_plan_create(context(), &c, &gm_plan_));
      }
      io_.ensure(Nsrc, 5);
      io_.up(src, Nsrc);
      float *d_prev = (float *)prev_[fc].ensure(sizeof(float) * (uint64_t)Nsrc);
      check(smilehip_spectral_gemaps_frames(gm_plan_, io_.d_in, Nsrc, d_prev, seen_[fc] ? 0 : 1, io_.d_out, 5, g_blk.n, nullptr));
      seen_[fc] = true;
      const float *five = io_.down_rows();
      for (long f = 0; f < g_blk.n; ++f, five += 5, dst += g_blk.ld_dst)
        if (gemaps_ == 1) memcpy(dst, five, sizeof(float) * 4); else dst[0] = five[4];
      g_frames[12] += g_blk.n;
      return (int)Ndst;
    }
    const bool compare_set = plain_ && (Nsrc == 129 || Nsrc == 257 || Nsrc == 513) && Ndst == 12 + (int)sel_[0] + (int)sel_[1] + (int)sel_[2];
    if (!compare_set && general_ < 0) {
      // everything the linear-spectrum branch of spectral.cpp:586-1560 offers except the alphaRatio / hammarbergIndex / tonality outputs
      // (round 6: + slopes[], specDiff, specPosDiff, fluxCentroid, fluxAtFluxCentroid, standardDeviation)
      std::memset(&gen_opts_, 0, sizeof(gen_opts_));
      const int nb = getArraySize("bands") > 0 ? getArraySize("bands") : 0, nr = getArraySize("rollOff") > 0 ? getArraySize("rollOff") : 0;
      const int nsl = getArraySize("slopes") > 0 ? getArraySize("slopes") : 0;
      bool ok = nb <= 16 && nr <= 16 && nsl <= 16;
      for (int b = 0; ok && b < nsl; ++b) {              // slopes[b] = "lo-hi" in Hz, integers (spectral.cpp:327-338)
        const char *v = getStr_f(myvprint("slopes[%i]", b));
        int lo = -1, hi = -1, used = 0;
        ok = v && sscanf(v, "%d-%d%n", &lo, &hi, &used) == 2 && v[used] == 0 && lo >= 0 && hi > lo;
        gen_opts_.slope_lo[b] = lo; gen_opts_.slope_hi[b] = hi;
      }
      gen_opts_.n_slopes = nsl;
      gen_opts_.spec_diff = getInt("specDiff"); gen_opts_.spec_pos_diff = getInt("specPosDiff"); gen_opts_.flux_centroid = getInt("fluxCentroid");
      gen_opts_.flux_at_flux_centroid = getInt("fluxAtFluxCentroid"); gen_opts_.standard_deviation = getInt("standardDeviation");
      for (int b = 0; ok && b < nb; ++b) {               // bands[b] = "lo-hi" in Hz, integers (spectral.cpp:163-190)
        const char *v = getStr_f(myvprint("bands[%i]", b));
        int lo = -1, hi = -1, used = 0;
        ok = v && sscanf(v, "%d-%d%n", &lo, &hi, &used) == 2 && v[used] == 0 && lo >= 0 && hi > lo;
        gen_opts_.band_lo[b] = lo; gen_opts_.band_hi[b] = hi;
      }
      for (int i = 0; ok && i < nr; ++i) gen_opts_.rolloff[i] = getDouble_f(myvprint("rollOff[%i]", i));
      gen_opts_.n_bands = nb; gen_opts_.n_rolloff = nr;
      gen_opts_.flux = getInt("flux"); gen_opts_.centroid = getInt("centroid"); gen_opts_.max_pos = getInt("maxPos"); gen_opts_.min_pos = getInt("minPos");
      gen_opts_.entropy = getInt("entropy"); gen_opts_.variance = getInt("variance"); gen_opts_.skewness = getInt("skewness");
      gen_opts_.kurtosis = getInt("kurtosis"); gen_opts_.slope = getInt("slope"); gen_opts_.sharpness = getInt("sharpness");
      gen_opts_.harmonicity = getInt("harmonicity"); gen_opts_.flatness = getInt("flatness"); gen_opts_.log_flatness = getInt("logFlatness");
      static const char *const off[] = {"normBandEnergies", "alphaRatio", "hammarbergIndex", "tonality", "buggyRollOff", "useLogSpectrum"};
      for (const char *o : off) ok = ok && getInt(o) == 0;
      ok = ok && getInt("squareInput") != 0 && ((!gen_opts_.slope && !nsl) || getInt("oldSlopeScale") != 0);
      const char *fr = getStr("freqRange");
      ok = ok && fr && !strcmp(fr, "0-0");
      gen_n_out_ = ok ? smilehip_spectral_opts_count(&gen_opts_) : 0;
      general_ = (ok && gen_n_out_ > 0) ? 1 : 0;
    }
    if (!compare_set && general_ == 1 && Nsrc >= 9 && ((Nsrc - 1) & (Nsrc - 2)) == 0 && Ndst == gen_n_out_ && fc >= 0 && fc < 8) {   // (2^k + 1 bins: an FFT magnitude level, linear axis)
      if (!gen_op_[fc]) check(smilehip_spectral_op_create(context(), &gen_opts_, Nsrc, reader_->getLevelConfig()->frameSizeSec, &gen_op_[fc]));
      io_.ensure(Nsrc, gen_n_out_);
      io_.up(src, Nsrc);
      float *d_prev = (float *)prev_[fc].ensure(sizeof(float) * (uint64_t)Nsrc);
      check(smilehip_spectral_op_frames(gen_op_[fc], io_.d_in, Nsrc, d_prev, seen_[fc] ? 0 : 1, io_.d_out, gen_n_out_, g_blk.n, nullptr));
      seen_[fc] = true;
      io_.down(dst, gen_n_out_);
      g_frames[12] += g_blk.n;
      return (int)Ndst;
    }
    if (!compare_set || fc < 0 || fc >= 8) {
      HIP_FALLTHROUGH(12, "cSpectral: the linear-spectrum descriptor sets (bands, slopes, rollOff points, specDiff, specPosDiff, flux, fluxCentroid, fluxAtFluxCentroid, centroid, "
                          "maxPos, minPos, entropy, standardDeviation, variance, skewness, kurtosis, slope, sharpness, harmonicity, flatness; freqRange 0-0) and the two GeMAPS sets (log-spectrum slopes + alphaRatio + hammarbergIndex; flux over 0-5000 Hz) are built");
      return cSpectral::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    smilehip_plan *&pl = plans_.at(fc);
    if (!pl) {
      const sDmLevelConfig *lc = reader_->getLevelConfig();
      smilehip_lld_config c = base_config((Nsrc - 1) * 2, SMILEHIP_STAGE_SPECTRAL);
      c.force_fft_frame_size_sec = lc->frameSizeSec;    // fsSec, spectral.cpp:382-385
      for (int b = 0; b < 2; ++b) { c.spectral_band_lo[b] = band_lo_[b]; c.spectral_band_hi[b] = band_hi_[b]; }
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, 15);
    io_.up(src, Nsrc);
    float *d_prev = (float *)prev_[fc].ensure(sizeof(float) * (uint64_t)Nsrc);
    check(smilehip_spectral_frames(pl, io_.d_in, Nsrc, d_prev, seen_[fc] ? 0 : 1, io_.d_out, 15, g_blk.n, nullptr));
    seen_[fc] = true;
    if (Ndst == 15) io_.down(dst, 15);
    else {
      const float *v = io_.down_rows();
      for (long f = 0; f < g_blk.n; ++f, v += 15, dst += g_blk.ld_dst) {
        long n = 0;
        for (int k = 0; k < 15; ++k)
          if ((k != 7 || sel_[0]) && (k != 13 || sel_[1]) && (k != 14 || sel_[2])) dst[n++] = v[k];
      }
    }
    g_frames[12] += g_blk.n;
    return (int)Ndst;
  }
 public:
  explicit cHipSpectral(const char *n) : BlockVP<cSpectral>(n) {}
  ~cHipSpectral() override {
    if (gm_plan_) smilehip_plan_destroy(gm_plan_);
    for (auto *op : gen_op_) if (op) smilehip_spectral_op_destroy(op);
  }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipSpectral(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R8  cPlp::processVector as auditory spectrum, with or without newRASTA  (src/lldcore/plp.cpp:416-593)
class cHipPlp : public BlockVP<cPlp> {
  int fused_ = -1;
  const FusedLevel *fcols_ = nullptr;
  long fnext_ = 0;
  FrameIO io_;
  bool cpu_warned_ = false;
  DevBytes eql_[8], state_[8], cos_[8], sin_[8];
  bool ready_[8] = {false, false, false, false, false, false, false, false};
  int plain_ = -1, newRasta_ = 0, oldRasta_ = 0, cc_ = 0, part_ = 0, lpOrder_ = 0, firstCC_ = 0, htk_ = 0;
  FLOAT_DMEM compression_ = 0, melfloor_ = 0;
  float coef_[6] = {0, 0, 0, 0, 0, 0};
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
    return BlockVP<cPlp>::myTick(t);
  }
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (plain_ < 0) {                                    // cPlp::myFetchConfig, plp.cpp:90-176
      int doLP = getInt("doLP"), doLpToCeps = getInt("doLpToCeps"), doIDFT = getInt("doIDFT");
      if (getInt("lpOrder") <= 0) { doLP = 0; doLpToCeps = 0; }
      if (doLpToCeps) doLP = 1;
      if (doLP) doIDFT = 1;
      newRasta_ = getInt("newRASTA");
      const int rasta = newRasta_ ? 0 : getInt("RASTA");
      oldRasta_ = rasta;
      compression_ = (FLOAT_DMEM)getDouble("compression");
      if (compression_ < 0.0) compression_ = 0.0;
      melfloor_ = (FLOAT_DMEM)getDouble("melfloor");
      htk_ = getInt("htkcompatible") ? 1 : 0;              // forces melfloor = 1, doAud = 1, doLog = doInvLog = 0 (plp.cpp:151-161)
      const bool logs_ok = (newRasta_ || rasta) ? true : (htk_ || (!getInt("doLog") && !getInt("doInvLog")));   // (either RASTA form forces doLog = doInvLog = 1, :168-174)
      plain_ = ((htk_ || getInt("doAud")) && !doIDFT && !doLP && logs_ok && !(htk_ && (newRasta_ || rasta))) ? 1 : 0;
      if (htk_) melfloor_ = 1.0;                           // the HTK-style auditory spectrum alone (config/audspec/audspec.conf)
      // PLP cepstra in HTK mode (config/plp/*.conf): doAud -> IDFT -> LP -> cepstra, c0 last
      lpOrder_ = getInt("lpOrder");
      const int lastCC = getInt("lastCC"), nCeps = getInt("nCeps");
      firstCC_ = (int)getInt("firstCC");                   // 1 (config/plp/PLP_E_*): c1 .. c12 -- the same values without the trailing c0
      cc_ = (getInt("htkcompatible") && doIDFT && doLP && doLpToCeps && !rasta && !newRasta_ && (firstCC_ == 0 || firstCC_ == 1) &&
             lpOrder_ >= 1 && lpOrder_ <= 15 && (lastCC < 0 || lastCC == lpOrder_) && (nCeps < 0 || nCeps == lpOrder_ + 1 - firstCC_)) ? 1 : 0;
      // round 6: the same chain cut short (plp.cpp:573-583) -- the autocorrelation (doIDFT = 1, doLP = 0) or the LP coefficients
      // (doLP = 1, doLpToCeps = 0) as the component's output
      part_ = 0;
      if (getInt("htkcompatible") && doIDFT && !doLpToCeps && !rasta && !newRasta_ && lpOrder_ >= 1 && lpOrder_ <= 15) part_ = doLP ? 2 : 1;
      if (part_) cc_ = 1;                                 // (the tables and the equal-loudness weights are the cepstral mode's)
      if (cc_) { plain_ = 1; melfloor_ = 1.0; }           // htkcompatible forces melfloor = 1, doAud = 1, no logs (plp.cpp:150-160)
      if (newRasta_ || oldRasta_) {                      // initTables, plp.cpp:361-399 (the same coefficients for both forms)
        const FLOAT_DMEM lo = (FLOAT_DMEM)getDouble("rastaLowerCutoff"), up = (FLOAT_DMEM)getDouble("rastaUpperCutoff");
        coef_[0] = (FLOAT_DMEM)(1.0 - sin(2.0 * M_PI * lo * reader_->getLevelT()));
        const FLOAT_DMEM om = (FLOAT_DMEM)cos(2.0 * M_PI * up * reader_->getLevelT());
        const FLOAT_DMEM norm = (FLOAT_DMEM)sqrt(10.0 * (32.0 * om * om + 8.0));
        coef_[1] = (FLOAT_DMEM)(2.0 / norm);
        coef_[2] = (FLOAT_DMEM)(-4.0 * om / norm);
        coef_[3] = 0.0;
        coef_[4] = -coef_[2];
        coef_[5] = -coef_[1];
      }
    }
    const int fc = getFconf(idxi);
    const FrameMetaInfo *fmeta = reader_->getFrameMetaInfo();
    const long n_cc = part_ == 1 ? lpOrder_ + 1 : (part_ == 2 ? lpOrder_ : lpOrder_ + 1 - firstCC_);
    if (!plain_ || (cc_ ? Ndst != n_cc : Nsrc != Ndst) || Nsrc > 64 || fc < 0 || fc >= 8 || !fmeta || idxi >= fmeta->N ||
        (long)(fmeta->field[idxi].infoSize / sizeof(double)) != Nsrc)
      { HIP_FALLTHROUGH(13, "cPlp: only the auditory spectrum (plain, RASTA, newRASTA) and the HTK modes (PLP-CC, or the chain cut behind the IDFT / the LP analysis) are built"); return cPlp::processVector(src, dst, Nsrc, Ndst, idxi); }
    if (!ready_[fc]) {                                   // equal-loudness curve at the band centres, plp.cpp:335-357
      const double *frq = (const double *)(fmeta->field[idxi].info);
      std::vector<float> e((size_t)Nsrc), st((size_t)(6 * Nsrc + 2), 0.0f);
      for (long i = 0; i < Nsrc; ++i) {
        e[(size_t)i] = (cc_ || htk_) ? (FLOAT_DMEM)smileDsp_equalLoudnessWeight_htk((double)frq[i])
                           : (FLOAT_DMEM)smileDsp_equalLoudnessWeight((double)frq[i]);
        if (newRasta_ || oldRasta_) e[(size_t)i] = log(e[(size_t)i]);
      }
      if (cc_) {                                         // IDFT cosine table and lifter, plp.cpp:288-334
        const int nFreq = (int)Nsrc + 2, nAuto = lpOrder_ + 1;
        std::vector<float> ct((size_t)nAuto * nFreq), sn((size_t)nAuto);
        const FLOAT_DMEM a = (FLOAT_DMEM)M_PI / (FLOAT_DMEM)(nFreq - 1);
        for (int i = 0; i < nAuto; i++) {
          const int ib = i * nFreq;
          int m;
          ct[(size_t)ib] = 1.0;
          for (m = 1; m < (nFreq - 1); m++) ct[(size_t)(m + ib)] = (FLOAT_DMEM)(2.0 * cos(a * (double)i * (double)m));
          ct[(size_t)(m + ib)] = (FLOAT_DMEM)(cos(a * (double)i * (double)m));
        }
        const FLOAT_DMEM L = (FLOAT_DMEM)getInt("cepLifter");
        for (int i = 0; i < nAuto; i++)
          sn[(size_t)i] = (L > 0.0) ? ((FLOAT_DMEM)1.0 + L / (FLOAT_DMEM)2.0 * sin((FLOAT_DMEM)M_PI * ((FLOAT_DMEM)(i)) / L)) : (FLOAT_DMEM)1.0;
        void *d_c = cos_[fc].ensure(sizeof(float) * ct.size());
        void *d_n = sin_[fc].ensure(sizeof(float) * sn.size());
        if (smilehip_copy_to_device(context(), d_c, ct.data(), sizeof(float) * ct.size(), nullptr) ||
            smilehip_copy_to_device(context(), d_n, sn.data(), sizeof(float) * sn.size(), nullptr))
          COMP_ERR("libsmilehip: %s", smilehip_last_error());
      }
      void *d_e = eql_[fc].ensure(sizeof(float) * e.size());
      void *d_s = state_[fc].ensure(sizeof(float) * st.size());
      if (smilehip_copy_to_device(context(), d_e, e.data(), sizeof(float) * e.size(), nullptr) ||
          smilehip_copy_to_device(context(), d_s, st.data(), sizeof(float) * st.size(), nullptr))
        COMP_ERR("libsmilehip: %s", smilehip_last_error());
      ready_[fc] = true;
    }
    io_.ensure(Nsrc, cc_ ? lpOrder_ + 1 : Ndst);
    io_.up(src, Nsrc);
    if (part_)
      check(smilehip_plp_stage_frames(context(), io_.d_in, Nsrc, (int)Nsrc, (const float *)eql_[fc].d, melfloor_, compression_, lpOrder_,
                                      (const float *)cos_[fc].d, part_, io_.d_out, lpOrder_ + 1, g_blk.n, nullptr));
    else if (cc_)
      check(smilehip_plp_cc_frames(context(), io_.d_in, Nsrc, (int)Nsrc, (const float *)eql_[fc].d, melfloor_, compression_, lpOrder_,
                                   (const float *)cos_[fc].d, (const float *)sin_[fc].d, io_.d_out, lpOrder_ + 1, g_blk.n, nullptr));
    else
      check(smilehip_plp_audspec_frames(context(), io_.d_in, Nsrc, (int)Nsrc, (const float *)eql_[fc].d, melfloor_, compression_,
                                        newRasta_ ? 1 : (oldRasta_ ? 2 : 0), coef_, (float *)state_[fc].d, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[13] += g_blk.n;
    return (int)Ndst;
  }
 public:
  explicit cHipPlp(const char *n) : BlockVP<cPlp>(n) {}
  ~cHipPlp() override { delete fblock_; }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipPlp(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};
