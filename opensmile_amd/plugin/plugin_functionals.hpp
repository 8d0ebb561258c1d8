// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): SURVEY 8(f) rank 1: cFunctionals
// SURVEY 8f rank 1  cFunctionals::doProcess (src/functionals/functionals.cpp:320-389): one input row (one LLD contour)
// per call through smilehip_funcspec_matrix. The instance's configuration -- functionalsEnabled and the options of the
// cFunctional* children, time norms resolved with the reference's precedence (the family's own `norm` if set, else
// masterTimeNorm, else the family's default; functionalComponent.hpp:68-76) -- is translated into a smilehip_func_spec
// once; an instance that uses an option the spec cannot express stays on the reference's own code.
class cHipFunctionals : public cFunctionals {
  int fused_ = -1;
  FusedChain::FuncAt fat_{0, 0};
  FrameIO io_;
  bool cpu_warned_ = false;
  int state_ = -1;                                       // -1 = not examined, 0 = not expressible -> reference code, 1 = spec_
  smilehip_func_spec spec_;
  std::vector<float> all_;                               // the values of every contour of the current tick's matrix ([contour][functional])
  const cMatrix *all_of_ = nullptr;
  long all_nT_ = 0, all_n_ = 0;
  int opt_int(const char *fam, const char *o) { return (int)getInt_f(myvprint("%s.%s", fam, o)); }
  double opt_dbl(const char *fam, const char *o) { return getDouble_f(myvprint("%s.%s", fam, o)); }
  bool opt_set(const char *fam, const char *o) {
    char *k = myvprint("%s.%s", fam, o);
    const bool r = isSet(k) != 0;
    free(k);
    return r;
  }
  static int parse_norm(const char *n, int fallback) {
    if (!n) return fallback;
    if (!strncmp(n, "tur", 3) || !strncmp(n, "seg", 3)) return SMILEHIP_NORM_SEGMENT;
    if (!strncmp(n, "sec", 3)) return SMILEHIP_NORM_SECOND;
    if (!strncmp(n, "fra", 3)) return SMILEHIP_NORM_FRAME;
    return fallback;
  }
  int time_norm(const char *fam) {                        // parseTimeNormOption + setTimeNorm
    const int own = parse_norm(getStr_f(myvprint("%s.norm", fam)), SMILEHIP_NORM_SEGMENT);
    if (opt_set(fam, "norm")) return own;
    if (isSet("masterTimeNorm")) {
      const char *m = getStr("masterTimeNorm");
      if (m && (!strncmp(m, "seg", 3) || !strncmp(m, "tur", 3) || !strncmp(m, "sec", 3) || !strncmp(m, "fra", 3)))
        return parse_norm(m, own);
    }
    return own;
  }
  uint32_t mask_of(const char *fam, const char *const *names, int n) {
    uint32_t m = 0;
    for (int k = 0; k < n; ++k)
      if (opt_int(fam, names[k])) m |= 1u << k;
    return m;
  }
  bool build_spec() {
    smilehip_func_spec &s = spec_;
    std::memset(&s, 0, sizeof(s));
    s.period = getInputPeriod();
    if (!(s.period > 0.0)) s.period = 1.0;                // only second-normalised values use it
    s.non_zero_functs = (int)getInt("nonZeroFuncts");
    s.ext_norm = s.means_norm = s.times_norm = s.seg_norm = s.pk_norm = s.ons_norm = s.pko_norm = SMILEHIP_NORM_SEGMENT;
    s.reg_centroid_norm = SMILEHIP_NORM_SEGMENT;
    s.seg_max_num = 20; s.seg_min_lng = 3; s.seg_pause_min_lng = 2; s.lpc_order = 5;
    const int n = getArraySize("functionalsEnabled");
    if (n < 1 || n > 12) return false;
    for (int i = 0; i < n; ++i) {
      const char *f = getStr_f(myvprint("functionalsEnabled[%i]", i));
      if (!f) return false;
      if (!strcmp(f, "Extremes")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_EXTREMES;
        static const char *const o[8] = {"max", "min", "range", "maxpos", "minpos", "amean", "maxameandist", "minameandist"};
        s.ext_mask = mask_of(f, o, 8);
        s.ext_norm = time_norm(f);
      } else if (!strcmp(f, "Means")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_MEANS;
        static const char *const o[17] = {"amean", "absmean", "qmean", "nzamean", "nzabsmean", "nzqmean", "nzgmean", "nnz", "flatness",
                                          "posamean", "negamean", "posqmean", "posrqmean", "negqmean", "negrqmean", "rqmean", "nzrqmean"};
        s.means_mask = mask_of(f, o, 17);
        s.means_norm = time_norm(f);
      } else if (!strcmp(f, "Moments")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_MOMENTS;
        static const char *const o[5] = {"variance", "stddev", "skewness", "kurtosis", "amean"};
        s.mom_mask = mask_of(f, o, 5);
        s.mom_stddev_norm = opt_int(f, "stddevNorm");
        if (s.mom_stddev_norm == 1 || s.mom_stddev_norm == 2) s.mom_mask |= 1u << 5;
        s.mom_ratio_limit = opt_int(f, "doRatioLimit");
      } else if (!strcmp(f, "Regression")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_REGRESSION;
        static const char *const o[18] = {"linregc1", "linregc2", "linregerrA", "linregerrQ", "qregc1", "qregc2", "qregc3", "qregerrA",
                                          "qregerrQ", "centroid", "qregls", "qregrs", "qregx0", "qregy0", "qregyr", "qregy0nn",
                                          "qregc3nn", "qregyrnn"};
        s.reg_mask = mask_of(f, o, 18);
        const char *cn = getStr_f(myvprint("%s.centroidNorm", f));
        if (!cn || (strncmp(cn, "sec", 3) && strncmp(cn, "fra", 3) && strncmp(cn, "seg", 3))) return false;
        s.reg_centroid_norm = parse_norm(cn, SMILEHIP_NORM_SEGMENT);
        s.reg_norm_coeff = opt_int(f, "normRegCoeff");
        s.reg_norm_inputs = opt_int(f, "normInputs");
        s.reg_centroid_abs = opt_int(f, "centroidUseAbsValues");
        s.reg_centroid_limit = opt_int(f, "centroidRatioLimit");
        s.reg_ratio_limit = opt_int(f, "doRatioLimit");
        s.reg_old_buggy_qerr = opt_int(f, "oldBuggyQerr");
      } else if (!strcmp(f, "Percentiles")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_PERCENTILES;
        static const char *const q[3] = {"quartile1", "quartile2", "quartile3"}, *const r[3] = {"iqr12", "iqr23", "iqr13"};
        uint32_t m = 0;
        for (int k = 0; k < 3; ++k) if (opt_int(f, q[k])) m |= 1u << k;
        if (opt_set(f, "quartiles")) m = opt_int(f, "quartiles") ? 7u : 0u;
        uint32_t mr = 0;
        for (int k = 0; k < 3; ++k) if (opt_int(f, r[k])) mr |= 1u << (3 + k);
        if (opt_set(f, "iqr")) mr = opt_int(f, "iqr") ? 0x38u : 0u;
        s.pct_mask = m | mr;
        s.pct_interp = opt_int(f, "interp");
        char *k = myvprint("%s.percentile", f);
        s.n_pctl = getArraySize(k); free(k);
        k = myvprint("%s.pctlrange", f);
        s.n_range = getArraySize(k); free(k);
        k = myvprint("%s.pctlquotient", f);
        const int nq = getArraySize(k); free(k);
        if (s.n_pctl < 0 || s.n_pctl > 8 || s.n_range < 0 || s.n_range > 8 || nq < 0 || nq > 8) return false;
        if (s.n_pctl == 0) s.n_range = 0;
        s.n_quot = s.n_pctl > 0 ? nq : 0;                 // (read inside `if (nPctl > 0)`, functionalPercentiles.cpp:114-232)
        for (int j = 0; j < s.n_quot; ++j) {
          const char *t = getStr_f(myvprint("%s.pctlquotient[%i]", f, j));
          int a = -1, b = -1;
          if (!t || sscanf(t, "%d-%d", &a, &b) != 2 || a < 0 || b < 0 || a >= s.n_pctl || b >= s.n_pctl) return false;
          s.quot_a[j] = a; s.quot_b[j] = b;
        }
        for (int j = 0; j < s.n_pctl; ++j) {
          double v = getDouble_f(myvprint("%s.percentile[%i]", f, j));
          s.pctl[j] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        }
        for (int j = 0; j < s.n_range; ++j) {
          const char *t = getStr_f(myvprint("%s.pctlrange[%i]", f, j));
          int a = -1, b = -1;
          if (!t || sscanf(t, "%d-%d", &a, &b) != 2) return false;
          s.range_a[j] = a; s.range_b[j] = b;
        }
      } else if (!strcmp(f, "Times")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_TIMES;
        static const char *const o[13] = {"upleveltime25", "downleveltime25", "upleveltime50", "downleveltime50", "upleveltime75",
                                          "downleveltime75", "upleveltime90", "downleveltime90", "risetime", "falltime", "leftctime",
                                          "rightctime", "duration"};
        s.times_mask = mask_of(f, o, 13);
        s.times_norm = time_norm(f);
        s.times_buggy_sec_norm = opt_int(f, "buggySecNorm");
        char *k = myvprint("%s.upleveltime", f);
        const int nu = getArraySize(k); free(k);
        k = myvprint("%s.downleveltime", f);
        const int nd = getArraySize(k); free(k);
        if (nu < 0 || nu > 8 || nd < 0 || nd > 8 || opt_int(f, "useRobustPercentileRange")) return false;
        s.n_ul = nu; s.n_dl = nd;
        for (int j = 0; j < nu; ++j) { const double v = getDouble_f(myvprint("%s.upleveltime[%i]", f, j)); s.ul[j] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
        for (int j = 0; j < nd; ++j) { const double v = getDouble_f(myvprint("%s.downleveltime[%i]", f, j)); s.dl[j] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
      } else if (!strcmp(f, "Segments")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_SEGMENTS;
        static const char *const o[5] = {"numSegments", "meanSegLen", "maxSegLen", "minSegLen", "segLenStddev"};
        s.seg_mask = mask_of(f, o, 5);
        s.seg_norm = time_norm(f);
        const char *alg = getStr_f(myvprint("%s.segmentationAlgorithm", f));
        if (!alg) return false;
        // the reference's own prefix tests in its own order (functionalSegments.cpp:120-155); ltX / gtX / geqX / leqX and any
        // unknown name end in the delta method there (:152-154, :872-874)
        if (!strncmp(alg, "delta", 5)) s.seg_algo = SMILEHIP_SEG_DELTA;
        else if (!strncmp(alg, "delt2", 5)) s.seg_algo = SMILEHIP_SEG_DELTA2;
        else if (!strncmp(alg, "relTh", 5)) s.seg_algo = SMILEHIP_SEG_RELTH;
        else if (!strncmp(alg, "mrelTh", 6)) s.seg_algo = SMILEHIP_SEG_MRELTH;
        else if (!strncmp(alg, "absTh", 5)) s.seg_algo = SMILEHIP_SEG_ABSTH;
        else if (!strncmp(alg, "NArelTh", 7)) s.seg_algo = SMILEHIP_SEG_NARELTH;
        else if (!strncmp(alg, "mNArelTh", 8) || !strncmp(alg, "NAmrelTh", 8)) s.seg_algo = SMILEHIP_SEG_NAMRELTH;
        else if (!strncmp(alg, "NAabsTh", 7)) s.seg_algo = SMILEHIP_SEG_NAABSTH;
        else if (!strncmp(alg, "chX", 3)) s.seg_algo = SMILEHIP_SEG_CHX;
        else if (!strncmp(alg, "nonX", 4)) s.seg_algo = SMILEHIP_SEG_NONX;
        else if (!strncmp(alg, "eqX", 3)) s.seg_algo = SMILEHIP_SEG_EQX;
        else s.seg_algo = SMILEHIP_SEG_DELTA;
        if (opt_int(f, "growDynSegBuffer") || opt_int(f, "useOldBuggyChX")) return false;
        s.seg_max_num = opt_int(f, "maxNumSeg");
        s.seg_min_lng = opt_int(f, "segMinLng");
        if (s.seg_min_lng < 1) s.seg_min_lng = 1;
        s.seg_auto_min_lng = opt_set(f, "segMinLng") ? 0 : 1;
        s.seg_pause_min_lng = opt_int(f, "pauseMinLng");
        if (s.seg_pause_min_lng < 1) s.seg_pause_min_lng = 1;
        s.seg_x = (float)opt_dbl(f, "X");
        s.seg_x_is_rel = opt_int(f, "XisRel");
        s.seg_range_rel_threshold = (float)opt_dbl(f, "rangeRelThreshold");
        s.seg_ravg_lng = opt_int(f, "ravgLng");
        if ((s.seg_algo == SMILEHIP_SEG_DELTA || s.seg_algo == SMILEHIP_SEG_DELTA2) && s.seg_ravg_lng <= 0 && s.seg_max_num < 2)
          return false;                                  // Nin / (maxNumSeg / 2): the reference divides by zero
        if (s.seg_algo == SMILEHIP_SEG_RELTH || s.seg_algo == SMILEHIP_SEG_NARELTH || s.seg_algo == SMILEHIP_SEG_NAABSTH ||
            s.seg_algo == SMILEHIP_SEG_MRELTH || s.seg_algo == SMILEHIP_SEG_NAMRELTH) {                           // :179-200
          const bool clamp = s.seg_algo == SMILEHIP_SEG_RELTH || s.seg_algo == SMILEHIP_SEG_NARELTH;
          char *k = myvprint("%s.thresholds", f);
          s.seg_n_thresholds = getArraySize(k); free(k);
          if (s.seg_n_thresholds < 0 || s.seg_n_thresholds > 8) return false;
          for (int j = 0; j < s.seg_n_thresholds; ++j) {
            float v = (float)getDouble_f(myvprint("%s.thresholds[%i]", f, j));
            s.seg_thresholds[j] = !clamp ? v : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
          }
        }
      } else if (!strcmp(f, "Lpc")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_LPC;
        s.lpc_first = opt_int(f, "firstCoeff");
        if (s.lpc_first < 0) s.lpc_first = 0;
        s.lpc_order = opt_int(f, "order");
        if (s.lpc_order <= s.lpc_first) return false;
        s.lpc_gain = opt_int(f, "lpGain") ? 1 : 0;
        s.lpc_coeffs = opt_int(f, "lpc") ? 1 : 0;
      } else if (!strcmp(f, "Peaks2")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_PEAKS2;
        static const char *const o[32] = {"numPeaks", "meanPeakDist", "meanPeakDistDelta", "peakDistStddev", "peakRangeAbs", "peakRangeRel",
                                          "peakMeanAbs", "peakMeanMeanDist", "peakMeanRel", "ptpAmpMeanAbs", "ptpAmpMeanRel",
                                          "ptpAmpStddevAbs", "ptpAmpStddevRel", "minRangeAbs", "minRangeRel", "minMeanAbs",
                                          "minMeanMeanDist", "minMeanRel", "mtmAmpMeanAbs", "mtmAmpMeanRel", "mtmAmpStddevAbs",
                                          "mtmAmpStddevRel", "meanRisingSlope", "maxRisingSlope", "minRisingSlope",
                                          "stddevRisingSlope", "meanFallingSlope", "maxFallingSlope", "minFallingSlope",
                                          "stddevFallingSlope", "covFallingSlope", "covRisingSlope"};
        s.pk_mask = mask_of(f, o, 32);
        s.pk_norm = time_norm(f);
        if (opt_int(f, "noClearPeakList")) return false;
        const char *dbg = getStr_f(myvprint("%s.posDbgOutp", f));
        if ((dbg && *dbg) || opt_int(f, "consoleDbg")) return false;
        s.pk_ratio_limit = opt_int(f, "doRatioLimit");
        s.pk_dyn_rel = opt_int(f, "dynRelThresh");
        float rt = (float)opt_dbl(f, "relThresh");
        if (rt < 0) rt = 0.0f;
        else if (rt > 1.0f && !s.pk_dyn_rel) rt = 1.0f;
        s.pk_rel_thresh = rt;
        if (opt_set(f, "absThresh")) {
          s.pk_use_abs = 1;
          s.pk_abs_thresh = (float)opt_dbl(f, "absThresh");
          s.pk_dyn_rel = 0;
        }
      } else if (!strcmp(f, "Peaks")) {
        if (!opt_int(f, "overlapFlag")) return false;      // overlapFlag = 0 carries the last two values from call to call
        s.fam[s.n_fam++] = SMILEHIP_FAM_PEAKS;
        static const char *const o[5] = {"numPeaks", "meanPeakDist", "peakMean", "peakMeanMeanDist", "peakDistStddev"};
        s.pko_mask = mask_of(f, o, 5);
        s.pko_norm = time_norm(f);
      } else if (!strcmp(f, "Crossings")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_CROSSINGS;
        static const char *const o[3] = {"zcr", "mcr", "amean"};
        s.crs_mask = mask_of(f, o, 3);
      } else if (!strcmp(f, "DCT")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_DCT;
        s.dct_first = opt_int(f, "firstCoeff");
        if (s.dct_first < 0) s.dct_first = 0;
        s.dct_last = opt_set(f, "nCoeffs") ? s.dct_first + opt_int(f, "nCoeffs") - 1 : opt_int(f, "lastCoeff");
        if (s.dct_last < s.dct_first || s.dct_last - s.dct_first >= 64) return false;
      } else if (!strcmp(f, "Samples")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_SAMPLES;
        char *k = myvprint("%s.samplepos", f);
        const int ns = getArraySize(k); free(k);
        if (ns > 8) return false;
        if (ns > 0) {
          s.n_samples = ns;
          for (int j = 0; j < ns; ++j) {
            double v = getDouble_f(myvprint("%s.samplepos[%i]", f, j));
            s.sample_pos[j] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
          }
        } else {                                          // DEFAULT_NR_SAMPLES = 5 (functionalSamples.cpp:27, 78-84)
          s.n_samples = 5;
          for (int j = 0; j < 5; ++j) s.sample_pos[j] = (double)j / (5 - 1.0);
        }
      } else if (!strcmp(f, "Modulation")) {
        // cFunctionalModulation::myFetchConfig (functionalModulation.cpp:375-418) and the first call of ::process (:483-496)
        s.fam[s.n_fam++] = SMILEHIP_FAM_MODULATION;
        const double per = getInputPeriod();
        if (!(per > 0.0)) return false;                   // (the reference cannot compute it either: T = 0)
        double win_sec = opt_dbl(f, "stftWinSizeSec"), step_sec = opt_dbl(f, "stftWinStepSec");
        if (step_sec == 0.0) step_sec = win_sec;
        long wf = 0, sf = 0;
        if (opt_set(f, "stftWinSizeFrames")) { wf = opt_int(f, "stftWinSizeFrames"); win_sec = 0.0; }
        if (opt_set(f, "stftWinStepFrames")) { sf = opt_int(f, "stftWinStepFrames"); step_sec = 0.0; }
        if (sf == 0) sf = wf;
        const float T = (float)per;
        if (wf == 0) { wf = (long)(win_sec / T); sf = (long)(step_sec / T); }
        if (wf < 33 || wf > 1024 || sf < 1) return false; // (0: one transform over the whole contour -- not built)
        s.mod_win_frames = (int32_t)wf;
        s.mod_step_frames = (int32_t)sf;
        s.mod_min_freq = opt_dbl(f, "modSpecMinFreq");
        s.mod_max_freq = opt_dbl(f, "modSpecMaxFreq");
        if (opt_set(f, "modSpecNumBins")) s.mod_n_bins = opt_int(f, "modSpecNumBins");
        else s.mod_n_bins = (int)round((s.mod_max_freq - s.mod_min_freq) / opt_dbl(f, "modSpecResolution")) + 1;
        if (s.mod_n_bins < 1 || s.mod_n_bins > 128) return false;
        const char *wfn = getStr_f(myvprint("%s.fftWinFunc", f));
        int wid = winfunc_id(wfn);
        if (wid < 0 || wid == SMILEHIP_WIN_GAUSS) wid = SMILEHIP_WIN_RECT;      // allocateWinFunc :133-146: any other shape falls back to the rectangle
        s.mod_win_func = wid;
        s.mod_remove_nz_mean = opt_int(f, "removeNonZeroMean") ? 1 : 0;
      } else if (!strcmp(f, "Onset")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_ONSET;
        static const char *const o[5] = {"onsetPos", "offsetPos", "numOnsets", "numOffsets", "onsetRate"};
        s.ons_mask = mask_of(f, o, 5);
        s.ons_norm = time_norm(f);
        s.ons_use_abs = opt_int(f, "useAbsVal");
        s.ons_thr_on = s.ons_thr_off = (float)opt_dbl(f, "threshold");            // functionalOnset.cpp:72-76
        if (opt_set(f, "thresholdOnset")) s.ons_thr_on = (float)opt_dbl(f, "thresholdOnset");
        if (opt_set(f, "thresholdOffset")) s.ons_thr_off = (float)opt_dbl(f, "thresholdOffset");
      } else {
        return false;                                     // a family that is not built
      }
    }
    return smilehip_funcspec_count(&s) == nFunctValues;
  }
 protected:
  int doProcess(int i, cMatrix *row, FLOAT_DMEM *y) override {
    if (fused_ < 0) {                                      // big-set fused mode: the instance's values of the fused batch's functionals vector
      g_fused.init();
      fused_ = 0;
      if (g_fused.big) {
        auto it = g_fused.func_levels.find(getStr("writer.dmLevel"));
        if (it != g_fused.func_levels.end() && it->second.count == nFunctValues) { fat_ = it->second; fused_ = 1; }
      }
    }
    if (fused_ && row->nT > 0) {
      const long at = fat_.base + (long)i * fat_.count;
      if (at + fat_.count > (long)g_fused.func.size()) COMP_ERR("libsmilehip plugin: fused mode: functionals element %d outside the batch's vector", i);
      for (int k = 0; k < fat_.count; ++k) y[k] = g_fused.func[(size_t)(at + k)];
      g_fused.served++;
      return nFunctValues;
    }
    if (state_ < 0) state_ = build_spec() ? 1 : 0;
    if (!state_ || row->nT <= 0) { if (!state_) HIP_FALLTHROUGH(14, "cFunctionals: a functional family or option of this instance is not built (Modulation over the whole contour, Times.useRobustPercentileRange, Segments.growDynSegBuffer, Peaks.overlapFlag = 0, ...)"); return cFunctionals::doProcess(i, row, y); }
    // The Modulation family on a contour of fewer than 34 values: the reference transforms it with 4 .. 32 points, which the
    // library does not build (include/smilehip.h) -- refused by name (or the reference's own code under SMILEHIP_PLUGIN_ALLOW_CPU=1),
    // never answered with NaN
    if (row->nT < 34) {
      bool has_mod = false;
      for (int q = 0; q < spec_.n_fam; ++q) has_mod = has_mod || spec_.fam[q] == SMILEHIP_FAM_MODULATION;
      if (has_mod) { HIP_FALLTHROUGH(14, "cFunctionals: Modulation on a contour of fewer than 34 values (4 .. 32-point transforms are not built)"); return cFunctionals::doProcess(i, row, y); }
    }
    // Every contour of the tick in ONE operator call (round 6): cWinToVecProcessor::myTick hands doProcess the rows of ONE matrix one
    // after the other (winToVecProcessor.cpp:1037-1052), and that matrix is still the reader's (cDataReader::m, dataReader.cpp:446-536).
    // At the tick's first row the whole matrix goes to the device and the operator runs over all its columns -- a thread per contour
    // instead of one thread in all --, the later rows take their values from that result. (A row that is not the matrix's -- checked
    // on its first and last value -- takes the single-contour call below.)
    {
      cMatrix *whole = reader_->m;
      const long nT = row->nT;
      if (block_mode() && whole && whole->nT == nT && whole->N > 1 && i >= 0 && i < whole->N && whole->N <= 4096 &&
          !memcmp(&whole->data[i], &row->data[0], sizeof(FLOAT_DMEM)) &&
          !memcmp(&whole->data[(size_t)(nT - 1) * (size_t)whole->N + (size_t)i], &row->data[nT - 1], sizeof(FLOAT_DMEM))) {
        if (i == 0 || all_of_ != whole || all_nT_ != nT || all_n_ != whole->N) {
          const long N = whole->N;
          io_.ensure(nT * N, N * nFunctValues);
          check(smilehip_copy_to_device(context(), io_.own_in, whole->data, sizeof(float) * (uint64_t)(nT * N), nullptr));
          check(smilehip_funcspec_matrix(context(), &spec_, io_.own_in, N, nT, (int32_t)N, io_.d_out, nullptr));
          all_.resize((size_t)(N * nFunctValues));
          check(smilehip_copy_to_host(context(), all_.data(), io_.d_out, sizeof(float) * all_.size(), nullptr));
          check(smilehip_stream_synchronize(context(), nullptr));
          all_of_ = whole; all_nT_ = nT; all_n_ = N;
        }
        memcpy(y, &all_[(size_t)i * (size_t)nFunctValues], sizeof(float) * (size_t)nFunctValues);
        g_frames[14]++;
        return nFunctValues;
      }
    }
    io_.ensure(row->nT, nFunctValues);
    io_.up(row->data, row->nT);
    check(smilehip_funcspec_matrix(context(), &spec_, io_.d_in, 1, row->nT, 1, io_.d_out, nullptr));
    io_.down(y, nFunctValues);
    g_frames[14]++;
    return nFunctValues;
  }
 public:
  explicit cHipFunctionals(const char *n) : cFunctionals(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipFunctionals(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};
