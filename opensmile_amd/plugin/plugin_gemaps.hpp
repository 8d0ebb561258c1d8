// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): SURVEY 8(f) rank 3: cSpecResample, cLpc, cFormantLpc, cHarmonics, cPitchJitter
// cSpecResample::processVector (src/dsp/specResample.cpp:175-185) for [gemapsv01b_resampLpc]
class cHipSpecResample : public BlockVP<cSpecResample> {
  FrameIO io_;
  DevBytes cos_, sin_;
  bool cpu_warned_ = false;
  int usable_ = -1;                                       // 1: eGeMAPS' fused geometry (plan tables), 2: any geometry (the instance's own tables)
  long rate_ = 0;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (usable_ < 0) {
      const sDmLevelConfig *c = reader_->getLevelConfig();
      rate_ = c->basePeriod > 0.0 ? std::lround(1.0 / c->basePeriod) : 0;
      usable_ = !isSet("resampleRatio") && getDouble("targetFs") == 11000.0 && !getStr("inputFieldPartial") &&
                (Nsrc == 256 || Nsrc == 512 || Nsrc == 1024) && Ndst == 220 && rate_ >= 8000 && rate_ <= 48000 &&
                std::fabs(c->lastFrameSizeSec - 0.020) < 1e-4;
      if (!usable_ && dftWork && dftWork->K == Nsrc && dftWork->I == Ndst && Nsrc >= 2 && Nsrc <= 8192 && dftWork->kMax >= 2 &&
          dftWork->kMax <= Nsrc && !(dftWork->kMax & 1)) {
        // any other geometry: smileDsp_irdft with the tables smileDsp_initIrdft built for THIS instance (smileUtil.c:1752-1820)
        const uint64_t bytes = sizeof(float) * (uint64_t)(dftWork->kMax / 2) * (uint64_t)dftWork->I;
        if (smilehip_copy_to_device(context(), cos_.ensure(bytes), dftWork->costable, bytes, nullptr) ||
            smilehip_copy_to_device(context(), sin_.ensure(bytes), dftWork->sintable, bytes, nullptr))
          COMP_ERR("libsmilehip: %s", smilehip_last_error());
        usable_ = 2;
      }
    }
    if (!usable_) {
      HIP_FALLTHROUGH(17, "cSpecResample: spectra of more than 8192 values are not built");
      return cSpecResample::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    if (usable_ == 1) check(smilehip_specresample_frames(gemaps_plan(rate_), io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
    else check(smilehip_specresample_table_frames(context(), io_.d_in, Nsrc, Nsrc, Ndst, dftWork->kMax, (const float *)cos_.d,
                                                  (const float *)sin_.d, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[17] += g_blk.n;
    return (int)Ndst;
  }
 public:
  explicit cHipSpecResample(const char *n) : BlockVP<cSpecResample>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipSpecResample(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cLpc::processVector (src/lld/lpc.cpp:171-213) with method = acf, saveLPCoeff only: p = 11 on 220 samples through the eGeMAPS plan,
// any other frame length and order p <= 32 through smilehip_lpc_acf_frames
class cHipLpc : public BlockVP<cLpc> {
  FrameIO io_;
  bool cpu_warned_ = false;
  int usable_ = -1;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (usable_ < 0) {
      const char *met = getStr("method");
      const bool plain = met && !strncasecmp(met, "acf", 3) && getInt("saveLPCoeff") == 1 && !getInt("saveRefCoeff") && !getInt("lpGain") &&
                         !getInt("residual") && !getInt("lpSpectrum") && Ndst == p;
      usable_ = (plain && p == 11 && Nsrc == 220) ? 1 : ((plain && p >= 1 && p <= 32 && Nsrc > p && Nsrc <= 15000) ? 2 : 0);
    }
    if (!usable_) {
      HIP_FALLTHROUGH(18, "cLpc: only method = acf with saveLPCoeff alone (p <= 32) is built");
      return cLpc::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    if (usable_ == 1) check(smilehip_lpc_frames(gemaps_plan(), io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
    else check(smilehip_lpc_acf_frames(context(), io_.d_in, Nsrc, Nsrc, (int32_t)p, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[18] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipLpc(const char *n) : BlockVP<cLpc>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipLpc(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cFormantLpc::processVector (src/lld/formantLpc.cpp:192-290): 5 formants + bandwidths from 11 LP coefficients at 11 kHz
class cHipFormantLpc : public BlockVP<cFormantLpc> {
  FrameIO io_;
  bool cpu_warned_ = false;
  int usable_ = -1;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (usable_ < 0) {
      const sDmLevelConfig *c = reader_->getLevelConfig();
      usable_ = getInt("nFormants") == 5 && getInt("saveFormants") == 1 && getInt("saveBandwidths") == 1 && !getInt("saveIntensity") &&
                !getInt("saveNumberOfValidFormants") && !getInt("useLpSpec") && !getInt("medianFilter") && !getInt("octaveCorrection") &&
                getDouble("minF") == 50.0 && getDouble("maxF") > 50.0 && getDouble("maxF") == std::floor(getDouble("maxF")) && Nsrc == 11 && Ndst == 10 &&
                std::fabs(c->basePeriod - 1.0 / 11000.0) < 1e-12;
    }
    if (!usable_) {
      HIP_FALLTHROUGH(19, "cFormantLpc: only nFormants = 5 with bandwidths, minF 50, no median filter / octave correction on "
                          "11 coefficients at 11 kHz is built");
      return cFormantLpc::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_formantlpc_frames(gemaps_plan(0, (long)getDouble("maxF")), io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[19] += g_blk.n;
    return (int)Ndst;
  }
 public:
  explicit cHipFormantLpc(const char *n) : BlockVP<cFormantLpc>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipFormantLpc(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cHarmonics::processVector (src/lld/harmonics.cpp:743-1031) with [gemapsv01b_harmonics]'s options: the input vector holds the F0
// element, the formant frequency / bandwidth fields and the 513-bin magnitude field; the positions are looked up by name as the
// reference does in setupNewNames (:226-307).
class cHipHarmonics : public BlockVP<cHarmonics> {
  FrameIO io_;
  bool cpu_warned_ = false;
  int usable_ = -1;                                        // 1: GeMAPS' six outputs; 2: the ACF harmonics-to-noise ratio alone
  long iF0_ = -1, iSpec_ = -1, iFf_ = -1, iFb_ = -1, nSpec_ = 0, nFf_ = 0, nFb_ = 0, rate_ = 0;
  DevBytes fm_, f0_;
  std::vector<float> fmh_, f0h_;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (usable_ < 0) {
      static const char *const diffs[2] = {"H1-H2", "H1-A3"};
      bool ok = getInt("nHarmonics") == 100 && getInt("nHarmonicMagnitudes") == 0 && getInt("harmonicDifferencesLog") == 1 &&
                !getInt("harmonicDifferencesRatioLinear") && getInt("formantAmplitudes") == 1 && getInt("formantAmplitudesLogRel") == 1 &&
                !getInt("formantAmplitudesLinear") && getInt("formantAmplitudesStart") == 1 && getInt("formantAmplitudesEnd") == 3 &&
                getInt("computeAcfHnrLogdB") == 1 && !getInt("computeAcfHnrLinear") && getDouble("logRelValueFloorUnvoiced") == -201.0 &&
                getArraySize("harmonicDifferences") == 2 && Ndst == 6 && idxi == 0;
      for (int i = 0; ok && i < 2; ++i) {
        const char *v = getStr_f(myvprint("harmonicDifferences[%i]", i));
        ok = v && !strcmp(v, diffs[i]);
      }
      // computeAcfHnrLogdB alone (prosodyShsViterbiLoudness.conf): the ratio comes from the ACF of the squared magnitudes and the F0
      // lag (harmonics.cpp:590-712) -- no harmonic search, no formants; column 0 of the operator's row
      const bool hnr_only = !ok && getInt("nHarmonicMagnitudes") == 0 && getArraySize("harmonicDifferences") <= 0 && !getInt("formantAmplitudes") &&
                            getInt("computeAcfHnrLogdB") == 1 && !getInt("computeAcfHnrLinear") && Ndst == 1 && idxi == 0;
      if (ok || hnr_only) {
        iF0_ = findElement(getStr("f0ElementName"), getInt("f0ElementNameIsFull"), NULL, NULL, NULL);
        int specField = -1;
        iSpec_ = findField(getStr("magSpecFieldName"), getInt("magSpecFieldNameIsFull"), &nSpec_, NULL, -1, NULL, &specField);
        const char *ff = getStr("formantFrequencyFieldName"), *fb = getStr("formantBandwidthFieldName");
        if (ff && fb) {
          iFf_ = findField(ff, getInt("formantFrequencyFieldNameIsFull"), &nFf_, NULL, -1, NULL);
          iFb_ = findField(fb, getInt("formantBandwidthFieldNameIsFull"), &nFb_, NULL, -1, NULL);
        }
        // the frequency axis the reference reads from the magnitude field's meta data (harmonics.cpp:753-777): linear, bin 0 at 0 Hz
        // (513 bins of 15.625 Hz at 16 kHz); the operator's axis is i / fsSec of the plan's 60 ms spectrum
        const FrameMetaInfo *fmeta = reader_->getFrameMetaInfo();
        bool axis = false;
        const bool size_ok = nSpec_ == 257 || nSpec_ == 513 || nSpec_ == 1025 || nSpec_ == 2049;
        if (size_ok && fmeta && specField >= 0 && specField < fmeta->N && fmeta->field[specField].info &&
            fmeta->field[specField].infoSize == nSpec_ * (long)sizeof(double)) {
          const double *frq = (const double *)fmeta->field[specField].info;
          // GeMAPS' set: the plan of the rate seen last (cSpecResample / cSpectral run before this component in every tick);
          // alone: the rate the axis itself tells. Its 60 ms spectrum must be this one -- same number of bins, same axis step
          rate_ = hnr_only ? std::lround(frq[1] * (double)(2 * (nSpec_ - 1))) : g_gm_rate;
          if (rate_ >= 8000 && rate_ <= 48000) {
            smilehip_geometry g;
            check(smilehip_plan_geometry(gemaps_plan(rate_), &g));
            const double step = (double)rate_ / (double)(2 * (nSpec_ - 1));
            axis = frq[0] == 0.0 && frq[1] == step && frq[nSpec_ - 1] == step * (double)(nSpec_ - 1);
          }
        }
        const bool common = axis && iF0_ >= 0 && iSpec_ >= 0 && size_ok && iSpec_ + nSpec_ <= Nsrc;
        if (hnr_only) { usable_ = common ? 2 : 0; ok = false; }
        else ok = common && iFf_ >= 0 && iFb_ >= 0 && nFf_ == 5 && nFb_ == 5;
      }
      if (usable_ < 0) usable_ = ok ? 1 : 0;
    }
    if (!usable_) {
      HIP_FALLTHROUGH(20, "cHarmonics: only GeMAPS' option set (H1-H2, H1-A3, formant amplitudes 1..3, ACF HNR in dB; 5 formants, the "
                          "spectrum of 60 ms frames at 8 .. 48 kHz) and the ACF HNR in dB alone are built");
      return cHarmonics::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    io_.ensure(nSpec_, 6);
    io_.up(src + iSpec_, nSpec_);
    // the F0 element and the formant fields of the call's frames (1, or the frames of a block tick -- plugin_block.hpp)
    const long nf = g_blk.n;
    fmh_.assign((size_t)nf * 10, 0.0f);
    f0h_.resize((size_t)nf);
    for (long f = 0; f < nf; ++f) {
      const FLOAT_DMEM *row = src + (size_t)f * (size_t)g_blk.ld_src;
      if (usable_ == 1) {
        memcpy(&fmh_[(size_t)f * 10], row + iFf_, sizeof(float) * 5);
        memcpy(&fmh_[(size_t)f * 10 + 5], row + iFb_, sizeof(float) * 5);
      }
      f0h_[(size_t)f] = row[iF0_];
    }
    float *d_fm = (float *)fm_.ensure(sizeof(float) * 10 * (uint64_t)nf), *d_f0 = (float *)f0_.ensure(sizeof(float) * (uint64_t)nf);
    if (smilehip_copy_to_device(context(), d_fm, fmh_.data(), sizeof(float) * fmh_.size(), nullptr) ||
        smilehip_copy_to_device(context(), d_f0, f0h_.data(), sizeof(float) * f0h_.size(), nullptr))
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
    check(smilehip_harmonics_frames(usable_ == 2 ? gemaps_plan(rate_) : gemaps_plan(), d_f0, d_fm, 10, io_.d_in, nSpec_, io_.d_out, 6, nf, nullptr));
    io_.down(dst, usable_ == 2 ? 1 : 6);
    g_frames[20] += nf;
    return 1;
  }
 public:
  explicit cHipHarmonics(const char *n) : BlockVP<cHarmonics>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipHarmonics(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cPitchJitter::myTick (src/lld/pitchJitter.cpp:591-1084) -- tick-level: per F0 frame the component reads a stretch of the
// wave level whose position and length depend on what the previous frames left over, matches pitch periods in it and
// carries period / jitter / shimmer values on. The prologue (:604-668: which samples to read) runs here on the base class's
// own members, the samples go to the device as 16-bit PCM, the matching itself and the carried values live in the
// device-resident stream (smilehip_jitter_stream_push: the fused path's kernel, one frame per launch).
class cHipPitchJitter : public cPitchJitter {
  smilehip_jitter_stream *js_ = nullptr;
  bool ready_ = false, usable_ = false, cpu_warned_ = false;
  std::vector<int16_t> pcm_;
  // block ticks (plugin_block.hpp): the samples of the wave level go to the device-resident stream as they arrive (pend_: what has been
  // read from the level and not pushed yet, samples [dev_upto_, dev_upto_ + pend_.size())), every F0 frame the pitch level holds in one
  // smilehip_jitter_stream_push_frames call; the read positions are the stream's own (the device function the batch path runs)
  BlockGate gate_;
  BlockMat bout_;
  std::vector<int16_t> pend_;
  std::vector<float> f0s_, outs_;
  long dev_upto_ = 0;
  bool block_started_ = false;
  void make_stream(const TimeMetaInfo *tm) {
    const double T = reader_->getLevelT();
    const double pitchT = tm->period;
    const long H = (long)round(pitchT / T), N = (long)round(tm->lengthSec / T);   // (lenF itself is N or N + 1: rounding of the time stamps)
    check(smilehip_jitter_stream_create(context(), T, N, H, pitchT, searchRangeRel, useBrokenJitterThresh_, &js_));
    // the first F0 frame's time stamp: frame 0 behind the Viterbi smoother, frame 1 behind cPitchSmoother (one frame of delay, the
    // time meta data of the frame it was called with)
    check(smilehip_jitter_stream_set_time_offset(js_, std::lround(tm->time / pitchT)));
  }
  bool blockTick(eTickResult &res) {
    if (!block_mode() || reader_->getNLevels() != 1) return false;
    // the wave level's new samples (the level holds s / 32767: exact as 16-bit values)
    const long curW = reader_->dm->getCurW(reader_->level[0]);
    const long have = dev_upto_ + (long)pend_.size();
    if (curW > have) {
      cMatrix *mat = reader_->getMatrix(have, curW - have);
      if (mat == NULL || mat->nT != curW - have || mat->N != 1) {
        if (block_started_) COMP_ERR("libsmilehip plugin: cPitchJitter: samples %ld .. %ld of the wave level are not readable any more", have, curW);
        return false;
      }
      const size_t o = pend_.size();
      pend_.resize(o + (size_t)mat->nT);
      for (long i = 0; i < mat->nT; ++i) pend_[o + (size_t)i] = (int16_t)lrintf(mat->data[i] * 32767.0f);
      reader_->catchupCurR(curW);
      block_started_ = true;
    }
    cDataReader *fr = F0reader;
    const long avail = fr->getNAvail();
    if (avail < 1 || fr->curR < 0) { res = TICK_SOURCE_NOT_AVAIL; return true; }
    if (gate_.input(avail)) { res = TICK_SUCCESS; return true; }
    const long n = gate_.frames(writer_, avail);
    if (n == 0) { res = TICK_SUCCESS; return true; }
    if (n < 1) { res = TICK_DEST_NO_SPACE; return true; }
    cMatrix *fm = fr->getMatrix(fr->curR, n);
    if (fm == NULL || fm->nT != n) { res = TICK_SOURCE_NOT_AVAIL; return true; }
    fr->curR += n;
    fr->catchupCurR(fr->curR);
    if (!js_) make_stream(fm->tmeta);
    f0s_.resize((size_t)n);
    for (long f = 0; f < n; ++f) f0s_[(size_t)f] = F0fieldIdx < fm->N ? fm->data[(size_t)f * (size_t)fm->N + (size_t)F0fieldIdx] : 0.0f;
    outs_.resize((size_t)n * 5);
    int64_t li = 0, lm = 0;
    check(smilehip_jitter_stream_push_frames(js_, f0s_.data(), (int32_t)n, pend_.data(), dev_upto_, (int64_t)pend_.size(), outs_.data(), &li, &lm));
    dev_upto_ += (long)pend_.size();
    pend_.clear();
    lastIdx = (long)li; lastMis = (long)lm;
    g_frames[23] += n;
    g_block_ticks++;
    g_block_frames += n;
    res = TICK_SUCCESS;
    if (Nout == 0) { res = TICK_INACTIVE; return true; }   // (:941-947)
    cMatrix *o = bout_.get(Nout, n);
    long m = 0;
    for (long f = 0; f < n; ++f) {
      if (onlyVoiced && (f0s_[(size_t)f] == 0.0)) continue;
      const float *out5 = &outs_[(size_t)f * 5];
      FLOAT_DMEM *d = o->data + (size_t)m * (size_t)Nout;
      long k = 0;
      if (jitterLocal) d[k++] = out5[0];
      if (jitterDDP) d[k++] = out5[1];
      if (shimmerLocal) d[k++] = out5[2];
      if (shimmerLocalDB) d[k++] = out5[4];
      if (logHNR) d[k++] = out5[3];
      o->tmeta[m] = fm->tmeta[f];
      ++m;
    }
    if (m > 0) { o->nT = m; writer_->setNextMatrix(o); }
    return true;
  }
 protected:
  int configureWriter(sDmLevelConfig &c) override {
    const int r = cPitchJitter::configureWriter(c);
    if (r && block_mode() && c.blocksizeWriter < block_frames()) c.blocksizeWriter = block_frames();   // (plugin_shared.hpp: block_frames)
    return r;
  }
  eTickResult myTick(long long t) override {
    g_fused.init();
    if (g_fused.big) {                                     // big-set fused mode: the jitter columns come from the batch; the wave level's samples are only passed by
      reader_->catchupCurR();
      return TICK_INACTIVE;
    }
    if (!ready_) {
      ready_ = true;
      usable_ = !jitterLocalEnv && !jitterDDPEnv && !shimmerLocalEnv && !shimmerLocalDBEnv && !shimmerUseRmsAmplitude && !harmonicERMS &&
                !noiseERMS && !linearHNR && !sourceQualityRange && !sourceQualityMean && !refinedF0 &&   // (periodLengths / periodStarts: members the reference never sets or reads)
                !usePeakToPeakPeriodLength_ && minNumPeriods == 2 && filehandle == NULL &&
                (useBrokenJitterThresh_ || threshCC_ == (FLOAT_DMEM)0.5) && lgHNRfloor == (FLOAT_DMEM)-100.0 && reader_->getLevelN() == 1;
    }
    if (!usable_) {
      HIP_FALLTHROUGH(23, "cPitchJitter: only jitterLocal / jitterDDP / shimmerLocal / shimmerLocalDB / logHNR with minNumPeriods = 2, minCC = 0.5 "
                          "(or useBrokenJitterThresh), lgHNRfloor = -100 on a mono wave level are built");
      return cPitchJitter::myTick(t);
    }
    if (isEOI()) return TICK_INACTIVE;
    { eTickResult r; if (blockTick(r)) return r; }
    if (!writer_->checkWrite(1)) return TICK_DEST_NO_SPACE;
    cVector *fvec = F0reader->getNextFrame();
    if (fvec == NULL) return TICK_SOURCE_NOT_AVAIL;
    FLOAT_DMEM F0 = 0.0;
    if (F0fieldIdx < fvec->N) F0 = fvec->data[F0fieldIdx];
    const long lenF = (long)ceil(fvec->tmeta->lengthSec / fvec->tmeta->framePeriod);
    const double T = reader_->getLevelT();
    const long startVidx = (long)round(fvec->tmeta->time / T);
    const double pitchT = fvec->tmeta->period;
    const long ppLen = (long)ceil(pitchT / T);
    if (!js_) {
      const long H = (long)round(pitchT / T), N = (long)round(fvec->tmeta->lengthSec / T);   // (lenF itself is N or N + 1: rounding of the time stamps)
      check(smilehip_jitter_stream_create(context(), T, N, H, pitchT, searchRangeRel, useBrokenJitterThresh_, &js_));
      // the first F0 frame's time stamp: frame 0 behind the Viterbi smoother, frame 1 behind cPitchSmoother (one frame of delay, the
      // time meta data of the frame it was called with)
      check(smilehip_jitter_stream_set_time_offset(js_, std::lround(fvec->tmeta->time / pitchT)));
    }
    const long toRead0 = ppLen + lastMis;
    long toRead = toRead0;
    if (F0 > 0.0) {
      const double Tf = (1.0 / F0) / T;
      const long T0maxF = (long)ceil((1.0 + searchRangeRel) * Tf);
      const long two_pp = minNumPeriods * T0maxF + minNumPeriods;
      if (toRead < two_pp) toRead = two_pp;
    }
    long maxRead = lastMis + lenF;
    if (toRead > maxRead) toRead = maxRead;
    if (startVidx - lastMis != lastIdx) {
      lastIdx = startVidx;
      if (toRead > lenF) toRead = lenF;
      if (maxRead > lenF) maxRead = lenF;
    }
    cMatrix *mat = reader_->getMatrix(lastIdx, toRead);
    float out5[5] = {0, 0, 0, 0, 0};
    int64_t li = 0, lm = 0;
    if (mat == NULL) {                                     // (:660-665) the position still moves on
      check(smilehip_jitter_stream_push(js_, F0, nullptr, 0, 0, out5, &li, &lm));
      lastIdx = (long)li; lastMis = (long)lm;
      return TICK_SOURCE_NOT_AVAIL;
    }
    if (maxRead < 1 || mat->data == NULL) return TICK_INACTIVE;
    pcm_.resize((size_t)mat->nT);
    for (long i = 0; i < mat->nT; ++i) pcm_[(size_t)i] = (int16_t)lrintf(mat->data[i] * 32767.0f);   // the level holds s / 32767: exact
    check(smilehip_jitter_stream_push(js_, F0, pcm_.data(), lastIdx, mat->nT, out5, &li, &lm));
    lastIdx = (long)li; lastMis = (long)lm;
    g_frames[23]++;
    if (Nout == 0) return TICK_INACTIVE;                    // (:941-947)
    if (onlyVoiced && (F0 == 0.0)) return TICK_INACTIVE;
    if (out == NULL) out = new cVector(Nout);
    long n = 0;
    if (jitterLocal) out->data[n++] = out5[0];
    if (jitterDDP) out->data[n++] = out5[1];
    if (shimmerLocal) out->data[n++] = out5[2];
    if (shimmerLocalDB) out->data[n++] = out5[4];
    if (logHNR) out->data[n++] = out5[3];
    out->setTimeMeta(fvec->tmeta);
    writer_->setNextFrame(out);
    return TICK_SUCCESS;
  }
 public:
  explicit cHipPitchJitter(const char *n) : cPitchJitter(n) {}
  ~cHipPitchJitter() override { if (js_) smilehip_jitter_stream_destroy(js_); }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipPitchJitter(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};
