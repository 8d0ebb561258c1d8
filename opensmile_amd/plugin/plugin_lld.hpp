// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): R9, R10, R12: cEnergy, cMZcr, cAcf, cPitchACF
// R12  cEnergy::processVector  (src/lldcore/energy.cpp:152-185): the double-accumulated sum of
// squares comes from the device, the rms / squared / log expressions are the reference's
class cHipEnergy : public BlockVP<cEnergy> {
  int fused_ = -1;
  const FusedLevel *fcols_ = nullptr;
  long fnext_ = 0;
  FrameIO io_;
  bool cpu_warned_ = false;
  DevBytes res_;
  std::vector<double> sums_;
  int htk_ = 0, erms_ = 0, e2_ = 0, elog_ = 0;
  FLOAT_DMEM sRms_ = 1, sLog_ = 1, sSq_ = 1, bLog_ = 0, bRms_ = 0, bSq_ = 0;
  bool ready_ = false;
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
    return BlockVP<cEnergy>::myTick(t);
  }
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (Nsrc == 0) return 0;
    if (!ready_) {                                       // cEnergy::myFetchConfig, energy.cpp:58-81
      htk_ = getInt("htkcompatible");
      erms_ = getInt("rms"); e2_ = getInt("energy2"); elog_ = getInt("log");
      if (htk_) { elog_ = 1; erms_ = 0; }
      bLog_ = (FLOAT_DMEM)getDouble("ebiasLog"); bRms_ = (FLOAT_DMEM)getDouble("ebiasRms"); bSq_ = (FLOAT_DMEM)getDouble("ebiasSquare");
      sRms_ = (FLOAT_DMEM)getDouble("escaleRms"); sSq_ = (FLOAT_DMEM)getDouble("escaleSquare"); sLog_ = (FLOAT_DMEM)getDouble("escaleLog");
      ready_ = true;
    }
    io_.ensure(Nsrc, 1);
    io_.up(src, Nsrc);
    const long nf = g_blk.n;                              // 1, or the frames of a block tick (plugin_block.hpp)
    double *d_d = (double *)res_.ensure(sizeof(double) * (uint64_t)nf);
    check(smilehip_sumsq_frames(context(), io_.d_in, Nsrc, Nsrc, nf, d_d, nullptr));
    sums_.resize((size_t)nf);
    res_.down(sums_.data(), sizeof(double) * (uint64_t)nf);
    int n = 0;
    for (long f = 0; f < nf; ++f, dst += g_blk.ld_dst) {
      double d = sums_[(size_t)f];
      n = 0;
      if (erms_) dst[n++] = (FLOAT_DMEM)sqrt(d / (FLOAT_DMEM)Nsrc) * sRms_ + bRms_;
      if (e2_) dst[n++] = (FLOAT_DMEM)(d / (double)Nsrc) * sSq_ + bSq_;
      if (elog_) {
        const double minE = 8.674676e-019;
        if (!htk_) {
          d /= (FLOAT_DMEM)Nsrc;
          if (d < minE) d = minE;
          dst[n++] = (FLOAT_DMEM)log(d) * sLog_ + bLog_;
        } else {
          d *= 32767.0 * 32767.0;
          if (d <= 1.0) d = 1.0;
          dst[n++] = (FLOAT_DMEM)log(d) * sLog_ + bLog_;
        }
      }
    }
    g_frames[6] += nf;
    return n;
  }
 public:
  explicit cHipEnergy(const char *n) : BlockVP<cEnergy>(n) {}
  ~cHipEnergy() override { delete fblock_; }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipEnergy(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cWaveSource (src/iocore/waveSource.cpp:240-294): the reference's source, untouched -- it reads the file and writes its level block
// after block. The override only tells the fused chain (plugin_shared.hpp) when the last block has been written: the batch over the
// whole input starts then.
class cHipWaveSource : public cWaveSource {
 protected:
  eTickResult myTick(long long t) override {
    const eTickResult r = cWaveSource::myTick(t);
    if (eof) g_fused.source_eof = true;
    return r;
  }
 public:
  explicit cHipWaveSource(const char *n) : cWaveSource(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipWaveSource(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cVectorConcat (src/other/vectorConcat.cpp) as the last component of a fused cepstral chain: the level the sinks read gets the
// batch's finished rows (static | delta | acceleration, mean-normalised where the file says so) at the tick level. Anywhere
// else it is the reference's component, untouched.
class cHipVectorConcat : public BlockVP<cVectorConcat> {
  int fused_ = -1;
  const FusedLevel *fcols_ = nullptr;
  long fnext_ = 0;
  cMatrix *fblock_ = nullptr;
 protected:
  eTickResult myTick(long long t) override {
    if (fused_ < 0) {
      g_fused.init();
      fcols_ = (g_fused.active && g_fused.final_level) ? g_fused.static_level(getStr("writer.dmLevel")) : nullptr;
      fused_ = fcols_ ? 1 : 0;
    }
    if (fused_) {
      if (isEOI()) return TICK_INACTIVE;
      return g_fused.tick_write(*fcols_, writer_, fnext_, fblock_, blocksizeW_);
    }
    return BlockVP<cVectorConcat>::myTick(t);
  }
  // cVectorConcat::processVector (src/other/vectorConcat.cpp:48-53: the field's values copied to their place in the output vector) for
  // the frames of the call -- one on the reference's ticks, every frame the input levels hold on a block tick (plugin_block.hpp);
  // no arithmetic, the rows stay on the host
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    const size_t w = sizeof(FLOAT_DMEM) * (size_t)(Ndst < Nsrc ? Ndst : Nsrc);
    for (long f = 0; f < g_blk.n; ++f) {
      const FLOAT_DMEM *a = src + (size_t)f * (size_t)g_blk.ld_src;
      FLOAT_DMEM *b = dst + (size_t)f * (size_t)g_blk.ld_dst;
      if (a != b) memcpy(b, a, w);
    }
    return 1;
  }
 public:
  explicit cHipVectorConcat(const char *n) : BlockVP<cVectorConcat>(n) {}
  ~cHipVectorConcat() override { delete fblock_; }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipVectorConcat(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R12  cMZcr::processVector, zero-crossing rate  (src/lldcore/mzcr.cpp:109-150)
class cHipMZcr : public BlockVP<cMZcr> {
  FrameIO io_;
  bool cpu_warned_ = false;
  DevBytes res_;
  std::vector<int32_t> counts_;
  int plain_ = -1, flags_ = 0;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (plain_ < 0) {
      plain_ = (getInt("zcr") && !getInt("mcr") && !getInt("amax") && !getInt("maxmin") && !getInt("dc")) ? 1 : 0;
      flags_ = (getInt("zcr") ? SMILEHIP_MZCR_ZCR : 0) | (getInt("mcr") ? SMILEHIP_MZCR_MCR : 0) | (getInt("amax") ? SMILEHIP_MZCR_AMAX : 0) |
               (getInt("maxmin") ? SMILEHIP_MZCR_MAXMIN : 0) | (getInt("dc") ? SMILEHIP_MZCR_DC : 0);
    }
    if (Nsrc == 0) return 0;                             // mzcr.cpp:112
    if (!plain_ && flags_ && Nsrc <= 32768) {            // mcr / amax / maxmin / dc (mzcr.cpp:119-150)
      const int n_out = ((flags_ & 1) ? 1 : 0) + ((flags_ & 2) ? 1 : 0) + ((flags_ & 4) ? 1 : 0) + ((flags_ & 8) ? 2 : 0) + ((flags_ & 16) ? 1 : 0);
      io_.ensure(Nsrc, n_out);
      io_.up(src, Nsrc);
      check(smilehip_mzcr_frames(context(), io_.d_in, Nsrc, Nsrc, g_blk.n, flags_, io_.d_out, n_out, nullptr));
      io_.down(dst, n_out);
      g_frames[7] += g_blk.n;
      return n_out;
    }
    if (!plain_) { HIP_FALLTHROUGH(7, "cMZcr: no output selected, or a frame longer than 32768 samples"); return cMZcr::processVector(src, dst, Nsrc, Ndst, idxi); }
    io_.ensure(Nsrc, 1);
    io_.up(src, Nsrc);
    const long nf = g_blk.n;                              // 1, or the frames of a block tick (plugin_block.hpp)
    int32_t *d_c = (int32_t *)res_.ensure(sizeof(int32_t) * (uint64_t)nf);
    check(smilehip_zcr_count_frames(context(), io_.d_in, Nsrc, Nsrc, nf, d_c, nullptr));
    counts_.resize((size_t)nf);
    res_.down(counts_.data(), sizeof(int32_t) * (uint64_t)nf);
    for (long f = 0; f < nf; ++f, dst += g_blk.ld_dst) {
      FLOAT_DMEM nzc = (FLOAT_DMEM)counts_[(size_t)f];
      nzc /= (FLOAT_DMEM)Nsrc;
      dst[0] = nzc;
    }
    g_frames[7] += nf;
    return 1;
  }
 public:
  explicit cHipMZcr(const char *n) : BlockVP<cMZcr>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipMZcr(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R9  cAcf::processVector, forward path  (src/dspcore/acf.cpp:249-349)
class cHipAcf : public BlockVP<cAcf> {
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
  int plain_ = -1, use_power_ = 0, cepstrum_ = 0, norm_ = 0, abs_ceps_ = 0;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (plain_ < 0) {                                    // cAcf::myFetchConfig, acf.cpp:77-110
      cepstrum_ = getInt("cepstrum");
      use_power_ = cepstrum_ ? (isSet("usePower") ? getInt("usePower") : 0) : getInt("usePower");
      norm_ = getInt("acfCepsNormOutput");
      abs_ceps_ = getInt("absCepstrum");
      plain_ = (!getInt("inverse") && !getInt("cosLifterCepstrum")) ? 1 : 0;
      if (cepstrum_ && getInt("oldCompatCepstrum")) cepstrum_ = 2;      // log(x) of the inner bins, DC and Nyquist as they are (acf.cpp:275-286)
    }
    const long N = (Nsrc - 1) * 2;
    if (!plain_ || Nsrc < 5 || (N & (N - 1)) != 0 || Ndst > N / 2)
      { HIP_FALLTHROUGH(8, "cAcf: inverse / cosLifterCepstrum / expBeforeAbs or this field size are not built"); return cAcf::processVector(src, dst, Nsrc, Ndst, idxi); }
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      smilehip_lld_config c = base_config(N, SMILEHIP_STAGE_FFT);
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_acf_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, Ndst, g_blk.n, use_power_, cepstrum_, norm_, abs_ceps_, nullptr));
    io_.down(dst, Ndst);
    g_frames[8] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipAcf(const char *n) : BlockVP<cAcf>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipAcf(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R10  cPitchACF::processVector  (src/lldcore/pitchACF.cpp:137-247), all of it on the device: the voicing probability and the
// cepstral peak (smilehip_pitchacf_frames), then F0, the voicing cut-off, the causal F0 contour and its envelope
// (smilehip_pitchacf_contour_step -- the device function the batch chain runs, its state in device memory). The host side
// only maps the harmonics-to-noise ratio of two ACF values it already holds onto the three HNR scales (:310-361).
class cHipPitchACF : public BlockVP<cPitchACF> {
  FrameIO io_;
  bool cpu_warned_ = false;
  DevBytes res_;
  bool state_ready_ = false;
  int plain_ = -1, voiceProb_ = 0, F0_ = 0, F0raw_ = 0, F0env_ = 0, HNR_ = 0, HNRdB_ = 0, linHNR_ = 0, voiceQual_ = 0;
  double maxPitch_ = 0.0, voicingCutoff_ = 0.0;
  float fsSec_ = -1.0f;
  std::vector<unsigned char> host_;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (plain_ < 0) {                                    // cPitchACF::myFetchConfig, pitchACF.cpp:75-104
      voiceProb_ = getInt("voiceProb"); F0_ = getInt("F0"); F0raw_ = getInt("F0raw"); F0env_ = getInt("F0env");
      voicingCutoff_ = getDouble("voicingCutoff");
      if (voicingCutoff_ > 1.0) voicingCutoff_ = 1.0;
      if (voicingCutoff_ < 0.0) voicingCutoff_ = 0.0;
      maxPitch_ = getDouble("maxPitch");
      if (maxPitch_ < 0.0) maxPitch_ = 0.0;
      fsSec_ = (float)(reader_->getLevelConfig()->frameSizeSec);          // setupNewNames, :110-114
      HNR_ = getInt("HNR"); HNRdB_ = getInt("HNRdB"); linHNR_ = getInt("linHNR"); voiceQual_ = getInt("voiceQual");
      plain_ = 1;
    }
    const long N = (int)floor(Nsrc / 2.0);
    if (N < 4 || 2 * N != Nsrc) { HIP_FALLTHROUGH(9, "cPitchACF: the input is not [acf | cepstrum] of equal, even size"); return cPitchACF::processVector(src, dst, Nsrc, Ndst, idxi); }
    io_.ensure(Nsrc, 1);
    io_.up(src, Nsrc);
    // device results of the call's frames (1, or the frames of a block tick -- plugin_block.hpp): the contour state (8 words), then
    // per frame the voicing probability, the peak index, the ACF zero-crossing rate, the four contour outputs
    const long nf = g_blk.n;
    const uint64_t o_v = 32, o_i = o_v + 8 * (uint64_t)nf, o_z = (o_i + 4 * (uint64_t)nf + 7) & ~(uint64_t)7, o_f = o_z + 8 * (uint64_t)nf,
                   total = o_f + 16 * (uint64_t)nf;
    if (!state_ready_ || total > res_.cap) {               // a stream starts with an all-zero contour; a larger block keeps the state
      float st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (state_ready_) res_.down(st, sizeof(st));
      res_.ensure(total > 4096 ? total : 4096);
      check(smilehip_copy_to_device(context(), res_.d, st, sizeof(st), nullptr));
      check(smilehip_stream_synchronize(context(), nullptr));
      state_ready_ = true;
    }
    unsigned char *r = (unsigned char *)res_.d;
    const double Tsamp = fsSec_ / (double)Nsrc;
    check(smilehip_pitchacf_frames(context(), io_.d_in, Nsrc, N, nf, (double)fsSec_, maxPitch_, (double *)(r + o_v), (int32_t *)(r + o_i), nullptr));
    if (voiceQual_) check(smilehip_pitchacf_zcr_frames(context(), io_.d_in, Nsrc, N, nf, (double)fsSec_, maxPitch_, (double *)(r + o_z), nullptr));
    const bool contour = F0_ || F0env_ || F0raw_ || voiceQual_;
    if (contour)
      check(smilehip_pitchacf_contour_frames(context(), (const double *)(r + o_v), (const int32_t *)(r + o_i), Tsamp, voicingCutoff_, (float *)r,
                                             (float *)(r + o_f), nf, nullptr));
    host_.resize((size_t)(total - o_v));
    check(smilehip_copy_to_host(context(), host_.data(), r + o_v, total - o_v, nullptr));
    check(smilehip_stream_synchronize(context(), nullptr));
    const double *h_v = (const double *)host_.data(), *h_z = (const double *)(host_.data() + (o_z - o_v));
    const int32_t *h_i = (const int32_t *)(host_.data() + (o_i - o_v));
    const float *h_f = (const float *)(host_.data() + (o_f - o_v));
    int n = 0;
    for (long f = 0; f < nf; ++f, src += g_blk.ld_src, dst += g_blk.ld_dst) {
    struct { double voicing; int32_t idx; double acfZcr; float f0[4]; } h = {h_v[f], h_i[f], voiceQual_ ? h_z[f] : 0.0, {h_f[4 * f], h_f[4 * f + 1], h_f[4 * f + 2], h_f[4 * f + 3]}};
    const long peak = h.idx;
    n = 0;
    if (voiceProb_) dst[n++] = (FLOAT_DMEM)h.voicing;
    if (HNR_ || HNRdB_ || linHNR_) {
      // harmonics-to-noise ratio acf[peak] / (acf[0] - acf[peak]), `pure` where the denominator vanishes. The difference is a
      // float; the natural-log scale divides in float (:315), the dB and linear scales hold the difference in a double and
      // divide in double (:331-335, :351-355). Natural-log scale floored at 1e-11, dB scale limited to -100 .. 100, linear
      // scale limited to 1e-2 .. 1e4
      const FLOAT_DMEM noise = src[0] - src[peak];
      const auto ratio = [&](double pure) { return noise == 0.0 ? pure : (double)src[peak] / (double)noise; };
      if (HNR_) { const double q = noise == 0.0 ? 1e20 : (double)(src[peak] / noise); dst[n++] = (FLOAT_DMEM)(10.0 * log(q > 0.00000000001 ? q : 0.00000000001)); }
      if (HNRdB_) { const double q = ratio(10e10); dst[n++] = (FLOAT_DMEM)(q <= 10e-10 ? -100.0 : (q >= 10e10 ? 100.0 : 10.0 * log(q) / log(10.0))); }
      if (linHNR_) { const double q = ratio(10e3); dst[n++] = (FLOAT_DMEM)(q <= 10e-3 ? 10e-3 : (q >= 10e3 ? 10e3 : q)); }
    }
    if (contour) {
      if (voiceQual_) {                                   // :178-181
        FLOAT_DMEM vq = ((FLOAT_DMEM)maxPitch_ - (FLOAT_DMEM)fabs((h.acfZcr * maxPitch_) - ((FLOAT_DMEM)1.0 / ((FLOAT_DMEM)(peak) * (FLOAT_DMEM)Tsamp)))) * (FLOAT_DMEM)h.voicing;
        dst[n++] = peak == 0 ? (FLOAT_DMEM)0.0 : vq;
      }
      if (F0_) dst[n++] = h.f0[0];
      if (F0raw_) dst[n++] = h.f0[1];
      if (F0env_) dst[n++] = h.f0[2];
    }
    }
    g_frames[9] += nf;
    return n;
  }
 public:
  explicit cHipPitchACF(const char *n) : BlockVP<cPitchACF>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipPitchACF(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};
