// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): SURVEY 8(f) rank 2, per frame: cSpecScale, cPitchShs
// SURVEY 8(f) rank 2, per component. An F0-chain plan carries cSpecScale's spline / weighting tables and cPitchShs'
// shifts for one spectrum geometry (bins, frameSizeSec of the magnitude level).
static smilehip_plan *f0_component_plan(long K, double frame_size_sec, double min_pitch, double max_pitch, double cutoff,
                                        int n_harm, double compression, double min_f = 25.0, int n_cand = 6, int old_peaks = 0,
                                        int specscale_off = 0) {
  smilehip_lld_config c;
  smilehip_config_compare16_f0(&c);
  c.force_fft_frame_size_sec = frame_size_sec;
  c.force_frame_size = 2 * (K - 1);                      // the spectrum the component sees: K bins of a 2 (K - 1)-point transform
  c.pitch_min = min_pitch;
  c.pitch_max = max_pitch;
  c.voicing_cutoff = cutoff;
  c.shs_n_harmonics = n_harm;
  c.shs_compression = (float)compression;
  c.specscale_min_f = min_f;
  c.shs_n_candidates = n_cand;
  c.shs_old_peak_algo = old_peaks;
  c.specscale_off = specscale_off;
  smilehip_plan *pl = nullptr;
  check(smilehip_plan_create(context(), &c, &pl));
  smilehip_geometry g;
  check(smilehip_plan_geometry(pl, &g));
  if (g.n_bins != K) {                                   // (K - 1 not a power of two)
    smilehip_plan_destroy(pl);
    return nullptr;
  }
  return pl;
}

// cSpecScale::processVector (src/dsp/specScale.cpp:305-357) for the option set the F0 chains use (octave target scale,
// spline interpolation, minF 25, maxF -1, nPointsTarget 0, smoothing + enhancement + auditory weighting); anything else
// stays on the reference's CPU code. Names, frequency-axis info and the level meta data cPitchShs reads are inherited.
class cHipSpecScale : public BlockVP<cSpecScale> {
  FrameIO io_;
  bool cpu_warned_ = false;
  smilehip_plan *pl_ = nullptr;
  int usable_ = -1;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (usable_ < 0) {
      const char *sc = getStr("scale"), *ss = getStr("sourceScale"), *im = getStr("interpMethod");
      // the octave axis: scale = octave, or scale = log with logScaleBase 2 (specScale.cpp:100-111)
      const bool octave = sc && (!strncasecmp(sc, "oct", 3) || (!strncasecmp(sc, "log", 3) && getDouble("logScaleBase") == 2.0));
      usable_ = octave && ss && !strncasecmp(ss, "lin", 3) && im && !strncasecmp(im, "spl", 3) &&
                getDouble("minF") > 0.0 && getDouble("maxF") == -1.0 && getInt("nPointsTarget") <= 0 && Nsrc == Ndst;
      if (usable_) {
        const int off = (getInt("specEnhance") ? 0 : 1) | (getInt("specSmooth") ? 0 : 2) | (getInt("auditoryWeighting") ? 0 : 4);
        pl_ = f0_component_plan(Nsrc, (double)(float)reader_->getLevelConfig()->frameSizeSec, 52.0, 620.0, 0.7, 15, 0.85, getDouble("minF"), 6, 0, off);
        if (!pl_) usable_ = 0;
      }
    }
    if (!usable_) { HIP_FALLTHROUGH(15, "cSpecScale: only the octave (log2) target scale from a linear source by spline interpolation, minF > 0, maxF -1, on spectra of 512 .. 4096 points is built"); return cSpecScale::processVector(src, dst, Nsrc, Ndst, idxi); }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_specscale_frames(pl_, io_.d_in, Nsrc, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[15] += g_blk.n;
    return (int)Ndst;
  }
 public:
  explicit cHipSpecScale(const char *n) : BlockVP<cSpecScale>(n) {}
  ~cHipSpecScale() override { if (pl_) smilehip_plan_destroy(pl_); }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipSpecScale(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cPitchBase::processVector around cPitchShs::pitchDetect (src/lldcore/pitchBase.cpp:187-310, src/lld/pitchShs.cpp:214-347)
// for six candidates with scores + voicing, F0raw + voicingClip, greedyPeakAlgo, no octave correction / lfCut / SHS dump.
class cHipPitchShs : public BlockVP<cPitchShs> {
  FrameIO io_;
  bool cpu_warned_ = false;
  smilehip_plan *pl_ = nullptr;
  int usable_ = -1;
  bool raw_ = true, clip_ = true;
  int nc_ = 6;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (usable_ < 0) {
      nc_ = (int)getInt("nCandidates");
      usable_ = nc_ >= 1 && nc_ <= 6 && getInt("scores") == 1 && getInt("voicing") == 1 && getInt("F0C1") == 0 &&
                getInt("voicingC1") == 0 && getInt("octaveCorrection") == 0 &&
                getInt("shsSpectrumOutput") == 0 && getDouble("lfCut") <= 0.0 &&
                Ndst == 1 + 3 * nc_ + (getInt("F0raw") ? 1 : 0) + (getInt("voicingClip") ? 1 : 0) && reader_->getLevelNf() == 1;
      raw_ = getInt("F0raw") != 0;
      clip_ = getInt("voicingClip") != 0;
      cVectorMeta *md = reader_->getLevelMetaDataPtr();     // cSpecScale's minF (pitchShs.cpp:166-176): the octave axis' first point
      const double min_f = md ? (double)md->fData[0] : 25.0;
      if (!(min_f > 0.0)) usable_ = 0;
      if (usable_) {
        pl_ = f0_component_plan(Nsrc, (double)(float)reader_->getLevelConfig()->frameSizeSec, getDouble("minPitch"),
                                getDouble("maxPitch"), (double)(float)getDouble("voicingCutoff"), getInt("nHarmonics"),
                                (double)(float)getDouble("compressionFactor"), min_f, nc_, getInt("greedyPeakAlgo") ? 0 : 1);
        if (!pl_) usable_ = 0;
      }
    }
    if (!usable_) { HIP_FALLTHROUGH(16, "cPitchShs: only up to six candidates with scores and voicing (F0raw / voicingClip optional), no octaveCorrection / lfCut are built"); return cPitchShs::processVector(src, dst, Nsrc, Ndst, idxi); }
    io_.ensure(Nsrc, 21);
    io_.up(src, Nsrc);
    check(smilehip_pitchshs_frames(pl_, io_.d_in, Nsrc, io_.d_out, 21, g_blk.n, nullptr));
    if (raw_ && clip_ && nc_ == 6) io_.down(dst, 21);
    else {                                                  // [nCandidates | F0Cand | candVoicing | candScores] (+ F0raw) (+ voicingClip)
      const float *v = io_.down_rows();                     // (the device rows keep six slots per field)
      for (long fr = 0; fr < g_blk.n; ++fr, v += 21, dst += g_blk.ld_dst) {
        long n = 0;
        dst[n++] = v[0];
        for (int f = 0; f < 3; ++f)
          for (int c = 0; c < nc_; ++c) dst[n++] = v[1 + 6 * f + c];
        if (raw_) dst[n++] = v[19];
        if (clip_) dst[n++] = v[20];
      }
    }
    g_frames[16] += g_blk.n;
    return (int)Ndst;
  }
 public:
  explicit cHipPitchShs(const char *n) : BlockVP<cPitchShs>(n) {}
  ~cHipPitchShs() override { if (pl_) smilehip_plan_destroy(pl_); }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipPitchShs(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// the formant / voice-quality components of GeMAPSv01b_core.lld.conf.inc, per component, on ONE shared
// eGeMAPS plan (its tables fix the geometry: 16 kHz, 512-point spectrum of 20 ms frames -> 220 samples at 11 kHz, p = 11;
// 1024-point spectrum of 60 ms frames).
// Round 3: one plan per sample rate (8 .. 48 kHz). The rate is what the components that see it report (cSpecResample: the level's
// basePeriod; cSpectral with the GeMAPS options: bins and frameSizeSec of its spectrum); cLpc / cFormantLpc / cHarmonics, which run after
// them in every tick, use the plan of the rate seen last.
std::map<std::pair<long, long>, smilehip_plan *> g_gm_plans;      // (sample rate, cFormantLpc maxF in Hz)
long g_gm_rate = 16000, g_gm_maxf = 5450;
smilehip_plan *gemaps_plan(long rate = 0, long maxf = 0) {
  if (rate > 0) g_gm_rate = rate;
  if (maxf > 0) g_gm_maxf = maxf;
  smilehip_plan *&pl = g_gm_plans[std::make_pair(g_gm_rate, g_gm_maxf)];
  if (!pl) {
    smilehip_lld_config c;
    smilehip_config_egemapsv02(&c);
    c.sample_rate = (double)g_gm_rate;
    c.formant_max_freq = (double)g_gm_maxf;              // 5450 in GeMAPSv01b / eGeMAPSv02, 5500 in the v01a files (formantLpc.cpp:224-231)
    check(smilehip_plan_create(context(), &c, &pl));
  }
  return pl;
}
