// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): fused mode behind the component API: cHipLldSource
// ---------------------------------------------------------------------------------------------
// Fused mode behind the component API: ONE data source that owns a whole file and replaces the wave source plus
// every component of the chain. It runs the fused kernels once (smilehip_lld_run_host) and then feeds the finished
// feature rows into the level the chain's last component used to write, so that every sink / functional of a config
// keeps working (conf/MFCC12_0_D_A_hip.conf). A new component type with its own options:
//   filename    the RIFF/WAVE file (16-bit mono)
//   featureSet  mfcc12_{0,e}_d_a[_z] | plp_{0,e}_d_a[_z]   (the sets whose rows are frames: row time = row * frameStep)
#define COMPONENT_NAME_CHIPLLDSOURCE "cHipLldSource"
#define COMPONENT_DESCRIPTION_CHIPLLDSOURCE "Reads a wave file and writes the LLD rows of a whole feature set, computed by the fused HIP kernels of libsmilehip, to a dataMemory level."
class cHipLldSource : public cDataSource {
  // the sets: the eight HTK-style files (rows = frames), the LLD levels of the three big sets (their own row counts and
  // end-of-input time stamps, smilehip_row_time), and the functionals levels (ONE vector per input)
  enum Set { kHtkVariant, kIs09, kCompare16, kIs13, kEgemaps };
  std::string filename_, set_;
  std::vector<float> rows_;
  std::vector<double> times_;
  std::vector<std::string> names_;
  long n_rows_ = 0, next_ = 0;
  int n_cols_ = 0, n_lld_ = 0;
  double period_sec_ = 0.01, frame_size_sec_ = 0.025;
  bool ran_ = false;
  Set kind_ = kHtkVariant;
  bool func_ = false;                                    // featureSet <set>_func: the functionals level, one vector
  cMatrix *block_ = nullptr;

  void config_for(smilehip_lld_config &c) {
    switch (kind_) {
      case kIs09: smilehip_config_is09_lld(&c); return;
      case kCompare16: smilehip_config_compare16(&c); return;
      case kIs13: smilehip_config_is13_compare(&c); return;
      case kEgemaps: smilehip_config_egemapsv02(&c); return;
      default: break;
    }
    std::string up;                                      // any of the eight files of config/mfcc and config/plp, by name
    for (char ch : set_) up += (char)toupper((unsigned char)ch);
    if (smilehip_config_htk_variant(&c, up.c_str()) != SMILEHIP_OK)
      COMP_ERR("cHipLldSource: unknown featureSet '%s' (mfcc12_{0,e}_d_a[_z], plp_{0,e}_d_a[_z], is09_{lld,func}, compare16_{lld,func}, "
               "is13_compare_{lld,func}, egemapsv02_{lld,func})", set_.c_str());
  }
  void run_once() {
    smilehip_host::WaveInfo wi;
    std::vector<unsigned char> raw;
    std::string err;
    if (!smilehip_host::read_wave_file(filename_, wi, raw, err)) COMP_ERR("cHipLldSource: %s", err.c_str());
    if (wi.sample_type != 1 || wi.n_bps != 2 || wi.n_chan != 1) COMP_ERR("cHipLldSource: '%s' is not 16-bit mono PCM", filename_.c_str());
    smilehip_lld_config c;
    config_for(c);
    c.sample_rate = (double)wi.sample_rate;
    smilehip_plan *pl = nullptr;
    check(smilehip_plan_create(context(), &c, &pl));
    const int64_t n = (int64_t)(raw.size() / 2);
    const int64_t off[2] = {0, n};
    smilehip_batch *b = nullptr;
    check(smilehip_batch_create(pl, off, 1, &b));
    const int64_t lld_rows = smilehip_batch_total_rows(b);
    n_rows_ = func_ ? (lld_rows > 0 ? 1 : 0) : (long)lld_rows;   // no frame -> the reference writes no functionals instance
    rows_.assign((size_t)(n_rows_ > 0 ? n_rows_ : 1) * n_cols_, 0.0f);
    if (lld_rows > 0) {
      // the LLD level (and its functionals) stay on the device; only what the level below gets comes back
      void *d_pcm = nullptr, *d_lld = nullptr, *d_func = nullptr;
      check(smilehip_alloc(context(), (uint64_t)n * 2, &d_pcm));
      check(smilehip_alloc(context(), (uint64_t)lld_rows * n_lld_ * 4, &d_lld));
      check(smilehip_copy_to_device(context(), d_pcm, raw.data(), (uint64_t)n * 2, nullptr));
      check(smilehip_lld_run(pl, b, (const int16_t *)d_pcm, (float *)d_lld, n_lld_, nullptr));
      if (func_) {
        check(smilehip_alloc(context(), (uint64_t)n_cols_ * 4, &d_func));
        switch (kind_) {
          case kIs09:
            check(smilehip_batch_functionals(pl, b, (const float *)d_lld, n_lld_, smilehip_functionals_is09_mask(), (float *)d_func, n_cols_, nullptr));
            break;
          case kEgemaps: check(smilehip_batch_functionals_egemaps(pl, b, (float *)d_func, n_cols_, nullptr)); break;
          case kIs13: check(smilehip_batch_functionals_is13_compare(pl, b, (const float *)d_lld, n_lld_, (float *)d_func, n_cols_, nullptr)); break;
          default: check(smilehip_batch_functionals_compare16(pl, b, (const float *)d_lld, n_lld_, (float *)d_func, n_cols_, nullptr)); break;
        }
        check(smilehip_copy_to_host(context(), rows_.data(), d_func, (uint64_t)n_cols_ * 4, nullptr));
      } else {
        check(smilehip_copy_to_host(context(), rows_.data(), d_lld, (uint64_t)lld_rows * n_lld_ * 4, nullptr));
      }
      check(smilehip_stream_synchronize(context(), nullptr));
      smilehip_free(context(), d_pcm); smilehip_free(context(), d_lld);
      if (d_func) smilehip_free(context(), d_func);
    }
    // frame time stamps of the rows: the rows a window processor emits at end of input repeat the last frame's
    times_.assign((size_t)(n_rows_ > 0 ? n_rows_ : 1), 0.0);
    if (!func_) {
      const int64_t n_frames = (kind_ == kCompare16 || kind_ == kIs13 || kind_ == kEgemaps) ? lld_rows - 1 : smilehip_num_frames(pl, n);
      for (long t = 0; t < n_rows_; ++t) times_[(size_t)t] = smilehip_row_time(pl, n_frames, t);
    }
    smilehip_batch_destroy(b);
    smilehip_plan_destroy(pl);
    ran_ = true;
  }
 protected:
  SMILECOMPONENT_STATIC_DECL_PR
  void myFetchConfig() override {
    cDataSource::myFetchConfig();
    filename_ = getStr("filename") ? getStr("filename") : "";
    set_ = getStr("featureSet") ? getStr("featureSet") : "mfcc12_0_d_a";
    std::string lo;
    for (char ch : set_) lo += (char)tolower((unsigned char)ch);
    auto ends = [&](const char *suf) { const size_t k = strlen(suf); return lo.size() > k && lo.compare(lo.size() - k, k, suf) == 0; };
    func_ = ends("_func");
    const std::string base = (func_ || ends("_lld")) ? lo.substr(0, lo.rfind('_')) : lo;
    kind_ = base == "is09" ? kIs09 : base == "compare16" ? kCompare16 : base == "is13_compare" ? kIs13 : base == "egemapsv02" ? kEgemaps : kHtkVariant;
    if (kind_ == kHtkVariant && (func_ || ends("_lld")))
      COMP_ERR("cHipLldSource: unknown featureSet '%s'", set_.c_str());
    smilehip_lld_config c;
    config_for(c);
    period_sec_ = c.frame_step_sec;
    frame_size_sec_ = c.frame_size_sec;
    std::vector<std::string> lld_names;
    switch (kind_) {
      case kIs09: lld_names = smilehip_host::lld_names_is09(); break;
      case kCompare16: case kIs13: lld_names = smilehip_host::lld_names_compare16(); break;
      case kEgemaps: lld_names = smilehip_host::lld_names_egemaps(); break;
      default: lld_names = smilehip_host::lld_names_htk_variant(c.chain_kind == SMILEHIP_CHAIN_PLP, c.append_log_energy != 0); break;
    }
    n_lld_ = (int)lld_names.size();
    if (func_) {
      names_ = kind_ == kIs09 ? smilehip_host::func_names_is09() : kind_ == kEgemaps ? smilehip_host::func_names_egemaps() : smilehip_host::func_names_compare16();
      period_sec_ = 0.0;                                  // one vector per input, as cFunctionals in frameMode = full writes
    } else {
      names_ = lld_names;
    }
    n_cols_ = (int)names_.size();
  }
  int configureWriter(sDmLevelConfig &c) override {
    c.T = period_sec_;                                  // the level the chain's cVectorConcat writes: period = frameStep
    c.frameSizeSec = frame_size_sec_;
    c.basePeriod = period_sec_;
    return 1;
  }
  int setupNewNames(long) override {
    // element names "base[i]" back into array fields (field name, size, first index), as the chain's components add them
    size_t i = 0;
    while (i < names_.size()) {
      const std::string &nm = names_[i];
      const size_t br = nm.rfind('[');
      // functional names carry the element index in the middle ("mfcc_sma[3]_range"): one field each
      if (br == std::string::npos || nm.back() != ']') { writer_->addField(nm.c_str(), 1); ++i; continue; }
      const std::string base = nm.substr(0, br);
      const int first = atoi(nm.c_str() + br + 1);
      size_t j = i;
      while (j < names_.size() && names_[j].compare(0, br + 1, base + "[") == 0 && names_[j].rfind('[') == br) ++j;
      writer_->addField(base.c_str(), (int)(j - i), first);
      i = j;
    }
    namesAreSet_ = 1;
    return 1;
  }
  eTickResult myTick(long long) override {
    if (isEOI()) return TICK_INACTIVE;
    if (!ran_) run_once();
    long n = n_rows_ - next_;
    if (n <= 0) return TICK_INACTIVE;
    if (n > blocksizeW_ && blocksizeW_ > 0) n = blocksizeW_;
    if (n > 64) n = 64;
    if (!writer_->checkWrite(n)) {
      n = 1;
      if (!writer_->checkWrite(1)) return TICK_DEST_NO_SPACE;
    }
    if (!block_ || block_->nT != n) {
      delete block_;
      block_ = new cMatrix(n_cols_, n);
    }
    memcpy(block_->data, rows_.data() + (size_t)next_ * n_cols_, sizeof(float) * (size_t)n * n_cols_);   // data[el + t*N]
    for (long t = 0; t < n; ++t) {                      // frame time stamps as the framer gives them: vIdx * frameStep
      block_->tmeta[t].time = times_[(size_t)(next_ + t)];
      block_->tmeta[t].lengthSec = frame_size_sec_;
      block_->tmeta[t].period = period_sec_;
    }
    writer_->setNextMatrix(block_);
    next_ += n;
    return TICK_SUCCESS;
  }
 public:
  SMILECOMPONENT_STATIC_DECL
  explicit cHipLldSource(const char *n) : cDataSource(n) {}
  ~cHipLldSource() override { delete block_; }
};

SMILECOMPONENT_STATICS(cHipLldSource)

SMILECOMPONENT_REGCOMP(cHipLldSource) {
  SMILECOMPONENT_REGCOMP_INIT
  scname = COMPONENT_NAME_CHIPLLDSOURCE;
  sdescription = COMPONENT_DESCRIPTION_CHIPLLDSOURCE;
  SMILECOMPONENT_INHERIT_CONFIGTYPE("cDataSource")
  SMILECOMPONENT_IFNOTREGAGAIN(
    ct->setField("filename", "The RIFF/WAVE file to process (16-bit mono PCM)", "input.wav");
    ct->setField("featureSet", "The feature set whose rows are produced. Named after its file in config/mfcc or config/plp: mfcc12_0_d_a, mfcc12_e_d_a, mfcc12_0_d_a_z, mfcc12_e_d_a_z, plp_0_d_a, plp_e_d_a, plp_0_d_a_z, plp_e_d_a_z (LLD rows). <set>_lld with <set> = is09 | compare16 | is13_compare | egemapsv02: the LLD level of is09-13/IS09_emotion.conf (32 columns), compare16/ComParE_2016.conf / is09-13/IS13_ComParE.conf (130), egemaps/v02/eGeMAPSv02.conf (25), with the rows and time stamps the reference's LLD sinks see. <set>_func: the functionals level of the same files, one vector of 384 / 6373 / 6373 / 88 values per input", "mfcc12_0_d_a");
  )
  SMILECOMPONENT_MAKEINFO(cHipLldSource);
}

SMILECOMPONENT_CREATE(cHipLldSource)
