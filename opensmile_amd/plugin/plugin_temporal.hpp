// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): R13: cDeltaRegression, cContourSmoother (window processors)
// R13  cWindowProcessor::processBuffer of cDeltaRegression (src/dspcore/deltaRegression.cpp:113-170) and
// cContourSmoother (src/dspcore/contourSmoother.cpp:85-118): one row of the block the window
// processor's tick hands over, valid on [-pre, nT+post)
// fused mode (plugin_shared.hpp): the window processor whose writer level the fused batch supplies writes the level's rows
// in blocks, whatever its reader holds (nothing: the wave source idles, no component upstream ever sees data)
struct FusedTick {
  const FusedLevel *lvl = nullptr;
  int tried = 0;
  long next = 0;
  cMatrix *block = nullptr;
  ~FusedTick() { delete block; }
  bool mine(const char *writer_level) {
    if (!tried) { tried = 1; g_fused.init(); if (g_fused.big && g_fused.tick_mode) lvl = g_fused.static_level(writer_level); }
    return g_fused.big && g_fused.tick_mode && lvl && !lvl->cols.empty();
  }
};

struct RowIO {
  FrameIO io;
  void run(cMatrix *in, cMatrix *out, int pre, int post, int kind, int W) {
    const long nT = out->nT;
    if (nT <= 0) return;
    io.ensure(nT + pre + post, nT);
    io.up(in->data - pre, nT + pre + post);
    check(smilehip_window_op_row_ex(context(), io.d_in + pre, io.d_out, nT, kind, W, d_norm, nullptr));
    io.down(out->data, nT);
  }
  void run_delta(cMatrix *in, cMatrix *out, int pre, int post, int W, int flags) {     // cDeltaRegression with option variants
    const long nT = out->nT;
    if (nT <= 0) return;
    io.ensure(nT + pre + post, nT);
    io.up(in->data - pre, nT + pre + post);
    check(smilehip_delta_op_row(context(), io.d_in + pre, io.d_out, nT, W, flags, d_norm, nullptr));
    io.down(out->data, nT);
  }
  float *d_norm = nullptr;                                 // kind 3: the instance's carried divisor (one device float)
};

class cHipDeltaRegression : public cDeltaRegression {
  RowIO row_;
  FusedTick ftick_;
  WinBlock wblock_;
  bool cpu_warned_ = false;
  int plain_ = -1, W_ = 0, segs_ = 0, flags_ = 0;
  DevBytes norm_;
  void options() {
    if (plain_ < 0) {
      W_ = getInt("deltawin");
      segs_ = getInt("onlyInSegments") ? 1 : 0;
      if (W_ < 0) W_ = 0;                                    // :72-75
      const int hw = getInt("halfWaveRect");
      flags_ = (getInt("relativeDelta") ? SMILEHIP_DELTA_RELATIVE : 0) | (hw ? SMILEHIP_DELTA_HALFWAVE : 0) |
               ((!hw && getInt("absOutput")) ? SMILEHIP_DELTA_ABS : 0) | (segs_ ? SMILEHIP_DELTA_SEGMENTS : 0);
      plain_ = (W_ > 0 && !(flags_ & ~SMILEHIP_DELTA_SEGMENTS)) ? 1 : 0;     // the two forms the window-op kernels have had since round 1 / 2
      if (segs_) {                                         // the norm member the onlyInSegments branch keeps adding to (:77-79, :129)
        float n0 = 0.0f;
        for (int i = 1; i <= W_; i++) n0 += (float)i * (float)i;
        n0 *= 2.0f;
        row_.d_norm = (float *)norm_.ensure(sizeof(float));
        check(smilehip_copy_to_device(context(), row_.d_norm, &n0, sizeof(float), nullptr));
      }
    }
  }
 protected:
  eTickResult myTick(long long t) override {
    if (ftick_.mine(getStr("writer.dmLevel"))) {
      if (isEOI()) return TICK_INACTIVE;
      return g_fused.tick_write(*ftick_.lvl, writer_, ftick_.next, ftick_.block, blocksizeW_);
    }
    // every block between the padded ones at the two ends of the input in one tick (plugin_block.hpp); onlyInSegments carries its
    // divisor from value to value in the order of the reference's ticks: its block operator walks the block in that order
    wblock_.restore(this);
    options();
    if (wblock_.tick(this, segs_ ? 3 : 0, W_, flags_ & ~SMILEHIP_DELTA_SEGMENTS, &g_frames[10], row_.d_norm)) return TICK_SUCCESS;
    return cDeltaRegression::myTick(t);
  }
  int configureWriter(sDmLevelConfig &c) override {
    const int r = cDeltaRegression::configureWriter(c);
    if (r) wblock_.configure(this, c);
    return r;
  }
  int processBuffer(cMatrix *in, cMatrix *out, int pre, int post) override {
    options();
    if (pre < (W_ > 0 ? W_ : 1) || post < W_) { HIP_FALLTHROUGH(10, "cDeltaRegression: a block without its window's history"); return cDeltaRegression::processBuffer(in, out, pre, post); }
    if (plain_) row_.run(in, out, pre, post, segs_ ? 3 : 0, W_);
    else row_.run_delta(in, out, pre, post, W_, flags_);   // relativeDelta / halfWaveRect / absOutput / deltawin = 0
    g_frames[10] += out->nT;
    return 1;
  }
 public:
  explicit cHipDeltaRegression(const char *n) : cDeltaRegression(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipDeltaRegression(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

class cHipContourSmoother : public cContourSmoother {
  RowIO row_;
  FusedTick ftick_;
  WinBlock wblock_;
  bool cpu_warned_ = false;
  int plain_ = -1, W_ = 0, nz_ = 0;
  void options() {
    if (plain_ < 0) {
      const int w = smaWin;                              // the member: myFetchConfig has made an even value odd (contourSmoother.cpp:64-67)
      W_ = w / 2;
      plain_ = ((w & 1) && W_ >= 1) ? 1 : 0;
      nz_ = getInt("noZeroSma") ? 1 : 0;
    }
  }
 protected:
  eTickResult myTick(long long t) override {
    if (ftick_.mine(getStr("writer.dmLevel"))) {
      if (isEOI()) return TICK_INACTIVE;
      return g_fused.tick_write(*ftick_.lvl, writer_, ftick_.next, ftick_.block, blocksizeW_);
    }
    wblock_.restore(this);
    options();
    if (plain_ && wblock_.tick(this, nz_ ? 2 : 1, W_, 0, &g_frames[11])) return TICK_SUCCESS;   // (plugin_block.hpp)
    return cContourSmoother::myTick(t);
  }
  int configureWriter(sDmLevelConfig &c) override {
    const int r = cContourSmoother::configureWriter(c);
    if (r) wblock_.configure(this, c);
    return r;
  }
  int processBuffer(cMatrix *in, cMatrix *out, int pre, int post) override {
    options();
    if (!plain_ || pre < W_ || post < W_) { HIP_FALLTHROUGH(11, "cContourSmoother: smaWin = 1 (no smoothing) is not built as a per-component operator"); return cContourSmoother::processBuffer(in, out, pre, post); }
    row_.run(in, out, pre, post, nz_ ? 2 : 1, W_);
    g_frames[11] += out->nT;
    return 1;
  }
 public:
  explicit cHipContourSmoother(const char *n) : cContourSmoother(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipContourSmoother(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};
