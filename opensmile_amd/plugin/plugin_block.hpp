// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, after plugin_shared.hpp):
// block-per-tick ticks behind the data memory.
//
// The reference's tick loop hands every component ONE frame per tick: cVectorProcessor::myTick (src/core/vectorProcessor.cpp:290-394)
// reads with getNextFrame, cWinToVecProcessor::myTick (src/core/winToVecProcessor.cpp:868-1098) cuts one frame out of the sample
// level with getNextMatrix (:983; src/core/dataReader.cpp:558-591), cWindowProcessor::myTick (src/core/windowProcessor.cpp:171-236)
// one block of `blocksize` frames (1 in the shipped files). A device round trip per frame and component is what made the
// per-component plugin 100 x slower than the CPU. Here an override takes EVERY frame its reader's level holds -- what the wave
// source wrote in its last tick (blocksize_sec, 1 s by default: src/iocore/waveSource.cpp:52) or what a host program pushed
// through cExternalAudioSource::writeData (src/iocore/externalAudioSource.cpp:132-160) -- in ONE tick: one getMatrix, the
// operator on n frames, one setNextMatrix. Reader and writer stay the component's own; every level gets real data, in the
// reference's order, with the reference's time stamps; the rows a tick does not take (a single frame, the padded blocks at the
// start and the end of input, flushes) go through the reference's own tick and the per-frame path below it.

// most frames the writer's level can take now (0: none)
template <class W>
long block_room(W *writer, long n) {
  if (n <= 0 || writer->checkWrite(n)) return n;
  const long f = writer->getNFree();
  return f < n ? (f > 0 ? f : 0) : n;
}

// How many of `avail` frames a block tick moves. A sink takes one frame per tick (cDataSink::myTick), so once the level in front of
// it is full every level upstream fills up too and room appears one frame at a time: taking it as it comes would put the whole graph
// back on one-frame ticks. A component whose writer's level is being drained therefore WAITS (its tick reports success and moves
// nothing) until a block's worth of room is there -- a quarter of the level, at most 256 frames -- and takes what there is as soon
// as a tick passes without the room growing (nobody drains: waiting longer could stall the graph). Returns the frames to move
// (>= 1), 0 = wait, -1 = nothing to do here (the reference's own tick reports why).
struct BlockGate {
  long last_room = -1, last_avail = -1;
  // true: fewer than block_min() frames (ticks' worth: units of `unit` frames) are there and more arrived since the last tick -- wait
  // for a block; false: take what is there (enough, or the input has stopped growing)
  bool input(long avail, long unit = 1) {
    if (avail >= block_min() * unit || avail <= last_avail) { last_avail = -1; return false; }
    last_avail = avail;
    return true;
  }
  template <class W>
  long frames(W *writer, long avail, long unit = 1) {
    long n = avail < block_cap() ? avail : block_cap();
    n -= n % unit;
    if (n < 1) return -1;
    long room = block_room(writer, n);
    room -= room % unit;
    if (room >= n) { last_room = -1; return n; }
    const sDmLevelConfig *c = writer->getLevelConfig();
    long target = c ? c->nT / 4 : 64;
    target = target > 256 ? 256 : (target < 2 ? 2 : target);
    target -= target % unit;
    if (target < unit) target = unit;
    if (target > n) target = n;
    if (room >= target) { last_room = -1; return room; }
    if (room > last_room) { last_room = room; return 0; }
    last_room = -1;
    return room >= 1 ? room : -1;
  }
};

// frames written so far to every level a reader reads (the slowest of them)
inline long reader_cur_w(cDataReader *r) {
  long w = r->dm->getCurW(r->level[0]);
  for (int i = 1; i < r->nLevels; ++i) { const long x = r->dm->getCurW(r->level[i]); if (x < w) w = x; }
  return w;
}

// rows [start, start + n) of a reader's level, if the override that wrote them left them on the device
inline const float *block_dev_rows(cDataReader *r, long start, long n, long N) {
  if (r->nLevels != 1) return nullptr;
  auto it = g_dev_rows.find(r->level[0]);
  if (it == g_dev_rows.end()) return nullptr;
  const DevRows &d = it->second;
  if (!d.d || d.N != N || start < d.start || start + n > d.start + d.n) return nullptr;
  return d.d + (size_t)(start - d.start) * (size_t)N;
}

// a matrix with room for `cap` frames whose nT is set per tick (cMatrix frees data / tmeta whatever nT says)
struct BlockMat {
  cMatrix *m = nullptr;
  long cap = 0, N = 0;
  bool pinned = false;
  void drop() {
    if (!m) return;
    if (pinned && g_ctx) smilehip_host_unregister(g_ctx, m->data);
    pinned = false;
    m->nT = cap;
    delete m;
    m = nullptr;
  }
  cMatrix *get(long n_el, long n) {
    if (!m || n > cap || n_el != N) {
      drop();
      cap = n > 256 ? n : 256;
      N = n_el;
      m = new cMatrix((int)n_el, (int)cap);
      // page-locked: the block's download is a DMA at the link's rate (SMILEHIP_NO_PINNED=1: plain memory)
      static const bool pin = !(getenv("SMILEHIP_NO_PINNED") && getenv("SMILEHIP_NO_PINNED")[0] == '1');
      pinned = pin && smilehip_host_register(context(), m->data, sizeof(FLOAT_DMEM) * (uint64_t)n_el * (uint64_t)cap) == SMILEHIP_OK;
    }
    m->nT = n;
    return m;
  }
  ~BlockMat() { drop(); }
};

// ---- cVectorProcessor descendants: the block tick around the component's own processVector
template <class B>
class BlockVP : public B {
 protected:
  BlockMat bout_;
  BlockGate gate_;
  // may this instance's processVector stand for a block (every operator call in it takes g_blk.n frames, its host-side
  // arithmetic loops over them)? Overrides whose option set has no such form return false and keep the one-frame ticks.
  virtual bool blockCapable() { return true; }
  int configureWriter(sDmLevelConfig &c) override {
    const int r = B::configureWriter(c);
    if (r && block_mode()) {                               // room for blocks in the levels either side (plugin_shared.hpp: block_frames)
      this->reader_->updateBlocksize(block_frames());
      if (c.blocksizeWriter < block_frames()) c.blocksizeWriter = block_frames();
    }
    return r;
  }
  eTickResult myTick(long long t) override {
    if (!block_mode() || this->isEOI() || this->processArrayFields == 2 || !blockCapable()) return B::myTick(t);
    cDataReader *rd = this->reader_;
    const long avail = rd->getNAvail();
    if (avail < 1 || rd->curR < 0) return B::myTick(t);
    if (gate_.input(avail)) return TICK_SUCCESS;           // a block is still arriving (BlockGate::input)
    if (avail < 2) return B::myTick(t);
    long n = gate_.frames(this->writer_, avail);
    BLOCK_DBG("%s: vector tick: curR %ld avail %ld n %ld", this->getInstName(), rd->curR, avail, n);
    if (n == 0) return TICK_SUCCESS;                       // waiting for a block's worth of room (BlockGate)
    if (n < 2) return B::myTick(t);
    const long s = rd->curR;
    const double t0 = now_sec();
    cMatrix *mat = rd->getMatrix(s, n);
    if (!mat) return B::myTick(t);
    if (mat->nT < n) n = mat->nT;
    rd->curR = s + n;
    rd->catchupCurR(s + n);                                // every frame of the block is consumed (n per-frame reads leave the level there)
    cMatrix *out = bout_.get(this->No, n);
    g_blk.n = n;
    g_blk.ld_src = mat->N;
    g_blk.ld_dst = this->No;
    g_blk.d_rows = block_dev_rows(rd, s, n, mat->N);
    g_blk.w_level = this->writer_->level;
    g_blk.w_start = this->writer_->dm->getCurW(this->writer_->level);
    const double t1 = now_sec();
    // the field walk of cVectorProcessor::myTick (vectorProcessor.cpp:344-377), one call per field for the whole block
    FLOAT_DMEM *dFi = mat->data, *dFo = out->data;
    int iO = 0, ret = 1, toSet = 1;
    for (int i = 0; i < this->Nfi; i++) {
      if ((this->fNi[i] == 1 && this->includeSingleElementFields == 0 && this->processArrayFields == 1) || (this->fNi[i] < 1)) continue;
      if (this->fNo[iO] <= 0) { g_blk = BlockCtx(); COMP_ERR("libsmilehip plugin: output field size for field %i is 0", iO); }
      const int res = this->processVector(dFi, dFo, this->fNi[i], this->fNo[iO], i);
      if (res == 0) ret = 0;
      else if (res < 0) toSet = 0;
      dFi += this->fNi[i];
      dFo += this->fNo[iO];
      iO++;
    }
    g_blk = BlockCtx();
    const double t2 = now_sec();
    if (!ret) toSet = 0;
    if (toSet) {
      out->setTimeMeta(mat->tmeta);                        // vecO->setTimeMeta(vec->tmeta), frame by frame
      this->writer_->setNextMatrix(out);
      out->setTimeMeta(nullptr);
    }
    g_t_read += t1 - t0; g_t_op += t2 - t1; g_t_write += now_sec() - t2;
    g_block_ticks++;
    g_block_frames += n;
    return ret ? TICK_SUCCESS : TICK_INACTIVE;
  }
 public:
  explicit BlockVP(const char *n) : B(n) {}
};

// ---- cFramer (cWinToVecProcessor in frameMode = fixed, one channel): every complete frame the sample level holds, in one tick
// (winToVecProcessor.cpp:983 getNextMatrix with stepM / lengthM, :1037-1052 the row copy, :1064-1072 the frame's time stamp)
class cHipFramer : public cFramer {
  BlockMat bout_;
  BlockGate gate_;
  FrameIO io_;
 protected:
  int configureWriter(sDmLevelConfig &c) override {
    const int r = cFramer::configureWriter(c);
    if (r && block_mode() && c.blocksizeWriter < block_frames()) c.blocksizeWriter = block_frames();   // (plugin_shared.hpp: block_frames)
    return r;
  }
  eTickResult myTick(long long t) override {
    cDataReader *rd = reader_;
    g_fused.init();
    // fused mode (plugin_shared.hpp): the samples of the wave level go to the batch, the frame level stays empty
    if (g_fused.active && rd->nLevels == 1 && Ni == 1) return isEOI() ? TICK_INACTIVE : g_fused.feed_tick(rd, this);
    if (!block_mode() || isEOI() || frameMode != FRAMEMODE_FIXED || allow_last_frame_incomplete_ || wholeMatrixMode || Ni != 1 ||
        rd->nLevels != 1 || rd->stepM <= 0 || rd->lengthM <= 0 || rd->curR < 0 || No != rd->lengthM)
      return cFramer::myTick(t);
    const long s = rd->curR, step = rd->stepM, len = rd->lengthM;
    const long have = rd->dm->getCurW(rd->level[0]) - s;
    const long avail = have >= len ? (have - len) / step + 1 : 0;
    if (avail < 2) return cFramer::myTick(t);
    const long n = gate_.frames(writer_, avail);
    BLOCK_DBG("%s: framer tick: curR %ld have %ld n %ld", getInstName(), s, have, n);
    if (n == 0) return TICK_SUCCESS;                       // waiting for a block's worth of room (BlockGate)
    if (n < 2) return cFramer::myTick(t);
    const long span = (n - 1) * step + len;
    cMatrix *mat = rd->getMatrix(s, span);
    if (!mat || mat->nT != span || mat->N != 1) return cFramer::myTick(t);    // (a read at s leaves the level where the reference's own read at s does)
    rd->curR = s + n * step;
    rd->catchupCurR(s + (n - 1) * step + 1);               // where n single reads leave the level's read index (dataMemoryLevel.cpp validateIdxRangeR)
    cMatrix *out = bout_.get(No, n);
    for (long f = 0; f < n; ++f) {
      memcpy(out->data + (size_t)f * (size_t)No, mat->data + (size_t)f * (size_t)step, sizeof(FLOAT_DMEM) * (size_t)len);
      // squashTimeMeta of the frame's sample matrix, then tmpVec->setTimeMeta (winToVecProcessor.cpp:1064-1072)
      TimeMetaInfo &tm = out->tmeta[f];
      const TimeMetaInfo &a = mat->tmeta[f * step], &z = mat->tmeta[f * step + len - 1];
      tm = a;
      tm.framePeriod = a.period;
      tm.lengthSec = z.time - a.time + z.lengthSec;
      if (frameCenterFrames > 0) tm.time += frameCenter;
    }
    // the same rows on the device for the override that reads this level next (no upload there): samples up, frames cut by the device
    io_.ensure(span, n * No);
    check(smilehip_copy_to_device(context(), io_.own_in, mat->data, sizeof(float) * (uint64_t)span, nullptr));
    check(smilehip_frame_rows(context(), io_.own_in, len, step, n, io_.d_out, No, nullptr));
    const long w0 = writer_->dm->getCurW(writer_->level);
    writer_->setNextMatrix(out);
    DevRows r; r.d = io_.d_out; r.start = w0; r.n = n; r.N = No;
    g_dev_rows[writer_->level] = r;
    io_.dev_level = writer_->level;
    g_block_ticks++;
    g_block_frames += n;
    g_framer_frames += n;
    return TICK_SUCCESS;
  }
 public:
  explicit cHipFramer(const char *n) : cFramer(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipFramer(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// ---- cWindowProcessor descendants (cDeltaRegression, cContourSmoother): the blocks between the padded ones at the two ends of the
// input, all of them in one tick. A tick of the reference reads frames [curR, curR + blocksize + pre + post) and writes `blocksize`
// frames (windowProcessor.cpp:171-236); every output value depends on its own window only, so k such ticks are one block of
// k * blocksize frames. The first block (curR = -pre: frames before the input padded by the reader) and the blocks at the end of
// input (padded, the R13 end-of-input rule) stay with the reference's tick.
struct WinBlock {
  BlockMat out;
  BlockGate gate;
  FrameIO io;
  bool waiting = false;                                    // the last tick() was a wait: the caller's tick reports success
  // room for blocks in the levels either side (plugin_shared.hpp: block_frames). cDataProcessor::myConfigureInstance copies the
  // writer's block size into blocksizeW_, which cWindowProcessor::myTick asks its level for before it writes ONE block
  // (windowProcessor.cpp:176): restore() puts the component's own value back.
  template <class C>
  void configure(C *c, sDmLevelConfig &cfg) {
    if (!block_mode()) return;
    own_bs_w = cfg.blocksizeWriter;
    c->reader_->updateBlocksize(block_frames() + c->winsize);
    if (cfg.blocksizeWriter < block_frames()) cfg.blocksizeWriter = block_frames();
  }
  long own_bs_w = -1;
  template <class C>
  void restore(C *c) { if (own_bs_w > 0) { c->blocksizeW_ = own_bs_w; own_bs_w = -1; } }
  // op: 0 cDeltaRegression, 1 cContourSmoother, 2 with noZeroSma, 3 cDeltaRegression with onlyInSegments (d_norm: its carried
  // divisor). Returns false when this tick is not a block tick. One device round trip for the block, whatever its size: a single
  // tick's worth too (the reference's own tick calls processBuffer once per ELEMENT -- a round trip each on the per-row path).
  template <class C>
  bool tick(C *c, int op, int W, int delta_flags, long *counter, float *d_norm = nullptr) {
    cDataReader *rd = c->reader_;
    const long bs = rd->stepM, win = c->winsize, pre = c->pre, post = c->post;
    if (!block_mode() || c->isEOI() || rd->nLevels < 1 || bs < 1 || rd->lengthM != bs + win || rd->curR < 0 || c->multiplier != 1 ||
        win != pre + post || pre < (W > 0 ? W : 1) || post < W) {
      BLOCK_DBG("%s: window tick refused: eoi %d levels %d bs %ld lengthM %ld win %ld curR %ld mult %d pre %ld post %ld W %d", c->getInstName(), (int)c->isEOI(),
                rd->nLevels, bs, rd->lengthM, win, rd->curR, (int)c->multiplier, pre, post, W);
      return false;
    }
    const long s = rd->curR;
    const long have = reader_cur_w(rd) - s;
    const long k = have >= bs + win ? (have - win) / bs : 0;
    waiting = false;
    if (k < 1) return false;
    if (gate.input(k * bs, bs)) { waiting = true; return true; }   // a block is still arriving (BlockGate::input)
    const long n = gate.frames(c->writer_, k * bs, bs);
    BLOCK_DBG("%s: window tick: curR %ld have %ld k %ld n %ld", c->getInstName(), s, have, k, n);
    if (n == 0) { waiting = true; return true; }           // waiting for a block's worth of room (BlockGate)
    if (n < bs) return false;
    cMatrix *mat = rd->getMatrix(s, n + win);
    if (!mat) return false;
    const long N = mat->N;
    if (mat->nT != n + win) { rd->catchupCurR(s); return false; }
    rd->curR = s + n;
    rd->catchupCurR(s + n - bs + 1);
    cMatrix *o = out.get(N, n);
    io.ensure((n + win) * N, n * N);
    check(smilehip_copy_to_device(context(), io.own_in, mat->data, sizeof(float) * (uint64_t)((n + win) * N), nullptr));
    if (op == 3)
      check(smilehip_delta_segments_block(context(), io.own_in + (size_t)pre * (size_t)N, N, io.d_out, N, n / bs, (int32_t)bs, (int32_t)N, W, delta_flags,
                                          d_norm, nullptr));
    else
      check(smilehip_window_op_block(context(), io.own_in + (size_t)pre * (size_t)N, N, io.d_out, N, n, (int32_t)N, op, W, delta_flags, nullptr));
    check(smilehip_copy_to_host(context(), o->data, io.d_out, sizeof(float) * (uint64_t)(n * N), nullptr));
    check(smilehip_stream_synchronize(context(), nullptr));
    o->setTimeMeta(mat->tmeta + pre);                      // matnew->setTimeMeta(mat->tmeta + pre), windowProcessor.cpp:222-224
    const long w0 = c->writer_->dm->getCurW(c->writer_->level);
    c->writer_->setNextMatrix(o);
    o->setTimeMeta(nullptr);
    DevRows r; r.d = io.d_out; r.start = w0; r.n = n; r.N = N;
    g_dev_rows[c->writer_->level] = r;
    io.dev_level = c->writer_->level;
    c->isFirstFrame = 0;
    g_block_ticks++;
    g_block_frames += n;
    *counter += n * N;                                     // the trace counts rows x elements, as the per-row path does
    return true;
  }
};

// ---- the sinks (cHtkSink, cCsvSink, cArffSink, cExternalSink: src/iocore/htkSink.cpp:183-202, csvSink.cpp:195-260, arffSink.cpp:336-400).
// A sink's tick writes ONE frame, so the whole graph's tick loop runs once per output frame however large the blocks upstream
// are: 60 components x 60 000 iterations of mostly idle ticks for ten minutes of audio. The override runs the reference's OWN tick
// -- same code, same bytes -- as often as its level holds frames, within one tick of the loop.
template <class S>
class LoopSink : public S {
 protected:
  eTickResult myTick(long long t) override {
    eTickResult r = S::myTick(t);
    if (!block_mode()) return r;
    for (long i = 1; i < block_cap() && (r == TICK_SUCCESS || r == TICK_INACTIVE) && this->reader_->getNAvail() > 0; ++i) {
      const eTickResult r2 = S::myTick(t);
      if (r2 == TICK_SUCCESS) r = r2;
      else if (r2 != TICK_INACTIVE) break;
    }
    return r;
  }
 public:
  explicit LoopSink(const char *n) : S(n) {}
};
#define LOOP_SINK(NAME, BASE)                                \
  class NAME : public LoopSink<BASE> {                       \
   public:                                                   \
    explicit NAME(const char *n) : LoopSink<BASE>(n) {}      \
    static cSmileComponent *create(const char *n) {          \
      cSmileComponent *c = new NAME(n);                      \
      c->setComponentInfo(scname, sdescription);             \
      return c;                                              \
    }                                                        \
  };
LOOP_SINK(cHipHtkSink, cHtkSink)
LOOP_SINK(cHipCsvSink, cCsvSink)
LOOP_SINK(cHipArffSink, cArffSink)
LOOP_SINK(cHipExternalSink, cExternalSink)
