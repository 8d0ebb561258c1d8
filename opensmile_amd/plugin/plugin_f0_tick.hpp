// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): SURVEY 8(f) rank 2, tick level: cValbasedSelector, cPitchSmootherViterbi
// cValbasedSelector::myTick (src/other/valbasedSelector.cpp:139-247) -- tick-level: a frame may be handed on, replaced by a
// constant vector, or dropped, so myTick itself is replaced. The decision and the output vector come from the device
// (smilehip_valbased_select_frames); the adaptive (running-average) threshold is not built.
class cHipValbasedSelector : public cValbasedSelector {
  FrameIO io_;
  DevBytes keep_;
  bool ready_ = false, cpu_warned_ = false;
  long idx_ = 0;
  int removeIdx_ = 0, invert_ = 0, allowEqual_ = 0, zerovec_ = 0, adaptive_ = 0;
  FLOAT_DMEM outputVal_ = 0, threshold_ = 0;
  cVector *my_ = nullptr;
  BlockGate gate_;
  BlockMat bout_;
  std::vector<int32_t> keeps_;
  std::vector<float> rows_;
  // every frame the reader's level holds in one tick (plugin_block.hpp): the decisions and the rows of the block from one operator
  // call, the frames that are handed on written as one matrix, each with its own time stamp. Returns false: not a block tick.
  bool blockTick(eTickResult &res) {
    cDataReader *rd = reader_;
    if (!block_mode() || isEOI() || rd->curR < 0) return false;
    const long avail = rd->getNAvail();
    if (avail < 1) return false;
    if (gate_.input(avail)) { res = TICK_SUCCESS; return true; }
    if (avail < 2) return false;
    const long n = gate_.frames(writer_, avail);
    if (n == 0) { res = TICK_SUCCESS; return true; }
    if (n < 2) return false;
    const long s = rd->curR;
    cMatrix *mat = rd->getMatrix(s, n);
    if (!mat || mat->nT != n) return false;
    const long N = mat->N, nOut = removeIdx_ ? N - 1 : N;
    if (nOut < 1) return false;
    rd->curR = s + n;
    rd->catchupCurR(s + n);
    io_.ensure(n * N, n * nOut);
    check(smilehip_copy_to_device(context(), io_.own_in, mat->data, sizeof(float) * (uint64_t)(n * N), nullptr));
    int32_t *d_keep = (int32_t *)keep_.ensure(sizeof(int32_t) * (uint64_t)n);
    check(smilehip_valbased_select_frames(context(), io_.own_in, N, N, n, (int32_t)idx_, threshold_, invert_, allowEqual_, zerovec_, removeIdx_,
                                          outputVal_, io_.d_out, nOut, d_keep, nullptr));
    rows_.resize((size_t)(n * nOut));
    keeps_.resize((size_t)n);
    check(smilehip_copy_to_host(context(), rows_.data(), io_.d_out, sizeof(float) * rows_.size(), nullptr));
    keep_.down(keeps_.data(), sizeof(int32_t) * (uint64_t)n);
    cMatrix *out = bout_.get(nOut, n);
    long m = 0;
    for (long f = 0; f < n; ++f) {
      if (!keeps_[(size_t)f]) continue;
      memcpy(out->data + (size_t)m * (size_t)nOut, rows_.data() + (size_t)f * (size_t)nOut, sizeof(float) * (size_t)nOut);
      out->tmeta[m] = mat->tmeta[f];                       // my_->setTimeMeta(vec->tmeta)
      ++m;
    }
    g_frames[22] += n;
    g_block_ticks++;
    g_block_frames += n;
    if (m > 0) { out->nT = m; writer_->setNextMatrix(out); }
    res = TICK_SUCCESS;
    return true;
  }
 protected:
  int configureWriter(sDmLevelConfig &c) override {
    const int r = cValbasedSelector::configureWriter(c);
    if (r && block_mode()) {                               // room for blocks in the levels either side (plugin_shared.hpp: block_frames)
      reader_->updateBlocksize(block_frames());
      if (c.blocksizeWriter < block_frames()) c.blocksizeWriter = block_frames();
    }
    return r;
  }
  eTickResult myTick(long long t) override {
    g_fused.init();
    if (g_fused.big) return TICK_INACTIVE;                 // big-set fused mode: the selected levels come from the batch; no frame ever arrives here
    if (!ready_) {
      threshold_ = (FLOAT_DMEM)getDouble("threshold");
      adaptive_ = (int)getInt("adaptiveThreshold");
      idx_ = getInt("idx"); invert_ = getInt("invert"); allowEqual_ = getInt("allowEqual");
      removeIdx_ = getInt("removeIdx"); zerovec_ = getInt("zeroVec");
      outputVal_ = (FLOAT_DMEM)getDouble("outputVal");
      ready_ = true;
    }
    if (adaptive_) { HIP_FALLTHROUGH(22, "cValbasedSelector: adaptiveThreshold = 1 is not built"); return cValbasedSelector::myTick(t); }
    { eTickResult r; if (blockTick(r)) return r; }
    if (!writer_->checkWrite(1)) return TICK_DEST_NO_SPACE;
    cVector *vec = reader_->getNextFrame();
    if (vec == NULL) return TICK_SOURCE_NOT_AVAIL;
    const long N = vec->N, nOut = removeIdx_ ? N - 1 : N;
    if (nOut < 1) { HIP_FALLTHROUGH(22, "cValbasedSelector: removeIdx on a one-element vector"); return TICK_INACTIVE; }
    io_.ensure(N, nOut);
    io_.up(vec->data, N);
    int32_t *d_keep = (int32_t *)keep_.ensure(sizeof(int32_t));
    check(smilehip_valbased_select_frames(context(), io_.own_in, N, N, 1, (int32_t)idx_, threshold_, invert_, allowEqual_, zerovec_, removeIdx_,
                                          outputVal_, io_.d_out, nOut, d_keep, nullptr));
    if (my_ == NULL || my_->N != nOut) { delete my_; my_ = new cVector((int)nOut); }
    io_.down(my_->data, nOut);
    int32_t keep = 0;
    keep_.down(&keep, sizeof(keep));
    g_frames[22]++;
    if (keep) {
      my_->setTimeMeta(vec->tmeta);
      writer_->setNextFrame(my_);
    }
    return TICK_SUCCESS;
  }
 public:
  explicit cHipValbasedSelector(const char *n) : cValbasedSelector(n) {}
  ~cHipValbasedSelector() override { delete my_; }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipValbasedSelector(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cPitchSmootherViterbi::myTick (src/lld/pitchSmootherViterbi.cpp:451-564) -- a TICK-LEVEL override: the component keeps its
// own buffering (frames are released when all surviving paths agree, or when the path buffer is full, the rest at end of
// input), so what is replaced is myTick itself. One frame of candidates per tick goes to the device-resident trellis
// (smilehip_viterbi_stream_push); the frames it reports as decided are written at this very tick, exactly as the
// reference's incremental scheme does -- the components behind (cPitchJitter does not run during end-of-input ticks, the
// window processors pad at end of input) see the same frames at the same ticks. Configuration, names, the second reader
// for the time meta and the writer are the base class's. Six candidates (cPitchShs nCandidates = 6) are built.
class cHipPitchSmootherViterbi : public cPitchSmootherViterbi {
  smilehip_viterbi_stream *vs_ = nullptr;
  bool ready_ = false, usable_ = false, cpu_warned_ = false, flushed_ = false;
  long nCand_ = 0;
  long f0I_ = -1, cvI_ = -1, rawI_ = -1, clipI_ = -1, c1I_ = -1;
  int oF0_ = 0, oLog_ = 0, oEnv_ = 0, oEnvLog_ = 0, oVc_ = 0, oVu_ = 0, oRaw_ = 0, oC1_ = 0, oClip_ = 0;
  FLOAT_DMEM thresh_ = 0, lastValid_ = 0;
  std::vector<std::vector<FLOAT_DMEM>> hist_;            // what the reference keeps per frame: (F0, voicing) x 6 | F0raw | voicingClip | voicingC1 | vIdx
  std::vector<std::pair<int, int>> queue_;               // decided (frame, state), not yet written
  size_t qpos_ = 0;
  cVector *vec_ = nullptr;
  long outN_ = 0;

  void setup() {
    ready_ = true;
    oF0_ = getInt("F0final"); oLog_ = getInt("F0finalLog"); oEnv_ = getInt("F0finalEnv"); oEnvLog_ = getInt("F0finalEnvLog");
    oVc_ = getInt("voicingFinalClipped"); oVu_ = getInt("voicingFinalUnclipped");
    oRaw_ = getInt("F0raw"); oC1_ = getInt("voicingC1"); oClip_ = getInt("voicingClip");
    outN_ = oF0_ + oLog_ + oEnv_ + oEnvLog_ + oVc_ + oVu_ + oRaw_ + oC1_ + oClip_;
    int more = 0;
    f0I_ = findField("F0Cand", 0, &nCand_, NULL, -1, &more);
    cvI_ = findField("candVoicing");
    if (oRaw_) rawI_ = findField("F0raw");
    if (oClip_) clipI_ = findField("voicingClip");
    if (oC1_) c1I_ = findField("voicingC1");
    cVectorMeta *md = reader_->getLevelMetaDataPtr(0);
    if (md != NULL) thresh_ = md->fData[0];                // the voicing cut-off cPitchShs publishes with its level
    const int buflen = getInt("bufferLength");
    usable_ = f0I_ >= 0 && cvI_ >= 0 && nCand_ >= 1 && nCand_ <= 6 && more == 0 && buflen >= 2 && buflen <= 128 && reader_->getNLevels() == 1;
    if (!usable_) return;
    // cSmileViterbiPitchSmooth::setWeights stores tvv into wTvvd as well (pitchSmootherViterbi.hpp:291-299)
    const double w[6] = {getDouble("wLocal"), getDouble("wTvv"), getDouble("wTvv"), getDouble("wTvuv"), getDouble("wThr"), getDouble("wRange")};
    check(smilehip_viterbi_stream_create(context(), buflen, thresh_, w, &vs_));
    check(smilehip_viterbi_stream_set_candidates(vs_, (int32_t)nCand_));
  }
  static FLOAT_DMEM semitone(FLOAT_DMEM f0) {            // :512-519, in the reference's own float arithmetic
    FLOAT_DMEM sc = 0.0;
    if (f0 > 29.136) sc = (FLOAT_DMEM)12.0 * log(f0 / (FLOAT_DMEM)27.5) / log((FLOAT_DMEM)2.0);
    else if (f0 > 0.0) sc = 1.0;
    return sc;
  }
  // the decided frames that are not written yet, as many as the writer's level takes (pitchSmootherViterbi.cpp:497-560)
  eTickResult write_decided() {
    if (qpos_ >= queue_.size()) return TICK_INACTIVE;
    if (vec_ == NULL) vec_ = new cVector((int)outN_);
    const size_t first = qpos_;
    for (; qpos_ < queue_.size(); ++qpos_) {
      if (!writer_->checkWrite(1)) return qpos_ == first ? TICK_DEST_NO_SPACE : TICK_SUCCESS;
      const std::vector<FLOAT_DMEM> &h = hist_[(size_t)queue_[qpos_].first];
      const int state = queue_[qpos_].second;
      FLOAT_DMEM f0 = state < nCand_ ? h[(size_t)(2 * state)] : 0.0f;            // getStateValueFromFrame
      long k = 0;
      if (oF0_) vec_->data[k++] = f0;
      if (oLog_) vec_->data[k++] = semitone(f0);
      if (oEnv_ || oEnvLog_) {
        if (f0 <= 0.0) f0 = lastValid_; else lastValid_ = f0;
        if (oEnv_) vec_->data[k++] = f0;
        if (oEnvLog_) vec_->data[k++] = semitone(f0);
      }
      const FLOAT_DMEM vp = state < nCand_ ? h[(size_t)(2 * state + 1)] : h[1];
      if (oVc_) vec_->data[k++] = vp >= thresh_ ? vp : 0.0f;
      if (oVu_) vec_->data[k++] = vp;
      if (oRaw_) vec_->data[k++] = h[(size_t)(2 * nCand_)];
      if (oC1_) vec_->data[k++] = h[(size_t)(2 * nCand_ + 1)];
      if (oClip_) vec_->data[k++] = h[(size_t)(2 * nCand_ + 2)];
      cVector *vin = reader2->getFrame((long)h[(size_t)(2 * nCand_ + 3)]);
      if (vin != NULL) vec_->setTimeMeta(vin->tmeta);
      writer_->setNextFrame(vec_);
    }
    return TICK_SUCCESS;
  }
  BlockGate gate_;
  std::vector<int32_t> fr_, st_;
  std::vector<float> cf_, cv_;
 protected:
  int configureWriter(sDmLevelConfig &c) override {
    const int r = cPitchSmootherViterbi::configureWriter(c);
    if (r && block_mode()) {                               // room for blocks in the levels either side (plugin_shared.hpp: block_frames)
      reader_->updateBlocksize(block_frames());
      if (c.blocksizeWriter < block_frames()) c.blocksizeWriter = block_frames();
    }
    return r;
  }
  eTickResult myTick(long long t) override {
    g_fused.init();
    if (!ready_) setup();
    if (!usable_) {
      HIP_FALLTHROUGH(21, "cPitchSmootherViterbi: only one input level with up to six candidates and bufferLength <= 128 is built");
      return cPitchSmootherViterbi::myTick(t);
    }
    int32_t n = 0;
    fr_.resize(128); st_.resize(128);
    int32_t *fr = fr_.data(), *st = st_.data();
    if (g_fused.big) {
      return TICK_INACTIVE;                                // big-set fused mode: the pitch contour comes from the batch; no candidates ever arrive here
    } else if (isEOI()) {
      if (!flushed_) {
        check(smilehip_viterbi_stream_flush(vs_, &n, fr, st, 128));
        flushed_ = true;
      }
    } else {
      // every frame of candidates the input level holds goes to the trellis in ONE call (plugin_block.hpp; one frame: the reference's
      // own pace); what becomes decided is written below, in order, as the single pushes would have reported it
      cDataReader *rd = reader_;
      long avail = block_mode() ? rd->getNAvail() : 1;
      if (avail >= 1 && block_mode() && qpos_ >= queue_.size() && gate_.input(avail)) return TICK_SUCCESS;   // a block is still arriving (BlockGate::input)
      if (avail > block_cap()) avail = block_cap();
      cMatrix *mat = (avail >= 2 && rd->curR >= 0) ? rd->getMatrix(rd->curR, avail) : nullptr;
      if (mat != NULL && mat->nT == avail) {
        const long nf = avail, N = mat->N;
        rd->curR += nf;
        rd->catchupCurR(rd->curR);
        cf_.assign((size_t)nf * 6, 0.0f); cv_.assign((size_t)nf * 6, 0.0f);
        for (long f = 0; f < nf; ++f) {
          const FLOAT_DMEM *row = mat->data + (size_t)f * (size_t)N;
          std::vector<FLOAT_DMEM> h((size_t)(2 * nCand_ + 4), 0.0f);
          for (long i = 0; i < nCand_; i++) {
            h[(size_t)(2 * i)] = cf_[(size_t)f * 6 + (size_t)i] = row[f0I_ + i];
            h[(size_t)(2 * i + 1)] = cv_[(size_t)f * 6 + (size_t)i] = row[cvI_ + i];
          }
          h[(size_t)(2 * nCand_)] = rawI_ >= 0 ? row[rawI_] : 0.0f;
          h[(size_t)(2 * nCand_ + 1)] = clipI_ > 0 ? row[clipI_] : 0.0f;
          h[(size_t)(2 * nCand_ + 2)] = c1I_ > 0 ? row[c1I_] : 0.0f;
          h[(size_t)(2 * nCand_ + 3)] = (FLOAT_DMEM)mat->tmeta[f].vIdx;
          hist_.push_back(h);
        }
        fr_.resize((size_t)nf + 128); st_.resize((size_t)nf + 128);
        fr = fr_.data(); st = st_.data();
        check(smilehip_viterbi_stream_push_frames(vs_, cf_.data(), cv_.data(), 6, (int32_t)nf, &n, fr, st, (int32_t)nf + 128));
        g_frames[21] += nf;
        g_block_ticks++;
        g_block_frames += nf;
      } else {
      cVector *vec = reader_->getNextFrame();
      if (vec == NULL) return qpos_ < queue_.size() ? write_decided() : TICK_SOURCE_NOT_AVAIL;
      std::vector<FLOAT_DMEM> h((size_t)(2 * nCand_ + 4), 0.0f);
      float cf[6], cv[6];
      for (long i = 0; i < nCand_; i++) {
        h[(size_t)(2 * i)] = cf[i] = vec->data[f0I_ + i];
        h[(size_t)(2 * i + 1)] = cv[i] = vec->data[cvI_ + i];
      }
      h[(size_t)(2 * nCand_)] = rawI_ >= 0 ? vec->data[rawI_] : 0.0f;
      h[(size_t)(2 * nCand_ + 1)] = clipI_ > 0 ? vec->data[clipI_] : 0.0f;       // (the reference tests > 0 for these two, :478-482)
      h[(size_t)(2 * nCand_ + 2)] = c1I_ > 0 ? vec->data[c1I_] : 0.0f;
      h[(size_t)(2 * nCand_ + 3)] = (FLOAT_DMEM)vec->tmeta->vIdx;
      hist_.push_back(h);
      check(smilehip_viterbi_stream_push(vs_, cf, cv, &n, fr, st, 128));
      g_frames[21]++;
      }
    }
    for (int i = 0; i < n; ++i) queue_.push_back(std::make_pair((int)fr[i], (int)st[i]));
    return write_decided();
  }
 public:
  explicit cHipPitchSmootherViterbi(const char *n) : cPitchSmootherViterbi(n) {}
  ~cHipPitchSmootherViterbi() override {
    if (vs_) smilehip_viterbi_stream_destroy(vs_);
    delete vec_;
  }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipPitchSmootherViterbi(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};
