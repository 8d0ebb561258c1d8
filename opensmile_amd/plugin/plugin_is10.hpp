// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): the components of IS10_paraling / IS11_speaker_state / IS12_speaker_trait: cIntensity, cLsp, cPitchSmoother, cVectorOperation
// ---- the components the other INTERSPEECH sets of config/is09-13 add (IS10_paraling, IS11_speaker_state, IS12_speaker_trait) ----
// cIntensity::processVector (src/lldcore/intensity.cpp:125-145)
class cHipIntensity : public BlockVP<cIntensity> {
  FrameIO io_;
  bool cpu_warned_ = false;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (Nsrc == 0) return 0;
    const int flags = (intensity ? 1 : 0) | (loudness ? 2 : 0);
    if (!hamWin || nWin != Nsrc || !flags || Ndst != (intensity ? 1 : 0) + (loudness ? 1 : 0)) {
      HIP_FALLTHROUGH(24, "cIntensity: a window of another length than the frame is not built");
      return cIntensity::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_intensity_frames(context(), io_.d_in, Nsrc, Nsrc, flags, io_.d_out, Ndst, g_blk.n, nullptr));
    io_.down(dst, Ndst);
    g_frames[24] += g_blk.n;
    return (int)Ndst;
  }
 public:
  explicit cHipIntensity(const char *n) : BlockVP<cIntensity>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipIntensity(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cLsp::processVector (src/lld/lsp.cpp:289-312)
class cHipLsp : public BlockVP<cLsp> {
  FrameIO io_;
  bool cpu_warned_ = false;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (Ndst < Nsrc) return 0;
    if ((lpcIdx == -1) || (nLpc <= 0)) return 0;
    if (nLpc < 2 || nLpc > 32 || lpcIdx + nLpc > Nsrc) {
      HIP_FALLTHROUGH(25, "cLsp: more than 32 LP coefficients are not built");
      return cLsp::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    io_.ensure(nLpc, nLpc);
    io_.up(src + lpcIdx, nLpc);
    check(smilehip_lsp_frames(context(), io_.d_in, nLpc, (int32_t)nLpc, io_.d_out, nLpc, g_blk.n, nullptr));
    io_.down(dst, nLpc);
    g_frames[25] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipLsp(const char *n) : BlockVP<cLsp>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipLsp(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cPitchSmoother::processVector (src/lldcore/pitchSmoother.cpp:236-425): one input level, medianFilter0 = 0, post smoothing none /
// simple. The state the component carries from frame to frame lives on the device (32 bytes), the frame is one row of
// [F0Cand | candVoicing | candScore].
class cHipPitchSmoother : public cPitchSmoother {
  FrameIO io_;
  DevBytes state_, written_;
  std::vector<float> row_;
  bool cpu_warned_ = false;
  int usable_ = -1;
  bool started_ = false;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    const int flags = (F0final ? 1 : 0) | (F0finalEnv ? 2 : 0) | (voicingFinalClipped ? 4 : 0) | (voicingFinalUnclipped ? 8 : 0);
    if (usable_ < 0) {
      usable_ = nInputLevels == 1 && medianFilter0 == 0 && postSmoothingMethod != POSTSMOOTHING_MEDIAN && !no0f0 && !F0raw && !voicingC1 &&
                !voicingClip && flags && nCandidates[0] >= 1 && nCandidates[0] <= 16 && f0candI[0] >= 0 && candVoiceI[0] >= 0 &&
                candScoreI[0] >= 0;
    }
    if (!usable_) {
      HIP_FALLTHROUGH(26, "cPitchSmoother: several input levels, medianFilter0, median post smoothing, no0f0 and the copied fields are not built");
      return cPitchSmoother::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    const int c = nCandidates[0];
    int n_out = 0;
    for (int b = 0; b < 4; ++b) n_out += (flags >> b) & 1;
    row_.resize(3 * (size_t)c);
    for (int j = 0; j < c; ++j) {
      row_[j] = src[f0candI[0] + j];
      row_[c + j] = src[candVoiceI[0] + j];
      row_[2 * c + j] = src[candScoreI[0] + j];
    }
    io_.ensure(3 * c, 4);
    io_.up(row_.data(), 3 * c);
    const bool simple = postSmoothing && postSmoothingMethod == POSTSMOOTHING_SIMPLE;
    check(smilehip_pitch_smoother_rows(context(), c, voicingCutoff[0], octaveCorrection, simple ? 1 : 0, flags, io_.d_in, 3 * c, nullptr, 1, 1,
                                       state_.ensure(32), started_ ? 1 : 0, io_.d_out, 4, (int64_t *)written_.ensure(8), nullptr));
    started_ = true;
    int64_t wrote = 0;
    written_.down(&wrote, 8);
    g_frames[26]++;
    if (wrote < 1) return 0;                               // the first frame with simple post smoothing: no output (:331)
    float out[4];
    io_.down(out, 4);
    for (int i = 0; i < n_out && i < Ndst; ++i) dst[i] = out[i];
    return n_out;
  }
 public:
  explicit cHipPitchSmoother(const char *n) : cPitchSmoother(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipPitchSmoother(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cVectorOperation::processVector, the element-wise operations (src/other/vectorOperation.cpp:360-435, 508-527)
class cHipVectorOperation : public BlockVP<cVectorOperation> {
  FrameIO io_;
  bool cpu_warned_ = false;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    int op = -1;
    switch (operation) {
      case VOP_ADD: op = SMILEHIP_VOP_ADD; break;
      case VOP_MUL: op = SMILEHIP_VOP_MUL; break;
      case VOP_LOG: op = SMILEHIP_VOP_LOG; break;
      case VOP_LOGA: op = SMILEHIP_VOP_LOGA; break;
      case VOP_SQRT: op = SMILEHIP_VOP_SQRT; break;
      case VOP_E: op = SMILEHIP_VOP_E; break;
      case VOP_ABS: op = SMILEHIP_VOP_ABS; break;
      case VOP_DB_POW: op = SMILEHIP_VOP_DB_POW; break;
      case VOP_DB_MAG: op = SMILEHIP_VOP_DB_MAG; break;
      case VOP_X_SUM: op = SMILEHIP_VOP_X_SUM; break;
      case VOP_X_SUMSQ: op = SMILEHIP_VOP_X_SUMSQ; break;
      case VOP_X_L1: op = SMILEHIP_VOP_X_L1; break;
      case VOP_X_L2: op = SMILEHIP_VOP_X_L2; break;
    }
    if (op < 0) {
      HIP_FALLTHROUGH(27, "cVectorOperation: only add, mul, log, lgA, sqr, ee, abs, dBp, dBv and sum, ssm, ll1, ll2 are built");
      return cVectorOperation::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    const bool reduce = op >= SMILEHIP_VOP_X_SUM;
    const long n = reduce ? Nsrc : (Nsrc < Ndst ? Nsrc : Ndst);
    if (n < 1 || Ndst < 1) return 0;
    io_.ensure(n, n);
    io_.up(src, n);
    check(smilehip_vecop_frames(context(), op, param1, logfloor, io_.d_in, n, (int32_t)n, io_.d_out, n, g_blk.n, nullptr));
    io_.down(dst, reduce ? 1 : n);
    g_frames[27] += g_blk.n;
    return 1;
  }
 public:
  explicit cHipVectorOperation(const char *n) : BlockVP<cVectorOperation>(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipVectorOperation(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};
