// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): shared state of the overrides: context, frame counters, the CPU fall-through rule, frame I/O buffers, fused mode for unmodified configuration files

// ---------------------------------------------------------------- shared state
smilehip_context *g_ctx = nullptr;
cConfigManager *g_confman = nullptr;     // the loader's configuration manager (registerPluginComponent's first argument)
constexpr int kNumOverrides = 28;
long g_frames[kNumOverrides] = {0};
long g_cpu[kNumOverrides] = {0};       // frames an overridden component handed to the reference's own CPU code (option set not built)
const char *const g_names[kNumOverrides] = {"cVectorPreemphasis", "cWindower", "cTransformFFT", "cFFTmagphase", "cMelspec", "cMfcc",
                                            "cEnergy", "cMZcr", "cAcf", "cPitchACF", "cDeltaRegression", "cContourSmoother", "cSpectral", "cPlp", "cFunctionals", "cSpecScale",
                                            "cPitchShs", "cSpecResample", "cLpc", "cFormantLpc", "cHarmonics", "cPitchSmootherViterbi", "cValbasedSelector", "cPitchJitter",
                                            "cIntensity", "cLsp", "cPitchSmoother", "cVectorOperation"};

// An override whose option set the HIP path does not cover is an ERROR of the component (COMP_ERR, like any configuration the
// reference cannot run): "the plugin is loaded" means "the frames were computed on the GPU". The shipped library has no other
// behaviour. A DEVELOPMENT build (make DEV_CPU_FALLTHROUGH=1: -DSMILEHIP_DEV_CPU_FALLTHROUGH) can, with SMILEHIP_PLUGIN_ALLOW_CPU=1
// in the environment, let such an instance run the reference's own code -- logged once (level-1 warning) and counted per frame
// (trace line "<type>.cpu <n>") -- to find out what ELSE of a new configuration file is not built; it keeps the one-frame ticks.
#ifdef SMILEHIP_DEV_CPU_FALLTHROUGH
inline bool allow_cpu() {
  static const int v = [] { const char *e = getenv("SMILEHIP_PLUGIN_ALLOW_CPU"); return (e && e[0] == '1') ? 1 : 0; }();
  return v != 0;
}
#define HIP_FALLTHROUGH(idx, why)                                                                                  \
  do {                                                                                                             \
    if (!allow_cpu())                                                                                              \
      COMP_ERR("libsmilehip plugin: %s (development build: SMILEHIP_PLUGIN_ALLOW_CPU=1 runs this instance on the reference's CPU code)", why); \
    if (!cpu_warned_) {                                                                                            \
      SMILE_IWRN(1, "libsmilehip plugin: %s -- this instance runs the reference's CPU code", why);                 \
      cpu_warned_ = true;                                                                                          \
    }                                                                                                              \
    g_cpu[idx]++;                                                                                                  \
  } while (0)
#else
inline bool allow_cpu() { return false; }
#define HIP_FALLTHROUGH(idx, why)                                                                                  \
  do {                                                                                                             \
    (void)cpu_warned_;                                                                                             \
    COMP_ERR("libsmilehip plugin: %s -- this option set is not built for the HIP path", why);                      \
  } while (0)
#endif

smilehip_context *context() {
  if (!g_ctx) {
    const char *dev = getenv("SMILEHIP_DEVICE");
    if (smilehip_init(dev ? atoi(dev) : 0, &g_ctx) != SMILEHIP_OK)
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
  }
  return g_ctx;
}

// ---------------------------------------------------------------- block-per-tick state (plugin_block.hpp)
// The reference moves ONE frame per component and tick (cVectorProcessor::myTick, src/core/vectorProcessor.cpp:290-394;
// cWinToVecProcessor::myTick, src/core/winToVecProcessor.cpp:868-1098). The overrides take every frame their reader's level
// holds in one tick: ONE getMatrix, one upload (none when the level's rows are still on the device), the component's operator on
// n frames, one download, ONE setNextMatrix. While such a tick runs, g_blk tells the component's own processVector -- the same
// code the per-frame path runs -- that its call stands for n frames: FrameIO sizes, uploads and downloads blocks.
struct BlockCtx {
  long n = 1;                            // frames the current processVector call stands for (1: the reference's own per-frame call)
  long ld_src = 0, ld_dst = 0;           // row pitch (floats) of the host matrices the call's src / dst point into
  const float *d_rows = nullptr;         // the reader level's rows of this block on the device ([n][ld_src]) when its writer left them there
  int w_level = -1;                      // the writer's level and the index its first row of the block gets: where a device copy of
  long w_start = 0;                      // the output is registered for the next component
};
BlockCtx g_blk;
long g_block_ticks = 0, g_block_frames = 0, g_block_dev_rows = 0, g_framer_frames = 0;   // trace: block ticks, frames moved by them, frames whose upload was skipped

// rows of a level that are still on the device (written by the override that filled the level, valid until that override's next tick)
struct DevRows { const float *d = nullptr; long start = 0, n = 0, N = 0; };
std::map<int, DevRows> g_dev_rows;        // by data-memory level index

inline bool block_mode() {               // SMILEHIP_PLUGIN_BLOCK=0: every override on the reference's own one-frame ticks (the round-3 parity vehicle)
  static const int v = [] { const char *e = getenv("SMILEHIP_PLUGIN_BLOCK"); return (e && e[0] == '0') ? 0 : 1; }();
  return v != 0 && !allow_cpu();
}
// The shipped files of the big sets give most levels a ring buffer of FIVE frames (config/shared/BufferModeRb.conf.inc: one-frame
// ticks need no more). A block-capable override announces block_frames() as its read and write block size while the levels are
// configured (cDataReader::updateBlocksize, sDmLevelConfig::blocksizeWriter): the data memory then sizes the level for it
// (cDataMemoryLevel::finaliseLevel, src/core/dataMemoryLevel.cpp:1343-1356: nT >= blocksizeReader + 2 blocksizeWriter). Capacity
// only: what is written to and read from a level does not change.
inline long block_frames() {
  static const long v = [] { const char *e = getenv("SMILEHIP_PLUGIN_BLOCK_FRAMES"); const long x = e ? atol(e) : 0; return x >= 1 ? x : 256L; }();
  return v;
}
// A component behind one that still moves a frame per tick (a component the plugin does not override, a tick-level override) would
// see one new frame per tick: it waits while its input keeps growing, until block_min() frames are there (BlockGate::input).
inline long block_min() {
  static const long v = [] { const char *e = getenv("SMILEHIP_PLUGIN_BLOCK_MIN"); const long x = e ? atol(e) : 0; return x >= 1 ? x : 32L; }();
  return v;
}
inline long block_cap() {                // most frames one block tick takes
  static const long v = [] { const char *e = getenv("SMILEHIP_PLUGIN_BLOCK_MAX"); const long x = e ? atol(e) : 0; return x >= 2 ? x : 4096L; }();
  return v;
}

// SMILEHIP_PLUGIN_DEBUG=1: why a tick was (not) a block tick, the first 200 decisions on stderr
inline bool block_debug() {
  static const int v = [] { const char *e = getenv("SMILEHIP_PLUGIN_DEBUG"); return (e && e[0] == '1') ? 1 : 0; }();
  static int left = 200;
  return v != 0 && left-- > 0;
}
#define BLOCK_DBG(...) do { if (block_debug()) { fprintf(stderr, "[smilehip block] " __VA_ARGS__); fputc('\n', stderr); } } while (0)

// device scratch for the frames of one processVector call: one frame in / one frame out on the reference's ticks, g_blk.n of each
// on a block tick
struct FrameIO {
  float *d_in = nullptr, *d_out = nullptr;               // what the kernels get
  float *own_in = nullptr;
  long cap_in = 0, cap_out = 0, w_in = 0, w_out = 0;
  int dev_level = -1;                                    // the level whose DevRows entry points at d_out
  void forget() { if (dev_level >= 0) { g_dev_rows.erase(dev_level); dev_level = -1; } }
  void ensure(long n_in, long n_out) {
    forget();                                            // d_out is about to be rewritten
    w_in = n_in; w_out = n_out;
    const long need_in = n_in * g_blk.n, need_out = n_out * g_blk.n;
    if (need_in > cap_in) {
      if (own_in) smilehip_free(context(), own_in);
      own_in = nullptr;
      if (smilehip_alloc(context(), sizeof(float) * (uint64_t)need_in, (void **)&own_in)) COMP_ERR("libsmilehip: %s", smilehip_last_error());
      cap_in = need_in;
    }
    if (need_out > cap_out) {
      if (d_out) smilehip_free(context(), d_out);
      d_out = nullptr;
      if (smilehip_alloc(context(), sizeof(float) * (uint64_t)need_out, (void **)&d_out)) COMP_ERR("libsmilehip: %s", smilehip_last_error());
      cap_out = need_out;
    }
    d_in = own_in;
  }
  void up(const FLOAT_DMEM *src, long n) {
    if (g_blk.n == 1) {
      if (smilehip_copy_to_device(context(), own_in, src, sizeof(float) * (uint64_t)n, nullptr)) COMP_ERR("libsmilehip: %s", smilehip_last_error());
      return;
    }
    if (g_blk.d_rows && n == g_blk.ld_src && n == w_in) {   // the field is the level's whole row and the rows never left the device
      d_in = const_cast<float *>(g_blk.d_rows);
      g_block_dev_rows += g_blk.n;
      return;
    }
    if (smilehip_copy_to_device_2d(context(), own_in, sizeof(float) * (uint64_t)w_in, src, sizeof(float) * (uint64_t)g_blk.ld_src,
                                   sizeof(float) * (uint64_t)n, (uint64_t)g_blk.n, nullptr))
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
  }
  // n floats of every frame's device row (pitch w_out) to the host rows at dst (pitch 1 frame / g_blk.ld_dst)
  void down(FLOAT_DMEM *dst, long n) {
    if (g_blk.n == 1) {
      if (smilehip_copy_to_host(context(), dst, d_out, sizeof(float) * (uint64_t)n, nullptr) || smilehip_stream_synchronize(context(), nullptr))
        COMP_ERR("libsmilehip: %s", smilehip_last_error());
      return;
    }
    if (smilehip_copy_to_host_2d(context(), dst, sizeof(float) * (uint64_t)g_blk.ld_dst, d_out, sizeof(float) * (uint64_t)w_out,
                                 sizeof(float) * (uint64_t)n, (uint64_t)g_blk.n, nullptr) ||
        smilehip_stream_synchronize(context(), nullptr))
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
    if (n == g_blk.ld_dst && n == w_out && g_blk.w_level >= 0) {   // the level's whole rows: the next override may take them from here
      DevRows r; r.d = d_out; r.start = g_blk.w_start; r.n = g_blk.n; r.N = n;
      g_dev_rows[g_blk.w_level] = r;
      dev_level = g_blk.w_level;
    }
  }
  // the device rows of the call's frames on the host as they are ([frames][w_out]), for outputs the component re-orders itself
  std::vector<float> rows_;
  const float *down_rows() {
    rows_.resize((size_t)w_out * (size_t)g_blk.n);
    if (smilehip_copy_to_host(context(), rows_.data(), d_out, sizeof(float) * (uint64_t)rows_.size(), nullptr) ||
        smilehip_stream_synchronize(context(), nullptr))
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
    return rows_.data();
  }
  ~FrameIO() {
    forget();
    if (g_ctx) {
      if (own_in) smilehip_free(g_ctx, own_in);
      if (d_out) smilehip_free(g_ctx, d_out);
    }
  }
};

void check(int rc) {
  if (rc != SMILEHIP_OK) COMP_ERR("libsmilehip: %s", smilehip_last_error());
}

// ---------------------------------------------------------------- fused mode for UNMODIFIED configuration files
// SMILEHIP_PLUGIN_FUSE=1: the first overridden component that is asked for a frame reads the process's own command line
// (-C file.conf and the options the file defines), parses the file with the host library's reader (conf_plan.cpp) and, if
// the graph is a cepstral chain, runs the WHOLE input file through the fused kernels in one batch. From then on the
// per-frame stages of the chain (pre-emphasis .. mel bank) only mark their frames, and cMfcc / cPlp / cEnergy copy their
// rows out of the batch result: one device round trip per file instead of one per frame and component. Everything
// downstream (mean normalisation, delta regression, concatenation, sinks) runs the reference's own code on those rows,
// at the reference's own ticks. Graphs that are not expressible stay on the per-component path (a warning says why).
long g_fused_stage = 0;
// a level the fused batch supplies: rows of a host matrix, a column per element of the level (no columns: zeros)
struct FusedLevel {
  const std::vector<float> *M = nullptr;
  int ld = 0;
  long n_rows = 0;
  std::vector<int> cols;
  std::vector<float> extra;      // one more row after the matrix's (ComParE group B's level holds row T60 + 1, which only its functionals read)
  bool lld_times = true;         // rows carry the LLD level's time stamps (FusedChain::times); false: row index x frame period (levels only functionals read)
};
struct FusedChain {
  bool tried = false, active = false;
  // big = an unmodified big-set file (IS09_emotion, ComParE_2016, IS13_ComParE, eGeMAPSv02 and its sub-graphs): the whole LLD
  // level comes from ONE fused batch; the components that write the levels the sinks and the cFunctionals instances read
  // (the final cContourSmoother / cDeltaRegression instances, eGeMAPS' energy level) hand out its rows, every overridden
  // component upstream of them writes zeros (nobody downstream of the fused levels reads those), the cFunctionals overrides
  // run as HIP operators on the handed-out levels.
  bool big = false;
  smilehip_host::ConfPlan plan;
  std::vector<float> rows, fin, b_extra;
  long f0_frames = 0, f0_pending = 0;                     // 60 ms frames of the file, and how many the Viterbi pass left to the end-of-input flush
  // the functionals vector of the fused batch and where a cFunctionals instance's values start in it (by the instance's writer level);
  // instances that are not listed run as HIP operators on the handed-out levels
  std::vector<float> func;
  struct FuncAt { long base; int count; };
  std::map<std::string, FuncAt> func_levels;
  long n_rows = 0;
  int n_cols = 0;
  long served = 0;
  std::map<std::string, FusedLevel> levels;
  FusedLevel zero_level;

  bool stage_level(const char *lvl) const {
    if (!active || !lvl) return false;
    if (big) return levels.find(lvl) == levels.end();      // everything that is not handed out is a stage
    for (const std::string &l : plan.stage_levels) if (l == lvl) return true;
    return false;
  }
  const FusedLevel *static_level(const char *lvl) const {
    if (!active || !lvl) return nullptr;
    auto it = levels.find(lvl);
    if (it != levels.end()) return &it->second;
    return big ? &zero_level : nullptr;
  }
  void add_level(const std::string &name, const std::vector<float> *M, int ld, long nr, int c0, int n) {
    FusedLevel L;
    L.M = M; L.ld = ld; L.n_rows = nr;
    for (int i = 0; i < n; ++i) L.cols.push_back(c0 + i);
    levels[name] = L;
  }
  // the big sets: one batch of the preset's chain over the whole file
  bool init_big(const smilehip_host::WaveInfo &wi, const std::vector<unsigned char> &raw) {
    const std::string &ps = plan.preset;
    smilehip_lld_config c;
    const bool egm = ps == "egemapsv02";                   // (the GeMAPS sub-graph files have other level names: per-component path)
    if (ps == "is09_emotion") smilehip_config_is09_lld(&c);
    else if (ps == "compare16") smilehip_config_compare16(&c);
    else if (ps == "is13_compare") smilehip_config_is13_compare(&c);
    else if (egm) smilehip_config_egemapsv02(&c);
    else return false;
    smilehip_host::conf_apply_f0_params(plan, c);
    c.sample_rate = (double)wi.sample_rate;
    smilehip_plan *pl = nullptr;
    check(smilehip_plan_create(context(), &c, &pl));
    smilehip_geometry g;
    check(smilehip_plan_geometry(pl, &g));
    const int64_t n = (int64_t)(raw.size() / 2);
    const int64_t off[2] = {0, n};
    smilehip_batch *b = nullptr;
    check(smilehip_batch_create(pl, off, 1, &b));
    n_rows = (long)smilehip_batch_total_rows(b);
    n_cols = g.n_out;
    rows.assign((size_t)(n_rows > 0 ? n_rows : 1) * n_cols, 0.0f);
    long fin_rows = 0;
    if (n_rows > 0) {
      void *d_pcm = nullptr, *d_lld = nullptr;
      check(smilehip_alloc(context(), (uint64_t)(n > 0 ? n : 1) * 2, &d_pcm));
      check(smilehip_alloc(context(), (uint64_t)n_rows * n_cols * 4, &d_lld));
      check(smilehip_copy_to_device(context(), d_pcm, raw.data(), (uint64_t)n * 2, nullptr));
      check(smilehip_lld_run(pl, b, (const int16_t *)d_pcm, (float *)d_lld, n_cols, nullptr));
      check(smilehip_copy_to_host(context(), rows.data(), d_lld, (uint64_t)n_rows * n_cols * 4, nullptr));
      if (ps != "is09_emotion") {
        const int32_t *d_pend = nullptr;
        int32_t pend = 0;
        check(smilehip_batch_f0_pending(b, &d_pend));
        check(smilehip_copy_to_host(context(), &pend, d_pend, sizeof(pend), nullptr));
        check(smilehip_stream_synchronize(context(), nullptr));
        f0_pending = pend;
        f0_frames = n_rows - 1;                            // rows = T60 + 1
      }
      if (ps == "compare16" || ps == "is13_compare") {     // row T60 + 1 of group B's smoothed / delta levels (its functionals read it)
        const float *d_ex = nullptr;
        check(smilehip_batch_compare_b_extra(b, &d_ex));
        b_extra.assign(110, 0.0f);
        check(smilehip_copy_to_host(context(), b_extra.data(), d_ex, 110 * 4, nullptr));
      }
      {                                                    // the set's functionals level, on the device-resident LLD matrix
        int nf = 0;
        if (ps == "is09_emotion") nf = 384;
        else if (ps == "compare16" || ps == "is13_compare") nf = smilehip_functionals_compare16_count();
        if (nf > 0) {
          void *d_func = nullptr;
          check(smilehip_alloc(context(), (uint64_t)nf * 4, &d_func));
          if (ps == "is09_emotion") check(smilehip_batch_functionals(pl, b, (const float *)d_lld, n_cols, smilehip_functionals_is09_mask(), (float *)d_func, nf, nullptr));
          else if (ps == "is13_compare") check(smilehip_batch_functionals_is13_compare(pl, b, (const float *)d_lld, n_cols, (float *)d_func, nf, nullptr));
          else check(smilehip_batch_functionals_compare16(pl, b, (const float *)d_lld, n_cols, (float *)d_func, nf, nullptr));
          func.assign((size_t)nf, 0.0f);
          check(smilehip_copy_to_host(context(), func.data(), d_func, (uint64_t)nf * 4, nullptr));
          check(smilehip_stream_synchronize(context(), nullptr));
          smilehip_free(context(), d_func);
          if (ps == "is09_emotion") func_levels["is09_func"] = FuncAt{0, 12};
          else {
            static const struct { const char *inst; int elems; } order[] = {{"A", 8}, {"B", 110}, {"Nz", 12}, {"F0", 1}, {"LLD", 59}, {"Delta", 59}};
            long base = 0;
            for (const auto &o : order) {
              smilehip_func_spec fs;
              check(ps == "is13_compare" ? smilehip_funcspec_is13_compare(o.inst, &fs) : smilehip_funcspec_compare16(o.inst, &fs));
              const int cnt = smilehip_funcspec_count(&fs);
              func_levels[std::string("is13_functionals") + o.inst] = FuncAt{base, cnt};
              base += (long)o.elems * cnt;
            }
            if (base != nf) COMP_ERR("libsmilehip plugin: fused mode: the functionals layout does not add up (%ld of %d)", base, nf);
          }
        }
      }
      if (egm) {                                           // the levels the functionals read (smilehip_batch_egemaps_taps)
        const float *d_fin = nullptr;
        check(smilehip_batch_egemaps_taps(b, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &d_fin, nullptr, nullptr));
        fin_rows = (long)smilehip_batch_total_frames(b) + 1;            // T20 + 1
        fin.assign((size_t)fin_rows * 36, 0.0f);
        check(smilehip_copy_to_host(context(), fin.data(), d_fin, (uint64_t)fin_rows * 36 * 4, nullptr));
      }
      check(smilehip_stream_synchronize(context(), nullptr));
      smilehip_free(context(), d_pcm); smilehip_free(context(), d_lld);
    }
    {                                                      // tick-level hand-out (round 5): the LLD level's time stamps, as cHipLldSource gives them
      const char *tm = getenv("SMILEHIP_PLUGIN_FUSE_TICK");
      // (eGeMAPS keeps the per-frame hand-out: its cFunctionals instances run as operators on the handed-out levels and count the
      // rows the reference's end-of-input rules leave in them; the other sets' functionals come from the batch's own vector)
      tick_mode = !egm && !(tm && !strcmp(tm, "0"));
      const int64_t n_frames = (ps == "is09_emotion") ? smilehip_num_frames(pl, n) : (int64_t)n_rows - 1;
      times.assign((size_t)(n_rows > 0 ? n_rows : 1), 0.0);
      for (long t = 0; t < n_rows; ++t) times[(size_t)t] = smilehip_row_time(pl, n_frames, t);
      frame_size_sec = c.frame_size_sec;
      frame_period_sec = c.frame_step_sec;
    }
    smilehip_batch_destroy(b);
    smilehip_plan_destroy(pl);
    if (ps == "is09_emotion") {
      add_level("is09_lld", &rows, n_cols, n_rows, 0, 16);
      add_level("is09_lld_de", &rows, n_cols, n_rows, 16, 16);
    } else if (ps == "compare16" || ps == "is13_compare") {
      add_level("is13_lld_nzsmo", &rows, n_cols, n_rows, 0, 6);
      add_level("is13_lldA_smo", &rows, n_cols, n_rows, 6, 4);
      add_level("is13_lldB_smo", &rows, n_cols, n_rows, 10, 55);
      add_level("is13_lld_nzsmo_de", &rows, n_cols, n_rows, 65, 6);
      add_level("is13_lldA_smo_de", &rows, n_cols, n_rows, 71, 4);
      add_level("is13_lldB_smo_de", &rows, n_cols, n_rows, 75, 55);
      if (b_extra.size() == 110) {
        levels["is13_lldB_smo"].extra.assign(b_extra.begin(), b_extra.begin() + 55);
        levels["is13_lldB_smo_de"].extra.assign(b_extra.begin() + 55, b_extra.end());
      }
    } else {
      const std::string g1 = "gemapsv01b";               // names of the included core file
      // the LLD level's two halves, and the levels the functionals read (lld_params.hpp: func_in's 36 columns)
      add_level("egemapsv02_lldsetE_smo", &rows, n_cols, n_rows, 0, 10);
      add_level("egemapsv02_lldsetF_smo", &rows, n_cols, n_rows, 10, 15);
      add_level(g1 + "_loudness_smo", &fin, 36, fin_rows, 0, 1);
      add_level("egemapsv02_lldSetNoF0AndLoudnessZ_smo", &fin, 36, fin_rows, 1, 5);
      add_level(g1 + "_lld_single_logF0_smo", &fin, 36, fin_rows, 6, 1);
      add_level("egemapsv02_lldSetNoF0AndLoudnessNz_smo", &fin, 36, fin_rows, 7, 14);
      add_level("egemapsv02_lldSetSpectralNz_smo", &fin, 36, fin_rows, 21, 9);
      add_level("egemapsv02_lldSetSpectralZ_smo", &fin, 36, fin_rows, 30, 5);
      add_level("egemapsv02_energyRMS", &fin, 36, fin_rows, 35, 1);
      for (auto &kv : levels) if (kv.second.M == &fin) kv.second.lld_times = false;
    }
    big = true;
    active = true;
    SMILE_MSG(2, "libsmilehip plugin: fused mode -- %s: %ld rows of '%s' in one batch", plan.describe.c_str(), n_rows, plan.wave_file.c_str());
    return true;
  }
  void init() {
    if (tried) return;
    tried = true;
    // Round 5: fused mode is the DEFAULT whenever the file's graph is one the host library's reader recognises (conf_plan.cpp) and
    // the input is a 16-bit mono wave file; SMILEHIP_PLUGIN_FUSE=0 keeps every component on its own operator (one device round
    // trip per frame and component: the parity vehicle, ~100 x slower than the fused batch), =1 asks for it loudly (warnings say
    // why a file does not fuse; by default those are messages of level 3).
    const char *on = getenv("SMILEHIP_PLUGIN_FUSE");
    if (on && !strcmp(on, "0")) return;
    const bool loud = on && *on;
#define FUSE_NOTE(...) do { if (loud) { SMILE_WRN(1, __VA_ARGS__); } else { SMILE_MSG(3, __VA_ARGS__); } } while (0)
    // The graph is the one the loader's cConfigManager holds (registerPluginComponent stored the pointer): its file reader keeps
    // every section of the configuration file as raw "field = value" lines, includes expanded (configManager.hpp:475-482, :567-610),
    // and its command-line parser knows the effective value of every \\cm[...] option the file defines -- whether the host is
    // SMILExtract (-C file -I wav ...) or a program that called smile_initialize with a file and an option list.
    std::string err;
    smilehip_host::ConfFile cf;
    {
      std::vector<smilehip_host::ConfRawSection> sections;
      cConfigManager *cm = g_confman;
      for (int r = 0; cm && r < cm->nReaders; ++r) {
        cFileConfigReader *fr = dynamic_cast<cFileConfigReader *>(cm->reader[r]);
        for (int i = 0; fr && i < fr->nInst_; ++i) {
          const fileInstance &fi = fr->inst_[i];
          smilehip_host::ConfRawSection sec;
          sec.name = fi.name ? fi.name : "";
          sec.type = fi.type ? fi.type : "";
          for (int l = 0; l < fi.N; ++l) if (fi.lines[l]) sec.lines.push_back(fi.lines[l]);
          sections.push_back(sec);
        }
      }
      cCommandlineParser *cp = cm ? cm->cmdparser : nullptr;
      const smilehip_host::ConfCmValue cm_value = [cp](const std::string &name, std::string &value) {
        const sCmdlineOpt *o = cp ? cp->findOpt(name.c_str()) : nullptr;
        if (!o) return false;
        char buf[64];
        switch (o->type) {                                 // (the host program may have registered an option with a type of its own: SMILExtract's -start / -end are doubles)
          case eCmdlineOptType::Str: value = o->getStr(); return true;
          case eCmdlineOptType::Int: if (!o->isSet) return false; snprintf(buf, sizeof(buf), "%d", o->getInt()); value = buf; return true;
          case eCmdlineOptType::Double: if (!o->isSet) return false; snprintf(buf, sizeof(buf), "%.17g", o->getDouble()); value = buf; return true;
          default: if (!o->isSet) return false; value = o->getBoolean() ? "1" : "0"; return true;
        }
      };
      if (sections.empty() || !smilehip_host::conf_from_sections(sections, cm_value, cf, err)) {
        FUSE_NOTE("libsmilehip plugin: fused mode: the configuration manager holds no file the host library's reader understands (%s) -- block-per-tick path", err.c_str());
        return;
      }
    }
    if (!smilehip_host::conf_to_plan(cf, plan, err)) {
      FUSE_NOTE("libsmilehip plugin: fused mode: %s -- block-per-tick path", err.c_str());
      return;
    }
    smilehip_host::WaveInfo wi;
    std::vector<unsigned char> raw;
    if (!smilehip_host::read_wave_file(plan.wave_file, wi, raw, err) || wi.sample_type != 1 || wi.n_bps != 2 || wi.n_chan != 1) {
      FUSE_NOTE("libsmilehip plugin: fused mode: '%s' is not a 16-bit mono PCM file (%s) -- block-per-tick path", plan.wave_file.c_str(), err.c_str());
      return;
    }
    if (!plan.preset.empty()) {
      // the big sets fuse only with EVERY override registered: the final smoother / delta instances must be the ones that hand out rows
      const char *only = getenv("SMILEHIP_PLUGIN_COMPONENTS");
      const bool all = !only || !*only || !strcmp(only, "all");
      if (!all || plan.last_mfcc > 0 || !plan.func_enabled.empty() || !init_big(wi, raw))
        FUSE_NOTE("libsmilehip plugin: fused mode: this big-set file does not fuse inside the reference process -- block-per-tick path");
      return;
    }
    smilehip_lld_config c = plan.cfg;
    c.sample_rate = (double)wi.sample_rate;
    // Where the rows are handed out. FINAL (round 5): the sinks read one level written by a cVectorConcat, every override is
    // registered -> the batch computes the file's whole output level (mean normalisation, regression stages: what smilextract_hip
    // writes for the same file) and cHipVectorConcat writes it at the tick level; nothing upstream of it ever ticks with data.
    // Otherwise the static block only: mean normalisation, deltas and concatenation stay with the reference's components.
    {
      const char *only = getenv("SMILEHIP_PLUGIN_COMPONENTS"), *tm = getenv("SMILEHIP_PLUGIN_FUSE_TICK");
      const bool all = !only || !*only || !strcmp(only, "all");
      final_level = all && plan.out_writer_type == "cVectorConcat" && !(tm && (!strcmp(tm, "0") || !strcmp(tm, "static")));
    }
    if (!final_level) {
      c.n_delta = 0;                                      // the static block is all the chain components hand on;
      c.cms = 0;                                          // mean normalisation and deltas stay with the reference's components
    }
    smilehip_plan *pl = nullptr;
    check(smilehip_plan_create(context(), &c, &pl));
    smilehip_geometry g;
    check(smilehip_plan_geometry(pl, &g));
    const int64_t n = (int64_t)(raw.size() / 2);
    const int64_t off[2] = {0, n};
    smilehip_batch *b = nullptr;
    check(smilehip_batch_create(pl, off, 1, &b));
    n_rows = (long)smilehip_batch_total_rows(b);
    n_cols = g.n_out;
    rows.assign((size_t)(n_rows > 0 ? n_rows : 1) * n_cols, 0.0f);
    if (n_rows > 0) check(smilehip_lld_run_host(pl, b, reinterpret_cast<const int16_t *>(raw.data()), n, rows.data()));
    {                                                      // tick-level hand-out: needs the wave source's override (it idles) -- every component registered
      const char *only = getenv("SMILEHIP_PLUGIN_COMPONENTS"), *tm = getenv("SMILEHIP_PLUGIN_FUSE_TICK");
      const bool all = !only || !*only || !strcmp(only, "all");
      tick_mode = all && !(tm && !strcmp(tm, "0"));        // SMILEHIP_PLUGIN_FUSE_TICK=0: the per-frame hand-out of round 3 (A/B switch)
      const int64_t n_frames = smilehip_num_frames(pl, n);
      times.assign((size_t)(n_rows > 0 ? n_rows : 1), 0.0);
      for (long t = 0; t < n_rows; ++t) times[(size_t)t] = smilehip_row_time(pl, n_frames, t);
      frame_size_sec = c.frame_size_sec;
      frame_period_sec = c.frame_step_sec;
    }
    smilehip_batch_destroy(b);
    smilehip_plan_destroy(pl);
    if (final_level) {
      add_level(plan.out_levels, &rows, n_cols, n_rows, 0, n_cols);
    } else {
      for (const auto &kv : plan.static_levels) {
        FusedLevel L;
        L.M = &rows; L.ld = n_cols; L.n_rows = n_rows; L.cols = kv.second;
        levels[kv.first] = L;
      }
    }
    active = true;
    SMILE_MSG(2, "libsmilehip plugin: fused mode -- %s: %ld frames of '%s' in one batch", plan.describe.c_str(), n_rows, plan.wave_file.c_str());
  }
  // ---- tick-level hand-out (round 5; the cepstral chains). The per-frame hand-out above still walks the reference's tick loop
  // once per frame for every component of the chain (framer, pre-emphasis, window, transform, magnitudes, mel bank: six levels
  // written and read per frame to carry zeros) -- an hour of audio took LONGER than the CPU binary. Here the wave source idles
  // (cHipWaveSource: nothing upstream of the chain's last components ever sees data) and those last components write their rows
  // straight to their levels, a block of frames per tick, with the time stamps the framer would have given them.
  bool tick_mode = false;
  bool final_level = false;                               // the rows are the sinks' own level (cHipVectorConcat hands them out)
  std::vector<double> times;                              // frame time stamps of the batch's rows (smilehip_row_time)
  double frame_size_sec = 0.0, frame_period_sec = 0.0;
  // writes the next rows of L to `writer`; next = rows written so far by this component; block: the component's own matrix
  eTickResult tick_write(const FusedLevel &L, cDataWriter *writer, long &next, cMatrix *&block, long blocksize_w) {
    const long total = L.n_rows + (L.extra.empty() ? 0 : 1);
    long n = total - next;
    if (n <= 0 || L.cols.empty()) return TICK_INACTIVE;
    if (blocksize_w > 0 && n > blocksize_w) n = blocksize_w;
    if (n > 256) n = 256;
    while (n > 1 && !writer->checkWrite(n)) n >>= 1;
    if (!writer->checkWrite(n)) return TICK_DEST_NO_SPACE;
    const long N = (long)L.cols.size();
    if (!block || block->nT != n || block->N != N) { delete block; block = new cMatrix((int)N, (int)n); }
    for (long t = 0; t < n; ++t) {
      const long row = next + t;
      FLOAT_DMEM *d = block->data + (size_t)t * N;         // data[el + t * N]
      if (row < L.n_rows) {
        const float *r = L.M->data() + (size_t)row * L.ld;
        for (long k = 0; k < N; ++k) d[k] = r[L.cols[(size_t)k]];
      } else {
        for (long k = 0; k < N; ++k) d[k] = L.extra[(size_t)k];
      }
      block->tmeta[t].time = (L.lld_times && row < (long)times.size()) ? times[(size_t)row] : (double)row * frame_period_sec;
      block->tmeta[t].lengthSec = frame_size_sec;
      block->tmeta[t].period = frame_period_sec;
    }
    writer->setNextMatrix(block);
    next += n;
    served += n;
    return TICK_SUCCESS;
  }
  // row `frame` of a fused level
  void copy(const FusedLevel &L, long frame, FLOAT_DMEM *dst, long Ndst) {
    if (L.cols.empty()) { for (long k = 0; k < Ndst; ++k) dst[k] = 0; return; }
    if (frame >= L.n_rows) COMP_ERR("libsmilehip plugin: fused mode: the graph asks for frame %ld, the batch has %ld", frame, L.n_rows);
    const float *r = L.M->data() + (size_t)frame * L.ld;
    for (long k = 0; k < Ndst && k < (long)L.cols.size(); ++k) dst[k] = r[L.cols[k]];
    ++served;
  }
  float at(const FusedLevel &L, long frame, int elem) {
    if (L.cols.empty()) return 0.0f;
    if (elem >= (int)L.cols.size())
      COMP_ERR("libsmilehip plugin: fused mode: the graph asks for element %d, the level has %d", elem, (int)L.cols.size());
    if (frame == L.n_rows && !L.extra.empty()) return L.extra[(size_t)elem];
    if (frame >= L.n_rows) { ++beyond; return 0.0f; }      // rows a window processor emits at the end of input that no reader of the level uses
    return (*L.M)[(size_t)frame * L.ld + L.cols[(size_t)elem]];
  }
  long beyond = 0;
};
// big-set fused mode: an overridden component that does not write a handed-out level fills its output with zeros
#define FUSED_BIG_STAGE(ret)                                                                                       \
  do {                                                                                                             \
    g_fused.init();                                                                                                \
    if (g_fused.big) { for (long k_ = 0; k_ < Ndst; ++k_) dst[k_] = 0; g_fused_stage++; return (ret); }           \
  } while (0)
FusedChain g_fused;

smilehip_lld_config base_config(long N, uint32_t stages) {
  smilehip_lld_config c;
  smilehip_config_mfcc12_0_d_a(&c);
  c.force_frame_size = N;
  c.stage_mask = stages;
  c.n_delta = 0;
  return c;
}

int winfunc_id(const char *s) {
  // same prefixes cWindower accepts (winFuncToInt, smileUtil.c)
  if (!s) return -1;
  if (!strncasecmp(s, "han", 3)) return SMILEHIP_WIN_HANN;
  if (!strncasecmp(s, "ham", 3)) return SMILEHIP_WIN_HAMM;
  if (!strncasecmp(s, "rec", 3)) return SMILEHIP_WIN_RECT;
  if (!strncasecmp(s, "gau", 3)) return SMILEHIP_WIN_GAUSS;
  if (!strncasecmp(s, "sin", 3) || !strncasecmp(s, "cos", 3)) return SMILEHIP_WIN_SINE;
  if (!strncasecmp(s, "tri", 3)) return SMILEHIP_WIN_TRI;
  if (!strncasecmp(s, "bar", 3) && strncasecmp(s, "barth", 5)) return SMILEHIP_WIN_BARTLETT;
  if (!strncasecmp(s, "lac", 3)) return SMILEHIP_WIN_LANCZOS;
  return -1;
}
