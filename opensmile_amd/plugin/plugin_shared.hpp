// libsmilehip_plugin.so, part of smilehip_plugin.cpp (included there, inside its unnamed namespace, in this order;
// one translation unit: the parts share the state of plugin_shared.hpp): shared state of the overrides: context, frame counters, the CPU fall-through rule, frame I/O buffers, fused mode for unmodified configuration files

// ---------------------------------------------------------------- shared state
smilehip_context *g_ctx = nullptr;
cConfigManager *g_confman = nullptr;     // the loader's configuration manager and component manager (registerPluginComponent's arguments)
cComponentManager *g_compman = nullptr;
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

// The device is opened (runtime start-up, code objects, a first launch: ~0.3 s) on a thread of its own as soon as the loader asks the
// plugin for its components, beside the host's own start-up -- reading the configuration, creating the instances, the wave source's
// first blocks. The first override that needs the device waits for it here.
struct DeviceStart {
  std::thread th;
  smilehip_context *ctx = nullptr;
  int rc = 0;
  std::string err;
  bool started = false;
  void start() {
    if (started) return;
    started = true;
    th = std::thread([this] {
      const char *dev = getenv("SMILEHIP_DEVICE");
      rc = smilehip_init(dev ? atoi(dev) : 0, &ctx);
      if (rc != SMILEHIP_OK) { err = smilehip_last_error(); return; }
      void *d = nullptr;                                   // the library's code objects load with the first launch
      if (smilehip_alloc(ctx, 16, &d) == SMILEHIP_OK) {
        smilehip_pcm16_to_float(ctx, (const int16_t *)d, 1, (float *)((char *)d + 8), nullptr);
        smilehip_stream_synchronize(ctx, nullptr);
        smilehip_free(ctx, d);
      }
    });
  }
  ~DeviceStart() { if (th.joinable()) th.join(); }
} g_device_start;

smilehip_context *context() {
  if (!g_ctx) {
    if (g_device_start.started) {
      if (g_device_start.th.joinable()) g_device_start.th.join();
      if (g_device_start.rc != SMILEHIP_OK) COMP_ERR("libsmilehip: %s", g_device_start.err.c_str());
      g_ctx = g_device_start.ctx;
    } else {
      const char *dev = getenv("SMILEHIP_DEVICE");
      if (smilehip_init(dev ? atoi(dev) : 0, &g_ctx) != SMILEHIP_OK)
        COMP_ERR("libsmilehip: %s", smilehip_last_error());
    }
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

// SMILEHIP_PLUGIN_TIMING=1: where a block tick's time goes (seconds, summed over the run; trace lines "time.<phase>")
double g_t_read = 0, g_t_op = 0, g_t_write = 0, g_t_win = 0, g_t_framer = 0, g_t_up = 0, g_t_down = 0;
inline bool block_timing() {
  static const int v = [] { const char *e = getenv("SMILEHIP_PLUGIN_TIMING"); return (e && e[0] == '1') ? 1 : 0; }();
  return v != 0;
}
inline double now_sec() {
  if (!block_timing()) return 0.0;
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

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
    const double t0 = now_sec();
    up_(src, n);
    g_t_up += now_sec() - t0;
  }
  void down(FLOAT_DMEM *dst, long n) {
    const double t0 = now_sec();
    down_(dst, n);
    g_t_down += now_sec() - t0;
  }
  void up_(const FLOAT_DMEM *src, long n) {
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
  void down_(FLOAT_DMEM *dst, long n) {
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
// A graph the host library's reader recognises (a cepstral chain of config/mfcc / config/plp, or one of the big sets:
// IS09_emotion, ComParE_2016, IS13_ComParE, eGeMAPSv02) whose audio comes from a cWaveSource -- a FINITE input -- runs through the
// fused kernels as ONE batch, behind the data memory on both sides:
//   * the graph is the one the loader's cConfigManager holds (no second look at the command line, no second reading of the file);
//   * the samples are the ones the reference's own cWaveSource writes to its level: the framer overrides (cHipFramer) take them
//     from the level with their readers, block after block, and keep them; nothing else of the chain sees data (the levels between
//     the framers and the chain's last components stay empty -- not zero-filled);
//   * when the source has delivered its last block, the batch runs once (smilehip_lld_run on the int16 image of the samples when
//     every sample is one -- a 16-bit file --, smilehip_lld_run_f32 otherwise: any sample format / channel mix-down), and the
//     chain's LAST components write the finished rows to their own levels with setNextMatrix, a block per tick, with the time
//     stamps the framer would have given them (FusedChain::tick_write). Sinks, cFunctionals instances and everything else
//     downstream read them as they read the reference's rows.
// An input that is not a cWaveSource (cExternalAudioSource, a live source: no end to wait for), a graph the reader does not
// recognise, or a run with only some overrides registered takes the block-per-tick path (plugin_block.hpp) instead.
// SMILEHIP_PLUGIN_FUSE=0: block-per-tick everywhere; =1: say loudly why a graph does not fuse.
long g_fused_stage = 0;
// a level the fused batch supplies: rows of a host matrix, a column per element of the level
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
  bool ran = false;                                       // the batch has run: the rows exist
  bool source_eof = false;                                // the wave source has written its last block (cHipWaveSource)
  std::vector<float> pcm;                                 // the wave level's samples, as the framer overrides took them from it
  const void *feeder = nullptr;                           // the framer instance that keeps the samples (the others only drain their readers)
  double sample_rate = 16000.0;
  // big = an unmodified big-set file (IS09_emotion, ComParE_2016, IS13_ComParE, eGeMAPSv02): the whole LLD level comes from ONE
  // fused batch; the components that write the levels the sinks and the cFunctionals instances read (the final cContourSmoother /
  // cDeltaRegression instances, eGeMAPS' energy level) hand out its rows; the functionals of IS09 / ComParE / IS13 are the batch's
  // own vector, eGeMAPS' cFunctionals instances run as HIP operators on the handed-out levels.
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
  void add_level(const std::string &name, const std::vector<float> *M, int ld, int c0, int n, bool lld_t = true) {
    FusedLevel L;
    L.M = M; L.ld = ld; L.n_rows = 0; L.lld_times = lld_t;
    for (int i = 0; i < n; ++i) L.cols.push_back(c0 + i);
    levels[name] = L;
  }
  void config_big(smilehip_lld_config &c) const {
    const std::string &ps = plan.preset;
    if (ps == "is09_emotion") smilehip_config_is09_lld(&c);
    else if (ps == "compare16") smilehip_config_compare16(&c);
    else if (ps == "is13_compare") smilehip_config_is13_compare(&c);
    else smilehip_config_egemapsv02(&c);
    smilehip_host::conf_apply_f0_params(plan, c);
    c.sample_rate = sample_rate;
  }
  // the levels the batch will supply, by name (the overrides look theirs up at their first tick; the rows arrive with run())
  void declare_levels() {
    const std::string &ps = plan.preset;
    if (ps.empty()) {
      if (final_level) add_level(plan.out_levels, &rows, 0, 0, 0);
      else for (const auto &kv : plan.static_levels) { FusedLevel L; L.M = &rows; L.cols = kv.second; levels[kv.first] = L; }
    } else if (ps == "is09_emotion") {
      add_level("is09_lld", &rows, 32, 0, 16);
      add_level("is09_lld_de", &rows, 32, 16, 16);
    } else if (ps == "compare16" || ps == "is13_compare") {
      add_level("is13_lld_nzsmo", &rows, 130, 0, 6);
      add_level("is13_lldA_smo", &rows, 130, 6, 4);
      add_level("is13_lldB_smo", &rows, 130, 10, 55);
      add_level("is13_lld_nzsmo_de", &rows, 130, 65, 6);
      add_level("is13_lldA_smo_de", &rows, 130, 71, 4);
      add_level("is13_lldB_smo_de", &rows, 130, 75, 55);
    } else {
      const std::string g1 = "gemapsv01b";               // names of the included core file
      // the LLD level's two halves, and the levels the functionals read (lld_params.hpp: func_in's 36 columns)
      add_level("egemapsv02_lldsetE_smo", &rows, 25, 0, 10);
      add_level("egemapsv02_lldsetF_smo", &rows, 25, 10, 15);
      add_level(g1 + "_loudness_smo", &fin, 36, 0, 1, false);
      add_level("egemapsv02_lldSetNoF0AndLoudnessZ_smo", &fin, 36, 1, 5, false);
      add_level(g1 + "_lld_single_logF0_smo", &fin, 36, 6, 1, false);
      add_level("egemapsv02_lldSetNoF0AndLoudnessNz_smo", &fin, 36, 7, 14, false);
      add_level("egemapsv02_lldSetSpectralNz_smo", &fin, 36, 21, 9, false);
      add_level("egemapsv02_lldSetSpectralZ_smo", &fin, 36, 30, 5, false);
      add_level("egemapsv02_energyRMS", &fin, 36, 35, 1, false);
    }
  }
  void set_rows(const std::vector<float> *M, long nr, int ld) {
    for (auto &kv : levels) if (kv.second.M == M) { kv.second.n_rows = nr; kv.second.ld = ld; }
  }
  // The batch, once: every sample the wave source wrote is in `pcm`.
  void run() {
    if (ran) return;
    ran = true;
    const int64_t n = (int64_t)pcm.size();
    // the int16 image of the samples, when every sample is one (cWaveSource's 16-bit conversion is s / 32767.0f, smileUtil.c:2527-2535)
    std::vector<int16_t> s16((size_t)(n > 0 ? n : 1));
    bool is16 = true;
    for (int64_t i = 0; i < n && is16; ++i) {
      const float r = nearbyintf(pcm[(size_t)i] * 32767.0f);
      is16 = r >= -32768.0f && r <= 32767.0f && (float)(int16_t)r / 32767.0f == pcm[(size_t)i];
      s16[(size_t)i] = (int16_t)r;
    }
    smilehip_lld_config c;
    const std::string &ps = plan.preset;
    const bool egm = ps == "egemapsv02";
    if (!ps.empty()) config_big(c);
    else {
      c = plan.cfg;
      c.sample_rate = sample_rate;
      if (!final_level) { c.n_delta = 0; c.cms = 0; }    // the static block only: mean normalisation and deltas stay with the reference's components
    }
    smilehip_plan *pl = nullptr;
    check(smilehip_plan_create(context(), &c, &pl));
    smilehip_geometry g;
    check(smilehip_plan_geometry(pl, &g));
    const int64_t off[2] = {0, n};
    smilehip_batch *b = nullptr;
    check(smilehip_batch_create(pl, off, 1, &b));
    n_rows = (long)smilehip_batch_total_rows(b);
    n_cols = g.n_out;
    rows.assign((size_t)(n_rows > 0 ? n_rows : 1) * n_cols, 0.0f);
    long fin_rows = 0;
    if (n_rows > 0) {
      void *d_pcm = nullptr, *d_lld = nullptr;
      check(smilehip_alloc(context(), (uint64_t)(n > 0 ? n : 1) * (is16 ? 2 : 4), &d_pcm));
      check(smilehip_alloc(context(), (uint64_t)n_rows * n_cols * 4, &d_lld));
      if (is16) {
        check(smilehip_copy_to_device(context(), d_pcm, s16.data(), (uint64_t)n * 2, nullptr));
        check(smilehip_lld_run(pl, b, (const int16_t *)d_pcm, (float *)d_lld, n_cols, nullptr));
      } else {
        check(smilehip_copy_to_device(context(), d_pcm, pcm.data(), (uint64_t)n * 4, nullptr));
        check(smilehip_lld_run_f32(pl, b, (const float *)d_pcm, (float *)d_lld, n_cols, nullptr));
      }
      check(smilehip_copy_to_host(context(), rows.data(), d_lld, (uint64_t)n_rows * n_cols * 4, nullptr));
      if (big && ps != "is09_emotion") {
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
      if (big) {                                           // the set's functionals level, on the device-resident LLD matrix
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
    {                                                      // the rows' time stamps, as the framer and the window processors give them
      const int64_t n_frames = (!big || ps == "is09_emotion") ? smilehip_num_frames(pl, n) : (int64_t)n_rows - 1;
      times.assign((size_t)(n_rows > 0 ? n_rows : 1), 0.0);
      for (long t = 0; t < n_rows; ++t) times[(size_t)t] = smilehip_row_time(pl, n_frames, t);
      frame_size_sec = c.frame_size_sec;
      frame_period_sec = c.frame_step_sec;
    }
    smilehip_batch_destroy(b);
    smilehip_plan_destroy(pl);
    set_rows(&rows, n_rows, n_cols);
    if (final_level) { FusedLevel &L = levels[plan.out_levels]; L.cols.clear(); for (int i = 0; i < n_cols; ++i) L.cols.push_back(i); }
    if (b_extra.size() == 110) {
      levels["is13_lldB_smo"].extra.assign(b_extra.begin(), b_extra.begin() + 55);
      levels["is13_lldB_smo_de"].extra.assign(b_extra.begin() + 55, b_extra.end());
    }
    if (egm && n_rows > 0) {
      // The rows a cFunctionals instance of the eGeMAPS files summarises are the ones its input level holds at its first
      // end-of-input tick (csrc/smilehip_funcspec.cpp: kEgemapsParts, measured against the binary): with T20 / T60 the frames of
      // the 20 ms / 60 ms framers and P the frames the Viterbi pass had not decided at the end of input -- 20 ms levels T20; the
      // levels behind the Viterbi smoother max(1, T60 - P); the voiced-only levels, which also wait for cPitchJitter, T60 - P, or
      // T60 when P = T60. Those rows are handed out; what the window processors would add afterwards nobody reads.
      const long T20 = fin_rows - 1, T60 = n_rows - 1, P = f0_pending;
      const long behind = T60 - P > 1 ? T60 - P : 1, voiced = P < T60 ? T60 - P : T60;
      set_rows(&fin, 0, 36);
      const std::string g1 = "gemapsv01b";
      levels[g1 + "_loudness_smo"].n_rows = T20;
      levels["egemapsv02_lldSetNoF0AndLoudnessZ_smo"].n_rows = T20;
      levels["egemapsv02_energyRMS"].n_rows = T20;
      levels[g1 + "_lld_single_logF0_smo"].n_rows = behind;
      levels["egemapsv02_lldSetSpectralZ_smo"].n_rows = behind;
      levels["egemapsv02_lldSetNoF0AndLoudnessNz_smo"].n_rows = voiced;
      levels["egemapsv02_lldSetSpectralNz_smo"].n_rows = voiced;
    }
    SMILE_MSG(2, "libsmilehip plugin: fused mode -- %s: %ld rows from %ld samples of the wave level in one batch (%s samples)", plan.describe.c_str(), n_rows,
              (long)n, is16 ? "16-bit" : "float");
    std::vector<float>().swap(pcm);
  }
  void init() {
    if (tried) return;
    tried = true;
    const char *on = getenv("SMILEHIP_PLUGIN_FUSE");
    if (on && !strcmp(on, "0")) return;
    const bool loud = on && *on;
#define FUSE_NOTE(...) do { if (loud) { SMILE_WRN(1, __VA_ARGS__); } else { SMILE_MSG(3, __VA_ARGS__); } } while (0)
    {                                                      // every override registered: the chain's last components must be the ones that hand out rows
      const char *only = getenv("SMILEHIP_PLUGIN_COMPONENTS");
      if (only && *only && strcmp(only, "all")) { FUSE_NOTE("libsmilehip plugin: fused mode needs every override registered -- block-per-tick path"); return; }
    }
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
    if (!plan.preset.empty() && (plan.last_mfcc > 0 || !plan.func_enabled.empty())) {
      FUSE_NOTE("libsmilehip plugin: fused mode: an edited big-set file (lastMfcc / functionalsEnabled) -- block-per-tick path");
      return;
    }
    // The input must end: ONE cWaveSource writes the level every cFramer reads, one channel (a file of one, or monoMixdown).
    {
      std::string src_name, src_level;
      int n_src = 0;
      for (const smilehip_host::ConfInstance &i : cf.inst) {
        if (i.type == "cWaveSource") { ++n_src; src_name = i.name; const std::string *l = i.find("writer.dmLevel"); src_level = l ? *l : ""; }
        else if (i.type.size() > 6 && i.type.compare(i.type.size() - 6, 6, "Source") == 0 && i.type != "cHipLldSource") n_src += 2;
      }
      bool ok = n_src == 1 && !src_level.empty();
      for (const smilehip_host::ConfInstance &i : cf.inst)
        if (ok && i.type == "cFramer") { const std::string *l = i.find("reader.dmLevel"); ok = l && *l == src_level; }
      cWaveSource *ws = ok && g_compman ? dynamic_cast<cWaveSource *>(g_compman->getComponentInstance(src_name.c_str())) : nullptr;
      if (!ws || !(ws->monoMixdown || ws->pcmParam.nChan == 1) || ws->pcmParam.sampleRate <= 0) {
        FUSE_NOTE("libsmilehip plugin: fused mode: the audio does not come from one cWaveSource with one (mixed-down) channel -- block-per-tick path");
        return;
      }
      sample_rate = (double)ws->pcmParam.sampleRate;
    }
    big = !plan.preset.empty();
    // Where the rows of a cepstral chain are handed out. FINAL: the sinks read one level written by a cVectorConcat -> the batch computes
    // the file's whole output level (mean normalisation, regression stages) and cHipVectorConcat writes it; otherwise the static block
    // only: mean normalisation, deltas and concatenation stay with the reference's components (block-per-tick).
    final_level = !big && plan.out_writer_type == "cVectorConcat";
    declare_levels();
    tick_mode = true;
    active = true;
    SMILE_MSG(2, "libsmilehip plugin: fused mode -- %s: the samples of the wave level go through the fused kernels in one batch (SMILEHIP_PLUGIN_FUSE=0: block-per-tick)",
              plan.describe.c_str());
  }
  // A framer override's tick in fused mode: every sample its reader's level holds is taken (and kept, by the first framer that asks);
  // when the source is done and the level is drained, the batch runs.
  eTickResult feed_tick(cDataReader *rd, const void *who) {
    if (ran) return TICK_INACTIVE;
    if (!feeder) feeder = who;
    if (rd->curR < 0) rd->curR = 0;
    const long have = rd->dm->getCurW(rd->level[0]) - rd->curR;
    if (have > 0) {
      cMatrix *mat = rd->getMatrix(rd->curR, have);
      if (!mat || mat->nT != have || mat->N != 1) COMP_ERR("libsmilehip plugin: fused mode: cannot read %ld samples of the wave level at %ld", have, rd->curR);
      if (feeder == who) pcm.insert(pcm.end(), mat->data, mat->data + have);
      rd->curR += have;
      rd->catchupCurR(rd->curR);
      return TICK_SUCCESS;
    }
    if (feeder == who && source_eof) { run(); return TICK_SUCCESS; }
    return TICK_INACTIVE;
  }
  // ---- the hand-out: the chain's last components write their rows straight to their own levels, a block of frames per tick, with
  // the time stamps the framer would have given them (nothing between the framers and them ever sees data).
  bool tick_mode = false;
  bool final_level = false;                               // the rows are the sinks' own level (cHipVectorConcat hands them out)
  std::vector<double> times;                              // frame time stamps of the batch's rows (smilehip_row_time)
  double frame_size_sec = 0.0, frame_period_sec = 0.0;
  // writes the next rows of L to `writer`; next = rows written so far by this component; block: the component's own matrix
  eTickResult tick_write(const FusedLevel &L, cDataWriter *writer, long &next, cMatrix *&block, long blocksize_w) {
    if (!ran) return TICK_INACTIVE;                        // the source is still writing: the batch has not run yet
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
};
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
