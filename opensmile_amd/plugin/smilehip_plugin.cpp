// libsmilehip_plugin.so -- openSMILE-side adapter of libsmilehip (C++).
//
// Dropped into ./plugins/ of a SHARED openSMILE build it is dlopen'ed by
// cComponentManager::registerPlugins (src/core/componentManager.cpp:289-424),
// which calls the C symbol registerPluginComponent below. The components it
// returns carry the BUILT-IN type names (cVectorPreemphasis, cWindower,
// cTransformFFT, cFFTmagphase, cMelspec, cMfcc, cEnergy, cMZcr, cAcf, cPitchACF,
// cDeltaRegression, cContourSmoother, cSpectral, cPlp, cFunctionals, cSpecScale, cPitchShs, cSpecResample, cLpc, cFormantLpc,
// cHarmonics), so their factories replace the
// built-in ones (componentManager.cpp:104-129) while the built-in ConfigTypes --
// every existing option -- stay (configManager.cpp:2818-2827): unmodified
// config files run through the HIP kernels.
//
// Each override derives from the reference's own class: configuration, field
// naming, frequency-axis metadata and the tick logic are inherited; only the
// per-frame operator processVector() is replaced by a call into the C ABI
// (include/smilehip.h). Errors from the C layer become COMP_ERR
// (src/include/core/exceptions.hpp:137) -- nothing throws across the C ABI.
//
// This file is compiled with the HOST g++ against the reference's headers
// (ABI-coupled to libopensmile.so); it contains no HIP code.
#include <core/componentManager.hpp>
#include <core/dataSource.hpp>
#include <core/smileCommon.hpp>
#include <dsp/specResample.hpp>
#include <dsp/specScale.hpp>
#include <dspcore/acf.hpp>
#include <dspcore/contourSmoother.hpp>
#include <dspcore/deltaRegression.hpp>
#include <dspcore/fftmagphase.hpp>
#include <dspcore/transformFft.hpp>
#include <dspcore/vectorPreemphasis.hpp>
#include <dspcore/windower.hpp>
#include <functionals/functionals.hpp>
#include <lld/formantLpc.hpp>
#include <lld/harmonics.hpp>
#include <lld/lpc.hpp>
#include <lld/lsp.hpp>
#include <lld/pitchShs.hpp>
#include <lld/pitchSmootherViterbi.hpp>
// cPitchJitter keeps its second reader, its options and the state it carries from frame to frame private; a tick-level
// override has to use them (the base class's own myTick must still work when an option set is not built). This translation
// unit is compiled with -fno-access-control (plugin/Makefile): the compiler skips access CHECKS, no keyword is redefined and
// the header is the library's, token for token -- same layout, same mangled names.
#include <lld/pitchJitter.hpp>
#include <lldcore/energy.hpp>
#include <lldcore/intensity.hpp>
#include <lldcore/melspec.hpp>
#include <lldcore/mfcc.hpp>
#include <lldcore/mzcr.hpp>
#include <lldcore/pitchACF.hpp>
#include <lldcore/pitchSmoother.hpp>
#include <lldcore/plp.hpp>
#include <lldcore/spectral.hpp>
#include <other/valbasedSelector.hpp>
#include <other/vectorOperation.hpp>
#include <smileutil/smileUtil.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <map>
#include <string>
#include <vector>

#include "../../include/smilehip.h"
#include "../host/smilehip_host.hpp"
#include "../host/conf_plan.hpp"

#define MODULE "smilehipPlugin"

namespace {

// ---------------------------------------------------------------- shared state
smilehip_context *g_ctx = nullptr;
constexpr int kNumOverrides = 28;
long g_frames[kNumOverrides] = {0};
long g_cpu[kNumOverrides] = {0};       // frames an overridden component handed to the reference's own CPU code (option set not built)
const char *const g_names[kNumOverrides] = {"cVectorPreemphasis", "cWindower", "cTransformFFT", "cFFTmagphase", "cMelspec", "cMfcc",
                                            "cEnergy", "cMZcr", "cAcf", "cPitchACF", "cDeltaRegression", "cContourSmoother", "cSpectral", "cPlp", "cFunctionals", "cSpecScale",
                                            "cPitchShs", "cSpecResample", "cLpc", "cFormantLpc", "cHarmonics", "cPitchSmootherViterbi", "cValbasedSelector", "cPitchJitter",
                                            "cIntensity", "cLsp", "cPitchSmoother", "cVectorOperation"};

// An override whose option set the HIP path does not cover runs the reference's own code -- never silently: the instance
// says so once (level-1 warning in the reference's log) and every such frame is counted (trace line "<type>.cpu <n>").
// Round 3: that is opt-in. By default an option set that is not built is an ERROR of the component (COMP_ERR, like any
// configuration the reference cannot run) -- "the plugin is loaded" then means "the frames were computed on the GPU";
// SMILEHIP_PLUGIN_ALLOW_CPU=1 brings back the logged and counted fall-through.
inline bool allow_cpu() {
  static const int v = [] { const char *e = getenv("SMILEHIP_PLUGIN_ALLOW_CPU"); return (e && e[0] == '1') ? 1 : 0; }();
  return v != 0;
}
#define HIP_FALLTHROUGH(idx, why)                                                                                  \
  do {                                                                                                             \
    if (!allow_cpu())                                                                                              \
      COMP_ERR("libsmilehip plugin: %s (set SMILEHIP_PLUGIN_ALLOW_CPU=1 to run this instance on the reference's CPU code)", why); \
    if (!cpu_warned_) {                                                                                            \
      SMILE_IWRN(1, "libsmilehip plugin: %s -- this instance runs the reference's CPU code", why);                 \
      cpu_warned_ = true;                                                                                          \
    }                                                                                                              \
    g_cpu[idx]++;                                                                                                  \
  } while (0)

smilehip_context *context() {
  if (!g_ctx) {
    const char *dev = getenv("SMILEHIP_DEVICE");
    if (smilehip_init(dev ? atoi(dev) : 0, &g_ctx) != SMILEHIP_OK)
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
  }
  return g_ctx;
}

// device scratch for one frame in / one frame out
struct FrameIO {
  float *d_in = nullptr, *d_out = nullptr;
  long cap_in = 0, cap_out = 0;
  void ensure(long n_in, long n_out) {
    if (n_in > cap_in) {
      if (d_in) smilehip_free(context(), d_in);
      if (smilehip_alloc(context(), sizeof(float) * (uint64_t)n_in, (void **)&d_in)) COMP_ERR("libsmilehip: %s", smilehip_last_error());
      cap_in = n_in;
    }
    if (n_out > cap_out) {
      if (d_out) smilehip_free(context(), d_out);
      if (smilehip_alloc(context(), sizeof(float) * (uint64_t)n_out, (void **)&d_out)) COMP_ERR("libsmilehip: %s", smilehip_last_error());
      cap_out = n_out;
    }
  }
  void up(const FLOAT_DMEM *src, long n) {
    if (smilehip_copy_to_device(context(), d_in, src, sizeof(float) * (uint64_t)n, nullptr)) COMP_ERR("libsmilehip: %s", smilehip_last_error());
  }
  void down(FLOAT_DMEM *dst, long n) {
    if (smilehip_copy_to_host(context(), dst, d_out, sizeof(float) * (uint64_t)n, nullptr) ||
        smilehip_stream_synchronize(context(), nullptr))
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
  }
  ~FrameIO() {
    if (g_ctx) {
      if (d_in) smilehip_free(g_ctx, d_in);
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
    }
    big = true;
    active = true;
    SMILE_MSG(2, "libsmilehip plugin: fused mode -- %s: %ld rows of '%s' in one batch", plan.describe.c_str(), n_rows, plan.wave_file.c_str());
    return true;
  }
  void init() {
    if (tried) return;
    tried = true;
    const char *on = getenv("SMILEHIP_PLUGIN_FUSE");
    if (!on || !*on || !strcmp(on, "0")) return;
    std::vector<std::string> args;
    if (FILE *f = fopen("/proc/self/cmdline", "rb")) {
      std::string cur;
      int ch;
      while ((ch = fgetc(f)) != EOF) { if (ch == 0) { args.push_back(cur); cur.clear(); } else cur += (char)ch; }
      if (!cur.empty()) args.push_back(cur);
      fclose(f);
    }
    std::string conf;
    std::map<std::string, std::string> cl;
    for (size_t i = 1; i < args.size(); ++i) {
      if (args[i].size() < 2 || args[i][0] != '-') continue;
      const bool has_val = i + 1 < args.size() && (args[i + 1].empty() || args[i + 1][0] != '-' || isdigit((unsigned char)args[i + 1][1]));
      const std::string key = args[i].substr(1), val = has_val ? args[i + 1] : "1";
      if (key == "C" || key == "configfile") conf = val; else cl[key] = val;
      if (has_val) ++i;
    }
    std::string err;
    smilehip_host::ConfFile cf;
    if (conf.empty() || !smilehip_host::conf_parse(conf, cl, cf, err)) {
      SMILE_WRN(1, "libsmilehip plugin: SMILEHIP_PLUGIN_FUSE: cannot read the configuration file (%s) -- per-component path", err.c_str());
      return;
    }
    if (!smilehip_host::conf_to_plan(cf, plan, err)) {
      SMILE_WRN(1, "libsmilehip plugin: SMILEHIP_PLUGIN_FUSE: %s -- per-component path", err.c_str());
      return;
    }
    smilehip_host::WaveInfo wi;
    std::vector<unsigned char> raw;
    if (!smilehip_host::read_wave_file(plan.wave_file, wi, raw, err) || wi.sample_type != 1 || wi.n_bps != 2 || wi.n_chan != 1) {
      SMILE_WRN(1, "libsmilehip plugin: SMILEHIP_PLUGIN_FUSE: '%s' is not a 16-bit mono PCM file (%s) -- per-component path", plan.wave_file.c_str(), err.c_str());
      return;
    }
    if (!plan.preset.empty()) {
      // the big sets fuse only with EVERY override registered: the final smoother / delta instances must be the ones that hand out rows
      const char *only = getenv("SMILEHIP_PLUGIN_COMPONENTS");
      const bool all = !only || !*only || !strcmp(only, "all");
      if (!all || plan.last_mfcc > 0 || !plan.func_enabled.empty() || !init_big(wi, raw))
        SMILE_WRN(1, "libsmilehip plugin: SMILEHIP_PLUGIN_FUSE: this big-set file does not fuse inside the reference process -- per-component path");
      return;
    }
    smilehip_lld_config c = plan.cfg;
    c.sample_rate = (double)wi.sample_rate;
    c.n_delta = 0;                                        // the static block is all the chain components hand on;
    c.cms = 0;                                            // mean normalisation and deltas stay with the reference's components
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
    smilehip_batch_destroy(b);
    smilehip_plan_destroy(pl);
    for (const auto &kv : plan.static_levels) {
      FusedLevel L;
      L.M = &rows; L.ld = n_cols; L.n_rows = n_rows; L.cols = kv.second;
      levels[kv.first] = L;
    }
    active = true;
    SMILE_MSG(2, "libsmilehip plugin: fused mode -- %s: %ld frames of '%s' in one batch", plan.describe.c_str(), n_rows, plan.wave_file.c_str());
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

// ---------------------------------------------------------------- overrides
// a small device buffer of raw bytes (results that are not float frames)
struct DevBytes {
  void *d = nullptr;
  uint64_t cap = 0;
  void *ensure(uint64_t bytes) {
    if (bytes > cap) {
      if (d) smilehip_free(context(), d);
      if (smilehip_alloc(context(), bytes, &d)) COMP_ERR("libsmilehip: %s", smilehip_last_error());
      cap = bytes;
    }
    return d;
  }
  void down(void *h, uint64_t bytes) {
    if (smilehip_copy_to_host(context(), h, d, bytes, nullptr) || smilehip_stream_synchronize(context(), nullptr))
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
  }
  ~DevBytes() { if (g_ctx && d) smilehip_free(g_ctx, d); }
};

// R2  cVectorPreemphasis::processVector  (src/dspcore/vectorPreemphasis.cpp:89-107)
class cHipVectorPreemphasis : public cVectorPreemphasis {
  int fused_ = -1;
  FrameIO io_;
  bool cpu_warned_ = false;
  float k_ = 0.f;
  int de_ = 0;
  bool ready_ = false;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (fused_ < 0) { g_fused.init(); fused_ = g_fused.stage_level(getStr("writer.dmLevel")) ? 1 : 0; }
    if (fused_) { for (long k = 0; k < Ndst; ++k) dst[k] = 0; g_fused_stage++; return 1; }   // fused mode: cMfcc / cPlp hand out the batch's rows
    if (!ready_) {
      double f = isSet("f") ? getDouble("f") : -1.0;
      k_ = (FLOAT_DMEM)getDouble("k");
      if (f >= 0.0) k_ = (FLOAT_DMEM)exp(-2.0 * M_PI * f * getBasePeriod());   // vectorPreemphasis.cpp:78-86
      de_ = getInt("de");
      ready_ = true;
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_preemphasis_frames(context(), io_.d_in, Nsrc, io_.d_out, Ndst, 1, Ndst, k_, de_, nullptr));
    io_.down(dst, Ndst);
    g_frames[0]++;
    return 1;
  }
 public:
  explicit cHipVectorPreemphasis(const char *n) : cVectorPreemphasis(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipVectorPreemphasis(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// one plan per input field configuration
template <int NPLANS = 8>
struct PlanSet {
  smilehip_plan *p[NPLANS];
  PlanSet() { for (auto &x : p) x = nullptr; }
  ~PlanSet() { for (auto &x : p) if (x) smilehip_plan_destroy(x); }
  smilehip_plan *&at(int i) {
    if (i < 0 || i >= NPLANS) COMP_ERR("libsmilehip plugin: more than %d differently sized fields", NPLANS);
    return p[i];
  }
};

// R3  cWindower::processVector  (src/dspcore/windower.cpp:221-229)
class cHipWindower : public cWindower {
  int fused_ = -1;
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (fused_ < 0) { g_fused.init(); fused_ = g_fused.stage_level(getStr("writer.dmLevel")) ? 1 : 0; }
    if (fused_) { for (long k = 0; k < Ndst; ++k) dst[k] = 0; g_fused_stage++; return 1; }   // fused mode: cMfcc / cPlp hand out the batch's rows
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      if (getDouble("fade") > 0.0 || getInt("squareRoot") || getDouble("xshift") != 0.0)
        COMP_ERR("libsmilehip plugin: cWindower options fade/squareRoot/xshift are not supported on the HIP path");
      smilehip_lld_config c = base_config(Nsrc, SMILEHIP_STAGE_WINDOW);
      c.win_func = winfunc_id(getStr("winFunc"));
      if (c.win_func < 0) COMP_ERR("libsmilehip plugin: window function '%s' not supported on the HIP path", getStr("winFunc"));
      c.win_sigma = getDouble("sigma");
      c.win_gain = getDouble("gain");
      c.win_offset = getDouble("offset");
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_window_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[1]++;
    return 1;
  }
 public:
  explicit cHipWindower(const char *n) : cWindower(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipWindower(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R4  cTransformFFT::processVector, forward  (src/dspcore/transformFft.cpp:165-223)
class cHipTransformFFT : public cTransformFFT {
  int fused_ = -1;
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (fused_ < 0) { g_fused.init(); fused_ = g_fused.stage_level(getStr("writer.dmLevel")) ? 1 : 0; }
    if (fused_) { for (long k = 0; k < Ndst; ++k) dst[k] = 0; g_fused_stage++; return 1; }   // fused mode: cMfcc / cPlp hand out the batch's rows
    if (getInt("inverse")) {                             // rdft(N, -1) and the 2 / N scaling (transformFft.cpp:196-216)
      if (Nsrc != Ndst || Ndst < 64 || Ndst > 8192 || (Ndst & (Ndst - 1))) {
        HIP_FALLTHROUGH(2, "cTransformFFT inverse = 1: only whole packed spectra of 64 .. 8192 values are built");
        return cTransformFFT::processVector(src, dst, Nsrc, Ndst, idxi);
      }
      smilehip_plan *&pli = plans_.at(getFconf(idxi));
      if (!pli) {
        smilehip_lld_config c = base_config(Nsrc, SMILEHIP_STAGE_FFT);
        check(smilehip_plan_create(context(), &c, &pli));
      }
      io_.ensure(Nsrc, Ndst);
      io_.up(src, Nsrc);
      check(smilehip_irfft_frames(pli, io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
      io_.down(dst, Ndst);
      g_frames[2]++;
      return 1;
    }
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      smilehip_lld_config c = base_config(Nsrc, SMILEHIP_STAGE_FFT);
      c.zero_pad_symmetric = getInt("zeroPadSymmetric");
      check(smilehip_plan_create(context(), &c, &pl));
      smilehip_geometry g;
      check(smilehip_plan_geometry(pl, &g));
      if (g.fft_size != Ndst) COMP_ERR("libsmilehip plugin: FFT size mismatch (%ld vs %ld)", (long)g.fft_size, Ndst);
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_rfft_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[2]++;
    return 1;
  }
 public:
  explicit cHipTransformFFT(const char *n) : cTransformFFT(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipTransformFFT(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R5  cFFTmagphase::processVector, magnitude branch  (src/dspcore/fftmagphase.cpp:215-221)
class cHipFFTmagphase : public cFFTmagphase {
  int fused_ = -1;
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
  int plain_ = -1, modes_ = 0;
  bool other_ok_ = false;
  float dbp_norm_ = 0.0f, min_dbp_ = 0.0f;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (fused_ < 0) { g_fused.init(); fused_ = g_fused.stage_level(getStr("writer.dmLevel")) ? 1 : 0; }
    if (fused_) { for (long k = 0; k < Ndst; ++k) dst[k] = 0; g_fused_stage++; return 1; }   // fused mode: cMfcc / cPlp hand out the batch's rows
    if (plain_ < 0) {
      plain_ = (!getInt("inverse") && getInt("magnitude") && !getInt("phase") && !getInt("normalise") &&
                !getInt("power") && !getInt("dBpsd")) ? 1 : 0;
      // every other output mode of :215-287 but the two the reference itself cannot produce properly: inverse = 1 (mag / phase ->
      // complex) and magnitude + phase as separate fields (:268-276, "this check is wrong" upstream)
      const int mag = getInt("magnitude"), ph = getInt("phase");
      modes_ = (mag ? SMILEHIP_MAGPHASE_MAGNITUDE : 0) | (ph ? SMILEHIP_MAGPHASE_PHASE : 0) |
               (getInt("normalise") ? SMILEHIP_MAGPHASE_NORMALISE : 0) | (getInt("power") ? SMILEHIP_MAGPHASE_POWER : 0) |
               (getInt("dBpsd") ? SMILEHIP_MAGPHASE_DBPSD : 0);
      other_ok_ = !getInt("inverse") && (mag || ph) && !(mag && ph && !getInt("joinMagphase"));
      dbp_norm_ = dBpnorm;                               // the members cFFTmagphase::myFetchConfig filled (:93-98: mindBp >= dBpnorm - 120 enforced)
      min_dbp_ = mindBp;
    }
    if (!plain_ && other_ok_) {
      const long K = Nsrc / 2 + 1;
      const long n_out = ((modes_ & 1) ? K : 0) + ((modes_ & 2) ? K : 0);
      if (n_out <= Ndst && Nsrc >= 4 && !(Nsrc & 1)) {
        io_.ensure(Nsrc, Ndst);
        io_.up(src, Nsrc);
        check(smilehip_fftmagphase_frames(context(), io_.d_in, Nsrc, Nsrc, modes_, dbp_norm_, min_dbp_, io_.d_out, Ndst, 1, nullptr));
        io_.down(dst, n_out);
        g_frames[3]++;
        return 1;
      }
    }
    if (!plain_) { HIP_FALLTHROUGH(3, "cFFTmagphase: inverse = 1 and magnitude + phase as separate fields are not built"); return cFFTmagphase::processVector(src, dst, Nsrc, Ndst, idxi); }
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      smilehip_lld_config c = base_config(Nsrc, SMILEHIP_STAGE_FFT);
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_fftmag_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[3]++;
    return 1;
  }
 public:
  explicit cHipFFTmagphase(const char *n) : cFFTmagphase(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipFFTmagphase(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R6  cMelspec::processVector  (src/lldcore/melspec.cpp:519-570)
class cHipMelspec : public cMelspec {
  int fused_ = -1;
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
  int plain_ = -1;
  DevBytes tab_coef_[8], tab_map_[8];
  bool tab_ready_[8] = {false, false, false, false, false, false, false, false};
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (fused_ < 0) { g_fused.init(); fused_ = g_fused.stage_level(getStr("writer.dmLevel")) ? 1 : 0; }
    if (fused_) { for (long k = 0; k < Ndst; ++k) dst[k] = 0; g_fused_stage++; return 1; }   // fused mode: cMfcc / cPlp hand out the batch's rows
    if (plain_ < 0) {
      const char *bw = getStr("bwMethod");
      const char *sc = getStr("specScale");
      const bool mel = getInt("htkcompatible") || (sc && !strcasecmp(sc, "mel"));
      plain_ = (!getInt("inverse") && mel && bw && !strncasecmp(bw, "lr", 2)) ? 1 : 0;
    }
    if (!plain_ && !getInt("inverse")) {
      // any other bank cMelspec::computeFilters built (other spectral scales, bwMethod, HFCC, custom bandwidth): the component's own
      // tables go to the device once, the frames through the table-driven operator
      const int fc = getFconf(idxi);
      if (fc >= 0 && fc < 8 && filterCoeffs_ && chanMap_ && filterCoeffs_[fc] && chanMap_[fc] && Ndst == nBands_ && Nsrc <= 8193) {
        const bool dense = hfcc_ || customBandwidth_;
        if (!tab_ready_[fc]) {
          const size_t nc = dense ? (size_t)nBands_ * (size_t)Nsrc : (size_t)Nsrc, nm = dense ? (size_t)2 * nBands_ : (size_t)Nsrc;
          std::vector<float> cf(filterCoeffs_[fc], filterCoeffs_[fc] + nc);
          std::vector<int32_t> cm(nm);
          for (size_t i = 0; i < nm; ++i) cm[i] = (int32_t)chanMap_[fc][i];
          void *d_c = tab_coef_[fc].ensure(sizeof(float) * nc);
          void *d_m = tab_map_[fc].ensure(sizeof(int32_t) * nm);
          if (smilehip_copy_to_device(context(), d_c, cf.data(), sizeof(float) * nc, nullptr) ||
              smilehip_copy_to_device(context(), d_m, cm.data(), sizeof(int32_t) * nm, nullptr))
            COMP_ERR("libsmilehip: %s", smilehip_last_error());
          tab_ready_[fc] = true;
        }
        const float scale = htkcompatible_ ? (usePower_ ? (FLOAT_DMEM)(32767.0 * 32767.0) : (FLOAT_DMEM)32767.0) : 1.0f;
        io_.ensure(Nsrc, Ndst);
        io_.up(src, Nsrc);
        check(smilehip_melspec_table_frames(context(), io_.d_in, Nsrc, Nsrc, nBands_, dense ? 1 : 0, (const float *)tab_coef_[fc].d,
                                            (const int32_t *)tab_map_[fc].d, (int32_t)nLoF_[fc], (int32_t)nHiF_[fc], usePower_, scale,
                                            io_.d_out, Ndst, 1, nullptr));
        io_.down(dst, Ndst);
        g_frames[4]++;
        return 1;
      }
    }
    if (!plain_) { HIP_FALLTHROUGH(4, "cMelspec: inverse = 1 is not built"); return cMelspec::processVector(src, dst, Nsrc, Ndst, idxi); }
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      // frame size of the input spectrum, cMelspec::configureField (melspec.cpp:150-173)
      const sDmLevelConfig *lc = reader_->getLevelConfig();
      double fss = isSet("overrideFrameSizeSec") ? getDouble("overrideFrameSizeSec")
                                                 : (lc->frameSizeSec > 0.0 ? lc->frameSizeSec : lc->lastFrameSizeSec);
      smilehip_lld_config c = base_config((Nsrc - 1) * 2, SMILEHIP_STAGE_MEL);
      c.force_fft_frame_size_sec = fss;
      c.n_bands = getInt("nBands");
      c.lofreq = (FLOAT_DMEM)getDouble("lofreq");
      c.hifreq = (FLOAT_DMEM)getDouble("hifreq");
      c.use_power = getInt("usePower");
      c.mel_htk_compatible = getInt("htkcompatible");
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_melspec_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[4]++;
    return 1;
  }
 public:
  explicit cHipMelspec(const char *n) : cMelspec(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipMelspec(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R7  cMfcc::processVector, forward  (src/lldcore/mfcc.cpp:239-273)
class cHipMfcc : public cMfcc {
  int fused_ = -1;
  const FusedLevel *fcols_ = nullptr;
  long fframe_ = 0, fnext_ = 0;
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (fused_ < 0) { g_fused.init(); fcols_ = g_fused.static_level(getStr("writer.dmLevel")); fused_ = fcols_ ? 1 : 0; }
    if (fused_) { if (idxi == 0) fframe_ = fnext_++; g_fused.copy(*fcols_, fframe_, dst, Ndst); return 1; }   // fused mode: rows of the whole-file batch
    if (getInt("inverse") || !getInt("doLog")) { HIP_FALLTHROUGH(5, "cMfcc: inverse = 1 / doLog = 0 are not built"); return cMfcc::processVector(src, dst, Nsrc, Ndst, idxi); }
    smilehip_plan *&pl = plans_.at(getFconf(idxi));
    if (!pl) {
      smilehip_lld_config c = base_config(512, SMILEHIP_STAGE_MFCC);
      c.n_bands = (int)Nsrc;
      c.first_mfcc = getInt("firstMfcc");
      c.last_mfcc = getInt("lastMfcc");
      if (!isSet("lastMfcc") && isSet("nMfcc")) c.last_mfcc = c.first_mfcc + getInt("nMfcc") - 1;   // mfcc.cpp:77-82
      c.cep_lifter = (FLOAT_DMEM)getDouble("cepLifter");
      c.mfcc_htk_compatible = getInt("htkcompatible");
      c.melfloor = (FLOAT_DMEM)getDouble("melfloor");
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_mfcc_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[5]++;
    return 1;
  }
 public:
  explicit cHipMfcc(const char *n) : cMfcc(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipMfcc(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R12  cEnergy::processVector  (src/lldcore/energy.cpp:152-185): the double-accumulated sum of
// squares comes from the device, the rms / squared / log expressions are the reference's
class cHipEnergy : public cEnergy {
  int fused_ = -1;
  const FusedLevel *fcols_ = nullptr;
  long fframe_ = 0, fnext_ = 0;
  FrameIO io_;
  bool cpu_warned_ = false;
  DevBytes res_;
  int htk_ = 0, erms_ = 0, e2_ = 0, elog_ = 0;
  FLOAT_DMEM sRms_ = 1, sLog_ = 1, sSq_ = 1, bLog_ = 0, bRms_ = 0, bSq_ = 0;
  bool ready_ = false;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (fused_ < 0) { g_fused.init(); fcols_ = g_fused.static_level(getStr("writer.dmLevel")); fused_ = fcols_ ? 1 : 0; }
    if (fused_) { if (idxi == 0) fframe_ = fnext_++; g_fused.copy(*fcols_, fframe_, dst, Ndst); return 1; }   // fused mode: rows of the whole-file batch
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
    double *d_d = (double *)res_.ensure(sizeof(double));
    check(smilehip_sumsq_frames(context(), io_.d_in, Nsrc, Nsrc, 1, d_d, nullptr));
    double d = 0.0;
    res_.down(&d, sizeof(double));
    int n = 0;
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
    g_frames[6]++;
    return n;
  }
 public:
  explicit cHipEnergy(const char *n) : cEnergy(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipEnergy(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R12  cMZcr::processVector, zero-crossing rate  (src/lldcore/mzcr.cpp:109-150)
class cHipMZcr : public cMZcr {
  FrameIO io_;
  bool cpu_warned_ = false;
  DevBytes res_;
  int plain_ = -1, flags_ = 0;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE((int)Ndst);
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
      check(smilehip_mzcr_frames(context(), io_.d_in, Nsrc, Nsrc, 1, flags_, io_.d_out, n_out, nullptr));
      io_.down(dst, n_out);
      g_frames[7]++;
      return n_out;
    }
    if (!plain_) { HIP_FALLTHROUGH(7, "cMZcr: no output selected, or a frame longer than 32768 samples"); return cMZcr::processVector(src, dst, Nsrc, Ndst, idxi); }
    io_.ensure(Nsrc, 1);
    io_.up(src, Nsrc);
    int32_t *d_c = (int32_t *)res_.ensure(sizeof(int32_t));
    check(smilehip_zcr_count_frames(context(), io_.d_in, Nsrc, Nsrc, 1, d_c, nullptr));
    int32_t c = 0;
    res_.down(&c, sizeof(c));
    FLOAT_DMEM nzc = (FLOAT_DMEM)c;
    nzc /= (FLOAT_DMEM)Nsrc;
    dst[0] = nzc;
    g_frames[7]++;
    return 1;
  }
 public:
  explicit cHipMZcr(const char *n) : cMZcr(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipMZcr(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R9  cAcf::processVector, forward path  (src/dspcore/acf.cpp:249-349)
class cHipAcf : public cAcf {
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
  int plain_ = -1, use_power_ = 0, cepstrum_ = 0, norm_ = 0, abs_ceps_ = 0;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE(1);
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
    check(smilehip_acf_frames(pl, io_.d_in, Nsrc, io_.d_out, Ndst, Ndst, 1, use_power_, cepstrum_, norm_, abs_ceps_, nullptr));
    io_.down(dst, Ndst);
    g_frames[8]++;
    return 1;
  }
 public:
  explicit cHipAcf(const char *n) : cAcf(n) {}
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
class cHipPitchACF : public cPitchACF {
  FrameIO io_;
  bool cpu_warned_ = false;
  DevBytes res_;
  bool state_ready_ = false;
  int plain_ = -1, voiceProb_ = 0, F0_ = 0, F0raw_ = 0, F0env_ = 0, HNR_ = 0, HNRdB_ = 0, linHNR_ = 0, voiceQual_ = 0;
  double maxPitch_ = 0.0, voicingCutoff_ = 0.0;
  float fsSec_ = -1.0f;
  // device result block: voicing | peak index | ACF zero-crossing rate | F0, F0raw, F0env, 0 | contour state (8 words)
  struct Result { double voicing; int32_t idx; int32_t pad; double acfZcr; float f0[4]; float state[8]; };
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE((int)Ndst);
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
    unsigned char *r = (unsigned char *)res_.ensure(sizeof(Result));
    if (!state_ready_) {                                 // a stream starts with an all-zero contour
      const Result zero = {};
      check(smilehip_copy_to_device(context(), r, &zero, sizeof(Result), nullptr));
      state_ready_ = true;
    }
    const double Tsamp = fsSec_ / (double)Nsrc;
    check(smilehip_pitchacf_frames(context(), io_.d_in, Nsrc, N, 1, (double)fsSec_, maxPitch_, (double *)(r + offsetof(Result, voicing)),
                                   (int32_t *)(r + offsetof(Result, idx)), nullptr));
    if (voiceQual_) check(smilehip_pitchacf_zcr_frames(context(), io_.d_in, Nsrc, N, 1, (double)fsSec_, maxPitch_, (double *)(r + offsetof(Result, acfZcr)), nullptr));
    const bool contour = F0_ || F0env_ || F0raw_ || voiceQual_;
    if (contour)
      check(smilehip_pitchacf_contour_step(context(), (const double *)(r + offsetof(Result, voicing)), (const int32_t *)(r + offsetof(Result, idx)),
                                           Tsamp, voicingCutoff_, (float *)(r + offsetof(Result, state)), (float *)(r + offsetof(Result, f0)), nullptr));
    Result h = {};
    res_.down(&h, offsetof(Result, state));
    const long peak = h.idx;
    int n = 0;
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
    g_frames[9]++;
    return n;
  }
 public:
  explicit cHipPitchACF(const char *n) : cPitchACF(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipPitchACF(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R13  cWindowProcessor::processBuffer of cDeltaRegression (src/dspcore/deltaRegression.cpp:113-170) and
// cContourSmoother (src/dspcore/contourSmoother.cpp:85-118): one row of the block the window
// processor's tick hands over, valid on [-pre, nT+post)
// big-set fused mode for the window processors (cWindowProcessor::myTick calls processBuffer once per element row of a block,
// rows 0 .. N-1 in order, windowProcessor.cpp:188-200): the n-th call of a block is element n, the blocks follow each other in time
struct FusedRows {
  const FusedLevel *lvl = nullptr;
  int tried = 0;
  long call = 0, t_base = 0;
  // true: `out` has been filled (with the level's rows, or zeros for a level nobody downstream of the fused ones reads)
  bool serve(const char *writer_level, long n_elems, cMatrix *out) {
    if (!tried) { tried = 1; g_fused.init(); if (g_fused.big) lvl = g_fused.static_level(writer_level); }
    if (!g_fused.big || !lvl) return false;
    const long N = n_elems > 0 ? n_elems : 1;
    const int e = (int)(call % N);
    for (long t = 0; t < out->nT; ++t) out->data[t] = g_fused.at(*lvl, t_base + t, e);
    if (e == N - 1) t_base += out->nT;
    ++call;
    if (!lvl->cols.empty()) g_fused.served += out->nT; else g_fused_stage++;
    return true;
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
  float *d_norm = nullptr;                                 // kind 3: the instance's carried divisor (one device float)
};

class cHipDeltaRegression : public cDeltaRegression {
  RowIO row_;
  FusedRows frows_;
  bool cpu_warned_ = false;
  int plain_ = -1, W_ = 0, segs_ = 0;
  DevBytes norm_;
 protected:
  int processBuffer(cMatrix *in, cMatrix *out, int pre, int post) override {
    if (plain_ < 0) {
      W_ = getInt("deltawin");
      segs_ = getInt("onlyInSegments") ? 1 : 0;
      plain_ = (W_ > 0 && !getInt("relativeDelta") && !getInt("halfWaveRect") && !getInt("absOutput")) ? 1 : 0;
      if (plain_ && segs_) {                               // the norm member the onlyInSegments branch keeps adding to (:77-79, :129)
        float n0 = 0.0f;
        for (int i = 1; i <= W_; i++) n0 += (float)i * (float)i;
        n0 *= 2.0f;
        row_.d_norm = (float *)norm_.ensure(sizeof(float));
        check(smilehip_copy_to_device(context(), row_.d_norm, &n0, sizeof(float), nullptr));
      }
    }
    if (frows_.serve(getStr("writer.dmLevel"), Ni, out)) return 1;                    // big-set fused mode: rows of the whole-file batch
    if (g_fused.active) return cDeltaRegression::processBuffer(in, out, pre, post);   // fused mode: the rows are already on the host, the reference's own regression is cheaper than a device round trip per block
    if (!plain_ || pre < W_ || post < W_) { HIP_FALLTHROUGH(10, "cDeltaRegression: relativeDelta / absOutput / halfWaveRect are not built"); return cDeltaRegression::processBuffer(in, out, pre, post); }
    row_.run(in, out, pre, post, segs_ ? 3 : 0, W_);
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
  FusedRows frows_;
  bool cpu_warned_ = false;
  int plain_ = -1, W_ = 0, nz_ = 0;
 protected:
  int processBuffer(cMatrix *in, cMatrix *out, int pre, int post) override {
    if (plain_ < 0) {
      const int w = smaWin;                              // the member: myFetchConfig has made an even value odd (contourSmoother.cpp:64-67)
      W_ = w / 2;
      plain_ = ((w & 1) && W_ >= 1) ? 1 : 0;
      nz_ = getInt("noZeroSma") ? 1 : 0;
    }
    if (frows_.serve(getStr("writer.dmLevel"), Ni, out)) return 1;                    // big-set fused mode: rows of the whole-file batch
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

// R11  cSpectral::processVector with ComParE_2016's option set  (src/lldcore/spectral.cpp:586-1560)
class cHipSpectral : public cSpectral {
  FrameIO io_;
  bool cpu_warned_ = false;
  PlanSet<> plans_;
  DevBytes prev_[8];
  bool seen_[8] = {false, false, false, false, false, false, false, false};
  int plain_ = -1, gemaps_ = -1;
  smilehip_plan *gm_plan_ = nullptr;
  int band_lo_[2] = {250, 1000}, band_hi_[2] = {650, 4000};
  bool sel_[3] = {true, true, true};
  // the general option set (smilehip_spectral_op_*: any bands / rollOff points, every descriptor optional): one operator per field
  int general_ = -1, gen_n_out_ = 0;
  smilehip_spectral_opts gen_opts_;
  smilehip_spectral_op *gen_op_[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool array_is(const char *name, int n, const char *const *vals) {
    if (getArraySize(name) != n) return false;
    for (int i = 0; i < n; ++i) {
      const char *v = getStr_f(myvprint("%s[%i]", name, i));
      if (!v || strcmp(v, vals[i]) != 0) return false;
    }
    return true;
  }
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE((int)Ndst);
    if (plain_ < 0) {
      bool ok = getArraySize("bands") == 2 && getArraySize("rollOff") == 4 && getArraySize("slopes") <= 0;
      for (int b = 0; ok && b < 2; ++b) {                  // bands[b] = "lo-hi" in Hz, integers (spectral.cpp:163-190)
        const char *v = getStr_f(myvprint("bands[%i]", b));
        int lo = -1, hi = -1, used = 0;
        ok = v && sscanf(v, "%d-%d%n", &lo, &hi, &used) == 2 && v[used] == 0 && lo >= 0 && hi > lo;
        band_lo_[b] = lo; band_hi_[b] = hi;
      }
      static const double ro[4] = {0.25, 0.50, 0.75, 0.90};
      for (int i = 0; ok && i < 4; ++i) ok = getDouble_f(myvprint("rollOff[%i]", i)) == ro[i];
      // the optional outputs of the fifteen (their values do not enter the others: the centroid is computed whenever a moment or the
      // slope is on, spectral.cpp:1262): column 7, 13, 14 of the device row
      sel_[0] = getInt("centroid") != 0; sel_[1] = getInt("sharpness") != 0; sel_[2] = getInt("harmonicity") != 0;
      static const char *const on[] = {"squareInput", "flux", "entropy", "variance", "skewness", "kurtosis", "slope", "oldSlopeScale"};
      static const char *const off[] = {"normBandEnergies", "specDiff", "specPosDiff", "fluxCentroid", "fluxAtFluxCentroid", "maxPos",
                                        "minPos", "standardDeviation", "alphaRatio", "hammarbergIndex", "tonality", "flatness",
                                        "logFlatness", "buggyRollOff", "useLogSpectrum"};
      for (const char *o : on) ok = ok && getInt(o) != 0;
      for (const char *o : off) ok = ok && getInt(o) == 0;
      const char *fr = getStr("freqRange");
      ok = ok && fr && !strcmp(fr, "0-0");
      plain_ = ok ? 1 : 0;
    }
    const int fc = getFconf(idxi);
    if (gemaps_ < 0) {
      // the two GeMAPS option sets (GeMAPSv01b_core.lld.conf.inc [gemapsv01b_logSpectral], eGeMAPSv02_core.lld.conf.inc
      // [egemapsv02_logSpectral_flux]): log-spectrum slopes 0-500 / 500-1500 + alpha ratio + Hammarberg index, or flux alone
      static const char *const slopes[2] = {"0-500", "500-1500"};
      static const char *const off[] = {"specDiff", "specPosDiff", "fluxCentroid", "fluxAtFluxCentroid", "centroid", "maxPos", "minPos",
                                        "entropy", "standardDeviation", "variance", "skewness", "kurtosis", "slope", "sharpness",
                                        "tonality", "harmonicity", "flatness", "logFlatness", "buggyRollOff", "oldSlopeScale"};
      bool ok = getInt("squareInput") == 1 && getInt("useLogSpectrum") == 1 && getInt("normBandEnergies") == 1 &&
                getArraySize("bands") <= 0 && getArraySize("rollOff") <= 0 && getDouble("specFloor") == 0.0000001;
      for (const char *o : off) ok = ok && getInt(o) == 0;
      const char *fr = getStr("freqRange");
      ok = ok && fr && !strcmp(fr, "0-5000");
      gemaps_ = 0;
      if (ok && array_is("slopes", 2, slopes) && getInt("alphaRatio") == 1 && getInt("hammarbergIndex") == 1 && getInt("flux") == 0)
        gemaps_ = 1;                                     // 4 outputs
      else if (ok && getArraySize("slopes") <= 0 && getInt("alphaRatio") == 0 && getInt("hammarbergIndex") == 0 && getInt("flux") == 1)
        gemaps_ = 2;                                     // 1 output
    }
    if (gemaps_ > 0 && (Nsrc == 129 || Nsrc == 257 || Nsrc == 513) && Ndst == (gemaps_ == 1 ? 4 : 1) && fc >= 0 && fc < 8) {
      if (!gm_plan_) {
        const sDmLevelConfig *lc = reader_->getLevelConfig();
        smilehip_lld_config c;
        smilehip_config_egemapsv02(&c);
        c.sample_rate = std::round(2.0 * (double)(Nsrc - 1) / lc->frameSizeSec);       // the spectrum level's frameSizeSec = Nfft / rate
        if (!(c.sample_rate >= 7999.0 && c.sample_rate <= 48001.0))
          COMP_ERR("libsmilehip plugin: cSpectral (GeMAPS options): the HIP path is built for 20 ms frames at 8 .. 48 kHz (this level: %ld bins, %g s)", Nsrc, lc->frameSizeSec);
        check(smilehip_plan_create(context(), &c, &gm_plan_));
      }
      io_.ensure(Nsrc, 5);
      io_.up(src, Nsrc);
      float *d_prev = (float *)prev_[fc].ensure(sizeof(float) * (uint64_t)Nsrc);
      check(smilehip_spectral_gemaps_frames(gm_plan_, io_.d_in, Nsrc, d_prev, seen_[fc] ? 0 : 1, io_.d_out, 5, 1, nullptr));
      seen_[fc] = true;
      float five[5];
      io_.down(five, 5);
      if (gemaps_ == 1) memcpy(dst, five, sizeof(float) * 4); else dst[0] = five[4];
      g_frames[12]++;
      return (int)Ndst;
    }
    const bool compare_set = plain_ && (Nsrc == 129 || Nsrc == 257 || Nsrc == 513) && Ndst == 12 + (int)sel_[0] + (int)sel_[1] + (int)sel_[2];
    if (!compare_set && general_ < 0) {
      // everything the linear-spectrum branch of spectral.cpp:586-1560 offers except the slopes[] / alphaRatio / hammarbergIndex /
      // specDiff / fluxCentroid / standardDeviation / tonality / flatness outputs
      std::memset(&gen_opts_, 0, sizeof(gen_opts_));
      const int nb = getArraySize("bands") > 0 ? getArraySize("bands") : 0, nr = getArraySize("rollOff") > 0 ? getArraySize("rollOff") : 0;
      bool ok = nb <= 16 && nr <= 16 && getArraySize("slopes") <= 0;
      for (int b = 0; ok && b < nb; ++b) {               // bands[b] = "lo-hi" in Hz, integers (spectral.cpp:163-190)
        const char *v = getStr_f(myvprint("bands[%i]", b));
        int lo = -1, hi = -1, used = 0;
        ok = v && sscanf(v, "%d-%d%n", &lo, &hi, &used) == 2 && v[used] == 0 && lo >= 0 && hi > lo;
        gen_opts_.band_lo[b] = lo; gen_opts_.band_hi[b] = hi;
      }
      for (int i = 0; ok && i < nr; ++i) gen_opts_.rolloff[i] = getDouble_f(myvprint("rollOff[%i]", i));
      gen_opts_.n_bands = nb; gen_opts_.n_rolloff = nr;
      gen_opts_.flux = getInt("flux"); gen_opts_.centroid = getInt("centroid"); gen_opts_.max_pos = getInt("maxPos"); gen_opts_.min_pos = getInt("minPos");
      gen_opts_.entropy = getInt("entropy"); gen_opts_.variance = getInt("variance"); gen_opts_.skewness = getInt("skewness");
      gen_opts_.kurtosis = getInt("kurtosis"); gen_opts_.slope = getInt("slope"); gen_opts_.sharpness = getInt("sharpness");
      gen_opts_.harmonicity = getInt("harmonicity"); gen_opts_.flatness = getInt("flatness"); gen_opts_.log_flatness = getInt("logFlatness");
      static const char *const off[] = {"normBandEnergies", "specDiff", "specPosDiff", "fluxCentroid", "fluxAtFluxCentroid", "standardDeviation",
                                        "alphaRatio", "hammarbergIndex", "tonality", "buggyRollOff", "useLogSpectrum"};
      for (const char *o : off) ok = ok && getInt(o) == 0;
      ok = ok && getInt("squareInput") != 0 && (!gen_opts_.slope || getInt("oldSlopeScale") != 0);
      const char *fr = getStr("freqRange");
      ok = ok && fr && !strcmp(fr, "0-0");
      gen_n_out_ = ok ? smilehip_spectral_opts_count(&gen_opts_) : 0;
      general_ = (ok && gen_n_out_ > 0) ? 1 : 0;
    }
    if (!compare_set && general_ == 1 && Nsrc >= 9 && ((Nsrc - 1) & (Nsrc - 2)) == 0 && Ndst == gen_n_out_ && fc >= 0 && fc < 8) {   // (2^k + 1 bins: an FFT magnitude level, linear axis)
      if (!gen_op_[fc]) check(smilehip_spectral_op_create(context(), &gen_opts_, Nsrc, reader_->getLevelConfig()->frameSizeSec, &gen_op_[fc]));
      io_.ensure(Nsrc, gen_n_out_);
      io_.up(src, Nsrc);
      float *d_prev = (float *)prev_[fc].ensure(sizeof(float) * (uint64_t)Nsrc);
      check(smilehip_spectral_op_frames(gen_op_[fc], io_.d_in, Nsrc, d_prev, seen_[fc] ? 0 : 1, io_.d_out, gen_n_out_, 1, nullptr));
      seen_[fc] = true;
      io_.down(dst, gen_n_out_);
      g_frames[12]++;
      return (int)Ndst;
    }
    if (!compare_set || fc < 0 || fc >= 8) {
      HIP_FALLTHROUGH(12, "cSpectral: the linear-spectrum descriptor sets (bands, rollOff points, flux, centroid, maxPos, minPos, entropy, variance, skewness, kurtosis, slope, "
                          "sharpness, harmonicity, flatness; freqRange 0-0) and the two GeMAPS sets (log-spectrum slopes + alphaRatio + hammarbergIndex; flux over 0-5000 Hz) are built");
      return cSpectral::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    smilehip_plan *&pl = plans_.at(fc);
    if (!pl) {
      const sDmLevelConfig *lc = reader_->getLevelConfig();
      smilehip_lld_config c = base_config((Nsrc - 1) * 2, SMILEHIP_STAGE_SPECTRAL);
      c.force_fft_frame_size_sec = lc->frameSizeSec;    // fsSec, spectral.cpp:382-385
      for (int b = 0; b < 2; ++b) { c.spectral_band_lo[b] = band_lo_[b]; c.spectral_band_hi[b] = band_hi_[b]; }
      check(smilehip_plan_create(context(), &c, &pl));
    }
    io_.ensure(Nsrc, 15);
    io_.up(src, Nsrc);
    float *d_prev = (float *)prev_[fc].ensure(sizeof(float) * (uint64_t)Nsrc);
    check(smilehip_spectral_frames(pl, io_.d_in, Nsrc, d_prev, seen_[fc] ? 0 : 1, io_.d_out, 15, 1, nullptr));
    seen_[fc] = true;
    if (Ndst == 15) io_.down(dst, 15);
    else {
      float v[15];
      io_.down(v, 15);
      long n = 0;
      for (int k = 0; k < 15; ++k)
        if ((k != 7 || sel_[0]) && (k != 13 || sel_[1]) && (k != 14 || sel_[2])) dst[n++] = v[k];
    }
    g_frames[12]++;
    return (int)Ndst;
  }
 public:
  explicit cHipSpectral(const char *n) : cSpectral(n) {}
  ~cHipSpectral() override {
    if (gm_plan_) smilehip_plan_destroy(gm_plan_);
    for (auto *op : gen_op_) if (op) smilehip_spectral_op_destroy(op);
  }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipSpectral(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// R8  cPlp::processVector as auditory spectrum, with or without newRASTA  (src/lldcore/plp.cpp:416-593)
class cHipPlp : public cPlp {
  int fused_ = -1;
  const FusedLevel *fcols_ = nullptr;
  long fframe_ = 0, fnext_ = 0;
  FrameIO io_;
  bool cpu_warned_ = false;
  DevBytes eql_[8], state_[8], cos_[8], sin_[8];
  bool ready_[8] = {false, false, false, false, false, false, false, false};
  int plain_ = -1, newRasta_ = 0, oldRasta_ = 0, cc_ = 0, lpOrder_ = 0, firstCC_ = 0, htk_ = 0;
  FLOAT_DMEM compression_ = 0, melfloor_ = 0;
  float coef_[6] = {0, 0, 0, 0, 0, 0};
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    if (fused_ < 0) { g_fused.init(); fcols_ = g_fused.static_level(getStr("writer.dmLevel")); fused_ = fcols_ ? 1 : 0; }
    if (fused_) { if (idxi == 0) fframe_ = fnext_++; g_fused.copy(*fcols_, fframe_, dst, Ndst); return 1; }   // fused mode: rows of the whole-file batch
    if (plain_ < 0) {                                    // cPlp::myFetchConfig, plp.cpp:90-176
      int doLP = getInt("doLP"), doLpToCeps = getInt("doLpToCeps"), doIDFT = getInt("doIDFT");
      if (getInt("lpOrder") <= 0) { doLP = 0; doLpToCeps = 0; }
      if (doLpToCeps) doLP = 1;
      if (doLP) doIDFT = 1;
      newRasta_ = getInt("newRASTA");
      const int rasta = newRasta_ ? 0 : getInt("RASTA");
      oldRasta_ = rasta;
      compression_ = (FLOAT_DMEM)getDouble("compression");
      if (compression_ < 0.0) compression_ = 0.0;
      melfloor_ = (FLOAT_DMEM)getDouble("melfloor");
      htk_ = getInt("htkcompatible") ? 1 : 0;              // forces melfloor = 1, doAud = 1, doLog = doInvLog = 0 (plp.cpp:151-161)
      const bool logs_ok = (newRasta_ || rasta) ? true : (htk_ || (!getInt("doLog") && !getInt("doInvLog")));   // (either RASTA form forces doLog = doInvLog = 1, :168-174)
      plain_ = ((htk_ || getInt("doAud")) && !doIDFT && !doLP && logs_ok && !(htk_ && (newRasta_ || rasta))) ? 1 : 0;
      if (htk_) melfloor_ = 1.0;                           // the HTK-style auditory spectrum alone (config/audspec/audspec.conf)
      // PLP cepstra in HTK mode (config/plp/*.conf): doAud -> IDFT -> LP -> cepstra, c0 last
      lpOrder_ = getInt("lpOrder");
      const int lastCC = getInt("lastCC"), nCeps = getInt("nCeps");
      firstCC_ = (int)getInt("firstCC");                   // 1 (config/plp/PLP_E_*): c1 .. c12 -- the same values without the trailing c0
      cc_ = (getInt("htkcompatible") && doIDFT && doLP && doLpToCeps && !rasta && !newRasta_ && (firstCC_ == 0 || firstCC_ == 1) &&
             lpOrder_ >= 1 && lpOrder_ <= 15 && (lastCC < 0 || lastCC == lpOrder_) && (nCeps < 0 || nCeps == lpOrder_ + 1 - firstCC_)) ? 1 : 0;
      if (cc_) { plain_ = 1; melfloor_ = 1.0; }           // htkcompatible forces melfloor = 1, doAud = 1, no logs (plp.cpp:150-160)
      if (newRasta_ || oldRasta_) {                      // initTables, plp.cpp:361-399 (the same coefficients for both forms)
        const FLOAT_DMEM lo = (FLOAT_DMEM)getDouble("rastaLowerCutoff"), up = (FLOAT_DMEM)getDouble("rastaUpperCutoff");
        coef_[0] = (FLOAT_DMEM)(1.0 - sin(2.0 * M_PI * lo * reader_->getLevelT()));
        const FLOAT_DMEM om = (FLOAT_DMEM)cos(2.0 * M_PI * up * reader_->getLevelT());
        const FLOAT_DMEM norm = (FLOAT_DMEM)sqrt(10.0 * (32.0 * om * om + 8.0));
        coef_[1] = (FLOAT_DMEM)(2.0 / norm);
        coef_[2] = (FLOAT_DMEM)(-4.0 * om / norm);
        coef_[3] = 0.0;
        coef_[4] = -coef_[2];
        coef_[5] = -coef_[1];
      }
    }
    const int fc = getFconf(idxi);
    const FrameMetaInfo *fmeta = reader_->getFrameMetaInfo();
    if (!plain_ || (cc_ ? Ndst != lpOrder_ + 1 - firstCC_ : Nsrc != Ndst) || Nsrc > 64 || fc < 0 || fc >= 8 || !fmeta || idxi >= fmeta->N ||
        (long)(fmeta->field[idxi].infoSize / sizeof(double)) != Nsrc)
      { HIP_FALLTHROUGH(13, "cPlp: only the auditory spectrum (plain, RASTA, newRASTA) and the HTK PLP-CC mode are built (no partial IDFT / LP modes)"); return cPlp::processVector(src, dst, Nsrc, Ndst, idxi); }
    if (!ready_[fc]) {                                   // equal-loudness curve at the band centres, plp.cpp:335-357
      const double *frq = (const double *)(fmeta->field[idxi].info);
      std::vector<float> e((size_t)Nsrc), st((size_t)(6 * Nsrc + 2), 0.0f);
      for (long i = 0; i < Nsrc; ++i) {
        e[(size_t)i] = (cc_ || htk_) ? (FLOAT_DMEM)smileDsp_equalLoudnessWeight_htk((double)frq[i])
                           : (FLOAT_DMEM)smileDsp_equalLoudnessWeight((double)frq[i]);
        if (newRasta_ || oldRasta_) e[(size_t)i] = log(e[(size_t)i]);
      }
      if (cc_) {                                         // IDFT cosine table and lifter, plp.cpp:288-334
        const int nFreq = (int)Nsrc + 2, nAuto = lpOrder_ + 1;
        std::vector<float> ct((size_t)nAuto * nFreq), sn((size_t)nAuto);
        const FLOAT_DMEM a = (FLOAT_DMEM)M_PI / (FLOAT_DMEM)(nFreq - 1);
        for (int i = 0; i < nAuto; i++) {
          const int ib = i * nFreq;
          int m;
          ct[(size_t)ib] = 1.0;
          for (m = 1; m < (nFreq - 1); m++) ct[(size_t)(m + ib)] = (FLOAT_DMEM)(2.0 * cos(a * (double)i * (double)m));
          ct[(size_t)(m + ib)] = (FLOAT_DMEM)(cos(a * (double)i * (double)m));
        }
        const FLOAT_DMEM L = (FLOAT_DMEM)getInt("cepLifter");
        for (int i = 0; i < nAuto; i++)
          sn[(size_t)i] = (L > 0.0) ? ((FLOAT_DMEM)1.0 + L / (FLOAT_DMEM)2.0 * sin((FLOAT_DMEM)M_PI * ((FLOAT_DMEM)(i)) / L)) : (FLOAT_DMEM)1.0;
        void *d_c = cos_[fc].ensure(sizeof(float) * ct.size());
        void *d_n = sin_[fc].ensure(sizeof(float) * sn.size());
        if (smilehip_copy_to_device(context(), d_c, ct.data(), sizeof(float) * ct.size(), nullptr) ||
            smilehip_copy_to_device(context(), d_n, sn.data(), sizeof(float) * sn.size(), nullptr))
          COMP_ERR("libsmilehip: %s", smilehip_last_error());
      }
      void *d_e = eql_[fc].ensure(sizeof(float) * e.size());
      void *d_s = state_[fc].ensure(sizeof(float) * st.size());
      if (smilehip_copy_to_device(context(), d_e, e.data(), sizeof(float) * e.size(), nullptr) ||
          smilehip_copy_to_device(context(), d_s, st.data(), sizeof(float) * st.size(), nullptr))
        COMP_ERR("libsmilehip: %s", smilehip_last_error());
      ready_[fc] = true;
    }
    io_.ensure(Nsrc, cc_ ? lpOrder_ + 1 : Ndst);
    io_.up(src, Nsrc);
    if (cc_)
      check(smilehip_plp_cc_frames(context(), io_.d_in, Nsrc, (int)Nsrc, (const float *)eql_[fc].d, melfloor_, compression_, lpOrder_,
                                   (const float *)cos_[fc].d, (const float *)sin_[fc].d, io_.d_out, lpOrder_ + 1, 1, nullptr));
    else
      check(smilehip_plp_audspec_frames(context(), io_.d_in, Nsrc, (int)Nsrc, (const float *)eql_[fc].d, melfloor_, compression_,
                                        newRasta_ ? 1 : (oldRasta_ ? 2 : 0), coef_, (float *)state_[fc].d, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[13]++;
    return (int)Ndst;
  }
 public:
  explicit cHipPlp(const char *n) : cPlp(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipPlp(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// SURVEY 8f rank 1  cFunctionals::doProcess (src/functionals/functionals.cpp:320-389): one input row (one LLD contour)
// per call through smilehip_funcspec_matrix. The instance's configuration -- functionalsEnabled and the options of the
// cFunctional* children, time norms resolved with the reference's precedence (the family's own `norm` if set, else
// masterTimeNorm, else the family's default; functionalComponent.hpp:68-76) -- is translated into a smilehip_func_spec
// once; an instance that uses an option the spec cannot express stays on the reference's own code.
class cHipFunctionals : public cFunctionals {
  int fused_ = -1;
  FusedChain::FuncAt fat_{0, 0};
  FrameIO io_;
  bool cpu_warned_ = false;
  int state_ = -1;                                       // -1 = not examined, 0 = not expressible -> reference code, 1 = spec_
  smilehip_func_spec spec_;
  int opt_int(const char *fam, const char *o) { return (int)getInt_f(myvprint("%s.%s", fam, o)); }
  double opt_dbl(const char *fam, const char *o) { return getDouble_f(myvprint("%s.%s", fam, o)); }
  bool opt_set(const char *fam, const char *o) {
    char *k = myvprint("%s.%s", fam, o);
    const bool r = isSet(k) != 0;
    free(k);
    return r;
  }
  static int parse_norm(const char *n, int fallback) {
    if (!n) return fallback;
    if (!strncmp(n, "tur", 3) || !strncmp(n, "seg", 3)) return SMILEHIP_NORM_SEGMENT;
    if (!strncmp(n, "sec", 3)) return SMILEHIP_NORM_SECOND;
    if (!strncmp(n, "fra", 3)) return SMILEHIP_NORM_FRAME;
    return fallback;
  }
  int time_norm(const char *fam) {                        // parseTimeNormOption + setTimeNorm
    const int own = parse_norm(getStr_f(myvprint("%s.norm", fam)), SMILEHIP_NORM_SEGMENT);
    if (opt_set(fam, "norm")) return own;
    if (isSet("masterTimeNorm")) {
      const char *m = getStr("masterTimeNorm");
      if (m && (!strncmp(m, "seg", 3) || !strncmp(m, "tur", 3) || !strncmp(m, "sec", 3) || !strncmp(m, "fra", 3)))
        return parse_norm(m, own);
    }
    return own;
  }
  uint32_t mask_of(const char *fam, const char *const *names, int n) {
    uint32_t m = 0;
    for (int k = 0; k < n; ++k)
      if (opt_int(fam, names[k])) m |= 1u << k;
    return m;
  }
  bool build_spec() {
    smilehip_func_spec &s = spec_;
    std::memset(&s, 0, sizeof(s));
    s.period = getInputPeriod();
    if (!(s.period > 0.0)) s.period = 1.0;                // only second-normalised values use it
    s.non_zero_functs = (int)getInt("nonZeroFuncts");
    s.ext_norm = s.means_norm = s.times_norm = s.seg_norm = s.pk_norm = s.ons_norm = s.pko_norm = SMILEHIP_NORM_SEGMENT;
    s.reg_centroid_norm = SMILEHIP_NORM_SEGMENT;
    s.seg_max_num = 20; s.seg_min_lng = 3; s.seg_pause_min_lng = 2; s.lpc_order = 5;
    const int n = getArraySize("functionalsEnabled");
    if (n < 1 || n > 12) return false;
    for (int i = 0; i < n; ++i) {
      const char *f = getStr_f(myvprint("functionalsEnabled[%i]", i));
      if (!f) return false;
      if (!strcmp(f, "Extremes")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_EXTREMES;
        static const char *const o[8] = {"max", "min", "range", "maxpos", "minpos", "amean", "maxameandist", "minameandist"};
        s.ext_mask = mask_of(f, o, 8);
        s.ext_norm = time_norm(f);
      } else if (!strcmp(f, "Means")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_MEANS;
        static const char *const o[17] = {"amean", "absmean", "qmean", "nzamean", "nzabsmean", "nzqmean", "nzgmean", "nnz", "flatness",
                                          "posamean", "negamean", "posqmean", "posrqmean", "negqmean", "negrqmean", "rqmean", "nzrqmean"};
        s.means_mask = mask_of(f, o, 17);
        s.means_norm = time_norm(f);
      } else if (!strcmp(f, "Moments")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_MOMENTS;
        static const char *const o[5] = {"variance", "stddev", "skewness", "kurtosis", "amean"};
        s.mom_mask = mask_of(f, o, 5);
        s.mom_stddev_norm = opt_int(f, "stddevNorm");
        if (s.mom_stddev_norm == 1 || s.mom_stddev_norm == 2) s.mom_mask |= 1u << 5;
        s.mom_ratio_limit = opt_int(f, "doRatioLimit");
      } else if (!strcmp(f, "Regression")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_REGRESSION;
        static const char *const o[18] = {"linregc1", "linregc2", "linregerrA", "linregerrQ", "qregc1", "qregc2", "qregc3", "qregerrA",
                                          "qregerrQ", "centroid", "qregls", "qregrs", "qregx0", "qregy0", "qregyr", "qregy0nn",
                                          "qregc3nn", "qregyrnn"};
        s.reg_mask = mask_of(f, o, 18);
        const char *cn = getStr_f(myvprint("%s.centroidNorm", f));
        if (!cn || (strncmp(cn, "sec", 3) && strncmp(cn, "fra", 3) && strncmp(cn, "seg", 3))) return false;
        s.reg_centroid_norm = parse_norm(cn, SMILEHIP_NORM_SEGMENT);
        s.reg_norm_coeff = opt_int(f, "normRegCoeff");
        s.reg_norm_inputs = opt_int(f, "normInputs");
        s.reg_centroid_abs = opt_int(f, "centroidUseAbsValues");
        s.reg_centroid_limit = opt_int(f, "centroidRatioLimit");
        s.reg_ratio_limit = opt_int(f, "doRatioLimit");
        s.reg_old_buggy_qerr = opt_int(f, "oldBuggyQerr");
      } else if (!strcmp(f, "Percentiles")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_PERCENTILES;
        static const char *const q[3] = {"quartile1", "quartile2", "quartile3"}, *const r[3] = {"iqr12", "iqr23", "iqr13"};
        uint32_t m = 0;
        for (int k = 0; k < 3; ++k) if (opt_int(f, q[k])) m |= 1u << k;
        if (opt_set(f, "quartiles")) m = opt_int(f, "quartiles") ? 7u : 0u;
        uint32_t mr = 0;
        for (int k = 0; k < 3; ++k) if (opt_int(f, r[k])) mr |= 1u << (3 + k);
        if (opt_set(f, "iqr")) mr = opt_int(f, "iqr") ? 0x38u : 0u;
        s.pct_mask = m | mr;
        s.pct_interp = opt_int(f, "interp");
        char *k = myvprint("%s.percentile", f);
        s.n_pctl = getArraySize(k); free(k);
        k = myvprint("%s.pctlrange", f);
        s.n_range = getArraySize(k); free(k);
        k = myvprint("%s.pctlquotient", f);
        const int nq = getArraySize(k); free(k);
        if (s.n_pctl < 0 || s.n_pctl > 8 || s.n_range < 0 || s.n_range > 8) return false;
        if (s.n_pctl > 0 && nq > 0) return false;         // quotients are not expressible
        if (s.n_pctl == 0) s.n_range = 0;
        for (int j = 0; j < s.n_pctl; ++j) {
          double v = getDouble_f(myvprint("%s.percentile[%i]", f, j));
          s.pctl[j] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        }
        for (int j = 0; j < s.n_range; ++j) {
          const char *t = getStr_f(myvprint("%s.pctlrange[%i]", f, j));
          int a = -1, b = -1;
          if (!t || sscanf(t, "%d-%d", &a, &b) != 2) return false;
          s.range_a[j] = a; s.range_b[j] = b;
        }
      } else if (!strcmp(f, "Times")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_TIMES;
        static const char *const o[13] = {"upleveltime25", "downleveltime25", "upleveltime50", "downleveltime50", "upleveltime75",
                                          "downleveltime75", "upleveltime90", "downleveltime90", "risetime", "falltime", "leftctime",
                                          "rightctime", "duration"};
        s.times_mask = mask_of(f, o, 13);
        s.times_norm = time_norm(f);
        s.times_buggy_sec_norm = opt_int(f, "buggySecNorm");
        char *k = myvprint("%s.upleveltime", f);
        const int nu = getArraySize(k); free(k);
        k = myvprint("%s.downleveltime", f);
        const int nd = getArraySize(k); free(k);
        if (nu > 0 || nd > 0 || opt_int(f, "useRobustPercentileRange")) return false;
      } else if (!strcmp(f, "Segments")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_SEGMENTS;
        static const char *const o[5] = {"numSegments", "meanSegLen", "maxSegLen", "minSegLen", "segLenStddev"};
        s.seg_mask = mask_of(f, o, 5);
        s.seg_norm = time_norm(f);
        const char *alg = getStr_f(myvprint("%s.segmentationAlgorithm", f));
        if (!alg) return false;
        if (!strncmp(alg, "relTh", 5)) s.seg_algo = SMILEHIP_SEG_RELTH;
        else if (!strncmp(alg, "nonX", 4)) s.seg_algo = SMILEHIP_SEG_NONX;
        else if (!strncmp(alg, "eqX", 3)) s.seg_algo = SMILEHIP_SEG_EQX;
        else return false;
        if (opt_int(f, "growDynSegBuffer") || opt_int(f, "useOldBuggyChX")) return false;
        s.seg_max_num = opt_int(f, "maxNumSeg");
        s.seg_min_lng = opt_int(f, "segMinLng");
        if (s.seg_min_lng < 1) s.seg_min_lng = 1;
        s.seg_auto_min_lng = opt_set(f, "segMinLng") ? 0 : 1;
        s.seg_pause_min_lng = opt_int(f, "pauseMinLng");
        if (s.seg_pause_min_lng < 1) s.seg_pause_min_lng = 1;
        s.seg_x = (float)opt_dbl(f, "X");
        s.seg_x_is_rel = opt_int(f, "XisRel");
        if (s.seg_algo == SMILEHIP_SEG_RELTH) {
          char *k = myvprint("%s.thresholds", f);
          s.seg_n_thresholds = getArraySize(k); free(k);
          if (s.seg_n_thresholds < 0 || s.seg_n_thresholds > 8) return false;
          for (int j = 0; j < s.seg_n_thresholds; ++j) {
            float v = (float)getDouble_f(myvprint("%s.thresholds[%i]", f, j));
            s.seg_thresholds[j] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
          }
        }
      } else if (!strcmp(f, "Lpc")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_LPC;
        s.lpc_first = opt_int(f, "firstCoeff");
        if (s.lpc_first < 0) s.lpc_first = 0;
        s.lpc_order = opt_int(f, "order");
        if (s.lpc_order <= s.lpc_first) return false;
        s.lpc_gain = opt_int(f, "lpGain") ? 1 : 0;
        s.lpc_coeffs = opt_int(f, "lpc") ? 1 : 0;
      } else if (!strcmp(f, "Peaks2")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_PEAKS2;
        static const char *const o[32] = {"numPeaks", "meanPeakDist", "meanPeakDistDelta", "peakDistStddev", "peakRangeAbs", "peakRangeRel",
                                          "peakMeanAbs", "peakMeanMeanDist", "peakMeanRel", "ptpAmpMeanAbs", "ptpAmpMeanRel",
                                          "ptpAmpStddevAbs", "ptpAmpStddevRel", "minRangeAbs", "minRangeRel", "minMeanAbs",
                                          "minMeanMeanDist", "minMeanRel", "mtmAmpMeanAbs", "mtmAmpMeanRel", "mtmAmpStddevAbs",
                                          "mtmAmpStddevRel", "meanRisingSlope", "maxRisingSlope", "minRisingSlope",
                                          "stddevRisingSlope", "meanFallingSlope", "maxFallingSlope", "minFallingSlope",
                                          "stddevFallingSlope", "covFallingSlope", "covRisingSlope"};
        s.pk_mask = mask_of(f, o, 32);
        s.pk_norm = time_norm(f);
        if (opt_int(f, "noClearPeakList")) return false;
        const char *dbg = getStr_f(myvprint("%s.posDbgOutp", f));
        if ((dbg && *dbg) || opt_int(f, "consoleDbg")) return false;
        s.pk_ratio_limit = opt_int(f, "doRatioLimit");
        s.pk_dyn_rel = opt_int(f, "dynRelThresh");
        float rt = (float)opt_dbl(f, "relThresh");
        if (rt < 0) rt = 0.0f;
        else if (rt > 1.0f && !s.pk_dyn_rel) rt = 1.0f;
        s.pk_rel_thresh = rt;
        if (opt_set(f, "absThresh")) {
          s.pk_use_abs = 1;
          s.pk_abs_thresh = (float)opt_dbl(f, "absThresh");
          s.pk_dyn_rel = 0;
        }
      } else if (!strcmp(f, "Peaks")) {
        if (!opt_int(f, "overlapFlag")) return false;      // overlapFlag = 0 carries the last two values from call to call
        s.fam[s.n_fam++] = SMILEHIP_FAM_PEAKS;
        static const char *const o[5] = {"numPeaks", "meanPeakDist", "peakMean", "peakMeanMeanDist", "peakDistStddev"};
        s.pko_mask = mask_of(f, o, 5);
        s.pko_norm = time_norm(f);
      } else if (!strcmp(f, "Crossings")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_CROSSINGS;
        static const char *const o[3] = {"zcr", "mcr", "amean"};
        s.crs_mask = mask_of(f, o, 3);
      } else if (!strcmp(f, "DCT")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_DCT;
        s.dct_first = opt_int(f, "firstCoeff");
        if (s.dct_first < 0) s.dct_first = 0;
        s.dct_last = opt_set(f, "nCoeffs") ? s.dct_first + opt_int(f, "nCoeffs") - 1 : opt_int(f, "lastCoeff");
        if (s.dct_last < s.dct_first || s.dct_last - s.dct_first >= 64) return false;
      } else if (!strcmp(f, "Samples")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_SAMPLES;
        char *k = myvprint("%s.samplepos", f);
        const int ns = getArraySize(k); free(k);
        if (ns > 8) return false;
        if (ns > 0) {
          s.n_samples = ns;
          for (int j = 0; j < ns; ++j) {
            double v = getDouble_f(myvprint("%s.samplepos[%i]", f, j));
            s.sample_pos[j] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
          }
        } else {                                          // DEFAULT_NR_SAMPLES = 5 (functionalSamples.cpp:27, 78-84)
          s.n_samples = 5;
          for (int j = 0; j < 5; ++j) s.sample_pos[j] = (double)j / (5 - 1.0);
        }
      } else if (!strcmp(f, "Onset")) {
        s.fam[s.n_fam++] = SMILEHIP_FAM_ONSET;
        static const char *const o[5] = {"onsetPos", "offsetPos", "numOnsets", "numOffsets", "onsetRate"};
        s.ons_mask = mask_of(f, o, 5);
        s.ons_norm = time_norm(f);
        s.ons_use_abs = opt_int(f, "useAbsVal");
        s.ons_thr_on = s.ons_thr_off = (float)opt_dbl(f, "threshold");            // functionalOnset.cpp:72-76
        if (opt_set(f, "thresholdOnset")) s.ons_thr_on = (float)opt_dbl(f, "thresholdOnset");
        if (opt_set(f, "thresholdOffset")) s.ons_thr_off = (float)opt_dbl(f, "thresholdOffset");
      } else {
        return false;                                     // a family that is not built (ModulationSpec, ...)
      }
    }
    return smilehip_funcspec_count(&s) == nFunctValues;
  }
 protected:
  int doProcess(int i, cMatrix *row, FLOAT_DMEM *y) override {
    if (fused_ < 0) {                                      // big-set fused mode: the instance's values of the fused batch's functionals vector
      g_fused.init();
      fused_ = 0;
      if (g_fused.big) {
        auto it = g_fused.func_levels.find(getStr("writer.dmLevel"));
        if (it != g_fused.func_levels.end() && it->second.count == nFunctValues) { fat_ = it->second; fused_ = 1; }
      }
    }
    if (fused_ && row->nT > 0) {
      const long at = fat_.base + (long)i * fat_.count;
      if (at + fat_.count > (long)g_fused.func.size()) COMP_ERR("libsmilehip plugin: fused mode: functionals element %d outside the batch's vector", i);
      for (int k = 0; k < fat_.count; ++k) y[k] = g_fused.func[(size_t)(at + k)];
      g_fused.served++;
      return nFunctValues;
    }
    if (state_ < 0) state_ = build_spec() ? 1 : 0;
    if (!state_ || row->nT <= 0) { if (!state_) HIP_FALLTHROUGH(14, "cFunctionals: a functional family or option of this instance is not built (Crossings, DCT, Onset, Peaks, Samples, ModulationSpec, pctlquotient, ...)"); return cFunctionals::doProcess(i, row, y); }
    io_.ensure(row->nT, nFunctValues);
    io_.up(row->data, row->nT);
    check(smilehip_funcspec_matrix(context(), &spec_, io_.d_in, 1, row->nT, 1, io_.d_out, nullptr));
    io_.down(y, nFunctValues);
    g_frames[14]++;
    return nFunctValues;
  }
 public:
  explicit cHipFunctionals(const char *n) : cFunctionals(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipFunctionals(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// SURVEY 8(f) rank 2, per component. An F0-chain plan carries cSpecScale's spline / weighting tables and cPitchShs'
// shifts for one spectrum geometry (bins, frameSizeSec of the magnitude level).
static smilehip_plan *f0_component_plan(long K, double frame_size_sec, double min_pitch, double max_pitch, double cutoff,
                                        int n_harm, double compression, double min_f = 25.0, int n_cand = 6, int old_peaks = 0) {
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
class cHipSpecScale : public cSpecScale {
  FrameIO io_;
  bool cpu_warned_ = false;
  smilehip_plan *pl_ = nullptr;
  int usable_ = -1;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE((int)Ndst);
    if (usable_ < 0) {
      const char *sc = getStr("scale"), *ss = getStr("sourceScale"), *im = getStr("interpMethod");
      usable_ = sc && !strncasecmp(sc, "oct", 3) && ss && !strncasecmp(ss, "lin", 3) && im && !strncasecmp(im, "spl", 3) &&
                getDouble("minF") > 0.0 && getDouble("maxF") == -1.0 && getInt("nPointsTarget") <= 0 && getInt("specSmooth") == 1 &&
                getInt("specEnhance") == 1 && getInt("auditoryWeighting") == 1 && Nsrc == Ndst;
      if (usable_) {
        pl_ = f0_component_plan(Nsrc, (double)(float)reader_->getLevelConfig()->frameSizeSec, 52.0, 620.0, 0.7, 15, 0.85, getDouble("minF"));
        if (!pl_) usable_ = 0;
      }
    }
    if (!usable_) { HIP_FALLTHROUGH(15, "cSpecScale: only the octave-scale spline set of the F0 chains on spectra of 512 .. 4096 points is built"); return cSpecScale::processVector(src, dst, Nsrc, Ndst, idxi); }
    io_.ensure(Nsrc, Ndst);
    io_.up(src, Nsrc);
    check(smilehip_specscale_frames(pl_, io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[15]++;
    return (int)Ndst;
  }
 public:
  explicit cHipSpecScale(const char *n) : cSpecScale(n) {}
  ~cHipSpecScale() override { if (pl_) smilehip_plan_destroy(pl_); }
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipSpecScale(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cPitchBase::processVector around cPitchShs::pitchDetect (src/lldcore/pitchBase.cpp:187-310, src/lld/pitchShs.cpp:214-347)
// for six candidates with scores + voicing, F0raw + voicingClip, greedyPeakAlgo, no octave correction / lfCut / SHS dump.
class cHipPitchShs : public cPitchShs {
  FrameIO io_;
  bool cpu_warned_ = false;
  smilehip_plan *pl_ = nullptr;
  int usable_ = -1;
  bool raw_ = true, clip_ = true;
  int nc_ = 6;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE((int)Ndst);
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
    check(smilehip_pitchshs_frames(pl_, io_.d_in, Nsrc, io_.d_out, 21, 1, nullptr));
    if (raw_ && clip_ && nc_ == 6) io_.down(dst, 21);
    else {                                                  // [nCandidates | F0Cand | candVoicing | candScores] (+ F0raw) (+ voicingClip)
      float v[21];                                          // (the device rows keep six slots per field)
      io_.down(v, 21);
      long n = 0;
      dst[n++] = v[0];
      for (int f = 0; f < 3; ++f)
        for (int c = 0; c < nc_; ++c) dst[n++] = v[1 + 6 * f + c];
      if (raw_) dst[n++] = v[19];
      if (clip_) dst[n++] = v[20];
    }
    g_frames[16]++;
    return (int)Ndst;
  }
 public:
  explicit cHipPitchShs(const char *n) : cPitchShs(n) {}
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

// cSpecResample::processVector (src/dsp/specResample.cpp:175-185) for [gemapsv01b_resampLpc]
class cHipSpecResample : public cSpecResample {
  FrameIO io_;
  DevBytes cos_, sin_;
  bool cpu_warned_ = false;
  int usable_ = -1;                                       // 1: eGeMAPS' fused geometry (plan tables), 2: any geometry (the instance's own tables)
  long rate_ = 0;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE(1);
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
    if (usable_ == 1) check(smilehip_specresample_frames(gemaps_plan(rate_), io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
    else check(smilehip_specresample_table_frames(context(), io_.d_in, Nsrc, Nsrc, Ndst, dftWork->kMax, (const float *)cos_.d,
                                                  (const float *)sin_.d, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[17]++;
    return (int)Ndst;
  }
 public:
  explicit cHipSpecResample(const char *n) : cSpecResample(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipSpecResample(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cLpc::processVector (src/lld/lpc.cpp:171-213) with method = acf, saveLPCoeff only: p = 11 on 220 samples through the eGeMAPS plan,
// any other frame length and order p <= 32 through smilehip_lpc_acf_frames
class cHipLpc : public cLpc {
  FrameIO io_;
  bool cpu_warned_ = false;
  int usable_ = -1;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE(1);
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
    if (usable_ == 1) check(smilehip_lpc_frames(gemaps_plan(), io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
    else check(smilehip_lpc_acf_frames(context(), io_.d_in, Nsrc, Nsrc, (int32_t)p, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[18]++;
    return 1;
  }
 public:
  explicit cHipLpc(const char *n) : cLpc(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipLpc(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cFormantLpc::processVector (src/lld/formantLpc.cpp:192-290): 5 formants + bandwidths from 11 LP coefficients at 11 kHz
class cHipFormantLpc : public cFormantLpc {
  FrameIO io_;
  bool cpu_warned_ = false;
  int usable_ = -1;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE((int)Ndst);
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
    check(smilehip_formantlpc_frames(gemaps_plan(0, (long)getDouble("maxF")), io_.d_in, Nsrc, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[19]++;
    return (int)Ndst;
  }
 public:
  explicit cHipFormantLpc(const char *n) : cFormantLpc(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipFormantLpc(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cHarmonics::processVector (src/lld/harmonics.cpp:743-1031) with [gemapsv01b_harmonics]'s options: the input vector holds the F0
// element, the formant frequency / bandwidth fields and the 513-bin magnitude field; the positions are looked up by name as the
// reference does in setupNewNames (:226-307).
class cHipHarmonics : public cHarmonics {
  FrameIO io_;
  bool cpu_warned_ = false;
  int usable_ = -1;
  long iF0_ = -1, iSpec_ = -1, iFf_ = -1, iFb_ = -1, nSpec_ = 0, nFf_ = 0, nFb_ = 0;
  DevBytes fm_, f0_;
 protected:
  int processVector(const FLOAT_DMEM *src, FLOAT_DMEM *dst, long Nsrc, long Ndst, int idxi) override {
    FUSED_BIG_STAGE(1);
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
      if (ok) {
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
          smilehip_geometry g;
          check(smilehip_plan_geometry(gemaps_plan(), &g));
          // the plan of the rate seen last (cSpecResample / cSpectral run before this component in every tick): its 60 ms
          // spectrum must be this one -- same number of bins, same axis step
          const double step = (double)g_gm_rate / (double)(2 * (nSpec_ - 1));
          axis = frq[0] == 0.0 && frq[1] == step && frq[nSpec_ - 1] == step * (double)(nSpec_ - 1);
        }
        ok = axis && iF0_ >= 0 && iSpec_ >= 0 && iFf_ >= 0 && iFb_ >= 0 && size_ok && nFf_ == 5 && nFb_ == 5 && iSpec_ + nSpec_ <= Nsrc;
      }
      usable_ = ok ? 1 : 0;
    }
    if (!usable_) {
      HIP_FALLTHROUGH(20, "cHarmonics: only GeMAPS' option set (H1-H2, H1-A3, formant amplitudes 1..3, ACF HNR in dB; 5 formants, the "
                          "spectrum of 60 ms frames at 8 .. 48 kHz) is built");
      return cHarmonics::processVector(src, dst, Nsrc, Ndst, idxi);
    }
    io_.ensure(nSpec_, 6);
    io_.up(src + iSpec_, nSpec_);
    float fm[10];
    memcpy(fm, src + iFf_, sizeof(float) * 5);
    memcpy(fm + 5, src + iFb_, sizeof(float) * 5);
    float *d_fm = (float *)fm_.ensure(sizeof(float) * 10), *d_f0 = (float *)f0_.ensure(sizeof(float));
    if (smilehip_copy_to_device(context(), d_fm, fm, sizeof(fm), nullptr) ||
        smilehip_copy_to_device(context(), d_f0, src + iF0_, sizeof(float), nullptr))
      COMP_ERR("libsmilehip: %s", smilehip_last_error());
    check(smilehip_harmonics_frames(gemaps_plan(), d_f0, d_fm, 10, io_.d_in, nSpec_, io_.d_out, 6, 1, nullptr));
    io_.down(dst, 6);
    g_frames[20]++;
    return 1;
  }
 public:
  explicit cHipHarmonics(const char *n) : cHarmonics(n) {}
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
 protected:
  eTickResult myTick(long long t) override {
    g_fused.init();
    if (g_fused.big) return cPitchJitter::myTick(t);            // big-set fused mode: a stage on zero-filled levels (the reference's own tick code keeps the frame bookkeeping)
    if (!ready_) {
      ready_ = true;
      usable_ = !jitterLocalEnv && !jitterDDPEnv && !shimmerLocalEnv && !shimmerLocalDBEnv && !shimmerUseRmsAmplitude && !harmonicERMS &&
                !noiseERMS && !linearHNR && !sourceQualityRange && !sourceQualityMean && !periodLengths && !periodStarts && !refinedF0 &&
                !usePeakToPeakPeriodLength_ && minNumPeriods == 2 && filehandle == NULL &&
                (useBrokenJitterThresh_ || threshCC_ == (FLOAT_DMEM)0.5) && lgHNRfloor == (FLOAT_DMEM)-100.0 && reader_->getLevelN() == 1;
    }
    if (!usable_) {
      HIP_FALLTHROUGH(23, "cPitchJitter: only jitterLocal / jitterDDP / shimmerLocal / shimmerLocalDB / logHNR with minNumPeriods = 2, minCC = 0.5 "
                          "(or useBrokenJitterThresh), lgHNRfloor = -100 on a mono wave level are built");
      return cPitchJitter::myTick(t);
    }
    if (isEOI()) return TICK_INACTIVE;
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

// ---- the components the other INTERSPEECH sets of config/is09-13 add (IS10_paraling, IS11_speaker_state, IS12_speaker_trait) ----
// cIntensity::processVector (src/lldcore/intensity.cpp:125-145)
class cHipIntensity : public cIntensity {
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
    check(smilehip_intensity_frames(context(), io_.d_in, Nsrc, Nsrc, flags, io_.d_out, Ndst, 1, nullptr));
    io_.down(dst, Ndst);
    g_frames[24]++;
    return (int)Ndst;
  }
 public:
  explicit cHipIntensity(const char *n) : cIntensity(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipIntensity(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

// cLsp::processVector (src/lld/lsp.cpp:289-312)
class cHipLsp : public cLsp {
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
    check(smilehip_lsp_frames(context(), io_.d_in, nLpc, (int32_t)nLpc, io_.d_out, nLpc, 1, nullptr));
    io_.down(dst, nLpc);
    g_frames[25]++;
    return 1;
  }
 public:
  explicit cHipLsp(const char *n) : cLsp(n) {}
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
class cHipVectorOperation : public cVectorOperation {
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
    check(smilehip_vecop_frames(context(), op, param1, logfloor, io_.d_in, n, (int32_t)n, io_.d_out, n, 1, nullptr));
    io_.down(dst, reduce ? 1 : n);
    g_frames[27]++;
    return 1;
  }
 public:
  explicit cHipVectorOperation(const char *n) : cVectorOperation(n) {}
  static cSmileComponent *create(const char *n) {
    cSmileComponent *c = new cHipVectorOperation(n);
    c->setComponentInfo(scname, sdescription);
    return c;
  }
};

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
 protected:
  eTickResult myTick(long long t) override {
    g_fused.init();
    if (g_fused.big) return cValbasedSelector::myTick(t);            // big-set fused mode: a stage on zero-filled levels (the reference's own tick code keeps the frame bookkeeping)
    if (!ready_) {
      threshold_ = (FLOAT_DMEM)getDouble("threshold");
      adaptive_ = (int)getInt("adaptiveThreshold");
      idx_ = getInt("idx"); invert_ = getInt("invert"); allowEqual_ = getInt("allowEqual");
      removeIdx_ = getInt("removeIdx"); zerovec_ = getInt("zeroVec");
      outputVal_ = (FLOAT_DMEM)getDouble("outputVal");
      ready_ = true;
    }
    if (adaptive_) { HIP_FALLTHROUGH(22, "cValbasedSelector: adaptiveThreshold = 1 is not built"); return cValbasedSelector::myTick(t); }
    if (!writer_->checkWrite(1)) return TICK_DEST_NO_SPACE;
    cVector *vec = reader_->getNextFrame();
    if (vec == NULL) return TICK_SOURCE_NOT_AVAIL;
    const long N = vec->N, nOut = removeIdx_ ? N - 1 : N;
    if (nOut < 1) { HIP_FALLTHROUGH(22, "cValbasedSelector: removeIdx on a one-element vector"); return TICK_INACTIVE; }
    io_.ensure(N, nOut);
    io_.up(vec->data, N);
    int32_t *d_keep = (int32_t *)keep_.ensure(sizeof(int32_t));
    check(smilehip_valbased_select_frames(context(), io_.d_in, N, N, 1, (int32_t)idx_, threshold_, invert_, allowEqual_, zerovec_, removeIdx_,
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
 protected:
  eTickResult myTick(long long t) override {
    g_fused.init();
    if (!ready_) setup();
    if (!usable_) {
      HIP_FALLTHROUGH(21, "cPitchSmootherViterbi: only one input level with up to six candidates and bufferLength <= 128 is built");
      return cPitchSmootherViterbi::myTick(t);
    }
    int32_t n = 0, fr[128], st[128];
    if (g_fused.big) {
      // big-set fused mode: this level is a stage (zeros), but WHEN its frames appear shapes every end-of-input rule downstream (the
      // frames the Viterbi pass has not decided when the input ends arrive in the flush): frames are "decided" at once except the
      // last P, P as the fused batch's own pass left it
      const long T = g_fused.f0_frames, P = g_fused.f0_pending;
      if (isEOI()) {
        if (!flushed_) {
          for (long f = (T - P > 0 ? T - P : 0); f < (long)hist_.size() && n < 128; ++f) { fr[n] = (int32_t)f; st[n] = (int32_t)nCand_; ++n; }
          flushed_ = true;
        }
      } else {
        cVector *vec = reader_->getNextFrame();
        if (vec == NULL) return TICK_SOURCE_NOT_AVAIL;
        std::vector<FLOAT_DMEM> h((size_t)(2 * nCand_ + 4), 0.0f);
        h[(size_t)(2 * nCand_ + 3)] = (FLOAT_DMEM)vec->tmeta->vIdx;
        hist_.push_back(h);
        const long f = (long)hist_.size() - 1;
        if (f < T - P) { fr[0] = (int32_t)f; st[0] = (int32_t)nCand_; n = 1; }
        g_fused_stage++;
      }
    } else if (isEOI()) {
      if (!flushed_) {
        check(smilehip_viterbi_stream_flush(vs_, &n, fr, st, 128));
        flushed_ = true;
      }
    } else {
      cVector *vec = reader_->getNextFrame();
      if (vec == NULL) return TICK_SOURCE_NOT_AVAIL;
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
    for (int i = 0; i < n; ++i) queue_.push_back(std::make_pair((int)fr[i], (int)st[i]));
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

// optional usage trace: SMILEHIP_PLUGIN_TRACE=<file> gets one line per overridden
// component with the number of frames it pushed through the HIP kernels
struct TraceAtExit {
  ~TraceAtExit() {
    const char *path = getenv("SMILEHIP_PLUGIN_TRACE");
    if (!path) return;
    FILE *f = fopen(path, "a");
    if (!f) return;
    for (int i = 0; i < kNumOverrides; ++i) fprintf(f, "%s %ld\n", g_names[i], g_frames[i]);
    for (int i = 0; i < kNumOverrides; ++i) fprintf(f, "%s.cpu %ld\n", g_names[i], g_cpu[i]);
    fprintf(f, "fused.rows %ld\nfused.stage_frames %ld\nfused.batch_frames %ld\n", g_fused.served, g_fused_stage, g_fused.active ? g_fused.n_rows : 0L);
    fclose(f);
  }
} g_trace;

typedef sComponentInfo *(*regfn)(cConfigManager *, cComponentManager *, int);
typedef cSmileComponent *(*createfn)(const char *);

sComponentInfo *override_of(regfn builtin, createfn mine, cConfigManager *c, cComponentManager *m, int it,
                            sComponentInfo *next) {
  // the built-in registerComponent builds the info (and re-offers the ConfigType,
  // which the config manager ignores because the type already exists)
  sComponentInfo *ci = builtin(c, m, it);
  if (!ci) return next;
  ci->create = mine;
  ci->builtIn = 0;
  ci->next = next;
  return ci;
}

}  // namespace

// The loader's entry point: type registerFunction, src/include/core/componentManager.hpp:23
extern "C" sComponentInfo *registerPluginComponent(cConfigManager *confman, cComponentManager *compman, int iteration) {
  sComponentInfo *head = nullptr;
  const char *only = getenv("SMILEHIP_PLUGIN_COMPONENTS");   // e.g. "cMelspec,cMfcc"; default: all twenty-eight
  auto want = [&](const char *name) {                       // whole names of the comma-separated list
    if (!only) return true;
    const size_t n = strlen(name);
    for (const char *p = only; (p = strstr(p, name)) != nullptr; p += n)
      if ((p == only || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
  };
  if (want("cHipLldSource")) {                             // a NEW type (fused mode), not an override
    sComponentInfo *ci = cHipLldSource::registerComponent(confman, compman, iteration);
    if (ci) { ci->builtIn = 0; ci->next = head; head = ci; }
  }
  if (want("cVectorOperation")) head = override_of(&cVectorOperation::registerComponent, &cHipVectorOperation::create, confman, compman, iteration, head);
  if (want("cPitchSmoother")) head = override_of(&cPitchSmoother::registerComponent, &cHipPitchSmoother::create, confman, compman, iteration, head);
  if (want("cLsp")) head = override_of(&cLsp::registerComponent, &cHipLsp::create, confman, compman, iteration, head);
  if (want("cIntensity")) head = override_of(&cIntensity::registerComponent, &cHipIntensity::create, confman, compman, iteration, head);
  if (want("cPitchJitter")) head = override_of(&cPitchJitter::registerComponent, &cHipPitchJitter::create, confman, compman, iteration, head);
  if (want("cValbasedSelector")) head = override_of(&cValbasedSelector::registerComponent, &cHipValbasedSelector::create, confman, compman, iteration, head);
  if (want("cPitchSmootherViterbi")) head = override_of(&cPitchSmootherViterbi::registerComponent, &cHipPitchSmootherViterbi::create, confman, compman, iteration, head);
  if (want("cHarmonics")) head = override_of(&cHarmonics::registerComponent, &cHipHarmonics::create, confman, compman, iteration, head);
  if (want("cFormantLpc")) head = override_of(&cFormantLpc::registerComponent, &cHipFormantLpc::create, confman, compman, iteration, head);
  if (want("cLpc")) head = override_of(&cLpc::registerComponent, &cHipLpc::create, confman, compman, iteration, head);
  if (want("cSpecResample")) head = override_of(&cSpecResample::registerComponent, &cHipSpecResample::create, confman, compman, iteration, head);
  if (want("cPitchShs")) head = override_of(&cPitchShs::registerComponent, &cHipPitchShs::create, confman, compman, iteration, head);
  if (want("cSpecScale")) head = override_of(&cSpecScale::registerComponent, &cHipSpecScale::create, confman, compman, iteration, head);
  if (want("cFunctionals")) head = override_of(&cFunctionals::registerComponent, &cHipFunctionals::create, confman, compman, iteration, head);
  if (want("cPlp")) head = override_of(&cPlp::registerComponent, &cHipPlp::create, confman, compman, iteration, head);
  if (want("cSpectral")) head = override_of(&cSpectral::registerComponent, &cHipSpectral::create, confman, compman, iteration, head);
  if (want("cContourSmoother")) head = override_of(&cContourSmoother::registerComponent, &cHipContourSmoother::create, confman, compman, iteration, head);
  if (want("cDeltaRegression")) head = override_of(&cDeltaRegression::registerComponent, &cHipDeltaRegression::create, confman, compman, iteration, head);
  if (want("cPitchACF")) head = override_of(&cPitchACF::registerComponent, &cHipPitchACF::create, confman, compman, iteration, head);
  if (want("cAcf")) head = override_of(&cAcf::registerComponent, &cHipAcf::create, confman, compman, iteration, head);
  if (want("cMZcr")) head = override_of(&cMZcr::registerComponent, &cHipMZcr::create, confman, compman, iteration, head);
  if (want("cEnergy")) head = override_of(&cEnergy::registerComponent, &cHipEnergy::create, confman, compman, iteration, head);
  if (want("cMfcc")) head = override_of(&cMfcc::registerComponent, &cHipMfcc::create, confman, compman, iteration, head);
  if (want("cMelspec")) head = override_of(&cMelspec::registerComponent, &cHipMelspec::create, confman, compman, iteration, head);
  if (want("cFFTmagphase")) head = override_of(&cFFTmagphase::registerComponent, &cHipFFTmagphase::create, confman, compman, iteration, head);
  if (want("cTransformFFT")) head = override_of(&cTransformFFT::registerComponent, &cHipTransformFFT::create, confman, compman, iteration, head);
  if (want("cWindower")) head = override_of(&cWindower::registerComponent, &cHipWindower::create, confman, compman, iteration, head);
  if (want("cVectorPreemphasis")) head = override_of(&cVectorPreemphasis::registerComponent, &cHipVectorPreemphasis::create, confman, compman, iteration, head);
  return head;
}
