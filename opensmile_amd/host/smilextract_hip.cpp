// smilextract_hip -- batch front end of the fused GPU path for the two feature sets whose
// whole graph the C ABI covers: config/mfcc/MFCC12_0_D_A.conf, config/plp/PLP_0_D_A.conf and
// config/is09-13/IS09_emotion.conf.
// It is NOT a re-implementation of SMILExtract's config language (that stays with the
// reference; unmodified configs run through the plugin): the set is picked
// by name and the file options keep the names those configs declare via \cm[...]:
//
//   smilextract_hip --set mfcc12_0_d_a|mfcc12_e_d_a|mfcc12_0_d_a_z|mfcc12_e_d_a_z|plp_0_d_a|plp_e_d_a|plp_0_d_a_z|plp_e_d_a_z
//                   (-I in.wav | -filelist list.txt) [-O lld.htk] [-csvoutput lld.csv]
//   smilextract_hip --set is09_emotion  (-I in.wav | -filelist list.txt) [-O func.arff] [-csvoutput func.csv]
//   smilextract_hip --set compare16_lld (-I in.wav | -filelist list.txt) [-lldcsvoutput lld.csv] [-lldhtkoutput lld.htk]
//                   (the 130-column LLD level of ComParE_2016 only)
//   smilextract_hip --set compare16     same options as is09_emotion: the whole ComParE_2016.conf, LLD level + 6373 functionals
//   smilextract_hip --set is13_compare  the same for config/is09-13/IS13_ComParE.conf
//   smilextract_hip --set egemapsv02    the whole config/egemaps/v02/eGeMAPSv02.conf: 25-column LLD level + 88 functionals
//   smilextract_hip --set gemapsv01a | egemapsv01a   the v01a files: the same columns, zeroPadSymmetric = 0, useBrokenJitterThresh = 1, maxF = 5500
//   smilextract_hip --set gemapsv01b | egemapsv01b   config/gemaps/v01b/GeMAPSv01b.conf (18 LLDs, 62 functionals) / config/egemaps/v01b/
//                                       eGeMAPSv01b.conf (23, 88): sub-graphs of eGeMAPSv02.conf, written as column subsets of its levels
//                   [-htkoutput func.htk] [-lldcsvoutput lld.csv] [-lldhtkoutput lld.htk]
//   common: [-instname name] [-outdir dir] [--device d] [--rank r --world n] [--chunk-files n]
//
// -filelist: one "wav[<TAB>instname]" per line. Per-file outputs (HTK, LLD CSV) of a list go to
// -outdir/<basename>.<ext>; summary outputs (func ARFF/CSV) append one row per file, like the
// reference's append=1 default. --rank/--world: this process takes its share of an LPT partition of the list by file size
// (utterances shard with no communication; one process per GPU); the summary outputs of rank r then go to
// <name>.rank<r><ext> -- ranks never share a file -- and are concatenated afterwards.
// All files of a chunk are packed into one device batch: one kernel sequence per chunk.
#include <dlfcn.h>
#include <sched.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "conf_plan.hpp"
#include "smilehip.h"
#include "smilehip_comm.h"
#include "smilehip_host.hpp"

using namespace smilehip_host;

namespace {

struct Job { std::string wav, inst; };

std::function<void()> g_before_exit;                    // waits for the ingest thread that reads ahead (exit() under a running thread is a crash at best)
[[noreturn]] void die(const std::string &m) {
  fprintf(stderr, "smilextract_hip: %s\n", m.c_str());
  if (g_before_exit) { auto f = g_before_exit; g_before_exit = nullptr; f(); }
  exit(1);
}
void check(int rc, const char *what) {
  if (rc != SMILEHIP_OK) die(std::string(what) + ": " + smilehip_last_error());
}
std::string basename_noext(const std::string &p) {
  size_t s = p.find_last_of('/');
  std::string b = (s == std::string::npos) ? p : p.substr(s + 1);
  size_t d = b.find_last_of('.');
  return d == std::string::npos ? b : b.substr(0, d);
}


// f(i) for i in [0, n) on up to 16 threads (SMILEHIP_IO_THREADS; measured on the 256-core host of an MI355X box, 8000 x 320 KB files on tmpfs:
// 8 threads 60 GB/s, 16 threads 76-108, 32 threads 42-126, 64 threads 25-40, 128 threads 13-28 -- more threads than that only contend) (file ingest and the per-file sinks: one open / read-or-write / close each, which a
// single thread spends most of its time waiting on); the first error message wins and is reported after the join
template <class F>
void parallel_for(size_t n, F f) {
  unsigned hw = std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) hw = (unsigned)CPU_COUNT(&set);      // cgroup / taskset limits
  static const unsigned cap = [] { const char *e = getenv("SMILEHIP_IO_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? (unsigned)v : 16u; }();
  const size_t nt = std::min<size_t>(n, std::min<unsigned>(cap, std::max(1u, hw)));
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; ++t)
    th.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
  for (auto &t : th) t.join();
}

struct Chunk {                                             // the files [j0, j1) of the list, read
  size_t j0 = 0, j1 = 0;
  std::vector<std::vector<unsigned char>> raw;             // per file (the general route)
  std::vector<WaveInfo> info;
  bool fast = false;                                       // every file 16-bit mono PCM at one rate: the samples are in `pcm` already,
  int16_t *pcm = nullptr;                                  // a page-locked buffer, utterance k = [true_off[k], true_off[k + 1])
  std::vector<int64_t> true_off;
  std::string err;
};

}  // namespace

// what --serve keeps from list to list: the device context (0.3 s of runtime start-up, code objects, first launch) and the plans
struct Persist {
  smilehip_context *ctx = nullptr;
  std::map<long, smilehip_plan *> plans;
  std::string key;                                         // the options the plans were built for
};

static int run_main(int argc, char **argv, Persist *ps) {
  std::map<std::string, std::string> opt, conf_cmdline;
  const char *with_value[] = {"--set", "-C", "-I", "-filelist", "-O", "-csvoutput", "-htkoutput", "-lldcsvoutput", "-lldhtkoutput",
                              "-instname", "-N", "-outdir", "--device", "--rank", "--world", "--chunk-files", "--master-addr", "--master-port"};
  bool with_conf = false, print_fingerprint = false, describe_only = false, gather = false;
  for (int i = 1; i < argc; ++i)
    if (!strcmp(argv[i], "-C")) with_conf = true;
  for (int i = 1; i < argc; ++i) {
    bool known = false;
    if (!strcmp(argv[i], "--fingerprint")) { print_fingerprint = true; continue; }
    if (!strcmp(argv[i], "--describe")) { describe_only = true; continue; }
    if (!strcmp(argv[i], "--gather")) { gather = true; continue; }
    for (const char *w : with_value)
      if (!strcmp(argv[i], w)) {
        if (i + 1 >= argc) die(std::string("option ") + w + " needs a value");
        opt[w] = argv[i + 1];
        known = true;
      }
    // with -C every option is also offered to the file's \cm[...] references (long or one-letter name), as SMILExtract does;
    // options the file does not define are refused after parsing
    if (with_conf && argv[i][0] == '-' && argv[i][1] != '-' && strcmp(argv[i], "-C")) {
      const bool has_val = i + 1 < argc && (argv[i + 1][0] != '-' || isdigit((unsigned char)argv[i + 1][1]) || argv[i + 1][1] == 0);
      conf_cmdline[argv[i] + 1] = has_val ? argv[i + 1] : "1";
      if (has_val) ++i;
      continue;
    }
    if (known) { ++i; continue; }
    die(std::string("unknown option ") + argv[i]);
  }
  // ---- -C file.conf: the plan comes from the configuration file (conf_plan.cpp)
  ConfPlan conf_plan;
  if (with_conf) {
    if (opt.count("--set")) die("-C and --set exclude each other");
    ConfFile cf;
    std::string cerr_;
    if (!conf_parse(opt["-C"], conf_cmdline, cf, cerr_)) die("-C " + opt["-C"] + ": " + cerr_);
    if (print_fingerprint) {
      printf("%016llx %016llx %016llx\n", (unsigned long long)conf_fingerprint(cf), (unsigned long long)conf_fingerprint_masked(cf),
             (unsigned long long)conf_fingerprint_masked2(cf));
      return 0;
    }
    static const char *builtin[] = {"l", "loglevel", "nologfile", "noconsoleoutput", "logfile", "appendLogfile", "t", "nticks", "d", "debug",
                                    "C", "configfile", "N", "instname", "filelist", "outdir"};
    // the \cm options this program acts on (with their one-letter forms); every other option the file defines is accepted only
    // with the file's own default value -- it would be parsed and then silently ignored otherwise (-start, -end, -arffoutput,
    // -timestampcsv, -appendcsv, -relation, -frameTimeAdd, ...)
    static const char *honoured[] = {"inputfile", "I", "output", "O", "csvoutput", "htkoutput", "lldcsvoutput", "D", "lldhtkoutput",
                                     "instname", "N"};
    for (const auto &kv : conf_cmdline) {
      bool ok = cf.cm_defaults.count(kv.first) != 0;
      std::string long_name = kv.first;
      for (const auto &sh : cf.cm_short) if (sh.second == kv.first) { ok = true; long_name = sh.first; }
      bool is_builtin = false, acted_on = false;
      for (const char *b : builtin) is_builtin = is_builtin || kv.first == b;
      for (const char *h : honoured) acted_on = acted_on || kv.first == h || long_name == h;
      if (!ok && !is_builtin) die("option -" + kv.first + " is not defined by " + opt["-C"]);
      if (is_builtin || acted_on) continue;
      const auto dflt = cf.cm_defaults.find(long_name);
      if (dflt == cf.cm_defaults.end() || dflt->second != kv.second)
        die("option -" + kv.first + " " + kv.second + ": " + opt["-C"] + " defines it, but smilextract_hip does not implement it (only its "
            "default" + (dflt != cf.cm_defaults.end() ? " '" + dflt->second + "'" : "") + "); implemented: -I -O -csvoutput -htkoutput "
            "-lldcsvoutput -lldhtkoutput -instname");
    }
    // an output option the file gives a default file name (MFCC12_0_D_A.conf: output(O){output.htk}) is written there, as the
    // reference does, unless the command line says otherwise
    {
      static const std::pair<const char *, const char *> outs[] = {{"output", "-O"}, {"csvoutput", "-csvoutput"}, {"htkoutput", "-htkoutput"},
                                                                    {"lldcsvoutput", "-lldcsvoutput"}, {"lldhtkoutput", "-lldhtkoutput"}};
      for (const auto &o : outs) {
        const auto dflt = cf.cm_defaults.find(o.first);
        if (dflt != cf.cm_defaults.end() && dflt->second != "?" && !dflt->second.empty() && !opt.count(o.second) && !opt.count("-filelist"))
          opt[o.second] = dflt->second;
      }
    }
    if (!conf_to_plan(cf, conf_plan, cerr_)) die("-C " + opt["-C"] + " cannot run on the fused path: " + cerr_);
    fprintf(stderr, "smilextract_hip: %s -> %s\n", opt["-C"].c_str(), conf_plan.describe.c_str());
    if (!conf_plan.preset.empty()) opt["--set"] = conf_plan.preset;
    if (describe_only) {                                   // what the file maps to, without touching a device
      const smilehip_lld_config &c = conf_plan.cfg;
      printf("preset=%s\n", conf_plan.preset.c_str());
      for (const auto &kv : conf_plan.f0_params) printf("param.%s=%.9g\n", kv.first.c_str(), kv.second);
      if (conf_plan.preset.empty()) {
        printf("chain_kind=%d\nframe_size_sec=%.17g\nframe_step_sec=%.17g\npreemph=%d\npreemph_k=%.9g\npreemph_de=%d\nwin_func=%d\n"
               "win_sigma=%.17g\nwin_gain=%.17g\nwin_offset=%.17g\nzero_pad_symmetric=%d\nn_bands=%d\nlofreq=%.9g\nhifreq=%.9g\n"
               "use_power=%d\nmel_htk_compatible=%d\nfirst_mfcc=%d\nlast_mfcc=%d\ncep_lifter=%.9g\nmfcc_htk_compatible=%d\nmelfloor=%.9g\n"
               "n_delta=%d\ndelta_win=%d\nplp_lp_order=%d\nplp_compression=%.9g\nappend_log_energy=%d\ncms=%d\nparm_kind=%d\n",
               c.chain_kind, c.frame_size_sec, c.frame_step_sec, c.preemph, (double)c.preemph_k, c.preemph_de, c.win_func, c.win_sigma,
               c.win_gain, c.win_offset, c.zero_pad_symmetric, c.n_bands, (double)c.lofreq, (double)c.hifreq, c.use_power,
               c.mel_htk_compatible, c.first_mfcc, c.last_mfcc, (double)c.cep_lifter, c.mfcc_htk_compatible, (double)c.melfloor, c.n_delta,
               c.delta_win, c.plp_lp_order, (double)c.plp_compression, c.append_log_energy, c.cms, conf_plan.parm_kind);
        printf("names=");
        for (size_t k = 0; k < conf_plan.lld_names.size(); ++k) printf("%s%s", k ? ";" : "", conf_plan.lld_names[k].c_str());
        printf("\n");
      }
      return 0;
    }
  }
  const bool free_chain = with_conf && conf_plan.preset.empty();
  const std::string set = opt.count("--set") ? opt["--set"] : "";
  const bool is09 = set == "is09_emotion";
  const bool is13 = set == "is13_compare";                     // config/is09-13/IS13_ComParE.conf: same elements, IS13 options
  const bool cmp16f = set == "compare16" || is13;              // the whole ComParE_2016.conf: LLD level + 6373 functionals
  const bool cmp16 = set == "compare16_lld" || cmp16f;
  // GeMAPSv01b.conf / eGeMAPSv01b.conf: sub-graphs of eGeMAPSv02.conf -- the v02 chain runs, the set's columns are written
  // GeMAPSv01a.conf / eGeMAPSv01a.conf: the same sub-graphs with three option values of openSMILE 2.2 (smilehip_config_egemapsv01a)
  const bool egm_v01a = set == "gemapsv01a" || set == "egemapsv01a";
  const bool egm_subset = set == "gemapsv01b" || set == "egemapsv01b" || egm_v01a;
  const bool egm = set == "egemapsv02" || egm_subset;            // config/egemaps/v02/eGeMAPSv02.conf
  std::vector<int> sel_lld = egemaps_subset_columns(set, false), sel_func = egemaps_subset_columns(set, true);
  // ComParE_2016 / IS13_ComParE files that lower lastMfcc or leave functional families out (conf_plan.hpp): the shipped graph runs, the
  // remaining outputs are written
  bool cmp_subset = false;
  if (with_conf && cmp16 && (conf_plan.last_mfcc > 0 || !conf_plan.func_enabled.empty())) {
    std::string e;
    if (!compare16_selection(is13, conf_plan.last_mfcc, conf_plan.func_enabled, sel_lld, sel_func, e)) die(e);
    cmp_subset = sel_lld.size() != 130 || sel_func.size() != 6373;
  }
  const bool out_subset = egm_subset || cmp_subset;
  if (out_subset && gather) die("--gather is not available with a set that writes a column selection (" + set + "): gather the whole vectors and select");
  const bool has_func = is09 || cmp16f || egm;
  // the eight files of config/mfcc and config/plp, by their names in lower case
  std::string variant;                             // upper-case config name for smilehip_config_htk_variant
  for (char ch : set) variant += (char)toupper((unsigned char)ch);
  smilehip_lld_config vcfg;
  if (free_chain) vcfg = conf_plan.cfg;
  const bool htk_variant = free_chain || (!is09 && !cmp16 && !egm && smilehip_config_htk_variant(&vcfg, variant.c_str()) == SMILEHIP_OK);
  const bool plp = htk_variant && vcfg.chain_kind == SMILEHIP_CHAIN_PLP;
  if (!is09 && !cmp16 && !egm && !htk_variant)
    die("--set must be mfcc12_{0,e}_d_a[_z], plp_{0,e}_d_a[_z], is09_emotion, compare16, compare16_lld, is13_compare, egemapsv02, gemapsv01b, egemapsv01b, gemapsv01a or egemapsv01a (or use -C file.conf)");
  // parmKind of the files' own cHtkSink sections (the _Z files); the others write through standard_data_output_lldonly (9)
  int parm_kind = 9;
  if (variant == "MFCC12_0_D_A_Z") parm_kind = 11014;
  else if (variant == "MFCC12_E_D_A_Z") parm_kind = 2886;
  else if (variant == "PLP_0_D_A_Z") parm_kind = 11019;
  else if (variant == "PLP_E_D_A_Z") parm_kind = 8971;
  if (free_chain) parm_kind = conf_plan.parm_kind;
  const bool lld_opts = is09 || cmp16 || egm;           // LLD files through -lldhtkoutput / -lldcsvoutput as in the reference
  std::string instname = opt.count("-instname") ? opt["-instname"] : (opt.count("-N") ? opt["-N"] : "unknown");

  std::vector<Job> jobs;
  if (opt.count("-I")) jobs.push_back({opt["-I"], instname});
  if (opt.count("-filelist")) {
    FILE *f = fopen(opt["-filelist"].c_str(), "r");
    if (!f) die("cannot open file list '" + opt["-filelist"] + "'");
    char line[8192];
    while (fgets(line, sizeof(line), f)) {
      std::string s(line);
      while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
      if (s.empty()) continue;
      const size_t tab = s.find('\t');
      jobs.push_back(tab == std::string::npos ? Job{s, basename_noext(s)} : Job{s.substr(0, tab), s.substr(tab + 1)});
    }
    fclose(f);
  }
  if (jobs.empty()) die("no input (-I or -filelist)");
  // one process per GPU: --rank / --world, or the RANK / WORLD_SIZE / LOCAL_RANK variables of torch.distributed.run
  auto env_int = [](const char *name, int dflt) { const char *v = getenv(name); return v && *v ? atoi(v) : dflt; };
  const int rank = opt.count("--rank") ? atoi(opt["--rank"].c_str()) : env_int("RANK", 0);
  const int world = opt.count("--world") ? atoi(opt["--world"].c_str()) : env_int("WORLD_SIZE", 1);
  if (world < 1 || rank < 0 || rank >= world) die("bad --rank/--world");
  const int device = opt.count("--device") ? atoi(opt["--device"].c_str()) : env_int("LOCAL_RANK", 0);
  const std::vector<Job> all_jobs = jobs;                  // rank 0 names the gathered rows
  // Sharding: longest-processing-time first on the files' sizes (a WAV's size is its frame count up to the header),
  // computed identically by every rank from the list alone -- no communication; within a rank the list order is kept.
  // shard[r] = list indices of rank r.
  std::vector<std::vector<size_t>> shard((size_t)world);
  if (world > 1) {
    std::vector<std::pair<long long, size_t>> by_size(jobs.size());
    for (size_t i = 0; i < jobs.size(); ++i) {
      struct stat st;
      by_size[i] = {stat(jobs[i].wav.c_str(), &st) == 0 ? (long long)st.st_size : 0LL, i};
    }
    std::stable_sort(by_size.begin(), by_size.end(), [](const std::pair<long long, size_t> &a, const std::pair<long long, size_t> &b) { return a.first > b.first; });
    std::vector<long long> load((size_t)world, 0);
    for (const auto &f : by_size) {
      size_t best = 0;
      for (size_t r = 1; r < (size_t)world; ++r) if (load[r] < load[best]) best = r;
      load[best] += f.first > 0 ? f.first : 1;
      shard[best].push_back(f.second);
    }
    for (auto &v : shard) std::sort(v.begin(), v.end());
    std::vector<Job> mine;
    for (size_t i : shard[(size_t)rank]) mine.push_back(jobs[i]);
    jobs.swap(mine);
  } else {
    for (size_t i = 0; i < jobs.size(); ++i) shard[0].push_back(i);
  }
  const bool list_mode = opt.count("-filelist") != 0;
  const std::string outdir = opt.count("-outdir") ? opt["-outdir"] : "";
  if (list_mode && outdir.empty() && (opt.count("-lldhtkoutput") || opt.count("-lldcsvoutput") || opt.count("-htkoutput") ||
                                       (!lld_opts && (opt.count("-O") || opt.count("-csvoutput")))))
    die("per-file outputs of a file list need -outdir (the file options then only switch the output on)");
  auto per_file = [&](const Job &j, const std::string &optname, const char *ext) {
    return list_mode ? outdir + "/" + basename_noext(j.wav) + ext : opt[optname];
  };
  if (list_mode && !outdir.empty()) {
    // per-file outputs are written by several threads: two list entries with the same base name (different directories) would
    // write the same path at the same time -- refused (over the WHOLE list: every rank sees the same answer)
    std::map<std::string, size_t> seen;
    for (size_t i = 0; i < all_jobs.size(); ++i) {
      const auto ins = seen.insert({basename_noext(all_jobs[i].wav), i});
      if (!ins.second)
        die("file list entries " + std::to_string(ins.first->second + 1) + " and " + std::to_string(i + 1) + " ('" + all_jobs[i].wav +
            "') have the same base name: their per-file outputs in -outdir would be the same file");
    }
  }

  // The header walk of EVERY file of the list starts now, beside the device's start-up (a quarter of a second during which the
  // host would otherwise idle); the chunks' ingest then only lays the samples out and reads them.
  std::vector<WaveInfo> all_info(jobs.size());
  std::vector<std::string> all_probe_err(jobs.size());
  const bool no_pinned = getenv("SMILEHIP_NO_PINNED") != nullptr;   // A/B switch: the pageable, serial route of round 3 (same files)
  std::shared_future<void> probed = std::async(std::launch::async, [&] {
    if (no_pinned) return;
    parallel_for(jobs.size(), [&](size_t k) { probe_wave_file(jobs[k].wav, all_info[k], all_probe_err[k]); });
  }).share();
  g_before_exit = [&] { probed.wait(); };
  smilehip_context *ctx = nullptr;
  if (ps && ps->ctx) ctx = ps->ctx;
  else check(smilehip_init(device, &ctx), "smilehip_init");
  if (ps) ps->ctx = ctx;
  // --gather: the summary rows of every rank travel to rank 0 over RCCL (libsmilehip_comm.so, loaded only here) and rank 0
  // writes ONE file in list order; without it every rank writes <name>.rank<r><ext>
  smilehip_comm *comm = nullptr;
  decltype(&smilehip_comm_create) comm_create = nullptr;
  decltype(&smilehip_comm_destroy) comm_destroy = nullptr;
  decltype(&smilehip_comm_allgather_count) comm_count = nullptr;
  decltype(&smilehip_comm_gather_rows) comm_gather = nullptr;
  decltype(&smilehip_comm_last_error) comm_error = nullptr;
  if (gather) {
    void *h = dlopen("libsmilehip_comm.so", RTLD_NOW);
    if (!h) die(std::string("--gather: ") + dlerror());
    comm_create = (decltype(comm_create))dlsym(h, "smilehip_comm_create");
    comm_destroy = (decltype(comm_destroy))dlsym(h, "smilehip_comm_destroy");
    comm_count = (decltype(comm_count))dlsym(h, "smilehip_comm_allgather_count");
    comm_gather = (decltype(comm_gather))dlsym(h, "smilehip_comm_gather_rows");
    comm_error = (decltype(comm_error))dlsym(h, "smilehip_comm_last_error");
    if (!comm_create || !comm_destroy || !comm_count || !comm_gather || !comm_error) die("--gather: libsmilehip_comm.so lacks a symbol");
    const char *ma = getenv("MASTER_ADDR"), *mp = getenv("MASTER_PORT");
    const std::string addr = opt.count("--master-addr") ? opt["--master-addr"] : (ma && *ma ? ma : "127.0.0.1");
    const int port = opt.count("--master-port") ? atoi(opt["--master-port"].c_str()) : (mp && *mp ? atoi(mp) + 1 : 29411);
    if (comm_create(device, rank, world, addr.c_str(), port, &comm) != 0) die(std::string("--gather: ") + comm_error());
  }
  std::vector<float> gathered;                              // this rank's summary rows: n_func values + a "has an instance" flag each
  int gathered_cols = 0;
  std::map<long, smilehip_plan *> own_plans;
  if (ps) {                                                // (--serve: the plans of the list before, if it asked for the same set / file)
    const std::string key = (opt.count("--set") ? opt["--set"] : "") + "|" + (opt.count("-C") ? opt["-C"] : "");
    if (key != ps->key) {
      for (auto &kv : ps->plans) smilehip_plan_destroy(kv.second);
      ps->plans.clear();
      ps->key = key;
    }
  }
  std::map<long, smilehip_plan *> &plans = ps ? ps->plans : own_plans;   // one plan per sample rate
  const long chunk_files_l = opt.count("--chunk-files") ? atol(opt["--chunk-files"].c_str()) : 256;
  if (chunk_files_l < 1) die("--chunk-files must be a positive number");
  const size_t chunk_files = (size_t)chunk_files_l;
  // Summary sinks (one row per file appended to ONE file) of several ranks must not share a file: each rank of a
  // --world > 1 run writes <name>.rank<r><ext>; concatenate them afterwards (the ARFF header is the same in each).
  auto summary_path = [&](const std::string &p) {
    if (world <= 1 || gather) return p;
    const size_t d = p.find_last_of('.'), sl = p.find_last_of('/');
    const std::string tag = ".rank" + std::to_string(rank);
    return (d == std::string::npos || (sl != std::string::npos && d < sl)) ? p + tag : p.substr(0, d) + tag + p.substr(d);
  };
  const std::vector<std::string> lld_names =
      free_chain ? conf_plan.lld_names
                 : (is09 ? lld_names_is09() : (cmp16 ? (cmp_subset ? select_names(lld_names_compare16(), sel_lld) : lld_names_compare16())
                    : (egm ? (egm_subset ? select_names(lld_names_egemaps(), sel_lld) : lld_names_egemaps())
                           : lld_names_htk_variant(plp, htk_variant && vcfg.append_log_energy))));
  const std::vector<std::string> fnames =
      is09 ? func_names_is09() : (cmp16f ? (cmp_subset ? select_names(func_names_compare16(), sel_func) : func_names_compare16())
           : (egm ? (egm_subset ? select_names(func_names_egemaps(), sel_func) : func_names_egemaps()) : std::vector<std::string>()));
  const uint32_t fmask = smilehip_functionals_is09_mask();
  std::string err;

  // ---- staging (round 5). The route is bound by the host, not the device (the MFCC kernel takes 0.4 ms per 1000 x 10 s files,
  // PCIe 6 ms): the samples are read straight into PAGE-LOCKED memory (probe the headers, lay the chunk out, pread every data
  // chunk to its place: no per-file vector, no packing copy), the device buffers are kept from chunk to chunk, the feature rows
  // come back into page-locked memory -- big-endian already when HTK files are all that is written (smilehip_htk_rows_be), so a
  // file's sink is one writev() -- and the sinks of chunk k run beside the device work of chunk k + 1 and the ingest of chunk
  // k + 2. SMILEHIP_NO_PINNED=1: the pageable, serial route of round 3 (A/B switch; same files).
  const bool timing = getenv("SMILEHIP_TIMING") != nullptr;
  struct HostBuf { void *p = nullptr; size_t cap = 0; };
  HostBuf pcm_slot[3], out_slot[2];                          // (three sample slots: chunk k + 2 is read while chunk k's copy-in may still be under way)
  auto host_reserve = [&](HostBuf &hb, size_t bytes) -> void * {
    if (bytes <= hb.cap) return hb.p;
    if (hb.p) smilehip_free_host(ctx, hb.p);
    hb.p = nullptr; hb.cap = 0;
    const size_t cap = bytes + bytes / 8 + 4096;
    if (smilehip_alloc_host(ctx, cap, &hb.p) != SMILEHIP_OK) return nullptr;
    hb.cap = cap;
    return hb.p;
  };
  HostBuf dev_pcm[2], dev_lld[2], dev_func[2];              // device buffers, kept and grown; two sets taken in turn by the chunks
  // (round 6) Chunk k's copy-in, kernels and copy-out go to stream k & 1 and the host does NOT wait for them: it reads and enqueues
  // chunk k + 1 first -- whose copy-in and kernels then run beside chunk k's copy-out, the link's two directions at once -- and only
  // then finishes chunk k (synchronise its stream, destroy its batch, start its sinks, write its summary rows). Chunks that are not
  // the plain case (one sample rate, 16-bit mono, page-locked slots) are finished at once, as before.
  // The KERNELS of consecutive chunks stay in order (the library's context-wide scratch -- the functionals' -- serves one run at
  // a time): chunk k + 1's first kernel waits for the event behind chunk k's last one; the copies either side do not.
  void *chunk_stream[2] = {nullptr, nullptr}, *chunk_computed[2] = {nullptr, nullptr};
  bool computed_recorded[2] = {false, false};
  if (!no_pinned && !getenv("SMILEHIP_E2E_SERIAL")) {
    for (void *&cs : chunk_stream) check(smilehip_stream_create(ctx, &cs), "smilehip_stream_create");
    for (void *&ce : chunk_computed) check(smilehip_event_create(ctx, &ce), "smilehip_event_create");
    check(smilehip_alloc_cache(ctx, (uint64_t)1 << 30), "smilehip_alloc_cache");   // a batch's blocks go to the next batch: no hipFree (= device-wide wait) per chunk
  }
  std::function<void()> pending;                            // what the previous chunk still has to do
  auto dev_reserve = [&](HostBuf &db, size_t bytes) -> void * {
    if (bytes <= db.cap && db.p) return db.p;
    if (db.p) smilehip_free(ctx, db.p);
    db.p = nullptr; db.cap = 0;
    const size_t cap = bytes + bytes / 8 + 4096;
    check(smilehip_alloc(ctx, cap, &db.p), "smilehip_alloc");
    db.cap = cap;
    return db.p;
  };
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  double t_wait_ingest = 0.0, t_device = 0.0, t_wait_sink = 0.0, t_dev_create = 0.0, t_dev_enqueue = 0.0, t_dev_reserve = 0.0, t_dev_sync = 0.0;
  std::atomic<int64_t> us_probe{0}, us_reserve{0}, us_read{0}, us_sink{0};   // inside the ingest / sink threads (SMILEHIP_TIMING)

  // ---- ingest: chunk k + 1 is read while chunk k is on the device and chunk k - 1 in the sinks (16-bit mono PCM is what the
  // fused kernels take)
  auto ingest = [&](size_t j0, size_t slot) {
    Chunk c;
    c.j0 = j0;
    c.j1 = std::min(jobs.size(), j0 + chunk_files);
    const size_t n = c.j1 - c.j0;
    c.raw.resize(n);
    c.info.resize(n);
    std::mutex m;
    auto fail1 = [&](const std::string &e) { std::lock_guard<std::mutex> g(m); if (c.err.empty()) c.err = e; };
    auto format_ok = [&](size_t k) {
      const WaveInfo &wi = c.info[k];
      // integer PCM of any width, or 32-bit IEEE float (the one float format the reference converts, smileUtil.c:2653-2662)
      const bool int_ok = wi.sample_type == 1 && wi.n_bps >= 1 && wi.n_bps <= 4;
      const bool float_ok = wi.sample_type == 3 && wi.n_bps == 4 && wi.n_bits == 32;
      if (wi.n_chan < 1 || !(int_ok || float_ok))
        fail1("'" + jobs[c.j0 + k].wav + "': integer PCM (8 / 16 / 24 / 32 bit) or 32-bit IEEE float, any number of channels, "
              "is what the reference converts (smilePcm_convertSamples / smilePcm_convertFloatSamples); this file is neither");
    };
    double ti = now();
    auto lap = [&](std::atomic<int64_t> &acc) { const double t = now(); acc += (int64_t)((t - ti) * 1e6); ti = t; };
    if (!no_pinned) {
      probed.wait();
      for (size_t k = 0; k < n; ++k) {
        if (!all_probe_err[c.j0 + k].empty()) { fail1(all_probe_err[c.j0 + k]); break; }
        c.info[k] = all_info[c.j0 + k];
        format_ok(k);
      }
      lap(us_probe);
      if (!c.err.empty()) return c;
      c.fast = n > 0;
      for (size_t k = 0; k < n && c.fast; ++k) {
        const WaveInfo &wi = c.info[k];
        c.fast = wi.sample_type == 1 && wi.n_bps == 2 && wi.n_chan == 1 && wi.sample_rate == c.info[0].sample_rate;
      }
    }
    if (c.fast) {
      c.true_off.assign(n + 1, 0);
      for (size_t k = 0; k < n; ++k) c.true_off[k + 1] = c.true_off[k] + (int64_t)c.info[k].n_blocks;
      int16_t *pcm = static_cast<int16_t *>(host_reserve(pcm_slot[slot], ((size_t)c.true_off[n] + 2) * 2));
      if (!pcm) { c.err = std::string("page-locked staging buffer: ") + smilehip_last_error(); return c; }
      c.pcm = pcm;
      pcm[c.true_off[n]] = 0; pcm[c.true_off[n] + 1] = 0;
      lap(us_reserve);
      parallel_for(n, [&](size_t k) {
        std::string e;
        const long want = c.info[k].n_blocks;
        if (want <= 0) return;
        const long got = read_wave_data(jobs[c.j0 + k].wav, c.info[k], pcm + c.true_off[k], (size_t)want * 2, e);
        if (got < 0) fail1(e);
        else if (got != want) fail1("'" + jobs[c.j0 + k].wav + "' changed while it was read");
      });
      lap(us_read);
      return c;
    }
    parallel_for(n, [&](size_t k) {
      std::string e;
      if (!read_wave_file(jobs[c.j0 + k].wav, c.info[k], c.raw[k], e)) { fail1(e); return; }
      format_ok(k);
    });
    return c;
  };
  // what a chunk's per-file sinks need once the device is done with it (they run beside the next chunk's device work)
  struct SinkWork {
    std::vector<size_t> idx;
    std::vector<int64_t> row_off, true_off;
    std::vector<float> func, lld_vec;
    const float *lld = nullptr;
    int n_out = 0, n_func = 0;
    bool be = false;
    smilehip_geometry g;
    smilehip_plan *plan = nullptr;
    std::string err;
  };
  std::future<std::string> sink_fut[2];
  auto sink_wait = [&](size_t slot) {
    if (!sink_fut[slot].valid()) return;
    const double t0 = now();
    const std::string e = sink_fut[slot].get();
    t_wait_sink += now() - t0;
    if (!e.empty()) die(e);
  };
  std::future<Chunk> ahead = std::async(std::launch::async, ingest, (size_t)0, (size_t)0);
  g_before_exit = [&] {
    probed.wait();
    if (ahead.valid()) ahead.wait();
    for (auto &f : sink_fut) if (f.valid()) f.wait();
  };
  auto make_plan = [&](long rate) -> smilehip_plan * {
    smilehip_plan *&plan = plans[rate];
    if (!plan) {
      smilehip_lld_config cfg;
      if (is09) smilehip_config_is09_lld(&cfg);
      else if (is13) smilehip_config_is13_compare(&cfg);
      else if (cmp16) smilehip_config_compare16(&cfg);
      else if (egm_v01a) smilehip_config_egemapsv01a(&cfg);
      else if (egm) smilehip_config_egemapsv02(&cfg);
      else cfg = vcfg;
      if (with_conf) conf_apply_f0_params(conf_plan, cfg);     // an edited big-set file: its own pitch range, harmonics, buffer ...
      cfg.sample_rate = (double)rate;
      check(smilehip_plan_create(ctx, &cfg, &plan), "smilehip_plan_create");
    }
    return plan;
  };
  // (the first file's rate is almost always every file's: its plan -- tables, uploads -- is built while the first chunk is read)
  if (!no_pinned && !jobs.empty()) { probed.wait(); if (all_probe_err[0].empty() && all_info[0].sample_rate > 0) make_plan(all_info[0].sample_rate); }
  size_t chunk_no = 0;
  for (size_t j0 = 0; j0 < jobs.size(); j0 += chunk_files, ++chunk_no) {
    const size_t j1 = std::min(jobs.size(), j0 + chunk_files);
    const size_t slot = chunk_no & 1;
    double t0 = now();
    Chunk chunk = ahead.get();
    t_wait_ingest += now() - t0;
    if (j1 < jobs.size()) ahead = std::async(std::launch::async, ingest, j1, (chunk_no + 1) % 3);
    if (!chunk.err.empty()) die(chunk.err);
    t0 = now();
    std::map<long, std::vector<size_t>> by_rate;
    std::vector<std::vector<unsigned char>> &raw = chunk.raw;
    for (size_t j = j0; j < j1; ++j) by_rate[chunk.info[j - j0].sample_rate].push_back(j);
    // per job of the chunk; empty = no instance (no frame). On the heap: the chunk may be finished an iteration later.
    std::shared_ptr<std::vector<std::vector<float>>> func_rows_p = std::make_shared<std::vector<std::vector<float>>>(j1 - j0);
    std::vector<std::vector<float>> &func_rows = *func_rows_p;
    std::vector<std::function<void()>> finish_groups;       // one per rate group
    const bool defer = chunk_stream[0] != nullptr && chunk.fast && by_rate.size() == 1;
    void *const st = defer ? chunk_stream[slot] : nullptr;
    if (!defer && pending) { pending(); pending = nullptr; }   // (a chunk on the null stream: the one before it is finished first)
    for (auto &kv : by_rate) {
      smilehip_plan *plan = make_plan(kv.first);
      std::shared_ptr<SinkWork> w = std::make_shared<SinkWork>();
      check(smilehip_plan_geometry(plan, &w->g), "smilehip_plan_geometry");
      const smilehip_geometry &g = w->g;
      w->plan = plan;
      w->idx = kv.second;
      const std::vector<size_t> &idx = w->idx;
      // exact packing: utterance u = samples [off[u], off[u+1]) of one buffer (the kernels use
      // dword PCM loads when every offset is even, 16-bit loads otherwise). 16-bit mono files go to the device as they are
      // (the kernels convert at the load); any other integer format / channel count and IEEE float are converted on the device by
      // smilehip_pcm_convert / smilehip_pcm_convert_float -- cWaveSource's monoMixdown = 1 of every shipped file (standard_wave_input.conf.inc) -- and the chain
      // reads floats (smilehip_lld_run_f32)
      bool all_s16_mono = true;
      for (size_t i = 0; i < idx.size(); ++i) {
        const WaveInfo &wi = chunk.info[idx[i] - j0];
        all_s16_mono = all_s16_mono && wi.sample_type == 1 && wi.n_bps == 2 && wi.n_chan == 1;
      }
      std::vector<int64_t> &true_off = w->true_off;
      std::vector<int16_t> pcm_vec;
      const int16_t *pcm_host = nullptr;
      if (chunk.fast) {                                   // (one rate group: the chunk itself, in list order)
        true_off = chunk.true_off;
        pcm_host = chunk.pcm;
      } else {
        true_off.assign(idx.size() + 1, 0);
        for (size_t i = 0; i < idx.size(); ++i) {
          const WaveInfo &wi = chunk.info[idx[i] - j0];
          true_off[i + 1] = true_off[i] + (int64_t)(raw[idx[i] - j0].size() / (size_t)(wi.n_bps * wi.n_chan));
        }
        pcm_vec.assign(all_s16_mono ? (size_t)true_off.back() + 2 : 2, 0);
        for (size_t i = 0; i < idx.size() && all_s16_mono; ++i) {
          const auto &r = raw[idx[i] - j0];
          if (!r.empty()) std::memcpy(&pcm_vec[(size_t)true_off[i]], r.data(), r.size() & ~(size_t)1);
        }
        pcm_host = pcm_vec.data();
      }
      smilehip_batch *b = nullptr;
      double td = now();
      auto dlap = [&](double &acc) { const double t = now(); acc += t - td; td = t; };
      check(smilehip_batch_create(plan, true_off.data(), (int32_t)idx.size(), &b), "smilehip_batch_create");
      dlap(t_dev_create);
      const int64_t rows = smilehip_batch_total_rows(b);
      std::vector<int64_t> &row_off = w->row_off;
      row_off.assign(idx.size() + 1, 0);
      check(smilehip_batch_frame_offsets(b, row_off.data()), "smilehip_batch_frame_offsets");
      const int n_out = g.n_out;
      w->n_out = n_out;
      void *d_func = nullptr;
      const uint64_t pcm_bytes = (uint64_t)std::max<int64_t>(true_off.back(), 2) * 2;
      void *d_pcm = dev_reserve(dev_pcm[slot], pcm_bytes);
      void *d_lld = dev_reserve(dev_lld[slot], (uint64_t)std::max<int64_t>(rows, 1) * n_out * 4);
      void *d_f32 = nullptr;
      if (all_s16_mono) {
        check(smilehip_copy_to_device(ctx, d_pcm, pcm_host, (uint64_t)true_off.back() * 2, st), "copy_to_device");
        if (defer && computed_recorded[slot ^ 1]) check(smilehip_stream_wait_event(ctx, st, chunk_computed[slot ^ 1]), "stream_wait_event");
        check(smilehip_lld_run(plan, b, (const int16_t *)d_pcm, (float *)d_lld, n_out, st), "smilehip_lld_run");
      } else {
        check(smilehip_alloc(ctx, (uint64_t)std::max<int64_t>(true_off.back(), 1) * 4, &d_f32), "smilehip_alloc");
        size_t biggest = 1;
        for (size_t i = 0; i < idx.size(); ++i) biggest = std::max(biggest, raw[idx[i] - j0].size());
        void *d_raw = nullptr;
        check(smilehip_alloc(ctx, biggest, &d_raw), "smilehip_alloc");
        for (size_t i = 0; i < idx.size(); ++i) {
          const WaveInfo &wi = chunk.info[idx[i] - j0];
          const auto &r = raw[idx[i] - j0];
          const int64_t n = true_off[i + 1] - true_off[i];
          if (n <= 0) continue;
          check(smilehip_copy_to_device(ctx, d_raw, r.data(), (uint64_t)n * wi.n_bps * wi.n_chan, nullptr), "copy_to_device");
          if (wi.sample_type == 3)
            check(smilehip_pcm_convert_float(ctx, (const float *)d_raw, wi.n_chan, 1, n, (float *)d_f32 + true_off[i], nullptr), "smilehip_pcm_convert_float");
          else
            check(smilehip_pcm_convert(ctx, d_raw, wi.n_bps, wi.n_bits, wi.n_chan, 1, n, (float *)d_f32 + true_off[i], nullptr), "smilehip_pcm_convert");
          check(smilehip_stream_synchronize(ctx, nullptr), "stream_synchronize");      // d_raw is reused by the next file
        }
        smilehip_free(ctx, d_raw);
        check(smilehip_lld_run_f32(plan, b, (const float *)d_f32, (float *)d_lld, n_out, nullptr), "smilehip_lld_run_f32");
      }
      const int n_func = is09 ? n_out * smilehip_functionals_count(fmask)
                              : (cmp16f ? smilehip_functionals_compare16_count() : (egm ? smilehip_functionals_egemaps_count() : 0));
      w->n_func = n_func;
      std::vector<float> &func = w->func;
      if (has_func) {
        d_func = dev_reserve(dev_func[slot], (uint64_t)idx.size() * n_func * 4);
        if (is09)
          check(smilehip_batch_functionals(plan, b, (const float *)d_lld, n_out, fmask, (float *)d_func, n_func, st),
                "smilehip_batch_functionals");
        else if (egm)
          check(smilehip_batch_functionals_egemaps(plan, b, (float *)d_func, n_func, st), "smilehip_batch_functionals_egemaps");
        else
          check((is13 ? smilehip_batch_functionals_is13_compare : smilehip_batch_functionals_compare16)(
                    plan, b, (const float *)d_lld, n_out, (float *)d_func, n_func, st),
                "smilehip_batch_functionals_compare16");
        func.resize(idx.size() * (size_t)n_func);
        check(smilehip_copy_to_host(ctx, func.data(), d_func, (uint64_t)func.size() * 4, st), "copy_to_host");
      }
      const std::string lld_htk_opt = lld_opts ? "-lldhtkoutput" : "-O", lld_csv_opt = lld_opts ? "-lldcsvoutput" : "-csvoutput";
      const bool want_lld_htk = opt.count(lld_htk_opt) && opt[lld_htk_opt] != "?";
      const bool want_lld_csv = opt.count(lld_csv_opt) && opt[lld_csv_opt] != "?";
      // the feature rows: into this chunk's page-locked slot (its previous user's sinks are awaited first); big-endian on the
      // device when HTK files are their only reader
      dlap(t_dev_enqueue);
      sink_wait(slot);
      td = now();
      const size_t lld_bytes = (size_t)std::max<int64_t>(rows, 1) * n_out * 4;
      float *lld = nullptr;
      if (!no_pinned && by_rate.size() == 1) lld = static_cast<float *>(host_reserve(out_slot[slot], lld_bytes));
      if (!lld) { w->lld_vec.resize(lld_bytes / 4); lld = w->lld_vec.data(); }
      w->lld = lld;
      dlap(t_dev_reserve);
      w->be = !no_pinned && want_lld_htk && !want_lld_csv && !out_subset && rows > 0;
      if (w->be) check(smilehip_htk_rows_be(ctx, (const float *)d_lld, rows * n_out, d_lld, st), "smilehip_htk_rows_be");
      if (defer) { check(smilehip_event_record(ctx, chunk_computed[slot], st), "event_record"); computed_recorded[slot] = true; }
      if (rows > 0) check(smilehip_copy_to_host(ctx, lld, d_lld, (uint64_t)rows * n_out * 4, st), "copy_to_host");
      dlap(t_dev_enqueue);
      // ---- from here on: what the chunk does once its stream has drained (at once, or after the next chunk has been enqueued).
      // Everything of this iteration it needs is taken by value.
      const bool one_group = by_rate.size() == 1;
      finish_groups.push_back([&, w, b, st, d_f32, n_func, j0, slot, func_rows_p, want_lld_htk, want_lld_csv, lld_htk_opt, lld_csv_opt, one_group]() {
      std::vector<std::vector<float>> &func_rows = *func_rows_p;
      const std::vector<size_t> &idx = w->idx;
      const std::vector<int64_t> &row_off = w->row_off;
      const std::vector<float> &func = w->func;
      double td = now();
      auto dlap = [&](double &acc) { const double t = now(); acc += t - td; td = t; };
      check(smilehip_stream_synchronize(ctx, st), "stream_synchronize");
      dlap(t_dev_sync);
      if (d_f32) smilehip_free(ctx, d_f32);
      smilehip_batch_destroy(b);
      dlap(t_dev_create);
      // the summary rows of the chunk (appended in list order below)
      for (size_t i = 0; i < idx.size() && has_func; ++i) {
        if (row_off[i + 1] - row_off[i] <= 0) continue;     // no frame -> the reference writes no instance
        const float *fv = func.data() + i * (size_t)n_func;
        if (out_subset) func_rows[idx[i] - j0] = select_columns(fv, 1, n_func, sel_func);
        else func_rows[idx[i] - j0].assign(fv, fv + n_func);
      }
      // ---- sinks: the per-file outputs of the chunk on up to 16 threads (different files), beside the next chunk's device work
      auto sinks = [&, w, want_lld_htk, want_lld_csv, lld_htk_opt, lld_csv_opt]() -> std::string {
        std::mutex sink_m;
        std::string sink_err;
        const double ts0 = now();
        struct Lap { std::atomic<int64_t> &a; double t0; std::function<double()> nowf; ~Lap() { a += (int64_t)((nowf() - t0) * 1e6); } } lap_sink{us_sink, ts0, now};
        const std::vector<size_t> &idx = w->idx;
        const int n_out = w->n_out, n_func = w->n_func;
        const smilehip_geometry &g = w->g;
        smilehip_plan *plan = w->plan;
        parallel_for(idx.size(), [&](size_t i) {
          std::string err;                                  // (shadows the function's: one per thread)
          auto die = [&](const std::string &m) { std::lock_guard<std::mutex> gd(sink_m); if (sink_err.empty()) sink_err = m; };   // (the caller returns: no further sink of this file)
          const Job &job = jobs[idx[i]];
          const float *x = w->lld + (size_t)w->row_off[i] * n_out;
          const int64_t r = w->row_off[i + 1] - w->row_off[i];
          int n_w = n_out;                                  // columns written (a subset preset writes its selection)
          std::vector<float> x_sel;
          if (out_subset) { x_sel = select_columns(x, r, n_out, sel_lld); x = x_sel.data(); n_w = (int)sel_lld.size(); }
          if (want_lld_htk) {
            const std::string path = per_file(job, lld_htk_opt, lld_opts ? ".lld.htk" : ".htk");
            const bool ok = w->be ? write_htk_be(path, x, r, n_w, g.frame_period, lld_opts ? 9 : parm_kind, err)
                                  : write_htk(path, x, r, n_w, n_w, g.frame_period, lld_opts ? 9 : parm_kind, err);
            if (!ok) { die(err); return; }
          }
          if (want_lld_csv) {
            CsvOptions co;
            co.instance_name = job.inst;
            // rows of the ComParE level follow the 60 ms framer: T60 + 1
            const int64_t n_frames = (cmp16 || egm) ? r - 1 : smilehip_num_frames(plan, w->true_off[i + 1] - w->true_off[i]);
            std::vector<double> times((size_t)r);
            for (int64_t t = 0; t < r; ++t) times[(size_t)t] = smilehip_row_time(plan, n_frames, t);
            if (!write_csv(per_file(job, lld_csv_opt, lld_opts ? ".lld.csv" : ".csv"), lld_names, x, r, n_w, n_w, g.frame_period,
                           times.data(), co, err)) {
              die(err);
              return;
            }
          }
          if (has_func && r > 0) {                          // no frame -> the reference writes no instance
            const float *fv = w->func.data() + i * (size_t)n_func;
            int n_fw = n_func;
            std::vector<float> f_sel;
            if (out_subset) { f_sel = select_columns(fv, 1, n_func, sel_func); fv = f_sel.data(); n_fw = (int)sel_func.size(); }
            if (opt.count("-htkoutput") && opt.at("-htkoutput") != "?")
              if (!write_htk(per_file(job, "-htkoutput", ".func.htk"), fv, 1, n_fw, n_fw, 0.0, 9, err)) { die(err); return; }
          }
        });
        return sink_err;
      };
      if (!no_pinned && one_group) {
        sink_fut[slot] = std::async(std::launch::async, sinks);
      } else {
        const std::string e = sinks();
        if (!e.empty()) die(e);
      }
      });
      if (!defer) finish_groups.back()();
    }
    t_device += now() - t0;
    auto finish_chunk = [&, j0, j1, func_rows_p, finish_groups, defer]() {
    std::vector<std::vector<float>> &func_rows = *func_rows_p;
    const double tf0 = now();
    if (defer) for (const auto &fg : finish_groups) fg();
    // summary sinks in file-list order (the rate groups above may have processed the chunk's files in another order)
    for (size_t j = j0; j < j1 && has_func && gather; ++j) {       // kept for the gather at the end
      const std::vector<float> &fv = func_rows[j - j0];
      const int n_func = is09 ? (int)lld_names.size() * smilehip_functionals_count(fmask)
                              : (cmp16f ? smilehip_functionals_compare16_count() : smilehip_functionals_egemaps_count());
      gathered_cols = n_func + 1;
      const size_t at = gathered.size();
      gathered.resize(at + (size_t)gathered_cols, 0.0f);
      if (!fv.empty()) { std::copy(fv.begin(), fv.end(), gathered.begin() + (long)at); gathered[at + (size_t)n_func] = 1.0f; }
    }
    for (size_t j = j0; j < j1 && has_func && !gather; ++j) {
      const std::vector<float> &fv = func_rows[j - j0];
      if (fv.empty()) continue;
      const int n_func = (int)fv.size();
      if (opt.count("-O") && opt["-O"] != "?") {
        ArffOptions ao;
        ao.instance_name = jobs[j].inst;
        if (!write_arff(summary_path(opt["-O"]), fnames, fv.data(), 1, n_func, n_func, 0.0, ao, err)) die(err);
      }
      if (opt.count("-csvoutput") && opt["-csvoutput"] != "?") {
        CsvOptions co;
        co.instance_name = jobs[j].inst;
        co.append = true;
        if (!write_csv(summary_path(opt["-csvoutput"]), fnames, fv.data(), 1, n_func, n_func, 0.0, nullptr, co, err)) die(err);
      }
    }
    t_device += now() - tf0;
    };
    // the chunk before this one is finished now that this one's work is enqueued behind it; this one waits for the next
    if (pending) { pending(); pending = nullptr; }
    if (defer) pending = finish_chunk;
    else finish_chunk();
  }
  if (pending) { pending(); pending = nullptr; }
  for (size_t s2 = 0; s2 < 2; ++s2) sink_wait(s2);
  if (timing)
    fprintf(stderr, "smilextract_hip timing: files %zu, chunks %zu; since the first ingest %.3f s: waiting for ingest %.3f, device stage (pack, copies, "
            "kernels, summary rows) %.3f, waiting for sinks %.3f; inside the ingest thread: header probes %.3f, staging buffer %.3f, sample reads %.3f; "
            "inside the sink threads %.3f; of the device stage: batch create / destroy %.3f, enqueue (buffers, copy in, kernels) %.3f, output buffer %.3f, "
            "copy out + synchronize %.3f\n", jobs.size(), chunk_no, now() - t_start, t_wait_ingest, t_device, t_wait_sink,
            us_probe.load() * 1e-6, us_reserve.load() * 1e-6, us_read.load() * 1e-6, us_sink.load() * 1e-6, t_dev_create, t_dev_enqueue,
            t_dev_reserve, t_dev_sync);
  for (HostBuf *hb : {&pcm_slot[0], &pcm_slot[1], &pcm_slot[2], &out_slot[0], &out_slot[1]}) if (hb->p) smilehip_free_host(ctx, hb->p);
  for (HostBuf *db : {&dev_pcm[0], &dev_pcm[1], &dev_lld[0], &dev_lld[1], &dev_func[0], &dev_func[1]}) if (db->p) smilehip_free(ctx, db->p);
  for (void *cs : chunk_stream) if (cs) smilehip_stream_destroy(ctx, cs);
  for (void *ce : chunk_computed) if (ce) smilehip_event_destroy(ctx, ce);
  if (gather && has_func) {
    // rank r holds the rows of files r, r + world, ...: counts to everyone, rows to rank 0, rank 0 writes in list order
    std::vector<int64_t> counts((size_t)world, 0);
    if (comm_count(comm, (int64_t)jobs.size(), counts.data(), nullptr) != 0) die(std::string("--gather: ") + comm_error());
    int cols = gathered_cols;
    if (cols == 0) cols = 1 + (is09 ? (int)lld_names.size() * smilehip_functionals_count(fmask)
                                    : (cmp16f ? smilehip_functionals_compare16_count() : smilehip_functionals_egemaps_count()));
    int64_t total = 0;
    for (int64_t c : counts) total += c;
    void *d_mine = nullptr, *d_all = nullptr;
    check(smilehip_alloc(ctx, std::max<uint64_t>(gathered.size(), 1) * 4, &d_mine), "smilehip_alloc");
    if (!gathered.empty()) check(smilehip_copy_to_device(ctx, d_mine, gathered.data(), (uint64_t)gathered.size() * 4, nullptr), "copy_to_device");
    if (rank == 0) check(smilehip_alloc(ctx, (uint64_t)std::max<int64_t>(total, 1) * cols * 4, &d_all), "smilehip_alloc");
    check(smilehip_stream_synchronize(ctx, nullptr), "stream_synchronize");
    if (comm_gather(comm, (const float *)d_mine, counts.data(), cols, (float *)d_all, nullptr) != 0) die(std::string("--gather: ") + comm_error());
    if (rank == 0) {
      std::vector<float> all((size_t)total * cols);
      if (total > 0) check(smilehip_copy_to_host(ctx, all.data(), d_all, (uint64_t)all.size() * 4, nullptr), "copy_to_host");
      check(smilehip_stream_synchronize(ctx, nullptr), "stream_synchronize");
      std::vector<int64_t> first((size_t)world + 1, 0);
      for (int r = 0; r < world; ++r) first[(size_t)r + 1] = first[(size_t)r] + counts[(size_t)r];
      const int n_func = cols - 1;
      std::vector<std::pair<int, int64_t>> where(all_jobs.size());      // list index -> (rank, position in that rank's rows)
      for (int r = 0; r < world; ++r) {
        if ((int64_t)shard[(size_t)r].size() != counts[(size_t)r]) die("--gather: the ranks disagree about the file list");
        for (size_t k = 0; k < shard[(size_t)r].size(); ++k) where[shard[(size_t)r][k]] = {r, (int64_t)k};
      }
      for (size_t j = 0; j < all_jobs.size(); ++j) {
        const int r = where[j].first;
        const int64_t k = where[j].second;
        const float *fv = all.data() + (size_t)(first[(size_t)r] + k) * cols;
        if (fv[n_func] == 0.0f) continue;                   // no frame -> the reference writes no instance
        if (opt.count("-O") && opt["-O"] != "?") {
          ArffOptions ao;
          ao.instance_name = all_jobs[j].inst;
          if (!write_arff(opt["-O"], fnames, fv, 1, n_func, n_func, 0.0, ao, err)) die(err);
        }
        if (opt.count("-csvoutput") && opt["-csvoutput"] != "?") {
          CsvOptions co;
          co.instance_name = all_jobs[j].inst;
          co.append = true;
          if (!write_csv(opt["-csvoutput"], fnames, fv, 1, n_func, n_func, 0.0, nullptr, co, err)) die(err);
        }
      }
      smilehip_free(ctx, d_all);
    } else {
      check(smilehip_stream_synchronize(ctx, nullptr), "stream_synchronize");
    }
    smilehip_free(ctx, d_mine);
  }
  if (comm) comm_destroy(comm);
  if (!ps) {
    for (auto &kv : plans) smilehip_plan_destroy(kv.second);
    smilehip_shutdown(ctx);
  }
  g_before_exit = nullptr;
  return 0;
}

// smilextract_hip [options] --serve: the options on the command line are every list's; then one line per list on standard input --
// more options, blank-separated (typically `-filelist <file> -outdir <dir>`; no blanks inside a value) -- each run like a command of
// its own, on the SAME process, device context and plans: what a list costs is its files, not the 0.3 s of start-up. After a list:
// one line `done <files' exit status> <seconds>` on standard output. An error in a list ends the server (as it ends the command).
int main(int argc, char **argv) {
  bool serve = false;
  std::vector<char *> base;
  for (int i = 0; i < argc; ++i) {
    if (i > 0 && !strcmp(argv[i], "--serve")) { serve = true; continue; }
    base.push_back(argv[i]);
  }
  if (!serve) return run_main(argc, argv, nullptr);
  Persist ps;
  char *line = nullptr;
  size_t cap = 0;
  while (getline(&line, &cap, stdin) > 0) {
    std::vector<std::string> tok;
    std::string cur;
    for (const char *c = line; *c; ++c) {
      if (*c == ' ' || *c == '\t' || *c == '\n' || *c == '\r') { if (!cur.empty()) { tok.push_back(cur); cur.clear(); } }
      else cur += *c;
    }
    if (!cur.empty()) tok.push_back(cur);
    if (tok.empty()) continue;
    if (tok[0] == "quit") break;
    std::vector<char *> av = base;
    for (std::string &t : tok) av.push_back(&t[0]);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = run_main((int)av.size(), av.data(), &ps);
    printf("done %d %.4f\n", rc, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    fflush(stdout);
  }
  free(line);
  for (auto &kv : ps.plans) smilehip_plan_destroy(kv.second);
  if (ps.ctx) smilehip_shutdown(ps.ctx);
  return 0;
}
