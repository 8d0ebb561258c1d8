// Plans from openSMILE configuration files (smilextract_hip -C file.conf): an own reader of the reference's config
// format and a symbolic walk over the component graph that either yields a smilehip_lld_config (+ column names) or says
// which component / option of the graph the fused path cannot express. No reference code runs here.
//
// Format, as the reference's cFileConfigReader reads it (src/core/configManager.cpp:1744-2060 file level, :2170-2560
// instance level): sections "[instance:cType]" (a section may be opened several times), "field = value" lines (the
// instance name may prefix the field), comments = lines starting with ; # % // and everything after // on a line,
// C-style /* */ blocks, "\{path}" includes (tried as given, relative to the top-level file, relative to the including
// file), "\cm[long(short){default}:description]" command-line placeholders (also inside an include path) and
// "\cm[long]" references to an option defined earlier. The component list is the instance[NAME].type = cType entries
// of the [componentInstances:cComponentManager] sections.
#include "conf_plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>

namespace smilehip_host {

namespace {

std::string trim(const std::string &s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) --b;
  return s.substr(a, b - a);
}
std::string dir_of(const std::string &p) {
  const size_t s = p.find_last_of('/');
  return s == std::string::npos ? std::string(".") : p.substr(0, s);
}
bool file_exists(const std::string &p) {
  std::ifstream f(p);
  return f.good();
}
std::string lower(std::string s) {
  for (char &c : s) c = (char)tolower((unsigned char)c);
  return s;
}

struct Parser {
  ConfFile &out;
  const std::map<std::string, std::string> &cmdline;
  std::string top_dir, err;
  int depth = 0;
  std::string cur_name, cur_type;      // the open section: an include is textual, it continues (and may change) the section
  const ConfCmValue *cm_value = nullptr;   // conf_from_sections: the host's own command-line parser answers (effective value of an option)

  // "\cm[long(short){dflt}:descr]trailing" -> value of the option (command line, else default) + trailing
  bool resolve_cm(const std::string &v, std::string &res) {
    if (v.compare(0, 4, "\\cm[") != 0) { res = v; return true; }
    const size_t end = v.find(']');
    if (end == std::string::npos) { err = "missing ']' in command-line reference '" + v + "'"; return false; }
    const std::string body = v.substr(4, end - 4), trailing = v.substr(end + 1);
    std::string name = body, dflt;
    bool has_dflt = false;
    const size_t colon = body.find(':');
    std::string head = colon == std::string::npos ? body : body.substr(0, colon);
    const size_t br = head.find('{');
    if (br != std::string::npos) {
      const size_t be = head.find('}', br);
      if (be == std::string::npos) { err = "missing '}' in command-line reference '" + v + "'"; return false; }
      dflt = head.substr(br + 1, be - br - 1);
      has_dflt = true;
      head = head.substr(0, br);
    }
    const size_t par = head.find('(');
    name = trim(par == std::string::npos ? head : head.substr(0, par));
    std::string short_name;
    if (par != std::string::npos) {
      const size_t pe = head.find(')', par);
      if (pe != std::string::npos) short_name = trim(head.substr(par + 1, pe - par - 1));
    }
    if (!short_name.empty()) out.cm_short[name] = short_name;
    if (has_dflt && !out.cm_defaults.count(name)) out.cm_defaults[name] = dflt;
    std::string host_val;
    if (cm_value && *cm_value && (*cm_value)(name, host_val)) { res = host_val + trailing; out.cm_used.insert(name); return true; }
    auto it = cmdline.find(name);
    if (it == cmdline.end() && out.cm_short.count(name)) it = cmdline.find(out.cm_short[name]);
    if (it != cmdline.end()) res = it->second + trailing;
    else if (out.cm_defaults.count(name)) res = out.cm_defaults[name] + trailing;
    else { err = "command-line option '" + name + "' is referenced before it is defined"; return false; }
    out.cm_used.insert(name);
    return true;
  }

  ConfInstance &instance(const std::string &name, const std::string &type) {
    for (ConfInstance &i : out.inst)
      if (i.name == name) {
        if (i.type.empty()) i.type = type;
        return i;
      }
    out.inst.push_back(ConfInstance{name, type, {}, false});
    return out.inst.back();
  }

  // one line of a section body or a section header (comments at the start of the line and includes are the caller's business)
  bool body_line(std::string l, const std::string &real) {
      const size_t eol = l.find("//");
      if (eol != std::string::npos) l = trim(l.substr(0, eol));
      if (l.empty()) return true;
      if (l[0] == '[') {
        const size_t c = l.find(':'), e = l.find(']');
        if (c == std::string::npos || e == std::string::npos || c > e) { err = "bad section header '" + l + "' in " + real; return false; }
        cur_name = trim(l.substr(1, c - 1));
        cur_type = trim(l.substr(c + 1, e - c - 1));
        if (cur_type != "cComponentManager") instance(cur_name, cur_type).has_section = true;
        return true;
      }
      const size_t eq = l.find('=');
      if (eq == std::string::npos) { err = "missing '=' in line '" + l + "' of " + real; return false; }
      std::string key = trim(l.substr(0, eq)), val;
      if (!resolve_cm(trim(l.substr(eq + 1)), val)) return false;
      val = trim(val);
      if (cur_name.empty()) { err = "field '" + key + "' outside a section in " + real; return false; }
      if (key.compare(0, cur_name.size() + 1, cur_name + ".") == 0) key = key.substr(cur_name.size() + 1);
      if (cur_type == "cComponentManager") {
        if (key.compare(0, 9, "instance[") == 0) {
          const size_t e = key.find(']');
          if (e != std::string::npos && key.substr(e + 1) == ".type") instance(key.substr(9, e - 9), val);
        }
        return true;                                       // nThreads, printLevelStats ...: the manager's own business
      }
      ConfInstance &ci = instance(cur_name, cur_type);
      bool replaced = false;
      for (auto &kv : ci.opts)
        if (kv.first == key) { kv.second = val; replaced = true; }
      if (!replaced) ci.opts.push_back({key, val});
        return true;
  }

  bool parse_file(const std::string &path, const std::string &parent) {
    if (++depth > 32) { err = "include depth exceeded (loop?) at '" + path + "'"; return false; }
    std::string real;
    for (const std::string &cand : {path, top_dir + "/" + path, dir_of(parent) + "/" + path})
      if (file_exists(cand)) { real = cand; break; }
    if (real.empty()) { err = "cannot open config file '" + path + "'"; return false; }
    std::ifstream f(real);
    std::string line;
    bool in_block_comment = false;
    while (std::getline(f, line)) {
      std::string l = trim(line);
      if (in_block_comment) {
        const size_t e = l.find("*/");
        if (e == std::string::npos) continue;
        l = trim(l.substr(e + 2));
        in_block_comment = false;
      }
      if (l.compare(0, 2, "/*") == 0) {
        if (l.find("*/", 2) == std::string::npos) in_block_comment = true;
        continue;
      }
      if (l.empty() || l[0] == ';' || l[0] == '#' || l[0] == '%' || l.compare(0, 2, "//") == 0) continue;
      if (l.size() > 3 && l[0] == '\\' && l[1] == '{' && l.back() == '}') {               // include
        std::string inc;
        if (!resolve_cm(l.substr(2, l.size() - 3), inc)) return false;
        if (!parse_file(inc, real)) return false;
        continue;
      }
      if (!body_line(l, real)) return false;
    }
    --depth;
    return true;
  }
};

bool is_io_type(const std::string &t) {
  return t == "cDataMemory" || t == "cWaveSource" || t == "cComponentManager" || (t.size() > 4 && t.compare(t.size() - 4, 4, "Sink") == 0);
}

std::string canonical_value(const std::string &v) {
  char *end = nullptr;
  const double d = strtod(v.c_str(), &end);
  if (end != v.c_str() && *end == 0) {
    char buf[64];
    snprintf(buf, sizeof(buf), "%.17g", d);
    return buf;
  }
  return v;
}

// ---- symbolic columns of a data-memory level of the cepstral chains
struct Col {
  int kind;        // 0 cepstrum, 1 log energy
  int index;       // cepstrum number as the component names it
  int delta;       // order of delta regression applied
  bool cms;        // cFullinputMean applied (before any delta)
};

struct OptReader {
  const ConfInstance &ci;
  std::set<std::string> seen;
  std::string err;
  explicit OptReader(const ConfInstance &c) : ci(c) {}
  bool has(const std::string &k) const { return ci.find(k) != nullptr; }
  std::string str(const std::string &k, const std::string &d) {
    seen.insert(k);
    const std::string *v = ci.find(k);
    return v ? *v : d;
  }
  double num(const std::string &k, double d) {
    seen.insert(k);
    const std::string *v = ci.find(k);
    if (!v) return d;
    char *end = nullptr;
    const double x = strtod(v->c_str(), &end);
    if (end == v->c_str()) { err = "[" + ci.name + ":" + ci.type + "] " + k + " = '" + *v + "' is not a number"; return d; }
    return x;
  }
  // option must have this value (given or by default)
  bool require(const std::string &k, double want, double dflt) {
    const double x = num(k, dflt);
    if (x != want && err.empty()) {
      std::ostringstream o;
      o << "[" << ci.name << ":" << ci.type << "] " << k << " = " << x << " is not expressible on the fused path (needs " << want << ")";
      err = o.str();
    }
    return err.empty();
  }
  // every option of the section must have been looked at, or be one of the benign ones
  bool finish(const std::set<std::string> &benign) {
    if (!err.empty()) return false;
    for (const auto &kv : ci.opts) {
      if (seen.count(kv.first) || benign.count(kv.first)) continue;
      if (kv.first.compare(0, 17, "writer.levelconf.") == 0 || kv.first.compare(0, 17, "reader.levelconf.") == 0) continue;
      err = "[" + ci.name + ":" + ci.type + "] option '" + kv.first + "' is not expressible on the fused path";
      return false;
    }
    return true;
  }
};

const std::set<std::string> kBenign = {"reader.dmLevel", "writer.dmLevel", "copyInputName", "buffersize", "buffersize_sec", "blocksize",
                                       "blocksizeR", "blocksizeW", "blocksize_sec", "blocksizeR_sec", "blocksizeW_sec", "EOIlevel"};

int win_func_of(const std::string &w) {
  const std::string s = lower(w);
  if (s == "han" || s == "hann" || s == "hanning") return SMILEHIP_WIN_HANN;
  if (s == "ham" || s == "hamming") return SMILEHIP_WIN_HAMM;
  if (s == "rec" || s == "rectangular" || s == "none") return SMILEHIP_WIN_RECT;
  if (s == "gau" || s == "gauss" || s == "gaussian") return SMILEHIP_WIN_GAUSS;
  if (s == "sin" || s == "sine" || s == "cos" || s == "cosine") return SMILEHIP_WIN_SINE;
  if (s == "tri" || s == "triangle") return SMILEHIP_WIN_TRI;
  if (s == "bar" || s == "bartlett") return SMILEHIP_WIN_BARTLETT;
  if (s == "lac" || s == "lanczos") return SMILEHIP_WIN_LANCZOS;
  return -1;
}

std::vector<std::string> split_levels(const std::string &s) {
  std::vector<std::string> r;
  std::stringstream ss(s);
  std::string item;
  while (std::getline(ss, item, ';'))
    if (!trim(item).empty()) r.push_back(trim(item));
  return r;
}

}  // namespace

const std::string *ConfInstance::find(const std::string &key) const {
  for (const auto &kv : opts)
    if (kv.first == key) return &kv.second;
  return nullptr;
}

bool conf_parse(const std::string &path, const std::map<std::string, std::string> &cmdline, ConfFile &out, std::string &err) {
  out = ConfFile();
  Parser p{out, cmdline, dir_of(path), "", 0, "", ""};
  if (!p.parse_file(path, path)) { err = p.err; return false; }
  for (const ConfInstance &i : out.inst)
    if (i.type.empty()) { err = "instance '" + i.name + "' has no type"; return false; }
  return true;
}

bool conf_from_sections(const std::vector<ConfRawSection> &sections, const ConfCmValue &cm_value, ConfFile &out, std::string &err) {
  out = ConfFile();
  static const std::map<std::string, std::string> none;
  Parser p{out, none, ".", "", 0, "", ""};
  p.cm_value = &cm_value;
  for (const ConfRawSection &sec : sections) {
    p.cur_name = sec.name;
    p.cur_type = sec.type;
    if (sec.type != "cComponentManager") p.instance(sec.name, sec.type).has_section = true;
    for (const std::string &raw : sec.lines) {
      const std::string l = trim(raw);
      if (l.empty() || l[0] == ';' || l[0] == '#' || l[0] == '%' || l.compare(0, 2, "//") == 0) continue;
      if (!p.body_line(l, "(configuration manager)")) { err = p.err; return false; }
    }
  }
  for (const ConfInstance &i : out.inst)
    if (i.type.empty()) { err = "instance '" + i.name + "' has no type"; return false; }
  return true;
}

// FNV-1a over the processing components (sources, sinks and the data memory left out: they follow the command line),
// sections in instantiation order, options sorted, numbers in canonical form
uint64_t conf_fingerprint(const ConfFile &f) {
  uint64_t h = 1469598103934665603ull;
  auto eat = [&](const std::string &s) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    h ^= '\n'; h *= 1099511628211ull;
  };
  for (const ConfInstance &i : f.inst) {
    if (is_io_type(i.type)) continue;
    eat(i.name + ":" + i.type);
    std::vector<std::pair<std::string, std::string>> o = i.opts;
    std::sort(o.begin(), o.end());
    for (const auto &kv : o) eat(kv.first + "=" + canonical_value(kv.second));
  }
  return h;
}

// The options of the F0 group the chains take as parameters (smilehip_lld_config fields): cPitchShs (pitchShs.cpp /
// pitchBase.cpp options), cPitchSmootherViterbi bufferLength, cPitchJitter searchRangeRel / useBrokenJitterThresh and the
// threshold of the energy gate ([is13_volmerge] / [gemapsv01b_volmerge]). Returns the config field's name, or nullptr.
static const char *conf_is_f0_param(const ConfInstance &i, const std::string &key) {
  if (i.type == "cPitchShs") {
    if (key == "minPitch") return "pitch_min";
    if (key == "maxPitch") return "pitch_max";
    if (key == "voicingCutoff") return "voicing_cutoff";
    if (key == "nHarmonics") return "shs_n_harmonics";
    if (key == "compressionFactor") return "shs_compression";
  }
  if (i.type == "cPitchSmootherViterbi" && key == "bufferLength") return "vit_buffer_len";
  if (i.type == "cPitchJitter") {
    if (key == "searchRangeRel") return "jitter_search_range";
    if (key == "useBrokenJitterThresh") return "jitter_broken_thresh";
  }
  if (i.type == "cValbasedSelector" && key == "threshold" && i.name.size() >= 8 && i.name.compare(i.name.size() - 8, 8, "volmerge") == 0)
    return "f0_min_energy";
  return nullptr;
}

uint64_t conf_fingerprint_masked(const ConfFile &f) {
  uint64_t h = 1469598103934665603ull;
  auto eat = [&](const std::string &s) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    h ^= '\n'; h *= 1099511628211ull;
  };
  for (const ConfInstance &i : f.inst) {
    if (is_io_type(i.type)) continue;
    eat(i.name + ":" + i.type);
    std::vector<std::pair<std::string, std::string>> o = i.opts;
    std::sort(o.begin(), o.end());
    // a parameter's VALUE is masked, its presence is not: a file that leaves the line out gets the component's own default
    // from the reference (pitchJitter.cpp:47/77, pitchSmootherViterbi.cpp:42 ...), not the shipped file's value
    for (const auto &kv : o)
      eat(conf_is_f0_param(i, kv.first) ? kv.first + "=<parameter>" : kv.first + "=" + canonical_value(kv.second));
  }
  return h;
}

static bool conf_is_selection_param(const ConfInstance &i, const std::string &key) {
  return (i.type == "cMfcc" && key == "lastMfcc") || (i.type == "cFunctionals" && key == "functionalsEnabled");
}
uint64_t conf_fingerprint_masked2(const ConfFile &f) {
  uint64_t h = 1469598103934665603ull;
  auto eat = [&](const std::string &s) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    h ^= '\n'; h *= 1099511628211ull;
  };
  for (const ConfInstance &i : f.inst) {
    if (is_io_type(i.type)) continue;
    eat(i.name + ":" + i.type);
    std::vector<std::pair<std::string, std::string>> o = i.opts;
    std::sort(o.begin(), o.end());
    for (const auto &kv : o)
      eat((conf_is_f0_param(i, kv.first) || conf_is_selection_param(i, kv.first)) ? kv.first + "=<parameter>"
                                                                                  : kv.first + "=" + canonical_value(kv.second));
  }
  return h;
}

void conf_apply_f0_params(const ConfPlan &p, smilehip_lld_config &c) {
  for (const auto &kv : p.f0_params) {
    if (kv.first == "pitch_min") c.pitch_min = kv.second;
    else if (kv.first == "pitch_max") c.pitch_max = kv.second;
    else if (kv.first == "voicing_cutoff") c.voicing_cutoff = kv.second;
    else if (kv.first == "shs_n_harmonics") c.shs_n_harmonics = (int32_t)kv.second;
    else if (kv.first == "shs_compression") c.shs_compression = (float)kv.second;
    else if (kv.first == "vit_buffer_len") c.vit_buffer_len = (int32_t)kv.second;
    else if (kv.first == "jitter_search_range") c.jitter_search_range = kv.second;
    else if (kv.first == "jitter_broken_thresh") c.jitter_broken_thresh = (int32_t)kv.second;
    else if (kv.first == "f0_min_energy") c.f0_min_energy = (float)kv.second;
  }
}

namespace {
struct KnownSet { uint64_t fingerprint; const char *set; const char *file; uint64_t masked; uint64_t masked2 = 0; };
// fingerprints of the reference's own files (smilextract_hip -C <file> --fingerprint prints them), computed from
// config/is09-13/IS09_emotion.conf, config/compare16/ComParE_2016.conf, config/is09-13/IS13_ComParE.conf and
// config/egemaps/v02/eGeMAPSv02.conf (and its two sub-graphs, GeMAPSv01b.conf / eGeMAPSv01b.conf) with their includes and every command-line option at its default
const KnownSet kKnownSets[] = {
    {0x84b42d91f07908acull, "is09_emotion", "IS09_emotion.conf", 0},
    {0x9c0cc5d423cf1caeull, "compare16", "ComParE_2016.conf", 0xfae69d014d0dfa60ull, 0x4f1f91fabefddeebull},
    {0xb409f82a744a62d2ull, "is13_compare", "IS13_ComParE.conf", 0x5a6b988a3b68a67full, 0xfe35c75d8b8a8138ull},
    {0xcb39106ceca6480cull, "egemapsv02", "eGeMAPSv02.conf", 0xb17c462e03023f64ull},
    // sub-graphs of eGeMAPSv02.conf: their levels are column subsets of its levels (smilehip_host.hpp, egemaps_subset_columns)
    {0x07901132e1b570f7ull, "gemapsv01b", "GeMAPSv01b.conf", 0xc9092680fbe30c4full},
    {0xe0e523408610f510ull, "egemapsv01b", "eGeMAPSv01b.conf", 0x3cde1ce6e1776678ull},
    // the v01a files: the same sub-graphs with zeroPadSymmetric = 0, useBrokenJitterThresh = 1, maxF = 5500 (smilehip_config_egemapsv01a)
    {0x54c6dceabc3deb6full, "gemapsv01a", "GeMAPSv01a.conf", 0x20e46a265c0b88beull},
    {0x378e8dfc3678ce85ull, "egemapsv01a", "eGeMAPSv01a.conf", 0x6956ca30c993252cull},
};
}  // namespace

bool conf_check_io(const ConfFile &f, const std::set<std::string> &produced, std::string &err) {
  auto num_is = [](const std::string &v, double want) {
    char *end = nullptr;
    const double d = strtod(v.c_str(), &end);
    return end != v.c_str() && d == want;
  };
  for (const ConfInstance &i : f.inst) {
    if (i.type == "cWaveSource") {
      // waveSource.cpp:52-68: the fused path packs whole files
      static const std::pair<const char *, double> whole[] = {{"start", 0}, {"end", -1}, {"endrel", 0}, {"startSamples", 0},
                                                               {"endSamples", -1}, {"endrelSamples", 0}, {"noHeader", 0}};
      for (const auto &w : whole) {
        const std::string *v = i.find(w.first);
        if (v && !num_is(*v, w.second)) {
          err = "[" + i.name + ":cWaveSource] " + w.first + " = " + *v + ": the fused path reads whole files (only " + w.first + " = " +
                canonical_value(std::to_string(w.second)) + " is implemented)";
          return false;
        }
      }
      const std::string *seg = i.find("segmentList");
      if (seg && !seg->empty()) { err = "[" + i.name + ":cWaveSource] segmentList is not implemented by the fused path"; return false; }
      continue;
    }
    const bool sink = i.type.size() > 4 && i.type.compare(i.type.size() - 4, 4, "Sink") == 0;
    if (!sink) continue;
    const std::string *fn = i.find("filename");
    if (!fn || *fn == "?") continue;                       // a disabled sink (the shared output files' default)
    if (i.type != "cHtkSink" && i.type != "cCsvSink" && i.type != "cArffSink") {
      err = "[" + i.name + ":" + i.type + "] is active (filename = " + *fn + ") but smilextract_hip has no writer for it";
      return false;
    }
    const std::string *lv = i.find("reader.dmLevel");
    if (lv && !produced.empty()) {
      size_t a = 0;
      while (a <= lv->size()) {                            // "lld;lld_de": every level the sink reads
        const size_t b = lv->find(';', a);
        const std::string one = lv->substr(a, b == std::string::npos ? std::string::npos : b - a);
        if (!one.empty() && !produced.count(one)) {
          err = "[" + i.name + ":" + i.type + "] reads level '" + one + "', which the fused path does not write (it keeps only the "
                "output levels; intermediate levels: per-component operators through the plugin)";
          return false;
        }
        if (b == std::string::npos) break;
        a = b + 1;
      }
    }
    const bool on_func = lv && *lv == "func";
    // The values the sink would run with -- the option if the file sets it, else the component's own default (csvSink.cpp:43-51,
    // 95-104; arffSink.cpp:42-51, 104-120; htkSink.cpp:33-37), `frameIndex` / `frameTime` overriding their synonyms `number` /
    // `timestamp` when set -- against the one value the writers of opensmile_amd/host/sinks.cpp implement for this kind of sink.
    auto value_of = [&](const char *key, double dflt, bool &bad) {
      const std::string *v = i.find(key);
      if (!v) return dflt;
      char *end = nullptr;
      const double d = strtod(v->c_str(), &end);
      if (end == v->c_str()) { bad = true; return dflt; }
      return d;
    };
    // what the file asks of a sink option against what smilextract_hip's writers do: `was_set` = the file names the option (under
    // either of its spellings), `exact` = the value itself must match (offsets), not only its being zero or not (switches)
    struct Want { const char *what; double has; double implemented; bool was_set; bool exact; };
    std::vector<Want> wants;
    bool bad = false;
    const auto set = [&](const char *a, const char *b = nullptr) { return i.find(a) != nullptr || (b && i.find(b) != nullptr); };
    const double number = i.find("frameIndex") ? value_of("frameIndex", 1, bad) : value_of("number", 1, bad);
    const double stamp = i.find("frameTime") ? value_of("frameTime", 1, bad) : value_of("timestamp", 1, bad);
    const bool number_set = set("frameIndex", "number"), stamp_set = set("frameTime", "timestamp");
    const auto plain = [&](const char *key, double dflt, double implemented, bool exact = false) {
      return Want{key, value_of(key, dflt, bad), implemented, set(key), exact};
    };
    if (i.type == "cCsvSink")
      wants = {{"timestamp / frameTime", stamp, 1, stamp_set, false}, plain("printHeader", 1, 1), {"number / frameIndex", number, 0, number_set, false},
               plain("frameLength", 0, 0), plain("lag", 0, 0, true), plain("append", 0, on_func ? 1.0 : 0.0)};
    if (i.type == "cArffSink")
      wants = {{"number / frameIndex", number, 0, number_set, false}, {"timestamp / frameTime", stamp, on_func ? 0.0 : 1.0, stamp_set, false},
               plain("frameTimeAdd", 0, 0, true), plain("frameLength", 0, 0), plain("lag", 0, 0, true), plain("append", 0, 1)};
    if (i.type == "cHtkSink") wants = {plain("append", 0, 0), plain("lag", 0, 0, true)};
    if (bad) { err = "[" + i.name + ":" + i.type + "] has a numeric option that does not parse"; return false; }
    for (const Want &w : wants)
      if (w.exact ? w.has != w.implemented : (w.has != 0) != (w.implemented != 0)) {
        err = "[" + i.name + ":" + i.type + "] " + w.what + " = " + canonical_value(std::to_string(w.has)) + (w.was_set ? "" : " (the component's default where the file is silent)") +
              " is not implemented by smilextract_hip's writer (only " + canonical_value(std::to_string(w.implemented)) + ")";
        return false;
      }
    if (i.type == "cArffSink" && !on_func) {
      err = "[" + i.name + ":cArffSink] an ARFF file of the LLD level (-lldarffoutput) is not implemented by smilextract_hip";
      return false;
    }
    const std::string *delim = i.find("delimChar");
    if (delim && *delim != ";") { err = "[" + i.name + ":cCsvSink] delimChar = " + *delim + " is not implemented (only ';')"; return false; }
  }
  return true;
}

static bool conf_to_plan_inner(const ConfFile &f, ConfPlan &p, std::string &err);
bool conf_to_plan(const ConfFile &f, ConfPlan &p, std::string &err) {
  const bool ok = conf_to_plan_inner(f, p, err);
  if (ok && p.wave_file.empty())                         // (the big sets return before the cepstral walk that records it)
    for (const ConfInstance &i : f.inst)
      if (i.type == "cWaveSource") { const std::string *fn = i.find("filename"); if (fn) p.wave_file = *fn; }
  return ok;
}
static bool conf_to_plan_inner(const ConfFile &f, ConfPlan &p, std::string &err) {
  p = ConfPlan();
  const uint64_t fp = conf_fingerprint(f);
  for (const KnownSet &k : kKnownSets)
    if (k.fingerprint == fp && fp != 0) {
      if (!conf_check_io(f, {"lld", "lld_de", "func"}, err)) return false;   // the shared output file's levels
      p.preset = k.set;
      p.describe = std::string("the graph of ") + k.file + " (every processing component and option identical)";
      return true;
    }
  // the same graph with other values of the options the kernels take as parameters (the F0 group's pitch range, harmonics,
  // compression, voicing cutoff, Viterbi buffer, jitter search range / threshold rule, energy gate)
  const uint64_t fpm = conf_fingerprint_masked(f);
  for (const KnownSet &k : kKnownSets)
    if (k.masked != 0 && k.masked == fpm) {
      if (!conf_check_io(f, {"lld", "lld_de", "func"}, err)) return false;
      p.preset = k.set;
      std::string what;
      for (const ConfInstance &i : f.inst)
        for (const auto &kv : i.opts) {
          const char *field = conf_is_f0_param(i, kv.first);
          if (!field) continue;
          char *end = nullptr;
          const double v = strtod(kv.second.c_str(), &end);
          if (end == kv.second.c_str()) { err = "[" + i.name + ":" + i.type + "] " + kv.first + " = " + kv.second + " is not a number"; return false; }
          p.f0_params[field] = v;
          what += (what.empty() ? "" : ", ") + kv.first + " = " + canonical_value(kv.second);
        }
      p.describe = std::string("the graph of ") + k.file + " with the file's own parameter values (" + what + ")";
      return true;
    }
  // the ComParE graphs with a lower lastMfcc and / or functional families removed: outputs of the shipped graph left out
  const uint64_t fpm2 = conf_fingerprint_masked2(f);
  for (const KnownSet &k : kKnownSets)
    if (k.masked2 != 0 && k.masked2 == fpm2) {
      if (!conf_check_io(f, {"lld", "lld_de", "func"}, err)) return false;
      p.preset = k.set;
      std::string what;
      for (const ConfInstance &i : f.inst)
        for (const auto &kv : i.opts) {
          if (const char *field = conf_is_f0_param(i, kv.first)) {
            char *end = nullptr;
            const double v = strtod(kv.second.c_str(), &end);
            if (end == kv.second.c_str()) { err = "[" + i.name + ":" + i.type + "] " + kv.first + " = " + kv.second + " is not a number"; return false; }
            p.f0_params[field] = v;
          } else if (i.type == "cMfcc" && kv.first == "lastMfcc") {
            p.last_mfcc = atoi(kv.second.c_str());
            if (p.last_mfcc < 1 || p.last_mfcc > 14) { err = "[" + i.name + ":cMfcc] lastMfcc = " + kv.second + ": the fused ComParE chain computes MFCC 1 .. 14 (a lower lastMfcc drops columns; a higher one is not built)"; return false; }
            what += (what.empty() ? "" : ", ") + std::string("lastMfcc = ") + kv.second;
          } else if (i.type == "cFunctionals" && kv.first == "functionalsEnabled") {
            const size_t at = i.name.find("functionals");
            const std::string inst = at == std::string::npos ? i.name : i.name.substr(at + 11);
            std::vector<std::string> fams;
            std::string tok;
            for (char ch : kv.second + ";") {
              if (ch == ';') { if (!tok.empty()) fams.push_back(tok); tok.clear(); }
              else if (!isspace((unsigned char)ch)) tok += ch;
            }
            p.func_enabled[inst] = fams;
            what += (what.empty() ? "" : ", ") + i.name + ".functionalsEnabled = " + kv.second;
          }
        }
      p.describe = std::string("the graph of ") + k.file + " with outputs left out (" + what + ")";
      return true;
    }
  // ---- a cepstral chain: framer -> [pre-emphasis] -> window -> FFT -> magnitude -> mel -> MFCC | PLP, optional log energy,
  // optional mean normalisation, delta regression, concatenation
  std::map<std::string, std::vector<const ConfInstance *>> by_type;
  const ConfInstance *htk_sink = nullptr, *any_sink = nullptr;
  for (const ConfInstance &i : f.inst) {
    if (i.type == "cHtkSink" && !htk_sink) htk_sink = &i;
    if (i.type.size() > 4 && i.type.compare(i.type.size() - 4, 4, "Sink") == 0 && !any_sink) any_sink = &i;
    if (!is_io_type(i.type)) by_type[i.type].push_back(&i);
  }
  static const std::set<std::string> chain_types = {"cFramer", "cVectorPreemphasis", "cWindower", "cTransformFFT", "cFFTmagphase", "cMelspec",
                                                    "cMfcc", "cPlp", "cEnergy", "cFullinputMean", "cDeltaRegression", "cVectorConcat"};
  for (const auto &kv : by_type)
    if (!chain_types.count(kv.first)) {
      err = "component [" + kv.second[0]->name + ":" + kv.first + "] is not part of a graph the fused path knows (the cepstral chains of "
            "config/mfcc and config/plp with any option values; IS09_emotion, ComParE_2016, IS13_ComParE, eGeMAPSv02 as shipped)";
      return false;
    }
  auto one = [&](const char *t, bool required) -> const ConfInstance * {
    auto it = by_type.find(t);
    if (it == by_type.end()) {
      if (required) err = std::string("the graph has no ") + t;
      return nullptr;
    }
    if (it->second.size() != 1) { err = std::string("more than one ") + t + " (one cepstral chain per file is expressible)"; return nullptr; }
    return it->second[0];
  };
  const ConfInstance *fr = one("cFramer", true), *win = one("cWindower", true), *fft = one("cTransformFFT", true),
                     *mag = one("cFFTmagphase", true), *mel = one("cMelspec", true);
  if (!err.empty()) return false;
  const ConfInstance *pe = one("cVectorPreemphasis", false), *en = one("cEnergy", false);
  one("cFullinputMean", false);
  if (!err.empty()) return false;
  const ConfInstance *mf = one("cMfcc", false), *pl = one("cPlp", false);
  if (!err.empty()) return false;
  if ((mf != nullptr) == (pl != nullptr)) { err = "the graph needs exactly one of cMfcc / cPlp"; return false; }
  smilehip_lld_config &c = p.cfg;
  smilehip_config_mfcc12_0_d_a(&c);
  p.plp = pl != nullptr;
  if (p.plp) smilehip_config_plp_0_d_a(&c);
  auto lvl = [](const ConfInstance *i, const char *k) { const std::string *v = i->find(k); return v ? *v : std::string(); };
  auto wired = [&](const ConfInstance *i, const std::string &want) {
    if (lvl(i, "reader.dmLevel") == want) return true;
    err = "[" + i->name + ":" + i->type + "] reads '" + lvl(i, "reader.dmLevel") + "', expected '" + want + "' (not the cepstral chain's wiring)";
    return false;
  };
  // cFramer (src/core/winToVecProcessor.cpp:51-83)
  {
    OptReader o(*fr);
    if (!wired(fr, "wave")) return false;
    c.frame_size_sec = o.num("frameSize", 0.025);
    c.frame_step_sec = o.num("frameStep", 0.0);
    if (c.frame_step_sec == 0.0) c.frame_step_sec = c.frame_size_sec;
    if (lower(o.str("frameMode", "fixed")) != "fixed") o.err = "[" + fr->name + ":cFramer] only frameMode = fixed is expressible";
    if (lower(o.str("frameCenterSpecial", "left")) != "left") o.err = "[" + fr->name + ":cFramer] only frameCenterSpecial = left is expressible";
    o.require("noPostEOIprocessing", 1, 1);
    o.require("frameSizeFrames", 0, 0); o.require("frameStepFrames", 0, 0); o.require("frameCenter", 0, 0);
    o.require("frameCenterFrames", 0, 0); o.require("allowLastFrameIncomplete", 0, 0);
    if (!o.finish(kBenign)) { err = o.err; return false; }
  }
  std::string level = lvl(fr, "writer.dmLevel");
  const std::string frames_level = level;
  for (const ConfInstance *st : {pe, win, fft, mag, mel})
    if (st) p.stage_levels.push_back(lvl(st, "writer.dmLevel"));
  for (const ConfInstance &i : f.inst)
    if (i.type == "cWaveSource") { const std::string *fn = i.find("filename"); if (fn) p.wave_file = *fn; }
  c.preemph = 0;
  if (pe) {                                             // src/dspcore/vectorPreemphasis.cpp:33-35
    OptReader o(*pe);
    if (!wired(pe, level)) return false;
    c.preemph = 1;
    c.preemph_k = (float)o.num("k", 0.97);
    c.preemph_de = (int)o.num("de", 0);
    o.require("f", 0, 0);
    if (!o.finish(kBenign)) { err = o.err; return false; }
    level = lvl(pe, "writer.dmLevel");
  }
  {                                                     // src/dspcore/windower.cpp:36-49
    OptReader o(*win);
    if (!wired(win, level)) return false;
    const std::string wf = o.str("winFunc", "Han");
    c.win_func = win_func_of(wf);
    if (c.win_func < 0) o.err = "[" + win->name + ":cWindower] winFunc = " + wf + " is not expressible (Hann, Hamming, rectangular, Gauss, sine, triangle, Bartlett, Lanczos are)";
    c.win_gain = o.num("gain", 1.0);
    c.win_offset = o.num("offset", 0.0);
    c.win_sigma = o.num("sigma", 0.4);
    o.require("fade", 0, 0); o.require("squareRoot", 0, 0); o.require("xshift", 0, 0); o.require("processArrayFields", 1, 1);
    if (!o.finish(kBenign)) { err = o.err; return false; }
    level = lvl(win, "writer.dmLevel");
  }
  {                                                     // src/dspcore/transformFft.cpp:34-35
    OptReader o(*fft);
    if (!wired(fft, level)) return false;
    o.require("inverse", 0, 0);
    c.zero_pad_symmetric = (int)o.num("zeroPadSymmetric", 1);
    o.require("processArrayFields", 1, 1);
    if (!o.finish(kBenign)) { err = o.err; return false; }
    level = lvl(fft, "writer.dmLevel");
  }
  {                                                     // src/dspcore/fftmagphase.cpp:39-49
    OptReader o(*mag);
    if (!wired(mag, level)) return false;
    o.require("inverse", 0, 0); o.require("magnitude", 1, 1); o.require("phase", 0, 0); o.require("joinMagphase", 0, 0);
    o.require("normalise", 0, 0); o.require("power", 0, 0); o.require("dBpsd", 0, 0); o.require("processArrayFields", 1, 1);
    if (!o.finish(kBenign)) { err = o.err; return false; }
    level = lvl(mag, "writer.dmLevel");
  }
  {                                                     // src/lldcore/melspec.cpp:32-44
    OptReader o(*mel);
    if (!wired(mel, level)) return false;
    c.n_bands = (int)o.num("nBands", 26);
    c.lofreq = (float)o.num("lofreq", 20.0);
    c.hifreq = (float)o.num("hifreq", 8000.0);
    c.use_power = (int)o.num("usePower", 0);
    c.mel_htk_compatible = (int)o.num("htkcompatible", 1);
    if (lower(o.str("specScale", "mel")) != "mel") o.err = "[" + mel->name + ":cMelspec] only specScale = mel is expressible";
    if (lower(o.str("bwMethod", "lr")) != "lr") o.err = "[" + mel->name + ":cMelspec] only bwMethod = lr is expressible";
    o.require("inverse", 0, 0); o.require("showFbank", 0, 0); o.require("processArrayFields", 1, 1);
    if (!o.finish(kBenign)) { err = o.err; return false; }
    level = lvl(mel, "writer.dmLevel");
  }
  std::map<std::string, std::vector<Col>> levels;        // what each level of the rest of the graph holds
  std::string cep_name;
  if (mf) {                                             // src/lldcore/mfcc.cpp:33-44
    OptReader o(*mf);
    if (!wired(mf, level)) return false;
    c.first_mfcc = (int)o.num("firstMfcc", 1);
    c.last_mfcc = o.has("lastMfcc") ? (int)o.num("lastMfcc", 12) : c.first_mfcc + (int)o.num("nMfcc", 12) - 1;
    c.cep_lifter = (float)o.num("cepLifter", 22.0);
    c.mfcc_htk_compatible = (int)o.num("htkcompatible", 1);
    c.melfloor = (float)o.num("melfloor", 0.00000001);
    o.require("doLog", 1, 1); o.require("inverse", 0, 0); o.require("printDctBaseFunctions", 0, 0); o.require("processArrayFields", 1, 1);
    if (o.has("nameAppend") && o.str("nameAppend", "mfcc") != "mfcc") o.err = "[" + mf->name + ":cMfcc] nameAppend must stay 'mfcc'";
    if (!o.finish(kBenign)) { err = o.err; return false; }
    if (c.first_mfcc < 0 || c.last_mfcc < c.first_mfcc) { err = "[" + mf->name + ":cMfcc] bad firstMfcc / lastMfcc"; return false; }
    std::vector<Col> cols;
    for (int i = c.first_mfcc; i <= c.last_mfcc; ++i) cols.push_back(Col{0, i, 0, false});
    levels[lvl(mf, "writer.dmLevel")] = cols;
    cep_name = "pcm_fftMag_mfcc";
  } else {                                              // src/lldcore/plp.cpp:45-67
    OptReader o(*pl);
    if (!wired(pl, level)) return false;
    c.plp_lp_order = (int)o.num("lpOrder", 5);
    const int first_cc = (int)o.num("firstCC", 1);
    int last_cc = (int)o.num("lastCC", -1);
    const int n_ceps = (int)o.num("nCeps", -1);
    if (last_cc < 0) last_cc = n_ceps < 0 ? c.plp_lp_order : n_ceps;
    if (last_cc != c.plp_lp_order || first_cc < 0 || first_cc > 1) o.err = "[" + pl->name + ":cPlp] only firstCC = 0|1 with lastCC = lpOrder is expressible";
    c.first_mfcc = first_cc;
    c.last_mfcc = c.plp_lp_order;
    c.cep_lifter = (float)o.num("cepLifter", 0.0);
    c.plp_compression = (float)o.num("compression", 0.33);
    o.require("htkcompatible", 1, 1);
    // the PLP-CC branch the fused kernel implements is the one of config/plp: power mel spectrum in, no log / inverse log
    o.require("doLog", 0, 1); o.require("doAud", 1, 1); o.require("RASTA", 0, 0); o.require("newRASTA", 0, 0);
    o.require("doInvLog", 0, 1); o.require("doIDFT", 1, 1); o.require("doLP", 1, 1); o.require("doLpToCeps", 1, 1);
    o.require("processArrayFields", 1, 1);
    if (!o.finish(kBenign)) { err = o.err; return false; }
    std::vector<Col> cols;
    for (int i = first_cc; i <= c.plp_lp_order; ++i) cols.push_back(Col{0, i, 0, false});
    levels[lvl(pl, "writer.dmLevel")] = cols;
    cep_name = "PlpCC";
  }
  c.append_log_energy = 0;
  if (en) {                                             // src/lldcore/energy.cpp:33-44: HTK log energy of the raw frame
    OptReader o(*en);
    if (!wired(en, frames_level)) return false;
    o.require("htkcompatible", 1, 0); o.require("log", 1, 1); o.require("rms", 0, 1); o.require("energy2", 0, 0);
    o.require("escaleLog", 1, 1); o.require("ebiasLog", 0, 0); o.require("processArrayFields", 0, 0);
    if (o.str("nameAppend", "energy") != "energy") o.err = "[" + en->name + ":cEnergy] nameAppend must stay 'energy'";
    if (!o.finish(kBenign)) { err = o.err; return false; }
    levels[lvl(en, "writer.dmLevel")] = {Col{1, 0, 0, false}};
    c.append_log_energy = 1;
  }
  // the rest, as soon as the levels a component reads are known: cFullinputMean, cDeltaRegression, cVectorConcat
  int delta_win = -1;
  std::string delta_suffix = "de";
  std::vector<const ConfInstance *> pending;
  for (const ConfInstance &i : f.inst)
    if (i.type == "cFullinputMean" || i.type == "cDeltaRegression" || i.type == "cVectorConcat") pending.push_back(&i);
  while (!pending.empty()) {
    bool progress = false;
    for (size_t pi = 0; pi < pending.size(); ++pi) {
      const ConfInstance &i = *pending[pi];
      std::vector<Col> in;
      bool ready = true;
      for (const std::string &l : split_levels(lvl(&i, "reader.dmLevel"))) {
        auto it = levels.find(l);
        if (it == levels.end()) { ready = false; break; }
        in.insert(in.end(), it->second.begin(), it->second.end());
      }
      if (!ready) continue;
      OptReader o(i);
      if (i.type == "cFullinputMean") {                   // src/dspcore/fullinputMean.cpp:35-45
        if (lower(o.str("meanNorm", "amean")) != "amean") o.err = "[" + i.name + ":cFullinputMean] only meanNorm = amean is expressible";
        o.require("mvn", 0, 0); o.require("symmSubtract", 0, 0); o.require("subtractClipToZero", 0, 0); o.require("specEnorm", 0, 0);
        o.require("htkLogEnorm", 0, 0); o.require("multiLoopMode", 0, 0); o.require("excludeZeros", 0, 0);
        for (Col &cc : in) {
          if (cc.delta != 0 || cc.kind != 0) o.err = "[" + i.name + ":cFullinputMean] mean normalisation is expressible on the static cepstra only";
          cc.cms = true;
        }
      } else if (i.type == "cDeltaRegression") {          // src/dspcore/deltaRegression.cpp:34-41
        const int w = (int)o.num("deltawin", 2);
        if (delta_win >= 0 && w != delta_win) o.err = "[" + i.name + ":cDeltaRegression] all delta stages must share one deltawin";
        delta_win = w;
        o.require("absOutput", 0, 0); o.require("halfWaveRect", 0, 0); o.require("onlyInSegments", 0, 0); o.require("relativeDelta", 0, 0);
        o.require("noPostEOIprocessing", 0, 0);
        delta_suffix = o.str("nameAppend", "de");
        o.str("zeroSegBound", "1");
        for (Col &cc : in) {
          if (cc.cms) o.err = "[" + i.name + ":cDeltaRegression] deltas of the mean-normalised level are not expressible (take them from the un-normalised level)";
          ++cc.delta;
        }
      } else {
        o.str("includeSingleElementFields", "0"); o.str("processArrayFields", "1");
      }
      if (!o.finish(kBenign)) { err = o.err; return false; }
      levels[lvl(&i, "writer.dmLevel")] = in;
      pending.erase(pending.begin() + (long)pi);
      progress = true;
      break;
    }
    if (!progress) {
      err = "[" + pending[0]->name + ":" + pending[0]->type + "] reads level '" + lvl(pending[0], "reader.dmLevel") + "', which the cepstral chain does not produce";
      return false;
    }
  }
  {
    const int n_cep_ = p.plp ? c.plp_lp_order - c.first_mfcc + 1 : c.last_mfcc - c.first_mfcc + 1;
    for (const auto &kv : levels) {
      std::vector<int> cols;
      bool plain = true;
      for (const Col &cc : kv.second) {
        if (cc.delta != 0 || cc.cms) { plain = false; break; }
        cols.push_back(cc.kind == 1 ? n_cep_ : cc.index - c.first_mfcc);
      }
      if (plain && !cols.empty()) p.static_levels[kv.first] = cols;
    }
  }
  // what the sink reads
  if (!any_sink) { err = "the graph has no sink"; return false; }
  std::vector<Col> outc;
  for (const std::string &l : split_levels(lvl(any_sink, "reader.dmLevel"))) {
    auto it = levels.find(l);
    if (it == levels.end()) { err = "[" + any_sink->name + ":" + any_sink->type + "] reads level '" + l + "', which the cepstral chain does not produce"; return false; }
    outc.insert(outc.end(), it->second.begin(), it->second.end());
  }
  // every other active sink must read the same levels (the fused path writes one output matrix; a sink on a stage level or on
  // another combination would silently get nothing), and the source / sink options must be the ones the writers implement
  {
    const std::string want = lvl(any_sink, "reader.dmLevel");
    for (const ConfInstance &i : f.inst) {
      const bool is_sink = i.type.size() > 4 && i.type.compare(i.type.size() - 4, 4, "Sink") == 0;
      const std::string *fn = i.find("filename");
      if (!is_sink || &i == any_sink || !fn || *fn == "?") continue;
      if (lvl(&i, "reader.dmLevel") != want) {
        err = "[" + i.name + ":" + i.type + "] reads '" + lvl(&i, "reader.dmLevel") + "' while [" + any_sink->name + "] reads '" + want +
              "': the fused path writes one output matrix (a sink on a stage level: per-component operators through the plugin)";
        return false;
      }
    }
    std::set<std::string> produced;
    for (const std::string &l : split_levels(want)) produced.insert(l);
    if (!conf_check_io(f, produced, err)) return false;
    p.out_levels = want;
    if (produced.size() == 1)
      for (const ConfInstance &i : f.inst) {
        const std::string *w = i.find("writer.dmLevel");
        if (w && *w == want) { p.out_writer_name = i.name; p.out_writer_type = i.type; }
      }
  }
  // must be [cepstra (, energy)] x (1 + n_delta) blocks, the cepstra of block 0 either all normalised or none
  const int n_cep = p.plp ? c.plp_lp_order - c.first_mfcc + 1 : c.last_mfcc - c.first_mfcc + 1;
  const int block = n_cep + (c.append_log_energy ? 1 : 0);
  if (block == 0 || outc.size() % (size_t)block != 0 || outc.size() / (size_t)block > 3) {
    err = "the sink's columns are not [static | delta | acceleration] blocks of the cepstral chain";
    return false;
  }
  c.n_delta = (int)(outc.size() / (size_t)block) - 1;
  c.delta_win = delta_win < 0 ? 2 : delta_win;
  c.cms = outc[0].cms ? 1 : 0;
  for (size_t k = 0; k < outc.size(); ++k) {
    const int b = (int)(k / (size_t)block), j = (int)(k % (size_t)block);
    const Col &cc = outc[k];
    const bool want_energy = c.append_log_energy && j == n_cep;
    const int first = c.first_mfcc;
    const bool ok = cc.delta == b && (want_energy ? cc.kind == 1 : (cc.kind == 0 && cc.index == first + j)) &&
                    cc.cms == (b == 0 && !want_energy && c.cms);
    if (!ok) { err = "the sink's column order is not [cepstra, energy | their deltas | their accelerations]"; return false; }
  }
  if (!c.append_log_energy && en) { err = "cEnergy is computed but not written"; return false; }
  // names as the reference's data memory builds them (field name + [index]); parmKind of the file's own HTK sink
  const char *sfx[3] = {"", "_", "_"};
  for (int b = 0; b <= c.n_delta; ++b) {
    std::string suffix;
    for (int q = 0; q < b; ++q) suffix += std::string(sfx[1]) + delta_suffix;
    for (int j = 0; j < n_cep; ++j) {
      char idx[32];
      snprintf(idx, sizeof(idx), "[%d]", p.plp ? j : c.first_mfcc + j);
      p.lld_names.push_back(cep_name + suffix + idx);
    }
    if (c.append_log_energy) p.lld_names.push_back("pcm_LOGenergy" + suffix);
  }
  p.parm_kind = 9;
  if (htk_sink) {
    const std::string *pk = htk_sink->find("parmKind");
    if (pk) p.parm_kind = atoi(pk->c_str());
  }
  std::ostringstream d;
  d << (p.plp ? "PLP" : "MFCC") << " chain: " << c.frame_size_sec * 1e3 << " ms / " << c.frame_step_sec * 1e3 << " ms frames, " << c.n_bands
    << " bands " << c.lofreq << "-" << c.hifreq << " Hz, " << n_cep << " cepstra" << (c.append_log_energy ? " + log energy" : "")
    << (c.cms ? ", mean-normalised" : "") << ", " << c.n_delta << " delta stage(s)";
  p.describe = d.str();
  return true;
}

}  // namespace smilehip_host
