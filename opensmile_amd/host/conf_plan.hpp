// smilextract_hip -C file.conf: the reference's configuration file format read by this host library and mapped to a plan
// of the fused path (see conf_plan.cpp). No reference code is linked.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "smilehip.h"

namespace smilehip_host {

struct ConfInstance {
  std::string name, type;
  std::vector<std::pair<std::string, std::string>> opts;      // in file order, later assignments replace earlier ones
  bool has_section = false;
  const std::string *find(const std::string &key) const;
};

struct ConfFile {
  std::vector<ConfInstance> inst;                              // in instantiation order
  std::map<std::string, std::string> cm_defaults;              // \cm[name{default}:...] options the file defines
  std::map<std::string, std::string> cm_short;                 // long name -> its one-letter form
  std::set<std::string> cm_used;
};

// cmdline: values for the file's \cm[...] options, keyed by the long or the one-letter name (no dash)
bool conf_parse(const std::string &path, const std::map<std::string, std::string> &cmdline, ConfFile &out, std::string &err);

// The same ConfFile from sections a host has ALREADY read: the reference's own cFileConfigReader keeps every section of the file
// (includes expanded, block comments removed) as its raw "field = value" lines (src/include/core/configManager.hpp:475-482
// fileInstance; src/core/configManager.cpp:1684-1699 addLine, :1747-2110 openInput) -- what the plugin hands over from the
// cConfigManager the component loader gave it. cm_value: the effective value of a \cm[...] option (command line, else default)
// from the host's own command-line parser; false = not known there (the default the placeholder itself names is used).
struct ConfRawSection { std::string name, type; std::vector<std::string> lines; };
typedef std::function<bool(const std::string &name, std::string &value)> ConfCmValue;
bool conf_from_sections(const std::vector<ConfRawSection> &sections, const ConfCmValue &cm_value, ConfFile &out, std::string &err);

// hash of every processing component (name, type, options; sources / sinks / data memory excluded)
uint64_t conf_fingerprint(const ConfFile &f);
// the same hash with the options left out that the kernels of the big sets take as parameters (conf_plan.cpp: conf_is_f0_param):
// a file that differs from a shipped set only in those is the set with other parameter values
uint64_t conf_fingerprint_masked(const ConfFile &f);
// ... and with the options left out whose edits only REMOVE outputs of the ComParE graphs (cMfcc lastMfcc, the cFunctionals
// instances' functionalsEnabled lists): every output of those graphs is computed independently of the others, so a file
// that lowers lastMfcc or drops functional families writes a column subset of what the shipped file writes
uint64_t conf_fingerprint_masked2(const ConfFile &f);

struct ConfPlan {
  std::string preset;                    // "is09_emotion", "compare16", "is13_compare", "egemapsv02", or "" = cfg below
  smilehip_lld_config cfg;               // a cepstral chain (SMILEHIP_CHAIN_MFCC / _PLP) with the file's option values
  bool plp = false;
  int parm_kind = 9;                     // of the file's own cHtkSink
  std::vector<std::string> lld_names;    // element names of the output level
  std::string describe;
  // cepstral chains: the levels between the framer and the cepstra (what the per-frame stages write), and the levels that
  // hold plain static columns of the fused static block (cepstrum j -> j, log energy -> number of cepstra)
  std::vector<std::string> stage_levels;
  std::map<std::string, std::vector<int>> static_levels;
  // cepstral chains: the level(s) the sinks read ("lld", or "a;b;c" if they read several) and, when it is ONE level, the instance
  // that writes it (name, type) -- the plugin's fused mode hands the finished rows out there (plugin_shared.hpp)
  std::string out_levels, out_writer_name, out_writer_type;
  std::string wave_file;                 // the wave source's filename option, command-line options applied
  // big sets recognised through the masked fingerprint: the values of the parameter options the file sets, keyed by the
  // smilehip_lld_config field they map to (pitch_min, pitch_max, voicing_cutoff, shs_n_harmonics, shs_compression,
  // vit_buffer_len, jitter_search_range, jitter_broken_thresh, f0_min_energy)
  std::map<std::string, double> f0_params;
  // ComParE graphs recognised through the second masked fingerprint: lastMfcc (0 = as shipped: 14) and, per cFunctionals
  // instance ("A", "B", "F0", "Nz", "LLD", "Delta"), the family names the file enables (absent = as shipped)
  int last_mfcc = 0;
  std::map<std::string, std::vector<std::string>> func_enabled;
};
void conf_apply_f0_params(const ConfPlan &p, smilehip_lld_config &cfg);

// false + err: the graph (component, option) the fused path cannot express
bool conf_to_plan(const ConfFile &f, ConfPlan &p, std::string &err);

// Sources and sinks are outside the fingerprint and the plan: this checks them. Every cWaveSource must read the whole file
// (start 0, end -1, no sample-based range, a RIFF header), every ACTIVE sink (filename other than "?") must be one
// smilextract_hip writes (cHtkSink / cCsvSink on the lld level, cArffSink / cCsvSink / cHtkSink on the functionals level, or
// the cepstral files' own cHtkSink) with the option values its writers implement; produced: the levels the plan writes
// (a sink reading any other level would get nothing). false + err names the instance and option.
bool conf_check_io(const ConfFile &f, const std::set<std::string> &produced, std::string &err);

}  // namespace smilehip_host
