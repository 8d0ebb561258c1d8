// Host-side C++ above the C ABI (include/smilehip.h): the data formats either side of the
// LLD path -- RIFF/WAVE ingest and the HTK / CSV / ARFF writers, byte
// layouts and text formats as the reference's source and sinks produce them. No HIP headers:
// the device is reached only through libsmilehip's extern "C" entry points.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "smilehip.h"

namespace smilehip_host {

// sWaveParameters as filled by smilePcm_readWaveHeader (src/smileutil/smileUtil.c:2374-2487)
struct WaveInfo {
  long sample_rate = 0;
  int sample_type = 0;      // 1 = PCM, 3 = IEEE float
  int n_chan = 0;
  int block_size = 0;
  int n_bps = 0;            // bytes per sample
  int n_bits = 0;
  long n_blocks = 0;        // sample frames in the data chunk
  long header_offset = 0;   // byte offset of the first sample
};

// Parses the header the way the reference does (RIFF/WAVE magic, chunks before "fmt " and
// "data" skipped with their pad byte, fmt size 16/18/40, PCM or IEEE float) and reads the
// data chunk (clipped to what the file really holds). Returns false + err on failure.
bool read_wave_file(const std::string &path, WaveInfo &info, std::vector<unsigned char> &data, std::string &err);
// The same in two steps, for hosts that read the samples straight into a staging buffer of their own: the header walk alone
// (info.n_blocks clipped to what the file holds), then `bytes` of the data chunk from info.header_offset into dst (pread).
// read_wave_data returns the sample frames it got (fewer than asked for if the file ended), -1 + err on an I/O error.
bool probe_wave_file(const std::string &path, WaveInfo &info, std::string &err);
long read_wave_data(const std::string &path, const WaveInfo &info, void *dst, size_t bytes, std::string &err);

// Element names of a level, as cCsvSink/cArffSink obtain them from the data memory
// (src/core/dataMemoryLevel.cpp naming: field name + "[index]" for array fields).
std::vector<std::string> lld_names_mfcc12_0_d_a();
std::vector<std::string> lld_names_plp_0_d_a();
std::vector<std::string> lld_names_htk_variant(bool plp, bool energy);   // MFCC12_{0,E}_D_A[_Z], PLP_{0,E}_D_A[_Z]
std::vector<std::string> lld_names_is09();
std::vector<std::string> lld_names_compare16();
std::vector<std::string> func_names_is09();     // 384: <lld>_<functional>
std::vector<std::string> func_names_compare16();   // 6373, the functionals level of ComParE_2016.conf
std::vector<std::string> lld_names_egemaps();      // 25, the LLD level of eGeMAPSv02.conf
std::vector<std::string> func_names_egemaps();     // 88, its functionals level
// GeMAPSv01b.conf and eGeMAPSv01b.conf are sub-graphs of eGeMAPSv02.conf whose outputs are column subsets of its outputs (same
// names, same values, same rows: measured against the binary, tests/test_gemaps_subsets.py): the columns of the v02 LLD level
// (func = false) / functionals level (func = true) that `set` ("gemapsv01b" | "egemapsv01b") writes, in its order; empty for
// any other name.
std::vector<int> egemaps_subset_columns(const std::string &set, bool func);
bool compare16_selection(bool is13, int last_mfcc, const std::map<std::string, std::vector<std::string>> &func_enabled,
                         std::vector<int> &sel_lld, std::vector<int> &sel_func, std::string &err);
std::vector<std::string> select_names(const std::vector<std::string> &names, const std::vector<int> &cols);
// rows x cols.size() matrix of the selected columns of x (rows x ld)
std::vector<float> select_columns(const float *x, int64_t rows, int64_t ld, const std::vector<int> &cols);
// value-name suffixes of one cFunctionals instance in output order (name_append = its functNameAppend option)
std::vector<std::string> funcspec_value_names(const smilehip_func_spec &spec, const std::string &name_append = "");

// cHtkSink (src/iocore/htkSink.cpp:90-105, 183-213): 12-byte big-endian header
// {nSamples u32, samplePeriod u32 [100 ns], sampleSize u16, parmKind u16} + big-endian float32 rows.
bool write_htk(const std::string &path, const float *x, int64_t rows, int cols, int64_t ld, double period_sec,
               int parm_kind, std::string &err);
// the same file from dense rows that are big-endian already (smilehip_htk_rows_be): one writev()
bool write_htk_be(const std::string &path, const void *be_rows, int64_t rows, int cols, double period_sec, int parm_kind,
                  std::string &err);

// printf("%e", v) / printf("%.0f", v) for a float argument without printf: the decimal digits by exact integer arithmetic
// (the value is m * 2^e with m < 2^24: 128-bit integers hold every case with |v| in [1e-21, 3.4e38]; round-half-even like
// glibc), snprintf for the rest (subnormal, tiny, inf, nan). Writes at dst (at least 48 bytes), returns the length; no
// terminating NUL. The CSV / ARFF writers spend their time here: 2.7 M values/s with fprintf, see DESIGN.md.
int format_e6(float v, char *dst);
int format_f0(float v, char *dst);                   // "%.0f"; exact for every float

// cCsvSink (src/iocore/csvSink.cpp:157-240): optional header line, 'name';frameTime;values with
// "%.0f" for integral values and "%e" otherwise, ';' delimiter.
struct CsvOptions {
  bool append = false, print_header = true, timestamp = true;
  std::string instance_name = "unknown";
};
// times: frameTime of each row (smilehip_row_time), or nullptr for row * period_sec
bool write_csv(const std::string &path, const std::vector<std::string> &names, const float *x, int64_t rows, int cols,
               int64_t ld, double period_sec, const double *times, const CsvOptions &opt, std::string &err);

// cArffSink (src/iocore/arffSink.cpp:246-333 header, 336-430 rows) with the shared
// arff_targets.conf.inc defaults: one numeric attribute "class", target "?" for every instance.
struct ArffOptions {
  bool append = true, timestamp = false;
  std::string relation = "openSMILE_features", instance_name = "unknown", class_type = "numeric", class_value = "?";
};
std::string arff_escape(const std::string &s);
bool write_arff(const std::string &path, const std::vector<std::string> &names, const float *x, int64_t rows, int cols,
                int64_t ld, double period_sec, const ArffOptions &opt, std::string &err);

}  // namespace smilehip_host
