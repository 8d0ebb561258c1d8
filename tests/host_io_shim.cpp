// Test-only C shim over the C++ host I/O library (opensmile_amd/host) so that the CPU tests
// can drive the writers and the wave reader through ctypes.
#include <cstring>

#include "smilehip_host.hpp"

using namespace smilehip_host;

extern "C" int shim_write_htk(const char *path, const float *x, int64_t rows, int cols, double period) {
  std::string err;
  return write_htk(path, x, rows, cols, cols, period, 9, err) ? 1 : 0;
}

// names: 0 = MFCC12_0_D_A lld, 1 = IS09 lld, 2 = IS09 func
extern "C" int shim_write_csv(const char *path, int names, const float *x, int64_t rows, int cols, double period,
                              const char *inst, int append, const double *times) {
  std::string err;
  CsvOptions o;
  o.instance_name = inst;
  o.append = append != 0;
  const std::vector<std::string> n = names == 0 ? lld_names_mfcc12_0_d_a() : (names == 1 ? lld_names_is09() : func_names_is09());
  if ((int)n.size() != cols) return -1;
  return write_csv(path, n, x, rows, cols, cols, period, times, o, err) ? 1 : 0;
}

extern "C" int shim_write_arff(const char *path, const float *x, int cols, const char *inst) {
  std::string err;
  ArffOptions o;
  o.instance_name = inst;
  const std::vector<std::string> n = func_names_is09();
  if ((int)n.size() != cols) return -1;
  return write_arff(path, n, x, 1, cols, cols, 0.0, o, err) ? 1 : 0;
}

// info: sample_rate, sample_type, n_chan, n_bps, n_bits, n_blocks, block_size, header_offset; returns bytes or -1
extern "C" long shim_read_wave(const char *path, long *info, void *buf, int64_t cap) {
  WaveInfo w;
  std::vector<unsigned char> d;
  std::string err;
  if (!read_wave_file(path, w, d, err)) return -1;
  info[0] = w.sample_rate; info[1] = w.sample_type; info[2] = w.n_chan; info[3] = w.n_bps; info[4] = w.n_bits;
  info[5] = w.n_blocks; info[6] = w.block_size; info[7] = w.header_offset;
  if (buf && cap > 0) std::memcpy(buf, d.data(), (size_t)std::min<int64_t>(cap, (int64_t)d.size()));
  return (long)d.size();
}

// the two-step reader of the file-to-file route: probe_wave_file (the quick in-memory walk, or the general one) + read_wave_data.
// Same answer as read_wave_file for every file it accepts; returns the bytes read, -1 on failure.
extern "C" long shim_probe_read_wave(const char *path, long *info, void *buf, int64_t cap) {
  WaveInfo w;
  std::string err;
  if (!probe_wave_file(path, w, err)) return -1;
  info[0] = w.sample_rate; info[1] = w.sample_type; info[2] = w.n_chan; info[3] = w.n_bps; info[4] = w.n_bits;
  info[5] = w.n_blocks; info[6] = w.block_size; info[7] = w.header_offset;
  const size_t bytes = (size_t)w.n_blocks * (size_t)w.block_size;
  if (!buf || cap <= 0) return (long)bytes;
  std::vector<unsigned char> d(bytes ? bytes : 1);
  const long got = read_wave_data(path, w, d.data(), bytes, err);
  if (got < 0) return -1;
  std::memcpy(buf, d.data(), (size_t)std::min<int64_t>(cap, (int64_t)got * w.block_size));
  return got * w.block_size;
}
// write_htk_be on rows the caller has made big-endian (what smilehip_htk_rows_be does on the device)
extern "C" int shim_write_htk_be(const char *path, const void *be_rows, int64_t rows, int cols, double period) {
  std::string err;
  return write_htk_be(path, be_rows, rows, cols, period, 9, err) ? 1 : 0;
}

// names of a level, '\n'-joined: 3 = ComParE_2016 LLD (130), 4 = ComParE_2016 functionals (6373); returns the byte count
extern "C" long shim_names(int which, char *buf, long cap) {
  const std::vector<std::string> n = which == 3 ? lld_names_compare16() : func_names_compare16();
  std::string all;
  for (const std::string &x : n) { all += x; all += '\n'; }
  if ((long)all.size() > cap) return -(long)all.size();
  std::memcpy(buf, all.data(), all.size());
  return (long)all.size();
}

// format_e6 / format_f0 on n floats against snprintf: returns the number of values whose text differs (first one in *bad)
extern "C" long shim_format_check(const float *x, long n, int use_f0, long *bad) {
  long diff = 0;
  char a[64], b[64];
  for (long i = 0; i < n; ++i) {
    const int la = use_f0 ? format_f0(x[i], a) : format_e6(x[i], a);
    const int lb = snprintf(b, sizeof b, use_f0 ? "%.0f" : "%e", x[i]);
    if (la != lb || memcmp(a, b, (size_t)la) != 0) { if (!diff && bad) *bad = i; ++diff; }
  }
  return diff;
}

// egemaps_subset_columns: writes the indices to out (capacity cap), returns their number
extern "C" int shim_egemaps_subset(const char *set, int func, int *out, int cap) {
  const std::vector<int> c = egemaps_subset_columns(set, func != 0);
  for (size_t i = 0; i < c.size() && (int)i < cap; ++i) out[i] = c[i];
  return (int)c.size();
}
// names of the eGeMAPSv02 levels joined with ';' (func: the 88 functionals, else the 25 LLDs)
extern "C" int shim_egemaps_names(int func, char *out, int cap) {
  const std::vector<std::string> n = func ? func_names_egemaps() : lld_names_egemaps();
  std::string s;
  for (size_t i = 0; i < n.size(); ++i) s += (i ? ";" : "") + n[i];
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)n.size();
}
