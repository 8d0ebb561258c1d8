// HTK / CSV / ARFF writers with the reference sinks' byte layouts and printf formats.
#include <cmath>
#include <cstring>

#include "smilehip_host.hpp"

namespace smilehip_host {

namespace {
void be32(unsigned char *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
void be16(unsigned char *p, uint16_t v) { p[0] = v >> 8; p[1] = v; }
bool file_exists(const std::string &p) {
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return false;
  fclose(f);
  return true;
}
}  // namespace

bool write_htk(const std::string &path, const float *x, int64_t rows, int cols, int64_t ld, double period_sec,
               int parm_kind, std::string &err) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) { err = "cannot open '" + path + "' for writing"; return false; }
  unsigned char h[12];
  be32(h, (uint32_t)rows);
  // htkSink.cpp:93-99: a level without a period gets the dummy 0.01 s
  be32(h + 4, period_sec <= 0.0 ? 100000u : (uint32_t)std::round(period_sec * 10000000.0));
  be16(h + 8, (uint16_t)(sizeof(float) * (size_t)cols));
  be16(h + 10, (uint16_t)parm_kind);
  bool ok = fwrite(h, 1, 12, f) == 12;
  std::vector<unsigned char> row((size_t)cols * 4);
  for (int64_t t = 0; ok && t < rows; ++t) {
    for (int c = 0; c < cols; ++c) {
      uint32_t u;
      std::memcpy(&u, &x[t * ld + c], 4);
      be32(&row[(size_t)c * 4], u);
    }
    ok = fwrite(row.data(), 1, row.size(), f) == row.size();
  }
  if (fclose(f) != 0) ok = false;
  if (!ok) err = "error writing '" + path + "'";
  return ok;
}

bool write_csv(const std::string &path, const std::vector<std::string> &names, const float *x, int64_t rows, int cols,
               int64_t ld, double period_sec, const double *times, const CsvOptions &opt, std::string &err) {
  const bool ap = opt.append && file_exists(path);
  FILE *f = fopen(path.c_str(), ap ? "a" : "w");
  if (!f) { err = "cannot open '" + path + "' for writing"; return false; }
  const char d = ';';
  if (!ap && opt.print_header) {
    fprintf(f, "name%c", d);
    if (opt.timestamp) fprintf(f, "frameTime%c", d);
    for (int c = 0; c < cols - 1; ++c) fprintf(f, "%s%c", names[(size_t)c].c_str(), d);
    fprintf(f, "%s\n", names[(size_t)cols - 1].c_str());
  }
  for (int64_t t = 0; t < rows; ++t) {
    fprintf(f, "'%s'%c", opt.instance_name.c_str(), d);
    if (opt.timestamp) fprintf(f, "%f%c", times ? times[t] : (double)t * period_sec, d);
    for (int c = 0; c < cols; ++c) {
      const float v = x[t * ld + c];
      const char *end = (c == cols - 1) ? "\n" : ";";
      if (v == std::floor(v)) fprintf(f, "%.0f%s", v, end);      // csvSink.cpp:224-235
      else fprintf(f, "%e%s", v, end);
    }
  }
  const bool ok = fclose(f) == 0;
  if (!ok) err = "error writing '" + path + "'";
  return ok;
}

std::string arff_escape(const std::string &s) {           // cArffSink::escape, arffSink.cpp:190-244
  if (s.empty()) return "''";
  bool quote = false;
  std::string e;
  for (char c : s) {
    switch (c) {
      case '"': case '\'': case '%': case '\\': e += '\\'; e += c; quote = true; break;
      case '\r': e += "\\r"; quote = true; break;
      case '\n': e += "\\n"; quote = true; break;
      case '\t': e += "\\t"; quote = true; break;
      case ' ': case ',': case '{': case '}': e += c; quote = true; break;
      default: e += c;
    }
  }
  return quote ? "'" + e + "'" : e;
}

bool write_arff(const std::string &path, const std::vector<std::string> &names, const float *x, int64_t rows, int cols,
                int64_t ld, double period_sec, const ArffOptions &opt, std::string &err) {
  const bool ap = opt.append && file_exists(path);
  FILE *f = fopen(path.c_str(), ap ? "a" : "w");
  if (!f) { err = "cannot open '" + path + "' for writing"; return false; }
  const bool prname = !opt.instance_name.empty() && (opt.instance_name[0] != '-' || opt.instance_name.size() > 1);
  if (!ap) {
    fprintf(f, "@relation %s\n\n", arff_escape(opt.relation).c_str());
    if (prname) fprintf(f, "@attribute name string\n");
    if (opt.timestamp) fprintf(f, "@attribute frameTime numeric\n");
    for (int c = 0; c < cols; ++c) fprintf(f, "@attribute %s numeric\n", arff_escape(names[(size_t)c]).c_str());
    if (opt.class_type.empty()) fprintf(f, "@attribute class numeric\n");
    else fprintf(f, "@attribute class %s\n", opt.class_type.c_str());
    fprintf(f, "\n@data\n\n");
  }
  for (int64_t t = 0; t < rows; ++t) {
    if (prname) fprintf(f, "%s,", arff_escape(opt.instance_name).c_str());
    if (opt.timestamp) fprintf(f, "%f,", (double)t * period_sec);
    fprintf(f, "%e", x[t * ld]);
    for (int c = 1; c < cols; ++c) fprintf(f, ",%e", x[t * ld + c]);
    fprintf(f, ",%s\n", opt.class_value.empty() ? "NULL" : opt.class_value.c_str());
  }
  const bool ok = fclose(f) == 0;
  if (!ok) err = "error writing '" + path + "'";
  return ok;
}

}  // namespace smilehip_host
