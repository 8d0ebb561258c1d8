// HTK / CSV / ARFF writers with the reference sinks' byte layouts and printf formats.
#include <fcntl.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

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

typedef unsigned __int128 u128;
const u128 kPow10[39] = {
    (u128)1ull, (u128)10ull, (u128)100ull, (u128)1000ull, (u128)10000ull, (u128)100000ull, (u128)1000000ull, (u128)10000000ull,
    (u128)100000000ull, (u128)1000000000ull, (u128)10000000000ull, (u128)100000000000ull, (u128)1000000000000ull,
    (u128)10000000000000ull, (u128)100000000000000ull, (u128)1000000000000000ull, (u128)10000000000000000ull,
    (u128)100000000000000000ull, (u128)1000000000000000000ull, (u128)10000000000000000000ull,
    (u128)10000000000000000000ull * 10u, (u128)10000000000000000000ull * 100u, (u128)10000000000000000000ull * 1000u,
    (u128)10000000000000000000ull * 10000u, (u128)10000000000000000000ull * 100000u, (u128)10000000000000000000ull * 1000000u,
    (u128)10000000000000000000ull * 10000000u, (u128)10000000000000000000ull * 100000000u,
    (u128)10000000000000000000ull * 1000000000u, (u128)10000000000000000000ull * 10000000000ull,
    (u128)10000000000000000000ull * 100000000000ull, (u128)10000000000000000000ull * 1000000000000ull,
    (u128)10000000000000000000ull * 10000000000000ull, (u128)10000000000000000000ull * 100000000000000ull,
    (u128)10000000000000000000ull * 1000000000000000ull, (u128)10000000000000000000ull * 10000000000000000ull,
    (u128)10000000000000000000ull * 100000000000000000ull, (u128)10000000000000000000ull * 1000000000000000000ull,
    (u128)10000000000000000000ull * 10000000000000000000ull};

// a buffered file: the writers below append text with memcpy and flush in 1 MB pieces
struct OutBuf {
  FILE *f;
  std::vector<char> b;
  size_t n = 0;
  bool ok = true;
  explicit OutBuf(FILE *file) : f(file), b((size_t)1 << 20) {}
  char *room(size_t need) {
    if (n + need > b.size()) flush();
    if (need > b.size()) b.resize(2 * need);
    return b.data() + n;
  }
  void put(const char *s, size_t len) {
    if (len > b.size() / 2) { flush(); ok = ok && fwrite(s, 1, len, f) == len; return; }
    std::memcpy(room(len), s, len);
    n += len;
  }
  void put(const std::string &s) { put(s.data(), s.size()); }
  void put(char c) { *room(1) = c; ++n; }
  void flush() {
    if (n) ok = ok && fwrite(b.data(), 1, n, f) == n;
    n = 0;
  }
};

// rows of text, formatted by fmt(t, dst) -> bytes (at most max_row bytes each), written in order. Long matrices are
// formatted by several threads, 512 rows per thread and round, and written in order; SMILEHIP_HOST_THREADS overrides the
// thread count (1 = the sequential path).
template <class Fmt>
bool write_text_rows(FILE *f, int64_t rows, size_t max_row, Fmt fmt) {
  unsigned T = std::thread::hardware_concurrency();
  if (const char *e = getenv("SMILEHIP_HOST_THREADS")) T = (unsigned)atoi(e);
  if (T > 32) T = 32;
  constexpr int64_t kSlice = 512;
  if (T < 2 || rows < 8 * kSlice) {
    OutBuf o(f);
    for (int64_t t = 0; t < rows; ++t) o.n += fmt(t, o.room(max_row));
    o.flush();
    return o.ok;
  }
  std::vector<std::vector<char>> buf(T, std::vector<char>((size_t)kSlice * max_row));
  std::vector<size_t> used(T, 0);
  bool ok = true;
  for (int64_t t0 = 0; ok && t0 < rows; t0 += (int64_t)T * kSlice) {
    std::vector<std::thread> th;
    for (unsigned i = 0; i < T; ++i) {
      const int64_t a = t0 + (int64_t)i * kSlice, b = std::min(rows, a + kSlice);
      used[i] = 0;
      if (a >= b) continue;
      th.emplace_back([&, i, a, b] {
        size_t n = 0;
        for (int64_t t = a; t < b; ++t) n += fmt(t, buf[i].data() + n);
        used[i] = n;
      });
    }
    for (auto &x : th) x.join();
    for (unsigned i = 0; ok && i < T; ++i)
      if (used[i]) ok = fwrite(buf[i].data(), 1, used[i], f) == used[i];
  }
  return ok;
}
}  // namespace

int format_f0(float v, char *dst) {
  if (!(std::fabs(v) < 9.0e18f)) return snprintf(dst, 48, "%.0f", v);          // huge, inf, nan
  const double d = (double)v;
  const double r = std::nearbyint(d);                     // round-half-even, as printf rounds
  char tmp[24];
  int n = 0;
  unsigned long long a = (unsigned long long)std::fabs(r);
  do { tmp[n++] = (char)('0' + a % 10); a /= 10; } while (a);
  int len = 0;
  if (std::signbit(d)) dst[len++] = '-';                  // "-0" for values in (-0.5, -0], as printf prints them
  while (n) dst[len++] = tmp[--n];
  return len;
}

int format_e6(float v, char *dst) {
  uint32_t bits;
  std::memcpy(&bits, &v, 4);
  const uint32_t ex = (bits >> 23) & 0xffu;
  int len = 0;
  if (bits >> 31) dst[len++] = '-';
  if ((bits & 0x7fffffffu) == 0) { std::memcpy(dst + len, "0.000000e+00", 12); return len + 12; }
  const float av = std::fabs(v);
  if (ex == 0 || ex == 0xff || av < 1e-21f) return snprintf(dst, 48, "%e", v);   // subnormal, tiny, inf, nan
  const uint64_t m64 = (bits & 0x7fffffu) | 0x800000u;
  const u128 m = m64;
  const int e = (int)ex - 150;                             // v = m * 2^e
  // decimal exponent: floor(log10 2^(ex-127)) is floor(log10 v) or one less; the loop below corrects it
  int k = (((int)ex - 127) * 1233) >> 12;
  {
    static const float *const pow10f = [] {                // 10^k as floats, k = -23 .. 38 (rounded: the loop absorbs it)
      static float t[62];
      for (int i = 0; i < 62; ++i) t[i] = (float)std::pow(10.0, (double)(i - 23));
      return t;
    }();
    if (k + 1 >= -23 && k + 1 <= 38 && av >= pow10f[k + 1 + 23]) ++k;
  }
  unsigned long long q = 0;
  for (int attempt = 0; attempt < 3; ++attempt) {
    // digits = round(v / 10^(k-6)) = round(N / D), half to even
    if (e < 0 && e >= -40 && k >= -5 && k <= 6) {          // the common case in 64 bits: N = m 10^(6-k) < 2^63, D = 2^-e
      static const uint64_t p10[12] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull,
                                       1000000000ull, 10000000000ull, 100000000000ull};
      const uint64_t N = m64 * p10[6 - k], D = 1ull << (-e);
      uint64_t qq = N >> (-e);
      if (qq < 1000000u) { --k; continue; }                // v < 10^k: the estimate was one too high (tested BEFORE rounding)
      if (qq >= 10000000u) { ++k; continue; }              // v >= 10^(k+1): one too low
      const uint64_t twice = (N & (D - 1)) << 1;
      if (twice > D || (twice == D && (qq & 1))) ++qq;
      if (qq == 10000000u) { qq = 1000000u; ++k; }         // 9.9999995.. rounds up to 1.000000e(k+1)
      q = qq;
      break;
    }
    u128 N, D;
    int shift = -1;                                        // D = 2^shift when >= 0
    if (e >= 0) { N = m << e; D = kPow10[k - 6 > 0 ? k - 6 : 0]; if (k < 6) N *= kPow10[6 - k]; }
    else if (k <= 6) { N = m * kPow10[6 - k]; D = (u128)1 << (-e); shift = -e; }
    else { N = m; D = ((u128)1 << (-e)) * kPow10[k - 6]; }
    u128 qq, rr;
    if (shift >= 0) { qq = N >> shift; rr = N & (D - 1); }
    else { qq = N / D; rr = N - qq * D; }
    if (qq < 1000000u) { --k; continue; }
    if (qq >= 10000000u) { ++k; continue; }
    const u128 twice = rr << 1;
    if (twice > D || (twice == D && (qq & 1))) ++qq;
    if (qq == 10000000u) { qq = 1000000u; ++k; }
    q = (unsigned long long)qq;
    break;
  }
  if (q == 0) return snprintf(dst, 48, "%e", v);           // (cannot happen; keeps the output right if it does)
  char dg[7];
  for (int i = 6; i >= 0; --i) { dg[i] = (char)('0' + q % 10); q /= 10; }
  dst[len++] = dg[0];
  dst[len++] = '.';
  std::memcpy(dst + len, dg + 1, 6);
  len += 6;
  dst[len++] = 'e';
  dst[len++] = k < 0 ? '-' : '+';
  const int ak = k < 0 ? -k : k;
  dst[len++] = (char)('0' + ak / 10);
  dst[len++] = (char)('0' + ak % 10);
  return len;
}

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

// The same file from rows that already are big-endian (smilehip_htk_rows_be did cHtkSink's swap on the device), dense
// (ld = cols): header and rows leave in ONE writev().
bool write_htk_be(const std::string &path, const void *be_rows, int64_t rows, int cols, double period_sec, int parm_kind,
                  std::string &err) {
  const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) { err = "cannot open '" + path + "' for writing"; return false; }
  unsigned char h[12];
  be32(h, (uint32_t)rows);
  be32(h + 4, period_sec <= 0.0 ? 100000u : (uint32_t)std::round(period_sec * 10000000.0));
  be16(h + 8, (uint16_t)(sizeof(float) * (size_t)cols));
  be16(h + 10, (uint16_t)parm_kind);
  const size_t body = (size_t)rows * (size_t)cols * 4;
  struct iovec iov[2] = {{h, 12}, {const_cast<void *>(be_rows), body}};
  size_t done = 0;
  const size_t total = 12 + body;
  bool ok = true;
  while (ok && done < total) {
    struct iovec cur[2];
    int n = 0;
    size_t skip = done;
    for (int i = 0; i < 2; ++i) {
      if (skip >= iov[i].iov_len) { skip -= iov[i].iov_len; continue; }
      cur[n].iov_base = static_cast<char *>(iov[i].iov_base) + skip;
      cur[n].iov_len = iov[i].iov_len - skip;
      skip = 0;
      ++n;
    }
    const ssize_t k = writev(fd, cur, n);
    if (k <= 0) ok = false; else done += (size_t)k;
  }
  if (close(fd) != 0) ok = false;
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
  const std::string name = "'" + opt.instance_name + "'" + d;
  const size_t max_row = name.size() + 64 + (size_t)cols * 50;
  const bool wrote = write_text_rows(f, rows, max_row, [&](int64_t t, char *p) {
    size_t n = name.size();
    std::memcpy(p, name.data(), n);
    if (opt.timestamp) n += (size_t)snprintf(p + n, 64, "%f%c", times ? times[t] : (double)t * period_sec, d);
    for (int c = 0; c < cols; ++c) {
      const float v = x[t * ld + c];
      n += (size_t)((v == std::floor(v)) ? format_f0(v, p + n) : format_e6(v, p + n));      // csvSink.cpp:224-235
      p[n++] = (c == cols - 1) ? '\n' : ';';
    }
    return n;
  });
  const bool ok = (fclose(f) == 0) && wrote;
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
  {
    const std::string name = prname ? arff_escape(opt.instance_name) + "," : std::string();
    const std::string tail = "," + (opt.class_value.empty() ? std::string("NULL") : opt.class_value) + "\n";
    const size_t max_row = name.size() + tail.size() + 64 + (size_t)cols * 50;
    const bool wrote = write_text_rows(f, rows, max_row, [&](int64_t t, char *p) {
      size_t n = name.size();
      std::memcpy(p, name.data(), n);
      if (opt.timestamp) n += (size_t)snprintf(p + n, 64, "%f,", (double)t * period_sec);
      for (int c = 0; c < cols; ++c) {
        if (c) p[n++] = ',';
        n += (size_t)format_e6(x[t * ld + c], p + n);
      }
      std::memcpy(p + n, tail.data(), tail.size());
      return n + tail.size();
    });
    const bool ok = (fclose(f) == 0) && wrote;
    if (!ok) err = "error writing '" + path + "'";
    return ok;
  }
}

}  // namespace smilehip_host
