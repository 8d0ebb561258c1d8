// Test helper (tests/test_glibc_float.py): sweeps float arguments through opensmile_amd/csrc/glibc_float.hpp compiled for
// the host and through the real libm; returns the number of arguments whose results differ in any bit (NaNs compare equal).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../opensmile_amd/csrc/glibc_float.hpp"

static inline bool same(float a, float b) {
  uint32_t x, y;
  memcpy(&x, &a, 4); memcpy(&y, &b, 4);
  return x == y || (a != a && b != b);
}

// which: 0 logf, 1 expf, 2 log10f, 3 atanf, 4 acosf; arguments = bit patterns lo, lo + step, ... < hi
extern "C" long long glibc_float_sweep(int which, unsigned long long lo, unsigned long long hi, unsigned long long step,
                                       unsigned int *first_bad) {
  long long bad = 0;
  for (unsigned long long u = lo; u < hi; u += step) {
    uint32_t b = (uint32_t)u;
    float x;
    memcpy(&x, &b, 4);
    float r, g;
    if (which == 0) { r = logf(x); g = smilehip::glibc_logf(x); }
    else if (which == 1) { r = expf(x); g = smilehip::glibc_expf(x); }
    else if (which == 2) { r = log10f(x); g = smilehip::glibc_log10f(x); }
    else if (which == 3) { r = atanf(x); g = smilehip::glibc_atanf(x); }
    else { r = acosf(x); g = smilehip::glibc_acosf(x); }
    if (!same(r, g)) { if (!bad && first_bad) *first_bad = b; ++bad; }
  }
  return bad;
}

// atan2f on n pseudo-random pairs (xorshift seed; a third of them with close exponents, special values mixed in)
extern "C" long long glibc_atan2f_pairs(unsigned long long n, unsigned long long seed, unsigned int *first_bad_y, unsigned int *first_bad_x) {
  unsigned long long rs = seed ? seed : 88172645463325252ull;
  auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); };
  const uint32_t sp[] = {0, 0x80000000u, 0x3f800000u, 0xbf800000u, 0x7f800000u, 0xff800000u, 0x00000001u, 0x80000001u, 0x7f7fffffu, 0x00800000u, 0x34000000u};
  long long bad = 0;
  for (unsigned long long it = 0; it < n; ++it) {
    uint32_t by = rnd(), bx = rnd();
    if (it % 97 == 0) by = sp[rnd() % 11];
    if (it % 89 == 0) bx = sp[rnd() % 11];
    if (it % 3 == 0) { const uint32_t e = (((by >> 23) & 0xff) + (rnd() % 13) - 6) & 0xff; bx = (bx & 0x807fffffu) | (e << 23); }
    float y, x;
    memcpy(&y, &by, 4); memcpy(&x, &bx, 4);
    if (y != y || x != x) continue;
    if (!same(atan2f(y, x), smilehip::glibc_atan2f(y, x))) { if (!bad) { if (first_bad_y) *first_bad_y = by; if (first_bad_x) *first_bad_x = bx; } ++bad; }
  }
  return bad;
}
