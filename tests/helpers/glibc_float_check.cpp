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

// which: 0 logf, 1 expf, 2 log10f; arguments = bit patterns lo, lo + step, ... < hi
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
    else { r = log10f(x); g = smilehip::glibc_log10f(x); }
    if (!same(r, g)) { if (!bad && first_bad) *first_bad = b; ++bad; }
  }
  return bad;
}
