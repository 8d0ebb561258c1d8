/* TEST INFRASTRUCTURE (tests/test_log_d_host.py): the arithmetic of log_d (opensmile_amd/csrc/lld_device.hpp) restated in C for
 * the host -- the same table (opensmile_amd/csrc/log_table.inc), the same operations in the same order, fma() where the device
 * code calls fma -- so that the algorithm's accuracy can be measured without a GPU. Compile with -ffp-contract=off. */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "log_table.inc"

static const double kTab[128][2] = {SMILEHIP_LOG_TABLE};

static double log_d_host(double x) {
  uint64_t ix;
  memcpy(&ix, &x, 8);
  if (!(ix - 0x0010000000000000ull < 0x7fe0000000000000ull)) return log(x);
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)((tmp >> 45) & 127);
  const int64_t k = (int64_t)tmp >> 52;
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  double z;
  memcpy(&z, &iz, 8);
  const double kd = (double)k;
  const double r = fma(z, kTab[i][0], -1.0);
  const double w = kd * SMILEHIP_LN2HI + kTab[i][1];
  const double hi = w + r;
  const double lo = w - hi + r + kd * SMILEHIP_LN2LO;
  double q = 1.0 / 9.0;
  q = fma(q, r, -1.0 / 8.0); q = fma(q, r, 1.0 / 7.0); q = fma(q, r, -1.0 / 6.0); q = fma(q, r, 1.0 / 5.0);
  q = fma(q, r, -1.0 / 4.0); q = fma(q, r, 1.0 / 3.0); q = fma(q, r, -0.5);
  return lo + (r * r) * q + hi;
}

void log_d_host_array(const double *in, double *out, long n) {
  for (long i = 0; i < n; ++i) out[i] = log_d_host(in[i]);
}
