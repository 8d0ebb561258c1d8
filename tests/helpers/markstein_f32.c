/* Test infrastructure: the five-operation sequence of oo_quad_irfft_even_real (opensmile_amd/csrc/lld_ooura_quad.hpp) against the
 * division, for every float significand of the dividend (three binades, both signs) and the divisors given on the command line.
 * Prints the number of dividends whose result differs from a / b. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static float seq(float a, float b, float y)
{
  const float q0 = a * y;
  const float r0 = fmaf(-q0, b, a);
  const float q1 = fmaf(r0, y, q0);
  const float r1 = fmaf(-q1, b, a);
  return fmaf(r1, y, q1);
}
int main(int argc, char **argv)
{
  long bad = 0;
  for (int k = 1; k < argc; k++) {
    const float b = strtof(argv[k], NULL), y = 1.0f / b;
    for (int e = -3; e <= 3; e += 3)
      for (uint32_t m = 0; m < (1u << 23); m++) {
        const uint32_t u = ((uint32_t)(127 + e) << 23) | m;
        float a;
        memcpy(&a, &u, 4);
        if (a / b != seq(a, b, y)) bad++;
        if (-a / b != seq(-a, b, y)) bad++;
      }
  }
  printf("%ld\n", bad);
  return 0;
}
