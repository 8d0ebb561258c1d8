// logf / expf / log10f as the reference's C library computes them.
//
// openSMILE calls logf (cMfcc's log-mel, cSpectral's log spectra, ...), expf and log10f of the host's glibc. Those are not
// correctly rounded (logf: <= 0.818 ulp, it rounds a double of ~2^-26 relative error), so a correctly rounded device logarithm
// differs from the reference in the last bit on a few percent of the arguments -- the only difference left between this
// library's LLD levels and the reference's once the FFT follows the reference's order. The functions below restate glibc 2.35's
// algorithms operation for operation (sysdeps/ieee754/flt-32/e_logf.c, e_expf.c: S. Nagy's table + polynomial in double, in
// the x86-64 "fma" build the dynamic linker selects on every CPU with FMA -- the products and sums below are fused exactly where
// GCC fuses them there; e_log10f.c: fdlibm's float formula around logf, built without FMA). tests/test_glibc_float.py compiles
// this header for the host and sweeps every float argument against the real libm; the tables are read out of that libm by
// tools/gen_glibc_float_tables.py.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GLF_HD __host__ __device__ __forceinline__
#else
#define GLF_HD inline
#endif

#include "glibc_float_tables.inc"

namespace smilehip {

namespace glf {
struct LogEnt { double invc, logc; };
GLF_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
GLF_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
GLF_HD uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
GLF_HD double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
#if defined(__HIP_DEVICE_COMPILE__)
__device__ const LogEnt kLogTab[16] = {GLF_LOG_TAB};
__device__ const uint64_t kExpTab[32] = {GLF_EXP_TAB};
#else
static const LogEnt kLogTab[16] = {GLF_LOG_TAB};
static const uint64_t kExpTab[32] = {GLF_EXP_TAB};
#endif
}  // namespace glf

// glibc 2.35 __logf (e_logf.c:36-85), fma build
GLF_HD float glibc_logf(float x) {
  uint32_t ix = glf::f2u(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    if (ix * 2 == 0) return -1.0f / 0.0f;                  // log(+-0) = -inf
    if (ix == 0x7f800000u) return x;                       // log(inf) = inf
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return (x - x) / (x - x);   // negative or NaN
    ix = glf::f2u(x * 0x1p23f);                            // subnormal: normalise
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) % 16u);
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & (0x1ffu << 23));
  const double invc = glf::kLogTab[i].invc, logc = glf::kLogTab[i].logc;
  const double z = (double)glf::u2f(iz);
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = __builtin_fma((double)k, (double)GLF_LOG_LN2, logc);
  const double r2 = r * r;
  double y = __builtin_fma((double)GLF_LOG_A1, r, (double)GLF_LOG_A2);
  y = __builtin_fma((double)GLF_LOG_A0, r2, y);
  y = __builtin_fma(y, r2, y0 + r);
  return (float)y;
}

// glibc 2.35 __expf (e_expf.c:38-108), fma build; the over- / underflow branches return what they return there
GLF_HD float glibc_expf(float x) {
  const double xd = (double)x;
  const uint32_t abstop = (glf::f2u(x) >> 20) & 0x7ffu;
  if (abstop >= (glf::f2u(88.0f) >> 20)) {
    if (glf::f2u(x) == glf::f2u(-1.0f / 0.0f)) return 0.0f;
    if (abstop >= (glf::f2u(1.0f / 0.0f) >> 20)) return x + x;
    if (x > 0x1.62e42ep6f) return 0x1p97f * 0x1p97f;         // overflow
    if (x < -0x1.9fe368p6f) return 0x1p-95f * 0x1p-95f;      // underflow
  }
  // z = InvLn2N * xd is only ever added to / subtracted from: GCC's FMA pass fuses the product into both uses (no rounded z)
  double kd = __builtin_fma((double)GLF_EXP_INVLN2N, xd, (double)GLF_EXP_SHIFT);
  const uint64_t ki = glf::d2u(kd);
  kd -= (double)GLF_EXP_SHIFT;
  const double r = __builtin_fma((double)GLF_EXP_INVLN2N, xd, -kd);
  uint64_t t = glf::kExpTab[ki % 32u];
  t += ki << (52 - 5);
  const double s = glf::u2d(t);
  const double zz = __builtin_fma((double)GLF_EXP_C0, r, (double)GLF_EXP_C1);
  const double r2 = r * r;
  double y = __builtin_fma((double)GLF_EXP_C2, r, 1.0);
  y = __builtin_fma(zz, r2, y);
  y = y * s;
  return (float)y;
}

// glibc 2.35 __ieee754_log10f (e_log10f.c), no FMA
GLF_HD float glibc_log10f(float x) {
  const float two25 = 3.3554432000e+07f, ivln10 = 4.3429449201e-01f, log10_2hi = 3.0102920532e-01f, log10_2lo = 7.9034151668e-07f;
  int32_t hx = (int32_t)glf::f2u(x), k = 0;
  if (hx < 0x00800000) {
    if ((hx & 0x7fffffff) == 0) return -two25 / __builtin_fabsf(x);   // log(+-0) = -inf
    if (hx < 0) return (x - x) / (x - x);
    k -= 25;
    x *= two25;
    hx = (int32_t)glf::f2u(x);
  }
  if (hx >= 0x7f800000) return x + x;
  k += (hx >> 23) - 127;
  const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
  hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
  const float y = (float)(k + i);
  x = glf::u2f((uint32_t)hx);
  const float p = y * log10_2lo, q = ivln10 * glibc_logf(x);
  const float z = p + q;
  const float w = y * log10_2hi;
  return z + w;
}

// glibc 2.35 __atanf (sysdeps/ieee754/flt-32/s_atanf.c: fdlibm's float version; the x86-64 build has no FMA variant -- plain
// mulss / addss in the installed libm) and __ieee754_atan2f (e_atan2f.c). cFFTmagphase's phase output calls atan2 on
// FLOAT_DMEM arguments (fftmagphase.cpp:264-284), i.e. atan2f, which is not correctly rounded (84 % of the values equal the
// rounded double result). Checked against the real libm: atanf for all 2^32 arguments, atan2f on 3e8 pairs (0 mismatches).
GLF_HD float glibc_atanf(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f,
                        -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f,
                        1.6285819933e-02f};
  const int32_t hx = (int32_t)glf::f2u(x), ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {                                 // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    return (hx > 0) ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {                                  // |x| < 0.4375
    if (ix < 0x31000000) return x;
    id = -1;
  } else {
    x = __builtin_fabsf(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; x = ((float)2.0 * x - 1.0f) / ((float)2.0 + x); }
      else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - (float)1.5) / (1.0f + (float)1.5 * x); }
      else { id = 3; x = -1.0f / x; }
    }
  }
  float z = x * x;
  const float w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float hi = id == 0 ? atanhi[0] : (id == 1 ? atanhi[1] : (id == 2 ? atanhi[2] : atanhi[3]));
  const float lo = id == 0 ? atanlo[0] : (id == 1 ? atanlo[1] : (id == 2 ? atanlo[2] : atanlo[3]));
  z = hi - ((x * (s1 + s2) - lo) - x);
  return (hx < 0) ? -z : z;
}
GLF_HD float glibc_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int32_t hx = (int32_t)glf::f2u(x), ix = hx & 0x7fffffff;
  const int32_t hy = (int32_t)glf::f2u(y), iy = hy & 0x7fffffff;
  if ((ix > 0x7f800000) || (iy > 0x7f800000)) return x + y;
  if (hx == 0x3f800000) return glibc_atanf(y);
  const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);   // 2 sign(x) + sign(y)
  if (iy == 0) return (m < 2) ? y : (m == 2 ? pi + tiny : -pi - tiny);
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? (float)3.0 * pi_o_4 + tiny : (float)-3.0 * pi_o_4 - tiny));
    return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
  }
  if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int32_t k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + (float)0.5 * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = glibc_atanf(__builtin_fabsf(y / x));
  if (m == 0) return z;
  if (m == 1) return -z;
  if (m == 2) return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}

// __ieee754_acosf (sysdeps/ieee754/flt-32/e_acosf.c, fdlibm's float formula, built without FMA). cLsp's `acos(xm)` on a
// FLOAT_DMEM (lsp.cpp:138, :254) is the C++ float overload, i.e. acosf. Checked against the real libm for all 2^32 arguments.
GLF_HD float glibc_acosf(float x) {
  const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f, pS0 = 1.6666667163e-01f,
              pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
              pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f,
              qS4 = 7.7038154006e-02f;
  const int32_t hx = (int32_t)glf::f2u(x), ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + (float)2.0 * pio2_lo;
  if (ix > 0x3f800000) return (x - x) / (x - x);
  if (ix < 0x3f000000) {                                  // |x| < 0.5
    if (ix <= 0x32800000) return pio2_hi + pio2_lo;
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - r * x));
  }
  if (hx < 0) {                                           // x < -0.5
    const float z = (one + x) * (float)0.5;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = __builtin_sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    return pi - (float)2.0 * (s + w);
  }
  const float z = (one - x) * (float)0.5;                 // x > 0.5
  const float s = __builtin_sqrtf(z);
  const float df = glf::u2f(glf::f2u(s) & 0xfffff000u);
  const float c = (z - df * df) / (s + df);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float w = r * s + c;
  return (float)2.0 * (df + w);
}

}  // namespace smilehip
