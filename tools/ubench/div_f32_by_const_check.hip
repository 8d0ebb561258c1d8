// Exhaustive check of the five-operation float quotient used by lld_f0_cand's f0_shs (a / b for a divisor whose correctly rounded
// reciprocal y = RN(1 / b) is at hand: q0 = a y, two residual corrections by FMA -- the float form of f0_div_by, lld_f0.hip) against
// the division instruction sequence (IEEE, correctly rounded): every one of the 2^23 mantissas of a at the exponents given (a
// quotient's rounding does not depend on a's exponent as long as nothing under- or overflows; -100 is the guard of the kernel, the
// others are ordinary), every divisor 1 .. 32. Prints the number of differing results (must be 0).
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off div_f32_by_const_check.hip -o div_f32_by_const_check
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void check(int e, int b, unsigned long long *bad) {
  const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;          // 2^23 mantissas
  const float a = __uint_as_float(((unsigned)(e + 127) << 23) | m);
  const float fb = (float)b;
  const float y = 1.0f / fb;
  const float q0 = a * y;
  const float r0 = __builtin_fmaf(-q0, fb, a);
  const float q1 = __builtin_fmaf(r0, y, q0);
  const float r1 = __builtin_fmaf(-q1, fb, a);
  const float q = __builtin_fmaf(r1, y, q1);
  const float ref = a / fb;
  if (__float_as_uint(q) != __float_as_uint(ref)) atomicAdd(bad, 1ull);
}

int main() {
  unsigned long long *d_bad, bad = 0, total = 0;
  (void)hipMalloc(&d_bad, 8);
  (void)hipMemset(d_bad, 0, 8);
  const int exps[] = {-100, -99, -60, -1, 0, 1, 23, 60, 100, 120};
  for (int e : exps)
    for (int b = 1; b <= 32; ++b) {
      hipLaunchKernelGGL(check, dim3(1 << 15), dim3(256), 0, 0, e, b, d_bad);
      total += 1ull << 23;
    }
  (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
  printf("{\"quotients_checked\": %llu, \"divisors\": \"1..32\", \"exponents_of_a\": [-100, -99, -60, -1, 0, 1, 23, 60, 100, 120], \"differ_from_the_division\": %llu}\n", total, bad);
  return bad != 0;
}
