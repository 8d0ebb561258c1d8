// Exhaustive check of sqrt_rn_normal (lld_device.hpp: v_sqrt_f32 + the library's two residual tests, WITHOUT its scaling of small
// arguments and its zero / infinity test) against sqrtf for every float in [2^-96, infinity): prints the number of differing results.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off sqrt_f32_normal_check.hip -o sqrt_f32_normal_check
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float sqrt_rn_normal(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  s = (0.0f >= rm) ? sm : s;
  s = (0.0f < rp) ? sp : s;
  return s;
}
__global__ void check(unsigned base, unsigned long long *bad) {
  const unsigned bits = base + blockIdx.x * blockDim.x + threadIdx.x;
  if (bits < 0x0f800000u || bits >= 0x7f800000u) return;
  const float x = __uint_as_float(bits);
  if (__float_as_uint(sqrt_rn_normal(x)) != __float_as_uint(sqrtf(x))) atomicAdd(bad, 1ull);
}
int main() {
  unsigned long long *d_bad, bad = 0;
  (void)hipMalloc(&d_bad, 8);
  (void)hipMemset(d_bad, 0, 8);
  for (unsigned long long base = 0x0f800000ull; base < 0x7f800000ull; base += 1ull << 26)
    hipLaunchKernelGGL(check, dim3(1 << 18), dim3(256), 0, 0, (unsigned)base, d_bad);
  (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
  printf("{\"arguments_checked\": %llu, \"range\": \"[2^-96, infinity)\", \"differ_from_sqrtf\": %llu}\n", 0x7f800000ull - 0x0f800000ull, bad);
  return bad != 0;
}
