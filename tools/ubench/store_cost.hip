// What does a coalesced global store cost a wave that goes on computing? 16 waves per CU (256 workgroups x 1024 threads), every wave
// loops: V dependent double-precision FMAs (V = 64 .. 1024), then S (0, 1, 4) stores of 64 x 16 bytes = one contiguous kilobyte each,
// to the wave's own 128 KB region (or always to the same 4 KB). Prints cycles (2.4 GHz) per iteration per wave and the difference
// to S = 0. Build: hipcc --offload-arch=gfx950 -O2 store_cost.hip -o store_cost
#include <hip/hip_runtime.h>

#include <cstdio>

template <int S, bool SAME>
__global__ void __launch_bounds__(1024) k(float4 *out, int iters, int v64, double c) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 *base = out + ((size_t)(blockIdx.x * 16 + wave) * 32) * 256 + lane;
  double d0 = lane + 1.5, d1 = lane + 2.5;
  for (int it = 0; it < iters; ++it) {
    for (int r = 0; r < v64; ++r) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d1) : "v"(c));
      }
    }
    const float4 v = make_float4((float)d0, (float)d1, (float)it, 1.0f);
    float4 *p = base + (SAME ? 0 : (it & 31) * 256);
#pragma unroll
    for (int s = 0; s < S; ++s) p[64 * s] = v;
  }
  if (d0 + d1 == 1234.5) out[0] = make_float4(0, 0, 0, 0);
}

template <int S, bool SAME>
float run(float4 *out, int iters, int v64) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<S, SAME>), dim3(256), dim3(1024), 0, 0, out, iters, v64, 1.0000001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  float4 *out;
  hipMalloc(&out, (size_t)4096 * 32 * 256 * sizeof(float4));
  printf("[\n");
  for (int v64 : {1, 4, 16}) {
    const int iters = 4096 / v64;
    const float t0 = run<0, false>(out, iters, v64), t1 = run<1, false>(out, iters, v64), t4 = run<4, false>(out, iters, v64), t4s = run<4, true>(out, iters, v64);
    const double cyc = 2.4e6 / iters;                      // ms -> cycles per iteration (per wave: all waves run the whole kernel)
    printf("  {\"fma_per_iteration\": %d, \"cycles_per_iteration\": {\"no_store\": %.0f, \"one_store\": %.0f, \"four_stores\": %.0f, \"four_stores_same_place\": %.0f}},\n",
           64 * v64, t0 * cyc, t1 * cyc, t4 * cyc, t4s * cyc);
  }
  printf("  {}\n]\n");
  return 0;
}
