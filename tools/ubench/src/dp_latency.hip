// Micro-benchmark: issue cost / dependent latency of FP64 and FP32 VALU operations on one wave (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 -o dp_latency dp_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(double *out, unsigned long long *ticks, double a, double b, int n) {
  double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4;
  float f0 = (float)a + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) {            // dependent mul,add chain (no fma contraction: compiled with -ffp-contract=off)
#pragma unroll
      for (int q = 0; q < 16; ++q) { x0 = x0 * b; x0 = x0 + a; }
    } else if (MODE == 1) {     // 4 independent chains
#pragma unroll
      for (int q = 0; q < 8; ++q) { x0 = x0 * b; x1 = x1 * b; x2 = x2 * b; x3 = x3 * b; x0 = x0 + a; x1 = x1 + a; x2 = x2 + a; x3 = x3 + a; }
    } else if (MODE == 2) {     // dependent fp32 chain
#pragma unroll
      for (int q = 0; q < 16; ++q) { f0 = f0 * (float)b; f0 = f0 + (float)a; }
    } else if (MODE == 3) {     // dependent fma f64
#pragma unroll
      for (int q = 0; q < 32; ++q) x0 = __builtin_fma(x0, b, a);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + f0;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
  double *d; unsigned long long *t;
  hipMalloc(&d, 1 << 20); hipMalloc(&t, 4096);
  const int n = 2000;
  const char *names[] = {"f64 dependent mul+add (32 ops/iter)", "f64 4 chains (64 ops/iter)", "f32 dependent mul+add (32 ops/iter)", "f64 dependent fma (32 ops/iter)"};
  for (int lanes : {64, 1}) {
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(lanes), 0, 0, d, t, 1.0000001, 0.9999999, n);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(lanes), 0, 0, d, t, 1.0000001, 0.9999999, n);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(lanes), 0, 0, d, t, 1.0000001, 0.9999999, n);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(lanes), 0, 0, d, t, 1.0000001, 0.9999999, n);
        hipDeviceSynchronize();
      }
      unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
      const int ops = (mode == 1) ? 64 : 32;
      printf("lanes %2d  %-40s %.2f memtime ticks/op\n", lanes, names[mode], (double)h / n / ops);
    }
  }
  // the same dependent chain with the whole device busy (one wave per SIMD, then four): issue rate under load
  for (int grid : {1024, 4096}) {
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, d, t, 1.0000001, 0.9999999, n); hipDeviceSynchronize(); }
    unsigned long long hh[8]; hipMemcpy(hh, t, 64, hipMemcpyDeviceToHost);
    printf("grid %4d x 64  f64 dependent mul+add: %.2f memtime ticks/op (block 0)\n", grid, (double)hh[0] / n / 32);
  }
  // clock: ticks per microsecond
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, t, 1.0000001, 0.9999999, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("memtime ticks per us: %.1f (kernel %.3f ms)\n", (double)h / (ms * 1e3), ms);
  return 0;
}
