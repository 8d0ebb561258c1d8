// Micro-benchmark: issue cost of the f32 MFMA forms a mel contraction could use on gfx950, cycles per instruction per
// SIMD at 1..4 waves per SIMD, next to v_fma_f32 / v_pk_fma_f32. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(float *out, int iters) {
  const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f16 d0 = {0}, d1 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {        // v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4x4 outer products, 4 independent accumulators
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
      } else {                // v_mfma_f32_32x32x2_f32, 2 independent accumulators
        d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
      }
    }
  }
  float s = c0.x + c1.y + c2.z + c3.w;
  for (int q = 0; q < 16; ++q) s += d0[q] + d1[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int per_iter) {
  float *out;
  hipMalloc(&out, 1 << 24);
  for (int wps : {1, 2, 4}) {
    const int iters = 2000, threads = 64 * 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_inst = double(iters) * per_iter;      // per wave
    printf("%-26s waves/SIMD=%d  %.2f ns per wave-instr per SIMD (= %.1f cycles at 2.4 GHz)\n", name, wps,
           ms * 1e6 / (n_inst * wps), ms * 1e6 / (n_inst * wps) * 2.4);
  }
}

int main() {
  run<0>("v_mfma_f32_4x4x1_16B_f32", 32);
  run<1>("v_mfma_f32_32x32x2_f32", 16);
  return 0;
}
