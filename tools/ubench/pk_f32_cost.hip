// Issue cost of the packed FP32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two results per instruction) beside their
// scalar forms on gfx950, 8 independent chains, 4 waves per SIMD: cycles (2.4 GHz) per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 pk_f32_cost.hip -o pk_f32_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(512) k(float *out, int iters, float c) {
  float2v d[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] = float2v{threadIdx.x + 1.5f + j, threadIdx.x + 2.5f + j};
  const float2v c2 = {c, c};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      float2v &x = d[u & 7];
      if (OP == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c2));
      if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(c2));
      if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c2));
      if (OP == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x.x) : "v"(c));
      if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x.x) : "v"(c));
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += d[j].x + d[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(512), dim3(512), 0, 0, out, iters, 1.0000001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("  \"%s\": %.3f,\n", name, best * 1e6 / (double(iters) * 64 * 4) * 2.4);
}
int main() {
  float *out;
  (void)hipMalloc(&out, 512 * 512 * sizeof(float));
  printf("{\"cycles_per_wave_instruction_per_simd_at_4_waves\": {\n");
  run<0>("v_pk_mul_f32", out); run<1>("v_pk_add_f32", out); run<2>("v_pk_fma_f32", out); run<3>("v_mul_f32", out); run<4>("v_add_f32", out);
  printf("  \"_\": 0}}\n");
  return 0;
}
