// Dependent-issue cost of double-precision vector instructions on gfx950: a loop of 64 v_fma_f64 / v_mul_f64 / v_add_f64 per
// iteration arranged as C independent chains (C = 1, 2, 4, 8: instruction i depends on instruction i - C), at 1, 2 and 4 waves per
// SIMD. Prints cycles (at 2.4 GHz) per wave-instruction per SIMD. If a chain's latency is L and W waves share the SIMD, the SIMD
// issues one instruction per max(issue cost, L / (C W)) cycles.
// Build: hipcc --offload-arch=gfx950 -O2 f64_dep_chain.hip -o f64_dep_chain
#include <hip/hip_runtime.h>

#include <cstdio>

template <int OP, int C>
__global__ void __launch_bounds__(512) k(double *out, int iters, double c) {
  double d[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] = threadIdx.x + 1.5 + j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      double &x = d[u % C];
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(c));
      if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(c));
      if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
      if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(reinterpret_cast<float &>(x)) : "v"((float)c));
      if (OP == 4) asm volatile("v_fma_f64 %0, -%0, %1, %0" : "+v"(x) : "s"(c));          // one operand a scalar register pair
      if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "s"(c));
      if (OP == 6) asm volatile("v_fmac_f64_e32 %0, %1, %0" : "+v"(x) : "s"(c));
      if (OP == 7) asm volatile("v_ldexp_f64 %0, %0, -2" : "+v"(x));
      if (OP == 8) asm volatile("v_fmac_f64_e32 %0, %1, %0" : "+v"(x) : "v"(c));
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += d[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int C>
void run(const char *name, double *out) {
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {
    const int threads = 64 * 4 * (wps >= 2 ? 2 : 1), blocks = 256 * (wps == 4 ? 2 : 1);   // one block per CU (two for 4 waves per SIMD)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<OP, C>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const double n_inst = double(iters) * 64 * wps;               // per SIMD
    printf("  {\"op\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"cycles_per_inst_per_simd\": %.2f, \"cycles_per_inst_per_wave\": %.2f},\n", name, C,
           wps, best * 1e6 / n_inst * 2.4, best * 1e6 / n_inst * 2.4 * wps);
  }
}

int main() {
  double *out;
  hipMalloc(&out, 512 * 512 * sizeof(double));
  printf("[\n");
  run<0, 1>("v_fma_f64", out); run<0, 2>("v_fma_f64", out); run<0, 4>("v_fma_f64", out); run<0, 8>("v_fma_f64", out);
  run<1, 1>("v_mul_f64", out); run<1, 2>("v_mul_f64", out); run<1, 8>("v_mul_f64", out);
  run<2, 1>("v_add_f64", out); run<2, 2>("v_add_f64", out); run<2, 8>("v_add_f64", out);
  run<3, 1>("v_fma_f32", out); run<3, 8>("v_fma_f32", out);
  run<4, 2>("v_fma_f64 sgpr", out); run<4, 8>("v_fma_f64 sgpr", out); run<5, 2>("v_mul_f64 sgpr", out); run<5, 8>("v_mul_f64 sgpr", out);
  run<6, 2>("v_fmac_f64 sgpr", out); run<6, 8>("v_fmac_f64 sgpr", out); run<7, 8>("v_ldexp_f64", out); run<8, 8>("v_fmac_f64 vgpr", out);
  printf("  {}\n]\n");
  return 0;
}
