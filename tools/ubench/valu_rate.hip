// Micro-benchmark: issue cost of f32 VALU forms on gfx950 (wave64), cycles per
// instruction per SIMD, for 1..4 waves per SIMD. Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, long long *cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float c = 1.0001f;
  const float2v pc = {1.0001f, 0.9999f};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {  // v_add_f32 x8 independent
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                     "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (MODE == 1) {  // v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                     "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (MODE == 2) {  // v_pk_add_f32
        asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                     "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
      } else if (MODE == 3) {  // v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                     "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
      } else if (MODE == 4) {  // v_mov_b32
        asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                     "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (MODE == 6) {  // dependent chain of v_add_f32
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n"
                     "v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (MODE == 7) {  // dependent chain of v_pk_add_f32
        asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %0, %0, %8\n"
                     "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %0, %0, %8\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
      } else if (MODE == 5) {  // v_mul_f32 dpp row_shr
        asm volatile("v_add_f32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name) {
  float *out; long long *cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  for (int wps : {1, 2, 3, 4, 5, 8}) {
    const int iters = 2000;
    const int threads = 64 * 4 * wps;  // one block per CU, wps waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_inst = double(iters) * 64;   // per wave
    // wall-clock based: instr issued per SIMD = n_inst * wps ; assume 2.4 GHz
    printf("%-14s waves/SIMD=%d  s_memtime cycles/instr/wave=%.2f   wall: %.2f ns per wave-instr per SIMD (=%.2f cyc @2.4GHz)\n",
           name, wps, double(c) / n_inst, ms * 1e6 / (n_inst * wps), ms * 1e6 / (n_inst * wps) * 2.4);
  }
}
int main() {
  run<0>("v_add_f32"); run<1>("v_fma_f32"); run<2>("v_pk_add_f32"); run<3>("v_pk_fma_f32"); run<4>("v_mov_b32"); run<5>("v_add_f32_dpp"); run<6>("v_add dep"); run<7>("v_pk_add dep");
  return 0;
}
