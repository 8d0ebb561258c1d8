// Issue cost of the VALU instruction classes the SQ counters distinguish (SQ_INSTS_VALU_ADD_F32 / MUL_F32 / FMA_F32 / TRANS_F32 /
// CVT / INT32 / ADD_F64 / MUL_F64 / FMA_F64 / INT64 and "other": moves, selects, DPP moves, lane reads) on gfx950 at 4 waves per SIMD
// (two blocks of eight waves per CU: the frame kernels' occupancy): wall time per wave-instruction per SIMD. With the per-class
// dynamic counts of a kernel (rocprofv3 --pmc) this gives the kernel's instruction-issue floor: sum over classes of count x cost
// (bench.py: roofline.issue). Validated on lld_mfcc512, whose whole VALU stream tools/ubench/valu_replay_gen.py replays.
// Build: hipcc --offload-arch=gfx950 -O2 valu_classes.hip -o valu_classes ; prints one JSON object.
#include <hip/hip_runtime.h>

#include <cstdio>

#define B2(op, k) op " %" #k ", %" #k ", %8\n"
#define B3(op, k) op " %" #k ", %" #k ", %8, %8\n"
#define U1(op, k) op " %" #k ", %" #k "\n"
#define R8(F, op) F(op, 0) F(op, 1) F(op, 2) F(op, 3) F(op, 4) F(op, 5) F(op, 6) F(op, 7)
#define F32_OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)
#define F64_OPS "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dc)
#define I32_OPS "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(ic)

template <int MODE>
__global__ void __launch_bounds__(512) k(float *out, int iters) {
  float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  const float c = 1.0001f;
  const double dc = 1.0001;
  const int ic = 3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile(R8(B2, "v_add_f32") : F32_OPS);
      if (MODE == 1) asm volatile(R8(B2, "v_mul_f32") : F32_OPS);
      if (MODE == 2) asm volatile(R8(B3, "v_fma_f32") : F32_OPS);
      if (MODE == 3) asm volatile(R8(U1, "v_rcp_f32") : F32_OPS);
      if (MODE == 4) asm volatile(R8(U1, "v_log_f32") : F32_OPS);
      if (MODE == 5) asm volatile(R8(U1, "v_sqrt_f32") : F32_OPS);
      if (MODE == 6) asm volatile(R8(U1, "v_cvt_f32_i32") : F32_OPS);
      if (MODE == 7) asm volatile(R8(B2, "v_add_u32") : I32_OPS);
      if (MODE == 8) asm volatile(R8(B2, "v_lshlrev_b32") : I32_OPS);
      if (MODE == 9) asm volatile(R8(B2, "v_mul_lo_u32") : I32_OPS);
      if (MODE == 10) asm volatile(R8(B2, "v_add_f64") : F64_OPS);
      if (MODE == 11) asm volatile(R8(B2, "v_mul_f64") : F64_OPS);
      if (MODE == 12) asm volatile(R8(B3, "v_fma_f64") : F64_OPS);
      if (MODE == 13) asm volatile(R8(U1, "v_mov_b32") : F32_OPS);
      if (MODE == 14)
        asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %2, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %4, %5 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %6, %7 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                     : F32_OPS);
      if (MODE == 15) asm volatile(R8(B2, "v_cndmask_b32") : F32_OPS : "vcc");
      if (MODE == 16) asm volatile(R8(B2, "v_sub_f32") : F32_OPS);
      if (MODE == 17) asm volatile(R8(B2, "v_fmac_f32") : F32_OPS);
      if (MODE == 18) asm volatile(R8(U1, "v_rcp_f64") : F64_OPS);
      if (MODE == 19) asm volatile(R8(B2, "v_max_f32") : F32_OPS);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) +
                                               (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
}

template <int MODE>
void run(const char *name, const char *cls, bool last = false) {
  static float *out = nullptr;
  if (!out) hipMalloc(&out, 512 * 512 * sizeof(float));
  const int iters = 2000, blocks = 512, threads = 512;        // 2 blocks x 8 waves per CU = 4 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double n_inst = double(iters) * 64 * 4;               // per SIMD: 64 instructions per iteration and wave, 4 waves
  printf("  \"%s\": {\"class\": \"%s\", \"ns_per_inst_per_simd\": %.4f, \"cycles_at_2.4GHz\": %.3f}%s\n", name, cls, best * 1e6 / n_inst,
         best * 1e6 / n_inst * 2.4, last ? "" : ",");
}

int main() {
  printf("{\"waves_per_simd\": 4, \"ops\": {\n");
  run<0>("v_add_f32", "ADD_F32"); run<16>("v_sub_f32", "ADD_F32"); run<19>("v_max_f32", "ADD_F32?");
  run<1>("v_mul_f32", "MUL_F32"); run<2>("v_fma_f32", "FMA_F32"); run<17>("v_fmac_f32", "FMA_F32");
  run<3>("v_rcp_f32", "TRANS_F32"); run<4>("v_log_f32", "TRANS_F32"); run<5>("v_sqrt_f32", "TRANS_F32");
  run<6>("v_cvt_f32_i32", "CVT"); run<7>("v_add_u32", "INT32"); run<8>("v_lshlrev_b32", "INT32"); run<9>("v_mul_lo_u32", "INT32");
  run<10>("v_add_f64", "ADD_F64"); run<11>("v_mul_f64", "MUL_F64"); run<12>("v_fma_f64", "FMA_F64"); run<18>("v_rcp_f64", "TRANS_F64");
  run<13>("v_mov_b32", "other"); run<14>("v_mov_b32_dpp", "other"); run<15>("v_cndmask_b32", "other", true);
  printf("}}\n");
  return 0;
}
