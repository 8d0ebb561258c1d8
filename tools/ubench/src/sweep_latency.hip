// Micro-benchmark: the forward sweep of the spline recurrence as lld_f0_frame runs it (eight steps per round, operands
// prefetched from LDS into alternating register sets), with 1 / 3 / 64 active lanes and uniform / per-lane array bases.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int K = 513, KP = 516, R = 8;
template <int MODE>   // 0: per-lane bases, lanes < nact active; 1: same but without LDS stores; 2: no LDS at all (register chain only)
__global__ void k(double *out, unsigned long long *ticks, int nact, int reps) {
  extern __shared__ double lds[];
  double2 *sp = reinterpret_cast<double2 *>(lds);            // [KP]
  double *Bbase = lds + 2 * KP;                               // nact arrays of KP (+pad)
  const int lane = threadIdx.x;
  for (int i = lane; i < KP; i += 64) sp[i] = make_double2(0.25 + 1e-3 * i, 0.5 - 1e-4 * i);
  for (int i = lane; i < KP * 4; i += 64) Bbase[i] = 1.0 + 1e-3 * (i % 97);
  __syncthreads();
  double up = 0.0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (lane < nact) {
    double *B = Bbase + lane * (KP + 534);                   // 8400-byte stride like the kernel's frame regions
    for (int rep = 0; rep < reps; ++rep) {
      constexpr int nfw = (K - 2) / R;
      double ca[R], cb[R];
      double2 sa[R], sb[R];
      auto ld = [&](double (&c)[R], double2 (&s)[R], int i) {
#pragma unroll
        for (int q = 0; q < R; ++q) { c[q] = (MODE == 2) ? 1.0 : B[i + q]; s[q] = (MODE == 2) ? make_double2(0.3, 0.4) : sp[i + q]; }
      };
      auto run = [&](double (&c)[R], double2 (&s)[R], int i) {
#pragma unroll
        for (int q = 0; q < R; ++q) { up = s[q].y * (c[q] - s[q].x * up); c[q] = up; }
        if (MODE == 0) {
#pragma unroll
          for (int q = 0; q < R; ++q) B[i + q] = c[q];
        }
      };
      ld(ca, sa, 1);
      int r = 0;
      for (; r + 2 <= nfw; r += 2) {
        ld(cb, sb, 1 + (r + 1) * R);
        __builtin_amdgcn_sched_barrier(0);
        run(ca, sa, 1 + r * R);
        __builtin_amdgcn_sched_barrier(0);
        if (r + 2 < nfw) ld(ca, sa, 1 + (r + 2) * R);
        __builtin_amdgcn_sched_barrier(0);
        run(cb, sb, 1 + (r + 1) * R);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (r < nfw) run(ca, sa, 1 + r * R);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = up;
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  double *d; unsigned long long *t;
  hipMalloc(&d, 1 << 22); hipMalloc(&t, 1 << 16);
  const int reps = 50;
  const size_t lds = sizeof(double) * (2 * KP + 4 * (KP + 534) + 64);
  for (int grid : {1, 1024})
    for (int nact : {1, 3, 64 > 4 ? 4 : 4}) {
      for (int mode = 0; mode < 3; ++mode) {
        for (int w = 0; w < 2; ++w) {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), lds, 0, d, t, nact, reps);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), lds, 0, d, t, nact, reps);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), lds, 0, d, t, nact, reps);
          hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        printf("grid %4d lanes %d mode %d (%s): %.1f ticks per step (3 dependent FP64 ops)\n", grid, nact, mode,
               mode == 0 ? "LDS loads + stores" : mode == 1 ? "LDS loads only" : "registers only", (double)h / reps / 504.0);
      }
    }
  return 0;
}
