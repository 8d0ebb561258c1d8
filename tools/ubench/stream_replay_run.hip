// Runs the code objects of tools/ubench/stream_replay_gen.py: stream_replay_run <dir> <name> <blocks> <threads> <iters> <valu per iteration>
// Prints, for <name>_valu and <name>_scal, the time and the cycles (2.4 GHz) per vector instruction per SIMD (waves per SIMD =
// blocks x threads / 64 / 1024 when the grid is one resident round).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  const std::string dir = argv[1], name = argv[2];
  const int blocks = atoi(argv[3]), threads = atoi(argv[4]), iters = atoi(argv[5]), n_valu = atoi(argv[6]);
  void *buf;
  CK(hipMalloc(&buf, 1 << 20));
  CK(hipMemset(buf, 0, 1 << 20));
  for (const char *v : {"valu", "scal"}) {   // (a missing code object is skipped)
    hipModule_t mod;
    hipFunction_t fn;
    const std::string kn = name + "_" + v;
    if (hipModuleLoad(&mod, (dir + "/" + kn + ".co").c_str()) != hipSuccess) { fprintf(stderr, "skip %s\n", kn.c_str()); continue; }
    CK(hipModuleGetFunction(&fn, mod, kn.c_str()));
    struct { int iters; int pad; void *p; } args = {iters, 0, buf};
    size_t sz = sizeof(args);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, 0));
      CK(hipModuleLaunchKernel(fn, blocks, 1, 1, threads, 1, 1, 0, 0, nullptr, cfg));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    const double waves_per_simd = double(blocks) * (threads / 64) / 1024.0;
    printf("{\"kernel\": \"%s\", \"blocks\": %d, \"threads\": %d, \"iters\": %d, \"ms\": %.4f, \"waves_per_simd\": %.2f, \"valu_per_iter\": %d, "
           "\"cycles_per_valu_per_simd\": %.3f}\n", kn.c_str(), blocks, threads, iters, best, waves_per_simd, n_valu,
           best * 1e-3 * 2.4e9 / (double(iters) * n_valu * waves_per_simd));
    CK(hipModuleUnload(mod));
  }
  return 0;
}
