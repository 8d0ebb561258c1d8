// Runs the code objects tools/ubench/valu_replay_gen.py writes (the headline kernel's own VALU stream in a counted loop) at the
// kernel's occupancy: 512 blocks x 8 waves = two blocks per CU, 4 waves per SIMD. Prints, per variant, the time the 249 500 passes
// of the bench's launch (998 000 frames / 4) take when ONLY their vector-ALU instructions are issued -- the issue-time floor --
// and the nanoseconds per VALU wave-instruction per SIMD. Build: hipcc --offload-arch=gfx950 -O2 valu_replay.hip -o valu_replay
// usage: valu_replay <dir with valu_replay_<variant>.co + valu_replay_info.json> [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int json_int(const std::string &js, const std::string &variant, const char *key) {
  size_t p = js.find("\"" + variant + "\": {");
  if (p == std::string::npos) return -1;
  p = js.find(std::string("\"") + key + "\":", p);
  return p == std::string::npos ? -1 : atoi(js.c_str() + p + strlen(key) + 3);
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp/valu_replay";
  const int iters = argc > 2 ? atoi(argv[2]) : 6100;
  std::string js;
  if (FILE *f = fopen((dir + "/valu_replay_info.json").c_str(), "r")) {
    char buf[4096]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) js.append(buf, n);
    fclose(f);
  }
  const char *variants[] = {"base", "nodft2", "nodft", "mfma16", "mfma32", "mfma16_nodft2", "mfma32_nodft",
                            "split_valu_only", "split_mfma_only", "split_both"};
  const int blocks = 512, threads = 512;                     // 2 blocks x 8 waves per CU
  const double passes_bench = 998000.0 / 4.0;
  const double passes_run = double(blocks) * (threads / 64) * iters;
  printf("{\"grid\": %d, \"block\": %d, \"iters\": %d, \"passes_per_run\": %.0f, \"variants\": {\n", blocks, threads, iters, passes_run);
  bool first = true;
  for (const char *v : variants) {
    hipModule_t mod;
    hipFunction_t fn;
    const std::string co = dir + "/valu_replay_" + v + ".co";
    if (hipModuleLoad(&mod, co.c_str()) != hipSuccess) { fprintf(stderr, "skip %s\n", co.c_str()); continue; }
    CK(hipModuleGetFunction(&fn, mod, (std::string("replay_") + v).c_str()));
    struct { int iters; int pad; } args = {iters, 0};
    size_t sz = sizeof(args);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {                      // (first: clocks, code object)
      CK(hipEventRecord(e0, 0));
      CK(hipModuleLaunchKernel(fn, blocks, 1, 1, threads, 1, 1, 0, 0, nullptr, cfg));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    const int n_valu = json_int(js, v, "valu"), n_mfma = json_int(js, v, "mfma");
    const double ms_bench = best * passes_bench / passes_run;
    const double ns_per_inst = n_valu > 0 ? best * 1e6 / (double(iters) * 4 /* waves per SIMD */ * n_valu) : 0.0;
    printf("%s \"%s\": {\"ms_run\": %.4f, \"ms_per_bench_launch\": %.4f, \"valu_per_pass\": %d, \"mfma_per_pass\": %d, "
           "\"ns_per_valu_inst_per_simd\": %.4f, \"cycles_at_2.4GHz\": %.3f}", first ? "" : ",\n", v, best, ms_bench, n_valu, n_mfma,
           ns_per_inst, ns_per_inst * 2.4);
    first = false;
    CK(hipModuleUnload(mod));
  }
  printf("\n}}\n");
  return 0;
}
