// Live per-kernel timing of the batch chains (bench.py's roofline objects): when switched on (smilehip_kernel_timing), every
// launch of the batch kernels is bracketed by two HIP events ON THE STREAM THE KERNEL IS LAUNCHED ON; the report sums the elapsed
// times by kernel name. Off (the default) a launch costs one predictable branch more.
#pragma once
#include <hip/hip_runtime.h>

namespace smilehip {
void kernel_mark_begin(const char *name, hipStream_t s);
void kernel_mark_end(hipStream_t s);
extern bool g_kernel_timing_on;
struct KernelMark {
  hipStream_t s;
  bool on;
  KernelMark(const char *name, hipStream_t stream) : s(stream), on(g_kernel_timing_on) { if (on) kernel_mark_begin(name, s); }
  ~KernelMark() { if (on) kernel_mark_end(s); }
};
}  // namespace smilehip

#define SMILEHIP_KLAUNCH(kernel, grid, block, lds, stream, ...)               \
  do {                                                                       \
    smilehip::KernelMark kernel_mark_(#kernel, stream);                      \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);       \
  } while (0)
