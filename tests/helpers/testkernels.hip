// Test helper library (tests/helpers/libsmilehip_testkernels.so, built by __graft_entry__.build(); NOT part of the product
// library): device entry points tests/test_gpu_fft.py uses to look at building blocks of the kernels in isolation -- the
// round-2 fused transform against the in-place radix-2 form (kept for SMILEHIP_FFT=radix2), and the table logarithm log_d.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "../../opensmile_amd/csrc/lld_blocks.hpp"
#include "../../opensmile_amd/csrc/lld_device.hpp"
#include "../../opensmile_amd/csrc/lld_fft.hpp"

namespace smilehip {
template <int LOGM>
__global__ void __launch_bounds__(64) fft_check_kernel(const float2 *in, const float2 *twh, float2 *out_r2, float2 *out_fused) {
  using F = WaveFft<LOGM>;
  constexpr int M = F::M;
  __shared__ float re[M], im[M];
  __shared__ float2 z[F::kZ];
  const int lane = threadIdx.x;
  const float2 *x = in + (size_t)blockIdx.x * M;
  for (int i = lane; i < M; i += 64) {
    const int r = (int)(__brev((unsigned)i) >> (32 - LOGM));
    re[r] = x[i].x;
    im[r] = x[i].y;
  }
  WaveG::sync();
  group_cfft_radix2<WaveG>(re, im, M, twh);
  for (int k = lane; k < M; k += 64) out_r2[(size_t)blockIdx.x * M + k] = make_float2(re[k], im[k]);
  F::forward(z, twh, lane, [&](int i) { return x[i]; });
  for (int k = lane; k < M; k += 64) out_fused[(size_t)blockIdx.x * M + k] = z[F::pos(k)];
}
}  // namespace smilehip

namespace smilehip {
__global__ void log_check_kernel(const double *x, double *y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = log_d(x[i]);
}
}  // namespace smilehip
// test entry: log_d (lld_device.hpp) on n host doubles
extern "C" int smilehip_debug_log_d(const double *in, double *out, int n) {
  using namespace smilehip;
  double *d_in = nullptr, *d_out = nullptr;
  int rc = -2;
  if (n > 0 && hipMalloc(&d_in, (size_t)n * 8) == hipSuccess && hipMalloc(&d_out, (size_t)n * 8) == hipSuccess &&
      hipMemcpy(d_in, in, (size_t)n * 8, hipMemcpyHostToDevice) == hipSuccess) {
    hipLaunchKernelGGL(log_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_in, d_out, n);
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
  }
  (void)hipFree(d_in); (void)hipFree(d_out);
  return rc;
}

extern "C" int smilehip_debug_fft_check(int logM, const float *in_pairs, float *out_r2, float *out_fused, int n_transforms) {
  using namespace smilehip;
  if ((logM != 8 && logM != 9) || n_transforms < 1) return -1;
  const size_t M = (size_t)1 << logM, nb = M * (size_t)n_transforms * sizeof(float2);
  std::vector<float2> twh(M / 2);
  for (size_t j = 0; j < M / 2; ++j) {
    const double a = -2.0 * M_PI * double(j) / double(M);
    twh[j] = make_float2(float(std::cos(a)), float(std::sin(a)));
  }
  float2 *d_in = nullptr, *d_tw = nullptr, *d_a = nullptr, *d_b = nullptr;
  int rc = -2;
  if (hipMalloc(&d_in, nb) == hipSuccess && hipMalloc(&d_tw, twh.size() * sizeof(float2)) == hipSuccess &&
      hipMalloc(&d_a, nb) == hipSuccess && hipMalloc(&d_b, nb) == hipSuccess &&
      hipMemcpy(d_in, in_pairs, nb, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(d_tw, twh.data(), twh.size() * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess) {
    if (logM == 9) hipLaunchKernelGGL(fft_check_kernel<9>, dim3((unsigned)n_transforms), dim3(64), 0, 0, d_in, d_tw, d_a, d_b);
    else hipLaunchKernelGGL(fft_check_kernel<8>, dim3((unsigned)n_transforms), dim3(64), 0, 0, d_in, d_tw, d_a, d_b);
    if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(out_r2, d_a, nb, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(out_fused, d_b, nb, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
  }
  (void)hipFree(d_in); (void)hipFree(d_tw); (void)hipFree(d_a); (void)hipFree(d_b);
  return rc;
}
