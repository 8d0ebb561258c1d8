// libsmilehip, C ABI part 1: errors, context life cycle, device-memory plumbing (include/smilehip.h).
#include "smilehip_internal.hpp"
#include "lld_stage.hpp"
#include "kernel_timing.hpp"

#include <map>
#include <mutex>
#include <vector>

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

extern "C" const char *smilehip_last_error(void) { return g_err.c_str(); }
extern "C" int smilehip_version(void) { return SMILEHIP_VERSION; }

// ------------------------------------------------------------- life cycle
extern "C" int smilehip_init(int device, smilehip_context **out) {
  if (!out) return fail(SMILEHIP_ERR_INVALID, "smilehip_init: null output pointer");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(SMILEHIP_ERR_NO_DEVICE, "no HIP device visible (libsmilehip has no CPU fallback)");
  if (device < 0 || device >= n) return fail(SMILEHIP_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
  auto *c = new (std::nothrow) smilehip_context();
  if (!c) return fail(SMILEHIP_ERR_NOMEM, "out of host memory");
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess) {
    delete c;
    return fail(SMILEHIP_ERR_HIP, "cannot open HIP device %d", device);
  }
  if (std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
    std::string arch = c->prop.gcnArchName;
    delete c;
    return fail(SMILEHIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                arch.c_str());
  }
  *out = c;
  return SMILEHIP_OK;
}

extern "C" void smilehip_shutdown(smilehip_context *ctx) { delete ctx; }

extern "C" int smilehip_device_name(smilehip_context *ctx, char *buf, int buflen) {
  if (!ctx || !buf || buflen <= 0) return fail(SMILEHIP_ERR_INVALID, "smilehip_device_name: bad argument");
  snprintf(buf, buflen, "%s (%s, %d CUs)", ctx->prop.name, ctx->prop.gcnArchName, ctx->prop.multiProcessorCount);
  return SMILEHIP_OK;
}

extern "C" int smilehip_alloc(smilehip_context *ctx, uint64_t bytes, void **d_ptr) {
  if (!ctx || !d_ptr) return fail(SMILEHIP_ERR_INVALID, "smilehip_alloc: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
  return SMILEHIP_OK;
}
extern "C" int smilehip_free(smilehip_context *ctx, void *d_ptr) {
  if (!ctx) return fail(SMILEHIP_ERR_INVALID, "smilehip_free: null context");
  if (d_ptr) HIP_TRY(hipFree(d_ptr));
  return SMILEHIP_OK;
}
extern "C" int smilehip_copy_to_device(smilehip_context *ctx, void *d_dst, const void *h_src, uint64_t bytes, void *stream) {
  if (!ctx || (bytes && (!d_dst || !h_src))) return fail(SMILEHIP_ERR_INVALID, "smilehip_copy_to_device: null argument");
  if (bytes) HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return SMILEHIP_OK;
}
extern "C" int smilehip_copy_to_host(smilehip_context *ctx, void *h_dst, const void *d_src, uint64_t bytes, void *stream) {
  if (!ctx || (bytes && (!h_dst || !d_src))) return fail(SMILEHIP_ERR_INVALID, "smilehip_copy_to_host: null argument");
  if (bytes) HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return SMILEHIP_OK;
}
// rows x width_bytes between a host matrix and a device matrix with their own row pitches (one level of the data memory is
// [frames x N] floats; a field of it, or a compact device block, has another pitch)
extern "C" int smilehip_copy_to_device_2d(smilehip_context *ctx, void *d_dst, uint64_t d_pitch, const void *h_src, uint64_t h_pitch,
                                          uint64_t width_bytes, uint64_t rows, void *stream) {
  if (!ctx || d_pitch < width_bytes || h_pitch < width_bytes || (rows && width_bytes && (!d_dst || !h_src)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_copy_to_device_2d: bad argument");
  if (!rows || !width_bytes) return SMILEHIP_OK;
  if (d_pitch == width_bytes && h_pitch == width_bytes)
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, width_bytes * rows, hipMemcpyHostToDevice, (hipStream_t)stream));
  else
    HIP_TRY(hipMemcpy2DAsync(d_dst, d_pitch, h_src, h_pitch, width_bytes, rows, hipMemcpyHostToDevice, (hipStream_t)stream));
  return SMILEHIP_OK;
}
extern "C" int smilehip_copy_to_host_2d(smilehip_context *ctx, void *h_dst, uint64_t h_pitch, const void *d_src, uint64_t d_pitch,
                                        uint64_t width_bytes, uint64_t rows, void *stream) {
  if (!ctx || d_pitch < width_bytes || h_pitch < width_bytes || (rows && width_bytes && (!h_dst || !d_src)))
    return fail(SMILEHIP_ERR_INVALID, "smilehip_copy_to_host_2d: bad argument");
  if (!rows || !width_bytes) return SMILEHIP_OK;
  if (d_pitch == width_bytes && h_pitch == width_bytes)
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, width_bytes * rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
  else
    HIP_TRY(hipMemcpy2DAsync(h_dst, h_pitch, d_src, d_pitch, width_bytes, rows, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return SMILEHIP_OK;
}
extern "C" int smilehip_alloc_host(smilehip_context *ctx, uint64_t bytes, void **h_ptr) {
  if (!ctx || !h_ptr) return fail(SMILEHIP_ERR_INVALID, "smilehip_alloc_host: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return SMILEHIP_OK;
}
extern "C" int smilehip_free_host(smilehip_context *ctx, void *h_ptr) {
  if (!ctx) return fail(SMILEHIP_ERR_INVALID, "smilehip_free_host: null context");
  if (h_ptr) HIP_TRY(hipHostFree(h_ptr));
  return SMILEHIP_OK;
}
// page-locks host memory the caller owns (a block matrix of the plugin): copies to and from it then run at the link's rate
extern "C" int smilehip_host_register(smilehip_context *ctx, void *h_ptr, uint64_t bytes) {
  if (!ctx || !h_ptr || !bytes) return fail(SMILEHIP_ERR_INVALID, "smilehip_host_register: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipHostRegister(h_ptr, bytes, hipHostRegisterDefault));
  return SMILEHIP_OK;
}
extern "C" int smilehip_host_unregister(smilehip_context *ctx, void *h_ptr) {
  if (!ctx) return fail(SMILEHIP_ERR_INVALID, "smilehip_host_unregister: null context");
  if (h_ptr) HIP_TRY(hipHostUnregister(h_ptr));
  return SMILEHIP_OK;
}
extern "C" int smilehip_htk_rows_be(smilehip_context *ctx, const float *d_src, int64_t n, void *d_dst, void *stream) {
  if (!ctx || n < 0 || (n && (!d_src || !d_dst))) return fail(SMILEHIP_ERR_INVALID, "smilehip_htk_rows_be: bad argument");
  const hipError_t e = smilehip::stage_htk_rows_be(d_src, n, static_cast<uint32_t *>(d_dst), (hipStream_t)stream);
  if (e != hipSuccess) return fail(SMILEHIP_ERR_HIP, "smilehip_htk_rows_be: %s", hipGetErrorString(e));
  return SMILEHIP_OK;
}
extern "C" int smilehip_stream_create(smilehip_context *ctx, void **stream) {
  if (!ctx || !stream) return fail(SMILEHIP_ERR_INVALID, "smilehip_stream_create: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return SMILEHIP_OK;
}
extern "C" int smilehip_stream_destroy(smilehip_context *ctx, void *stream) {
  if (!ctx) return fail(SMILEHIP_ERR_INVALID, "smilehip_stream_destroy: null context");
  if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return SMILEHIP_OK;
}
extern "C" int smilehip_event_create(smilehip_context *ctx, void **event) {
  if (!ctx || !event) return fail(SMILEHIP_ERR_INVALID, "smilehip_event_create: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  hipEvent_t e = nullptr;
  HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = e;
  return SMILEHIP_OK;
}
extern "C" int smilehip_event_destroy(smilehip_context *ctx, void *event) {
  if (!ctx) return fail(SMILEHIP_ERR_INVALID, "smilehip_event_destroy: null context");
  if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event));
  return SMILEHIP_OK;
}
extern "C" int smilehip_event_record(smilehip_context *ctx, void *event, void *stream) {
  if (!ctx || !event) return fail(SMILEHIP_ERR_INVALID, "smilehip_event_record: null argument");
  HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return SMILEHIP_OK;
}
extern "C" int smilehip_stream_wait_event(smilehip_context *ctx, void *stream, void *event) {
  if (!ctx || !event) return fail(SMILEHIP_ERR_INVALID, "smilehip_stream_wait_event: null argument");
  HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return SMILEHIP_OK;
}
extern "C" int smilehip_stream_synchronize(smilehip_context *ctx, void *stream) {
  if (!ctx) return fail(SMILEHIP_ERR_INVALID, "smilehip_stream_synchronize: null context");
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return SMILEHIP_OK;
}

// ------------------------------------------------------------------ block cache of the batches' and plans' device memory
#include <unordered_map>
namespace smilehip {
namespace {
std::mutex g_dc_mu;
size_t g_dc_limit = 0, g_dc_held = 0;
std::unordered_map<void *, size_t> g_dc_size;             // blocks handed out while the cache is on
std::multimap<size_t, void *> g_dc_free;                  // blocks kept, by size
}  // namespace
hipError_t dev_malloc(void **p, size_t bytes) {
  bytes = bytes ? bytes : 1;
  {
    std::lock_guard<std::mutex> lock(g_dc_mu);
    if (g_dc_limit) {
      auto it = g_dc_free.find(bytes);
      if (it != g_dc_free.end()) {
        *p = it->second;
        g_dc_free.erase(it);
        g_dc_held -= bytes;
        g_dc_size[*p] = bytes;
        return hipSuccess;
      }
    }
  }
  const hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(g_dc_mu);
    if (g_dc_limit) g_dc_size[*p] = bytes;
  }
  return e;
}
void dev_free(void *p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(g_dc_mu);
    auto it = g_dc_size.find(p);
    if (it != g_dc_size.end()) {
      const size_t b = it->second;
      g_dc_size.erase(it);
      if (g_dc_limit && g_dc_held + b <= g_dc_limit) {
        g_dc_free.emplace(b, p);
        g_dc_held += b;
        return;
      }
    }
  }
  (void)hipFree(p);
}
namespace {
constexpr size_t kStageBytes = (size_t)4 << 20;
std::mutex g_up_mu;
void *g_up_stage = nullptr;
hipStream_t g_up_stream = nullptr;
__global__ void k_upload_words(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace
hipError_t dev_upload(void *d_dst, const void *h_src, size_t bytes) {
  if (!bytes) return hipSuccess;
  bool fast;
  {
    std::lock_guard<std::mutex> lock(g_dc_mu);
    fast = g_dc_limit != 0;
  }
  if (!fast || bytes > kStageBytes || (bytes & 3)) return hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice);
  std::lock_guard<std::mutex> lock(g_up_mu);
  hipError_t e;
  if (!g_up_stage) {
    if ((e = hipHostMalloc(&g_up_stage, kStageBytes, hipHostMallocDefault)) != hipSuccess) { g_up_stage = nullptr; return hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice); }
    if ((e = hipStreamCreateWithFlags(&g_up_stream, hipStreamNonBlocking)) != hipSuccess) return e;
  }
  memcpy(g_up_stage, h_src, bytes);
  const size_t n = bytes / 4;
  const unsigned grid = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(k_upload_words, dim3(grid), dim3(256), 0, g_up_stream, static_cast<const uint32_t *>(g_up_stage), static_cast<uint32_t *>(d_dst), n);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return hipStreamSynchronize(g_up_stream);
}
}  // namespace smilehip

extern "C" int smilehip_alloc_cache(smilehip_context *ctx, uint64_t bytes) {
  if (!ctx) return fail(SMILEHIP_ERR_INVALID, "smilehip_alloc_cache: null context");
  std::vector<void *> drop;
  {
    std::lock_guard<std::mutex> lock(smilehip::g_dc_mu);
    smilehip::g_dc_limit = (size_t)bytes;
    while (!smilehip::g_dc_free.empty() && smilehip::g_dc_held > smilehip::g_dc_limit) {
      auto it = std::prev(smilehip::g_dc_free.end());
      smilehip::g_dc_held -= it->first;
      drop.push_back(it->second);
      smilehip::g_dc_free.erase(it);
    }
    if (!bytes) smilehip::g_dc_size.clear();
  }
  for (void *p : drop) (void)hipFree(p);
  return SMILEHIP_OK;
}

// ------------------------------------------------------------------ live per-kernel timing (kernel_timing.hpp)
namespace smilehip {
bool g_kernel_timing_on = false;
namespace {
struct KRec { const char *name; hipEvent_t a, b; };
std::mutex g_kt_mu;
std::vector<KRec> g_kt_recs;
std::vector<hipEvent_t> g_kt_pool;
thread_local std::vector<size_t> g_kt_open;               // records begun on this thread and not ended yet (launches do not nest across threads)
hipEvent_t kt_event() {
  if (!g_kt_pool.empty()) { hipEvent_t e = g_kt_pool.back(); g_kt_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace
void kernel_mark_begin(const char *name, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_kt_mu);
  KRec r{name, kt_event(), kt_event()};
  if (r.a) (void)hipEventRecord(r.a, s);
  g_kt_recs.push_back(r);
  g_kt_open.push_back(g_kt_recs.size() - 1);
}
void kernel_mark_end(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_kt_mu);
  if (g_kt_open.empty()) return;
  const size_t i = g_kt_open.back();
  g_kt_open.pop_back();
  if (i < g_kt_recs.size() && g_kt_recs[i].b) (void)hipEventRecord(g_kt_recs[i].b, s);
}
}  // namespace smilehip

extern "C" int smilehip_kernel_timing(int enable) {
  std::lock_guard<std::mutex> lock(smilehip::g_kt_mu);
  for (auto &r : smilehip::g_kt_recs) { if (r.a) smilehip::g_kt_pool.push_back(r.a); if (r.b) smilehip::g_kt_pool.push_back(r.b); }
  smilehip::g_kt_recs.clear();
  smilehip::g_kernel_timing_on = enable != 0;
  return SMILEHIP_OK;
}

extern "C" int64_t smilehip_kernel_timing_report(char *buf, int64_t buflen) {
  if (!buf || buflen < 1) return -1;
  std::lock_guard<std::mutex> lock(smilehip::g_kt_mu);
  std::map<std::string, std::pair<double, long>> sum;
  for (auto &r : smilehip::g_kt_recs) {
    float ms = 0.0f;
    if (!r.a || !r.b || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    std::string n = r.name;
    while (!n.empty() && (n.front() == '(' || n.front() == ' ')) n.erase(n.begin());
    while (!n.empty() && (n.back() == ')' || n.back() == ' ')) n.pop_back();
    auto &e = sum[n];
    e.first += ms;
    e.second += 1;
  }
  std::string out;
  char line[256];
  for (auto &kv : sum) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\n", kv.first.c_str(), kv.second.second, kv.second.first);
    out += line;
  }
  if ((int64_t)out.size() + 1 > buflen) return -(int64_t)out.size() - 1;
  memcpy(buf, out.c_str(), out.size() + 1);
  return (int64_t)out.size();
}
