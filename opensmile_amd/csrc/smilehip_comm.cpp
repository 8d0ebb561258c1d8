// libsmilehip_comm.so: RCCL gather of result rows to rank 0 (include/smilehip_comm.h).
#include "../../include/smilehip_comm.h"

#include <arpa/inet.h>
#include <hip/hip_runtime.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <poll.h>
#include <unistd.h>

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {
thread_local char g_err[512] = "";
int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
bool send_all(int fd, const void *p, size_t n) {
  const char *c = static_cast<const char *>(p);
  while (n) {
    const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) return false;
    c += k; n -= (size_t)k;
  }
  return true;
}
bool recv_all(int fd, void *p, size_t n) {
  char *c = static_cast<char *>(p);
  while (n) {
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) return false;
    c += k; n -= (size_t)k;
  }
  return true;
}
}  // namespace

struct smilehip_comm {
  ncclComm_t nccl = nullptr;
  int rank = 0, world = 1, device = 0;
  int64_t *d_counts = nullptr;       // [world + 1] device scratch of the count exchange
  hipStream_t piece_stream = nullptr; // the pieces of a chunked gather travel here, beside the caller's kernels
  hipEvent_t piece_ready = nullptr, piece_done = nullptr;
};

extern "C" const char *smilehip_comm_last_error(void) { return g_err; }

extern "C" int smilehip_comm_bootstrap_bcast(int rank, int world, const char *master_addr, int master_port, void *buf, int32_t len) {
  if (world <= 1) return 0;
  if (!master_addr || master_port <= 0 || !buf || len <= 0 || rank < 0 || rank >= world) return fail("bootstrap: bad argument");
  sockaddr_in sa;
  memset(&sa, 0, sizeof(sa));
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)master_port);
  if (inet_pton(AF_INET, master_addr, &sa.sin_addr) != 1) return fail("bootstrap: '%s' is not an IPv4 address", master_addr);
  if (rank == 0) {
    const int ls = socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) return fail("bootstrap: socket: %s", strerror(errno));
    const int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (bind(ls, reinterpret_cast<sockaddr *>(&sa), sizeof(sa)) != 0 || listen(ls, world) != 0) {
      const int e = errno;
      close(ls);
      return fail("bootstrap: cannot listen on %s:%d: %s", master_addr, master_port, strerror(e));
    }
    std::vector<char> seen((size_t)world, 0);
    for (int k = 1; k < world; ++k) {
      // a peer that died before connecting must not hang rank 0 forever: wait at most two minutes per peer
      pollfd pf{ls, POLLIN, 0};
      const int pr = poll(&pf, 1, 120 * 1000);
      if (pr <= 0) { close(ls); return fail("bootstrap: %d of %d peers connected within 120 s", k - 1, world - 1); }
      const int fd = accept(ls, nullptr, nullptr);
      int32_t peer = -1;
      const bool ok = fd >= 0 && recv_all(fd, &peer, sizeof(peer)) && peer > 0 && peer < world && !seen[(size_t)peer] && send_all(fd, buf, (size_t)len);
      if (fd >= 0) close(fd);
      if (!ok) { close(ls); return fail("bootstrap: hand-over to a peer failed (rank %d)", (int)peer); }
      seen[(size_t)peer] = 1;
    }
    close(ls);
    return 0;
  }
  for (int attempt = 0; attempt < 600; ++attempt) {         // rank 0 may still be starting: retry for a minute
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return fail("bootstrap: socket: %s", strerror(errno));
    if (connect(fd, reinterpret_cast<sockaddr *>(&sa), sizeof(sa)) == 0) {
      const int32_t me = rank;
      const bool ok = send_all(fd, &me, sizeof(me)) && recv_all(fd, buf, (size_t)len);
      close(fd);
      return ok ? 0 : fail("bootstrap: rank 0 closed the connection");
    }
    close(fd);
    usleep(100 * 1000);
  }
  return fail("bootstrap: rank 0 not reachable at %s:%d", master_addr, master_port);
}

static int comm_join(int device, int rank, int world, const ncclUniqueId &id, smilehip_comm **out) {
  smilehip_comm *c = new smilehip_comm;
  c->rank = rank; c->world = world; c->device = device;
  const ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) { delete c; return fail("ncclCommInitRank: %s", ncclGetErrorString(r)); }
  if (hipMalloc(&c->d_counts, sizeof(int64_t) * (size_t)(world + 1)) != hipSuccess ||
      hipStreamCreateWithFlags(&c->piece_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->piece_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->piece_done, hipEventDisableTiming) != hipSuccess) {
    smilehip_comm_destroy(c);
    return fail("smilehip_comm_create: device resources (count scratch, piece stream, events) failed");
  }
  *out = c;
  return 0;
}

extern "C" int smilehip_comm_create(int device, int rank, int world, const char *master_addr, int master_port, smilehip_comm **out) {
  if (!out || world < 1 || rank < 0 || rank >= world) return fail("smilehip_comm_create: bad argument");
  if (hipSetDevice(device) != hipSuccess) return fail("smilehip_comm_create: hipSetDevice(%d) failed", device);
  ncclUniqueId id;
  memset(&id, 0, sizeof(id));
  if (rank == 0 && ncclGetUniqueId(&id) != ncclSuccess) return fail("ncclGetUniqueId failed");
  if (smilehip_comm_bootstrap_bcast(rank, world, master_addr, master_port, &id, (int32_t)sizeof(id)) != 0) return -1;
  return comm_join(device, rank, world, id, out);
}

static_assert(sizeof(ncclUniqueId) == SMILEHIP_COMM_ID_BYTES, "SMILEHIP_COMM_ID_BYTES");

extern "C" int smilehip_comm_unique_id(void *id) {
  if (!id) return fail("smilehip_comm_unique_id: null argument");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return fail("ncclGetUniqueId failed");
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int smilehip_comm_create_from_id(int device, int rank, int world, const void *id, smilehip_comm **out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail("smilehip_comm_create_from_id: bad argument");
  if (hipSetDevice(device) != hipSuccess) return fail("smilehip_comm_create_from_id: hipSetDevice(%d) failed", device);
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  return comm_join(device, rank, world, u, out);
}

extern "C" int smilehip_comm_destroy(smilehip_comm *c) {
  if (!c) return 0;
  if (c->piece_stream) { (void)hipStreamSynchronize(c->piece_stream); (void)hipStreamDestroy(c->piece_stream); }
  if (c->piece_ready) (void)hipEventDestroy(c->piece_ready);
  if (c->piece_done) (void)hipEventDestroy(c->piece_done);
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->nccl) ncclCommDestroy(c->nccl);
  delete c;
  return 0;
}

extern "C" int smilehip_comm_allgather_count(smilehip_comm *c, int64_t n, int64_t *counts, void *stream) {
  if (!c || !counts) return fail("smilehip_comm_allgather_count: null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(c->d_counts + c->world, &n, sizeof(n), hipMemcpyHostToDevice, s) != hipSuccess) return fail("count upload failed");
  const ncclResult_t r = ncclAllGather(c->d_counts + c->world, c->d_counts, 1, ncclInt64, c->nccl, s);
  if (r != ncclSuccess) return fail("ncclAllGather: %s", ncclGetErrorString(r));
  if (hipMemcpyAsync(counts, c->d_counts, sizeof(int64_t) * (size_t)c->world, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return fail("count download failed");
  return 0;
}

extern "C" int smilehip_comm_gather_rows(smilehip_comm *c, const float *d_rows, const int64_t *counts, int32_t cols, float *d_all, void *stream) {
  if (!c || !counts || cols <= 0) return fail("smilehip_comm_gather_rows: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t mine = (size_t)counts[c->rank] * (size_t)cols;
  // every argument is checked BEFORE the group opens: a return between ncclGroupStart and ncclGroupEnd would leave the
  // communicator inside an open group
  if (c->rank == 0 && !d_all) return fail("smilehip_comm_gather_rows: rank 0 needs d_all");
  if (mine && !d_rows) return fail("smilehip_comm_gather_rows: rank %d has %lld rows but no d_rows", c->rank, (long long)counts[c->rank]);
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return fail("ncclGroupStart: %s", ncclGetErrorString(r));
  if (c->rank == 0) {
    size_t off = (size_t)counts[0] * (size_t)cols;
    for (int p = 1; p < c->world; ++p) {
      const size_t n = (size_t)counts[p] * (size_t)cols;
      if (n && (r = ncclRecv(d_all + off, n, ncclFloat, p, c->nccl, s)) != ncclSuccess) break;
      off += n;
    }
  } else if (mine) {
    r = ncclSend(d_rows, mine, ncclFloat, 0, c->nccl, s);
  }
  const ncclResult_t r2 = ncclGroupEnd();
  if (r != ncclSuccess || r2 != ncclSuccess) return fail("gather: %s", ncclGetErrorString(r != ncclSuccess ? r : r2));
  if (c->rank == 0 && mine && hipMemcpyAsync(d_all, d_rows, mine * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return fail("gather: local copy failed");
  if (!s && hipStreamSynchronize(s) != hipSuccess) return fail("gather: synchronize failed");      // null stream: blocking call
  return 0;
}

extern "C" int64_t smilehip_comm_gather_pieces(const int64_t *counts, int world, int64_t piece_rows) {
  if (!counts || world < 1 || piece_rows <= 0) return -1;
  int64_t mx = 0;
  for (int r = 0; r < world; ++r) mx = counts[r] > mx ? counts[r] : mx;
  return (mx + piece_rows - 1) / piece_rows;
}

extern "C" int smilehip_comm_piece_rows(const int64_t *counts, int world, int64_t piece_rows, int64_t k, int rank, int64_t *first, int64_t *n,
                                        int64_t *dst_row) {
  if (!counts || world < 1 || piece_rows <= 0 || k < 0 || rank < 0 || rank >= world || !first || !n || !dst_row)
    return fail("smilehip_comm_piece_rows: bad argument");
  int64_t block0 = 0;
  for (int r = 0; r < rank; ++r) block0 += counts[r];
  *first = k * piece_rows;
  int64_t m = counts[rank] - *first;
  *n = m < 0 ? 0 : (m > piece_rows ? piece_rows : m);
  *dst_row = block0 + *first;
  return 0;
}

extern "C" int smilehip_comm_gather_rows_piece(smilehip_comm *c, const float *d_rows, const int64_t *counts, int32_t cols, float *d_all,
                                               int64_t piece_rows, int64_t k, void *after_stream) {
  if (!c || !counts || cols <= 0 || piece_rows <= 0 || k < 0) return fail("smilehip_comm_gather_rows_piece: bad argument");
  if (c->rank == 0 && !d_all) return fail("smilehip_comm_gather_rows_piece: rank 0 needs d_all");
  int64_t my_first = 0, my_n = 0, my_dst = 0;
  if (smilehip_comm_piece_rows(counts, c->world, piece_rows, k, c->rank, &my_first, &my_n, &my_dst) != 0) return -1;
  if (my_n && !d_rows) return fail("smilehip_comm_gather_rows_piece: rank %d has rows in piece %lld but no d_rows", c->rank, (long long)k);
  hipStream_t ps = c->piece_stream;
  if (after_stream) {                                                  // the piece's rows are written by work enqueued there
    if (hipEventRecord(c->piece_ready, static_cast<hipStream_t>(after_stream)) != hipSuccess ||
        hipStreamWaitEvent(ps, c->piece_ready, 0) != hipSuccess)
      return fail("smilehip_comm_gather_rows_piece: event hand-over failed");
  }
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return fail("ncclGroupStart: %s", ncclGetErrorString(r));
  if (c->rank == 0) {
    for (int p = 1; p < c->world; ++p) {
      int64_t first = 0, n = 0, dst = 0;
      (void)smilehip_comm_piece_rows(counts, c->world, piece_rows, k, p, &first, &n, &dst);
      if (n && (r = ncclRecv(d_all + (size_t)dst * (size_t)cols, (size_t)n * (size_t)cols, ncclFloat, p, c->nccl, ps)) != ncclSuccess) break;
    }
  } else if (my_n) {
    r = ncclSend(d_rows + (size_t)my_first * (size_t)cols, (size_t)my_n * (size_t)cols, ncclFloat, 0, c->nccl, ps);
  }
  const ncclResult_t r2 = ncclGroupEnd();
  if (r != ncclSuccess || r2 != ncclSuccess) return fail("gather piece: %s", ncclGetErrorString(r != ncclSuccess ? r : r2));
  if (c->rank == 0 && my_n &&
      hipMemcpyAsync(d_all + (size_t)my_dst * (size_t)cols, d_rows + (size_t)my_first * (size_t)cols, (size_t)my_n * (size_t)cols * sizeof(float),
                     hipMemcpyDeviceToDevice, ps) != hipSuccess)
    return fail("gather piece: local copy failed");
  return 0;
}

extern "C" int smilehip_comm_gather_wait(smilehip_comm *c, void *stream) {
  if (!c) return fail("smilehip_comm_gather_wait: null argument");
  if (!stream) return hipStreamSynchronize(c->piece_stream) == hipSuccess ? 0 : fail("gather wait: synchronize failed");
  if (hipEventRecord(c->piece_done, c->piece_stream) != hipSuccess ||
      hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->piece_done, 0) != hipSuccess)
    return fail("gather wait: event hand-over failed");
  return 0;
}

// The point-to-point primitives of the gather on ONE device: a grouped ncclSend / ncclRecv pair whose peer is the caller's own
// rank (RCCL pairs them inside the group). What a box with a single GPU can execute of smilehip_comm_gather_rows' send / receive
// path -- no xGMI link is crossed; tests/test_gpu_comm.py.
extern "C" int smilehip_comm_self_sendrecv(smilehip_comm *c, const float *d_src, float *d_dst, int64_t n, void *stream) {
  if (!c || !d_src || !d_dst || n <= 0) return fail("smilehip_comm_self_sendrecv: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return fail("ncclGroupStart: %s", ncclGetErrorString(r));
  const ncclResult_t rs = ncclSend(d_src, (size_t)n, ncclFloat, c->rank, c->nccl, s);
  const ncclResult_t rr = ncclRecv(d_dst, (size_t)n, ncclFloat, c->rank, c->nccl, s);
  const ncclResult_t r2 = ncclGroupEnd();
  if (rs != ncclSuccess || rr != ncclSuccess || r2 != ncclSuccess)
    return fail("self send/recv: %s", ncclGetErrorString(rs != ncclSuccess ? rs : (rr != ncclSuccess ? rr : r2)));
  if (!s && hipStreamSynchronize(s) != hipSuccess) return fail("self send/recv: synchronize failed");
  return 0;
}

extern "C" int smilehip_comm_rccl_version(int *version) {
  if (!version) return fail("smilehip_comm_rccl_version: null argument");
  return ncclGetVersion(version) == ncclSuccess ? 0 : fail("ncclGetVersion failed");
}
