/* smilehip_comm.h -- the one exchange step of the path: the variable-length gather of result rows
 * (functionals vectors, or LLD blocks) from every rank to rank 0, over RCCL. One process per GPU; utterances are sharded
 * with no data-path collective, so this is the only collective and it is optional (every rank can write its own files).
 * The reference has no multi-device path; what this replaces is the concatenation of per-process outputs a
 * SMILExtract-per-file deployment does on the host.
 *
 * Separate shared library (libsmilehip_comm.so, links librccl) so that single-GPU users of libsmilehip.so do not load
 * RCCL. Plain C: int status, pointers and sizes; smilehip_comm_last_error() for the message. */
#ifndef SMILEHIP_COMM_H
#define SMILEHIP_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct smilehip_comm smilehip_comm;

/* Rank 0 creates the RCCL unique id and hands it to the other ranks over TCP at master_addr:master_port (the
 * MASTER_ADDR / MASTER_PORT convention of torch.distributed.run; use 127.0.0.1 on one node), then every rank joins
 * the communicator on HIP device `device`. world = 1 is allowed (no socket is opened). */
int smilehip_comm_create(int device, int rank, int world, const char *master_addr, int master_port, smilehip_comm **out);
int smilehip_comm_destroy(smilehip_comm *c);

/* counts[r] = n of rank r, on every rank (ncclAllGather of one int64 per rank; host pointers) */
int smilehip_comm_allgather_count(smilehip_comm *c, int64_t n, int64_t *counts, void *stream);

/* Rows of `cols` floats: rank r sends its counts[r] rows (device pointer d_rows); rank 0 receives them in rank order
 * into d_all (device, sum(counts) rows; its own block is a device copy). One grouped ncclSend/ncclRecv: every peer's
 * block travels on its own xGMI link to rank 0, no ring. d_all is ignored on ranks != 0. */
int smilehip_comm_gather_rows(smilehip_comm *c, const float *d_rows, const int64_t *counts, int32_t cols, float *d_all, void *stream);

/* the TCP hand-over alone (what smilehip_comm_create uses for the unique id): rank 0 sends buf to every other rank.
 * Exported for the CPU tests. */
int smilehip_comm_bootstrap_bcast(int rank, int world, const char *master_addr, int master_port, void *buf, int32_t len);

/* One grouped ncclSend / ncclRecv pair to the caller's own rank (n floats, device pointers, non-overlapping): the gather's
 * point-to-point primitives on a single device. Diagnostic / test entry point. */
int smilehip_comm_self_sendrecv(smilehip_comm *c, const float *d_src, float *d_dst, int64_t n, void *stream);
/* RCCL's version code (ncclGetVersion) */
int smilehip_comm_rccl_version(int *version);

const char *smilehip_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
