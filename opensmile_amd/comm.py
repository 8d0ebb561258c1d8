"""ctypes binding of libsmilehip_comm.so (include/smilehip_comm.h): the path's one exchange step, the gather of result rows to rank 0
over RCCL. One process per GPU; torch.distributed is used for nothing but the rendezvous -- rank 0's RCCL unique id
travels to the other ranks through the process group the launcher set up (`Comm.from_process_group`), the transfers themselves are
the library's grouped ncclSend / ncclRecv.

`PieceGather` is the call order of the gather in pieces (piece k of every rank, k = 0, 1, ...; then wait) over either transport:
the RCCL library on devices, or -- for the world-size-2 CPU tests, where no device exists -- torch.distributed point-to-point on
the gloo backend, moving exactly the rows `smilehip_comm_piece_rows` (the library's own host arithmetic) names."""
import ctypes as C
import os

import numpy as np

_LIB = None
ID_BYTES = 128
_vp, _i64 = C.c_void_p, C.c_int64


class CommError(RuntimeError):
    pass


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsmilehip_comm.so")
        if not os.path.exists(path):
            raise CommError(f"{path} is not built (make -C opensmile_amd/csrc)")
        L = C.CDLL(path)
        L.smilehip_comm_last_error.restype = C.c_char_p
        L.smilehip_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(_vp)]
        L.smilehip_comm_unique_id.argtypes = [_vp]
        L.smilehip_comm_create_from_id.argtypes = [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp)]
        L.smilehip_comm_destroy.argtypes = [_vp]
        L.smilehip_comm_allgather_count.argtypes = [_vp, _i64, _vp, _vp]
        L.smilehip_comm_gather_rows.argtypes = [_vp, _vp, _vp, C.c_int32, _vp, _vp]
        L.smilehip_comm_gather_pieces.argtypes = [_vp, C.c_int, _i64]
        L.smilehip_comm_gather_pieces.restype = _i64
        L.smilehip_comm_piece_rows.argtypes = [_vp, C.c_int, _i64, _i64, C.c_int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]
        L.smilehip_comm_gather_rows_piece.argtypes = [_vp, _vp, _vp, C.c_int32, _vp, _i64, _i64, _vp]
        L.smilehip_comm_gather_wait.argtypes = [_vp, _vp]
        L.smilehip_comm_self_sendrecv.argtypes = [_vp, _vp, _vp, _i64, _vp]
        L.smilehip_comm_rccl_version.argtypes = [C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise CommError((load().smilehip_comm_last_error() or b"").decode())


def n_pieces(counts, piece_rows):
    c = np.ascontiguousarray(counts, np.int64)
    n = load().smilehip_comm_gather_pieces(c.ctypes.data, len(c), piece_rows)
    if n < 0:
        raise CommError("smilehip_comm_gather_pieces: bad argument")
    return int(n)


def piece_rows_of(counts, piece_rows, k, rank):
    """(first row of the rank's block, rows, first row in the gathered matrix) of piece k: the library's own arithmetic"""
    c = np.ascontiguousarray(counts, np.int64)
    first, n, dst = _i64(), _i64(), _i64()
    _check(load().smilehip_comm_piece_rows(c.ctypes.data, len(c), piece_rows, k, rank, C.byref(first), C.byref(n), C.byref(dst)))
    return first.value, n.value, dst.value


class Comm:
    """one RCCL communicator of `world` ranks, this process being `rank` on HIP device `device`"""

    def __init__(self, handle, rank, world):
        self._h, self.rank, self.world = handle, rank, world

    @classmethod
    def from_process_group(cls, dist, device, group=None):
        """rendezvous through an initialised torch.distributed process group (any backend): rank 0's unique id is broadcast as a
        Python object, then every rank joins"""
        L = load()
        rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist is not None else (0, 1)
        ident = C.create_string_buffer(ID_BYTES)
        if rank == 0:
            _check(L.smilehip_comm_unique_id(ident))
        if world > 1:
            box = [ident.raw if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = C.create_string_buffer(box[0], ID_BYTES)
        h = _vp()
        _check(L.smilehip_comm_create_from_id(device, rank, world, ident, C.byref(h)))
        return cls(h, rank, world)

    @classmethod
    def single(cls, device=0):
        h = _vp()
        _check(load().smilehip_comm_create(device, 0, 1, b"127.0.0.1", 0, C.byref(h)))
        return cls(h, 0, 1)

    def allgather_count(self, n, stream=None):
        counts = np.zeros(self.world, np.int64)
        _check(load().smilehip_comm_allgather_count(self._h, int(n), counts.ctypes.data, stream))
        return counts

    def gather_rows(self, d_rows, counts, cols, d_all, stream=None):
        c = np.ascontiguousarray(counts, np.int64)
        _check(load().smilehip_comm_gather_rows(self._h, d_rows, c.ctypes.data, cols, d_all, stream))

    def gather_rows_piece(self, d_rows, counts, cols, d_all, piece_rows, k, after_stream=None):
        c = np.ascontiguousarray(counts, np.int64)
        _check(load().smilehip_comm_gather_rows_piece(self._h, d_rows, c.ctypes.data, cols, d_all, piece_rows, k, after_stream))

    def wait(self, stream=None):
        _check(load().smilehip_comm_gather_wait(self._h, stream))

    def close(self):
        if self._h:
            load().smilehip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PieceGather:
    """The gather of every rank's (rows_r x cols) float32 matrix to rank 0 in pieces of `piece_rows` rows.

        g = PieceGather(local, transport)            # counts exchanged; rank 0 allocates the (sum rows x cols) result
        for k in range(g.pieces): g.piece(k, after_stream)
        out = g.finish(stream)                       # rank 0: the gathered matrix; other ranks: None

    transport: a `Comm` (device tensors, RCCL) or a torch.distributed module whose backend moves CPU tensors (gloo)."""

    def __init__(self, local, transport, piece_rows=1 << 16, group=None):
        import torch
        self.t, self.local, self.piece_rows, self.group = transport, local.contiguous(), int(piece_rows), group
        self.cols = int(local.shape[1])
        self.is_comm = isinstance(transport, Comm)
        if self.is_comm:
            self.rank, self.world = transport.rank, transport.world
            self.counts = transport.allgather_count(local.shape[0])
        else:
            dist = transport
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            n = torch.tensor([local.shape[0]], dtype=torch.int64)
            got = [torch.zeros_like(n) for _ in range(self.world)]
            dist.all_gather(got, n, group=group)
            self.counts = np.array([int(x.item()) for x in got], np.int64)
        self.pieces = n_pieces(self.counts, self.piece_rows)
        self.out = torch.empty((int(self.counts.sum()), self.cols), dtype=local.dtype, device=local.device) if self.rank == 0 else None

    def rebind(self, local):
        """the next matrix of the same shape (a double-buffered producer): same counts, same result buffer"""
        assert tuple(local.shape) == tuple(self.local.shape) and local.is_contiguous()
        self.local = local

    def piece(self, k, after_stream=None):
        if self.is_comm:
            self.t.gather_rows_piece(self.local.data_ptr() if self.local.numel() else None, self.counts, self.cols,
                                     self.out.data_ptr() if self.out is not None else None, self.piece_rows, k, after_stream)
            return
        dist, ops = self.t, []
        if self.rank == 0:
            first, n, dst = piece_rows_of(self.counts, self.piece_rows, k, 0)
            if n:
                self.out[dst:dst + n] = self.local[first:first + n]
            for p in range(1, self.world):
                first, n, dst = piece_rows_of(self.counts, self.piece_rows, k, p)
                if n:
                    ops.append(dist.P2POp(dist.irecv, self.out[dst:dst + n], p, self.group))
        else:
            first, n, _dst = piece_rows_of(self.counts, self.piece_rows, k, self.rank)
            if n:
                ops.append(dist.P2POp(dist.isend, self.local[first:first + n], 0, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def finish(self, stream=None):
        if self.is_comm:
            self.t.wait(stream)
        return self.out
