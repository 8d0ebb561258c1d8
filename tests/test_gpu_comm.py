"""The RCCL gather step (include/smilehip_comm.h) executed IN THIS PROCESS on the one device a gpurun box has:
a world-size-1 communicator (ncclGetUniqueId, ncclCommInitRank), the count all-gather, smilehip_comm_gather_rows (rank 0's own
block), and the send / receive primitives as a grouped self pair. No xGMI link is crossed: a multi-GPU run has never happened
(DESIGN 7)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def comm():
    L = C.CDLL(os.path.join(ROOT, "opensmile_amd", "libsmilehip_comm.so"))
    L.smilehip_comm_last_error.restype = C.c_char_p
    L.smilehip_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    L.smilehip_comm_destroy.argtypes = [C.c_void_p]
    L.smilehip_comm_allgather_count.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.smilehip_comm_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.smilehip_comm_self_sendrecv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.smilehip_comm_rccl_version.argtypes = [C.POINTER(C.c_int)]
    h = C.c_void_p()
    assert L.smilehip_comm_create(0, 0, 1, b"127.0.0.1", 0, C.byref(h)) == 0, L.smilehip_comm_last_error()
    yield L, h
    assert L.smilehip_comm_destroy(h) == 0


def test_rccl_version(comm):
    L, h = comm
    v = C.c_int(0)
    assert L.smilehip_comm_rccl_version(C.byref(v)) == 0 and v.value > 20000


def test_world1_count_and_gather_rows(comm):
    """config 5's gather: 88 functionals per utterance -- rank 0's block arrives in d_all"""
    import torch
    L, h = comm
    rows = torch.arange(1250 * 88, dtype=torch.float32, device="cuda").reshape(1250, 88)
    counts = np.zeros(1, np.int64)
    assert L.smilehip_comm_allgather_count(h, 1250, counts.ctypes.data, None) == 0, L.smilehip_comm_last_error()
    assert counts.tolist() == [1250]
    allr = torch.zeros_like(rows)
    assert L.smilehip_comm_gather_rows(h, rows.data_ptr(), counts.ctypes.data, 88, allr.data_ptr(), None) == 0, L.smilehip_comm_last_error()
    torch.cuda.synchronize()
    assert torch.equal(allr, rows)
    # an empty shard (a rank whose utterances were all too short)
    counts[0] = 0
    assert L.smilehip_comm_gather_rows(h, None, counts.ctypes.data, 88, allr.data_ptr(), None) == 0


def test_self_send_recv_pair(comm):
    """ncclSend + ncclRecv inside one group, peer = own rank: 44 MB (config 5's block per rank)"""
    import torch
    L, h = comm
    n = 125000 * 88
    src = torch.rand(n, dtype=torch.float32, device="cuda")
    dst = torch.zeros(n, dtype=torch.float32, device="cuda")
    assert L.smilehip_comm_self_sendrecv(h, src.data_ptr(), dst.data_ptr(), n, None) == 0, L.smilehip_comm_last_error()
    torch.cuda.synchronize()
    assert torch.equal(src, dst)


def test_world1_gather_in_pieces():
    """smilehip_comm_gather_rows_piece / _wait on the one device: rank 0's block travels piece by piece on the communicator's own
    stream behind the caller's stream (the event hand-over), the gathered matrix equals the block; rendezvous by unique id"""
    import torch
    from opensmile_amd import comm
    c = comm.Comm.from_process_group(None, 0)                 # smilehip_comm_unique_id + smilehip_comm_create_from_id, world 1
    stream = torch.cuda.current_stream().cuda_stream
    rows = torch.zeros((12500, 130), dtype=torch.float32, device="cuda")
    g = comm.PieceGather(rows, c, piece_rows=3000)
    assert g.pieces == 5 and g.counts.tolist() == [12500]
    rows.copy_(torch.arange(12500 * 130, dtype=torch.float32, device="cuda").reshape(12500, 130))    # enqueued on the caller's stream
    for k in range(g.pieces):
        g.piece(k, after_stream=stream)
    out = g.finish(stream)                                   # the caller's stream waits for the pieces
    torch.cuda.synchronize()
    assert torch.equal(out, rows)
    # an empty block, and finish on the host
    g2 = comm.PieceGather(rows[:0], c, piece_rows=3000)
    assert g2.pieces == 0 and g2.finish(None).shape == (0, 130)
    c.close()
