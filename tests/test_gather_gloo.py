"""N > 1 path on CPU: world_size-2 gloo. The path's only exchange is the
variable-length gather of per-rank feature matrices to rank 0
(opensmile_amd/gather.py); sharding is a static utterance partition."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opensmile_amd import gather
    torch.manual_seed(100 + rank)
    local = torch.full((rows[rank], 39), float(rank)) + torch.arange(rows[rank] * 39, dtype=torch.float32).reshape(rows[rank], 39)
    out = gather.gather_features(local, dst=0)
    if rank == 0:
        q.put([o.numpy().copy() for o in out])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rows", [(5, 7), (998, 0), (0, 3)])
def test_gather_features_world2(rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(out) == 2
    for r in range(2):
        exp = np.full((rows[r], 39), float(r), np.float32) + np.arange(rows[r] * 39, dtype=np.float32).reshape(rows[r], 39)
        assert out[r].shape == (rows[r], 39)
        assert np.array_equal(out[r], exp)


def test_shard_utterances_partition():
    from opensmile_amd.gather import shard_utterances
    # equal lengths -> contiguous blocks
    parts = shard_utterances([998] * 1000, 8)
    assert [len(p) for p in parts] == [125] * 8
    assert parts[0] == list(range(125)) and parts[7][-1] == 999
    # ragged -> every utterance exactly once, loads balanced within the longest item
    rng = np.random.default_rng(0)
    fc = rng.integers(1, 3000, size=517).tolist()
    parts = shard_utterances(fc, 8)
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(517))
    loads = [sum(fc[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(fc)
    # more ranks than utterances
    parts = shard_utterances([10, 20], 4)
    assert sorted(i for p in parts for i in p) == [0, 1]


def _piece_worker(rank, world, port, rows, piece_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opensmile_amd import comm
    local = torch.full((rows[rank], 130), float(rank)) + torch.arange(rows[rank] * 130, dtype=torch.float32).reshape(rows[rank], 130)
    g = comm.PieceGather(local, dist, piece_rows=piece_rows)          # counts exchanged, rank 0 allocates
    for k in range(g.pieces):                                         # every rank: piece 0, 1, ... in the same order
        g.piece(k)
    out = g.finish()
    if rank == 0:
        q.put((g.pieces, g.counts.tolist(), out.numpy().copy()))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rows,piece_rows", [((1000, 2500), 700), ((0, 33), 16), ((998, 0), 4096), ((64, 64), 64)])
def test_piece_gather_world2(rows, piece_rows):
    """The call order of the gather in pieces (opensmile_amd/comm.py: PieceGather -- what bench.py's N > 1 config-4 gather runs over
    libsmilehip_comm.so) with two gloo ranks: the rows libsmilehip_comm's own piece arithmetic (smilehip_comm_piece_rows) names,
    moved by torch.distributed point-to-point instead of RCCL; ragged blocks, an empty rank, blocks that end inside a piece."""
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opensmile_amd", "libsmilehip_comm.so")
    if not os.path.exists(lib):
        pytest.skip("libsmilehip_comm.so not built")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_piece_worker, args=(r, 2, port, rows, piece_rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    pieces, counts, out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert counts == list(rows) and pieces == -(-max(rows) // piece_rows)
    exp = np.concatenate([np.full((rows[r], 130), float(r), np.float32) + np.arange(rows[r] * 130, dtype=np.float32).reshape(rows[r], 130)
                          for r in range(2)])
    assert out.shape == exp.shape and np.array_equal(out, exp)


def test_piece_rows_cover_every_row_once():
    """smilehip_comm_piece_rows (host arithmetic of the library): over all pieces and ranks every row of the gathered matrix is
    written exactly once"""
    from opensmile_amd import comm
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opensmile_amd", "libsmilehip_comm.so")
    if not os.path.exists(lib):
        pytest.skip("libsmilehip_comm.so not built")
    rng = np.random.default_rng(3)
    for world in (1, 2, 8):
        counts = rng.integers(0, 5000, size=world)
        counts[rng.integers(0, world)] = 0
        for piece in (1, 37, 4096, 100000):
            hit = np.zeros(int(counts.sum()), np.int32)
            for k in range(comm.n_pieces(counts, piece)):
                for r in range(world):
                    first, n, dst = comm.piece_rows_of(counts, piece, k, r)
                    assert n == 0 or (0 <= first and first + n <= counts[r])
                    hit[dst:dst + n] += 1
            assert (hit == 1).all()
