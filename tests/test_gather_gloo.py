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
