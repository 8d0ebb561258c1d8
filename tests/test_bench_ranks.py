"""bench.py's N > 1 protocol. On CPU: `python bench.py --gpus 2 --plumbing-only` launches its two ranks itself
(torch.distributed.run, gloo) and runs every step of the protocol except the device work -- rendezvous, barrier, per-rank
times, frame sum, the gather-v to rank 0, one JSON line from rank 0. On the GPU box: the real line with two ranks
(sharing the one device over gloo when there is only one)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus2_launches_its_own_ranks_plumbing_only():
    r = _run(["--gpus", "2", "--plumbing-only", "--steps", "2"], 300)
    assert r["n_gpus"] == 2 and r["value"] is None and r["metric"].startswith("PLUMBING ONLY")
    assert r["config"]["ranks_seen"] == 2
    assert r["config"]["frames_total"] == 998 * 10 + 998 * 11
    assert r["config"]["gathered_rows"] == [998 * 10, 998 * 11]
    assert r["config"]["gathered_in_pieces"] == {"pieces": 3, "rows": 998 * 21, "equal": True}      # ceil(998 * 11 / 4096) pieces
    assert r["ranks"]["backend"] == "gloo" and len(r["ranks"]["rank_devices"]) == 2


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--plumbing-only"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_bench_gpus2_real_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "2"], 600)
    assert r["n_gpus"] == 2 and r["config"]["ranks_seen"] == 2
    assert r["config"]["frames_rank0"] == 998000 and r["value"] > 1e8
    assert r["gather_ms"] > 0 and 0 < r["value_incl_gather"] < r["value"]
    assert len(r["ranks"]["rank_devices"]) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("config", [2, 3, 4])
def test_bench_gather_through_the_comm_library_world1(config):
    """--with-gather at N = 1: the gather section of the N > 1 line with a world-size-1 RCCL communicator -- the same calls into
    libsmilehip_comm.so (config 4: the LLD level in pieces on the communicator's stream beside the next step's kernels)"""
    extra = ["--no-configs", "--no-h2d"] if config == 2 else ["--config", str(config)]
    r = _run(extra + ["--utts", "96", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--with-gather"], 600)
    assert r["n_gpus"] == 1 and r["gather_via"].startswith("libsmilehip_comm.so")
    if config == 2:
        assert r["gather_ms"] > 0
    else:
        assert r["gather_check"]["rank0_block_equal"] is True
    if config == 4:
        assert r["gather_check"]["pieces_per_step"] >= 1 and r["gather_check"]["value_incl_gather_overlapped"] > 0
