"""The path's only exchange step: variable-length gather of per-rank feature
matrices to one rank. The reference has nothing distributed;
its only "data parallelism" is one SMILExtract process per file
(scripts/extract_features_batch.pl), whose outputs land in one directory --
this gather is the multi-GPU equivalent of that.

RCCL has no gatherv: counts are all-gathered, then every non-root rank sends
its block straight to the root and the root posts one receive per peer, all in
ONE batched P2P group (ncclGroupStart/End under torch.distributed). xGMI is
point-to-point, so each peer's block travels over its own link to the root --
no ring, no per-link serialisation of other ranks' data.
"""
import torch
import torch.distributed as dist


def gather_features(local, dst=0, group=None):
    """local: (rows_r x cols) tensor on this rank's device. Returns, on `dst`,
    the list of every rank's block in rank order (dst's own block is `local`
    itself, not a copy); None on the other ranks."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [local]
    local = local.contiguous()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    ops, outs = [], None
    if rank == dst:
        outs = []
        for r in range(world):
            if r == dst:
                outs.append(local)
                continue
            buf = torch.empty((counts[r],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            outs.append(buf)
            if counts[r] > 0:
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
    elif local.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, local, dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return outs


def shard_utterances(frame_counts, world):
    """Static partition of utterances over ranks: greedy
    longest-processing-time on frame counts; equal-length corpora degrade to
    contiguous blocks. Returns a list of index lists, one per rank; every
    utterance appears exactly once; deterministic."""
    order = sorted(range(len(frame_counts)), key=lambda i: (-int(frame_counts[i]), i))
    loads = [0] * world
    parts = [[] for _ in range(world)]
    if len(set(int(c) for c in frame_counts)) <= 1:
        per = -(-len(frame_counts) // world) if frame_counts else 0
        for r in range(world):
            parts[r] = list(range(r * per, min(len(frame_counts), (r + 1) * per)))
        return parts
    for i in order:
        r = min(range(world), key=lambda q: (loads[q], q))
        parts[r].append(i)
        loads[r] += int(frame_counts[i])
    for p in parts:
        p.sort()
    return parts
