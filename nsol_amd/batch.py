"""Batch of independent volumes sharded over the GPUs of one node.

The reference has no notion of a batch or of several devices; one volume is one
independent solve, so the shard rule is: volume i -> rank i % world_size, no
per-iteration communication, and ONE gather of the results at the end
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for
tests).
"""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """Indices owned by `rank` (round-robin)."""
    return list(range(rank, n_items, world_size))


def solve_batch(solve_one, n_items, group=None, dst=0):
    """Run solve_one(i) -> 1-D tensor for the items this rank owns, then gather
    all results on rank `dst` in item order.

    Returns the list of n_items result tensors on rank dst and None elsewhere.
    Every result must have the same length and dtype (one reconstruction per
    volume)."""
    if not dist.is_initialized():
        return [solve_one(i) for i in range(n_items)]
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    mine = shard_indices(n_items, rank, world)
    local = [solve_one(i) for i in mine]
    rounds = (n_items + world - 1) // world
    out = [None] * n_items
    for r in range(rounds):
        have = r < len(local)
        # ranks without an item in the last (ragged) round send a dummy
        ref = local[0] if local else None
        if ref is None:
            raise RuntimeError("rank %d owns no volume: use world_size <= "
                               "n_items" % rank)
        send = local[r] if have else torch.zeros_like(ref)
        if send.is_cuda and dist.get_backend(group) == "gloo":
            send = send.cpu()      # rehearsal without RCCL: stage through host
        if rank == dst:
            bucket = [torch.empty_like(send) for _ in range(world)]
            dist.gather(send, gather_list=bucket, dst=dst, group=group)
            for src in range(world):
                i = r * world + src
                if i < n_items:
                    out[i] = bucket[src]
        else:
            dist.gather(send, gather_list=None, dst=dst, group=group)
    return out if rank == dst else None
