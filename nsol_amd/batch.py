"""Batch of independent volumes sharded over the GPUs of one node.

The reference has no notion of a batch or of several devices; one volume is one
independent solve, so the shard rule is: volume i -> rank i % world_size, no
per-iteration communication, and ONE gather of the results at the end (a rank's
results as one buffer)
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for
tests).
"""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """Indices owned by `rank` (round-robin)."""
    return list(range(rank, n_items, world_size))


def _settle():
    """Results built from ops.pd_run directly (not through Solver.run(), which
    settles itself): the verdict of the persistent runs before anything is sent."""
    import sys
    ops = sys.modules.get("nsol_amd.ops")
    if ops is not None:
        ops.settle_persist_runs()


def solve_batch(solve_one, n_items, group=None, dst=0):
    """Run solve_one(i) -> 1-D tensor for the items this rank owns, then gather
    all results on rank `dst` in item order.

    Returns the list of n_items result tensors on rank dst and None elsewhere.
    Every result must have the same length and dtype (one reconstruction per
    volume)."""
    if not dist.is_initialized():
        out = [solve_one(i) for i in range(n_items)]
        _settle()
        return out
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    mine = shard_indices(n_items, rank, world)
    local = [solve_one(i) for i in mine]
    _settle()
    if n_items == 0:
        return [] if rank == dst else None
    # Length, dtype and device of a result are agreed before the first gather,
    # so a rank that owns no volume (n_items < world_size) can take part with a
    # dummy instead of leaving the others waiting inside the collective.
    meta = (int(local[0].numel()), str(local[0].dtype).split(".")[-1],
            bool(local[0].is_cuda)) if local else None
    metas = [None] * world
    dist.all_gather_object(metas, meta, group=group)
    known = [m for m in metas if m is not None]
    if any(m != known[0] for m in known):
        raise ValueError("solve_batch: results differ in length / dtype / "
                         "device across ranks: %r" % (metas,))
    numel, dtype, on_gpu = known[0][0], getattr(torch, known[0][1]), known[0][2]
    if any(t.numel() != numel or t.dtype != dtype for t in local):
        raise ValueError("solve_batch: every result must have the same length "
                         "and dtype")
    stage_host = on_gpu and dist.get_backend(group) == "gloo"
    device = torch.device("cuda", torch.cuda.current_device()) \
        if on_gpu and not stage_host else torch.device("cpu")
    # ONE gather: a rank's results travel as one buffer of `rounds` slots (a rank that
    # owns fewer volumes -- n_items not a multiple of world_size -- leaves its last
    # slot unused), and the root hands out views of what arrived, in item order.
    rounds = (n_items + world - 1) // world
    if len(local) == rounds == 1:
        send = local[0].contiguous().view(-1)
        if stage_host:
            send = send.cpu()      # rehearsal without RCCL: stage through host
    else:
        alloc = torch.zeros if len(local) < rounds else torch.empty
        send = alloc(rounds * numel, dtype=dtype, device=device)
        for r, t in enumerate(local):
            send[r * numel:(r + 1) * numel].copy_(t.contiguous().view(-1))
    if rank != dst:
        dist.gather(send, gather_list=None, dst=dst, group=group)
        return None
    bucket = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, gather_list=bucket, dst=dst, group=group)
    return [bucket[i % world][(i // world) * numel:(i // world + 1) * numel]
            for i in range(n_items)]
