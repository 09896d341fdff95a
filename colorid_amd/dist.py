"""Multi-GPU plumbing for the read-sharded, index-replicated placement: one process per GPU,
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The only exchange step on the path is the reduction of per-accession counters: 3*C u64 sums for the proportional
search, and a bitwise AND of W words for the perfect search (RCCL has no bitwise reduction, so the partial AND
vectors are all-gathered and combined locally).  read_id needs no collective: per-read rows are independent.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_units: int, rank: int, world: int):
    """Contiguous, balanced partition of n_units over world ranks: sizes differ by at most one."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _active():
    return dist.is_available() and dist.is_initialized()


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    """In-place SUM over ranks.  RCCL ("nccl") reduces device tensors in place over xGMI; under "gloo" (the CPU tests, and the
    two-ranks-on-one-GPU test) a device tensor takes the round trip through host memory."""
    if not _active():
        return t
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_counts(counts: torch.Tensor) -> torch.Tensor:
    """In-place SUM over ranks of an int64 tensor of per-colour counters (hits | n_unique | sum_unique_freq)."""
    return all_reduce_sum(counts)


def allgather_and(words: torch.Tensor, any_missing: bool):
    """Perfect search across ranks: AND of the per-rank AND vectors, OR of the 'a row was absent' flags."""
    if not _active():
        return words, any_missing
    buf = torch.cat([words.to(torch.int64).reshape(-1), torch.tensor([1 if any_missing else 0], dtype=torch.int64, device=words.device)])
    if buf.is_cuda and dist.get_backend() == "gloo":
        buf = buf.cpu()
    gathered = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, buf)
    gathered = [x.to(words.device) for x in gathered]
    out = gathered[0][:-1].clone()
    missing = bool(gathered[0][-1].item())
    for g in gathered[1:]:
        out &= g[:-1]
        missing = missing or bool(g[-1].item())
    if missing:
        out.zero_()
    return out.to(words.dtype).reshape(words.shape), missing
