"""Multi-GPU plumbing for the data-parallel path (one process per GPU, torch.distributed; backend "nccl" == RCCL).

The hot path shards naturally: every image is an independent forward (the reference is batch 1, dinov2.cpp:630), so
ranks never exchange activations.  The only collective is the ONE-TIME broadcast of rank 0's converted weight arena
over xGMI; timing uses a barrier + max-over-ranks.  Backend-agnostic so the same code runs
under gloo on CPU in tests/test_distributed_cpu.py.
"""
from __future__ import annotations


def shard_range(global_batch: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous split of a global batch: rank r owns images [lo, hi).  Remainder goes to the low ranks."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    q, r = divmod(global_batch, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class DevPtr:
    """Expose a raw device pointer (the model's weight arena) to torch, zero-copy."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _via_host(dist, t) -> bool:
    """gloo moves host memory: device tensors are staged through the host for it (bench.py's `--backend gloo` dry run of the
    N > 1 path on a 1-GPU box, and the CPU tests); under nccl (= RCCL) device tensors go out as they are."""
    return getattr(t, "is_cuda", False) and dist.get_backend() == "gloo"


def broadcast_weights(dist, arena, src: int = 0, chunk_bytes: int = 1 << 30):
    """Broadcast a flat uint8 tensor in <= 1 GiB messages (ring broadcast over xGMI is per-link bound; a few large
    messages, not many small ones)."""
    n = arena.numel()
    for off in range(0, n, chunk_bytes):
        chunk = arena[off:min(n, off + chunk_bytes)]
        if _via_host(dist, chunk):
            host = chunk.cpu()
            dist.broadcast(host, src=src)
            chunk.copy_(host)
        else:
            dist.broadcast(chunk, src=src)


def max_over_ranks(dist, torch, value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(dist, torch, local_rows, world: int):
    """Optional all-gather of per-rank output rows (e.g. logits [B/G, C]) into rank order."""
    src = local_rows.cpu() if _via_host(dist, local_rows) else local_rows
    outs = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(outs, src)
    return torch.cat(outs, dim=0).to(local_rows.device)
