"""Multi-GPU: the `num_samples` replicas are independent units (no cross-sample op anywhere in
/root/reference/slm/models/model.py:543-607; the reference already splits them into independent batches,
sample_esmdiff.py:181-216), so they shard over ranks with no data-path collective.  The only exchange is one
all-gather of the final ids (int16: ids <= 4100) at the very end — RCCL over xGMI with backend "nccl".
Noise is indexed by the GLOBAL sample number (esmdiff_rng.sample_offset), so results do not depend on the
number of ranks."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_samples(num_samples: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of samples for `rank`: (global offset, count); the first `num_samples % world` ranks
    get one extra."""
    base, extra = divmod(num_samples, world_size)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def gather_rows(local: torch.Tensor, num_samples: int) -> torch.Tensor:
    """All ranks' (count_r, ...) blocks of any dtype -> (num_samples, ...) on every rank, in global sample order: ONE
    all_gather of equally sized byte buffers (the shards differ by at most one sample)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    cap = -(-num_samples // world)
    buf = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    raw = buf.reshape(cap, -1).view(torch.uint8)    # all_gather is type-agnostic; neither gloo nor RCCL has an int16 type
    out: List[torch.Tensor] = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(out, raw)
    parts = [out[r].view(local.dtype).reshape((cap,) + tuple(local.shape[1:]))[: shard_samples(num_samples, world, r)[1]]
             for r in range(world)]
    return torch.cat(parts, 0)


def gather_ids(local_ids: torch.Tensor, num_samples: int) -> torch.Tensor:
    """All ranks' (count_r, L) id blocks -> (num_samples, L) int64 on every rank, in global sample order (int16 on the wire:
    ids <= 4100)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_ids
    return gather_rows(local_ids.to(torch.int16).contiguous(), num_samples).to(torch.int64)
