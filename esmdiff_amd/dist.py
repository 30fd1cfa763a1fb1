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


def pin_to_gpu_numa(local_rank):
    """Pin this process's host threads to the CPUs of the NUMA node its GPU hangs off (one process per GPU: the sampling loop's
    launch thread, the weight upload and the PDB writer should not cross sockets).  Returns {"node", "cpus"} or None when
    the topology files are missing, the node is unknown (-1) or there is no GPU (stub engines).  Never raises."""
    import glob
    import os
    try:
        if local_rank is None or not torch.cuda.is_available():
            return None
        p = torch.cuda.get_device_properties(local_rank)
        want = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
        for d in glob.glob("/sys/bus/pci/devices/*"):
            if os.path.basename(d).startswith(want):
                node = int(open(os.path.join(d, "numa_node")).read().strip())
                if node < 0:
                    return None
                cpus = set()
                for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                    a, _, b = part.partition("-")
                    cpus.update(range(int(a), int(b or a) + 1))
                cpus &= os.sched_getaffinity(0)
                if not cpus:
                    return None
                os.sched_setaffinity(0, cpus)
                return {"node": node, "cpus": len(cpus)}
    except Exception:
        return None
    return None
