"""Multi-GPU: the `num_samples` replicas are independent units (no cross-sample op anywhere in
/root/reference/slm/models/model.py:543-607; the reference already splits them into independent batches,
sample_esmdiff.py:181-216), so they shard over ranks with no data-path collective.  The only exchange is one
all-gather of the final ids (int16: ids <= 4100) at the very end — RCCL over xGMI with backend "nccl".
Noise is indexed by the GLOBAL sample number (esmdiff_rng.sample_offset), so results do not depend on the
number of ranks."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
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


def broadcast_state_dict(loader, device, src: int = 0):
    """Checkpoint fan-out: ONE rank reads the file, every rank gets the tensors.

    The reference loads its 5.5 GB float32 checkpoint with torch.load in the single process it runs
    (/root/reference/slm/utils/checkpoint_utils.py:59-73).  With one process per GPU, eight ranks doing that at once read and
    unpickle the same file eight times and hold 44 GB of host copies; here rank `src` calls `loader()` (-> dict name -> CPU
    tensor), the others receive (name, shape, dtype) by one object broadcast and the values by ONE broadcast of a flat byte
    buffer — over RCCL / xGMI from rank src's GPU when the backend is "nccl" (5.5 GB ~ 0.1 s on a ring), over gloo through host
    memory otherwise (CPU tests; two ranks sharing one GPU).  Without a process group (or a world of 1) it is loader() + upload.

    With a process group this is a COLLECTIVE: every rank must call it (load_state_dict_from_lightning_ckpt does).  A failure of
    `loader()` on the reading rank (missing file, no 'module' entry, no net.* keys ...) is broadcast as a header before anything
    else, so that EVERY rank raises the reader's error instead of waiting in a collective until the backend's timeout.

    Returns (state dict of views into one flat buffer on `device`, timings {"read_s", "upload_s", "broadcast_s", "load_s",
    "bytes"}); the Engine copies what it needs at create time, after which the flat buffer can be dropped — callers should
    `del` the returned dict once their engines exist (the float32 checkpoint is 5.5 GB per rank while it lives)."""
    import time
    device = torch.device(device)
    t0 = time.perf_counter()
    multi = dist.is_available() and dist.is_initialized()      # (a world of 1 still goes through the collectives: same code path)
    rank = dist.get_rank() if multi else src
    sd, err = None, None
    if rank == src:
        try:
            sd = loader()
        except Exception as ex:      # reported to every rank below; without a process group it is simply re-raised
            if not multi:
                raise
            err = f"{type(ex).__name__}: {ex}"
    t_read = time.perf_counter()
    meta = [{"error": err, "meta": None if sd is None else [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]}] \
        if rank == src else [None]
    if multi:
        dist.broadcast_object_list(meta, src=src)
    if meta[0]["error"] is not None:
        raise RuntimeError(f"checkpoint loading failed on rank {src} (reported to all {dist.get_world_size() if multi else 1} ranks): "
                           f"{meta[0]['error']}")
    meta = meta[0]["meta"]
    sizes = [int(torch.empty(0, dtype=getattr(torch, dt)).element_size()) * int(np.prod(shape, dtype=np.int64)) for _, shape, dt in meta]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 255) // 256 * 256                 # every tensor starts 256-byte aligned
    via_host = multi and dist.get_backend() != "nccl" and device.type == "cuda"
    flat = torch.empty(tot, dtype=torch.uint8, device="cpu" if via_host else device)
    if rank == src:
        for (k, _, _), o, n in zip(meta, offs, sizes):
            flat[o:o + n].copy_(sd[k].detach().contiguous().reshape(-1).view(torch.uint8), non_blocking=False)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t_up = time.perf_counter()
    if multi:
        dist.broadcast(flat, src=src)
        if device.type == "cuda" and not via_host:
            torch.cuda.synchronize(device)
    t_bc = time.perf_counter()
    if via_host:
        flat = flat.to(device)
        torch.cuda.synchronize(device)
    out = {k: flat[o:o + n].view(getattr(torch, dt)).reshape(shape) for (k, shape, dt), o, n in zip(meta, offs, sizes)}
    t_end = time.perf_counter()
    return out, {"read_s": round(t_read - t0, 3), "upload_s": round(t_up - t_read, 3), "broadcast_s": round(t_bc - t_up, 3),
                 "load_s": round(t_end - t0, 3), "bytes": int(tot), "reader_rank": src, "rank": rank,
                 "world": dist.get_world_size() if multi else 1,
                 "path": "no process group" if not multi else ("gloo via host memory" if via_host or device.type == "cpu" else "RCCL from the reader's GPU")}


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
