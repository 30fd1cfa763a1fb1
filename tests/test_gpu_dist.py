"""GPU tests of the multi-process path (run with -m gpu on an MI355X box).

SURVEY.md section 8(e): one process per GPU, samples sharded by global index, ONE all_gather of the int16 ids over RCCL.
The reference has no counterpart (it is single-device, /root/reference/slm/sample_esmdiff.py:34); the split these tests
must reproduce is the reference's own batch split (:181-216): any sharding gives the ids a single process gives.

  * the launcher branch of bench.py with the REAL engine and the "nccl" (= RCCL) backend — on a one-GPU box as a world of 1
    (`--spawn`), which still goes through torch.distributed.run, init_process_group, all_gather and the barrier;
  * Engine + RCCL at world size 2 against a single process — runs whenever the box shows >= 2 GPUs, skipped otherwise.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(extra, timeout=900):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_launcher_path_rccl_world1():
    """bench.py re-executing itself under torch.distributed.run with the real engine: process group on "nccl", the int16
    all_gather and the device barrier all run (world of 1 on a one-GPU box), ONE JSON line comes back, and the same
    arguments without the launcher give the same kind of line."""
    common = ["--tiny", "--steps", "2", "--warmup", "1", "--samples-per-gpu", "8", "--residues", "30", "--no-cpu-baseline"]
    a = _bench(["--gpus", "1", "--spawn"] + common)
    b = _bench(["--gpus", "1"] + common)
    for out in (a, b):
        assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0 and out["data"] == "debug-tiny-model"
        assert out["roofline"]["launches"] > 0 and out["roofline"]["union_busy_ms"] > 0
    assert "RCCL all_gather" in a["config"]["parallelism"] and "no process group" in b["config"]["parallelism"]
    assert a["config"]["samples_per_gpu"] == b["config"]["samples_per_gpu"] == 8


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from esmdiff_amd.config import TINY
    from esmdiff_amd.dist import gather_ids, shard_samples
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    from esmdiff_amd.weights import random_init_state_dict
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    N, L = 7, 40
    sd = random_init_state_dict(TINY, seed=3)
    eng = Engine(TINY, sd, max_batch=N, max_len=L, device=rank)
    g = torch.Generator().manual_seed(1)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])
    off, cnt = shard_samples(N, world, rank)
    ids = eng.ddpm_sample(seq[None].repeat(cnt, 1).cuda(rank), ddpm_schedule(5, freq_dim=TINY.freq_dim), seed=9, sample_offset=off)
    allids = gather_ids(ids, N)
    if rank == 0:
        np.save(Path(tmp) / "gathered.npy", allids.cpu().numpy())
    eng.close()
    dist.barrier(device_ids=[rank])
    dist.destroy_process_group()


def test_engine_rccl_world2_equals_single_process(tmp_path):
    """Two ranks, two GPUs, the real Engine and RCCL: shard 7 samples 4 + 3, sample, gather — the ensemble equals the one a
    single process draws (Philox keyed by the global sample index; row-independent kernels).  Needs 2 visible GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs, this box shows {torch.cuda.device_count()} (the driver's multi-GPU tier runs it)")
    import torch.multiprocessing as mp
    from esmdiff_amd.config import TINY
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    from esmdiff_amd.weights import random_init_state_dict
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    N, L = 7, 40
    eng = Engine(TINY, random_init_state_dict(TINY, seed=3), max_batch=N, max_len=L)
    g = torch.Generator().manual_seed(1)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])
    want = eng.ddpm_sample(seq[None].repeat(N, 1).cuda(), ddpm_schedule(5, freq_dim=TINY.freq_dim), seed=9).cpu().numpy()
    eng.close()
    assert np.array_equal(got, want)


def _worker_shared_gpu(rank, world, port, tmp):
    """Two ranks, ONE GPU (both on device 0), the real Engine, gloo for the exchange (RCCL refuses two ranks on one device)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from esmdiff_amd.config import TINY
    from esmdiff_amd.dist import gather_ids, shard_samples
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    from esmdiff_amd.weights import random_init_state_dict
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, L = 7, 40
    g = torch.Generator().manual_seed(1)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])
    off, cnt = shard_samples(N, world, rank)
    out = {}
    for prec in ("bf16", "f16", "f32_split"):
        eng = Engine(TINY, random_init_state_dict(TINY, seed=3), max_batch=N, max_len=L, device=0, precision=prec)
        eng.set_step0_sharing(True)
        eng.set_final_skip(True)
        ids = eng.ddpm_sample(seq[None].repeat(cnt, 1).cuda(0), ddpm_schedule(5, freq_dim=TINY.freq_dim), seed=9, sample_offset=off)
        out[prec] = gather_ids(ids.cpu(), N).numpy()          # the exchange itself over gloo (CPU tensors)
        eng.close()
    if rank == 0:
        np.savez(Path(tmp) / "gathered_shared.npz", **out)
    dist.barrier()
    dist.destroy_process_group()


def test_engine_two_ranks_one_gpu_equal_single_process(tmp_path):
    """The multi-rank path with the REAL engine on a one-GPU box: two processes share device 0, each samples its shard (4 + 3 of 7
    samples, Philox keyed by the global sample index, exact shortcuts on), gloo gathers the int16 ids — the ensemble must equal the
    one a single process draws, for the bf16, f16 and f32_split engines.  (What this cannot cover is RCCL itself at world > 1:
    test_engine_rccl_world2_equals_single_process does, on a node with two GPUs.)"""
    import torch.multiprocessing as mp
    from esmdiff_amd.config import TINY
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    from esmdiff_amd.weights import random_init_state_dict
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_shared_gpu, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered_shared.npz")
    N, L = 7, 40
    g = torch.Generator().manual_seed(1)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])
    for prec in ("bf16", "f16", "f32_split"):
        eng = Engine(TINY, random_init_state_dict(TINY, seed=3), max_batch=N, max_len=L, precision=prec)
        want = eng.ddpm_sample(seq[None].repeat(N, 1).cuda(), ddpm_schedule(5, freq_dim=TINY.freq_dim), seed=9).cpu().numpy()
        eng.close()
        assert np.array_equal(got[prec], want), prec
