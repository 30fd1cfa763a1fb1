#!/usr/bin/env python3
"""bench.py — conformation samples/sec of the ESMDiff ancestral sampler on MI355X.

One "step" = one pass of the hot path over one batch: `samples_per_gpu` independent conformations of a
256-residue protein (L_tok = 258), 25 reverse-diffusion updates + the noise-removal pass = 26 forwards of
the ESM3-open-sized (1.4 B parameter, random-init, bf16 MFMA) structure-token transformer plus 26 fused
sampler launches (BASELINE.json configs[1]; reference loop: slm/models/model.py:543-581 driven by
slm/sample_esmdiff.py:137-233).  Inputs (tokens, weights) are resident in HBM when the timed region starts.

    python bench.py [--gpus N --steps K --warmup W]        (N > 1 without a launcher: re-executes itself under
                                                            torch.distributed.run, one rank per GPU, still ONE JSON line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (FFN-up GEMM, bf16 MFMA): algorithmic FLOP per launch / mean launch duration
                measured with HIP events on the launch stream inside the timed region.
  cpu_baseline  the oracle (torch-CPU float32 restatement, oracle/) timed on this box's host cores on a
                bounded sample; kind = "port" (the reference's own Python cannot run here: esm==3.0.4 absent).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense)


def flops_forward_per_sample(L: int, cfg) -> float:
    """2 FLOP/MAC; linears + attention per layer + head."""
    D, FH, V = cfg.d_model, cfg.ffn_hidden, cfg.n_structure_heads
    lin = 2 * D * 3 * D + 2 * D * D + 2 * D * 2 * FH + 2 * FH * D
    attn = 4 * D * L     # QK^T + PV: 2 * 2 * L * 64 * heads
    head = 2 * D * D + 2 * D * V
    return L * (cfg.n_layers * (lin + attn) + head)


def usable_cores() -> int:
    """Host cores this process may really use: min(affinity, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_model_name() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, sd, L, T, engines=None, seconds_budget=90.0):
    """The CPU leg, by the protocol BASELINE.md section 3 wrote down, on this box's host cores with the oracle (torch-CPU f32
    restatement of the network + the C-oracle sampler; kind "port": the reference's own Python needs esm==3.0.4):
      * configs[0] IN FULL: BPTI length (58 residues, L_tok = 60), 4 samples, 25 updates + the noise-removal pass = 26 forwards
        of the whole network; its wall time is the reference's "Sampling token time" window for that config;
      * configs[1] (the metric's config) EXTRAPOLATED, and labelled so: ONE whole update (forward + sampler) at B = 4,
        L_tok = L, times (T + 1) x (100 / 4) — a full run is ~1.9 PFLOP, about an hour on these cores.
    `value` is the extrapolated configs[1] rate (the unit of the metric).  The B = 4 forward's logits double as the checker of
    `parity_spot` (the engine's logits on the same tokens, compared once).  Bounded: the configs[0] chain stops early when
    `seconds_budget` is exceeded (then its wall time is itself extrapolated from the updates done, and says so)."""
    from esmdiff_amd.schedule import ddpm_schedule
    from oracle import c_oracle
    from oracle.esm3_ref import build_from_state_dict
    cores = usable_cores()
    torch.set_num_threads(cores)
    net, emb = build_from_state_dict(cfg, {k: v.cpu() for k, v in sd.items()})
    g = torch.Generator().manual_seed(0)

    def update(x, seq, sch, i, fin, seed):
        Bc, Lc = x.shape
        with torch.no_grad():
            cond = torch.tile(emb(sch.sigma_t[i] * torch.ones(Bc))[:, None, :], (1, Lc, 1))
            lg = net(structure_tokens=torch.from_numpy(x), sequence_tokens=seq, auxiliary_embeddings=cond).structure_logits
        x2 = c_oracle.ddpm_step(x, lg.numpy(), 0.0 if fin else sch.mc_t[i].item(), 0.0 if fin else sch.mc_s[i].item(), final=fin,
                                seed=seed, step=i)
        return x2, lg

    # configs[0] in full
    import numpy as np
    B0, L0, T0 = 4, 60, 25
    seq0 = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L0 - 2,), generator=g), torch.tensor([2])])[None].repeat(B0, 1)
    sch0 = ddpm_schedule(T0)
    x = np.full((B0, L0), 4096, dtype=np.int64)
    done, t0 = 0, time.perf_counter()
    for i in range(T0 + 1):
        x, _ = update(x, seq0, sch0, i, i == T0, 0)
        done += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    el0 = time.perf_counter() - t0
    full0 = done == T0 + 1
    wall0 = el0 if full0 else el0 / done * (T0 + 1)
    # one whole update at B = 4, L_tok = L
    Bc = 4
    seq1 = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])
    seq = seq1[None].repeat(Bc, 1)
    sch = ddpm_schedule(T)
    xb = np.full((Bc, L), 4096, dtype=np.int64)
    t1 = time.perf_counter()
    _, lg = update(xb, seq, sch, 0, False, 0)
    per_update = time.perf_counter() - t1
    spot = {}
    for name, eng in (engines or {}).items():
        if eng is None:
            continue
        got = eng.forward_logits(torch.from_numpy(xb).to(eng.device), seq.to(eng.device), sch.t_freq[0]).float().cpu()
        err = (got - lg).abs()
        spot[name] = {"what": f"engine (precision {eng.precision}, head {eng.head_precision}) vs oracle (f32 torch) logits of ONE forward of "
                              f"the full {cfg.n_layers}-block model, B={Bc}, L_tok={L}, all positions masked, sigma of step 0",
                      "max_abs_logit_err": round(float(err.max()), 6), "mean_abs_logit_err": round(float(err.mean()), 7),
                      "logit_std": round(float(lg.std()), 4),
                      "cosine": round(float(torch.nn.functional.cosine_similarity(got.flatten().double(), lg.flatten().double(), dim=0)), 7),
                      "argmax_agreement": round(float((got.argmax(-1) == lg.argmax(-1)).float().mean()), 4)}
    n_ref = 100
    return {"value": Bc / (per_update * (T + 1)), "unit": "samples/s", "cores": cores, "kind": "port",
            "protocol": "BASELINE.md section 3",
            "sample": (f"configs[1] EXTRAPOLATED from ONE whole reverse-diffusion update (f32 torch forward of all {cfg.n_layers} blocks + C "
                       f"sampler) at B={Bc}, L_tok={L}: {per_update:.2f} s; a {n_ref}-sample run = {T + 1} x {n_ref // Bc} x that = "
                       f"{per_update * (T + 1) * n_ref / Bc:.0f} s"),
            "configs0_full_run": {"what": f"BASELINE configs[0] {'in full' if full0 else 'EXTRAPOLATED from %d of %d updates' % (done, T0 + 1)}: "
                                          f"B={B0}, L_tok={L0}, {T0} updates + noise removal = {T0 + 1} forwards of all {cfg.n_layers} blocks "
                                          "+ C sampler ('Sampling token time' window, sample_esmdiff.py:177-223)",
                                  "wall_s": round(wall0, 2), "samples_per_s": round(B0 / wall0, 4), "updates_run": done},
            "cpu_model": cpu_model_name(), "cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(),
            "torch": torch.__version__}, spot


def workload_name(args, world):
    """Which BASELINE.json configuration the arguments actually are (never a fixed label)."""
    R, B, T = args.residues, args.samples_per_gpu, args.num_steps
    base = f"{R}-residue synthetic sequence, num_steps={T}, num_samples={B}/GPU, ESM3-open-sized random-init weights, bf16 MFMA"
    if args.tiny:
        return "debug: tiny model, " + base
    if args.mode == "ddpm" and not args.inpaint and (R, B, T) == (256, 100, 25):
        if world == 1:
            return "BASELINE configs[1]: single MI355X, " + base
        return (f"BASELINE configs[2]{'' if world == 8 else ' scaling series'}: {world}xMI355X, {B * world} samples sharded "
                f"{B}/GPU, one RCCL all_gather of the ids at the end, ") + base
    if args.mode == "ddpm" and not args.inpaint and world == 1 and (R, B, T) == (1024, 32, 25):
        return "BASELINE configs[3]: single MI355X, 1024-residue long chain, " + base
    if args.mode == "ddpm" and args.inpaint and world == 1 and (R, B, T) == (256, 100, 50):
        return f"BASELINE configs[4]: single MI355X, inpainting prior with residues {args.inpaint} masked (partial-mask resample), " + base
    if args.mode == "ddpm" and not args.inpaint and world == 1 and (R, B, T) == (58, 4, 25):
        return "BASELINE configs[0] shape on the GPU (BPTI length, 4 samples), " + base
    return (f"not a BASELINE configuration: mode={args.mode}" + (f", inpaint {args.inpaint}" if args.inpaint else "")
            + f", {world} GPU(s), " + base)


class PowerSampler:
    """Package power and shader clock of THIS rank's GPU from the amdgpu hwmon files, sampled every 50 ms by a thread
    while the timed region runs (a report beside the number: this workload sits at the package power cap).
    Silent when the files are not there."""

    def __init__(self, device_index: int):
        import glob
        import threading
        self.dir = None
        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        try:
            p = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(d)).startswith(want):
                    hw = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
                    if hw and os.path.exists(os.path.join(hw[0], "power1_input")):
                        self.dir = hw[0]
        except Exception:
            self.dir = None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as fh:
            return float(fh.read().strip())

    def _loop(self):
        while not self._stop.wait(0.05):
            try:
                self.samples.append((self._read("power1_input") * 1e-6, self._read("freq1_input") * 1e-6))
            except Exception:
                return

    def start(self):
        if self.dir is not None:
            import threading
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join()
        if not self.samples:
            return None
        w = [a for a, _ in self.samples]
        f = [b for _, b in self.samples]
        try:
            cap = self._read("power1_cap") * 1e-6
        except Exception:
            cap = None
        return {"cap_w": cap, "mean_w": round(sum(w) / len(w), 1), "max_w": round(max(w), 1),
                "mean_sclk_mhz": round(sum(f) / len(f), 1), "samples": len(w),
                "source": "amdgpu hwmon power1_input / freq1_input every 50 ms over the timed region"}


def roofline_report(args, cfg, B, L, n_fwd_sample, prof_dom, prof):
    """The dominant kernel (FFN-up GEMM [M,1536] x [8192,1536]^T, SwiGLU epilogue) against the bf16 MFMA peak.

    The engine runs a large batch as sub-batches on separate HIP streams, so an FFN-up launch covers M / streams rows and
    shares the GPU with the other stream's kernels.  Three figures, all from HIP events on the launch streams:
      roofline.achieved / frac   TIMED REGION, the contract's figure: algorithmic FLOP of one launch / mean launch duration
                                 (what rocprofv3 --kernel-trace --stats reports as the kernel's average; the launch shares
                                 the GPU with the other stream's kernels).  r04: this is the top-level figure again, so that it
                                 follows from a committed rocprof artifact and does not move with bookkeeping
      roofline.union             labelled extra, device level: FFN-up FLOP of ALL streams / the UNION of the launches' busy
                                 intervals on the device timeline (no instant counted twice)
      roofline.exclusive         labelled extra: the same kernel ALONE on the GPU at the full M (single-stream pass, untimed)"""
    M = B * L
    timed = prof_dom["gemm_ffn_up"]["launches"] > 0
    up = prof_dom["gemm_ffn_up"] if timed else prof["gemm_ffn_up"]
    expect = (args.steps if timed else 1) * n_fwd_sample * cfg.n_layers
    parts = max(1, round(up["launches"] / expect))
    flop_up_full = 2.0 * M * cfg.d_model * 2 * cfg.ffn_hidden
    flop_up = flop_up_full / parts
    ms_up = up["ms"] / max(up["launches"], 1)
    ach_launch = flop_up / (ms_up * 1e-3) / 1e12 if ms_up > 0 else 0.0
    un = prof_dom.get("gemm_ffn_up_union", {"ms": 0.0, "launches": 0})
    if timed and un["launches"] == up["launches"] and un["ms"] > 0:
        union_ms = un["ms"]
    else:                                                       # no timed events (--no-profile): single-stream sections
        union_ms = up["ms"]
    ach = flop_up * up["launches"] / (union_ms * 1e-3) / 1e12 if union_ms > 0 else 0.0
    ex = prof["gemm_ffn_up"]
    ms_ex = ex["ms"] / max(ex["launches"], 1)
    ach_ex = flop_up_full / (ms_ex * 1e-3) / 1e12 if ms_ex > 0 else 0.0
    sections = {k: v for k, v in prof.items() if k != "gemm_ffn_up_union"}
    gemm_ms = sum(sections[s]["ms"] for s in ("gemm_qkv", "gemm_out", "gemm_ffn_up", "gemm_ffn_down", "head"))
    tot_ms = sum(v["ms"] for v in sections.values())
    lin_flop_fwd = M * (cfg.n_layers * (2 * cfg.d_model * (3 * cfg.d_model + cfg.d_model + 2 * cfg.ffn_hidden)
                                        + 2 * cfg.ffn_hidden * cfg.d_model)
                        + 2 * cfg.d_model * cfg.d_model + 2 * cfg.d_model * cfg.n_structure_heads)
    n_fwd = n_fwd_sample                                        # the breakdown pass is one step
    # roofline.traffic: the newest profiles/r*_gemm_traffic.json written by tools/pmc_traffic.py (separate rocprofv3 --pmc passes of
    # this kernel at this M) — reported only while the kernel source it was measured on is the source of the library in use
    traffic, traffic_note = None, "no PMC pass on record (tools/pmc_traffic.py measures it on the GPU box)"
    if not args.tiny:
        import hashlib
        src_hash = hashlib.sha256((ROOT / "esmdiff_amd" / "csrc" / "gemm256w4.hip").read_bytes()).hexdigest()
        for tp in sorted((ROOT / "profiles").glob("r*_gemm_traffic.json"), reverse=True):
            doc = json.loads(tp.read_text())
            rec = doc.get("by_rows", {}).get(str(M // parts))
            if not rec:
                continue
            if doc.get("kernel_source_sha256") != src_hash:
                traffic_note = (f"stale: profiles/{tp.name} was measured on another version of csrc/gemm256w4.hip "
                                f"({str(doc.get('kernel_source_sha256'))[:12]} vs {src_hash[:12]} now); re-run tools/pmc_traffic.py")
                break
            traffic = rec["traffic_bytes_per_launch"]
            traffic_note = (f"profiles/{tp.name} (tools/pmc_traffic.py; kernel source sha256 {src_hash[:12]} = the library's): "
                            "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, fabric level, Infinity Cache hits included")
            break
    passes = 3 if args.precision == "f32_split" else 1      # f32_split: three f16 MFMA products per split operand pair
    kname = {"bf16": "g4::gemm256w4_kernel<SWIGLU, 0>", "f16": "ed16::g4::gemm256w4_kernel<SWIGLU, 0>",
             "f32_split": "g4::gemm256w4_kernel<4, SPLIT> (fused SwiGLU, 3 f16 MFMA passes)", "f32": "gemm_f32_kernel<STORE>"}[args.precision]
    rep = {
        "roofline": {"bound": "mfma", "kernel": "%s (FFN-up, M=%d N=%d K=%d)" % (kname, M // parts, 2 * cfg.ffn_hidden, cfg.d_model),
                     "mfma_passes": passes,
                     # top level = the contract's figure: algorithmic FLOP of ONE launch / its mean duration in the timed
                     # configuration (HIP events on the launch stream); it is the number profiles/*_kernel_stats.txt
                     # (rocprofv3 --kernel-trace --stats of the same command) shows as the kernel's average
                     "achieved": round(ach_launch, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach_launch / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                     "traffic_unit": "bytes/launch", "traffic_source": traffic_note,
                     "what": ("timed region: algorithmic FLOP of one FFN-up launch / mean launch duration, HIP events on the launch "
                              "stream (%d stream(s): a launch covers M / streams rows and shares the GPU with the other stream's "
                              "kernels); reproducible from the rocprofv3 kernel-trace average" % parts),
                     "algorithmic_flop_per_launch": flop_up, "launches": up["launches"], "streams": parts,
                     "launch_ms": round(ms_up, 4),
                     "union": {"what": "labelled extra, device level: FFN-up FLOP of all streams / the UNION of the launches' busy "
                                       "intervals (engine-side merge of the event intervals; tools/rocprof_summary.py --union "
                                       "recomputes it from a kernel trace)",
                               "union_busy_ms": round(union_ms, 3), "achieved": round(ach, 1), "frac": round(ach / PEAK_BF16_TFLOPS, 4)},
                     "union_busy_ms": round(union_ms, 3),
                     "exclusive": {"what": "labelled extra: same kernel alone on the GPU at M=%d (single-stream breakdown pass, HIP events)" % M,
                                   "launch_ms": round(ms_ex, 4), "achieved": round(ach_ex, 1),
                                   "frac": round(ach_ex / PEAK_BF16_TFLOPS, 4)},
                     "all_gemm_tflops": round(lin_flop_fwd * n_fwd / (gemm_ms * 1e-3) / 1e12, 1) if gemm_ms else None},
        "sections_ms_per_forward": {k: round(v["ms"] / n_fwd, 3) for k, v in sections.items()},
        "sections_note": "per-launch HIP events of one extra untimed single-stream step; roofline.* is from the timed region",
        "device_ms_per_forward": round(tot_ms / n_fwd, 3),
    }
    if passes > 1:   # the figure above counts the ALGORITHMIC product once; the matrix pipes do `passes` times that
        rep["roofline"]["mfma_work_frac"] = round(passes * ach_launch / PEAK_BF16_TFLOPS, 4)
    return rep


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run this same command line under torch.distributed.run, one rank per
    GPU of this node (rendezvous on 127.0.0.1, a free port), and hand its exit code back.  Rank 0 of the child prints the ONE
    JSON line on the inherited stdout; the launcher's own chatter goes to stderr."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples-per-gpu", type=int, default=100)
    ap.add_argument("--residues", type=int, default=256)
    ap.add_argument("--num-steps", type=int, default=25)
    ap.add_argument("--tiny", action="store_true", help="small model (debug only; prints data=debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="debug: timed region without the in-stream HIP events")
    ap.add_argument("--no-breakdown", action="store_true",
                    help="skip the extra UNTIMED single-stream step that produces sections_ms_per_forward and roofline.exclusive: under "
                         "rocprofv3 every launch of the dominant kernel is then a launch of the timed configuration, so the trace's "
                         "average for it is directly the figure roofline.frac is computed from")
    ap.add_argument("--no-step0-sharing", action="store_true", help="skip the second, labelled run with exact step-0 sharing")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mode", choices=["ddpm", "gibbs"], default="ddpm",
                    help="gibbs: the CLI's default mode (entropy-ordered unmasking, temperature 1.4, top-p 0.9: "
                         "sample_esmdiff.py:66-130) — num_steps forwards instead of num_steps + 1; not the headline metric")
    ap.add_argument("--inpaint", type=str, default=None, metavar="A:B",
                    help="BASELINE configs[4]: residues A..B-1 start as MASK, all others carry fixed (synthetic) structure "
                         "tokens through input_prior (sample_esmdiff.py:196-209); use with --num-steps 50")
    ap.add_argument("--stub-engine", action="store_true",
                    help="CI only: a CPU stand-in engine (tests/standin_engine.py) over the gloo backend — exercises the launch, "
                         "process-group, gather and reporting path of this script without a GPU; prints data=debug-stub-engine")
    ap.add_argument("--precision", choices=["bf16", "f16", "f32", "f32_split"], default="bf16",
                    help="engine arithmetic; the headline metric is quoted at bf16 (BASELINE configs[1]); f32 / f32_split are the "
                         "float32-grade paths whose ids equal the float32 chain's")
    ap.add_argument("--head-precision", choices=["bf16", "f32"], default="bf16",
                    help="bf16 engine only: final LayerNorm + output head in float32 grade (esmdiff_config.head_precision)")
    ap.add_argument("--no-head-f32-leg", action="store_true", help="skip the second, labelled run with the float32-grade head")
    ap.add_argument("--checkpoint-fanout", choices=["auto", "on", "off"], default="auto",
                    help="load the weights the way a real run does: rank 0 writes the synthetic state dict to a file (untimed), then "
                         "rank 0 READS it and every rank receives the tensors by one broadcast (esmdiff_amd.dist.broadcast_state_dict); "
                         "per_rank reports load_s next to create_s.  auto: on under a launcher (N > 1 or --spawn), off otherwise")
    ap.add_argument("--alt-steps", type=int, default=5, help="timed steps of every labelled extra leg (alt_precisions, certified); >= 5")
    ap.add_argument("--spawn", action="store_true",
                    help="go through the launcher path (torch.distributed.run, process group, RCCL gather) even at --gpus 1: the "
                         "multi-GPU code path on a one-GPU box")
    args = ap.parse_args()

    # A number from this script is only as good as its environment: the launch-skipping debug switch makes
    # results wrong by construction — refuse to measure with it (the product library refuses too); every ESMDIFF_* variable and
    # the library actually loaded go into the line.
    env_seen = {k: v for k, v in sorted(os.environ.items()) if k.startswith("ESMDIFF_")}
    if "ESMDIFF_DEBUG_SKIP" in env_seen:
        raise SystemExit("bench.py: ESMDIFF_DEBUG_SKIP is set — it drops kernels from the forward (timing experiments only); "
                         "unset it, a number measured with it is invalid")
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run
    if not args.stub_engine and not launched and args.gpus > torch.cuda.device_count():
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {torch.cuda.device_count()} GPU(s) visible on this node")
    if (args.gpus > 1 or args.spawn) and not launched:
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): pass --nproc-per-node {args.gpus}")
    import torch.distributed as dist
    use_dist = launched
    stub = args.stub_engine
    if stub:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world)

    from esmdiff_amd.config import ESM3_OPEN, TINY
    from esmdiff_amd.schedule import ddpm_schedule
    from esmdiff_amd.weights import random_init_state_dict

    from esmdiff_amd.dist import pin_to_gpu_numa
    cfg = TINY if args.tiny else ESM3_OPEN
    B, L, T = args.samples_per_gpu, args.residues + 2, args.num_steps
    numa = pin_to_gpu_numa(None if stub else local_rank) if world > 1 else None   # host threads next to the rank's GPU
    load_rec = None
    t_create = time.perf_counter()
    if stub:
        from tests.standin_engine import StandinEngine
        sd = {}
        eng = StandinEngine(cfg, max_batch=B, max_len=L)
    else:
        from esmdiff_amd.engine import Engine
        sd = random_init_state_dict(cfg, seed=args.seed, device=str(dev), with_geom=True)   # same weights on every rank (GPU generator)
        fanout = args.checkpoint_fanout == "on" or (args.checkpoint_fanout == "auto" and launched)
        if fanout:
            # The start-up cost of a real run (SURVEY 8e names it THE scaling risk): the 5.5 GB float32 checkpoint of
            # checkpoint_utils.py:59-73.  Rank 0 writes the synthetic weights in that file format (untimed), then the weights are
            # loaded as the CLI loads them: one reader, one broadcast.
            import tempfile
            from esmdiff_amd.dist import broadcast_state_dict
            from esmdiff_amd.weights import load_checkpoint_state_dict
            ck = Path(tempfile.gettempdir()) / f"esmdiff_bench_ckpt_{os.environ.get('MASTER_PORT', 'solo')}_{cfg.n_layers}.pt"
            if rank == 0:
                torch.save({"module": {k: v.cpu() for k, v in sd.items()}}, ck)
            del sd
            if use_dist:
                dist.barrier(device_ids=[local_rank])
            torch.cuda.empty_cache()
            t_create = time.perf_counter()
            sd, load_rec = broadcast_state_dict(lambda: load_checkpoint_state_dict(ck), dev)
            t_create2 = time.perf_counter()
            if use_dist:
                dist.barrier(device_ids=[local_rank])
            if rank == 0:
                ck.unlink(missing_ok=True)
            t_create += time.perf_counter() - t_create2      # (the clean-up barrier is not part of create_s)
        eng = Engine(cfg, sd, max_batch=B, max_len=L, device=local_rank, precision=args.precision,
                     head_precision="f32" if args.head_precision == "f32" else None)
    create_s = time.perf_counter() - t_create - (load_rec["load_s"] if load_rec else 0.0)
    g = torch.Generator().manual_seed(args.seed)
    seq1 = torch.cat([torch.tensor([0]), torch.randint(4, 24, (args.residues,), generator=g), torch.tensor([2])])
    seq = seq1[None].repeat(B, 1).to(dev)
    sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
    gathered = [torch.empty(B, L * 2, dtype=torch.uint8, device=dev) for _ in range(world)] if use_dist else None

    prior = None
    if args.inpaint:
        a, b = (int(v) for v in args.inpaint.split(":"))
        prior = torch.randint(0, 4096, (1, L), generator=g).repeat(B, 1)
        prior[:, a:b] = 4096                                     # token-space indices, as the reference uses them (:200-201)
        prior = prior.to(dev)

    if args.mode == "gibbs":
        from esmdiff_amd.gibbs import unmask_schedule
        x0 = torch.full((B, L), 4096, dtype=torch.int64)
        x0[:, 0], x0[:, -1] = 4098, 4097                          # structure BOS / EOS
        table = torch.tensor(unmask_schedule(L - 2, T), dtype=torch.int32)[:, None].repeat(1, B)
        x0 = x0.to(dev)
        if args.inpaint:   # the CLI's inpainting (sample_esmdiff.py:88-96): known backbone conditions through block 0's
            from esmdiff_amd.geometry import build_affine3d_from_coordinates   # geometric attention, masked residues = Inf
            a, b = (int(v) for v in args.inpaint.split(":"))
            ca = torch.cumsum(torch.randn(L, 3, generator=g) * 2.2, 0)
            xyz = torch.stack([ca + torch.randn(L, 3, generator=g) * 0.8, ca, ca + torch.randn(L, 3, generator=g) * 0.8], 1)
            xyz[0], xyz[-1], xyz[a:b] = float("nan"), float("nan"), float("inf")
            eng.set_frames(*build_affine3d_from_coordinates(xyz[None].repeat(B, 1, 1, 1)))
            prior = None

    phase_marks = []      # per timed step: (start, sampled, gathered) stream events (GPU) or host times (stub)

    def mark():
        if stub:
            return time.perf_counter()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    # samples/s = N / wall-time(tokenise -> final ids on HOST).  A timed step therefore starts from the residue
    # STRING (tokenised on the host, sdk.encode_sequence, repeated over the batch and uploaded) and ends with the ids in pinned
    # host memory (an asynchronous D2H copy on the launch stream; the synchronisation that closes the region waits for it).
    seq_string = None
    if not stub and not args.inpaint:
        from esmdiff_amd import constants as C_
        from esmdiff_amd.sdk import encode_sequence
        seq_string = "".join(C_.SEQUENCE_VOCAB[int(i)] for i in seq1[1:-1])
        assert torch.equal(encode_sequence(seq_string), seq1)
    ids_host = None if stub else torch.empty(B * (world if use_dist else 1), L, dtype=torch.int16, pin_memory=True)

    def one_step(step_idx, engine=None, timed=False):
        e_ = eng if engine is None else engine
        m0 = mark() if timed else None
        seq_ = seq
        if timed and seq_string is not None:
            seq_ = encode_sequence(seq_string)[None].repeat(B, 1).pin_memory().to(dev, non_blocking=True)
        if args.mode == "gibbs":
            ids = e_.gibbs_sample(seq_, x0, table, 1.4, 0.9, seed=args.seed + step_idx, sample_offset=rank * B)
        else:
            ids = e_.ddpm_sample(seq_, sch, seed=args.seed + step_idx, sample_offset=rank * B, input_prior=prior)
        m1 = mark() if timed else None
        if use_dist:                                             # one exchange at the end: int16 ids over RCCL
            dist.all_gather(gathered, ids.to(torch.int16).view(torch.uint8))
        if timed and ids_host is not None:                       # ... and the ids to the host
            if use_dist:
                ids_host.copy_(torch.cat(gathered, 0).view(torch.int16), non_blocking=True)
            else:
                ids_host.copy_(ids.to(torch.int16), non_blocking=True)
        if timed:
            phase_marks.append((m0, m1, mark()))
        return ids

    def sync():
        if not stub:
            torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier() if stub else dist.barrier(device_ids=[local_rank])
            if not stub:
                torch.cuda.synchronize(dev)

    for w in range(args.warmup):
        one_step(1000 + w)
    sync()
    # HIP events on the launch stream around the dominant kernel only (96 per forward: <0.1 % overhead; bracketing
    # every launch costs 2.3 %), collected after the region
    eng.set_profiling(0 if args.no_profile else 2)
    power = PowerSampler(-1 if stub else local_rank)
    power.start()
    t0 = time.perf_counter()
    ids0 = None
    for k in range(args.steps):
        ids = one_step(k, timed=True)
        if k == 0:
            ids0 = ids
    sync()
    t1 = time.perf_counter()
    power_rec = power.stop()
    # per-rank breakdown (no extra synchronisation inside the region: stream events, read after it)
    if stub:
        sample_ms = [1e3 * (b - a) for a, b, _ in phase_marks]
        gather_ms = [1e3 * (c - b) for _, b, c in phase_marks]
    else:
        sample_ms = [a.elapsed_time(b) for a, b, _ in phase_marks]
        gather_ms = [b.elapsed_time(c) for _, b, c in phase_marks]
    rank_rec = {"rank": rank, "create_s": round(create_s, 2), "load_s": None if load_rec is None else load_rec["load_s"],
                "load": load_rec, "elapsed_s": round(t1 - t0, 4),
                "sample_ms_per_step": round(sum(sample_ms) / max(len(sample_ms), 1), 2),
                "gather_ms_per_step": round(sum(gather_ms) / max(len(gather_ms), 1), 3),
                "power_mean_w": None if not power_rec else power_rec["mean_w"],
                "sclk_mhz": None if not power_rec else power_rec["mean_sclk_mhz"], "numa": numa}
    per_rank = [rank_rec]
    if use_dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, rank_rec)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    prof_dom = eng.get_profile()
    sync_local = (lambda: None) if stub else (lambda: torch.cuda.synchronize(dev))
    if args.no_breakdown:
        prof = {k: {"ms": 0.0, "launches": 0} for k in prof_dom}
    else:
        eng.set_profiling(1)                                     # one extra, UNTIMED pass for the per-section breakdown
        one_step(2000)
        sync_local()
        prof = eng.get_profile()
    eng.set_profiling(0)
    assert int((ids == 4096).sum()) == 0
    # Second, LABELLED figure (never `value`): the same workload with the exact shortcuts — step-0 sharing (every sample of a
    # step starts from identical tokens, so the first of the T + 1 forwards runs on a sub-batch and serves all samples) and the
    # noise-removal skip (forward T + 1 only for samples that still hold a MASK); ids are bit-identical to the run above
    # (tests: test_step0_sharing_is_exact, test_final_skip_is_exact).  FLOP accounting uses the rows really executed.
    shared_rec = None
    if world == 1 and not stub and not args.no_step0_sharing:
        eng.set_step0_sharing(True)
        if args.mode == "ddpm":
            eng.set_final_skip(True)                             # the other exact shortcut (esmdiff_set_final_skip)
        same = bool(torch.equal(one_step(0), ids0)) if ids0 is not None else None   # step 0 again (same seed), shared this time
        sync_local()
        eng.counters(reset=True)
        ks = max(2, min(args.steps, 3))
        ts0 = time.perf_counter()
        for k in range(ks):
            one_step(k)
        sync_local()
        ts1 = time.perf_counter()
        cnt = eng.counters(reset=True)
        eng.set_step0_sharing(False)
        eng.set_final_skip(False)
        shared_rec = {"value": round(B * ks / (ts1 - ts0), 3), "unit": "samples/s", "steps": ks,
                      "forwards_executed_per_sample": round(cnt["token_rows"] / (B * L * ks), 4),
                      "token_rows_executed_per_step": cnt["token_rows"] // ks,
                      "ids_equal_to_unshared_run": same,
                      "what": "same workload with the two EXACT shortcuts on (ids bit-identical to the plain run, asserted here and in "
                              "tests/test_gpu_fullwidth.py): esmdiff_set_step0_sharing — at step 0 all samples have identical inputs, "
                              "one sub-batch forward serves them all; esmdiff_set_final_skip — the noise-removal forward runs only for "
                              "samples that still hold a MASK (none, almost always).  NOT the headline value"}

    # Further LABELLED figures (never `value`): the same workload at the other precisions of the 16-bit throughput path, each
    # timed in this process next to a fresh run of the headline engine (A/B inside one thermal state):
    #   head_f32      final LayerNorm + output head in float32 grade (esmdiff_config.head_precision = 1): the head is 0.6 % of the
    #                 FLOP and 64 % of the bf16 logit-error variance (profiles/r04_head_decomposition.json)
    #   f16           IEEE-half operands instead of bfloat16 on the f16 forms of the same MFMA instructions (csrc/ed_half.h):
    #                 same rate, same bytes, 1/8 of the operand rounding
    #   f16_head_f32  both
    # Their logit errors against the oracle are parity_spot_<name>; the id-level figures (flips per draw against the float32
    # chain at configs[1]'s full size) are measured by tests/test_gpu_strict.py and recorded in profiles/r04_parity_strict.json.
    alt_recs, alt_engines = {}, {}
    if world == 1 and not stub and args.precision == "bf16" and args.head_precision == "bf16" and not args.no_head_f32_leg:
        from esmdiff_amd.engine import Engine
        ks = max(5, args.alt_steps)

        def timed(engine):
            sync_local()
            ps = PowerSampler(local_rank)
            ps.start()
            ta = time.perf_counter()
            for k in range(ks):
                one_step(k, engine=engine)
            sync_local()
            dt = time.perf_counter() - ta
            pw = ps.stop()
            return dt, pw

        for name, kw in (("head_f32", {"head_precision": "f32"}), ("f16", {"precision": "f16"}),
                         ("f16_head_f32", {"precision": "f16", "head_precision": "f32"})):
            e2 = Engine(cfg, sd, max_batch=B, max_len=L, device=local_rank, **kw)
            one_step(1000, engine=e2)
            t_alt, pw_alt = timed(e2)
            t_base, pw_base = timed(eng)
            alt_engines[name] = e2
            pw = lambda r: None if not r else {"mean_w": r["mean_w"], "mean_sclk_mhz": r["mean_sclk_mhz"]}  # noqa: E731
            alt_recs[name] = {"value": round(B * ks / t_alt, 3), "unit": "samples/s", "steps": ks,
                              "headline_engine_same_session": round(B * ks / t_base, 3),
                              "cost_frac": round(t_alt / t_base - 1.0, 4), "engine": kw,
                              "power": pw(pw_alt), "headline_engine_power": pw(pw_base),
                              "what": "same workload, labelled extra — NOT the headline value"}

    # One more LABELLED figure: certified sampling (esmdiff_amd/certified.py, r05 form) — the f16 engine (f32-grade head) draws every
    # update of every sample; the sampler kernel flags the samples with a close call (winner within exp(2 eps) of the runner-up);
    # flagged sample-updates — and a random 2 % of the unflagged ones, the audit — are verified in batches of >= 32 on an F32_SPLIT
    # engine while the flagged samples continue speculatively; a verification that disagrees rolls the sample back.  Ids are
    # compared here with the F32_SPLIT engine's own chain for the same seeds (first and last timed step).
    cert_rec = None
    if alt_engines.get("f16_head_f32") is not None and args.mode == "ddpm" and not args.inpaint:
        from esmdiff_amd.certified import CertifiedSampler
        from esmdiff_amd.engine import Engine
        ex = Engine(cfg, sd, max_batch=B, max_len=L, device=local_rank, precision="f32_split")
        cs = CertifiedSampler(alt_engines["f16_head_f32"], ex)
        cs.ddpm_sample(seq, sch, seed=args.seed + 1000, sample_offset=rank * B)          # cold call: probe + allocations
        cold = cs.stats
        sync_local()
        ks, got, stats = max(5, args.alt_steps), [], []
        psamp = PowerSampler(local_rank)
        psamp.start()
        tc0 = time.perf_counter()
        for k in range(ks):
            got.append(cs.ddpm_sample(seq, sch, seed=args.seed + k, sample_offset=rank * B))
            stats.append(cs.stats)
        sync_local()
        tc1 = time.perf_counter()
        pw_c = psamp.stop()
        want = [ex.ddpm_sample(seq, sch, seed=args.seed + k, sample_offset=rank * B) for k in (0, ks - 1)]
        sync_local()
        tc2 = time.perf_counter()
        plain = alt_engines["f16_head_f32"].ddpm_sample(seq, sch, seed=args.seed, sample_offset=rank * B)
        # the same ks x B samples as ONE stream (the CLI's batch loop handed to the sampler in one call): a fast forward always takes
        # the B unfinished samples with the lowest indices, so the roll-back tail of one batch runs inside the next batch's forwards
        seq_all = seq[:1].expand(ks * B, L).contiguous()
        sync_local()
        tsa = time.perf_counter()
        got_stream = cs.ddpm_sample(seq_all, sch, seed=args.seed + 2000, sample_offset=rank * B)
        sync_local()
        tsb = time.perf_counter()
        st_stream = cs.stats
        stream_ok = all(bool(torch.equal(got_stream[k * B:(k + 1) * B],
                                         ex.ddpm_sample(seq, sch, seed=args.seed + 2000, sample_offset=rank * B + k * B))) for k in (0, ks - 1))
        tot = lambda key: sum(s_[key] for s_ in stats)    # noqa: E731
        cert_rec = {"value": round(B * ks / (tc1 - tc0), 3), "unit": "samples/s", "steps": ks,
                    "seconds_per_step": round((tc1 - tc0) / ks, 3), "tail_seconds_per_step": round(tot("tail_seconds") / ks, 3),
                    "streamed": {"value": round(ks * B / (tsb - tsa), 3), "unit": "samples/s", "samples": ks * B, "lane_width": B,
                                 "ids_equal_to_f32_split_chain": stream_ok, "audit_checked": st_stream["audit_checked"],
                                 "audit_mismatches": st_stream["audit_mismatches"], "eps_violations": st_stream["eps_violations"],
                                 "corrections": st_stream["corrections"], "tail_seconds": st_stream["tail_seconds"],
                                 "what": f"the same {ks} x {B} samples handed to the sampler in ONE call (one seed, global sample "
                                         "indices): no per-batch tail — roll-backs of one batch ride in the next batch's forwards"},
                    "certificate": stats[-1]["certificate"],
                    "eps": f"auto: pair bound P = max({cs.k_sigma:g} x r.m.s. neighbouring-pair error, {cs.max_factor:g} x largest row RANGE "
                           "(max e - min e: bounds ANY pair's error) seen), eps = P / 2",
                    "max_range_err_seen": stats[-1]["max_range_err_all_calls"],
                    "eps_used": [min(s_["eps_min_used"] for s_ in stats), max(s_["eps_max_used"] for s_ in stats)],
                    "sigma_pair_err": stats[-1]["sigma_pair_err"], "max_pair_err_seen": stats[-1]["max_pair_err_all_calls"],
                    "max_logit_err_seen": stats[-1]["max_logit_err_all_calls"],
                    "fast_engine": "f16 + f32-grade head", "exact_engine": "f32_split", "verify_batch": cs.verify_batch,
                    "f32_split_engine_alone_same_session": round(2 * B / (tc2 - tc1), 3),
                    "ids_equal_to_f32_split_chain": bool(torch.equal(got[0], want[0])) and bool(torch.equal(got[-1], want[1])),
                    "ids_checked_steps": [0, ks - 1],
                    "samples_identical_without_certification": int((plain == want[0]).all(1).sum()),
                    "rerun_share": round(tot("flagged") / max(1, tot("sample_forwards_fast")), 4),
                    "rerun_share_vs_eps": stats[-1]["rerun_share_vs_eps"],
                    "sample_forwards_fast": tot("sample_forwards_fast"), "sample_forwards_exact": tot("sample_forwards_exact"),
                    "corrections": tot("corrections"), "rollback_updates_discarded": tot("rollback_updates_discarded"),
                    "verify_batch_sizes": [s_["verify_batch_sizes"] for s_ in stats],
                    "eps_violations": tot("eps_violations"),
                    "audit": {"rate": cs.audit_rate, "rate_steady": cs.audit_rate_steady, "steady_after_clean_audits": cs.audit_clean_target,
                              "rate_now": stats[-1]["audit_rate_now"], "audit_checked": tot("audit_checked"), "audit_mismatches": tot("audit_mismatches"),
                              "audit_eps_violations": tot("audit_eps_violations"),
                              "audit_max_logit_err": max(s_["audit_max_logit_err"] for s_ in stats),
                              "audit_max_pair_err": max(s_["audit_max_pair_err"] for s_ in stats)},
                    "cold_call": {"eps_used": [cold["eps_min_used"], cold["eps_max_used"]], "eps_violations": cold["eps_violations"],
                                  "audit_checked": cold["audit_checked"], "audit_mismatches": cold["audit_mismatches"]},
                    "power": None if not pw_c else {"mean_w": pw_c["mean_w"], "mean_sclk_mhz": pw_c["mean_sclk_mhz"]},
                    "what": "same workload, ids of the float32-grade chain (tests/test_gpu_strict.py::"
                            "test_certified_sampler_equals_float32_chain_configs1_full_batch checks them against the exact-f32 engine "
                            "too).  Labelled extra — NOT the headline value"}
        # ... and the CLI's DEFAULT mode (gibbs: entropy-ordered unmasking, temperature 1.4, top-p 0.9; sample_esmdiff.py:66-130, :241)
        # through the same sampler: CertifiedSampler.gibbs_sample at configs[1]'s shape (all 256 residues sampled, 25 steps) and at
        # configs[4]'s shape (residues 96..159 sampled over 50 steps, the rest of a synthetic backbone conditions block 0's geometric
        # attention).  Two timed jobs each (the F32_SPLIT referee costs 5-11 s per job); ids of the first job against the F32_SPLIT
        # engine's own gibbs chain.
        gibbs_rec = None
        try:
            from esmdiff_amd.geometry import build_affine3d_from_coordinates
            from esmdiff_amd.gibbs import unmask_schedule
            gibbs_rec = {}
            gq = torch.Generator().manual_seed(args.seed + 7)
            for gname, n_masked, gsteps, with_xyz in (("configs1_shape", L - 2, T, False), ("configs4_shape_inpaint", 64, 50, True)):
                if with_xyz and not getattr(ex, "has_geom", False):
                    continue
                x0g = torch.full((B, L), 4096, dtype=torch.int64)
                x0g[:, 0], x0g[:, -1] = 4098, 4097
                frames = None
                if with_xyz:
                    x0g[:, 1:-1] = torch.randint(0, 4096, (1, L - 2), generator=gq)
                    x0g[:, 97:161] = 4096
                    ca = torch.cumsum(torch.nn.functional.normalize(torch.randn(L, 3, generator=gq), dim=-1) * 3.8, 0)
                    xyz = torch.stack([ca + torch.tensor([-1.2, 0.7, 0.0]), ca, ca + torch.tensor([1.3, 0.6, 0.1])], 1)
                    xyz[97:161] = float("inf")
                    xyz[0] = xyz[-1] = float("nan")
                    frames = build_affine3d_from_coordinates(xyz[None].repeat(B, 1, 1, 1))
                tab = torch.tensor(unmask_schedule(n_masked, gsteps), dtype=torch.int32)[:, None].repeat(1, B)
                run = lambda sd_: cs.gibbs_sample(seq, x0g, tab, 1.4, 0.9, seed=sd_, sample_offset=rank * B, frames=frames)   # noqa: E731
                run(args.seed + 3000)                                                   # cold for this mode: the entropy bound starts here
                sync_local()
                tg0 = time.perf_counter()
                g_out = [run(args.seed + 10 + k) for k in range(2)]
                g_stats = cs.stats
                sync_local()
                tg1 = time.perf_counter()
                if frames is not None:
                    ex.set_frames(*frames)
                g_want = ex.gibbs_sample(seq, x0g, tab, 1.4, 0.9, seed=args.seed + 10, sample_offset=rank * B)
                ex.set_frames(None) if frames is not None else None
                sync_local()
                tg2 = time.perf_counter()
                gibbs_rec[gname] = {
                    "value": round(2 * B / (tg1 - tg0), 3), "unit": "samples/s", "steps": 2, "masked_residues": n_masked, "num_steps": gsteps,
                    "f32_split_engine_alone_same_session": round(B / (tg2 - tg1), 3),
                    "ratio_to_f32_split": round((tg2 - tg1) * 2 / (tg1 - tg0), 3),
                    "ids_equal_to_f32_split_chain": bool(torch.equal(g_out[0], g_want)), "ids_checked_steps": [0],
                    "certificate": g_stats["certificate"], "flagged": g_stats["flagged"], "flag_reasons": g_stats["flag_reasons"],
                    "corrections": g_stats["corrections"], "audit_checked": g_stats["audit_checked"], "audit_mismatches": g_stats["audit_mismatches"],
                    "eps_violations": g_stats["eps_violations"], "entropy_violations": g_stats["entropy_violations"],
                    "sample_forwards_fast": g_stats["sample_forwards_fast"], "sample_forwards_exact": g_stats["sample_forwards_exact"],
                    "sample_forwards_direct": g_stats["sample_forwards_direct"], "direct_lane_switches": g_stats["direct_lane_switches"],
                    "pair_bound": None if g_stats["eps_max_used"] is None else 2 * g_stats["eps_max_used"],
                    "entropy_bound": g_stats.get("entropy_eps_max_used"),
                    "sigma_entropy_err": g_stats["sigma_entropy_err"]}
            gibbs_rec["what"] = ("the CLI's default mode through CertifiedSampler.gibbs_sample: ids of the F32_SPLIT engine's gibbs chain.  Speculation "
                                 "pays where few decisions are open (configs[4]'s shape); where most are (all positions nearly uniform at random "
                                 "initialisation: the ORDER of two entropies ~4e-5 apart is below what f16 resolves) the sampler runs the "
                                 "F32_SPLIT engine directly and costs what that engine costs.  Labelled extra — NOT the headline value")
            cert_rec["gibbs"] = gibbs_rec
        except Exception as ex_:      # a labelled extra must never take the headline line down with it
            cert_rec["gibbs"] = {"error": f"{type(ex_).__name__}: {ex_}", "partial": gibbs_rec}
        ex.close()

    if rank == 0:
        total_samples = B * world * args.steps
        value = total_samples / elapsed
        n_fwd_sample = T + 1 if args.mode == "ddpm" else int(table.shape[0])
        f_sample = flops_forward_per_sample(L, cfg) * n_fwd_sample
        out = {
            "metric": ("conformation samples/sec (256-res, 25 steps)" if (args.mode, args.residues, T) == ("ddpm", 256, 25) and not args.inpaint
                       else f"conformation samples/sec ({args.residues}-res, {T} steps, {args.mode}{', inpaint' if args.inpaint else ''})"),
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "f16": "f16", "f32": "f32", "f32_split": "f16x2-split (f32 grade)"}[args.precision],
            "data": "debug-stub-engine" if stub else ("synthetic" if not args.tiny else "debug-tiny-model"),
            "config": {"workload": workload_name(args, world),
                       "samples_per_gpu": B, "L_tok": L, "num_steps": T, "forwards_per_sample": n_fwd_sample, "mode": args.mode,
                       "layers": cfg.n_layers, "d_model": cfg.d_model, "noise": "philox4x32-10",
                       "parallelism": (f"sample-sharded x{world}, one RCCL all_gather of int16 ids per step" if use_dist else
                                       "single process, one GPU, no process group (nothing crosses RCCL at N=1)")},
            "flop_per_sample": f_sample,
            "mfma_frac_whole_job": round(value / world * f_sample / (PEAK_BF16_TFLOPS * 1e12), 4),
            "timed_region": ("BASELINE.md section 4, tokenise -> final ids on host.  t0: engines created, warm-up done, device synchronised "
                             "+ barrier; the input is the residue STRING.  Inside: K x [tokenise on the host, repeat over the batch, "
                             "upload; whole sampling loop on the device (T + 1 forwards + fused sampler launches)" +
                             (", one RCCL all_gather of the int16 ids" if use_dist else "") +
                             "; asynchronous D2H copy of the int16 ids into pinned host memory].  t1: after device synchronise "
                             "(+ barrier): the ids of every step are on the HOST.  Not inside: VQ-VAE decode, PDB I/O"),
            "environment": {"lib_path": None if stub else str(__import__("esmdiff_amd._native", fromlist=["lib_path"]).lib_path()),
                            "esmdiff_env": env_seen, "precision": args.precision, "head_precision": args.head_precision},
            "per_rank": per_rank,
            "slowest_rank": max(per_rank, key=lambda r: r["elapsed_s"])["rank"],
        }
        # What the per-rank figures PREDICT for the scaling efficiency the driver computes from the per-N values (weak scaling,
        # no data-path collective): inside the timed region only the gather and rank skew can cost anything; start-up (checkpoint
        # fan-out + engine create) is outside it and is reported as the job-level figure a user of the CLI would see.
        slow = max(per_rank, key=lambda r: r["elapsed_s"])
        samp = [r["sample_ms_per_step"] for r in per_rank]
        startup = max((r["create_s"] or 0.0) + (r["load_s"] or 0.0) for r in per_rank)
        out["scaling_prediction"] = {
            "timed_region_efficiency": round(min(samp) / max(1e-9, slow["sample_ms_per_step"] + slow["gather_ms_per_step"]), 4),
            "gather_share_of_step": round(slow["gather_ms_per_step"] / max(1e-9, slow["sample_ms_per_step"] + slow["gather_ms_per_step"]), 5),
            "rank_skew": round(max(samp) / max(1e-9, min(samp)) - 1.0, 4),
            "startup_s_max_rank": round(startup, 2),
            "job_efficiency_incl_startup_at_these_steps": round(elapsed / (elapsed + startup), 4),
            "what": "timed_region_efficiency = fastest rank's sampling time / (slowest rank's sampling + gather time): the efficiency "
                    "value_N / (N x value_1) should show if every GPU holds the clock the N = 1 run held; startup = load_s "
                    "(one reader + one broadcast of the 5.5 GB checkpoint) + create_s, outside the timed region"}
        out["config"]["precision"] = args.precision
        out["config"]["head_precision"] = args.head_precision
        if stub:
            out["roofline"] = None
            out["note"] = "stand-in engine on CPU over gloo: launch / process-group / reporting path only, not a measurement"
        else:
            out.update(roofline_report(args, cfg, B, L, n_fwd_sample, prof_dom, prof))
            out["power"] = power_rec
            if alt_recs:
                out["alt_precisions"] = alt_recs
            if cert_rec is not None:
                if "gibbs" in cert_rec:
                    out["certified_gibbs"] = cert_rec.pop("gibbs")
                out["certified"] = cert_rec
            if shared_rec is not None:
                shared_rec["flop_per_sample_executed"] = flops_forward_per_sample(L, cfg) * shared_rec["forwards_executed_per_sample"]
                shared_rec["mfma_frac_whole_job"] = round(shared_rec["value"] * shared_rec["flop_per_sample_executed"] / (PEAK_BF16_TFLOPS * 1e12), 4)
                out["exact_shortcuts"] = shared_rec
        if world == 1 and not args.no_cpu_baseline and not stub:
            try:
                out["cpu_baseline"], spots = cpu_baseline(cfg, sd, L, T, {"value": eng, **alt_engines})
                out["parity_spot"] = spots.get("value")
                for name in alt_engines:
                    out["parity_spot_" + name] = spots.get(name)
            except Exception as ex:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    eng.close()
    for e2 in alt_engines.values():
        e2.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
