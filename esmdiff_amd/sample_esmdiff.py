"""Sampling CLI with the reference's interface (/root/reference/slm/sample_esmdiff.py:236-294):

    python -m esmdiff_amd.sample_esmdiff --input data/targets/bpti --ckpt release_v0.pt --mode ddpm \\
           --num_steps 25 --num_samples 100 --output output/inference_esmdiff

Same flags and defaults (--input --ckpt --output --mode{gibbs,ddpm} --num_steps --num_samples --mask_ids).
Extensions: --seed, --random_init (ESM3-open-sized random weights when no checkpoint is available offline),
--synthetic_len (a random sequence instead of --input), --n_max_residue_square (the reference's batching
heuristic, sample_esmdiff.py:146; the default here is larger because an MI355X holds 288 GB), --parity
(draw uniforms like the reference's CPU path).  Multi-GPU: launch with torch.distributed.run; the samples are
sharded over ranks and rank 0 writes the output.

Both sampling modes are built: `--mode ddpm` (MDLM ancestral sampler) and the reference's default `--mode gibbs`
(entropy-ordered iterative unmasking, temperature 1.4 / top-p 0.9).  `--precision` defaults to `certified` in both modes (r06): the
ids of a float32-grade run of the same seed (esmdiff_amd/certified.py: an f16 engine draws, the decisions its measured error
leaves open are verified on an f32_split engine; a statistical certificate whose counters go into the run's json); `--precision
bf16` is the throughput path the benchmark's headline is quoted on.

Output: the reference decodes tokens to backbone coordinates with ESM3's VQ-VAE decoder and writes a
multi-MODEL PDB (sample_esmdiff.py:225-231).  With --decoder_ckpt (esm's StructureTokenDecoder weights, which cannot be
downloaded here) every rank decodes its own samples on the device and rank 0 writes `<out>/<name>.pdb` (one MODEL per
sample, B-factor = pLDDT); the structure TOKENS are always written to `<out>/<name>.tokens.npy` (N, L int16) next to
a JSON with the run settings.
"""
from __future__ import annotations

import argparse
import json
import os
from pathlib import Path
from time import strftime, time

import numpy as np
import torch

from . import constants as C
from .dist import gather_ids, shard_samples
from .pdbio import timer
from .sdk import ESMProtein, encode_sequence

DEFAULT_NMAX = 1026 * 1026 * 32


def batch_sizes(n_tokens: int, num_samples: int, n_max_residue_square: int = DEFAULT_NMAX, cap: int = 0):
    """sample_esmdiff.py:181-193 (the length is the TOKEN count in ddpm mode, the residue count in gibbs mode).  The
    reference's arithmetic can make the remainder batch larger than the regular ones; `cap` (the engine's max_batch)
    additionally splits any batch the engine could not hold — samples are independent and the Philox noise is keyed by the
    global sample index, so the split draws the same noise for every sample.  The LOGITS may differ at bf16 rounding level
    between two splits (the engine picks its GEMM kernel, stream count and small-batch path by row count, DESIGN 3.1b /
    3.8), so a near-tie can resolve to another id; an engine created with precision="f32" is batch-independent bit for bit.
    In --parity mode (the reference's torch.rand stream, drawn per batch) a cap split would reorder that stream: the CLI
    refuses it there."""
    sq = n_tokens * n_tokens
    total = sq * num_samples
    bsz = [n_max_residue_square // sq] * (total // n_max_residue_square)
    if total % n_max_residue_square > 0:
        bsz.append(num_samples - sum(bsz))
    assert sum(bsz) == num_samples, f"{sum(bsz)} != {num_samples}"
    if cap > 0:
        bsz = [c for b in bsz for c in ([cap] * (b // cap) + ([b % cap] if b % cap else []))]
    return bsz


def engine_capacity(lengths, per_rank: int, n_max_residue_square: int, mode: str) -> int:
    """max_batch an engine needs so that every batch `batch_sizes` will issue for these targets fits: ddpm mode batches
    by token count (L + 2), gibbs mode by residue count (sample_esmdiff.py:97-105 vs :181-193)."""
    need = 1
    for n_res in lengths:
        n = n_res + 2 if mode == "ddpm" else n_res
        need = max([need] + batch_sizes(n, per_rank, n_max_residue_square))
    return need


@torch.no_grad()
def decode_tokens(tokens: torch.Tensor, decoder, chunk: int = 0, return_ptm: bool = False):
    """sample_esmdiff.py:40-61 without the file: structure tokens (n, L) WITHOUT BOS/EOS -> (coords (n, L, 3, 3) float32,
    plddt (n, L) float32 or None) on the decoder's device.  The reference decodes one sample per esm3.decode call; the
    decoder engine takes them `chunk` (default: its max_batch) at a time.  return_ptm: a third value, pTM (n,) or None
    (decoder_output["ptm"], /root/reference/slm/models/utils.py:73-76) when the decoder carries the pairwise head."""
    n, L_ = tokens.shape
    dev = decoder.device
    chunk = chunk or getattr(decoder, "max_batch", 64)
    want_ptm = return_ptm and getattr(decoder, "has_ptm", False)
    if n == 0:
        empty = (torch.empty(0, L_, 3, 3, device=dev), (torch.empty(0, L_, device=dev) if decoder.has_plddt else None))
        return empty + ((torch.empty(0, device=dev) if want_ptm else None),) if return_ptm else empty
    t = tokens.to(dev, torch.int64)
    full = torch.cat([torch.full((n, 1), C.STRUCTURE_BOS_TOKEN, dtype=torch.int64, device=dev), t,
                      torch.full((n, 1), C.STRUCTURE_EOS_TOKEN, dtype=torch.int64, device=dev)], 1)
    cs, ps, ts = [], [], []
    for i in range(0, n, chunk):
        if want_ptm:
            c, pl, tm = decoder.decode(full[i:i + chunk], return_plddt=True, return_ptm=True)
            ts.append(tm.clone())
        else:
            c, pl = decoder.decode(full[i:i + chunk], return_plddt=True)
        cs.append(c.clone())
        ps.append(None if pl is None else pl.clone())
    out = (torch.cat(cs, 0), (None if ps[0] is None else torch.cat(ps, 0)))
    return out + ((torch.cat(ts, 0) if want_ptm else None),) if return_ptm else out


def write_models_pdb(coords, plddt, sequence: str, save_to: Path, sample_basename: str):
    """sample_esmdiff.py:225-231: one PDB per sample in a temporary directory (B-factor column = pLDDT, as
    ESMProtein.to_pdb writes it) -> merge_pdbfiles into `<basename>.pdb` (MODEL n ... ENDMDL blocks)."""
    import tempfile
    from .pdbio import merge_pdbfiles, write_backbone_pdb
    coords = coords.cpu().numpy()
    plddt = None if plddt is None else plddt.cpu().numpy()
    seq = sequence.replace(C.MASK_RESIDUE, "X")
    with tempfile.TemporaryDirectory() as tmpdirname:
        saved = []
        for i in range(coords.shape[0]):
            tmp = Path(tmpdirname) / f"{sample_basename}.{i}.pdb"
            write_backbone_pdb(tmp, seq, coords[i], None if plddt is None else plddt[i])
            saved.append(tmp)
        merge_pdbfiles(saved, save_to, verbose=False)


def decode_shard_and_gather(local_tokens: torch.Tensor, decoder, num_samples: int, return_ptm: bool = False):
    """Multi-GPU tail: EVERY rank decodes the samples it drew, then one all_gather of coordinates (+ pLDDT, + pTM with
    return_ptm) — the decode time stays 1/N of the ensemble instead of rank 0 decoding all N shards after the id gather."""
    from .dist import gather_rows
    coords, plddt, ptm = decode_tokens(local_tokens, decoder, return_ptm=True)
    coords = gather_rows(coords.contiguous(), num_samples)
    plddt = None if plddt is None else gather_rows(plddt.contiguous(), num_samples)
    if not return_ptm:
        return coords, plddt
    ptm = None if ptm is None else gather_rows(ptm.reshape(-1, 1).contiguous(), num_samples)[:, 0]
    return coords, plddt, ptm


@torch.no_grad()
def decode_to_pdb(tokens: torch.Tensor, sequence: str, decoder, save_to: Path, sample_basename: str, chunk: int = 0):
    """Single-process convenience: decode_tokens + write_models_pdb."""
    coords, plddt = decode_tokens(tokens, decoder, chunk)
    write_models_pdb(coords, plddt, sequence, save_to, sample_basename)


def certified_record(cs) -> dict:
    """What the run's json says about a certified call (esmdiff_amd/certified.py): the kind of certificate and its counters."""
    keys = ("certificate", "mode", "eps_min_used", "eps_max_used", "entropy_eps_max_used", "flagged", "flag_reasons", "corrections",
            "audit_checked", "audit_mismatches", "audit_rate_now", "eps_violations", "entropy_violations", "sample_forwards_fast",
            "sample_forwards_exact")
    return {k: cs.stats.get(k) for k in keys if k in cs.stats}


@timer
@torch.no_grad()
def ddpm_sample_by_esm(sequence, pl_model, output_dir: Path, sample_basename: str, num_samples: int = 5,
                       num_steps: int = 10, eps: float = 1e-5, n_max_residue_square: int = DEFAULT_NMAX,
                       coordinates=None, mask_ids=None, structure_tokens=None, sample_max_t: float = 1.0,
                       seed: int = 0, noise: str = "philox", timestamp: bool = True, decoder=None, encoder=None):
    """sample_esmdiff.py:137-233.  With a `decoder` (esmdiff_amd.engine.StructureDecoder, one per rank) every rank decodes
    its own shard and rank 0 writes the reference's artefact, `<basename>.pdb` with one MODEL per sample; the token file
    is written either way.  `structure_tokens` (L+2, with BOS/EOS) replaces what the reference gets from
    ESM3.encode(coordinates) for the inpainting prior (:196-201); without it mask_ids cannot be honoured."""
    model = pl_model
    str_time = ("_" + strftime("%Y%m%d-%H%M%S")) if timestamp else ""
    output_dir = Path(output_dir) / f"step{num_steps}_eps{eps}_N{num_samples}{str_time}"
    save_to = output_dir / f"{sample_basename}.tokens.npy"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        print(f"Results will save to {save_to}")
    if save_to.exists():
        print(f"Skip existing {save_to}")
        return None
    if mask_ids is not None and structure_tokens is None and coordinates is not None and encoder is not None:
        # protseq_to_data (models/utils.py:116-137): the masked residues lose their coordinates, the rest is tokenised by
        # the VQ-VAE encoder; BOS / EOS as esm's tokenizer adds them
        xyz = torch.as_tensor(coordinates, dtype=torch.float32).clone()
        for idx in mask_ids:
            assert 0 <= idx < len(sequence), f"Invalid mask index {idx} for sequence of length {len(sequence)}"
            xyz[idx] = float("Inf")
        body = encoder.encode(xyz[None, :, :3, :])[0].cpu()
        structure_tokens = torch.cat([torch.tensor([C.STRUCTURE_BOS_TOKEN]), body, torch.tensor([C.STRUCTURE_EOS_TOKEN])])
    if mask_ids is not None:
        assert structure_tokens is not None, "Need structure tokens of the known residues (or coordinates + an encoder) for masking"
        seq_l = list(sequence)
        for idx in mask_ids:
            assert 0 <= idx < len(seq_l), f"Invalid mask index {idx} for sequence of length {len(seq_l)}"
            seq_l[idx] = C.MASK_RESIDUE
        sequence = "".join(seq_l)
    seq_tok = encode_sequence(sequence)
    start_t = time()
    offset, count = shard_samples(num_samples, world, rank)
    outs = []
    done = 0
    cap = getattr(getattr(model, "net", model), "max_batch", 0)
    if noise == "torch-cpu" and count and batch_sizes(seq_tok.numel(), count, n_max_residue_square, cap) != \
            batch_sizes(seq_tok.numel(), count, n_max_residue_square):
        raise ValueError(f"--parity replays the reference's per-batch torch.rand stream, but the engine capacity ({cap}) would "
                         "split the reference's batches and reorder it: build the engine with a larger max_batch")
    sizes = batch_sizes(seq_tok.numel(), count, n_max_residue_square, cap) if count else []
    if getattr(model, "certified", None) is not None and noise == "philox" and count:
        # certified sampling streams: the reference's batches (sample_esmdiff.py:181-216) go to the sampler as ONE call; it runs
        # the fast engine's max_batch unfinished samples with the lowest indices per forward, so a sample that is rolled back rides
        # along with later ones instead of ending its batch with a tail of small forwards.  Same ids (Philox by global index).
        sizes = [count]
    for bs in sizes:
        batch = seq_tok[None, :].repeat(bs, 1)
        prior = None
        if mask_ids is not None:
            prior = torch.as_tensor(structure_tokens, dtype=torch.int64)[None, :].repeat(bs, 1)
            for idx in mask_ids:                       # token-space index, exactly like sample_esmdiff.py:200-201
                prior[:, idx] = C.STRUCTURE_MASK_TOKEN
        outs.append(model.ddpm_sample(num_steps=num_steps, sequence_tokens=batch, eps=eps, input_prior=prior,
                                      sample_max_t=sample_max_t, seed=seed, sample_offset=offset + done, noise=noise))
        done += bs
    L = seq_tok.numel()
    local = torch.cat(outs, 0) if outs else torch.empty(0, L, dtype=torch.int64, device=model.device)
    tokens = gather_ids(local, num_samples)[:, 1:-1]          # remove bos and eos positions (:220-221)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    sample_t = time() - start_t
    coords = plddt = ptm = None
    if decoder is not None:                                   # every rank: decode the local shard, then one gather
        coords, plddt, ptm = decode_shard_and_gather(local[:, 1:-1], decoder, num_samples, return_ptm=True)
    if rank == 0:
        print(f"Sampling token time: {sample_t:.2f}s")
        eng_ = getattr(model, "fast", None) or getattr(model, "net", None)
        if hasattr(eng_, "describe_plan"):       # what produced these ids: the library build and the dispatch plan of the last batch
            from . import _native
            print(f"Engine: {_native.build_info()}")
            if outs:
                print(f"Plan:   {eng_.describe_plan(int(outs[-1].shape[0]), int(local.shape[1]))}")
        output_dir.mkdir(parents=True, exist_ok=True)
        np.save(save_to, tokens.cpu().numpy().astype(np.int16))
        (output_dir / f"{sample_basename}.json").write_text(json.dumps(
            {"sequence": sequence, "num_steps": num_steps, "num_samples": num_samples, "eps": eps, "seed": seed,
             "noise": noise, "world_size": world, "sampling_seconds": round(sample_t, 3),
             "precision": "certified" if getattr(model, "certified", None) is not None else getattr(getattr(model, "net", model), "precision", None),
             "head_precision": getattr(getattr(model, "fast", None) or getattr(model, "net", model), "head_precision", None),
             **({} if getattr(model, "certified", None) is None else {"certified": certified_record(model.certified)}),
             "decoder_precision": getattr(decoder, "precision", None),
             **({} if ptm is None else {"ptm": [round(float(v), 4) for v in ptm.cpu()]})}, indent=1))
        if coords is not None:
            write_models_pdb(coords, plddt, sequence, output_dir / f"{sample_basename}.pdb", sample_basename)
        print(f"Total time: {time() - start_t:.2f}s")
    return []


@timer
@torch.no_grad()
def minibatch_gibbs_by_esm(protseq, esm3_model, output_dir: Path, sample_basename: str, num_samples: int = 10,
                           num_steps: int = 16, temperature: float = 1.4, top_p: float = 0.9,
                           n_max_residue_square: int = DEFAULT_NMAX, coordinates=None, mask_ids=None,
                           structure_tokens=None, seed: int = 0, timestamp: bool = True, decoder=None):
    """sample_esmdiff.py:66-130: batches of ESMProtein copies through iterative_sampling_raw with
    GenerationConfig(track="structure", num_steps, temperature, top_p).  Inpainting as the reference does it (:88-96):
    `mask_ids` needs `coordinates` (L, >=3, 3); the masked residues get sequence '_' and coordinates Inf, and the known
    backbone conditions the model through block 0's geometric attention while every structure token is sampled
    (esm's condition_on_coordinates_only default).  Extension: `structure_tokens` (L,) of the known residues, when the
    caller has them, are kept fixed as well."""
    from .gibbs import iterative_sampling_raw
    from .sdk import GenerationConfig
    str_time = ("_" + strftime("%Y%m%d-%H%M%S")) if timestamp else ""
    output_dir = Path(output_dir) / f"T{temperature}_step{num_steps}_topp{top_p}_N{num_samples}{str_time}"
    save_to = output_dir / f"{sample_basename}.tokens.npy"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        print(f"Results will save to {save_to}")
    if save_to.exists():
        print(f"Skip existing {save_to}")
        return None
    st = None if structure_tokens is None else torch.as_tensor(structure_tokens, dtype=torch.int64).clone()
    if coordinates is not None:
        coordinates = torch.as_tensor(coordinates, dtype=torch.float32).clone()
    if mask_ids is not None:
        print(f"Masking {len(mask_ids)} residues and inpainting...")
        assert coordinates is not None or st is not None, "Need to provide coordinates for masking"
        protseq = list(protseq)
        for idx in mask_ids:
            assert 0 <= idx < len(protseq), f"Invalid mask index {idx} for sequence of length {len(protseq)}"
            protseq[idx] = C.MASK_RESIDUE
            if coordinates is not None:
                coordinates[idx] = float("Inf")
            if st is not None:
                st[idx] = C.STRUCTURE_MASK_TOKEN
        protseq = "".join(protseq)
    start_t = time()
    offset, count = shard_samples(num_samples, world, rank)
    out_list, done = [], 0
    cap = getattr(getattr(esm3_model, "net", esm3_model), "max_batch", 0)
    sizes = batch_sizes(len(protseq), count, n_max_residue_square, cap) if count else []
    if getattr(esm3_model, "certified", None) is not None and count:
        sizes = [count]                    # certified sampling streams all of a target's prompts through its fast lane (see ddpm_sample_by_esm)
    for bs in sizes:
        prot_list = [ESMProtein(sequence=protseq, coordinates=coordinates, structure_tokens=st) for _ in range(bs)]
        if rank == 0:
            print(f"Generating {len(prot_list)} samples for {protseq}...")
        cfg_list = [GenerationConfig(track="structure", num_steps=num_steps, temperature=temperature, top_p=top_p)
                    for _ in range(bs)]
        outs = iterative_sampling_raw(esm3_model, proteins=prot_list, configs=cfg_list, seed=seed,
                                      sample_offset=offset + done, decoder=decoder)
        out_list += outs
        done += bs
    dev = getattr(esm3_model, "net", esm3_model).device
    out_tokens = [o.structure_tokens for o in out_list]
    local = (torch.stack(out_tokens) if out_tokens else torch.empty(0, len(protseq), dtype=torch.int64)).to(dev)
    tokens = gather_ids(local, num_samples)
    coords = plddt = ptm = None
    if decoder is not None:       # the proteins already carry the coordinates their rank decoded (iterative_sampling_raw)
        from .dist import gather_rows
        Lr = len(protseq)
        if getattr(decoder, "has_ptm", False):
            lt = torch.stack([torch.as_tensor(o.ptm) for o in out_list]).reshape(-1, 1) if out_list else torch.empty(0, 1)
            ptm = gather_rows(lt.to(dev, torch.float32).contiguous(), num_samples)[:, 0]
        lc = torch.stack([torch.as_tensor(o.coordinates) for o in out_list]) if out_list else torch.empty(0, Lr, 3, 3)
        coords = gather_rows(lc.to(dev, torch.float32).contiguous(), num_samples)
        if decoder.has_plddt:
            lp = torch.stack([torch.as_tensor(o.plddt) for o in out_list]) if out_list else torch.empty(0, Lr)
            plddt = gather_rows(lp.to(dev, torch.float32).contiguous(), num_samples)
    if rank == 0:
        print(f"Sampling token time: {time() - start_t:.2f}s")
        output_dir.mkdir(parents=True, exist_ok=True)
        np.save(save_to, tokens.cpu().numpy().astype(np.int16))
        (output_dir / f"{sample_basename}.json").write_text(json.dumps(
            {"sequence": protseq, "mode": "gibbs", "num_steps": num_steps, "num_samples": num_samples,
             "temperature": temperature, "top_p": top_p, "seed": seed, "world_size": world,
             "sampling_seconds": round(time() - start_t, 3),
             "precision": "certified" if getattr(esm3_model, "certified", None) is not None else getattr(getattr(esm3_model, "net", esm3_model), "precision", None),
             **({} if getattr(esm3_model, "certified", None) is None else {"certified": certified_record(esm3_model.certified)}),
             **({} if ptm is None else {"ptm": [round(float(v), 4) for v in ptm.cpu()]})}, indent=1))
        if coords is not None:
            write_models_pdb(coords, plddt, protseq, output_dir / f"{sample_basename}.pdb", sample_basename)
    return out_list


def get_argparser(argv=None):
    p = argparse.ArgumentParser(description="Evaluate the ensemble of protein structures.")
    p.add_argument("--input", type=str, default="data/targets/bpti", help="Path to the data directory.")
    p.add_argument("--ckpt", type=str, default=None, help="Path to the model checkpoint.")
    p.add_argument("--output", type=str, default="output/inference_esmdiff")
    p.add_argument("--mode", type=str, default="gibbs", choices=["gibbs", "ddpm"])
    p.add_argument("--num_steps", type=int, default=25, help="Number of denoising steps.")
    p.add_argument("--num_samples", type=int, default=10, help="Number of samples to generate.")
    p.add_argument("--mask_ids", type=str, default=None, help="Comma-separated list of masked indices.")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--random_init", action="store_true", help="ESM3-open-sized random weights instead of --ckpt")
    p.add_argument("--esm3_ckpt", type=str, default=None,
                   help="state dict of the stock esm3_sm_open_v1 model (torch.save): what the reference samples from in gibbs "
                        "mode when --ckpt is absent (it downloads it; this engine cannot)")
    p.add_argument("--tiny", action="store_true", help=argparse.SUPPRESS)   # tests: 2-block model with --random_init
    p.add_argument("--decoder_ckpt", type=str, default=None,
                   help="state dict of esm's StructureTokenDecoder (esm3_structure_decoder_v0): also write <name>.pdb "
                        "with one MODEL per sample, as the reference does; without it only <name>.tokens.npy is written")
    p.add_argument("--random_init_decoder", action="store_true", help="random decoder weights (plumbing / tests)")
    p.add_argument("--encoder_ckpt", type=str, default=None,
                   help="state dict of esm's StructureTokenEncoder (esm3_structure_encoder_v0): lets --mode ddpm --mask_ids "
                        "build its prior from the input PDB's coordinates, as the reference does")
    p.add_argument("--random_init_encoder", action="store_true", help="random encoder weights (plumbing / tests)")
    p.add_argument("--synthetic_len", type=int, default=0, help="sample a random sequence of this length")
    p.add_argument("--n_max_residue_square", type=int, default=DEFAULT_NMAX)
    p.add_argument("--parity", action="store_true", help="uniforms from torch's CPU generator, like the reference")
    p.add_argument("--precision", choices=["bf16", "f16", "f32", "f32_split", "certified"], default="certified",
                   help="arithmetic of the sampling network.  certified (default, both modes) = the ids of the float32-grade (f32_split) "
                        "chain at ~2.4x its rate: an f16 engine draws every update and only the decisions its measured logit error leaves "
                        "open are verified on an f32_split engine, in batches, with an audit (a statistical certificate: k-sigma of the "
                        "measured error + audit, counters in the run's json); bf16 = the MFMA throughput path (the benchmark's headline; "
                        "near-tie draws can differ from a float32 run); f16 = the bf16 path with IEEE-half operands (same speed, 1/8 of the "
                        "rounding error); f32_split = float32-grade linears as three f16 MFMA passes over split operands (~1/3 of bf16); f32 = "
                        "the strict path, the reference's own float32 arithmetic on the f32-input MFMA (~1/10; the referee)")
    p.add_argument("--head_precision", choices=["body", "f32", "bf16", "f16"], default=None,
                   help="bf16 / f16 networks: 'f32' = final LayerNorm + output head in float32 grade (+1 %% time, fewer near-tie "
                        "flips), 'body' (aliases: 'bf16', 'f16') = the head in the network's own precision.  Default: 'body', except "
                        "--precision certified, whose fast engine gets the float32-grade head (the validated configuration)")
    p.add_argument("--decoder_precision", choices=["f32", "f32_split", "bf16"], default="f32",
                   help="arithmetic of the VQ-VAE structure decoder (and encoder): f32 (default, backbone within 1e-4 A of a float32 "
                        "decode, encoder codes equal to a float32 encoder's) or bf16")
    p.add_argument("--no_timestamp", action="store_true")
    return p.parse_args(argv)


def main(argv=None):
    args = get_argparser(argv)
    if args.head_precision in ("body", "bf16", "f16"):   # explicit: the head in the body's precision, also for the certified sampler's fast engine
        args.head_precision = {"bf16": "bf16", "f16": "f16", "certified": "f16"}.get(args.precision)
    if args.parity and args.precision == "certified":
        args.precision = "f32_split"       # the reference's torch.rand stream is drawn per batch on one engine: the float32-grade one
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    from .dist import pin_to_gpu_numa
    numa = pin_to_gpu_numa(local_rank)     # one process per GPU: launch thread, weight upload and PDB writer stay on the GPU's socket
    if args.ckpt is None and not args.random_init and not args.esm3_ckpt:
        # the reference falls back to the stock esm3_sm_open_v1 weights in gibbs mode (sample_esmdiff.py:252-255);
        # they cannot be fetched offline, so a checkpoint (or --random_init) is required in both modes here
        assert args.mode == "gibbs" or args.ckpt is not None, \
            "Only Gibbs sampling is supported for the pre-trained ESM3 model."
        raise SystemExit("no weights: pass --ckpt <release_v0.pt>, --esm3_ckpt <esm3_sm_open_v1 state dict> (gibbs mode) or "
                         "--random_init (synthetic weights)")
    if args.esm3_ckpt and args.ckpt is None:
        assert args.mode == "gibbs", "Only Gibbs sampling is supported for the pre-trained ESM3 model."
    if args.parity and world > 1:
        raise SystemExit("--parity replays the reference's single-process torch.rand stream; it cannot be sharded over "
                         "ranks (every rank would draw the same uniforms) — run it on one GPU")
    mask_ids = [int(i) for i in args.mask_ids.split(",")] if args.mask_ids else None   # 0-based index
    if mask_ids is not None and args.mode == "ddpm" and not (args.encoder_ckpt or args.random_init_encoder):
        raise SystemExit("--mode ddpm --mask_ids builds its prior with the VQ-VAE structure encoder: pass --encoder_ckpt "
                         "<StructureTokenEncoder state dict> (or call ddpm_sample_by_esm(structure_tokens=...) from Python, "
                         "or use the default gibbs mode, which conditions on the coordinates directly)")
    from .model import load_state_dict_from_lightning_ckpt, random_init_model

    targets, coords_of = [], {}
    if args.synthetic_len:
        g = torch.Generator().manual_seed(args.seed)
        ids = torch.randint(4, 24, (args.synthetic_len,), generator=g)
        targets.append((f"synthetic{args.synthetic_len}", "".join(C.SEQUENCE_VOCAB[int(i)] for i in ids)))
    else:
        data_path = Path(args.input)
        assert data_path.is_dir(), f"Invalid directory {data_path} (Currently we only support pdb files in a folder as input)."
        for p in sorted(q for q in data_path.iterdir() if q.suffix == ".pdb"):
            prot = ESMProtein.from_pdb(p)
            targets.append((p.stem, prot.sequence))                        # throw away other entities
            coords_of[p.stem] = prot.coordinates
    max_len = max(len(s) for _, s in targets) + 2
    per_rank = -(-args.num_samples // world)
    # every batch the per-target splitters will issue must fit (ddpm: token count, gibbs: residue count; the remainder
    # batch can exceed the regular one); bounded so that one 288 GB GPU holds the workspace, larger batches are chunked
    max_b = engine_capacity([len(s) for _, s in targets], per_rank, args.n_max_residue_square, args.mode)
    max_b = max(1, min(max_b, per_rank, max(1, (4 * DEFAULT_NMAX) // (max_len * max_len))))
    if args.random_init:
        from .config import ESM3_OPEN, TINY
        model = random_init_model(TINY if args.tiny else ESM3_OPEN, seed=args.seed, max_batch=max_b, max_len=max_len,
                                  device=local_rank, precision=args.precision, head_precision=args.head_precision)
    elif args.ckpt is None:
        from .model import load_stock_esm3
        model = load_stock_esm3(args.esm3_ckpt, device=f"cuda:{local_rank}", max_batch=max_b, max_len=max_len,
                                precision=args.precision, head_precision=args.head_precision)
    else:
        model = load_state_dict_from_lightning_ckpt(args.ckpt, device=f"cuda:{local_rank}", max_batch=max_b,
                                                    max_len=max_len, precision=args.precision, head_precision=args.head_precision)
    decoder = None
    if args.decoder_ckpt or args.random_init_decoder:          # one per rank: each rank decodes its own shard
        from .config import STRUCTURE_DECODER_V0, TINY_DECODER
        from .engine import StructureDecoder
        from .weights import random_init_decoder_state_dict
        dcfg = TINY_DECODER if args.tiny else STRUCTURE_DECODER_V0
        if args.decoder_ckpt:
            dsd = torch.load(args.decoder_ckpt, map_location="cpu", weights_only=True)
        else:
            dsd = random_init_decoder_state_dict(dcfg, seed=args.seed, device=f"cuda:{local_rank}")
        decoder = StructureDecoder(dcfg, dsd, max_batch=64, max_len=max_len, device=local_rank,
                                   precision=args.decoder_precision)
    encoder = None
    if args.encoder_ckpt or args.random_init_encoder:
        from .config import STRUCTURE_ENCODER_V0, TINY_ENCODER
        from .engine import StructureEncoder
        from .weights import random_init_encoder_state_dict
        ecfg = TINY_ENCODER if args.tiny else STRUCTURE_ENCODER_V0
        esd = (torch.load(args.encoder_ckpt, map_location="cpu", weights_only=True) if args.encoder_ckpt
               else random_init_encoder_state_dict(ecfg, seed=args.seed, device=f"cuda:{local_rank}"))
        encoder = StructureEncoder(ecfg, esd, device=local_rank,     # same switch as the decoder (the encoder has no split form)
                                   precision="f32" if args.decoder_precision == "f32_split" else args.decoder_precision)
    lt = getattr(model, "load_timings", None)
    print(f"[rank {rank}] weights ready: load_s = {lt['load_s'] if lt else 'n/a (random init)'}"
          f"{'' if not lt else ' (' + lt['path'] + ')'}, NUMA pinning = {numa}", flush=True)
    if rank == 0:
        print(f">>> Sampling mode = {args.mode}, precision = {args.precision} ...")
    for name, seq in targets:
        if args.mode == "gibbs":
            coordinates = coords_of.get(name) if mask_ids is not None else None   # sample_esmdiff.py:286-289
            if mask_ids is not None and coordinates is None:
                raise SystemExit("--mask_ids needs the coordinates of an input PDB")
            minibatch_gibbs_by_esm(seq, model, Path(args.output), name, num_samples=args.num_samples,
                                   num_steps=args.num_steps, n_max_residue_square=args.n_max_residue_square,
                                   coordinates=coordinates, mask_ids=mask_ids,
                                   seed=args.seed, timestamp=not args.no_timestamp, decoder=decoder)
        else:
            ddpm_sample_by_esm(seq, model, Path(args.output), name, num_samples=args.num_samples,
                               num_steps=args.num_steps, n_max_residue_square=args.n_max_residue_square,
                               coordinates=coords_of.get(name) if mask_ids is not None else None, mask_ids=mask_ids,
                               encoder=encoder,
                               seed=args.seed, noise="torch-cpu" if args.parity else "philox",
                               timestamp=not args.no_timestamp, decoder=decoder)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
