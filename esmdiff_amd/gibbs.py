"""`iterative_sampling_raw` for the structure track — the call the reference's "gibbs" mode makes
(/root/reference/slm/sample_esmdiff.py:114-122):

    out_list += iterative_sampling_raw(esm3_model, proteins=prot_list, configs=cfg_list)

[ESM-RECALL] The function belongs to the un-vendored esm==3.0.4 package; semantics per SURVEY.md Appendix B:
encode every protein, batch them, `num_steps = min(num_steps, #masked)`, and at step t unmask the
`still_masked - int(cos((t+1)/T * pi/2) * total + 0.1)` lowest-entropy masked positions with tokens drawn from the
temperature / top-p filtered distribution.  One forward + one fused kernel pair per step, all on the device.
Proteins that carry coordinates condition the model through block 0's geometric attention (frames from
esmdiff_amd.geometry, residues with non-finite coordinates have no frame — the inpainting driver marks masked residues
with Inf, sample_esmdiff.py:88-96); with `condition_on_coordinates_only` (the default) no structure tokens are derived
from them.  esm finishes with `client.decode` per protein (tokens -> coordinates + pLDDT), which is what lets the
reference call `prot.to_pdb(tmp)` on every output (sample_esmdiff.py:124-128): pass `decoder=` (an
esmdiff_amd.engine.StructureDecoder) and the returned proteins carry coordinates and pLDDT as well.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch

from . import constants as C
from .sdk import ESMProtein, GenerationConfig, encode_sequence


def cosine_schedule(t: float) -> float:
    return math.cos(t * math.pi / 2)


# fraction of the positions still masked after a step at progress t in (0, 1]: esm.utils.noise_schedules
# NOISE_SCHEDULE_REGISTRY [ESM-RECALL]; the reference only ever uses the default, "cosine" (sample_esmdiff.py:116-119)
NOISE_SCHEDULES = {
    "cosine": cosine_schedule,
    "linear": lambda t: 1.0 - t,
    "square_root_schedule": lambda t: 1.0 - math.sqrt(t),
    "cubic": lambda t: 1.0 - t ** 3,
    "square": lambda t: 1.0 - t ** 2,
}


def unmask_schedule(total_to_sample: int, num_steps: int, schedule: str = "cosine") -> List[int]:
    if schedule not in NOISE_SCHEDULES:
        raise ValueError(f"unknown schedule {schedule!r}; one of {sorted(NOISE_SCHEDULES)}")
    fn = NOISE_SCHEDULES[schedule]
    T = min(num_steps, total_to_sample)
    out, still = [], total_to_sample
    for t in range(T):
        after = int(fn((t + 1) / T) * total_to_sample + 0.1)
        k = max(still - after, 0)
        out.append(k)
        still -= k
    return out


def encode_structure_prior(protein: ESMProtein, n_tokens: int) -> torch.Tensor:
    """Structure track with BOS/EOS; known tokens where the protein carries them, MASK elsewhere."""
    x = torch.full((n_tokens,), C.STRUCTURE_MASK_TOKEN, dtype=torch.int64)
    x[0], x[-1] = C.STRUCTURE_BOS_TOKEN, C.STRUCTURE_EOS_TOKEN
    if protein.structure_tokens is not None:
        st = torch.as_tensor(protein.structure_tokens, dtype=torch.int64)
        assert st.numel() == n_tokens - 2
        x[1:-1] = st
    return x


@torch.no_grad()
def iterative_sampling_raw(client, proteins: Sequence[ESMProtein], configs: Sequence[GenerationConfig], *,
                           seed: int = 0, sample_offset: int = 0, decoder=None, encoder=None) -> List[ESMProtein]:
    """client: an esmdiff_amd Engine, or an object with a `.net` Engine (the model wrapper); `decoder` / `encoder` default to
    the client's own `.decoder` / `.encoder` attributes when it has them.  `GenerationConfig.condition_on_coordinates_only =
    False` [ESM-RECALL: esm then also feeds the structure TOKENS its VQ-VAE encoder derives from the coordinates]: proteins
    that carry coordinates but no structure tokens get them from `encoder` (an esmdiff_amd.engine.StructureEncoder) for every
    residue with finite coordinates; those positions are then known tokens, not MASK, and are not sampled."""
    eng = getattr(client, "net", client)
    # precision "certified" (esmdiff_amd/certified.py): `net` is the f32-grade engine, `fast` the f16 one that draws every step; the
    # decisions of a step that the f16 logits leave open are verified on `net` — the ids of `net`'s own chain at ~2x its rate
    cert = getattr(client, "certified", None)
    engines = [eng] + ([client.fast] if cert is not None else [])
    if decoder is None:
        decoder = getattr(client, "decoder", None)
    if encoder is None:
        encoder = getattr(client, "encoder", None)
    assert len(proteins) == len(configs) and len(proteins) > 0
    cfg0 = configs[0]
    for c in configs:
        if c.track != "structure":
            raise NotImplementedError("only the structure track is sampled by this engine")
        if (c.num_steps, c.temperature, c.top_p, c.schedule, c.strategy, list(c.invalid_ids or [])) != \
                (cfg0.num_steps, cfg0.temperature, cfg0.top_p, cfg0.schedule, cfg0.strategy, list(cfg0.invalid_ids or [])):
            raise NotImplementedError("one batch shares num_steps / temperature / top_p / schedule / strategy / invalid_ids "
                                      "(the reference passes copies)")
        if c.strategy not in ("entropy", "random"):
            raise ValueError(f"strategy {c.strategy!r}: 'entropy' or 'random'")
    seqs = [encode_sequence(p.sequence) for p in proteins]
    L = seqs[0].numel()
    if any(s.numel() != L for s in seqs):
        raise ValueError("all proteins of a batch must have the same length")
    seq = torch.stack(seqs)
    x0 = torch.stack([encode_structure_prior(p, L) for p in proteins])
    has_xyz = [p.coordinates is not None for p in proteins]
    for b, (p, c) in enumerate(zip(proteins, configs)):
        if not getattr(c, "condition_on_coordinates_only", True) and p.coordinates is not None and p.structure_tokens is None:
            if encoder is None:
                raise RuntimeError("condition_on_coordinates_only=False needs the VQ-VAE structure encoder: pass encoder= "
                                   "(esmdiff_amd.engine.StructureEncoder; CLI: --encoder_ckpt)")
            cb = torch.as_tensor(p.coordinates, dtype=torch.float32)
            if cb.shape[0] != L - 2:
                raise ValueError(f"coordinates cover {cb.shape[0]} residues, sequence has {L - 2}")
            x0[b, 1:-1] = encoder.encode(cb[None]).reshape(-1).to(x0.device)     # MASK where a residue has no coordinates
    frames = None
    if any(has_xyz):
        from .geometry import build_affine3d_from_coordinates
        if not getattr(eng, "has_geom", False):
            raise RuntimeError("proteins carry coordinates but the loaded weights have no transformer.blocks.0.geom_attn.* "
                               "tensors (geometric attention)")
        xyz = torch.full((len(proteins), L, 3, 3), float("nan"))
        for b, p in enumerate(proteins):
            if p.coordinates is not None:
                cb = torch.as_tensor(p.coordinates, dtype=torch.float32)
                if cb.shape[0] != L - 2:
                    raise ValueError(f"coordinates cover {cb.shape[0]} residues, sequence has {L - 2}")
                xyz[b, 1:-1] = cb[:, :3, :]                  # BOS / EOS carry no coordinates
        frames = build_affine3d_from_coordinates(xyz)
        if cert is None:
            eng.set_frames(*frames)
    totals = (x0 == C.STRUCTURE_MASK_TOKEN).sum(1).tolist()
    T = max(min(cfg0.num_steps, t) for t in totals) if max(totals) > 0 else 0
    if T == 0:
        out_x = x0
    else:
        table = torch.zeros(T, len(proteins), dtype=torch.int32)
        for b, tot in enumerate(totals):
            sch = unmask_schedule(tot, cfg0.num_steps, cfg0.schedule)
            table[: len(sch), b] = torch.tensor(sch, dtype=torch.int32)
        custom = cfg0.strategy != "entropy" or bool(cfg0.invalid_ids)
        if custom:
            for e_ in engines:
                e_.set_gibbs_options(cfg0.strategy, cfg0.invalid_ids or ())
        try:
            if cert is not None:     # (more prompts than the engines' max_batch are streamed through the fast lane)
                out_x = cert.gibbs_sample(seq, x0, table, cfg0.temperature, cfg0.top_p, seed=seed, sample_offset=sample_offset,
                                          frames=frames).cpu()
            else:
                out_x = eng.gibbs_sample(seq, x0, table, cfg0.temperature, cfg0.top_p, seed=seed,
                                         sample_offset=sample_offset).cpu()
        finally:
            if custom:
                for e_ in engines:
                    e_.set_gibbs_options()
    if any(has_xyz) and cert is None:
        eng.set_frames(None)
    coords = plddt = ptm = None
    if decoder is not None:                        # esm: client.decode(tensor) -> ESMProtein with coordinates, pLDDT, pTM
        from .sample_esmdiff import decode_tokens
        coords, plddt, ptm = decode_tokens(out_x[:, 1:-1], decoder, return_ptm=True)
        coords, plddt = coords.cpu(), (None if plddt is None else plddt.cpu())
        ptm = None if ptm is None else ptm.cpu()
    return [ESMProtein(sequence=p.sequence, coordinates=None if coords is None else coords[b],
                       structure_tokens=out_x[b, 1:-1].clone(), plddt=None if plddt is None else plddt[b],
                       ptm=None if ptm is None else ptm[b])
            for b, p in enumerate(proteins)]
