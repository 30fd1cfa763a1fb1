"""The SDK names the sampling CLI uses from `esm.sdk.api` (/root/reference/slm/sample_esmdiff.py:13-18),
re-created so that the reference's call sites keep working against this engine: ESMProtein,
ESMProteinTensor, GenerationConfig, and the sequence tokenizer ([ESM-RECALL] vocabulary, constants.py).

`ESMProtein.to_pdb` writes the backbone the structure decoder produced (B-factor column = pLDDT, as esm does); proteins
that came out of a sampler run without a decoder attached carry tokens only and refuse to be written.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from . import constants as C
from .pdbio import read_pdb_backbone, write_backbone_pdb

_TOK = {t: i for i, t in enumerate(C.SEQUENCE_VOCAB)}


def encode_sequence(sequence: str) -> torch.Tensor:
    """'<cls>' + residues + '<eos>' (ids 0 … 2); '_' is the mask residue (slm/models/utils.py:121)."""
    ids = [C.SEQUENCE_BOS_TOKEN]
    for ch in sequence:
        ids.append(C.SEQUENCE_MASK_TOKEN if ch == C.MASK_RESIDUE else _TOK.get(ch, _TOK["X"]))
    ids.append(C.SEQUENCE_EOS_TOKEN)
    return torch.tensor(ids, dtype=torch.int64)


def decode_sequence(tokens) -> str:
    return "".join(C.MASK_RESIDUE if int(t) == C.SEQUENCE_MASK_TOKEN else C.SEQUENCE_VOCAB[int(t)] for t in tokens)


@dataclass
class ESMProtein:
    sequence: Optional[str] = None
    coordinates: Optional[torch.Tensor] = None   # (L, 3 | 37, 3)
    structure_tokens: Optional[torch.Tensor] = None  # (L,) ids without BOS/EOS (this engine's output)
    plddt: Optional[torch.Tensor] = None
    ptm: Optional[torch.Tensor] = None           # scalar, from the decoder's pairwise head (esm: ESMProtein.ptm)

    @classmethod
    def from_pdb(cls, path, chain_id: Optional[str] = None) -> "ESMProtein":
        seq, xyz = read_pdb_backbone(path, chain_id)
        return cls(sequence=seq, coordinates=torch.from_numpy(xyz))

    def to_pdb(self, path) -> None:
        if self.coordinates is None:
            raise ValueError(
                "to_pdb needs coordinates: this protein carries structure TOKENS only — pass decoder=StructureDecoder(...) to "
                "iterative_sampling_raw (or decode with esmdiff_amd.sample_esmdiff.decode_tokens) first")
        seq = (self.sequence or "").replace(C.MASK_RESIDUE, "X")
        write_backbone_pdb(path, seq, np.asarray(self.coordinates)[:, :3, :],
                           None if self.plddt is None else np.asarray(self.plddt))

    def __len__(self):
        return len(self.sequence) if self.sequence is not None else 0


@dataclass
class ESMProteinTensor:
    sequence: Optional[torch.Tensor] = None    # with BOS/EOS
    structure: Optional[torch.Tensor] = None   # with BOS/EOS
    coordinates: Optional[torch.Tensor] = None

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return ESMProteinTensor(mv(self.sequence), mv(self.structure), mv(self.coordinates))


@dataclass
class GenerationConfig:
    track: str = "structure"
    num_steps: int = 16
    temperature: float = 1.0
    top_p: float = 1.0
    schedule: str = "cosine"
    strategy: str = "entropy"
    invalid_ids: Optional[List[int]] = None
    condition_on_coordinates_only: bool = True


def encode_decode(encoder, decoder, pdb):
    """The VQ-VAE round trip of `encode_decode(model, pdb)` (/root/reference/slm/models/utils.py:166-194) with this repository's
    engines: backbone of a PDB file (or of an ESMProtein) -> structure tokens (esmdiff_amd.engine.StructureEncoder) -> decoded
    backbone (StructureDecoder).  Returns (coords, coords_pred), both (L, 3, 3) float32 N / CA / C on the CPU — the reference
    returns the same pair as atom37 arrays; residues without coordinates are tokenised as MASK and decoded like any other token."""
    from pathlib import Path
    if isinstance(pdb, (str, Path)):
        assert Path(pdb).exists(), f"File {pdb} does not exist."
        _, xyz = read_pdb_backbone(pdb)
        coords = torch.from_numpy(np.asarray(xyz, dtype=np.float32))
    elif isinstance(pdb, ESMProtein):
        if pdb.coordinates is None:
            raise ValueError("encode_decode: the protein carries no coordinates")
        coords = torch.as_tensor(pdb.coordinates, dtype=torch.float32)[:, :3]
    else:
        raise ValueError(f"Invalid input type: {type(pdb)}: {pdb}")
    body = encoder.encode(coords[None])[0]                                      # (L,) ids, MASK where a residue has no frame
    tokens = torch.cat([torch.tensor([C.STRUCTURE_BOS_TOKEN], device=body.device), body,
                        torch.tensor([C.STRUCTURE_EOS_TOKEN], device=body.device)])
    pred = decoder.decode(tokens[None])[0]
    return coords.cpu(), pred.cpu()
