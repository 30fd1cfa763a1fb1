"""TEST INFRASTRUCTURE — torch restatement of esm's StructureTokenDecoder (tokens -> backbone coordinates).

Reference call site: /root/reference/slm/sample_esmdiff.py:40-61 (`esm3.decode(ESMProteinTensor(structure=ids))` ->
ESMProtein -> to_pdb), looped per sample at :225-230.  The module lives in the un-vendored esm==3.0.4
(esm.models.vqvae.StructureTokenDecoder, esm.layers.structure_proj.Dim6RotStructureHead): [ESM-RECALL] restated from
memory, PARITY UNPINNED.  Restated: the backbone (N, CA, C) output and the pLDDT head (RegressionHead(d, 50)
on the same hidden state; value = mean of the categorical mixture over 50 bins of [0, 1], which ESMProtein.to_pdb writes
into the B-factor column).  The pairwise pTM / PAE head is not.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .esm3_ref import TransformerRef
from .geom_ref import graham_schmidt

BB_COORDINATES = [[0.5256, 1.3612, 0.0000], [0.0000, 0.0000, 0.0000], [-1.5251, 0.0000, 0.0000]]  # N, CA, C


class Dim6RotStructureHeadRef(nn.Module):
    def __init__(self, d: int, trans_scale: float = 10.0):
        super().__init__()
        self.ffn1 = nn.Linear(d, d)
        self.norm = nn.LayerNorm(d)
        self.proj = nn.Linear(d, 9 + 7 * 2)
        self.trans_scale = trans_scale

    def forward(self, x):
        x = self.norm(torch.nn.functional.gelu(self.ffn1(x)))
        trans, ax, ay, _ = self.proj(x).split([3, 3, 3, 14], dim=-1)
        trans = trans * self.trans_scale
        ax = ax / (ax.norm(dim=-1, keepdim=True) + 1e-5)
        ay = ay / (ay.norm(dim=-1, keepdim=True) + 1e-5)
        # Affine3D.from_graham_schmidt(neg_x_axis = ax + trans, origin = trans, xy_plane = ay + trans)
        rot = graham_schmidt(trans - (ax + trans), (ay + trans) - trans)
        bb = torch.tensor(BB_COORDINATES, dtype=x.dtype)
        return torch.einsum("...ij,aj->...ai", rot, bb) + trans[..., None, :]


class StructureTokenDecoderRef(nn.Module):
    def __init__(self, cfg, with_plddt: bool = False):
        super().__init__()
        self.embed = nn.Embedding(4096 + 5, cfg.d_model)
        self.decoder_stack = TransformerRef(cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.ffn_hidden)
        for b in self.decoder_stack.blocks:       # scale_residue=False
            b.scale = 1.0
        self.affine_output_projection = Dim6RotStructureHeadRef(cfg.d_model, cfg.trans_scale)
        if with_plddt:
            d = cfg.d_model
            self.plddt_head = nn.Sequential(nn.Linear(d, d), nn.GELU(), nn.LayerNorm(d), nn.Linear(d, 50))

    def forward(self, structure_tokens, return_plddt: bool = False):
        x, _ = self.decoder_stack(self.embed(structure_tokens))
        bb = self.affine_output_projection(x)[:, 1:-1]        # drop BOS / EOS
        if not return_plddt:
            return bb
        logits = self.plddt_head(x)
        n = logits.shape[-1]
        centers = (torch.arange(n, dtype=logits.dtype) + 0.5) / n      # CategoricalMixture(bins=50, start=0, end=1).mean()
        return bb, (logits.softmax(-1) * centers).sum(-1)[:, 1:-1]


def build_decoder_from_state_dict(cfg, state_dict):
    net = StructureTokenDecoderRef(cfg, with_plddt="plddt_head.3.weight" in state_dict)
    net.load_state_dict({k: v.float() for k, v in state_dict.items()}, strict=True)
    return net.eval()
