"""TEST INFRASTRUCTURE — torch restatement of esm's StructureTokenDecoder (tokens -> backbone coordinates).

Reference call site: /root/reference/slm/sample_esmdiff.py:40-61 (`esm3.decode(ESMProteinTensor(structure=ids))` ->
ESMProtein -> to_pdb), looped per sample at :225-230.  The module lives in the un-vendored esm==3.0.4
(esm.models.vqvae.StructureTokenDecoder, esm.layers.structure_proj.Dim6RotStructureHead): [ESM-RECALL] restated from
memory, PARITY UNPINNED.  Restated: the backbone (N, CA, C) output and the pLDDT head (RegressionHead(d, 50)
on the same hidden state; value = mean of the categorical mixture over 50 bins of [0, 1], which ESMProtein.to_pdb writes
into the B-factor column), and the pairwise head's predicted-aligned-error slice with the pTM / PAE values computed
from it (PairwisePredictionHead(d, 128, 128, 64 + 96 + 64 bins, bias=False); compute_tm / compute_predicted_aligned_error
of esm.utils.structure.predicted_aligned_error, max_bin = 31) — what `decoder_output["ptm"]` is at
/root/reference/slm/models/utils.py:73-76.  The distogram / direction slices of the pairwise head are training targets
only and are not restated.
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


PAIRWISE_BINS = (64, 96, 64)     # distogram, 16 direction bins x 6, predicted aligned error
MAX_PAE_BIN = 31.0


class PairwisePredictionHeadRef(nn.Module):
    """esm.layers.regression_head / structure heads: downproject to 128, split into q | k (64 each), pair features
    [q_j * k_i | q_j - k_i] for the pair (i, j), Linear -> GELU -> LayerNorm -> Linear, all without bias."""

    def __init__(self, d: int, downproject_dim: int = 128, hidden_dim: int = 128, n_bins: int = sum(PAIRWISE_BINS)):
        super().__init__()
        self.downproject = nn.Linear(d, downproject_dim, bias=False)
        self.linear1 = nn.Linear(downproject_dim, hidden_dim, bias=False)
        self.norm = nn.LayerNorm(hidden_dim)
        self.linear2 = nn.Linear(hidden_dim, n_bins, bias=False)

    def forward(self, x):
        q, k = self.downproject(x).chunk(2, dim=-1)
        prod = q[:, None, :, :] * k[:, :, None, :]
        diff = q[:, None, :, :] - k[:, :, None, :]
        h = self.norm(torch.nn.functional.gelu(self.linear1(torch.cat([prod, diff], dim=-1))))
        return self.linear2(h)                                 # [B, L, L, n_bins]


def pae_bins(max_bin: float = MAX_PAE_BIN, num_bins: int = 64) -> torch.Tensor:
    bins = torch.linspace(0, max_bin, steps=num_bins - 1)
    step = max_bin / (num_bins - 2)
    centers = bins + step / 2
    return torch.cat([centers, (centers[-1] + step)[None]])


def _pae_probs(logits, aa_mask):
    square = (aa_mask[:, :, None] & aa_mask[:, None, :])
    probs = logits.float().masked_fill(~square[..., None], torch.finfo(torch.float32).min).softmax(-1)
    return probs, square


def compute_predicted_aligned_error(logits, aa_mask, max_bin: float = MAX_PAE_BIN):
    probs, _ = _pae_probs(logits, aa_mask)
    return (probs * pae_bins(max_bin, logits.shape[-1])).sum(-1)


def compute_tm(logits, aa_mask, max_bin: float = MAX_PAE_BIN):
    probs, square = _pae_probs(logits, aa_mask)
    seqlens = aa_mask.sum(-1, keepdim=True)
    d0 = 1.24 * (seqlens.clamp_min(19) - 15).float() ** (1.0 / 3.0) - 1.8           # [B, 1]
    f_d = 1.0 / (1.0 + (pae_bins(max_bin, logits.shape[-1])[None] / d0) ** 2)        # [B, bins]
    tm = (probs * f_d[:, None, None, :]).sum(-1)                                     # [B, L, L]
    tm = (square * tm).sum(-1) / (1e-10 + square.sum(-1))                            # masked mean over j
    return tm.max(-1).values


class StructureTokenDecoderRef(nn.Module):
    def __init__(self, cfg, with_plddt: bool = False, with_pairwise: bool = False):
        super().__init__()
        self.embed = nn.Embedding(4096 + 5, cfg.d_model)
        self.decoder_stack = TransformerRef(cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.ffn_hidden)
        for b in self.decoder_stack.blocks:       # scale_residue=False
            b.scale = 1.0
        self.affine_output_projection = Dim6RotStructureHeadRef(cfg.d_model, cfg.trans_scale)
        if with_plddt:
            d = cfg.d_model
            self.plddt_head = nn.Sequential(nn.Linear(d, d), nn.GELU(), nn.LayerNorm(d), nn.Linear(d, 50))
        if with_pairwise:
            self.pairwise_classification_head = PairwisePredictionHeadRef(cfg.d_model)

    def confidence(self, structure_tokens):
        """-> (pTM [B], predicted aligned error [B, L, L]) with BOS / EOS / special tokens excluded from the mask."""
        x, _ = self.decoder_stack(self.embed(structure_tokens))
        pae_logits = self.pairwise_classification_head(x)[..., PAIRWISE_BINS[0] + PAIRWISE_BINS[1]:]
        aa_mask = structure_tokens < 4096
        return compute_tm(pae_logits, aa_mask), compute_predicted_aligned_error(pae_logits, aa_mask)

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
    net = StructureTokenDecoderRef(cfg, with_plddt="plddt_head.3.weight" in state_dict,
                                   with_pairwise="pairwise_classification_head.linear2.weight" in state_dict)
    net.load_state_dict({k: v.float() for k, v in state_dict.items()}, strict=True)
    return net.eval()
