"""TEST INFRASTRUCTURE — torch restatement of esm's StructureTokenEncoder (backbone coordinates -> structure tokens).

Reference call sites: /root/reference/slm/models/utils.py:136-137 (`model.encode(ESMProtein(coordinates=...))` inside
protseq_to_data) reached from /root/reference/slm/sample_esmdiff.py:196-201 (the DDPM inpainting prior).  The module is
esm==3.0.4's esm.models.vqvae.StructureTokenEncoder (un-vendored): [ESM-RECALL] restated from memory, PARITY UNPINNED.
  frames per residue (geom_ref.build_affine3d_from_coordinates) -> 16 nearest residues by CA distance (sequence distance
  for residues without coordinates) -> relative-position embedding of the neighbours' offsets -> 2 blocks of
  [geometric attention (v_heads 128, with bias) + SwiGLU FFN (expansion 4, with bias)] over each 16-residue neighbourhood ->
  the query residue's embedding -> Linear(1024, 128) -> nearest of 4096 codebook vectors.
Checkable without esm: tokens are invariant under rigid motions of the input and do not depend on far-away residues."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .geom_ref import build_affine3d_from_coordinates

MAX_SUPPORTED_DISTANCE = 1e6
MASK_TOKEN = 4096


def knn_edges(ca: torch.Tensor, has: torch.Tensor, knn: int) -> torch.Tensor:
    """(B, L, 3), (B, L) -> (B, L, min(knn, L)) neighbour indices, nearest first (the residue itself comes first)."""
    B, L, _ = ca.shape
    ca = torch.where(has[..., None], ca, torch.zeros_like(ca))
    both = has[:, :, None] & has[:, None, :]
    dists = (ca[:, :, None] - ca[:, None, :]).norm(dim=-1)
    ar = torch.arange(L)
    seq = (ar[:, None] - ar[None, :]).abs().to(dists.dtype) * 1e2 + MAX_SUPPORTED_DISTANCE
    key = torch.where(both, dists, seq[None].expand(B, L, L))
    return key.sort(dim=-1, descending=False, stable=True)[1][..., : min(knn, L)]


class GeomAttnBias(nn.Module):
    """GeometricReasoningOriginalImpl(c_s, v_heads, bias=True, mask_and_zero_frameless=False)."""

    def __init__(self, c_s, v_heads):
        super().__init__()
        self.v_heads = v_heads
        self.s_norm = nn.LayerNorm(c_s, bias=False)
        self.proj = nn.Linear(c_s, v_heads * 15, bias=True)
        self.out_proj = nn.Linear(v_heads * 3, c_s, bias=True)
        self.distance_scale_per_head = nn.Parameter(torch.zeros(v_heads))
        self.rotation_scale_per_head = nn.Parameter(torch.zeros(v_heads))

    def forward(self, s, rot, trans, mask):
        N, K, _ = s.shape
        H = self.v_heads
        p = self.proj(self.s_norm(s))
        vec_rot, vec_dist = p.split([H * 9, H * 6], dim=-1)
        vec_rot = vec_rot.reshape(N, K, 3 * H, 3)
        vec_dist = vec_dist.reshape(N, K, 2 * H, 3)
        rotated = torch.einsum("nkij,nkhj->nkhi", rot, vec_rot)
        q_rot, k_rot, value = rotated.split([H, H, H], dim=2)
        moved = torch.einsum("nkij,nkhj->nkhi", rot, vec_dist) + trans[:, :, None]
        q_dist, k_dist = moved.chunk(2, dim=2)
        q_rot, k_rot, value, q_dist, k_dist = (t.permute(0, 2, 1, 3) for t in (q_rot, k_rot, value, q_dist, k_dist))
        logits = (q_rot @ k_rot.transpose(-1, -2)) / math.sqrt(3) * F.softplus(self.rotation_scale_per_head)[None, :, None, None] \
            - (q_dist[:, :, :, None] - k_dist[:, :, None]).norm(dim=-1) / math.sqrt(3) * F.softplus(self.distance_scale_per_head)[None, :, None, None]
        logits = logits.masked_fill(~mask[:, None, None, :], torch.finfo(logits.dtype).min)
        out = (torch.softmax(logits, dim=-1) @ value).permute(0, 2, 1, 3)
        out = torch.einsum("nkji,nkhj->nkhi", rot, out).reshape(N, K, 3 * H)
        return self.out_proj(out)


class EncoderBlock(nn.Module):
    def __init__(self, d, v_heads, hidden):
        super().__init__()
        self.geom_attn = GeomAttnBias(d, v_heads)
        self.ffn = nn.Sequential(nn.LayerNorm(d), nn.Linear(d, 2 * hidden, bias=True), nn.Identity(), nn.Linear(hidden, d, bias=True))

    def forward(self, x, rot, trans, mask):
        x = x + self.geom_attn(x, rot, trans, mask)
        h = self.ffn[1](self.ffn[0](x))
        a, b = h.chunk(2, dim=-1)
        return x + self.ffn[3](F.silu(a) * b)


class StructureTokenEncoderRef(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        d = cfg.d_model
        self.relative_positional_embedding = nn.Module()
        self.relative_positional_embedding.embedding = nn.Embedding(2 * cfg.relpos_bins + 2, d)
        self.transformer = nn.Module()
        self.transformer.blocks = nn.ModuleList([EncoderBlock(d, cfg.v_heads, cfg.ffn_hidden) for _ in range(cfg.n_layers)])
        self.pre_vq_proj = nn.Linear(d, cfg.d_out)
        self.codebook = nn.Module()
        self.codebook.embeddings = nn.Parameter(torch.zeros(cfg.n_codes, cfg.d_out))

    def forward(self, coords, return_z: bool = False):
        """coords (B, L, 3, 3) -> tokens (B, L) int64, MASK where the residue has no coordinates."""
        cfg = self.cfg
        rot, trans, has = build_affine3d_from_coordinates(coords)
        B, L = has.shape
        edges = knn_edges(coords[..., 1, :].float(), has, cfg.knn)                 # (B, L, K)
        K = edges.shape[-1]
        bi = torch.arange(B)[:, None, None]
        nrot = rot[bi, edges].reshape(B * L, K, 3, 3)
        ntrans = trans[bi, edges].reshape(B * L, K, 3)
        nmask = has[bi, edges].reshape(B * L, K)
        diff = (edges - edges[..., :1]).clamp(-cfg.relpos_bins, cfg.relpos_bins) + cfg.relpos_bins + 1
        z = self.relative_positional_embedding.embedding(diff.reshape(B * L, K))
        for blk in self.transformer.blocks:
            z = blk(z, nrot, ntrans, nmask)
        z = z.reshape(B, L, K, -1)[:, :, 0]
        z = z.masked_fill(~has[..., None], 0)
        z = self.pre_vq_proj(z)
        e = self.codebook.embeddings
        d2 = (z ** 2).sum(-1, keepdim=True) - 2 * z @ e.t() + (e ** 2).sum(-1)[None, None]
        tok = d2.argmin(-1).masked_fill(~has, MASK_TOKEN)
        return (tok, z, d2) if return_z else tok


def build_encoder_from_state_dict(cfg, sd):
    net = StructureTokenEncoderRef(cfg)
    net.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    return net.eval()
