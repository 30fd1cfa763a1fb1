"""TEST INFRASTRUCTURE — torch-CPU float32 restatement of the network on the hot path.

What is restated: CustomizedESM3.forward (/root/reference/slm/models/net.py:371-483) with the default
tracks it injects (:410-436), StructureOutputHeads (:298-320), and the pieces of the third-party
`esm==3.0.4` package it calls (EncodeInputs, TransformerStack / UnifiedTransformerBlock /
MultiHeadAttention with QK-LayerNorm + rotary, swiglu_ln_ffn, RegressionHead), which are NOT in the
reference tree and not installable here.

PARITY UNPINNED for the esm parts: restated from the published architecture (SURVEY.md Appendix A,
[ESM-RECALL]); nothing in the reference pins their arithmetic (the reference has no tests).  The in-tree
anchors are the constructor arguments (net.py:325-346: d_model 1536, 24 heads, 48 layers,
mask_and_zero_frameless=True), the head (net.py:301) and the call order of forward.

Module / parameter names follow the reference's state-dict layout so the same
state dict feeds this oracle and the HIP engine.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

SEQ_BOS, SEQ_PAD, SEQ_EOS, SEQ_CHAINBREAK, SEQ_MASK = 0, 1, 2, 31, 32
ST_MASK, ST_EOS, ST_BOS, ST_PAD, ST_CHAINBREAK = 4096, 4097, 4098, 4099, 4100


def rbf(values, v_min, v_max, n_bins=16):
    centers = torch.linspace(v_min, v_max, n_bins, dtype=values.dtype)
    std = (v_max - v_min) / n_bins
    z = (values.unsqueeze(-1) - centers) / std
    return torch.exp(-z ** 2)


class EncodeInputsRef(nn.Module):
    """esm EncodeInputs restricted to what the hot path feeds it (Appendix A.1)."""

    def __init__(self, d_model):
        super().__init__()
        self.sequence_embed = nn.Embedding(64, d_model)
        self.plddt_projection = nn.Linear(16, d_model)
        self.structure_per_res_plddt_projection = nn.Linear(16, d_model)
        self.structure_tokens_embed = nn.Embedding(4096 + 5, d_model)
        self.ss8_embed = nn.Embedding(8 + 3, d_model)
        self.sasa_embed = nn.Embedding(16 + 3, d_model)
        # function_embed / residue_embed: pad id 0 embeds to zero (padding_idx=0) -> no contribution

    def forward(self, sequence_tokens, structure_tokens, average_plddt, per_res_plddt, ss8_tokens, sasa_tokens):
        return (self.sequence_embed(sequence_tokens)
                + self.plddt_projection(rbf(average_plddt, 0.0, 1.0, 16))
                + self.structure_per_res_plddt_projection(rbf(per_res_plddt, 0.0, 1.0, 16))
                + self.structure_tokens_embed(structure_tokens)
                + self.ss8_embed(ss8_tokens)
                + self.sasa_embed(sasa_tokens))


def rotate_half(x):
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


class MultiHeadAttentionRef(nn.Module):
    """Appendix A.3: LN -> Linear(D,3D) -> full-width q/k LayerNorm -> rotary -> SDPA -> Linear(D,D)."""

    def __init__(self, d_model, n_heads):
        super().__init__()
        self.d_model, self.n_heads, self.d_head = d_model, n_heads, d_model // n_heads
        self.layernorm_qkv = nn.Sequential(nn.LayerNorm(d_model), nn.Linear(d_model, 3 * d_model, bias=False))
        self.q_ln = nn.LayerNorm(d_model, bias=False)
        self.k_ln = nn.LayerNorm(d_model, bias=False)
        self.out_proj = nn.Linear(d_model, d_model, bias=False)

    def _rope(self, q, k):
        B, L = q.shape[:2]
        inv_freq = 1.0 / (10000 ** (torch.arange(0, self.d_head, 2, dtype=torch.float32) / self.d_head))
        freqs = torch.outer(torch.arange(L, dtype=torch.float32), inv_freq)
        cos = torch.cat([freqs.cos(), freqs.cos()], -1)[None, :, None, :]
        sin = torch.cat([freqs.sin(), freqs.sin()], -1)[None, :, None, :]
        q = q.view(B, L, self.n_heads, self.d_head)
        k = k.view(B, L, self.n_heads, self.d_head)
        return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin

    def forward(self, x, return_parts=False):
        B, L, _ = x.shape
        qkv = self.layernorm_qkv(x)
        q, k, v = torch.chunk(qkv, 3, dim=-1)
        q, k = self.q_ln(q), self.k_ln(k)
        q, k = self._rope(q, k)
        q, k = q.transpose(1, 2), k.transpose(1, 2)
        v = v.view(B, L, self.n_heads, self.d_head).transpose(1, 2)
        ctx = F.scaled_dot_product_attention(q, k, v)
        ctx = ctx.transpose(1, 2).reshape(B, L, self.d_model)
        out = self.out_proj(ctx)
        return (out, qkv, ctx) if return_parts else out


class SwiGLU(nn.Module):
    def forward(self, x):
        x1, x2 = x.chunk(2, dim=-1)
        return F.silu(x1) * x2


class BlockRef(nn.Module):
    """Appendix A.2: pre-LN, residuals divided by sqrt(n_layers/36); geometric attention (block 0) is
    exactly zero without coordinates (net.py:433-441 + mask_and_zero_frameless) and is omitted."""

    def __init__(self, d_model, n_heads, ffn_hidden, scale, v_heads=0):
        super().__init__()
        self.attn = MultiHeadAttentionRef(d_model, n_heads)
        if v_heads:                                   # block 0 only (Appendix A.2: use_geom_attn = (i == 0))
            from .geom_ref import GeometricAttentionRef
            self.geom_attn = GeometricAttentionRef(d_model, v_heads)
        self.ffn = nn.Sequential(nn.LayerNorm(d_model), nn.Linear(d_model, 2 * ffn_hidden, bias=False), SwiGLU(),
                                 nn.Linear(ffn_hidden, d_model, bias=False))
        self.scale = scale

    def forward(self, x, frames=None):
        x = x + self.attn(x) / self.scale
        if frames is not None and hasattr(self, "geom_attn"):
            x = x + self.geom_attn(x, *frames) / self.scale
        x = x + self.ffn(x) / self.scale
        return x


class TransformerRef(nn.Module):
    def __init__(self, d_model, n_heads, n_layers, ffn_hidden, v_heads=0):
        super().__init__()
        scale = math.sqrt(n_layers / 36)
        self.blocks = nn.ModuleList([BlockRef(d_model, n_heads, ffn_hidden, scale, v_heads if i == 0 else 0)
                                     for i in range(n_layers)])
        self.norm = nn.LayerNorm(d_model, bias=False)

    def forward(self, x, frames=None):
        for b in self.blocks:
            x = b(x, frames)
        return self.norm(x), x


class OutputHeadsRef(nn.Module):
    """StructureOutputHeads (net.py:298-308): RegressionHead = Linear -> GELU -> LayerNorm -> Linear."""

    def __init__(self, d_model, n_out, n_sequence_heads=0):
        super().__init__()
        self.structure_head = nn.Sequential(nn.Linear(d_model, d_model), nn.GELU(), nn.LayerNorm(d_model),
                                            nn.Linear(d_model, n_out))
        if n_sequence_heads:       # net.py:302-303: the optional sequence head is the same RegressionHead
            self.sequence_head = nn.Sequential(nn.Linear(d_model, d_model), nn.GELU(), nn.LayerNorm(d_model),
                                               nn.Linear(d_model, n_sequence_heads))


class ESM3Ref(nn.Module):
    def __init__(self, cfg, with_geom=False, n_sequence_heads=0):
        super().__init__()
        self.cfg = cfg
        self.encoder = EncodeInputsRef(cfg.d_model)
        self.transformer = TransformerRef(cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.ffn_hidden,
                                          cfg.v_heads if with_geom else 0)
        self.output_heads = OutputHeadsRef(cfg.d_model, cfg.n_structure_heads, n_sequence_heads)

    def embed(self, structure_tokens, sequence_tokens, auxiliary_embeddings=None):
        """net.py:410-466 with every optional track at its default."""
        L = structure_tokens.shape[1]
        if sequence_tokens is None:
            sequence_tokens = torch.full((1, L), SEQ_MASK, dtype=torch.long)
        ss8 = torch.zeros(1, L, dtype=torch.long)
        sasa = torch.zeros(1, L, dtype=torch.long)
        average_plddt = torch.ones(1, L)
        per_res_plddt = torch.zeros(1, L)
        st = (structure_tokens.masked_fill(structure_tokens == -1, ST_MASK)
              .masked_fill(sequence_tokens == SEQ_BOS, ST_BOS)
              .masked_fill(sequence_tokens == SEQ_PAD, ST_PAD)
              .masked_fill(sequence_tokens == SEQ_EOS, ST_EOS)
              .masked_fill(sequence_tokens == SEQ_CHAINBREAK, ST_CHAINBREAK))
        x = self.encoder(sequence_tokens, st, average_plddt, per_res_plddt, ss8, sasa)
        if auxiliary_embeddings is not None:
            x = x + auxiliary_embeddings
        return x

    def forward(self, structure_tokens=None, sequence_tokens=None, auxiliary_embeddings=None, labels=None,
                structure_coords=None, **kw):
        x = self.embed(structure_tokens, sequence_tokens, auxiliary_embeddings)
        frames = None
        if structure_coords is not None:              # net.py:433-441; all-NaN coordinates give an exactly zero branch
            from .geom_ref import build_affine3d_from_coordinates
            frames = build_affine3d_from_coordinates(structure_coords)
        x, emb = self.transformer(x, frames)
        seq_head = getattr(self.output_heads, "sequence_head", None)       # net.py:310-311 (a dummy zero tensor without the head)
        return SimpleNamespace(structure_logits=self.output_heads.structure_head(x), embeddings=emb,
                               sequence_logits=None if seq_head is None else seq_head(x))


def build_from_state_dict(cfg, state_dict, prefix="net."):
    """(net, sigma_embedder) loaded strictly from a reference-layout state dict."""
    from .sampler_ref import TimestepEmbedderRef

    sub = {k[len(prefix):]: v.float() for k, v in state_dict.items() if k.startswith(prefix)}
    n_seq = sub["output_heads.sequence_head.3.weight"].shape[0] if "output_heads.sequence_head.3.weight" in sub else 0
    net = ESM3Ref(cfg, with_geom=any("geom_attn" in k for k in sub), n_sequence_heads=n_seq)
    missing, unexpected = net.load_state_dict(sub, strict=False)
    assert not missing, missing
    unexpected = [k for k in unexpected if "geom_attn" not in k and "function_embed" not in k
                  and "residue_embed" not in k]
    assert not unexpected, unexpected
    emb = TimestepEmbedderRef(cfg.d_model, cfg.freq_dim)
    emb.load_state_dict({k[len("sigma_embedder."):]: v.float() for k, v in state_dict.items()
                         if k.startswith("sigma_embedder.")})
    return net.eval(), emb.eval()
