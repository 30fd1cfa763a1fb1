"""Weights: random initialisation with the reference's state-dict key layout, and checkpoint loading.

Key layout: the ESMDiff checkpoint stores a DeepSpeed 'module' dict
(/root/reference/slm/utils/checkpoint_utils.py:62-64) whose keys are `net.<esm3 key>` plus
`sigma_embedder.mlp.{0,2}.{weight,bias}` (net.py:489-492).
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Dict

import torch

from .config import ModelConfig


def random_init_state_dict(cfg: ModelConfig, seed: int = 0, prefix: str = "net.",
                           device: str = "cpu", with_geom: bool = False) -> Dict[str, torch.Tensor]:
    """ESM3-architecture random weights (PyTorch default initialisers per layer type), float32.

    `device` selects where the numbers are drawn (the CPU and GPU generators give different streams; tests
    use "cpu", the full-size benchmark draws its 1.4 B parameters on the GPU to save a minute of host time).
    LayerNorm weights are drawn around 1 (not exactly 1) and biases around 0 so that tests exercise them."""
    g = torch.Generator(device=device).manual_seed(seed)
    D, FH, V = cfg.d_model, cfg.ffn_hidden, cfg.n_structure_heads
    sd: Dict[str, torch.Tensor] = {}

    class _R:
        @staticmethod
        def randn(*shape, generator=None):
            return torch.randn(*shape, generator=g, device=device)

    def _linear(_g, out_f, in_f, bias):
        # nn.Linear default init: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in)) for weight and bias
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g, device=device) * 2 - 1) * bound
        b = (torch.rand(out_f, generator=g, device=device) * 2 - 1) * bound if bias else None
        return w, b

    def ln(name, bias=True):
        sd[name + ".weight"] = 1.0 + 0.1 * _R.randn(D)
        if bias:
            sd[name + ".bias"] = 0.05 * _R.randn(D)

    e = prefix + "encoder."
    sd[e + "sequence_embed.weight"] = _R.randn(64, D)
    sd[e + "structure_tokens_embed.weight"] = _R.randn(4096 + 5, D)
    sd[e + "ss8_embed.weight"] = _R.randn(8 + 3, D)
    sd[e + "sasa_embed.weight"] = _R.randn(16 + 3, D)
    for nm in ("plddt_projection", "structure_per_res_plddt_projection"):
        w, b = _linear(g, D, 16, True)
        sd[e + nm + ".weight"], sd[e + nm + ".bias"] = w, b
    for i in range(cfg.n_layers):
        b = f"{prefix}transformer.blocks.{i}."
        ln(b + "attn.layernorm_qkv.0")
        sd[b + "attn.layernorm_qkv.1.weight"] = _linear(g, 3 * D, D, False)[0]
        sd[b + "attn.q_ln.weight"] = 1.0 + 0.1 * _R.randn(D)
        sd[b + "attn.k_ln.weight"] = 1.0 + 0.1 * _R.randn(D)
        sd[b + "attn.out_proj.weight"] = _linear(g, D, D, False)[0]
        ln(b + "ffn.0")
        sd[b + "ffn.1.weight"] = _linear(g, 2 * FH, D, False)[0]
        sd[b + "ffn.3.weight"] = _linear(g, D, FH, False)[0]
    sd[prefix + "transformer.norm.weight"] = 1.0 + 0.1 * _R.randn(D)
    h = prefix + "output_heads.structure_head."
    sd[h + "0.weight"], sd[h + "0.bias"] = _linear(g, D, D, True)
    ln(h + "2")
    sd[h + "3.weight"], sd[h + "3.bias"] = _linear(g, V, D, True)
    s = "sigma_embedder.mlp."
    sd[s + "0.weight"], sd[s + "0.bias"] = _linear(g, D, cfg.freq_dim, True)
    sd[s + "2.weight"], sd[s + "2.bias"] = _linear(g, D, D, True)
    if with_geom:   # block 0's geometric attention; drawn last so that the other tensors keep their stream
        ga, VH = f"{prefix}transformer.blocks.0.geom_attn.", cfg.v_heads
        sd[ga + "s_norm.weight"] = 1.0 + 0.1 * _R.randn(D)
        sd[ga + "proj.weight"] = _linear(g, VH * 15, D, False)[0]
        sd[ga + "out_proj.weight"] = _linear(g, D, VH * 3, False)[0]
        sd[ga + "distance_scale_per_head"] = 0.5 * _R.randn(VH)
        sd[ga + "rotation_scale_per_head"] = 0.5 * _R.randn(VH)
    return sd


def random_init_decoder_state_dict(cfg, seed: int = 0, device: str = "cpu", with_plddt: bool = True,
                                   with_pairwise: bool = True) -> Dict[str, torch.Tensor]:
    """Random weights with the key layout of esm's StructureTokenDecoder: embed, decoder_stack.*,
    affine_output_projection.* — for tests and offline plumbing (the real esm3_structure_decoder_v0 cannot be fetched)."""
    g = torch.Generator(device=device).manual_seed(seed)
    D, FH = cfg.d_model, cfg.ffn_hidden

    def randn(*shape):
        return torch.randn(*shape, generator=g, device=device)

    def linear(out_f, in_f, bias):
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g, device=device) * 2 - 1) * bound
        return w, ((torch.rand(out_f, generator=g, device=device) * 2 - 1) * bound if bias else None)

    sd: Dict[str, torch.Tensor] = {"embed.weight": randn(4096 + 5, D)}
    for i in range(cfg.n_layers):
        b = f"decoder_stack.blocks.{i}."
        sd[b + "attn.layernorm_qkv.0.weight"] = 1.0 + 0.1 * randn(D)
        sd[b + "attn.layernorm_qkv.0.bias"] = 0.05 * randn(D)
        sd[b + "attn.layernorm_qkv.1.weight"] = linear(3 * D, D, False)[0]
        sd[b + "attn.q_ln.weight"] = 1.0 + 0.1 * randn(D)
        sd[b + "attn.k_ln.weight"] = 1.0 + 0.1 * randn(D)
        sd[b + "attn.out_proj.weight"] = linear(D, D, False)[0]
        sd[b + "ffn.0.weight"] = 1.0 + 0.1 * randn(D)
        sd[b + "ffn.0.bias"] = 0.05 * randn(D)
        sd[b + "ffn.1.weight"] = linear(2 * FH, D, False)[0]
        sd[b + "ffn.3.weight"] = linear(D, FH, False)[0]
    sd["decoder_stack.norm.weight"] = 1.0 + 0.1 * randn(D)
    h = "affine_output_projection."
    sd[h + "ffn1.weight"], sd[h + "ffn1.bias"] = linear(D, D, True)
    sd[h + "norm.weight"], sd[h + "norm.bias"] = 1.0 + 0.1 * randn(D), 0.05 * randn(D)
    sd[h + "proj.weight"], sd[h + "proj.bias"] = linear(23, D, True)
    if with_plddt:       # esm RegressionHead(d, 50): drawn last so that the other tensors keep their stream
        sd["plddt_head.0.weight"], sd["plddt_head.0.bias"] = linear(D, D, True)
        sd["plddt_head.2.weight"], sd["plddt_head.2.bias"] = 1.0 + 0.1 * randn(D), 0.05 * randn(D)
        sd["plddt_head.3.weight"], sd["plddt_head.3.bias"] = linear(50, D, True)
    if with_pairwise:    # esm PairwisePredictionHead(d, 128, 128, 64 + 96 + 64, bias=False): the pTM / PAE head
        p = "pairwise_classification_head."
        sd[p + "downproject.weight"] = linear(128, D, False)[0]
        sd[p + "linear1.weight"] = linear(128, 128, False)[0] * 4.0      # spread the PAE logits (a flat softmax tests nothing)
        sd[p + "norm.weight"], sd[p + "norm.bias"] = 1.0 + 0.1 * randn(128), 0.05 * randn(128)
        sd[p + "linear2.weight"] = linear(224, 128, False)[0] * 4.0
    return sd


def random_init_encoder_state_dict(cfg, seed: int = 0, device: str = "cpu") -> Dict[str, torch.Tensor]:
    """Random weights with the key layout of esm's StructureTokenEncoder — tests / offline plumbing."""
    g = torch.Generator(device=device).manual_seed(seed)
    D, FH, VH = cfg.d_model, cfg.ffn_hidden, cfg.v_heads

    def randn(*shape):
        return torch.randn(*shape, generator=g, device=device)

    def linear(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        return ((torch.rand(out_f, in_f, generator=g, device=device) * 2 - 1) * bound,
                (torch.rand(out_f, generator=g, device=device) * 2 - 1) * bound)

    sd: Dict[str, torch.Tensor] = {"relative_positional_embedding.embedding.weight": randn(2 * cfg.relpos_bins + 2, D)}
    for i in range(cfg.n_layers):
        b = f"transformer.blocks.{i}."
        sd[b + "geom_attn.s_norm.weight"] = 1.0 + 0.1 * randn(D)
        sd[b + "geom_attn.proj.weight"], sd[b + "geom_attn.proj.bias"] = linear(15 * VH, D)
        sd[b + "geom_attn.out_proj.weight"], sd[b + "geom_attn.out_proj.bias"] = linear(D, 3 * VH)
        sd[b + "geom_attn.out_proj.weight"] *= 8.0     # let the geometry dominate the (residue-independent) offset
        sd[b + "geom_attn.distance_scale_per_head"] = 0.5 * randn(VH)
        sd[b + "geom_attn.rotation_scale_per_head"] = 0.5 * randn(VH)
        sd[b + "ffn.0.weight"], sd[b + "ffn.0.bias"] = 1.0 + 0.1 * randn(D), 0.05 * randn(D)
        sd[b + "ffn.1.weight"], sd[b + "ffn.1.bias"] = linear(2 * FH, D)
        sd[b + "ffn.3.weight"], sd[b + "ffn.3.bias"] = linear(D, FH)
    sd["pre_vq_proj.weight"], sd["pre_vq_proj.bias"] = linear(cfg.d_out, D)
    sd["codebook.embeddings"] = randn(cfg.n_codes, cfg.d_out) * 0.3
    return sd


def _torch_load_tensors(path):
    """torch.load restricted to tensors and plain containers.  DeepSpeed / Lightning `mp_rank_00_model_states.pt` files
    carry client state next to 'module' (omegaconf nodes, functools.partial, ds_config ...) that the restricted
    unpickler refuses; say what to do instead of failing with a bare UnpicklingError."""
    import pickle
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as ex:
        raise RuntimeError(
            f"{path} holds pickled Python objects besides tensors ({str(ex).splitlines()[0]}); this loader only unpickles "
            "tensors (weights_only=True).  Re-export the weights once where the checkpoint is trusted:\n"
            "    sd = torch.load(path, map_location='cpu', weights_only=False)['module']\n"
            "    torch.save({'module': sd}, 'release_v0.module.pt')") from ex


def checkpoint_file_and_config(path):
    """checkpoint_utils.py:43-50: (file to load, the run's .hydra/config.yaml next to it or None)."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    if path.suffix not in (".ckpt", ".pt"):
        raise ValueError(f"Unsupported ckpt format: {path}")
    if path.is_dir():
        file = path / "checkpoint/mp_rank_00_model_states.pt"
        cfg = file.parent.parent.parent.parent / ".hydra/config.yaml"
    else:
        file, cfg = path, path.parent.parent / ".hydra/config.yaml"
    if file.suffix != ".pt":
        raise ValueError(f"Unsupported ckpt format: {file}")       # checkpoint_utils.py:65-66
    return file, (cfg if cfg.exists() else None)


def load_checkpoint_state_dict(path) -> Dict[str, torch.Tensor]:
    """The reference's format (checkpoint_utils.py:41-64): a .pt whose 'module' entry is the state dict with the
    `net.*` and `sigma_embedder.*` keys of MaskedDiffusionLanguageModeling."""
    file, _ = checkpoint_file_and_config(path)
    blob = _torch_load_tensors(file)
    if not isinstance(blob, dict) or "module" not in blob:
        raise KeyError(f"{file}: no 'module' entry (the reference reads torch.load(...)['module'], checkpoint_utils.py:63)")
    sd = blob["module"]
    if not any(k.startswith("net.") for k in sd):
        raise KeyError(f"{file}: 'module' holds no net.* keys — not an ESMDiff task-module state dict")
    return sd
