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


def trained_like_state_dict(cfg: ModelConfig, seed: int = 0, prefix: str = "net.", device: str = "cpu", with_geom: bool = False,
                            head_scale: float = 10.0, ln_gain_max: float = 30.0, n_gain_channels: int = 12, n_outlier_channels: int = 4,
                            outlier_magnitude: float = 1e3, ffn_row_scale: float = 50.0, n_ffn_rows: int = 8) -> Dict[str, torch.Tensor]:
    """A synthetic state dict with the statistics of a TRAINED checkpoint instead of a fresh initialisation (VERDICT r04 item 2:
    the released `release_v0.pt` the reference's README runs, /root/reference/README.md:65, cannot be fetched offline, and every
    earlier parity number was taken at PyTorch-default-init scale, logit std 0.6).  Starting from random_init_state_dict:

      * output head: structure_head.3.weight x head_scale -> peaked logits (std ~ 5-6 instead of 0.6);
      * LayerNorm gains: in every LayerNorm of the body (block norms, q / k norms, final norm) n_gain_channels random channels get
        gains log-uniform in [ln_gain_max / 6, ln_gain_max] with random sign kept positive — the heavy-tailed gain vectors trained
        transformers show;
      * outlier residual channels: n_outlier_channels channels carry +-outlier_magnitude in every row of the sequence and
        structure embeddings (the "massive activations" of trained models: a constant, token-independent offset of ~1e3 in a few
        channels, which the first LayerNorms see next to O(1) signal);
      * heavy FFN units: in every block n_ffn_rows hidden units have their gate and up rows (ffn.1) scaled by ffn_row_scale
        (row norm 50x the others: SwiGLU products 2 500x), with the matching ffn.3 columns divided by ffn_row_scale ** 1.5 so that
        the branch output stays finite-scale, as a trained network's would.

    Deterministic in (seed, device).  The result is not a model of anything — it puts the engines' scale bounds (F32_SPLIT's
    power-of-two scalings, f16's range) and the certified sampler's error estimate in the regime a real checkpoint puts them."""
    sd = random_init_state_dict(cfg, seed=seed, prefix=prefix, device=device, with_geom=with_geom)
    g = torch.Generator(device=device).manual_seed(seed + 7919)
    D, FH = cfg.d_model, cfg.ffn_hidden

    def pick(n, hi):
        return torch.randperm(hi, generator=g, device=device)[:n]

    def loguniform(n, lo, hi):
        return torch.exp(torch.rand(n, generator=g, device=device) * (math.log(hi) - math.log(lo)) + math.log(lo))

    sd[prefix + "output_heads.structure_head.3.weight"] = sd[prefix + "output_heads.structure_head.3.weight"] * head_scale
    for name, w in sd.items():
        is_ln = w.dim() == 1 and w.numel() == D and name.endswith(".weight") and (
            ".layernorm_qkv.0." in name or ".q_ln." in name or ".k_ln." in name or ".ffn.0." in name or name.endswith("transformer.norm.weight")
            or ".s_norm." in name)      # (not the head's own LayerNorm: head_scale alone sets the logit scale)
        if is_ln and ln_gain_max > 1 and n_gain_channels > 0:
            w = w.clone()
            w[pick(n_gain_channels, D)] = loguniform(n_gain_channels, ln_gain_max / 6.0, ln_gain_max)
            sd[name] = w
    if n_outlier_channels > 0 and outlier_magnitude > 0:
        ch = pick(n_outlier_channels, D)
        sign = torch.where(torch.rand(n_outlier_channels, generator=g, device=device) < 0.5, -1.0, 1.0)
        for nm in ("sequence_embed.weight", "structure_tokens_embed.weight"):
            w = sd[prefix + "encoder." + nm].clone()
            w[:, ch] = 0.5 * outlier_magnitude * sign + 0.01 * outlier_magnitude * torch.randn(w.shape[0], n_outlier_channels, generator=g, device=device)
            sd[prefix + "encoder." + nm] = w
    if n_ffn_rows > 0 and ffn_row_scale != 1:
        for i in range(cfg.n_layers):
            b = f"{prefix}transformer.blocks.{i}."
            rows = pick(n_ffn_rows, FH)
            up = sd[b + "ffn.1.weight"].clone()
            up[rows] *= ffn_row_scale                      # gate rows (x1 of the chunk)
            up[rows + FH] *= ffn_row_scale                 # the matching up rows (x2)
            sd[b + "ffn.1.weight"] = up
            down = sd[b + "ffn.3.weight"].clone()
            down[:, rows] /= ffn_row_scale ** 1.5
            sd[b + "ffn.3.weight"] = down
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
