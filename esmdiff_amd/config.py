"""Static model configuration — the inference-time content of the reference's Hydra config
(/root/reference/configs/experiment/mdlm.yaml:26-58) and of CustomizedESM3's defaults
(/root/reference/slm/models/net.py:322-332)."""
from __future__ import annotations

import math
from dataclasses import dataclass


@dataclass(frozen=True)
class ModelConfig:
    d_model: int = 1536            # net.py:325
    n_heads: int = 24              # net.py:326
    v_heads: int = 256             # net.py:327 (block 0 geometric attention; live only with coordinates)
    n_layers: int = 48             # net.py:328
    n_structure_heads: int = 4101  # mdlm.yaml:57
    freq_dim: int = 256            # TimestepEmbedder.frequency_embedding_size, net.py:487
    time_conditioning: bool = True # mdlm.yaml:41
    noise_eps: float = 1e-3        # LogLinearNoise(eps), noise_utils.py:196
    noise_removal: bool = True     # checkpoint_utils.py:71

    @property
    def ffn_hidden(self) -> int:   # esm swiglu_ln_ffn: ceil(8/3 d / 256) * 256
        return int(((8.0 / 3.0 * self.d_model) + 255) // 256 * 256)

    @property
    def residue_scale(self) -> float:  # esm TransformerStack: sqrt(n_layers / 36)
        return math.sqrt(self.n_layers / 36)


ESM3_OPEN = ModelConfig()
# the stock esm3_sm_open_v1 network the reference samples from when no --ckpt is given (sample_esmdiff.py:37, :252-255):
# 4096-way structure head, no time conditioning
ESM3_OPEN_STOCK = ModelConfig(n_structure_heads=4096, time_conditioning=False)
# small configuration with the same structure, for tests (d_model must be a multiple of 512)
TINY = ModelConfig(d_model=512, n_heads=8, v_heads=128, n_layers=2)  # v_heads: the engine needs a multiple of 128


@dataclass(frozen=True)
class DecoderConfig:
    """esm StructureTokenDecoder(d_model=1280, n_heads=20, n_layers=30) as ESM3.decode reaches it
    (/root/reference/slm/sample_esmdiff.py:56-59) [ESM-RECALL, SURVEY.md 8f-1]: same blocks as ESM3, no residue scaling."""
    d_model: int = 1280
    n_heads: int = 20
    n_layers: int = 30
    trans_scale: float = 10.0      # Dim6RotStructureHead(trans_scale_factor=10)

    @property
    def ffn_hidden(self) -> int:
        return int(((8.0 / 3.0 * self.d_model) + 255) // 256 * 256)


STRUCTURE_DECODER_V0 = DecoderConfig()
TINY_DECODER = DecoderConfig(d_model=768, n_heads=12, n_layers=2)   # 768: exercises the half-slab q/k LayerNorm path


@dataclass(frozen=True)
class EncoderConfig:
    """esm StructureTokenEncoder(d_model=1024, n_heads=1, v_heads=128, n_layers=2, d_out=128, n_codes=4096) as ESM3.encode
    reaches it (/root/reference/slm/models/utils.py:136-137) [ESM-RECALL, SURVEY.md 8f-4]."""
    d_model: int = 1024
    v_heads: int = 128
    n_layers: int = 2
    d_out: int = 128
    n_codes: int = 4096
    knn: int = 16
    relpos_bins: int = 32

    @property
    def ffn_hidden(self) -> int:       # swiglu_ln_ffn(d, expansion_ratio=4)
        return int(((4.0 * self.d_model) + 255) // 256 * 256)


STRUCTURE_ENCODER_V0 = EncoderConfig()
TINY_ENCODER = EncoderConfig(d_model=512, n_layers=2)
