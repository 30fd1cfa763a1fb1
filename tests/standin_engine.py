"""CPU stand-in for esmdiff_amd.engine.Engine — TEST INFRASTRUCTURE, used by tests/ and by `bench.py --stub-engine` only.

It has the call shapes bench.py and the CLI drivers use (ddpm_sample / gibbs_sample / set_profiling / get_profile / close)
and returns ids that are a pure function of (seed, GLOBAL sample index, position) — like the engine's Philox noise — so any
launch, sharding or gather bug shows up as a wrong ensemble.  No arithmetic of the network is restated here: it exists so
that the multi-process launch path (self-spawn under torch.distributed.run, process group, all_gather, max-over-ranks
timing, the single JSON line) runs in CI on a box without GPUs.
"""
from __future__ import annotations

import time

import torch

SECTIONS = ["embed", "layernorm", "gemm_qkv", "qk_norm_rope", "attention", "gemm_out", "gemm_ffn_up", "gemm_ffn_down", "head",
            "sampler", "gemm_ffn_up_union"]


class StandinEngine:
    device = torch.device("cpu")
    has_geom = False

    def __init__(self, cfg=None, state_dict=None, max_batch: int = 2, max_len: int = 1026, device: int = 0, step_seconds: float = 0.01):
        self.cfg, self.max_batch, self.max_len, self.step_seconds = cfg, max_batch, max_len, step_seconds

    @staticmethod
    def ids(n, L, seed, offset):
        b = torch.arange(offset, offset + n)[:, None]
        return (seed * 7 + b * 131 + torch.arange(L)[None] * 17) % 4096

    def ddpm_sample(self, sequence_tokens, schedule, *, seed, sample_offset=0, input_prior=None):
        n, L = sequence_tokens.shape
        assert n <= self.max_batch and L <= self.max_len, "the driver must chunk to the engine capacity"
        time.sleep(self.step_seconds)
        ids = self.ids(n, L, seed, sample_offset)
        return ids if input_prior is None else torch.where(input_prior == 4096, ids, input_prior)

    def gibbs_sample(self, seq, x0, table, temperature, top_p, *, seed, sample_offset=0):
        n, L = x0.shape
        assert n <= self.max_batch
        time.sleep(self.step_seconds)
        return torch.where(x0 == 4096, self.ids(n, L, seed, sample_offset), x0)

    def set_profiling(self, mode):
        pass

    def get_profile(self):
        return {s: {"ms": 0.0, "launches": 0} for s in SECTIONS}

    def set_frames(self, *a, **k):
        raise RuntimeError("stand-in engine: no coordinate conditioning")

    def close(self):
        pass
