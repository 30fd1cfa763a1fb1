"""Backbone frames for the coordinate-conditioned path (host side, torch on CPU or GPU).

The reference builds them inside CustomizedESM3.forward: `structure_coords[..., :3, :]` ->
esm's `build_affine3d_from_coordinates` (/root/reference/slm/models/net.py:433-441), and the inpainting driver marks
unknown residues with Inf coordinates (/root/reference/slm/sample_esmdiff.py:88-96).  esm==3.0.4 is not vendored:
the construction below follows SURVEY.md A.4 [ESM-RECALL] (Gram-Schmidt frame with origin CA, first axis CA - C,
second axis in the plane of N; residues without finite coordinates get the frame of the average known backbone, or the
identity when nothing is known) and is cross-checked against oracle/geom_ref.py in tests/test_geom_cpu.py.
The engine consumes the result through esmdiff_set_frames (include/esmdiff_hip.h).
"""
from __future__ import annotations

from typing import Tuple

import torch

_MAX_SUPPORTED_DISTANCE = 1e6


def _normalise(v: torch.Tensor, eps: float) -> torch.Tensor:
    return v / torch.sqrt((v * v).sum(-1, keepdim=True) + eps)


def _frames(n: torch.Tensor, ca: torch.Tensor, c: torch.Tensor, eps: float = 1e-12) -> Tuple[torch.Tensor, torch.Tensor]:
    e0 = _normalise(ca - c, eps)
    v = n - ca
    e1 = _normalise(v - e0 * (e0 * v).sum(-1, keepdim=True), eps)
    e2 = torch.cross(e0, e1, dim=-1)
    return torch.stack([e0, e1, e2], dim=-1), ca            # columns of R are the axes


def build_affine3d_from_coordinates(coords: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """coords (B, L, >=3, 3) with atoms N, CA, C first (atom37 / atom14 layouts are fine) ->
    rot (B, L, 3, 3), trans (B, L, 3), has_frame (B, L) bool."""
    if coords.dim() != 4 or coords.shape[-1] != 3 or coords.shape[-2] < 3:
        raise ValueError(f"coordinates must be (B, L, >=3, 3), got {tuple(coords.shape)}")
    xyz = coords[..., :3, :].to(torch.float32)
    has = (torch.isfinite(xyz) & (xyz < _MAX_SUPPORTED_DISTANCE)).all(-1).all(-1)
    xyz = torch.where(has[..., None, None], xyz, torch.zeros_like(xyz))
    B, L = has.shape
    mean = xyz.sum(1) / (has.sum(-1)[..., None, None].to(xyz.dtype) + 1e-8)
    rot_bh, trans_bh = _frames(mean[..., 0, :], mean[..., 1, :], mean[..., 2, :])
    eye = torch.eye(3, dtype=xyz.dtype, device=xyz.device)
    rot_bh = torch.where(has.any(-1)[:, None, None], rot_bh, eye.expand(B, 3, 3))
    rot, trans = _frames(xyz[..., 0, :], xyz[..., 1, :], xyz[..., 2, :])
    rot = torch.where(has[..., None, None], rot, rot_bh[:, None].expand(B, L, 3, 3))
    trans = torch.where(has[..., None], trans, trans_bh[:, None].expand(B, L, 3))
    return rot.contiguous(), trans.contiguous(), has
