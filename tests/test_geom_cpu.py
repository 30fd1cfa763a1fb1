"""Coordinate conditioning (block 0's geometric attention) — CPU checks of the oracle and of the host geometry.

The arithmetic belongs to the un-vendored esm==3.0.4 (PARITY UNPINNED): there is no reference vector
to pin it to.  What is checked here is what the construction must satisfy whatever its details: the conditioned
network is invariant under a global rigid motion of the coordinates, all-unknown coordinates reproduce the
unconditioned network exactly (net.py:433-441: the DDPM path), unknown residues neither contribute nor receive, and
the product's host-side frame builder agrees with the oracle's."""
import math

import pytest
import torch

from esmdiff_amd.config import TINY
from esmdiff_amd.geometry import build_affine3d_from_coordinates
from esmdiff_amd.weights import random_init_state_dict
from oracle import geom_ref
from oracle.esm3_ref import build_from_state_dict


def _backbone(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    ca = torch.cumsum(torch.randn(B, L, 3, generator=g) * 2.2, 1)
    n = ca + torch.randn(B, L, 3, generator=g) * 0.8
    c = ca + torch.randn(B, L, 3, generator=g) * 0.8
    return torch.stack([n, ca, c], 2)


@pytest.fixture(scope="module")
def net():
    sd = random_init_state_dict(TINY, seed=1, with_geom=True)
    return build_from_state_dict(TINY, sd)[0]


def test_frames_are_rotations_and_match_the_oracle():
    xyz = _backbone(2, 17, 0)
    xyz[0, 3] = float("inf")
    xyz[0, 5, 1, 2] = float("nan")
    xyz[1] = float("nan")
    rot, trans, has = build_affine3d_from_coordinates(xyz)
    r2, t2, h2 = geom_ref.build_affine3d_from_coordinates(xyz)
    assert torch.equal(has, h2) and torch.equal(rot, r2) and torch.equal(trans, t2)
    assert has[0].sum() == 15 and not has[1].any()
    eye = torch.eye(3).expand(2, 17, 3, 3)
    assert float((rot.transpose(-1, -2) @ rot - eye).abs().max()) < 1e-5          # orthonormal
    assert float((torch.linalg.det(rot) - 1).abs().max()) < 1e-5                  # right-handed
    assert torch.equal(rot[1], eye[1])                                            # nothing known: identity
    assert torch.equal(trans[0, 3], trans[0, 5])                                  # both take the mean-backbone frame
    # atom37-style input: only the first three atoms are used
    xyz37 = torch.cat([xyz, torch.randn(2, 17, 34, 3)], 2)
    assert torch.equal(build_affine3d_from_coordinates(xyz37)[0], rot)
    with pytest.raises(ValueError):
        build_affine3d_from_coordinates(torch.zeros(17, 3, 3))


def test_conditioned_network_is_rigid_motion_invariant_and_nan_is_unconditioned(net):
    B, L = 2, 20
    g = torch.Generator().manual_seed(4)
    xyz = _backbone(B, L, 1)
    xyz[:, 0] = float("nan")
    xyz[:, -1] = float("nan")
    xyz[:, 6:10] = float("inf")                          # the inpainting driver's marker (sample_esmdiff.py:95)
    seq = torch.randint(4, 24, (B, L), generator=g)
    seq[:, 0], seq[:, -1] = 0, 2
    x = torch.full((B, L), 4096)
    A = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    if torch.det(A) < 0:
        A[:, 0] = -A[:, 0]
    with torch.no_grad():
        y0 = net(structure_tokens=x, sequence_tokens=seq).structure_logits
        y1 = net(structure_tokens=x, sequence_tokens=seq, structure_coords=xyz).structure_logits
        y2 = net(structure_tokens=x, sequence_tokens=seq, structure_coords=xyz @ A.T + torch.tensor([30.0, -7.0, 11.0])).structure_logits
        y3 = net(structure_tokens=x, sequence_tokens=seq, structure_coords=torch.full((B, L, 3, 3), float("nan"))).structure_logits
    assert float((y1 - y0).abs().max()) > 1e-2           # the coordinates do something
    assert float((y2 - y1).abs().max()) < 5e-5           # ... that does not depend on the global pose
    assert torch.equal(y3, y0)                           # no coordinates: exactly the DDPM-path network


def test_frameless_residues_neither_give_nor_receive():
    torch.manual_seed(0)
    ga = geom_ref.GeometricAttentionRef(64, 8)
    with torch.no_grad():
        ga.rotation_scale_per_head.normal_()
        ga.distance_scale_per_head.normal_()
    B, L = 1, 12
    xyz = _backbone(B, L, 2)
    xyz[:, 4] = float("inf")
    s = torch.randn(B, L, 64)
    rot, trans, has = geom_ref.build_affine3d_from_coordinates(xyz)
    with torch.no_grad():
        y, _, pre = ga(s, rot, trans, has, return_parts=True)
        s2 = s.clone()
        s2[:, 4] += 5.0                                  # perturb the frameless residue's features
        y2 = ga(s2, rot, trans, has)
    assert torch.equal(pre[:, 4], torch.zeros(1, 24))    # zeroed before out_proj (mask_and_zero_frameless)
    keep = [i for i in range(L) if i != 4]
    assert float((y2[:, keep] - y[:, keep]).abs().max()) < 1e-6   # nobody attends to it
