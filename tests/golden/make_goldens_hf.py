"""Golden vectors for two conventions of the un-vendored `esm` package, taken from INDEPENDENT public code that ships in the
build image (transformers 5.15; nothing here imports /root/reference or esm):

  * rotary position embedding, rotate-half (NeoX) form: transformers.models.esm.modeling_esm.{rotate_half,
    apply_rotary_pos_emb, EsmRotaryEmbedding} — the ESM family's own rotary in Hugging Face's port (inv_freq =
    base^(-2i/d), cos / sin tables cat(freqs, freqs), x cos + rotate_half(x) sin);
  * Gram-Schmidt backbone frames: transformers.models.esm.openfold_utils.rigid_utils.Rigid.from_3_points (AlphaFold-2
    algorithm 21) with (p_neg_x_axis, origin, p_xy_plane) = (C, CA, N) — the argument order esm's
    build_affine3d_from_coordinates uses (Affine3D.from_graham_schmidt(C, CA, N)) [ESM-RECALL for that call; the
    construction itself — which axis is first, the handedness of the third, columns-are-axes — is what this pins].

This does NOT pin esm's arithmetic (parity stays "unpinned" for everything esm owns); it removes the two
conventions most likely to be mis-remembered.

    python tests/golden/make_goldens_hf.py      ->  tests/golden/g11_hf_conventions.npz
"""
from pathlib import Path

import numpy as np
import torch
from transformers.models.esm.modeling_esm import apply_rotary_pos_emb, rotate_half
from transformers.models.esm.openfold_utils.rigid_utils import Rigid

OUT = Path(__file__).resolve().parent / "g11_hf_conventions.npz"


def main():
    g = torch.Generator().manual_seed(11)
    # rotary: (B, H, L, d) with d = 64 as in ESM3's heads, positions 0..L-1, base 10000
    B, H, L, d = 2, 3, 9, 64
    q = torch.randn(B, H, L, d, generator=g)
    k = torch.randn(B, H, L, d, generator=g)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float) / d))
    freqs = (inv_freq[None, :, None] @ torch.arange(L, dtype=torch.float)[None, None, :]).transpose(1, 2)   # EsmRotaryEmbedding.forward
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos(), emb.sin()                                                                          # (1, L, d)
    q_rot, k_rot = apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=1)
    # frames
    n = torch.randn(5, 7, 3, generator=g) * 3
    ca = n + torch.randn(5, 7, 3, generator=g)
    c = ca + torch.randn(5, 7, 3, generator=g)
    r = Rigid.from_3_points(c, ca, n)
    np.savez(OUT, q=q.numpy(), k=k.numpy(), q_rot=q_rot.numpy(), k_rot=k_rot.numpy(), rotate_half_q=rotate_half(q).numpy(),
             cos=cos.numpy(), sin=sin.numpy(), n=n.numpy(), ca=ca.numpy(), c=c.numpy(),
             rot=r.get_rots().get_rot_mats().numpy(), trans=r.get_trans().numpy())
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
