"""max |err| of qk_norm_rope + attention vs f32 SDPA for a list of (B, L) (debugging aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd.config import TINY
from esmdiff_amd.engine import Engine
from esmdiff_amd.weights import random_init_state_dict
from oracle.esm3_ref import build_from_state_dict
torch.set_num_threads(8)
cfg = TINY
sd = random_init_state_dict(cfg, seed=1)
eng = Engine(cfg, sd, max_batch=8, max_len=1100)
net, _ = build_from_state_dict(cfg, sd)
attn = net.transformer.blocks[0].attn
for B, L in [(1, 64), (2, 64), (1, 128), (1, 63), (1, 65), (1, 258), (2, 33), (1, 1026)]:
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B, L, 3 * cfg.d_model, generator=g).to(torch.bfloat16)
    with torch.no_grad():
        q, k, v = torch.chunk(qkv.float(), 3, dim=-1)
        q, k = attn._rope(attn.q_ln(q), attn.k_ln(k))
        v = v.view(B, L, cfg.n_heads, 64).transpose(1, 2)
        ref = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v)
        ref = ref.transpose(1, 2).reshape(B * L, cfg.d_model)
    got = eng.attention(qkv.reshape(B * L, -1).contiguous().cuda(), attn.q_ln.weight.cuda(), attn.k_ln.weight.cuda(), B, L).float().cpu()
    err = (got - ref).abs()
    bad = (err > 6e-2).view(B, L, cfg.n_heads, 64).any(-1)
    print(f"B={B} L={L}: max {float(err.max()):.4f} mean {float(err.mean()):.5f} nan {int(torch.isnan(got).sum())} "
          f"bad rows {int(bad.any(-1).sum())} first bad {bad.nonzero()[:4].tolist()}")
