#!/usr/bin/env python3
"""Golden vectors of the network arithmetic from the REAL `esm` package -> tests/golden/esm/*.npz.

    python tools/dump_esm_vectors.py [--out tests/golden/esm] [--with-coordinates]

WHY.  Everything `esm==3.0.4` owns on the hot path — the ESM3 block stack the reference drives at
/root/reference/slm/models/net.py:336-356, 441, 455-469, the iterative sampler it calls at
/root/reference/slm/sample_esmdiff.py:114-122, the VQ-VAE encoder / decoder behind models/utils.py:136-137 and
sample_esmdiff.py:40-61 — is restated in oracle/*_ref.py FROM MEMORY, because the package is in neither /root/reference nor
the build image (requirements.txt:30).  Parity for those parts is therefore "unpinned".  This script is the other half of the
pin: run it ONCE on any machine where `pip install esm==3.0.4` works (CPU is enough, a minute), commit the .npz files it writes,
and `python -m pytest tests/test_oracle_golden.py -k esm_golden` turns from "skipped: no vectors" into a real comparison of
every restatement with esm's own output (tolerances stated in the tests).

WHAT IT WRITES (seeded, small configurations — KB to a few MB each; float32 on CPU):
  esm3_stack.npz        EncodeInputs + TransformerStack(d 512, 8 heads, 128 vector heads, 2 blocks, mask_and_zero_frameless)
                        + the reference's 4101-way structure head (RegressionHead), called exactly as
                        CustomizedESM3.forward does (net.py:410-469): state dict, tokens, auxiliary embedding ->
                        structure_logits, embeddings; with --with-coordinates also a run with backbone coordinates
                        (block 0's geometric attention live) and the affine frames esm derives from them
  sampling.npz          esm.utils.sampling on seeded logits: top_p_logits outputs, the entropy / position-selection inputs of
                        one iterative_sampling_raw-style step (only functions that exist in the installed version are dumped)
  structure_decoder.npz StructureTokenDecoder(d 256, 4 heads, 2 blocks): state dict, tokens -> bb_pred (N, CA, C), plddt, ptm
  structure_encoder.npz StructureTokenEncoder(d 128, 2 blocks, 64 codes, d_out 16): state dict, coordinates -> codes
Each section is independent: one that fails (an API that moved) is reported and skipped, the others are still written.

This file imports NOTHING from this repository and is never shipped to the GPU box (.gpurunignore); the consuming tests import
only numpy + the oracle.  The key names written are esm's own state-dict keys (the ones esmdiff_amd/weights.py documents).
"""
import argparse
import json
import sys
import traceback
from pathlib import Path

import numpy as np
import torch


def _sd(module):
    return {"sd::" + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def dump_esm3_stack(out: Path, with_coordinates: bool):
    from esm.layers.regression_head import RegressionHead
    from esm.layers.transformer_stack import TransformerStack
    from esm.models.esm3 import EncodeInputs
    from esm.utils.constants import esm3 as C
    from esm.utils.structure.affine3d import build_affine3d_from_coordinates
    torch.manual_seed(0)
    d, heads, v_heads, layers, vocab = 512, 8, 128, 2, 4101
    enc = EncodeInputs(d)
    stack = TransformerStack(d, heads, v_heads, layers, mask_and_zero_frameless=True)
    head = RegressionHead(d, vocab)
    for m in (enc, stack, head):
        m.eval()
    with torch.no_grad():                                   # spread the parameters a little: default inits leave LayerNorms at 1 / 0
        for p in list(stack.parameters()) + list(head.parameters()):
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    B, L = 2, 24
    g = torch.Generator().manual_seed(1)
    seq = torch.randint(4, 24, (B, L), generator=g)
    seq[:, 0], seq[:, -1] = C.SEQUENCE_BOS_TOKEN, C.SEQUENCE_EOS_TOKEN
    st = torch.full((B, L), C.STRUCTURE_MASK_TOKEN, dtype=torch.long)
    st[:, 3:11] = torch.randint(0, 4096, (B, 8), generator=g)
    aux = 0.3 * torch.randn(B, 1, d, generator=g).expand(B, L, d).contiguous()

    def forward(coords):
        # CustomizedESM3.forward, net.py:410-469, every optional track at its default
        ss8 = torch.full((1, L), C.SS8_PAD_TOKEN, dtype=torch.long)
        sasa = torch.full((1, L), C.SASA_PAD_TOKEN, dtype=torch.long)
        chain_id = torch.zeros(1, L, dtype=torch.long)
        avg_plddt, res_plddt = torch.ones(1, L), torch.zeros(1, L)
        res_ann = torch.full((1, L, 16), C.RESIDUE_PAD_TOKEN, dtype=torch.long)
        func = torch.full((1, L, 8), C.INTERPRO_PAD_TOKEN, dtype=torch.long)
        xyz = torch.full((1, L, 3, 3), float("nan")) if coords is None else coords
        affine, affine_mask = build_affine3d_from_coordinates(xyz[..., :3, :])
        s = (st.masked_fill(st == -1, C.STRUCTURE_MASK_TOKEN)
             .masked_fill(seq == C.SEQUENCE_BOS_TOKEN, C.STRUCTURE_BOS_TOKEN)
             .masked_fill(seq == C.SEQUENCE_PAD_TOKEN, C.STRUCTURE_PAD_TOKEN)
             .masked_fill(seq == C.SEQUENCE_EOS_TOKEN, C.STRUCTURE_EOS_TOKEN)
             .masked_fill(seq == C.SEQUENCE_CHAINBREAK_TOKEN, C.STRUCTURE_CHAINBREAK_TOKEN))
        x = enc(seq, s, avg_plddt, res_plddt, ss8, sasa, func, res_ann) + aux
        res = stack(x, None, affine, affine_mask, chain_id)
        x, emb = res[0], res[1]
        return head(x), emb, affine, affine_mask

    rec = {"cfg_json": json.dumps({"d_model": d, "n_heads": heads, "v_heads": v_heads, "n_layers": layers, "n_structure_heads": vocab}),
           "in::sequence_tokens": seq.numpy(), "in::structure_tokens": st.numpy(), "in::auxiliary_embeddings": aux.numpy()}
    for prefix, m in (("encoder.", enc), ("transformer.", stack), ("output_heads.structure_head.", head)):
        rec.update({"sd::" + prefix + k: v.detach().numpy() for k, v in m.state_dict().items()})
    with torch.no_grad():
        lg, emb, _, _ = forward(None)
        rec["out::structure_logits"], rec["out::embeddings"] = lg.numpy(), emb.numpy()
        if with_coordinates:
            ca = torch.cumsum(torch.randn(B, L, 3, generator=g) * 2.2, 1)
            xyz = torch.stack([ca + 0.8 * torch.randn(B, L, 3, generator=g), ca, ca + 0.8 * torch.randn(B, L, 3, generator=g)], 2)
            xyz[:, 0], xyz[:, -1] = float("nan"), float("nan")
            xyz[0, 7:10] = float("inf")
            lg2, emb2, affine, mask = forward(xyz)
            rec["in::structure_coords"] = xyz.numpy()
            rec["out::structure_logits_with_coords"], rec["out::embeddings_with_coords"] = lg2.numpy(), emb2.numpy()
            rec["out::affine_mask"] = mask.numpy()
            try:
                rec["out::affine_rot"] = affine.rot.tensor.numpy()
                rec["out::affine_trans"] = affine.trans.numpy()
            except Exception:                                           # the frame container's attribute names differ between versions
                pass
    np.savez_compressed(out / "esm3_stack.npz", **rec)


def dump_sampling(out: Path):
    import esm.utils.sampling as S
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(2)
    logits = 3.0 * torch.randn(3, 12, 4101, generator=g)
    rec = {"in::logits": logits.numpy(), "functions_present": np.array(sorted(n for n in dir(S) if not n.startswith("_")))}
    if hasattr(S, "top_p_logits"):
        for p in (0.9, 0.5, 1.0):
            rec[f"out::top_p_logits_{p}"] = S.top_p_logits(logits.clone(), p).numpy()
    if hasattr(S, "sample_logits"):
        torch.manual_seed(5)
        rec["out::sample_logits_t0"] = S.sample_logits(logits.clone(), temperature=0.0).numpy()
    if hasattr(S, "_tensorize_like") or hasattr(S, "get_default_sampling_config"):
        pass                                                            # (nothing numeric to pin)
    try:                                                                # the entropy the "entropy" strategy sorts by
        lp = torch.log_softmax(logits, -1)
        rec["out::entropy"] = (-(lp.exp() * lp).sum(-1)).numpy()
    except Exception:
        pass
    try:
        from esm.sdk.api import GenerationConfig
        cfg = GenerationConfig(track="structure", num_steps=8, temperature=1.4, top_p=0.9)
        rec["generation_config_json"] = json.dumps({k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v))
                                                    for k, v in vars(cfg).items()})
    except Exception:
        pass
    try:                                                                # the unmasking schedule of iterative_sampling_raw
        from esm.utils.noise_schedules import NOISE_SCHEDULE_REGISTRY
        t = torch.linspace(0, 1, 9)
        for name, fn in NOISE_SCHEDULE_REGISTRY.items():
            rec[f"out::noise_schedule_{name}"] = fn(t).numpy()
    except Exception:
        pass
    np.savez_compressed(out / "sampling.npz", **rec)


def dump_structure_decoder(out: Path):
    from esm.models.vqvae import StructureTokenDecoder
    torch.manual_seed(0)
    d, heads, layers = 256, 4, 2
    dec = StructureTokenDecoder(d_model=d, n_heads=heads, n_layers=layers).eval()
    g = torch.Generator().manual_seed(3)
    tok = torch.randint(0, 4096, (2, 20), generator=g)
    tok[:, 0], tok[:, -1] = 4098, 4097
    with torch.no_grad():
        res = dec.decode(tok)
    rec = {"cfg_json": json.dumps({"d_model": d, "n_heads": heads, "n_layers": layers}), "in::structure_tokens": tok.numpy(), **_sd(dec)}
    for k, v in res.items():
        if torch.is_tensor(v):
            rec["out::" + k] = v.detach().numpy()
    np.savez_compressed(out / "structure_decoder.npz", **rec)


def dump_structure_encoder(out: Path):
    from esm.models.vqvae import StructureTokenEncoder
    torch.manual_seed(0)
    d, heads, v_heads, layers, d_out, n_codes = 128, 1, 128, 2, 16, 64
    enc = StructureTokenEncoder(d_model=d, n_heads=heads, v_heads=v_heads, n_layers=layers, d_out=d_out, n_codes=n_codes).eval()
    g = torch.Generator().manual_seed(4)
    B, L = 2, 30
    ca = torch.cumsum(torch.randn(B, L, 3, generator=g) * 2.2, 1)
    xyz = torch.stack([ca + 0.8 * torch.randn(B, L, 3, generator=g), ca, ca + 0.8 * torch.randn(B, L, 3, generator=g)], 2)
    xyz[1, 11:14] = float("inf")
    with torch.no_grad():
        res = enc.encode(xyz)
    z, codes = (res if isinstance(res, tuple) else (None, res))
    rec = {"cfg_json": json.dumps({"d_model": d, "n_heads": heads, "v_heads": v_heads, "n_layers": layers, "d_out": d_out, "n_codes": n_codes}),
           "in::coordinates": xyz.numpy(), "out::codes": codes.numpy(), **_sd(enc)}
    if z is not None:
        rec["out::z_q"] = z.detach().numpy()
    np.savez_compressed(out / "structure_encoder.npz", **rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(Path(__file__).resolve().parent.parent / "tests" / "golden" / "esm"))
    ap.add_argument("--with-coordinates", action="store_true", default=True)
    args = ap.parse_args()
    try:
        import esm
    except ImportError:
        sys.exit("tools/dump_esm_vectors.py needs the `esm` package (pip install esm==3.0.4, the reference's requirements.txt:30); "
                 "it is not part of this repository's image — run it where the reference itself runs")
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    ok = {}
    for name, fn in (("esm3_stack", lambda: dump_esm3_stack(out, args.with_coordinates)), ("sampling", lambda: dump_sampling(out)),
                     ("structure_decoder", lambda: dump_structure_decoder(out)), ("structure_encoder", lambda: dump_structure_encoder(out))):
        try:
            fn()
            ok[name] = "written"
        except Exception as ex:                                         # one moved API must not lose the other sections
            ok[name] = f"FAILED: {type(ex).__name__}: {ex}"
            traceback.print_exc()
    (out / "MANIFEST.json").write_text(json.dumps({"esm_version": getattr(esm, "__version__", "unknown"), "torch": torch.__version__,
                                                    "sections": ok}, indent=1))
    print(json.dumps(ok, indent=1))


if __name__ == "__main__":
    main()
