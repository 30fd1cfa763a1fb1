"""Consumers of tests/golden/esm/*.npz — the vectors tools/dump_esm_vectors.py writes where `esm==3.0.4` is installed.

Each check_* loads one file, rebuilds the matching restatement in oracle/ from the state dict INSIDE the file and compares its
output with esm's own.  tests/test_oracle_golden.py calls them when the file exists (skip with the reason otherwise) and, in
any case, runs them once on files written by write_selfcheck_vectors() — the same format filled by the oracle itself — so the
harness (key names, shapes, tolerance code) is known to work before the first real vector arrives."""
import json
from pathlib import Path

import numpy as np
import torch

ESM_DIR = Path(__file__).resolve().parent / "golden" / "esm"


def _load(path):
    z = np.load(path, allow_pickle=False)
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    cfg = json.loads(str(z["cfg_json"])) if "cfg_json" in z.files else {}
    return z, sd, cfg


def check_esm3_stack(path, tol=2e-4):
    """EncodeInputs + TransformerStack + structure head (oracle/esm3_ref.py) vs esm: logits and the
    pre-norm hidden state without coordinates, and — when the file carries them — with block 0's geometric attention live."""
    from esmdiff_amd.config import ModelConfig
    from oracle.esm3_ref import ESM3Ref
    z, sd, c = _load(path)
    cfg = ModelConfig(d_model=c["d_model"], n_heads=c["n_heads"], v_heads=c["v_heads"], n_layers=c["n_layers"],
                      n_structure_heads=c["n_structure_heads"])
    net = ESM3Ref(cfg, with_geom=any("geom_attn" in k for k in sd))
    missing, unexpected = net.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not missing, f"the restatement has parameters esm's state dict lacks: {missing}"
    unexpected = [k for k in unexpected if "function_embed" not in k and "residue_embed" not in k]
    assert not unexpected, f"esm parameters the restatement does not model: {unexpected}"
    net.eval()
    seq, st = torch.from_numpy(z["in::sequence_tokens"]), torch.from_numpy(z["in::structure_tokens"])
    aux = torch.from_numpy(z["in::auxiliary_embeddings"])
    out = {}
    with torch.no_grad():
        r = net(structure_tokens=st, sequence_tokens=seq, auxiliary_embeddings=aux)
        out["logits"] = float((r.structure_logits - torch.from_numpy(z["out::structure_logits"])).abs().max())
        out["embeddings"] = float((r.embeddings - torch.from_numpy(z["out::embeddings"])).abs().max())
        if "in::structure_coords" in z.files:
            r2 = net(structure_tokens=st, sequence_tokens=seq, auxiliary_embeddings=aux, structure_coords=torch.from_numpy(z["in::structure_coords"]))
            out["logits_with_coords"] = float((r2.structure_logits - torch.from_numpy(z["out::structure_logits_with_coords"])).abs().max())
            assert float((r2.structure_logits - r.structure_logits).abs().max()) > 1e-3, "coordinates changed nothing"
    scale = float(np.abs(z["out::structure_logits"]).max())
    for k, v in out.items():
        assert v <= tol * max(1.0, scale), (k, v, out)
    return out


def check_structure_decoder(path, tol=2e-4):
    """StructureTokenDecoder (oracle/decoder_ref.py) vs esm: backbone N / CA / C per residue, and pLDDT / pTM when both sides
    have the heads."""
    from esmdiff_amd.config import DecoderConfig
    from oracle.decoder_ref import StructureTokenDecoderRef
    from oracle.geom_ref import backbone_rmsd
    z, sd, c = _load(path)
    cfg = DecoderConfig(d_model=c["d_model"], n_heads=c["n_heads"], n_layers=c["n_layers"])
    net = StructureTokenDecoderRef(cfg, with_plddt="plddt_head.3.weight" in sd, with_pairwise="pairwise_classification_head.linear2.weight" in sd)
    missing, unexpected = net.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not missing, missing
    net.eval()
    tok = torch.from_numpy(z["in::structure_tokens"])
    key = next(k for k in ("out::bb_pred", "out::bb_coords") if k in z.files)
    want = torch.from_numpy(z[key])
    if want.shape[1] == tok.shape[1]:
        want = want[:, 1:-1]                                     # esm keeps the BOS / EOS rows, the restatement drops them
    with torch.no_grad():
        got = net(tok)
    out = {"backbone_max_abs": float((got - want[..., :3, :]).abs().max()), "backbone_rmsd": float(backbone_rmsd(got, want[..., :3, :]).max()),
           "esm_keys_not_modelled": sorted(unexpected)}
    assert out["backbone_max_abs"] <= tol * max(1.0, float(want.abs().max())), out
    return out


def check_structure_encoder(path):
    """StructureTokenEncoder (oracle/encoder_ref.py) vs esm: the codes, identical except at exact nearest-code ties."""
    from esmdiff_amd.config import EncoderConfig
    from oracle.encoder_ref import StructureTokenEncoderRef
    z, sd, c = _load(path)
    cfg = EncoderConfig(d_model=c["d_model"], v_heads=c["v_heads"], n_layers=c["n_layers"], d_out=c["d_out"], n_codes=c["n_codes"])
    net = StructureTokenEncoderRef(cfg)
    missing, unexpected = net.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not missing, missing
    net.eval()
    with torch.no_grad():
        got = net(torch.from_numpy(z["in::coordinates"]))
    want = torch.from_numpy(z["out::codes"])
    agree = float((got == want).float().mean())
    assert agree >= 0.98, (agree, sorted(unexpected))
    return {"codes_agreement": agree, "esm_keys_not_modelled": sorted(unexpected)}


def check_sampling(path, tol=1e-6):
    """esm.utils.sampling.top_p_logits vs oracle/gibbs_ref.top_p_logits on the same logits (kept entries equal, cut entries at
    the same positions), and the entropy the "entropy" strategy sorts positions by."""
    from oracle.gibbs_ref import top_p_logits
    z = np.load(path, allow_pickle=False)
    logits = torch.from_numpy(z["in::logits"])
    out = {}
    for k in z.files:
        if k.startswith("out::top_p_logits_"):
            p = float(k.rsplit("_", 1)[1])
            want = torch.from_numpy(z[k])
            got = top_p_logits(logits.clone(), p)
            kept_w, kept_g = want > -1e30, got > -1e30
            assert torch.equal(kept_w, kept_g), (p, int((kept_w != kept_g).sum()))
            assert float((got[kept_g] - want[kept_w]).abs().max()) <= tol
            out[f"top_p_{p}_kept"] = int(kept_g.sum())
    if "out::entropy" in z.files:
        lp = torch.log_softmax(logits, -1)
        assert float(((-(lp.exp() * lp).sum(-1)) - torch.from_numpy(z["out::entropy"])).abs().max()) <= 1e-5
    return out


CHECKS = {"esm3_stack.npz": check_esm3_stack, "structure_decoder.npz": check_structure_decoder,
          "structure_encoder.npz": check_structure_encoder, "sampling.npz": check_sampling}


def write_selfcheck_vectors(out: Path):
    """The dump tool's file format, filled by the ORACLE (random small models): exercises every consumer above end to end."""
    from esmdiff_amd.config import DecoderConfig, EncoderConfig, ModelConfig
    from esmdiff_amd.weights import random_init_decoder_state_dict, random_init_encoder_state_dict, random_init_state_dict
    from oracle.decoder_ref import build_decoder_from_state_dict
    from oracle.encoder_ref import build_encoder_from_state_dict
    from oracle.esm3_ref import ESM3Ref
    from oracle.gibbs_ref import top_p_logits
    out.mkdir(parents=True, exist_ok=True)
    g = torch.Generator().manual_seed(1)
    cfg = ModelConfig(d_model=512, n_heads=8, v_heads=128, n_layers=2)
    sd = {k[4:]: v for k, v in random_init_state_dict(cfg, seed=3, with_geom=True).items() if k.startswith("net.")}
    net = ESM3Ref(cfg, with_geom=True)
    net.load_state_dict(sd, strict=False)
    net.eval()
    B, L = 2, 24
    seq = torch.randint(4, 24, (B, L), generator=g)
    seq[:, 0], seq[:, -1] = 0, 2
    st = torch.full((B, L), 4096, dtype=torch.long)
    st[:, 3:11] = torch.randint(0, 4096, (B, 8), generator=g)
    aux = 0.3 * torch.randn(B, 1, cfg.d_model, generator=g).expand(B, L, cfg.d_model).contiguous()
    ca = torch.cumsum(torch.randn(B, L, 3, generator=g) * 2.2, 1)
    xyz = torch.stack([ca + 0.8 * torch.randn(B, L, 3, generator=g), ca, ca + 0.8 * torch.randn(B, L, 3, generator=g)], 2)
    xyz[:, 0], xyz[:, -1] = float("nan"), float("nan")
    with torch.no_grad():
        r = net(structure_tokens=st, sequence_tokens=seq, auxiliary_embeddings=aux)
        r2 = net(structure_tokens=st, sequence_tokens=seq, auxiliary_embeddings=aux, structure_coords=xyz)
    np.savez_compressed(out / "esm3_stack.npz", cfg_json=json.dumps({"d_model": 512, "n_heads": 8, "v_heads": 128, "n_layers": 2, "n_structure_heads": 4101}),
                        **{"sd::" + k: v.numpy() for k, v in sd.items()}, **{"in::sequence_tokens": seq.numpy(), "in::structure_tokens": st.numpy(),
                        "in::auxiliary_embeddings": aux.numpy(), "in::structure_coords": xyz.numpy(), "out::structure_logits": r.structure_logits.numpy(),
                        "out::embeddings": r.embeddings.numpy(), "out::structure_logits_with_coords": r2.structure_logits.numpy()})
    dcfg = DecoderConfig(d_model=256, n_heads=4, n_layers=2)
    dsd = random_init_decoder_state_dict(dcfg, seed=2, with_plddt=False, with_pairwise=False)
    tok = torch.randint(0, 4096, (2, 20), generator=g)
    tok[:, 0], tok[:, -1] = 4098, 4097
    with torch.no_grad():
        bb = build_decoder_from_state_dict(dcfg, dsd)(tok)
    np.savez_compressed(out / "structure_decoder.npz", cfg_json=json.dumps({"d_model": 256, "n_heads": 4, "n_layers": 2}),
                        **{"sd::" + k: v.numpy() for k, v in dsd.items()}, **{"in::structure_tokens": tok.numpy(), "out::bb_pred": bb.numpy()})
    ecfg = EncoderConfig(d_model=128, v_heads=128, n_layers=2, d_out=16, n_codes=64)
    esd = random_init_encoder_state_dict(ecfg, seed=5)
    with torch.no_grad():
        codes = build_encoder_from_state_dict(ecfg, esd)(xyz[:, 1:-1])
    np.savez_compressed(out / "structure_encoder.npz", cfg_json=json.dumps({"d_model": 128, "n_heads": 1, "v_heads": 128, "n_layers": 2, "d_out": 16, "n_codes": 64}),
                        **{"sd::" + k: v.numpy() for k, v in esd.items()}, **{"in::coordinates": xyz[:, 1:-1].numpy(), "out::codes": codes.numpy()})
    logits = 3.0 * torch.randn(3, 12, 4101, generator=g)
    lp = torch.log_softmax(logits, -1)
    np.savez_compressed(out / "sampling.npz", **{"in::logits": logits.numpy(), "out::top_p_logits_0.9": top_p_logits(logits.clone(), 0.9).numpy(),
                        "out::top_p_logits_0.5": top_p_logits(logits.clone(), 0.5).numpy(), "out::entropy": (-(lp.exp() * lp).sum(-1)).numpy()})
