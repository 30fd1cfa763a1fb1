"""r05: is F32_SPLIT float32-grade on trained-like weights?  Engines (f32, f32_split) against a float64 evaluation of the oracle
network (torch CPU, .double()), for truncated stacks of k blocks: logits and the pre-norm hidden state.  WEIGHTS=random for the
default-init control."""
import dataclasses, json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from esmdiff_amd.config import ESM3_OPEN
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import trained_like_state_dict, random_init_state_dict
from oracle.esm3_ref import build_from_state_dict
torch.set_num_threads(os.cpu_count())
kind = os.environ.get("WEIGHTS", "trained")
ks = [int(v) for v in os.environ.get("KS", "1,4,12,48").split(",")]
mk = trained_like_state_dict if kind == "trained" else random_init_state_dict
opts = json.loads(os.environ.get("OPTS", "{}"))
sd_full = mk(ESM3_OPEN, seed=11, device="cpu", **opts)
n, L = 2, 258
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(n, 1)
x = torch.full((n, L), 4096, dtype=torch.int64)
x[:, 1::3] = torch.randint(0, 4096, (n, len(range(1, L, 3))), generator=g)
sch = ddpm_schedule(25, freq_dim=256)
for k in ks:
    cfg = dataclasses.replace(ESM3_OPEN, n_layers=k)
    sd = {kk: v for kk, v in sd_full.items() if ".blocks." not in kk or int(kk.split(".blocks.")[1].split(".")[0]) < k}
    net, emb = build_from_state_dict(cfg, sd)
    t0 = time.time()
    net64, emb64 = net.double(), emb.double()
    rec = {"k": k}
    for prec in ("f32", "f32_split"):
        eng = Engine(cfg, sd, max_batch=n, max_len=L, precision=prec)
        lg = eng.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[5]).double().cpu()
        hid = eng.embeddings(n, L).double().cpu()
        eng.close()
        rec[prec] = (lg, hid)
    torch.set_default_dtype(torch.float64)
    with torch.no_grad():
        # conditioning exactly as the engine gets it: the sinusoid row of the schedule through the float64 embedder MLP
        c64 = emb64.mlp(sch.t_freq[5].double()[None].repeat(n, 1))
        out = net64(structure_tokens=x, sequence_tokens=seq, auxiliary_embeddings=torch.tile(c64[:, None, :], (1, L, 1)))
        ref_lg, ref_hid = out.structure_logits, out.embeddings
    torch.set_default_dtype(torch.float32)
    res = {"k": k, "weights": kind, "oracle_f64_seconds": round(time.time() - t0, 1), "logit_std": float(ref_lg.std()),
           "hidden_absmax": float(ref_hid.abs().max())}
    for prec in ("f32", "f32_split"):
        lg, hid = rec[prec]
        res[prec] = {"logit_max_err": float((lg - ref_lg).abs().max()), "logit_rms_err": float((lg - ref_lg).pow(2).mean().sqrt()),
                     "hidden_max_err": float((hid - ref_hid).abs().max()), "hidden_rms_err": float((hid - ref_hid).pow(2).mean().sqrt())}
    res["split_vs_f32_logit_max"] = float((rec["f32"][0] - rec["f32_split"][0]).abs().max())
    print(json.dumps(res), flush=True)
