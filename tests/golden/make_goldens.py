#!/usr/bin/env python3
"""Generate golden vectors by importing the REFERENCE's own sampler code.

Runs ONLY in the build container (needs /root/reference).  Nothing in the
tests, smoke() or bench.py imports this file or reads /root/reference; they
read the small .npz/.json fixtures this script writes next to itself.

How the reference is imported: its sampler layer (slm/models/model.py,
slm/utils/noise_utils.py, slm/models/net.py::TimestepEmbedder,
slm/utils/eval_utils.py::merge_pdbfiles) depends on packages that are not
installed here (lightning, torchmetrics, esm, hydra, biotite, ...).  A
sys.meta_path finder serves inert stub modules for those names; four stubs
carry the few attributes the sampler really touches (LightningModule ->
nn.Module with .device, Mean/MinMetric -> no-op modules, esm constants, an
empty ESM3 base class).  The arithmetic that is recorded below is executed by
the reference's code, not by a restatement.

Fixtures:
  G1 schedule tables          LogLinearNoise / CosineNoise sigma, dsigma, move chance
  G1b other schedules         CosineSqrNoise / Linear / GeometricNoise tables, importance_sampling_transformation
  G2 timestep embedding       TimestepEmbedder.timestep_embedding + MLP
  G3 logits_parameterization  seeded logits with mixed masks
  G4 _sample_categorical      with the uniforms torch drew recorded
  G5 one _ddpm_update         around the lookup stand-in net
  G6 full ddpm_sample         T=5 / 25, prior None and partial prior; x after every step
  G7 batch split lists        sample_esmdiff.py:104-112 / :181-193
  G8 merge_pdbfiles           two 3-residue inputs
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch
from torch import nn

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT.parent.parent))  # repo root, for tests.standin_net

# --------------------------------------------------------------------------
# stub machinery
# --------------------------------------------------------------------------
STUB_ROOTS = {
    "lightning", "torchmetrics", "esm", "hydra", "omegaconf", "rootutils",
    "biotite", "Bio", "tree", "deeptime", "mdtraj", "lightning_utilities",
    "rich", "wandb",
}


class _Anything:
    """Inert attribute sink: any attribute / call returns another sink."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root in STUB_ROOTS and root not in _REAL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


_REAL = set()
for _r in list(STUB_ROOTS):
    try:
        if importlib.util.find_spec(_r) is not None and _r == "rich":
            _REAL.add(_r)
    except Exception:
        pass


def _populate(module):
    name = module.__name__
    if name == "lightning":
        class LightningModule(nn.Module):
            @property
            def device(self):
                try:
                    return next(self.parameters()).device
                except StopIteration:
                    return torch.device("cpu")

            def log(self, *a, **k):
                pass

        module.LightningModule = LightningModule
        module.LightningDataModule = object
        module.Callback = object
        module.Trainer = object
    elif name == "torchmetrics":
        class _M(nn.Module):
            def forward(self, *a, **k):
                return None

            def reset(self):
                pass

            def compute(self):
                return torch.tensor(0.0)

        module.MeanMetric = _M
        module.MinMetric = _M
        module.MaxMetric = _M
    elif name == "esm.utils.constants.esm3":
        # [ESM-RECALL] cross-checked in-tree: model.py:380, configs/model/default.yaml:39-41
        module.VQVAE_CODEBOOK_SIZE = 4096
        module.STRUCTURE_MASK_TOKEN = 4096
        module.STRUCTURE_EOS_TOKEN = 4097
        module.STRUCTURE_BOS_TOKEN = 4098
        module.STRUCTURE_PAD_TOKEN = 4099
        module.STRUCTURE_CHAINBREAK_TOKEN = 4100
        module.SEQUENCE_BOS_TOKEN = 0
        module.SEQUENCE_PAD_TOKEN = 1
        module.SEQUENCE_EOS_TOKEN = 2
        module.SEQUENCE_CHAINBREAK_TOKEN = 31
        module.SEQUENCE_MASK_TOKEN = 32
        module.SS8_PAD_TOKEN = 0
        module.SASA_PAD_TOKEN = 0
        module.RESIDUE_PAD_TOKEN = 0
        module.INTERPRO_PAD_TOKEN = 0
    elif name == "tree":
        # dm-tree's map_structure on nested lists/tuples/dicts (residue_constants.py:753 needs it at import)
        def map_structure(fn, s):
            if isinstance(s, (list, tuple)):
                return type(s)(map_structure(fn, e) for e in s)
            if isinstance(s, dict):
                return {k: map_structure(fn, v) for k, v in s.items()}
            return fn(s)

        module.map_structure = map_structure
    elif name == "esm.utils.constants":
        pass
    elif name == "esm.models.esm3":
        from dataclasses import dataclass

        class ESM3(nn.Module):
            pass

        @dataclass
        class ESMOutput:
            sequence_logits: object = None
            structure_logits: object = None
            secondary_structure_logits: object = None
            sasa_logits: object = None
            function_logits: object = None
            residue_logits: object = None
            embeddings: object = None

        module.ESM3 = ESM3
        module.ESMOutput = ESMOutput
        module.EncodeInputs = _Anything
        module.OutputHeads = _Anything


sys.meta_path.insert(0, _StubFinder())
sys.path.insert(0, str(REF))
os.chdir(tempfile.gettempdir())

# the reference's own modules ------------------------------------------------
import esm.utils.constants.esm3 as C  # noqa: E402  (stub with the constants above)
import esm.utils.constants as _ec  # noqa: E402
_ec.esm3 = C
from slm.utils import noise_utils  # noqa: E402
from slm.models import model as ref_model  # noqa: E402
from slm.models.net import TimestepEmbedder  # noqa: E402
from slm.utils.eval_utils import merge_pdbfiles  # noqa: E402

from tests.standin_net import StandinNet, standin_sigma_embedder_state  # noqa: E402

torch.set_num_threads(1)
V = 4101
MASK = 4096


def build_model(sigma_hidden=32):
    emb = TimestepEmbedder(sigma_hidden)
    emb.load_state_dict(standin_sigma_embedder_state(sigma_hidden))
    net = StandinNet(sigma_hidden)
    m = ref_model.MaskedDiffusionLanguageModeling(
        net=net, optimizer=None, scheduler=None, compile=False,
        noise_schedule=noise_utils.LogLinearNoise(),
        sigma_embedder=emb, time_conditioning=True, T=0, sampling_eps=1e-3,
        noise_removal=True,
    )
    m.eval()
    return m


def g1_schedules():
    out = {}
    for T in (5, 25, 50):
        for eps in (1e-5,):
            ts = torch.linspace(1.0, eps, T + 1)
            dt = (1 - eps) / T
            out[f"timesteps_T{T}"] = ts.numpy()
            out[f"dt_T{T}"] = np.float64(dt)
            for nm, sched in (("loglinear", noise_utils.LogLinearNoise()),
                              ("cosine", noise_utils.CosineNoise(eps=1e-3))):
                t = ts[:, None]
                sig_t, dsig_t = sched(t)
                sig_s, _ = sched(t - dt)
                out[f"{nm}_sigma_t_T{T}"] = sig_t.squeeze(-1).numpy()
                out[f"{nm}_dsigma_t_T{T}"] = dsig_t.squeeze(-1).numpy()
                out[f"{nm}_sigma_s_T{T}"] = sig_s.squeeze(-1).numpy()
                out[f"{nm}_mc_t_T{T}"] = (1 - torch.exp(-sig_t.squeeze(-1))).numpy()
                out[f"{nm}_mc_s_T{T}"] = (1 - torch.exp(-sig_s.squeeze(-1))).numpy()
    ll = noise_utils.LogLinearNoise()
    out["loglinear_sigma_max"] = ll.sigma_max.numpy()
    out["loglinear_sigma_min"] = ll.sigma_min.numpy()
    np.savez_compressed(OUT / "g1_schedules.npz", **out)


def g1b_schedules_other():
    """The schedules `get_noise` can also build (noise_utils.py:75-90, :138-185): CosineSqrNoise, Linear, GeometricNoise, at the
    sampler's T = 25 grid; plus importance_sampling_transformation of Linear and LogLinearNoise on a t grid."""
    out = {}
    T, eps = 25, 1e-5
    ts = torch.linspace(1.0, eps, T + 1)
    dt = (1 - eps) / T
    t = ts[:, None]
    for nm, sched in (("cosinesqr", noise_utils.CosineSqrNoise(eps=1e-3)), ("linear", noise_utils.Linear(0, 10)),
                      ("geometric", noise_utils.GeometricNoise(1e-3, 1))):
        sig_t, dsig_t = sched(t)
        sig_s, _ = sched(t - dt)
        out[f"{nm}_sigma_t"] = sig_t.squeeze(-1).numpy()
        out[f"{nm}_dsigma_t"] = (dsig_t * torch.ones_like(t)).squeeze(-1).numpy()
        out[f"{nm}_sigma_s"] = sig_s.squeeze(-1).numpy()
        out[f"{nm}_mc_t"] = (1 - torch.exp(-sig_t.squeeze(-1))).numpy()
        out[f"{nm}_mc_s"] = (1 - torch.exp(-sig_s.squeeze(-1))).numpy()
    grid = torch.linspace(0.0, 1.0, 21)
    out["t_grid"] = grid.numpy()
    out["linear_importance"] = noise_utils.Linear(1e-3, 10).importance_sampling_transformation(grid).numpy()
    out["linear_importance_sigma_min0"] = noise_utils.Linear(0, 10).importance_sampling_transformation(grid).numpy()   # degenerate: 0 / nan
    out["loglinear_importance"] = noise_utils.LogLinearNoise().importance_sampling_transformation(grid).numpy()
    np.savez_compressed(OUT / "g1b_schedules_other.npz", **out)


def g2_timestep():
    sig = torch.tensor([0.0, 1e-5, 0.01, 0.5, 1.0, 3.3, 6.9077683], dtype=torch.float32)
    out = {"sigma": sig.numpy()}
    out["freq_embedding_256"] = TimestepEmbedder.timestep_embedding(sig, 256).numpy()
    out["freq_embedding_7"] = TimestepEmbedder.timestep_embedding(sig, 7).numpy()
    for h in (32, 64):
        emb = TimestepEmbedder(h)
        emb.load_state_dict(standin_sigma_embedder_state(h))
        with torch.no_grad():
            out[f"mlp_out_h{h}"] = emb(sig).numpy()
    np.savez_compressed(OUT / "g2_timestep.npz", **out)


def g3_logits_param():
    m = build_model()
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(2, 6, V, generator=g) * 3.0
    xt = torch.full((2, 6), MASK, dtype=torch.long)
    xt[0, 1] = 17
    xt[0, 4] = 4095
    xt[1, 0] = 4098
    xt[1, 5] = 4097
    xt[1, 2] = 0
    lin = logits.clone()
    lp = m.logits_parameterization(logits=logits.clone(), xt=xt)
    np.savez_compressed(OUT / "g3_logits_param.npz", logits=lin.numpy(), xt=xt.numpy(),
                        log_p=lp.numpy())


def g4_categorical():
    g = torch.Generator().manual_seed(4)
    probs = torch.softmax(torch.randn(3, 7, V, generator=g) * 2.0, -1) * 0.04
    probs[..., MASK] = 0.9
    torch.manual_seed(1234)
    u = torch.rand_like(probs)
    torch.manual_seed(1234)
    ids = ref_model._sample_categorical(probs)
    np.savez_compressed(OUT / "g4_categorical.npz", probs=probs.numpy(), u=u.numpy(),
                        ids=ids.numpy(), seed=np.int64(1234))


def g5_ddpm_update():
    m = build_model()
    B, L = 3, 9
    seq = torch.tensor([[0, 5, 9, 13, 7, 4, 20, 11, 2]]).repeat(B, 1)
    x = torch.full((B, L), MASK, dtype=torch.long)
    x[0, 3] = 100
    x[2, 7] = 4000
    T = 25
    ts = torch.linspace(1.0, 1e-5, T + 1)
    dt = (1 - 1e-5) / T
    i = 7
    t = ts[i] * torch.ones(B, 1)
    with torch.no_grad():
        sigma_t, _ = m.noise(t)
        logp, _ = m._model_wrapper(x.clone(), seq, sigma_t)
        torch.manual_seed(55)
        u = torch.rand(B, L, V)
        torch.manual_seed(55)
        x_new = m._ddpm_update(x.clone(), t, sequence_tokens=seq, dt=dt)
    np.savez_compressed(OUT / "g5_ddpm_update.npz", seq=seq.numpy(), x=x.numpy(), t=t.numpy(),
                        dt=np.float64(dt), log_p=logp.numpy(), u=u.numpy(), x_new=x_new.numpy(),
                        seed=np.int64(55), step=np.int64(i), T=np.int64(T))


def g6_ddpm_sample():
    out = {}
    seq1 = torch.tensor([0] + [4 + (7 * i) % 20 for i in range(10)] + [2])
    for tag, T, B, prior_kind, seed in (("T5_noprior", 5, 2, None, 11),
                                        ("T25_noprior", 25, 3, None, 12),
                                        ("T25_prior", 25, 2, "partial", 13),
                                        ("T5_prior", 5, 4, "partial", 14)):
        m = build_model()
        seq = seq1[None].repeat(B, 1)
        L = seq.shape[1]
        prior = None
        if prior_kind == "partial":
            prior = torch.tensor([4098] + [(37 * i + 5) % 4096 for i in range(L - 2)] + [4097])[None].repeat(B, 1)
            prior[:, 3] = MASK
            prior[:, 6] = MASK
        traj = []
        orig = m._ddpm_update

        def hooked(x, t, sequence_tokens, dt, _o=orig, _traj=traj):
            r = _o(x, t, sequence_tokens=sequence_tokens, dt=dt)
            _traj.append(r.clone())
            return r

        m._ddpm_update = hooked
        torch.manual_seed(seed)
        xf = m.ddpm_sample(sequence_tokens=seq, num_steps=T, eps=1e-5,
                           input_prior=None if prior is None else prior.clone(), sample_max_t=1.0)
        out[f"{tag}_seq"] = seq.numpy()
        out[f"{tag}_traj"] = torch.stack(traj).numpy()
        out[f"{tag}_final"] = xf.numpy()
        out[f"{tag}_seed"] = np.int64(seed)
        out[f"{tag}_T"] = np.int64(T)
        if prior is not None:
            out[f"{tag}_prior"] = prior.numpy()
    np.savez_compressed(OUT / "g6_ddpm_sample.npz", **out)


def g7_batch_split():
    # the arithmetic of sample_esmdiff.py:104-112 (gibbs; L=len(protseq)) and :181-193 (ddpm; L=#tokens),
    # executed verbatim as an expression of the reference's lines (no import possible: module-level
    # code there loads ESM3 weights).  n_max_residue_square = 200*200*105.
    def split(Lsq_len, num_samples, nmax=200 * 200 * 105):
        bsz = []
        target_size = Lsq_len * Lsq_len * num_samples
        n_batch = target_size // nmax
        residual_size = target_size % nmax
        batch_size = nmax // int(Lsq_len * Lsq_len)
        for _ in range(n_batch):
            bsz.append(batch_size)
        if residual_size > 0:
            bsz.append(num_samples - sum(bsz))
        return bsz

    cases = {}
    for L, N in ((60, 4), (258, 100), (1026, 32), (58, 4), (256, 100), (1024, 32), (258, 800), (60, 10)):
        cases[f"{L},{N}"] = split(L, N)
    (OUT / "g7_batch_split.json").write_text(json.dumps(cases, indent=1))


def g8_merge_pdb():
    a = ("ATOM      1  N   ALA A   1       1.000   2.000   3.000  1.00  0.00           N  \n"
         "ATOM      2  CA  ALA A   1       2.000   2.000   3.000  1.00  0.00           C  \n"
         "ATOM      3  C   ALA A   1       3.000   2.000   3.000  1.00  0.00           C  \n"
         "TER\nEND\n")
    b = ("HEADER    test\n"
         "ATOM      1  N   GLY A   1      -1.000   0.500   3.250  1.00 50.00           N  \n"
         "ATOM      2  CA  GLY A   1      -2.000   0.500   3.250  1.00 50.00           C  \n"
         "HETATM    3  O   HOH A   2       0.000   0.000   0.000  1.00  0.00           O  \n"
         "TER       4      GLY A   1\nEND\n")
    with tempfile.TemporaryDirectory() as d:
        pa, pb = Path(d) / "a.pdb", Path(d) / "b.pdb"
        pa.write_text(a)
        pb.write_text(b)
        merge_pdbfiles([pa, pb], Path(d) / "m.pdb", verbose=False)
        merged = (Path(d) / "m.pdb").read_text()
    (OUT / "g8_merge_pdb.json").write_text(json.dumps({"a": a, "b": b, "merged": merged}, indent=1))


if __name__ == "__main__":
    if sys.argv[1:] == ["g1b"]:          # later addition: only this fixture (the others are unchanged)
        g1b_schedules_other()
        sys.exit(0)
    g1_schedules()
    g1b_schedules_other()
    g2_timestep()
    g3_logits_param()
    g4_categorical()
    g5_ddpm_update()
    g6_ddpm_sample()
    g7_batch_split()
    g8_merge_pdb()
    print("goldens written to", OUT)
    for p in sorted(OUT.iterdir()):
        print(f"  {p.name:28s} {p.stat().st_size:>9d} B")
