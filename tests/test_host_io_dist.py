"""CPU tests: SDK / PDB I/O against the reference's data + goldens, the batching heuristic, and the N>1
sharding + gather path on the gloo backend (world_size 2) with the oracle standing in for the GPU compute."""
import json
import os
import socket
import tempfile
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent

BPTI_HEAD = """ATOM      1  N   ARG A   1       4.481  11.293   0.204  1.00  0.00           N  
ATOM      2  CA  ARG A   1       5.137  10.135   0.877  1.00  0.00           C  
ATOM      3  C   ARG A   1       5.356   9.056  -0.221  1.00  0.00           C  
ATOM      4  O   ARG A   1       4.487   8.823  -1.054  1.00  0.00           O  
ATOM     12  N   PRO A   2       6.483   8.417  -0.280  1.00  0.00           N  
ATOM     13  CA  PRO A   2       6.740   7.345  -1.220  1.00  0.00           C  
ATOM     14  C   PRO A   2       5.696   6.227  -1.193  1.00  0.00           C  
ATOM     19  N   ASP A   3       5.540   5.671  -2.406  1.00  0.00           N  
ATOM     20  CA  ASP A   3       4.355   4.870  -2.726  1.00  0.00           C  
TER
END
"""


def test_pdb_reader_and_tokenizer(tmp_path):
    from esmdiff_amd.sdk import ESMProtein, decode_sequence, encode_sequence
    p = tmp_path / "x.pdb"
    p.write_text(BPTI_HEAD)          # first residues of the reference's data/targets/bpti/bpti.pdb
    prot = ESMProtein.from_pdb(p)
    assert prot.sequence == "RPD" and prot.coordinates.shape == (3, 3, 3)
    assert abs(float(prot.coordinates[1, 1, 0]) - 6.740) < 1e-6 and torch.isnan(prot.coordinates[2, 2]).all()
    tok = encode_sequence("RPD_")
    assert tok.tolist() == [0, 10, 14, 13, 32, 2]      # <cls> R P D <mask> <eos>
    assert decode_sequence(tok[1:-1]) == "RPD_"
    out = tmp_path / "y.pdb"
    ESMProtein(sequence="RPD", coordinates=torch.nan_to_num(prot.coordinates)).to_pdb(out)
    assert ESMProtein.from_pdb(out).sequence == "RPD"
    with pytest.raises(ValueError, match="decoder"):
        ESMProtein(sequence="RPD").to_pdb(out)
    # B-factor column = pLDDT, as esm's to_pdb writes it
    ESMProtein(sequence="RPD", coordinates=torch.nan_to_num(prot.coordinates), plddt=torch.tensor([0.25, 0.5, 0.875])).to_pdb(out)
    bf = [float(ln[60:66]) for ln in out.read_text().splitlines() if ln.startswith("ATOM")]
    assert bf[:3] == [0.25] * 3 and bf[-1] == 0.88


def test_infer_oxygen_geometry(tmp_path):
    """Carbonyl O as the reference's decode path adds it (models/utils.py:78-79, esm infer_oxygen; constants from memory,
    so the test pins the chemistry they imply): |C=O| = 1.231 A, CA-C-O = 120.5 deg, O opposite N(+1) across the CA->C
    axis and in the peptide plane; none on the last residue; rigid motions commute; the PDB writer emits it."""
    from esmdiff_amd.pdbio import infer_oxygen, read_pdb_backbone, write_backbone_pdb
    g = np.random.default_rng(3)
    L = 12
    ca = np.cumsum(g.normal(size=(L, 3)) * 2.2, 0)
    xyz = np.stack([ca + g.normal(size=ca.shape) * 0.8, ca, ca + g.normal(size=ca.shape) * 0.8], 1).astype(np.float32)
    o = infer_oxygen(xyz)
    assert o.shape == (L, 3) and np.isnan(o[-1]).all() and np.isfinite(o[:-1]).all()
    c, n1 = xyz[:-1, 2], xyz[1:, 0]
    co, cca, cn = o[:-1] - c, xyz[:-1, 1] - c, n1 - c
    np.testing.assert_allclose(np.linalg.norm(co, axis=-1), 1.2312, atol=2e-3)
    cosang = (co * cca).sum(-1) / (np.linalg.norm(co, axis=-1) * np.linalg.norm(cca, axis=-1))
    np.testing.assert_allclose(np.degrees(np.arccos(cosang)), 120.46, atol=0.1)
    normal = np.cross(cca, cn)
    normal /= np.linalg.norm(normal, axis=-1, keepdims=True)
    assert np.abs((co * normal).sum(-1)).max() < 0.02                      # in the CA-C-N(+1) plane (0.0103 A off)
    e0 = -cca / np.linalg.norm(cca, axis=-1, keepdims=True)
    perp = lambda v: v - e0 * (v * e0).sum(-1, keepdims=True)
    assert ((perp(co) * perp(cn)).sum(-1) < 0).all()                        # O and N(+1) on opposite sides of the axis
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
    np.testing.assert_allclose(infer_oxygen(xyz @ R.T + 5.0)[:-1], o[:-1] @ R.T + 5.0, atol=2e-4)
    write_backbone_pdb(tmp_path / "o.pdb", "A" * L, xyz)
    lines = [ln for ln in (tmp_path / "o.pdb").read_text().splitlines() if ln.startswith("ATOM")]
    assert len(lines) == 4 * L - 1 and sum(ln[12:16].strip() == "O" and ln[76:78].strip() == "O" for ln in lines) == L - 1
    seq, back = read_pdb_backbone(tmp_path / "o.pdb")
    assert seq == "A" * L and np.abs(back - xyz).max() < 1e-3
    write_backbone_pdb(tmp_path / "n.pdb", "A" * L, xyz, with_oxygen=False)
    assert sum(ln.startswith("ATOM") for ln in (tmp_path / "n.pdb").read_text().splitlines()) == 3 * L


def test_load_coords_reads_back_what_the_cli_writes(tmp_path):
    """models/utils.py:274-317 `load_coords`: the evaluation scripts' way from an ensemble on disk to the (n, L, 3) array the
    metrics take — here on this repository's own output: per-sample backbone PDBs merged into one multi-MODEL file."""
    from esmdiff_amd.pdbio import load_coords, merge_pdbfiles, write_backbone_pdb
    rng = np.random.default_rng(0)
    n, L = 7, 12
    seq = "ACDEFGHIKLMN"
    bb = np.cumsum(rng.normal(size=(n, L, 3, 3)) * 0.3 + np.array([1.3, 0.2, 0.1]), axis=1).astype(np.float32)
    for i in range(n):
        write_backbone_pdb(tmp_path / f"s_{i}.pdb", seq, bb[i])
    merge_pdbfiles([tmp_path / f"s_{i}.pdb" for i in range(n)], tmp_path / "ens.pdb", verbose=False)
    ca = load_coords(tmp_path / "ens.pdb", verbose=False)
    assert ca.shape == (n, L, 3) and np.allclose(ca, bb[:, :, 1], atol=6e-4)              # %8.3f records
    full = load_coords(tmp_path / "ens.pdb", ca_only=False, verbose=False)
    assert full.shape == (n, L, 3, 3) and np.allclose(full, bb, atol=6e-4)
    # more models than wanted: uniform stride n // max (the reference's rule: 7 // 3 = 2 -> models 0, 2, 4, 6), or the first ones
    assert np.array_equal(load_coords(tmp_path / "ens.pdb", max_n_model=3, verbose=False), ca[::2])
    assert np.array_equal(load_coords(tmp_path / "ens.pdb", max_n_model=3, uniform_sample=False, verbose=False), ca[:3])
    # a single-model file without MODEL records, a directory of them, a glob pattern, an .npy in nm
    assert load_coords(tmp_path / "s_3.pdb", verbose=False).shape == (1, L, 3)
    d = tmp_path / "dir"
    d.mkdir()
    for i in (1, 4):
        (d / f"m{i}.pdb").write_text((tmp_path / f"s_{i}.pdb").read_text())
    got = load_coords(d, verbose=False)
    assert got.shape == (2, L, 3) and {tuple(np.round(g[0], 3)) for g in got} == {tuple(np.round(bb[i, 0, 1], 3)) for i in (1, 4)}
    assert load_coords(str(tmp_path / "s_*.pdb"), verbose=False).shape == (n, L, 3)
    np.save(tmp_path / "traj.npy", bb[:, :, 1] * 0.1)
    assert np.allclose(load_coords(tmp_path / "traj.npy", verbose=False), bb[:, :, 1], rtol=1e-6)
    with pytest.raises(AssertionError, match="does not exist"):
        load_coords(tmp_path / "nope.pdb")
    (tmp_path / "bad.pdb").write_text((tmp_path / "ens.pdb").read_text().replace(" CA  ASP A   3", " CB  ASP A   3", 1))
    with pytest.raises(ValueError, match="unequal|different lengths"):
        load_coords(tmp_path / "bad.pdb", verbose=False)


def test_load_coords_filters_like_biotite(tmp_path):
    """The reference reads ensembles with biotite (models/utils.py:240-249: get_structure -> filter_backbone): backbone atoms of
    AMINO ACIDS only — HETATM selenomethionine counts, ligand / water / nucleotide atoms named N / CA / C do not — the first
    alternate location seen per residue, and a short or ragged ATOM line is not an IndexError."""
    from esmdiff_amd.pdbio import load_coords

    def atom(rec, serial, name, alt, res, chain, num, x):
        return f"{rec:<6}{serial:>5} {name:<4}{alt}{res:>3} {chain}{num:>4}    {x:8.3f}{0.0:8.3f}{0.0:8.3f}  1.00  0.00\n"

    lines = ["MODEL        1\n"]
    k = 0
    for num, (rec, res) in enumerate([("ATOM", "ALA"), ("HETATM", "MSE"), ("ATOM", "GLY")], start=1):
        for nm in ("N", "CA", "C", "O"):
            k += 1
            lines.append(atom(rec, k, nm, " ", res, "A", num, float(k)))
    # residue 4: altlocs B first, then A -> the B atoms are the ones kept
    for alt, base in (("B", 100.0), ("A", 200.0)):
        for j, nm in enumerate(("N", "CA", "C")):
            k += 1
            lines.append(atom("ATOM", k, nm, alt, "SER", "A", 4, base + j))
    lines.append(atom("HETATM", 90, "CA", " ", " CA", "A", 201, 555.0))      # a calcium ion
    lines.append(atom("HETATM", 91, "N", " ", "HOH", "A", 301, 666.0))
    lines.append(atom("ATOM", 92, "C", " ", " DA", "B", 1, 777.0))           # a nucleotide
    lines.append("ATOM     93  CA  ALA A   9\n")                             # truncated record
    lines.append("ENDMDL\nEND\n")
    p = tmp_path / "mixed.pdb"
    p.write_text("".join(lines))
    full = load_coords(p, ca_only=False, verbose=False)
    assert full.shape == (1, 4, 3, 3)
    assert np.allclose(full[0, :3, :, 0], [[1, 2, 3], [5, 6, 7], [9, 10, 11]])
    assert np.allclose(full[0, 3, :, 0], [100, 101, 102])
    ca = load_coords(p, verbose=False)
    assert ca.shape == (1, 4, 3) and np.allclose(ca[0, :, 0], [2, 6, 10, 101])
    # a residue name outside the reader's amino-acid list that carries a whole backbone is skipped LOUDLY: biotite
    # would keep any CCD "L-peptide linking" residue, so the chain lengths would differ silently otherwise
    extra = "".join(atom("HETATM", 200 + j, nm, " ", "ZZQ", "A", 5, 900.0 + j) for j, nm in enumerate(("N", "CA", "C")))
    q = tmp_path / "odd.pdb"
    q.write_text("".join(lines[:-1]) + extra + "ENDMDL\nEND\n")
    with pytest.warns(UserWarning, match="ZZQ"):
        assert load_coords(q, verbose=False).shape == (1, 4, 3)


def test_merge_pdbfiles_matches_reference_golden(golden_dir, tmp_path):
    from esmdiff_amd.pdbio import merge_pdbfiles
    g = json.loads((golden_dir / "g8_merge_pdb.json").read_text())
    (tmp_path / "a.pdb").write_text(g["a"])
    (tmp_path / "b.pdb").write_text(g["b"])
    merge_pdbfiles([tmp_path / "a.pdb", tmp_path / "b.pdb"], tmp_path / "m.pdb", verbose=False)
    assert (tmp_path / "m.pdb").read_text() == g["merged"]


def test_batch_sizes_match_reference_golden(golden_dir):
    from esmdiff_amd.sample_esmdiff import batch_sizes
    cases = json.loads((golden_dir / "g7_batch_split.json").read_text())
    for key, want in cases.items():
        L, N = (int(v) for v in key.split(","))
        assert batch_sizes(L, N, 200 * 200 * 105) == want
    assert batch_sizes(258, 100) == [100]              # MI355X default: one batch


def test_timer_contract():
    from esmdiff_amd.pdbio import timer
    assert timer(lambda: None)() is None
    r = timer(lambda: [])()
    assert isinstance(r, tuple) and len(r) == 1 and isinstance(r[0], float)


def test_shard_samples_partition():
    from esmdiff_amd.dist import shard_samples
    for n, w in ((800, 8), (10, 4), (3, 8), (100, 1)):
        parts = [shard_samples(n, w, r) for r in range(w)]
        assert sum(c for _, c in parts) == n
        assert all(parts[r][0] == sum(c for _, c in parts[:r]) for r in range(w))


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from esmdiff_amd.dist import gather_ids, shard_samples
    from oracle import c_oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, L = 5, 9
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(1, L, 4104, generator=g).repeat(N, 1, 1).numpy()   # identical rows, noise differs
    off, cnt = shard_samples(N, world, rank)
    x = np.full((cnt, L), 4096, np.int64)
    x = c_oracle.ddpm_step(x, logits[off:off + cnt], 0.6, 0.5, seed=5, sample_offset=off, step=3)
    allids = gather_ids(torch.from_numpy(x), N)
    if rank == 0:
        np.save(Path(tmp) / "gathered.npy", allids.numpy())
    dist.destroy_process_group()


def test_sharded_equals_unsharded_gloo_world2(tmp_path):
    """world_size 2 on gloo: shard -> sample -> gather reproduces the single-process result exactly."""
    import torch.multiprocessing as mp
    from oracle import c_oracle
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(1, 9, 4104, generator=g).repeat(5, 1, 1).numpy()
    want = c_oracle.ddpm_step(np.full((5, 9), 4096, np.int64), logits, 0.6, 0.5, seed=5, sample_offset=0, step=3)
    assert np.array_equal(got, want)
    assert len({tuple(r) for r in got.tolist()}) > 1      # samples differ: the noise is per global sample index


class _StubEngine:
    """Stands in for esmdiff_amd.engine.Engine in the CPU tests of the driver functions: ids are a pure function of the
    GLOBAL sample index (like the Philox noise), so any sharding / chunking must reproduce the same ensemble."""
    max_batch = 2
    device = torch.device("cpu")
    has_geom = False

    @staticmethod
    def ids(n, L, seed, offset):
        b = torch.arange(offset, offset + n)[:, None]
        return (seed * 7 + b * 131 + torch.arange(L)[None] * 17) % 4096

    def gibbs_sample(self, seq, x0, table, temperature, top_p, *, seed, sample_offset=0):
        n, L = x0.shape
        assert n <= self.max_batch
        return torch.where(x0 == 4096, self.ids(n, L, seed, sample_offset), x0)


class _StubModel:
    def __init__(self):
        self.net = _StubEngine()
        self.device = self.net.device

    def ddpm_sample(self, num_steps, sequence_tokens, eps, input_prior, sample_max_t, seed, sample_offset, noise):
        n, L = sequence_tokens.shape
        assert n <= self.net.max_batch, "the driver must chunk to the engine capacity"
        return _StubEngine.ids(n, L, seed, sample_offset)


class _StubDecoder:
    device = torch.device("cpu")
    has_plddt = True
    max_batch = 3

    def decode(self, full, return_plddt=False):
        body = full[:, 1:-1].to(torch.float32)
        coords = body[:, :, None, None] * 1e-3 + torch.arange(9, dtype=torch.float32).view(1, 1, 3, 3)
        return (coords, (body % 100) / 100) if return_plddt else coords


def _cli_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from esmdiff_amd.sample_esmdiff import ddpm_sample_by_esm, minibatch_gibbs_by_esm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seq = "ACDEFGHIKL"
    ddpm_sample_by_esm(seq, _StubModel(), Path(tmp) / "ddpm", "t", num_samples=7, num_steps=3, seed=4, timestamp=False,
                       decoder=_StubDecoder())
    minibatch_gibbs_by_esm(seq, _StubModel(), Path(tmp) / "gibbs", "t", num_samples=7, num_steps=3, seed=4,
                           timestamp=False, decoder=_StubDecoder())
    dist.destroy_process_group()


def test_cli_drivers_shard_decode_gather_gloo_world2(tmp_path):
    """world_size 2, odd sample count, engine capacity 2 (forces chunking), decoder capacity 3: every rank samples and
    DECODES its own shard, one gather per artefact, rank 0 writes the token file and the multi-MODEL PDB in global sample
    order — identical to what a single process computes."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_cli_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    N, L = 7, 10
    for mode, sub, toks in (("ddpm", "step3_eps1e-05_N7", _StubEngine.ids(N, L + 2, 4, 0)[:, 1:-1]),
                            ("gibbs", "T1.4_step3_topp0.9_N7", _StubEngine.ids(N, L + 2, 4, 0)[:, 1:-1])):
        d = tmp_path / mode / sub
        got = np.load(d / "t.tokens.npy")
        assert got.shape == (N, L) and np.array_equal(got, toks.numpy()), mode
        meta = json.loads((d / "t.json").read_text())
        assert meta["world_size"] == 2 and meta["num_samples"] == N
        text = (d / "t.pdb").read_text().splitlines()
        assert sum(ln.startswith("MODEL") for ln in text) == N
        assert sum(ln.startswith("ATOM") and ln[12:16].strip() == "O" for ln in text) == N * (L - 1)   # inferred carbonyl O
        atoms = [ln for ln in text if ln.startswith("ATOM") and ln[12:16].strip() != "O"]
        assert len(atoms) == N * L * 3
        # sample i, residue j: x of atom N = token * 1e-3 + 0; B-factor = (token % 100) / 100
        for i in (0, 3, 6):
            for j in (0, L - 1):
                ln = atoms[(i * L + j) * 3]
                assert abs(float(ln[30:38]) - float(toks[i, j]) * 1e-3) < 2e-3, (mode, i, j)
                assert abs(float(ln[60:66]) - float(toks[i, j] % 100) / 100) < 6e-3


def _cli_worker_800(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from esmdiff_amd.sample_esmdiff import ddpm_sample_by_esm
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _StubModel()
    model.net.max_batch = 128                                   # configs[2]: 100 samples per rank in one batch
    ddpm_sample_by_esm("ACDEFGHIKLMNPQRS", model, Path(tmp) / f"w{world}", "t", num_samples=800, num_steps=3, seed=4, timestamp=False,
                       decoder=_StubDecoder())
    if world > 1:
        dist.destroy_process_group()


def test_cli_configs2_split_world_1_2_8_gloo(tmp_path):
    """BASELINE configs[2]'s split without the node: 800 samples through the reference's driver (sample_esmdiff.py:137-233)
    at world sizes 1, 2 and 8 on gloo with the stand-in engine (ids a pure function of the GLOBAL sample index, like the Philox
    noise) — 100 samples per rank at world 8, one int16 all_gather, every rank decodes its shard, rank 0 writes ONE file with 800
    MODELs.  The token files and the PDB files of the three runs must be identical byte for byte."""
    import torch.multiprocessing as mp
    outs = {}
    for world in (1, 2, 8):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        if world == 1:
            _cli_worker_800(0, 1, port, str(tmp_path))
            for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                os.environ.pop(k, None)
        else:
            mp.spawn(_cli_worker_800, args=(world, port, str(tmp_path)), nprocs=world, join=True)
        d = tmp_path / f"w{world}" / "step3_eps1e-05_N800"
        toks = np.load(d / "t.tokens.npy")
        pdb = (d / "t.pdb").read_bytes()
        meta = json.loads((d / "t.json").read_text())
        assert toks.shape == (800, 16) and meta["world_size"] == world and pdb.count(b"MODEL ") == 800
        assert len(list(d.glob("*.pdb"))) == 1                  # one multi-MODEL file, written by rank 0 only
        outs[world] = (toks, pdb)
    want = _StubEngine.ids(800, 18, 4, 0)[:, 1:-1].numpy()
    for world in (1, 2, 8):
        assert np.array_equal(outs[world][0], want), world
        assert outs[world][1] == outs[1][1], world


def _fanout_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from esmdiff_amd.dist import broadcast_state_dict
    from esmdiff_amd.weights import load_checkpoint_state_dict
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def loader():
        calls.append(rank)
        return load_checkpoint_state_dict(Path(tmp) / "ck.pt")

    sd, t = broadcast_state_dict(loader, "cpu")
    assert calls == ([0] if rank == 0 else []), calls           # only the reader opened the file
    assert t["rank"] == rank and t["reader_rank"] == 0 and t["bytes"] >= 4 * (7 * 5 + 3) + 2 * 4 and t["load_s"] >= 0
    torch.save({k: v.clone() for k, v in sd.items()}, Path(tmp) / f"got{rank}.pt")
    dist.destroy_process_group()


def test_checkpoint_fanout_one_reader_gloo_world3(tmp_path):
    """dist.broadcast_state_dict: rank 0 reads the checkpoint (checkpoint_utils.py:59-64's file format), the other ranks never
    touch it and receive names, shapes, dtypes and values — mixed dtypes, a 0-d tensor, an odd-sized one (256-byte aligned flat
    buffer) — bit for bit."""
    import torch.multiprocessing as mp
    g = torch.Generator().manual_seed(0)
    sd = {"net.a.weight": torch.randn(7, 5, generator=g), "net.b.bias": torch.randn(3, generator=g).to(torch.bfloat16),
          "sigma_embedder.mlp.0.weight": torch.randn(2, 2, generator=g), "net.count": torch.tensor(5, dtype=torch.int64),
          "net.half": torch.randn(129, generator=g).half()}
    torch.save({"module": sd}, tmp_path / "ck.pt")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_fanout_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        got = torch.load(tmp_path / f"got{r}.pt", weights_only=True)
        assert list(got) == list(sd)
        for k in sd:
            assert got[k].dtype == sd[k].dtype and got[k].shape == sd[k].shape and torch.equal(got[k], sd[k]), (r, k)
    from esmdiff_amd.dist import broadcast_state_dict
    alone, t = broadcast_state_dict(lambda: sd, "cpu")          # no process group: loader + copy
    assert t["path"] == "no process group" and all(torch.equal(alone[k], sd[k]) for k in sd)


def _fanout_fail_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    import torch.distributed as dist
    from esmdiff_amd.dist import broadcast_state_dict
    from esmdiff_amd.weights import load_checkpoint_state_dict
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        broadcast_state_dict(lambda: load_checkpoint_state_dict(Path(tmp) / "nothing_here.pt"), "cpu")
        msg = "no error"
    except RuntimeError as ex:
        msg = str(ex)
    (Path(tmp) / f"err{rank}.txt").write_text(msg)
    dist.barrier()                                              # every rank is still alive and in step
    dist.destroy_process_group()


def test_checkpoint_fanout_reader_failure_reaches_every_rank(tmp_path):
    """loading is a collective, so a failure of the ONE reader (missing file, no 'module' entry ...) must fail every
    rank — the others would otherwise sit in the broadcast until the backend's timeout.  gloo, world 2: both ranks raise within
    seconds with the reader's error text."""
    import time
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    t0 = time.time()
    mp.spawn(_fanout_fail_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert time.time() - t0 < 45
    for r in range(2):
        msg = (tmp_path / f"err{r}.txt").read_text()
        assert "checkpoint loading failed on rank 0" in msg and "nothing_here.pt" in msg, (r, msg)
    from esmdiff_amd.dist import broadcast_state_dict
    with pytest.raises(FileNotFoundError):                      # without a process group the loader's own exception comes through
        broadcast_state_dict(lambda: torch.load(tmp_path / "nothing_here.pt"), "cpu")


def test_engine_capacity_covers_every_issued_batch():
    """max_batch must come from the batches the splitters really issue (per target, per mode, remainder batch
    included), and any batch can be chunked to a smaller engine."""
    from esmdiff_amd.sample_esmdiff import DEFAULT_NMAX, batch_sizes, engine_capacity
    assert engine_capacity([100, 500], 1000, DEFAULT_NMAX, "ddpm") == 1000      # the 100-residue target is issued as one batch
    assert engine_capacity([200], 1000, DEFAULT_NMAX, "gibbs") == max(batch_sizes(200, 1000, DEFAULT_NMAX)) == 842
    assert engine_capacity([200], 1000, DEFAULT_NMAX, "ddpm") == max(batch_sizes(202, 1000, DEFAULT_NMAX))
    nmax = 200 * 200 * 105
    for n_tok, n in ((60, 4), (258, 100), (258, 1000), (1026, 33)):
        ref = batch_sizes(n_tok, n, nmax)
        for cap in (1, 7, 64):
            got = batch_sizes(n_tok, n, nmax, cap)
            assert sum(got) == n and max(got) <= cap
        assert batch_sizes(n_tok, n, nmax, max(ref)) == ref                      # a big enough engine: the reference's split


def test_hydra_config_next_to_checkpoint(tmp_path):
    """checkpoint_utils.py:45-57: the run's .hydra/config.yaml decides noise schedule / time_conditioning / head width."""
    from esmdiff_amd.model import config_from_hydra_yaml
    from esmdiff_amd.schedule import CosineNoise, LogLinearNoise
    from esmdiff_amd.weights import checkpoint_file_and_config
    run = tmp_path / "run"
    (run / ".hydra").mkdir(parents=True)
    (run / "checkpoints").mkdir()
    ck = run / "checkpoints" / "last.pt"
    ck.write_bytes(b"")
    assert checkpoint_file_and_config(ck) == (ck, None)
    (run / ".hydra" / "config.yaml").write_text(
        "model:\n  noise_schedule:\n    _target_: slm.utils.noise_utils.CosineNoise\n    eps: 0.002\n"
        "  time_conditioning: false\n  net:\n    n_structure_heads: 4101\n")
    f, y = checkpoint_file_and_config(ck)
    assert f == ck and y == run / ".hydra" / "config.yaml"
    cfg, noise = config_from_hydra_yaml(y)
    assert isinstance(noise, CosineNoise) and noise.eps == 0.002 and cfg.time_conditioning is False
    (run / ".hydra" / "config.yaml").write_text("model:\n  noise_schedule:\n    _target_: slm.utils.noise_utils.LogLinearNoise\n")
    cfg, noise = config_from_hydra_yaml(y)
    assert isinstance(noise, LogLinearNoise) and cfg.time_conditioning is True
    # the other schedule classes of noise_utils.py (constructor keys from the yaml, as hydra passes them)
    from esmdiff_amd.schedule import CosineSqrNoise, GeometricNoise, Linear
    (run / ".hydra" / "config.yaml").write_text(
        "model:\n  noise_schedule:\n    _target_: slm.utils.noise_utils.GeometricNoise\n    sigma_min: 0.01\n    sigma_max: 2\n")
    cfg, noise = config_from_hydra_yaml(y)
    assert isinstance(noise, GeometricNoise) and [round(float(v), 4) for v in noise.sigmas] == [0.01, 2.0]
    (run / ".hydra" / "config.yaml").write_text("model:\n  noise_schedule:\n    _target_: slm.utils.noise_utils.Linear\n    sigma_max: 8\n")
    cfg, noise = config_from_hydra_yaml(y)
    assert isinstance(noise, Linear) and float(noise.sigma_min) == 0.0 and float(noise.sigma_max) == 8.0
    (run / ".hydra" / "config.yaml").write_text("model:\n  noise_schedule:\n    _target_: slm.utils.noise_utils.CosineSqrNoise\n")
    assert isinstance(config_from_hydra_yaml(y)[1], CosineSqrNoise)
    (run / ".hydra" / "config.yaml").write_text("model:\n  noise_schedule:\n    _target_: slm.utils.noise_utils.SigmoidNoise\n")
    with pytest.raises(NotImplementedError):
        config_from_hydra_yaml(y)
    with pytest.raises(FileNotFoundError):
        checkpoint_file_and_config(tmp_path / "nope.pt")
    with pytest.raises(ValueError):
        (tmp_path / "x.bin").write_bytes(b"")
        checkpoint_file_and_config(tmp_path / "x.bin")


def test_checkpoint_loader_needs_module_dict(tmp_path):
    from esmdiff_amd.weights import load_checkpoint_state_dict
    torch.save({"net.a": torch.zeros(1)}, tmp_path / "flat.pt")
    with pytest.raises(KeyError, match="module"):
        load_checkpoint_state_dict(tmp_path / "flat.pt")
    torch.save({"module": {"foo": torch.zeros(1)}}, tmp_path / "other.pt")
    with pytest.raises(KeyError, match="net"):
        load_checkpoint_state_dict(tmp_path / "other.pt")
    torch.save({"module": {"net.x": torch.ones(2)}, "ds_version": "0.1"}, tmp_path / "ok.pt")
    assert list(load_checkpoint_state_dict(tmp_path / "ok.pt")) == ["net.x"]
    import functools
    torch.save({"module": {"net.x": torch.ones(2)}, "hyper_parameters": functools.partial(print)}, tmp_path / "obj.pt")
    with pytest.raises(RuntimeError, match="Re-export"):
        load_checkpoint_state_dict(tmp_path / "obj.pt")


def test_cli_argparser_matches_reference_defaults():
    from esmdiff_amd.sample_esmdiff import get_argparser
    a = get_argparser([])
    assert (a.input, a.ckpt, a.output, a.mode, a.num_steps, a.num_samples, a.mask_ids) == (
        "data/targets/bpti", None, "output/inference_esmdiff", "gibbs", 25, 10, None)


def _run_bench(extra, timeout=600):
    import json
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # ONE JSON line whatever the number of ranks
    return json.loads(lines[0])


def test_bench_self_spawns_ranks_world2_gloo_stub():
    """`python bench.py --gpus 2` with NO launcher (the form the driver used for --gpus 1): the script re-executes itself under
    torch.distributed.run, both ranks join a process group, shard by global sample index, all_gather the int16 ids, take the
    max-over-ranks time, and rank 0 prints ONE JSON line.  CPU stand-in engine + gloo here; the GPU suite runs the same
    branch with the real engine over RCCL (tests/test_gpu_dist.py)."""
    out = _run_bench(["--gpus", "2", "--stub-engine", "--tiny", "--steps", "2", "--warmup", "1", "--samples-per-gpu", "3",
                      "--residues", "10"])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["data"] == "debug-stub-engine" and out["roofline"] is None
    assert "sample-sharded x2" in out["config"]["parallelism"]
    # whole-job value: samples of BOTH ranks / max-over-ranks time
    assert abs(out["value"] - 2 * 3 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-3
    # the N > 1 line explains its own efficiency: one record per rank with engine-create time, sampling and
    # gather time per step, power, NUMA pinning; the slowest rank is named; the environment the number was taken in is on the line
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and out["slowest_rank"] in (0, 1)
    for r in out["per_rank"]:
        assert r["create_s"] >= 0 and r["sample_ms_per_step"] > 0 and r["gather_ms_per_step"] >= 0 and r["elapsed_s"] > 0
        assert "power_mean_w" in r and "numa" in r
    assert out["environment"]["esmdiff_env"] == {k: v for k, v in os.environ.items() if k.startswith("ESMDIFF_")}
    assert isinstance(out["timed_region"], str) and "all_gather" in out["timed_region"]


def test_bench_fails_fast_and_refuses_debug_switches(monkeypatch):
    """`--gpus N` with fewer visible GPUs ends in seconds with one line (no launcher started); ESMDIFF_DEBUG_SKIP in the
    environment — the launch-skipping switch of -DED_DEBUG builds — makes bench.py refuse to measure at all."""
    import subprocess
    import sys
    import time
    root = Path(__file__).resolve().parent.parent
    t0 = time.time()
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=120, cwd=root)
    assert r.returncode != 0 and "--gpus 64 but" in r.stderr and time.time() - t0 < 30, r.stderr[-500:]
    assert len([ln for ln in r.stderr.splitlines() if ln.strip()]) == 1
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--stub-engine", "--tiny"], capture_output=True, text=True, timeout=120,
                       cwd=root, env=dict(os.environ, ESMDIFF_DEBUG_SKIP="3"))
    assert r.returncode != 0 and "ESMDIFF_DEBUG_SKIP" in r.stderr


def test_bench_refuses_mismatched_world(monkeypatch):
    """Under a launcher the world size must equal --gpus (a silent mismatch would mis-report n_gpus)."""
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--stub-engine", "--tiny"], capture_output=True,
                       text=True, timeout=120, env=env, cwd=root)
    assert r.returncode != 0 and "--nproc-per-node 2" in r.stderr
