"""PDB text I/O used by the sampling CLI.

  read_pdb_backbone   what the CLI needs from ESMProtein.from_pdb (/root/reference/slm/sample_esmdiff.py:278-283):
                      the one-letter sequence of the first chain and its N/CA/C coordinates
  merge_pdbfiles      /root/reference/slm/utils/eval_utils.py:437-492 — one multi-MODEL file, 80-column lines
  timer               /root/reference/slm/utils/eval_utils.py:24-34
"""
from __future__ import annotations

import time
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

THREE_TO_ONE = {
    "ALA": "A", "ARG": "R", "ASN": "N", "ASP": "D", "CYS": "C", "GLN": "Q", "GLU": "E", "GLY": "G", "HIS": "H",
    "ILE": "I", "LEU": "L", "LYS": "K", "MET": "M", "PHE": "F", "PRO": "P", "SER": "S", "THR": "T", "TRP": "W",
    "TYR": "Y", "VAL": "V", "SEC": "U", "PYL": "O",
}
ONE_TO_THREE = {v: k for k, v in THREE_TO_ONE.items()}


def read_pdb_backbone(path, chain: Optional[str] = None) -> Tuple[str, np.ndarray]:
    """-> (sequence, coords [L,3,3] for N, CA, C; NaN where an atom is missing).  First MODEL, first chain."""
    seq: List[str] = []
    coords: List[np.ndarray] = []
    index = {}
    want = {"N": 0, "CA": 1, "C": 2}
    with open(path) as fh:
        for line in fh:
            rec = line[:6]
            if rec.startswith("ENDMDL"):
                break
            if not rec.startswith("ATOM"):
                if rec.startswith("TER") and seq:
                    break
                continue
            ch = line[21]
            if chain is None:
                chain = ch
            if ch != chain:
                continue
            if line[16] not in (" ", "A"):
                continue
            key = (line[22:27], line[17:20])
            if key not in index:
                index[key] = len(seq)
                seq.append(THREE_TO_ONE.get(line[17:20].strip(), "X"))
                coords.append(np.full((3, 3), np.nan, dtype=np.float32))
            name = line[12:16].strip()
            if name in want:
                coords[index[key]][want[name]] = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
    if not seq:
        raise ValueError(f"no ATOM records in {path}")
    return "".join(seq), np.stack(coords)


_O_LOCAL = np.array([0.6240, -1.0613, 0.0103], dtype=np.float64)   # [ESM-RECALL: esm ProteinChain.infer_oxygen]


def infer_oxygen(coords: np.ndarray) -> np.ndarray:
    """Carbonyl O from N/CA/C, as the reference's decode path places it (`chain.infer_oxygen()`,
    /root/reference/slm/models/utils.py:78-79; esm ProteinChain.infer_oxygen): residue i's O is a fixed vector in the
    Gram-Schmidt frame with origin C_i, x axis CA_i -> C_i and the N_{i+1} side of the plane as +y; the last residue has
    no following N and gets NaN.  coords [L, >=3, 3] (N, CA, C first) -> [L, 3].
    The local vector gives |C=O| = 1.231 A, CA-C-O = 120.5 deg, O-C-N(+1) = 123.5 deg in the peptide plane."""
    xyz = np.asarray(coords, dtype=np.float64)
    L = xyz.shape[0]
    out = np.full((L, 3), np.nan)
    if L < 2:
        return out.astype(np.float32)
    ca, c, n_next = xyz[:-1, 1], xyz[:-1, 2], xyz[1:, 0]
    with np.errstate(invalid="ignore", divide="ignore"):
        e0 = c - ca
        e0 = e0 / np.linalg.norm(e0, axis=-1, keepdims=True)
        v = n_next - c
        e1 = v - e0 * np.sum(e0 * v, axis=-1, keepdims=True)
        e1 = e1 / np.linalg.norm(e1, axis=-1, keepdims=True)
        e2 = np.cross(e0, e1)
        out[:-1] = c + _O_LOCAL[0] * e0 + _O_LOCAL[1] * e1 + _O_LOCAL[2] * e2
    return out.astype(np.float32)


def write_backbone_pdb(path, sequence: str, coords: np.ndarray, bfactor: Optional[np.ndarray] = None,
                       with_oxygen: bool = True) -> None:
    """N/CA/C backbone (+ the inferred carbonyl O, as the reference's decoded chains carry it) as ATOM records."""
    lines, serial = [], 1
    coords = np.asarray(coords)
    if coords.ndim == 3:
        coords = coords[:, :3]       # N, CA, C only: a 4th / 5th input column (atom37: CB, O) is never written under another name
    n_atoms = 3
    if with_oxygen and coords.ndim == 3 and coords.shape[0] == len(sequence) and coords.shape[0] > 0:
        coords = np.concatenate([coords, infer_oxygen(coords)[:, None]], axis=1)
        n_atoms = 4                  # the O column exists only when it was inferred here
    atoms = (("N", "N"), ("CA", "C"), ("C", "C"), ("O", "O"))[:n_atoms]
    for i, aa in enumerate(sequence):
        for j, (name, elem) in enumerate(atoms):
            x, y, z = (float(v) for v in coords[i, j])
            if not np.isfinite([x, y, z]).all():
                continue
            b = 0.0 if bfactor is None else float(bfactor[i])
            lines.append(f"ATOM  {serial:5d}  {name:<3s} {ONE_TO_THREE.get(aa, 'UNK'):>3s} A{i + 1:4d}    "
                         f"{x:8.3f}{y:8.3f}{z:8.3f}{1.0:6.2f}{b:6.2f}          {elem:>2s}  ")
            serial += 1
    lines += ["TER", "END"]
    Path(path).write_text("\n".join(lines) + "\n")


def merge_pdbfiles(input, save_to: Path, verbose: bool = True) -> None:
    """Ordered merge of single- or multi-model PDB files into one multi-MODEL file."""
    if isinstance(input, Path):
        pdb_files = [f for f in input.iterdir() if f.suffix == ".pdb"]
    elif isinstance(input, (list, tuple)):
        pdb_files = list(input)
    else:
        raise ValueError(f"Unrecognized input type: {type(input)}")
    assert len(pdb_files) > 0
    save_to = Path(save_to)
    save_to.parent.mkdir(parents=True, exist_ok=True)
    n_model, out = 0, []
    for f in pdb_files:
        lines = Path(f).read_text().splitlines()
        if not any(ln.startswith(("MODEL", "ENDMDL")) for ln in lines):
            n_model += 1
            out.append(f"MODEL     {n_model}")
            out.extend(ln.strip() for ln in lines if ln.startswith(("TER", "ATOM")))
            out.append("ENDMDL")
            continue
        for ln in lines:
            if ln.startswith("MODEL"):
                n_model += 1
                if n_model > 1:
                    out.append("ENDMDL")
                out.append(f"MODEL     {n_model}")
            elif ln.startswith("END"):
                continue
            elif ln.startswith(("TER", "ATOM")):
                out.append(ln.strip())
    out += ["ENDMDL", "END"]
    save_to.write_text("\n".join(ln.ljust(80) for ln in out) + "\n")
    if verbose:
        print(f"Merged {len(pdb_files)} PDB files into {save_to} with {n_model} models.")


# residue names biotite's filter_amino_acids accepts that occur in practice: the 20 standard ones, SEC / PYL, the ambiguity codes
# and the common modified L-peptide residues (the full list is the Chemical Component Dictionary's "L-peptide linking" class)
_AMINO_ACID_RESNAMES = frozenset("""ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL SEC PYL ASX GLX UNK
MSE HYP SEP TPO PTR CSO CSD CME CSX OCS KCX LLP MLY M3L MLZ ALY PCA NLE ABA AIB ORN DAL HIC HID HIE HIP CYX ASH GLH LYN""".split())


def _backbone_coords_from_pdb(pdb_path, target_atoms=("N", "CA", "C")) -> np.ndarray:
    """models/utils.py:240-267 without biotite: the backbone atoms named in `target_atoms` of every MODEL of a PDB file (a file
    without MODEL records is one model) -> (n_models, L, len(target_atoms), 3), or (n_models, L, 3) for a single atom name.
    ATOM records only, first alternate location; every model must hold the same number of each atom."""
    want = {a: j for j, a in enumerate(target_atoms)}
    models, cur = [], [[] for _ in target_atoms]

    def close():
        if any(cur):
            n = {len(c) for c in cur}
            if len(n) != 1:
                raise ValueError(f"{pdb_path}: a model with unequal numbers of {list(target_atoms)} atoms ({[len(c) for c in cur]})")
            arr = np.stack([np.asarray(c, dtype=np.float32) for c in cur], axis=1)        # L, na, 3
            models.append(arr[:, 0] if len(target_atoms) == 1 else arr)
            for c in cur:
                c.clear()

    # biotite's reader as the reference uses it (models/utils.py:240-249: PDBFile.get_structure() -> struct.filter_backbone): N / CA /
    # C atoms of AMINO-ACID residues, ATOM and HETATM records alike (selenomethionine and other modified residues are HETATM),
    # nothing of ligands, waters or nucleic acids; of alternate locations the first one seen in each residue (altloc="first").
    first_alt = {}
    skipped = {}       # residues outside the amino-acid list that nevertheless carry backbone atoms: (chain, number, name) -> atom names
    with open(pdb_path) as fh:
        for line in fh:
            name = line[:6].strip()
            if name in ("MODEL", "ENDMDL"):
                close()
                first_alt.clear()
            elif name in ("ATOM", "HETATM") and len(line) >= 54 and line[17:20].strip() not in _AMINO_ACID_RESNAMES:
                if line[12:16].strip() in ("N", "CA", "C"):
                    skipped.setdefault((line[21], line[22:27], line[17:20].strip()), set()).add(line[12:16].strip())
            elif name in ("ATOM", "HETATM") and len(line) >= 54:
                alt = line[16]
                if alt != " ":
                    res = (line[21], line[22:27])                      # chain, residue number + insertion code
                    if first_alt.setdefault(res, alt) != alt:
                        continue
                j = want.get(line[12:16].strip())
                if j is not None:
                    cur[j].append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
    close()
    odd = sorted({k[2] for k, atoms in skipped.items() if atoms >= {"N", "CA", "C"}})
    if odd:      # biotite's filter_amino_acids knows the whole CCD "L-peptide linking" class; this reader a hand-written list
        import warnings
        warnings.warn(f"{pdb_path}: residues {odd} carry N / CA / C atoms but are not in this reader's amino-acid list and were skipped — "
                      "the chain is shorter than biotite (the reference's reader) would make it; add the residue name to "
                      "esmdiff_amd.pdbio._AMINO_ACID_RESNAMES if it is a modified amino acid", stacklevel=2)
    if not models:
        raise ValueError(f"no backbone ATOM records in {pdb_path}")
    if len({m.shape for m in models}) != 1:
        raise ValueError(f"{pdb_path}: models of different lengths {sorted({m.shape[0] for m in models})}")
    return np.stack(models, axis=0)


def load_coords(input_path, max_n_model: Optional[int] = 10000, uniform_sample: bool = True, ca_only: bool = True,
                verbose: bool = True) -> np.ndarray:
    """models/utils.py:274-317: the ensemble behind a path as an array for the metrics — a (multi-MODEL) .pdb, a .npy in nm
    (scaled to Angstrom), a directory of .pdb files, or a glob pattern; CA only -> (n, L, 3), else N / CA / C -> (n, L, 3, 3).
    More than max_n_model models: every (n // max_n_model)-th one (uniform_sample) or the first max_n_model."""
    import glob as _glob
    import os
    assert os.path.exists(input_path) or _glob.glob(str(input_path)), f"File {input_path} does not exist."
    input_path = Path(input_path)
    atoms = ("CA",) if ca_only else ("N", "CA", "C")
    if input_path.name.endswith(".pdb") and input_path.exists():
        coords = _backbone_coords_from_pdb(input_path, atoms)
    elif input_path.name.endswith(".npy"):
        coords = np.load(input_path) / 0.1                    # nm -> Angstrom (coordinate_scale, models/utils.py:270-271)
    elif input_path.is_dir():
        coords = np.concatenate([_backbone_coords_from_pdb(f, atoms) for f in input_path.iterdir() if f.name.endswith(".pdb")], axis=0)
    else:
        # (the reference asserts os.path.exists first, which no pattern passes; here a pattern that matches files is accepted)
        print("[Warning] Unrecoginzed path, infer as glob pattern")
        coords = np.concatenate([_backbone_coords_from_pdb(f, atoms) for f in sorted(_glob.glob(str(input_path))) if f.endswith(".pdb")], axis=0)
    if max_n_model is not None and len(coords) > max_n_model > 0:
        coords = coords[::len(coords) // max_n_model] if uniform_sample else coords[:max_n_model]
    if verbose:
        print(f"Loaded {len(coords)} models from input: {input_path} (shape={coords.shape})")
    return coords


def split_pdbfile(input, output_dir=None, sep: str = "_", verbose: bool = True):
    """eval_utils.py:495-530, the inverse of merge_pdbfiles: one PDB string per MODEL block — its ATOM / TER records, closed by
    'END' — and, with output_dir, one file `<stem><sep><i>.pdb` per block.  Returns the list of strings."""
    input = Path(input)
    assert input.exists() and input.suffix == ".pdb", f"File {input} does not exist or not a .pdb file."
    blocks, records = [], []
    for line in input.read_text().splitlines(keepends=True):
        name = line[:6].strip()                      # PDB record name, columns 1-6
        if name == "MODEL":
            records = []
        elif name in ("ATOM", "TER"):
            records.append(line)
        elif name in ("ENDMDL", "END"):
            if records:                              # (the file's last END follows an ENDMDL: nothing pending)
                blocks.append("".join(records) + "END\n")
                records = []
        elif verbose:
            print(f"Warning: line '{line}' is not recognized. Skip.")
    if output_dir is not None:
        output_dir = Path(output_dir)
        output_dir.mkdir(parents=True, exist_ok=True)
        for i, text in enumerate(blocks):
            (output_dir / f"{input.stem}{sep}{i}.pdb").write_text(text)
    if verbose:
        print(f">>> Split pdb {input} into {len(blocks)} structures.")
    return blocks


def timer(func):
    """Prints the elapsed time and appends it to a tuple result; a None result passes through."""
    def wrapper(*args, **kwargs):
        t0 = time.time()
        result = func(*args, **kwargs)
        if result is None:
            return None
        dt = time.time() - t0
        print(f"Elapsed time ({func.__name__}): {dt:.2f} sec")
        return (*result, float(f"{dt:.2f}"))
    return wrapper
