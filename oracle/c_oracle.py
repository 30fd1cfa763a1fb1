"""TEST INFRASTRUCTURE — ctypes access to oracle/build/liboracle_sampler.so (oracle/csrc/sampler_oracle.c)."""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "build" / "liboracle_sampler.so"
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> Path:
    src = _HERE / "csrc" / "sampler_oracle.c"
    hdr = _HERE.parent / "esmdiff_amd" / "csrc" / "ed_math.h"
    stale = (not _LIB.exists()) or force
    if not stale:
        try:
            stale = _LIB.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime)
        except FileNotFoundError:
            stale = False
    if stale:
        subprocess.run(["make", "-C", str(_HERE)] + (["-B"] if force else []), check=True,
                       capture_output=True)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_LIB))
        _lib.oracle_ddpm_step.argtypes = [_i64p, _f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_float, ctypes.c_int, _f32p, ctypes.c_uint64,
                                          ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.oracle_ddpm_step.restype = None
        _lib.oracle_logits_parameterization.argtypes = [_f32p, ctypes.c_int64, _i64p, ctypes.c_int64,
                                                        ctypes.c_int, _f32p]
        _lib.oracle_logits_parameterization.restype = None
        for nm in ("oracle_expf_array", "oracle_logf_array"):
            getattr(_lib, nm).argtypes = [_f32p, _f32p, ctypes.c_int64]
            getattr(_lib, nm).restype = None
        _lib.oracle_philox.argtypes = [ctypes.c_uint32] * 6 + [ctypes.POINTER(ctypes.c_uint32)]
        _lib.oracle_philox.restype = None
        _lib.oracle_philox_uniforms.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                                ctypes.c_uint32, ctypes.c_int, _f32p]
        _lib.oracle_philox_uniforms.restype = None
        _lib.oracle_gibbs_step.argtypes = [_i64p, _i64p, _f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                           ctypes.POINTER(ctypes.c_int32), _f32p, ctypes.c_uint64, ctypes.c_uint64,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p,
                                           ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
        _lib.oracle_gibbs_step.restype = None
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def ddpm_step(x, logits, mc_t, mc_s, final=False, u=None, seed=0, sample_offset=0, step=0, vocab=4101):
    """x [B,L] int64 (copied), logits [B,L,ld>=vocab] f32 -> new x.  u [B,L,vocab] or None (Philox)."""
    x = np.array(x, dtype=np.int64, order="C", copy=True)
    B, L = x.shape
    lg, lgp = _f32(logits)
    assert lg.shape[:2] == (B, L) and lg.shape[2] >= vocab
    up = None
    if u is not None:
        ua, up = _f32(u)
        assert ua.shape == (B, L, vocab)
    lib().oracle_ddpm_step(x.ctypes.data_as(_i64p), lgp, lg.shape[2], vocab, np.float32(mc_t),
                           np.float32(mc_s), int(bool(final)), up, int(seed), int(sample_offset), int(step),
                           B, L)
    return x


def logits_parameterization(logits, x, vocab=4101):
    x = np.ascontiguousarray(x, dtype=np.int64)
    lg, lgp = _f32(logits)
    R = x.size
    out = np.empty((R, vocab), dtype=np.float32)
    lib().oracle_logits_parameterization(lgp, lg.shape[-1], x.ctypes.data_as(_i64p), R, vocab,
                                         out.ctypes.data_as(_f32p))
    return out.reshape(x.shape + (vocab,))


def expf(x):
    a, ap = _f32(x)
    y = np.empty_like(a)
    lib().oracle_expf_array(ap, y.ctypes.data_as(_f32p), a.size)
    return y


def logf(x):
    a, ap = _f32(x)
    y = np.empty_like(a)
    lib().oracle_logf_array(ap, y.ctypes.data_as(_f32p), a.size)
    return y


def philox(c, k):
    out = (ctypes.c_uint32 * 4)()
    lib().oracle_philox(*[int(v) for v in c], *[int(v) for v in k], out)
    return [int(v) for v in out]


def philox_uniforms(seed, sample, step, l, vocab=4101):
    out = np.empty(vocab, dtype=np.float32)
    lib().oracle_philox_uniforms(int(seed), int(sample), int(step), int(l), vocab, out.ctypes.data_as(_f32p))
    return out


def gibbs_step(x, seq, logits, temperature, top_p, n_unmask, u=None, seed=0, sample_offset=0, step=0,
               return_aux=False, vocab=4096, strategy="entropy", invalid_ids=()):
    """One entropy-ordered unmasking step (oracle_gibbs_step).  x, seq [B,L]; logits [B,L,ld>=vocab]; `vocab` = width of
    the structure head's row (4096 stock ESM3, 4101 ESMDiff): entropy and nucleus run over all of it, draws over the
    4096 codebook ids; n_unmask [B]; u [B,L,4096] or None (Philox)."""
    assert 4096 <= vocab <= 4352
    x = np.array(x, dtype=np.int64, order="C", copy=True)
    seq = np.ascontiguousarray(seq, dtype=np.int64)
    B, L = x.shape
    lg, lgp = _f32(logits)
    nu = np.ascontiguousarray(n_unmask, dtype=np.int32)
    up = None
    if u is not None:
        ua, up = _f32(u)
        assert ua.shape == (B, L, 4096)
    ent = np.full((B, L), np.inf, dtype=np.float32)
    smp = np.full((B, L), -1, dtype=np.int32)
    assert lg.shape[2] >= vocab
    assert strategy in ("entropy", "random") and (strategy == "entropy" or u is None)
    mask = np.zeros(128, dtype=np.uint32)
    for v in invalid_ids:
        if 0 <= int(v) < 4096:
            mask[int(v) >> 5] |= np.uint32(1 << (int(v) & 31))
    lib().oracle_gibbs_step(x.ctypes.data_as(_i64p), seq.ctypes.data_as(_i64p), lgp, lg.shape[2], int(vocab),
                            np.float32(temperature), np.float32(top_p), nu.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                            up, int(seed), int(sample_offset), int(step), B, L, ent.ctypes.data_as(_f32p),
                            smp.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 1 if strategy == "random" else 0,
                            mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)) if len(invalid_ids) else None)
    return (x, ent, smp) if return_aux else x
