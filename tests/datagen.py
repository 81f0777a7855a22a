"""Deterministic input generators shared by oracle/make_golden.py, the tests and bench.py.

Inputs follow SURVEY.md §8(d): block b of size S is bytes [b*S,(b+1)*S) of the infinite cyclic
repetition of a base quality text (q4 / q8 / q40+dir, first column, newlines removed — what
tests/rans4x16.test:11 feeds the reference), plus synthetic edge-case generators.
Only numpy's legacy RandomState is used, so streams are stable across numpy versions.
"""
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BASE_NAMES = ("q4", "q8", "q40+dir", "qvar")
_base_cache = {}


def base_text(name):
    """Stripped base text as a uint8 array (committed under tests/golden/dat/<name>.nl)."""
    if name not in _base_cache:
        with open(os.path.join(GOLDEN, "dat", name + ".nl"), "rb") as f:
            _base_cache[name] = np.frombuffer(f.read(), dtype=np.uint8)
    return _base_cache[name]


def tile(name, size, block=0, offset=None):
    """Block `block` of `size` bytes of the cyclic repetition of base text `name`."""
    t = base_text(name)
    start = (block * size if offset is None else offset) % len(t)
    reps = (start + size + len(t) - 1) // len(t) + 1
    return np.tile(t, reps)[start:start + size].copy()


def const(n, byte=65):
    return np.full(n, byte, dtype=np.uint8)


def rand(n, seed=1, nsym=256, lo=0):
    return (np.random.RandomState(seed).randint(0, nsym, size=n) + lo).astype(np.uint8)


def weighted(n, weights, seed=1):
    """i.i.d. bytes drawn with the given (unnormalised) weights for symbols 0..len-1."""
    w = np.asarray(weights, dtype=np.float64)
    cdf = np.cumsum(w / w.sum())
    u = np.random.RandomState(seed).random_sample(n)
    return np.minimum(np.searchsorted(cdf, u), len(w) - 1).astype(np.uint8)


def runs(n, nsym=6, mean_run=12, seed=1, lo=48):
    """Run-heavy data: geometric run lengths, random symbol per run (exercises X_RLE)."""
    rs = np.random.RandomState(seed)
    out = np.empty(0, dtype=np.uint8)
    while len(out) < n:
        k = max(16, n // mean_run)
        lens = rs.geometric(1.0 / mean_run, size=k)
        syms = (rs.randint(0, nsym, size=k) + lo).astype(np.uint8)
        out = np.concatenate([out, np.repeat(syms, lens)])
    return out[:n].copy()


def markov(n, nsym=40, seed=1, lo=33, stick=0.55):
    """First-order chain: stays near the previous symbol — rewards order-1 modelling."""
    rs = np.random.RandomState(seed)
    steps = rs.randint(-2, 3, size=n)
    jump = rs.random_sample(n) > stick
    far = rs.randint(0, nsym, size=n)
    out = np.empty(n, dtype=np.int64)
    cur = 0
    for i in range(n):                       # small sizes only; use tile() for big inputs
        cur = far[i] if jump[i] else (cur + steps[i]) % nsym
        out[i] = cur
    return (out + lo).astype(np.uint8)


def make(spec):
    """spec = [kind, *args] (JSON-serialisable) -> uint8 array."""
    kind, args = spec[0], spec[1:]
    return {
        "tile": tile, "const": const, "rand": rand, "weighted": weighted,
        "runs": runs, "markov": markov,
        "bytes": lambda hexstr: np.frombuffer(bytes.fromhex(hexstr), dtype=np.uint8).copy(),
    }[kind](*args)
