"""Deterministic inputs for the golden fixtures (tests/golden/flat_golden.npz).

Used by oracle/make_golden.py (to feed the real reference) and by the tests (to re-create the
same inputs; every array's sha256 is checked against the fixture before use).  Only numpy's
legacy RandomState and elementwise float32 arithmetic are used, so the streams are stable.
"""
import hashlib

import numpy as np

DOT_DIMS = (768, 384, 300, 100, 37, 31, 16, 15, 13, 8, 7, 5, 3)

# name -> shape of the flat-scan case.  "dup": rows duplicated so exact-score ties exist.
CASES = {
    "main768": dict(n=3000, d=768, nq=8, k=10, seed=11, gtbin=True),
    "tail100": dict(n=700, d=100, nq=4, k=7, seed=12),          # dim % 8 != 0 and % 16 != 0 tails
    "ties64": dict(n=600, d=64, nq=4, k=10, seed=13, dup=True),   # exact duplicate rows
    "tiny": dict(n=5, d=32, nq=2, k=10, seed=14),                 # k > N -> clamp
    "k1": dict(n=257, d=128, nq=3, k=1, seed=15),
    "k64": dict(n=1500, d=256, nq=3, k=64, seed=16),
}


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def make_case_inputs(name):
    s = CASES[name]
    rs = np.random.RandomState(s["seed"])
    n, d, nq = s["n"], s["d"], s["nq"]
    scale = np.float32(1.0 / np.sqrt(np.float32(d)))
    if s.get("dup"):
        # small-integer rows (exactly representable in f16/int8 products) with many duplicates
        uniq = rs.randint(-3, 4, size=(n // 6, d)).astype(np.float32) * np.float32(0.125)
        idx = rs.randint(0, uniq.shape[0], size=n)
        base = uniq[idx].copy()
        queries = rs.randint(-3, 4, size=(nq, d)).astype(np.float32) * np.float32(0.25)
    else:
        base = rs.standard_normal((n, d)).astype(np.float32) * scale
        queries = rs.standard_normal((nq, d)).astype(np.float32) * scale
        # half of the queries are corpus rows (self-match, like tools/nvdb_make_query.cpp samples)
        for i in range(0, nq, 2):
            queries[i] = base[(i * 7919 + 13) % n]
    return np.ascontiguousarray(base), np.ascontiguousarray(queries)


def make_dot_inputs(d, m=48):
    rs = np.random.RandomState(1000 + d)
    q = rs.standard_normal((m, d)).astype(np.float32)
    x32 = rs.standard_normal((m, d)).astype(np.float32)
    x16 = rs.standard_normal((m, d)).astype(np.float16).view(np.uint16)  # numpy RNE cast; any half bits are valid inputs
    x8 = rs.randint(-127, 128, size=(m, d)).astype(np.int8)
    sc = (rs.rand(m).astype(np.float32) * np.float32(0.01) + np.float32(1e-3)).astype(np.float32)
    return q, x32, np.ascontiguousarray(x16), x8, sc


def make_f16_specials():
    """float32 values probing every branch of the f32->f16 conversion."""
    v = [0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 70000.0, -70000.0, np.inf, -np.inf,
         2.0 ** -14, 2.0 ** -15, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25, 2.0 ** -26, 1e-40, -1e-40,
         1.0 + 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -20, 1.0 + 3 * 2.0 ** -11, 2047.5, 2048.5, 0.1, -0.3333333,
         (2.0 ** -14) * (1 - 2.0 ** -11), (2.0 ** -14) * (1 - 2.0 ** -12), 6.0e-8, 5.9e-8, 3.0e-8, 2.9e-8]
    rs = np.random.RandomState(77)
    extra = (rs.standard_normal(200) * np.exp(rs.uniform(-20, 12, 200))).astype(np.float32)
    return np.concatenate([np.array(v, dtype=np.float32), extra]).astype(np.float32)
