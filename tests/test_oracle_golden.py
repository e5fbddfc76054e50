"""CPU tests: the C restatement (oracle/nvdb_oracle.c) against golden vectors produced by the REAL
reference (oracle/make_golden.py -> tests/golden/flat_golden.npz).  This is what "pins" the oracle."""
import numpy as np
import pytest

import pyoracle as po
from golden_inputs import CASES, DOT_DIMS, make_case_inputs, make_dot_inputs, make_f16_specials, sha
from parity import assert_topk_equal


def _sha_ok(golden, key, *arrays):
    assert bytes(golden[key]).hex() == sha(*arrays), f"input drift for {key}: numpy stream changed?"


@pytest.mark.parametrize("d", DOT_DIMS)
def test_dot_kernels_bitexact(oracle, golden, d):
    q, x32, x16, x8, sc = make_dot_inputs(d)
    _sha_ok(golden, f"dot{d}_sha", q, x32, x16, x8, sc)
    L = oracle.lib
    m = len(q)
    f32 = np.array([L.oracle_dot_f32(po._p(q[i], po._f32p), po._p(x32[i], po._f32p), d) for i in range(m)], dtype=np.float32)
    f16 = np.array([L.oracle_dot_f32_f16base(po._p(q[i], po._f32p), x16[i].ctypes.data, d) for i in range(m)], dtype=np.float32)
    i8 = np.array([L.oracle_dot_f32_i8base(po._p(q[i], po._f32p), x8[i].ctypes.data, d, float(sc[i])) for i in range(m)], dtype=np.float32)
    assert np.array_equal(f32.view(np.uint32), golden[f"dot{d}_simd_f32"])
    assert np.array_equal(f16.view(np.uint32), golden[f"dot{d}_simd_f16"])
    assert np.array_equal(i8.view(np.uint32), golden[f"dot{d}_simd_i8"])
    # forced-scalar paths (double accumulation); f16 has no force-scalar switch
    s32 = np.array([L.oracle_dot_f32_scalar(po._p(q[i], po._f32p), po._p(x32[i], po._f32p), d) for i in range(m)], dtype=np.float32)
    s8 = np.array([L.oracle_dot_f32_i8base_scalar(po._p(q[i], po._f32p), x8[i].ctypes.data, d, float(sc[i])) for i in range(m)], dtype=np.float32)
    assert np.array_equal(s32.view(np.uint32), golden[f"dot{d}_scalar_f32"])
    assert np.array_equal(s8.view(np.uint32), golden[f"dot{d}_scalar_i8"])
    assert np.array_equal(golden[f"dot{d}_scalar_f16"], golden[f"dot{d}_simd_f16"])


def test_f16_conversion_specials(oracle, golden):
    sp = make_f16_specials()
    got_a = oracle.f32_to_f16(sp.reshape(1, -1)).reshape(-1)
    finite_or_inf = ~np.isnan(sp)
    assert np.array_equal(got_a[finite_or_inf], golden["f16_specials_a"][finite_or_inf])
    spb = sp[: (len(sp) // 8) * 8 - 3]
    got_b = oracle.f32_to_f16(spb.reshape(1, -1)).reshape(-1)
    assert np.array_equal(got_b, golden["f16_specials_b"])
    # half -> float is exact and inverts the conversion on representable values
    back = oracle.f16_to_f32(got_a[np.isfinite(sp)])
    again = oracle.f32_to_f16(back)
    assert np.array_equal(again, got_a[np.isfinite(sp)])


@pytest.mark.parametrize("name", list(CASES))
def test_converters_match_reference_tools(oracle, golden, name):
    base32, queries = make_case_inputs(name)
    _sha_ok(golden, f"{name}_sha", base32, queries)
    b16 = oracle.f32_to_f16(base32)
    b8, sc8 = oracle.quantize_i8(base32)
    assert bytes(golden[f"{name}_f16_sha"]).hex() == sha(b16)
    assert bytes(golden[f"{name}_i8_sha"]).hex() == sha(b8, sc8)
    assert np.array_equal(b16[:4], golden[f"{name}_f16_head"])
    assert np.array_equal(b8[:4], golden[f"{name}_i8_head"])
    assert np.array_equal(sc8[:16].view(np.uint32), golden[f"{name}_i8_scales_head"].view(np.uint32))


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("tag", ["f32", "f16", "i8"])
def test_flat_topk_matches_reference(oracle, golden, name, tag):
    base32, queries = make_case_inputs(name)
    k = CASES[name]["k"]
    scales = None
    if tag == "f32":
        base, dt = base32, po.DT_F32
    elif tag == "f16":
        base, dt = oracle.f32_to_f16(base32), po.DT_F16
    else:
        (base, scales), dt = oracle.quantize_i8(base32), po.DT_I8
    ids, sc = oracle.flat_topk(base, dt, queries, k, scales)
    for variant in ("st", "omp"):
        rids, rsc = golden[f"{name}_{tag}_{variant}_ids"], golden[f"{name}_{tag}_{variant}_scores"].view(np.float32)
        assert ids.shape == rids.shape == (len(queries), min(k, len(base)))
        for qi in range(len(queries)):
            allsc = oracle.scores(base, dt, queries[qi], scales)
            assert_topk_equal(ids[qi], sc[qi], rids[qi], rsc[qi], score_of=lambda i: allsc[i],
                              what=f"{name}/{tag}/{variant}/q{qi}")


def test_tie_case_really_has_ties(oracle, golden):
    base32, queries = make_case_inputs("ties64")
    ids, sc = oracle.flat_topk(base32, po.DT_F32, queries, 10)
    assert any(len(np.unique(sc[q])) < sc.shape[1] for q in range(sc.shape[0]))
    # canonical order inside ties: id ascending
    for q in range(sc.shape[0]):
        for j in range(1, sc.shape[1]):
            if sc[q, j] == sc[q, j - 1]:
                assert ids[q, j] > ids[q, j - 1]


def test_gtbin_and_search_stdout(oracle, golden):
    base32, queries = make_case_inputs("main768")
    b16 = oracle.f32_to_f16(base32)
    ids, _ = oracle.flat_topk(b16, po.DT_F16, queries, 10)
    assert np.array_equal(ids.astype(np.uint32), golden["main768_gtbin_f16_ids"])   # nvdb_gt_build GT_MODE=st
    hdr = bytes(golden["main768_gtbin_raw"])
    import struct
    magic, ver, metric, k, dim, Q, N = struct.unpack("<QIIIIQQ", hdr[:40])
    assert (magic, ver, metric, k, dim, Q, N) == (po.GT_MAGIC, 1, 1, 10, 768, 8, 3000)
    # nvdb_search prints query 0's top-k with 6 decimals (apps/nvdb_search.cpp:31-39)
    ids32, sc32 = oracle.flat_topk(base32, po.DT_F32, queries[:1], 10)
    lines = bytes(golden["main768_search_stdout"]).decode().strip().splitlines()[1:]
    for j, line in enumerate(lines):
        assert line == f"#{j + 1} row={ids32[0, j]} score={sc32[0, j]:.6f}"


def test_refine_orders_agree_within_tolerance(oracle):
    """Refine restatements (parity UNPINNED vs the CUDA kernel): the fp32 GPU-order distance and
    the CPU double-precision distance must agree to fp32 rounding, and ids wherever gaps allow."""
    rs = np.random.RandomState(5)
    n, d, Q, R, K = 4000, 768, 6, 300, 10
    base = oracle.f32_to_f16((rs.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32))
    queries = (rs.standard_normal((Q, d)) / np.sqrt(d)).astype(np.float32)
    cand = rs.randint(0, n, size=(Q, R)).astype(np.uint32)
    cand[:, ::17] = 0xFFFFFFFF
    cand[0, 5] = n + 3            # out of range -> skipped (cuda_refine.cu:437)
    i0, d0 = oracle.refine(base, po.DT_F16, queries, cand, K, mode=0)
    i1, d1 = oracle.refine(base, po.DT_F16, queries, cand, K, mode=1)
    assert np.allclose(d0, d1, rtol=1e-5, atol=1e-6)
    assert (i0 == i1).mean() > 0.95
    assert not np.any(i0 == 0xFFFFFFFF) and np.all(np.diff(d0, axis=1) >= 0)
    # fewer valid candidates than K -> padding 0xFFFFFFFF / 1e30 (cuda_refine.cu:892-894)
    cand2 = np.full((1, 8), 0xFFFFFFFF, dtype=np.uint32)
    cand2[0, :3] = [7, 9, 11]
    i2, d2 = oracle.refine(base, po.DT_F16, queries[:1], cand2, 5, mode=0)
    assert list(i2[0, 3:]) == [0xFFFFFFFF] * 2 and np.all(d2[0, 3:] == np.float32(1e30))
