"""C++ host layer called directly (lib/libnvdb_host_capi.so, test hooks over nvdb::) and the oracle, both against
goldens produced by the REAL reference:

  * tests/golden/refine_conv_golden.npz -- nvdb::f16_to_f32_scalar on all 65 536 half patterns and
    nvdb::base_row_to_f32 on rows of each base dtype (reference include/nvdb/f16_scalar.h, to_f32_row.h): the
    row-conversion half of the CPU refine.  This pins oracle_refine_*'s mode 1 conversion; its
    kernel-order half (mode 0, cuda_refine.cu:326-392) stays "parity unpinned" (DESIGN.md section 6).
  * tests/golden/flat_golden.npz -- the dot kernels on both dispatch branches (SIMD and forced scalar,
    reference src/simd_dot.cpp:52-64, 127-136, 202-213).
"""
import ctypes as C
import os

import numpy as np
import pytest

import pyoracle as po
from golden_inputs import DOT_DIMS, make_case_inputs, make_dot_inputs, sha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAPI = os.path.join(ROOT, "nano-vectordb_amd", "lib", "libnvdb_host_capi.so")
_f32p = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(CAPI):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "nano-vectordb_amd"), "-j8"])
    L = C.CDLL(CAPI)
    L.nvdb_host_last_error.restype = C.c_char_p
    for name, args in (("nvdb_host_dot_f32", [_f32p, _f32p, C.c_uint32]), ("nvdb_host_dot_f32_f16base", [_f32p, C.c_void_p, C.c_uint32]),
                       ("nvdb_host_dot_f32_f16base_scalar", [_f32p, C.c_void_p, C.c_uint32]),
                       ("nvdb_host_dot_f32_i8base", [_f32p, C.c_void_p, C.c_uint32, C.c_float])):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = args
    L.nvdb_host_f16_to_f32.argtypes = [C.c_void_p, C.c_uint64, _f32p]
    L.nvdb_host_dataset_open.restype = C.c_void_p
    L.nvdb_host_dataset_open.argtypes = [C.c_char_p]
    L.nvdb_host_dataset_close.argtypes = [C.c_void_p]
    L.nvdb_host_base_row_to_f32.argtypes = [C.c_void_p, C.c_uint64, _f32p]
    L.nvdb_host_refine_topk_l2.argtypes = [C.c_void_p, _f32p, C.POINTER(C.c_int64), C.c_int, C.c_uint32, C.POINTER(C.c_uint64), _f32p]
    return L


@pytest.fixture(scope="module")
def conv_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "refine_conv_golden.npz"))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------------------------------------ half -> float
def test_f16_to_f32_all_patterns_oracle_and_host(oracle, host, conv_golden):
    want = conv_golden["f16_to_f32_all"]
    allh = np.arange(65536, dtype=np.uint16)
    assert (want[0x7C01] & 0x7FC00000) == 0x7F800000                       # signalling NaNs stay signalling: results were written to memory
    assert np.array_equal(oracle.base_row_to_f32(allh, po.DT_F16).view(np.uint32), want)   # NaN payloads included: compared as bits
    out = np.empty(65536, dtype=np.float32)
    host.nvdb_host_f16_to_f32(allh.ctypes.data, 65536, out.ctypes.data_as(_f32p))
    assert np.array_equal(out.view(np.uint32), want)
    # numpy's own half -> float agrees on everything that is not a NaN (an independent third opinion)
    npf = allh.view(np.float16).astype(np.float32)
    ok = ~np.isnan(npf)
    assert np.array_equal(npf.view(np.uint32)[ok], want[ok])


# ------------------------------------------------------------------------------------------------ base_row_to_f32
def _main768_as(oracle, tag):
    base32, _ = make_case_inputs("main768")
    if tag == "f32":
        return base32, po.DT_F32, None
    if tag == "f16":
        return oracle.f32_to_f16(base32), po.DT_F16, None
    b8, sc = oracle.quantize_i8(base32)
    return b8, po.DT_I8, sc


@pytest.mark.parametrize("tag", ["f32", "f16", "i8"])
def test_base_row_to_f32_oracle_and_host(oracle, host, conv_golden, tag, tmp_path):
    base, dt, sc = _main768_as(oracle, tag)
    if tag == "f16":
        assert sha(base) == bytes(conv_golden["main768_f16_sha"]).hex()    # same corpus bits the reference's tool wrote
    if tag == "i8":
        assert sha(base, sc) == bytes(conv_golden["main768_i8_sha"]).hex()
    want = conv_golden[f"main768_{tag}_rows_f32"]
    rows = [int(r) for r in conv_golden["rows"]]
    got = np.stack([oracle.base_row_to_f32(base[r], dt, sc[r] if sc is not None else 0.0) for r in rows])
    assert np.array_equal(_bits(got), want)
    p = str(tmp_path / f"b_{tag}.vecbin")
    po.write_vecbin(p, base, dt, sc)
    h = host.nvdb_host_dataset_open(p.encode())
    assert h, host.nvdb_host_last_error()
    out = np.empty(base.shape[1], dtype=np.float32)
    for i, r in enumerate(rows):
        assert host.nvdb_host_base_row_to_f32(h, r, out.ctypes.data_as(_f32p)) == 0
        assert np.array_equal(out.view(np.uint32), want[i]), (tag, r)
    host.nvdb_host_dataset_close(h)


# ------------------------------------------------------------------------------------------------ CPU refine (a12)
@pytest.mark.parametrize("tag", ["f32", "f16"])
def test_cpu_refine_host_matches_oracle_mode1(oracle, host, tag, tmp_path):
    """nvdb::refine_topk_l2_ids (host/include/nvdb/cpu_refine.h) vs the oracle's mode 1: both restate
    apps/nvdb_ivf_eval.cpp:232-240, 278-307 over the (now pinned) row conversion; distances must agree bit for bit and
    ids wherever distances are distinct."""
    base, dt, _ = _main768_as(oracle, tag)
    _, queries = make_case_inputs("main768")
    n, d = base.shape
    rs = np.random.RandomState(9)
    R, K = 300, 10
    cand = rs.randint(0, n, size=(len(queries), R)).astype(np.uint32)
    cand[rs.rand(*cand.shape) < 0.02] = 0xFFFFFFFF
    cand[1, 5:] = 0xFFFFFFFF                                     # fewer candidates than K
    oid, odist = oracle.refine(base, dt, queries, cand, K, mode=1)
    p = str(tmp_path / f"b_{tag}.vecbin")
    po.write_vecbin(p, base, dt)
    h = host.nvdb_host_dataset_open(p.encode())
    for qi in range(len(queries)):
        c64 = np.where(cand[qi] == 0xFFFFFFFF, -1, cand[qi].astype(np.int64)).astype(np.int64)
        ids = np.empty(K, dtype=np.uint64)
        dist = np.empty(K, dtype=np.float32)
        got = host.nvdb_host_refine_topk_l2(h, queries[qi].ctypes.data_as(_f32p), c64.ctypes.data_as(C.POINTER(C.c_int64)), R, K,
                                            ids.ctypes.data_as(C.POINTER(C.c_uint64)), dist.ctypes.data_as(_f32p))
        nvalid = int((oid[qi] != 0xFFFFFFFF).sum())
        assert got == nvalid
        assert np.array_equal(dist[:got].view(np.uint32), odist[qi, :got].view(np.uint32)), qi
        distinct = np.r_[True, dist[1:got] != dist[:got - 1]] & np.r_[dist[:got - 1] != dist[1:got], True]
        assert np.array_equal(ids[:got][distinct], oid[qi, :got].astype(np.uint64)[distinct]), qi
    host.nvdb_host_dataset_close(h)


# ------------------------------------------------------------------------------------------------ dot kernels, both branches
@pytest.mark.parametrize("d", DOT_DIMS)
def test_host_dot_kernels_simd_and_scalar_paths(host, golden, d):
    q, x32, x16, x8, sc = make_dot_inputs(d)
    assert sha(q, x32, x16, x8, sc) == bytes(golden[f"dot{d}_sha"]).hex()
    if not host.nvdb_host_simd_available():
        pytest.skip("host CPU lacks avx2/fma/f16c: only the scalar branch exists here")

    def run(force):
        host.nvdb_host_set_force_scalar(force)
        f32 = np.array([host.nvdb_host_dot_f32(q[i].ctypes.data_as(_f32p), x32[i].ctypes.data_as(_f32p), d) for i in range(len(q))], dtype=np.float32)
        f16 = np.array([host.nvdb_host_dot_f32_f16base(q[i].ctypes.data_as(_f32p), x16[i].ctypes.data, d) for i in range(len(q))], dtype=np.float32)
        i8 = np.array([host.nvdb_host_dot_f32_i8base(q[i].ctypes.data_as(_f32p), x8[i].ctypes.data, d, float(sc[i])) for i in range(len(q))], dtype=np.float32)
        host.nvdb_host_set_force_scalar(0)
        return f32, f16, i8
    for tag, force in (("simd", 0), ("scalar", 1)):
        f32, f16, i8 = run(force)
        assert np.array_equal(_bits(f32), golden[f"dot{d}_{tag}_f32"]), (d, tag, "f32")
        assert np.array_equal(_bits(f16), golden[f"dot{d}_{tag}_f16"]), (d, tag, "f16")      # the fp16 kernel ignores the switch (reference quirk)
        assert np.array_equal(_bits(i8), golden[f"dot{d}_{tag}_i8"]), (d, tag, "i8")


def test_host_f16_scalar_fallback_is_double_accumulation(host, oracle):
    """What a host WITHOUT avx2/fma/f16c computes for fp16 rows (reference src/simd_dot.cpp:133-135): sequential double
    accumulation of double(q) * double(half), cast to float.  The reference never takes this branch on this CPU, so the
    check is against the definition (products of two floats are exact in double; the sum is sequential)."""
    for d in (768, 37, 5):
        q, _, x16, _, _ = make_dot_inputs(d)
        xf = oracle.f16_to_f32(x16).astype(np.float64)
        for i in range(8):
            s = 0.0
            for a, b in zip(q[i].astype(np.float64), xf[i]):
                s += a * b
            got = host.nvdb_host_dot_f32_f16base_scalar(q[i].ctypes.data_as(_f32p), x16[i].ctypes.data, d)
            assert np.float32(got).view(np.uint32) == np.float32(s).view(np.uint32), (d, i)
