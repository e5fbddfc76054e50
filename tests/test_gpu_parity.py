"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(libnvdb_hip.so via nvdb_amd); the oracle (oracle/liboracle.so) and the committed golden vectors
produced by the real reference are the checkers.

Bars: ids and score BITS identical to the reference CPU path (tie groups set-wise, tests/parity.py);
refine: ids and distance bits identical to the restated reference kernel order, and within 1e-5
relative of the reference's CPU (double) refine.
"""
import numpy as np
import pytest

import nvdb_amd
import pyoracle as po
from golden_inputs import CASES, make_case_inputs
from parity import assert_topk_equal

pytestmark = pytest.mark.gpu

SEED = 20240613


@pytest.fixture(scope="module")
def ctx():
    c = nvdb_amd.HipContext(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_dev():
    """A context of libnvdb_hip_dev.so: the same ABI plus the kernel variants that lost their A/B (32x32x16 fp16 build for
    batches > 128, two-plane int8 kernel, filter_i8w_kernel at 64 queries per wave, 8-wave and 32x32x32 int8 builds); the
    product library compiles none of them and rejects the options that select them."""
    c = nvdb_amd.HipContext(0, dev=True)
    yield c
    c.close()


def _as_dtype(oracle, base32, tag):
    if tag == "f32":
        return base32, po.DT_F32, None
    if tag == "f16":
        return oracle.f32_to_f16(base32), po.DT_F16, None
    b8, sc = oracle.quantize_i8(base32)
    return b8, po.DT_I8, sc


def _check_against_oracle(oracle, base, dt, scales, queries, ids, sc, k, what):
    oid, osc = oracle.flat_topk(base, dt, queries, k, scales)
    assert ids.shape == oid.shape, f"{what}: {ids.shape} vs {oid.shape}"
    for qi in range(len(queries)):
        allsc = None

        def score_of(i, qi=qi):
            nonlocal allsc
            if allsc is None:
                allsc = oracle.scores(base, dt, queries[qi], scales)
            return allsc[i]
        assert_topk_equal(ids[qi], sc[qi], oid[qi], osc[qi], score_of=score_of, what=f"{what}/q{qi}")
        # our own order is canonical: (score desc, id asc)
        assert np.array_equal(ids[qi], oid[qi]), f"{what}/q{qi}: canonical order differs"


# ----------------------------------------------------------------------------- generator
@pytest.mark.parametrize("dtype", [nvdb_amd.DT_F32, nvdb_amd.DT_F16, nvdb_amd.DT_I8])
def test_device_generator_matches_host(ctx, dtype):
    n, d, base_row = 5000, 768, 123456789012
    ctx.generate_corpus(SEED, n, d, dtype, row_base=base_row)
    dev, dsc = ctx.download_rows(0, n)
    host, hsc = nvdb_amd.synth_corpus(SEED, base_row, n, d, dtype)
    assert np.array_equal(dev, host)
    if dtype == nvdb_amd.DT_I8:
        assert np.array_equal(dsc.view(np.uint32), hsc.view(np.uint32))
    info = ctx.corpus_info()
    assert info["n"] == n and info["dim"] == d and info["row_base"] == base_row
    assert 0.99 < info["max_row_norm"] < 1.02


# ----------------------------------------------------------------------------- golden cases, exact path
@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("tag", ["f32", "f16", "i8"])
def test_golden_cases_exact_path(ctx, oracle, golden, name, tag):
    base32, queries = make_case_inputs(name)
    k = CASES[name]["k"]
    base, dt, scales = _as_dtype(oracle, base32, tag)
    ctx.upload_corpus(base, dt, scales)
    ctx.set_option("path", 1)
    ids, sc = ctx.search_batch(queries, k)
    ctx.set_option("path", 0)
    assert ctx.stats()["path"] == 1
    for variant in ("st", "omp"):
        rids, rsc = golden[f"{name}_{tag}_{variant}_ids"], golden[f"{name}_{tag}_{variant}_scores"].view(np.float32)
        assert ids.shape == rids.shape
        for qi in range(len(queries)):
            allsc = oracle.scores(base, dt, queries[qi], scales)
            assert_topk_equal(ids[qi], sc[qi], rids[qi], rsc[qi], score_of=lambda i: allsc[i],
                              what=f"{name}/{tag}/{variant}/q{qi}")
    _check_against_oracle(oracle, base, dt, scales, queries, ids, sc, k, f"{name}/{tag}/oracle")


def test_single_query_and_edge_arguments(ctx, oracle):
    base32, queries = make_case_inputs("main768")
    ctx.upload_corpus(base32, po.DT_F32)
    ids, sc = ctx.search_batch(queries[0], 10)               # 1-D query
    oid, osc = oracle.flat_topk(base32, po.DT_F32, queries[:1], 10)
    assert np.array_equal(ids, oid) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
    ids0, sc0 = ctx.search_batch(queries, 0)                 # k == 0 -> empty (flat_index.cpp:18)
    assert ids0.shape == (len(queries), 0)
    i65, s65 = ctx.search_batch(queries, 65)                 # k is bounded by N only (flat_index.cpp:24): the any-k path
    o65, os65 = oracle.flat_topk(base32, po.DT_F32, queries, 65)
    assert ctx.stats()["path"] == 3 and np.array_equal(i65, o65) and np.array_equal(s65.view(np.uint32), os65.view(np.uint32))
    idx = nvdb_amd.FlatIndexHIP(base32, po.DT_F32)
    r = idx.search_topk_dot(queries[1], 3)
    assert [i for i, _ in r] == oid[0:0].tolist() or len(r) == 3
    o1, s1 = oracle.flat_topk(base32, po.DT_F32, queries[1:2], 3)
    assert [i for i, _ in r] == o1[0].tolist() and [np.float32(s) for _, s in r] == s1[0].tolist()
    assert idx.search_topk_dot(queries[1], 0) == []
    with pytest.raises(RuntimeError):
        nvdb_amd.FlatIndexHIP(np.zeros((0, 8), dtype=np.float32), po.DT_F32)   # "Empty base"
    idx.ctx.close()


def test_search_before_corpus_is_an_error():
    c = nvdb_amd.HipContext(0)
    with pytest.raises(nvdb_amd.NvdbError) as e:
        c.search_batch(np.zeros((1, 8), dtype=np.float32), 1)
    assert e.value.status == 4 and "Empty base" in str(e.value)
    c.close()


# ----------------------------------------------------------------------------- MFMA filter path
@pytest.mark.parametrize("nq,k", [(64, 10), (300, 10), (17, 1), (256, 64)])
def test_filter_path_f16_matches_oracle(ctx, oracle, nq, k):
    n, d = 150000 + 13, 768                                  # ragged: not a multiple of the 32-row tile
    ctx.generate_corpus(SEED, n, d, nvdb_amd.DT_F16)
    base, _ = nvdb_amd.synth_corpus(SEED, 0, n, d, nvdb_amd.DT_F16)
    queries = nvdb_amd.synth_rows_f32(SEED + 1, 0, nq, d)
    queries[::3] = oracle.f16_to_f32(base[(np.arange(0, nq, 3) * 7919) % n]) if nq < 100 else queries[::3]
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(queries, k)
    st = ctx.stats()
    ctx.set_option("path", 0)
    assert st["path"] == 2 and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    assert st["chunks"] >= 2 and st["candidates"] >= nq * k
    _check_against_oracle(oracle, base, po.DT_F16, None, queries, ids, sc, k, f"filter/nq{nq}/k{k}")


@pytest.mark.parametrize("nq", [5, 200])
def test_filter_path_fp32_corpus_via_fp16_shadow(ctx, oracle, golden, nq):
    """fp32 corpus: the MFMA filter streams an fp16 shadow copy, survivors are re-scored from the fp32 rows in
    the reference's order (simd_dot.cpp:26-49) -> ids and score bits identical to the CPU path."""
    n, d, k = 70000 + 3, 768, 10
    ctx.generate_corpus(SEED + 60, n, d, nvdb_amd.DT_F32)
    base, _ = nvdb_amd.synth_corpus(SEED + 60, 0, n, d, nvdb_amd.DT_F32)
    queries = nvdb_amd.synth_rows_f32(SEED + 61, 0, nq, d)
    queries[0] = base[12345]
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(queries, k)
    st = ctx.stats()
    ctx.set_option("path", 0)
    assert st["path"] == 2 and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    _check_against_oracle(oracle, base, po.DT_F32, None, queries, ids, sc, k, f"f32-shadow/nq{nq}")
    # the reference's own golden case through the automatic path choice
    base32, gq = make_case_inputs("main768")
    ctx.upload_corpus(base32, po.DT_F32)
    gi, gs = ctx.search_batch(gq, 10)
    assert ctx.stats()["path"] == 2
    assert np.array_equal(gi, golden["main768_f32_st_ids"]) and np.array_equal(gs.view(np.uint32), golden["main768_f32_st_scores"])


def test_filter_path_mfma16_variant_matches_default(ctx_dev, oracle):
    """The 16x16x32-MFMA build of the filter kernel (the product's) must give the same bits as the 32x32x16 build (developer library)."""
    ctx = ctx_dev
    n, d, nq, k = 90000 + 5, 768, 300, 10
    ctx.generate_corpus(SEED + 50, n, d, nvdb_amd.DT_F16)
    base, _ = nvdb_amd.synth_corpus(SEED + 50, 0, n, d, nvdb_amd.DT_F16)
    queries = nvdb_amd.synth_rows_f32(SEED + 51, 0, nq, d)
    ctx.set_option("path", 2)
    res = {}
    for m16 in (0, 1):
        ctx.set_option("mfma16", m16)
        res[m16] = ctx.search_batch(queries, k)
        st = ctx.stats()
        assert st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    ctx.set_option("mfma16", 1)
    ctx.set_option("path", 0)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
    _check_against_oracle(oracle, base, po.DT_F16, None, queries, res[1][0], res[1][1], k, "mfma16")


@pytest.mark.parametrize("nq,k,d", [(64, 10, 768), (300, 10, 768), (40, 64, 256)])
def test_filter_path_int8_matches_oracle(ctx, oracle, nq, k, d):
    """int8(+scale) corpus on the integer matrix cores (two-plane int8 query), exact rescore in the
    reference's dequantise-and-FMA order (simd_dot.cpp:160-199): ids and score bits must match."""
    n = 120000 + 7
    ctx.generate_corpus(SEED + 40, n, d, nvdb_amd.DT_I8)
    base, scales = nvdb_amd.synth_corpus(SEED + 40, 0, n, d, nvdb_amd.DT_I8)
    queries = nvdb_amd.synth_rows_f32(SEED + 41, 0, nq, d)
    queries[1] *= np.float32(123.0)
    queries[2, :3] *= np.float32(25.0)
    res = {}
    for path in (2, 1):
        ctx.set_option("path", path)
        res[path] = ctx.search_batch(queries, k)
        st = ctx.stats()
        assert st["path"] == path and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    ctx.set_option("path", 0)
    assert np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1].view(np.uint32), res[2][1].view(np.uint32))
    _check_against_oracle(oracle, base, po.DT_I8, scales, queries, res[2][0], res[2][1], k, f"filter-i8/nq{nq}")


@pytest.mark.parametrize("nq,k,d", [(300, 10, 768), (1024, 10, 768), (130, 64, 512), (200, 5, 256), (64, 10, 768), (1, 10, 768), (300, 10, 384), (100, 64, 384), (260, 10, 640), (128, 10, 768), (33, 64, 768), (100, 10, 512)])
def test_int8_two_stage_kernel_matches_two_plane_kernel(ctx_dev, oracle, nq, k, d):
    """(developer library: it holds the builds the product's kernel is compared with)  int8 corpora run the two-stage kernel: hi plane always, lo plane only for tiles whose hi-plane value
    could reach the threshold (64 queries per wave for batches > 128, 32 below).  It must log exactly the survivors of the two-plane kernel, so ids, score
    bits and the number of rescored candidates are identical; both match the CPU int8 path."""
    ctx = ctx_dev
    n = 200000 + 9
    ctx.generate_corpus(SEED + 80, n, d, nvdb_amd.DT_I8)
    base, scales = nvdb_amd.synth_corpus(SEED + 80, 0, n, d, nvdb_amd.DT_I8)
    queries = nvdb_amd.synth_rows_f32(SEED + 81, 0, nq, d)
    if nq > 5:
        queries[3] *= np.float32(977.0)
        queries[4, :7] *= np.float32(31.0)                   # heavy-tailed query: large hi-plane, small lo-plane share
        queries[5] = np.round(queries[5] * 40) / 40          # few distinct levels
    ctx.set_option("path", 2)
    res, stats = {}, {}
    # 0: two-plane kernel; 1: two-stage, software-pipelined build for batches > 128, first-stage survivors logged and finished
    # after the stream (the default); 2: two-stage, filter_i8w_kernel; 3: the pipelined build with the in-loop second stage
    # (deferred v_dot4 slots); 4: the same on 8 waves of 32 queries; 5: the product's default: the logged build on
    # v_mfma_i32_16x16x64_i8 (d >= 384; d = 256 keeps the 32x32x32 build)
    variants = ((0, 1, 0, 0, 0), (1, 1, 0, 0, 0), (1, 0, 0, 0, 0), (1, 1, 0, 1, 0), (1, 1, 1, 1, 0), (1, 1, 0, 0, 1), (1, 1, 1, 0, 1))   # last: 16x16x64 on 8 waves (d = 768)
    for var, (wide, pipe, w8, defer, m16) in enumerate(variants):
        ctx.set_option("i8_wide", wide)
        ctx.set_option("i8_pipe", pipe)
        ctx.set_option("i8_waves8", w8)
        ctx.set_option("i8_defer", defer)
        ctx.set_option("i8_mfma16", m16)
        res[var] = ctx.search_batch(queries, k)
        stats[var] = ctx.stats()
        assert stats[var]["path"] == 2 and stats[var]["bound_violations"] == 0 and stats[var]["overflow_queries"] == 0, stats[var]
    ctx.set_option("i8_wide", 1)
    ctx.set_option("i8_pipe", 1)
    ctx.set_option("i8_waves8", 0)
    ctx.set_option("i8_defer", 0)
    ctx.set_option("i8_mfma16", 1)
    if (8 < nq <= 128 and d in (512, 768)) or (64 < nq <= 128 and d == 384):
        # batches <= 128 run the 16x16x64 logged build on 8 waves (the variants above that do not defer all took it):
        # against filter_i8w_kernel<DIM, 1>, the kernel these batches ran before
        ctx.set_option("i8_small8", 0)
        old = ctx.search_batch(queries, k)
        st_old = ctx.stats()
        ctx.set_option("i8_small8", 1)
        assert np.array_equal(old[0], res[1][0]) and np.array_equal(old[1].view(np.uint32), res[1][1].view(np.uint32))
        assert st_old["candidates"] == stats[1]["candidates"]
    ctx.set_option("path", 0)
    for var in range(1, len(variants)):
        assert np.array_equal(res[0][0], res[var][0]) and np.array_equal(res[0][1].view(np.uint32), res[var][1].view(np.uint32)), var
        assert stats[0]["candidates"] == stats[var]["candidates"], var
        assert stats[var]["i8_stage1_tiles"] > 0, stats[var]       # some values passed the hi-plane test and were finished exactly
    assert stats[0]["i8_stage1_tiles"] == 0
    _check_against_oracle(oracle, base, po.DT_I8, scales, queries, res[1][0], res[1][1], k, f"i8-wide/nq{nq}")


@pytest.mark.parametrize("nq,k,d", [(1, 10, 384), (48, 10, 384), (128, 64, 384), (200, 10, 384), (1024, 10, 384), (700, 3, 384), (100, 10, 640), (520, 10, 640)])
def test_int8_d384_native(ctx, oracle, nq, k, d):
    """d = 384 is the reference's own data dimension (all-MiniLM-L6-v2, Performance.md): int8 rows of 384 bytes stream as they
    are -- the LDS image of a tile uses the swizzle for row strides that are odd multiples of 128 bytes (kernels_filter.h,
    swz_chunk).  Filter path == exact path == oracle for every batch regime (32 / 64 queries per wave, 1..4 query tiles)."""
    n = 300000 + 37                                              # (d = 640: the same swizzle family, five 128-byte groups per row)
    ctx.generate_corpus(SEED + 86, n, d, nvdb_amd.DT_I8)
    base, scales = nvdb_amd.synth_corpus(SEED + 86, 0, n, d, nvdb_amd.DT_I8)
    queries = nvdb_amd.synth_rows_f32(SEED + 87, 0, nq, d)
    queries[0] = base[n - 1].astype(np.float32) * scales[n - 1]          # the last row (tail tile) is somebody's best match
    if nq > 8:
        queries[3] *= np.float32(512.0)
        queries[5, 100:] = 0.0
    res = {}
    for path in (1, 2):
        ctx.set_option("path", path)
        res[path] = ctx.search_batch(queries, k)
        st = ctx.stats()
        assert st["path"] == path and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    ctx.set_option("path", 0)
    assert res[2][0][0, 0] == n - 1
    assert np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1].view(np.uint32), res[2][1].view(np.uint32))
    sub = slice(0, min(nq, 16))
    _check_against_oracle(oracle, base, po.DT_I8, scales, queries[sub], res[2][0][sub], res[2][1][sub], k, f"i8 d={d} nq={nq}")


@pytest.mark.parametrize("d,nq,k", [(300, 200, 10), (320, 48, 10), (100, 70, 5), (600, 300, 10)])
def test_int8_other_dims_through_the_zero_padded_shadow(ctx, oracle, d, nq, k):
    """int8 corpora whose dim is not a multiple of 128 in [256, 1536]: the filter streams a copy with rows zero-padded to the
    next one (300 -> 384, 100 -> 256, 600 -> 640); the rescore reads the original rows."""
    n = 90000 + 3
    ctx.generate_corpus(SEED + 84, n, d, nvdb_amd.DT_I8)
    base, scales = nvdb_amd.synth_corpus(SEED + 84, 0, n, d, nvdb_amd.DT_I8)
    queries = nvdb_amd.synth_rows_f32(SEED + 85, 0, nq, d)
    queries[2] = base[5555].astype(np.float32) * scales[5555]
    res = {}
    for path in (1, 2):
        ctx.set_option("path", path)
        res[path] = ctx.search_batch(queries, k)
        st = ctx.stats()
        assert st["path"] == path and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    ctx.set_option("path", 0)
    assert np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1].view(np.uint32), res[2][1].view(np.uint32))
    _check_against_oracle(oracle, base, po.DT_I8, scales, queries, res[2][0], res[2][1], k, f"i8-pad/d{d}")


@pytest.mark.parametrize("d,nq,k", [(1536, 300, 10), (1024, 64, 10), (1280, 130, 5), (1000, 200, 10), (1100, 33, 64), (1536, 1, 10), (1400, 1024, 10), (896, 150, 10), (1152, 64, 10), (1408, 300, 5)])
def test_int8_dims_up_to_1536_on_integer_mfma(ctx, oracle, d, nq, k):
    """768 < dim <= 1536 (the reference takes any dim, src/simd_dot.cpp:160-213): the two-stage int8 kernel on 32-row tiles
    with 32 queries per wave, K-step count up to 48; every multiple of 128 streams as it is (896 / 1152 / 1408: the LDS swizzle for
    odd multiples of 128 bytes), other dims stream a zero-padded copy.
    Results come from the original rows either way."""
    n = 70_000 + 13
    base32 = nvdb_amd.synth_rows_f32(SEED + 140 + d, 0, n, d)
    base, scales = oracle.quantize_i8(base32)
    queries = nvdb_amd.synth_rows_f32(SEED + 141 + d, 0, nq, d)
    if nq > 4:
        queries[3] *= 37.0                                        # a scaled query: thresholds live in the query's own units
        queries[4, : d // 2] = 0.0                                # and a half-empty one
    ctx.upload_corpus(base, po.DT_I8, scales)
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(queries, k)
    st = ctx.stats()
    ctx.set_option("path", 1)
    ei, es = ctx.search_batch(queries, k)
    ctx.set_option("path", 0)
    assert st["path"] == 2 and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    assert np.array_equal(ids, ei) and np.array_equal(sc.view(np.uint32), es.view(np.uint32))
    sub = slice(0, min(nq, 12))
    _check_against_oracle(oracle, base, po.DT_I8, scales, queries[sub], ids[sub], sc[sub], k, f"i8 d={d}")


def test_int8_big_dim_flat_queries_and_dense_blocks(ctx, oracle):
    """d = 1536 with queries whose components all sit at full magnitude (the hi plane's L1 cap of prep_q8_kernel quantises them
    more coarsely, so that |H| < 2^23 and (H << 7) + L stays inside int32 at any dim) against rows of all +-127; plus a
    near-duplicate block so that many values of one (row block, query block) pass the first stage (lo-plane MFMAs)."""
    n, d, nq, k = 40_000, 1536, 140, 10
    rs = np.random.RandomState(77)
    base32 = nvdb_amd.synth_rows_f32(SEED + 150, 0, n, d)
    base, scales = oracle.quantize_i8(base32)
    base[100] = 127; base[101] = -127; base[102] = np.where(rs.rand(d) < 0.5, 127, -127).astype(np.int8)
    base[2000:2040] = base[1999]                                  # exact duplicates inside one tile
    scales[2000:2040] = scales[1999]
    queries = nvdb_amd.synth_rows_f32(SEED + 151, 0, nq, d)
    queries[0] = 1.0; queries[1] = -1.0; queries[2] = np.sign(base[102]).astype(np.float32)
    queries[5] = base32[1999]
    ctx.upload_corpus(base, po.DT_I8, scales)
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(queries, k)
    st = ctx.stats()
    ctx.set_option("path", 0)
    assert st["path"] == 2 and st["bound_violations"] == 0, st
    _check_against_oracle(oracle, base, po.DT_I8, scales, queries[:8], ids[:8], sc[:8], k, "i8 d=1536 flat")


@pytest.mark.parametrize("d", [256, 384])
def test_int8_two_stage_kernel_with_a_negative_row_scale(ctx, oracle, d):
    """A negative row scale is never produced by the reference quantiser but is legal in the file format; both
    stages of the kernel are sign-agnostic (per-value products; |scale| in the lo-plane bound)."""
    n, nq, k = 50000, 160, 10
    base, scales = nvdb_amd.synth_corpus(SEED + 82, 0, n, d, nvdb_amd.DT_I8)
    base, scales = base.copy(), scales.copy()
    base[100] = -base[100]; scales[100] = -scales[100]       # same dequantised row, negative scale
    queries = nvdb_amd.synth_rows_f32(SEED + 83, 0, nq, d)
    queries[0] = base[100].astype(np.float32) * scales[100]
    ctx.upload_corpus(base, po.DT_I8, scales)
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(queries, k)
    st = ctx.stats()
    ctx.set_option("path", 0)
    assert st["path"] == 2 and st["i8_stage1_tiles"] > 0 and st["bound_violations"] == 0, st
    assert ids[0, 0] == 100
    _check_against_oracle(oracle, base, po.DT_I8, scales, queries, ids, sc, k, "i8-negscale")


@pytest.mark.parametrize("d,nq", [(768, 300), (512, 200)])
def test_int8_biased_accumulators_at_their_limits(ctx, oracle, d, nq):
    """The default int8 kernel for batches > 128 starts its accumulators at the bits of 2^23 and reads the sums as floats
    (kernels_filter.h, I8_ACC_BIAS): exact for 0 <= H < 2^23, monotone (halved) for negative H.  Exercise the edges:
      * flat queries (every component at full magnitude: the hi plane's L1 norm would reach 768 * 127 > I8_HI_L1_MAX, so
        prep_q8_kernel quantises them more coarsely), against rows of constant sign at full magnitude (largest |H|);
      * queries whose every score is negative (negative thresholds, negative H everywhere);
      * the in-loop second-stage build (no bias) must log the same survivors."""
    n, k = 150000 + 5, 10
    base, scales = nvdb_amd.synth_corpus(SEED + 90, 0, n, d, nvdb_amd.DT_I8)
    base, scales = base.copy(), scales.copy()
    base[:64] = 127; base[64:128] = -128; base[128:160] = np.where(np.arange(d) % 2 == 0, 127, -127).astype(np.int8)   # extreme rows
    base[200:20000] = np.abs(base[200:20000].astype(np.int16)).clip(0, 127).astype(np.int8)                             # a block of non-negative rows
    queries = nvdb_amd.synth_rows_f32(SEED + 91, 0, nq, d)
    queries[0] = 1.0                                               # flat, positive: H = 127 * L1(hi) against the all-127 rows
    queries[1] = -1.0                                              # flat, negative: largest H against the all-(-128) rows
    queries[2] = np.where(np.arange(d) % 2 == 0, 1.0, -1.0)
    queries[3] = np.sign(queries[3]) * np.float32(3.25)            # flat magnitude, random signs
    queries[4] = -np.abs(queries[4])                               # against the non-negative block: every score there is negative
    queries[5] = np.abs(queries[5]) * np.float32(1e-3)
    ctx.upload_corpus(base, po.DT_I8, scales)
    ctx.set_option("path", 2)
    res, stats = {}, {}
    for var, defer in enumerate((0, 1)):
        ctx.set_option("i8_defer", defer)
        res[var] = ctx.search_batch(queries, k)
        stats[var] = ctx.stats()
        assert stats[var]["path"] == 2 and stats[var]["bound_violations"] == 0 and stats[var]["overflow_queries"] == 0, stats[var]
    ctx.set_option("i8_defer", 0)
    ctx.set_option("path", 0)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
    assert stats[0]["candidates"] == stats[1]["candidates"]
    _check_against_oracle(oracle, base, po.DT_I8, scales, queries, res[0][0], res[0][1], k, f"i8-biased/d{d}")
    # all-negative scores: a corpus of non-negative rows only, a batch of non-positive queries
    sub = np.ascontiguousarray(base[200:20000])
    ssc = np.ascontiguousarray(np.abs(scales[200:20000]))
    qn = -np.abs(nvdb_amd.synth_rows_f32(SEED + 92, 0, 200, d))
    ctx.upload_corpus(sub, po.DT_I8, ssc)
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(qn, k)
    st = ctx.stats()
    ctx.set_option("path", 0)
    assert st["path"] == 2 and st["bound_violations"] == 0, st
    assert np.all(sc < 0)
    _check_against_oracle(oracle, sub, po.DT_I8, ssc, qn, ids, sc, k, f"i8-biased-negative/d{d}")


def test_filter_path_scaled_and_skewed_queries(ctx, oracle):
    """Query scale must not matter (per-query power-of-two prescale), nor heavy-tailed elements."""
    n, d, nq, k = 100000, 768, 48, 10
    ctx.generate_corpus(SEED + 5, n, d, nvdb_amd.DT_F16)
    base, _ = nvdb_amd.synth_corpus(SEED + 5, 0, n, d, nvdb_amd.DT_F16)
    rs = np.random.RandomState(21)
    queries = nvdb_amd.synth_rows_f32(SEED + 6, 0, nq, d)
    queries[:16] *= np.float32(3.7e4)
    queries[16:32] *= np.float32(2.2e-6)
    queries[32:, :5] *= np.float32(40.0)
    queries[40] = 0.0                                        # zero query: every score ties at 0
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(queries, k)
    st = ctx.stats()
    ctx.set_option("path", 0)
    assert st["bound_violations"] == 0, st
    _check_against_oracle(oracle, base, po.DT_F16, None, queries, ids, sc, k, "filter/scaled")


def test_filter_path_with_exact_duplicates(ctx, oracle):
    """Exact-score ties at the k boundary: every tied row must survive the filter; order is id asc."""
    n, d, nq, k = 60000, 768, 32, 10
    base, _ = nvdb_amd.synth_corpus(SEED + 9, 0, n, d, nvdb_amd.DT_F16)
    base = base.copy()
    rs = np.random.RandomState(4)
    for src in rs.randint(0, n, size=40):                    # 40 rows, each copied to 7 other places
        base[rs.randint(0, n, size=7)] = base[src]
    queries = oracle.f16_to_f32(base[rs.randint(0, n, size=nq)])
    ctx.upload_corpus(base, po.DT_F16)
    for path in (1, 2):
        ctx.set_option("path", path)
        ids, sc = ctx.search_batch(queries, k)
        assert ctx.stats()["bound_violations"] == 0
        _check_against_oracle(oracle, base, po.DT_F16, None, queries, ids, sc, k, f"dups/path{path}")
    ctx.set_option("path", 0)


@pytest.mark.parametrize("d,nq", [(384, 40), (128, 200), (256, 33), (512, 150), (640, 300), (640, 40)])
def test_filter_and_exact_paths_agree_on_other_dims(ctx, oracle, d, nq):
    n, k = 80000, 10                                         # 384 is the reference's own data dimension
    ctx.generate_corpus(SEED + 2, n, d, nvdb_amd.DT_F16)
    base, _ = nvdb_amd.synth_corpus(SEED + 2, 0, n, d, nvdb_amd.DT_F16)
    queries = nvdb_amd.synth_rows_f32(SEED + 3, 0, nq, d)
    res = {}
    for path in (1, 2):
        ctx.set_option("path", path)
        res[path] = ctx.search_batch(queries, k)
        assert ctx.stats()["path"] == path
    ctx.set_option("path", 0)
    assert np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1].view(np.uint32), res[2][1].view(np.uint32))
    _check_against_oracle(oracle, base, po.DT_F16, None, queries, res[2][0], res[2][1], k, f"dim{d}")


@pytest.mark.parametrize("tag,d,nq", [("f16", 100, 70), ("f16", 300, 33), ("f16", 600, 200), ("f32", 200, 64), ("f32", 760, 9)])
def test_filter_path_on_zero_padded_shadow_for_odd_dims(ctx, oracle, tag, d, nq):
    """Dims the MFMA kernels are not instantiated for: the filter streams an fp16 shadow whose rows are
    zero-padded to the next instantiated dim; survivors are re-scored from the ORIGINAL rows in the reference's
    order (incl. its tail rules, simd_dot.cpp:26-49, 95-150), so ids and score bits still match the CPU path."""
    n, k = 50000 + 11, 10
    dt = nvdb_amd.DT_F16 if tag == "f16" else nvdb_amd.DT_F32
    ctx.generate_corpus(SEED + 70, n, d, dt)
    base, _ = nvdb_amd.synth_corpus(SEED + 70, 0, n, d, dt)
    queries = nvdb_amd.synth_rows_f32(SEED + 71, 0, nq, d)
    queries[0] = oracle.f16_to_f32(base[777]) if tag == "f16" else base[777]
    res = {}
    for path in (1, 2):
        ctx.set_option("path", path)
        res[path] = ctx.search_batch(queries, k)
        st = ctx.stats()
        assert st["path"] == path and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    ctx.set_option("path", 0)
    assert np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1].view(np.uint32), res[2][1].view(np.uint32))
    _check_against_oracle(oracle, base, po.DT_F16 if tag == "f16" else po.DT_F32, None, queries, res[2][0], res[2][1], k, f"pad/{tag}/d{d}")


@pytest.mark.parametrize("tag,d,nq", [("f16", 1536, 300), ("f16", 1024, 64), ("f16", 1000, 130), ("f32", 1536, 40), ("f16", 1280, 33),
                                      ("f16", 3072, 200), ("f16", 2048, 70), ("f16", 2500, 130), ("f32", 3072, 20), ("f16", 1600, 600),
                                      ("f16", 896, 300), ("f16", 1152, 64), ("f16", 1408, 1024), ("f16", 2560, 100), ("f32", 1280, 150)])
def test_filter_path_for_dims_up_to_1536(ctx, oracle, tag, d, nq):
    """768 < dim <= 1536: the 16-row-tile build of the fp16 kernel (32 queries per wave); 1536 < dim <= 3072: the K-split
    build (16 queries per wave, a tile streamed as two half-K stages); other dims through the zero-padded shadow.  Same
    exact rescore, so ids and score bits match the CPU path."""
    n, k = 60000 + 5, 10
    dt = nvdb_amd.DT_F16 if tag == "f16" else nvdb_amd.DT_F32
    ctx.generate_corpus(SEED + 90, n, d, dt)
    base, _ = nvdb_amd.synth_corpus(SEED + 90, 0, n, d, dt)
    queries = nvdb_amd.synth_rows_f32(SEED + 91, 0, nq, d)
    queries[1] = oracle.f16_to_f32(base[4321]) if tag == "f16" else base[4321]
    res = {}
    for path in (1, 2):
        ctx.set_option("path", path)
        res[path] = ctx.search_batch(queries, k)
        st = ctx.stats()
        assert st["path"] == path and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    ctx.set_option("path", 0)
    assert np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1].view(np.uint32), res[2][1].view(np.uint32))
    _check_against_oracle(oracle, base, po.DT_F16 if tag == "f16" else po.DT_F32, None, queries, res[2][0], res[2][1], k, f"bigdim/{tag}/d{d}")


@pytest.mark.parametrize("tag,d,nq,k", [("f16", 768, 300, 10), ("f16", 768, 1, 10), ("f16", 384, 70, 10), ("f32", 768, 130, 10), ("f16", 1024, 64, 5),
                                        ("f32", 512, 1024, 10), ("f16", 768, 40, 100)])
def test_int8_filter_shadow_of_an_fp16_or_fp32_corpus_stays_exact(oracle, tag, d, nq, k):
    """Option q8_shadow: an fp16 / fp32 corpus is FILTERED through an int8 copy of itself (the reference's per-row quantiser) on the
    integer matrix cores; the rows' largest quantisation residual joins the filter's error bound and every survivor is re-scored from
    the ORIGINAL rows.  Ids and score bits must equal the oracle's fp16 / fp32 answer, the self-check (|filter - exact| <= bound for
    every survivor) must stay silent, and the exact path must agree for every query."""
    n = 150_000 + 17
    base32 = nvdb_amd.synth_rows_f32(SEED + 200, 0, n, d)                # unit-norm rows, like embeddings: every row quantises about equally well
    base32[201] = 0.0
    base32[100:140] *= (np.float32(10.0) ** np.random.RandomState(d).randint(-3, 1, size=(40, 1))).astype(np.float32)   # some much SHORTER rows
    base, dt, scales = _as_dtype(oracle, base32, tag)
    queries = nvdb_amd.synth_rows_f32(SEED + 201, 0, nq, d)
    queries[0] = base32[300]
    c = nvdb_amd.HipContext(0)
    c.set_option("q8_shadow", 1)
    c.upload_corpus(base, dt, scales)
    ids, sc = c.search_batch(queries, k)
    st = c.stats()
    assert st["path"] == 2 and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    c.set_option("path", 1)
    ei, es = c.search_batch(queries, k)
    c.close()
    assert np.array_equal(ids, ei) and np.array_equal(sc.view(np.uint32), es.view(np.uint32))
    sub = np.unique(np.r_[0, nq // 2, nq - 1])
    _check_against_oracle(oracle, base, dt, scales, queries[sub], ids[sub], sc[sub], k, f"q8 shadow/{tag}/d{d}")
    # without the option the same corpus takes the fp16 filter: same answer
    c2 = nvdb_amd.HipContext(0)
    c2.upload_corpus(base, dt, scales)
    i2, s2 = c2.search_batch(queries, k)
    c2.close()
    assert np.array_equal(ids, i2) and np.array_equal(sc.view(np.uint32), s2.view(np.uint32))


def test_int8_filter_shadow_of_rows_that_quantise_badly_stays_exact(oracle):
    """The corpus side of the shadow's error bound is ||q|| x the LARGEST quantisation residual of any row: one row with a dominant
    element (coarse scale) or rows ten times longer than the rest widen the band for every query.  The lists then overflow and the host
    API redoes the batch on the exact path -- slower, never wrong."""
    n, d, nq, k = 120_000, 768, 50, 10
    base32 = nvdb_amd.synth_rows_f32(SEED + 205, 0, n, d)
    base32[100:140] *= np.float32(10.0)                                  # ten times longer rows
    base32[200, 5] = np.float32(0.9)                                     # one dominant element
    base = oracle.f32_to_f16(base32)
    queries = nvdb_amd.synth_rows_f32(SEED + 206, 0, nq, d)
    queries[0] = base32[200]
    c = nvdb_amd.HipContext(0)
    c.set_option("q8_shadow", 1)
    c.upload_corpus(base, nvdb_amd.DT_F16)
    ids, sc = c.search_batch(queries, k)
    assert c.stats()["bound_violations"] == 0
    c.close()
    sub = np.r_[0:3, nq - 1]
    _check_against_oracle(oracle, base, po.DT_F16, None, queries[sub], ids[sub], sc[sub], k, "q8 shadow/badly quantising rows")


def test_int8_filter_shadow_with_near_duplicates_falls_back_but_stays_exact(oracle):
    """Thousands of rows inside the (wider) error band of the int8 shadow: the candidate lists overflow, the host API redoes the batch with
    the longest lists / on the exact path -- the answer stays the oracle's."""
    n, d, nq, k = 200_000, 768, 20, 10
    rs = np.random.RandomState(5)
    centre = nvdb_amd.synth_rows_f32(SEED + 210, 0, 1, d)[0]
    base32 = nvdb_amd.synth_rows_f32(SEED + 211, 0, n, d)
    base32[:6000] = centre + np.float32(2e-4) * rs.randn(6000, d).astype(np.float32)       # 6000 near-duplicates of one direction
    base = oracle.f32_to_f16(base32)
    queries = nvdb_amd.synth_rows_f32(SEED + 212, 0, nq, d)
    queries[0] = centre
    c = nvdb_amd.HipContext(0)
    c.set_option("q8_shadow", 1)
    c.upload_corpus(base, nvdb_amd.DT_F16)
    ids, sc = c.search_batch(queries, k)
    assert c.stats()["bound_violations"] == 0
    c.close()
    _check_against_oracle(oracle, base, po.DT_F16, None, queries[:3], ids[:3], sc[:3], k, "q8 shadow/near-duplicates")


@pytest.mark.parametrize("tag", ["f16", "i8"])
def test_non_finite_queries_take_the_exact_path(ctx, oracle, tag):
    """A query with a NaN or an infinite element has no usable filter bound: it is flagged like a list overflow and
    its sub-batch is redone on the exact kernel.  The finite queries of the batch still match the CPU path."""
    n, d, nq, k = 40000, 768, 24, 10
    dt = nvdb_amd.DT_F16 if tag == "f16" else nvdb_amd.DT_I8
    ctx.generate_corpus(SEED + 97, n, d, dt)
    base, scales = nvdb_amd.synth_corpus(SEED + 97, 0, n, d, dt)
    queries = nvdb_amd.synth_rows_f32(SEED + 98, 0, nq, d)
    queries[3, 5] = np.nan
    queries[7, 100] = np.inf
    queries[9] *= np.float32(3e37)                            # finite elements, norm^2 beyond fp32
    ctx.set_option("path", 1)
    ei, es = ctx.search_batch(queries, k)
    ctx.set_option("path", 2)
    fi, fs = ctx.search_batch(queries, k)
    st = ctx.stats()
    ctx.set_option("path", 0)
    assert st["overflow_queries"] >= 3 and st["bound_violations"] == 0, st      # flagged -> sub-batch redone on the exact path
    assert np.array_equal(fi, ei) and np.array_equal(fs.view(np.uint32), es.view(np.uint32))
    good = [i for i in range(nq) if i not in (3, 7, 9)]
    _check_against_oracle(oracle, base, po.DT_F16 if tag == "f16" else po.DT_I8, scales, queries[good], fi[good], fs[good], k, "nonfinite/rest")


@pytest.mark.parametrize("tag,d", [("f16", 768), ("i8", 768), ("f16", 384)])
def test_corpus_stored_in_cluster_order_does_not_flood_the_lists(ctx, oracle, tag, d):
    """Rows grouped by cluster (a corpus ingested topic by topic): a query's whole cluster sits in one stretch of rows.
    In storage order the last chunks would meet thousands of rows above thresholds learnt from unrelated clusters; the
    filter kernels stream the tiles in a permuted order instead, so every chunk is a fair sample.  No overflow, and
    ids / score bits equal the CPU path.  (Also the case that showed why thresholds must never fall: the bootstrap
    rows are not part of the first chunk any more.)"""
    n, nq, k, C = 160000 + 33, 300, 10, 40
    rs = np.random.RandomState(12)
    cent = rs.randn(C, d).astype(np.float32); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    cid = np.arange(n) * C // n                               # cluster-ordered
    x = cent[cid] + np.float32(0.5) * rs.randn(n, d).astype(np.float32) / np.float32(np.sqrt(d))
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    if tag == "f16":
        base, dt, scales = oracle.f32_to_f16(x), po.DT_F16, None
    else:
        (base, scales), dt = oracle.quantize_i8(x), po.DT_I8
    q = cent[rs.randint(0, C, size=nq)] + np.float32(0.5) * rs.randn(nq, d).astype(np.float32) / np.float32(np.sqrt(d))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    ctx.upload_corpus(base, dt, scales)
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(q, k)
    st = ctx.stats()
    ctx.set_option("tile_permute", 0)
    ctx.search_batch(q, k)
    st_storage_order = ctx.stats()
    ctx.set_option("tile_permute", 1)
    ctx.set_option("path", 0)
    assert st["path"] == 2 and st["overflow_queries"] == 0 and st["bound_violations"] == 0, st
    assert st_storage_order["overflow_queries"] > 0, st_storage_order          # what the permutation is for
    _check_against_oracle(oracle, base, dt, scales, q[:40], ids[:40], sc[:40], k, f"cluster-order/{tag}/d{d}")


def test_near_duplicate_clusters_overflow_the_short_lists_but_stay_exact(ctx, oracle):
    """Tight clusters: thousands of rows lie inside the filter's error band of the k-th score, more than the default
    2048-entry lists hold.  The search is redone with the longest lists (and, failing that, on the exact path); either
    way ids and score bits equal the CPU path."""
    n, d, nq, k, C = 150000, 768, 200, 10, 40               # ~3750 rows per cluster
    rs = np.random.RandomState(11)
    cent = rs.randn(C, d).astype(np.float32); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    x = cent[rs.randint(0, C, size=n)] + np.float32(0.05) * rs.randn(n, d).astype(np.float32) / np.float32(np.sqrt(d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    base = oracle.f32_to_f16(x.astype(np.float32))
    q = cent[rs.randint(0, C, size=nq)] + np.float32(0.05) * rs.randn(nq, d).astype(np.float32) / np.float32(np.sqrt(d))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    ctx.upload_corpus(base, po.DT_F16)
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(q, k)
    st = ctx.stats()
    ctx.set_option("path", 0)
    assert st["overflow_queries"] > 0 and st["bound_violations"] == 0, st       # the first attempt did overflow
    _check_against_oracle(oracle, base, po.DT_F16, None, q[:48], ids[:48], sc[:48], k, "near-dup clusters")


@pytest.mark.parametrize("permute", [0, 1])
def test_overflow_falls_back_to_exact_path(ctx, oracle, permute):
    """Candidate lists that overflow (tiny lists here; with the tiles in storage order also the adversarial order: rows
    sorted by score ascending for query 0): the library must notice, redo the batch with longer lists or on the exact
    path, and still return the exact answer."""
    n, d, k = 40000, 768, 10
    base32 = nvdb_amd.synth_rows_f32(SEED + 11, 0, n, d)
    q = nvdb_amd.synth_rows_f32(SEED + 12, 0, 16, d)
    order = np.argsort(base32 @ q[0])                        # ascending similarity to query 0
    base = oracle.f32_to_f16(base32[order])
    ctx.upload_corpus(base, po.DT_F16)
    ctx.set_option("path", 2)
    ctx.set_option("cand_cap", 64)
    ctx.set_option("tile_permute", permute)
    ids, sc = ctx.search_batch(q, k)
    st = ctx.stats()
    ctx.set_option("cand_cap", 0)
    ctx.set_option("tile_permute", 1)
    ctx.set_option("path", 0)
    assert st["overflow_queries"] >= 1
    _check_against_oracle(oracle, base, po.DT_F16, None, q, ids, sc, k, "overflow")


# ----------------------------------------------------------------------------- sharding (multi-GPU logic on one GPU)
def test_row_sharded_search_merges_to_the_unsharded_answer(oracle):
    n, d, nq, k, shards = 120000, 768, 24, 10, 3
    full = nvdb_amd.HipContext(0)
    full.generate_corpus(SEED + 20, n, d, nvdb_amd.DT_F16)
    queries = nvdb_amd.synth_rows_f32(SEED + 21, 0, nq, d)
    fi, fs = full.search_batch(queries, k)
    parts_i, parts_s = [], []
    for s in range(shards):
        lo, hi = n * s // shards, n * (s + 1) // shards
        c = nvdb_amd.HipContext(0)
        c.generate_corpus(SEED + 20, hi - lo, d, nvdb_amd.DT_F16, row_base=lo)   # global ids = base + local
        i, sc = c.search_batch(queries, k)
        parts_i.append(i)
        parts_s.append(sc)
        c.close()
    mi, ms = nvdb_amd.merge_topk_host(np.stack(parts_i), np.stack(parts_s))
    assert np.array_equal(mi, fi) and np.array_equal(ms.view(np.uint32), fs.view(np.uint32))
    full.close()


def test_device_merge_matches_host_merge(ctx):
    rs = np.random.RandomState(8)
    S, nq, k = 8, 50, 10
    sc = np.sort(rs.rand(S, nq, k).astype(np.float32), axis=2)[:, :, ::-1].copy()
    sc[:, :, 3] = sc[:, :, 2]                                 # ties inside and across shards
    sc[1] = sc[0]
    ids = (rs.permutation(S * nq * k).astype(np.uint64)).reshape(S, nq, k)
    hi, hs = nvdb_amd.merge_topk_host(ids, sc)
    import torch
    dev = torch.device("cuda", 0)
    d_ids, d_sc = torch.from_numpy(ids.view(np.int64)).to(dev), torch.from_numpy(sc).to(dev)
    d_oi = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    d_os = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.merge_topk_dev(d_ids.data_ptr(), d_sc.data_ptr(), S, nq, k, d_oi.data_ptr(), d_os.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    oi, os_ = d_oi.cpu().numpy().view(np.uint64), d_os.cpu().numpy()
    assert np.array_equal(oi, hi) and np.array_equal(os_.view(np.uint32), hs.view(np.uint32))


@pytest.mark.parametrize("S,k,nq", [(8, 1000, 33), (2, 3000, 7), (5, 900, 4), (8, 513, 1)])
def test_device_merge_of_long_sorted_lists_matches_host_merge(ctx, S, k, nq):
    """nshards * k > 4096 (beyond the LDS-ranked merge): merge_topk_sorted_kernel, one binary search per other shard over lists
    that arrive sorted by (score desc, id asc) with their padding last -- cross-shard score ties, duplicated whole lists' scores
    and short (padded) shards included.  The reference bounds k by N only (src/flat_index.cpp:24)."""
    import torch
    rs = np.random.RandomState(S * k)
    sc = rs.randint(0, 4 * k, size=(S, nq, k)).astype(np.float32) / np.float32(4 * k)     # coarse grid: many equal scores across shards
    ids = rs.permutation(S * nq * k).astype(np.uint64).reshape(S, nq, k)                   # distinct global ids
    if S > 1:
        sc[1] = sc[0]
    for s_ in range(S):                                         # every list sorted (score desc, id asc)
        for q in range(nq):
            o = np.lexsort((ids[s_, q], -sc[s_, q].astype(np.float64)))
            sc[s_, q], ids[s_, q] = sc[s_, q][o], ids[s_, q][o]
    pad = k // 3                                                # last shard holds fewer than k rows: padding (id ~0, -inf) at the end
    sc[S - 1, :, k - pad:] = -np.inf
    ids[S - 1, :, k - pad:] = np.iinfo(np.uint64).max
    if S > 2:                                                   # ... and so does another one: equal pads from two shards
        sc[S - 2, :, k - 5:] = -np.inf
        ids[S - 2, :, k - 5:] = np.iinfo(np.uint64).max
    hi, hs = nvdb_amd.merge_topk_host(ids, sc)
    dev = torch.device("cuda", 0)
    d_ids, d_sc = torch.from_numpy(ids.view(np.int64)).to(dev), torch.from_numpy(sc).to(dev)
    d_oi = torch.zeros((nq, k), dtype=torch.int64, device=dev)
    d_os = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.merge_topk_dev(d_ids.data_ptr(), d_sc.data_ptr(), S, nq, k, d_oi.data_ptr(), d_os.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    oi, os_ = d_oi.cpu().numpy().view(np.uint64), d_os.cpu().numpy()
    assert np.array_equal(oi, hi) and np.array_equal(os_.view(np.uint32), hs.view(np.uint32))


# ----------------------------------------------------------------------------- refine
@pytest.mark.parametrize("tag,d,R,K", [("f16", 768, 1024, 10), ("f16", 384, 500, 10), ("f32", 768, 300, 64),
                                       ("f16", 100, 77, 5), ("f32", 37, 40, 3), ("f16", 1536, 1024, 10), ("f16", 1024, 333, 64),
                                       ("f16", 1536, 17, 3)])
def test_refine_matches_restated_reference_order(ctx, oracle, tag, d, R, K):
    n, Q = 30000, 40
    rs = np.random.RandomState(d + R)
    base32 = nvdb_amd.synth_rows_f32(SEED + 30, 0, n, d)
    base, dt = (oracle.f32_to_f16(base32), po.DT_F16) if tag == "f16" else (base32, po.DT_F32)
    queries = nvdb_amd.synth_rows_f32(SEED + 31, 0, Q, d)
    cand = rs.randint(0, n, size=(Q, R)).astype(np.uint32)
    cand[rs.rand(Q, R) < 0.01] = 0xFFFFFFFF                 # skipped slots (nvdb_ivf_eval.cpp:513-516)
    cand[0, 1] = n + 5                                      # out of range -> skipped (cuda_refine.cu:437)
    cand[1, :] = 0xFFFFFFFF                                 # no valid candidate at all -> all padding
    cand[2, 5:] = 0xFFFFFFFF                                # fewer valid candidates than K (when K > 5)
    cand[3, 10:20] = cand[3, 0]                             # duplicates of one id stay separate entries
    ctx.upload_corpus(base, dt)
    ids, dist, t = ctx.refine_l2_topk(queries, cand, K, want_timing=True)
    oid, odist = oracle.refine(base, dt, queries, cand, K, mode=0)
    assert np.array_equal(ids, oid)
    assert np.array_equal(dist.view(np.uint32), odist.view(np.uint32))
    assert (ids[1] == 0xFFFFFFFF).all() and (dist[1] == np.float32(1e30)).all()
    # against the reference's CPU refine (double accumulation): distances to fp32 rounding
    cid, cdist = oracle.refine(base, dt, queries, cand, K, mode=1)
    valid = ids != 0xFFFFFFFF
    assert np.allclose(dist[valid], cdist[valid], rtol=1e-5, atol=1e-7)
    assert t.K == K and t.R == R and t.kernel_ms > 0 and abs(t.total_ms - (t.h2d_ms + t.kernel_ms + t.d2h_ms)) < 1e-3
    ids_only, none = ctx.refine_l2_topk(queries, cand, K, want_dist=False)   # CUDA_RETURN_DIST=0
    assert none is None and np.array_equal(ids_only, ids)
    ctx.set_option("refine_pinned", 1)                                       # CUDA_PINNED=1 (cuda_refine.cu:875, 902-914)
    pi, pd = ctx.refine_l2_topk(queries, cand, K)
    ctx.set_option("refine_pinned", 0)
    assert np.array_equal(pi, ids) and np.array_equal(pd.view(np.uint32), dist.view(np.uint32))


def test_refine_argument_conventions(ctx, oracle):
    base = oracle.f32_to_f16(nvdb_amd.synth_rows_f32(SEED, 0, 1000, 64))
    ctx.upload_corpus(base, po.DT_F16)
    q = nvdb_amd.synth_rows_f32(SEED + 1, 0, 2, 64)
    ids, dist = ctx.refine_l2_topk(q, np.zeros((2, 0), dtype=np.uint32), 10)      # R == 0 -> nothing to do
    assert (ids == 0xFFFFFFFF).all()
    with pytest.raises(nvdb_amd.NvdbError):
        ctx.refine_l2_topk(q, np.zeros((2, 4), dtype=np.uint32), 65)             # K > 64 (cuda_refine.cu:858-862)
    b8, sc = oracle.quantize_i8(nvdb_amd.synth_rows_f32(SEED, 0, 100, 64))
    ctx.upload_corpus(b8, po.DT_I8, sc)
    with pytest.raises(nvdb_amd.NvdbError):
        ctx.refine_l2_topk(q, np.zeros((2, 4), dtype=np.uint32), 3)              # int8 base unsupported (nvdb_ivf_eval.cpp:519-525)


def test_non_power_of_two_candidate_capacity(ctx, oracle):
    """cand_cap = 3000 with lists longer than 2048 entries: the bitonic select pads to 4096 entries, its LDS must be
    sized for that (not for cap).  Near-duplicate rows make the lists long; results stay exact."""
    n, d, nq, k = 60000, 768, 16, 10
    rs = np.random.RandomState(17)
    base32 = nvdb_amd.synth_rows_f32(SEED + 40, 0, n, d)
    centre = base32[7].copy()
    dup = rs.choice(n, 2600, replace=False)
    base32[dup] = centre + rs.standard_normal((2600, d)).astype(np.float32) * np.float32(2e-5)     # 2600 rows inside the filter's error band
    base = oracle.f32_to_f16(base32)
    queries = nvdb_amd.synth_rows_f32(SEED + 41, 0, nq, d)
    queries[0] = centre
    ctx.upload_corpus(base, po.DT_F16)
    ctx.set_option("path", 2)
    ctx.set_option("cand_cap", 3000)
    ids, sc = ctx.search_batch(queries, k)
    st = ctx.stats()
    ctx.set_option("cand_cap", 0)
    ctx.set_option("path", 0)
    assert st["bound_violations"] == 0, st
    oid, osc = oracle.flat_topk(base, po.DT_F16, queries, k)
    for qi in range(nq):
        allsc = oracle.scores(base, po.DT_F16, queries[qi], None)
        assert_topk_equal(ids[qi], sc[qi], oid[qi], osc[qi], score_of=lambda i: allsc[i], what=f"cap3000/q{qi}")


# ----------------------------------------------------------------------------- any k (reference: k clamped to N only)
@pytest.mark.parametrize("tag", ["f32", "f16", "i8"])
@pytest.mark.parametrize("name,k", [("main768", 65), ("main768", 100), ("main768", 1000), ("main768", 3000), ("main768", 5000),
                                    ("ties64", 100), ("ties64", 600), ("tail100", 699)])
def test_any_k_matches_oracle(ctx, oracle, name, k, tag):
    """k > 64 (src/flat_index.cpp:24 clamps k to N and nothing else): scores of every row in the reference's fp32 order,
    radix select of the k-th (score desc, id asc) key, sort.  Golden-case inputs incl. exact duplicates (ties64),
    k == N and k > N."""
    base32, queries = make_case_inputs(name)
    base, dt, scales = _as_dtype(oracle, base32, tag)
    ctx.upload_corpus(base, dt, scales)
    ids, sc = ctx.search_batch(queries, k)
    assert ctx.stats()["path"] == 3
    oid, osc = oracle.flat_topk(base, dt, queries, k, scales)
    assert ids.shape == oid.shape == (len(queries), min(k, len(base)))
    for qi in range(len(queries)):
        allsc = oracle.scores(base, dt, queries[qi], scales)
        assert_topk_equal(ids[qi], sc[qi], oid[qi], osc[qi], score_of=lambda i: allsc[i], what=f"anyk/{name}/{tag}/k{k}/q{qi}")
        order = np.lexsort((ids[qi], -sc[qi].astype(np.float64)))
        assert np.array_equal(order, np.arange(ids.shape[1])), "canonical (score desc, id asc) order"


@pytest.mark.parametrize("tag,k,nq,d", [("f16", 100, 300, 768), ("f16", 1000, 200, 768), ("i8", 100, 256, 768), ("i8", 1024, 40, 768), ("f32", 200, 64, 768),
                                        ("f16", 100, 200, 1024), ("f16", 300, 70, 1536), ("i8", 100, 130, 1024), ("f16", 128, 40, 3072), ("f32", 100, 64, 1280),
                                        ("i8", 65, 33, 1408)])
def test_k_up_to_1024_rides_the_filter_path(oracle, tag, k, nq, d):
    """64 < k <= 1024: the MFMA filter's lists (8192 entries), a bootstrap over 8k tile maxima and small chunks; results must
    equal the any-k path (forced with path = 1) for every query and the oracle for a few.  768 < dim (kernels without an MFMA
    bootstrap build): the bootstrap is exact -- the any-k machinery on the first 8k tiles' rows seeds the lists."""
    n = 600_000 + 7 if d <= 1536 else 200_000 + 7
    dt = {"f16": nvdb_amd.DT_F16, "i8": nvdb_amd.DT_I8, "f32": nvdb_amd.DT_F32}[tag]
    c = nvdb_amd.HipContext(0)
    c.generate_corpus(SEED + 7, n, d, dt)
    queries = nvdb_amd.synth_rows_f32(SEED + 8, 0, nq, d)
    ids, sc = c.search_batch(queries, k)
    st = c.stats()
    assert st["path"] == 2 and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    c.set_option("path", 1)
    ide, sce = c.search_batch(queries, k)
    assert c.stats()["path"] == 3
    c.set_option("path", 0)
    assert np.array_equal(ids, ide) and np.array_equal(sc.view(np.uint32), sce.view(np.uint32))
    base, scales = c.download_rows(0, n)
    sub = [0, nq // 2, nq - 1]
    oid, osc = oracle.flat_topk(base, {"f16": po.DT_F16, "i8": po.DT_I8, "f32": po.DT_F32}[tag], queries[sub], k, scales)
    assert np.array_equal(ids[sub], oid) and np.array_equal(sc[sub].view(np.uint32), osc.view(np.uint32))
    c.close()


def test_any_k_large_lists_sorted_in_global_memory(oracle):
    """k = 20 000 of 50 000 rows (lists longer than the 8192 entries LDS sorts), 11 queries in sub-batches of 4 (tiny
    score-matrix budget), and the device-buffer entry point."""
    n, d, nq, k = 50_000, 128, 11, 20_000
    c = nvdb_amd.HipContext(0)
    c.generate_corpus(SEED + 5, n, d, nvdb_amd.DT_F16)
    base, _ = nvdb_amd.synth_corpus(SEED + 5, 0, n, d, nvdb_amd.DT_F16)
    queries = nvdb_amd.synth_rows_f32(SEED + 6, 0, nq, d)
    c.set_option("largek_budget_mb", 1)
    ids, sc = c.search_batch(queries, k)
    st = c.stats()
    assert st["path"] == 3 and st["chunks"] >= 3, st
    oid, osc = oracle.flat_topk(base, po.DT_F16, queries, k)
    assert np.array_equal(ids, oid) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
    c.close()


# ----------------------------------------------------------------------------- BASELINE full sizes: size-independent properties
@pytest.mark.parametrize("dtype,tag", [(nvdb_amd.DT_F16, "f16"), (nvdb_amd.DT_I8, "i8")])
def test_full_size_10M_properties(oracle, dtype, tag):
    """N = 10M, d = 768, B = 1024, k = 10 (BASELINE configs[1] / [2]).  The oracle cannot scan 10M x 1024 in
    seconds, so parity is checked through properties that do not depend on size:
      (a) every returned score is, bit for bit, the reference CPU score of the returned row (oracle dot on
          rows copied back from HBM) and lists are in canonical (score desc, id asc) order;
      (b) no row of a 200K-row sample beats a query's k-th score (oracle scan of the sample);
      (c) sharding: top-k of [0,6M) merged with top-k of [6M,10M) == top-k of [0,10M);
      (d) idempotence, and the MFMA path == the exact fp32-order kernel for ALL 1024 queries of the batch."""
    n, d, nq, k = 10_000_000, 768, 1024, 10
    dt_o = po.DT_F16 if dtype == nvdb_amd.DT_F16 else po.DT_I8
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(SEED, n, d, dtype)
    queries = nvdb_amd.synth_rows_f32(SEED + 1, 0, nq, d)
    ids, sc = ctx.search_batch(queries, k)
    st = ctx.stats()
    assert st["path"] == 2 and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    assert ids.shape == (nq, k)
    # (a)
    for qi in range(0, nq, 97):
        for j in range(k):
            row, rsc = ctx.download_rows(int(ids[qi, j]), 1)
            if dtype == nvdb_amd.DT_F16:
                s = oracle.lib.oracle_dot_f32_f16base(po._p(queries[qi], po._f32p), row.ctypes.data, d)
            else:
                s = oracle.lib.oracle_dot_f32_i8base(po._p(queries[qi], po._f32p), row.ctypes.data, d, float(rsc[0]))
            assert np.float32(s).view(np.uint32) == sc[qi, j].view(np.uint32), (qi, j)
    assert np.all((sc[:, :-1] > sc[:, 1:]) | ((sc[:, :-1] == sc[:, 1:]) & (ids[:, :-1] < ids[:, 1:])))
    assert all(len(set(r.tolist())) == k for r in ids)
    # (b)
    lo = 4_321_000
    sample, ssc = ctx.download_rows(lo, 200_000)
    for qi in (0, 511, 1023):
        allsc = oracle.scores(sample, dt_o, queries[qi], ssc)
        better = np.flatnonzero(allsc > sc[qi, -1]) + lo
        assert set(better.tolist()) <= set(ids[qi].tolist()), qi
    # (d) idempotence + exact kernel on a subset
    ids2, sc2 = ctx.search_batch(queries, k)
    assert np.array_equal(ids, ids2) and np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))
    ctx.set_option("path", 1)                                 # ALL 1024 queries on the exact fp32-order kernel (~0.8 s)
    ide, sce = ctx.search_batch(queries, k)
    assert ctx.stats()["path"] == 1
    ctx.set_option("path", 0)
    assert np.array_equal(ide, ids) and np.array_equal(sce.view(np.uint32), sc.view(np.uint32))
    ctx.close()
    # (c)
    parts = []
    for lo_, hi_ in ((0, 6_000_000), (6_000_000, n)):
        c = nvdb_amd.HipContext(0)
        c.generate_corpus(SEED, hi_ - lo_, d, dtype, row_base=lo_)
        parts.append(c.search_batch(queries, k))
        c.close()
    mi, ms = nvdb_amd.merge_topk_host(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert np.array_equal(mi, ids) and np.array_equal(ms.view(np.uint32), sc.view(np.uint32))


def test_int8_recall_and_score_deltas_against_fp32_at_10M():
    """BASELINE configs[2]: "INT8(+scale) flat-scan top-10, N=10M d=768, batch=1024 ... recall vs fp32 reported" -- at the
    configuration's own size, all 1024 queries.  The int8 corpus is the fp32 corpus quantised per row by the reference's rule
    (apps/nvdb_quantize_i8.cpp:71-80; the device generator and nvdb_quantize_i8_rows are byte-compared with the reference tool in
    test_converters_match_reference_goldens), so what is measured is the quantiser's loss, not the kernels': the int8 ids are the
    exact-kernel int8 ids (asserted), the fp32 ids come from exact fp32-order scores of the fp32 rows."""
    n, d, nq, k = 10_000_000, 768, 1024, 10
    queries = nvdb_amd.synth_rows_f32(SEED + 1, 0, nq, d)
    c32 = nvdb_amd.HipContext(0)
    c32.generate_corpus(SEED, n, d, nvdb_amd.DT_F32)
    g_ids, g_sc = c32.search_batch(queries, k)
    assert c32.stats()["path"] == 2 and c32.stats()["bound_violations"] == 0
    c32.close()
    c8 = nvdb_amd.HipContext(0)
    c8.generate_corpus(SEED, n, d, nvdb_amd.DT_I8)
    i_ids, i_sc = c8.search_batch(queries, k)
    assert c8.stats()["path"] == 2
    c8.set_option("path", 1)
    e_ids, e_sc = c8.search_batch(queries, k)                  # the exact fp32-order kernel over the int8 rows, all 1024 queries
    c8.close()
    assert np.array_equal(i_ids, e_ids) and np.array_equal(i_sc.view(np.uint32), e_sc.view(np.uint32))
    recall = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(g_ids, i_ids)]))
    assert recall >= 0.97, recall
    same = i_ids[:, :, None] == g_ids[:, None, :]
    delta = np.abs(i_sc[:, :, None].astype(np.float64) - g_sc[:, None, :].astype(np.float64))[same]
    # unit rows, |q| = 1: a row's quantisation error is ~ scale / sqrt(12) per element, scale = max|x| / 127 ~ 1e-3
    assert delta.size >= 0.97 * nq * k and delta.max() < 2e-3 and delta.mean() < 4e-4, (delta.max(), delta.mean())


def test_xcd_balanced_partition_tiles_the_corpus_and_leaves_results_unchanged(oracle):
    """The persistent filter kernels give the row streams of XCD label x a share of the tiles proportional to that XCD's
    measured speed (kernels_filter.h stream_tile_range / record_xcd_speed; weights adapted by select_kernel).
      (1) the device's partition function tiles [0, T) exactly -- no tile skipped, none visited twice -- for any weights
          in the cage, any T and any stream count (developer library: the function itself, evaluated on the GPU);
      (2) searches with the balance off, on, and on again after the weights have adapted return identical bits;
      (3) the adapted weights stay inside the cage and average 1."""
    import ctypes as C
    c = nvdb_amd.HipContext(0, dev=True)
    u32p, f32p = C.POINTER(C.c_uint32), C.POINTER(C.c_float)

    def ranges(T, S, w):
        lo, hi, wout = np.empty(S, np.uint32), np.empty(S, np.uint32), np.empty(8, np.float32)
        st = c.lib.nvdb_hip_debug_tile_ranges(c.h, T, S, w.ctypes.data_as(f32p) if w is not None else None,
                                              lo.ctypes.data_as(u32p), hi.ctypes.data_as(u32p), wout.ctypes.data_as(f32p))
        assert st == 0, c.lib.nvdb_hip_last_error(c.h)
        return lo.astype(np.int64), hi.astype(np.int64), wout
    rs = np.random.RandomState(5)
    for T, S in ((78125, 64), (781250, 256), (4883, 256), (300, 64), (7, 8), (1_000_003, 512), (0, 8), (4_294_967, 8), (12345, 24), (999, 12)):
        for trial in range(5):
            w = np.ones(8, np.float32) if trial == 0 else rs.uniform(0.9, 1.1, 8).astype(np.float32)
            if trial == 4:
                w = np.array([0.9, 1.1] * 4, np.float32)
            lo, hi, _ = ranges(T, S, w)
            assert lo[0] == 0 and hi[-1] == T and np.array_equal(lo[1:], hi[:-1]) and np.all(hi >= lo), (T, S, trial)
            if trial == 0 or S % 8:                                  # equal weights (or no XCD labels): the plain equal split, give or take a rounding
                assert np.all(np.abs(lo - (T * np.arange(S, dtype=np.int64)) // S) <= 1), (T, S)
            elif T >= 64 * S:                                        # shares follow the weights
                per = (hi - lo).reshape(-1, 8).sum(axis=0).astype(np.float64)
                assert np.allclose(per / per.sum(), w.astype(np.float64) / w.astype(np.float64).sum(), atol=2e-3), (T, S, trial)
    n, d, nq, k = 2_000_000, 768, 1024, 10
    queries = nvdb_amd.synth_rows_f32(SEED + 1, 0, nq, d)
    for dtype in (nvdb_amd.DT_F16, nvdb_amd.DT_I8):
        c.generate_corpus(SEED, n, d, dtype)
        c.set_option("xcd_balance", 0)
        ids0, sc0 = c.search_batch(queries, k)
        assert c.stats()["path"] == 2
        c.set_option("xcd_balance", 1)
        for rep in range(4):
            ids, sc = c.search_batch(queries, k)
            st = c.stats()
            assert st["path"] == 2 and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
            assert np.array_equal(ids, ids0) and np.array_equal(sc.view(np.uint32), sc0.view(np.uint32)), (dtype, rep)
        _, _, w = ranges(1000, 64, None)
        assert np.all(w >= 0.88) and np.all(w <= 1.12) and abs(float(w.mean()) - 1.0) < 1e-3, w
    c.close()


def _rescore_returned_rows(oracle, ctx, queries, ids, sc, d, every):
    """(a) of the full-size properties: every `every`-th query's returned scores are, bit for bit, the reference CPU
    score (oracle dot, simd_dot.cpp:102-124) of the returned rows copied back from HBM."""
    base = ctx.corpus_info()["row_base"]
    for qi in range(0, len(queries), every):
        for j in range(ids.shape[1]):
            row, _ = ctx.download_rows(int(ids[qi, j]) - base, 1)
            s = oracle.lib.oracle_dot_f32_f16base(po._p(queries[qi], po._f32p), row.ctypes.data, d)
            assert np.float32(s).view(np.uint32) == sc[qi, j].view(np.uint32), (qi, j)


def test_full_size_100M_properties(oracle):
    """BASELINE configs[3]: fp16, N = 100M, d = 768, B = 1024, k = 10 -- the whole corpus (153.6 GB) resident on ONE
    GPU, and the same rows as eight 12.5M-row shards (what each of 8 GPUs holds) merged on the device.
      (a) returned scores == oracle dot of the returned rows (bits), lists canonical, ids distinct and < N;
      (b) MFMA filter path == exact fp32-order kernel (src/flat_index.cpp:16-48 semantics) for ALL 1024 queries;
      (c) eight shards generated one after another with their global row base, each searched, lists merged with
          nvdb_hip_merge_topk_dev == the unsharded answer bit for bit (ids and scores);
      (d) no row of a 200K-row sample from the last 4 GiB of the corpus beats a query's k-th score."""
    import torch
    n, d, nq, k, shards = 100_000_000, 768, 1024, 10, 8
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(SEED, n, d, nvdb_amd.DT_F16)
    queries = nvdb_amd.synth_rows_f32(SEED + 1, 0, nq, d)
    ids, sc = ctx.search_batch(queries, k)
    st = ctx.stats()
    assert st["path"] == 2 and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    assert ids.shape == (nq, k) and int(ids.max()) < n
    assert (ids > 2**32 // 1536).any() and (ids > 90_000_000).any()   # results from beyond 4 GiB and beyond 128 GiB of row offsets
    _rescore_returned_rows(oracle, ctx, queries, ids, sc, d, every=97)
    assert np.all((sc[:, :-1] > sc[:, 1:]) | ((sc[:, :-1] == sc[:, 1:]) & (ids[:, :-1] < ids[:, 1:])))
    assert all(len(set(r.tolist())) == k for r in ids)
    lo = n - 2_500_000
    sample, _ = ctx.download_rows(lo, 200_000)
    for qi in (3, 700):
        allsc = oracle.scores(sample, po.DT_F16, queries[qi], None)
        better = np.flatnonzero(allsc > sc[qi, -1]) + lo
        assert set(better.tolist()) <= set(ids[qi].tolist()), qi
    del sample
    ctx.set_option("path", 1)
    ide, sce = ctx.search_batch(queries, k)                    # 1024 queries x 100M rows on the exact kernel (~8 s)
    assert ctx.stats()["path"] == 1
    assert np.array_equal(ide, ids) and np.array_equal(sce.view(np.uint32), sc.view(np.uint32))
    ctx.close()
    # (c) what 8 GPUs would each do, one shard at a time on this one, then the device merge of the gathered lists
    dev = torch.device("cuda", 0)
    g_ids = torch.empty((shards, nq, k), dtype=torch.int64, device=dev)
    g_sc = torch.empty((shards, nq, k), dtype=torch.float32, device=dev)
    t_q = torch.from_numpy(queries).to(dev)
    stream = torch.cuda.Stream()
    from nvdb_amd.sharding import shard_range
    c = nvdb_amd.HipContext(0)
    for s in range(shards):
        lo_, hi_ = shard_range(n, s, shards)
        assert hi_ - lo_ == 12_500_000
        c.generate_corpus(SEED, hi_ - lo_, d, nvdb_amd.DT_F16, row_base=lo_)
        with torch.cuda.stream(stream):
            c.search_batch_dev(t_q.data_ptr(), nq, k, g_ids[s].data_ptr(), g_sc[s].data_ptr(), stream.cuda_stream)
        stream.synchronize()
        s_st = c.search_check()
        assert s_st["path"] == 2 and s_st["bound_violations"] == 0 and s_st["overflow_queries"] == 0, (s, s_st)
    m_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    m_sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
    c.merge_topk_dev(g_ids.data_ptr(), g_sc.data_ptr(), shards, nq, k, m_ids.data_ptr(), m_sc.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    c.close()
    mi, ms = m_ids.cpu().numpy().view(np.uint64), m_sc.cpu().numpy()
    assert np.array_equal(mi, ids) and np.array_equal(ms.view(np.uint32), sc.view(np.uint32))
    hi_m, hs_m = nvdb_amd.merge_topk_host(g_ids.cpu().numpy().view(np.uint64), g_sc.cpu().numpy())
    assert np.array_equal(hi_m, ids) and np.array_equal(hs_m.view(np.uint32), sc.view(np.uint32))


def _config5_candidates(rs, n, Q, R, hot_lo):
    """SURVEY 8(d) config 5 candidate rule without FAISS: random distinct-ish ids, 1 % of the slots 0xFFFFFFFF; here at
    least half of every list is drawn from rows >= hot_lo (byte offsets beyond 4 GiB)."""
    cand = rs.randint(0, n, size=(Q, R)).astype(np.uint32)
    hot = rs.rand(Q, R) < 0.55
    cand[hot] = rs.randint(hot_lo, n, size=int(hot.sum())).astype(np.uint32)
    cand[rs.rand(Q, R) < 0.01] = 0xFFFFFFFF                  # skipped slots (nvdb_ivf_eval.cpp:513-516)
    cand[5, 7] = n + 11                                      # out of range -> skipped (cuda_refine.cu:437)
    cand[6, :] = 0xFFFFFFFF
    cand[7, 3:] = 0xFFFFFFFF                                 # fewer valid candidates than K
    cand[8, 100:140] = n - 1                                 # the very last row, forty times
    return cand


def test_refine_config5_full_size(oracle):
    """BASELINE configs[4]: exact-L2 refine, fp16 base N = 2.9M x 768 (4.45 GB), Q = 10 000, R = 1024, K = 10.
    Rows >= 2 796 203 start beyond 4 GiB: their gather needs the 64-bit row offset (kernels_refine.h).  More than half
    of every candidate list comes from rows >= 2.8M.  Checked against the oracle's restatement of the reference
    kernel's fp32 order (cuda_refine.cu:326-382, :437; parity unpinned -- DESIGN.md section 6) on the rows copied
    back from HBM, ids AND distance bits, for 96 queries (first, last and a middle block), for both refine kernels;
    the two kernels must agree on all 10 000 queries; and against the CPU refine's double accumulation
    (nvdb_ivf_eval.cpp:232-240, 278-307) within 1e-5 relative."""
    n, d, Q, R, K = 2_900_000, 768, 10_000, 1024, 10
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(SEED, n, d, nvdb_amd.DT_F16)
    base, _ = ctx.download_rows(0, n)                          # 4.45 GB to the host: the oracle reads the same bits
    queries = nvdb_amd.synth_rows_f32(SEED + 2, 0, Q, d)
    queries[::5] = oracle.f16_to_f32(base[(np.arange(0, Q, 5, dtype=np.int64) * 7919 + 2_800_000) % n])   # self-matches: dist 0
    queries[8] = oracle.f16_to_f32(base[n - 1:n])[0]
    rs = np.random.RandomState(55)
    cand = _config5_candidates(rs, n, Q, R, 2_800_000)
    cand[::5, 17] = ((np.arange(0, Q, 5, dtype=np.int64) * 7919 + 2_800_000) % n).astype(np.uint32)
    assert (cand[cand != 0xFFFFFFFF].astype(np.int64) * d * 2 > 2**32).mean() > 0.5
    sub = np.r_[0:32, 5000:5032, Q - 32:Q]
    oid, odist = oracle.refine(base, po.DT_F16, queries[sub], cand[sub], K, mode=0)
    cid, cdist = oracle.refine(base, po.DT_F16, queries[sub], cand[sub], K, mode=1)
    results = {}
    for v2 in (2, 1, 0):
        ctx.set_option("refine_v2", v2)
        ids, dist, t = ctx.refine_l2_topk(queries, cand, K, want_timing=True)
        results[v2] = (ids, dist)
        assert np.array_equal(ids[sub], oid), f"refine_v2={v2}: ids differ from the restated kernel order"
        assert np.array_equal(dist[sub].view(np.uint32), odist.view(np.uint32)), f"refine_v2={v2}: distance bits differ"
        valid = ids[sub] != 0xFFFFFFFF
        assert np.allclose(dist[sub][valid], cdist[valid], rtol=1e-5, atol=1e-7)
        assert (ids[6] == 0xFFFFFFFF).all() and (dist[6] == np.float32(1e30)).all()
        assert (ids[7, 3:] == 0xFFFFFFFF).all() and (ids[7, :3] != 0xFFFFFFFF).all()
        assert (ids[8] == n - 1).all() and (dist[8] == 0.0).all()      # forty copies of the last row, which is query 8 itself
        assert (dist[::5, 0] == 0.0).all() and np.array_equal(ids[::5, 0], cand[::5, 17])      # the self-matches win
        assert t.kernel_ms > 0 and t.R == R and t.K == K
    ctx.set_option("refine_v2", 2)
    for a in (0, 1):
        assert np.array_equal(results[a][0], results[2][0]) and np.array_equal(results[a][1].view(np.uint32), results[2][1].view(np.uint32)), a
    ctx.close()


def test_refine_f32_rows_beyond_4GiB(oracle):
    """fp32 base (cuda_refine.cu:383-392, which the reference declares but never launches, :1055-1085): 1.5M x 768 x 4 B =
    4.6 GB, candidates concentrated in the rows whose byte offset exceeds 4 GiB."""
    n, d, Q, R, K = 1_500_000, 768, 256, 512, 10
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(SEED + 3, n, d, nvdb_amd.DT_F32)
    hot_lo = 1_400_000
    tail, _ = ctx.download_rows(hot_lo, n - hot_lo)
    queries = nvdb_amd.synth_rows_f32(SEED + 4, 0, Q, d)
    rs = np.random.RandomState(56)
    cand = rs.randint(hot_lo, n, size=(Q, R)).astype(np.uint32)
    cand[rs.rand(Q, R) < 0.01] = 0xFFFFFFFF
    local = np.where(cand == 0xFFFFFFFF, cand, cand - np.uint32(hot_lo))
    oid, odist = oracle.refine(tail, po.DT_F32, queries, local, K, mode=0)
    oid = np.where(oid == 0xFFFFFFFF, oid, oid + np.uint32(hot_lo))
    for v2 in (2, 1, 0):
        ctx.set_option("refine_v2", v2)
        ids, dist = ctx.refine_l2_topk(queries, cand, K)
        assert np.array_equal(ids, oid) and np.array_equal(dist.view(np.uint32), odist.view(np.uint32)), v2
    ctx.close()


# ----------------------------------------------------------------------------- device-buffer entry points (torch as plumbing)
@pytest.mark.parametrize("tag,nq", [("f16", 1500), ("i8", 2000), ("f16", 1025)])
def test_device_api_with_more_than_1024_queries_in_one_call(oracle, tag, nq):
    """nvdb_hip_search_batch_dev takes up to 2048 queries per call (the host API splits at 1024): more query tiles
    than the XCD-aware grid / sibling rendezvous cover, so the kernels fall back to the plain block mapping."""
    import torch
    n, d, k = 120000 + 17, 768, 10
    dt = nvdb_amd.DT_F16 if tag == "f16" else nvdb_amd.DT_I8
    c = nvdb_amd.HipContext(0)
    c.generate_corpus(SEED + 95, n, d, dt)
    base, scales = nvdb_amd.synth_corpus(SEED + 95, 0, n, d, dt)
    queries = nvdb_amd.synth_rows_f32(SEED + 96, 0, nq, d)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        t_q = torch.from_numpy(queries).to(dev)
        t_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
        t_sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
        c.set_option("path", 2)
        c.search_batch_dev(t_q.data_ptr(), nq, k, t_ids.data_ptr(), t_sc.data_ptr(), st.cuda_stream)
    st.synchronize()
    s = c.search_check()
    assert s["path"] == 2 and s["bound_violations"] == 0 and s["overflow_queries"] == 0, s
    ids, sc = t_ids.cpu().numpy().astype(np.uint64), t_sc.cpu().numpy()
    sub = np.r_[0:40, nq - 40:nq]                              # the oracle on a subset keeps the test short
    oid, osc = oracle.flat_topk(base, po.DT_F16 if tag == "f16" else po.DT_I8, queries[sub], k, scales)
    assert np.array_equal(ids[sub], oid) and np.array_equal(sc[sub].view(np.uint32), osc.view(np.uint32))
    c.set_option("path", 1)                                    # and the whole batch against the exact kernel
    hi_, hs_ = c.search_batch(queries, k)
    assert np.array_equal(ids, hi_) and np.array_equal(sc.view(np.uint32), hs_.view(np.uint32))
    c.close()


@pytest.mark.parametrize("tag", ["f16", "i8", "f32"])
def test_adopted_corpus_and_device_buffers(oracle, tag):
    """nvdb_hip_adopt_corpus (corpus already in HBM, not owned, NOT padded -> ragged tail on the exact kernel) and
    nvdb_hip_search_batch_dev / nvdb_hip_search_check (queries and results in HBM, caller's stream)."""
    import torch
    n, d, nq, k = 50000 + 21, 768, 96, 10
    dt = {"f16": nvdb_amd.DT_F16, "i8": nvdb_amd.DT_I8, "f32": nvdb_amd.DT_F32}[tag]
    base, scales = nvdb_amd.synth_corpus(SEED + 70, 0, n, d, dt)
    queries = nvdb_amd.synth_rows_f32(SEED + 71, 0, nq, d)
    dev = torch.device("cuda", 0)
    view = base.view(np.int16) if tag == "f16" else base
    t_rows = torch.from_numpy(view).to(dev)
    t_scales = torch.from_numpy(scales).to(dev) if scales is not None else None
    t_q = torch.from_numpy(queries).to(dev)
    t_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    t_sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
    ctx = nvdb_amd.HipContext(0)
    ctx.adopt_corpus(t_rows.data_ptr(), n, d, dt, t_scales.data_ptr() if t_scales is not None else None, row_base=1000)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx.search_batch_dev(t_q.data_ptr(), nq, k, t_ids.data_ptr(), t_sc.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    st = ctx.search_check()
    assert st["path"] == 2 and st["bound_violations"] == 0 and st["overflow_queries"] == 0, st
    ids = t_ids.cpu().numpy().astype(np.uint64) - np.uint64(1000)
    sc = t_sc.cpu().numpy()
    _check_against_oracle(oracle, base, {"f16": po.DT_F16, "i8": po.DT_I8, "f32": po.DT_F32}[tag], scales, queries, ids, sc, k,
                          f"adopt/{tag}")
    # the last rows (ragged tail, < 32) must be reachable: query = one of them
    qtail = (oracle.f16_to_f32(base[n - 3]) if tag == "f16" else
             (base[n - 3].astype(np.float32) * scales[n - 3] if tag == "i8" else base[n - 3]))[None, :].astype(np.float32)
    t_q1 = torch.from_numpy(qtail).to(dev)
    ctx.search_batch_dev(t_q1.data_ptr(), 1, k, t_ids.data_ptr(), t_sc.data_ptr(), None)
    torch.cuda.synchronize()
    ctx.search_check()
    assert int(t_ids[0, 0].item()) == 1000 + n - 3
    ctx.close()
    # refine with device buffers
    if tag != "i8":
        ctx = nvdb_amd.HipContext(0)
        ctx.adopt_corpus(t_rows.data_ptr(), n, d, dt, None)
        rs = np.random.RandomState(2)
        cand = rs.randint(0, n, size=(nq, 200)).astype(np.uint32)
        t_c = torch.from_numpy(cand.view(np.int32)).to(dev)
        t_oi = torch.empty((nq, k), dtype=torch.int32, device=dev)
        t_od = torch.empty((nq, k), dtype=torch.float32, device=dev)
        ctx.refine_l2_topk_dev(t_q.data_ptr(), t_c.data_ptr(), nq, 200, k, t_oi.data_ptr(), t_od.data_ptr(), None)
        torch.cuda.synchronize()
        oid, od = oracle.refine(base, po.DT_F16 if tag == "f16" else po.DT_F32, queries, cand, k, mode=0)
        assert np.array_equal(t_oi.cpu().numpy().view(np.uint32), oid) and np.array_equal(t_od.cpu().numpy().view(np.uint32), od.view(np.uint32))
        ctx.close()


def test_host_api_leaves_the_device_api_sticky_flags_alone(oracle):
    """Sticky self-check words (misc[12..14]) belong to the device API: a host-API search on the same context -- clean or
    one that trips and recovers by itself -- must neither erase what an earlier, unchecked device-API search left, nor
    be failed by it (the nq > 1024 host path used to redo every sub-batch on the exact path)."""
    import torch
    n, d, k = 40000, 768, 10
    base32 = nvdb_amd.synth_rows_f32(SEED + 11, 0, n, d)
    q = nvdb_amd.synth_rows_f32(SEED + 12, 0, 1200, d)
    base = oracle.f32_to_f16(base32[np.argsort(base32 @ q[0])])          # ascending similarity to query 0 (adversarial for it alone)
    c = nvdb_amd.HipContext(0)
    c.upload_corpus(base, po.DT_F16)
    dev = torch.device("cuda", 0)
    t_q = torch.from_numpy(q[:16]).to(dev)
    t_ids = torch.empty((16, k), dtype=torch.int64, device=dev)
    t_sc = torch.empty((16, k), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    c.set_option("path", 2); c.set_option("cand_cap", 64); c.set_option("tile_permute", 0)
    c.search_batch_dev(t_q.data_ptr(), 16, k, t_ids.data_ptr(), t_sc.data_ptr(), None)      # overflows: sticky flag set, not checked yet
    torch.cuda.synchronize()
    c.set_option("cand_cap", 0); c.set_option("tile_permute", 1); c.set_option("path", 0)
    ids, sc = c.search_batch(q[100:], k)                                 # host API, 1100 other queries: two sub-batches, both clean
    st = c.stats()
    assert st["path"] == 2 and st["overflow_queries"] == 0, st          # not dragged onto the exact path by the stale flag
    oid, osc = oracle.flat_topk(base, po.DT_F16, q[100:124], k)
    assert np.array_equal(ids[:24], oid) and np.array_equal(sc[:24].view(np.uint32), osc.view(np.uint32))
    c.set_option("path", 2); c.set_option("cand_cap", 64); c.set_option("tile_permute", 0)
    c.search_batch(q[:16], k)                                            # a host search that trips and recovers by itself
    c.set_option("cand_cap", 0); c.set_option("tile_permute", 1); c.set_option("path", 0)
    with pytest.raises(nvdb_amd.NvdbError) as e:                        # the device-API search's flag is still there
        c.search_check()
    assert e.value.status == 5
    assert c.search_check()["sticky_overflow"] == 0                     # ... and the check cleared it
    c.close()


def test_wide_k_without_the_mfma_bootstrap_takes_the_any_k_path(oracle):
    """64 < k <= 1024 rides the filter path only with the MFMA bootstrap; when that is unavailable (boot_tiles larger
    than the corpus here) the search must go to the any-k path, not to the exact bootstrap chunk whose wavefront lists
    hold 64 entries."""
    n, d, nq, k = 60000, 768, 40, 100
    c = nvdb_amd.HipContext(0)
    c.generate_corpus(SEED + 97, n, d, nvdb_amd.DT_F16)
    queries = nvdb_amd.synth_rows_f32(SEED + 98, 0, nq, d)
    c.set_option("path", 2)
    c.set_option("boot_tiles", 4096)
    c.set_option("chunk0_rows", 4096)
    ids, sc = c.search_batch(queries, k)
    assert c.stats()["path"] == 3
    base, _ = nvdb_amd.synth_corpus(SEED + 97, 0, n, d, nvdb_amd.DT_F16)
    oid, osc = oracle.flat_topk(base, po.DT_F16, queries[:8], k)
    assert np.array_equal(ids[:8], oid) and np.array_equal(sc[:8].view(np.uint32), osc.view(np.uint32))
    c.close()


def test_int8_corpus_with_a_huge_row_scale_takes_the_unbiased_build(oracle):
    """A row scale beyond ~1e30 would make the biased-accumulator test compute inf - inf; such corpora are routed to the
    in-loop build like corpora with negative scales."""
    n, d, nq, k = 600_000, 768, 200, 10
    c = nvdb_amd.HipContext(0)
    base, scales = nvdb_amd.synth_corpus(SEED + 99, 0, n, d, nvdb_amd.DT_I8)
    scales = scales.copy()
    scales[12345] = np.float32(3e31)
    c.upload_corpus(base, po.DT_I8, scales)
    queries = nvdb_amd.synth_rows_f32(SEED + 98, 0, nq, d)
    c.set_option("path", 2)
    ids, sc = c.search_batch(queries, k)
    st = c.stats()
    c.set_option("path", 1)
    ei, es = c.search_batch(queries, k)
    c.close()
    assert np.array_equal(ids, ei) and np.array_equal(sc.view(np.uint32), es.view(np.uint32))
    oid, osc = oracle.flat_topk(base, po.DT_I8, queries[:4], k, scales)
    assert np.array_equal(ids[:4], oid) and np.array_equal(sc[:4].view(np.uint32), osc.view(np.uint32))


@pytest.mark.parametrize("tag", ["f16", "f32", "i8"])
@pytest.mark.parametrize("d,nq", [(768, 64), (768, 9), (768, 33), (384, 100), (128, 17), (512, 48), (256, 130), (128, 64), (768, 200), (640, 70), (640, 10)])
def test_exact_scores_on_the_fp32_matrix_cores(oracle, tag, d, nq):
    """Path 1 with more than 8 queries on dims that are whole MFMA K-steps runs on v_mfma_f32_16x16x4_f32 (eight accumulator
    tiles = the reference's eight stride-8 fma chains, kernels_exact_mfma.h): ids and score BITS must equal the VALU kernels'
    (option exact_mfma = 0) and the oracle's, including rows / queries with zeros, huge and tiny magnitudes (subnormal
    products and sums) and a row count that is not a multiple of the 16-row tile."""
    n, k = 20_000 + 7, 10
    rs = np.random.RandomState(d + nq)
    base32 = nvdb_amd.synth_rows_f32(SEED + 160, 0, n, d)
    base32[11] = 0.0
    base32[12] *= np.float32(1e-30); base32[13] *= np.float32(3e18); base32[14, ::2] = 0.0
    base32[15] = base32[16]                                              # an exact tie
    base32[500:600] *= (np.float32(10.0) ** rs.randint(-20, 8, size=(100, 1))).astype(np.float32)
    base, dt, scales = _as_dtype(oracle, base32, tag)
    queries = nvdb_amd.synth_rows_f32(SEED + 161, 0, nq, d)
    queries[1] *= np.float32(1e-12); queries[2] *= np.float32(1e15); queries[3, : d // 2] = 0.0; queries[4] = -queries[4]
    queries[5] = base32[15]
    c = nvdb_amd.HipContext(0)
    c.upload_corpus(base, dt, scales)
    c.set_option("path", 1)
    # kernel builds: fp32 LDS image (fp16 / int8 rows, full groups of 64 queries; the default), raw tiles staged through LDS,
    # register-direct loads only, and the VALU kernels
    MODES = {"img": dict(exact_mfma=1, exact_img=1, exact_lds=1), "lds": dict(exact_mfma=1, exact_img=0, exact_lds=2),
             "reg": dict(exact_mfma=1, exact_img=0, exact_lds=0), "valu": dict(exact_mfma=0, exact_img=0, exact_lds=0)}

    def run(qs, kk):
        out = {}
        for name, opts in MODES.items():
            for key, val in opts.items():
                c.set_option(key, val)
            out[name] = c.search_batch(qs, kk)
        return out

    res = run(queries, k)
    assert c.stats()["path"] == 1
    for name in ("img", "lds", "reg"):
        assert np.array_equal(res[name][0], res["valu"][0]) and np.array_equal(res[name][1].view(np.uint32), res["valu"][1].view(np.uint32)), name
    sub = np.r_[0:8, nq - 1]
    _check_against_oracle(oracle, base, dt, scales, queries[sub], res["img"][0][sub], res["img"][1][sub], k, f"exact-mfma/{tag}/d{d}")
    # SUBNORMAL scores as the top-k (the tiny rows above never reach a top-10): every query scaled by 1e-37, so every query element (~4e-39),
    # product (~1e-40) and nearly every sum (~4e-39) is an fp32 subnormal -- a matrix core that flushed subnormal inputs or outputs
    # would return zeros here; the fma chain and the oracle do not
    tiny_q = (queries * np.float32(1e-37)).astype(np.float32)
    tiny_q[2] = queries[0] * np.float32(1e-37)                           # (query 2 was scaled up above)
    rt = run(tiny_q, k)
    for name in ("img", "lds", "reg"):
        assert np.array_equal(rt[name][0], rt["valu"][0]) and np.array_equal(rt[name][1].view(np.uint32), rt["valu"][1].view(np.uint32)), ("subnormal", name)
    _check_against_oracle(oracle, base, dt, scales, tiny_q[sub], rt["img"][0][sub], rt["img"][1][sub], k, f"exact-mfma subnormal/{tag}/d{d}")
    # the any-k path's score matrix comes from the same tiles (k = 100 on 20K rows is off the filter path)
    res = run(queries, 100)
    assert c.stats()["path"] == 3
    for name in ("img", "lds", "reg"):
        assert np.array_equal(res[name][0], res["valu"][0]) and np.array_equal(res[name][1].view(np.uint32), res["valu"][1].view(np.uint32)), name
    rt = run(tiny_q, n)                                                  # k = n: EVERY row's score, in order -- most of them subnormal
    a0 = np.abs(rt["valu"][1][0])
    assert np.count_nonzero((a0 > 0) & (a0 < np.float32(1.17e-38))) > n // 2, "the tiny queries must produce subnormal, non-zero scores"
    for name in ("img", "lds", "reg"):
        assert np.array_equal(rt[name][0], rt["valu"][0]) and np.array_equal(rt[name][1].view(np.uint32), rt["valu"][1].view(np.uint32)), ("subnormal any-k", name)
    c.close()


@pytest.mark.parametrize("tag", ["f16", "i8", "f32"])
def test_exact_path_prescan_and_grid_options_leave_the_results_unchanged(oracle, tag):
    """Path 1 on a corpus of >= 2^20 rows scans the first 1/64 of the rows alone, turns it into the exact k-th best score per query
    (a select with slack 0) and starts the scan of the rest with that bar (option exact_prescan); the scan grid is one workgroup per
    CU in all (exact_wgs).  Neither may change a bit: 70 queries (one full group of 64 on the image / LDS build + 6 on the
    register-direct build), duplicate rows across the prescan boundary (ties at the bar), k = 1 and k = 64."""
    n, d, nq = (1 << 20) + 12_345, 128, 70
    dt = {"f16": nvdb_amd.DT_F16, "i8": nvdb_amd.DT_I8, "f32": nvdb_amd.DT_F32}[tag]
    base, scales = nvdb_amd.synth_corpus(SEED + 190, 0, n, d, dt)
    head = max(1 << 15, (n >> 6) & ~255)
    base[head + 5] = base[head - 7]; base[n - 1] = base[3]                    # exact score ties on both sides of the prescan boundary
    if scales is not None:
        scales[head + 5] = scales[head - 7]; scales[n - 1] = scales[3]
    queries = nvdb_amd.synth_rows_f32(SEED + 191, 0, nq, d)
    as_f32 = (lambda r, i: oracle.f16_to_f32(r) if tag == "f16" else (r.astype(np.float32) * (scales[i] if tag == "i8" else 1.0)))
    queries[0] = as_f32(base[head - 7], head - 7).astype(np.float32)         # its two best rows are the tie pair
    queries[1] = as_f32(base[3], 3).astype(np.float32)
    c = nvdb_amd.HipContext(0)
    c.upload_corpus(base, dt, scales)
    c.set_option("path", 1)
    po_dt = {"f16": po.DT_F16, "i8": po.DT_I8, "f32": po.DT_F32}[tag]
    for k in (10, 1, 64):
        res = {}
        for pre, wgs in ((1, 1), (0, 1), (1, 2), (0, 3)):
            c.set_option("exact_prescan", pre); c.set_option("exact_wgs", wgs)
            res[(pre, wgs)] = c.search_batch(queries, k)
            assert c.stats()["path"] == 1 and c.stats()["chunks"] == (2 if pre else 1)
        ref = res[(0, 1)]
        for key, (i_, s_) in res.items():
            assert np.array_equal(i_, ref[0]) and np.array_equal(s_.view(np.uint32), ref[1].view(np.uint32)), (tag, k, key)
        sub = np.r_[0:3, 63:66, nq - 1]
        _check_against_oracle(oracle, base, po_dt, scales, queries[sub], ref[0][sub], ref[1][sub], k, f"exact prescan/{tag}/k{k}")
    assert set(res[(1, 1)][0][0][:2].tolist()) == {head - 7, head + 5} if tag != "i8" else True
    c.close()


def test_exact_mfma_lds_kernel_on_an_adopted_unpadded_corpus(oracle):
    """The LDS-staged exact kernel brings tiles in with direct-to-LDS loads; on an ADOPTED corpus (the caller's buffer, no zero
    padding behind the last row) a ragged last tile must be clamped into the row range, for rows and int8 scales alike."""
    import torch
    n, d, nq, k = 4096 + 5, 768, 128, 10                         # 5 rows in the last 16-row tile
    dev = torch.device("cuda", 0)
    for tag, dt in (("f16", nvdb_amd.DT_F16), ("i8", nvdb_amd.DT_I8), ("f32", nvdb_amd.DT_F32)):
        base, scales = nvdb_amd.synth_corpus(SEED + 180, 0, n, d, dt)
        queries = nvdb_amd.synth_rows_f32(SEED + 181, 0, nq, d)
        queries[7] = (oracle.f16_to_f32(base[n - 2]) if tag == "f16" else (base[n - 2].astype(np.float32) * (scales[n - 2] if tag == "i8" else 1.0))).astype(np.float32)
        view = base.view(np.int16) if tag == "f16" else base
        t_rows = torch.from_numpy(view).to(dev).clone()           # exactly n rows: nothing behind them belongs to us
        t_scales = torch.from_numpy(scales).to(dev).clone() if scales is not None else None
        c = nvdb_amd.HipContext(0)
        c.adopt_corpus(t_rows.data_ptr(), n, d, dt, t_scales.data_ptr() if t_scales is not None else None)
        c.set_option("path", 1)
        gi, gs = c.search_batch(queries, k)                        # defaults: the fp32-image build for fp16 / int8 rows, raw LDS stages for fp32 rows
        c.set_option("exact_lds", 2)                               # the raw LDS-staged build for every dtype
        ids, sc = c.search_batch(queries, k)
        c.set_option("exact_mfma", 0)
        ei, es = c.search_batch(queries, k)
        c.close()
        assert np.array_equal(ids, ei) and np.array_equal(sc.view(np.uint32), es.view(np.uint32)), tag
        assert np.array_equal(gi, ei) and np.array_equal(gs.view(np.uint32), es.view(np.uint32)), (tag, "default build")
        assert ids[7, 0] == n - 2
        _check_against_oracle(oracle, base, {"f16": po.DT_F16, "i8": po.DT_I8, "f32": po.DT_F32}[tag], scales, queries[5:9], ids[5:9], sc[5:9], k, f"exact-lds adopt/{tag}")


def test_exact_mfma_pruned_by_thresholds_and_small_row_ranges(oracle):
    """The MFMA scan as the filter path's helper: ragged tails / bootstrap chunks call it with per-query thresholds and
    arbitrary row ranges; k = 64 fills the wavefront-sized lists; 3 query blocks leave one wave of the workgroup idle."""
    n, d, nq, k = 9_000 + 5, 768, 40, 64
    c = nvdb_amd.HipContext(0)
    c.generate_corpus(SEED + 170, n, d, nvdb_amd.DT_F16)
    base, _ = nvdb_amd.synth_corpus(SEED + 170, 0, n, d, nvdb_amd.DT_F16)
    queries = nvdb_amd.synth_rows_f32(SEED + 171, 0, nq, d)
    c.set_option("path", 1)
    ids, sc = c.search_batch(queries, k)
    c.set_option("exact_mfma", 0)
    ei, es = c.search_batch(queries, k)
    c.close()
    assert np.array_equal(ids, ei) and np.array_equal(sc.view(np.uint32), es.view(np.uint32))
    _check_against_oracle(oracle, base, po.DT_F16, None, queries[:6], ids[:6], sc[:6], k, "exact-mfma k=64")


def test_product_library_rejects_developer_variants(ctx):
    for key, val in (("mfma16", 0), ("i8_wide", 0), ("i8_pipe", 0), ("i8_waves8", 1), ("i8_mfma16", 0), ("i8_small8", 0)):
        with pytest.raises(nvdb_amd.NvdbError) as e:
            ctx.set_option(key, val)
        assert e.value.status == 3 and "developer-build" in str(e.value)
        ctx.set_option(key, 1 - val)                         # the default value is accepted


def test_randomised_cross_check():
    """tools_dev/fuzz.py: 40 random (n, dim, dtype, nq, k, options) cases; the automatic path must equal the exact path
    (ids and score bits), and both the oracle on a few queries per case."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools_dev", "fuzz.py"), "7", "40"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_randomised_refine_check():
    """tools_dev/fuzz_refine.py: 30 random (n, dim, dtype, Q, R, K) refine cases with invalid / out-of-range / duplicate
    candidate ids on both refine kernels; ids and distance bits must equal the oracle's restated kernel order."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools_dev", "fuzz_refine.py"), "5", "30"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
