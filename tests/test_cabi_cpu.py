"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/nvdb_hip.h declares, fails loudly without a GPU, and its host-side helpers (the functions
that define the corpus bits) agree with the oracle."""
import os
import re

import numpy as np
import pytest

import nvdb_amd
import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(nvdb_amd.LIB_PATH):
        g.build()
    return nvdb_amd.load_library()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "nvdb_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nvdb_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in nvdb_hip.h but not exported"
    assert declared == set(nvdb_amd.EXPORTS), declared ^ set(nvdb_amd.EXPORTS)
    assert lib.nvdb_hip_abi_version() == 3


def test_product_library_carries_only_the_drop_in_surface(lib):
    """Developer aids (timing-only ablation launchers, in-kernel clock stamps) live in libnvdb_hip_dev.so behind
    include/nvdb_hip_dev.h: the product library exports none of them, the dev library all of them plus the whole ABI."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", nvdb_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b(nvdb_[a-z0-9_]+)\b", syms))
    assert exported == set(nvdb_amd.EXPORTS), exported ^ set(nvdb_amd.EXPORTS)
    assert not any("debug" in e for e in exported)
    dev = nvdb_amd.load_dev_library()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nvdb_hip_dev.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(nvdb_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(nvdb_amd.DEV_EXPORTS), declared ^ set(nvdb_amd.DEV_EXPORTS)
    for name in list(nvdb_amd.EXPORTS) + list(nvdb_amd.DEV_EXPORTS):
        assert hasattr(dev, name), name


def test_timing_struct_mirrors_reference_layout():
    # nvdb::CudaRefineTiming (reference include/nvdb/cuda_refine.h:7-22): 4 floats, 4 u32, size_t, u32, 6 doubles
    names = [f for f, _ in nvdb_amd.Timing._fields_]
    assert names == ["h2d_ms", "kernel_ms", "d2h_ms", "total_ms", "threads", "nwarps", "K", "R", "shmem_bytes", "dbg_q",
                     "dbg_dist_cycles_avg", "dbg_write_cycles_avg", "dbg_merge_cycles_avg", "dbg_dist_pct",
                     "dbg_write_pct", "dbg_merge_pct"]
    import ctypes
    assert ctypes.sizeof(nvdb_amd.Timing) == 96


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_fails_loudly_without_gpu(lib):
    assert lib.nvdb_hip_device_count() <= 0
    with pytest.raises(nvdb_amd.NvdbError) as e:
        nvdb_amd.HipContext(0)
    assert e.value.status == 2 and "HIP" in str(e.value)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_device_group_fails_loudly_without_gpu(lib):
    """The multi-GPU group has no CPU fallback either, and creating one must not need RCCL to be loadable."""
    with pytest.raises(nvdb_amd.NvdbError) as e:
        nvdb_amd.DeviceGroup([0, 1])
    assert e.value.status == 2 and "HIP" in str(e.value)
    import ctypes
    assert lib.nvdb_hip_group_create(None, 0, ctypes.byref(ctypes.c_void_p())) == 1          # empty device list: INVALID
    assert lib.nvdb_hip_group_size(None) == 0 and lib.nvdb_hip_group_exchange(None, None) == -1


def test_product_library_does_not_link_rccl(lib):
    """RCCL is bound with dlopen when a group is created; single-GPU users of libnvdb_hip.so never load it."""
    import subprocess
    needed = subprocess.run(["objdump", "-p", nvdb_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "librccl" not in needed and "libamdhip64" in needed


def test_host_f16_conversion_matches_oracle(lib, oracle):
    from golden_inputs import make_f16_specials
    sp = make_f16_specials()
    sp = sp[~np.isnan(sp)]
    mine = nvdb_amd.f32_to_f16(sp)
    ref = np.array([oracle.lib.oracle_f32_to_f16(float(v)) for v in sp], dtype=np.uint16)   # F16C semantics
    assert np.array_equal(mine, ref)
    rs = np.random.RandomState(9)
    x = (rs.standard_normal(200000) * np.exp(rs.uniform(-25, 14, 200000))).astype(np.float32)
    assert np.array_equal(nvdb_amd.f32_to_f16(x), oracle.f32_to_f16(x))


def test_host_int8_quantiser_matches_oracle(lib, oracle):
    rs = np.random.RandomState(10)
    rows = (rs.standard_normal((300, 768)) / 27.7).astype(np.float32)
    rows[7] = 0.0
    a, sa = nvdb_amd.quantize_i8(rows)
    b, sb = oracle.quantize_i8(rows)
    assert np.array_equal(a, b) and np.array_equal(sa.view(np.uint32), sb.view(np.uint32))


def test_converters_match_reference_goldens(lib, golden):
    """The product's converters reproduce the reference tools' output files (dim % 8 == 0 cases;
    for dim % 8 != 0 the reference's scalar tail has two defects that are not copied)."""
    from golden_inputs import CASES, make_case_inputs, sha
    for name, spec in CASES.items():
        base32, _ = make_case_inputs(name)
        if spec["d"] % 8 == 0:
            assert bytes(golden[f"{name}_f16_sha"]).hex() == sha(nvdb_amd.f32_to_f16(base32)), name
        b8, sc = nvdb_amd.quantize_i8(base32)
        assert bytes(golden[f"{name}_i8_sha"]).hex() == sha(b8, sc), name


def test_synth_generator_is_deterministic_and_normalised(lib):
    a = nvdb_amd.synth_rows_f32(20240613, 1000, 64, 768)
    b = nvdb_amd.synth_rows_f32(20240613, 1032, 32, 768)
    assert np.array_equal(a[32:], b)                                  # keyed by absolute row id
    assert np.allclose(np.linalg.norm(a.astype(np.float64), axis=1), 1.0, atol=1e-6)
    assert abs(a.mean()) < 1e-3 and 0.9 < a.std() * np.sqrt(768) < 1.1
    c = nvdb_amd.synth_rows_f32(20240614, 1000, 64, 768)
    assert not np.array_equal(a, c)
    big = nvdb_amd.synth_rows_f32(7, (1 << 33) + 5, 2, 64)            # 64-bit row ids
    assert np.isfinite(big).all()


def test_merge_topk_host_canonical_order(lib):
    ids = np.array([[[5, 9, 11]], [[2, 7, 30]]], dtype=np.uint64)                  # 2 shards, 1 query, k=3
    sc = np.array([[[0.9, 0.5, 0.5]], [[0.9, 0.5, 0.1]]], dtype=np.float32)
    oi, os_ = nvdb_amd.merge_topk_host(ids, sc)
    assert oi.tolist() == [[2, 5, 7]] and os_.tolist() == [[np.float32(0.9), np.float32(0.9), np.float32(0.5)]]


def test_tile_permutation_is_a_bijection():
    """The filter kernels stream logical tile g from physical tile nvdb_permuted_tile(g, T); every tile must be visited
    exactly once for any corpus size (odd multiplier modulo the next power of two, cycle-walked into range)."""
    lib = nvdb_amd.load_dev_library()                          # a developer aid: only the dev build exports it
    for T in (1, 2, 63, 64, 65, 100, 1023, 1024, 1025, 1876, 3751, 4097, 65536, 78125, 312500):
        img = np.fromiter((lib.nvdb_permuted_tile(g, T) for g in range(T)), dtype=np.uint32, count=T)
        assert img.max() == T - 1 and len(np.unique(img)) == T, T
        if T >= 1024:                                          # and it spreads: consecutive logical tiles are far apart
            assert np.median(np.abs(np.diff(img.astype(np.int64)))) > T // 8, T


def test_makefile_targets_that_link_the_hip_library_depend_on_it():
    """A parallel build of a fresh tree (what the driver's build check does after a clone: the .so files are not in
    the history) must not link -lnvdb_hip before lib/libnvdb_hip.so exists: every rule whose recipe links it lists it."""
    import re
    mk = open(os.path.join(ROOT, "nano-vectordb_amd", "Makefile")).read()
    rules = re.findall(r"^([^\s#:][^:\n]*):([^\n]*)\n((?:\t[^\n]*\n)+)", mk, flags=re.M)
    linking = [(t.strip(), deps) for t, deps, recipe in rules if "-lnvdb_hip" in recipe]
    assert len(linking) >= 5, linking
    for target, deps in linking:
        assert "lib/libnvdb_hip.so" in deps.split(), target
