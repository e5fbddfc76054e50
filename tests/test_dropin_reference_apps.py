"""Drop-in proof for SURVEY section 8(b): the reference's UNMODIFIED app and tool mains compile against the product's
headers (nano-vectordb_amd/host/include) and link against libnvdb_host.a + libnvdb_hip.so -- sources compiled where
they lie under /root/reference by `make -C oracle dropin` (outputs in oracle/_ref/dropin, git-ignored), nothing copied.
The binaries are then run on the golden cases and compared with tests/golden/ (bytes of the converters' files, the
.gtbin, nvdb_search's stdout) and, for what the goldens do not hold (nvdb_bench's sink and line keys, nvdb_make_query /
nvdb_slice / nvdb_dump / nvdb_sanity output), with the real reference binaries built from the same mains against the
reference's own library (oracle/_ref/bin).

Build container only: skipped where /root/reference does not exist (the GPU box)."""
import os
import re
import subprocess

import numpy as np
import pytest

import pyoracle as po
from golden_inputs import make_case_inputs, sha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DROP = os.path.join(ROOT, "oracle", "_ref", "dropin")
MAINS = {"nvdb_bench": "apps", "nvdb_search": "apps", "nvdb_gt_build": "apps", "nvdb_quantize_i8": "apps", "nvdb_dump": "apps",
         "nvdb_sanity": "apps", "nvdb_convert_f16": "tools", "nvdb_make_query": "tools", "nvdb_slice": "tools"}

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "apps")), reason="/root/reference is not present here")


@pytest.fixture(scope="module")
def dropin():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "nano-vectordb_amd"), "-j8"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all", "dropin", "-j8"], stdout=subprocess.DEVNULL)
    for name in MAINS:
        assert os.path.exists(os.path.join(DROP, name)), name
    return DROP


def drun(tool, *args, env=None, where=DROP):
    e = dict(os.environ, OMP_NUM_THREADS="2")
    e.update(env or {})
    return subprocess.run([os.path.join(where, tool), *map(str, args)], check=True, capture_output=True, text=True, env=e).stdout


def rrun(tool, *args, env=None):
    return drun(tool, *args, env=env, where=po.Reference().bin)


@pytest.fixture(scope="module")
def files(tmp_path_factory, dropin):
    d = tmp_path_factory.mktemp("dropin")
    base32, queries = make_case_inputs("main768")
    p = {k: str(d / v) for k, v in dict(b32="b32.vecbin", q="q.raw12", b16="b16.vecbin", b8="b8.vecbin", gt="gt.gtbin").items()}
    po.write_vecbin(p["b32"], base32, po.DT_F32)
    po.write_raw12(p["q"], queries)
    drun("nvdb_convert_f16", p["b32"], p["b16"])
    drun("nvdb_quantize_i8", p["b32"], p["b8"])
    p["dir"] = str(d)
    return p


def test_the_mains_compiled_are_the_reference_files_themselves(dropin):
    """The recipe names the sources under /root/reference directly: no copy of them exists in the repo."""
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    assert "$(REF)/apps/%.cpp" in mk and "$(REF)/tools/%.cpp" in mk and "-I$(PKG)/host/include" in mk
    dr = mk[mk.index("DROPFLAGS"):]
    assert "$(REF)/include" not in dr, "the drop-in build must not see the reference's headers"
    for name, sub in MAINS.items():
        assert not os.path.exists(os.path.join(ROOT, "nano-vectordb_amd", "host", sub, name + ".cpp")) or sub == "apps"
    # every header the reference's mains include from nvdb/ exists in the product's include directory
    want = set()
    for name, sub in MAINS.items():
        want |= set(re.findall(r'#include "nvdb/([a-zA-Z0-9_]+\.h)"', open(os.path.join(REF, sub, name + ".cpp")).read()))
    have = set(os.listdir(os.path.join(ROOT, "nano-vectordb_amd", "host", "include", "nvdb")))
    assert want <= have, sorted(want - have)
    assert {"flat_index_omp.h", "flat_index_async.h", "flat_index_pool.h", "score_dispatch.h"} <= want


def test_converters_write_the_golden_bytes(files, golden):
    b16 = po.read_vecbin(files["b16"])[0]
    b8, _, sc = po.read_vecbin(files["b8"])
    assert bytes(golden["main768_f16_sha"]).hex() == sha(b16)
    assert bytes(golden["main768_i8_sha"]).hex() == sha(b8, sc)


def test_nvdb_search_prints_the_golden_lines(files, golden):
    assert drun("nvdb_search", files["b32"], files["q"], 10) == bytes(golden["main768_search_stdout"]).decode()


@pytest.mark.parametrize("mode", ["st", "omp"])
def test_nvdb_gt_build_writes_the_golden_gtbin(files, golden, mode):
    drun("nvdb_gt_build", files["b16"], files["q"], 10, files["gt"], env={"GT_MODE": mode, "OMP_NUM_THREADS": "3"})
    ids, _ = po.read_gtbin(files["gt"])
    assert np.array_equal(ids, golden["main768_gtbin_f16_ids"])
    assert np.fromfile(files["gt"], dtype=np.uint8)[:64].tobytes() == bytes(golden["main768_gtbin_raw"])


@pytest.mark.parametrize("base", ["b32", "b16", "b8"])
@pytest.mark.parametrize("mode,extra", [("st", []), ("omp", ["2"]), ("async", ["3"]), ("pool", ["3"]),
                                        ("st", ["1", "1", "4", "512", "0"]), ("omp", ["2", "1", "4", "512", "8"]),
                                        ("pool", ["2", "1", "4", "256", "0"])])
def test_nvdb_bench_matches_the_reference_binary(files, base, mode, extra):
    """st / omp / async / pool, per query and batched (the triple loop of apps/nvdb_bench.cpp:47-251 runs through the
    product's score_dispatch.h + topK.h here): same `sink`, same line keys as the real reference binary on the same file."""
    if base == "b8" and len(extra) >= 5 and extra[4] != "0":
        # prefetch_dist > 0 on an int8 base: the reference's own prefetch helper throws inside the OpenMP region
        # (apps/nvdb_bench.cpp:29-43) and the process aborts -- with the product's headers exactly as with its own
        for where in (DROP, po.Reference().bin):
            r = subprocess.run([os.path.join(where, "nvdb_bench"), files[base], files["q"], "10", mode, *extra], capture_output=True)
            assert r.returncode == -6, (where, r.returncode)
        return
    out = drun("nvdb_bench", files[base], files["q"], 10, mode, *extra)
    ref = rrun("nvdb_bench", files[base], files["q"], 10, mode, *extra)
    keys = lambda s: [re.split(r"[=:]", l)[0] for l in s.strip().splitlines()]
    assert keys(out) == keys(ref)
    assert re.search(r"sink=(\S+)", out).group(1) == re.search(r"sink=(\S+)", ref).group(1)
    assert out.splitlines()[1] == ref.splitlines()[1]                       # Base count= ... line
    assert re.search(r"bytes_per_query=\d+", out).group(0) == re.search(r"bytes_per_query=\d+", ref).group(0)


def test_small_tools_match_the_reference_binaries(files):
    d = files["dir"]
    for args in (("8", "42", "random"), ("5", "7", "first")):
        a, b = os.path.join(d, "mq_a.vecbin"), os.path.join(d, "mq_b.vecbin")
        drun("nvdb_make_query", files["b32"], a, *args)
        rrun("nvdb_make_query", files["b32"], b, *args)
        assert open(a, "rb").read() == open(b, "rb").read()
    assert drun("nvdb_make_query", files["b32"], os.path.join(d, "mq_c.vecbin"), "3").count("\n") >= 1


def test_slice_dump_sanity_run_on_every_dtype(files):
    """nvdb_slice / nvdb_dump / nvdb_sanity are not built into oracle/_ref/bin; the reference library's answers for them
    come from its own VectorDataset, so compare with a same-main build against the reference's library made on the spot."""
    d = files["dir"]
    refbin = os.path.join(d, "refbin")
    os.makedirs(refbin, exist_ok=True)
    objs = [os.path.join(ROOT, "oracle", "_ref", "obj", o) for o in os.listdir(os.path.join(ROOT, "oracle", "_ref", "obj"))]
    for name in ("nvdb_slice", "nvdb_dump", "nvdb_sanity"):
        subprocess.check_call(["g++", "-O2", "-std=gnu++17", "-mavx2", "-mfma", "-pthread", "-fopenmp", "-DNVDB_HAS_OPENMP=1", f"-I{REF}/include",
                               "-o", os.path.join(refbin, name), os.path.join(REF, MAINS[name], name + ".cpp"), *objs])
    def both(tool, *args):
        """run the drop-in build and the reference build; same exit status (the reference's tools let some exceptions
        escape, e.g. nvdb_slice on a non-fp32 file) and same stdout"""
        r = [subprocess.run([os.path.join(w, tool), *map(str, args)], capture_output=True, text=True) for w in (DROP, refbin)]
        assert r[0].returncode == r[1].returncode, (tool, args, r[0].returncode, r[1].returncode, r[0].stderr[-200:], r[1].stderr[-200:])
        assert r[0].stdout == r[1].stdout, (tool, args)
        if r[0].returncode != 0:
            assert r[0].stderr.strip().splitlines()[-1:] == r[1].stderr.strip().splitlines()[-1:]
        return r[0].returncode
    for key in ("b32", "b16", "b8"):
        a, b = os.path.join(d, "sl_a.vecbin"), os.path.join(d, "sl_b.vecbin")
        for w, o in ((DROP, a), (refbin, b)):
            subprocess.run([os.path.join(w, "nvdb_slice"), files[key], o, "100"], capture_output=True)
        both("nvdb_slice", files[key], a, 100)
        if os.path.exists(a) or os.path.exists(b):
            assert open(a, "rb").read() == open(b, "rb").read()
            os.remove(a), os.remove(b)
        both("nvdb_dump", files[key], 2, 6)
        both("nvdb_sanity", files[key], 5)
    assert both("nvdb_dump", files["b32"], 2, 6) == 0 and both("nvdb_sanity", files["b32"], 5) == 0
