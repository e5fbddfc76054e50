"""Host C++ layer (nano-vectordb_amd/host): the reference's nvdb:: surface re-implemented over the C ABI,
and the CLI tools.  CPU tests pin the CPU modes against the reference goldens; GPU tests (marked) run the
gpu modes and the refine harness."""
import os
import re
import subprocess
import struct

import numpy as np
import pytest

import pyoracle as po
from golden_inputs import CASES, make_case_inputs, sha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "nano-vectordb_amd", "bin")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(os.path.join(BIN, "nvdb_bench")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "nano-vectordb_amd"), "-j8"])
    return BIN


def run(tool, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([os.path.join(BIN, tool), *map(str, args)], check=True, capture_output=True, text=True, env=e).stdout


@pytest.fixture(scope="module")
def files(tmp_path_factory, built):
    d = tmp_path_factory.mktemp("vecbin")
    base32, queries = make_case_inputs("main768")
    p = dict(b32=str(d / "b32.vecbin"), q=str(d / "q.raw12"), b16=str(d / "b16.vecbin"), b8=str(d / "b8.vecbin"), gt=str(d / "gt.gtbin"))
    po.write_vecbin(p["b32"], base32, po.DT_F32)
    po.write_raw12(p["q"], queries)
    run("nvdb_convert_f16", p["b32"], p["b16"])
    run("nvdb_quantize_i8", p["b32"], p["b8"])
    return p


def test_converter_tools_write_the_reference_bytes(files, golden):
    b16 = po.read_vecbin(files["b16"])[0]
    b8, _, sc = po.read_vecbin(files["b8"])
    assert bytes(golden["main768_f16_sha"]).hex() == sha(b16)
    assert bytes(golden["main768_i8_sha"]).hex() == sha(b8, sc)


def test_nvdb_search_cpu_prints_the_reference_lines(files, golden):
    assert run("nvdb_search", files["b32"], files["q"], 10) == bytes(golden["main768_search_stdout"]).decode()


def test_nvdb_gt_build_cpu_modes_write_the_reference_gtbin(files, golden):
    for mode in ("st", "omp", "async", "pool"):
        run("nvdb_gt_build", files["b16"], files["q"], 10, files["gt"], env={"GT_MODE": mode, "OMP_NUM_THREADS": "3"})
        ids, meta = po.read_gtbin(files["gt"])
        assert np.array_equal(ids, golden["main768_gtbin_f16_ids"])
        assert np.fromfile(files["gt"], dtype=np.uint8)[:64].tobytes() == bytes(golden["main768_gtbin_raw"])


@pytest.mark.parametrize("mode,extra", [("st", []), ("omp", ["2"]), ("omp", ["2", "1", "4", "512", "0"]),
                                        ("async", ["3"]), ("pool", ["3"]), ("pool", ["2", "1", "4", "512", "0"])])
def test_nvdb_bench_cpu_output_lines(files, mode, extra):
    out = run("nvdb_bench", files["b16"], files["q"], 10, mode, *extra, env={"OMP_NUM_THREADS": "2"})
    lines = out.strip().splitlines()
    assert lines[0].startswith(f"mode={mode} threads=")
    assert lines[1] == "Base count=3000 dim=768 | Query count=8 | k=10 | warmup=" + (extra[1] if len(extra) > 1 else "5")
    keys = [re.split(r"[=:]", l)[0] for l in lines[2:]]
    if len(extra) > 2:   # batch_q = 4
        assert keys == ["batch_samples", "Total", "Avg_query", "Avg_batch", "batch_p50", "batch_p95", "batch_p99", "sink",
                        "bytes_per_query", "payload_equiv_bandwidth_GBps", "(note) payload_equiv_bandwidth_GBps may exceed DRAM peak due to cache reuse",
                        "batch_q"]
    else:
        assert keys == ["Total", "Avg_query", "p50", "p95", "p99", "sink", "bytes_per_query", "payload_equiv_bandwidth_GBps", "batch_q"]
    assert "bytes_per_query=4608000" in out                     # 3000 * 768 * 2
    if po.Reference.available():                                # same sink (= sum of top-1 scores) as the real reference binary
        ref = po.Reference().run_tool("nvdb_bench", files["b16"], files["q"], 10, mode, *extra, env={"OMP_NUM_THREADS": "2"})
        assert re.search(r"sink=(\S+)", out).group(1) == re.search(r"sink=(\S+)", ref).group(1)
        assert [re.split(r"[=:]", l)[0] for l in ref.strip().splitlines()[2:]] == keys


def test_nvdb_bench_rejects_bad_input(files, built):
    r = subprocess.run([os.path.join(BIN, "nvdb_bench")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Usage: nvdb_bench")
    base32, _ = make_case_inputs("tiny")
    p = os.path.join(os.path.dirname(files["q"]), "tiny.vecbin")
    po.write_vecbin(p, base32, po.DT_F32)
    r = subprocess.run([os.path.join(BIN, "nvdb_bench"), p, files["q"], "3"], capture_output=True, text=True)
    assert r.returncode == 2 and "Dim mismatch" in r.stderr
    r = subprocess.run([os.path.join(BIN, "nvdb_bench"), files["b16"], files["q"], "3", "async", "2", "0", "4"], capture_output=True, text=True)
    assert r.returncode == 3 and "batch_q>1 supported only" in r.stderr       # reference: uncaught runtime_error, apps/nvdb_bench.cpp:345
    r = subprocess.run([os.path.join(BIN, "nvdb_bench"), files["b16"], files["q"], "3", "warp"], capture_output=True, text=True)
    assert r.returncode == 3 and "Unknown mode" in r.stderr


def test_loader_rejects_what_the_reference_rejects(files, built, tmp_path):
    """Loader validation (reference src/vector_dataset.cpp:47-70, 98-108): size mismatch under a valid vecbin header,
    files too small for either header, raw12 with zero dim, raw12 size mismatch.  Same messages; like the reference's
    tools ours let the exception escape (terminate -> SIGABRT), and both binaries are compared when the real one exists."""
    good = open(files["b16"], "rb").read()
    cases = {
        "truncated.vecbin": (good[:-2], "VecbinHeader ok but file size mismatch"),
        "padded.vecbin": (good + b"\0" * 8, "VecbinHeader ok but file size mismatch"),
        "tiny.bin": (b"\1\2\3\4", "File too small (neither vecbin64 nor raw12)"),
        "zero_dim.raw12": (struct.pack("<III", 5, 0, 0) + b"\0" * 64, "raw12 header invalid (count/dim == 0)"),
        "short.raw12": (struct.pack("<III", 5, 0, 8) + b"\0" * 100, "raw12 header parsed but file size mismatch"),
    }
    for name, (blob, msg) in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        r = subprocess.run([os.path.join(BIN, "nvdb_search"), p, files["q"], "3"], capture_output=True, text=True)
        assert r.returncode != 0 and msg in r.stderr, (name, r.returncode, r.stderr[-300:])
        if po.Reference.available():
            ref = subprocess.run([os.path.join(po.Reference().bin, "nvdb_search"), p, files["q"], "3"], capture_output=True, text=True)
            assert ref.returncode == r.returncode and msg in ref.stderr, (name, ref.returncode, ref.stderr[-300:])


def test_cpu_modes_under_address_and_ub_sanitizers(files, golden, tmp_path):
    """The host layer (loader, SIMD dots, top-k buffers, the four CPU threadings, converters, gt writer) built with
    -fsanitize=address,undefined must run clean and still produce the reference's bytes.  CPU build only."""
    pkg = os.path.join(ROOT, "nano-vectordb_amd")
    subprocess.check_call(["make", "-C", pkg, "asan", "-j4"], stdout=subprocess.DEVNULL)
    abin = os.path.join(pkg, "bin_asan")
    env = dict(os.environ, OMP_NUM_THREADS="2", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")

    def arun(tool, *args, extra=None):
        e = dict(env); e.update(extra or {})
        r = subprocess.run([os.path.join(abin, tool), *map(str, args)], capture_output=True, text=True, env=e)
        assert r.returncode == 0, f"{tool} {args}: rc={r.returncode}\n{r.stderr[-2000:]}"
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
        return r.stdout
    b16, b8 = str(tmp_path / "b16.vecbin"), str(tmp_path / "b8.vecbin")
    arun("nvdb_convert_f16", files["b32"], b16)
    arun("nvdb_quantize_i8", files["b32"], b8)
    assert open(b16, "rb").read() == open(files["b16"], "rb").read() and open(b8, "rb").read() == open(files["b8"], "rb").read()
    assert arun("nvdb_search", files["b32"], files["q"], 10) == bytes(golden["main768_search_stdout"]).decode()
    sinks = set()
    for mode, extra in (("st", []), ("omp", ["2"]), ("async", ["3"]), ("pool", ["3"]), ("omp", ["2", "1", "4", "512", "0"]), ("pool", ["2", "1", "4", "512", "0"])):
        for base in (files["b16"], b8):
            out = arun("nvdb_bench", base, files["q"], 10, mode, *extra)
            sinks.add((base == b8, re.search(r"sink=(\S+)", out).group(1)))
    assert len(sinks) == 2                                      # one sink per dtype, whatever the threading
    gt = str(tmp_path / "gt.gtbin")
    for mode in ("st", "pool"):
        arun("nvdb_gt_build", files["b16"], files["q"], 10, gt, extra={"GT_MODE": mode})
        assert np.array_equal(po.read_gtbin(gt)[0], golden["main768_gtbin_f16_ids"])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_modes_of_the_tools(files, golden):
    assert run("nvdb_search", files["b32"], files["q"], 10, "gpu") == bytes(golden["main768_search_stdout"]).decode()
    run("nvdb_gt_build", files["b16"], files["q"], 10, files["gt"], env={"GT_MODE": "gpu"})
    assert np.array_equal(po.read_gtbin(files["gt"])[0], golden["main768_gtbin_f16_ids"])
    # k = 100 (> 64, the any-k path): the reference's tool takes any k (apps/nvdb_gt_build.cpp, flat_index.cpp:24)
    run("nvdb_gt_build", files["b16"], files["q"], 100, files["gt"], env={"GT_MODE": "gpu"})
    g100 = po.read_gtbin(files["gt"])[0].copy()
    run("nvdb_gt_build", files["b16"], files["q"], 100, files["gt"], env={"OMP_NUM_THREADS": "3"})        # default mode = omp, as in the reference
    assert g100.shape == (8, 100) and np.array_equal(g100, po.read_gtbin(files["gt"])[0])
    cpu = run("nvdb_bench", files["b16"], files["q"], 10, "st")
    for extra in ([], ["0", "2", "4"]):
        out = run("nvdb_bench", files["b16"], files["q"], 10, "gpu", *extra)
        assert re.search(r"sink=(\S+)", out).group(1) == re.search(r"sink=(\S+)", cpu).group(1)
        assert "gpu_kernel_ms_total=" in out.splitlines()[-1]
    # row-sharded over "two GPUs" (the same device twice on this box): same sink as the unsharded run
    out = run("nvdb_bench", files["b16"], files["q"], 10, "gpu", "0", "1", "8", env={"NVDB_GPU_DEVICES": "0,0,0"})
    assert re.search(r"sink=(\S+)", out).group(1) == re.search(r"sink=(\S+)", cpu).group(1)
    assert "gpu_shards=3" in out.splitlines()[-1]


@pytest.mark.gpu
@pytest.mark.parametrize("dtkey,pinned", [("b16", "0"), ("b32", "0"), ("b16", "1")])
def test_nvdb_cuda_refine_eval(files, dtkey, pinned):
    out = run("nvdb_cuda_refine_eval", files[dtkey], files["q"], 10, env={"REFINE_K": "256", "CUDA_PINNED": pinned})
    assert out.startswith("CUDA_REFINE=1 refine_ms_total=")
    res = dict(kv.split("=") for kv in out.strip().splitlines()[-1].split()[1:])
    assert res["refine_k"] == "256" and res["Q"] == "8" and res["k"] == "10" and res["refine_backend"] == "cuda"
    assert float(res["recall_vs_cpu"]) == 1.0 and res["cuda_pinned"] == pinned
    assert "identical_rows=8/8" in out


@pytest.mark.parametrize("sanitize", [False, True])
def test_call_coalescer_without_a_gpu(tmp_path, sanitize):
    """nvdb::detail::CallCoalescer (what lets FlatIndexHIP / FlatIndexHIPSharded serve overlapping callers with ONE GPU batch) driven by
    tests/coalescer_check.cpp: a stand-in batch function that sleeps 200 us whatever the batch size; six threads x 50 calls
    (single queries, some 3-query requests, some with another k).  Every caller gets its own rows, no two batches overlap, batches
    are shared, an injected failure reaches its callers and only them.  Once plain (timing: six threads in well under 6x the solo
    time) and once under ThreadSanitizer."""
    exe = str(tmp_path / "coalescer_check")
    flags = ["-O1", "-g", "-fsanitize=thread"] if sanitize else ["-O2"]
    subprocess.run(["g++", "-std=c++17", *flags, "-I", os.path.join(ROOT, "nano-vectordb_amd", "host", "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "coalescer_check.cpp"), "-lpthread"], check=True)
    r = subprocess.run([exe, "6", "50"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    kv = dict(p.split("=") for p in r.stdout.splitlines()[0].split())
    assert int(kv["overlaps"]) == 0 and int(kv["wrong"]) == 0 and int(kv["max_batch"]) > 1
    if not sanitize:
        assert float(kv["par_ms"]) < 3.0 * float(kv["solo_ms"]), kv          # serialised it would be 6x
    one = subprocess.run([exe, "1", "20"], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stdout + one.stderr
