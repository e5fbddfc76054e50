"""Device group of the C ABI (nvdb_hip_group_*: one process, several GPUs, RCCL all-gather of the per-shard top-k, merge
on devices[0]) and the RCCL branch of bench.py, on the ONE GPU a test box has:

  * devices = [0]        -> a real RCCL communicator of one rank: ncclCommInitAll, ncclGroupStart / ncclAllGather / End on
                            the group's stream, merge_topk_kernel out of the gathered buffer.  Result == the plain search.
  * devices = [0, 0, 0]  -> RCCL refuses a device listed twice; the exchange is three peer copies into devices[0] and the
                            same merge kernel: three shards with global ids == the unsharded search, bit for bit.
  * bench.py with NVDB_BENCH_FORCE_COLLECTIVE=1 and one rank: init_process_group("nccl"), the packed uint8 all-gather,
    nvdb_hip_merge_topk_strided_dev, merge_check -- the code path the 8-GPU run takes.
  * devices = [0] * G with NVDB_GROUP_RCCL_LIB naming tests/loopback_rccl's stand-in -> the RCCL branch with G = 2, 3, 8 ranks.
The 8-GPU exchange itself cannot run here; what these tests show is that every call of that path initialises, orders
its streams and merges correctly."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import nvdb_amd
import pyoracle as po

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20240613


def _reference_search(n, d, dtype, queries, k, seed=SEED + 30):
    c = nvdb_amd.HipContext(0)
    c.generate_corpus(seed, n, d, dtype)
    ids, sc = c.search_batch(queries, k)
    c.close()
    return ids, sc


@pytest.mark.parametrize("devices,mode", [([0], "rccl"), ([0, 0, 0], "peer-copy")])
@pytest.mark.parametrize("dtype", [nvdb_amd.DT_F16, nvdb_amd.DT_I8])
def test_group_search_equals_the_unsharded_search(devices, mode, dtype):
    n, d, nq, k = 700_000, 768, 300, 10                        # shards big enough for the MFMA filter path
    queries = nvdb_amd.synth_rows_f32(SEED + 31, 0, nq, d)
    fi, fs = _reference_search(n, d, dtype, queries, k)
    g = nvdb_amd.DeviceGroup(devices)
    got_mode, why = g.exchange()
    assert got_mode == mode, (got_mode, why)
    assert g.size() == len(devices)
    g.generate_corpus(SEED + 30, n, d, dtype)
    for rep in range(2):                                        # second call reuses buffers and communicator
        ids, sc, st = g.search_batch(queries, k, want_stats=True)
        assert st["shards"] == len(devices) and st["exchange"] == (1 if mode == "rccl" else 0)
        assert st["host_merge_fallbacks"] == 0 and st["bytes_per_rank"] == nq * k * 12
        assert np.array_equal(ids, fi) and np.array_equal(sc.view(np.uint32), fs.view(np.uint32)), (mode, rep)
    assert g.shard_stats(0)["path"] == 2
    g.close()


LOOPBACK = os.path.join(ROOT, "tests", "loopback_rccl", "libloopback_rccl.so")


def _loopback_counters():
    import ctypes as C
    L = C.CDLL(LOOPBACK, mode=C.RTLD_GLOBAL)
    out = (C.c_int * 4)()
    L.nvdb_loopback_rccl_counters(out)
    return list(out)


@pytest.mark.parametrize("G", [2, 3, 8])
def test_group_rccl_branch_with_several_ranks_through_the_loopback_stand_in(G, monkeypatch):
    """The group's RCCL branch -- ncclCommInitAll, ncclGroupStart / G x ncclAllGather / ncclGroupEnd on the G streams, the merge out
    of devices[0]'s gathered buffer -- with G ranks on ONE device.  RCCL itself refuses a repeated device; NVDB_GROUP_RCCL_LIB
    names a stand-in (tests/loopback_rccl, test infrastructure) that implements the all-gather's ordering semantics with stream
    events and copies into recv + rank * count.  What this executes of the 8-GPU path: receive offsets, which buffers are handed
    to which rank's call, stream order between search, exchange, merge and the next batch, communicator reuse."""
    if not os.path.exists(LOOPBACK):
        subprocess.run(["make", "-C", os.path.dirname(LOOPBACK)], check=True)
    monkeypatch.setenv("NVDB_GROUP_RCCL_LIB", LOOPBACK)
    n, d, k = 600_000, 768, 10
    batches = [nvdb_amd.synth_rows_f32(SEED + 70 + i, 0, nq, d) for i, nq in enumerate((300, 64, 1024, 5))]
    want = [_reference_search(n, d, nvdb_amd.DT_F16, q, k) for q in batches]
    c0 = _loopback_counters()
    g = nvdb_amd.DeviceGroup([0] * G)
    mode, why = g.exchange()
    assert mode == "rccl" and "NVDB_GROUP_RCCL_LIB" in why, (mode, why)
    g.generate_corpus(SEED + 30, n, d, nvdb_amd.DT_F16)
    calls = 0
    for rep in range(2):                                        # second round: same communicators, same buffers
        for q, (fi, fs) in zip(batches, want):                  # different batch sizes back to back: a stale block would show
            ids, sc, st = g.search_batch(q, k, want_stats=True)
            calls += 1
            assert st["shards"] == G and st["exchange"] == 1 and st["host_merge_fallbacks"] == 0
            assert st["bytes_per_rank"] == len(q) * k * 12
            assert np.array_equal(ids, fi) and np.array_equal(sc.view(np.uint32), fs.view(np.uint32)), (G, rep, len(q))
    c1 = _loopback_counters()
    assert c1[0] - c0[0] == 1, "communicators must be created once per group"
    assert c1[1] - c0[1] == G * calls and c1[2] - c0[2] == calls
    g.close()
    assert _loopback_counters()[3] - c0[3] == G


def test_group_uploads_a_host_corpus_in_row_shards(oracle):
    n, d, nq, k = 90_000, 768, 40, 10
    base = oracle.f32_to_f16(nvdb_amd.synth_rows_f32(SEED + 32, 0, n, d))
    queries = nvdb_amd.synth_rows_f32(SEED + 33, 0, nq, d)
    oid, osc = oracle.flat_topk(base, po.DT_F16, queries[:16], k)
    for devices in ([0], [0, 0], [0, 0, 0, 0, 0]):
        g = nvdb_amd.DeviceGroup(devices)
        g.upload_corpus(base, nvdb_amd.DT_F16)
        ids, sc = g.search_batch(queries, k)
        assert np.array_equal(ids[:16], oid) and np.array_equal(sc[:16].view(np.uint32), osc.view(np.uint32)), devices
        one, ones = g.search_batch(queries[3], k)               # single query, 1-D
        assert np.array_equal(one[0], ids[3]) and np.array_equal(ones[0].view(np.uint32), sc[3].view(np.uint32))
        g.close()


def test_group_more_than_1024_queries_and_large_k():
    n, d = 200_000, 768
    queries = nvdb_amd.synth_rows_f32(SEED + 34, 0, 1300, d)
    g = nvdb_amd.DeviceGroup([0, 0])
    g.generate_corpus(SEED + 30, n, d, nvdb_amd.DT_F16)
    fi, fs = _reference_search(n, d, nvdb_amd.DT_F16, queries, 10)
    ids, sc, st = g.search_batch(queries, 10, want_stats=True)
    assert st["host_merge_fallbacks"] == 0
    assert np.array_equal(ids, fi) and np.array_equal(sc.view(np.uint32), fs.view(np.uint32))
    # 2 shards x k = 3000 > 4096 entries per query: beyond the LDS-ranked merge -> the sorted-list merge on the device, same answer
    fi, fs = _reference_search(n, d, nvdb_amd.DT_F16, queries[:20], 3000)
    ids, sc, st = g.search_batch(queries[:20], 3000, want_stats=True)
    assert st["host_merge_fallbacks"] == 0
    assert np.array_equal(ids, fi) and np.array_equal(sc.view(np.uint32), fs.view(np.uint32))
    # k = 1000 at 8 shards (8000 entries per query; the reference bounds k by N only, src/flat_index.cpp:24)
    g8 = nvdb_amd.DeviceGroup([0] * 8)
    g8.generate_corpus(SEED + 30, n, d, nvdb_amd.DT_F16)
    fi, fs = _reference_search(n, d, nvdb_amd.DT_F16, queries[:12], 1000)
    ids, sc, st = g8.search_batch(queries[:12], 1000, want_stats=True)
    assert st["host_merge_fallbacks"] == 0 and st["shards"] == 8
    assert np.array_equal(ids, fi) and np.array_equal(sc.view(np.uint32), fs.view(np.uint32))
    g8.close()
    # k beyond a shard's row count: clamped to the corpus size, padded shard lists merge correctly
    g2 = nvdb_amd.DeviceGroup([0, 0, 0])
    g2.generate_corpus(SEED + 30, 3000, d, nvdb_amd.DT_F16)
    fi, fs = _reference_search(3000, d, nvdb_amd.DT_F16, queries[:5], 1200)
    ids, sc = g2.search_batch(queries[:5], 1200)
    assert ids.shape == (5, 1200) and np.array_equal(ids, fi) and np.array_equal(sc.view(np.uint32), fs.view(np.uint32))
    g2.close()
    g.close()


def test_group_shard_overflow_goes_through_the_host_merge(oracle):
    """A shard whose candidate lists overflow (tiny lists forced here) trips its self-check after the exchange: the
    sub-batch is redone through the per-shard host API (which retries by itself) + host merge; the answer stays exact."""
    n, d, k = 120_000, 768, 10
    base32 = nvdb_amd.synth_rows_f32(SEED + 11, 0, n, d)
    q = nvdb_amd.synth_rows_f32(SEED + 12, 0, 16, d)
    base = oracle.f32_to_f16(base32[np.argsort(base32 @ q[0])])          # ascending similarity to query 0
    g = nvdb_amd.DeviceGroup([0, 0])
    g.upload_corpus(base, nvdb_amd.DT_F16)
    g.set_option("path", 2)
    g.set_option("cand_cap", 64)
    g.set_option("tile_permute", 0)
    ids, sc, st = g.search_batch(q, k, want_stats=True)
    assert st["host_merge_fallbacks"] == 1, st
    oid, osc = oracle.flat_topk(base, po.DT_F16, q, k)
    assert np.array_equal(ids, oid) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
    g.close()


def test_group_argument_conventions():
    g = nvdb_amd.DeviceGroup([0])
    with pytest.raises(nvdb_amd.NvdbError) as e:
        g.search_batch(np.zeros((1, 8), np.float32), 3)
    assert e.value.status == 4 and "Empty base" in str(e.value)          # flat_index.cpp:17
    g.generate_corpus(SEED, 1000, 64, nvdb_amd.DT_F32)
    ids, sc = g.search_batch(np.zeros((2, 64), np.float32), 0)           # k == 0 -> nothing (flat_index.cpp:18)
    assert ids.shape == (2, 0)
    with pytest.raises(nvdb_amd.NvdbError):
        g.set_option("no_such_option", 1)
    g.close()
    with pytest.raises(nvdb_amd.NvdbError) as e:
        nvdb_amd.DeviceGroup([99])
    assert e.value.status in (1, 2)


def test_sharded_cli_uses_the_group(tmp_path, oracle):
    """nvdb_bench ... gpu with NVDB_GPU_DEVICES: FlatIndexHIPSharded over the group API; same sink as the CPU mode."""
    from golden_inputs import make_case_inputs
    base32, queries = make_case_inputs("main768")
    b16, qf = str(tmp_path / "b16.vecbin"), str(tmp_path / "q.raw12")
    po.write_vecbin(b16, oracle.f32_to_f16(base32), po.DT_F16)
    po.write_raw12(qf, queries)
    tool = os.path.join(ROOT, "nano-vectordb_amd", "bin", "nvdb_bench")
    cpu = subprocess.run([tool, b16, qf, "10", "st"], check=True, capture_output=True, text=True).stdout
    for devs, want in (("0,0,0", "peer-copy"), ):
        out = subprocess.run([tool, b16, qf, "10", "gpu", "0", "1", "8"], check=True, capture_output=True, text=True,
                             env=dict(os.environ, NVDB_GPU_DEVICES=devs)).stdout
        assert re.search(r"sink=(\S+)", out).group(1) == re.search(r"sink=(\S+)", cpu).group(1)
        last = out.splitlines()[-1]
        assert "gpu_shards=3" in last and f"gpu_exchange={want}" in last and "gpu_host_merge_fallbacks=0" in last and "gpu_upload_s=" in last


def test_bench_rccl_branch_with_one_rank():
    """bench.py's N > 1 step with world = 1: init_process_group("nccl"), the packed all-gather on the bench stream,
    nvdb_hip_merge_topk_strided_dev, and merge_check (merged lists == a second, unsharded context's answer)."""
    env = dict(os.environ, NVDB_BENCH_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--rows", "2000000",
                        "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["merge_check"] is True, line
    assert "backend nccl" in line["exchange"] and "world 1" in line["exchange"]
    assert line["parity"].startswith("ok") and line["scan"]["path"] == 2 and line["scan"]["bound_violations"] == 0
    assert line["n_gpus"] == 1 and line["value"] > 0
    # every N > 1 line is self-contained: its own 1-GPU point of the same corpus and the search / all-gather / merge split
    one = line["one_gpu_same_corpus"]
    assert one["qps"] > 0 and 0.7 < one["strong_scaling_efficiency"] < 1.3, one          # world 1: the "sharded" run IS the one-GPU run
    sp = line["step_split_ms"]
    assert sp["search_ms"] > 0 and sp["allgather_ms"] >= 0 and sp["merge_ms"] > 0
    assert sp["search_ms"] + sp["allgather_ms"] + sp["merge_ms"] < 1.3 * line["ms_per_step"], (sp, line["ms_per_step"])


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_index_objects_take_concurrent_callers(tmp_path, oracle, devices):
    """nvdb::FlatIndexHIP / FlatIndexHIPSharded::search_topk_dot are const like the reference's FlatIndex, whose callers may
    overlap (threading); the device context is single-owner, so the classes serialise.  Six host threads, one
    query at a time each, against the batched answer."""
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "nano-vectordb_amd", "lib", "libnvdb_host_capi.so"))
    L.nvdb_host_dataset_open.restype = C.c_void_p
    L.nvdb_host_dataset_open.argtypes = [C.c_char_p]
    L.nvdb_host_dataset_close.argtypes = [C.c_void_p]
    L.nvdb_host_last_error.restype = C.c_char_p
    L.nvdb_host_hip_concurrent_search.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    n, d, nq, k = 150_000, 384, 96, 10
    base = oracle.f32_to_f16(nvdb_amd.synth_rows_f32(SEED + 60, 0, n, d))
    queries = np.ascontiguousarray(nvdb_amd.synth_rows_f32(SEED + 61, 0, nq, d))
    p = str(tmp_path / "b16.vecbin")
    po.write_vecbin(p, base, po.DT_F16)
    c = nvdb_amd.HipContext(0)
    c.upload_corpus(base, nvdb_amd.DT_F16)
    want_i, want_s = c.search_batch(queries, k)
    c.close()
    h = L.nvdb_host_dataset_open(p.encode())
    assert h, L.nvdb_host_last_error()
    ids = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32)
    dev = (C.c_int * len(devices))(*devices) if devices else None
    rc = L.nvdb_host_hip_concurrent_search(h, queries.ctypes.data, nq, k, 6, dev, len(devices) if devices else 0, ids.ctypes.data, sc.ctypes.data)
    L.nvdb_host_dataset_close(h)
    assert rc == 0, L.nvdb_host_last_error()
    assert np.array_equal(ids, want_i) and np.array_equal(sc.view(np.uint32), want_s.view(np.uint32))


def test_overlapping_single_query_callers_share_gpu_batches(tmp_path, oracle):
    """nvdb::FlatIndexHIP::search_topk_dot coalesces callers that overlap (host/src/flat_index_hip.cpp, CallCoalescer): six threads
    x 50 single queries must finish in less than three times one thread's 50 (measured 1.2-1.4x; serialised under a mutex ~6x), and
    every query's rows equal the batched answer bit for bit.  Reference: FlatIndex::search_topk_dot is const and re-entrant
    (include/nvdb/flat_index.h:11-16)."""
    import ctypes as C
    L = C.CDLL(os.path.join(ROOT, "nano-vectordb_amd", "lib", "libnvdb_host_capi.so"))
    L.nvdb_host_dataset_open.restype = C.c_void_p
    L.nvdb_host_dataset_open.argtypes = [C.c_char_p]
    L.nvdb_host_dataset_close.argtypes = [C.c_void_p]
    L.nvdb_host_last_error.restype = C.c_char_p
    L.nvdb_host_hip_concurrent_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    n, d, nq, k, threads, reps = 500_000, 384, 96, 10, 6, 50      # the reference's own corpus size and dimension (Performance.md:43)
    base = oracle.f32_to_f16(nvdb_amd.synth_rows_f32(SEED + 62, 0, n, d))
    queries = np.ascontiguousarray(nvdb_amd.synth_rows_f32(SEED + 63, 0, nq, d))
    p = str(tmp_path / "b16.vecbin")
    po.write_vecbin(p, base, po.DT_F16)
    c = nvdb_amd.HipContext(0)
    c.upload_corpus(base, nvdb_amd.DT_F16)
    want_i, want_s = c.search_batch(queries, k)
    c.close()
    h = L.nvdb_host_dataset_open(p.encode())
    assert h, L.nvdb_host_last_error()
    ids = np.zeros((nq, k), np.uint64); sc = np.zeros((nq, k), np.float32)
    ms = (C.c_double * 2)()
    best = None
    for attempt in range(3):                                    # wall-clock ratio on a shared host: best of three
        rc = L.nvdb_host_hip_concurrent_timing(h, queries.ctypes.data, nq, k, threads, reps, ids.ctypes.data, sc.ctypes.data, ms)
        assert rc == 0, L.nvdb_host_last_error()
        assert np.array_equal(ids, want_i) and np.array_equal(sc.view(np.uint32), want_s.view(np.uint32))
        ratio = ms[1] / ms[0]
        best = ratio if best is None else min(best, ratio)
        print(f"one thread x {reps}: {ms[0]:.2f} ms; {threads} threads x {reps}: {ms[1]:.2f} ms; ratio {ratio:.2f}")
        if best < 2.0:
            break
    L.nvdb_host_dataset_close(h)
    assert best < 3.0, best
