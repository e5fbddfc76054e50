#!/usr/bin/env python3
"""bench.py -- flat-scan top-10, d=768, fp16 corpus, batch 1024 (BASELINE.json configs[1] on one GPU, configs[3] on N > 1).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process NEVER touches the GPU; it starts N ranks of
itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) and waits for them.
Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks are already there.

A "step" is one pass of the hot path over one batch: 1024 fp32 queries (already in HBM) against the resident corpus
-> exact top-10 (ids + scores) in HBM.
  * N = 1: corpus = 10M rows (configs[1]).
  * N > 1: corpus = 100M rows (configs[3]), row-sharded: rank r holds rows [r*N_rows/N, (r+1)*N_rows/N) with global ids,
    every rank scans its shard for the whole batch, ONE RCCL all-gather of the packed per-shard top-k (B*k*12 bytes per
    rank) and a k-way merge on every rank.  Total work is fixed -> "scaling": "strong"; the 1-GPU point of this series
    is `extras.fp16_100M_batch1024` of the --gpus 1 line (the whole 100M corpus resident on one GPU).

Rank 0 prints ONE JSON line.  `value` = queries/s with queries and results resident in HBM (contract); the same
step through the host API (pageable host queries in, results out, self-check read back: PCIe-inclusive) is
`host_api`.  `roofline`: the dominant kernel timed live by HIP events attached to its launches.  `cpu_baseline`: the
REAL reference (oracle/_ref: its nvdb_bench binary, AVX2+FMA+F16C, OpenMP) on this box's host cores over the full
10M-row corpus -- per-query OMP (the north star's "AVX2+OMP CPU path"), single thread, and the batched OMP loop.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SEED = 20240613
PEAK_F16_TFLOPS = 2500.0     # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_I8_TOPS = 5000.0        # int8 MFMA = 2x the bf16 rate per clock (same guide, "Matrix cores")
PEAK_HBM_GBPS = 8000.0       # HBM3E 8 TB/s spec (same guide)
ROWS_1GPU = 10_000_000       # BASELINE configs[1]
ROWS_SHARDED = 100_000_000   # BASELINE configs[3]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=0, help="total corpus rows (all GPUs together); default 10M on one GPU, 100M on several")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--dtype", default="f16", choices=["f16", "i8"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="CPU time the cpu_baseline leg may spend on its timed queries (bounded sample)")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 exact kernel, 2 MFMA filter")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (nvdb_hip_set_option), repeatable")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (HBM-bound point, int8, refine, 100M)")
    ap.add_argument("--no-verify-merge", action="store_true", help="N>1: skip the merged == unsharded check of the first step")
    ap.add_argument("--sweep", default="", help="comma list of extra batch sizes to time (N=1 only), e.g. 1,16,64,256")
    ap.add_argument("--dry-launch", action="store_true", help="start the ranks, rendezvous over gloo on the CPU, exchange one tensor, exit (no GPU)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ launcher (no GPU touched)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """Parent of a `--gpus N` run started without a launcher: spawn the N ranks as children of a process that has not
    initialised HIP (no torch.cuda call, libnvdb_hip not loaded) and only waits.  Rank 0 inherits stdout."""
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), LOCAL_WORLD_SIZE=str(n), NVDB_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = None
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0:
                rc = rc or code
                deadline = deadline or time.time() + 30          # a rank died: give the others a moment, then stop them
        if deadline and time.time() > deadline:
            for p in procs:
                p.kill()
        time.sleep(0.05)
    return rc


def dry_launch(args, rank, world):
    """Launcher rehearsal on the CPU: gloo rendezvous + one all-gather of the packed result shape + host merge."""
    import torch
    import torch.distributed as dist
    import nvdb_amd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, K = 4, args.k
    ids = (torch.arange(B * K, dtype=torch.int64).view(B, K) * world + rank)
    sc = (-(torch.arange(B * K, dtype=torch.float32).view(B, K) * world + rank))
    g_ids = torch.empty((world * B, K), dtype=torch.int64)
    g_sc = torch.empty((world * B, K), dtype=torch.float32)
    dist.all_gather_into_tensor(g_ids, ids)
    dist.all_gather_into_tensor(g_sc, sc)
    mi, ms = nvdb_amd.merge_topk_host(g_ids.view(world, B, K).numpy().view(np.uint64), g_sc.view(world, B, K).numpy())
    ok = bool((mi[:, 0] == np.arange(B) * K * world).all())
    dist.barrier()
    if rank == 0:
        rows = args.rows or (ROWS_1GPU if world == 1 else ROWS_SHARDED)
        print(json.dumps({"dry_launch": True, "world": world, "merge_ok": ok, "rows_total": rows,
                          "rows_per_gpu": rows // world, "launcher": "self" if os.environ.get("NVDB_BENCH_SELF_LAUNCHED") else "external"}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


# ------------------------------------------------------------------------------------------ CPU baseline (reference binaries)
def host_cpu_info():
    """threads this process may use (affinity, cgroup quota, NVDB_CPU_THREADS, at most 16 per visible GPU = the box's stated
    CPU share), plus what the host is."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    if os.environ.get("NVDB_CPU_THREADS"):
        n = max(1, int(os.environ["NVDB_CPU_THREADS"]))
    else:
        n = min(n, 16)
    model, sockets, phys = "unknown", 1, None
    try:
        txt = open("/proc/cpuinfo").read()
        m = re.search(r"model name\s*:\s*(.+)", txt)
        model = m.group(1).strip() if m else model
        sockets = max(1, len(set(re.findall(r"physical id\s*:\s*(\d+)", txt))))
        phys = len(set(re.findall(r"physical id\s*:\s*(\d+)\n(?:.*\n)*?core id\s*:\s*(\d+)", txt))) or None
    except Exception:
        pass
    return {"threads_used": n, "cpu_model": model, "sockets": sockets, "physical_cores_visible": phys, "logical_cpus_visible": os.cpu_count()}


def scratch_dir(need_bytes):
    """/dev/shm when it has the room (the file then sits in RAM like the reference's warm page cache), else $TMPDIR or /tmp."""
    for d in ("/dev/shm", os.environ.get("TMPDIR", ""), "/tmp"):
        try:
            if d and os.path.isdir(d):
                v = os.statvfs(d)
                if v.f_bavail * v.f_frsize > need_bytes * 1.05:
                    return d
        except OSError:
            pass
    return "/tmp"


def _ref_bench(ref_bin, base, qfile, k, mode, threads, warmup, batch_q=1, tile=512, timeout=600):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
    cmd = [os.path.join(ref_bin, "nvdb_bench"), base, qfile, str(k), mode, str(threads), str(warmup)]
    if batch_q > 1:
        cmd += [str(batch_q), str(tile), "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, check=True).stdout
    m = re.search(r"Avg_query:\s*([\d.]+) ms/query\s*\(([\d.]+) QPS\)", out)
    sink = re.search(r"sink=(\S+)", out)
    return float(m.group(1)), float(m.group(2)), sink.group(1) if sink else None


def cpu_baseline(ctx, args, nvdb_amd, N):
    """The reference's CPU path on the FULL corpus of this run (no extrapolation): rows copied back from HBM into a
    vecbin in /dev/shm, then the reference's own nvdb_bench binary (oracle/_ref/bin, built from /root/reference by
    oracle/Makefile) -- per-query OpenMP = FlatIndexOMP (the headline `value`), single thread = FlatIndex, and the
    bench-side batched OpenMP loop (apps/nvdb_bench.cpp:47-159, batch_q=8 tile_vecs=512, the published setting).
    Query counts are sized from a first timed query so that the leg stays within --cpu-seconds."""
    import pyoracle as po
    info = host_cpu_info()
    T = info["threads_used"]
    D, K = args.dim, args.k
    dt_o = po.DT_F16 if args.dtype == "f16" else po.DT_I8
    bpr = D * 2 if args.dtype == "f16" else D + 4
    shm = scratch_dir(N * bpr + (64 << 20))
    base_p = os.path.join(shm, f"nvdb_bench_{os.getpid()}.vecbin")
    q_p = os.path.join(shm, f"nvdb_bench_{os.getpid()}_q.raw12")
    res = {"unit": "queries/s", "cores": T, "host": info,
           "cores_note": f"{T} threads = min(CPU affinity, cgroup quota, 16 per GPU of the lease): a one-GPU lease of this pool is entitled to 16 host "
                         f"cores (the other {max(0, (info['logical_cpus_visible'] or 0) - T)} visible logical CPUs belong to the other seven GPUs' tenants); NVDB_CPU_THREADS overrides"}
    try:
        # corpus file: header + payload streamed from HBM in slabs (+ scales for int8); 15.36 GB for the headline config
        import struct
        slab = 500_000
        with open(base_p, "wb") as f:
            f.write(struct.pack("<QIIIIQ", po.VEC_MAGIC, 1, dt_o, D, 0, N) + b"\0" * 32)
            scales = []
            for r0 in range(0, N, slab):
                rows, sc = ctx.download_rows(r0, min(slab, N - r0))
                f.write(rows.tobytes())
                if sc is not None:
                    scales.append(sc)
            if scales:
                f.write(np.concatenate(scales).astype(np.float32).tobytes())
        if not args.no_extras:
            res["_cli"] = cli_legs(nvdb_amd, args, base_p, N, shm)     # the named CLI at this config, while the file exists
        try:
            queries = nvdb_amd.synth_rows_f32(SEED + 1, 0, 256, D)
            if not po.Reference.available():
                raise RuntimeError("oracle/_ref is missing (built only where /root/reference exists)")
            ref_bin = po.Reference().bin
            budget = args.cpu_seconds

            def timed(mode, threads, batch_q, share):
                # probe with one query (plus the reference's own warm-up query), then size the run
                po.write_raw12(q_p, queries[:max(1, batch_q)])
                ms1, _, _ = _ref_bench(ref_bin, base_p, q_p, K, mode, threads, 1, batch_q)
                nq = int(max(batch_q, min(256, (budget * share * 1e3) / max(ms1, 1e-3))))
                nq = max(batch_q, nq // batch_q * batch_q)
                po.write_raw12(q_p, queries[:nq])
                ms, qps, sink = _ref_bench(ref_bin, base_p, q_p, K, mode, threads, 1, batch_q)
                return {"qps": qps, "ms_per_query": ms, "queries": nq, "threads": threads, "GBps": N * bpr * qps / 1e9, "sink": sink}
            omp = timed("omp", T, 1, 0.45)
            st = timed("st", 1, 1, 0.2)
            bat = timed("omp", T, 8, 0.35)
            res.update({"value": omp["qps"], "kind": "reference",
                        "sample": f"{omp['queries']} queries one at a time, FlatIndexOMP with {T} OpenMP threads (OMP_PROC_BIND=close OMP_PLACES=cores), "
                                  f"over ALL {N} rows of the same {args.dtype} corpus (d={D}, k={K}); reference binary oracle/_ref/bin/nvdb_bench",
                        "per_query_omp": omp, "single_thread": st, "batched_omp_batch8_tile512": bat})
        except Exception as e:          # the CLI legs above must survive a failure of the reference leg
            res.update({"value": None, "kind": "error", "sample": repr(e)})
    finally:
        for p in (base_p, q_p):
            if os.path.exists(p):
                os.remove(p)
    return res


def _kv_line(line):
    return dict(kv.split("=", 1) for kv in line.split() if "=" in kv)


def cli_legs(nvdb_amd, args, base_p, N, tmpdir):
    """The CLI surface the north star names, timed at BASELINE configs in fresh child processes (nothing of this
    process' GPU state is shared with them):
      * `nvdb_bench <base> <4096 queries> 10 gpu 0 1 1024` on the 10M-row vecbin of the cpu_baseline leg (configs[1]):
        one-time upload, then 16 batches of 1024 through nvdb::FlatIndexHIP -> C ABI host entry (PCIe-inclusive);
      * `nvdb_cuda_refine_eval <base> <10000 queries> 10` with REFINE_K=1024 on the first 2.9M rows (configs[4]):
        synthetic candidates, CPU refine (OpenMP) beside the drop-in nvdb::cuda_l2_topk_batch call.
    Output lines are the reference's (apps/nvdb_bench.cpp:379-425; apps/nvdb_ivf_eval.cpp:572-576, 743-779)."""
    import pyoracle as po
    import struct
    D, K = args.dim, args.k
    bin_dir = os.path.join(ROOT, "nano-vectordb_amd", "bin")
    out = {}
    q_p = os.path.join(tmpdir, f"nvdb_cli_{os.getpid()}_q.raw12")
    r_p = os.path.join(tmpdir, f"nvdb_cli_{os.getpid()}_refine.vecbin")
    rq_p = os.path.join(tmpdir, f"nvdb_cli_{os.getpid()}_rq.raw12")
    env = dict(os.environ, OMP_NUM_THREADS=str(host_cpu_info()["threads_used"]))
    try:
        if args.dtype == "f16" or args.dtype == "i8":
            nq_cli = 16 * args.batch                                                     # 16 batches: a fresh process needs a few passes to settle (XCD shares, clocks)
            po.write_raw12(q_p, np.tile(nvdb_amd.synth_rows_f32(SEED + 1, 0, 4 * args.batch, D), (4, 1)))    # the bench's own four query batches, four times
            # under the reference's own timing rules (the query mmap's first-touch page faults inside the timed loop, as in the
            # CPU modes and the reference's harness), then once more with the query file pre-faulted (NVDB_BENCH_PREFAULT=1)
            cmd = [os.path.join(bin_dir, "nvdb_bench"), base_p, q_p, str(K), "gpu", "0", "1", str(args.batch)]
            t0 = time.perf_counter()
            txt = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, check=True).stdout
            wall = time.perf_counter() - t0
            txt_pf = subprocess.run(cmd, env=dict(env, NVDB_BENCH_PREFAULT="1"), capture_output=True, text=True, timeout=600, check=True).stdout
            m = re.search(r"Avg_query:\s*([\d.]+) ms/query\s*\(([\d.]+) QPS\)", txt)
            m_pf = re.search(r"Avg_query:\s*([\d.]+) ms/query\s*\(([\d.]+) QPS\)", txt_pf)
            g = _kv_line(txt.strip().splitlines()[-1])
            out["nvdb_bench_cli"] = {
                "cmd": f"nvdb_bench <{N}x{D} {args.dtype} vecbin> <{nq_cli} queries> {K} gpu 0 1 {args.batch}",
                "Avg_query_ms": float(m.group(1)), "QPS": float(m.group(2)), "sink": re.search(r"sink=(\S+)", txt).group(1),
                "batch_p50_ms": float(re.search(r"batch_p50:\s*([\d.]+)", txt).group(1)),
                "upload_s": float(g["gpu_upload_s"]), "upload_GBps": float(g["gpu_upload_GBps"]), "hip_init_s": float(g.get("gpu_init_s", "nan")),
                "upload_note": "index construction alone (device buffers, mmap page-in through the threaded pinned staging, H2D, row-norm pass); the HIP runtime / device bring-up of a fresh process is hip_init_s",
                "gpu_kernel_ms_total": float(g["gpu_kernel_ms_total"]), "gpu_passes": int(g["gpu_passes"]),
                "gpu_algorithmic_GBps": float(g["gpu_algorithmic_GBps"]), "process_wall_s": wall,
                "timing_rule": "the reference's: query-file page faults inside the timed loop (gpu_prefault=0)",
                "QPS_query_file_prefaulted": float(m_pf.group(2))}
            # `nvdb_search <base> <queries> 10 gpu` on the same file: a fresh process whose whole cost is context + upload + ONE query
            t0 = time.perf_counter()
            stxt = subprocess.run([os.path.join(bin_dir, "nvdb_search"), base_p, q_p, str(K), "gpu"], env=env, capture_output=True, text=True, timeout=600, check=True).stdout
            out["nvdb_search_cli"] = {"cmd": f"nvdb_search <{N}x{D} {args.dtype} vecbin> <queries> {K} gpu", "process_wall_s": time.perf_counter() - t0,
                                      "top1": stxt.splitlines()[1] if len(stxt.splitlines()) > 1 else None,
                                      "note": "wall of the whole process: context creation, corpus upload (threaded pinned staging), one search"}
        if args.dtype == "f16":
            NR, QR = min(N, 2_900_000), 10_000
            with open(base_p, "rb") as src, open(r_p, "wb") as dst:                       # same generator, same rows: a prefix of the 10M-row file
                src.seek(64)
                dst.write(struct.pack("<QIIIIQ", po.VEC_MAGIC, 1, po.DT_F16, D, 0, NR) + b"\0" * 32)
                left = NR * D * 2
                while left:
                    buf = src.read(min(left, 64 << 20))
                    dst.write(buf)
                    left -= len(buf)
            po.write_raw12(rq_p, nvdb_amd.synth_rows_f32(SEED + 2, 0, QR, D))
            t0 = time.perf_counter()
            txt = subprocess.run([os.path.join(bin_dir, "nvdb_cuda_refine_eval"), r_p, rq_p, str(K)], env=dict(env, REFINE_K="1024"),
                                 capture_output=True, text=True, timeout=900, check=True).stdout
            wall = time.perf_counter() - t0
            r = _kv_line([l for l in txt.splitlines() if l.startswith("RESULT")][-1])
            out["nvdb_cuda_refine_eval_cli"] = {
                "cmd": f"REFINE_K=1024 nvdb_cuda_refine_eval <{NR}x{D} f16 vecbin> <{QR} queries> {K}",
                "refine_ms_total": float(r["refine_ms_total"]), "refine_h2d_ms": float(r["refine_h2d_ms"]), "refine_kernel_ms": float(r["refine_kernel_ms"]),
                "refine_d2h_ms": float(r["refine_d2h_ms"]), "refine_kernel_us_per_q": float(r["refine_kernel_ms_per_q"]) * 1e3,
                "gather_GBps": float(r["gather_GBps"]), "hbm_frac": float(r["gather_GBps"]) / PEAK_HBM_GBPS,
                "recall_vs_cpu": float(r["recall_vs_cpu"]), "cpu_refine_ms_total": float(r["cpu_refine_ms_total"]),
                "cpu_refine_threads": int(env["OMP_NUM_THREADS"]), "process_wall_s": wall}
    except Exception as e:
        out["error"] = repr(e) + (" | " + getattr(e, "stderr", "")[-400:] if getattr(e, "stderr", None) else "")
    finally:
        for p in (q_p, r_p, rq_p):
            if os.path.exists(p):
                os.remove(p)
    return out


def _pinned_child(cmd, env, cpu, timeout=300):
    """run a single-threaded CPU child pinned to one CPU (both binaries of an A/B on the SAME core)"""
    # no preexec_fn: this process holds a HIP context (runtime threads alive), and Python code between fork and exec in a
    # multi-threaded parent can deadlock.  The affinity is set on THIS thread and inherited by the child, then restored.
    if cpu is None:
        return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, check=True).stdout
    aff = os.sched_getaffinity(0)
    try:
        os.sched_setaffinity(0, {cpu})
        return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, check=True).stdout
    finally:
        os.sched_setaffinity(0, aff)


def config0_plumbing(nvdb_amd, args):
    """BASELINE configs[0]: fp32 flat scan, N=500K d=768, single-thread CPU via nvdb_bench (no GPU in the measured path).
    The product's host tool (nano-vectordb_amd/bin/nvdb_bench, mode st) and the reference's binary run on the same
    vecbin; their `sink` (sum of top-1 scores) must agree.  Corpus rows come from the device generator.
    A/B hygiene (round 2 measured 52.9 vs 35.1 ms on a 2-socket host with unpinned children): the file is written, and
    BOTH children run, pinned to ONE cpu (same core, same NUMA node as the file's pages); the two binaries alternate
    A B A B and the best of each is reported, with every sample kept."""
    import pyoracle as po
    n, D, K, nq = 500_000, args.dim, args.k, 16
    shm = scratch_dir(n * D * 4 + (64 << 20))
    base_p, q_p = os.path.join(shm, f"nvdb_cfg0_{os.getpid()}.vecbin"), os.path.join(shm, f"nvdb_cfg0_{os.getpid()}_q.raw12")
    out = {"workload": f"fp32 flat-scan top-{K}, N={n} d={D}, single thread, nvdb_bench st"}
    aff = os.sched_getaffinity(0)
    cpu = min(aff)
    try:
        c = nvdb_amd.HipContext(0)
        c.generate_corpus(SEED, n, D, nvdb_amd.DT_F32)
        rows, _ = c.download_rows(0, n)
        c.close()
        os.sched_setaffinity(0, {cpu})                     # first touch of the file's pages from the cpu the children will use
        po.write_vecbin(base_p, rows, po.DT_F32)
        os.sched_setaffinity(0, aff)
        del rows
        po.write_raw12(q_p, nvdb_amd.synth_rows_f32(SEED + 1, 0, nq, D))
        env = dict(os.environ, OMP_NUM_THREADS="1")
        bins = {"host_tool": os.path.join(ROOT, "nano-vectordb_amd", "bin", "nvdb_bench")}
        if po.Reference.available():
            bins["reference"] = os.path.join(po.Reference().bin, "nvdb_bench")
        samples = {k: [] for k in bins}
        sinks = {}
        for _ in range(2):
            for name, exe in bins.items():
                txt = _pinned_child([exe, base_p, q_p, str(K), "st", "1", "1"], env, cpu)
                m = re.search(r"Avg_query:\s*([\d.]+) ms/query\s*\(([\d.]+) QPS\)", txt)
                samples[name].append(float(m.group(1)))
                sinks[name] = re.search(r"sink=(\S+)", txt).group(1)
        for name in bins:
            best = min(samples[name])
            out[name] = {"ms_per_query": best, "qps": 1e3 / best, "sink": sinks[name], "samples_ms": samples[name]}
        out["pinned_cpu"] = cpu
        out["order"] = "host_tool, reference, host_tool, reference (best of 2 each)"
        if "reference" in bins:
            out["sink_equal"] = sinks["reference"] == sinks["host_tool"]
            out["host_tool_over_reference"] = out["host_tool"]["ms_per_query"] / out["reference"]["ms_per_query"]
    finally:
        os.sched_setaffinity(0, aff)
        for p in (base_p, q_p):
            if os.path.exists(p):
                os.remove(p)
    return out


# ------------------------------------------------------------------------------------------ the benchmark
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))          # nothing above this line has touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_launch:
        sys.exit(dry_launch(args, rank, world))

    import torch
    import nvdb_amd

    # NVDB_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box): every rank uses device 0 and the exchange goes through
    # gloo on host copies -- same sharding / all-gather / merge logic, no RCCL (RCCL refuses two ranks on one device).
    share_gpu = os.environ.get("NVDB_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Everything below runs on ONE explicit stream: torch's default stream has the handle 0, which the C ABI reads as
    # "use the context's own (non-blocking) stream" -- searches would then not be ordered with torch's copies and
    # collectives (a stale result buffer would be gathered).
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    # NVDB_BENCH_FORCE_COLLECTIVE=1: take the N > 1 code path (process group, packed all-gather, device merge, merge_check)
    # with whatever world size there is -- with one rank it shows on a one-GPU box that the RCCL branch initialises, orders
    # its streams and reproduces the local search (tests/test_gpu_parity.py::test_bench_rccl_branch_with_one_rank).
    use_dist = world > 1 or os.environ.get("NVDB_BENCH_FORCE_COLLECTIVE") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    dt = nvdb_amd.DT_F16 if args.dtype == "f16" else nvdb_amd.DT_I8
    bpr = args.dim * 2 if args.dtype == "f16" else args.dim + 4      # algorithmic bytes per corpus row
    N = args.rows or (ROWS_1GPU if world == 1 else ROWS_SHARDED)
    B, D, K = args.batch, args.dim, args.k
    from nvdb_amd.sharding import shard_range
    lo, hi = shard_range(N, rank, world)
    ctx = nvdb_amd.HipContext(local_rank)
    ctx.generate_corpus(SEED, hi - lo, D, dt, row_base=lo)     # one-time, excluded like the reference's base H2D
    ctx.set_option("path", args.path)
    for kv in args.opt:
        key, val = kv.split("=")
        ctx.set_option(key, int(val))

    # query batches: independent synthetic rows (no self-match), resident in HBM before timing
    nbatches = 4
    qhost = nvdb_amd.synth_rows_f32(SEED + 1, 0, nbatches * B, D)
    qdev = torch.from_numpy(qhost).to(dev).contiguous()
    # one packed result buffer per rank: [ids: B*K int64 | scores: B*K float32] -> ONE all-gather per step
    PACK = B * K * 12
    packed = torch.empty(PACK, dtype=torch.uint8, device=dev)
    out_ids = packed[:B * K * 8].view(torch.int64).view(B, K)
    out_sc = packed[B * K * 8:].view(torch.float32).view(B, K)
    if use_dist:
        gathered = torch.empty(world * PACK, dtype=torch.uint8, device=dev)
        m_ids = torch.empty((B, K), dtype=torch.int64, device=dev)
        m_sc = torch.empty((B, K), dtype=torch.float32, device=dev)

    def step(i, batch=B, c=None):
        c = c or ctx
        stream = torch.cuda.current_stream().cuda_stream
        q = qdev[(i % nbatches) * B:(i % nbatches) * B + batch]
        c.search_batch_dev(q.data_ptr(), batch, K, out_ids.data_ptr(), out_sc.data_ptr(), stream)
        if use_dist:
            if share_gpu:
                cg = torch.empty(world * PACK, dtype=torch.uint8)
                dist.all_gather_into_tensor(cg, packed.cpu())
                gathered.copy_(cg)
            else:
                dist.all_gather_into_tensor(gathered, packed)   # RCCL over xGMI: B*k*(8+4) = 123 KB per rank, one collective
            c.merge_topk_strided_dev(gathered.data_ptr(), gathered.data_ptr() + B * K * 8, PACK, PACK, world, batch, K,
                                     m_ids.data_ptr(), m_sc.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(nsteps, first=0, batch=B, c=None):
        """barrier + sync, nsteps steps, barrier + sync; MAX over ranks.  The self-check of EVERY step is read afterwards
        (sticky flags of nvdb_hip_search_check; raises on overflow / bound violation)."""
        barrier()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(first + i, batch, c)
        barrier()
        el = time.perf_counter() - t0
        st = (c or ctx).search_check()
        if use_dist:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, st

    # ---- untimed: warm-up + parity self-check ---------------------------------------------------------
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    ctx.search_check()                                         # raises on overflow / bound violation in any warm-up step
    parity = "skipped"
    if rank == 0:
        # Self-check without the oracle (the oracle only serves the cpu_baseline leg here; tests/ pin the exact
        # fp32 kernel against it, and compare ALL queries at 10M and 100M rows): the MFMA filter path must reproduce
        # the exact fp32-order kernel bit for bit.  Local search only -- no collective in this branch.
        npar = 64 if world == 1 else 8
        strm = torch.cuda.current_stream().cuda_stream
        ctx.search_batch_dev(qdev[:B].data_ptr(), B, K, out_ids.data_ptr(), out_sc.data_ptr(), strm)
        torch.cuda.synchronize()
        fi, fs = out_ids[:npar].cpu().numpy().astype(np.uint64), out_sc[:npar].cpu().numpy()
        ctx.set_option("path", 1)
        ctx.search_batch_dev(qdev[:npar].data_ptr(), npar, K, out_ids.data_ptr(), out_sc.data_ptr(), strm)
        torch.cuda.synchronize()
        ei, es = out_ids[:npar].cpu().numpy().astype(np.uint64), out_sc[:npar].cpu().numpy()
        ctx.set_option("path", args.path)
        ctx.search_check()
        ok = np.array_equal(fi, ei) and np.array_equal(fs.view(np.uint32), es.view(np.uint32))
        parity = f"ok: MFMA path == exact fp32-order kernel (ids and score bits, {npar} queries)" if ok else "FAILED"
        if not ok:
            raise SystemExit("parity self-check failed: filter path != exact path")
    barrier()

    # ---- N > 1: the merged lists of one step against the unsharded corpus on rank 0 -------------------------
    merge_check = None
    one_gpu_ms = None
    if use_dist and not args.no_verify_merge:
        step(0)
        torch.cuda.synchronize()
        if rank == 0:
            try:
                full = nvdb_amd.HipContext(local_rank)
                full.generate_corpus(SEED, N, D, dt, row_base=0)       # 153.6 GB at N=100M: fits beside the rank's shard
                fi, fs = full.search_batch(qhost[:B], K)
                # the SAME corpus on ONE GPU, timed here so that this line carries its own 1-GPU point (3 passes after the warm one above,
                # device-resident like the timed region): strong_scaling_efficiency = value / (N x this)
                strm = torch.cuda.current_stream().cuda_stream
                o_i = torch.empty((B, K), dtype=torch.int64, device=dev)
                o_s = torch.empty((B, K), dtype=torch.float32, device=dev)
                full.search_batch_dev(qdev[:B].data_ptr(), B, K, o_i.data_ptr(), o_s.data_ptr(), strm)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(3):
                    full.search_batch_dev(qdev[(i % nbatches) * B:(i % nbatches + 1) * B].data_ptr(), B, K, o_i.data_ptr(), o_s.data_ptr(), strm)
                torch.cuda.synchronize()
                one_gpu_ms = (time.perf_counter() - t0) * 1e3 / 3
                full.search_check()
                full.close()
                mi, ms_ = m_ids.cpu().numpy().astype(np.uint64), m_sc.cpu().numpy()
                merge_check = bool(np.array_equal(mi, fi) and np.array_equal(ms_.view(np.uint32), fs.view(np.uint32)))
                if not merge_check:
                    bad = np.argwhere(mi != fi)
                    detail = "; ".join(f"q{a} #{j}: full ({fi[a, j]}, {fs[a, j]:.6f}) merged ({mi[a, j]}, {ms_[a, j]:.6f})" for a, j in bad[:6])
                    raise SystemExit(f"sharded + merged result differs from the unsharded search: {len(bad)} of {mi.size} ids; {detail}")
            except nvdb_amd.NvdbError as e:
                merge_check = f"skipped: {e}"
        barrier()

    # ---- timed region ---------------------------------------------------------------------------------
    ctx.set_option("time_kernels", 0 if os.environ.get("NVDB_BENCH_NO_KERNEL_EVENTS") == "1" else 1)   # diagnosis only: roofline becomes null
    elapsed, stats = timed_steps(args.steps, first=args.warmup)
    ctx.set_option("time_kernels", 0)
    kt = ctx.collect_kernel_times()

    qps = args.steps * B / elapsed
    ms_per_step = elapsed * 1e3 / args.steps

    # ---- N > 1: where a step's time goes (events on the bench stream around search / all-gather / merge; a few extra, untimed
    # steps after the timed region; per phase the MAX over ranks of the per-rank mean) ---------------------------------------
    step_split = None
    if use_dist and not share_gpu:
        nsp = max(2, min(args.steps, 8))
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(nsp)]
        barrier()
        for i in range(nsp):
            stream = torch.cuda.current_stream().cuda_stream
            q = qdev[(i % nbatches) * B:(i % nbatches + 1) * B]
            evs[i][0].record()
            ctx.search_batch_dev(q.data_ptr(), B, K, out_ids.data_ptr(), out_sc.data_ptr(), stream)
            evs[i][1].record()
            dist.all_gather_into_tensor(gathered, packed)
            evs[i][2].record()
            ctx.merge_topk_strided_dev(gathered.data_ptr(), gathered.data_ptr() + B * K * 8, PACK, PACK, world, B, K, m_ids.data_ptr(), m_sc.data_ptr(), stream)
            evs[i][3].record()
        barrier()
        ctx.search_check()
        mine = [sum(e[j].elapsed_time(e[j + 1]) for e in evs) / nsp for j in range(3)]
        tt = torch.tensor(mine, dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        sp = [float(x) for x in tt.tolist()]
        step_split = {"search_ms": sp[0], "allgather_ms": sp[1], "merge_ms": sp[2], "steps": nsp,
                      "note": "HIP events on the bench stream around the three phases of a step, extra untimed steps; per phase the MAX over "
                              "ranks of the per-rank mean.  allgather_ms includes waiting for the slowest rank's search (the collective "
                              "starts when every rank has reached it)"}
    out = {
        "metric": "QPS + effective HBM GB/s, flat-scan top-10 d=768",
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16" if args.dtype == "f16" else "i8", "data": "synthetic",
        "config": {"workload": f"{'fp16' if args.dtype == 'f16' else 'int8+scale'} flat-scan top-{K}, N={N} d={D}, batch={B}"
                               + (" (BASELINE configs[1])" if (N, world) == (ROWS_1GPU, 1) else " (BASELINE configs[3])" if N == ROWS_SHARDED else ""),
                   "rows_total": N, "rows_per_gpu": hi - lo, "batch": B, "k": K, "dim": D,
                   "parallelism": "1 GPU" if world == 1 else f"corpus row-sharded x{world}, RCCL all-gather of per-shard top-k",
                   "series": "strong scaling over the 100M-row corpus; its 1-GPU point is measured in THIS run (one_gpu_same_corpus: the unsharded corpus on rank 0's GPU) and also reported as extras.fp16_100M_batch1024 of the --gpus 1 line.  Do NOT divide this value by the --gpus 1 line's value: that one is the 10M-row corpus of configs[1]" if (world > 1 and N == ROWS_SHARDED) else None},
        "effective_hbm_GBps": (N * bpr / 1e9) / (ms_per_step * 1e-3),          # corpus bytes / pass time, all GPUs
        "parity": parity, "merge_check": merge_check,
        "exchange": (("gloo on host copies (NVDB_BENCH_SHARE_GPU rehearsal)" if share_gpu else f"RCCL all-gather of {PACK} packed bytes per rank, backend {dist.get_backend()}, world {world}")
                     + (" -- forced at world 1 (NVDB_BENCH_FORCE_COLLECTIVE)" if world == 1 else "")) if use_dist else None,
        "step_split_ms": step_split,
        "one_gpu_same_corpus": ({"qps": B / (one_gpu_ms * 1e-3), "ms_per_step": one_gpu_ms,
                                 "what": f"the UNSHARDED {N}-row corpus resident on rank 0's GPU alone (the context merge_check builds), same batch, "
                                         "device-resident queries and results, 3 timed passes -- this line's own 1-GPU point",
                                 "strong_scaling_efficiency": qps / (world * B / (one_gpu_ms * 1e-3)),
                                 "efficiency_is": "value / (n_gpus x one_gpu_same_corpus.qps)"} if one_gpu_ms else None),
        "self_check": "every timed step checked (sticky flags): no list overflow, no bound violation",
        "scan": {"path": stats["path"], "chunks": stats["chunks"], "candidates_per_query": stats["candidates"] / max(B, 1),
                 "bound_violations": stats["bound_violations"], "overflow_queries": stats["overflow_queries"]},
    }
    if kt["launches"]:
        sec = kt["ms"] * 1e-3
        ach = kt["flops"] / sec / 1e12
        peak = PEAK_F16_TFLOPS if args.dtype == "f16" else PEAK_I8_TOPS
        kname = (("filter_f16_m16_kernel<768> (8 waves x 32 queries)" if B > 128 else "filter_f16_kernel<768,1>") if args.dtype == "f16" else ("filter_i8s_kernel<768> (two-stage build on v_mfma_i32_16x16x64_i8)" if B > 128 else ("filter_i8s_kernel<768> on 8 waves of 32 queries" if B > 8 else "filter_i8w_kernel<768,1>")))
        gbps = kt["bytes"] / sec / 1e9
        # ridge point: intensity = 2*B*dim flop per row / row bytes  vs  peak flop / peak bytes
        hbm_bound = (2.0 * B * D / bpr) < (peak * 1e12 / (PEAK_HBM_GBPS * 1e9))
        traffic, traffic_source = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get(f"{'fp16' if args.dtype == 'f16' else 'int8'} N={N} d={D} batch={B}")
            if ent and world == 1:
                traffic = ent["bytes_per_launch"]
                traffic_source = ("profiles/traffic.json: FETCH_SIZE x2 (gfx950) + WRITE_SIZE of a separate rocprofv3 --pmc pass of this "
                                  f"command ({ent.get('source', 'see profiles/README.md')}); not measured in this run")
        except Exception:
            pass
        common = {"traffic": traffic, "traffic_source": traffic_source, "kernel": kname, "launches": kt["launches"],
                  "avg_launch_ms": kt["ms"] / kt["launches"], "kernel_time_share": kt["ms"] / (elapsed * 1e3), "mfma_T_per_s": ach,
                  "mfma_frac": ach / peak, "hbm_GBps_algorithmic": gbps, "hbm_frac": gbps / PEAK_HBM_GBPS}
        if hbm_bound:
            out["roofline"] = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS, **common}
        else:
            # int8: algorithmic ops stay 2*B*N*d whatever the kernel issues (hi plane always, lo plane on demand)
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s" if args.dtype == "f16" else "TOP/s",
                               "frac": ach / peak, **common}
    else:
        out["roofline"] = None

    # ---- the same step through the host API of the C ABI (PCIe-inclusive; never `value`) --------------------------
    if world == 1:
        reps = max(2, min(args.steps, 8))
        ctx.search_batch(qhost[:B], K)
        t0 = time.perf_counter()
        for i in range(reps):
            ctx.search_batch(qhost[(i % nbatches) * B:(i % nbatches + 1) * B], K)
        el = (time.perf_counter() - t0) / reps
        out["host_api"] = {"call": "nvdb_hip_search_batch: pageable host queries in (3.1 MB H2D), ids + scores out (123 KB D2H), self-check read back, one synchronisation per call",
                           "qps": B / el, "ms_per_step": el * 1e3, "overhead_vs_device_resident_ms": el * 1e3 - ms_per_step}

    # ---- N > 1: the 10M-row corpus of configs[1] sharded the same way (small shards: fixed costs weigh more) -----------
    if world > 1 and not args.no_extras:
        lo2, hi2 = shard_range(ROWS_1GPU, rank, world)
        ctx.generate_corpus(SEED, hi2 - lo2, D, dt, row_base=lo2)
        for i in range(2):
            step(i)
        el2, _ = timed_steps(args.steps)
        out["extras"] = {"strong_10M": {"workload": f"fp16 flat-scan top-{K}, N={ROWS_1GPU} d={D}, batch={B}, row-sharded x{world}",
                                        "rows_per_gpu": hi2 - lo2, "qps": args.steps * B / el2, "ms_per_step": el2 * 1e3 / args.steps}}

    # ---- optional batch sweep (HBM-bound points), N=1 only ---------------------------------------------
    if args.sweep and world == 1:
        sweep = []
        for b in [int(x) for x in args.sweep.split(",") if x]:
            b = min(b, B)
            for i in range(2):
                step(i, b)
            ctx.set_option("time_kernels", 1)
            reps = max(2, min(args.steps, 8))
            el, s2 = timed_steps(reps, batch=b)
            ctx.set_option("time_kernels", 0)
            k2 = ctx.collect_kernel_times()
            sweep.append({"batch": b, "qps": reps * b / el, "ms_per_pass": el * 1e3 / reps, "path": s2["path"],
                          "hbm_GBps": N * bpr / 1e9 / (el / reps), "hbm_frac": N * bpr / 1e9 / (el / reps) / PEAK_HBM_GBPS,
                          "tflops": 2.0 * b * N * D / (el / reps) / 1e12,
                          "filter_kernel_ms_per_pass": (k2["ms"] / reps) if k2["launches"] else None})
        out["sweep"] = sweep

    # ---- secondary measurements (N=1 only; the headline `value` above is unaffected) ------------------------
    if rank == 0 and world == 1 and not args.no_extras and args.dtype == "f16":
        extras = {}
        try:
            def timed_passes(c, b, reps=6):
                for i in range(2):
                    step(i, b, c)
                el, st_ = timed_steps(reps, batch=b, c=c)
                timed_passes.last_stats = st_
                return el / reps
            # (1) HBM-bound point of the same fp16 corpus: batch 64
            el = timed_passes(ctx, 64)
            extras["fp16_batch64"] = {"workload": f"fp16 flat-scan top-{K}, N={N} d={D}, batch=64", "qps": 64 / el, "ms_per_pass": el * 1e3,
                                      "hbm_GBps": N * D * 2 / 1e9 / el, "hbm_frac": N * D * 2 / 1e9 / el / PEAK_HBM_GBPS}
            # (1b) SURVEY 8(d) batch sweep, both fractions at every point (B = 64 and 1024 are the lines above / the headline)
            def sweep_points(c, row_bytes, peak_ops):
                pts = []
                for bb in (1, 8, 128, 256, 512):
                    el_ = timed_passes(c, bb, reps=4)
                    pts.append({"batch": bb, "qps": bb / el_, "ms_per_pass": el_ * 1e3, "path": timed_passes.last_stats["path"],
                                "hbm_GBps": N * row_bytes / 1e9 / el_, "hbm_frac": N * row_bytes / 1e9 / el_ / PEAK_HBM_GBPS,
                                "mfma_T_per_s": 2.0 * bb * N * D / el_ / 1e12, "mfma_frac": 2.0 * bb * N * D / el_ / 1e12 / peak_ops,
                                "bound": "hbm" if (2.0 * bb * D / row_bytes) < (peak_ops * 1e12 / (PEAK_HBM_GBPS * 1e9)) else "mfma"})
                return pts
            extras["sweep"] = {"note": "whole-pass wall time per point (bootstrap, selects and rescore included), 4 timed passes each; "
                                       "fractions against 8 TB/s and the dense MFMA peak of the dtype",
                               "fp16": sweep_points(ctx, D * 2, PEAK_F16_TFLOPS)}
            # (1c) the exact fp32-order path (fallback / any-k / this script's own parity check): 64 queries against the whole corpus
            ctx.set_option("path", 1)
            ex = {}
            for mf in (1, 0):
                ctx.set_option("exact_mfma", mf)
                el_ = timed_passes(ctx, 64, reps=2)
                ex["fp32_mfma" if mf else "valu"] = {"ms_per_pass": el_ * 1e3, "TFLOPs": 2.0 * 64 * N * D / el_ / 1e12}
            ctx.set_option("exact_mfma", 1)
            el_ = timed_passes(ctx, 256, reps=2)
            ex["fp32_mfma_256_queries"] = {"ms_per_pass": el_ * 1e3, "TFLOPs": 2.0 * 256 * N * D / el_ / 1e12}
            ctx.set_option("path", args.path)
            extras["exact_path_batch64"] = {"workload": f"exact fp32-order scan (path 1), fp16 corpus N={N} d={D}, 64 queries (fp32_mfma_256_queries: 256)", **ex,
                                            "kernels": "exact_mfma_img_kernel (v_mfma_f32_16x16x4_f32, eight accumulator tiles = the reference's eight fma chains; rows converted "
                                                       "once per workgroup into an fp32 LDS image) vs scan_exact_kernel (VALU)",
                                            "fp32_matrix_peak_TFLOPs": 157.3}
            # (2) BASELINE configs[2]: int8(+scale), same shape
            c8 = nvdb_amd.HipContext(local_rank)
            c8.generate_corpus(SEED, N, D, nvdb_amd.DT_I8)
            for bb in (B, 64):
                el = timed_passes(c8, bb)
                extras[f"int8_batch{bb}"] = {"workload": f"int8+scale flat-scan top-{K}, N={N} d={D}, batch={bb}", "qps": bb / el, "ms_per_pass": el * 1e3,
                                             "hbm_GBps": N * (D + 4) / 1e9 / el, "hbm_frac": N * (D + 4) / 1e9 / el / PEAK_HBM_GBPS,
                                             "algorithmic_TOPs": 2.0 * bb * N * D / el / 1e12,
                                             "two_stage": {"tiles_past_quick_test": timed_passes.last_stats.get("i8_stage1_tiles"),
                                                           "lo_plane_blocks": timed_passes.last_stats.get("i8_stage2_blocks"),
                                                           "wave_tiles": (N // 64) * ((bb + 63) // 64 if bb > 128 else (bb + 31) // 32)}}
            extras["sweep"]["int8"] = sweep_points(c8, D + 4, PEAK_I8_TOPS)
            c8.close()
            # recall@10 and score deltas of the int8 corpus against the fp32 corpus at configs[2]'s OWN size: all N rows, all B queries.
            # Same synthetic rows (seed, row id), quantised per row as the reference's tool does (apps/nvdb_quantize_i8.cpp:71-80:
            # scale = max|x| / 127, lrint(x / scale), clamp to [-127, 127]).  The fp32 corpus (30.7 GB + its fp16 filter shadow) is
            # searched on the filter path: exact fp32-order scores of the fp32 rows.
            c32 = nvdb_amd.HipContext(local_rank)
            c32.generate_corpus(SEED, N, D, nvdb_amd.DT_F32)
            g_ids, g_sc = c32.search_batch(qhost[:B], K)
            c32.close()
            c8s = nvdb_amd.HipContext(local_rank)
            c8s.generate_corpus(SEED, N, D, nvdb_amd.DT_I8)
            i_ids, i_sc = c8s.search_batch(qhost[:B], K)
            c8s.close()
            extras["int8_recall_at_10_vs_fp32"] = float(np.mean([len(set(a.tolist()) & set(b_.tolist())) / K for a, b_ in zip(g_ids, i_ids)]))
            same = i_ids[:, :, None] == g_ids[:, None, :]                               # [q][j int8][j' fp32]: the same row in both lists
            d_same = np.abs(i_sc[:, :, None].astype(np.float64) - g_sc[:, None, :].astype(np.float64))[same]
            d_rank = np.abs(i_sc.astype(np.float64) - g_sc.astype(np.float64))
            extras["int8_vs_fp32"] = {"workload": f"int8+scale vs fp32 flat-scan top-{K}, N={N} d={D}, {B} queries (BASELINE configs[2]: 'recall vs fp32 reported')",
                                      "recall_at_k": extras["int8_recall_at_10_vs_fp32"], "k": K, "queries": B, "rows": N,
                                      "queries_with_identical_id_sets": int(sum(set(a.tolist()) == set(b_.tolist()) for a, b_ in zip(g_ids, i_ids))),
                                      "top1_agrees": float(np.mean(i_ids[:, 0] == g_ids[:, 0])),
                                      "score_delta_same_row": {"max": float(d_same.max()), "mean": float(d_same.mean()), "pairs": int(d_same.size),
                                                               "what": "|score_int8 - score_fp32| of every row that is in both returned lists"},
                                      "score_delta_rankwise": {"max": float(d_rank.max()), "mean": float(d_rank.mean()),
                                                               "what": "|j-th best int8 score - j-th best fp32 score| over all queries and ranks"},
                                      "quantiser": "per-row scale = max|x|/127, lrint, clamp [-127,127] (reference apps/nvdb_quantize_i8.cpp:71-80)"}
            # (2b) NOT a BASELINE config and never `value`: the same fp16 corpus FILTERED through an int8 shadow of itself (option q8_shadow, off by
            #      default: + N x d bytes of HBM) -- integer MFMAs over half the bytes, survivors re-scored from the fp16 rows, same ids and score bits
            cq = nvdb_amd.HipContext(local_rank)
            cq.set_option("q8_shadow", 1)
            cq.generate_corpus(SEED, N, D, nvdb_amd.DT_F16)
            shadow = {}
            for bb in (B, 64):
                el = timed_passes(cq, bb)
                shadow[f"batch{bb}"] = {"qps": bb / el, "ms_per_pass": el * 1e3, "candidates_per_query": timed_passes.last_stats["candidates"] / bb,
                                        "hbm_GBps_shadow_bytes": N * (D + 4) / 1e9 / el}
            strm = torch.cuda.current_stream().cuda_stream
            cq.search_batch_dev(qdev[:B].data_ptr(), B, K, out_ids.data_ptr(), out_sc.data_ptr(), strm)
            torch.cuda.synchronize(); cq.search_check()
            qi_, qs_ = out_ids.cpu().numpy().copy(), out_sc.cpu().numpy().copy()
            ctx.search_batch_dev(qdev[:B].data_ptr(), B, K, out_ids.data_ptr(), out_sc.data_ptr(), strm)
            torch.cuda.synchronize(); ctx.search_check()
            same = bool(np.array_equal(qi_, out_ids.cpu().numpy()) and np.array_equal(qs_.view(np.uint32), out_sc.cpu().numpy().view(np.uint32)))
            cq.close()
            extras["fp16_corpus_int8_filter_shadow"] = {"workload": f"fp16 flat-scan top-{K}, N={N} d={D}: filter on an int8 shadow of the fp16 rows (option q8_shadow=1, off by default), exact rescore from the fp16 rows",
                                                        **shadow, "identical_to_the_fp16_filter": same, "extra_hbm_bytes": N * (D + 4)}
            if not same:
                raise RuntimeError("q8_shadow: results differ from the fp16 filter path")
            # (3) BASELINE configs[4]: exact-L2 refine, N=2.9M fp16, Q=10000, R=1024, K=10, synthetic candidates
            NR, QR, RR = min(N, 2_900_000), 10_000, 1024
            cr = nvdb_amd.HipContext(local_rank)
            cr.generate_corpus(SEED, NR, D, nvdb_amd.DT_F16)
            rq = nvdb_amd.synth_rows_f32(SEED + 2, 0, QR, D)
            rs_ = np.random.RandomState(1)
            cand = rs_.randint(0, NR, size=(QR, RR)).astype(np.uint32)
            cand[rs_.rand(QR, RR) < 0.01] = 0xFFFFFFFF
            best = None
            for _ in range(3):
                _, _, t = cr.refine_l2_topk(rq, cand, K, want_timing=True)
                best = t.kernel_ms if best is None else min(best, t.kernel_ms)
            cr.set_option("refine_pinned", 1)                  # the reference's CUDA_PINNED=1: pinned host staging of the call's buffers
            _, _, tp = cr.refine_l2_topk(rq, cand, K, want_timing=True)
            _, _, tp = cr.refine_l2_topk(rq, cand, K, want_timing=True)
            cr.close()
            gb = float((cand != 0xFFFFFFFF).sum()) * D * 2 / 1e9
            extras["refine"] = {"workload": f"exact-L2 refine N={NR} Q={QR} R={RR} K={K} fp16", "kernel": "refine_l2_rows_kernel<768>",
                                "kernel_ms": best, "us_per_query": best * 1e3 / QR, "h2d_ms": t.h2d_ms, "d2h_ms": t.d2h_ms,
                                "pinned": {"h2d_ms": tp.h2d_ms, "kernel_ms": tp.kernel_ms, "d2h_ms": tp.d2h_ms, "total_ms": tp.total_ms},
                                "gather_GBps": gb / (best * 1e-3), "hbm_frac": gb / (best * 1e-3) / PEAK_HBM_GBPS}
            del cand, rq
            # (3b) the shape the reference publishes for its CUDA refine (Performance_CUDA.md:54: RTX 3080, fp16 base 500K x 384, R=500,
            #      K=10, ids only: 29.86 ms per 10 000 queries, H2D 1.45 / kernel 28.39 / D2H 0.02) -- other hardware, orientation only
            cp = nvdb_amd.HipContext(local_rank)
            cp.generate_corpus(SEED, 500_000, 384, nvdb_amd.DT_F16)
            pq = nvdb_amd.synth_rows_f32(SEED + 3, 0, 10_000, 384)
            pc = np.random.RandomState(2).randint(0, 500_000, size=(10_000, 500)).astype(np.uint32)
            bt = None
            for _ in range(3):
                _, _, t2 = cp.refine_l2_topk(pq, pc, K, want_dist=False, want_timing=True)
                bt = t2 if bt is None or t2.kernel_ms < bt.kernel_ms else bt
            cp.close()
            extras["refine_published_shape"] = {"workload": "exact-L2 refine N=500000 d=384 fp16, Q=10000 R=500 K=10, ids only", "kernel": "refine_l2_rows_kernel<384>",
                                                "h2d_ms": bt.h2d_ms, "kernel_ms": bt.kernel_ms, "d2h_ms": bt.d2h_ms, "total_ms": bt.total_ms,
                                                "us_per_query": bt.total_ms * 1e3 / 10_000,
                                                "reference_published": "29.86 ms total (kernel 28.39) on an RTX 3080, Performance_CUDA.md:54"}
            del pq, pc
            # (4) the north star's target point: N=100M (153.6 GB resident on this one GPU), batch 64 (HBM-bound) and 1024
            c100 = nvdb_amd.HipContext(local_rank)
            c100.generate_corpus(SEED, ROWS_SHARDED, D, nvdb_amd.DT_F16)
            for bb, reps in ((64, 4), (B, 3)):
                el = timed_passes(c100, bb, reps)
                extras[f"fp16_100M_batch{bb}"] = {"workload": f"fp16 flat-scan top-{K}, N={ROWS_SHARDED} d={D}, batch={bb}, one GPU", "qps": bb / el,
                                                  "ms_per_pass": el * 1e3, "hbm_GBps": ROWS_SHARDED * D * 2 / 1e9 / el,
                                                  "hbm_frac": ROWS_SHARDED * D * 2 / 1e9 / el / PEAK_HBM_GBPS,
                                                  "mfma_TFLOPs": 2.0 * bb * ROWS_SHARDED * D / el / 1e12,
                                                  "mfma_frac": 2.0 * bb * ROWS_SHARDED * D / el / 1e12 / PEAK_F16_TFLOPS}
            c100.close()
        except Exception as e:
            extras["error"] = repr(e)
        out["extras"] = extras

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(ctx, args, nvdb_amd, N)
        except Exception as e:                                   # never let the baseline leg kill the bench line
            out["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": host_cpu_info()["threads_used"], "kind": "error",
                                   "sample": repr(e)}
        cli = out["cpu_baseline"].pop("_cli", None) if isinstance(out.get("cpu_baseline"), dict) else None
        if cli is not None:
            out.setdefault("extras", {}).update(cli)
            nb, ha = cli.get("nvdb_bench_cli"), out.get("host_api")
            if nb and ha:
                nb["QPS_over_host_api"] = nb["QPS"] / ha["qps"]
        ctx.close()
        try:
            out["cpu_baseline"]["config0_fp32_500K_st"] = config0_plumbing(nvdb_amd, args)
        except Exception as e:
            out["cpu_baseline"]["config0_fp32_500K_st"] = {"error": repr(e)}
    if rank == 0:
        # One point of each strong-scaling series per line, so the 1/2/4/8 curves can be read off the per-N lines without
        # mixing corpora: `value` is the 10M-row point at N = 1 and the 100M-row point at N > 1 (BASELINE configs[1] / [3]).
        ex = out.get("extras") or {}
        if args.dtype == "f16" and B == 1024:
            if world == 1 and N == ROWS_1GPU:
                out["scaling_points"] = {"fp16_10M_batch1024_qps": qps, "fp16_100M_batch1024_qps": (ex.get("fp16_100M_batch1024") or {}).get("qps")}
            elif world > 1 and N == ROWS_SHARDED:
                out["scaling_points"] = {"fp16_100M_batch1024_qps": qps, "fp16_10M_batch1024_qps": (ex.get("strong_10M") or {}).get("qps")}
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
