"""bench.py's own launcher, on the CPU: `python bench.py --gpus N` with no WORLD_SIZE in the environment must start
its N ranks itself (from a parent that never touches the GPU) and the ranks must rendezvous; `--dry-launch` makes them
meet over gloo, all-gather a result-shaped tensor, merge it with the product's merge and exit.  The same file under
torch.distributed.run (the driver's way) must take the ranks it is given."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE", "NVDB_BENCH_SELF_LAUNCHED")}
    e["OMP_NUM_THREADS"] = "1"
    return e


def test_bench_spawns_its_own_ranks():
    for n in (2, 3, 8):                                # 8: what the driver's scaling run starts (N = 1, 2, 4, 8)
        r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-launch"], capture_output=True, text=True, env=_env(), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out == {"dry_launch": True, "world": n, "merge_ok": True, "rows_total": 100_000_000, "rows_per_gpu": 100_000_000 // n, "launcher": "self"}


def test_bench_under_torch_distributed_run():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert out["world"] == 2 and out["merge_ok"] and out["launcher"] == "external" and out["rows_per_gpu"] == 50_000_000


def test_bench_parent_does_not_touch_the_gpu():
    """The spawning parent must not import torch or load the HIP library before it forks the ranks."""
    src = open(BENCH).read()
    main_src = src[src.index("def main():"):]
    head = main_src[:main_src.index("sys.exit(launch_ranks(args.gpus))")]
    assert "import torch" not in head and "import nvdb_amd" not in head
    launch = src[src.index("def launch_ranks"):src.index("def dry_launch")]
    assert "import torch" not in launch and "import nvdb_amd" not in launch and "load_library" not in launch
    top = src[:src.index("def parse():")]
    assert "import torch" not in top and "import nvdb_amd" not in top                  # module level: numpy and the standard library only


def test_one_gpu_default_is_config_1():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-launch"], capture_output=True, text=True, env=dict(_env(), MASTER_PORT="29533"), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["world"] == 1 and out["rows_total"] == 10_000_000
