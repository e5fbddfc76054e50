"""N>1 path on CPU: W processes (gloo, W = 2, 3, 8), each owns one row shard; per-shard top-k (from the oracle here, the
GPU kernel on the box) -> all-gather -> merge with the product's merge -> must equal the unsharded answer.  Two exchanges per
run: the two-tensor all-gather of nvdb_amd.sharding and bench.py's ONE packed uint8 buffer [ids | scores] per rank, unpacked with
the strides nvdb_hip_merge_topk_strided_dev takes.  World 8 runs k = 64 and k = 600: the latter is longer than a shard
(padded lists: id ~0, -inf) and nshards * k = 4800 > 4096, the size beyond which the device merge switches kernels."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, nq, k, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "nano-vectordb_amd")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import nvdb_amd
    from nvdb_amd import sharding
    import pyoracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = po.Oracle()
    lo, hi = sharding.shard_range(n, rank, world)
    base, _ = nvdb_amd.synth_corpus(77, lo, hi - lo, d, nvdb_amd.DT_F16)          # rows keyed by GLOBAL row id
    queries = nvdb_amd.synth_rows_f32(78, 0, nq, d)
    ids, sc = orc.flat_topk(base, po.DT_F16, queries, k)                           # min(k, shard rows) entries per query
    ids = ids + np.uint64(lo)                                                      # global ids = base + local
    if ids.shape[1] < k:                                                           # a shard shorter than k pads its lists, as the search does
        padn = k - ids.shape[1]
        ids = np.concatenate([ids, np.full((nq, padn), np.iinfo(np.uint64).max, np.uint64)], axis=1)
        sc = np.concatenate([sc, np.full((nq, padn), -np.inf, np.float32)], axis=1)
    g_ids, g_sc = sharding.all_gather_topk(dist, torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(sc), world)
    m_ids, m_sc = sharding.merge_host(g_ids.numpy(), g_sc.numpy())
    # bench.py's exchange: ONE packed buffer per rank, [ids: nq*k*8 bytes | scores: nq*k*4 bytes], one all-gather, block r at r * PACK
    PACK = nq * k * 12
    packed = torch.from_numpy(np.concatenate([np.ascontiguousarray(ids).view(np.uint8).ravel(), np.ascontiguousarray(sc).view(np.uint8).ravel()]))
    gathered = torch.empty(world * PACK, dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, packed)
    gb = gathered.numpy()
    p_ids = np.stack([gb[r * PACK:r * PACK + nq * k * 8].view(np.uint64).reshape(nq, k) for r in range(world)])
    p_sc = np.stack([gb[r * PACK + nq * k * 8:(r + 1) * PACK].view(np.float32).reshape(nq, k) for r in range(world)])
    pm_ids, pm_sc = nvdb_amd.merge_topk_host(p_ids, p_sc)
    assert np.array_equal(pm_ids, m_ids) and np.array_equal(pm_sc.view(np.uint32), m_sc.view(np.uint32)), "packed exchange != two-tensor exchange"
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ids=m_ids, sc=m_sc)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,k", [(2, 10), (3, 10), (8, 64), (8, 600)])
def test_gloo_shard_exchange_merge(tmp_path, oracle, world, k):
    import torch.multiprocessing as mp
    import nvdb_amd
    import pyoracle as po
    n, d, nq = 3001, 128, 6
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, d, nq, k, str(tmp_path)), nprocs=world, join=True)
    base, _ = nvdb_amd.synth_corpus(77, 0, n, d, nvdb_amd.DT_F16)
    queries = nvdb_amd.synth_rows_f32(78, 0, nq, d)
    ref_ids, ref_sc = oracle.flat_topk(base, po.DT_F16, queries, k)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["ids"], ref_ids), f"rank {r}"
        assert np.array_equal(got["sc"].view(np.uint32), ref_sc.view(np.uint32)), f"rank {r}"


def test_shard_ranges_partition_the_corpus():
    from nvdb_amd.sharding import shard_range
    for n in (1, 7, 10_000_000, 100_000_003):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1
