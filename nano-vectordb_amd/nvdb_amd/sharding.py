"""Row-sharding of the corpus across ranks and the one exchange step of the multi-GPU flat scan.

shard r of W holds rows [r*N//W, (r+1)*N//W); ids it returns are global (base + local row).
exchange = all-gather of the per-shard [nq][k] ids and scores (RCCL on GPUs, gloo in the CPU tests),
then a k-way merge in the product's canonical (score desc, id asc) order.  Because top-k of a union is
the top-k of the per-part top-k lists and the order relation is total, the merged result is
bit-identical to the unsharded search.
"""
import numpy as np


def shard_range(n_rows, rank, world):
    return n_rows * rank // world, n_rows * (rank + 1) // world


def all_gather_topk(dist, ids_t, scores_t, world):
    """ids_t: int64 [nq,k] tensor, scores_t: float32 [nq,k] tensor (any device) -> ([W,nq,k],[W,nq,k])."""
    import torch
    nq, k = ids_t.shape
    g_ids = torch.empty((world * nq, k), dtype=ids_t.dtype, device=ids_t.device)     # concatenated along dim 0:
    g_sc = torch.empty((world * nq, k), dtype=scores_t.dtype, device=scores_t.device)  # the form gloo and RCCL both take
    dist.all_gather_into_tensor(g_ids, ids_t.contiguous())
    dist.all_gather_into_tensor(g_sc, scores_t.contiguous())
    return g_ids.view(world, nq, k), g_sc.view(world, nq, k)


def merge_host(g_ids, g_sc):
    """CPU merge through the C ABI's nvdb_merge_topk_host."""
    from . import merge_topk_host
    return merge_topk_host(np.ascontiguousarray(g_ids).view(np.uint64), np.ascontiguousarray(g_sc))
