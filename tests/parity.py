"""Tie-aware comparison of top-k results (SURVEY.md section 4 / 8d "parity gate").

The reference's own variants (st / omp / async / pool) return different ids inside groups of
exactly equal scores, so "bit-exact ids" is defined as:
  * the score sequences are bit-identical position by position, and
  * for every score value strictly greater than the k-th score the id SETS are equal, and
  * ids in the last (boundary) score group all carry exactly that score (`score_of(id)`).
"""
import numpy as np


def assert_topk_equal(ids, scores, ref_ids, ref_scores, score_of=None, what=""):
    ids, ref_ids = np.asarray(ids).astype(np.int64), np.asarray(ref_ids).astype(np.int64)
    s, rs = np.asarray(scores, dtype=np.float32), np.asarray(ref_scores, dtype=np.float32)
    assert ids.shape == ref_ids.shape, f"{what}: shape {ids.shape} vs {ref_ids.shape}"
    assert np.array_equal(s.view(np.uint32), rs.view(np.uint32)), \
        f"{what}: score bits differ\n got {s}\n ref {rs}"
    if ids.size == 0:
        return
    last = rs[-1]
    for v in np.unique(rs):
        mine, theirs = set(ids[s == v].tolist()), set(ref_ids[rs == v].tolist())
        if v > last:
            assert mine == theirs, f"{what}: ids differ inside score group {v}: {sorted(mine)} vs {sorted(theirs)}"
        elif mine != theirs:
            assert score_of is not None, f"{what}: boundary tie group differs and no score_of() given"
            for i in mine:
                got = np.float32(score_of(i))
                assert got.view(np.uint32) == np.float32(v).view(np.uint32), \
                    f"{what}: boundary id {i} has score {got}, expected {v}"
    assert len(set(ids.tolist())) == ids.size, f"{what}: duplicate ids in result"
