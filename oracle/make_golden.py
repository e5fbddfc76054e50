#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REAL REFERENCE (oracle/_ref, built from
/root/reference by oracle/Makefile).  Run in the build container only:

    make -C oracle && python oracle/make_golden.py

Inputs are re-creatable from seeds (numpy legacy RandomState, elementwise float32 ops only), so
the fixtures hold just the seeds, a sha256 of every input array and the reference's OUTPUTS
(ids, score bit patterns, converter bytes).  tests/golden_inputs.py re-creates the inputs and
verifies the hashes before any comparison.
"""
import hashlib
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import pyoracle as po  # noqa: E402
from golden_inputs import (CASES, DOT_DIMS, make_case_inputs, make_dot_inputs, make_f16_specials,  # noqa: E402
                           sha)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = po.Reference()
    tmp = tempfile.mkdtemp(prefix="nvdb_golden_")
    g = {}

    # ---- 1. raw dot kernels (a1-a3), SIMD path and forced-scalar path
    for d in DOT_DIMS:
        q, x32, x16, x8, sc = make_dot_inputs(d)
        g[f"dot{d}_sha"] = np.frombuffer(bytes.fromhex(sha(q, x32, x16, x8, sc)), dtype=np.uint8)
        for tag, force in (("simd", 0), ("scalar", 1)):
            ref.lib.ref_set_force_scalar(force)
            f32 = np.array([ref.lib.ref_dot_f32(po._p(q[i], po._f32p), po._p(x32[i], po._f32p), d) for i in range(len(q))], dtype=np.float32)
            f16 = np.array([ref.lib.ref_dot_f32_f16base(po._p(q[i], po._f32p), x16[i].ctypes.data, d) for i in range(len(q))], dtype=np.float32)
            i8 = np.array([ref.lib.ref_dot_f32_i8base(po._p(q[i], po._f32p), x8[i].ctypes.data, d, float(sc[i])) for i in range(len(q))], dtype=np.float32)
            g[f"dot{d}_{tag}_f32"], g[f"dot{d}_{tag}_f16"], g[f"dot{d}_{tag}_i8"] = bits(f32), bits(f16), bits(i8)
        ref.lib.ref_set_force_scalar(0)

    # ---- 2. converters on special values (tools/nvdb_convert_f16.cpp) : one row of dim=len
    sp = make_f16_specials()
    p_in, p_out = os.path.join(tmp, "sp32.vecbin"), os.path.join(tmp, "sp16.vecbin")
    for name, arr in (("a", sp), ("b", sp[: (len(sp) // 8) * 8 - 3])):  # b: dim%8 != 0 -> scalar tail
        po.write_vecbin(p_in, arr.reshape(1, -1), po.DT_F32)
        ref.run_tool("nvdb_convert_f16", p_in, p_out)
        g[f"f16_specials_{name}"] = po.read_vecbin(p_out)[0].reshape(-1).copy()

    # ---- 3. flat-scan cases through the reference's own tools and classes
    for name, spec in CASES.items():
        base32, queries = make_case_inputs(name)
        n, d = base32.shape
        k = spec["k"]
        g[f"{name}_sha"] = np.frombuffer(bytes.fromhex(sha(base32, queries)), dtype=np.uint8)
        pb32 = os.path.join(tmp, f"{name}_f32.vecbin")
        pb16 = os.path.join(tmp, f"{name}_f16.vecbin")
        pb8 = os.path.join(tmp, f"{name}_i8.vecbin")
        pq = os.path.join(tmp, f"{name}_q.raw12")
        po.write_vecbin(pb32, base32, po.DT_F32)
        po.write_raw12(pq, queries)                       # queries are raw12 f32, like nvdb_make_query writes
        ref.run_tool("nvdb_convert_f16", pb32, pb16)      # defines the f16 corpus bits
        ref.run_tool("nvdb_quantize_i8", pb32, pb8)       # defines the int8 corpus bits + scales
        b16 = po.read_vecbin(pb16)[0]
        b8, _, sc8 = po.read_vecbin(pb8)
        g[f"{name}_f16_sha"] = np.frombuffer(bytes.fromhex(sha(b16)), dtype=np.uint8)
        g[f"{name}_i8_sha"] = np.frombuffer(bytes.fromhex(sha(b8, sc8)), dtype=np.uint8)
        g[f"{name}_f16_head"] = b16[:4].copy()
        g[f"{name}_i8_head"] = b8[:4].copy()
        g[f"{name}_i8_scales_head"] = sc8[:16].copy()
        for tag, path in (("f32", pb32), ("f16", pb16), ("i8", pb8)):
            h = ref.open(path)
            for mode, mname, thr in ((0, "st", 0), (1, "omp", 3)):
                ids, sc, _ = ref.flat_search(h, queries, k, mode=mode, threads=thr)
                g[f"{name}_{tag}_{mname}_ids"] = ids
                g[f"{name}_{tag}_{mname}_scores"] = bits(sc)
            ref.close(h)
        if spec.get("gtbin"):
            pg = os.path.join(tmp, f"{name}.gtbin")
            ref.run_tool("nvdb_gt_build", pb16, pq, k, pg, env={"GT_MODE": "st", "WARMUP": "0"})
            gt, meta = po.read_gtbin(pg)
            g[f"{name}_gtbin_f16_ids"] = gt
            g[f"{name}_gtbin_raw"] = np.fromfile(pg, dtype=np.uint8)[:64]
            txt = ref.run_tool("nvdb_search", pb32, pq, k)
            g[f"{name}_search_stdout"] = np.frombuffer(txt.encode(), dtype=np.uint8)

    np.savez_compressed(os.path.join(OUT, "flat_golden.npz"), **g)
    sz = os.path.getsize(os.path.join(OUT, "flat_golden.npz"))
    print(f"wrote tests/golden/flat_golden.npz: {len(g)} arrays, {sz} bytes")


CONV_ROWS = (0, 1, 2, 1499, 2999)            # rows of case "main768" whose float32 widening is kept


def main_refine_conv():
    """tests/golden/refine_conv_golden.npz: the row conversion under the reference's CPU refine --
    nvdb::f16_to_f32_scalar (include/nvdb/f16_scalar.h:8-38) on ALL 65 536 half bit patterns, and
    nvdb::base_row_to_f32 (include/nvdb/to_f32_row.h:10-34) on rows of the "main768" case in each base dtype
    (the f16 / int8 files are written by the reference's own converter tools, as in main())."""
    os.makedirs(OUT, exist_ok=True)
    ref = po.Reference()
    tmp = tempfile.mkdtemp(prefix="nvdb_golden_")
    g = {}
    allh, allf = np.arange(65536, dtype=np.uint16), np.empty(65536, dtype=np.float32)
    ref.lib.ref_f16_to_f32_array(allh.ctypes.data, 65536, po._p(allf, po._f32p))      # written to memory: NaN payload bits intact
    g["f16_to_f32_all"] = allf.view(np.uint32).copy()
    base32, _ = make_case_inputs("main768")
    d = base32.shape[1]
    pb32, pb16, pb8 = (os.path.join(tmp, f"main768_{t}.vecbin") for t in ("f32", "f16", "i8"))
    po.write_vecbin(pb32, base32, po.DT_F32)
    ref.run_tool("nvdb_convert_f16", pb32, pb16)
    ref.run_tool("nvdb_quantize_i8", pb32, pb8)
    g["main768_f16_sha"] = np.frombuffer(bytes.fromhex(sha(po.read_vecbin(pb16)[0])), dtype=np.uint8)
    b8, _, sc8 = po.read_vecbin(pb8)
    g["main768_i8_sha"] = np.frombuffer(bytes.fromhex(sha(b8, sc8)), dtype=np.uint8)
    g["rows"] = np.array(CONV_ROWS, dtype=np.uint64)
    for tag, path in (("f32", pb32), ("f16", pb16), ("i8", pb8)):
        h = ref.open(path)
        g[f"main768_{tag}_rows_f32"] = bits(np.stack([ref.base_row_to_f32(h, r, d) for r in CONV_ROWS]))
        ref.close(h)
    np.savez_compressed(os.path.join(OUT, "refine_conv_golden.npz"), **g)
    sz = os.path.getsize(os.path.join(OUT, "refine_conv_golden.npz"))
    print(f"wrote tests/golden/refine_conv_golden.npz: {len(g)} arrays, {sz} bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "refine_conv":
        main_refine_conv()
    else:
        main()
        main_refine_conv()
