"""ctypes front-end for the TEST-ONLY checkers in oracle/ (see nvdb_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
`Oracle`    -> oracle/liboracle.so        (our C restatement; always available after `make -C oracle`)
`Reference` -> oracle/_ref/libnvdb_ref.so (the real reference compiled from /root/reference;
                                           present only where that tree existed at build time)
Also holds small numpy helpers for the reference's on-disk formats (vecbin64 / raw12 / gtbin:
include/nvdb/vecbin_format.h:7-59, src/vector_dataset.cpp:11-22, include/nvdb/gtbin_format.h:7-35)
so that tests can write inputs the reference binaries accept.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
VEC_MAGIC = 0x4E56444256454331
GT_MAGIC = 0x4E56444247543031
DT_F32, DT_F16, DT_I8 = 1, 2, 3

_f32p = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def build(force=False):
    """Compile liboracle.so (and _ref/ when /root/reference is present)."""
    if force or not os.path.exists(os.path.join(HERE, "liboracle.so")):
        subprocess.check_call(["make", "-C", HERE, "-j4"], stdout=subprocess.DEVNULL)
    else:
        subprocess.check_call(["make", "-C", HERE, "-j4"], stdout=subprocess.DEVNULL)


# ---------------------------------------------------------------------------- file formats
def write_vecbin(path, rows, dtype, scales=None):
    """rows: [N, D] array of float32 / uint16 (half bits) / int8."""
    n, d = rows.shape
    hdr = struct.pack("<QIIIIQ", VEC_MAGIC, 1, dtype, d, 0, n) + b"\0" * 32
    assert len(hdr) == 64
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(np.ascontiguousarray(rows).tobytes())
        if dtype == DT_I8:
            f.write(np.ascontiguousarray(scales, dtype=np.float32).tobytes())


def write_raw12(path, rows_f32):
    n, d = rows_f32.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<III", n, 0, d))
        f.write(np.ascontiguousarray(rows_f32, dtype=np.float32).tobytes())


def read_vecbin(path):
    raw = np.fromfile(path, dtype=np.uint8)
    magic, ver, dtype, dim, _, count = struct.unpack("<QIIIIQ", raw[:32].tobytes())
    assert magic == VEC_MAGIC and ver == 1
    body = raw[64:]
    if dtype == DT_F32:
        return body.view(np.float32).reshape(count, dim), dtype, None
    if dtype == DT_F16:
        return body.view(np.uint16).reshape(count, dim), dtype, None
    rows = body[: count * dim].view(np.int8).reshape(count, dim)
    return rows, dtype, body[count * dim:].view(np.float32).copy()


def read_gtbin(path):
    raw = np.fromfile(path, dtype=np.uint8)
    magic, ver, metric, k, dim, q, n = struct.unpack("<QIIIIQQ", raw[:40].tobytes())
    assert magic == GT_MAGIC and ver == 1
    return raw[64:].view(np.uint32).reshape(q, k), dict(metric=metric, k=k, dim=dim, Q=q, N=n)


# ---------------------------------------------------------------------------- restatement
class Oracle:
    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        L.oracle_f16_to_f32.restype = C.c_float
        L.oracle_f16_to_f32.argtypes = [C.c_uint16]
        L.oracle_f32_to_f16.restype = C.c_uint16
        L.oracle_f32_to_f16.argtypes = [C.c_float]
        L.oracle_convert_f32_to_f16.argtypes = [_f32p, C.c_void_p, C.c_uint64]
        L.oracle_quantize_i8_row.restype = C.c_float
        L.oracle_quantize_i8_row.argtypes = [_f32p, C.c_uint32, C.c_void_p]
        for name in ("oracle_dot_f32", "oracle_dot_f32_scalar"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [_f32p, _f32p, C.c_uint32]
        L.oracle_dot_f32_f16base.restype = C.c_float
        L.oracle_dot_f32_f16base.argtypes = [_f32p, C.c_void_p, C.c_uint32]
        for name in ("oracle_dot_f32_i8base", "oracle_dot_f32_i8base_scalar"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [_f32p, C.c_void_p, C.c_uint32, C.c_float]
        L.oracle_scores.argtypes = [C.c_void_p, _f32p, C.c_uint32, C.c_uint64, C.c_uint32, _f32p, _f32p]
        L.oracle_flat_topk.restype = C.c_uint32
        L.oracle_flat_topk.argtypes = [C.c_void_p, _f32p, C.c_uint32, C.c_uint64, C.c_uint32, _f32p,
                                       C.c_uint32, C.c_uint32, _u64p, _f32p]
        L.oracle_flat_topk_omp.restype = C.c_uint32
        L.oracle_flat_topk_omp.argtypes = [C.c_void_p, _f32p, C.c_uint32, C.c_uint64, C.c_uint32, _f32p,
                                           C.c_uint32, C.c_int, _u64p, _f32p]
        L.oracle_topk_of_scores.restype = C.c_uint32
        L.oracle_topk_of_scores.argtypes = [_f32p, C.c_uint64, C.c_uint32, _u64p, _f32p]
        L.oracle_refine_l2_topk.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, _f32p, _u32p,
                                            C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _u32p, _f32p]
        for name in ("oracle_l2_f16_gpu_order",):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [_f32p, C.c_void_p, C.c_uint32]
        L.oracle_base_row_to_f32.argtypes = [C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, _f32p]
        L.oracle_l2_cpu_double.restype = C.c_float
        L.oracle_l2_cpu_double.argtypes = [_f32p, C.c_void_p, C.c_uint32, C.c_uint32]

    # -- conversions
    def f32_to_f16(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(x.shape, dtype=np.uint16)
        if x.ndim == 2:  # row by row, as the reference tool does (tail elements use the scalar path)
            for i in range(x.shape[0]):
                self.lib.oracle_convert_f32_to_f16(_p(x[i], _f32p), out[i].ctypes.data, x.shape[1])
        else:
            self.lib.oracle_convert_f32_to_f16(_p(x, _f32p), out.ctypes.data, x.size)
        return out

    def f16_to_f32(self, h):
        h = np.ascontiguousarray(h, dtype=np.uint16)
        return np.array([self.lib.oracle_f16_to_f32(int(v)) for v in h.ravel()], dtype=np.float32).reshape(h.shape)

    def base_row_to_f32(self, row, dtype, scale=0.0):
        """to_f32_row.h:10-34 on one row (any dtype)."""
        row = np.ascontiguousarray(row)
        out = np.empty(row.shape[0], dtype=np.float32)
        self.lib.oracle_base_row_to_f32(row.ctypes.data, float(scale), dtype, row.shape[0], _p(out, _f32p))
        return out

    def quantize_i8(self, rows_f32):
        rows = np.ascontiguousarray(rows_f32, dtype=np.float32)
        out = np.empty(rows.shape, dtype=np.int8)
        scales = np.empty(rows.shape[0], dtype=np.float32)
        for i in range(rows.shape[0]):
            scales[i] = self.lib.oracle_quantize_i8_row(_p(rows[i], _f32p), rows.shape[1], out[i].ctypes.data)
        return out, scales

    # -- scoring / selection
    def scores(self, base, dtype, q, scales=None):
        base = np.ascontiguousarray(base)
        q = np.ascontiguousarray(q, dtype=np.float32)
        n, d = base.shape
        out = np.empty(n, dtype=np.float32)
        self.lib.oracle_scores(base.ctypes.data, _p(scales, _f32p), dtype, n, d, _p(q, _f32p), _p(out, _f32p))
        return out

    def flat_topk(self, base, dtype, queries, k, scales=None):
        base = np.ascontiguousarray(base)
        queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, base.shape[1])
        n, d = base.shape
        nq = queries.shape[0]
        ke = min(k, n)
        ids = np.zeros((nq, max(ke, 1)), dtype=np.uint64)
        sc = np.zeros((nq, max(ke, 1)), dtype=np.float32)
        got = self.lib.oracle_flat_topk(base.ctypes.data, _p(scales, _f32p), dtype, n, d, _p(queries, _f32p),
                                        nq, k, _p(ids, _u64p), _p(sc, _f32p))
        return ids[:, :got], sc[:, :got]

    def flat_topk_omp(self, base, dtype, q, k, threads, scales=None):
        base = np.ascontiguousarray(base)
        n, d = base.shape
        ids = np.zeros(k, dtype=np.uint64)
        sc = np.zeros(k, dtype=np.float32)
        got = self.lib.oracle_flat_topk_omp(base.ctypes.data, _p(scales, _f32p), dtype, n, d,
                                            _p(np.ascontiguousarray(q, dtype=np.float32), _f32p), k, threads,
                                            _p(ids, _u64p), _p(sc, _f32p))
        return ids[:got], sc[:got]

    def refine(self, base, dtype, queries, cand, K, mode=0):
        base = np.ascontiguousarray(base)
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        cand = np.ascontiguousarray(cand, dtype=np.uint32)
        n, d = base.shape
        Q, R = cand.shape
        ids = np.empty((Q, K), dtype=np.uint32)
        dist = np.empty((Q, K), dtype=np.float32)
        self.lib.oracle_refine_l2_topk(base.ctypes.data, dtype, n, d, _p(queries, _f32p), _p(cand, _u32p),
                                       Q, R, K, mode, _p(ids, _u32p), _p(dist, _f32p))
        return ids, dist


# ---------------------------------------------------------------------------- real reference
class Reference:
    @staticmethod
    def available():
        return os.path.exists(os.path.join(HERE, "_ref", "libnvdb_ref.so"))

    def __init__(self):
        L = self.lib = C.CDLL(os.path.join(HERE, "_ref", "libnvdb_ref.so"))
        L.ref_last_error.restype = C.c_char_p
        L.ref_dot_f32.restype = C.c_float
        L.ref_dot_f32.argtypes = [_f32p, _f32p, C.c_uint32]
        L.ref_dot_f32_f16base.restype = C.c_float
        L.ref_dot_f32_f16base.argtypes = [_f32p, C.c_void_p, C.c_uint32]
        L.ref_dot_f32_i8base.restype = C.c_float
        L.ref_dot_f32_i8base.argtypes = [_f32p, C.c_void_p, C.c_uint32, C.c_float]
        L.ref_set_force_scalar.argtypes = [C.c_int]
        L.ref_dataset_open.restype = C.c_void_p
        L.ref_dataset_open.argtypes = [C.c_char_p]
        L.ref_dataset_close.argtypes = [C.c_void_p]
        L.ref_dataset_count.restype = C.c_uint64
        L.ref_dataset_count.argtypes = [C.c_void_p]
        L.ref_flat_search.restype = C.c_int
        L.ref_flat_search.argtypes = [C.c_void_p, _f32p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, _u64p, _f32p,
                                      C.POINTER(C.c_double)]
        L.ref_omp_max_threads.restype = C.c_int
        if hasattr(L, "ref_f16_to_f32_array"):
            L.ref_f16_to_f32_array.argtypes = [C.c_void_p, C.c_uint64, _f32p]
            L.ref_base_row_to_f32.restype = C.c_int
            L.ref_base_row_to_f32.argtypes = [C.c_void_p, C.c_uint64, _f32p]
        self.bin = os.path.join(HERE, "_ref", "bin")

    def open(self, path):
        h = self.lib.ref_dataset_open(path.encode())
        if not h:
            raise RuntimeError(self.lib.ref_last_error().decode())
        return h

    def close(self, h):
        self.lib.ref_dataset_close(h)

    def base_row_to_f32(self, h, row, dim):
        out = np.empty(dim, dtype=np.float32)
        if self.lib.ref_base_row_to_f32(h, row, _p(out, _f32p)) != 0:
            raise RuntimeError(self.lib.ref_last_error().decode())
        return out

    def flat_search(self, h, queries, k, mode=0, threads=0, want_results=True):
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        nq = queries.shape[0]
        ids = np.zeros((nq, k), dtype=np.uint64) if want_results else None
        sc = np.zeros((nq, k), dtype=np.float32) if want_results else None
        ms = C.c_double(0.0)
        got = self.lib.ref_flat_search(h, _p(queries, _f32p), nq, k, mode, threads, _p(ids, _u64p), _p(sc, _f32p),
                                       C.byref(ms))
        if got < 0:
            raise RuntimeError(self.lib.ref_last_error().decode())
        if want_results:
            return ids[:, :got], sc[:, :got], ms.value
        return None, None, ms.value

    def run_tool(self, name, *args, env=None):
        e = dict(os.environ)
        if env:
            e.update(env)
        return subprocess.run([os.path.join(self.bin, name), *map(str, args)], check=True, capture_output=True,
                              text=True, env=e).stdout
