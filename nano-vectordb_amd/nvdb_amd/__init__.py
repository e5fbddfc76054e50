"""nvdb_amd -- thin ctypes binding of libnvdb_hip.so (include/nvdb_hip.h).

This is plumbing for the tests and bench.py; the product is the C-ABI library and the C++ host
layer.  There is no CPU implementation behind this module: if the HIP library is missing, or no
GPU is present when a computation is requested, the call raises.

Names follow the reference's host API: `FlatIndexHIP.search_topk_dot` mirrors
nvdb::FlatIndex::search_topk_dot (reference include/nvdb/flat_index.h:11-13), `l2_topk_batch`
mirrors nvdb::cuda_l2_topk_batch (include/nvdb/cuda_refine.h:25-38).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libnvdb_hip.so")
DEV_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libnvdb_hip_dev.so")   # developer build: + nvdb_hip_debug_* (include/nvdb_hip_dev.h)

DT_F32, DT_F16, DT_I8 = 1, 2, 3
REFINE_KMAX = 64
_NP_OF = {DT_F32: np.float32, DT_F16: np.uint16, DT_I8: np.int8}

EXPORTS = [
    "nvdb_hip_abi_version", "nvdb_hip_device_count", "nvdb_hip_create", "nvdb_hip_destroy", "nvdb_hip_last_error",
    "nvdb_hip_upload_corpus", "nvdb_hip_adopt_corpus", "nvdb_hip_generate_corpus", "nvdb_hip_corpus_info",
    "nvdb_hip_download_rows", "nvdb_hip_search_batch", "nvdb_hip_search_batch_dev", "nvdb_hip_search_check",
    "nvdb_hip_get_stats", "nvdb_hip_collect_kernel_times", "nvdb_hip_merge_topk_dev", "nvdb_hip_merge_topk_strided_dev", "nvdb_merge_topk_host", "nvdb_hip_set_option",
    "nvdb_hip_refine_l2_topk", "nvdb_hip_refine_l2_topk_dev", "nvdb_synth_rows_f32", "nvdb_f32_to_f16",
    "nvdb_quantize_i8_rows",
    "nvdb_hip_group_create", "nvdb_hip_group_destroy", "nvdb_hip_group_last_error", "nvdb_hip_group_size", "nvdb_hip_group_ctx",
    "nvdb_hip_group_exchange", "nvdb_hip_group_upload_corpus", "nvdb_hip_group_generate_corpus", "nvdb_hip_group_set_option",
    "nvdb_hip_group_search_batch",
]
# only in libnvdb_hip_dev.so; the product library must NOT export them (tests/test_cabi_cpu.py)
DEV_EXPORTS = ["nvdb_hip_debug_filter_variant", "nvdb_hip_debug_clock", "nvdb_hip_debug_clock_i8", "nvdb_permuted_tile", "nvdb_hip_debug_tile_ranges"]


class NvdbError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"nvdb_hip status {status}: {msg}")
        self.status = status


class Timing(C.Structure):
    _fields_ = [("h2d_ms", C.c_float), ("kernel_ms", C.c_float), ("d2h_ms", C.c_float), ("total_ms", C.c_float),
                ("threads", C.c_uint32), ("nwarps", C.c_uint32), ("K", C.c_uint32), ("R", C.c_uint32),
                ("shmem_bytes", C.c_size_t), ("dbg_q", C.c_uint32),
                ("dbg_dist_cycles_avg", C.c_double), ("dbg_write_cycles_avg", C.c_double),
                ("dbg_merge_cycles_avg", C.c_double), ("dbg_dist_pct", C.c_double), ("dbg_write_pct", C.c_double),
                ("dbg_merge_pct", C.c_double)]


class ScanStats(C.Structure):
    _fields_ = [("path", C.c_uint32), ("chunks", C.c_uint32), ("rows_scanned", C.c_uint64), ("candidates", C.c_uint64),
                ("overflow_queries", C.c_uint32), ("bound_violations", C.c_uint32), ("filter_kernel_ms", C.c_float),
                ("other_kernel_ms", C.c_float), ("i8_stage1_tiles", C.c_uint32), ("i8_stage2_blocks", C.c_uint32),
                ("sticky_overflow", C.c_uint32), ("sticky_violations", C.c_uint32)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class GroupStats(C.Structure):
    _fields_ = [("shards", C.c_uint32), ("exchange", C.c_uint32), ("host_merge_fallbacks", C.c_uint32), ("bytes_per_rank", C.c_uint64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


_lib = None
_dev_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so.7; libnvdb_hip.so is linked
    against /opt/rocm's, which has the same soname.  Whichever is loaded first serves both, and torch cannot
    initialise ("No HIP GPUs are available") on top of /opt/rocm's.  bench.py and the tests share device
    buffers and streams with torch, so when torch is installed its runtime is loaded first -- without importing
    torch.  Stand-alone C++ users of libnvdb_hip.so (the tools in bin/) are unaffected."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _share_rccl_with_torch():
    """Same for RCCL, which the device group binds with dlopen("librccl.so.1") on first use: PyTorch's bundled copy is built
    against PyTorch's HIP runtime, so it is the one to have in the process.  PyTorch is IMPORTED for that (only DeviceGroup
    does this): a bare dlopen of its librccl.so ahead of a later `import torch` changed the order in which the two tear their
    statics down and aborted the interpreter at exit ("double free or corruption") once both had been used."""
    import importlib.util
    if importlib.util.find_spec("torch") is None:
        return
    import torch  # noqa: F401  (loads libamdhip64 / librccl in its own order)


def _bind(L, dev):
    vp, u32, u64, i64, f32p = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.POINTER(C.c_float)
    L.nvdb_hip_abi_version.restype = C.c_int
    L.nvdb_hip_device_count.restype = C.c_int
    L.nvdb_hip_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.nvdb_hip_destroy.argtypes = [vp]
    L.nvdb_hip_destroy.restype = None
    L.nvdb_hip_last_error.argtypes = [vp]
    L.nvdb_hip_last_error.restype = C.c_char_p
    L.nvdb_hip_upload_corpus.argtypes = [vp, vp, vp, u64, u32, u32, u64]
    L.nvdb_hip_adopt_corpus.argtypes = [vp, vp, vp, u64, u32, u32, u64]
    L.nvdb_hip_generate_corpus.argtypes = [vp, u64, u64, u32, u32, u64]
    L.nvdb_hip_corpus_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), f32p]
    L.nvdb_hip_download_rows.argtypes = [vp, u64, u64, vp, vp]
    L.nvdb_hip_search_batch.argtypes = [vp, vp, u32, u32, vp, vp, C.POINTER(u32), C.POINTER(Timing)]
    L.nvdb_hip_search_batch_dev.argtypes = [vp, vp, u32, u32, vp, vp, vp]
    L.nvdb_hip_search_check.argtypes = [vp, C.POINTER(ScanStats)]
    L.nvdb_hip_get_stats.argtypes = [vp, C.POINTER(ScanStats)]
    L.nvdb_hip_collect_kernel_times.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                C.POINTER(C.c_double)]
    L.nvdb_hip_merge_topk_dev.argtypes = [vp, vp, vp, u32, u32, u32, vp, vp, vp]
    L.nvdb_hip_merge_topk_strided_dev.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, u32, u32, u32, vp, vp, vp]
    L.nvdb_merge_topk_host.argtypes = [vp, vp, u32, u32, u32, vp, vp]
    L.nvdb_hip_set_option.argtypes = [vp, C.c_char_p, i64]
    L.nvdb_hip_refine_l2_topk.argtypes = [vp, vp, vp, u32, u32, u32, vp, vp, C.POINTER(Timing)]
    L.nvdb_hip_refine_l2_topk_dev.argtypes = [vp, vp, vp, u32, u32, u32, vp, vp, vp]
    L.nvdb_synth_rows_f32.argtypes = [u64, u64, u64, u32, vp]
    L.nvdb_synth_rows_f32.restype = None
    L.nvdb_f32_to_f16.argtypes = [vp, vp, u64]
    L.nvdb_f32_to_f16.restype = None
    L.nvdb_quantize_i8_rows.argtypes = [vp, u64, u32, vp, vp]
    L.nvdb_quantize_i8_rows.restype = None
    L.nvdb_hip_group_create.argtypes = [C.POINTER(C.c_int), u32, C.POINTER(vp)]
    L.nvdb_hip_group_destroy.argtypes = [vp]
    L.nvdb_hip_group_destroy.restype = None
    L.nvdb_hip_group_last_error.argtypes = [vp]
    L.nvdb_hip_group_last_error.restype = C.c_char_p
    L.nvdb_hip_group_size.argtypes = [vp]
    L.nvdb_hip_group_size.restype = u32
    L.nvdb_hip_group_ctx.argtypes = [vp, u32]
    L.nvdb_hip_group_ctx.restype = vp
    L.nvdb_hip_group_exchange.argtypes = [vp, C.POINTER(C.c_char_p)]
    L.nvdb_hip_group_exchange.restype = C.c_int
    L.nvdb_hip_group_upload_corpus.argtypes = [vp, vp, vp, u64, u32, u32]
    L.nvdb_hip_group_generate_corpus.argtypes = [vp, u64, u64, u32, u32]
    L.nvdb_hip_group_set_option.argtypes = [vp, C.c_char_p, i64]
    L.nvdb_hip_group_search_batch.argtypes = [vp, vp, u32, u32, vp, vp, C.POINTER(u32), C.POINTER(GroupStats)]
    for name in EXPORTS:
        getattr(L, name)
    if dev:
        L.nvdb_hip_debug_filter_variant.argtypes = [vp, C.c_int, u32, u32, f32p]
        L.nvdb_hip_debug_clock.argtypes = [vp, C.c_int, u32, C.c_float, f32p]
        L.nvdb_hip_debug_clock_i8.argtypes = [vp, C.c_int, u32, C.c_float, f32p]
        L.nvdb_permuted_tile.argtypes = [u32, u32]
        L.nvdb_hip_debug_tile_ranges.argtypes = [vp, u32, u32, f32p, C.POINTER(u32), C.POINTER(u32), f32p]
        L.nvdb_permuted_tile.restype = u32
    return L


def load_library():
    """Load libnvdb_hip.so (the product library); raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C nano-vectordb_amd` "
                          "(python -c 'import __graft_entry__ as g; g.build()')")
    _share_hip_runtime_with_torch()
    _lib = _bind(C.CDLL(LIB_PATH), dev=False)
    return _lib


def load_dev_library():
    """Load libnvdb_hip_dev.so: the same ABI plus the developer entry points (tools_dev/ only)."""
    global _dev_lib
    if _dev_lib is not None:
        return _dev_lib
    if not os.path.exists(DEV_LIB_PATH):
        raise ImportError(f"{DEV_LIB_PATH} is missing: build it with `make -C nano-vectordb_amd`")
    _share_hip_runtime_with_torch()
    _dev_lib = _bind(C.CDLL(DEV_LIB_PATH), dev=True)
    return _dev_lib


# ------------------------------------------------------------------------------- host-side helpers (no GPU)
def synth_rows_f32(seed, row0, nrows, dim):
    out = np.empty((nrows, dim), dtype=np.float32)
    load_library().nvdb_synth_rows_f32(seed, row0, nrows, dim, out.ctypes.data)
    return out


def f32_to_f16(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    load_library().nvdb_f32_to_f16(x.ctypes.data, out.ctypes.data, x.size)
    return out


def quantize_i8(rows):
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.empty(rows.shape, dtype=np.int8)
    scales = np.empty(rows.shape[0], dtype=np.float32)
    load_library().nvdb_quantize_i8_rows(rows.ctypes.data, rows.shape[0], rows.shape[1], out.ctypes.data,
                                         scales.ctypes.data)
    return out, scales


def synth_corpus(seed, row0, nrows, dim, dtype):
    """CPU twin of HipContext.generate_corpus (same bits)."""
    f = synth_rows_f32(seed, row0, nrows, dim)
    if dtype == DT_F32:
        return f, None
    if dtype == DT_F16:
        return f32_to_f16(f), None
    return quantize_i8(f)


def merge_topk_host(ids, scores):
    """ids/scores: [nshards, nq, k] -> ([nq,k],[nq,k]) in (score desc, id asc) order."""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    s, nq, k = ids.shape
    oi = np.empty((nq, k), dtype=np.uint64)
    os_ = np.empty((nq, k), dtype=np.float32)
    st = load_library().nvdb_merge_topk_host(ids.ctypes.data, scores.ctypes.data, s, nq, k, oi.ctypes.data,
                                             os_.ctypes.data)
    if st:
        raise NvdbError(st, "merge_topk_host")
    return oi, os_


# ------------------------------------------------------------------------------- device context
class HipContext:
    """One GPU: resident corpus + workspace (nvdb_hip_ctx)."""

    def __init__(self, device=0, dev=False):
        self.lib = load_dev_library() if dev else load_library()   # dev=True: a context of the developer build (tools_dev/)
        h = C.c_void_p()
        st = self.lib.nvdb_hip_create(device, C.byref(h))
        if st:
            raise NvdbError(st, self.lib.nvdb_hip_last_error(None).decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.nvdb_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        if st:
            raise NvdbError(st, self.lib.nvdb_hip_last_error(self.h).decode())

    # -- corpus
    def upload_corpus(self, rows, dtype, scales=None, row_base=0):
        if dtype == DT_F16 and getattr(rows, "dtype", None) == np.float16:
            rows = rows.view(np.uint16)                     # IEEE half bits, not a value conversion to integers
        rows = np.ascontiguousarray(rows, dtype=_NP_OF[dtype])
        sc = np.ascontiguousarray(scales, dtype=np.float32) if scales is not None else None
        self._chk(self.lib.nvdb_hip_upload_corpus(self.h, rows.ctypes.data, sc.ctypes.data if sc is not None else None,
                                                  rows.shape[0], rows.shape[1], dtype, row_base))

    def adopt_corpus(self, dev_ptr, n, dim, dtype, dev_scales_ptr=None, row_base=0):
        self._chk(self.lib.nvdb_hip_adopt_corpus(self.h, dev_ptr, dev_scales_ptr, n, dim, dtype, row_base))

    def generate_corpus(self, seed, n, dim, dtype, row_base=0):
        self._chk(self.lib.nvdb_hip_generate_corpus(self.h, seed, n, dim, dtype, row_base))

    def corpus_info(self):
        n, dim, dt, base, mx = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_float()
        self._chk(self.lib.nvdb_hip_corpus_info(self.h, C.byref(n), C.byref(dim), C.byref(dt), C.byref(base), C.byref(mx)))
        return dict(n=n.value, dim=dim.value, dtype=dt.value, row_base=base.value, max_row_norm=mx.value)

    def download_rows(self, row0, nrows):
        info = self.corpus_info()
        out = np.empty((nrows, info["dim"]), dtype=_NP_OF[info["dtype"]])
        sc = np.empty(nrows, dtype=np.float32) if info["dtype"] == DT_I8 else None
        self._chk(self.lib.nvdb_hip_download_rows(self.h, row0, nrows, out.ctypes.data,
                                                  sc.ctypes.data if sc is not None else None))
        return out, sc

    def set_option(self, key, value):
        self._chk(self.lib.nvdb_hip_set_option(self.h, key.encode(), int(value)))

    # -- flat scan
    def search_batch(self, queries, k, want_timing=False):
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim == 1:
            queries = queries[None, :]
        nq = queries.shape[0]
        ids = np.full((nq, max(k, 1)), np.iinfo(np.uint64).max, dtype=np.uint64)
        scores = np.full((nq, max(k, 1)), -np.inf, dtype=np.float32)
        keff = C.c_uint32(0)
        t = Timing()
        self._chk(self.lib.nvdb_hip_search_batch(self.h, queries.ctypes.data, nq, k, ids.ctypes.data, scores.ctypes.data,
                                                 C.byref(keff), C.byref(t) if want_timing else None))
        ke = keff.value if k > 0 else 0
        if want_timing:
            return ids[:, :ke], scores[:, :ke], t
        return ids[:, :ke], scores[:, :ke]

    def search_batch_dev(self, dev_q, nq, k, dev_out_ids, dev_out_scores, stream=None):
        self._chk(self.lib.nvdb_hip_search_batch_dev(self.h, dev_q, nq, k, dev_out_ids, dev_out_scores, stream))

    def search_check(self):
        s = ScanStats()
        self._chk(self.lib.nvdb_hip_search_check(self.h, C.byref(s)))
        return s.as_dict()

    def stats(self):
        s = ScanStats()
        self._chk(self.lib.nvdb_hip_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    def collect_kernel_times(self):
        n, ms, fl, by = C.c_uint32(), C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.nvdb_hip_collect_kernel_times(self.h, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
        return dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)

    def merge_topk_dev(self, dev_ids, dev_scores, nshards, nq, k, dev_out_ids, dev_out_scores, stream=None):
        self._chk(self.lib.nvdb_hip_merge_topk_dev(self.h, dev_ids, dev_scores, nshards, nq, k, dev_out_ids,
                                                   dev_out_scores, stream))

    def merge_topk_strided_dev(self, dev_ids, dev_scores, stride_ids, stride_scores, nshards, nq, k, dev_out_ids, dev_out_scores,
                               stream=None):
        self._chk(self.lib.nvdb_hip_merge_topk_strided_dev(self.h, dev_ids, dev_scores, stride_ids, stride_scores, nshards, nq, k,
                                                           dev_out_ids, dev_out_scores, stream))

    # -- refine
    def refine_l2_topk(self, queries, cand_ids, K, want_dist=True, want_timing=False):
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        cand_ids = np.ascontiguousarray(cand_ids, dtype=np.uint32)
        Q, R = cand_ids.shape if cand_ids.ndim == 2 else (0, 0)
        ids = np.full((Q, K), 0xFFFFFFFF, dtype=np.uint32)
        dist = np.full((Q, K), 1e30, dtype=np.float32) if want_dist else None
        t = Timing()
        self._chk(self.lib.nvdb_hip_refine_l2_topk(self.h, queries.ctypes.data, cand_ids.ctypes.data, Q, R, K,
                                                   ids.ctypes.data, dist.ctypes.data if dist is not None else None,
                                                   C.byref(t) if want_timing else None))
        return (ids, dist, t) if want_timing else (ids, dist)

    def refine_l2_topk_dev(self, dev_q, dev_cand, Q, R, K, dev_out_ids, dev_out_dist, stream=None):
        self._chk(self.lib.nvdb_hip_refine_l2_topk_dev(self.h, dev_q, dev_cand, Q, R, K, dev_out_ids, dev_out_dist, stream))


class DeviceGroup:
    """One process, several GPUs (nvdb_hip_group): corpus row-sharded over `devices`, per-shard top-k exchanged by an RCCL
    all-gather (or peer copies when a device is listed twice), merged on devices[0]."""

    def __init__(self, devices):
        self.lib = load_library()
        if self.lib.nvdb_hip_device_count() > 0:             # (no GPU: creation fails below without ever needing RCCL)
            _share_rccl_with_torch()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        st = self.lib.nvdb_hip_group_create(arr, len(devices), C.byref(h))
        if st:
            raise NvdbError(st, self.lib.nvdb_hip_group_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.nvdb_hip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        if st:
            raise NvdbError(st, self.lib.nvdb_hip_group_last_error(self.h).decode())

    def size(self):
        return self.lib.nvdb_hip_group_size(self.h)

    def exchange(self):
        why = C.c_char_p()
        mode = self.lib.nvdb_hip_group_exchange(self.h, C.byref(why))
        return ("rccl" if mode == 1 else "peer-copy"), (why.value or b"").decode()

    def shard_stats(self, shard):
        s = ScanStats()
        st = self.lib.nvdb_hip_get_stats(self.lib.nvdb_hip_group_ctx(self.h, shard), C.byref(s))
        if st:
            raise NvdbError(st, "get_stats")
        return s.as_dict()

    def upload_corpus(self, rows, dtype, scales=None):
        if dtype == DT_F16 and getattr(rows, "dtype", None) == np.float16:
            rows = rows.view(np.uint16)
        rows = np.ascontiguousarray(rows, dtype=_NP_OF[dtype])
        sc = np.ascontiguousarray(scales, dtype=np.float32) if scales is not None else None
        self._chk(self.lib.nvdb_hip_group_upload_corpus(self.h, rows.ctypes.data, sc.ctypes.data if sc is not None else None,
                                                        rows.shape[0], rows.shape[1], dtype))

    def generate_corpus(self, seed, n, dim, dtype):
        self._chk(self.lib.nvdb_hip_group_generate_corpus(self.h, seed, n, dim, dtype))

    def set_option(self, key, value):
        self._chk(self.lib.nvdb_hip_group_set_option(self.h, key.encode(), int(value)))

    def search_batch(self, queries, k, want_stats=False):
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim == 1:
            queries = queries[None, :]
        nq = queries.shape[0]
        ids = np.full((nq, max(k, 1)), np.iinfo(np.uint64).max, dtype=np.uint64)
        scores = np.full((nq, max(k, 1)), -np.inf, dtype=np.float32)
        keff = C.c_uint32(0)
        gs = GroupStats()
        self._chk(self.lib.nvdb_hip_group_search_batch(self.h, queries.ctypes.data, nq, k, ids.ctypes.data, scores.ctypes.data,
                                                       C.byref(keff), C.byref(gs)))
        ke = keff.value if k > 0 else 0
        if want_stats:
            return ids[:, :ke], scores[:, :ke], gs.as_dict()
        return ids[:, :ke], scores[:, :ke]


class FlatIndexHIP:
    """GPU flat index with the reference's FlatIndex surface (include/nvdb/flat_index.h:11-13):
    constructed over a dataset, `search_topk_dot(q, k)` -> list of (id, score), best first; plus the
    batched entry the reference lacks."""

    def __init__(self, rows, dtype, scales=None, device=0, row_base=0):
        if rows is None or len(rows) == 0:
            raise RuntimeError("Empty base")          # src/flat_index.cpp:17
        self.ctx = HipContext(device)
        self.ctx.upload_corpus(rows, dtype, scales, row_base)

    def search_topk_dot(self, q, k):
        if q is None:
            raise RuntimeError("Null query")          # src/flat_index_pool.cpp:196
        if k == 0:
            return []                                 # src/flat_index.cpp:18
        ids, sc = self.ctx.search_batch(np.asarray(q, dtype=np.float32)[None, :], k)
        return [(int(i), float(s)) for i, s in zip(ids[0], sc[0])]

    def search_topk_dot_batch(self, queries, k):
        return self.ctx.search_batch(queries, k)


def l2_topk_batch(ctx, queries_f32, cand_ids, K, return_dist=True):
    """nvdb::cuda_l2_topk_batch on the context's resident corpus; returns (ids, dist, Timing)."""
    return ctx.refine_l2_topk(queries_f32, cand_ids, K, want_dist=return_dist, want_timing=True)
