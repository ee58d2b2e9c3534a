"""ctypes view of libpalace_hip.so (the C ABI declared in include/palace_hip.h).

This is plumbing for tests/ and bench.py: it adds no compute of its own and has no CPU
fallback -- if the HIP library is missing or a call fails, it raises.  The product's host side
is the C++ under palace_amd/host/, which links the same library.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("PALACE_HIP_SO") or os.path.join(_HERE, "libpalace_hip.so")     # override: A/B timing of two builds
_LIB = None


class PalaceError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.run(args + ["clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)
    return SO_PATH


class GraphParams(C.Structure):
    """palace_graph_params (defaults = generate_graph.cpp:20-44)."""
    _fields_ = [("max_end", C.c_int32), ("min_mapq", C.c_int32), ("max_nm", C.c_int32), ("enable_paired", C.c_int32),
                ("both_order", C.c_int32), ("reserved", C.c_int32), ("max_span_frac", C.c_double)]

    @classmethod
    def default(cls):
        return cls(300, 0, 5, 1, 0, 0, 0.80)


class BamCols(C.Structure):
    _fields_ = [("n", C.c_int64)] + [(k, C.c_void_p) for k in
                                     ("tid", "pos", "mtid", "mpos", "nm", "ref_len", "read_len", "clip_s", "clip_e",
                                      "flag", "mapq", "qkey", "sa_off")]


class Stage04Inputs(C.Structure):
    """palace_stage04_inputs"""
    _fields_ = [("n_segs", C.c_int32), ("min_count", C.c_int32), ("seed", C.c_void_p), ("tlen", C.c_void_p), ("rank", C.c_void_p),
                ("name_len", C.c_void_p), ("n_paths", C.c_int64), ("path_off", C.c_void_p), ("path_tok", C.c_void_p)]


SA_ITEM_DTYPE = np.dtype([(k, np.int32) for k in ("tid2", "pos2", "mapq2", "nm2", "clip_s2", "clip_e2", "len2", "rev2")])
CAND_DTYPE = np.dtype([("ord", np.int64), ("qkey", np.uint64), ("left", np.int32), ("right", np.int32),
                       ("mtid", np.int32), ("ref_len", np.int32), ("dL", np.int32), ("dR", np.int32),
                       ("nmL", np.int32), ("nmR", np.int32), ("mapqL", np.int16), ("mapqR", np.int16),
                       ("kind", np.uint8), ("cls", np.uint8), ("found", np.uint8), ("in_fastg", np.uint8),
                       ("oL", np.uint8), ("oR", np.uint8), ("pad0", np.uint8), ("pad1", np.uint8), ("sa_index", np.int32)])
EDGE_DTYPE = np.dtype([("left", np.int32), ("right", np.int32), ("counts", np.uint32, 4), ("oL", np.uint8),
                       ("oR", np.uint8), ("pad", np.uint8, 6)])
assert CAND_DTYPE.itemsize == 64 and EDGE_DTYPE.itemsize == 32 and SA_ITEM_DTYPE.itemsize == 32

_SIGS = {
    "palace_ctx_create": [C.c_int, C.POINTER(C.c_void_p)],
    "palace_ctx_create_prio": [C.c_int, C.c_int, C.POINTER(C.c_void_p)],
    "palace_ctx_create_on_stream": [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)],
    "palace_ctx_destroy": [C.c_void_p],
    "palace_sync": [C.c_void_p],
    "palace_malloc": [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)],
    "palace_free": [C.c_void_p, C.c_void_p],
    "palace_memset": [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t],
    "palace_h2d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "palace_d2h": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "palace_d2d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "palace_host_alloc": [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)],
    "palace_host_free": [C.c_void_p, C.c_void_p],
    "palace_d2h_async": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "palace_h2d_async": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "palace_mark_wait": [C.c_void_p, C.c_int],
    "palace_mark_wait_for": [C.c_void_p, C.c_int, C.c_double],
    "palace_wait_for_mark": [C.c_void_p, C.c_void_p, C.c_int],
    "palace_timer_begin": [C.c_void_p],
    "palace_timer_end": [C.c_void_p, C.POINTER(C.c_float)],
    "palace_mark": [C.c_void_p, C.c_int],
    "palace_mark_elapsed": [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)],
    "palace_eref_set_coder": [C.c_void_p, C.c_void_p],
    "palace_eref_index_refs": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    "palace_eref_table_reset": [C.c_void_p],
    "palace_eref_reserve": [C.c_void_p, C.c_int64],
    "palace_eref_count_reads": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64],
    "palace_eref_pack_reads": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p],
    "palace_eref_count_reads_packed": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64],
    "palace_eref_set_count_mode": [C.c_void_p, C.c_int, C.c_int64],
    "palace_eref_set_key_buckets": [C.c_void_p, C.c_void_p],
    "palace_eref_set_option": [C.c_void_p, C.c_char_p, C.c_int64],
    "palace_eref_scan_refs": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                              C.c_void_p],
    "palace_eref_probe_index_build": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)],
    "palace_eref_attach_probe_index": [C.c_void_p, C.c_void_p],
    "palace_eref_entry_layout": [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
    "palace_eref_entry_buffers_attach": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "palace_eref_entry_buffers": [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)],
    "palace_eref_entry_hits_from_counts": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t],
    "palace_eref_entry_hits_complete": [C.c_void_p, C.c_void_p, C.c_int64],
    "palace_eref_entry_counts_valid": [C.c_void_p, C.c_void_p],
    "palace_eref_probe_index_free": [C.c_void_p, C.c_void_p],
    "palace_eref_scan_refs_indexed": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                      C.c_void_p],
    "palace_eref_table_planes": [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)],
    "palace_eref_table_attach": [C.c_void_p, C.POINTER(C.c_void_p)],
    "palace_eref_table_invalidate": [C.c_void_p],
    "palace_eref_table_merge_slices": [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t],
    "palace_eref_table_merge_slices_packed": [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t],
    "palace_eref_table_pack_low": [C.c_void_p, C.c_void_p],
    "palace_eref_plane_pack": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "palace_eref_plane_unpack": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "palace_eref_table_lookup": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "palace_eref_table_popcounts": [C.c_void_p, C.POINTER(C.c_uint64)],
    "palace_graph_classify": [C.c_void_p, C.POINTER(BamCols), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_int64, C.POINTER(GraphParams), C.c_int64, C.c_void_p, C.c_void_p,
                              C.c_int64, C.POINTER(C.c_int64)],
    "palace_depth_sum_covered": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "palace_depth_per_contig": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p],
    "palace_graph_copy_numbers": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p],
    "palace_match_greedy": [C.c_void_p, C.c_int32, C.c_int64] + [C.c_void_p] * 10 + [C.POINTER(C.c_int32)],
    "palace_match_arcs_from_edges": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)],
    "palace_match_set_option": [C.c_void_p, C.c_char_p, C.c_int64],
    "palace_match_decompose": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                               C.c_int32, C.POINTER(C.c_void_p)],
    "palace_match_decompose_ex": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.c_int32, C.c_int32, C.POINTER(C.c_void_p)],
    "palace_graph_resolve": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(GraphParams), C.c_void_p,
                             C.c_void_p, C.c_int64, C.POINTER(C.c_int64)],
    "palace_graph_classify_ex": [C.c_void_p, C.POINTER(BamCols), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int64, C.POINTER(GraphParams), C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "palace_graph_classify_ix": [C.c_void_p, C.POINTER(BamCols), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(GraphParams), C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "palace_graph_fastg_offsets": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p],
    "palace_bgzf_inflate": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "palace_graph_score_border": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(GraphParams)],
    "palace_graph_resolve_ex": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(GraphParams), C.c_void_p,
                                C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)],
    "palace_stage04_create": [C.c_void_p, C.POINTER(Stage04Inputs), C.POINTER(C.c_void_p)],
    "palace_stage04_destroy": [C.c_void_p, C.c_void_p],
    "palace_stage04_reserve": [C.c_void_p, C.c_void_p, C.c_int64],
    "palace_stage04_filter": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64],
    "palace_stage04_flags": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64],
    "palace_stage04_counts": [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)],
    "palace_stage04_match": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32],
    "palace_stage04_result": [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)],
}


def declared_symbols():
    """Every function include/palace_hip.h declares (parsed from the header itself)."""
    import re
    text = open(os.path.join(_HERE, "..", "include", "palace_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(palace_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise PalaceError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _LIB = C.CDLL(SO_PATH)
        _LIB.palace_last_error.restype = C.c_char_p
        _LIB.palace_version.restype = C.c_char_p
        _LIB.palace_stream.restype = C.c_void_p
        _LIB.palace_stream.argtypes = [C.c_void_p]
        _LIB.palace_eref_packed_bytes.restype = C.c_size_t
        _LIB.palace_eref_packed_bytes.argtypes = [C.c_int64]
        for nm, rt in (("count", C.c_int64), ("bare_count", C.c_int64), ("bare", C.POINTER(C.c_uint64)), ("offsets", C.POINTER(C.c_int64)), ("verts", C.POINTER(C.c_int32)),
                       ("kind", C.POINTER(C.c_uint8)), ("iter", C.POINTER(C.c_int32)), ("open_at", C.POINTER(C.c_int32))):
            fn = getattr(_LIB, "palace_match_result_" + nm)
            fn.argtypes = [C.c_void_p]
            fn.restype = rt
        _LIB.palace_match_result_free.argtypes = [C.c_void_p]
        _LIB.palace_match_result_free.restype = None
        for name, sig in _SIGS.items():
            fn = getattr(_LIB, name)
            fn.argtypes = sig
            fn.restype = C.c_int
    return _LIB


def _check(rc: int, what: str):
    if rc != 0:
        raise PalaceError(f"{what} -> {rc}: {lib().palace_last_error().decode()}")


class DevBuf:
    """A device allocation owned through palace_malloc/palace_free."""

    def __init__(self, ctx: "Ctx", nbytes: int, dtype=np.uint8, shape=None):
        self.ctx, self.nbytes, self.dtype, self.shape = ctx, int(nbytes), np.dtype(dtype), shape
        p = C.c_void_p()
        _check(lib().palace_malloc(ctx.h, self.nbytes, C.byref(p)), "palace_malloc")
        self.ptr = p.value

    def to_host(self) -> np.ndarray:
        out = np.empty(self.nbytes // self.dtype.itemsize, dtype=self.dtype)
        _check(lib().palace_d2h(self.ctx.h, out.ctypes.data, self.ptr, self.nbytes), "palace_d2h")
        return out.reshape(self.shape) if self.shape is not None else out

    def free(self):
        if self.ptr:
            lib().palace_free(self.ctx.h, self.ptr)
            self.ptr = None


class Ctx:
    """One device context (one HIP stream).  `calls` go straight to the C ABI."""

    def __init__(self, device: int = 0, high_priority: bool = False, stream: int | None = None):
        """stream: a hipStream_t of the caller's to run on instead of a stream of the context's own (palace_ctx_create_on_stream)"""
        h = C.c_void_p()
        if stream:
            _check(lib().palace_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h)), "palace_ctx_create_on_stream")
        else:
            _check(lib().palace_ctx_create_prio(device, int(high_priority), C.byref(h)), "palace_ctx_create_prio")
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            lib().palace_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- memory ---------------------------------------------------------------------------
    def upload(self, arr: np.ndarray) -> DevBuf:
        a = np.ascontiguousarray(arr)
        b = DevBuf(self, max(a.nbytes, 1), a.dtype, a.shape)
        if a.nbytes:
            _check(lib().palace_h2d(self.h, b.ptr, a.ctypes.data, a.nbytes), "palace_h2d")
        return b

    def empty(self, shape, dtype) -> DevBuf:
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return DevBuf(self, max(n, 1), dtype, tuple(np.atleast_1d(shape)))

    def d2d(self, dst_ptr: int, src_ptr: int, nbytes: int):
        _check(lib().palace_d2d(self.h, dst_ptr, src_ptr, nbytes), "palace_d2d")

    def sync(self):
        _check(lib().palace_sync(self.h), "palace_sync")

    def timer_begin(self):
        _check(lib().palace_timer_begin(self.h), "palace_timer_begin")

    def timer_end(self) -> float:
        ms = C.c_float()
        _check(lib().palace_timer_end(self.h, C.byref(ms)), "palace_timer_end")
        return ms.value

    def mark(self, i: int):
        _check(lib().palace_mark(self.h, i), "palace_mark")

    def mark_wait(self, i: int):
        """the host waits for mark i of this context's stream"""
        _check(lib().palace_mark_wait(self.h, i), "palace_mark_wait")

    def wait_for_mark(self, other: "Ctx", i: int):
        _check(lib().palace_wait_for_mark(self.h, other.h, i), "palace_wait_for_mark")

    def mark_elapsed(self, a: int, b: int) -> float:
        ms = C.c_float()
        _check(lib().palace_mark_elapsed(self.h, a, b, C.byref(ms)), "palace_mark_elapsed")
        return ms.value

    @property
    def stream(self) -> int:
        return lib().palace_stream(self.h)

    # -- eref -----------------------------------------------------------------------------
    def eref_set_coder(self, header400: np.ndarray):
        h = np.ascontiguousarray(header400, dtype=np.uint8)
        assert h.size == 400
        _check(lib().palace_eref_set_coder(self.h, h.ctypes.data), "palace_eref_set_coder")

    def eref_table_reset(self):
        _check(lib().palace_eref_table_reset(self.h), "palace_eref_table_reset")

    def eref_count_reads(self, d_bases: DevBuf, d_offsets: DevBuf, n_reads: int, d_keep: DevBuf | None = None,
                         total_bases: int = -1):
        _check(lib().palace_eref_count_reads(self.h, d_bases.ptr, d_offsets.ptr, n_reads,
                                             d_keep.ptr if d_keep else None, total_bases), "palace_eref_count_reads")

    def eref_pack_reads(self, d_bases: DevBuf, d_offsets: DevBuf, n_reads: int, d_keep, total_bases: int, d_p0: DevBuf, d_p1: DevBuf, d_u: DevBuf):
        _check(lib().palace_eref_pack_reads(self.h, d_bases.ptr, d_offsets.ptr, n_reads, d_keep.ptr if d_keep else None, total_bases,
                                            d_p0.ptr, d_p1.ptr, d_u.ptr), "palace_eref_pack_reads")

    def eref_count_reads_packed(self, d_p0: DevBuf, d_p1: DevBuf, d_u: DevBuf, n_positions: int, n_reads_hint: int = 0):
        _check(lib().palace_eref_count_reads_packed(self.h, d_p0.ptr, d_p1.ptr, d_u.ptr, n_positions, n_reads_hint),
               "palace_eref_count_reads_packed")

    def eref_set_count_mode(self, mode: int, bucket_cap: int = 0):
        _check(lib().palace_eref_set_count_mode(self.h, mode, bucket_cap), "palace_eref_set_count_mode")

    def eref_set_key_buckets(self, buckets=None):
        """the level-1 buckets (0..127) count calls take in; None = all"""
        m = (C.c_uint32 * 4)(*([0xFFFFFFFF] * 4 if buckets is None else [0] * 4))
        for b in ([] if buckets is None else buckets):
            m[b >> 5] |= 1 << (b & 31)
        _check(lib().palace_eref_set_key_buckets(self.h, m), "palace_eref_set_key_buckets")

    def eref_set_option(self, name: str, value: int):
        _check(lib().palace_eref_set_option(self.h, name.encode(), value), "palace_eref_set_option")

    def eref_scan_refs(self, d_bases: DevBuf, d_offsets: DevBuf, n_refs: int, total_bases: int,
                       one_min: int, three_min: int, d_rows: DevBuf):
        _check(lib().palace_eref_scan_refs(self.h, d_bases.ptr, d_offsets.ptr, n_refs, total_bases,
                                           one_min, three_min, d_rows.ptr), "palace_eref_scan_refs")

    def eref_probe_index_build(self, d_bases: DevBuf, d_offsets: DevBuf, n_refs: int, total_bases: int) -> C.c_void_p:
        """Per-DB probe index (device resident); free with eref_probe_index_free."""
        h = C.c_void_p()
        _check(lib().palace_eref_probe_index_build(self.h, d_bases.ptr, d_offsets.ptr, n_refs, total_bases, C.byref(h)),
               "palace_eref_probe_index_build")
        return h

    def eref_attach_probe_index(self, index):
        """count calls that run as the final count also probe channel 0 of this DB (palace_eref_attach_probe_index); None detaches"""
        _check(lib().palace_eref_attach_probe_index(self.h, index), "palace_eref_attach_probe_index")

    def eref_entry_layout(self, index):
        """-> (bytes of the partial-count block, bytes of the hit-bit block) of a probe index (palace_eref_entry_layout)"""
        cb, hb = C.c_size_t(), C.c_size_t()
        _check(lib().palace_eref_entry_layout(index, C.byref(cb), C.byref(hb)), "palace_eref_entry_layout")
        return int(cb.value), int(hb.value)

    def eref_entry_buffers_attach(self, index, counts_ptr: int | None, hits_ptr: int | None):
        """the caller's device buffers stand in for the index's count / hit-bit blocks (None: the index's own)"""
        _check(lib().palace_eref_entry_buffers_attach(self.h, index, counts_ptr, hits_ptr), "palace_eref_entry_buffers_attach")

    def eref_entry_hits_from_counts(self, index, parts_ptr: int, n_parts: int, part_stride: int, off: int, nbytes: int):
        """sum n_parts partial-count arrays over the count block's bytes [off, off + nbytes) into the hit bits of that entry range"""
        _check(lib().palace_eref_entry_hits_from_counts(self.h, index, parts_ptr, n_parts, part_stride, off, nbytes), "palace_eref_entry_hits_from_counts")

    def eref_entry_counts_valid(self, index) -> bool:
        """a count call of this context (option probe_all_sets 2) stands behind the count block `index` points at"""
        return bool(lib().palace_eref_entry_counts_valid(self.h, index))

    def eref_entry_hits_complete(self, index, keys_counted: int = -1):
        """the hit-bit block of the attached index is whole: the next indexed scan starts from it"""
        _check(lib().palace_eref_entry_hits_complete(self.h, index, keys_counted), "palace_eref_entry_hits_complete")

    def eref_probe_index_free(self, index: C.c_void_p):
        _check(lib().palace_eref_probe_index_free(self.h, index), "palace_eref_probe_index_free")

    def eref_scan_refs_indexed(self, index: C.c_void_p, d_bases: DevBuf, d_offsets: DevBuf, n_refs: int, total_bases: int,
                               one_min: int, three_min: int, d_rows: DevBuf):
        _check(lib().palace_eref_scan_refs_indexed(self.h, index, d_bases.ptr, d_offsets.ptr, n_refs, total_bases,
                                                   one_min, three_min, d_rows.ptr), "palace_eref_scan_refs_indexed")

    def eref_index_refs(self, d_bases: DevBuf, d_offsets: DevBuf, n_refs: int, d_out: DevBuf, d_out_offsets: DevBuf):
        _check(lib().palace_eref_index_refs(self.h, d_bases.ptr, d_offsets.ptr, n_refs, d_out.ptr,
                                            d_out_offsets.ptr), "palace_eref_index_refs")

    def eref_table_lookup(self, keys: np.ndarray) -> np.ndarray:
        k = self.upload(np.ascontiguousarray(keys, dtype=np.uint32))
        out = self.empty(len(keys), np.uint8)
        _check(lib().palace_eref_table_lookup(self.h, k.ptr, len(keys), out.ptr), "palace_eref_table_lookup")
        res = out.to_host()
        k.free()
        out.free()
        return res

    def eref_table_popcounts(self):
        out = (C.c_uint64 * 3)()
        _check(lib().palace_eref_table_popcounts(self.h, out), "palace_eref_table_popcounts")
        return [int(v) for v in out]

    def eref_table_planes(self):
        ptrs = (C.c_void_p * 3)()
        nbytes = C.c_size_t()
        _check(lib().palace_eref_table_planes(self.h, ptrs, C.byref(nbytes)), "palace_eref_table_planes")
        return [int(p) for p in ptrs], int(nbytes.value)

    def eref_table_attach(self, ptrs):
        arr = (C.c_void_p * 3)(*[int(p) for p in ptrs])
        _check(lib().palace_eref_table_attach(self.h, arr), "palace_eref_table_attach")

    def match_set_option(self, name: str, value: int):
        _check(lib().palace_match_set_option(self.h, name.encode(), value), "palace_match_set_option")

    def eref_table_invalidate(self):
        _check(lib().palace_eref_table_invalidate(self.h), "palace_eref_table_invalidate")

    def eref_table_merge_slices(self, parts_ptr: int, n_parts: int, slice_off: int, slice_bytes: int, packed: bool = False):
        fn = lib().palace_eref_table_merge_slices_packed if packed else lib().palace_eref_table_merge_slices
        _check(fn(self.h, parts_ptr, n_parts, slice_off, slice_bytes), "palace_eref_table_merge_slices")

    def eref_table_pack_low(self, low_ptr: int):
        _check(lib().palace_eref_table_pack_low(self.h, low_ptr), "palace_eref_table_pack_low")

    @staticmethod
    def _bucket_mask(buckets):
        m = (C.c_uint32 * 4)(0, 0, 0, 0)
        for b in buckets:
            m[b >> 5] |= 1 << (b & 31)
        return m

    def eref_plane_pack(self, buckets, counts_ptr: int, keys_ptr: int, cap_keys: int, first_ptr: int):
        """the '>= 3' plane of the level-1 buckets `buckets` in sparse form (palace_eref_plane_pack): device pointers to
        512 * len(buckets) uint32 counts, cap_keys uint16 keys, 512 * len(buckets) + 1 uint64 prefix entries"""
        _check(lib().palace_eref_plane_pack(self.h, self._bucket_mask(buckets), counts_ptr, keys_ptr, cap_keys, first_ptr), "palace_eref_plane_pack")

    def eref_plane_unpack(self, buckets, counts_ptr: int, keys_ptr: int, cap_keys: int, first_ptr: int):
        _check(lib().palace_eref_plane_unpack(self.h, self._bucket_mask(buckets), counts_ptr, keys_ptr, cap_keys, first_ptr), "palace_eref_plane_unpack")


_ARC_BUFFERS = {}


def match_arcs_from_edges(cn: np.ndarray, edges: np.ndarray, min_count: int = 5, reuse: bool = False):
    """palace_match_arcs_from_edges (host code of the library) -> (copies, src, dst, weight); `edges` is an
    EDGE_DTYPE array as palace_graph_resolve writes it.  reuse=True hands out the same output arrays on every call
    of this size (valid until the next call) instead of fresh ones."""
    cn = np.ascontiguousarray(cn, dtype=np.int32)
    e = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
    key = (len(cn), len(e))
    if reuse and key in _ARC_BUFFERS:
        copies, src, dst, w = _ARC_BUFFERS[key]
    else:
        copies = np.empty(len(cn), np.int64)
        src = np.empty(2 * len(e), np.int32); dst = np.empty(2 * len(e), np.int32); w = np.empty(2 * len(e), np.int64)
        if reuse:
            _ARC_BUFFERS.clear()
            _ARC_BUFFERS[key] = (copies, src, dst, w)
    n = C.c_int64()
    _check(lib().palace_match_arcs_from_edges(cn.ctypes.data, len(cn), e.ctypes.data, len(e), min_count, copies.ctypes.data,
                                              src.ctypes.data, dst.ctypes.data, w.ctypes.data, C.byref(n)),
           "palace_match_arcs_from_edges")
    return copies, src[:n.value], dst[:n.value], w[:n.value]


class MatchResult:
    """Views into a palace_match_result (no copies); call free() -- or use as a context manager -- when done."""

    def __init__(self, handle):
        L = lib()
        self._h = handle
        n = self.n = L.palace_match_result_count(handle)
        self.off = np.ctypeslib.as_array(L.palace_match_result_offsets(handle), shape=(n + 1,))
        nv = int(self.off[-1])
        self.verts = np.ctypeslib.as_array(L.palace_match_result_verts(handle), shape=(max(nv, 1),))[:nv]
        mk = lambda f, dt: (np.ctypeslib.as_array(f(handle), shape=(max(n, 1),))[:n] if n else np.zeros(0, dt))
        self.kind = mk(L.palace_match_result_kind, np.uint8)
        self.iter = mk(L.palace_match_result_iter, np.int32)
        self.open_at = mk(L.palace_match_result_open_at, np.int32)

    def free(self):
        if self._h is not None:
            self.off = self.verts = self.kind = self.iter = self.open_at = None
            lib().palace_match_result_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()


def match_decompose_views(ctx: "Ctx", copies: np.ndarray, src: np.ndarray, dst: np.ndarray, iterations: int = 10,
                          aggressive: bool = False, compact: bool = False) -> MatchResult:
    cp = np.ascontiguousarray(copies, dtype=np.int64)
    s = np.ascontiguousarray(src, dtype=np.int32)
    d = np.ascontiguousarray(dst, dtype=np.int32)
    res = C.c_void_p()
    _check(lib().palace_match_decompose_ex(ctx.h, len(cp), cp.ctypes.data, len(s), s.ctypes.data, d.ctypes.data,
                                           iterations, int(aggressive), int(compact), C.byref(res)), "palace_match_decompose")
    r = MatchResult(res)
    if compact:
        r.n_bare = int(lib().palace_match_result_bare_count(res))
        words = (len(cp) + 63) // 64
        r.bare = np.ctypeslib.as_array(lib().palace_match_result_bare(res), shape=(max(1, words),))[:words]
    return r


def match_decompose(ctx: "Ctx", copies: np.ndarray, src: np.ndarray, dst: np.ndarray, iterations: int = 10,
                    aggressive: bool = False):
    """palace_match_decompose -> (offsets, verts, kind, iter, open_at) as numpy copies."""
    with match_decompose_views(ctx, copies, src, dst, iterations, aggressive) as r:
        return r.off.copy(), r.verts.copy(), r.kind.copy(), r.iter.copy(), r.open_at.copy()


class Stage04:
    """The resident stage-04 object (palace_stage04_*): filter_graph.py's selection + matching on the device."""
    COUNT_NAMES = ("edges", "juncs", "kept_pass2", "kept_pass3_more", "segs_selected", "segs_rescued", "arcs", "segs_filtered")

    def __init__(self, ctx: "Ctx", seed, tlen, rank, name_len, path_off, path_tok, min_count: int = 5):
        self.ctx = ctx
        self._keep = [np.ascontiguousarray(seed, np.uint8), np.ascontiguousarray(tlen, np.int32), np.ascontiguousarray(rank, np.int32),
                      np.ascontiguousarray(name_len, np.int32), np.ascontiguousarray(path_off, np.int64),
                      np.ascontiguousarray(path_tok, np.int32)]
        sd, tl, rk, nl, po, pt = self._keep
        assert len(sd) == len(tl) == len(rk) == len(nl) and len(po) >= 1
        inp = Stage04Inputs(len(sd), min_count, sd.ctypes.data, tl.ctypes.data, rk.ctypes.data, nl.ctypes.data, len(po) - 1,
                            po.ctypes.data, pt.ctypes.data if len(pt) else None)
        h = C.c_void_p()
        _check(lib().palace_stage04_create(ctx.h, C.byref(inp), C.byref(h)), "palace_stage04_create")
        self.h, self.n_segs = h, len(sd)

    def filter(self, d_edges_ptr: int, d_n_edges_ptr: int, edge_bound: int):
        _check(lib().palace_stage04_filter(self.ctx.h, self.h, d_edges_ptr, d_n_edges_ptr, edge_bound), "palace_stage04_filter")

    def counts(self) -> dict:
        out = (C.c_int64 * 8)()
        _check(lib().palace_stage04_counts(self.ctx.h, self.h, out), "palace_stage04_counts")
        return dict(zip(self.COUNT_NAMES, (int(v) for v in out)))

    def flags(self, n_edges: int):
        seg = np.zeros(self.n_segs, np.uint8)
        edge = np.zeros(n_edges, np.uint8)
        _check(lib().palace_stage04_flags(self.ctx.h, self.h, seg.ctypes.data, edge.ctypes.data if n_edges else None, n_edges),
               "palace_stage04_flags")
        return seg, edge

    def match(self, d_edges_ptr: int, d_cn_ptr: int, iterations: int = 10, aggressive: bool = False, use_paths: bool = True):
        _check(lib().palace_stage04_match(self.ctx.h, self.h, d_edges_ptr, d_cn_ptr, iterations, int(aggressive), int(use_paths)), "palace_stage04_match")

    def result(self):
        """-> (MatchResult view with .bare / .n_bare, contig_of array view); valid until the next match() / close()"""
        res, cof, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        _check(lib().palace_stage04_result(self.ctx.h, self.h, C.byref(res), C.byref(cof), C.byref(n)), "palace_stage04_result")
        r = MatchResult(res)
        r.n_bare = int(lib().palace_match_result_bare_count(res))
        words = (n.value + 63) // 64
        r.bare = np.ctypeslib.as_array(lib().palace_match_result_bare(res), shape=(max(1, words),))[:words]
        contig_of = np.ctypeslib.as_array(C.cast(cof, C.POINTER(C.c_int32)), shape=(max(1, n.value),))[: n.value]
        return r, contig_of

    def close(self):
        if self.h:
            lib().palace_stage04_destroy(self.ctx.h, self.h)
            self.h = None


def window_minimums(hit_ratio: float, perfect_ratio: float):
    """int(500 * float32(ratio)) -- the expression of extract_ref.cpp:513-514."""
    w = np.float32(500)
    return int(w * np.float32(hit_ratio)), int(w * np.float32(perfect_ratio))
