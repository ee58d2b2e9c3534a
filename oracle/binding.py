"""ctypes binding of oracle/libpalace_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package (palace_amd/).  Build with `make -C oracle libpalace_oracle.so`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libpalace_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("eref_oracle.c", "graph_oracle.cpp", "match_oracle.cpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "libpalace_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_table_new.restype = C.c_void_p
        L.orc_table_free.argtypes = [C.c_void_p]
        L.orc_table_clear.argtypes = [C.c_void_p]
        L.orc_count_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_index_ref.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_scan_ref.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int]
        L.orc_format_line.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_build_index_file.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_scan_index_file.argtypes = [C.c_char_p, C.c_void_p, C.c_float, C.c_float, C.c_char_p, C.c_size_t]
        L.orc_scan_index_file.restype = C.c_long
        L.orc_sample_ratio.argtypes = [C.c_int64]
        L.orc_rand_new.restype = C.c_void_p
        L.orc_rand_new.argtypes = [C.c_uint]
        L.orc_rand_next.argtypes = [C.c_void_p]
        L.orc_rand_free.argtypes = [C.c_void_p]
        L.orc_header_to_cc.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_cc_to_header.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_cc_from_picks.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_eref_reference_dead_cost.restype = C.c_uint64
    return _LIB


def _p(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def header_to_cc(header: np.ndarray) -> np.ndarray:
    cc = np.zeros(96, dtype=np.int16)
    h = np.ascontiguousarray(header, dtype=np.uint8)
    lib().orc_header_to_cc(_p(h), _p(cc))
    return cc


def header_from_picks(picks) -> np.ndarray:
    """400-byte index header for a choice of one of the 6 projection orders per k-mer offset."""
    p = np.ascontiguousarray(picks, dtype=np.uint8)
    assert p.shape == (32,) and p.max() < 6
    cc = np.zeros(96, dtype=np.int16)
    hdr = np.zeros(400, dtype=np.uint8)
    lib().orc_cc_from_picks(_p(p), _p(cc))
    lib().orc_cc_to_header(_p(cc), _p(hdr))
    return hdr


class CountTable:
    """The reference's 2^32-entry byte table (extract_ref.cpp:25-26), lazily paged."""

    def __init__(self):
        self.ptr = lib().orc_table_new()
        if not self.ptr:
            raise MemoryError("oracle count table (4 GiB virtual)")
        self.view = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint8)), shape=(1 << 32,))

    def count(self, bases: np.ndarray, offsets: np.ndarray, cc: np.ndarray, keep: np.ndarray | None = None):
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        k = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        lib().orc_count_reads(_p(b), _p(o), len(o) - 1, None if k is None else _p(k), _p(cc), self.ptr)

    def count_mt(self, bases: np.ndarray, offsets: np.ndarray, cc: np.ndarray, threads: int, keep: np.ndarray | None = None):
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        k = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        lib().orc_count_reads_mt(_p(b), _p(o), C.c_int64(len(o) - 1), None if k is None else _p(k), _p(cc), C.c_void_p(self.ptr), C.c_int(threads))

    def clear(self):
        lib().orc_table_clear(self.ptr)

    def lookup(self, keys: np.ndarray) -> np.ndarray:
        return self.view[np.asarray(keys, dtype=np.int64)]

    def free(self):
        if self.ptr:
            lib().orc_table_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()


def index_ref(seq: np.ndarray, cc: np.ndarray) -> np.ndarray:
    s = np.ascontiguousarray(seq, dtype=np.uint8)
    n = max(0, len(s) - 31)
    out = np.zeros(3 * n, dtype=np.uint32)
    if n:
        lib().orc_index_ref(_p(s), len(s), _p(cc), _p(out))
    return out


def scan_ref(idx: np.ndarray, ref_len: int, table: CountTable, hit_ratio: float, perfect_ratio: float):
    """-> (printed, n_intervals, el, intervals[n,2])"""
    n_int, el = C.c_int(0), C.c_int(0)
    cap = max(4, 2 * ref_len // 500 + 4)
    iv = np.zeros(2 * cap, dtype=np.int32)
    i = np.ascontiguousarray(idx, dtype=np.uint32)
    printed = lib().orc_scan_ref(_p(i), ref_len, table.ptr, hit_ratio, perfect_ratio, C.byref(n_int),
                                 C.byref(el), _p(iv), cap)
    return bool(printed), n_int.value, el.value, iv[: 2 * n_int.value].reshape(-1, 2).copy()


def format_line(ref_index: int, n_int: int, el: int, ref_len: int) -> bytes:
    buf = C.create_string_buffer(256)
    n = lib().orc_format_line(buf, 256, ref_index, n_int, el, ref_len)
    return buf.raw[:n]


def build_index_file(fasta: str, header: np.ndarray, index_path: str, len_path: str) -> None:
    h = np.ascontiguousarray(header, dtype=np.uint8)
    rc = lib().orc_build_index_file(fasta.encode(), _p(h), index_path.encode(), len_path.encode())
    if rc:
        raise OSError("orc_build_index_file failed")


def scan_index_file(index_path: str, table: CountTable, hit_ratio: float, perfect_ratio: float) -> bytes:
    cap = 1 << 24
    buf = C.create_string_buffer(cap)
    n = lib().orc_scan_index_file(index_path.encode(), table.ptr, hit_ratio, perfect_ratio, buf, cap)
    if n < 0:
        raise OSError("orc_scan_index_file failed")
    return buf.raw[:n]


def sample_ratio(fq1_bases: int) -> int:
    return lib().orc_sample_ratio(fq1_bases)


def glibc_rand_stream(seed: int, n: int) -> np.ndarray:
    st = lib().orc_rand_new(seed)
    out = np.array([lib().orc_rand_next(st) for _ in range(n)], dtype=np.int64)
    lib().orc_rand_free(st)
    return out


# ---- generateGraph restatement (oracle/graph_oracle.cpp) -------------------------------------
class _OrcRecords(C.Structure):
    _fields_ = [("n", C.c_int64), ("flag", C.c_void_p), ("tid", C.c_void_p), ("pos", C.c_void_p),
                ("mtid", C.c_void_p), ("mpos", C.c_void_p), ("mapq", C.c_void_p), ("nm", C.c_void_p),
                ("cigar_off", C.c_void_p), ("cigar", C.c_void_p), ("qname_off", C.c_void_p), ("qname", C.c_void_p),
                ("sa_off", C.c_void_p), ("sa", C.c_void_p), ("has_sa", C.c_void_p)]


class GraphOpts(C.Structure):
    _fields_ = [("max_end", C.c_int), ("min_mapq", C.c_int), ("max_nm", C.c_int), ("enable_paired", C.c_int),
                ("both_order", C.c_int), ("min_count", C.c_int), ("max_span_frac", C.c_double), ("debug", C.c_int)]


def graph_default_opts() -> GraphOpts:
    o = GraphOpts()
    lib().orc_graph_default_opts(C.byref(o))
    return o


def _concat(strings):
    bs = [s.encode() if isinstance(s, str) else s for s in strings]
    off = np.zeros(len(bs) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    return np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy(), off


class GraphInput:
    """BAM-level records marshalled into the arrays orc_graph_run reads (done once, outside any timing)."""

    def __init__(self, records, targets):
        from palace_amd.synth import parse_cigar
        n = len(records)
        self.n, self.n_targets = n, len(targets)
        flag = np.array([r.flag for r in records], dtype=np.uint16)
        tid = np.array([r.tid for r in records], dtype=np.int32)
        pos = np.array([r.pos for r in records], dtype=np.int32)
        mtid = np.array([r.mtid for r in records], dtype=np.int32)
        mpos = np.array([r.mpos for r in records], dtype=np.int32)
        mapq = np.array([r.mapq for r in records], dtype=np.uint8)
        nm = np.array([0 if r.nm is None else r.nm for r in records], dtype=np.int32)
        cig = [np.array([(ln << 4) | op for ln, op in parse_cigar(r.cigar)], dtype=np.uint32) for r in records]
        cigar_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(c) for c in cig], out=cigar_off[1:])
        cigar = np.concatenate(cig) if n else np.zeros(0, np.uint32)
        cigar = np.ascontiguousarray(np.append(cigar, np.uint32(0)))
        qn, qoff = _concat([r.qname for r in records])
        sa, saoff = _concat([r.sa or "" for r in records])
        has_sa = np.array([r.sa is not None for r in records], dtype=np.uint8)
        self.tn, self.toff = _concat([t[0] for t in targets])
        self.tlen = np.array([t[1] for t in targets], dtype=np.int32)
        self._keep = [flag, tid, pos, mtid, mpos, mapq, nm, cigar_off, cigar, qn, qoff, sa, saoff, has_sa]
        self.R = _OrcRecords(n, *(a.ctypes.data for a in (flag, tid, pos, mtid, mpos, mapq, nm, cigar_off, cigar, qoff,
                                                          qn, saoff, sa, has_sa)))

    @classmethod
    def from_columns(cls, col, sa_off, sa, names, lens):
        """The same records from DECODED COLUMNS (what bench.py holds in HBM and palace_amd/bin/synthbam writes as a BAM), built with
        numpy so that millions of records marshal in seconds: record i is qname "q<qkey & 2^48-1 in hex>", CIGAR <ref_len>M[<clip_e>S]
        (150M when nothing is clipped), NM, and for a record with SA items the tag text "<name>,<pos>,<+|->,<clip_s>S<len-clip_s>M,
        <mapq>,<nm>;" per item -- character for character what synthbam puts into the file (palace_amd/host/synthbam_main.cpp:98-130).
        col: dict of arrays tid pos mtid mpos nm ref_len clip_e flag mapq qkey; sa_off: n+1; sa: rows of 8 int32
        (tid2 pos2 mapq2 nm2 clip_s2 clip_e2 len2 rev2)."""
        self = cls.__new__(cls)
        n = len(col["tid"])
        self.n, self.n_targets = n, len(names)
        i32 = lambda k: np.ascontiguousarray(col[k], dtype=np.int32)
        flag = np.ascontiguousarray(np.asarray(col["flag"]).view(np.uint16) if np.asarray(col["flag"]).dtype.itemsize == 2 else col["flag"], dtype=np.uint16)
        tid, pos, mtid, mpos, nm = i32("tid"), i32("pos"), i32("mtid"), i32("mpos"), i32("nm")
        mapq = np.ascontiguousarray(col["mapq"], dtype=np.uint8)
        ref_len, clip_e = i32("ref_len").astype(np.uint32), i32("clip_e").astype(np.uint32)
        clipped = clip_e != 0
        cigar_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(1 + clipped.astype(np.int64), out=cigar_off[1:])
        cigar = np.zeros(int(cigar_off[-1]) + 1, dtype=np.uint32)
        cigar[cigar_off[:-1]] = np.where(clipped, ref_len << np.uint32(4), np.uint32(150 << 4))          # M = 0
        cigar[cigar_off[:-1][clipped] + 1] = (clip_e[clipped] << np.uint32(4)) | np.uint32(4)            # S = 4
        # qname: "q" + hex of the low 48 bits without leading zeros ("q0" for zero), as printf("%llx")
        q = np.asarray(col["qkey"]).view(np.uint64) & np.uint64(0xffffffffffff)
        nib = ((q[:, None] >> (np.arange(11, -1, -1, dtype=np.uint64) * np.uint64(4))[None, :]) & np.uint64(15)).astype(np.uint8)
        lead = np.minimum((np.cumsum(nib != 0, axis=1) == 0).sum(axis=1), 11)                            # leading zero nibbles (keep one digit)
        qlen = 13 - lead                                                                                   # 'q' + digits
        qoff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(qlen, out=qoff[1:])
        qn = np.zeros(int(qoff[-1]) + 1, dtype=np.uint8)
        qn[qoff[:-1]] = ord("q")
        hexch = np.frombuffer(b"0123456789abcdef", dtype=np.uint8)[nib]
        keep = np.arange(12)[None, :] >= lead[:, None]
        dst = (qoff[:-1] + 1 - lead)[:, None] + np.arange(12)[None, :]
        qn[dst[keep]] = hexch[keep]
        del nib, hexch, keep, dst
        # SA text: only the few per cent of records that carry items take the Python loop
        so = np.asarray(sa_off, dtype=np.int64)
        cnt = so[1:] - so[:-1]
        has_sa = (cnt > 0).astype(np.uint8)
        rows = np.asarray(sa).reshape(-1, 8)
        texts = [b""] * n
        for i in np.flatnonzero(cnt > 0).tolist():
            texts[i] = "".join(f"{names[it[0]]},{it[1]},{'-' if it[7] else '+'},{it[4]}S{it[6] - it[4]}M,{it[2]},{it[3]};"
                               for it in rows[so[i]:so[i + 1]].tolist()).encode()
        saoff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.fromiter((len(t) for t in texts), dtype=np.int64, count=n), out=saoff[1:])
        sa_txt = np.frombuffer(b"".join(texts) + b"\0", dtype=np.uint8).copy()
        self.tn, self.toff = _concat(names)
        self.tlen = np.ascontiguousarray(lens, dtype=np.int32)
        self._keep = [flag, tid, pos, mtid, mpos, mapq, nm, cigar_off, cigar, qn, qoff, sa_txt, saoff, has_sa]
        self.R = _OrcRecords(n, *(a.ctypes.data for a in (flag, tid, pos, mtid, mpos, mapq, nm, cigar_off, cigar, qoff,
                                                          qn, saoff, sa_txt, has_sa)))
        return self

    def run(self, fastg_fai: str, avg_depth: float, opts: GraphOpts | None = None) -> bytes:
        o = opts or graph_default_opts()
        cap = 64 * 1024 * 1024 + 200 * self.n_targets
        buf = C.create_string_buffer(cap)
        L = lib()
        L.orc_graph_run.restype = C.c_long
        L.orc_graph_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_double,
                                    C.c_void_p, C.c_char_p, C.c_size_t]
        got = L.orc_graph_run(C.byref(self.R), self.n_targets, self.tn.ctypes.data, self.toff.ctypes.data,
                              self.tlen.ctypes.data, fastg_fai.encode(), avg_depth, C.byref(o), buf, cap)
        if got < 0:
            raise OSError("orc_graph_run: output buffer too small")
        return buf.raw[:got]


def graph_trace(records, targets, fastg_fai: str, avg_depth: float, opts: GraphOpts | None = None):
    """-> (graph text, the text `generateGraph --debug` writes to stderr on the way: generate_graph.cpp:454-458, :607-609, :711-853)"""
    gin = GraphInput(records, targets)
    o = opts or graph_default_opts()
    cap = 64 * 1024 * 1024 + 200 * gin.n_targets
    buf, tbuf = C.create_string_buffer(cap), C.create_string_buffer(cap)
    tlen = C.c_long()
    L = lib()
    L.orc_graph_run_trace.restype = C.c_long
    L.orc_graph_run_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_double,
                                      C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_long)]
    got = L.orc_graph_run_trace(C.byref(gin.R), gin.n_targets, gin.tn.ctypes.data, gin.toff.ctypes.data, gin.tlen.ctypes.data,
                                fastg_fai.encode(), avg_depth, C.byref(o), buf, cap, tbuf, cap, C.byref(tlen))
    if got < 0:
        raise OSError("orc_graph_run_trace: output buffer too small")
    return buf.raw[:got], tbuf.raw[:tlen.value]


def depth_mean(records, targets):
    """-> (text awk would print, sum, NR) for `samtools depth | awk '{sum+=$3} END {print sum/NR}'` (palace:538-552)."""
    gin = GraphInput(records, targets)
    L = lib()
    L.orc_depth_mean.restype = C.c_long
    L.orc_depth_mean.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    buf = C.create_string_buffer(64)
    s, nr = C.c_uint64(), C.c_uint64()
    n = L.orc_depth_mean(C.byref(gin.R), gin.n_targets, gin.tlen.ctypes.data, buf, 64, C.byref(s), C.byref(nr))
    return (None if n < 0 else buf.raw[:n].decode()), s.value, nr.value


def graph_run(records, targets, fastg_fai: str, avg_depth: float, opts: GraphOpts | None = None) -> bytes:
    """records: list of palace_amd.synth.BamRecord (file order); targets: [(name, len)]."""
    return GraphInput(records, targets).run(fastg_fai, avg_depth, opts)


# ---- matching (oracle/match_oracle.cpp: this repository's own algorithm, reference absent) ------
def match_run(graph_path: str, paths_path: str | None, iterations: int = 10, self_loops: bool = False,
              break_cycles: bool = False, aggressive: bool = False, cap: int = 32 * 1024 * 1024):
    """-> (linear_bytes, cycle_bytes)"""
    L = lib()
    L.orc_match_run.restype = C.c_long
    L.orc_match_run.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                C.c_char_p, C.c_size_t, C.POINTER(C.c_long)]
    lin, cyc = C.create_string_buffer(cap), C.create_string_buffer(cap)
    n_cyc = C.c_long(0)
    n = L.orc_match_run(graph_path.encode(), (paths_path or "").encode(), iterations, int(self_loops),
                        int(break_cycles), int(aggressive), lin, cap, cyc, cap, C.byref(n_cyc))
    if n < 0:
        raise OSError("orc_match_run: output too large")
    return lin.raw[:n], cyc.raw[:n_cyc.value]
