"""Seeded synthetic inputs for the conjugate-graph hot path.

Build code, not reference code: nothing here is derived from /root/reference.  The generator
produces every file the three executables consume:

  phage DB FASTA                      -> eref argv[3]          (extract_ref.cpp:1221)
  paired 4-line FASTQ                 -> eref argv[1], argv[2] (extract_ref.cpp:1222-1223)
  BAM (BGZF, coordinate sorted)       -> generateGraph <BAM>   (generate_graph.cpp:600)
  assembly_graph.fastg.fai            -> generateGraph <FASTG_FAI> (generate_graph.cpp:601)
  assembly_graph.fasta.fai, .blast, hit_seqs.out, node_scores.out, contigs.paths
                                      -> filter_graph.py argv  (filter_graph.py:8-20)

Everything is driven by numpy's PCG64 so a (seed, sizes) pair names a workload exactly.
"""
from __future__ import annotations

import struct
import zlib
from dataclasses import dataclass, field

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
    _COMP[_a] = _b


def rng_for(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def random_dna(rng: np.random.Generator, n: int) -> np.ndarray:
    return ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def mutate(rng: np.random.Generator, seq: np.ndarray, rate: float) -> np.ndarray:
    """Substitute each base with probability `rate` by a different base."""
    out = seq.copy()
    if rate <= 0 or len(seq) == 0:
        return out
    hit = np.nonzero(rng.random(len(seq)) < rate)[0]
    for p in hit:
        cur = out[p]
        alt = ACGT[rng.integers(0, 4)]
        while alt == cur:
            alt = ACGT[rng.integers(0, 4)]
        out[p] = alt
    return out


# --------------------------------------------------------------------------------------
# phage DB + reads (eref inputs)
# --------------------------------------------------------------------------------------
@dataclass
class PhageDB:
    names: list            # FASTA header text after '>' (may contain spaces / '/')
    seqs: list             # list of np.uint8 ASCII arrays

    def write_fasta(self, path: str, width: int = 80) -> None:
        with open(path, "wb") as f:
            for name, seq in zip(self.names, self.seqs):
                f.write(b">" + name.encode() + b"\n")
                b = seq.tobytes()
                for i in range(0, len(b), width):
                    f.write(b[i:i + width] + b"\n")


def make_phage_db(rng, n_refs: int, len_lo: int, len_hi: int, name_prefix: str = "phage") -> PhageDB:
    names, seqs = [], []
    for i in range(n_refs):
        L = int(rng.integers(len_lo, len_hi + 1))
        names.append(f"{name_prefix}_{i + 1}|len{L} synthetic")
        seqs.append(random_dna(rng, L))
    return PhageDB(names, seqs)


@dataclass
class ReadSet:
    """Concatenated read bases (ASCII) with offsets; reads[i] = bases[off[i]:off[i+1]]."""
    bases: np.ndarray
    offsets: np.ndarray
    names: list = field(default_factory=list)

    @property
    def n(self) -> int:
        return len(self.offsets) - 1

    def read(self, i: int) -> np.ndarray:
        return self.bases[self.offsets[i]:self.offsets[i + 1]]

    def write_fastq(self, path: str, tag: str) -> None:
        with open(path, "wb") as f:
            for i in range(self.n):
                s = self.read(i).tobytes()
                nm = self.names[i] if self.names else f"r{i}"
                f.write(b"@" + nm.encode() + b"/" + tag.encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")


def reads_from_list(seqs: list, names: list | None = None) -> ReadSet:
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    bases = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
    return ReadSet(bases.astype(np.uint8), off, names or [])


def sample_pairs(rng, template: np.ndarray, n_pairs: int, read_len: int, insert_mu: float,
                 insert_sd: float, err: float, lo: int = 0, hi: int | None = None):
    """Sample FR read pairs whose fragment lies inside template[lo:hi]."""
    hi = len(template) if hi is None else hi
    r1, r2 = [], []
    for _ in range(n_pairs):
        ins = int(max(read_len, min(hi - lo, rng.normal(insert_mu, insert_sd))))
        st = int(rng.integers(lo, max(lo + 1, hi - ins + 1)))
        frag = template[st:st + ins]
        a = frag[:read_len]
        b = revcomp(frag[-read_len:])
        if rng.random() < 0.5:                       # fragment from the other strand
            a, b = b, a
        r1.append(mutate(rng, a, err))
        r2.append(mutate(rng, b, err))
    return r1, r2


def vector_reads(rng, pool: np.ndarray, n: int, read_len: int) -> ReadSet:
    """Fast path for big workloads: n fixed-length reads cut from a random pool (vectorised)."""
    starts = rng.integers(0, len(pool) - read_len, size=n)
    idx = starts[:, None] + np.arange(read_len)[None, :]
    bases = pool[idx].reshape(-1)
    off = np.arange(n + 1, dtype=np.int64) * read_len
    return ReadSet(bases, off)


# --------------------------------------------------------------------------------------
# BAM writer (BGZF over zlib) -- SAM spec v1 section 4; test/bench infrastructure only
# --------------------------------------------------------------------------------------
_CIGAR_OPS = "MIDNSHP=X"


def parse_cigar(text: str):
    ops, n = [], 0
    for ch in text:
        if ch.isdigit():
            n = n * 10 + ord(ch) - 48
        else:
            ops.append((n, _CIGAR_OPS.index(ch)))
            n = 0
    return ops


def _reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


@dataclass
class BamRecord:
    qname: str
    flag: int
    tid: int
    pos: int            # 0-based
    mapq: int
    cigar: str
    mtid: int = -1
    mpos: int = -1
    tlen: int = 0
    nm: int | None = 0
    sa: str | None = None
    nm_type: str = "C"  # aux integer type used for NM: c C s S i I

    def encode(self) -> bytes:
        ops = parse_cigar(self.cigar)
        qlen = sum(n for n, op in ops if op in (0, 1, 4, 7, 8))
        rlen = sum(n for n, op in ops if op in (0, 2, 3, 7, 8))
        name = self.qname.encode() + b"\0"
        bin_ = _reg2bin(max(self.pos, 0), max(self.pos, 0) + max(rlen, 1))
        body = struct.pack("<iiBBHHHiiii", self.tid, self.pos, len(name), self.mapq, bin_, len(ops),
                           self.flag, qlen, self.mtid, self.mpos, self.tlen)
        body += name
        body += b"".join(struct.pack("<I", (n << 4) | op) for n, op in ops)
        body += b"\x11" * ((qlen + 1) // 2)          # sequence: all 'A' (=1) nibbles
        body += b"\xff" * qlen                        # qualities absent
        # an unrelated tag first, so aux scanning has to skip something
        body += b"ASC" + struct.pack("<B", 30)
        if self.nm is not None:
            fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[self.nm_type]
            body += b"NM" + self.nm_type.encode() + struct.pack(fmt, self.nm)
        body += b"XSi" + struct.pack("<i", -7)
        if self.sa is not None:
            body += b"SAZ" + self.sa.encode() + b"\0"
        return struct.pack("<i", len(body)) + body


def _bgzf_block(data: bytes, level: int = 1) -> bytes:
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def write_bam(path: str, targets: list, records: list, block: int = 0xFF00, level: int = 1) -> None:
    """targets: [(name, length)], records: iterable of BamRecord (caller sorts)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in targets)
    raw = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(targets)))
    for n, l in targets:
        nb = n.encode() + b"\0"
        raw += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    for r in records:
        raw += r.encode()
    with open(path, "wb") as f:
        for i in range(0, len(raw), block):
            f.write(_bgzf_block(bytes(raw[i:i + block]), level))
        f.write(_BGZF_EOF)


# --------------------------------------------------------------------------------------
# contigs / graph-side inputs
# --------------------------------------------------------------------------------------
def contig_names(rng, n: int, median: float = 800.0, sigma: float = 1.0, min_len: int = 56,
                 long_mode: bool = False):
    """EDGE_<id>_length_<len>_cov_<cov> names (the naming split_fastg.py:55-66 produces)."""
    if long_mode:      # N50 ~ 50 kb with a tail > 120 kb (exercises the G5 underflow gate)
        lens = np.maximum(min_len, rng.lognormal(np.log(30000.0), 0.9, size=n)).astype(np.int64)
    else:
        lens = np.maximum(min_len, rng.lognormal(np.log(median), sigma, size=n)).astype(np.int64)
    ids = rng.permutation(np.arange(1, 4 * n + 1))[:n]           # non-contiguous ids, random order
    covs = rng.gamma(2.0, 8.0, size=n)
    names = [f"EDGE_{int(i)}_length_{int(l)}_cov_{c:.6f}" for i, l, c in zip(ids, lens, covs)]
    return names, lens


def fastg_fai_lines(rng, names: list, lens, links_per_contig: float = 1.3):
    """Lines of assembly_graph.fastg.fai: both strands of every edge with their successors."""
    n = len(names)
    lines = []
    for strand in ("", "'"):
        for i, nm in enumerate(names):
            k = int(rng.poisson(links_per_contig))
            succ = []
            for _ in range(k):
                j = int(rng.integers(0, n))
                succ.append(names[j] + ("'" if rng.random() < 0.5 else ""))
            head = nm + strand
            col0 = head + (":" + ",".join(succ) if succ else "") + ";"
            lines.append(f"{col0}\t{int(lens[i])}\t0\t60\t61\n")
    return lines


# --------------------------------------------------------------------------------------
# random generateGraph scenario (records + targets + fastg.fai) for parity tests
# --------------------------------------------------------------------------------------
def random_graph_case(rng, n_contigs: int = 60, n_events: int = 3000, long_mode: bool = False,
                      read_len: int = 100):
    names, lens = contig_names(rng, n_contigs, median=1500.0, sigma=0.9, min_len=90, long_mode=long_mode)
    lens = [int(x) for x in lens]
    targets = list(zip(names, lens))
    fai = fastg_fai_lines(rng, names, lens, 1.3)
    recs = []
    mapqs = [0, 20, 40, 60, 60, 60]

    def end_pos(L):     # 0-based pos whose 1-based value lies in the END region (default MAX_END)
        lo = max(L - 300, L // 2)
        return int(rng.integers(lo, max(lo + 1, L - 1)))

    def start_pos(L):
        hi = min(300, L // 2)
        return int(rng.integers(0, max(1, hi)))

    def any_pos(L):
        return int(rng.integers(0, max(1, L - 1)))

    hot = [(int(rng.integers(0, n_contigs)), int(rng.integers(0, n_contigs)), int(rng.integers(0, 8)))
           for _ in range(max(4, n_contigs // 3))]
    nm_types = "cCsSiI"
    for ev in range(n_events):
        q = f"q{ev}"
        kind = rng.random()
        mq = int(mapqs[int(rng.integers(0, len(mapqs)))])
        nm = int(rng.choice([0, 0, 0, 1, 2, 5, 6, 9]))
        nmt = nm_types[int(rng.integers(0, 6))]
        if kind < 0.45:                                    # ordinary intra-contig pair
            a = int(rng.integers(0, n_contigs)); L = lens[a]
            p1 = any_pos(L); p2 = min(L - 1, p1 + int(rng.integers(0, 300)))
            cg = str(rng.choice(["100M", "50M2I48M", "50M3D50M", "30S70M", "70M30S", "10H90M", "100M"]))
            recs.append(BamRecord(q, 0x63, a, p1, mq, cg, a, p2, nm=nm, nm_type=nmt))
            recs.append(BamRecord(q, 0x93, a, p2, mq, "100M", a, p1, nm=None if rng.random() < 0.2 else nm))
        elif kind < 0.75:                                  # cross-contig pair
            if rng.random() < 0.7:
                a, b, t = hot[int(rng.integers(0, len(hot)))]
            else:
                a, b, t = int(rng.integers(0, n_contigs)), int(rng.integers(0, n_contigs)), int(rng.integers(0, 8))
            if a == b:
                b = (a + 1) % n_contigs
            rev1, rev2 = bool(t & 1), bool(t & 2)
            p1 = (start_pos if rev1 else end_pos)(lens[a]) if rng.random() < 0.85 else any_pos(lens[a])
            if t & 4:
                p2 = (end_pos if not rev2 else start_pos)(lens[b])
            else:
                p2 = (start_pos if rev2 else end_pos)(lens[b])
            if rng.random() < 0.1:
                p2 = any_pos(lens[b])
            f1 = 0x41 | (0x10 if rev1 else 0) | (0x20 if rev2 else 0)
            f2 = 0x81 | (0x10 if rev2 else 0) | (0x20 if rev1 else 0)
            mq2 = int(mapqs[int(rng.integers(0, len(mapqs)))])
            recs.append(BamRecord(q, f1, a, p1, mq, "100M", b, p2, nm=nm, nm_type=nmt))
            if rng.random() < 0.9:                         # sometimes the mate record is absent
                recs.append(BamRecord(q, f2, b, p2, mq2, "90M10S", a, p1, nm=int(rng.choice([0, 1, 7]))))
        elif kind < 0.95:                                  # split read with SA tag(s)
            if rng.random() < 0.7:
                a, b, t = hot[int(rng.integers(0, len(hot)))]
            else:
                a, b, t = int(rng.integers(0, n_contigs)), int(rng.integers(0, n_contigs)), int(rng.integers(0, 8))
            k = int(rng.integers(25, 76)); m = read_len - k
            rev1, rev2 = bool(t & 1), bool(t & 2)
            # primary covers read bases [1,k] (left part), SA covers [k+1, read_len]
            cg1 = f"{m}S{k}M" if rev1 else f"{k}M{m}S"
            cg2 = f"{m}M{k}S" if rev2 else f"{k}S{m}M"
            if t & 4:                                      # swap roles: primary is the right part
                cg1 = f"{k}M{m}S" if rev1 else f"{m}S{k}M"
                cg2 = f"{k}S{m}M" if rev2 else f"{m}M{k}S"
            left_a = not (t & 4)
            pa = ((start_pos if rev1 else end_pos) if left_a else (end_pos if rev1 else start_pos))(lens[a])
            pb = ((end_pos if rev2 else start_pos) if left_a else (start_pos if rev2 else end_pos))(lens[b])
            if rng.random() < 0.1:
                pa = any_pos(lens[a])
            items = [f"{names[b]},{pb + 1},{'-' if rev2 else '+'},{cg2},{int(mapqs[int(rng.integers(0, 6))])},{int(rng.choice([0, 1, 6]))}"]
            r = rng.random()
            if r < 0.08:
                items.insert(0, f"{names[a]},5,+,{cg2},60,0")                 # same contig as primary
            elif r < 0.16:
                items.insert(0, "nosuchcontig,5,+,50S50M,60,0")               # unknown name
            elif r < 0.22:
                items.insert(0, f"{names[b]},7,+,50S50M,60")                  # 5 fields only
            elif r < 0.28:
                items.insert(0, f" {names[b]} , {pb + 1} ,{'-' if rev2 else '+'}, {cg2} , 60 , 0 ")   # blanks
            elif r < 0.32:
                items.append(f"{names[(b + 1) % n_contigs]},{start_pos(lens[(b + 1) % n_contigs]) + 1},+,{k}S{m}M,60,0")
            elif r < 0.35:
                items.insert(0, "")
            sa = ";".join(items) + (";" if rng.random() < 0.9 else "")
            flag = (0x10 if rev1 else 0)
            paired = rng.random() < 0.5
            if paired:                                     # split read that is also a cross-contig pair
                c = int(rng.integers(0, n_contigs))
                flag |= 0x41 | (0x20 if rng.random() < 0.5 else 0)
                pc = start_pos(lens[c]) if rng.random() < 0.5 else end_pos(lens[c])
                recs.append(BamRecord(q, flag, a, pa, mq, cg1, c, pc, nm=nm, sa=sa, nm_type=nmt))
                recs.append(BamRecord(q, 0x81 | (0x10 if flag & 0x20 else 0) | (0x20 if rev1 else 0), c, pc, 60, "100M", a, pa, nm=0))
            else:
                recs.append(BamRecord(q, flag, a, pa, mq, cg1, nm=nm, sa=sa, nm_type=nmt))
        else:                                              # records the reference skips outright
            a = int(rng.integers(0, n_contigs))
            f = int(rng.choice([0x4, 0x100, 0x800, 0x904]))
            recs.append(BamRecord(q, f, a if f != 0x4 else -1, any_pos(lens[a]) if f != 0x4 else -1, mq, "100M" if f != 0x4 else "", nm=nm))
    order = sorted(range(len(recs)), key=lambda i: (recs[i].tid if recs[i].tid >= 0 else 1 << 30, recs[i].pos, i))
    recs = [recs[i] for i in order]
    total = sum(sum(n for n, op in parse_cigar(r.cigar) if op in (0, 2, 3, 7, 8)) for r in recs)
    avg_depth = float(f"{total / max(1, sum(lens)):.6g}")
    return targets, "".join(fai), recs, avg_depth


# --------------------------------------------------------------------------------------
# side inputs of the graph filter (filter_graph.py argv) for a set of contigs
# --------------------------------------------------------------------------------------
def filter_side_files(rng, names, lens, ref_names=("phageA", "phageB", "phageC")):
    """-> dict of file texts: fasta_fai, blast, hit_seqs, node_scores, contigs_paths."""
    n = len(names)
    fasta_fai = "".join(f"{nm}\t{int(L)}\t{7 + i * 100}\t60\t61\n" for i, (nm, L) in enumerate(zip(names, lens)))
    # blastn outfmt 6 (palace:524-528): qseqid sseqid pident length mismatch gapopen qstart qend sstart send evalue bitscore qlen slen
    blast = []
    for i in rng.permutation(n)[: max(1, n // 8)]:
        L = int(lens[i])
        for ref in rng.permutation(len(ref_names))[: int(rng.integers(1, 3))]:
            for _ in range(int(rng.integers(1, 4))):
                ident = float(rng.choice([99.5, 85.0, 70.0, 69.9, 55.0]))
                al = int(max(30, L * rng.choice([0.1, 0.3, 0.5, 0.8])))
                blast.append(f"{names[i]}\t{ref_names[ref]}\t{ident:.3f}\t{al}\t3\t0\t1\t{al}\t100\t{100 + al}\t1e-50\t200\t{L}\t40000\n")
    hit = "".join(f"{names[i]}\t{int(rng.integers(1, 9))}\n" for i in rng.permutation(n)[: max(1, n // 12)])
    scores = []
    for i in range(n):
        r = rng.random()
        if r < 0.1:
            s = f"{rng.random() * 9:.4f}e-05"
        elif r < 0.15:
            s = "1.0"
        else:
            s = f"{rng.random():.6f}"
        scores.append(f"{names[i]}\t{s}\n")
    ids = [nm.split("_")[1] for nm in names]
    paths = []
    for k in range(max(1, n // 3)):
        m = int(rng.integers(1, 6))
        members = [int(x) for x in rng.integers(0, n, size=m)]
        fwd = [ids[j] + ("+" if rng.random() < 0.5 else "-") for j in members]
        rc = [t[:-1] + ("-" if t[-1] == "+" else "+") for t in reversed(fwd)]
        sep = ";\n" if (m > 2 and rng.random() < 0.3) else None
        for tag, toks in (("", fwd), ("'", rc)):
            paths.append(f"NODE_{k + 1}_length_{sum(int(lens[j]) for j in members)}_cov_9.5{tag}\n")
            if sep:
                paths.append(",".join(toks[:2]) + sep + ",".join(toks[2:]) + "\n")
            else:
                paths.append(",".join(toks) + "\n")
    return dict(fasta_fai=fasta_fai, blast="".join(blast), hit_seqs=hit, node_scores="".join(scores),
                contigs_paths="".join(paths))


# --------------------------------------------------------------------------------------
# BASELINE config[1]: 50k contigs / 5k-phage ref DB (eref side), vectorised so it is quick at 200 Mb
# --------------------------------------------------------------------------------------
def eref_config_inputs(seed: int = 20261003, n_refs: int = 5000, n_pairs: int = 166_666, read_len: int = 150,
                       pool_bases: int = 60_000_000, arrays: bool = False, n_present: int | None = None,
                       n_phage_pairs: int | None = None):
    """Phage DB + read pairs of the 50k-contig configuration (3.33 read pairs per contig as in SURVEY.md section 8(d):
    5e8 fq1 bases per 1M contigs): refs U[20 kb, 60 kb]; 1 % of them present, covered by a fifth of the pairs with
    0.5 % substitutions; the rest of the pairs from a random contig pool.  Returns (fasta_bytes, fq1_bytes, fq2_bytes), or
    with arrays=True also (refs: list of base arrays, r1, r2: [n_pairs, read_len] base matrices in file order).
    tools/ref_compare_eref.py and tests/golden/make_eref_50k_golden.py feed exactly these bytes to the compiled
    reference; tests regenerate them from the seed."""
    rng = rng_for(seed)
    lens = rng.integers(20000, 60001, size=n_refs)
    refs = [random_dna(rng, int(L)) for L in lens]
    fa = []
    for i, s in enumerate(refs):
        b = s.tobytes()
        fa.append(b">phage_%d synthetic\n" % (i + 1) + b"\n".join(b[k:k + 80] for k in range(0, len(b), 80)) + b"\n")
    present = rng.choice(n_refs, size=max(1, n_refs // 100) if n_present is None else n_present, replace=False)
    pool = random_dna(rng, pool_bases)
    n_ph = n_pairs // 5 if n_phage_pairs is None else n_phage_pairs
    ar = np.arange(read_len)

    def cut(src, st):
        return src[st[:, None] + ar[None, :]]

    ins = np.clip(rng.normal(400, 40, size=n_pairs), read_len, 800).astype(np.int64)
    st_pool = rng.integers(0, len(pool) - 1000, size=n_pairs - n_ph)
    r1 = [cut(pool, st_pool)]
    r2 = [cut(pool, st_pool + ins[: n_pairs - n_ph] - read_len)]
    which = rng.integers(0, len(present), size=n_ph)
    a1 = np.zeros((n_ph, read_len), dtype=np.uint8)
    a2 = np.zeros((n_ph, read_len), dtype=np.uint8)
    for k in range(len(present)):
        idx = np.nonzero(which == k)[0]
        s = refs[present[k]]
        st = rng.integers(0, len(s) - 900, size=len(idx))
        a1[idx] = cut(s, st)
        a2[idx] = cut(s, st + ins[n_pairs - n_ph:][idx] - read_len)
    for a in (a1, a2):
        m = rng.random(a.shape) < 0.005
        a[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
    r1 = np.concatenate(r1 + [a1])
    r2 = _COMP[np.concatenate(r2 + [a2])[:, ::-1]]
    perm = rng.permutation(n_pairs)
    q = b"I" * read_len
    fq = []
    r1, r2 = r1[perm], r2[perm]
    for tag, arr in ((b"1", r1), (b"2", r2)):
        fq.append(b"".join(b"@r%d/%s\n" % (i, tag) + arr[i].tobytes() + b"\n+\n" + q + b"\n" for i in range(n_pairs)))
    if arrays:
        return b"".join(fa), fq[0], fq[1], refs, r1, r2
    return b"".join(fa), fq[0], fq[1]
