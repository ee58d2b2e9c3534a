"""BGZF / BAM fixtures written byte by byte from the SAM specification (sections 4.1, 4.2), independently of
palace_amd.synth.write_bam, and fed to the host reader of the drop-in generateGraph (palace_amd/host/bam.cpp) through
`hostdump`: auxiliary arrays, the CG long-CIGAR tag, files without an EOF block, empty blocks, records that span blocks --
and files that are damaged: the reader must stop or fail like htslib's readers do (generate_graph.cpp:611-622, 644), never
read outside its buffers.  The same cases run a second time under AddressSanitizer + UBSan (CPU build of hostdump)."""
import os
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "palace_amd", "host")
HOSTDUMP = os.path.join(ROOT, "palace_amd", "bin", "hostdump")
HOSTDUMP_ASAN = os.path.join(ROOT, "palace_amd", "bin", "hostdump_asan")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-C", HOST, os.path.join("..", "bin", "hostdump"), os.path.join("..", "bin", "hostdump_asan")],
                   check=True, stdout=subprocess.DEVNULL)


def run(tool, path, threads="3"):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([tool, "bam", path, threads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert b"AddressSanitizer" not in p.stderr and b"runtime error" not in p.stderr, p.stderr.decode()[:2000]
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def both(path):
    a = run(HOSTDUMP, path)
    b = run(HOSTDUMP_ASAN, path)
    assert a[:2] == b[:2]
    return a


# ---- the specification, restated ---------------------------------------------------------------------------------
def bgzf_member(data: bytes, extra_subfields: bytes = b"", level: int = 6) -> bytes:
    """SAM spec 4.1: gzip member, FLG.FEXTRA set, extra subfield BC (SLEN 2) = total member size - 1."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = co.compress(data) + co.flush()
    xlen = 6 + len(extra_subfields)
    bsize = 12 + xlen + len(cdata) + 8 - 1
    assert bsize < 65536
    head = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, xlen)
    return (head + extra_subfields + b"BC" + struct.pack("<HH", 2, bsize) + cdata +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


EOF_MEMBER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
OPS = "MIDNSHP=X"


def cigar_words(text):
    out, n = [], 0
    for ch in text:
        if ch.isdigit():
            n = n * 10 + int(ch)
        else:
            out.append((n << 4) | OPS.index(ch))
            n = 0
    return out


def record(qname, flag, tid, pos, mapq, cigar, mtid=-1, mpos=-1, l_seq=None, aux=b"", name_extra_nul=0):
    """SAM spec 4.2: block_size, refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, next_refID, next_pos, tlen,
    read_name (NUL terminated), cigar, seq (4 bit), qual, aux."""
    ops = cigar_words(cigar) if isinstance(cigar, str) else cigar
    if l_seq is None:
        l_seq = sum(w >> 4 for w in ops if (w & 15) in (0, 1, 4, 7, 8))
    name = qname.encode() + b"\0" * (1 + name_extra_nul)
    body = struct.pack("<iiBBHHHIiii", tid, pos, len(name), mapq, 4680, len(ops), flag, l_seq, mtid, mpos, 0)
    body += name + b"".join(struct.pack("<I", w) for w in ops) + b"\x11" * ((l_seq + 1) // 2) + b"\x28" * l_seq + aux
    return struct.pack("<I", len(body)) + body


def header(targets, text="@HD\tVN:1.6\tSO:coordinate\n"):
    raw = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(targets))
    for n, l in targets:
        raw += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    return raw


def aux_i(tag, v): return tag.encode() + b"i" + struct.pack("<i", v)
def aux_C(tag, v): return tag.encode() + b"C" + struct.pack("<B", v)
def aux_Z(tag, s): return tag.encode() + b"Z" + s.encode() + b"\0"
def aux_A(tag, c): return tag.encode() + b"A" + c.encode()
def aux_f(tag, v): return tag.encode() + b"f" + struct.pack("<f", v)
def aux_H(tag, s): return tag.encode() + b"H" + s.encode() + b"\0"


def aux_B(tag, sub, vals):
    fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
    return tag.encode() + b"B" + sub.encode() + struct.pack("<i", len(vals)) + b"".join(struct.pack("<" + fmt, v) for v in vals)


TARGETS = [("EDGE_1_length_5000_cov_3.5", 5000), ("EDGE_2_length_900_cov_7.25", 900), ("EDGE_3_length_70000_cov_1.0", 70000)]


def fields(line):
    return line.split("\t")


def write(path, members):
    with open(path, "wb") as f:
        for m in members:
            f.write(m)


# ---- well-formed files -------------------------------------------------------------------------------------------
def test_aux_arrays_cg_tag_block_spanning_and_no_eof(tmp_path):
    n1, n2, n3 = (t[0] for t in TARGETS)
    long_ops = cigar_words("10S") + cigar_words("1M1D" * 40) + cigar_words("50M5S")       # the real CIGAR of r_cg
    l_seq_cg = 10 + 40 + 50 + 5
    ref_span = 80 + 50
    recs = [
        # every aux type in front of NM / SA, incl. B arrays of each subtype
        record("r_aux", 0, 0, 100, 60, "20S80M", aux=aux_A("XA", "q") + aux_B("ZB", "c", [-1, 2]) + aux_B("ZC", "S", [1, 65535, 3]) +
               aux_B("ZD", "f", [1.5]) + aux_B("ZE", "I", []) + aux_f("XF", 2.5) + aux_H("XH", "1AE3") + aux_C("NM", 4) +
               aux_Z("SA", f"{n2},15,-,30S70M,40,2;") + aux_i("NM", 99)),
        # CG:B,I long CIGAR behind the <l_seq>S<ref>N placeholder (SAM spec 4.2.2); htslib swaps it in
        record("r_cg", 0, 2, 7, 30, [(l_seq_cg << 4) | 4, (ref_span << 4) | 3], l_seq=l_seq_cg,
               aux=aux_i("NM", 1) + aux_B("CG", "I", long_ops)),
        # same placeholder shape but no CG tag: stays what it is
        record("r_fake", 0, 2, 9, 30, [(50 << 4) | 4, (60 << 4) | 3], l_seq=50),
        # name with extra NULs (l_read_name counts them), unmapped mate fields, no aux at all
        record("r_nul", 99, 1, 0, 0, "100M", mtid=1, mpos=300, name_extra_nul=3),
        # no CIGAR
        record("r_nocig", 4, -1, -1, 0, "", l_seq=30),
    ]
    stream = header(TARGETS) + b"".join(recs)
    # members cut at arbitrary places (records and even their block_size words span members), an empty member in the
    # middle, an unknown extra subfield before BC, and NO EOF marker
    cuts = [0, 3, 90, 91, 200, 333, len(stream) - 2, len(stream)]
    members = []
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        members.append(bgzf_member(stream[a:b], extra_subfields=b"XY" + struct.pack("<H", 3) + b"abc" if i == 2 else b""))
        if i == 3:
            members.append(bgzf_member(b""))
    path = str(tmp_path / "spec.bam")
    write(path, members)
    rc, out, err = both(path)
    assert rc == 0, err
    lines = out.strip().split("\n")
    assert [fields(l)[1:] for l in lines[:3]] == [[n, str(l)] for n, l in TARGETS]
    body = {fields(l)[0]: fields(l) for l in lines[3:]}
    assert list(body) == ["r_aux", "r_cg", "r_fake", "r_nul", "r_nocig"]
    # qname flag tid pos mapq mtid mpos nm ref_len read_len clip_s clip_e [SA:tid2,pos2,rev2,mapq2,nm2,clip_s2,clip_e2,len2]
    assert body["r_aux"][1:] == ["0", "0", "100", "60", "-1", "-1", "4", "80", "100", "20", "0", "SA:1,15,1,40,2,30,0,100"]
    assert body["r_cg"][1:] == ["0", "2", "7", "30", "-1", "-1", "1", str(ref_span), str(l_seq_cg), "10", "5"]
    assert body["r_fake"][1:] == ["0", "2", "9", "30", "-1", "-1", "0", "60", "50", "50", "0"]
    assert body["r_nul"][1:] == ["99", "1", "0", "0", "1", "300", "0", "100", "100", "0", "0"]
    assert body["r_nocig"][1:] == ["4", "-1", "-1", "0", "-1", "-1", "0", "0", "0", "-1", "0"]
    # with the EOF marker the result is the same
    write(path, members + [EOF_MEMBER])
    assert both(path)[1] == out


def test_duplicate_and_unsorted_sq_lines(tmp_path):
    targets = [("b", 10), ("a", 20), ("b", 30)]                     # name_to_tid: the last duplicate wins (:624-627)
    recs = [record("q", 0, 1, 1, 60, "10M", aux=aux_Z("SA", "b,1,+,10M,60,0;"))]
    path = str(tmp_path / "dup.bam")
    write(path, [bgzf_member(header(targets) + b"".join(recs)), EOF_MEMBER])
    rc, out, _ = both(path)
    assert rc == 0
    assert fields(out.strip().split("\n")[-1])[-1].startswith("SA:2,")


# ---- damaged files -----------------------------------------------------------------------------------------------
def good_file():
    recs = [record(f"r{i}", 0, 0, 10 * i, 60, "50M", aux=aux_C("NM", i)) for i in range(40)]
    stream = header(TARGETS) + b"".join(recs)
    return [bgzf_member(stream[a:a + 700]) for a in range(0, len(stream), 700)], recs


def test_truncated_and_corrupt_bgzf(tmp_path):
    members, _ = good_file()
    whole = b"".join(members)
    path = str(tmp_path / "bad.bam")
    # cut inside a member: its BSIZE points past the end of the file
    open(path, "wb").write(whole[:len(members[0]) + len(members[1]) // 2])
    rc, out, err = both(path)
    assert rc == 1 and "BGZF" in err
    # fewer than 18 bytes left after the last whole member: ignored, like a missing EOF marker
    open(path, "wb").write(whole + b"\x1f\x8b\x08")
    assert both(path)[0] == 0

    def patched(off, data):
        b = bytearray(whole)
        b[off:off + len(data)] = data
        open(path, "wb").write(bytes(b))
        return both(path)

    assert patched(0, b"\x00")[0] == 1                                     # magic
    assert patched(10, struct.pack("<H", 60000))[0] == 1                   # XLEN runs past the member / the file
    assert patched(14, struct.pack("<H", 200))[0] == 1                     # SLEN runs past XLEN
    assert patched(16, struct.pack("<H", 5))[0] == 1                       # BSIZE smaller than header + trailer
    assert patched(16, struct.pack("<H", 65535))[0] == 1                   # BSIZE past the end of the file
    isize_at = len(members[0]) - 4
    assert patched(isize_at, struct.pack("<I", 0x7fffffff))[0] == 1        # ISIZE must not size a 2 GiB buffer
    assert patched(isize_at, struct.pack("<I", 5))[0] == 1                 # ISIZE smaller than the data: inflate fails
    assert patched(30, b"\xff\xff\xff\xff")[0] == 1                        # garbage in the deflate stream
    # not a BAM inside
    open(path, "wb").write(bgzf_member(b"SAM\1" + b"\0" * 40))
    assert both(path)[0] == 1
    # header that claims more references / text than the stream holds
    open(path, "wb").write(bgzf_member(b"BAM\1" + struct.pack("<i", 1 << 30)))
    assert both(path)[0] == 1
    open(path, "wb").write(bgzf_member(header(TARGETS)[:-10]))
    assert both(path)[0] == 1


def test_malformed_records_end_the_stream_like_sam_read1(tmp_path):
    """htslib's bam_read1 fails on a record whose variable-length fields do not fit its block_size, which ends the
    reference's read loop (generate_graph.cpp:644): everything before that record is kept, nothing behind it is read."""
    path = str(tmp_path / "rec.bam")
    good = [record(f"g{i}", 0, 0, i, 60, "50M", aux=aux_C("NM", 1)) for i in range(3)]
    tail = record("after", 0, 0, 9, 60, "50M")

    def n_records(bad):
        write(path, [bgzf_member(header(TARGETS) + b"".join(good) + bad + tail), EOF_MEMBER])
        rc, out, err = both(path)
        assert rc == 0, err
        return len(out.strip().split("\n")) - len(TARGETS)

    ok = record("x", 0, 0, 5, 60, "50M")

    def poke(rec, off, fmt, val):
        b = bytearray(rec)
        b[4 + off:4 + off + struct.calcsize(fmt)] = struct.pack(fmt, val)
        return bytes(b)

    assert n_records(ok) == 5
    assert n_records(poke(ok, 8, "<B", 200)) == 3                          # l_read_name past the record
    assert n_records(poke(ok, 8, "<B", 0)) == 3                            # l_read_name 0
    assert n_records(poke(ok, 12, "<H", 60000)) == 3                       # n_cigar_op past the record
    assert n_records(poke(ok, 16, "<I", 1 << 20)) == 3                     # l_seq past the record
    assert n_records(poke(ok, 16, "<I", 0xfffffff0)) == 3                  # l_seq negative as int32
    assert n_records(struct.pack("<I", 20) + b"\0" * 20) == 3              # block_size < 32
    assert n_records(struct.pack("<I", 1 << 28) + ok[4:]) == 3             # block_size past the stream
    # aux damage does not end the stream: the walk over the aux fields stops, the record itself is delivered
    trunc_z = record("z", 0, 0, 5, 60, "50M", aux=b"SAZ" + b"no terminator")
    assert n_records(trunc_z) == 5
    big_b = record("b", 0, 0, 5, 60, "50M", aux=b"ZBBi" + struct.pack("<i", 0x7fffffff) + aux_C("NM", 3))
    assert n_records(big_b) == 5
    bad_sub = record("b", 0, 0, 5, 60, "50M", aux=b"ZBB?" + struct.pack("<i", 1) + b"\0" * 4)
    assert n_records(bad_sub) == 5


# ---- the record walk ahead of the walker (round 6: the inflate threads walk 1 MiB segments speculatively) ----------------------
def run_mode(path, threads, serial):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    if serial:
        env["PALACE_BAM_SERIAL_WALK"] = "1"
    p = subprocess.run([HOSTDUMP, "bam", path, str(threads)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    return p.returncode, p.stdout, p.stderr.decode()[-300:]


def test_the_walk_ahead_gives_the_serial_walk_on_streams_full_of_fake_record_chains(tmp_path):
    """A ~9 MB stream (nine segments) in which most records carry an auxiliary byte array that READS like five well-formed records in a
    row -- what the speculation's entry search takes for a record start --, two records larger than a segment, records of every size in
    between; the same file cut short at member boundaries and inside a record.  The loader with the segments walked ahead must print what
    the walker alone prints (same records, same columns, same SA items, same exit status), for 1, 3 and 8 threads."""
    import random
    rnd = random.Random(11)
    fake_one = lambda i: (lambda body: struct.pack("<I", len(body)) + body)(
        struct.pack("<iiBBHHHIiii", i % 3, 10 * i, 2, 30, 4680, 0, 0, 0, -1, -1, 0) + b"x\0")
    fake = b"".join(fake_one(i) for i in range(5))
    recs = []
    pos = 0
    for i in range(21000):
        tid = 0 if i < 9000 else 1 if i < 12000 else 2
        if i in (9000, 12000):
            pos = 0
        pos += rnd.randrange(0, 3)
        L = rnd.choice((36, 100, 150, 150, 150, 251, 1000))
        if i in (4000, 15000):
            L = 800_000                                                  # 1.2 MB records: longer than a segment
        aux = aux_C("NM", i % 7)
        if i % 4:
            aux += aux_B("ZF", "C", list(fake))
        if i % 50 == 0:
            aux += aux_Z("SA", f"{TARGETS[2][0]},{100 + i},+,40M{L - 40}S,60,1;")
        recs.append(record(f"r{i}", 99 if i % 2 else 147, tid, min(pos, TARGETS[tid][1] - 1), 60, f"{L}M", mtid=(tid + i % 2) % 3, mpos=5, aux=aux))
    raw = header(TARGETS) + b"".join(recs)
    assert len(raw) > 8 * (1 << 20)
    members = [bgzf_member(raw[i:i + 60000], level=1) for i in range(0, len(raw), 60000)]
    full = str(tmp_path / "fake_chains.bam")
    write(full, members + [EOF_MEMBER])
    want = run_mode(full, 3, True)
    assert want[0] == 0 and want[1].count(b"\n") == len(recs) + len(TARGETS)
    for threads in (1, 3, 8):
        assert run_mode(full, threads, False) == want, threads
    tr = subprocess.run([HOSTDUMP, "bamtime", full, "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, PALACE_TRACE="1"), timeout=300)
    import re
    m = re.search(r"(\d+) of (\d+) record boundaries came from the segments walked ahead", tr.stderr.decode())
    assert m and int(m.group(2)) == len(recs) and int(m.group(1)) > 0, tr.stderr.decode()[-500:]      # (the speculation did take part)
    asan = run(HOSTDUMP_ASAN, full, "4")                                    # (and under AddressSanitizer + UBSan)
    assert asan[0] == 0 and asan[1].encode() == want[1]
    # cut short: whole members missing (no EOF block), and the stream ending inside a record
    for k, cut in enumerate((len(members) // 3, len(members) - 2)):
        part = str(tmp_path / f"cut{k}.bam")
        write(part, members[:cut])
        a = run_mode(part, 3, True)
        assert a[1].count(b"\n") < want[1].count(b"\n")
        for threads in (2, 8):
            assert run_mode(part, threads, False) == a, (k, threads)
