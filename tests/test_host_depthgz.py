"""`bamdepth --depth-gz` (palace_amd/host/depthgz.hpp; SURVEY row N2, palace:541-545): the per-base depth text of `samtools depth`,
BGZF-compressed, and the .tbi `tabix -s 1 -b 2 -e 2` makes.  samtools / bgzip / tabix / pysam are absent from the image: the text is
checked against a per-position statement of samtools' counting rules written here, the containers against a reader written from the
specifications (tests/tabix_reader.py) and against a fixture whose index was worked out BY HAND from them."""
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as orc
from palace_amd import synth
from tests import tabix_reader as tr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAMDEPTH = os.path.join(ROOT, "palace_amd", "bin", "bamdepth")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(os.path.join(ROOT, "palace_amd", "libpalace_hip.so")):
        pytest.skip("libpalace_hip.so not built")
    subprocess.run(["make", "-C", os.path.join(ROOT, "palace_amd", "host"), os.path.join("..", "bin", "bamdepth")], check=True, stdout=subprocess.DEVNULL)


def expected_depth_text(targets, records):
    """samtools depth (>= 1.13 defaults): records with UNMAP / SECONDARY / QCFAIL / DUP skipped, M / = / X counted, D / N advance the
    reference without counting, no depth cap, positions with depth 0 not printed; contigs in header order"""
    diff = [np.zeros(l + 1, dtype=np.int64) for _, l in targets]
    for r in records:
        if r.flag & 0x704 or r.tid < 0 or r.pos < 0:
            continue
        p, L = r.pos, targets[r.tid][1]
        for n, op in synth.parse_cigar(r.cigar):
            if op in (0, 7, 8):
                a, b = min(p, L), min(p + n, L)
                diff[r.tid][a] += 1; diff[r.tid][b] -= 1
                p += n
            elif op in (2, 3):
                p += n
    lines, total, nr = [], 0, 0
    per_contig = {}
    for (name, l), d in zip(targets, diff):
        depth = np.cumsum(d[:-1])
        pos = np.flatnonzero(depth > 0)
        mine = [f"{name}\t{p + 1}\t{depth[p]}\n".encode() for p in pos.tolist()]
        if mine:
            per_contig[name] = mine
        lines += mine
        total += int(depth[pos].sum()); nr += len(pos)
    return lines, per_contig, total, nr


def run(tmp_path, targets, records, block=0xFF00):
    bam, gz = str(tmp_path / "t.bam"), str(tmp_path / "t.bam.depth.gz")
    synth.write_bam(bam, targets, records, block=block)
    p = subprocess.run([BAMDEPTH, "--depth-gz", gz, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p, gz


def test_hand_derived_index(tmp_path):
    """Two contigs, two reads; every number of the index below is worked out by hand from the format descriptions.
    text:  c1 <TAB> 11..15 <TAB> 1   = 5 lines of 8 bytes  -> text offsets [0, 40)
           c2 <TAB> 16381..16390 <TAB> 1 = 10 lines of 11 bytes -> [40, 150); their 0-based positions 16380..16389 fall into the 16 kb
           windows 0 (16380..16383: 4 lines, [40, 84)) and 1 (16384..16389: 6 lines, [84, 150))
    one BGZF member at file offset 0 holds all 150 bytes, so virtual offset = text offset."""
    targets = [("c1", 100), ("c2", 20000)]
    recs = [synth.BamRecord("r1", 0, 0, 10, 60, "5M"), synth.BamRecord("r2", 0, 1, 16380, 60, "10M")]
    p, gz = run(tmp_path, targets, recs)
    assert p.returncode == 0, p.stderr
    assert p.stdout == b"1\n"                                           # awk: 15 / 15
    data = open(gz, "rb").read()
    mem = tr.bgzf_members(data)
    assert len(mem) == 2 and mem[0][0] == 0 and mem[1][1] == b"" and data[-28:] == tr.EOF_MEMBER
    want = b"".join(b"c1\t%d\t1\n" % q for q in range(11, 16)) + b"".join(b"c2\t%d\t1\n" % q for q in range(16381, 16391))
    assert mem[0][1] == want and len(want) == 150
    tbi = tr.read_tbi(gz + ".tbi")
    assert (tbi["format"], tbi["col_seq"], tbi["col_beg"], tbi["col_end"], tbi["meta"], tbi["skip"]) == (0, 1, 2, 2, ord("#"), 0)
    assert tbi["names"] == [b"c1", b"c2"] and tbi["n_no_coor"] == 0
    c1, c2 = tbi["refs"]
    assert c1["bins"] == {4681: [(0, 40)], 37450: [(0, 40), (5, 0)]} and c1["ioff"] == [0]
    assert c2["bins"] == {4681: [(40, 84)], 4682: [(84, 150)], 37450: [(40, 150), (10, 0)]} and c2["ioff"] == [40, 84]
    f = tr.TabixFile(gz)
    assert f.fetch("c2", 16383, 16385) == [b"c2\t16384\t1\n", b"c2\t16385\t1\n"]
    assert f.fetch("c1") == [b"c1\t%d\t1\n" % q for q in range(11, 16)]


@pytest.mark.parametrize("seed,long_mode", [(3, False), (4, True)])
def test_random_bam_text_members_and_index(tmp_path, seed, long_mode):
    rng = synth.rng_for(seed)
    targets, _, recs, _ = synth.random_graph_case(rng, 60 if long_mode else 300, 6000 if long_mode else 20000, long_mode=long_mode)
    # the flags samtools skips, deletions / skips / clips / insertions in CIGARs, reads running over the contig end
    extra = []
    for k in range(200):
        t = int(rng.integers(0, len(targets)))
        L = targets[t][1]
        cig = ["20M5D30M", "10S40M", "25M3I25M2N20M", "30=5X15M", "50M"][k % 5]
        extra.append(synth.BamRecord(f"x{k}", [0, 0x400, 0x100, 0x200, 0x4, 0x800, 16][k % 7], t, int(rng.integers(0, max(1, L - 10))), 60, cig))
    recs = sorted(recs + extra, key=lambda r: (r.tid if r.tid >= 0 else 1 << 30, r.pos))
    p, gz = run(tmp_path, targets, recs)
    assert p.returncode == 0, p.stderr
    lines, per_contig, total, nr = expected_depth_text(targets, recs)
    f = tr.TabixFile(gz)
    assert f.text == b"".join(lines)
    assert all(len(x) == 0xff00 for _, x in f.members[:-2]) and 0 < len(f.members[-2][1]) <= 0xff00 and len(f.members) > 3
    # the awk number: sum / NR as awk prints it -- and as the oracle's depth stage has it
    mean = total / nr
    assert p.stdout.decode().strip() == (str(int(mean)) if mean == int(mean) else "%.6g" % mean)
    text, s, n = orc.depth_mean(recs, targets)
    assert (s, n) == (total, nr) and text == p.stdout.decode().strip()
    # every contig through the index; contigs without coverage are not in it
    assert f.tbi["names"] == [n.encode() for n, _ in targets if n in per_contig]
    for name, mine in per_contig.items():
        assert f.fetch(name) == mine, name
    # windows inside the longest covered contig, across 16 kb borders
    name = max(per_contig, key=lambda n: len(per_contig[n]))
    L = dict(targets)[name]
    for beg, end in [(0, 1), (16383, 16385), (16384, 40000), (L - 100, L), (L // 2, L // 2 + 20000)]:
        want = [l for l in per_contig[name] if beg < int(l.split(b"\t")[1]) <= end]
        assert f.fetch(name, beg, end) == want, (name, beg, end)
    if long_mode:
        assert L > 3 * 16384 and max(len(r["ioff"]) for r in f.tbi["refs"]) > 3
    with pytest.raises(ValueError):
        f.fetch("no_such_contig")


def test_nothing_covered(tmp_path):
    targets = [("c1", 100)]
    p, gz = run(tmp_path, targets, [synth.BamRecord("r1", 4, -1, -1, 0, "")])
    assert p.returncode == 2 and b"division by zero" in p.stderr
    assert open(gz, "rb").read() == tr.EOF_MEMBER                      # an empty BGZF file
    assert tr.read_tbi(gz + ".tbi")["names"] == []
