"""End-to-end parity of the drop-in executables (palace_amd/bin/{eref,generateGraph}) on a GPU:
same argv, same files, byte-identical outputs vs the golden reference outputs / the oracle."""
import hashlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import binding as orc
from palace_amd import synth
from tests import graph_cases as gc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "palace_amd", "bin")


def run(cmd, **kw):
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


# ------------------------------------------------------------------------------------------------
# eref
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def eref_files(golden_eref, tmp_path_factory):
    g = golden_eref
    d = tmp_path_factory.mktemp("eref_cli")
    fa = str(d / "db.fa")
    open(fa, "wb").write(g["db_fasta"].tobytes())
    synth.ReadSet(g["r1_bases"], g["r1_offsets"]).write_fastq(str(d / "r_1.fq"), "1")
    synth.ReadSet(g["r2_bases"], g["r2_offsets"]).write_fastq(str(d / "r_2.fq"), "2")
    return d, fa


@pytest.mark.parametrize("key,hr,pr", [("stdout_090_085", "0.9", "0.85"), ("stdout_080_050", "0.8", "0.5"),
                                       ("stdout_095_090", "0.95", "0.9")])
def test_eref_cached_index_equals_reference_stdout(eref_files, golden_eref, key, hr, pr):
    d, fa = eref_files
    # index as the reference left it beside the DB (same coder permutation as the golden run)
    orc.build_index_file(fa, golden_eref["index_header"], fa + ".k32.index.dat", fa + ".genome.len.txt")
    tmp = str(d / "tmp.txt")
    open(tmp, "w").write("stale")
    for threads, reads_as in (("1", "packed"), ("8", "packed"), ("8", "ascii")):     # the parser threads pack the reads (default) or ship bytes
        p = run([os.path.join(BIN, "eref"), str(d / "r_1.fq"), str(d / "r_2.fq"), fa, tmp, hr, pr, threads],
                env=dict(os.environ, PALACE_EREF_INPUT=reads_as))
        assert p.returncode == 0, p.stderr
        assert p.stdout == golden_eref[key].tobytes()
        assert os.path.getsize(tmp) == 0                      # extract_ref.cpp:825, 899


def test_eref_builds_index_like_reference(eref_files, golden_eref, tmp_path):
    d, fa0 = eref_files
    fa = str(tmp_path / "db2.fa")
    open(fa, "wb").write(open(fa0, "rb").read())
    p = run([os.path.join(BIN, "eref"), str(d / "r_1.fq"), str(d / "r_2.fq"), fa, str(tmp_path / "t.txt"), "0.9", "0.85", "4"],
            env=dict(os.environ, PALACE_CODER_SEED="77"))
    assert p.returncode == 0, p.stderr
    idx = open(fa + ".k32.index.dat", "rb").read()
    hdr = np.frombuffer(idx[:400], dtype=np.uint8)
    ofa = str(tmp_path / "oracle.fa")
    open(ofa, "wb").write(open(fa0, "rb").read())
    orc.build_index_file(ofa, hdr, ofa + ".k32.index.dat", ofa + ".genome.len.txt")
    assert hashlib.sha256(idx).digest() == hashlib.sha256(open(ofa + ".k32.index.dat", "rb").read()).digest()
    assert open(fa + ".genome.len.txt", "rb").read() == golden_eref["genome_len_txt"].tobytes()
    # stdout under that permutation == oracle under the same permutation
    cc = orc.header_to_cc(hdr)
    t = orc.CountTable()
    t.count(golden_eref["r1_bases"], golden_eref["r1_offsets"], cc)
    t.count(golden_eref["r2_bases"], golden_eref["r2_offsets"], cc)
    assert p.stdout == orc.scan_index_file(ofa + ".k32.index.dat", t, 0.9, 0.85)
    t.free()
    # second run finds the index it wrote and gives the same answer
    p2 = run([os.path.join(BIN, "eref"), str(d / "r_1.fq"), str(d / "r_2.fq"), fa, str(tmp_path / "t.txt"), "0.9", "0.85", "1"])
    assert p2.returncode == 0 and p2.stdout == p.stdout


def test_eref_rejects_foreign_index(eref_files, tmp_path):
    d, fa0 = eref_files
    fa = str(tmp_path / "db3.fa")
    open(fa, "wb").write(open(fa0, "rb").read())
    open(fa + ".k32.index.dat", "wb").write(b"\0" * 1000)
    p = run([os.path.join(BIN, "eref"), str(d / "r_1.fq"), str(d / "r_2.fq"), fa, str(tmp_path / "t.txt"), "0.9", "0.85", "1"])
    assert p.returncode != 0 and b"does not belong" in p.stderr


# ------------------------------------------------------------------------------------------------
# generateGraph
# ------------------------------------------------------------------------------------------------
def graph_cli(tmp_path, targets, fai_text, recs, avg, extra=()):
    bam, fai, out = str(tmp_path / "t.bam"), str(tmp_path / "g.fastg.fai"), str(tmp_path / "graph.txt")
    synth.write_bam(bam, targets, recs, block=4000)            # small blocks: records straddle BGZF blocks
    open(fai, "w").write(fai_text)
    p = run([os.path.join(BIN, "generateGraph"), *extra, bam, fai, out, f"{avg:.6g}"])
    assert p.returncode == 0, p.stderr
    return open(out, "rb").read(), fai


def test_graph_hand_case(tmp_path):
    got, _ = graph_cli(tmp_path, gc.TARGETS, gc.FASTG_FAI, gc.records(), gc.AVG_DEPTH)
    assert got == gc.EXPECTED
    got, _ = graph_cli(tmp_path, gc.TARGETS, gc.FASTG_FAI, gc.records(), gc.AVG_DEPTH, ["--debug"])
    assert got == gc.EXPECTED_DEBUG


@pytest.mark.parametrize("seed,n_contigs,n_events", [(1, 40, 3000), (2, 12, 2000), (3, 300, 20000)])
def test_graph_random_equals_oracle(tmp_path, seed, n_contigs, n_events):
    targets, fai_text, recs, avg = synth.random_graph_case(synth.rng_for(seed), n_contigs, n_events)
    got, fai = graph_cli(tmp_path, targets, fai_text, recs, avg)
    want = orc.graph_run(recs, targets, fai, float(f"{avg:.6g}"))
    assert got == want
    assert got.count(b"JUNC") > 3


@pytest.mark.parametrize("extra,opt", [
    (["-e", "150", "-n", "2", "--min-count", "2"], dict(max_end=150, max_nm=2, min_count=2)),
    (["-P", "0", "--min-count", "1"], dict(enable_paired=0, min_count=1)),
    (["--max-span-frac", "0.1", "-q", "30", "--min-count", "3"], dict(max_span_frac=0.1, min_mapq=30, min_count=3)),
    (["--both-order", "1", "--min-count", "2", "--lib", "FR", "-p", "0.5", "--min-score", "0.1"], dict(both_order=1, min_count=2)),
])
def test_graph_options(tmp_path, extra, opt):
    targets, fai_text, recs, avg = synth.random_graph_case(synth.rng_for(11), 30, 4000)
    got, fai = graph_cli(tmp_path, targets, fai_text, recs, avg, extra)
    o = orc.graph_default_opts()
    for k, v in opt.items():
        setattr(o, k, v)
    assert got == orc.graph_run(recs, targets, fai, float(f"{avg:.6g}"), o)


@pytest.mark.parametrize("seed,extra,opt", [(1, [], {}), (3, ["--min-count", "1"], dict(min_count=1)),
                                            (5, ["--both-order", "1", "--min-count", "2"], dict(both_order=1, min_count=2))])
def test_graph_debug_lists_the_supporting_reads(tmp_path, seed, extra, opt):
    """--debug (generate_graph.cpp:1068-1073): every JUNC line ends in ' READS: name(flag) ...', the evidence in the order the
    reference meets it (records in file order, the SA items of a record in list order)."""
    targets, fai_text, recs, avg = synth.random_graph_case(synth.rng_for(seed), 40, 6000)
    got, fai = graph_cli(tmp_path, targets, fai_text, recs, avg, ["--debug"] + extra)
    o = orc.graph_default_opts()
    o.debug = 1
    for k, v in opt.items():
        setattr(o, k, v)
    want = orc.graph_run(recs, targets, fai, float(f"{avg:.6g}"), o)
    assert got == want
    juncs = [l for l in got.split(b"\n") if l.startswith(b"JUNC")]
    assert len(juncs) > 3 and all(b" READS: " in l for l in juncs)
    # the list of a junction is as long as its two counters say
    for l in juncs:
        t = l.split(b" READS:")[0].split()
        assert len(l.split(b" READS:")[1].split()) == int(t[5]) + int(t[6])
    plain, _ = graph_cli(tmp_path, targets, fai_text, recs, avg, extra)
    assert plain == b"".join(l.split(b" READS:")[0] + b"\n" for l in got.split(b"\n") if l)


def test_graph_with_members_inflated_on_the_device(tmp_path):
    """A BAM of thousands of small BGZF members: with device helpers (bam_device.hpp; the default) part of them is inflated by
    palace_bgzf_inflate, the output is the one of the host alone and the oracle's.  (PALACE_BAM_HOST_SHARE holds the host threads
    back at 30 % of the file while a helper works: a file this small would be done before the HIP runtime is up.)"""
    targets, fai_text, recs, avg = synth.random_graph_case(synth.rng_for(9), 200, 60000)
    bam, fai = str(tmp_path / "t.bam"), str(tmp_path / "g.fastg.fai")
    synth.write_bam(bam, targets, recs, block=1500)
    open(fai, "w").write(fai_text)
    outs = {}
    for dev in ("0", "2"):
        out = str(tmp_path / f"graph{dev}.txt")
        p = run([os.path.join(BIN, "generateGraph"), bam, fai, out, f"{avg:.6g}"], env=dict(os.environ, PALACE_BAM_DEVICE=dev, PALACE_TRACE="1", PALACE_BAM_HOST_SHARE="30"))
        assert p.returncode == 0, p.stderr
        outs[dev] = open(out, "rb").read()
        if dev == "2":
            m = re.search(rb"(\d+) of (\d+) members were inflated by helpers", p.stderr)
            assert m and int(m.group(2)) > 3000 and int(m.group(1)) > 0, p.stderr[-2000:]
    assert outs["0"] == outs["2"] == orc.graph_run(recs, targets, fai, float(f"{avg:.6g}"))


def test_graph_damaged_member_fails_the_same_with_and_without_the_device(tmp_path):
    """a BGZF member near the end of the file whose DEFLATE data are damaged: the device decoder refuses it (or the host's does), zlib
    then refuses it too, and generateGraph fails with the loader's message either way -- never a graph from garbage"""
    import struct
    targets, fai_text, recs, avg = synth.random_graph_case(synth.rng_for(10), 100, 30000)
    bam, fai = str(tmp_path / "t.bam"), str(tmp_path / "g.fastg.fai")
    synth.write_bam(bam, targets, recs, block=1500)
    open(fai, "w").write(fai_text)
    raw = bytearray(open(bam, "rb").read())
    starts, p = [], 0
    while p < len(raw):
        starts.append(p)
        p += struct.unpack_from("<H", raw, p + 16)[0] + 1
    assert len(starts) > 3000
    m = starts[-40]                                             # (the last member is the empty EOF marker)
    size = struct.unpack_from("<H", raw, m + 16)[0] + 1
    for k in range(m + 18 + 2, m + size - 8):                   # the middle of its DEFLATE payload
        raw[k] ^= 0x5A
    open(bam, "wb").write(bytes(raw))
    msgs = []
    for dev in ("0", "2"):
        p = run([os.path.join(BIN, "generateGraph"), bam, fai, str(tmp_path / f"o{dev}.txt"), f"{avg:.6g}"],
                env=dict(os.environ, PALACE_BAM_DEVICE=dev, PALACE_BAM_HOST_SHARE="30"))
        assert p.returncode == 1, (dev, p.stderr[-500:])
        msgs.append(p.stderr.strip().splitlines()[-1])
    assert msgs[0] == msgs[1] and b"BGZF" in msgs[0]


def test_graph_long_contigs_underflow_gate(tmp_path):
    """N50 ~ 50 kb with a >120 kb tail: '-' orientations on long contigs underflow exp() to 0."""
    targets, fai_text, recs, avg = synth.random_graph_case(synth.rng_for(21), 25, 6000, long_mode=True)
    assert max(l for _, l in targets) > 120000
    got, fai = graph_cli(tmp_path, targets, fai_text, recs, avg)
    assert got == orc.graph_run(recs, targets, fai, float(f"{avg:.6g}"))


def test_graph_usage_and_errors(tmp_path):
    p = run([os.path.join(BIN, "generateGraph"), "only", "three", "args"])
    assert p.returncode == 1 and b"Usage:" in p.stderr
    p = run([os.path.join(BIN, "generateGraph"), str(tmp_path / "missing.bam"), "x", str(tmp_path / "o"), "1"])
    assert p.returncode == 1 and b"Failed to open BAM" in p.stderr


# ------------------------------------------------------------------------------------------------
# matching (own algorithm; checked against oracle/match_oracle.cpp -- the reference binary is absent)
# ------------------------------------------------------------------------------------------------
FILTER_G = np.load(os.path.join(ROOT, "tests", "golden", "filter_cases.npz"))


def ring_graph():
    """three rings (one with a repeat segment of copy number 2), a self loop and a chain"""
    n = lambda i, L=1000: f"EDGE_{i}_length_{L}_cov_5.0"
    seg = [f"SEG {n(i)} 10 {2 if i == 3 else 1} 0 0.500 0\n" for i in range(1, 15)]
    j = []
    ring = lambda ids, w: [f"JUNC {n(a)} + {n(b)} + {w} 0\n" for a, b in zip(ids, ids[1:] + ids[:1])]
    j += ring([1, 2, 3], 9) + ring([3, 4, 5, 6], 7)          # two rings sharing segment 3 (cn 2)
    j += [f"JUNC {n(7)} + {n(7)} + 8 0\n"]                    # self loop
    j += [f"JUNC {n(8)} + {n(9)} - 6 1\n", f"JUNC {n(9)} - {n(10)} + 5 0\n"]
    j += ring([11, 12], 5) + [f"JUNC {n(12)} + {n(13)} + 5 0\n", f"JUNC {n(13)} + {n(14)} - 2 0\n"]
    paths = "NODE_1_length_3000_cov_5\n13+,14+\nNODE_1_length_3000_cov_5'\n14-,13-\n"
    return "".join(seg + j), paths


def match_cli(tmp_path, graph_text, paths_text, flags):
    g, p = str(tmp_path / "g.txt"), str(tmp_path / "contigs.paths")
    open(g, "w").write(graph_text)
    open(p, "w").write(paths_text)
    lin, cyc = str(tmp_path / "lin.txt"), str(tmp_path / "cyc.txt")
    r = run([os.path.join(BIN, "matching"), "-g", g, "-r", lin, "-c", cyc, *flags, "-l", p])
    assert r.returncode == 0, r.stderr
    it = int(flags[flags.index("-i") + 1]) if "-i" in flags else 10
    want = orc.match_run(g, p, it, "-s" in flags, "-b" in flags, "--aggressive" in flags)
    return (open(lin, "rb").read(), open(cyc, "rb").read()), want


@pytest.mark.parametrize("flags", [["-s", "-i", "10"], ["-i", "10", "-b", "--aggressive"], ["-i", "2"], ["-s", "-b", "-i", "1"]])
def test_matching_rings(tmp_path, flags):
    g, p = ring_graph()
    got, want = match_cli(tmp_path, g, p, flags)
    assert got == want
    assert b"iter 0\n" in got[1]
    if "-s" in flags:
        assert b"self\nEDGE_7_length_1000_cov_5.0+\n" in got[1]


@pytest.mark.parametrize("case", [0, 1, 2])
@pytest.mark.parametrize("flags", [["-s", "-i", "10"], ["-i", "10", "-b", "--aggressive"]])
def test_matching_filtered_graphs(tmp_path, case, flags):
    got, want = match_cli(tmp_path, FILTER_G[f"case{case}_pre"].tobytes().decode(),
                          FILTER_G[f"case{case}_contigs_paths"].tobytes().decode(), flags)
    assert got == want and got[0]


def test_matching_large_random(tmp_path):
    rng = synth.rng_for(5)
    names, lens = synth.contig_names(rng, 20000)
    seg = "".join(f"SEG {nm} {rng.random() * 30:.4g} {int(rng.integers(0, 4))} 0 0.100 0\n" for nm in names)
    junc = []
    for _ in range(30000):
        a, b = int(rng.integers(0, 20000)), int(rng.integers(0, 20000))
        junc.append(f"JUNC {names[a]} {'+-'[int(rng.integers(0, 2))]} {names[b]} {'+-'[int(rng.integers(0, 2))]} "
                    f"{int(rng.integers(1, 40))} {int(rng.integers(0, 5))}\n")
    side = synth.filter_side_files(rng, names, lens)
    got, want = match_cli(tmp_path, seg + "".join(junc), side["contigs_paths"], ["-s", "-i", "10", "-b"])
    assert got == want
    assert got[1].count(b"iter") > 0


def test_matching_hubs_and_ties(tmp_path):
    """segments with thousands of junctions (long arc lists at one vertex), most weights equal (the rank is then decided by the
    second key word), every orientation: the vertex-side search of the best arc against the oracle's sort"""
    rng = synth.rng_for(77)
    names, lens = synth.contig_names(rng, 6000)
    seg = "".join(f"SEG {nm} 12.5 {int(rng.integers(1, 4))} 0 0.100 0\n" for nm in names)
    junc = []
    for hub in (0, 1, 2):
        for b in rng.choice(np.arange(3, 6000), size=2500, replace=False):
            a, b = (hub, int(b)) if rng.integers(0, 2) else (int(b), hub)
            junc.append(f"JUNC {names[a]} {'+-'[int(rng.integers(0, 2))]} {names[b]} {'+-'[int(rng.integers(0, 2))]} {int(rng.choice([7, 7, 7, 9]))} 0\n")
    for _ in range(4000):
        a, b = int(rng.integers(0, 6000)), int(rng.integers(0, 6000))
        junc.append(f"JUNC {names[a]} {'+-'[int(rng.integers(0, 2))]} {names[b]} {'+-'[int(rng.integers(0, 2))]} 7 0\n")
    side = synth.filter_side_files(rng, names, lens)
    for flags in (["-s", "-i", "10"], ["-i", "6", "-b", "--aggressive"]):
        got, want = match_cli(tmp_path, seg + "".join(junc), side["contigs_paths"], flags)
        assert got == want
    assert got[0].count(b"\n") > 1000


def test_matching_usage():
    p = run([os.path.join(BIN, "matching"), "-g", "x"])
    assert p.returncode == 1 and b"Usage" in p.stderr


def test_eref_subsampling_follows_glibc_rand_stream(eref_files, golden_eref):
    """E3: when 2 * sum(fq1 bases) exceeds the sampling target, one rand() % 100 draw per sequence line
    (fq1 then fq2, seed 1) decides which reads count (extract_ref.cpp:955-960, 1124-1148, 1239-1240)."""
    d, fa = eref_files
    g = golden_eref
    orc.build_index_file(fa, g["index_header"], fa + ".k32.index.dat", fa + ".genome.len.txt")
    n1, n2 = len(g["r1_offsets"]) - 1, len(g["r2_offsets"]) - 1
    fq1_bases = int(g["r1_offsets"][-1])
    target = fq1_bases                                        # ratio = 100 * target / (2 * fq1_bases) = 50
    ratio = 100 * target // (2 * fq1_bases)
    assert ratio == 50
    draws = orc.glibc_rand_stream(1, n1 + n2) % 100
    keep1, keep2 = (draws[:n1] < ratio).astype(np.uint8), (draws[n1:] < ratio).astype(np.uint8)
    cc = orc.header_to_cc(g["index_header"])
    t = orc.CountTable()
    t.count(g["r1_bases"], g["r1_offsets"], cc, keep1)
    t.count(g["r2_bases"], g["r2_offsets"], cc, keep2)
    want = orc.scan_index_file(fa + ".k32.index.dat", t, 0.8, 0.5)
    t.free()
    # bin/eref_testhooks = eref_main.cpp compiled with -DPALACE_TEST_HOOKS (a lowered sampling target; the shipped eref has
    # no such knob).  The same path at its real threshold (> 1 Gbase in fq1) is pinned by tests/test_gpu_configs.py against a
    # run of the compiled reference.
    for reads_as in ("packed", "ascii"):
        p = run([os.path.join(BIN, "eref_testhooks"), str(d / "r_1.fq"), str(d / "r_2.fq"), fa, str(d / "tmp.txt"), "0.8", "0.5", "2"],
                env=dict(os.environ, PALACE_EREF_SAMPLE_TARGET=str(target), PALACE_EREF_INPUT=reads_as))
        assert p.returncode == 0, p.stderr
        assert p.stdout == want
    assert want != g["stdout_080_050"].tobytes()              # sampling really changed the answer


# ------------------------------------------------------------------------------------------------
# N1: eref's optional 8th/9th arguments == eref stdout -> get_ref_by_index.py (palace:483-498)
# ------------------------------------------------------------------------------------------------
def test_eref_folded_ref_outputs_equal_the_script_chain(eref_files, golden_eref, tmp_path):
    d, fa = eref_files
    orc.build_index_file(fa, golden_eref["index_header"], fa + ".k32.index.dat", fa + ".genome.len.txt")
    out_fa, out_pc = str(tmp_path / "phage_refs.fasta"), str(tmp_path / "ref_percent.txt")
    p = run([os.path.join(BIN, "eref"), str(d / "r_1.fq"), str(d / "r_2.fq"), fa, str(d / "tmp.txt"), "0.8", "0.5", "4", out_fa, out_pc])
    assert p.returncode == 0, p.stderr
    assert p.stdout == golden_eref["stdout_080_050"].tobytes() and p.stdout.count(b"\n") >= 3
    # the two-step way: stdout file + a .fai of the DB (samtools faidx layout: name, length, offset, line bases, line width)
    names_txt = str(tmp_path / "ref_names.txt")
    open(names_txt, "wb").write(p.stdout)
    fai = str(tmp_path / "db.fa.fai")
    with open(fai, "w") as f:
        off = 0
        for rec in open(fa, "rb").read().split(b">")[1:]:
            head, _, body = rec.partition(b"\n")
            seq = body.replace(b"\n", b"")
            f.write(f"{head.split()[0].decode() if head.split() else ''}\t{len(seq)}\t{off + len(head) + 2}\t80\t81\n")
            off += len(rec) + 1
    want_fa, want_pc = str(tmp_path / "w.fasta"), str(tmp_path / "w.txt")
    q = run([sys.executable, os.path.join(ROOT, "palace_amd", "scripts", "get_ref_by_index.py"), fa, fai, names_txt, want_fa, want_pc])
    assert q.returncode == 0, q.stderr
    assert open(out_fa, "rb").read() == open(want_fa, "rb").read()
    assert open(out_pc, "rb").read() == open(want_pc, "rb").read()
    assert open(out_pc, "rb").read().count(b"\n") == p.stdout.count(b"\n")


# ------------------------------------------------------------------------------------------------
# N2: depth stage (samtools depth | awk) in-process
# ------------------------------------------------------------------------------------------------
def test_bamdepth_hand_case(tmp_path):
    bam = str(tmp_path / "d.bam")
    synth.write_bam(bam, gc.DEPTH_TARGETS, gc.depth_records())
    p = run([os.path.join(BIN, "bamdepth"), bam])
    assert p.returncode == 0, p.stderr
    assert p.stdout.decode() == gc.DEPTH_TEXT + "\n"
    synth.write_bam(bam, gc.DEPTH_TARGETS, [synth.BamRecord("u", 4, -1, -1, 0, "")])
    assert run([os.path.join(BIN, "bamdepth"), bam]).returncode == 2          # nothing covered: awk would divide by zero


@pytest.mark.parametrize("seed,n_contigs,n_events,long_mode", [(5, 40, 4000, False), (6, 300, 30000, False), (7, 20, 5000, True)])
def test_depth_stage_random_equals_oracle_and_auto_graph(tmp_path, seed, n_contigs, n_events, long_mode):
    targets, fai_text, recs, _ = synth.random_graph_case(synth.rng_for(seed), n_contigs, n_events, long_mode=long_mode)
    text, s, nr = orc.depth_mean(recs, targets)
    bam = str(tmp_path / "r.bam")
    synth.write_bam(bam, targets, recs, block=5000)
    p = run([os.path.join(BIN, "bamdepth"), bam])
    assert p.returncode == 0, p.stderr
    assert p.stdout.decode().strip() == text and nr > 100
    # generateGraph ... auto == generateGraph ... <that text>
    fai = str(tmp_path / "g.fastg.fai")
    open(fai, "w").write(fai_text)
    a, b = str(tmp_path / "auto.txt"), str(tmp_path / "given.txt")
    pa = run([os.path.join(BIN, "generateGraph"), bam, fai, a, "auto"])
    pb = run([os.path.join(BIN, "generateGraph"), bam, fai, b, text])
    assert pa.returncode == 0 and pb.returncode == 0, (pa.stderr, pb.stderr)
    assert open(a, "rb").read() == open(b, "rb").read() == orc.graph_run(recs, targets, fai, float(text))
    assert f"Average sequencing depth: {text}".encode() in pa.stderr
    # per contig (what step 5 takes from the tabix-indexed depth file): restated here with per-base arrays
    depth = {i: np.zeros(L, dtype=np.int64) for i, (_, L) in enumerate(targets)}
    for r in recs:
        if (r.flag & 0x704) or r.tid < 0 or r.pos < 0:
            continue
        at = r.pos
        for n, op in synth.parse_cigar(r.cigar):
            if op in (0, 7, 8):
                depth[r.tid][at:at + n] += 1
            if op in (0, 2, 3, 7, 8):
                at += n
    want = "".join(f"{targets[i][0]}\t{int(d.sum())}\t{int((d > 0).sum())}\n" for i, d in depth.items() if (d > 0).any())
    pc = run([os.path.join(BIN, "bamdepth"), "--per-contig", bam])
    assert pc.returncode == 0, pc.stderr
    assert pc.stdout.decode() == want
    assert sum(int(l.split("\t")[1]) for l in want.splitlines()) == s and sum(int(l.split("\t")[2]) for l in want.splitlines()) == nr


def test_matching_batch_equals_one_process_per_graph(tmp_path):
    """N3: `matching --batch <list>` (every *.second sub-graph of step 5 in ONE process / one GPU run, palace:651-806)
    writes byte for byte what one `matching -g ... -b --aggressive` process per graph writes -- and what the oracle writes.
    Sub-graphs share contigs (the same SEG name in several graphs), carry the 7th SEG column (create_sub_graph.py:77,89)
    and include an empty one."""
    rng = synth.rng_for(31)
    names, lens = synth.contig_names(rng, 3000)
    side = synth.filter_side_files(rng, names, lens)
    paths = str(tmp_path / "contigs.paths")
    open(paths, "w").write(side["contigs_paths"])
    flags = ["-i", "10", "-b", "--aggressive"]
    jobs = []
    for k in range(40):
        m = int(rng.integers(0, 150)) if k != 7 else 0
        mine = rng.choice(len(names), size=m, replace=False) if m else []
        seg = "".join(f"SEG {names[i]} {rng.random() * 30:.4g} {int(rng.integers(0, 4))} 0 0.100 1 {int(rng.integers(-1, 9))}\n" for i in mine)
        junc = []
        for _ in range(2 * m):
            a, b = int(rng.choice(mine)), int(rng.choice(mine))
            junc.append(f"JUNC {names[a]} {'+-'[int(rng.integers(0, 2))]} {names[b]} {'+-'[int(rng.integers(0, 2))]} "
                        f"{int(rng.integers(1, 40))} {int(rng.integers(0, 5))}\n")
        g = str(tmp_path / f"s_ref{k}ref.second")
        open(g, "w").write(seg + "".join(junc))
        jobs.append(g)
    lst = str(tmp_path / "batch.txt")
    open(lst, "w").write("".join(f"{g}\t{g[:-7]}_linear.txt\t{g[:-7]}_cycle.txt\n" for g in jobs))
    r = run([os.path.join(BIN, "matching"), "--batch", lst, *flags, "-l", paths])
    assert r.returncode == 0, r.stderr
    n_cyc = 0
    for g in jobs:
        lin1, cyc1 = g + ".lin1", g + ".cyc1"
        r = run([os.path.join(BIN, "matching"), "-g", g, "-r", lin1, "-c", cyc1, *flags, "-l", paths])
        assert r.returncode == 0, r.stderr
        got = (open(g[:-7] + "_linear.txt", "rb").read(), open(g[:-7] + "_cycle.txt", "rb").read())
        assert got == (open(lin1, "rb").read(), open(cyc1, "rb").read()), g
        assert got == orc.match_run(g, paths, 10, False, True, True), g
        n_cyc += got[1].count(b"iter")
    assert n_cyc > 10


def test_graph_every_layout_combination(tmp_path):
    """All 16 + 32 orientation / region combinations of the pair and split-read layout checks (tests/graph_cases.py, with
    and without the name swap), each on its own contig pair, in ONE BAM: the JUNC block must be exactly the rule table's."""
    import dataclasses
    import itertools
    targets, recs, want = [], [], []
    cases = []
    for b in itertools.product([False, True], repeat=4):
        for swap in (False, True):
            cases.append(gc.paired_case(*b, swap))
            cases.append(gc.split_case(*b, True, swap))
            cases.append(gc.split_case(*b, False, swap))
    for k, (rs, line) in enumerate(cases):
        base = len(targets)
        names = [f"c{k:03d}A", f"c{k:03d}B"]
        targets += [(names[0], 2000), (names[1], 2000)]
        for r in rs:
            sa = r.sa.replace("ctgA", names[0]).replace("ctgB", names[1]) if r.sa else r.sa
            recs.append(dataclasses.replace(r, qname=f"k{k}_{r.qname}", tid=base + r.tid, mtid=(base + r.mtid if r.mtid >= 0 else r.mtid), sa=sa))
        if line:
            want.append(line.replace("ctgA", names[0]).replace("ctgB", names[1]))
    recs.sort(key=lambda r: (r.tid, r.pos))
    got, fai = graph_cli(tmp_path, targets, "", recs, 1.0)
    juncs = [l + "\n" for l in got.decode().splitlines() if l.startswith("JUNC")]
    assert sorted(juncs) == sorted(want) and len(want) == 24
    assert got == orc.graph_run(recs, targets, fai, 1.0)
