"""palace_amd/scripts/create_sub_graph.py against expectations derived by hand from the reference's rules
(share/palace/scripts/create_sub_graph.py:31-93, 186-378).  The reference script itself cannot run here (pysam is
absent), so there is no golden from it: parity is UNPINNED and this file is the check of the restatement."""
import gzip
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "palace_amd", "scripts", "create_sub_graph.py")

E1, E2, E3, E4 = ("EDGE_1_length_1000_cov_10.0", "EDGE_2_length_500_cov_20.0", "EDGE_3_length_2000_cov_5.0",
                  "EDGE_4_length_300_cov_7.0")
GRAPH = (f"SEG {E1} 12.5 1 1 0.900 1\nSEG {E2} 25 2 0 0.100 0\nSEG {E3} 5 1 0 0.500 1\nSEG {E4} 7 1 0 0.000 0\n"
         f"JUNC {E1} + {E2} + 9 0\nJUNC {E2} + {E3} - 6 1\nJUNC {E3} + {E4} + 8 0\n")
# need_second_match.txt: <path tokens ...> <reference>; the first token is the group key
MATCH = f"{E1}+{E2}+ phageA\n{E1}+{E2}+ phageB\n{E3}- phageC\n"
PERCENT = "phageA\t0.95\nphageB\t0.8\nphageC\t0.9\n"
# blast outfmt 6 + qlen slen (palace:524-528): E1 placed at 101..1100 of phageA; E2 covers 39.8 % of itself -> unplaced (-2)
BLAST = (f"{E1}\tphageA\t99.0\t1000\t0\t0\t1\t1000\t101\t1100\t0\t2000\t1000\t40000\n"
         f"{E2}\tphageA\t99.0\t200\t0\t0\t1\t200\t5000\t5199\t0\t400\t500\t40000\n")
# per-contig depth: E1 mean 12.0 over 1000 positions, E2 mean 25.0 over 480; E3 / E4 not listed -> from the contig name
DEPTH_TSV = f"{E1}\t12000\t1000\n{E2}\t12000\t480\n"

# phageA: weighted mean = 24000 / 1480 = 16.2162; copy numbers round(12 / 16.2162) = 1, round(25 / 16.2162) = 2
WANT_A = (f"SEG {E1} 12.0 1 1 0.900 1 101\nSEG {E2} 25.0 2 0 0.100 -1 -2\nJUNC {E1} + {E2} + 9 0\n")
# phageB: pruned from its group (0.8 < 0.85 and not the best).  phageC: no BLAST rows -> the order list of the previous
# reference is still in force (l.64-65), E3 is not in it -> -2 and the 7th column becomes -1; depth from the name
WANT_C = f"SEG {E3} 5.0 1 0 0.500 -1 -2\n"
WANT_REMAIN = f"SEG {E4} 7 1 0 0.000 0 -1\n"
WANT_SIMILAR = "phageA\nphageC\n"


def run_case(tmp_path, depth_name, depth_bytes):
    f = {k: str(tmp_path / k) for k in ("graph.txt", "match.txt", "blast.txt", "percent.txt", "similar.txt")}
    for k, v in (("graph.txt", GRAPH), ("match.txt", MATCH), ("blast.txt", BLAST), ("percent.txt", PERCENT)):
        open(f[k], "w").write(v)
    depth = str(tmp_path / depth_name)
    open(depth, "wb").write(depth_bytes)
    prefix = str(tmp_path / "out" / "s")
    os.makedirs(os.path.dirname(prefix))
    p = subprocess.run([sys.executable, SCRIPT, f["graph.txt"], prefix, f["match.txt"], "samtools", depth, f["blast.txt"], f["similar.txt"],
                        f["percent.txt"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    got = {n: open(os.path.join(os.path.dirname(prefix), n)).read() for n in sorted(os.listdir(os.path.dirname(prefix)))}
    return got, open(f["similar.txt"]).read()


def test_hand_case_with_contig_table(tmp_path):
    got, similar = run_case(tmp_path, "contig_depth.tsv", DEPTH_TSV.encode())
    assert similar == WANT_SIMILAR
    assert got == {"s_refphageAref.second": WANT_A, "s_refphageCref.second": WANT_C, "s_refremainref.second": WANT_REMAIN}


def test_hand_case_with_depth_gz(tmp_path):
    """the same through the bgzip'ed `samtools depth` text the reference reads (one line per covered position)"""
    lines = [f"{E1}\t{p + 1}\t12\n" for p in range(1000)] + [f"{E2}\t{p + 11}\t25\n" for p in range(480)]
    got, _ = run_case(tmp_path, "reads.bam.depth.gz", gzip.compress("".join(lines).encode()))
    assert got == {"s_refphageAref.second": WANT_A, "s_refphageCref.second": WANT_C, "s_refremainref.second": WANT_REMAIN}


def test_copy_number_floor_and_duplicates(tmp_path):
    """a reference that lists a contig twice weights it twice and prints it twice; a copy number of 0 becomes 1"""
    global MATCH
    keep = MATCH
    try:
        MATCH = f"{E1}+{E2}+{E1}- phageA\n"
        got, _ = run_case(tmp_path, "contig_depth.tsv", f"{E1}\t1000\t1000\n{E2}\t48000\t480\n".encode())
    finally:
        MATCH = keep
    # means 1.0 (x2, weight 1000 each) and 100.0 (weight 480): overall = 50000 / 2480 = 20.16 -> round(0.0496) = 0 -> 1 ; round(4.96) = 5
    a = got["s_refphageAref.second"].splitlines()
    assert a[:3] == [f"SEG {E1} 1.0 1 1 0.900 1 101", f"SEG {E2} 100.0 5 0 0.100 -1 -2", f"SEG {E1} 1.0 1 1 0.900 1 101"]
    assert got["s_refremainref.second"].splitlines()[0] == f"SEG {E3} 5 1 0 0.500 1 -1"
