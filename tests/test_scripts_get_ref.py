"""get_ref_by_index.py counterpart (SURVEY "next" row N1).  The reference script needs Biopython, which this image
lacks, so these are hand-derived cases of share/palace/scripts/get_ref_by_index.py:6-89, not reference runs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "palace_amd", "scripts", "get_ref_by_index.py")


def run(tmp_path, fasta, fai, refs):
    p = lambda n: str(tmp_path / n)
    for n, text in (("db.fa", fasta), ("db.fa.fai", fai), ("refs.txt", refs)):
        with open(p(n), "w") as f:
            f.write(text)
    r = subprocess.run([sys.executable, SCRIPT, p("db.fa"), p("db.fa.fai"), p("refs.txt"), p("out.fa"), p("out.txt")],
                       capture_output=True, text=True, check=True)
    return open(p("out.fa")).read(), open(p("out.txt")).read(), r.stdout


FASTA = ">phA some description\nACGT\nAC GT\n\n>phB\nTTTT\r\n>phC/1 x\nGG\nCC\n"
FAI = "phA\t8\t20\t4\t5\nphB\t4\t35\t4\t6\nphC/1\t4\t50\t2\t3\n"


def test_sorted_by_index_last_float_and_multiline_sequences(tmp_path):
    refs = ("noise line\n"
            "ref_index\t3\t2\t3900\t4000\t0.975\n"
            "ref_index\t1\t1\t7000\t8000\t0.875\n"
            "  ref_index 2 5 10 10 1\n")
    fa, pc, out = run(tmp_path, FASTA, FAI, refs)
    assert fa == ">phA\nACGTACGT\n>phB\nTTTT\n>phC/1\nGGCC\n"
    assert pc == "phA\t0.875\nphB\t1.0\nphC/1\t0.975\n"          # float(...) formatted by str(): "1" -> "1.0"
    assert out.endswith("Processing complete!\n")


def test_repeated_index_keeps_the_later_line_and_missing_rows_only_warn(tmp_path):
    refs = ("ref_index\t2\t1\t10\t10\t0.8\n"
            "ref_index\t9\t1\t10\t10\t0.9\n"                       # no such .fai row
            "ref_index\t2\t1\t10\t10\t0.85\n"
            "ref_index\tx\t1\t10\t10\t0.5\n")                      # "x" is skipped: the first all-digit field is 1
    fai = FAI + "phD\t4\t60\t4\t5\n"
    fa, pc, out = run(tmp_path, FASTA, fai, refs + "ref_index\t4\t1\t1\t1\t0.99\n")
    assert fa == ">phA\nACGTACGT\n>phB\nTTTT\n"
    assert pc == "phA\t0.5\nphB\t0.85\n"
    assert "Warning: Index 9 not found in FAI file" in out and "Warning: Sequence 'phD' not found in FASTA file" in out


def test_eref_stdout_format_round_trip(tmp_path):
    """The lines our eref prints (ref_index\\t%d\\t%d\\t%d\\t%d\\t%g) are consumed as index / ratio."""
    refs = "ref_index\t2\t1\t4\t4\t1\nref_index\t1\t3\t7\t8\t0.875\n"
    _, pc, _ = run(tmp_path, FASTA, FAI, refs)
    assert pc == "phA\t0.875\nphB\t1.0\n"
