"""Hand-built generateGraph scenarios shared by the oracle test and the GPU/CLI parity tests.
Expected text is derived by hand from generate_graph.cpp (see comments)."""
from palace_amd.synth import BamRecord

TARGETS = [("ctgA", 1000), ("ctgB", 2000), ("ctgC", 400)]
FASTG_FAI = ("ctgA:ctgB;\t1000\t6\t60\t61\n"      # (A,B,+,+) and twin (B,A,-,-)
             "ctgC':ctgA';\t400\t6\t60\t61\n"      # (C,A,-,+) and twin (A,C,-,+)... see generate_graph.cpp:150-164
             "ctgB;\t2000\t6\t60\t61\n")
AVG_DEPTH = 0.5


def records():
    rec = []
    # five split reads: primary fwd on ctgA END region, SA fwd on ctgB START region -> A+ -> B+, in FASTG
    for i in range(5):
        rec.append(BamRecord(f"s{i}", 0, 0, 850, 60, "60M40S", nm=1, sa="ctgB,5,+,60S40M,60,0;"))
    # same geometry but the SA item fails NM<=5 -> no evidence, unpaired -> depth only
    rec.append(BamRecord("sx", 0, 0, 850, 60, "60M40S", nm=0, sa="ctgB,5,+,60S40M,60,6;"))
    # five cross-contig pairs: read1 fwd at ctgA END, mate rev at ctgC START -> A+ -> C+ (not in FASTG)
    for i in range(5):
        rec.append(BamRecord(f"p{i}", 0x61, 0, 900, 60, "100M", mtid=2, mpos=10, nm=0))
    # two split reads: primary REV on ctgB START, SA REV on ctgA END: layout (B-, A-) canonicalises to
    # (A+, B+) but the FASTG lookup uses (A, B, '-', '-') -> supplementCountNoFastg (generate_graph.cpp:863)
    for i in range(2):
        rec.append(BamRecord(f"t{i}", 0x10, 1, 4, 60, "40S60M", nm=0, sa="ctgA,900,-,60S40M,60,0;"))
    # the mates of p*: qname already in processedPairedReads -> refConsumed[ctgA] += 100 (:891)
    for i in range(5):
        rec.append(BamRecord(f"p{i}", 0x91, 2, 10, 60, "100M", mtid=0, mpos=900, nm=0))
    return rec


# depth ctgA = (6*60 + 5*100 + 5*100)/1000 = 1.36 -> cn floor(2.72+.5)=3 ; ctgB = 120/2000 = 0.06 -> 0 ;
# ctgC = 500/400 = 1.25 -> floor(2.5+.5)=3
EXPECTED = (b"SEG ctgA 1.36 3\nSEG ctgB 0.06 0\nSEG ctgC 1.25 3\n"
            b"JUNC ctgA + ctgB + 7 0\nJUNC ctgA + ctgC + 0 5\n")
# --debug (generate_graph.cpp:1068-1073): supportingReads in the order the evidence is met (:872, :1008), `name(flag)`, flag in decimal
EXPECTED_DEBUG = (b"SEG ctgA 1.36 3\nSEG ctgB 0.06 0\nSEG ctgC 1.25 3\n"
                  b"JUNC ctgA + ctgB + 7 0 READS: s0(0) s1(0) s2(0) s3(0) s4(0) t0(16) t1(16)\n"
                  b"JUNC ctgA + ctgC + 0 5 READS: p0(97) p1(97) p2(97) p3(97) p4(97)\n")


# ---- depth stage (palace:538-552): samtools depth | awk '{sum+=$3} END {print sum/NR}', derived by hand -------------
DEPTH_TARGETS = [("ctgA", 100), ("ctgB", 50)]


def depth_records():
    B = BamRecord
    return [
        B("d1", 0, 0, 10, 60, "5S20M"),             # ctgA [10,30)
        B("d2", 0, 0, 20, 60, "10M5D10M"),          # ctgA [20,30) and [35,45): the deletion is not counted (no -J)
        B("d3", 0x100, 0, 0, 60, "50M"),            # secondary: skipped
        B("d4", 0x400, 0, 0, 60, "50M"),            # duplicate: skipped
        B("d5", 0x800, 0, 50, 60, "10M"),           # supplementary: counted by samtools depth  [50,60)
        B("d6", 0x4, -1, -1, 0, ""),                # unmapped
        B("d7", 0, 0, 70, 60, "5M10N5M"),           # reference skip not counted  [70,75) [85,90)
        B("d8", 0, 0, 0, 60, "3=2X"),               # = and X count  [0,5)
        B("d9", 0, 0, 90, 60, "4M3I4M"),            # insertion consumes no reference  [90,98)
        B("d10", 0x200, 0, 0, 60, "50M"),           # QC fail: skipped
        B("d11", 0, 1, 45, 60, "10M"),              # ctgB [45,50): runs to the contig end exactly... (45+10 = 55 > 50: cut at 50)
    ]


# ctgA: [0,5) x1, [10,20) x1, [20,30) x2, [35,45) x1, [50,60) x1, [70,75) x1, [85,90) x1, [90,98) x1 ; ctgB: [45,50) x1
DEPTH_SUM = 5 + 10 + 20 + 10 + 10 + 5 + 5 + 8 + 5          # 78
DEPTH_NR = 5 + 10 + 10 + 10 + 10 + 5 + 5 + 8 + 5           # 68
DEPTH_TEXT = "1.14706"                                     # 78 / 68 = 1.147058...  ("%.6g")


# ---- every orientation / region combination of the two layout checks, expectation derived from the rules --------------
# checkPairedEndLayout (generate_graph.cpp:465-506): in the layout the LEFT read must run forward on the left contig's
# physical right end, the RIGHT read reverse on the right contig's physical left end.  A read qualifies for either side iff
# it is (forward, END) or (reverse, START) -- the same set for both sides -- so a pair is evidence iff both reads are in that
# set, and since order 0 (this record's read on the left) is tried first (:916) it always wins:
#     oL = '+' for (forward, END), '-' for (reverse, START);   oR = '+' for (reverse, START), '-' for (forward, END).
# checkSplitReadLayout (:510-538): both segments must run forward in the layout; the one that comes first in READ
# coordinates (canStitchReadIntervals, :401-428) is the left one:
#     left  qualifies as (forward, END) -> '+' or (reverse, START) -> '-'
#     right qualifies as (forward, START) -> '+' or (reverse, END) -> '-'.
# A layout with cR < cL (byte order of the names) is reported as (cR, flip(oR), cL, flip(oL)) (:855-861, :991-997).
LAYOUT_TARGETS = [("ctgA", 2000), ("ctgB", 2000)]
START_POS, END_POS = 9, 1899                      # 0-based; 1-based 10 <= min(300, L/2) and 1900 > max(L-300, L/2)


def flip(o):
    return "-" if o == "+" else "+"


def canonical(cl, ol, cr, o_r):
    return (cr, flip(o_r), cl, flip(ol)) if cr < cl else (cl, ol, cr, o_r)


def paired_case(rev1, end1, rev2, end2, swap_names=False):
    """5 records of read 1 only (the mate records are left out so that the :891 quirk stays out of the picture)."""
    a, b = (1, 0) if swap_names else (0, 1)
    flag = 0x41 | (0x10 if rev1 else 0) | (0x20 if rev2 else 0)
    recs = [BamRecord(f"p{i}", flag, a, END_POS if end1 else START_POS, 60, "100M", mtid=b, mpos=END_POS if end2 else START_POS, nm=0)
            for i in range(5)]
    left_ok = (not rev1 and end1) or (rev1 and not end1)
    right_ok = (rev2 and not end2) or (not rev2 and end2)
    if not (left_ok and right_ok):
        return recs, None
    ol = "+" if (not rev1 and end1) else "-"
    o_r = "+" if (rev2 and not end2) else "-"
    cl, cr = LAYOUT_TARGETS[a][0], LAYOUT_TARGETS[b][0]
    k = canonical(cl, ol, cr, o_r)
    return recs, f"JUNC {k[0]} {k[1]} {k[2]} {k[3]} 0 5\n"        # not in the FASTG: spanCountNoFastg (second number)


def split_case(rev_p, end_p, rev_s, end_s, primary_first=True, swap_names=False):
    """5 split reads of 100 bases: one part covers read bases 1-60, the other 61-100 (forward: 60M40S / 60S40M; on the
    reverse strand the same read intervals are 40S60M / 40M60S, generate_graph.cpp:330-383)."""
    a, b = (1, 0) if swap_names else (0, 1)
    first_cigar = lambda rev: "40S60M" if rev else "60M40S"       # read bases 1-60
    second_cigar = lambda rev: "40M60S" if rev else "60S40M"      # read bases 61-100
    cg_p = first_cigar(rev_p) if primary_first else second_cigar(rev_p)
    cg_s = second_cigar(rev_s) if primary_first else first_cigar(rev_s)
    pos_p = END_POS if end_p else START_POS
    pos_s = (END_POS if end_s else START_POS) + 1                  # SA positions are 1-based text
    sa = f"{LAYOUT_TARGETS[b][0]},{pos_s},{'-' if rev_s else '+'},{cg_s},60,0;"
    recs = [BamRecord(f"s{i}", 0x10 if rev_p else 0, a, pos_p, 60, cg_p, nm=0, sa=sa) for i in range(5)]
    (rev_l, end_l, c_l), (rev_r, end_r, c_r) = ((rev_p, end_p, a), (rev_s, end_s, b)) if primary_first else ((rev_s, end_s, b), (rev_p, end_p, a))
    left_ok = (not rev_l and end_l) or (rev_l and not end_l)
    right_ok = (not rev_r and not end_r) or (rev_r and end_r)
    if not (left_ok and right_ok):
        return recs, None
    ol = "+" if not rev_l else "-"
    o_r = "+" if not rev_r else "-"
    k = canonical(LAYOUT_TARGETS[c_l][0], ol, LAYOUT_TARGETS[c_r][0], o_r)
    return recs, f"JUNC {k[0]} {k[1]} {k[2]} {k[3]} 5 0\n"        # supplementCountNoFastg counts into the first number
