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
