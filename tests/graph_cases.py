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
