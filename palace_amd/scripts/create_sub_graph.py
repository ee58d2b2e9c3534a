#!/usr/bin/env python3
"""Cut the filtered conjugate graph into one sub-graph per candidate reference (`*_ref<ref>ref.second`) plus the rest
(`*_refremainref.second`) for the second round of matching.

Counterpart of the reference's share/palace/scripts/create_sub_graph.py (call site palace:651-662; SURVEY.md "next" row
N3), same eight positional arguments:

    create_sub_graph.py <filtered_graph> <out prefix> <need_second_match.txt> <samtools (unused)> <depth>
                        <assembly.blast> <similar_ref.txt out> <ref_percent.txt>

`<depth>` is what the reference opens with pysam.TabixFile: the bgzip'ed `samtools depth` text (`contig<TAB>pos<TAB>depth`,
only positions with depth > 0).  pysam is not needed here: a `.gz` argument is streamed once and reduced to per-contig
(sum, covered positions); any other file is read as the table `bamdepth --per-contig <bam>` writes
(`contig<TAB>sum<TAB>covered`, contigs without coverage absent) -- the same two numbers without the 40 GB text.

Output rules, restated from the reference (create_sub_graph.py:31-93 main, 186-262 update_segs_with_depth, 265-283
parse_graph_file, 284-326 parse_match_file, 328-378 parse_blast):
  * a sub-graph SEG line is `SEG <contig> <mean depth> <copy number> <gene> <score> 1 <order>`: mean depth = depth sum /
    covered positions of the contig (from its name `..._length_<L>_cov_<c>` when the depth file does not list it), copy
    number = round(mean / length-weighted mean over the reference's contigs), at least 1; gene / score = columns 4 and 5
    of the graph's SEG line; order = start of the contig's BLAST hit on the reference (-2: unplaced, then the 7th column
    becomes -1 as well);
  * its JUNC lines are the graph's junctions whose two contigs both belong to the reference, sorted as text;
  * the remain graph holds every SEG of the graph that no reference sub-graph used, with ` -1` appended, and the
    junctions among those.
Parity status: UNPINNED -- the reference script needs pysam, which this image lacks; tests hold hand-derived cases.
"""
import gzip
import re
import sys
from collections import defaultdict

EDGE_TOKEN = re.compile(r"(EDGE_[\w_]+_cov_[\d.]+)([+-])")


class ContigDepth:
    """per contig: (sum of per-base depths, number of positions listed) -- what `fetch(contig)` of the tabix file yields."""

    def __init__(self, path):
        self.table = {}
        if path.endswith(".gz"):
            with gzip.open(path, "rt") as f:
                cur, s, n = None, 0, 0
                for line in f:
                    name, _, depth = line.rstrip("\n").split("\t")
                    if name != cur:
                        if cur is not None:
                            ps, pn = self.table.get(cur, (0, 0))
                            self.table[cur] = (ps + s, pn + n)
                        cur, s, n = name, 0, 0
                    s += int(depth)
                    n += 1
                if cur is not None:
                    ps, pn = self.table.get(cur, (0, 0))
                    self.table[cur] = (ps + s, pn + n)
        else:
            with open(path) as f:
                for line in f:
                    cols = line.rstrip("\n").split("\t")
                    if len(cols) >= 3:
                        self.table[cols[0]] = (int(cols[1]), int(cols[2]))

    def mean_and_len(self, contig):
        """(mean depth, length used as weight) or None when the contig contributes nothing (create_sub_graph.py:206-234)."""
        if contig not in self.table:                       # pysam raises ValueError for a contig the index does not know
            parts = contig.split("_")
            return float(parts[-1]), int(parts[-3])
        s, n = self.table[contig]
        if n == 0:
            return None
        return s / n, n


def read_percent(path):
    out = {}
    with open(path) as f:
        for line in f.readlines():
            cols = line.split("\t")
            out[cols[0]] = float(cols[-1])
    return out


def read_graph(path):
    segs, juncs = {}, {}
    with open(path) as f:
        for line in f:
            parts = line.strip().split()
            if not parts:
                continue
            if parts[0] == "SEG":
                segs[parts[1]] = parts[2:]
            elif parts[0] == "JUNC":
                juncs[(parts[1], parts[2], parts[3], parts[4])] = parts
    return segs, juncs


def read_match(path, percent):
    """-> ({ref: [(contig, orient), ...]}, {group: [refs kept]}) (create_sub_graph.py:284-326)."""
    groups, members = {}, {}
    with open(path) as f:
        for line in f:
            parts = line.strip().split()
            if not parts:
                continue
            ref = parts[-1]
            groups.setdefault(parts[0], []).append(ref)
            members.setdefault(ref, []).extend((m.group(1), m.group(2)) for m in EDGE_TOKEN.finditer(" ".join(parts[:-1])))
    for key, refs in groups.items():
        best, best_ref = 0, ""
        for ref in refs[:]:
            if best < percent[ref]:
                best, best_ref = percent[ref], ref
            elif percent[ref] < 0.85:
                groups[key].remove(ref)
        if not groups[key]:
            groups[key].append(best_ref)
    return members, groups


def read_blast_order(path):
    """per reference: sorted [(start or -2/-1/0 marker, end, contig, covered fraction)] (create_sub_graph.py:328-378)."""
    hits = defaultdict(list)
    with open(path) as f:
        for line in f:
            p = line.strip().split("\t")
            if len(p) < 12:
                continue
            query, subject = p[0], p[1]
            lo, hi = min(int(p[8]), int(p[9])), max(int(p[8]), int(p[9]))
            sub_len, query_len = int(p[13]), int(p[12])
            span = hi - lo
            found = False
            rows = hits[subject]
            for i, row in enumerate(rows):
                if query != row[2]:
                    continue
                frac = rows[i][3] + span / query_len
                if abs(lo - hi) > abs(row[0] - row[1]):
                    rows[i] = (lo, hi, query, frac)
                elif lo - 1 < 10:
                    if sub_len - row[1] < 50:              # wraps around a circular reference
                        rows[i] = ((0 if hi == int(p[9]) else -1), hi, query, frac)
                else:
                    rows[i] = (rows[i][0], rows[i][1], rows[i][2], frac)
                found = True
            if not found:
                rows.append((lo, hi, query, span / query_len))
    placed = {ref: [((-2, b, c, d) if d < 0.5 else (a, b, c, d)) for a, b, c, d in rows] for ref, rows in hits.items()}
    for ref in placed:
        placed[ref].sort()
    return placed


def order_of(rows, contig):
    for row in rows:
        if contig == row[2]:
            return row[0]
    return -2


def segs_with_depth(ref_members, depth, graph_segs):
    weight_sum, len_sum, known = 0, 0, {}
    for contig, _ in ref_members:
        got = depth.mean_and_len(contig)
        if got is None:
            continue
        known[contig] = got
        weight_sum += got[0] * got[1]
        len_sum += got[1]
    if len_sum == 0:
        return []
    overall = weight_sum / len_sum
    out = []
    for contig, _ in ref_members:
        if contig not in known:
            continue
        mean = known[contig][0]
        cn = round(mean / overall) or 1
        info = graph_segs.get(contig)
        out.append(["SEG", contig, str(mean), str(cn), info[2] if info else "0", info[3] if info else "0", "1"])
    return out


def juncs_among(names, graph_juncs):
    names = set(names)
    return {" ".join(parts) for key, parts in graph_juncs.items() if key[0] in names and key[2] in names}


def run(argv):
    graph_path, prefix, match_path, _samtools, depth_path, blast_path, similar_out, percent_path = argv[:8]
    percent = read_percent(percent_path)
    graph_segs, graph_juncs = read_graph(graph_path)
    depth = ContigDepth(depth_path)
    members, groups = read_match(match_path, percent)
    placed = read_blast_order(blast_path)
    with open(similar_out, "w") as f:
        for key in sorted(groups):
            f.write(",".join(groups[key]) + "\n")
    wanted = [ref for key in sorted(groups) for ref in groups[key]]
    used = set()
    rows = []                                           # (the reference keeps the last reference's order list when a
    for ref, ref_members in sorted(members.items()):    #  reference has no BLAST rows of its own: l.64-65)
        if ref not in wanted:
            continue
        if ref in placed:
            rows = placed[ref]
        segs = segs_with_depth(ref_members, depth, graph_segs)
        if not segs:
            continue
        with open(f"{prefix}_ref{ref}ref.second", "w") as f:
            for seg in segs:
                used.add(seg[1])
                order = order_of(rows, seg[1])
                if order == -2:
                    seg[-1] = "-1"
                f.write(" ".join(seg) + " " + str(order) + "\n")
            names = [x for pair in ref_members for x in pair]          # (flattened as the reference does: names and signs)
            for j in sorted(juncs_among(names, graph_juncs)):
                f.write(j + "\n")
    rest = [name for name in graph_segs if name not in used]
    with open(f"{prefix}_refremainref.second", "w") as f:
        for name in rest:
            f.write(f"SEG {name} {' '.join(graph_segs[name])} -1\n")
        for j in sorted(juncs_among(rest, graph_juncs)):
            f.write(j + "\n")
    return 0


if __name__ == "__main__":
    if len(sys.argv) < 9:
        sys.stderr.write(__doc__)
        sys.exit(2)
    sys.exit(run(sys.argv[1:]))
