#!/usr/bin/env python3
"""Select the phage-relevant part of a conjugate graph (`*_graph.txt` -> `*_filtered_graph_pre.txt`).

Counterpart of the reference's share/palace/scripts/filter_graph.py (call site palace:568-579): the
same 13 positional arguments, the same output grammar:

  SEG <name> <depth> <cn> <gene 0|1> <score %.3f> <blast 0|1>      seed / neighbourhood / path SEGs
  JUNC ...                                                          copied verbatim, first occurrence only

and `all_hit_segs.txt` (SAMPLE<TAB>name<TAB>ref+score+gene+ flags) in SEG order of the input graph.
The reference emits its SEG block by iterating two Python sets, i.e. in a PYTHONHASHSEED-dependent
order (filter_graph.py:41, 253-258): the selected SEG lines, then the path-rescued ones.  This
implementation emits each of the two groups in the order of the SEG lines of the input graph -- one
of the orders the reference can produce, and the one the in-memory filter of the HIP library
(palace_stage04_filter) numbers the segments by, so that `matching` sees the same graph either way.
The JUNC block order is identical to the reference's.

argv: fastg.fai graph.txt out.txt depth f_th hit_seqs.out node_scores.out contigs.blast blast_ratio
      contigs.fasta.fai all_hit_segs.txt contigs.paths score_threshold

When `../bin/filter_graph` (palace_amd/host/filter_graph_main.cpp: the same selection in C++, byte-identical output,
~10x faster on a million SEG lines) has been built, this script hands the call over to it; PALACE_FILTER_PY=1 keeps
the Python implementation below (tests run both against the reference's golden outputs).
"""
import os
import subprocess
import sys


class GraphFilter:
    def __init__(self, blast_ratio: float, score_threshold: float):
        self.blast_ratio = blast_ratio
        self.score_threshold = score_threshold
        self.contig_len = {}        # name -> length (fasta .fai)
        self.id_to_name = {}        # "123" -> EDGE_123_length_...
        self.blast_hit = set()
        self.gene_hit = {}          # name -> '1'
        self.score_text = {}        # name -> "%.3f" text
        self.score_hit = set()

    # -- inputs ---------------------------------------------------------------------------------
    def load_fasta_index(self, path):
        with open(path) as f:
            for line in f:
                cols = line.strip().split("\t")
                self.contig_len[cols[0]] = int(cols[1])
                self.id_to_name[cols[0].split("_")[1]] = cols[0]

    def _blast_group_done(self, name, aligned):
        if aligned / self.contig_len[name] > self.blast_ratio or aligned > 2000:
            self.blast_hit.add(name)

    def load_blast(self, path):
        """Consecutive rows of one (query, subject) pair form a group; rows above the identity cut
        add their alignment length; a group passes by aligned fraction or by > 2000 aligned bases
        (filter_graph.py:66-94)."""
        cur_q = cur_s = ""
        aligned = 0
        cut = self.blast_ratio * 100
        with open(path) as f:
            for line in f:
                cols = line.strip().split("\t")
                q, s, ident, alen = cols[0], cols[1], float(cols[2]), int(cols[3])
                new_group = (cur_q != q and cur_q != "") or (cur_s != s and cur_s != "")
                if new_group:
                    self._blast_group_done(cur_q, aligned)
                    aligned = alen if ident > cut else 0
                elif ident > cut:
                    aligned += alen
                cur_q, cur_s = q, s
        if cur_q and cur_q in self.contig_len:
            self._blast_group_done(cur_q, aligned)

    def load_gene_hits(self, path):
        with open(path) as f:
            for line in f:
                self.gene_hit[line.split("\t")[0]] = "1"          # first column, untrimmed (l.101)

    def load_scores(self, path):
        with open(path) as f:
            for line in f:
                cols = line.strip().split("\t")
                text = "0.0" if "e" in cols[1].lower() else f"{float(cols[1]):.3f}"   # l.108-111
                self.score_text[cols[0]] = text
                if float(text) > self.score_threshold:
                    self.score_hit.add(cols[0])

    # -- per-segment facts ----------------------------------------------------------------------
    def _score_of(self, name):
        return float(self.score_text.get(name, "0"))

    def hit_flags(self, name):
        flags = ""
        if name in self.blast_hit:
            flags += "ref+"
        if self._score_of(name) > self.score_threshold:
            flags += "score+"
        if name in self.gene_hit:
            flags += "gene+"
        return flags

    @staticmethod
    def _plain_number(tok):
        """Fields written in scientific notation become plain: integers as integers, the rest with
        three decimals and trailing zeros/dot removed (filter_graph.py:178-188)."""
        if "e" not in tok.lower():
            return tok
        try:
            v = float(tok)
        except ValueError:
            return tok
        if v.is_integer():
            return str(int(v))
        return f"{v:.3f}".rstrip("0").rstrip(".")

    def seg_line(self, name, raw):
        cols = raw.strip().split()
        cols = cols[:2] + [self._plain_number(c) for c in cols[2:]]
        return "{} {} {} {}\n".format(" ".join(cols), self.gene_hit.get(name, "0"),
                                      self.score_text.get(name, "0.000"), "1" if name in self.blast_hit else "0")

    # -- contigs.paths rescue (filter_graph.py:126-151) -------------------------------------------
    def rescued_by_paths(self, path, support):
        rescued = []
        seen = set()
        with open(path) as f:
            for line in f:
                line = line.strip().replace(";", "")
                if line.startswith("NODE"):
                    continue
                members = [self.id_to_name[tok[:-1]] for tok in line.split(",")]
                total = sum(int(m.split("_")[3]) for m in members)
                backed = sum(int(m.split("_")[3]) for m in members if m in support)
                if backed > 0 and (backed / total >= 0.5 or backed > 2000):
                    for m in members:
                        if m not in seen:
                            seen.add(m)
                            rescued.append(m)
        return rescued


def run(argv):
    (fastg_fai, graph_path, out_path, _depth, _f_th, gene_file, score_file, blast_file, blast_ratio, fasta_fai,
     hit_segs_path, paths_file, score_threshold) = argv[:13]
    int(float(_depth))                                        # the reference parses it (l.11); unused after
    flt = GraphFilter(float(blast_ratio), float(score_threshold))
    flt.load_fasta_index(fasta_fai)
    flt.load_blast(blast_file)
    flt.load_gene_hits(gene_file)
    flt.load_scores(score_file)
    with open(fastg_fai) as f:                                # read like the reference (l.114-120); content unused
        for _ in f:
            pass
    with open(graph_path) as f:
        lines = f.readlines()

    raw_seg = {}
    raw_at = {}                                               # name -> line number of its latest SEG line
    hit_rows = []
    out_segs, out_seg_lines = set(), []                       # (line number of the SEG line a text was made from, text)

    def select(name):
        text = flt.seg_line(name, raw_seg[name])
        if text not in out_segs:
            out_segs.add(text)
            out_seg_lines.append((raw_at[name], text))

    juncs = []
    seeds = set()
    for at, line in enumerate(lines):                         # pass 1: SEG lines, seeds
        cols = line.rstrip().split(" ")
        if cols[0] != "SEG":
            continue
        name = cols[1]
        raw_seg[name] = line
        raw_at[name] = at
        flags = flt.hit_flags(name)
        if flags:
            seeds.add(name)
            select(name)
        hit_rows.append((name, flags))
    ends = [(ln, ln.rstrip().split(" ")) for ln in lines if ln.rstrip().split(" ")[0] != "SEG"]
    hop1 = set()
    for line, cols in ends:                                   # pass 2: junctions touching a seed (or self loops)
        left, right = cols[1], cols[3]
        if left == right or left in seeds or right in seeds:
            juncs.append(line)
            select(left)
            select(right)
            hop1.update((left, right))
    near = seeds | hop1
    for line, cols in ends:                                   # pass 3: junctions touching seeds or their neighbours
        left, right = cols[1], cols[3]
        if left in near or right in near:
            juncs.append(line)
            select(left)
            select(right)

    support = flt.blast_hit | set(flt.gene_hit) | flt.score_hit
    rescued = flt.rescued_by_paths(paths_file, support)
    out_seg_lines.sort(key=lambda x: x[0])                    # graph order (stable: texts of one line keep their order)
    already = {text.split(" ")[1] for _, text in out_seg_lines}
    with open(out_path, "w") as out:
        out.writelines(text for _, text in out_seg_lines)
        for name in sorted((n for n in rescued if n not in already), key=lambda n: raw_at[n]):
            out.write(f"{raw_seg[name].strip()} 0 1.0 0\n")
        emitted = set()
        for j in juncs:
            if j not in emitted:
                emitted.add(j)
                out.write(j)
    # hit_segs keeps the last flags per name at its first position (dict semantics of l.169, 266-269)
    last = {}
    for name, flags in hit_rows:
        if flags:
            last[name] = flags
    with open(hit_segs_path, "w") as out:
        for name, flags in last.items():
            out.write(f"SAMPLE\t{name}\t{flags}\n")
    return 0


if __name__ == "__main__":
    if len(sys.argv) < 14:
        sys.stderr.write(__doc__)
        sys.exit(2)
    native = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "filter_graph")
    if os.environ.get("PALACE_FILTER_PY") != "1" and os.access(native, os.X_OK):
        sys.exit(subprocess.call([native] + sys.argv[1:14]))
    sys.exit(run(sys.argv[1:]))
