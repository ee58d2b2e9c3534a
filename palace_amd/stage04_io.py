"""Host-side plumbing around the resident stage-04 object (palace_stage04_* of include/palace_hip.h) for tests and bench.py:
the per-sample inputs of filter_graph.py as arrays, and the text `matching` writes for a result.  No compute of its own that
the library does: parsing on the way in, formatting on the way out (palace_amd/host/matching_main.cpp is the product's
formatter; this is its Python twin for results that never were files).
"""
from __future__ import annotations

import numpy as np


def name_ranks(names) -> np.ndarray:
    """dense rank of every name in byte order = position of its SEG line in `_graph.txt` (generate_graph.cpp:1048-1050)"""
    order = np.argsort(np.array([n.encode() for n in names], dtype="S"), kind="stable")
    rank = np.empty(len(names), np.int32)
    rank[order] = np.arange(len(names), dtype=np.int32)
    return rank


def name_lengths(names) -> np.ndarray:
    """the token between the 3rd and 4th '_' (get_edge_len, filter_graph.py:50-52)"""
    return np.array([int(n.split("_")[3]) for n in names], dtype=np.int32)


def paths_csr(paths_text: str, names):
    """contigs.paths as the filter and matching read it: every line that does not start with NODE is one path line of
    comma-separated `<id><+|->` tokens (';' dropped, filter_graph.py:131-137); -> (offsets, tokens = 2 * contig + minus, -1 =
    unknown id or malformed token)."""
    by_id = {}
    for i, n in enumerate(names):
        by_id[n.split("_")[1]] = i                               # a later name with the same id wins, as in a dict
    off, tok = [0], []
    for line in paths_text.splitlines():
        line = line.strip().replace(";", "")
        if line.startswith("NODE"):
            continue
        for t in line.split(","):
            t = t.strip()
            c = by_id.get(t[:-1], -1) if len(t) >= 2 and t[-1] in "+-" else -1
            tok.append(-1 if c < 0 else 2 * c + (t[-1] == "-"))
        off.append(len(tok))
    return np.array(off, np.int64), np.array(tok, np.int32)


def matching_text(res, contig_of, names, self_loops: bool = True, break_cycles: bool = False):
    """(linear file, cycle file) as palace_amd/host/matching_main.cpp writes them for a compact result `res` of the
    filtered graph whose segment f is contig contig_of[f]: components round by round in first-vertex order, the bare
    segments (one-vertex paths of round 0) merged in between; a path line once; a bare segment only in round 0; cycles as
    `iter <n>` + line, one-vertex cycles as `self` + line behind the others when -s; with -b every cycle also opened at its
    weakest arc."""
    off, verts, kind, it, open_at = (np.asarray(x) for x in (res.off, res.verts, res.kind, res.iter, res.open_at))
    nm = [names[int(c)] for c in contig_of]
    tok = lambda v: nm[v >> 1] + "+-"[v & 1]
    n_f = len(contig_of)
    bare = np.flatnonzero(np.unpackbits(np.asarray(res.bare).view(np.uint8), bitorder="little")[:n_f]) if n_f else np.zeros(0, np.int64)
    lin, cyc, selfs, seen_l, seen_c = [], [], [], set(), set()
    c, n_comp = 0, len(kind)
    rounds = int(it.max()) + 1 if n_comp else 1
    for t in range(rounds):
        c_end = c
        while c_end < n_comp and it[c_end] == t:
            c_end += 1
        items = [(int(verts[off[k]]), k) for k in range(c, c_end)]
        if t == 0:
            items += [(2 * int(s), -1) for s in bare]
            items.sort()
        for first, k in items:
            if k < 0:
                body = tok(first) + "\n"
                if body not in seen_l:
                    seen_l.add(body); lin.append(body)
                continue
            vs = [int(v) for v in verts[off[k]:off[k + 1]]]
            line_of = lambda start: "\t".join(tok(vs[(start + i) % len(vs)]) for i in range(len(vs))) + "\n"
            if not kind[k]:
                if len(vs) == 1 and t > 0:
                    continue
                body = line_of(0)
                if body not in seen_l:
                    seen_l.add(body); lin.append(body)
                continue
            body = line_of(0)
            if body in seen_c:
                continue
            seen_c.add(body)
            if len(vs) == 1 and self_loops:
                selfs.append("self\n" + body)
            else:
                cyc.append(f"iter {t}\n" + body)
            if break_cycles:
                opened = line_of(int(open_at[k]))
                if opened not in seen_l:
                    seen_l.add(opened); lin.append(opened)
        c = c_end
    return "".join(lin), "".join(cyc + selfs)
