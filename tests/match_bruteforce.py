"""A second, independent statement of this repository's `matching` for tiny graphs: plain Python written
from the prose, not from oracle/match_oracle.cpp, so that the two restatements can disagree.  (The reference's binary is
absent, SURVEY.md F1: parity stays unpinned; this only guards the product's two checkers against a shared misreading.)
O(everything) on purpose: sets, sorts and walks, no cleverness."""


def decompose(seg_names, copies, juncs, iterations=10, aggressive=False, self_loops=False, break_cycles=False):
    """seg_names: list; copies: list of int (>= 1 enforced here); juncs: (left index, '+'|'-', right index, '+'|'-', weight).
    -> (linear text, cycle text)"""
    n = len(seg_names)
    flip = lambda v: v ^ 1
    weight = {}
    for l, ol, r, orr, w in juncs:
        u, v = 2 * l + (ol == "-"), 2 * r + (orr == "-")
        weight[(u, v)] = weight.get((u, v), 0) + w
        twin = (flip(v), flip(u))
        if twin != (u, v):
            weight[twin] = weight.get(twin, 0) + w
    klass = lambda a: min(a, (flip(a[1]), flip(a[0])))
    ranked = sorted(weight, key=lambda a: (-weight[a], klass(a), a))
    rank_of = {a: i for i, a in enumerate(ranked)}
    left = [max(1, c) for c in copies]
    tok = lambda v: seg_names[v >> 1] + "+-"[v & 1]
    line = lambda vs: "\t".join(tok(v) for v in vs) + "\n"
    lin, cyc, selfs, seen_lin, seen_cyc = [], [], [], set(), set()
    rounds = iterations + (1 if aggressive else 0)
    for t in range(rounds):
        if aggressive and t == rounds - 1:
            left = [1] * n
        live = {v for v in range(2 * n) if left[v >> 1] > 0}
        if not live:
            continue
        succ, pred = {}, {}
        for a in ranked:                                     # greedy in rank order
            u, v = a
            if u in live and v in live and u not in succ and v not in pred:
                succ[u], pred[v] = v, u
        comps, done = [], set()
        for v in sorted(live):                               # open walks start where nothing leads in
            if v in pred or v in done:
                continue
            walk = [v]
            while walk[-1] in succ:
                walk.append(succ[walk[-1]])
            twin = [flip(x) for x in reversed(walk)]
            done.update(walk); done.update(twin)
            comps.append((min(walk, twin, key=lambda w: w[0]), False))
        for v in sorted(live):                               # the rest are closed walks
            if v in done:
                continue
            walk = [v]
            while succ[walk[-1]] != v:
                walk.append(succ[walk[-1]])
            twin = [flip(x) for x in reversed(walk)]
            done.update(walk); done.update(twin)
            rot = lambda w: w[w.index(min(w)):] + w[:w.index(min(w))]
            comps.append((min(rot(walk), rot(twin), key=lambda w: w[0]), True))
        comps.sort(key=lambda c: c[0][0])
        for vs, closed in comps:
            uses = {}
            for x in vs:
                uses[x >> 1] = uses.get(x >> 1, 0) + 1
            pay = max(1, min(left[s] // k for s, k in uses.items()))
            for s, k in uses.items():
                left[s] = max(0, left[s] - pay * k)
            if not closed:
                if len(vs) == 1 and t > 0:
                    continue
                if line(vs) not in seen_lin:
                    seen_lin.add(line(vs)); lin.append(line(vs))
                continue
            if line(vs) in seen_cyc:
                continue
            seen_cyc.add(line(vs))
            if len(vs) == 1 and self_loops:
                selfs.append("self\n" + line(vs))
            else:
                cyc.append("iter %d\n" % t + line(vs))
            if break_cycles:                                 # opened behind its weakest (worst ranked) arc
                arcs = [(vs[i], vs[(i + 1) % len(vs)]) for i in range(len(vs))]
                worst = max(range(len(vs)), key=lambda i: rank_of[arcs[i]])
                opened = vs[worst + 1:] + vs[:worst + 1]
                if line(opened) not in seen_lin:
                    seen_lin.add(line(opened)); lin.append(line(opened))
    return "".join(lin), "".join(cyc + selfs)
