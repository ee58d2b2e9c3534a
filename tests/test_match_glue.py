"""palace_match_arcs_from_edges (host code inside libpalace_hip.so) against a plain numpy statement of the same
rule: JUNC filter, arc + conjugate, equal arcs merged, rank order (weight desc, class asc, (u, v) asc)."""
import numpy as np
import pytest

from palace_amd import capi


def numpy_arcs(cn, edges, min_count):
    n_segs = len(cn)
    V = 2 * n_segs
    tot = edges["counts"].astype(np.int64).sum(axis=1)
    keep = tot >= min_count
    e, w = edges[keep], tot[keep]
    u = 2 * e["left"].astype(np.int64) + e["oL"]
    v = 2 * e["right"].astype(np.int64) + e["oR"]
    selfc = (v ^ 1) == u
    uu = np.concatenate([u, (v ^ 1)[~selfc]]); vv = np.concatenate([v, (u ^ 1)[~selfc]]); ww = np.concatenate([w, w[~selfc]])
    pair, inv = np.unique(uu * V + vv, return_inverse=True)
    wsum = np.zeros(len(pair), np.int64)
    np.add.at(wsum, inv, ww)
    pu, pv = pair // V, pair % V
    cls = np.minimum(pair, (pv ^ 1) * V + (pu ^ 1))
    order = np.lexsort((pair, cls, -wsum))
    return np.maximum(1, cn).astype(np.int64), pu[order].astype(np.int32), pv[order].astype(np.int32), wsum[order]


@pytest.mark.parametrize("n_segs,n_edges,seed", [(50, 400, 1), (3000, 20000, 2), (1, 3, 3), (7, 0, 4)])
def test_arcs_from_edges_equal_numpy(n_segs, n_edges, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    cn = rng.integers(-1, 5, size=n_segs).astype(np.int32)
    e = np.zeros(n_edges, dtype=capi.EDGE_DTYPE)
    e["left"] = rng.integers(0, n_segs, n_edges); e["right"] = rng.integers(0, n_segs, n_edges)
    e["oL"] = rng.integers(0, 2, n_edges); e["oR"] = rng.integers(0, 2, n_edges)
    e["counts"] = rng.integers(0, 4, size=(n_edges, 4))            # many ties, many below the filter
    if n_edges > 10:                                                 # an edge next to its own conjugate, and a self-conjugate one
        e[1] = e[0]; e[1]["left"], e[1]["right"] = e[0]["right"], e[0]["left"]
        e[1]["oL"], e[1]["oR"] = 1 - e[0]["oR"], 1 - e[0]["oL"]
        e[2]["right"] = e[2]["left"]; e[2]["oR"] = 1 - e[2]["oL"]
    want = numpy_arcs(cn, e, 5)
    got = capi.match_arcs_from_edges(cn, e, 5)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
