"""Time the matching host path alone (no other GPU work): palace_match_arcs_from_edges + palace_match_decompose on a
bench-shaped graph (1M segments, ~50k junctions)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from palace_amd import capi

rng = np.random.Generator(np.random.PCG64(5))
n = 1_000_000
cn = rng.integers(0, 4, n).astype(np.int32)
e = np.zeros(50_000, dtype=capi.EDGE_DTYPE)
e["left"] = rng.integers(0, n, len(e)); e["right"] = rng.integers(0, n, len(e))
e["oL"] = rng.integers(0, 2, len(e)); e["oR"] = rng.integers(0, 2, len(e))
e["counts"] = rng.integers(1, 6, size=(len(e), 4))
with capi.Ctx(0) as ctx:
    for it in range(6):
        t0 = time.perf_counter()
        copies, src, dst, w = capi.match_arcs_from_edges(cn, e, 5)
        t1 = time.perf_counter()
        r = capi.match_decompose_views(ctx, copies, src, dst, 10, False)
        t2 = time.perf_counter()
        n_comp = r.n
        r.free()
        print(f"glue {1e3*(t1-t0):.2f} ms  decompose {1e3*(t2-t1):.2f} ms  comps {n_comp} arcs {len(src)}")
