"""Property tests (hypothesis) of the glue scripts' pure functions: size-independent invariants next to the golden cases."""
import os
import sys

from hypothesis import given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "palace_amd", "scripts"))
import get_ref_by_index as grbi  # noqa: E402
import remove_cycle_dup as rcd  # noqa: E402

line = st.text(alphabet="ab+-\t", min_size=0, max_size=6).map(lambda s: s + "\n")


@settings(max_examples=200, deadline=None)
@given(st.lists(line, max_size=40))
def test_remove_cycle_dup_is_idempotent_and_order_preserving(lines):
    once = rcd.dedup_records(list(lines))
    assert rcd.dedup_records(list(once)) == once                         # idempotent
    assert len(once) % 2 == 0
    pairs = list(zip(once[0::2], once[1::2]))
    assert len(set(pairs)) == len(pairs)                                 # no pair twice
    padded = list(lines) + (["\n"] if len(lines) % 2 else [])
    src = list(zip(padded[0::2], padded[1::2]))
    first_seen = list(dict.fromkeys(src))                                # first occurrences, in order
    assert pairs == first_seen


@settings(max_examples=200, deadline=None)
@given(st.lists(st.tuples(st.integers(1, 50), st.integers(0, 9), st.floats(0.0, 1.0, allow_nan=False)), max_size=30))
def test_get_ref_by_index_parse_takes_first_index_and_last_float(tmp_path_factory, rows):
    p = tmp_path_factory.mktemp("g") / "refs.txt"
    with open(p, "w") as f:
        f.write("header line\n")
        for idx, n_int, ratio in rows:
            f.write(f"ref_index\t{idx}\t{n_int}\t100\t200\t{ratio:g}\n")
    got = grbi.reported_rows(str(p))
    want = {}
    for idx, _, ratio in rows:
        want[idx] = float(f"{ratio:g}")                                  # a repeated index keeps the later line
    assert got == want
