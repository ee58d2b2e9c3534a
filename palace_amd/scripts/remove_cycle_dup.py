#!/usr/bin/env python3
"""Drop repeated two-line records from a cycle file, keeping first occurrences in order.

Counterpart of the reference's share/palace/scripts/remove_cycle_dup.py (call site palace:594-597):
same command line (`input_file output_file`), same output bytes.  Records are consecutive line
pairs; an odd trailing line is paired with an empty line ("\\n"), as the reference does (l.9-10).
"""
import sys


def dedup_records(lines):
    if len(lines) % 2:
        lines = lines + ["\n"]
    kept, seen = [], set()
    for head, body in zip(lines[0::2], lines[1::2]):
        if (head, body) not in seen:
            seen.add((head, body))
            kept.append(head)
            kept.append(body)
    return kept


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        print("usage: remove_cycle_dup.py input_file output_file", file=sys.stderr)
        return 2
    with open(argv[0]) as f:
        lines = f.readlines()
    with open(argv[1], "w") as f:
        f.writelines(dedup_records(lines))
    print(f"Unique pairs have been written to {argv[1]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
