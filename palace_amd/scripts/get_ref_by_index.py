#!/usr/bin/env python3
"""Turn eref's stdout into the reference FASTA and the per-reference percentage table.

Counterpart of the reference's share/palace/scripts/get_ref_by_index.py (call site palace:483-498; SURVEY.md
"next" row N1): same command line
    get_ref_by_index.py <db.fasta> <db.fasta.fai> <eref stdout> <out.fasta> <out_percent.txt>
and the same output rules, without the Biopython dependency (the reference loads the whole DB through
Bio.SeqIO just to look sequences up by name):
  * of every line starting with "ref_index" take the first all-digit field after the tag as the 1-based
    `.fai` row and the last field that parses as a float as the percentage (l.6-37); a repeated index
    keeps the later percentage;
  * in ascending index order write ">name\\nsequence\\n" and "name\\tpercentage\\n" (l.74-89), name = column 0
    of the `.fai` row, sequence = the record whose id (header up to the first white space) equals it,
    with line breaks removed; indices without a `.fai` row and names without a record only produce a
    warning on stdout, as in the reference.
Parity status: UNPINNED -- the reference script cannot run here (Bio is absent); tests hold hand-derived cases.
"""
import sys


def _as_float(token):
    try:
        return float(token)
    except ValueError:
        return None


def reported_rows(eref_stdout_path):
    """{1-based .fai row: percentage text value} of eref's report lines.  A report line is `ref_index`, then white-space
    separated fields: the row is the first field made of digits only, the percentage the LAST field float() accepts
    (eref prints it last: `ref_index<TAB>row<TAB>n_intervals<TAB>el<TAB>ref_len<TAB>ratio`).  Lines without both are skipped;
    when a row is reported twice the later line wins."""
    rows = {}
    with open(eref_stdout_path) as f:
        for raw in f:
            fields = raw.split()
            if not fields or not fields[0].startswith("ref_index"):
                continue
            row = next((int(t) for t in fields[1:] if t.isdigit()), None)
            pct = next((v for v in map(_as_float, reversed(fields)) if v is not None), None)
            if row is not None and pct is not None:
                rows[row] = pct
    return rows


def fai_row_names(fai_path):
    """Column 0 of every `.fai` line, as a list: names[k] belongs to 1-based row k + 1."""
    with open(fai_path) as f:
        return [line.split("\t")[0] for line in f]


def fasta_records(path, wanted):
    """{id: sequence} of the records whose id is in `wanted`; a repeated id anywhere in the file is an error,
    as with Bio.SeqIO.to_dict."""
    seen, out = set(), {}
    name, chunks = None, []

    def close():
        if name is not None and name in wanted:
            out[name] = "".join(chunks)

    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                close()
                title = line[1:].rstrip()
                name = title.split(None, 1)[0] if title.split() else ""
                if name in seen:
                    raise ValueError(f"Duplicate key '{name}'")
                seen.add(name)
                chunks = []
            elif name is not None:
                chunks.append(line.rstrip().replace(" ", "").replace("\r", ""))
        close()
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 5:
        print("usage: get_ref_by_index.py db.fasta db.fasta.fai ref_names.txt out.fasta out_percent.txt", file=sys.stderr)
        return 2
    fasta_file, fai_file, ref_file, out_fasta, out_percent = argv
    print("Loading FAI index...")
    names = fai_row_names(fai_file)
    print("Processing reference file...")
    hits = reported_rows(ref_file)
    known = {row: names[row - 1] for row in hits if 1 <= row <= len(names)}
    print("Loading FASTA sequences...")
    records = fasta_records(fasta_file, set(known.values()))
    print("Writing output files...")
    with open(out_fasta, "w") as fa, open(out_percent, "w") as pc:
        for row in sorted(hits):
            if row not in known:
                print(f"Warning: Index {row} not found in FAI file")
                continue
            name = known[row]
            if name not in records:
                print(f"Warning: Sequence '{name}' not found in FASTA file")
                continue
            fa.write(f">{name}\n{records[name]}\n")
            pc.write(f"{name}\t{hits[row]}\n")
    print("Processing complete!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
