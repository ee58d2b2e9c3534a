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


def parse_ref_file(path):
    ref = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line.startswith("ref_index"):
                continue
            parts = line.split()
            index = next((int(p) for p in parts[1:] if p.isdigit()), None)
            percentage = None
            for p in reversed(parts):
                try:
                    percentage = float(p)
                    break
                except ValueError:
                    continue
            if index is not None and percentage is not None:
                ref[index] = percentage
    return ref


def load_fai_index(path):
    with open(path) as f:
        return {i: line.split("\t")[0] for i, line in enumerate(f, 1)}


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
    index_to_name = load_fai_index(fai_file)
    print("Processing reference file...")
    ref_data = parse_ref_file(ref_file)
    print("Loading FASTA sequences...")
    records = fasta_records(fasta_file, {index_to_name[i] for i in ref_data if i in index_to_name})
    print("Writing output files...")
    with open(out_fasta, "w") as fa, open(out_percent, "w") as pc:
        for index, percentage in sorted(ref_data.items()):
            if index not in index_to_name:
                print(f"Warning: Index {index} not found in FAI file")
                continue
            name = index_to_name[index]
            if name not in records:
                print(f"Warning: Sequence '{name}' not found in FASTA file")
                continue
            fa.write(f">{name}\n{records[name]}\n")
            pc.write(f"{name}\t{percentage}\n")
    print("Processing complete!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
