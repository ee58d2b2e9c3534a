"""The headline bench in parts (the command line is ../bench.py):
  sample        the synthetic sample, generated on the device
  step          the HBM-resident step, its timing and the JSON line
  e2e           the same sample as files, and the chain of executables on them
  cpu_baseline  the CPU path timed beside it (the only part that may touch oracle/)
"""
from .sample import (HBM_PEAK_GBS, READ_LEN, SEED, contig_lengths, graph_to_arcs, make_graph_sample, make_sample,  # noqa: F401
                     make_side_inputs, paths_text)
