"""k_count_merged: time per look-up against the size of a contig's index (is the config-4 shape's count kernel bound by the
index not fitting an XCD's L2?).  One contig of a given length at the config-4 densities (100 000 segments and 1 000 tracks x
10 000 intervals per 3.1 Gb), count kernel time / look-ups.  usage: tools/exp_index_size.py [samples]"""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from gat_amd import _lib, problem, synthetic

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = _lib.Context(0)
G = 3.1e9
for mb in (24, 48, 96, 160, 249, 498):
    size = mb * 1000000
    contigs = collections.OrderedDict([("c", size)])
    f = size / G
    segs = synthetic.random_segments(contigs, int(100000 * f), 500, 11)
    annos = [("a%d" % i, synthetic.random_segments(contigs, max(2, int(10000 * f)), 2000, 100 + i)) for i in range(1000)]
    flat = problem.flatten_arrays(segs, annos, synthetic.workspace_contigs(contigs), None)
    P = _lib.Problem(ctx, flat)
    dev = ctx.alloc(1000 * S * 8)
    P.sample_and_count_device(["nucleotide-overlap"], 1, 0, S, dev)
    st = P.sample_and_count_device(["nucleotide-overlap"], 1, 0, S, dev)
    look, words = st["n_index_lookups"], st["n_index_entries"]
    print("contig %4d Mb: index %6.2f MB, %5d segments/sample, count kernel %7.3f ms, %6.3f ns per look-up, %5.1f words per look-up, form %d" %
          (mb, len(flat["annos"]) * 8 / 1e6, len(flat["segs"]), st["ms_count_main"], st["ms_count_main"] * 1e6 / max(1, look), words / max(1, look),
           st["merged_form"]), flush=True)
    ctx.free(dev)
    P.close()
