"""Generate tests/golden/adjust.npz with the compiled reference (oracle/_ref):
  * AlignmentAdjuster::AdjustAlignment (AlignmentAdjuster.cpp:33-190) on 3 000 made-up results over the golden index of make_golden.py
    (tests/adjust_util.adjust_cases: shifted locations, both ends of every contig, both strands);
  * BaseAligner::AlignRead with -om 1 and -ae (ignoreAlignmentAdjustmentsForOm = false) on 1 200 reads made to need the adjuster
    (adjust_util.adjust_reads), aligner objects newly constructed for every read.
The genome is rebuilt from make_golden.py's seeds, so locations are those of tiny_index.npz."""
import os, sys, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from oracle import ref
from tests import util, adjust_util

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
W = '/tmp/snap_golden_adjust'
shutil.rmtree(W, ignore_errors=True); os.makedirs(W)
g = synth.make_genome(20260925, 100_000, n_contigs=2, repeat_frac=0.4, max_copies=60, repeat_len=(150, 1200), n_run_frac=0.004)
rng = np.random.default_rng(99)
alt = g[0][1][20_000:32_000].copy()
mut = rng.random(alt.size) < 0.01
alt[mut] = synth._ACGT[rng.integers(0, 4, size=int(mut.sum()))]
g.append(('chrA_alt1', alt))
synth.write_fasta(W + '/ref.fa', g)
ref.build_index(W + '/ref.fa', W + '/idx', 20, threads=4, extra=['-altContigName', 'chrA_alt1'])
idx = GenomeIndex.load_from_directory(W + '/idx')
gold = util.load_golden_index()
assert (idx.contig_begin == gold.contig_begin).all() and (idx.genome_padded == gold.genome_padded).all()
ri = ref.RefIndex(W + '/idx')

contigs, cstart = adjust_util.golden_contigs(gold)
L = 100
cb, cres = adjust_util.adjust_cases(20260927, contigs, cstart, 3000, L)
off = np.arange(cb.shape[0], dtype=np.uint64) * L; length = np.full(cb.shape[0], L, np.int32)
cexp = ri.adjust_alignments(cb, off, length, cres)

rb, rq = adjust_util.adjust_reads(20260928, contigs, 1200, L)
roff = np.arange(rb.shape[0] + 1, dtype=np.uint64) * L
p = abi.default_params(max_k=10, max_read_len=160, extra_search_depth=2)
with ref.fresh_objects(), ref.adjust_alignments():
    prim, _, sec, nsec = ri.align_single_secondary(p, 1, rb, rq, roff, threads=4)
with ref.fresh_objects():
    prim0, _, sec0, nsec0 = ri.align_single_secondary(p, 1, rb, rq, roff, threads=4)
stride = max(1, int(nsec.max()))
np.savez_compressed(OUT + '/adjust.npz', case_bases=cb, case_in=cres, case_out=cexp,
                    read_bases=rb, read_quals=rq, primary=prim, secondary=sec[:, :stride], nsec=nsec)
print('cases', cb.shape[0], 'moved', int((cexp['location'] != cres['location']).sum()), 'clipped', int((cexp['clipping_for_read_adjustment'] != 0).sum()),
      'dropped', int(((cexp['status'] == 0) & (cres['status'] != 0)).sum()))
print('reads', rb.shape[0], 'aligned', int((prim['status'] != 0).sum()), 'secondary', int(nsec.sum()), '(without -ae', int(nsec0.sum()), ')',
      'primaries changed by -ae', int(((prim['location'] != prim0['location']) | (prim['score'] != prim0['score']) | (prim['status'] != prim0['status'])).sum()))
print('wrote', OUT + '/adjust.npz', os.path.getsize(OUT + '/adjust.npz'), 'bytes')
