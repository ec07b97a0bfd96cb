"""Generate tests/golden/paired_index.npz + paired_reads.npz with the compiled reference (oracle/_ref).
Run here (where /root/reference exists); the fixtures travel to the GPU box, the reference sources do not.

  paired_index.npz   a SNAP index (seed 20) of a 240 kb three-contig synthetic genome with repeat families, no ALT contigs
  paired_reads.npz   2 x 1500 "hard" 150 bp pairs + 600 2x100 bp pairs and the reference's PairedAlignmentResult for each
                     (ChimericPairedEndAligner::align over IntersectingPairedEndAligner::align, stage 0, and the
                     intersecting aligner alone, stage 1) under three option sets
"""
import os, sys, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from oracle import ref
from tests.pairs_util import hard_pairs, alt_liftover_genome

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
W = '/tmp/snap_golden_paired'
shutil.rmtree(W, ignore_errors=True); os.makedirs(W); os.makedirs(OUT, exist_ok=True)

g = synth.make_genome(20260926, 240_000, n_contigs=3, repeat_frac=0.35, max_copies=40, repeat_len=(150, 1500), n_run_frac=0.003)
synth.write_fasta(W + '/ref.fa', g)
ref.build_index(W + '/ref.fa', W + '/idx', 20, threads=4)
idx = GenomeIndex.load_from_directory(W + '/idx')
np.savez_compressed(OUT + '/paired_index.npz',
                    meta=np.array([idx.seed_len, idx.key_bytes, idx.n_hash_tables, int(idx.large), idx.location_size,
                                   idx.chromosome_padding, idx.n_bases], dtype=np.int64),
                    overflow=idx.overflow, hash_blob=idx.hash_blob, table_offset=idx.table_offset,
                    table_size=idx.table_size, genome_padded=idx.genome_padded,
                    contig_begin=idx.contig_begin, contig_is_alt=np.array([c.is_alt for c in idx.contigs]),
                    contig_names=np.array([c.name for c in idx.contigs]))

ri = ref.RefIndex(W + '/idx')
p150 = hard_pairs(7, g, 1500, 150, insert_mean=380)
p100 = hard_pairs(8, g, 600, 100, insert_mean=300, insert_min=100)
out = dict(b150=p150['bases'], q150=p150['quals'], o150=p150['offsets'], b100=p100['bases'], q100=p100['quals'], o100=p100['offsets'])
opts = dict(default_d8=(dict(max_k=8), {}), default_d27=(dict(max_k=27), {}), lvonly_d12=(dict(max_k=12, use_affine_gap=0), {}),
            spacing_d8=(dict(max_k=8), dict(min_spacing=100, max_spacing=600, num_seeds=12)))
for name, (kw, pkw) in opts.items():
    p = abi.default_params(max_read_len=160, **kw)
    pp = abi.default_paired_params(**pkw)
    for tag, pr in (('150', p150), ('100', p100)):
        for stage in (0, 1):
            prim, alt, cnt, _ = ri.align_paired(p, pp, pr['bases'], pr['quals'], pr['offsets'], threads=1, stage=stage)
            # reference nondeterminism (stale affine-gap traceback cells): vary the history and mark what moves
            n = prim.size
            unstable = np.zeros(n, bool)
            lens = np.diff(pr['offsets'].astype(np.int64))
            for order, th in ((np.arange(n)[::-1].copy(), 1), (np.random.default_rng(5).permutation(n), 3)):
                ro = np.empty(2 * n, dtype=np.int64); ro[0::2] = 2 * order; ro[1::2] = 2 * order + 1
                starts = pr['offsets'][:-1].astype(np.int64)
                bb = np.concatenate([pr['bases'][starts[i]:starts[i] + lens[i]] for i in ro])
                qq = np.concatenate([pr['quals'][starts[i]:starts[i] + lens[i]] for i in ro])
                oo = np.concatenate([[0], np.cumsum(lens[ro])]).astype(np.uint64)
                pv, _, _, _ = ri.align_paired(p, pp, bb, qq, oo, threads=th, stage=stage)
                back = np.empty_like(pv); back[order] = pv
                for f in ('status', 'location', 'score', 'mapq', 'ag_score', 'direction'):
                    m = prim[f] != back[f]
                    if f != 'status':
                        m &= prim['status'] != 0      # fields of NotFound reads are whatever the reference's stack held
                    unstable |= m.any(axis=1)
            print(name, tag, 'stage', stage, 'reference-unstable pairs:', np.nonzero(unstable)[0].tolist(),
                  'status hist', np.bincount(prim['status'].ravel(), minlength=3).tolist(), 'as pair %.3f' % prim['aligned_as_pair'].mean())
            key = '%s_%s_s%d' % (name, tag, stage)
            out[key + '_primary'] = prim
            out[key + '_alt'] = alt
            out[key + '_unstable'] = unstable
            out[key + '_counters'] = np.array([cnt['lv'], cnt['ag']], dtype=np.int64)
np.savez_compressed(OUT + '/paired_reads.npz', **out)

# ---- ALT liftover: the same kind of genome plus two ALT contigs and an -altLiftoverFile
#   chrA_alt1  forward copy of chrA[20000:32000] with a 10-base deletion and a 20-base insertion   5000M10D2990M20I4000M
#   chrB_alt2  reverse-complement copy of chrB[40000:49000] behind 100 novel bases                  100S9000M (flag 16)
W2 = W + '_alt'
shutil.rmtree(W2, ignore_errors=True); os.makedirs(W2)
g2, sam, alt_args = alt_liftover_genome()
synth.write_fasta(W2 + '/ref.fa', g2)
open(W2 + '/lift.sam', 'w').write(sam)
ref.build_index(W2 + '/ref.fa', W2 + '/idx', 20, threads=4, extra=alt_args + ['-altLiftoverFile', W2 + '/lift.sam'])
idx = GenomeIndex.load_from_directory(W2 + '/idx')
assert [c.is_alt for c in idx.contigs] == [False, False, False, True, True] and idx.contigs[4].proj_rc
np.savez_compressed(OUT + '/paired_alt_index.npz',
                    meta=np.array([idx.seed_len, idx.key_bytes, idx.n_hash_tables, int(idx.large), idx.location_size,
                                   idx.chromosome_padding, idx.n_bases], dtype=np.int64),
                    overflow=idx.overflow, hash_blob=idx.hash_blob, table_offset=idx.table_offset,
                    table_size=idx.table_size, genome_padded=idx.genome_padded,
                    contig_begin=idx.contig_begin, contig_is_alt=np.array([c.is_alt for c in idx.contigs]),
                    contig_names=np.array([c.name for c in idx.contigs]),
                    contig_proj_begin=np.array([c.proj_begin for c in idx.contigs], dtype=np.int64),
                    contig_proj_rc=np.array([c.proj_rc for c in idx.contigs]),
                    contig_proj_cigar=np.array([c.proj_cigar for c in idx.contigs]))
ri = ref.RefIndex(W2 + '/idx')
pa = hard_pairs(3, g2[3:], 700, 150, insert_mean=380)            # pairs drawn from the ALT contigs: these get lifted over
pb = hard_pairs(4, g2, 700, 150, insert_mean=380)
pr = dict(bases=np.concatenate([pa['bases'], pb['bases']]), quals=np.concatenate([pa['quals'], pb['quals']]),
          offsets=np.concatenate([pa['offsets'], pb['offsets'][1:] + pa['offsets'][-1]]).astype(np.uint64))
out = dict(b=pr['bases'], q=pr['quals'], o=pr['offsets'])
for name, kw in dict(default_d8=dict(max_k=8), default_d27=dict(max_k=27), emitalt_d8=dict(max_k=8, emit_alt_alignments=1)).items():
    p = abi.default_params(max_read_len=160, **kw)
    pp = abi.default_paired_params()
    for stage in (0, 1):
        prim, alt, cnt, _ = ri.align_paired(p, pp, pr['bases'], pr['quals'], pr['offsets'], threads=1, stage=stage)
        n = prim.size
        unstable = np.zeros(n, bool)
        lens = np.diff(pr['offsets'].astype(np.int64)); starts = pr['offsets'][:-1].astype(np.int64)
        for order, th in ((np.arange(n)[::-1].copy(), 1), (np.random.default_rng(5).permutation(n), 3)):
            ro = np.empty(2 * n, dtype=np.int64); ro[0::2] = 2 * order; ro[1::2] = 2 * order + 1
            bb = np.concatenate([pr['bases'][starts[i]:starts[i] + lens[i]] for i in ro])
            qq = np.concatenate([pr['quals'][starts[i]:starts[i] + lens[i]] for i in ro])
            oo = np.concatenate([[0], np.cumsum(lens[ro])]).astype(np.uint64)
            pv, _, _, _ = ri.align_paired(p, pp, bb, qq, oo, threads=th, stage=stage)
            back = np.empty_like(pv); back[order] = pv
            for f in ('status', 'location', 'score', 'mapq', 'ag_score', 'direction', 'liftover'):
                m = prim[f] != back[f]
                if f != 'status':
                    m &= prim['status'] != 0
                unstable |= m.any(axis=1)
        print('alt', name, 'stage', stage, 'lifted pairs', int(prim['liftover'].all(axis=1).sum()), 'first-ALT results', int((alt['status'] != 0).any(axis=1).sum()),
              'unstable', np.nonzero(unstable)[0].tolist())
        key = '%s_s%d' % (name, stage)
        out[key + '_primary'] = prim; out[key + '_alt'] = alt; out[key + '_unstable'] = unstable
np.savez_compressed(OUT + '/paired_alt_reads.npz', **out)
print('wrote', OUT)
