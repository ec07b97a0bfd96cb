"""First GPU bring-up: lookups, LV and AlignRead (LV-only, -G-) parity against the reference."""
import os, sys, time, json, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth, abi
from snap_amd.index import GenomeIndex
from snap_amd.aligner import BaseAligner
from oracle import ref

W = '/tmp/w1'
os.makedirs(W, exist_ok=True)
g = synth.make_genome(1, 2_000_000, n_contigs=2, repeat_frac=0.5, max_copies=200, n_run_frac=0.002)
synth.write_fasta(W + '/ref.fa', g)
ref.build_index(W + '/ref.fa', W + '/idx', 20)
idx = GenomeIndex.load_from_directory(W + '/idx')
ri = ref.RefIndex(W + '/idx')
p = abi.default_params(max_k=8, use_affine_gap=0, max_read_len=160)
al = BaseAligner(idx, p)
out = {}

# tables
t = al.debug_tables()
ph, ind, pf = ref.tables()
out['tables_equal'] = bool((t['phred'] == ph).all() and (t['indel'] == ind).all() and (t['perfect'] == pf).all())
out['seed_prob_equal'] = t['seed_prob'] == ref.seed_prob(20)
out['wrapped_equal'] = all(int(t['wrapped'][i]) == ref.wrapped_next_seed(20, i) for i in range(20))
rng = np.random.default_rng(5)
bad = 0
for _ in range(20000):
    pa = rng.random() * 5; pb = pa * rng.random()
    # host check of thresholds
    x = 1 - pb / max(pa, pb)
    m = int(np.searchsorted(-t['mapq_threshold'][:71], -x, side='right')) - 1 if pb/max(pa,pb) < 1 else 70
    if m != ref.compute_mapq(pa, pb, 0, 0): bad += 1
out['mapq_threshold_mismatch'] = bad

# lookups
cat = np.concatenate([x for _, x in g])
pos = rng.integers(0, len(cat) - 20, size=20000)
seeds = np.stack([cat[q:q+20] for q in pos])
seeds[::7] = synth._ACGT[rng.integers(0, 4, size=seeds[::7].shape)]   # random (mostly absent) seeds
nh_r, h_r = ri.lookup_seeds(seeds, 512)
t0 = time.time(); nh_g, h_g = al.lookupSeed32(seeds, 512); out['lookup_s'] = time.time() - t0
out['lookup_nhits_equal'] = bool((nh_r == nh_g).all())
out['lookup_hits_equal'] = bool((h_r == h_g).all())
out['lookup_max_hits'] = int(nh_r.max())

# LV fuzz
def mutate(s, rng, rate):
    o = bytearray()
    for c in s:
        r = rng.random()
        if r < rate: o.append(b'ACGT'[rng.integers(0, 4)])
        elif r < rate * 1.3: continue
        elif r < rate * 1.6: o.append(c); o.append(b'ACGT'[rng.integers(0, 4)])
        else: o.append(c)
    return bytes(o)
texts, pats, quals, ks = [], [], [], []
for i in range(4000):
    L = int(rng.integers(1, 150)); q = int(rng.integers(0, len(cat) - 400))
    t_ = cat[q:q + L + 40].tobytes()
    pt = mutate(t_[:L], rng, rng.choice([0.0, 0.01, 0.03, 0.1]))[:L]
    if not pt: pt = b'A'
    texts.append(t_); pats.append(pt); quals.append(bytes(rng.integers(35, 74, size=len(pt), dtype=np.uint8))); ks.append(int(rng.integers(0, 31)))
for d in (1, -1):
    tt = texts if d == 1 else [x[::-1] for x in texts]
    r = ref.landau_vishkin(d, tt, pats, quals, ks)
    gq = al.computeEditDistance(d, tt, pats, quals, ks)
    for key in r:
        eq = (r[key] == gq[key])
        out['lv_%d_%s_mismatch' % (d, key)] = int((~eq).sum())

# AlignRead, LV only
reads = synth.make_reads(2, g, 20000, 150)
prim_r, alt_r, cnt_r, secs = ri.align_single(p, reads['bases'], reads['quals'], reads['offsets'], threads=8)
t0 = time.time(); prim_g, alt_g = al.AlignRead(reads['bases'], reads['quals'], reads['offsets']); dt = time.time() - t0
out['align_ref_s'] = secs; out['align_gpu_s'] = dt; out['ref_counters'] = cnt_r; out['gpu_counters'] = al.counters()
out['kernel_ms'] = al.kernel_time()
fields = [n for n in prim_r.dtype.names if n not in ('reserved',)]
mism = {}
found = prim_r['status'] != 0
for f in fields:
    a, b = prim_r[f], prim_g[f]
    if f in ('match_probability', 'probability_all_candidates', 'orig_location', 'popular_seeds_skipped'):
        ne = (a != b) & found        # undefined in the reference for NotFound results
    else:
        ne = a != b
    mism[f] = int(ne.sum())
out['align_mismatch'] = mism
out['status_counts'] = dict(collections.Counter(prim_g['status'].tolist()))
anybad = np.zeros(len(prim_r), bool)
for f in fields:
    if f in ('match_probability', 'probability_all_candidates', 'orig_location', 'popular_seeds_skipped'):
        anybad |= (prim_r[f] != prim_g[f]) & found
    else:
        anybad |= prim_r[f] != prim_g[f]
out['n_bad_reads'] = int(anybad.sum())
bi = np.nonzero(anybad)[0][:5]
out['examples'] = [dict(i=int(i), ref=str(prim_r[i]), gpu=str(prim_g[i])) for i in bi]
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/check1.json', 'w'), indent=1, default=str)
print(json.dumps(out, indent=1, default=str))
