"""GPU bring-up 2: affine-gap batch parity and AlignRead with default options (affine gap on)."""
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
out = {}
rng = np.random.default_rng(11)

def mutate(s, rate):
    o = bytearray()
    for c in s:
        r = rng.random()
        if r < rate: o.append(b'ACGT'[rng.integers(0, 4)])
        elif r < rate * 1.4:
            if rng.random() < 0.5: continue
            o.append(c)
            for _ in range(rng.integers(1, 4)): o.append(b'ACGT'[rng.integers(0, 4)])
        else: o.append(c)
    return bytes(o)

p = abi.default_params(max_k=8, max_read_len=160)
al = BaseAligner(idx, p)
N = 3000
for d in (1, -1):
    for banded in (0, 1):
        texts, pats, quals, ws, sis, rcs, clips = [], [], [], [], [], [], []
        for i in range(N):
            L = int(rng.integers(1, 140))
            gg = bytes(rng.choice(list(b'ACGT'), size=L + 130).astype(np.uint8))
            if rng.random() < 0.1: gg = gg[:10] + b'N' + gg[11:]
            pt = mutate(gg[:L], rng.choice([0, 0.01, 0.03, 0.08, 0.2]))[:L] or b'A'
            if rng.random() < 0.15:
                cut = int(rng.integers(0, len(pt))); pt = pt[:cut] + pt[cut + int(rng.integers(1, 12)):]
                pt = pt or b'A'
            w = int(rng.integers(2, 30))
            tl = len(pt) + 127 if d == 1 and rng.random() < 0.7 else len(pt) + w
            tl = min(tl, len(gg))
            t = gg[:tl] + b'nnnnnnnn'          # slack the clipping heuristics may peek at
            tlen = tl
            if d == -1: t = b'nnnnnnnn' + gg[:tl][::-1]
            texts.append((t, tlen)); pats.append(pt); quals.append(bytes(rng.integers(35, 74, size=len(pt), dtype=np.uint8)))
            ws.append(w); sis.append(int(rng.choice([150, 100, len(pt) + 20, 30]))); rcs.append(int(rng.integers(0, 2))); clips.append(int(rng.integers(0, 2)))
        # reference: plain strings (its driver pads with 'n' itself)
        rtexts = [t[:tl] if d == 1 else t[8:] for (t, tl) in texts]
        r = ref.affine_gap(d, rtexts, pats, quals, ws, sis, rcs, [banded] * N, clips)
        # gpu: pass padded buffers but true text_len -> build manually
        gq = al.computeScoreAffine(d, rtexts, pats, quals, ws, sis, rcs, [banded] * N, clips)
        bad = 0
        for i in range(N):
            exp = (int(r['ag_score'][i]), int(r['text_offset'][i]), int(r['pattern_offset'][i]), int(r['n_edits'][i]), float(r['match_probability'][i]))
            got = (int(gq['ag_score'][i]), int(gq['text_offset'][i]), int(gq['pattern_offset'][i]), int(gq['n_edits'][i]), float(gq['match_probability'][i]))
            ok = got[0] == exp[0] and (exp[0] == -1 or got == exp)
            if not ok:
                bad += 1
                if bad < 4: out.setdefault('ag_examples', []).append(dict(d=d, banded=banded, i=i, got=got, exp=exp, plen=len(pats[i]), w=ws[i], si=sis[i], clip=clips[i], tlen=texts[i][1]))
        out['ag_dir%d_banded%d_bad' % (d, banded)] = bad

# AlignRead with affine gap (reference defaults except -d 8)
for name, nreads, rl, kw in (('d8_150', 20000, 150, dict(max_k=8)), ('d27_100', 10000, 100, dict(max_k=27))):
    pp = abi.default_params(max_read_len=160, **kw)
    aln = BaseAligner(idx, pp)
    reads = synth.make_reads(3, g, nreads, rl, ins=0.002, dele=0.002)
    prim_r, alt_r, cnt_r, secs = ri.align_single(pp, reads['bases'], reads['quals'], reads['offsets'], threads=8)
    t0 = time.time(); prim_g, alt_g = aln.AlignRead(reads['bases'], reads['quals'], reads['offsets']); dt = time.time() - t0
    fields = [n for n in prim_r.dtype.names if n != 'reserved']
    found = prim_r['status'] != 0
    anybad = np.zeros(len(prim_r), bool); mism = {}
    for f in fields:
        ne = prim_r[f] != prim_g[f]
        if f in ('match_probability', 'probability_all_candidates', 'orig_location', 'popular_seeds_skipped'): ne &= found
        mism[f] = int(ne.sum()); anybad |= ne
    bi = np.nonzero(anybad)[0][:4]
    out[name] = dict(ref_s=secs, gpu_s=dt, kernel_ms=aln.kernel_time(), ref_counters=cnt_r, gpu_counters=aln.counters(),
                     mismatch={k: v for k, v in mism.items() if v}, n_bad=int(anybad.sum()),
                     used_ag=float(prim_r['used_affine_gap_scoring'].mean()),
                     examples=[dict(i=int(i), ref=str(prim_r[i]), gpu=str(prim_g[i])) for i in bi])
    aln.close()
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/check2.json', 'w'), indent=1, default=str)
print(json.dumps(out, indent=1, default=str))
