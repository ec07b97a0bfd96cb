"""TEST INFRASTRUCTURE (host only): adversarial pairs for the paired-end SAM writer ("redo loop"): both mates get one or two
indels within their first 7 reference bases and their leftmost ends lie 0-7 bases apart, so that leading-indel adjustments and write-order flips
(SNAPLib/ReadWriter.cpp:447-500) are the norm.  The FASTQs go through the reference CLI (oracle/_ref/snap-aligner paired -t 1) and through
snapgpu-sam paired built against the wavefront emulator (tests/emu/_build/snapgpu-sam-emu); every record must be identical.
  python scripts/fuzz_paired_writer.py <seed> <n_pairs> [options passed to both tools, e.g. -G- -d 12]"""
import os, subprocess, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snap_amd import synth
from oracle import ref
seed = int(sys.argv[1]); n = int(sys.argv[2])
d = os.environ.get("SNAP_FUZZ_DIR", "/tmp/snap_fuzz_writer"); os.makedirs(d, exist_ok=True)
g = synth.make_genome(7, 1_000_000, n_contigs=2, repeat_frac=0.02)
if not os.path.exists(d + "/idx/GenomeIndex"):
    synth.write_fasta(d + "/g.fa", g); ref.build_index(d + "/g.fa", d + "/idx", seed_len=20, threads=8)
rng = np.random.default_rng(seed)
comp = np.zeros(256, np.uint8); comp[ord('A')] = ord('T'); comp[ord('T')] = ord('A'); comp[ord('C')] = ord('G'); comp[ord('G')] = ord('C'); comp[ord('N')] = ord('N')
ACGT = np.frombuffer(b"ACGT", np.uint8)
def edit_left2(seg):
    out = []; o = 0
    evs = sorted(set(int(x) for x in rng.integers(0, 7, size=int(rng.integers(1, 3)))))
    while len(out) < 150 and o < len(seg):
        if evs and o >= evs[0]:
            evs.pop(0)
            k = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                out.extend(int(x) for x in ACGT[rng.integers(0, 4, size=k)])
            else:
                o += k
                continue
        out.append(int(seg[o])); o += 1
    return np.array(out[:150], np.uint8)
with open(d + "/f1.fq", "wb") as f1, open(d + "/f2.fq", "wb") as f2:
    for i in range(n):
        c = g[int(rng.integers(0, 2))][1]
        p = int(rng.integers(1000, len(c) - 2000))
        delta = int(rng.integers(0, 8))
        a = edit_left2(c[p:p + 200]) if rng.random() < 0.8 else c[p:p + 150].copy()
        bseg = edit_left2(c[p + delta:p + delta + 200]) if rng.random() < 0.8 else c[p + delta:p + delta + 150].copy()
        b = comp[bseg[::-1]]
        if rng.random() < 0.5: a, b = b, a                  # either mate may be the forward one
        q = b"I" * 150
        f1.write(b"@p%d/1\n" % i + a.tobytes() + b"\n+\n" + q + b"\n")
        f2.write(b"@p%d/2\n" % i + b.tobytes() + b"\n+\n" + q + b"\n")
extra = sys.argv[3:]
outs = {}
for tag, cmd in (("ref", [ref.CLI_PATH, "paired", d + "/idx", d + "/f1.fq", d + "/f2.fq", "-t", "1"] + extra),
                 ("emu", [os.path.join(ROOT, "tests", "emu", "_build", "snapgpu-sam-emu"), "paired", d + "/idx", d + "/f1.fq", d + "/f2.fq"] + extra)):
    t0 = time.time()
    r = subprocess.run(cmd + ["-o", d + "/%s.sam" % tag], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=3000)
    print(tag, r.returncode, round(time.time() - t0, 1), r.stdout.decode()[-150:].replace("\n", " | "))
    outs[tag] = [l for l in open(d + "/%s.sam" % tag) if not l.startswith("@")]
bad = [(x, y) for x, y in zip(outs["ref"], outs["emu"]) if x != y]
print("records", len(outs["ref"]), len(outs["emu"]), "differing", len(bad))
flips = sum(1 for k in range(0, len(outs["ref"]) - 1, 2) if outs["ref"][k].split("\t")[0] == outs["ref"][k + 1].split("\t")[0] and (int(outs["ref"][k].split("\t")[1]) & 0x80))
print("pairs written second-mate-first:", flips)
for x, y in bad[:6]:
    print("REF", x.split("\t")[:9]); print("EMU", y.split("\t")[:9])
