#!/usr/bin/env python
"""TEST INFRASTRUCTURE (build container; needs oracle/_ref): the inputs of scripts/gpu_r04_sam.sh under gpurun_in/sam_check/ (git-ignored; it
travels to the GPU box with the snapshot) -- a seeded 4 Mb genome indexed by the reference's own indexer, 100 000 x 150 bp reads, 10 000 pairs,
and the md5 of the reference CLI's SAM for each (-d 8 -t 1: its record order is the input's; every line but @PG), which the GPU box
compares snapgpu-sam's output with.  No Python runs on the box: a whole check is ~30 s of box time.

    python scripts/make_sam_check.py [n_reads=100000] [n_pairs=10000]
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snap_amd import synth          # noqa: E402
from oracle import ref              # noqa: E402


def md5_records(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for line in f:
            if not line.startswith(b"@PG"):
                h.update(line)
    return h.hexdigest()


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    d = os.path.join(ROOT, "gpurun_in", "sam_check")
    os.makedirs(d, exist_ok=True)
    g = synth.make_genome(4242, 4_000_000, n_contigs=3, repeat_frac=0.08)
    fasta = os.path.join(d, "g.fa")
    synth.write_fasta(fasta, g)
    index = os.path.join(d, "index")
    ref.build_index(fasta, index, seed_len=20, threads=max(1, min(8, os.cpu_count() or 1)))
    synth.write_fastq(os.path.join(d, "single.fq"), synth.make_reads(777, g, n_reads, 150))
    pr = synth.make_pairs(778, g, n_pairs, 150)
    files = [open(os.path.join(d, "r%d.fq" % (w + 1)), "wb") for w in (0, 1)]
    for i in range(n_pairs):
        for w in (0, 1):
            files[w].write(b"@pair%d/%d\n" % (i, w + 1) + pr["bases"][2 * i + w].tobytes() + b"\n+\n" + pr["quals"][2 * i + w].tobytes() + b"\n")
    for f in files:
        f.close()
    tmp = os.environ.get("TMPDIR", "/tmp")
    for tag, inputs in (("single", [os.path.join(d, "single.fq")]), ("paired", [os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")])):
        sam = os.path.join(tmp, "sam_check_%s_ref.sam" % tag)
        r = subprocess.run([ref.CLI_PATH, tag, index] + inputs + ["-o", sam, "-d", "8", "-t", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
        with open(os.path.join(d, tag + ".md5"), "w") as f:
            f.write(md5_records(sam) + "\n")
        print(tag, open(os.path.join(d, tag + ".md5")).read().strip())


if __name__ == "__main__":
    main()
