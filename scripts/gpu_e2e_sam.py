"""MI355X: FASTQ -> SAM end to end on the bench workload (VERDICT r01 "report end-to-end"): the same FASTQ of N bench reads through
  (1) snap_amd/snapgpu-sam single        (C++ host program over the C ABI: alignment AND the SAM fields on the GPU),
  (2) oracle/_ref/snap-aligner-gpu       (SNAP's own CLI with shim/GpuAlignerExtension.cpp: alignment on the GPU, SAM text by the reference),
  (3) oracle/_ref/snap-aligner -t nproc  (the unmodified reference),
wall time of each process (index load included: 2.4 GB here) and the "reads/s" SNAP itself prints; the three SAM files must hold the same
records.  Uses the genome / index bench.py left under $SNAP_BENCH_DIR (run bench.py first).  Prints one JSON line."""
import hashlib, json, os, re, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snap_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
work = os.path.join(os.environ.get("SNAP_BENCH_DIR", "/tmp/snap_bench"), "g256_s20_seed20260925")
idx = os.path.join(work, "idx")
assert os.path.exists(os.path.join(idx, "GenomeIndex")), "run bench.py first"
genome = synth.make_genome(20260925, 256_000_000, n_contigs=24, repeat_frac=0.30, max_copies=5000, repeat_len=(200, 3000), max_divergence=0.05)
reads = synth.make_reads(20260925 + 1000, genome, n, 150)
fq = os.path.join(work, "e2e.fq")
t0 = time.time()
names = np.char.add("@r", np.arange(n).astype("U9")).astype("S")
with open(fq, "wb") as f:                       # one write per 50 000 reads
    for a in range(0, n, 50000):
        b = min(n, a + 50000)
        f.write(b"".join(names[i] + b"\n" + reads["bases"][i].tobytes() + b"\n+\n" + reads["quals"][i].tobytes() + b"\n" for i in range(a, b)))
t_fq = time.time() - t0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cores = os.cpu_count() or 8
out = {"reads": n, "fastq_write_s": t_fq}


def run(tag, cmd):
    sam = os.path.join(work, tag + ".sam")
    t0 = time.time()
    r = subprocess.run(cmd + ["-o", sam], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=400)
    dt = time.time() - t0
    txt = r.stdout.decode(errors="replace")
    m = re.findall(r"([\d,]+)\s*$", txt.strip().splitlines()[-1]) if txt.strip() else []
    h = hashlib.md5()
    nrec = 0
    recs = sorted(l for l in open(sam, "rb") if not l.startswith(b"@"))
    for l in recs:
        h.update(l); nrec += 1
    out[tag] = {"rc": r.returncode, "wall_s": dt, "reads_per_s_wall": n / dt, "records": nrec, "md5_sorted_records": h.hexdigest(),
                "tool_last_line": txt.strip().splitlines()[-1][:300] if txt.strip() else ""}
    os.remove(sam)


run("snapgpu_sam", [os.path.join(ROOT, "snap_amd", "snapgpu-sam"), "single", idx, fq, "-d", "8"])
run("snap_aligner_gpu_shim", [os.path.join(ROOT, "oracle", "_ref", "snap-aligner-gpu"), "single", idx, fq, "-d", "8", "-t", "8"])
run("snap_aligner_reference", [os.path.join(ROOT, "oracle", "_ref", "snap-aligner"), "single", idx, fq, "-d", "8", "-t", str(cores)])
out["identical_records"] = len({out[k]["md5_sorted_records"] for k in ("snapgpu_sam", "snap_aligner_gpu_shim", "snap_aligner_reference")}) == 1
os.remove(fq)
print(json.dumps(out))
