"""MI355X: FASTQ -> SAM end to end on the bench workload, at a size where the streaming rate shows (>= 20 M reads):
the same FASTQ of N bench reads through
  (1) snap_amd/snapgpu-sam single        (C++ host program over the C ABI: alignment AND the SAM fields on the GPU),
  (2) oracle/_ref/snap-aligner -t nproc  (the unmodified reference), unless --skip-reference,
  (3) oracle/_ref/snap-aligner-gpu       (SNAP's own CLI with shim/GpuAlignerExtension.cpp), only with --shim.
Per tool: wall time of the process, the rate AFTER the index is resident (snapgpu-sam prints it; the reference prints its own reads/s over
the alignment phase, AlignerContext.cpp:489-543), record count and an order-independent hash of the records (sum of xxh3-64 of every line):
the files must hold the same records.  Prints one JSON line.

    python scripts/gpu_e2e_sam.py [N=20000000] [--genome-mb 256] [--skip-reference] [--shim] [--threads T]
"""
import argparse, json, os, re, subprocess, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snap_amd import synth
import bench

ap = argparse.ArgumentParser()
ap.add_argument("n", nargs="?", type=int, default=20_000_000)
ap.add_argument("--genome-mb", type=int, default=256)
ap.add_argument("--skip-reference", action="store_true")
ap.add_argument("--shim", action="store_true")
ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
ap.add_argument("--keep", action="store_true")
ap.add_argument("--tool-args", default="", help="extra arguments for snapgpu-sam, one string")
ap.add_argument("--no-hash", action="store_true", help="do not hash the records of the sweep's runs (the first run is always hashed)")
a = ap.parse_args()
n = a.n

bargs = bench.parse_args(["--genome-mb", str(a.genome_mb)])
genome, idx, built, info = bench.ensure_index(bargs, 0, 0)
if built is not None:
    built.close()
work = os.path.dirname(idx)
fq = os.path.join(work, "e2e.fq")
out = {"reads": n, "genome_mb": a.genome_mb}

# ---- the FASTQ: bench reads (the same generator, 1 M at a time, drawn by several threads), fixed-width names, written as one byte matrix per piece
t0 = time.time()
L, PIECE = 150, 1_000_000
name_w = 11                                     # "r" + 10 digits
rec_w = 1 + name_w + 1 + L + 3 + L + 1


def piece(k):
    m = min(PIECE, n - k * PIECE)
    rd = synth.make_reads(20260925 + 1000 + 7919 * k, genome, m, L)
    rec = np.empty((m, rec_w), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    ids = np.arange(k * PIECE, k * PIECE + m, dtype=np.int64)
    for d in range(10):
        rec[:, 2 + 9 - d] = ord("0") + (ids // 10 ** d) % 10
    c = 1 + name_w
    rec[:, c] = 10; rec[:, c + 1:c + 1 + L] = rd["bases"]; c += 1 + L
    rec[:, c] = 10; rec[:, c + 1] = ord("+"); rec[:, c + 2] = 10; c += 3
    rec[:, c:c + L] = rd["quals"]; rec[:, c + L] = 10
    return rec


with open(fq, "wb") as f, ThreadPoolExecutor(max_workers=6) as ex:
    for rec in ex.map(piece, range((n + PIECE - 1) // PIECE)):
        f.write(rec.tobytes())
out["fastq_write_s"] = time.time() - t0
out["fastq_bytes"] = os.path.getsize(fq)


def hash_records(sam):
    import xxhash
    h, nrec = 0, 0
    with open(sam, "rb", buffering=1 << 24) as f:
        for line in f:
            if line[:1] == b"@":
                continue
            h = (h + xxhash.xxh3_64_intdigest(line)) & 0xFFFFFFFFFFFFFFFF
            nrec += 1
    return nrec, "%016x" % h


def run(tag, cmd, env=None, hash_it=True):
    sam = os.path.join(work, re.sub(r"[^A-Za-z0-9_.-]", "_", tag) + ".sam")
    t0 = time.time()
    r = subprocess.run(cmd + ["-o", sam], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=1500, env=env)
    dt = time.time() - t0
    txt = r.stdout.decode(errors="replace")
    t1 = time.time()
    nrec, hx = hash_records(sam) if (r.returncode == 0 and hash_it) else (0, "")
    o = {"rc": r.returncode, "wall_s": dt, "reads_per_s_wall": n / dt, "records": nrec, "records_hash": hx, "hash_s": time.time() - t1,
         "sam_bytes": os.path.getsize(sam) if os.path.exists(sam) else 0, "tool_tail": [l[:300] for l in txt.strip().splitlines()[-4:]]}
    m = re.search(r"index resident after ([\d.]+) s; FASTQ -> \w+ in ([\d.]+) s = (\d+) reads/s", txt)
    if m:
        o["index_load_s"], o["stream_s"], o["reads_per_s_streaming"] = float(m.group(1)), float(m.group(2)), float(m.group(3))
    m = re.findall(r"([\d,]+)\s*$", txt.strip().splitlines()[-1]) if txt.strip() else []
    out[tag] = o
    if not a.keep and os.path.exists(sam):
        os.remove(sam)


run("snapgpu_sam", [os.path.join(ROOT, "snap_amd", "snapgpu-sam"), "single", idx, fq, "-d", "8"] + a.tool_args.split(), env=dict(os.environ, SNAPGPU_SAM_VERBOSE="1"))
for k, extra in enumerate([x for x in os.environ.get("E2E_SWEEP", "").split(";") if x.strip()]):          # e.g. E2E_SWEEP="-b 1048576 -q 3;SNAPGPU_SAM_PIN=0 -q 4"
    toks = extra.split()
    envs = {t.split("=", 1)[0]: t.split("=", 1)[1] for t in toks if re.match(r"^[A-Z_0-9]+=", t)}          # leading NAME=VALUE words go to the environment
    run("snapgpu_sam #%d %s" % (k, extra.strip()), [os.path.join(ROOT, "snap_amd", "snapgpu-sam"), "single", idx, fq, "-d", "8"] + [t for t in toks if not re.match(r"^[A-Z_0-9]+=", t)],
        env=dict(os.environ, SNAPGPU_SAM_VERBOSE="1", **envs), hash_it=not a.no_hash)
if a.shim:
    run("snap_aligner_gpu_shim", [os.path.join(ROOT, "oracle", "_ref", "snap-aligner-gpu"), "single", idx, fq, "-d", "8", "-t", "8"])
if not a.skip_reference:
    run("snap_aligner_reference", [os.path.join(ROOT, "oracle", "_ref", "snap-aligner"), "single", idx, fq, "-d", "8", "-t", str(a.threads)])
    # the reference prints "... reads/s" in its summary line: total, aligned, ..., reads/s, time
    tail = out["snap_aligner_reference"]["tool_tail"][-1] if out["snap_aligner_reference"]["tool_tail"] else ""
    nums = re.findall(r"[\d,]+", tail)
    if len(nums) >= 2:
        try:
            out["snap_aligner_reference"]["reads_per_s_own_figure"] = int(nums[-2].replace(",", ""))
        except ValueError:
            pass
    out["identical_records"] = (out["snapgpu_sam"]["records_hash"] == out["snap_aligner_reference"]["records_hash"] and
                                out["snapgpu_sam"]["records"] == out["snap_aligner_reference"]["records"] == n)
    if out["snapgpu_sam"].get("reads_per_s_streaming") and out["snap_aligner_reference"].get("reads_per_s_own_figure"):
        out["speedup_streaming_vs_reference_own_figure"] = out["snapgpu_sam"]["reads_per_s_streaming"] / out["snap_aligner_reference"]["reads_per_s_own_figure"]
    out["speedup_wall"] = out["snap_aligner_reference"]["wall_s"] / out["snapgpu_sam"]["wall_s"]
if not a.keep:
    os.remove(fq)
print(json.dumps(out))
