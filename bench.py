#!/usr/bin/env python
"""bench.py -- aligned reads/s of the single-end hot path on N MI355X (BASELINE.json metric).

One "step" = one pass of BaseAligner::AlignRead (reference defaults, -d 8, seed 20) over a batch
of synthetic 150 bp reads whose bytes already sit in HBM when the clock starts; results stay in
HBM.  Workload = BASELINE.json configs[1] ("1M synthetic 150 bp single-end reads vs GRCh38,
seed=20, maxDist=8, 1xMI355X") with GRCh38 replaced by the seeded synthetic genome BASELINE.md
prescribes when GRCh38 is unavailable (no network here): --genome-mb Mb (default: 3100 = GRCh38
scale, a ~31 GB index built on the GPU in the reference's format and kept resident in HBM, when the
device has >= 64 GB free; 256 otherwise, and the line says so), 30 % of bases in planted repeat
families (copy number 2-5000, 0-5 % divergence).

The default one-GPU run adds three legs to the same JSON line (skip with --no-extra-legs):
  paired       configs[2] (2x150 bp pairs through the paired-end path) over the SAME resident
               index: value, ms_per_step, parity_check and cpu_baseline of its own
  c5           configs[4] on one GPU (2x250 bp pairs, -d 20, insert N(600, 80^2), 0.2 % long
               indels: the affine-gap code at limit 21), contexts with options of their own over
               the same resident index; parity_check and cpu_baseline of its own
  e2e          the PRODUCT rate: --e2e-reads reads as a FASTQ file through snap_amd/snapgpu-sam
               (FASTQ in, SAM out, index load reported apart), the reference CLI's own Reads/s on
               the first --e2e-ref-reads of them beside it, records compared by hash
  genome_256mb (--standin-leg) the single-end line on the 256 Mb stand-in of rounds 1-3

N > 1: one process per GPU (torch.distributed.run), reads sharded (each rank aligns its own
--reads reads: weak scaling), no data-path collective; the index is read by rank 0 and
broadcast once over RCCL into every rank's HBM before the clock starts.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     dominant kernel (k_align_single) against the HBM roofline: algorithmic bytes per
               launch (DESIGN.md "Algorithmic bytes") / average launch duration measured with
               hipEvents on the launch stream inside libsnapgpu.so
  cpu_baseline the compiled reference (oracle/_ref, unmodified SNAP 2.0.5) timed on this box's
               host cores on a bounded sample of the same reads (N = 1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

# (before torch starts the HIP runtime: eight hardware queues for the feeders' streams instead of four -- snapgpu.hip: snapgpu_hw_queues; libsnapgpu.so sets
#  the same default itself when it is loaded first, as in snapgpu-sam and the shim)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")


_T_PROCESS = time.time()


def log(*a):
    print("[bench +%.1fs]" % (time.time() - _T_PROCESS), *a, file=sys.stderr, flush=True)


def kernel_source_hash():
    """sha256, first 16 hex digits, over what the TIMED kernels are compiled from: the include closure (quoted includes, resolved relative to
    the including file, so include/snapgpu.h is in it) of the translation units that define k_align_single / k_align_paired
    (snap_amd/csrc/single_*_k.hip, paired_k.hip), snapgpu.hip -- the host side that sizes and launches them -- and the compile flags that are
    not fixed in __graft_entry__.build_library (SNAPGPU_BUILD_FLAGS).  What a committed PMC summary must carry for bench.py to replay its
    counters next to this build's timings (scripts/pmc_collect.py writes it).  The SAM-side and index-builder sources (cigar_k.hip,
    cigar_ag.h, sam_fields.h, index_build.*) are not part of those kernels and are not hashed."""
    import hashlib
    import re
    d = os.path.join(ROOT, "snap_amd", "csrc")
    roots = [os.path.join(d, f) for f in sorted(os.listdir(d)) if re.match(r"(single_.*_k|paired_k)\.hip$", f)] + [os.path.join(d, "snapgpu.hip")]
    seen, todo = set(), list(roots)
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        seen.add(f)
        if os.path.basename(f) == "snapgpu.hip":
            continue                                    # (its own text is hashed; what it includes beyond the align kernels' closure is the SAM side)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f, errors="replace").read(), re.M):
            todo.append(os.path.join(os.path.dirname(f), inc))
    h = hashlib.sha256()
    for f in sorted(seen):
        h.update(os.path.relpath(f, ROOT).encode()); h.update(open(f, "rb").read())
    h.update(("flags:" + " ".join(os.environ.get("SNAPGPU_BUILD_FLAGS", "").split())).encode())
    return h.hexdigest()[:16]


def ensure_index(args, rank, device=0, multi=False, need_dir=True):
    """Genome + SNAP index directory under /tmp, built once per box.  The directory is in the reference's on-disk format either way:
    --indexer gpu (default) builds it with this repo's GPU index builder (include/snapgpu.h: snapgpu_index_build_from_fasta; parity with
    the reference's builder: tests/test_zx_gpu_index_build.py) and ALSO returns the index still resident in HBM, so that the run aligns
    over it without reading the files back; --indexer reference runs the reference's own `snap-aligner index`,
    ~90 s at 256 Mb and ~20 minutes at GRCh38 scale.  The reference (cpu_baseline / parity_check) loads the same directory.
    N > 1 (`multi`): rank 0 draws the genome ONCE and leaves it in a file the other ranks map (they need it to draw their own reads),
    builds the index in its HBM and keeps it there for the RCCL broadcast -- no 31 GB directory is written or read back on that path
    (only a one-GPU run has a CPU baseline that needs one)."""
    from snap_amd import synth
    tag = "g%d_s%d_seed%d_%s" % (args.genome_mb, args.seed_len, args.seed, args.indexer)
    work = os.path.join(args.workdir, tag)
    done = os.path.join(work, "idx", "GenomeIndex")
    job = os.environ.get("MASTER_PORT", "0") + "_" + os.environ.get("TORCHELASTIC_RUN_ID", "x")
    ready = os.path.join(work, "ready_%s" % job)             # rank 0 -> the others: genome file written, index built
    gfile = os.path.join(work, "genome_%s" % job)
    t0 = time.time()
    if rank == 0 or not multi:
        genome = synth.make_genome(args.seed, args.genome_mb * 1_000_000, n_contigs=max(1, min(24, args.genome_mb // 8)),
                                   repeat_frac=0.30, max_copies=5000, repeat_len=(200, 3000), max_divergence=0.05)
        log("genome %d Mb generated in %.1fs" % (args.genome_mb, time.time() - t0))
        if multi:
            os.makedirs(work, exist_ok=True)
            np.concatenate([g for _, g in genome]).tofile(gfile + ".u8")
            json.dump([[nm, int(len(g))] for nm, g in genome], open(gfile + ".json", "w"))
    else:
        while not os.path.exists(ready):
            time.sleep(0.5)
        meta = json.load(open(gfile + ".json"))
        flat = np.memmap(gfile + ".u8", dtype=np.uint8, mode="r")
        genome, at = [], 0
        for nm, ln in meta:
            genome.append((nm, flat[at:at + ln])); at += ln
        log("rank %d: genome mapped from rank 0's file in %.1fs" % (rank, time.time() - t0))
    built, info = None, {"indexer": args.indexer, "cached": os.path.exists(done)}
    # A run whose directory is already there (an earlier run on this box) still BUILDS the index in HBM -- 9 s at 3.1 Gb against reading
    # 31 GB of files back and copying them up -- and just does not save it again; an N > 1 run never saves one.
    in_hbm_only = args.indexer == "gpu" and (os.path.exists(done) or (multi and not need_dir))
    if rank == 0 and (not os.path.exists(done) or in_hbm_only):
        os.makedirs(work, exist_ok=True)
        fa = os.path.join(work, "ref.fa")
        t1 = time.time()
        synth.write_fasta(fa, genome)
        info["s_fasta_written"] = time.time() - t1
        t1 = time.time()
        if args.indexer == "gpu":
            from snap_amd.index import build_index
            stats, built = build_index(fa, None if in_hbm_only else os.path.join(work, "idx"), seed_len=args.seed_len, device=device, keep=True)
            info.update(stats)
            info["s_build_and_save"] = time.time() - t1
            log("GPU index build%s: %.1fs (device %.0f ms: seeds %.0f, sort %.0f, runs %.0f, tables %.0f; FASTA read %.1fs)"
                % ("" if in_hbm_only else " + save", time.time() - t1, stats["ms_total_device"], stats["ms_keys"], stats["ms_sort"], stats["ms_runs"], stats["ms_tables"], stats["s_fasta"]))
        else:
            from oracle import ref          # reference index builder == the cpu_baseline's own set-up step
            ref.build_index(fa, os.path.join(work, "idx"), args.seed_len, threads=os.cpu_count() or 8)
            info["s_build_and_save"] = time.time() - t1
            log("reference index build: %.1fs" % (time.time() - t1))
        os.remove(fa)
    if multi and rank == 0:
        open(ready, "w").write("ok")
    return genome, os.path.join(work, "idx"), built, info


def summarize_launch_profile(lp):
    """Per-wave residency (out-of-reads clock minus first-read clock: the s_memtime counters of different XCDs are not synchronised, so
    only differences inside one wave mean anything) and the most expensive reads of a launch of the TIMED instantiation."""
    ok = (lp["wave_finish"] > 0) & (lp["wave_start"] > 0)
    dur = np.sort((lp["wave_finish"][ok].astype(np.int64) - lp["wave_start"][ok].astype(np.int64)))
    longest = max(1, int(dur[-1]))
    worst_i = np.argsort(lp["wave_worst_read_cycles"])[::-1][:8]
    return {"waves": int(dur.size), "longest_wave_cycles": longest,
            "wave_residency_fraction_of_longest": {"p10": float(dur[int(0.10 * dur.size)]) / longest, "p50": float(dur[dur.size // 2]) / longest,
                                                   "p90": float(dur[int(0.90 * dur.size)]) / longest, "p99": float(dur[int(0.99 * dur.size)]) / longest},
            "mean_wave_residency": float(dur.mean()) / longest,
            "worst_reads_fraction_of_longest_wave": [float(lp["wave_worst_read_cycles"][i]) / longest for i in worst_i],
            "worst_reads_ag_calls": [int(lp["wave_worst_read_ag_calls"][i]) for i in worst_i],
            "read_cycles_log2_hist": {str(i): int(v) for i, v in enumerate(lp["read_cycles_log2_hist"]) if v}}


def algorithmic_bytes(c, read_len, n_reads, ref_walk_slots=None):
    """SURVEY.md 8(d) / DESIGN.md: bytes the algorithm is entitled to move for the work done.  The probe term is that of the REFERENCE's
    slot walk (8 B per slot its quadratic / linear probe examines), whatever layout the kernel actually reads: `ref_walk_slots` is that
    count for this batch (bench.py counts it with one launch on a context that keeps the reference layout); the kernel's own counter
    (with the bucket tables: 8 x the 64-byte lines read) is reported separately as `probe_layout_bytes`."""
    slots = c["n_hash_slots_probed"] if ref_walk_slots is None else ref_walk_slots
    probe = 8 * slots + 4 * c["n_overflow_lists"] + 4 * c["n_hits_consumed"]
    lv = c["n_lv_ref_bytes"]
    reads = (2 * read_len + 88) * n_reads
    # what the kernel actually stages per scored location since round 3: 27 plane words (216 B: planes.h) for Landau-Vishkin, and the byte
    # window (read + 2 x 128 B) only where affine gap runs -- against (read - seed + 2k) reference BYTES the algorithm is entitled to
    staged = 216 * c["n_lv_locations"] + (read_len + 256) * c["n_ag_locations"]
    return probe + lv + reads, dict(probe=probe, lv_reference=lv, reads_and_results=reads, lv_reference_staged_bytes=staged,
                                    probe_layout_bytes=8 * c["n_hash_slots_probed"] + 4 * c["n_overflow_lists"] + 4 * c["n_hits_consumed"],
                                    probe_basis="reference slot walk, counted" if ref_walk_slots is not None else "the kernel's own table layout")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (0 = auto: 6, a multiple of the feeders)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mb", type=int, default=0,
                    help="synthetic genome size in Mb; 0 = auto: 3100 (GRCh38 scale, ~31 GB index in HBM) when the device has >= 64 GB free and the "
                         "host >= 48 GB available, else the 256 Mb stand-in")
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--seed-len", type=int, default=20)
    ap.add_argument("--max-k", type=int, default=8)
    ap.add_argument("--seed", type=int, default=20260925)
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads (pairs) for the CPU baseline (0 = auto: >= 15 s of reference work, over as many of the rotated batches as that takes)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target seconds of reference work for the CPU baseline when --cpu-sample is 0")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every GPU aligns --reads reads per step; strong: --reads is the whole job's batch, split evenly over the GPUs")
    ap.add_argument("--feeders", type=int, default=0,
                    help="contexts per GPU, each with its own stream and result buffer, that take the steps in turn so that consecutive "
                         "batches overlap on the GPU (what snapgpu-sam's feeder threads do).  0 = auto: 3 "
                         "(a launch ends with a tail of few, heavy reads / pairs that leaves most of the chip idle: measured, profiles/r02i, "
                         "1 -> 2 feeders: 4.20 -> 6.28 M reads/s single-end; 1 / 2 / 3 / 4 feeders: 119 / 174 / 210 / 189 k reads/s paired-end)")
    ap.add_argument("--batches", type=int, default=6, help="distinct read batches rotated through the timed steps")
    ap.add_argument("--insert-mean", type=float, default=400.0, help="paired: insert size mean (C3: 400; C5 as SURVEY.md 8(d) defines it: 600)")
    ap.add_argument("--insert-sd", type=float, default=50.0, help="paired: insert size s.d. (C3: 50; C5: 80)")
    ap.add_argument("--long-indel-frac", type=float, default=0.0, help="paired: fraction of reads with one extra indel event of length 1-10 (C5: 0.002)")
    ap.add_argument("--skip-refwalk", action="store_true", help="skip the untimed launches that count the reference's slot walk (roofline numerator)")
    ap.add_argument("--skip-breakdown", action="store_true", help="skip the untimed launch with phase timers (roofline.wave_cycle_breakdown)")
    ap.add_argument("--indexer", choices=["gpu", "reference"], default="gpu", help="who builds the index directory (see ensure_index)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-probe", action="store_true", help="skip the stand-alone index-probe measurement (roofline.probe)")
    ap.add_argument("--workdir", default=os.environ.get("SNAP_BENCH_DIR", "/tmp/snap_bench"))
    ap.add_argument("--workload", choices=["single", "paired"], default="single",
                    help="single = configs[1] (the metric's config); paired = configs[2], 2x150 bp FR pairs through the paired-end path")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="one GPU, --workload single only: do not add the short paired-end leg (`paired`) and the 256 Mb leg (`genome_256mb`)")
    ap.add_argument("--paired-leg-steps", type=int, default=12, help="timed steps of the extra paired-end leg")
    ap.add_argument("--standin-mb", type=int, default=256, help="genome size of the extra `genome_256mb` leg (tests shrink it)")
    ap.add_argument("--standin-leg", action="store_true", help="add the `genome_256mb` leg (the line of rounds 1-3 on the 256 Mb stand-in; in the default run until round 4)")
    ap.add_argument("--no-c5-leg", action="store_true", help="do not add the `c5` leg (configs[4] on one GPU: 2 x 250 bp pairs, -d 20, insert N(600, 80^2), 0.2 % long indels)")
    ap.add_argument("--c5-leg-steps", type=int, default=12)
    ap.add_argument("--c5-reads", type=int, default=200_000, help="reads per step of the c5 leg")
    ap.add_argument("--no-e2e-leg", action="store_true", help="do not add the `e2e` leg (FASTQ -> SAM through snap_amd/snapgpu-sam)")
    ap.add_argument("--e2e-reads", type=int, default=20_000_000, help="reads of the e2e leg's FASTQ")
    ap.add_argument("--e2e-ref-reads", type=int, default=1_000_000, help="leading reads of that FASTQ the reference CLI aligns beside it (its own Reads/s; records compared)")
    ap.add_argument("--e2e-tool-args", default="", help="extra arguments for snapgpu-sam in the e2e leg, one string")
    ap.add_argument("--e2e-passes", type=int, default=3, help="passes of snapgpu-sam over the e2e FASTQ with the index resident (`-passes`): value = the median pass, min / max beside it")
    args = ap.parse_args(argv)
    if args.steps <= 0:
        args.steps = 6
    return args


class Bed:
    """One genome's set-up, shared by the legs that run over it: the synthetic genome, the index directory (the reference's format), the
    index resident in HBM and the context that owns it (`owner`; every other context of a leg is a replica over the same blobs)."""
    pass


def make_bed(args, env, paired_owner):
    import torch
    from snap_amd import abi
    from snap_amd import dist as sd
    from snap_amd.aligner import BaseAligner, ChimericPairedEndAligner
    from snap_amd.index import GenomeIndex
    rank, world, local_rank, dev = env["rank"], env["world"], env["local_rank"], env["dev"]
    bed = Bed()
    bed.args = args
    bed.pparams = abi.default_paired_params()
    cls = ChimericPairedEndAligner if paired_owner else BaseAligner
    # The index directory is built BEFORE the process group exists: a build can take minutes at GRCh38 scale, and a rank that sits in
    # a collective that long runs into the NCCL watchdog.  Ranks other than 0 wait for the directory's last file on the file system.
    # (the directory itself is only needed where the reference runs beside the GPU: the CPU baseline of a one-process run)
    bed.genome, bed.idx_dir, built, bed.index_info = ensure_index(args, rank, local_rank, multi=world > 1 or env["force_dist"],
                                                                  need_dir=world == 1 and not args.skip_cpu)
    if (world > 1 or env["force_dist"]) and env.get("dist") is None:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env["dist"] = sd.init_process_group("nccl")
        env["dist"].barrier()
    dist = env.get("dist")
    bed.params = bed.owner_params = abi.default_params(max_k=args.max_k, max_read_len=((args.read_len + 15) // 16) * 16)
    t0 = time.time()
    index = None
    if dist is None and built is not None:      # the index this process has just built is still in HBM: adopt it, no file is read back
        bed.owner = cls.from_built_index(built, None, bed.params, device=local_rank, paired_params=bed.pparams if paired_owner else None)
        bed.keep = built
        bed.index_bytes = int(bed.index_info["hash_blob_bytes"] + 4 * bed.index_info["overflow_table_size"] + bed.index_info["n_bases"] + 2048)
    elif dist is None:
        index = GenomeIndex.load_from_directory(bed.idx_dir)
        bed.owner = cls(index, bed.params, bed.pparams, device=local_rank) if paired_owner else cls(index, bed.params, device=local_rank)
        bed.keep = None
    else:
        # N > 1: the index rank 0 has just built is in ITS HBM: the RCCL broadcast reads it there (no file, no host copy); a rank 0 without
        # one (--indexer reference) loads the directory and sends it up in pinned pieces (snap_amd/dist.py)
        src_ptrs = None
        if rank == 0 and built is not None:
            v = built.view()
            index = sd.index_meta_from_view(v)
            src_ptrs = (int(v.hash_blob), int(v.overflow), int(v.genome) - int(v.genome_pad))
        elif rank == 0:
            index = GenomeIndex.load_from_directory(bed.idx_dir)
        index, blobs = sd.broadcast_index(index, dev, src_device_ptrs=src_ptrs)          # RCCL broadcast HBM -> HBM
        bed.keep = (blobs, built)                  # (rank 0's tensors are views of the built index: it must outlive them)
        ptrs = (blobs[0].data_ptr(), blobs[1].data_ptr(), blobs[2].data_ptr())
        bed.owner = (cls(index, bed.params, bed.pparams, device=local_rank, device_index_ptrs=ptrs) if paired_owner
                     else cls(index, bed.params, device=local_rank, device_index_ptrs=ptrs))
    if index is not None:
        hb_, ow_, gb_ = getattr(index, "_device_sizes", (index.hash_blob.size, index.overflow.size, index.genome_padded.size))
        bed.index_bytes = int(hb_) + 4 * int(ow_) + int(gb_)
    bed.index_resident_s = time.time() - t0          # adopting / loading / broadcasting the blobs (the build itself: config.index_build)
    bed.index_path = ("adopted from the GPU build" if dist is None and built is not None else "loaded from the directory" if dist is None
                      else "RCCL broadcast from rank 0's HBM (built there)" if built is not None or rank != 0 else "RCCL broadcast, rank 0 loaded the directory")
    log("rank %d: %d Mb genome, index (%.1f GB) resident in HBM after %.1fs" % (rank, args.genome_mb, bed.index_bytes / 1e9, bed.index_resident_s))
    return bed


def close_bed(bed):
    import torch
    bed.owner.close()
    keep = bed.keep
    bed.keep = None
    if isinstance(keep, tuple):                    # (blob tensors, the built index they may be views of)
        blobs_, built_ = keep
        del blobs_
        keep = built_
    if keep is not None and hasattr(keep, "close"):
        keep.close()
    del keep
    bed.genome = None
    bed.ref_index = None
    torch.cuda.empty_cache()


def run_leg(args, env, bed, workload, primary):
    """One leg = warm-up + exactly args.steps timed steps of `workload` over `bed`, then (rank 0) the untimed diagnostics, the CPU baseline
    and the parity check.  Returns the JSON object on rank 0, None elsewhere."""
    import torch
    from snap_amd import abi, synth
    from snap_amd import dist as sd
    from snap_amd.aligner import BaseAligner, ChimericPairedEndAligner
    rank, world, local_rank, dev, dist = env["rank"], env["world"], env["local_rank"], env["dev"], env.get("dist")
    paired = workload == "paired"
    params, pparams, genome, idx_dir, index_info, index_bytes = bed.params, bed.pparams, bed.genome, bed.idx_dir, bed.index_info, bed.index_bytes
    owner_is_paired = isinstance(bed.owner, ChimericPairedEndAligner)

    # a leg whose options differ from the owner's (the c5 leg: -d 20, 250 bp reads) gets contexts of its own over the owner's resident index
    p_over = params if params is not getattr(bed, "owner_params", params) else None

    def new_context():
        if paired and not owner_is_paired:
            return ChimericPairedEndAligner.over(bed.owner, pparams, params=p_over)
        return bed.owner.replica(params=p_over)

    # --batches DISTINCT read batches rotate through the timed steps (step k aligns batch k mod B), so that no step finds the previous
    # step's probes / reference windows in L2 or MALL by construction.  Batch 0 is the first one the parity check and the CPU baseline use.
    n_batches = max(1, min(args.batches, args.steps))

    def make_batch(b):
        sd_ = args.seed + 1000 + rank + 7919 * b
        if paired:      # FR pairs; C3: insert N(400, 50^2) clipped to [150, 1000]; C5: N(600, 80^2) + 0.2 % indel events of length 1-10
            return synth.make_pairs(sd_, genome, args.reads // 2, args.read_len, insert_mean=args.insert_mean, insert_sd=args.insert_sd,
                                    long_indel_frac=args.long_indel_frac)
        return synth.make_reads(sd_, genome, args.reads, args.read_len)   # 1% sub, .05% ins/del, 50% RC, Q20-40
    t0 = time.time()
    from concurrent.futures import ThreadPoolExecutor
    # (numpy releases the GIL in the large operations: the batches are drawn side by side -- ~5 GB of temporaries each, so fewer at a time
    #  when several ranks share the host)
    with ThreadPoolExecutor(max_workers=max(1, min(n_batches, 6 if world == 1 else 2))) as ex:
        batches = list(ex.map(make_batch, range(n_batches)))
    log("%s: %d read batch(es) generated in %.1fs" % (workload, n_batches, time.time() - t0))
    reads = batches[0]
    n = args.reads                      # reads per GPU per step (a pair is two reads)
    n_units = n // 2 if paired else n   # alignment problems per launch
    res_dtype = abi.PAIRED_RESULT_DTYPE if paired else abi.RESULT_DTYPE
    d_batches = [(torch.from_numpy(b_["bases"].reshape(-1)).to(dev), torch.from_numpy(b_["quals"].reshape(-1)).to(dev),
                  torch.from_numpy(b_["offsets"].astype(np.int64)).to(dev)) for b_ in batches]
    d_bases, d_quals, d_offs = d_batches[0]
    # Feeders: contexts over the one resident index (snapgpu_create_replica, share_index), each with its own stream, slabs and result
    # buffer.  Feeder f runs steps f, f + F, f + 2F, ... from a host thread of its own (the C ABI call blocks until its batch is done),
    # so the tail of one batch -- a few heavy pairs on a few wavefronts -- overlaps the bulk of the next.  A step is still one pass of
    # the hot path over one batch, and exactly --steps of them are inside the timed region.
    n_feed = args.feeders if args.feeders > 0 else 3
    n_feed = max(1, min(n_feed, max(1, args.steps)))
    own_first = paired == owner_is_paired and p_over is None      # the bed's owner is itself a context of this leg's kind (and options)
    feeders = ([bed.owner] if own_first else []) + [new_context() for _ in range(n_feed - (1 if own_first else 0))]
    aligner = feeders[0]
    d_prims = [torch.zeros(n_units * res_dtype.itemsize, dtype=torch.uint8, device=dev) for _ in range(n_feed)]
    d_prim = d_prims[0]
    torch.cuda.synchronize()

    call_ms = []                        # host-side duration of every align call of the timed region (the call blocks until its launch is done)
    event_ms = []                       # hipEvent duration of the same launches, one entry per call (read from the context right after the call)

    def run_steps(k_steps, only_batch=None, record=None):
        def feed(f):
            for k in range(f, k_steps, n_feed):
                db, dq, do = d_batches[(k % n_batches) if only_batch is None else only_batch]
                t_c = time.perf_counter()
                feeders[f].align_device(n_units, db.data_ptr(), dq.data_ptr(), do.data_ptr(), d_prims[f].data_ptr())
                if record is not None:
                    record.append(1e3 * (time.perf_counter() - t_c))      # (list.append is atomic under the GIL)
                    ms_k, nl_k = feeders[f].kernel_time(reset=True)       # this context's launches since its last call: this call's
                    event_ms.append((ms_k, nl_k))
        if n_feed == 1:
            feed(0)
            return
        import threading
        errs = []

        def guarded(f):
            try:
                feed(f)
            except BaseException as e:          # noqa: BLE001 -- re-raised in the main thread
                errs.append(e)
        ts = [threading.Thread(target=guarded, args=(f,)) for f in range(n_feed)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]

    run_steps(max(args.warmup, n_feed if args.warmup else 0))      # (every feeder gets at least one warm-up batch)
    for a_ in feeders:
        a_.counters(reset=True)
        a_.kernel_time(reset=True)

    # ---------------------------------------------------------------- timed region
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    run_steps(args.steps, record=call_ms)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        elapsed = sd.max_over_ranks(elapsed, dev)
    log("%s: %d timed steps in %.2fs" % (workload, args.steps, elapsed))

    counters = {}
    kernel_ms, launches = sum(m_ for m_, _ in event_ms), sum(n_ for _, n_ in event_ms)
    for a_ in feeders:
        for k_, v_ in a_.counters().items():
            counters[k_] = counters.get(k_, 0) + v_
    # a dedicated, untimed step: EVERY feeder aligns batch 0 -- their results must be the same bytes, and they are what the parity check
    # below compares with the reference
    run_steps(n_feed, only_batch=0)
    torch.cuda.synchronize()
    prim = np.frombuffer(d_prim.cpu().numpy().tobytes(), dtype=res_dtype)
    for d_other in d_prims[1:]:
        if not torch.equal(d_other, d_prim):
            raise SystemExit("bench.py: two feeders disagree on the same batch")

    def close_leg():
        for a_ in feeders:
            if a_ is not bed.owner:
                a_.close()
    if rank != 0:
        close_leg()
        return None

    def gpu_results_of_batch(b):          # untimed: the timed kernel's answer for batch b (parity check over more than batch 0)
        if b == 0:
            return prim
        db, dq, do = d_batches[b]
        aligner.align_device(n_units, db.data_ptr(), dq.data_ptr(), do.data_ptr(), d_prims[0].data_ptr())
        torch.cuda.synchronize()
        return np.frombuffer(d_prims[0].cpu().numpy().tobytes(), dtype=res_dtype)

    # ---------------------------------------------------------------- the reference's slot walk, counted (the numerator SURVEY.md 8(d) defines)
    # The timed contexts probe the device-native bucket tables (one 64-byte line per strand); the ALGORITHMIC bytes of a lookup are those of
    # the reference's walk over its own slot arrays (8 B per slot examined).  One untimed launch per distinct batch on a context that keeps
    # the reference layout (SNAPGPU_NO_BUCKETS=1) counts exactly those slots for exactly these reads.
    ref_walk_slots = None
    walker = None
    if not args.skip_refwalk:
        os.environ["SNAPGPU_NO_BUCKETS"] = "1"
        try:
            walker = new_context()
        finally:
            del os.environ["SNAPGPU_NO_BUCKETS"]
        walker.counters(reset=True)
        for db, dq, do in d_batches:
            walker.align_device(n_units, db.data_ptr(), dq.data_ptr(), do.data_ptr(), d_prims[0].data_ptr())
        wc = walker.counters(reset=True)
        ref_walk_slots = wc["n_hash_slots_probed"] / len(d_batches)
    # ---------------------------------------------------------------- where the wave cycles go: one launch of the instantiation that carries the phase timers
    breakdown = None
    if not args.skip_breakdown and not paired:
        os.environ["SNAPGPU_PHASE_TIMERS"] = "1"
        try:
            timed = aligner.replica()
        finally:
            del os.environ["SNAPGPU_PHASE_TIMERS"]
        timed.counters(reset=True)
        timed.align_device(n_units, d_bases.data_ptr(), d_quals.data_ptr(), d_offs.data_ptr(), d_prims[0].data_ptr())
        tc = timed.counters(reset=True)
        try:        # where the launch's time goes wave by wave: finish-time distribution and the most expensive reads (profiles/)
            lp = timed.launch_profile()
            launch_profile = summarize_launch_profile(lp)
        except Exception as e_:          # noqa: BLE001 -- diagnostics only
            launch_profile = {"error": str(e_)}
        tot_c = max(1, tc.get("cycles_total", 0))
        breakdown = {"fractions": {k[7:]: tc[k] / tot_c for k in ("cycles_lookup", "cycles_hits", "cycles_lv", "cycles_ag") if k in tc},
                     "wave_cycles_per_read": tc.get("cycles_total", 0) / max(1, tc["n_reads"]),
                     "launch_profile": launch_profile,
                     "source": "one untimed launch of k_align_single<3, false, true, TIMED> (SNAPGPU_PHASE_TIMERS=1) on batch 0; the timed kernels carry no timers"}
        timed.close()

    # ---------------------------------------------------------------- the index-probe kernel on its own (north_star: HBM roofline of the probe)
    # k_lookup_seeds -- the lookupSeed32 entry of the C ABI, NOT a stage of AlignRead (the align kernel probes inline) -- over >= 10^7 seeds
    # drawn from the bench reads (every 13th offset of every read), hit lists read as BaseAligner consumes them (at most -h 300 per
    # direction) but not stored; timed with hipEvents on the launch stream inside libsnapgpu.so.  Numerator = the REFERENCE's slot walk for
    # these seeds (8 B per slot its probe sequence examines, counted by the same kernel on a context that keeps the reference's table
    # layout), as SURVEY.md 8(d) defines it; the 64-byte bucket lines the timed kernel actually reads are reported beside it.
    probe = None
    if not paired and not args.skip_probe:
        L, S = args.read_len, args.seed_len
        offs13 = np.arange(0, L - S + 1, 13)
        rb = reads["bases"].reshape(n, L)
        seeds = np.ascontiguousarray(np.stack([rb[:, o:o + S] for o in offs13], axis=1).reshape(-1, S))
        n_seeds = seeds.shape[0]
        d_seeds = torch.from_numpy(seeds.reshape(-1)).to(dev)
        d_nh = torch.zeros(2 * n_seeds, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        aligner.lookup_device(n_seeds, d_seeds.data_ptr(), d_nh.data_ptr(), 0, 300)          # warm-up
        aligner.counters(reset=True); aligner.kernel_time(reset=True)
        reps = 5
        for _ in range(reps):
            aligner.lookup_device(n_seeds, d_seeds.data_ptr(), d_nh.data_ptr(), 0, 300)
        pc = aligner.counters(reset=True); pms, pl = aligner.kernel_time(reset=True)
        per_rep = {k_: v_ / reps for k_, v_ in pc.items()}
        lists_hits_io = 4 * per_rep["n_overflow_lists"] + 4 * per_rep["n_hits_consumed"] + S * n_seeds + 16 * n_seeds
        layout_bytes = 8 * per_rep["n_hash_slots_probed"] + lists_hits_io          # 8 x 8 B = the 64-byte bucket line per strand
        ref_slots = None
        if walker is not None:
            walker.counters(reset=True)
            walker.lookup_device(n_seeds, d_seeds.data_ptr(), d_nh.data_ptr(), 0, 300)
            ref_slots = walker.counters(reset=True)["n_hash_slots_probed"]
        pb = (8 * ref_slots + lists_hits_io) if ref_slots is not None else layout_bytes
        pavg = pms / max(1, pl)
        probe = {"kernel": "k_lookup_seeds (the C ABI's lookupSeed32 entry; not a stage of AlignRead, which probes inline)",
                 "seeds_per_launch": n_seeds, "avg_launch_ms": pavg, "lookups_per_s": n_seeds / (pavg * 1e-3),
                 "algorithmic_bytes_per_launch": pb, "achieved": pb / (pavg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": pb / (pavg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "numerator_basis": "reference slot walk, counted (8 B per slot examined)" if ref_slots is not None else "the kernel's own table layout (64-byte bucket lines)",
                 "bucket_line_bytes_per_launch": layout_bytes, "frac_bucket_lines": layout_bytes / (pavg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "ref_slots_per_lookup": (ref_slots / max(1, per_rep["n_hash_table_lookups"])) if ref_slots is not None else None,
                 "hits_per_lookup": per_rep["n_hits_consumed"] / max(1, per_rep["n_hash_table_lookups"]),
                 # the sector-level figure SURVEY.md 8(d) asks for next to the algorithmic one: every slot walk touches whole 64 B lines.
                 # Lower bound on lines: one per direction for the walk + one per 16 hits read + one for the seed text / counts
                 "min_64B_lines_per_lookup": 2 + per_rep["n_hits_consumed"] / max(1, per_rep["n_hash_table_lookups"]) / 16 + per_rep["n_overflow_lists"] / max(1, per_rep["n_hash_table_lookups"]),
                 "note": "algorithmic bytes = 8 B per slot the reference's walk examines + 4 B per overflow count word + 4 B per hit read + seed text in + hit counts out"}
        del d_seeds, d_nh
    if walker is not None:
        walker.close()

    # ---------------------------------------------------------------- report (rank 0)
    total_reads = n * world * args.steps
    value = total_reads / elapsed
    per_launch = {k: v / max(1, launches) for k, v in counters.items()}
    alg_bytes, parts = algorithmic_bytes(per_launch, args.read_len, n, ref_walk_slots)
    avg_ms = kernel_ms / max(1, launches)
    # (with several feeders the launches overlap, so each one's hipEvent time is longer than its share of the chip: the rate is then
    #  taken over the step time, bytes of one batch / (elapsed / steps))
    achieved = alg_bytes / ((avg_ms if n_feed == 1 else 1e3 * elapsed / args.steps) * 1e-3) / 1e9
    genome_desc = ("seeded synthetic %d Mb genome with 30%% planted repeats (GRCh38 scale: GRCh38 itself is unavailable here)" % args.genome_mb if args.genome_mb >= 3000
                   else "seeded synthetic %d Mb genome with 30%% planted repeats (stand-in: %s)" % (args.genome_mb, env.get("genome_choice", "--genome-mb")))
    out = {
        "metric": "aligned reads/sec (whole node), %d bp %s vs synthetic %d Mb genome (GRCh38 unavailable), seed=20, maxDist=%d"
                  % (args.read_len, "paired-end (2x%d FR pairs)" % args.read_len if paired else "single-end", args.genome_mb, args.max_k),
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "u8 bases / int32 DP / f64 match probability", "data": "synthetic",
        "config": {"workload": ("%s: %d pairs of 2 x %d bp (FR, insert N(%g,%g^2), long-indel fraction %g) per GPU per step, ChimericPairedEndAligner over IntersectingPairedEndAligner defaults (-n 8 -H 4000 -s 0 1000 -i 40, affine gap + soft clipping on), -d %d, index seed %d (directory in the reference's format), genome = %s"
                                % ("configs[4] on one GPU" if getattr(args, "pmc_tag", "") == "c5" else "configs[2]", n_units, args.read_len, args.insert_mean, args.insert_sd, args.long_indel_frac, args.max_k, args.seed_len, genome_desc)) if paired else
                               ("configs[1]: %d x %d bp single-end reads per GPU per step, BaseAligner::AlignRead defaults (-n 25 -h 300 -D 1, affine gap on, ALT-aware), -d %d, index seed %d (directory in the reference's format), genome = %s"
                                % (n, args.read_len, args.max_k, args.seed_len, genome_desc)),
                   "genome_mb": args.genome_mb, "genome_choice": env.get("genome_choice", "--genome-mb"),
                   "reads_per_gpu": n, "read_len": args.read_len, "index_bytes_hbm": index_bytes,
                   "parallelism": "reads sharded over %d GPU(s), index replicated%s" % (world, " by RCCL broadcast" if world > 1 else ""),
                   "feeders_per_gpu": n_feed, "index_build": index_info, "index_resident_s": getattr(bed, "index_resident_s", None),
                   "index_path": getattr(bed, "index_path", None), "kernel_source_hash": kernel_source_hash()},
        "roofline": {"kernel": "k_align_paired" if paired else "k_align_single", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "bytes_breakdown": parts, "avg_launch_ms": avg_ms,
                     # with F feeders F launches share the chip, so one launch's hipEvent duration (= rocprofv3's kernel average) is ~F x
                     # its share: `achieved` is then bytes per batch / time per batch; the per-launch figure is kept beside it
                     "achieved_basis": "bytes per launch / hipEvent launch duration" if n_feed == 1 else "bytes per batch / (elapsed / steps): %d launches overlap" % n_feed,
                     "achieved_per_overlapped_launch": alg_bytes / (avg_ms * 1e-3) / 1e9,
                     # the same bytes over ONE launch's own duration (hipEvents = rocprofv3's kernel average), whatever else shares the chip
                     "frac_per_launch": alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "batches_rotated": n_batches,
                     "per_read": {"hash_lookups": per_launch["n_hash_table_lookups"] / n, "hash_slots": per_launch["n_hash_slots_probed"] / n,
                                  "hits": per_launch["n_hits_consumed"] / n, "lv_locations": per_launch["n_lv_locations"] / n,
                                  "ag_locations": per_launch["n_ag_locations"] / n}},
        "aligned_fraction": float((prim["status"] != 0).mean()),
    }
    if probe is not None:
        out["roofline"]["probe"] = probe
    if os.environ.get("SNAPGPU_PHASE_TIMERS") == "1" and not paired:      # a diagnostic run: the timed contexts themselves carry the timers
        try:
            out["roofline"]["launch_profile_of_last_timed_launch"] = summarize_launch_profile(aligner.launch_profile())
        except Exception as e_:          # noqa: BLE001
            out["roofline"]["launch_profile_of_last_timed_launch"] = {"error": str(e_)}
    if breakdown is not None:
        out["roofline"]["wave_cycle_breakdown"] = breakdown["fractions"]
        out["roofline"]["wave_cycles_per_read"] = breakdown["wave_cycles_per_read"]
        out["roofline"]["wave_cycle_breakdown_source"] = breakdown["source"]
        out["roofline"]["launch_profile"] = breakdown["launch_profile"]
    elif counters.get("cycles_total", 0):      # (a -DSNAPGPU_PHASE_TIMERS build of the paired kernels)
        tot = counters["cycles_total"]
        out["roofline"]["wave_cycle_breakdown"] = {k[7:]: counters[k] / tot for k in
                                                  ("cycles_lookup", "cycles_hits", "cycles_lv", "cycles_ag") + (("cycles_single_fallback",) if paired else ())
                                                  if k in counters}   # paired: lookup = Phase 1, hits = Phase 2 (set intersection), lv/ag = paired scoring
        out["roofline"]["wave_cycles_per_read"] = counters["cycles_total"] / max(1, counters["n_reads"])
    if paired:
        out["roofline"]["phase4_help"] = {"watchdog_events": counters.get("help_watchdog_events", 0), "min_candidates": os.environ.get("SNAPGPU_PAIRED_HELP_MIN", "64 (default)"),
                                          "lists_published": counters.get("help_lists_published", 0), "answers_used": counters.get("help_answers_used", 0)}
    else:       # se_help.h: forced walks published for idle waves, stored evaluations the owners' ordered walks took (summed over the timed launches)
        out["roofline"]["heavy_read_help"] = {"enabled": os.environ.get("SNAPGPU_SINGLE_HELP", "1") != "0", "lists_published": counters.get("help_lists_published", 0),
                                              "evaluations_taken": counters.get("help_answers_used", 0), "watchdog_events": counters.get("help_watchdog_events", 0),
                                              "evaluations_stored": counters.get("cycles_single_fallback", 0) >> 32,
                                              "refused_other_band_or_decision": (counters.get("cycles_single_fallback", 0) >> 16) & 0xffff,
                                              "refused_skipped_or_limit": counters.get("cycles_single_fallback", 0) & 0xffff}
    attach_pmc(out, args, getattr(args, "pmc_tag", workload), n, n_feed, avg_ms, elapsed)
    finish_roofline(out, call_ms, elapsed, args.steps, [m_ for m_, _ in event_ms])

    if world == 1 and not args.skip_cpu:
        from oracle import ref                                       # cpu_baseline leg only
        cores = os.cpu_count() or 1
        t0 = time.time()
        ri = getattr(bed, "ref_index", None)
        if ri is None:
            ri = bed.ref_index = ref.RefIndex(idx_dir)
            log("%s: reference loaded the index directory in %.1fs" % (workload, time.time() - t0))
        per_unit = 2 if paired else 1

        def sample_arrays(k):          # the first k alignment problems of batches 0, 1, ... concatenated (reads of a pair stay together)
            bs, qs, os_, base, left, b = [], [], [0], 0, k, 0
            while left > 0:
                bt = batches[b]
                take = min(left, n_units)
                ro = bt["offsets"].astype(np.int64)
                end = int(ro[per_unit * take])
                bs.append(bt["bases"].reshape(-1)[:end]); qs.append(bt["quals"].reshape(-1)[:end])
                os_.append(ro[1:per_unit * take + 1] + base)
                base += end; left -= take; b += 1
            return (np.concatenate(bs), np.concatenate(qs), np.concatenate([np.asarray(o_, dtype=np.int64).reshape(-1) for o_ in os_]).astype(np.uint64))

        def run_ref(k):             # k = alignment problems (reads, or pairs)
            sb, sq, so = sample_arrays(k)
            if paired:
                return ri.align_paired(params, pparams, sb, sq, so, threads=cores, stage=0), (sb, sq, so)
            return ri.align_single(params, sb, sq, so, threads=cores), (sb, sq, so)
        cap = n_units * n_batches
        sample = min(cap, args.cpu_sample or min(n_units, 50_000))
        (pr, _, _, secs), arrs = run_ref(sample)
        if not args.cpu_sample:         # grow the sample until it is >= --cpu-seconds of reference work (a small first sample under-estimates the
            for _ in range(3):          # rate: 256 threads each construct their aligner first), capped by the distinct batches there are
                if secs >= 0.9 * args.cpu_seconds or sample >= cap:
                    break
                sample = int(min(cap, max(2 * sample, sample / secs * args.cpu_seconds * 1.15)))
                (pr, _, _, secs), arrs = run_ref(sample)
        log("%s: reference aligned %d %s in %.1fs" % (workload, sample, "pairs" if paired else "reads", secs))
        out["cpu_baseline"] = {"value": per_unit * sample / secs, "unit": "reads/s", "cores": cores, "kind": "reference", "seconds": secs,
                               "sample": "first %d %s of the rotated batches (batch 0 first), %s via oracle/_ref (SNAP 2.0.5 built -O3), %d threads, align phase only"
                                         % (sample, "pairs" if paired else "reads", "ChimericPairedEndAligner::align" if paired else "BaseAligner::AlignRead", cores)}
        # The baseline's results double as a parity check of the timed GPU output: EVERY unit of the sample is compared, none excluded.
        # The reads / pairs whose banded affine-gap traceback left the band (`reserved` != 0) were redone on the GPU the way a newly
        # constructed reference aligner does them (exact replay); for those the expectation is the reference run with fresh objects
        # (oracle/ref_driver.cpp: ZeroedArena), because the long-lived objects of the timed run answer such reads from their history.
        gpu = np.concatenate([gpu_results_of_batch(b)[:min(n_units, sample - b * n_units)] for b in range((sample + n_units - 1) // n_units)])
        sb, sq, so = arrs
        flagged = (gpu["reserved"] & 0x3fffffff) != 0         # any traceback step outside the band (superset of what was replayed)
        replayed = ((gpu["flags"] & 4) != 0) if paired else ((gpu["reserved"] & 0x80000000) != 0)
        fi = np.nonzero(flagged)[0]
        history_dependent = 0
        if fi.size:
            ro = so.astype(np.int64)
            sel = np.concatenate([np.arange(per_unit * i, per_unit * i + per_unit) for i in fi])
            fb = np.concatenate([sb[ro[j]:ro[j + 1]] for j in sel])
            fq = np.concatenate([sq[ro[j]:ro[j + 1]] for j in sel])
            fo = np.concatenate([[0], np.cumsum([ro[j + 1] - ro[j] for j in sel])]).astype(np.uint64)
            with ref.fresh_objects():
                if paired:
                    fr = ri.align_paired(params, pparams, fb, fq, fo, threads=min(cores, 16), stage=0)[0]
                else:
                    fr = ri.align_single(params, fb, fq, fo, threads=min(cores, 16))[0]
            pr = pr.copy()
            if paired:
                from tests.pairs_util import compare_paired as _cp
                history_dependent = int(_cp(pr[fi], fr, verbose=0).sum())
            else:
                from tests.util import compare_results as _cr
                history_dependent = sum(1 for i, j in enumerate(fi) if _cr(pr[j:j + 1], fr[i:i + 1]))
            pr[fi] = fr
        if paired:
            from tests.pairs_util import compare_paired
            bad = compare_paired(pr, gpu, verbose=0)
            out["parity_check"] = {"pairs": sample, "mismatching_pairs": int(bad.sum()), "excluded": 0, "exact_replayed": int(replayed.sum()), "left_the_band": int(flagged.sum()),
                                   "of_which_reference_history_dependent": history_dependent}
        else:
            from tests.util import compare_results
            problems = compare_results(pr, gpu)
            out["parity_check"] = {"reads": sample, "mismatching_fields": problems, "excluded": 0, "exact_replayed": int(replayed.sum()), "left_the_band": int(flagged.sum()),
                                   "of_which_reference_history_dependent": history_dependent}
    close_leg()
    del d_batches, d_prims
    torch.cuda.empty_cache()
    return out


def attach_pmc(out, args, workload, n, n_feed, avg_ms, elapsed):
    """roofline.traffic / roofline.valu_issue: PMC counters need their own rocprofv3 --pmc passes (scripts/pmc_collect.py); what is reported
    here is REPLAYED from the committed summary of the last such pass -- and only when that summary was taken from THIS build's device
    sources (kernel_source_hash), on this workload, genome size and feeder count; otherwise both stay null and the line says why."""
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(pmc):
        return
    try:
        t = json.load(open(pmc))
        ents = t.get("entries", [t])
        want = kernel_source_hash()
        for e in ents:
            if e.get("workload", "single") != workload or e.get("reads_per_launch") != n or e.get("genome_mb") != args.genome_mb:
                continue
            if e.get("kernel_source_hash") != want:
                out["roofline"]["traffic_source"] = ("profiles/pmc_latest.json has a pass for this workload, but of other device sources (%s, this build: %s): not replayed"
                                                     % (e.get("kernel_source_hash", "no hash"), want))
                continue
            out["roofline"]["traffic"] = e.get("hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = ("replayed from profiles/pmc_latest.json (rocprofv3 --pmc passes '%s' of this build's device sources, hash %s, %s feeder(s)), not measured in this run"
                                                 % (e.get("source", "?"), want, e.get("feeders", "?")))
            out["roofline"]["traffic_fetch_write_kb"] = [e.get("fetch_size_kb"), e.get("write_size_kb")]
            if e.get("valu_insts_per_launch"):
                # the LV / affine-gap work is integer VALU, not HBM.  Wave-level VALU instructions of one launch
                # (rocprofv3 --pmc SQ_INSTS_VALU, profiles/) over this run's launch time, against the issue peak of the chip:
                # 256 CUs x 4 SIMD32s, a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md: v_fma_f32 wave64 =
                # 2 cycles), 2.4 GHz  =>  256 * 4 * 2.4e9 / 2 = 1228.8 G wave-instructions/s
                peak = 256 * 4 * 2.4e9 / 2 / 1e9
                ach = e["valu_insts_per_launch"] / ((avg_ms if n_feed == 1 else 1e3 * elapsed / args.steps) * 1e-3) / 1e9
                vi = {"achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak,
                      "valu_insts_per_read": e["valu_insts_per_launch"] / n, "salu_insts_per_read": e.get("salu_insts_per_launch", 0) / n,
                      "source": "instruction counts replayed from profiles/pmc_latest.json ('%s'); time per batch of this run" % e.get("source", "?")}
                if e.get("thread_cycles_valu") and e.get("valu_insts_per_launch"):
                    # SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = lanes active per VALU instruction (x4 cycles per wave64 instruction on a SIMD16 pass)
                    vi["active_lanes_per_valu_inst"] = e["thread_cycles_valu"] / e["valu_insts_per_launch"] / e.get("thread_cycles_per_lane_inst", 1.0)
                if e.get("active_inst_valu"):
                    # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md) and comes out at 1.0 per VALU instruction of this kernel:
                    # a wave64 integer VALU instruction keeps its SIMD's VALU busy for 4 cycles.  Busy fraction of the chip's 1 024 SIMDs
                    # over this run's time per batch, at the 2.4 GHz peak clock (a lower sustained clock makes the true figure higher)
                    t_batch = (avg_ms if n_feed == 1 else 1e3 * elapsed / args.steps) * 1e-3
                    vi["valu_busy_frac_at_2p4GHz"] = e["active_inst_valu"] * 4.0 / (256 * 4 * 2.4e9 * t_batch)
                    vi["quad_cycles_per_valu_inst"] = e["active_inst_valu"] / e["valu_insts_per_launch"]
                if e.get("wave_cycles"):
                    vi["active_fraction_of_wave_cycles"] = e.get("active_inst_any", 0) / e["wave_cycles"]
                    vi["waiting_fraction_of_wave_cycles"] = e.get("wait_inst_any", 0) / e["wave_cycles"]
                out["roofline"]["valu_issue"] = vi
                out["roofline"]["valu_issue_frac"] = vi["frac"]
                if e.get("salu_insts_per_launch"):
                    # The scalar unit: ONE scalar-ALU instruction can issue per CU per cycle (a CU's arbiter visits one of its four SIMDs
                    # each cycle and issues at most one instruction per category from that SIMD's waves; the CU has one scalar ALU --
                    # MI355X_MICROARCH.md / cdna_hip_programming.md "1 scalar unit per CU"), so the chip's scalar issue peak is
                    # 256 CUs x 2.4 GHz = 614.4 G wave-instructions/s (lower at the sustained clock, which makes the true fraction higher).
                    t_batch = (avg_ms if n_feed == 1 else 1e3 * elapsed / args.steps) * 1e-3
                    s_peak = 256 * 2.4e9 / 1e9
                    s_ach = e["salu_insts_per_launch"] / t_batch / 1e9
                    out["roofline"]["salu_issue"] = {"achieved": s_ach, "peak": s_peak, "unit": "G wave-instructions/s", "frac": s_ach / s_peak,
                                                     "salu_insts_per_read": e["salu_insts_per_launch"] / n,
                                                     "peak_derivation": "256 CUs x 1 scalar-ALU issue per CU per cycle x 2.4 GHz peak clock",
                                                     "source": vi["source"]}
                    out["roofline"]["salu_issue_frac"] = s_ach / s_peak
                if e.get("wave_cycles") and e.get("wait_any"):
                    out["roofline"]["wait_any_frac_of_wave_cycles"] = e["wait_any"] / e["wave_cycles"]
            if e.get("probe_fetch_size_kb") and "probe" in out["roofline"]:
                out["roofline"]["probe"]["traffic"] = e["probe_fetch_size_kb"] * 1024.0
                out["roofline"]["probe"]["traffic_source"] = "FETCH_SIZE of k_lookup_seeds, same passes"
            break
    except Exception as e_:          # noqa: BLE001 -- the replay is optional
        out["roofline"]["traffic_source"] = "profiles/pmc_latest.json unreadable: %s" % e_


def finish_roofline(out, call_ms, elapsed, steps, event_ms=()):
    """What the line says ABOUT its roofline numbers: which resource the kernel is closest to by the counters (`bound`), the figures a reader
    needs as top-level scalars of `roofline` (the driver's record keeps scalars), and the spread of the launches' durations."""
    r = out["roofline"]
    if call_ms:
        # two different quantities: how long the HOST waited in each blocking align call -- with other feeders' launches
        # queued on the same GPU in front of it -- and how long each call's launches lasted by hipEvents on their stream
        cm = sorted(call_ms)
        r["blocking_call_ms_min"], r["blocking_call_ms_median"], r["blocking_call_ms_max"] = cm[0], cm[len(cm) // 2], cm[-1]
        r["blocking_call_ms_basis"] = "host wall time of each blocking align call of the timed region (%d calls; the call's own launches AND whatever other feeders had queued in front)" % len(cm)
    if event_ms:
        em = sorted(event_ms)
        r["launch_event_ms_min"], r["launch_event_ms_median"], r["launch_event_ms_max"] = em[0], em[len(em) // 2], em[-1]
        r["launch_event_ms_basis"] = "hipEvent time of the launches of each align call of the timed region (%d calls); avg_launch_ms is their average per launch" % len(em)
    if isinstance(r.get("probe"), dict):
        r["probe_frac"] = r["probe"].get("frac")
        r["probe_frac_bucket_lines"] = r["probe"].get("frac_bucket_lines")
    lp = r.get("launch_profile")
    if isinstance(lp, dict) and "mean_wave_residency" in lp:
        r["mean_wave_residency"] = lp["mean_wave_residency"]
    fr = {"hbm": r.get("frac") or 0.0}
    if r.get("traffic"):
        r["traffic_frac_of_hbm_peak"] = r["traffic"] / (1e-3 * 1e3 * elapsed / steps) / 1e9 / HBM_PEAK_GBS
        r["traffic_over_algorithmic"] = r["traffic"] / max(1.0, r["algorithmic_bytes_per_launch"])
        fr["hbm (measured traffic)"] = r["traffic_frac_of_hbm_peak"]
    if r.get("valu_issue_frac") is not None:
        fr["valu_issue"] = r["valu_issue_frac"]
    if r.get("salu_issue_frac") is not None:
        fr["salu_issue"] = r["salu_issue_frac"]
    # `frac` / `achieved` / `peak` stay the HBM figures the metric asks for; `bound` names the roof the counters put the
    # kernel closest to, which for this integer / control-flow path is an instruction-issue roof, not the HBM one
    r["nominal_bound"] = "hbm"
    r["bound"] = max(fr, key=lambda k: fr[k]) if len(fr) > 1 else "hbm (no PMC pass of this build: issue fractions unknown)"
    r["bound_fractions"] = fr


def run_e2e(args, idx_dir, genome):
    """The `e2e` object: the product rate.  --e2e-reads bench reads are written as a FASTQ file, `snap_amd/snapgpu-sam single` (the C++ host
    program over the C ABI: FASTQ batches in, alignment AND SAM fields on the GPU, SAM text out) aligns them over the SAME index directory
    the line's own leg used (its own process: it loads the directory itself and says how long that took), and the unmodified reference CLI
    (oracle/_ref/snap-aligner, all host threads) aligns the first --e2e-ref-reads of them: its own `Reads/s` figure is the baseline beside
    `value`, and an order-independent hash of its records (sum of xxh3-64 of every line) must equal that of snapgpu-sam's records for the
    same reads.  Reference side of the path: FASTQ.h:67, ReadSupplierQueue.h:76, SAM.cpp:1898-2354."""
    import re
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from snap_amd import synth
    n, n_ref, L = args.e2e_reads, min(args.e2e_ref_reads, args.e2e_reads), args.read_len
    work = os.path.dirname(idx_dir)
    fq, fq_ref = os.path.join(work, "e2e.fq"), os.path.join(work, "e2e_ref.fq")
    o = {"metric": "FASTQ -> SAM reads/s (snap_amd/snapgpu-sam single, index resident, index load reported apart)", "unit": "reads/s", "reads": n,
         "genome_mb": args.genome_mb}
    t0 = time.time()
    PIECE, name_w = 1_000_000, 11                # names "r" + 10 digits
    rec_w = 1 + name_w + 1 + L + 3 + L + 1

    def piece(k):
        m = min(PIECE, n - k * PIECE)
        rd = synth.make_reads(args.seed + 500_000 + 7919 * k, genome, m, L)
        rec = np.empty((m, rec_w), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
        ids = np.arange(k * PIECE, k * PIECE + m, dtype=np.int64)
        for d in range(10):
            rec[:, 2 + 9 - d] = ord("0") + (ids // 10 ** d) % 10
        c = 1 + name_w
        rec[:, c] = 10; rec[:, c + 1:c + 1 + L] = rd["bases"]; c += 1 + L
        rec[:, c] = 10; rec[:, c + 1] = ord("+"); rec[:, c + 2] = 10; c += 3
        rec[:, c:c + L] = rd["quals"]; rec[:, c + L] = 10
        return rec.tobytes()
    with open(fq, "wb") as f, open(fq_ref, "wb") as fr, ThreadPoolExecutor(max_workers=min(12, os.cpu_count() or 4)) as ex:
        done = 0
        for raw in ex.map(piece, range((n + PIECE - 1) // PIECE)):
            f.write(raw)
            if done < n_ref:
                fr.write(raw[:min(len(raw), (n_ref - done) * rec_w)])
            done += len(raw) // rec_w
    o["fastq_write_s"] = time.time() - t0
    log("e2e: %d reads written as FASTQ in %.1fs" % (n, o["fastq_write_s"]))

    def hash_records(sam, first=None):
        import xxhash
        h, nrec, names_ok = 0, 0, True
        with open(sam, "rb", buffering=1 << 24) as f:
            for line in f:
                if line[:1] == b"@":
                    continue
                if first is not None:
                    if nrec >= first:
                        break
                    if int(line[1:11]) >= first:           # (snapgpu-sam writes in input order: the first `first` records are the subset's)
                        names_ok = False
                h = (h + xxhash.xxh3_64_intdigest(line)) & 0xFFFFFFFFFFFFFFFF
                nrec += 1
        return nrec, "%016x" % h, names_ok
    sam, sam_ref = os.path.join(work, "e2e.sam"), os.path.join(work, "e2e_ref.sam")
    tool = os.path.join(ROOT, "snap_amd", "snapgpu-sam")
    t0 = time.time()
    r = subprocess.run([tool, "single", idx_dir, fq, "-d", str(args.max_k), "-o", sam, "-passes", str(max(1, args.e2e_passes))] + args.e2e_tool_args.split(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       stdin=subprocess.DEVNULL, timeout=900, env=dict(os.environ, SNAPGPU_SAM_VERBOSE="1"))
    o["tool_wall_s"] = time.time() - t0
    txt = r.stdout.decode(errors="replace")
    o["tool_tail"] = [l[:600] for l in txt.strip().splitlines()[-(5 + max(1, args.e2e_passes)):]]
    if r.returncode != 0:
        raise RuntimeError("snapgpu-sam failed (%d): %s" % (r.returncode, txt[-600:]))
    m = re.search(r"index resident after ([\d.]+) s; FASTQ -> \w+ in ([\d.]+) s = (\d+) reads/s", txt)
    if not m:
        raise RuntimeError("snapgpu-sam printed no rate: %s" % txt[-400:])
    o["index_load_s"], o["stream_s"], o["value"] = float(m.group(1)), float(m.group(2)), float(m.group(3))
    passes = [float(x) for x in re.findall(r"pass \d+ of \d+: \d+ reads in [\d.]+ s = (\d+) reads/s", txt)]
    if passes:                                  # (-passes N: every pass streams the whole file over the resident index; the line's figure is the MEDIAN pass)
        ps = sorted(passes)
        o["pass_values"] = passes
        o["value_min"], o["value_max"], o["value_median"] = ps[0], ps[-1], ps[len(ps) // 2] if len(ps) % 2 else 0.5 * (ps[len(ps) // 2 - 1] + ps[len(ps) // 2])
        o["value"] = o["value_median"]
        o["min_over_median"] = o["value_min"] / o["value_median"] if o["value_median"] else None
    mw = re.search(r"wall per pass if alone: (.*)", txt)
    if mw:
        o["stage_wall_per_pass"] = mw.group(1)[:500]
    o["reads_per_s_wall_incl_index_load"] = n * max(1, len(passes)) / o["tool_wall_s"]
    log("e2e: snapgpu-sam %.0f reads/s (index load %.1fs, stream %.1fs)" % (o["value"], o["index_load_s"], o["stream_s"]))
    nrec, hx_sub, names_ok = hash_records(sam, first=n_ref)
    o["sam_bytes"] = os.path.getsize(sam)
    os.remove(sam); os.remove(fq)
    ref_tool = os.path.join(ROOT, "oracle", "_ref", "snap-aligner")
    if os.path.exists(ref_tool):
        t0 = time.time()
        r2 = subprocess.run([ref_tool, "single", idx_dir, fq_ref, "-d", str(args.max_k), "-t", str(os.cpu_count() or 8), "-o", sam_ref], stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=1200)
        o["reference_cli"] = {"wall_s": time.time() - t0, "reads": n_ref, "threads": os.cpu_count() or 8}
        t2 = r2.stdout.decode(errors="replace")
        tail = t2.strip().splitlines()[-1] if t2.strip() else ""
        nums = re.findall(r"[\d,]+", tail)          # the summary line: total, aligned ..., reads/s, time
        try:
            o["reference_cli"]["reads_per_s_own_figure"] = int(nums[-2].replace(",", ""))
        except Exception:          # noqa: BLE001
            o["reference_cli"]["tail"] = tail[:300]
        if r2.returncode == 0:
            nr2, hx_ref, _ = hash_records(sam_ref)
            o["records_compared"] = nr2
            o["records_hash"], o["reference_records_hash"] = hx_sub, hx_ref
            o["identical_records"] = bool(hx_sub == hx_ref and nr2 == nrec == n_ref and names_ok)
            if o["reference_cli"].get("reads_per_s_own_figure"):
                o["speedup_vs_reference_cli_own_figure"] = o["value"] / o["reference_cli"]["reads_per_s_own_figure"]
        else:
            o["reference_cli"]["error"] = t2[-400:]
        if os.path.exists(sam_ref):
            os.remove(sam_ref)
    else:
        o["reference_cli"] = {"error": "oracle/_ref/snap-aligner not built"}
    os.remove(fq_ref)
    return o


def parity_verdict(out):
    """Which legs of the line disagree with the reference (empty: none), and -- because the driver's record keeps scalars of `config` and
    `roofline` but only the key names of everything else -- the parity figures and the legs' values lifted into `config` as scalars."""
    cfg, fails = out["config"], []

    def one(tag, leg):
        if not isinstance(leg, dict):
            return
        if "error" in leg and "value" not in leg:
            cfg[tag + "error"] = str(leg["error"])[:200]
            fails.append("%sleg failed: %s" % (tag, str(leg["error"])[:120]))
            return
        if tag:
            cfg[tag + "value"] = leg.get("value")
        if isinstance(leg.get("cpu_baseline"), dict):
            cfg[tag + "cpu_baseline_value"] = leg["cpu_baseline"].get("value")
        pc = leg.get("parity_check")
        if isinstance(pc, dict):
            units = pc.get("reads", pc.get("pairs"))
            mm = pc.get("mismatching_pairs") if "mismatching_pairs" in pc else len(pc.get("mismatching_fields") or [])
            cfg[tag + "parity_units"], cfg[tag + "parity_mismatching"] = units, mm
            cfg[tag + "parity_exact_replayed"] = pc.get("exact_replayed")
            if mm:
                fails.append("%sparity_check: %s" % (tag, json.dumps(pc)[:300]))
    one("", out)
    for tag in ("paired", "c5", "genome_256mb"):
        if tag in out:
            one(tag + "_", out[tag])
    e = out.get("e2e")
    if isinstance(e, dict):
        if "error" in e and "value" not in e:
            cfg["e2e_error"] = str(e["error"])[:200]
            fails.append("e2e leg failed: %s" % str(e["error"])[:120])
        else:
            for k in ("value", "value_min", "value_median", "index_load_s", "records_compared", "identical_records"):
                if k in e:
                    cfg["e2e_" + k] = e[k]
            if e.get("identical_records") is False:
                fails.append("e2e: records differ from the reference CLI's (%s vs %s)" % (e.get("records_hash"), e.get("reference_records_hash")))
    cfg["parity_failures"] = len(fails)
    return fails


def compact_leg(o):
    """What an extra leg contributes to the line."""
    keep = {k: o[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step") if k in o}
    keep["config"] = {k: o["config"][k] for k in ("workload", "genome_mb", "feeders_per_gpu", "index_bytes_hbm") if k in o["config"]}
    r = o["roofline"]
    keep["roofline"] = {k: r[k] for k in ("kernel", "bound", "nominal_bound", "bound_fractions", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_over_algorithmic",
                                          "algorithmic_bytes_per_launch", "avg_launch_ms", "blocking_call_ms_min", "blocking_call_ms_median", "blocking_call_ms_max",
                                          "launch_event_ms_min", "launch_event_ms_median", "launch_event_ms_max",
                                          "achieved_basis", "per_read", "valu_issue", "valu_issue_frac", "salu_issue", "salu_issue_frac", "wait_any_frac_of_wave_cycles",
                                          "phase4_help") if k in r}
    for k in ("cpu_baseline", "parity_check", "aligned_fraction"):
        if k in o:
            keep[k] = o[k]
    return keep


def main():
    args = parse_args()
    # stdout carries exactly ONE JSON line: anything libraries print to fd 1 (RCCL prints a version
    # banner there) is sent to stderr instead; the JSON goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import copy
    import torch
    from snap_amd import dist as sd
    rank, world, local_rank = sd.env_rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    if args.scaling == "strong":        # the whole job's batch split over the ranks (each rank still draws its own reads: they are i.i.d.)
        args.reads = max(2, (args.reads // max(1, world)) & ~1)
    torch.cuda.set_device(local_rank)
    env = {"rank": rank, "world": world, "local_rank": local_rank, "dev": torch.device("cuda", local_rank), "dist": None,
           "force_dist": os.environ.get("SNAP_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ}   # exercise the RCCL path on one GPU
    if args.genome_mb <= 0:             # auto: the metric's scale where it fits (every MI355X: 288 GB), the stand-in elsewhere
        free_b, _tot = torch.cuda.mem_get_info(local_rank)
        try:
            import psutil
            host_avail = psutil.virtual_memory().available
        except Exception:          # noqa: BLE001
            host_avail = 1 << 62
        # (host: each rank holds the 3.1 GB genome + its read batches; rank 0 also the FASTA text while it writes it)
        fits = free_b >= 64e9 and host_avail >= 20e9 * world + 40e9
        args.genome_mb = 3100 if fits else 256
        env["genome_choice"] = ("auto: GRCh38 scale (device has %.0f GB free)" % (free_b / 1e9) if fits
                                else "auto: the 256 Mb stand-in, because the device has %.0f GB free / the host %.0f GB available (3100 Mb needs 64 GB of HBM and 40 + 20 per rank GB of host memory)" % (free_b / 1e9, host_avail / 1e9))
        log(env["genome_choice"])

    # ---------------------------------------------------------------- the line's own leg
    bed = make_bed(args, env, paired_owner=args.workload == "paired")
    out = run_leg(args, env, bed, args.workload, primary=True)
    extra = world == 1 and not env["force_dist"] and args.workload == "single" and not args.no_extra_legs
    if extra and rank == 0:
        # ---------------------------------------------------------------- configs[2] over the same resident index (so that C3 is in the driver's line)
        try:
            pa = copy.copy(args)
            pa.steps, pa.warmup, pa.batches = max(1, args.paired_leg_steps), 1, min(args.batches, max(1, args.paired_leg_steps))
            pa.skip_refwalk = pa.skip_breakdown = pa.skip_probe = True
            pa.cpu_sample = 0
            out["paired"] = compact_leg(run_leg(pa, env, bed, "paired", primary=False))
        except BaseException as e_:          # noqa: BLE001 -- the line's own leg must survive a failing extra
            out["paired"] = {"error": "%s: %s" % (type(e_).__name__, e_)}
        # ---------------------------------------------------------------- configs[4] on one GPU (2 x 250 bp, -d 20, affine gap: AffineGapVectorized at limit 21)
        if not args.no_c5_leg:
            try:
                from snap_amd import abi as _abi
                ca = copy.copy(args)
                ca.read_len, ca.max_k, ca.insert_mean, ca.insert_sd, ca.long_indel_frac = 250, 20, 600.0, 80.0, 0.002
                ca.reads = args.c5_reads
                ca.steps, ca.warmup, ca.batches = max(1, args.c5_leg_steps), 1, min(args.batches, max(1, args.c5_leg_steps))
                ca.skip_refwalk = ca.skip_breakdown = ca.skip_probe = True
                ca.cpu_sample, ca.cpu_seconds, ca.pmc_tag = 0, min(args.cpu_seconds, 12.0), "c5"
                bed_c5 = copy.copy(bed)
                bed_c5.params = _abi.default_params(max_k=20, max_read_len=256)
                bed_c5.ref_index = getattr(bed, "ref_index", None)
                out["c5"] = compact_leg(run_leg(ca, env, bed_c5, "paired", primary=False))
                bed.ref_index = getattr(bed_c5, "ref_index", None)
            except BaseException as e_:          # noqa: BLE001
                out["c5"] = {"error": "%s: %s" % (type(e_).__name__, e_)}
    idx_dir_main, genome_main = bed.idx_dir, bed.genome
    close_bed(bed)
    if extra and rank == 0 and not args.no_e2e_leg:
        # ---------------------------------------------------------------- the product rate: FASTQ -> SAM through snap_amd/snapgpu-sam over the same index directory
        try:
            out["e2e"] = run_e2e(args, idx_dir_main, genome_main)
        except BaseException as e_:          # noqa: BLE001
            out["e2e"] = {"error": "%s: %s" % (type(e_).__name__, e_)}
    genome_main = None
    if extra and rank == 0 and args.standin_leg and args.genome_mb != args.standin_mb:
        # ---------------------------------------------------------------- the 256 Mb stand-in of rounds 1-3, same command otherwise
        try:
            sa = copy.copy(args)
            sa.genome_mb = args.standin_mb
            sa.skip_breakdown = sa.skip_probe = True
            sa.cpu_seconds = min(args.cpu_seconds, 5.0)
            env2 = dict(env); env2["genome_choice"] = "the stand-in of rounds 1-3, kept for continuity"
            bed2 = make_bed(sa, env2, paired_owner=False)
            out["genome_256mb"] = compact_leg(run_leg(sa, env2, bed2, "single", primary=False))
            close_bed(bed2)
        except BaseException as e_:          # noqa: BLE001
            out["genome_256mb"] = {"error": "%s: %s" % (type(e_).__name__, e_)}
    failures = []
    if rank == 0:
        failures = parity_verdict(out)
        out["bench_wall_s"] = time.time() - _T_PROCESS
        try:
            import resource
            out["host_peak_rss_gb"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6          # (rank 0: the rank that loads / builds the index)
        except Exception:          # noqa: BLE001
            pass
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if failures:              # the line is written (it says what differed), and the process fails: a parity regression must not look like a clean run
        log("PARITY FAILURE: " + "; ".join(failures))
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(3)


if __name__ == "__main__":
    main()
