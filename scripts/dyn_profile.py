#!/usr/bin/env python
"""TEST INFRASTRUCTURE / analysis: an estimated DYNAMIC instruction profile of a kernel by source line, without a GPU.

PC sampling and thread trace are not available on this pool, so the two halves come from different places:
  * how often each source line RUNS: the wavefront emulator's build of the device code compiled `--coverage` (gcov counts per line and per
    template instance; a wave-uniform line is executed by all 64 lane fibers, so count / 64 = executions per wavefront);
  * what each source line COSTS: hipcc's gfx950 listing of the same translation unit with `-gline-tables-only` (`.loc` before every
    instruction): scalar / vector / lane-move / LDS / memory / control instructions per line, in the kernel that ran and in the functions
    it calls.
dynamic(line) = static(line) / copies x runs(line) / 64.  `copies` is 1 unless --copies says otherwise for a line range (a function the
compiler inlined at several sites); compile the listing with -DSNAPGPU_SCORE_NOINLINE to take the biggest duplicate (score(), three sites)
out.  Lines guarded by `lane == 0` are undercounted (they run in one fiber and cost a wave instruction) -- those are vector / memory
instructions, not the scalar ones this tool is for.

    python scripts/dyn_profile.py single [n_reads] [--top N]
    python scripts/dyn_profile.py paired [n_pairs] [--read-len 150|250] [--max-k 8|20]
"""
import collections
import gzip
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COV = "/tmp/snapgpu_emu_cov"
os.environ["SNAPGPU_EMU_BDIR"] = COV
os.environ.setdefault("SNAPGPU_EMU_CUS", "8")
HIPCC = "/opt/rocm/bin/hipcc"
CSRC = os.path.join(ROOT, "snap_amd", "csrc")


def klass(op):
    if op.startswith(("v_writelane", "v_readlane", "v_readfirstlane")): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_setprio", "s_barrier")): return "wait"
    if op.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc", "s_getpc", "s_call")): return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    return "other"


LLVM = "/opt/rocm/lib/llvm/bin"
HELPER_FILES = ("dev_common.h", "amd_device_functions.h", "amd_warp_functions.h", "amd_hip_runtime.h", "amd_hip_atomic.h", "amd_hip_unsafe_atomics.h", "hip_ldg.h",
                "amd_math_functions.h", "math_fwd.h", "__clang_hip_math.h", "__clang_hip_libdevice_declares.h", "type_traits", "ockl_image.h")


def static_by_chain(co, want):
    """[(key (file, line), copy id, class, inclusive function names)] for every instruction of the functions of code object `co` whose mangled
    name contains one of `want`.  The inline chain of each instruction comes from llvm-symbolizer; an instruction of a small helper (first_u32,
    lane_id, the HIP headers' intrinsics wrappers) is attributed to the line that called it."""
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    ins, inside, per_fn = [], False, collections.Counter()
    for l in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            inside = any(w in m.group(1) for w in want); fn = m.group(1); continue
        if not inside:
            continue
        m = re.match(r"^\s+(\S+).*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1))); per_fn[fn] += 1
    sym = subprocess.run([LLVM + "/llvm-symbolizer", "--obj=" + co, "--inlines", "--functions=short", "--basenames", "--verbose"],
                         input="\n".join("0x%x" % a for a, _ in ins), capture_output=True, text=True).stdout
    blocks = [b.strip("\n").splitlines() for b in sym.split("\n\n") if b.strip()]
    assert len(blocks) == len(ins), (len(blocks), len(ins))
    out = []
    for (addr, op), blk in zip(ins, blocks):
        frames = []
        for l in blk:
            if not l.startswith(" "):
                frames.append([l, "?", 0, 0, 0])                     # function, file, line, column, discriminator
            else:
                kv = l.strip().split(": ", 1)
                if kv[0] == "Filename": frames[-1][1] = kv[1]
                elif kv[0] == "Line": frames[-1][2] = int(kv[1])
                elif kv[0] == "Column": frames[-1][3] = int(kv[1])
                elif kv[0] == "Discriminator": frames[-1][4] = int(kv[1])
        frames = [tuple(f) for f in frames]
        while len(frames) > 1 and (frames[0][1] in HELPER_FILES or frames[0][2] == 0):
            frames = frames[1:]
        key = (frames[0][1], frames[0][2])
        copy = (frames[0][4],) + tuple(frames[1:])       # (the line's own discriminator: copies an unrolled loop made)
        out.append((key, copy, klass(op), tuple(f[0] for f in frames)))
    return out, per_fn


_dem = {}


def gcov_counts(gcda_stem, want_fn=None):
    """{(file, line): executions summed over the function instances whose demangled name contains want_fn}"""
    out = subprocess.run(["gcov", "--json-format", "--stdout", "-m", gcda_stem + ".gcda"], cwd=COV, capture_output=True)
    cnt = collections.Counter(); fn_of = {}
    for doc in out.stdout.decode(errors="replace").splitlines():
        if not doc.startswith("{"):
            continue
        j = json.loads(doc)
        for f in j["files"]:
            base = os.path.basename(f["file"])
            for ln in f["lines"]:
                name = ln.get("function_name", "")
                if name not in _dem:
                    _dem[name] = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() if name.startswith("_Z") else name
                name = _dem[name]
                if want_fn and not any(w in name for w in want_fn):
                    continue
                cnt[(base, ln["line_number"])] += ln["count"]
                fn_of[(base, ln["line_number"])] = name
    return cnt, fn_of


_srcs = {}


def _src(base):
    if base not in _srcs:
        _srcs[base] = open(os.path.join(CSRC, base), errors="replace").read().splitlines()
    return _srcs[base]


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<[^<>]*(<[^<>]*>[^<>]*)*>", "<>", name)
    return name[-60:]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "single"
    args = [a for a in sys.argv[2:] if not a.startswith("--")]
    opt = {sys.argv[i][2:]: sys.argv[i + 1] for i in range(2, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
    n = int(args[0]) if args else 1500
    top = int(opt.get("top", 60))
    read_len = int(opt.get("read-len", 150)); max_k = int(opt.get("max-k", 8))

    unit, asm_flags, src = "single_sec_k3", ["-DSINGLE_AGC=3"], "single_sec_k.hip"
    kern = ["k_align_singleILi3ELb0ELb1ELb0ELb0EE", "lv_compute_fn", "ag_dispatch_fnILi3ELb1E"]
    gfn = None
    if mode.endswith("paired"):          # the paired-end kernel of the read length's class (2 x 150: AGC 3; 2 x 250: AGC 4), fast form
        agc = 3 if read_len <= 170 else 6
        unit, asm_flags, src = "paired_k%d" % agc, ["-DPAIRED_AGC=%d" % agc], "paired_k.hip"
        kern = ["k_align_pairedILi%dELb0ELb0E" % agc, "lv_compute_fn", "ag_dispatch_fnILi%dELb0E" % agc]
    if mode.startswith("run-"):          # the child: run the workload on the coverage build; the counters are written when the process exits
        import tests.emu.build as eb
        os.makedirs(COV, exist_ok=True)
        for f in os.listdir(COV):
            if f.endswith(".gcda"):
                os.unlink(os.path.join(COV, f))
        eb.FLAGS += ["--coverage", "-fprofile-update=atomic"]
        run0 = eb._run
        eb._run = lambda cmd: run0(cmd + (["--coverage"] if "-shared" in cmd else []))
        lib_path = eb.build(verbose=True)
        import numpy as np
        import snap_amd.aligner as al
        al._lib, al.LIB_PATH = None, lib_path
        from snap_amd import abi, synth
        from tests import util
        if "genome-mb" in opt:             # a genome drawn the way bench.py draws its own (30 % planted repeats), indexed by the reference's indexer
            from oracle import ref
            from snap_amd.index import GenomeIndex
            mb = int(opt["genome-mb"])
            d = "/tmp/snapgpu_dyn_genome_%d" % mb
            contigs = synth.make_genome(20260925, mb * 1_000_000, n_contigs=max(1, min(24, mb // 8)), repeat_frac=0.30, max_copies=5000, repeat_len=(200, 3000), max_divergence=0.05)
            if not os.path.exists(os.path.join(d, "ix", "GenomeIndexHash")):
                os.makedirs(d, exist_ok=True)
                synth.write_fasta(os.path.join(d, "g.fa"), contigs)
                ref.build_index(os.path.join(d, "g.fa"), os.path.join(d, "ix"), seed_len=20, threads=8)
            ix = GenomeIndex.load_from_directory(os.path.join(d, "ix"))
        else:
            ix = util.load_golden_index()
            pad = (ix.genome_padded.size - ix.n_bases) // 2
            ends = [c.begin for c in ix.contigs[1:]] + [ix.n_bases]
            contigs = [(c.name, ix.genome_padded[pad + c.begin: pad + e - ix.chromosome_padding]) for c, e in zip(ix.contigs, ends)]
        if mode == "run-single":
            os.environ["SNAPGPU_SINGLE_HELP"] = "0"          # the bench's path (several feeders): one pass of the exact form, no help protocol
            from snap_amd.aligner import BaseAligner
            rd = synth.make_reads(20260925, contigs, n, read_len)
            a = BaseAligner(ix, abi.default_params(max_k=max_k, max_read_len=((read_len + 31) // 32) * 32))
            offs = np.arange(n + 1, dtype=np.uint64) * read_len
            a.AlignRead(rd["bases"], rd["quals"], offs)
            print("ran %d reads; counters: %s" % (n, a.counters()))
            a.close()
        else:
            from snap_amd.aligner import ChimericPairedEndAligner
            if read_len > 170:
                pr = synth.make_pairs(20260925, contigs, n, read_len, insert_mean=600, insert_sd=80, long_indel_frac=0.002)
            else:
                pr = synth.make_pairs(20260925, contigs, n, read_len)
            a = ChimericPairedEndAligner(ix, abi.default_params(max_k=max_k, max_read_len=((read_len + 31) // 32) * 32), abi.default_paired_params())
            a.align(pr["bases"].reshape(-1), pr["quals"].reshape(-1), pr["offsets"])
            print("ran %d pairs; counters: %s" % (n, a.counters()))
            a.close()
        return
    subprocess.run([sys.executable, os.path.abspath(__file__), "run-" + mode] + sys.argv[2:], check=True)
    cnt, fn_of = gcov_counts(os.path.join(COV, unit), gfn)

    obj, co = os.path.join(COV, unit + ".dev.o"), os.path.join(COV, unit + ".co")
    extra = os.environ.get("DYN_PROFILE_FLAGS", "").split()
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fPIC", "-O3", "-gline-tables-only", "-fdebug-info-for-profiling", "--cuda-device-only", "-c"]
                   + asm_flags + extra + [os.path.join(CSRC, src), "-o", obj], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + obj, "--output=" + co], check=True)
    recs, per_fn = static_by_chain(co, kern)
    print("static instructions:", dict(per_fn))
    copies = collections.defaultdict(set)
    for key, copy, k, fns in recs:
        copies[key].add(copy)
    tot = collections.Counter(); by_line = collections.defaultdict(collections.Counter); incl = collections.defaultdict(collections.Counter); self_ = collections.defaultdict(collections.Counter)
    for key, copy, k, fns in recs:
        w = cnt.get(key, 0) / 64.0 / n / len(copies[key])
        if not w:
            continue
        tot[k] += w; by_line[key][k] += w
        self_[fns[0]][k] += w
        for f in set(fns):
            incl[f][k] += w
    sc = lambda d: d["salu"] + d["smem"]
    print("\nestimated dynamic instructions per read: " + "  ".join("%s %.0f" % kv for kv in sorted(tot.items(), key=lambda x: -x[1])) + "   all %.0f" % sum(tot.values()))
    print("\nby function, INCLUSIVE of what it inlines (a callee reached by a real call -- Landau-Vishkin, affine gap -- is its own root): scalar | branch | wait | vector | lane moves | lds | memory")
    for f, d in sorted(incl.items(), key=lambda x: -sc(x[1]))[:45]:
        print("  %-46s %7.0f %6.0f %6.0f %7.0f %6.0f %6.0f %6.0f     self: scalar %6.0f vector %6.0f" % (f[:46], sc(d), d["branch"], d["wait"], d["valu"], d["lane"], d["lds"], d["vmem"] + d["scratch"],
                                                                                                        sc(self_[f]), self_[f]["valu"] + self_[f]["lane"]))
    print("\ntop %d lines by scalar instructions: scalar | vector+lane | runs per read | copies" % top)
    for key, d in sorted(by_line.items(), key=lambda x: -sc(x[1]))[:top]:
        try:
            text = _src(key[0])[key[1] - 1].strip()[:110]
        except Exception:
            text = ""
        print("  %-16s %5d  %7.0f %7.0f   runs %8.2f  copies %3d   | %s" % (key[0], key[1], sc(d), d["valu"] + d["lane"], cnt.get(key, 0) / 64.0 / n, len(copies[key]), text))


if __name__ == "__main__":
    main()
