"""TEST INFRASTRUCTURE: build tests/emu/_build/libsnapgpu_emu.so -- snap_amd/csrc compiled with g++ against the wavefront
emulator (tests/emu/include/hip/hip_runtime.h + wave_emu.cpp).  Same translation units and -D switches as
__graft_entry__.build(); nothing here is used by the product."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "snap_amd", "csrc")
BDIR = os.environ.get("SNAPGPU_EMU_BDIR") or os.path.join(HERE, "_build")      # (SNAPGPU_EMU_BDIR: a second build beside a test run that is using the first)
LIB = os.path.join(BDIR, "libsnapgpu_emu.so")
TOOL = os.path.join(BDIR, "snapgpu-sam-emu")
CXX = os.environ.get("CXX", "g++")
# -fno-reorder-blocks: the emulator orders divergent lanes by code address (see wave_emu.cpp)
FLAGS = ["-x", "c++", "-std=c++17", "-O1", "-fno-reorder-blocks", "-fno-reorder-blocks-and-partition", "-ffp-contract=off",
         "-fPIC", "-g1", "-w", "-I", os.path.join(HERE, "include")]


def units():
    u = [("snapgpu.o", os.path.join(CSRC, "snapgpu.hip"), []), ("cigar_k.o", os.path.join(CSRC, "cigar_k.hip"), []),
         ("single_timed_k.o", os.path.join(CSRC, "single_timed_k.hip"), []),
         ("index_build.o", os.path.join(CSRC, "index_build.hip"), [])]
    u += [("paired_k%d.o" % v, os.path.join(CSRC, "paired_k.hip"), ["-DPAIRED_AGC=%d" % v]) for v in (3, 4, 6, 0)]
    u += [("single_sec_k%d.o" % v, os.path.join(CSRC, "single_sec_k.hip"), ["-DSINGLE_AGC=%d" % v]) for v in (3, 4, 6, 0)]
    u += [("single_planes_k%d.o" % v, os.path.join(CSRC, "single_planes_k.hip"), ["-DSINGLE_AGC=%d" % v]) for v in (3, 4, 6, 0)]
    u += [("paired_sec_k%d.o" % v, os.path.join(CSRC, "paired_k.hip"), ["-DPAIRED_AGC=%d" % v, "-DPAIRED_SEC"]) for v in (3, 0)]
    # -fsanitize=thread only for its instrumentation: wave_emu.cpp supplies the __tsan_* hooks (stores become rendezvous points)
    u = [(o, src, fl + ["-fsanitize=thread", "--param", "tsan-instrument-func-entry-exit=0"]) for (o, src, fl) in u]
    u += [("wave_emu.o", os.path.join(HERE, "wave_emu.cpp"), [])]
    return u


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout.decode(errors="replace")[-6000:]))


def build(verbose=False):
    os.makedirs(BDIR, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "snapgpu.h"),
            os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    newest = max(os.path.getmtime(d) for d in deps)
    us = units()

    def is_stale(u):
        o = os.path.join(BDIR, u[0])
        if not os.path.exists(o):
            return True
        t = os.path.getmtime(o)
        return t < (max(newest, os.path.getmtime(u[1])) if u[1].startswith(CSRC) else
                    max(os.path.getmtime(u[1]), os.path.getmtime(os.path.join(HERE, "include", "hip", "hip_runtime.h"))))
    stale = [u for u in us if is_stale(u)]
    if stale:
        if verbose:
            print("emu build: compiling", [u[0] for u in stale], file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(len(stale), os.cpu_count() or 4)) as ex:
            list(ex.map(lambda u: _run([CXX] + FLAGS + u[2] + ["-c", u[1], "-o", os.path.join(BDIR, u[0])]), stale))
    if stale or not os.path.exists(LIB):
        _run([CXX, "-shared", "-fPIC", "-o", LIB] + [os.path.join(BDIR, u[0]) for u in us] + ["-lpthread"])
    # the native FASTQ -> SAM host program, linked against the emulated library (same source as snap_amd/snapgpu-sam)
    tool_src = os.path.join(CSRC, "host", "snapgpu_sam.cpp")
    if not os.path.exists(TOOL) or os.path.getmtime(TOOL) < max(os.path.getmtime(tool_src), os.path.getmtime(LIB)):
        _run([CXX, "-O2", "-std=c++17", "-o", TOOL, tool_src, "-L" + BDIR, "-lsnapgpu_emu", "-Wl,-rpath," + BDIR, "-lpthread", "-lz", "-ldl"])
    itool_src = os.path.join(CSRC, "host", "snapgpu_index.cpp")
    itool = os.path.join(BDIR, "snapgpu-index-emu")
    if not os.path.exists(itool) or os.path.getmtime(itool) < max(os.path.getmtime(itool_src), os.path.getmtime(LIB)):
        _run([CXX, "-O2", "-std=c++17", "-o", itool, itool_src, "-L" + BDIR, "-lsnapgpu_emu", "-Wl,-rpath," + BDIR, "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
