#!/usr/bin/env python
"""VGPRs / SGPRs / scratch / LDS / occupancy of the kernels in a build directory's gfx950 objects (the metadata note of each code object).

    python scripts/kernel_resources.py [build-dir] [name-substring ...]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def notes(obj):
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "co")
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + obj, "--output=" + co], capture_output=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            # a host object with an embedded fat binary: pull the section out first
            fb = os.path.join(td, "fb")
            subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, obj], capture_output=True)
            if not os.path.exists(fb):
                return ""
            subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + fb, "--output=" + co], capture_output=True)
        if not os.path.exists(co):
            return ""
        return subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout


def main():
    bdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snap_amd", "build")
    pats = sys.argv[2:]
    for obj in sorted(glob.glob(os.path.join(bdir, "*.o"))):
        txt = notes(obj)
        for blk in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
            name = g("name")
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem)
            if pats and not any(p in dem for p in pats):
                continue
            print("%-22s %-70s vgpr %3s sgpr %3s spill(v/s) %s/%s scratch %5s B/lane lds %6s" % (
                os.path.basename(obj), dem[:70], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                g("private_segment_fixed_size"), g("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
