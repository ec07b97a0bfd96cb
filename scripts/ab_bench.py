#!/usr/bin/env python
"""A/B runs: bench.py over another build of the C ABI library.

    python scripts/ab_bench.py build <name> [hipcc flags ...]     # here (no GPU needed): snap_amd/ab/libsnapgpu_<name>.so, objects in snap_amd/ab/build_<name>/
    python scripts/ab_bench.py run <name> [bench.py arguments]    # on the GPU box: bench.py with snap_amd.aligner.LIB_PATH pointing at that build

Measurement tooling only: the product opens snap_amd/libsnapgpu.so and nothing else.  The variants are git-ignored (*.so) and travel to
the GPU box with the snapshot, like the product library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lib_of(name):
    return os.path.join(ROOT, "snap_amd", "ab", "libsnapgpu_%s.so" % name)


def main():
    mode, name = sys.argv[1], sys.argv[2]
    if mode == "build":
        import __graft_entry__ as g
        print(g.build_library(out=lib_of(name), bdir=os.path.join(ROOT, "snap_amd", "ab", "build_" + name), extra_flags=sys.argv[3:]))
        return
    import snap_amd.aligner as al
    al.LIB_PATH = lib_of(name)
    al._lib = None
    if not os.path.exists(al.LIB_PATH):
        raise SystemExit("no such variant: " + al.LIB_PATH)
    sys.argv = ["bench.py"] + sys.argv[3:]
    import bench
    bench.main()


if __name__ == "__main__":
    main()
