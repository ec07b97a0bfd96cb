"""The reference's own full-path fixture through the product: tests/datatest/datatest.fa + datatest.fq of the
reference tree (committed under tests/golden/datatest/) indexed by the reference's indexer, aligned by `snap_amd/snapgpu-sam` and by the
reference CLI with the shim (oracle/_ref/snap-aligner-gpu), and compared with the SAM the reference's authors committed as the expected
output, correct-fq-datatest.sam: QNAME, FLAG, RNAME, POS, MAPQ, CIGAR, SEQ, QUAL and NM of both reads."""
import os
import subprocess

import pytest

from oracle import ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = os.path.join(ROOT, "tests", "golden", "datatest")
TOOL = os.environ.get("SNAPGPU_TEST_TOOL") or os.path.join(ROOT, "snap_amd", "snapgpu-sam")
SHIM = os.environ.get("SNAPGPU_TEST_SHIM") or os.path.join(ROOT, "oracle", "_ref", "snap-aligner-gpu")


def fields(path):
    out = []
    for line in open(path):
        if line.startswith("@"):
            continue
        t = line.rstrip("\n").split("\t")
        nm = [x for x in t[11:] if x.startswith("NM:i:")]
        out.append((t[0], int(t[1]), t[2], int(t[3]), int(t[4]), t[5], t[9], t[10], nm[0] if nm else None))
    return out


@pytest.fixture(scope="module")
def datatest_index(tmp_path_factory):
    if not ref.available() or not os.path.exists(ref.CLI_PATH):
        pytest.skip("oracle/_ref not on this box")
    d = str(tmp_path_factory.mktemp("datatest"))
    ref.build_index(os.path.join(DT, "datatest.fa"), d + "/idx", 16, threads=1)      # the README's own command: index ... -s 16
    return d


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL, timeout=600)
    assert r.returncode == 0, "%s failed:\n%s" % (cmd[0], r.stdout.decode(errors="replace")[-2000:])


def test_datatest_through_snapgpu_sam(datatest_index):
    d = datatest_index
    exp = fields(os.path.join(DT, "correct-fq-datatest.sam"))
    assert [(e[3], e[4], e[5], e[8]) for e in exp] == [(1, 70, "101=", "NM:i:0"), (102, 70, "101=", "NM:i:0")]
    _run([TOOL, "single", d + "/idx", os.path.join(DT, "datatest.fq"), "-o", d + "/new.sam", "-="])
    assert fields(d + "/new.sam") == exp
    # ... and the reference CLI of this tree writes the same records (header and aux fields apart: the committed file is from an older SNAP)
    _run([ref.CLI_PATH, "single", d + "/idx", os.path.join(DT, "datatest.fq"), "-o", d + "/ref.sam", "-=", "-t", "1"])
    assert fields(d + "/ref.sam") == exp


def test_datatest_through_the_shim(datatest_index):
    if not os.path.exists(SHIM):
        pytest.skip("oracle/_ref/snap-aligner-gpu not built")
    d = datatest_index
    _run([SHIM, "single", d + "/idx", os.path.join(DT, "datatest.fq"), "-o", d + "/shim.sam", "-=", "-t", "1"])
    assert fields(d + "/shim.sam") == fields(os.path.join(DT, "correct-fq-datatest.sam"))
