"""bench.py itself on the MI355X, at a size that takes seconds: the line the driver reads (its extra legs included) and the N > 1 code
path -- RCCL initialisation, the index broadcast HBM -> HBM, device-pointer adoption -- forced onto one rank, so that the driver's
`pytest -m gpu` exercises both every round."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--genome-mb", "24", "--reads", "20000", "--steps", "3", "--warmup", "1", "--batches", "2"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    e["SNAP_BENCH_DIR"] = e.get("SNAP_BENCH_DIR", "/tmp/snap_bench_test")
    r = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line on stdout, got %d" % len(lines)
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_with_its_extra_legs_small():
    """The default-shaped run (single-end leg + the paired-end and c5 legs over the same resident index + the FASTQ -> SAM leg through
    snap_amd/snapgpu-sam, plus the opt-in stand-in leg), every sampled read and pair compared with the compiled reference inside bench.py
    itself, the e2e leg's records with the reference CLI's."""
    o = _run([sys.executable, "bench.py"] + SMALL + ["--paired-leg-steps", "2", "--standin-leg", "--standin-mb", "12", "--cpu-seconds", "1",
                                                     "--c5-reads", "6000", "--c5-leg-steps", "2", "--e2e-reads", "300000", "--e2e-ref-reads", "100000"], timeout=900)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity_check"):
        assert k in o, k
    assert o["n_gpus"] == 1 and o["steps"] == 3 and o["value"] > 0 and o["unit"] == "reads/s"
    assert o["config"]["genome_mb"] == 24 and o["config"]["kernel_source_hash"]
    rf = o["roofline"]
    for k in ("bound", "nominal_bound", "achieved", "peak", "unit", "frac", "traffic", "probe", "probe_frac", "blocking_call_ms_min", "blocking_call_ms_median", "blocking_call_ms_max",
              "launch_event_ms_min", "launch_event_ms_median", "launch_event_ms_max", "mean_wave_residency"):
        assert k in rf, k
    assert rf["blocking_call_ms_min"] <= rf["blocking_call_ms_median"] <= rf["blocking_call_ms_max"]
    assert rf["launch_event_ms_min"] <= rf["launch_event_ms_median"] <= rf["launch_event_ms_max"]
    # the parity figures and the legs' values as scalars of `config` (what the driver's record keeps)
    cf = o["config"]
    assert cf["parity_failures"] == 0 and cf["parity_units"] >= 20000 and cf["parity_mismatching"] == 0
    assert cf["paired_value"] == o["paired"]["value"] and cf["paired_parity_mismatching"] == 0 and cf["paired_parity_units"] >= 10000
    assert cf["c5_value"] == o["c5"]["value"] and cf["c5_parity_mismatching"] == 0
    assert cf["e2e_value"] == o["e2e"]["value"] and cf["e2e_identical_records"] is True
    assert rf["probe"]["numerator_basis"].startswith("reference slot walk")
    assert rf["probe"]["algorithmic_bytes_per_launch"] <= rf["probe"]["bucket_line_bytes_per_launch"] * 1.5
    pc = o["parity_check"]
    assert pc["reads"] >= 20000 and pc["mismatching_fields"] == [] and pc["excluded"] == 0
    assert o["cpu_baseline"]["kind"] == "reference" and o["cpu_baseline"]["value"] > 0
    p = o["paired"]
    assert "error" not in p, p
    assert p["value"] > 0 and p["parity_check"]["mismatching_pairs"] == 0 and p["parity_check"]["pairs"] >= 10000 and p["cpu_baseline"]["value"] > 0
    assert p["config"]["index_bytes_hbm"] == o["config"]["index_bytes_hbm"]           # the same resident index
    c5 = o["c5"]
    assert "error" not in c5, c5
    assert c5["value"] > 0 and c5["parity_check"]["mismatching_pairs"] == 0 and c5["parity_check"]["pairs"] >= 3000 and c5["cpu_baseline"]["value"] > 0
    assert "2 x 250 bp" in c5["config"]["workload"] and "-d 20" in c5["config"]["workload"]
    e = o["e2e"]
    assert "error" not in e, e
    assert e["value"] > 0 and e["reads"] == 300000 and e["identical_records"] is True and e["records_compared"] == 100000
    assert e["reference_cli"]["reads_per_s_own_figure"] > 0
    g = o["genome_256mb"]
    assert "error" not in g, g
    assert g["config"]["genome_mb"] == 12 and g["value"] > 0 and g["parity_check"]["mismatching_fields"] == []


@pytest.mark.gpu
def test_bench_fails_loudly_on_a_kernel_that_answers_differently():
    """snap_amd/ab/libsnapgpu_broken.so (__graft_entry__.build_broken_variant: the product's objects with ONE translation unit rebuilt
    -DSNAPGPU_TEST_BREAK_PARITY, which flips the MAPQ's low bit of every 997th read in the single-end kernel the bench times) under
    bench.py: the line is still written and says what differed, and the process exits non-zero."""
    lib = os.path.join(ROOT, "snap_amd", "ab", "libsnapgpu_broken.so")
    if not os.path.exists(lib):
        pytest.skip("the broken variant was not built (python -c 'import __graft_entry__ as g; g.build_broken_variant()')")
    e = dict(os.environ)
    e["SNAP_BENCH_DIR"] = e.get("SNAP_BENCH_DIR", "/tmp/snap_bench_test")
    r = subprocess.run([sys.executable, "scripts/ab_bench.py", "run", "broken"] + SMALL + ["--no-extra-legs", "--cpu-seconds", "1", "--skip-probe", "--skip-refwalk",
                                                                                          "--skip-breakdown"], cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr.decode(errors="replace")[-2000:])
    assert b"PARITY FAILURE" in r.stderr
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    o = json.loads(lines[0])
    assert o["config"]["parity_failures"] == 1 and o["config"]["parity_mismatching"] >= 1
    assert any("mapq" in str(f) for f in o["parity_check"]["mismatching_fields"])


@pytest.mark.gpu
def test_bench_multi_gpu_path_on_one_rank():
    """SNAP_BENCH_FORCE_DIST=1 under torch.distributed.run with one process: process group over RCCL, rank 0 loads the directory, the index
    blobs go through dist.broadcast into tensors the context then adopts by device pointer (snap_amd/dist.py), timing by barrier + max."""
    port = _free_port()
    o = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
              "bench.py", "--gpus", "1"] + SMALL + ["--skip-probe", "--skip-refwalk", "--skip-breakdown", "--cpu-seconds", "1"],
             env={"SNAP_BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert o["n_gpus"] == 1 and o["value"] > 0
    assert "paired" not in o and "genome_256mb" not in o and "e2e" not in o and "c5" not in o                               # the extra legs belong to the plain one-GPU run
    assert o["parity_check"]["mismatching_fields"] == [] and o["parity_check"]["reads"] >= 20000
