"""TEST INFRASTRUCTURE: the single-end fixtures through an emulator build whose candidate-table directory holds THREE elements (-DSNAPGPU_DIR_CAP=3; align_single.h:
REGDIR), so that nearly every read crosses the directory-to-hash transition.  ~8 min on 8 cores; python scripts/emu_dircap_check.py"""
import os, sys
sys.path.insert(0, "/root/repo")
os.environ["SNAPGPU_EMU_BDIR"] = "/tmp/snapgpu_emu_dircap"
import tests.emu.build as eb
eb.FLAGS.append("-DSNAPGPU_DIR_CAP=3")
lib = eb.build(verbose=True)
os.environ["SNAPGPU_TEST_LIB"] = lib
import subprocess
r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_secondary.py", "-q", "-x", "-m", "gpu", "-k", "align_read_vs_reference_fixture or ragged or secondary", "--timeout", "3000"], cwd="/root/repo", env=dict(os.environ, SNAPGPU_SINGLE_HELP="0"))
sys.exit(r.returncode)
