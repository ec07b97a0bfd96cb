"""Host-side mirror of the reference's operator interface over include/snapgpu.h."""
import os as _os

# Eight hardware queues for the feeders' streams instead of the HIP runtime's four (snap_amd/csrc/snapgpu.hip: snapgpu_hw_queues): the runtime
# reads this once, when it starts -- importing this package before the first GPU call is early enough; a value already in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
