"""spectrobot_amd -- MI355X-native engine for SpectRobot's spectral hot path.

Importing the package loads libspectrobot_hip.so (spectrobot_amd/lib/); there is
no CPU fallback and the import fails loudly when the library is not built.
"""
import os as _os

# The coefficient op runs on six HIP streams (the caller's, table preparation, two for the far-field chain, the zones
# kernel, a copy stream).  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default: two of the six then
# share a queue and one's kernels wait behind the other's although no event orders them -- the next call's table
# preparation sat behind the 3.6 ms zones kernel of the current one.  6 or more: 5.47 instead of 5.63 ms per BASELINE
# step (profiles/r05_hw_queues_ab.txt).  The runtime reads the variable when it initialises, i.e. at the first HIP
# call of the process: set here unless the caller already did (it has no effect if HIP is up already).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: F401,E402
from . import engine, synthetic, distributed, compat  # noqa: F401,E402
from . import spect_base_module, spect_classes, spect_main_module  # noqa: F401,E402
