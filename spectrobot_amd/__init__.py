"""spectrobot_amd -- MI355X-native engine for SpectRobot's spectral hot path.

Importing the package loads libspectrobot_hip.so (spectrobot_amd/lib/); there is
no CPU fallback and the import fails loudly when the library is not built.
"""
from . import _lib  # noqa: F401
from . import engine, synthetic, distributed, compat  # noqa: F401
from . import spect_base_module, spect_classes, spect_main_module  # noqa: F401
