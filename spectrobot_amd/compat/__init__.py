"""f2py-shaped drop-ins for the reference's three extension modules.

`install()` registers them in sys.modules under the names the reference imports
(`import lineshape`, `import fparts_mod`, `import curgods`; spect_classes.py:18,
1685), which is all an unmodified caller needs.
"""
import sys

from . import curgods, fparts_mod, lineshape  # noqa: F401


def install():
    sys.modules["lineshape"] = lineshape
    sys.modules["fparts_mod"] = fparts_mod
    sys.modules["curgods"] = curgods
