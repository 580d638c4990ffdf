"""Drop-in for the reference's f2py module `fparts_mod` (fparts_mod.f)."""
import ctypes as C

import numpy as np

from .._lib import lib, check, dp


def bd_tips_2003(mol, iso):
    """gi, t_grid, QT_grid = fparts_mod.bd_tips_2003(MOL, ISO)   (fparts_mod.f:33-295)"""
    gi = C.c_double(0.0)
    t = np.zeros(119)
    q = np.zeros(119)
    check(lib.sr_bd_tips_2003(int(mol), int(iso), C.byref(gi), t.ctypes.data_as(dp), q.ctypes.data_as(dp)),
          "fparts_mod.bd_tips_2003")
    return gi.value, t, q
