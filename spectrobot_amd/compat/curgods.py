"""Drop-in for the reference's f2py module `curgods` (curgods.f): Curtis-Godson
column integrals over one LOS step, evaluated on the GPU."""
import numpy as np

from .._lib import lib, check, dp, ip

imxstp = 8000  # parameters.inc:64


def _call(which, n_p, nd, x, vmr=None, f=None):
    n_p = int(n_p)
    if n_p > imxstp:
        raise ValueError("n_p = %d exceeds imxstp = %d" % (n_p, imxstp))
    arrs = []
    for a in (nd, vmr, f, x):
        if a is None:
            arrs.append(None)
        else:
            arrs.append(np.ascontiguousarray(np.asarray(a, dtype=np.float64)[:n_p]))
    off = np.array([0, n_p], dtype=np.int32)
    res = np.zeros(1)
    p = [a.ctypes.data_as(dp) if a is not None else None for a in arrs]
    check(lib.sr_curgod(which, p[0], p[1], p[2], p[3], off.ctypes.data_as(ip), 1, res.ctypes.data_as(dp)),
          "curgods.curgod_fort_%d" % which)
    return float(res[0])


def curgod_fort_1(nd, x, n_p):
    """res = curgods.curgod_fort_1(nd, x, n_p)   (curgods.f:2-21)"""
    return _call(1, n_p, nd, x)


def curgod_fort_2(nd, vmr, x, n_p):
    """res = curgods.curgod_fort_2(nd, vmr, x, n_p)   (curgods.f:24-45)"""
    return _call(2, n_p, nd, x, vmr)


def curgod_fort_3(nd, vmr, f, x, n_p):
    """res = curgods.curgod_fort_3(nd, vmr, f, x, n_p)   (curgods.f:48-73)"""
    return _call(3, n_p, nd, x, vmr, f)


def curgod_fort_4(nd, vmr, f, x, n_p):
    """res = curgods.curgod_fort_4(nd, vmr, f, x, n_p)   (curgods.f:76-98)"""
    return _call(4, n_p, nd, x, vmr, f)


def curgod_batch(which, nd, x, off, vmr=None, f=None):
    """All LOS steps of a ray set in one launch: segment s covers samples
    off[s]..off[s+1]-1 of the concatenated arrays."""
    off = np.ascontiguousarray(off, dtype=np.int32)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in (nd, vmr, f, x)]
    res = np.zeros(off.size - 1)
    p = [a.ctypes.data_as(dp) if a is not None else None for a in arrs]
    check(lib.sr_curgod(which, p[0], p[1], p[2], p[3], off.ctypes.data_as(ip), off.size - 1,
                        res.ctypes.data_as(dp)), "curgods.curgod_batch")
    return res
