"""Drop-in for the reference's f2py module `lineshape` (lineshape.f) on the GPU.

Same call shapes as f2py generates from the Cf2py directives
(lineshape.f:4-5, 227-228); every call runs a HIP kernel through the C ABI.
"""
import numpy as np

from .._lib import lib, check, dp, ip, IMXSIG

imxsig = IMXSIG          # parameters.inc:65
imxlines = 40000         # parameters.inc:64
imxsig_long = 2000000    # parameters.inc:64


def humliv_bb(x, i1, i2, x0, lw, dw):
    """y = lineshape.humliv_bb(x, i1, i2, x0, lw, dw)   (lineshape.f:226-569)
    x: 13010 float64 (the f2py wrapper fixes the length), i1/i2 1-based."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim != 1 or x.size != IMXSIG:
        raise ValueError("0-th dimension must be fixed to %d but got %d" % (IMXSIG, x.size))
    y = np.zeros(IMXSIG)
    check(lib.sr_humliv_bb(x.ctypes.data_as(dp), IMXSIG, int(i1), int(i2), float(x0), float(lw), float(dw),
                           y.ctypes.data_as(dp)), "lineshape.humliv_bb")
    return y


def sum_all_lines(spe_ini, matrix, init, fin, n_lines, n_spe):
    """spe_fin = lineshape.sum_all_lines(spe_ini, matrix, init, fin, n_lines, n_spe)
    (lineshape.f:2-25).  matrix[ilin, i]; init/fin 1-based inclusive.  The
    fixed Fortran extents (2e6 / 40000 x 13010) are upper limits here, not
    required sizes."""
    spe = np.array(spe_ini, dtype=np.float64)  # copy: intent(out) spe_fin
    matrix = np.asarray(matrix, dtype=np.float64)
    n_lines = int(n_lines)
    if n_lines > imxlines:
        raise ValueError("%d are too many lines (imxlines = %d)" % (n_lines, imxlines))
    if spe.size > imxsig_long:
        raise ValueError("spectrum longer than imxsig_long = %d" % imxsig_long)
    rows = np.ascontiguousarray(matrix[:n_lines])
    a = np.ascontiguousarray(np.asarray(init)[:n_lines], dtype=np.int32)
    b = np.ascontiguousarray(np.asarray(fin)[:n_lines], dtype=np.int32)
    check(lib.sr_sum_all_lines(spe.ctypes.data_as(dp), spe.size, rows.ctypes.data_as(dp), a.ctypes.data_as(ip),
                               b.ctypes.data_as(ip), n_lines, rows.shape[1] if rows.ndim == 2 else 1),
          "lineshape.sum_all_lines")
    return spe
