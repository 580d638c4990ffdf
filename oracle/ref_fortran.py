"""ctypes access to oracle/_ref/*.so -- the reference's own Fortran, compiled from
/root/reference by `make -C oracle ref` (amdflang).  TEST INFRASTRUCTURE ONLY.

Used (a) here, in the build container, by tests/golden/make_golden.py to
generate the committed fixtures and by tests that cross-check the C
restatement against the compiled reference when oracle/_ref exists, and
(b) on the GPU box as bench.py's cpu_baseline kind "reference" (the .so
travels, the sources do not).  Flang symbols: lower-case + trailing underscore,
every argument by reference.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_libs = {}
IMXSIG = 13010   # parameters.inc:65
IMXSTP = 8000    # parameters.inc:64


def available():
    return all(os.path.exists(os.path.join(_REF, "lib%s.so" % n))
               for n in ("lineshape", "curgods", "fparts_mod"))


def _lib(name):
    if name not in _libs:
        _libs[name] = C.CDLL(os.path.join(_REF, "lib%s.so" % name))
    return _libs[name]


def _r(v, t):
    return C.byref(t(v))


def humliv_bb(x, i1, i2, x0, lw, dw):
    """lineshape.f:226; x must hold IMXSIG doubles (f2py contract)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert x.size == IMXSIG
    y = np.zeros(IMXSIG)
    _lib("lineshape").humliv_bb_(x.ctypes.data_as(_dp), _r(i1, C.c_int), _r(i2, C.c_int),
                                 _r(x0, C.c_double), _r(lw, C.c_double), _r(dw, C.c_double),
                                 y.ctypes.data_as(_dp))
    return y


def humli_bb(rx, ry):
    out = C.c_double(0.0)
    _lib("lineshape").humli_bb_(_r(rx, C.c_double), _r(ry, C.c_double), C.byref(out))
    return out.value


def bd_tips_2003(mol, iso):
    """fparts_mod.f:33 -> gi, t_grid[119], QT_grid[119]"""
    gi = C.c_double(0.0)
    t = np.zeros(119)
    q = np.zeros(119)
    _lib("fparts_mod").bd_tips_2003_(_r(mol, C.c_int), _r(iso, C.c_int), C.byref(gi),
                                     t.ctypes.data_as(_dp), q.ctypes.data_as(_dp))
    return gi.value, t, q


def _pad(a):
    b = np.zeros(IMXSTP)
    b[:len(a)] = a
    return b


def curgod(which, nd, x, vmr=None, f=None):
    """curgods.f:2-98"""
    n_p = len(nd)
    res = C.c_double(0.0)
    args = [_pad(nd)]
    if which >= 2:
        args.append(_pad(vmr))
    if which >= 3:
        args.append(_pad(f))
    args.append(_pad(x))
    ptrs = [a.ctypes.data_as(_dp) for a in args]
    getattr(_lib("curgods"), "curgod_fort_%d_" % which)(*ptrs, _r(n_p, C.c_int), C.byref(res))
    return res.value


def time_humliv_bb(n_calls=2000, n_warm=200, x0_off=1e-4, lw=1e-4, dw=4e-3, step=5e-4, centre=3000.0):
    """Per-call wall time (microseconds) of the reference's compiled humliv_bb on one 13010-point window
    (i1=1, i2=13010: what MakeShape passes, spect_classes.py:1999), arguments prepared once so that only
    the Fortran call itself (through ctypes, by reference) is inside the timed region.  Returns
    dict(median_us, min_us, n_calls, params)."""
    import time
    lin = np.arange(-IMXSIG * step / 2, IMXSIG * step / 2, step)
    x = np.ascontiguousarray(lin + centre)
    y = np.zeros(IMXSIG)
    f = _lib("lineshape").humliv_bb_
    xp, yp = x.ctypes.data_as(_dp), y.ctypes.data_as(_dp)
    i1, i2 = C.c_int(1), C.c_int(IMXSIG)
    a0, a1, a2 = C.c_double(float(x[IMXSIG // 2] + x0_off)), C.c_double(lw), C.c_double(dw)
    args = (xp, C.byref(i1), C.byref(i2), C.byref(a0), C.byref(a1), C.byref(a2), yp)
    for _ in range(n_warm):
        f(*args)
    t = np.empty(n_calls)
    for i in range(n_calls):
        t0 = time.perf_counter()
        f(*args)
        t[i] = time.perf_counter() - t0
    return dict(median_us=float(np.median(t) * 1e6), min_us=float(t.min() * 1e6), n_calls=int(n_calls),
                params=dict(window=IMXSIG, step=step, centre=centre, x0_offset=x0_off, lw=lw, dw_over_sqrt_ln2=dw,
                            ry=lw / dw))
