"""Host-side mirror of the reference's `spect_main_module` coefficient layer.

make_abscoeff_isomolec keeps the reference's signature and return types
(spect_main_module.py:1880-2131) for the direct, `useLUTs=False` route; instead of
calc_shapes_lines + LutSet.add_PT per (P,T) + a pickle round trip + the
population-weighted combine, it makes ONE call into the HIP engine, which
returns the abs/emi coefficient spectra of every LOS step.
"""
import numpy as np

from . import engine
from . import spect_classes as spcl

n_threads = 4  # spect_main_module.py:27 (signature compatibility)


def prepare_spe_grid(wn_range, sp_step=5.e-4, units='cm_1'):
    """spect_main_module.py:1262-1272"""
    spoffo = np.arange(wn_range[0], wn_range[1] + sp_step / 2, sp_step, dtype=float)
    spect_grid = spcl.SpectralGrid(spoffo, units=units)
    return spcl.SpectralObject(np.zeros(len(spect_grid.grid), dtype=float), spect_grid)


class AbsSetLOS(object):
    """Set of abs / emi coefficient spectra along a LOS (spect_main_module.py:1179-1258).
    Kept in memory (.set); the reference's pickle streaming is a RAM workaround of its
    CPU path and is not mirrored.  .device holds the same data as one CUDA tensor
    [n_steps, n_grid] for consumers that stay on the GPU (radiance recursion)."""

    def __init__(self, filename=None, spectral_grid=None, indices=None):
        self.indices = indices if indices is not None else []
        self.counter = 0
        self.remaining = 0
        self.filename = filename
        self.set = []
        self.spectral_grid = spectral_grid
        self.device = None

    def add_set(self, set_):
        self.set.append(set_)
        self.counter += 1

    def read_one(self):
        set_ = self.set[self.counter - self.remaining] if self.remaining else self.set[0]
        self.remaining = max(self.remaining - 1, 0)
        return set_

    def prepare_read(self, read_spectral_grid=True):
        self.remaining = self.counter


def make_abscoeff_isomolec(wn_range_tot, isomolec, Temps, Press, LTE=True, allLUTs=None, useLUTs=False,
                           lines=None, store_in_memory=False, tagLOS=None, cartDROP=None, track_levels=None,
                           n_threads=n_threads, lineset=None, to_host=True):
    """Absorption and emission coefficients of `isomolec` at every (Press[i], Temps[i])
    (spect_main_module.py:1880-2131, useLUTs=False route).  Non-LTE: every level of
    isomolec.levels carries .local_vibtemp (one value per step).

    Returns (abs_coeffs, emi_coeffs): AbsSetLOS whose .set holds one SpectralObject per
    step (when to_host) and whose .device is the CUDA tensor [n_steps, n_grid].
    `lineset` may carry an engine.LineSet built earlier from the same lines/grid so
    that the upload is not repeated."""
    if useLUTs:
        raise NotImplementedError('the LUT route (interpolation of stored G coefficients, '
                                  'spect_main_module.py:997-1066) is a disk cache of the CPU path; '
                                  'the engine recomputes: call with useLUTs=False')
    if track_levels is not None:
        raise NotImplementedError('track_levels is not supported yet')
    try:
        len(Press)
        len(Temps)
    except TypeError:
        Press, Temps = [Press], [Temps]
    if lineset is None and lines is None:
        raise ValueError('when calling smm.make_abscoeff_isomolec() with useLUTs = False, you need to give '
                         'the list of spectral lines of isomolec as input')   # spect_main_module.py:1965
    coso = prepare_spe_grid(wn_range_tot)
    spectral_grid = coso.spectral_grid
    levels = [getattr(isomolec, lev) for lev in isomolec.levels]
    if lineset is None:
        lines = [lin for lin in lines if lin.Mol == isomolec.mol and lin.Iso == isomolec.iso]  # :1968
        soa = spcl.lines_to_soa(lines, isomolec)
        lineset = engine.LineSet(soa, spectral_grid.grid, isomolec.mol, isomolec.iso, isomolec.MM,
                                 [lv.energy for lv in levels])
    tvib = None
    if levels and not LTE:
        tvib = np.array([lv.local_vibtemp for lv in levels], dtype=float)   # :2065
    ab, em = lineset.abscoeff_layers(np.asarray(Temps, float), np.asarray(Press, float), tvib=tvib)
    abs_coeffs = AbsSetLOS(None, spectral_grid=spectral_grid)
    emi_coeffs = AbsSetLOS(None, spectral_grid=spectral_grid)
    abs_coeffs.device, emi_coeffs.device = ab, em
    if to_host:
        abh, emh = ab.cpu().numpy(), em.cpu().numpy()
        for i in range(abh.shape[0]):
            abs_coeffs.add_set(spcl.SpectralObject(abh[i], spectral_grid, link_grid=True))
            emi_coeffs.add_set(spcl.SpectralObject(emh[i], spectral_grid, link_grid=True))
    return abs_coeffs, emi_coeffs
