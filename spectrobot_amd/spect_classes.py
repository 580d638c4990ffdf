"""Host-side mirror of the reference's `spect_classes` interface for the hot path.

Same names, argument meaning and error behaviour as the reference
(spect_classes.py, cited per item) for what the spectral hot path touches:
SpectLine, SpectralGrid, SpectralObject, the width / Einstein / partition-sum
helpers, MakeShape and calc_shapes_lines.  The per-line physics that the
reference evaluates in Python + f2py is evaluated on the GPU:

  * MakeShape / MakeShapeLine        -> lineshape.humliv_bb shim (HIP kernel)
  * calc_shapes_lines + BuildCoeff + make_abscoeff_isomolec
                                     -> one coarse-grained call, see
                                        spect_main_module.make_abscoeff_isomolec
  * CalcPartitionSum                 -> library TIPS-2003 tables + Lagrange

The scalar helpers (Lorenz_width, Doppler_width, Einstein_*, ...) are the
reference's closed-form expressions; they exist for callers that inspect single
lines (CheckWidths, Calc_Gcoeffs) and are not used by the GPU path, which
computes the same quantities in sr_prep_kernel.
"""
import copy
import ctypes as C
import math as mt

import numpy as np

from . import _lib
from ._lib import lib, check, dp
from .compat import lineshape

n_threads = 4           # spect_classes.py:24 (kept for signature compatibility; unused)
imxsig = 13010          # spect_classes.py:27
imxlines = 40000        # spect_classes.py:28
imxsig_long = 2000000   # spect_classes.py:29
T_ref = 296.0           # spect_classes.py:39
hpa_to_atm = 0.00098692326671601  # spect_classes.py:40

# spect_classes.py:44-47 with scipy.constants (CODATA-2018) values written out
h_cgs = 6.62607015e-34 * 1.e7
c_cgs = 299792458.0 * 1.e2
k_cgs = 1.380649e-23 * 1.e7
c2 = h_cgs * c_cgs / k_cgs
_AVOGADRO = 6.02214076e23

cose = ('Mol', 'Iso', 'Freq', 'Strength', 'A_coeff', 'Air_broad', 'Self_broad', 'E_lower', 'T_dep_broad',
        'P_shift', 'Up_lev_str', 'Lo_lev_str', 'Q_num_up', 'Q_num_lo')                    # spect_classes.py:50
cose_hit = cose + ('others', 'g_up', 'g_lo')                                               # spect_classes.py:51
cose_mas = ('Up_lev_id', 'Lo_lev_id', 'Up_lev', 'Lo_lev')                                 # spect_classes.py:53


# ----------------------------------------------------------------------------
# scalar helpers
# ----------------------------------------------------------------------------
def convert_to_atm(Pres, units='hPa'):
    """spect_classes.py:2029-2036"""
    if units == 'hPa':
        return Pres * hpa_to_atm
    raise ValueError('units {} not recognized'.format(units))


def Lorenz_width(Temp, Pres_atm, T_dep_broad, Air_broad, Self_broad=0.0, Self_pres_atm=0.0):
    """spect_classes.py:1967-1974"""
    return (T_ref / Temp) ** T_dep_broad * (Air_broad * (Pres_atm - Self_pres_atm) + Self_broad * Self_pres_atm)


def Doppler_width(Temp, MM, wn_0):
    """spect_classes.py:1976-1986 (half width at half maximum; humliv_bb takes dw/sqrt(ln2))"""
    return wn_0 / c_cgs * mt.sqrt(2 * _AVOGADRO * k_cgs * Temp * mt.log(2.0) / MM)


def Boltz_ratio_nodeg(wavenumber, temp):
    """spect_classes.py:1876-1878"""
    return np.exp(-c2 * wavenumber / temp)


def Calc_BB_single(nu, T):
    """spect_classes.py:1895-1903, units erg s-1 cm-2 sr-1 (cm-1)-1"""
    return 2 * h_cgs * c_cgs ** 2 * nu ** 3 / (np.exp(c2 * nu / T) - 1)


def Einstein_A_to_B(A_coeff, wavenumber, units='cm3ergcm2'):
    """spect_classes.py:1736-1754 (only the unit system the path uses)"""
    if units != 'cm3ergcm2':
        raise ValueError("only units='cm3ergcm2' is supported on this path")
    return A_coeff / (2 * h_cgs * c_cgs ** 2 * wavenumber ** 3)


def Einstein_B21_to_B12(B_21, g_1, g_2):
    """spect_classes.py:1778-1785"""
    return B_21 * g_2 / g_1


def Einstein_A_to_Gcoeff_abs(line, Temp, E_vib):
    """spect_classes.py:1806-1819"""
    B_21 = line.Einstein_A_to_B()
    B_12 = Einstein_B21_to_B12(B_21, line.g_lo, line.g_up)
    rot_pop = line.g_lo * Boltz_ratio_nodeg(line.E_lower - E_vib, Temp)
    return h_cgs * c_cgs * line.Freq * rot_pop * B_12 / (4 * np.pi)


def Einstein_A_to_Gcoeff_indem(line, Temp, E_vib):
    """spect_classes.py:1832-1842"""
    B_21 = line.Einstein_A_to_B()
    rot_pop = line.g_up * Boltz_ratio_nodeg(line.E_lower + line.Freq - E_vib, Temp)
    return h_cgs * c_cgs * line.Freq * rot_pop * B_21 / (4 * np.pi)


def Einstein_A_to_Gcoeff_spem(line, Temp, E_vib):
    """spect_classes.py:1845-1853"""
    rot_pop = line.g_up * Boltz_ratio_nodeg(line.E_lower + line.Freq - E_vib, Temp)
    return h_cgs * c_cgs * line.Freq * rot_pop * line.A_coeff / (4 * np.pi)


def Einstein_A_to_LineStrength_hitran(A_coeff, wavenumber, temp, Q_part, g_upper, E_lower, iso_ab=1.0):
    """spect_classes.py:1856-1863"""
    return iso_ab * A_coeff * g_upper * np.exp(-c2 * E_lower / temp) * (1 - np.exp(-c2 * wavenumber / temp)) / (
        8 * np.pi * c_cgs * wavenumber ** 2 * Q_part)


def ImportPartitionSumTable(mol, iso):
    """spect_classes.py:1680-1689 -> gi, T_grid, Q_grid (library TIPS-2003 tables)"""
    from .compat import fparts_mod
    return fparts_mod.bd_tips_2003(mol, iso)


def CalcPartitionSum(mol, iso, temp=296.0):
    """spect_classes.py:1692-1710 (4-point Lagrange through the TIPS-2003 table)"""
    t = np.ascontiguousarray(np.atleast_1d(temp), dtype=np.float64)
    q = np.zeros_like(t)
    check(lib.sr_calc_partition_sum(int(mol), int(iso), t.ctypes.data_as(dp), t.size, q.ctypes.data_as(dp)),
          "CalcPartitionSum")
    return q if np.ndim(temp) else float(q[0])


def CalcPartitionSum_dT(mol, iso, temp=296.0):
    """d Q / d T of the interpolant CalcPartitionSum evaluates (spect_classes.py:1692-1710: the Lagrange polynomial
    through the two table temperatures <= T and the two > T, three points where the table ends), differentiated
    exactly -- the population part of a temperature Jacobian needs (1 / Q)' = -Q' / Q^2 and the reference has no
    temperature derivative of its own.  Piecewise: at a table temperature the right-hand polynomial."""
    gi = C.c_double(0)
    tg, qg = np.zeros(119), np.zeros(119)
    check(lib.sr_bd_tips_2003(int(mol), int(iso), C.byref(gi), tg.ctypes.data_as(dp), qg.ctypes.data_as(dp)), "bd_tips_2003")
    t_all = np.atleast_1d(np.asarray(temp, dtype=np.float64))
    t, inv = np.unique(t_all, return_inverse=True)      # a 3-D path has thousands of steps on a few hundred temperatures
    out = np.zeros_like(t)
    for i, T in enumerate(t):
        n_le = int(np.searchsorted(tg, T, side="right"))
        sel = list(range(max(0, n_le - 2), n_le)) + list(range(n_le, min(len(tg), n_le + 2)))
        xs, qs = tg[sel], qg[sel]
        d = 0.0
        for j in range(len(xs)):          # sum_j q_j l_j'(T),  l_j' = sum_(m != j) prod_(k != j, m) (T - x_k) / prod_(k != j) (x_j - x_k)
            den = np.prod([xs[j] - xs[k] for k in range(len(xs)) if k != j])
            num = sum(np.prod([T - xs[k] for k in range(len(xs)) if k not in (j, m)]) for m in range(len(xs)) if m != j)
            d += qs[j] * num / den
        out[i] = d
    out = out[inv].reshape(t_all.shape)
    return out if np.ndim(temp) else float(out[0])


def closest_grid(wn_arr, wn_0):
    """spect_classes.py:1937-1943 -> (index, grid value) of the closest grid point"""
    ind = np.argmin(np.abs(wn_arr.grid - wn_0))
    return ind, wn_arr.grid[ind]


# ----------------------------------------------------------------------------
# containers
# ----------------------------------------------------------------------------
_C_SI = 299792458.0   # scipy.constants.c (spect_classes.py:409, 429)

# Unit hub of SpectralGrid / SpectralObject conversions (spect_classes.py:395-432, 771-807): every
# conversion goes through nm.  Per unit: (grid -> nm, grid reversed?, spectrum factor given the grid in
# THAT unit) for "to nm", and the same from nm.  The expressions keep the reference's order of operations.
_TO_NM = {
    'nm': (lambda g: g, False, None),
    'mum': (lambda g: g * 1.e3, False, lambda sp, g: sp * 1.e-3),
    'cm_1': (lambda g: 1.e7 / g, True, lambda sp, g: sp * g ** 2 * 1.e-7),
    'hz': (lambda g: _C_SI / (1.e-9 * g), True, lambda sp, g: sp * g ** 2 * 1.e-9 / _C_SI),
}
_FROM_NM = {
    'nm': (lambda g: g, False, None),
    'mum': (lambda g: g * 1.e-3, False, lambda sp, g: sp * 1.e3),
    'cm_1': (lambda g: 1.e7 / g, True, lambda sp, g: sp * g ** 2 * 1.e-7),
    'hz': (lambda g: _C_SI / (1.e-9 * g), True, lambda sp, g: sp * g ** 2 * 1.e-9 / _C_SI),
}


class SpectralGrid(object):
    """spect_classes.py:354-432: grid + units with in-place conversions between 'nm', 'mum', 'cm_1', 'hz'
    (a conversion that inverts the axis also reverses the array, so grids stay ascending)."""

    def __init__(self, spectral_grid, units='nm'):
        self.grid = copy.deepcopy(np.asarray(spectral_grid, dtype=float))
        self.units = units
        if len(spectral_grid) > imxsig_long:
            raise ValueError('Grid longer that the max value imxsig_long set to {}'.format(imxsig_long))

    def _convert(self, units):
        if units not in _FROM_NM:
            raise ValueError('Cannot recognize units {}'.format(units))
        if self.units != 'nm':
            to_nm, rev, _ = _TO_NM[self.units]
            g = to_nm(self.grid)
            self.grid, self.units = (g[::-1].copy() if rev else g), 'nm'
        if units != 'nm':
            from_nm, rev, _ = _FROM_NM[units]
            g = from_nm(self.grid)
            self.grid, self.units = (g[::-1].copy() if rev else g), units
        return self.grid

    def convertto_nm(self):
        return self._convert('nm')

    def convertto_cm_1(self):
        return self._convert('cm_1')

    def convertto_mum(self):
        return self._convert('mum')

    def convertto_hz(self):
        return self._convert('hz')

    def half_precision(self):
        """spect_classes.py:381-386 (float16 grid for saving)."""
        self.grid = self.grid.astype(np.float16)

    def double_precision(self):
        self.grid = self.grid.astype(float)

    def step(self):
        return self.grid[1] - self.grid[0]

    def len_wn(self):
        return len(self.grid)

    def wn_range(self):
        return [self.min_wn(), self.max_wn()]

    def min_wn(self):
        return min(self.grid)

    def max_wn(self):
        return max(self.grid)


class SpectralObject(object):
    """spect_classes.py:435-500, 670-691, 810-830, 929-953: spectrum on a grid with + - * and
    add_to_spectrum / integrate."""

    def __init__(self, spectrum, spectral_grid, direction=None, units='', link_grid=False):
        # link_grid=True shares the caller's grid object (one grid for all the steps of a LOS)
        self.spectral_grid = spectral_grid if link_grid else copy.deepcopy(spectral_grid)
        self.spectrum = np.array(spectrum, dtype=float) if isinstance(spectrum, np.ndarray) else copy.deepcopy(spectrum)
        self.direction = copy.deepcopy(direction)
        self.units = units

    def n_points(self):
        return len(self.spectrum)

    def _combined(self, other, sign):
        """self + sign * other on a copy: same-length operands point by point, a shorter spectrum is
        placed by its grid (add_to_spectrum), anything else is broadcast by numpy."""
        out = copy.deepcopy(self)
        if not isinstance(other, SpectralObject):
            out.spectrum = out.spectrum + sign * other
        elif len(other.spectrum) == len(self.spectrum):
            out.spectrum = out.spectrum + sign * other.spectrum
        else:
            out.add_to_spectrum(other, Strength=sign)
        return out

    def __add__(self, obj2):
        return self._combined(obj2, 1.0)

    def __sub__(self, obj2):
        return self._combined(obj2, -1.0)

    def __mul__(self, obj2):
        out = copy.deepcopy(self)
        out.spectrum = out.spectrum * (obj2.spectrum if isinstance(obj2, SpectralObject) else obj2)
        return out

    def __truediv__(self, obj2):
        """spect_classes.py:493-499 (__div__ of the Python-2 reference)."""
        out = copy.deepcopy(self)
        out.spectrum = out.spectrum / (obj2.spectrum if isinstance(obj2, SpectralObject) else obj2)
        return out

    __div__ = __truediv__

    def __getitem__(self, key):
        """spectrum[(w1, w2)]: the part of the spectrum with w1 - step/2 < grid < w2 + step/2
        (spect_classes.py:451-460); None when empty."""
        half = self.spectral_grid.step() / 2.0
        cond = (self.spectral_grid.grid > key[0] - half) & (self.spectral_grid.grid < key[1] + half)
        if not np.any(cond):
            return None
        out = copy.deepcopy(self)
        out.spectral_grid = SpectralGrid(self.spectral_grid.grid[cond], units=self.units)   # sic: the object's units
        out.spectrum = self.spectrum[cond]
        return out

    def interp_to_grid(self, nugrid):
        """Linear interpolation onto another SpectralGrid, zero outside (spect_classes.py:501-507)."""
        out = copy.deepcopy(self)
        out.spectrum = np.interp(nugrid.grid, self.spectral_grid.grid, self.spectrum, left=0.0, right=0.0)
        out.spectral_grid = copy.deepcopy(nugrid)
        return out

    def max(self):
        return np.max(self.spectrum)

    def min(self):
        return np.min(self.spectrum)

    def convert_grid_to(self, units):
        """Grid AND spectral density to `units` ('nm', 'mum', 'cm_1', 'hz'), in place: the density picks up
        |d old / d new| and the arrays are reversed with the axis (spect_classes.py:757-807).
        Returns (grid, spectrum)."""
        if units not in _FROM_NM:
            raise ValueError('Cannot recognize units {}'.format(units))
        cur = self.spectral_grid.units
        if cur != 'nm':
            _, rev, fac = _TO_NM[cur]
            sp = fac(self.spectrum, self.spectral_grid.grid)
            self.spectrum = sp[::-1] if rev else sp
            self.spectral_grid.convertto_nm()
        if units != 'nm':
            _, rev, fac = _FROM_NM[units]
            sp = fac(self.spectrum, self.spectral_grid.grid)       # grid in nm here, as in the reference
            self.spectrum = sp[::-1] if rev else sp
            self.spectral_grid._convert(units)
        return self.spectral_grid.grid, self.spectrum

    def convertto_nm(self):
        return self.convert_grid_to('nm')

    def convertto_mum(self):
        return self.convert_grid_to('mum')

    def convertto_cm_1(self):
        return self.convert_grid_to('cm_1')

    def convertto_hz(self):
        return self.convert_grid_to('hz')

    def half_precision(self):
        """spect_classes.py:722-737: float32 spectrum (sic), float16 grid."""
        self.spectrum = self.spectrum.astype(np.float32)
        if self.spectral_grid is not None:
            self.spectral_grid.half_precision()

    def double_precision(self):
        self.spectrum = self.spectrum.astype(float)
        if self.spectral_grid is not None:
            self.spectral_grid.double_precision()

    # ---- fine-grained drop-in of the reference's accumulate (the production path never materialises
    # per-line shapes: engine.LineSet.abscoeff_layers / gcoeff_layers) ----
    def prepare_fortran_sum(self, lines, fix_length=imxsig):
        """Rows, init, fin for lineshape.sum_all_lines (spect_classes.py:1100-1147): every line's part
        inside this grid, zero-padded to fix_length on the right -- or on the left, with init shifted,
        when the padded row would pass the end of the grid.  1-based, inclusive."""
        spino = self.spectral_grid.step() / 10.
        g = self.spectral_grid.grid
        rows = np.zeros((len(lines), fix_length))
        init, fin = np.zeros(len(lines), np.int32), np.zeros(len(lines), np.int32)
        for n, line in enumerate(lines):
            lg = line.spectral_grid.grid
            inside = np.flatnonzero((g > lg[0] - spino) & (g < lg[-1] + spino))
            part = line.spectrum[(lg > g[0] - spino) & (lg < g[-1] + spino)]
            first, last = inside[0] + 1, inside[-1] + 1
            pad = fix_length - (last - first + 1)
            if pad <= 0:
                rows[n, :] = line.spectrum
            elif last + pad < self.n_points():
                rows[n, :len(part)] = part
                last += pad
            else:
                rows[n, pad:] = part
                first -= pad
            init[n], fin[n] = first, last
        return rows, init, fin

    def add_lines_to_spectrum(self, lines, Strengths=None, fix_length=imxsig, n_threads=n_threads):
        """spect_classes.py:1016-1097 with the sum on the GPU (compat lineshape.sum_all_lines, Fortran
        summation order).  As in the reference the lines are only added when Strengths is given (:1038-1044)."""
        n_lines = len(lines)
        if n_lines == 0:
            return self.spectrum
        if n_lines > imxlines:
            raise ValueError('{} are too many lines!! Increase the thresold imxlines (now {}) or decrease num of '
                             'lines..'.format(n_lines, imxlines))
        if self.n_points() > imxsig_long:
            raise ValueError('The input spectrum is too long!! Increase the thresold imxsig_long or decrease num of wn..')
        scaled = [] if Strengths is None else [ln.multiply(st, save=False) for ln, st in zip(lines, Strengths)]
        if not scaled:
            return self.spectrum
        rows, init, fin = self.prepare_fortran_sum(scaled, fix_length=fix_length)
        if init.min() < 1:
            raise ValueError('grid shorter than one line window: the reference writes in front of its array here '
                             '(spect_classes.py:1132-1134)')
        self.spectrum = lineshape.sum_all_lines(self.spectrum, rows, init, fin, len(scaled), self.n_points())
        return self.spectrum

    def multiply(self, factor, save=True):
        if save:
            self.spectrum = self.spectrum * factor
            return
        coso = copy.deepcopy(self)
        coso.spectrum = factor * self.spectrum
        return coso

    def integrate(self, w1=None, w2=None):
        cond = ~np.isnan(self.spectrum)
        if w1 is not None:
            cond &= self.spectral_grid.grid >= w1
        if w2 is not None:
            cond &= self.spectral_grid.grid <= w2
        return np.trapezoid(self.spectrum[cond], x=self.spectral_grid.grid[cond])

    def add_to_spectrum(self, spectrum2, Strength=None, sumcheck=10.):
        """spect_classes.py:929-953: spectrum2's grid is (partly) inside self's, same step."""
        spino = self.spectral_grid.step() / 10.
        g1, g2 = self.spectral_grid.grid, spectrum2.spectral_grid.grid
        ok = (g1 > g2[0] - spino) & (g1 < g2[-1] + spino)
        ok2 = (g2 > g1[0] - spino) & (g2 < g1[-1] + spino)
        if Strength is not None:
            self.spectrum[ok] += Strength * spectrum2.spectrum[ok2]
        else:
            self.spectrum[ok] += spectrum2.spectrum[ok2]

    def erase_grid(self):
        self.spectral_grid = None

    def restore_grid(self, spectral_grid, link_grid=False):
        self.spectral_grid = spectral_grid if link_grid else copy.deepcopy(spectral_grid)


class SpectralIntensity(SpectralObject):
    """spect_classes.py:1167-1243: spectral intensity ('ergscm2' | 'Wm2' | 'nWcm2').  hires_to_lowres
    runs the Gaussian-ILS degradation on the GPU."""

    def __init__(self, intensity, spectral_grid, direction=None, units='ergscm2'):
        self.spectrum = copy.deepcopy(intensity)
        self.direction = copy.deepcopy(direction)
        self.spectral_grid = copy.deepcopy(spectral_grid)
        self.units = units

    _TO_WM2 = {'Wm2': 1.0, 'ergscm2': 1.e-3, 'nWcm2': 1.e-5}      # spect_classes.py:1213-1223
    _FROM_WM2 = {'Wm2': 1.0, 'ergscm2': 1.e3, 'nWcm2': 1.e5}      # :1225-1235

    def convertto(self, new_units):
        """In-place conversion between 'Wm2', 'ergscm2', 'nWcm2' through W m-2 (spect_classes.py:1200-1235)."""
        if new_units not in self._FROM_WM2:
            raise ValueError('No method for units ' + new_units)
        if self.units != 'Wm2':
            self.spectrum = self.spectrum * self._TO_WM2[self.units]
            self.units = 'Wm2'
        if new_units != 'Wm2':
            self.spectrum = self.spectrum * self._FROM_WM2[new_units]
            self.units = new_units
        return self.spectrum

    def convertto_Wm2(self):
        return self.convertto('Wm2')

    def convertto_ergscm2(self):
        return self.convertto('ergscm2')

    def convertto_nWcm2(self):
        return self.convertto('nWcm2')

    def add_noise(self, noise):
        self.noise = copy.deepcopy(noise)

    def add_mask(self, mask):
        self.mask = copy.deepcopy(mask)

    def add_bands(self, bands):
        self.bands = copy.deepcopy(bands)

    def hires_to_lowres(self, lowres_obs, spectral_widths=None, keep_original_hires=True):
        """spect_classes.py:1180-1191.  self: hi-res on a cm_1 np.arange grid in 'ergscm2';
        lowres_obs: object with .spectral_grid (units 'nm') and .units; spectral_widths: Gaussian
        sigma per band (nm); default = the low-res grid step, as in the reference (spcl:890-892)."""
        import torch
        from . import engine
        if self.spectral_grid.units != 'cm_1' or self.units != 'ergscm2':
            raise ValueError("hires_to_lowres on the GPU takes a cm_1 grid and 'ergscm2' intensities")
        if lowres_obs.spectral_grid.units != 'nm':
            raise ValueError("the low-resolution grid must be in nm")
        new_len = len(lowres_obs.spectral_grid.grid)
        if spectral_widths is None:
            spectral_widths = [lowres_obs.spectral_grid.step()] * new_len
        elif type(spectral_widths) is int or type(spectral_widths) is float:
            spectral_widths = [spectral_widths] * new_len
        if len(spectral_widths) != new_len:
            raise ValueError('{} spectral widths for {} grid points'.format(len(spectral_widths), new_len))
        dev = torch.as_tensor(np.ascontiguousarray(self.spectrum, dtype=np.float64), device="cuda")
        low = engine.hires_to_lowres(dev, self.spectral_grid.grid, lowres_obs.spectral_grid.grid, spectral_widths,
                                     out_units=lowres_obs.units)[0]
        return SpectralIntensity(low, lowres_obs.spectral_grid, units=lowres_obs.units)


class SpectralGcoeff(SpectralObject):
    """G_abs / G_spem / G_indem spectrum of ONE level of an iso-molecule at one (P, T)
    (spect_classes.py:1247-1375): what the non-LTE look-up tables hold."""

    ctypes = ('sp_emission', 'ind_emission', 'absorption')

    def __init__(self, ctype, spectral_grid, mol, iso, MM, minimal_level_string, unidentified_lines=False,
                 spectrum=None, Pres=None, Temp=None, link_grid=False):
        self.mol, self.iso, self.MM = mol, iso, MM
        self.unidentified_lines = bool(unidentified_lines)
        self.lev_string = None if unidentified_lines else minimal_level_string
        self.ctype = ctype
        self.spectral_grid = spectral_grid if link_grid else copy.deepcopy(spectral_grid)
        self.spectrum = np.zeros(len(spectral_grid.grid), dtype=float) if spectrum is None else spectrum
        self.direction = None
        self.units = ''
        if Pres is not None and Temp is not None:
            self.pres, self.temp = Pres, Temp

    def BuildCoeff(self, lines, Temp, Pres, n_threads=n_threads, preCalc_shapes=False, debug=False, isomolec=None):
        """Sum of G_ctype * shape over the lines of this level: upper level for the two emission types, lower
        level for absorption; every line of the iso-molecule for the 'all' set (spect_classes.py:1277-1337).
        `lines` carry .shape / .G_coeffs (calc_shapes_lines) when preCalc_shapes; otherwise they are computed
        here, which needs `isomolec` (the reference's own call at :1302 is one argument short of :1340).
        This is the per-line drop-in route (GPU shims); LutSet.add_PT with an engine.LineSet is the fast one."""
        if self.ctype not in self.ctypes:
            raise ValueError('ctype has to be one among {}, {} and {}. {} not recognized'.format(*self.ctypes, self.ctype))
        self.temp, self.pres = Temp, Pres
        if len(lines) == 0:
            return self.spectrum
        if preCalc_shapes:
            if not (hasattr(lines[0], 'shape') and hasattr(lines[0], 'G_coeffs')):
                raise ValueError('preCalc_shapes is set as True but the lines do not contain the attribute << shapes >>. '
                                 'Are you sure you precalculated the line shapes? Run calc_shapes_lines on your line '
                                 'set first.')
        else:
            if isomolec is None:
                raise ValueError('BuildCoeff without preCalc_shapes needs isomolec= to run calc_shapes_lines')
            lines = calc_shapes_lines(self.spectral_grid, lines, Temp, Pres, isomolec)
        mine = [lin for lin in lines if lin.Mol == self.mol and lin.Iso == self.iso]
        if not self.unidentified_lines:
            if self.ctype == 'absorption':
                mine = [lin for lin in mine if self.lev_string == lin.minimal_level_string_lo()]
            else:
                mine = [lin for lin in mine if self.lev_string == lin.minimal_level_string_up()]
        if mine:
            self.add_lines_to_spectrum([lin.shape for lin in mine], Strengths=[lin.G_coeffs[self.ctype] for lin in mine],
                                       n_threads=n_threads)
        return self.spectrum

    def interpolate(self, coeff2, Pres=None, Temp=None):
        """Linear interpolation between two G spectra that share T (give Pres) or P (give Temp)
        (spect_classes.py:1349-1375; sbm.weight(.., itype='lin') is in the absent module: linear weights)."""
        from . import spect_base_module as sbm
        if coeff2 is None:
            return None
        same_t, same_p = sbm.isclose(self.temp, coeff2.temp), sbm.isclose(self.pres, coeff2.pres)
        if not same_t and not same_p:
            raise ValueError('The two coeffs have both different temperatures and pressures! cannot interpolate')
        if Pres is not None:
            if not same_t:
                raise ValueError('The two coeffs have different temperatures! You should specify the interpolation '
                                 'temperature, not the pressure')
            w1, w2 = sbm.weight(Pres, self.pres, coeff2.pres, itype='lin')
            new_p, new_t = Pres, self.temp
        elif Temp is not None:
            if not same_p:
                raise ValueError('The two coeffs have different pressures! You should specify the interpolation '
                                 'pressure, not the temperature')
            w1, w2 = sbm.weight(Temp, self.temp, coeff2.temp, itype='lin')
            new_p, new_t = self.pres, Temp
        else:
            raise ValueError('give Pres or Temp')
        return SpectralGcoeff(self.ctype, self.spectral_grid, self.mol, self.iso, self.MM, self.lev_string,
                              unidentified_lines=self.unidentified_lines,
                              spectrum=w1 * self.spectrum + w2 * coeff2.spectrum, Pres=new_p, Temp=new_t)


def Calc_BB(spectral_grid, T, units='ergscm2'):
    """Planck spectrum on a cm_1 grid as a SpectralIntensity (spect_classes.py:1881-1892)."""
    spectrum = 2 * h_cgs * c_cgs ** 2 * (spectral_grid.grid) ** 3 / (np.exp(c2 * spectral_grid.grid / T) - 1)
    bb = SpectralIntensity(spectrum, spectral_grid, units='ergscm2')
    if units != 'ergscm2':
        bb.convertto(units)
    return bb


class SpectLine(object):
    """One HITRAN line (spect_classes.py:56-351).  linea: dict, numpy record or sequence + nomi."""

    def __init__(self, linea, nomi=None):
        if nomi is None:
            if isinstance(linea, dict):
                nomi = tuple(linea.keys())
            elif isinstance(linea, np.void):
                nomi = linea.dtype.names
            else:
                raise ValueError('Missing names for line quantities')
        else:
            linea = dict(zip(nomi, linea))
        for nome in nomi:
            setattr(self, nome, linea[nome])
        for nome in cose_mas:
            setattr(self, nome, None)
        for nome in cose_hit:
            if not hasattr(self, nome):
                setattr(self, nome, None)
        self.E_vib_up = None
        self.E_vib_lo = None

    def minimal_level_string_up(self):
        return self.Up_lev_str.strip() if self.Up_lev_str is not None else None

    def minimal_level_string_lo(self):
        return self.Lo_lev_str.strip() if self.Lo_lev_str is not None else None

    def LinkToMolec(self, isomolec):
        """spect_classes.py:122-150 (including its if/elif: a line whose two levels are the
        same level never gets its lower level linked and is reported unlinked)."""
        if isomolec is None:
            return False
        self.Up_lev_id = self.E_vib_up = self.Lo_lev_id = self.E_vib_lo = None
        for lev in isomolec.levels:
            Level = getattr(isomolec, lev)
            if Level.minimal_level_string() == self.minimal_level_string_up():
                self.Up_lev_id = lev
                self.E_vib_up = Level.energy
            elif Level.minimal_level_string() == self.minimal_level_string_lo():
                self.Lo_lev_id = lev
                self.E_vib_lo = Level.energy
        return not (self.Up_lev_id is None or self.Lo_lev_id is None)

    def Einstein_A_to_B(self):
        return Einstein_A_to_B(self.A_coeff, self.Freq, units='cm3ergcm2')

    def CheckWidths(self, Temp, Pres, MM):
        """spect_classes.py:161-171 -> (dw, lw, p_shift)"""
        Pres_atm = convert_to_atm(Pres, units='hPa')
        return (Doppler_width(Temp, MM, self.Freq),
                Lorenz_width(Temp, Pres_atm, self.T_dep_broad, self.Air_broad), self.P_shift * Pres_atm)

    def MakeShapeLine(self, Temp, Pres, grid=None, MM=None, Strength=1.0, verbose=False, keep_memory=False):
        """spect_classes.py:174-206.  The pressure shift is computed but, as in the
        reference (line 197), not applied; self broadening is not used (line 190)."""
        if MM is None:  # spect_classes.py:178-179: from molparam.txt
            from . import spect_base_module as sbm
            MM = sbm.find_molec_metadata(self.Mol, self.Iso)['iso_MM']
        if grid is None:
            sp_step = 5.e-4
            grid = np.arange(-imxsig * sp_step / 2, imxsig * sp_step / 2, sp_step, dtype=float)
            grid = SpectralGrid(grid + self.Freq, units='cm_1')
        Pres_atm = convert_to_atm(Pres, units='hPa')
        lw = Lorenz_width(Temp, Pres_atm, self.T_dep_broad, self.Air_broad)
        dw = Doppler_width(Temp, MM, self.Freq)
        shape = MakeShape(grid, self.Freq, lw, dw, Strength=Strength)
        if keep_memory:
            self.shape = shape
        return shape

    def Calc_Gcoeffs(self, Temp, isomolec=None):
        """spect_classes.py:312-343 -> {'sp_emission', 'ind_emission', 'absorption'}"""
        ctypes_ = ['sp_emission', 'ind_emission', 'absorption']
        ok = self.LinkToMolec(isomolec)
        lev_energy_lo = self.E_vib_lo if ok else 0.0
        lev_energy_up = self.E_vib_up if ok else 0.0
        if self.A_coeff != 0.0 and self.g_lo != 0.0 and self.g_up != 0.0:
            values = [Einstein_A_to_Gcoeff_spem(self, Temp, lev_energy_up),
                      Einstein_A_to_Gcoeff_indem(self, Temp, lev_energy_up),
                      Einstein_A_to_Gcoeff_abs(self, Temp, lev_energy_lo)]
        else:
            values = [0., 0., 0.]
        self.G_coeffs = dict(zip(ctypes_, values))
        return self.G_coeffs


def MakeShape(wn_arr, wn_0, lw, dw, Strength=1.0):
    """spect_classes.py:1990-2008: unit-area Voigt on the 13010-point window wn_arr
    (humliv_bb evaluated on the GPU)."""
    fac = float(dw * mt.sqrt(np.pi / mt.log(2.0)))
    y = lineshape.humliv_bb(wn_arr.grid, 1, len(wn_arr.grid), wn_0, lw, dw / mt.sqrt(mt.log(2.0)))
    y = Strength * y / fac
    return SpectralObject(y, wn_arr)


# ----------------------------------------------------------------------------
# line databases
# ----------------------------------------------------------------------------
_HITRAN_WIDTHS = (2, 1, 12, 10, 10, 5, 5, 10, 4, 8, 15, 15, 15, 15, 19, 7, 7)   # 160 columns, HITRAN 2004+
_GBB_WIDTHS = (2, 1, 12, 10, 10, 6, 6, 10, 4, 8, 15, 15, 15, 15)


def _num(txt, kind):
    txt = txt.strip()
    if not txt:
        return -1 if kind is int else float('nan')     # what np.genfromtxt yields for an empty field
    try:
        return kind(txt)
    except ValueError:
        return -1 if kind is int else float('nan')


def read_line_database(nome_sp, mol=None, iso=None, up_lev=None, down_lev=None, fraction_to_keep=None,
                       db_format='HITRAN', freq_range=None, n_skip=0, link_to_isomolecs=None, verbose=False):
    """spect_classes.py:1532-1601: fixed-width HITRAN ('HITRAN', 160 columns with g_up / g_lo) or
    MAKE_MW ('gbb') line files -> list of SpectLine.  Same selection rules: mol / iso / level-string
    filters, freq_range (the file is assumed sorted: reading stops past the upper bound), zero
    broadening coefficients replaced by 0.05 (air) and 0.07 (self), fraction_to_keep by line strength."""
    if db_format == 'gbb':
        widths, names = _GBB_WIDTHS, cose
    elif db_format == 'HITRAN':
        widths, names = _HITRAN_WIDTHS, cose_hit
    else:
        raise ValueError('Allowed values for db_format: {}, {}'.format('gbb', 'HITRAN'))
    edges = np.concatenate([[0], np.cumsum(widths)])
    kinds = [int, int] + [float] * 8 + [str] * (5 if db_format == 'HITRAN' else 4) + \
        ([float, float] if db_format == 'HITRAN' else [])
    linee_ok = []
    with open(nome_sp, 'r') as infi:
        if n_skip == -1:
            from . import spect_base_module as sbm
            sbm.trova_spip(infi)               # :1558-1559: the data start behind the header's '#' line
        for _ in range(max(n_skip, 0)):
            infi.readline()
        for raw in infi:
            raw = raw.rstrip('\n').rstrip('\r')
            if not raw.strip():
                continue
            vals = []
            for a, b, kind in zip(edges[:-1], edges[1:], kinds):
                cell = raw[a:b]
                vals.append(cell if kind is str else _num(cell, kind))
            linea = dict(zip(names, vals))
            if freq_range is not None:
                if linea['Freq'] < freq_range[0]:
                    continue
                if linea['Freq'] > freq_range[1]:
                    break
            if (linea['Mol'] == mol or mol is None) and (linea['Iso'] == iso or iso is None) and \
                    (linea['Up_lev_str'] == up_lev or up_lev is None) and \
                    (linea['Lo_lev_str'] == down_lev or down_lev is None):
                line = SpectLine(linea)
                if line.Air_broad == 0.0:
                    line.Air_broad = 0.05      # spect_classes.py:1578-1579
                if line.Self_broad == 0.0:
                    line.Self_broad = 0.07     # spect_classes.py:1580-1581
                if link_to_isomolecs is not None:
                    cand = [m for m in link_to_isomolecs if (m.mol == line.Mol and m.iso == line.Iso)]
                    if len(cand) > 1:
                        raise ValueError('Multiple levels corresponding to line! WTF?')
                    if cand:
                        line.LinkToMolec(cand[0])
                linee_ok.append(line)
    if fraction_to_keep is not None and linee_ok:
        essort = np.sort(np.array([lin.Strength for lin in linee_ok]))[int(fraction_to_keep * (len(linee_ok) - 1))]
        return [lin for lin in linee_ok if lin.Strength >= essort]
    return linee_ok


# ----------------------------------------------------------------------------
# line list -> structure of arrays for the engine
# ----------------------------------------------------------------------------
def lines_to_soa(lines, isomolec=None):
    """SpectLine objects -> the SoA dict the engine uploads.  Level indices are
    resolved here once (the reference string-matches every level for every line
    at every (P,T): LinkToMolec, spect_classes.py:122-150); -1 = not in the
    level list."""
    n = len(lines)
    out = {k: np.zeros(n) for k in ("freq", "a_coeff", "e_lower", "g_up", "g_lo", "air_broad", "t_dep_broad")}
    out["lev_up"] = np.full(n, -1, np.int32)
    out["lev_lo"] = np.full(n, -1, np.int32)
    names = []
    if isomolec is not None:
        names = [getattr(isomolec, lev).minimal_level_string() for lev in isomolec.levels]
    for i, l in enumerate(lines):
        out["freq"][i], out["a_coeff"][i], out["e_lower"][i] = l.Freq, l.A_coeff, l.E_lower
        out["g_up"][i], out["g_lo"][i] = l.g_up, l.g_lo
        out["air_broad"][i], out["t_dep_broad"][i] = l.Air_broad, l.T_dep_broad
        if names:
            up, lo = l.minimal_level_string_up(), l.minimal_level_string_lo()
            # same loop as LinkToMolec: a level that matches the upper string is never
            # also taken as the lower one (if/elif); a later duplicate overrides
            for j, nm in enumerate(names):
                if nm == up:
                    out["lev_up"][i] = j
                elif nm == lo:
                    out["lev_lo"][i] = j
    return out


def calc_shapes_lines(wn_arr, lines, Temp, Pres, isomolec, n_threads=n_threads):
    """spect_classes.py:1378-1415.  Kept for callers that want per-line shapes:
    attaches .shape (13010-point SpectralObject) and .G_coeffs to every line that
    links to the iso-molecule.  The production path does NOT go through per-line
    shapes: use spect_main_module.make_abscoeff_isomolec."""
    if len(isomolec.levels) > 0:
        lines = [lin for lin in lines if lin.LinkToMolec(isomolec)]
    sp_step = wn_arr.step()
    lin_grid = np.arange(-imxsig * sp_step / 2, imxsig * sp_step / 2, sp_step, dtype=float)
    for lin in lines:
        ind_ok, fr_grid_ok = closest_grid(wn_arr, lin.Freq)
        lin_grid_ok = SpectralGrid(lin_grid + fr_grid_ok, units='cm_1')
        lin.MakeShapeLine(Temp, Pres, grid=lin_grid_ok, MM=isomolec.MM, keep_memory=True)
        lin.Calc_Gcoeffs(Temp, isomolec=isomolec)
    return lines
