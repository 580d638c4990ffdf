"""Host-side mirror of the reference's `spect_main_module` coefficient layer.

make_abscoeff_isomolec keeps the reference's signature and return types
(spect_main_module.py:1880-2131) for the direct, `useLUTs=False` route; instead of
calc_shapes_lines + LutSet.add_PT per (P,T) + a pickle round trip + the
population-weighted combine, it makes ONE call into the HIP engine, which
returns the abs/emi coefficient spectra of every LOS step.
"""
import copy
import os
import pickle
import time

import numpy as np

from . import engine
from . import spect_classes as spcl

n_threads = 4  # spect_main_module.py:27 (signature compatibility)


def prepare_spe_grid(wn_range, sp_step=5.e-4, units='cm_1'):
    """spect_main_module.py:1262-1272"""
    spoffo = np.arange(wn_range[0], wn_range[1] + sp_step / 2, sp_step, dtype=float)
    spect_grid = spcl.SpectralGrid(spoffo, units=units)
    return spcl.SpectralObject(np.zeros(len(spect_grid.grid), dtype=float), spect_grid, units=units)


class AbsSetLOS(object):
    """Set of abs / emi coefficient spectra along a LOS (spect_main_module.py:1179-1258): kept in memory
    (.set, add_set) or streamed through a pickle file (prepare_export / add_dump / finalize_IO, then
    prepare_read / read_one), the reference's way of saving RAM on long LOS.  read_one() serves both.
    .device holds the same data as one CUDA tensor [n_steps, n_grid] for consumers that stay on the GPU."""

    def __init__(self, filename=None, spectral_grid=None, indices=None):
        self.indices = indices if indices is not None else []
        self.counter = 0
        self.remaining = 0
        self.filename = filename
        self.temp_file = None
        self.set = []
        self.spectral_grid = spectral_grid
        self.device = None

    def _open(self, mode):
        if self.filename is None:
            raise ValueError('ERROR!: NO filename set for LutSet.')
        self.temp_file = open(self.filename, mode)

    def prepare_export(self):
        self._open('wb')
        if self.spectral_grid is not None:
            pickle.dump(self.spectral_grid, self.temp_file, protocol=-1)

    def add_dump(self, set_, no_spectral_grid=True):
        if no_spectral_grid:
            for obj in (set_.values() if type(set_) is dict else [set_]):
                obj.erase_grid()
        pickle.dump(set_, self.temp_file, protocol=-1)
        self.counter += 1

    def finalize_IO(self):
        self.temp_file.close()
        self.temp_file = None

    def add_set(self, set_):
        self.set.append(set_)
        self.counter += 1

    def prepare_read(self, read_spectral_grid=True):
        self.remaining = self.counter
        if self.set:            # in memory
            return
        self._open('rb')
        if read_spectral_grid:
            self.spectral_grid = pickle.load(self.temp_file)

    def read_one(self):
        if self.set:
            if self.remaining <= 0:
                self.remaining = self.counter
            set_ = self.set[self.counter - self.remaining]
            self.remaining -= 1
            return set_
        if self.temp_file is None:
            self.prepare_read()
        set_ = pickle.load(self.temp_file)
        for obj in (set_.values() if type(set_) is dict else [set_]):
            obj.restore_grid(self.spectral_grid, link_grid=True)
        self.remaining -= 1
        return set_


def find_free_name(filename, maxnum=1000, split_at='.'):
    """First of filename, name_001.ext, name_002.ext, ... that does not exist yet (spect_main_module.py:34-51): a
    second table built on the same day must not overwrite the first."""
    form = '_{:02d}' if maxnum <= 100 else ('_{:03d}' if maxnum <= 1000 else '_{:05d}')
    num, fileorig = 1, filename
    ind = fileorig.index(split_at)
    while os.path.isfile(filename):
        filename = fileorig[:ind] + form.format(num) + fileorig[ind:]
        num += 1
        if num > maxnum:
            raise ValueError('Check filenames! More than {} with the same name'.format(maxnum))
    return filename


def date_stamp():
    t = time.ctime().split()
    return '_' + t[2] + '-' + t[1] + '-' + t[4]          # spect_main_module.py:30-32


def lut_name(mol, iso, LTE):
    return 'LUT_mol{:02d}_iso{:1d}_{}'.format(mol, iso, 'LTE' if LTE else 'nonLTE')   # :659-664


ctypes_G = ['sp_emission', 'ind_emission', 'absorption']


def _as_lineset(lines, spectral_grid, isomolec):
    """engine.LineSet of the iso-molecule's lines on the grid (lines: SpectLine list or a LineSet)."""
    if isinstance(lines, engine.LineSet):
        return lines
    lines = [lin for lin in lines if lin.Mol == isomolec.mol and lin.Iso == isomolec.iso]
    levels = [getattr(isomolec, lev) for lev in isomolec.levels]
    return engine.LineSet(spcl.lines_to_soa(lines, isomolec), spectral_grid.grid, isomolec.mol, isomolec.iso,
                          isomolec.MM, [lv.energy for lv in levels])


class LutSet(object):
    """Look-up table entry of ONE level of an iso-molecule: the three G spectra at every tabulated (P, T)
    (spect_main_module.py:841-1176).  The table lives in HBM, .device = CUDA float64 [3, n_PT, n_grid]
    (ctype order of ctypes_G); .sets[i][ctype] are host views created on demand.  The reference streams the
    same content through pickle files to save RAM; filename is kept for that interface (add_dump / load)."""

    def __init__(self, mol, iso, MM, level=None, filename=None, level_index=None):
        self.mol, self.iso, self.MM = mol, iso, MM
        self.level = copy.deepcopy(level)
        self.unidentified_lines = level is None
        self.level_index = level_index        # index into the engine's level table (-1 / None: every line)
        self.filename = filename
        self.filenames = [filename]
        self.spectral_grid = None
        self.PTcouples = None
        self.device = None
        self.temp_file = None
        self._host = {}

    # ---- table content ----
    def _append(self, g3, PT):
        import torch
        g3 = g3 if g3.dim() == 3 else g3[:, None, :]
        self.device = g3 if self.device is None else torch.cat([self.device, g3], dim=1)
        if self.PTcouples is None:
            self.PTcouples = []
        self.PTcouples += [list(pt) for pt in PT]
        self._host = {}

    def add_PT(self, spectral_grid, lines, Pres, Temp, keep_memory=False, control=True, n_threads=n_threads):
        """Adds one (P, T) couple (spect_main_module.py:1122-1168).  lines: an engine.LineSet -- one
        sr_gcoeff_layers_dev call -- or SpectLine objects processed by spcl.calc_shapes_lines, which go
        through SpectralGcoeff.BuildCoeff(preCalc_shapes=True) as in the reference (per-line drop-in route)."""
        import torch
        if self.spectral_grid is None:
            self.spectral_grid = copy.deepcopy(spectral_grid)
        if isinstance(lines, engine.LineSet):
            lev = -1 if (self.unidentified_lines or self.level_index is None) else self.level_index
            g3 = lines.gcoeff_layers([Temp], [Pres], level=0 if lines.level_energies.size == 0 else lev)
        else:
            mls = '' if self.unidentified_lines else self.level.minimal_level_string()
            rows = []
            for ctype in ctypes_G:
                gigi = spcl.SpectralGcoeff(ctype, spectral_grid, self.mol, self.iso, self.MM, mls,
                                           unidentified_lines=self.unidentified_lines, link_grid=True)
                rows.append(gigi.BuildCoeff(lines, Temp, Pres, preCalc_shapes=True, n_threads=n_threads))
            g3 = torch.as_tensor(np.array(rows), device='cuda')[:, None, :]
        if self.temp_file is not None:
            self.add_dump(self._host_set(g3[:, 0].cpu().numpy(), Pres, Temp, grid=False))
        self._append(g3, [[Pres, Temp]])

    def _host_set(self, rows, Pres, Temp, grid=True):
        mls = '' if self.unidentified_lines else self.level.minimal_level_string()
        return {ct: spcl.SpectralGcoeff(ct, self.spectral_grid, self.mol, self.iso, self.MM, mls,
                                        unidentified_lines=self.unidentified_lines, spectrum=np.array(rows[c]),
                                        Pres=Pres, Temp=Temp, link_grid=True) for c, ct in enumerate(ctypes_G)}

    @property
    def sets(self):
        """[{ctype: SpectralGcoeff}] per PT couple, on the host (copied from HBM on first access)."""
        if 'sets' not in self._host:
            tab = self.device.cpu().numpy() if self.device is not None else np.zeros((3, 0, 0))
            self._host['sets'] = [self._host_set(tab[:, i], pt[0], pt[1]) for i, pt in enumerate(self.PTcouples or [])]
        return self._host['sets']

    def free_memory(self):
        self._host = {}

    def find(self, Pres, Temp):
        if [Pres, Temp] not in self.PTcouples:
            raise ValueError('{} couple not found!'.format([Pres, Temp]))
        return self.PTcouples.index([Pres, Temp])

    # ---- interpolation (spect_main_module.py:997-1066) ----
    def interp_plan(self, Pres, Temp):
        """Table rows and weights of LutSet.calculate for one (P, T): ([i1, i2, i3, i4], [wP1, wP2, wT1, wT2])."""
        from . import spect_base_module as sbm
        Ps = np.unique(np.array([PT[0] for PT in self.PTcouples]))
        Ts = np.unique(np.array([PT[1] for PT in self.PTcouples]))
        order_t = np.argsort(np.abs(Ts - Temp))
        T1, T2 = Ts[np.argmin(np.abs(Ts - Temp))], Ts[order_t[1]]
        wt = sbm.weight(Temp, T1, T2, itype='lin')
        if Pres <= np.min(Ps):
            P1 = np.min(Ps)
            return [self.find(P1, T1), self.find(P1, T2), -1, -1], [0.0, 0.0, wt[0], wt[1]]
        if Pres <= np.max(Ps):
            P1, P2 = Ps[np.argmin(np.abs(Ps - Pres))], Ps[np.argsort(np.abs(Ps - Pres))[1]]
            wp = sbm.weight(Pres, P1, P2, itype='lin')
            return ([self.find(P1, T1), self.find(P1, T2), self.find(P2, T1), self.find(P2, T2)],
                    [wp[0], wp[1], wt[0], wt[1]])
        raise ValueError('Extrapolating in P')

    def _plan(self, Press, Temps):
        plans = [self.interp_plan(P, T) for P, T in zip(Press, Temps)]
        return (np.array([p[0] for p in plans], np.int32).reshape(-1, 4),
                np.array([p[1] for p in plans], np.float64).reshape(-1, 4))

    def calculate_steps(self, Press, Temps):
        """Interpolated G spectra at every (Press[i], Temps[i]): CUDA [3, n_steps, n_grid]."""
        return engine.lut_interp(self.device, *self._plan(Press, Temps))

    def combine_steps(self, Press, Temps, pops, abs_dev, emi_dev):
        """abs += pop (Gabs - Gind), emi += pop Gsp at every step (spect_main_module.py:2073-2080)."""
        idx, w = self._plan(Press, Temps)
        engine.lut_interp(self.device, idx, w, pops=pops, out=(abs_dev, emi_dev))

    def calculate(self, Pres, Temp):
        """{ctype: SpectralGcoeff} at (Pres, Temp), spect_main_module.py:997-1066."""
        g3 = self.calculate_steps([Pres], [Temp])[:, 0].cpu().numpy()
        return self._host_set(g3, Pres, Temp)

    # ---- file streaming of the reference (RAM workaround), optional ----
    def prepare_export(self, PTcouples, spectral_grid):
        if self.filename is None:
            raise ValueError('ERROR!: NO filename set for LutSet.')
        self.temp_file = open(self.filename, 'wb')
        self.spectral_grid = spectral_grid
        pickle.dump([list(pt) for pt in PTcouples], self.temp_file, protocol=-1)

    def add_dump(self, set_):
        for obj in set_.values():
            obj.erase_grid()
        pickle.dump(set_, self.temp_file, protocol=-1)

    def finalize_IO(self):
        if self.temp_file is not None:
            self.temp_file.close()
        self.temp_file = None

    def load_from_file(self, load_just_PT=False, spectral_grid=None):
        """Reads a stream written by prepare_export / add_PT back into HBM."""
        import torch
        with open(self.filename, 'rb') as fh:
            PT = pickle.load(fh)
            if load_just_PT:
                self.PTcouples = PT
                return
            rows = [[pickle.load(fh)[ct].spectrum for ct in ctypes_G] for _ in PT]
        if spectral_grid is not None and self.spectral_grid is None:
            self.spectral_grid = spectral_grid
        self.device, self.PTcouples, self._host = None, None, {}
        self._append(torch.as_tensor(np.array(rows, dtype=float).transpose(1, 0, 2).copy(), device='cuda'), PT)


class LookUpTable(object):
    """Look-up table of one iso-molecule: one LutSet per level in non-LTE, one 'all' set in LTE
    (spect_main_module.py:682-838); built on the GPU, resident in HBM."""

    def __init__(self, isomolec, wn_range, LTE):
        self.tag = lut_name(isomolec.mol, isomolec.iso, LTE)
        self.wn_range = copy.deepcopy(wn_range)
        self.mol, self.iso, self.MM = isomolec.mol, isomolec.iso, isomolec.MM
        self.isomolec = copy.deepcopy(isomolec)
        self.sets = dict()
        self.PTcouples = []
        self.LTE = LTE

    def make(self, spectral_grid, lines, PTcouples, export_levels=True, cartLUTs=None, control=True,
             n_threads=n_threads, pt_batch=64):
        """G spectra of every level at every [P, T] of PTcouples (spect_main_module.py:718-788): the
        reference's loop over PT couples x levels x ctypes is one sr_gcoeff_levels_dev call per batch of pt_batch
        couples (non-LTE: all levels at once), or one sr_gcoeff_layers_dev call per batch for the LTE 'all' set.
        cartLUTs: when given, every LutSet is also streamed to a file there."""
        self.PTcouples = copy.deepcopy(PTcouples)
        self.spectral_grid = copy.deepcopy(spectral_grid)
        lineset = _as_lineset(lines, spectral_grid, self.isomolec)
        Ps = np.array([pt[0] for pt in PTcouples], float)
        Ts = np.array([pt[1] for pt in PTcouples], float)
        if not self.LTE:
            todo = [(lev, getattr(self.isomolec, lev), i) for i, lev in enumerate(self.isomolec.levels)]
        else:
            todo = [('all', None, -1 if self.isomolec.levels else 0)]
        sets = []
        for name, level, index in todo:
            fn = None if cartLUTs is None else find_free_name(                                      # :737
                cartLUTs + self.tag + '_' + (name if level is not None else 'alllev') + date_stamp() + '.pic', maxnum=10, split_at='.pic')
            st = LutSet(self.mol, self.iso, self.MM, level=level, filename=fn, level_index=index)
            st.spectral_grid = self.spectral_grid
            if fn is not None:
                st.prepare_export(PTcouples, self.spectral_grid)
            sets.append((name, st, index, fn))
        # Round 6: ALL levels of a batch of couples from one multi-channel pass (engine.LineSet.gcoeff_levels: every line
        # evaluated once, its three G-weighted shapes added to the spectra of its two levels) instead of one
        # sr_gcoeff_layers_dev call per level -- the reference's loop order (:759-772: couples outside, levels inside)
        all_levels = (not self.LTE) and lineset.level_energies.size == len(todo)
        for b0 in range(0, len(PTcouples), pt_batch):
            sl = slice(b0, b0 + pt_batch)
            g_all = lineset.gcoeff_levels(Ts[sl], Ps[sl]) if all_levels else None
            for name, st, index, fn in sets:
                g3 = g_all[index] if g_all is not None else lineset.gcoeff_layers(Ts[sl], Ps[sl], level=index)
                if fn is not None:
                    gh = g3.cpu().numpy()
                    for i, pt in enumerate(PTcouples[sl]):
                        st.add_dump(st._host_set(gh[:, i], pt[0], pt[1], grid=False))
                st._append(g3, PTcouples[sl])
        for name, st, index, fn in sets:
            st.finalize_IO()
            self.sets[name] = st

    def CPU_time_estimate(self, lines, PTcouples):
        """The reference's own estimate for ITS path, minutes (spect_main_module.py:791-801)."""
        n_lin = len([lin for lin in lines if (lin.Mol == self.mol and lin.Iso == self.iso)])
        return n_lin * 3. / 30000. * len(PTcouples)

    def find_lev(self, lev_string):
        for lev in self.sets.keys():
            if self.sets[lev].level is not None and self.sets[lev].level.equiv(lev_string):
                return True, lev
        return False, None

    def merge(self, LUT):
        """One table from two with different PT couples and the same levels (spect_main_module.py:699-716)."""
        if self.wn_range != LUT.wn_range:
            raise ValueError('Incompatible LUTs, different wn_ranges: {} {}'.format(self.wn_range, LUT.wn_range))
        for lev_name in self.sets.keys():
            lev1, lev2 = self.sets[lev_name], LUT.sets[lev_name]
            if not self.LTE and not lev1.level.equiv(lev2.level.lev_string):
                raise ValueError('Levels are different, cannot merge LUTs')
            lev1._append(lev2.device, lev2.PTcouples)
        self.PTcouples += LUT.PTcouples


def calc_PT_couples_atmosphere(lines, molecs, atmosphere, pres_step_log=0.4, temp_step=5.0, max_pres=None, thres=0.01,
                               add_lowpres=True):
    """The [P, T] couples a look-up table needs to cover `atmosphere` (.pres [hPa], .temp [K] arrays):
    pressure levels on a log grid, at each level the temperatures met within one level either side, rounded
    outwards to temp_step; below the pressure where the most-broadened line is Doppler dominated
    (lw < thres * dw) one low-pressure level stands for all (spect_main_module.py:1746-1844)."""
    import math as mt
    pres, temp = np.asarray(atmosphere.pres, float), np.asarray(atmosphere.temp, float)
    top = np.log(np.max(pres)) if max_pres is None else np.log(max_pres)
    log_hi = mt.ceil(top / pres_step_log) * pres_step_log
    log_lo = mt.floor(np.log(np.min(pres)) / pres_step_log) * pres_step_log
    pressures = np.exp(log_lo + np.arange(0, (log_hi - log_lo) + 0.5 * pres_step_log, pres_step_log))
    windows = [(-np.inf, pressures[1])] + list(zip(pressures[:-2], pressures[2:])) + [(pressures[-2], pressures[-1])]
    PTcouples = []
    for p, (lo, hi) in zip(pressures, windows):
        tt = temp[(pres >= lo) & (pres <= hi)]
        t_0 = (np.floor(np.min(tt) / temp_step) - 1) * temp_step
        t_1 = (np.ceil(np.max(tt) / temp_step) + 1) * temp_step
        PTcouples += [[p, t] for t in np.arange(t_0, t_1 + 0.5 * temp_step, temp_step)]
    mms = []
    for mol in (molecs if type(molecs) is list else [molecs]):
        if hasattr(mol, 'all_iso'):
            mms += [getattr(mol, isom).MM for isom in mol.all_iso]
        else:
            mms.append(mol.MM)
    widest = lines[int(np.argmax(np.array([lin.Air_broad for lin in lines])))]
    keep, temps_lowpres, pres_0 = [], [], 1.e-8
    for Pres, Temp in PTcouples:
        dw, lw, _ = widest.CheckWidths(Temp, Pres, min(mms))
        if lw < thres * dw:
            pres_0 = max(pres_0, Pres)
            if Temp not in temps_lowpres:
                temps_lowpres.append(Temp)
        else:
            keep.append([Pres, Temp])
    for Temp in temps_lowpres:
        keep.insert(0, [pres_0, Temp])
    if add_lowpres:
        for Temp in temps_lowpres:
            keep.insert(0, [np.exp(log_lo), Temp])
    return keep


def makeLUT_nonLTE_Gcoeffs(spectral_grid, lines, isomol, LTE, atmosphere=None, PTcouples=None, cartLUTs=None,
                           pres_step_log=0.4, temp_step=5.0, save_LUTs=True, n_threads=n_threads, test=False, thres=0.01,
                           max_pres=None, check_num_couples=False):
    """spect_main_module.py:1847-1877: the look-up table of isomol over the atmosphere's (P, T) couples."""
    if PTcouples is None:
        PTcouples = calc_PT_couples_atmosphere(lines, isomol, atmosphere, pres_step_log=pres_step_log,
                                               temp_step=temp_step, max_pres=max_pres, thres=thres)
    if check_num_couples:
        return PTcouples
    if test:
        PTcouples = PTcouples[:10]
    LUT = LookUpTable(isomol, spectral_grid.wn_range(), LTE)
    LUT.make(spectral_grid, lines, PTcouples, export_levels=True, cartLUTs=cartLUTs if save_LUTs else None,
             n_threads=n_threads)
    return LUT


def _populations(isomolec, levels, Temps, LTE):
    """[n_levels or 1, n_steps]: exp(-c2 E_L / Tvib_L) / Q(T), or 1 / Q(T) for the 'all' set
    (spect_main_module.py:2049-2073)."""
    Q = np.atleast_1d(spcl.CalcPartitionSum(isomolec.mol, isomolec.iso, temp=np.asarray(Temps, float)))
    if not levels:
        return (1 / Q)[None, :]
    return np.array([spcl.Boltz_ratio_nodeg(lv.energy, np.asarray(Temps, float) if LTE else lv.local_vibtemp) / Q
                     for lv in levels])


def make_abscoeff_isomolec(wn_range_tot, isomolec, Temps, Press, LTE=True, allLUTs=None, useLUTs=False,
                           lines=None, store_in_memory=False, tagLOS=None, cartDROP=None, track_levels=None,
                           n_threads=n_threads, lineset=None, to_host=True):
    """Absorption and emission coefficients of `isomolec` at every (Press[i], Temps[i])
    (spect_main_module.py:1880-2131).  Non-LTE: every level of isomolec.levels carries .local_vibtemp (one
    value per step).

    useLUTs=False: lines (SpectLine list) or lineset (an engine.LineSet built earlier from the same lines and
    grid) -- ONE coefficient op on the GPU instead of calc_shapes_lines + LutSet.add_PT per (P, T), a pickle
    round trip and the population-weighted combine.  useLUTs=True: allLUTs[(isomolec.mol_name, isomolec.iso)]
    is a LookUpTable; its G spectra are interpolated to every step and combined on the GPU (:1992-2017).
    track_levels: level names whose own share of the coefficients is returned too.

    Returns (abs_coeffs, emi_coeffs) or, with track_levels, (abs_coeffs, emi_coeffs, emi_coeffs_tracked,
    abs_coeffs_tracked): AbsSetLOS with .device (CUDA [n_steps, n_grid]) and, when to_host, one
    SpectralObject per step -- in .set, or in the pickle stream `cartDROP + 'abscoeff_' + tagLOS...` when
    store_in_memory is True (the reference's name for spilling to disk; it forces it for more than 10 steps
    to save RAM, which is not done here: read_one() / prepare_read() serve the in-memory set the same way)."""
    import torch
    try:
        len(Press)
        len(Temps)
    except TypeError:
        Press, Temps = [Press], [Temps]
    Temps, Press = np.asarray(Temps, float), np.asarray(Press, float)
    levels = [getattr(isomolec, lev) for lev in isomolec.levels]
    if track_levels is not None:
        for lev in track_levels:
            if lev not in isomolec.levels:
                raise ValueError('level {} is not a level of mol {} iso {}'.format(lev, isomolec.mol, isomolec.iso))
    tracked = {}
    if useLUTs:
        LUTs = allLUTs[(isomolec.mol_name, isomolec.iso)]
        spectral_grid = LUTs.spectral_grid
        n_grid = len(spectral_grid.grid)
        ab = torch.zeros((len(Temps), n_grid), dtype=torch.float64, device='cuda')
        em = torch.zeros_like(ab)
        pops = _populations(isomolec, levels, Temps, LTE)
        if not levels:
            LUTs.sets['all'].combine_steps(Press, Temps, pops[0], ab, em)
        for li, (lev, levello) in enumerate(zip(isomolec.levels, levels)):
            ok, lev_lut = LUTs.find_lev(levello.lev_string)
            if not ok:
                raise ValueError('mol {} iso {} Level {} not found'.format(isomolec.mol, isomolec.iso, levello.lev_string))
            LUTs.sets[lev_lut].combine_steps(Press, Temps, pops[li], ab, em)
            if track_levels is not None and lev in track_levels:
                ta, te = torch.zeros_like(ab), torch.zeros_like(em)
                LUTs.sets[lev_lut].combine_steps(Press, Temps, pops[li], ta, te)
                tracked[lev] = (ta, te)
    else:
        if lineset is None and lines is None:
            raise ValueError('when calling smm.make_abscoeff_isomolec() with useLUTs = False, you need to give '
                             'the list of spectral lines of isomolec as input')   # spect_main_module.py:1965
        spectral_grid = prepare_spe_grid(wn_range_tot).spectral_grid
        if lineset is None:
            lineset = _as_lineset(lines, spectral_grid, isomolec)                  # :1968 filter inside
        tvib = None
        if levels and not LTE:
            tvib = np.array([lv.local_vibtemp for lv in levels], dtype=float)      # :2065
        # The reference evaluates the shapes once per DISTINCT (P, T) of the step list only in its LUT routes; its direct
        # route repeats them per step (:1974-1990).  Here: when at least three steps share every (P, T) row on average
        # (a 3-D path: kinetic state on (latitude box, altitude), vibrational temperatures by the local SZA), the
        # level-factored route -- per-level pair spectra on the distinct rows + one population-weighted combine
        # (:2036-2106), engine.LevelFactored -- replaces the folded op over the steps.  Same numbers to ~1e-13.
        T_rows, P_rows, step_row = engine.LevelFactored.unique_rows(Temps, Press)
        if len(T_rows) * 3 <= len(Temps):
            ab, em = engine.LevelFactored(lineset, T_rows, P_rows).steps(step_row, tvib=tvib)
        else:
            ab, em = lineset.abscoeff_layers(Temps, Press, tvib=tvib)
        for lev in (track_levels or []):
            tracked[lev] = lineset.abscoeff_level(Temps, Press, isomolec.levels.index(lev), tvib=tvib)

    tagLOS = 'LOS' if tagLOS is None else tagLOS
    if store_in_memory and to_host:
        if cartDROP is None:
            cartDROP = 'stuff_' + date_stamp()
            if not os.path.exists(cartDROP):
                os.mkdir(cartDROP)
            cartDROP += '/'
    name = lambda kind, extra='': None if not (store_in_memory and to_host) else \
        cartDROP + kind + '_' + tagLOS + '_mol_{}_iso_{}{}.pic'.format(isomolec.mol, isomolec.iso, extra)

    def fill(aset, dev):
        aset.device = dev
        if not to_host:
            return aset
        host = dev.cpu().numpy()
        if store_in_memory:
            aset.prepare_export()
        for row in host:
            obj = spcl.SpectralObject(row, spectral_grid, units='cm_1', link_grid=True)   # as prepare_spe_grid, :2039
            aset.add_dump(obj) if store_in_memory else aset.add_set(obj)
        if store_in_memory:
            aset.finalize_IO()
        return aset

    abs_coeffs = fill(AbsSetLOS(name('abscoeff'), spectral_grid=spectral_grid), ab)
    emi_coeffs = fill(AbsSetLOS(name('emicoeff'), spectral_grid=spectral_grid), em)
    if track_levels is None:
        return abs_coeffs, emi_coeffs
    emi_tracked, abs_tracked = dict(), dict()
    for lev in track_levels:
        ta, te = tracked[lev]
        emi_tracked[lev] = fill(AbsSetLOS(name('tracklevel_emicoeff', '_' + lev), spectral_grid=spectral_grid), te)
        # sic: the reference stores the level's EMISSION coefficient in abs_coeffs_tracked too
        # (spect_main_module.py:2097, 2106); the level's absorption share is kept beside it as .true_abs
        abs_tracked[lev] = fill(AbsSetLOS(name('tracklevel_abscoeff', '_' + lev), spectral_grid=spectral_grid), te)
        # always an AbsSetLOS carrying .device (never written to a file: the reference has no such file)
        tset = AbsSetLOS(None, spectral_grid=spectral_grid)
        tset.device = ta
        if to_host and not store_in_memory:
            for row in ta.cpu().numpy():
                tset.add_set(spcl.SpectralObject(row, spectral_grid, units='cm_1', link_grid=True))
        abs_tracked[lev].true_abs = tset
    return abs_coeffs, emi_coeffs, emi_tracked, abs_tracked


def make_abscoeff_LUTS_fast(spectral_grid, isomolec, Temps, Press, LTE=True, tagLOS=None, allLUTs=None, cartDROP=None,
                            store_in_memory=False, track_levels=None, time_control=False, to_host=True):
    """The in-memory LUT route of the fast retrieval (spect_main_module.py:2134-2299): same result as
    make_abscoeff_isomolec(useLUTs=True) on the LUT's own grid; (None, None) when the iso-molecule has no
    table in this spectral range (:2159-2164)."""
    if allLUTs[(isomolec.mol_name, isomolec.iso)] is None:
        return (None, None) if track_levels is None else (None, None, None, None)
    return make_abscoeff_isomolec(None, isomolec, Temps, Press, LTE=LTE, allLUTs=allLUTs, useLUTs=True,
                                  store_in_memory=store_in_memory, tagLOS=tagLOS, cartDROP=cartDROP,
                                  track_levels=track_levels, to_host=to_host)


# ----------------------------------------------------------------------------
# optimal-estimation algebra (spect_main_module.py:3399-3469)
# ----------------------------------------------------------------------------
def genvec(obs, sims, noise, masks=None):
    """Concatenate observation / simulation / noise spectra (optionally masked), spect_main_module.py:3399-3423."""
    cat = lambda objs: np.concatenate([np.asarray(o.spectrum, dtype=float) for o in objs])
    obs_vec, sim_vec, noi_vec = cat(obs), cat(sims), cat(noise)
    if masks is not None:
        masktot = np.concatenate([np.asarray(m, dtype=bool) for m in masks])
        obs_vec, sim_vec, noi_vec = obs_vec[masktot], sim_vec[masktot], noi_vec[masktot]
    return obs_vec, sim_vec, noi_vec


def chicalc(obs, sims, noise, masks, n_ret):
    """Reduced chi square, spect_main_module.py:3426-3431."""
    obs_vec, sim_vec, noi_vec = genvec(obs, sims, noise, masks=masks)
    return np.sum(((obs_vec - sim_vec) / noi_vec) ** 2) / (len(obs_vec) - n_ret)


def inversion_algebra(obs, sims, noise, bayes_set, lambda_LM=0.1, L1_reg=False, masks=None):
    """One Levenberg-Marquardt step of the Bayesian optimal estimation, spect_main_module.py:3433-3469:
    dx = (K^T Sy^-1 K + Sa^-1 + lambda diag(.))^-1 (K^T Sy^-1 (y - F) + Sa^-1 (xa - x)); stores the
    averaging kernel and the retrieval covariance in bayes_set.  Small dense algebra (n_par ~ 10-50),
    host side as in the reference; S_y is diagonal, so it is applied as a row scaling instead of
    inverting an n_obs x n_obs matrix."""
    jac = np.asarray(bayes_set.build_jacobian(masks=masks), dtype=float)
    obs_vec, sim_vec, noi_vec = genvec(obs, sims, noise, masks=masks)
    inversion_algebra_arrays(jac, obs_vec, sim_vec, noi_vec, bayes_set, lambda_LM=lambda_LM, L1_reg=L1_reg)


def inversion_algebra_arrays(jac, obs_vec, sim_vec, noi_vec, bayes_set, lambda_LM=0.1, L1_reg=False, Sa_inv=None):
    """inversion_algebra on the vectors themselves (jac [n_obs, n_par] as build_jacobian returns it, the vectors as
    genvec does): the retrieval loop keeps its spectra as arrays and wraps them into spectrum objects once, at the
    end (retrieval.inversion_fast_limb).  Sa_inv: the inverse a-priori covariance when the caller kept it (it
    does not change between iterations)."""
    xi = np.asarray(bayes_set.param_vector(), dtype=float)
    x_ap = np.asarray(bayes_set.apriori_vector(), dtype=float)
    if Sa_inv is None:
        Sa_inv = np.linalg.inv(np.asarray(bayes_set.VCM_apriori(), dtype=float))
    KtSy = jac.T / noi_vec ** 2.0
    G_inv = KtSy @ jac
    S_inv = G_inv + Sa_inv
    LM = np.diag(np.diag(S_inv))
    S_x = np.linalg.inv(S_inv)
    AVK = S_x @ G_inv
    rhs = KtSy @ (obs_vec - sim_vec) + Sa_inv @ (x_ap - xi)
    deltax = np.linalg.solve(S_inv + lambda_LM * LM, rhs)
    bayes_set.update_params(deltax)
    bayes_set.store_avk(AVK)
    bayes_set.store_VCM(S_x)


# ----------------------------------------------------------------------------
# retrieval parameter space (spect_main_module.py:169-665) and FOV integration
# (N2; spect_main_module.py:3342-3374).  Host-side bookkeeping around the GPU forward model: the VMR
# profile of a gas is sum_p mask_p(z) * x_p, so the absorber columns are linear in the parameters and
# engine.radiance_jacobian (sr_radiance_jac_dev) returns d(radiance)/dx_p from dcol[s][p].
# The reference builds its masks on spect_base_module.AtmGrid / AtmGridMask, which are not in the
# tree; GridMask is the minimal stand-in (coordinates, mask values, interpolation tag).
# ----------------------------------------------------------------------------
class GridMask(object):
    def __init__(self, coords, mask, interp):
        self.grid = np.asarray(coords, dtype=float)
        self.mask = np.asarray(mask, dtype=float)
        self.interp = interp

    def __mul__(self, value):
        return self.mask * value

    __rmul__ = __mul__


def alt_triangle(alt_grid, node_alt, step=None, node_lo=None, node_up=None, first=False, last=False):
    """Triangular weight of one altitude node on alt_grid (spect_main_module.py:319-351): 1 at the
    node, linear to 0 at the neighbouring nodes; the first (last) node keeps weight 1 below (above)."""
    z = np.asarray(alt_grid, dtype=float)
    if step is not None:
        node_lo, node_up = node_alt - step, node_alt + step
    w = np.zeros(z.shape)
    if first:
        up = (z >= node_alt) & (z < node_up)
        w[z < node_alt] = 1.0
        w[up] = 1.0 - (z[up] - node_alt) / (node_up - node_alt)
    elif last:
        lo = (z <= node_alt) & (z > node_lo)
        w[z > node_alt] = 1.0
        w[lo] = 1.0 - (node_alt - z[lo]) / (node_alt - node_lo)
    else:
        up = (z >= node_alt) & (z <= node_up)
        lo = (z < node_alt) & (z >= node_lo)
        w[up] = 1.0 - (z[up] - node_alt) / (node_up - node_alt)
        w[lo] = 1.0 - (node_alt - z[lo]) / (node_alt - node_lo)
    return GridMask(z, w, 'lin')


def lat_box(lat_limits, lat_ok):
    """Box mask over latitude bands that start at lat_limits (spect_main_module.py:354-373): 1 for the
    band holding lat_ok; the last band is open-ended and, as in the reference, excludes its own start."""
    lim = np.asarray(lat_limits, dtype=float)
    w = np.zeros(len(lim))
    w[:-1] = (lat_ok >= lim[:-1]) & (lat_ok < lim[1:])
    w[-1] = lat_ok > lim[-1]
    return GridMask(lim, w, 'box')


def centre_boxes(lat_limits):
    """Band centres of consecutive limits (spect_main_module.py:376-384)."""
    lim = np.asarray(lat_limits, dtype=float)
    return list((lim[:-1] + lim[1:]) / 2.0)


class RetParam(object):
    """One retrieved parameter (spect_main_module.py:587-644)."""

    def __init__(self, nameset, key, maskgrid, apriori, apriori_err, first_guess=None, constrain_positive=True):
        self.nameset, self.key = nameset, key
        self.maskgrid = copy.deepcopy(maskgrid)
        self.value = apriori if first_guess is None else first_guess
        self.apriori, self.apriori_err = apriori, apriori_err
        self.derivatives, self.old_values = [], []
        self.constrain_positive = constrain_positive
        self.not_involved = False
        self.is_used = False

    def set_not_involved(self):
        self.not_involved = True

    def set_involved(self):
        self.not_involved = False

    def set_used(self):
        self.is_used = True

    def update_par(self, delta_par):
        """value += delta; a step that would leave a positive-constrained parameter <= 0 is halved
        until it does not (spect_main_module.py:616-624)."""
        self.old_values.append(self.value)
        if self.constrain_positive:
            while self.value + delta_par <= 0.0:
                delta_par /= 2
        self.value = self.value + delta_par

    def add_hires_deriv(self, derivative):
        self.hires_deriv = copy.deepcopy(derivative)

    def erase_hires_deriv(self):
        self.hires_deriv = None

    def store_deriv(self, derivative, num):
        """Derivative spectrum of observation `num` (replaces an existing entry, else appends)."""
        if 0 <= num < len(self.derivatives):
            self.derivatives[num] = _clone_spectrum(derivative)
        else:
            self.derivatives.append(_clone_spectrum(derivative))


class RetSet(object):
    """Parameters of one quantity, e.g. the nodes of a VMR profile (spect_main_module.py:256-283)."""

    def __init__(self, name, params):
        self.name = name
        self.set = [copy.deepcopy(p) for p in params]
        self.n_par = len(self.set)

    def keys(self):
        return [p.key for p in self.set]

    def items(self):
        return list(zip(self.keys(), self.set))


class LinearProfile_1D_new(RetSet):
    """Profile by linear interpolation between altitude nodes (spect_main_module.py:450-492):
    one RetParam per node, triangular masks on alt_grid."""

    def __init__(self, name, alt_grid, alt_nodes, apriori_prof, apriori_prof_err, first_guess_prof=None):
        z = np.asarray(alt_grid.grid[0] if hasattr(alt_grid, 'grid') else alt_grid, dtype=float)
        nodes = list(alt_nodes)
        fg = apriori_prof if first_guess_prof is None else first_guess_prof
        self.name, self.alts, self.n_par, self.set = name, nodes, len(nodes), []
        for i, node in enumerate(nodes):
            if i == 0:
                mask = alt_triangle(z, node, node_up=nodes[1], first=True)
            elif i == len(nodes) - 1:
                mask = alt_triangle(z, node, node_lo=nodes[-2], last=True)
            else:
                mask = alt_triangle(z, node, node_lo=nodes[i - 1], node_up=nodes[i + 1])
            self.set.append(RetParam(name, node, mask, apriori_prof[i], apriori_prof_err[i], first_guess=fg[i]))

    def profile(self):
        """sum_p mask_p * value_p on the altitude grid."""
        return sum(p.maskgrid * p.value for p in self.set)

    def mask_matrix(self):
        """[n_par, n_alt] weights: d(profile)/d(parameter), the input of the column Jacobian."""
        return np.array([p.maskgrid.mask for p in self.set])

    def check_involved(self, parkey, coord_range):
        """A node is not involved in a path that starts above the next node (spect_main_module.py:482-492)."""
        i = self.alts.index(parkey)
        return i == len(self.alts) - 1 or not coord_range['alt'][0] > self.alts[i + 1]


class LinearProfile_1D(LinearProfile_1D_new):
    """The older constructor (spect_main_module.py:543-585): takes the atmosphere (its altitude grid is
    atmosphere.grid.grid[0]) and -- sic -- gives the MIDDLE nodes the a priori / first guess / error of the node
    before them (the zip at :560 runs over the unsliced profiles)."""

    def __init__(self, name, atmosphere, alt_nodes, apriori_prof, apriori_prof_err, first_guess_prof=None):
        z = np.asarray(atmosphere.grid.grid[0], dtype=float)
        nodes = list(alt_nodes)
        fg = apriori_prof if first_guess_prof is None else first_guess_prof
        self.name, self.alts, self.n_par, self.set = name, nodes, len(nodes), []
        self.set.append(RetParam(name, nodes[0], alt_triangle(z, nodes[0], node_up=nodes[1], first=True),
                                 apriori_prof[0], apriori_prof_err[0], first_guess=fg[0]))
        for j in range(1, len(nodes) - 1):
            self.set.append(RetParam(name, nodes[j], alt_triangle(z, nodes[j], node_lo=nodes[j - 1], node_up=nodes[j + 1]),
                                     apriori_prof[j - 1], apriori_prof_err[j - 1], first_guess=fg[j - 1]))
        self.set.append(RetParam(name, nodes[-1], alt_triangle(z, nodes[-1], node_lo=nodes[-2], last=True),
                                 apriori_prof[-1], apriori_prof_err[-1], first_guess=fg[-1]))
        self.orig_atmosphere = atmosphere


class GridMask2D(object):
    """Mask over (latitude box, altitude): what cos.maskgrid.merge(latbox) builds at spect_main_module.py:412
    (AtmGridMask.merge is in the absent module): the outer product of the latitude-box mask and the altitude mask."""

    def __init__(self, lat_mask, alt_mask):
        self.grid = (lat_mask.grid, alt_mask.grid)
        self.mask = np.outer(lat_mask.mask, alt_mask.mask)
        self.interp = {'lat': lat_mask.interp, 'alt': alt_mask.interp}

    def __mul__(self, value):
        return self.mask * value

    __rmul__ = __mul__


class LinearProfile_2D(RetSet):
    """Altitude nodes x latitude boxes (spect_main_module.py:389-447): one LinearProfile_1D_new per box, every
    node's altitude mask merged with the box mask; keys are (lat, alt_node)."""

    def __init__(self, name, atmosphere, alt_nodes, lat_limits, apriori_profs, apriori_prof_errs, first_guess_profs=None):
        self.name, self.set = name, []
        self.n_par = len(alt_nodes) * len(lat_limits)
        self.alts, self.lats = list(alt_nodes), list(lat_limits)
        z = np.asarray(atmosphere.grid.coords['alt'], dtype=float)
        for apriori_prof, apriori_prof_err, lat in zip(apriori_profs, apriori_prof_errs, lat_limits):
            one_d = LinearProfile_1D_new(name, z, alt_nodes, apriori_prof, apriori_prof_err)   # sic: the first guess
            latbox = lat_box(lat_limits, lat)                                                   # is not passed on (:409)
            for cos in one_d.set:
                self.set.append(RetParam(name, (lat, cos.key), GridMask2D(latbox, cos.maskgrid), cos.apriori, cos.apriori_err))

    def profile(self):
        """[n_lat, n_alt]: sum of mask x value."""
        return sum(p.maskgrid * p.value for p in self.set)

    def check_involved(self, parkey, coord_range):
        """spect_main_module.py:430-447: not involved when the path starts above the next altitude node or lies
        outside the parameter's latitude box."""
        i = self.alts.index(parkey[1])
        involved = i == len(self.alts) - 1 or not coord_range['alt'][0] > self.alts[i + 1]
        latz = coord_range['lat']
        j = self.lats.index(parkey[0])
        if j == len(self.lats) - 1:
            if latz[1] < parkey[0]:
                involved = False
        elif latz[1] < parkey[0] or latz[0] > self.lats[j + 1]:
            involved = False
        return involved


class BayesSet(object):
    """The full parameter space of a retrieval: ordered RetSets (spect_main_module.py:169-253)."""

    def __init__(self, tag=None):
        self.tag = tag
        self.sets, self.order, self.old_params = dict(), [], []
        self.n_tot = 0

    def add_set(self, set_):
        self.sets[set_.name] = copy.deepcopy(set_)
        self.order.append(set_.name)
        self.n_tot += set_.n_par

    def params(self):
        return [p for name in self.order for p in self.sets[name].set]

    def values(self):
        return [p.value for p in self.params()]

    def param_vector(self):
        return np.array(self.values())

    def apriori_vector(self):
        return np.array([p.apriori for p in self.params()])

    def VCM_apriori(self):
        return np.diag(np.array([p.apriori_err for p in self.params()], dtype=float) ** 2)

    def n_used_par(self):
        return sum(p.is_used for p in self.params())

    def build_jacobian(self, masks=None):
        """[n_obs_total, n_tot]: per parameter the derivative spectra of all observations, concatenated
        (and masked like the observation vector of genvec)."""
        rows = [np.concatenate([np.asarray(d.spectrum, dtype=float) for d in p.derivatives]) for p in self.params()]
        jac = np.array(rows)
        if masks is not None:
            jac = jac[:, np.concatenate([np.asarray(m, dtype=bool) for m in masks])]
        self.jacobian = jac.T
        return self.jacobian

    def update_params(self, delta_x):
        self.old_params.append(self.values())
        for p, d in zip(self.params(), delta_x):
            p.update_par(d)

    def store_avk(self, av_kernel):
        self.av_kernel = np.array(av_kernel, copy=True)   # (the reference deep-copies: a private array is the same thing)

    def store_VCM(self, VCM):
        self.VCM = np.array(VCM, copy=True)

    def update_parerror(self):
        for i, p in enumerate(self.params()):
            p.ret_error = np.sqrt(self.VCM[i, i])


def retrieval_converged(chi, chi_old, chi_threshold=0.01):
    """Stopping rule of the retrieval loop (spect_main_module.py:2960-2973): relative change of the
    reduced chi square below the threshold, or chi square increased.  Returns '' (continue),
    'converged' or 'raised'."""
    if chi_old is None:
        return ''
    if abs(chi - chi_old) / chi_old < chi_threshold:
        return 'converged'
    return 'raised' if chi > chi_old else ''


def _clone_spectrum(obj, spectrum=None):
    """A private copy of a spectrum holder: its own spectrum array, the (immutable) grid shared.  The reference
    deep-copies here (spect_main_module.py:3372, 640-645); python's deepcopy of these small objects was 2.5 of the 5.5 ms
    of a retrieval iteration."""
    out = object.__new__(type(obj))          # (copy.copy's __reduce_ex__ round trip was a third of the call)
    out.__dict__.update(obj.__dict__)
    out.spectrum = np.array(obj.spectrum if spectrum is None else spectrum, dtype=float, copy=True)
    return out


def make_group_observations(pixels, alt_step=50., alt_first_los=None):
    """The set of tangent altitudes a group of pixels with similar geometry is simulated on, a regular ladder of
    step alt_step from the lowest pixel's lower LOS to (just beyond) the highest pixel's upper LOS
    (spect_main_module.py:3290-3338).  Sorts `pixels` by tangent altitude in place, like the reference.  The pixels
    need limb_tg_alt and the three lines of sight (low_LOS() / LOS() / up_LOS() objects with get_tangent_altitude(), or
    los_alts() of the stand-in retrieval.LimbPixel).  Returns (alts, mean) with mean = {lat, lon, sza} of the pixels
    that carry limb_tg_lat / limb_tg_lon / limb_tg_sza (the reference builds its LineOfSight objects, absent module,
    from these means and the first pixel's spacecraft position and sub-solar point)."""
    pixels.sort(key=lambda x: x.limb_tg_alt)                                              # :3298

    def tg_alts(pix):
        if hasattr(pix, "los_alts"):
            return pix.los_alts()
        return [lo.get_tangent_altitude() for lo in (pix.low_LOS(), pix.LOS(), pix.up_LOS())]
    lo0, _, up0 = tg_alts(pixels[0])
    first = lo0 if not lo0 > pixels[0].limb_tg_alt else up0                               # :3301-3303
    loN, _, upN = tg_alts(pixels[-1])
    last = upN if not upN < pixels[-1].limb_tg_alt else loN                               # :3304-3306
    alt_range = [first, last]
    if alt_first_los is None or alt_first_los > alt_range[0]:                             # :3309-3312
        alt_first_los = alt_range[0]
    alts = np.arange(alt_first_los, alt_range[1] + alt_step, alt_step)                    # :3329
    mean = {}
    for key, attr in (("lat", "limb_tg_lat"), ("lon", "limb_tg_lon"), ("sza", "limb_tg_sza")):
        vals = [getattr(pi, attr) for pi in pixels if hasattr(pi, attr)]
        if len(vals) == len(pixels):
            mean[key] = np.mean(vals)                                                      # :3322-3324
    return alts, mean


def make_radtran_spline(alts, radtrans):
    """Interpolation of simulated LOS spectra to an arbitrary tangent altitude: the reference's
    RectBivariateSpline(alts, grid, spectra, kx=2, ky=2) (spect_main_module.py:3377-3396), evaluated on the spectra's own
    grid -- there the tensor spline is, column by column, FITPACK's quadratic interpolating spline in altitude.
    radtrans: spectrum objects (.spectrum, .spectral_grid.grid) or an array [n_alts, n_grid] with `grid` taken as its
    index.  Returns f(x) -> the interpolated spectrum (object like radtrans[0], or an array for array input); the whole
    grid in one evaluation (the reference evaluates point by point: the same numbers)."""
    from scipy.interpolate import RectBivariateSpline as spline2D
    alts = np.array(alts, dtype=float)
    as_arrays = isinstance(radtrans, np.ndarray)
    spectrums = np.asarray(radtrans, dtype=float) if as_arrays else np.array([rad.spectrum for rad in radtrans])
    grid = np.arange(spectrums.shape[1], dtype=float) if as_arrays else np.asarray(radtrans[0].spectral_grid.grid, dtype=float)
    intens_spl = spline2D(alts, grid, spectrums, kx=2, ky=2)

    def radtran_alt(x):
        res = np.array(intens_spl(float(x), grid)).reshape(-1)
        if as_arrays:
            return res
        res_spe = copy.deepcopy(radtrans[0])
        res_spe.spectrum = res
        return res_spe

    return radtran_alt


def fov_closed_form(s0, s1, s2, pixel_rot=0.0):
    """The closed form of FOV_integr_1D on arrays of any (common) shape: the spectra of the three lines of sight -- or
    stacks of them, e.g. a pixel's radiances and all its parameter derivatives at once (the integral is linear in them)."""
    if np.ndim(pixel_rot) > 0:
        # one rotation per leading index of the stacks (pixels with different rotations in one pass): the same elementwise
        # operations with the geometry factors as [n_pix, 1, ...] columns, the edge term where there is an edge
        rot = np.abs(np.deg2rad(np.asarray(pixel_rot, dtype=float))).reshape((-1,) + (1,) * (np.ndim(s1) - 1))
        dmax = np.sqrt(2.0) / 2.0 * np.cos(np.pi / 4 - rot)
        delta = dmax - np.sin(rot)
        esse = 1.0 / np.cos(rot)
        c = (s0 + s2 - 2.0 * s1) / (2.0 * dmax ** 2)
        edge = dmax - delta
        total = 2.0 * (s1 * delta + c * delta ** 3 / 3.0)
        has_edge = edge > 1e-14 * dmax
        m2 = dmax * (dmax ** 3 - delta ** 3) / 3.0 - (dmax ** 4 - delta ** 4) / 4.0
        with_edge = total + s1 * edge + 2.0 * c * m2 / np.where(has_edge, edge, 1.0)
        return esse * np.where(has_edge, with_edge, total)
    rot = abs(np.deg2rad(pixel_rot))
    dmax = np.sqrt(2.0) / 2.0 * np.cos(np.pi / 4 - rot)
    delta = dmax - np.sin(rot)
    esse = 1.0 / np.cos(rot)
    c = (s0 + s2 - 2.0 * s1) / (2.0 * dmax ** 2)
    edge = dmax - delta
    total = 2.0 * (s1 * delta + c * delta ** 3 / 3.0)
    if edge > 1e-14 * dmax:
        m2 = dmax * (dmax ** 3 - delta ** 3) / 3.0 - (dmax ** 4 - delta ** 4) / 4.0  # int x^2 (dmax - x)
        total = total + s1 * edge + 2.0 * c * m2 / edge
    return esse * total


def FOV_integr_1D(radtrans, pixel_rot=0.0, closed_form=False):
    """Field-of-view integration over a square pixel rotated by pixel_rot degrees, from the spectra of
    three lines of sight at -dmax, 0, +dmax across it (spect_main_module.py:3342-3374): the spectrum
    is interpolated across the pixel by a degree-2 spline through the three points -- the parabola
    q(x) = s1 + b x + c x^2 -- and integrated with the weight esse for |x| <= delta, falling
    linearly to 0 at |x| = dmax.

    The reference integrates with scipy's quad at its default tolerances (epsabs 1.49e-8).  For
    radiances of ~1e-6 that absolute tolerance is met by the first 21-point Gauss-Kronrod pass, which
    does not resolve the kinks of the weight at +-delta: the reference's value is 2.5e-4 below the
    integral for a 20-degree rotation.  Parity is with the reference, so the default path runs the
    same quadrature on the same integrand; closed_form=True returns the integral itself,
    I = esse [ 2 (s1 delta + c delta^3/3) + s1 e + 2 c (dmax (dmax^3-delta^3)/3 - (dmax^4-delta^4)/4)/e ],
    e = dmax - delta (the odd term of q drops out)."""
    s0, s1, s2 = (np.asarray(r.spectrum, dtype=float) for r in radtrans)
    if closed_form:
        return _clone_spectrum(radtrans[0], fov_closed_form(s0, s1, s2, pixel_rot))
    rot = abs(np.deg2rad(pixel_rot))
    dmax = np.sqrt(2.0) / 2.0 * np.cos(np.pi / 4 - rot)
    delta = dmax - np.sin(rot)
    esse = 1.0 / np.cos(rot)
    b = (s2 - s0) / (2.0 * dmax)
    c = (s0 + s2 - 2.0 * s1) / (2.0 * dmax ** 2)
    edge = dmax - delta
    from scipy import integrate

    def weighted(x, j):
        q = s1[j] + x * (b[j] + x * c[j])
        return q * esse if abs(x) <= delta else q * esse * abs(dmax - abs(x)) / edge

    spet_fov = np.array([integrate.quad(weighted, -dmax, dmax, args=(j,))[0] for j in range(len(s1))])
    return _clone_spectrum(radtrans[0], spet_fov)
