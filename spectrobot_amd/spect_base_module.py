"""Minimal stand-ins for the `spect_base_module` (sbm) objects the hot path takes.

The reference imports sbm everywhere but does not contain it, so
these classes are defined from the call-site contract alone: what
make_abscoeff_isomolec reads from an iso-molecule and its levels
(spect_main_module.py:1909-1953, 2059-2073; spect_classes.py:135-142).
"""
import numpy as np


def isclose(a, b, rtol=1e-9, atol=0.0):
    return bool(np.isclose(a, b, rtol=rtol, atol=atol))


class Level(object):
    """A vibrational level: .energy (cm-1), .lev_string, .minimal_level_string(),
    .local_vibtemp (one vibrational temperature per LOS step, spect_main_module.py:2065)."""

    def __init__(self, lev_string, energy, local_vibtemp=None):
        self.lev_string = lev_string
        self.energy = float(energy)
        self.local_vibtemp = None if local_vibtemp is None else np.asarray(local_vibtemp, dtype=float)

    def minimal_level_string(self):
        return self.lev_string.strip()

    def add_local_vibtemp(self, temp):
        self.local_vibtemp = np.asarray(temp, dtype=float)


class IsoMolec(object):
    """.mol .iso .MM .mol_name .levels (attribute names, in order) and one attribute per level."""

    def __init__(self, mol, iso, MM, mol_name='', ratio=1.0):
        self.mol, self.iso, self.MM = int(mol), int(iso), float(MM)
        self.mol_name = mol_name
        self.ratio = ratio
        self.levels = []
        self.is_in_LTE = True

    def add_level(self, lev_string, energy, local_vibtemp=None):
        name = 'lev_{:02d}'.format(len(self.levels))
        setattr(self, name, Level(lev_string, energy, local_vibtemp))
        self.levels.append(name)
        return name
