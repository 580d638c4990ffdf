"""Minimal stand-ins for the `spect_base_module` (sbm) objects the hot path takes.

The reference imports sbm everywhere but does not contain it, so
these classes are defined from the call-site contract alone: what
make_abscoeff_isomolec reads from an iso-molecule and its levels
(spect_main_module.py:1909-1953, 2059-2073; spect_classes.py:135-142).
"""
import numpy as np


def isclose(a, b, rtol=1e-9, atol=0.0):
    return bool(np.isclose(a, b, rtol=rtol, atol=atol))


def weight(x, x1, x2, itype='lin'):
    """(w1, w2) with w1 * f(x1) + w2 * f(x2) the interpolant at x (call sites spect_classes.py:1365, 1371;
    the function itself is in the absent module): linear, or linear in log x for itype='exp'."""
    if itype == 'lin':
        w2 = (x - x1) / (x2 - x1)
    elif itype == 'exp':
        w2 = (np.log(x) - np.log(x1)) / (np.log(x2) - np.log(x1))
    else:
        raise ValueError('itype {} not recognized'.format(itype))
    return 1.0 - w2, w2


class Level(object):
    """A vibrational level: .energy (cm-1), .lev_string, .minimal_level_string(),
    .local_vibtemp (one vibrational temperature per LOS step, spect_main_module.py:2065)."""

    def __init__(self, lev_string, energy, local_vibtemp=None):
        self.lev_string = lev_string
        self.energy = float(energy)
        self.local_vibtemp = None if local_vibtemp is None else np.asarray(local_vibtemp, dtype=float)

    def minimal_level_string(self):
        return self.lev_string.strip()

    def equiv(self, lev_string):
        """Same level as the one labelled lev_string (call site spect_main_module.py:711, 809)."""
        return self.minimal_level_string() == lev_string.strip()

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


# ----------------------------------------------------------------------------
# input files.  Both readers live in the absent spect_base_module; they are written
# from their call sites and from the files they read, which ARE in the reference tree.
# ----------------------------------------------------------------------------
def trova_spip(ifile, hasha='#', read_past=False):
    """Advance an open text file past its header: read lines up to and including the first one that starts with
    `hasha` ('#'); with read_past return the rest of that line.  Call site: read_line_database(n_skip=-1)
    (spect_classes.py:1558-1559); the function itself is in the absent module (unpinned: restated from the call
    site and the author's .dat file convention -- free-text header, one '#' line, data).  A file without such a
    line is an error here, not an endless loop."""
    while True:
        linea = ifile.readline()
        if linea == '':
            raise ValueError("trova_spip: no line starting with {!r} in the file header".format(hasha))
        if linea[:1] == hasha:
            return linea[1:] if read_past else None


def read_molparam(filename):
    """HITRAN `molparam.txt` (the file shipped with the reference): blocks `NAME (mol)` followed by one
    row per isotopologue `code abundance Q(296K) gj molar_mass`.  Returns {(mol, iso): dict}, iso =
    1-based position inside the block (HITRAN's isotopologue number)."""
    table = {}
    mol, name, iso = None, '', 0
    with open(filename) as fh:
        for raw in fh:
            parts = raw.split()
            if len(parts) == 2 and parts[1].startswith('(') and parts[1].endswith(')') and parts[1][1:-1].isdigit():
                name, mol, iso = parts[0], int(parts[1][1:-1]), 0
            elif mol is not None and len(parts) == 5 and parts[0].isdigit():
                iso += 1
                table[(mol, iso)] = dict(mol_name=name, iso_name=parts[0], iso_ratio=float(parts[1]),
                                         Q_296=float(parts[2]), gj=int(parts[3]), iso_MM=float(parts[4]))
    return table


_MOLPARAM = {}


def find_molec_metadata(mol, iso, filename=None):
    """Metadata of isotopologue `iso` of HITRAN molecule `mol`: the keys the reference reads are
    'iso_MM' (spect_classes.py:179) and 'iso_ratio' (spect_classes.py:233, 270, 301).  The table comes
    from `filename`, else from $SPECTROBOT_MOLPARAM (a molparam.txt); there is no built-in copy."""
    import os
    path = filename or os.environ.get('SPECTROBOT_MOLPARAM')
    if path is None:
        raise ValueError('no molparam.txt: pass filename= or set SPECTROBOT_MOLPARAM')
    if path not in _MOLPARAM:
        _MOLPARAM[path] = read_molparam(path)
    try:
        return _MOLPARAM[path][(int(mol), int(iso))]
    except KeyError:
        raise ValueError('molecule {} isotopologue {} not in {}'.format(mol, iso, path))


def read_inputs(filename, keys, n_lines=None, itype=None, defaults=None, verbose=False):
    """`[key]` input files of the drivers (inputs_spect_robot_SAMPLE.in; call sites
    radtran_3D_ch4.py:45-49): a line `[key]` is followed by the value on the next non-empty line
    (n_lines[i] values for list-valued keys).  Values are converted with itype[i]; keys missing from
    the file take defaults[i].  Text outside `[key]` blocks and lines starting with '#' are ignored."""
    keys = list(keys)
    itype = [str] * len(keys) if itype is None else list(itype)
    defaults = [None] * len(keys) if defaults is None else list(defaults)
    n_lines = [1] * len(keys) if n_lines is None else list(n_lines)
    with open(filename) as fh:
        lines = [ln.strip() for ln in fh]
    lines = [ln for ln in lines if ln and not ln.startswith('#')]

    def convert(text, typ):
        if typ is bool:
            if text.lower() in ('true', 't', '1', 'yes'):
                return True
            if text.lower() in ('false', 'f', '0', 'no'):
                return False
            raise ValueError('not a boolean: {!r}'.format(text))
        if isinstance(typ, (list, tuple)):  # several values on one line, one type each
            return [t(v) for t, v in zip(typ, text.split())]
        return typ(text)

    out = {}
    for key, typ, default, n in zip(keys, itype, defaults, n_lines):
        tag = '[{}]'.format(key)
        if tag in lines:
            at = lines.index(tag)
            vals = []
            for ln in lines[at + 1:at + 1 + n]:
                if ln.startswith('[') and ln.endswith(']'):
                    break
                vals.append(convert(ln, typ))
            if len(vals) < n:
                raise ValueError('key {} needs {} value line(s) in {}'.format(key, n, filename))
            out[key] = vals[0] if n == 1 else vals
        else:
            out[key] = default
        if verbose:
            print('{:>16s}: {}'.format(key, out[key]))
    return out
