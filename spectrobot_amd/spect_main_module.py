"""Host-side mirror of the reference's `spect_main_module` coefficient layer.

make_abscoeff_isomolec keeps the reference's signature and return types
(spect_main_module.py:1880-2131) for the direct, `useLUTs=False` route; instead of
calc_shapes_lines + LutSet.add_PT per (P,T) + a pickle round trip + the
population-weighted combine, it makes ONE call into the HIP engine, which
returns the abs/emi coefficient spectra of every LOS step.
"""
import copy

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
    xi = np.asarray(bayes_set.param_vector(), dtype=float)
    obs_vec, sim_vec, noi_vec = genvec(obs, sims, noise, masks=masks)
    S_ap = np.asarray(bayes_set.VCM_apriori(), dtype=float)
    x_ap = np.asarray(bayes_set.apriori_vector(), dtype=float)
    KtSy = jac.T / noi_vec ** 2.0
    G_inv = KtSy @ jac
    Sa_inv = np.linalg.inv(S_ap)
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
            self.derivatives[num] = copy.deepcopy(derivative)
        else:
            self.derivatives.append(copy.deepcopy(derivative))


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
        self.av_kernel = copy.deepcopy(av_kernel)

    def store_VCM(self, VCM):
        self.VCM = copy.deepcopy(VCM)

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
    rot = abs(np.deg2rad(pixel_rot))
    dmax = np.sqrt(2.0) / 2.0 * np.cos(np.pi / 4 - rot)
    delta = dmax - np.sin(rot)
    esse = 1.0 / np.cos(rot)
    s0, s1, s2 = (np.asarray(r.spectrum, dtype=float) for r in radtrans)
    b = (s2 - s0) / (2.0 * dmax)
    c = (s0 + s2 - 2.0 * s1) / (2.0 * dmax ** 2)
    edge = dmax - delta
    if closed_form:
        total = 2.0 * (s1 * delta + c * delta ** 3 / 3.0)
        if edge > 1e-14 * dmax:
            m2 = dmax * (dmax ** 3 - delta ** 3) / 3.0 - (dmax ** 4 - delta ** 4) / 4.0  # int x^2 (dmax - x)
            total = total + s1 * edge + 2.0 * c * m2 / edge
        spet_fov = esse * total
    else:
        from scipy import integrate

        def weighted(x, j):
            q = s1[j] + x * (b[j] + x * c[j])
            return q * esse if abs(x) <= delta else q * esse * abs(dmax - abs(x)) / edge

        spet_fov = np.array([integrate.quad(weighted, -dmax, dmax, args=(j,))[0] for j in range(len(s1))])
    out = copy.deepcopy(radtrans[0])
    out.spectrum = spet_fov
    return out
